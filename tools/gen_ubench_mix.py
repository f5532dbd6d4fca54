#!/usr/bin/env python3
"""Generate tools/ubench_mix.hip: does VALU op X issue in the shadow of MFMA op Y on one gfx950 SIMD?
Pattern per step: 1 MFMA (rotating over 4 independent accumulators) + k independent VALU ops.
Reports cycles per step for one wave per SIMD and two waves per SIMD (s_memtime)."""
VALU = {
    "pk_fma_f32": "v_pk_fma_f32 v[{d}:{d1}], v[4:5], v[6:7], v[{d}:{d1}]",
    "fma_f32": "v_fma_f32 v{d}, v4, v6, v{d}",
    "mul_f32": "v_mul_f32 v{d}, v4, v6",
    "pk_mul_f32": "v_pk_mul_f32 v[{d}:{d1}], v[4:5], v[6:7]",
    "pk_add_f32": "v_pk_add_f32 v[{d}:{d1}], v[4:5], v[6:7]",
    "cvt_pk_bf16_f32": "v_cvt_pk_bf16_f32 v{d}, v4, v5",
    "cvt_scalef32_pk_f32_fp8": "v_cvt_scalef32_pk_f32_fp8 v[{d}:{d1}], v4, v6",
    "cvt_scalef32_pk_bf16_fp8": "v_cvt_scalef32_pk_bf16_fp8 v{d}, v4, v6",
    "perm_b32": "v_perm_b32 v{d}, v4, v5, v6",
    "and_b32": "v_and_b32 v{d}, v4, v5",
    "lshrrev_b32": "v_lshrrev_b32 v{d}, 4, v5",
    "dot2_f32_bf16": "v_dot2_f32_bf16 v{d}, v4, v5, v{d}",
    "and_or_b32": "v_and_or_b32 v{d}, v4, v5, v6",
}
MFMA = {
    "none": None,
    "4x4x4": "v_mfma_f32_4x4x4_16b_bf16 v[{a}:{a3}], v[8:9], v[10:11], v[{a}:{a3}]",
    "16x16x32": "v_mfma_f32_16x16x32_bf16 v[{a}:{a3}], v[8:11], v[12:15], v[{a}:{a3}]",
    "16x16x16": "v_mfma_f32_16x16x16_bf16 v[{a}:{a3}], v[8:9], v[10:11], v[{a}:{a3}]",
}
KS = [0, 2, 4, 6, 8]
STEPS = 16
clob = ", ".join('"v%d"' % i for i in range(4, 128))
out = ['#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <vector>', '#include <string>',
       'typedef void (*KFn)(unsigned long long*, int);', 'struct K { const char* v; const char* m; int k; KFn fn; };']
kerns = []
for vn, vt in VALU.items():
    for mn, mt in MFMA.items():
        for k in KS:
            if mt is None and k == 0:
                continue
            if k == 0 and vn != "fma_f32":
                continue
            lines = []
            for s in range(STEPS):
                if mt is not None:
                    a = 64 + 4 * (s % 4)
                    lines.append(mt.format(a=a, a3=a + 3))
                for j in range(k):
                    d = 16 + 2 * ((s * k + j) % 16)
                    lines.append(vt.format(d=d, d1=d + 1))
            body = "\\n\\t".join(lines)
            name = f"k_{vn}_{mn}_{k}"
            out.append(f'__global__ __launch_bounds__(512) void {name}(unsigned long long* o, int iters) {{\n'
                       f'  unsigned long long t0 = __builtin_amdgcn_s_memtime();\n'
                       f'  for (int i = 0; i < iters; ++i) asm volatile("{body}" ::: {clob});\n'
                       f'  asm volatile("s_nop 7\\n\\ts_nop 7" ::: "memory");\n'
                       f'  unsigned long long t1 = __builtin_amdgcn_s_memtime();\n'
                       f'  if ((threadIdx.x & 63) == 0) o[threadIdx.x >> 6] = t1 - t0;\n}}')
            kerns.append((vn, mn, k, name))
out.append('static K ks[] = {' + ", ".join(f'{{"{v}", "{m}", {k}, {n}}}' for v, m, k, n in kerns) + '};')
out.append('''
int main() {
  unsigned long long* d; hipMalloc(&d, 4096);
  const int iters = 500;
  printf("%%-28s %%-9s %%2s %%12s %%12s   (cycles per step: 1 MFMA + k VALU)\\n", "valu", "mfma", "k", "1w/SIMD", "2w/SIMD(max)");
  for (auto& k : ks) {
    double r[2];
    int cfg[2] = {256, 512};
    for (int c = 0; c < 2; ++c) {
      hipMemset(d, 0, 4096);
      hipLaunchKernelGGL(k.fn, dim3(1), dim3(cfg[c]), 0, 0, d, iters);
      hipLaunchKernelGGL(k.fn, dim3(1), dim3(cfg[c]), 0, 0, d, iters);
      hipDeviceSynchronize();
      unsigned long long h[8]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
      unsigned long long mx = 0; for (int w = 0; w < cfg[c] / 64; ++w) mx = h[w] > mx ? h[w] : mx;
      r[c] = (double)mx / (iters * %d);
    }
    printf("%%-28s %%-9s %%2d %%12.2f %%12.2f\\n", k.v, k.m, k.k, r[0], r[1]);
  }
  return 0;
}
''' % STEPS)
open(__file__.replace("gen_ubench_mix.py", "ubench_mix.hip"), "w").write("\n".join(out))
print(len(kerns), "kernels")
