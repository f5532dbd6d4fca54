"""Generate + build tools/ubench_valu (gfx950 VALU/MFMA issue-rate microbenchmark).

Each kernel runs ITER x UNROLL copies of one instruction on 8 independent
register chains and reports shader cycles per wave-instruction (s_memtime),
for 1 wave/SIMD and for 4 waves/SIMD.  Used to price the int4 dequant sequence.
"""
import os
import subprocess

OPS = {
    # name: (asm template with {d} dst, {a} {b} srcs; register class note)
    "v_fma_f32": "v_fma_f32 {d}, {a}, {b}, {d}",
    "v_mul_f32": "v_mul_f32 {d}, {a}, {d}",
    "v_pk_fma_f32": "v_pk_fma_f32 {D}, {A}, {B}, {D}",
    "v_pk_add_f32": "v_pk_add_f32 {D}, {A}, {D}",
    "v_pk_mul_f32": "v_pk_mul_f32 {D}, {A}, {D}",
    "v_cvt_pk_bf16_f32": "v_cvt_pk_bf16_f32 {d}, {a}, {d}",
    "v_cvt_f32_ubyte0": "v_cvt_f32_ubyte0 {d}, {a}",
    "v_cvt_f32_ubyte1": "v_cvt_f32_ubyte1 {d}, {a}",
    "v_cvt_f32_ubyte3": "v_cvt_f32_ubyte3 {d}, {a}",
    "v_cvt_f32_u32": "v_cvt_f32_u32 {d}, {a}",
    "v_cvt_f32_bf16": "v_cvt_f32_bf16 {d}, {a}",
    "v_lshlrev_b32": "v_lshlrev_b32 {d}, 16, {a}",
    "v_and_b32": "v_and_b32 {d}, 0xffff0000, {a}",
    "v_and_b32_inl": "v_and_b32 {d}, 15, {a}",
    "v_and_or_b32": "v_and_or_b32 {d}, {a}, {b}, {d}",
    "v_bfe_u32": "v_bfe_u32 {d}, {a}, 4, 4",
    "v_bfi_b32": "v_bfi_b32 {d}, {a}, {b}, {d}",
    "v_perm_b32": "v_perm_b32 {d}, {a}, {b}, {d}",
    "v_cndmask_b32": "v_cndmask_b32 {d}, {a}, {b}, vcc",
    "v_add_u32": "v_add_u32 {d}, {a}, {d}",
    "v_lshl_add_u32": "v_lshl_add_u32 {d}, {a}, 2, {d}",
    "v_mul_u32_u24": "v_mul_u32_u24 {d}, {a}, {d}",
    "v_dot2_f32_bf16": "v_dot2_f32_bf16 {d}, {a}, {b}, {d}",
    "v_dot2c_f32_bf16": "v_dot2c_f32_bf16 {d}, {a}, {b}",
    "v_pk_add_f16": "v_pk_add_f16 {d}, {a}, {d}",
    "v_pk_fma_f16": "v_pk_fma_f16 {d}, {a}, {b}, {d}",
    "v_cvt_pk_fp8_f32": "v_cvt_pk_fp8_f32 {d}, {a}, {b}",
    "v_rndne_f32": "v_rndne_f32 {d}, {a}",
    "v_max_f32": "v_max_f32 {d}, {a}, {d}",
    "v_cvt_i32_f32": "v_cvt_i32_f32 {d}, {a}",
    "v_fmac_f32": "v_fmac_f32 {d}, {a}, {b}",
    "v_add_f32": "v_add_f32 {d}, {a}, {d}",
    "v_sub_f32": "v_sub_f32 {d}, {a}, {d}",
    "v_or_b32": "v_or_b32 {d}, {a}, {d}",
    "v_xor_b32": "v_xor_b32 {d}, {a}, {d}",
    "v_mov_b32": "v_mov_b32 {d}, {a}",
    "v_lshrrev_b32": "v_lshrrev_b32 {d}, 4, {a}",
    "v_fmamk_f32": "v_fmamk_f32 {d}, {a}, 0x40490fdb, {d}",
    "v_mad_u32_u24": "v_mad_u32_u24 {d}, {a}, {b}, {d}",
    "v_cvt_pk_f32_fp8": "v_cvt_pk_f32_fp8 {D}, {a}",
    "v_cvt_scalef32_pk_f32_fp8": "v_cvt_scalef32_pk_f32_fp8 {D}, {a}, {b}",
    "v_cvt_scalef32_pk_bf16_fp8": "v_cvt_scalef32_pk_bf16_fp8 {d}, {a}, {b}",
    "v_cvt_scalef32_pk_f32_fp4": "v_cvt_scalef32_pk_f32_fp4 {D}, {a}, {b}",
    "v_cvt_f32_ubyte1_sdwa": "v_cvt_f32_u32_sdwa {d}, {a} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1",
    "v_add_f32_sdwa_word1": "v_add_f32_sdwa {d}, {a}, {d} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD",
    "v_pk_add_f32_opsel": "v_pk_add_f32 {D}, {A}, {D} op_sel_hi:[0,1]",
    "v_dot4_i32_i8": "v_dot4_i32_i8 {d}, {a}, {b}, {d}",
    "v_alignbit_b32": "v_alignbit_b32 {d}, {a}, {b}, 16",
    "v_cvt_f32_f16": "v_cvt_f32_f16 {d}, {a}",
}
MFMA = {
    "v_mfma_f32_16x16x32_bf16": ("v_mfma_f32_16x16x32_bf16 {C4}, {A4}, {B4}, {C4}", 4),
    "v_mfma_f32_32x32x16_bf16": ("v_mfma_f32_32x32x16_bf16 {C16}, {A4}, {B4}, {C16}", 16),
    "v_mfma_f32_4x4x4_16b_bf16": ("v_mfma_f32_4x4x4_16b_bf16 {C4}, {A2}, {B2}, {C4}", 4),
    "v_mfma_f32_16x16x16_bf16": ("v_mfma_f32_16x16x16_bf16 {C4}, {A2}, {B2}, {C4}", 4),
    "v_mfma_i32_32x32x32_i8": ("v_mfma_i32_32x32x32_i8 {C16}, {A4}, {B4}, {C16}", 16),
    "v_mfma_i32_16x16x64_i8": ("v_mfma_i32_16x16x64_i8 {C4}, {A4}, {B4}, {C4}", 4),
    "v_mfma_f32_32x32x16_fp8_fp8": ("v_mfma_f32_32x32x16_fp8_fp8 {C16}, {A2}, {B2}, {C16}", 16),
    "v_mfma_scale_f32_32x32x64_f8f6f4": ("v_mfma_scale_f32_32x32x64_f8f6f4 {C16}, {A8}, {B8}, {C16}, v100, v101 op_sel_hi:[0,0,0]", 16),
    "v_mfma_scale_f32_16x16x128_f8f6f4": ("v_mfma_scale_f32_16x16x128_f8f6f4 {C4}, {A8}, {B8}, {C4}, v100, v101 op_sel_hi:[0,0,0]", 4),
}
UNROLL = 8
CHAINS = 8


def valu_body(tmpl):
    lines = []
    for _ in range(UNROLL):
        for c in range(CHAINS):
            d = f"v{20 + 2 * c}"
            D = f"v[{20 + 2 * c}:{21 + 2 * c}]"
            lines.append(tmpl.format(d=d, a="v4", b="v6", D=D, A="v[4:5]", B="v[6:7]"))
    return lines


def mfma_body(tmpl, nacc):
    lines = []
    chains = 4 if nacc == 16 else 8
    for _ in range(UNROLL):
        for c in range(chains):
            base = 32 + c * nacc
            lines.append(
                tmpl.format(
                    C4=f"v[{base}:{base + 3}]", C16=f"v[{base}:{base + 15}]",
                    A2="v[4:5]", B2="v[6:7]", A4="v[4:7]", B4="v[8:11]", A8="v[4:11]", B8="v[12:19]",
                )
            )
    return lines, chains


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    src = ['#include <hip/hip_runtime.h>', '#include <stdio.h>', '#include <string.h>', '#include <vector>',
           'struct K { const char* name; void (*fn)(unsigned long long*, int); int per_iter; };']
    table = []
    clob = ", ".join(f'"v{i}"' for i in range(4, 128))
    for name, tmpl in list(OPS.items()) + [(n, None) for n in MFMA]:
        if tmpl is None:
            lines, chains = mfma_body(*MFMA[name])
            per_iter = UNROLL * chains
        else:
            lines = valu_body(tmpl)
            per_iter = UNROLL * CHAINS
        body = "\\n\\t".join(lines)
        fn = "k_" + name
        src.append(f'''__global__ __launch_bounds__(1024) void {fn}(unsigned long long* out, int iters) {{
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {{
    asm volatile("{body}" ::: {clob}, "vcc");
  }}
  asm volatile("s_nop 7\\n\\ts_nop 7" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}}''')
        table.append(f'{{"{name}", {fn}, {per_iter}}}')
    valu_lines = "\\n\\t".join(valu_body(OPS["v_pk_fma_f32"]))
    ml, _ = mfma_body(*MFMA["v_mfma_f32_4x4x4_16b_bf16"])
    mfma_lines = "\\n\\t".join(ml)
    ml2, _ = mfma_body(*MFMA["v_mfma_f32_16x16x32_bf16"])
    mfma2_lines = "\\n\\t".join(ml2)
    for nm, ml_ in (("mixed4x4", mfma_lines), ("mixed16x16x32", mfma2_lines)):
        src.append(f"""__global__ __launch_bounds__(1024) void k_{nm}(unsigned long long* out, int iters, int mode) {{
  // mode 0: all waves VALU; 1: all waves MFMA; 2: waves 0-3 VALU + waves 4-7 MFMA (same SIMDs)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool do_mfma = (mode == 1) || (mode == 2 && wave >= 4);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (do_mfma) {{
    for (int i = 0; i < iters; ++i) asm volatile("{ml_}" ::: {clob}, "vcc");
  }} else {{
    for (int i = 0; i < iters; ++i) asm volatile("{valu_lines}" ::: {clob}, "vcc");
  }}
  asm volatile("s_nop 7\\n\\ts_nop 7" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
}}""")
    src.append("static K kernels[] = {" + ", ".join(table) + "};")
    src.append(r'''
int main(int argc, char** argv) {
  unsigned long long* d; hipMalloc(&d, 1 << 16);
  const int iters = 2000;
  printf("%-36s %10s %10s %10s\n", "instruction", "1w/SIMD", "2w/SIMD", "4w/SIMD");
  for (auto& k : kernels) {
    double res[3];
    int cfg[3] = {256, 512, 1024};
    for (int c = 0; c < 3; ++c) {
      hipMemset(d, 0, 1 << 16);
      hipLaunchKernelGGL(k.fn, dim3(1), dim3(cfg[c]), 0, 0, d, iters);   // warm
      hipLaunchKernelGGL(k.fn, dim3(1), dim3(cfg[c]), 0, 0, d, iters);
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(cfg[c] / 64);
      hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
      unsigned long long mx = 0; for (auto v : h) mx = v > mx ? v : mx;
      // cycles per wave-instruction as seen by ONE SIMD: total cycles / (instr per wave * waves per SIMD)
      res[c] = (double)mx / ((double)iters * k.per_iter * (cfg[c] / 256));
    }
    printf("%-36s %10.2f %10.2f %10.2f\n", k.name, res[0], res[1], res[2]);
  }
  for (int which = 0; which < 2; ++which) {
    for (int mode = 0; mode < 3; ++mode) {
      hipMemset(d, 0, 1 << 16);
      for (int r = 0; r < 2; ++r) {
        if (which == 0) hipLaunchKernelGGL(k_mixed4x4, dim3(1), dim3(512), 0, 0, d, iters, mode);
        else hipLaunchKernelGGL(k_mixed16x16x32, dim3(1), dim3(512), 0, 0, d, iters, mode);
      }
      hipDeviceSynchronize();
      unsigned long long h[8]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
      printf("mixed %s mode %d (0=VALU x8 waves,1=MFMA x8,2=4 VALU + 4 MFMA): cycles/iter-block per wave:", which ? "16x16x32" : "4x4x4", mode);
      for (int w = 0; w < 8; ++w) printf(" %.0f", (double)h[w] / iters);
      printf("\n");
    }
  }
  return 0;
}''')
    path = os.path.join(here, "ubench_valu.hip")
    with open(path, "w") as f:
        f.write("\n".join(src))
    exe = os.path.join(here, "ubench_valu")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-o", exe, path], check=True)
    print(exe)


if __name__ == "__main__":
    main()
