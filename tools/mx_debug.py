import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ao_amd import _lib, ops
lib = _lib.lib()
def run(n, k, sizes, tune, variant=0):
    lib.ao_gemm8_set_variant(variant)
    lib.ao_gemm8_set_tuning(9, 0)
    for kv in filter(None, tune.split(",")):
        a, b = kv.split("="); _lib.check(lib.ao_gemm8_set_tuning(int(a), int(b)))
    return ops.mxfp8_grouped_mm(aq, a_s, wq, ws, offs).float().cpu().numpy()
for n, k in ((14336, 4096), (4096, 14336)):
    sizes = [32, 0, 0, 0, 32, 64, 0, 0]
    E, rows = len(sizes), sum(sizes)
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn(rows, k, device="cuda", dtype=torch.bfloat16, generator=g)
    w = torch.randn(E, n, k, device="cuda", dtype=torch.bfloat16, generator=g) * 0.02
    wq, ws = ops.mxfp8_quantize(w); del w
    aq, a_s = ops.mxfp8_quantize(a)
    offs = torch.tensor(np.cumsum(sizes), dtype=torch.int32, device="cuda")
    ref = run(n, k, sizes, "", 113)
    for tune, var in (("9=3", 0), ("9=1", 0), ("", 0), ("", 116), ("9=3", 116)):
        for rep in range(4):
            y = run(n, k, sizes, tune, var)
            bad = np.argwhere(np.abs(y - ref) > 1e-2 * np.abs(ref).max())
            if len(bad):
                tiles = sorted({(int(r) // 64 if r < 64 else (1 if r < 64 else 2), int(c) // 128) for r, c in bad})
                rowsb = sorted({int(r) for r, c in bad}); 
                print(f"N={n} K={k} tune[{tune}] var {var} rep{rep}: {len(bad)} bad elements; rows {rowsb[0]}..{rowsb[-1]} ({len(rowsb)} rows); col tiles {sorted({int(c)//128 for r,c in bad})[:12]}; max err {np.abs(y-ref).max():.3f} ref max {np.abs(ref).max():.3f}")
            else:
                print(f"N={n} K={k} tune[{tune}] var {var} rep{rep}: ok (max diff {np.abs(y-ref).max():.2e})")
