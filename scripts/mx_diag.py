"""GPU diagnostic: error structure of the mxfp8 grouped mm vs the numpy oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ao_amd import ops
from oracle import mx_ref as MX
from oracle import bf16 as B

DEV = "cuda"
def randn_bf16(shape, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    v = (rng.standard_normal(shape) * scale).astype(np.float32)
    return torch.from_numpy(v).to(torch.bfloat16)

for sizes in ([16, 16, 16, 16], [128, 0, 0, 0]):
    E, N, K = len(sizes), 64, 512
    M = sum(sizes)
    a = randn_bf16((M, K), 21); w = randn_bf16((E, N, K), 22, 0.1)
    offs = torch.tensor(np.cumsum(sizes), dtype=torch.int32)
    a_d, a_s = ops.mxfp8_quantize(a.to(DEV), "rceil")
    w_d, w_s = ops.mxfp8_quantize(w.to(DEV), "rceil")
    y = ops.mxfp8_grouped_mm(a_d, a_s, w_d, w_s, offs.to(DEV))
    an, asn = a_d.view(torch.uint8).cpu().numpy(), a_s.view(torch.uint8).cpu().numpy()
    wn, wsn = w_d.view(torch.uint8).cpu().numpy(), w_s.view(torch.uint8).cpu().numpy()
    y_ref = MX.grouped_mm(an, asn, wn, wsn, offs.numpy())
    yn = y.float().cpu().numpy()
    d = np.abs(yn - y_ref)
    i = np.unravel_index(np.argmax(d / (np.abs(y_ref) * 2.0**-7 + 1e-5)), d.shape)
    print("sizes", sizes, "rel", np.linalg.norm(yn - y_ref) / np.linalg.norm(y_ref), "max abs diff", d.max(),
          "worst elem", i, "y", yn[i], "ref", y_ref[i], "n_bad", int((d > np.abs(y_ref) * 2.0**-7 + 1e-5).sum()), "of", d.size)
    # exact f64 value for the worst element, before bf16 rounding
    A = MX.mx_dequant_bf16(an, asn).astype(np.float64)
    e = int(np.searchsorted(np.cumsum(sizes), i[0], side="right"))
    Bd = MX.mx_dequant_bf16(wn[e], wsn[e]).astype(np.float64)
    exact = float(A[i[0]] @ Bd[i[1]])
    terms = np.abs(A[i[0]] * Bd[i[1]])
    print("   exact f64", exact, "sum|terms|", terms.sum(), "max|term|", terms.max())
