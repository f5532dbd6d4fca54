"""GPU parity tests: int8 dynamic, float8 rowwise and MXFP8 kernels (through the
C ABI) against the CPU oracles and the reference-generated golden fixtures."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, bf16_bits_to_f32, np_from_torch_bf16, torch_bf16_from_f32
from oracle import bf16, fp8_ref as F, int8_ref as I, mx_ref as MX

pytestmark = pytest.mark.gpu

from ao_amd import ops  # noqa: E402

DEV = "cuda"


def _rel(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def _randn_bf16(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16)


def _nan_aware_equal(got, exp):
    nan_e, nan_g = (exp & 0x7F) == 0x7F, (got & 0x7F) == 0x7F
    return np.array_equal(nan_e, nan_g) and np.array_equal(got[~nan_g], exp[~nan_e])


@pytest.fixture(scope="module")
def g8():
    return np.load(os.path.join(GOLDEN, "int8_fp8.npz"))


@pytest.fixture(scope="module")
def gmx():
    return np.load(os.path.join(GOLDEN, "mx.npz"))


# ---- int8 -------------------------------------------------------------------------
def test_int8_quantize_golden(g8):
    for t in ("x", "w"):
        q, s = ops.int8_quantize_rowwise(torch_bf16_from_f32(bf16_bits_to_f32(g8[t])))
        assert np.array_equal(q.cpu().numpy(), g8[f"int8_{t}q"])
        assert np.array_equal(s.flatten().cpu().numpy(), g8[f"int8_{t}s"])


@pytest.mark.parametrize("m,k", [(1, 4096), (7, 128), (300, 14336), (64, 8)])
def test_int8_quantize_vs_oracle(m, k):
    x = _randn_bf16((m, k), m + k, 3.0)
    x[0, : min(k, 64)] = 0
    q, s = ops.int8_quantize_rowwise(x.to(DEV))
    qo, so = I.quantize_rowwise(x.float().numpy())
    assert np.array_equal(s.flatten().cpu().numpy(), so)
    assert np.array_equal(q.cpu().numpy(), qo)


def test_int8_scaled_mm_golden(g8):
    y = ops.int8_scaled_mm(
        torch.from_numpy(g8["int8_xq"]).to(DEV), torch.from_numpy(g8["int8_xs"]).to(DEV),
        torch.from_numpy(g8["int8_wq"]).to(DEV), torch.from_numpy(g8["int8_ws"]).to(DEV),
        torch_bf16_from_f32(bf16_bits_to_f32(g8["bias"])),
    )
    assert np.array_equal(y.view(torch.int16).cpu().numpy().view(np.uint16), g8["int8_y"])


@pytest.mark.parametrize("m,n,k", [(1, 16, 64), (5, 48, 256), (128, 128, 128), (130, 200, 1040), (513, 384, 4096)])
@pytest.mark.parametrize("bias", [False, True])
def test_int8_linear_vs_oracle(m, n, k, bias):
    x = _randn_bf16((m, k), 11 * m + k)
    w = _randn_bf16((n, k), 13 * n + k, 0.05)
    b = _randn_bf16((n,), 5) if bias else None
    xq, xs = ops.int8_quantize_rowwise(x.to(DEV))
    wq, ws = ops.int8_quantize_rowwise(w.to(DEV))
    y = ops.int8_scaled_mm(xq, xs, wq, ws, None if b is None else b.to(DEV))
    y_ref = I.linear(x.float().numpy(), w.float().numpy(), None if b is None else b.float().numpy())
    # integer GEMM + the reference's rounding sequence: bit-exact
    assert np.array_equal(np_from_torch_bf16(y), y_ref)
    c = ops.int_mm(xq, wq.t())
    assert np.array_equal(c.cpu().numpy(), I.int_mm(xq.cpu().numpy(), wq.cpu().numpy()))


@pytest.mark.parametrize("m,n,k,bias", [(1, 4096, 14336, False), (16, 6144, 4096, True), (128, 4096, 4096, True), (200, 48, 512, True),
                                        (64, 28672, 4096, False), (1, 16, 128, False)])
def test_int8_weight_streaming_small_m(m, n, k, bias):
    """M <= 32 takes the per-tile streaming kernel, few output tiles the LDS-staged one (rb8_kernel<INT8>), both forced here in
    turn: integer (split-K) sums + the reference's two roundings stay bit-exact, and equal to the tiled GEMM's bits."""
    from ao_amd import _lib

    lib = _lib.lib()
    x = _randn_bf16((m, k), 17 * m + k)
    w = _randn_bf16((n, k), 19 * n + k, 0.05)
    b = _randn_bf16((n,), 5) if bias else None
    xq, xs = ops.int8_quantize_rowwise(x.to(DEV))
    wq, ws = ops.int8_quantize_rowwise(w.to(DEV))
    bd = None if b is None else b.to(DEV)
    y = ops.int8_scaled_mm(xq, xs, wq, ws, bd)
    try:
        lib.ao_gemm8_set_variant(100)  # never a weight-streaming kernel
        y_gemm = ops.int8_scaled_mm(xq, xs, wq, ws, bd)
        lib.ao_gemm8_set_variant(101)  # always the LDS-staged one
        y_rb = ops.int8_scaled_mm(xq, xs, wq, ws, bd)
    finally:
        lib.ao_gemm8_set_variant(0)
    assert torch.equal(y, y_gemm) and torch.equal(y_rb, y_gemm)
    y_ref = I.linear(x.float().numpy(), w.float().numpy(), None if b is None else b.float().numpy())
    assert np.array_equal(np_from_torch_bf16(y), y_ref)


@pytest.mark.parametrize("kind", ["int8", "fp8"])
@pytest.mark.parametrize("m,n,k,bias", [(1, 4096, 4096, False), (4, 6144, 4096, True), (15, 256, 4096, True), (3, 48, 14336, False),
                                        (1, 16, 128, True), (16, 64, 2048, False)])
def test_fused_dynamic_linear_equals_cast_plus_matmul(kind, m, n, k, bias):
    """the decode-size linears with the activation cast fused in give the bits of the two-launch path
    (and therefore the oracle's for int8)."""
    x = _randn_bf16((m, k), 23 * m + k)
    x[0, :7] = torch.tensor([0.0, -0.0, 1e-30, 3.0e38, -3.0e38, 448.0, -57344.0]).to(torch.bfloat16)[: min(7, k)]
    w = _randn_bf16((n, k), 29 * n + k, 0.05)
    b = _randn_bf16((n,), 5) if bias else None
    bd = None if b is None else b.to(DEV)
    assert ops.dynamic_linear_fits(m, n, k)
    if kind == "int8":
        wq, ws = ops.int8_quantize_rowwise(w.to(DEV))
        xq, xs = ops.int8_quantize_rowwise(x.to(DEV))
        two = ops.int8_scaled_mm(xq, xs, wq, ws, bd)
        one = ops.int8_dynamic_linear(x.to(DEV), wq, ws, bd)
        y_ref = I.linear(x.float().numpy(), w.float().numpy(), None if b is None else b.float().numpy())
        assert np.array_equal(np_from_torch_bf16(one), y_ref)
    else:
        wq, ws = ops.fp8_quantize_rowwise(w.to(DEV))
        xq, xs = ops.fp8_quantize_rowwise(x.to(DEV))
        two = ops.fp8_scaled_mm(xq, wq.t(), xs, ws.t(), bd)
        one = ops.fp8_dynamic_linear(x.to(DEV), wq, ws, bd)
    assert torch.equal(one, two)


@pytest.mark.parametrize("kind", ["int8", "fp8"])
@pytest.mark.parametrize("m,n,k,bias", [(1, 1280, 8192, False), (1, 512, 1024, True), (1, 256, 3584, False), (2, 64, 4096, True), (5, 48, 896, True),
                                        (16, 32, 2048, False), (3, 64, 14336, False), (1, 16, 128, True), (16, 16, 3968, True), (7, 80, 256, False),
                                        # K that does not factor into <= 16 waves x {8, 7, 4, 2, 1} steps: the loop form (Llama-2-7B's 11008, 70B's unsharded 28672)
                                        (1, 64, 11008, False), (4, 32, 128 * 17, True), (16, 48, 128 * 19, True), (1, 128, 28672, False), (2, 32, 128 * 23, False)])
def test_decode_kernel_every_form(kind, m, n, k, bias):
    """Round 4: the straight-line decode kernel (dec8_kernels.hip: weights as full lines in a register ring, transposed through a
    wave-private LDS slab) in every ring depth it is built with, with half-line loads, and against the round-3 kernels: int8 bit-exact
    against the oracle in every form; fp8 within 1e-3 of the oracle, the fused and the two-launch call of one form bit-identical."""
    from ao_amd import _lib

    lib = _lib.lib()
    x = _randn_bf16((m, k), 31 * m + k)
    w = _randn_bf16((n, k), 37 * n + k, 0.05)
    b = _randn_bf16((n,), 5) if bias else None
    bd = None if b is None else b.to(DEV)
    xd = x.to(DEV)
    bn = None if b is None else b.float().numpy()
    if kind == "int8":
        wq, ws = ops.int8_quantize_rowwise(w.to(DEV))
        xq, xs = ops.int8_quantize_rowwise(xd)
        y_ref = I.linear(x.float().numpy(), w.float().numpy(), bn)
    else:
        wq, ws = ops.fp8_quantize_rowwise(w.to(DEV))
        xq, xs = ops.fp8_quantize_rowwise(xd)
        y_ref = F.linear(x.float().numpy(), w.float().numpy(), bn)
    ran = 0
    for variant in (0, 299, 290, 291, 292, 201, 202, 204, 207, 208):  # (291 / 292: 8-row tiles never / wherever the form allows, round 6)
        lib.ao_gemm8_set_variant(variant)
        try:
            if kind == "int8":
                two = ops.int8_scaled_mm(xq, xs, wq, ws, bd)
                one = ops.int8_dynamic_linear(xd, wq, ws, bd) if ops.dynamic_linear_fits(m, n, k) else two
            else:
                two = ops.fp8_scaled_mm(xq, wq.t(), xs, ws.t(), bd)
                one = ops.fp8_dynamic_linear(xd, wq, ws, bd) if ops.dynamic_linear_fits(m, n, k) else two
        finally:
            lib.ao_gemm8_set_variant(0)
        ran += 1
        assert torch.equal(one, two), (variant, "fused != two-launch")
        if kind == "int8":
            assert np.array_equal(np_from_torch_bf16(two), y_ref), variant
        else:
            assert _rel(two.float().cpu().numpy(), np.asarray(y_ref, dtype=np.float32)) <= 1e-3, variant
    assert ran == 10


@pytest.mark.parametrize("kind", ["int8", "fp8"])
@pytest.mark.parametrize("m,n,k,bias", [(128, 1280, 8192, False), (128, 8192, 1024, True), (17, 64, 512, True), (33, 200 * 16, 2048, False), (64, 7168, 4096, True),
                                        (100, 48, 3584, True), (256, 512, 1536, False), (129, 4096, 3584, False), (32, 16, 8192, True)])
def test_mid_m_register_ring_kernel(kind, m, n, k, bias):
    """Round 4: mid8_kernel (16 < M <= 256, few output tiles: weights through a register ring, two workgroups per CU, two-level K
    meeting) -- the product dispatch, every forced number of K parts that divides the shape, and the round-3 kernels give the oracle's
    result (int8: bit for bit, also between all forms; fp8 within 1e-3), and the FUSED form (the cast shared out among the workgroups of
    the same launch) gives the bits of cast + matmul on the same kernel form."""
    from ao_amd import _lib

    lib = _lib.lib()
    x = _randn_bf16((m, k), 41 * m + k)
    x[0, :7] = torch.tensor([0.0, -0.0, 1e-30, 3.0e38, -3.0e38, 448.0, -57344.0]).to(torch.bfloat16)
    w = _randn_bf16((n, k), 43 * n + k, 0.05)
    b = _randn_bf16((n,), 5) if bias else None
    bd = None if b is None else b.to(DEV)
    xd = x.to(DEV)
    bn = None if b is None else b.float().numpy()
    if kind == "int8":
        wq, ws = ops.int8_quantize_rowwise(w.to(DEV))
        xq, xs = ops.int8_quantize_rowwise(xd)
        y_ref = I.linear(x.float().numpy(), w.float().numpy(), bn)
    else:
        wq, ws = ops.fp8_quantize_rowwise(w.to(DEV))
        xq, xs = ops.fp8_quantize_rowwise(xd)
        y_ref = F.linear(x.float().numpy(), w.float().numpy(), bn)
    groups = k // 512
    variants = [0, 300, 301] + [310 + s for s in (1, 2, 4, 7, 8, 16) if groups % s == 0]
    for variant in variants:
        lib.ao_gemm8_set_variant(variant)
        try:
            fits = ops.dynamic_linear_fits(m, n, k)
            if kind == "int8":
                two = ops.int8_scaled_mm(xq, xs, wq, ws, bd)
                one = ops.int8_dynamic_linear(xd, wq, ws, bd) if fits else two
                again = ops.int8_dynamic_linear(xd, wq, ws, bd) if fits else two  # the ticket counters are back at zero
            else:
                two = ops.fp8_scaled_mm(xq, wq.t(), xs, ws.t(), bd)
                one = ops.fp8_dynamic_linear(xd, wq, ws, bd) if fits else two
                again = ops.fp8_dynamic_linear(xd, wq, ws, bd) if fits else two
        finally:
            lib.ao_gemm8_set_variant(0)
        assert fits == (variant != 300 and (variant != 0 or (m <= 32 and k >= 4096)) and k <= 16384), (variant, fits, "which shapes the fused mid-M form takes")
        assert torch.equal(one, two) and torch.equal(again, two), (variant, "fused != two-launch")
        if kind == "int8":
            assert np.array_equal(np_from_torch_bf16(two), y_ref), variant
        else:
            # (row 0 carries +-3e38 / 1e-30 for the fused == two-launch check above: acc * sx overflows fp32 there, in every kernel of
            # the library and in _scaled_mm alike, while the float64 oracle does not -- the oracle comparison takes the other rows)
            got, want = two.float().cpu().numpy()[1:], np.asarray(y_ref, dtype=np.float32)[1:]
            assert np.isfinite(got).all() and _rel(got, want) <= 1e-3, variant


def test_fused_dynamic_linear_shape_limits():
    # 16 < M <= 256: the form with the cast shared out inside the launch (mid8_kernels.hip) -- by itself only where it is the faster
    # kernel (M <= 32, K >= 4096, K % 512 == 0, K <= 16384), anywhere the shape allows when forced (ao_gemm8_set_variant 301)
    assert ops.dynamic_linear_fits(16, 64, 2048) and not ops.dynamic_linear_fits(17, 64, 2048)
    assert ops.dynamic_linear_fits(17, 64, 4096) and ops.dynamic_linear_fits(32, 1280, 8192) and not ops.dynamic_linear_fits(33, 1280, 8192)
    assert not ops.dynamic_linear_fits(17, 64, 4096 + 128) and not ops.dynamic_linear_fits(32, 64, 32768)
    assert not ops.dynamic_linear_fits(8, 64, 14336) and not ops.dynamic_linear_fits(1, 40, 4096) and not ops.dynamic_linear_fits(1, 64, 4000)
    x = torch.zeros(8, 14336, dtype=torch.bfloat16, device=DEV)
    wq = torch.zeros(64, 14336, dtype=torch.int8, device=DEV)
    with pytest.raises(ValueError, match="fused form holds the cast activation in LDS"):
        ops.int8_dynamic_linear(x, wq, torch.ones(64, device=DEV))


def test_int8_extremes_exact():
    """int32 accumulation at full range: K * 127 * 128 stays exact."""
    m, n, k = 33, 64, 8192
    a = torch.full((m, k), -128, dtype=torch.int8)
    b = torch.full((n, k), 127, dtype=torch.int8)
    a[1::2] = 127
    c = ops.int_mm(a.to(DEV), b.to(DEV).t())
    assert np.array_equal(c.cpu().numpy(), I.int_mm(a.numpy(), b.numpy()))


# ---- fp8 -------------------------------------------------------------------------
def test_fp8_quantize_golden(g8):
    q, s = ops.fp8_quantize_rowwise(torch_bf16_from_f32(bf16_bits_to_f32(g8["fp8_x"])))
    assert np.array_equal(s.flatten().cpu().numpy(), g8["fp8_xs"])
    assert np.array_equal(q.view(torch.uint8).cpu().numpy(), g8["fp8_xq"])
    q, s = ops.fp8_quantize_rowwise(torch.zeros(2, 256, dtype=torch.bfloat16, device=DEV))
    assert np.all((q.view(torch.uint8).cpu().numpy() & 0x7F) == 0x7F)  # 0/0 -> NaN like the reference
    assert np.all(s.cpu().numpy() == 0)


@pytest.mark.parametrize("m,k", [(1, 4096), (9, 128), (257, 8192)])
def test_fp8_quantize_vs_oracle(m, k):
    x = _randn_bf16((m, k), 3 * m + k, 5.0)
    x[0, 0] = 1e-30  # exercises fp8 subnormals after scaling
    q, s = ops.fp8_quantize_rowwise(x.to(DEV))
    qo, so = F.quantize_rowwise(x.float().numpy())
    assert np.array_equal(s.flatten().cpu().numpy(), so)
    assert np.array_equal(q.view(torch.uint8).cpu().numpy(), qo)


def test_fp8_cast_every_representable_neighbourhood():
    """RNE of the hardware cast around every e4m3 value and midpoint."""
    pos = F.E4M3[:127].astype(np.float64)
    pts = np.concatenate([pos, (pos[1:] + pos[:-1]) / 2, np.nextafter(((pos[1:] + pos[:-1]) / 2).astype(np.float32), 0), np.nextafter(((pos[1:] + pos[:-1]) / 2).astype(np.float32), 1e9)])
    pts = np.concatenate([pts, -pts]).astype(np.float32)
    pts = pts[np.abs(pts) <= 448]
    k = ((len(pts) + 255) // 256) * 256
    buf = np.zeros(k, np.float32); buf[: len(pts)] = pts
    buf[-1] = 448.0  # amax = 448 -> scale exactly 1.0
    xb = bf16.bf16_round(buf)  # inputs must be bf16
    q, s = ops.fp8_quantize_rowwise(torch_bf16_from_f32(xb[None, :]))
    qo, so = F.quantize_rowwise(xb[None, :])
    assert float(s[0, 0]) == 1.0 == float(so[0])
    assert np.array_equal(q.view(torch.uint8).cpu().numpy(), qo)


@pytest.mark.parametrize("m,n,k", [(1, 16, 128), (3, 48, 256), (64, 128, 4096), (65, 128, 512), (300, 208, 1040), (2048, 256, 1024)])
@pytest.mark.parametrize("bias", [False, True])
def test_fp8_linear_vs_oracle(m, n, k, bias):
    x = _randn_bf16((m, k), 7 * m + k)
    w = _randn_bf16((n, k), 3 * n + k, 0.05)
    b = _randn_bf16((n,), 9) if bias else None
    xq, xs = ops.fp8_quantize_rowwise(x.to(DEV))
    wq, ws = ops.fp8_quantize_rowwise(w.to(DEV))
    y = ops.fp8_scaled_mm(xq, wq.t(), xs, ws.t(), None if b is None else b.to(DEV))
    y_ref = F.scaled_mm(
        xq.view(torch.uint8).cpu().numpy(), wq.view(torch.uint8).cpu().numpy(),
        xs.flatten().cpu().numpy(), ws.flatten().cpu().numpy(), None if b is None else b.float().numpy(),
    )
    yn = np_from_torch_bf16(y)
    assert _rel(yn, y_ref) <= 1e-3  # BASELINE.json tolerance
    # elementwise: one bf16 ulp, plus fp32 accumulation-order noise on cancelling sums
    atol = 1e-5 * np.sqrt(k) * float(np.abs(y_ref).max())
    assert np.all(np.abs(yn - y_ref) <= np.abs(y_ref) * 2.0 ** -7 + atol)


def test_fp8_vs_torch_scaled_mm_if_available():
    x = _randn_bf16((32, 512), 1).to(DEV)
    w = _randn_bf16((64, 512), 2, 0.05).to(DEV)
    xq, xs = ops.fp8_quantize_rowwise(x)
    wq, ws = ops.fp8_quantize_rowwise(w)
    try:
        ref = torch._scaled_mm(xq, wq.t(), scale_a=xs, scale_b=ws.t(), out_dtype=torch.bfloat16)
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        pytest.skip(f"torch._scaled_mm rowwise unavailable here: {e}")
    y = ops.fp8_scaled_mm(xq, wq.t(), xs, ws.t())
    assert _rel(np_from_torch_bf16(y), np_from_torch_bf16(ref)) < 5e-3


# ---- MXFP8 -------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["rceil", "floor"])
def test_mxfp8_semantic_contract(gmx, name):
    x = torch_bf16_from_f32(bf16_bits_to_f32(gmx[f"sem_{name}_x"]))
    q, s = ops.mxfp8_quantize(x, name)
    assert np.array_equal(s.view(torch.uint8).flatten().cpu().numpy(), gmx[f"sem_{name}_scale"].flatten())
    got, exp = q.view(torch.uint8).cpu().numpy(), gmx[f"sem_{name}_data"]
    for i, nm in enumerate(gmx[f"sem_{name}_names"]):
        assert _nan_aware_equal(got[i], exp[i]), nm


@pytest.mark.parametrize("name", ["rceil", "floor"])
def test_mxfp8_quantize_golden_and_oracle(gmx, name):
    x = torch_bf16_from_f32(bf16_bits_to_f32(gmx["x"]))
    q, s = ops.mxfp8_quantize(x, name)
    assert np.array_equal(s.view(torch.uint8).cpu().numpy(), gmx[f"{name}_scale"])
    assert _nan_aware_equal(q.view(torch.uint8).cpu().numpy(), gmx[f"{name}_data"])
    big = _randn_bf16((64, 4096), 3, 10.0)
    q, s = ops.mxfp8_quantize(big.to(DEV), name)
    qo, so = MX.to_mx(big.float().numpy(), MX.RCEIL if name == "rceil" else MX.FLOOR)
    assert np.array_equal(s.view(torch.uint8).cpu().numpy(), so)
    assert np.array_equal(q.view(torch.uint8).cpu().numpy(), qo)


def test_mxfp8_grouped_mm_golden(gmx):
    y = ops.mxfp8_grouped_mm(
        torch.from_numpy(gmx["g_a_data"]).to(DEV), torch.from_numpy(gmx["g_a_scale"]).to(DEV),
        torch.from_numpy(gmx["g_w_data"]).to(DEV), torch.from_numpy(gmx["g_w_scale"]).to(DEV),
        torch.from_numpy(gmx["g_offs"]).to(DEV),
    )
    ref = bf16_bits_to_f32(gmx["g_y"])
    rows = int(gmx["g_offs"][-1])
    yn = np_from_torch_bf16(y)[:rows]
    assert _rel(yn, ref[:rows]) <= 1e-3
    mag = MX.grouped_mm(gmx["g_a_data"], gmx["g_a_scale"], gmx["g_w_data"], gmx["g_w_scale"], gmx["g_offs"], return_abs=True)[1]
    assert np.all(np.abs(yn - ref[:rows]) <= np.abs(ref[:rows]) * 2.0 ** -7 + mag[:rows] * 2.0 ** -16)


@pytest.mark.parametrize("sizes", [[16, 16, 16, 16], [32, 0, 5, 27], [1, 70, 3, 0], [128, 0, 0, 0]])
def test_mxfp8_grouped_mm_vs_oracle(sizes):
    E, N, K = len(sizes), 64, 512
    M = sum(sizes)
    a = _randn_bf16((M, K), 21)
    w = _randn_bf16((E, N, K), 22, 0.1)
    offs = torch.tensor(np.cumsum(sizes), dtype=torch.int32)
    a_d, a_s = ops.mxfp8_quantize(a.to(DEV), "rceil")
    w_d, w_s = ops.mxfp8_quantize(w.to(DEV), "rceil")
    y = ops.mxfp8_grouped_mm(a_d, a_s, w_d, w_s, offs.to(DEV))
    y_ref, mag = MX.grouped_mm(
        a_d.view(torch.uint8).cpu().numpy(), a_s.view(torch.uint8).cpu().numpy(),
        w_d.view(torch.uint8).cpu().numpy(), w_s.view(torch.uint8).cpu().numpy(), offs.numpy(), return_abs=True,
    )
    yn = np_from_torch_bf16(y)
    assert _rel(yn, y_ref) <= 1e-3
    # element-wise: one bf16 ulp of the result + fp32 accumulation-order noise, which scales
    # with sum|a||b| (cancelling outputs), 2^-18 of it measured on the scaled MFMA
    assert np.all(np.abs(yn - y_ref) <= np.abs(y_ref) * 2.0 ** -7 + mag * 2.0 ** -16)
    # SQNR vs the unquantised bf16 grouped matmul: reference bar >= 27 dB (test_mxfp8_grouped_mm.py:120-122)
    full = np.zeros_like(y_ref)
    st = 0
    for e, sz in enumerate(sizes):
        full[st : st + sz] = a[st : st + sz].float().numpy() @ w[e].float().numpy().T
        st += sz
    sqnr = 20 * np.log10(np.linalg.norm(full) / np.linalg.norm(full - yn))
    assert sqnr >= 27, sqnr


@pytest.mark.parametrize(
    "sizes,n,k",
    [([8, 8, 8, 8], 80, 2048), ([16, 16, 16, 16], 256, 4096), ([40, 0, 5, 27], 80, 2048), ([70, 1, 3, 0], 144, 4096),
     ([3, 0, 0, 1], 2048, 2048), ([32, 32, 32, 32, 0, 0, 0, 0], 192, 6144)],
)
def test_mxfp8_grouped_mm_a_stationary_kernel(sizes, n, k):
    """With variant 111, K % 2048 == 0 takes the A-stationary kernel (mx_grouped_kernel): every m-tiling (avg group <= 8,
    <= 16, larger), groups larger than one pass, empty experts, N not a multiple of the tile group."""
    E = len(sizes)
    M = sum(sizes)
    a = _randn_bf16((M, k), 31 + n)
    w = _randn_bf16((E, n, k), 32 + k, 0.1)
    offs = torch.tensor(np.cumsum(sizes), dtype=torch.int32)
    a_d, a_s = ops.mxfp8_quantize(a.to(DEV), "rceil")
    w_d, w_s = ops.mxfp8_quantize(w.to(DEV), "rceil")
    from ao_amd import _lib

    try:
        _lib.lib().ao_gemm8_set_variant(111)  # the older kernels (the LDS-staged one is the default path)
        y = ops.mxfp8_grouped_mm(a_d, a_s, w_d, w_s, offs.to(DEV))
    finally:
        _lib.lib().ao_gemm8_set_variant(0)
    y_ref, mag = MX.grouped_mm(
        a_d.view(torch.uint8).cpu().numpy(), a_s.view(torch.uint8).cpu().numpy(),
        w_d.view(torch.uint8).cpu().numpy(), w_s.view(torch.uint8).cpu().numpy(), offs.numpy(), return_abs=True,
    )
    yn = np_from_torch_bf16(y)
    assert _rel(yn, y_ref) <= 1e-3
    assert np.all(np.abs(yn - y_ref) <= np.abs(y_ref) * 2.0 ** -7 + mag * 2.0 ** -16)


@pytest.mark.parametrize(
    "sizes,n,k",
    [([8, 8, 8, 8], 80, 2048), ([16, 16, 16, 16], 256, 4096), ([40, 0, 5, 27], 80, 512), ([200, 1, 3, 0], 144, 1024), ([0, 0, 130], 4096, 384),
     ([16, 32, 16, 0, 32, 0, 16, 16], 2048, 2048),
     # the device-side slab enumeration: every m-tile count of a 64-row slab (1, 17, 33, 49 rows), a group that straddles
     # two slabs next to empty ones, and more than 64 experts (second pass of the 64-lane scan) with the tokens at the far end
     ([1, 17, 33, 49, 0, 64, 65, 0], 80, 512), ([0] * 66 + [5, 0, 70, 3], 64, 256), ([2] * 70, 48, 384),
     ([130, 0, 0, 257, 1], 96, 256)],
)
def test_mxfp8_grouped_mm_lds_staged_kernel(sizes, n, k):
    """The LDS-staged weight-streaming form (rb8_kernel<RB8_MX>, forced with variant 110): groups larger than one 128-row
    slab, empty experts, both tile widths, N not a multiple of the tile; same bits on repeated launches and agreement
    with the A-stationary / per-tile kernels up to accumulation order."""
    from ao_amd import _lib

    lib = _lib.lib()
    E = len(sizes)
    M = sum(sizes)
    a = _randn_bf16((M, k), 41 + n)
    w = _randn_bf16((E, n, k), 42 + k, 0.1)
    offs = torch.tensor(np.cumsum(sizes), dtype=torch.int32)
    a_d, a_s = ops.mxfp8_quantize(a.to(DEV), "rceil")
    w_d, w_s = ops.mxfp8_quantize(w.to(DEV), "rceil")
    try:
        lib.ao_gemm8_set_variant(110)
        y = ops.mxfp8_grouped_mm(a_d, a_s, w_d, w_s, offs.to(DEV))
        assert torch.equal(ops.mxfp8_grouped_mm(a_d, a_s, w_d, w_s, offs.to(DEV)), y)
        lib.ao_gemm8_set_variant(111)
        y_other = ops.mxfp8_grouped_mm(a_d, a_s, w_d, w_s, offs.to(DEV))
    finally:
        lib.ao_gemm8_set_variant(0)
    y_ref, mag = MX.grouped_mm(
        a_d.view(torch.uint8).cpu().numpy(), a_s.view(torch.uint8).cpu().numpy(),
        w_d.view(torch.uint8).cpu().numpy(), w_s.view(torch.uint8).cpu().numpy(), offs.numpy(), return_abs=True,
    )
    yn = np_from_torch_bf16(y)
    assert _rel(yn, y_ref) <= 1e-3
    assert np.all(np.abs(yn - y_ref) <= np.abs(y_ref) * 2.0 ** -7 + mag * 2.0 ** -16)
    assert _rel(np_from_torch_bf16(y_other), yn) <= 1e-3


@pytest.mark.parametrize("variant", [0, 129])
@pytest.mark.parametrize(
    "sizes,n,k",
    [([16, 16, 16, 16], 256, 4096),          # 16 tiles x 32 steps over 32 shares: every tile cut in two
     ([40, 0, 5, 27], 80, 512),              # 24 steps in all: ONE workgroup walks six tiles of three experts (N not a multiple of 64)
     ([1, 17, 33, 49, 0, 64, 65, 0], 80, 512),  # every m-tile count, a group of two slabs
     ([2] * 64, 48, 384),                    # 64 experts (the whole group table), 3-step tiles cut at every offset
     ([3, 0, 0, 1], 64, 14336),              # 112-step tiles in seven pieces each (more pieces than one batch of the reducer)
     ([3, 0, 0, 1], 2048, 2048),             # shares that are whole tiles: nothing meets
     ([32, 0, 0, 0, 32, 64, 0, 0], 1024, 4096)],  # BASELINE config 5's routing
)
def test_mxfp8_grouped_mm_stream_k_kernel(sizes, n, k, variant):
    """mx_stream_kernel (decode-size groups).  0: the product -- 16 waves / 256-column tiles, ONE workgroup per CU, block scales fetched per 4 k steps
    where K % 512 == 0; other K, and every K under variant 129, the 8-wave / 128-column form with the scales fetched per step, two per CU:
    shares of the (slab, tile, k step) space that cross tile and expert boundaries, pieces of cut tiles meeting through the
    split-K workspace.  Against the oracle, same bits on repeated launches (the pieces are added in k order), and agreement with the
    one-workgroup-per-tile kernel (variant 113) up to accumulation order."""
    from ao_amd import _lib

    lib = _lib.lib()
    E, M = len(sizes), sum(sizes)
    assert M <= 48 * E  # the dispatch's decode condition
    a = _randn_bf16((M, k), 61 + n)
    w = _randn_bf16((E, n, k), 62 + k, 0.1)
    offs = torch.tensor(np.cumsum(sizes), dtype=torch.int32)
    a_d, a_s = ops.mxfp8_quantize(a.to(DEV), "rceil")
    w_d, w_s = ops.mxfp8_quantize(w.to(DEV), "rceil")
    try:
        lib.ao_gemm8_set_variant(variant)
        y = ops.mxfp8_grouped_mm(a_d, a_s, w_d, w_s, offs.to(DEV))
        for _ in range(3):  # tickets are left ready for the next launch; the sum does not depend on arrival order
            assert torch.equal(ops.mxfp8_grouped_mm(a_d, a_s, w_d, w_s, offs.to(DEV)), y)
        lib.ao_gemm8_set_variant(113)
        y_tile = ops.mxfp8_grouped_mm(a_d, a_s, w_d, w_s, offs.to(DEV))
    finally:
        lib.ao_gemm8_set_variant(0)
    y_ref, mag = MX.grouped_mm(
        a_d.view(torch.uint8).cpu().numpy(), a_s.view(torch.uint8).cpu().numpy(),
        w_d.view(torch.uint8).cpu().numpy(), w_s.view(torch.uint8).cpu().numpy(), offs.numpy(), return_abs=True,
    )
    yn = np_from_torch_bf16(y)
    assert _rel(yn, y_ref) <= 1e-3
    assert np.all(np.abs(yn - y_ref) <= np.abs(y_ref) * 2.0 ** -7 + mag * 2.0 ** -16)
    yt = np_from_torch_bf16(y_tile)  # the one-workgroup-per-tile kernel (scales per 4 steps when K % 512 == 0) against the oracle too
    assert _rel(yt, yn) <= 1e-3
    assert np.all(np.abs(yt - y_ref) <= np.abs(y_ref) * 2.0 ** -7 + mag * 2.0 ** -16)


@pytest.mark.parametrize("mode", ["rceil", "floor"])
@pytest.mark.parametrize(
    "sizes,n,k",
    [([16, 16, 16, 16], 256, 4096),            # every tile cut in two
     ([40, 0, 5, 27], 80, 512),                # one workgroup walks tiles of three experts; N not a multiple of the 256-column tile
     ([1, 17, 33, 49, 0, 64, 65, 0], 80, 512),  # every m-tile count, a group of two slabs (rows the waves do not fetch)
     ([2] * 64, 48, 1024),                     # 64 experts
     ([3, 0, 0, 1], 64, 14336),                # 112-step tiles in many pieces
     ([32, 0, 0, 0, 32, 64, 0, 0], 1024, 4096),  # BASELINE config 5's routing
     ([32, 0, 0, 0, 32, 64, 0, 0], 3584, 4096)],  # ... on a width that is not a multiple of 256 x 8
)
def test_mxfp8_grouped_mm_fused_activation_cast(sizes, n, k, mode):
    """SURVEY 8 f1 for the MX format (round 6): ao_mxfp8_grouped_mm_dyn casts the bf16 activations 1 x 32 inside the grouped GEMM's A-fill
    (reference call order mxfp8_grouped_mm.py:330-371: to_mx(A) then the grouped mm).  Bit for bit the two-launch path (same cast function,
    same kernel, same k order), against the oracle, same bits on repeated launches; inf / NaN / zero blocks go through the cast as they do
    stand-alone; shapes it does not take are refused."""
    E, M = len(sizes), sum(sizes)
    a = _randn_bf16((M, k), 71 + n, 3.0)
    a[0, :32] = 0.0
    a[min(1, M - 1), 40] = float("inf")
    if M > 2:
        a[2, 70] = float("nan")
    w = _randn_bf16((E, n, k), 72 + k, 0.1)
    offs = torch.tensor(np.cumsum(sizes), dtype=torch.int32).to(DEV)
    w_d, w_s = ops.mxfp8_quantize(w.to(DEV), "rceil")
    ad = a.to(DEV)
    assert ops.mxfp8_grouped_mm_dyn_fits(M, n, k, E)
    y = ops.mxfp8_grouped_mm_dyn(ad, w_d, w_s, offs, mode)
    for _ in range(3):
        assert torch.equal(ops.mxfp8_grouped_mm_dyn(ad, w_d, w_s, offs, mode).view(torch.int16), y.view(torch.int16))
    a_d, a_s = ops.mxfp8_quantize(ad, mode)
    y2 = ops.mxfp8_grouped_mm(a_d, a_s, w_d, w_s, offs)
    rows = int(offs[-1])
    assert torch.equal(y[:rows].view(torch.int16), y2[:rows].view(torch.int16))  # (NaN rows included: compared as bits)
    # and the two-launch path is the oracle's (on the finite rows)
    y_ref, mag = MX.grouped_mm(a_d.view(torch.uint8).cpu().numpy(), a_s.view(torch.uint8).cpu().numpy(), w_d.view(torch.uint8).cpu().numpy(),
                               w_s.view(torch.uint8).cpu().numpy(), offs.cpu().numpy(), return_abs=True)
    yn = np_from_torch_bf16(y)[:rows]
    fin = np.isfinite(y_ref[:rows]).all(axis=1) & np.isfinite(yn).all(axis=1)
    assert fin.sum() >= rows - 3
    assert np.all(np.abs(yn[fin] - y_ref[:rows][fin]) <= np.abs(y_ref[:rows][fin]) * 2.0 ** -7 + mag[:rows][fin] * 2.0 ** -16)


@pytest.mark.parametrize("sizes,n,k", [([16, 16, 16, 16], 256, 4096), ([1, 17, 33, 49, 0, 64, 65, 0], 80, 512), ([32, 0, 0, 0, 32, 64, 0, 0], 3584, 4096),
                                       ([3, 0, 0, 1], 64, 14336)])
def test_mxfp8_grouped_mm_pair_is_two_single_products(sizes, n, k):
    """ao_mxfp8_grouped_mm_dyn_pair / _pair: an MoE layer's x @ w1 and x @ w3 in ONE launch (the second weight tensor's column tiles follow the
    first's in every slab of the stream-K space): both outputs bit for bit the single-product launches (these shapes' shares cut the tiles at the
    same k steps in both forms; where they do not, single elements round the other way: the race screen below), fused cast and pre-cast activations,
    repeated launches, and through the mirror (_to_mxfp8_then_scaled_grouped_mm_pair)."""
    from ao_amd.prototype import mx as MXP

    E, M = len(sizes), sum(sizes)
    a = _randn_bf16((M, k), 81 + n, 2.0).to(DEV)
    w1 = _randn_bf16((E, n, k), 82 + k, 0.1).to(DEV)
    w3 = _randn_bf16((E, n, k), 83 + k, 0.1).to(DEV)
    offs = torch.tensor(np.cumsum(sizes), dtype=torch.int32).to(DEV)
    rows = int(offs[-1])
    w1d, w1s = ops.mxfp8_quantize(w1, "rceil")
    w3d, w3s = ops.mxfp8_quantize(w3, "rceil")
    assert ops.mxfp8_grouped_mm_pair_fits(M, n, k, E)
    want1 = ops.mxfp8_grouped_mm_dyn(a, w1d, w1s, offs)[:rows]
    want3 = ops.mxfp8_grouped_mm_dyn(a, w3d, w3s, offs)[:rows]
    for _ in range(3):
        y1, y3 = ops.mxfp8_grouped_mm_pair(a, w1d, w1s, w3d, w3s, offs)
        assert torch.equal(y1[:rows], want1) and torch.equal(y3[:rows], want3)
    aq, asc = ops.mxfp8_quantize(a, "rceil")
    y1, y3 = ops.mxfp8_grouped_mm_pair(aq, w1d, w1s, w3d, w3s, offs, a_scale=asc)
    assert torch.equal(y1[:rows], want1) and torch.equal(y3[:rows], want3)
    m1, m3 = MXP._to_mxfp8_then_scaled_grouped_mm_pair(a, w1.transpose(-2, -1), w3.transpose(-2, -1), offs)
    assert torch.equal(m1[:rows], want1) and torch.equal(m3[:rows], want3)
    assert torch.equal(MXP._to_mxfp8_then_scaled_grouped_mm(a, w1.transpose(-2, -1), offs)[:rows], want1)


def test_mxfp8_grouped_mm_pair_shares_that_begin_inside_a_tile_race_screen():
    """Round-6 regression (found by tools/fuzz_long.py): with the cast fused and a share that BEGINS inside a tile, wave 1 takes the head piece's
    ticket from inside the k loop, and the two waits after it allowed one request too many in flight -- in ~1 % of such launches wave 1 cast
    rows 4 .. 7 of one k step's tile from a raw stage that had not landed (4 rows x 256 columns of garbage, not reproducible).  E = 8,
    N = K = 4096 as a pair launch has 28- / 20- / 12-step shares against 32-step tiles: every launch has head pieces.  480 launches: every one
    gives the bits of the first for its routing, and the single-product launches' values up to the summation order of the cut tiles."""
    E, n, k = 8, 4096, 4096
    w1 = _randn_bf16((E, n, k), 91, 0.1).to(DEV)
    w3 = _randn_bf16((E, n, k), 92, 0.1).to(DEV)
    w1d, w1s = ops.mxfp8_quantize(w1, "rceil")
    w3d, w3s = ops.mxfp8_quantize(w3, "rceil")
    del w1, w3
    rng = np.random.default_rng(6)
    for draw in range(60):
        sizes = [int(v) for v in rng.choice([0, 0, 1, 5, 16, 31, 33, 48], size=E)]
        if sum(sizes) == 0:
            sizes[0] = 3
        M = sum(sizes)
        a = _randn_bf16((M, k), 1000 + draw, 2.0).to(DEV)
        offs = torch.tensor(np.cumsum(sizes), dtype=torch.int32).to(DEV)
        assert ops.mxfp8_grouped_mm_pair_fits(M, n, k, E)
        aq, asc = ops.mxfp8_quantize(a, "rceil")
        want = [ops.mxfp8_grouped_mm(aq, asc, w1d, w1s, offs)[:M].float(), ops.mxfp8_grouped_mm(aq, asc, w3d, w3s, offs)[:M].float()]
        first = None
        for rep in range(8):
            got = [t[:M] for t in ops.mxfp8_grouped_mm_pair(a, w1d, w1s, w3d, w3s, offs)]
            if first is None:
                first = got
                for y, ref in zip(got, want):  # another cut of the tiles: single elements may round the other way, nothing more
                    ulp = torch.exp2(torch.floor(torch.log2(ref.abs().clamp_min(1e-30))) - 7)
                    off = (y.float() - ref).abs() > 2 * ulp  # (2: a flip across a power of two is one ulp of the upper binade)
                    assert int(off.sum()) == 0, (draw, sizes, int(off.sum()), off.nonzero()[:4].tolist())
                    assert int((y.float() != ref).sum()) <= 8, (draw, sizes, int((y.float() != ref).sum()))
            else:
                assert torch.equal(got[0], first[0]) and torch.equal(got[1], first[1]), (draw, rep, sizes)


def test_mxfp8_grouped_mm_fused_cast_refuses_other_shapes_and_the_mirror_falls_back():
    from ao_amd.prototype import mx as MXP

    assert not ops.mxfp8_grouped_mm_dyn_fits(128, 256, 384, 4)      # K % 512 != 0
    assert not ops.mxfp8_grouped_mm_dyn_fits(4 * 49, 256, 512, 4)   # groups beyond decode size
    assert not ops.mxfp8_grouped_mm_dyn_fits(64, 256, 512, 65)      # more experts than the group table holds
    a = _randn_bf16((64, 384), 5).to(DEV)
    w = _randn_bf16((2, 64, 384), 6, 0.1).to(DEV)
    w_d, w_s = ops.mxfp8_quantize(w, "rceil")
    offs = torch.tensor([32, 64], dtype=torch.int32, device=DEV)
    with pytest.raises(ValueError, match="not a decode-size grouped product"):
        ops.mxfp8_grouped_mm_dyn(a, w_d, w_s, offs)
    # the mirror (_to_mxfp8_then_scaled_grouped_mm) takes the fused launch where it fits and the two launches elsewhere: same bits either way
    for k in (384, 1024):
        a = _randn_bf16((64, k), 7).to(DEV)
        b_t = _randn_bf16((2, 128, k), 8, 0.1).to(DEV).transpose(-2, -1)
        want = None
        for fuse in (False, True):
            MXP.FUSE_ACTIVATION_CAST = fuse
            try:
                got = MXP._to_mxfp8_then_scaled_grouped_mm(a, b_t, offs)
            finally:
                MXP.FUSE_ACTIVATION_CAST = True
            want = got if want is None else want
            assert torch.equal(got, want)


@pytest.mark.parametrize("m,n,k,bias", [(128, 1024, 8192, False), (128, 7168, 8192, True), (200, 8192, 1024, True), (2048, 1024, 1024, False),
                                        (96, 48, 256, True), (65, 4096, 3584, False), (1000, 208, 384, True)])
def test_fp8_weight_streaming_mid_m(m, n, k, bias):
    """64 < M on narrow (TP-sharded) weights takes fp8_rb_kernel: both tile widths, split-K from 1 to 16 parts, ragged M,
    N not a multiple of the tile, bias; forced on for every shape (variant 101) and bit-reproducible."""
    from ao_amd import _lib

    lib = _lib.lib()
    x = _randn_bf16((m, k), 11 * m + k)
    w = _randn_bf16((n, k), 13 * n + k, 0.05)
    b = _randn_bf16((n,), 9) if bias else None
    xq, xs = ops.fp8_quantize_rowwise(x.to(DEV))
    wq, ws = ops.fp8_quantize_rowwise(w.to(DEV))
    bd = None if b is None else b.to(DEV)
    try:
        lib.ao_gemm8_set_variant(101)
        y = ops.fp8_scaled_mm(xq, wq.t(), xs, ws.t(), bd)
        for _ in range(5):
            assert torch.equal(ops.fp8_scaled_mm(xq, wq.t(), xs, ws.t(), bd), y)
        lib.ao_gemm8_set_variant(100)
        y_gemm = ops.fp8_scaled_mm(xq, wq.t(), xs, ws.t(), bd)
    finally:
        lib.ao_gemm8_set_variant(0)
    y_ref = F.scaled_mm(
        xq.view(torch.uint8).cpu().numpy(), wq.view(torch.uint8).cpu().numpy(),
        xs.flatten().cpu().numpy(), ws.flatten().cpu().numpy(), None if b is None else b.float().numpy(),
    )
    yn = np_from_torch_bf16(y)
    assert _rel(yn, y_ref) <= 1e-3
    atol = 1e-5 * np.sqrt(k) * float(np.abs(y_ref).max())
    assert np.all(np.abs(yn - y_ref) <= np.abs(y_ref) * 2.0 ** -7 + atol)
    assert _rel(np_from_torch_bf16(y_gemm), yn) <= 1e-3  # the GEMM kernels agree up to accumulation order


@pytest.mark.parametrize("variant", [1, 2, 4, 8, 16, 32])
@pytest.mark.parametrize("m,n,k", [(130, 208, 1152), (300, 528, 256), (513, 384, 4096), (257, 272, 128)])  # N % 16 == 0 (fp8 requirement)
def test_gemm8_every_kernel_variant(variant, m, n, k):
    """register-staged, 128x128 / 256x128 / 256x256 LDS-DMA kernels: same bits for int8 (integer GEMM + the
    reference's rounding sequence), <= 1e-3 for fp8; ragged M and N against every tile shape."""
    from ao_amd import _lib

    lib = _lib.lib()
    x = _randn_bf16((m, k), 3 * m + k)
    w = _randn_bf16((n, k), 5 * n + k, 0.05)
    b = _randn_bf16((n,), 9)
    try:
        lib.ao_gemm8_set_variant(variant)
        xq, xs = ops.int8_quantize_rowwise(x.to(DEV))
        wq, ws = ops.int8_quantize_rowwise(w.to(DEV))
        y8 = ops.int8_scaled_mm(xq, xs, wq, ws, b.to(DEV))
        c = ops.int_mm(xq, wq.t())
        fq, fs = ops.fp8_quantize_rowwise(x.to(DEV))
        gq, gs = ops.fp8_quantize_rowwise(w.to(DEV))
        yf = ops.fp8_scaled_mm(fq, gq.t(), fs, gs.t(), b.to(DEV))
    finally:
        lib.ao_gemm8_set_variant(0)
    assert np.array_equal(np_from_torch_bf16(y8), I.linear(x.float().numpy(), w.float().numpy(), b.float().numpy()))
    assert np.array_equal(c.cpu().numpy(), I.int_mm(xq.cpu().numpy(), wq.cpu().numpy()))
    assert _rel(np_from_torch_bf16(yf), F.linear(x.float().numpy(), w.float().numpy(), b.float().numpy())) <= 1e-3


@pytest.mark.parametrize("variant", [32, 33])  # 256 x 256 tiles; 256 x 128 tiles (round 5)
@pytest.mark.parametrize("m,n,k", [(2048, 4096, 4096), (4096, 1024, 14336), (1000, 784, 2048), (768, 1296, 128), (130, 208, 384)])
def test_gemm8_phase_interleaved_kernel_race_screen(m, n, k, variant):
    """gemm8_p8_kernel (variant 32): staggered wave rows, counted vmcnt, LDS slots refilled two phases after their last read --
    an ordering mistake would show as rare wrong tiles.  The int8 GEMM is exact, so ANY difference from the two-stage kernel
    (variant 8) in any of 6 runs on fresh random operands is a failure; fp8 within accumulation order."""
    from ao_amd import _lib

    lib = _lib.lib()
    for run in range(6):
        g = torch.Generator(device=DEV).manual_seed(1000 * run + m)
        a = torch.randint(-128, 128, (m, k), device=DEV, dtype=torch.int8, generator=g)
        b = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
        f = torch.randn(m, k, device=DEV, generator=g).to(torch.bfloat16)
        h = (torch.randn(n, k, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
        fq, fs = ops.fp8_quantize_rowwise(f)
        hq, hs = ops.fp8_quantize_rowwise(h)
        try:
            lib.ao_gemm8_set_variant(8)
            want = ops.int_mm(a, b.t())
            want_f = ops.fp8_scaled_mm(fq, hq.t(), fs, hs.t())
            lib.ao_gemm8_set_variant(variant)
            got = ops.int_mm(a, b.t())
            got_f = ops.fp8_scaled_mm(fq, hq.t(), fs, hs.t())
            raw = ops.fp8_mm_f32(fq, hq.t())
        finally:
            lib.ao_gemm8_set_variant(0)
        assert torch.equal(got, want), (run, int((got != want).sum()))
        assert _rel(np_from_torch_bf16(got_f), np_from_torch_bf16(want_f)) <= 1e-3
        assert _rel(np_from_torch_bf16((raw * fs * hs.t()).to(torch.bfloat16)), np_from_torch_bf16(want_f)) <= 2e-3


@pytest.mark.parametrize("m,n,k,bias", [(2048, 4096, 4096, False), (4096, 5120, 1024, True), (8192, 6144, 384, False), (256, 256, 256, True),
                                        (16384, 4096, 4096, True), (2304, 14336, 256, False)])
def test_gemm8_persistent_form_equals_one_workgroup_per_tile(m, n, k, bias):
    """gemm8_p8p_kernel (round 6): one workgroup per CU walks its XCD's share of the 256 x 256 tiles with the K-tile stream running across
    tile boundaries, swapped MFMA operands and a register-only epilogue (v_permlane32_swap / v_permlane16_swap instead of the LDS transposition).
    Against gemm8_p8_kernel (one workgroup per tile) on fresh operands: int32 and the scaled bf16 output of the int8 GEMM bit for bit (the
    epilogue's rounding sequence is the same), fp8 within accumulation order; 1, 2 and 3+ tiles per workgroup, odd K-tile counts (the buffer
    parity then flips from tile to tile), a grid with idle workgroups, with and without bias."""
    from ao_amd import _lib

    lib = _lib.lib()
    for run in range(3):
        g = torch.Generator(device=DEV).manual_seed(77 * run + m + k)
        a = torch.randint(-128, 128, (m, k), device=DEV, dtype=torch.int8, generator=g)
        b = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
        sa = torch.rand(m, 1, device=DEV, generator=g) * 0.01 + 1e-3
        sb = torch.rand(n, 1, device=DEV, generator=g) * 0.01 + 1e-3
        bv = (torch.randn(n, device=DEV, generator=g)).to(torch.bfloat16) if bias else None
        f = torch.randn(m, k, device=DEV, generator=g).to(torch.bfloat16)
        h = (torch.randn(n, k, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
        fq, fs = ops.fp8_quantize_rowwise(f)
        hq, hs = ops.fp8_quantize_rowwise(h)
        out = {}
        try:
            lib.ao_gemm8_set_variant(32)
            for form in (1, 2):  # 1: never persistent, 2: persistent wherever the shape allows
                lib.ao_gemm8_set_tuning(6, form)
                out[form] = (ops.int_mm(a, b.t()), ops.int8_scaled_mm(a, sa, b, sb, bv), ops.fp8_scaled_mm(fq, hq.t(), fs, hs.t(), bv),
                             ops.fp8_mm_f32(fq, hq.t()))
        finally:
            lib.ao_gemm8_set_tuning(6, 0)
            lib.ao_gemm8_set_variant(0)
        for form in (2,):
            assert torch.equal(out[1][0], out[form][0]), (run, form, int((out[1][0] != out[form][0]).sum()))
            assert torch.equal(out[1][1].view(torch.int16), out[form][1].view(torch.int16)), (run, form)
            assert _rel(np_from_torch_bf16(out[form][2]), np_from_torch_bf16(out[1][2])) <= 1e-3
            assert _rel(out[form][3].cpu().numpy(), out[1][3].cpu().numpy()) <= 1e-5
    assert lib.ao_gemm8_set_variant(0) == 0


# ---- round 5: the LDS-transposed epilogue, the 32-column tiles; round 6: 64-row slabs above 64 rows ------------------------------------------------
@pytest.mark.parametrize("kind", ["int8", "fp8"])
@pytest.mark.parametrize("m,n,k,bias", [(128, 1280, 8192, False), (200, 1296, 2048, True), (33, 8192, 1024, False), (128, 4096, 4096, True),
                                        (512, 1280, 8192, False), (97, 48, 3584, True)])
def test_rb8_every_tile_width_and_part_count(kind, m, n, k, bias):
    """The weight-streaming kernel over slab rows 64 / 128, tile widths 32 / 64 / 128 and 1 .. 16 K parts (one- and two-level write-through
    meetings) against the oracle: the same parts are added in the same order, so every run of a form gives the same bits (int8: every form the oracle's)."""
    from ao_amd import _lib

    lib = _lib.lib()
    x = _randn_bf16((m, k), 31 * m + k)
    w = _randn_bf16((n, k), 37 * n + k, 0.05)
    b = _randn_bf16((n,), 7) if bias else None
    bd = None if b is None else b.to(DEV)
    if kind == "int8":
        xq, xs = ops.int8_quantize_rowwise(x.to(DEV))
        wq, ws = ops.int8_quantize_rowwise(w.to(DEV))
        run = lambda: ops.int8_scaled_mm(xq, xs, wq, ws, bd)  # noqa: E731
        y_ref = I.linear(x.float().numpy(), w.float().numpy(), None if b is None else b.float().numpy())
    else:
        xq, xs = ops.fp8_quantize_rowwise(x.to(DEV))
        wq, ws = ops.fp8_quantize_rowwise(w.to(DEV))
        run = lambda: ops.fp8_scaled_mm(xq, wq.t(), xs, ws.t(), bd)  # noqa: E731
        y_ref = F.linear(x.float().numpy(), w.float().numpy(), None if b is None else b.float().numpy())
    outs = {}
    try:
        lib.ao_gemm8_set_variant(101)  # always the weight-streaming kernel
        for bm in ((64, 128) if m > 64 else (64,)):  # (round 6) slab rows: 64-row slabs serve M > 64 too; 128-row slabs never run M <= 64
            for bn in (32, 64, 128):
                for split in (1, 2, 5, 16):
                    lib.ao_gemm8_set_tuning(3, bm)
                    lib.ao_gemm8_set_tuning(1, bn)
                    lib.ao_gemm8_set_tuning(2, split)
                    outs[(bm, bn, split)] = run().clone()
                    assert torch.equal(run(), outs[(bm, bn, split)])
    finally:
        lib.ao_gemm8_set_variant(0)
        for key in (1, 2, 3):
            lib.ao_gemm8_set_tuning(key, 0)
    torch.cuda.synchronize()
    yn = np_from_torch_bf16(next(iter(outs.values())))
    if kind == "int8":
        for key, y in outs.items():
            assert np.array_equal(np_from_torch_bf16(y), y_ref), f"int8 not bit-exact at {key}"
    else:
        for key, y in outs.items():
            assert _rel(np_from_torch_bf16(y), y_ref) <= 1e-3, f"fp8 off at {key}"
    assert np.isfinite(yn).all()


@pytest.mark.parametrize("m,n,k,bias", [(1024, 7168, 8192, False), (768, 1280, 4096, True), (512, 1024, 4096, False), (300, 528, 2048, True)])
@pytest.mark.parametrize("variant", [33])  # the 256 x 128 form (the 256 x 256 kernel's K parts were never dispatched: removed in round 6)
def test_gemm8_p8_split_k_every_part_count(m, n, k, bias, variant):
    """gemm8_p8_kernel / gemm8_p8h_kernel with the K range shared among 1 .. 16 workgroups per tile (split_k_meet2 in batches of 8 registers: the
    parts parked through to memory, summed in part order by the last arriver).  int8: integer partial sums, so every part count gives the
    UNSPLIT kernel's bits (int32 output and the scaled bf16 epilogue -- the oracle's bits); fp8: fp32 partial sums in a fixed order --
    reproducible run to run, <= 1e-3 from the oracle, the raw fp32 output within fp32 summation noise of the unsplit one."""
    from ao_amd import _lib

    lib = _lib.lib()
    x = _randn_bf16((m, k), 41 * m + k)
    w = _randn_bf16((n, k), 43 * n + k, 0.05)
    b = _randn_bf16((n,), 9) if bias else None
    bd = None if b is None else b.to(DEV)
    xq8, xs8 = ops.int8_quantize_rowwise(x.to(DEV))
    wq8, ws8 = ops.int8_quantize_rowwise(w.to(DEV))
    xqf, xsf = ops.fp8_quantize_rowwise(x.to(DEV))
    wqf, wsf = ops.fp8_quantize_rowwise(w.to(DEV))
    outs = {}
    try:
        lib.ao_gemm8_set_variant(8)  # the two-stage tile kernel
        want32 = ops.int_mm(xq8, wq8.t()).clone()
        want8 = ops.int8_scaled_mm(xq8, xs8, wq8, ws8, bd).clone()
        wantf = ops.fp8_scaled_mm(xqf, wqf.t(), xsf, wsf.t(), bd).clone()
        lib.ao_gemm8_set_variant(variant)  # always the phase-interleaved kernel of that tile
        for split in (1, 2, 3, 4, 5, 8, 13, 16):
            lib.ao_gemm8_set_tuning(7, split)
            for rep in range(2):
                outs[(split, rep)] = (ops.int_mm(xq8, wq8.t()).clone(), ops.int8_scaled_mm(xq8, xs8, wq8, ws8, bd).clone(),
                                      ops.fp8_scaled_mm(xqf, wqf.t(), xsf, wsf.t(), bd).clone(), ops.fp8_mm_f32(xqf, wqf.t()).clone())
    finally:
        lib.ao_gemm8_set_variant(0)
        lib.ao_gemm8_set_tuning(7, 0)
    torch.cuda.synchronize()
    i32, y8, yf, raw = outs[(1, 0)]
    assert torch.equal(i32, want32), "the unsplit 256 x 256 kernel differs from the two-stage tile kernel"
    # the CPU oracle where it takes seconds; the BASELINE-size shape is held to the unsplit kernels (themselves held to the oracle by
    # tests/test_baseline_scale_gpu.py) -- numpy's integer matmul would take minutes there
    small = m * n * k <= 1 << 32
    y8_ref = I.linear(x.float().numpy(), w.float().numpy(), None if b is None else b.float().numpy()) if small else np_from_torch_bf16(want8)
    yf_ref = F.linear(x.float().numpy(), w.float().numpy(), None if b is None else b.float().numpy()) if small else np_from_torch_bf16(wantf)
    for (split, rep), (a, c, d, e) in outs.items():
        assert torch.equal(a, i32), f"int32 differs at {split} parts"
        assert torch.equal(c, y8), f"int8 epilogue differs at {split} parts"
        assert torch.equal(d, outs[(split, 0)][2]) and torch.equal(e, outs[(split, 0)][3]), f"fp8 not reproducible at {split} parts"
        assert _rel(np_from_torch_bf16(d), yf_ref) <= 1e-3, f"fp8 off at {split} parts"
        assert float((e - raw).abs().max()) <= 1e-4 * float(raw.abs().max()), f"fp8 raw sums off at {split} parts"
    assert np.array_equal(np_from_torch_bf16(y8), y8_ref)


def test_empty_inputs_empty_groups_and_malformed_offsets():
    """The edges the reference's tests visit: zero rows through every cast and linear, grouped products whose groups are all empty or end before
    M_total (the rows past offs[-1]: zero-filled by the two-launch entry points), and offsets a router would never send -- decreasing, past
    M_total, negative -- which must not fault (the kernels clamp them into [0, M_total] and make them monotone)."""
    n, k, E = 64, 256, 4
    w = _randn_bf16((n, k), 1, 0.05).to(DEV)
    wq8, ws8 = ops.int8_quantize_rowwise(w)
    wqf, wsf = ops.fp8_quantize_rowwise(w)
    x0 = torch.zeros(0, k, dtype=torch.bfloat16, device=DEV)
    assert [tuple(v.shape) for v in ops.int8_quantize_rowwise(x0)] == [(0, k), (0, 1)]
    assert [tuple(v.shape) for v in ops.fp8_quantize_rowwise(x0)] == [(0, k), (0, 1)]
    assert [tuple(v.shape) for v in ops.mxfp8_quantize(x0, "rceil")] == [(0, k), (0, k // 32)]
    assert tuple(ops.int8_linear(x0, wq8, ws8).shape) == (0, n) and tuple(ops.fp8_linear(x0, wqf, wsf).shape) == (0, n)
    xq, xs = ops.int8_quantize_rowwise(_randn_bf16((3, k), 2).to(DEV))
    fq, fs = ops.fp8_quantize_rowwise(_randn_bf16((3, k), 3).to(DEV))
    assert tuple(ops.int8_scaled_mm(xq[:0], xs[:0], wq8, ws8).shape) == (0, n)
    assert tuple(ops.fp8_scaled_mm(fq[:0], wqf.t(), fs[:0], wsf.t()).shape) == (0, n)
    zeros = torch.zeros(E, dtype=torch.int32, device=DEV)
    we = _randn_bf16((E, n, k), 4, 0.1).to(DEV)
    weq, wes = ops.mxfp8_quantize(we, "rceil")
    a = _randn_bf16((8, k), 5).to(DEV)
    aq, asc = ops.mxfp8_quantize(a, "rceil")
    assert tuple(ops.mxfp8_grouped_mm(aq[:0], asc[:0], weq, wes, zeros).shape) == (0, n)
    assert float(ops.mxfp8_grouped_mm(aq, asc, weq, wes, zeros).abs().max()) == 0.0
    short = ops.mxfp8_grouped_mm(aq, asc, weq, wes, torch.tensor([2, 2, 5, 5], dtype=torch.int32, device=DEV))
    assert float(short[5:].abs().max()) == 0.0 and float(short[:5].abs().max()) > 0.0
    k2 = 512
    we2 = _randn_bf16((E, n, k2), 6, 0.1).to(DEV)
    we2q, we2s = ops.mxfp8_quantize(we2, "rceil")
    a2 = _randn_bf16((8, k2), 7).to(DEV)
    assert tuple(ops.mxfp8_grouped_mm_dyn(a2, we2q, we2s, zeros).shape) == (8, n)
    assert tuple(ops.mxfp8_grouped_mm_dyn(a2[:0], we2q, we2s, zeros).shape) == (0, n)
    assert [tuple(v.shape) for v in ops.mxfp8_grouped_mm_pair(a2, we2q, we2s, we2q, we2s, zeros)] == [(8, n), (8, n)]
    wgq, wgs = ops.fp8_quantize_rowwise(we.reshape(E * n, k))
    assert float(ops.fp8_grouped_mm(fq, fs, wgq.reshape(E, n, k), wgs.reshape(E, n), zeros).abs().max()) == 0.0
    assert tuple(ops.fp8_grouped_mm(fq[:0], fs[:0], wgq.reshape(E, n, k), wgs.reshape(E, n), zeros).shape) == (0, n)
    for bad in ([5, 2, 8, 8], [2, 4, 6, 100], [-3, 4, 6, 8]):
        offs = torch.tensor(bad, dtype=torch.int32, device=DEV)
        assert tuple(ops.mxfp8_grouped_mm(aq, asc, weq, wes, offs).shape) == (8, n)
        assert tuple(ops.mxfp8_grouped_mm_dyn(a2, we2q, we2s, offs).shape) == (8, n)
        torch.cuda.synchronize()  # a fault would surface here
    with pytest.raises(ValueError, match="multiple of 8"):
        ops.int8_quantize_rowwise(_randn_bf16((2, 100), 8).to(DEV))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.int8_quantize_rowwise(_randn_bf16((2, 128), 9))
