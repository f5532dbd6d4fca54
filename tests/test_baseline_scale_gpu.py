"""GPU parity at the sizes BASELINE.json quotes ("parity holes at BASELINE scale"), HIP path vs the oracle:

* int4 tinygemm linear at the merged gate_up_proj width (28672 x 4096, M = 1, g128) -- the headline's biggest launch;
* MXFP8 grouped GEMM on the Mixtral-8x7B expert shapes (E = 8, 14336 x 4096 and 4096 x 14336) with the seeded multinomial
  group sizes of SURVEY.md 8(d) (and the uniform 16-rows-per-expert case);
* int8 dynamic / fp8 rowwise linears at M = 2048 on the four Llama-3-8B shapes and the Llama-3-70B TP=8 shards;
* fp8_dynamic_linear (cast fused into the matmul) directly against oracle/fp8_ref;
* the K-sharded (row-parallel) 8-bit protocol of ao_amd/parallel.py against the UNSHARDED oracle linear.

The oracle side is bounded by sampling output rows / columns (the kernels still run the full shape): oracle/lowbit_ref.c
(OpenMP; itself pinned to the numpy oracles in tests/test_oracle_c.py) or numpy on the sample.
"""
import numpy as np
import pytest
import torch

from conftest import np_from_torch_bf16, torch_bf16_from_f32
from oracle import bf16, c_ref, fp8_ref as F, int4_ref as R, int8_ref as I, mx_ref as MX

pytestmark = pytest.mark.gpu

from ao_amd import _lib, ops  # noqa: E402

DEV = "cuda"


def _rel(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def _randn_bf16(shape, seed, scale=1.0, device="cpu"):
    g = torch.Generator(device=device).manual_seed(seed)
    return (torch.randn(*shape, generator=g, device=device) * scale).to(torch.bfloat16)


def _bits(t):
    """bf16 torch tensor -> uint16 numpy"""
    return t.view(torch.int16).cpu().numpy().view(np.uint16)


# ---- int4: the merged gate_up_proj launch of the headline ---------------------------------------------------------
@pytest.mark.parametrize("n,k,m", [(28672, 4096, 1), (14336, 4096, 1), (4096, 14336, 1), (28672, 4096, 3),
                                   (4096, 12288, 1), (5120, 13824, 1)])  # round 6: the 8- and 9-block straight-line forms (K of other models)
def test_int4_mm_baseline_shapes_vs_oracle(n, k, m):
    g = 128
    w = _randn_bf16((n, k), n + k, 0.02, DEV)
    x = _randn_bf16((m, k), n + k + 1, 1.0, DEV)
    qdata, sz = ops.int4_quantize_tinygemm(w, g)
    y = np_from_torch_bf16(ops.weight_int4pack_mm(x, qdata, g, sz))
    # the weight prep itself against the numpy oracle on a slab of rows (bit-exact), the matmul against the C port of
    # the reference dequant -> matmul path on the GPU-produced (= oracle-identical) packed weight
    wn = w[:64].float().cpu().numpy()
    s, z = R.choose_qparams_tinygemm(wn, g)
    q = R.quantize_tinygemm(wn, s, z, g)
    assert np.array_equal(qdata[:8].cpu().numpy(), R.convert_weight_to_int4pack(R.nibble_pack(q)))
    y_ref = bf16.from_bits(c_ref.int4_linear(_bits(x), qdata.cpu().numpy(), _bits(sz), n, k, g))
    assert _rel(y, y_ref) <= 1e-3
    assert np.all(np.abs(y - y_ref) <= np.abs(y_ref) * 2.0 ** -7 + 1e-6)
    assert np.mean(y == y_ref) > 0.97


# ---- int4 at bs = 128: the BASELINE shapes through the DEFAULT dispatch, i.e. the launches bench.py's int4_bs128 config times ----
@pytest.mark.parametrize("n,k,g,kernel", [(14336, 4096, 128, "int4_mm_w32_kernel"), (4096, 14336, 128, "int4_mm_rb_kernel"),
                                          (4096, 4096, 128, "int4_mm_rb_kernel"), (6144, 4096, 128, "int4_mm_rb_kernel"),
                                          (14336, 4096, 32, "int4_mm_w32_kernel"), (4096, 14336, 32, "int4_mm_rb_kernel")])
def test_int4_mm_bs128_baseline_shapes_default_dispatch(n, k, g, kernel):
    """int4_tile_packed_to_4d_tensor.py:243-299 at M = 128 on gate / up, down, o and qkv of Llama-3-8B: no forced mode, the kernel the
    product dispatch picks is the one named here (and by bench.py's roofline entry), output against the C port of the reference's
    dequant -> bf16 matmul path on a sample of columns, every row."""
    from ao_amd import _lib

    m = 128
    assert _lib.lib().ao_int4_mm_kernel_name(m, n, k, g).decode() == kernel
    w = _randn_bf16((n, k), n + k + g, 0.02, DEV)
    x = _randn_bf16((m, k), n + k + g + 1, 1.0, DEV)
    qdata, sz = ops.int4_quantize_tinygemm(w, g)
    y = np_from_torch_bf16(ops.weight_int4pack_mm(x, qdata, g, sz))
    # oracle on every 9th 16-row tile of the packed weight (+ the last): qdata dim 0 is N / 8 (the C port reads it in pairs: one 16-row
    # MFMA tile = two 8-row blocks), scales dim 1 is N
    tiles = np.unique(np.concatenate([np.arange(0, n // 16, 9), [n // 16 - 1]]))
    blocks = (tiles[:, None] * 2 + np.arange(2)[None, :]).reshape(-1)
    cols = (tiles[:, None] * 16 + np.arange(16)[None, :]).reshape(-1)
    bi, ci = torch.from_numpy(blocks).to(DEV), torch.from_numpy(cols).to(DEV)
    y_ref = bf16.from_bits(c_ref.int4_linear(_bits(x), qdata[bi].cpu().numpy(), _bits(sz[:, ci].contiguous()), len(cols), k, g))
    ys = y[:, cols]
    assert _rel(ys, y_ref) <= 1e-3
    assert np.all(np.abs(ys - y_ref) <= np.abs(y_ref) * 2.0 ** -7 + np.abs(y_ref).max() * 2.0 ** -12)
    assert np.mean(ys == y_ref) > 0.9


# ---- int4 at 9 .. 16 rows: round 6 moved weights of >= 288 n-tiles from the 16-row per-tile build to the batched kernel's 16-row slabs with K parts ----
@pytest.mark.parametrize("m,n,k,g,kernel", [(12, 14336, 4096, 128, "int4_mm_rb_kernel"), (16, 14336, 4096, 128, "int4_mm_rb_kernel"), (9, 6144, 4096, 128, "int4_mm_rb_kernel"),
                                            (16, 4608, 3584, 32, "int4_mm_rb_kernel"), (16, 5120, 13824, 64, "int4_mm_rb_kernel"), (16, 4096, 4096, 128, "int4_mm_kernel"),
                                            (8, 14336, 4096, 128, "int4_mm_kernel"), (6, 28672, 4096, 128, "int4_mm_rb_kernel")])
def test_int4_mm_small_batch_default_dispatch(m, n, k, g, kernel):
    """int4_tile_packed_to_4d_tensor.py:243-299 at 5 .. 16 rows through the default dispatch (profiles/int4_forms_r06.jsonl: which kernel serves
    which weight width); output against the C port of the reference's dequant -> bf16 matmul path on a sample of columns, every row."""
    assert _lib.lib().ao_int4_mm_kernel_name(m, n, k, g).decode() == kernel
    w = _randn_bf16((n, k), n + k + g + m, 0.02, DEV)
    x = _randn_bf16((m, k), n + k + g + m + 1, 1.0, DEV)
    qdata, sz = ops.int4_quantize_tinygemm(w, g)
    y = np_from_torch_bf16(ops.weight_int4pack_mm(x, qdata, g, sz))
    assert np.array_equal(np_from_torch_bf16(ops.weight_int4pack_mm(x, qdata, g, sz)), y)  # K parts meet in part order: reproducible
    tiles = np.unique(np.concatenate([np.arange(0, n // 16, 7), [n // 16 - 1]]))
    blocks = (tiles[:, None] * 2 + np.arange(2)[None, :]).reshape(-1)
    cols = (tiles[:, None] * 16 + np.arange(16)[None, :]).reshape(-1)
    bi, ci = torch.from_numpy(blocks).to(DEV), torch.from_numpy(cols).to(DEV)
    y_ref = bf16.from_bits(c_ref.int4_linear(_bits(x), qdata[bi].cpu().numpy(), _bits(sz[:, ci].contiguous()), len(cols), k, g))
    ys = y[:, cols]
    assert _rel(ys, y_ref) <= 1e-3
    assert np.all(np.abs(ys - y_ref) <= np.abs(y_ref) * 2.0 ** -7 + np.abs(y_ref).max() * 2.0 ** -12)
    assert np.mean(ys == y_ref) > 0.9


# ---- MXFP8 grouped GEMM: Mixtral-8x7B expert shapes ------------------------------------------------------------------
def _mixtral_offs(kind, rows=128, experts=8, seed=0):
    if kind == "uniform16":
        sizes = np.full(experts, rows // experts)
    else:  # offs = 32 * round(multinomial) cumulative (seeded), 64 tokens x top-2 = 128 rows
        rng = np.random.default_rng(seed)
        sizes = 32 * rng.multinomial(rows // 32, np.full(experts, 1.0 / experts))
    return np.cumsum(sizes).astype(np.int32), sizes


@pytest.mark.parametrize("kind", ["multinomial32", "uniform16"])
@pytest.mark.parametrize("n,k", [(14336, 4096), (4096, 14336)])
def test_mxfp8_grouped_mm_mixtral_vs_oracle(n, k, kind):
    E = 8
    offs, sizes = _mixtral_offs(kind)
    M = int(offs[-1])
    a = _randn_bf16((M, k), 31, 1.0, DEV)
    w = _randn_bf16((E, n, k), 32, 0.02, DEV)
    w_d, w_s = ops.mxfp8_quantize(w, "rceil")
    del w
    a_d, a_s = ops.mxfp8_quantize(a, "rceil")
    y = np_from_torch_bf16(ops.mxfp8_grouped_mm(a_d, a_s, w_d, w_s, torch.from_numpy(offs).to(DEV)))
    # oracle on a sample of output columns (every 7th 16-wide tile and the last one), every row
    tiles = np.unique(np.concatenate([np.arange(0, n // 16, 7), [n // 16 - 1]]))
    cols = (tiles[:, None] * 16 + np.arange(16)[None, :]).reshape(-1)
    ci = torch.from_numpy(cols).to(DEV)
    wq_s = w_d.view(torch.uint8)[:, ci].cpu().numpy()
    ws_s = w_s.view(torch.uint8)[:, ci].cpu().numpy()
    y_ref = bf16.from_bits(c_ref.mxfp8_grouped_mm(_bits(a), wq_s, ws_s, offs))
    # the cast of the activations on the GPU is bit-exact against the oracle's (so both sides multiply the same codes)
    aq, asc = MX.to_mx(a.float().cpu().numpy(), MX.RCEIL)
    assert np.array_equal(a_d.view(torch.uint8).cpu().numpy(), aq) and np.array_equal(a_s.view(torch.uint8).cpu().numpy(), asc)
    ys = y[:, cols]
    assert _rel(ys, y_ref) <= 1e-3
    # one bf16 ulp + fp32 accumulation-order noise relative to the column's magnitude
    scale = np.abs(y_ref).max() + 1e-30
    assert np.all(np.abs(ys - y_ref) <= np.abs(y_ref) * 2.0 ** -7 + scale * 2.0 ** -12)


# ---- int8 dynamic / fp8 rowwise at M = 2048 ---------------------------------------------------------------------------
LLAMA8B = [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336)]
LLAMA70B_TP8 = [(1024, 8192), (8192, 1024), (7168, 8192), (8192, 3584)]


def _sample(n, m, seed):
    rng = np.random.default_rng(seed)
    rows = np.unique(np.concatenate([[0, m - 1], rng.integers(0, m, 14)]))
    tiles = np.unique(np.concatenate([[0, n // 16 - 1], rng.integers(0, n // 16, 30)]))
    cols = (tiles[:, None] * 16 + np.arange(16)[None, :]).reshape(-1)
    return rows, cols


@pytest.mark.parametrize("n,k", LLAMA8B + LLAMA70B_TP8)
def test_int8_dynamic_linear_m2048_vs_oracle(n, k):
    m = 2048
    x = _randn_bf16((m, k), n + 3, 1.0, DEV)
    w = _randn_bf16((n, k), k + 5, 0.02, DEV)
    wq, ws = ops.int8_quantize_rowwise(w)
    xq, xs = ops.int8_quantize_rowwise(x)
    y = ops.int8_scaled_mm(xq, xs, wq, ws)
    rows, cols = _sample(n, m, n + k)
    ri, ci = torch.from_numpy(rows).to(DEV), torch.from_numpy(cols).to(DEV)
    y_ref = c_ref.int8_dynamic_linear(_bits(x[ri]), wq[ci].cpu().numpy(), ws.flatten()[ci].cpu().numpy())
    got = _bits(y[ri][:, ci])
    assert np.array_equal(got, y_ref)  # int32 accumulation + the reference's two roundings: bit for bit


@pytest.mark.parametrize("n,k", [(14336, 4096), (4096, 14336), (6144, 4096), (4096, 4096)])
def test_int8_dynamic_linear_benched_chunk_vs_oracle(n, k):
    """The launches bench.py's int8 config times: one 16384-row chunk per call (int8/kernels.py:114-144 + int8_tensor.py:305-359),
    cast and matmul through the default dispatch; oracle on sampled rows x columns, bit for bit."""
    m = 16384
    x = _randn_bf16((m, k), n + 11, 1.0, DEV)
    w = _randn_bf16((n, k), k + 13, 0.02, DEV)
    wq, ws = ops.int8_quantize_rowwise(w)
    xq, xs = ops.int8_quantize_rowwise(x)
    y = ops.int8_scaled_mm(xq, xs, wq, ws)
    rng = np.random.default_rng(n + k)
    rows = np.unique(np.concatenate([[0, 255, 256, m - 257, m - 1], rng.integers(0, m, 40)]))
    tiles = np.unique(np.concatenate([[0, n // 16 - 1], rng.integers(0, n // 16, 30)]))
    cols = (tiles[:, None] * 16 + np.arange(16)[None, :]).reshape(-1)
    ri, ci = torch.from_numpy(rows).to(DEV), torch.from_numpy(cols).to(DEV)
    y_ref = c_ref.int8_dynamic_linear(_bits(x[ri]), wq[ci].cpu().numpy(), ws.flatten()[ci].cpu().numpy())
    assert np.array_equal(_bits(y[ri][:, ci]), y_ref)


@pytest.mark.parametrize("n,k", LLAMA8B + LLAMA70B_TP8)
def test_fp8_rowwise_linear_m2048_vs_oracle(n, k):
    m = 2048
    x = _randn_bf16((m, k), n + 7, 1.0, DEV)
    w = _randn_bf16((n, k), k + 9, 0.02, DEV)
    wq, ws = ops.fp8_quantize_rowwise(w)
    xq, xs = ops.fp8_quantize_rowwise(x)
    y = ops.fp8_scaled_mm(xq, wq.t(), xs, ws.t())
    rows, cols = _sample(n, m, n + k + 1)
    ri, ci = torch.from_numpy(rows).to(DEV), torch.from_numpy(cols).to(DEV)
    y_ref = bf16.from_bits(c_ref.fp8_rowwise_linear(_bits(x[ri]), wq.view(torch.uint8)[ci].cpu().numpy(), ws.flatten()[ci].cpu().numpy()))
    got = np_from_torch_bf16(y[ri][:, ci])
    assert _rel(got, y_ref) <= 1e-3
    assert np.all(np.abs(got - y_ref) <= np.abs(y_ref) * 2.0 ** -7 + np.abs(y_ref).max() * 2.0 ** -14)


# Row counts off the power-of-two grid, where round 6 moved dispatch seams (profiles/midm_offgrid_r06.jsonl): two tile rows with K parts on the
# 256 x 128 kernel (ragged last tile row), an odd count of 128-row slabs on the weight-streaming kernel, the quantisation hole at 1152 rows.
OFF_GRID = [(320, 4096, 14336, "gemm8_p8h_kernel"), (384, 7168, 8192, "gemm8_p8h_kernel"), (449, 7168, 8192, "gemm8_p8h_kernel"),
            (576, 6144, 4096, "rb8_kernel"), (640, 6144, 4096, "rb8_kernel"), (1152, 7168, 8192, "gemm8_dma_kernel<128x128>"),
            # shapes of other models (Qwen2-7B, Llama-2-13B) at decode / small-batch rows, where round 6 moved what the decode kernels leave to
            # the weight-streaming kernel, and 9 .. 16 rows where the decode kernel now picks its ring by occupancy
            (24, 37888, 3584, "rb8_kernel"), (64, 37888, 3584, "rb8_kernel"), (8, 5120, 13824, "rb8_kernel"), (24, 3584, 18944, "rb8_kernel"),
            (16, 8192, 1024, "dec8_kernel"), (12, 4096, 4096, "dec8_kernel"), (16, 15360, 5120, "dec8_kernel"), (9, 3584, 3584, "dec8_kernel")]


@pytest.mark.parametrize("int8", [0, 1])
@pytest.mark.parametrize("m,n,k,kernel", OFF_GRID)
def test_8bit_linear_off_grid_rows_vs_oracle(m, n, k, kernel, int8):
    """float8_tensor.py:410-458 / int8_tensor.py:305-359 through the default dispatch at row counts between the tuned grid points; the kernel each
    takes is asserted (a moved rule has to show up here); int8 bit for bit, fp8 within 1e-3."""
    assert _lib.lib().ao_gemm8_kernel_name(int8, m, n, k).decode() == kernel
    x = _randn_bf16((m, k), n + m, 1.0, DEV)
    w = _randn_bf16((n, k), k + m, 0.02, DEV)
    rows, cols = _sample(n, m, n + k + m)
    ri, ci = torch.from_numpy(rows).to(DEV), torch.from_numpy(cols).to(DEV)
    if int8:
        wq, ws = ops.int8_quantize_rowwise(w)
        xq, xs = ops.int8_quantize_rowwise(x)
        y = ops.int8_scaled_mm(xq, xs, wq, ws)
        y_ref = c_ref.int8_dynamic_linear(_bits(x[ri]), wq[ci].cpu().numpy(), ws.flatten()[ci].cpu().numpy())
        assert np.array_equal(_bits(y[ri][:, ci]), y_ref)
    else:
        wq, ws = ops.fp8_quantize_rowwise(w)
        xq, xs = ops.fp8_quantize_rowwise(x)
        y = ops.fp8_scaled_mm(xq, wq.t(), xs, ws.t())
        y_ref = bf16.from_bits(c_ref.fp8_rowwise_linear(_bits(x[ri]), wq.view(torch.uint8)[ci].cpu().numpy(), ws.flatten()[ci].cpu().numpy()))
        got = np_from_torch_bf16(y[ri][:, ci])
        assert _rel(got, y_ref) <= 1e-3
        assert np.all(np.abs(got - y_ref) <= np.abs(y_ref) * 2.0 ** -7 + np.abs(y_ref).max() * 2.0 ** -14)


@pytest.mark.parametrize("m,n,k", [(1, 1024, 8192), (1, 8192, 3584), (4, 4096, 4096), (2, 256, 1024)])
def test_fp8_dynamic_linear_vs_oracle_directly(m, n, k):
    if not ops.dynamic_linear_fits(m, n, k):
        pytest.skip("shape outside the fused kernel's range")
    x = _randn_bf16((m, k), 3 * n + m, 1.0)
    w = _randn_bf16((n, k), 5 * k + m, 0.02)
    bias = _randn_bf16((n,), 17)
    wq, ws = ops.fp8_quantize_rowwise(w.to(DEV))
    y = np_from_torch_bf16(ops.fp8_dynamic_linear(x.to(DEV), wq, ws, bias.to(DEV)))
    y_ref = F.linear(x.float().numpy(), w.float().numpy(), bias.float().numpy())
    assert _rel(y, y_ref) <= 1e-3
    assert np.all(np.abs(y - y_ref) <= np.abs(y_ref) * 2.0 ** -7 + np.abs(y_ref).max() * 2.0 ** -14)


# ---- K-sharded (row-parallel) 8-bit linears: the unsharded oracle from sharded operands ---------------------------------
@pytest.mark.parametrize("kind", ["int8", "fp8"])
@pytest.mark.parametrize("world", [2, 8])
def test_row_parallel_blocks_match_unsharded_oracle(kind, world):
    """Every rank's steps of ao_amd.parallel.RowParallelLinear's exact protocol, run on one GPU and combined the way the
    collectives would (max of the amax, integer / fp32 sum of the accumulators), against the UNSHARDED oracle linear."""
    from ao_amd.parallel import shard_bounds

    n, k, m = 512, 3584, 37  # 3584 / 128 = 28 k-units: uneven shards at world 8
    x = _randn_bf16((m, k), 41, 1.0)
    x[:, 777] *= 32.0  # the row amax sits in one shard
    w = _randn_bf16((n, k), 42, 0.05)
    bias = _randn_bf16((n,), 43)
    ref = I if kind == "int8" else F
    want = ref.linear(x.float().numpy(), w.float().numpy(), bias.float().numpy())
    xd, wd = x.to(DEV), w.to(DEV)
    wq, ws = (ops.int8_quantize_rowwise if kind == "int8" else ops.fp8_quantize_rowwise)(wd)
    spans = [shard_bounds(k, world, r, 128) for r in range(world)]
    amax = torch.stack([ops.rowwise_amax(xd[:, k0:k1]) for k0, k1 in spans]).amax(dim=0)  # all_reduce(MAX)
    assert torch.equal(amax, xd.float().abs().amax(dim=1))
    acc = None
    for k0, k1 in spans:
        if kind == "int8":
            xq, xs = ops.int8_quantize_rowwise_amax(xd[:, k0:k1], amax)  # strided view: no copy
            part = ops.int_mm(xq, wq[:, k0:k1].contiguous().t())
        else:
            xq, xs = ops.fp8_quantize_rowwise_amax(xd[:, k0:k1], amax)
            part = ops.fp8_mm_f32(xq, wq[:, k0:k1].contiguous().t())
        acc = part if acc is None else acc + part  # all_reduce(SUM)
    # the shard casts are the shards of the unsharded cast, the scale is the unsharded scale
    xq_full, xs_full = (ops.int8_quantize_rowwise if kind == "int8" else ops.fp8_quantize_rowwise)(xd)
    assert torch.equal(xs, xs_full)
    assert torch.equal(xq.view(torch.uint8), xq_full.view(torch.uint8)[:, spans[-1][0]:])
    y = (ops.int8_scale_epilogue if kind == "int8" else ops.fp8_scale_epilogue)(acc, xs, ws, bias.to(DEV))
    yn = np_from_torch_bf16(y)
    if kind == "int8":
        assert np.array_equal(yn, want)
    else:
        assert _rel(yn, want) <= 1e-3
        assert np.all(np.abs(yn - want) <= np.abs(want) * 2.0 ** -7 + np.abs(want).max() * 2.0 ** -14)
    # and it equals the unsharded product path on the same GPU
    y_full = (ops.int8_scaled_mm(xq_full, xs_full, wq, ws, bias.to(DEV)) if kind == "int8"
              else ops.fp8_scaled_mm(xq_full, wq.t(), xs_full, ws.t(), bias.to(DEV)))
    if kind == "int8":
        assert torch.equal(y, y_full)
    else:
        assert _rel(yn, np_from_torch_bf16(y_full)) <= 1e-3


# ---- Float8Tensor _grouped_mm (rowwise) and the cached MXFP8 expert weights ------------------------------------------------
@pytest.mark.parametrize("sizes,n,k", [([16, 16, 16, 16], 64, 512), ([32, 0, 5, 27], 256, 2048), ([1, 70, 3, 0], 144, 4096),
                                       ([0] * 65 + [40, 0, 9], 64, 256), ([1, 17, 33, 49, 0, 64, 65, 0], 80, 512),
                                       ([128, 0, 64, 200, 0, 0, 8, 0], 1024, 1024)])
def test_fp8_grouped_mm_vs_oracle(sizes, n, k):
    """torch._grouped_mm(x, Float8Tensor weight.transpose(-2, -1), offs) -- float8_tensor.py:1085-1122 -- against the numpy oracle
    (rowwise e4m3 cast of the tokens, per-expert rowwise scaled matmul)."""
    from ao_amd.quantization import Float8Tensor
    from ao_amd.quantization.float8_tensor import QuantizeTensorToFloat8Kwargs

    E, M = len(sizes), sum(sizes)
    a = _randn_bf16((M, k), 51)
    w = _randn_bf16((E, n, k), 52, 0.05)
    offs = torch.tensor(np.cumsum(sizes), dtype=torch.int32)
    wt = Float8Tensor.from_hp(w.to(DEV), act_quant_kwargs=QuantizeTensorToFloat8Kwargs())
    assert tuple(wt.shape) == (E, n, k) and tuple(wt.scale.shape) == (E, n, 1)
    y = torch._grouped_mm(a.to(DEV), wt.transpose(-2, -1), offs=offs.to(DEV))
    assert y.dtype == torch.bfloat16 and tuple(y.shape) == (M, n)
    wq, ws = F.quantize_rowwise(w.float().numpy().reshape(E * n, k))
    assert np.array_equal(wt.qdata.view(torch.uint8).cpu().numpy().reshape(E * n, k), wq)
    aq, a_s = F.quantize_rowwise(a.float().numpy())
    y_ref = F.grouped_mm(aq, wq.reshape(E, n, k), a_s, ws.reshape(E, n), offs.numpy())
    yn = np_from_torch_bf16(y)
    assert _rel(yn, y_ref) <= 1e-3
    assert np.all(np.abs(yn - y_ref) <= np.abs(y_ref) * 2.0 ** -7 + np.abs(y_ref).max() * 2.0 ** -14)


def test_fp8_grouped_mm_rejects_per_tensor_activations():
    """Float8Tensor _grouped_mm is PerRow x PerRow like the reference (float8_tensor.py:1098-1101): PerTensor activation kwargs fail with a
    message that names the reason instead of surfacing later in the kernel's scale-shape check."""
    from ao_amd.quantization import Float8Tensor
    from ao_amd.quantization.float8_tensor import QuantizeTensorToFloat8Kwargs
    from ao_amd.quantization.granularity import PerTensor

    w = _randn_bf16((2, 64, 256), 53, 0.05).to(DEV)
    wt = Float8Tensor.from_hp(w, act_quant_kwargs=QuantizeTensorToFloat8Kwargs(granularity=PerTensor()))
    a = _randn_bf16((32, 256), 54).to(DEV)
    offs = torch.tensor([16, 32], dtype=torch.int32, device=DEV)
    with pytest.raises(NotImplementedError, match="PerRow"):
        torch._grouped_mm(a, wt.transpose(-2, -1), offs=offs)


def test_mxfp8_moe_forward_with_cached_expert_weights():
    """_to_mxfp8_then_scaled_grouped_mm: casting the expert weights once (MXFP8ExpertWeights / cache_weights=True) gives the bits of
    casting them in every call (the reference's forward), and an in-place weight update invalidates the memo."""
    from ao_amd.prototype.mx import MXFP8ExpertWeights, _to_mxfp8_then_scaled_grouped_mm

    E, n, k = 4, 256, 1024
    a = _randn_bf16((96, k), 61).to(DEV)
    b_t = _randn_bf16((E, n, k), 62, 0.05).to(DEV).transpose(-2, -1)  # [E, K, N] view of [E, N, K]
    offs = torch.tensor([32, 32, 64, 96], dtype=torch.int32, device=DEV)
    want = _to_mxfp8_then_scaled_grouped_mm(a, b_t, offs)
    assert torch.equal(_to_mxfp8_then_scaled_grouped_mm(a, MXFP8ExpertWeights.from_hp(b_t), offs), want)
    assert torch.equal(_to_mxfp8_then_scaled_grouped_mm(a, b_t, offs, cache_weights=True), want)
    assert torch.equal(_to_mxfp8_then_scaled_grouped_mm(a, b_t, offs, cache_weights=True), want)  # memo hit
    b_t.mul_(2.0)  # in-place update bumps the version counter: the memo must not serve the old cast
    got = _to_mxfp8_then_scaled_grouped_mm(a, b_t, offs, cache_weights=True)
    assert torch.equal(got, _to_mxfp8_then_scaled_grouped_mm(a, b_t, offs)) and not torch.equal(got, want)
