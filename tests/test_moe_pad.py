"""MoE token-group padding: oracle vs the reference's outputs (CPU), HIP kernels vs oracle (GPU, bit exact)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, np_from_torch_bf16, torch_bf16_from_f32  # noqa: F401
from oracle import moe_ref as R
M = R  # the regrouping tests below name it M (R is a rank count there)

CASES = ["ragged", "aligned16", "single", "odd_dim"]


@pytest.fixture(scope="module")
def golden_moe():
    return np.load(os.path.join(GOLDEN, "moe_pad.npz"))


@pytest.mark.parametrize("case", CASES)
def test_oracle_matches_reference_outputs(golden_moe, case):
    d = golden_moe
    p, s, e = R.pad_token_groups(d[f"{case}_x"], d[f"{case}_offs"], int(d[f"{case}_align"]))
    assert np.array_equal(p, d[f"{case}_padded"]) and np.array_equal(s, d[f"{case}_starts"]) and np.array_equal(e, d[f"{case}_ends"])
    u = R.unpad_token_groups(p, d[f"{case}_offs"], s, d[f"{case}_x"].shape[0])
    assert np.array_equal(u, d[f"{case}_x"])


def test_oracle_unpad_size_mismatch_raises():
    x = np.zeros((4, 2), np.float32)
    p, s, _ = R.pad_token_groups(x, [1, 4], 4)
    with pytest.raises(RuntimeError, match="Unpad output size mismatch"):
        R.unpad_token_groups(p, [1, 4], s, 5)


# ---- GPU ---------------------------------------------------------------------------------
def _to_dev(a):
    """golden array -> device tensor (uint16 = bf16 bit patterns)"""
    if a.dtype == np.uint16:
        return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16).cuda()
    return torch.from_numpy(a.copy()).cuda()


def _bits(t):
    t = t.cpu().contiguous()
    return t.view(torch.int16).numpy().view(np.uint16) if t.dtype == torch.bfloat16 else t.numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_kernels_match_reference_outputs(golden_moe, case):
    from ao_amd import ops

    d = golden_moe
    x, offs, align = _to_dev(d[f"{case}_x"]), torch.from_numpy(d[f"{case}_offs"]).cuda(), int(d[f"{case}_align"])
    p, s, e = ops.fused_pad_token_groups(x, offs, align)
    assert np.array_equal(_bits(p), d[f"{case}_padded"])
    assert np.array_equal(s.cpu().numpy(), d[f"{case}_starts"]) and np.array_equal(e.cpu().numpy(), d[f"{case}_ends"])
    u = ops.fused_unpad_token_groups(p, offs, s, x.shape[0], align)
    assert torch.equal(u, x)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("groups,dim,align", [(8, 7168, 32), (128, 2048, 32), (200, 96, 16), (3, 5, 32)])
def test_kernels_vs_oracle_random_groups(groups, dim, align, dtype):
    """DeepSeek-V3-sized rows, more than 64 groups (two ballot passes), empty groups, rows that are not a multiple
    of 16 bytes; output rows past the last group must be zero even though the buffer starts uninitialised."""
    from ao_amd import ops
    from ao_amd.prototype.mx import pad_token_groups, unpad_token_groups

    rng = np.random.default_rng(groups * 1000 + dim)
    sizes = rng.integers(0, 70, size=groups)
    sizes[rng.integers(0, groups)] = 0
    ends = np.cumsum(sizes).astype(np.int32)
    tokens = int(ends[-1])
    x = torch.randn(tokens, dim, generator=torch.Generator().manual_seed(1)).to(dtype).cuda()
    offs = torch.from_numpy(ends).cuda()
    torch.full((tokens + groups * align + align, dim), float("nan"), device="cuda", dtype=dtype)  # dirty the allocator's block
    p, s, e = pad_token_groups(x, offs, align)
    xr = _bits(x)
    pr, sr, er = R.pad_token_groups(xr, ends, align)
    assert p.shape == pr.shape and np.array_equal(_bits(p), pr)
    assert np.array_equal(s.cpu().numpy(), sr) and np.array_equal(e.cpu().numpy(), er)
    u = unpad_token_groups(p, offs, s, tokens, align)
    assert torch.equal(u, x)
    # the dispatcher route (what a torch.ops.torchao.* call site would be retargeted to)
    import ao_amd.torch_ops  # noqa: F401

    p2, s2, e2 = torch.ops.ao_mi355.fused_pad_token_groups(x, offs, align)
    assert torch.equal(p2, p) and torch.equal(s2, s) and torch.equal(e2, e)
    assert torch.equal(torch.ops.ao_mi355.fused_unpad_token_groups(p2, offs, s2, tokens, align), x)


@pytest.mark.gpu
def test_padded_mxfp8_grouped_mm_round_trip():
    """pad -> quantise -> grouped GEMM on aligned groups -> unpad equals the grouped GEMM on the ragged groups
    (zero rows quantise to zero and never reach a real token's output)."""
    from ao_amd import ops
    from ao_amd.prototype.mx import _to_mxfp8_then_scaled_grouped_mm, pad_token_groups, unpad_token_groups

    E, N, K = 4, 64, 256
    ends = np.array([5, 5, 40, 77], dtype=np.int32)
    gen = torch.Generator().manual_seed(3)
    a = torch.randn(int(ends[-1]), K, generator=gen).to(torch.bfloat16).cuda()
    w = (torch.randn(E, N, K, generator=gen) * 0.1).to(torch.bfloat16).cuda()
    offs = torch.from_numpy(ends).cuda()
    direct = _to_mxfp8_then_scaled_grouped_mm(a, w.transpose(-2, -1), offs)
    pa, ps, pe = pad_token_groups(a, offs, 32)
    padded_out = _to_mxfp8_then_scaled_grouped_mm(pa, w.transpose(-2, -1), pe)
    assert torch.equal(unpad_token_groups(padded_out, offs, ps, a.shape[0], 32), direct)


@pytest.mark.gpu
def test_argument_errors():
    from ao_amd import ops

    x = torch.zeros(4, 8, device="cuda", dtype=torch.bfloat16)
    offs = torch.tensor([4], dtype=torch.int32, device="cuda")
    with pytest.raises(AssertionError, match="offsets must be int32"):
        ops.fused_pad_token_groups(x, offs.long(), 32)
    with pytest.raises(AssertionError, match="float32 or bfloat16"):
        ops.fused_pad_token_groups(x.half(), offs, 32)
    with pytest.raises(AssertionError, match="2d"):
        ops.fused_pad_token_groups(x[0], offs, 32)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.fused_pad_token_groups(x.cpu(), offs.cpu(), 32)
    p, s, e = ops.fused_pad_token_groups(x[:0], torch.tensor([0], dtype=torch.int32, device="cuda"), 32)  # no tokens
    assert p.shape == (32, 8) and not p.any() and s.item() == 0 and e.item() == 0


# ---- expert-parallel regrouping: generate_permute_indices / permute / unpermute (ep/kernels.py, permute.py, unpermute.py) -------------
def _permute_cases():
    d = np.load(os.path.join(GOLDEN, "moe_permute.npz"))
    ci = 0
    while f"c{ci}_meta" in d:
        E, R, align, dim, T, max_len = (int(v) for v in d[f"c{ci}_meta"])
        yield ci, d, E, R, align, dim, T, max_len
        ci += 1


def test_permute_oracle_matches_the_reference_fixtures():
    """oracle/moe_ref.py vs outputs of the reference's own generate_permute_indices(use_cpu=True) / row gather / row scatter"""
    n = 0
    for ci, d, E, R, align, dim, T, max_len in _permute_cases():
        idx, m_sizes, m_offsets = M.generate_permute_indices(d[f"c{ci}_counts"], E, R, max_len, align)
        assert np.array_equal(idx, d[f"c{ci}_idx"]) and np.array_equal(m_sizes, d[f"c{ci}_m_sizes"]) and np.array_equal(m_offsets, d[f"c{ci}_m_offsets"])
        assert np.array_equal(M.gather_rows(d[f"c{ci}_x"], idx), d[f"c{ci}_permuted"])
        assert np.array_equal(M.scatter_rows(d[f"c{ci}_y"], idx, T), d[f"c{ci}_unpermuted"])
        # properties: a permutation of the real rows plus padding; group sizes aligned and >= alignment
        real = idx[idx >= 0]
        assert sorted(real.tolist()) == list(range(T)) and (m_sizes % align == 0).all() and (m_sizes >= align).all()
        n += 1
    assert n >= 4


@pytest.mark.gpu
def test_permute_kernels_match_oracle_and_fixtures():
    import torch
    from ao_amd import ops
    from ao_amd.prototype import ep

    dev = torch.device("cuda", 0)
    for ci, d, E, R, align, dim, T, max_len in _permute_cases():
        counts = torch.from_numpy(d[f"c{ci}_counts"]).to(dev)
        idx, m_sizes, m_offsets = ops.generate_permute_indices(counts, E, R, max_len, align)
        assert np.array_equal(idx.cpu().numpy(), d[f"c{ci}_idx"])
        assert np.array_equal(m_sizes.cpu().numpy(), d[f"c{ci}_m_sizes"]) and np.array_equal(m_offsets.cpu().numpy(), d[f"c{ci}_m_offsets"])
        x = torch.from_numpy(d[f"c{ci}_x"].view(np.int16)).view(torch.bfloat16).to(dev)
        shape, xp, idx2, sizes2, offs2 = ep.permute_and_pad(x, counts, R, E, align)
        assert tuple(shape) == (T + 1, dim) and torch.equal(idx2, idx)
        assert np.array_equal(xp.view(torch.int16).cpu().numpy().view(np.uint16), d[f"c{ci}_permuted"])
        y = torch.from_numpy(d[f"c{ci}_y"].view(np.int16)).view(torch.bfloat16).to(dev)
        un = ep.unpermute_hp_fwd(y, idx, shape)
        assert np.array_equal(un.view(torch.int16).cpu().numpy().view(np.uint16), d[f"c{ci}_unpermuted"])
        # round trip: unpermute(permute(x)) == x
        assert torch.equal(ep.unpermute_hp_fwd(xp, idx, shape), x)
    # a larger seeded case against the oracle, odd row width (byte path), uint8 rows (e8m0 scales), many experts
    rng = np.random.default_rng(5)
    E, R, align = 96, 3, 32
    counts_np = rng.integers(0, 40, size=R * E).astype(np.int32)
    T = int(counts_np.sum())
    max_len = (T + E * align + align - 1) // align * align
    want_idx, want_sizes, want_offs = M.generate_permute_indices(counts_np, E, R, max_len, align)
    idx, sizes, offs = ops.generate_permute_indices(torch.from_numpy(counts_np).to(dev), E, R, max_len, align)
    assert np.array_equal(idx.cpu().numpy(), want_idx) and np.array_equal(sizes.cpu().numpy(), want_sizes) and np.array_equal(offs.cpu().numpy(), want_offs)
    for width, dt in ((45, np.uint8), (224, np.uint8), (7168, np.uint16)):
        x_np = rng.integers(0, np.iinfo(dt).max, size=(T, width)).astype(dt)
        xt = torch.from_numpy(x_np.view(np.int16 if dt == np.uint16 else np.uint8)).to(dev)
        got = ops.gather_rows(xt, idx).cpu().numpy().view(dt)
        assert np.array_equal(got, M.gather_rows(x_np, want_idx))
        back = ops.scatter_rows(ops.gather_rows(xt, idx), idx, T).cpu().numpy().view(dt)
        assert np.array_equal(back, x_np)


# ---- the 128 x 4 blocked layout of the E8M0 scales (torchao::mx_block_rearrange_2d_M_groups / to_blocked) --------------------------------
TB_CASES = ["one_block", "ragged", "wide", "tiny", "k4096"]
MG_CASES = ["groups", "single", "aligned", "k224"]


@pytest.fixture(scope="module")
def golden_blocked():
    return np.load(os.path.join(GOLDEN, "mx_blocked.npz"))


@pytest.mark.parametrize("case", TB_CASES)
def test_oracle_to_blocked_matches_reference(golden_blocked, case):
    from oracle import mx_ref

    d = golden_blocked
    assert np.array_equal(mx_ref.to_blocked(d[f"tb_{case}_in"]), d[f"tb_{case}_out"])


@pytest.mark.parametrize("case", MG_CASES)
def test_oracle_blocked_m_groups_matches_reference(golden_blocked, case):
    from oracle import mx_ref

    d = golden_blocked
    out, starts = mx_ref.to_blocked_2d_M_groups(d[f"mg_{case}_in"], d[f"mg_{case}_offs"])
    assert np.array_equal(out, d[f"mg_{case}_out"]) and np.array_equal(starts, d[f"mg_{case}_starts"])


@pytest.mark.gpu
@pytest.mark.parametrize("case", TB_CASES)
def test_to_blocked_kernel_matches_reference(golden_blocked, case):
    from ao_amd import ops

    d = golden_blocked
    got = ops.mx_to_blocked(torch.from_numpy(d[f"tb_{case}_in"].copy()).cuda())
    assert np.array_equal(got.cpu().numpy(), d[f"tb_{case}_out"])
    e8 = ops.mx_to_blocked(torch.from_numpy(d[f"tb_{case}_in"].copy()).cuda().view(torch.float8_e8m0fnu))
    assert e8.dtype == torch.float8_e8m0fnu and torch.equal(e8.view(torch.uint8), got)


@pytest.mark.gpu
@pytest.mark.parametrize("case", MG_CASES)
def test_blocked_m_groups_kernel_matches_reference(golden_blocked, case):
    from ao_amd import ops

    d = golden_blocked
    s, offs = torch.from_numpy(d[f"mg_{case}_in"].copy()).cuda(), torch.from_numpy(d[f"mg_{case}_offs"].copy()).cuda()
    torch.full((d[f"mg_{case}_out"].size + 4096,), 0xAB, dtype=torch.uint8, device="cuda")  # dirty the allocator's block
    got = ops.mx_block_rearrange_2d_M_groups(s, offs)
    assert got.shape == d[f"mg_{case}_out"].shape and np.array_equal(got.cpu().numpy(), d[f"mg_{case}_out"])
    # the mirror of the reference's Python entry points: same bytes, and the start rows the reference's torch restatement returns
    from ao_amd.prototype import mx

    assert torch.equal(mx.mx_block_rearrange_2d_M_groups_cuda(s, offs), got)
    sizes, starts = mx.compute_blocked_scale_offsets_for_M_groups(offs)
    assert np.array_equal(starts.cpu().numpy().astype(np.int64), d[f"mg_{case}_starts"])
    assert np.array_equal(sizes.cpu().numpy(), np.diff(d[f"mg_{case}_offs"], prepend=0))


@pytest.mark.gpu
@pytest.mark.parametrize("groups,cols,top", [(8, 128, 300), (64, 224, 70), (3, 5, 200), (200, 16, 40), (1, 448, 1000)])
def test_blocked_m_groups_kernel_vs_oracle_random(groups, cols, top):
    """Mixtral / DeepSeek-sized scale rows (K / 32 = 128, 224, 448), more groups than the reference's 32, empty groups, a column count
    that is not a multiple of 4 (byte path); the tail of the upper-bound buffer must be zero although it starts uninitialised."""
    from ao_amd import ops
    from oracle import mx_ref

    rng = np.random.default_rng(groups * 100 + cols)
    sizes = rng.integers(0, top, size=groups)
    if groups > 1:
        sizes[rng.integers(0, groups)] = 0
    ends = np.cumsum(sizes).astype(np.int32)
    s = rng.integers(0, 256, size=(int(ends[-1]), cols), dtype=np.uint8)
    want, _ = mx_ref.to_blocked_2d_M_groups(s, ends)
    torch.full((want.size + 4096,), 0xCD, dtype=torch.uint8, device="cuda")
    got = ops.mx_block_rearrange_2d_M_groups(torch.from_numpy(s).cuda(), torch.from_numpy(ends).cuda(), chunks_per_tb=8)
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.gpu
def test_blocked_m_groups_refusals_and_edges():
    from ao_amd import ops

    s = torch.zeros((4, 8), dtype=torch.uint8, device="cuda")
    offs = torch.tensor([4], dtype=torch.int32, device="cuda")
    with pytest.raises(AssertionError, match="chunks_per_tb"):
        ops.mx_block_rearrange_2d_M_groups(s, offs, 3)
    with pytest.raises(AssertionError, match="int32"):
        ops.mx_block_rearrange_2d_M_groups(s, offs.long())
    with pytest.raises(AssertionError, match="uint8"):
        ops.mx_block_rearrange_2d_M_groups(s.float(), offs)
    # no rows at all: the buffer of the upper bound, all zero
    z = ops.mx_block_rearrange_2d_M_groups(torch.zeros((0, 8), dtype=torch.uint8, device="cuda"), torch.tensor([0, 0], dtype=torch.int32, device="cuda"))
    assert z.shape == (256, 8) and not z.any()
    # malformed offsets (decreasing, past the rows) do not fault
    ops.mx_block_rearrange_2d_M_groups(s, torch.tensor([3, 1, 900], dtype=torch.int32, device="cuda"))
    torch.cuda.synchronize()
