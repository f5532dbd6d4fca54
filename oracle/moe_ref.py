"""CPU oracle (test infrastructure only): MoE token-group padding.

numpy restatement of torch_pad_token_groups / torch_unpad_token_groups
(torchao/prototype/moe_training/kernels/mxfp8/quant.py:368-430, 433-480), which the reference itself uses as the
checker of its CUDA kernels torchao::fused_pad_token_groups / fused_unpad_token_groups.  Pinned against outputs of
those reference functions in tests/golden/moe_pad.npz (tests/golden/make_golden.py:make_moe).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.
"""
import numpy as np


def pad_token_groups(inputs, group_offsets, alignment_size):
    """inputs [T, D]; group_offsets int32 [E] cumulative ends -> (padded [R, D], starts int32 [E], ends int32 [E])."""
    inputs = np.ascontiguousarray(inputs)
    offs = np.asarray(group_offsets, dtype=np.int64)
    tokens, groups = inputs.shape[0], offs.shape[0]
    sizes = np.diff(offs, prepend=0)                                  # quant.py:397-400
    padded_sizes = (sizes + alignment_size - 1) // alignment_size * alignment_size  # :403-405
    ends = np.cumsum(padded_sizes)                                    # :408
    rows = tokens + groups * alignment_size                           # :412
    rows = (rows + alignment_size - 1) // alignment_size * alignment_size  # :413-415
    out = np.zeros((rows, inputs.shape[1]), dtype=inputs.dtype)       # :416-418
    starts = ends - padded_sizes                                      # :424
    first = 0
    for g in range(groups):                                           # :425-429
        out[starts[g] : starts[g] + sizes[g]] = inputs[first : first + sizes[g]]
        first += sizes[g]
    return out, starts.astype(np.int32), ends.astype(np.int32)


def unpad_token_groups(padded, group_offsets, padded_group_start_offsets, num_tokens):
    """Inverse gather (quant.py:455-478); raises like the reference when the sizes do not add up."""
    offs = np.asarray(group_offsets, dtype=np.int64)
    sizes = np.diff(offs, prepend=0)
    chunks = [padded[s : s + n] for s, n in zip(np.asarray(padded_group_start_offsets, dtype=np.int64), sizes)]
    out = np.concatenate(chunks, axis=0)
    if out.shape[0] != num_tokens:
        raise RuntimeError(f"Unpad output size mismatch: expected {num_tokens} tokens but got {out.shape[0]} tokens. ")
    return out


# ---- expert-parallel regrouping (torchao/prototype/moe_training/ep/) ------------------------------------------------------------
def generate_permute_indices(tokens_per_expert_group, experts_per_rank, num_ranks, max_len, alignment):
    """numpy restatement of generate_permute_indices + fill_indices_cpu (ep/kernels.py:94-129, 132-214): tokens arrive rank-major
    (for rank r: expert 0's tokens, expert 1's, ...); the result lists, expert-major and with every expert's group padded to
    `alignment` rows (an empty expert gets one aligned block), the source row of each position, -1 for padding.
    Pinned by tests/golden/moe_permute.npz (make_golden.py:make_moe_permute)."""
    c = np.asarray(tokens_per_expert_group, dtype=np.int64)
    start = np.cumsum(c) - c                                              # :166-168
    total = np.maximum(c.reshape(num_ranks, -1).sum(0), alignment)        # :171-174
    m_sizes = ((total + alignment - 1) // alignment * alignment).astype(np.int32)  # :177-179
    m_offsets = np.cumsum(m_sizes).astype(np.int32)                       # :183
    write = m_offsets - m_sizes
    idx = np.full(max_len, -1, dtype=np.int32)
    for e in range(experts_per_rank):                                     # fill_indices_cpu :110-128
        w = int(write[e])
        for r in range(num_ranks):
            i = r * experts_per_rank + e
            n = int(c[i])
            if n > 0:
                end = min(w + n, max_len)
                idx[w:end] = np.arange(start[i], start[i] + (end - w), dtype=np.int32)
            w += n
    return idx, m_sizes, m_offsets


def gather_rows(x, idx):
    """`vstack(x, 0)[idx]` (ep/permute.py:86-96, 195-198): index -1 (or len(x)) is the appended zero row."""
    xp = np.concatenate([x, np.zeros((1,) + x.shape[1:], x.dtype)], axis=0)
    return xp[np.asarray(idx, dtype=np.int64)]


def scatter_rows(y, idx, num_rows):
    """`out = empty(num_rows + 1); out[idx] = y; out[:-1]` (ep/unpermute.py:36-41, 152-158); rows nothing names stay zero here."""
    out = np.zeros((num_rows + 1,) + y.shape[1:], y.dtype)
    out[np.asarray(idx, dtype=np.int64)] = y
    return out[:-1]


def a2a_v(rows_per_rank, splits_per_rank, rank):
    """What rank `rank` holds after the all-to-all-v of the reference's _mxfp8_all_to_all_v_kernel / _exchange_row_offsets
    (torchao/prototype/moe_training/kernels/mxfp8/comms.py:318-460): rows_per_rank[q] = rank q's rows ordered by destination,
    splits_per_rank[q][r] = how many of them go to rank r.  Returns (rows received, ordered by source rank; output_splits)."""
    import numpy as np

    got, out_splits = [], []
    for q, (rows, splits) in enumerate(zip(rows_per_rank, splits_per_rank)):
        in_off = int(sum(splits[:rank]))
        n = int(splits[rank])
        got.append(rows[in_off:in_off + n])
        out_splits.append(n)
    return np.concatenate(got, axis=0), np.asarray(out_splits, dtype=np.int64)
