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
