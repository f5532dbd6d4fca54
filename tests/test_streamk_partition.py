"""Host-side model of mx_stream_kernel's share arithmetic (ao_amd/csrc/rb8_kernels.hip): the (slab, tile, k step) space is cut into
W contiguous shares at multiples of QS steps; a tile cut by share boundaries leaves one piece per share it touches, parked in slot
2 v (the piece workgroup v's share BEGINS with) or 2 v + 1 (the piece it ENDS with, when that is another one).  The kernel's writer
and reader derive the slot from the same rule and the reader enumerates a tile's pieces through owner(); this test restates those
formulas and checks, over random problems, that every step is covered once, that writer and reader agree on every slot, that no two
live pieces share a slot, and that a workgroup never has more than two pieces."""
import random


def shares(G, W_grid, QS, min_share=16):
    """Round 6: the first sr workgroups take sq + 1 units of QS steps, the others sq (one division; rounds 3 - 5: GQ * v // W per boundary)."""
    GQ = G // QS
    W = min(W_grid, max(1, G // min_share))
    sq, sr = divmod(GQ, W)
    B = lambda v: v * sq + min(v, sr)  # noqa: E731
    return W, GQ, (sq, sr), [(B(w) * QS, B(w + 1) * QS) for w in range(W)]


def owner(g, W, GQ, QS, wts):
    """The kernel's owner(): one division."""
    sq, sr = wts
    Q, big = g // QS, sr * (sq + 1)
    return Q // (sq + 1) if Q < big else sr + (Q - big) // sq


def test_stream_k_partition_invariants():
    rng = random.Random(0)
    for _ in range(400):
        QS = rng.choice([1, 4])
        ksteps = rng.choice([3, 4, 16, 32, 112]) * (QS if QS == 4 else 1)
        if QS == 4 and ksteps % 4:
            ksteps *= 4
        tiles = rng.randint(1, 1200)
        W_grid = rng.choice([512, 768, 7, 64, 512, 512])
        G = tiles * ksteps
        W, GQ, wts, sh = shares(G, W_grid, QS)
        # contiguous cover, cut at multiples of QS, no empty share among the first W
        assert sh[0][0] == 0 and sh[-1][1] == G
        for (a0, a1), (b0, b1) in zip(sh, sh[1:]):
            assert a1 == b0
        for g0, g1 in sh:
            assert g0 % QS == 0 and g1 % QS == 0 and g1 > g0
        # owner() inverts the shares
        for g in [0, G - 1] + [rng.randrange(G) for _ in range(50)]:
            v = owner(g, W, GQ, QS, wts)
            assert sh[v][0] <= g < sh[v][1]
        # pieces: writer side (per workgroup) vs reader side (per tile)
        written = {}
        for v, (g0, g1) in enumerate(sh):
            t_first, t_last = g0 // ksteps, (g1 - 1) // ksteps
            pieces = []
            for t in range(t_first, t_last + 1):
                lo, hi = max(g0, t * ksteps), min(g1, (t + 1) * ksteps)
                if hi - lo < ksteps:  # cut tile: this workgroup holds a piece of it
                    pieces.append(t)
            assert len(pieces) <= 2
            # only the share's first and last tile can be cut
            assert all(t in (t_first, t_last) for t in pieces)
            for t in pieces:
                slot = 2 * v + (1 if t * ksteps > g0 else 0)  # the kernel's park()
                assert slot not in written
                written[slot] = t
        for t in range(tiles):
            t0 = t * ksteps
            wf, wl = owner(t0, W, GQ, QS, wts), owner(t0 + ksteps - 1, W, GQ, QS, wts)
            if wf == wl:
                continue  # whole tile inside one share: stored from the loop, nothing parked
            first_odd = 1 if t0 > sh[wf][0] else 0  # the kernel's cut_of(): only the tile's FIRST piece can be the END of a share
            for vq in range(wf, wl + 1):
                slot = 2 * vq + (first_odd if vq == wf else 0)  # the kernel's gather()
                assert written.get(slot) == t, (t, vq, slot)
        # every parked piece is read by exactly one tile's gather
        assert sorted(written.values()) == sorted(t for t in range(tiles)
                                                  for _ in range(owner(t * ksteps, W, GQ, QS, wts), owner(t * ksteps + ksteps - 1, W, GQ, QS, wts) + 1)
                                                  if owner(t * ksteps, W, GQ, QS, wts) != owner(t * ksteps + ksteps - 1, W, GQ, QS, wts))
