"""Host-side model of mx_stream_kernel's share arithmetic (ao_amd/csrc/rb8_kernels.hip): the (slab, tile, k step) space is cut into
W contiguous shares at single steps (round 6: the first G % W workgroups take one step more); a tile cut by share boundaries leaves one
piece per share it touches, parked in slot 2 v (the piece workgroup v's share BEGINS with) or 2 v + 1 (the piece it ENDS with, when
that is another one).  The kernel's writer and reader derive the slot from the same rule and the reader enumerates a tile's pieces
through owner(); this test restates those formulas and checks, over random problems, that every step is covered once, that no share
is empty, that owner() inverts the shares, that a share which begins inside a 4-step scale block finds that block inside its first tile,
that writer and reader agree on every slot, that no two live pieces share a slot, and that a workgroup never has more than two pieces."""
import random


def shares(G, W_grid, min_share=16):
    """The first sr workgroups take sq + 1 steps, the others sq (one division; rounds 3 - 5: GQ * v // W per boundary, in units of 4 steps)."""
    W = min(W_grid, max(1, G // min_share))
    sq, sr = divmod(G, W)
    B = lambda v: v * sq + min(v, sr)  # noqa: E731
    return W, (sq, sr), [(B(w), B(w + 1)) for w in range(W)]


def owner(g, wts):
    """The kernel's owner(): one division."""
    sq, sr = wts
    big = sr * (sq + 1)
    return g // (sq + 1) if g < big else sr + (g - big) // sq


def cut_of(tile, ksteps, v, sh, wts):
    """The kernel's cut_of() as workgroup v calls it for one of ITS cut tiles: the end of the tile that lies inside v's share is v's."""
    g0, g1 = sh[v]
    T0 = tile * ksteps
    wf = v if T0 >= g0 else owner(T0, wts)
    wl = v if T0 + ksteps <= g1 else owner(T0 + ksteps - 1, wts)
    return wl - wf + 1, wf, (1 if T0 > sh[wf][0] else 0)


def test_stream_k_partition_invariants():
    rng = random.Random(0)
    for _ in range(400):
        ksteps = rng.choice([3, 4, 16, 32, 112])
        tiles = rng.randint(1, 1200)
        W_grid = rng.choice([256, 512, 768, 7, 64, 512])
        G = tiles * ksteps
        W, wts, sh = shares(G, W_grid)
        # contiguous cover, no empty share, equal to within one step
        assert sh[0][0] == 0 and sh[-1][1] == G
        for (a0, a1), (b0, b1) in zip(sh, sh[1:]):
            assert a1 == b0
        for g0, g1 in sh:
            assert g1 > g0 and (g1 - g0) - G // W in (0, 1)
            if ksteps % 4 == 0:  # QS == 4: the scale block a share begins in starts at gv = g0 - (k00 & 3), inside the share's first tile
                k00 = g0 % ksteps
                gv = g0 - (k00 & 3)
                assert gv // ksteps == g0 // ksteps and (gv % ksteps) % 4 == 0
        # owner() inverts the shares
        for g in [0, G - 1] + [rng.randrange(G) for _ in range(50)]:
            v = owner(g, wts)
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
                # what v works out for this tile agrees with the tile's own view
                t0 = t * ksteps
                wf, wl = owner(t0, wts), owner(t0 + ksteps - 1, wts)
                assert cut_of(t, ksteps, v, sh, wts) == (wl - wf + 1, wf, 1 if t0 > sh[wf][0] else 0)
        cut = 0
        for t in range(tiles):
            t0 = t * ksteps
            wf, wl = owner(t0, wts), owner(t0 + ksteps - 1, wts)
            if wf == wl:
                continue  # whole tile inside one share: stored from the loop, nothing parked
            first_odd = 1 if t0 > sh[wf][0] else 0  # only the tile's FIRST piece can be the END of a share
            for vq in range(wf, wl + 1):
                slot = 2 * vq + (first_odd if vq == wf else 0)  # the kernel's gather()
                assert written.get(slot) == t, (t, vq, slot)
                cut += 1
        # every parked piece is read by exactly one tile's gather
        assert cut == len(written)
