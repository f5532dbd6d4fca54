"""CPU: the dispatch tables of the 8-bit and int4 linears, through the library's host-only introspection entries (no launch, no GPU).

DESIGN.md 4.4-4.5g describes which kernel serves which (M, N, K) band and why (each boundary was measured); these tests pin the table so
that a change of a rule shows up as a diff here.  Shapes: Llama-3-70B TP = 8 shards and Llama-3-8B linears of BASELINE.json's configs.
"""
import pytest

from ao_amd import _lib

TABLE = [
    # decode: the straight-line register-ring kernel while the activation codes + slabs fit the CU's 160 KiB of LDS (round 6; rounds 4 - 5:
    # 64 KiB of codes), else the round-3 per-tile kernel
    ((1, 7168, 8192), "dec8_kernel"),
    ((4, 8192, 1024), "dec8_kernel"),
    ((16, 8192, 1024), "dec8_kernel"),
    ((16, 7168, 8192), "dec8_kernel"),     # 128.25 KiB of codes + 8 slabs: 154.8 KiB
    ((8, 4096, 14336), "dec8_kernel"),     # 16 waves: 112 KiB + 16 slabs + partials
    ((9, 4096, 14336), "rb8_kernel"),      # round 6: what the decode kernels leave at 8 .. 64 rows on weights of >= 16 MB (was stream8_kernel; level
    ((16, 4096, 14336), "rb8_kernel"),     # here, 1.3 - 2.3 x ahead on the shapes of profiles/other_shapes_forms_r06.jsonl)
    ((7, 5120, 13824), "dec8_kernel"),     # (7 rows of K = 13824 still fit the decode kernel's LDS; 8 do not)
    ((8, 5120, 13824), "rb8_kernel"),
    ((24, 37888, 3584), "rb8_kernel"),     # K < 4096: not mid8; 296 column tiles
    ((64, 37888, 3584), "rb8_kernel"),
    ((96, 37888, 3584), "gemm8_p8_kernel"),
    ((24, 3584, 18944), "rb8_kernel"),     # mid8 refuses a K it cannot split (37 groups of 512)
    ((24, 5120, 13824), "rb8_kernel"),     # 40 column tiles x 3 parts: half the chip idle in mid8
    ((24, 8192, 7168), "rb8_kernel"),      # 64 x 2
    ((24, 15360, 5120), "mid8_kernel"),
    ((24, 3584, 3584), "stream8_kernel"),  # 12.8 MB
    ((24, 8192, 3584), "rb8_kernel"),      # the 70B / TP8 down shard
    # 17 .. 32 rows on long K: the register-ring mid-M kernel; short K stays with the per-tile kernel
    ((17, 1280, 8192), "mid8_kernel"),
    ((32, 1280, 8192), "mid8_kernel"),
    ((32, 8192, 1024), "stream8_kernel"),
    # from 33 rows: the LDS-staged weight-streaming kernel up to one 128 x 128 workgroup per CU (256 tiles; round 4: was 190)
    ((33, 8192, 1024), "rb8_kernel"),
    ((64, 7168, 8192), "rb8_kernel"),
    ((64, 28672, 4096), "rb8_kernel"),
    ((128, 8192, 3584), "rb8_kernel"),
    ((128, 28672, 4096), "rb8_kernel"),
    ((512, 4096, 4096), "rb8_kernel"),
    ((512, 8192, 1024), "rb8_kernel"),
    ((768, 1280, 8192), "rb8_kernel"),
    # round 5: more than 128 and at most 256 tiles of 256 x 128 (above 128 rows): the phase-interleaved 256 x 128 GEMM
    ((768, 7168, 8192), "gemm8_p8h_kernel"),
    ((768, 8192, 1024), "gemm8_p8h_kernel"),
    ((640, 8192, 1024), "gemm8_p8h_kernel"),
    ((1024, 7168, 8192), "gemm8_p8h_kernel"),
    ((1024, 8192, 1024), "gemm8_p8h_kernel"),
    ((256, 28672, 4096), "gemm8_p8h_kernel"),
    ((2048, 4096, 14336), "gemm8_p8h_kernel"),
    ((2048, 4096, 4096), "gemm8_p8h_kernel"),
    ((1024, 4096, 4096), "rb8_kernel"),          # exactly 128 such tiles: the weight-streaming kernel
    # ... and with 2 - 4 K parts on long K (>= 8192) from 512 rows where that makes 128 .. 256 workgroups
    ((512, 7168, 8192), "gemm8_p8h_kernel"),
    ((1024, 1280, 8192), "gemm8_p8h_kernel"),
    ((2048, 1280, 8192), "gemm8_p8h_kernel"),
    ((768, 4096, 14336), "gemm8_p8h_kernel"),
    ((256, 7168, 8192), "rb8_kernel"),           # one tile row: never with parts
    ((128, 28672, 4096), "rb8_kernel"),          # 128 rows: never
    # round 6 (profiles/midm_offgrid_r06.jsonl): two tile rows take the parts as well (was: from 512 rows) ...
    ((320, 4096, 14336), "gemm8_p8h_kernel"),
    ((384, 7168, 8192), "gemm8_p8h_kernel"),
    ((320, 1280, 8192), "rb8_kernel"),           # 20 tiles x 4 parts: below 128 workgroups
    ((448, 8192, 3584), "rb8_kernel"),           # K < 8192
    # ... and an odd count of 128-row slabs that fits one round of the chip stays with the weight-streaming kernel at K >= 4096
    ((576, 6144, 4096), "rb8_kernel"),           # 3 x 48 = 144 tiles of 256 x 128, but 5 x 48 = 240 slabs
    ((640, 6144, 4096), "rb8_kernel"),
    ((768, 6144, 4096), "gemm8_p8h_kernel"),     # 6 x 48 slabs: a second round
    # two rounds of 128 x 128 tiles and more: the tiled GEMMs -- 256 x 256 phase-interleaved from 160 such tiles on (from 128 at short K / > 512 small tiles)
    ((1024, 28672, 4096), "gemm8_p8p_kernel"),  # 448 full tiles: the persistent form (round 6)
    ((2048, 8192, 4096), "gemm8_p8_kernel"),   # 256 tiles of 256 x 256
    ((512, 16512, 4096), "gemm8_p8_kernel"),   # 130 tiles of 256 x 256 and K <= 4096 (round 4); 258 tiles of 256 x 128 would need a second round
    ((1280, 7168, 8192), "gemm8_p8_kernel"),   # 140 such tiles, but 560 of 128 x 128: a second round of the chip otherwise
    ((2048, 7168, 8192), "gemm8_p8_kernel"),
    ((16384, 14336, 4096), "gemm8_p8p_kernel"),
    ((16384, 4096, 14336), "gemm8_p8p_kernel"),
    ((2048, 8192, 1024), "gemm8_p8p_kernel"),   # 256 tiles, K < 4096: one tile per workgroup, the register-only epilogue still pays
    ((2048, 8192, 3584), "gemm8_p8p_kernel"),
    # K not a multiple of 128: the register-staged tile kernel
    ((2048, 4096, 4000), "gemm8_kernel"),
    ((0, 64, 1024), "invalid"),
]


@pytest.mark.parametrize("int8", [0, 1])
def test_8bit_dispatch_table(int8):
    lib = _lib.lib()
    got = {shape: lib.ao_gemm8_kernel_name(int8, *shape).decode() for shape, _ in TABLE}
    assert got == dict(TABLE)


def test_fp8_needs_n_multiple_of_16_int8_does_not():
    lib = _lib.lib()
    assert lib.ao_gemm8_kernel_name(0, 128, 200, 1024).decode() == "invalid"
    assert lib.ao_gemm8_kernel_name(1, 128, 200, 1024).decode() == "gemm8_dma_kernel<128x128>"


def test_int4_dispatch_bands():
    lib = _lib.lib()
    name = lambda m, n, k, g=128: lib.ao_int4_mm_kernel_name(m, n, k, g).decode()  # noqa: E731
    # per-tile kernel (1-, 4-, 8- and 16-row builds) up to 16 rows; wide weights (>= 1024 n-tiles) switch to the batched kernel from 5 rows
    assert [name(m, 14336, 4096) for m in (1, 4, 5, 8)] == ["int4_mm_kernel"] * 4
    # round 6: 9 .. 16 rows take the batched kernel's 16-row slabs from 288 n-tiles (profiles/int4_forms_r06.jsonl); narrower weights stay
    assert [name(m, 14336, 4096) for m in (9, 16)] == ["int4_mm_rb_kernel"] * 2 and [name(m, 4608, 3584) for m in (8, 9)] == ["int4_mm_kernel", "int4_mm_rb_kernel"]
    assert [name(m, 4096, 4096) for m in (9, 16)] == ["int4_mm_kernel"] * 2 and name(16, 4096, 14336) == "int4_mm_kernel"
    assert [name(m, 28672, 4096) for m in (1, 4)] == ["int4_mm_kernel"] * 2
    assert [name(m, 28672, 4096) for m in (5, 16)] == ["int4_mm_rb_kernel"] * 2
    assert name(17, 4096, 4096) == "int4_mm_rb_kernel" and name(128, 6144, 4096) == "int4_mm_rb_kernel"
    # round 5: the 128 x 128 / 32 x 32 x 16 kernel from 129 rows on, and on wide weights (>= 64 column tiles of 128) from 65 rows
    assert name(128, 14336, 4096) == "int4_mm_w32_kernel" and name(129, 4096, 4096) == "int4_mm_w32_kernel" and name(2048, 6144, 4096) == "int4_mm_w32_kernel"
    # round 6: 128 x 256 tiles (64-column wave tiles) from 512 rows where one K part of them fills >= 7/8 of every round of the chip, g >= 128
    big = "int4_mm_w32_kernel<128x256>"
    assert name(2048, 4096, 14336) == big and name(2048, 4096, 4096) == big and name(512, 14336, 4096) == big and name(2048, 14336, 4096) == big
    assert name(512, 4096, 14336) == "int4_mm_w32_kernel" and name(256, 14336, 4096) == "int4_mm_w32_kernel"  # 64 / 112 such tiles: K parts would be needed
    # last pass of round 6 (profiles/int4_forms_big_r06.jsonl): within one round of the chip from 144 such tiles, from 256 rows; over several rounds from 0.8 of full
    assert name(1024, 6144, 4096) == big and name(1024, 5120, 5120) == big and name(256, 18944, 3584) == big and name(1024, 13824, 5120) == big
    assert name(512, 18944, 3584) == "int4_mm_w32_kernel" and name(512, 8192, 8192) == "int4_mm_w32_kernel" and name(2048, 6144, 4096) == "int4_mm_w32_kernel"
    assert name(2048, 4096, 4096, 32) == "int4_mm_w32_kernel" and name(2048, 4096, 4096, 256) == big  # the weight rings of groups of 32 / 64 do not fit
    assert name(64, 14336, 4096) == "int4_mm_rb_kernel"


def test_fp8_int4_tile_forms():
    """one workgroup per 16 x 16 outputs up to 16 rows; two m-tiles beyond; two n-tiles as well above 64 rows (above 32 on
    K >= 8192) for groups of 128 / 256 when N is a multiple of 32 (profiles/fp8_int4_mt_nt_ab_r05.jsonl)."""
    lib = _lib.lib()
    name = lambda m, n, k, g=128: lib.ao_fp8_int4_kernel_name(m, n, k, g).decode()  # noqa: E731
    assert [name(m, 14336, 4096) for m in (1, 16, 17, 64, 65, 512)] == ["fp8_int4_mm_kernel<1x1>"] * 2 + ["fp8_int4_mm_kernel<2x1>"] * 2 + ["fp8_int4_mm_kernel<2x2>"] * 2
    assert [name(m, 4096, 14336) for m in (32, 33, 64)] == ["fp8_int4_mm_kernel<2x1>", "fp8_int4_mm_kernel<2x2>", "fp8_int4_mm_kernel<2x2>"]
    assert name(128, 4096, 4096, 64) == "fp8_int4_mm_kernel<2x1>" and name(128, 4096, 4096, 256) == "fp8_int4_mm_kernel<2x2>"
    assert name(128, 4112, 4096) == "fp8_int4_mm_kernel<2x1>"  # N % 32 != 0
    assert name(128, 4096, 4000) == "invalid" and name(0, 4096, 4096) == "invalid" and name(8, 4096, 4096, 48) == "invalid"


def test_8bit_launch_plans_on_the_sweep_shapes():
    """ao_gemm8_plan / ao_gemm8_plan_rows: tile rows / width / K parts of the product dispatch on the shapes of profiles/midm_final_r06.jsonl
    (fp8 = int8).  The weight-streaming kernel's pick comes from a cost model fitted to the round-6 grid (rb8_plan;
    profiles/rb8_grid_r06_*.jsonl: 64-row slabs serve 65 .. 512 rows where they win): this table is the fit's output at the time of the
    committed measurements -- a change of its constants has to show up here."""
    import ctypes

    lib = _lib.lib()
    shapes = {"qkv70b": (1280, 8192), "o70b": (8192, 1024), "gate70b": (7168, 8192), "down70b": (8192, 3584),
              "qkv8b": (6144, 4096), "o8b": (4096, 4096), "gate_up8b": (28672, 4096), "down8b": (4096, 14336)}
    want = {  # M: (kernel, tile rows, tile columns, K parts)
        "qkv70b": {96: ("rb8", 64, 64, 4), 128: ("rb8", 64, 64, 4), 256: ("rb8", 64, 64, 3), 512: ("rb8", 64, 128, 3), 768: ("rb8", 128, 128, 4), 1024: ("p8h", 256, 128, 4), 2048: ("p8h", 256, 128, 3)},
        "o70b": {96: ("rb8", 64, 64, 1), 128: ("rb8", 64, 64, 1), 256: ("rb8", 64, 128, 1), 512: ("rb8", 128, 128, 1), 768: ("p8h", 256, 128, 1), 1024: ("p8h", 256, 128, 1), 2048: ("p8", 256, 256, 1)},
        "gate70b": {96: ("rb8", 64, 128, 2), 128: ("rb8", 64, 128, 2), 256: ("rb8", 64, 128, 1), 512: ("p8h", 256, 128, 2), 768: ("p8h", 256, 128, 1), 1024: ("p8h", 256, 128, 1), 2048: ("p8", 256, 256, 1)},
        "down70b": {96: ("rb8", 64, 64, 1), 128: ("rb8", 64, 64, 1), 256: ("rb8", 64, 128, 1), 512: ("rb8", 128, 128, 1), 768: ("p8h", 256, 128, 1), 1024: ("p8h", 256, 128, 1), 2048: ("p8", 256, 256, 1)},
        "qkv8b": {96: ("rb8", 64, 128, 2), 128: ("rb8", 64, 128, 2), 256: ("rb8", 64, 128, 1), 512: ("rb8", 128, 128, 1), 768: ("p8h", 256, 128, 1), 1024: ("p8h", 256, 128, 1), 2048: ("p8", 256, 256, 1)},
        "o8b": {96: ("rb8", 64, 64, 2), 128: ("rb8", 64, 64, 2), 256: ("rb8", 64, 128, 2), 512: ("rb8", 64, 128, 1), 768: ("rb8", 128, 128, 1), 1024: ("rb8", 128, 128, 1), 2048: ("p8h", 256, 128, 1)},
        "gate_up8b": {96: ("rb8", 128, 128, 1), 128: ("rb8", 128, 128, 1), 256: ("p8h", 256, 128, 1), 512: ("p8", 256, 256, 1), 1024: ("p8", 256, 256, 1), 2048: ("p8", 256, 256, 1)},
        "down8b": {96: ("rb8", 64, 128, 4), 128: ("rb8", 64, 128, 4), 256: ("rb8", 128, 128, 4), 512: ("p8h", 256, 128, 4), 768: ("p8h", 256, 128, 2), 1024: ("p8h", 256, 128, 2), 2048: ("p8h", 256, 128, 1)},
    }
    short = {"rb8_kernel": "rb8", "gemm8_p8h_kernel": "p8h", "gemm8_p8_kernel": "p8", "gemm8_p8p_kernel": "p8"}
    for name, (n, k) in shapes.items():
        for m, expect in want[name].items():
            for int8 in (0, 1):
                rows, cols, parts = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
                _lib.check(lib.ao_gemm8_plan(int8, m, n, k, ctypes.byref(cols), ctypes.byref(parts)))
                _lib.check(lib.ao_gemm8_plan_rows(int8, m, n, k, ctypes.byref(rows)))
                got = (short[lib.ao_gemm8_kernel_name(int8, m, n, k).decode()], rows.value, cols.value, parts.value)
                assert got == expect, (name, m, int8, got, expect)
    # the per-tile streaming kernels report their 16-wide n-tiles; a shape no kernel takes is an error
    cols, parts = ctypes.c_int(), ctypes.c_int()
    _lib.check(lib.ao_gemm8_plan(0, 1, 8192, 1024, ctypes.byref(cols), ctypes.byref(parts)))
    assert (cols.value, parts.value) == (16, 1)
    with pytest.raises(ValueError):
        _lib.check(lib.ao_gemm8_plan(0, 0, 64, 1024, ctypes.byref(cols), ctypes.byref(parts)))
