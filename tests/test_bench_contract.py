"""CPU: the committed bench line (profiles/bench_r04.json, written by scripts/gpu_profile.sh on an MI355X) carries every field the
driver's contract and the roofline / CPU-baseline sections ask for, for the headline and for each BASELINE config."""
import json
import os

from conftest import ROOT


def _line():
    with open(os.path.join(ROOT, "profiles", "bench_r04.json")) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_headline_fields():
    d = _line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "tokens/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # achieved = algorithmic bytes per launch / the event-timed average launch duration
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_kernel_us"] * 1e-6) / 1e9) / r["achieved"] < 1e-6
    assert r["traffic"] is None or r["traffic"] > 0.9 * r["algorithmic_bytes_per_launch"]
    assert set(r["per_shape"]) == {"qkv", "o", "gate", "up", "down"}  # the five shapes
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    # whole-job value = tokens / max-over-ranks time
    assert abs(d["value"] - d["n_gpus"] * 1000.0 / d["ms_per_step"]) / d["value"] < 1e-6


def test_every_baseline_config_has_its_roofline_and_cpu_baseline():
    d = _line()
    cfg = d["configs"]
    assert set(cfg) >= {"int4_bs128", "int8_dyn_bs128x2048", "fp8_tp8_shards", "mxfp8_mixtral_bs64"}
    for name, c in cfg.items():
        assert "error" not in c, (name, c)
        assert c["value"] > 0 and c["workload"]
        r = c["roofline"]
        assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] <= 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6
        assert c["cpu_baseline"]["value"] > 0 and c["cpu_baseline"]["kind"] in ("port", "reference")
        if name != "int4_bs1_merged":  # (the merged layout shares the headline's CPU measurement: same weights, other module boundaries)
            assert c["cpu_baseline"]["kind"] == "port"
    assert cfg["int8_dyn_bs128x2048"]["roofline"]["peak"] == 5000.0 and cfg["int4_bs128"]["roofline"]["peak"] == 2500.0
    # round 4: the real reference's CPU path on the GPU host is the headline's baseline, the C port rides along
    assert d["cpu_baseline"]["kind"] == "reference" and d["cpu_baseline_port"]["kind"] == "port" and d["cpu_baseline_port"]["value"] > 0
    # ... the workloads added this round: int8 decode, 16 tokens on every expert, the vLLM module layout
    assert 0 < cfg["int8_dyn_bs128x2048"]["decode_M1"]["frac"] <= 1 and cfg["int8_dyn_bs128x2048"]["decode_M1"]["bound"] == "hbm"
    u = cfg["mxfp8_mixtral_bs64"]["uniform16"]
    assert u["value"] > 0 and 0 < u["roofline"]["frac"] <= 1 and abs(u["roofline"]["frac"] - u["roofline"]["achieved"] / 8000.0) < 1e-9
    assert cfg["int4_bs1_merged"]["roofline"]["bound"] == "hbm" and cfg["int4_bs1_merged"]["value"] > 0
    assert cfg["fp8_tp8_shards"]["by_M"]["M1"]["frac"] > 0.4  # the round-4 decode kernel (round 3: 0.34)
