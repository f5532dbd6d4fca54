"""GPU spec table for roofline fractions: the gfx950 entry the reference's table lacks.

torchao keeps `gpu_name_to_specs` in torchao/testing/training/roofline_utils.py:18-100 (H100 / B200 / MI300X ...; SURVEY.md section 6:
no MI350 / MI355 entry) and divides achieved throughput by these peaks in its float8 / mx roofline benchmarks
(benchmarks/float8/float8_inference_roofline.py).  Same keys here so the entry can be dropped into that dict; `bench.py` takes
every peak it divides by from this module.  Values: MI355X_MICROARCH.md ("Chip-level parameters": dense peaks -- AMD's headline
figures include 2:1 sparsity and are never used; "HBM": 8.0 TB/s spec, 6.29 TB/s measured float4 copy).
"""
from typing import Optional

gpu_name_to_specs = {
    "AMD Instinct MI355X": {
        "arch": "gfx950",
        "cus": 256,
        "max_clock_hz": 2.4e9,
        # dense MFMA peaks (no sparsity): 256 CU x 4096 bf16 / 8192 fp8 FLOP per clock x 2.4 GHz
        "bf16_peak_tops": 2.5e15,
        "fp8_peak_tops": 5.0e15,
        "int8_peak_tops": 5.0e15,  # nominal (bench.py divides by it); the guide's int8 micro-benchmark ceiling is 3.94e15
        "fp4_peak_tops": 10.0e15,
        "fp32_vector_peak_tops": 157.3e12,
        # HBM3E
        "peak_mem_bw_bytes_sec": 8.0e12,
        "hbm_bytes": 288e9,
        "lds_bytes_per_cu": 160 * 1024,
        "l2_bytes_per_xcd": 4 << 20,
        "infinity_cache_bytes": 256 << 20,
        # measured on this hardware (MI355X_MICROARCH.md): bf16 32x32x16 micro-benchmark 2495 TF of 2500; float4 copy 6.29 of 8.0 TB/s
        "pct_achievable_gemm_tops": 0.95,
        "pct_achievable_mem_bw": 0.79,
        # measured in round 3 on the pool's boxes (tools/mfma_rate.hip -> profiles/mfma_rate_r03.txt: every SIMD issuing independent MFMAs back
        # to back; the chip holds ~2.07 - 2.3 GHz under that load, not 2.4): what "100 % of the matrix pipe" is in practice
        "bf16_measured_mfma_tops": 2.39e15,   # 16x16x32, 16 cycles
        "fp8_measured_mfma_tops": 4.52e15,    # f8f6f4 16x16x128 / 32x32x64 with e4m3 operands, 32 / 64 cycles
        "int8_measured_mfma_tops": 4.22e15,   # i32_16x16x64_i8, 16 cycles
        # streaming reads of 60 - 500 MB launches top out here in every kernel of this repo
        "measured_read_bw_bytes_sec": 4.5e12,
        # xGMI: 7 links x ~153 GB/s per GPU, fully connected 8-GPU node
        "xgmi_links": 7,
        "xgmi_link_bytes_sec": 153e9,
    },
}
# names torch.cuda.get_device_name() has been seen to return for the same part
_ALIASES = {"AMD Instinct MI355X": ("AMD Instinct MI355X", "AMD Instinct MI355", "gfx950")}


def get_roofline_gpu_name(gpu_name: Optional[str] = None) -> str:
    """reference roofline_utils.py:103-117: exact name, else prefix match; unknown names fall back to the MI355X entry (this backend
    runs on nothing else)."""
    if gpu_name is None:
        try:
            import torch

            gpu_name = torch.cuda.get_device_name(0)
        except Exception:  # noqa: BLE001 -- no GPU in the build container
            gpu_name = "AMD Instinct MI355X"
    for known, aliases in _ALIASES.items():
        if any(gpu_name.startswith(a) for a in aliases):
            return known
    return "AMD Instinct MI355X"


def get_specs(gpu_name: Optional[str] = None) -> dict:
    return gpu_name_to_specs[get_roofline_gpu_name(gpu_name)]


def hbm_roofline_seconds(nbytes: float, gpu_name: Optional[str] = None) -> float:
    return nbytes / get_specs(gpu_name)["peak_mem_bw_bytes_sec"]


def mfma_roofline_seconds(flops: float, dtype: str = "bf16", gpu_name: Optional[str] = None) -> float:
    """dtype: bf16 | fp8 | int8 | fp4"""
    return flops / get_specs(gpu_name)[f"{dtype}_peak_tops"]
