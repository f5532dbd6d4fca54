"""ao_amd: MI355X (gfx950) native low-bit linear backend with torchao's
quantize_ / tensor-subclass interface for the linear hot path.

    from ao_amd.quantization import quantize_, Int4WeightOnlyConfig
    quantize_(model, Int4WeightOnlyConfig(group_size=128))

The arithmetic lives in hand-written HIP kernels behind a C ABI
(include/ao_mi355.h -> ao_amd/_C_mi355.so); there is no CPU or eager fallback.
"""
__version__ = "0.1.0"
