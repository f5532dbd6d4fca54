#!/bin/bash
# compact per-kernel resource table of one .hip source:  bash scripts/kres.sh ao_amd/csrc/int4_kernels.hip [filter]
src=$1; filt=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-strict-aliasing -fno-vectorize -Wno-unused-result \
  -Rpass-analysis=kernel-resource-usage -c "$src" -o /tmp/kres.o 2>&1 | grep -E "error|Function Name|VGPRs:|ScratchSize|Occupancy|LDS Size" \
  | sed -E 's/.*(Function Name: |VGPRs: |ScratchSize \[bytes\/lane\]: |Occupancy \[waves\/SIMD\]: |LDS Size \[bytes\/block\]: )/\1/; s/ \[-Rpass.*//' \
  | paste - - - - - | awk '{print}' | c++filt 2>/dev/null | grep -E "$filt" | sed -E 's/Function Name: //; s/\(unsigned short const\*.*\)//' | cut -c1-200
