#!/bin/bash
# VGPRs / LDS / occupancy of every kernel of one source file, as the compiler reports them:
#   scripts/kernel_regs.sh pnr_amd/csrc/smc_phased.hip
# (ph_sample must stay at <= 96 VGPRs and ph_sums<false> at <= 96: four sampling waves + one sums wave share a SIMD's 512 registers)
f=${1:-pnr_amd/csrc/smc_phased.hip}
cd "$(dirname "$f")" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -Rpass-analysis=kernel-resource-usage -c "$(basename "$f")" -o /dev/null 2>&1 |
  grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|LDS Size" | sed -e 's/^.*remark: [^ ]* *//' -e 's/ \[-Rpass.*$//' | paste - - - - -
