#!/bin/bash
# The compiler's resource table of every kernel instantiation (VGPRs, AGPRs, scratch, occupancy):
#   tools/kernel_resources.sh > profiles/rNN_kernel_resources.txt
# rocprofv3's per-dispatch VGPR_Count column reports HALF of these numbers on gfx950 (92 for a 184-register kernel),
# so occupancy must be read from this table, not from the trace.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=on -Rpass-analysis=kernel-resource-usage "$@" \
  $R/quadruped_control_amd/csrc/qc_balance.hip -o $T/lib.so 2>&1 \
  | grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|Occupancy" | sed 's/.*remark: //; s/ \[-Rpass-analysis=kernel-resource-usage\]//' \
  | paste - - - - - | sed 's/Function Name: _ZN2qc14balance_kernelI//; s/EEvPKNS_9DevParamsElNS_7BatchInEPKjNS_8BatchOutEli//' | awk '{$1=$1};1' | sort
rm -rf $T
