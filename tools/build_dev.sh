#!/bin/bash
# dev helper: build the HIP library with resource remarks + ISA dump + loop histogram
set -e
R=/root/repo
mkdir -p /tmp/isa && cd /tmp/isa
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=on -save-temps -Rpass-analysis=kernel-resource-usage \
  $R/quadruped_control_amd/csrc/qc_balance.hip -o $R/quadruped_control_amd/libqc_balance.so 2>&1 \
  | grep -E "error|Function Name|VGPRs:|AGPRs|Scratch|Occupancy|SGPRs Spill" | sed 's/.*remark: //' || true
python $R/tools/isa_loop_hist.py /tmp/isa/qc_balance-hip-amdgcn-amd-amdhsa-gfx950.s ${1:-EqpDiagW} ${2:-12}
