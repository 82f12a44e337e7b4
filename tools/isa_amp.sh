#!/bin/bash
# device ISA + register report of csrc/amp_fused.hip (development helper)
mkdir -p /tmp/isa
cd /root/repo/promptttspp_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -Wno-unused-result -Wno-return-type -S --cuda-device-only amp_fused.hip -o /tmp/isa/amp_fused.s 2>&1 | grep -E "error" -A8 | head -20
grep -E "^\s+\.(vgpr_count|vgpr_spill_count|name):" /tmp/isa/amp_fused.s | paste - - - | awk '{print $2, $4, $6}' | grep -v red_sum
