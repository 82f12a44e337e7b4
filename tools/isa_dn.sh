#!/bin/bash
# device ISA + register report of csrc/diffnet_layer.hip (development helper)
mkdir -p /tmp/isa
cd /root/repo/promptttspp_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-return-type -S --cuda-device-only diffnet_layer.hip -o /tmp/isa/diffnet.s 2>&1 | grep -E "error" -A8 | head -40
python3 - <<'PY'
import re
t=open('/tmp/isa/diffnet.s').read()
for m in re.finditer(r"\.name:\s+(\S+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", t, re.S):
    n=m.group(1)
    if 'diffnet_layer_kernel' in n and 'Li0E' in n and 'Lb1E' in n[60:]:
        print(n[32:80], m.group(2), m.group(3))
PY
