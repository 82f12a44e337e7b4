"""No product kernel of libptpp_hip.so may use scratch (private segment) memory: a spilled register file or -- worse -- a
device function that was not inlined and gets the kernel's argument block and accumulators through memory.  (A generic
lambda of the activation dispatch that stopped being inlined cost the 128 x 128 conv kernel 1.9x before anything failed,
DESIGN.md section 5c.)  Read from the kernel descriptors' metadata of the built library; no GPU needed."""
import os
import re
import shutil
import subprocess

import pytest
from conftest import ROOT

LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(ROOT, "promptttspp_amd", "csrc", "libptpp_hip.so")
# experiments kept in the library behind environment switches (not the product path)
EXPERIMENTAL = ("amp_layer_mfma_kernel",)


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-readelf")) or not os.path.exists(LIB),
                    reason="needs the built library and the ROCm LLVM tools")
def test_no_product_kernel_uses_scratch_memory(tmp_path):
    so = shutil.copy(LIB, tmp_path / "libptpp_hip.so")
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", str(so)], check=True, capture_output=True, cwd=tmp_path)
    objs = [f for f in os.listdir(tmp_path) if "gfx950" in f]
    assert objs, "no gfx950 code objects found in the library"
    kernels, bad = 0, []
    for f in objs:
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", str(tmp_path / f)], check=True, capture_output=True,
                               text=True).stdout
        for name, size, spill in re.findall(r"\.name:\s+(\S+)\n\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)",
                                            notes):
            kernels += 1
            if (int(size) or int(spill)) and not any(e in name for e in EXPERIMENTAL):
                bad.append((name, int(size), int(spill)))
    assert kernels > 100, kernels  # the library holds a few hundred kernel instantiations
    assert not bad, bad
