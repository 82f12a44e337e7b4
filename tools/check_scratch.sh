#!/bin/bash
# Lists every kernel of promptttspp_amd/csrc whose code object uses scratch (private segment) memory -- spilled registers
# or, worse, a non-inlined device function with the argument block / accumulators passed through memory (a generic
# lambda that was not inlined cost the 128 x 128 conv kernel 1.9x).  Product kernels must print nothing;
# the experimental persistent amp_layer_mfma_kernel<..., true> variants are known to spill.
cd "$(dirname "$0")/../promptttspp_amd/csrc" || exit 1
for f in *.hip; do
  extra=""; [ "$f" = amp_layer.hip ] && extra="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -w $extra -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
    grep -E "Function Name|ScratchSize" | sed 's/.*remark: *//; s/\[-Rpass-analysis=kernel-resource-usage\]//' | paste - - |
    awk -F'\t' -v F="$f" '{split($2, a, " "); if (a[3] + 0 > 0) print F ": " $2 " | " $1}' | c++filt
done
