mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_bigvgan.py tests/test_stack_drivers.py -x -q -s -k "amp_layer or snake_conv or bigvgan or refuses or conv1d_fwd" 2>&1 | grep -v "Warn\|warn\|^$\|WeightNorm" | tail -n 12
for dt in bf16 f16; do echo "== $dt"; DTYPE=$dt timeout 300 python tools/bench_vocoder_stages.py 2>&1 | grep -v "amdgpu\|Warn\|WeightNorm"; done
