mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests/test_hip_kernels.py tests/test_hip_bigvgan.py -x -q -k "amp_layer or snake_conv or bigvgan" 2>&1 | tail -n 8
timeout 300 python tools/bench_snake_conv.py 2>&1 | grep -v amdgpu | grep "skip 0"
timeout 300 python tools/bench_vocoder_stages.py 2>&1 | grep -v "amdgpu\|Warn\|WeightNorm"
