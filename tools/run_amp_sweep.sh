mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests/test_hip_kernels.py tests/test_hip_bigvgan.py -x -q -k "amp_layer or snake_conv or bigvgan" 2>&1 | tail -n 4
timeout 300 python tools/bench_amp_layer.py 2>&1 | grep -v amdgpu | cut -c1-30 | paste - - -
SKIPS=0,5,10,15 timeout 300 python tools/bench_amp_phases.py 2>&1 | grep -v amdgpu
timeout 300 python tools/bench_vocoder_stages.py 2>&1 | grep -v "amdgpu\|Warn\|WeightNorm"
