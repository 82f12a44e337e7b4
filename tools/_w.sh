mkdir -p gpurun_out
python -m pytest tests/test_hip_kernels.py -q -x -m gpu -k "wgrad" 2>&1 | tail -3
PTPP_WGRAD_TAPS=3 python -m pytest tests/test_hip_kernels.py -q -x -m gpu -k "wgrad" 2>&1 | tail -2
PTPP_WGRAD_TAPS=1 python tools/bench_wgrad.py wgrad > gpurun_out/wgrad_old.txt 2>&1
python tools/bench_wgrad.py wgrad > gpurun_out/wgrad_new.txt 2>&1
PTPP_WGRAD_TAPS=3 python tools/bench_wgrad.py wgrad > gpurun_out/wgrad_new3.txt 2>&1
paste -d'\n' gpurun_out/wgrad_old.txt gpurun_out/wgrad_new.txt gpurun_out/wgrad_new3.txt | grep -v amdgpu | grep "ks=[35]" | cut -c1-150
