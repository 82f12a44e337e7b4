mkdir -p gpurun_out/r05
echo "== two blocks per CU"; SKIPS=0,5,10 python tools/bench_amp_phases.py 2>&1 | grep -v amdgpu
echo "== one block per CU"; PTPP_AMP_ONE_BLOCK=1 SKIPS=0,5,10 python tools/bench_amp_phases.py 2>&1 | grep -v amdgpu
