mkdir -p gpurun_out/r05
for sk in 0 256 512; do echo "== pipelined, skip $sk"; PTPP_AMP_VARIANT=8 SKIPS=$sk timeout 300 python tools/bench_amp_phases.py 2>&1 | grep -v amdgpu; done
