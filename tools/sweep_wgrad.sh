echo default; python tools/bench_wgrad.py wgrad 2>&1 | grep wgrad
for ns in 2 4 8 16 32; do echo "nsplit=$ns"; for s in 0 2 4 7 12; do PTPP_TUNE_NSPLIT=$ns python tools/bench_wgrad.py wgrad $s 2>&1 | grep wgrad; done; done
