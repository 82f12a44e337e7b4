out=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --no-cpu-baseline --no-vocoder --no-app > $out/bench_train_base.json 2> $out/bench_train_base.err
tail -c 600 $out/bench_train_base.json
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_train -o t -- python $R/bench.py --steps 6 --warmup 3 --preheat 0 --no-cpu-baseline --no-vocoder --no-app > $out/train_prof.log 2>&1
python $R/tools/prof_main_order.py /tmp/p_train > $out/main_order_base.txt 2>&1
python $R/tools/prof_streams.py /tmp/p_train > $out/streams_base.txt 2>&1
head -5 $out/streams_base.txt
