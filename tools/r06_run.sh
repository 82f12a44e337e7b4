# usage: bash tools/r06_run.sh <tag> [pytest args...]   -- GPU tests, then the training leg of bench.py, then the ordered launch list
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $out
R=$GRAFT_REPO_ROOT
cd $R
if [ -n "$*" ]; then timeout 1500 python -m pytest "$@" -x -q 2>&1 | tail -25 > $out/pytest_$tag.txt; tail -15 $out/pytest_$tag.txt; fi
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --no-cpu-baseline --no-vocoder --no-app > $out/bench_train_$tag.json 2> $out/bench_train_$tag.err
python - <<PY
import json
try:
    d=json.loads(open("$out/bench_train_$tag.json").read().strip().splitlines()[-1])
    print("ms_per_step", d["ms_per_step"], "host", d["config"].get("host_enqueue_ms_per_step"), "frac", d["roofline"]["frac"], "loss", d["config"].get("final_loss"))
except Exception as e:
    print("bench failed", e); print(open("$out/bench_train_$tag.err").read()[-3000:])
PY
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_train -o t -- python $R/bench.py --steps 6 --warmup 3 --preheat 0 --no-cpu-baseline --no-vocoder --no-app > $out/train_prof_$tag.log 2>&1
python $R/tools/prof_main_order.py /tmp/p_train > $out/main_order_$tag.txt 2>&1
python $R/tools/prof_main_order.py /tmp/p_train spin > $out/main_order_spin_$tag.txt 2>&1
python $R/tools/prof_streams.py /tmp/p_train > $out/streams_$tag.txt 2>&1
for k in 1 2 3; do python $R/tools/prof_main_order.py /tmp/p_train spin $k > $out/order_s${k}_$tag.txt 2>&1; done
python $R/tools/prof_step_tail.py /tmp/p_train > $out/step_tail_$tag.txt 2>&1
head -3 $out/main_order_$tag.txt; head -8 $out/streams_$tag.txt | cut -c1-200
