# usage: bash tools/r06_ab.sh <tag> "ENV1=.. ENV2=.." "ENV=.." ...   -- training leg of bench.py under each environment, two rounds, alternating
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
: > $out/ab_$tag.txt
for round in $(seq 1 ${AB_ROUNDS:-2}); do
  i=0
  for envs in "default" "$@"; do
    if [ "$envs" = "default" ]; then e=""; else e="$envs"; fi
    env $e python $R/bench.py --steps ${AB_STEPS:-20} --no-cpu-baseline --no-vocoder --no-app > /tmp/ab.json 2> /tmp/ab.err
    python - "$envs" <<'PY' >> $out/ab_$tag.txt
import json, sys
try:
    d = json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:40s} ms_per_step {d['ms_per_step']:.3f} host {d['config'].get('host_enqueue_ms_per_step')} loss {d['config'].get('final_loss')}")
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("/tmp/ab.err").read()[-1500:])
PY
    i=$((i+1))
  done
done
cat $out/ab_$tag.txt
