"""`python bench.py --gpus N` must start its own ranks when no launcher set WORLD_SIZE (the reference's trainer spawns
its workers from a plain `python train.py`: /root/reference/promptttspp/trainers/tts.py:40-48), and must keep working
under `python -m torch.distributed.run`.  CPU: the launcher + rendezvous + one gloo collective (PTPP_BENCH_LAUNCH_PROBE).
GPU: the real two-rank benchmark on one device over gloo (PTPP_BENCH_SELFTEST)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(kw)
    return env


def _one_json_line(out):
    lines = [l for l in out.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [2, 3])
def test_plain_python_spawns_its_ranks(n):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], env=_env(PTPP_BENCH_LAUNCH_PROBE="1"),
                       capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = _one_json_line(r.stdout)
    assert d == {"probe": True, "n_gpus": n, "sum": float(n * (n + 1) // 2)}


def test_under_torch_distributed_run():
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", "2"],
                       env=_env(PTPP_BENCH_LAUNCH_PROBE="1"), capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert _one_json_line(r.stdout)["n_gpus"] == 2


def test_a_failing_rank_fails_the_launch():
    # rank 1 exits before the rendezvous: rank 0 would wait for it forever; the launcher stops it and reports failure
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=_env(PTPP_BENCH_LAUNCH_PROBE="fail1"),
                       capture_output=True, timeout=120)
    assert r.returncode != 0
    assert r.stdout.strip() == b""


@pytest.mark.gpu
def test_two_ranks_on_one_device_selftest():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-vocoder",
                        "--no-cpu-baseline", "--no-app"], env=_env(PTPP_BENCH_SELFTEST="1", PTPP_RESERVE_GIB="4"),
                       capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    d = _one_json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["config"]["parallelism"] == "dp2" and d["value"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("backend,port", [("torch", "29653"), ("native", "29655")])
def test_one_rank_over_rccl_fills_the_dp_object(backend, port):
    """What the first 8-GPU driver run will exercise, as far as one GPU can: the data-parallel
    machinery over RCCL itself -- process group "nccl", parameter broadcast, per-forward BatchNorm-buffer broadcast (the default,
    trainers/tts.py:117 `broadcast_buffers`), gradient hooks, bucket all-reduces on their stream, finish() -- with ONE rank
    (PTPP_DP_FORCE_COLLECTIVES), and the `dp` diagnostics of the bench line populated so that a scaling run is readable."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--preheat", "0",
                        "--no-vocoder", "--no-cpu-baseline", "--no-app"],
                       env=_env(PTPP_DP_FORCE_COLLECTIVES="1", PTPP_RESERVE_GIB="4", MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                                PTPP_DP_BACKEND=backend),
                       capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    d = _one_json_line(r.stdout)
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0
    dp = d["dp"]
    assert dp is not None and dp["steps"] == 2 and dp["world"] == 1
    # (round 6: the same machinery through the C ABI's own communicator -- ptpp_comm_init / ptpp_allreduce_mean / ptpp_broadcast --
    #  the entry points a non-Python host would bind, include/ptpp.h)
    assert dp["backend"] == ("native-rccl" if backend == "native" else "nccl") and dp["buckets"] >= 1 and len(dp["bucket_mb"]) == dp["buckets"]
    assert dp["join_gradient_streams_ms"] is not None and dp["exposed_allreduce_ms"] is not None and dp["exposed_allreduce_ms_max"] is not None
    assert sum(dp["bucket_mb"]) > 250  # ~298 MB of f32 gradients
