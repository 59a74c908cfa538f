"""bench.py's multi-GPU launch contract on CPU (gloo, two processes, train step stubbed): `python bench.py --gpus N` launches N
ranks itself when no launcher environment is present, reports n_gpus == N, and refuses to print a line for a world size that is
not the one asked for (reference: MirroredStrategy over every visible GPU, tensorflow_asr/utils/env_util.py:57-70)."""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(TFASR_BENCH_STUB="1", **kw)
    return env


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_relaunch_cmd_shapes():
    b = _bench_module()
    assert b.relaunch_cmd(1, ["--gpus", "1"], {}) is None
    assert b.relaunch_cmd(4, ["--gpus", "4"], {"WORLD_SIZE": "4"}) is None  # already a rank of a launcher
    cmd = b.relaunch_cmd(4, ["--gpus", "4", "--steps", "3"], {})
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5] == BENCH


def test_self_launch_two_ranks_gloo():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 2 * 32 and out["config"]["parallelism"] == "dp2"
    assert out["value"] > 0
    # three timed regions by default, the line reports the median one; per-rank wire accounting of the data-parallel route
    reg = out["regions"]
    assert len(reg["ms_per_step"]) == 3 and reg["min"] <= reg["median"] <= reg["max"] and out["ms_per_step"] == reg["median"]
    assert len(reg["host_enqueue_ms_per_step"]) == 3
    dpr = out["dp"]
    assert dpr["rank0"]["syncbn_wait_calls_per_step"] == 1.0 and dpr["rank_step_ms_min"] <= dpr["rank_step_ms_max"]
    assert set(dpr["max_over_ranks"]) == {"syncbn_wait_ms", "grad_allreduce_exposed_ms"} and dpr["stats_communicator"] == "own"


def test_refuses_world_mismatch():
    # a launcher environment of ONE rank with --gpus 2: no line, non-zero exit
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_env(WORLD_SIZE="1", RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "refusing" in (r.stderr + r.stdout)
