"""bench.py --gpus N without a launcher starts its own ranks (as the reference does with mp.spawn,
example/quantization/DDP_RootQ_train.py:30-34): the parent - which never touches a GPU - runs torch.distributed.run as a child,
forwards rank 0's single JSON line and exits with the child's status.  CPU rehearsal: --backend gloo --dry-run (no GPU work,
the line says so)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=240, env=e)


def test_gpus_2_spawns_two_ranks_and_prints_one_line():
    r = _run("--gpus", "2", "--backend", "gloo", "--dry-run", "--steps", "3", "--warmup", "1", "--model", "repvgg_a1", "--scaling", "strong",
             "--global-batch", "4096")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "strong"
    assert d["config"]["global_batch"] == 4096 and d["config"]["parallelism"].startswith("dp2")
    assert "dry-run" in d["data"] and d["value"] == 0.0          # a dry line carries no measurement and says so
    # what the collective library saw (the object a real N > 1 line carries: bench.collective_report)
    c = d["collective"]
    assert c["backend"] == "gloo" and c["world"] == 2 and c["ranks_seen"] == 2 and c["rccl_version"] is None
    assert c["observer_allreduce_calls_timed"] == 54 and c["observer_allreduce_ms"] > 0 and c["steady_state_collectives_per_step"] == 0


def test_world_size_mismatch_is_an_error_not_an_assert():
    r = _run("--gpus", "2", "--dry-run", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_failing_rank_fails_the_parent():
    # an unknown backend name is rejected by argparse in every child: the launcher must return non-zero and print no line
    r = _run("--gpus", "2", "--dry-run", "--backend", "mpi")
    assert r.returncode != 0 and not r.stdout.strip()
