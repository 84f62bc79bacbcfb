"""bench.py --gpus N must run N ranks whatever launched it (missing #1): with no launcher around it, it
starts the ranks itself; under a launcher whose world size differs from --gpus it refuses to print a line.  Rehearsed
on the CPU with --dry-run (the ranks form a gloo group and count themselves; nothing touches a GPU)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, drop=("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, BENCH] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    return p.returncode, lines, p.stderr


def test_gpus_2_without_a_launcher_starts_two_ranks():
    rc, lines, err = _run(["--gpus", "2", "--dry-run"])
    assert rc == 0, err[-2000:]
    assert len(lines) == 1, lines                       # rank 0 alone prints
    assert lines[0]["n_gpus"] == 2 and lines[0]["config"]["rccl_ranks"] == 2


def test_gpus_1_stays_one_process():
    rc, lines, err = _run(["--gpus", "1", "--dry-run"])
    assert rc == 0, err[-2000:]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 1 and lines[0]["config"]["rccl_ranks"] == 1


def test_world_size_that_differs_from_gpus_is_refused():
    # what `WORLD_SIZE=1 python bench.py --gpus 8` used to report as n_gpus: 1
    rc, lines, err = _run(["--gpus", "8", "--dry-run"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert rc != 0 and not lines
    assert "WORLD_SIZE=1" in err


def test_under_torchrun_the_ranks_are_not_started_twice():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29731", BENCH, "--gpus", "2", "--dry-run"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2
