"""bench.py --gpus N without a launcher around it starts its N ranks itself (how the driver calls it); CPU only: the ranks meet over
gloo and report what they saw (--launcher-selftest), no GPU is touched."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus_n_launches_n_ranks_itself():
    r = _run(["--gpus", "2", "--launcher-selftest"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                               # rank 0's line only
    assert lines[0]["n_gpus"] == 2 and lines[0]["ranks_seen"] == 2 and lines[0]["world_size"] == 2


def test_gpus_must_match_the_launcher():
    r = _run(["--gpus", "3", "--launcher-selftest"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr
