"""bench.py's launch contract without a GPU: `python bench.py --gpus N` with WORLD_SIZE unset must start its own ranks as a
CHILD `torch.distributed.run` (never replace itself: a process that has touched the GPU must not exec) and propagate the
child's return code.  Without a GPU every rank fails loudly (engine/runtime.py: no CPU fallback), so the parent must too."""
import os
import socket
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_bench_self_launch_propagates_child_failure():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    env.update({"MASTER_PORT": str(port), "WDG_DIST_BACKEND": "gloo", "WDG_DEVICE": "0"})
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--batch", "2", "--size", "32", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], env=env, cwd=str(ROOT), capture_output=True, text=True, timeout=600)
    import torch
    if torch.cuda.is_available():
        assert r.returncode == 0, r.stderr[-2000:]
        return
    assert r.returncode != 0
    assert "launch with torch.distributed.run" not in r.stderr           # the old behaviour: refuse instead of launching
    assert "torch.distributed" in r.stderr or "ChildFailedError" in r.stderr or "rank" in r.stderr.lower()
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_rejects_mismatched_world():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4"], env=env, cwd=str(ROOT), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
