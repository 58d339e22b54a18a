"""Process-wide HIP backend handle (one process per GPU: device = LOCAL_RANK)."""
import os

_ops = None


def get_ops():
    """The HipOps instance of this process.  Raises (never falls back) when there is no GPU or the
    HIP library is missing."""
    global _ops
    if _ops is None:
        from .hipops import HipOps
        _ops = HipOps(f"cuda:{int(os.environ.get('WDG_DEVICE', os.environ.get('LOCAL_RANK', 0)))}")
    return _ops


def set_ops(ops):
    """Inject an operator backend (tests use this to run the host logic on the CPU oracle backend)."""
    global _ops
    _ops = ops
