"""Small backend-neutral helpers of the layer engine."""
from dataclasses import dataclass


@dataclass(frozen=True)
class ConvGeom:
    kh: int
    kw: int
    stride: int
    pad: int


def round4(c: int) -> int:
    return (c + 3) // 4 * 4


def v2(t):
    """[N,H,W,C] view -> [N*H*W, C] view (no copy; works for channel slices of dense buffers)."""
    return t.view(-1, t.shape[-1])
