"""sharding.py — screen-strip sharding of one frame over the GPUs of a node (no reference counterpart).

Every rank holds the full scene, transforms all vertices, and rasterises + shades only pixel rows [y0, y1).  There is no
exchange inside the frame; the one collective is the all-gather of the RGBA16F strips at the end.  Strips are
ceil(H/N) rows (the last may be shorter) and are padded to equal size so a single all_gather_into_tensor suffices.
"""
from __future__ import annotations

from typing import Tuple


def strip_rows(height: int, world: int, rank: int) -> Tuple[int, int, int]:
    """(y0, y1, rows_per_strip) of `rank`'s strip."""
    per = (height + world - 1) // world
    y0 = min(rank * per, height)
    return y0, min(y0 + per, height), per


def gather_image(strip, full, world: int):
    """All-gather equal-size strips (torch tensors [per, W, 4]) into full [world*per, W, 4]; rows >= H are padding."""
    import torch.distributed as dist
    if world > 1:
        dist.all_gather_into_tensor(full, strip)
    else:
        full[: strip.shape[0]].copy_(strip)
    return full
