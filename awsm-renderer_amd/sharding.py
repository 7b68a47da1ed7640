"""sharding.py — screen-strip sharding of one frame over the GPUs of a node (no reference counterpart).

Every rank holds the full scene, transforms all vertices, and rasterises + shades only pixel rows [y0, y1).  There is no
exchange inside the frame; the one collective is the all-gather of the RGBA16F strips at the end.  Strips are
ceil(H/N) rows (the last may be shorter) and are padded to equal size so a single all_gather_into_tensor suffices.

Two ways to cut the frame (both leave a shard's rows bit-identical to the unsharded frame's):
  strips  contiguous row ranges (awsm_hip_set_shard_rows) — simplest, but the expensive part of a frame (distant, densely
          tessellated geometry with minified textures) usually sits in one horizontal band, i.e. on one or two ranks;
  bands   32-row tile rows dealt round-robin, rank r owns tile rows r, r+N, r+2N, ... (awsm_hip_set_shard_bands) — every
          rank gets 1/N of every region.  The compact output of rank r is [L, 32, W] with L = bands_per_rank; the
          all-gather yields [N, L, 32, W] and image row y lives at [ (y//32) % N, (y//32) // N, y % 32 ].
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


TILE = 32


def bands_per_rank(height: int, world: int) -> int:
    """Tile rows (32 px) per rank in band mode, padded so that every rank gathers the same amount."""
    rows = (height + TILE - 1) // TILE
    return (rows + world - 1) // world


def band_rows(height: int, world: int, rank: int):
    """Absolute pixel rows owned by `rank` in band mode, in the order of its compact output."""
    rows = (height + TILE - 1) // TILE
    out = []
    for ty in range(rank, rows, world):
        out.extend(range(ty * TILE, min((ty + 1) * TILE, height)))
    return out


def bands_to_image(gathered, height: int, world: int):
    """[world, L, 32, W, C] (all-gathered compact band outputs; torch tensor or numpy array) -> [height, W, C] image.
    A permutation of the leading axes: rank-major -> band-major.  `.reshape` copies once (the de-interleave)."""
    n, L, t = gathered.shape[0], gathered.shape[1], gathered.shape[2]
    assert n == world and t == TILE
    perm = (1, 0, 2) + tuple(range(3, gathered.ndim))
    img = gathered.permute(*perm) if hasattr(gathered, "permute") else gathered.transpose(perm)
    return img.reshape((L * world * TILE,) + tuple(gathered.shape[3:]))[:height]
