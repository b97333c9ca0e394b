"""Multi-GPU film merge: the one exchange step of the path.

The reference merges ImageBlock work results by per-pixel summation under a mutex
(BlockedRenderProcess::processResult -> Film::putImageBlock, src/librender/renderproc.cpp:123-130,
src/films/mfilm.cpp:118-143).  Here every rank renders the tiles (tx, ty) with morton(tx, ty) % world == rank
(mtsgpu_set_tiles) into a full-frame [H][W][5] film and the films are summed once with a reduce over RCCL/xGMI.
With the box filter every pixel is non-zero on exactly one rank, so the sum is exact in any order."""
import numpy as np


def tile_morton(tx, ty):
    """bits of tx and ty interleaved, tx lowest: the tile-ownership key of mtsgpu_set_tiles (include/mtsgpu.h)"""
    m = 0
    for b in range(16):
        m |= ((tx >> b) & 1) << (2 * b) | ((ty >> b) & 1) << (2 * b + 1)
    return m


def tiles_of_rank(width, height, block_size, rank, world):
    """pixel keys (y*W+x) of the tiles owned by `rank`, in the order mtsgpu_render walks them"""
    tx = (width + block_size - 1) // block_size
    ty = (height + block_size - 1) // block_size
    keys = []
    for t in range(tx * ty):
        if tile_morton(t % tx, t // tx) % world != rank:
            continue
        x0, y0 = (t % tx) * block_size, (t // tx) * block_size
        ys = np.arange(y0, min(y0 + block_size, height))
        xs = np.arange(x0, min(x0 + block_size, width))
        keys.append((ys[:, None] * width + xs[None, :]).reshape(-1))
    return np.concatenate(keys).astype(np.uint32) if keys else np.zeros(0, dtype=np.uint32)


def reduce_film(film, dst=0, group=None):
    """Sum the per-rank films into rank `dst` (torch.distributed; backend nccl == RCCL on ROCm, gloo on CPU)."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return film
    dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return film
