"""Row-block sharding of the N x N genome-pair loop across the GPUs of a node.

Rows of the count matrix (one row = one subject against every query,
src/dist_hack.h:46-68) are independent, so rank r owns a contiguous block of
subjects, holds only their indexes, and no data-path collective is needed; the
one exchange is the final gather of the row blocks (68 bytes per ordered pair)
on rank 0 — RCCL over xGMI on the GPU box, gloo in the CPU tests.
"""
import math

import numpy as np


def row_block(total, world, rank):
    """Contiguous block [start, stop) of subject rows owned by `rank`; block
    sizes differ by at most one."""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def max_rows(total, world):
    return max(row_block(total, world, r)[1] - row_block(total, world, r)[0] for r in range(world))


def weak_scaling_set_size(n_gpus, base=29):
    """Genomes in the set so that every GPU does about one base x base tile of
    the matrix: G ~ base * sqrt(n_gpus), a multiple of n_gpus."""
    table = {1: 29, 2: 42, 4: 60, 8: 80}
    if base == 29 and n_gpus in table:
        return table[n_gpus]
    return n_gpus * max(1, round(base / math.sqrt(n_gpus)))


def gather_matrix(block, total, dist=None, world=1, rank=0, force=False, rows=None):
    """All ranks call with their padded row block, a (max_rows, total, 17) int32
    tensor; returns the full (rows, total, 17) uint32 matrix on rank 0 and None
    elsewhere (rows = number of subject rows of the job, default: total).
    force=True goes through the collective even with one rank."""
    import torch
    rows = total if rows is None else rows
    if world == 1 and not (force and dist is not None):
        a, b = row_block(rows, 1, 0)
        return block[: b - a].cpu().numpy().view(np.uint32)
    parts = [torch.empty_like(block) for _ in range(world)]
    dist.all_gather(parts, block)
    if rank != 0:
        return None
    out = []
    for r in range(world):
        a, b = row_block(rows, world, r)
        out.append(parts[r][: b - a].cpu().numpy().view(np.uint32))
    return np.concatenate(out, axis=0)
