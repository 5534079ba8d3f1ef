"""Splitting a progressive render over the GPUs of one node (one process per GPU, torch.distributed).

The reference renders on one device; what it does provide are the hooks that make the frame divisible:
image_offset_x/y and free work-group counts in the push constants (shaders/ray_tracer.comp:423-424,1660;
pt_render_window.gd:179-189), and a seed that depends only on the absolute pixel and on `time`
(ray_tracer.comp:1665).  Two partitionings follow from that:

* tiles   -- every rank renders a band of 8-pixel tile rows of every sample; bands are disjoint, so the
             assembled image is bit-identical to a one-GPU render and the only exchange is a gather of
             bands when the image is wanted.
* samples -- every rank renders the whole frame for its own batch of K samples and keeps its own running
             mean; the means are averaged with one reduce per batch.  The running mix of
             ray_tracer.comp:1693-1697 weights a sample by 1 / max(1, frame), so frames 0 and 1 both
             overwrite: a batch numbered 0..K-1 would throw its first sample away on every rank.  A
             batch is therefore numbered 1..K (batch_frame): the mix then is the arithmetic mean of all K
             samples, and the average over N ranks is the mean of all N*K samples up to rounding.

The functions here work on torch tensors of any device, so the same code runs over RCCL on GPUs and
over gloo in the CPU tests.
"""
from __future__ import annotations

import dataclasses
from typing import List, Tuple

import torch
import torch.distributed as dist


def sample_index(rank: int, steps_per_rank: int, k: int) -> int:
    """Global index of the k-th sample of `rank`'s batch: batches are consecutive and disjoint."""
    return rank * steps_per_rank + k


def batch_frame(k: int) -> int:
    """The `frame` push constant of the k-th sample of a rank's batch: numbering starts at 1, where the
    running mix keeps every sample (frame 0 would be overwritten by frame 1)."""
    return k + 1


def row_bands(groups_y: int, world: int) -> List[Tuple[int, int]]:
    """(first tile row, tile row count) per rank: contiguous bands whose sizes differ by at most one."""
    if world <= 0 or groups_y < 0:
        raise ValueError("world must be positive and groups_y non-negative")
    base, extra = divmod(groups_y, world)
    bands, start = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        bands.append((start, n))
        start += n
    return bands


def band_window(window, rank: int, world: int):
    """Copy of a PTRenderWindow restricted to `rank`'s band (same image size, shifted origin)."""
    first, count = row_bands(window.work_group_height, world)[rank]
    return dataclasses.replace(window, y_offset=window.y_offset + first * 8, work_group_height=count)


def reduce_batch_means(local_mean: torch.Tensor, dst: int = 0) -> torch.Tensor:
    """Average of the per-rank accumulators, left in `local_mean` on rank dst (other ranks' tensors are
    clobbered by the reduction, as with any in-place reduce)."""
    world = dist.get_world_size()
    dist.reduce(local_mean, dst=dst, op=dist.ReduceOp.SUM)
    if dist.get_rank() == dst:
        local_mean.mul_(1.0 / world)
    return local_mean


def gather_bands(local_image: torch.Tensor, tile_rows: int, dst: int = 0) -> torch.Tensor:
    """Assemble the full image on rank dst from images in which every rank has rendered only its own
    band of tile rows.  local_image is H x W x 4; rows outside the rank's band are ignored."""
    world, rank = dist.get_world_size(), dist.get_rank()
    bands = row_bands(tile_rows, world)
    most = max(n for _, n in bands) * 8
    first, count = bands[rank]
    part = torch.zeros((most,) + tuple(local_image.shape[1:]), dtype=local_image.dtype, device=local_image.device)
    part[: count * 8] = local_image[first * 8 : (first + count) * 8]
    parts = [torch.empty_like(part) for _ in range(world)] if rank == dst else None
    dist.gather(part, parts, dst=dst)
    if rank != dst:
        return local_image
    for r, (f, n) in enumerate(bands):
        local_image[f * 8 : (f + n) * 8] = parts[r][: n * 8]
    return local_image
