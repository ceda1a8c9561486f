"""Data-parallel crop encoding: one process per GPU, contiguous shard per rank, one all-gather of the
[B_local, E] embeddings (RCCL over xGMI on MI355X; gloo on CPU for the host-logic tests).

The reference has no counterpart: it never gathers embeddings across ranks.  Crops
are independent, so the data path needs no other collective; weights are replicated.
The gathered result is bit-identical, row for row, to encoding the whole batch on one GPU (no
reduction is involved, so no re-ordering error).
"""
from __future__ import annotations

from typing import Callable, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block partition of n rows: the first n % world ranks get one extra row."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_rows(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """Concatenate per-rank row blocks (sizes from ``shard_bounds``) in rank order -> [n_total, E]."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_bounds(n_total, world, r) for r in range(world)]
    assert local.shape[0] == sizes[rank][1] - sizes[rank][0], "local shard has the wrong number of rows"
    if n_total % world == 0:
        out = torch.empty(n_total, *local.shape[1:], dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    # ragged tail: pad every shard to the largest one, gather, drop the padding
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros(mx, *local.shape[1:], dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    buf = torch.empty(world * mx, *local.shape[1:], dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    return torch.cat([buf[r * mx: r * mx + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], dim=0)


def encode_image_sharded(encode: Callable[[torch.Tensor], torch.Tensor], images: torch.Tensor,
                         group=None) -> torch.Tensor:
    """Every rank holds (or can index) the same global batch ``images [B,3,R,R]``; rank r encodes rows
    ``shard_bounds(B, W, r)`` with ``encode`` (e.g. ``model.encode_image``) and all ranks receive the full
    ``[B, E]`` embedding matrix."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return encode(images)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(images.shape[0], world, rank)
    local = encode(images[lo:hi])
    return all_gather_rows(local, images.shape[0], group)
