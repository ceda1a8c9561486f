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


class ShardedEncoder:
    """Steady-state form of ``encode_image_sharded`` for a stream of equal-sized local batches (BASELINE config 5:
    2048 crops per step over 8 GPUs): the rank's ``[B_local, E]`` fp32 embeddings are written straight into its
    slice of a pre-allocated ``[W * B_local, E]`` buffer and the all-gather (in place, RCCL over xGMI) is issued on a
    side stream, so it overlaps the next batch's encode.  Two buffers alternate; ``step`` returns the buffer and
    an event that marks its gather complete.

    ``encode_into(images, out)`` must write ``out [B_local, E]`` on the current stream
    (``VisionTransformer.encode_into``); on CPU (gloo tests) streams and events are skipped.
    """

    def __init__(self, encode_into: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], b_local: int, embed_dim: int,
                 device: torch.device, group=None, n_buffers: int = 2, force_comm: bool = False):
        self.encode_into, self.group = encode_into, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.b_local = b_local
        self.cuda = device.type == "cuda"
        self.bufs = [torch.empty(self.world * b_local, embed_dim, dtype=torch.float32, device=device)
                     for _ in range(n_buffers)]
        # force_comm: run the side-stream event chain and the in-place all-gather even with one rank (a 1-rank RCCL
        # group executes the collective as a no-op copy) - the single-GPU test of the multi-GPU leg
        self.force_comm = bool(force_comm and dist.is_initialized())
        self.comm = torch.cuda.Stream(device) if (self.cuda and (self.world > 1 or self.force_comm)) else None
        self.done = [None] * n_buffers        # event: the gather into buffer i has completed
        self.i = 0

    def step(self, images: torch.Tensor, consumed=None):
        """Encode this rank's batch and start the all-gather -> (gathered [W*B_local, E], done_event | None).
        Wait for the event (or call ``finish``) before reading rows of other ranks.

        Buffer reuse: ``n_buffers`` steps later the same buffer is overwritten, ordered only after ITS OWN gather (the
        ``done`` event), on the stream that calls ``step``.  A consumer must therefore read the returned buffer on that
        same stream (after ``wait_event(done_event)``), or finish reading before the ``n_buffers``-th next ``step``;
        a consumer on another stream passes the event that marks its read complete as ``consumed``."""
        buf, i = self.bufs[self.i], self.i
        self.i = (self.i + 1) % len(self.bufs)
        lo = self.rank * self.b_local
        mine = buf[lo: lo + self.b_local]
        if self.comm is not None and self.done[i] is not None:
            torch.cuda.current_stream().wait_event(self.done[i])    # the buffer's previous gather has completed
        if self.cuda and consumed is not None:
            torch.cuda.current_stream().wait_event(consumed)        # ... and its reader on another stream has finished
        self.encode_into(images, mine)
        if self.world == 1 and not self.force_comm:
            return buf, None
        if self.comm is None:                                       # CPU / gloo
            dist.all_gather_into_tensor(buf, mine.clone(), group=self.group)
            return buf, None
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(ready)
            dist.all_gather_into_tensor(buf, mine, group=self.group)      # in place: input = this rank's slice
            ev = torch.cuda.Event()
            ev.record()
        self.done[i] = ev
        return buf, ev

    def finish(self):
        """Make the current stream wait for every outstanding gather."""
        if self.comm is not None:
            torch.cuda.current_stream().wait_stream(self.comm)
