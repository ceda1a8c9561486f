"""Multi-rank host logic on CPU (gloo, world_size 2): shard bounds, rank-major all-gather, and that the
sharded encode equals the single-rank encode row for row."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hoigen_amd.distributed import ShardedEncoder, all_gather_rows, encode_image_sharded, shard_bounds


def test_shard_bounds_cover_exactly():
    for n in (0, 1, 7, 8, 255, 256, 2048):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def _fake_encode(x):                     # deterministic per-row "embedding"
    return torch.stack([x.flatten(1).sum(1), x.flatten(1).max(1).values, x[:, 0, 0, 0]], dim=1)


def _worker(rank, world, port, n, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    images = torch.randn(n, 3, 4, 4, generator=g)
    full = encode_image_sharded(_fake_encode, images)
    ok = torch.equal(full, _fake_encode(images))
    lo, hi = shard_bounds(n, world, rank)
    ok2 = torch.equal(all_gather_rows(_fake_encode(images[lo:hi]), n), _fake_encode(images))
    # steady-state form: equal local batches written into the rank's slice of a pre-allocated buffer
    ok3 = True
    if n % world == 0:
        bl = n // world

        def into(x, out):
            out.copy_(_fake_encode(x))
            return out

        enc = ShardedEncoder(into, bl, 3, torch.device("cpu"))
        for it in range(3):                                   # buffers alternate and are reused
            imgs = images + it
            buf, ev = enc.step(imgs[rank * bl:(rank + 1) * bl])
            ok3 = ok3 and ev is None and torch.equal(buf, _fake_encode(imgs))
        enc.finish()
    out_q.put((rank, bool(ok and ok2 and ok3)))
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [8, 7])
def test_sharded_encode_equals_single(n):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]
