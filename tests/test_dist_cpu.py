"""CPU, world_size 2 over gloo: batch sharding + all-gather of logits equals the single-process run
(SURVEY.md section 8e).  The per-image function here is a stand-in for the conv stack (any function
without cross-image terms shards the same way); the collective logic is what is under test."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from quantize_amd import dist as qdist


def test_shard_bounds_cover_the_batch():
    for total, world in [(2048, 8), (256, 1), (10, 4), (7, 8), (513, 2)]:
        spans = [qdist.shard_bounds(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    assert qdist.shard_bounds(2048, 8, 3) == (768, 1024)  # rank r gets images [256r, 256r+256)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _per_image_logits(x, w):
    # no cross-image term, like the conv op (quantconv2d.cu:83)
    return torch.relu(x.flatten(1)) @ w


def _worker(rank, world, port, totals, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok, shape = True, None
        # several global batches in ONE process group: 7 -> 8 gives rank 0 the same local shape (4 rows) in both
        # calls while rank 1 goes 3 -> 4 (the round-1 hang: a per-rank cached equal/unequal decision)
        for total in totals:
            g = torch.Generator().manual_seed(total)
            x = torch.randn(total, 3, 4, 4, generator=g)
            w = torch.randn(48, 10, generator=g)
            mine = qdist.shard_batch(x)
            logits = qdist.gather_logits(_per_image_logits(mine, w))
            full = _per_image_logits(x, w)
            ok = ok and torch.equal(logits, full) and torch.equal(qdist.top1(logits), full.argmax(1))
            ok = ok and torch.equal(qdist.gather_logits(_per_image_logits(mine, w)), full)
            if total % world == 0:   # the bench's form: the caller vouches for equal shards, no size exchange
                ok = ok and torch.equal(qdist.gather_logits(_per_image_logits(mine, w), equal_shards=True), full)
            shape = tuple(logits.shape)
        q.put((rank, bool(ok), shape))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("totals", [(16,), (13,), (7, 8, 7, 9, 16)])
def test_two_rank_gather_matches_single_process(totals):
    total = totals[-1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, totals, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), res
    assert all(r[2] == (total, 10) for r in res)
