"""The N>1 path on CPU: world_size-2 (and 3, ragged) gloo runs of the caption sharding used by bench.py /
the pipeline: scatter the caption conditioning from rank 0, compute locally, gather in caption order."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from text_to_sound_synthesis_amd import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        cond_all = torch.arange(n_items * 6, dtype=torch.float32).view(n_items, 2, 3) if rank == 0 else None
        mine = shard.scatter_conditions(cond_all, n_items, (2, 3), torch.device("cpu"))
        lo, hi = shard.shard_bounds(n_items, world, rank)
        assert mine.shape == (hi - lo, 2, 3)
        # stand-in for the per-caption pipeline: a function of the caption's conditioning and of
        # noise keyed by the GLOBAL caption index (independent of rank / batch position)
        noise = shard.per_caption_noise(range(lo, hi), step=7, shape_tail=(4,), device=torch.device("cpu"))
        local = mine.sum((1, 2))[:, None] + noise
        out = shard.gather_outputs(local, n_items)
        if rank == 0:
            q.put(out)
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_items", [(2, 8), (2, 5), (3, 7)])
def test_scatter_compute_gather(world, n_items):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cond_all = torch.arange(n_items * 6, dtype=torch.float32).view(n_items, 2, 3)
    noise = shard.per_caption_noise(range(n_items), step=7, shape_tail=(4,), device=torch.device("cpu"))
    want = cond_all.sum((1, 2))[:, None] + noise          # what a single process would produce
    assert torch.equal(out, want)


def test_shard_bounds_cover_everything():
    for n in (1, 7, 64, 512, 513):
        for w in (1, 2, 3, 8):
            spans = [shard.shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
