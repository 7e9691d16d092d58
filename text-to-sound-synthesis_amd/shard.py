"""Multi-GPU sharding of the generation path: captions are independent units, so each rank (one
process per GPU, torch.distributed 'nccl' = RCCL over xGMI) owns a contiguous slice of them and
runs the whole pipeline locally; the only communication is scattering the caption conditioning
from rank 0 and gathering the waveforms back.  The reference samples in a
single process; Codebook/evaluation/generate_samples_caps.py:153,306 is the pattern followed
(rank-sharded sampler, no collective inside the loop).

Noise is drawn per *caption* (generator seeded by the global caption index), not per batch, so a
caption produces the same tokens whichever rank and batch position it lands on.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_items, world_size, rank):
    """Contiguous, balanced slices: first (n % world) ranks get one extra item."""
    q, r = divmod(n_items, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def scatter_conditions(cond_all, n_items, feat_shape, device, group=None, dtype=torch.float32):
    """rank 0 holds cond_all [n_items, *feat_shape] (caption token ids i64[.,77] or embeddings
    f32[.,77,512]); every rank receives its slice."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return cond_all.to(device=device, dtype=dtype)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(n_items, world, rank)
    mine = torch.empty((hi - lo,) + tuple(feat_shape), device=device, dtype=dtype)
    if rank == 0:
        parts = []
        for r in range(world):
            a, b = shard_bounds(n_items, world, r)
            parts.append(cond_all[a:b].to(device=device, dtype=dtype).contiguous())
        if all(p.shape == parts[0].shape for p in parts):
            dist.scatter(mine, parts, src=0, group=group)
        else:  # ragged shards: point-to-point
            mine.copy_(parts[0])
            for r in range(1, world):
                dist.send(parts[r], dst=r, group=group)
    else:
        a, b = shard_bounds(n_items, world, 0)
        even = n_items % world == 0
        if even:
            dist.scatter(mine, None, src=0, group=group)
        else:
            dist.recv(mine, src=0, group=group)
    return mine


def gather_outputs(local, n_items, group=None):
    """Gather per-rank outputs [n_local, ...] to rank 0 in caption order (None elsewhere)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    local = local.contiguous()
    if n_items % world == 0:
        bufs = [torch.empty_like(local) for _ in range(world)] if rank == 0 else None
        dist.gather(local, bufs, dst=0, group=group)
        return torch.cat(bufs, 0) if rank == 0 else None
    if rank == 0:
        outs = [local]
        for r in range(1, world):
            a, b = shard_bounds(n_items, world, r)
            buf = torch.empty((b - a,) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
            dist.recv(buf, src=r, group=group)
            outs.append(buf)
        return torch.cat(outs, 0)
    dist.send(local, dst=0, group=group)
    return None


def per_caption_noise(global_ids, step, shape_tail, device, base_seed=1234):
    """Uniform noise [n, *shape_tail] where row i depends only on (base_seed, global_ids[i], step)."""
    out = torch.empty((len(global_ids),) + tuple(shape_tail), device=device, dtype=torch.float32)
    g = torch.Generator(device=device)
    for i, gid in enumerate(global_ids):
        g.manual_seed((base_seed * 1000003 + int(gid)) * 1009 + int(step))
        out[i] = torch.rand(shape_tail, device=device, generator=g)
    return out


def allreduce_gradients(grads, bucket_bytes=256 << 20, group=None, average=True):
    """Data-parallel gradient reduction for the training step (engine/solver_spec.py:109 wraps the model in DDP):
    the gradient tensors (dict name -> tensor, same keys and shapes on every rank) are packed into flat buckets in
    name order and all-reduced bucket by bucket -- with backend 'nccl' that is RCCL over xGMI; a few large messages
    instead of one per parameter (per-link-bound ring: message count matters) -- then averaged
    over the ranks like DDP does.  In place; returns grads."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return grads
    world = dist.get_world_size(group)
    names = sorted(grads)
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([grads[n].reshape(-1) for n in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat.div_(world)
        off = 0
        for n in bucket:
            k = grads[n].numel()
            grads[n].copy_(flat[off:off + k].view_as(grads[n]))
            off += k
        bucket, size = [], 0

    for n in names:
        b = grads[n].numel() * grads[n].element_size()
        if bucket and size + b > bucket_bytes:
            flush()
        bucket.append(n)
        size += b
    flush()
    return grads

