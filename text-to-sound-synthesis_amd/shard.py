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


def _single(group, always_collective):
    """True when the call should not touch the process group: none is initialised, or it has one rank and the caller did
    not ask for the collective anyway (`always_collective`: the world-size-1 RCCL test and bench.py's communicator check
    run the very same scatter / gather / all-reduce calls an N-rank job issues)."""
    if not (dist.is_available() and dist.is_initialized()):
        return True
    return dist.get_world_size(group) == 1 and not always_collective


def scatter_conditions(cond_all, n_items, feat_shape, device, group=None, dtype=torch.float32, always_collective=False):
    """rank 0 holds cond_all [n_items, *feat_shape] (caption token ids i64[.,77] or embeddings
    f32[.,77,512]); every rank receives its slice."""
    if _single(group, always_collective):
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


def scatter_captions(captions, n_items, device, tokenize, order=None, group=None, always_collective=False):
    """Per-rank BPE: every rank holds the caption LIST (as every rank of the reference's sharded sampler opens the same
    dataset, Codebook/evaluation/generate_samples_caps.py:147-153), rank 0 decides which captions run -- `order`, n_items
    indices into the list, default 0 .. n_items-1 -- and scatters the INDICES (8 bytes per caption instead of 616 bytes of
    token ids); every rank then tokenises its own strings with `tokenize(list of str) -> i64[n, 77]` (host) and moves the
    ids to `device`.  Rank 0's host work no longer grows with the world size (1.9 ms per 64 captions, on every rank's
    critical path through the scatter before).  Returns (token ids i64[n_local, 77] on device, the local indices as a list)."""
    idx_all = None
    if not (dist.is_available() and dist.is_initialized()) or dist.get_rank(group) == 0:
        idx_all = torch.arange(n_items, dtype=torch.long) if order is None else torch.as_tensor(order, dtype=torch.long)
        assert idx_all.shape == (n_items,)
    if _single(group, always_collective):
        mine = idx_all.tolist()            # one process: the indices never leave the host (no device round trip, no sync)
    else:
        mine = scatter_conditions(idx_all, n_items, (), device, group=group, dtype=torch.long,
                                  always_collective=always_collective).tolist()
    toks = tokenize([captions[i] for i in mine])
    return toks.to(device=device, dtype=torch.long), mine


def gather_outputs(local, n_items, group=None, always_collective=False):
    """Gather per-rank outputs [n_local, ...] to rank 0 in caption order (None elsewhere)."""
    if _single(group, always_collective):
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


# ---- per-caption noise: host mirror of the sampler's in-kernel Philox stream (csrc/sampler.hip) ------------------------
# The product draws the Gumbel noise of a reverse step INSIDE ds_sample_tail (ds_denoiser_step_rng / _sample_rng): the
# uniform of class c at grid position pos of global caption id gid in sampler call `call` is a pure function of
# (seed, gid, call, pos, c) -- independent of batch composition, batch position and rank.  The functions below restate
# that function in numpy; they are what tests and the oracle-driven parity checks feed to the `u` path to reproduce a
# device draw on the host (and are checked against the Random123 known-answer vectors in tests/test_philox.py).
_PHILOX_M0, _PHILOX_M1 = 0xD2511F53, 0xCD9E8D57
_PHILOX_W0, _PHILOX_W1 = 0x9E3779B9, 0xBB67AE85


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al., SC'11): counter words c0..c3 and key words k0, k1 (numpy-broadcastable unsigned
    32-bit values) -> the four output words, uint32 arrays."""
    import numpy as np
    c0, c1, c2, c3, k0, k1 = (np.asarray(v).astype(np.uint64) & 0xFFFFFFFF for v in (c0, c1, c2, c3, k0, k1))
    c0, c1, c2, c3, k0, k1 = np.broadcast_arrays(c0, c1, c2, c3, k0, k1)
    for _ in range(10):
        p0, p1 = c0 * _PHILOX_M0, c2 * _PHILOX_M1
        hi0, lo0, hi1, lo1 = p0 >> 32, p0 & 0xFFFFFFFF, p1 >> 32, p1 & 0xFFFFFFFF
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0, k1 = (k0 + _PHILOX_W0) & 0xFFFFFFFF, (k1 + _PHILOX_W1) & 0xFFFFFFFF
    return tuple(v.astype(np.uint32) for v in (c0, c1, c2, c3))


def caption_uniforms(global_ids, call, n_codes, seq_len, seed, rng_stream=0):
    """f32[n, n_codes + 1, seq_len] (the reference's rand_like(logits) layout): the uniforms ds_sample_tail_rng
    (rng_stream 0) / ds_q_sample_rng (1) draw for captions `global_ids` in sampler call `call`.  Class c = 64 j + lane
    takes word j & 3 of Philox(counter = (64 (j >> 2) + lane, pos | rng_stream << 16, call, gid), key = seed), mapped to
    [0, 1) as (word >> 8) * 2^-24 (include/diffsound_hip.h)."""
    import numpy as np
    gid = np.asarray(list(global_ids), dtype=np.uint64).reshape(-1, 1, 1)
    c = np.arange(n_codes + 1, dtype=np.uint64).reshape(1, -1, 1)
    pos = np.arange(seq_len, dtype=np.uint64).reshape(1, 1, -1)
    j, lane = c >> 6, c & 63
    words = philox4x32_10((j >> 2) * 64 + lane, pos | (int(rng_stream) << 16), int(call), gid,
                          int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF)
    sel = np.broadcast_to(j & 3, words[0].shape)
    w = np.choose(sel.astype(np.int64), words)
    return torch.from_numpy(((w >> 8).astype(np.float32) * np.float32(2.0 ** -24)))


def per_caption_noise(global_ids, step, shape_tail, device, base_seed=1234):
    """Uniform noise [n, K + 1, L] where row i depends only on (base_seed, global_ids[i], step): the host mirror of
    the product's in-kernel draw (caption_uniforms above), kept under its round-1 name for callers that inject noise."""
    k1, length = shape_tail
    return caption_uniforms(global_ids, step, k1 - 1, length, base_seed).to(device)


def allreduce_gradients(grads, bucket_bytes=256 << 20, group=None, average=True, always_collective=False):
    """Data-parallel gradient reduction for the training step (engine/solver_spec.py:109 wraps the model in DDP):
    the gradient tensors (dict name -> tensor, same keys and shapes on every rank) are packed into flat buckets in
    name order and all-reduced bucket by bucket -- with backend 'nccl' that is RCCL over xGMI; a few large messages
    instead of one per parameter (per-link-bound ring: message count matters) -- then averaged
    over the ranks like DDP does.  In place; returns grads."""
    if _single(group, always_collective):
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



class GradientReducer:
    """The same reduction OVERLAPPED with the backward (what DDP's bucketing does for the reference, engine/solver_spec.py:109):
    the training step hands over gradients as soon as they are final -- `ready(named)`, once per transformer block, in the
    order the backward produces them -- and every bucket that fills up is flattened and all-reduced ASYNCHRONOUSLY
    (`async_op=True`; on GPUs on a communication stream that first waits for the streams that produced the tensors), while
    the backward of the next block runs.  `finish(grads)` reduces what has not been handed over (the small gradients the
    step un-scales at its very end), waits for everything, averages and writes the results back into the caller's tensors.
    Buckets of ~one block (50 MB) keep the messages large: the xGMI ring is per-link bound.
    Every rank must call ready() with the same names in the same order.  Without a process group it does nothing."""

    def __init__(self, bucket_bytes=48 << 20, group=None, average=True, always_collective=False):
        self.bucket_bytes, self.group, self.average = bucket_bytes, group, average
        self.always_collective = always_collective
        self._pending, self._pending_bytes, self._inflight, self._done = [], 0, [], set()
        self._comm = None

    def _active(self):
        return not _single(self.group, self.always_collective)

    def ready(self, named, streams=()):
        """named: {name: tensor} of gradients that will not change any more; streams: the device streams that wrote them
        (default: the current one)."""
        if not self._active():
            return
        for n, t in named.items():
            if n in self._done:
                raise RuntimeError("gradient %r handed over twice" % n)
            self._done.add(n)
            self._pending.append((n, t, tuple(streams)))
            self._pending_bytes += t.numel() * t.element_size()
        if self._pending_bytes >= self.bucket_bytes:
            self._flush()

    def _flush(self):
        if not self._pending:
            return
        bucket, self._pending, self._pending_bytes = self._pending, [], 0
        dev = bucket[0][1].device
        if dev.type == "cuda":
            if self._comm is None:
                self._comm = torch.cuda.Stream(dev)
            producers = {torch.cuda.current_stream(dev)}
            for _, _, ss in bucket:
                producers.update(ss)
            for st in producers:
                self._comm.wait_stream(st)
            with torch.cuda.stream(self._comm):
                flat = torch.cat([t.reshape(-1) for _, t, _ in bucket])
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            for _, t, _ in bucket:
                t.record_stream(self._comm)
        else:
            flat = torch.cat([t.reshape(-1) for _, t, _ in bucket])
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._inflight.append((work, flat, [(n, t) for n, t, _ in bucket]))

    def finish(self, grads):
        """Reduce the gradients of `grads` that were never handed over, wait for every bucket, average, copy back (in place
        into the tensors the caller holds).  Returns grads."""
        if not self._active():
            self._done.clear()
            return grads
        rest = {n: grads[n] for n in sorted(grads) if n not in self._done}
        unknown = self._done - set(grads)
        if unknown:
            raise RuntimeError("gradients handed over but not in the final dict: %s" % sorted(unknown)[:4])
        for n, t in rest.items():
            self._pending.append((n, t, ()))
        self._flush()
        world = dist.get_world_size(self.group)
        for work, flat, items in self._inflight:
            work.wait()                               # (NCCL: makes the current stream wait for the communication)
            dev = flat.device
            if dev.type == "cuda":
                torch.cuda.current_stream(dev).wait_stream(self._comm)
            if self.average:
                flat.div_(world)
            off = 0
            for n, t in items:
                k = t.numel()
                t.copy_(flat[off:off + k].view_as(t))
                off += k
        self._inflight, self._done = [], set()
        return grads
