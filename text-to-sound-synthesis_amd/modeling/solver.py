"""Host logic around the training step (scope row 8f-3): what engine/solver_spec.py:308-331 does per iteration --

    loss.backward() -> clip_grad_norm -> optimizer.step() -> scheduler.step(loss) -> ema.update(iteration)

-- restated over gradient *dicts* (name -> tensor) instead of torch.optim objects, because the gradients here come from
`modeling.train.TrainStep` (HIP kernels, no autograd graph) and the update is the `ds_adamw` kernel.  The classes keep
the reference's names for their knobs (configs/caps.yaml:88-131) and its quirks:

  * PlateauWarmupLR   = engine/lr_scheduler.py:14-209  ReduceLROnPlateauWithWarmup
  * GradClipWindow    = engine/clip_grad_norm.py:8-29   ClipGradNorm (+ torch.nn.utils.clip_grad_norm_)
  * EMA               = engine/ema.py:8-72

Nothing here needs the GPU library on the CPU (an EMA that lives on the GPU updates through ds_ema_multi), so the CPU test suite
drives the whole iteration order against a golden run of the reference's own classes (tests/golden/solver_schedule.npz).
"""
import math

import torch


class PlateauWarmupLR:
    """Linear warm-up from the initial lr to `warmup_lr` over `warmup` steps (one equal increment per step, computed
    once from the lr at construction), then reduce-on-plateau: the lr is multiplied by `factor` (floored at `min_lr`)
    when the metric has not improved on `best` (relative/absolute `threshold`) for more than `patience` steps.  The
    metric seen during warm-up is ignored, as in the reference (lr_scheduler.py:124-147)."""

    def __init__(self, lr, mode="min", factor=0.1, patience=10, threshold=1e-4, threshold_mode="rel", cooldown=0,
                 min_lr=0.0, eps=1e-8, warmup_lr=None, warmup=0):
        if factor >= 1.0:
            raise ValueError("Factor should be < 1.0.")
        if mode not in ("min", "max"):
            raise ValueError("mode " + mode + " is unknown!")
        if threshold_mode not in ("rel", "abs"):
            raise ValueError("threshold mode " + threshold_mode + " is unknown!")
        self.lr = float(lr)
        self.mode, self.factor, self.patience = mode, factor, patience
        self.threshold, self.threshold_mode = threshold, threshold_mode
        self.cooldown, self.min_lr, self.eps = cooldown, float(min_lr), eps
        self.warmup_lr, self.warmup = warmup_lr, warmup
        self.best = math.inf if mode == "min" else -math.inf
        self.num_bad_epochs = 0
        self.cooldown_counter = 0
        self.last_epoch = 0
        self.warmup_lr_step = None
        if warmup > 0 and warmup_lr is not None:
            self.warmup_lr_step = max(0.0, (float(warmup_lr) - self.lr) / float(warmup))

    def is_better(self, a, best):
        if self.mode == "min":
            return a < best * (1.0 - self.threshold) if self.threshold_mode == "rel" else a < best - self.threshold
        return a > best * (1.0 + self.threshold) if self.threshold_mode == "rel" else a > best + self.threshold

    def step(self, metric):
        """One scheduler step with this iteration's loss; returns the lr for the next iteration.  The metric is read
        (`float(metric)`: a host synchronisation when it is a device scalar) only once the warm-up is over -- the reference
        converts it first and then ignores it for `warmup` steps (lr_scheduler.py:124-131), same schedule, one stall per
        iteration more."""
        self.last_epoch += 1
        if self.last_epoch <= self.warmup:
            if self.warmup_lr_step is None:
                raise RuntimeError("warmup > 0 needs warmup_lr")      # the reference fails here too (None in a sum)
            self.lr = max(self.lr + self.warmup_lr_step, self.min_lr)
            return self.lr
        current = float(metric)
        if self.is_better(current, self.best):
            self.best = current
            self.num_bad_epochs = 0
        else:
            self.num_bad_epochs += 1
        if self.cooldown_counter > 0:
            self.cooldown_counter -= 1
            self.num_bad_epochs = 0
        if self.num_bad_epochs > self.patience:
            new_lr = max(self.lr * self.factor, self.min_lr)
            if self.lr - new_lr > self.eps:
                self.lr = new_lr
            self.cooldown_counter = self.cooldown
            self.num_bad_epochs = 0
        return self.lr

    def state_dict(self):
        return dict(self.__dict__)

    def load_state_dict(self, state):
        self.__dict__.update(state)


class GradClipWindow:
    """Global-L2-norm clipping of a gradient dict to `max_norm`.  The reference's window test
    (clip_grad_norm.py:21-28) clips when `it >= start_iteration` OR (`end_iteration > 0` and `it < end_iteration`),
    which with the shipped start_iteration = 0 means every iteration; reproduced as written.  The scale is torch's:
    coef = max_norm / (total_norm + 1e-6), applied only when < 1."""

    def __init__(self, start_iteration=0, end_iteration=-1, max_norm=0.5):
        self.start_iteration, self.end_iteration, self.max_norm = start_iteration, end_iteration, max_norm
        self.last_epoch = -1

    def __call__(self, grads):
        """In place; returns the pre-clip total norm (0-dim tensor) or None when this iteration is outside the window."""
        self.last_epoch += 1
        clip = self.last_epoch >= self.start_iteration
        if self.end_iteration > 0 and self.last_epoch < self.end_iteration:
            clip = True
        if not clip or not grads:
            return None
        tensors = list(grads.values())
        total = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(tensors)))
        coef = torch.clamp(self.max_norm / (total + 1e-6), max=1.0)
        torch._foreach_mul_(tensors, coef)
        return total

    def state_dict(self):
        return dict(self.__dict__)

    def load_state_dict(self, state):
        self.__dict__.update(state)


class EMA:
    """Exponential moving average of `model.get_ema_model().state_dict()` (parameters *and* buffers, as the reference's
    state-dict loop does), refreshed every `update_interval` iterations: ema = ema * decay + current * (1 - decay).
    The shipped config keeps the average on the CPU (configs/caps.yaml:98-101); `device` may equally be the GPU."""

    def __init__(self, model, decay=0.99, update_interval=1, device=torch.device("cpu")):
        self.decay, self.update_interval, self.device = decay, update_interval, torch.device(device)
        self.model = model
        self.ema = {k: v.detach().clone().to(self.device) for k, v in self._target().state_dict().items()}
        self.cur = None

    def _target(self):
        m = self.model
        return m.get_ema_model() if callable(getattr(m, "get_ema_model", None)) else m

    @torch.no_grad()
    def update(self, iteration):
        if (iteration + 1) % self.update_interval != 0:
            return
        cur = self._target().state_dict()
        rest = self.ema
        if self.device.type == "cuda":
            # the average lives on the GPU: every fp32 entry in ONE pass (ds_ema_multi: the reference's expression term for
            # term; 12 launches instead of three elementwise launches per tensor, ~25 ms per update at 19 layers)
            import ctypes
            from .. import _lib
            fast = [k for k, e in self.ema.items() if e.dtype == torch.float32 and cur[k].dtype == torch.float32 and
                    cur[k].device == e.device and e.is_contiguous() and cur[k].is_contiguous() and e.numel() > 0]
            if fast:
                rec = (ctypes.c_int64 * (3 * len(fast)))()
                for i, k in enumerate(fast):
                    rec[3 * i:3 * i + 3] = [self.ema[k].data_ptr(), cur[k].data_ptr(), self.ema[k].numel()]
                _lib.check(_lib.lib().ds_ema_multi(ctypes.cast(rec, ctypes.c_void_p), len(fast), float(self.decay), float(1 - self.decay), _lib.stream()))
                done = set(fast)
                rest = {k: e for k, e in self.ema.items() if k not in done}
        for k, e in rest.items():
            e.copy_(e * self.decay + cur[k].detach().to(self.device) * (1 - self.decay))

    def state_dict(self):
        return self.ema

    def load_state_dict(self, state_dict, strict=True):
        missing = [k for k in self.ema if k not in state_dict]
        extra = [k for k in state_dict if k not in self.ema]
        if strict and (missing or extra):
            raise RuntimeError("EMA.load_state_dict: missing %s, unexpected %s" % (missing, extra))
        for k, v in state_dict.items():
            if k in self.ema:
                self.ema[k].copy_(v.to(self.device))

    @torch.no_grad()
    def modify_to_inference(self):
        """Put the averaged weights into the live model (keeping a copy of the live ones for modify_to_train)."""
        tgt = self._target()
        self.cur = {k: v.detach().clone().to(self.device) for k, v in tgt.state_dict().items()}
        tgt.load_state_dict({k: v.to(_model_device(tgt)) for k, v in self.ema.items()})
        _invalidate(tgt)

    @torch.no_grad()
    def modify_to_train(self):
        tgt = self._target()
        tgt.load_state_dict({k: v.to(_model_device(tgt)) for k, v in self.cur.items()})
        _invalidate(tgt)


def _model_device(m):
    d = getattr(m, "device", None)
    if isinstance(d, torch.device):
        return d
    return next(iter(m.state_dict().values())).device


def _invalidate(m):
    """Weight packs cached by the HIP modules (split fp16 planes, AdaLN tables) are stale after a weight swap -- and so
    are the per-matrix pre-scales 2^s and the loss scale a TrainStep derived from the old weights (it registers itself on
    its DiffusionTransformer as a `_scale_clients` weak reference)."""
    for sub in m.modules():
        if hasattr(sub, "invalidate"):
            sub.invalidate()           # also destroys the native handle that points at the old packs
        elif hasattr(sub, "_packed"):
            sub._packed = None
        for ref in getattr(sub, "_scale_clients", ()):
            client = ref()
            if client is not None:
                client.reset_scales()


def _window_always_clips(clip):
    """ClipGradNorm's window test (clip_grad_norm.py:21-28) is true for every iteration >= 0?"""
    return clip.start_iteration <= 0 or (clip.end_iteration > 0 and clip.end_iteration >= clip.start_iteration)


def _batch_tensors(solver, batch):
    """step()'s arguments -> (x0, cond_emb, t, pt, noise): a single dict is the reference's batch and goes through the
    prologue (needs Solver(model=...)); five tensors pass through.  A batch whose prologue was started by `prefetch` (same
    dict object) only waits for it."""
    if len(batch) == 1 and isinstance(batch[0], dict):
        if solver.model is None:
            raise ValueError("step(batch dict) needs the DALLE model: Solver(..., model=dalle)")
        from .train import training_draws, training_prologue
        pre = getattr(solver, "_prefetched", None)
        solver._prefetched = None                  # (a prefetch for ANOTHER batch than the one that arrives is dropped)
        if pre is not None and pre[0] is batch[0]:
            _, x0, cond_emb, side = pre
            cur = torch.cuda.current_stream(x0.device)
            cur.wait_stream(side)
            x0.record_stream(cur)
            cond_emb.record_stream(cur)
        else:
            x0, cond_emb = training_prologue(solver.model, batch[0])
        return training_draws(solver.model, x0, cond_emb, generator=solver.generator)
    return batch


def _prefetch(solver, batch, ready=None):
    """Enqueue the mel / caption prologue of a LATER batch on a side stream now (modeling.train.training_prologue: BPE, CLIP, VQ
    encoder -- frozen weights only, nothing of the training state), so that it runs beside the iteration in flight; step(batch)
    with the same dict then only waits for it.  The reference's DataLoader workers do the host half of this; the device half
    exists because the reference encodes inside the iteration (dalle_spec.py:93-133)."""
    if solver.model is None:
        raise ValueError("prefetch(batch dict) needs the DALLE model: Solver(..., model=dalle)")
    from .train import training_prologue
    dev = solver.model.transformer.device
    if getattr(solver, "_side_stream", None) is None:
        solver._side_stream = torch.cuda.Stream(dev)
    side = solver._side_stream
    # `ready`: the stream (or event) after which the batch's device tensors are valid -- a data-loading stream; default: the
    # caller's current stream, which is correct but waits for everything enqueued on it (the replay in flight included)
    if ready is None:
        if any(torch.is_tensor(v) and v.is_cuda for v in batch.values()):
            side.wait_stream(torch.cuda.current_stream(dev))
    elif isinstance(ready, torch.cuda.Event):
        side.wait_event(ready)
    else:
        side.wait_stream(ready)
    for v in batch.values():                      # the batch's tensors are read on the side stream: the caching allocator must know
        if torch.is_tensor(v) and v.is_cuda:
            v.record_stream(side)
    with torch.cuda.stream(side):
        x0, cond_emb = training_prologue(solver.model, batch)
    solver._prefetched = (batch, x0, cond_emb, side)


class Solver:
    """One training iteration in the reference's order (engine/solver_spec.py:308-331).  `train_step` supplies
    `loss_and_grads(*batch) -> (loss, {name: grad})` and `adamw_step(grads, state, step, lr, betas, eps, weight_decay)`
    -- `modeling.train.TrainStep` on the GPU; the gradients are averaged over the data-parallel ranks (bucketed RCCL
    all-reduce, shard.allreduce_gradients) before clipping, which is where DDP's reduction lands too."""

    def __init__(self, train_step, lr=3.0e-6, betas=(0.9, 0.96), eps=1e-8, weight_decay=4.5e-2, scheduler=None,
                 clip_grad_norm=None, ema=None, allreduce=None, reducer=None, model=None, generator=None):
        """allreduce: callable(grads) run after the backward (shard.allreduce_gradients); reducer: a shard.GradientReducer --
        the same reduction overlapped with the backward (buckets are all-reduced while earlier blocks are still being
        differentiated).  Give one or neither.
        model: the DALLE drop-in whose `transformer` the train_step differentiates -- with it `step(batch)` takes the
        reference's batch dict {'image': mel, 'text': captions} (modeling.train.training_inputs: BPE -> CLIP -> VQ encode ->
        sample_time -> noise); generator: torch.Generator for the timesteps and the q_sample noise."""
        assert allreduce is None or reducer is None
        self.model, self.generator = model, generator
        self.train_step, self.lr = train_step, float(lr)
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.scheduler, self.clip_grad_norm, self.ema, self.allreduce = scheduler, clip_grad_norm, ema, allreduce
        self.reducer = reducer
        self.opt_state = {}
        self.last_iter = -1

    def prefetch(self, batch, ready=None):
        _prefetch(self, batch, ready)

    def step(self, *batch):
        """step(batch_dict) -- the reference's `self.model(batch, return_loss=True)` entry -- or step(x0, cond_emb, t, pt,
        noise) with the five tensors of TrainStep.loss_and_grads."""
        batch = _batch_tensors(self, batch)
        if self.reducer is not None:
            loss, grads = self.train_step.loss_and_grads(*batch, on_grads=self.reducer.ready)
            self.reducer.finish(grads)
        else:
            loss, grads = self.train_step.loss_and_grads(*batch)
        if self.allreduce is not None:
            self.allreduce(grads)
        total = self.clip_grad_norm(grads) if self.clip_grad_norm is not None else None
        self.last_iter += 1
        self.train_step.adamw_step(grads, self.opt_state, self.last_iter + 1, self.lr, betas=self.betas, eps=self.eps,
                                   weight_decay=self.weight_decay)
        if self.scheduler is not None:
            self.lr = self.scheduler.step(loss)
        # guards of the split backend's loss scale: the device-side saturation monitor (max |scaled dY| of every GEMM
        # input; works without clipping, one host sync every `monitor_interval` iterations), and -- when a norm exists --
        # its drift since calibration
        if hasattr(self.train_step, "check_loss_scale"):
            self.train_step.check_loss_scale()
        if total is not None and self.last_iter % 16 == 0 and hasattr(self.train_step, "observe_grad_norm"):
            self.train_step.observe_grad_norm(float(total))
        if self.ema is not None:
            self.ema.update(iteration=self.last_iter)
        return {"loss": loss, "lr": self.lr, "grad_norm": total}

    def state_dict(self):
        out = {"last_iter": self.last_iter, "lr": self.lr, "optimizer": self.opt_state}
        if self.scheduler is not None:
            out["scheduler"] = self.scheduler.state_dict()
        if self.clip_grad_norm is not None:
            out["clip_grad_norm"] = self.clip_grad_norm.state_dict()
        if self.ema is not None:
            out["ema"] = self.ema.state_dict()
        return out

    def load_state_dict(self, state):
        self.last_iter, self.lr, self.opt_state = state["last_iter"], state["lr"], state["optimizer"]
        if self.scheduler is not None and "scheduler" in state:
            self.scheduler.load_state_dict(state["scheduler"])
        if self.clip_grad_norm is not None and "clip_grad_norm" in state:
            self.clip_grad_norm.load_state_dict(state["clip_grad_norm"])
        if self.ema is not None and "ema" in state:
            self.ema.load_state_dict(state["ema"])
        if hasattr(self.train_step, "reset_scales"):      # a resumed run usually follows a weight load: re-derive the scales
            self.train_step.reset_scales()


class GraphSolver:
    """`Solver.step` with  gradients -> global-norm clip -> AdamW  replayed as ONE hipGraph (`TrainStep.capture`), or -- data
    parallel, `reduce` = e.g. `shard.allreduce_gradients` -- as TWO graphs per rank with the bucketed all-reduce enqueued
    between the replays (engine/solver_spec.py:109: DDP reduces before the optimizer step).  Per iteration the host copies the batch into the graph's static tensors, writes four scalars
    (lr, the two bias corrections, -- the clip coefficient is computed inside the graph) and launches the graph; the LR
    schedule and the EMA stay host-driven, in the reference's order (engine/solver_spec.py:308-331).  The graph is
    captured on the first batch (shapes are fixed from then on)."""

    def __init__(self, train_step, lr=3.0e-6, betas=(0.9, 0.96), eps=1e-8, weight_decay=4.5e-2, scheduler=None,
                 clip_grad_norm=None, ema=None, reduce=None, model=None, generator=None):
        """model / generator: as in Solver -- `step(batch_dict)` then runs the caption / mel prologue eagerly on the current
        stream (it is enqueued while the previous replay is still executing) and replays the captured iteration on its output."""
        self.model, self.generator = model, generator
        self.train_step, self.lr = train_step, float(lr)
        self.reduce = reduce
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.scheduler, self.clip_grad_norm, self.ema = scheduler, clip_grad_norm, ema
        # the clip coefficient is computed inside the captured graph on every replay: a window that skips iterations
        # cannot be honoured there (the shipped configs/caps.yaml window -- start_iteration 0 -- clips always)
        if clip_grad_norm is not None and not _window_always_clips(clip_grad_norm):
            raise NotImplementedError("GraphSolver clips on every iteration: a ClipGradNorm window that skips iterations "
                                      "(start_iteration %r, end_iteration %r) needs the eager Solver"
                                      % (clip_grad_norm.start_iteration, clip_grad_norm.end_iteration))
        self.iteration_graph = None
        self._pending_state = None          # load_state_dict before the first batch: applied right after the capture
        self.last_iter = -1

    def prefetch(self, batch, ready=None):
        """the next batch's BPE / CLIP / VQ-encode prologue on a side stream, beside the replay in flight (solver._prefetch)"""
        _prefetch(self, batch, ready)

    def step(self, *batch):
        batch = _batch_tensors(self, batch)
        if self.iteration_graph is None:
            max_norm = self.clip_grad_norm.max_norm if self.clip_grad_norm is not None else None
            self.iteration_graph = self.train_step.capture(*batch, betas=self.betas, eps=self.eps,
                                                           weight_decay=self.weight_decay, max_norm=max_norm,
                                                           **({"reduce": self.reduce} if self.reduce is not None else {}))
            if self._pending_state is not None:
                self.iteration_graph.load_state_dict(self._pending_state)
                self._pending_state = None
        out = self.iteration_graph(*batch, lr=self.lr)
        self.last_iter += 1
        if self.clip_grad_norm is not None:
            self.clip_grad_norm.last_epoch += 1        # the window's counter advances as in the eager solver (state_dict parity)
        if self.scheduler is not None:
            self.lr = self.scheduler.step(out["loss"])
        self.iteration_graph.check_loss_scale()        # periodic re-capture / sampled saturation monitor (no sync in between)
        if self.ema is not None:
            self.ema.update(iteration=self.last_iter)
        return {"loss": out["loss"], "lr": self.lr, "grad_norm": out["grad_norm"]}

    def state_dict(self):
        """Same layout as Solver.state_dict (a run can move between the two): the optimizer moments come out of the
        graph's static tensors, the bias-correction counter rides along as `graph_iteration`."""
        g = self.iteration_graph.state_dict() if self.iteration_graph is not None else \
            (self._pending_state or {"iteration": 0, "optimizer": {}})
        out = {"last_iter": self.last_iter, "lr": self.lr, "optimizer": g["optimizer"], "graph_iteration": g["iteration"]}
        if self.scheduler is not None:
            out["scheduler"] = self.scheduler.state_dict()
        if self.clip_grad_norm is not None:
            out["clip_grad_norm"] = self.clip_grad_norm.state_dict()
        if self.ema is not None:
            out["ema"] = self.ema.state_dict()
        return out

    def load_state_dict(self, state):
        self.last_iter, self.lr = state["last_iter"], state["lr"]
        g = {"iteration": state.get("graph_iteration", state["last_iter"] + 1), "optimizer": state["optimizer"]}
        if self.iteration_graph is not None:
            self.iteration_graph.load_state_dict(g)       # in place, into the tensors the captured graph updates
        else:
            self._pending_state = g
        if self.scheduler is not None and "scheduler" in state:
            self.scheduler.load_state_dict(state["scheduler"])
        if self.clip_grad_norm is not None and "clip_grad_norm" in state:
            self.clip_grad_norm.load_state_dict(state["clip_grad_norm"])
        if self.ema is not None and "ema" in state:
            self.ema.load_state_dict(state["ema"])
        if hasattr(self.train_step, "reset_scales"):
            self.train_step.reset_scales()
            if self.iteration_graph is not None:
                self.iteration_graph._replays = 1 << 30    # frozen constants of the old weights: re-capture at the next check
