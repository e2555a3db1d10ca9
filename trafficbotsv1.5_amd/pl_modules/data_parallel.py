"""Data-parallel training step: one process per GPU, scenes sharded across ranks, ONE exchange step per optimizer
step = gradient all-reduce over RCCL/xGMI (torch.distributed backend "nccl" is RCCL on ROCm).

Replaces Lightning's `strategy="ddp"` (run.py:50-52) for this path. Differences that matter on MI355X:
  * the 2,745,030 parameters that never receive a gradient (std-normal prior copies, unused norm_tgt, action-head log_std:
    SURVEY.md finding 3) are excluded statically - no per-iteration unused-parameter graph walk;
  * all live gradients (~31.6 MB fp32) travel as ONE flat all-reduce after backward. With 8 fully connected xGMI
    peers a single large message amortises launch latency best; it is ~1 % of a training step, so it is not overlapped
    with backward (nothing to hide).
"""
import os
import threading
from typing import Dict, Iterable, List, Optional

import torch
import torch.distributed as dist
from torch import Tensor


def live_parameters(module: torch.nn.Module) -> List[torch.nn.Parameter]:
    """Parameters that hold a gradient after a backward pass (call after the first backward)."""
    return [p for p in module.parameters() if p.requires_grad and p.grad is not None]


class FlatGrads:
    """ONE persistent fp32 buffer whose slices ARE the live parameters' `.grad` tensors (what DDP calls gradient_as_bucket_view):
    backward accumulates straight into it, the exchange is one all-reduce on the buffer itself - no gather copy before it, no
    per-parameter copy back after it - and a captured training step writes the same addresses at every replay.
    Zeroed by `zero()` (a fill kernel: capturable, unlike a memset node on this runtime) before each backward."""

    def __init__(self, params: Iterable[torch.nn.Parameter], align: int = 1) -> None:
        """align: every slice starts at a multiple of `align` elements (the gaps stay zero). 1 = packed; FlatAdamW lays the parameters'
        DATA out like their gradients and the kernels read weights through 16-byte loads: GraphedTrainStep asks for 64 (256 bytes, what
        the allocator gives a tensor of its own)."""
        self.params = [p for p in params]
        assert self.params, "no live parameters"
        dev = self.params[0].device
        up = lambda n: -(-n // align) * align
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += up(p.numel())
        self.flat = torch.zeros(self.offsets[-1] + self.params[-1].numel() if align == 1 else off, dtype=torch.float32, device=dev)
        self.views = []
        for p, o in zip(self.params, self.offsets):
            v = self.flat[o:o + p.numel()].view_as(p)
            if p.grad is not None:
                v.copy_(p.grad)
            self.views.append(v)
        self.attach()

    def attach(self) -> None:
        """(Re-)point every parameter's .grad at its slice (a zero_grad(set_to_none=True) elsewhere detaches them)."""
        for p, v in zip(self.params, self.views):
            p.grad = v

    def zero(self) -> None:
        self.flat.fill_(0.0)

    def detach(self) -> None:
        """.grad = None on every parameter: the next backward HANDS each parameter its gradient (autograd's AccumulateGrad keeps the
        incoming tensor) instead of adding it to a zeroed slice - one add kernel per parameter (~700 per step) and the zero fill less.
        Follow the backward with gather()."""
        for p in self.params:
            p.grad = None

    def gather(self) -> None:
        """The gradients the backward left on the parameters -> the flat buffer's slices (a few multi-tensor copy launches), and .grad
        re-pointed at the slices: the exchange, the clip and the optimizer see ONE buffer as before. Capturable (static addresses:
        the gradient tensors of a captured backward live in the graph's pool)."""
        grads = [p.grad for p in self.params]
        have = [(v, g) for v, g in zip(self.views, grads) if g is not None]
        if len(have) < len(grads):
            self.zero()  # (a parameter without a gradient this step keeps a zero slice)
        torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        self.attach()

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * self.flat.element_size()

    def allreduce(self, world_size: Optional[int] = None, group=None) -> int:
        """Average across ranks, in place: one all-reduce on the buffer (RCCL over xGMI) + one scale. Returns the bytes exchanged."""
        if not dist.is_available() or not dist.is_initialized():
            return 0
        world_size = world_size or dist.get_world_size(group)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        if world_size > 1:
            self.flat.mul_(1.0 / world_size)
        return self.nbytes


class FlatAdamW:
    """torch.optim.AdamW's update of the live parameters as ONE fused launch per run of consecutive parameters of a parameter group.

    torch's fused AdamW is a multi-tensor kernel: ~30 tensors per launch, 874 tensors = 1.30 ms of device time behind every training
    step; over one flat tensor of the same 10.7 M elements the same kernel takes 0.09 ms (tools/opt_time.py). Here the live parameters'
    data, both moment estimates and the step counters become slices of flat buffers (the gradients already are: FlatGrads) - every
    `p.data`, `optimizer.state[p]["exp_avg" | "exp_avg_sq" | "step"]` stays a tensor of its own shape that `state_dict()`, a direct
    `optimizer.step()` or a checkpoint load see as before - and `step()` runs `torch._fused_adamw_` on the flat slices with the group's
    current hyper-parameters (the lr scheduler keeps working on `optimizer.param_groups`). Same elementwise arithmetic, same results
    (tests/test_hip_data_parallel.py). Needs a fused, non-amsgrad, non-maximize AdamW on the device; anything else: `usable()` is False
    and the caller keeps `optimizer.step()`."""

    @staticmethod
    def usable(opt, flat: "FlatGrads") -> bool:
        return (type(opt) is torch.optim.AdamW and bool(opt.defaults.get("fused")) and flat.flat.is_cuda and hasattr(torch, "_fused_adamw_")
                and all(not g.get("amsgrad") and not g.get("maximize") and not g.get("capturable") and not g.get("differentiable")
                        and not torch.is_tensor(g["lr"]) for g in opt.param_groups))

    def __init__(self, opt, flat: "FlatGrads") -> None:
        assert self.usable(opt, flat)
        self.opt, self.grads = opt, flat
        params, dev = flat.params, flat.flat.device
        group_of = {id(p): gi for gi, g in enumerate(opt.param_groups) for p in g["params"]}
        assert all(id(p) in group_of for p in params), "a live parameter the optimizer does not own"
        n = flat.flat.numel()
        self.p, self.m, self.v = (torch.zeros(n, dtype=torch.float32, device=dev) for _ in range(3))
        self.steps = torch.zeros(len(params), dtype=torch.float32, device=dev)
        self.runs = []  # (group index, first element, one past the last, index of the run's first parameter); gaps of an aligned layout
        # lie inside the runs: zero parameters with zero gradients stay zero under AdamW
        for i, (q, off) in enumerate(zip(params, flat.offsets)):
            gi, k = group_of[id(q)], q.numel()
            assert q.dtype == torch.float32
            st = opt.state.get(q, {})
            sl = slice(off, off + k)
            self.p[sl].view_as(q).copy_(q.data)
            if "exp_avg" in st:
                self.m[sl].view_as(q).copy_(st["exp_avg"]), self.v[sl].view_as(q).copy_(st["exp_avg_sq"])
                self.steps[i] = float(st["step"])
            q.data = self.p[sl].view_as(q)
            opt.state[q] = {"step": self.steps[i], "exp_avg": self.m[sl].view_as(q), "exp_avg_sq": self.v[sl].view_as(q)}
            if self.runs and self.runs[-1][0] == gi:
                self.runs[-1][2] = off + k
            else:
                self.runs.append([gi, off, off + k, i])
        assert len(set(self.steps.tolist())) <= 1, "parameters at different optimizer steps"

    @torch.no_grad()
    def step(self) -> None:
        q0, q1 = self.grads.params[0], self.grads.params[-1]
        if (q0.data_ptr() != self.p.data_ptr() + 4 * self.grads.offsets[0] or q1.data_ptr() != self.p.data_ptr() + 4 * self.grads.offsets[-1]
                or q0.grad is None or q0.grad.data_ptr() != self.grads.flat.data_ptr() + 4 * self.grads.offsets[0]):
            raise RuntimeError("FlatAdamW: the parameters (or their gradients) no longer live in the flat buffers (model.to(), a swapped "
                               "`.data`, zero_grad(set_to_none=True) without FlatGrads.attach()): build a new GraphedTrainStep")
        self.steps.add_(1.0)  # every parameter's counter (views of one buffer)
        for gi, a, b, i0 in self.runs:
            g = self.opt.param_groups[gi]
            torch._fused_adamw_([self.p[a:b]], [self.grads.flat[a:b]], [self.m[a:b]], [self.v[a:b]], [], [self.steps[i0]], amsgrad=False,
                                lr=float(g["lr"]), beta1=float(g["betas"][0]), beta2=float(g["betas"][1]), weight_decay=float(g["weight_decay"]),
                                eps=float(g["eps"]), maximize=False, grad_scale=None, found_inf=None)


def clip_gradients(params, max_norm: float):
    """torch.nn.utils.clip_grad_norm_ (trainer/default.yaml:13 gradient_clip_val). With a FlatGrads the gradients ARE one buffer: its
    2-norm is the norm of the per-tensor norms, and one in-place scale replaces the foreach passes over ~700 tensors (4 launches)."""
    if not (max_norm and max_norm > 0):
        return None
    if isinstance(params, FlatGrads):
        total = torch.linalg.vector_norm(params.flat)
        params.flat.mul_((max_norm / (total + 1e-6)).clamp(max=1.0))  # (clip_grad_norm_'s coefficient)
        return total
    return torch.nn.utils.clip_grad_norm_(params, max_norm)


def allreduce_gradients(params, world_size: Optional[int] = None, group=None) -> int:
    """Average the gradients of `params` across ranks in place with one flat all-reduce. Returns the bytes exchanged.
    `params`: a FlatGrads (the gradients already live in one buffer: nothing is copied) or an iterable of parameters (their
    gradients are gathered into a temporary flat buffer and scattered back by two multi-tensor copies)."""
    if isinstance(params, FlatGrads):
        return params.allreduce(world_size, group)
    params = [p for p in params if p.grad is not None]
    if not params or not dist.is_available() or not dist.is_initialized():
        return 0
    world_size = world_size or dist.get_world_size(group)
    # (a one-rank group still takes the exchange: the flat buffer, the RCCL call and the copy back are then exercised on one GPU)
    flat = torch.empty(sum(p.grad.numel() for p in params), dtype=params[0].grad.dtype, device=params[0].grad.device)
    views, off = [], 0
    for p in params:
        views.append(flat[off:off + p.grad.numel()].view_as(p.grad))
        off += p.grad.numel()
    grads = [p.grad for p in params]
    torch._foreach_copy_(views, grads)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(world_size)
    torch._foreach_copy_(grads, views)
    return flat.numel() * flat.element_size()


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> int:
    """Rank `src`'s parameters and buffers to every rank, as ONE flat broadcast per dtype (what DDP's constructor does with
    its `_sync_params_and_buffers`): identical initial weights by construction, whatever each rank's RNG did before. Returns the
    bytes sent. No-op without an initialised process group."""
    if not dist.is_available() or not dist.is_initialized():
        return 0
    sent = 0
    with torch.no_grad():
        tensors = [t for t in list(module.parameters()) + list(module.buffers()) if t.numel() > 0]
        for dt in sorted({t.dtype for t in tensors}, key=str):
            ts = [t for t in tensors if t.dtype == dt]
            flat = torch.cat([t.reshape(-1) for t in ts])
            if dt == torch.bool:
                flat = flat.to(torch.uint8)
            dist.broadcast(flat, src=src, group=group)
            off = 0
            for t in ts:
                t.copy_(flat[off:off + t.numel()].view_as(t).to(dt))
                off += t.numel()
            sent += flat.numel() * flat.element_size()
    return sent


def rank_seed(base: int, rank: Optional[int] = None) -> int:
    """Seed of a rank's private random streams (dropout masks, latent noise, forcing draws): distinct per rank, so the ranks'
    scenes see independent noise, while the weights stay identical (`broadcast_parameters`)."""
    if rank is None:
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    return int(base) + 1000003 * int(rank)


def parameters_checksum(module: torch.nn.Module) -> Tensor:
    """[sum, sum of squares] of all parameters in float64 on their device: equal across ranks iff the replicas agree."""
    ps = [p.detach().double().reshape(-1) for p in module.parameters()]
    flat = torch.cat(ps)
    return torch.stack([flat.sum(), (flat * flat).sum()])


def train_step(wm, optimizer: torch.optim.Optimizer, batch: Dict[str, Tensor], clip_grad_norm: float = 5.0,
               live=None) -> Dict[str, Tensor]:
    """fwd + bwd + gradient all-reduce + clip (trainer/default.yaml:13 gradient_clip_val 5.0) + optimizer step.
    live: None (first step: the live parameters are found after the backward), a list of parameters, or a FlatGrads over them
    (gradients accumulate into its buffer, which is what travels)."""
    optimizer.zero_grad(set_to_none=True)  # (the parameters without a gradient path stay None; the live ones are handed theirs)
    loss = wm.training_step(batch, 0)
    from ..train_graph import backward_pack_scope

    with backward_pack_scope(wm):
        loss.backward()
    if isinstance(live, FlatGrads):
        live.gather()
    params = live if live is not None else live_parameters(wm.model)
    allreduce_gradients(params)
    clip_gradients(params, clip_grad_norm)
    optimizer.step()
    return wm.last_metrics


class GraphedTrainStep:
    """The same training step with forward + backward replayed as ONE hipGraph.

    The eager step issues ~10^5 small launches (90 closed-loop steps x the per-step schedule, forward and backward) and is
    bound by the host's launch rate; shapes are static on synthetic / fixed-size batches, the custom kernels only
    enqueue on the current stream, and the step's two host-drawn random inputs (latent noise, prior/posterior choice)
    are fed as device tensors, so the whole of fwd + bwd captures. Per call: copy the batch into the static input
    buffers, refill the random inputs, replay, then (eagerly, outside the graph) the flat gradient all-reduce over
    RCCL, the clip and the optimizer step - the exchange keeps its own place between backward and update, as in
    `train_step`. Device-side generators (dropout, teacher-forcing Bernoulli draws) advance per replay through
    torch's graph-safe Philox offsets.

    Re-capture when the epoch changes (TeacherForcing's schedules read `current_epoch` on the host).

    ROCm caveat (7.0.x, measured with tools/hipgraph_memset_repro.py): on the runtime's AQL-packet fast path hipGraph memset nodes
    are not ordered with the kernels around them, and torch's multi-block reductions zero their semaphores with one - a
    replay then returns bias gradients of the PREVIOUS replay. `DEBUG_CLR_GRAPH_PACKET_CAPTURE=0` (read when HIP initialises)
    selects the ordered path at no measurable cost; the constructor refuses to capture without it.
    """

    def __init__(self, wm, optimizer: torch.optim.Optimizer, example_batch: Dict[str, Tensor], clip_grad_norm: float = 5.0,
                 warmup: int = 2, verbose: bool = False) -> None:
        if os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") != "0":
            raise RuntimeError("GraphedTrainStep needs DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in the environment before HIP "
                               "initialises (hipGraph memset nodes replay out of order otherwise: stale gradients)")
        self.wm, self.opt, self.clip = wm, optimizer, clip_grad_norm
        dev = next(wm.model.parameters()).device
        self.static = self._pre(example_batch)  # static input buffers: the re-keyed (sc/*, gt/*, ref/*) batch
        n, A = example_batch["agent/valid"].shape[:2]
        self.noise = torch.zeros(n, A, wm.model.latent_encoder.out_dim, device=dev)
        self.use_prior = torch.zeros((), dtype=torch.bool, device=dev)
        self._noise_pin, self._noise_turn = None, 0  # pinned staging of the host-drawn noise (_refill)
        wm.attn_dropout_seed = self.drop_seed = torch.zeros(1, dtype=torch.int64, device=dev)  # static: the graph reads it
        self.live: Optional[List[torch.nn.Parameter]] = None
        self._say = (lambda *a: print("[GraphedTrainStep]", *a, flush=True)) if verbose else (lambda *a: None)
        self._dev, self._warmup = dev, warmup
        self.graph = self.flat = self.flat_opt = None
        self._recapture()

    def _recapture(self) -> None:
        """Warm-up + capture for the module's CURRENT epoch (constructor; again from __call__ when the epoch has changed). The old
        graph and its private pool are released first; the live-parameter list and the optimizer are kept."""
        wm, dev = self.wm, self._dev
        if self.graph is not None:
            torch.cuda.synchronize(dev)
            if self.flat is not None:
                self.flat.detach()  # (the parameters' .grad views of the old static buffer)
            self.graph = self.flat = None
            import gc

            gc.collect()
        self.epoch = wm.current_epoch
        # hipStreamEndCapture walks the captured graph recursively (~10^5 nodes in a chain: the default 8 MiB stack overflows,
        # measured); warm-up + capture therefore run on a thread with a 1 GiB (virtual, lazily committed) stack
        err: List[BaseException] = []

        def work():
            try:
                torch.cuda.set_device(dev)
                self._capture(self.opt, dev, self._warmup)
            except BaseException as e:  # noqa: BLE001 - re-raised on the caller's thread
                err.append(e)

        old = threading.stack_size(1 << 30)
        try:
            t = threading.Thread(target=work, name="tbx-train-capture")
            t.start()
            t.join()
        finally:
            threading.stack_size(old)
        if err:
            raise err[0]
        self.metrics = dict(wm.last_metrics)

    @property
    def grads(self) -> List[Tensor]:
        """The live parameters' gradients: slices of the static flat buffer every replay rewrites."""
        return self.flat.views

    def _capture(self, optimizer, dev, warmup: int) -> None:
        wm, say = self.wm, self._say
        # warm-up on the stream the capture will use (allocator / library workspaces / lazy inits must not happen inside the
        # capture, and autograd's AccumulateGrad nodes must not outlive an iteration on another stream)
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for i in range(warmup):
                self._refill()
                optimizer.zero_grad(set_to_none=True)
                self._fwd_bwd()
                self.live = self.live or live_parameters(wm.model)
                say("warm-up", i, "done")
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        optimizer.zero_grad(set_to_none=True)
        # the live gradients as views of ONE static buffer: the captured backward accumulates into it, the exchange runs on it
        want_flat_opt = os.environ.get("TBX_FLAT_ADAMW", "1") != "0"
        self.flat = FlatGrads(self.live, align=64 if (want_flat_opt or self.flat_opt is not None) else 1)
        if self.flat_opt is None and want_flat_opt and FlatAdamW.usable(optimizer, self.flat):
            self.flat_opt = FlatAdamW(optimizer, self.flat)  # (parameter storage moves: before the capture reads it)
        elif self.flat_opt is not None:
            self.flat_opt.grads = self.flat  # (a re-capture: the same parameters, a new gradient buffer of the same layout)
        wm.last_metrics = None
        getattr(wm, "logged", {}).clear()
        self.graph = torch.cuda.CUDAGraph()
        say("capture begins")
        # thread_local: only this thread's calls are policed during the capture (the RCCL watchdog / heartbeat threads of an
        # initialised process group may query events at any time); the autograd engine's launches still land on the capturing
        # stream and are captured
        with torch.cuda.graph(self.graph, stream=s, capture_error_mode="thread_local"):
            self.flat.detach()
            self._fwd_bwd()  # every live parameter is handed its gradient (a tensor of the graph's pool: static across replays) ...
            self.flat.gather()  # ... and a few multi-tensor copies move them into the flat buffer the exchange runs on
        say("capture done")

    @torch.no_grad()
    def _pre(self, batch: Dict[str, Tensor]) -> Dict[str, Tensor]:
        return {k: v for k, v in self.wm.pre_processing({k: v.clone() for k, v in batch.items()}).items() if torch.is_tensor(v)}

    def _fwd_bwd(self) -> None:
        from .. import hip_base

        loss = self.wm.training_step(dict(self.static), 0, noise=self.noise, use_prior=self.use_prior)
        # The backward packs weight images too (TallLinearFn.backward: the W^T image of every tall LINEAR's input gradient). Outside a
        # scope those come from the per-Parameter cache, stamped by the weights' version - and the warm-up passes (no optimizer step
        # between them and the capture) leave that cache at the CAPTURE-time version: the captured backward would hit it, its
        # tbx_pack_weight launch would not be in the graph, and every replay after the first AdamW step would multiply by the W^T of
        # the warm-up weights. With a scope of its own the packing is part of the backward - eager or captured.
        # ... in the scope of the forward that produced `loss`: the W^T images of last step's list were packed with the forward's.
        from ..train_graph import backward_pack_scope

        with backward_pack_scope(self.wm):
            loss.backward()

    def _refill(self) -> None:
        # The latent noise comes from the CPU generator, as on the reference's CPU path. A copy from pageable memory blocks the host until
        # the stream has drained - i.e. until the PREVIOUS step's replay, gather, clip and AdamW are done - and the next replay is only
        # enqueued after it: the device idled between steps. Two pinned staging buffers + an event each: the copy is asynchronous, the host
        # runs one step ahead and only waits if it gets two ahead.
        if self._noise_pin is None:
            self._noise_pin = [(torch.empty(self.noise.shape, dtype=self.noise.dtype).pin_memory(), torch.cuda.Event()) for _ in range(2)]
            self._noise_turn = 0
        buf, ev = self._noise_pin[self._noise_turn]
        self._noise_turn ^= 1
        ev.synchronize()  # (the copy that last read this buffer has run; a fresh event is complete)
        torch.randn(self.noise.shape, out=buf)
        self.noise.copy_(buf, non_blocking=True)
        ev.record()
        self.use_prior.fill_(bool(torch.rand(1) < self.wm.hp.p_training_rollout_prior))
        self.drop_seed.random_()  # new attention-dropout masks for this replay

    def __call__(self, batch: Dict[str, Tensor]) -> Dict[str, Tensor]:
        if self.wm.current_epoch != self.epoch:
            # TeacherForcing's schedules read `current_epoch` on the host while the step is captured (teacher_forcing.py:86-106: the
            # forcing horizon / probabilities decrease per epoch): a new epoch is a new graph. The weights, the optimizer (its
            # state included) and the flat gradient buffer's role stay; warm-up + capture run again (a few seconds, once per epoch).
            self._say("epoch", self.epoch, "->", self.wm.current_epoch, ": re-capturing")
            self._recapture()
        b = self._pre(batch)  # eager (tiny); its advanced-indexing index tensors are host data, not capturable
        for k, v in self.static.items():
            v.copy_(b[k])
        self._refill()
        self.graph.replay()
        self.flat.attach()  # a zero_grad(set_to_none=True) elsewhere must not detach the static buffer's views
        allreduce_gradients(self.flat)
        clip_gradients(self.flat, self.clip)
        if self.flat_opt is not None and self.flat_opt.opt is self.opt:
            self.flat_opt.step()
        else:  # (not a fused AdamW on the device, or the caller has swapped the optimizer)
            self.opt.step()
        return self.metrics
