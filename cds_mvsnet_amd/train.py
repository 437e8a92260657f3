"""Training loop pieces around ``CDSMVSNet.train()``: what the reference's ``trainer/trainer.py`` and
``train.py`` do, laid out for one process per GPU over RCCL instead of ``nn.DataParallel``.

* :func:`temperature_for_epoch` — DynamicConv softmax temperature schedule (trainer/trainer.py:45-49).
* :func:`make_optimizer` / :func:`make_scheduler` — SGD(lr 1e-4, weight_decay 0.01) + StepLR(step 3, gamma 0.5)
  (configs/config_blended.json:34-50); the scheduler steps once per epoch (trainer.py:94).
* :class:`GradAllReducer` — data-parallel gradient averaging: all gradients are packed into flat fp32 buckets
  (981 622 parameters = 3.9 MB -> one bucket, one RCCL all-reduce per step: the exchange is latency-bound on xGMI, so
  fewer/larger messages, not NCCL-style 25 MB overlap buckets), divided by the world size and unpacked.  BatchNorm
  statistics stay per replica like ``nn.DataParallel``.
* :func:`train_step` — forward, ``final_loss``, backward, gradient exchange, optimizer step (trainer.py:69-82), all fp32.

Precision: the reference trains in fp32 and has no AMP; fp32 is the default here.  BASELINE config 5 is labelled bf16:
``train_step(..., activation_storage="bf16")`` (or ``CDS_TRAIN_ACT_STORAGE=bf16``) selects bf16 STORAGE / fp32 ACCUMULATE for the
FeatureNet activations - the layer inputs, the DynamicConv branch responses and the pre-normalisation maps are kept as bfloat16 from
the forward to the backward pass, the kernels widen on load and accumulate in fp32 / fp64 (``train2d_ops.activation_storage``,
csrc/train2d.hip).  Weights, gradients, statistics, the cost-volume path (K1 / K3 / CostRegNet / soft-argmin) and the loss stay
fp32.  Acceptance against the reference's own step G7 (tests/test_train_bf16_gpu.py): loss within 1e-2 relative, cosine >= 0.99 on the
nine full gradient tensors.  (The rounds-2/3 experiment, ``torch.autocast(bf16)`` around fp32 kernels, only added casts and was removed.)
"""
from __future__ import annotations

import bisect
import contextlib
import weakref
import os
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import _scratch
from .losses import final_loss

SIDE_STREAM_WGRAD = os.environ.get("CDS_TRAIN_SIDE_STREAM", "1") != "0"   # A/B knob: weight-gradient kernels on a second stream

Tensor = torch.Tensor


def temperature_for_epoch(epoch: int) -> float:
    """epoch is 1-based: 1, 10^-0.5, 0.1, 10^-1.5 for epochs 1-4, then 0.01 (trainer/trainer.py:45-49)."""
    if epoch <= 4:
        return float(10.0 ** (-(epoch - 1) / 2.0))
    return 0.01


def make_optimizer(model: torch.nn.Module, lr: float = 1e-4, weight_decay: float = 0.01) -> torch.optim.Optimizer:
    return torch.optim.SGD(model.parameters(), lr=lr, weight_decay=weight_decay)


def make_scheduler(optimizer: torch.optim.Optimizer, step_size: int = 3, gamma: float = 0.5):
    return torch.optim.lr_scheduler.StepLR(optimizer, step_size=step_size, gamma=gamma)


@torch.no_grad()
def broadcast_model_(module: torch.nn.Module, src: int = 0, group: Optional["dist.ProcessGroup"] = None) -> int:
    """Make every rank start from rank ``src``'s parameters AND buffers (BatchNorm running statistics): what
    ``nn.DataParallel`` gets for free by replicating one module (base/base_trainer.py:17-18) has to be done explicitly with
    one process per GPU — each rank's ``CDSMVSNet()`` draws its own random initial weights.  One flat fp32 broadcast
    for the floating-point tensors, one int64 broadcast for the ``num_batches_tracked`` counters.  Returns the number
    of tensors synchronised (0 when there is a single rank)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    src_global = dist.get_global_rank(group, src) if group is not None else src
    tensors = list(module.parameters()) + list(module.buffers())
    n = 0
    for is_float in (True, False):
        sel = [t for t in tensors if t.is_floating_point() == is_float]
        if not sel:
            continue
        flat = torch.cat([t.detach().reshape(-1).to(torch.float32 if is_float else torch.int64) for t in sel])
        dist.broadcast(flat, src=src_global, group=group)
        off = 0
        for t in sel:
            k = t.numel()
            t.copy_(flat[off:off + k].view_as(t).to(t.dtype))
            off += k
        n += len(sel)
    if hasattr(module, "repack"):
        module.repack()          # the copies above bypass nothing, but be explicit: packed eval weights are stale now
    return n


class GradAllReducer:
    """Flat-bucket gradient averaging over a process group (default group; RCCL when the tensors are on the GPU).
    Pass ``module=`` to also broadcast rank 0's initial parameters and buffers at construction (:func:`broadcast_model_`):
    averaged gradients only make sense on identical replicas."""

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 64 << 20,
                 group: Optional["dist.ProcessGroup"] = None, module: Optional[torch.nn.Module] = None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        if module is not None:
            broadcast_model_(module, 0, group)
        self.buckets: List[List[torch.nn.Parameter]] = [[]]
        size = 0
        for p in self.params:
            nbytes = p.numel() * 4
            if self.buckets[-1] and size + nbytes > bucket_bytes:
                self.buckets.append([])
                size = 0
            self.buckets[-1].append(p)
            size += nbytes
        self._flat: List[Optional[Tensor]] = [None] * len(self.buckets)

    def world_size(self) -> int:
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    @torch.no_grad()
    def reduce(self) -> int:
        """Average ``p.grad`` over the ranks in place; parameters without a gradient contribute zeros (a rank whose
        batch did not touch a parameter must still take part in the collective).  Returns the number of collectives."""
        world = self.world_size()
        if world == 1:
            return 0
        for i, bucket in enumerate(self.buckets):
            n = sum(p.numel() for p in bucket)
            dev = bucket[0].device
            flat = self._flat[i]
            if flat is None or flat.device != dev:
                flat = self._flat[i] = torch.empty(n, dtype=torch.float32, device=dev)
            off = 0
            for p in bucket:
                k = p.numel()
                if p.grad is None:
                    flat[off:off + k].zero_()
                else:
                    flat[off:off + k].copy_(p.grad.reshape(-1))
                off += k
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            flat.div_(world)
            off = 0
            for p in bucket:
                k = p.numel()
                if p.grad is None:
                    p.grad = flat[off:off + k].view_as(p).clone()
                else:
                    p.grad.copy_(flat[off:off + k].view_as(p))
                off += k
        return len(self.buckets)


def train_step(model: torch.nn.Module, optimizer: torch.optim.Optimizer, sample: Dict[str, object], temperature: float,
               dlossw: Sequence[float] = (0.5, 1.0, 2.0), reducer: Optional[GradAllReducer] = None,
               activation_storage: Optional[str] = None) -> Tuple[float, float]:
    """One optimisation step on ``sample`` = {imgs, proj_matrices, depth_values, depth: {stageK}, mask: {stageK}}
    (already on the model's device).  Returns (loss, depth_loss) as Python floats.  activation_storage: None = the process default
    (fp32 unless CDS_TRAIN_ACT_STORAGE=bf16), "f32" or "bf16" (module docstring)."""
    loss, depth_loss = _step_tensors(model, optimizer, sample, temperature, dlossw, reducer, activation_storage)
    return float(loss), float(depth_loss)


def _step_tensors(model, optimizer, sample, temperature, dlossw, reducer, activation_storage, geo=None, update: bool = True):
    """train_step without the host read of the loss: returns the two 0-dim device tensors.  geo: the step's geometry block
    (training.train_geometry, uploaded); None = built from the sample here.  update=False stops after the backward pass."""
    from . import train2d_ops, training
    if not model.training:                                   # walking ~1 400 modules costs 1 ms of a CPU-bound 28 ms step
        model.train()
    optimizer.zero_grad(set_to_none=True)
    imgs = sample["imgs"]
    if geo is None:
        geo = training.train_geometry(model, sample["proj_matrices"], sample["depth_values"], imgs.shape[1]).upload(imgs.device)
    with (train2d_ops.activation_storage(activation_storage) if activation_storage is not None else contextlib.nullcontext()):
        with torch.cuda.device(imgs.device):
            outputs = training.forward_train(model, imgs.float(), None, None, sample["depth"], temperature, geo=geo)
    outputs = _to_float(outputs)
    loss, depth_loss = final_loss(outputs, sample["depth"], sample["mask"], dlossw=list(dlossw), depth_interval=geo["dint"])
    # weight gradients on a side stream, joined before anything reads .grad (CDS_TRAIN_SIDE_STREAM=0: everything on one stream)
    _backward(model, loss)
    if update:
        if reducer is not None:
            reducer.reduce()
        optimizer.step()
    return loss.detach(), depth_loss.detach()


_SIDE_VERDICT: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()   # model -> (step structure key, side stream is sound)
_PARAM_CACHE: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()    # model -> [steps until re-read, trainable parameters, key]
_PARAM_RECHECK = 64


def _backward(model: torch.nn.Module, loss: torch.Tensor) -> None:
    """loss.backward() with the weight-gradient kernels on a side stream when that is SOUND for this model (see _scratch.py): the first
    backward of a model runs on one stream and audits that every buffer a weight-gradient kernel filled ended up as the storage of a
    parameter's `.grad` - all of its bytes - and that every parameter accumulated exactly one gradient in the backward (so
    AccumulateGrad launched nothing on a buffer: one gradient per parameter, handed over untouched).  A parameter
    that is used twice (CDS_TRAIN_BATCH_FEATURES=0 runs FeatureNet 2 V times on shared weights) fails the audit and keeps the single
    stream: `grad += dw` on the main stream would race with the kernel still writing `dw`."""
    from . import training
    dev = loss.device
    if not (SIDE_STREAM_WGRAD and dev.type == "cuda"):
        loss.backward()
        return
    # the trainable set is re-read every _PARAM_RECHECK steps, not every step: walking ~1 400 modules for ~390 parameters is ~0.3 ms of
    # a host-bound step; a parameter frozen / unfrozen in between is picked up at the next re-read, and until then the
    # side stream only ever runs weight-gradient kernels whose buffers the audit saw handed over untouched
    cached = _PARAM_CACHE.get(model)
    if cached is None or cached[0] <= 0:
        params = [p for p in model.parameters() if p.requires_grad]
        cached = _PARAM_CACHE[model] = [_PARAM_RECHECK, params, (training.BATCH_FEATURES, tuple(id(p) for p in params))]
    cached[0] -= 1
    params, key = cached[1], (training.BATCH_FEATURES, cached[2][1])    # re-audit when the trainable set (or the batching) changes
    verdict = _SIDE_VERDICT.get(model)
    if verdict is None or verdict[0] != key:
        # the audit: (1) every parameter's gradient is accumulated exactly ONCE in this backward (a second contribution - from any op,
        # not only from a weight-gradient kernel - would be `grad += X` on the main stream while the side-stream kernel still writes the
        # buffer); (2) the gradients found inside a buffer a weight-gradient kernel filled cover ALL of its bytes (a DynamicConv's dw is
        # split into two parameters' gradients: both must have taken their part, untouched)
        counts: dict = {}
        hooks = [p.register_post_accumulate_grad_hook(lambda q: counts.__setitem__(id(q), counts.get(id(q), 0) + 1)) for p in params]
        _scratch.audit_begin(dev)
        try:
            loss.backward()
        finally:
            filled = _scratch.audit_end(dev)
            for h in hooks:
                h.remove()
        grads = sorted((p.grad.data_ptr(), p.grad.numel() * p.grad.element_size()) for p in params if p.grad is not None)
        starts = [g[0] for g in grads]
        ok = bool(filled) and all(c == 1 for c in counts.values())
        for ptr, nbytes in filled:
            i = bisect.bisect_left(starts, ptr)
            covered = 0
            while i < len(grads) and grads[i][0] < ptr + nbytes:
                covered += grads[i][1]
                i += 1
            if covered != nbytes:
                ok = False
                break
        _SIDE_VERDICT[model] = (key, ok)
        return
    with _scratch.side_stream_weight_gradients(dev, verdict[1]):
        loss.backward()


def _to_float(x):
    if isinstance(x, torch.Tensor):
        return x.float() if x.is_floating_point() else x
    if isinstance(x, dict):
        return {k: _to_float(v) for k, v in x.items()}
    return x


# ---------------------------------------------------------------------------------------------------------------------------------------
# the training step as a hipGraph
# ---------------------------------------------------------------------------------------------------------------------------------------
class CapturedTrainStep:
    """``train_step`` captured into a hipGraph and replayed.  An eager step is ~1 000 kernel launches behind ~140
    autograd nodes: ~21 ms of Python / ctypes / autograd-engine time for a step whose kernels finish 0.35 ms after the last launch
     - and the loss read at its end keeps the host from running ahead.  Everything per-sample that the kernels
    need as NUMBERS (epipoles, homographies, depth range, spacings) is device data in the step's geometry block
    (training.train_geometry), so forward + loss + backward (+ the SGD update when there is one rank) are recorded ONCE per
    ``(sample shapes, temperature, learning rate, weight decay, storage policy)`` key and replayed with

        the sample copied into the graph's static input tensors, the geometry block rewritten, one hipGraphLaunch;
        with several ranks: the flat-bucket gradient all-reduce and the optimizer step, eagerly, after the replay.

    The first ``eager_steps`` calls of a key run ``train_step``'s eager path (real steps on real samples: they are also the side-stream
    audit of ``_backward`` and the allocator / LDS-attribute warm-up a capture must not contain).  Returns the step's loss and depth
    loss as 0-dim DEVICE tensors that the next call overwrites (static graph outputs): read them with ``float()`` when a number is
    wanted - reading every step synchronises the host with the GPU, reading every n-th lets the host prepare the next samples'
    geometry while the GPU trains.  A change of the learning rate (StepLR) or the temperature (trainer.py:45-49) is a new key: one
    capture per epoch boundary.  Parity: ``tests/test_graphed_gpu.py`` - captured and eager steps from the same weights give equal
    losses and the same updated weights up to the fp32 atomics noise of the gradients."""

    def __init__(self, model: torch.nn.Module, optimizer: torch.optim.Optimizer, reducer: Optional[GradAllReducer] = None,
                 dlossw: Sequence[float] = (0.5, 1.0, 2.0), activation_storage: Optional[str] = None, eager_steps: int = 2,
                 max_graphs: int = 2):
        self.model, self.optimizer, self.reducer = model, optimizer, reducer
        self.dlossw, self.activation_storage = tuple(dlossw), activation_storage
        # the first backward of a model is the single-stream audit of `_backward`: with eager_steps = 0 and a model that has not run a
        # backward yet, that audit happens inside the capture and the graph keeps the weight gradients on one stream
        self.eager_steps = max(0, int(eager_steps))
        self.max_graphs = max_graphs
        self._entries: Dict[tuple, dict] = {}
        self._stream: Optional["torch.cuda.Stream"] = None
        self.captures = 0

    def _multi_rank(self) -> bool:
        return self.reducer is not None and self.reducer.world_size() > 1

    @staticmethod
    def _sample_tensors(sample) -> List[Tuple[str, Tensor]]:
        out = [("imgs", sample["imgs"])]
        for grp in ("depth", "mask"):
            out += [(f"{grp}.{k}", sample[grp][k]) for k in sorted(sample[grp])]
        return out

    def _key(self, sample, temperature: float, geo) -> tuple:
        groups = tuple((g["lr"], g["weight_decay"], g.get("momentum", 0)) for g in self.optimizer.param_groups)
        shapes = tuple((n, tuple(t.shape), t.dtype) for n, t in self._sample_tensors(sample))
        return (shapes, float(temperature), groups, geo.layout(), self.activation_storage, self._multi_rank())

    def _capture(self, sample, temperature: float, geo) -> dict:
        dev = sample["imgs"].device
        e: dict = {"static": {}, "block": torch.empty((geo.numel(),), dtype=torch.float32, device=dev)}
        seen: Dict[int, Tensor] = {}
        for name, t in self._sample_tensors(sample):        # tensors that alias in the sample (stage4 = stage3) alias in the copy
            e["static"][name] = seen.setdefault(t.data_ptr(), t.detach().clone())
        st_sample = {"imgs": e["static"]["imgs"],
                     "depth": {k: e["static"][f"depth.{k}"] for k in sample["depth"]},
                     "mask": {k: e["static"][f"mask.{k}"] for k in sample["mask"]}}
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=dev)
        st, cur = self._stream, torch.cuda.current_stream(dev)
        geo.upload(dev, into=e["block"])
        st.wait_stream(cur)
        self.optimizer.zero_grad(set_to_none=True)           # the captured backward allocates the gradients from the graph's pool
        e["graph"] = torch.cuda.CUDAGraph()
        with torch.cuda.stream(st):
            with torch.cuda.graph(e["graph"], stream=st):
                e["loss"], e["depth_loss"] = _step_tensors(self.model, self.optimizer, st_sample, temperature, self.dlossw, None,
                                                           self.activation_storage, geo=geo.bind(e["block"]),
                                                           update=not self._multi_rank())
        cur.wait_stream(st)
        self.captures += 1
        e["calls"] = 0
        e["grads"] = [(p, p.grad) for p in self.model.parameters() if p.grad is not None]   # this graph's gradient tensors (its pool)
        return e

    def __call__(self, sample: Dict[str, object], temperature: float) -> Tuple[Tensor, Tensor]:
        from . import training
        imgs = sample["imgs"]
        with torch.cuda.device(imgs.device):
            geo = training.train_geometry(self.model, sample["proj_matrices"], sample["depth_values"], imgs.shape[1])
            key = self._key(sample, temperature, geo)
            e = self._entries.get(key)
            if e is None:
                e = self._entries[key] = {"eager": 0}
                while len(self._entries) > self.max_graphs:
                    self._entries.pop(next(iter(self._entries)))
            if "graph" not in e:
                if e["eager"] < self.eager_steps:            # real steps, eagerly: audit + warm-up
                    e["eager"] += 1
                    return _step_tensors(self.model, self.optimizer, sample, temperature, self.dlossw, self.reducer,
                                         self.activation_storage, geo=geo.upload(imgs.device))
                e.update(self._capture(sample, temperature, geo))
            for name, t in self._sample_tensors(sample):
                dst = e["static"][name]
                if dst.data_ptr() != t.data_ptr():
                    dst.copy_(t, non_blocking=True)
            geo.upload(imgs.device, into=e["block"])
            e["graph"].replay()
            e["calls"] += 1
            for p, g in e["grads"]:                          # `.grad` shows THIS graph's gradients (another key's capture may have re-bound it)
                p.grad = g
            if self._multi_rank():
                self.reducer.reduce()
                self.optimizer.step()
        return e["loss"], e["depth_loss"]
