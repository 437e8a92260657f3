"""hipGraph replay of ``CDSMVSNet``'s inference forward.

An eager forward is ~250 kernel launches behind ~3 ms of Python / ctypes work; at 640x512 that host time IS the forward (3.7 ms
with 3.3 ms of enqueue), and a single 1600x1184 forward after a synchronisation pays it up front (18.3 vs
15.2 ms).  The launches of a forward depend on the image size, the number of views and the model configuration only - every per-call
NUMBER (homographies, epipoles, depth range and hypothesis spacings) is device data in the call's geometry block
(``geometry.GeoBlock``, since round 6) - so the device side (``CDSMVSNet.forward_device``) is captured once per
``(image shape, temperature, geometry layout)`` key into a hipGraph and replayed with

    1. the host camera algebra of THIS call (``CDSMVSNet.geometry_block``), packed into pinned memory,
    2. one asynchronous copy of it into the graph's static geometry block (+ one device copy of the images into the static input),
    3. one ``hipGraphLaunch``.

The eager path stays the reference: ``tests/test_graphed_gpu.py`` asserts ``torch.equal`` between captured and eager outputs (same
kernels, same launch order, same arithmetic), also when the replay is fed other cameras than the capture saw.

    runner = CapturedForward(model)                      # model.eval(), on the GPU
    out = runner(imgs, proj_matrices, depth_values, temperature=0.01)

``out`` holds the graph's static output tensors: valid until the next call with the same key (``clone=True`` copies them out).
``CDSMVSNet.use_graphs(True)`` routes ``model(...)`` through a runner with ``clone=True`` - the nn.Module surface of
models/model.py:140 unchanged.  Weights are baked into a graph by address: a runner re-captures when the packed weights of the model
were rebuilt (``load_state_dict``, ``.to()``, ``train()`` / ``eval()``, optimizer steps: the ``_Packed`` signatures), at the price of
reading ~390 tensor versions per call (~0.1 ms); ``check_weights=False`` skips that for frozen models.
"""
from __future__ import annotations

import weakref
from typing import Dict, Optional, Tuple

import torch

from . import geometry as _geometry

Tensor = torch.Tensor


def _clone_tree(x):
    if isinstance(x, torch.Tensor):
        return x.clone()
    if isinstance(x, dict):
        return {k: _clone_tree(v) for k, v in x.items()}
    return x


class _Entry:
    __slots__ = ("graph", "imgs", "block", "outputs", "weights", "replays")

    def __init__(self):
        self.graph = None
        self.imgs: Optional[Tensor] = None
        self.block: Optional[Tensor] = None
        self.outputs = None
        self.weights = None
        self.replays = 0


class CapturedForward:
    """Per-key hipGraphs of ``model.forward_device`` (module docstring).  One runner per model and device; not thread-safe (like the
    module it wraps: one forward at a time per instance)."""

    def __init__(self, model, check_weights: bool = True, warmup: int = 2, max_graphs: int = 8, weak: bool = False):
        # weak: the runner does not keep the model alive (CDSMVSNet.use_graphs registers runners in a WeakKeyDictionary keyed by the
        # model: a strong reference from the value would pin every model - and the static pools of its graphs - for the process)
        self._model_ref = weakref.ref(model) if weak else (lambda m=model: m)
        self.check_weights = check_weights
        self.warmup = max(1, int(warmup))
        self.max_graphs = max_graphs
        self._entries: Dict[tuple, _Entry] = {}
        self._stream: Optional["torch.cuda.Stream"] = None
        self.captures = 0

    @property
    def model(self):
        m = self._model_ref()
        if m is None:
            raise RuntimeError("CapturedForward: the model of this runner was garbage-collected")
        return m

    # ---- weights -------------------------------------------------------------------------------------------------------------------
    def _weights_signature(self) -> tuple:
        from .model import _PackedHolder
        sig = []
        for m in self.model.modules():
            if isinstance(m, _PackedHolder):
                sig.append((m._packed.generation,) + tuple((t.data_ptr(), t._version) for t in m._packed._tensors(m)))
        return tuple(sig)

    # ---- capture -------------------------------------------------------------------------------------------------------------------
    def _capture(self, imgs: Tensor, geo: "_geometry.GeoBlock", T: float) -> _Entry:
        dev = imgs.device
        e = _Entry()
        e.imgs = torch.empty_like(imgs)
        e.block = torch.empty((geo.numel(),), dtype=torch.float32, device=dev)
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=dev)
        cur = torch.cuda.current_stream(dev)
        e.imgs.copy_(imgs)
        geo.upload(dev, into=e.block)
        st = self._stream
        st.wait_stream(cur)
        with torch.cuda.stream(st), torch.no_grad():
            for _ in range(self.warmup):      # packs the weights, creates the side streams, opts the kernels into > 64 KB of LDS: nothing
                self.model.forward_device(e.imgs, geo, T)   # of that may happen for the first time inside a capture
            st.synchronize()
            e.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(e.graph, stream=st):
                e.outputs = self.model.forward_device(e.imgs, geo.bind(e.block), T)
        cur.wait_stream(st)
        e.weights = self._weights_signature() if self.check_weights else None
        self.captures += 1
        return e

    def _key(self, imgs: Tensor, geo: "_geometry.GeoBlock", T: float) -> tuple:
        m = self.model
        sh = m._view_shard
        if sh is not None:
            raise RuntimeError("CapturedForward: the view-sharded forward exchanges data through RCCL / gloo between the launches; "
                               "capture covers the single-GPU forward")
        return (tuple(imgs.shape), imgs.dtype, imgs.device.index, float(T), geo.layout(), m.refine, m.ndepths)

    # ---- call ----------------------------------------------------------------------------------------------------------------------
    def input_buffer(self, imgs_like: Tensor, proj_matrices, depth_values, temperature: float = 0.001) -> Tensor:
        """The static image tensor of the graph for this key (captured on first use): a data loader that writes the next images straight
        into it saves the device copy of the call."""
        geo = self.model.geometry_block(proj_matrices, depth_values, imgs_like.shape[1])
        return self._entry(imgs_like, geo, float(temperature)).imgs

    def _entry(self, imgs: Tensor, geo, T: float) -> _Entry:
        key = self._key(imgs, geo, T)
        e = self._entries.get(key)
        if e is not None and self.check_weights and e.weights != self._weights_signature():
            e = None                                  # the packed weights were rebuilt: the graph points at the old ones
        if e is None:
            if len(self._entries) >= self.max_graphs and key not in self._entries:
                self._entries.pop(next(iter(self._entries)))      # oldest key out (its graph and static pool are freed)
            e = self._entries[key] = self._capture(imgs, geo, T)
        return e

    @torch.no_grad()
    def __call__(self, imgs: Tensor, proj_matrices, depth_values, temperature: float = 0.001, clone: bool = False):
        m = self.model
        if m.training:
            raise RuntimeError("CapturedForward replays the inference forward: call model.eval() (the training step has its own capture, "
                               "train.CapturedTrainStep)")
        if not imgs.is_cuda:
            raise RuntimeError("cds_mvsnet_amd runs on a ROCm device only (no CPU fallback)")
        B, N, _, Him, Wim = imgs.shape
        H, W = (Him // 2, Wim // 2) if m.refine else (Him, Wim)
        if H % 32 or W % 32:
            raise ValueError("internal resolution must be a multiple of 32 (three stride-2 levels at 1/4 scale)")
        with torch.cuda.device(imgs.device):
            geo = m.geometry_block(proj_matrices, depth_values, N)           # host camera algebra of THIS call
            e = self._entry(imgs, geo, float(temperature))
            if imgs.data_ptr() != e.imgs.data_ptr():
                e.imgs.copy_(imgs, non_blocking=True)
            geo.upload(imgs.device, into=e.block)                            # pinned host -> the graph's static block, asynchronous
            e.graph.replay()
            e.replays += 1
        return _clone_tree(e.outputs) if clone else e.outputs
