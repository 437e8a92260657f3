"""Per-step scratch of the training ops: a zero-filled arena (ONE fill per step instead of one `torch.zeros` per weight-gradient / statistics
buffer - the step is launch-bound) and deferred `num_batches_tracked` increments (one `_foreach_add_` per forward instead of ~80 launches).

`training.forward_train` opens a step; the autograd functions ask `zeros()` for accumulation buffers in forward AND backward.  A slice is
handed out once and never reused (a weight gradient carved from the arena may live on as `param.grad`, which keeps the arena's storage
alive), every step gets a fresh arena, and without an open step - or when the arena is exhausted - `zeros()` is `torch.zeros`."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

ARENA_BYTES = 8 << 20


class _Step:
    def __init__(self, device: torch.device):
        self.buf = torch.zeros((ARENA_BYTES // 8,), dtype=torch.float64, device=device)
        self.off = 0
        self.counters: Optional[List[torch.Tensor]] = []
        self.bumps: List[int] = []


_steps: Dict[int, _Step] = {}


def begin_step(device: torch.device) -> None:
    if device.type == "cuda":
        _steps[device.index if device.index is not None else torch.cuda.current_device()] = _Step(device)


def flush_counters(device: torch.device) -> None:
    st = _steps.get(device.index if device.index is not None else -1) if device.type == "cuda" else None
    if st is not None and st.counters is not None:
        if st.counters:
            torch._foreach_add_(st.counters, st.bumps)
        st.counters = None                                   # the forward is over: later increments (ops called directly) apply at once


def bump(counter: torch.Tensor, n: int = 1) -> None:
    """counter += n, deferred to the end of the forward when a step is open on the counter's device."""
    st = _steps.get(counter.device.index) if counter.is_cuda else None
    if st is None or st.counters is None:
        counter += n
        return
    for i, c in enumerate(st.counters):
        if c is counter:
            st.bumps[i] += n
            return
    st.counters.append(counter)
    st.bumps.append(n)


def zeros(shape: Sequence[int], dtype: torch.dtype, device: torch.device) -> torch.Tensor:
    st = _steps.get(device.index) if device.type == "cuda" else None
    numel = 1
    for s in shape:
        numel *= int(s)
    item = 8 if dtype == torch.float64 else 4
    if st is None or dtype not in (torch.float32, torch.float64) or numel == 0:
        return torch.zeros(tuple(shape), dtype=dtype, device=device)
    n64 = (numel * item + 15) // 16 * 2
    if st.off + n64 > st.buf.numel():
        return torch.zeros(tuple(shape), dtype=dtype, device=device)
    sl = st.buf[st.off:st.off + n64]
    st.off += n64
    if dtype == torch.float32:
        sl = sl.view(torch.float32)
    return sl[:numel].view(tuple(shape))


# ---- weight gradients on a side stream ----------------------------------------------------------------------------------------------
# A weight gradient is a leaf of the backward pass: nothing downstream waits for it until the optimiser runs.  Launched on a second
# stream it overlaps with the data-gradient chain, whose kernels are mostly too small to fill 256 CUs.  OPT-IN (`train.train_step`
# enables it around `loss.backward()` and joins the streams before the gradient all-reduce / optimiser step): a caller that runs
# `backward()` itself and reads `.grad` right away must not have gradients still in flight on a stream it does not know about.
_side: Dict[int, "torch.cuda.Stream"] = {}
_side_enabled = False


class side_stream_weight_gradients:
    """Context manager: weight-gradient kernels launched inside go to a per-device side stream; leaving it makes the current stream
    wait for them."""

    def __init__(self, device: torch.device, enabled: bool = True):
        self.device, self.enabled = device, enabled and device.type == "cuda"

    def __enter__(self):
        global _side_enabled
        self.prev = _side_enabled
        _side_enabled = self.enabled
        return self

    def __exit__(self, *exc):
        global _side_enabled
        _side_enabled = self.prev
        if self.enabled:
            idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
            if idx in _side:
                torch.cuda.current_stream(self.device).wait_stream(_side[idx])
        return False


def side_stream(device: torch.device) -> Optional["torch.cuda.Stream"]:
    """The side stream for a leaf kernel on `device` (it already waits for everything queued on the current stream), or None."""
    if not _side_enabled or device.type != "cuda":
        return None
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _side.get(idx)
    if st is None:
        st = _side[idx] = torch.cuda.Stream(device=device)
    st.wait_stream(torch.cuda.current_stream(device))
    return st
