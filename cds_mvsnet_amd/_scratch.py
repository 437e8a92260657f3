"""Per-step scratch of the training ops: a zero-filled arena (ONE fill per step instead of one `torch.zeros` per weight-gradient / statistics
buffer - the step is launch-bound) and deferred `num_batches_tracked` increments (one `_foreach_add_` per forward instead of ~80 launches).

`training.forward_train` opens a step; the autograd functions ask `zeros()` for accumulation buffers in forward AND backward.  A slice is
handed out once and never reused (a weight gradient carved from the arena may live on as `param.grad`, which keeps the arena's storage
alive), every step gets a fresh arena, and without an open step - or when the arena is exhausted - `zeros()` is `torch.zeros`."""
from __future__ import annotations

import threading
from typing import Dict, List, Optional, Sequence, Tuple

import torch

ARENA_BYTES = 8 << 20


class _Step:
    def __init__(self, device: torch.device):
        self.buf = torch.zeros((ARENA_BYTES // 8,), dtype=torch.float64, device=device)
        self.off = 0
        self.counters: Optional[List[torch.Tensor]] = []
        self.bumps: List[int] = []

    def flush(self) -> None:
        if self.counters:
            torch._foreach_add_(self.counters, self.bumps)
        self.counters = None                                 # the forward is over: later increments (ops called directly) apply at once


class _DevState:
    """Everything this module keeps, per DEVICE: nn.DataParallel drives one replica per device from its own thread, and autograd runs
    a device's backward on that device's worker thread, so per-device state is per-replica state; there are no process-wide flags."""

    def __init__(self):
        self.step: Optional[_Step] = None
        self.side: Optional["torch.cuda.Stream"] = None
        self.side_enabled = False
        self.audit: Optional[List[Tuple[int, int]]] = None


_state: Dict[int, _DevState] = {}
_lock = threading.Lock()


def _idx(device: torch.device) -> int:
    """ONE convention for the device key: an index-less torch.device('cuda') means the current device."""
    return device.index if device.index is not None else torch.cuda.current_device()


def _dev_state(device: torch.device) -> Optional[_DevState]:
    if device.type != "cuda":
        return None
    i = _idx(device)
    st = _state.get(i)
    if st is None:
        with _lock:
            st = _state.setdefault(i, _DevState())
    return st


def begin_step(device: torch.device) -> None:
    ds = _dev_state(device)
    if ds is None:
        return
    if ds.step is not None and ds.step.counters is not None:
        ds.step.flush()                                      # a forward that raised half way: its increments are not dropped
    ds.step = _Step(device)


def flush_counters(device: torch.device) -> None:
    ds = _dev_state(device)
    if ds is not None and ds.step is not None and ds.step.counters is not None:
        ds.step.flush()


def bump(counter: torch.Tensor, n: int = 1) -> None:
    """counter += n, deferred to the end of the forward when a step is open on the counter's device."""
    ds = _dev_state(counter.device)
    st = ds.step if ds is not None else None
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
    ds = _dev_state(device)
    st = ds.step if ds is not None else None
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
# stream it overlaps with the data-gradient chain, whose kernels are mostly too small to fill 256 CUs.  Autograd does not know about
# that stream, so this is only sound when AccumulateGrad TAKES the returned tensor as `param.grad` without launching a kernel on it:
# every parameter must receive exactly one gradient per backward and the tensor a weight-gradient kernel wrote must be handed over
# untouched.  `train.train_step` therefore AUDITS one single-stream backward per model first (`audit_begin` / `audit_end`: every
# buffer a weight-gradient kernel filled must have become some parameter's `.grad` storage) and only then enables the side stream; it
# joins the streams before the gradient all-reduce / optimiser step.


class side_stream_weight_gradients:
    """Context manager: weight-gradient kernels launched inside go to the device's side stream; leaving it makes the current stream
    wait for them."""

    def __init__(self, device: torch.device, enabled: bool = True):
        self.device, self.enabled = device, enabled and device.type == "cuda"

    def __enter__(self):
        ds = _dev_state(self.device)
        if ds is not None:
            self.prev = ds.side_enabled
            ds.side_enabled = self.enabled
        return self

    def __exit__(self, *exc):
        ds = _dev_state(self.device)
        if ds is not None:
            ds.side_enabled = self.prev
            if self.enabled and ds.side is not None:
                torch.cuda.current_stream(self.device).wait_stream(ds.side)
        return False


def side_stream(device: torch.device) -> Optional["torch.cuda.Stream"]:
    """The side stream for a leaf kernel on `device` (it already waits for everything queued on the current stream), or None."""
    ds = _dev_state(device)
    if ds is None or not ds.side_enabled:
        return None
    if ds.side is None:
        ds.side = torch.cuda.Stream(device=device)
    ds.side.wait_stream(torch.cuda.current_stream(device))
    return ds.side


def audit_begin(device: torch.device) -> None:
    ds = _dev_state(device)
    if ds is not None:
        ds.audit = []


def audit_note(dw: torch.Tensor) -> None:
    """Called by the weight-gradient launchers with the buffer their kernel fills."""
    ds = _dev_state(dw.device)
    if ds is not None and ds.audit is not None:
        ds.audit.append((dw.data_ptr(), dw.numel() * dw.element_size()))


def audit_end(device: torch.device) -> List[Tuple[int, int]]:
    ds = _dev_state(device)
    if ds is None or ds.audit is None:
        return []
    out, ds.audit = ds.audit, None
    return out
