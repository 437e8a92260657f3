"""Depth-map inference harness (the counterpart of the reference's test.py:153-265 `save_depth`).

    python -m cds_mvsnet_amd.infer --testpath <scenes> --testlist <list.txt> --outdir <out> \
        [--resume ckpt.pth] [--refine] [--num_view 5] [--numdepth 192] [--max_h 512 --max_w 640] [--temperature 0.01]

One process per GPU: with `python -m torch.distributed.run --nproc-per-node N -m cds_mvsnet_amd.infer ...` every rank takes
the reference views `idx % world == rank` (independent depth maps, no collective).  Outputs follow the reference layout:
`<out>/<scan>/depth_est/%08d.pfm`, `confidence/%08d.pfm` (3 channels = stage 1-3 confidences), `cams/%08d_cam.txt`,
`images/%08d.jpg`, ready for the fusion step.
"""
from __future__ import annotations

import argparse
import os
import time

import numpy as np
import torch

from . import CDSMVSNet, seeded_init_
from .mvs_io import EvalScenes, save_outputs


class _Opaque:
    """Inert stand-in for a class the checkpoint pickles but this process does not have (the reference's
    ``parse_config.ConfigParser`` rides along in its checkpoints): accepts any construction / state, does nothing."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        pass

    def __call__(self, *a, **k):
        return _Opaque()


_SAFE_BUILTINS = {"set", "frozenset", "list", "dict", "tuple", "int", "float", "bool", "str", "bytes", "bytearray",
                  "complex", "slice", "range", "object"}


# exact globals a tensor state dict needs; everything else in the pickle stream becomes an inert _Opaque
_SAFE_GLOBALS = {
    ("collections", "OrderedDict"), ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_parameter"),
    ("torch._utils", "_rebuild_tensor"), ("torch", "Size"), ("torch", "device"), ("torch", "dtype"),
    ("torch.serialization", "_get_layout"), ("torch._tensor", "_rebuild_from_type_v2"),
    ("_codecs", "encode"), ("numpy", "dtype"), ("numpy", "ndarray"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"),
    ("numpy._core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "scalar"),
}
_SAFE_TORCH_ATTRS = ({n for n in dir(torch) if n.endswith("Storage")}
                     | {n for n in dir(torch) if isinstance(getattr(torch, n, None), torch.dtype)})


class _placeholder_pickle:
    """A ``pickle_module`` for ``torch.load`` whose unpickler resolves ONLY an allowlist of globals (the tensor / storage
    rebuild helpers, ``torch.Size`` / ``dtype`` / ``device``, ``OrderedDict``, numpy array reconstruction, a few builtin
    containers) and replaces every other global by :class:`_Opaque` instead of importing it: ``torch.hub.load``,
    ``torch.jit.load`` and the like are NOT resolvable through it -- nor is ``torch.storage._load_from_bytes``, which is a plain
    ``torch.load(..., weights_only=False)`` with the default pickle module in disguise (zip-format state dicts never reference it)."""
    import pickle as _pickle
    __name__ = "cds_mvsnet_amd.infer._placeholder_pickle"

    class Unpickler(_pickle.Unpickler):
        def find_class(self, module, name):
            if (module, name) in _SAFE_GLOBALS or (module == "torch" and name in _SAFE_TORCH_ATTRS) \
                    or (module == "torch.storage" and name in ("TypedStorage", "UntypedStorage")) \
                    or (module == "builtins" and name in _SAFE_BUILTINS):
                return super().find_class(module, name)
            return _Opaque

    @staticmethod
    def load(f, **kw):
        return _placeholder_pickle.Unpickler(f, **kw).load()


def load_checkpoint(model: torch.nn.Module, path: str, trust_pickle: bool = False) -> None:
    """Reference checkpoints: {'state_dict': ...} with a 'module.' prefix when saved under DataParallel (test.py:180-187).

    Loaded with ``weights_only=True`` (tensors only).  The checkpoints the reference ships also pickle their
    ``ConfigParser``, which that mode refuses: pass ``trust_pickle=True`` (``--trust-checkpoint``) to read such a file
    through a restricted unpickler that turns every non-torch class into an inert placeholder (the reference itself
    does a full ``torch.load``).  Keys are checked: anything missing from the file, or unexpected in it, raises (the
    reference loads with ``strict=False`` and silently keeps random weights); exempt are ``num_batches_tracked`` counters and,
    for a model built with ``refine=False``, the ``refine_network.*`` entries of a checkpoint trained with refinement."""
    import pickle
    try:
        ck = torch.load(path, map_location="cpu", weights_only=True)
    except (pickle.UnpicklingError, RuntimeError) as e:          # refused global / legacy format; I/O errors propagate
        if isinstance(e, RuntimeError) and "eights only" not in str(e) and "nsupported" not in str(e):
            raise
        if not trust_pickle:
            raise RuntimeError(f"{path}: not loadable with weights_only=True ({type(e).__name__}: {str(e)[:200]}). "
                               "If the file is trusted, retry with trust_pickle=True / --trust-checkpoint.") from e
        ck = torch.load(path, map_location="cpu", weights_only=False, pickle_module=_placeholder_pickle)
    sd = ck["state_dict"] if isinstance(ck, dict) and "state_dict" in ck else ck
    sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}
    res = model.load_state_dict(sd, strict=False)
    has_refine = hasattr(model, "refine_network")
    missing = [k for k in res.missing_keys if not k.endswith("num_batches_tracked")]
    unexpected = [k for k in res.unexpected_keys
                  if not k.endswith("num_batches_tracked") and (has_refine or not k.startswith("refine_network."))]
    if missing or unexpected:
        raise RuntimeError(f"{path}: state dict does not match the model: {len(missing)} missing "
                           f"(e.g. {missing[:3]}), {len(unexpected)} unexpected (e.g. {unexpected[:3]})")


def run(args) -> float:
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    with open(args.testlist) as f:
        scans = [ln.strip() for ln in f if ln.strip()]
    data = EvalScenes(args.testpath, scans, nviews=args.num_view, ndepths=args.numdepth,
                      interval_scale=args.interval_scale, max_h=args.max_h, max_w=args.max_w, refine=args.refine,
                      dataset=args.dataset)
    model = CDSMVSNet(refine=args.refine, ndepths=(48, 32, 8), depth_interals_ratio=(4.0, 1.5, 0.75))
    if args.resume:
        load_checkpoint(model, args.resume, trust_pickle=args.trust_checkpoint)
    else:
        seeded_init_(model, 0)  # no checkpoint given: deterministic synthetic weights (plumbing runs)
    model = model.to(dev).eval()
    last = "stage4" if args.refine else "stage3"
    times = []
    with torch.no_grad():
        for idx in range(rank, len(data), world):
            s = data[idx]
            t0 = time.time()
            imgs = torch.from_numpy(s["imgs"]).unsqueeze(0).to(dev)
            # cameras and the depth range stay on the host: the model needs them there (12 floats per view become kernel
            # arguments), so no device round trip
            cams = {k: torch.from_numpy(v).unsqueeze(0) for k, v in s["proj_matrices"].items()}
            dv = torch.from_numpy(s["depth_values"]).unsqueeze(0)
            out = model(imgs, cams, dv, temperature=args.temperature)
            torch.cuda.synchronize()
            times.append(time.time() - t0)
            confs = [out["stage1"]["photometric_confidence"][0].cpu().numpy(),
                     out["stage2"]["photometric_confidence"][0].cpu().numpy(),
                     out["photometric_confidence"][0].cpu().numpy()]
            save_outputs(args.outdir, s["filename"], out["refined_depth"][0].cpu().numpy(), confs,
                         s["proj_matrices"][last][0], s["imgs"][0])
            print(f"[{rank}] {idx + 1}/{len(data)} {s['filename'].format('depth_est', '.pfm')} {times[-1] * 1e3:.1f} ms", flush=True)
    avg = float(np.mean(times)) if times else 0.0
    print(f"[{rank}] average time: {avg:.4f} s over {len(times)} depth maps")
    if args.fuse:
        # step 2 of the reference's test.py (pcd_filter, test.py:386-396): scans are independent -> shard over ranks
        from .fusion import filter_depth
        if world > 1:  # every rank's depth maps must be on disk before any scan is fused
            if not torch.distributed.is_initialized():
                torch.distributed.init_process_group("nccl", device_id=dev)
            torch.distributed.barrier()
        for i, scan in enumerate(scans):
            if i % world != rank:
                continue
            info = filter_depth(os.path.join(args.testpath, scan), os.path.join(args.outdir, scan),
                                os.path.join(args.outdir, f"{scan}.ply"), conf=[float(c) for c in args.conf.split(",")],
                                thres_disp=args.thres_disp, thres_view=args.thres_view, device=str(dev))
            print(f"[{rank}] {scan}.ply: {info['points']} points, final mask {info['mean_final_mask']:.3f}", flush=True)
    return avg


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--testpath", required=True)
    ap.add_argument("--testlist", required=True)
    ap.add_argument("--outdir", required=True)
    ap.add_argument("--resume", default=None)
    ap.add_argument("--trust-checkpoint", dest="trust_checkpoint", action="store_true",
                    help="allow full unpickling of --resume (the reference's shipped checkpoints need it; runs arbitrary code)")
    ap.add_argument("--refine", action="store_true")
    ap.add_argument("--num_view", type=int, default=5)
    ap.add_argument("--numdepth", type=int, default=192)
    ap.add_argument("--interval_scale", type=float, default=1.06)
    ap.add_argument("--max_h", type=int, default=512)
    ap.add_argument("--max_w", type=int, default=640)
    ap.add_argument("--temperature", type=float, default=0.01)
    ap.add_argument("--dataset", default="dtu", choices=["dtu", "tt", "general"])
    ap.add_argument("--fuse", action="store_true", help="filter + fuse the saved depth maps into <outdir>/<scan>.ply")
    ap.add_argument("--conf", default="0.0,0.0,0.0", help="per-stage confidence thresholds (test.py:61)")
    ap.add_argument("--thres_view", type=int, default=3)
    ap.add_argument("--thres_disp", type=float, default=1.0)
    run(ap.parse_args(argv))


if __name__ == "__main__":
    main()
