"""cds_mvsnet_amd — MI355X-native (gfx950) implementation of the CDS-MVSNet plane-sweep hot path.

``CDSMVSNet`` mirrors ``models.model.CDSMVSNet`` of the reference (constructor, forward signature, output
dict, state-dict keys); the arithmetic runs in hand-written HIP kernels (``libcdsmvs_hip.so``) reached
through the C ABI of ``include/cds_mvsnet_hip.h``.  No CPU / PyTorch fallback exists on the product path.
"""
from .model import CDSMVSNet, CostRegNet, FeatureNet, Refinement, StageNet  # noqa: F401
from .init import seeded_init_  # noqa: F401
from .losses import final_loss  # noqa: F401
from . import train  # noqa: F401  (schedules, gradient all-reduce, train_step)

__all__ = ["CDSMVSNet", "CostRegNet", "FeatureNet", "Refinement", "StageNet", "seeded_init_", "final_loss"]
__version__ = "0.1.0"
