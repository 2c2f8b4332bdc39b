"""bbbp-multi-modal-deep-ensemble-framework_amd: MI355X (gfx950) implementation of the neural hot path of
FengDushuo/BBBP-Multi-Modal-Deep-Ensemble-Framework (import it as ``bbbp_amd``).

Layout:  csrc/ (HIP kernels + the C ABI of include/bbbp_hip.h), _lib.py (ctypes binding), ops.py (tensor
front end), models.py (the reference's nn.Module interface), optim.py (fused AdamW), ensemble.py (stacked
predict surface), trees.py / boosters.py (random-forest and XGBoost prediction), distributed.py (one process per GPU, RCCL
gradient all-reduce).
"""
from .models import (ConcatMixedInputModel, MixedDataset, MixedInputModel, MSELoss, TwoBranchConcatModel,
                     MultiHeadAttentionFusion, flatten_parameters, reference_nhead)  # noqa: F401
from . import ops  # noqa: F401

__all__ = ["ConcatMixedInputModel", "TwoBranchConcatModel", "MixedDataset", "MixedInputModel", "MSELoss", "MultiHeadAttentionFusion", "flatten_parameters", "reference_nhead", "ops"]
