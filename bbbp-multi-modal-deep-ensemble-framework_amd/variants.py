"""The reference's MLP-only variants of ``MixedInputModel`` (same class name in every script; here one class per
script, same constructor signature ``(fingerprint_size, image_feature_size)`` and ``state_dict`` keys), running on
the HIP GEMM / BatchNorm / dropout / fusion ops through per-op autograd nodes.

* ``PCAFusionModel``  -- Models/multi_input_data_regression_opt_transformer_cnn_opt.py:72-105 (also _morgan.py):
  PCA-reduced fingerprint and image each through Linear+ReLU, attention fusion, 256->128->64->1.  This is the
  architecture of the shipped ``best_nn_model*.pth``.
* ``RdkitPCAFusionModel`` -- Models/multi_input_data_regression_opt_transformer_cnn_rdkit.py:53-105: the same with the
  single-head ``AttentionFusion``.
* ``OptMoreFusionModel`` -- Models/multi_input_data_regression_opt_transformer_cnn_opt_more.py:80-107: the PCA-MLP fusion model with
  256-wide branches, ReLU -> BatchNorm1d -> Dropout(0.3) in each, fusion over 512 columns and a 512->256->128->1 BatchNorm head.
* ``DenseMLPModel``   -- Models/multi_input_data_regression_opt.py:41-85: raw fingerprint F->512->256->128 and raw image
  49152->1024->256->128 with ReLU -> BatchNorm1d -> Dropout(0.2), concat, BatchNorm head.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from .functional import run_sequential
from .models import MixedInputModel as _FusedMixedInputModel
from .models import MultiHeadAttentionFusion, flatten_parameters


def _need_cuda(x):
    if not x.is_cuda:
        raise RuntimeError("this model runs on MI355X only (HIP kernels); there is no CPU fallback")


class PCAFusionModel(nn.Module):
    def __init__(self, fingerprint_size, image_feature_size):
        super().__init__()
        self.fingerprint_fc = nn.Sequential(nn.Linear(fingerprint_size, 128), nn.ReLU())
        self.image_fc = nn.Sequential(nn.Linear(image_feature_size, 128), nn.ReLU())
        self.attention_fusion = MultiHeadAttentionFusion(256)
        self.fc = nn.Sequential(nn.Linear(256, 128), nn.ReLU(), nn.Linear(128, 64), nn.ReLU(), nn.Linear(64, 1))
        flatten_parameters(self)

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        flatten_parameters(self)
        return out

    def forward(self, fingerprint, image):
        _need_cuda(fingerprint)
        a = run_sequential(self.fingerprint_fc, fingerprint.float().contiguous())
        b = run_sequential(self.image_fc, image.float().contiguous())
        return run_sequential(self.fc, self.attention_fusion(a, b))


class AttentionFusion(nn.Module):
    """Single-head fusion block of Models/multi_input_data_regression_opt_transformer_cnn_rdkit.py:53-66:
    ``Linear(input_dim,128) -> Tanh -> Linear(128,1) -> Softmax(dim=1)`` then ``weights * combined``.  The softmax normalises
    a size-1 dimension, so ``weights == 1`` whatever the scorer says: output = ``cat(x1, x2)`` exactly, scorer gradients are
    exact zeros.  Same parameter container (``attention.{0,2}.*`` keys) as the reference."""

    def __init__(self, input_dim):
        super().__init__()
        self.attention = nn.Sequential(nn.Linear(input_dim, 128), nn.Tanh(), nn.Linear(128, 1), nn.Softmax(dim=1))

    def forward(self, x1, x2):
        from .functional import attention_fusion
        return attention_fusion(self, x1, x2, heads=[self.attention])


class RdkitPCAFusionModel(nn.Module):
    """``MixedInputModel`` of Models/multi_input_data_regression_opt_transformer_cnn_rdkit.py:69-105: the PCA-MLP fusion
    model above with the single-head ``AttentionFusion``."""

    def __init__(self, fingerprint_size, image_feature_size):
        super().__init__()
        self.fingerprint_fc = nn.Sequential(nn.Linear(fingerprint_size, 128), nn.ReLU())
        self.image_fc = nn.Sequential(nn.Linear(image_feature_size, 128), nn.ReLU())
        self.attention_fusion = AttentionFusion(256)
        self.fc = nn.Sequential(nn.Linear(256, 128), nn.ReLU(), nn.Linear(128, 64), nn.ReLU(), nn.Linear(64, 1))
        flatten_parameters(self)

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        flatten_parameters(self)
        return out

    def forward(self, fingerprint, image):
        _need_cuda(fingerprint)
        a = run_sequential(self.fingerprint_fc, fingerprint.float().contiguous())
        b = run_sequential(self.image_fc, image.float().contiguous())
        return run_sequential(self.fc, self.attention_fusion(a, b))


class OptMoreFusionModel(nn.Module):
    """``MixedInputModel`` of Models/multi_input_data_regression_opt_transformer_cnn_opt_more.py:80-107 (the Descriptors/ copy is the
    same class): PCA-reduced fingerprint and image each through ``Linear(n, 256) -> ReLU -> BatchNorm1d(256) -> Dropout(0.3)``,
    ``MultiHeadAttentionFusion(512)``, head ``Linear(512,256) -> ReLU -> BatchNorm1d(256) -> Linear(256,128) -> ReLU -> Linear(128,1)``.
    Same constructor signature and ``state_dict`` keys as the reference class."""

    def __init__(self, fingerprint_size, image_feature_size):
        super().__init__()
        self.fingerprint_fc = nn.Sequential(nn.Linear(fingerprint_size, 256), nn.ReLU(), nn.BatchNorm1d(256), nn.Dropout(0.3))
        self.image_fc = nn.Sequential(nn.Linear(image_feature_size, 256), nn.ReLU(), nn.BatchNorm1d(256), nn.Dropout(0.3))
        self.attention_fusion = MultiHeadAttentionFusion(512)
        self.fc = nn.Sequential(nn.Linear(512, 256), nn.ReLU(), nn.BatchNorm1d(256), nn.Linear(256, 128), nn.ReLU(), nn.Linear(128, 1))
        flatten_parameters(self)

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        flatten_parameters(self)
        return out

    def forward(self, fingerprint, image):
        _need_cuda(fingerprint)
        a = run_sequential(self.fingerprint_fc, fingerprint.float().contiguous())
        b = run_sequential(self.image_fc, image.float().contiguous())
        return run_sequential(self.fc, self.attention_fusion(a, b))


class DenseMLPModel(nn.Module):
    def __init__(self, fingerprint_size, image_feature_size):
        super().__init__()
        def branch(n_in, wide):
            return nn.Sequential(nn.Linear(n_in, wide), nn.ReLU(), nn.BatchNorm1d(wide), nn.Dropout(0.2),
                                 nn.Linear(wide, 256), nn.ReLU(), nn.BatchNorm1d(256), nn.Linear(256, 128), nn.ReLU())
        self.fingerprint_fc = branch(fingerprint_size, 512)
        self.image_fc = branch(image_feature_size, 1024)
        self.fc = nn.Sequential(nn.Linear(128 + 128, 256), nn.ReLU(), nn.BatchNorm1d(256), nn.Linear(256, 128), nn.ReLU(),
                                nn.Linear(128, 64), nn.ReLU(), nn.Linear(64, 1))
        flatten_parameters(self)

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        flatten_parameters(self)
        return out

    def forward(self, fingerprint, image):
        _need_cuda(fingerprint)
        a = run_sequential(self.fingerprint_fc, fingerprint.float().contiguous())
        b = run_sequential(self.image_fc, image.float().contiguous())
        return run_sequential(self.fc, torch.cat((a, b), dim=1))


class MultiModalAttentionFusion(nn.Module):
    """Fusion block of the wide/deep variant (Models/multi_input_data_regression_opt_transformer_cnn_opt_20250107_network.py:
    51-105).  Faithful to the reference's arithmetic including its accidental broadcast: ``attention_weights[:, 0:1]`` is
    [B,1,1] and multiplies ``fingerprint`` [B,512] into [B,B,512]; the following ``mean(dim=1)`` then averages over the
    BATCH, i.e. row i of the 'weighted' features is a_i * column-mean(features).  Here that is two GEMMs."""

    def __init__(self, fingerprint_dim, image_dim, hidden_dim=128):
        super().__init__()
        self.fingerprint_attention = nn.Sequential(nn.Linear(fingerprint_dim, hidden_dim), nn.Tanh(), nn.Linear(hidden_dim, 1))
        self.image_attention = nn.Sequential(nn.Linear(image_dim, hidden_dim), nn.Tanh(), nn.Linear(hidden_dim, 1))
        self.cross_modal_attention = nn.Sequential(nn.Linear(fingerprint_dim + image_dim, hidden_dim), nn.Tanh(),
                                                   nn.Linear(hidden_dim, fingerprint_dim))
        self.softmax = nn.Softmax(dim=1)

    def forward(self, fingerprint, image):
        from .functional import matmul, softmax_lastdim
        B = fingerprint.shape[0]
        fw = run_sequential(self.fingerprint_attention, fingerprint)          # [B,1]
        iw = run_sequential(self.image_attention, image)                      # [B,1]
        cross = run_sequential(self.cross_modal_attention, torch.cat((fingerprint, image), dim=1))
        aw = softmax_lastdim(torch.cat((fw, iw), dim=1))                      # softmax over the two modality logits
        ones = torch.full((1, B), 1.0 / B, device=fingerprint.device, dtype=torch.float32)
        fp_mean = matmul(ones, fingerprint)                                   # [1,512] batch mean of the features
        img_mean = matmul(ones, image)
        fpw = matmul(aw[:, 0:1].contiguous(), fp_mean)                        # [B,1] x [1,512]
        imgw = matmul(aw[:, 1:2].contiguous(), img_mean)
        return torch.cat((fpw, imgw, cross), dim=1)


_SIDE_STREAMS = {}


def _side_stream(device):
    """One side stream per device for the variants that overlap their branches at the Python level."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


class WideDeepMixedInputModel(nn.Module):
    """The wide/deep ``MixedInputModel`` (same file, :109-174): 12-layer encoder (nhead search from 8 down), 3-stage CNN
    64/128/256, Linear(65536,512), Dropout 0.3, MultiModalAttentionFusion(512,512), 6-layer BatchNorm head from 1536."""

    def __init__(self, fingerprint_size, image_feature_size):
        super().__init__()
        nhead = 8
        while fingerprint_size % nhead != 0 and nhead > 1:
            nhead -= 1
        if fingerprint_size % nhead != 0:
            raise ValueError(f"fingerprint_size={fingerprint_size} must be divisible by nhead={nhead}.")
        self.nhead = nhead
        self.fingerprint_transformer = nn.TransformerEncoder(
            nn.TransformerEncoderLayer(d_model=fingerprint_size, nhead=nhead), num_layers=12, enable_nested_tensor=False)
        self.fingerprint_fc = nn.Sequential(nn.Linear(fingerprint_size, 512), nn.ReLU(), nn.Dropout(0.3))
        S = image_feature_size
        self.image_cnn = nn.Sequential(
            nn.Conv2d(3, 64, kernel_size=3, stride=1, padding=1), nn.ReLU(), nn.MaxPool2d(kernel_size=2, stride=2),
            nn.Conv2d(64, 128, kernel_size=3, stride=1, padding=1), nn.ReLU(), nn.MaxPool2d(kernel_size=2, stride=2),
            nn.Conv2d(128, 256, kernel_size=3, stride=1, padding=1), nn.ReLU(), nn.MaxPool2d(kernel_size=2, stride=2),
            nn.Flatten(), nn.Linear(256 * (S // 8) * (S // 8), 512), nn.ReLU(), nn.Dropout(0.3))
        self.attention_fusion = MultiModalAttentionFusion(512, 512)
        self.fc = nn.Sequential(nn.Linear(512 + 512 + 512, 1024), nn.ReLU(), nn.BatchNorm1d(1024), nn.Linear(1024, 512), nn.ReLU(),
                                nn.Dropout(0.3), nn.Linear(512, 256), nn.ReLU(), nn.Linear(256, 128), nn.ReLU(),
                                nn.Linear(128, 64), nn.ReLU(), nn.Linear(64, 1))
        if S != 128:
            raise ValueError("image_feature_size must be 128: forward reshapes images to 3x128x128")
        flatten_parameters(self)

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        flatten_parameters(self)
        return out

    def forward(self, fingerprint, image):
        from .functional import conv3x3_relu_pool, transformer_encoder
        _need_cuda(fingerprint)
        # round 4: the two branches are independent up to the fusion block, and they are opposites -- the 12-layer encoder is ~250 small
        # launches bound by latency, the CNN three persistent conv stages bound by the matrix pipe -- so the encoder runs on a side stream
        # beside the CNN (the flagship's engine does the same inside its C call).  Autograd replays every node on the stream its forward ran
        # on and synchronises at the boundaries, so the backward pass overlaps the same way.  BBBP_WIDE_OVERLAP=0: one stream.
        cur = torch.cuda.current_stream(fingerprint.device)
        side = _side_stream(fingerprint.device) if os.environ.get("BBBP_WIDE_OVERLAP", "1") != "0" else None
        fingerprint = fingerprint.float()
        if side is not None:
            side.wait_stream(cur)
            fingerprint.record_stream(side)
            with torch.cuda.stream(side):
                x = transformer_encoder(fingerprint, self.fingerprint_transformer, self.nhead, self.training)
                fp_out = run_sequential(self.fingerprint_fc, x)
        else:
            x = transformer_encoder(fingerprint, self.fingerprint_transformer, self.nhead, self.training)
            fp_out = run_sequential(self.fingerprint_fc, x)
        img = image.float().contiguous().view(-1, 3, 128, 128)
        cnn = self.image_cnn
        h = conv3x3_relu_pool(img, cnn[0], side is not None)
        h = conv3x3_relu_pool(h, cnn[3], side is not None)
        h = conv3x3_relu_pool(h, cnn[6], side is not None)
        img_out = run_sequential(nn.Sequential(*list(cnn)[10:]), h.flatten(1))
        if side is not None:
            cur.wait_stream(side)
            fp_out.record_stream(cur)
        return run_sequential(self.fc, self.attention_fusion(fp_out, img_out))


class ExactBatchMixedInputModel(_FusedMixedInputModel):
    """The flagship ``MixedInputModel`` (Models/multi_input_data_regression_opt_transformer_cnn_20250113.py:68-119, same
    parameters and ``state_dict``) for EXACT-global-batch data parallelism (SURVEY.md 8e mode 2): every rank holds B/N
    molecules, and the two places where the reference's function couples the molecules of a mini-batch are made global --
    the encoder attends over the keys/values of ALL ranks (one all-gather of K|V per layer forward, one reduce-scatter of
    dK|dV backward) and the head's BatchNorm1d uses global batch statistics.  With gradients averaged over ranks
    (``distributed.allreduce_gradients``) an N-rank step equals the single-GPU step at batch B up to rounding.
    Round 3: runs on the FUSED engine (one C call per direction, three streams, the fused head) -- the engine calls back for its
    collectives (``bbbp_mixed_desc.collective``, models._make_collective) between the launches that produce and consume them.
    Attention dropout draws its mask per LOCAL row, so with dropout on an N-rank step is a different (equally valid) sample than the
    single-process step."""

    def __init__(self, fingerprint_size, image_feature_size, group=None):
        super().__init__(fingerprint_size, image_feature_size)
        self.group = group
        self.exact_batch = True

    def __getstate__(self):
        state = self.__dict__.copy()
        state["group"] = None                     # process groups do not pickle; a restored model joins the default group
        return state


class PerOpExactBatchMixedInputModel(nn.Module):
    """The same mode composed from per-op autograd nodes on the HIP ops (rounds 1-2; 2.4x slower than the fused engine at one rank):
    kept as the independent cross-check of the fused exact-global-batch engine (tests/test_gpu_exact_batch.py)."""

    def __init__(self, fingerprint_size, image_feature_size, group=None):
        super().__init__()
        from .models import MixedInputModel
        inner = MixedInputModel(fingerprint_size, image_feature_size)      # same module tree => same keys and seeded init
        self.nhead = inner.nhead
        self.fingerprint_transformer = inner.fingerprint_transformer
        self.fingerprint_fc = inner.fingerprint_fc
        self.image_cnn = inner.image_cnn
        self.attention_fusion = inner.attention_fusion
        self.fc = inner.fc
        self.group = group
        flatten_parameters(self)

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        flatten_parameters(self)
        return out

    def forward(self, fingerprint, image):
        import torch.distributed as dist
        from .functional import conv3x3_relu_pool, linear, sync_batchnorm1d, transformer_encoder
        _need_cuda(fingerprint)
        group = self.group if (dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1) else None
        if group is None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            group = dist.group.WORLD
        x = transformer_encoder(fingerprint.float(), self.fingerprint_transformer, self.nhead, self.training, group=group)
        fp_out = run_sequential(self.fingerprint_fc, x)
        img = image.float().contiguous().view(-1, 3, 128, 128)
        h = conv3x3_relu_pool(conv3x3_relu_pool(img, self.image_cnn[0]), self.image_cnn[3])
        img_out = linear(h.flatten(1), self.image_cnn[7].weight, self.image_cnn[7].bias, "relu")
        fused = self.attention_fusion(fp_out, img_out)
        fc = self.fc
        h = linear(fused, fc[0].weight, fc[0].bias, "relu")
        h = sync_batchnorm1d(h, fc[2], group) if group is not None else run_sequential(nn.Sequential(fc[2]), h)
        return run_sequential(nn.Sequential(*list(fc)[3:]), h)
