"""CPU oracle for the input pipeline (TEST INFRASTRUCTURE ONLY; see oracle/reference_cpu.py for the rules).

Restates Descriptors/multi_input_data_preprocess_maccs_opt_IsolationForest_fixed_1.py with the reference's own
third-party pieces where they are installed (Pillow, scikit-learn) and a numpy restatement of torchvision's
Resize/ToTensor (torchvision is not installed; on a PIL image Resize((128,128)) IS ``img.resize((128,128),
Image.BILINEAR)`` and ToTensor IS uint8 HWC -> float32 CHW / 255).
"""
from __future__ import annotations

import numpy as np


def load_image_features(path: str) -> np.ndarray:
    """:56-71 -> flat [49152] float32 (CHW)."""
    from PIL import Image
    img = Image.open(path).convert("RGB").resize((128, 128), Image.BILINEAR)
    a = np.asarray(img)                                     # [128,128,3] uint8
    return (a.transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)).reshape(-1)


def resized_bytes(path: str) -> np.ndarray:
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB").resize((128, 128), Image.BILINEAR))


def pil_resize_restated(a: np.ndarray, bounds_x, kk_x, bounds_y, kk_y) -> np.ndarray:
    """Pillow's 8-bit two-pass resampling in numpy integers (validates the coefficient tables used by the HIP kernel)."""
    PB = 32 - 8 - 2
    Hs, Ws, C = a.shape
    Wo, Ho = len(bounds_x), len(bounds_y)
    tmp = np.zeros((Hs, Wo, C), dtype=np.uint8)
    ai = a.astype(np.int64)
    for xo in range(Wo):
        x0, n = bounds_x[xo]
        ss = (ai[:, x0:x0 + n, :] * kk_x[xo, :n].astype(np.int64)[None, :, None]).sum(axis=1) + (1 << (PB - 1))
        tmp[:, xo, :] = np.clip(ss >> PB, 0, 255).astype(np.uint8)
    out = np.zeros((Ho, Wo, C), dtype=np.uint8)
    ti = tmp.astype(np.int64)
    for yo in range(Ho):
        y0, n = bounds_y[yo]
        ss = (ti[y0:y0 + n, :, :] * kk_y[yo, :n].astype(np.int64)[:, None, None]).sum(axis=0) + (1 << (PB - 1))
        out[yo] = np.clip(ss >> PB, 0, 255).astype(np.uint8)
    return out


def standardize_features(maccs_u8: np.ndarray, images_f32: np.ndarray, batch_size: int = 100):
    """:86-101, with scikit-learn's own StandardScaler exactly as the reference calls it."""
    from sklearn.preprocessing import StandardScaler
    scaler = StandardScaler()
    out = []
    for i in range(0, len(maccs_u8), batch_size):
        feats = np.hstack([maccs_u8[i:i + batch_size], images_f32[i:i + batch_size]])
        out.extend(scaler.fit_transform(feats))
    out = np.array(out, dtype=np.float32)
    F = maccs_u8.shape[1]
    return out[:, :F], out[:, F:]
