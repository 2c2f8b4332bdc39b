"""CPU: the coefficient tables the HIP resize kernel consumes reproduce Pillow bit for bit on the reference's own molecule
drawings (tests/golden/img/*.png are data files from Descriptors/img_output/)."""
import glob
import os

import numpy as np

from bbbp_amd.preprocess import pil_resample_coeffs
from oracle import preprocess_cpu as oracle
from helpers import GOLDEN

PNGS = sorted(glob.glob(os.path.join(GOLDEN, "img", "*.png")))


def test_coefficients_reproduce_pillow_bit_exactly():
    from PIL import Image
    assert len(PNGS) == 8
    bx, kx, ksx = pil_resample_coeffs(300, 128)
    assert ksx == 7 and bx.shape == (128, 2) and kx.shape == (128, 7)
    assert int(kx.sum(axis=1).min()) >= (1 << 22) - 4 and int(kx.sum(axis=1).max()) <= (1 << 22) + 4     # weights sum to ~1.0
    for p in PNGS:
        a = np.asarray(Image.open(p).convert("RGB"))
        assert a.shape == (300, 300, 3)
        got = oracle.pil_resize_restated(a, bx, kx, bx, kx)
        assert np.array_equal(got, oracle.resized_bytes(p)), p
    # non-square and upscaling cases
    rng = np.random.default_rng(0)
    for (h, w, ho, wo) in ((50, 70, 128, 128), (301, 299, 64, 32), (128, 128, 128, 128)):
        a = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        bxx, kxx, _ = pil_resample_coeffs(w, wo)
        byy, kyy, _ = pil_resample_coeffs(h, ho)
        want = np.asarray(Image.fromarray(a).resize((wo, ho), Image.BILINEAR))
        assert np.array_equal(oracle.pil_resize_restated(a, bxx, kxx, byy, kyy), want), (h, w, ho, wo)


def test_oracle_image_features_layout():
    f = oracle.load_image_features(PNGS[0])
    assert f.shape == (49152,) and f.dtype == np.float32 and 0.0 <= f.min() and f.max() <= 1.0
    b = oracle.resized_bytes(PNGS[0])
    assert f[128 * 128 + 5 * 128 + 7] == np.float32(b[5, 7, 1]) / np.float32(255)      # CHW: channel 1, row 5, col 7
