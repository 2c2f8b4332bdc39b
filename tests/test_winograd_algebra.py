"""CPU restatement of the Winograd F(2x2, 3x3) algebra that csrc/conv_wino.hip implements (numpy, float32, same operation
order per tile): the transform matrices, the position layout pos = 4 r + c, the 180-degree rotation used for the data
gradient, the bias riding in position (1,1) and the tie stability of flat patches.  Pins the formulas without a GPU."""
import numpy as np
import torch
import torch.nn.functional as F

G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float32)


def transform_filters(w):
    """U[co][ci][r][c] = G g G^T  (wino_prep_kernel: row r of G on the left, row c on the right)."""
    return np.einsum("ra,ocab,sb->ocrs", G, w.astype(np.float32), G).astype(np.float32)


def winograd_conv(x, w, bias=None):
    """x [ci][H][W] -> y [co][H][W], pad 1, tile by tile in float32 like the kernel (row transform, column transform,
    sequential accumulation over channels, A^T m A)."""
    ci, H, W = x.shape
    U = transform_filters(w)
    xp = np.pad(x.astype(np.float32), ((0, 0), (1, 1), (1, 1)))
    y = np.zeros((w.shape[0], H, W), np.float32)
    for ty in range(H // 2):
        for tx in range(W // 2):
            d = xp[:, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]
            t = np.stack([d[:, 0] - d[:, 2], d[:, 1] + d[:, 2], d[:, 2] - d[:, 1], d[:, 1] - d[:, 3]], 1)          # B^T d
            V = np.stack([t[:, :, 0] - t[:, :, 2], t[:, :, 1] + t[:, :, 2], t[:, :, 2] - t[:, :, 1], t[:, :, 1] - t[:, :, 3]], 2)
            m = np.zeros((w.shape[0], 4, 4), np.float32)
            if bias is not None:
                m[:, 1, 1] = bias                                   # a constant at position (1,1) lands on all four outputs
            for c in range(ci):
                m += U[:, c] * V[c][None]
            s0 = m[:, 0] + m[:, 1] + m[:, 2]
            s1 = m[:, 1] - m[:, 2] - m[:, 3]
            y[:, 2 * ty, 2 * tx] = s0[:, 0] + s0[:, 1] + s0[:, 2]
            y[:, 2 * ty, 2 * tx + 1] = s0[:, 1] - s0[:, 2] - s0[:, 3]
            y[:, 2 * ty + 1, 2 * tx] = s1[:, 0] + s1[:, 1] + s1[:, 2]
            y[:, 2 * ty + 1, 2 * tx + 1] = s1[:, 1] - s1[:, 2] - s1[:, 3]
    return y


def test_forward_matches_direct_convolution():
    rng = np.random.default_rng(0)
    x = np.maximum(rng.standard_normal((8, 12, 12)), 0).astype(np.float32)
    w = (0.2 * rng.standard_normal((6, 8, 3, 3))).astype(np.float32)
    b = (0.1 * rng.standard_normal(6)).astype(np.float32)
    ref = F.conv2d(torch.from_numpy(x).double()[None], torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=1)[0].numpy()
    got = winograd_conv(x, w, b)
    assert np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max()


def test_data_gradient_is_the_same_transform_of_the_rotated_filters():
    """dX = conv(dY, w rotated by 180 degrees with the channel roles swapped) -- the DGRAD branch of wino_prep_kernel."""
    rng = np.random.default_rng(1)
    w = (0.2 * rng.standard_normal((6, 4, 3, 3))).astype(np.float32)          # [co][ci]
    dy = rng.standard_normal((6, 8, 8)).astype(np.float32)
    xt = torch.zeros(1, 4, 8, 8, dtype=torch.float64, requires_grad=True)
    F.conv2d(xt, torch.from_numpy(w).double(), padding=1).backward(torch.from_numpy(dy).double()[None])
    w_rot = np.ascontiguousarray(np.flip(w, (2, 3)).transpose(1, 0, 2, 3))   # g'[m = ci][k = co][a][b] = w[co][ci][2-a][2-b]
    got = winograd_conv(dy, w_rot)
    assert np.abs(got - xt.grad[0].numpy()).max() <= 2e-6 * np.abs(xt.grad[0].numpy()).max()


def test_flat_patches_give_bit_equal_outputs_in_a_window():
    """A flat 4x4 patch transforms to a single non-zero position, so the four outputs of its 2x2 window are the SAME float
    and max-pool keeps the first maximum exactly as the direct form does (white image background)."""
    rng = np.random.default_rng(2)
    x = rng.standard_normal((8, 16, 16)).astype(np.float32)
    x[:, 4:12, :] = 0.73
    w = (0.2 * rng.standard_normal((6, 8, 3, 3))).astype(np.float32)
    y = winograd_conv(x, w, np.zeros(6, np.float32))
    for ty in range(3, 5):                                           # windows whose patches (rows 2ty-1 .. 2ty+2) are inside rows 4..11
        for tx in range(1, 7):
            win = y[:, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2].reshape(6, 4)
            assert (win == win[:, :1]).all()
