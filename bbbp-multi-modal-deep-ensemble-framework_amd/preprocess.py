"""GPU input pipeline at the tensor boundary: the host-side mirror of the reference's preprocessing functions
(Descriptors/multi_input_data_preprocess_maccs_opt_IsolationForest_fixed_1.py):

* ``load_image_features``  (:56-71)  PNG -> convert('RGB') -> Resize((128,128)) -> ToTensor -> flatten  = [49152] f32
* ``standardize_features`` (:86-101) StandardScaler().fit_transform on every chunk of 100 rows of hstack([MACCS, image])

PNG decoding happens on the host (PIL); everything after the decoded bytes runs in libbbbp_hip.so and is bit-identical to
Pillow / torchvision / scikit-learn (tests/test_gpu_preprocess.py).  Fingerprint GENERATION (RDKit) is not part of this
path: fingerprints arrive as uint8 bit vectors.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import _lib, ops

PRECISION_BITS = 32 - 8 - 2


def pil_resample_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """Pillow's antialiased BILINEAR coefficients for one axis (src/libImaging/Resample.c: precompute_coeffs +
    normalize_coeffs_8bpc), the same double arithmetic in the same order.  Returns (bounds [out,2] int32 = first source
    index and count, kk [out, ksize] int32 fixed-point weights, ksize)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale                      # bilinear filter support = 1.0
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = np.zeros(ksize, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            v = (x + xmin - center + 0.5) * ss
            v = -v if v < 0.0 else v
            w = (1.0 - v) if v < 1.0 else 0.0
            k[x] = w
            ww += w
        for x in range(xmax):
            if ww != 0.0:
                k[x] /= ww
        bounds[xx] = (xmin, xmax)
        for x in range(ksize):
            kk[xx, x] = int(-0.5 + k[x] * (1 << PRECISION_BITS)) if k[x] < 0 else int(0.5 + k[x] * (1 << PRECISION_BITS))
    return bounds, kk, ksize


_coeff_cache = {}


def _coeffs_on(device, in_size, out_size):
    key = (str(device), in_size, out_size)
    if key not in _coeff_cache:
        b, k, ks = pil_resample_coeffs(in_size, out_size)
        _coeff_cache[key] = (torch.from_numpy(b).to(device), torch.from_numpy(k).to(device), ks)
    return _coeff_cache[key]


def resize_totensor(images_u8: torch.Tensor, size: Tuple[int, int] = (128, 128), return_bytes: bool = False):
    """[N, Hs, Ws, 3] uint8 (device) -> [N, 3*H*W] float32 in CHW order, values byte/255: the reference's
    Resize((128,128)) + ToTensor() + flatten.  ``return_bytes`` also returns the resized [N, H, W, 3] uint8 image."""
    if not images_u8.is_cuda or images_u8.dtype != torch.uint8 or images_u8.dim() != 4 or images_u8.shape[3] != 3:
        raise RuntimeError("resize_totensor: expected a CUDA uint8 tensor [N, H, W, 3]")
    x = images_u8.contiguous()
    N, Hs, Ws, _ = x.shape
    Ho, Wo = size
    bx, kx, ksx = _coeffs_on(x.device, Ws, Wo)
    by, ky, ksy = _coeffs_on(x.device, Hs, Ho)
    tmp = torch.empty((N, Hs, Wo, 3), dtype=torch.uint8, device=x.device)
    out8 = torch.empty((N, Ho, Wo, 3), dtype=torch.uint8, device=x.device) if return_bytes else None
    out = torch.empty((N, 3 * Ho * Wo), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().bbbp_resize_bilinear_totensor(ops._stream(), x.data_ptr(), tmp.data_ptr(),
                                                        out8.data_ptr() if out8 is not None else None, out.data_ptr(),
                                                        bx.data_ptr(), kx.data_ptr(), ksx, by.data_ptr(), ky.data_ptr(), ksy,
                                                        N, Hs, Ws, Ho, Wo), "bbbp_resize_bilinear_totensor")
    return (out, out8) if return_bytes else out


def load_image_features(paths: Sequence[str], device="cuda") -> torch.Tensor:
    """Batched ``load_image_features`` (reference :56-71): decode with PIL on the host, resize + ToTensor on the GPU.
    All images of a batch must share one size (the reference's RDKit drawings are 300x300)."""
    from PIL import Image
    arrs: List[np.ndarray] = [np.asarray(Image.open(p).convert("RGB")) for p in paths]
    if len({a.shape for a in arrs}) != 1:
        raise RuntimeError("load_image_features: images of one batch must have the same size")
    batch = torch.from_numpy(np.stack(arrs)).to(device, non_blocking=True)
    return resize_totensor(batch)


def standardize_features(fingerprints_u8: torch.Tensor, images: torch.Tensor, batch_size: int = 100):
    """Reference ``standardize_features`` (:86-101): a fresh StandardScaler fit on every chunk of ``batch_size`` rows of
    hstack([fingerprint, image]).  Returns (fingerprint_normalized [N,F] f32, image_normalized [N,I] f32)."""
    if not (fingerprints_u8.is_cuda and images.is_cuda):
        raise RuntimeError("standardize_features: expected CUDA tensors; there is no CPU fallback")
    if fingerprints_u8.dtype != torch.uint8 or images.dtype != torch.float32:
        raise RuntimeError("standardize_features: fingerprints must be uint8 bit vectors and images float32")
    fp = fingerprints_u8.contiguous()
    img = images.contiguous()
    N, F = fp.shape
    I = img.shape[1]
    if img.shape[0] != N:
        raise RuntimeError("standardize_features: row counts differ")
    fo = torch.empty((N, F), dtype=torch.float32, device=fp.device)
    io = torch.empty((N, I), dtype=torch.float32, device=fp.device)
    L = _lib.lib()
    for i in range(0, N, batch_size):
        n = min(batch_size, N - i)
        _lib.check(L.bbbp_standardize_chunk(ops._stream(), fp[i:i + n].data_ptr(), img[i:i + n].data_ptr(), fo[i:i + n].data_ptr(),
                                            io[i:i + n].data_ptr(), None, None, n, F, I), "bbbp_standardize_chunk")
    return fo, io


class HostFedBatches:
    """Streaming input for data that lives in HOST memory (the step before the hot path: the reference's DataLoader hands over
    per-sample ``torch.tensor`` copies and one synchronous ``.to(device)`` per batch, Models/...20250113.py:31-45,184-186 --
    197 KB per molecule, 101 MB per batch of 512).  Batches are consecutive rows (wrapping around), copied by DMA on a dedicated copy
    stream into one of two device buffers ONE BATCH AHEAD of the step that consumes them, so the transfer of batch k+1 runs under
    the compute of batch k.  Host memory: a dataset that is already pinned (or ``pin="all"``, for sets that comfortably fit) is
    read in place; otherwise the rows of a batch are first gathered into one of TWO pinned batch-sized staging buffers -- a
    screening library is never duplicated into page-locked memory.  ``next()`` makes the current stream wait for the batch's copy
    event and returns device views; a buffer is refilled only after the step that read it has been enqueued (event on the
    compute stream).  The tensors are bit-identical to ``host_rows.to(device)``.

    Shuffled epochs over a dataset that fits in HBM (B3DB: 208 MB) should stay device-resident (``training.train_fold``); this
    class is for libraries that do not (screening)."""

    def __init__(self, fingerprints, images, labels, batch_size: int, device, start: int = 0, pin: str = "staging"):
        self.host = tuple(torch.as_tensor(t).contiguous() for t in (fingerprints, images, labels))
        n = self.host[0].shape[0]
        if any(t.shape[0] != n for t in self.host) or batch_size < 1 or n < 1:
            raise ValueError("HostFedBatches: fingerprints, images and labels must have the same, non-zero number of rows")
        if pin not in ("staging", "all"):
            raise ValueError("HostFedBatches: pin must be 'staging' or 'all'")
        if pin == "all":
            self.host = tuple(t if t.is_pinned() else t.pin_memory() for t in self.host)
        self.n, self.batch, self.device = n, int(batch_size), torch.device(device)
        # tensors that are not page-locked go through two pinned batch-sized staging buffers (one per device buffer)
        self.stage = [tuple(None if t.is_pinned() else torch.empty((self.batch,) + tuple(t.shape[1:]), dtype=t.dtype).pin_memory()
                            for t in self.host) for _ in range(2)]
        self.staged = [torch.cuda.Event() for _ in range(2)]       # the DMA out of staging buffer `slot` has finished
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.bufs = [tuple(torch.empty((self.batch,) + tuple(t.shape[1:]), dtype=t.dtype, device=self.device) for t in self.host)
                     for _ in range(2)]
        self.ready = [torch.cuda.Event() for _ in range(2)]
        self.free = [torch.cuda.Event() for _ in range(2)]
        self.pos, self.k = start % n, 0
        self._issue(0)

    def _rows(self, pos):
        """Row ranges of the batch that starts at ``pos`` (two ranges when it wraps around the end of the dataset)."""
        out, need, p = [], self.batch, pos
        while need:
            take = min(need, self.n - p)
            out.append((p, p + take)); need -= take; p = (p + take) % self.n
        return out

    def _issue(self, slot):
        with torch.cuda.stream(self.copy_stream):
            if self.k >= 2:
                self.copy_stream.wait_event(self.free[slot])          # the step that read this buffer has been enqueued and must finish
            if self.k >= 2 and any(s is not None for s in self.stage[slot]):
                self.staged[slot].synchronize()                       # host: the previous DMA out of this staging buffer is done
            off = 0
            for lo, hi in self._rows(self.pos):
                for dst, src, stg in zip(self.bufs[slot], self.host, self.stage[slot]):
                    if stg is not None:
                        stg[off:off + hi - lo].copy_(src[lo:hi])      # host gather into page-locked memory (memcpy)
                        src_rows = stg[off:off + hi - lo]
                    else:
                        src_rows = src[lo:hi]
                    dst[off:off + hi - lo].copy_(src_rows, non_blocking=True)
                off += hi - lo
            self.staged[slot].record(self.copy_stream)
            self.ready[slot].record(self.copy_stream)
        self.pos = (self.pos + self.batch) % self.n

    def next(self):
        """Device tensors (fingerprints, images, labels) of the next batch; valid until the call after next."""
        cur = torch.cuda.current_stream(self.device)
        slot = self.k % 2
        other = slot ^ 1
        self.k += 1
        if self.k >= 2:
            self.free[other].record(cur)             # everything that read the other buffer is on the compute stream by now
        cur.wait_event(self.ready[slot])
        out = self.bufs[slot]
        self._issue(other)                           # batch k+1 streams in while batch k is computed
        return out

    def host_batch(self, index: int):
        """The host rows of batch ``index`` counted from ``start`` (for checks and CPU baselines)."""
        rows = torch.cat([torch.arange(lo, hi) for lo, hi in self._rows((index * self.batch) % self.n)])
        return tuple(t[rows] for t in self.host)
