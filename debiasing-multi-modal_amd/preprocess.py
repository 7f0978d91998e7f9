"""Device-side CLIP preprocessing: the reference's `_transform`
(clip/clip.py:79-86: Resize(n_px, BICUBIC) -> CenterCrop(n_px) -> RGB -> ToTensor -> Normalize)
for decoded RGB uint8 images that are already on the GPU.

The reference's Resize is torchvision handing the PIL image to Pillow's resampler, which for
8-bit images is INTEGER arithmetic (Pillow src/libImaging/Resample.c: double-precision
coefficients normalised per output pixel, rounded to 22-bit fixed point, a horizontal and a
vertical pass each rounded to uint8).  This module rebuilds those coefficient tables on the
host (`resample_coeffs`, same operation order in float64, so the integers are identical) and
the HIP kernels apply them, which makes the uint8 image bit-identical to PIL's and the fp32
tensor bit-identical to ToTensor + Normalize of it (tests/test_gpu_preprocess.py, against PIL
itself).  Only the n_px x n_px crop window is computed.

Difference to the reference: the input must already be RGB (the reference converts after the
crop; for RGB sources the order does not matter).  JPEG decode stays on the host.
"""
import functools
import math

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, require_cuda, stream

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
PRECISION_BITS = 32 - 8 - 2
BICUBIC_SUPPORT = 2.0


def _bicubic(x):
    """Pillow's bicubic_filter (a = -0.5), vectorised; float64 in the C operation order"""
    a = -0.5
    x = np.abs(x)
    near = ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    far = (((x - 5) * x + 8) * x - 4) * a
    return np.where(x < 1.0, near, np.where(x < 2.0, far, 0.0))


def resample_coeffs(in_size, out_size):
    """precompute_coeffs + normalize_coeffs_8bpc of Pillow's Resample.c for the full-image box:
    (bounds int32 [out,2] = (first source index, tap count), coeffs int32 [out, ksize])."""
    scale = in_size / out_size                      # (in1 - in0) / outSize with in0 = 0
    filterscale = max(scale, 1.0)
    support = BICUBIC_SUPPORT * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    xx = np.arange(out_size, dtype=np.float64)
    center = 0.0 + (xx + 0.5) * scale
    xmin = np.trunc(center - support + 0.5).astype(np.int64)        # C (int) cast
    xmin = np.maximum(xmin, 0)
    xmax = np.trunc(center + support + 0.5).astype(np.int64)
    xmax = np.minimum(xmax, in_size) - xmin
    x = np.arange(ksize, dtype=np.int64)[None, :]
    valid = x < xmax[:, None]
    w = _bicubic(((x + xmin[:, None]).astype(np.float64) - center[:, None] + 0.5) * ss)
    w = np.where(valid, w, 0.0)
    ww = np.cumsum(w, axis=1)[:, -1]                # sequential accumulation like the C loop
    k = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    k = np.where(valid, k, 0.0)
    fixed = np.where(k < 0, np.trunc(-0.5 + k * (1 << PRECISION_BITS)), np.trunc(0.5 + k * (1 << PRECISION_BITS)))
    bounds = np.stack([xmin, xmax], axis=1).astype(np.int32)
    return bounds, fixed.astype(np.int32)


def resized_size(w, h, n_px):
    """torchvision Resize(int): smaller edge -> n_px, the other int(n_px * long / short)"""
    return (n_px, int(n_px * h / w)) if w <= h else (int(n_px * w / h), n_px)


@functools.lru_cache(maxsize=256)
def _plan_host(H, W, n_px):
    nw, nh = resized_size(W, H, n_px)
    left, top = int(round((nw - n_px) / 2.0)), int(round((nh - n_px) / 2.0))   # CenterCrop
    hb, hk = resample_coeffs(W, nw)
    vb, vk = resample_coeffs(H, nh)
    hb, hk = hb[left:left + n_px].copy(), hk[left:left + n_px].copy()
    vb, vk = vb[top:top + n_px].copy(), vk[top:top + n_px].copy()
    row0 = int(vb[:, 0].min())
    nrows = int((vb[:, 0] + vb[:, 1]).max()) - row0
    vb[:, 0] -= row0
    return dict(hb=hb, hk=hk, vb=vb, vk=vk, row0=row0, nrows=nrows, size=(nw, nh), crop=(left, top))


_dev_plans = {}


def plan(H, W, n_px, device):
    """coefficient tables of one (H, W) -> n_px geometry, resident on `device` (cached)"""
    key = (H, W, n_px, str(device))
    p = _dev_plans.get(key)
    if p is None:
        h = _plan_host(H, W, n_px)
        p = dict(h, **{k: torch.from_numpy(np.ascontiguousarray(h[k])).to(device) for k in ("hb", "hk", "vb", "vk")})
        if len(_dev_plans) > 256:
            _dev_plans.clear()
        _dev_plans[key] = p
    return p


_F3 = _lib.ctypes.c_float * 3


def preprocess_u8(img_hwc, n_px, out=None, return_u8=False):
    """img_hwc: uint8 [H, W, 3] RGB tensor on the GPU -> float32 [3, n_px, n_px] (what the
    reference's `preprocess(PIL image)` returns), written into `out` when given."""
    require_cuda(img_hwc)
    if img_hwc.dtype != torch.uint8 or img_hwc.dim() != 3 or img_hwc.shape[2] != 3 or not img_hwc.is_contiguous():
        raise _lib.DbmmError("preprocess_u8 expects a contiguous uint8 [H, W, 3] tensor")
    H, W = int(img_hwc.shape[0]), int(img_hwc.shape[1])
    p = plan(H, W, n_px, img_hwc.device)
    if out is None:
        out = torch.empty((3, n_px, n_px), device=img_hwc.device, dtype=torch.float32)
    u8 = torch.empty((n_px, n_px, 3), device=img_hwc.device, dtype=torch.uint8) if return_u8 else None
    ws = torch.empty(p["nrows"] * n_px * 3, device=img_hwc.device, dtype=torch.uint8)
    check(_lib.lib().dbmm_resize_crop_normalize_u8(
        ptr(img_hwc), H, W, ptr(p["hb"]), ptr(p["hk"]), p["hk"].shape[1], ptr(p["vb"]), ptr(p["vk"]), p["vk"].shape[1],
        p["row0"], p["nrows"], n_px, _F3(*CLIP_MEAN), _F3(*CLIP_STD), ptr(out), ptr(u8), ptr(ws), ws.numel(), stream()),
        "resize_crop_normalize_u8")
    return (out, u8) if return_u8 else out


def preprocess_uniform(images_bhwc, n_px, out=None):
    """uint8 [B, H, W, 3] RGB batch of ONE geometry on the GPU (CelebA: every image 218 x 178) -> float32 [B, 3, n_px, n_px] in two
    launches; bit-identical to preprocess_u8 image by image."""
    require_cuda(images_bhwc)
    if images_bhwc.dtype != torch.uint8 or images_bhwc.dim() != 4 or images_bhwc.shape[3] != 3 or not images_bhwc.is_contiguous():
        raise _lib.DbmmError("preprocess_uniform expects a contiguous uint8 [B, H, W, 3] tensor")
    B, H, W = (int(v) for v in images_bhwc.shape[:3])
    p = plan(H, W, n_px, images_bhwc.device)
    if out is None:
        out = torch.empty((B, 3, n_px, n_px), device=images_bhwc.device, dtype=torch.float32)
    ws = torch.empty(B * p["nrows"] * n_px * 3, device=images_bhwc.device, dtype=torch.uint8)
    check(_lib.lib().dbmm_resize_crop_normalize_u8_batch(
        ptr(images_bhwc), B, H, W, ptr(p["hb"]), ptr(p["hk"]), p["hk"].shape[1], ptr(p["vb"]), ptr(p["vk"]), p["vk"].shape[1],
        p["row0"], p["nrows"], n_px, _F3(*CLIP_MEAN), _F3(*CLIP_STD), ptr(out), None, ptr(ws), ws.numel(), stream()),
        "resize_crop_normalize_u8_batch")
    return out


def preprocess_batch(images, n_px):
    """list of uint8 [H_i, W_i, 3] GPU tensors (ragged sizes) -> float32 [B, 3, n_px, n_px]"""
    if not images:
        raise _lib.DbmmError("preprocess_batch: empty image list")
    out = torch.empty((len(images), 3, n_px, n_px), device=images[0].device, dtype=torch.float32)
    for i, im in enumerate(images):
        preprocess_u8(im, n_px, out=out[i])
    return out
