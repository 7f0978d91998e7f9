"""CPU: the host-side coefficient tables of the device preprocessing path equal Pillow's --
checked end to end by running the same integer two-pass resampler in numpy with OUR tables
and comparing the uint8 result with Pillow's resize + crop bit for bit."""
import numpy as np
import pytest

import preprocess_oracle as PO
from dbmm_amd import preprocess as PP


def _apply_tables(img, p, n_px):
    """numpy restatement of the two HIP kernels (int32 arithmetic, >> 22, clip to uint8)"""
    hb, hk, vb, vk, row0, nrows = p["hb"], p["hk"], p["vb"], p["vk"], p["row0"], p["nrows"]
    tmp = np.zeros((nrows, n_px, 3), np.uint8)
    src = img[row0:row0 + nrows].astype(np.int64)
    for x in range(n_px):
        x0, n = int(hb[x, 0]), int(hb[x, 1])
        acc = (1 << 21) + (src[:, x0:x0 + n, :] * hk[x, :n].astype(np.int64)[None, :, None]).sum(1)
        tmp[:, x, :] = np.clip(acc >> 22, 0, 255)
    out = np.zeros((n_px, n_px, 3), np.uint8)
    t64 = tmp.astype(np.int64)
    for y in range(n_px):
        y0, n = int(vb[y, 0]), int(vb[y, 1])
        acc = (1 << 21) + (t64[y0:y0 + n] * vk[y, :n].astype(np.int64)[:, None, None]).sum(0)
        out[y] = np.clip(acc >> 22, 0, 255)
    return out


@pytest.mark.parametrize("H,W,n_px", [(300, 400, 224), (400, 300, 224), (224, 224, 224), (500, 333, 224),
                                      (218, 178, 224), (97, 131, 64), (1024, 768, 336), (64, 64, 32), (225, 1000, 224)])
def test_tables_reproduce_pillow(H, W, n_px):
    rng = np.random.RandomState(H * 7 + W)
    img = rng.randint(0, 256, (H, W, 3)).astype(np.uint8)
    img[: H // 3] = (np.linspace(0, 255, W)[None, :, None] + rng.randint(0, 3, (H // 3, W, 3))).clip(0, 255).astype(np.uint8)
    p = PP._plan_host(H, W, n_px)
    ours = _apply_tables(img, p, n_px)
    ref = PO.resize_crop_u8(img, n_px)
    assert ours.shape == ref.shape == (n_px, n_px, 3)
    assert np.array_equal(ours, ref)


def test_resized_size_and_crop_follow_torchvision():
    assert PP.resized_size(400, 300, 224) == (298, 224)
    assert PP.resized_size(178, 218, 224) == (224, 274)          # CelebA aligned images
    p = PP._plan_host(218, 178, 224)
    assert p["size"] == (224, 274) and p["crop"] == (0, 25)
    assert p["hk"].dtype == np.int32 and p["hb"].shape == (224, 2)
    # every coefficient row sums to 2^22 up to the per-tap rounding
    assert np.all(np.abs(p["hk"].sum(1) - (1 << 22)) <= p["hk"].shape[1])
