"""GPU: device preprocessing (Resize BICUBIC + CenterCrop + ToTensor + Normalize) against the
Pillow-based oracle: uint8 image bit-exact, fp32 tensor bit-exact (same IEEE fp32 ops)."""
import numpy as np
import pytest
import torch

import preprocess_oracle as PO
from dbmm_amd import _lib, preprocess as PP

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _img(H, W, seed):
    rng = np.random.RandomState(seed)
    img = rng.randint(0, 256, (H, W, 3)).astype(np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    smooth = (127 + 120 * np.sin(xx / 17.0) * np.cos(yy / 23.0)).astype(np.uint8)
    img[H // 2:] = smooth[H // 2:, :, None]           # half noise (worst case for ringing/clipping), half smooth
    return img


@pytest.mark.parametrize("H,W,n_px", [(300, 400, 224), (400, 300, 224), (224, 224, 224), (500, 333, 224),
                                      (218, 178, 224), (97, 131, 64), (1024, 768, 336), (2000, 3000, 224),
                                      (64, 64, 32), (225, 1000, 224)])
def test_preprocess_matches_pillow_bit_exact(H, W, n_px):
    img = _img(H, W, H + W)
    ref, ref_u8 = PO.transform(img, n_px)
    out, u8 = PP.preprocess_u8(torch.from_numpy(img).to(DEV), n_px, return_u8=True)
    assert np.array_equal(u8.cpu().numpy(), ref_u8)
    assert torch.equal(out.cpu(), ref)


def test_preprocess_batch_ragged_and_errors():
    sizes = [(300, 400), (218, 178), (500, 333), (224, 224)]
    imgs = [_img(h, w, i) for i, (h, w) in enumerate(sizes)]
    out = PP.preprocess_batch([torch.from_numpy(i).to(DEV) for i in imgs], 224)
    assert out.shape == (4, 3, 224, 224)
    for i, im in enumerate(imgs):
        assert torch.equal(out[i].cpu(), PO.transform(im, 224)[0])
    with pytest.raises(_lib.DbmmError):
        PP.preprocess_u8(torch.zeros(10, 10, 3), 8)                              # CPU tensor: no fallback
    with pytest.raises(_lib.DbmmError):
        PP.preprocess_u8(torch.zeros(10, 10, 4, dtype=torch.uint8, device=DEV), 8)
    with pytest.raises(_lib.DbmmError):
        PP.preprocess_batch([], 224)


def test_preprocess_feeds_encode_image():
    """device-preprocessed batch == host-preprocessed batch through the tiny RN tower"""
    from dbmm_amd import synth
    from dbmm_amd.clip.model import build_model
    from dbmm_amd.clip.clip import _transform
    from PIL import Image
    model = build_model(synth.clip_state_dict(2, "tiny-RN")).to(DEV)
    R = model.visual.input_resolution
    imgs = [_img(90 + 7 * i, 120 - 5 * i, i) for i in range(3)]
    host = torch.stack([_transform(R)(Image.fromarray(im, "RGB")) for im in imgs]).to(DEV)
    dev = PP.preprocess_batch([torch.from_numpy(im).to(DEV) for im in imgs], R)
    assert torch.equal(dev, host)
    assert torch.equal(model.encode_image(dev), model.encode_image(host))


def test_uniform_batch_equals_per_image_calls():
    """a batch of one geometry (CelebA's 218 x 178) in two launches == the per-image entry, bit for bit"""
    rng = np.random.RandomState(5)
    imgs = torch.from_numpy(rng.randint(0, 256, (7, 218, 178, 3)).astype(np.uint8)).cuda()
    a = PP.preprocess_uniform(imgs, 224)
    b = PP.preprocess_batch([imgs[i] for i in range(7)], 224)
    assert a.dtype == torch.float32 and torch.equal(a, b)
