"""TEST INFRASTRUCTURE ONLY (imported by tests/ and nothing else).

CPU restatement of the reference's image preprocessing, `_transform` in
/root/reference/clip/clip.py:79-86:

    Compose([Resize(n_px, interpolation=BICUBIC), CenterCrop(n_px), _convert_image_to_rgb,
             ToTensor(), Normalize(mean, std)])

torchvision (absent here) implements Resize / CenterCrop on PIL images by calling Pillow --
`img.resize((ow, oh), BICUBIC)` with the smaller edge set to n_px and the other to
int(n_px * long / short), then `img.crop` around int(round((size - n_px) / 2.0)) -- so this
oracle calls Pillow (12.2.0 in this image) directly.  Pinned: Pillow IS the implementation the
reference runs; tests compare the HIP path's uint8 image with Pillow's bit for bit.
"""
import numpy as np
import torch
from PIL import Image

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def resize_crop_u8(img_u8_hwc, n_px):
    """numpy uint8 [H, W, 3] -> uint8 [n_px, n_px, 3]: Resize(BICUBIC) + CenterCrop via Pillow"""
    im = Image.fromarray(np.ascontiguousarray(img_u8_hwc), "RGB")
    w, h = im.size
    nw, nh = (n_px, int(n_px * h / w)) if w <= h else (int(n_px * w / h), n_px)
    im = im.resize((nw, nh), Image.BICUBIC)
    left, top = int(round((nw - n_px) / 2.0)), int(round((nh - n_px) / 2.0))
    im = im.crop((left, top, left + n_px, top + n_px)).convert("RGB")
    return np.asarray(im, dtype=np.uint8)


def transform(img_u8_hwc, n_px):
    """-> float32 [3, n_px, n_px]: ToTensor (/255) + Normalize, as torch fp32 ops"""
    u8 = resize_crop_u8(img_u8_hwc, n_px)
    x = torch.from_numpy(u8.copy()).permute(2, 0, 1).float().div(255.0)
    mean = torch.tensor(CLIP_MEAN).view(3, 1, 1)
    std = torch.tensor(CLIP_STD).view(3, 1, 1)
    return (x - mean) / std, u8
