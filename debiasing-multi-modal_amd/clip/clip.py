"""`clip.load / clip.tokenize / clip.available_models` with the reference's signatures
(/root/reference/clip/clip.py:89-237) on top of the MI355X model in model.py.

Differences that follow from the environment, not from the algorithm:
  * no network: `load(name)` resolves a model *name* only to a file already present in
    `download_root` (default ~/.cache/clip); it never downloads (clip/clip.py:43-72 would).
  * checkpoints are opened with `torch.load(weights_only=True)` (plain state dict) and, if
    that fails, as a TorchScript archive whose state_dict is taken -- same two formats as
    clip/clip.py:126-136.
  * `jit=True` is rejected: the product path is the HIP plan, not a traced graph.
  * preprocessing is implemented with PIL + torch (torchvision is not required): bicubic
    resize of the short side, centre crop, RGB, [0,1] scaling, CLIP mean/std
    (clip/clip.py:79-86).
"""
import os
import warnings
from typing import List, Union

import numpy as np
import torch

from .model import build_model
from .simple_tokenizer import SimpleTokenizer as _Tokenizer

__all__ = ["available_models", "load", "tokenize"]

_MODEL_FILES = {
    "RN50": "RN50.pt", "RN101": "RN101.pt", "RN50x4": "RN50x4.pt", "RN50x16": "RN50x16.pt", "RN50x64": "RN50x64.pt",
    "ViT-B/32": "ViT-B-32.pt", "ViT-B/16": "ViT-B-16.pt", "ViT-L/14": "ViT-L-14.pt",
    "ViT-L/14@336px": "ViT-L-14-336px.pt",
}
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)

_tokenizer = None


def _get_tokenizer():
    global _tokenizer
    if _tokenizer is None:
        _tokenizer = _Tokenizer()
    return _tokenizer


def available_models() -> List[str]:
    return list(_MODEL_FILES.keys())


class _Preprocess:
    """PIL image -> float32 [3, n_px, n_px], normalised with the CLIP statistics."""

    def __init__(self, n_px):
        self.n_px = n_px

    def __call__(self, image):
        from PIL import Image
        n = self.n_px
        w, h = image.size
        if w <= h:
            nw, nh = n, int(n * h / w)
        else:
            nw, nh = int(n * w / h), n
        image = image.resize((nw, nh), Image.BICUBIC)
        left, top = int(round((nw - n) / 2.0)), int(round((nh - n) / 2.0))
        image = image.crop((left, top, left + n, top + n)).convert("RGB")
        x = torch.from_numpy(np.asarray(image, dtype=np.uint8).copy()).permute(2, 0, 1).float().div_(255.0)
        mean = torch.tensor(CLIP_MEAN).view(3, 1, 1)
        std = torch.tensor(CLIP_STD).view(3, 1, 1)
        return (x - mean) / std

    def __repr__(self):
        return f"ClipPreprocess(n_px={self.n_px})"


def _transform(n_px):
    return _Preprocess(n_px)


def _read_state_dict(path):
    try:
        sd = torch.load(path, map_location="cpu", weights_only=True)
        if isinstance(sd, dict) and "state_dict" in sd and "text_projection" not in sd:
            sd = sd["state_dict"]
        return sd
    except Exception:
        with open(path, "rb") as f:
            return torch.jit.load(f, map_location="cpu").eval().state_dict()


def load(name: str, device: Union[str, torch.device] = "cuda" if torch.cuda.is_available() else "cpu",
         jit: bool = False, download_root: str = None):
    """Returns (model, preprocess) like clip/clip.py:94."""
    if jit:
        raise RuntimeError("jit=True is not supported by the MI355X build (use the default jit=False)")
    if name in _MODEL_FILES:
        root = download_root or os.path.expanduser("~/.cache/clip")
        path = os.path.join(root, _MODEL_FILES[name])
        if not os.path.isfile(path):
            raise RuntimeError(f"Model {name}: {path} not found and this build never downloads "
                               f"(no network); place the checkpoint there or pass a file path")
    elif os.path.isfile(name):
        path = name
    else:
        raise RuntimeError(f"Model {name} not found; available models = {available_models()}")
    model = build_model(_read_state_dict(path)).to(device)
    if str(device) == "cpu":
        warnings.warn("dbmm_amd CLIP on device='cpu': parameters load, but encode_* needs an MI355X (no CPU path)")
    return model, _transform(model.visual.input_resolution)


def tokenize(texts: Union[str, List[str]], context_length: int = 77, truncate: bool = False) -> torch.IntTensor:
    """int32 [n, context_length] rows: <sot> ids <eot> then zero padding (clip/clip.py:197-237)."""
    if isinstance(texts, str):
        texts = [texts]
    tok = _get_tokenizer()
    sot, eot = tok.encoder["<|startoftext|>"], tok.encoder["<|endoftext|>"]
    result = torch.zeros(len(texts), context_length, dtype=torch.int)
    for i, text in enumerate(texts):
        ids = [sot] + tok.encode(text) + [eot]
        if len(ids) > context_length:
            if not truncate:
                raise RuntimeError(f"Input {texts[i]} is too long for context length {context_length}")
            ids = ids[:context_length]
            ids[-1] = eot
        result[i, :len(ids)] = torch.tensor(ids, dtype=torch.int)
    return result
