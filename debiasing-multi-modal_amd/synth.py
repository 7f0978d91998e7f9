"""Deterministic synthetic weights / inputs (build-owned; no torch RNG involved).

Every tensor is a pure function of (seed, name, shape): a splitmix64 counter hash
-> 24-bit uniforms -> (for normals) an Irwin-Hall sum of 4 uniforms.  Only integer
ops and float adds/multiplies are used, so the values are bit-identical on any host
(no libm transcendental whose last bit could differ between machines).  Weights are
therefore *regenerated* on both sides of a parity test and never committed
(SURVEY.md section 8c / 8d).

State-dict layouts follow the OpenAI CLIP key names the reference consumes
(/root/reference/clip/model.py:399-436 infers the architecture from them).
"""
import zlib
from collections import OrderedDict

import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _splitmix64(x):
    x = (x + _GOLD) & _MASK
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
    return z ^ (z >> np.uint64(31))


def _stream(seed, name, n):
    """n uint64 hashes for tensor `name` under `seed`."""
    key = np.uint64(zlib.crc32(name.encode("utf-8"))) | (np.uint64(seed & 0xFFFFFFFF) << np.uint64(32))
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([key], dtype=np.uint64))[0]
        idx = np.arange(n, dtype=np.uint64)
        return _splitmix64(base + idx * _GOLD)


def uniform(seed, name, shape, lo=0.0, hi=1.0):
    n = int(np.prod(shape)) if len(shape) else 1
    u = (_stream(seed, name, n) >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32).reshape(shape))


def normal(seed, name, shape, std=1.0, mean=0.0):
    """Approximate N(mean, std^2): Irwin-Hall(4), exact in float64 adds."""
    n = int(np.prod(shape)) if len(shape) else 1
    h = _stream(seed, name, n)
    m16 = np.uint64(0xFFFF)
    s = ((h & m16).astype(np.float64) + ((h >> np.uint64(16)) & m16).astype(np.float64)
         + ((h >> np.uint64(32)) & m16).astype(np.float64) + ((h >> np.uint64(48)) & m16).astype(np.float64))
    s = (s * (1.0 / 65536.0) - 2.0) * 1.7320508075688772  # var(sum of 4 U(0,1)) = 1/3
    return torch.from_numpy((mean + std * s).astype(np.float32).reshape(shape))


def integers(seed, name, shape, high):
    n = int(np.prod(shape)) if len(shape) else 1
    v = (_stream(seed, name, n) >> np.uint64(33)) % np.uint64(high)
    return torch.from_numpy(v.astype(np.int64).reshape(shape))


# --------------------------------------------------------------------------------------
# CLIP state dicts (OpenAI key names)
# --------------------------------------------------------------------------------------

def _bn(sd, seed, prefix, c, gamma=(0.5, 1.5)):
    sd[prefix + ".weight"] = uniform(seed, prefix + ".weight", (c,), *gamma)
    sd[prefix + ".bias"] = normal(seed, prefix + ".bias", (c,), 0.1)
    sd[prefix + ".running_mean"] = normal(seed, prefix + ".running_mean", (c,), 0.1)
    sd[prefix + ".running_var"] = uniform(seed, prefix + ".running_var", (c,), 0.5, 1.5)
    sd[prefix + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.int64)


def _conv(sd, seed, name, cout, cin, k):
    std = (2.0 / (cin * k * k)) ** 0.5  # He init keeps activations O(1) through 55 convs
    sd[name] = normal(seed, name, (cout, cin, k, k), std)


def _linear(sd, seed, prefix, out_f, in_f, std=None, bias_std=0.02):
    sd[prefix + ".weight"] = normal(seed, prefix + ".weight", (out_f, in_f), std or in_f ** -0.5)
    sd[prefix + ".bias"] = normal(seed, prefix + ".bias", (out_f,), bias_std)


def _ln(sd, seed, prefix, c):
    sd[prefix + ".weight"] = uniform(seed, prefix + ".weight", (c,), 0.5, 1.5)
    sd[prefix + ".bias"] = normal(seed, prefix + ".bias", (c,), 0.1)


def _resblocks(sd, seed, prefix, width, layers):
    proj_std = (width ** -0.5) * ((2 * layers) ** -0.5)
    for i in range(layers):
        p = f"{prefix}.resblocks.{i}"
        sd[p + ".attn.in_proj_weight"] = normal(seed, p + ".attn.in_proj_weight", (3 * width, width), width ** -0.5)
        sd[p + ".attn.in_proj_bias"] = normal(seed, p + ".attn.in_proj_bias", (3 * width,), 0.02)
        _linear(sd, seed, p + ".attn.out_proj", width, width, proj_std)
        _ln(sd, seed, p + ".ln_1", width)
        _linear(sd, seed, p + ".mlp.c_fc", 4 * width, width, (2 * width) ** -0.5)
        _linear(sd, seed, p + ".mlp.c_proj", width, 4 * width, proj_std)
        _ln(sd, seed, p + ".ln_2", width)


def modified_resnet_state_dict(seed, layers=(3, 4, 6, 3), width=64, output_dim=1024,
                               input_resolution=224, prefix="visual."):
    """ModifiedResNet weights with *non-trivial* BN stats and bn3.weight != 0.

    The reference zero-inits every bn3.weight (clip/model.py:311-314) which would
    multiply 2/3 of the network by zero; fixtures must not do that.
    """
    sd = OrderedDict()
    p = prefix
    _conv(sd, seed, p + "conv1.weight", width // 2, 3, 3); _bn(sd, seed, p + "bn1", width // 2)
    _conv(sd, seed, p + "conv2.weight", width // 2, width // 2, 3); _bn(sd, seed, p + "bn2", width // 2)
    _conv(sd, seed, p + "conv3.weight", width, width // 2, 3); _bn(sd, seed, p + "bn3", width)
    inplanes = width
    for li, (nblocks, planes_mul) in enumerate(zip(layers, (1, 2, 4, 8)), start=1):
        planes = width * planes_mul
        for bi in range(nblocks):
            stride = 2 if (li > 1 and bi == 0) else 1
            q = f"{p}layer{li}.{bi}."
            _conv(sd, seed, q + "conv1.weight", planes, inplanes, 1); _bn(sd, seed, q + "bn1", planes)
            _conv(sd, seed, q + "conv2.weight", planes, planes, 3); _bn(sd, seed, q + "bn2", planes)
            _conv(sd, seed, q + "conv3.weight", planes * 4, planes, 1)
            _bn(sd, seed, q + "bn3", planes * 4, gamma=(0.2, 0.6))
            if stride > 1 or inplanes != planes * 4:
                _conv(sd, seed, q + "downsample.0.weight", planes * 4, inplanes, 1)
                _bn(sd, seed, q + "downsample.1", planes * 4, gamma=(0.4, 0.9))
            inplanes = planes * 4
    e = width * 32
    sp = input_resolution // 32
    sd[p + "attnpool.positional_embedding"] = normal(seed, p + "attnpool.positional_embedding", (sp * sp + 1, e), e ** -0.5)
    for nm, o in (("k_proj", e), ("q_proj", e), ("v_proj", e), ("c_proj", output_dim)):
        _linear(sd, seed, p + "attnpool." + nm, o, e, e ** -0.5)
    return sd


def vision_transformer_state_dict(seed, input_resolution=224, patch_size=32, width=768, layers=12,
                                  output_dim=512, prefix="visual."):
    sd = OrderedDict()
    p = prefix
    sd[p + "conv1.weight"] = normal(seed, p + "conv1.weight", (width, 3, patch_size, patch_size),
                                    (3 * patch_size * patch_size) ** -0.5)
    scale = width ** -0.5
    sd[p + "class_embedding"] = normal(seed, p + "class_embedding", (width,), scale)
    g = input_resolution // patch_size
    sd[p + "positional_embedding"] = normal(seed, p + "positional_embedding", (g * g + 1, width), scale)
    _ln(sd, seed, p + "ln_pre", width)
    _resblocks(sd, seed, p + "transformer", width, layers)
    _ln(sd, seed, p + "ln_post", width)
    sd[p + "proj"] = normal(seed, p + "proj", (width, output_dim), scale)
    return sd


def text_state_dict(seed, embed_dim=1024, context_length=77, vocab_size=49408, width=512, layers=12):
    sd = OrderedDict()
    sd["positional_embedding"] = normal(seed, "positional_embedding", (context_length, width), 0.01)
    sd["text_projection"] = normal(seed, "text_projection", (width, embed_dim), width ** -0.5)
    sd["logit_scale"] = torch.tensor(float(np.log(1 / 0.07)), dtype=torch.float32)
    _resblocks(sd, seed, "transformer", width, layers)
    sd["token_embedding.weight"] = normal(seed, "token_embedding.weight", (vocab_size, width), 0.02)
    _ln(sd, seed, "ln_final", width)
    return sd


def clip_state_dict(seed, arch="RN50", **over):
    """Full CLIP state dict for a named architecture (random init, synthetic)."""
    cfgs = {
        "RN50": dict(kind="rn", layers=(3, 4, 6, 3), width=64, embed=1024, res=224, twidth=512, tlayers=12),
        "RN101": dict(kind="rn", layers=(3, 4, 23, 3), width=64, embed=512, res=224, twidth=512, tlayers=12),
        "RN50x4": dict(kind="rn", layers=(4, 6, 10, 6), width=80, embed=640, res=288, twidth=640, tlayers=12),
        "ViT-B/32": dict(kind="vit", patch=32, width=768, layers=12, embed=512, res=224, twidth=512, tlayers=12),
        "ViT-L/14@336px": dict(kind="vit", patch=14, width=1024, layers=24, embed=768, res=336, twidth=768, tlayers=12),
        # tiny configs used by the full-tensor parity fixtures
        "tiny-RN": dict(kind="rn", layers=(1, 1, 1, 1), width=64, embed=128, res=64, twidth=64, tlayers=2, vocab=512),
        "tiny-RN-w32": dict(kind="rn", layers=(1, 2, 1, 1), width=32, embed=64, res=96, twidth=64, tlayers=2, vocab=512),
        "tiny-ViT": dict(kind="vit", patch=16, width=128, layers=2, embed=64, res=64, twidth=64, tlayers=2, vocab=512),
        # ViT-L/14@336px geometry (577 tokens, 16 heads, patch K = 588) with 2 layers: exercises the
        # multi-tile attention core and the K-tail GEMM path without the 24-layer cost
        "ViT-L14-336-2L": dict(kind="vit", patch=14, width=1024, layers=2, embed=768, res=336, twidth=64, tlayers=1,
                               vocab=512),
    }
    c = dict(cfgs[arch]); c.update(over)
    if c["kind"] == "rn":
        sd = modified_resnet_state_dict(seed, c["layers"], c["width"], c["embed"], c["res"])
    else:
        sd = vision_transformer_state_dict(seed, c["res"], c["patch"], c["width"], c["layers"], c["embed"])
    sd.update(text_state_dict(seed, c["embed"], 77, c.get("vocab", 49408), c["twidth"], c["tlayers"]))
    # The official checkpoints store fp16 values and the reference's build_model round-trips
    # conv/linear/attention/projection weights through fp16 (clip/model.py:375-396,434-435):
    # make every value fp16-representable so that round trip is lossless, as with real files.
    for k, v in sd.items():
        if v.dtype == torch.float32 and v.dim() > 0:
            sd[k] = v.half().float()
    return sd


def images(seed, batch, res):
    return normal(seed, "images", (batch, 3, res, res))


# --------------------------------------------------------------------------------------
# adapter-side synthetic data
# --------------------------------------------------------------------------------------

def adapter_state_dict(seed, input_dim=1024, hidden_dim=128, prefix="layers."):
    """Keys of reference Adapter.state_dict() (final_main.py:160-174)."""
    sd = OrderedDict()
    b1 = input_dim ** -0.5
    sd[prefix + "0.weight"] = uniform(seed, prefix + "0.weight", (hidden_dim, input_dim), -b1, b1)
    sd[prefix + "0.bias"] = uniform(seed, prefix + "0.bias", (hidden_dim,), -b1, b1)
    sd[prefix + "1.weight"] = uniform(seed, prefix + "1.weight", (hidden_dim,), 0.8, 1.2)
    sd[prefix + "1.bias"] = normal(seed, prefix + "1.bias", (hidden_dim,), 0.05)
    sd[prefix + "1.running_mean"] = torch.zeros(hidden_dim)
    sd[prefix + "1.running_var"] = torch.ones(hidden_dim)
    sd[prefix + "1.num_batches_tracked"] = torch.tensor(0, dtype=torch.int64)
    b2 = hidden_dim ** -0.5
    sd[prefix + "3.weight"] = uniform(seed, prefix + "3.weight", (input_dim, hidden_dim), -b2, b2)
    sd[prefix + "3.bias"] = uniform(seed, prefix + "3.bias", (input_dim,), -b2, b2)
    return sd


def labels(seed, batch):
    """y ~ Bernoulli(.25); confounder = y w.p. .95 (Waterbirds-like); g = 2y+c."""
    y = (uniform(seed, "y", (batch,)) < 0.25).long()
    flip = (uniform(seed, "flip", (batch,)) < 0.05).long()
    c = (y ^ flip).long()
    return y, c, 2 * y + c


def text_matrix(seed, dim, n_cls, name="text"):
    """[D, C] text feature matrix (column = prompt), un-normalised like the JSON files."""
    return normal(seed, name, (dim, n_cls))


def embedding_dataset(seed, split, n, dim=1024, p_y=0.25, p_agree=0.85, noise=0.5, s_class=0.025, s_spur=0.045):
    """(embeddings [n, dim] fp32, y [n], confounder [n]) of one split of a synthetic Waterbirds / CelebA-like embedding set: a class
    direction and a STRONGER spurious direction (shared by all splits of a seed) under isotropic noise, y ~ Bernoulli(p_y), the
    confounder agrees with y w.p. p_agree -- so that ERM leans on the spurious direction and the minority groups are the hard ones."""
    u_y, u_c = normal(seed, "dir_class", (dim,)), normal(seed, "dir_spur", (dim,))
    y = (uniform(seed, split + "/y", (n,)) < p_y).long()
    c = (y ^ (uniform(seed, split + "/flip", (n,)) < 1.0 - p_agree).long()).long()
    x = noise * normal(seed, split + "/noise", (n, dim)) + 0.1 * normal(seed, "common", (1, dim))
    x = x + s_class * (2 * y.float().unsqueeze(1) - 1) * u_y + s_spur * (2 * c.float().unsqueeze(1) - 1) * u_c
    return x.contiguous(), y, c


def embedding_text(seed, dim=1024, spread=0.6):
    """prompt matrices for embedding_dataset: (class [dim, 2], spurious [dim, 2], group [dim, 4]); a column = common part + its
    attribute's direction(s) + its own noise"""
    u_y, u_c = normal(seed, "dir_class", (dim,)), normal(seed, "dir_spur", (dim,))
    base = normal(seed, "text_common", (dim,))
    col = lambda nm, sy, sc: base + spread * (sy * u_y + sc * u_c) + 0.3 * normal(seed, "text/" + nm, (dim,))
    tcls = torch.stack([col("c0", -1, 0), col("c1", 1, 0)], dim=1)
    tspu = torch.stack([col("s0", 0, -1), col("s1", 0, 1)], dim=1)
    tgrp = torch.stack([col(f"g{g}", 2 * (g // 2) - 1, 2 * (g % 2) - 1) for g in range(4)], dim=1)
    return tcls.contiguous(), tspu.contiguous(), tgrp.contiguous()
