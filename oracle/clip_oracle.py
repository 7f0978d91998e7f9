"""ORACLE (test infrastructure, not product code) -- CPU restatement of the reference's
CLIP embedding path in plain torch-CPU fp32 math, written against explicit state dicts.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product path (debiasing-multi-modal_amd/) never does and fails loudly when
its HIP library is missing.

Pinned by: tests/golden/*.npz, which were produced by importing the reference's own
modules (/root/reference/clip/model.py) in the build container -- see
oracle/make_golden.py.  tests/test_oracle_golden.py re-checks this file against those
vectors on every run.

Each function cites the reference lines it restates (paths relative to /root/reference).
All tensors fp32, CPU path semantics of clip/clip.py:139-141 (`model.float()` on cpu).
"""
import math

import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------
# ModifiedResNet  (clip/model.py:94-154)
# ---------------------------------------------------------------------------------------

def _bn_eval(x, sd, p, eps=1e-5):
    """BatchNorm2d in eval mode (build_model returns .eval(), clip/model.py:436):
    y = (x - running_mean) / sqrt(running_var + eps) * weight + bias."""
    shape = (1, -1, 1, 1)
    inv = torch.rsqrt(sd[p + ".running_var"].reshape(shape) + eps)
    return (x - sd[p + ".running_mean"].reshape(shape)) * inv * sd[p + ".weight"].reshape(shape) \
        + sd[p + ".bias"].reshape(shape)


def _bottleneck(x, sd, p, stride):
    """Bottleneck.forward, clip/model.py:42-55.  All convs stride 1; avgpool(stride) after
    conv2 and in front of the downsample conv (clip/model.py:25,36-40)."""
    out = F.relu(_bn_eval(F.conv2d(x, sd[p + "conv1.weight"]), sd, p + "bn1"))
    out = F.relu(_bn_eval(F.conv2d(out, sd[p + "conv2.weight"], padding=1), sd, p + "bn2"))
    if stride > 1:
        out = F.avg_pool2d(out, stride)
    out = _bn_eval(F.conv2d(out, sd[p + "conv3.weight"]), sd, p + "bn3")
    identity = x
    if (p + "downsample.0.weight") in sd:
        identity = F.avg_pool2d(x, stride) if stride > 1 else x   # AvgPool2d(1) == identity
        identity = _bn_eval(F.conv2d(identity, sd[p + "downsample.0.weight"]), sd, p + "downsample.1")
    return F.relu(out + identity)


def attention_pool(x, sd, p, num_heads):
    """AttentionPool2d.forward, clip/model.py:68-91, as explicit math.

    tokens = [mean(x), x_0..x_{HW-1}] + pos; q from token 0 only; q scaled by hd^-0.5
    (F.multi_head_attention_forward); softmax over all HW+1 keys; c_proj."""
    B, C, H, W = x.shape
    t = x.flatten(2).permute(0, 2, 1)                                    # [B, HW, C]
    t = torch.cat([t.mean(dim=1, keepdim=True), t], dim=1)               # [B, HW+1, C]
    t = t + sd[p + "positional_embedding"][None]
    hd = C // num_heads
    q = F.linear(t[:, :1], sd[p + "q_proj.weight"], sd[p + "q_proj.bias"]) * (hd ** -0.5)
    k = F.linear(t, sd[p + "k_proj.weight"], sd[p + "k_proj.bias"])
    v = F.linear(t, sd[p + "v_proj.weight"], sd[p + "v_proj.bias"])
    L = t.shape[1]
    q = q.reshape(B, 1, num_heads, hd).transpose(1, 2)                   # [B, h, 1, hd]
    k = k.reshape(B, L, num_heads, hd).transpose(1, 2)
    v = v.reshape(B, L, num_heads, hd).transpose(1, 2)
    att = torch.softmax(q @ k.transpose(-1, -2), dim=-1)                 # [B, h, 1, L]
    o = (att @ v).transpose(1, 2).reshape(B, C)
    return F.linear(o, sd[p + "c_proj.weight"], sd[p + "c_proj.bias"])


def rn_layout(sd, prefix="visual."):
    """(layers, width) inferred from key shapes like build_model, clip/model.py:409-411."""
    layers = tuple(len({k.split(".")[2] for k in sd if k.startswith(f"{prefix}layer{b}.")}) for b in (1, 2, 3, 4))
    return layers, sd[prefix + "layer1.0.conv1.weight"].shape[0]


def rn_encode_image(sd, image, prefix="visual.", return_stages=False):
    """CLIP.encode_image -> ModifiedResNet.forward, clip/model.py:340-341,138-154."""
    p = prefix
    layers, width = rn_layout(sd, p)
    heads = width * 32 // 64                                             # clip/model.py:263
    x = image.float()
    x = F.relu(_bn_eval(F.conv2d(x, sd[p + "conv1.weight"], stride=2, padding=1), sd, p + "bn1"))
    x = F.relu(_bn_eval(F.conv2d(x, sd[p + "conv2.weight"], padding=1), sd, p + "bn2"))
    x = F.relu(_bn_eval(F.conv2d(x, sd[p + "conv3.weight"], padding=1), sd, p + "bn3"))
    x = F.avg_pool2d(x, 2)
    stages = {"stem": x}
    for li, nblocks in enumerate(layers, start=1):
        for bi in range(nblocks):
            stride = 2 if (li > 1 and bi == 0) else 1
            x = _bottleneck(x, sd, f"{p}layer{li}.{bi}.", stride)
        stages[f"layer{li}"] = x
    out = attention_pool(x, sd, p + "attnpool.", heads)
    return (out, stages) if return_stages else out


# ---------------------------------------------------------------------------------------
# Transformer blocks (clip/model.py:157-203), ViT (206-240), text tower (343-356)
# ---------------------------------------------------------------------------------------

def _layer_norm(x, sd, p, eps=1e-5):
    """LayerNorm.forward, clip/model.py:157-163 (fp32 statistics, biased variance)."""
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * sd[p + ".weight"] + sd[p + ".bias"]


def _mha(x, sd, p, heads, causal):
    """nn.MultiheadAttention(d, heads)(x, x, x, attn_mask) of clip/model.py:185-187 on
    batch-first [B, L, E]: packed in_proj, q scaled by hd^-0.5, additive -inf mask strictly
    above the diagonal when `causal` (clip/model.py:328-334), out_proj."""
    B, L, E = x.shape
    hd = E // heads
    qkv = F.linear(x, sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"])
    q, k, v = qkv.split(E, dim=-1)
    q = q.reshape(B, L, heads, hd).transpose(1, 2) * (hd ** -0.5)
    k = k.reshape(B, L, heads, hd).transpose(1, 2)
    v = v.reshape(B, L, heads, hd).transpose(1, 2)
    s = q @ k.transpose(-1, -2)
    if causal:
        s = s + torch.full((L, L), float("-inf")).triu_(1)
    o = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, L, E)
    return F.linear(o, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def _resblock(x, sd, p, heads, causal):
    """ResidualAttentionBlock.forward, clip/model.py:189-192; QuickGELU :166-168."""
    x = x + _mha(_layer_norm(x, sd, p + ".ln_1"), sd, p + ".attn", heads, causal)
    h = F.linear(_layer_norm(x, sd, p + ".ln_2"), sd[p + ".mlp.c_fc.weight"], sd[p + ".mlp.c_fc.bias"])
    h = h * torch.sigmoid(1.702 * h)
    return x + F.linear(h, sd[p + ".mlp.c_proj.weight"], sd[p + ".mlp.c_proj.bias"])


def _n_resblocks(sd, prefix):
    return len({k.split(".")[len(prefix.split(".")) + 1] for k in sd if k.startswith(prefix + ".resblocks.")})


def vit_encode_image(sd, image, prefix="visual."):
    """CLIP.encode_image -> VisionTransformer.forward, clip/model.py:223-240."""
    p = prefix
    w = sd[p + "conv1.weight"]
    width, patch = w.shape[0], w.shape[-1]
    heads = width // 64                                                   # clip/model.py:272
    x = F.conv2d(image.float(), w, stride=patch)                          # [B, width, g, g]
    x = x.reshape(x.shape[0], width, -1).permute(0, 2, 1)                 # [B, g*g, width]
    cls = sd[p + "class_embedding"].expand(x.shape[0], 1, width)
    x = torch.cat([cls, x], dim=1) + sd[p + "positional_embedding"]
    x = _layer_norm(x, sd, p + "ln_pre")
    for i in range(_n_resblocks(sd, p + "transformer")):
        x = _resblock(x, sd, f"{p}transformer.resblocks.{i}", heads, causal=False)
    x = _layer_norm(x[:, 0, :], sd, p + "ln_post")
    return x @ sd[p + "proj"]


def encode_text(sd, text):
    """CLIP.encode_text, clip/model.py:343-356: token gather, +pos, causal transformer,
    ln_final, row at argmax(text) (EOT = largest id), @ text_projection."""
    width = sd["ln_final.weight"].shape[0]
    heads = width // 64                                                   # clip/model.py:421
    x = sd["token_embedding.weight"][text.long()] + sd["positional_embedding"]
    for i in range(_n_resblocks(sd, "transformer")):
        x = _resblock(x, sd, f"transformer.resblocks.{i}", heads, causal=True)
    x = _layer_norm(x, sd, "ln_final")
    x = x[torch.arange(x.shape[0]), text.argmax(dim=-1)]
    return x @ sd["text_projection"]


def encode_image(sd, image):
    return vit_encode_image(sd, image) if "visual.proj" in sd else rn_encode_image(sd, image)


# ---------------------------------------------------------------------------------------
# zero-shot tail of clip_inference.py:207-216 and the minority flags :219-233
# ---------------------------------------------------------------------------------------

def zeroshot_tail(image_features, zeroshot_weights, temperature=0.02):
    """normalize(f) @ W / 0.02 -> softmax -> max.  Returns logits, probs, pred (int64)."""
    f = image_features / image_features.norm(dim=-1, keepdim=True)
    logits = f @ zeroshot_weights / temperature
    probs = logits.softmax(dim=-1)
    _, pred = torch.max(probs, dim=1)
    return logits, probs, pred


def minority_flags(dataset, target, target_s, pred):
    """clip_inference.py:219-233 -> (is_minor, is_minor_pred), int64."""
    if dataset == "waterbirds":
        is_minor_pred = (((target == 0) & (pred == 1)) | ((target == 1) & (pred == 0))).long()
        is_minor = (((target == 0) & (target_s == 1)) | ((target == 1) & (target_s == 0))).long()
    else:
        is_minor_pred = ((target == 1) & (pred == 1)).long()
        is_minor = ((target == 1) & (target_s == 1)).long()
    return is_minor, is_minor_pred
