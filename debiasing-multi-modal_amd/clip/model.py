"""CLIP model surface of the reference (`/root/reference/clip/model.py`) over MI355X kernels.

What is kept: the OpenAI state-dict key names, `build_model`
inferring the architecture from key shapes (clip/model.py:399-436), `CLIP.encode_image`,
`CLIP.encode_text`, `CLIP.forward`, `.dtype`, `.visual.input_resolution`, `.logit_scale`
(clip/model.py:336-372), `convert_weights`, and the nn.Module protocol (`.cuda()`, `.eval()`,
`state_dict()`).

What is different: modules here are *parameter holders*; there is no per-layer forward.
Each tower compiles an execution plan on first use and `forward` walks that plan calling the C ABI
(ops.py).  Two arithmetic modes:

  * fp32 parity mode (default; `model.dtype == float32`, what `build_model` returns): activations are fp32
    NHWC / batch-first tokens in HBM.  Convolutions and GEMMs run on the 16-bit matrix cores with every
    activation split into an fp16 (hi, lo) pair under an exact per-tensor power-of-two scale and the
    checkpoint's fp16-exact weight as ONE fp16 plane (two partial products, fp32 accumulate): fp32-level
    accuracy, which the 1e-3 logit parity against the reference's CPU path needs.  Eval-mode BatchNorm is a
    per-channel scale / bias in the conv epilogue; ReLU, residual adds and the 2x2 average pools are fused
    there too; in layers 1-2 conv3 + residual and the next block's conv1 are one launch.
    set_plan_option("conv_split", "bf16" / "off") selects the bf16-triple / fp32-input-MFMA variants of the same arithmetic.
  * fp16 throughput mode (`convert_weights(model)` or `model.half()`, the reference's GPU path,
    clip/model.py:375-396): every tower keeps fp16 activations in HBM, one fp16 MFMA per product, fp32 accumulation,
    fp32 LayerNorm / softmax statistics / BatchNorm arithmetic (csrc/f16_ops.hip, csrc/conv_f16.hip).  RN towers whose
    width is not a multiple of 64 (RN50x4, RN50x16, toy towers) compute fp32-accurately in this mode and round their
    output to fp16.
"""
import os
import re
from collections import OrderedDict
from typing import Tuple, Union

import numpy as np
import torch
from torch import nn

from .. import ops


# ---------------------------------------------------------------------------------------
# parameter holders (names = the reference's state-dict keys)
# ---------------------------------------------------------------------------------------

class _ConvW(nn.Module):
    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))


class _BNParams(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.eps = 1e-5


class _LinearW(nn.Module):
    def __init__(self, fin, fout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(fout, fin))
        self.bias = nn.Parameter(torch.zeros(fout))
        self.in_features, self.out_features = fin, fout


class _LNParams(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.eps = 1e-5


class _AttnParams(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = _LinearW(d, d)


class _Downsample(nn.Module):
    """keys downsample.0.weight / downsample.1.* (clip/model.py:36-40)."""
    def __init__(self, cin, cout):
        super().__init__()
        self.add_module("0", _ConvW(cin, cout, 1))
        self.add_module("1", _BNParams(cout))


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1, self.bn1 = _ConvW(inplanes, planes, 1), _BNParams(planes)
        self.conv2, self.bn2 = _ConvW(planes, planes, 3), _BNParams(planes)
        self.conv3, self.bn3 = _ConvW(planes, planes * 4, 1), _BNParams(planes * 4)
        self.stride = stride
        self.downsample = _Downsample(inplanes, planes * 4) if (stride > 1 or inplanes != planes * 4) else None


class AttentionPool2d(nn.Module):
    def __init__(self, spacial_dim, embed_dim, num_heads, output_dim=None):
        super().__init__()
        self.positional_embedding = nn.Parameter(torch.randn(spacial_dim ** 2 + 1, embed_dim) / embed_dim ** 0.5)
        self.k_proj = _LinearW(embed_dim, embed_dim)
        self.q_proj = _LinearW(embed_dim, embed_dim)
        self.v_proj = _LinearW(embed_dim, embed_dim)
        self.c_proj = _LinearW(embed_dim, output_dim or embed_dim)
        self.num_heads = num_heads


def _fold_bn(conv_w, bn):
    """conv (bias-free) followed by eval-mode BN == conv with w*s and bias beta - mu*s."""
    s = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
    w = conv_w.double() * s.reshape(-1, 1, 1, 1)
    b = bn.bias.double() - bn.running_mean.double() * s
    return w, b


def _conv_bn(conv, bn):
    """operands of one conv + eval-mode BatchNorm (see _pack_conv)"""
    w, b = _fold_bn(conv.weight, bn)
    scale = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
    return _pack_conv(w, b, raw=conv.weight.detach().float(), scale=scale)


# Plan options: how a tower's execution plan is built.  ONE table, set by name (`set_plan_option`), each entry seeded once at
# import from DBMM_<NAME> like the library's own option table (csrc/options.hip) so that developer A/B runs still work from the
# shell; nothing below reads the environment.  All settings are fp32-accurate; they exist for ablation runs.
#   conv_split  how operands are split for the matrix cores: "f16" = fp16 pair + per-tensor power-of-two scales (two / three
#               partial products), "bf16" = bf16 triple (six), "off" = fp32-input MFMA only
#   conv_k_order  K order of the packed KxK conv weights: "chunk32" = (cin/32, kh, kw, 32), the taps of a 32-channel slab are
#               consecutive K chunks so the KH*KW re-reads of an input pixel are L2 hits (+1.4 ... +9.8 % per 3x3 layer, HBM-side
#               fetch -75 %; pack_conv_weight falls back to tap-major when Cin % 32 != 0); "tap" = tap-major
#   fuse_ds     conv3 + downsample branch of a stage's first block as one dual-source GEMM (0: two launches)
#   fuse_chain  conv3 + residual chained with the next block's conv1 in one launch (0: separate launches)
#   fuse_conv2  ... and the block's own 3x3 conv2 inside that launch too (0: conv2 as its own launch)
#   seam_pool   the chain launch at a stage seam also writes AvgPool2d(2) of the stage output for the next stage's downsample branch:
#               2 (default) = and ONLY that, not the un-pooled tensor nobody else reads (the stage outputs of return_stages=True are
#               always written); 1 = both; 0 = fp16 mode: neither, a separate pooling launch (the fp32-accurate plan: as 1)
_PLAN_OPTIONS = {"conv_split": ("f16", ("f16", "bf16", "off")), "conv_k_order": ("chunk32", ("chunk32", "tap")),
                 "fuse_ds": (1, (0, 1)), "fuse_chain": (1, (0, 1)), "fuse_conv2": (1, (0, 1)), "seam_pool": (2, (0, 1, 2))}
_opt = {}
_opt_version = [0]


def set_plan_option(name, value):
    """set a plan option by name; returns the previous value.  Plans compiled under other settings are rebuilt on next use."""
    default, allowed = _PLAN_OPTIONS[name]
    if isinstance(default, int):
        value = int(value)
    if value not in allowed:
        raise ValueError(f"plan option {name}: {value!r} not in {allowed}")
    old = _opt.get(name, default)
    _opt[name] = value
    _opt_version[0] += 1
    return old


def plan_option(name):
    return _opt[name]


for _name in _PLAN_OPTIONS:
    set_plan_option(_name, os.environ.get("DBMM_" + _name.upper(), _PLAN_OPTIONS[_name][0]))


def _k_order():
    return False if _opt["conv_k_order"] == "tap" else 32


def _pack_conv(w64, bias, raw=None, scale=None):
    """[Cout][Cin][kh][kw] fp64 (BN folded) -> dict of ops.conv_bn_act operands: packed fp32 weight,
    layout id, bias, and the pre-split planes for the split-precision kernels (shapes those do not
    cover fall back to the fp32-MFMA kernel inside the library).
    raw / scale: the conv weight as stored and the BatchNorm scale gamma / sqrt(var + eps).  The
    reference keeps conv weights in fp16 (build_model, clip/model.py:433), so `raw` times a power
    of two is exact in fp16: the fp16-pair kernel then needs ONE weight plane and two partial
    products, with the BatchNorm scale applied to the accumulator in the epilogue (`sc`) instead
    of being folded into (and de-fp16-ing) the weights."""
    if raw is not None and _opt["conv_split"] == "f16" and raw.is_cuda:
        w, wl = ops.pack_conv_weight(raw, chunk_major=_k_order())
        K, cin = w.shape[1], raw.shape[1]
        if K % 32 == 0 and (raw.shape[2] * raw.shape[3] == 1 or cin % 32 == 0):
            ph, we, n = ops.split_planes_f16(w, allow_single=True)
            if n == 1:
                return dict(w=w, wl=wl, b=bias.float().contiguous(), p3=None, ph=ph, we=we,
                            sc=scale.float().contiguous())
    w, wl = ops.pack_conv_weight(w64, chunk_major=_k_order() if _opt["conv_split"] == "f16" else False)
    c = dict(w=w, wl=wl, b=bias.float().contiguous(), p3=None, ph=None, we=0, sc=None)
    if w.is_cuda and w.shape[1] % 16 == 0 and (w.shape[0] > 32 or _opt["conv_split"] == "f16"):
        if _opt["conv_split"] == "f16":
            c["ph"], c["we"], _ = ops.split_planes_f16(w)
        elif _opt["conv_split"] == "bf16":
            c["p3"] = ops.split_planes(w)
    return c


class ModifiedResNet(nn.Module):
    """Parameter layout of clip/model.py:94-136; forward = compiled plan over HIP kernels."""

    def __init__(self, layers, output_dim, heads, input_resolution=224, width=64):
        super().__init__()
        self.output_dim, self.input_resolution = output_dim, input_resolution
        self.conv1, self.bn1 = _ConvW(3, width // 2, 3), _BNParams(width // 2)
        self.conv2, self.bn2 = _ConvW(width // 2, width // 2, 3), _BNParams(width // 2)
        self.conv3, self.bn3 = _ConvW(width // 2, width, 3), _BNParams(width)
        inplanes = width
        for li, (n, mul) in enumerate(zip(layers, (1, 2, 4, 8)), start=1):
            blocks = []
            for bi in range(n):
                blocks.append(Bottleneck(inplanes, width * mul, 2 if (li > 1 and bi == 0) else 1))
                inplanes = width * mul * 4
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
        self.attnpool = AttentionPool2d(input_resolution // 32, width * 32, heads, output_dim)
        self._plan = None
        self._plan16 = None

    def refresh_plan(self):
        """Re-fold the weights (call after loading new parameters)."""
        self._plan = self._plan16 = None

    def _apply(self, fn, *a, **k):
        self._plan = self._plan16 = None
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):        # load_state_dict(): the folded / packed plan is stale
        self._plan = self._plan16 = None
        return super()._load_from_state_dict(*a, **k)

    def _param_key(self):
        """changes whenever a parameter or buffer is replaced or edited in place (optimizer step, .copy_())"""
        return (_opt_version[0],) + tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))

    @torch.no_grad()
    def _compile(self):
        P = {}
        w, b = _fold_bn(self.conv1.weight.float(), self.bn1)
        P["stem1"] = (w.permute(2, 3, 1, 0).contiguous().float(), b.float().contiguous())   # [kh][kw][cin][cout]
        for i in (2, 3):
            P[f"stem{i}"] = _conv_bn(getattr(self, f"conv{i}"), getattr(self, f"bn{i}"))
        blocks = []
        for li in (1, 2, 3, 4):
            for blk in getattr(self, f"layer{li}"):
                e = {"stride": blk.stride}
                for i in (1, 2, 3):
                    e[f"c{i}"] = _conv_bn(getattr(blk, f"conv{i}"), getattr(blk, f"bn{i}"))
                if blk.downsample is not None:
                    e["ds"] = _conv_bn(getattr(blk.downsample, "0"), getattr(blk.downsample, "1"))
                    c3, ds = e["c3"], e["ds"]
                    # conv3 + downsample branch as one dual-source GEMM: needs both weights as single exact
                    # fp16 planes and a non-zero bn3 scale (the accumulators are rescaled by scale_d / scale_3)
                    if (c3["sc"] is not None and ds["sc"] is not None and c3["ph"] is not None and ds["ph"] is not None
                            and c3["ph"].shape[0] == 1 and ds["ph"].shape[0] == 1
                            and float(c3["sc"].abs().min()) > 1e-20     # the rescale divides by bn3's scale ...
                            and float(ds["sc"].abs().max()) < 1e12 * float(c3["sc"].abs().min())):   # ... and must stay finite
                        ratio = ds["sc"].double() / c3["sc"].double() * 2.0 ** (c3["we"] - ds["we"])
                        e["dual"] = dict(ratio=ratio.float().contiguous(), bias=(c3["b"].double() + ds["b"].double()).float())
                blocks.append(e)
        P["blocks"] = blocks
        ap = self.attnpool
        P["attn"] = dict(
            pos=ap.positional_embedding.detach().float().contiguous(),
            wq=ap.q_proj.weight.detach().float().contiguous(), bq=ap.q_proj.bias.detach().float().contiguous(),
            wkv=torch.cat([ap.k_proj.weight, ap.v_proj.weight], 0).detach().float().contiguous(),
            bkv=torch.cat([ap.k_proj.bias, ap.v_proj.bias], 0).detach().float().contiguous(),
            wc=ap.c_proj.weight.detach().float().contiguous(), bc=ap.c_proj.bias.detach().float().contiguous())
        self._plan = P
        self._plan_key = self._param_key()
        return P

    def _f16_eligible(self):
        """the fp16 kernels' shapes: every 1x1 conv reduces over a multiple of 64 channels, the stem convs over 32, and every map
        that is average-pooled has even sides (input resolution a multiple of 32: the pooled 3x3 kernels and avgpool2_f16 take
        even maps only, where the reference's AvgPool2d(2) floors); other towers keep the fp32-accurate plan"""
        return (self.conv2.weight.shape[1] % 32 == 0 and self.conv3.weight.shape[0] % 64 == 0 and self.conv1.weight.shape[0] in (32, 64)
                and self.input_resolution % 32 == 0)

    @torch.no_grad()
    def _compile_f16(self):
        """fp16 mode (clip/model.py:146, 375-396): conv weights as the checkpoint's fp16 values ([Cout][Cin] for 1x1,
        [Cout][(cin/32, kh, kw, 32)] for 3x3), eval-mode BatchNorm as fp32 per-channel scale / bias for the conv epilogues;
        the 3-channel stem conv likewise (fp32 container of the fp16 values, [kh][kw][cin][cout]: its MFMA gather kernel packs them)."""
        def bn_sb(bn):
            sc = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
            return sc.float().contiguous(), (bn.bias.detach().double() - bn.running_mean.detach().double() * sc).float().contiguous()

        def c1x1(conv, bn):
            w = conv.weight.detach()
            return (w.reshape(w.shape[0], w.shape[1]).to(torch.float16).contiguous(),) + bn_sb(bn)

        def c3x3(conv, bn):
            w, _ = ops.pack_conv_weight(conv.weight.detach().float(), chunk_major=32)
            return (w.to(torch.float16).contiguous(),) + bn_sb(bn)
        P = {}
        sc1, b1 = bn_sb(self.bn1)
        P["stem1"] = (self.conv1.weight.detach().float().permute(2, 3, 1, 0).contiguous(), b1, sc1)      # (w, bias, scale)
        P["stem2"], P["stem3"] = c3x3(self.conv2, self.bn2), c3x3(self.conv3, self.bn3)
        blocks = []
        for li in (1, 2, 3, 4):
            for blk in getattr(self, f"layer{li}"):
                e = {"stride": blk.stride, "c1": c1x1(blk.conv1, blk.bn1), "c2": c3x3(blk.conv2, blk.bn2), "c3": c1x1(blk.conv3, blk.bn3)}
                if blk.downsample is not None:
                    e["ds"] = c1x1(getattr(blk.downsample, "0"), getattr(blk.downsample, "1"))
                    s3, sd = e["c3"][1], e["ds"][1]
                    if float(s3.abs().min()) > 1e-20:                       # conv3 + downsample as one dual-source GEMM (ratio of the BN scales)
                        e["dual"] = ((sd / s3).contiguous(), (e["c3"][2] + e["ds"][2]).contiguous())
                blocks.append(e)
        P["blocks"] = blocks
        ap = self.attnpool
        f = lambda t: t.detach().float().contiguous()
        P["attn"] = dict(pos=f(ap.positional_embedding), wq=f(ap.q_proj.weight), bq=f(ap.q_proj.bias),
                         wkv=f(torch.cat([ap.k_proj.weight, ap.v_proj.weight], 0)), bkv=f(torch.cat([ap.k_proj.bias, ap.v_proj.bias], 0)),
                         wc=f(ap.c_proj.weight), bc=f(ap.c_proj.bias))
        self._plan16 = (self._param_key(), P)
        return P

    @torch.no_grad()
    def forward_f16(self, x, return_stages=False):
        """fp16 mode: fp16 NHWC activations in HBM, one fp16 MFMA per product, BatchNorm / ReLU / residual / average pool in the
        fp32 epilogues (csrc/conv_f16.hip, csrc/f16_ops.hip).  Bottleneck.forward (clip/model.py:42-55) block by block."""
        if getattr(self, "_plan16", None) is None or self._plan16[0] != self._param_key():
            self._compile_f16()
        P = self._plan16[1]
        x = x.contiguous()
        if x.dtype not in (torch.float16, torch.float32):
            x = x.float()
        if x.shape[2] % 32 or x.shape[3] % 32:
            raise ops.DbmmUnsupported(f"fp16 mode of the ModifiedResNet tower needs image sides that are multiples of 32, got {tuple(x.shape)}")
        x = ops.conv_stem_s2_f16(x, *P["stem1"])                       # NCHW image -> fp16 NHWC
        x = ops.conv3x3_f16(x, *P["stem2"])
        x = ops.conv3x3_f16(x, *P["stem3"], pool=2)                     # + the stem's AvgPool2d(2)
        stages = {"stem": x}
        bi = 0
        y1_next = None                                                  # conv1 of this block, when the previous block's chain launch made it
        x_pooled = None                                                 # AvgPool2d(2) of x, when that launch made it too
        for li in (1, 2, 3, 4):
            nblk = len(getattr(self, f"layer{li}"))
            for k_in_stage in range(nblk):
                e = P["blocks"][bi]; bi += 1
                out = y1_next if y1_next is not None else ops.conv1x1_f16(x, *e["c1"])
                y1_next = None
                out = ops.conv3x3_f16(out, *e["c2"], pool=2 if e["stride"] == 2 else 1)     # conv2 + bn2 + relu (+ avgpool)
                if "ds" not in e and bi < len(P["blocks"]) and _opt["fuse_chain"]:
                    # conv3 + residual -> conv1 of the NEXT block (of this stage, or the first block of the next stage: its conv1 runs on
                    # the un-pooled map too) in one launch (layers 1 - 2 and the 1 -> 2 seam; None elsewhere)
                    nxt = P["blocks"][bi]
                    # ... and AvgPool2d(2) of x for the next stage's downsample branch; the un-pooled x only when the caller wants the stages
                    seam = "ds" in nxt and nxt["stride"] == 2 and _opt["seam_pool"] > 0
                    r = ops.chain_f16(out, e["c3"], x, nxt["c1"], pooled=seam, keep_full=return_stages or _opt["seam_pool"] < 2)
                    if r is not None:
                        x, x_pooled, y1_next = r if seam else (r[0], None, r[1])
                        continue
                identity = x
                if "ds" in e:
                    if e["stride"] == 2:
                        identity = x_pooled if x_pooled is not None else ops.avgpool2_f16(x)
                        x_pooled = None
                    fused = None
                    if "dual" in e and _opt["fuse_ds"] and _opt["fuse_chain"] and bi < len(P["blocks"]):
                        # ... and conv1 of the next block chained on (layer 1's first block; None elsewhere)
                        r = ops.chain_dual_f16(out, e["c3"][0], e["c3"][1], identity, e["ds"][0], *e["dual"], P["blocks"][bi]["c1"])
                        if r is not None:
                            x, y1_next = r
                            continue
                    if "dual" in e and _opt["fuse_ds"]:                     # conv3 + downsample branch in one launch where the library has the shape
                        fused = ops.conv1x1_dual_f16(out, e["c3"][0], e["c3"][1], identity, e["ds"][0], *e["dual"])
                    if fused is not None:
                        x = fused
                        continue
                    identity = ops.conv1x1_f16(identity, *e["ds"], act=ops.ACT_NONE)
                if bi < len(P["blocks"]) and "ds" in P["blocks"][bi] and P["blocks"][bi]["stride"] == 2 and _opt["seam_pool"] > 0:
                    r = ops.conv1x1_res_pool_f16(out, e["c3"], identity)         # a seam without a chain kernel (layer 3 -> 4): the same launch
                    if r is not None:                                           # also writes AvgPool2d(2) of x for the downsample branch
                        x, x_pooled = r
                        continue
                x = ops.conv1x1_f16(out, *e["c3"], residual=identity)   # bn3(conv3) + identity, ReLU
            stages[f"layer{li}"] = x
        a = P["attn"]
        # attention pool: the fp32-accurate kernels on the fp16 feature map (0.2 MB per image, read as fp16), result rounded to fp16
        out = ops.attnpool(x, a["pos"], a["wq"], a["bq"], a["wkv"], a["bkv"], a["wc"], a["bc"], self.attnpool.num_heads)
        out = out.to(torch.float16)
        return (out, stages) if return_stages else out

    def rounds_fp32_images_itself(self):
        """fp16 mode on the fp16 kernels: the stem conv reads an fp32 image and rounds it to fp16 as it gathers (the values of the
        reference's `image.type(self.dtype)`, clip/model.py:341), so the caller's cast pass (0.9 MB per image) is not needed"""
        return (self.conv1.weight.dtype == torch.float16 and self._f16_eligible()
                and all(b.stride in (1, 2) for li in (1, 2, 3, 4) for b in getattr(self, f"layer{li}")))

    @torch.no_grad()
    def forward(self, x, return_stages=False):
        if self.conv1.weight.dtype == torch.float16:
            if self._f16_eligible() and all(b.stride in (1, 2) for li in (1, 2, 3, 4) for b in getattr(self, f"layer{li}")):
                return self.forward_f16(x, return_stages)
            if not return_stages:
                # widths the fp16 kernels do not serve (RN50x4: 80, RN50x16: 96, toy towers): the fp32-accurate plan on the
                # fp16-stored weights, embedding rounded to the model dtype
                return self.forward(x, return_stages=True)[0].to(torch.float16)
        if self._plan is not None and self._plan_key != self._param_key():
            self._plan = None
        P = self._plan or self._compile()
        x = x.float().contiguous()                      # NCHW image at the boundary
        # One device scalar per conv output: its epilogue leaves max|y| there and the consumer
        # derives its fp16 scale from it (an average pool passes its input's bound on).
        n_slots = 3 + 4 * len(P["blocks"])
        amax = torch.zeros(n_slots, device=x.device, dtype=torch.float32)
        slot = [1]
        track = _opt["conv_split"] == "f16"          # the maxima are only needed by the fp16-pair kernels
        x = ops.conv_stem_s2(x, *P["stem1"], y_absmax=amax[0:1] if track else None)   # -> NHWC from here on

        def conv(t, t_am, c, res, k, pad, act, pool=1, keep_full=False):
            y_am = amax[slot[0]:slot[0] + 1] if track else None; slot[0] += 1
            y = ops.conv_bn_act(t, c["w"], c["b"], res, k, k, 1, pad, act, c["wl"], w_planes=c["p3"],
                                w_planes_f16=c["ph"], w_exp=c["we"], x_absmax=t_am, y_absmax=y_am, out_scale=c["sc"],
                                pool=pool, keep_full=keep_full)
            return y, y_am

        x, am = conv(x, amax[0:1] if track else None, P["stem2"], None, 3, 1, ops.ACT_RELU)
        x, am = conv(x, am, P["stem3"], None, 3, 1, ops.ACT_RELU, pool=2)      # + the stem's AvgPool2d(2)
        stages = {"stem": x}
        blocks = P["blocks"]
        x_pooled = None          # AvgPool2d(2) of x when the previous conv3 already produced it
        y1_next = None           # (conv1 output, its max slot) of the next block when the previous chain launch made it
        bi = 0
        for li in (1, 2, 3, 4):
            for _ in getattr(self, f"layer{li}"):
                e = blocks[bi]; bi += 1
                if y1_next is not None:
                    (out, oam), y1_next = y1_next, None
                else:
                    out, oam = conv(x, am, e["c1"], None, 1, 0, ops.ACT_RELU)
                nxt = blocks[bi] if bi < len(blocks) else None
                if (e["stride"] == 1 and track and _opt["fuse_chain"] and _opt["fuse_conv2"] and nxt is not None and nxt["stride"] == 1
                        and all(c["sc"] is not None and c["ph"] is not None for c in (e["c2"], e["c3"], nxt["c1"]))
                        and ("ds" not in e or ("dual" in e and _opt["fuse_ds"]))):
                    # conv2 -> conv3 + residual (or downsample branch) -> next conv1: the whole rest of the block in ONE launch
                    ndsl = 3 if "ds" in e else 2                # slots the separate launches would use after conv2's
                    x_am, y1_am = amax[slot[0] + 1:slot[0] + 2], amax[slot[0] + ndsl:slot[0] + ndsl + 1]
                    dual = dict(a2=x, a2_absmax=am, ds=e["ds"], ratio=e["dual"]["ratio"], bias=e["dual"]["bias"]) if "ds" in e else None
                    r = ops.bottleneck_block_chain(out, oam, e["c2"], e["c3"], nxt["c1"], residual=None if dual else x, dual=dual,
                                                   x_absmax=x_am, y1n_absmax=y1_am)
                    if r is not None:
                        slot[0] += ndsl + 1
                        x, am, x_pooled, y1_next = r[0], x_am, None, (r[1], y1_am)
                        continue
                if e["stride"] == 2:      # conv2 + bn2 + ReLU + AvgPool2d(2) in one epilogue
                    out, oam = conv(out, oam, e["c2"], None, 3, 1, ops.ACT_RELU, pool=2)
                else:
                    out, oam = conv(out, oam, e["c2"], None, 3, 1, ops.ACT_RELU)
                    if e["stride"] > 1:
                        out = ops.avgpool2d(out, e["stride"])
                identity = x
                fused = None
                if "ds" in e:
                    if e["stride"] > 1:
                        identity = x_pooled if (x_pooled is not None and e["stride"] == 2) else ops.avgpool2d(x, e["stride"])
                    nxt = blocks[bi] if bi < len(blocks) else None
                    if ("dual" in e and track and _opt["fuse_ds"] and _opt["fuse_chain"] and e["stride"] == 1 and nxt is not None
                            and nxt["c1"]["sc"] is not None and nxt["c1"]["ph"] is not None and "ds" not in nxt):
                        # ... and the same launch continues into the next block's conv1 (layer 1's first block)
                        x_am, y1_am = amax[slot[0]:slot[0] + 1], amax[slot[0] + 2:slot[0] + 3]
                        r = ops.bottleneck_chain_dual(out, oam, e["c3"], identity, am, e["ds"], e["dual"]["ratio"],
                                                      e["dual"]["bias"], nxt["c1"], x_am, y1_am)
                        if r is not None:
                            slot[0] += 3                  # conv3's and the branch's slots, and the next conv1's
                            x, am, x_pooled, y1_next = r[0], x_am, None, (r[1], y1_am)
                            continue
                    if "dual" in e and track and _opt["fuse_ds"]:
                        # out = relu(bn3(conv3(out)) + bn_d(conv_d(identity))) in one launch: the branch
                        # output is never materialised
                        y_am = amax[slot[0]:slot[0] + 1]
                        fused = ops.gemm_dual(out, oam, e["c3"]["ph"], e["c3"]["we"], e["c3"]["sc"], identity, am,
                                              e["ds"]["ph"], e["dual"]["ratio"], e["dual"]["bias"], ops.ACT_RELU, y_am)
                    if fused is not None:
                        slot[0] += 2                      # the two slots the unfused calls would have used
                        x, am, x_pooled = fused, y_am, None
                        continue
                    identity, _ = conv(identity, am, e["ds"], None, 1, 0, ops.ACT_NONE)
                # conv3 + bn3, residual add and the final ReLU fused into one epilogue; when the next
                # block downsamples, the same launch also writes AvgPool2d(2) of its output for that
                # block's downsample branch
                nxt = blocks[bi] if bi < len(blocks) else None
                want_pool = nxt is not None and nxt["stride"] == 2 and "ds" in nxt and identity.shape[1] % 2 == 0 and identity.shape[2] % 2 == 0
                if (nxt is not None and track and _opt["fuse_chain"] and e["c3"]["sc"] is not None and nxt["c1"]["sc"] is not None
                        and e["c3"]["ph"] is not None and nxt["c1"]["ph"] is not None
                        and (want_pool or not (nxt["stride"] == 2 and "ds" in nxt))):
                    # the same launch continues into the next block's conv1: x is written once, not re-read
                    x_am, y1_am = amax[slot[0]:slot[0] + 1], amax[slot[0] + 1:slot[0] + 2]
                    # (at a seam the un-pooled x is read by nobody else: it is only written when the caller wants the stage outputs)
                    r = ops.bottleneck_chain(out, oam, e["c3"], identity, nxt["c1"], x_am, y1_am, pooled=want_pool,
                                             keep_full=return_stages or _opt["seam_pool"] < 2)
                    if r is not None:
                        slot[0] += 2                      # this conv3's slot and the next conv1's
                        if want_pool:
                            x, x_pooled, y1 = r
                        else:
                            (x, y1), x_pooled = r, None
                        am, y1_next = x_am, (y1, y1_am)
                        continue
                if want_pool:
                    (x_pooled, x), am = conv(out, oam, e["c3"], identity, 1, 0, ops.ACT_RELU, pool=2, keep_full=True)
                else:
                    x_pooled = None
                    x, am = conv(out, oam, e["c3"], identity, 1, 0, ops.ACT_RELU)
            stages[f"layer{li}"] = x
        a = P["attn"]
        out = ops.attnpool(x, a["pos"], a["wq"], a["bq"], a["wkv"], a["bkv"], a["wc"], a["bc"], self.attnpool.num_heads)
        return (out, stages) if return_stages else out


# ---------------------------------------------------------------------------------------
# transformer towers
# ---------------------------------------------------------------------------------------

class _MLP(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.c_fc = _LinearW(d, 4 * d)
        self.c_proj = _LinearW(4 * d, d)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model, n_head, attn_mask=None):
        super().__init__()
        self.attn = _AttnParams(d_model)
        self.ln_1 = _LNParams(d_model)
        self.mlp = _MLP(d_model)
        self.ln_2 = _LNParams(d_model)
        self.n_head = n_head


class Transformer(nn.Module):
    def __init__(self, width, layers, heads, attn_mask=None):
        super().__init__()
        self.width, self.layers, self.heads = width, layers, heads
        self.causal = attn_mask is not None
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads) for _ in range(layers)])

    def _apply(self, fn, *a, **k):
        self._planes = None
        return super()._apply(fn, *a, **k)

    def _weight_key(self):
        return (_opt_version[0],) + tuple((w.data_ptr(), w._version) for b in self.resblocks
                                          for w in (b.attn.in_proj_weight, b.attn.out_proj.weight, b.mlp.c_fc.weight, b.mlp.c_proj.weight))

    @torch.no_grad()
    def _split_planes(self):
        """pre-split planes of the four projection weights of every block for the split-precision
        GEMMs: fp16 (one plane when the weight is exact in fp16, as after build_model; else hi + lo)
        or, with conv_split = "bf16", the bf16 triple; "off" = fp32-input MFMA."""
        def one(w):
            w = w.detach().contiguous()
            if _opt["conv_split"] == "f16" and w.shape[1] % 16 == 0:
                ph, we, _ = ops.split_planes_f16(w, allow_single=True)
                return dict(w_planes_f16=ph, w_exp=we)
            if _opt["conv_split"] == "bf16":
                return dict(w_planes=ops.split_planes(w))
            return {}
        self._planes = [tuple(one(w) for w in (b.attn.in_proj_weight, b.attn.out_proj.weight, b.mlp.c_fc.weight,
                                                b.mlp.c_proj.weight)) for b in self.resblocks]
        self._planes_key = self._weight_key()
        return self._planes

    @torch.no_grad()
    def _f16_plan(self):
        """fp16 mode: projection weights as contiguous fp16 [N][K] (already fp16 after convert_weights), biases and
        LayerNorm parameters in fp32; rebuilt when a parameter is replaced or edited."""
        key = tuple((t.data_ptr(), t._version, t.dtype) for t in self.parameters())
        if getattr(self, "_f16", None) is None or self._f16[0] != key:
            h = lambda t: t.detach().to(torch.float16).contiguous()
            f = lambda t: t.detach().float().contiguous()
            plan = []
            for b in self.resblocks:
                plan.append(dict(ln1=(f(b.ln_1.weight), f(b.ln_1.bias)), ln2=(f(b.ln_2.weight), f(b.ln_2.bias)),
                                 w_in=h(b.attn.in_proj_weight), b_in=f(b.attn.in_proj_bias),
                                 w_out=h(b.attn.out_proj.weight), b_out=f(b.attn.out_proj.bias),
                                 w_fc=h(b.mlp.c_fc.weight), b_fc=f(b.mlp.c_fc.bias),
                                 w_proj=h(b.mlp.c_proj.weight), b_proj=f(b.mlp.c_proj.bias)))
            self._f16 = (key, plan)
        return self._f16[1]

    @torch.no_grad()
    def run_f16(self, x, B, L):
        """fp16 mode (clip/model.py:185-192 with fp16 activations): x f16 [B*L, E]; bias / QuickGELU / residual adds in
        the GEMM epilogues, LayerNorm statistics and softmax in fp32."""
        E = self.width
        for e in self._f16_plan():
            h = ops.layernorm_f16(x, *e["ln1"])
            qkv = ops.gemm_f16(h, e["w_in"], e["b_in"])
            o = ops.mha_core_f16(qkv, B, L, E, self.heads, self.causal)
            x = ops.gemm_f16(o, e["w_out"], e["b_out"], residual=x)
            h = ops.layernorm_f16(x, *e["ln2"])
            h = ops.gemm_f16(h, e["w_fc"], e["b_fc"], act=ops.ACT_QUICKGELU)
            x = ops.gemm_f16(h, e["w_proj"], e["b_proj"], residual=x)
        return x

    @torch.no_grad()
    def run(self, x, B, L):
        """x [B*L, E] batch-first rows.  Pre-LN blocks (clip/model.py:189-192): the residual
        adds and QuickGELU are GEMM epilogues.  Every GEMM input carries a device scalar with (a
        bound of) its maximum for the fp16-pair kernel: written by the LayerNorm kernel and the
        GEMM epilogues; the attention core's output is a convex combination of V rows, so the
        qkv GEMM's scalar bounds it."""
        E = self.width
        planes = getattr(self, "_planes", None)
        if planes is None or self._planes_key != self._weight_key():     # load_state_dict / in-place edits
            planes = self._split_planes()
        amax = torch.zeros(4 * len(self.resblocks), device=x.device, dtype=torch.float32)
        f16 = _opt["conv_split"] == "f16"
        for i, (blk, (p_in, p_out, p_fc, p_proj)) in enumerate(zip(self.resblocks, planes)):
            a = [amax[4 * i + j:4 * i + j + 1] if f16 else None for j in range(4)]
            sc = (lambda am: dict(a_absmax=am)) if f16 else (lambda am: {})
            h = ops.layernorm(x, blk.ln_1.weight, blk.ln_1.bias, y_absmax=a[0])
            qkv = ops.gemm(h, blk.attn.in_proj_weight, blk.attn.in_proj_bias, c_absmax=a[1], **sc(a[0]), **p_in)
            o = ops.mha_core(qkv, B, L, E, self.heads, self.causal, qkv_absmax=a[1])
            x = ops.gemm(o, blk.attn.out_proj.weight, blk.attn.out_proj.bias, residual=x, **sc(a[1]), **p_out)
            h = ops.layernorm(x, blk.ln_2.weight, blk.ln_2.bias, y_absmax=a[2])
            h = ops.gemm(h, blk.mlp.c_fc.weight, blk.mlp.c_fc.bias, act=ops.ACT_QUICKGELU, c_absmax=a[3], **sc(a[2]), **p_fc)
            x = ops.gemm(h, blk.mlp.c_proj.weight, blk.mlp.c_proj.bias, residual=x, **sc(a[3]), **p_proj)
        return x


class VisionTransformer(nn.Module):
    def __init__(self, input_resolution, patch_size, width, layers, heads, output_dim):
        super().__init__()
        self.input_resolution, self.output_dim, self.patch_size = input_resolution, output_dim, patch_size
        self.conv1 = _ConvW(3, width, patch_size)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = _LNParams(width)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = _LNParams(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))

    def rounds_fp32_images_itself(self):
        """fp16 mode: the patch gather reads an fp32 image and rounds it to fp16 (the values of `image.type(self.dtype)`, clip/model.py:341)"""
        return self.conv1.weight.dtype == torch.float16

    @torch.no_grad()
    def forward_f16(self, x):
        """fp16 mode: patch GEMM (K zero-padded to a multiple of 64), fp16 token stream, fp16 output [B, D]."""
        x = x.contiguous()
        if x.dtype not in (torch.float16, torch.float32):
            x = x.float()
        B = x.shape[0]
        W = self.conv1.weight.shape[0]
        L = self.positional_embedding.shape[0]
        K = 3 * self.patch_size ** 2
        Kp = (K + 63) // 64 * 64
        key = tuple((t.data_ptr(), t._version, t.dtype) for t in (self.conv1.weight, self.proj, self.class_embedding,
                                                                    self.positional_embedding, self.ln_pre.weight, self.ln_post.weight))
        if getattr(self, "_f16", None) is None or self._f16[0] != key:
            w1 = torch.zeros((W, Kp), device=x.device, dtype=torch.float16)
            w1[:, :K] = self.conv1.weight.detach().reshape(W, K).to(torch.float16)
            f = lambda t: t.detach().float().contiguous()
            self._f16 = (key, dict(w1=w1, proj_t=self.proj.detach().t().to(torch.float16).contiguous(), cls=f(self.class_embedding),
                                   pos=f(self.positional_embedding), ln_pre=(f(self.ln_pre.weight), f(self.ln_pre.bias)),
                                   ln_post=(f(self.ln_post.weight), f(self.ln_post.bias))))
        e = self._f16[1]
        cols = ops.im2col_patch_f16(x, self.patch_size, Kp)                       # [B*g*g, Kp]
        patches = ops.gemm_f16(cols, e["w1"])                                      # conv1 has no bias
        t = ops.vit_tokens_f16(patches, e["cls"], e["pos"], B)
        t = ops.layernorm_f16(t, *e["ln_pre"])
        t = self.transformer.run_f16(t, B, L)
        c = ops.layernorm_f16(t, *e["ln_post"], rows=B, ldx=L * W)                 # token 0 only
        return ops.gemm_f16(c, e["proj_t"])

    @torch.no_grad()
    def forward(self, x):
        if self.conv1.weight.dtype == torch.float16:
            return self.forward_f16(x)
        x = x.float().contiguous()
        B = x.shape[0]
        W = self.conv1.weight.shape[0]
        L = self.positional_embedding.shape[0]
        w1 = self.conv1.weight.reshape(W, -1)
        kw = {}
        x_am = None
        if _opt["conv_split"] == "f16" and w1.shape[1] % 16 == 0:                 # fp16-pair GEMM: planes cached per weight
            key = (w1.data_ptr(), w1._version, _opt_version[0])
            if getattr(self, "_conv1_planes", (None,))[0] != key:
                ph, we, _ = ops.split_planes_f16(w1.detach().contiguous(), allow_single=True)
                self._conv1_planes = (key, ph, we)
            x_am = torch.zeros(1, device=x.device, dtype=torch.float32)   # max|x|, left there by the im2col kernel
            kw = dict(w_planes_f16=self._conv1_planes[1], w_exp=self._conv1_planes[2], a_absmax=x_am)
        cols = ops.im2col_patch(x, self.patch_size, out_absmax=x_am)      # [B*g*g, 3*P*P]
        patches = ops.gemm(cols, w1, **kw)                                # conv1 has no bias
        t = ops.vit_tokens(patches, self.class_embedding, self.positional_embedding, B)
        t = ops.layernorm(t, self.ln_pre.weight, self.ln_pre.bias)        # [B*L, W]
        t = self.transformer.run(t, B, L)
        c = ops.layernorm(t, self.ln_post.weight, self.ln_post.bias, rows=B, ldx=L * W)   # token 0 only
        return ops.gemm(c, self.proj, trans_w=True)


class CLIP(nn.Module):
    def __init__(self, embed_dim: int, image_resolution: int, vision_layers: Union[Tuple[int, int, int, int], int],
                 vision_width: int, vision_patch_size: int, context_length: int, vocab_size: int,
                 transformer_width: int, transformer_heads: int, transformer_layers: int):
        super().__init__()
        self.context_length = context_length
        if isinstance(vision_layers, (tuple, list)):
            self.visual = ModifiedResNet(layers=vision_layers, output_dim=embed_dim, heads=vision_width * 32 // 64,
                                         input_resolution=image_resolution, width=vision_width)
        else:
            self.visual = VisionTransformer(input_resolution=image_resolution, patch_size=vision_patch_size,
                                            width=vision_width, layers=vision_layers, heads=vision_width // 64,
                                            output_dim=embed_dim)
        self.transformer = Transformer(width=transformer_width, layers=transformer_layers, heads=transformer_heads,
                                       attn_mask=True)
        self.vocab_size = vocab_size
        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, transformer_width))
        self.ln_final = _LNParams(transformer_width)
        self.text_projection = nn.Parameter(torch.empty(transformer_width, embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    def encode_image(self, image):
        if image.dtype == torch.float32 and self.dtype == torch.float16 and getattr(self.visual, "rounds_fp32_images_itself", lambda: False)():
            return self.visual(image)                  # (same values as the cast below, without the extra pass over the batch)
        return self.visual(image.type(self.dtype))

    @torch.no_grad()
    def encode_text(self, text):
        if self.dtype == torch.float16:                 # fp16 mode (clip/model.py:343-356 with fp16 activations)
            x, tok = ops.embed_gather_f16(text, self.token_embedding.weight.float(), self.positional_embedding.float())
            n, L, W = x.shape
            x = self.transformer.run_f16(x.view(n * L, W), n, L)
            e = ops.gather_eot_f16(tok, x.view(n, L, W))
            e = ops.layernorm_f16(e, self.ln_final.weight.float().contiguous(), self.ln_final.bias.float().contiguous())
            return ops.gemm_f16(e, self.text_projection.detach().t().to(torch.float16).contiguous())
        x, tok = ops.embed_gather(text, self.token_embedding.weight, self.positional_embedding)
        n, L, W = x.shape
        x = self.transformer.run(x.view(n * L, W), n, L)
        # LayerNorm is per row, so gather the EOT row first and normalise only that row
        e = ops.gather_eot(tok, x.view(n, L, W))
        e = ops.layernorm(e, self.ln_final.weight, self.ln_final.bias)
        return ops.gemm(e, self.text_projection, trans_w=True)

    def forward(self, image, text):
        image_features = self.encode_image(image)
        text_features = self.encode_text(text)
        # clip/model.py:362-370: cosine similarity as logits -- row normalisation and the [B, n] product on the kernels
        # (fp16 mode: on the .float() embeddings, result cast back)
        out_dtype = image_features.dtype
        image_features = ops.l2norm_rows(image_features.float().contiguous())
        text_features = ops.l2norm_rows(text_features.float().contiguous())
        logits_per_image = ops.gemm(image_features, text_features, alpha=float(self.logit_scale.exp())).to(out_dtype)
        return logits_per_image, logits_per_image.t()


def convert_weights(model: nn.Module):
    """clip/model.py:375-396: conv / linear / attention-projection weights and biases, `text_projection` and `proj` to
    fp16 in place (BatchNorm, LayerNorm, embeddings stay fp32).  `model.dtype` becomes float16 and the transformer
    towers switch to the fp16 throughput mode (module docstring); `model.float()` switches back."""
    def _half(t):
        if t is not None and t.is_floating_point():
            t.data = t.data.half()

    for m in model.modules():
        if isinstance(m, (_ConvW, _LinearW)):
            _half(m.weight); _half(getattr(m, "bias", None))
        if isinstance(m, _AttnParams):
            _half(m.in_proj_weight); _half(m.in_proj_bias)
        for name in ("text_projection", "proj"):
            if isinstance(getattr(m, name, None), nn.Parameter):
                _half(getattr(m, name))
    return model


def build_model(state_dict: dict):
    """Infer the architecture from key shapes exactly as clip/model.py:399-436 does."""
    vit = "visual.proj" in state_dict
    if vit:
        vision_width = state_dict["visual.conv1.weight"].shape[0]
        vision_layers = len([k for k in state_dict if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
        vision_patch_size = state_dict["visual.conv1.weight"].shape[-1]
        grid = round((state_dict["visual.positional_embedding"].shape[0] - 1) ** 0.5)
        image_resolution = vision_patch_size * grid
    else:
        vision_layers = tuple(len({k.split(".")[2] for k in state_dict if k.startswith(f"visual.layer{b}")})
                              for b in (1, 2, 3, 4))
        vision_width = state_dict["visual.layer1.0.conv1.weight"].shape[0]
        out_w = round((state_dict["visual.attnpool.positional_embedding"].shape[0] - 1) ** 0.5)
        vision_patch_size = None
        if out_w ** 2 + 1 != state_dict["visual.attnpool.positional_embedding"].shape[0]:
            raise RuntimeError("attnpool.positional_embedding is not (n*n + 1) rows")
        image_resolution = out_w * 32
    embed_dim = state_dict["text_projection"].shape[1]
    context_length = state_dict["positional_embedding"].shape[0]
    vocab_size = state_dict["token_embedding.weight"].shape[0]
    transformer_width = state_dict["ln_final.weight"].shape[0]
    transformer_layers = len({k.split(".")[2] for k in state_dict if k.startswith("transformer.resblocks")})
    model = CLIP(embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size, context_length,
                 vocab_size, transformer_width, transformer_width // 64, transformer_layers)
    sd = OrderedDict((k, _as_loaded(k, v)) for k, v in state_dict.items()
                     if k not in ("input_resolution", "context_length", "vocab_size"))
    model.load_state_dict(sd)
    return model.eval()


# The reference converts conv / linear / attention-projection weights and biases and the two
# projection matrices to fp16 BEFORE load_state_dict (convert_weights, clip/model.py:375-396,433),
# so whatever the checkpoint holds is rounded to fp16 there; on the CPU path the model is then
# cast back with .float() (clip/clip.py:139-141).  BatchNorm / LayerNorm parameters, positional
# and class embeddings, the token table and logit_scale stay fp32.
_FP16_KEYS = re.compile(r"(conv\d\.weight|downsample\.0\.weight|_proj\.(weight|bias)|in_proj_(weight|bias)"
                        r"|c_fc\.(weight|bias))$")


def _as_loaded(key, v):
    if not v.is_floating_point():
        return v
    if _FP16_KEYS.search(key) or key in ("text_projection", "visual.proj"):
        return v.half().float()
    return v.float()
