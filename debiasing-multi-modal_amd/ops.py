"""Tensor-level wrappers over the C ABI (include/dbmm.h).  Plumbing only: allocate outputs
with torch's caching allocator, pass raw pointers + the current HIP stream, raise on error.
Every function requires HIP tensors -- there is no CPU path here.
"""
import ctypes
import math
import os

import torch

from . import _lib
from ._lib import check, ptr, require_cuda, stream

ACT_NONE, ACT_RELU, ACT_QUICKGELU = 0, 1, 2


class DbmmUnsupported(_lib.DbmmError):
    """a wrapper's C entry answered DBMM_E_UNSUPPORTED (no kernel for this valid request)"""


def set_option(name, value):
    """library option by name (include/dbmm.h dbmm_set_option; names in csrc/options.hip); returns the previous value"""
    old = get_option(name)
    check(_lib.lib().dbmm_set_option(name.encode(), int(value)), f"set_option({name})")
    return old


def get_option(name):
    v = ctypes.c_int()
    check(_lib.lib().dbmm_get_option(name.encode(), ctypes.byref(v)), f"get_option({name})")
    return v.value


# every output / workspace of this module comes from here (torch's caching allocator); tests swap in an allocator
# that surrounds each tensor with guard zones to catch writes outside the tensor (tests/test_gpu_headline.py)
_empty = torch.empty

# ---- optional per-launch timing of the MFMA kernel family (bench.py roofline leg) ----------
_prof = None
_prof_conv_only = False
_prof_paused = False


def profile_begin(conv_only=False):
    """start collecting (kernel tag, flops, start event, end event) for every igemm launch --
    with conv_only, just the KxK convolutions (the event pairs are stream work themselves, ~4 us
    per launch, so a bench that only needs the dominant kernel should not pay for all)."""
    global _prof, _prof_conv_only, _prof_paused
    _prof, _prof_conv_only, _prof_paused = [], bool(conv_only), False


def profile_sample(on):
    """inside a profile_begin() ... profile_end() window: switch the per-launch events off / on again (bench.py times the launches of
    every k-th step only: an event pair is ~4 us of stream time, ~60 launches per RN50 step)"""
    global _prof_paused
    _prof_paused = not on


def profile_end():
    """stop collecting; returns {tag: (launches, total_flops, total_ms, total_algorithmic_bytes)} (synchronises)."""
    global _prof
    rec, _prof = _prof or [], None
    torch.cuda.synchronize()
    out = {}
    for tag, flops, nbytes, e0, e1 in rec:
        n, f, ms, by = out.get(tag, (0, 0.0, 0.0, 0.0))
        out[tag] = (n + 1, f + flops, ms + e0.elapsed_time(e1), by + nbytes)
    return out


def _last_igemm_tag():
    """exact instantiation of the igemm launch just issued, spelled like rocprofv3's kernel name"""
    cfg = (ctypes.c_int * 11)()
    _lib.lib().dbmm_debug_last_igemm(cfg)
    if cfg[8] == 5:        # conv3 + downsample dual-source GEMM: the x3 kernel with TWO = 1
        return f"igemm_x3_kernel<{cfg[0]}, {cfg[1]}, {cfg[2]}, {cfg[3]}, 0, {cfg[7]}, {cfg[9]}, 2, 1, 32, 1>"
    if cfg[8] == 8:        # the dual-source GEMM on the deep-pipelined kernel: <ACT, RES, TWO = 1>
        return "gemm_pair_8ph_kernel<dual>"
    if cfg[8] == 9:        # conv3 + residual, short K into many channels (conv1x1_res_stream.hip): <K, POOL>
        return f"conv1x1_res_stream_kernel<{cfg[0]}, {cfg[5]}>"
    if cfg[8] == 6:        # deep-pipelined parity GEMM (gemm_pair_8ph.hip): <ACT, RES> are not reported
        return "gemm_pair_8ph_kernel"
    if cfg[8] == 7:        # eight-phase 3x3 halo kernel (conv3x3_halo8.hip): <POOL, ACT>, ACT not reported
        return f"conv3x3_halo8{'n' if cfg[1] == 128 else ''}_kernel<{cfg[5]}>"
    if cfg[8] == 4:        # 3x3 halo kernel: <BN, WAVES_M, WAVES_N, MINB, SK, POOL>
        return f"igemm_halo_kernel<{cfg[1]}, {cfg[2]}, {cfg[3]}, {cfg[7]}, {cfg[9]}, {cfg[5]}>"
    if cfg[8] in (2, 3):   # split-precision kernels: <BM, BN, WAVES_M, WAVES_N, AMODE, MINB, SK, NP, NW, BK, TWO>
        return (f"igemm_x3_kernel<{cfg[0]}, {cfg[1]}, {cfg[2]}, {cfg[3]}, {cfg[4]}, {cfg[7]}, {cfg[9]}, {cfg[8]}, "
                f"{cfg[10]}, {cfg[6]}, 0>")
    return "igemm_f32_kernel<" + ", ".join(str(v) for v in cfg) + ">"


class _Timed:
    """HIP events (on the launch stream = torch's current stream) around one igemm launch.  nbytes = the launch's
    ALGORITHMIC bytes: every operand read once, every output written once."""
    def __init__(self, M, N, K, amode, wmode, nbytes=0):
        self.args = (M, N, K, amode, wmode)
        self.nbytes = float(nbytes)

    def __enter__(self):
        if _TRACE_LAUNCHES:
            with open(_TRACE_LAUNCHES, "a") as f:
                f.write(f"igemm {self.args} ... ")
        self.on = _prof is not None and not _prof_paused and (self.args[3] == 1 or not _prof_conv_only)
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if _TRACE_LAUNCHES:
            torch.cuda.synchronize()
            with open(_TRACE_LAUNCHES, "a") as f:
                f.write("ok\n")
        if self.on and _prof is not None and exc[0] is None:
            M, N, K, amode, wmode = self.args
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            tag = _chain_tag if amode == -1 else _last_igemm_tag()
            _prof.append((tag, 2.0 * M * N * K, self.nbytes, self.e0, e1))


_ws_cache = {}


def igemm_workspace(device):
    """per-device scratch for the stream-K work split (allocated once, reused by every launch on
    the stream; launches on one stream are ordered, so sharing it is safe)."""
    key = (device.type, device.index)
    ws = _ws_cache.get(key)
    if ws is None:
        ws = _empty(_lib.lib().dbmm_workspace_bytes_igemm() // 4, device=device, dtype=torch.float32)
        _ws_cache[key] = ws
    return ws


def _sized(name, t, n):
    """an optional per-channel / per-row operand the library will index as n elements (it only sees the pointer)"""
    if t is not None and t.numel() != n:
        raise RuntimeError(f"{name}: {t.numel()} elements where the kernels will index {n}")


def _f32c(t):
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise _lib.DbmmError(f"expected a contiguous float32 tensor, got {t.dtype} contiguous={t.is_contiguous()}")
    return t


def split_planes(w):
    """fp32 [N][K] weight -> three bf16 planes [3][N][K] (hi, mid, lo) for the split-precision kernels"""
    require_cuda(w)
    _f32c(w)
    N, K = w.shape[0], w.numel() // w.shape[0]
    planes = _empty((3, N, K), device=w.device, dtype=torch.bfloat16)
    check(_lib.lib().dbmm_split_weight_planes(ptr(w), ptr(planes), N, K, stream()), "split_weight_planes")
    return planes


def split_planes_f16(w, allow_single=False):
    """fp32 [N][K] weight -> (fp16 planes [n][N][K] of w * 2^w_exp, w_exp, n) for the fp16-pair
    kernel; w_exp puts max|w| just below 2^14 (one host read of the maximum, at plan time).
    n = 2 (hi, lo) in general.  With allow_single, n = 1 when the scaled weight is exactly
    representable in fp16 -- the case for every conv / linear weight that went through the
    reference's build_model, which stores them in fp16 -- and K is a multiple of 32."""
    require_cuda(w)
    _f32c(w)
    N, K = w.shape[0], w.numel() // w.shape[0]
    m = float(w.abs().max())
    w_exp = 0 if not (m > 0.0 and math.isfinite(m)) else max(-40, min(40, 13 - math.frexp(m)[1] + 1))
    if allow_single and K % 32 == 0:
        ws = w.reshape(N, K) * (2.0 ** w_exp)
        h = ws.half()
        if torch.equal(h.float(), ws):
            return h.reshape(1, N, K).contiguous(), w_exp, 1
    planes = _empty((2, N, K), device=w.device, dtype=torch.float16)
    check(_lib.lib().dbmm_split_weight_planes_f16(ptr(w), ptr(planes), N, K, w_exp, stream()), "split_weight_planes_f16")
    return planes, w_exp, 2


def gemm(a, w, bias=None, residual=None, act=ACT_NONE, alpha=1.0, trans_a=False, trans_w=False,
         M=None, N=None, K=None, lda=None, out=None, w_planes=None, w_planes_f16=None, w_exp=0, a_absmax=None,
         c_absmax=None):
    """c = act(alpha * (op(a) @ op(w)^T + bias) + residual); see dbmm_gemm_bias_act.
    w_planes: bf16 triple; w_planes_f16 / w_exp + a_absmax: fp16-pair kernel (split_planes_f16);
    c_absmax (1-element device tensor, zeroed by the caller) receives max|c|."""
    require_cuda(a, w)
    _f32c(w)
    if a.dtype != torch.float32:
        raise _lib.DbmmError("gemm needs float32")
    if lda is None:
        _f32c(a)
        lda = a.shape[-1]
    if M is None:
        M = a.shape[-1] if trans_a else a.numel() // a.shape[-1]
    if K is None:
        K = a.numel() // a.shape[-1] if trans_a else a.shape[-1]
    if N is None:
        N = w.shape[1] if trans_w else w.shape[0]
    if out is None:
        out = _empty((M, N), device=a.device, dtype=torch.float32)
    ldr = residual.shape[-1] if residual is not None else 0
    # the library sees raw pointers and leading dimensions: what it will index must be there
    if w.numel() != N * K or (not trans_a and a.numel() < (M - 1) * lda + K) or out.numel() < (M - 1) * out.shape[-1] + N:
        raise RuntimeError(f"gemm: operands {tuple(a.shape)}, {tuple(w.shape)}, out {tuple(out.shape)} do not hold an M = {M}, N = {N}, K = {K} product")
    _sized("gemm bias", bias, N)
    if residual is not None and (ldr < N or residual.numel() < (M - 1) * ldr + N):
        raise RuntimeError(f"gemm: residual {tuple(residual.shape)} for an [{M}, {N}] output")
    if w_planes_f16 is not None and tuple(w_planes_f16.shape[1:]) != (N, K):
        raise RuntimeError(f"gemm: weight planes {tuple(w_planes_f16.shape)} for a [{N}, {K}] weight")
    ws = igemm_workspace(a.device)
    # algorithmic bytes: A and the residual read once, C written once, W once in the form the kernel reads it
    nby = 4 * (M * K + M * N * (2 if residual is not None else 1)) + \
        (w_planes_f16.numel() * 2 if w_planes_f16 is not None else (w_planes.numel() * 2 if w_planes is not None else 4 * N * K))
    if (w_planes_f16 is not None or c_absmax is not None) and not trans_a and not trans_w:
        nw = 2 if w_planes_f16 is None else int(w_planes_f16.shape[0])
        with _Timed(M, N, K, 0, 0, nby):
            check(_lib.lib().dbmm_gemm_bias_act_x2(ptr(a), lda, ptr(a_absmax), ptr(w), ptr(w_planes_f16), nw, int(w_exp),
                                                   w.shape[-1], None, ptr(bias), ptr(residual), ldr, ptr(out),
                                                   out.shape[-1], ptr(c_absmax), M, N, K, float(alpha), act, ptr(ws),
                                                   ws.numel() * 4, stream()), "gemm_bias_act_x2")
        return out
    if w_planes is not None and not trans_a and not trans_w:
        with _Timed(M, N, K, 0, 0, nby):
            check(_lib.lib().dbmm_gemm_bias_act_x3(ptr(a), lda, ptr(w), ptr(w_planes), w.shape[-1], ptr(bias),
                                                   ptr(residual), ldr, ptr(out), out.shape[-1], M, N, K, float(alpha),
                                                   act, ptr(ws), ws.numel() * 4, stream()), "gemm_bias_act_x3")
        return out
    with _Timed(M, N, K, 2 if trans_a else 0, int(trans_w), nby):
        check(_lib.lib().dbmm_gemm_bias_act_ws(ptr(a), lda, int(trans_a), ptr(w), w.shape[-1], int(trans_w), ptr(bias),
                                               ptr(residual), ldr, ptr(out), out.shape[-1], M, N, K, float(alpha), act,
                                               ptr(ws), ws.numel() * 4, stream()), "gemm_bias_act")
    return out


WL_TAP_MAJOR, WL_CHUNK_MAJOR, WL_CHUNK32_MAJOR = 0, 1, 2


def pack_conv_weight(w_oihw, chunk_major=False):
    """[Cout][Cin][kh][kw] -> (packed fp32 [Cout][K], layout id).

    Default K order (kh, kw, cin).  chunk_major=True packs (cin/16, kh, kw, 16), chunk_major=32
    packs (cin/32, kh, kw, 32) -- the taps of a
    channel slab adjacent along K -- which cuts the fabric re-reads of the 3x3 gather (each
    input pixel is read by 9 taps) but measured 1-5 % *slower* on MI355X at B = 512: the
    Infinity Cache already absorbs the re-reads and tap-major uses both halves of every 128-B
    line back to back.  Kept as an option (needs Cin % 16 == 0)."""
    Cout, Cin, kh, kw = w_oihw.shape
    slab = 32 if chunk_major == 32 else 16
    if chunk_major and kh * kw > 1 and Cin % slab == 0:
        w = w_oihw.reshape(Cout, Cin // slab, slab, kh, kw).permute(0, 1, 3, 4, 2)
        return w.contiguous().float().reshape(Cout, kh * kw * Cin), (WL_CHUNK32_MAJOR if slab == 32 else WL_CHUNK_MAJOR)
    return w_oihw.permute(0, 2, 3, 1).contiguous().float().reshape(Cout, kh * kw * Cin), WL_TAP_MAJOR


def conv_bn_act(x, w, bias, residual, kh, kw, stride, pad, act, w_layout=WL_TAP_MAJOR, w_planes=None,
                w_planes_f16=None, w_exp=0, x_absmax=None, y_absmax=None, out_scale=None, pool=1, keep_full=False):
    """x NHWC [B,H,W,Cin]; w packed [Cout][K] (BN folded) in `w_layout` order (default
    [Cout][kh][kw][Cin]; see pack_conv_weight); returns NHWC.
    w_planes: bf16 triple (split_planes) -> split-precision kernel.
    w_planes_f16 / w_exp (split_planes_f16) + x_absmax (1-element device tensor >= max|x|) ->
    fp16-pair kernel; y_absmax (1-element device tensor, zeroed by the caller) receives max|y|.
    out_scale ([Cout], x2 entry only): per-channel scale of the accumulator before the bias.
    pool = 2: followed by AvgPool2d(2) (returns the pooled tensor); fused into the conv epilogue
    when the library has a kernel for it, otherwise the two calls are composed here.
    keep_full (with pool = 2): returns (pooled, un-pooled) -- both written by the one launch."""
    require_cuda(x, w)
    _f32c(x); _f32c(w)
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    if pool not in (1, 2):
        raise _lib.DbmmError("conv_bn_act: pool must be 1 or 2")
    if w.numel() != Cout * kh * kw * Cin:
        raise RuntimeError(f"conv_bn_act: packed weight {tuple(w.shape)} for a {kh} x {kw} conv over {Cin} channels")
    _sized("conv_bn_act bias", bias, Cout); _sized("conv_bn_act out_scale", out_scale, Cout)
    if residual is not None and tuple(residual.shape) != (B, Ho, Wo, Cout):
        raise RuntimeError(f"conv_bn_act: residual {tuple(residual.shape)} for a {(B, Ho, Wo, Cout)} output")
    if w_planes_f16 is not None and w_planes_f16.numel() != w_planes_f16.shape[0] * w.numel():
        raise RuntimeError(f"conv_bn_act: weight planes {tuple(w_planes_f16.shape)} for a packed weight {tuple(w.shape)}")
    plain = kh == 1 and kw == 1 and stride == 1 and pad == 0
    split = w_planes_f16 is not None or y_absmax is not None or out_scale is not None
    # algorithmic bytes: input map and residual read once, output(s) written once, weights once
    wby = w_planes_f16.numel() * 2 if w_planes_f16 is not None else (w_planes.numel() * 2 if w_planes is not None else w.numel() * 4)
    oel = B * Ho * Wo * Cout
    nby_pool = 4 * (x.numel() + (residual.numel() if residual is not None else 0) + oel // 4 + (oel if keep_full else 0)) + wby
    nby = 4 * (x.numel() + (residual.numel() if residual is not None else 0) + oel) + wby
    # the 32-channel stem convs: persistent patch kernel (csrc/conv_patch.hip); option conv_patch = 0 disables
    if (Cin == 32 and kh == 3 and kw == 3 and stride == 1 and pad == 1 and act == ACT_RELU and residual is None and not keep_full
            and w_planes_f16 is not None and w_planes_f16.shape[0] == 1 and out_scale is not None and x_absmax is not None
            and Cout in (32, 64) and H % 4 == 0 and W % 28 == 0 and (pool == 1 or (H % 2 == 0 and W % 2 == 0))
            and w_layout in (WL_TAP_MAJOR, WL_CHUNK32_MAJOR) and get_option("conv_patch")):
        y = _empty((B, H // pool, W // pool, Cout), device=x.device, dtype=torch.float32)
        global _chain_tag
        _chain_tag = f"conv3x3_c32_kernel<{Cout}, {int(pool == 2)}>"
        t = _Timed(B * H * W, Cout, 9 * Cin, -1, 0, 4 * (x.numel() + y.numel()) + 2 * Cout * 9 * Cin)
        t.__enter__()
        rc = _lib.lib().dbmm_conv3x3_c32_bn_relu_x2(ptr(x), ptr(x_absmax), ptr(w_planes_f16), int(w_exp), ptr(out_scale), ptr(bias),
                                                   ptr(y), ptr(y_absmax), B, H, W, Cin, Cout, 2 if pool == 2 else 0, stream())
        if rc == 0:
            t.__exit__(None, None, None)
            return y
        if rc not in (_lib.E_UNSUPPORTED, _lib.E_ALIGN):      # a shape / alignment the patch kernel does not serve: the general kernels below do
            check(rc, "conv3x3_c32_bn_relu_x2")
        del y
    if pool == 2 and split and Ho % 2 == 0 and Wo % 2 == 0:
        y = _empty((B, Ho // 2, Wo // 2, Cout), device=x.device, dtype=torch.float32)
        yf = _empty((B, Ho, Wo, Cout), device=x.device, dtype=torch.float32) if keep_full else None
        t = _Timed(B * Ho * Wo, Cout, kh * kw * Cin, 0 if plain else 1, 0, nby_pool)
        t.__enter__()
        rc = _conv_x2(x, w, bias, residual, y, kh, kw, stride, pad, act, w_layout, w_planes_f16, w_exp, x_absmax,
                      y_absmax, out_scale, 2, yf)
        if rc == 0:
            t.__exit__(None, None, None)
            return (y, yf) if keep_full else y
        if rc != _lib.E_UNSUPPORTED:
            check(rc, "conv_bn_act_x2(pool)")
    y = _empty((B, Ho, Wo, Cout), device=x.device, dtype=torch.float32)
    with _Timed(B * Ho * Wo, Cout, kh * kw * Cin, 0 if plain else 1, 0, nby):
        ws = igemm_workspace(x.device)
        if split:
            check(_conv_x2(x, w, bias, residual, y, kh, kw, stride, pad, act, w_layout, w_planes_f16, w_exp, x_absmax,
                           y_absmax, out_scale, 0, None), "conv_bn_act_x2")
        elif w_planes is not None:
            check(_lib.lib().dbmm_conv_bn_act_x3(ptr(x), ptr(w), ptr(w_planes), ptr(bias), ptr(residual), ptr(y), B, H, W,
                                                 Cin, Cout, kh, kw, stride, pad, act, int(w_layout), ptr(ws),
                                                 ws.numel() * 4, stream()), "conv_bn_act_x3")
        else:
            check(_lib.lib().dbmm_conv_bn_act_ws(ptr(x), ptr(w), ptr(bias), ptr(residual), ptr(y), B, H, W, Cin, Cout, kh,
                                                 kw, stride, pad, act, int(w_layout), ptr(ws), ws.numel() * 4, stream()),
                  "conv_bn_act")
    if pool == 2:      # (y_absmax of the unpooled map bounds the pooled one)
        return (avgpool2d(y, 2), y) if keep_full else avgpool2d(y, 2)
    return y


def _conv_x2(x, w, bias, residual, y, kh, kw, stride, pad, act, w_layout, w_planes_f16, w_exp, x_absmax, y_absmax,
             out_scale, pool, y_full):
    B, H, W, Cin = x.shape
    ws = igemm_workspace(x.device)
    nw = 2 if w_planes_f16 is None else int(w_planes_f16.shape[0])
    return _lib.lib().dbmm_conv_bn_act_x2(ptr(x), ptr(x_absmax), ptr(w), ptr(w_planes_f16), nw, int(w_exp), ptr(out_scale),
                                          ptr(bias), ptr(residual), ptr(y), ptr(y_full), ptr(y_absmax), B, H, W, Cin,
                                          w.shape[0], kh,
                                          kw, stride, pad, act, pool, int(w_layout), ptr(ws), ws.numel() * 4, stream())


def gemm_dual(a, a_absmax, w_plane, w_exp, out_scale, a2, a2_absmax, w2_plane, ratio, bias, act=ACT_RELU, c_absmax=None):
    """c = act((a @ w^T) * out_scale + (a2 @ w2^T) * out_scale2 + bias) as one launch -- conv3 and the
    downsample branch of a bottleneck's first block (see dbmm_gemm_dual_bn_act_x2).  a [.., K],
    a2 [.., K2] with the same leading shape; planes [1][N][K] / [1][N][K2].  Returns None when the
    library has no kernel for the shape (the caller then runs the two convs)."""
    require_cuda(a, a2)
    _f32c(a); _f32c(a2)
    K, K2 = a.shape[-1], a2.shape[-1]
    M, N = a.numel() // K, w_plane.shape[1]
    if a2.numel() // K2 != M:
        raise _lib.DbmmError("gemm_dual: operand row counts differ")
    c = _empty(tuple(a.shape[:-1]) + (N,), device=a.device, dtype=torch.float32)
    ws = igemm_workspace(a.device)
    t = _Timed(M, N, K + K2, 0, 0, 4 * (M * K + M * K2 + M * N) + 2 * (N * K + N * K2))
    t.__enter__()
    rc = _lib.lib().dbmm_gemm_dual_bn_act_x2(ptr(a), K, ptr(a_absmax), ptr(w_plane), int(w_exp), K, K, ptr(out_scale),
                                             ptr(a2), K2, ptr(a2_absmax), ptr(w2_plane), K2, K2, ptr(ratio), ptr(bias),
                                             ptr(c), N, ptr(c_absmax), M, N, act, ptr(ws), ws.numel() * 4, stream())
    if rc == _lib.E_UNSUPPORTED:
        return None
    check(rc, "gemm_dual_bn_act_x2")
    t.__exit__(None, None, None)
    return c


def bottleneck_chain(y2, y2_absmax, c3, residual, c1, x_absmax=None, y1_absmax=None, pooled=False, keep_full=True):
    """x' = relu(bn3(conv3(y2)) + residual); y1' = relu(bn1'(conv1'(x'))) in ONE launch (dbmm_bottleneck_chain_x2);
    c3 / c1 = plan entries of the two 1x1 convs (single exact fp16 plane `ph`, `we`, `sc`, `b`).  y2 NHWC [B,H,W,K].
    Returns (x', y1') or (x', x'_pooled, y1') with pooled=True -- x' is None with keep_full=False (pooled only: the un-pooled
    tensor is not written); None when the library has no kernel for the shape (the caller then runs the two convs)."""
    require_cuda(y2, residual)
    _f32c(y2); _f32c(residual)
    B, H, W, K = y2.shape
    N, P = c3["ph"].shape[1], c1["ph"].shape[1]
    if c3["ph"].shape[0] != 1 or c1["ph"].shape[0] != 1 or tuple(residual.shape) != (B, H, W, N):
        return None
    chain8 = K == 256 and P == 256 and not pooled and get_option("chain8")       # the layer-3 geometry: eight-wave kernel
    if not chain8 and (K not in (64, 128) or P not in (64, 128)):                 # shapes the library serves
        return None
    if N % 64 or (pooled and (H % 2 or W % 2)):
        return None
    dev = y2.device
    full = keep_full or not pooled
    x = _empty((B, H, W, N), device=dev, dtype=torch.float32) if full else None
    y1 = _empty((B, H, W, P), device=dev, dtype=torch.float32)
    xp = _empty((B, H // 2, W // 2, N), device=dev, dtype=torch.float32) if pooled else None
    M = B * H * W
    # tagged like an igemm launch for bench.py's per-kernel table: FLOPs of both GEMMs, algorithmic bytes
    global _chain_tag
    _chain_tag = "bottleneck_chain8_kernel" if chain8 else f"bottleneck_chain_kernel<{K}, {P}, {int(pooled) + int(not full)}, 0>"
    t = _Timed(M, N, K + P, -1, 0, 4 * (M * K + (2 if full else 1) * M * N + M * P + (M // 4 * N if pooled else 0)) + 2 * (N * K + P * N))
    t.__enter__()
    rc = _lib.lib().dbmm_bottleneck_chain_x2(ptr(y2), ptr(y2_absmax), ptr(c3["ph"]), int(c3["we"]), ptr(c3["sc"]), ptr(c3["b"]),
                                            ptr(residual), ptr(x), ptr(xp), ptr(x_absmax), ptr(c1["ph"]), int(c1["we"]),
                                            ptr(c1["sc"]), ptr(c1["b"]), ptr(y1), ptr(y1_absmax), B, H, W, K, N, P, stream())
    if rc == _lib.E_UNSUPPORTED:
        return None
    check(rc, "bottleneck_chain_x2")
    t.__exit__(None, None, None)
    return (x, xp, y1) if pooled else (x, y1)


_chain_tag = None


def bottleneck_block_chain(y1, y1_absmax, c2, c3, c1, residual=None, dual=None, x_absmax=None, y1n_absmax=None):
    """one stride-1 bottleneck block from its conv1 output on, continued into the next block's conv1, ONE launch
    (dbmm_bottleneck_block_chain_x2): y1 -> conv2 3x3 -> conv3 + residual (or, dual = dict(a2, a2_absmax, ds, ratio, bias):
    + downsample branch) -> x' -> conv1' -> y1'.  c2 / c3 / c1 = plan entries (c2 in the chunk32-major K order).
    Returns (x', y1') or None when the library has no kernel for the shape."""
    require_cuda(y1)
    _f32c(y1)
    B, H, W, K = y1.shape
    N, P = c3["ph"].shape[1], c1["ph"].shape[1]
    M = B * H * W
    if (c2["ph"] is None or c2["ph"].shape[0] != 1 or c2["wl"] != WL_CHUNK32_MAJOR or tuple(c2["ph"].shape[1:]) != (K, 9 * K)
            or c3["ph"].shape[0] != 1 or c1["ph"].shape[0] != 1 or N % 64 or M % 4):
        return None
    if dual is None:
        if (K, P) not in ((64, 64), (128, 128)) or residual is None or tuple(residual.shape) != (B, H, W, N):
            return None
    elif (K, P) != (64, 64) or dual["ds"]["ph"].shape[0] != 1 or tuple(dual["a2"].shape) != (B, H, W, 64):
        return None
    x = _empty((B, H, W, N), device=y1.device, dtype=torch.float32)
    y1n = _empty((B, H, W, P), device=y1.device, dtype=torch.float32)
    global _chain_tag
    _chain_tag = f"bottleneck_chain_kernel<{K}, {P}, 0, {int(dual is not None)}, 1>"
    K2 = 64 if dual is not None else 0
    nby = 4 * (M * K + M * N * (1 if dual is not None else 2) + M * K2 + M * P) + 2 * (K * 9 * K + N * K + N * K2 + P * N)
    t = _Timed(M, 1, 9 * K * K + N * (K + K2 + P), -1, 0, nby)
    t.__enter__()
    d = dual or {}
    bias3 = d["bias"] if dual is not None else c3["b"]
    rc = _lib.lib().dbmm_bottleneck_block_chain_x2(
        ptr(y1), ptr(y1_absmax), ptr(c2["ph"]), int(c2["we"]), ptr(c2["sc"]), ptr(c2["b"]), ptr(c3["ph"]), int(c3["we"]),
        ptr(c3["sc"]), ptr(bias3), ptr(residual) if dual is None else None, ptr(d.get("a2")), ptr(d.get("a2_absmax")),
        ptr(d["ds"]["ph"]) if dual is not None else None, ptr(d.get("ratio")), ptr(x), ptr(x_absmax), ptr(c1["ph"]), int(c1["we"]),
        ptr(c1["sc"]), ptr(c1["b"]), ptr(y1n), ptr(y1n_absmax), B, H, W, K, N, P, stream())
    if rc == _lib.E_UNSUPPORTED:
        return None
    check(rc, "bottleneck_block_chain_x2")
    t.__exit__(None, None, None)
    return x, y1n


def bottleneck_chain_dual(y2, y2_absmax, c3, a2, a2_absmax, ds, ratio, bias, c1, x_absmax=None, y1_absmax=None):
    """first block of a stage at unchanged resolution: x' = relu(bn3(conv3(y2)) + bn_d(conv_d(a2))), then the next
    block's y1' = relu(bn1'(conv1'(x'))), one launch (dbmm_bottleneck_chain_dual_x2).  Returns (x', y1') or None."""
    require_cuda(y2, a2)
    _f32c(y2); _f32c(a2)
    K, K2 = y2.shape[-1], a2.shape[-1]
    N, P = c3["ph"].shape[1], c1["ph"].shape[1]
    M = y2.numel() // K
    if (c3["ph"].shape[0] != 1 or ds["ph"].shape[0] != 1 or c1["ph"].shape[0] != 1 or a2.numel() // K2 != M
            or K != 64 or K2 != 64 or N % 64 or P not in (64, 128) or M % 4):
        return None
    x = _empty(tuple(y2.shape[:-1]) + (N,), device=y2.device, dtype=torch.float32)
    y1 = _empty(tuple(y2.shape[:-1]) + (P,), device=y2.device, dtype=torch.float32)
    global _chain_tag
    _chain_tag = f"bottleneck_chain_kernel<{K}, {P}, 0, 1>"
    t = _Timed(M, N, K + K2 + P, -1, 0, 4 * (M * K + M * K2 + M * N + M * P) + 2 * (N * K + N * K2 + P * N))
    t.__enter__()
    rc = _lib.lib().dbmm_bottleneck_chain_dual_x2(ptr(y2), ptr(y2_absmax), ptr(c3["ph"]), int(c3["we"]), ptr(c3["sc"]), ptr(bias),
                                                 ptr(a2), ptr(a2_absmax), ptr(ds["ph"]), ptr(ratio), ptr(x), ptr(x_absmax),
                                                 ptr(c1["ph"]), int(c1["we"]), ptr(c1["sc"]), ptr(c1["b"]), ptr(y1),
                                                 ptr(y1_absmax), M, K, K2, N, P, stream())
    if rc == _lib.E_UNSUPPORTED:
        return None
    check(rc, "bottleneck_chain_dual_x2")
    t.__exit__(None, None, None)
    return x, y1


def conv_stem_s2(x_nchw, w, bias, y_absmax=None):
    require_cuda(x_nchw, w)
    _f32c(x_nchw); _f32c(w)
    B, C, H, W = x_nchw.shape
    if C != 3:
        raise _lib.DbmmError("stem conv expects 3 input channels")
    Cout = w.shape[-1]
    y = _empty((B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cout), device=x_nchw.device, dtype=torch.float32)
    check(_lib.lib().dbmm_conv_stem_s2(ptr(x_nchw), ptr(w), ptr(bias), ptr(y), ptr(y_absmax), B, H, W, Cout, stream()),
          "conv_stem_s2")
    return y


def avgpool2d(x, k):
    require_cuda(x)
    _f32c(x)
    B, H, W, C = x.shape
    y = _empty((B, H // k, W // k, C), device=x.device, dtype=torch.float32)
    check(_lib.lib().dbmm_avgpool2d(ptr(x), ptr(y), B, H, W, C, k, stream()), "avgpool2d")
    return y


def attnpool(x, pos, wq, bq, wkv, bkv, wc, bc, heads):
    """x NHWC [B,h,w,C] feature map, fp32 or (fp16 mode) fp16 -> fp32 [B, Dout]."""
    require_cuda(x)
    if x.dtype != torch.float16:
        _f32c(x)
    elif not x.is_contiguous():
        raise _lib.DbmmError("attnpool: expected a contiguous tensor")
    B, H, W, C = x.shape
    HW = H * W
    Dout = wc.shape[0]
    # the library sees raw pointers: the operand shapes are checked here.  A feature map of another resolution than the positional table
    # was built for fails like the reference's `x + self.positional_embedding[:, None, :]` (clip/model.py:72), not as a stray read
    if tuple(pos.shape) != (HW + 1, C):
        raise RuntimeError(f"attnpool: positional embedding {tuple(pos.shape)} does not match a {H} x {W} x {C} feature map "
                           f"(expected ({HW + 1}, {C})): the input resolution differs from the one the model was built for")
    if (tuple(wq.shape) != (C, C) or tuple(wkv.shape) != (2 * C, C) or wc.shape[1] != C or bq.numel() != C or bkv.numel() != 2 * C
            or bc.numel() != Dout):
        raise RuntimeError(f"attnpool: projection shapes {tuple(wq.shape)}, {tuple(wkv.shape)}, {tuple(wc.shape)} do not match C = {C}")
    for t in (pos, wq, bq, wkv, bkv, wc, bc):
        require_cuda(t)
        _f32c(t)
    nbytes = _lib.lib().dbmm_workspace_bytes_attnpool(B, HW, C)
    ws = _empty(nbytes // 4, device=x.device, dtype=torch.float32)
    out = _empty((B, Dout), device=x.device, dtype=torch.float32)
    check(_lib.lib().dbmm_attnpool_x(ptr(x), int(x.dtype == torch.float16), ptr(pos), ptr(wq), ptr(bq), ptr(wkv), ptr(bkv), ptr(wc), ptr(bc),
                                     ptr(out), B, HW, C, heads, Dout, ptr(ws), nbytes, stream()), "attnpool")
    return out


def layernorm(x, gamma, beta, rows=None, ldx=None, eps=1e-5, y_absmax=None):
    require_cuda(x)
    E = gamma.numel()
    if rows is None:
        rows = x.numel() // E
    if ldx is None:
        ldx = E
    _sized("layernorm beta", beta, E)
    if ldx < E or x.numel() < (rows - 1) * ldx + E:
        raise RuntimeError(f"layernorm: input {tuple(x.shape)} for {rows} rows of {E} at pitch {ldx}")
    y = _empty((rows, E), device=x.device, dtype=torch.float32)
    check(_lib.lib().dbmm_layernorm(ptr(x), ldx, ptr(gamma), ptr(beta), ptr(y), E, rows, E, eps, ptr(y_absmax), stream()),
          "layernorm")
    return y


def mha_core(qkv, B, L, E, heads, causal, qkv_absmax=None):
    """softmax(q k^T / 8) v per (image, head).  With qkv_absmax (device scalar >= max|qkv|, the qkv GEMM's c_absmax) the
    fp16-pair kernel runs (16-bit matrix cores, three partial products, fp32 accuracy); without it, or with
    option mha_x2 = 0, the fp32-input-MFMA kernel."""
    require_cuda(qkv)
    if qkv.numel() != B * L * 3 * E or E != heads * 64:
        raise RuntimeError(f"mha_core: qkv {tuple(qkv.shape)} for B = {B}, L = {L}, E = {E}, {heads} heads of 64")
    out = _empty((B * L, E), device=qkv.device, dtype=torch.float32)
    if qkv_absmax is not None and get_option("mha_x2"):
        with _TimedTag("mha_pair_kernel", 4.0 * B * heads * L * L * 64, 4 * (B * L * 4 * E)):
            check(_lib.lib().dbmm_mha_core_x2(ptr(qkv), ptr(qkv_absmax), ptr(out), B, L, E, heads, int(causal), stream()),
                  "mha_core_x2")
        return out
    with _TimedTag("mha_mfma_kernel", 4.0 * B * heads * L * L * 64, 4 * (B * L * 4 * E)):
        check(_lib.lib().dbmm_mha_core(ptr(qkv), ptr(out), B, L, E, heads, int(causal), stream()), "mha_core")
    return out


def embed_gather(tokens, table, pos):
    require_cuda(tokens, table)
    if tokens.dtype != torch.int32:
        tokens = tokens.to(torch.int32)
    tokens = tokens.contiguous()
    n, L = tokens.shape
    W = table.shape[1]
    if tuple(pos.shape) != (L, W):          # (clip/model.py:346: `x + self.positional_embedding` needs context_length tokens)
        raise RuntimeError(f"embed_gather: {L} tokens per prompt against a positional embedding of shape {tuple(pos.shape)}")
    out = _empty((n, L, W), device=table.device, dtype=torch.float32)
    check(_lib.lib().dbmm_embed_gather(ptr(tokens), ptr(table), ptr(pos), ptr(out), n, L, W, table.shape[0], stream()),
          "embed_gather")
    return out, tokens


def im2col_patch(x_nchw, P, out_absmax=None):
    """[B,3,R,R] -> patch rows [B*g*g, 3*P*P]; out_absmax (1-element device tensor, zeroed by the caller) receives max|x|"""
    require_cuda(x_nchw)
    _f32c(x_nchw)
    B, C, R, _ = x_nchw.shape
    g = R // P
    out = _empty((B * g * g, 3 * P * P), device=x_nchw.device, dtype=torch.float32)
    check(_lib.lib().dbmm_im2col_patch(ptr(x_nchw), ptr(out), ptr(out_absmax), B, R, P, stream()), "im2col_patch")
    return out


def _vit_tokens_check(patches, cls, pos, B):
    # raw pointers below: a patch count other than the positional table's (another input resolution) fails like the reference's
    # `x + self.positional_embedding` (clip/model.py:229), not as a stray read
    L, W = pos.shape
    if patches.numel() != B * (L - 1) * W or cls.numel() != W:
        raise RuntimeError(f"vit_tokens: {patches.numel() // max(1, B * W)} patches of width {W} per image do not match a positional embedding of "
                           f"{L} rows: the input resolution differs from the one the model was built for")


def vit_tokens(patches, cls, pos, B):
    L, W = pos.shape
    _vit_tokens_check(patches, cls, pos, B)
    out = _empty((B, L, W), device=patches.device, dtype=torch.float32)
    check(_lib.lib().dbmm_vit_tokens(ptr(patches), ptr(cls), ptr(pos), ptr(out), B, L, W, stream()), "vit_tokens")
    return out


def gather_eot(tokens_i32, x):
    n, L, W = x.shape
    out = _empty((n, W), device=x.device, dtype=torch.float32)
    check(_lib.lib().dbmm_gather_eot(ptr(tokens_i32), ptr(x), ptr(out), n, L, W, stream()), "gather_eot")
    return out


def text_colnorm(text):
    """[D, C] -> normalised, transposed [C, D]."""
    require_cuda(text)
    _f32c(text)
    D, C = text.shape
    tn = _empty((C, D), device=text.device, dtype=torch.float32)
    check(_lib.lib().dbmm_text_colnorm(ptr(text), ptr(tn), D, C, stream()), "text_colnorm")
    return tn


def l2norm_rows(x):
    """x / x.norm(dim=1, keepdim=True) for fp32 [B, D] (CLIP.forward, clip/model.py:362-363)"""
    require_cuda(x)
    _f32c(x)
    y = _empty(tuple(x.shape), device=x.device, dtype=torch.float32)
    check(_lib.lib().dbmm_l2norm_rows(ptr(x), ptr(y), x.shape[0], x.shape[1], stream()), "l2norm_rows")
    return y


def colsum(x):
    """x.sum(0) of fp32 [B, N] (N % 4 == 0) in a fixed order"""
    require_cuda(x)
    _f32c(x)
    out = _empty((x.shape[1],), device=x.device, dtype=torch.float32)
    check(_lib.lib().dbmm_colsum(ptr(x), ptr(out), x.shape[0], x.shape[1], stream()), "colsum")
    return out


def l2norm_sim_ce_fwd(z, tn, temperature, labels=None, z_old=None, ebd_weight=0.5, want_loss=True, want_pred=False):
    require_cuda(z, tn)
    _f32c(z)
    B, D = z.shape
    C = tn.shape[0]
    dev = z.device
    logits = _empty((B, C), device=dev, dtype=torch.float32)
    inv_norm = _empty((B,), device=dev, dtype=torch.float32)
    loss_rows = loss_mean = pred = None
    if labels is not None and want_loss:
        loss_rows = _empty((B,), device=dev, dtype=torch.float32)
        loss_mean = _empty((), device=dev, dtype=torch.float32)
    if want_pred:
        pred = _empty((B,), device=dev, dtype=torch.int64)
    check(_lib.lib().dbmm_l2norm_sim_ce_fwd(ptr(z), ptr(z_old), float(ebd_weight), ptr(tn), ptr(labels),
                                            float(temperature), ptr(logits), ptr(loss_rows), ptr(loss_mean), ptr(pred),
                                            ptr(inv_norm), B, D, C, stream()), "l2norm_sim_ce_fwd")
    return logits, loss_rows, loss_mean, pred, inv_norm


def l2norm_sim_ce_bwd(z, inv_norm, tn, temperature, logits=None, labels=None, dlogits=None, blended=False,
                      ebd_weight=0.5, grad_scale=1.0):
    B, D = z.shape
    C = tn.shape[0]
    dz = _empty(tuple(z.shape), device=z.device, dtype=z.dtype)
    check(_lib.lib().dbmm_l2norm_sim_ce_bwd(ptr(z), ptr(inv_norm), float(ebd_weight), int(blended), ptr(tn), ptr(logits),
                                            ptr(labels), ptr(dlogits), float(temperature), float(grad_scale), ptr(dz),
                                            B, D, C, stream()), "l2norm_sim_ce_bwd")
    return dz


def adapter_fwd(x, w1, b1, gamma, beta, running_mean, running_var, nbt, w2, b2, train, eps=1e-5, momentum=0.1):
    require_cuda(x, w1)
    _f32c(x)
    B, D = x.shape
    H = w1.shape[0]
    dev = x.device
    h = _empty((B, H), device=dev, dtype=torch.float32)
    r = _empty((B, H), device=dev, dtype=torch.float32)
    z = _empty((B, D), device=dev, dtype=torch.float32)
    mean = _empty((H,), device=dev, dtype=torch.float32) if train else None
    invstd = _empty((H,), device=dev, dtype=torch.float32) if train else None
    check(_lib.lib().dbmm_adapter_fwd(ptr(x), ptr(w1), ptr(b1), ptr(gamma), ptr(beta), ptr(running_mean),
                                      ptr(running_var), ptr(nbt), ptr(w2), ptr(b2), ptr(h), ptr(mean), ptr(invstd),
                                      ptr(r), ptr(z), B, D, H, int(train), eps, momentum, stream()), "adapter_fwd")
    return z, h, mean, invstd, r


def adapter_bwd(x, dz, h, mean, invstd, r, gamma, beta, w2):
    B, D = x.shape
    H = h.shape[1]
    dev = x.device
    f = dict(device=dev, dtype=torch.float32)
    dw1, db1 = _empty((H, D), **f), _empty((H,), **f)
    dgamma, dbeta = _empty((H,), **f), _empty((H,), **f)
    dw2, db2 = _empty((D, H), **f), _empty((D,), **f)
    nbytes = _lib.lib().dbmm_workspace_bytes_adapter_bwd(B, D, H)
    ws = _empty(nbytes // 4, **f)
    check(_lib.lib().dbmm_adapter_bwd(ptr(x), ptr(dz), ptr(h), ptr(mean), ptr(invstd), ptr(r), ptr(gamma), ptr(beta),
                                      ptr(w2), ptr(dw1), ptr(db1), ptr(dgamma), ptr(dbeta), ptr(dw2), ptr(db2), B, D, H,
                                      ptr(ws), nbytes, stream()), "adapter_bwd")
    return dw1, db1, dgamma, dbeta, dw2, db2, ws[B * H:2 * B * H].view(B, H)   # last = dh (for an optional dx)


def sgd_momentum(params, grads, bufs, lr, momentum, weight_decay, first_step):
    n = len(params)
    for i in range(0, n, 16):
        ps, gs, bs = params[i:i + 16], grads[i:i + 16], bufs[i:i + 16]
        m = len(ps)
        P = (ctypes.c_void_p * m)(*[p.data_ptr() for p in ps])
        G = (ctypes.c_void_p * m)(*[g.data_ptr() for g in gs])
        Bf = (ctypes.c_void_p * m)(*[b.data_ptr() for b in bs])
        S = (ctypes.c_int64 * m)(*[p.numel() for p in ps])
        check(_lib.lib().dbmm_sgd_momentum(m, P, G, Bf, S, float(lr), float(momentum), float(weight_decay),
                                           int(first_step), stream()), "sgd_momentum")


def group_count(logits, y, g, counts):
    """counts int64 [G,2] accumulated in place: (n, correct) per group."""
    require_cuda(logits, y, g, counts)
    B, C = logits.shape
    check(_lib.lib().dbmm_group_count(ptr(logits), ptr(y), ptr(g), ptr(counts), B, C, counts.shape[0], stream()),
          "group_count")
    return counts


def group_loss_sum(loss_rows, g, sums):
    check(_lib.lib().dbmm_group_loss_sum(ptr(loss_rows), ptr(g), ptr(sums), loss_rows.shape[0], sums.shape[0], stream()),
          "group_loss_sum")
    return sums


_step_ws = {}


def adapter_step_launches(B, D, H, with_old=False):
    """kernel launches of one dbmm_adapter_train_step call on the purpose-built kernels (csrc/adapter_step.hip): forward 3
    (K-split fc1, BatchNorm statistics, BatchNorm + ReLU + fc2; 3 more for a frozen old adapter), cosine logits + CE forward
    and backward in one, backward 3 (dW2 / dr + the loss mean, BatchNorm backward + dW2 sums, dW1), SGD (+ dW1 sums); None for
    shapes that take the general GEMM kernel (H != 128 or D % 128 != 0)"""
    if H != 128 or D % 128 or not get_option("adapter_step_fused"):
        return None
    return 8 + (3 if with_old else 0)


def adapter_step_args(new, bufs, old):
    """the 24 parameter / buffer pointers of dbmm_adapter_train_step, converted once: `new` / `old` are tuples
    (w1, b1, gamma, beta, running_mean, running_var, nbt, w2, b2); `bufs` the six momentum buffers in the order
    (w1, b1, gamma, beta, w2, b2); `old` may be None.  Valid while those tensors keep their storage."""
    require_cuda(*new)
    return tuple(ptr(t) for t in list(new) + list(bufs)) + (tuple(ptr(t) for t in old) if old is not None else (None,) * 9)


def adapter_train_step(x, labels, args, H, with_old, ebd_weight, tn, temperature, lr, momentum, weight_decay, first_step):
    """one fused training-step body (dbmm_adapter_train_step); `args` from adapter_step_args()."""
    if not (x.is_cuda and labels.is_cuda and tn.is_cuda):
        require_cuda(x, labels, tn)
    _f32c(x)
    B, D = x.shape
    C = tn.shape[0]
    dev = x.device
    key = (dev.index, B, D, H, with_old)
    ws = _step_ws.get(key)
    if ws is None:
        nbytes = _lib.lib().dbmm_workspace_bytes_adapter_train_step(B, D, H, int(with_old))
        ws = _empty(nbytes // 4, device=dev, dtype=torch.float32)
        if len(_step_ws) > 8:
            _step_ws.clear()
        _step_ws[key] = ws
    logits = _empty((B, C), device=dev, dtype=torch.float32)
    loss = _empty((B + 1,), device=dev, dtype=torch.float32)        # per-row losses, then their mean: one allocation
    rc = _lib.lib().dbmm_adapter_train_step(
        x.data_ptr(), labels.data_ptr(), *args, float(ebd_weight), tn.data_ptr(), float(temperature), float(lr), float(momentum),
        float(weight_decay), int(first_step), logits.data_ptr(), loss.data_ptr(), loss.data_ptr() + 4 * B, B, D, H, C, ws.data_ptr(),
        ws.numel() * 4, stream())
    if rc:
        check(rc, "adapter_train_step")
    return loss[B], logits, loss[:B]


def gather_rows(table, idx):
    """out[i] = table[idx[i]] for a device-resident [N, D] fp32 table and int64 indices."""
    require_cuda(table, idx)
    _f32c(table)
    idx = idx.contiguous()
    out = _empty((idx.numel(), table.shape[1]), device=table.device, dtype=torch.float32)
    check(_lib.lib().dbmm_gather_rows(ptr(table), ptr(idx), ptr(out), table.shape[0], idx.numel(), table.shape[1],
                                      stream()), "gather_rows")
    return out


# ---- fp16 mode of the transformer towers (csrc/f16_ops.hip) -------------------------------------------------------

def _f16c(t):
    if t.dtype != torch.float16 or not t.is_contiguous():
        raise _lib.DbmmError(f"expected a contiguous float16 tensor, got {t.dtype} contiguous={t.is_contiguous()}")
    return t


def _gemm_f16_tag(M, N):
    """instantiation dbmm_gemm_f16 picks for a shape the deep-pipelined kernel does not take (its own rule), as rocprofv3 prints it"""
    wide = get_option("f16_bn256") and N >= 768 and N % 256 == 0 and M >= 32768
    return "gemm_f16_kernel<32, 256, 2>" if wide else "gemm_f16_kernel<64, 128, 2>"


def gemm_f16(a, w, bias=None, residual=None, act=ACT_NONE, M=None, lda=None):
    """c f16 = act(a @ w^T + bias) + residual; a f16 [M][K] (row pitch lda), w f16 [N][K], bias f32, residual f16."""
    require_cuda(a, w)
    _f16c(w)
    if a.dtype != torch.float16:
        raise _lib.DbmmError("gemm_f16 needs float16 activations")
    N, K = w.shape
    if lda is None:
        _f16c(a)
        lda = a.shape[-1]
    if M is None:
        M = a.numel() // a.shape[-1]
    if a.numel() < (M - 1) * lda + K or lda < K:
        raise RuntimeError(f"gemm_f16: activations {tuple(a.shape)} (row pitch {lda}) for M = {M}, K = {K}")
    _sized("gemm_f16 bias", bias, N); _sized("gemm_f16 residual", residual, M * N)
    c = _empty((M, N), device=a.device, dtype=torch.float16)
    deep = N % 256 == 0 and K % 128 == 0 and M >= 16384 and get_option("f16_8ph")   # dbmm_gemm_f16's own rule
    with _TimedTag(f"gemm_f16_8ph_kernel<{act}, {int(residual is not None)}>" if deep else _gemm_f16_tag(M, N), 2.0 * M * N * K,
                   2 * (M * K + N * K + M * N * (2 if residual is not None else 1))):
        ws = igemm_workspace(a.device)
        check(_lib.lib().dbmm_gemm_f16_ws(ptr(a), lda, ptr(w), K, ptr(bias), ptr(residual), N if residual is not None else 0, ptr(c), N,
                                          M, N, K, act, ptr(ws), ws.numel() * 4, stream()), "gemm_f16")
    return c


def mha_core_f16(qkv, B, L, E, heads, causal):
    require_cuda(qkv)
    _f16c(qkv)
    if qkv.numel() != B * L * 3 * E or E != heads * 64:
        raise RuntimeError(f"mha_core_f16: qkv {tuple(qkv.shape)} for B = {B}, L = {L}, E = {E}, {heads} heads of 64")
    out = _empty((B * L, E), device=qkv.device, dtype=torch.float16)
    with _TimedTag("mha_f16_kernel", 4.0 * B * heads * L * L * 64, 2 * (B * L * 4 * E)):
        check(_lib.lib().dbmm_mha_core_f16(ptr(qkv), ptr(out), B, L, E, heads, int(causal), stream()), "mha_core_f16")
    return out


def layernorm_f16(x, gamma, beta, rows=None, ldx=None, eps=1e-5):
    require_cuda(x)
    E = gamma.numel()
    if rows is None:
        rows = x.numel() // E
    if ldx is None:
        ldx = E
    _sized("layernorm_f16 beta", beta, E)
    if ldx < E or x.numel() < (rows - 1) * ldx + E:
        raise RuntimeError(f"layernorm_f16: input {tuple(x.shape)} for {rows} rows of {E} at pitch {ldx}")
    y = _empty((rows, E), device=x.device, dtype=torch.float16)
    check(_lib.lib().dbmm_layernorm_f16(ptr(x), ldx, ptr(gamma), ptr(beta), ptr(y), E, rows, E, eps, stream()), "layernorm_f16")
    return y


def im2col_patch_f16(x_nchw, P, Kp):
    require_cuda(x_nchw)
    if x_nchw.dtype not in (torch.float16, torch.float32) or not x_nchw.is_contiguous():
        raise _lib.DbmmError("im2col_patch_f16 needs a contiguous float16 / float32 image batch")
    B, C, R, _ = x_nchw.shape
    g = R // P
    out = _empty((B * g * g, Kp), device=x_nchw.device, dtype=torch.float16)
    check(_lib.lib().dbmm_im2col_patch_f16(ptr(x_nchw), int(x_nchw.dtype == torch.float16), ptr(out), B, R, P, Kp, stream()),
          "im2col_patch_f16")
    return out


def vit_tokens_f16(patches, cls, pos, B):
    L, W = pos.shape
    _vit_tokens_check(patches, cls, pos, B)
    out = _empty((B, L, W), device=patches.device, dtype=torch.float16)
    check(_lib.lib().dbmm_vit_tokens_f16(ptr(patches), ptr(cls), ptr(pos), ptr(out), B, L, W, stream()), "vit_tokens_f16")
    return out


def embed_gather_f16(tokens, table, pos):
    require_cuda(tokens, table)
    if tokens.dtype != torch.int32:
        tokens = tokens.to(torch.int32)
    tokens = tokens.contiguous()
    n, L = tokens.shape
    W = table.shape[1]
    if tuple(pos.shape) != (L, W):
        raise RuntimeError(f"embed_gather_f16: {L} tokens per prompt against a positional embedding of shape {tuple(pos.shape)}")
    out = _empty((n, L, W), device=table.device, dtype=torch.float16)
    check(_lib.lib().dbmm_embed_gather_f16(ptr(tokens), ptr(table), ptr(pos), ptr(out), n, L, W, table.shape[0], stream()),
          "embed_gather_f16")
    return out, tokens


def gather_eot_f16(tokens_i32, x):
    n, L, W = x.shape
    out = _empty((n, W), device=x.device, dtype=torch.float16)
    check(_lib.lib().dbmm_gather_eot_f16(ptr(tokens_i32), ptr(x), ptr(out), n, L, W, stream()), "gather_eot_f16")
    return out


# ---- fp16 mode of the ModifiedResNet towers (csrc/conv_f16.hip, csrc/f16_ops.hip) ------------------------------------------

def conv1x1_f16(x, w, scale, bias, residual=None, act=ACT_RELU):
    """y f16 NHWC = act(conv1x1(x) * scale + bias + residual); x f16 [..., Cin], w f16 [Cout][Cin]; None when the library has no
    kernel for the shape (Cin % 64, Cout % 8)."""
    require_cuda(x, w)
    _f16c(x); _f16c(w)
    Cout, Cin = w.shape
    if x.shape[-1] != Cin:
        raise RuntimeError(f"conv1x1_f16: {x.shape[-1]} input channels against a weight {tuple(w.shape)}")
    M = x.numel() // Cin
    _sized("conv1x1_f16 scale", scale, Cout); _sized("conv1x1_f16 bias", bias, Cout); _sized("conv1x1_f16 residual", residual, M * Cout)
    y = _empty(tuple(x.shape[:-1]) + (Cout,), device=x.device, dtype=torch.float16)
    deep = Cout % 256 == 0 and Cin % 128 == 0 and M >= 16384                            # what dbmm_gemm_f16 gives the eight-phase kernel
    mode = get_option("conv1x1_stream")
    # dbmm_conv1x1_bn_act_f16's own routing rule: the streaming kernel where the GEMM is not deep-pipelined, and for conv3 + residual with K <= 256
    stream_k = ((mode == 2 or (mode == 1 and not (deep and not (residual is not None and Cin <= 256))) or (mode == 3 and not deep))
                and act in (ACT_NONE, ACT_RELU) and Cin % 32 == 0)           # (mode 3: round 3's rule, without the conv3 + residual exception)
    res = int(residual is not None)
    tag = (f"conv1x1_f16_kernel<{'4, 1, 2' if Cout <= 64 else '2, 2, 4'}, {res}>" if stream_k
           else (f"gemm_f16_8ph_kernel<{act}, {res}>" if deep and get_option("f16_8ph") else _gemm_f16_tag(M, Cout)))
    if residual is not None and act == ACT_RELU and Cin == 256 and Cout >= 1024 and Cout % 64 == 0 and M >= 131072 and get_option("conv1x1_res_stream"):
        tag = "conv1x1_res_stream_f16_kernel<256, 0>"            # layer 3's conv3 + residual: the row-owning streaming kernel
    t = _TimedTag(tag, 2.0 * M * Cout * Cin,
                  2 * (M * Cin + Cout * Cin + M * Cout * (2 if residual is not None else 1)))
    t.__enter__()
    ws = igemm_workspace(x.device)
    rc = _lib.lib().dbmm_conv1x1_bn_act_f16_ws(ptr(x), ptr(w), ptr(scale), ptr(bias), ptr(residual), ptr(y), M, Cin, Cout, act, ptr(ws),
                                               ws.numel() * 4, stream())
    t.__exit__(None if rc == 0 else DbmmUnsupported, None, None)      # nothing is recorded for a launch that did not happen
    if rc == _lib.E_UNSUPPORTED:
        return None
    check(rc, "conv1x1_bn_act_f16")
    return y


def conv1x1_res_pool_f16(x, c3, residual):
    """fp16 mode, the last block of a stage: (y, AvgPool2d(2) of y) with y = relu(conv1x1(x) * scale + bias + residual) in one launch
    (dbmm_conv1x1_res_pool_f16); x NHWC f16 [B,H,W,128 | 256]; None when the library has no kernel for the shape."""
    require_cuda(x, residual)
    _f16c(x); _f16c(residual)
    w, scale, bias = c3
    Cout, Cin = w.shape
    if x.dim() != 4 or x.shape[1] % 2 or x.shape[2] % 2:
        return None
    B, H, W = x.shape[:3]
    M = B * H * W
    y = _empty((B, H, W, Cout), device=x.device, dtype=torch.float16)
    yp = _empty((B, H // 2, W // 2, Cout), device=x.device, dtype=torch.float16)
    t = _TimedTag(f"conv1x1_res_stream_f16_kernel<{Cin}, 1>", 2.0 * M * Cout * Cin, 2 * (M * Cin + Cout * Cin + 2 * M * Cout + M // 4 * Cout))
    t.__enter__()
    rc = _lib.lib().dbmm_conv1x1_res_pool_f16(ptr(x), ptr(w), ptr(scale), ptr(bias), ptr(residual), ptr(y), ptr(yp), B, H, W, Cin, Cout, stream())
    t.__exit__(None if rc == 0 else DbmmUnsupported, None, None)
    if rc == _lib.E_UNSUPPORTED:
        return None
    check(rc, "conv1x1_res_pool_f16")
    return y, yp


def chain_f16(y2, c3, residual, c1, pooled=False, keep_full=True):
    """fp16 mode: x' = relu(conv3(y2) * s3 + b3 + residual) and y1' = relu(conv1'(x') * s1 + b1) in one launch (dbmm_bottleneck_chain_f16);
    c3 / c1 = (w f16, scale f32, bias f32).  Returns (x', y1'), or with pooled=True (y2 NHWC [B,H,W,K], a stage seam) (x', AvgPool2d(2) of x',
    y1') where x' is None with keep_full=False (not written); None when the library has no kernel for the shape."""
    require_cuda(y2, residual)
    _f16c(y2); _f16c(residual)
    (w3, s3, b3), (w1, s1, b1) = c3, c1
    N, K = w3.shape
    P = w1.shape[0]
    M = y2.numel() // K
    full = keep_full or not pooled
    if pooled and (y2.dim() != 4 or y2.shape[1] % 2 or y2.shape[2] % 2):
        return None
    x = _empty(tuple(y2.shape[:-1]) + (N,), device=y2.device, dtype=torch.float16) if full else None
    y1 = _empty(tuple(y2.shape[:-1]) + (P,), device=y2.device, dtype=torch.float16)
    xp = _empty((y2.shape[0], y2.shape[1] // 2, y2.shape[2] // 2, N), device=y2.device, dtype=torch.float16) if pooled else None
    tag = f"chain_f16_kernel<{K}, {P}, 0, {int(pooled) + int(not full)}>" if pooled else f"chain_f16_kernel<{K}, {P}>"
    t = _TimedTag(tag, 2.0 * M * N * (K + P), 2 * (M * (K + (2 if full else 1) * N + P) + (M // 4 * N if pooled else 0) + N * (K + P)))
    t.__enter__()
    if pooled:
        B, H, W = y2.shape[:3]
        rc = _lib.lib().dbmm_bottleneck_chain_pool_f16(ptr(y2), ptr(w3), ptr(s3), ptr(b3), ptr(residual), ptr(x), ptr(xp), ptr(w1), ptr(s1), ptr(b1),
                                                       ptr(y1), B, H, W, K, N, P, stream())
    else:
        rc = _lib.lib().dbmm_bottleneck_chain_f16(ptr(y2), ptr(w3), ptr(s3), ptr(b3), ptr(residual), ptr(x), ptr(w1), ptr(s1), ptr(b1), ptr(y1),
                                                  M, K, N, P, stream())
    t.__exit__(None if rc == 0 else DbmmUnsupported, None, None)
    if rc == _lib.E_UNSUPPORTED:
        return None
    check(rc, "bottleneck_chain_f16")
    return (x, xp, y1) if pooled else (x, y1)


def chain_dual_f16(y2, w3, scale3, xp, wd, ratio, bias, c1):
    """fp16 mode: the first block of a stage -- x' = relu(conv3(y2) * scale3 + conv_d(xp) * scale_d + bias) (conv1x1_dual_f16's arguments) and
    y1' = relu(conv1'(x') * s1 + b1) in one launch (dbmm_bottleneck_chain_dual_f16).  Returns (x', y1') or None."""
    require_cuda(y2, xp)
    _f16c(y2); _f16c(xp); _f16c(w3); _f16c(wd)
    w1, s1, b1 = c1
    N, K = w3.shape
    K2 = wd.shape[1]
    P = w1.shape[0]
    M = y2.numel() // K
    x = _empty(tuple(y2.shape[:-1]) + (N,), device=y2.device, dtype=torch.float16)
    y1 = _empty(tuple(y2.shape[:-1]) + (P,), device=y2.device, dtype=torch.float16)
    t = _TimedTag(f"chain_f16_kernel<{K}, {P}, 1>", 2.0 * M * N * (K + K2 + P), 2 * (M * (K + K2 + N + P) + N * (K + K2 + P)))
    t.__enter__()
    rc = _lib.lib().dbmm_bottleneck_chain_dual_f16(ptr(y2), ptr(w3), ptr(scale3), ptr(xp), ptr(wd), ptr(ratio), ptr(bias), ptr(x), ptr(w1), ptr(s1),
                                                   ptr(b1), ptr(y1), M, K, K2, N, P, stream())
    t.__exit__(None if rc == 0 else DbmmUnsupported, None, None)
    if rc == _lib.E_UNSUPPORTED:
        return None
    check(rc, "bottleneck_chain_dual_f16")
    return x, y1


def conv1x1_dual_f16(y2, w3, scale3, xp, wd, ratio, bias, act=ACT_RELU):
    """fp16 mode: act(conv3(y2) * scale3 + conv_d(xp) * scale_d + bias) as one dual-source GEMM (dbmm_conv1x1_dual_bn_act_f16);
    ratio = scale_d / scale3, bias = both BatchNorm biases.  None when the library has no kernel for the shape."""
    require_cuda(y2, xp)
    _f16c(y2); _f16c(xp); _f16c(w3); _f16c(wd)
    Cout, K = w3.shape
    K2 = wd.shape[1]
    M = y2.numel() // K
    out = _empty(tuple(y2.shape[:-1]) + (Cout,), device=y2.device, dtype=torch.float16)
    t = _TimedTag("gemm_f16_8ph_kernel<dual>", 2.0 * M * Cout * (K + K2), 2 * (M * (K + K2) + Cout * (K + K2) + M * Cout))
    t.__enter__()
    rc = _lib.lib().dbmm_conv1x1_dual_bn_act_f16(ptr(y2), ptr(w3), ptr(scale3), ptr(xp), ptr(wd), ptr(ratio), ptr(bias), ptr(out), M, K, K2, Cout,
                                                 act, stream())
    t.__exit__(None if rc == 0 else DbmmUnsupported, None, None)
    if rc == _lib.E_UNSUPPORTED:
        return None
    check(rc, "conv1x1_dual_bn_act_f16")
    return out


def conv3x3_f16(x, w, scale, bias, pool=1):
    """y f16 NHWC = [AvgPool2d(2)] relu(conv3x3(x, stride 1, pad 1) * scale + bias); x f16 [B,H,W,Cin], w f16 [Cout][(cin/32, kh, kw, 32)];
    None when the library has no kernel for the shape."""
    require_cuda(x, w)
    _f16c(x); _f16c(w)
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    if w.numel() != Cout * 9 * Cin:
        raise RuntimeError(f"conv3x3_f16: packed weight {tuple(w.shape)} for a 3 x 3 conv over {Cin} channels")
    _sized("conv3x3_f16 scale", scale, Cout); _sized("conv3x3_f16 bias", bias, Cout)
    y = _empty((B, H // pool, W // pool, Cout), device=x.device, dtype=torch.float16)
    geo = "2, 2, 2" if Cout > 64 else ("4, 1, 2" if Cout > 32 else "4, 1, 1")
    deep = Cout % 256 == 0 and B * H * W >= 16384 and get_option("f16_conv_8ph")     # dbmm_conv3x3_bn_relu_f16's own routing rule
    tag = f"conv3x3_f16_8ph_kernel<{int(pool == 2)}>" if deep else f"conv3x3_f16_kernel<{geo}, {int(pool == 2)}>"
    if Cin == 32 and Cout in (32, 64) and H % 4 == 0 and W % 28 == 0 and get_option("conv_patch"):      # the stem convs: persistent patch kernel
        tag = f"conv3x3_c32_f16_kernel<{Cout}, {int(pool == 2)}>"
    t = _TimedTag(tag, 2.0 * B * H * W * Cout * 9 * Cin, 2 * (x.numel() + y.numel() + w.numel()))
    t.__enter__()
    rc = _lib.lib().dbmm_conv3x3_bn_relu_f16(ptr(x), ptr(w), ptr(scale), ptr(bias), ptr(y), B, H, W, Cin, Cout, 2 if pool == 2 else 0, stream())
    t.__exit__(None if rc == 0 else DbmmUnsupported, None, None)
    if rc == _lib.E_UNSUPPORTED:
        return None
    check(rc, "conv3x3_bn_relu_f16")
    return y


def conv_stem_s2_f16(x_nchw, w, bias, scale=None):
    """stem conv1 (3x3, stride 2, BatchNorm, ReLU): NCHW f32 / f16 image -> f16 NHWC.  w fp32 [kh][kw][cin][cout].  Without `scale`: w carries
    the folded BatchNorm and is used as it is (FMA kernel); with it: w = the model's fp16 conv weights, BatchNorm = scale / bias on the fp32
    accumulator (dbmm_conv_stem_s2_bn_f16, MFMA gather kernel)."""
    require_cuda(x_nchw, w)
    if x_nchw.dtype not in (torch.float16, torch.float32) or not x_nchw.is_contiguous():
        raise _lib.DbmmError("conv_stem_s2_f16 needs a contiguous float16 / float32 image batch")
    B, C, H, W = x_nchw.shape
    if C != 3:
        raise _lib.DbmmError("stem conv expects 3 input channels")
    Cout = w.shape[-1]
    y = _empty((B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cout), device=x_nchw.device, dtype=torch.float16)
    mfma = scale is not None and get_option("stem_mfma")
    with _TimedTag("stem_s2_f16_mfma_kernel" if mfma else "stem_s2_f16_kernel", 2.0 * y.numel() * 27,
                   x_nchw.numel() * x_nchw.element_size() + 2 * y.numel()):
        if scale is None:
            check(_lib.lib().dbmm_conv_stem_s2_f16(ptr(x_nchw), int(x_nchw.dtype == torch.float16), ptr(w), ptr(bias), ptr(y), B, H, W, Cout,
                                                   stream()), "conv_stem_s2_f16")
        else:
            check(_lib.lib().dbmm_conv_stem_s2_bn_f16(ptr(x_nchw), int(x_nchw.dtype == torch.float16), ptr(w), ptr(scale), ptr(bias), ptr(y), B, H, W,
                                                      Cout, stream()), "conv_stem_s2_bn_f16")
    return y


def avgpool2_f16(x):
    require_cuda(x)
    _f16c(x)
    B, H, W, C = x.shape
    y = _empty((B, H // 2, W // 2, C), device=x.device, dtype=torch.float16)
    with _TimedTag("avgpool2_f16_kernel", 0.0, 2 * (x.numel() + y.numel())):
        check(_lib.lib().dbmm_avgpool2_f16(ptr(x), ptr(y), B, H, W, C, stream()), "avgpool2_f16")
    return y


_TRACE_LAUNCHES = os.environ.get("DBMM_TRACE_LAUNCHES")


class _TimedTag:
    """profile hook for kernels outside the igemm family: fixed tag, caller-supplied FLOPs and algorithmic bytes"""
    def __init__(self, tag, flops, nbytes):
        self.tag, self.flops, self.nbytes = tag, float(flops), float(nbytes)

    def __enter__(self):
        if _TRACE_LAUNCHES:                 # developer aid (DBMM_TRACE_LAUNCHES=<file>): tag before the launch, "ok" after a device sync
            with open(_TRACE_LAUNCHES, "a") as f:
                f.write(self.tag + " ... ")
        self.on = _prof is not None and not _prof_paused and not _prof_conv_only
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if _TRACE_LAUNCHES:
            torch.cuda.synchronize()
            with open(_TRACE_LAUNCHES, "a") as f:
                f.write("ok\n")
        if self.on and _prof is not None and exc[0] is None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            _prof.append((self.tag, self.flops, self.nbytes, self.e0, e1))
