"""Correctness at the size the metric is quoted on (CLIP-RN50 224 px, bs = 1024 on one GPU; reference loop
clip_inference.py:203-216 over clip/model.py:42-55, 138-154) and memory safety of the branch-free epilogues.

  * the whole RN50 plan on 1024 images in ONE call: the golden's images sit at rows 0 and 1023, fresh images around row 668
    (where the 3.29 GB layer-1 maps cross 2 GiB); rows must equal the reference-generated golden and the same images
    encoded in a small batch, and the kernels that serve this batch (chain / block chain / patch / pooled halo) must
    have been the ones that ran;
  * the chain kernels on operands past 2 GiB against fp64 (first / straddling / last images);
  * guard zones: every output of a kernel whose epilogue leaves rows >= M to the buffer range check is allocated between
    sentinel-filled zones; ragged tiles (M % tile = 5, 77) must leave them untouched.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import relerr
from dbmm_amd import ops, synth
from dbmm_amd.clip.model import build_model

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_rn50_bs1024_one_call_rows_vs_golden_and_small_batch(golden):
    g = golden("clip_RN50.npz")
    seed = int(g["seed"])
    model = build_model(synth.clip_state_dict(seed, "RN50")).cuda()
    gold_img = synth.images(seed + 100, int(g["batch"]), 224)            # the fixture's two images
    fresh = synth.images(4242, 3, 224)
    base = synth.images(977, 64, 224)
    B = 1024
    scale = torch.linspace(0.7, 1.3, B // 64).repeat_interleave(64).view(-1, 1, 1, 1)
    img = (base.repeat(B // 64, 1, 1, 1) * scale).contiguous()
    img[0], img[B - 1] = gold_img[0], gold_img[1]
    img[667:670] = fresh                                                 # image 668 straddles byte 2^31 of a [B,56,56,256] fp32 map
    assert 668 * 56 * 56 * 256 * 4 < 2 ** 31 < 669 * 56 * 56 * 256 * 4
    img = img.to(DEV)
    ops.profile_begin()
    out = model.encode_image(img)
    prof = ops.profile_end()
    assert tuple(out.shape) == (B, 1024) and torch.isfinite(out).all()
    ref = torch.from_numpy(g["embedding"])
    e0, e1 = relerr(out[0:1].cpu(), ref[0:1]), relerr(out[B - 1:B].cpu(), ref[1:2])
    assert e0 < 5e-5 and e1 < 5e-5, f"rows 0 / 1023 of the bs=1024 call vs the reference golden: {e0:.2e} / {e1:.2e}"
    small = model.encode_image(img[667:670].contiguous())
    es = relerr(out[667:670].cpu(), small.cpu())
    assert es < 1e-5, f"rows 667..669 of the bs=1024 call vs the same images in a batch of 3: {es:.2e}"
    eg = relerr(model.encode_image(torch.stack([img[0], img[B - 1]])).cpu(), ref)
    assert eg < 5e-5
    # the launches that make the headline number were the ones checked
    tags = set(prof)
    for want in ("bottleneck_chain_kernel<64, 64, 0, 1, 1>", "bottleneck_chain_kernel<64, 64, 0, 0, 1>", "bottleneck_chain_kernel<64, 128, 2, 0>",
                 "bottleneck_chain_kernel<128, 128, 0, 0, 1>", "conv3x3_c32_kernel<32, 0>", "conv3x3_c32_kernel<64, 1>"):
        assert want in tags, (want, sorted(tags))
    assert "conv3x3_halo8n_kernel<1>" in tags, sorted(tags)                                                # layer 2's pooled 3x3
    assert "conv3x3_halo8_kernel<0>" in tags and "conv3x3_halo8_kernel<1>" in tags, sorted(tags)           # layers 3 / 4: eight-phase halo
    assert not any(t.startswith("igemm_f32_kernel<") and prof[t][1] > 1e12 for t in tags), sorted(tags)    # no fp32 fallback on a conv


def test_chain8_on_operands_over_2gib(option):
    """layer-3 geometry: x / x' are [B,14,14,1024] fp32 = 802,816 B per image, > 2 GiB from B = 2675; the eight-wave chain at B = 2800:
    first image, the images around the 2 GiB line and the last one against fp64"""
    option("chain8", 1)
    B, H, K, N, P = 2800, 14, 256, 1024, 256
    HW = H * H
    g = torch.Generator(device=DEV); g.manual_seed(13)
    rn = lambda *sh: torch.randn(sh, device=DEV, generator=g)
    c3 = _entry((rn(N, K) * K ** -0.5).half().float(), g); c1 = _entry((rn(P, N) * N ** -0.5).half().float(), g)
    y2 = torch.relu(rn(B, H, H, K)); res = torch.relu(rn(B, H, H, N) * 2.0)
    assert res.numel() * 4 > 2 ** 31
    xam, yam = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    r = ops.bottleneck_chain(y2, (y2.abs().max() * 1.1).reshape(1), c3, res, c1, xam, yam)
    assert r is not None and ops._chain_tag == "bottleneck_chain8_kernel"
    x, y1 = r
    s = 2 ** 31 // (HW * N * 4)
    for i in (0, s - 1, s, s + 1, B - 1):
        xr = torch.relu(y2[i].view(HW, K).double() @ c3["w"].double().t() * c3["sc"].double() + c3["b"].double() + res[i].view(HW, N).double())
        yr = torch.relu(xr @ c1["w"].double().t() * c1["sc"].double() + c1["b"].double())
        assert relerr(x[i].view(HW, N).double().cpu(), xr.cpu()) < 5e-6, i
        assert relerr(y1[i].view(HW, P).double().cpu(), yr.cpu()) < 5e-6, i
    assert xam.item() == x.abs().max().item() and yam.item() == y1.abs().max().item()


def _entry(w, g, bias_std=0.1):
    n = w.shape[0]
    ph, we, k = ops.split_planes_f16(w, allow_single=True)
    assert k == 1
    return dict(w=w, ph=ph, we=we, sc=0.5 + torch.rand((n,), device=DEV, generator=g), b=torch.randn((n,), device=DEV, generator=g) * bias_std)


def test_chain_kernels_on_operands_over_2gib():
    """layer-1 geometry at the headline batch: x / x' are [B,56,56,256] fp32 = 3.2 MB per image, > 2 GiB from B = 669.
    bottleneck_chain (pooled), bottleneck_chain_dual and bottleneck_block_chain (plain and dual) at B = 720: first image,
    the images around the 2 GiB line and the last one against fp64."""
    B, H, K, N, P = 720, 56, 64, 256, 64
    HW = H * H
    M = B * HW
    g = torch.Generator(device=DEV); g.manual_seed(11)
    rn = lambda *sh: torch.randn(sh, device=DEV, generator=g)
    w3 = (rn(N, K) * K ** -0.5).half().float(); w1 = (rn(P, N) * N ** -0.5).half().float(); wd = (rn(N, 64) * 0.125).half().float()
    w2 = (rn(K, K, 3, 3) * (9 * K) ** -0.5).half().float()
    c3, c1, ds = _entry(w3, g), _entry(w1, g), _entry(wd, g)
    w2p, wl = ops.pack_conv_weight(w2, chunk_major=32)
    c2 = _entry(w2p, g); c2["wl"] = wl
    ratio = (ds["sc"].double() / c3["sc"].double() * 2.0 ** (c3["we"] - ds["we"])).float()
    bias_dual = c3["b"] + ds["b"]
    imgs = (0, 667, 668, 669, B - 1)

    def bn(t, c):
        return t * c["sc"].double() + c["b"].double()

    def conv2_ref(y1_img):                                              # [H,H,K] -> [HW,K] fp64
        v = F.conv2d(y1_img.permute(2, 0, 1)[None].double(), w2.double(), None, padding=1)[0].permute(1, 2, 0).reshape(HW, K)
        return torch.relu(bn(v, c2))

    y2 = torch.relu(rn(B, H, H, K))
    res = torch.relu(rn(B, H, H, N) * 2.0)
    assert res.numel() * 4 > 2 ** 31
    ya = (y2.abs().max() * 1.1).reshape(1)
    # --- conv3 + residual -> next conv1, with the pooled copy (the layer-1 -> layer-2 seam of the plan)
    xam, yam = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    c1p = _entry((rn(128, N) * N ** -0.5).half().float(), g)
    r = ops.bottleneck_chain(y2, ya, c3, res, c1p, xam, yam, pooled=True)
    assert r is not None
    x, xp, y1 = r
    for i in imgs:
        xr = torch.relu(bn(y2[i].view(HW, K).double() @ w3.double().t(), c3) + res[i].view(HW, N).double())
        yr = torch.relu(bn(xr @ c1p["w"].double().t(), c1p))
        assert relerr(x[i].view(HW, N).double().cpu(), xr.cpu()) < 5e-6, i
        assert relerr(y1[i].view(HW, 128).double().cpu(), yr.cpu()) < 5e-6, i
        pr = F.avg_pool2d(xr.view(H, H, N).permute(2, 0, 1)[None], 2)[0].permute(1, 2, 0)
        assert relerr(xp[i].double().cpu(), pr.cpu()) < 5e-6, i
    assert xam.item() == x.abs().max().item() and yam.item() == y1.abs().max().item()
    del r, x, xp, y1
    torch.cuda.empty_cache()
    # --- block chain, plain: conv2 -> conv3 + residual -> next conv1 (y2 here plays the block's conv1 output y1)
    xam.zero_(); yam.zero_()
    r = ops.bottleneck_block_chain(y2, ya, c2, c3, c1, residual=res, x_absmax=xam, y1n_absmax=yam)
    assert r is not None
    x, y1n = r
    for i in imgs:
        xr = torch.relu(bn(conv2_ref(y2[i]) @ w3.double().t(), c3) + res[i].view(HW, N).double())
        yr = torch.relu(bn(xr @ w1.double().t(), c1))
        assert relerr(x[i].view(HW, N).double().cpu(), xr.cpu()) < 5e-6, i
        assert relerr(y1n[i].view(HW, P).double().cpu(), yr.cpu()) < 5e-6, i
    assert xam.item() == x.abs().max().item() and yam.item() == y1n.abs().max().item()
    del r, x, y1n, res
    torch.cuda.empty_cache()
    # --- the stage's first block: downsample branch instead of the residual (chain_dual and block chain dual)
    a2 = torch.relu(rn(B, H, H, 64) * 3.0)
    aa = (a2.abs().max() * 1.3).reshape(1)
    xam.zero_(); yam.zero_()
    r = ops.bottleneck_chain_dual(y2.view(M, K), ya, c3, a2.view(M, 64), aa, ds, ratio, bias_dual, c1, xam, yam)
    assert r is not None
    x, y1 = r
    assert x.numel() * 4 > 2 ** 31
    for i in imgs:
        sl = slice(i * HW, (i + 1) * HW)
        xr = torch.relu(y2.view(M, K)[sl].double() @ w3.double().t() * c3["sc"].double()
                        + a2.view(M, 64)[sl].double() @ wd.double().t() * ds["sc"].double() + bias_dual.double())
        yr = torch.relu(bn(xr @ w1.double().t(), c1))
        assert relerr(x[sl].double().cpu(), xr.cpu()) < 5e-6, i
        assert relerr(y1[sl].double().cpu(), yr.cpu()) < 5e-6, i
    del r, x, y1
    torch.cuda.empty_cache()
    xam.zero_(); yam.zero_()
    dd = dict(a2=a2, a2_absmax=aa, ds=ds, ratio=ratio, bias=bias_dual)
    r = ops.bottleneck_block_chain(y2, ya, c2, c3, c1, dual=dd, x_absmax=xam, y1n_absmax=yam)
    assert r is not None
    x, y1n = r
    for i in imgs:
        xr = torch.relu(conv2_ref(y2[i]) @ w3.double().t() * c3["sc"].double()
                        + a2[i].view(HW, 64).double() @ wd.double().t() * ds["sc"].double() + bias_dual.double())
        yr = torch.relu(bn(xr @ w1.double().t(), c1))
        assert relerr(x[i].view(HW, N).double().cpu(), xr.cpu()) < 5e-6, i
        assert relerr(y1n[i].view(HW, P).double().cpu(), yr.cpu()) < 5e-6, i
    assert xam.item() == x.abs().max().item() and yam.item() == y1n.abs().max().item()


# ---------------------------------------------------------------------------------------------------------------------------
# guard zones
# ---------------------------------------------------------------------------------------------------------------------------

class GuardedAlloc:
    """stand-in for ops._empty: every tensor sits between two sentinel-filled zones, each at least 256 rows of the tensor's
    last dimension long (a ragged 128/256-row tile that ignored M would land there)"""
    S = -7.0

    def __init__(self):
        self.bufs = []

    def __call__(self, shape, device=None, dtype=torch.float32, **kw):
        shape = (shape,) if isinstance(shape, int) else tuple(shape)
        n = math.prod(shape)
        pad = max(1 << 16, 288 * (shape[-1] if shape else 1))
        pad = min(pad, 1 << 24)                                         # (a flat workspace of 10^8 elements is not "288 rows" of itself)
        pad = (pad + 63) // 64 * 64                                     # keeps 16-byte alignment of the payload
        raw = torch.empty(n + 2 * pad, device=device, dtype=dtype)
        raw[:pad] = self.S; raw[pad + n:] = self.S
        self.bufs.append((raw, pad, n))
        return raw[pad:pad + n].view(shape)

    def check(self):
        assert self.bufs
        for raw, pad, n in self.bufs:
            lo, hi = raw[:pad], raw[pad + n:]
            assert bool((lo == self.S).all()) and bool((hi == self.S).all()), \
                f"guard zone of a {n}-element {raw.dtype} output was written: {(lo != self.S).sum().item()} before, {(hi != self.S).sum().item()} after"


@pytest.fixture
def guarded(monkeypatch):
    ga = GuardedAlloc()
    monkeypatch.setattr(ops, "_empty", ga)
    return ga


@pytest.mark.parametrize("tail", [5, 77])
def test_ragged_tiles_do_not_write_outside_their_outputs(tail, guarded, option):
    g = torch.Generator(device=DEV); g.manual_seed(tail)
    rn = lambda *sh: torch.randn(sh, device=DEV, generator=g)
    # --- fp16-pair igemm, direct epilogue (ViT shapes: ragged M on every launch), with and without a residual
    for (M, N, K, res, act) in ((128 * 100 + tail, 256, 128, True, 0), (128 * 200 + tail, 768, 768, True, 2), (128 * 300 + tail, 64, 64, False, 1),
                                 (64 * 3 + tail, 256, 128, True, 0)):
        a = rn(M, K); w = (rn(N, K) * K ** -0.5).half().float(); b = rn(N); r = rn(M, N) if res else None
        ph, we, n = ops.split_planes_f16(w, allow_single=True)
        cam = torch.zeros(1, device=DEV)
        out = ops.gemm(a, w, b, residual=r, act=act, w_planes_f16=ph, w_exp=we, a_absmax=a.abs().max().reshape(1), c_absmax=cam)
        assert M < 10000 or ops._last_igemm_tag().startswith("igemm_x3_kernel<"), ops._last_igemm_tag()   # (small grids: fp32 64x64 tile)
        v = a.double() @ w.double().t() + b.double()
        if res:
            v = v + r.double()                                          # (the igemm epilogue activates after the residual)
        v = {0: v, 1: torch.relu(v), 2: v * torch.sigmoid(1.702 * v)}[act]
        assert relerr(out.double().cpu(), v.cpu()) < 5e-6 and cam.item() == out.abs().max().item()
    # --- the eight-phase parity GEMM (256-row tiles, 64 rows per wave)
    M, N, K = 256 * 70 + tail, 3072, 1024
    a = rn(M, K); w = (rn(N, K) * K ** -0.5).half().float(); b = rn(N)
    ph, we, n = ops.split_planes_f16(w, allow_single=True)
    cam = torch.zeros(1, device=DEV)
    out = ops.gemm(a, w, b, act=2, w_planes_f16=ph, w_exp=we, a_absmax=a.abs().max().reshape(1), c_absmax=cam)
    assert ops._last_igemm_tag() == "gemm_pair_8ph_kernel", ops._last_igemm_tag()
    rows = torch.cat([torch.arange(0, 300, device=DEV), torch.arange(M - 300, M, device=DEV)])
    v = a[rows].double() @ w.double().t() + b.double()
    assert relerr(out[rows].double().cpu(), (v * torch.sigmoid(1.702 * v)).cpu()) < 5e-6 and cam.item() == out.abs().max().item()
    M, N, K = 256 * 66 + tail, 1024, 4096
    a = rn(M, K); w = (rn(N, K) * K ** -0.5).half().float(); b = rn(N); r = rn(M, N)
    ph, we, n = ops.split_planes_f16(w, allow_single=True)
    out = ops.gemm(a, w, b, residual=r, w_planes_f16=ph, w_exp=we, a_absmax=a.abs().max().reshape(1))
    assert ops._last_igemm_tag() == "gemm_pair_8ph_kernel", ops._last_igemm_tag()
    v = a[rows % M].double() @ w.double().t() + b.double() + r[rows % M].double()
    assert relerr(out[rows % M].double().cpu(), v.cpu()) < 5e-6
    # --- fp16 mode: the eight-phase GEMM (128 rows per wave) and the two-barrier kernel
    for (M, N, K, res) in ((16384 + tail, 512, 256, True), (16384 + 128 + tail, 1024, 1024, False), (128 * 3 + tail, 256, 128, True)):
        a = rn(M, K).half(); w = (rn(N, K) * K ** -0.5).half(); b = rn(N); r = rn(M, N).half() if res else None
        out = ops.gemm_f16(a, w, b, residual=r)
        v = a.double() @ w.double().t() + b.double()
        if res:
            v = v + r.double()
        assert relerr(out.double().cpu(), v.cpu()) < 1.5e-3
    # --- chain kernels (rows masked by lane; M % 4 == 0 is their contract, so the tail is a multiple of 4)
    t4 = (tail + 3) // 4 * 4
    Hh, Ww = 4, (128 * 2 + t4) // 4
    K, N, P = 64, 256, 64
    y2 = torch.relu(rn(1, Hh, Ww, K)); res = torch.relu(rn(1, Hh, Ww, N))
    w3 = (rn(N, K) * K ** -0.5).half().float(); w1 = (rn(P, N) * N ** -0.5).half().float()
    c3, c1 = _entry(w3, g), _entry(w1, g)
    r_ = ops.bottleneck_chain(y2, y2.abs().max().reshape(1), c3, res, c1, torch.zeros(1, device=DEV), torch.zeros(1, device=DEV))
    assert r_ is not None
    Mc = Hh * Ww
    xr = torch.relu(y2.view(Mc, K).double() @ w3.double().t() * c3["sc"].double() + c3["b"].double() + res.view(Mc, N).double())
    assert relerr(r_[0].view(Mc, N).double().cpu(), xr.cpu()) < 5e-6
    w2 = (rn(K, K, 3, 3) * (9 * K) ** -0.5).half().float()
    w2p, wl = ops.pack_conv_weight(w2, chunk_major=32)
    c2 = _entry(w2p, g); c2["wl"] = wl
    r_ = ops.bottleneck_block_chain(y2, y2.abs().max().reshape(1), c2, c3, c1, residual=res, x_absmax=torch.zeros(1, device=DEV),
                                    y1n_absmax=torch.zeros(1, device=DEV))
    assert r_ is not None
    # ... and the eight-wave chain of the layer-3 geometry (K = P = 256)
    option("chain8", 1)
    y2b = torch.relu(rn(1, Hh, Ww, 256)); resb = torch.relu(rn(1, Hh, Ww, 128))
    c3b, c1b = _entry((rn(128, 256) * 0.0625).half().float(), g), _entry((rn(256, 128) * 128 ** -0.5).half().float(), g)
    r_ = ops.bottleneck_chain(y2b, y2b.abs().max().reshape(1), c3b, resb, c1b, torch.zeros(1, device=DEV), torch.zeros(1, device=DEV))
    assert r_ is not None and ops._chain_tag == "bottleneck_chain8_kernel"
    xr = torch.relu(y2b.view(Mc, 256).double() @ c3b["w"].double().t() * c3b["sc"].double() + c3b["b"].double() + resb.view(Mc, 128).double())
    assert relerr(r_[0].view(Mc, 128).double().cpu(), xr.cpu()) < 5e-6
    # --- stem convs: stride-2 MFMA gather kernel and the 32-channel patch kernels
    x = rn(3, 3, 64, 64)
    wst = (rn(3, 3, 3, 32) * 0.2)
    ys = ops.conv_stem_s2(x, wst.contiguous(), rn(32) * 0.1, y_absmax=torch.zeros(1, device=DEV))
    assert tuple(ys.shape) == (3, 32, 32, 32)
    xs = torch.relu(rn(2, 8, 28, 32))
    wq = (rn(64, 32, 3, 3) * 288 ** -0.5).half().float()
    wqp, wql = ops.pack_conv_weight(wq, chunk_major=32)
    cq = _entry(wqp, g)
    for pool in (1, 2):
        ops.conv_bn_act(xs, wqp, cq["b"], None, 3, 3, 1, 1, ops.ACT_RELU, wql, w_planes_f16=cq["ph"], w_exp=cq["we"],
                        x_absmax=xs.abs().max().reshape(1), y_absmax=torch.zeros(1, device=DEV), out_scale=cq["sc"], pool=pool)
    # --- fp16-mode conv kernels (csrc/conv_f16.hip): LDS-DMA loaders, packed stores from the accumulator layout, rows masked by lane
    xh = torch.relu(rn(1, 4, (256 * 2 + t4) // 4, 64)).half()                                   # M = 2 tiles + tail
    for cout, pool in ((64, 1), (136, 1), (32, 1), (64, 2), (136, 2)):
        wh = (rn(cout, 9 * 64) * 0.05).half()
        yh = ops.conv3x3_f16(xh, wh, torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV), pool=pool)
        assert yh is not None and torch.isfinite(yh).all()
    for cin, cout, res in ((64, 64, False), (64, 256, True), (128, 136, True)):
        xa = rn(256 * 2 + tail, cin).half(); wa = (rn(cout, cin) * cin ** -0.5).half()
        ra = rn(256 * 2 + tail, cout).half() if res else None
        for mode in (0, 2):
            ops.set_option("conv1x1_stream", mode)
            ya = ops.conv1x1_f16(xa, wa, torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV), residual=ra)
            v = xa.double() @ wa.double().t() + (ra.double() if res else 0.0)
            assert relerr(ya.double().cpu(), torch.relu(v).cpu()) < 1.5e-3
        ops.set_option("conv1x1_stream", 1)
    ys = ops.conv_stem_s2_f16(rn(3, 3, 40, 40), (rn(3, 3, 3, 32) * 0.2).contiguous(), rn(32) * 0.1)
    ops.avgpool2_f16(ys)
    torch.cuda.synchronize()
    guarded.check()
