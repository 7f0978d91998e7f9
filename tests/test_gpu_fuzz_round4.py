"""Seeded random shapes for the fp16 ModifiedResNet kernels added in round 4 (clip/model.py:42-55, 108-116, 141-148): every launch writes
into the middle of a canary buffer and is compared with the kernel it replaces (bit for bit where the arithmetic is the same), so an index
that leaves the tensor, a ragged tile or a window wrap that a hand-picked shape happened to miss shows up here."""
import random

import pytest
import torch
import torch.nn.functional as F

from conftest import relerr
from dbmm_amd import ops, _lib

pytestmark = pytest.mark.gpu
DEV = "cuda"
GUARD = 2048


def _bn(g, n):
    return 0.5 + torch.rand((n,), device=DEV, generator=g), torch.randn((n,), device=DEV, generator=g) * 0.1


def _guarded(shape):
    n = 1
    for s in shape:
        n *= s
    buf = torch.full((n + 2 * GUARD,), 7.0, device=DEV, dtype=torch.float16)
    return buf, buf[GUARD:GUARD + n].view(shape)


def _intact(buf, n):
    return bool((buf[:GUARD] == 7.0).all() and (buf[GUARD + n:] == 7.0).all())


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_chain_and_seam_kernels_f16(seed):
    rnd = random.Random(1000 + seed)
    g = torch.Generator(device=DEV); g.manual_seed(seed)
    K = rnd.choice([64, 128]); P = rnd.choice([64, 128]); N = 64 * rnd.randint(1, 9)
    B = rnd.randint(1, 9); H = 2 * rnd.randint(1, 15); W = 2 * rnd.randint(1, 15)
    y2 = torch.relu(torch.randn((B, H, W, K), device=DEV, generator=g)).half()
    res = torch.relu(torch.randn((B, H, W, N), device=DEV, generator=g, dtype=torch.float16) * 2.0)
    w3 = (torch.randn((N, K), device=DEV, generator=g) * K ** -0.5).half(); w1 = (torch.randn((P, N), device=DEV, generator=g) * N ** -0.5).half()
    (s3, b3), (s1, b1) = _bn(g, N), _bn(g, P)
    x_ref = ops.conv1x1_f16(y2, w3, s3, b3, residual=res)
    y1_ref = ops.conv1x1_f16(x_ref, w1, s1, b1)
    xp_ref = ops.avgpool2_f16(x_ref)
    M = B * H * W
    for pooled, full in ((False, True), (True, True), (True, False)):
        bx, x = _guarded((B, H, W, N)); by, y1 = _guarded((B, H, W, P)); bp, xp = _guarded((B, H // 2, W // 2, N))
        if pooled:
            rc = _lib.lib().dbmm_bottleneck_chain_pool_f16(y2.data_ptr(), w3.data_ptr(), s3.data_ptr(), b3.data_ptr(), res.data_ptr(),
                                                           x.data_ptr() if full else None, xp.data_ptr(), w1.data_ptr(), s1.data_ptr(), b1.data_ptr(),
                                                           y1.data_ptr(), B, H, W, K, N, P, _lib.stream())
        else:
            rc = _lib.lib().dbmm_bottleneck_chain_f16(y2.data_ptr(), w3.data_ptr(), s3.data_ptr(), b3.data_ptr(), res.data_ptr(), x.data_ptr(),
                                                      w1.data_ptr(), s1.data_ptr(), b1.data_ptr(), y1.data_ptr(), M, K, N, P, _lib.stream())
        torch.cuda.synchronize()
        assert rc == 0, (rc, B, H, W, K, N, P)
        assert _intact(bx, M * N) and _intact(by, M * P) and _intact(bp, M // 4 * N), (B, H, W, K, N, P, pooled, full)
        assert torch.equal(y1, y1_ref), (B, H, W, K, N, P, pooled, full)
        assert torch.equal(x, x_ref) if full else bool((x == 7.0).all())
        assert torch.equal(xp, xp_ref) if pooled else bool((xp == 7.0).all())
    # the seam without a chain kernel: conv3 + residual + pooled copy in one launch
    bx, x = _guarded((B, H, W, N)); bp, xp = _guarded((B, H // 2, W // 2, N))
    for Kc in (128, 256):
        a = torch.relu(torch.randn((B, H, W, Kc), device=DEV, generator=g)).half(); wc = (torch.randn((N, Kc), device=DEV, generator=g) * Kc ** -0.5).half()
        rc = _lib.lib().dbmm_conv1x1_res_pool_f16(a.data_ptr(), wc.data_ptr(), s3.data_ptr(), b3.data_ptr(), res.data_ptr(), x.data_ptr(), xp.data_ptr(),
                                                  B, H, W, Kc, N, _lib.stream())
        torch.cuda.synchronize()
        ref = ops.conv1x1_f16(a, wc, s3, b3, residual=res)
        assert rc == 0 and _intact(bx, M * N) and _intact(bp, M // 4 * N), (B, H, W, Kc, N)
        assert torch.equal(x, ref) and torch.equal(xp, ops.avgpool2_f16(ref)), (B, H, W, Kc, N)


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_stem_kernels_f16(seed, option):
    rnd = random.Random(2000 + seed)
    g = torch.Generator(device=DEV); g.manual_seed(seed)
    # stem conv1 on the MFMA gather kernel: any image size, fp32 or fp16 input
    B = rnd.randint(1, 6); R = rnd.randint(5, 70); R2 = rnd.randint(5, 70); Cout = rnd.choice([32, 64]); half_in = rnd.random() < 0.5
    img = torch.randn((B, 3, R, R2), device=DEV, generator=g)
    w = (torch.randn((3, 3, 3, Cout), device=DEV, generator=g) * 0.2).half().float()
    sc, b = _bn(g, Cout)
    Ho, Wo = (R - 1) // 2 + 1, (R2 - 1) // 2 + 1
    buf, y = _guarded((B, Ho, Wo, Cout))
    xin = img.half() if half_in else img
    rc = _lib.lib().dbmm_conv_stem_s2_bn_f16(xin.data_ptr(), int(half_in), w.data_ptr(), sc.data_ptr(), b.data_ptr(), y.data_ptr(), B, R, R2, Cout, _lib.stream())
    torch.cuda.synchronize()
    assert rc == 0 and _intact(buf, y.numel()), (B, R, R2, Cout)
    ref = F.conv2d(img.half().double(), w.permute(3, 2, 0, 1).double(), None, stride=2, padding=1) * sc.double().view(1, -1, 1, 1) + b.double().view(1, -1, 1, 1)
    assert relerr(y.double().cpu(), torch.relu(ref).permute(0, 2, 3, 1).cpu()) < 1e-3, (B, R, R2, Cout)
    # the parity mode's stem conv on the same gather (images with fewer than 31 output pixels: a block of 32 touches more than two images)
    w32 = torch.randn((3, 3, 3, Cout), device=DEV, generator=g) * 0.2; b32 = torch.randn((Cout,), device=DEV, generator=g) * 0.1
    am = torch.zeros(1, device=DEV)
    y32 = ops.conv_stem_s2(img, w32, b32, y_absmax=am)
    ref32 = torch.relu(F.conv2d(img.double(), w32.permute(3, 2, 0, 1).double(), b32.double(), stride=2, padding=1)).permute(0, 2, 3, 1)
    assert relerr(y32.double().cpu(), ref32.cpu()) < 5e-6 and am.item() == y32.abs().max().item(), (B, R, R2, Cout)
    # the 32-channel convs on the patch kernel: maps of 4 i x 28 j pixels
    B = rnd.randint(1, 5); H = 4 * rnd.randint(1, 8); W = 28 * rnd.randint(1, 3); Cout = rnd.choice([32, 64]); pool = rnd.choice([1, 2])
    x = torch.relu(torch.randn((B, H, W, 32), device=DEV, generator=g)).half()
    wc = (torch.randn((Cout, 32, 3, 3), device=DEV, generator=g) * 288 ** -0.5).half()
    wp, _ = ops.pack_conv_weight(wc.float(), chunk_major=32)
    wh = wp.half().contiguous()
    sc, b = _bn(g, Cout)
    buf, y = _guarded((B, H // pool, W // pool, Cout))
    rc = _lib.lib().dbmm_conv3x3_bn_relu_f16(x.data_ptr(), wh.data_ptr(), sc.data_ptr(), b.data_ptr(), y.data_ptr(), B, H, W, 32, Cout, 2 if pool == 2 else 0,
                                             _lib.stream())
    torch.cuda.synchronize()
    assert rc == 0 and _intact(buf, y.numel()), (B, H, W, Cout, pool)
    option("conv_patch", 0)
    assert torch.equal(y, ops.conv3x3_f16(x, wh, sc, b, pool=pool)), (B, H, W, Cout, pool)


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_parity_seam_kernels(seed, option):
    """fp32-accurate mode: the chain launch that writes only the pooled stage output (bottleneck_chain_kernel<.., POOL = 2>) and the row-owning
    conv3 + residual kernel (conv1x1_res_stream_kernel<256, POOL>) on random maps, against the launches they replace, canaries around every output"""
    rnd = random.Random(3000 + seed)
    g = torch.Generator(device=DEV); g.manual_seed(seed)
    mk = lambda n: (0.5 + torch.rand((n,), device=DEV, generator=g), torch.randn((n,), device=DEV, generator=g) * 0.1)
    K = rnd.choice([64, 128]); P = rnd.choice([64, 128]); N = 64 * rnd.randint(1, 8)
    B = rnd.randint(1, 7); H = 2 * rnd.randint(1, 14); W = 2 * rnd.randint(1, 14)
    y2 = torch.relu(torch.randn((B, H, W, K), device=DEV, generator=g)); res = torch.relu(torch.randn((B, H, W, N), device=DEV, generator=g) * 2.0)
    w3 = (torch.randn((N, K), device=DEV, generator=g) * K ** -0.5).half().float(); w1 = (torch.randn((P, N), device=DEV, generator=g) * N ** -0.5).half().float()
    (s3, b3), (s1, b1) = mk(N), mk(P)
    p3, e3, _ = ops.split_planes_f16(w3, allow_single=True); p1, e1, _ = ops.split_planes_f16(w1, allow_single=True)
    c3 = dict(w=w3, ph=p3, we=e3, sc=s3, b=b3); c1 = dict(w=w1, ph=p1, we=e1, sc=s1, b=b1)
    ya = (y2.abs().max() * 1.1).reshape(1)
    xa, ya1 = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    x, xp, y1 = ops.bottleneck_chain(y2, ya, c3, res, c1, xa, ya1, pooled=True)
    xb, yb = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    M = B * H * W
    bp = torch.full((M // 4 * N + 2 * GUARD,), 7.0, device=DEV); by = torch.full((M * P + 2 * GUARD,), 7.0, device=DEV)
    op, oy = bp[GUARD:GUARD + M // 4 * N].view(B, H // 2, W // 2, N), by[GUARD:GUARD + M * P].view(B, H, W, P)
    rc = _lib.lib().dbmm_bottleneck_chain_x2(y2.data_ptr(), ya.data_ptr(), p3.data_ptr(), int(e3), s3.data_ptr(), b3.data_ptr(), res.data_ptr(), None,
                                            op.data_ptr(), xb.data_ptr(), p1.data_ptr(), int(e1), s1.data_ptr(), b1.data_ptr(), oy.data_ptr(), yb.data_ptr(),
                                            B, H, W, K, N, P, _lib.stream())
    torch.cuda.synchronize()
    assert rc == 0, (rc, B, H, W, K, N, P)
    assert (bp[:GUARD] == 7.0).all() and (bp[GUARD + M // 4 * N:] == 7.0).all() and (by[:GUARD] == 7.0).all() and (by[GUARD + M * P:] == 7.0).all()
    assert torch.equal(op, xp) and torch.equal(oy, y1) and xb.item() == xa.item() and yb.item() == ya1.item(), (B, H, W, K, N, P)
    # conv1x1_res_stream_kernel: K = 256, >= 131,072 rows
    H = 2 * rnd.randint(4, 20); W = 2 * rnd.randint(4, 20); B = 131072 // (H * W) + rnd.randint(1, 5); N = 32 * rnd.randint(32, 36)
    x = torch.relu(torch.randn((B, H, W, 256), device=DEV, generator=g)); res = torch.relu(torch.randn((B, H, W, N), device=DEV, generator=g) * 2.0)
    w = (torch.randn((N, 256, 1, 1), device=DEV, generator=g) * 256 ** -0.5).half().float()
    sc, b = mk(N)
    wp, wl = ops.pack_conv_weight(w, chunk_major=32)
    ph, we, _ = ops.split_planes_f16(wp, allow_single=True)
    xam = (x.abs().max() * 1.2).reshape(1)
    kw = dict(w_planes_f16=ph, w_exp=we, x_absmax=xam, out_scale=sc)
    M = B * H * W
    for pool in (1, 2):
        option("conv1x1_res_stream", 0)
        r0 = ops.conv_bn_act(x, wp, b, res, 1, 1, 1, 0, ops.ACT_RELU, wl, pool=pool, keep_full=pool == 2, **kw)
        option("conv1x1_res_stream", 1)
        bf = torch.full((M * N + 2 * GUARD,), 7.0, device=DEV); bq = torch.full((M // 4 * N + 2 * GUARD,), 7.0, device=DEV)
        of, oq = bf[GUARD:GUARD + M * N].view(B, H, W, N), bq[GUARD:GUARD + M // 4 * N].view(B, H // 2, W // 2, N)
        rc = ops._conv_x2(x, wp, b, res, oq if pool == 2 else of, 1, 1, 1, 0, ops.ACT_RELU, wl, ph, we, xam, None, sc, pool if pool == 2 else 0, of if pool == 2 else None)
        torch.cuda.synchronize()
        assert rc == 0 and ops._last_igemm_tag() == f"conv1x1_res_stream_kernel<256, {int(pool == 2)}>", (rc, ops._last_igemm_tag(), B, H, W, N)
        assert (bf[:GUARD] == 7.0).all() and (bf[GUARD + M * N:] == 7.0).all() and (bq[:GUARD] == 7.0).all() and (bq[GUARD + M // 4 * N:] == 7.0).all()
        if pool == 2:
            assert relerr(oq.cpu(), r0[0].cpu()) < 2e-6 and relerr(of.cpu(), r0[1].cpu()) < 2e-6, (B, H, W, N)
        else:
            assert relerr(of.cpu(), r0.cpu()) < 2e-6 and (oq == 7.0).all(), (B, H, W, N)


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_eight_phase_3x3_kernels(seed, option):
    """the eight-phase 3x3 kernels of both modes (conv3x3_halo8 / halo8n, conv3x3_f16_8ph) on random maps with >= 16,384 pixels: equal to the
    kernels they replace (bit for bit; whole tiles: tail_split = 0), canaries around the output"""
    rnd = random.Random(4000 + seed)
    g = torch.Generator(device=DEV); g.manual_seed(seed)
    Cin = rnd.choice([64, 128, 256]); Cout = rnd.choice([128, 256, 384, 512]); pool = rnd.choice([1, 2])
    H = 2 * rnd.randint(2, 16); W = 2 * rnd.randint(2, 16); B = 16384 // (H * W) + rnd.randint(1, 40)
    sc = 0.5 + torch.rand((Cout,), device=DEV, generator=g); b = torch.randn((Cout,), device=DEV, generator=g) * 0.1
    w = (torch.randn((Cout, Cin, 3, 3), device=DEV, generator=g) * (9 * Cin) ** -0.5).half().float()
    wp, wl = ops.pack_conv_weight(w, chunk_major=32)
    Ho, Wo = H // pool, W // pool
    n_out = B * Ho * Wo * Cout
    # parity mode
    x = torch.relu(torch.randn((B, H, W, Cin), device=DEV, generator=g))
    ph, we, _ = ops.split_planes_f16(wp, allow_single=True)
    xam = x.abs().max().reshape(1)
    option("igemm_streamk", 0); option("tail_split", 0)
    option("halo8", 0)
    y0 = ops.conv_bn_act(x, wp, b, None, 3, 3, 1, 1, ops.ACT_RELU, wl, w_planes_f16=ph, w_exp=we, x_absmax=xam, out_scale=sc, pool=pool)
    option("halo8", 2)
    buf = torch.full((n_out + 2 * GUARD,), 7.0, device=DEV)
    out = buf[GUARD:GUARD + n_out].view(B, Ho, Wo, Cout)
    rc = ops._conv_x2(x, wp, b, None, out, 3, 3, 1, 1, ops.ACT_RELU, wl, ph, we, xam, None, sc, 2 if pool == 2 else 0, None)
    torch.cuda.synchronize()
    assert rc == 0 and ops._last_igemm_tag().startswith("conv3x3_halo8"), (rc, ops._last_igemm_tag(), B, H, W, Cin, Cout, pool)
    assert (buf[:GUARD] == 7.0).all() and (buf[GUARD + n_out:] == 7.0).all() and torch.equal(out, y0), (B, H, W, Cin, Cout, pool)
    # fp16 mode (the eight-phase kernel takes Cout % 256 == 0)
    if Cout % 256 == 0:
        xh = x.half(); wh = wp.half().contiguous()
        option("f16_conv_8ph", 0)
        yh0 = ops.conv3x3_f16(xh, wh, sc, b, pool=pool)
        option("f16_conv_8ph", 1)
        bh = torch.full((n_out + 2 * GUARD,), 7.0, device=DEV, dtype=torch.float16)
        oh = bh[GUARD:GUARD + n_out].view(B, Ho, Wo, Cout)
        rc = _lib.lib().dbmm_conv3x3_bn_relu_f16(xh.data_ptr(), wh.data_ptr(), sc.data_ptr(), b.data_ptr(), oh.data_ptr(), B, H, W, Cin, Cout,
                                                 2 if pool == 2 else 0, _lib.stream())
        torch.cuda.synchronize()
        assert rc == 0 and (bh[:GUARD] == 7.0).all() and (bh[GUARD + n_out:] == 7.0).all() and torch.equal(oh, yh0), (B, H, W, Cin, Cout, pool)


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_eight_phase_gemms(seed, option):
    """the eight-phase GEMM kernels of both modes (gemm_pair_8ph, gemm_f16_8ph; K cut / row split of a short last round wherever it applies:
    tail_split = 2) on random (M, N, K) with bias, residual and both activations: against the two-barrier kernels (gemm_8ph / f16_8ph = 0)
    -- the cut changes the order of summation, so to rounding -- and with the maximum scalar; outputs sit inside canary buffers"""
    rnd = random.Random(5000 + seed)
    g = torch.Generator(device=DEV); g.manual_seed(seed)
    M = rnd.randint(16384, 60000); N = 256 * rnd.randint(1, 6); K = 128 * rnd.randint(1, 16)
    act = rnd.choice([ops.ACT_NONE, ops.ACT_RELU, ops.ACT_QUICKGELU]); use_res = rnd.random() < 0.5
    a = torch.randn((M, K), device=DEV, generator=g); w = (torch.randn((N, K), device=DEV, generator=g) * K ** -0.5).half().float()
    bias = torch.randn((N,), device=DEV, generator=g) * 0.1
    r = torch.randn((M, N), device=DEV, generator=g) if use_res else None
    ph, we, n = ops.split_planes_f16(w, allow_single=True)
    aam = a.abs().max().reshape(1)
    option("tail_split", 2)
    option("gemm_8ph", 0)
    am0 = torch.zeros(1, device=DEV)
    c0 = ops.gemm(a, w, bias, residual=r, act=act, w_planes_f16=ph, w_exp=we, a_absmax=aam, c_absmax=am0)
    option("gemm_8ph", 2)
    buf = torch.full((M * N + 2 * GUARD,), 7.0, device=DEV)
    out = buf[GUARD:GUARD + M * N].view(M, N)
    am = torch.zeros(1, device=DEV)
    c = ops.gemm(a, w, bias, residual=r, act=act, w_planes_f16=ph, w_exp=we, a_absmax=aam, c_absmax=am, out=out)
    torch.cuda.synchronize()
    assert ops._last_igemm_tag().startswith("gemm_pair_8ph_kernel"), (ops._last_igemm_tag(), M, N, K)
    assert (buf[:GUARD] == 7.0).all() and (buf[GUARD + M * N:] == 7.0).all(), (M, N, K)
    assert relerr(c.cpu(), c0.cpu()) < 3e-6 and am.item() == c.abs().max().item(), (M, N, K, act, use_res, relerr(c.cpu(), c0.cpu()))
    # fp16 mode
    ah = a.half(); wh = w.half(); rh = r.half() if use_res else None
    option("f16_8ph", 0)
    h0 = ops.gemm_f16(ah, wh, bias, residual=rh, act=act)
    option("f16_8ph", 1)
    h = ops.gemm_f16(ah, wh, bias, residual=rh, act=act)
    tol = 4e-3 * max(1.0, h0.float().abs().max().item())
    assert (h.float() - h0.float()).abs().max().item() <= tol, (M, N, K, act, use_res)


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_attention_cores(seed, option):
    """softmax(q k^T / 8) v per (image, head) (clip/model.py:185-189 through nn.MultiheadAttention) on random sequence lengths around the
    workgroup sizes (1 ... 150 tokens: two-wave workgroups up to 64, four-wave above), heads and batch, causal or not, both modes, against
    fp64; with mha_short = 0 the same results bit for bit"""
    rnd = random.Random(6000 + seed)
    g = torch.Generator(device=DEV); g.manual_seed(seed)
    L = rnd.choice([1, 2, 7, 31, 32, 33, 49, 50, 63, 64, 65, 77, 96, 127, 128, 129, 150]); heads = rnd.randint(1, 12); B = rnd.randint(1, 40)
    causal = rnd.random() < 0.5
    E = 64 * heads
    qkv = torch.randn((B * L, 3 * E), device=DEV, generator=g)
    q, k, v = [t.view(B, L, heads, 64).permute(0, 2, 1, 3).double() for t in qkv.split(E, dim=1)]
    s = q @ k.transpose(-1, -2) / 8.0
    if causal:
        s = s + torch.full((L, L), float("-inf"), device=DEV, dtype=torch.float64).triu(1)
    ref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B * L, E)
    am = qkv.abs().max().reshape(1)
    o = ops.mha_core(qkv, B, L, E, heads, causal, qkv_absmax=am)
    assert relerr(o.double().cpu(), ref.cpu()) < 1e-5, (B, L, heads, causal)
    qh = qkv.half()
    qr, kr, vr = [t.view(B, L, heads, 64).permute(0, 2, 1, 3).double() for t in qh.split(E, dim=1)]
    sr = qr @ kr.transpose(-1, -2) / 8.0
    if causal:
        sr = sr + torch.full((L, L), float("-inf"), device=DEV, dtype=torch.float64).triu(1)
    refh = (torch.softmax(sr, -1) @ vr).permute(0, 2, 1, 3).reshape(B * L, E)
    oh = ops.mha_core_f16(qh, B, L, E, heads, causal)
    assert (oh.double() - refh).abs().max().item() <= 6e-3 * max(1.0, refh.abs().max().item()), (B, L, heads, causal)
    option("mha_short", 0)
    assert torch.equal(ops.mha_core(qkv, B, L, E, heads, causal, qkv_absmax=am), o) and torch.equal(ops.mha_core_f16(qh, B, L, E, heads, causal), oh)


def test_other_input_resolutions_raise_in_the_vit_towers_too():
    """ViT-B/32 built for 224 px fed 160 / 256 px images: the reference fails at `x + self.positional_embedding` (clip/model.py:229)"""
    from dbmm_amd import synth
    from dbmm_amd.clip.model import build_model, convert_weights
    sd = synth.clip_state_dict(5, "ViT-B/32")
    for R in (160, 256):
        img = F.interpolate(synth.images(3, 2, 224).cuda(), size=(R, R), mode="bilinear", align_corners=False).contiguous()
        with pytest.raises(RuntimeError, match="positional embedding"):
            build_model(sd).cuda().encode_image(img)
        with pytest.raises(RuntimeError, match="positional embedding"):
            convert_weights(build_model(sd).cuda()).encode_image(img)
    torch.cuda.synchronize()


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_whole_towers_inside_guard_zones(seed, monkeypatch):
    """the full RN50 tower in both modes at random batch sizes (ragged tiles in every layer), with EVERY tensor the library allocates between
    sentinel zones; rows of a call == the same images in other batches (no cross-row leakage), fp16 mode within its distance of the
    fp32-accurate embedding; other input resolutions (other map sizes in every stage up to the attention pool) raise as the reference does"""
    from test_gpu_headline import GuardedAlloc
    from dbmm_amd import synth
    from dbmm_amd.clip.model import build_model, convert_weights
    rnd = random.Random(7000 + seed)
    B = rnd.randint(1, 23); R = 224 if rnd.random() < 0.6 else 32 * rnd.randint(2, 8)
    sd = synth.clip_state_dict(5, "RN50")
    model = build_model(sd).cuda()
    img = synth.images(100 + seed, B, 224).cuda()
    if R != 224:
        img = F.interpolate(img, size=(R, R), mode="bilinear", align_corners=False).contiguous()
    ga = GuardedAlloc()
    monkeypatch.setattr(ops, "_empty", ga)
    if R != 224:
        # another input resolution than the model was built for: the reference fails at `x + self.positional_embedding` (clip/model.py:72);
        # here the wrapper checks the operand shapes before the library sees raw pointers -- both modes raise, nothing faults
        with pytest.raises(RuntimeError, match="positional embedding"):
            model.encode_image(img)
        with pytest.raises(RuntimeError, match="positional embedding"):
            convert_weights(build_model(sd).cuda()).encode_image(img)
        torch.cuda.synchronize()
        ga.check()
        return
    out = model.encode_image(img)
    ga.check()
    k = rnd.randint(0, B - 1)
    one = model.encode_image(img[k:k + 1].contiguous())
    assert relerr(out[k:k + 1].cpu(), one.cpu()) < 1e-5, (B, k)
    m16 = convert_weights(build_model(sd).cuda())
    ga16 = GuardedAlloc()
    monkeypatch.setattr(ops, "_empty", ga16)
    o16 = m16.encode_image(img)
    ga16.check()
    assert o16.dtype == torch.float16 and relerr(o16.float().cpu(), out.cpu()) < 3e-2, (B, relerr(o16.float().cpu(), out.cpu()))
    assert torch.equal(m16.encode_image(img[k:k + 1].contiguous()), o16[k:k + 1]) or relerr(m16.encode_image(img[k:k + 1].contiguous()).float().cpu(), o16[k:k + 1].float().cpu()) < 2e-3


def test_wrappers_check_operand_shapes_before_launching():
    """mis-shaped operands raise (the library would index past them): attention pool with another map size, a conv with a weight of another
    depth, a bias / scale / residual of another width, a qkv buffer of another sequence length"""
    z = lambda *s, dt=torch.float32: torch.zeros(s, device=DEV, dtype=dt)
    with pytest.raises(RuntimeError, match="positional embedding"):
        ops.attnpool(z(2, 3, 3, 64), z(5, 64), z(64, 64), z(64), z(128, 64), z(128), z(32, 64), z(32), 1)
    with pytest.raises(RuntimeError, match="projection shapes"):
        ops.attnpool(z(2, 2, 2, 64), z(5, 64), z(64, 32), z(64), z(128, 64), z(128), z(32, 64), z(32), 1)
    with pytest.raises(RuntimeError, match="packed weight"):
        ops.conv_bn_act(z(1, 4, 4, 32), z(16, 9 * 16), None, None, 3, 3, 1, 1, ops.ACT_RELU)
    with pytest.raises(RuntimeError, match="bias"):
        ops.conv_bn_act(z(1, 4, 4, 32), z(16, 32), z(8), None, 1, 1, 1, 0, ops.ACT_RELU)
    with pytest.raises(RuntimeError, match="residual"):
        ops.conv_bn_act(z(1, 4, 4, 32), z(16, 32), None, z(1, 4, 4, 8), 1, 1, 1, 0, ops.ACT_RELU)
    with pytest.raises(RuntimeError, match="gemm"):
        ops.gemm(z(8, 32), z(16, 64))
    with pytest.raises(RuntimeError, match="residual"):
        ops.gemm(z(8, 32), z(16, 32), residual=z(8, 8))
    with pytest.raises(RuntimeError, match="scale"):
        ops.conv1x1_f16(z(8, 64, dt=torch.float16), z(64, 64, dt=torch.float16), z(32), z(64))
    with pytest.raises(RuntimeError, match="input channels"):
        ops.conv1x1_f16(z(8, 32, dt=torch.float16), z(64, 64, dt=torch.float16), z(64), z(64))
    with pytest.raises(RuntimeError, match="packed weight"):
        ops.conv3x3_f16(z(1, 4, 4, 32, dt=torch.float16), z(32, 9 * 64, dt=torch.float16), z(32), z(32))
    with pytest.raises(RuntimeError, match="qkv"):
        ops.mha_core(z(10, 3 * 64), 2, 7, 64, 1, False)
    with pytest.raises(RuntimeError, match="qkv"):
        ops.mha_core_f16(z(10, 3 * 64, dt=torch.float16), 2, 7, 64, 1, False)
    with pytest.raises(RuntimeError, match="layernorm"):
        ops.layernorm(z(4, 64), z(64), z(32))
    with pytest.raises(RuntimeError, match="tokens per prompt"):
        ops.embed_gather(torch.zeros((2, 9), device=DEV, dtype=torch.int32), z(100, 64), z(77, 64))
    torch.cuda.synchronize()


@pytest.mark.parametrize("mode", ["f16", "f32"])
def test_rn50_at_the_headline_batch_inside_guard_zones(mode, monkeypatch):
    """the RN50 tower at B = 1024 (the size bench.py's headline and rn50_f16_bs1024 legs run) in both modes with every tensor the library
    allocates between sentinel zones; rows 0 / 511 / 1023 == the same images in a batch of three (fp16 mode: bit for bit -- every kernel's
    rows are independent; fp32-accurate mode: to rounding -- the per-tensor fp16 scales depend on the batch maximum)"""
    from test_gpu_headline import GuardedAlloc
    from dbmm_amd import synth
    from dbmm_amd.clip.model import build_model, convert_weights
    model = build_model(synth.clip_state_dict(5, "RN50")).cuda()
    if mode == "f16":
        model = convert_weights(model)
    base = synth.images(977, 64, 224)
    scale = torch.linspace(0.7, 1.3, 16).repeat_interleave(64).view(-1, 1, 1, 1)
    img = (base.repeat(16, 1, 1, 1) * scale).contiguous().cuda()
    ga = GuardedAlloc()
    monkeypatch.setattr(ops, "_empty", ga)
    out = model.encode_image(img)
    torch.cuda.synchronize()
    ga.check()
    assert tuple(out.shape) == (1024, 1024) and torch.isfinite(out.float()).all()
    pick = torch.tensor([0, 511, 1023], device=DEV)
    ga.bufs.clear()
    small = model.encode_image(img[pick].contiguous())
    if mode == "f16":
        assert out.dtype == torch.float16 and torch.equal(small, out[pick]), (small != out[pick]).sum().item()
    else:
        assert relerr(small.cpu(), out[pick].cpu()) < 1e-5


@pytest.mark.parametrize("mode", ["f16", "f32"])
def test_vit_b32_and_text_tower_inside_guard_zones(mode, monkeypatch):
    """ViT-B/32 at 512 images (the size bench.py's vit_b32 legs run) and the text tower on 8 prompts, both modes, every tensor the library
    allocates between sentinel zones; a random batch size too (ragged GEMM tiles and attention blocks in every layer)"""
    from test_gpu_headline import GuardedAlloc
    from dbmm_amd import synth
    from dbmm_amd.clip.model import build_model, convert_weights
    model = build_model(synth.clip_state_dict(5, "ViT-B/32")).cuda()
    if mode == "f16":
        model = convert_weights(model)
    img = synth.images(31, 64, 224).repeat(8, 1, 1, 1).contiguous().cuda()
    ga = GuardedAlloc()
    monkeypatch.setattr(ops, "_empty", ga)
    out = model.encode_image(img)
    odd = model.encode_image(img[:37].contiguous())
    tok = torch.randint(1, 49000, (8, 77), device=DEV, dtype=torch.int32)
    tok[:, 0] = 49406; tok[:, 20] = 49407; tok[:, 21:] = 0
    txt = model.encode_text(tok)
    torch.cuda.synchronize()
    ga.check()
    assert tuple(out.shape) == (512, 512) and torch.isfinite(out.float()).all() and torch.isfinite(txt.float()).all()
    tol = 2e-3 if mode == "f16" else 1e-5
    assert relerr(odd.float().cpu(), out[:37].float().cpu()) < tol
