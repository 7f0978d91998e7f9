"""BASELINE configs[3] / configs[4] at the size ONE GPU runs them (reference towers: clip/model.py:185-192, 223-240), and the fp16 /
fp16-pair transformer kernels on operands past 2 GiB.

  * ViT-B/32, 512 images in one call (one GPU's share of configs[3]), parity and fp16 mode: the reference golden's images sit at
    rows 0 / 511, fresh images in the middle must equal a batch-of-3 encode, and the deep-pipelined GEMMs that only serve
    M >= 16,384 token rows must be the kernels that ran;
  * ViT-L/14@336px fp16, 1024 images in one call (one GPU's share of configs[4]; bench.py's per-GPU default): golden image at
    rows 0 / 1023, fresh images at the rows where the qkv tensor (3.5 MB per image) and the MLP hidden tensor (4.7 MB per
    image) cross byte 2^31, against a small-batch encode;
  * gemm_f16 (eight-phase and two-barrier kernels), mha_core_f16, layernorm_f16 and mha_core (fp16-pair kernel) on operands
    > 2 GiB against fp64 at the first / straddling / last rows;
  * guard zones around the outputs of the attention cores and LayerNorm on ragged shapes.
"""
import pytest
import torch
import torch.nn.functional as F

from conftest import relerr
from dbmm_amd import ops, synth
from dbmm_amd.clip.model import build_model, convert_weights
from test_gpu_headline import GuardedAlloc  # noqa: F401

pytestmark = pytest.mark.gpu
DEV = "cuda"


def gname(arch, f16=False):
    return "clip_" + arch.replace("/", "-").replace("@", "-") + ("_f16" if f16 else "") + ".npz"


def _batch(seed, B, res, gold_img, fresh, fresh_rows):
    """B images: a 16-image base block tiled with per-block scales, the golden's images at rows 0 / B - 1, fresh images at fresh_rows"""
    base = synth.images(seed, 16, res)
    scale = torch.linspace(0.7, 1.3, B // 16).repeat_interleave(16).view(-1, 1, 1, 1)
    img = (base.repeat(B // 16, 1, 1, 1) * scale).contiguous()
    img[0], img[B - 1] = gold_img[0], gold_img[-1]
    for r, f in zip(fresh_rows, fresh):
        img[r] = f
    return img


@pytest.mark.parametrize("mode", ["f32", "f16"])
def test_vit_b32_512_images_one_call(mode, golden):
    arch, B = "ViT-B/32", 512
    g, h = golden(gname(arch)), golden(gname(arch, True))
    seed = int(g["seed"])
    model = build_model(synth.clip_state_dict(seed, arch)).cuda()
    if mode == "f16":
        convert_weights(model)
    gold_img = synth.images(seed + 100, int(g["batch"]), 224)
    fresh = synth.images(5151, 3, 224)
    rows = [255, 256, 257]
    img = _batch(991, B, 224, gold_img, fresh, rows).to(DEV)
    ops.profile_begin()
    out = model.encode_image(img)
    prof = ops.profile_end()
    assert tuple(out.shape) == (B, 512) and torch.isfinite(out).all()
    small = model.encode_image(img[255:258].contiguous())
    if mode == "f32":
        ref = torch.from_numpy(g["embedding"])
        e0, e1 = relerr(out[0:1].cpu(), ref[0:1]), relerr(out[B - 1:B].cpu(), ref[-1:])
        assert e0 < 5e-5 and e1 < 5e-5, f"rows 0 / 511 vs the reference golden: {e0:.2e} / {e1:.2e}"
        assert relerr(out[255:258].cpu(), small.cpu()) < 1e-5
        # 25,600 token rows: every projection on the split-precision MFMA kernels (nothing fell back to the fp32-input MFMA)
        big = [t for t in prof if prof[t][1] > 1e10]
        assert big and all(t.startswith(("igemm_x3_kernel<", "gemm_pair_8ph_kernel", "mha_pair_kernel")) for t in big), sorted(prof)
        assert "mha_pair_kernel" in prof
    else:
        ref16 = torch.from_numpy(h["embedding"])
        tol = 3.0 * float(h["f16_vs_f32"])
        e0, e1 = relerr(out[0:1].float().cpu(), ref16[0:1]), relerr(out[B - 1:B].float().cpu(), ref16[-1:])
        assert e0 < tol and e1 < tol, f"rows 0 / 511 vs the reference's fp16 path: {e0:.2e} / {e1:.2e} (tolerance {tol:.2e})"
        assert relerr(out[255:258].float().cpu(), small.float().cpu()) < 2e-3
        # the eight-phase GEMM serves every projection of the 25,600-row token stream (qkv, out + residual, fc + QuickGELU, proj + residual)
        for want in ("gemm_f16_8ph_kernel<0, 0>", "gemm_f16_8ph_kernel<0, 1>", "gemm_f16_8ph_kernel<2, 0>", "mha_f16_kernel"):
            assert want in prof, (want, sorted(prof))
        assert prof["gemm_f16_8ph_kernel<0, 1>"][0] == 24 and prof["gemm_f16_8ph_kernel<2, 0>"][0] == 12


def test_vit_l14_336_fp16_1024_images_one_call(golden):
    arch, B, R = "ViT-L/14@336px", 1024, 336
    h = golden(gname(arch, True))
    seed = int(h["seed"])
    model = convert_weights(build_model(synth.clip_state_dict(seed, arch)).cuda())
    L, E = 577, 1024
    qkv_row = 2 ** 31 // (L * 3 * E * 2)                 # the image whose qkv rows straddle byte 2^31 (3.5 MB per image)
    hid_row = 2 ** 31 // (L * 4 * E * 2)                 # ... and of the MLP hidden tensor (4.7 MB per image)
    assert qkv_row * L * 3 * E * 2 < 2 ** 31 < (qkv_row + 1) * L * 3 * E * 2 and B * L * E * 2 < 2 ** 31 < B * L * 3 * E * 2
    rows = [hid_row - 1, hid_row, qkv_row, qkv_row + 1]
    gold_img = synth.images(seed + 100, int(h["batch"]), R)
    fresh = synth.images(6161, 4, R)
    img = _batch(771, B, R, gold_img, fresh, rows).to(DEV)
    ops.profile_begin()
    out = model.encode_image(img)
    prof = ops.profile_end()
    assert tuple(out.shape) == (B, 768) and torch.isfinite(out).all()
    ref16 = torch.from_numpy(h["embedding"])
    tol = 3.0 * float(h["f16_vs_f32"])
    e0, e1 = relerr(out[0:1].float().cpu(), ref16[0:1]), relerr(out[B - 1:B].float().cpu(), ref16[-1:])
    print(f"ViT-L/14@336px fp16, 1024 images: rows 0 / 1023 vs the reference's fp16 path {e0:.2e} / {e1:.2e} (tolerance {tol:.2e})")
    assert e0 < tol and e1 < tol
    small = model.encode_image(torch.stack([img[r] for r in rows]).contiguous())
    es = relerr(out[rows].float().cpu(), small.float().cpu())
    assert es < 2e-3, f"rows {rows} of the 1024-image call vs the same images in a batch of 4: {es:.2e}"
    assert prof["gemm_f16_8ph_kernel<0, 1>"][0] == 48 and prof["mha_f16_kernel"][0] == 24
    del out, img
    torch.cuda.empty_cache()


def _straddle_rows(M, row_bytes, n=96):
    s = 2 ** 31 // row_bytes
    return torch.cat([torch.arange(0, n), torch.arange(s - n // 2, s + n // 2), torch.arange(M - n, M)]).to(DEV)


@pytest.mark.parametrize("deep", [1, 0])
def test_gemm_f16_on_operands_over_2gib(deep, option):
    """A, the residual and C each past 2 GiB (ViT-L/14@336px at 1024 images: qkv 3.6 GB, MLP hidden 4.8 GB): eight-phase kernel and
    the two-barrier kernel (option f16_8ph = 0), ragged M, against fp64 at the first / straddling / last rows"""
    option("f16_8ph", deep)
    g = torch.Generator(device=DEV); g.manual_seed(17 + deep)
    # --- wide A (K = 2048): A = 2.4 GB
    M, N, K = 590_848 + 77, 256, 2048
    a = torch.randn((M, K), device=DEV, generator=g, dtype=torch.float16)
    assert a.numel() * 2 > 2 ** 31
    w = (torch.randn((N, K), device=DEV, generator=g) * K ** -0.5).half(); b = torch.randn((N,), device=DEV, generator=g)
    out = ops.gemm_f16(a, w, b, act=2)
    rows = _straddle_rows(M, K * 2)
    v = a[rows].double() @ w.double().t() + b.double()
    assert torch.allclose(out[rows].double(), v * torch.sigmoid(1.702 * v), rtol=2e-3, atol=2e-3)
    del a, out
    torch.cuda.empty_cache()
    # --- wide C + residual (N = 2048): C and the residual = 2.4 GB each
    M, N, K = 590_848 + 77, 2048, 128
    a = torch.randn((M, K), device=DEV, generator=g, dtype=torch.float16)
    w = (torch.randn((N, K), device=DEV, generator=g) * K ** -0.5).half(); b = torch.randn((N,), device=DEV, generator=g)
    r = torch.randn((M, N), device=DEV, generator=g, dtype=torch.float16)
    assert r.numel() * 2 > 2 ** 31
    out = ops.gemm_f16(a, w, b, residual=r)
    rows = _straddle_rows(M, N * 2)
    v = a[rows].double() @ w.double().t() + b.double() + r[rows].double()
    assert torch.allclose(out[rows].double(), v, rtol=2e-3, atol=2e-3)
    # every tile wrote its rows: column sums of random row blocks are non-zero everywhere
    assert bool((out[-300:].float().abs().sum(1) > 0).all()) and bool((out[M // 2:M // 2 + 300].float().abs().sum(1) > 0).all())


def _mha_ref(qkv_img, L, heads, causal=False):
    E = heads * 64
    q, k, v = (t.view(L, heads, 64).permute(1, 0, 2).double() for t in qkv_img.view(L, 3, E).unbind(1))
    s = q @ k.transpose(-1, -2) / 8.0
    if causal:
        s = s + torch.full((L, L), float("-inf"), device=s.device, dtype=torch.float64).triu(1)
    return (torch.softmax(s, -1) @ v).permute(1, 0, 2).reshape(L, E)


def test_mha_core_f16_on_operands_over_2gib():
    B, L, heads = 640, 577, 16                       # qkv = 640 x 577 x 3072 fp16 = 2.27 GB
    E = heads * 64
    g = torch.Generator(device=DEV); g.manual_seed(23)
    qkv = (torch.randn((B * L, 3 * E), device=DEV, generator=g) * 1.5).half()
    assert qkv.numel() * 2 > 2 ** 31
    out = ops.mha_core_f16(qkv, B, L, E, heads, False)
    s = 2 ** 31 // (L * 3 * E * 2)
    for i in (0, s - 1, s, s + 1, B - 1):
        ref = _mha_ref(qkv[i * L:(i + 1) * L], L, heads)
        assert relerr(out[i * L:(i + 1) * L].double().cpu(), ref.cpu()) < 3e-3, i


def test_mha_core_pair_kernel_on_operands_over_2gib():
    B, L, heads = 320, 577, 16                       # fp32 qkv = 320 x 577 x 3072 x 4 B = 2.27 GB
    E = heads * 64
    g = torch.Generator(device=DEV); g.manual_seed(29)
    qkv = torch.randn((B * L, 3 * E), device=DEV, generator=g) * 1.5
    assert qkv.numel() * 4 > 2 ** 31
    am = (qkv.abs().max() * 1.2).reshape(1)
    out = ops.mha_core(qkv, B, L, E, heads, False, qkv_absmax=am)
    s = 2 ** 31 // (L * 3 * E * 4)
    for i in (0, s - 1, s, s + 1, B - 1):
        ref = _mha_ref(qkv[i * L:(i + 1) * L], L, heads)
        assert relerr(out[i * L:(i + 1) * L].double().cpu(), ref.cpu()) < 2e-5, i


def test_layernorm_f16_on_operands_over_2gib():
    rows, E = 1_100_003, 1024                        # 2.25 GB in, 2.25 GB out
    g = torch.Generator(device=DEV); g.manual_seed(31)
    x = (torch.randn((rows, E), device=DEV, generator=g) * 3 + 0.5).half()
    assert x.numel() * 2 > 2 ** 31
    ga, be = torch.randn((E,), device=DEV, generator=g), torch.randn((E,), device=DEV, generator=g)
    out = ops.layernorm_f16(x, ga, be)
    sel = _straddle_rows(rows, E * 2)
    ref = F.layer_norm(x[sel].double(), (E,), ga.double(), be.double(), 1e-5)
    assert relerr(out[sel].double().cpu(), ref.cpu()) < 1e-3
    # token-0 gather form (ln_post: rows = images, pitch = L * E) reaching past 2 GiB
    Bn, Ltok = rows // 577, 577
    out0 = ops.layernorm_f16(x.view(-1), ga, be, rows=Bn, ldx=Ltok * E)
    ref0 = F.layer_norm(x[:Bn * Ltok].view(Bn, Ltok, E)[:, 0].double(), (E,), ga.double(), be.double(), 1e-5)
    assert relerr(out0.double().cpu(), ref0.cpu()) < 1e-3


@pytest.fixture
def guarded(monkeypatch):
    ga = GuardedAlloc()
    monkeypatch.setattr(ops, "_empty", ga)
    return ga


def test_attention_and_layernorm_outputs_stay_inside_their_buffers(guarded):
    """ragged shapes of the kernels the transformer towers launch: sequence lengths that are not tile multiples (L = 577, 50, 77, 1),
    row counts that are not wave multiples; every output sits between sentinel zones"""
    g = torch.Generator(device=DEV); g.manual_seed(41)
    for (B, L, heads, causal) in ((3, 577, 2, False), (5, 50, 3, False), (2, 77, 1, True), (7, 1, 1, False), (1, 129, 2, True)):
        E = heads * 64
        q16 = (torch.randn((B * L, 3 * E), device=DEV, generator=g) * 1.5).half()
        o16 = ops.mha_core_f16(q16, B, L, E, heads, causal)
        q32 = torch.randn((B * L, 3 * E), device=DEV, generator=g) * 1.5
        o32 = ops.mha_core(q32, B, L, E, heads, causal, qkv_absmax=(q32.abs().max() * 1.1).reshape(1))
        for i in (0, B - 1):
            assert relerr(o16[i * L:(i + 1) * L].double().cpu(), _mha_ref(q16[i * L:(i + 1) * L], L, heads, causal).cpu()) < 3e-3
            assert relerr(o32[i * L:(i + 1) * L].double().cpu(), _mha_ref(q32[i * L:(i + 1) * L], L, heads, causal).cpu()) < 2e-5
    for rows, E in ((7, 768), (1, 1024), (130, 512), (3, 4096)):
        x = torch.randn((rows, E), device=DEV, generator=g)
        ga, be = torch.randn((E,), device=DEV, generator=g), torch.randn((E,), device=DEV, generator=g)
        y16 = ops.layernorm_f16(x.half(), ga, be)
        y32 = ops.layernorm(x, ga, be, y_absmax=torch.zeros(1, device=DEV))
        ref = F.layer_norm(x.double(), (E,), ga.double(), be.double(), 1e-5)
        assert relerr(y32.double().cpu(), ref.cpu()) < 1e-5 and relerr(y16.double().cpu(), F.layer_norm(x.half().double(), (E,), ga.double(), be.double(), 1e-5).cpu()) < 1e-3
    torch.cuda.synchronize()
    guarded.check()


def test_fp16_resnet_kernels_on_tensors_over_2gib():
    """the fp16 ModifiedResNet kernels of round 4 where a tensor passes 2 GiB (fp16 RN50 from B = 1338 at layer 1, from B = 2675 at the
    stem): every tile rebases its descriptors on a 64-bit base, so the images around byte 2^31 and the last one must match fp64 --
    chain_f16_kernel (plain, and the seam mode whose pooled copy is written from window-major tiles), conv3x3_c32_f16_kernel,
    stem_s2_f16_mfma_kernel (clip/model.py:42-55, 108-116, 141-148)."""
    g = torch.Generator(device=DEV); g.manual_seed(77)
    bn = lambda n: (0.5 + torch.rand((n,), device=DEV, generator=g), torch.randn((n,), device=DEV, generator=g) * 0.1)
    # ---- conv3 + residual -> next conv1: the 256-channel layer-1 map, 1,605,632 B per image ----
    B, H, K, N, P = 1400, 56, 64, 256, 128
    y2 = torch.relu(torch.randn((B, H, H, K), device=DEV, generator=g)).half()
    res = torch.relu(torch.randn((B, H, H, N), device=DEV, generator=g, dtype=torch.float16))
    assert res.numel() * 2 > 2 ** 31
    w3 = (torch.randn((N, K), device=DEV, generator=g) * K ** -0.5).half(); w1 = (torch.randn((P, N), device=DEV, generator=g) * N ** -0.5).half()
    (s3, b3), (s1, b1) = bn(N), bn(P)
    x, y1 = ops.chain_f16(y2, (w3, s3, b3), res, (w1, s1, b1))
    xs, xp, y1s = ops.chain_f16(y2, (w3, s3, b3), res, (w1, s1, b1), pooled=True)
    s = 2 ** 31 // (H * H * N * 2)
    for i in (0, s - 1, s, s + 1, B - 1):
        xr = torch.relu(y2[i].double().view(-1, K) @ w3.double().t() * s3.double() + b3.double() + res[i].double().view(-1, N))
        assert torch.allclose(x[i].double().view(-1, N), xr, rtol=2e-3, atol=2e-3), i
        y1r = torch.relu(x[i].double().view(-1, N) @ w1.double().t() * s1.double() + b1.double())
        assert torch.allclose(y1[i].double().view(-1, P), y1r, rtol=2e-3, atol=2e-3), i
        assert torch.equal(xs[i], x[i]) and torch.equal(y1s[i], y1[i]) and torch.equal(xp[i], ops.avgpool2_f16(x[i:i + 1])[0]), i
    del y2, res, x, y1, xs, xp, y1s
    # ---- the 32-channel stem convs: 802,816 B per image in, the same out (plain) ----
    B, R = 2800, 112
    xin = torch.relu(torch.randn((B, R, R, 32), device=DEV, generator=g, dtype=torch.float16))
    assert xin.numel() * 2 > 2 ** 31
    w = (torch.randn((32, 32, 3, 3), device=DEV, generator=g) * 288 ** -0.5).half()
    wp, _ = ops.pack_conv_weight(w.float(), chunk_major=32)
    sc, b = bn(32)
    ops.profile_begin()
    y = ops.conv3x3_f16(xin, wp.half().contiguous(), sc, b)
    yp = ops.conv3x3_f16(xin, wp.half().contiguous(), sc, b, pool=2)
    assert list(ops.profile_end()) == ["conv3x3_c32_f16_kernel<32, 0>", "conv3x3_c32_f16_kernel<32, 1>"]
    s = 2 ** 31 // (R * R * 32 * 2)
    for i in (0, s - 1, s, s + 1, B - 1):
        ref = torch.relu(F.conv2d(xin[i].permute(2, 0, 1)[None].double(), w.double(), None, padding=1) * sc.double().view(1, -1, 1, 1) + b.double().view(1, -1, 1, 1))
        assert relerr(y[i].double().cpu(), ref[0].permute(1, 2, 0).cpu()) < 1.5e-3, i
        assert relerr(yp[i].double().cpu(), F.avg_pool2d(ref, 2)[0].permute(1, 2, 0).cpu()) < 1.5e-3, i
    del xin, y, yp
    # ---- stem conv1: the fp32 NCHW image batch itself passes 2 GiB (602,112 B per image) ----
    B, R = 3600, 224
    img = torch.randn((B, 3, R, R), device=DEV, generator=g)
    assert img.numel() * 4 > 2 ** 31
    w = (torch.randn((3, 3, 3, 32), device=DEV, generator=g) * 0.2).half().float()
    sc, b = bn(32)
    y = ops.conv_stem_s2_f16(img, w, b, sc)
    s = 2 ** 31 // (3 * R * R * 4)
    for i in (0, s - 1, s, s + 1, B - 1):
        ref = F.conv2d(img[i:i + 1].half().double(), w.permute(3, 2, 0, 1).double(), None, stride=2, padding=1) * sc.double().view(1, -1, 1, 1) + b.double().view(1, -1, 1, 1)
        assert relerr(y[i].double().cpu(), torch.relu(ref)[0].permute(1, 2, 0).cpu()) < 5e-4, i
