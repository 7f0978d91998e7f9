"""fp16 throughput mode (csrc/f16_ops.hip; the reference's GPU path, convert_weights + fp16 activations,
clip/model.py:146, 157-163, 341, 375-396).

Pinned to the reference's OWN fp16 path: oracle/make_golden.py builds the reference model without `.float()` (fp16
weights and activations, fp32 LayerNorm statistics, fp32 BatchNorm parameters) and runs it on the build container's
CPU; tests/golden/clip_<arch>_f16.npz hold its image / text embeddings and `f16_vs_f32`, the distance of that fp16
result from the reference's fp32 result.  Two fp16 evaluations of the same network differ by their rounding histories
(where a tensor is rounded to fp16, the order of the fp32 accumulation), i.e. by about sqrt(2) x that distance; the
tower tests hold the HIP path to 3 x `f16_vs_f32` (2.4e-3 ... 3.3e-3 of the embedding maximum) and print what they
measured.  Kernel tests: every fp16 kernel against an fp64 evaluation of the same op on the SAME fp16-rounded inputs --
the only differences are fp32 accumulation order and the single fp16 rounding of the stored result (2^-11 relative)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import relerr
from dbmm_amd import ops, synth
from dbmm_amd.clip.model import build_model, convert_weights

pytestmark = pytest.mark.gpu
DEV = "cuda"


def gname(arch):
    return "clip_" + arch.replace("/", "-").replace("@", "-") + ".npz"


@pytest.mark.parametrize("M,N,K,res,act", [(300, 256, 128, False, 0), (1000, 768, 768, True, 0), (577 * 3, 3072, 1024, False, 2),
                                           (128, 64, 64, True, 1), (4, 512, 768, False, 0), (25600, 2304, 768, False, 0)])
def test_gemm_f16(M, N, K, res, act):
    g = torch.Generator(device=DEV); g.manual_seed(M + N + K)
    a = torch.randn((M, K), device=DEV, generator=g).half(); w = (torch.randn((N, K), device=DEV, generator=g) * K ** -0.5).half()
    b = torch.randn((N,), device=DEV, generator=g); r = torch.randn((M, N), device=DEV, generator=g).half() if res else None
    v = a.double() @ w.double().t() + b.double()
    v = {0: v, 1: torch.relu(v), 2: v * torch.sigmoid(1.702 * v)}[act]
    if res:
        v = v + r.double()
    out = ops.gemm_f16(a, w, b, residual=r, act=act)
    assert out.dtype == torch.float16 and tuple(out.shape) == (M, N)
    assert relerr(out.double().cpu(), v.cpu()) < 1.5e-3           # 2^-11 output rounding + fp32 accumulation


@pytest.mark.parametrize("M,N,K,res,act", [(16384 + 77, 512, 256, True, 2), (32768, 256, 128, False, 0), (20000, 1024, 4096, True, 0),
                                           (577 * 64, 3072, 1024, False, 2), (16384, 768, 3072, True, 0),
                                           # a short last round cut along K (tail_split): 300 tiles = 256 + 44 x 5 slices (the ViT-B/32 projections at
                                           # 512 images), 274 = 256 + 18 x 4; above: 316 = 256 + 60 x 4
                                           (25600, 768, 768, True, 0), (25600, 768, 3072, True, 0), (70000, 256, 512, False, 1)])
def test_gemm_f16_deep_pipelined_kernel(M, N, K, res, act, option):
    """the 256 x 256 x 64 eight-phase kernel (N % 256 == 0, K % 128 == 0, M >= 16384) against fp64 ELEMENT-wise -- a staging
    race would show as a few wrong tiles, which a norm-wise error hides -- over repeated launches, and against the
    two-barrier kernel (DBMM_F16_8PH=0) on the same operands"""
    g = torch.Generator(device=DEV); g.manual_seed(M + N + K)
    a = torch.randn((M, K), device=DEV, generator=g).half(); w = (torch.randn((N, K), device=DEV, generator=g) * K ** -0.5).half()
    b = torch.randn((N,), device=DEV, generator=g); r = torch.randn((M, N), device=DEV, generator=g).half() if res else None
    rows = torch.cat([torch.arange(0, 600, device=DEV), torch.randint(0, M, (3000,), device=DEV, generator=g), torch.arange(M - 300, M, device=DEV)])
    v = a[rows].double() @ w.double().t() + b.double()
    v = {0: v, 1: torch.relu(v), 2: v * torch.sigmoid(1.702 * v)}[act]
    if res:
        v = v + r[rows].double()
    option("f16_8ph", "0")
    base = ops.gemm_f16(a, w, b, residual=r, act=act)
    option("f16_8ph", "1")
    for it in range(4):
        option("tail_split", 2 if it & 1 else 1)       # 1: by the library's rule, 2: a short last round's tiles cut along K wherever that applies
        out = ops.gemm_f16(a, w, b, residual=r, act=act)
        assert torch.allclose(out[rows].double(), v, rtol=2e-3, atol=2e-3)
        # same products, fp32 accumulation in another order, one fp16 rounding: at most an ulp or two apart, everywhere
        assert (out.float() - base.float()).abs().max().item() <= 4e-3 * max(1.0, base.float().abs().max().item())
        assert (out != base).float().mean().item() < 0.05


def test_gemm_f16_strided_rows_and_rejects():
    g = torch.Generator(device=DEV); g.manual_seed(5)
    t = torch.randn((9, 7, 64), device=DEV, generator=g).half(); w = torch.randn((32, 64), device=DEV, generator=g).half()
    out = ops.gemm_f16(t, w, M=9, lda=7 * 64)                                      # token 0 of every image (ln_post -> proj)
    assert relerr(out.double().cpu(), (t[:, 0].double() @ w.double().t()).cpu()) < 1.5e-3
    from dbmm_amd import _lib
    with pytest.raises(_lib.DbmmError):
        ops.gemm_f16(torch.zeros((8, 40), device=DEV, dtype=torch.float16), torch.zeros((16, 40), device=DEV, dtype=torch.float16))   # K % 64
    with pytest.raises(_lib.DbmmError):
        ops.gemm_f16(torch.zeros((8, 64), device=DEV), torch.zeros((16, 64), device=DEV, dtype=torch.float16))                        # f32 input


@pytest.mark.parametrize("B,L,heads,causal", [(3, 50, 2, False), (2, 77, 8, True), (2, 130, 1, False), (1, 577, 4, False),
                                              (2, 1, 1, False), (1, 128, 2, True), (2, 129, 1, True),
                                              (2, 64, 2, True), (3, 33, 1, True)])          # <= 64 tokens: the two-wave workgroups
def test_mha_core_f16(B, L, heads, causal):
    E = heads * 64
    g = torch.Generator(device=DEV); g.manual_seed(L + heads)
    qkv = (torch.randn((B * L, 3 * E), device=DEV, generator=g) * 1.5).half()
    out = ops.mha_core_f16(qkv, B, L, E, heads, causal)
    q, k, v = (t.view(B, L, heads, 64).permute(0, 2, 1, 3).double() for t in qkv.view(B, L, 3, E).unbind(2))
    s = q @ k.transpose(-1, -2) / 8.0
    if causal:
        s = s + torch.full((L, L), float("-inf"), device=DEV, dtype=torch.float64).triu(1)
    ref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B * L, E)
    assert out.dtype == torch.float16 and relerr(out.double().cpu(), ref.cpu()) < 3e-3   # P is rounded to fp16 before P V


def test_mha_core_f16_spiked_scores():
    B, L, heads, E = 1, 200, 2, 128
    g = torch.Generator(device=DEV); g.manual_seed(1)
    qkv = torch.randn((B * L, 3 * E), device=DEV, generator=g)
    qkv[:, :E] *= 12.0                                                             # softmax close to one-hot
    qkv = qkv.half()
    out = ops.mha_core_f16(qkv, B, L, E, heads, False)
    q, k, v = (t.view(B, L, heads, 64).permute(0, 2, 1, 3).double() for t in qkv.view(B, L, 3, E).unbind(2))
    ref = (torch.softmax(q @ k.transpose(-1, -2) / 8.0, -1) @ v).permute(0, 2, 1, 3).reshape(B * L, E)
    assert torch.isfinite(out).all() and relerr(out.double().cpu(), ref.cpu()) < 3e-3


@pytest.mark.parametrize("rows,E", [(7, 64), (100, 768), (33, 1024), (5, 2048), (3, 4096)])
def test_layernorm_f16(rows, E):
    g = torch.Generator(device=DEV); g.manual_seed(rows + E)
    x = (torch.randn((rows, E), device=DEV, generator=g) * 3 + 0.5).half()
    ga, be = torch.randn((E,), device=DEV, generator=g), torch.randn((E,), device=DEV, generator=g)
    out = ops.layernorm_f16(x, ga, be)
    ref = F.layer_norm(x.double(), (E,), ga.double(), be.double(), 1e-5)
    assert out.dtype == torch.float16 and relerr(out.double().cpu(), ref.cpu()) < 1e-3
    strided = ops.layernorm_f16(x.view(-1), ga[:64].contiguous(), be[:64].contiguous(), rows=rows, ldx=E) if E >= 64 else None
    if strided is not None:
        assert relerr(strided.double().cpu(), F.layer_norm(x[:, :64].double(), (64,), ga[:64].double(), be[:64].double(), 1e-5).cpu()) < 1e-3


def gname16(arch):
    return gname(arch).replace(".npz", "_f16.npz")


@pytest.mark.parametrize("arch", ["tiny-ViT", "ViT-B/32", "ViT-L/14@336px"])
def test_fp16_mode_towers_vs_reference_fp16_path(arch, golden):
    g, h = golden(gname(arch)), golden(gname16(arch))
    seed, B, res = int(h["seed"]), int(h["batch"]), int(h["res"])
    model = convert_weights(build_model(synth.clip_state_dict(seed, arch)).cuda())
    assert model.dtype == torch.float16
    img = synth.images(seed + 100, B, res).cuda()
    out = model.encode_image(img)
    assert out.dtype == torch.float16 and tuple(out.shape) == h["embedding"].shape
    ref16, ref32 = torch.from_numpy(h["embedding"]), torch.from_numpy(g["embedding"])
    tol = 3.0 * float(h["f16_vs_f32"])
    e16, e32 = relerr(out.float().cpu(), ref16), relerr(out.float().cpu(), ref32)
    print(f"{arch}: image vs reference fp16 path {e16:.2e} (tol {tol:.2e}), vs fp32 golden {e32:.2e}")
    assert e16 < tol, f"{arch}: image embedding {e16:.2e} from the reference's fp16 path (tolerance {tol:.2e})"
    assert e32 < tol, f"{arch}: image embedding {e32:.2e} from the fp32 golden (tolerance {tol:.2e})"
    assert F.cosine_similarity(out.float().cpu(), ref16, dim=1).min() > 0.99999
    txt = model.encode_text(torch.from_numpy(h["tokens"]).cuda())
    tref16 = torch.from_numpy(h["text_embedding"])
    ttol = 3.0 * float(h["text_f16_vs_f32"])
    t16 = relerr(txt.float().cpu(), tref16)
    print(f"{arch}: text vs reference fp16 path {t16:.2e} (tol {ttol:.2e})")
    assert txt.dtype == torch.float16 and t16 < ttol, f"{arch}: text embedding {t16:.2e} (tolerance {ttol:.2e})"
    assert F.cosine_similarity(txt.float().cpu(), tref16, dim=1).min() > 0.99999
    # back to the parity mode
    model.float()
    assert model.dtype == torch.float32 and relerr(model.encode_image(img).cpu(), ref32) < 5e-5


def test_fp16_mode_rn_towers_without_fp16_kernels_round_their_fp32_accurate_output(golden):
    """widths the fp16 conv kernels do not serve (not a multiple of 64: RN50x4, RN50x16, the toy towers) keep the fp32-accurate
    plan on the fp16-stored weights and round the embedding; still within the reference fp16 path's own rounding distance"""
    for arch in ("tiny-RN", "tiny-RN-w32"):
        h = golden(gname16(arch))
        seed, B, res = int(h["seed"]), int(h["batch"]), int(h["res"])
        model = build_model(synth.clip_state_dict(seed, arch)).cuda()
        img = synth.images(seed + 100, B, res).cuda()
        ref = model.encode_image(img)
        convert_weights(model)
        out = model.encode_image(img)
        assert out.dtype == torch.float16 and relerr(out.float().cpu(), ref.cpu()) < 1e-3
        assert relerr(out.float().cpu(), torch.from_numpy(h["embedding"])) < 3.0 * float(h["f16_vs_f32"])


# ---- fp16 mode of the ModifiedResNet towers (csrc/conv_f16.hip) -----------------------------------------------------------

def _bn(g, n):
    return 0.5 + torch.rand((n,), device=DEV, generator=g), torch.randn((n,), device=DEV, generator=g) * 0.1


@pytest.mark.parametrize("B,H,W,Cin,Cout,pool", [(3, 14, 14, 256, 256, 1), (2, 56, 56, 64, 64, 1), (5, 7, 7, 512, 512, 1), (2, 28, 28, 128, 128, 2),
                                                 (3, 12, 20, 32, 32, 1), (2, 16, 24, 32, 64, 2), (9, 10, 6, 64, 192, 1), (4, 14, 14, 96, 136, 2),
                                                 (1, 2, 2, 32, 8, 2), (1, 1, 1, 64, 64, 1), (40, 28, 28, 128, 128, 1), (33, 14, 14, 256, 256, 2)])
def test_conv3x3_f16_kernel(B, H, W, Cin, Cout, pool):
    """3x3 / stride 1 / pad 1 conv + BatchNorm + ReLU (+ AvgPool2d(2)) with fp16 operands against fp64 on the SAME fp16 values:
    every tile geometry (256 x 128, 512 x 64, 512 x 32), image borders on 1x1 ... 56x56 maps, tiles that span several images
    and end in a ragged tail, Cout tails, one / many channel slabs, pooled (window-major) and standard row order."""
    g = torch.Generator(device=DEV); g.manual_seed(B * 131 + H * 7 + Cin + Cout + pool)
    x = torch.relu(torch.randn((B, H, W, Cin), device=DEV, generator=g) * 1.5).half()
    w = (torch.randn((Cout, Cin, 3, 3), device=DEV, generator=g) * (9 * Cin) ** -0.5).half()
    sc, b = _bn(g, Cout)
    wp, wl = ops.pack_conv_weight(w.float(), chunk_major=32)
    assert wl == ops.WL_CHUNK32_MAJOR
    y = ops.conv3x3_f16(x, wp.half().contiguous(), sc, b, pool=pool)
    assert y is not None and y.dtype == torch.float16
    ref = torch.relu(F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, padding=1) * sc.double().view(1, -1, 1, 1) + b.double().view(1, -1, 1, 1))
    if pool == 2:
        ref = F.avg_pool2d(ref, 2)
    ref = ref.permute(0, 2, 3, 1)
    assert tuple(y.shape) == tuple(ref.shape)
    assert relerr(y.double().cpu(), ref.cpu()) < 1.5e-3            # one fp16 rounding of the result + fp32 accumulation


@pytest.mark.parametrize("B,H,W,Cin,Cout,pool", [
    (128, 14, 14, 256, 256, 1),          # layer 3's conv2: 98 whole tiles
    (512, 7, 7, 512, 512, 1),            # layer 4's: two N tiles per row block, 392 tiles = two per workgroup, 7 x 7 maps
    (131, 14, 14, 64, 256, 1),           # ragged last tile, two slabs
    (30, 26, 30, 96, 256, 1),            # non-square maps, three slabs
    (24, 28, 28, 256, 256, 2),           # layer 3's first conv2: pooled, window-major rows
    (100, 14, 14, 512, 512, 2),          # layer 4's first conv2
    (33, 26, 26, 64, 256, 2),            # pooled, ragged, windows wrap over pooled rows inside a tile
    (700, 6, 6, 32, 256, 2),             # one slab; every tile wraps over rows and images
    (3, 80, 72, 64, 256, 1),             # a tile inside one image row block (W > 64)
])
def test_conv3x3_f16_eight_phase_kernel(B, H, W, Cin, Cout, pool, option):
    """conv3x3_f16_8ph_kernel (Cout % 256 == 0, >= 16,384 pixels): the eight-phase 256 x 256 variant keeps conv3x3_f16_kernel's arithmetic and
    order of accumulation, so the two agree BIT FOR BIT; both against fp64 on the same fp16 values; guard zone behind the output."""
    g = torch.Generator(device=DEV); g.manual_seed(B * 131 + H * 7 + Cin + Cout + pool)
    x = torch.relu(torch.randn((B, H, W, Cin), device=DEV, generator=g) * 1.5).half()
    w = (torch.randn((Cout, Cin, 3, 3), device=DEV, generator=g) * (9 * Cin) ** -0.5).half()
    sc, b = _bn(g, Cout)
    wp, wl = ops.pack_conv_weight(w.float(), chunk_major=32)
    wh = wp.half().contiguous()
    option("f16_conv_8ph", 1)
    ops.profile_begin()
    y8 = ops.conv3x3_f16(x, wh, sc, b, pool=pool)
    assert list(ops.profile_end()) == [f"conv3x3_f16_8ph_kernel<{int(pool == 2)}>"]
    option("f16_conv_8ph", 0)
    y = ops.conv3x3_f16(x, wh, sc, b, pool=pool)
    assert torch.equal(y8, y), ((y8 != y).sum().item(), relerr(y8.double().cpu(), y.double().cpu()))
    ref = torch.relu(F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, padding=1) * sc.double().view(1, -1, 1, 1) + b.double().view(1, -1, 1, 1))
    if pool == 2:
        ref = F.avg_pool2d(ref, 2)
    assert relerr(y8.double().cpu(), ref.permute(0, 2, 3, 1).cpu()) < 1.5e-3
    from dbmm_amd import _lib
    n_out = y.numel()
    buf = torch.full((n_out + 4096,), 7.0, device=DEV, dtype=torch.float16)
    option("f16_conv_8ph", 1)
    rc = _lib.lib().dbmm_conv3x3_bn_relu_f16(x.data_ptr(), wh.data_ptr(), sc.data_ptr(), b.data_ptr(), buf.data_ptr(), B, H, W, Cin, Cout, 2 if pool == 2 else 0,
                                             _lib.stream())
    torch.cuda.synchronize()
    assert rc == 0 and torch.equal(buf[:n_out].view(y.shape), y) and (buf[n_out:] == 7.0).all()


@pytest.mark.parametrize("B,H,W,Cout,pool", [(3, 112, 112, 32, 1), (3, 112, 112, 64, 2), (2, 8, 28, 32, 2), (5, 4, 56, 64, 1), (1, 4, 28, 32, 1),
                                             (40, 20, 84, 64, 2), (300, 12, 28, 32, 1)])
def test_conv3x3_c32_f16_patch_kernel(B, H, W, Cout, pool, option):
    """the 32-channel stem convs (clip/model.py:108-116, 141-143: conv2 / conv3 + bn + relu, conv3 followed by AvgPool2d(2)) on the persistent
    patch kernel: same products and K order as conv3x3_f16_kernel, so the two agree BIT FOR BIT; both against fp64 on the same fp16 values;
    one tile per image row block ... more tiles than workgroup slots (B = 300: 900 tiles over 1024 slots; B = 40 x 15 = 600), image borders on
    every side of a tile, guard zone behind the output"""
    Cin = 32
    g = torch.Generator(device=DEV); g.manual_seed(B * 131 + H * 7 + W + Cout + pool)
    x = torch.relu(torch.randn((B, H, W, Cin), device=DEV, generator=g) * 1.5).half()
    w = (torch.randn((Cout, Cin, 3, 3), device=DEV, generator=g) * (9 * Cin) ** -0.5).half()
    sc, b = _bn(g, Cout)
    wp, wl = ops.pack_conv_weight(w.float(), chunk_major=32)
    wh = wp.half().contiguous()
    ops.profile_begin()
    yp = ops.conv3x3_f16(x, wh, sc, b, pool=pool)
    assert list(ops.profile_end()) == [f"conv3x3_c32_f16_kernel<{Cout}, {int(pool == 2)}>"]
    option("conv_patch", 0)
    y = ops.conv3x3_f16(x, wh, sc, b, pool=pool)
    assert torch.equal(yp, y), ((yp != y).sum().item(), relerr(yp.double().cpu(), y.double().cpu()))
    ref = torch.relu(F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, padding=1) * sc.double().view(1, -1, 1, 1) + b.double().view(1, -1, 1, 1))
    if pool == 2:
        ref = F.avg_pool2d(ref, 2)
    assert relerr(yp.double().cpu(), ref.permute(0, 2, 3, 1).cpu()) < 1.5e-3
    option("conv_patch", 1)
    from dbmm_amd import _lib
    n_out = y.numel()
    buf = torch.full((n_out + 4096,), 7.0, device=DEV, dtype=torch.float16)
    rc = _lib.lib().dbmm_conv3x3_bn_relu_f16(x.data_ptr(), wh.data_ptr(), sc.data_ptr(), b.data_ptr(), buf.data_ptr(), B, H, W, Cin, Cout, 2 if pool == 2 else 0,
                                             _lib.stream())
    torch.cuda.synchronize()
    assert rc == 0 and torch.equal(buf[:n_out].view(y.shape), y) and (buf[n_out:] == 7.0).all()


@pytest.mark.parametrize("M,Cin,Cout,res,act", [(56 * 56 * 3, 64, 256, True, 1), (28 * 28 * 5, 512, 128, False, 1), (14 * 14 * 90, 1024, 256, False, 1),
                                                (14 * 14 * 90 + 12, 256, 1024, True, 1), (7 * 7 * 40, 2048, 512, False, 1), (300, 64, 64, False, 0),
                                                (17000, 128, 512, True, 1), (256 * 3 + 77, 96, 136, True, 1), (5, 32, 8, False, 1)])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_conv1x1_f16(M, Cin, Cout, res, act, mode, option):
    """1x1 conv + BatchNorm (+ residual added BEFORE the ReLU, clip/model.py:53-54) vs fp64: on the two fp16 GEMM kernels
    (conv1x1_stream = 0), the streaming kernel everywhere (2) and the default routing (1)"""
    option("conv1x1_stream", mode)
    g = torch.Generator(device=DEV); g.manual_seed(M + Cin + Cout)
    x = torch.relu(torch.randn((M, Cin), device=DEV, generator=g)).half()
    w = (torch.randn((Cout, Cin), device=DEV, generator=g) * Cin ** -0.5).half()
    sc, b = _bn(g, Cout)
    r = (torch.randn((M, Cout), device=DEV, generator=g) * 2.0).half() if res else None
    y = ops.conv1x1_f16(x, w, sc, b, residual=r, act=act)
    v = x.double() @ w.double().t() * sc.double() + b.double()
    if res:
        v = v + r.double()
    if act == 1:
        v = torch.relu(v)
    if y is None:
        assert mode == 0 and Cin % 64                       # the GEMM kernels need K % 64 == 0
        return
    assert y.dtype == torch.float16 and relerr(y.double().cpu(), v.cpu()) < 1.5e-3


@pytest.mark.parametrize("B,R,Cout,half_in", [(3, 64, 32, False), (2, 224, 32, True), (2, 96, 64, False), (1, 50, 32, False)])
def test_stem_and_avgpool_f16(B, R, Cout, half_in):
    g = torch.Generator(device=DEV); g.manual_seed(R + Cout)
    x = torch.randn((B, 3, R, R), device=DEV, generator=g)
    w = torch.randn((3, 3, 3, Cout), device=DEV, generator=g) * 0.2; b = torch.randn((Cout,), device=DEV, generator=g) * 0.1
    y = ops.conv_stem_s2_f16(x.half() if half_in else x, w, b)
    ref = torch.relu(F.conv2d(x.half().double(), w.permute(3, 2, 0, 1).double(), b.double(), stride=2, padding=1)).permute(0, 2, 3, 1)
    assert y.dtype == torch.float16 and tuple(y.shape) == tuple(ref.shape) and relerr(y.double().cpu(), ref.cpu()) < 1e-3
    if y.shape[1] % 2 == 0:
        p = ops.avgpool2_f16(y)
        assert relerr(p.double().cpu(), F.avg_pool2d(y.double().permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1).cpu()) < 1e-3


@pytest.mark.parametrize("B,R,Cout,half_in", [(3, 64, 32, False), (2, 224, 32, True), (5, 38, 64, False), (1, 18, 32, True), (130, 32, 32, False)])
def test_stem_f16_mfma_gather_kernel(B, R, Cout, half_in, option):
    """stem conv1 + BatchNorm + ReLU in fp16 mode on the MFMA gather kernel (clip/model.py:146-148 with fp16 weights): against fp64 of the
    fp16-rounded image x fp16 weights, and against the FMA kernel given the same operands (same exact products, another order of the fp32
    sums); odd output sides (R = 38, 18), blocks of 32 pixels straddling images, a ragged last block, guard zone behind the output"""
    g = torch.Generator(device=DEV); g.manual_seed(R + Cout + B)
    x = torch.randn((B, 3, R, R), device=DEV, generator=g)
    w = (torch.randn((3, 3, 3, Cout), device=DEV, generator=g) * 0.2).half().float()
    sc = 0.5 + torch.rand((Cout,), device=DEV, generator=g); b = torch.randn((Cout,), device=DEV, generator=g) * 0.1
    xin = x.half() if half_in else x
    y = ops.conv_stem_s2_f16(xin, w, b, sc)
    ref = F.conv2d(x.half().double(), w.permute(3, 2, 0, 1).double(), None, stride=2, padding=1) * sc.double().view(1, -1, 1, 1) + b.double().view(1, -1, 1, 1)
    ref = torch.relu(ref).permute(0, 2, 3, 1)
    assert y.dtype == torch.float16 and tuple(y.shape) == tuple(ref.shape) and relerr(y.double().cpu(), ref.cpu()) < 5e-4
    assert (y.double() - ref).abs().max().item() <= 2e-3 * max(1.0, ref.abs().max().item())
    option("stem_mfma", 0)
    y0 = ops.conv_stem_s2_f16(xin, w, b, sc)
    assert (y.float() - y0.float()).abs().max().item() <= 2e-3 * max(1.0, y0.float().abs().max().item())     # one fp16 ulp apart at most
    assert (y != y0).float().mean().item() < 0.02
    option("stem_mfma", 1)
    from dbmm_amd import _lib
    n = y.numel()
    buf = torch.full((n + 4096,), 7.0, device=DEV, dtype=torch.float16)
    rc = _lib.lib().dbmm_conv_stem_s2_bn_f16(xin.data_ptr(), int(half_in), w.data_ptr(), sc.data_ptr(), b.data_ptr(), buf.data_ptr(), B, R, R, Cout, _lib.stream())
    torch.cuda.synchronize()
    assert rc == 0 and torch.equal(buf[:n].view_as(y), y) and (buf[n:] == 7.0).all()


def test_fp16_mode_rn50_vs_reference_fp16_path(golden):
    """the whole RN50 tower in fp16 mode (fp16 NHWC activations, the fp16 conv / GEMM kernels) against the reference's OWN
    fp16 path run on CPU: per-stage samples and the embedding within 3 x the reference's fp16-vs-fp32 distance"""
    arch = "RN50"
    g, h = golden(gname(arch)), golden(gname16(arch))
    seed, B, res = int(h["seed"]), int(h["batch"]), int(h["res"])
    model = convert_weights(build_model(synth.clip_state_dict(seed, arch)).cuda())
    img = synth.images(seed + 100, B, res).cuda()
    out, stages = model.visual(img.half(), return_stages=True)
    assert out.dtype == torch.float16 and all(t.dtype == torch.float16 for t in stages.values())
    tol = 3.0 * float(h["f16_vs_f32"])
    from conftest import summary
    for k, t in stages.items():
        sums, sample = summary(t.float())
        ref = h[f"{k}_sample"]
        e = np.abs(sample - ref).max() / np.abs(ref).max()
        print(f"RN50 fp16 {k}: sample error {e:.2e} of the maximum")
        assert e < tol, (k, e)
        assert abs(sums[1] - h[f"{k}_sums"][1]) <= 2e-3 * h[f"{k}_sums"][1], k
    e16, e32 = relerr(out.float().cpu(), torch.from_numpy(h["embedding"])), relerr(out.float().cpu(), torch.from_numpy(g["embedding"]))
    print(f"RN50 fp16: embedding vs reference fp16 path {e16:.2e} (tol {tol:.2e}), vs fp32 golden {e32:.2e}")
    assert e16 < tol and e32 < tol
    assert F.cosine_similarity(out.float().cpu(), torch.from_numpy(h["embedding"]), dim=1).min() > 0.99999
    # encode_image with an fp32 batch takes the same path (the image is rounded to fp16 like the reference's cast)
    assert torch.equal(model.encode_image(img), out)
    ops.profile_begin()
    model.encode_image(img)
    tags = set(ops.profile_end())
    assert any(t.startswith("conv3x3_f16_kernel<") for t in tags) and "stem_s2_f16_mfma_kernel" in tags, sorted(tags)


@pytest.mark.parametrize("B", [1, 3])
def test_fp16_mode_rn50_small_and_odd_batches(B, golden):
    """B = 1 and an odd batch (every conv tile is a ragged tail; pooled 3x3 tiles hold a fraction of their windows): rows equal the
    fixture images encoded in the fixture's own batch (bit-identical arithmetic per image up to tile order)"""
    h = golden(gname16("RN50"))
    seed = int(h["seed"])
    model = convert_weights(build_model(synth.clip_state_dict(seed, "RN50")).cuda())
    img = synth.images(seed + 100, 2, 224)
    batch = torch.cat([img, synth.images(5, 1, 224)])[:B].cuda() if B > 1 else img[:1].cuda()
    out = model.encode_image(batch)
    ref = torch.from_numpy(h["embedding"])[:min(B, 2)]
    assert out.dtype == torch.float16 and tuple(out.shape) == (B, 1024) and torch.isfinite(out).all()
    assert relerr(out[:min(B, 2)].float().cpu(), ref) < 3.0 * float(h["f16_vs_f32"])


def test_fp16_mode_rn50_large_batch_rows_are_independent():
    """a batch the fixture does not cover (every conv tile geometry full): any row equals the same image in a small batch"""
    model = convert_weights(build_model(synth.clip_state_dict(2, "RN50")).cuda())
    img = synth.images(9, 80, 224).cuda()
    big = model.encode_image(img)
    small = model.encode_image(img[37:40].contiguous())
    assert torch.isfinite(big).all() and relerr(big[37:40].float().cpu(), small.float().cpu()) < 2e-3
    assert torch.equal(big, model.encode_image(img))


def test_model_half_is_the_same_mode_as_convert_weights(golden):
    """`model.half()` (every floating-point tensor fp16, LayerNorm / embeddings included) selects the same fp16 plan as
    convert_weights; LayerNorm parameters and embeddings are widened back to fp32 at plan time, so the result may differ
    from convert_weights' only by their fp16 rounding."""
    arch = "tiny-ViT"
    g = golden(gname(arch))
    seed, B, res = int(g["seed"]), int(g["batch"]), int(g["res"])
    sd = synth.clip_state_dict(seed, arch)
    img = synth.images(seed + 100, B, res).cuda()
    a = convert_weights(build_model(sd).cuda()).encode_image(img)
    m = build_model(sd).cuda().half()
    assert m.dtype == torch.float16
    b = m.encode_image(img)
    t = m.encode_text(torch.from_numpy(g["tokens"]).cuda())
    assert b.dtype == torch.float16 and relerr(b.float().cpu(), a.float().cpu()) < 5e-3
    assert relerr(t.float().cpu(), torch.from_numpy(golden(gname16(arch))["text_embedding"])) < 5e-3
    # fp16 input images are accepted as they are (the reference casts the image to model.dtype, clip/model.py:341)
    c = m.encode_image(img.half())
    assert relerr(c.float().cpu(), b.float().cpu()) < 5e-3


def test_fp16_mode_large_batch_rows_are_independent():
    """ViT-B/32 fp16 at a batch the goldens do not cover: any row of a big batch equals the same image encoded in a
    small batch (tile schedules differ, so not bit-exact), and repeated runs are bit-identical."""
    model = convert_weights(build_model(synth.clip_state_dict(2, "ViT-B/32")).cuda())
    img = synth.images(9, 96, 224).cuda()
    big = model.encode_image(img)
    small = model.encode_image(img[40:43].contiguous())
    assert torch.isfinite(big).all() and relerr(big[40:43].float().cpu(), small.float().cpu()) < 2e-3
    assert torch.equal(big, model.encode_image(img))


def test_fp16_conv_kernels_on_operands_over_2gib():
    """fp16 NHWC maps past 2 GiB (RN50 layer 1 from B = 1338): every tile of the conv kernels rebases its buffer descriptors on a
    64-bit base; first / straddling / last images against fp64 on the same fp16 values."""
    g = torch.Generator(device=DEV); g.manual_seed(3)
    H = 56
    # --- 1x1 streaming kernel: 256 -> 64 channels (K-loop of 8 chunks) and 64 -> 256 + residual on 1400 images
    B, Cin, Cout = 1400, 256, 64
    x = torch.relu(torch.randn((B, H, H, Cin), device=DEV, generator=g, dtype=torch.float16))
    assert x.numel() * 2 > 2 ** 31
    w = (torch.randn((Cout, Cin), device=DEV, generator=g) * Cin ** -0.5).half()
    sc, b = _bn(g, Cout)
    y = ops.conv1x1_f16(x, w, sc, b)
    w3 = (torch.randn((Cin, Cout), device=DEV, generator=g) * Cout ** -0.5).half()
    s3, b3 = _bn(g, Cin)
    x2 = ops.conv1x1_f16(y, w3, s3, b3, residual=x)
    assert x2.numel() * 2 > 2 ** 31
    straddle = 2 ** 31 // (H * H * Cin * 2)
    for i in (0, straddle, straddle + 1, B - 1):
        yr = torch.relu(x[i].view(-1, Cin).double() @ w.double().t() * sc.double() + b.double())
        assert relerr(y[i].view(-1, Cout).double().cpu(), yr.cpu()) < 1.5e-3, i
        xr = torch.relu(y[i].view(-1, Cout).double() @ w3.double().t() * s3.double() + b3.double() + x[i].view(-1, Cin).double())
        assert relerr(x2[i].view(-1, Cin).double().cpu(), xr.cpu()) < 1.5e-3, i
    del x, x2, y
    torch.cuda.empty_cache()
    # --- 3x3 kernel, 64 -> 64 channels on 56 x 56 maps: 5600 images = 2.25 GB input; plain and pooled
    B, C = 5600, 64
    x = torch.relu(torch.randn((B, H, H, C), device=DEV, generator=g, dtype=torch.float16))
    assert x.numel() * 2 > 2 ** 31
    w = (torch.randn((C, C, 3, 3), device=DEV, generator=g) * (9 * C) ** -0.5).half()
    wp, _ = ops.pack_conv_weight(w.float(), chunk_major=32)
    sc, b = _bn(g, C)
    straddle = 2 ** 31 // (H * H * C * 2)
    for pool in (1, 2):
        y = ops.conv3x3_f16(x, wp.half().contiguous(), sc, b, pool=pool)
        for i in (0, straddle, straddle + 1, B - 1):
            ref = torch.relu(F.conv2d(x[i].permute(2, 0, 1)[None].double(), w.double(), None, padding=1) * sc.double().view(1, -1, 1, 1)
                             + b.double().view(1, -1, 1, 1))
            if pool == 2:
                ref = F.avg_pool2d(ref, 2)
            assert relerr(y[i].double().cpu(), ref[0].permute(1, 2, 0).cpu()) < 1.5e-3, (pool, i)
        del y


@pytest.mark.parametrize("M,K,K2,N", [(28 * 28 * 24, 128, 256, 512), (14 * 14 * 90 + 12, 256, 512, 1024), (7 * 7 * 340, 512, 1024, 2048),
                                      # the streaming kernel's dual-source mode: layer 1 (64 + 64 channels), a small ragged problem, Cout % 128 != 0
                                      (56 * 56 * 6, 64, 64, 256), (1000 + 77, 128, 256, 512), (56 * 56 * 2 + 5, 64, 96, 200)])
def test_conv1x1_dual_f16(M, K, K2, N):
    """fp16 mode, first block of a stage: relu(bn3(conv3(y2)) + bn_d(conv_d(xp))) as ONE dual-source GEMM on the eight-phase kernel
    (clip/model.py:36-38, 42-55) against fp64 and against the two launches it replaces (one fp16 rounding less: the identity is never
    rounded to fp16); shapes the library does not take answer None"""
    g = torch.Generator(device=DEV); g.manual_seed(M + K)
    y2 = torch.relu(torch.randn((M, K), device=DEV, generator=g)).half(); xp = torch.relu(torch.randn((M, K2), device=DEV, generator=g) * 2.0).half()
    w3 = (torch.randn((N, K), device=DEV, generator=g) * K ** -0.5).half(); wd = (torch.randn((N, K2), device=DEV, generator=g) * K2 ** -0.5).half()
    s3 = 0.5 + torch.rand((N,), device=DEV, generator=g); sd = 0.5 + torch.rand((N,), device=DEV, generator=g)
    b3 = torch.randn((N,), device=DEV, generator=g) * 0.1; bd = torch.randn((N,), device=DEV, generator=g) * 0.1
    out = ops.conv1x1_dual_f16(y2, w3, s3, xp, wd, (sd / s3).contiguous(), (b3 + bd).contiguous())
    assert out is not None and out.dtype == torch.float16 and tuple(out.shape) == (M, N)
    rows = torch.cat([torch.arange(0, 300, device=DEV), torch.randint(0, M, (1500,), device=DEV, generator=g), torch.arange(M - 300, M, device=DEV)])
    ref = torch.relu(y2[rows].double() @ w3.double().t() * s3.double() + xp[rows].double() @ wd.double().t() * sd.double() + (b3 + bd).double())
    assert torch.allclose(out[rows].double(), ref, rtol=2e-3, atol=2e-3)
    ident = ops.conv1x1_f16(xp, wd, sd, bd, act=ops.ACT_NONE)
    two = ops.conv1x1_f16(y2, w3, s3, b3, residual=ident)
    assert (out.float() - two.float()).abs().max().item() <= 8e-3 * max(1.0, two.float().abs().max().item())
    # guard zone: nothing written past the (possibly ragged) output
    buf = torch.full((M * N + 4096,), 7.0, device=DEV, dtype=torch.float16)
    from dbmm_amd import _lib
    ratio, bsum = (sd / s3).contiguous(), (b3 + bd).contiguous()           # (kept alive: the call takes raw pointers)
    rc = _lib.lib().dbmm_conv1x1_dual_bn_act_f16(y2.data_ptr(), w3.data_ptr(), s3.data_ptr(), xp.data_ptr(), wd.data_ptr(), ratio.data_ptr(),
                                                 bsum.data_ptr(), buf.data_ptr(), M, K, K2, N, ops.ACT_RELU, _lib.stream())
    torch.cuda.synchronize()
    assert rc == 0 and torch.equal(buf[:M * N].view(M, N), out) and (buf[M * N:] == 7.0).all()


@pytest.mark.parametrize("M,K,N,P", [(56 * 56 * 3, 64, 256, 64), (28 * 28 * 9 + 5, 128, 512, 128), (1000, 64, 192, 128), (130, 128, 64, 64), (7, 64, 64, 64)])
def test_bottleneck_chain_f16(M, K, N, P):
    """fp16 mode: conv3 + BatchNorm + residual + ReLU of a block and conv1 + BatchNorm + ReLU of the next block as one launch
    (clip/model.py:42-55 twice): equal BIT FOR BIT to the two launches it replaces (x' is rounded to fp16 before it feeds conv1'), against
    fp64 on sample rows, ragged tiles, guard zones behind both outputs; other shapes answer None"""
    g = torch.Generator(device=DEV); g.manual_seed(M + K + N + P)
    y2 = torch.relu(torch.randn((M, K), device=DEV, generator=g)).half(); res = torch.relu(torch.randn((M, N), device=DEV, generator=g) * 2.0).half()
    w3 = (torch.randn((N, K), device=DEV, generator=g) * K ** -0.5).half(); w1 = (torch.randn((P, N), device=DEV, generator=g) * N ** -0.5).half()
    s3, b3 = _bn(g, N); s1, b1 = _bn(g, P)
    r = ops.chain_f16(y2, (w3, s3, b3), res, (w1, s1, b1))
    assert r is not None
    x, y1 = r
    x_two = ops.conv1x1_f16(y2, w3, s3, b3, residual=res)
    y1_two = ops.conv1x1_f16(x_two, w1, s1, b1)
    assert torch.equal(x, x_two), (x != x_two).sum().item()
    assert torch.equal(y1, y1_two), (y1 != y1_two).sum().item()
    xr = torch.relu(y2.double() @ w3.double().t() * s3.double() + b3.double() + res.double())
    assert torch.allclose(x.double(), xr, rtol=2e-3, atol=2e-3)
    y1r = torch.relu(x.double() @ w1.double().t() * s1.double() + b1.double())           # from the fp16 x', like the reference's next conv
    assert torch.allclose(y1.double(), y1r, rtol=2e-3, atol=2e-3)
    from dbmm_amd import _lib
    bx = torch.full((M * N + 4096,), 7.0, device=DEV, dtype=torch.float16); by = torch.full((M * P + 4096,), 7.0, device=DEV, dtype=torch.float16)
    rc = _lib.lib().dbmm_bottleneck_chain_f16(y2.data_ptr(), w3.data_ptr(), s3.data_ptr(), b3.data_ptr(), res.data_ptr(), bx.data_ptr(), w1.data_ptr(),
                                              s1.data_ptr(), b1.data_ptr(), by.data_ptr(), M, K, N, P, _lib.stream())
    torch.cuda.synchronize()
    assert rc == 0 and torch.equal(bx[:M * N].view(M, N), x) and torch.equal(by[:M * P].view(M, P), y1)
    assert (bx[M * N:] == 7.0).all() and (by[M * P:] == 7.0).all()
    assert ops.chain_f16(torch.zeros((M, 256), device=DEV, dtype=torch.float16), (torch.zeros((N, 256), device=DEV, dtype=torch.float16), s3, b3), res, (w1, s1, b1)) is None


@pytest.mark.parametrize("M,N", [(56 * 56 * 3, 256), (1000 + 37, 192), (7, 128)])
def test_bottleneck_chain_dual_f16(M, N):
    """fp16 mode, layer 1's first block: conv3 + downsample branch + ReLU and conv1 of the next block as one launch (clip/model.py:36-38,
    42-55, then 42 of the next block): equal BIT FOR BIT to the dual launch + conv1 launch it replaces, against fp64, ragged tiles, guard
    zones behind both outputs; other depths answer None"""
    K = K2 = P = 64
    g = torch.Generator(device=DEV); g.manual_seed(M + N)
    y2 = torch.relu(torch.randn((M, K), device=DEV, generator=g)).half(); xp = torch.relu(torch.randn((M, K2), device=DEV, generator=g) * 2.0).half()
    w3 = (torch.randn((N, K), device=DEV, generator=g) * K ** -0.5).half(); wd = (torch.randn((N, K2), device=DEV, generator=g) * K2 ** -0.5).half()
    w1 = (torch.randn((P, N), device=DEV, generator=g) * N ** -0.5).half()
    s3 = 0.5 + torch.rand((N,), device=DEV, generator=g); sd = 0.5 + torch.rand((N,), device=DEV, generator=g)
    b3 = torch.randn((N,), device=DEV, generator=g) * 0.1; bd = torch.randn((N,), device=DEV, generator=g) * 0.1
    s1, b1 = _bn(g, P)
    ratio, bsum = (sd / s3).contiguous(), (b3 + bd).contiguous()
    r = ops.chain_dual_f16(y2, w3, s3, xp, wd, ratio, bsum, (w1, s1, b1))
    assert r is not None
    x, y1 = r
    x_two = ops.conv1x1_dual_f16(y2, w3, s3, xp, wd, ratio, bsum)
    assert x_two is not None
    y1_two = ops.conv1x1_f16(x_two, w1, s1, b1)
    assert torch.equal(x, x_two), (x != x_two).sum().item()
    assert torch.equal(y1, y1_two), (y1 != y1_two).sum().item()
    xr = torch.relu(y2.double() @ w3.double().t() * s3.double() + xp.double() @ wd.double().t() * sd.double() + bsum.double())
    assert torch.allclose(x.double(), xr, rtol=2e-3, atol=2e-3)
    y1r = torch.relu(x.double() @ w1.double().t() * s1.double() + b1.double())
    assert torch.allclose(y1.double(), y1r, rtol=2e-3, atol=2e-3)
    from dbmm_amd import _lib
    bx = torch.full((M * N + 4096,), 7.0, device=DEV, dtype=torch.float16); by = torch.full((M * P + 4096,), 7.0, device=DEV, dtype=torch.float16)
    rc = _lib.lib().dbmm_bottleneck_chain_dual_f16(y2.data_ptr(), w3.data_ptr(), s3.data_ptr(), xp.data_ptr(), wd.data_ptr(), ratio.data_ptr(),
                                                   bsum.data_ptr(), bx.data_ptr(), w1.data_ptr(), s1.data_ptr(), b1.data_ptr(), by.data_ptr(),
                                                   M, K, K2, N, P, _lib.stream())
    torch.cuda.synchronize()
    assert rc == 0 and torch.equal(bx[:M * N].view(M, N), x) and torch.equal(by[:M * P].view(M, P), y1)
    assert (bx[M * N:] == 7.0).all() and (by[M * P:] == 7.0).all()
    w3b = (torch.randn((N, 128), device=DEV, generator=g) * 0.1).half()
    assert ops.chain_dual_f16(torch.zeros((M, 128), device=DEV, dtype=torch.float16), w3b, s3, xp, wd, ratio, bsum, (w1, s1, b1)) is None


@pytest.mark.parametrize("B,H,W,K,N,P", [(3, 56, 56, 64, 256, 128), (5, 6, 10, 64, 192, 64), (1, 2, 2, 64, 64, 64), (9, 28, 28, 128, 512, 128),
                                         (7, 14, 6, 128, 128, 64)])
def test_bottleneck_chain_f16_at_a_stage_seam(B, H, W, K, N, P):
    """fp16 mode, the last block of a stage: the chain launch also writes AvgPool2d(2) of x' for the next stage's downsample branch
    (clip/model.py:36-38, 52) -- window-major tile rows; x', y1' equal the plain chain launch and the pooled copy equals avgpool2_f16 of x',
    bit for bit; pooled-only (x' not written) gives the same pooled copy and y1'; ragged tiles, windows wrapping over pooled rows and images,
    guard zones"""
    g = torch.Generator(device=DEV); g.manual_seed(B + H + K + N + P)
    y2 = torch.relu(torch.randn((B, H, W, K), device=DEV, generator=g)).half(); res = torch.relu(torch.randn((B, H, W, N), device=DEV, generator=g) * 2.0).half()
    w3 = (torch.randn((N, K), device=DEV, generator=g) * K ** -0.5).half(); w1 = (torch.randn((P, N), device=DEV, generator=g) * N ** -0.5).half()
    s3, b3 = _bn(g, N); s1, b1 = _bn(g, P)
    x0, y0 = ops.chain_f16(y2, (w3, s3, b3), res, (w1, s1, b1))
    r = ops.chain_f16(y2, (w3, s3, b3), res, (w1, s1, b1), pooled=True)
    assert r is not None
    x, xp, y1 = r
    assert torch.equal(x, x0) and torch.equal(y1, y0)
    assert tuple(xp.shape) == (B, H // 2, W // 2, N) and torch.equal(xp, ops.avgpool2_f16(x))
    ref_p = F.avg_pool2d(x.float().permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    assert torch.allclose(xp.float(), ref_p, rtol=1e-3, atol=1e-3)
    x2, xp2, y12 = ops.chain_f16(y2, (w3, s3, b3), res, (w1, s1, b1), pooled=True, keep_full=False)
    assert x2 is None and torch.equal(xp2, xp) and torch.equal(y12, y1)
    from dbmm_amd import _lib
    M = B * H * W
    bx = torch.full((M * N + 4096,), 7.0, device=DEV, dtype=torch.float16); by = torch.full((M * P + 4096,), 7.0, device=DEV, dtype=torch.float16)
    bp = torch.full((M // 4 * N + 4096,), 7.0, device=DEV, dtype=torch.float16)
    rc = _lib.lib().dbmm_bottleneck_chain_pool_f16(y2.data_ptr(), w3.data_ptr(), s3.data_ptr(), b3.data_ptr(), res.data_ptr(), bx.data_ptr(), bp.data_ptr(),
                                                   w1.data_ptr(), s1.data_ptr(), b1.data_ptr(), by.data_ptr(), B, H, W, K, N, P, _lib.stream())
    torch.cuda.synchronize()
    assert rc == 0 and torch.equal(bx[:M * N].view_as(x), x) and torch.equal(by[:M * P].view_as(y1), y1) and torch.equal(bp[:M // 4 * N].view_as(xp), xp)
    assert (bx[M * N:] == 7.0).all() and (by[M * P:] == 7.0).all() and (bp[M // 4 * N:] == 7.0).all()
    # odd map sides: no kernel
    assert ops.chain_f16(y2[:, :H - 1].contiguous(), (w3, s3, b3), res[:, :H - 1].contiguous(), (w1, s1, b1), pooled=True) is None


@pytest.mark.parametrize("B,H,W,N", [(1024, 14, 14, 1024), (673, 14, 14, 1088), (170, 28, 28, 1024), (46, 54, 58, 1024)])
def test_conv1x1_res_stream_f16_kernel(B, H, W, N, option):
    """fp16 mode: conv3 + BatchNorm + residual + ReLU with K = 256 into >= 1024 channels (clip/model.py:50-54, layer 3) on the row-owning
    streaming kernel, through dbmm_conv1x1_bn_act_f16: BIT-EQUAL to conv1x1_f16_kernel (option conv1x1_res_stream = 0), against fp64; the
    stage's last block (dbmm_conv1x1_res_pool_f16): same y, pooled copy == avgpool2_f16 of y bit for bit; a ragged last tile (B = 673), a
    slab count that does not divide over the workgroups (N = 1088: 17 slabs), windows wrapping over pooled rows and images, guard zones"""
    K = 256
    g = torch.Generator(device=DEV); g.manual_seed(B + N + W)
    x = torch.relu(torch.randn((B, H, W, K), device=DEV, generator=g)).half()
    res = torch.relu(torch.randn((B, H, W, N), device=DEV, generator=g, dtype=torch.float16) * 2.0)
    w = (torch.randn((N, K), device=DEV, generator=g) * K ** -0.5).half()
    sc, b = _bn(g, N)
    ops.profile_begin()
    y = ops.conv1x1_f16(x, w, sc, b, residual=res)
    assert list(ops.profile_end()) == ["conv1x1_res_stream_f16_kernel<256, 0>"]
    r = ops.conv1x1_res_pool_f16(x, (w, sc, b), res)
    assert r is not None
    y2, yp = r
    option("conv1x1_res_stream", 0)
    ops.profile_begin()
    y0 = ops.conv1x1_f16(x, w, sc, b, residual=res)
    assert not any(t.startswith("conv1x1_res_stream") for t in ops.profile_end())
    assert ops.conv1x1_res_pool_f16(x, (w, sc, b), res) is None
    option("conv1x1_res_stream", 1)
    assert torch.equal(y, y0), (y != y0).sum().item()
    assert torch.equal(y2, y) and torch.equal(yp, ops.avgpool2_f16(y))
    M = B * H * W
    rows = torch.cat([torch.arange(0, 300, device=DEV), torch.randint(0, M, (1500,), device=DEV, generator=g), torch.arange(M - 300, M, device=DEV)])
    ref = torch.relu(x.view(M, K)[rows].double() @ w.double().t() * sc.double() + b.double() + res.view(M, N)[rows].double())
    assert torch.allclose(y.view(M, N)[rows].double(), ref, rtol=2e-3, atol=2e-3)
    from dbmm_amd import _lib
    bf = torch.full((M * N + 4096,), 7.0, device=DEV, dtype=torch.float16); bp = torch.full((M // 4 * N + 4096,), 7.0, device=DEV, dtype=torch.float16)
    rc = _lib.lib().dbmm_conv1x1_res_pool_f16(x.data_ptr(), w.data_ptr(), sc.data_ptr(), b.data_ptr(), res.data_ptr(), bf.data_ptr(), bp.data_ptr(),
                                              B, H, W, K, N, _lib.stream())
    torch.cuda.synchronize()
    assert rc == 0 and torch.equal(bf[:M * N].view_as(y), y) and torch.equal(bp[:M // 4 * N].view_as(yp), yp)
    assert (bf[M * N:] == 7.0).all() and (bp[M // 4 * N:] == 7.0).all()
    # smaller problems keep the other kernels
    ops.profile_begin()
    ops.conv1x1_f16(x[:40].contiguous(), w, sc, b, residual=res[:40].contiguous())
    assert not any(t.startswith("conv1x1_res_stream") for t in ops.profile_end())


@pytest.mark.parametrize("B,H,W,N", [(128, 28, 28, 512), (37, 30, 26, 576), (3, 2, 2, 64)])
def test_conv1x1_res_pool_f16_layer2_depth(B, H, W, N):
    """the same pooled launch at layer 2's depth (K = 128, the layer 2 -> 3 seam): y bit-equal to conv1x1_f16, pooled copy == avgpool2_f16 of y"""
    K = 128
    g = torch.Generator(device=DEV); g.manual_seed(B + N + W)
    x = torch.relu(torch.randn((B, H, W, K), device=DEV, generator=g)).half()
    res = torch.relu(torch.randn((B, H, W, N), device=DEV, generator=g, dtype=torch.float16) * 2.0)
    w = (torch.randn((N, K), device=DEV, generator=g) * K ** -0.5).half()
    sc, b = _bn(g, N)
    ops.profile_begin()
    r = ops.conv1x1_res_pool_f16(x, (w, sc, b), res)
    assert r is not None and list(ops.profile_end()) == ["conv1x1_res_stream_f16_kernel<128, 1>"]
    y, yp = r
    y0 = ops.conv1x1_f16(x, w, sc, b, residual=res)
    assert torch.equal(y, y0) and torch.equal(yp, ops.avgpool2_f16(y0))
    M = B * H * W
    ref = torch.relu(x.view(M, K)[:2000].double() @ w.double().t() * sc.double() + b.double() + res.view(M, N)[:2000].double())
    assert torch.allclose(y.view(M, N)[:2000].double(), ref, rtol=2e-3, atol=2e-3)
    assert ops.conv1x1_res_pool_f16(torch.zeros((B, H, W, 64), device=DEV, dtype=torch.float16), (w[:, :64].contiguous(), sc, b), res) is None


def test_fp16_mode_encode_image_takes_fp32_images_without_a_cast_pass():
    """CLIP.encode_image casts the batch to the model dtype (clip/model.py:340-341); in fp16 mode on the fp16 kernels the stem conv rounds an
    fp32 image itself (so does the ViT towers' patch gather): same embedding bit for bit as for the pre-cast batch, no cast kernel in between"""
    for arch in ("RN50", "ViT-B/32"):
        model = convert_weights(build_model(synth.clip_state_dict(3, arch)).cuda())
        img = synth.images(11, 3, 224).cuda()
        assert model.visual.rounds_fp32_images_itself()
        a = model.encode_image(img)
        b = model.encode_image(img.half())
        assert a.dtype == torch.float16 and torch.equal(a, b), arch
