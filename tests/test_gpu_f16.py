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
                                           (577 * 64, 3072, 1024, False, 2), (16384, 768, 3072, True, 0)])
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
    for _ in range(4):
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
                                              (2, 1, 1, False), (1, 128, 2, True), (2, 129, 1, True)])
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


def test_fp16_mode_rn_tower_rounds_its_fp32_accurate_output():
    sd = synth.clip_state_dict(3, "tiny-RN")
    model = build_model(sd).cuda()
    img = synth.images(103, 2, 64).cuda()
    ref = model.encode_image(img)
    convert_weights(model)
    out = model.encode_image(img)
    assert out.dtype == torch.float16 and relerr(out.float().cpu(), ref.cpu()) < 1e-3


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
