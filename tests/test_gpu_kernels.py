"""Kernel-level parity: every C-ABI entry point against a plain torch-CPU fp32 computation of
the same op (the floating-point oracle for a single kernel), through the ctypes binding."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import clip_oracle as CO
from conftest import ROOT, relerr
from dbmm_amd import ops, synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(seed, name, shape, std=1.0):
    return synth.normal(seed, name, shape, std)


GEMM_SHAPES = [
    # M, N, K  -> exercises the 128x128, 128x64, 128x32 and 64x64 tiles, ragged edges, K tails
    (512, 256, 128), (1000, 128, 64), (300, 64, 96), (257, 32, 288), (64, 1024, 2048),
    (200, 136, 588), (50, 20, 36), (1024, 128, 1024), (4, 1024, 128),
]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("ta,tw", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm_modes(M, N, K, ta, tw):
    if ta and M % 4:
        pytest.skip("K-major A needs M % 4 == 0")
    if tw and N % 4:
        pytest.skip("K-major W needs N % 4 == 0")
    a = rnd(1, "a", (M, K)); w = rnd(2, "w", (N, K), K ** -0.5); b = rnd(3, "b", (N,))
    ref = a @ w.t() + b
    ad = (a.t().contiguous() if ta else a).to(DEV)
    wd = (w.t().contiguous() if tw else w).to(DEV)
    out = ops.gemm(ad, wd, b.to(DEV), trans_a=ta, trans_w=tw)
    assert out.shape == (M, N)
    assert relerr(out.cpu(), ref) < 2e-5


@pytest.mark.parametrize("act", [ops.ACT_NONE, ops.ACT_RELU, ops.ACT_QUICKGELU])
def test_gemm_epilogue(act):
    M, N, K = 333, 192, 256
    a = rnd(1, "a", (M, K)); w = rnd(2, "w", (N, K), K ** -0.5); b = rnd(3, "b", (N,)); r = rnd(4, "r", (M, N))
    v = (a @ w.t() + b) * 0.125 + r
    ref = {0: v, 1: F.relu(v), 2: v * torch.sigmoid(1.702 * v)}[act]
    out = ops.gemm(a.to(DEV), w.to(DEV), b.to(DEV), residual=r.to(DEV), act=act, alpha=0.125)
    assert relerr(out.cpu(), ref) < 2e-5


def test_gemm_strided_rows():
    """row stride > K: the q projection of attention-pool reads token 0 of every image."""
    B, L, C = 37, 5, 64
    t = rnd(1, "t", (B, L, C)); w = rnd(2, "w", (C, C), C ** -0.5)
    out = ops.gemm(t.to(DEV), w.to(DEV), M=B, K=C, lda=L * C)
    assert relerr(out.cpu(), t[:, 0] @ w.t()) < 2e-5


def test_gemm_rejects_bad_shapes():
    from dbmm_amd._lib import DbmmError
    a = torch.zeros(8, 6, device=DEV); w = torch.zeros(4, 6, device=DEV)
    with pytest.raises(DbmmError):
        ops.gemm(a, w)                       # K % 4 != 0
    with pytest.raises(DbmmError):
        ops.gemm(torch.zeros(8, 8), torch.zeros(4, 8))   # CPU tensors: no fallback


CONV_CASES = [
    # B, H, W, Cin, Cout, k, stride, pad
    (2, 16, 16, 32, 32, 3, 1, 1), (2, 16, 16, 32, 64, 3, 1, 1), (3, 14, 14, 64, 128, 3, 1, 1),
    (2, 7, 7, 16, 48, 3, 1, 1), (1, 9, 11, 8, 40, 3, 1, 1), (2, 12, 12, 64, 256, 1, 1, 0),
    (2, 13, 13, 32, 64, 3, 2, 1), (5, 7, 7, 512, 512, 3, 1, 1), (2, 56, 56, 64, 64, 3, 1, 1),
]


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,stride,pad", CONV_CASES)
def test_conv_bn_act(B, H, W, Cin, Cout, k, stride, pad):
    x = rnd(1, "x", (B, Cin, H, W)); w = rnd(2, "w", (Cout, Cin, k, k), (Cin * k * k) ** -0.5)
    bias = rnd(3, "b", (Cout,), 0.1)
    ref = F.conv2d(x, w, bias, stride=stride, padding=pad)
    res = rnd(4, "r", tuple(ref.shape))
    ref = F.relu(ref + res).permute(0, 2, 3, 1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wd = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    out = ops.conv_bn_act(xd, wd, bias.to(DEV), res.permute(0, 2, 3, 1).contiguous().to(DEV), k, k, stride, pad,
                          ops.ACT_RELU)
    assert relerr(out.cpu(), ref) < 2e-5


@pytest.mark.parametrize("B,H,Cin,Cout", [(2, 16, 32, 32), (3, 14, 64, 128), (2, 9, 16, 48), (5, 7, 512, 256)])
def test_conv_chunk_major_weights(B, H, Cin, Cout):
    """packed K order (cin/16, kh, kw, 16): same convolution, taps adjacent along K."""
    x = rnd(1, "x", (B, Cin, H, H)); w = rnd(2, "w", (Cout, Cin, 3, 3), (Cin * 9) ** -0.5); b = rnd(3, "b", (Cout,), 0.1)
    ref = F.relu(F.conv2d(x, w, b, padding=1)).permute(0, 2, 3, 1)
    wp, wl = ops.pack_conv_weight(w.to(DEV), chunk_major=True)
    assert wl == ops.WL_CHUNK_MAJOR
    out = ops.conv_bn_act(x.permute(0, 2, 3, 1).contiguous().to(DEV), wp, b.to(DEV), None, 3, 3, 1, 1, ops.ACT_RELU, wl)
    assert relerr(out.cpu(), ref) < 2e-5
    wp0, wl0 = ops.pack_conv_weight(rnd(4, "w8", (Cout, 8, 3, 3)).to(DEV), chunk_major=True)   # Cin % 16 != 0 -> tap-major
    assert wl0 == ops.WL_TAP_MAJOR and wp0.shape == (Cout, 72)


@pytest.mark.parametrize("B,R,Cout", [(2, 64, 32), (3, 33, 16), (1, 224, 32), (3, 33, 32), (2, 50, 40), (1, 36, 48), (5, 30, 64)])
def test_conv_stem_s2(B, R, Cout):
    x = rnd(1, "x", (B, 3, R, R)); w = rnd(2, "w", (Cout, 3, 3, 3), 27 ** -0.5); b = rnd(3, "b", (Cout,), 0.1)
    ref = F.relu(F.conv2d(x, w, b, stride=2, padding=1)).permute(0, 2, 3, 1)
    am = torch.zeros(1, device=DEV)
    out = ops.conv_stem_s2(x.to(DEV), w.permute(2, 3, 1, 0).contiguous().to(DEV), b.to(DEV), y_absmax=am)
    assert relerr(out.cpu(), ref) < 1e-5
    assert am.item() == out.max().item()


@pytest.mark.parametrize("k", [2, 7])
def test_avgpool(k):
    x = rnd(1, "x", (3, 32, 14, 14))
    ref = F.avg_pool2d(x, k).permute(0, 2, 3, 1)
    out = ops.avgpool2d(x.permute(0, 2, 3, 1).contiguous().to(DEV), k)
    assert relerr(out.cpu(), ref) < 1e-6


def test_attnpool():
    B, C, S, heads, Dout = 5, 256, 3, 4, 96
    sd = {}
    p = "a."
    sd[p + "positional_embedding"] = rnd(1, "pos", (S * S + 1, C), C ** -0.5)
    for nm, o in (("q_proj", C), ("k_proj", C), ("v_proj", C), ("c_proj", Dout)):
        sd[p + nm + ".weight"] = rnd(2, nm, (o, C), C ** -0.5)
        sd[p + nm + ".bias"] = rnd(3, nm + "b", (o,), 0.1)
    x = rnd(4, "x", (B, C, S, S))
    ref = CO.attention_pool(x, sd, p, heads)
    d = {k: v.to(DEV) for k, v in sd.items()}
    out = ops.attnpool(x.permute(0, 2, 3, 1).contiguous().to(DEV), d[p + "positional_embedding"],
                       d[p + "q_proj.weight"], d[p + "q_proj.bias"],
                       torch.cat([d[p + "k_proj.weight"], d[p + "v_proj.weight"]]).contiguous(),
                       torch.cat([d[p + "k_proj.bias"], d[p + "v_proj.bias"]]).contiguous(),
                       d[p + "c_proj.weight"], d[p + "c_proj.bias"], heads)
    assert relerr(out.cpu(), ref) < 1e-5
    # the fp16 mode hands its fp16 feature map over as it is (dbmm_attnpool_x): same as casting it to fp32 first, bit for bit
    xh = x.permute(0, 2, 3, 1).contiguous().to(DEV).half()
    args = (d[p + "positional_embedding"], d[p + "q_proj.weight"], d[p + "q_proj.bias"],
            torch.cat([d[p + "k_proj.weight"], d[p + "v_proj.weight"]]).contiguous(),
            torch.cat([d[p + "k_proj.bias"], d[p + "v_proj.bias"]]).contiguous(), d[p + "c_proj.weight"], d[p + "c_proj.bias"], heads)
    assert torch.equal(ops.attnpool(xh, *args), ops.attnpool(xh.float(), *args))


@pytest.mark.parametrize("rows,E", [(7, 64), (100, 768), (33, 1024), (5, 100)])
def test_layernorm(rows, E):
    x = rnd(1, "x", (rows, E), 2.0) + 0.5
    g = synth.uniform(2, "g", (E,), 0.5, 1.5); b = rnd(3, "b", (E,), 0.1)
    ref = F.layer_norm(x, (E,), g, b, 1e-5)
    am = torch.zeros(1, device=DEV)
    out = ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), y_absmax=am)
    assert relerr(out.cpu(), ref) < 1e-5
    assert am.item() == out.abs().max().item()


@pytest.mark.parametrize("E", [64, 320, 768, 1024, 1280, 2052, 4096])
def test_layernorm_widths(E):
    """register-resident variants (E <= 2048) and the generic fallback"""
    x = rnd(1, "x", (37, E), 3.0) - 1.0
    g = synth.uniform(2, "g", (E,), 0.5, 1.5); b = rnd(3, "b", (E,), 0.1)
    out = ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV))
    assert relerr(out.cpu(), F.layer_norm(x, (E,), g, b, 1e-5)) < 1e-5


def test_layernorm_strided_rows():
    B, L, E = 6, 5, 128
    x = rnd(1, "x", (B, L, E))
    g = synth.uniform(2, "g", (E,), 0.5, 1.5); b = rnd(3, "b", (E,), 0.1)
    out = ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), rows=B, ldx=L * E)
    assert relerr(out.cpu(), F.layer_norm(x[:, 0], (E,), g, b)) < 1e-5


@pytest.mark.parametrize("B,L,heads,causal", [(3, 50, 2, False), (2, 77, 8, True), (2, 130, 1, False),
                                              (1, 130, 2, True), (2, 5, 1, False), (1, 577, 1, False),
                                              (2, 197, 3, False), (2, 300, 2, True), (1, 256, 1, False),
                                              (1, 257, 2, True)])
def test_mha_core(B, L, heads, causal):
    """matrix-core attention kernel (default) against torch softmax attention"""
    E = heads * 64
    qkv = rnd(1, "qkv", (B, L, 3 * E))
    q, k, v = qkv.split(E, dim=-1)
    sh = lambda t: t.reshape(B, L, heads, 64).transpose(1, 2)
    s = (sh(q) * 0.125) @ sh(k).transpose(-1, -2)
    if causal:
        s = s + torch.full((L, L), float("-inf")).triu_(1)
    ref = (torch.softmax(s, -1) @ sh(v)).transpose(1, 2).reshape(B * L, E)
    out = ops.mha_core(qkv.to(DEV).view(B * L, 3 * E), B, L, E, heads, causal)
    assert relerr(out.cpu(), ref) < 1e-5


def test_mha_core_spiked_scores():
    """forces large running-max jumps between key groups (online-softmax rescale path)."""
    B, L, E = 1, 100, 64
    qkv = rnd(1, "qkv", (B, L, 3 * E))
    qkv[0, :, E:2 * E][37] *= 40.0
    qkv[0, :, E:2 * E][90] *= -60.0
    q, k, v = qkv.split(E, dim=-1)
    ref = torch.softmax((q * 0.125) @ k.transpose(-1, -2), -1) @ v
    out = ops.mha_core(qkv.to(DEV).view(L, 3 * E), B, L, E, 1, False)
    assert relerr(out.cpu(), ref.reshape(L, E)) < 1e-5


def test_mha_core_valu_kernel_agrees():
    """DBMM_MHA_VALU=1 (lane-per-query VALU kernel, kept for ablation) gives the same result"""
    import os, subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, torch
        sys.path.insert(0, %r)
        import dbmm_amd
        from dbmm_amd import ops, synth
        B, L, heads = 2, 77, 3
        E = heads * 64
        qkv = synth.normal(1, "qkv", (B, L, 3 * E))
        q, k, v = qkv.split(E, dim=-1)
        sh = lambda t: t.reshape(B, L, heads, 64).transpose(1, 2)
        s = (sh(q) * 0.125) @ sh(k).transpose(-1, -2) + torch.full((L, L), float("-inf")).triu_(1)
        ref = (torch.softmax(s, -1) @ sh(v)).transpose(1, 2).reshape(B * L, E)
        out = ops.mha_core(qkv.cuda().view(B * L, 3 * E), B, L, E, heads, True).cpu()
        assert ((out - ref).abs().max() / ref.abs().max()).item() < 1e-5
        print("ok")
    """ % ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, DBMM_MHA_VALU="1"), capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_embed_gather_and_eot():
    n, L, W, V = 5, 77, 64, 300
    table = rnd(1, "t", (V, W)); pos = rnd(2, "p", (L, W))
    tok = torch.zeros(n, L, dtype=torch.int32)
    lens = [3, 77, 10, 2, 40]
    for i, ln in enumerate(lens):
        tok[i, :ln - 1] = synth.integers(3 + i, "tk", (ln - 1,), V - 1).int()
        tok[i, ln - 1] = V - 1
    x, tk = ops.embed_gather(tok.to(DEV), table.to(DEV), pos.to(DEV))
    ref = table[tok.long()] + pos
    assert torch.equal(x.cpu(), ref)
    e = ops.gather_eot(tk, x)
    assert torch.equal(e.cpu(), ref[torch.arange(n), tok.argmax(-1)])


def test_gather_eot_tie_takes_first():
    tok = torch.tensor([[5, 9, 9, 1], [7, 7, 7, 7]], dtype=torch.int32)
    x = rnd(1, "x", (2, 4, 8))
    e = ops.gather_eot(tok.to(DEV), x.to(DEV))
    assert torch.equal(e.cpu(), torch.stack([x[0, 1], x[1, 0]]))


@pytest.mark.parametrize("R,P", [(64, 16), (28, 14), (224, 32)])
def test_im2col_patch_and_tokens(R, P):
    B, W = 2, 64
    x = rnd(1, "x", (B, 3, R, R))
    g = R // P
    am = torch.zeros(1, device=DEV)
    cols = ops.im2col_patch(x.to(DEV), P, out_absmax=am)
    assert am.item() == x.abs().max().item()
    ref = F.unfold(x, P, stride=P).transpose(1, 2).reshape(B * g * g, 3 * P * P)
    assert torch.equal(cols.cpu(), ref)
    patches = rnd(2, "pt", (B * g * g, W)); cls = rnd(3, "c", (W,)); pos = rnd(4, "pos", (g * g + 1, W))
    t = ops.vit_tokens(patches.to(DEV), cls.to(DEV), pos.to(DEV), B)
    rt = torch.cat([cls.expand(B, 1, W), patches.view(B, g * g, W)], 1) + pos
    assert torch.equal(t.cpu(), rt)


@pytest.mark.parametrize("B,H", [(4, 128), (256, 128), (1000, 136)])
def test_bn1d_stats_and_relu(B, H):
    h = rnd(1, "h", (B, H), 2.0) + 0.3
    bn = torch.nn.BatchNorm1d(H)
    bn.weight.data = synth.uniform(2, "g", (H,), 0.5, 1.5); bn.bias.data = rnd(3, "b", (H,), 0.1)
    bn.running_mean.data = rnd(4, "rm", (H,), 0.1); bn.running_var.data = synth.uniform(5, "rv", (H,), 0.5, 1.5)
    rm, rv = bn.running_mean.clone().to(DEV), bn.running_var.clone().to(DEV)
    nbt = torch.zeros((), dtype=torch.int64, device=DEV)
    bn.train()
    ref = F.relu(bn(h))
    mean = torch.empty(H, device=DEV); invstd = torch.empty(H, device=DEV)
    from dbmm_amd import _lib
    L = _lib.lib()
    hd = h.to(DEV)                      # keep alive: raw pointers are passed below
    _lib.check(L.dbmm_bn1d_stats(hd.data_ptr(), B, H, 1e-5, 0.1, mean.data_ptr(), invstd.data_ptr(),
                                 rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(), _lib.stream()))
    r = torch.empty(B, H, device=DEV)
    gam, bet = bn.weight.data.to(DEV), bn.bias.data.to(DEV)
    _lib.check(L.dbmm_bn1d_relu(hd.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gam.data_ptr(),
                                bet.data_ptr(), r.data_ptr(), B, H, 0, 1e-5, _lib.stream()))
    assert relerr(r.cpu(), ref.detach()) < 1e-5
    assert relerr(rm.cpu(), bn.running_mean) < 1e-5 and relerr(rv.cpu(), bn.running_var) < 1e-5
    assert int(nbt) == 1


@pytest.mark.parametrize("C", [2, 4, 7])
@pytest.mark.parametrize("blended", [False, True])
def test_l2norm_sim_ce(C, blended):
    B, D, T = 77, 256, 0.01
    z = rnd(1, "z", (B, D)).requires_grad_(True)
    zo = rnd(2, "zo", (B, D)) if blended else None
    text = rnd(3, "t", (D, C))
    y = synth.integers(4, "y", (B,), C)
    f = z / z.norm(dim=-1, keepdim=True)
    if blended:
        f = 0.3 * (zo / zo.norm(dim=-1, keepdim=True)) + 0.7 * f
    logits = f @ (text / text.norm(dim=0, keepdim=True)) / T
    loss = F.cross_entropy(logits, y)
    loss.backward()
    tn = ops.text_colnorm(text.to(DEV))
    assert relerr(tn.cpu(), (text / text.norm(dim=0, keepdim=True)).t()) < 1e-6
    zd = z.detach().to(DEV)
    lg, lrows, lmean, pred, inv = ops.l2norm_sim_ce_fwd(zd, tn, T, labels=y.to(DEV), z_old=zo.to(DEV) if blended else None,
                                                        ebd_weight=0.3, want_pred=True)
    assert (lg.cpu() - logits.detach()).abs().max() < 1e-3          # BASELINE.json: logits within 1e-3
    assert abs(lmean.item() - loss.item()) < 1e-4 * max(1.0, abs(loss.item()))
    assert relerr(lrows.cpu(), F.cross_entropy(logits.detach(), y, reduction="none")) < 1e-4
    assert torch.equal(pred.cpu(), logits.detach().argmax(1))
    dz = ops.l2norm_sim_ce_bwd(zd, inv, tn, T, logits=lg, labels=y.to(DEV), blended=blended, ebd_weight=0.3)
    assert relerr(dz.cpu(), z.grad) < 2e-4
    # external-criterion path: feed d(loss)/d(logits)
    dl = (torch.softmax(logits.detach(), 1) - F.one_hot(y, C)) / B
    dz2 = ops.l2norm_sim_ce_bwd(zd, inv, tn, T, dlogits=dl.float().to(DEV), blended=blended, ebd_weight=0.3)
    assert relerr(dz2.cpu(), z.grad) < 2e-4


def test_sgd_matches_torch():
    shapes = [(128, 1024), (128,), (1024, 128), (1024,), (7,)]
    ps = [torch.nn.Parameter(rnd(i, "p", s)) for i, s in enumerate(shapes)]
    opt = torch.optim.SGD(ps, lr=0.1, momentum=0.9, weight_decay=5e-5)
    dps = [p.detach().clone().to(DEV) for p in ps]
    bufs = [torch.zeros_like(p) for p in dps]
    for step in range(3):
        gs = [rnd(10 * step + i, "g", s) for i, s in enumerate(shapes)]
        for p, g in zip(ps, gs):
            p.grad = g.clone()
        opt.step()
        ops.sgd_momentum(dps, [g.to(DEV) for g in gs], bufs, 0.1, 0.9, 5e-5, step == 0)
    for p, d in zip(ps, dps):
        assert relerr(d.cpu(), p.detach()) < 1e-6


def test_group_count_bit_exact():
    B, C, G = 1000, 2, 4
    logits = rnd(1, "l", (B, C))
    logits[5] = 0.25                                           # tie -> first index
    y, c, g = synth.labels(2, B)
    counts = torch.zeros(G, 2, dtype=torch.int64, device=DEV)
    ops.group_count(logits.to(DEV), y.to(DEV), g.to(DEV), counts)
    ops.group_count(logits.to(DEV), y.to(DEV), g.to(DEV), counts)      # accumulates
    import adapter_oracle as AO
    assert (counts.cpu().numpy() == 2 * AO.group_counts(logits, y, g, G)).all()


def test_gemm_streamk_split_shapes():
    """784 tiles over 256 CUs triggers the stream-K work split (partials + fix-up kernel);
    3-image conv below cuts tiles in the middle of a tap as well."""
    M, N, K = 25088, 512, 1152
    a = rnd(1, "a", (M, K)); w = rnd(2, "w", (N, K), K ** -0.5); b = rnd(3, "b", (N,)); r = rnd(4, "r", (M, N))
    ref = F.relu(a @ w.t() + b + r)
    out = ops.gemm(a.to(DEV), w.to(DEV), b.to(DEV), residual=r.to(DEV), act=ops.ACT_RELU)
    assert relerr(out.cpu(), ref) < 2e-5
    B, H, Cin, Cout = 128, 14, 256, 256            # M = 25088 -> 392 tiles of 128x128
    x = rnd(5, "x", (B, Cin, H, H)); wc = rnd(6, "wc", (Cout, Cin, 3, 3), (Cin * 9) ** -0.5)
    refc = F.relu(F.conv2d(x, wc, None, padding=1)).permute(0, 2, 3, 1)
    outc = ops.conv_bn_act(x.permute(0, 2, 3, 1).contiguous().to(DEV), wc.permute(0, 2, 3, 1).contiguous().to(DEV),
                           None, None, 3, 3, 1, 1, ops.ACT_RELU)
    assert relerr(outc.cpu(), refc) < 2e-5


X3_SHAPES = [(512, 256, 1024), (1000, 128, 64), (300, 64, 576), (25088, 512, 1152), (130, 192, 48)]


@pytest.mark.parametrize("M,N,K", X3_SHAPES)
def test_gemm_split_precision(M, N, K):
    """three-bf16-plane operands, six partial products: fp32-level accuracy (checked against an
    fp64 reference, and no worse than the fp32-MFMA kernel), incl. a stream-K shape."""
    a = rnd(1, "a", (M, K)); w = rnd(2, "w", (N, K), K ** -0.5); b = rnd(3, "b", (N,)); r = rnd(4, "r", (M, N))
    ref = torch.relu(a.double() @ w.double().t() + b.double() + r.double())
    ad, wd, bd, rd = a.to(DEV), w.to(DEV), b.to(DEV), r.to(DEV)
    planes = ops.split_planes(wd)
    assert planes.shape == (3, N, K) and planes.dtype == torch.bfloat16
    # the three planes sum back to the fp32 weight exactly
    assert torch.equal(planes.float().sum(0), wd) or relerr(planes.double().sum(0).cpu(), w.double()) < 1e-9
    o3 = ops.gemm(ad, wd, bd, residual=rd, act=ops.ACT_RELU, w_planes=planes)
    o32 = ops.gemm(ad, wd, bd, residual=rd, act=ops.ACT_RELU)
    e3, e32 = relerr(o3.cpu().double(), ref), relerr(o32.cpu().double(), ref)
    assert e3 < 5e-6 and e3 < 5 * e32 + 5e-7, (e3, e32)


@pytest.mark.parametrize("B,H,Cin,Cout,k", [(2, 16, 32, 128, 3), (3, 14, 64, 64, 3), (2, 9, 16, 48, 3),
                                            (4, 7, 512, 256, 3), (128, 14, 256, 256, 3), (2, 12, 64, 256, 1)])
def test_conv_split_precision(B, H, Cin, Cout, k):
    x = rnd(1, "x", (B, Cin, H, H)); w = rnd(2, "w", (Cout, Cin, k, k), (Cin * k * k) ** -0.5); b = rnd(3, "b", (Cout,), 0.1)
    ref = torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=k // 2)).permute(0, 2, 3, 1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wp, wl = ops.pack_conv_weight(w.to(DEV))
    planes = ops.split_planes(wp)
    o3 = ops.conv_bn_act(xd, wp, b.to(DEV), None, k, k, 1, k // 2, ops.ACT_RELU, wl, planes)
    o32 = ops.conv_bn_act(xd, wp, b.to(DEV), None, k, k, 1, k // 2, ops.ACT_RELU, wl)
    e3, e32 = relerr(o3.cpu().double(), ref), relerr(o32.cpu().double(), ref)
    assert e3 < 5e-6 and e3 < 5 * e32 + 5e-7, (e3, e32)


def test_split_precision_falls_back_when_ineligible():
    """K % 16 != 0 -> the library runs the fp32-MFMA kernel even when planes are given"""
    M, N, K = 200, 136, 588
    a = rnd(1, "a", (M, K)); w = rnd(2, "w", (N, K), K ** -0.5)
    planes = ops.split_planes(w.to(DEV))
    out = ops.gemm(a.to(DEV), w.to(DEV), w_planes=planes)
    assert relerr(out.cpu(), a @ w.t()) < 2e-5


def _x2_conv(xd, wp, wl, bd, rd, k, act, x_bound):
    ph, we, _ = ops.split_planes_f16(wp)
    yam = torch.zeros(1, device=DEV)
    out = ops.conv_bn_act(xd, wp, bd, rd, k, k, 1, k // 2, act, wl, w_planes_f16=ph, w_exp=we,
                          x_absmax=x_bound, y_absmax=yam)
    return out, yam, ph, we


@pytest.mark.parametrize("B,H,Cin,Cout,k,res", [(2, 16, 32, 128, 3, False), (3, 14, 64, 64, 3, True), (2, 9, 16, 48, 3, False),
                                                (4, 7, 512, 256, 3, False), (128, 14, 256, 256, 3, True),
                                                (2, 12, 64, 256, 1, True), (64, 28, 128, 128, 3, False),
                                                (4, 20, 32, 32, 3, False), (3, 11, 48, 32, 3, True)])
def test_conv_fp16_pair(B, H, Cin, Cout, k, res):
    """fp16 hi+lo operands with power-of-two scales, three partial products: fp32-level accuracy
    against an fp64 reference (no worse than the fp32-MFMA kernel); the output-maximum scalar
    equals max|y| exactly; the planes reproduce the scaled weight to 2^-22."""
    x = rnd(1, "x", (B, Cin, H, H)); w = rnd(2, "w", (Cout, Cin, k, k), (Cin * k * k) ** -0.5); b = rnd(3, "b", (Cout,), 0.1)
    r = rnd(4, "r", (B, Cout, H, H)) if res else None
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=k // 2)
    if res:
        ref = ref + r.double()
    ref = torch.relu(ref).permute(0, 2, 3, 1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    rd = r.permute(0, 2, 3, 1).contiguous().to(DEV) if res else None
    wp, wl = ops.pack_conv_weight(w.to(DEV))
    o2, yam, ph, we = _x2_conv(xd, wp, wl, b.to(DEV), rd, k, ops.ACT_RELU, xd.abs().max().reshape(1))
    assert ph.shape == (2, Cout, Cin * k * k) and ph.dtype == torch.float16
    back = ph.double().sum(0) * 2.0 ** -we
    assert (back - wp.double()).abs().max().item() <= 2.0 ** -21 * wp.abs().max().item()
    assert 2.0 ** 13 <= wp.abs().max().item() * 2.0 ** we < 2.0 ** 14
    o32 = ops.conv_bn_act(xd, wp, b.to(DEV), rd, k, k, 1, k // 2, ops.ACT_RELU, wl)
    e2, e32 = relerr(o2.cpu().double(), ref), relerr(o32.cpu().double(), ref)
    assert e2 < 5e-6 and e2 < 5 * e32 + 5e-7, (e2, e32)
    assert yam.item() == o2.abs().max().item()


def test_conv_fp16_pair_loose_bound_and_outliers():
    """any upper bound of max|x| works (an average pool hands its input's bound on), and a few
    huge activations / tiny weights do not cost accuracy on the rest (absolute floor 2^-39 of the
    tensor maximum, far below fp32 rounding of the sums)."""
    B, H, Cin, Cout, k = 4, 14, 64, 128, 3
    x = rnd(1, "x", (B, Cin, H, H)); w = rnd(2, "w", (Cout, Cin, k, k), (Cin * k * k) ** -0.5); b = rnd(3, "b", (Cout,), 0.1)
    x[:, ::7, ::5, ::3] *= 300.0
    x[:, 1::4] *= 1e-4
    w[::5, ::11] *= 50.0
    ref = torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1)).permute(0, 2, 3, 1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wp, wl = ops.pack_conv_weight(w.to(DEV))
    o32 = ops.conv_bn_act(xd, wp, b.to(DEV), None, k, k, 1, 1, ops.ACT_RELU, wl)
    e32 = relerr(o32.cpu().double(), ref)
    for slack in (1.0, 3.7, 64.0):
        o2, yam, _, _ = _x2_conv(xd, wp, wl, b.to(DEV), None, k, ops.ACT_RELU, (xd.abs().max() * slack).reshape(1))
        e2 = relerr(o2.cpu().double(), ref)
        assert e2 < 5e-6 and e2 < 5 * e32 + 5e-7, (slack, e2, e32)
        assert yam.item() == o2.abs().max().item()


def test_conv_fp16_pair_stream_k():
    """forced stream-K: the fix-up kernel applies the same power-of-two rescale"""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import torch, torch.nn.functional as F, sys
        sys.path.insert(0, %r)
        import dbmm_amd
        from dbmm_amd import ops, synth
        x = synth.normal(1, "x", (128, 256, 14, 14)); w = synth.normal(2, "w", (256, 256, 3, 3), 2304 ** -0.5)
        ref = torch.relu(F.conv2d(x.double(), w.double(), None, padding=1)).permute(0, 2, 3, 1)
        xd = x.permute(0, 2, 3, 1).contiguous().cuda(); wp, wl = ops.pack_conv_weight(w.cuda())
        ph, we, _ = ops.split_planes_f16(wp); yam = torch.zeros(1, device="cuda")
        o = ops.conv_bn_act(xd, wp, None, None, 3, 3, 1, 1, ops.ACT_RELU, wl, w_planes_f16=ph, w_exp=we,
                            x_absmax=xd.abs().max().reshape(1), y_absmax=yam)
        tag = ops._last_igemm_tag()
        e = ((o.cpu().double() - ref).abs().max() / ref.abs().max()).item()
        assert tag.startswith("igemm_x3_kernel<") and tag.endswith(", 1, 2, 2, 32, 0>"), tag   # SK = 1, NP = 2, NW = 2, BK = 32, TWO = 0
        assert e < 5e-6, e
        assert yam.item() == o.abs().max().item()
        print("ok")
    """ % ROOT)
    import os
    env = dict(os.environ, DBMM_IGEMM_STREAMK="2")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_conv_fp16_pair_without_bound_uses_fp32_kernel():
    B, H, Cin, Cout, k = 2, 10, 32, 64, 3
    x = rnd(1, "x", (B, Cin, H, H)); w = rnd(2, "w", (Cout, Cin, k, k), 0.05)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV); wp, wl = ops.pack_conv_weight(w.to(DEV))
    ph, we, _ = ops.split_planes_f16(wp); yam = torch.zeros(1, device=DEV)
    o = ops.conv_bn_act(xd, wp, None, None, k, k, 1, 1, ops.ACT_NONE, wl, w_planes_f16=ph, w_exp=we, y_absmax=yam)
    assert ops._last_igemm_tag().startswith("igemm_f32_kernel<")
    ref = F.conv2d(x, w, None, padding=1).permute(0, 2, 3, 1)
    assert relerr(o.cpu(), ref) < 2e-5
    assert yam.item() == o.abs().max().item()


@pytest.mark.parametrize("B,H,Cin,Cout,k,res,x2", [(4, 14, 64, 128, 3, True, False), (128, 14, 256, 256, 3, False, True),
                                                   (2, 12, 64, 256, 1, True, False), (8, 28, 128, 64, 3, False, True),
                                                   (4, 20, 32, 32, 3, False, True), (64, 7, 512, 512, 3, False, False),
                                                   (256, 14, 256, 256, 1, True, True), (256, 14, 128, 384, 3, True, True)])
def test_conv_fp16_single_plane_with_out_scale(B, H, Cin, Cout, k, res, x2):
    """weights that are exact in fp16 (what the reference's build_model holds): one weight plane,
    two partial products, BatchNorm scale applied to the accumulator per output channel.
    Reference: fp64 conv with the fp16 weights, times the scale, plus bias."""
    x = rnd(1, "x", (B, Cin, H, H)); w = rnd(2, "w", (Cout, Cin, k, k), (Cin * k * k) ** -0.5).half().float()
    b = rnd(3, "b", (Cout,), 0.1); sc = 0.5 + synth.uniform(4, "sc", (Cout,))
    r = rnd(5, "r", (B, Cout, H, H)) if res else None
    ref = F.conv2d(x.double(), w.double(), None, padding=k // 2) * sc.double().view(1, -1, 1, 1) + b.double().view(1, -1, 1, 1)
    if res:
        ref = ref + r.double()
    ref = torch.relu(ref).permute(0, 2, 3, 1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    rd = r.permute(0, 2, 3, 1).contiguous().to(DEV) if res else None
    wp, wl = ops.pack_conv_weight(w.to(DEV))
    ph, we, n = ops.split_planes_f16(wp, allow_single=True)
    assert n == 1 and ph.shape == (1, Cout, Cin * k * k)
    assert torch.equal(ph[0].float() * 2.0 ** -we, wp)                      # the plane IS the weight
    yam = torch.zeros(1, device=DEV)
    o = ops.conv_bn_act(xd, wp, b.to(DEV), rd, k, k, 1, k // 2, ops.ACT_RELU, wl, w_planes_f16=ph, w_exp=we,
                        x_absmax=xd.abs().max().reshape(1), y_absmax=yam, out_scale=sc.to(DEV))
    tag = ops._last_igemm_tag()
    if x2:       # (small problems take the 64x64 fp32-MFMA tile, which also honours out_scale)
        assert tag.startswith("igemm_x3_kernel<") and tag.endswith(", 2, 1, 32, 0>"), tag   # NP = 2, NW = 1, BK = 32, TWO = 0
    # same op on the fp32-MFMA kernel (no planes): identical semantics of out_scale
    o32 = ops.conv_bn_act(xd, wp, b.to(DEV), rd, k, k, 1, k // 2, ops.ACT_RELU, wl, out_scale=sc.to(DEV))
    assert ops._last_igemm_tag().startswith("igemm_f32_kernel<")
    e1, e32 = relerr(o.cpu().double(), ref), relerr(o32.cpu().double(), ref)
    assert e1 < 5e-6 and e1 < 5 * e32 + 5e-7, (e1, e32)
    assert yam.item() == o.abs().max().item()


def test_split_planes_f16_single_only_when_exact():
    w = rnd(1, "w", (64, 96)).to(DEV)                                       # not fp16-representable
    assert ops.split_planes_f16(w, allow_single=True)[2] == 2
    assert ops.split_planes_f16(w.half().float(), allow_single=True)[2] == 1
    assert ops.split_planes_f16(w.half().float()[:, :80].contiguous(), allow_single=True)[2] == 2   # K % 32 != 0
    assert ops.split_planes_f16(w.half().float())[2] == 2                   # not asked for


@pytest.mark.parametrize("B,H,Cin,Cout,single,halo_pool", [
    (8, 28, 128, 64, True, 1), (4, 20, 32, 32, True, 1), (64, 28, 128, 128, True, 1), (64, 28, 128, 128, True, 0),
    (64, 28, 128, 128, False, 1), (256, 14, 64, 256, True, 1), (256, 14, 64, 256, True, 2), (3, 12, 48, 64, False, 1),
    (9, 56, 128, 128, True, 1), (131, 14, 64, 128, True, 1), (37, 26, 64, 64, True, 1), (690, 6, 64, 128, True, 1),
    (131, 14, 96, 128, True, 1), (64, 28, 128, 512, True, 2), (7, 64, 32, 96, True, 1)])
def test_conv_pool2_fused_equals_conv_then_avgpool(B, H, Cin, Cout, single, halo_pool, option):
    """conv + ReLU + AvgPool2d(2) in one epilogue (rows walked 2x2-window-major) is bit-identical
    to the same conv followed by dbmm_avgpool2d, and agrees with an fp64 reference.  halo_pool = DBMM_IGEMM_HALO_POOL:
    1 puts the pooled conv on the window-major halo kernel where Cout % 256 != 0, 2 (default) everywhere, 0 nowhere
    (per-tap kernel).  Widths 26 / 14 / 6: the 32 windows of a tile wrap over 3 / 5 / 11+ pooled rows and over images."""
    option("igemm_halo_pool", str(halo_pool))
    option("halo8", 0)                  # (Cout % 256 == 0 shapes would take conv3x3_halo8_kernel: tests/test_gpu_halo8.py)
    x = rnd(1, "x", (B, Cin, H, H)); w = rnd(2, "w", (Cout, Cin, 3, 3), (Cin * 9) ** -0.5)
    if single:
        w = w.half().float()
    b = rnd(3, "b", (Cout,), 0.1); sc = 0.5 + synth.uniform(4, "sc", (Cout,))
    ref = F.conv2d(x.double(), w.double(), None, padding=1) * sc.double().view(1, -1, 1, 1) + b.double().view(1, -1, 1, 1)
    ref = F.avg_pool2d(torch.relu(ref), 2).permute(0, 2, 3, 1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    # K order of the model (clip/model.py: 32-channel slabs, taps adjacent) for the halo_pool cases, tap-major otherwise
    wp, wl = ops.pack_conv_weight(w.to(DEV), chunk_major=32 if (halo_pool != 1 or B * H * H >= 128 * 192) else False)
    ph, we, n = ops.split_planes_f16(wp, allow_single=single)
    kw = dict(w_planes_f16=ph, w_exp=we, x_absmax=xd.abs().max().reshape(1), out_scale=sc.to(DEV))
    am1, am2 = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    fused = ops.conv_bn_act(xd, wp, b.to(DEV), None, 3, 3, 1, 1, ops.ACT_RELU, wl, y_absmax=am1, pool=2, **kw)
    tag = ops._last_igemm_tag()
    unf = ops.avgpool2d(ops.conv_bn_act(xd, wp, b.to(DEV), None, 3, 3, 1, 1, ops.ACT_RELU, wl, y_absmax=am2, **kw), 2)
    assert fused.shape == (B, H // 2, H // 2, Cout)
    stream_k = tag.startswith("igemm_x3_kernel<") and tag.split(", ")[6] == "1"
    if stream_k:     # cut tiles differ between the two row orders: sums of K segments in another order
        assert relerr(fused.cpu(), unf.cpu()) < 1e-6
    else:
        assert torch.equal(fused, unf)
    assert relerr(fused.cpu().double(), ref) < 5e-6
    big = single and Cin % 32 == 0 and B * H * H >= 192 * 128
    if big and (halo_pool == 2 or (halo_pool == 1 and Cout % 256)) and Cout > 32:
        assert tag.startswith("igemm_halo_kernel<") and tag.endswith(", 1>"), tag
    if big and halo_pool == 0:
        assert tag.startswith("igemm_x3_kernel<"), tag
    if tag.startswith("igemm_x3_kernel<") or tag.startswith("igemm_halo_kernel<"):   # fused kernel ran: its scalar is the pooled maximum
        assert am1.item() == fused.abs().max().item()
    assert am1.item() <= am2.item()


def test_conv_pool2_unsupported_shapes_compose():
    """odd output size / no planes: the library reports DBMM_E_UNSUPPORTED and ops composes conv + pool"""
    x = rnd(1, "x", (2, 32, 14, 14)); w = rnd(2, "w", (64, 32, 3, 3), 0.06)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV); wp, wl = ops.pack_conv_weight(w.to(DEV))
    ref = F.avg_pool2d(torch.relu(F.conv2d(x, w, None, padding=1)), 2).permute(0, 2, 3, 1)
    am = torch.zeros(1, device=DEV)
    out = ops.conv_bn_act(xd, wp, None, None, 3, 3, 1, 1, ops.ACT_RELU, wl, y_absmax=am, pool=2)   # no planes -> fp32 kernel
    assert relerr(out.cpu(), ref) < 2e-5
    out = ops.conv_bn_act(xd, wp, None, None, 3, 3, 1, 1, ops.ACT_RELU, wl, pool=2)                # plain entry point
    assert relerr(out.cpu(), ref) < 2e-5


@pytest.mark.parametrize("M,N,K,res,act", [(25637, 256, 256, True, 1), (25605, 200, 64, True, 0), (25600, 768, 768, False, 2), (128 * 300 + 5, 1024, 256, True, 1)])
def test_direct_epilogue_equals_staged_epilogue(M, N, K, res, act, option):
    """the fp16-pair GEMM's two epilogues -- straight from the accumulator layout (default) and staged through LDS
    (DBMM_IGEMM_EPI_DIRECT=0) -- perform the same arithmetic per element: outputs and the output maximum are bit-identical;
    ragged M (rows past M fall off the descriptor), N not a multiple of 32 (columns past N are masked lanes)"""
    a = rnd(1, "a", (M, K), 2.0).to(DEV); w = rnd(2, "w", (N, K), K ** -0.5).half().float().to(DEV); b = rnd(3, "b", (N,), 0.1).to(DEV)
    r = rnd(4, "r", (M, N)).to(DEV) if res else None
    ph, we, n = ops.split_planes_f16(w, allow_single=True)
    outs = []
    for knob in ("1", "0"):
        option("igemm_epi_direct", knob)
        am = torch.zeros(1, device=DEV)
        y = ops.gemm(a, w, b, r, act=act, w_planes_f16=ph, w_exp=we, a_absmax=a.abs().max().reshape(1), c_absmax=am)
        assert ops._last_igemm_tag().startswith("igemm_x3_kernel<"), ops._last_igemm_tag()
        assert am.item() == y.abs().max().item()
        outs.append(y)
    assert torch.equal(outs[0], outs[1])
    if not res:
        ref = a.double() @ w.double().t() + b.double()
        ref = {0: ref, 1: torch.relu(ref), 2: ref * torch.sigmoid(1.702 * ref)}[act]
        assert relerr(outs[0].double().cpu(), ref.cpu()) < 5e-6
    # the same for a conv with BatchNorm scale and residual: 1x1 (GEMM kernel) and 3x3 (halo kernel), ragged pixel count
    B, H, C = 5, 14, 64
    x = rnd(6, "x", (B, H, H, C)).to(DEV); sc = (0.5 + synth.uniform(7, "sc", (N,))).to(DEV)
    for k in (1, 3):
        wk = rnd(8, "wk", (N, C, k, k), (C * k * k) ** -0.5).half().float().to(DEV)
        wp, wl = ops.pack_conv_weight(wk, chunk_major=32)
        pk, wek, _ = ops.split_planes_f16(wp, allow_single=True)
        rr = rnd(9, "rr", (B, H, H, N)).to(DEV)
        ys = []
        for knob in ("1", "0"):
            option("igemm_epi_direct", knob)
            am = torch.zeros(1, device=DEV)
            ys.append(ops.conv_bn_act(x, wp, b, rr, k, k, 1, k // 2, ops.ACT_RELU, wl, w_planes_f16=pk, w_exp=wek,
                                      x_absmax=x.abs().max().reshape(1), y_absmax=am, out_scale=sc))
            assert am.item() == ys[-1].abs().max().item()
        assert torch.equal(ys[0], ys[1])


@pytest.mark.parametrize("M,N,K,res,act", [(16384 + 77, 512, 128, True, 2), (25600, 2304, 768, False, 0), (20000, 1024, 4096, True, 0), (16500, 256, 256, False, 0),
                                           (577 * 32, 3072, 1024, False, 2), (16384, 768, 3072, True, 1),
                                           # a short last round cut along K (tail_split): 274 tiles = 256 + 18 x 8 slices of one trip; 300 tiles =
                                           # 256 + 44 x 5 slices of 2 / 3 trips (the ViT-B/32 projections at 512 images); above: 316 = 256 + 60 x 4, 876 = 768 + 108 x 2
                                           (70000, 256, 512, True, 1), (25600, 768, 768, True, 0)])
def test_gemm_pair_deep_pipelined_kernel(M, N, K, res, act, option):
    """parity-mode GEMM on the 256 x 256 eight-phase kernel (gemm_pair_8ph.hip; N % 256 == 0, K % 64 == 0, M >= 16384):
    element-wise against fp64 -- a staging race would show as a few wrong tiles -- over repeated launches, against the
    two-barrier fp16-pair kernel (DBMM_GEMM_8PH=0) on the same operands, and the output-maximum scalar"""
    g = torch.Generator(device=DEV); g.manual_seed(M + N + K)
    a = torch.randn((M, K), device=DEV, generator=g) * 2.0; w = (torch.randn((N, K), device=DEV, generator=g) * K ** -0.5).half().float()
    b = torch.randn((N,), device=DEV, generator=g) * 0.1; r = torch.randn((M, N), device=DEV, generator=g) if res else None
    ph, we, n = ops.split_planes_f16(w, allow_single=True)
    assert n == 1
    rows = torch.cat([torch.arange(0, 600, device=DEV), torch.randint(0, M, (2000,), device=DEV, generator=g), torch.arange(M - 300, M, device=DEV)])
    v = a[rows].double() @ w.double().t() + b.double()
    if res:
        v = v + r[rows].double()
    v = {0: v, 1: torch.relu(v), 2: v * torch.sigmoid(1.702 * v)}[act]
    aam = a.abs().max().reshape(1)
    option("gemm_8ph", "0")
    am0 = torch.zeros(1, device=DEV)
    base = ops.gemm(a, w, b, r, act=act, w_planes_f16=ph, w_exp=we, a_absmax=aam, c_absmax=am0)
    assert ops._last_igemm_tag().startswith("igemm_x3_kernel<")
    option("gemm_8ph", "2")
    scale = max(1.0, v.abs().max().item())
    for _ in range(4):
        am = torch.zeros(1, device=DEV)
        out = ops.gemm(a, w, b, r, act=act, w_planes_f16=ph, w_exp=we, a_absmax=aam, c_absmax=am)
        assert ops._last_igemm_tag() == "gemm_pair_8ph_kernel"
        assert (out[rows].double() - v).abs().max().item() < 1e-5 * scale
        assert (out - base).abs().max().item() < 1e-5 * scale           # same products, another summation order
        assert am.item() == out.abs().max().item()


@pytest.mark.parametrize("M,N,K,single,res,act", [(25600, 768, 768, True, True, 0), (25600, 2304, 768, True, False, 0),
                                                  (25600, 3072, 768, True, False, 2), (1000, 256, 64, True, True, 0),
                                                  (4096, 512, 2048, False, True, 0), (616, 192, 96, False, False, 2)])
def test_gemm_fp16_pair(M, N, K, single, res, act):
    """transformer GEMMs on the fp16-pair kernel (one exact weight plane or hi + lo), QuickGELU /
    residual epilogues, output-maximum scalar, LayerNorm as the producer of the input scalar"""
    xin = rnd(1, "x", (M, K), 2.0); g = 1.0 + rnd(2, "g", (K,), 0.1); be = rnd(3, "be", (K,), 0.1)
    w = rnd(4, "w", (N, K), K ** -0.5); b = rnd(5, "b", (N,), 0.1)
    if single:
        w = w.half().float()
    r = rnd(6, "r", (M, N)) if res else None
    a_am = torch.zeros(1, device=DEV)
    a = ops.layernorm(xin.to(DEV), g.to(DEV), be.to(DEV), y_absmax=a_am)
    assert a_am.item() == a.abs().max().item()
    ref = a.cpu().double() @ w.double().t() + b.double()
    if act == 2:
        ref = ref * torch.sigmoid(1.702 * ref)
    if res:
        ref = ref + r.double()
    ph, we, n = ops.split_planes_f16(w.to(DEV), allow_single=True)
    assert n == (1 if single and K % 32 == 0 else 2)
    c_am = torch.zeros(1, device=DEV)
    out = ops.gemm(a, w.to(DEV), b.to(DEV), residual=None if r is None else r.to(DEV), act=act, w_planes_f16=ph, w_exp=we,
                   a_absmax=a_am, c_absmax=c_am)
    tag = ops._last_igemm_tag()
    o32 = ops.gemm(a, w.to(DEV), b.to(DEV), residual=None if r is None else r.to(DEV), act=act)
    e2, e32 = relerr(out.cpu().double(), ref), relerr(o32.cpu().double(), ref)
    assert e2 < 5e-6 and e2 < 5 * e32 + 5e-7, (e2, e32, tag)
    assert c_am.item() == out.abs().max().item()
    if (M // 128) * (N // 128) >= 192:     # enough 128x128 tiles for the big-tile kernels
        assert tag.startswith(("igemm_x3_kernel<", "gemm_pair_8ph_kernel")), tag


@pytest.mark.parametrize("B,H,Cin,Cout,k,single", [(64, 28, 128, 256, 1, True), (256, 14, 256, 512, 1, True),
                                                   (16, 56, 64, 256, 1, True), (64, 28, 128, 128, 3, False),
                                                   (6, 10, 32, 64, 1, True)])
def test_conv_pool2_dual_output_with_residual(B, H, Cin, Cout, k, single):
    """conv3-style launch (1x1 or 3x3) + residual + ReLU that writes BOTH the un-pooled map and its
    AvgPool2d(2): equal to the plain launch followed by the pool kernel (bit for bit without
    stream-K), and to an fp64 reference."""
    x = rnd(1, "x", (B, Cin, H, H)); w = rnd(2, "w", (Cout, Cin, k, k), (Cin * k * k) ** -0.5)
    if single:
        w = w.half().float()
    b = rnd(3, "b", (Cout,), 0.1); sc = 0.5 + synth.uniform(4, "sc", (Cout,)); r = rnd(5, "r", (B, Cout, H, H))
    ref = torch.relu(F.conv2d(x.double(), w.double(), None, padding=k // 2) * sc.double().view(1, -1, 1, 1)
                     + b.double().view(1, -1, 1, 1) + r.double())
    ref_p = F.avg_pool2d(ref, 2).permute(0, 2, 3, 1); ref = ref.permute(0, 2, 3, 1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV); rd = r.permute(0, 2, 3, 1).contiguous().to(DEV)
    wp, wl = ops.pack_conv_weight(w.to(DEV))
    ph, we, n = ops.split_planes_f16(wp, allow_single=single)
    kw = dict(w_planes_f16=ph, w_exp=we, x_absmax=xd.abs().max().reshape(1), out_scale=sc.to(DEV))
    am1, am2 = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    pooled, full = ops.conv_bn_act(xd, wp, b.to(DEV), rd, k, k, 1, k // 2, ops.ACT_RELU, wl, y_absmax=am1, pool=2,
                                   keep_full=True, **kw)
    tag = ops._last_igemm_tag()
    plain = ops.conv_bn_act(xd, wp, b.to(DEV), rd, k, k, 1, k // 2, ops.ACT_RELU, wl, y_absmax=am2, **kw)
    stream_k = tag.startswith("igemm_x3_kernel<") and tag.split(", ")[6] == "1"
    if stream_k:
        assert relerr(full.cpu(), plain.cpu()) < 1e-6 and relerr(pooled.cpu(), ops.avgpool2d(plain, 2).cpu()) < 1e-6
    else:
        assert torch.equal(full, plain) and torch.equal(pooled, ops.avgpool2d(plain, 2))
    assert relerr(full.cpu().double(), ref) < 5e-6 and relerr(pooled.cpu().double(), ref_p) < 5e-6
    assert am1.item() == full.abs().max().item()


@pytest.mark.parametrize("B,H,Cin,Cout,single,pool", [(64, 28, 128, 128, True, 1), (8, 28, 64, 64, True, 2),
                                                      (128, 14, 256, 256, False, 1), (3, 10, 32, 48, True, 1)])
def test_conv_chunk32_major_layout(B, H, Cin, Cout, single, pool):
    """K packed (cin/32, kh, kw, 32) -- the nine taps of a 32-channel slab adjacent along K -- gives
    the same convolution on every kernel family (fp16-pair 32-deep, fp32-MFMA fallback)."""
    x = rnd(1, "x", (B, Cin, H, H)); w = rnd(2, "w", (Cout, Cin, 3, 3), (Cin * 9) ** -0.5)
    if single:
        w = w.half().float()
    b = rnd(3, "b", (Cout,), 0.1)
    ref = torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1))
    if pool == 2:
        ref = F.avg_pool2d(ref, 2)
    ref = ref.permute(0, 2, 3, 1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wp, wl = ops.pack_conv_weight(w.to(DEV), chunk_major=32)
    assert wl == ops.WL_CHUNK32_MAJOR
    wt, _ = ops.pack_conv_weight(w.to(DEV))
    assert torch.equal(wp.reshape(Cout, Cin // 32, 9, 32).permute(0, 2, 1, 3).reshape(Cout, -1), wt)
    ph, we, n = ops.split_planes_f16(wp, allow_single=single)
    am = torch.zeros(1, device=DEV)
    o2 = ops.conv_bn_act(xd, wp, b.to(DEV), None, 3, 3, 1, 1, ops.ACT_RELU, wl, w_planes_f16=ph, w_exp=we,
                         x_absmax=xd.abs().max().reshape(1), y_absmax=am, pool=pool)
    o32 = ops.conv_bn_act(xd, wp, b.to(DEV), None, 3, 3, 1, 1, ops.ACT_RELU, wl, pool=pool)     # fp32-MFMA kernel
    e2, e32 = relerr(o2.cpu().double(), ref), relerr(o32.cpu().double(), ref)
    assert e32 < 2e-5 and e2 < 5e-6 and e2 < 5 * e32 + 5e-7, (e2, e32)


@pytest.mark.parametrize("B,H,W,Cin,Cout,res", [(8, 56, 56, 64, 64, False), (128, 14, 14, 256, 256, False),
                                                (512, 7, 7, 512, 512, False), (33, 28, 28, 128, 128, False),
                                                (9, 40, 72, 64, 192, False), (70, 19, 19, 128, 128, True),
                                                (4, 112, 112, 32, 32, False), (16, 40, 40, 32, 64, False),
                                                (40, 26, 26, 96, 128, True)])
def test_conv3x3_halo_kernel(B, H, W, Cin, Cout, res, option):
    """3x3 conv with the activation tile reused across the kw taps (LDS rows shifted, border taps
    redirected to a zero row): equal to the per-tap kernel up to summation order, and to fp64.
    Shapes cover every image-border case (7x7 ... 56x56, non-square, M not a multiple of 128,
    stream-K and plain grids, residual)."""
    x = rnd(1, "x", (B, Cin, H, W)); w = rnd(2, "w", (Cout, Cin, 3, 3), (Cin * 9) ** -0.5).half().float()
    b = rnd(3, "b", (Cout,), 0.1); sc = 0.5 + synth.uniform(4, "sc", (Cout,))
    r = rnd(5, "r", (B, Cout, H, W)) if res else None
    ref = F.conv2d(x.double(), w.double(), None, padding=1) * sc.double().view(1, -1, 1, 1) + b.double().view(1, -1, 1, 1)
    if res:
        ref = ref + r.double()
    ref = torch.relu(ref).permute(0, 2, 3, 1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    rd = r.permute(0, 2, 3, 1).contiguous().to(DEV) if res else None
    wp, wl = ops.pack_conv_weight(w.to(DEV), chunk_major=32)
    ph, we, n = ops.split_planes_f16(wp, allow_single=True)
    assert n == 1
    kw = dict(w_planes_f16=ph, w_exp=we, x_absmax=xd.abs().max().reshape(1), out_scale=sc.to(DEV))
    am = torch.zeros(1, device=DEV)
    option("igemm_halo", "1")
    option("halo8", 0)                 # (this test is about igemm_halo_kernel; conv3x3_halo8_kernel: tests/test_gpu_halo8.py)
    option("conv_patch", "0")          # (the 32-channel stem shape would otherwise take the patch kernel)
    o = ops.conv_bn_act(xd, wp, b.to(DEV), rd, 3, 3, 1, 1, ops.ACT_RELU, wl, y_absmax=am, **kw)
    tag = ops._last_igemm_tag()
    assert tag.startswith("igemm_halo_kernel<"), tag
    option("igemm_halo", "0")
    o_tap = ops.conv_bn_act(xd, wp, b.to(DEV), rd, 3, 3, 1, 1, ops.ACT_RELU, wl, **kw)
    assert ops._last_igemm_tag().startswith("igemm_x3_kernel<")
    assert relerr(o.cpu(), o_tap.cpu()) < 2e-6
    assert relerr(o.cpu().double(), ref) < 5e-6
    assert am.item() == o.abs().max().item()


@pytest.mark.parametrize("eight", [0, 2])
@pytest.mark.parametrize("M,N,K,K2", [(25088, 256, 64, 64), (50176, 512, 128, 256), (25088, 1024, 256, 512),
                                      (25000, 2048, 512, 1024)])
def test_gemm_dual_conv3_plus_downsample(M, N, K, K2, eight, option):
    """relu(bn3(conv3(y)) + bn_d(conv_d(x))) as one dual-source GEMM == the two fp32-accurate
    launches (conv_d, then conv3 with the residual) and an fp64 reference; incl. stream-K shapes
    and an M tail."""
    y = torch.relu(rnd(1, "y", (M, K))); x = torch.relu(rnd(2, "x", (M, K2), 3.0))          # different ranges
    w = rnd(3, "w", (N, K), K ** -0.5).half().float(); w2 = rnd(4, "w2", (N, K2), K2 ** -0.5).half().float()
    sc = 0.5 + synth.uniform(5, "sc", (N,)); sc2 = 0.5 + synth.uniform(6, "sc2", (N,))
    b = rnd(7, "b", (N,), 0.1); b2 = rnd(8, "b2", (N,), 0.1)
    ref = torch.relu(y.double() @ w.double().t() * sc.double() + b.double() + x.double() @ w2.double().t() * sc2.double()
                     + b2.double())
    yd, xd = y.to(DEV), x.to(DEV)
    ph, we, n1 = ops.split_planes_f16(w.to(DEV), allow_single=True); ph2, we2, n2 = ops.split_planes_f16(w2.to(DEV), allow_single=True)
    assert n1 == 1 and n2 == 1
    ratio = (sc2.double() / sc.double() * 2.0 ** (we - we2)).float().to(DEV)
    ya, xa = yd.abs().max().reshape(1), (xd.abs().max() * 1.7).reshape(1)               # a bound, not the exact max
    cam = torch.zeros(1, device=DEV)
    option("dual_8ph", eight)        # 2: gemm_pair_8ph_kernel with TWO = 1 (every shape here suits it), 0: the 128 x 128 kernel
    out = ops.gemm_dual(yd, ya, ph, we, sc.to(DEV), xd, xa, ph2, ratio, (b + b2).to(DEV), ops.ACT_RELU, cam)
    assert out is not None
    assert ops._last_igemm_tag() == "gemm_pair_8ph_kernel<dual>" if eight else ops._last_igemm_tag().endswith(", 2, 1, 32, 1>")
    # unfused: identity = conv_d(x) * sc2 + b2 ; out = relu(conv3(y) * sc + b + identity)
    v = lambda t: t.reshape(1, 1, M, -1)
    ident = ops.conv_bn_act(v(xd), w2.to(DEV), b2.to(DEV), None, 1, 1, 1, 0, ops.ACT_NONE, w_planes_f16=ph2, w_exp=we2,
                            x_absmax=xa, out_scale=sc2.to(DEV))
    unf = ops.conv_bn_act(v(yd), w.to(DEV), b.to(DEV), ident, 1, 1, 1, 0, ops.ACT_RELU, w_planes_f16=ph, w_exp=we,
                          x_absmax=ya, out_scale=sc.to(DEV)).reshape(M, N)
    e_f, e_u = relerr(out.cpu().double(), ref), relerr(unf.cpu().double(), ref)
    assert e_f < 5e-6 and e_f < 5 * e_u + 5e-7, (e_f, e_u)
    assert cam.item() == out.abs().max().item()


def test_gemm_dual_reports_unsupported_shapes():
    y = rnd(1, "y", (256, 64)).to(DEV); x = rnd(2, "x", (256, 64)).to(DEV)
    w = rnd(3, "w", (128, 64)).half().float().to(DEV)
    ph, we, _ = ops.split_planes_f16(w, allow_single=True)
    one = torch.ones(128, device=DEV)
    assert ops.gemm_dual(y, y.abs().max().reshape(1), ph, we, one, x, x.abs().max().reshape(1), ph, one, one) is None


def _last_cfg():
    import ctypes
    from dbmm_amd import _lib
    cfg = (ctypes.c_int * 11)()
    _lib.lib().dbmm_debug_last_igemm(cfg)
    return list(cfg)


def test_operands_over_2gib_stay_on_the_split_kernels(option):
    """RN50 layer 1 at the headline batch 1024 holds 3.3 GB activation tensors.  Buffer descriptors
    address 32 bits, so every tile rebases its descriptor on a 64-bit base (a_desc in igemm_f32.hip):
    operands past 2 GiB must still run the fp16-pair / halo / dual-source kernels (cfg[8] in 2, 4, 5 --
    not the fp32 fallback) and give the same numbers as the same images in a small batch."""
    torch.manual_seed(0)
    g = torch.Generator(device=DEV); g.manual_seed(1)
    # --- 1x1 conv, 256 -> 64 channels on 56x56 maps: 720 images = 2.31 GB input
    B, H, Cin, Cout = 720, 56, 256, 64
    x = torch.randn((B, H, H, Cin), device=DEV, generator=g)
    assert x.numel() * 4 > 2 ** 31
    w = (torch.randn((Cout, Cin), device=DEV, generator=g) * Cin ** -0.5).half().float()
    sc = 0.5 + torch.rand((Cout,), device=DEV, generator=g); b = torch.randn((Cout,), device=DEV, generator=g) * 0.1
    ph, we, n = ops.split_planes_f16(w, allow_single=True)
    assert n == 1
    xa = x.abs().max().reshape(1)
    am = torch.zeros(1, device=DEV)
    y = ops.conv_bn_act(x, w, b, None, 1, 1, 1, 0, ops.ACT_RELU, w_planes_f16=ph, w_exp=we, x_absmax=xa, y_absmax=am,
                        out_scale=sc)
    assert _last_cfg()[8] == 2, _last_cfg()
    for i in (0, 359, B - 1):           # first image, one straddling the 2 GiB line's neighbourhood, the last
        ref = torch.relu(x[i].double() @ w.double().t() * sc.double() + b.double())
        assert relerr(y[i].double().cpu(), ref.cpu()) < 5e-6, i
    assert am.item() == y.abs().max().item()
    # --- the same tensor as the branch operand of the dual-source GEMM (conv3 + downsample)
    M = B * H * H
    y2 = torch.relu(torch.randn((M, 64), device=DEV, generator=g))
    w3 = (torch.randn((256, 64), device=DEV, generator=g) * 64 ** -0.5).half().float()
    wd = (torch.randn((256, Cin), device=DEV, generator=g) * Cin ** -0.5).half().float()
    p3, e3, _ = ops.split_planes_f16(w3, allow_single=True); pd, ed, _ = ops.split_planes_f16(wd, allow_single=True)
    s3 = 0.5 + torch.rand((256,), device=DEV, generator=g); sd = 0.5 + torch.rand((256,), device=DEV, generator=g)
    ratio = (sd.double() / s3.double() * 2.0 ** (e3 - ed)).float()
    bb = torch.randn((256,), device=DEV, generator=g) * 0.1
    for knob, kind in ((0, 5), (2, 8)):          # the 128 x 128 dual-source kernel, then the eight-phase one (both rebase their descriptors per tile)
        ops.set_option("dual_8ph", knob)
        out = ops.gemm_dual(y2, y2.abs().max().reshape(1), p3, e3, s3, x.view(M, Cin), xa, pd, ratio, bb, ops.ACT_RELU)
        assert out is not None and _last_cfg()[8] == kind, _last_cfg()
        assert out.numel() * 4 > 2 ** 31
        for r0 in (0, M // 2 - 64, M - 200):
            sl = slice(r0, r0 + 200)
            ref = torch.relu(y2[sl].double() @ w3.double().t() * s3.double() + x.view(M, Cin)[sl].double() @ wd.double().t() * sd.double()
                             + bb.double())
            assert relerr(out[sl].double().cpu(), ref.cpu()) < 5e-6, r0
    ops.set_option("dual_8ph", 1)
    # --- conv3-style 1x1 with a residual > 2 GiB and the pooled second output (64-bit row pointers in the epilogue)
    res = out.view(B, H, H, 256)
    (yp, yf), _ = (ops.conv_bn_act(y2.view(B, H, H, 64), w3, bb, res, 1, 1, 1, 0, ops.ACT_RELU, w_planes_f16=p3, w_exp=e3,
                                   x_absmax=y2.abs().max().reshape(1), out_scale=s3, pool=2, keep_full=True), None)
    assert _last_cfg()[8] == 2
    i = B - 1
    ref = torch.relu(y2.view(B, H * H, 64)[i].double() @ w3.double().t() * s3.double() + bb.double() + res[i].view(H * H, 256).double())
    assert relerr(yf[i].view(H * H, 256).double().cpu(), ref.cpu()) < 5e-6
    refp = F.avg_pool2d(ref.view(H, H, 256).permute(2, 0, 1)[None], 2)[0].permute(1, 2, 0)
    assert relerr(yp[i].double().cpu(), refp.cpu()) < 5e-6
    del out, res, yp, yf, y2, x, y
    torch.cuda.empty_cache()
    # --- 3x3 halo kernel, 64 -> 64 channels on 56x56 maps: 2880 images = 2.31 GB input
    B, Cin, Cout = 2880, 64, 64
    x = torch.randn((B, H, H, Cin), device=DEV, generator=g)
    assert x.numel() * 4 > 2 ** 31
    w = (torch.randn((Cout, Cin, 3, 3), device=DEV, generator=g) * (9 * Cin) ** -0.5).half().float()
    wp, wl = ops.pack_conv_weight(w, chunk_major=32)
    ph, we, n = ops.split_planes_f16(wp, allow_single=True)
    y = ops.conv_bn_act(x, wp, b, None, 3, 3, 1, 1, ops.ACT_RELU, wl, w_planes_f16=ph, w_exp=we,
                        x_absmax=x.abs().max().reshape(1), out_scale=sc)
    assert _last_cfg()[8] == 4, _last_cfg()
    for i in (0, 1439, 1440, B - 1):
        ref = F.conv2d(x[i].permute(2, 0, 1)[None].double(), w.double(), None, padding=1)[0].permute(1, 2, 0)
        ref = torch.relu(ref * sc.double() + b.double())
        assert relerr(y[i].double().cpu(), ref.cpu()) < 5e-6, i
    # --- the same tensor pooled (window-major rows): the halo kernel's pooled variant, then the per-tap conv kernel
    for knob, kind in (("2", 4), ("0", 2)):
        option("igemm_halo_pool", knob)
        yp = ops.conv_bn_act(x, wp, b, None, 3, 3, 1, 1, ops.ACT_RELU, wl, w_planes_f16=ph, w_exp=we,
                             x_absmax=x.abs().max().reshape(1), out_scale=sc, pool=2)
        cfg = _last_cfg()
        assert cfg[8] == kind and (cfg[5] == 1 if kind == 4 else cfg[4] == 1), cfg
        for i in (0, 1440, B - 1):
            assert relerr(yp[i].cpu(), F.avg_pool2d(y[i].permute(2, 0, 1)[None], 2)[0].permute(1, 2, 0).cpu()) < 2e-6, i
        del yp


@pytest.mark.parametrize("B,H,W,K,N,P,pooled", [(8, 56, 56, 64, 256, 64, False), (8, 56, 56, 64, 256, 128, True), (3, 10, 14, 64, 128, 64, False),
                                                (5, 6, 10, 64, 192, 128, True), (1, 2, 2, 64, 64, 64, True), (40, 56, 56, 64, 256, 64, True),
                                                (3, 10, 14, 64, 128, 128, False), (16, 28, 28, 128, 512, 128, False),
                                                (9, 28, 28, 128, 512, 128, True), (3, 10, 14, 128, 256, 64, False),
                                                (2, 6, 6, 128, 64, 64, True),
                                                # layer-3 geometry (K = P = 256): the eight-wave kernel of bottleneck_chain8.hip
                                                (64, 14, 14, 256, 1024, 256, False), (3, 10, 14, 256, 128, 256, False),
                                                (1, 2, 2, 256, 64, 256, False), (5, 6, 10, 256, 192, 256, False),
                                                (300, 14, 14, 256, 1024, 256, False)])
def test_bottleneck_chain_conv3_residual_then_next_conv1(B, H, W, K, N, P, pooled, option):
    """relu(bn3(conv3(y2)) + x) and the NEXT block's relu(bn1(conv1(.))) as one launch (clip/model.py:42-55 twice) ==
    fp64, and == the two separate fp32-accurate launches; optional 2x2-pooled copy == avg-pool of the written x';
    maxima == the written tensors' maxima.  Shapes: layer-1 / layer-2 / layer-3 geometry (K = 64 / 128 / 256), ragged M (not a
    multiple of 128), one window, more tiles than workgroup slots (B = 300 at K = 256: 460 tiles over 256 CUs)."""
    if K == 256:
        option("chain8", 1)                                           # (off by default: measured no faster than the two launches)
    g = torch.Generator(device=DEV); g.manual_seed(B * 1000 + N + P)
    y2 = torch.relu(torch.randn((B, H, W, K), device=DEV, generator=g))
    res = torch.relu(torch.randn((B, H, W, N), device=DEV, generator=g) * 3.0)
    res[0, 0, 0, :7] = 300.0                                          # one wave's slab maximum far above the others'
    w3 = (torch.randn((N, K), device=DEV, generator=g) * K ** -0.5).half().float()
    w1 = (torch.randn((P, N), device=DEV, generator=g) * N ** -0.5).half().float()
    mk = lambda n: (0.5 + torch.rand((n,), device=DEV, generator=g), torch.randn((n,), device=DEV, generator=g) * 0.1)
    (s3, b3), (s1, b1) = mk(N), mk(P)
    p3, e3, n3 = ops.split_planes_f16(w3, allow_single=True); p1, e1, n1 = ops.split_planes_f16(w1, allow_single=True)
    assert n3 == 1 and n1 == 1
    c3 = dict(w=w3, ph=p3, we=e3, sc=s3, b=b3); c1 = dict(w=w1, ph=p1, we=e1, sc=s1, b=b1)
    ya = (y2.abs().max() * 1.3).reshape(1)
    xam, yam = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    r = ops.bottleneck_chain(y2, ya, c3, res, c1, xam, yam, pooled=pooled)
    assert r is not None
    x, y1 = r[0], r[-1]
    M = B * H * W
    xr = torch.relu(y2.view(M, K).double() @ w3.double().t() * s3.double() + b3.double() + res.view(M, N).double())
    yr = torch.relu(xr @ w1.double().t() * s1.double() + b1.double())
    assert relerr(x.view(M, N).double().cpu(), xr.cpu()) < 5e-6
    assert relerr(y1.view(M, P).double().cpu(), yr.cpu()) < 5e-6
    assert xam.item() == max(x.abs().max().item(), r[1].abs().max().item() if pooled else 0.0) and yam.item() == y1.abs().max().item()
    if pooled:
        ref_p = F.avg_pool2d(x.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
        assert tuple(r[1].shape) == (B, H // 2, W // 2, N) and relerr(r[1].cpu(), ref_p.cpu()) < 2e-6
        # pooled only (a stage seam: nobody reads the un-pooled tensor): same pooled copy, conv1 output and maxima, bit for bit
        xam2, yam2 = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
        r2 = ops.bottleneck_chain(y2, ya, c3, res, c1, xam2, yam2, pooled=True, keep_full=False)
        assert r2 is not None and r2[0] is None and ops._chain_tag == f"bottleneck_chain_kernel<{K}, {P}, 2, 0>"
        assert torch.equal(r2[1], r[1]) and torch.equal(r2[2], y1) and xam2.item() == xam.item() and yam2.item() == yam.item()
    # the two separate launches
    xu = ops.conv_bn_act(y2, w3, b3, res, 1, 1, 1, 0, ops.ACT_RELU, w_planes_f16=p3, w_exp=e3, x_absmax=ya, out_scale=s3)
    yu = ops.conv_bn_act(xu, w1, b1, None, 1, 1, 1, 0, ops.ACT_RELU, w_planes_f16=p1, w_exp=e1,
                         x_absmax=xu.abs().max().reshape(1), out_scale=s1)
    assert relerr(x.cpu(), xu.cpu()) < 2e-6 and relerr(y1.cpu(), yu.cpu()) < 3e-6


def test_bottleneck_chain_reports_unsupported_shapes():
    g = torch.Generator(device=DEV); g.manual_seed(3)
    mk = lambda n, k: (torch.randn((n, k), device=DEV, generator=g) * k ** -0.5).half().float()
    def entry(w):
        ph, we, _ = ops.split_planes_f16(w, allow_single=True)
        return dict(w=w, ph=ph, we=we, sc=torch.ones(w.shape[0], device=DEV), b=torch.zeros(w.shape[0], device=DEV))
    y = torch.rand((2, 4, 4, 256), device=DEV); res = torch.rand((2, 4, 4, 1024), device=DEV)
    assert ops.bottleneck_chain(y, y.max().reshape(1), entry(mk(1024, 256)), res, entry(mk(128, 1024))) is None    # K = 256, P = 128
    y = torch.rand((2, 4, 4, 512), device=DEV); res = torch.rand((2, 4, 4, 2048), device=DEV)
    assert ops.bottleneck_chain(y, y.max().reshape(1), entry(mk(2048, 512)), res, entry(mk(512, 2048))) is None    # K = 512 (layer 4)
    y = torch.rand((2, 4, 4, 64), device=DEV); res = torch.rand((2, 4, 4, 256), device=DEV)
    assert ops.bottleneck_chain(y, y.max().reshape(1), entry(mk(256, 64)), res, entry(mk(256, 256))) is None        # P = 256
    y = torch.rand((1, 3, 3, 64), device=DEV); res = torch.rand((1, 3, 3, 256), device=DEV)
    assert ops.bottleneck_chain(y, y.max().reshape(1), entry(mk(256, 64)), res, entry(mk(64, 256))) is None         # M % 4 != 0


@pytest.mark.parametrize("M,N,P", [(25088, 256, 64), (25088, 256, 128), (300, 128, 64)])
def test_bottleneck_chain_dual_first_block(M, N, P):
    """first block of layer 1: relu(bn3(conv3(y2)) + bn_d(conv_d(x))) and the next block's conv1 as one launch == fp64
    and == the dual-source GEMM followed by the separate conv1."""
    K = K2 = 64
    g = torch.Generator(device=DEV); g.manual_seed(M + N + P)
    y2 = torch.relu(torch.randn((M, K), device=DEV, generator=g)); a2 = torch.relu(torch.randn((M, K2), device=DEV, generator=g) * 4.0)
    mkw = lambda n, k: (torch.randn((n, k), device=DEV, generator=g) * k ** -0.5).half().float()
    w3, wd, w1 = mkw(N, K), mkw(N, K2), mkw(P, N)
    mk = lambda n: (0.5 + torch.rand((n,), device=DEV, generator=g), torch.randn((n,), device=DEV, generator=g) * 0.1)
    (s3, b3), (sd, bd), (s1, b1) = mk(N), mk(N), mk(P)
    p3, e3, _ = ops.split_planes_f16(w3, allow_single=True); pd, ed, _ = ops.split_planes_f16(wd, allow_single=True)
    p1, e1, _ = ops.split_planes_f16(w1, allow_single=True)
    ratio = (sd.double() / s3.double() * 2.0 ** (e3 - ed)).float()
    c3 = dict(ph=p3, we=e3, sc=s3, b=b3); ds = dict(ph=pd, we=ed, sc=sd, b=bd); c1 = dict(ph=p1, we=e1, sc=s1, b=b1)
    ya, xa = y2.abs().max().reshape(1), (a2.abs().max() * 1.9).reshape(1)
    xam, yam = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    r = ops.bottleneck_chain_dual(y2, ya, c3, a2, xa, ds, ratio, b3 + bd, c1, xam, yam)
    assert r is not None
    x, y1 = r
    xr = torch.relu(y2.double() @ w3.double().t() * s3.double() + a2.double() @ wd.double().t() * sd.double() + (b3 + bd).double())
    yr = torch.relu(xr @ w1.double().t() * s1.double() + b1.double())
    assert relerr(x.double().cpu(), xr.cpu()) < 5e-6 and relerr(y1.double().cpu(), yr.cpu()) < 5e-6
    assert xam.item() == x.abs().max().item() and yam.item() == y1.abs().max().item()
    if M >= 24576:                                      # the dual-source GEMM serves grids of >= 192 tiles
        xu = ops.gemm_dual(y2, ya, p3, e3, s3, a2, xa, pd, ratio, b3 + bd, ops.ACT_RELU)
        assert xu is not None and relerr(x.cpu(), xu.cpu()) < 2e-6


@pytest.mark.parametrize("B,H,W,Cout,pool", [(2, 112, 112, 32, 1), (2, 112, 112, 64, 2), (3, 8, 28, 64, 1), (5, 4, 56, 32, 2),
                                             (1, 12, 84, 64, 2), (300, 16, 28, 32, 1)])
def test_conv3x3_c32_patch_kernel(B, H, W, Cout, pool, option):
    """the stem's 32-channel 3x3 convs on the persistent patch kernel == fp64 and == the implicit-GEMM kernels
    (DBMM_CONV_PATCH=0), un-pooled and with the fused 2x2 average pool; image borders, several tiles per row, more tiles
    than resident workgroups (B = 300: 1200 tiles over 768 slots)."""
    g = torch.Generator(device=DEV); g.manual_seed(B + H + W + Cout)
    x = torch.relu(torch.randn((B, H, W, 32), device=DEV, generator=g) * 2.0)
    w = (torch.randn((Cout, 32, 3, 3), device=DEV, generator=g) * 288 ** -0.5).half().float()
    sc = 0.5 + torch.rand((Cout,), device=DEV, generator=g); b = torch.randn((Cout,), device=DEV, generator=g) * 0.1
    wp, wl = ops.pack_conv_weight(w, chunk_major=32)
    ph, we, n = ops.split_planes_f16(wp, allow_single=True)
    assert n == 1
    xa = (x.abs().max() * 1.5).reshape(1)
    kw = dict(w_planes_f16=ph, w_exp=we, x_absmax=xa, out_scale=sc, pool=pool)
    am = torch.zeros(1, device=DEV)
    option("conv_patch", "1")
    y = ops.conv_bn_act(x, wp, b, None, 3, 3, 1, 1, ops.ACT_RELU, wl, y_absmax=am, **kw)
    ref = torch.relu(F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, padding=1) * sc.double().view(1, -1, 1, 1)
                     + b.double().view(1, -1, 1, 1))
    assert am.item() >= y.abs().max().item() and am.item() <= ref.max().item() * (1 + 1e-5)
    if pool == 2:
        ref = F.avg_pool2d(ref, 2)
    ref = ref.permute(0, 2, 3, 1)
    assert tuple(y.shape) == tuple(ref.shape) and relerr(y.double().cpu(), ref.cpu()) < 5e-6
    option("conv_patch", "0")
    y0 = ops.conv_bn_act(x, wp, b, None, 3, 3, 1, 1, ops.ACT_RELU, wl, **kw)
    assert relerr(y.cpu(), y0.cpu()) < 2e-6


@pytest.mark.parametrize("B,L,heads,causal", [(3, 50, 2, False), (2, 77, 8, True), (2, 130, 1, False), (1, 577, 2, False),
                                              (2, 1, 1, False), (1, 128, 2, True), (2, 129, 1, True), (2, 300, 2, True),
                                              (2, 64, 2, True), (3, 33, 1, True)])         # <= 64 tokens: the two-wave workgroups
@pytest.mark.parametrize("bound", [1.0, 37.0])
def test_mha_core_fp16_pair_kernel(B, L, heads, causal, bound):
    """parity-mode attention core on fp16-pair products (three partial products, scale from a bound of max|qkv|) against
    fp64 softmax attention, and no worse than the fp32-input-MFMA kernel; exact and loose bounds."""
    E = heads * 64
    qkv = rnd(1, "qkv", (B, L, 3 * E)) * 1.7
    q, k, v = qkv.double().split(E, dim=-1)
    sh = lambda t: t.reshape(B, L, heads, 64).transpose(1, 2)
    s = (sh(q) * 0.125) @ sh(k).transpose(-1, -2)
    if causal:
        s = s + torch.full((L, L), float("-inf"), dtype=torch.float64).triu_(1)
    ref = (torch.softmax(s, -1) @ sh(v)).transpose(1, 2).reshape(B * L, E)
    qd = qkv.to(DEV).view(B * L, 3 * E)
    out = ops.mha_core(qd, B, L, E, heads, causal, qkv_absmax=(qd.abs().max() * bound).reshape(1))
    base = ops.mha_core(qd, B, L, E, heads, causal)
    e_pair, e_f32 = relerr(out.double().cpu(), ref), relerr(base.double().cpu(), ref)
    assert e_pair < 5e-6 and e_pair < 4 * e_f32 + 1e-6, (e_pair, e_f32)


def test_mha_core_fp16_pair_spiked_scores():
    B, L, E = 1, 200, 64
    qkv = rnd(1, "qkv", (B, L, 3 * E))
    qkv[0, :, E:2 * E][37] *= 40.0
    qkv[0, :, E:2 * E][150] *= -60.0
    q, k, v = qkv.double().split(E, dim=-1)
    ref = torch.softmax((q * 0.125) @ k.transpose(-1, -2), -1) @ v
    qd = qkv.to(DEV).view(L, 3 * E)
    out = ops.mha_core(qd, B, L, E, 1, False, qkv_absmax=qd.abs().max().reshape(1))
    assert torch.isfinite(out).all() and relerr(out.double().cpu(), ref.reshape(L, E)) < 5e-6


@pytest.mark.parametrize("B,H,W,K,N,P,dual", [(8, 56, 56, 64, 256, 64, False), (8, 56, 56, 64, 256, 64, True), (16, 28, 28, 128, 512, 128, False),
                                              (3, 10, 14, 64, 128, 64, False), (5, 6, 6, 128, 256, 128, False), (3, 10, 14, 64, 128, 64, True),
                                              (70, 7, 8, 64, 64, 64, False)])
def test_bottleneck_block_chain_conv2_conv3_next_conv1(B, H, W, K, N, P, dual):
    """conv2 3x3 + bn + relu -> conv3 + bn + (residual | downsample branch) + relu -> next block's conv1 + bn + relu as ONE
    launch (clip/model.py:42-55) == fp64 and == the separate launches; image borders, images smaller than a tile (several
    images and an M tail in one 128-pixel tile), layer-1 / layer-2 geometry."""
    g = torch.Generator(device=DEV); g.manual_seed(B * 7 + H + K + N + int(dual))
    rn = lambda *sh: torch.randn(sh, device=DEV, generator=g)
    y1 = torch.relu(rn(B, H, W, K) * 1.5)
    w2 = (rn(K, K, 3, 3) * (9 * K) ** -0.5).half().float(); w3 = (rn(N, K) * K ** -0.5).half().float(); w1 = (rn(P, N) * N ** -0.5).half().float()
    mk = lambda n: (0.5 + torch.rand((n,), device=DEV, generator=g), rn(n) * 0.1)
    (s2, b2), (s3, b3), (s1, b1) = mk(K), mk(N), mk(P)
    w2p, wl = ops.pack_conv_weight(w2, chunk_major=32)
    p2, e2, n2 = ops.split_planes_f16(w2p, allow_single=True); p3, e3, _ = ops.split_planes_f16(w3, allow_single=True)
    p1, e1, _ = ops.split_planes_f16(w1, allow_single=True)
    assert n2 == 1 and wl == ops.WL_CHUNK32_MAJOR
    c2 = dict(w=w2p, wl=wl, ph=p2, we=e2, sc=s2, b=b2); c3 = dict(ph=p3, we=e3, sc=s3, b=b3); c1 = dict(ph=p1, we=e1, sc=s1, b=b1)
    ya = (y1.abs().max() * 1.2).reshape(1)
    xam, yam = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    M = B * H * W
    y2r = torch.relu(F.conv2d(y1.permute(0, 3, 1, 2).double(), w2.double(), None, padding=1) * s2.double().view(1, -1, 1, 1)
                     + b2.double().view(1, -1, 1, 1)).permute(0, 2, 3, 1).reshape(M, K)
    if dual:
        a2 = torch.relu(rn(B, H, W, 64) * 3.0); wd = (rn(N, 64) * 0.125).half().float(); sd, bd = mk(N)
        pd, ed, _ = ops.split_planes_f16(wd, allow_single=True)
        ratio = (sd.double() / s3.double() * 2.0 ** (e3 - ed)).float()
        dd = dict(a2=a2, a2_absmax=(a2.abs().max() * 1.4).reshape(1), ds=dict(ph=pd, we=ed, sc=sd, b=bd), ratio=ratio, bias=b3 + bd)
        r = ops.bottleneck_block_chain(y1, ya, c2, c3, c1, dual=dd, x_absmax=xam, y1n_absmax=yam)
        xr = torch.relu(y2r @ w3.double().t() * s3.double() + a2.view(M, 64).double() @ wd.double().t() * sd.double() + (b3 + bd).double())
    else:
        res = torch.relu(rn(B, H, W, N) * 2.0)
        r = ops.bottleneck_block_chain(y1, ya, c2, c3, c1, residual=res, x_absmax=xam, y1n_absmax=yam)
        xr = torch.relu(y2r @ w3.double().t() * s3.double() + b3.double() + res.view(M, N).double())
    assert r is not None
    x, y1n = r
    yr = torch.relu(xr @ w1.double().t() * s1.double() + b1.double())
    assert relerr(x.view(M, N).double().cpu(), xr.cpu()) < 5e-6
    assert relerr(y1n.view(M, P).double().cpu(), yr.cpu()) < 5e-6
    assert xam.item() == x.abs().max().item() and yam.item() == y1n.abs().max().item()
    if not dual:            # the separate launches: halo / per-tap conv2, then the conv3 -> conv1 chain
        y2 = ops.conv_bn_act(y1, w2p, b2, None, 3, 3, 1, 1, ops.ACT_RELU, wl, w_planes_f16=p2, w_exp=e2, x_absmax=ya, out_scale=s2)
        xu = ops.conv_bn_act(y2, w3, b3, res, 1, 1, 1, 0, ops.ACT_RELU, w_planes_f16=p3, w_exp=e3, x_absmax=y2.abs().max().reshape(1), out_scale=s3)
        assert relerr(x.cpu(), xu.cpu()) < 3e-6



@pytest.mark.parametrize("B,H,W,K,N", [(1024, 14, 14, 256, 1024), (671, 14, 14, 256, 1024), (700, 14, 14, 256, 1056), (170, 28, 28, 256, 1024)])
def test_conv1x1_res_stream_kernel(B, H, W, K, N, option):
    """conv3 + BatchNorm + residual + ReLU with a short reduction into many channels (clip/model.py:50-54; layer 3: 256 -> 1024) on
    conv1x1_res_stream_kernel, through the same C-ABI entry as every conv (dbmm_conv_bn_act_x2, option conv1x1_res_stream): against fp64 and
    against the 128 x 128-tile kernel it replaces; same output maximum; ragged last tile (B = 671: 131,516 rows), slab counts that do not
    divide over the workgroups (N = 1056: 33 slabs), guard zones around the output; smaller problems, K = 128 and other epilogues keep the
    tile kernels"""
    g = torch.Generator(device=DEV); g.manual_seed(B + K + N)
    x = torch.relu(torch.randn((B, H, W, K), device=DEV, generator=g))
    res = torch.relu(torch.randn((B, H, W, N), device=DEV, generator=g) * 2.0)
    w = (torch.randn((N, K, 1, 1), device=DEV, generator=g) * K ** -0.5).half().float()
    sc = 0.5 + torch.rand((N,), device=DEV, generator=g); b = torch.randn((N,), device=DEV, generator=g) * 0.1
    wp, wl = ops.pack_conv_weight(w, chunk_major=32)
    ph, we, n = ops.split_planes_f16(wp, allow_single=True)
    assert n == 1
    xam = (x.abs().max() * 1.2).reshape(1)
    kw = dict(w_planes_f16=ph, w_exp=we, x_absmax=xam, out_scale=sc)
    am, am0 = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    y = ops.conv_bn_act(x, wp, b, res, 1, 1, 1, 0, ops.ACT_RELU, wl, y_absmax=am, **kw)
    assert ops._last_igemm_tag() == f"conv1x1_res_stream_kernel<{K}, 0>", ops._last_igemm_tag()
    option("conv1x1_res_stream", 0)
    y0 = ops.conv_bn_act(x, wp, b, res, 1, 1, 1, 0, ops.ACT_RELU, wl, y_absmax=am0, **kw)
    assert not ops._last_igemm_tag().startswith("conv1x1_res_stream"), ops._last_igemm_tag()
    M = B * H * W
    ref = torch.relu(x.view(M, K).double() @ w.view(N, K).double().t() * sc.double() + b.double() + res.view(M, N).double())
    assert relerr(y.view(M, N).double().cpu(), ref.cpu()) < 5e-6 and relerr(y.cpu(), y0.cpu()) < 2e-6
    assert am.item() == y.abs().max().item() and am0.item() == y0.abs().max().item()
    # no residual / fewer rows keep the tile kernels
    ops.conv_bn_act(x, wp, b, None, 1, 1, 1, 0, ops.ACT_RELU, wl, **kw)
    assert not ops._last_igemm_tag().startswith("conv1x1_res_stream")
    ops.conv_bn_act(x[:100].contiguous(), wp, b, res[:100].contiguous(), 1, 1, 1, 0, ops.ACT_RELU, wl, **kw)
    assert not ops._last_igemm_tag().startswith("conv1x1_res_stream")
    option("conv1x1_res_stream", 1)
    guard = 4096
    buf = torch.full((M * N + 2 * guard,), 777.0, device=DEV)
    out = buf[guard:guard + M * N].view(B, H, W, N)
    rc = ops._conv_x2(x, wp, b, res, out, 1, 1, 1, 0, ops.ACT_RELU, wl, ph, we, xam, None, sc, 0, None)
    assert rc == 0 and ops._last_igemm_tag().startswith("conv1x1_res_stream"), (rc, ops._last_igemm_tag())
    torch.cuda.synchronize()
    assert (buf[:guard] == 777.0).all() and (buf[guard + M * N:] == 777.0).all() and torch.equal(out, y)


@pytest.mark.parametrize("B,H,W,N", [(1024, 14, 14, 1024), (673, 14, 14, 1056), (170, 28, 28, 1024), (46, 54, 58, 1024)])
def test_conv1x1_res_stream_kernel_pooled(B, H, W, N, option):
    """the last block of a stage: conv3 + BatchNorm + residual + ReLU writes the un-pooled map AND its AvgPool2d(2) (clip/model.py:36-38,
    50-54) -- conv1x1_res_stream_kernel<256, 1> walks 2x2-window-major tiles; against fp64, against the 128 x 128-tile kernel's dual-output
    epilogue, pooled == avg_pool2d of the written map; windows wrapping over pooled rows and images inside a tile (W / 2 = 7, 14, 29), a ragged
    last tile, guard zones around both outputs"""
    K = 256
    g = torch.Generator(device=DEV); g.manual_seed(B + N + W)
    x = torch.relu(torch.randn((B, H, W, K), device=DEV, generator=g))
    res = torch.relu(torch.randn((B, H, W, N), device=DEV, generator=g) * 2.0)
    w = (torch.randn((N, K, 1, 1), device=DEV, generator=g) * K ** -0.5).half().float()
    sc = 0.5 + torch.rand((N,), device=DEV, generator=g); b = torch.randn((N,), device=DEV, generator=g) * 0.1
    wp, wl = ops.pack_conv_weight(w, chunk_major=32)
    ph, we, n = ops.split_planes_f16(wp, allow_single=True)
    xam = (x.abs().max() * 1.2).reshape(1)
    kw = dict(w_planes_f16=ph, w_exp=we, x_absmax=xam, out_scale=sc)
    am, am0 = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    yp, y = ops.conv_bn_act(x, wp, b, res, 1, 1, 1, 0, ops.ACT_RELU, wl, y_absmax=am, pool=2, keep_full=True, **kw)
    assert ops._last_igemm_tag() == "conv1x1_res_stream_kernel<256, 1>", ops._last_igemm_tag()
    option("conv1x1_res_stream", 0)
    yp0, y0 = ops.conv_bn_act(x, wp, b, res, 1, 1, 1, 0, ops.ACT_RELU, wl, y_absmax=am0, pool=2, keep_full=True, **kw)
    assert not ops._last_igemm_tag().startswith("conv1x1_res_stream")
    M = B * H * W
    ref = torch.relu(x.view(M, K).double() @ w.view(N, K).double().t() * sc.double() + b.double() + res.view(M, N).double()).view(B, H, W, N)
    assert relerr(y.double().cpu(), ref.cpu()) < 5e-6 and relerr(y.cpu(), y0.cpu()) < 2e-6 and relerr(yp.cpu(), yp0.cpu()) < 2e-6
    assert tuple(yp.shape) == (B, H // 2, W // 2, N)
    assert relerr(yp.cpu(), F.avg_pool2d(y.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1).cpu()) < 2e-6
    assert am.item() == y.abs().max().item() and am0.item() == y0.abs().max().item()
    option("conv1x1_res_stream", 1)
    guard = 4096
    bf = torch.full((M * N + 2 * guard,), 777.0, device=DEV); bp = torch.full((M // 4 * N + 2 * guard,), 777.0, device=DEV)
    of, op = bf[guard:guard + M * N].view(B, H, W, N), bp[guard:guard + M // 4 * N].view(B, H // 2, W // 2, N)
    rc = ops._conv_x2(x, wp, b, res, op, 1, 1, 1, 0, ops.ACT_RELU, wl, ph, we, xam, None, sc, 2, of)
    assert rc == 0 and ops._last_igemm_tag() == "conv1x1_res_stream_kernel<256, 1>", (rc, ops._last_igemm_tag())
    torch.cuda.synchronize()
    assert (bf[:guard] == 777.0).all() and (bf[guard + M * N:] == 777.0).all() and (bp[:guard] == 777.0).all() and (bp[guard + M // 4 * N:] == 777.0).all()
    assert torch.equal(of, y) and torch.equal(op, yp)
