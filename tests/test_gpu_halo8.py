"""conv3x3_halo8_kernel (csrc/conv3x3_halo8.hip): the eight-phase 256 x 256 variant of the parity-mode 3x3 conv, through the same
C-ABI entry point as every other conv (dbmm_conv_bn_act_x2, option halo8).  It keeps igemm_halo_kernel's arithmetic, K order and
accumulation order, so the two must agree BIT FOR BIT; both are held to an fp64 reference of
/root/reference/clip/model.py:24-26, 44-45 (conv2 -> bn2 -> relu -> avgpool)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import relerr
from dbmm_amd import ops, synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(seed, name, shape, std=1.0):
    return synth.normal(seed, name, shape, std)


def _case(B, H, W, Cin, Cout, pool, relu, option, check_ref=True, head_rows=None):
    x = rnd(1, "x", (B, Cin, H, W)); w = rnd(2, "w", (Cout, Cin, 3, 3), (Cin * 9) ** -0.5).half().float()
    b = rnd(3, "b", (Cout,), 0.1); sc = 0.5 + synth.uniform(4, "sc", (Cout,))
    act = ops.ACT_RELU if relu else ops.ACT_NONE
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wp, wl = ops.pack_conv_weight(w.to(DEV), chunk_major=32)
    ph, we, n = ops.split_planes_f16(wp, allow_single=True)
    assert n == 1
    kw = dict(w_planes_f16=ph, w_exp=we, x_absmax=xd.abs().max().reshape(1), out_scale=sc.to(DEV))
    am8, am = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    option("igemm_streamk", 0)          # (stream-K cuts the old kernel's K loop into partial sums: another summation order)
    option("halo8", 2)
    o8 = ops.conv_bn_act(xd, wp, b.to(DEV), None, 3, 3, 1, 1, act, wl, y_absmax=am8, pool=2 if pool else 1, **kw)
    tag = ops._last_igemm_tag()
    assert tag == f"conv3x3_halo8{'n' if Cout % 256 else ''}_kernel<{1 if pool else 0}>", tag
    option("halo8", 0)
    o = ops.conv_bn_act(xd, wp, b.to(DEV), None, 3, 3, 1, 1, act, wl, y_absmax=am, pool=2 if pool else 1, **kw)
    assert ops._last_igemm_tag().startswith("igemm_halo_kernel<"), ops._last_igemm_tag()
    assert o8.shape == o.shape
    if head_rows is None:
        assert torch.equal(o8, o), (relerr(o8.cpu(), o.cpu()), (o8 != o).sum().item())
    else:       # tail_split: the tiles past the whole rounds are sums of K slices (another order of summation)
        hr = head_rows // 4 if pool else head_rows
        assert torch.equal(o8.view(-1, Cout)[:hr], o.view(-1, Cout)[:hr]) and relerr(o8.cpu(), o.cpu()) < 2e-6
        assert not torch.equal(o8, o)
    assert am8.item() == o8.abs().max().item() == am.item()
    if check_ref:
        ref = F.conv2d(x.double(), w.double(), None, padding=1) * sc.double().view(1, -1, 1, 1) + b.double().view(1, -1, 1, 1)
        if relu:
            ref = torch.relu(ref)
        if pool:
            ref = F.avg_pool2d(ref, 2)
        assert relerr(o8.cpu().double(), ref.permute(0, 2, 3, 1)) < 5e-6
    return o8


@pytest.mark.parametrize("B,H,W,Cin,Cout,pool,relu", [
    (128, 14, 14, 256, 256, 0, True),        # layer 3's conv2: 98 whole tiles
    (512, 7, 7, 512, 512, 0, True),          # layer 4's: two N tiles per row block, 7 x 7 maps (most fragment rows hold a border pixel)
    (131, 14, 14, 64, 256, 0, False),        # ragged last tile (25,676 rows), two groups per tap row only, no activation
    (30, 26, 30, 128, 256, 0, True),         # non-square maps
    (24, 28, 28, 256, 256, 2, True),         # layer 3's first conv2: pooled, window-major rows
    (100, 14, 14, 512, 512, 2, True),        # layer 4's first conv2
    (33, 26, 26, 64, 256, 2, True),          # pooled, ragged (5,577 pooled rows), windows wrap over pooled rows inside a tile
    (700, 6, 6, 64, 256, 2, False),          # 3 windows per pooled row: every tile wraps over rows and images
    (3, 80, 72, 64, 256, 0, True),           # a tile inside one image row block (W > 64)
    # Cout % 256 != 0: conv3x3_halo8n_kernel (256 x 128 tiles)
    (32, 28, 28, 128, 128, 0, True),         # layer 2's conv2
    (8, 56, 56, 128, 128, 2, True),          # layer 2's first conv2: pooled
    (131, 14, 14, 64, 128, 0, False),        # ragged last tile, one slab pair only, no activation
    (33, 26, 26, 192, 384, 2, True),         # three N tiles, pooled, ragged, windows wrap over pooled rows
    (700, 6, 6, 64, 128, 2, False),          # every tile wraps over rows and images
    (40, 26, 30, 128, 128, 0, True),         # non-square maps
])
def test_halo8_equals_halo_kernel_and_fp64(B, H, W, Cin, Cout, pool, relu, option):
    _case(B, H, W, Cin, Cout, pool, relu, option)


@pytest.mark.parametrize("B,Cin,pool", [(356, 64, 0), (356, 64, 2), (356, 256, 0), (400, 256, 2)])
def test_halo8_tail_split(B, Cin, pool, option):
    """one round of 256 tiles + a short one whose tiles are cut along K over the idle CUs (second launch sums the slices):
    17 tiles x 3 slices of one loop trip; 17 x 12 slices; 51 tiles x 5 slices of 2 / 3 trips.  tail_split = 0: one launch, bit-equal again."""
    _case(B, 14, 14, Cin, 256, pool, True, option, head_rows=65536)
    option("tail_split", 0)
    _case(B, 14, 14, Cin, 256, pool, True, option, check_ref=False)


def test_halo8_shapes_it_does_not_take_fall_back(option):
    """Cout % 128 != 0, Cin % 64 != 0, a residual, or a small problem: the library keeps igemm_halo_kernel (no error, same results)."""
    option("halo8", 2)
    for B, H, Cin, Cout, res in [(64, 14, 256, 64, False), (64, 14, 96, 256, False), (100, 14, 64, 256, True), (8, 14, 64, 256, False)]:
        x = rnd(1, "x", (B, Cin, H, H)); w = rnd(2, "w", (Cout, Cin, 3, 3), (Cin * 9) ** -0.5).half().float()
        xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
        r = rnd(5, "r", (B, H, H, Cout)).to(DEV) if res else None
        wp, wl = ops.pack_conv_weight(w.to(DEV), chunk_major=32)
        ph, we, n = ops.split_planes_f16(wp, allow_single=True)
        o = ops.conv_bn_act(xd, wp, None, r, 3, 3, 1, 1, ops.ACT_RELU, wl, w_planes_f16=ph, w_exp=we, x_absmax=xd.abs().max().reshape(1))
        assert not ops._last_igemm_tag().startswith("conv3x3_halo8"), ops._last_igemm_tag()
        ref = F.conv2d(x.double(), w.double(), None, padding=1).permute(0, 2, 3, 1)
        if res:
            ref = ref + r.cpu().double()
        assert relerr(o.cpu().double(), torch.relu(ref)) < 5e-6


def test_halo8_guard_zones(option):
    """the kernel writes exactly its output: canaries before and after the (ragged) output tensor survive"""
    B, H, Cin, Cout = 131, 14, 64, 256
    x = rnd(1, "x", (B, Cin, H, H)); w = rnd(2, "w", (Cout, Cin, 3, 3), (Cin * 9) ** -0.5).half().float()
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wp, wl = ops.pack_conv_weight(w.to(DEV), chunk_major=32)
    ph, we, n = ops.split_planes_f16(wp, allow_single=True)
    option("halo8", 2)
    for pool in (0, 2):
        Ho = H // 2 if pool else H
        n_out = B * Ho * Ho * Cout
        guard = 4096
        buf = torch.full((n_out + 2 * guard,), 777.0, device=DEV)
        out = buf[guard:guard + n_out].view(B, Ho, Ho, Cout)
        am = torch.zeros(1, device=DEV)
        rc = ops._conv_x2(xd, wp, None, None, out, 3, 3, 1, 1, ops.ACT_RELU, wl, ph, we, xd.abs().max().reshape(1), am, None, pool, None)
        assert rc == 0 and ops._last_igemm_tag().startswith("conv3x3_halo8"), (rc, ops._last_igemm_tag())
        torch.cuda.synchronize()
        assert (buf[:guard] == 777.0).all() and (buf[guard + n_out:] == 777.0).all()
        assert (out != 777.0).any() and am.item() == out.abs().max().item()


def test_halo8_on_operands_over_2gib(option):
    """layer 3's first conv2 at a batch whose input passes 2 GiB (802,816 B per image: from B = 2675): every tile rebases its descriptors, so
    the first image, the images around the 2 GiB line and the last one must match fp64 -- plain (output > 2 GiB too) and pooled"""
    B, H, C = 2800, 28, 256
    g = torch.Generator(device=DEV); g.manual_seed(17)
    x = torch.randn((B, H, H, C), device=DEV, generator=g)
    assert x.numel() * 4 > 2 ** 31
    w = (torch.randn((C, C, 3, 3), device=DEV, generator=g) * (9 * C) ** -0.5).half().float()
    sc = 0.5 + torch.rand((C,), device=DEV, generator=g); b = torch.randn((C,), device=DEV, generator=g) * 0.1
    wp, wl = ops.pack_conv_weight(w, chunk_major=32)
    ph, we, n = ops.split_planes_f16(wp, allow_single=True)
    option("halo8", 2)
    kw = dict(w_planes_f16=ph, w_exp=we, x_absmax=x.abs().max().reshape(1), out_scale=sc)
    am = torch.zeros(1, device=DEV)
    y = ops.conv_bn_act(x, wp, b, None, 3, 3, 1, 1, ops.ACT_RELU, wl, y_absmax=am, **kw)
    assert ops._last_igemm_tag() == "conv3x3_halo8_kernel<0>" and y.numel() * 4 > 2 ** 31
    yp = ops.conv_bn_act(x, wp, b, None, 3, 3, 1, 1, ops.ACT_RELU, wl, pool=2, **kw)
    assert ops._last_igemm_tag() == "conv3x3_halo8_kernel<1>"
    s = 2 ** 31 // (H * H * C * 4)
    for i in (0, s - 1, s, s + 1, B - 1):
        ref = F.conv2d(x[i].permute(2, 0, 1)[None].double().cpu(), w.double().cpu(), None, padding=1)[0]
        ref = torch.relu(ref * sc.double().cpu().view(-1, 1, 1) + b.double().cpu().view(-1, 1, 1))
        assert relerr(y[i].double().cpu(), ref.permute(1, 2, 0)) < 5e-6, i
        assert relerr(yp[i].double().cpu(), F.avg_pool2d(ref[None], 2)[0].permute(1, 2, 0)) < 5e-6, i
    assert am.item() == y.abs().max().item()
