"""Per-kernel parity: every C-ABI entry point of liblang2seg_hip.so against the CPU oracle
(oracle/*.py, torch-CPU fp32 autograd for float ops, numpy for integer/box work) on seeded
inputs.  Tolerances: fp32 kernels 1e-4 relative (exact-f32 MFMA), bf16 storage 2e-2;
integer / index outputs bit-exact."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import boxes as OB
from oracle import net as ON

DEV = 'cuda'


def ops():
    from lang2seg_amd import ops as O
    O.TRACK_PLAN = True            # ops.LAST_PLAN: the plan name of the last convolution launch (off outside tests / bench timers)
    return O


def rel_err(a, b):
    a = a.double().cpu(); b = b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def nhwc(x):  # NCHW cpu -> NHWC flat rows
    return x.permute(0, 2, 3, 1).contiguous()


def to_dev(t, dt):
    return t.to(DEV).to(torch.bfloat16 if dt == 1 else torch.float32).contiguous()


def ohwi(w):  # OIHW -> [O][KH][KW][I]
    return w.permute(0, 2, 3, 1).contiguous()


# outputs stored in bf16 carry half an ulp (2^-9 of the value): 5e-3 of the tensor's maximum bounds it; the oracle sees the same rounded operands
TOL = {0: 2e-5, 1: 5e-3}


@pytest.mark.parametrize('dt', [0, 1])
@pytest.mark.parametrize('cfg', [
    dict(n=1, H=19, W=23, Cin=64, Cout=64, k=3, s=1, p=1),
    dict(n=1, H=38, W=63, Cin=256, Cout=256, k=3, s=1, p=1),
    dict(n=3, H=7, W=7, Cin=128, Cout=192, k=3, s=1, p=1),
    dict(n=1, H=38, W=63, Cin=1024, Cout=72, k=1, s=1, p=0),
    dict(n=1, H=20, W=26, Cin=256, Cout=128, k=1, s=2, p=0),
    dict(n=40, H=7, W=7, Cin=512, Cout=512, k=3, s=1, p=1, tile=128),
    dict(n=21, H=7, W=7, Cin=128, Cout=200, k=3, s=1, p=1, tile=256),
    dict(n=23, H=7, W=7, Cin=128, Cout=136, k=3, s=1, p=1, tile=224),
    dict(n=1, H=9, W=11, Cin=96, Cout=40, k=1, s=1, p=0),
    dict(n=256, H=7, W=7, Cin=512, Cout=512, k=3, s=1, p=1),          # the dominant launch: layer4 @ 256 RoIs (M=12544, K=4608, 224x128 tile, tap-inner walk)
    dict(n=1, H=38, W=63, Cin=1024, Cout=512, k=3, s=1, p=1),         # RPN 3x3
    dict(n=256, H=7, W=7, Cin=1024, Cout=2048, k=1, s=1, p=0),        # layer4.0 downsample @ RoIs
    dict(n=1, H=75, W=125, Cin=128, Cout=128, k=3, s=1, p=1),         # layer2 3x3 (wide rows)
    dict(n=1, H=5, W=3, Cin=64, Cout=36, k=3, s=1, p=1),              # a map smaller than one pixel tile, ragged channel tile
    dict(n=2, H=19, W=23, Cin=64, Cout=64, k=3, s=1, p=1),            # two images
])
def test_conv_fwd(cfg, dt):
    O = ops()
    g = torch.Generator().manual_seed(1)
    n, H, W, Cin, Cout, k, s, p = [cfg[x] for x in ['n', 'H', 'W', 'Cin', 'Cout', 'k', 's', 'p']]
    x = torch.randn(n, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g)
    OH = (H + 2 * p - k) // s + 1; OW = (W + 2 * p - k) // s + 1
    res = torch.randn(n, Cout, OH, OW, generator=g)
    xd, wd, rd = to_dev(nhwc(x), dt), to_dev(ohwi(w), dt), to_dev(nhwc(res), dt)
    # the oracle sees the same rounded operands
    xr, wr, rr = xd.float().cpu().permute(0, 3, 1, 2), wd.float().cpu().permute(0, 3, 1, 2), rd.float().cpu().permute(0, 3, 1, 2)
    ref = F.relu(F.conv2d(xr, wr, b, stride=s, padding=p) + rr)
    y = O.empty((n * OH * OW, Cout), dt)
    O.conv_igemm(xd, wd, y, n, H, W, Cin, OH, OW, Cout, k, k, s, p, bias=b.to(DEV), add=rd, relu=True, tile=cfg.get('tile', 0))
    torch.cuda.synchronize()
    assert rel_err(y.float().view(n, OH, OW, Cout), nhwc(ref)) < TOL[dt]
    # split-K path (fp32 partial tiles in workspace slabs, added in slab order by a second launch that applies the same epilogue); the
    # workspace may hold anything, and two runs agree bit for bit (no atomics)
    if Cout % 4 == 0:
        ws = torch.full((3 * n * OH * OW * Cout,), 3e30, dtype=torch.float32, device=DEV)
        y3 = O.empty((n * OH * OW, Cout), dt)
        O.conv_igemm(xd, wd, y3, n, H, W, Cin, OH, OW, Cout, k, k, s, p, bias=b.to(DEV), add=rd, relu=True, tile=64, ws=ws, split_k=3)
        torch.cuda.synchronize()
        assert rel_err(y3.float().view(n, OH, OW, Cout), nhwc(ref)) < TOL[dt]
        y4 = O.empty((n * OH * OW, Cout), dt)
        O.conv_igemm(xd, wd, y4, n, H, W, Cin, OH, OW, Cout, k, k, s, p, bias=b.to(DEV), add=rd, relu=True, tile=64, ws=ws, split_k=3)
        torch.cuda.synchronize()
        assert torch.equal(y3, y4)
    # fp32 output + no epilogue
    y2 = torch.empty((n * OH * OW, Cout), dtype=torch.float32, device=DEV)
    O.conv_igemm(xd, wd, y2, n, H, W, Cin, OH, OW, Cout, k, k, s, p, out_f32=True)
    ref2 = F.conv2d(xr, wr, None, stride=s, padding=p)
    assert rel_err(y2.view(n, OH, OW, Cout), nhwc(ref2)) < (2e-5 if dt == 0 else 1e-4)


@pytest.mark.parametrize('H,W', [(150, 250), (13, 31), (6, 25), (41, 57)])
@pytest.mark.parametrize('down', [True, False])
def test_bottleneck64_fused(H, W, down):
    """csrc/bottleneck_fused.hip: a frozen 64-plane bottleneck behind its conv1 as one launch (3x3 -> 1x1 + shortcut -> ReLU, + the next block's
    conv1), bf16, against torch-CPU on the same rounded operands with the same rounding points (b and y rounded to bf16 where the kernel stores
    them): the 150x250 map of the 600x1000 step (no ragged tile), a map smaller than one tile row, exactly one tile, and a ragged 41x57 map; first
    block (Cx = 64, shortcut convolution) and later blocks (Cx = 256, identity).  5e-3 of the tensor's maximum = one bf16 ulp at that scale."""
    O = ops()
    g = torch.Generator().manual_seed(7 + H + int(down))
    Cx = 64 if down else 256
    rb = lambda t: t.bfloat16().float()
    a = rb(torch.randn(1, 64, H, W, generator=g).clamp(min=0)); x = rb(torch.randn(1, Cx, H, W, generator=g).clamp(min=0))
    w2 = rb(torch.randn(64, 64, 3, 3, generator=g) / np.sqrt(576)); w3 = rb(torch.randn(256, 64, 1, 1, generator=g) / 8)
    wd_ = rb(torch.randn(256, 64, 1, 1, generator=g) / 8); w1n = rb(torch.randn(64, 256, 1, 1, generator=g) / 16)
    b2, b3, bd, b1n = [torch.randn(n, generator=g) * 0.2 for n in (64, 256, 256, 64)]
    b = rb(F.relu(F.conv2d(a, w2, b2, padding=1)))
    sc = F.conv2d(x, wd_, bd) if down else x
    yr = rb(F.relu(F.conv2d(b, w3, b3) + sc))
    anr = rb(F.relu(F.conv2d(yr, w1n, b1n)))
    dev = lambda t: to_dev(nhwc(t), 1).view(-1, t.shape[1]).contiguous()
    dw = lambda t: to_dev(ohwi(t), 1).contiguous()
    f = lambda t: t.float().to(DEV).contiguous()
    ad, xd = dev(a), dev(x)
    for with_next in (True, False):
        y = torch.full((H * W, 256), float('nan'), device=DEV).bfloat16(); an = torch.full((H * W, 64), float('nan'), device=DEV).bfloat16()
        O.bottleneck64_fwd(ad, xd, dw(w2), f(b2), dw(w3), f(b3), y, H, W, wd=dw(wd_) if down else None, bd=f(bd) if down else None,
                           w1n=dw(w1n) if with_next else None, b1n=f(b1n) if with_next else None, a_next=an if with_next else None)
        torch.cuda.synchronize()
        assert rel_err(y.float().view(1, H, W, 256), nhwc(yr)) < 5e-3, (H, W, down, rel_err(y.float().view(1, H, W, 256), nhwc(yr)))
        if with_next:
            assert rel_err(an.float().view(1, H, W, 64), nhwc(anr)) < 5e-3, (H, W, down, rel_err(an.float().view(1, H, W, 64), nhwc(anr)))
        else:
            assert bool(torch.isnan(an.float()).all())


@pytest.mark.parametrize('cfg', [
    dict(n=256, H=7, W=7, Cin=512, Cout=512, k=3, s=1, p=1),          # the dominant launch (M = 49 x 256, no ragged tile)
    dict(n=23, H=7, W=7, Cin=128, Cout=200, k=3, s=1, p=1),           # ragged pixel tile (M = 1127) and ragged channel tile, borders of 7x7 maps
    dict(n=64, H=7, W=7, Cin=512, Cout=2048, k=1, s=1, p=0),          # layer4 1x1-out: 8 slices of K
    dict(n=40, H=7, W=7, Cin=192, Cout=128, k=1, s=1, p=0),           # exactly 3 slices of K: prologue and tail only
    dict(n=1, H=38, W=63, Cin=1024, Cout=512, k=3, s=1, p=1),         # RPN 3x3 on the feature map: image borders, 144 slices
    dict(n=1, H=20, W=26, Cin=256, Cout=128, k=1, s=2, p=0),          # strided 1x1
    dict(n=2, H=19, W=23, Cin=64, Cout=64, k=3, s=1, p=1),            # two small images, 9 slices
])
@pytest.mark.parametrize('ALGO_DMA', [2])
def test_conv_dma_tile(cfg, ALGO_DMA):
    """L2S_ALGO_DMA (256x128 tile filled by buffer_load ... lds, two wave groups alternating load / multiply) against torch on the same
    rounded bf16 operands: bias + residual + ReLU epilogue, and the data-gradient form (ReLU mask operand, no bias)."""
    O = ops()
    dt = 1
    g = torch.Generator().manual_seed(11)
    n, H, W, Cin, Cout, k, s, p = [cfg[x] for x in ['n', 'H', 'W', 'Cin', 'Cout', 'k', 's', 'p']]
    x = torch.randn(n, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g)
    OH = (H + 2 * p - k) // s + 1; OW = (W + 2 * p - k) // s + 1
    res = torch.randn(n, Cout, OH, OW, generator=g)
    xd, wd, rd = to_dev(nhwc(x), dt), to_dev(ohwi(w), dt), to_dev(nhwc(res), dt)
    xr, wr, rr = xd.float().cpu().permute(0, 3, 1, 2), wd.float().cpu().permute(0, 3, 1, 2), rd.float().cpu().permute(0, 3, 1, 2)
    conv = F.conv2d(xr, wr, None, stride=s, padding=p)
    y = torch.full((n * OH * OW, Cout), float('nan'), dtype=torch.bfloat16, device=DEV)
    O.conv_igemm(xd, wd, y, n, H, W, Cin, OH, OW, Cout, k, k, s, p, bias=b.to(DEV), add=rd, relu=True, algo=ALGO_DMA)
    torch.cuda.synchronize()
    assert rel_err(y.float().view(n, OH, OW, Cout), nhwc(F.relu(conv + b.view(1, -1, 1, 1) + rr))) < TOL[dt]
    y2 = torch.full((n * OH * OW, Cout), float('nan'), dtype=torch.bfloat16, device=DEV)
    O.conv_igemm(xd, wd, y2, n, H, W, Cin, OH, OW, Cout, k, k, s, p, ref=rd, algo=ALGO_DMA)
    torch.cuda.synchronize()
    assert rel_err(y2.float().view(n, OH, OW, Cout), nhwc(conv * (rr > 0))) < TOL[dt]
    # the same launch twice more: no dependence on what the LDS ring held before
    y3 = torch.empty_like(y2)
    for _ in range(2):
        O.conv_igemm(xd, wd, y3, n, H, W, Cin, OH, OW, Cout, k, k, s, p, ref=rd, algo=ALGO_DMA)
    torch.cuda.synchronize()
    assert torch.equal(y2, y3)


@pytest.mark.parametrize('cfg', [
    dict(n=256, H=7, W=7, Cin=512, Cout=2048),     # layer4 1x1-out on the RoIs: 784 tiles of 8 slices on 256 persistent workgroups (3 - 4 tiles each)
    dict(n=256, H=7, W=7, Cin=1024, Cout=2048),    # layer4[0].downsample: 16 slices
    dict(n=37, H=7, W=7, Cin=192, Cout=392),       # ragged pixel tile (M = 1813), ragged channel tile, 3 slices: 32 tiles on 32 workgroups (one each)
    dict(n=75, H=7, W=7, Cin=256, Cout=1048),      # 15 x 9 = 135 tiles on 136 workgroups: one workgroup without a tile
    dict(n=5, H=3, W=3, Cin=192, Cout=64),         # a single tile
])
def test_conv_persistent_dma_tile(cfg):
    """L2S_ALGO_PDMA (the LDS-DMA 256x128 tile as one persistent workgroup per CU that walks its tiles, the requests running ahead
    across tile boundaries, per-wave epilogues) against torch on the same rounded bf16 operands and BIT FOR BIT against the one-shot
    LDS-DMA tile (same slices in the same order): bias + residual + ReLU epilogue, and the data-gradient form (ReLU mask, no bias)."""
    O = ops()
    dt = 1
    g = torch.Generator().manual_seed(12)
    n, H, W, Cin, Cout = [cfg[x] for x in ['n', 'H', 'W', 'Cin', 'Cout']]
    x = torch.randn(n, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / np.sqrt(Cin)
    b = torch.randn(Cout, generator=g)
    res = torch.randn(n, Cout, H, W, generator=g)
    xd, wd, rd = to_dev(nhwc(x), dt), to_dev(ohwi(w), dt), to_dev(nhwc(res), dt)
    xr, wr, rr = xd.float().cpu().permute(0, 3, 1, 2), wd.float().cpu().permute(0, 3, 1, 2), rd.float().cpu().permute(0, 3, 1, 2)
    conv = F.conv2d(xr, wr, None)
    M = n * H * W
    ys = {}
    for algo in (7, 2):
        y = torch.full((M, Cout), float('nan'), dtype=torch.bfloat16, device=DEV)
        O.conv_igemm(xd, wd, y, n, H, W, Cin, H, W, Cout, 1, 1, 1, 0, bias=b.to(DEV), add=rd, relu=True, algo=algo)
        y2 = torch.full((M, Cout), float('nan'), dtype=torch.bfloat16, device=DEV)
        O.conv_igemm(xd, wd, y2, n, H, W, Cin, H, W, Cout, 1, 1, 1, 0, ref=rd, algo=algo)
        torch.cuda.synchronize()
        ys[algo] = (y, y2)
    y, y2 = ys[7]
    assert rel_err(y.float().view(n, H, W, Cout), nhwc(F.relu(conv + b.view(1, -1, 1, 1) + rr))) < TOL[dt]
    assert rel_err(y2.float().view(n, H, W, Cout), nhwc(conv * (rr > 0))) < TOL[dt]
    assert torch.equal(y, ys[2][0]) and torch.equal(y2, ys[2][1])
    y3 = torch.empty_like(y2)
    for _ in range(3):
        O.conv_igemm(xd, wd, y3, n, H, W, Cin, H, W, Cout, 1, 1, 1, 0, ref=rd, algo=7)
    torch.cuda.synchronize()
    assert torch.equal(y2, y3)


@pytest.mark.parametrize('cfg', [
    dict(n=256, H=7, W=7, Cin=512, Cout=512, k=3),      # the dominant launch: 64 x 4 = 256 tiles of 196 rows, no ragged tile
    dict(n=91, H=7, W=7, Cin=128, Cout=256, k=3),       # M = 4459 = 22 x 196 + 147: ragged last tile
    dict(n=37, H=7, W=7, Cin=192, Cout=200, k=3),       # ragged rows AND a ragged channel tile, three slices per tap
    dict(n=5, H=7, W=7, Cin=64, Cout=128, k=3),         # M = 245: two tiles, the second nearly empty
    dict(n=256, H=7, W=7, Cin=1024, Cout=512, k=1),     # layer4[0].conv1 on the RoIs (1x1)
    dict(n=3, H=9, W=11, Cin=128, Cout=136, k=3),       # maps that are not 7x7: rows of a tile cross images at arbitrary places
])
def test_conv_dma196_tile(cfg):
    """igemm_dma196_kernel (the 256x128 LDS-DMA pipeline on tiles of 196 rows: wave row 0 keeps four row fragments, wave rows 1-3 three; rows
    196-207 of a tile are computed and not stored) against torch on the same rounded bf16 operands and BIT FOR BIT against the 256-row tile
    (same k order inside every accumulator): bias + residual + ReLU, the data-gradient forms (ReLU mask; residual + mask), plain.  The tile is
    on request only (l2s_conv_desc.algo = L2S_ALGO_DMA196): 2-4 % faster alone, 1 % slower inside the step."""
    O = ops()
    dt = 1
    g = torch.Generator().manual_seed(17)
    n, H, W, Cin, Cout, k = [cfg[x] for x in ['n', 'H', 'W', 'Cin', 'Cout', 'k']]
    pd = k // 2
    x = torch.randn(n, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g)
    xd, wd = to_dev(nhwc(x), dt), to_dev(ohwi(w), dt)
    rd = to_dev(nhwc(torch.randn(n, Cout, H, W, generator=g)), dt); r2d = to_dev(nhwc(torch.randn(n, Cout, H, W, generator=g)), dt)
    xr, wr = xd.float().cpu().permute(0, 3, 1, 2), wd.float().cpu().permute(0, 3, 1, 2)
    rr, r2 = rd.float().cpu().permute(0, 3, 1, 2), r2d.float().cpu().permute(0, 3, 1, 2)
    conv = F.conv2d(xr, wr, None, padding=pd)
    M = n * H * W
    ys = {}
    for algo in (10, 2):
        outs = []
        for kw in (dict(bias=b.to(DEV), add=rd, relu=True), dict(ref=rd), dict(bias=b.to(DEV)), dict(add=rd, ref=r2d)):
            y = torch.full((M, Cout), float('nan'), dtype=torch.bfloat16, device=DEV)
            O.conv_igemm(xd, wd, y, n, H, W, Cin, H, W, Cout, k, k, 1, pd, algo=algo, **kw)
            assert ('dma196' in O.LAST_PLAN) == (algo == 10), O.LAST_PLAN
            outs.append(y)
        torch.cuda.synchronize()
        ys[algo] = outs
    y, y2, y3, y5 = ys[10]
    assert rel_err(y.float().view(n, H, W, Cout), nhwc(F.relu(conv + b.view(1, -1, 1, 1) + rr))) < TOL[dt]
    assert rel_err(y2.float().view(n, H, W, Cout), nhwc(conv * (rr > 0))) < TOL[dt]
    assert rel_err(y3.float().view(n, H, W, Cout), nhwc(conv + b.view(1, -1, 1, 1))) < TOL[dt]
    assert rel_err(y5.float().view(n, H, W, Cout), nhwc((conv + rr) * (r2 > 0))) < TOL[dt]
    for a, c in zip(ys[10], ys[2]):
        assert torch.equal(a, c)


@pytest.mark.parametrize('cfg', [
    dict(n=256, H=7, W=7, Cin=512, Cout=2048),     # layer4 conv3 on the RoIs: 392 tiles of 16 slices
    dict(n=256, H=7, W=7, Cin=1024, Cout=2048),    # layer4[0].downsample
    dict(n=256, H=7, W=7, Cin=2048, Cout=1024),    # the downsample's data gradient
    dict(n=91, H=7, W=7, Cin=192, Cout=1024),      # ragged pixel tile (M = 4459 = 17 x 256 + 107), six slices
    dict(n=100, H=7, W=7, Cin=320, Cout=1280),     # five channel tiles, ten slices
])
def test_conv_dma256_tile(cfg):
    """igemm_dma256_kernel (256x256 LDS-DMA tile, 32-channel slices, per-wave epilogue through LDS; the automatic choice for wide plain
    GEMMs) against torch on the same rounded bf16 operands and BIT FOR BIT against the 256x128 LDS-DMA tile (same k order inside every
    accumulator): bias + residual + ReLU epilogue, the data-gradient forms (ReLU mask; residual + mask), plain; the automatic plan picks it."""
    O = ops()
    dt = 1
    g = torch.Generator().manual_seed(13)
    n, H, W, Cin, Cout = [cfg[x] for x in ['n', 'H', 'W', 'Cin', 'Cout']]
    x = torch.randn(n, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / np.sqrt(Cin)
    b = torch.randn(Cout, generator=g)
    res = torch.randn(n, Cout, H, W, generator=g)
    xd, wd, rd = to_dev(nhwc(x), dt), to_dev(ohwi(w), dt), to_dev(nhwc(res), dt)
    r2d = to_dev(nhwc(torch.randn(n, Cout, H, W, generator=g)), dt)
    xr, wr, rr = xd.float().cpu().permute(0, 3, 1, 2), wd.float().cpu().permute(0, 3, 1, 2), rd.float().cpu().permute(0, 3, 1, 2)
    r2 = r2d.float().cpu().permute(0, 3, 1, 2)
    conv = F.conv2d(xr, wr, None)
    M = n * H * W
    ys = {}
    for algo in (9, 2, 0):
        outs = []
        for kw in (dict(bias=b.to(DEV), add=rd, relu=True), dict(ref=rd), dict(bias=b.to(DEV)), dict(add=rd, ref=r2d)):
            y = torch.full((M, Cout), float('nan'), dtype=torch.bfloat16, device=DEV)
            O.conv_igemm(xd, wd, y, n, H, W, Cin, H, W, Cout, 1, 1, 1, 0, algo=algo, **kw)
            if algo == 9 or (algo == 0 and M >= 4096):
                assert 'dma256' in O.LAST_PLAN, O.LAST_PLAN
            outs.append(y)
        torch.cuda.synchronize()
        ys[algo] = outs
    y, y2, y3, y5 = ys[9]
    assert rel_err(y5.float().view(n, H, W, Cout), nhwc((conv + rr) * (r2 > 0))) < TOL[dt]      # residual AND mask: a block's first 1x1, data gradient
    assert rel_err(y.float().view(n, H, W, Cout), nhwc(F.relu(conv + b.view(1, -1, 1, 1) + rr))) < TOL[dt]
    assert rel_err(y2.float().view(n, H, W, Cout), nhwc(conv * (rr > 0))) < TOL[dt]
    assert rel_err(y3.float().view(n, H, W, Cout), nhwc(conv + b.view(1, -1, 1, 1))) < TOL[dt]
    for a, c in zip(ys[9], ys[2]):
        assert torch.equal(a, c)
    for a, c in zip(ys[9], ys[0]):
        assert torch.equal(a, c)
    y4 = torch.empty_like(y2)
    for _ in range(3):
        O.conv_igemm(xd, wd, y4, n, H, W, Cin, H, W, Cout, 1, 1, 1, 0, ref=rd, algo=9)
    torch.cuda.synchronize()
    assert torch.equal(y2, y4)


@pytest.mark.parametrize('cfg', [dict(H=38, W=63, C=1024, R=256, N1=512, N2=2048), dict(H=20, W=26, C=1024, R=37, N1=512, N2=2048),
                                 dict(H=10, W=14, C=256, R=5, N1=128, N2=256)])
def test_roialign_fused_into_layer4_block0(cfg):
    """l2s_roialign_block0_fwd (crop-and-resize + layer4[0].conv1 + layer4[0].downsample in one launch, one workgroup per RoI) against
    (a) the oracle's _crop_pool_layer restatement followed by the two 1x1 convolutions in fp32 on the same rounded operands, and
    (b) the three launches it replaces (l2s_roialign_fwd + two l2s_conv_igemm), bit for bit: same crop arithmetic (roi_sample.h), same
    order of the K slices.  RoIs reach over the map's borders (zero padding) and include a degenerate one."""
    O = ops()
    H, W, C, R, N1, N2 = [cfg[k] for k in ('H', 'W', 'C', 'R', 'N1', 'N2')]
    P = 7
    g = torch.Generator().manual_seed(21)
    feat = torch.randn(H * W, C, generator=g)
    rs = np.random.RandomState(3)
    x1 = rs.uniform(-40, W * 16 - 60, R); y1 = rs.uniform(-40, H * 16 - 60, R)
    rois = np.stack([np.zeros(R), x1, y1, x1 + rs.uniform(8, 400, R), y1 + rs.uniform(8, 300, R)], 1).astype(np.float32)
    rois[0] = [0, 5.0, 7.0, 5.0, 7.0]                                  # a point
    w1 = torch.randn(N1, C, generator=g) / np.sqrt(C); w2 = torch.randn(N2, C, generator=g) / np.sqrt(C)
    b1 = torch.randn(N1, generator=g); b2 = torch.randn(N2, generator=g)
    fd, w1d, w2d = to_dev(feat, 1), to_dev(w1, 1), to_dev(w2, 1)
    rd = torch.from_numpy(rois).to(DEV)
    pooled = torch.full((R * P * P, C), float('nan'), dtype=torch.bfloat16, device=DEV)
    y1_ = torch.full((R * P * P, N1), float('nan'), dtype=torch.bfloat16, device=DEV)
    y2_ = torch.full((R * P * P, N2), float('nan'), dtype=torch.bfloat16, device=DEV)
    O.roialign_block0_fwd(fd, H, W, C, rd, R, P, 1.0 / 16.0, w1d, b1.to(DEV), N1, w2d, b2.to(DEV), N2, pooled, y1_, y2_)
    torch.cuda.synchronize()
    # (b) the unfused launches
    crop = torch.empty_like(pooled); u1 = torch.empty_like(y1_); u2 = torch.empty_like(y2_)
    O.roialign_fwd(fd, H, W, C, rd, R, P, 1.0 / 16.0, crop)
    O.conv_igemm(crop, w1d, u1, R, P, P, C, P, P, N1, bias=b1.to(DEV), relu=True)
    O.conv_igemm(crop, w2d, u2, R, P, P, C, P, P, N2, bias=b2.to(DEV))
    torch.cuda.synchronize()
    assert torch.equal(pooled, crop)
    assert torch.equal(y1_, u1) and torch.equal(y2_, u2)
    # (a) the oracle's crop on the rounded map, then fp32 products
    net = ON.OracleNet.__new__(ON.OracleNet); net.cfg = ON.DEFAULT_CFG; net.var = {}
    ref_crop = net.crop_pool(fd.float().cpu().view(1, H, W, C).permute(0, 3, 1, 2), torch.from_numpy(rois)).permute(0, 2, 3, 1).reshape(R * P * P, C)
    assert rel_err(pooled.float(), ref_crop) < 1e-2
    pc = pooled.float().cpu()
    assert rel_err(y1_.float(), F.relu(pc @ w1d.float().cpu().t() + b1)) < TOL[1]
    assert rel_err(y2_.float(), pc @ w2d.float().cpu().t() + b2) < TOL[1]


@pytest.mark.parametrize('cfg', [
    dict(H=19, W=32, Cin=512, Cout=512, k=3, p=1, split=4),           # 80 tiles x 72 slices in 4 slabs
    dict(H=19, W=32, Cin=2048, Cout=512, k=1, p=0, split=3),          # ragged slice ranges (32 slices in 3 slabs)
    dict(H=38, W=63, Cin=256, Cout=256, k=3, p=1, split=2),
    dict(H=6, W=5, Cin=1024, Cout=128, k=3, p=1, split=8),            # one pixel tile, 144 slices; ragged rows
])
def test_conv_splitk_slabs(cfg):
    """l2s_conv_desc.split_k (bf16, 64x64 wave-specialised tile): K cut over several workgroups, partial tiles in workspace slabs, added in
    slab order by a second launch that applies the epilogue.  Forward form (bias + residual + ReLU) and data-gradient form (add + ReLU mask)
    against fp64 convolution of the same rounded operands, equal to the unsplit launch within bf16 rounding, bit-identical from run to run,
    independent of what the workspace held; a workspace too small for the requested split lowers it.  (Never chosen automatically: the
    second launch costs more than the split saves on every shape measured)"""
    O = ops()
    g = torch.Generator().manual_seed(11)
    H, W, Cin, Cout, k, p = [cfg[x] for x in ['H', 'W', 'Cin', 'Cout', 'k', 'p']]
    M = H * W
    x = torch.randn(1, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g)
    res = torch.randn(1, Cout, H, W, generator=g)
    xd, wd, rd = to_dev(nhwc(x), 1), to_dev(ohwi(w), 1), to_dev(nhwc(res), 1)
    xr, wr, rr = [t.double().cpu().permute(0, 3, 1, 2) for t in (xd, wd, rd)]
    conv = F.conv2d(xr, wr, None, padding=p)
    ws = torch.full((cfg['split'] * M * Cout + 64,), float('nan'), dtype=torch.float32, device=DEV)
    for form in ('fwd', 'dgrad'):
        kw = dict(bias=b.to(DEV), add=rd, relu=True) if form == 'fwd' else dict(add=rd, ref=rd)
        ref = F.relu(conv + b.double().view(1, -1, 1, 1) + rr) if form == 'fwd' else (conv + rr) * (rr > 0)
        ya, yb, y1 = [O.empty((M, Cout), 1) for _ in range(3)]
        O.conv_igemm(xd, wd, ya, 1, H, W, Cin, H, W, Cout, k, k, 1, p, ws=ws, split_k=cfg['split'], tile=64, **kw)
        assert 'splitk' in O.LAST_PLAN, O.LAST_PLAN
        ws.fill_(-7e29)
        O.conv_igemm(xd, wd, yb, 1, H, W, Cin, H, W, Cout, k, k, 1, p, ws=ws, split_k=cfg['split'], tile=64, **kw)
        O.conv_igemm(xd, wd, y1, 1, H, W, Cin, H, W, Cout, k, k, 1, p, ws=ws, **kw)
        assert 'splitk' not in O.LAST_PLAN                         # a workspace alone never splits
        torch.cuda.synchronize()
        assert torch.equal(ya, yb)
        assert rel_err(ya.float().view(1, H, W, Cout), nhwc(ref.float())) < TOL[1], form
        assert rel_err(ya.float(), y1.float()) < 1e-2
    small = torch.empty(M * Cout * 2, dtype=torch.float32, device=DEV)
    O.conv_igemm(xd, wd, ya, 1, H, W, Cin, H, W, Cout, k, k, 1, p, ws=small, split_k=cfg['split'], tile=64, bias=b.to(DEV), add=rd, relu=True)
    torch.cuda.synchronize()
    assert rel_err(ya.float().view(1, H, W, Cout), nhwc(F.relu(conv + b.double().view(1, -1, 1, 1) + rr).float())) < TOL[1]
    tiny = torch.empty(M * Cout, dtype=torch.float32, device=DEV)
    O.conv_igemm(xd, wd, ya, 1, H, W, Cin, H, W, Cout, k, k, 1, p, ws=tiny, split_k=cfg['split'], tile=64, bias=b.to(DEV), add=rd, relu=True)
    assert 'splitk' not in O.LAST_PLAN


@pytest.mark.parametrize('algo', [5, 6])
@pytest.mark.parametrize('cfg', [
    dict(n=1, H=38, W=63, Cin=256, Cout=256, k=3, s=1, p=1),          # layer3 3x3: 18 slices of 128 channels, image borders, ragged pixel tile
    dict(n=1, H=38, W=63, Cin=1024, Cout=256, k=1, s=1, p=0),         # layer3 1x1-in
    dict(n=1, H=38, W=63, Cin=256, Cout=1024, k=1, s=1, p=0),         # layer3 1x1-out: two slices
    dict(n=1, H=75, W=125, Cin=128, Cout=128, k=3, s=1, p=1),         # layer2 3x3: one slice per tap, 294 tiles
    dict(n=1, H=75, W=125, Cin=128, Cout=512, k=1, s=1, p=0),         # a single slice of K
    dict(n=1, H=75, W=125, Cin=512, Cout=256, k=1, s=2, p=0),         # layer3.0 conv1: strided 1x1
    dict(n=5, H=7, W=7, Cin=128, Cout=200, k=3, s=1, p=1),            # several small images, ragged channel tile
])
def test_conv_ksplit_tile(cfg, algo):
    """L2S_ALGO_KSPLIT (64x64 tile, LDS-DMA fill by requester waves, the four multiplier waves split each slice's K and add their partial
    tiles in LDS in a fixed order; algo 6 = ring of three stages) against torch on the same rounded bf16 operands, in the forward form
    (bias + residual + ReLU), the data-gradient form (ReLU mask) and the strided-scatter form of a stride-2 data gradient."""
    O = ops()
    dt = 1
    g = torch.Generator().manual_seed(12)
    n, H, W, Cin, Cout, k, s, p = [cfg[x] for x in ['n', 'H', 'W', 'Cin', 'Cout', 'k', 's', 'p']]
    x = torch.randn(n, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g)
    OH = (H + 2 * p - k) // s + 1; OW = (W + 2 * p - k) // s + 1
    res = torch.randn(n, Cout, OH, OW, generator=g)
    xd, wd, rd = to_dev(nhwc(x), dt), to_dev(ohwi(w), dt), to_dev(nhwc(res), dt)
    xr, wr, rr = xd.float().cpu().permute(0, 3, 1, 2), wd.float().cpu().permute(0, 3, 1, 2), rd.float().cpu().permute(0, 3, 1, 2)
    conv = F.conv2d(xr, wr, None, stride=s, padding=p)
    y = torch.full((n * OH * OW, Cout), float('nan'), dtype=torch.bfloat16, device=DEV)
    O.conv_igemm(xd, wd, y, n, H, W, Cin, OH, OW, Cout, k, k, s, p, bias=b.to(DEV), add=rd, relu=True, algo=algo)
    torch.cuda.synchronize()
    assert rel_err(y.float().view(n, OH, OW, Cout), nhwc(F.relu(conv + b.view(1, -1, 1, 1) + rr))) < TOL[dt]
    y2 = torch.full((n * OH * OW, Cout), float('nan'), dtype=torch.bfloat16, device=DEV)
    O.conv_igemm(xd, wd, y2, n, H, W, Cin, OH, OW, Cout, k, k, s, p, ref=rd, algo=algo)
    torch.cuda.synchronize()
    assert rel_err(y2.float().view(n, OH, OW, Cout), nhwc(conv * (rr > 0))) < TOL[dt]
    y3 = torch.empty_like(y2)
    for _ in range(2):
        O.conv_igemm(xd, wd, y3, n, H, W, Cin, OH, OW, Cout, k, k, s, p, ref=rd, algo=algo)
    torch.cuda.synchronize()
    assert torch.equal(y2, y3)                                        # fixed summation order of the four K quarters
    if k == 1 and s == 1 and Cin % 128 == 0:
        # the data gradient of a stride-2 1x1 convolution: a 1x1 product whose rows are scattered to every second pixel of a larger map
        xr2 = xd.float().cpu().view(n * H * W, Cin)
        wr2 = wd.float().cpu().view(Cout, Cin)
        out = torch.zeros((n, 2 * H, 2 * W, Cout), dtype=torch.bfloat16, device=DEV)
        O.conv_igemm(xd, wd, out, n, H, W, Cin, H, W, Cout, 1, 1, 1, 0, scatter=(2 * H, 2 * W, 2), algo=algo)
        torch.cuda.synchronize()
        ref = (xr2 @ wr2.t()).view(n, H, W, Cout)
        assert rel_err(out.float()[:, ::2, ::2, :], ref) < TOL[dt]
        assert float(out.float()[:, 1::2, :, :].abs().max()) == 0.0


@pytest.mark.parametrize('cfg', [
    dict(H=38, W=63, Cin=256, Cout=256),          # layer3 conv2: 19 x 8 tiles of 128 pixels x 32 channels, ragged last pixel tile
    dict(H=38, W=63, Cin=512, Cout=512),          # layer4 on the map: 64-channel tiles
    dict(H=20, W=64, Cin=64, Cout=32),            # W + 1 = 65: the 384-row patch
    dict(H=9, W=127, Cin=64, Cout=48),            # the widest row the patch holds, ragged channel tile
    dict(H=75, W=125, Cin=128, Cout=128),         # layer2 conv2
    dict(H=1, W=1, Cin=64, Cout=16),              # one pixel: both edges on the same lane
    dict(H=7, W=2, Cin=192, Cout=80),             # every pixel is an edge pixel; six steps
    dict(H=3, W=200, Cin=64, Cout=64),            # a row wider than the patch: the plan falls back (still correct)
])
def test_conv_patch_tile(cfg):
    """L2S_ALGO_PATCH (3x3 / stride 1 / pad 1 on one map: 128 consecutive pixels x 32 or 64 channels per workgroup, the input patch staged
    once per 32-channel step, taps as shifted fragment reads, edge lanes zeroed in registers) against fp64 convolution of the same rounded
    bf16 operands: forward form (bias + residual + ReLU) and data-gradient form (add + ReLU mask); the automatic choice gives the same bits."""
    O = ops()
    g = torch.Generator().manual_seed(21)
    H, W, Cin, Cout = [cfg[x] for x in ['H', 'W', 'Cin', 'Cout']]
    M = H * W
    x = torch.randn(1, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / np.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g)
    res = torch.randn(1, Cout, H, W, generator=g)
    xd, wd, rd = to_dev(nhwc(x), 1), to_dev(ohwi(w), 1), to_dev(nhwc(res), 1)
    xr, wr, rr = [t.double().cpu().permute(0, 3, 1, 2) for t in (xd, wd, rd)]
    conv = F.conv2d(xr, wr, None, padding=1)
    for form in ('fwd', 'dgrad'):
        kw = dict(bias=b.to(DEV), add=rd, relu=True) if form == 'fwd' else dict(add=rd, ref=rd)
        ref = F.relu(conv + b.double().view(1, -1, 1, 1) + rr) if form == 'fwd' else (conv + rr) * (rr > 0)
        y = torch.full((M, Cout), float('nan'), dtype=torch.bfloat16, device=DEV)
        O.conv_igemm(xd, wd, y, 1, H, W, Cin, H, W, Cout, 3, 3, 1, 1, algo=3, **kw)
        plan = O.LAST_PLAN
        ya = torch.full((M, Cout), float('nan'), dtype=torch.bfloat16, device=DEV)
        O.conv_igemm(xd, wd, ya, 1, H, W, Cin, H, W, Cout, 3, 3, 1, 1, **kw)
        torch.cuda.synchronize()
        assert ('igemm_p3' in plan) == (W + 1 <= 128), plan
        assert rel_err(y.float().view(1, H, W, Cout), nhwc(ref.float())) < TOL[1], (form, plan)
        assert rel_err(ya.float().view(1, H, W, Cout), nhwc(ref.float())) < TOL[1], (form, O.LAST_PLAN)
        if 'igemm_p3' in O.LAST_PLAN:
            assert torch.equal(y, ya)


@pytest.mark.parametrize('dt', [0, 1])
@pytest.mark.parametrize('cfg', [
    dict(n=1, H=19, W=23, Cin=64, Cout=128, k=3, s=1, p=1),
    dict(n=1, H=38, W=63, Cin=256, Cout=256, k=3, s=1, p=1),
    dict(n=5, H=7, W=7, Cin=128, Cout=64, k=1, s=1, p=0),
    dict(n=1, H=20, W=26, Cin=256, Cout=128, k=1, s=2, p=0),
    dict(n=1, H=14, W=14, Cin=512, Cout=72, k=1, s=1, p=0),
    dict(n=256, H=7, W=7, Cin=512, Cout=512, k=3, s=1, p=1),          # dominant shape: dgrad on the 224x128 tile, wgrad over 12544 pixels
    dict(n=256, H=7, W=7, Cin=1024, Cout=512, k=1, s=1, p=0),         # layer4 1x1-in @ RoIs (the 128x128 wgrad tile)
    dict(n=1, H=38, W=63, Cin=256, Cout=1024, k=1, s=1, p=0),         # layer3 1x1-out
    dict(n=1, H=75, W=125, Cin=128, Cout=128, k=3, s=1, p=1),         # layer2 3x3
])
def test_conv_bwd(cfg, dt):
    """data gradient = igemm over dY with transposed/flipped weights; weight gradient = wgrad kernel."""
    O = ops()
    g = torch.Generator().manual_seed(2)
    n, H, W, Cin, Cout, k, s, p = [cfg[x] for x in ['n', 'H', 'W', 'Cin', 'Cout', 'k', 's', 'p']]
    OH = (H + 2 * p - k) // s + 1; OW = (W + 2 * p - k) // s + 1
    x = torch.randn(n, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)
    dy = torch.randn(n, Cout, OH, OW, generator=g)
    xd, dyd = to_dev(nhwc(x), dt), to_dev(nhwc(dy), dt)
    w_master = ohwi(w).to(DEV)
    scale = (torch.rand(Cout, generator=g) + 0.5)
    xr = xd.float().cpu().permute(0, 3, 1, 2).clone().requires_grad_(True)
    dyr = dyd.float().cpu().permute(0, 3, 1, 2)
    # effective (BN-folded) weight as the kernels see it
    wt = O.empty((Cin, k * k, Cout), dt)
    O.weight_transpose(w_master, scale.to(DEV), wt, Cout, k * k, Cin)
    wf = O.empty((Cout, k * k, Cin), dt)
    O.weight_cast(w_master, scale.to(DEV), wf, Cout, k * k, Cin)
    torch.cuda.synchronize()
    weff = wf.float().cpu().view(Cout, k, k, Cin).permute(0, 3, 1, 2).clone().requires_grad_(True)
    assert rel_err(wf.float().view(Cout, k, k, Cin), ohwi(w) * scale.view(-1, 1, 1, 1)) < (1e-6 if dt == 0 else 1e-2)
    out = F.conv2d(xr, weff, None, stride=s, padding=p)
    out.backward(dyr)
    # dgrad
    dx = torch.zeros((n * H * W, Cin), dtype=O.TORCH_DT[dt], device=DEV)
    if s == 1:
        O.conv_igemm(dyd, wt, dx, n, OH, OW, Cout, H, W, Cin, k, k, 1, k - 1 - p)
    else:
        O.conv_igemm(dyd, wt, dx, n, OH, OW, Cout, OH, OW, Cin, 1, 1, 1, 0, scatter=(H, W, s))
    torch.cuda.synchronize()
    assert rel_err(dx.float().view(n, H, W, Cin), nhwc(xr.grad)) < TOL[dt]
    # wgrad (accumulates on top of existing content); split-K partial slabs go through the workspace and are summed in a fixed order
    ws = torch.empty(16 << 20, dtype=torch.float32, device=DEV)
    dw = torch.ones((Cout, k * k * Cin), dtype=torch.float32, device=DEV)
    O.conv_wgrad(dyd, xd, dw, n, H, W, Cin, OH, OW, Cout, k, k, s, p, ws=ws)
    torch.cuda.synchronize()
    refw = ohwi(weff.grad).view(Cout, k * k * Cin) + 1.0
    # fp32 accumulation of exact products of the same rounded operands in both modes
    assert rel_err(dw, refw) < 1e-4
    # the weight gradient is reproducible bit for bit from run to run (no floating-point atomics)
    dw2 = torch.ones((Cout, k * k * Cin), dtype=torch.float32, device=DEV)
    O.conv_wgrad(dyd, xd, dw2, n, H, W, Cin, OH, OW, Cout, k, k, s, p, ws=ws)
    torch.cuda.synchronize()
    assert torch.equal(dw, dw2)
    # every tile / split choice (forced split-K, no workspace = unsplit, the other tile, per-tap tiles instead of shared filter rows)
    for kw in (dict(split_k=3, ws=ws), dict(), dict(tile=128, ws=ws), dict(tile=64, split_k=2, ws=ws)):
        dw3 = torch.ones((Cout, k * k * Cin), dtype=torch.float32, device=DEV)
        O.conv_wgrad(dyd, xd, dw3, n, H, W, Cin, OH, OW, Cout, k, k, s, p, **kw)
        torch.cuda.synchronize()
        assert rel_err(dw3, refw) < 1e-4, kw


@pytest.mark.parametrize('dt', [0, 1])
def test_conv_wgrad_grouped(dt):
    """l2s_conv_wgrad_grouped: the weight gradients of several convolutions in one launch per tile variant (what a backward stage of the
    step issues), incl. a tensor used twice (two pixel segments of different shape, like resnet.layer4 on the RoIs and on the map);
    against torch autograd on the same rounded operands, accumulating on top of existing content, bit-identical from run to run."""
    import ctypes as C
    from lang2seg_amd._lib import WgradProb, load
    O = ops()
    lib = load()
    g = torch.Generator().manual_seed(5)
    # (uses [(n, H, W)], Cin, Cout, k, stride, pad)
    convs = [([(1, 19, 23)], 64, 128, 3, 1, 1), ([(3, 7, 7), (1, 11, 13)], 128, 64, 3, 1, 1), ([(2, 9, 9)], 256, 72, 1, 1, 0),
             ([(40, 7, 7), (1, 10, 12)], 512, 512, 1, 1, 0), ([(1, 20, 26)], 256, 128, 1, 2, 0), ([(20, 7, 7)], 512, 512, 3, 1, 1),
             ([(170, 7, 7), (1, 10, 12)], 512, 768, 1, 1, 0),          # >= 8192 pixels, 256-multiples: the 8-wave 256x256 tile in bf16
             ([(170, 7, 7), (1, 10, 12)], 256, 384, 3, 1, 1), ([(180, 7, 7)], 128, 128, 3, 1, 1)]   # 3x3, >= 8192 pixels: the LDS-DMA filter-row tile in bf16
    byv, keep = {}, []
    for uses, Cin, Cout, k, s, p in convs:
        dw = torch.ones((Cout, k * k * Cin), dtype=torch.float32, device=DEV)
        ref = torch.zeros(Cout, Cin, k, k)
        q = WgradProb()
        q.dw, q.nseg, q.Cin, q.Cout, q.KH, q.KW, q.stride, q.pad = dw.data_ptr(), len(uses), Cin, Cout, k, k, s, p
        Mmax, same = 0, True
        for si, (n, H, W) in enumerate(uses):
            OH = (H + 2 * p - k) // s + 1; OW = (W + 2 * p - k) // s + 1
            x = torch.randn(n, Cin, H, W, generator=g); dy = torch.randn(n, Cout, OH, OW, generator=g)
            xd, dyd = to_dev(nhwc(x), dt), to_dev(nhwc(dy), dt)
            keep += [xd, dyd]
            w0 = torch.zeros(Cout, Cin, k, k, requires_grad=True)
            F.conv2d(xd.float().cpu().permute(0, 3, 1, 2), w0, None, stride=s, padding=p).backward(dyd.float().cpu().permute(0, 3, 1, 2))
            ref += w0.grad
            q.dy[si], q.x[si] = dyd.data_ptr(), xd.data_ptr()
            q.n_img[si], q.IH[si], q.IW[si], q.OH[si], q.OW[si], q.lddy[si], q.ldx[si] = n, H, W, OH, OW, Cout, Cin
            Mmax = max(Mmax, n * OH * OW); same = same and OH == H and OW == W
        v = int(lib.l2s_wgrad_variant(Cin, Cout, k, k, s, p, int(same), Mmax, 256 if dt == 1 else 0))
        byv.setdefault(v, []).append((q, dw, ohwi(ref).reshape(Cout, k * k * Cin) + 1.0))
    assert len(byv) >= 3                                        # per-tap and filter-row tiles, 64- and 128-wide
    assert (4 in byv) == (dt == 1) and (5 in byv) == (dt == 1)
    if 4 in byv:                                                # the LDS-DMA 256x256 tile (variant 6, on request only) on copies of the register-staged tile's problems
        byv[6] = []
        for q, dw, ref in byv[4]:
            q6 = WgradProb.from_buffer_copy(bytes(q)); dw6 = torch.ones_like(dw); q6.dw = dw6.data_ptr()
            byv[6].append((q6, dw6, ref))
    first = {}
    for rep in range(4):                                        # 2, 3: flags = 1 - the gradient is written, whatever dW held (unsplit / split)
        for v, lst in byv.items():
            arr = (WgradProb * len(lst))(*[q for q, _, _ in lst])
            if rep:
                for _, dw, _ in lst:
                    dw.fill_(1.0 if rep == 1 else float('nan'))
            for q_ in arr:
                q_.flags = int(rep >= 2)
            tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(DEV)
            if rep in (1, 3) and v not in (5, 6):                # pixels of the first problem cut into 3 ranges (slabs, fixed-order sum)
                arr[0].split, arr[0].ws_off = 3, 0
                tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(DEV)
            ws = torch.empty(max(8 << 20, int(lib.l2s_wgrad_grouped_ws_bytes(v)) // 4), dtype=torch.float32, device=DEV)
            O.call('l2s_conv_wgrad_grouped', tab.data_ptr(), C.cast(arr, C.c_void_p), len(lst), v, dt, ws.data_ptr(), ws.numel() * 4, O.stream())
            torch.cuda.synchronize()
            for i, (_, dw, ref) in enumerate(lst):
                assert rel_err(dw, ref - (1.0 if rep >= 2 else 0.0)) < 1e-4, (v, rep, i)
                if rep == 0:
                    first[(v, i)] = dw.clone()
                elif rep == 1 and (i > 0 or v in (5, 6)):
                    assert torch.equal(dw, first[(v, i)])       # unsplit problems: bit-identical from run to run


@pytest.mark.parametrize('dt', [0, 1])
def test_deconv_and_colsum(dt):
    O = ops()
    g = torch.Generator().manual_seed(3)
    n, Cin, Cq = 6, 256, 64
    x = torch.randn(n, Cin, 7, 7, generator=g)
    w = torch.randn(Cin, Cq, 2, 2, generator=g) * 0.05       # ConvTranspose2d layout (Cin, Cout, kh, kw)
    b = torch.randn(Cq, generator=g)
    xd = to_dev(nhwc(x), dt)
    wg = w.permute(2, 3, 1, 0).contiguous().view(4 * Cq, Cin)  # [(dy,dx,co)][ci]
    wd = to_dev(wg, dt)
    xr = xd.float().cpu().permute(0, 3, 1, 2)
    wr = wd.float().cpu().view(2, 2, Cq, Cin).permute(3, 2, 0, 1)
    ref = F.relu(F.conv_transpose2d(xr, wr, b, stride=2))
    y = O.empty((n * 14 * 14, Cq), dt)
    O.conv_igemm(xd, wd, y, n, 7, 7, Cin, 7, 7, 4 * Cq, bias=b.to(DEV), relu=True, deconv=True)
    torch.cuda.synchronize()
    assert rel_err(y.float().view(n, 14, 14, Cq), nhwc(ref)) < TOL[dt]
    out = torch.zeros(Cq, device=DEV)
    O.colsum(y, n * 196, Cq, Cq, out)
    torch.cuda.synchronize()
    assert rel_err(out, y.float().sum(0)) < 1e-4


@pytest.mark.parametrize('dt', [0, 1])
def test_stem_maxpool(dt):
    O = ops()
    g = torch.Generator().manual_seed(4)
    H, W = 61, 83
    img = torch.randn(1, H, W, 3, generator=g) * 50
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.01
    sc = torch.rand(64, generator=g) + 0.5; bi = torch.randn(64, generator=g) * 0.1
    OH, OW = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    y = O.empty((OH * OW, 64), dt)
    O.stem_conv(img.to(DEV), ohwi(w).to(DEV), sc.to(DEV), bi.to(DEV), y, H, W, OH, OW)
    ref = F.relu(F.conv2d(img.permute(0, 3, 1, 2), w, None, stride=2, padding=3) * sc.view(1, -1, 1, 1) + bi.view(1, -1, 1, 1))
    torch.cuda.synchronize()
    assert rel_err(y.float().view(1, OH, OW, 64), nhwc(ref)) < (1e-5 if dt == 0 else 1e-2)
    PH, PW = (OH + 2 - 3) // 2 + 1, (OW + 2 - 3) // 2 + 1
    yp = O.empty((PH * PW, 64), dt)
    O.maxpool(y, yp, OH, OW, 64, PH, PW)
    refp = F.max_pool2d(y.float().cpu().view(1, OH, OW, 64).permute(0, 3, 1, 2), 3, 2, 1)
    torch.cuda.synchronize()
    assert rel_err(yp.float().view(1, PH, PW, 64), nhwc(refp)) < 1e-6


def test_weight_transpose_batched_from_shadow():
    """l2s_weight_transpose_batched: the data-gradient copies [Cin][taps flipped][Cout] of several layers in one launch, read from the f32
    master (* scale) and - bf16 mode, trainable layers - from the bf16 shadow the update keeps current (l2s_transpose_desc.force_f32 bit 1):
    bit-identical outputs, equal to a torch permute of the rounded weights."""
    import ctypes as C
    from lang2seg_amd._lib import TransposeDesc
    O = ops()
    g = torch.Generator().manual_seed(9)
    shapes = [(512, 9, 512), (72, 1, 256), (2048, 1, 512), (256, 9, 64), (24, 1, 516)]
    keep, tabs = [], {}
    for mode in ('master', 'shadow'):
        arr = (TransposeDesc * len(shapes))()
        outs = []
        gg = torch.Generator().manual_seed(9)
        for i, (Cout, taps, Cin) in enumerate(shapes):
            w = (torch.randn(Cout, taps, Cin, generator=gg) * 0.1).to(DEV)
            sc = (torch.rand(Cout, generator=gg) + 0.5).to(DEV)
            sh = (w * sc.view(-1, 1, 1)).bfloat16().contiguous()
            dst = torch.full((Cin, taps, Cout), float('nan'), device=DEV).bfloat16()
            keep += [w, sc, sh, dst]
            if mode == 'master':
                arr[i].src, arr[i].scale, arr[i].force_f32 = w.data_ptr(), sc.data_ptr(), 0
            else:
                arr[i].src, arr[i].scale, arr[i].force_f32 = sh.data_ptr(), None, 2
            arr[i].dst, arr[i].Cout, arr[i].taps, arr[i].Cin = dst.data_ptr(), Cout, taps, Cin
            outs.append((dst, sh))
        tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(DEV)
        tiles = sum(((Cin + 63) // 64) * ((Cout + 63) // 64) * taps for (Cout, taps, Cin) in shapes)
        O.weight_transpose_batched(tab, len(shapes), tiles, 1)
        torch.cuda.synchronize()
        tabs[mode] = outs
    for (a, sh), (b, _) in zip(tabs['master'], tabs['shadow']):
        ref = sh.flip(1).permute(2, 1, 0).contiguous()
        assert torch.equal(a, ref) and torch.equal(b, ref)


@pytest.mark.parametrize('hw', [(600, 1000), (37, 53), (64, 64), (7, 250), (131, 9)])
def test_stem_pool_mfma(hw):
    """l2s_stem_pool_bf16 (stem + frozen-BN affine + ReLU + 3x3/2 pooling on the matrix cores, one launch; resnet_v1.py:121-126) against
    torch in f32 with the SAME bf16-rounded weights (what differs then is the image's 16-bit split and the summation order: far below the
    output's bf16 rounding), against f32 weights (the bf16 mode's weight rounding on top), and against the two-launch path it replaces."""
    O = ops()
    H, W = hw
    g = torch.Generator().manual_seed(11)
    img = torch.randn(1, H, W, 3, generator=g) * 60 + 10
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.02
    sc = torch.rand(64, generator=g) + 0.5; bi = torch.randn(64, generator=g) * 0.3
    OH, OW = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    PH, PW = (OH + 2 - 3) // 2 + 1, (OW + 2 - 3) // 2 + 1
    wd = ohwi(w).to(DEV)
    pack = O.stem_pack(wd)
    yp = torch.full((PH * PW, 64), float('nan'), device=DEV).bfloat16()
    O.stem_pool_bf16(img.to(DEV), pack, sc.to(DEV), bi.to(DEV), yp, H, W, OH, OW, PH, PW)
    torch.cuda.synchronize()
    def ref(wt):
        c = F.relu(F.conv2d(img.double().permute(0, 3, 1, 2), wt.double(), None, stride=2, padding=3) * sc.double().view(1, -1, 1, 1) + bi.double().view(1, -1, 1, 1))
        return nhwc(F.max_pool2d(c, 3, 2, 1)).float()
    got = yp.float().view(1, PH, PW, 64)
    assert bool(torch.isfinite(got).all())
    r_same = ref(w.bfloat16().float())
    assert rel_err(got, r_same.bfloat16().float()) < 4e-3          # one bf16 ulp where the f32 sums straddle a rounding boundary
    assert float(((got.cpu() - r_same).abs() / (r_same.abs() + 1.0)).mean()) < 2e-3
    assert rel_err(got, ref(w)) < 1e-2                              # the bf16 mode's weight rounding
    # the two-launch path (f32 weights, bf16 output)
    y = O.empty((OH * OW, 64), 1)
    O.stem_conv(img.to(DEV), wd, sc.to(DEV), bi.to(DEV), y, H, W, OH, OW)
    y2 = O.empty((PH * PW, 64), 1)
    O.maxpool(y, y2, OH, OW, 64, PH, PW)
    torch.cuda.synchronize()
    assert rel_err(got, y2.float().view(1, PH, PW, 64)) < 1e-2


@pytest.mark.parametrize('dt', [0, 1])
def test_pools(dt):
    O = ops()
    g = torch.Generator().manual_seed(5)
    n, hw, Cc = 5, 49, 192
    x = torch.randn(n, hw, Cc, generator=g)
    xd = to_dev(x, dt)
    y = O.empty((n, Cc), dt)
    O.avgpool_fwd(xd, y, n, hw, Cc)
    torch.cuda.synchronize()
    assert rel_err(y.float(), xd.float().mean(1)) < (1e-5 if dt == 0 else 1e-2)
    dy = to_dev(torch.randn(n, Cc, generator=g), dt)
    add = to_dev(torch.randn(n, hw, Cc, generator=g), dt)
    dx = O.empty((n, hw, Cc), dt)
    O.avgpool_bwd(dy, dx, add, xd, n, hw, Cc)
    ref = (dy.float().cpu().unsqueeze(1) / hw + add.float().cpu()) * (xd.float().cpu() > 0)
    torch.cuda.synchronize()
    assert rel_err(dx.float(), ref) < (1e-5 if dt == 0 else 1e-2)
    # adaptive pool 38x63 -> 14x14 with a pixel mask, forward and backward
    H, W, Cc = 38, 63, 128
    f = torch.randn(1, Cc, H, W, generator=g)
    pm = (torch.rand(H, W, generator=g) > 0.5).float()
    fd = to_dev(nhwc(f), dt)
    fr = fd.float().cpu().permute(0, 3, 1, 2).clone().requires_grad_(True)
    ya = F.adaptive_avg_pool2d(fr, [14, 14]); ym = F.adaptive_avg_pool2d(fr * pm, [14, 14])
    refy = torch.cat((ya.permute(0, 2, 3, 1), ym.permute(0, 2, 3, 1)), 3).reshape(196, 2 * Cc)
    yo = O.empty((196, 2 * Cc), dt)
    O.adaptive_pool_fwd(fd, None, yo, H, W, Cc, 14, 14, 2 * Cc)
    O.adaptive_pool_fwd(fd, pm.to(DEV), yo[:, Cc:], H, W, Cc, 14, 14, 2 * Cc)
    torch.cuda.synchronize()
    assert rel_err(yo.float(), refy) < (1e-5 if dt == 0 else 1e-2)
    dyo = to_dev(torch.randn(196, 2 * Cc, generator=g), dt)
    refy.backward(dyo.float().cpu())
    dxo = O.empty((H * W, Cc), dt)
    O.adaptive_pool_bwd(dyo, 2 * Cc, 0, Cc, pm.to(DEV), dxo, None, H, W, Cc, 14, 14)
    torch.cuda.synchronize()
    assert rel_err(dxo.float().view(1, H, W, Cc), nhwc(fr.grad)) < (1e-5 if dt == 0 else 1e-2)
    dxr = O.empty((H * W, Cc), dt)                               # with the ReLU mask of the pooled map (the caption branch's call)
    O.adaptive_pool_bwd(dyo, 2 * Cc, 0, Cc, pm.to(DEV), dxr, fd.view(H * W, Cc), H, W, Cc, 14, 14)
    torch.cuda.synchronize()
    assert torch.equal(dxr.float(), dxo.float() * (fd.view(H * W, Cc).float() > 0))
    # gt mask downsample
    m = (torch.rand(600, 1000, generator=g) > 0.6).to(torch.uint8)
    out = torch.empty(38 * 63, device=DEV)
    O.mask_downsample(m.to(DEV), out, 600, 1000, 38, 63)
    refm = (F.adaptive_avg_pool2d(m.float()[None, None], [38, 63]) >= 0.5).float().view(-1)
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), refm)


def _rpn_inputs(H, W, A, seed, gain=1.0):
    rs = np.random.RandomState(seed)
    heads = rs.normal(0, gain, (H * W, 6 * A + 8)).astype(np.float32)
    heads[:, 2 * A:6 * A] *= 0.3
    return heads


def test_rpn_decode_sort_nms():
    O = ops()
    H, W, A = 20, 26, 12
    heads = _rpn_inputs(H, W, A, 6)
    anchors, n = OB.generate_anchors_pre(H, W, 16, (4, 8, 16, 32), (0.5, 1, 2))
    base = OB.generate_anchors(ratios=(0.5, 1, 2), scales=(4, 8, 16, 32)).astype(np.float32)
    hd = torch.from_numpy(heads).to(DEV)
    prob = torch.empty(H * W, 2 * A, device=DEV); boxes = torch.empty(n, 4, device=DEV); scores = torch.empty(n, device=DEV)
    O.rpn_decode(hd, heads.shape[1], torch.from_numpy(base).to(DEV), H, W, A, 16, 320.0, 416.0, prob, boxes, scores)
    cls = torch.from_numpy(heads[:, :2 * A]).view(H * W, 2, A)
    p_ref = F.softmax(cls, 1).view(H * W, 2 * A)
    deltas = heads[:, 2 * A:6 * A].reshape(-1, 4)
    b_ref = OB.clip_boxes(OB.bbox_transform_inv(anchors, deltas), (320, 416))
    torch.cuda.synchronize()
    assert rel_err(prob, p_ref) < 1e-5
    assert np.abs(boxes.cpu().numpy() - b_ref).max() < 2e-3
    # stable sort + top-k: bit-exact order on the device's own scores
    sc = scores.cpu().numpy(); bx = boxes.cpu().numpy()
    k = 1500
    sb = torch.empty(k, 4, device=DEV); ss = torch.empty(k, device=DEV); si = torch.empty(k, dtype=torch.int32, device=DEV)
    sws = torch.empty(O.sort_ws_ints(n), dtype=torch.int32, device=DEV)
    O.sort_topk(scores, boxes, n, k, sws, sb, ss, si)
    order = OB.stable_desc_order(sc)[:k]
    torch.cuda.synchronize()
    assert np.array_equal(si.cpu().numpy(), order.astype(np.int32))
    assert np.array_equal(sb.cpu().numpy(), bx[order])
    # NMS, both comparators, bit-exact keep lists vs numpy and C oracles
    import ctypes as C, os
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle', '_build', 'liboracle_ref.so')
    clib = C.CDLL(so) if os.path.exists(so) else None
    for cmp_mode, name in [(0, 'ge'), (1, 'gt')]:
        for max_keep in [300, 5000]:
            ws = torch.empty(O.nms_workspace_bytes(k) // 8 + 8, dtype=torch.int64, device=DEV)
            keep = torch.full((max_keep,), -1, dtype=torch.int32, device=DEV); num = torch.zeros(1, dtype=torch.int32, device=DEV)
            O.nms(sb, k, 0.7, cmp_mode, max_keep, ws, keep, num)
            dets = np.hstack((bx[order], sc[order][:, None]))
            ref_keep = OB.nms(dets, 0.7, name)[:max_keep]
            torch.cuda.synchronize()
            nk = int(num.item())
            assert nk == len(ref_keep)
            assert np.array_equal(keep.cpu().numpy()[:nk], ref_keep.astype(np.int32))
            if clib is not None:
                sbx = np.ascontiguousarray(bx[order]); ko = np.zeros(k, np.int64)
                if cmp_mode == 0:
                    od = np.arange(k, dtype=np.int64)
                    cn = clib.oracle_cpu_nms(sbx.ctypes.data_as(C.c_void_p), od.ctypes.data_as(C.c_void_p), C.c_long(k), C.c_float(0.7), ko.ctypes.data_as(C.c_void_p))
                else:
                    cn = clib.oracle_gpu_nms(sbx.ctypes.data_as(C.c_void_p), C.c_long(k), C.c_float(0.7), ko.ctypes.data_as(C.c_void_p))
                assert np.array_equal(ko[:cn][:max_keep], ref_keep)
    rois = torch.empty(300, 5, device=DEV); rsc = torch.empty(300, device=DEV)
    keep = torch.full((300,), -1, dtype=torch.int32, device=DEV); num = torch.zeros(1, dtype=torch.int32, device=DEV)
    ws = torch.empty(O.nms_workspace_bytes(k) // 8 + 8, dtype=torch.int64, device=DEV)
    O.nms(sb, k, 0.7, 0, 300, ws, keep, num)
    O.gather_rois(sb, ss, keep, num, 300, rois, rsc)
    torch.cuda.synchronize()
    nk = int(num.item()); kk = keep.cpu().numpy()[:nk]
    assert np.array_equal(rois.cpu().numpy()[:nk, 1:], bx[order][kk]) and (rois.cpu().numpy()[nk:] == 0).all()


def _full_size_heads(dist, H, W, A, seed):
    """RPN head outputs [H*W][6A+8] for the BASELINE map (38x63, A=12 -> 28 728 anchors) in three score / box regimes"""
    rs = np.random.RandomState(seed)
    heads = np.zeros((H * W, 6 * A + 8), np.float32)
    if dist == 'fresh':            # a freshly initialised RPN (weights N(0, 0.01)): every score within ~1e-3 of 0.5, boxes ~ anchors
        heads[:, :2 * A] = rs.normal(0, 2e-3, (H * W, 2 * A))
        heads[:, 2 * A:6 * A] = rs.normal(0, 5e-3, (H * W, 4 * A))
    elif dist == 'ties':           # heavy ties: logits on a coarse grid -> a few hundred distinct scores, stable order decides
        heads[:, :2 * A] = np.round(rs.normal(0, 1.0, (H * W, 2 * A)) * 4) / 4
        heads[:, 2 * A:6 * A] = rs.normal(0, 0.2, (H * W, 4 * A))
    else:                          # 'clustered': a trained RPN looking at a few objects: high scores and boxes pulled onto 6 centres
        heads[:, :2 * A] = rs.normal(0, 2.0, (H * W, 2 * A))
        heads[:, 2 * A:6 * A] = rs.normal(0, 0.05, (H * W, 4 * A))
        cy, cx = np.mgrid[0:H, 0:W]
        for k in range(6):
            oy, ox = rs.randint(4, H - 4), rs.randint(6, W - 6)
            near = (np.abs(cy - oy) <= 3) & (np.abs(cx - ox) <= 4)
            idx = np.where(near.reshape(-1))[0]
            # dx, dy move the anchor centre towards the object centre (in units of the anchor size); fg logits raised
            for a in range(A):
                heads[idx, 2 * A + 4 * a + 0] += (ox - cx.reshape(-1)[idx]) * 16.0 / 128.0
                heads[idx, 2 * A + 4 * a + 1] += (oy - cy.reshape(-1)[idx]) * 16.0 / 128.0
                heads[idx, A + a] += 3.0
    return heads


@pytest.mark.parametrize('dist', ['fresh', 'ties', 'clustered'])
def test_sort_nms_full_size(dist):
    """BASELINE-size proposal chain (proposal_layer.py:42-62): 28 728 anchors -> stable top 12 000 -> NMS 0.7 -> top 2000 (and the
    uncapped keep list), both comparators (nms.c:35-63 `>=`, nms_kernel.cu:56-66 `>`): sorted indices and keep lists bit-exact
    against the numpy and the C oracle.  n = 12 000 is 188 column blocks of the 64x64 bit mask: the multi-phase software-pipelined
    reduce and its early stop at max_keep, which the 1500-box test (24 blocks) does not reach."""
    import ctypes as C, os
    O = ops()
    H, W, A = 38, 63, 12
    heads = _full_size_heads(dist, H, W, A, {'fresh': 11, 'ties': 12, 'clustered': 13}[dist])
    anchors, n = OB.generate_anchors_pre(H, W, 16, (4, 8, 16, 32), (0.5, 1, 2))
    assert n == 28728
    base = OB.generate_anchors(ratios=(0.5, 1, 2), scales=(4, 8, 16, 32)).astype(np.float32)
    hd = torch.from_numpy(heads).to(DEV)
    prob = torch.empty(H * W, 2 * A, device=DEV); boxes = torch.empty(n, 4, device=DEV); scores = torch.empty(n, device=DEV)
    O.rpn_decode(hd, heads.shape[1], torch.from_numpy(base).to(DEV), H, W, A, 16, 600.0, 1000.0, prob, boxes, scores)
    torch.cuda.synchronize()
    sc = scores.cpu().numpy(); bx = boxes.cpu().numpy()
    if dist == 'fresh':
        assert np.abs(sc - 0.5).max() < 5e-3
    if dist == 'ties':
        assert len(np.unique(sc)) < 2000
    k = 12000
    sb = torch.empty(k, 4, device=DEV); ss = torch.empty(k, device=DEV); si = torch.empty(k, dtype=torch.int32, device=DEV)
    sws = torch.empty(O.sort_ws_ints(n), dtype=torch.int32, device=DEV)
    O.sort_topk(scores, boxes, n, k, sws, sb, ss, si)
    torch.cuda.synchronize()
    order = OB.stable_desc_order(sc)[:k]
    assert np.array_equal(si.cpu().numpy(), order.astype(np.int32))
    assert np.array_equal(sb.cpu().numpy(), bx[order]) and np.array_equal(ss.cpu().numpy(), sc[order])
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle', '_build', 'liboracle_ref.so')
    clib = C.CDLL(so)
    sbx = np.ascontiguousarray(bx[order])
    if dist == 'ties':
        # plant 100 pairs whose IoU is exactly the threshold in fp32 (70 / 100 == 0.7f): `>=` suppresses the second box, `>` keeps it
        for i in range(100):
            r = 100 + 37 * i
            x0 = 3000.0 + 20.0 * i
            sbx[r] = [x0, 0, x0 + 9, 9]; sbx[r + 1] = [x0, 0, x0 + 9, 6]
        sb.copy_(torch.from_numpy(sbx))
    counts = {}
    for cmp_mode, name in [(0, 'ge'), (1, 'gt')]:
        ko = np.zeros(k, np.int64)
        if cmp_mode == 0:
            od = np.arange(k, dtype=np.int64)
            cn = clib.oracle_cpu_nms(sbx.ctypes.data_as(C.c_void_p), od.ctypes.data_as(C.c_void_p), C.c_long(k), C.c_float(0.7), ko.ctypes.data_as(C.c_void_p))
        else:
            cn = clib.oracle_gpu_nms(sbx.ctypes.data_as(C.c_void_p), C.c_long(k), C.c_float(0.7), ko.ctypes.data_as(C.c_void_p))
        ref_all = ko[:cn]
        counts[name] = cn
        # the sorted scores carry ties: the numpy oracle must walk the list in the given order
        np_keep = OB.nms(np.hstack((sbx, -np.arange(k, dtype=np.float32)[:, None])), 0.7, name)
        assert np.array_equal(np_keep, ref_all), (dist, name, len(np_keep), cn)
        for max_keep in (2000, 12000):
            ws = torch.empty(O.nms_workspace_bytes(k) // 8 + 8, dtype=torch.int64, device=DEV)
            keep = torch.full((max_keep,), -1, dtype=torch.int32, device=DEV); num = torch.zeros(1, dtype=torch.int32, device=DEV)
            O.nms(sb, k, 0.7, cmp_mode, max_keep, ws, keep, num)
            torch.cuda.synchronize()
            nk = int(num.item())
            ref_keep = ref_all[:max_keep]
            assert nk == len(ref_keep), (dist, name, max_keep, nk, len(ref_keep))
            assert np.array_equal(keep.cpu().numpy()[:nk], ref_keep.astype(np.int32)), (dist, name, max_keep)
        if dist == 'clustered':
            assert cn < 12000                       # suppression chains are exercised
    if dist == 'ties':
        assert counts['gt'] == counts['ge'] + 100   # the planted threshold-exact pairs tell the two comparators apart


@pytest.mark.parametrize('n', [12000, 8192, 4097, 4096, 4033, 1500, 65, 64, 1])
def test_nms_deep_scan(n):
    """The proposal chain's NMS in the regime of the training step itself: tests/golden/nms_deep_boxes.npy holds the 12000 sorted boxes of the
    bench step after 40 updates (tools/proposal_depth.py) - 721 survive, the scan visits every box, all three 4096-box stages of
    l2s_nms run with their carry-in.  Prefixes of the list put the stage boundary in every position (full, one box over, ragged
    last block, a single block, a single box); keep lists bit-exact against the C oracle, both comparators, capped and uncapped."""
    import ctypes as C, os
    O = ops()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sbx = np.ascontiguousarray(np.load(os.path.join(root, 'tests', 'golden', 'nms_deep_boxes.npy'))[:n])
    clib = C.CDLL(os.path.join(root, 'oracle', '_build', 'liboracle_ref.so'))
    sb = torch.from_numpy(sbx).to(DEV)
    for cmp_mode in (0, 1):
        ko = np.zeros(n, np.int64)
        if cmp_mode == 0:
            od = np.arange(n, dtype=np.int64)
            cn = clib.oracle_cpu_nms(sbx.ctypes.data_as(C.c_void_p), od.ctypes.data_as(C.c_void_p), C.c_long(n), C.c_float(0.7), ko.ctypes.data_as(C.c_void_p))
        else:
            cn = clib.oracle_gpu_nms(sbx.ctypes.data_as(C.c_void_p), C.c_long(n), C.c_float(0.7), ko.ctypes.data_as(C.c_void_p))
        if n == 12000:
            assert cn < 2000 and ko[cn - 1] > 11900          # the regime: fewer than RPN_POST_NMS_TOP_N survive, the scan reaches the end
        for max_keep in (2000, 300, n):
            ws = torch.empty(O.nms_workspace_bytes(n) // 8 + 8, dtype=torch.int64, device=DEV)
            keep = torch.full((max_keep,), -1, dtype=torch.int32, device=DEV); num = torch.full((1,), -7, dtype=torch.int32, device=DEV)
            for rep in range(2):                                 # (the second call finds the workspace and the counters as the first left them)
                O.nms(sb, n, 0.7, cmp_mode, max_keep, ws, keep, num)
            torch.cuda.synchronize()
            ref = ko[:cn][:max_keep]
            assert int(num.item()) == len(ref), (n, cmp_mode, max_keep, int(num.item()), len(ref))
            assert np.array_equal(keep.cpu().numpy()[:len(ref)], ref.astype(np.int32)), (n, cmp_mode, max_keep)


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_anchor_target(seed):
    O = ops()
    H, W, A = 20, 26, 12
    rs = np.random.RandomState(seed)
    n = H * W * A
    gt = np.array([[60 + 20 * seed, 40, 260 + 30 * seed, 250, 7]], np.float32)
    if seed == 2:
        gt = np.vstack((gt, [[10, 10, 90, 120, 3]])).astype(np.float32)
    anchors, _ = OB.generate_anchors_pre(H, W, 16, (4, 8, 16, 32), (0.5, 1, 2))
    base = OB.generate_anchors(ratios=(0.5, 1, 2), scales=(4, 8, 16, 32)).astype(np.float32)
    fgk = rs.permutation(n).astype(np.uint32); bgk = rs.permutation(n).astype(np.uint32)
    ct = dict(ON.DEFAULT_CFG['TRAIN']); ct['RPN_BATCHSIZE'] = 64 if seed else 256
    lab, tg, inw, outw = OB.anchor_target_layer(H, W, gt, np.array([320, 416, 1.0]), anchors, A, ct, fgk, bgk)
    labels = torch.empty(n, dtype=torch.int32, device=DEV)
    t = torch.empty(H * W, 4 * A, device=DEV); i_ = torch.empty_like(t); o_ = torch.empty_like(t)
    ws = torch.empty(O.anchor_target_ws_ints(n), dtype=torch.int32, device=DEV)
    O.anchor_target(torch.from_numpy(gt).to(DEV), gt.shape[0], torch.from_numpy(base).to(DEV), H, W, A, 16, 320.0, 416.0,
                    torch.from_numpy(fgk.astype(np.int64)).to(torch.int32).to(DEV) if False else torch.from_numpy(fgk.view(np.int32)).to(DEV),
                    torch.from_numpy(bgk.view(np.int32)).to(DEV), 0.3, 0.7, ct['RPN_BATCHSIZE'], 0.5, labels, t, i_, o_, ws)
    torch.cuda.synchronize()
    assert np.array_equal(labels.cpu().numpy(), lab.reshape(-1).astype(np.int32))
    assert np.abs(t.cpu().numpy() - tg.reshape(H * W, -1)).max() < 1e-5
    assert np.array_equal(i_.cpu().numpy(), inw.reshape(H * W, -1))
    assert np.abs(o_.cpu().numpy() - outw.reshape(H * W, -1)).max() < 1e-8


@pytest.mark.parametrize('case', ['normal', 'few_bg', 'no_fg', 'only_fg', 'only_fg_few'])
def test_proposal_target(case):
    O = ops()
    rs = np.random.RandomState(11)
    im_h, im_w = 320, 416
    gt = np.array([[100, 80, 300, 260, 17]], np.float32)
    yy, xx = np.mgrid[0:im_h, 0:im_w]
    gm = ((((xx - 200) / 100.) ** 2 + ((yy - 170) / 90.) ** 2) <= 1).astype(np.uint8)[None]
    n_max, n = 300, 240
    b = rs.uniform(0, 300, (n, 4)).astype(np.float32); b[:, 2:] = np.minimum(b[:, :2] + rs.uniform(20, 200, (n, 2)), [im_w - 1, im_h - 1])
    if case.startswith('only_fg'):
        # proposal_target_layer.py:155-158: every candidate overlaps the gt box by >= 0.5 -> all R sampled RoIs are foreground
        # (without replacement from 240 candidates; with replacement when there are fewer than R)
        b = (gt[0, :4] + rs.normal(0, 6, (n, 4))).astype(np.float32); b = np.clip(b, 0, [im_w - 1, im_h - 1, im_w - 1, im_h - 1]).astype(np.float32)
        if case == 'only_fg_few':
            n = 20
    elif case != 'no_fg':
        b[:40] = gt[0, :4] + rs.normal(0, 12, (40, 4)); b[:40] = np.clip(b[:40], 0, [im_w - 1, im_h - 1, im_w - 1, im_h - 1])
    else:
        b[:, 0] = np.minimum(b[:, 0], 60); b[:, 2] = np.minimum(b[:, 2], 90)
    if case == 'few_bg':
        n = 60
    rois = np.zeros((n_max, 5), np.float32); rois[:n, 1:] = b[:n].astype(np.float32)
    sc = rs.rand(n_max).astype(np.float32)
    fgk = rs.permutation(n_max + 1).astype(np.uint32); bgk = rs.permutation(n_max + 1).astype(np.uint32)
    bgr = rs.randint(0, 1 << 30, 64).astype(np.uint32)
    ct = dict(ON.DEFAULT_CFG['TRAIN']); ct['BATCH_SIZE'] = 32
    ref = OB.proposal_target_layer(rois[:n], sc[:n], gt, gm, 81, ct, 14, fgk[:n], bgk[:n], None, bgr)
    R, fg_max = 32, 8
    slots = R if case.startswith('only_fg') else fg_max          # mask-target slots (TRAIN.MASK_SLOTS_ALL sizes them for the no-background case)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    out_rois = torch.empty(R, 5, device=DEV); labels = torch.empty(R, dtype=torch.int32, device=DEV)
    bt = torch.empty(R, 324, device=DEV); bi = torch.empty_like(bt); bo = torch.empty_like(bt)
    mt = torch.empty(slots, 196, device=DEV); counts = torch.zeros(4, dtype=torch.int32, device=DEV)
    ws = torch.empty(4 * (n_max + 1) + R + 16, dtype=torch.int32, device=DEV)
    O.proposal_target(d(rois), d(sc), d(np.array([n], np.int32)), n_max, d(gt), 1, d(gm), im_h, im_w, d(fgk.view(np.int32)),
                      d(bgk.view(np.int32)), d(bgr.view(np.int32)), R, fg_max, slots, 0.5, 0.5, 0.0, d(np.zeros(4, np.float32)),
                      d(np.array([.1, .1, .2, .2], np.float32)), d(np.ones(4, np.float32)), 81, 14, out_rois, labels, bt, bi, bo, mt,
                      counts, ws)
    torch.cuda.synchronize()
    r_rois, _, r_lab, r_bt, r_bi, r_bo, r_mt, _ = ref
    nfg = int(counts[0].item())
    assert nfg == r_mt.shape[0]
    if case.startswith('only_fg'):
        assert nfg == R and int(counts[2].item()) == 0 and (r_lab > 0).all()
    assert np.array_equal(out_rois.cpu().numpy(), r_rois)
    assert np.array_equal(labels.cpu().numpy(), r_lab.reshape(-1).astype(np.int32))
    assert np.abs(bt.cpu().numpy() - r_bt).max() < 1e-4
    assert np.array_equal(bi.cpu().numpy(), r_bi) and np.array_equal(bo.cpu().numpy(), r_bo)
    assert np.array_equal(mt.cpu().numpy()[:nfg].reshape(nfg, 14, 14), r_mt)


@pytest.mark.parametrize('dt', [0, 1])
def test_roialign(dt):
    O = ops()
    g = torch.Generator().manual_seed(7)
    H, W, Cc, R = 20, 26, 96, 9
    feat = torch.randn(1, Cc, H, W, generator=g)
    rs = np.random.RandomState(3)
    rois = np.zeros((R, 5), np.float32)
    rois[:, 1] = rs.uniform(0, 300, R); rois[:, 2] = rs.uniform(0, 200, R)
    rois[:, 3] = np.minimum(rois[:, 1] + rs.uniform(10, 200, R), 415); rois[:, 4] = np.minimum(rois[:, 2] + rs.uniform(10, 200, R), 319)
    rois[0, 1:] = [0, 0, 415, 319]
    fd = to_dev(nhwc(feat), dt)
    fr = fd.float().cpu().permute(0, 3, 1, 2).clone().requires_grad_(True)
    net = ON.OracleNet.__new__(ON.OracleNet); net.cfg = ON.DEFAULT_CFG; net.var = {}
    ref = net.crop_pool(fr, torch.from_numpy(rois))
    out = O.empty((R * 49, Cc), dt)
    O.roialign_fwd(fd, H, W, Cc, torch.from_numpy(rois).to(DEV), R, 7, 1.0 / 16.0, out)
    torch.cuda.synchronize()
    assert rel_err(out.float().view(R, 7, 7, Cc), ref.permute(0, 2, 3, 1)) < (2e-5 if dt == 0 else 1e-2)
    dout = to_dev(torch.randn(R, 7, 7, Cc, generator=g), dt)
    ref.backward(dout.float().cpu().permute(0, 3, 1, 2))
    dfeat = torch.full((H * W, Cc), 3.0, device=DEV)   # (garbage: the launch writes every element)
    O.roialign_bwd(dout, H, W, Cc, torch.from_numpy(rois).to(DEV), R, 7, 1.0 / 16.0, dfeat)
    torch.cuda.synchronize()
    assert rel_err(dfeat.view(1, H, W, Cc), nhwc(fr.grad)) < 1e-4


def test_cropalign_with_max_pool():
    """_crop_pool_layer_align (NET:151-182): l2s_cropalign_fwd/bwd (grid from the RoI in image pixels over im_info) + the 2x2 max pool,
    against autograd of the oracle's crop_pool with POOLING_ALIGN"""
    import copy
    O = ops()
    g = torch.Generator().manual_seed(9)
    H, W, Cc, R = 20, 26, 64, 11
    im_h, im_w = 317.0, 409.0                             # not a multiple of 16: the two parameterisations differ
    rs = np.random.RandomState(4)
    rois = np.zeros((R, 5), np.float32)
    rois[:, 1] = rs.uniform(0, 300, R); rois[:, 2] = rs.uniform(0, 200, R)
    rois[:, 3] = np.minimum(rois[:, 1] + rs.uniform(10, 200, R), im_w - 1); rois[:, 4] = np.minimum(rois[:, 2] + rs.uniform(10, 200, R), im_h - 1)
    rois[0, 1:] = [0, 0, im_w - 1, im_h - 1]
    fr = torch.randn(1, Cc, H, W, generator=g).requires_grad_(True)
    net = ON.OracleNet.__new__(ON.OracleNet); net.cfg = copy.deepcopy(ON.DEFAULT_CFG); net.cfg['POOLING_ALIGN'] = True; net.var = {}
    net._im_info = np.array([[im_h, im_w, 1.0]], np.float32)
    ref = net.crop_pool(fr, torch.from_numpy(rois))        # (R, C, 7, 7)
    fd = nhwc(fr.detach()).to(DEV).contiguous().view(H * W, Cc)
    rd = torch.from_numpy(rois).to(DEV)
    crop = torch.empty(R * 14 * 14, Cc, device=DEV)
    O.cropalign_fwd(fd, H, W, Cc, rd, R, 14, im_h, im_w, crop)
    pool = torch.empty(R * 49, Cc, device=DEV)
    O.maxpool2x2_fwd(crop, pool, R, 14, 14, Cc)
    torch.cuda.synchronize()
    assert rel_err(pool.view(R, 7, 7, Cc), ref.permute(0, 2, 3, 1)) < 2e-5
    dout = torch.randn(R, 7, 7, Cc, generator=g)
    ref.backward(dout.permute(0, 3, 1, 2))
    dcrop = torch.empty(R * 14 * 14, Cc, device=DEV)
    O.maxpool2x2_bwd(dout.to(DEV).view(R * 49, Cc).contiguous(), crop, dcrop, R, 14, 14, Cc, False)
    dfeat = torch.full((H * W, Cc), 3.0, device=DEV)   # (garbage: the launch writes every element)
    O.cropalign_bwd(dcrop, H, W, Cc, rd, R, 14, im_h, im_w, dfeat)
    torch.cuda.synchronize()
    assert rel_err(dfeat.view(1, H, W, Cc), nhwc(fr.grad)) < 1e-4


def test_roialign_bwd_clustered_rois():
    """the gather-form backward on 300 RoIs, 200 of them tiny boxes on one spot (every bin of a RoI lands on the same pixel and the
    per-pixel sample list overflows its LDS window) plus duplicates, against autograd of the oracle's crop_pool"""
    O = ops()
    g = torch.Generator().manual_seed(17)
    H, W, Cc, R = 20, 26, 64, 300
    rs = np.random.RandomState(5)
    rois = np.zeros((R, 5), np.float32)
    rois[:, 1] = rs.uniform(0, 300, R); rois[:, 2] = rs.uniform(0, 200, R)
    rois[:, 3] = np.minimum(rois[:, 1] + rs.uniform(10, 200, R), 415); rois[:, 4] = np.minimum(rois[:, 2] + rs.uniform(10, 200, R), 319)
    rois[:200, 1] = 100.0 + rs.uniform(0, 2, 200); rois[:200, 2] = 84.0 + rs.uniform(0, 2, 200)
    rois[:200, 3] = rois[:200, 1] + rs.uniform(1, 10, 200); rois[:200, 4] = rois[:200, 2] + rs.uniform(1, 10, 200)
    rois[200:210] = rois[210:220]
    rois[220, 1:] = [0, 0, 415, 319]
    rois[221, 1:] = [64, 32, 64, 32]                      # degenerate: integer sample coordinates, zero-weight corners
    fr = torch.randn(1, Cc, H, W, generator=g).requires_grad_(True)
    net = ON.OracleNet.__new__(ON.OracleNet); net.cfg = ON.DEFAULT_CFG; net.var = {}
    ref = net.crop_pool(fr, torch.from_numpy(rois))
    dout = torch.randn(R, 7, 7, Cc, generator=g)
    ref.backward(dout.permute(0, 3, 1, 2))
    dfeat = torch.full((H * W, Cc), 3.0, device=DEV)   # (garbage: the launch writes every element)
    O.roialign_bwd(dout.to(DEV), H, W, Cc, torch.from_numpy(rois).to(DEV), R, 7, 1.0 / 16.0, dfeat)
    torch.cuda.synchronize()
    assert rel_err(dfeat.view(1, H, W, Cc), nhwc(fr.grad)) < 1e-4


def test_losses():
    O = ops()
    g = torch.Generator().manual_seed(8)
    H, W, A = 12, 16, 12
    n = H * W * A
    ldh = 6 * A + 8
    heads = torch.randn(H * W, ldh, generator=g)
    hr = heads.clone().requires_grad_(True)
    labels_hwa = torch.randint(-1, 2, (H, W, A), generator=g)
    lab_ahw = labels_hwa.permute(2, 0, 1).contiguous().view(-1)
    tg = torch.randn(H * W, 4 * A, generator=g) * 0.3
    inw = (labels_hwa == 1).float().view(H * W, A, 1).expand(-1, -1, 4).reshape(H * W, 4 * A).contiguous()
    outw = (labels_hwa >= 0).float().view(H * W, A, 1).expand(-1, -1, 4).reshape(H * W, 4 * A).contiguous() / 50.0
    # oracle: NET:377-390 on the NCHW-equivalent views
    cls = hr[:, :2 * A].view(H * W, 2, A)                       # [pix][k][a]
    cls_rs = cls.permute(2, 0, 1).reshape(-1, 2)                 # rows (a,h,w)
    sel = (lab_ahw != -1).nonzero().view(-1)
    l_cls = F.cross_entropy(cls_rs[sel], lab_ahw[sel])
    l_box = ON.OracleNet.smooth_l1(hr[:, 2 * A:6 * A].view(1, H, W, 4 * A), tg.view(1, H, W, -1), inw.view(1, H, W, -1), outw.view(1, H, W, -1), 3.0, [1, 2, 3])
    (l_cls + l_box).backward()
    loss = torch.zeros(8, device=DEV)
    dh = torch.full((H * W, ldh), 7.0, device=DEV)
    O.rpn_loss(heads.to(DEV), ldh, lab_ahw.to(torch.int32).to(DEV), tg.to(DEV), inw.to(DEV), outw.to(DEV), H, W, A, 3.0, 1.0, loss, dh, ldh)
    torch.cuda.synchronize()
    assert abs(loss[0].item() - l_cls.item()) < 1e-5 and abs(loss[1].item() - l_box.item()) < 1e-5
    assert rel_err(dh, hr.grad) < 1e-5
    # rcnn
    R, ncls = 32, 81
    ldh = 5 * ncls + 3
    heads = torch.randn(R, ldh, generator=g); hr = heads.clone().requires_grad_(True)
    lab = torch.randint(0, ncls, (R,), generator=g)
    bt = torch.randn(R, 4 * ncls, generator=g); bi = (torch.rand(R, 4 * ncls, generator=g) > 0.9).float(); bo = bi.clone()
    l1 = F.cross_entropy(hr[:, :ncls], lab)
    l2 = ON.OracleNet.smooth_l1(hr[:, ncls:5 * ncls], bt, bi, bo, 1.0, [1])
    (l1 + l2).backward()
    loss.zero_()
    dh = torch.full((R, ldh), 7.0, device=DEV)
    O.rcnn_loss(heads.to(DEV), ldh, lab.to(torch.int32).to(DEV), bt.to(DEV), bi.to(DEV), bo.to(DEV), R, ncls, 1.0, loss, dh, ldh)
    torch.cuda.synchronize()
    assert abs(loss[2].item() - l1.item()) < 1e-5 and abs(loss[3].item() - l2.item()) < 1e-5
    assert rel_err(dh, hr.grad) < 1e-5
    # mask loss + mask_pred backward
    fg_max, nfg, Cc = 8, 5, 64
    sc = torch.randn(fg_max * 196, ncls, generator=g); sr = sc.clone().requires_grad_(True)
    mt = (torch.rand(fg_max, 196, generator=g) > 0.5).float()
    labs = torch.randint(1, ncls, (fg_max,), generator=g)
    picked = sr.view(fg_max, 196, ncls)[:nfg].gather(2, labs[:nfg].view(nfg, 1, 1).expand(nfg, 196, 1)).squeeze(2)
    lm = F.binary_cross_entropy_with_logits(picked, mt[:nfg])
    lm.backward()
    loss.zero_()
    dsc = torch.empty(fg_max * 196, device=DEV)
    nf = torch.tensor([nfg], dtype=torch.int32, device=DEV)
    O.mask_loss(sc.to(DEV), ncls, labs.to(torch.int32).to(DEV), mt.to(DEV), nf, fg_max, 196, 1.0, loss, dsc)
    torch.cuda.synchronize()
    assert abs(loss[4].item() - lm.item()) < 1e-5
    dref = sr.grad.view(fg_max, 196, ncls).gather(2, labs.view(fg_max, 1, 1).expand(fg_max, 196, 1)).squeeze(2)
    assert rel_err(dsc.view(fg_max, 196), dref) < 1e-5
    u = F.relu(torch.randn(fg_max * 196, Cc, generator=g)); wm = torch.randn(ncls, Cc, generator=g)
    ur = u.clone().requires_grad_(True); wr = wm.clone().requires_grad_(True); br = torch.zeros(ncls, requires_grad=True)
    (F.linear(ur, wr, br) * sr.grad).sum().backward()
    dx = torch.empty(fg_max * 196, Cc, device=DEV); dw = torch.zeros(ncls, Cc, device=DEV); db = torch.zeros(ncls, device=DEV)
    O.maskpred_bwd(dsc, labs.to(torch.int32).to(DEV), nf, fg_max, 196, Cc, wm.to(DEV), u.to(DEV), u.to(DEV), dx, dw, db)
    torch.cuda.synchronize()
    assert rel_err(dx, ur.grad * (u > 0)) < 1e-5 and rel_err(dw, wr.grad) < 1e-4 and rel_err(db, br.grad) < 1e-4
    loss.zero_(); loss[:6] = torch.tensor([1., 2., 3., 4., 5., 6.])
    O.total_loss(loss, 0.5)
    torch.cuda.synchronize()
    assert abs(loss[6].item() - 18.0) < 1e-6


def test_captioner_projected_attention_step():
    """projected-attention form of one att2in2 step (P = att . W_a2c^T): forward (dots -> softmax + weighted columns of P + gates) and
    backward (gates + dweight = P . da2c; softmax backward + datt_h) against an fp64 torch restatement of AttModel.py:406-466;
    1e-5 relative (fp32, different summation order)."""
    O = ops()
    g = torch.Generator().manual_seed(23)
    L, D, R = 196, 512, 512
    dd = lambda *sh: torch.randn(*sh, generator=g, dtype=torch.float64)
    att = dd(L, R).clamp(min=0); patt = dd(L, D) * 0.5; att_h = dd(D) * 0.5; aw = dd(D) * 0.1; ab = dd(1) * 0.1
    W = dd(2 * R, R) / np.sqrt(R); b = dd(2 * R) * 0.1; sums = dd(5 * R); c_prev = dd(R)
    f = lambda x: x.float().to(DEV).contiguous()
    # reference forward
    th = torch.tanh(patt + att_h); dots = th @ aw + ab; w = torch.softmax(dots, 0); ares = w @ att; a2c = W @ ares + b
    ig, fg, og = torch.sigmoid(sums[:R]), torch.sigmoid(sums[R:2 * R]), torch.sigmoid(sums[2 * R:3 * R])
    t0, t1 = sums[3 * R:4 * R] + a2c[:R], sums[4 * R:] + a2c[R:]
    it = torch.maximum(t0, t1); cn = fg * c_prev + ig * it; hn = og * torch.tanh(cn)
    P = f(att @ W.t())
    tanh_ws = torch.empty(L, D, device=DEV); dts = torch.empty(256, device=DEV); wgt = torch.empty(L, device=DEV)
    c = torch.empty(R, device=DEV); h = torch.empty(R, device=DEV); save = torch.empty(6 * R, device=DEV)
    O.cap_att_dots_fwd(f(patt), f(att_h), f(aw), f(ab), L, D, tanh_ws, dts)
    O.cap_apply_gates_fwd(P, dts, f(b), f(sums), f(c_prev), c, h, save, wgt, L, R)
    torch.cuda.synchronize()
    assert rel_err(wgt, w.float()) < 1e-5 and rel_err(c, cn.float()) < 1e-5 and rel_err(h, hn.float()) < 1e-5
    assert rel_err(tanh_ws, th.float()) < 1e-6
    # backward of the step given dh, dc
    dh, dh2, dc = dd(R) * 0.1, dd(R) * 0.1, dd(R) * 0.1
    tc = torch.tanh(cn); dhj = dh + dh2; dcn = dc + dhj * og * (1 - tc * tc)
    dit = dcn * ig; sel = (t0 >= t1)
    d0 = torch.where(sel, dit, torch.zeros_like(dit)); d1 = torch.where(sel, torch.zeros_like(dit), dit)
    da2c_ref = torch.cat([d0, d1]); dares = W.t() @ da2c_ref; dwl = att @ dares
    ddot_ref = w * (dwl - (w * dwl).sum()); datt_h_ref = ((ddot_ref[:, None] * aw[None, :]) * (1 - th * th)).sum(0)
    dsums_ref = torch.cat([dcn * it * ig * (1 - ig), dcn * c_prev * fg * (1 - fg), dhj * tc * og * (1 - og), d0, d1])
    dsums = torch.empty(5 * R, device=DEV); da2c = torch.empty(2 * R, device=DEV); dcp = torch.empty(R, device=DEV); dw = torch.empty(256, device=DEV)
    ddot = torch.empty(L, device=DEV); datt_h = torch.empty(D + 256, device=DEV)
    O.cap_gates_bwd_dw(f(dh), f(dc), save, f(c_prev), P, dsums, da2c, dcp, dw, L, R, dh2=f(dh2))
    O.cap_attention_bwd_step2(dw, tanh_ws, wgt, f(aw), L, D, ddot, datt_h)
    torch.cuda.synchronize()
    assert rel_err(dsums, dsums_ref.float()) < 1e-5 and rel_err(da2c, da2c_ref.float()) < 1e-5 and rel_err(dcp, (dcn * fg).float()) < 1e-5
    assert rel_err(dw[:L], dwl.float()) < 1e-5 and rel_err(ddot, ddot_ref.float()) < 2e-5 and rel_err(datt_h[:D], datt_h_ref.float()) < 2e-5


def _cap_recurrence_ref(S, L, R, AH, W_h2h, b_h2h, W_att, b_att, patt, aw, ab, P, b_a2c, sums, dho):
    """fp64 autograd restatement of the att2in2 recurrence in projected form (ATT:406-423,446-466) with per-token probes whose gradients are
    the tensors the kernels hand on: d(sums), d(a2c), ddot, d(att_h)."""
    sums = sums.clone().requires_grad_(True)
    B = b_a2c[None].repeat(S, 1).requires_grad_(True)
    Zd = torch.zeros(S, L, dtype=torch.float64, requires_grad=True); Za = torch.zeros(S, AH, dtype=torch.float64, requires_grad=True)
    h = torch.zeros(R, dtype=torch.float64); c = torch.zeros(R, dtype=torch.float64)
    hs, cs, ws, ths = [], [], [], []
    for t in range(S):
        att_h = W_att @ h + b_att + Za[t]
        th = torch.tanh(patt + att_h[None]); dots = th @ aw + ab + Zd[t]; w = torch.softmax(dots, 0)
        a2c = w @ P + B[t]
        s = sums[t] + W_h2h @ h + b_h2h
        ig, fg, og = torch.sigmoid(s[:R]), torch.sigmoid(s[R:2 * R]), torch.sigmoid(s[2 * R:3 * R])
        it = torch.maximum(s[3 * R:4 * R] + a2c[:R], s[4 * R:] + a2c[R:])
        c = fg * c + ig * it; h = og * torch.tanh(c)
        hs.append(h); cs.append(c); ws.append(w); ths.append(th)
    (torch.stack(hs) * dho).sum().backward()
    return (torch.stack(hs).detach(), torch.stack(cs).detach(), torch.stack(ws).detach(), torch.stack(ths).detach(), sums.grad, B.grad, Zd.grad, Za.grad)


@pytest.mark.parametrize('S,L', [(21, 196), (1, 196), (4, 50), (11, 224)])
def test_captioner_resident_recurrence(S, L):
    """csrc/cap_recur.hip: the whole captioner recurrence as one resident launch per direction (32 workgroups, weights in registers, three granule
    exchanges per token) against an fp64 autograd restatement of AttModel.py:406-466 - states, attention weights, tanh values forward; d(sums),
    d(a2c), ddot, d(att_h) backward; 2e-5 relative (fp32, a summation order of its own).  Run three times on the same state buffers (the exchange
    epochs of consecutive launches) with other kernels on a second stream the third time: bit-identical each time, no give-up flag."""
    O = ops()
    R = AH = 512
    g = torch.Generator().manual_seed(31 + S)
    dd = lambda *sh: torch.randn(*sh, generator=g, dtype=torch.float64)
    W_h2h = dd(5 * R, R) / np.sqrt(R); b_h2h = dd(5 * R) * 0.1; W_att = dd(AH, R) / np.sqrt(R); b_att = dd(AH) * 0.1
    patt = dd(L, AH) * 0.5; aw = dd(AH) * 0.1; ab = dd(1) * 0.1; P = dd(L, 2 * R) * 0.5; b_a2c = dd(2 * R) * 0.1
    sums = dd(S, 5 * R) * 0.7; dho = dd(S, R) * 0.1
    assert O.cap_recur_supported(S, R, AH, L) and not O.cap_recur_supported(S, 256, AH, L) and not O.cap_recur_supported(S, R, AH, 225)
    hs_r, cs_r, w_r, th_r, dsums_r, da2c_r, ddot_r, datt_h_r = _cap_recurrence_ref(S, L, R, AH, W_h2h, b_h2h, W_att, b_att, patt, aw, ab, P, b_a2c, sums, dho)
    f = lambda x: x.float().to(DEV).contiguous()
    dv = dict(W_h2h=f(W_h2h), b_h2h=f(b_h2h), W_att=f(W_att), b_att=f(b_att), patt=f(patt), aw=f(aw), ab=f(ab), P=f(P), b_a2c=f(b_a2c), sums=f(sums), dho=f(dho))
    st = (O.cap_recur_state(False), O.cap_recur_state(True))
    side = torch.cuda.Stream()
    big = torch.randn(64 << 20, device=DEV)
    outs = []
    for rep in range(3):
        hs = torch.zeros(S + 1, R, device=DEV); cs = torch.zeros(S + 1, R, device=DEV); save = torch.full((S, 6 * R), float('nan'), device=DEV)
        tanh_ws = torch.full((S, L, AH), float('nan'), device=DEV); wgt = torch.full((S, L), float('nan'), device=DEV)
        dsums = torch.full((S, 5 * R), float('nan'), device=DEV); da2c = torch.full((S, 2 * R), float('nan'), device=DEV)
        ddot = torch.full((S, L), float('nan'), device=DEV); datt_h = torch.full((S, AH + 256), float('nan'), device=DEV)
        torch.cuda.synchronize()
        if rep == 2:                                  # uneven load: HBM-streaming kernels on another queue while the resident launches run
            with torch.cuda.stream(side):
                for _ in range(40):
                    big.mul_(1.0001)
        O.cap_recur_fwd(dv['W_h2h'], dv['b_h2h'], dv['W_att'], dv['b_att'], dv['patt'], dv['aw'], dv['ab'], dv['P'], dv['b_a2c'], dv['sums'], hs, cs, save,
                        tanh_ws, wgt, st[0], S, R, AH, L)
        O.cap_recur_bwd(dv['W_h2h'], dv['W_att'], dv['P'], dv['aw'], save, cs, wgt, tanh_ws, dv['dho'], dsums, da2c, ddot, datt_h, st[1], S, R, AH, L)
        torch.cuda.synchronize()
        assert int(st[0][1].item()) == 0 and int(st[1][1].item()) == 0, 'a bounded spin of the resident recurrence gave up'
        assert int(st[0][0].item()) == rep + 1 and int(st[1][0].item()) == rep + 1
        outs.append([x.clone() for x in (hs, cs, save, tanh_ws, wgt, dsums, da2c, ddot, datt_h[:, :AH])])
    hs, cs, save, tanh_ws, wgt, dsums, da2c, ddot, datt_h = outs[0]
    assert float(hs[0].abs().max()) == 0.0
    assert rel_err(hs[1:], hs_r.float()) < 2e-5 and rel_err(cs[1:], cs_r.float()) < 2e-5 and rel_err(wgt, w_r.float()) < 2e-5
    assert rel_err(tanh_ws, th_r.float()) < 2e-6
    assert rel_err(dsums, dsums_r.float()) < 2e-5 and rel_err(da2c, da2c_r.float()) < 2e-5, (rel_err(dsums, dsums_r.float()), rel_err(da2c, da2c_r.float()))
    assert rel_err(ddot, ddot_r.float()) < 5e-5 and rel_err(datt_h, datt_h_r.float()) < 5e-5, (rel_err(ddot, ddot_r.float()), rel_err(datt_h, datt_h_r.float()))
    for o in outs[1:]:
        for a_, b_ in zip(outs[0], o):
            assert torch.equal(a_, b_), 'the resident recurrence is not bit-reproducible across launches'


def test_row_batch_linears_mfma():
    """row batches (21 tokens, 196 attention locations) of the fp32 linears: exact-fp32 MFMA kernels, forward (bias / activation /
    accumulate, ragged M and N) and data gradient from the weight as stored (split contraction through a workspace, fused mask,
    accumulate); tolerance 1e-5 relative to the fp64 product (fp32 products, fp32 accumulation in a different order than torch's)."""
    O = ops()
    g = torch.Generator().manual_seed(19)
    for (M, N, K, act, acc) in [(21, 3350, 512, 0, False), (196, 512, 512, 0, False), (21, 2560, 512, 0, True), (2, 16, 16, 2, False),
                                (17, 100, 2560, 1, False), (33, 50, 36, 0, True), (196, 512, 4096, 1, False)]:
        x = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / np.sqrt(K); b = torch.randn(N, generator=g)
        y0 = torch.randn(M, N, generator=g)
        pre = (x.double() @ w.double().t() + b.double() + (y0.double() if acc else 0.0))
        ref = pre if act == 0 else (pre.clamp(min=0) if act == 1 else torch.tanh(pre))
        y = y0.to(DEV).clone()
        O.linear_fwd(x.to(DEV), w.to(DEV), b.to(DEV), y, M, N, K, act, accumulate=acc)
        torch.cuda.synchronize()
        assert rel_err(y, ref.float()) < 1e-5, (M, N, K, act, acc)
    for (M, N, K, use_ws, use_mul, acc) in [(21, 3350, 512, True, True, False), (21, 3350, 512, False, False, False), (196, 512, 512, True, False, True),
                                            (5, 70, 36, True, True, True), (21, 2560, 512, True, False, False), (40, 130, 200, False, True, False)]:
        dy = torch.randn(M, N, generator=g); w = torch.randn(N, K, generator=g) / np.sqrt(N)
        mul = (torch.rand(M, K, generator=g) > 0.5).float() * 2 if use_mul else None
        dx0 = torch.randn(M, K, generator=g)
        ref = dy.double() @ w.double()
        if mul is not None:
            ref = ref * mul.double()
        if acc:
            ref = ref + dx0.double()
        dx = dx0.to(DEV).clone()
        nws = O.linear_bwd_x_ws_floats(M, N, K)
        ws = torch.full((max(nws, 1),), float('nan'), device=DEV) if (use_ws and nws) else None
        O.linear_bwd_x(dy.to(DEV), w.to(DEV), dx, M, N, K, accumulate=acc, mul=None if mul is None else mul.to(DEV), ws=ws)
        torch.cuda.synchronize()
        assert rel_err(dx, ref.float()) < 1e-5, (M, N, K, use_ws, use_mul, acc)
        dx2 = dx0.to(DEV).clone()                              # bit-identical when repeated (no atomics)
        O.linear_bwd_x(dy.to(DEV), w.to(DEV), dx2, M, N, K, accumulate=acc, mul=None if mul is None else mul.to(DEV), ws=ws)
        torch.cuda.synchronize()
        assert torch.equal(dx, dx2)
    # a single row of a large matrix (the dynamic-filter layer: 7175 x 1024) also takes the split NN path
    dy = torch.randn(1, 7175, generator=g); w = torch.randn(7175, 1024, generator=g) / 80.0
    dx = torch.empty(1, 1024, device=DEV); ws = torch.full((O.linear_bwd_x_ws_floats(1, 7175, 1024),), float('nan'), device=DEV)
    O.linear_bwd_x(dy.to(DEV), w.to(DEV), dx, 1, 7175, 1024, ws=ws)
    torch.cuda.synchronize()
    assert rel_err(dx, (dy.double() @ w.double()).float()) < 1e-5
    # weight gradient of a row batch (TN): dw += dy^T x, db += column sums of dy; ragged N, dy rows that are only 8-byte aligned (3350)
    for (M, N, K, lddy) in [(21, 3350, 512, None), (196, 512, 512, None), (196, 1024, 512, None), (21, 196, 1024, None), (7, 100, 36, None), (21, 512, 512, 768),
                            (2, 64, 64, None)]:
        ld = N if lddy is None else lddy
        dyf = torch.randn(M, ld, generator=g); x = torch.randn(M, K, generator=g)
        dw0 = torch.randn(N, K, generator=g); db0 = torch.randn(N, generator=g)
        refw = dw0.double() + dyf[:, :N].double().t() @ x.double(); refb = db0.double() + dyf[:, :N].double().sum(0)
        dw = dw0.to(DEV).clone(); db = db0.to(DEV).clone()
        O.linear_bwd_w(dyf.to(DEV), x.to(DEV), dw, db, M, N, K, lddy=ld)
        torch.cuda.synchronize()
        assert rel_err(dw, refw.float()) < 1e-5 and rel_err(db, refb.float()) < 1e-5, (M, N, K, lddy)
        dw2 = dw0.to(DEV).clone()
        O.linear_bwd_w(dyf.to(DEV), x.to(DEV), dw2, None, M, N, K, lddy=ld)
        torch.cuda.synchronize()
        assert torch.equal(dw, dw2)


def test_linear_embed_lstm():
    O = ops()
    g = torch.Generator().manual_seed(9)
    for (M, N, K, act) in [(1, 2048, 512, 0), (21, 3350, 512, 0), (7, 512, 512, 1), (1, 7175, 1024, 2), (30, 64, 128, 0)]:
        x = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / np.sqrt(K); b = torch.randn(N, generator=g)
        xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
        pre = F.linear(xr, wr, br)
        yref = pre if act == 0 else (F.relu(pre) if act == 1 else torch.tanh(pre))
        y = torch.empty(M, N, device=DEV)
        O.linear_fwd(x.to(DEV), w.to(DEV), b.to(DEV), y, M, N, K, act)
        torch.cuda.synchronize()
        assert rel_err(y, yref) < 1e-5
        dy = torch.randn(M, N, generator=g)
        yref.backward(dy)
        dyd = dy.to(DEV).clone()
        if act:
            O.act_bwd(dyd, y, act)
        dx = torch.empty(M, K, device=DEV)
        O.linear_bwd_x(dyd, w.to(DEV), dx, M, N, K)
        dw = torch.zeros(N, K, device=DEV); db = torch.zeros(N, device=DEV)
        O.linear_bwd_w(dyd, x.to(DEV), dw, db, M, N, K)
        torch.cuda.synchronize()
        assert rel_err(dx, xr.grad) < 1e-4 and rel_err(dw, wr.grad) < 1e-4 and rel_err(db, br.grad) < 1e-4
    # embedding (+relu, mask) fwd/bwd with a repeated token
    V, D, T = 50, 512, 6
    tab = torch.randn(V, D, generator=g); ids = torch.tensor([3, 7, 3, 0, 49, 7])
    mask = (torch.rand(T, D, generator=g) > 0.5).float() * 2
    tr = tab.clone().requires_grad_(True)
    ref = F.relu(tr[ids]) * mask
    out = torch.empty(T, D, device=DEV)
    O.embed_fwd(tab.to(DEV), ids.to(DEV), mask.to(DEV), out, T, D, True)
    dout = torch.randn(T, D, generator=g)
    ref.backward(dout)
    dtab = torch.zeros(V, D, device=DEV)
    O.embed_bwd(dout.to(DEV), out, ids.to(DEV), mask.to(DEV), dtab, T, D, True)
    torch.cuda.synchronize()
    assert rel_err(out, ref) < 1e-6 and rel_err(dtab, tr.grad) < 1e-5
    # LSTM cell
    Hh = 512
    gates = torch.randn(4 * Hh, generator=g); c0 = torch.randn(Hh, generator=g)
    gr = gates.clone().requires_grad_(True); cr = c0.clone().requires_grad_(True)
    i, f, gg, o = torch.sigmoid(gr[:Hh]), torch.sigmoid(gr[Hh:2 * Hh]), torch.tanh(gr[2 * Hh:3 * Hh]), torch.sigmoid(gr[3 * Hh:])
    c1 = f * cr + i * gg; h1 = o * torch.tanh(c1)
    dh = torch.randn(Hh, generator=g); dc = torch.randn(Hh, generator=g)
    (h1 * dh + c1 * dc).sum().backward()
    c = torch.empty(Hh, device=DEV); h = torch.empty(Hh, device=DEV); act = torch.empty(4 * Hh, device=DEV)
    O.lstm_cell_fwd(gates.to(DEV), c0.to(DEV), c, h, act, Hh)
    dg = torch.empty(4 * Hh, device=DEV); dcp = torch.empty(Hh, device=DEV)
    O.lstm_cell_bwd(dh.to(DEV), dc.to(DEV), act, c0.to(DEV), c, dg, dcp, Hh)
    torch.cuda.synchronize()
    assert rel_err(h, h1) < 1e-5 and rel_err(c, c1) < 1e-5 and rel_err(dg, gr.grad) < 1e-4 and rel_err(dcp, cr.grad) < 1e-4


@pytest.mark.parametrize('dt', [0, 1])
def test_dynfilter(dt):
    O = ops()
    g = torch.Generator().manual_seed(10)
    H, W, Cc = 13, 17, 256
    x = F.relu(torch.randn(1, Cc, H, W, generator=g))
    filt = torch.tanh(torch.randn(7, Cc, generator=g)); r = torch.tanh(torch.randn(7, generator=g))
    xd = to_dev(nhwc(x), dt)
    xr = xd.float().cpu().permute(0, 3, 1, 2).clone().requires_grad_(True)
    fr = filt.clone().requires_grad_(True); rr = r.clone().requires_grad_(True)
    masks = torch.from_numpy(ON.spatial_masks(H, W))
    resp = [F.conv2d(xr * masks[k][None, None], fr[k].view(1, -1, 1, 1)) for k in range(7)]
    response = F.conv2d(torch.cat(resp, 1), rr.view(1, 7, 1, 1))
    yref = xr * response
    y = O.empty((H * W, Cc), dt); rs_ = torch.empty(H * W, device=DEV); rk = torch.empty(H * W, 7, device=DEV)
    O.dynfilter_fwd(xd, filt.to(DEV), r.to(DEV), y, rs_, rk, H, W, Cc)
    torch.cuda.synchronize()
    assert rel_err(y.float().view(1, H, W, Cc), nhwc(yref)) < (2e-5 if dt == 0 else 1e-2)
    assert rel_err(rs_.view(H, W), response[0, 0]) < 1e-5
    dy = to_dev(torch.randn(H * W, Cc, generator=g), dt)
    yref.backward(dy.float().cpu().view(1, H, W, Cc).permute(0, 3, 1, 2))
    dx = O.empty((H * W, Cc), dt); dfilt = torch.zeros(7, Cc, device=DEV); dr = torch.zeros(7, device=DEV); wsd = torch.empty(O.dynfilter_ws_floats(H, W, Cc), device=DEV)
    O.dynfilter_bwd(dy, xd, filt.to(DEV), r.to(DEV), rs_, rk, dx, xd, dfilt, dr, wsd, H, W, Cc)
    torch.cuda.synchronize()
    tol = 1e-4 if dt == 0 else 2e-2
    assert rel_err(dx.float().view(1, H, W, Cc), nhwc(xr.grad * (xr > 0))) < tol
    assert rel_err(dfilt, fr.grad) < tol and rel_err(dr, rr.grad) < tol
    # split form used by the train step: dx on the caller's stream, dfilt / dr finished later from the workspace -- bit-identical
    dx2 = O.empty((H * W, Cc), dt); dfilt2 = torch.zeros(7, Cc, device=DEV); dr2 = torch.zeros(7, device=DEV)
    O.dynfilter_bwd(dy, xd, filt.to(DEV), r.to(DEV), rs_, rk, dx2, xd, None, None, wsd, H, W, Cc)
    O.dynfilter_bwd_finish(wsd, rk, dfilt2, dr2, H, W, Cc)
    torch.cuda.synchronize()
    assert torch.equal(dx2.float(), dx.float()) and torch.equal(dfilt2, dfilt) and torch.equal(dr2, dr)


def test_captioner_pieces():
    O = ops()
    g = torch.Generator().manual_seed(12)
    L, D = 196, 512
    patt = torch.randn(L, D, generator=g); att = torch.randn(L, D, generator=g); ah = torch.randn(D, generator=g)
    aw = torch.randn(D, generator=g) * 0.1; ab = torch.randn(1, generator=g)
    pr, ar, hr, wr, br = [t.clone().requires_grad_(True) for t in (patt, att, ah, aw, ab)]
    dot = torch.tanh(pr + hr) @ wr + br
    wgt = F.softmax(dot, 0); res = wgt @ ar
    dres = torch.randn(D, generator=g)
    (res * dres).sum().backward()
    tws = torch.empty(L, D, device=DEV); wd = torch.empty(L, device=DEV); rd = torch.empty(D + 256, device=DEV)
    O.cap_attention_fwd(patt.to(DEV), att.to(DEV), ah.to(DEV), aw.to(DEV), ab.to(DEV), L, D, tws, wd, rd)
    dpatt = torch.zeros(L, D, device=DEV); datt = torch.zeros(L, D, device=DEV); dah = torch.empty(D + 256, device=DEV)
    daw = torch.zeros(D, device=DEV); dab = torch.zeros(1, device=DEV)
    O.cap_attention_bwd(dres.to(DEV), att.to(DEV), tws, wd, aw.to(DEV), L, D, dpatt, datt, dah, daw, dab)
    torch.cuda.synchronize()
    assert rel_err(wd, wgt) < 1e-5 and rel_err(rd[:D], res) < 1e-5
    assert rel_err(dpatt, pr.grad) < 1e-4 and rel_err(datt, ar.grad) < 1e-4 and rel_err(dah[:D], hr.grad) < 1e-4
    assert rel_err(daw, wr.grad) < 1e-4 and abs(dab.item() - br.grad.item()) < 1e-5
    # the split form used inside the recurrence: per-step (ddot, datt_h) + one batched launch for the step sums
    S2 = 3
    dres2 = torch.randn(S2, D, generator=g); ah2 = torch.randn(S2, D, generator=g)
    pr, ar, wr, br = [t.clone().requires_grad_(True) for t in (patt, att, aw, ab)]
    hr2 = ah2.clone().requires_grad_(True)
    tot = 0
    for i in range(S2):
        wgt_i = F.softmax(torch.tanh(pr + hr2[i]) @ wr + br, 0)
        tot = tot + ((wgt_i @ ar) * dres2[i]).sum()
    tot.backward()
    tws2 = torch.empty(S2, L, D, device=DEV); wd2 = torch.empty(S2, L, device=DEV); rd2 = torch.empty(S2, D + 256, device=DEV)
    ddot = torch.empty(S2, L, device=DEV); dah2 = torch.empty(S2, D + 256, device=DEV); dr2 = dres2.to(DEV)
    for i in range(S2):
        O.cap_attention_fwd(patt.to(DEV), att.to(DEV), ah2[i].to(DEV), aw.to(DEV), ab.to(DEV), L, D, tws2[i], wd2[i], rd2[i])
        O.cap_attention_bwd_step(dr2[i], att.to(DEV), tws2[i], wd2[i], aw.to(DEV), L, D, ddot[i], dah2[i])
    dpatt = torch.zeros(L, D, device=DEV); datt = torch.zeros(L, D, device=DEV); daw = torch.zeros(D, device=DEV); dab = torch.zeros(1, device=DEV)
    O.cap_attention_bwd_batched(ddot, wd2, dr2, D, tws2, aw.to(DEV), S2, L, D, dpatt, datt, daw, dab)
    torch.cuda.synchronize()
    assert rel_err(dah2[:, :D], hr2.grad) < 1e-4 and rel_err(dpatt, pr.grad) < 1e-4 and rel_err(datt, ar.grad) < 1e-4
    assert rel_err(daw, wr.grad) < 1e-4 and abs(dab.item() - br.grad.item()) < 1e-4
    # gates (maxout candidate, AttModel.py:449-462)
    R = 512
    s = torch.randn(5 * R, generator=g); a2c = torch.randn(2 * R, generator=g); c0 = torch.randn(R, generator=g)
    sr, a2r, cr = [t.clone().requires_grad_(True) for t in (s, a2c, c0)]
    sg = torch.sigmoid(sr[:3 * R]); it = sr[3 * R:] + a2r; it = torch.max(it[:R], it[R:])
    c1 = sg[R:2 * R] * cr + sg[:R] * it; h1 = sg[2 * R:] * torch.tanh(c1)
    dh = torch.randn(R, generator=g); dc = torch.randn(R, generator=g)
    (h1 * dh + c1 * dc).sum().backward()
    c = torch.empty(R, device=DEV); h = torch.empty(R, device=DEV); save = torch.empty(6 * R, device=DEV)
    O.cap_gates_fwd(s.to(DEV), a2c.to(DEV), c0.to(DEV), c, h, save, R)
    ds = torch.empty(5 * R, device=DEV); da = torch.empty(2 * R, device=DEV); dcp = torch.empty(R, device=DEV)
    O.cap_gates_bwd(dh.to(DEV), dc.to(DEV), save, c0.to(DEV), ds, da, dcp, R)
    torch.cuda.synchronize()
    assert rel_err(h, h1) < 1e-5 and rel_err(c, c1) < 1e-5
    assert rel_err(ds, sr.grad) < 1e-4 and rel_err(da, a2r.grad) < 1e-4 and rel_err(dcp, cr.grad) < 1e-4
    # dh given as two addends
    ds2 = torch.empty(5 * R, device=DEV); da2 = torch.empty(2 * R, device=DEV); dcp2 = torch.empty(R, device=DEV)
    O.cap_gates_bwd((dh * 0.25).to(DEV), dc.to(DEV), save, c0.to(DEV), ds2, da2, dcp2, R, dh2=(dh * 0.75).to(DEV))
    torch.cuda.synchronize()
    assert rel_err(ds2, sr.grad) < 1e-4 and rel_err(dcp2, cr.grad) < 1e-4
    # a2c Linear fused with the gates; two GEMVs from one input; two-input GEMV
    K = 512
    ares = torch.randn(K, generator=g); wa = torch.randn(2 * R, K, generator=g) / 22; ba = torch.randn(2 * R, generator=g)
    a2c_ref = wa @ ares + ba
    itr = s[3 * R:] + a2c_ref; itr = torch.max(itr[:R], itr[R:]); sgr = torch.sigmoid(s[:3 * R])
    c1r = sgr[R:2 * R] * c0 + sgr[:R] * itr; h1r = sgr[2 * R:] * torch.tanh(c1r)
    c = torch.empty(R, device=DEV); h = torch.empty(R, device=DEV); save2 = torch.empty(6 * R, device=DEV)
    O.cap_a2c_gates_fwd(ares.to(DEV), wa.to(DEV), ba.to(DEV), K, s.to(DEV), c0.to(DEV), c, h, save2, R)
    x = torch.randn(K, generator=g); w1 = torch.randn(300, K, generator=g) / 22; b1 = torch.randn(300, generator=g)
    w2 = torch.randn(1030, K, generator=g) / 22; b2 = torch.randn(1030, generator=g); y2_0 = torch.randn(1030, generator=g)
    y1 = torch.empty(300, device=DEV); y2 = y2_0.to(DEV)
    O.linear2_fwd(x.to(DEV), K, w1.to(DEV), b1.to(DEV), y1, 300, False, w2.to(DEV), b2.to(DEV), y2, 1030, True)
    x2 = torch.randn(256, generator=g); w3 = torch.randn(300, 256, generator=g) / 16
    y3 = torch.empty(300, device=DEV)
    O.linear_sum2_fwd(x.to(DEV), w1.to(DEV), K, x2.to(DEV), w3.to(DEV), 256, y3, 300)
    torch.cuda.synchronize()
    assert rel_err(h, h1r) < 1e-5 and rel_err(c, c1r) < 1e-5
    assert rel_err(y1, w1 @ x + b1) < 1e-5 and rel_err(y2, w2 @ x + b2 + y2_0) < 1e-5 and rel_err(y3, w1 @ x + w3 @ x2) < 1e-5
    # log-softmax + masked NLL
    S, V1 = 7, 1200
    lg = torch.randn(S, V1, generator=g) * 3; tgt = torch.randint(0, V1, (S,), generator=g); msk = torch.tensor([1, 1, 1, 1, 1, 0.0, 1])
    lr_ = lg.clone().requires_grad_(True)
    lp = F.log_softmax(lr_, 1)
    lossr = (-lp.gather(1, tgt.view(-1, 1)).squeeze(1) * msk).sum() / msk.sum()
    (lossr * 0.5).backward()
    slot = torch.zeros(1, device=DEV); dl = torch.empty(S, V1, device=DEV); lpo = torch.empty(S, V1, device=DEV)
    O.logsoftmax_nll(lg.to(DEV), tgt.to(DEV), msk.to(DEV), S, V1, 0.5, slot, dl, lpo)
    torch.cuda.synchronize()
    assert abs(slot.item() - lossr.item()) < 1e-5 and rel_err(dl, lr_.grad) < 1e-5 and rel_err(lpo, lp) < 1e-5


def test_sgd_and_misc():
    O = ops()
    import ctypes as C
    from lang2seg_amd._lib import SgdSeg
    g = torch.Generator().manual_seed(13)
    n1, rows, rl, n3 = 1000, 8, 96, 77
    tot = n1 + rows * rl + n3
    p = torch.randn(tot, generator=g); gr = torch.randn(tot, generator=g); m = torch.randn(tot, generator=g)
    rowscale = torch.rand(rows, generator=g) + 0.5
    segs = (SgdSeg * 3)()
    segs[0].offset, segs[0].count, segs[0].row_len, segs[0].weight_decay, segs[0].rowscale_off, segs[0].lr_mult = 0, n1, 1, 1, -1, 1.0
    segs[1].offset, segs[1].count, segs[1].row_len, segs[1].weight_decay, segs[1].rowscale_off, segs[1].lr_mult = n1, rows * rl, rl, 1, 0, 1.0
    segs[2].offset, segs[2].count, segs[2].row_len, segs[2].weight_decay, segs[2].rowscale_off, segs[2].lr_mult = n1 + rows * rl, n3, 1, 0, -1, 2.0
    run = 0
    for sg in segs:                                  # l2s_sgd_seg.chunk0: running count of the segments' work chunks (include/lang2seg_hip.h)
        sg.chunk0 = run; run += -(-int(sg.count) // O.sgd_chunk())
    sb = torch.frombuffer(bytearray(bytes(segs)), dtype=torch.uint8).to(DEV)
    pd, gd, md = p.to(DEV), gr.to(DEV), m.to(DEV)
    shadow = torch.empty(tot, dtype=torch.bfloat16, device=DEV)
    O.sgd_momentum(pd, gd, md, sb, 3, rowscale.to(DEV), 0.01, 0.9, 1e-2, shadow=shadow)
    torch.cuda.synchronize()
    assert torch.equal(gd.cpu(), gr)                 # clear_grad = 0: the gradients are left alone
    # the same update on the last two segments only, with the gradient clear folded in (the table may start at any segment)
    p2, g2, m2 = p.to(DEV), gr.to(DEV), m.to(DEV)
    O.sgd_momentum(p2, g2, m2, sb[C.sizeof(SgdSeg):], 2, rowscale.to(DEV), 0.01, 0.9, 1e-2, shadow=None, clear_grad=True)
    torch.cuda.synchronize()
    assert torch.equal(p2[n1:], pd[n1:]) and torch.equal(m2[n1:], md[n1:]) and torch.equal(p2[:n1].cpu(), p[:n1])
    assert float(g2[n1:].abs().max()) == 0.0 and torch.equal(g2[:n1].cpu(), gr[:n1])
    ge = gr.clone(); ge[n1:n1 + rows * rl] = (ge[n1:n1 + rows * rl].view(rows, rl) * rowscale.view(-1, 1)).view(-1)
    wd = torch.ones(tot) * 1e-2; wd[n1 + rows * rl:] = 0
    lr = torch.ones(tot) * 0.01; lr[n1 + rows * rl:] = 0.02
    mref = 0.9 * m + ge + wd * p; pref = p - lr * mref
    torch.cuda.synchronize()
    assert rel_err(md, mref) < 1e-6 and rel_err(pd, pref) < 1e-6
    sref = pref.clone(); sref[n1:n1 + rows * rl] = (sref[n1:n1 + rows * rl].view(rows, rl) * rowscale.view(-1, 1)).view(-1)
    assert rel_err(shadow.float(), sref) < 1e-2
    # l2s_sgd_momentum_range: the update cut into two "rank" slices at an arbitrary multiple of four elements (+ a tail every rank updates) gives
    # the single launch's result bit for bit (data parallel, sharded update); flags = 2 rewrites the shadow only
    from lang2seg_amd._lib import load as _load
    CH = O.sgd_chunk()
    offs = [0, n1, n1 + rows * rl]; cnts = [n1, rows * rl, n3]; ch0 = [0, -(-n1 // CH), -(-n1 // CH) + -(-(rows * rl) // CH)]
    def chunk_range(lo, hi):
        import bisect
        ends = [o + c for o, c in zip(offs, cnts)]
        s0 = bisect.bisect_right(ends, lo); s1 = bisect.bisect_left(offs, hi)
        if s0 >= s1:
            return 0, 0
        return ch0[s0] + max(0, lo - offs[s0]) // CH, ch0[s1 - 1] + -(-(min(hi, ends[s1 - 1]) - offs[s1 - 1]) // CH)
    p3, g3, m3 = p.to(DEV), gr.to(DEV), m.to(DEV)
    sh3 = torch.zeros(tot, dtype=torch.bfloat16, device=DEV)
    cut, tail0 = 1284, tot - 5
    for lo, hi in ((cut, tail0), (0, cut), (tail0, tot)):
        c_lo, c_hi = chunk_range(lo, hi)
        O.sgd_momentum_range(p3, g3, m3, sb, 3, rowscale.to(DEV), 0.01, 0.9, 1e-2, 1.0, None, 0, lo, hi, c_lo, c_hi)
    O.sgd_momentum_range(p3, g3, m3, sb, 3, rowscale.to(DEV), 0.0, 1.0, 0.0, 0.0, sh3, 2, 0, tot, 0, -1)
    torch.cuda.synchronize()
    assert torch.equal(p3, pd) and torch.equal(m3, md) and torch.equal(sh3, shadow) and torch.equal(g3.cpu(), gr)
    # dropout mask statistics + determinism, random keys
    ctr = torch.zeros(1, dtype=torch.int64, device=DEV)
    mk = torch.empty(100000, device=DEV); O.dropout_mask(mk, 0.5, ctr, 42)
    mk2 = torch.empty(100000, device=DEV); O.dropout_mask(mk2, 0.5, ctr, 42)
    O.counter_inc(ctr)
    mk3 = torch.empty(100000, device=DEV); O.dropout_mask(mk3, 0.5, ctr, 42)
    torch.cuda.synchronize()
    assert int(ctr.item()) == 1 and not torch.equal(mk, mk3)
    assert torch.equal(mk, mk2) and abs((mk > 0).float().mean().item() - 0.5) < 0.01 and set(mk.unique().tolist()) == {0.0, 2.0}
    a = torch.randn(1000, generator=g); b = torch.randn(1000, generator=g); c = torch.randn(1000, generator=g)
    out = torch.empty(1000, device=DEV); O.add3(a.to(DEV), b.to(DEV), c.to(DEV), out)
    ob = torch.empty(1000, dtype=torch.bfloat16, device=DEV); O.cast(out, ob)
    torch.cuda.synchronize()
    assert rel_err(out, a + b + c) < 1e-6 and rel_err(ob.float(), (a + b + c)) < 1e-2


@pytest.mark.parametrize('dt', [0, 1])
def test_vgg_kernels(dt):
    """conv1_1 (3 -> 64), 2x2 max pooling forward / backward (odd sizes, first-maximum rule, ReLU mask), dropout scale-mask."""
    O = ops()
    g = torch.Generator().manual_seed(21)
    H, W = 37, 45
    img = torch.randn(1, 3, H, W, generator=g) * 30
    w = torch.randn(64, 3, 3, 3, generator=g) * 0.02; b = torch.randn(64, generator=g) * 0.1
    ref = F.relu(F.conv2d(img, w, b, padding=1))
    y = O.empty((H * W, 64), dt)
    O.conv3x3_c3(nhwc(img).contiguous().to(DEV), ohwi(w).to(DEV), b.to(DEV), y, H, W)
    torch.cuda.synchronize()
    assert rel_err(y.float().view(1, H, W, 64), nhwc(ref)) < (1e-5 if dt == 0 else 1e-2)
    # max pool: quantised inputs so that ties occur and are resolved like ATen (first maximum in scan order)
    C, n = 24, 3
    x = (torch.randint(-2, 3, (n, C, H, W), generator=g).float() * 0.5)
    xd = to_dev(nhwc(x), dt)
    xr = xd.float().cpu().permute(0, 3, 1, 2).clone().requires_grad_(True)
    pr = F.max_pool2d(xr, 2, 2)
    OH, OW = H // 2, W // 2
    yd = O.empty((n * OH * OW, C), dt)
    O.maxpool2x2_fwd(xd, yd, n, H, W, C)
    torch.cuda.synchronize()
    assert torch.equal(yd.float().cpu().view(n, OH, OW, C), nhwc(pr.detach()))
    dy = torch.randn(n, C, OH, OW, generator=g)
    dyd = to_dev(nhwc(dy), dt)
    pr.backward(dyd.float().cpu().permute(0, 3, 1, 2))
    dxd = torch.full((n * H * W, C), 7.0, device=DEV).to(O.TORCH_DT[dt])
    O.maxpool2x2_bwd(dyd, xd, dxd, n, H, W, C, False)
    torch.cuda.synchronize()
    assert torch.equal(dxd.float().cpu().view(n, H, W, C), nhwc(xr.grad))
    O.maxpool2x2_bwd(dyd, xd, dxd, n, H, W, C, True)          # x taken as a ReLU output: no gradient where the maximum is <= 0
    torch.cuda.synchronize()
    pooled = F.max_pool2d(xd.float().cpu().permute(0, 3, 1, 2), 2, 2)
    up = F.interpolate((pooled > 0).float(), scale_factor=2, mode='nearest')
    exp = xr.grad.clone(); exp[:, :, :2 * OH, :2 * OW] *= up
    assert torch.equal(dxd.float().cpu().view(n, H, W, C), nhwc(exp))
    # dropout scale / mask
    v = torch.randn(500, generator=g); m = (torch.rand(500, generator=g) > 0.5).float() * 2; r = torch.randn(500, generator=g)
    vd, rd = to_dev(v, dt), to_dev(r, dt)
    out = O.empty((500,), dt)
    O.scale_mask(vd, m.to(DEV), rd, out)
    torch.cuda.synchronize()
    exp = vd.float().cpu() * m * (rd.float().cpu() > 0).float()
    assert rel_err(out.float(), exp) < (1e-6 if dt == 0 else 1e-2)


def test_response_loss_and_test_heads():
    O = ops()
    from oracle import boxes as OB
    g = torch.Generator().manual_seed(22)
    MH, MW, H, W = 320, 416, 20, 26
    mask = (torch.rand(MH, MW, generator=g) > 0.6).to(torch.uint8)
    resp = torch.randn(H, W, generator=g)
    tgt = torch.from_numpy(OB.imresize_nearest_u8(mask.numpy(), (H, W)).astype(np.float32))
    rr = resp.clone().requires_grad_(True)
    l = F.binary_cross_entropy_with_logits(rr, tgt); (l * 0.7).backward()
    loss = torch.zeros(8, device=DEV); dresp = torch.empty(H * W, device=DEV)
    O.response_loss(resp.to(DEV).view(-1), mask.to(DEV), MH, MW, H, W, 0.7, loss, dresp)
    torch.cuda.synchronize()
    assert abs(loss[7].item() - l.item()) < 1e-6 and rel_err(dresp.view(H, W), rr.grad) < 1e-5
    # TEST-mode heads
    R, nc = 37, 81
    heads = torch.randn(R, 408, generator=g)
    stds = torch.tensor([0.1, 0.1, 0.2, 0.2]); means = torch.tensor([0.0, 0.01, 0.0, -0.02])
    cp = torch.empty(R, nc, device=DEV); bp = torch.empty(R, 4 * nc, device=DEV)
    O.rcnn_predict(heads.to(DEV), 408, R, nc, stds.to(DEV), means.to(DEV), cp, bp)
    sc = torch.randn(5 * 196, nc, generator=g); lab = torch.tensor([3, 80, 0, 17, 42], dtype=torch.int32)
    mp_all = torch.empty(5 * 196, nc, device=DEV); mp_l = torch.empty(5 * 196, device=DEV)
    O.mask_prob(sc.to(DEV), nc, nc, None, 196, 5 * 196, mp_all)
    O.mask_prob(sc.to(DEV), nc, nc, lab.to(DEV), 196, 5 * 196, mp_l)
    torch.cuda.synchronize()
    assert rel_err(cp, F.softmax(heads[:, :nc], 1)) < 1e-6
    assert rel_err(bp, heads[:, nc:5 * nc] * stds.repeat(nc) + means.repeat(nc)) < 1e-6
    assert rel_err(mp_all, torch.sigmoid(sc)) < 1e-6
    assert rel_err(mp_l, torch.sigmoid(sc.view(5, 196, nc)[torch.arange(5), :, lab.long()]).reshape(-1)) < 1e-6


@pytest.mark.parametrize('dt', [0, 1])
def test_roipool(dt):
    """RoI max pooling (POOLING_MODE == 'pool') vs the numpy restatement of roi_pooling_kernel.cu: values and argmax exact."""
    O = ops()
    g = torch.Generator().manual_seed(31)
    H, W, Cc, R, P = 20, 26, 40, 11, 7
    feat = torch.randint(-8, 9, (Cc, H, W), generator=g).float() * 0.25      # ties on purpose
    rs = np.random.RandomState(4)
    rois = np.zeros((R, 5), np.float32)
    rois[:, 1] = rs.uniform(0, 300, R); rois[:, 2] = rs.uniform(0, 200, R)
    rois[:, 3] = np.minimum(rois[:, 1] + rs.uniform(2, 250, R), 415); rois[:, 4] = np.minimum(rois[:, 2] + rs.uniform(2, 250, R), 319)
    rois[0, 1:] = [0, 0, 415, 319]; rois[1, 1:] = [100.0, 90.0, 103.0, 92.0]            # full image, tiny roi (bins share pixels)
    rois[2, 1:] = [400.0, 300.0, 500.0, 400.0]                                           # partly outside: empty bins
    fd = to_dev(feat.permute(1, 2, 0).contiguous().view(H * W, Cc), dt)
    ref, arg = OB.roi_pool_fwd(fd.float().cpu().view(H, W, Cc).permute(2, 0, 1).numpy(), rois, P, 1.0 / 16.0)
    out = O.empty((R * P * P, Cc), dt); am = torch.empty((R * P * P, Cc), dtype=torch.int32, device=DEV)
    O.roipool_fwd(fd, H, W, Cc, torch.from_numpy(rois).to(DEV), R, P, 1.0 / 16.0, out, am)
    torch.cuda.synchronize()
    assert np.array_equal(out.float().cpu().view(R, P, P, Cc).permute(0, 3, 1, 2).numpy(), ref)
    assert np.array_equal(am.cpu().view(R, P, P, Cc).permute(0, 3, 1, 2).numpy(), arg)
    dout = to_dev(torch.randn(R * P * P, Cc, generator=g), dt)
    dfeat = torch.zeros(H * W, Cc, device=DEV)
    O.roipool_bwd(dout, am, R, P, Cc, dfeat)
    torch.cuda.synchronize()
    dref = OB.roi_pool_bwd(dout.float().cpu().view(R, P, P, Cc).permute(0, 3, 1, 2).numpy(), arg, H, W)
    assert rel_err(dfeat.view(H, W, Cc).permute(2, 0, 1), torch.from_numpy(dref)) < 1e-5


def test_tape_timing_events():
    """timing events recorded as tape ops: outside a recording the call is refused (-1); inside, every replay records the pair again on
    the stream and the elapsed time brackets the launches between them (here: fills of 64 MB vs of 1 MB)."""
    O = ops()
    s = torch.cuda.current_stream()
    big = torch.empty(16 << 20, dtype=torch.float32, device=DEV)
    small = torch.empty(1 << 18, dtype=torch.float32, device=DEV)
    assert O.tape_time_event() == -1
    h = O.tape_begin([s])
    a = O.tape_time_event(); O.memset_zero(big); b = O.tape_time_event()
    O.memset_zero(small); c = O.tape_time_event()
    O.tape_end(h)
    assert a >= 0 and b == a + 1 and c == b + 1
    torch.cuda.synchronize()
    for _ in range(3):
        big.fill_(1.0); small.fill_(1.0)
        O.tape_run(h, [s])
        torch.cuda.synchronize()
        t_big, t_small = O.time_event_elapsed(a, b), O.time_event_elapsed(b, c)
        assert float(big.abs().sum()) == 0.0 and float(small.abs().sum()) == 0.0
        assert 0.0 < t_small < t_big < 5.0, (t_small, t_big)        # 64 MB at >= 1 TB/s is well under a millisecond
        assert t_big > 0.008                                         # and cannot beat 8 TB/s
    O.tape_destroy(h)
    with pytest.raises(Exception):
        O.time_event_elapsed(a, 1 << 20)
