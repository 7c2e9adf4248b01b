"""ivln_gn_conv_f32 (csrc/gn_conv.hip): GroupNorm (+ second operand / residual) (+ ReLU) (+ MaxPool) of slab input fused
with the NEXT convolution(s), written as per-group partial slabs - against F.group_norm / F.conv2d (fp32, torch CPU)
on every layer shape of the DD-PPO depth ResNet (habitat-lab ResNetEncoder, restated in oracle/habitat_ext_ref.py),
and the whole encoder chained through it against the deferred conv + GroupNorm path and the oracle."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _slabs(t, splits, g):
    """NCHW tensor -> `splits` random slabs of its [C][N*HW] matrix that sum to it (a Deferred)."""
    from ivln_ce_amd import ops

    N, Cc, H, W = t.shape
    m = t.permute(1, 0, 2, 3).reshape(Cc, N * H * W)
    parts = [torch.randn(m.shape, generator=g) for _ in range(splits - 1)]
    parts.append(m - sum(parts) if parts else m)
    ws = torch.stack(parts).contiguous().view(-1).to(DEV)
    return ops.Deferred(ws, splits, N, Cc, H, W)


def _sum(d):
    """Deferred slabs -> NCHW (CPU)"""
    m = d.ws.view(d.splits, d.C, d.N, d.H, d.W).sum(0)
    return m.permute(1, 0, 2, 3).cpu()


def _gn(Cc, groups, g):
    m = torch.nn.GroupNorm(groups, Cc)
    m.weight.data, m.bias.data = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)
    return m


# N, C, H, W, splits, (Cout_a, k, stride, pad), (Cout_b, stride) | None, tail
_SHAPES = [
    (2, 32, 32, 32, 16, (32, 3, 1, 1), None, "plain"),          # layer1 GN1 -> 3x3
    (2, 32, 32, 32, 16, (128, 1, 1, 0), None, "plain"),         # layer1 GN2 -> 1x1 x4
    (2, 128, 32, 32, 16, (32, 1, 1, 0), None, "residual"),      # layer1 tail -> next conv1, identity kept
    (2, 128, 32, 32, 16, (64, 1, 1, 0), (256, 2), "second"),    # layer1 last tail -> layer2 conv1 + strided downsample
    (2, 64, 32, 32, 16, (64, 3, 2, 1), None, "plain"),          # layer2 stride-2 3x3
    (4, 64, 16, 16, 16, (256, 1, 1, 0), None, "plain"),
    (4, 256, 16, 16, 16, (64, 1, 1, 0), None, "residual"),
    (2, 256, 16, 16, 3, (128, 1, 1, 0), (512, 2), "residual"),  # cpg 16: two chunks of 8
    (4, 128, 16, 16, 16, (128, 3, 2, 1), None, "plain"),        # layer3
    (4, 128, 8, 8, 16, (512, 1, 1, 0), None, "plain"),
    (2, 512, 8, 8, 16, (256, 1, 1, 0), (1024, 2), "residual"),
    (4, 256, 8, 8, 16, (256, 3, 2, 1), None, "plain"),          # layer4: 16 px
    (4, 256, 4, 4, 16, (1024, 1, 1, 0), None, "plain"),
    (4, 1024, 4, 4, 16, (256, 1, 1, 0), None, "second"),
    (4, 1024, 4, 4, 16, (128, 3, 1, 1), None, "residual"),      # last tail -> compression conv
    (3, 64, 7, 7, 5, (24, 3, 1, 1), (40, 2), "plain"),          # odd sizes: 49 px, ragged channel slices
    (1, 32, 5, 3, 1, (7, 1, 1, 0), None, "residual"),
]


@pytest.mark.parametrize("N,Cc,H,W,splits,ca,cb,tail", _SHAPES)
def test_groupnorm_fused_with_the_next_conv(N, Cc, H, W, splits, ca, cb, tail):
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(Cc * 3 + H + ca[0])
    x = torch.randn(N, Cc, H, W, generator=g) * 2 + 0.3
    gn = _gn(Cc, 16, g)
    ref = F.group_norm(x, 16, gn.weight, gn.bias, 1e-5)
    kw = {}
    if tail == "second":
        x2, gn2 = torch.randn(N, Cc, H, W, generator=g), _gn(Cc, 16, g)
        ref = ref + F.group_norm(x2, 16, gn2.weight, gn2.bias, 1e-5)
        kw = dict(x2=_slabs(x2, 4, g), gn2=gn2.to(DEV))
    elif tail == "residual":
        res = torch.randn(N, Cc, H, W, generator=g)
        ref = ref + res
        kw = dict(residual=res.to(DEV))
    ref = F.relu(ref).detach()
    wa = torch.randn(ca[0], Cc, ca[1], ca[1], generator=g) / (Cc * ca[1] ** 2) ** 0.5
    ref_a = F.conv2d(ref, wa, None, ca[2], ca[3])
    conv_b, ref_b = None, None
    if cb is not None:
        wb = torch.randn(cb[0], Cc, 1, 1, generator=g) / Cc ** 0.5
        ref_b = F.conv2d(ref, wb, None, cb[1])
        conv_b = (wb.to(DEV), cb[1])
    r = ops.gn_conv(_slabs(x, splits, g), gn.to(DEV), relu=True, want_act=True, conv_a=(wa.to(DEV), ca[2], ca[3]),
                    conv_b=conv_b, **kw)
    assert r is not None, "shape must be inside the kernel's envelope"
    act, ya, yb = r
    tol = 5e-5 * max(1.0, float(ref.abs().max()))
    assert float((act.cpu() - ref).abs().max()) < tol
    assert ya.splits == 16 and (ya.N, ya.C, ya.H, ya.W) == tuple(ref_a.shape)
    assert float((_sum(ya) - ref_a).abs().max()) < 2 * tol
    if cb is not None:
        assert (yb.N, yb.C, yb.H, yb.W) == tuple(ref_b.shape)
        assert float((_sum(yb) - ref_b).abs().max()) < 2 * tol


def test_stem_groupnorm_relu_maxpool_and_both_convs_of_the_first_bottleneck():
    """GN(16, 32) + ReLU + MaxPool2d(3, 2, 1) on the 64x64 stem output, then layer1.0's conv1 and downsample conv."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(11)
    N = 3
    x = torch.randn(N, 32, 64, 64, generator=g)
    gn = _gn(32, 16, g)
    ref = F.max_pool2d(F.relu(F.group_norm(x, 16, gn.weight, gn.bias, 1e-5)), 3, 2, 1).detach()
    wa, wb = torch.randn(32, 32, 1, 1, generator=g) / 32 ** 0.5, torch.randn(128, 32, 1, 1, generator=g) / 32 ** 0.5
    r = ops.gn_conv(_slabs(x, 2, g), gn.to(DEV), relu=True, pool=True, want_act=True, conv_a=(wa.to(DEV), 1, 0),
                    conv_b=(wb.to(DEV), 1))
    assert r is not None
    act, ya, yb = r
    assert tuple(act.shape) == (N, 32, 32, 32)
    assert float((act.cpu() - ref).abs().max()) < 3e-5
    assert float((_sum(ya) - F.conv2d(ref, wa)).abs().max()) < 1e-4
    assert float((_sum(yb) - F.conv2d(ref, wb)).abs().max()) < 1e-4


@pytest.mark.parametrize("N,C0,H,W,Cc,ca,cb,tail", [
    (4, 128, 8, 8, 512, (128, 1, 1, 0), None, "residual"),       # layer3 bottleneck tail
    (2, 128, 8, 8, 512, (256, 1, 1, 0), (1024, 2), "residual"),  # layer3 last -> layer4 conv1 + downsample
    (4, 256, 4, 4, 1024, (256, 1, 1, 0), None, "second"),        # layer4 first block (downsample operand)
    (4, 256, 4, 4, 1024, (128, 3, 1, 1), None, "residual"),      # layer4 last -> compression conv
    (2, 64, 16, 16, 256, (64, 1, 1, 0), None, "residual"),       # layer2
    (3, 32, 5, 7, 96, (10, 3, 2, 1), (6, 1), "plain"),           # odd sizes
])
def test_two_conv_layers_per_launch_front_stage(N, C0, H, W, Cc, ca, cb, tail):
    """front = (x0 slabs, GroupNorm0, 1x1 conv w0): relu(GN0(x0)) -> conv w0 -> GN -> (+ ...) -> ReLU -> next conv(s),
    i.e. a Bottleneck's GN2 -> conv3 -> GN3 tail -> the next block's conv1 in ONE launch."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(C0 + Cc + H)
    x0 = torch.randn(N, C0, H, W, generator=g) * 1.5 + 0.2
    gn0, gn = _gn(C0, 16, g), _gn(Cc, 16, g)
    w0 = torch.randn(Cc, C0, 1, 1, generator=g) / C0 ** 0.5
    mid = F.conv2d(F.relu(F.group_norm(x0, 16, gn0.weight, gn0.bias, 1e-5)), w0)
    ref = F.group_norm(mid, 16, gn.weight, gn.bias, 1e-5)
    kw = {}
    if tail == "second":
        x2, gn2 = torch.randn(N, Cc, H, W, generator=g), _gn(Cc, 16, g)
        ref = ref + F.group_norm(x2, 16, gn2.weight, gn2.bias, 1e-5)
        kw = dict(x2=_slabs(x2, 16, g), gn2=gn2.to(DEV))
    elif tail == "residual":
        res = torch.randn(N, Cc, H, W, generator=g)
        ref = ref + res
        kw = dict(residual=res.to(DEV))
    ref = F.relu(ref).detach()
    wa = torch.randn(ca[0], Cc, ca[1], ca[1], generator=g) / (Cc * ca[1] ** 2) ** 0.5
    ref_a = F.conv2d(ref, wa, None, ca[2], ca[3])
    conv_b = ref_b = None
    if cb is not None:
        wb = torch.randn(cb[0], Cc, 1, 1, generator=g) / Cc ** 0.5
        ref_b = F.conv2d(ref, wb, None, cb[1])
        conv_b = (wb.to(DEV), cb[1])
    r = ops.gn_conv(None, gn.to(DEV), relu=True, want_act=True, conv_a=(wa.to(DEV), ca[2], ca[3]), conv_b=conv_b,
                    front=(_slabs(x0, 16, g), gn0.to(DEV), w0.to(DEV)), **kw)
    assert r is not None, "shape must be inside the kernel's envelope"
    act, ya, yb = r
    tol = 1e-4 * max(1.0, float(ref.abs().max()))
    assert float((act.cpu() - ref).abs().max()) < tol
    assert float((_sum(ya) - ref_a).abs().max()) < 2 * tol
    if cb is not None:
        assert float((_sum(yb) - ref_b).abs().max()) < 2 * tol


def test_refuses_shapes_outside_its_envelope():
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(1)
    w = torch.randn(8, 48, 1, 1, device=DEV)
    assert ops.gn_conv(_slabs(torch.randn(1, 48, 4, 4, generator=g), 1, g), _gn(48, 16, g).to(DEV), conv_a=(w, 1, 0)) is None  # 3 per group
    w = torch.randn(8, 32, 1, 1, device=DEV)
    assert ops.gn_conv(_slabs(torch.randn(1, 32, 96, 96, generator=g), 1, g), _gn(32, 16, g).to(DEV), conv_a=(w, 1, 0)) is None  # tile > LDS budget


@pytest.mark.parametrize("B,first,pair", [(1, 0, 16), (4, 0, 0), (8, 0, 16), (4, 3, 7), (4, 7, 7), (2, 16, 16), (8, 3, 3), (4, 3, 13),
                                          (8, -1, 16), (4, -1, 16), (6, 9, 16)])
def test_depth_encoder_chain_matches_pairwise_path_and_oracle(B, first, pair):
    """The whole ResNetEncoder through the gn_conv chain == the deferred conv + GroupNorm pairs == the oracle's
    torch restatement (oracle/habitat_ext_ref.py), random-init weights, fp32."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from ivln_ce_amd import ops
    from ivln_ce_amd.encoders import ResNetEncoder
    from oracle import habitat_ext_ref as R

    torch.manual_seed(5)
    enc = ResNetEncoder((256, 256, 1)).to(DEV).eval()
    depth = torch.rand(B, 256, 256, 1, generator=torch.Generator().manual_seed(B))
    old, old_first, old_pair = ops.CHAIN_GN_CONV, ops.CHAIN_FROM_BLOCK, ops.CHAIN_PAIR_FROM_BLOCK
    try:
        # `first`: the bottleneck at which the chain takes over; `pair`: the first one with two conv layers per launch
        ops.CHAIN_GN_CONV, ops.CHAIN_FROM_BLOCK, ops.CHAIN_PAIR_FROM_BLOCK = True, first, pair
        with torch.no_grad():
            a = enc({"depth": depth.to(DEV)}).cpu()
            a2 = enc({"depth": depth.to(DEV)}).cpu()
        ops.CHAIN_GN_CONV = False
        with torch.no_grad():
            b = enc({"depth": depth.to(DEV)}).cpu()
    finally:
        ops.CHAIN_GN_CONV, ops.CHAIN_FROM_BLOCK, ops.CHAIN_PAIR_FROM_BLOCK = old, old_first, old_pair
    assert torch.equal(a, a2), "the chain is deterministic"
    assert float((a - b).abs().max()) < 2e-4, float((a - b).abs().max())
    import types

    space = types.SimpleNamespace(spaces={"depth": types.SimpleNamespace(shape=(256, 256, 1))})
    ref = R.ResNetEncoder(space, baseplanes=32, ngroups=16, make_backbone=R.resnet50)
    ref.load_state_dict({k: v.cpu() for k, v in enc.state_dict().items()})
    # fp64 restatement: the fp32 CPU run is itself 1-1.5e-4 off after 50 layers, in a direction that depends on how many
    # threads torch splits its reductions over - the bar is the HIP path's distance from the exact arithmetic
    ref = ref.double()
    with torch.no_grad():
        r = ref({"depth": depth.double()}).float()
    assert float((a - r).abs().max()) < 2e-4, float((a - r).abs().max())
    assert np.isfinite(a.numpy()).all()


@pytest.mark.parametrize("N,Cc,H,W,ca,cb,mode", [
    (4, 32, 32, 32, (32, 1), (128, 1), "plain_in"),     # layer1.0: conv1 + downsample of the pooled stem activation
    (4, 32, 32, 32, (32, 3), None, "gn"),               # GN1 on load -> 3x3
    (2, 32, 32, 32, (128, 1), None, "gn"),              # GN2 on load -> 1x1 x4
    (2, 128, 32, 32, (32, 1), None, "gn_second"),       # tail of block 0: GN3 + GN_ds on load -> next conv1
    (2, 128, 32, 32, (32, 1), None, "gn_residual"),     # tail of an identity block
    (3, 64, 16, 16, (32, 3), None, "gn"),               # 16x16 maps: four rows per strip
    (1, 16, 8, 12, (6, 3), (10, 1), "gn_residual"),     # odd sizes, ragged last strip (rows_per_block 5)
    (2, 64, 32, 32, (64, 3, 2), None, "gn"),            # layer2.0 conv2: 3x3 stride 2
    (2, 128, 32, 32, (64, 1), (256, 1, 2), "gn_residual"),  # layer1 tail -> layer2.0 conv1 + stride-2 downsample
    (2, 64, 16, 16, (256, 1), None, "gn"),              # 256 output channels: split over blockIdx.z
])
def test_conv_with_groupnorm_on_load_and_statistics_out(N, Cc, H, W, ca, cb, mode):
    """ivln_nconv_f32: in = relu(GN(x) [+ GN2(x2)] [+ res]) built on load from the producer's (count, mean, M2)
    partials, conv(s) over the full K, raw outputs + the partials of THEIR statistics - against F.group_norm / F.conv2d;
    the emitted partials, merged, must reproduce the two-pass mean / variance of the outputs."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(Cc + H + ca[0])
    groups = 16 if Cc % 32 == 0 else 8
    rs = 5 if H == 8 else 0
    x = torch.randn(N, Cc, H, W, generator=g) * 1.7 + 0.4

    def partials(t, ngroups, strips, rows):
        """what a producer would have left: per strip (count, mean, M2) of every (image, group)"""
        n_, c_, h_, w_ = t.shape
        out = torch.zeros(strips, n_, ngroups, 3)
        for s in range(strips):
            blk = t[:, :, s * rows:(s + 1) * rows].reshape(n_, ngroups, -1)
            out[s, :, :, 0] = blk.shape[2]
            out[s, :, :, 1] = blk.mean(2)
            out[s, :, :, 2] = ((blk - blk.mean(2, keepdim=True)) ** 2).sum(2)
        return out

    rows_in = max(1, 64 // W) if rs == 0 else rs
    rows_in = min(rows_in, H)
    strips_in = (H + rows_in - 1) // rows_in
    kw, ref = {}, x
    gn = None
    if mode != "plain_in":
        gn = _gn(Cc, groups, g)
        ref = F.group_norm(x, groups, gn.weight, gn.bias, 1e-5)
        cnhw = lambda t: t.permute(1, 0, 2, 3).contiguous()  # noqa: E731  raw tensors are [C][N][H][W]
        xin = ops.RawStats(cnhw(x).to(DEV), partials(x, groups, strips_in, rows_in).to(DEV), strips_in, groups)
        if mode == "gn_second":
            x2, gn2 = torch.randn(N, Cc, H, W, generator=g) * 0.8 - 0.3, _gn(Cc, groups, g)
            ref = ref + F.group_norm(x2, groups, gn2.weight, gn2.bias, 1e-5)
            kw = dict(x2=ops.RawStats(cnhw(x2).to(DEV), partials(x2, groups, strips_in, rows_in).to(DEV), strips_in, groups), gn2=gn2.to(DEV))
        elif mode == "gn_residual":
            res = torch.randn(N, Cc, H, W, generator=g)
            ref = ref + res
            kw = dict(residual=res.to(DEV))
        ref = F.relu(ref)
        gn = gn.to(DEV)
    else:
        xin = x.to(DEV)
    ref = ref.detach()
    ga = 16 if ca[0] % 32 == 0 else (2 if ca[0] % 2 == 0 else 1)
    sa = ca[2] if len(ca) > 2 else 1
    wa = torch.randn(ca[0], Cc, ca[1], ca[1], generator=g) / (Cc * ca[1] ** 2) ** 0.5
    ref_a = F.conv2d(ref, wa, None, sa, ca[1] // 2)
    conv_b = ref_b = None
    if cb is not None:
        gb = 16 if cb[0] % 32 == 0 else 2
        sb = cb[2] if len(cb) > 2 else 1
        wb = torch.randn(cb[0], Cc, 1, 1, generator=g) / Cc ** 0.5
        ref_b = F.conv2d(ref, wb, None, sb)
        conv_b = (wb.to(DEV), gb, sb)
    r = ops.nconv(xin, gn, relu=(mode != "plain_in"), want_act=(mode != "plain_in" and sa == 1), conv_a=(wa.to(DEV), ga, sa), conv_b=conv_b,
                  rows_per_block=rs, **kw)
    assert r is not None, "shape must be inside the kernel's envelope"
    act, a, b = r
    tol = 5e-5 * max(1.0, float(ref.abs().max()))
    if act is not None:
        assert float((act.cpu() - ref).abs().max()) < tol
    for out, rr, ng in ((a, ref_a, ga), (b, ref_b, None if cb is None else gb)):
        if out is None:
            continue
        assert float((out.y.permute(1, 0, 2, 3).cpu() - rr).abs().max()) < 2 * tol
        st = out.stats.cpu().double()  # (strips, N, groups, 3): merge and compare with the direct statistics
        cnt = st[..., 0].sum(0)
        mean = (st[..., 0] * st[..., 1]).sum(0) / cnt
        M2 = (st[..., 2] + st[..., 0] * (st[..., 1] - mean) ** 2).sum(0)
        blk = rr.double().reshape(N, ng, -1)
        assert float((mean - blk.mean(2)).abs().max()) < 1e-5
        assert float((M2 / cnt - blk.var(2, unbiased=False)).abs().max()) < 1e-4 * max(1.0, float(blk.var(2).max()))
