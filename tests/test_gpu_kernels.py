"""Per-kernel GPU parity against plain torch fp32 on CPU (floating-point kernels: tolerance stated
per test; fp32 MFMA is an exact fmaf chain so only the summation order differs)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _close(got, ref, atol, rtol=1e-4):
    got, ref = got.detach().cpu(), ref.detach().cpu()
    err = (got - ref).abs().max().item()
    assert torch.allclose(got, ref, atol=atol, rtol=rtol), f"max err {err:.3e}, ref max {ref.abs().max().item():.3e}"


@pytest.mark.parametrize(
    "N,Cin,H,W,Cout,k,s,p",
    [
        (2, 1, 128, 128, 32, 7, 2, 3),    # DD-PPO stem (direct 7x7 at stride 2; the one input channel half-fills a chunk)
        (4, 3, 256, 256, 64, 7, 2, 3),    # RedNet's RGB stem: three channels = a full chunk + a ragged one
        (3, 1, 250, 250, 64, 7, 2, 3),    # RedNet's depth stem, ragged 125x125 output
        (4, 32, 32, 32, 32, 3, 1, 1),     # layer1 3x3 (M<=32 -> 32x128 tile)
        (3, 128, 32, 32, 64, 1, 1, 0),    # 1x1
        (2, 128, 32, 32, 256, 1, 2, 0),   # strided 1x1 downsample (float4-staged GEMM, two columns per 16-byte load)
        (8, 256, 64, 64, 512, 1, 2, 0),   # ... RedNet's size
        (3, 64, 30, 30, 96, 1, 2, 0),     # ... odd output width: the scalar-gather GEMM
        (2, 64, 32, 32, 64, 3, 2, 1),     # strided 3x3
        (8, 128, 32, 32, 128, 3, 2, 1),   # strided 3x3 on the direct kernel (patch staged as even | odd column planes)
        (4, 64, 64, 64, 64, 3, 2, 1),     # ... 32-wide output rows
        (8, 64, 30, 30, 64, 3, 2, 1),     # ... ragged 15x15 output
        (16, 256, 16, 16, 128, 3, 2, 1),  # ... 8x8 output, two images per pixel tile, channel chunks split
        (1, 256, 4, 4, 1024, 1, 1, 0),    # pixel-starved (N<=32 -> 128x32 tile)
        (4, 1024, 4, 4, 128, 3, 1, 1),    # compression conv: split-K
        (2, 14, 64, 64, 32, 7, 1, 3),     # map CNN layer 1
        (3, 128, 8, 8, 128, 7, 1, 3),     # map CNN layer 4
        (1, 3, 37, 53, 5, 3, 1, 1),       # ragged sizes
    ],
)
def test_conv2d(N, Cin, H, W, Cout, k, s, p):
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(N * 1000 + Cin + k)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x, w, b, stride=s, padding=p)
    got = ops.conv2d(x.to(DEV), w.to(DEV), stride=s, pad=p, shift=b.to(DEV))
    _close(got, ref, 2e-5)
    # fused epilogue: scale/shift + residual + relu
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    res = torch.randn_like(ref)
    ref2 = F.relu(F.conv2d(x, w, None, stride=s, padding=p) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + res)
    got2 = ops.conv2d(x.to(DEV), w.to(DEV), stride=s, pad=p, scale=sc.to(DEV), shift=sh.to(DEV),
                      residual=res.to(DEV), relu=True)
    _close(got2, ref2, 3e-5)


@pytest.mark.parametrize("N,Cin,H,W,Cout,k,s,p,op", [(2, 64, 8, 8, 32, 3, 2, 1, 1), (1, 32, 16, 16, 13, 2, 2, 0, 0),
                                                      (2, 16, 5, 7, 8, 3, 2, 1, 1)])
def test_conv_transpose2d(N, Cin, H, W, Cout, k, s, p, op):
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cin, Cout, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.conv_transpose2d(x, w, b, stride=s, padding=p, output_padding=op)
    got = ops.conv_transpose2d(x.to(DEV), w.permute(1, 0, 2, 3).contiguous().to(DEV), s, p, op, shift=b.to(DEV))
    _close(got, ref, 2e-5)


@pytest.mark.parametrize("rows,K,O", [(1, 512, 4), (4, 3072, 128), (8, 1184, 512), (5, 50, 512), (64, 416, 1536),
                                      (600, 50, 512), (30, 2048, 256)])
def test_linear(rows, K, O):
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(rows + K)
    x = torch.randn(rows, K, generator=g)
    w = torch.randn(O, K, generator=g) / K ** 0.5
    b = torch.randn(O, generator=g)
    _close(ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), relu=True), F.relu(F.linear(x, w, b)), 2e-5)
    # strided destination (concat-free writes)
    buf = torch.zeros(rows, O + 7, device=DEV)
    ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), out=buf[:, 3:3 + O]) if (rows > 16) else None
    if rows > 16:
        _close(buf[:, 3:3 + O], F.linear(x, w, b), 2e-5)
        assert float(buf[:, :3].abs().max()) == 0.0


@pytest.mark.parametrize("N,C,H,W,G,res", [(2, 32, 64, 64, 16, False), (3, 128, 32, 32, 16, True), (4, 128, 4, 4, 1, False),
                                           (2, 1024, 4, 4, 16, True)])
def test_groupnorm(N, C, H, W, G, res):
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(C)
    x = torch.randn(N, C, H, W, generator=g) * 2 + 0.3
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    r = torch.randn(N, C, H, W, generator=g) if res else None
    ref = F.group_norm(x, G, gamma, beta, 1e-5)
    if res:
        ref = ref + r
    ref = F.relu(ref)
    got = ops.groupnorm(x.to(DEV), gamma.to(DEV), beta.to(DEV), G, 1e-5, relu=True,
                        residual=r.to(DEV) if res else None)
    _close(got, ref, 2e-5)


def test_pools_and_map_features():
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 5, 33, 40, generator=g)
    _close(ops.pool2d(x.to(DEV), 3, 2, 1, "max"), F.max_pool2d(x, 3, 2, 1), 0)
    # MaxPool2d(3, 2, 1) on widths that are multiples of 8 (RedNet's stem): four outputs per thread from 16-byte loads - exact,
    # odd heights, the pad value never leaks (a row / column of -inf), widths that take the generic kernel beside it
    for shape in [(2, 5, 16, 16), (3, 4, 23, 8), (2, 64, 128, 128), (1, 3, 10, 40), (1, 2, 9, 12)]:
        xw = torch.randn(*shape, generator=g)
        xw[0, 0, 0, :] = float("-inf")
        xw[0, 0, :, 0] = float("-inf")
        assert torch.equal(ops.pool2d(xw.to(DEV), 3, 2, 1, "max").cpu(), F.max_pool2d(xw, 3, 2, 1)), shape
    x2 = torch.randn(2, 1, 64, 64, generator=g)
    _close(ops.pool2d(x2.to(DEV), 2, 2, 0, "avg"), F.avg_pool2d(x2, 2), 1e-6)
    occ = (torch.rand(3, 64, 64, generator=g) < 0.5).to(torch.uint8)
    sem = torch.randint(0, 13, (3, 64, 64), generator=g).to(torch.uint8)
    ref = torch.cat((occ.unsqueeze(1), F.one_hot(sem.long(), 13).permute(0, 3, 1, 2)), 1).float()
    _close(ops.map_features(occ.to(DEV), sem.to(DEV)), ref, 0)


@pytest.mark.parametrize("train", [False, True])
def test_cbra_block(train):
    import torch.nn as nn

    from ivln_ce_amd.encoders import CBRA

    torch.manual_seed(3)
    blk = CBRA(14, 32)
    blk.conv[1].running_mean.normal_(0, 0.1)
    blk.conv[1].running_var.uniform_(0.5, 1.5)
    blk.train(train)
    import copy

    ref_blk = copy.deepcopy(blk)
    x = torch.randn(4, 14, 64, 64)
    ref = ref_blk.conv(x)
    blk = blk.to(DEV)
    with torch.no_grad():
        got = blk.forward_hip(x.to(DEV))
    _close(got, ref, 3e-5)
    if train:
        _close(blk.conv[1].running_mean, ref_blk.conv[1].running_mean, 1e-6)
        _close(blk.conv[1].running_var, ref_blk.conv[1].running_var, 1e-6)


def test_lstm_bidir_matches_packed_torch_lstm():
    import torch.nn as nn

    from ivln_ce_amd import ops

    torch.manual_seed(5)
    B, L, E, H = 4, 37, 50, 128
    lens = [37, 5, 20, 1]
    rnn = nn.LSTM(E, H, bidirectional=True)
    emb_table = torch.randn(100, E)
    emb_table[0] = 0
    tokens = torch.zeros(B, L, dtype=torch.long)
    for b, n in enumerate(lens):
        tokens[b, :n] = torch.randint(1, 100, (n,))
    x = emb_table[tokens]
    packed = nn.utils.rnn.pack_padded_sequence(x, torch.tensor(lens), batch_first=True, enforce_sorted=False)
    ref = nn.utils.rnn.pad_packed_sequence(rnn(packed)[0], batch_first=True)[0].permute(0, 2, 1)
    emb, lengths = ops.embed_lengths(tokens.to(DEV), emb_table.to(DEV))
    assert lengths.cpu().tolist() == lens
    p = {k: v.detach().to(DEV) for k, v in rnn.named_parameters()}
    gx_f = ops.linear_gemm(emb, p["weight_ih_l0"], p["bias_ih_l0"])
    gx_r = ops.linear_gemm(emb, p["weight_ih_l0_reverse"], p["bias_ih_l0_reverse"])
    out, _, _ = ops.lstm_bidir(gx_f, gx_r, p["weight_hh_l0"], p["weight_hh_l0_reverse"], p["bias_hh_l0"],
                               p["bias_hh_l0_reverse"], lengths, B, L, H)
    _close(out, ref, 2e-5)
    # the over-subscribed launch (blocks draw their item when they start): the same bits for every item, whichever
    # block ran it, and the ticket word re-armed after each launch
    ticket = torch.zeros((1,), dtype=torch.int32, device=DEV)
    for spare in (2, 3, 8):
        for _ in range(3):
            o2, _, _ = ops.lstm_bidir(gx_f, gx_r, p["weight_hh_l0"], p["weight_hh_l0_reverse"], p["bias_hh_l0"],
                                      p["bias_hh_l0_reverse"], lengths, B, L, H, spare=spare, ticket=ticket)
            assert torch.equal(o2, out)
            assert int(ticket.item()) == 0
    with pytest.raises(ValueError):
        ops.lstm_bidir(gx_f, gx_r, p["weight_hh_l0"], p["weight_hh_l0_reverse"], p["bias_hh_l0"], p["bias_hh_l0_reverse"],
                       lengths, B, L, H, spare=2)


@pytest.mark.parametrize("rows,I", [(3, 416), (8, 512), (11, 416)])
def test_gru_step(rows, I):
    import torch.nn as nn

    from ivln_ce_amd import ops

    torch.manual_seed(rows)
    H = 512
    rnn = nn.GRU(I, H)
    x = torch.randn(rows, I)
    h = torch.randn(rows, 2, H)
    mask = (torch.rand(rows) < 0.6).to(torch.uint8)
    ref, _ = rnn(x.unsqueeze(0), (h[:, 1] * mask.view(-1, 1).float()).unsqueeze(0))
    p = {k: v.detach().to(DEV) for k, v in rnn.named_parameters()}
    hd = h.to(DEV)
    out = torch.zeros(rows, H + 8, device=DEV)
    out2 = torch.zeros(rows, 2, H, device=DEV)
    ops.gru_step(x.to(DEV), None, hd[:, 1], mask.to(DEV), p["weight_ih_l0"], p["weight_hh_l0"], p["bias_ih_l0"],
                 p["bias_hh_l0"], out[:, 4:4 + H], out2[:, 1])
    _close(out[:, 4:4 + H], ref[0], 2e-5)
    _close(out2[:, 1], ref[0], 2e-5)
    # precomputed-gi path (sequence mode)
    gi = F.linear(x, rnn.weight_ih_l0, rnn.bias_ih_l0)
    out3 = torch.zeros(rows, H, device=DEV)
    ops.gru_step(None, gi.detach().to(DEV), hd[:, 1], mask.to(DEV), p["weight_ih_l0"], p["weight_hh_l0"],
                 p["bias_ih_l0"], p["bias_hh_l0"], out3)
    _close(out3, ref[0], 2e-5)


def test_attention():
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(9)
    rows, Ck, Cv, I = 5, 256, 128, 200
    q = torch.randn(rows, Ck, generator=g)
    k = torch.randn(rows, Ck, I, generator=g)
    v = torch.randn(rows, Cv, I, generator=g)
    lens = torch.tensor([200, 3, 77, 1, 150], dtype=torch.int32)
    mask = torch.arange(I).view(1, -1) >= lens.view(-1, 1)
    logits = torch.einsum("nc,nci->ni", q, k) - mask.float() * 1e8
    ref = torch.einsum("ni,nci->nc", F.softmax(logits * 0.0625, 1), v)
    out = torch.zeros(rows, Cv + 3, device=DEV)
    ops.attn(q.to(DEV), k.to(DEV), v.to(DEV), lens.to(DEV), 0.0625, out[:, :Cv])
    _close(out[:, :Cv], ref, 2e-5)
    ref2 = torch.einsum("ni,nci->nc", F.softmax(torch.einsum("nc,nci->ni", q, k[:, :, :16]) * 0.0625, 1), v[:, :, :16])
    out2 = torch.zeros(rows, Cv, device=DEV)
    ops.attn(q.to(DEV), k[:, :, :16].contiguous().to(DEV), v[:, :, :16].contiguous().to(DEV), None, 0.0625, out2)
    _close(out2, ref2, 2e-5)


@pytest.mark.parametrize("N,Cin,H,W,Cout,k,s,p,G", [(4, 256, 8, 8, 128, 3, 1, 1, 16), (4, 1024, 4, 4, 128, 3, 1, 1, 1),
                                                     (2, 32, 32, 32, 128, 1, 1, 0, 16), (3, 1, 128, 128, 32, 7, 2, 3, 16)])
def test_deferred_conv_fused_into_groupnorm(N, Cin, H, W, Cout, k, s, p, G):
    """conv leaves split-K slabs in the workspace; GroupNorm reduces + normalises (+res, +ReLU)."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(Cin + k)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    gamma, beta = torch.randn(Cout, generator=g), torch.randn(Cout, generator=g)
    y = F.conv2d(x, w, None, stride=s, padding=p)
    res = torch.randn_like(y)
    ref = F.relu(F.group_norm(y, G, gamma, beta, 1e-5) + res)
    d = ops.conv2d(x.to(DEV), w.to(DEV), stride=s, pad=p, defer=True)
    assert isinstance(d, ops.Deferred) and d.splits >= 1
    got = ops.groupnorm(d, gamma.to(DEV), beta.to(DEV), G, 1e-5, relu=True, residual=res.to(DEV))
    _close(got, ref, 3e-5)


def test_deferred_conv_fused_into_bn_relu_pool():
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(11)
    x = torch.randn(4, 128, 8, 8, generator=g)
    w = torch.randn(128, 128, 7, 7, generator=g) / (128 * 49) ** 0.5
    sc, sh = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g)
    ref = F.avg_pool2d(F.relu(F.conv2d(x, w, None, padding=3) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)), 2)
    d = ops.conv2d(x.to(DEV), w.to(DEV), pad=3, defer=True)
    assert d.splits > 1
    _close(ops.scale_shift_relu_avgpool2(d, sc.to(DEV), sh.to(DEV)), ref, 3e-5)


@pytest.mark.parametrize("override", [1, 4, 5])
@pytest.mark.parametrize("N,Cin,H,W,Cout,k,s,p", [(2, 64, 32, 32, 192, 3, 1, 1), (1, 96, 40, 24, 130, 1, 1, 0),
                                                   (2, 24, 33, 31, 70, 7, 2, 3)])
def test_conv2d_every_block_tile(override, N, Cin, H, W, Cout, k, s, p):
    """The register-tiled 128x128 / 64x128 variants must agree with the 64x64 tile (ragged edges too)."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(override * 7 + k)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.relu(F.conv2d(x, w, b, stride=s, padding=p))
    ops.TILE_OVERRIDE = override
    try:
        got = ops.conv2d(x.to(DEV), w.to(DEV), stride=s, pad=p, shift=b.to(DEV), relu=True)
    finally:
        ops.TILE_OVERRIDE = 0
    _close(got, ref, 3e-5)


@pytest.mark.parametrize(
    "N,Cin,H,W,Cout,k,p",
    [
        (3, 14, 64, 64, 32, 7, 3),     # 32x4 pixel tile, 32-channel blocks (map CNN layer 1)
        (2, 32, 32, 32, 64, 7, 3),     # 32x4 tile, 64-channel blocks
        (5, 64, 16, 16, 128, 7, 3),    # 16x8 tile
        (5, 128, 8, 8, 128, 7, 3),     # 8x8 tile, two images per block (odd image count -> masked image)
        (9, 16, 4, 4, 48, 7, 3),       # 4x4 tile, eight images per block
        (2, 64, 40, 24, 70, 3, 1),     # ragged: tile overhang in both directions, 70 channels
        (3, 256, 8, 8, 256, 3, 1),
        (11, 512, 4, 4, 40, 3, 1),
        (2, 8, 21, 19, 24, 3, 0),      # no padding: output smaller than input
        (1, 16, 12, 12, 8, 7, 6),      # pad = KS-1 (the input-gradient form of a pad-0 conv)
    ],
)
def test_conv2d_direct_lds_patch_kernel(N, Cin, H, W, Cout, k, p):
    """conv_direct.hip forced (tile_override 6) against torch CPU fp32 and against the implicit GEMM;
    tolerance 3e-5 abs on O(1) outputs (only the fp32 summation order differs)."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(N * 100 + Cin + k)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    ref0 = F.conv2d(x, w, None, stride=1, padding=p)
    res = torch.randn_like(ref0)
    ref = F.relu(ref0 * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + res)
    xd, wd = x.to(DEV), w.to(DEV)
    try:
        ops.TILE_OVERRIDE = 6
        got = ops.conv2d(xd, wd, pad=p, scale=sc.to(DEV), shift=sh.to(DEV), residual=res.to(DEV), relu=True)
        got_nosplit = ops.conv2d(xd, wd, pad=p, splitk=False)
        # channel slice of a wider destination (concat-free skip connections)
        wide = torch.zeros(N, Cout + 5, *ref0.shape[2:], device=DEV)
        ops.conv2d(xd, wd, pad=p, out=wide[:, 3:], out_ctot=Cout + 5)
        # deferred raw slabs, reduced by the consumer
        dfr = ops.conv2d(xd, wd, pad=p, defer=True)
        slabs = dfr.ws[: dfr.splits * Cout * N * ref0.shape[2] * ref0.shape[3]].view(dfr.splits, Cout, N, *ref0.shape[2:]).sum(0)
        ops.TILE_OVERRIDE = 1
        gemm = ops.conv2d(xd, wd, pad=p, splitk=False)
    finally:
        ops.TILE_OVERRIDE = 0
    _close(got, ref, 3e-5)
    _close(got_nosplit, ref0, 3e-5)
    _close(got_nosplit, gemm, 3e-5)
    _close(wide[:, 3:3 + Cout], ref0, 3e-5)
    assert float(wide[:, :3].abs().max()) == 0.0 and float(wide[:, 3 + Cout:].abs().max()) == 0.0
    _close(slabs.permute(1, 0, 2, 3), ref0, 3e-5)


@pytest.mark.parametrize(
    "N,Cin,H,W,Cout,k",
    [
        (6, 14, 64, 64, 32, 7),     # map CNN layer 1: ragged 14-channel chunk, 32 x 512 tiles of 16 x 32 pixels
        (4, 32, 32, 32, 64, 7),     # layer 2: 64 x 512 tiles
        (5, 64, 16, 16, 128, 7),    # layer 3: 16 x 16 tiles (odd image count: 64 x 256 and 128 x 256 both see a masked image)
        (9, 128, 8, 8, 128, 7),     # layer 4: 8 x 8 tiles, four images per tile, ragged image group
        (3, 64, 32, 32, 32, 7),     # layer 2's input gradient
        (2, 64, 128, 128, 64, 3),   # RedNet layer 1
        (3, 48, 40, 24, 70, 3),     # ragged: tile overhang both ways, 70 channels (masked channel tile), 48 = 3 chunks
        (16, 256, 8, 8, 40, 3),     # eight images per tile
        (2, 16, 16, 48, 256, 3),    # two channel-tile rows of 128
    ],
)
def test_conv2d_split_bf16_kernel(N, Cin, H, W, Cout, k):
    """conv_bf3.hip forced (tile_override 9): both operands as three exact bf16 pieces, six of the nine piece products on the
    bf16 MFMA pipe, fp32 accumulation.  Against a float64 convolution its error has to be what the fp32 MFMA direct kernel's
    is (measured 0.7-2.4e-6 of the largest output for both; bar: 3e-6 of the largest output, and at most twice the fp32
    kernel's + 1e-6), and all the epilogue forms of ivln_gemm_f32 have to hold: scale / shift / residual / ReLU, a channel
    slice of a wider destination, image-grouped weights."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(N * 100 + Cin + k)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    ref0 = F.conv2d(x.double(), w.double(), None, stride=1, padding=k // 2)
    res = torch.randn(ref0.shape, generator=g)
    ref = F.relu(ref0 * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1) + res.double())
    xd, wd = x.to(DEV), w.to(DEV)
    try:
        ops.TILE_OVERRIDE = 9
        got = ops.conv2d(xd, wd, pad=k // 2, scale=sc.to(DEV), shift=sh.to(DEV), residual=res.to(DEV), relu=True)
        plain = ops.conv2d(xd, wd, pad=k // 2, splitk=False)
        again = ops.conv2d(xd, wd, pad=k // 2, splitk=False)
        wide = torch.zeros(N, Cout + 8, *ref0.shape[2:], device=DEV)
        ops.conv2d(xd, wd, pad=k // 2, out=wide[:, 4:], out_ctot=Cout + 8, splitk=False)
        # with a workspace on hand small grids split their channel chunks over blockIdx.z (raw slabs + k_splitk_epilogue)
        wide_k = torch.zeros(N, Cout + 8, *ref0.shape[2:], device=DEV)
        ops.conv2d(xd, wd, pad=k // 2, out=wide_k[:, 4:], out_ctot=Cout + 8, scale=sc.to(DEV), shift=sh.to(DEV), relu=True)
        got_k = ops.conv2d(xd, wd, pad=k // 2, scale=sc.to(DEV), shift=sh.to(DEV), residual=res.to(DEV), relu=True, splitk=True)
        ops.TILE_OVERRIDE = 6
        fp32 = ops.conv2d(xd, wd, pad=k // 2, splitk=False)
    finally:
        ops.TILE_OVERRIDE = 0
    scale = float(ref0.abs().max())
    e_split = float((plain.double().cpu() - ref0).abs().max()) / scale
    e_fp32 = float((fp32.double().cpu() - ref0).abs().max()) / scale
    assert e_split <= 3e-6 and e_split <= 2.0 * e_fp32 + 1e-6, (e_split, e_fp32)
    assert torch.equal(plain, again)  # (no atomics, fixed order: run-to-run identical)
    _close(got, ref.float(), 3e-5)
    assert torch.equal(wide[:, 4:4 + Cout], plain)
    assert float(wide[:, :4].abs().max()) == 0.0 and float(wide[:, 4 + Cout:].abs().max()) == 0.0
    _close(wide_k[:, 4:4 + Cout], F.relu(ref0 * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)).float(), 3e-5)
    _close(got_k, ref.float(), 3e-5)
    assert float(wide_k[:, :4].abs().max()) == 0.0 and float(wide_k[:, 4 + Cout:].abs().max()) == 0.0


@pytest.mark.parametrize("N,Cin,H,W,Cout", [(8, 3, 256, 256, 64), (8, 1, 256, 256, 64), (2, 3, 12, 512, 48), (3, 1, 6, 256, 72), (1, 3, 2, 256, 32)])
def test_conv2d_7x7_stride2_stems_on_the_split_bf16_kernel(N, Cin, H, W, Cout):
    """RedNet's stems (rednet.py:201-210, 190-199: 7x7, stride 2, pad 3, 3 | 1 -> 64 channels on 256 x 256 inputs) on
    k_conv7s2_bf3 (K as kernel rows x 8 columns, fragments built in registers): against the float64 conv, 3e-6 of the largest
    output and at or below twice the fp32 direct kernel's error; the folded BatchNorm + ReLU epilogue with the residual in front
    of and BEHIND the ReLU (the stems' fusion add); ragged channel counts, two row segments, maps of one output row; reproducible;
    a channel slice of a wider tensor as the destination and a strided input batch; other widths go to the fp32 kernel."""
    import ctypes as C

    from ivln_ce_amd import ops
    from ivln_ce_amd._lib import lib

    g = torch.Generator().manual_seed(N + Cin + Cout + H)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 7, 7, generator=g) / (Cin * 49) ** 0.5
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    res = torch.randn(N, Cout, H // 2, W // 2, generator=g)
    ref0 = F.conv2d(x.double(), w.double(), stride=2, padding=3)
    aff = ref0 * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    xd, wd, scd, shd, resd = x.to(DEV), w.to(DEV), sc.to(DEV), sh.to(DEV), res.to(DEV)
    L = lib()
    L.ivln_conv_split_counters.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.c_int]

    def run(stem, xin=None, **kw):
        L.ivln_conv_split_counters(None, None, 1)
        old = ops.BF3_STEM
        ops.BF3_STEM = stem
        try:
            y = ops.conv2d(xd if xin is None else xin, wd, stride=2, pad=3, **kw)
        finally:
            ops.BF3_STEM = old
        n = C.c_longlong(0)
        L.ivln_conv_split_counters(None, C.byref(n), 0)
        return y, n.value

    scale0 = float(ref0.abs().max())
    fp32, n0 = run(False)
    assert n0 == 0
    e_fp32 = float((fp32.double().cpu() - ref0).abs().max()) / scale0
    plain, n1 = run(True)
    assert n1 == 1, "the stem kernel declined a shape it is built for"
    e = float((plain.double().cpu() - ref0).abs().max()) / scale0
    assert e <= 3e-6 and e <= 2.0 * e_fp32 + 1e-6, (e, e_fp32)
    assert torch.equal(plain, run(True)[0])
    for name, kw, want in (("bn+relu", dict(scale=scd, shift=shd, relu=True), F.relu(aff)),
                           ("residual, relu", dict(scale=scd, shift=shd, relu=True, residual=resd), F.relu(aff + res.double())),
                           ("relu, residual behind it", dict(scale=scd, shift=shd, relu=True, residual=resd, residual_after_relu=True),
                            F.relu(aff) + res.double()),
                           ("shift only", dict(shift=shd), ref0 + sh.double().view(1, -1, 1, 1))):
        y, n = run(True, **kw)
        assert y is not None and n == 1, name
        assert float((y.double().cpu() - want).abs().max()) <= 3e-6 * float(want.abs().max()), name
    # destination = a channel slice of a wider tensor, input = every second image of a longer batch (in_img_stride)
    wide = torch.full((N, Cout + 8, H // 2, W // 2), 7.0, device=DEV)
    y, n = run(True, scale=scd, shift=shd, relu=True, out=wide[:, 4:4 + Cout], out_ctot=Cout + 8)
    assert n == 1 and float((wide[:, 4:4 + Cout].double().cpu() - F.relu(aff)).abs().max()) <= 3e-6 * float(aff.abs().max())
    assert bool((wide[:, :4] == 7.0).all()) and bool((wide[:, 4 + Cout:] == 7.0).all())
    x2 = torch.randn(2 * N, Cin, H, W, generator=g).to(DEV)
    x2[::2] = xd
    y, n = run(True, xin=x2[::2], in_img_stride=2 * Cin * H * W)
    assert n == 1 and torch.equal(y, plain)
    # a width the kernel is not built for: the fp32 direct kernel, same numbers as ever
    xs = xd[..., : W - 32].contiguous()
    ys, ns = run(True, xin=xs)
    assert ns == 0
    refs = F.conv2d(xs.double().cpu(), w.double(), stride=2, padding=3)
    assert float((ys.double().cpu() - refs).abs().max()) <= 3e-6 * float(refs.abs().max())


@pytest.mark.parametrize("N,Cin,H,W,Cout,G", [(16, 128, 64, 64, 128, 2), (16, 256, 32, 32, 256, 2), (16, 512, 16, 16, 512, 2), (3, 48, 24, 40, 40, 0),
                                              (2, 32, 16, 16, 24, 0)])
def test_conv2d_stride2_3x3_on_the_split_bf16_kernel(N, Cin, H, W, Cout, G):
    """RedNet's stride-2 3x3 convs (the entry blocks of layers 2-4, rednet.py:84-110, image-grouped over the stacked encoders)
    on the tiled split-bf16 kernel with its patch staged as four phase planes (k_conv_bf3<..., ST = 2>): against the float64
    conv with the folded BatchNorm + ReLU epilogue, 3e-6 of the largest output and at or below twice the fp32 direct kernel's
    error; every tile that fits (tile_override 22-26), with and without channel chunks split over blockIdx.z; reproducible;
    output written into a channel slice of a wider tensor; and the default dispatch takes this kernel at RedNet's sizes."""
    import ctypes as C

    from ivln_ce_amd import ops
    from ivln_ce_amd._lib import lib

    g = torch.Generator().manual_seed(N + Cin + Cout)
    x = torch.randn(N, Cin, H, W, generator=g)
    ws = [torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5 for _ in range(max(G, 1))]
    sc, sh = torch.rand(max(G, 1) * Cout, generator=g) + 0.5, torch.randn(max(G, 1) * Cout, generator=g)
    per = N // max(G, 1)
    ref0 = torch.cat([F.conv2d(x[i * per:(i + 1) * per].double(), ws[i].double(), stride=2, padding=1) for i in range(max(G, 1))])
    ref = torch.cat([F.relu(ref0[i * per:(i + 1) * per] * sc[i * Cout:(i + 1) * Cout].double().view(1, -1, 1, 1)
                            + sh[i * Cout:(i + 1) * Cout].double().view(1, -1, 1, 1)) for i in range(max(G, 1))])
    w = (torch.stack(ws) if G else ws[0]).contiguous().to(DEV)
    xd, scd, shd = x.to(DEV), sc.to(DEV), sh.to(DEV)
    L = lib()
    L.ivln_conv_split_counters.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.c_int]

    def run(override, **kw):
        L.ivln_conv_split_counters(None, None, 1)
        ops.TILE_OVERRIDE = override
        try:
            y = ops.conv2d(xd, w, stride=2, pad=1, **kw)
        finally:
            ops.TILE_OVERRIDE = 0
        n = C.c_longlong(0)
        L.ivln_conv_split_counters(None, C.byref(n), 0)
        return y, n.value

    scale0, scale = float(ref0.abs().max()), float(ref.abs().max())
    fp32, n0 = run(6, splitk=False)
    assert n0 == 0
    e_fp32 = float((fp32.double().cpu() - ref0).abs().max()) / scale0
    plain, n1 = run(9, splitk=False)
    assert n1 == 1, "the split-bf16 kernel declined a shape it is built for"
    e = float((plain.double().cpu() - ref0).abs().max()) / scale0
    assert e <= 3e-6 and e <= 2.0 * e_fp32 + 1e-6, (e, e_fp32)
    assert torch.equal(plain, run(9, splitk=False)[0])
    full, _ = run(9, scale=scd, shift=shd, relu=True)  # (with a workspace on hand small grids split their channel chunks)
    assert float((full.double().cpu() - ref).abs().max()) <= 3e-6 * scale
    for ov in (22, 23, 24, 25, 26):
        if (ov in (23, 25) and Cout < 128) or (G and N // G * (H // 2) * (W // 2) % (256 if ov in (22, 23, 26) else 128)):
            continue
        try:
            y, n = run(ov, scale=scd, shift=shd, relu=True, splitk=False)
        except Exception:  # noqa: BLE001 - a tile whose planes do not fit is declined (IVLN_E_UNSUPPORTED under an insisting override)
            continue
        assert n == 1 and float((y.double().cpu() - ref).abs().max()) <= 3e-6 * scale, ov
    wide = torch.zeros(N, Cout + 8, H // 2, W // 2, device=DEV)
    ops.TILE_OVERRIDE = 9
    try:
        ops.conv2d(xd, w, stride=2, pad=1, out=wide[:, 4:], out_ctot=Cout + 8, splitk=False)
    finally:
        ops.TILE_OVERRIDE = 0
    assert torch.equal(wide[:, 4:4 + Cout], plain) and float(wide[:, :4].abs().max()) == 0.0 and float(wide[:, 4 + Cout:].abs().max()) == 0.0
    if N == 16:  # RedNet's own shapes: what the default dispatch launches
        dflt, n3 = run(0, scale=scd, shift=shd, relu=True)
        assert n3 == 1 and float((dflt.double().cpu() - ref).abs().max()) <= 3e-6 * scale


@pytest.mark.parametrize("N,Cin,Cout,H,W,form", [(2, 64, 64, 32, 32, 13), (2, 256, 64, 16, 16, 13), (2, 512, 128, 8, 16, 12), (1, 1024, 256, 8, 8, 12),
                                                 (3, 80, 40, 4, 8, 13)])
def test_conv1x1_split_bf16_residual_behind_the_relu(N, Cin, Cout, H, W, form):
    """ivln_gemm_desc.residual_after_relu: relu(scale * conv + shift) + residual in the 1x1 kernel's epilogue - RedNet's decoder
    skips (rednet.py:244-263: x = deconv(x) + agant(fuse)) without the add launch.  Both forms of k_conv1x1_bf3_ks: bit-identical
    to the same form followed by the add kernel; within the split-bf16 bar of the float64 composition; a conv the 1x1 kernels do
    not take (3x3) comes back as None."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(Cin + Cout + H)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    up = torch.randn(N, Cout, H, W, generator=g)
    ref = F.relu(F.conv2d(x.double(), w.double()) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)) + up.double()
    d = [v.to(DEV) for v in (x, w, sc, sh, up)]
    got = ops.conv2d(d[0], d[1], scale=d[2], shift=d[3], residual=d[4], relu=True, residual_after_relu=True)
    assert got is not None, "the library declined an eligible shape"
    try:
        ops.TILE_OVERRIDE = form  # the same form, then the add kernel
        two = ops.add(d[4], ops.conv2d(d[0], d[1], scale=d[2], shift=d[3], relu=True))
    finally:
        ops.TILE_OVERRIDE = 0
    _close(got, ref.float(), 3e-5)
    assert torch.equal(got, two)
    w3 = torch.randn(Cout, Cin, 3, 3, generator=g).to(DEV)
    assert ops.conv2d(d[0], w3, pad=1, scale=d[2], shift=d[3], residual=d[4], relu=True, residual_after_relu=True) is None


@pytest.mark.parametrize("N,Cin,Cmid,Cout,H,W,G", [(2, 64, 64, 256, 32, 32, 0), (2, 128, 128, 512, 8, 32, 0), (4, 64, 64, 256, 16, 64, 2),
                                                   (3, 48, 64, 96, 4, 32, 0), (2, 128, 128, 512, 32, 32, 2), (32, 128, 128, 256, 32, 32, 2)])
def test_bottleneck_tail_as_one_launch(N, Cin, Cmid, Cout, H, W, G):
    """k_conv_bf3<..., FUSE> (ivln_gemm_desc.fuse_*): a ResNet bottleneck's conv2 3x3 + bn2 + ReLU + conv3 1x1 + bn3 + residual +
    ReLU (rednet.py:20-65) in one launch, the mid tensor in LDS only.  Against the float64 composition (the bar of the split-bf16
    kernels) and against the two separate launches (same arithmetic, same order per accumulator: equal to rounding of the mid
    tensor's different summation order at most); image-grouped weight pairs (the stacked RGB + depth encoders); a ragged
    channel count in front (48) and behind (96); run-to-run identical bits.  128 mid channels take the 128 x 64-pixel tile
    (four waves of 32 channels, two pixels per lane in the second stage) while the 128-pixel grid would leave CUs idle, the
    128 x 128 tile from 256 workgroups on (the 32-image case)."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(N * 1000 + Cmid + W)
    x = torch.randn(N, Cin, H, W, generator=g)
    sh2, sh3 = ((G, Cmid, Cin, 3, 3), (G, Cout, Cmid, 1, 1)) if G else ((Cmid, Cin, 3, 3), (Cout, Cmid, 1, 1))
    w2 = torch.randn(*sh2, generator=g) / (Cin * 9) ** 0.5
    w3 = torch.randn(*sh3, generator=g) / Cmid ** 0.5
    GG = max(G, 1)
    s2, b2 = torch.rand(GG * Cmid, generator=g) + 0.5, torch.randn(GG * Cmid, generator=g)
    s3, b3 = torch.rand(GG * Cout, generator=g) + 0.5, torch.randn(GG * Cout, generator=g)
    res = torch.randn(N, Cout, H, W, generator=g)
    B = N // GG
    outs = []
    for i in range(GG):
        xi = x[i * B:(i + 1) * B].double()
        wa, wb = (w2[i], w3[i]) if G else (w2, w3)
        y = F.relu(F.conv2d(xi, wa.double(), None, padding=1) * s2[i * Cmid:(i + 1) * Cmid].double().view(1, -1, 1, 1)
                   + b2[i * Cmid:(i + 1) * Cmid].double().view(1, -1, 1, 1))
        z = F.conv2d(y, wb.double()) * s3[i * Cout:(i + 1) * Cout].double().view(1, -1, 1, 1) + b3[i * Cout:(i + 1) * Cout].double().view(1, -1, 1, 1)
        outs.append(F.relu(z + res[i * B:(i + 1) * B].double()))
    ref = torch.cat(outs)
    d = [v.to(DEV) for v in (x, w2, s2, b2, w3, s3, b3, res)]
    got = ops.conv3x3_then_1x1(*d)
    assert got is not None, "the library declined an eligible shape"
    again = ops.conv3x3_then_1x1(*d)
    y = ops.conv2d(d[0], d[1], pad=1, scale=d[2], shift=d[3], relu=True)
    two = ops.conv2d(y, d[4], scale=d[5], shift=d[6], residual=d[7], relu=True)
    _close(got, ref.float(), 3e-5)
    assert torch.equal(got, again)
    assert float((got - two).abs().max()) <= 2e-5 * float(ref.abs().max())
    # shapes the kernel does not take come back as None (the caller then issues the two convs)
    assert ops.conv3x3_then_1x1(d[0][:, :, :, :16].contiguous(), *d[1:7], d[7][:, :, :, :16].contiguous()) is None


@pytest.mark.parametrize("N,Cin,H,Cout,G", [(8, 512, 8, 512, 0), (8, 256, 16, 256, 0), (4, 128, 32, 128, 0), (16, 256, 16, 256, 2),
                                           (3, 144, 8, 96, 0), (2, 128, 16, 40, 0), (1, 128, 32, 64, 0), (32, 128, 8, 256, 0)])
def test_conv2d_split_bf16_k_split_over_waves(N, Cin, H, Cout, G):
    """k_conv_bf3_ks (tile_override 10): the pixel-starved deep 3x3 convs of RedNet (rednet.py:190-263) with K split over the
    eight WAVES of a workgroup - wave-private patches, no slabs, no reduction launch.  Same error bar against a float64
    convolution as the tiled split-bf16 kernel and the fp32 MFMA kernel, every epilogue form (scale / shift / residual / ReLU,
    a channel slice of a wider destination), image-grouped weights (the stacked RGB + depth encoders), channel counts that
    leave some waves without a chunk (144 = 9 chunks) and a ragged channel tile (40 outputs), run-to-run identical bits.  Both
    tile sizes at every map width: 64 pixels per workgroup, and 32 where the 64-pixel grid would leave half of the CUs idle
    (the 512-channel 8 x 8 case, the three small ones)."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(N * 100 + Cin + H)
    x = torch.randn(N, Cin, H, H, generator=g)
    wshape = (G, Cout, Cin, 3, 3) if G else (Cout, Cin, 3, 3)
    w = torch.randn(*wshape, generator=g) / (Cin * 9) ** 0.5
    sc, sh = torch.rand(max(G, 1) * Cout, generator=g) + 0.5, torch.randn(max(G, 1) * Cout, generator=g)
    if G:
        B = N // G
        ref0 = torch.cat([F.conv2d(x[i * B:(i + 1) * B].double(), w[i].double(), None, padding=1) for i in range(G)])
        scv = torch.cat([sc[i * Cout:(i + 1) * Cout].view(1, -1, 1, 1).expand(B, -1, 1, 1) for i in range(G)]).double()
        shv = torch.cat([sh[i * Cout:(i + 1) * Cout].view(1, -1, 1, 1).expand(B, -1, 1, 1) for i in range(G)]).double()
    else:
        ref0 = F.conv2d(x.double(), w.double(), None, padding=1)
        scv, shv = sc.double().view(1, -1, 1, 1), sh.double().view(1, -1, 1, 1)
    res = torch.randn(ref0.shape, generator=g)
    ref = F.relu(ref0 * scv + shv + res.double())
    xd, wd = x.to(DEV), w.to(DEV)
    try:
        ops.TILE_OVERRIDE = 10
        got = ops.conv2d(xd, wd, pad=1, scale=sc.to(DEV), shift=sh.to(DEV), residual=res.to(DEV), relu=True)
        if not G:
            plain = ops.conv2d(xd, wd, pad=1, splitk=False)
            again = ops.conv2d(xd, wd, pad=1)
            wide = torch.zeros(N, Cout + 8, H, H, device=DEV)
            ops.conv2d(xd, wd, pad=1, out=wide[:, 4:], out_ctot=Cout + 8)
            ops.TILE_OVERRIDE = 6
            fp32 = ops.conv2d(xd, wd, pad=1, splitk=False)
    finally:
        ops.TILE_OVERRIDE = 0
    _close(got, ref.float(), 3e-5)
    if not G:
        scale = float(ref0.abs().max())
        e_split = float((plain.double().cpu() - ref0).abs().max()) / scale
        e_fp32 = float((fp32.double().cpu() - ref0).abs().max()) / scale
        assert e_split <= 3e-6 and e_split <= 2.0 * e_fp32 + 1e-6, (e_split, e_fp32)
        assert torch.equal(plain, again)
        assert torch.equal(wide[:, 4:4 + Cout], plain)
        assert float(wide[:, :4].abs().max()) == 0.0 and float(wide[:, 4 + Cout:].abs().max()) == 0.0


@pytest.mark.parametrize("N,Cin,H,Cout,G", [(8, 1024, 16, 256, 0), (16, 2048, 8, 512, 2), (3, 528, 8, 40, 0), (5, 512, 4, 96, 0),
                                           (2, 512, 32, 128, 0),
                                           # wave tiles (no K split: 64 ... 256 input channels)
                                           (4, 64, 64, 256, 0), (8, 128, 32, 512, 2), (4, 256, 16, 1024, 0), (3, 80, 8, 40, 0),
                                           (2, 64, 16, 64, 0)])
def test_conv1x1_split_bf16_deep_k_split_over_waves(N, Cin, H, Cout, G):
    """k_conv1x1_bf3_ks (tile_override 11): RedNet's deep-K 1x1 convs (bottleneck reductions 1024 -> 256 / 2048 -> 512, skip
    convs: rednet.py:20-65, 244-248) with K split over the eight waves of a workgroup and the B fragments built in registers
    (no LDS in the K loop).  Error bar of the other split-bf16 kernels against float64, every epilogue form, image-grouped
    weights, chunk counts that leave waves idle (528 = 33 chunks), a ragged channel tile, pixel counts that are not a
    multiple of the 128-pixel tile (3 x 64, 5 x 16), run-to-run identical bits."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(N * 10 + Cin + H)
    x = torch.randn(N, Cin, H, H, generator=g)
    wshape = (G, Cout, Cin, 1, 1) if G else (Cout, Cin, 1, 1)
    w = torch.randn(*wshape, generator=g) / Cin ** 0.5
    sc, sh = torch.rand(max(G, 1) * Cout, generator=g) + 0.5, torch.randn(max(G, 1) * Cout, generator=g)
    if G:
        B = N // G
        ref0 = torch.cat([F.conv2d(x[i * B:(i + 1) * B].double(), w[i].double()) for i in range(G)])
        scv = torch.cat([sc[i * Cout:(i + 1) * Cout].view(1, -1, 1, 1).expand(B, -1, 1, 1) for i in range(G)]).double()
        shv = torch.cat([sh[i * Cout:(i + 1) * Cout].view(1, -1, 1, 1).expand(B, -1, 1, 1) for i in range(G)]).double()
    else:
        ref0 = F.conv2d(x.double(), w.double())
        scv, shv = sc.double().view(1, -1, 1, 1), sh.double().view(1, -1, 1, 1)
    res = torch.randn(ref0.shape, generator=g)
    ref = F.relu(ref0 * scv + shv + res.double())
    xd, wd = x.to(DEV), w.to(DEV)
    try:
        ops.TILE_OVERRIDE = 11
        got = ops.conv2d(xd, wd, scale=sc.to(DEV), shift=sh.to(DEV), residual=res.to(DEV), relu=True)
        if not G:
            plain = ops.conv2d(xd, wd, splitk=False)
            again = ops.conv2d(xd, wd)
            wide = torch.zeros(N, Cout + 8, H, H, device=DEV)
            ops.conv2d(xd, wd, out=wide[:, 4:], out_ctot=Cout + 8)
            ops.TILE_OVERRIDE = 7
            fp32 = ops.conv2d(xd, wd, splitk=False)
    finally:
        ops.TILE_OVERRIDE = 0
    _close(got, ref.float(), 3e-5)
    if not G:
        scale = float(ref0.abs().max())
        e_split = float((plain.double().cpu() - ref0).abs().max()) / scale
        e_fp32 = float((fp32.double().cpu() - ref0).abs().max()) / scale
        assert e_split <= 3e-6 and e_split <= 2.0 * e_fp32 + 1e-6, (e_split, e_fp32)
        assert torch.equal(plain, again)
        assert torch.equal(wide[:, 4:4 + Cout], plain)
        assert float(wide[:, :4].abs().max()) == 0.0 and float(wide[:, 4 + Cout:].abs().max()) == 0.0


@pytest.mark.parametrize("case", ["offset_input_zero_sum_filters", "wide_dynamic_range", "tiny_values"])
def test_conv2d_split_bf16_accuracy_where_fp32_struggles(case):
    """Inputs that expose a lossy product: (1) activations 1000 + N(0, 1) against zero-sum filters - the exact result is O(1)
    while every product is O(1000), so a product error of 2^-16 (a bf16 x 3 scheme without the cross terms) would show as 1e-2
    and fp32's own 2^-24 as ~1e-4; (2) activations and weights spread over 12 decades; (3) values near 1e-30.  The split-bf16
    kernel has to stay within twice the fp32 MFMA kernel's error against float64 (+ 1e-7 of the product scale)."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(7)
    N, Cin, H, Cout, k = 4, 64, 32, 64, 3
    if case == "offset_input_zero_sum_filters":
        x = 1000.0 + torch.randn(N, Cin, H, H, generator=g)
        w = torch.randn(Cout, Cin, k, k, generator=g)
        w = w - w.mean(dim=(1, 2, 3), keepdim=True)
        prod = 1000.0 * float(w.abs().max())
    elif case == "wide_dynamic_range":
        x = torch.randn(N, Cin, H, H, generator=g) * 10.0 ** torch.randint(-6, 7, (N, Cin, H, H), generator=g).float()
        w = torch.randn(Cout, Cin, k, k, generator=g) * 10.0 ** torch.randint(-6, 7, (Cout, Cin, k, k), generator=g).float()
        prod = float(x.abs().max()) * float(w.abs().max())
    else:
        x = torch.randn(N, Cin, H, H, generator=g) * 1e-30
        w = torch.randn(Cout, Cin, k, k, generator=g) * 1e-3
        prod = 1e-33
    ref = F.conv2d(x.double(), w.double(), padding=1)
    try:
        ops.TILE_OVERRIDE = 9
        got = ops.conv2d(x.to(DEV), w.to(DEV), pad=1, splitk=False)
        ops.TILE_OVERRIDE = 6
        fp32 = ops.conv2d(x.to(DEV), w.to(DEV), pad=1, splitk=False)
    finally:
        ops.TILE_OVERRIDE = 0
    e_split = float((got.double().cpu() - ref).abs().max())
    e_fp32 = float((fp32.double().cpu() - ref).abs().max())
    assert e_split <= 2.0 * e_fp32 + 1e-7 * prod * (Cin * k * k) ** 0.5, (case, e_split, e_fp32, prod)


def test_conv2d_split_bf16_one_hot_input_takes_the_three_product_path_with_the_same_bits():
    """A chunk whose staged values are all exact in one bf16 piece (the map CNN's first layer reads occupancy + one-hot
    labels) skips the three products against the zero pieces.  Exact zeros add nothing: the result has to equal, bit for
    bit, the six-product result - forced here by making ONE input value inexact far away from the compared region."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(2)
    N, Cin, H, Cout = 6, 14, 64, 32
    x = (torch.rand(N, Cin, H, H, generator=g) > 0.85).float()
    w = torch.randn(Cout, Cin, 7, 7, generator=g) / (Cin * 49) ** 0.5
    x_full = x.clone()
    x_full[:, :, 0, 0] = 0.3   # an inexact value in the corner of every image: every chunk of its tile takes the six-product path
    ref = F.conv2d(x.double(), w.double(), padding=3)
    try:
        ops.TILE_OVERRIDE = 9
        lite = ops.conv2d(x.to(DEV), w.to(DEV), pad=3, splitk=False)
        full = ops.conv2d(x_full.to(DEV), w.to(DEV), pad=3, splitk=False)
    finally:
        ops.TILE_OVERRIDE = 0
    assert float((lite.double().cpu() - ref).abs().max()) <= 3e-6 * float(ref.abs().max())
    # rows 8.. of every image are computed by tiles whose patches never see pixel (0, 0) in the lite run, and outputs more than
    # 3 pixels away from (0, 0) do not depend on it in the full run: same inputs, different product count, same bits
    assert torch.equal(lite[:, :, 4:8, 4:], full[:, :, 4:8, 4:])


def test_conv2d_split_bf16_pieces_are_exact_and_specials_propagate():
    """The split itself: weights of ONE non-zero tap against an input of ONE non-zero pixel make every output a single
    product a * b - the six piece products must reproduce it to 2^-22 relative for values across the fp32 exponent range
    (three bf16 pieces each; the dropped products are below 2^-23 of a b), and an infinity / a NaN in the input never
    comes out finite (an infinity may come out as NaN: inf times a weight's zero or opposite-signed lower piece)."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(3)
    N, C, H, k = 16, 16, 32, 3
    mags = torch.tensor([1e-30, 1e-12, 1e-3, 1.0, 3.14159, 1e5, 1e15, 1e25])
    x = torch.zeros(N, C, H, H)
    w = torch.zeros(32, C, k, k)
    a = (torch.rand(32, generator=g) + 0.5) * (torch.randint(0, 2, (32,), generator=g) * 2 - 1)
    w[:, 5, 1, 1] = a * 1e-6
    vals = (torch.rand(N, H, H, generator=g) + 0.5) * mags[torch.randint(0, 8, (N, H, H), generator=g)]
    x[:, 5] = vals
    try:
        ops.TILE_OVERRIDE = 9
        y = ops.conv2d(x.to(DEV), w.to(DEV), pad=1, splitk=False).cpu()
        x2 = x.clone()
        x2[0, 5, 3, 3], x2[1, 5, 4, 4] = float("inf"), float("nan")
        y2 = ops.conv2d(x2.to(DEV), w.to(DEV), pad=1, splitk=False).cpu()
    finally:
        ops.TILE_OVERRIDE = 0
    ref = (a.double() * 1e-6).view(1, -1, 1, 1) * vals.double().unsqueeze(1)
    rel = ((y.double() - ref).abs() / ref.abs()).max()
    assert float(rel) <= 2.0 ** -22, float(rel)
    assert (~torch.isfinite(y2[0, :, 3, 3])).all() and torch.isnan(y2[1, :, 4, 4]).all()
    assert torch.isfinite(y2[2:]).all()


def test_conv2d_split_bf16_image_grouped_weights_and_epilogue_statistics():
    """Two weight sets over the two halves of the batch in one launch (RedNet's stacked encoders) == two launches, bit for
    bit; and the {count, mean, M2} partials the epilogue leaves per (128-pixel segment, channel) merge to the statistics
    of what was stored."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(11)
    N, Cin, H, Cout = 8, 64, 64, 64
    x = torch.randn(N, Cin, H, H, generator=g).to(DEV)
    w2 = (torch.randn(2, Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    try:
        ops.TILE_OVERRIDE = 9
        both = ops.conv2d(x, w2, pad=1, splitk=False)
        lo = ops.conv2d(x[:4].contiguous(), w2[0].contiguous(), pad=1, splitk=False)
        hi = ops.conv2d(x[4:].contiguous(), w2[1].contiguous(), pad=1, splitk=False)
        stats = []
        y = ops.conv2d(x, w2[0].contiguous(), pad=1, shift=b, stats=stats)
    finally:
        ops.TILE_OVERRIDE = 0
    assert torch.equal(both[:4], lo) and torch.equal(both[4:], hi)
    assert stats, "the split-bf16 kernel leaves its partials"
    bn = torch.nn.BatchNorm2d(Cout).to(DEV).train()
    sc, sh, sm, sr = (torch.empty(Cout, device=DEV) for _ in range(4))
    ops.bn_stats_from_partials(stats[0][0], stats[0][1], bn, sc, sh, sm, sr)
    _close(sm, y.double().mean((0, 2, 3)).float(), 1e-5)
    _close(sr, (1.0 / torch.sqrt(y.double().var((0, 2, 3), unbiased=False) + 1e-5)).float(), 1e-4)


@pytest.mark.parametrize("N,Cin,H,Cout,stride", [(16, 256, 16, 1024, 1), (8, 200, 32, 128, 1), (4, 128, 64, 512, 2), (16, 2048, 8, 64, 1)])
def test_conv2d_split_bf16_1x1_form(N, Cin, H, Cout, stride):
    """The kernel's 1x1 form (four 16-channel chunks staged per barrier pair, stride 1 or 2, ragged channel count): not on
    by default - the float4-staged fp32 GEMM is faster on these shapes - but it has to be right."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(N, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=stride)
    try:
        ops.TILE_OVERRIDE = 9
        got = ops.conv2d(x.to(DEV), w.to(DEV), stride=stride, shift=b.to(DEV))
        one = ops.conv2d(x.to(DEV), w.to(DEV), stride=stride, shift=b.to(DEV), splitk=False)
    finally:
        ops.TILE_OVERRIDE = 0
    for y in (got, one):
        assert float((y.double().cpu() - ref).abs().max()) <= 3e-6 * float(ref.abs().max())


@pytest.mark.parametrize("N,Cin,H,Cout,G", [(16, 256, 64, 512, 2), (16, 512, 32, 1024, 2), (16, 1024, 16, 2048, 2), (3, 64, 16, 96, 0)])
def test_stride2_1x1_conv_through_the_gather_equals_the_strided_read(N, Cin, H, Cout, G):
    """ops.conv2d gathers the input of a stride-2 1x1 conv (RedNet's downsample branches, rednet.py:226-232, image-grouped
    over the stacked encoders) into a dense quarter-size tensor and runs the stride-1 register-built split-bf16 kernel on it;
    IVLN_S2_GATHER=0 reads the input strided (the tiled 1x1 form / the fp32 GEMM).  Both against float64 (3e-6 of the largest
    output); the gather itself is a copy (bit-equal to slicing)."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(N + Cin + Cout)
    x = torch.randn(N, Cin, H, H, generator=g)
    ws = [torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5 for _ in range(max(G, 1))]
    sc, sh = torch.rand(max(G, 1) * Cout, generator=g) + 0.5, torch.randn(max(G, 1) * Cout, generator=g)
    per = N // max(G, 1)
    ref = torch.cat([F.conv2d(x[i * per:(i + 1) * per].double(), ws[i].double(), stride=2) * sc[i * Cout:(i + 1) * Cout].double().view(1, -1, 1, 1)
                     + sh[i * Cout:(i + 1) * Cout].double().view(1, -1, 1, 1) for i in range(max(G, 1))])
    w = (torch.stack(ws) if G else ws[0]).contiguous().to(DEV)
    xd = x.to(DEV)
    assert torch.equal(ops.pool2d(xd, 1, 2, 0, "max"), xd[:, :, ::2, ::2])
    outs = {}
    for gather in (True, False):
        prev, ops.S2_GATHER = ops.S2_GATHER, gather
        try:
            outs[gather] = ops.conv2d(xd, w, stride=2, scale=sc.to(DEV), shift=sh.to(DEV))
        finally:
            ops.S2_GATHER = prev
    for y in outs.values():
        assert y.shape == ref.shape
        assert float((y.double().cpu() - ref).abs().max()) <= 3e-6 * float(ref.abs().max())


@pytest.mark.parametrize(
    "N,Cin,H,Cout",
    [(6, 14, 64, 32),      # map CNN layer 1: 32 x 512 tile, two column tiles (686 columns: ragged), strips of two rows
     (8, 32, 32, 64),      # layer 2: 64 x 512
     (16, 64, 16, 128),    # layer 3: 128 x 256, strips of eight rows
     (33, 128, 8, 128),    # layer 4: strips of two whole images, odd image count (a half-empty last strip)
     (5, 20, 16, 70)],     # ragged everything: 70 channels, 980 columns
)
def test_conv2d_weight_gradient_split_bf16_kernel(N, Cin, H, Cout):
    """k_wgrad_bf3 forced (tile_override 9): the 7x7 weight gradient with dy and x as three bf16 pieces each.  Against the
    float64 weight gradient: as close as the fp32 MFMA weight-gradient kernel (bar: twice its error + 1e-6 of the largest
    element, and 5e-6 of it outright), deterministic, and with one slab or many."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(N + Cin + Cout)
    x = torch.randn(N, Cin, H, H, generator=g)
    dy = torch.randn(N, Cout, H, H, generator=g)
    w = torch.zeros(Cout, Cin, 7, 7, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w, padding=3).backward(dy.double())
    ref = w.grad
    xd, dyd = x.to(DEV), dy.to(DEV)
    try:
        ops.TILE_OVERRIDE = 9
        got = ops.conv2d_bwd_weight(dyd, xd, 7, 7, pad=3)
        again = ops.conv2d_bwd_weight(dyd, xd, 7, 7, pad=3)
        ops.TILE_OVERRIDE = 6
        fp32 = ops.conv2d_bwd_weight(dyd, xd, 7, 7, pad=3)
    finally:
        ops.TILE_OVERRIDE = 0
    scale = float(ref.abs().max())
    e_split = float((got.double().cpu() - ref).abs().max()) / scale
    e_fp32 = float((fp32.double().cpu() - ref).abs().max()) / scale
    assert e_split <= 5e-6 and e_split <= 2.0 * e_fp32 + 1e-6, (e_split, e_fp32)
    assert torch.equal(got, again)


@pytest.mark.parametrize("N,Cin,Cout", [(6, 14, 32), (64, 14, 32), (5, 9, 20)])
def test_conv2d_weight_gradient_with_bf16_exact_input_staged_as_one_piece(N, Cin, Cout):
    """k_wgrad_bf3<..., XE> (ivln_gemm_desc.split_ok = 2): the map CNN's first layer reads one-hot map features
    (map_encoder.py:60-75), whose values are exact in bf16 - x is staged as its upper 16 bits alone, the kernel needs half the
    LDS and two thirds of the registers.  Against the float64 weight gradient as close as the three-piece form (both drop only
    products with a zero factor) and as the fp32 kernel, deterministic, forced and by default dispatch; and a value that is NOT
    exact - the promise broken - gives NaNs, not a wrong gradient."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(N + Cin + Cout)
    x = (torch.rand(N, Cin, 64, 64, generator=g) < 0.3).float()
    dy = torch.randn(N, Cout, 64, 64, generator=g)
    w = torch.zeros(Cout, Cin, 7, 7, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w, padding=3).backward(dy.double())
    ref = w.grad
    xd, dyd = x.to(DEV), dy.to(DEV)
    try:
        ops.TILE_OVERRIDE = 9
        one = ops.conv2d_bwd_weight(dyd, xd, 7, 7, pad=3, x_exact_bf16=True)
        again = ops.conv2d_bwd_weight(dyd, xd, 7, 7, pad=3, x_exact_bf16=True)
        three = ops.conv2d_bwd_weight(dyd, xd, 7, 7, pad=3)
        ops.TILE_OVERRIDE = 6
        fp32 = ops.conv2d_bwd_weight(dyd, xd, 7, 7, pad=3)
        ops.TILE_OVERRIDE = 0
        default = ops.conv2d_bwd_weight(dyd, xd, 7, 7, pad=3, x_exact_bf16=True)
    finally:
        ops.TILE_OVERRIDE = 0
    scale = float(ref.abs().max())
    e1, e3, e32 = (float((t.double().cpu() - ref).abs().max()) / scale for t in (one, three, fp32))
    assert e1 <= 5e-6 and e1 <= 2.0 * e32 + 1e-6 and e1 <= 2.0 * e3 + 1e-6, (e1, e3, e32)
    assert torch.equal(one, again)
    assert float((default.double().cpu() - ref).abs().max()) <= 5e-6 * scale
    bad = xd.clone()
    bad[N // 2, Cin // 2, 17, 23] = 1.0 + 2.0 ** -12  # not a bf16 value
    try:
        ops.TILE_OVERRIDE = 9
        broken = ops.conv2d_bwd_weight(dyd, bad, 7, 7, pad=3, x_exact_bf16=True)
        honest = ops.conv2d_bwd_weight(dyd, bad, 7, 7, pad=3)
    finally:
        ops.TILE_OVERRIDE = 0
    assert bool(torch.isnan(broken).any()) and not bool(torch.isnan(honest).any())


def test_conv2d_split_bf16_refuses_what_it_is_not_built_for():
    from ivln_ce_amd import ops
    from ivln_ce_amd._lib import IvlnError

    x = torch.randn(8, 32, 32, 32, device=DEV)
    for w, kw in [(torch.randn(32, 32, 7, 7, device=DEV), dict(stride=2, pad=3)),       # strided 7x7 (stride 2 exists for 3x3 only: round 6)
                  (torch.randn(32, 32, 3, 3, device=DEV), dict(stride=2, pad=0)),         # stride-2 3x3 without its padding
                  (torch.randn(32, 32, 3, 3, device=DEV), dict(pad=0)),                   # not same-size
                  (torch.randn(32, 32, 1, 1, device=DEV), dict(pad=0)),                   # 1x1 into 32 channels (below its 64-channel tiles)
                  (torch.randn(32, 32, 5, 5, device=DEV), dict(pad=2))]:                  # 5x5
        try:
            ops.TILE_OVERRIDE = 9
            with pytest.raises(IvlnError):
                ops.conv2d(x, w, **kw)
        finally:
            ops.TILE_OVERRIDE = 0


def test_conv2d_direct_packed_and_unpacked_weights_agree():
    """The pre-arranged-weight variant of the direct kernel (default) against the variant that gathers OIHW
    weights itself: bit-identical (same MFMA order), and the packed cache follows in-place weight updates."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(5)
    for (N, Cin, H, W, Cout, k) in [(3, 14, 64, 64, 32, 7), (5, 64, 16, 16, 128, 7), (2, 64, 40, 24, 70, 3)]:
        x = torch.randn(N, Cin, H, W, generator=g).to(DEV)
        w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(DEV)
        try:
            ops.TILE_OVERRIDE = 6
            a = ops.conv2d(x, w, pad=k // 2)
            ops.PACK_WEIGHTS = False
            b = ops.conv2d(x, w, pad=k // 2)
            ops.PACK_WEIGHTS = True
            w.mul_(2.0)  # in-place update: the tensor version changes, the cache must re-pack
            c = ops.conv2d(x, w, pad=k // 2)
        finally:
            ops.TILE_OVERRIDE, ops.PACK_WEIGHTS = 0, True
        assert torch.equal(a, b)
        _close(c, 2.0 * a, 1e-5)


def test_conv2d_direct_refuses_ineligible_shapes():
    from ivln_ce_amd import ops
    from ivln_ce_amd._lib import IvlnError

    x = torch.randn(1, 3, 16, 16, device=DEV)   # 3 input channels: not a multiple of the channel chunk
    w = torch.randn(8, 3, 3, 3, device=DEV)
    ref = F.conv2d(x.cpu(), w.cpu(), padding=1)
    _close(ops.conv2d(x, w, pad=1), ref, 2e-5)  # auto: falls through to the implicit GEMM
    try:
        ops.TILE_OVERRIDE = 6
        with pytest.raises(IvlnError):
            ops.conv2d(x, w, pad=1)
    finally:
        ops.TILE_OVERRIDE = 0


@pytest.mark.parametrize(
    "N,Cin,H,W,Cout,k,p",
    [
        (5, 14, 64, 64, 32, 7, 3),     # map CNN layer 1 (686 columns: ragged last column tile)
        (4, 32, 32, 32, 64, 7, 3),
        (6, 64, 16, 16, 128, 7, 3),
        (7, 128, 8, 8, 128, 7, 3),     # two images per pixel tile, odd image count
        (9, 16, 4, 4, 48, 7, 3),
        (3, 64, 40, 24, 70, 3, 1),     # ragged spatial tile, 70 channels
        (10, 256, 4, 4, 40, 3, 1),
        (2, 8, 21, 19, 24, 3, 0),      # no padding
        (2, 6, 12, 12, 8, 7, 6),       # pad = KS-1
    ],
)
def test_conv2d_weight_gradient_direct_kernel(N, Cin, H, W, Cout, k, p):
    """k_wgrad_direct (forced, tile_override 6) against torch CPU autograd and the implicit GEMM path;
    sums run over N*Ho*Wo pixels: tolerance 2e-4 relative to the gradient's max magnitude."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(N * 10 + Cin + k)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g, requires_grad=True)
    y = F.conv2d(x, w, None, stride=1, padding=p)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    try:
        ops.TILE_OVERRIDE = 6
        got = ops.conv2d_bwd_weight(dy.to(DEV), x.to(DEV), k, k, stride=1, pad=p)
        ops.TILE_OVERRIDE = 1
        gemm = ops.conv2d_bwd_weight(dy.to(DEV), x.to(DEV), k, k, stride=1, pad=p)
    finally:
        ops.TILE_OVERRIDE = 0
    scale = w.grad.abs().max().item()
    _close(got / scale, w.grad / scale, 2e-5, rtol=2e-4)
    _close(got / scale, gemm / scale, 2e-5, rtol=2e-4)


@pytest.mark.parametrize(
    "rows,K,O",
    [
        (640, 256, 512),     # 64x128 tiles
        (100, 64, 96),       # 64x64 tiles, ragged rows / outputs
        (4096, 1184, 24),    # O <= 32: 32x128 tiles
        (24, 3072, 512),     # rows <= 32: 128x32 tiles, deep K (split-K)
        (333, 36, 260),      # K barely more than one 32-deep tile
    ],
)
def test_vector_load_gemm_all_operand_layouts(rows, K, O):
    """gemm_vec.hip forced (tile_override 7) on its four operand-layout combinations - linear forward
    ([m][k] x [n][k]), input gradient ([k][m] x [n][k]), weight gradient ([k][m] x [k][n]) and the 1x1
    convolution ([m][k] x [k][n]) - against torch CPU fp32; tolerance 3e-5 on O(1) results."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(rows + K + O)
    x = torch.randn(rows, K, generator=g)
    w = torch.randn(O, K, generator=g) / K ** 0.5
    b = torch.randn(O, generator=g)
    dy = torch.randn(rows, O, generator=g) / O ** 0.5
    try:
        ops.TILE_OVERRIDE = 7
        y = ops.linear_gemm(x.to(DEV), w.to(DEV), b.to(DEV), relu=True)
        dx = ops.linear_bwd_input(dy.to(DEV), w.to(DEV))
        dw = ops.linear_bwd_weight((dy / rows ** 0.5).to(DEV), x.to(DEV))
    finally:
        ops.TILE_OVERRIDE = 0
    _close(y, F.relu(x @ w.t() + b), 3e-5)
    _close(dx, dy @ w, 3e-5)
    _close(dw, (dy / rows ** 0.5).t() @ x, 3e-5)


@pytest.mark.parametrize("N,Cin,H,W,Cout", [(3, 128, 32, 32, 64), (8, 64, 64, 64, 256), (2, 256, 4, 4, 1024),
                                            (4, 1024, 4, 4, 128), (5, 48, 6, 10, 20)])
def test_vector_load_gemm_conv1x1(N, Cin, H, W, Cout):
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(N + Cin + Cout)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    ref0 = F.conv2d(x, w)
    res = torch.randn_like(ref0)
    try:
        ops.TILE_OVERRIDE = 7
        got = ops.conv2d(x.to(DEV), w.to(DEV), scale=sc.to(DEV), shift=sh.to(DEV), residual=res.to(DEV), relu=True)
        dfr = ops.conv2d(x.to(DEV), w.to(DEV), defer=True)
    finally:
        ops.TILE_OVERRIDE = 0
    _close(got, F.relu(ref0 * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + res), 3e-5)
    slabs = dfr.ws[: dfr.splits * Cout * N * H * W].view(dfr.splits, Cout, N, H, W).sum(0)
    _close(slabs.permute(1, 0, 2, 3), ref0, 3e-5)


@pytest.mark.parametrize("N,Cin,H,W,Cout,groups", [(2, 64, 64, 64, 256, 0), (1, 64, 128, 128, 64, 0), (4, 128, 32, 32, 512, 0),
                                                   (16, 256, 16, 16, 1024, 0), (1, 64, 128, 128, 52, 0), (8, 64, 64, 64, 256, 2),
                                                   (2, 256, 64, 64, 64, 0), (16, 128, 32, 32, 128, 2)])
def test_streaming_conv1x1_short_k_matches_torch_and_the_tiled_kernel(N, Cin, H, W, Cout, groups):
    """conv1x1_stream.hip (weights in registers, activations streamed as 16-byte loads, four interleaved pixel tiles per
    load; taken by ivln_gemm_f32 for K = 64 / 128 / 256 over >= 4096 pixels) against F.conv2d and against the float4-
    staged tiled kernel (tile_override 7) on RedNet's 1x1 shapes (rednet.py:190-263) - with the fused epilogue (folded
    BatchNorm, residual, ReLU), a channel count that does not fill the last row tile, image-grouped weights (the stacked
    rgb / depth encoders) and an output that is a channel slice of a wider buffer."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(N + Cin + Cout)
    x = torch.randn(N, Cin, H, W, generator=g)
    G = max(groups, 1)
    w = torch.randn(G, Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    sc, sh = torch.rand(G, Cout, generator=g) + 0.5, torch.randn(G, Cout, generator=g)
    per = N // G
    ref0 = torch.cat([F.conv2d(x[i * per:(i + 1) * per], w[i]) for i in range(G)])
    res = torch.randn_like(ref0)
    scb = torch.cat([sc[i].view(1, -1, 1, 1).expand(per, -1, 1, 1) for i in range(G)])
    shb = torch.cat([sh[i].view(1, -1, 1, 1).expand(per, -1, 1, 1) for i in range(G)])
    want = F.relu(ref0 * scb + shb + res)
    wd = (w if groups else w[0]).to(DEV).contiguous()
    scd, shd = (sc if groups else sc[0]).to(DEV).contiguous(), (sh if groups else sh[0]).to(DEV).contiguous()
    try:
        ops.TILE_OVERRIDE = 8  # insist on the streaming kernel (by default K = 64 stays with the tiled one)
        got = ops.conv2d(x.to(DEV), wd, scale=scd, shift=shd, residual=res.to(DEV), relu=True)
    finally:
        ops.TILE_OVERRIDE = 0
    _close(got, want, 3e-5)
    _close(ops.conv2d(x.to(DEV), wd, scale=scd, shift=shd, residual=res.to(DEV), relu=True), want, 3e-5)  # the default route
    try:
        ops.TILE_OVERRIDE = 7
        tiled = ops.conv2d(x.to(DEV), wd, scale=scd, shift=shd, residual=res.to(DEV), relu=True)
    finally:
        ops.TILE_OVERRIDE = 0
    _close(got, tiled, 2e-5)  # (same products, different summation order: k pairs in sequence vs 32-deep K tiles)
    if not groups:  # into a channel slice of a wider NCHW buffer, no epilogue
        wide = torch.full((N, Cout + 8, H, W), -3.0, device=DEV)
        try:
            ops.TILE_OVERRIDE = 8
            ops.conv2d(x.to(DEV), wd, out=wide[:, 8:], out_ctot=Cout + 8)
        finally:
            ops.TILE_OVERRIDE = 0
        _close(wide[:, 8:], ref0, 3e-5)
        assert float(wide[:, :8].max()) == -3.0


def test_vector_load_gemm_refuses_unaligned_shapes():
    from ivln_ce_amd import ops
    from ivln_ce_amd._lib import IvlnError

    x, w = torch.randn(40, 50, device=DEV), torch.randn(12, 50, device=DEV)  # K = 50: rows not 16-byte multiples
    _close(ops.linear_gemm(x, w), x.cpu() @ w.cpu().t(), 3e-5)  # auto: scalar-gather kernel
    try:
        ops.TILE_OVERRIDE = 7
        with pytest.raises(IvlnError):
            ops.linear_gemm(x, w)
    finally:
        ops.TILE_OVERRIDE = 0


@pytest.mark.parametrize("N,Cin,H,Cout,form", [(8, 512, 8, 256, "ks"), (8, 256, 16, 128, "ks"), (8, 128, 32, 64, "tiled"), (8, 64, 64, 64, "tiled"),
                                                (3, 128, 16, 24, "ks"), (2, 48, 32, 40, "tiled")])
def test_conv_transpose2d_stride2_stacked_classes_on_the_split_bf16_kernels(N, Cin, H, Cout, form):
    """RedNet's stride-2 3x3 transposed convs (rednet.py:152-181; the four decoder stages at 8 envs + two ragged shapes) as ONE
    launch of the stacked parity classes on the split-bf16 kernels (csrc/conv_bf3.hip, KS = 2 -> IVLN_D_NCHW_UP2X4): K split
    over the waves where pixels are few (tile_override 10), the tiled kernel beyond (tile_override 9 lets the library choose,
    24 pins the 64 x 128 tile).  Against the float64 transposed conv with scale / shift / residual / ReLU: 3e-6 of the largest
    output, at or below twice the fp32 direct kernel's error; reproducible; the default dispatch takes the same kernels."""
    import ctypes as C

    from ivln_ce_amd import ops
    from ivln_ce_amd._lib import lib

    g = torch.Generator().manual_seed(N + Cin + Cout)
    x = torch.randn(N, Cin, H, H, generator=g)
    w = torch.randn(Cin, Cout, 3, 3, generator=g) / (Cin * 9 / 4) ** 0.5
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    ref = F.conv_transpose2d(x.double(), w.double(), None, stride=2, padding=1, output_padding=1)
    res = torch.randn(ref.shape, generator=g)
    ref = F.relu(ref * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1) + res.double())
    cls = ops.convt_s2_classes(w.to(DEV), 1)
    stacked = ops.convt_s2_stack(cls)
    args = dict(scale=sc.to(DEV), shift=sh.to(DEV), residual=res.to(DEV), relu=True, stacked=stacked)
    L = lib()
    L.ivln_conv_split_counters.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.c_int]
    L.ivln_conv_split_kinds.argtypes = [C.POINTER(C.c_longlong), C.c_int]

    def run(override):
        L.ivln_conv_split_counters(None, None, 1)
        L.ivln_conv_split_kinds(None, 1)
        ops.TILE_OVERRIDE = override
        try:
            y = ops.conv_transpose2d_s2(x.to(DEV), cls, **args)
        finally:
            ops.TILE_OVERRIDE = 0
        f, n, kinds = C.c_double(0.0), C.c_longlong(0), (C.c_longlong * 4)()
        L.ivln_conv_split_counters(C.byref(f), C.byref(n), 0)
        L.ivln_conv_split_kinds(kinds, 0)
        return y, f.value, n.value, list(kinds)

    scale = float(ref.abs().max())
    fp32, _, n0, _ = run(6)  # the fp32 direct kernel (2 x 2 window)
    assert n0 == 0
    e_fp32 = float((fp32.double().cpu() - ref).abs().max()) / scale
    got, flops, n1, kinds = run(10 if form == "ks" else 9)
    assert n1 == 1 and kinds[1 if form == "ks" else 0] == 1, (n1, kinds)
    # the tally prices the nine real taps, not the sixteen executed ones (ivln_gemm_desc.real_taps)
    assert abs(flops - 2.0 * Cout * Cin * 9 * N * H * H) < 1e-6 * flops
    e = float((got.double().cpu() - ref).abs().max()) / scale
    assert e <= 3e-6 and e <= 2.0 * e_fp32 + 1e-6, (e, e_fp32)
    again, _, _, _ = run(10 if form == "ks" else 9)
    assert torch.equal(got, again)
    if form == "tiled":
        pinned, _, n2, k2 = run(24)
        assert n2 == 1 and k2[0] == 1 and float((pinned.double().cpu() - ref).abs().max()) / scale <= 3e-6
    if N == 8:  # RedNet's own shapes: what the default dispatch launches
        dflt, _, n3, k3 = run(0)
        assert n3 == 1 and float((dflt.double().cpu() - ref).abs().max()) / scale <= 3e-6, (n3, k3)


@pytest.mark.parametrize("N,Cin,H,Cout,form", [(8, 512, 8, 256, 12), (8, 256, 16, 128, 13), (8, 128, 32, 64, 13), (8, 64, 64, 64, 13), (8, 64, 128, 13, 13),
                                                (3, 96, 8, 20, 13), (2, 512, 16, 24, 12)])
def test_conv_transpose2d_2x2_stride2_on_the_split_bf16_1x1_kernels(N, Cin, H, Cout, form):
    """RedNet's 2 x 2 stride-2 transposed convs (the four upsampling branches and the final 64 -> 13 deconv, rednet.py:239-245,
    217-218) as the stacked one-tap classes on the register-built split-bf16 1x1 kernels, which store the 2 x 2 output blocks
    themselves (k_conv1x1_bf3_ks: both forms, tile_override 12 = K split over the waves, 13 = wave tiles).  Against the
    float64 transposed conv with scale / shift / residual / ReLU: 3e-6 of the largest output, at or below twice the fp32
    kernel's error; reproducible."""
    import ctypes as C

    from ivln_ce_amd import ops
    from ivln_ce_amd._lib import lib

    g = torch.Generator().manual_seed(N + Cin + Cout + form)
    x = torch.randn(N, Cin, H, H, generator=g)
    w = torch.randn(Cin, Cout, 2, 2, generator=g) / Cin ** 0.5
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    ref = F.conv_transpose2d(x.double(), w.double(), None, stride=2)
    res = torch.randn(ref.shape, generator=g)
    ref = F.relu(ref * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1) + res.double())
    cls = ops.convt_s2_classes(w.to(DEV), 0)
    stacked = ops.convt_s2_stack(cls)
    args = dict(scale=sc.to(DEV), shift=sh.to(DEV), residual=res.to(DEV), relu=True, stacked=stacked)
    L = lib()
    L.ivln_conv_split_counters.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.c_int]

    def run(override):
        L.ivln_conv_split_counters(None, None, 1)
        ops.TILE_OVERRIDE = override
        try:
            y = ops.conv_transpose2d_s2(x.to(DEV), cls, **args)
        finally:
            ops.TILE_OVERRIDE = 0
        n = C.c_longlong(0)
        L.ivln_conv_split_counters(None, C.byref(n), 0)
        return y, n.value

    scale = float(ref.abs().max())
    fp32, n0 = run(7)  # the float4-staged fp32 GEMM
    assert n0 == 0
    e_fp32 = float((fp32.double().cpu() - ref).abs().max()) / scale
    got, n1 = run(form)
    assert n1 == 1
    e = float((got.double().cpu() - ref).abs().max()) / scale
    assert e <= 3e-6 and e <= 2.0 * e_fp32 + 1e-6, (e, e_fp32)
    assert torch.equal(got, run(form)[0])


@pytest.mark.parametrize("N,Cin,H,W,Cout,k,p,op", [(2, 64, 8, 8, 32, 3, 1, 1), (3, 32, 16, 12, 13, 2, 0, 0),
                                                   (2, 16, 5, 7, 8, 3, 1, 1), (1, 128, 32, 32, 64, 3, 1, 1)])
def test_conv_transpose2d_stride2_parity_classes(N, Cin, H, W, Cout, k, p, op):
    """Stride-2 transposed conv as four output-parity sub-convolutions (RedNet upsampling blocks) against
    torch CPU, with the fused scale/shift + residual + ReLU epilogue; tolerance 3e-5."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(N + Cin + k)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cin, Cout, k, k, generator=g) / (Cin * k * k / 4) ** 0.5
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    ref0 = F.conv_transpose2d(x, w, None, stride=2, padding=p, output_padding=op)
    assert ref0.shape[2:] == (2 * H, 2 * W)
    res = torch.randn_like(ref0)
    cls = ops.convt_s2_classes(w.to(DEV), p)
    assert cls is not None and len(cls) == 4
    got = ops.conv_transpose2d_s2(x.to(DEV), cls, scale=sc.to(DEV), shift=sh.to(DEV), residual=res.to(DEV), relu=True)
    _close(got, F.relu(ref0 * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + res), 3e-5)
    # the same four classes stacked along M: ONE launch (D_NCHW_UP2X4), zero-padded taps for k=3
    stacked = ops.convt_s2_stack(cls)
    assert stacked.shape[0] == 4 * Cout
    got1 = ops.conv_transpose2d_s2(x.to(DEV), cls, scale=sc.to(DEV), shift=sh.to(DEV), residual=res.to(DEV), relu=True,
                                   stacked=stacked)
    _close(got1, F.relu(ref0 * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + res), 3e-5)
    # k=3, p=0 needs a negative input offset for the even rows: not decomposed this way
    assert ops.convt_s2_classes(torch.randn(4, 4, 3, 3, device=DEV), 0) is None


@pytest.mark.parametrize("N,C,H,W,G", [(2, 128, 16, 16, 16), (3, 64, 8, 8, 16), (2, 32, 5, 7, 4)])
def test_groupnorm_with_second_normalised_operand(N, C, H, W, G):
    """y = relu(GN(x) + GN2(x2)): the bottleneck's downsample GroupNorm folded into the block's last
    GroupNorm launch, x2 given as deferred conv slabs or as a plain tensor; tolerance 2e-5."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(N + C)
    x, x2 = torch.randn(N, C, H, W, generator=g), 2.0 * torch.randn(N, C, H, W, generator=g) + 0.5
    g1, b1, g2, b2 = (torch.randn(C, generator=g) for _ in range(4))
    ref = F.relu(F.group_norm(x, G, g1, b1, 1e-5) + F.group_norm(x2, G, g2, b2, 1e-5))
    got = ops.groupnorm(x.to(DEV), g1.to(DEV), b1.to(DEV), G, 1e-5, relu=True, x2=x2.to(DEV), gamma2=g2.to(DEV),
                        beta2=b2.to(DEV))
    _close(got, ref, 2e-5)
    if (H * W) % 4 == 0:
        # x2 as the raw slabs of a deferred 1x1 conv in the second workspace
        xin = torch.randn(N, 24, H, W, generator=g)
        w = torch.randn(C, 24, 1, 1, generator=g) / 24 ** 0.5
        ds = ops.conv2d(xin.to(DEV), w.to(DEV), defer=True, ws_slot=1)
        junk = ops.conv2d(xin.to(DEV), w.to(DEV), defer=True)  # a later deferred conv must not disturb slot 1
        ref2 = F.relu(F.group_norm(x, G, g1, b1, 1e-5) + F.group_norm(F.conv2d(xin, w), G, g2, b2, 1e-5))
        got2 = ops.groupnorm(x.to(DEV), g1.to(DEV), b1.to(DEV), G, 1e-5, relu=True, x2=ds, gamma2=g2.to(DEV),
                             beta2=b2.to(DEV))
        _close(got2, ref2, 3e-5)
        assert junk.splits >= 1


def test_attention_short_axis_two_sets_one_launch():
    """k_attn_small: both 16-position attentions of the rollout head (depth and map features, shared query)
    in one launch, against torch and against the two-kernel path; tolerance 2e-5."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(21)
    rows, Ck, I = 5, 256, 16
    q = torch.randn(rows, Ck, generator=g)
    kv0, kv1 = torch.randn(rows, Ck + 192, I, generator=g), torch.randn(rows, Ck + 128, I, generator=g)

    def ref(kv):
        k, v = kv[:, :Ck], kv[:, Ck:]
        return torch.einsum("ni,nci->nc", F.softmax(torch.einsum("nc,nci->ni", q, k) * 0.0625, 1), v)

    d0, d1 = kv0.to(DEV), kv1.to(DEV)
    x2 = torch.zeros(rows, 400, device=DEV)
    ops.attn_small2(q.to(DEV), d0[:, :Ck], d0[:, Ck:], x2[:, 10:202], d1[:, :Ck], d1[:, Ck:], x2[:, 250:378], 0.0625)
    _close(x2[:, 10:202], ref(kv0), 2e-5)
    _close(x2[:, 250:378], ref(kv1), 2e-5)
    assert float(x2[:, :10].abs().max()) == 0.0 and float(x2[:, 202:250].abs().max()) == 0.0
    two = torch.zeros(rows, 192, device=DEV)
    ops.attn(q.to(DEV), d0[:, :Ck], d0[:, Ck:], None, 0.0625, two)
    _close(x2[:, 10:202], two, 1e-6)


@pytest.mark.parametrize("rows", [1, 4, 8, 11, 40])
def test_action_head_linear_argmax_one_launch(rows):
    """k_linear_argmax == torch.argmax of the fp32 logits and == argmax_rows(linear(x)) (random logits: no
    near-ties), first maximum on exact ties."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, 512, generator=g)
    w, b = torch.randn(4, 512, generator=g) * 0.05, torch.randn(4, generator=g) * 0.01
    got = ops.linear_argmax(x.to(DEV), w.to(DEV), b.to(DEV))
    assert got.shape == (rows, 1) and got.dtype == torch.int64
    ref = torch.argmax(x @ w.t() + b, dim=1, keepdim=True)
    assert torch.equal(got.cpu(), ref)
    if rows <= 16:
        two = ops.argmax_rows(ops.linear(x.to(DEV), w.to(DEV), b.to(DEV)))
        assert torch.equal(got, two)
    # first maximum wins on exact ties (torch.argmax / distribution.mode())
    xt = torch.zeros(2, 512)
    assert ops.linear_argmax(xt.to(DEV), w.to(DEV), torch.zeros(4).to(DEV)).flatten().tolist() == [0, 0]


@pytest.mark.parametrize("N,Cin,H,W,Cout,k,s,p,override", [
    (5, 32, 32, 32, 64, 3, 1, 1, 6),     # direct conv, 40 x 1 workgroups
    (3, 14, 64, 64, 32, 7, 1, 3, 6),     # direct 7x7, grid not a multiple of 8
    (7, 96, 20, 12, 72, 1, 1, 0, 7),     # vector-load GEMM, ragged grid (27 x 2 tiles)
    (4, 1024, 4, 4, 128, 3, 1, 1, 1),    # scalar-gather GEMM with split-K (z dimension in the remap)
    (3, 64, 33, 17, 48, 3, 2, 1, 0),     # strided conv, default dispatch
])
def test_xcd_aware_workgroup_remap_is_a_pure_permutation(N, Cin, H, W, Cout, k, s, p, override):
    """ivln_gemm_desc.no_xcd_remap: handing each XCD a contiguous range of tiles only permutes which workgroup
    computes which tile - outputs (and split-K slab order) are bit-identical to the identity mapping, for grids
    whose size is not a multiple of the 8 XCDs too."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(N + Cin + Cout + k)
    x = torch.randn(N, Cin, H, W, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(DEV)
    sh = torch.randn(Cout, generator=g).to(DEV)
    out = {}
    try:
        ops.TILE_OVERRIDE = override
        for flag in (False, True):
            ops.NO_XCD_REMAP = flag
            out[flag] = ops.conv2d(x, w, stride=s, pad=p, shift=sh, relu=True).clone()
            gw = ops.conv2d_bwd_weight(out[flag], x, k, k, s, p) if (s == 1 and override in (0, 6)) else None
            out[(flag, "dw")] = gw.clone() if gw is not None else None
    finally:
        ops.TILE_OVERRIDE = 0
        ops.NO_XCD_REMAP = False
    assert torch.equal(out[False], out[True])
    if out[(False, "dw")] is not None:
        assert torch.equal(out[(False, "dw")], out[(True, "dw")])
    _close(out[False], F.relu(F.conv2d(x.cpu(), w.cpu(), sh.cpu(), stride=s, padding=p)), 3e-5)


@pytest.mark.parametrize("B,Cin,H,W,Cout,k,s,p", [(4, 64, 16, 16, 64, 3, 1, 1), (2, 256, 8, 8, 64, 1, 1, 0), (3, 64, 8, 8, 256, 1, 1, 0),
                                                   (2, 128, 16, 16, 128, 3, 2, 1), (2, 256, 16, 16, 512, 1, 2, 0),
                                                   (8, 128, 32, 32, 128, 3, 2, 1),
                                                   (8, 512, 8, 8, 512, 3, 1, 1), (1, 64, 32, 32, 64, 3, 1, 1)])
def test_image_grouped_conv_matches_two_separate_convs(B, Cin, H, W, Cout, k, s, p):
    """ivln_gemm_desc.grp_imgs: images [0, B) with weight set 0 and [B, 2B) with set 1 in ONE launch (RedNet's RGB and
    depth encoders stacked) == the two convs run separately, with per-set folded-BN scale / shift, residual, ReLU.
    Covers the direct 3x3 kernel (packed weights per set), the vector-load 1x1 GEMM and the strided gather GEMM."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(B * 100 + Cin + k)
    x = torch.randn(2 * B, Cin, H, W, generator=g).to(DEV)
    w = (torch.randn(2, Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(DEV)
    sc, sh = (1 + 0.2 * torch.randn(2 * Cout, generator=g)).to(DEV), torch.randn(2 * Cout, generator=g).to(DEV)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    res = torch.randn(2 * B, Cout, Ho, Wo, generator=g).to(DEV)
    got = ops.conv2d(x, w, stride=s, pad=p, scale=sc, shift=sh, residual=res, relu=True)
    for gi in range(2):
        sl = slice(gi * B, (gi + 1) * B)
        ref = ops.conv2d(x[sl], w[gi].contiguous(), stride=s, pad=p, scale=sc[gi * Cout:(gi + 1) * Cout].contiguous(),
                         shift=sh[gi * Cout:(gi + 1) * Cout].contiguous(), residual=res[sl], relu=True)
        tref = F.relu(F.conv2d(x[sl].cpu(), w[gi].cpu(), None, s, p) * sc[gi * Cout:(gi + 1) * Cout].cpu().view(1, -1, 1, 1)
                      + sh[gi * Cout:(gi + 1) * Cout].cpu().view(1, -1, 1, 1) + res[sl].cpu())
        _close(ref, tref, 2e-5)
        assert float((got[sl] - ref).abs().max()) < 2e-5, gi


@pytest.mark.parametrize("rows,Cc,H,W,Ckv,O", [(4, 192, 4, 4, 384, 128), (8, 128, 4, 4, 384, 128), (1, 192, 4, 4, 384, 128),
                                                (3, 20, 3, 4, 10, 7)])
def test_kv_projection_and_flatten_linear_in_one_launch(rows, Cc, H, W, Ckv, O):
    """ivln_kv_linear_f32: nn.Conv1d(C, Ckv, 1) over the positions + nn.Flatten -> nn.Linear -> ReLU of the same
    feature map (the MapCMA head's two consumers of an encoder output) against torch CPU; written into a column
    slice of a wider state buffer like the policy does."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(rows + Cc)
    feat = torch.randn(rows, Cc, H, W, generator=g)
    wkv, bkv = torch.randn(Ckv, Cc, 1, generator=g) / Cc ** 0.5, torch.randn(Ckv, generator=g)
    wl, bl = torch.randn(O, Cc * H * W, generator=g) / (Cc * H * W) ** 0.5, torch.randn(O, generator=g)
    ref_kv = F.conv1d(feat.view(rows, Cc, H * W), wkv, bkv)
    ref_l = F.relu(F.linear(feat.flatten(1), wl, bl))
    wide = torch.full((rows, O + 9), 5.0, device=DEV)
    kv = ops.kv_linear(feat.to(DEV), wkv.to(DEV), bkv.to(DEV), wl.to(DEV), bl.to(DEV), wide[:, 4:4 + O])
    assert kv is not None and tuple(kv.shape) == (rows, Ckv, 1, H * W)
    _close(kv.view(rows, Ckv, H * W), ref_kv, 2e-5)
    _close(wide[:, 4:4 + O], ref_l, 2e-5)
    assert float((wide[:, :4] - 5.0).abs().max()) == 0.0 and float((wide[:, 4 + O:] - 5.0).abs().max()) == 0.0
    assert ops.kv_linear(torch.randn(9, 8, 2, 2, device=DEV), wkv[:4, :8].contiguous().to(DEV), None, torch.randn(3, 32, device=DEV),
                         None, torch.empty(9, 3, device=DEV)) is None  # more than 8 rows: the caller's GEMM path


def _gru_seq_case(T, N, seed):
    H = 512
    g = torch.Generator().manual_seed(seed)
    gi = (torch.randn(T * N, 3 * H, generator=g) * 0.5).to(DEV)
    h0 = (torch.randn(N, H, generator=g) * 0.5).to(DEV)
    masks = (torch.rand(T * N, generator=g) > 0.1).to(torch.uint8)
    masks[:N] = 0
    w_hh = (torch.randn(3 * H, H, generator=g) * 0.05).to(DEV)
    b_hh = (torch.randn(3 * H, generator=g) * 0.1).to(DEV)
    d_out = (torch.randn(T * N, H, generator=g) * 0.1).to(DEV)
    return H, gi, h0, masks.to(DEV), w_hh, b_hh, d_out


def _gru_seq_run(ops, case, T, N, persistent, backward):
    H, gi, h0, masks, w_hh, b_hh, d_out = case
    ops.SEQ_PERSISTENT = persistent
    try:
        out = torch.full((T * N, H), float("nan"), device=DEV)   # poisoned: a row nobody wrote must show
        state = torch.full((N, H), float("nan"), device=DEV)
        saves = tuple(torch.full((T * N, H), float("nan"), device=DEV) for _ in range(4))
        ops.gru_seq(gi, h0, masks, w_hh, b_hh, out, state, T, N, saves)
        res = [out, state, *saves]
        if backward:
            whh_t = w_hh.t().contiguous()
            dgi = torch.full((T * N, 3 * H), float("nan"), device=DEV)
            dgh = torch.full((T * N, 3 * H), float("nan"), device=DEV)
            hp = torch.full((T * N, H), float("nan"), device=DEV)
            dhz = torch.empty((N, H), device=DEV)
            ops.gru_seq_bwd(d_out, *saves, out, h0, masks, whh_t, T, N, dgi, dgh, hp, dhz)
            res += [dgi, dgh, hp]
        ops.check_seq_sync()
        return res
    finally:
        ops.SEQ_PERSISTENT = True


@pytest.mark.parametrize("T,N", [(64, 8), (7, 5), (33, 16), (5, 3), (12, 40), (2, 1)])
def test_persistent_sequence_gru_matches_per_step_launches_and_torch(T, N):
    """ivln_cma_seq_fwd/bwd as ONE persistent launch each (csrc/gru_seq.hip) against the launch-per-timestep path of
    the same entry points (sync_ws = NULL): same summation order, 1e-6 (measured 2e-7: the compiler contracts the
    element formulas differently in the two kernels); the
    forward also against torch's own GRU arithmetic (masked nn.GRUCell recurrence, 1e-5).  N > 16 exercises the
    forward-only envelope (the BPTT kernel falls back to launches there)."""
    from ivln_ce_amd import ops

    case = _gru_seq_case(T, N, seed=T * 100 + N)
    H, gi, h0, masks, w_hh, b_hh, d_out = case
    backward = True
    a = _gru_seq_run(ops, case, T, N, persistent=False, backward=backward)
    b = _gru_seq_run(ops, case, T, N, persistent=True, backward=backward)
    for x, y in zip(a, b):
        assert torch.isfinite(y).all()
        _close(y, x, atol=1e-6)
    # torch reference of the recurrence (gate order r, z, n; hidden zeroed where mask == 0)
    h = h0.clone()
    outs = []
    for t in range(T):
        rows = slice(t * N, (t + 1) * N)
        h = h * masks[rows].float().unsqueeze(1)
        gh = h @ w_hh.t() + b_hh
        gi_t = gi[rows]
        r = torch.sigmoid(gi_t[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi_t[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi_t[:, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - z) * n + z * h
        outs.append(h)
    _close(b[0], torch.cat(outs), atol=1e-5)
    _close(b[1], h, atol=1e-5)


def test_persistent_sequence_gru_is_reproducible_under_load():
    """The in-launch exchange (write-through stores, one counter per step, sc1 loads) must not depend on timing or
    placement: 30 back-to-back runs, half of them beside a bandwidth-heavy stream, give the first run's bits, forward
    and backward, and no spin ever times out."""
    from ivln_ce_amd import ops

    T, N = 64, 8
    case = _gru_seq_case(T, N, seed=9)
    first = _gru_seq_run(ops, case, T, N, persistent=True, backward=True)
    side = torch.cuda.Stream()
    big = torch.empty(64 << 20, device=DEV)
    for rep in range(30):
        if rep % 2:
            with torch.cuda.stream(side):
                for _ in range(8):
                    big.mul_(1.0001)
        again = _gru_seq_run(ops, case, T, N, persistent=True, backward=True)
        for x, y in zip(first, again):
            assert torch.equal(x, y), rep
    torch.cuda.synchronize()


@pytest.mark.parametrize("rows", [1, 4, 8, 13])
def test_sampled_action_head_inverse_cdf_mixing_and_skip_rule(rows):
    """ivln_linear_sample_f32: a = first o with cumsum(exp(l - max))[o] > u * sum (torch restatement, exact on the
    kernel's own logits), `where(u_beta < beta, expert, a)`, 0 where the expert says -1; and the draws follow
    softmax(logits) (20 000 uniforms, 3 sigma)."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(rows)
    K, O = 512, 4
    x = torch.randn(rows, K, generator=g).to(DEV)
    w = (torch.randn(O, K, generator=g) * 0.05).to(DEV)
    b = torch.randn(O, generator=g).to(DEV)
    u = torch.rand(rows, generator=g).to(DEV)
    ub = torch.rand(rows, generator=g).to(DEV)
    expert = torch.randint(-1, 4, (rows,), generator=g).double().to(DEV)
    logits = torch.empty(rows, O, device=DEV)
    a = ops.linear_sample(x, w, b, u, ub, 0.4, expert, logits_out=logits)
    _close(logits, x.cpu() @ w.cpu().t() + b.cpu(), atol=1e-5)
    p = torch.exp(logits - logits.max(1, keepdim=True).values)
    pick = (torch.cumsum(p, 1) > (u * p.sum(1)).unsqueeze(1)).float().argmax(1)
    ref = torch.where(ub < 0.4, expert.long(), pick)
    ref = torch.where(expert.long() == -1, torch.zeros_like(ref), ref)
    assert torch.equal(a.view(-1), ref)
    plain = ops.linear_sample(x, w, b, u)   # no mixing: the bare draw
    assert torch.equal(plain.view(-1), pick)
    # distribution: one row, many uniforms
    n = 20000
    xs = x[:1].expand(n, K).contiguous()
    draws = ops.linear_sample(xs, w, b, torch.rand(n, generator=g).to(DEV)).view(-1)
    probs = torch.softmax(logits[0].double().cpu(), 0)
    freq = torch.bincount(draws.cpu(), minlength=O).double() / n
    assert ((freq - probs).abs() < 3 * (probs * (1 - probs) / n).sqrt() + 1e-3).all(), (freq, probs)


def test_family_timing_sink_counts_launches_and_leaves_results_alone():
    """ivln_family_timing_begin / _end (include/ivln_hip.h): every MFMA-family launch in between carries its own start /
    stop event; the sum of the durations and the number of launches come back, the kernels' results do not change."""
    import ctypes as C

    from ivln_ce_amd import ops
    from ivln_ce_amd._lib import check, lib

    L = lib()
    L.ivln_family_timing_begin.argtypes = [C.c_int]
    L.ivln_family_timing_end.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 64, 32, 32, generator=g).to(DEV)
    w3 = (torch.randn(64, 64, 3, 3, generator=g) / 24).to(DEV)
    w1 = (torch.randn(128, 64, 1, 1, generator=g) / 8).to(DEV)
    ref3, ref1 = ops.conv2d(x, w3, pad=1), ops.conv2d(x, w1)
    check(L.ivln_family_timing_begin(2), "begin")
    assert L.ivln_family_timing_begin(2) != 0  # one user at a time
    got3, got1 = ops.conv2d(x, w3, pad=1), ops.conv2d(x, w1)
    extra = ops.conv2d(x, w1)  # a third launch: beyond max_launches, goes out untimed
    ms, n, dropped = C.c_double(0), C.c_int(0), C.c_int(0)
    check(L.ivln_family_timing_end(C.byref(ms), C.byref(n), C.byref(dropped)), "end")
    assert n.value == 2 and dropped.value == 1 and 0.0 < ms.value < 5.0
    assert torch.equal(got3, ref3) and torch.equal(got1, ref1) and torch.equal(extra, ref1)
    assert L.ivln_family_timing_end(C.byref(ms), C.byref(n), C.byref(dropped)) != 0  # not armed any more


@pytest.mark.parametrize("N,Cin,H,W,Cout,k,s,p", [(8, 512, 8, 8, 512, 3, 1, 1), (4, 256, 16, 16, 256, 3, 1, 1), (2, 64, 64, 64, 64, 3, 1, 1),
                                                   (2, 64, 64, 64, 256, 1, 1, 0), (4, 3, 128, 128, 64, 7, 2, 3)])
def test_wide_and_narrow_epilogues_store_the_same_values(N, Cin, H, W, Cout, k, s, p):
    """NCHW tiles through LDS as 16-byte stores (k_gemm_vec, k_conv_direct, the 16-byte split-K reduction) against the
    MFMA layout's 4-byte stores (`ivln_gemm_desc.no_wide_epilogue`): bit-identical outputs, fused scale / shift /
    residual / ReLU included, split or not."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(N + Cin + k)
    x = torch.randn(N, Cin, H, W, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(DEV)
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).to(DEV), torch.randn(Cout, generator=g).to(DEV)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    res = torch.randn(N, Cout, Ho, Wo, generator=g).to(DEV)
    saved = ops.NO_WIDE_EPILOGUE
    out = []
    try:
        for narrow in (False, True):
            ops.NO_WIDE_EPILOGUE = narrow
            out.append(ops.conv2d(x, w, stride=s, pad=p, scale=sc, shift=sh, residual=res, relu=True))
    finally:
        ops.NO_WIDE_EPILOGUE = saved
    assert torch.equal(out[0], out[1])
    ref = F.relu(F.conv2d(x.cpu(), w.cpu(), None, s, p) * sc.cpu().view(1, -1, 1, 1) + sh.cpu().view(1, -1, 1, 1) + res.cpu())
    _close(out[0], ref, 3e-5)


def test_pixel_starved_conv_leaves_no_partials_and_the_block_falls_back_to_the_statistics_pass():
    """3 images of 16x16 are 12 direct-conv blocks: ivln_gemm_f32 sends the 7x7 conv to the split-K implicit GEMM, which
    has no statistics epilogue and says so (`stat_tiles` = 0).  CBRA.forward_hip then computes the BatchNorm statistics
    from a pass over the conv's output: same block output and running statistics as torch's train-mode BatchNorm."""
    import copy

    from ivln_ce_amd import ops
    from ivln_ce_amd.encoders import CBRA

    torch.manual_seed(11)
    x = torch.randn(3, 32, 16, 16)
    blk = CBRA(32, 128).train()
    ref_blk = copy.deepcopy(blk)
    ref = ref_blk.conv(x)
    blk = blk.to(DEV)
    stats = []
    ops.conv2d(x.to(DEV), blk.conv[0].weight, stride=1, pad=3, shift=blk.conv[0].bias, stats=stats)
    assert stats == []
    with torch.no_grad():
        got = blk.forward_hip(x.to(DEV))
    _close(got, ref, 3e-5)
    _close(blk.conv[1].running_mean, ref_blk.conv[1].running_mean, 1e-6)
    _close(blk.conv[1].running_var, ref_blk.conv[1].running_var, 1e-6)


def test_colsum_queue_equals_per_matrix_colsums_bit_for_bit():
    """ivln_colsum_multi_f32 (ops.ColsumQueue): many column sums in two launches, same partial / final order as
    ivln_colsum_f32 per matrix - bias gradients of one update (autograd's grad_output.sum(0))."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(3)
    shapes = [(512, 1536), (40960, 50), (512, 128), (7, 5), (4097, 513), (300, 64)]
    xs = [torch.randn(r, c, generator=g).to(DEV) for r, c in shapes]
    wide = torch.randn(512, 640, generator=g).to(DEV)
    xs.append(wide[:, 128:])  # a column slice: row stride != cols
    ref = [ops.colsum(x).clone() for x in xs]
    q = ops.ColsumQueue()
    outs = [q.add(x) for x in xs]
    q.flush()
    for o, r, x in zip(outs, ref, xs):
        assert torch.equal(o, r)
        _close(o, x.double().sum(0).float(), 2e-3 * max(1.0, x.shape[0] ** 0.5 / 20))


@pytest.mark.parametrize("N,C,H,W", [(64, 32, 64, 64), (64, 64, 32, 32), (512, 128, 8, 8), (256, 128, 16, 16)])
def test_batchnorm_statistics_from_the_conv_epilogue_match_the_pass_over_its_output(N, C, H, W):
    """conv2d(..., stats=[]) leaves per-tile {count, mean, M2} of what it stores; ivln_bn_stats_from_partials_f32 merges
    them into the scale / shift / saved statistics / running statistics that ivln_bn_train_stats_f32 computes from a
    pass over the conv's output (map_encoder.py:13-20 in train mode)."""
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(N + C)
    Cin = 32
    x = torch.randn(N, Cin, H, W, generator=g).to(DEV)
    w = (torch.randn(C, Cin, 7, 7, generator=g) / (Cin * 49) ** 0.5).to(DEV)
    b = torch.randn(C, generator=g).to(DEV)
    stats = []
    y = ops.conv2d(x, w, stride=1, pad=3, shift=b, stats=stats)
    res = []
    for use_partials in (False, True):
        bn = torch.nn.BatchNorm2d(C).to(DEV).train()
        sc, sh, sm, sr = (torch.empty(C, device=DEV) for _ in range(4))
        if use_partials:
            assert stats, "a trajectory-batch shape must take the direct kernel's wide epilogue and leave its partials"
            ops.bn_stats_from_partials(stats[0][0], stats[0][1], bn, sc, sh, sm, sr)
        else:
            ops.bn_train_stats(y, bn, sc, sh, sm, sr)
        res.append([t.clone() for t in (sc, sh, sm, sr, bn.running_mean, bn.running_var)])
    ref_mean, ref_var = y.double().mean((0, 2, 3)), y.double().var((0, 2, 3), unbiased=False)
    for r in res:
        _close(r[2], ref_mean.float(), 1e-5)
        _close(r[3], (1.0 / torch.sqrt(ref_var + 1e-5)).float(), 1e-4)
    for a, b in zip(res[0], res[1]):  # scale, shift, saved mean / rstd, running mean / var
        _close(a, b, 1e-5)


@pytest.mark.parametrize("B,H,W", [(8, 256, 256), (2, 6, 10), (1, 2, 2)])
def test_rgb_normalize_at_the_networks_size_is_the_resize_kernels_bits(B, H, W):
    """RedNet's input prep (mapper.py:715-736, 788-793) on frames that already have the network's size: the bilinear weights
    are exactly 1 and 0, so ivln_rgb_resize_normalize_f32 takes a four-pixels-per-thread kernel (k_rgb_normalize_x4) - against
    the general kernel on the same frames (forced by a destination that is not 16-byte aligned) bit for bit, and against
    (u8 / 255 - mean) / std in torch."""
    import ctypes as C

    from ivln_ce_amd import ops
    from ivln_ce_amd._lib import lib

    g = torch.Generator().manual_seed(B + H)
    rgb = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8).to(DEV)
    fast = ops.rgb_resize_normalize(rgb, H, W)
    L = lib()
    buf = torch.zeros(B * 3 * H * W + 4, dtype=torch.float32, device=DEV)
    rc = L.ivln_rgb_resize_normalize_f32(C.c_void_p(rgb.data_ptr()), B, H, W, H, W, C.c_void_p(buf.data_ptr() + 4), C.c_void_p(ops.stream_ptr()))
    assert rc == 0
    torch.cuda.synchronize()
    general = buf[1:1 + B * 3 * H * W].view(B, 3, H, W)
    assert torch.equal(fast, general)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    want = (rgb.cpu().permute(0, 3, 1, 2).float() / 255.0 - mean) / std
    assert float((fast.cpu() - want).abs().max()) <= 1e-6
