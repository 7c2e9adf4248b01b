"""Pins the CPU twin oracle/libivln_ref.so (same `ivln_*` symbols as the device library, host pointers): its GEMM /
conv against torch.nn.functional, its GroupNorm against F.group_norm, its mapper entry points against the goldens of
the reference's own MappingModule.  The GPU half - the same entry point run in BOTH libraries on the same bytes - is
tests/test_gpu_twin.py."""
import ctypes as C
import glob
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import twin as T

G = os.path.join(os.path.dirname(__file__), "golden")


def test_twin_exports_the_header_symbols_it_claims():
    L = T.twin()
    header = open(os.path.join(T.ROOT, "include", "ivln_hip.h")).read()
    for name in ["ivln_strerror", "ivln_version", "ivln_gemm_f32", "ivln_groupnorm_f32", "ivln_mapper_create",
                 "ivln_mapper_destroy", "ivln_mapper_reset", "ivln_mapper_frames", "ivln_mapper_step",
                 "ivln_mapper_step_posed", "ivln_mapper_step_begin", "ivln_mapper_step_finish",
                 "ivln_mapper_known_begin", "ivln_mapper_load_known", "ivln_mapper_known_raster", "ivln_mapper_status",
                 "ivln_mapper_world_export", "ivln_gn_conv_f32", "ivln_nconv_f32", "ivln_kv_linear_f32",
                 "ivln_cma_step_fwd", "ivln_cma_step_ws_floats"]:
        assert hasattr(L, name) and name + "(" in header, name
    assert L.ivln_strerror(-5) == b"unsupported configuration"


@pytest.mark.parametrize("N,Cin,H,W,Cout,k,s,p", [(2, 5, 9, 7, 6, 3, 1, 1), (1, 8, 8, 8, 4, 1, 1, 0), (2, 8, 9, 9, 4, 1, 2, 0),
                                                   (1, 2, 12, 10, 3, 7, 2, 3), (2, 4, 6, 6, 5, 3, 2, 1)])
def test_twin_conv_matches_torch(N, Cin, H, W, Cout, k, s, p):
    g = torch.Generator().manual_seed(N + Cin + k)
    x, w = torch.randn(N, Cin, H, W, generator=g), torch.randn(Cout, Cin, k, k, generator=g)
    sc, sh = torch.randn(Cout, generator=g), torch.randn(Cout, generator=g)
    ref = F.conv2d(x, w, None, s, p)
    res = torch.randn(ref.shape, generator=g)
    ref = F.relu(ref * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + res)
    xs, ws, out = T.np32(x), T.np32(w), np.zeros(ref.shape, np.float32)
    scs, shs, rs = T.np32(sc), T.np32(sh), T.np32(res)
    d = T.conv_desc(T.hp, xs, ws, out, stride=s, pad=p, scale=scs, shift=shs, residual=rs, relu=True)
    L = T.twin()
    T.check(L, L.ivln_gemm_f32(C.byref(d), None), "twin gemm")
    assert np.allclose(out, ref.numpy(), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("N,Cin,H,W,Cout,k", [(2, 6, 5, 4, 3, 3), (1, 4, 3, 6, 5, 2), (2, 3, 4, 5, 2, 3)])
def test_twin_stacked_transposed_conv_matches_torch(N, Cin, H, W, Cout, k):
    """RedNet's stride-2 transposed convs (rednet.py:210-216, 262-279) as ONE GEMM over the four output-parity classes:
    2x2 window (k=3) or 1x1 (k=2) into IVLN_D_NCHW_UP2X4, rows 4*channel + class."""
    c = T.convt_case(N * 5 + Cin + k, N, Cin, H, W, Cout, k)
    out = np.zeros(c["ref"].shape, np.float32)
    d = T.convt_desc(T.hp, c, c["x"], c["w"], out, c["scale"], c["shift"], c["residual"])
    L = T.twin()
    T.check(L, L.ivln_gemm_f32(C.byref(d), None), "twin stacked transposed conv")
    assert np.allclose(out, c["ref"], atol=2e-5, rtol=1e-5)


def test_twin_linear_and_image_grouped_conv_match_torch():
    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(3)
    L = T.twin()
    x, w, b = torch.randn(7, 12, generator=g), torch.randn(5, 12, generator=g), torch.randn(5, generator=g)
    xs, ws, bs, out = T.np32(x), T.np32(w), T.np32(b), np.zeros((7, 5), np.float32)
    d = T.desc_type()()
    d.A, d.B, d.D = T.hp(ws), T.hp(xs), T.hp(out)
    d.M, d.N, d.K = 5, 7, 12
    d.amode, d.bmode, d.dmode = ops.A_MK, ops.B_NK, ops.D_DENSE
    d.lda, d.ldb, d.sDm, d.sDn, d.HoWo = 12, 12, 1, 5, 1
    d.shift, d.splits = T.hp(bs), 1
    T.check(L, L.ivln_gemm_f32(C.byref(d), None), "twin linear")
    assert np.allclose(out, F.linear(x, w, b).numpy(), atol=1e-5)
    # two weight sets over the halves of a stacked batch (RedNet's encoders)
    B, Cin, Cout = 2, 4, 3
    x2, w2 = torch.randn(2 * B, Cin, 4, 8, generator=g), torch.randn(2, Cout, Cin, 3, 3, generator=g)
    sc, sh = torch.randn(2 * Cout, generator=g), torch.randn(2 * Cout, generator=g)
    ref = torch.cat([F.conv2d(x2[i * B:(i + 1) * B], w2[i], None, 1, 1) * sc[i * Cout:(i + 1) * Cout].view(1, -1, 1, 1)
                     + sh[i * Cout:(i + 1) * Cout].view(1, -1, 1, 1) for i in range(2)])
    xs, ws, out = T.np32(x2), T.np32(w2), np.zeros(ref.shape, np.float32)
    scs, shs = T.np32(sc), T.np32(sh)
    d = T.conv_desc(T.hp, xs, ws, out, pad=1, scale=scs, shift=shs, grouped=True)
    T.check(L, L.ivln_gemm_f32(C.byref(d), None), "twin grouped conv")
    assert np.allclose(out, ref.numpy(), atol=2e-5, rtol=1e-5)


def test_twin_groupnorm_matches_torch():
    g = torch.Generator().manual_seed(5)
    x, gamma, beta = torch.randn(3, 32, 4, 4, generator=g) * 2 + 0.5, torch.randn(32, generator=g), torch.randn(32, generator=g)
    res = torch.randn(3, 32, 4, 4, generator=g)
    ref = F.relu(F.group_norm(x, 16, gamma, beta, 1e-5) + res).numpy()
    xs, gs, bs, rs, out = T.np32(x), T.np32(gamma), T.np32(beta), T.np32(res), np.zeros((3, 32, 4, 4), np.float32)
    L = T.twin()
    T.check(L, L.ivln_groupnorm_f32(T.hp(xs), T.hp(gs), T.hp(bs), T.hp(rs), T.hp(out), 3, 32, 16, 16, 1e-5, 1, 0, 0, 1, 0, 0, 0,
                                    None, None, None), "twin groupnorm")
    assert np.allclose(out, ref, atol=2e-5)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "mapper_b*.npz"))), ids=lambda p: os.path.basename(p)[7:-4])
def test_twin_mapper_entry_points_match_reference_golden(path):
    """ivln_mapper_create / frames / step / status of the TWIN (the wrappers the GPU diff test calls) on the goldens
    produced by the reference's own MappingModule."""
    g = np.load(path)
    B, H, W = int(g["B"]), int(g["H"]), int(g["W"])
    L = T.twin()
    h = C.c_void_p()
    T.check(L, L.ivln_mapper_create(B, H, W, float(np.deg2rad(90.0 * H / W)), 6.4, 6.4, 0.1, 0, 0, C.byref(h)), "create")
    for t in range(int(g["steps"])):
        pose, orient = T.np32(g[f"pose_{t}"]), np.ascontiguousarray(g[f"orientation_{t}"], np.float64)
        Tm, rot = np.zeros((B, 4, 4), np.float32), np.zeros((B, 3, 3), np.float32)
        T.check(L, L.ivln_mapper_frames(T.hp(pose), T.hp(orient), B, T.hp(Tm), T.hp(rot), None), "frames")
        depth = T.np32(g[f"depth_{t}"]).reshape(B, H, W)
        labels = np.ascontiguousarray(g[f"semantic12_{t}"], np.uint8).reshape(B, H, W)
        nd = np.ascontiguousarray(g[f"not_done_{t}"], np.uint8).reshape(-1)
        occ, sem = np.zeros((B, 64, 64), np.uint8), np.zeros((B, 64, 64), np.uint8)
        T.check(L, L.ivln_mapper_step(h, T.hp(depth), T.hp(labels), T.hp(Tm), T.hp(pose), T.hp(rot), T.hp(nd), B, T.hp(occ),
                                      T.hp(sem), None), "step")
        assert np.array_equal(occ, g[f"occ_{t}"]) and np.array_equal(sem, g[f"sem_{t}"]), t
        n = C.c_int64(0)
        T.check(L, L.ivln_mapper_status(h, C.byref(n), None), "status")
        assert n.value == int(g[f"world_n_{t}"])
    L.ivln_mapper_destroy(h)


# ---- round-2 entry points -------------------------------------------------------------------------------------------
def _gn(x, G, gamma, beta):
    return F.group_norm(x, G, torch.from_numpy(gamma), torch.from_numpy(beta), 1e-5)


@pytest.mark.parametrize("second,residual,pool,ka,sa,Cb,sb", [(False, False, False, 3, 1, 0, 1), (True, False, False, 1, 1, 12, 2),
                                                              (False, True, False, 3, 2, 0, 1), (False, False, True, 1, 1, 8, 1)])
def test_twin_gn_conv_matches_torch(second, residual, pool, ka, sa, Cb, sb):
    """GroupNorm (+ second operand / residual / ReLU / MaxPool) and the NEXT conv as per-group partial slabs
    (ivln_gn_conv_f32) against F.group_norm + F.max_pool2d + F.conv2d over each group's channel slice."""
    N, Cc, H, W, G, Ca = 2, 8, 6, 8, 4, 6
    c = T.gn_conv_case(7 + ka + Cb, N, Cc, H, W, G, 3, second, residual, pool, ka, sa, Ca, Cb, sb)
    act = np.zeros((N, Cc, c["Hp"], c["Wp"]), np.float32)
    ya = np.zeros((G, Ca, N * c["Ho"] * c["Wo"]), np.float32)
    yb = np.zeros((G, max(Cb, 1), N * max(c["Hb"] * c["Wb"], 1)), np.float32)
    L = T._sigs2(T.twin())
    d = T.gn_conv_desc(T.hp, c, act, ya, yb)
    T.check(L, L.ivln_gn_conv_f32(C.byref(d), None), "twin gn_conv")
    x = torch.from_numpy(c["x"].sum(0)).view(Cc, N, H, W).permute(1, 0, 2, 3)
    ref = _gn(x, G, c["gamma"], c["beta"])
    if second:
        ref = ref + _gn(torch.from_numpy(c["x2"].sum(0)).view(Cc, N, H, W).permute(1, 0, 2, 3), G, c["gamma2"], c["beta2"])
    if residual:
        ref = ref + torch.from_numpy(c["residual"])
    ref = F.relu(ref)
    if pool:
        ref = F.max_pool2d(ref, 3, 2, 1)
    assert np.allclose(act, ref.numpy(), atol=2e-5)
    cpg = Cc // G
    for g in range(G):
        sl = slice(g * cpg, (g + 1) * cpg)
        ra = F.conv2d(ref[:, sl], torch.from_numpy(c["wa"][:, sl]), None, sa, ka // 2)
        assert np.allclose(ya[g], ra.permute(1, 0, 2, 3).reshape(Ca, -1).numpy(), atol=3e-5), g
        if Cb:
            rb = F.conv2d(ref[:, sl], torch.from_numpy(c["wb"][:, sl]), None, sb, 0)
            assert np.allclose(yb[g], rb.permute(1, 0, 2, 3).reshape(Cb, -1).numpy(), atol=3e-5), g


@pytest.mark.parametrize("second,residual,ka,sa,Cb,sb", [(False, False, 3, 1, 0, 1), (True, True, 1, 1, 8, 2), (False, False, 3, 2, 0, 1)])
def test_twin_nconv_matches_torch(second, residual, ka, sa, Cb, sb):
    """GroupNorm-on-load conv (ivln_nconv_f32): statistics merged from per-strip (count, mean, M2) partials, conv
    outputs in the [C][N][H][W] raw layout, and the OUTPUT's own partials - against torch on the full tensors."""
    N, Cc, H, W, G, Ca, ga, gb = 2, 8, 8, 8, 4, 8, 4, 2
    c = T.nconv_case(11 + ka + Cb, N, Cc, H, W, G, second, residual, ka, sa, Ca, ga, Cb, gb, sb, rows=4)
    act = np.zeros((N, Cc, H, W), np.float32) if sa == 1 else None
    ya, sta = np.zeros((Ca, N, c["Ho"], c["Wo"]), np.float32), np.zeros((c["strips"], N, ga, 3), np.float32)
    yb = np.zeros((max(Cb, 1), N, max(c["Hb"], 1), max(c["Wb"], 1)), np.float32)
    stb = np.zeros((c["strips"], N, gb, 3), np.float32)
    L = T._sigs2(T.twin())
    d = T.nconv_desc(T.hp, c, act, ya, sta, yb, stb)
    T.check(L, L.ivln_nconv_f32(C.byref(d), None), "twin nconv")
    ref = _gn(torch.from_numpy(c["x"]).permute(1, 0, 2, 3), G, c["gamma"], c["beta"])
    if second:
        ref = ref + _gn(torch.from_numpy(c["x2"]).permute(1, 0, 2, 3), G, c["gamma2"], c["beta2"])
    if residual:
        ref = ref + torch.from_numpy(c["residual"])
    ref = F.relu(ref)
    if act is not None:
        assert np.allclose(act, ref.numpy(), atol=3e-5)
    ra = F.conv2d(ref, torch.from_numpy(c["wa"]), None, sa, ka // 2).permute(1, 0, 2, 3)
    assert np.allclose(ya, ra.numpy(), atol=5e-5)
    mean, var = T.merged(sta)
    v = ra.reshape(ga, Ca // ga, N, -1).permute(2, 0, 1, 3).reshape(N, ga, -1).double()
    assert np.allclose(mean, v.mean(2).numpy(), atol=1e-5) and np.allclose(var, v.var(2, unbiased=False).numpy(), atol=1e-5)
    if Cb:
        rb = F.conv2d(ref, torch.from_numpy(c["wb"]), None, sb, 0).permute(1, 0, 2, 3)
        assert np.allclose(yb, rb.numpy(), atol=5e-5)
        mean, var = T.merged(stb)
        v = rb.reshape(gb, Cb // gb, N, -1).permute(2, 0, 1, 3).reshape(N, gb, -1).double()
        assert np.allclose(mean, v.mean(2).numpy(), atol=1e-5) and np.allclose(var, v.var(2, unbiased=False).numpy(), atol=1e-5)


def test_twin_kv_linear_and_cma_step_match_torch():
    """ivln_kv_linear_f32 against F.conv1d + F.linear, ivln_cma_step_fwd against the head's arithmetic in torch
    (map_cma_policy.py:305-353 on the folded operands: nn.GRUCell, masked softmax attention, compress, nn.GRUCell)."""
    rs = np.random.RandomState(4)
    rows, Cc, P, Ckv, O = 3, 6, 16, 10, 5
    feat = rs.randn(rows, Cc, P).astype(np.float32)
    wkv, bkv = rs.randn(Ckv, Cc).astype(np.float32), rs.randn(Ckv).astype(np.float32)
    wl, bl = (rs.randn(O, Cc * P) / 8).astype(np.float32), rs.randn(O).astype(np.float32)
    kv, lin = np.zeros((rows, Ckv, P), np.float32), np.zeros((rows, O), np.float32)
    L = T._sigs2(T.twin())
    T.check(L, L.ivln_kv_linear_f32(T.hp(feat), rows, Cc, P, T.hp(wkv), T.hp(bkv), Ckv, T.hp(kv), T.hp(wl), T.hp(bl), O, 1,
                                    T.hp(lin), O, None), "twin kv_linear")
    ft = torch.from_numpy(feat)
    assert np.allclose(kv, F.conv1d(ft, torch.from_numpy(wkv).unsqueeze(-1), torch.from_numpy(bkv)).numpy(), atol=2e-5)
    assert np.allclose(lin, F.relu(F.linear(ft.flatten(1), torch.from_numpy(wl), torch.from_numpy(bl))).numpy(), atol=2e-5)

    c = T.cma_step_case(9, rows=3, L=12, P=16, H=8, Hq=4, Ct=6, d_out=4, m_out=6, E=2)
    x2 = np.zeros((c["rows"], c["x2w"]), np.float32)
    x2[:, -c["E"]:] = c["prev"]
    h_out, feats = np.zeros((c["rows"], 2, c["H"]), np.float32), np.zeros((c["rows"], c["H"]), np.float32)
    d = T.cma_step_desc(T.hp, c, x2, h_out, feats, None)
    T.check(L, L.ivln_cma_step_fwd(C.byref(d), 0, None), "twin cma_step")
    t = {k: torch.from_numpy(v) for k, v in c.items() if isinstance(v, np.ndarray)}
    H, Hq = c["H"], c["Hq"]

    def gru(x, h, wi, wh, bi, bh):
        cell = torch.nn.GRUCell(x.shape[1], H)
        with torch.no_grad():
            cell.weight_ih.copy_(wi), cell.weight_hh.copy_(wh), cell.bias_ih.copy_(bi), cell.bias_hh.copy_(bh)
            return cell(x, h)

    mk = t["mask"].float().unsqueeze(1)
    state = gru(t["state_in"], t["h_in"][:, 0] * mk, t["w_ih1"], t["w_hh1"], t["b_ih1"], t["b_hh1"])
    logits = torch.einsum("rh,rhl->rl", state, t["Mq"][:, :H]) + t["Mq"][:, H]
    pad = (torch.arange(c["L"]).unsqueeze(0) >= t["lengths"].unsqueeze(1)).float()
    a = torch.softmax((logits - pad * 1e8) / 16.0, 1)
    text = torch.einsum("rl,rcl->rc", a, t["txt"])
    q2 = torch.einsum("rl,rcl->rc", a, t["TQb"])
    outs = []
    for kv_t in (t["dkv"], t["mkv"]):
        aa = torch.softmax(torch.einsum("rc,rcp->rp", q2, kv_t[:, :Hq]) / 16.0, 1)
        outs.append(torch.einsum("rp,rcp->rc", aa, kv_t[:, Hq:]))
    x2_ref = torch.cat([state, text, outs[0], outs[1], t["prev"]], 1)
    c2 = F.relu(F.linear(x2_ref, t["w_c"], t["b_c"]))
    f_ref = gru(c2, t["h_in"][:, 1] * mk, t["w_ih2"], t["w_hh2"], t["b_ih2"], t["b_hh2"])
    assert np.allclose(x2, x2_ref.detach().numpy(), atol=2e-5)
    assert np.allclose(feats, f_ref.detach().numpy(), atol=2e-5)
    assert np.allclose(h_out[:, 0], state.detach().numpy(), atol=2e-5) and np.allclose(h_out[:, 1], feats, atol=0)
