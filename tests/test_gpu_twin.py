"""The same `ivln_*` C-ABI entry point run in libivln_hip.so (device pointers) and in its CPU twin
oracle/libivln_ref.so (host pointers) on the same bytes, results diffed symbol for symbol.
The twin itself is pinned to torch / the reference goldens in tests/test_oracle_twin.py."""
import ctypes as C
import glob
import os

import numpy as np
import pytest
import torch

import twin as T

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
G = os.path.join(os.path.dirname(__file__), "golden")


def _dev(a):
    return None if a is None else torch.from_numpy(a).to(DEV)


def _dp(t):
    return None if t is None else t.data_ptr()


@pytest.mark.parametrize("N,Cin,H,W,Cout,k,s,p,grouped", [
    (4, 64, 16, 16, 64, 3, 1, 1, False),    # direct LDS-patch kernel
    (2, 256, 16, 16, 64, 1, 1, 0, False),   # vector-load 1x1 GEMM
    (2, 128, 16, 16, 256, 1, 2, 0, False),  # strided 1x1: gather GEMM
    (2, 1, 64, 64, 32, 7, 2, 3, False),     # stem
    (2, 64, 16, 16, 64, 3, 2, 1, False),
    (8, 128, 32, 32, 128, 3, 2, 1, False),  # stride-2 3x3 on the direct kernel (even / odd column planes in LDS)
    (4, 64, 64, 64, 64, 3, 2, 1, False),
    (8, 64, 30, 30, 64, 3, 2, 1, False),    # ... with a ragged 15x15 output
    (4, 64, 8, 8, 64, 3, 1, 1, True),       # image-grouped weight sets
    (4, 128, 8, 8, 64, 1, 1, 0, True),
])
def test_ivln_gemm_f32_device_vs_twin(N, Cin, H, W, Cout, k, s, p, grouped):
    from ivln_ce_amd._lib import lib, stream_ptr

    rs = np.random.RandomState(N * 7 + Cin + k)
    x = rs.randn(N, Cin, H, W).astype(np.float32)
    w = (rs.randn(*((2,) if grouped else ()), Cout, Cin, k, k) / np.sqrt(Cin * k * k)).astype(np.float32)
    ng = 2 if grouped else 1
    sc, sh = (1 + 0.2 * rs.randn(ng * Cout)).astype(np.float32), rs.randn(ng * Cout).astype(np.float32)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    res = rs.randn(N, Cout, Ho, Wo).astype(np.float32)
    out_h = np.zeros((N, Cout, Ho, Wo), np.float32)
    dh = T.conv_desc(T.hp, x, w, out_h, stride=s, pad=p, scale=sc, shift=sh, residual=res, relu=True, grouped=grouped)
    Lt = T.twin()
    T.check(Lt, Lt.ivln_gemm_f32(C.byref(dh), None), "twin")
    xd, wd, scd, shd, resd = _dev(x), _dev(w), _dev(sc), _dev(sh), _dev(res)
    out_d = torch.zeros((N, Cout, Ho, Wo), device=DEV)
    dd = T.conv_desc(_dp, xd, wd, out_d, stride=s, pad=p, scale=scd, shift=shd, residual=resd, relu=True, grouped=grouped)
    Ld = T._sigs(lib())
    T.check(Ld, Ld.ivln_gemm_f32(C.byref(dd), stream_ptr()), "device")
    err = float(np.abs(out_d.cpu().numpy() - out_h).max())
    assert err < 3e-5, err  # same fmaf arithmetic, different summation order (MFMA tiles vs a straight k loop)


@pytest.mark.parametrize("N,Cin,H,W,Cout,k,narrow", [
    (2, 64, 32, 32, 32, 3, False),   # 2x2-window direct kernel, 16-byte stores of whole 2x2 output blocks
    (2, 64, 32, 32, 32, 3, True),    # ... the same launch with the MFMA layout's 4-byte stores
    (4, 128, 8, 8, 64, 3, False),    # 8x8 class grid: two images per pixel tile, channel chunks split over blockIdx.z
    (2, 32, 16, 16, 16, 3, False),
    (1, 24, 6, 10, 8, 3, False),     # Cin % 16 != 0: the implicit GEMM's constant-division gather
    (2, 16, 5, 7, 8, 3, False),      # odd class-grid width: scalar stores
    (2, 64, 32, 32, 16, 2, False),   # k=2: a 1x1 GEMM per class, float4-staged kernel + wide stores
    (2, 64, 32, 32, 16, 2, True),
    (1, 20, 6, 6, 5, 2, False),
])
def test_stacked_transposed_conv_device_vs_twin(N, Cin, H, W, Cout, k, narrow):
    from ivln_ce_amd._lib import lib, stream_ptr

    c = T.convt_case(N * 5 + Cin + k, N, Cin, H, W, Cout, k)
    out_h = np.zeros(c["ref"].shape, np.float32)
    Lt = T.twin()
    T.check(Lt, Lt.ivln_gemm_f32(C.byref(T.convt_desc(T.hp, c, c["x"], c["w"], out_h, c["scale"], c["shift"],
                                                     c["residual"])), None), "twin")
    assert np.allclose(out_h, c["ref"], atol=3e-5, rtol=1e-5)
    xd, wd, scd, shd, resd = (_dev(c[k_]) for k_ in ("x", "w", "scale", "shift", "residual"))
    out_d = torch.zeros(c["ref"].shape, device=DEV)
    dd = T.convt_desc(_dp, c, xd, wd, out_d, scd, shd, resd)
    dd.no_wide_epilogue = int(narrow)
    Ld = T._sigs(lib())
    T.check(Ld, Ld.ivln_gemm_f32(C.byref(dd), stream_ptr()), "device")
    err = float(np.abs(out_d.cpu().numpy() - out_h).max())
    assert err < 3e-5, err


def test_ivln_groupnorm_f32_device_vs_twin():
    from ivln_ce_amd._lib import lib, stream_ptr

    rs = np.random.RandomState(2)
    N, Cc, HW, Gr = 3, 128, 64, 16
    x, ga, be = (rs.randn(N, Cc, HW) * 2 + 0.3).astype(np.float32), rs.randn(Cc).astype(np.float32), rs.randn(Cc).astype(np.float32)
    res = rs.randn(N, Cc, HW).astype(np.float32)
    args = (N, Cc, HW, Gr, 1e-5, 1, 0, 0, 1, 0, 0, 0, None, None)
    yh = np.zeros_like(x)
    Lt = T.twin()
    T.check(Lt, Lt.ivln_groupnorm_f32(T.hp(x), T.hp(ga), T.hp(be), T.hp(res), T.hp(yh), *args, None), "twin")
    xd, gd, bd, rd, yd = _dev(x), _dev(ga), _dev(be), _dev(res), torch.zeros((N, Cc, HW), device=DEV)
    Ld = T._sigs(lib())
    T.check(Ld, Ld.ivln_groupnorm_f32(_dp(xd), _dp(gd), _dp(bd), _dp(rd), _dp(yd), *args, stream_ptr()), "device")
    assert float(np.abs(yd.cpu().numpy() - yh).max()) < 2e-5


@pytest.mark.parametrize("posed", [False, True, "halves"], ids=["frames+step", "step_posed", "begin+finish"])
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "mapper_b*.npz"))), ids=lambda p: os.path.basename(p)[7:-4])
def test_ivln_mapper_entry_points_device_vs_twin(path, posed):
    """create / frames / step (or step_posed: the same from the sensor pose, six launches) / status through the raw C
    ABI of both libraries: maps bit-identical every step, frames bit-identical, world-cloud size equal."""
    from ivln_ce_amd._lib import lib, stream_ptr

    g = np.load(path)
    B, H, W = int(g["B"]), int(g["H"]), int(g["W"])
    Lt, Ld = T.twin(), T._sigs(lib())
    ht, hd = C.c_void_p(), C.c_void_p()
    vf = float(np.deg2rad(90.0 * H / W))
    T.check(Lt, Lt.ivln_mapper_create(B, H, W, vf, 6.4, 6.4, 0.1, 0, 0, C.byref(ht)), "twin create")
    T.check(Ld, Ld.ivln_mapper_create(B, H, W, vf, 6.4, 6.4, 0.1, 1 << 20, 1 << 22, C.byref(hd)), "device create")
    s = stream_ptr()
    ht2 = C.c_void_p()  # ("halves": a second twin handle stepped through begin + finish beside the single-call one)
    if posed == "halves":
        T.check(Lt, Lt.ivln_mapper_create(B, H, W, vf, 6.4, 6.4, 0.1, 0, 0, C.byref(ht2)), "twin create 2")
    for t in range(int(g["steps"])):
        pose, orient = T.np32(g[f"pose_{t}"]), np.ascontiguousarray(g[f"orientation_{t}"], np.float64)
        depth = T.np32(g[f"depth_{t}"]).reshape(B, H, W)
        labels = np.ascontiguousarray(g[f"semantic12_{t}"], np.uint8).reshape(B, H, W)
        nd = np.ascontiguousarray(g[f"not_done_{t}"], np.uint8).reshape(-1)
        Th, roth = np.zeros((B, 4, 4), np.float32), np.zeros((B, 3, 3), np.float32)
        occh, semh = np.zeros((B, 64, 64), np.uint8), np.zeros((B, 64, 64), np.uint8)
        T.check(Lt, Lt.ivln_mapper_frames(T.hp(pose), T.hp(orient), B, T.hp(Th), T.hp(roth), None), "twin frames")
        T.check(Lt, Lt.ivln_mapper_step(ht, T.hp(depth), T.hp(labels), T.hp(Th), T.hp(pose), T.hp(roth), T.hp(nd), B,
                                        T.hp(occh), T.hp(semh), None), "twin step")
        pd, od, dd, ld, ndd = _dev(pose), _dev(orient), _dev(depth), _dev(labels), _dev(nd)
        Td, rotd = torch.zeros((B, 4, 4), device=DEV), torch.zeros((B, 3, 3), device=DEV)
        occd = torch.zeros((B, 64, 64), dtype=torch.uint8, device=DEV)
        semd = torch.zeros((B, 64, 64), dtype=torch.uint8, device=DEV)
        if posed == "halves":
            # the step in two calls (ivln_mapper_step_begin on a SIDE stream - what a pred-semantics step does beside RedNet -,
            # _finish on the caller's behind an event), on both libraries; a finish for a batch beyond the handle's size is refused
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                T.check(Ld, Ld.ivln_mapper_step_begin(hd, _dp(dd), _dp(pd), _dp(od), _dp(ndd), B, _dp(occd), _dp(Td), _dp(rotd),
                                                      stream_ptr()), "device step_begin")
            torch.cuda.current_stream().wait_stream(side)
            assert Ld.ivln_mapper_step_finish(hd, _dp(dd), _dp(ld), _dp(Td), _dp(pd), _dp(rotd), _dp(ndd), 65, _dp(occd), _dp(semd), s) != 0  # (B > B_max)
            T.check(Ld, Ld.ivln_mapper_step_finish(hd, _dp(dd), _dp(ld), _dp(Td), _dp(pd), _dp(rotd), _dp(ndd), B, _dp(occd), _dp(semd), s),
                    "device step_finish")
            # ... and the twin's pair against its own single call (fresh handle state is the same: it was stepped above)
            Th2, roth2 = np.zeros_like(Th), np.zeros_like(roth)
            occ2, sem2 = np.zeros_like(occh), np.zeros_like(semh)
            T.check(Lt, Lt.ivln_mapper_step_begin(ht2, T.hp(depth), T.hp(pose), T.hp(orient), T.hp(nd), B, T.hp(occ2), T.hp(Th2),
                                                  T.hp(roth2), None), "twin step_begin")
            T.check(Lt, Lt.ivln_mapper_step_finish(ht2, T.hp(depth), T.hp(labels), T.hp(Th2), T.hp(pose), T.hp(roth2), T.hp(nd), B,
                                                   T.hp(occ2), T.hp(sem2), None), "twin step_finish")
            assert np.array_equal(occ2, occh) and np.array_equal(sem2, semh) and np.array_equal(Th2, Th)
        elif posed:
            T.check(Ld, Ld.ivln_mapper_step_posed(hd, _dp(dd), _dp(ld), _dp(pd), _dp(od), _dp(ndd), B, _dp(occd), _dp(semd),
                                                  _dp(Td), _dp(rotd), s), "device step_posed")
        else:
            T.check(Ld, Ld.ivln_mapper_frames(_dp(pd), _dp(od), B, _dp(Td), _dp(rotd), s), "device frames")
            T.check(Ld, Ld.ivln_mapper_step(hd, _dp(dd), _dp(ld), _dp(Td), _dp(pd), _dp(rotd), _dp(ndd), B, _dp(occd),
                                            _dp(semd), s), "device step")
        nt, ndv = C.c_int64(0), C.c_int64(0)
        T.check(Lt, Lt.ivln_mapper_status(ht, C.byref(nt), None), "twin status")
        T.check(Ld, Ld.ivln_mapper_status(hd, C.byref(ndv), s), "device status")
        assert np.array_equal(Td.cpu().numpy(), Th) and np.array_equal(rotd.cpu().numpy(), roth), f"frames step {t}"
        assert np.array_equal(occd.cpu().numpy(), occh) and np.array_equal(semd.cpu().numpy(), semh), f"maps step {t}"
        assert nt.value == ndv.value == int(g[f"world_n_{t}"])
    Lt.ivln_mapper_destroy(ht)
    Ld.ivln_mapper_destroy(hd)
    if posed == "halves":
        Lt.ivln_mapper_destroy(ht2)


# ---- round-2 entry points: device library vs CPU twin on the same bytes -----------------------------------------------
def _pair(c):
    """device copies of a case's numpy operands"""
    return {k: (_dev(v) if isinstance(v, np.ndarray) else v) for k, v in c.items()}


def _ptr_dev(t):
    return None if t is None else t.data_ptr()


@pytest.mark.parametrize("N,Cc,H,W,second,residual,pool,ka,sa,Ca,Cb,sb", [
    (4, 256, 16, 16, False, False, False, 3, 1, 64, 0, 1),     # layer-2 conv2 input: GN + 3x3
    (4, 256, 16, 16, False, True, False, 1, 1, 64, 0, 1),      # block tail + identity -> next conv1
    (4, 512, 8, 8, True, False, False, 1, 1, 128, 0, 1),       # tail with the downsample GroupNorm as second operand
    (4, 512, 8, 8, False, True, False, 1, 1, 256, 1024, 2),    # layer change: conv1 (A) + strided downsample (B)
    (2, 32, 64, 64, False, False, True, 1, 1, 32, 128, 1),     # stem: GN + ReLU + MaxPool -> layer-1 convs
])
def test_ivln_gn_conv_f32_device_vs_twin(N, Cc, H, W, second, residual, pool, ka, sa, Ca, Cb, sb):
    from ivln_ce_amd._lib import lib, stream_ptr

    G = 16
    c = T.gn_conv_case(N + Cc + ka, N, Cc, H, W, G, 16 if not pool else 1, second, residual, pool, ka, sa, Ca, Cb, sb)
    shp_act, shp_a = (N, Cc, c["Hp"], c["Wp"]), (G, Ca, N * c["Ho"] * c["Wo"])
    shp_b = (G, max(Cb, 1), N * max(c["Hb"] * c["Wb"], 1))
    act_h, ya_h, yb_h = np.zeros(shp_act, np.float32), np.zeros(shp_a, np.float32), np.zeros(shp_b, np.float32)
    Lt = T._sigs2(T.twin())
    T.check(Lt, Lt.ivln_gn_conv_f32(C.byref(T.gn_conv_desc(T.hp, c, act_h, ya_h, yb_h)), None), "twin")
    cd = _pair(c)
    act_d, ya_d, yb_d = torch.zeros(shp_act, device=DEV), torch.zeros(shp_a, device=DEV), torch.zeros(shp_b, device=DEV)
    Ld = T._sigs2(T._sigs(lib()))
    T.check(Ld, Ld.ivln_gn_conv_f32(C.byref(T.gn_conv_desc(_ptr_dev, cd, act_d, ya_d, yb_d)), stream_ptr()), "device")
    assert float(np.abs(act_d.cpu().numpy() - act_h).max()) < 3e-5
    assert float(np.abs(ya_d.cpu().numpy() - ya_h).max()) < 5e-5      # every partial slab, not just their sum
    if Cb:
        assert float(np.abs(yb_d.cpu().numpy() - yb_h).max()) < 5e-5


@pytest.mark.parametrize("N,Cc,H,W,second,residual,ka,sa,Ca,ga,Cb,gb,sb,rows", [
    (4, 32, 32, 32, False, False, 3, 1, 32, 16, 0, 16, 1, 2),       # layer-1 conv2
    (4, 32, 32, 32, False, False, 1, 1, 128, 16, 0, 16, 1, 2),      # layer-1 conv3
    (4, 128, 32, 32, True, False, 1, 1, 32, 16, 0, 16, 1, 2),       # first block's tail (+ downsample GN) -> conv1
    (4, 128, 32, 32, False, True, 1, 1, 64, 16, 256, 16, 2, 2),     # layer-1 exit: conv1 of layer 2 + its downsample
    (2, 64, 32, 32, False, False, 3, 2, 64, 16, 0, 16, 1, 2),       # the stride-2 3x3 of layer 2
])
def test_ivln_nconv_f32_device_vs_twin(N, Cc, H, W, second, residual, ka, sa, Ca, ga, Cb, gb, sb, rows):
    from ivln_ce_amd._lib import lib, stream_ptr

    c = T.nconv_case(N + Cc + ka + Ca, N, Cc, H, W, 16, second, residual, ka, sa, Ca, ga, Cb, gb, sb, rows)
    sa_, sb_ = (Ca, N, c["Ho"], c["Wo"]), (max(Cb, 1), N, max(c["Hb"], 1), max(c["Wb"], 1))
    act_h = np.zeros((N, Cc, H, W), np.float32) if sa == 1 else None
    ya_h, yb_h = np.zeros(sa_, np.float32), np.zeros(sb_, np.float32)
    sta_h, stb_h = np.zeros((c["strips"], N, ga, 3), np.float32), np.zeros((c["strips"], N, gb, 3), np.float32)
    Lt = T._sigs2(T.twin())
    T.check(Lt, Lt.ivln_nconv_f32(C.byref(T.nconv_desc(T.hp, c, act_h, ya_h, sta_h, yb_h, stb_h)), None), "twin")
    cd = _pair(c)
    act_d = torch.zeros((N, Cc, H, W), device=DEV) if sa == 1 else None
    ya_d, yb_d = torch.zeros(sa_, device=DEV), torch.zeros(sb_, device=DEV)
    sta_d, stb_d = torch.zeros((c["strips"], N, ga, 3), device=DEV), torch.zeros((c["strips"], N, gb, 3), device=DEV)
    Ld = T._sigs2(T._sigs(lib()))
    T.check(Ld, Ld.ivln_nconv_f32(C.byref(T.nconv_desc(_ptr_dev, cd, act_d, ya_d, sta_d, yb_d, stb_d)), stream_ptr()), "device")
    if act_h is not None:
        assert float(np.abs(act_d.cpu().numpy() - act_h).max()) < 5e-5
    assert float(np.abs(ya_d.cpu().numpy() - ya_h).max()) < 1e-4
    sd = sta_d.cpu().numpy()
    assert np.array_equal(sd[..., 0], sta_h[..., 0])                               # counts
    assert np.allclose(sd[..., 1], sta_h[..., 1], atol=2e-5)                       # per-strip means
    assert np.allclose(sd[..., 2], sta_h[..., 2], rtol=2e-4, atol=1e-4)            # per-strip M2
    if Cb:
        assert float(np.abs(yb_d.cpu().numpy() - yb_h).max()) < 1e-4
        assert np.allclose(stb_d.cpu().numpy(), stb_h, rtol=2e-4, atol=1e-4)


def test_ivln_kv_linear_and_cma_step_device_vs_twin():
    from ivln_ce_amd._lib import lib, stream_ptr

    rs = np.random.RandomState(6)
    rows, Cc, P, Ckv, O = 4, 192, 16, 384, 128   # the depth branch of the head
    feat = rs.randn(rows, Cc, P).astype(np.float32)
    wkv, bkv = (rs.randn(Ckv, Cc) / np.sqrt(Cc)).astype(np.float32), rs.randn(Ckv).astype(np.float32)
    wl, bl = (rs.randn(O, Cc * P) / np.sqrt(Cc * P)).astype(np.float32), rs.randn(O).astype(np.float32)
    kv_h, lin_h = np.zeros((rows, Ckv, P), np.float32), np.zeros((rows, O), np.float32)
    Lt, Ld = T._sigs2(T.twin()), T._sigs2(T._sigs(lib()))
    T.check(Lt, Lt.ivln_kv_linear_f32(T.hp(feat), rows, Cc, P, T.hp(wkv), T.hp(bkv), Ckv, T.hp(kv_h), T.hp(wl), T.hp(bl), O, 1,
                                      T.hp(lin_h), O, None), "twin")
    d = [_dev(a) for a in (feat, wkv, bkv, wl, bl)]
    kv_d, lin_d = torch.zeros((rows, Ckv, P), device=DEV), torch.zeros((rows, O), device=DEV)
    T.check(Ld, Ld.ivln_kv_linear_f32(_dp(d[0]), rows, Cc, P, _dp(d[1]), _dp(d[2]), Ckv, _dp(kv_d), _dp(d[3]), _dp(d[4]), O, 1,
                                      _dp(lin_d), O, stream_ptr()), "device")
    assert float(np.abs(kv_d.cpu().numpy() - kv_h).max()) < 2e-5 and float(np.abs(lin_d.cpu().numpy() - lin_h).max()) < 2e-5

    for rows, L in ((4, 80), (8, 200), (1, 24)):
        c = T.cma_step_case(rows + L, rows=rows, L=L, P=16)
        x2_h = np.zeros((rows, c["x2w"]), np.float32)
        x2_h[:, -c["E"]:] = c["prev"]
        ho_h, f_h = np.zeros((rows, 2, c["H"]), np.float32), np.zeros((rows, c["H"]), np.float32)
        T.check(Lt, Lt.ivln_cma_step_fwd(C.byref(T.cma_step_desc(T.hp, c, x2_h, ho_h, f_h, None)), 0, None), "twin")
        cd = _pair(c)
        x2_d = torch.zeros((rows, c["x2w"]), device=DEV)
        x2_d[:, -c["E"]:] = cd["prev"]
        ho_d, f_d = torch.zeros((rows, 2, c["H"]), device=DEV), torch.zeros((rows, c["H"]), device=DEV)
        ws = torch.zeros(int(Ld.ivln_cma_step_ws_floats(rows, L, 16, c["H"])) + 64, device=DEV)
        T.check(Ld, Ld.ivln_cma_step_fwd(C.byref(T.cma_step_desc(_ptr_dev, cd, x2_d, ho_d, f_d, ws)), 0, stream_ptr()), "device")
        assert float(np.abs(x2_d.cpu().numpy() - x2_h).max()) < 2e-5, (rows, L)
        assert float(np.abs(f_d.cpu().numpy() - f_h).max()) < 2e-5 and float(np.abs(ho_d.cpu().numpy() - ho_h).max()) < 2e-5
