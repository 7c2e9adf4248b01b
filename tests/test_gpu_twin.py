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


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "mapper_b*.npz"))), ids=lambda p: os.path.basename(p)[7:-4])
def test_ivln_mapper_entry_points_device_vs_twin(path):
    """create / frames / step / status through the raw C ABI of both libraries: maps bit-identical every step, frames
    bit-identical, world-cloud size equal."""
    from ivln_ce_amd._lib import lib, stream_ptr

    g = np.load(path)
    B, H, W = int(g["B"]), int(g["H"]), int(g["W"])
    Lt, Ld = T.twin(), T._sigs(lib())
    ht, hd = C.c_void_p(), C.c_void_p()
    vf = float(np.deg2rad(90.0 * H / W))
    T.check(Lt, Lt.ivln_mapper_create(B, H, W, vf, 6.4, 6.4, 0.1, 0, 0, C.byref(ht)), "twin create")
    T.check(Ld, Ld.ivln_mapper_create(B, H, W, vf, 6.4, 6.4, 0.1, 1 << 20, 1 << 22, C.byref(hd)), "device create")
    s = stream_ptr()
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
        T.check(Ld, Ld.ivln_mapper_frames(_dp(pd), _dp(od), B, _dp(Td), _dp(rotd), s), "device frames")
        T.check(Ld, Ld.ivln_mapper_step(hd, _dp(dd), _dp(ld), _dp(Td), _dp(pd), _dp(rotd), _dp(ndd), B, _dp(occd), _dp(semd), s),
                "device step")
        nt, ndv = C.c_int64(0), C.c_int64(0)
        T.check(Lt, Lt.ivln_mapper_status(ht, C.byref(nt), None), "twin status")
        T.check(Ld, Ld.ivln_mapper_status(hd, C.byref(ndv), s), "device status")
        assert np.array_equal(Td.cpu().numpy(), Th) and np.array_equal(rotd.cpu().numpy(), roth), f"frames step {t}"
        assert np.array_equal(occd.cpu().numpy(), occh) and np.array_equal(semd.cpu().numpy(), semh), f"maps step {t}"
        assert nt.value == ndv.value == int(g[f"world_n_{t}"])
    Lt.ivln_mapper_destroy(ht)
    Ld.ivln_mapper_destroy(hd)
