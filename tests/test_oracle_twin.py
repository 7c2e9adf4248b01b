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
                 "ivln_mapper_known_begin", "ivln_mapper_load_known", "ivln_mapper_known_raster", "ivln_mapper_status",
                 "ivln_mapper_world_export"]:
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
