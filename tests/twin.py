"""Test helper: call the SAME `ivln_*` C-ABI entry point in libivln_hip.so (device pointers) and in its CPU twin
oracle/libivln_ref.so (host pointers) - SURVEY.md section 8b, "a CPU twin exports the same symbols for parity tests".
Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_TWIN = None

vp, i32, i64, f32, f64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double


def desc_type():
    from ivln_ce_amd.ops import GemmDesc

    return GemmDesc


def _sigs(L):
    D = desc_type()
    L.ivln_gemm_f32.argtypes = [C.POINTER(D), vp]
    L.ivln_groupnorm_f32.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, i64, i64, i32, i64, i64, i64, vp, vp, vp]
    L.ivln_mapper_create.argtypes = [i32, i32, i32, f64, f64, f64, f64, i64, i64, C.POINTER(vp)]
    L.ivln_mapper_destroy.argtypes = [vp]
    L.ivln_mapper_frames.argtypes = [vp, vp, i32, vp, vp, vp]
    L.ivln_mapper_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp]
    L.ivln_mapper_step_posed.argtypes = [vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp]
    L.ivln_mapper_step_begin.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp, vp, vp]
    L.ivln_mapper_step_finish.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp]
    L.ivln_mapper_known_begin.argtypes = [vp, vp, i32, vp]
    L.ivln_mapper_load_known.argtypes = [vp, i32, vp, vp, i64, vp]
    L.ivln_mapper_known_raster.argtypes = [vp, vp, vp, i32, vp, vp, vp]
    L.ivln_mapper_status.argtypes = [vp, C.POINTER(i64), vp]
    L.ivln_strerror.restype = C.c_char_p
    L.ivln_strerror.argtypes = [i32]
    return L


def twin():
    """oracle/libivln_ref.so (built on demand by oracle/Makefile)."""
    global _TWIN
    if _TWIN is None:
        so = os.path.join(ROOT, "oracle", "libivln_ref.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libivln_ref.so"])
        _TWIN = _sigs(C.CDLL(so))
    return _TWIN


def hp(a):
    """host pointer of a contiguous numpy array (or None)"""
    return None if a is None else a.ctypes.data_as(vp)


def conv_desc(ptr, x, w, out, stride=1, pad=0, scale=None, shift=None, residual=None, relu=False, grouped=False):
    """ivln_gemm_desc of a forward NCHW convolution exactly as ivln_ce_amd.ops.conv2d fills it (non-deferred, split
    heuristic off: splits = 1); `ptr(array)` turns an operand into the pointer the library expects."""
    from ivln_ce_amd import ops

    N, Cin, H, W = x.shape
    Cout, _, KH, KW = w.shape[-4:]
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    d = desc_type()()
    d.A, d.B, d.D = ptr(w), ptr(x), ptr(out)
    d.M, d.N, d.K = Cout, N * Ho * Wo, Cin * KH * KW
    d.amode, d.dmode = ops.A_MK, ops.D_NCHW
    d.lda = d.K
    d.Cin, d.Hin, d.Win, d.Hout, d.Wout = Cin, H, W, Ho, Wo
    d.stride, d.pad, d.dil = stride, pad, 1
    d.HoWo = Ho * Wo
    d.bmode = ops.B_CONV1X1 if (KH == 1 and pad == 0) else (ops.B_CONV_K3 if KH == 3 else ops.B_CONV_K7)
    d.scale, d.shift, d.residual = ptr(scale), ptr(shift), ptr(residual)
    d.relu = int(relu)
    d.splits = 1
    if grouped:
        d.grp_imgs, d.a_grp_stride = N // w.shape[0], Cout * Cin * KH * KW
    return d


def convt_case(seed, N, Cin, H, W, Cout, k):
    """nn.ConvTranspose2d(Cin, Cout, k, stride=2) with output exactly (2H, 2W) (k=3: padding 1, output_padding 1; k=2:
    padding 0) as ops.conv_transpose2d_s2 launches it: the four output-parity classes stacked into one
    (4*Cout, Cin, t, t) weight, rows 4*co + cls.  Returns numpy operands and the torch result."""
    import torch
    import torch.nn.functional as F

    from ivln_ce_amd import ops

    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cin, Cout, k, k, generator=g) / (Cin * k * k / 4) ** 0.5
    sc, sh = 1 + 0.2 * torch.randn(Cout, generator=g), torch.randn(Cout, generator=g)
    pad, op = (1, 1) if k == 3 else (0, 0)
    ref = F.conv_transpose2d(x, w, None, stride=2, padding=pad, output_padding=op)
    res = torch.randn(ref.shape, generator=g)
    ref = F.relu(ref * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + res)
    stacked = ops.convt_s2_stack(ops.convt_s2_classes(w, pad))
    return {"x": np32(x), "w": np32(stacked), "scale": np32(sc), "shift": np32(sh), "residual": np32(res),
            "ref": ref.numpy(), "Cout": Cout}


def convt_desc(ptr, c, x, w, out, scale, shift, residual):
    """ivln_gemm_desc of the stacked transposed conv (IVLN_B_CONV_K2 / IVLN_B_CONV1X1 into IVLN_D_NCHW_UP2X4)."""
    from ivln_ce_amd import ops

    N, Cin, H, W = c["x"].shape
    t = c["w"].shape[2]
    d = desc_type()()
    d.A, d.B, d.D = ptr(w), ptr(x), ptr(out)
    d.M, d.N, d.K = 4 * c["Cout"], N * H * W, Cin * t * t
    d.amode, d.dmode = ops.A_MK, ops.D_NCHW_UP2X4
    d.bmode = ops.B_CONV_K2 if t == 2 else ops.B_CONV1X1
    d.lda = d.K
    d.Cin, d.Hin, d.Win, d.Hout, d.Wout = Cin, H, W, H, W
    d.stride, d.pad, d.dil = 1, 0, 1
    d.HoWo, d.Ctot = H * W, c["Cout"]
    d.scale, d.shift, d.residual = ptr(scale), ptr(shift), ptr(residual)
    d.relu, d.splits = 1, 1
    return d


def check(L, code, what):
    assert code == 0, f"{what}: {L.ivln_strerror(code).decode()} ({code})"


def np32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---- round-2 entry points: shared case builders (numpy operands + descriptors for either library) ---------------------
def _sigs2(L):
    from ivln_ce_amd.ops import GnConvDesc, NconvDesc

    L.ivln_gn_conv_f32.argtypes = [C.POINTER(GnConvDesc), vp]
    L.ivln_nconv_f32.argtypes = [C.POINTER(NconvDesc), vp]
    L.ivln_kv_linear_f32.argtypes = [vp, i32, i32, i32, vp, vp, i32, vp, vp, vp, i32, i32, vp, i64, vp]
    L.ivln_cma_step_fwd.argtypes = [C.POINTER(cma_desc_type()), i32, vp]
    L.ivln_cma_step_ws_floats.restype = i64
    L.ivln_cma_step_ws_floats.argtypes = [i32, i32, i32, i32]
    return L


def cma_desc_type():
    from ivln_ce_amd import ops

    return ops.CmaStepDesc


def gn_conv_case(seed, N, Cc, H, W, G, splits, second, residual, pool, ka, sa, Cout_a, Cout_b, sb):
    """Operands of one ivln_gn_conv_f32 launch: slabs of the raw [C][N*H*W] matrix (+ a second operand), affine
    parameters, conv A (k = ka, stride sa, pad ka // 2) and optionally conv B (1x1, stride sb)."""
    rs = np.random.RandomState(seed)
    M = N * H * W
    c = {"N": N, "C": Cc, "H": H, "W": W, "G": G, "splits": splits, "pool": pool, "ka": ka, "sa": sa, "sb": sb,
         "Cout_a": Cout_a, "Cout_b": Cout_b}
    c["x"] = (rs.randn(splits, Cc, M) * 0.7 + 0.1).astype(np.float32)
    c["gamma"], c["beta"] = (1 + 0.2 * rs.randn(Cc)).astype(np.float32), (0.3 * rs.randn(Cc)).astype(np.float32)
    if second:
        c["x2"] = (rs.randn(2, Cc, M) * 0.5).astype(np.float32)
        c["gamma2"], c["beta2"] = (1 + 0.2 * rs.randn(Cc)).astype(np.float32), (0.3 * rs.randn(Cc)).astype(np.float32)
    if residual:
        c["residual"] = rs.randn(N, Cc, H, W).astype(np.float32)
    c["wa"] = (rs.randn(Cout_a, Cc, ka, ka) / np.sqrt(Cc * ka * ka)).astype(np.float32)
    if Cout_b:
        c["wb"] = (rs.randn(Cout_b, Cc, 1, 1) / np.sqrt(Cc)).astype(np.float32)
    Hp, Wp = ((H - 1) // 2 + 1, (W - 1) // 2 + 1) if pool else (H, W)
    pad = ka // 2
    c["Hp"], c["Wp"] = Hp, Wp
    c["Ho"], c["Wo"] = (Hp + 2 * pad - ka) // sa + 1, (Wp + 2 * pad - ka) // sa + 1
    c["Hb"], c["Wb"] = ((Hp - 1) // sb + 1, (Wp - 1) // sb + 1) if Cout_b else (0, 0)
    return c


def gn_conv_desc(ptr, c, act, ya, yb):
    from ivln_ce_amd.ops import GnConvDesc

    d = GnConvDesc()
    M = c["N"] * c["H"] * c["W"]
    d.x, d.splits, d.slab_stride, d.gamma, d.beta = ptr(c["x"]), c["splits"], c["C"] * M, ptr(c["gamma"]), ptr(c["beta"])
    if "x2" in c:
        d.x2, d.splits2, d.slab_stride2, d.gamma2, d.beta2 = ptr(c["x2"]), 2, c["C"] * M, ptr(c["gamma2"]), ptr(c["beta2"])
    d.residual = ptr(c.get("residual"))
    d.N, d.C, d.H, d.W, d.groups, d.eps, d.relu, d.pool = c["N"], c["C"], c["H"], c["W"], c["G"], 1e-5, 1, int(c["pool"])
    d.act_out = ptr(act)
    d.wa, d.Cout_a, d.ka, d.stride_a, d.pad_a, d.ya = ptr(c["wa"]), c["Cout_a"], c["ka"], c["sa"], c["ka"] // 2, ptr(ya)
    if c["Cout_b"]:
        d.wb, d.Cout_b, d.stride_b, d.yb = ptr(c["wb"]), c["Cout_b"], c["sb"], ptr(yb)
    return d


def part_stats(y, groups, rows_per_part):
    """(count, mean, M2) of every (row strip, image, group) of a raw [C][N][H][W] tensor: the partials a producing
    ivln_nconv_f32 launch leaves for its consumer."""
    Cc, N, H, W = y.shape
    parts = (H + rows_per_part - 1) // rows_per_part
    cpg = Cc // groups
    st = np.zeros((parts, N, groups, 3), np.float32)
    for p in range(parts):
        for n in range(N):
            for g in range(groups):
                v = y[g * cpg:(g + 1) * cpg, n, p * rows_per_part:(p + 1) * rows_per_part].astype(np.float64)
                st[p, n, g] = (v.size, v.mean(), ((v - v.mean()) ** 2).sum())
    return st, parts


def nconv_case(seed, N, Cc, H, W, G, second, residual, ka, sa, Cout_a, ga, Cout_b, gb, sb, rows):
    rs = np.random.RandomState(seed)
    c = {"N": N, "C": Cc, "H": H, "W": W, "G": G, "ka": ka, "sa": sa, "sb": sb, "Cout_a": Cout_a, "ga": ga,
         "Cout_b": Cout_b, "gb": gb, "rows": rows}
    c["x"] = (rs.randn(Cc, N, H, W) * 0.8 + 0.2).astype(np.float32)
    c["stats"], c["parts"] = part_stats(c["x"], G, 4)
    c["gamma"], c["beta"] = (1 + 0.2 * rs.randn(Cc)).astype(np.float32), (0.3 * rs.randn(Cc)).astype(np.float32)
    if second:
        c["x2"] = (rs.randn(Cc, N, H, W) * 0.6).astype(np.float32)
        c["stats2"], c["parts2"] = part_stats(c["x2"], G, 8)
        c["gamma2"], c["beta2"] = (1 + 0.2 * rs.randn(Cc)).astype(np.float32), (0.3 * rs.randn(Cc)).astype(np.float32)
    if residual:
        c["residual"] = rs.randn(N, Cc, H, W).astype(np.float32)
    c["wa"] = (rs.randn(Cout_a, Cc, ka, ka) / np.sqrt(Cc * ka * ka)).astype(np.float32)
    if Cout_b:
        c["wb"] = (rs.randn(Cout_b, Cc, 1, 1) / np.sqrt(Cc)).astype(np.float32)
    ph = ka // 2
    c["Ho"], c["Wo"] = (H + 2 * ph - ka) // sa + 1, (W + 2 * ph - ka) // sa + 1
    c["Hb"], c["Wb"] = ((H - 1) // sb + 1, (W - 1) // sb + 1) if Cout_b else (0, 0)
    RS = min(rows, c["Ho"])
    if Cout_b and RS % sb:
        RS = (RS + sb - 1) // sb * sb
    c["strips"] = (c["Ho"] + RS - 1) // RS
    return c


def nconv_desc(ptr, c, act, ya, sta, yb, stb):
    from ivln_ce_amd.ops import NconvDesc

    d = NconvDesc()
    d.x, d.stats, d.parts, d.gamma, d.beta = ptr(c["x"]), ptr(c["stats"]), c["parts"], ptr(c["gamma"]), ptr(c["beta"])
    if "x2" in c:
        d.x2, d.stats2, d.parts2, d.gamma2, d.beta2 = ptr(c["x2"]), ptr(c["stats2"]), c["parts2"], ptr(c["gamma2"]), ptr(c["beta2"])
    d.residual = ptr(c.get("residual"))
    d.N, d.C, d.H, d.W, d.groups, d.eps, d.relu = c["N"], c["C"], c["H"], c["W"], c["G"], 1e-5, 1
    d.act_out = ptr(act)
    d.wa, d.Cout_a, d.ka, d.groups_a, d.ya, d.stats_a = ptr(c["wa"]), c["Cout_a"], c["ka"], c["ga"], ptr(ya), ptr(sta)
    if c["Cout_b"]:
        d.wb, d.Cout_b, d.groups_b, d.yb, d.stats_b = ptr(c["wb"]), c["Cout_b"], c["gb"], ptr(yb), ptr(stb)
    d.rows_per_block, d.stride_a, d.stride_b = c["rows"], c["sa"], c["sb"]
    return d


def merged(stats):
    """(mean, variance) per (image, group) of a [parts][N][groups][3] partial-statistics tensor."""
    cnt = stats[..., 0].astype(np.float64)
    tot = cnt.sum(0)
    mean = (cnt * stats[..., 1]).sum(0) / tot
    m2 = (stats[..., 2] + cnt * (stats[..., 1] - mean) ** 2).sum(0)
    return mean, m2 / tot


def cma_step_case(seed, rows, L, P, H=512, Hq=256, Ct=256, d_out=128, m_out=256, E=32):
    rs = np.random.RandomState(seed)
    f = lambda *s, k=1.0: (rs.randn(*s) * k).astype(np.float32)  # noqa: E731
    I1, x2w = d_out + m_out + E, H + Ct + d_out + m_out + E
    c = {"rows": rows, "L": L, "P": P, "H": H, "Hq": Hq, "Ct": Ct, "d_out": d_out, "m_out": m_out, "E": E, "x2w": x2w}
    c["state_in"], c["h_in"] = np.abs(f(rows, I1)), f(rows, 2, H, k=0.5)
    c["mask"] = (rs.rand(rows) > 0.3).astype(np.uint8)
    for nm, sh, k in (("w_ih1", (3 * H, I1), I1 ** -0.5), ("w_hh1", (3 * H, H), H ** -0.5), ("b_ih1", (3 * H,), 0.1),
                      ("b_hh1", (3 * H,), 0.1), ("w_c", (H, x2w), x2w ** -0.5), ("b_c", (H,), 0.1),
                      ("w_ih2", (3 * H, H), H ** -0.5), ("w_hh2", (3 * H, H), H ** -0.5), ("b_ih2", (3 * H,), 0.1),
                      ("b_hh2", (3 * H,), 0.1)):
        c[nm] = f(*sh, k=k)
    c["Mq"], c["TQb"] = f(rows, H + 1, L, k=H ** -0.5 * 4), f(rows, Hq, L, k=0.5)
    c["txt"] = f(rows, Ct, L)
    c["lengths"] = rs.randint(1, L + 1, size=rows).astype(np.int32)
    c["lengths"][0] = L
    for r in range(rows):
        c["txt"][r, :, c["lengths"][r]:] = 0
    c["dkv"], c["mkv"] = f(rows, Hq + d_out, P, k=0.5), f(rows, Hq + m_out, P, k=0.5)
    c["prev"] = f(rows, E)
    return c


def cma_step_desc(ptr, c, x2, h_out, feats, ws):
    d = cma_desc_type()()
    for k in ("rows", "L", "P", "H", "Hq", "Ct", "d_out", "m_out", "E", "x2w"):
        setattr(d, k, c[k])
    d.state_in, d.h_in, d.ld_h, d.mask = ptr(c["state_in"]), ptr(c["h_in"]), 2 * c["H"], ptr(c["mask"])
    for k in ("w_ih1", "w_hh1", "b_ih1", "b_hh1", "w_c", "b_c", "w_ih2", "w_hh2", "b_ih2", "b_hh2", "txt", "lengths", "dkv", "mkv"):
        setattr(d, k, ptr(c[k]))
    d.Mq, d.Mq_img, d.TQb, d.TQb_img = ptr(c["Mq"]), (c["H"] + 1) * c["L"], ptr(c["TQb"]), c["Hq"] * c["L"]
    d.scale = 1.0 / 16.0
    d.x2, d.h_out, d.ld_ho, d.feats, d.ws = ptr(x2), ptr(h_out), 2 * c["H"], ptr(feats), ptr(ws)
    return d
