"""Thin Python bindings of the HIP kernels (ctypes over the C ABI of include/ivln_hip.h).

Every function takes/returns torch GPU tensors, launches on torch's current stream and performs
no arithmetic itself; tensor allocation and views are the only torch operations used.
"""
import ctypes as C
import contextlib
import os
from typing import Optional

import torch

from . import _lib
from ._lib import check, f32, i32, i64, lib, stream_ptr, vp
from ._lib import dptr as _dptr


class GemmDesc(C.Structure):
    """Mirror of `ivln_gemm_desc` (include/ivln_hip.h) - field order must match."""

    _fields_ = [
        ("A", vp), ("B", vp), ("D", vp),
        ("M", i32), ("N", i32), ("K", i32),
        ("amode", i32), ("bmode", i32), ("dmode", i32),
        ("lda", i64), ("ldb", i64),
        ("Cin", i32), ("Hin", i32), ("Win", i32), ("Hout", i32), ("Wout", i32),
        ("stride", i32), ("pad", i32), ("dil", i32),
        ("koff", vp), ("kpos", vp),
        ("HoWo", i32), ("Ctot", i32),
        ("in_img_stride", i64),
        ("sDm", i64), ("sDn", i64),
        ("scale", vp), ("shift", vp), ("residual", vp),
        ("relu", i32), ("accumulate", i32),
        ("splits", i32),
        ("ws", vp), ("ws_floats", i64),
        ("defer_epilogue", i32), ("splits_used", C.POINTER(i32)),
        ("tile_override", i32),
        ("A_packed", vp),
        ("no_xcd_remap", i32),
        ("grp_imgs", i32), ("a_grp_stride", i64), ("a_packed_grp_stride", i64),
        ("no_wide_epilogue", i32),
        ("stat_partials", vp), ("stat_tiles", C.POINTER(i32)),
        ("A_split", vp), ("a_split_grp_stride", i64),
        ("split_ok", i32),
        ("img_run_flags", vp),
        ("fuse_A_split", vp), ("fuse_a_grp_stride", i64), ("fuse_scale", vp), ("fuse_shift", vp), ("fuse_M", i32),
        ("residual_after_relu", i32),
        ("real_taps", i32),
    ]


class RednetOp(C.Structure):
    """include/ivln_hip.h: ivln_rednet_op."""
    _fields_ = [("kind", i32), ("i", i32 * 7), ("f", f32 * 2), ("n", i64), ("src0", vp), ("src1", vp), ("dst", vp),
                ("gemm", GemmDesc)]


OP_GEMM, OP_ADD, OP_POOL, OP_RGB_NORM, OP_AFFINE, OP_ARGMAX_U8 = range(6)


class OpRecorder:
    """While active (`with ops.recording() as rec:`) every launch of the kinds ivln_rednet_fwd knows is ALSO appended
    to `rec.ops` with its pointers resolved, and every tensor whose pointer was handed to the library is kept alive in
    `rec.keep` - so the recorded table stays valid and one C call can repeat the whole sequence (rednet.RedNetPlan)."""

    def __init__(self):
        self.ops, self.keep, self.unsupported = [], [], []

    def table(self):
        arr = (RednetOp * len(self.ops))()
        for k, op in enumerate(self.ops):
            C.memmove(C.byref(arr, k * C.sizeof(RednetOp)), C.byref(op), C.sizeof(RednetOp))
        return arr


_REC = None


class recording:
    def __enter__(self):
        global _REC
        if _REC is not None:
            raise _lib.IvlnError("ops.recording() does not nest")
        _REC = OpRecorder()
        return _REC

    def __exit__(self, *exc):
        global _REC
        _REC = None
        return False


def _rec(kind, ints=(), floats=(), n=0, src0=None, src1=None, dst=None, gemm=None):
    op = RednetOp()
    op.kind = kind
    for k, v in enumerate(ints):
        op.i[k] = int(v)
    for k, v in enumerate(floats):
        op.f[k] = float(v)
    op.n, op.src0, op.src1, op.dst = int(n), src0, src1, dst
    if gemm is not None:
        C.memmove(C.byref(op, RednetOp.gemm.offset), C.byref(gemm), C.sizeof(GemmDesc))
        op.gemm.splits_used = None  # an output of the direct call only
    _REC.ops.append(op)


A_MK, A_KM, A_NCHW_P = 0, 1, 2
B_CONV, B_CONV1X1, B_KN, B_NK, B_IM2COL_T, B_CONVT, B_CONV_K3, B_CONV_K7, B_CONV_K2 = 0, 1, 2, 3, 4, 5, 6, 7, 8
D_NCHW, D_DENSE, D_NCHW_UP2, D_NCHW_UP2X4 = 0, 1, 2, 3

_sigs_done = False


def _L():
    global _sigs_done
    L = lib()
    if not _sigs_done:
        L.ivln_gemm_f32.argtypes = [C.POINTER(GemmDesc), vp]
        L.ivln_groupnorm_f32.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, i64, i64, i32, i64, i64, i64,
                                         vp, vp, vp]
        L.ivln_bn_fold_f32.argtypes = [vp, vp, vp, vp, vp, f32, i32, vp, vp, vp]
        L.ivln_bn_train_stats_f32.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp, vp, i64, vp]
        L.ivln_scale_shift_relu_avgpool2_f32.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i64, i64, i32, i64, vp]
        L.ivln_pool2d_f32.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
        L.ivln_map_features_f32.argtypes = [vp, vp, vp, i32, i32, i32, vp]
        L.ivln_embed_lengths.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp]
        L.ivln_lstm_bidir_fwd_f32.argtypes = [vp] * 7 + [i32, i32, i32, vp, vp, vp, vp]
        L.ivln_linear_skinny_f32.argtypes = [vp, i64, vp, vp, vp, i64, i32, i32, i32, i32, vp]
        L.ivln_gru_step_f32.argtypes = [vp, i64, i32, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, i64, vp, i64, i32, i32,
                                        vp, vp, vp, vp, vp]
        L.ivln_attn_fwd_f32.argtypes = [vp, i64, vp, i64, vp, i64, vp, f32, i32, i32, i32, i32, vp, i64, vp, vp, vp]
        L.ivln_prev_action_embed_f32.argtypes = [vp, vp, vp, i32, i32, i32, vp, i64, vp, i64, vp]
        L.ivln_argmax_rows.argtypes = [vp, i32, i32, vp, vp]
        L.ivln_argmax_channels_u8.argtypes = [vp, i32, i32, i32, vp, vp]
        L.ivln_rgb_resize_normalize_f32.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp]
        L.ivln_affine_f32.argtypes = [vp, vp, i64, f32, f32, vp]
        L.ivln_add_f32.argtypes = [vp, vp, vp, i64, i32, vp]
        L.ivln_copy2d_f32.argtypes = [vp, i64, vp, i64, i32, i32, i32, vp]
        L.ivln_tour_memory_f32.argtypes = [vp, i64, vp, i64, vp, i32, i32, vp, i64, vp, i64, vp]
        _sigs_done = True
    return L


def dptr(t: torch.Tensor) -> int:
    """Device pointer of a contiguous GPU tensor (kept alive by an active OpRecorder)."""
    if _REC is not None:
        _REC.keep.append(t)
    return _dptr(t)


def _p(t: Optional[torch.Tensor]):
    """Device pointer; unlike dptr() allows strided views (caller passes the strides)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.IvlnError("HIP hot path needs GPU tensors (no CPU fallback); got " + str(t.device))
    if _REC is not None:
        _REC.keep.append(t)
    return t.data_ptr()


# ---- per-device caches (index tables, split-K workspace) -----------------------------------------
_tables = {}
_ws = {}


def conv_tables(cin, kh, kw, hin, win, dil, device, transposed=False):
    key = (cin, kh, kw, hin, win, dil, str(device), transposed)
    t = _tables.get(key)
    if t is None:
        ci = torch.arange(cin, dtype=torch.int64).view(-1, 1, 1)
        a = torch.arange(kh, dtype=torch.int64).view(1, -1, 1)
        b = torch.arange(kw, dtype=torch.int64).view(1, 1, -1)
        if transposed:
            koff = (ci * hin * win + 0 * a + 0 * b).reshape(-1)
        else:
            koff = (ci * hin * win + a * dil * win + b * dil).reshape(-1)
        kpos = (0 * ci + (a << 16) + b).reshape(-1)
        t = (koff.to(torch.int32).to(device), kpos.to(torch.int32).to(device))
        _tables[key] = t
    return t


def splitk_ws(device, floats=8 << 20, slot=0):
    """Split-K / deferred-epilogue workspace, one per (device, stream): concurrent branches of the
    captured step graph must not share it.  slot 1 is a second workspace for a deferred conv whose slabs
    must outlive later deferred convs (the bottleneck's downsample branch)."""
    key = (str(device), stream_ptr(), slot)
    w = _ws.get(key)
    if w is None or w.numel() < floats:
        w = torch.empty(floats, dtype=torch.float32, device=device)
        _ws[key] = w
    return w


TILE_OVERRIDE = 0  # tuning/tests: force a block tile (1..5), see ivln_gemm_desc.tile_override
LINEAR_BWD_SPLIT = True  # split-K in Linear dX / accumulating dW
NO_XCD_REMAP = False  # tests / A-B: identity workgroup -> tile mapping (ivln_gemm_desc.no_xcd_remap)
NO_WIDE_EPILOGUE = False  # tests: 4-byte MFMA-layout stores for NCHW outputs
PACK_WEIGHTS = True  # A/B switch: pre-arranged weights for the direct conv kernel


def gemm(desc: GemmDesc):
    if TILE_OVERRIDE:
        desc.tile_override = TILE_OVERRIDE
    if NO_XCD_REMAP:
        desc.no_xcd_remap = 1
    if NO_WIDE_EPILOGUE:
        desc.no_wide_epilogue = 1
    check(_L().ivln_gemm_f32(C.byref(desc), stream_ptr()), "ivln_gemm_f32")
    if _REC is not None:
        _rec(OP_GEMM, gemm=desc)


def gemm_soft(desc: GemmDesc) -> bool:
    """ivln_gemm_f32 for descriptors that only some kernels take (fuse_*, residual_after_relu): False when the library
    declines (IVLN_E_UNSUPPORTED: the caller takes its other route), True when the launch went out."""
    rc = _L().ivln_gemm_f32(C.byref(desc), stream_ptr())
    if rc == _lib.IVLN_E_UNSUPPORTED:
        return False
    check(rc, "ivln_gemm_f32")
    if _REC is not None:
        _rec(OP_GEMM, gemm=desc)
    return True


_WORK_STREAMS = {}
EAGER_WORK_STREAM = os.environ.get("IVLN_EAGER_WORK_STREAM", "1") != "0"


@contextlib.contextmanager
def eager_work_stream(device=None):
    """Run an eager many-launch region on a dedicated stream that never launches hipGraphs.  Measured on MI355X /
    ROCm 7.2 (profiles/README.md, round 3): the same DAgger update takes 13.6 ms per step in a fresh process and 17.5 ms
    once the stream it runs on has replayed the rollout step's graphs - about 2 us more of host time on each of its
    ~2000 launches, with identical kernel time - whether that stream is the null stream or a stream of the caller's;
    on a stream of its own it is back at 13.8 ms.  The region is ordered behind the caller's stream on entry and the
    caller's stream behind it on exit, so surrounding code sees no difference."""
    if not EAGER_WORK_STREAM or not torch.cuda.is_available():
        yield
        return
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.type != "cuda" or torch.cuda.is_current_stream_capturing():
        yield
        return
    work = _WORK_STREAMS.get(dev.index)
    if work is None:
        work = _WORK_STREAMS[dev.index] = torch.cuda.Stream(dev)
    cur = torch.cuda.current_stream(dev)
    if cur == work:
        yield
        return
    work.wait_stream(cur)
    try:
        with torch.cuda.stream(work):
            yield
    finally:
        cur.wait_stream(work)


def _epilogue(d: GemmDesc, scale, shift, residual, relu, accumulate=False):
    d.scale, d.shift, d.residual = _p(scale), _p(shift), _p(residual)
    d.relu, d.accumulate = int(bool(relu)), int(bool(accumulate))


_packed = {}


def packed_conv_weights(w, cache=True, split=False):
    """Weights of a stride-1 3x3 / 7x7 / 2x2 conv in the direct kernel's LDS order - or, `split`, as three bf16 pieces per
    value in the bf16-MFMA direct conv's per-lane order (csrc/conv_bf3.hip; a float32 tensor of 4-byte words).  Long-lived
    tensors (module parameters) are cached until they change (tensor version / WEIGHT_EPOCH) or die (weak reference: a
    freed address may be handed to another tensor); temporaries (`cache=False`, e.g. the flipped weights of an input
    gradient) are packed into a fresh buffer every call so that nothing accumulates."""
    import weakref

    G = w.shape[0] if w.dim() == 5 else 1  # (G, Cout, Cin, k, k): image-grouped weight sets, packed one after another
    Cout, Cin, KH, KW = w.shape[-4:]
    L = _L()
    L.ivln_conv_packed_floats.restype = i64
    L.ivln_conv_packed_floats.argtypes = [i32, i32, i32]
    L.ivln_conv_pack_weights_f32.argtypes = [vp, i32, i32, i32, vp, vp]
    L.ivln_conv_split_words.restype = i64
    L.ivln_conv_split_words.argtypes = [i32, i32, i32]
    L.ivln_conv_split_weights_f32.argtypes = [vp, i32, i32, i32, vp, vp]
    L.ivln_conv_stem_split_words.restype = i64
    L.ivln_conv_stem_split_words.argtypes = [i32, i32]
    L.ivln_conv_stem_split_weights_f32.argtypes = [vp, i32, i32, vp, vp]
    stem = split == "stem"  # (7x7 stride-2 convs of 1 or 3 channels: K as kernel rows x 8 columns, k_conv7s2_bf3)
    if stem:
        n1 = L.ivln_conv_stem_split_words(Cout, Cin) if (G == 1 and KH == 7 and KW == 7) else 0
    else:
        n1 = L.ivln_conv_split_words(Cout, Cin, KH) if split else L.ivln_conv_packed_floats(Cout, Cin, KH)
    if n1 <= 0:
        return None
    n = n1 * G
    wsz = Cout * Cin * KH * KW
    fn, fn_name = ((L.ivln_conv_split_weights_f32, "ivln_conv_split_weights_f32") if split
                   else (L.ivln_conv_pack_weights_f32, "ivln_conv_pack_weights_f32"))

    def _pack(dst):
        if stem:
            check(L.ivln_conv_stem_split_weights_f32(w.data_ptr(), Cout, Cin, dst.data_ptr(), stream_ptr()), "ivln_conv_stem_split_weights_f32")
            return
        for g in range(G):
            check(fn(w.data_ptr() + 4 * g * wsz, Cout, Cin, KH, dst.data_ptr() + 4 * g * n1, stream_ptr()), fn_name)

    if not cache:
        out = torch.empty(n, dtype=torch.float32, device=w.device)
        _pack(out)
        return out
    key = (w.data_ptr(), tuple(w.shape), split)
    stamp = (w._version, WEIGHT_EPOCH)
    cur = torch.cuda.current_stream()
    hit = _packed.get(key)
    if hit is not None and hit.ref() is w and hit.stamp == stamp:
        if hit.settled or hit.stream == cur.cuda_stream:
            return hit.out
        if not torch.cuda.is_current_stream_capturing():
            cur.wait_event(hit.event)  # packed on another stream: order this stream behind it
            return hit.out
        # capturing on a stream that never saw the packing and nobody settled the cache: pack again (into a
        # buffer of the capture's own pool), as a node of this graph
    capturing = torch.cuda.is_current_stream_capturing()
    same = hit is not None and hit.ref() is w  # same live tensor, new contents: refresh its buffer in place
    if same and not capturing:
        if hit.stream != cur.cuda_stream:
            cur.wait_event(hit.event)
        out = hit.out
    else:
        out = torch.empty(n, dtype=torch.float32, device=w.device)
    _pack(out)
    if capturing:
        return out  # lives in this graph's pool, valid inside this graph only: not cached
    if len(_packed) > 1024:
        _packed.clear()
    ev = torch.cuda.Event()
    ev.record(cur)
    _packed[key] = _Packed(stamp, out, weakref.ref(w), cur.cuda_stream, ev)
    return out


class _Packed:
    __slots__ = ("stamp", "out", "ref", "stream", "event", "settled")

    def __init__(self, stamp, out, ref, stream, event):
        self.stamp, self.out, self.ref, self.stream, self.event, self.settled = stamp, out, ref, stream, event, False


def settle_packed_weights():
    """Call right after a device-wide synchronize: every packed buffer is complete, so any stream - and any
    graph captured from now on - may read it without ordering (graphed.py calls this before capturing; without
    it a capture on a stream other than the one that packed would record the packing kernels as graph nodes
    and replay them every step)."""
    for e in _packed.values():
        e.settled = True


class Deferred:
    """Raw conv output left in the split-K workspace: `splits` slabs of a [C][N*HW] matrix."""

    def __init__(self, ws, splits, N, C, H, W):
        self.ws, self.splits, self.N, self.C, self.H, self.W = ws, splits, N, C, H, W


# 3x3 / 7x7 convs with both operands as three bf16 pieces each on the bf16 MFMA pipe (csrc/conv_bf3.hip): exact pieces, six
# of the nine piece products, fp32 accumulation - as close to the exact conv as the fp32 MFMA kernels.
# IVLN_SPLIT_BF16=0 keeps the fp32 MFMA kernels everywhere (A/B); the C side has IVLN_NO_SPLIT_BF16 for the same.
SPLIT_BF16 = os.environ.get("IVLN_SPLIT_BF16", "1") != "0"
# the 7x7 weight gradients on the same arithmetic (k_wgrad_bf3; 1.1-1.7x the fp32 MFMA weight-gradient kernel at the update's shapes)
SPLIT_BF16_WGRAD = os.environ.get("IVLN_SPLIT_BF16_WGRAD", "1") != "0"
SPLIT_BF16_1X1 = int(os.environ.get("IVLN_SPLIT_BF16_1X1", "-1"))  # -1: by measured rule (ops.conv2d), 0 never, 1 always
BF3_1X1_KS = os.environ.get("IVLN_BF3_1X1_KS", "1") != "0"  # A/B: 0 = deep-K 1x1 convs stay on the fp32 GEMM kernels
BF3_CONVT = os.environ.get("IVLN_BF3_CONVT", "1") != "0"  # A/B: 0 = stride-2 3x3 transposed convs stay on the fp32 direct kernel
BF3_S2 = os.environ.get("IVLN_BF3_S2", "1") != "0"  # A/B: 0 = stride-2 3x3 convs stay on the fp32 direct kernel
BF3_STEM = os.environ.get("IVLN_BF3_STEM", "1") != "0"  # A/B: 0 = RedNet's 7x7 stride-2 stems stay on the fp32 direct kernel (+ the fusion add as a launch)
S2_GATHER = os.environ.get("IVLN_S2_GATHER", "1") != "0"  # A/B: 0 = stride-2 1x1 convs read their input strided (tiled 1x1 form)
SPLIT_BF16_MIN_OUT = 1 << 18  # output elements below which nothing is packed
_stat_ws = {}
CONV_STATS = os.environ.get("IVLN_CONV_STATS", "1") != "0"  # A/B: BatchNorm statistics from the conv's epilogue


def conv_stat_ws(device, floats):
    """Per-(device, stream) buffer for the per-tile statistics a conv launch leaves behind (ivln_gemm_desc.stat_partials)."""
    key = (str(device), stream_ptr())
    w = _stat_ws.get(key)
    if w is None or w.numel() < floats:
        w = torch.empty(floats, dtype=torch.float32, device=device)
        _stat_ws[key] = w
    return w


def conv2d(x, w, stride=1, pad=0, dil=1, scale=None, shift=None, residual=None, relu=False, out=None, out_ctot=0,
           in_img_stride=0, splitk=True, defer=False, ws_slot=0, weight_is_temp=False, stats=None, run_flags=None,
           residual_after_relu=False):
    """NCHW conv: x (N,Cin,H,W) [contiguous per image, image stride `in_img_stride`], w OIHW.
    out: optional destination (a channel slice of an (N,out_ctot,Ho,Wo) buffer).
    run_flags: optional int32 (N) on the device - output tiles all of whose images carry 0 are skipped and `out` (then a
    persistent buffer of the caller) keeps what it held (ivln_gemm_desc.img_run_flags).
    residual_after_relu: out = relu(scale * conv + shift) + residual (RedNet's decoder skips); only the stride-1 1x1
    split-bf16 kernels have that epilogue - returns None when the library declines (the caller issues conv + add)."""
    N, Cin, H, W = x.shape
    G = w.shape[0] if w.dim() == 5 else 0  # (G, Cout, Cin, k, k): weight set g for images [g*N/G, (g+1)*N/G)
    Cout, _, KH, KW = w.shape[-4:]
    if (S2_GATHER and stride == 2 and KH == 1 and KW == 1 and pad == 0 and SPLIT_BF16 and BF3_1X1_KS and not TILE_OVERRIDE
            and Cin >= 64 and Cin % 16 == 0 and H % 2 == 0 and W % 8 == 0 and in_img_stride == 0 and not defer and x.is_contiguous()
            and N * (H // 2) * (W // 2) * Cout >= SPLIT_BF16_MIN_OUT):
        # a stride-2 1x1 conv (RedNet's downsample branches, rednet.py:226-232) reads every second pixel of every second row:
        # gathered once into a dense quarter-size tensor (one pooling launch with a 1 x 1 window, a quarter of the input's
        # bytes), it IS a stride-1 1x1 conv and takes the register-built split-bf16 kernels (k_conv1x1_bf3_ks) instead of the
        # tiled 1x1 form - 47-52 us per launch at 8 + 8 images, the pipe a fifth busy (profiles/r05_predsem_B8_*)
        xs = pool2d(x, 1, 2, 0, "max")
        return conv2d(xs, w, 1, 0, dil, scale, shift, residual, relu, out, out_ctot, 0, splitk, defer, ws_slot, weight_is_temp,
                      stats, run_flags, residual_after_relu)
    Ho = (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1
    if defer:
        out = splitk_ws(x.device, slot=ws_slot)  # D is unused by the kernel in deferred mode
    elif out is None:
        out = torch.empty((N, Cout, Ho, Wo), dtype=torch.float32, device=x.device)
    d = GemmDesc()
    d.A, d.B, d.D = dptr(w), _p(x), _p(out)
    d.M, d.N, d.K = Cout, N * Ho * Wo, Cin * KH * KW
    d.amode, d.dmode = A_MK, D_NCHW
    d.lda = d.K
    d.Cin, d.Hin, d.Win, d.Hout, d.Wout = Cin, H, W, Ho, Wo
    d.stride, d.pad, d.dil = stride, pad, dil
    d.HoWo, d.Ctot = Ho * Wo, out_ctot
    d.in_img_stride = in_img_stride
    if G:
        if N % G or not w.is_contiguous():
            raise _lib.IvlnError("image-grouped conv: N must be a multiple of the weight sets, weights contiguous")
        d.grp_imgs, d.a_grp_stride = N // G, Cout * Cin * KH * KW
    if KH == 1 and KW == 1 and pad == 0:
        d.bmode = B_CONV1X1
        # short-K 1x1 convs over many pixels: the streaming kernel's per-lane weight image (conv1x1_stream.hip)
        if stride == 1 and Cin in (128, 256) and N * Ho * Wo >= 4096 and PACK_WEIGHTS and w.is_contiguous() and not defer:
            pk = packed_conv_weights(w, cache=not weight_is_temp)
            if pk is not None:
                d.A_packed = dptr(pk)
                d.a_packed_grp_stride = pk.numel() // max(G, 1)
        # 1x1 convs on the split-bf16 kernel (four 16-channel chunks staged per barrier pair): an activation element is
        # re-used only Cout times, so the split's VALU work is a large share - measured inside RedNet it wins on the stride-2
        # and the >= 4 GFLOP launches (53 vs 76, 50 vs 66, 42 vs 56, 46 vs 53 us) and ties or loses on the 2 GFLOP ones
        # (36-39 vs 32-37): on for the former only (IVLN_SPLIT_BF16_1X1=0 / 1 = never / always, tile_override 9 in tests)
        big_1x1 = stride == 2 or 2.0 * Cout * Cin * N * Ho * Wo >= 4e9
        # (round 5) stride-1 1x1 convs on k_conv1x1_bf3_ks, which builds its fragments in registers (no LDS in the K loop):
        # K split over the waves of a workgroup from 512 input channels, a 32 x 128 tile per wave up to 256 (the C side decides
        # by chunk count and grid size)
        # (maps only: Conv1d-shaped inputs - H = 1 - never fill a pixel tile, and their weights would be re-packed after every update)
        deep_1x1 = stride == 1 and Cin >= 64 and Cin % 16 == 0 and (Ho * Wo) % 4 == 0 and (Ho >= 2 or TILE_OVERRIDE >= 9) and BF3_1X1_KS
        want_1x1 = TILE_OVERRIDE >= 9 or SPLIT_BF16_1X1 == 1 or (SPLIT_BF16_1X1 != 0 and (big_1x1 or deep_1x1))
        if (want_1x1 and SPLIT_BF16 and stride in (1, 2) and (Cin >= 128 or deep_1x1) and (Cout >= 64 or deep_1x1 or TILE_OVERRIDE in (11, 12, 13)) and Wo % 4 == 0
                and (Wo >= 8 or deep_1x1 or TILE_OVERRIDE in (11, 12, 13)) and w.is_contiguous()
                and not defer and (N * Ho * Wo * Cout >= SPLIT_BF16_MIN_OUT or TILE_OVERRIDE >= 9 or residual_after_relu)):
            sp = packed_conv_weights(w, cache=not weight_is_temp, split=True)
            if sp is not None:
                d.A_split = dptr(sp)
                d.a_split_grp_stride = sp.numel() // max(G, 1)
    elif KH == KW and KH in (3, 7) and dil == 1:
        d.bmode = B_CONV_K3 if KH == 3 else B_CONV_K7
        if stride in (1, 2) and PACK_WEIGHTS and w.is_contiguous():
            pk = packed_conv_weights(w, cache=not weight_is_temp)
            if pk is not None:
                d.A_packed = dptr(pk)
                d.a_packed_grp_stride = pk.numel() // max(G, 1)
        # same-size stride-1 convs over enough pixels: both operands as three bf16 pieces on the bf16 MFMA pipe
        # (csrc/conv_bf3.hip; the C side decides per shape and falls back to the fp32 MFMA kernels)
        # ... and (round 6) RedNet's stride-2 3x3 convs: the same kernel with its patch staged as four phase planes
        s2_ok = stride == 2 and KH == 3 and BF3_S2 and H % 2 == 0 and W % 4 == 0 and Cin % 16 == 0
        # ... and RedNet's stems (7x7, stride 2, 3 | 1 channels, rednet.py:201-210): K as kernel rows x 8 columns, fragments built in
        # registers from four 16-byte loads per lane and kernel row (k_conv7s2_bf3) - the only kernel that reads THIS weight image
        stem_ok = (BF3_STEM and SPLIT_BF16 and stride == 2 and KH == 7 and pad == 3 and Cin in (1, 3) and not G and H == 2 * Ho
                   and W == 2 * Wo and Wo % 128 == 0 and w.is_contiguous() and not defer and not TILE_OVERRIDE and stats is None
                   and run_flags is None)
        if stem_ok:
            sp = packed_conv_weights(w, cache=not weight_is_temp, split="stem")
            if sp is not None:
                d.A_split = dptr(sp)
        if (not stem_ok and SPLIT_BF16 and (stride == 1 or s2_ok) and pad == KH // 2 and Wo % 4 == 0 and Wo >= 8 and w.is_contiguous() and not defer
                and (N * Ho * Wo * Cout >= SPLIT_BF16_MIN_OUT or TILE_OVERRIDE >= 9)):
            sp = packed_conv_weights(w, cache=not weight_is_temp, split=True)
            if sp is not None:
                d.A_split = dptr(sp)
                d.a_split_grp_stride = sp.numel() // max(G, 1)
    else:
        d.bmode = B_CONV
        koff, kpos = conv_tables(Cin, KH, KW, H, W, dil, x.device)
        d.koff, d.kpos = dptr(koff), dptr(kpos)
    _epilogue(d, scale, shift, residual, relu)
    if run_flags is not None:
        if defer or run_flags.dtype != torch.int32 or run_flags.numel() < N or not run_flags.is_cuda:
            raise _lib.IvlnError("conv2d(run_flags): int32 flags per image on the device, not with defer")
        d.img_run_flags, splitk = dptr(run_flags), False
        d._run_flags = run_flags  # (python-side handle: bench.py's instrumented pass reads it back to count executed FLOPs)
    if defer:
        ws = splitk_ws(x.device, slot=ws_slot)
        used = i32(0)
        d.ws, d.ws_floats, d.splits, d.defer_epilogue = dptr(ws), ws.numel(), 0, 1
        d.splits_used = C.pointer(used)
        gemm(d)
        return Deferred(ws, used.value, N, Cout, Ho, Wo)
    if splitk:
        ws = splitk_ws(x.device)
        d.ws, d.ws_floats, d.splits = dptr(ws), ws.numel(), 0
    else:
        d.splits = 1
    if residual_after_relu:
        if residual is None or defer or stats is not None or run_flags is not None or TILE_OVERRIDE:
            return None
        d.residual_after_relu, d.splits = 1, 1
        return out if gemm_soft(d) else None
    if stats is not None:  # `stats` = a list: receives (partials, tiles) when the launch produced per-tile statistics
        tiles_max = (N * Ho * Wo + 31) // 32  # (pixel tiles hold 128 outputs; 4x headroom for ragged tilings)
        sp = conv_stat_ws(x.device, tiles_max * Cout * 3)
        n_tiles = i32(0)
        d.stat_partials, d.stat_tiles = dptr(sp), C.pointer(n_tiles)
        gemm(d)
        if n_tiles.value > 0:
            stats.append((sp, n_tiles.value))
        return out
    gemm(d)
    return out


_ALL_ROWS = {}


def all_rows_flags(rows, device):
    """int32 ones (rows) on the device: the `run_flags` of a conv that recomputes every image (long-lived: captured graphs
    hold the address)."""
    k = (str(device), int(rows))
    if k not in _ALL_ROWS:
        _ALL_ROWS[k] = torch.ones((int(rows),), dtype=torch.int32, device=device)
    return _ALL_ROWS[k]


BF3_FUSE = os.environ.get("IVLN_BF3_FUSE", "1") != "0"


def conv3x3_then_1x1(x, w2, scale2, shift2, w3, scale3, shift3, residual):
    """A bottleneck's tail as ONE launch (rednet.py:20-65): relu(bn3(conv1x1(relu(bn2(conv3x3(x))))) + residual), the folded
    BatchNorms given as scale / shift.  Weights (Cout, Cin, 3, 3) / (Cout3, Cout, 1, 1), or image-grouped (G, ...) pairs.
    Returns None when the library declines the shape (ivln_gemm_desc.fuse_*: 64 or 128 mid channels, stride 1, width a
    multiple of 32, height of 4) - the caller then issues the two convs."""
    if not (BF3_FUSE and SPLIT_BF16) or TILE_OVERRIDE:
        return None
    N, Cin, H, W = x.shape
    G = w2.shape[0] if w2.dim() == 5 else 0
    Cmid, Cout = w2.shape[-4], w3.shape[-4]
    if (Cmid not in (64, 128) or w2.shape[-1] != 3 or w3.shape[-1] != 1 or w3.shape[-3] != Cmid or W % 32 or H % 4 or Cin % 16 or Cout % 32
            or (w3.dim() == 5) != bool(G) or not (w2.is_contiguous() and w3.is_contiguous() and x.is_contiguous()) or (G and N % G)):
        return None
    sp2, sp3 = packed_conv_weights(w2, split=True), packed_conv_weights(w3, split=True)
    if sp2 is None or sp3 is None:
        return None
    out = torch.empty((N, Cout, H, W), dtype=torch.float32, device=x.device)
    d = GemmDesc()
    d.A, d.B, d.D = dptr(w2), _p(x), _p(out)
    d.M, d.N, d.K = Cmid, N * H * W, Cin * 9
    d.amode, d.bmode, d.dmode = A_MK, B_CONV_K3, D_NCHW
    d.lda = d.K
    d.Cin, d.Hin, d.Win, d.Hout, d.Wout = Cin, H, W, H, W
    d.stride, d.pad, d.dil = 1, 1, 1
    d.HoWo, d.Ctot = H * W, Cout
    if G:
        d.grp_imgs, d.a_grp_stride = N // G, Cmid * Cin * 9
    d.A_split, d.a_split_grp_stride = dptr(sp2), sp2.numel() // max(G, 1)
    d.fuse_A_split, d.fuse_a_grp_stride, d.fuse_M = dptr(sp3), sp3.numel() // max(G, 1), Cout
    d.fuse_scale, d.fuse_shift = _p(scale3), _p(shift3)
    _epilogue(d, scale2, shift2, residual, True)
    d.splits = 1
    return out if gemm_soft(d) else None


def bn_stats_from_partials(partials, tiles, bn, scale, shift, save_mean=None, save_rstd=None, update_running=True):
    """BatchNorm (train mode) scale / shift / saved statistics / running statistics from the per-tile partials the
    producing conv left behind (conv2d(..., stats=[])): no pass over the conv's output."""
    mom = bn.momentum if bn.momentum is not None else 0.1
    L = _L()
    L.ivln_bn_stats_from_partials_f32.argtypes = [vp, i32, i32, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp, vp]
    check(
        L.ivln_bn_stats_from_partials_f32(dptr(partials), tiles, bn.num_features, dptr(bn.weight), dptr(bn.bias),
                                          dptr(bn.running_mean) if update_running else None,
                                          dptr(bn.running_var) if update_running else None, mom, bn.eps, dptr(scale),
                                          dptr(shift), _p(save_mean), _p(save_rstd), stream_ptr()),
        "ivln_bn_stats_from_partials_f32",
    )


def conv_transpose2d(x, w_oihw, stride, pad, out_pad, scale=None, shift=None, residual=None, relu=False, out=None):
    """nn.ConvTranspose2d with weights pre-arranged as (Cout,Cin,KH,KW)."""
    N, Cin, H, W = x.shape
    Cout, _, KH, KW = w_oihw.shape
    Ho = (H - 1) * stride - 2 * pad + KH + out_pad
    Wo = (W - 1) * stride - 2 * pad + KW + out_pad
    if out is None:
        out = torch.empty((N, Cout, Ho, Wo), dtype=torch.float32, device=x.device)
    d = GemmDesc()
    d.A, d.B, d.D = dptr(w_oihw), dptr(x), dptr(out)
    d.M, d.N, d.K = Cout, N * Ho * Wo, Cin * KH * KW
    d.amode, d.bmode, d.dmode = A_MK, B_CONVT, D_NCHW
    d.lda = d.K
    d.Cin, d.Hin, d.Win, d.Hout, d.Wout = Cin, H, W, Ho, Wo
    d.stride, d.pad, d.dil = stride, pad, 1
    d.HoWo = Ho * Wo
    koff, kpos = conv_tables(Cin, KH, KW, H, W, 1, x.device, transposed=True)
    d.koff, d.kpos = dptr(koff), dptr(kpos)
    _epilogue(d, scale, shift, residual, relu)
    ws = splitk_ws(x.device)
    d.ws, d.ws_floats, d.splits = dptr(ws), ws.numel(), 0
    gemm(d)
    return out


def convt_s2_classes(w_t, pad):
    """Split a stride-2 ConvTranspose2d weight (Cin,Cout,k,k) into its four output-parity classes.
    Output row oh = 2i + a only receives taps kh = a + pad - 2*dh (dh = input row offset, ih = i + dh), so
    each class is an ordinary correlation with a (1|2)x(1|2) kernel over the un-dilated input: 4x fewer
    MACs than gathering all k*k taps per output pixel (3 of 4 are structurally zero).
    Returns [(a, b, w_sub (Cout,Cin,ta,tb))] or None when a class would need a negative offset."""
    Cin, Cout, KH, KW = w_t.shape
    out = []
    for a in (0, 1):
        khs = [(dh, a + pad - 2 * dh) for dh in range(0, KH) if 0 <= a + pad - 2 * dh < KH]
        neg = [dh for dh in range(-KH, 0) if 0 <= a + pad - 2 * dh < KH]
        for b in (0, 1):
            kws = [(dw, b + pad - 2 * dw) for dw in range(0, KW) if 0 <= b + pad - 2 * dw < KW]
            negw = [dw for dw in range(-KW, 0) if 0 <= b + pad - 2 * dw < KW]
            if neg or negw or not khs or not kws or [d for d, _ in khs] != list(range(len(khs))) or \
                    [d for d, _ in kws] != list(range(len(kws))):
                return None
            sub = w_t[:, :, [k for _, k in khs]][:, :, :, [k for _, k in kws]]  # (Cin,Cout,ta,tb)
            out.append((a, b, sub.permute(1, 0, 2, 3).contiguous()))
    return out


def convt_s2_stack(classes):
    """The four parity classes of convt_s2_classes as ONE weight (4*Cout, Cin, t, t), rows 4*co + cls with
    cls = 2a + b (include/ivln_hip.h IVLN_D_NCHW_UP2X4: the four rows of a channel are one 2x2 output block per input
    pixel, so a tile leaves as 16-byte stores); classes with fewer taps are zero-padded to the common (t x t) window
    (k=3: 16 instead of 9 tap products per input pixel, bought back by one launch that reads the input once and
    writes whole output rows)."""
    t = max(max(w.shape[2], w.shape[3]) for _, _, w in classes)
    Cout, Cin = classes[0][2].shape[:2]
    W = torch.zeros((Cout, 4, Cin, t, t), dtype=torch.float32, device=classes[0][2].device)
    for a, b, w in classes:
        W[:, 2 * a + b, :, :w.shape[2], :w.shape[3]] = w
    return W.view(4 * Cout, Cin, t, t).contiguous()


STACK_CONVT = os.environ.get("IVLN_CONVT_STACK", "1") != "0"


def conv_transpose2d_s2(x, classes, scale=None, shift=None, residual=None, relu=False, out=None, stacked=None):
    """nn.ConvTranspose2d(stride=2) whose output is exactly (2H, 2W) (k=3,p=1,op=1 / k=2,p=0) through its four
    output-parity classes (convt_s2_classes) with the fused epilogue (rednet.py:210-216,262-279 upsampling blocks):
    ONE implicit GEMM over the stacked classes (`stacked` = convt_s2_stack(classes), D_NCHW_UP2X4), or one per class."""
    N, Cin, H, W = x.shape
    Cout = classes[0][2].shape[0]
    if out is None:
        out = torch.empty((N, Cout, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    ws = splitk_ws(x.device)
    todo = [(0, 0, stacked)] if (stacked is not None and STACK_CONVT) else classes
    for a, b, w in todo:
        ta, tb = w.shape[2], w.shape[3]
        d = GemmDesc()
        d.A, d.B, d.D = dptr(w), dptr(x), dptr(out)
        d.M, d.N, d.K = w.shape[0], N * H * W, Cin * ta * tb
        d.amode, d.dmode = A_MK, (D_NCHW_UP2X4 if w.shape[0] != Cout else D_NCHW_UP2)
        d.lda = d.K
        d.Cin, d.Hin, d.Win, d.Hout, d.Wout = Cin, H, W, H, W
        d.stride, d.pad, d.dil = 1, 0, 1
        d.HoWo = H * W
        d.Ctot = Cout
        d.sDm, d.sDn = a, b
        if ta == 1 and tb == 1:
            d.bmode = B_CONV1X1
            # the stacked one-tap classes of a 2 x 2 stride-2 transposed conv (RedNet's upsampling branches and final deconv) on the
            # register-built split-bf16 1x1 kernels, which store the 2 x 2 output blocks themselves (csrc/conv_bf3.hip)
            if (w is stacked and SPLIT_BF16 and BF3_CONVT and BF3_1X1_KS and Cin >= 64 and Cin % 16 == 0 and W % 4 == 0 and w.is_contiguous()
                    and (N * H * W * w.shape[0] >= SPLIT_BF16_MIN_OUT or TILE_OVERRIDE >= 9)):
                sp = packed_conv_weights(w, split=True)
                if sp is not None:
                    d.A_split = dptr(sp)
                    d.a_split_grp_stride = sp.numel()
        elif ta == 2 and tb == 2:  # the 2x2 window: LDS-staged direct kernel, no tap tables (csrc/conv_direct.hip)
            d.bmode = B_CONV_K2
            if PACK_WEIGHTS and w.is_contiguous():
                pk = packed_conv_weights(w)
                if pk is not None:
                    d.A_packed = dptr(pk)
            # the stacked classes on the split-bf16 kernels (csrc/conv_bf3.hip, KS = 2: K split over the waves for the
            # pixel-starved first upsampling stages, the tiled kernel beyond); the C side decides and falls back to the above
            if (w is stacked and SPLIT_BF16 and BF3_CONVT and Cin % 16 == 0 and W % 4 == 0 and W >= 8 and w.is_contiguous()
                    and (N * H * W * w.shape[0] >= SPLIT_BF16_MIN_OUT or TILE_OVERRIDE >= 9)):
                sp = packed_conv_weights(w, split=True)
                if sp is not None:
                    d.A_split = dptr(sp)
                    d.a_split_grp_stride = sp.numel()
                    d.real_taps = sum(c.shape[2] * c.shape[3] for _, _, c in classes)
        else:  # taps at input offsets (0..ta-1, 0..tb-1); rows/cols past the edge read as zero
            d.bmode = B_CONV
            koff, kpos = conv_tables(Cin, ta, tb, H, W, 1, x.device)
            d.koff, d.kpos = dptr(koff), dptr(kpos)
        _epilogue(d, scale, shift, residual, relu)
        d.ws, d.ws_floats, d.splits = dptr(ws), ws.numel(), 0
        if w is stacked:
            # the stacked form multiplies the zero padding of the classes' common window too (k = 3: 16 tap products per
            # input pixel for 9 real ones): FLOP counters price the ALGORITHMIC taps, not the executed ones
            d._algo_flops = 2 * Cout * d.N * Cin * sum(c.shape[2] * c.shape[3] for _, _, c in classes)
        gemm(d)
    return out


def linear_gemm(x, w, bias=None, relu=False, out=None):
    """y[r][o] = act(x[r] . w[o] + b[o]) through the MFMA GEMM (many rows)."""
    rows, K = x.shape
    O = w.shape[0]
    if out is None:
        out = torch.empty((rows, O), dtype=torch.float32, device=x.device)
    d = GemmDesc()
    d.A, d.B, d.D = dptr(w), _p(x), _p(out)
    d.M, d.N, d.K = O, rows, K
    d.amode, d.bmode, d.dmode = A_MK, B_NK, D_DENSE
    d.lda, d.ldb = K, x.stride(0)
    d.sDm, d.sDn = 1, out.stride(0)
    d.HoWo = 1
    _epilogue(d, None, bias, None, relu)
    ws = splitk_ws(x.device)
    d.ws, d.ws_floats, d.splits = dptr(ws), ws.numel(), 0
    gemm(d)
    return out


def linear(x, w, bias=None, relu=False, out=None):
    """nn.Linear forward; rows <= 16 use the wave-per-output-row kernel, else the MFMA GEMM."""
    rows, K = x.shape
    O = w.shape[0]
    if rows > 16:
        return linear_gemm(x, w, bias, relu, out)
    if out is None:
        out = torch.empty((rows, O), dtype=torch.float32, device=x.device)
    check(
        _L().ivln_linear_skinny_f32(_p(x), x.stride(0), dptr(w), _p(bias), _p(out), out.stride(0), rows, K, O,
                                    int(bool(relu)), stream_ptr()),
        "ivln_linear_skinny_f32",
    )
    return out


FUSE_KV_LINEAR = os.environ.get("IVLN_KV_LINEAR", "1") != "0"
FOLD_INSTRUCTION_GATES = os.environ.get("IVLN_FOLD_GATES", "1") != "0"  # inference: embedding + W_ih as one table lookup


def kv_linear(feat, w_kv, b_kv, w_lin, b_lin, lin_out, relu=True):
    """kv = Conv1d(C, Ckv, 1)(feat.view(rows, C, P)) and lin_out[:] = act(Linear(C*P, O)(feat.flatten(1))) in ONE launch
    (csrc/nn_ops.hip k_kv_linear) for rollout batches (rows <= 8).  Returns kv (rows, Ckv, 1, P) or None when the shape is
    outside the kernel's envelope."""
    if not FUSE_KV_LINEAR:
        return None
    rows, Cc = feat.shape[0], feat.shape[1]
    P = feat.numel() // (rows * Cc)
    Ckv, O = w_kv.shape[0], w_lin.shape[0]
    if rows > 8 or not feat.is_contiguous():
        return None
    kv = torch.empty((rows, Ckv, 1, P), dtype=torch.float32, device=feat.device)
    L = _L()
    L.ivln_kv_linear_f32.argtypes = [vp, i32, i32, i32, vp, vp, i32, vp, vp, vp, i32, i32, vp, i64, vp]
    code = L.ivln_kv_linear_f32(_p(feat), rows, Cc, P, dptr(w_kv), _p(b_kv), Ckv, _p(kv), dptr(w_lin), _p(b_lin), O,
                                int(bool(relu)), _p(lin_out), lin_out.stride(0), stream_ptr())
    if code == IVLN_E_UNSUPPORTED:
        return None
    check(code, "ivln_kv_linear_f32")
    return kv


def groupnorm(x, gamma, beta, groups, eps=1e-5, relu=False, residual=None, out=None, y_img_stride=0,
              r_img_stride=0, x2=None, gamma2=None, beta2=None):
    """x: NCHW tensor or a `Deferred` conv output (slab reduction fused).  `out` may be a channel slice
    of a wider NCHW buffer (y_img_stride = its image stride).  x2 (+gamma2, beta2): a second operand that is
    group-normalised (same groups) and added before the ReLU."""
    if isinstance(x, Deferred):
        N, Cc, H, W = x.N, x.C, x.H, x.W
        HW = H * W
        xp, x_img, x_chan, splits, slab = dptr(x.ws), HW, N * HW, x.splits, Cc * N * HW
        dev = x.ws.device
    else:
        N, Cc, H, W = x.shape
        HW = H * W
        xp, x_img, x_chan, splits, slab = _p(x), 0, 0, 1, 0
        dev = x.device
    if out is None:
        out = torch.empty((N, Cc, H, W), dtype=torch.float32, device=dev)
    if x2 is None:
        check(
            _L().ivln_groupnorm_f32(xp, dptr(gamma), dptr(beta), _p(residual), _p(out), N, Cc, HW, groups, eps,
                                    int(bool(relu)), x_img, x_chan, splits, slab, y_img_stride, r_img_stride, None,
                                    None, stream_ptr()),
            "ivln_groupnorm_f32",
        )
        return out
    if isinstance(x2, Deferred):
        assert (x2.N, x2.C, x2.H, x2.W) == (N, Cc, H, W)
        x2p, x2_img, x2_chan, splits2, slab2 = dptr(x2.ws), HW, N * HW, x2.splits, Cc * N * HW
    else:
        assert tuple(x2.shape) == (N, Cc, H, W)
        x2p, x2_img, x2_chan, splits2, slab2 = _p(x2), 0, 0, 1, 0
    L = _L()
    L.ivln_groupnorm2_f32.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, i64, i64, i32, i64, i64, i64, vp,
                                      vp, vp, vp, vp, i64, i64, i32, i64, vp]
    check(
        L.ivln_groupnorm2_f32(xp, dptr(gamma), dptr(beta), _p(residual), _p(out), N, Cc, HW, groups, eps,
                              int(bool(relu)), x_img, x_chan, splits, slab, y_img_stride, r_img_stride, None, None,
                              x2p, dptr(gamma2), dptr(beta2), x2_img, x2_chan, splits2, slab2, stream_ptr()),
        "ivln_groupnorm2_f32",
    )
    return out


IVLN_E_UNSUPPORTED = -5


# depth ResNet as a chain of GroupNorm+next-conv launches (csrc/gn_conv.hip); IVLN_GN_CONV=0 selects the deferred
# conv + GroupNorm pairs
# the whole depth encoder of a rollout batch (<= 8 images) as ONE persistent launch (csrc/depth_net.hip); 0: the per-layer
# launch chain below (A/B switch, and what runs where the persistent grid cannot be resident)
#   "1" (default): whenever the encoder is the latency-bound part of the step and the batch is <= 8 images - eager and
#       single-stream execution (one launch instead of 54: 512 vs 565 us of GPU time and no 54 host enqueues) and the split
#       replay (600-620 us per step at 1-5 envs against 640-715 for the chain, 743 against 811 at 8).  Beside the mapper /
#       map-CNN graph the persistent workgroups hold their CUs: with 218 VGPRs (two waves per SIMD leave 76 registers per
#       lane) the first neighbour kernel that needed more waited for this launch to end and the replay LOST (0.81 vs 0.72
#       ms at 4 envs); at 150 VGPRs and 117 KB of LDS the map CNN's convs fit beside it, the bi-LSTM (340 registers per
#       SIMD lane, fits beside nothing) is ordered last in its graph and, below 6 images, drawn by the blocks that land on
#       the free XCDs (lstm_bidir spare);  IVLN_DEPTH_NET_SPLIT_MIN raises the image count from which the split replay
#       takes it (A/B);
#   "2" always (<= 8 images); "0" never.
# The persistent launch spins on its own workgroups' arrivals: all 32 workgroups of a cluster have to be resident, which
# the residency check guarantees only when this process has the GPU to itself.  Two processes on one device (the one-device
# multi-rank smoke tests: IVLN_ONE_DEVICE / IVLN_BENCH_ONE_DEVICE) could each get half of the CUs and time each other out,
# so the default there is the launch chain; set IVLN_DEPTH_NET=0 for any other shared-GPU deployment.
DEPTH_NET = int(os.environ.get("IVLN_DEPTH_NET", "0" if (os.environ.get("IVLN_ONE_DEVICE") or os.environ.get("IVLN_BENCH_ONE_DEVICE")) else "1"))
DEPTH_NET_SPLIT_MIN = int(os.environ.get("IVLN_DEPTH_NET_SPLIT_MIN", "1"))
CHAIN_GN_CONV = os.environ.get("IVLN_GN_CONV", "1") != "0"
# The chain trades launches for slab bytes (16 partial slabs per conv), which pays while the step is latency-bound:
# measured 4 envs 5.1 K vs 4.0 K env-steps/s, 8 envs 7.5 K vs 6.9 K, but 16 envs 9.5 K vs 10.1 K and 32 envs 11.1 K vs
# 15.2 K - beyond 8 images per GPU the conv + GroupNorm pairs run.
CHAIN_MAX_IMAGES = 8
# the bottlenecks before the chain (layer 1) as ivln_nconv_f32 launches: GroupNorm on load, statistics out, no slabs
NCONV_FRONT = True
NCONV_BLOCKS = -1  # bottlenecks from the stem that run this way; -1: layer 1 (3) up to
# 5 images, layers 1-2 (7) beyond (measured at 8 envs: 1.017 -> 0.997 ms per step; at 4 envs the slab chain wins layer 2)
NCONV_ROWS = 0  # output rows per workgroup (0: 64 pixels)
# first bottleneck (0..16) that runs in the chain; earlier ones (large feature maps: 16 partial slabs of a 32x32 map
# are more traffic than the launches they save) stay conv + GroupNorm pairs.  0 = the whole backbone incl. the stem;
# 3 = from layer2 on (measured best at 4 envs: 0.790 ms/step vs 0.820 from the stem and 0.996 without the chain);
# at 8 envs layer 2's slabs are twice as large and starting at layer 3 is better (1.046 vs 1.100 ms/step).
CHAIN_FROM_BLOCK = -1  # -1: by batch size (3 up to 5 images, 7 = from layer 3 beyond)
# first bottleneck whose GN2 -> conv3 -> GN3 tail -> next conv1 run as ONE launch (the block re-normalises the whole
# 16-64 KB conv2 output of its image): 2 launches per bottleneck instead of 3.  Measured SLOWER (0.835 vs 0.805 ms per
# step from layer 3 on, profiles/r02_gn_conv_ab.txt: the merged launch takes 20 us against 8.5 + 9.8), so 16 = never.
CHAIN_PAIR_FROM_BLOCK = 16


class GnConvDesc(C.Structure):
    """Mirror of `ivln_gn_conv_desc` (include/ivln_hip.h) - field order must match."""

    _fields_ = [
        ("x", vp), ("splits", i32), ("slab_stride", i64), ("gamma", vp), ("beta", vp),
        ("x2", vp), ("splits2", i32), ("slab_stride2", i64), ("gamma2", vp), ("beta2", vp),
        ("residual", vp),
        ("N", i32), ("C", i32), ("H", i32), ("W", i32), ("groups", i32), ("eps", f32), ("relu", i32), ("pool", i32),
        ("act_out", vp),
        ("wa", vp), ("Cout_a", i32), ("ka", i32), ("stride_a", i32), ("pad_a", i32), ("ya", vp),
        ("wb", vp), ("Cout_b", i32), ("stride_b", i32), ("yb", vp),
        ("x0", vp), ("splits0", i32), ("slab_stride0", i64), ("C0", i32), ("groups0", i32), ("gamma0", vp), ("beta0", vp),
        ("w0", vp),
    ]


def gn_conv(x, gn, relu=True, pool=False, x2=None, gn2=None, residual=None, want_act=False, conv_a=None, conv_b=None,
            front=None):
    """act(GroupNorm(x) [+ GroupNorm2(x2)] [+ residual]) [-> MaxPool(3, 2, 1)] and the NEXT convolution(s) of that
    activation in one launch (csrc/gn_conv.hip).  x, x2: `Deferred` slabs; conv_a = (weight (Co, C, k, k), stride, pad),
    conv_b = (weight (Co, C, 1, 1), stride).  Returns (act | None, Deferred a | None, Deferred b | None); the conv
    outputs are `groups` partial slabs for the next gn_conv / groupnorm.  None when the shape is outside the kernel's
    envelope.
    front = (x0 Deferred, gn0, w0 (C, C0, 1, 1)) with x None: two conv layers per launch - the block first normalises
    (+ ReLU) the WHOLE previous layer x0 of its image and runs the 1x1 conv w0 for its own group over the full K; that
    tile is what `gn` then normalises (Bottleneck: GN2 -> ReLU -> conv3 -> GN3 -> + identity -> ReLU -> next conv1)."""
    d = GnConvDesc()
    if front is not None:
        assert x is None
        x0, gn0, w0 = front
        N, Cc, H, W = x0.N, w0.shape[0], x0.H, x0.W
        dev = x0.ws.device
        assert w0.shape[1] == x0.C and w0.shape[2] == 1 and gn0.eps == gn.eps
        d.x0, d.splits0, d.slab_stride0 = dptr(x0.ws), x0.splits, x0.C * N * H * W
        d.C0, d.groups0, d.gamma0, d.beta0, d.w0 = x0.C, gn0.num_groups, dptr(gn0.weight), dptr(gn0.bias), dptr(w0)
    else:
        N, Cc, H, W = x.N, x.C, x.H, x.W
        dev = x.ws.device
        d.x, d.splits, d.slab_stride = dptr(x.ws), x.splits, Cc * N * H * W
    d.gamma, d.beta = dptr(gn.weight), dptr(gn.bias)
    if x2 is not None:
        assert (x2.N, x2.C, x2.H, x2.W) == (N, Cc, H, W) and gn2.num_groups == gn.num_groups and gn2.eps == gn.eps
        d.x2, d.splits2, d.slab_stride2 = dptr(x2.ws), x2.splits, Cc * N * H * W
        d.gamma2, d.beta2 = dptr(gn2.weight), dptr(gn2.bias)
    d.residual = _p(residual)
    d.N, d.C, d.H, d.W, d.groups, d.eps, d.relu, d.pool = N, Cc, H, W, gn.num_groups, gn.eps, int(bool(relu)), int(bool(pool))
    Hp, Wp = ((H - 1) // 2 + 1, (W - 1) // 2 + 1) if pool else (H, W)
    act = torch.empty((N, Cc, Hp, Wp), dtype=torch.float32, device=dev) if want_act else None
    d.act_out = _p(act)
    G = gn.num_groups
    ya = yb = None
    if conv_a is not None:
        w, s, p = conv_a
        Co, _, k, _ = w.shape
        Ho, Wo = (Hp + 2 * p - k) // s + 1, (Wp + 2 * p - k) // s + 1
        ya = Deferred(torch.empty(G * Co * N * Ho * Wo, dtype=torch.float32, device=dev), G, N, Co, Ho, Wo)
        d.wa, d.Cout_a, d.ka, d.stride_a, d.pad_a, d.ya = dptr(w), Co, k, s, p, dptr(ya.ws)
    if conv_b is not None:
        w, s = conv_b
        Co = w.shape[0]
        Ho, Wo = (Hp - 1) // s + 1, (Wp - 1) // s + 1
        yb = Deferred(torch.empty(G * Co * N * Ho * Wo, dtype=torch.float32, device=dev), G, N, Co, Ho, Wo)
        d.wb, d.Cout_b, d.stride_b, d.yb = dptr(w), Co, s, dptr(yb.ws)
    L = _L()
    L.ivln_gn_conv_f32.argtypes = [C.POINTER(GnConvDesc), vp]
    code = L.ivln_gn_conv_f32(C.byref(d), stream_ptr())
    if code == IVLN_E_UNSUPPORTED:
        return None
    check(code, "ivln_gn_conv_f32")
    return act, ya, yb


class NconvDesc(C.Structure):
    """Mirror of `ivln_nconv_desc` (include/ivln_hip.h) - field order must match."""

    _fields_ = [
        ("x", vp), ("stats", vp), ("parts", i32), ("gamma", vp), ("beta", vp),
        ("x2", vp), ("stats2", vp), ("parts2", i32), ("gamma2", vp), ("beta2", vp),
        ("residual", vp),
        ("N", i32), ("C", i32), ("H", i32), ("W", i32), ("groups", i32), ("eps", f32), ("relu", i32),
        ("act_out", vp),
        ("wa", vp), ("Cout_a", i32), ("ka", i32), ("groups_a", i32), ("ya", vp), ("stats_a", vp),
        ("wb", vp), ("Cout_b", i32), ("groups_b", i32), ("yb", vp), ("stats_b", vp),
        ("rows_per_block", i32), ("stride_a", i32), ("stride_b", i32),
    ]


class RawStats:
    """A conv output that has not been group-normalised yet, channel-major over the batch ([C][N][H][W]), + the (count,
    mean, M2) partials its producer left per (strip, image, group): what the next ivln_nconv_f32 launch normalises on
    load; `deferred()` views it as the one-slab `Deferred` the ivln_gn_conv_f32 chain takes."""

    def __init__(self, y, stats, parts, groups):
        self.y, self.stats, self.parts, self.groups = y, stats, parts, groups

    def deferred(self):
        Cc, N, H, W = self.y.shape
        return Deferred(self.y.view(-1), 1, N, Cc, H, W)


def nconv(x, gn=None, x2=None, gn2=None, residual=None, relu=True, want_act=False, conv_a=None, conv_b=None, rows_per_block=0):
    """Conv with GroupNorm on its input applied on load and the GroupNorm statistics of its output(s) emitted as partials
    (csrc/gn_conv.hip k_nconv): in = act(GN(x) [+ GN2(x2)] [+ residual]); conv_a = (weight (Co, C, k, k), groups of the
    GroupNorm that follows[, stride 1 | 2]) with k = 1 | 3, pad (k-1)/2; conv_b likewise (1x1; needs a stride-1 conv_a).  x: a `RawStats` with `gn`, or an
    activated NCHW tensor (gn None).  Returns (act | None, RawStats a, RawStats b | None), or None outside the envelope."""
    d = NconvDesc()
    xt = x.y if isinstance(x, RawStats) else x
    if isinstance(x, RawStats):
        Cc, N, H, W = xt.shape  # raw conv outputs are [C][N][H][W] (the one-slab layout of the deferred convs)
    else:
        N, Cc, H, W = xt.shape
    dev = xt.device
    d.x = _p(xt)
    if isinstance(x, RawStats):
        d.stats, d.parts, d.gamma, d.beta, d.groups, d.eps = _p(x.stats), x.parts, dptr(gn.weight), dptr(gn.bias), x.groups, gn.eps
        assert gn.num_groups == x.groups
    if x2 is not None:
        d.x2, d.stats2, d.parts2, d.gamma2, d.beta2 = _p(x2.y), _p(x2.stats), x2.parts, dptr(gn2.weight), dptr(gn2.bias)
        assert x2.groups == x.groups and tuple(x2.y.shape) == (Cc, N, H, W)
    d.residual = _p(residual)
    d.N, d.C, d.H, d.W, d.relu = N, Cc, H, W, int(bool(relu))
    act = torch.empty((N, Cc, H, W), dtype=torch.float32, device=dev) if want_act else None
    d.act_out = _p(act)
    def cw3(cw):  # (weight, groups of the following GroupNorm[, stride])
        return (cw[0], cw[1], cw[2] if len(cw) > 2 else 1)

    wa, ga, sa = cw3(conv_a)
    k = wa.shape[2]
    Ho, Wo = (H + 2 * (k // 2) - k) // sa + 1, (W + 2 * (k // 2) - k) // sa + 1
    sb = cw3(conv_b)[2] if conv_b is not None else 1
    rs = rows_per_block if rows_per_block > 0 else (NCONV_ROWS if NCONV_ROWS > 0 else (1 if Wo >= 64 else 64 // Wo))
    rs = min(rs, Ho)
    ya = torch.empty((wa.shape[0], N, Ho, Wo), dtype=torch.float32, device=dev)
    d.wa, d.Cout_a, d.ka, d.groups_a, d.ya = dptr(wa), wa.shape[0], k, ga, _p(ya)
    yb = None
    if conv_b is not None:
        wb, gb, _ = cw3(conv_b)
        yb = torch.empty((wb.shape[0], N, (H - 1) // sb + 1, (W - 1) // sb + 1), dtype=torch.float32, device=dev)
        d.wb, d.Cout_b, d.groups_b, d.yb = dptr(wb), wb.shape[0], gb, _p(yb)
    L = _L()
    L.ivln_nconv_f32.argtypes = [C.POINTER(NconvDesc), vp]
    while True:  # rows per workgroup: halved until the strip and the weight slice fit the workgroup
        if conv_b is not None and rs % sb:
            rs = (rs + sb - 1) // sb * sb
        strips = (Ho + rs - 1) // rs
        d.rows_per_block, d.stride_a, d.stride_b = rs, sa, sb
        sta = torch.empty((strips, N, ga, 3), dtype=torch.float32, device=dev)
        d.stats_a = _p(sta)
        stb = None
        if conv_b is not None:
            stb = torch.empty((strips, N, gb, 3), dtype=torch.float32, device=dev)
            d.stats_b = _p(stb)
        code = L.ivln_nconv_f32(C.byref(d), stream_ptr())
        if code != IVLN_E_UNSUPPORTED:
            break
        if rows_per_block > 0 or rs <= sb:
            return None
        rs //= 2
    check(code, "ivln_nconv_f32")
    return act, RawStats(ya, sta, strips, ga), (RawStats(yb, stb, strips, gb) if conv_b is not None else None)


class CmaStepDesc(C.Structure):
    """Mirror of `ivln_cma_step_desc` (include/ivln_hip.h) - field order must match."""

    _fields_ = [
        ("rows", i32), ("L", i32), ("P", i32), ("H", i32), ("Hq", i32), ("Ct", i32), ("d_out", i32), ("m_out", i32),
        ("E", i32), ("x2w", i32),
        ("state_in", vp), ("h_in", vp), ("ld_h", i64), ("mask", vp),
        ("w_ih1", vp), ("w_hh1", vp), ("b_ih1", vp), ("b_hh1", vp),
        ("Mq", vp), ("Mq_img", i64), ("lengths", vp), ("txt", vp), ("TQb", vp), ("TQb_img", i64),
        ("dkv", vp), ("mkv", vp), ("scale", f32),
        ("w_c", vp), ("b_c", vp), ("w_ih2", vp), ("w_hh2", vp), ("b_ih2", vp), ("b_hh2", vp),
        ("x2", vp), ("h_out", vp), ("ld_ho", i64), ("feats", vp), ("ws", vp),
    ]


# rollout head: 0 = ivln_cma_step_fwd (folded operands, five phase kernels), -1 = the unfused chain of ten separate ops
# (A/B switch IVLN_CMA_STEP_MODE; measured 0.977 vs 0.988 ms per 4-env step)
CMA_STEP_MODE = int(os.environ.get("IVLN_CMA_STEP_MODE", "0"))
_cma_ws = {}
CMA_WS_OWNER = 0  # graphed.GraphedRollout sets its own id while it warms up / captures: every runner owns a workspace


def cma_step_ws(rows, L, P, H, device):
    """Scratch of the fused head, one per (device, shape, OWNER): phases 2-5 of ivln_cma_step_fwd hand logits / tables /
    partial states to each other through it, so two heads in flight at once - a graph replay on one stream beside an
    eager act() on another, two runners of two policies - must not share it.  The owner is 0 for eager calls and the
    capturing GraphedRollout's id for a captured step (its pointer is baked into that graph).  Not keyed by stream:
    the buffer must not be born inside a stream capture (it would belong to that graph's private pool and outlive it
    in this cache) - the warm-up steps that precede every capture create it, on another stream than the capture."""
    L_ = _L()
    L_.ivln_cma_step_ws_floats.restype = i64
    L_.ivln_cma_step_ws_floats.argtypes = [i32, i32, i32, i32]
    n = L_.ivln_cma_step_ws_floats(rows, L, P, H)
    key = (str(device), rows, L, P, H, CMA_WS_OWNER)
    w = _cma_ws.get(key)
    if w is None:
        if torch.cuda.is_current_stream_capturing():
            raise _lib.IvlnError("cma_step workspace must exist before stream capture (run one warm-up step first)")
        w = torch.zeros(n, dtype=torch.float32, device=device)
        _cma_ws[key] = w
    return w


def release_cma_ws(owner):
    for k in [k for k in _cma_ws if k[-1] == owner]:
        del _cma_ws[k]


def cma_step(d: CmaStepDesc, mode=None):
    L_ = _L()
    L_.ivln_cma_step_fwd.argtypes = [C.POINTER(CmaStepDesc), i32, vp]
    check(L_.ivln_cma_step_fwd(C.byref(d), 0, stream_ptr()), "ivln_cma_step_fwd")


WEIGHT_EPOCH = 0  # bumped by FlatAdam.step(): kernels update parameters through raw pointers, which
#                   torch's tensor version counters do not see


def bn_fold(bn, scale, shift, conv_bias=None):
    check(
        _L().ivln_bn_fold_f32(dptr(bn.weight), dptr(bn.bias), dptr(bn.running_mean), dptr(bn.running_var),
                              _p(conv_bias), bn.eps, bn.num_features, dptr(scale), dptr(shift), stream_ptr()),
        "ivln_bn_fold_f32",
    )


_red_ws = {}


def _reduce_ws(device):
    """Scratch for the two-stage per-channel reductions (C * 64 splits * 3 values, per stream)."""
    key = (str(device), stream_ptr())
    w = _red_ws.get(key)
    if w is None:
        w = torch.empty(1 << 20, dtype=torch.float32, device=device)
        _red_ws[key] = w
    return w


def bn_train_stats(x, bn, scale, shift, save_mean=None, save_rstd=None, update_running=True):
    N, Cc, H, W = x.shape
    mom = bn.momentum if bn.momentum is not None else 0.1
    check(
        _L().ivln_bn_train_stats_f32(dptr(x), N, Cc, H * W, dptr(bn.weight), dptr(bn.bias),
                                     dptr(bn.running_mean) if update_running else None,
                                     dptr(bn.running_var) if update_running else None, mom, bn.eps, dptr(scale),
                                     dptr(shift), _p(save_mean), _p(save_rstd), dptr(_reduce_ws(x.device)),
                                     _reduce_ws(x.device).numel(), stream_ptr()),
        "ivln_bn_train_stats_f32",
    )


def scale_shift_relu_avgpool2(x, scale, shift, out=None):
    """x: NCHW tensor or a `Deferred` conv output (slab reduction fused)."""
    if isinstance(x, Deferred):
        N, Cc, H, W = x.N, x.C, x.H, x.W
        xp, img_s, chan_s, splits, slab = dptr(x.ws), H * W, N * H * W, x.splits, Cc * N * H * W
        dev = x.ws.device
    else:
        N, Cc, H, W = x.shape
        xp, img_s, chan_s, splits, slab = dptr(x), 0, 0, 1, 0
        dev = x.device
    if out is None:
        out = torch.empty((N, Cc, H // 2, W // 2), dtype=torch.float32, device=dev)
    check(_L().ivln_scale_shift_relu_avgpool2_f32(xp, dptr(scale), dptr(shift), dptr(out), N, Cc, H, W, img_s, chan_s,
                                                   splits, slab, stream_ptr()), "ivln_scale_shift_relu_avgpool2_f32")
    return out


def pool2d(x, k, s, p, mode, out=None):
    N, Cc, H, W = x.shape
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    if out is None:
        out = torch.empty((N, Cc, Ho, Wo), dtype=torch.float32, device=x.device)
    check(_L().ivln_pool2d_f32(dptr(x), dptr(out), N * Cc, H, W, k, s, p, 0 if mode == "max" else 1, stream_ptr()),
          "ivln_pool2d_f32")
    if _REC is not None:
        _rec(OP_POOL, (N * Cc, H, W, k, s, p, 0 if mode == "max" else 1), src0=x.data_ptr(), dst=out.data_ptr())
    return out


def map_features(occ_u8, sem_u8, classes=13, out=None):
    B, R, Cc = occ_u8.shape
    if out is None:
        out = torch.empty((B, classes + 1, R, Cc), dtype=torch.float32, device=occ_u8.device)
    check(_L().ivln_map_features_f32(dptr(occ_u8), dptr(sem_u8), dptr(out), B, R * Cc, classes, stream_ptr()),
          "ivln_map_features_f32")
    return out


def embed_lengths(tokens_i64, table):
    B, L = tokens_i64.shape
    V, E = table.shape
    emb = torch.empty((B * L, E), dtype=torch.float32, device=table.device)
    lengths = torch.empty((B,), dtype=torch.int32, device=table.device)
    check(_L().ivln_embed_lengths(dptr(tokens_i64), dptr(table), B, L, E, V, dptr(emb), dptr(lengths), stream_ptr()),
          "ivln_embed_lengths")
    return emb, lengths


def embed_gates(tokens_i64, table, row_nonzero, cache=None):
    """tokens (B, L) -> gx_f, gx_r (B*L, G) looked up in the folded (V, 2G) table, lengths i32 (B) (k_embed_gates).
    cache: an `InstructionStepCache` - its persistent gx / lengths buffers are the outputs and only the rows whose tokens
    differ from the cached ones are written (ivln_embed_gates_cached_f32; cache.dirty says which)."""
    B, L = tokens_i64.shape
    V, G2 = table.shape
    G = G2 // 2
    Lb = _L()
    Lb.ivln_embed_gates_cached_f32.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp]
    if cache is not None:
        gx_f, gx_r, lengths = cache.gx_f, cache.gx_r, cache.lengths
        check(Lb.ivln_embed_gates_cached_f32(dptr(tokens_i64), dptr(table), dptr(row_nonzero), B, L, G, V, dptr(gx_f), dptr(gx_r),
                                             dptr(lengths), dptr(cache.tokens), dptr(cache.dirty), stream_ptr()),
              "ivln_embed_gates_cached_f32")
        return gx_f, gx_r, lengths
    gx_f = torch.empty((B * L, G), dtype=torch.float32, device=table.device)
    gx_r = torch.empty((B * L, G), dtype=torch.float32, device=table.device)
    lengths = torch.empty((B,), dtype=torch.int32, device=table.device)
    check(Lb.ivln_embed_gates_cached_f32(dptr(tokens_i64), dptr(table), dptr(row_nonzero), B, L, G, V, dptr(gx_f), dptr(gx_r),
                                         dptr(lengths), None, None, stream_ptr()), "ivln_embed_gates_f32")
    return gx_f, gx_r, lengths


class InstructionStepCache:
    """Persistent device buffers of the per-episode instruction cache for one (rows, L) batch shape (rollout steps only:
    map_cma_policy.py:293 re-encodes an episode's instruction at every step): the tokens every row encoded last, the per-row
    dirty flags of the current step, and everything the instruction branch produces - gate inputs, lengths, the bi-LSTM's
    output, the folded attention operands (`fold`, allocated by the policy on first use).  Allocated OUTSIDE any stream
    capture and shared by every graph of the policy, so that what one replayed step leaves is what the next one finds."""

    def __init__(self, rows, L, G, H, device, key):
        self.rows, self.L, self.key = rows, L, key
        self.tokens = torch.full((rows, L), -1, dtype=torch.int64, device=device)  # (-1: no row matches - everything dirty)
        self.dirty = torch.ones((rows,), dtype=torch.int32, device=device)
        self.gx_f = torch.zeros((rows * L, G), dtype=torch.float32, device=device)
        self.gx_r = torch.zeros((rows * L, G), dtype=torch.float32, device=device)
        self.lengths = torch.zeros((rows,), dtype=torch.int32, device=device)
        self.out = torch.zeros((rows, 2 * H, L), dtype=torch.float32, device=device)
        self.fold = None
        self.fold_key = None
        _STEP_CACHES.add(self)

    def invalidate(self):
        """Every row re-encodes at the next step (weights changed).  An eager fill on the current stream."""
        self.tokens.fill_(-1)


import weakref as _weakref  # noqa: E402

_STEP_CACHES = _weakref.WeakSet()
CACHE_INSTRUCTION = os.environ.get("IVLN_CACHE_INSTRUCTION", "1") != "0"  # A/B switch: 0 = re-encode at every step


def invalidate_step_caches():
    """Called when parameters changed behind torch's back (FlatAdam.step writes through raw pointers): a replayed graph
    would otherwise keep serving instruction encodings of the old weights."""
    for c in list(_STEP_CACHES):
        c.invalidate()


def lstm_bidir(gx_f, gx_r, whh_f, whh_r, bhh_f, bhh_r, lengths, B, L, H, save=False, spare=1, ticket=None, cache=None):
    """spare > 1: ivln_lstm_bidir_fwd_spread_f32 - 2B * spare blocks draw the 2B items in the order they start (for a
    replay beside a launch that fills some XCDs).  ticket: the caller's zeroed int32 word (one launch in flight per word).
    cache: the step cache whose `dirty` flags `embed_gates` just wrote - rows with dirty == 0 are not run and cache.out
    (the returned tensor) keeps their values."""
    if cache is not None and not save:
        tk = ticket if spare > 1 else None
        if spare > 1 and (tk is None or tk.dtype != torch.int32 or tk.device != gx_f.device):
            raise ValueError("lstm_bidir(spare>1) needs the caller's int32 ticket word on the same device")
        Lb = _L()
        Lb.ivln_lstm_bidir_fwd_cached_f32.argtypes = [vp] * 7 + [i32, i32, i32, vp, vp, vp, vp, i32, vp, vp]
        check(Lb.ivln_lstm_bidir_fwd_cached_f32(dptr(gx_f), dptr(gx_r), dptr(whh_f), dptr(whh_r), dptr(bhh_f), dptr(bhh_r),
                                                dptr(lengths), B, L, H, dptr(cache.out), None, None, _p(tk),
                                                int(spare) if tk is not None else 1, dptr(cache.dirty), stream_ptr()),
              "ivln_lstm_bidir_fwd_cached_f32")
        return cache.out, None, None
    out = torch.empty((B, 2 * H, L), dtype=torch.float32, device=gx_f.device)
    if spare > 1 and not save:
        tk = ticket
        if tk is None or tk.dtype != torch.int32 or tk.device != gx_f.device:
            raise ValueError("lstm_bidir(spare>1) needs the caller's int32 ticket word on the same device")
        Lb = _L()
        Lb.ivln_lstm_bidir_fwd_spread_f32.argtypes = [vp] * 7 + [i32, i32, i32, vp, vp, vp, vp, i32, vp]
        check(Lb.ivln_lstm_bidir_fwd_spread_f32(dptr(gx_f), dptr(gx_r), dptr(whh_f), dptr(whh_r), dptr(bhh_f), dptr(bhh_r),
                                                dptr(lengths), B, L, H, dptr(out), None, None, dptr(tk), int(spare),
                                                stream_ptr()), "ivln_lstm_bidir_fwd_spread_f32")
        return out, None, None
    gates = cs = None
    if save:
        gates = torch.zeros((B, 2, L, 4 * H), dtype=torch.float32, device=gx_f.device)
        cs = torch.zeros((B, 2, L, H), dtype=torch.float32, device=gx_f.device)
    check(
        _L().ivln_lstm_bidir_fwd_f32(dptr(gx_f), dptr(gx_r), dptr(whh_f), dptr(whh_r), dptr(bhh_f), dptr(bhh_r),
                                     dptr(lengths), B, L, H, dptr(out), _p(gates), _p(cs), stream_ptr()),
        "ivln_lstm_bidir_fwd_f32",
    )
    return out, gates, cs


def attn_small2(q, k0, v0, out0, k1, v1, out1, scale):
    """Two attentions over a short key axis (<= 32 positions) sharing the query, one launch."""
    rows = q.shape[0]
    I = v0.shape[2]
    L = _L()
    L.ivln_attn_small2_f32.argtypes = [vp, i64, f32, i32, i32, vp, i64, vp, i64, i32, i32, vp, i64, vp, i64, vp, i64,
                                       i32, i32, vp, i64, vp]
    check(
        L.ivln_attn_small2_f32(_p(q), q.stride(0), scale, rows, I, _p(k0), k0.stride(0), _p(v0), v0.stride(0),
                               k0.shape[1], v0.shape[1], _p(out0), out0.stride(0), _p(k1), k1.stride(0), _p(v1),
                               v1.stride(0), k1.shape[1], v1.shape[1], _p(out1), out1.stride(0), stream_ptr()),
        "ivln_attn_small2_f32",
    )


def gru_step(x, gi_pre, h_in, mask_u8, w_ih, w_hh, b_ih, b_hh, h_out, h_out2=None, saves=None):
    """One masked GRU step over `rows` rows; x (rows,I) or gi_pre (rows,3H); h_in/h_out strided."""
    rows = h_in.shape[0]
    H = w_hh.shape[1]
    s = saves or (None, None, None, None)
    check(
        _L().ivln_gru_step_f32(
            _p(x), x.stride(0) if x is not None else 0, w_ih.shape[1], _p(gi_pre),
            gi_pre.stride(0) if gi_pre is not None else 0, _p(h_in), h_in.stride(0), _p(mask_u8), dptr(w_ih),
            dptr(w_hh), dptr(b_ih), dptr(b_hh), _p(h_out), h_out.stride(0), _p(h_out2),
            h_out2.stride(0) if h_out2 is not None else 0, rows, H, _p(s[0]), _p(s[1]), _p(s[2]), _p(s[3]),
            stream_ptr(),
        ),
        "ivln_gru_step_f32",
    )


SEQ_PERSISTENT = os.environ.get("IVLN_SEQ_PERSISTENT", "1") != "0"  # A/B: 0 = one launch per timestep
_seq_sync_ws = {}


def _seq_ws(device):
    """256-byte counter / error workspace of the single-launch sequence GRU (csrc/gru_seq.hip), one per stream: two
    sequences in flight on different streams must not share a counter.  None -> the per-timestep launches."""
    if not SEQ_PERSISTENT:
        return None
    key = (str(device), stream_ptr())
    ws = _seq_sync_ws.get(key)
    if ws is None:
        ws = _seq_sync_ws[key] = torch.zeros(64, dtype=torch.int32, device=device)
    return ws


def seq_guard_word(device):
    """The sticky error word (a one-element int32 view) of the CURRENT stream's sequence-GRU workspace, or None: what
    FlatAdam hands the Adam kernel as its guard, so that an update whose persistent GRU timed out leaves the parameters
    alone."""
    ws = _seq_sync_ws.get((str(device), stream_ptr())) if SEQ_PERSISTENT else None
    return None if ws is None else ws[48:49]


def seq_failed():
    """A persistent sequence launch timed out (reads each workspace's sticky word: a 4-byte copy per workspace - call
    where the stream is idle anyway, e.g. right after the update's loss read-back)."""
    return any(int(ws[48]) != 0 for ws in _seq_sync_ws.values())


def seq_recover():
    """After `seq_failed`: clear the workspaces and run the sequence GRUs as per-timestep launches for the rest of the run."""
    global SEQ_PERSISTENT
    import logging

    L = _L()
    L.ivln_seq_sync_init.argtypes = [vp, vp]
    for ws in _seq_sync_ws.values():
        check(L.ivln_seq_sync_init(dptr(ws), stream_ptr()), "ivln_seq_sync_init")
    torch.cuda.synchronize()
    SEQ_PERSISTENT = False
    logging.getLogger("ivln_ce_amd").warning(
        "persistent sequence GRU: a bounded spin timed out (its workgroups were not all resident); the update is computed again "
        "with per-timestep launches, which stay on for the rest of the run")


def check_seq_sync():
    """Raise if a bounded spin of a persistent sequence launch timed out (synchronises the stream: call where the
    host waits anyway, e.g. after reading the loss).  A timed-out launch ends on its own; its outputs are garbage."""
    L = _L()
    L.ivln_seq_sync_status.argtypes = [vp, vp]
    for ws in _seq_sync_ws.values():
        check(L.ivln_seq_sync_status(dptr(ws), stream_ptr()), "ivln_seq_sync_status (persistent GRU spin timed out)")


_seq_polls = {}


def seq_sync_poll():
    """Non-blocking form of check_seq_sync: raises if an EARLIER poll's read of a workspace's sticky error word has
    arrived non-zero, then queues a fresh asynchronous read of every workspace on the current stream.  A timed-out
    persistent launch is therefore reported at the latest one call late, without the host ever waiting for the GPU."""
    for key, ws in _seq_sync_ws.items():
        slot = _seq_polls.get(key)
        if slot is None:
            slot = _seq_polls[key] = [torch.zeros(1, dtype=torch.int32).pin_memory(), torch.cuda.Event(), False]
        pin, ev, pending = slot
        if pending and ev.query():
            slot[2] = False
            if int(pin[0]) != 0:
                raise _lib.IvlnError("persistent sequence GRU: a bounded spin timed out (ivln_seq_sync_status)")
        if not slot[2]:
            pin.copy_(ws[48:49], non_blocking=True)
            ev.record()
            slot[2] = True


def gru_seq(gi, h0, masks_u8, w_hh, b_hh, out, state_out, T, N, saves=None):
    """Masked GRU over T timesteps of N rows in one C-ABI call: ONE persistent launch inside the kernel's envelope
    (H = 512, N <= 64), else T dependent launches enqueued from C."""
    H = w_hh.shape[1]
    sv = saves or (None, None, None, None)
    L = _L()
    L.ivln_cma_seq_fwd_f32.argtypes = [vp, vp, i64, vp, vp, vp, vp, i64, vp, i64, i32, i32, i32, vp, vp, vp, vp, vp, vp]
    check(L.ivln_cma_seq_fwd_f32(dptr(gi), _p(h0), h0.stride(0), dptr(masks_u8), dptr(w_hh), dptr(b_hh), _p(out),
                                 out.stride(0), _p(state_out), state_out.stride(0) if state_out is not None else 0, T, N,
                                 H, _p(sv[0]), _p(sv[1]), _p(sv[2]), _p(sv[3]), _p(_seq_ws(gi.device)), stream_ptr()),
          "ivln_cma_seq_fwd_f32")


def gru_seq_bwd(d_out, r, z, n, ghn, out, h0, masks_u8, whh_t, T, N, dgi, dgh, hp, dhz):
    H = r.shape[1]
    L = _T()
    L.ivln_cma_seq_bwd_f32.argtypes = [vp, i64, vp, vp, vp, vp, vp, i64, vp, i64, vp, vp, i32, i32, i32, vp, vp, vp, vp,
                                       vp, vp]
    check(L.ivln_cma_seq_bwd_f32(_p(d_out), d_out.stride(0), dptr(r), dptr(z), dptr(n), dptr(ghn), _p(out), out.stride(0),
                                 _p(h0), h0.stride(0), dptr(masks_u8), dptr(whh_t), T, N, H, dptr(dgi), dptr(dgh), dptr(hp),
                                 dptr(dhz), _p(_seq_ws(r.device)), stream_ptr()), "ivln_cma_seq_bwd_f32")


def attn(q, k, v, valid_len, scale, out, save_attn=None, row_index=None):
    """q (rows,Ck) strided rows; k (imgs,Ck,I), v (imgs,Cv,I) with image strides; out (rows,Cv) strided.
    row_index (rows,) i32: the key/value image of each row (None: row r uses image r); valid_len is per image."""
    rows, Ck = q.shape
    Cv, I = v.shape[1], v.shape[2]
    logits_ws = torch.empty((rows, I), dtype=torch.float32, device=q.device)
    L = _L()
    L.ivln_attn_fwd_idx_f32.argtypes = [vp, i64, vp, i64, vp, i64, vp, f32, i32, i32, i32, i32, vp, i64, vp, vp, vp, vp]
    check(
        L.ivln_attn_fwd_idx_f32(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(valid_len), scale, rows,
                                Ck, Cv, I, _p(out), out.stride(0), _p(save_attn), dptr(logits_ws), _p(row_index),
                                stream_ptr()),
        "ivln_attn_fwd_idx_f32",
    )
    return out


def prev_action_embed(prev_actions_i64, mask_u8, table, out1, out2=None):
    rows = prev_actions_i64.numel()
    n_emb, E = table.shape
    check(
        _L().ivln_prev_action_embed_f32(dptr(prev_actions_i64), dptr(mask_u8), dptr(table), rows, E, n_emb, _p(out1),
                                        out1.stride(0), _p(out2), out2.stride(0) if out2 is not None else 0,
                                        stream_ptr()),
        "ivln_prev_action_embed_f32",
    )


def linear_argmax(x, w, bias, out=None):
    """argmax_o (x . w[o] + b[o]) per row, one launch (deterministic action head); out (rows,1) int64."""
    rows, K = x.shape
    if out is None:
        out = torch.empty((rows, 1), dtype=torch.int64, device=x.device)
    L = _L()
    L.ivln_linear_argmax_f32.argtypes = [vp, i64, vp, vp, i32, i32, i32, vp, vp, vp]
    check(L.ivln_linear_argmax_f32(_p(x), x.stride(0), dptr(w), _p(bias), rows, K, w.shape[0], dptr(out), None,
                                   stream_ptr()), "ivln_linear_argmax_f32")
    return out


def linear_sample(x, w, bias, u_sample, u_beta=None, beta=0.0, expert=None, out=None, logits_out=None):
    """Sampled action head in one launch: inverse-CDF draw from softmax(x . w^T + b) with the uniforms `u_sample`
    (rows,) f32, beta-mixed with `expert` (rows,) f64 through `u_beta` (rows,) f32 and zeroed where expert == -1
    (ivln_linear_sample_f32); out (rows, 1) int64."""
    rows, K = x.shape
    if out is None:
        out = torch.empty((rows, 1), dtype=torch.int64, device=x.device)
    L = _L()
    L.ivln_linear_sample_f32.argtypes = [vp, i64, vp, vp, i32, i32, i32, vp, vp, f32, vp, vp, vp, vp]
    check(L.ivln_linear_sample_f32(_p(x), x.stride(0), dptr(w), _p(bias), rows, K, w.shape[0], dptr(u_sample),
                                   _p(u_beta), float(beta), _p(expert), dptr(out), _p(logits_out), stream_ptr()),
          "ivln_linear_sample_f32")
    return out


def rednet_fwd(table, n_ops, rgb_u8, depth, labels_out):
    """One C call walks a recorded op table (ivln_rednet_fwd): RedNet's whole forward for this step's frames."""
    L = _L()
    L.ivln_rednet_fwd.argtypes = [vp, i32, vp, vp, vp, vp]
    check(L.ivln_rednet_fwd(C.addressof(table), n_ops, _dptr(rgb_u8), _dptr(depth), _dptr(labels_out), stream_ptr()),
          "ivln_rednet_fwd")


def argmax_rows(x, out=None):
    rows, Cc = x.shape
    if out is None:
        out = torch.empty((rows, 1), dtype=torch.int64, device=x.device)
    check(_L().ivln_argmax_rows(dptr(x), rows, Cc, dptr(out), stream_ptr()), "ivln_argmax_rows")
    return out


def argmax_channels_u8(x):
    N, Cc, H, W = x.shape
    out = torch.empty((N, 1, H, W), dtype=torch.uint8, device=x.device)
    check(_L().ivln_argmax_channels_u8(dptr(x), N, Cc, H * W, dptr(out), stream_ptr()), "ivln_argmax_channels_u8")
    if _REC is not None:  # dst NULL: the labels go where the replaying call says
        _rec(OP_ARGMAX_U8, (N, Cc, H * W), src0=x.data_ptr(), dst=None)
    return out


def rgb_resize_normalize(rgb_u8_nhwc, Ho, Wo):
    B, Hi, Wi, _ = rgb_u8_nhwc.shape
    out = torch.empty((B, 3, Ho, Wo), dtype=torch.float32, device=rgb_u8_nhwc.device)
    check(_L().ivln_rgb_resize_normalize_f32(dptr(rgb_u8_nhwc), B, Hi, Wi, Ho, Wo, dptr(out), stream_ptr()),
          "ivln_rgb_resize_normalize_f32")
    if _REC is not None:  # src NULL: this step's frames come with the replaying call
        _rec(OP_RGB_NORM, (B, Hi, Wi, Ho, Wo), src0=None, dst=out.data_ptr())
    return out


def rgb_to_nchw(rgb_u8_nhwc, div=255.0):
    B, H, W, _ = rgb_u8_nhwc.shape
    out = torch.empty((B, 3, H, W), dtype=torch.float32, device=rgb_u8_nhwc.device)
    L = _L()
    L.ivln_rgb_to_nchw_f32.argtypes = [vp, i32, i32, i32, f32, vp, vp]
    check(L.ivln_rgb_to_nchw_f32(dptr(rgb_u8_nhwc), B, H, W, div, dptr(out), stream_ptr()), "ivln_rgb_to_nchw_f32")
    return out


def adaptive_avgpool2d(x, OH, OW, out=None, out_ctot=0):
    N, Cc, H, W = x.shape
    if out is None:
        out = torch.empty((N, Cc, OH, OW), dtype=torch.float32, device=x.device)
    L = _L()
    L.ivln_adaptive_avgpool2d_f32.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, i64, vp]
    check(L.ivln_adaptive_avgpool2d_f32(dptr(x), N, Cc, H, W, OH, OW, _p(out), out_ctot * OH * OW, stream_ptr()),
          "ivln_adaptive_avgpool2d_f32")
    return out


def affine(x, sub, div):
    out = torch.empty_like(x)
    check(_L().ivln_affine_f32(dptr(x), dptr(out), x.numel(), sub, div, stream_ptr()), "ivln_affine_f32")
    if _REC is not None:  # src NULL: this step's depth comes with the replaying call
        _rec(OP_AFFINE, floats=(sub, div), n=x.numel(), src0=None, dst=out.data_ptr())
    return out


def add(a, b, relu=False, out=None):
    if out is None:
        out = torch.empty_like(a)
    check(_L().ivln_add_f32(dptr(a), dptr(b), dptr(out), a.numel(), int(bool(relu)), stream_ptr()), "ivln_add_f32")
    if _REC is not None:
        _rec(OP_ADD, (int(bool(relu)),), n=a.numel(), src0=a.data_ptr(), src1=b.data_ptr(), dst=out.data_ptr())
    return out


def copy_multi(pairs):
    """[(src, dst)] contiguous same-shape/dtype device tensors copied by one launch (<= 8 per call)."""
    L = _L()
    L.ivln_copy_multi.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int,
                                  C.c_void_p]
    for k in range(0, len(pairs), 8):
        chunk = pairs[k:k + 8]
        n = len(chunk)
        srcs, dsts, nb = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_int64 * n)()
        for i, (s, d) in enumerate(chunk):
            if s.shape != d.shape or s.dtype != d.dtype or not s.is_contiguous() or not d.is_contiguous():
                raise _lib.IvlnError("copy_multi needs contiguous tensors of identical shape and dtype")
            srcs[i], dsts[i], nb[i] = _p(s), _p(d), s.numel() * s.element_size()
        check(L.ivln_copy_multi(srcs, dsts, nb, n, stream_ptr()), "ivln_copy_multi")


def add_multi(pairs):
    """[(src, dst)]: dst += src for contiguous float32 device tensors of equal size, 64 per launch."""
    L = _L()
    L.ivln_add_multi_f32.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int,
                                     C.c_void_p]
    for k in range(0, len(pairs), 64):
        chunk = pairs[k:k + 64]
        n = len(chunk)
        srcs, dsts, cnt = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_int64 * n)()
        for i, (s, d) in enumerate(chunk):
            if (s.numel() != d.numel() or s.dtype != torch.float32 or d.dtype != torch.float32
                    or not s.is_contiguous() or not d.is_contiguous()):
                raise _lib.IvlnError("add_multi needs contiguous float32 tensors of identical size")
            srcs[i], dsts[i], cnt[i] = _p(s), _p(d), s.numel()
        check(L.ivln_add_multi_f32(srcs, dsts, cnt, n, stream_ptr()), "ivln_add_multi_f32")


def tour_memory(mem, h, mask_u8, out1, out2=None):
    """out = mask * max(mem, h) (h optional), row-strided (N,H) views; see ivln_tour_memory_f32."""
    N, H = mem.shape
    check(
        _L().ivln_tour_memory_f32(_p(mem), mem.stride(0), _p(h) if h is not None else None, h.stride(0) if h is not None else 0,
                                  _p(mask_u8) if mask_u8 is not None else None, N, H, _p(out1), out1.stride(0),
                                  _p(out2) if out2 is not None else None, out2.stride(0) if out2 is not None else 0,
                                  stream_ptr()),
        "ivln_tour_memory_f32",
    )
    return out1


def copy2d(src, dst, rows, cols, broadcast_rows=False):
    check(
        _L().ivln_copy2d_f32(_p(src), src.stride(0) if src.dim() > 1 else cols, _p(dst), dst.stride(0), rows, cols,
                             int(bool(broadcast_rows)), stream_ptr()),
        "ivln_copy2d_f32",
    )


# ---- backward / loss / optimizer bindings (csrc/train_ops.hip) -------------------------------------
_tsigs_done = False


def _T():
    global _tsigs_done
    L = _L()
    if not _tsigs_done:
        L.ivln_relu_bwd_f32.argtypes = [vp, vp, vp, i32, i32, i64, i64, i64, vp]
        L.ivln_add2d_f32.argtypes = [vp, i64, vp, i64, vp, i64, i32, i32, vp]
        L.ivln_colsum_f32.argtypes = [vp, i64, i32, i32, vp, i32, vp, i64, vp]
        L.ivln_nchw_chansum_f32.argtypes = [vp, i32, i32, i32, vp, vp, i64, vp]
        L.ivln_transpose_f32.argtypes = [vp, vp, i32, i32, vp]
        L.ivln_weight_flip_transpose_f32.argtypes = [vp, vp, i32, i32, i32, i32, vp]
        L.ivln_attn_bwd_f32.argtypes = [vp, i64, vp, vp, i64, vp, i64, vp, i64, f32, i32, i32, i32, i32, vp, i64, vp,
                                        i64, vp, i64, vp]
        L.ivln_gru_bwd_elem_f32.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, i64, vp, i32, i32, vp, vp, vp, vp, vp]
        L.ivln_linear_skinny_ex_f32.argtypes = [vp, i64, vp, vp, i64, vp, vp, i64, i32, i32, i32, vp]
        L.ivln_lstm_bidir_bwd_f32.argtypes = [vp] * 7 + [i32, i32, i32, vp, vp, vp, vp, vp]
        L.ivln_cbra_bwd_f32.argtypes = [vp] * 6 + [i32, i32, i32, i32, i32, vp, vp, vp, vp, i64, vp]
        L.ivln_embedding_scatter_add_f32.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp]
        L.ivln_prev_action_embed_bwd_f32.argtypes = [vp, vp, vp, i64, vp, i64, i32, i32, i32, vp, vp]
        L.ivln_ce_iw_loss_f32.argtypes = [vp, vp, vp, i32, i32, i32, f32, vp, vp, vp]
        L.ivln_pm_loss_fwd_f32.argtypes = [vp, vp, i32, vp, vp, vp]
        L.ivln_pm_loss_bwd_f32.argtypes = [vp, vp, vp, i32, vp, vp]
        L.ivln_adam_step_f32.argtypes = [vp, vp, vp, vp, i64, f32, vp, vp, f32, f32, f32, i32, f32, i32, vp]
        _tsigs_done = True
    return L


def relu_bwd(dy, y, dx=None):
    """dx = dy * (y > 0); 2-D row-strided views allowed."""
    rows, cols = y.shape
    if dx is None:
        dx = torch.empty((rows, cols), dtype=torch.float32, device=y.device)
    check(_T().ivln_relu_bwd_f32(_p(dy), _p(y), _p(dx), rows, cols, dy.stride(0), y.stride(0), dx.stride(0),
                                 stream_ptr()), "ivln_relu_bwd_f32")
    return dx


def add2d(a, b, out=None):
    rows, cols = a.shape
    if out is None:
        out = torch.empty((rows, cols), dtype=torch.float32, device=a.device)
    check(_T().ivln_add2d_f32(_p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), rows, cols,
                              stream_ptr()), "ivln_add2d_f32")
    return out


_colsum_ws = {}


def colsum(x, out=None, accumulate=False):
    """Column sums of a 2-D (row-strided) matrix -> (cols,)."""
    rows, cols = x.shape
    if out is None:
        out = torch.empty((cols,), dtype=torch.float32, device=x.device)
    key = (str(x.device), stream_ptr())  # per stream: the training pass runs two at once
    ws = _colsum_ws.get(key)
    if ws is None or ws.numel() < 128 * cols:
        ws = torch.empty(max(128 * cols, 1 << 18), dtype=torch.float32, device=x.device)
        _colsum_ws[key] = ws
    check(_T().ivln_colsum_f32(_p(x), x.stride(0), rows, cols, dptr(out), int(accumulate), dptr(ws), ws.numel(),
                               stream_ptr()), "ivln_colsum_f32")
    return out


class ColsumQueue:
    """Column sums whose results are only needed at the end of a backward pass (the bias gradients): `add(x)` hands out
    the result tensor at once and keeps `x` alive, `flush()` computes all of them in two launches
    (ivln_colsum_multi_f32) instead of two per matrix."""

    def __init__(self):
        self.jobs = []

    def add(self, x):
        rows, cols = x.shape
        out = torch.empty((cols,), dtype=torch.float32, device=x.device)
        if not COLSUM_MULTI or x.stride(1) != 1:
            return colsum(x, out)
        self.jobs.append((x, out))
        if len(self.jobs) == 32:
            self.flush()
        return out

    def flush(self):
        jobs, self.jobs = self.jobs, []
        if not jobs:
            return
        n = len(jobs)
        dev = jobs[0][0].device
        cur = torch.cuda.current_stream(dev)
        for x, o in jobs:  # (queued on the instruction branch's side stream, summed on this one)
            x.record_stream(cur)
            o.record_stream(cur)
        need = sum(min(128, (x.shape[0] + 255) // 256) * x.shape[1] for x, _ in jobs)
        key = (str(dev), stream_ptr())
        ws = _colsum_ws.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.empty(max(need, 1 << 18), dtype=torch.float32, device=dev)
            _colsum_ws[key] = ws
        xs = (vp * n)(*[x.data_ptr() for x, _ in jobs])
        outs = (vp * n)(*[o.data_ptr() for _, o in jobs])
        lds = (i64 * n)(*[x.stride(0) for x, _ in jobs])
        rows = (i32 * n)(*[x.shape[0] for x, _ in jobs])
        cols = (i32 * n)(*[x.shape[1] for x, _ in jobs])
        L = _T()
        L.ivln_colsum_multi_f32.argtypes = [C.POINTER(vp), C.POINTER(i64), C.POINTER(i32), C.POINTER(i32), C.POINTER(vp), i32,
                                            vp, i64, vp]
        check(L.ivln_colsum_multi_f32(xs, lds, rows, cols, outs, n, dptr(ws), ws.numel(), stream_ptr()),
              "ivln_colsum_multi_f32")


COLSUM_MULTI = True  # one pair of launches for all bias gradients


def nchw_chansum(x):
    N, Cc, H, W = x.shape
    out = torch.empty((Cc,), dtype=torch.float32, device=x.device)
    ws = _reduce_ws(x.device)
    check(_T().ivln_nchw_chansum_f32(dptr(x), N, Cc, H * W, dptr(out), dptr(ws), ws.numel(), stream_ptr()),
          "ivln_nchw_chansum_f32")
    return out


def transpose(x):
    R, Cc = x.shape
    y = torch.empty((Cc, R), dtype=torch.float32, device=x.device)
    check(_T().ivln_transpose_f32(dptr(x), dptr(y), R, Cc, stream_ptr()), "ivln_transpose_f32")
    return y


def weight_flip_transpose(w):
    O, I, KH, KW = w.shape
    wt = torch.empty((I, O, KH, KW), dtype=torch.float32, device=w.device)
    check(_T().ivln_weight_flip_transpose_f32(dptr(w), dptr(wt), O, I, KH, KW, stream_ptr()),
          "ivln_weight_flip_transpose_f32")
    return wt


def attn_bwd(dout, attn_p, q, k, v, scale, dq, dk, dv, row_index=None):
    """dk / dv are per ROW (rows, C, I) even when rows share key/value images through row_index (index_sum folds them)."""
    rows, Ck = q.shape
    Cv, I = v.shape[1], v.shape[2]
    L = _T()
    L.ivln_attn_bwd_idx_f32.argtypes = [vp, i64, vp, vp, i64, vp, i64, vp, i64, f32, i32, i32, i32, i32, vp, i64, vp,
                                        i64, vp, i64, vp, vp]
    check(
        L.ivln_attn_bwd_idx_f32(_p(dout), dout.stride(0), dptr(attn_p), _p(q), q.stride(0), _p(k), k.stride(0), _p(v),
                                v.stride(0), scale, rows, Ck, Cv, I, _p(dq), dq.stride(0), _p(dk), dk.stride(0),
                                _p(dv), dv.stride(0), _p(row_index), stream_ptr()),
        "ivln_attn_bwd_idx_f32",
    )


def index_sum(src, index, U):
    """dst[u] = sum of the rows of `src` (rows, ...) whose index is u, in ascending row order -> (U, ...)."""
    rows = src.shape[0]
    M = src[0].numel()
    dst = torch.empty((U,) + tuple(src.shape[1:]), dtype=torch.float32, device=src.device)
    L = _T()
    L.ivln_index_sum_f32.argtypes = [vp, vp, i32, i64, i32, vp, vp]
    check(L.ivln_index_sum_f32(dptr(src), dptr(index), rows, M, U, dptr(dst), stream_ptr()), "ivln_index_sum_f32")
    return dst


def gru_bwd_elem(dout, dh_carry, r, z, n, ghn, h_prev, mask, dgi, dgh, dhz, hp_out):
    rows, H = r.shape
    check(
        _T().ivln_gru_bwd_elem_f32(_p(dout), dout.stride(0), _p(dh_carry), _p(r), _p(z), _p(n), _p(ghn), _p(h_prev),
                                   h_prev.stride(0), _p(mask), rows, H, _p(dgi), _p(dgh), _p(dhz), _p(hp_out),
                                   stream_ptr()),
        "ivln_gru_bwd_elem_f32",
    )


def gru_bwd_step(dgh_t, whh_t, mask_t, dout_prev, r, z, n, ghn, h_prev, mask_prev, dhz, dgi_prev, dgh_prev, hp_prev):
    """Fused BPTT step: carry of step t (matvec + dhz, masked) -> element part of step t-1."""
    rows, H = r.shape
    L = _T()
    L.ivln_gru_bwd_step_f32.argtypes = [vp, i64, vp, vp, vp, i64, vp, vp, vp, vp, vp, i64, vp, i32, i32, vp, vp, vp, vp,
                                        vp]
    check(
        L.ivln_gru_bwd_step_f32(_p(dgh_t), dgh_t.stride(0), dptr(whh_t), _p(mask_t), _p(dout_prev), dout_prev.stride(0),
                                _p(r), _p(z), _p(n), _p(ghn), _p(h_prev), h_prev.stride(0), _p(mask_prev), rows, H,
                                _p(dhz), _p(dgi_prev), _p(dgh_prev), _p(hp_prev), stream_ptr()),
        "ivln_gru_bwd_step_f32",
    )


def linear_skinny_ex(x, W, add, rowmask, out):
    rows, K = x.shape
    O = W.shape[0]
    check(
        _T().ivln_linear_skinny_ex_f32(_p(x), x.stride(0), dptr(W), _p(add), add.stride(0) if add is not None else 0,
                                       _p(rowmask), _p(out), out.stride(0), rows, K, O, stream_ptr()),
        "ivln_linear_skinny_ex_f32",
    )
    return out


def lstm_bidir_bwd(dout, out, gates, cs, whh_f, whh_r, lengths, B, L, H):
    dev = dout.device
    dgx_f = torch.empty((B * L, 4 * H), dtype=torch.float32, device=dev)
    dgx_r = torch.empty((B * L, 4 * H), dtype=torch.float32, device=dev)
    hp_f = torch.empty((B * L, H), dtype=torch.float32, device=dev)
    hp_r = torch.empty((B * L, H), dtype=torch.float32, device=dev)
    check(
        _T().ivln_lstm_bidir_bwd_f32(dptr(dout), dptr(out), dptr(gates), dptr(cs), dptr(whh_f), dptr(whh_r),
                                     dptr(lengths), B, L, H, dptr(dgx_f), dptr(dgx_r), dptr(hp_f), dptr(hp_r),
                                     stream_ptr()),
        "ivln_lstm_bidir_bwd_f32",
    )
    return dgx_f, dgx_r, hp_f, hp_r


def cbra_bwd(dout, y, scale, shift, mean, rstd, train):
    N, Cc, H, W = y.shape
    dgamma = torch.empty((Cc,), dtype=torch.float32, device=y.device)
    dbeta = torch.empty((Cc,), dtype=torch.float32, device=y.device)
    dy = torch.empty_like(y)
    check(
        _T().ivln_cbra_bwd_f32(dptr(dout), dptr(y), dptr(scale), dptr(shift), dptr(mean), dptr(rstd), N, Cc, H, W,
                               int(bool(train)), dptr(dgamma), dptr(dbeta), dptr(dy), dptr(_reduce_ws(y.device)),
                               _reduce_ws(y.device).numel(), stream_ptr()),
        "ivln_cbra_bwd_f32",
    )
    return dy, dgamma, dbeta


def embedding_scatter_add(tokens, d, grad, padding_idx):
    rows, E = d.shape
    check(_T().ivln_embedding_scatter_add_f32(dptr(tokens), dptr(d), rows, E, grad.shape[0],
                                              -1 if padding_idx is None else padding_idx, dptr(grad), stream_ptr()),
          "ivln_embedding_scatter_add_f32")


def prev_action_embed_bwd(prev_actions, mask, d1, d2, n_emb):
    rows = prev_actions.numel()
    E = d1.shape[1]
    grad = torch.empty((n_emb, E), dtype=torch.float32, device=d1.device)
    check(
        _T().ivln_prev_action_embed_bwd_f32(dptr(prev_actions), dptr(mask), _p(d1), d1.stride(0), _p(d2),
                                            d2.stride(0) if d2 is not None else 0, rows, E, n_emb, dptr(grad),
                                            stream_ptr()),
        "ivln_prev_action_embed_bwd_f32",
    )
    return grad


def ce_iw_loss(logits, targets, weights, loss_scale=1.0):
    """logits (T,N,A) f32, targets (T,N) i64, weights (T,N) f32 -> (loss scalar tensor, dlogits)."""
    T, N, A = logits.shape
    loss = torch.empty((1,), dtype=torch.float32, device=logits.device)
    dl = torch.empty_like(logits)
    check(_T().ivln_ce_iw_loss_f32(dptr(logits), dptr(targets), dptr(weights), T, N, A, loss_scale, dptr(loss),
                                   dptr(dl), stream_ptr()), "ivln_ce_iw_loss_f32")
    return loss, dl


def pm_loss_fwd(pre, progress):
    n = pre.numel()
    hat = torch.empty((n,), dtype=torch.float32, device=pre.device)
    Lm = torch.empty((n, n), dtype=torch.float32, device=pre.device)
    check(_T().ivln_pm_loss_fwd_f32(dptr(pre), dptr(progress), n, dptr(hat), dptr(Lm), stream_ptr()),
          "ivln_pm_loss_fwd_f32")
    return hat, Lm


def pm_masked_mean_fwd(pre, progress, mask):
    """mean over the mask-selected columns of the (n, n) progress-monitor matrix, never materialised; returns
    (out2 = [mean, count] device tensor, hat, dsum)."""
    n = pre.numel()
    hat = torch.empty((n,), dtype=torch.float32, device=pre.device)
    dsum = torch.empty((n,), dtype=torch.float32, device=pre.device)
    out2 = torch.empty((2,), dtype=torch.float32, device=pre.device)
    L = _T()
    L.ivln_pm_masked_mean_fwd_f32.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp]
    check(L.ivln_pm_masked_mean_fwd_f32(dptr(pre), dptr(progress), dptr(mask), n, dptr(hat), dptr(dsum), dptr(out2),
                                        stream_ptr()), "ivln_pm_masked_mean_fwd_f32")
    return out2, hat, dsum


def pm_masked_mean_bwd(gout, hat, dsum, mask, out2, alpha):
    n = hat.numel()
    dpre = torch.empty((n,), dtype=torch.float32, device=hat.device)
    L = _T()
    L.ivln_pm_masked_mean_bwd_f32.argtypes = [vp, vp, vp, vp, vp, i32, f32, vp, vp]
    check(L.ivln_pm_masked_mean_bwd_f32(dptr(gout), dptr(hat), dptr(dsum), dptr(mask), dptr(out2), n, float(alpha),
                                        dptr(dpre), stream_ptr()), "ivln_pm_masked_mean_bwd_f32")
    return dpre


def pm_loss_bwd(dL, hat, progress):
    n = hat.numel()
    dpre = torch.empty((n,), dtype=torch.float32, device=hat.device)
    check(_T().ivln_pm_loss_bwd_f32(dptr(dL), dptr(hat), dptr(progress), n, dptr(dpre), stream_ptr()),
          "ivln_pm_loss_bwd_f32")
    return dpre


def adam_step(params, grads, exp_avg, exp_avg_sq, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, seg_of=None,
              seg_lr=None, grad_scale=1.0, zero_grad=True, guard=None):
    """guard: a one-word device tensor; non-zero at execution time = the step is skipped on the device (seq_guard_word)."""
    T = _T()
    T.ivln_adam_step_guarded_f32.argtypes = [vp, vp, vp, vp, i64, f32, vp, vp, f32, f32, f32, i32, f32, i32, vp, vp]
    check(
        T.ivln_adam_step_guarded_f32(dptr(params), dptr(grads), dptr(exp_avg), dptr(exp_avg_sq), params.numel(), lr,
                                     _p(seg_of), _p(seg_lr), beta1, beta2, eps, step, grad_scale, int(bool(zero_grad)),
                                     _p(guard), stream_ptr()),
        "ivln_adam_step_f32",
    )


# ---- GEMM-shaped gradients ----------------------------------------------------------------------------
def linear_bwd_input(dy, w, out=None, accumulate=False):
    """dX[r][i] = sum_o dY[r][o] W[o][i]  (dy, out may be row-strided)."""
    rows, O = dy.shape
    I = w.shape[1]
    if out is None:
        out = torch.empty((rows, I), dtype=torch.float32, device=dy.device)
    d = GemmDesc()
    d.A, d.B, d.D = dptr(w), _p(dy), _p(out)
    d.M, d.N, d.K = I, rows, O
    d.amode, d.bmode, d.dmode = A_KM, B_NK, D_DENSE
    d.lda, d.ldb = I, dy.stride(0)
    d.sDm, d.sDn = 1, out.stride(0)
    d.HoWo = 1
    _epilogue(d, None, None, None, False, accumulate)
    # rows x I is a small output (512 x 416 = 56 tiles) under a deep K (up to 3072 outputs): split-K fills the chip
    if LINEAR_BWD_SPLIT:
        ws = splitk_ws(dy.device)
        d.ws, d.ws_floats, d.splits = dptr(ws), ws.numel(), 0
    else:
        d.splits = 1
    gemm(d)
    return out


def linear_bwd_weight(dy, x, out=None, accumulate=False):
    """dW[o][i] = sum_r dY[r][o] X[r][i]  -> (O, I) contiguous."""
    rows, O = dy.shape
    I = x.shape[1]
    if out is None:
        out = torch.empty((O, I), dtype=torch.float32, device=dy.device)
    d = GemmDesc()
    d.A, d.B, d.D = _p(dy), _p(x), dptr(out)
    d.M, d.N, d.K = O, I, rows
    d.amode, d.bmode, d.dmode = A_KM, B_KN, D_DENSE
    d.lda, d.ldb = dy.stride(0), x.stride(0)
    d.sDm, d.sDn = I, 1
    d.HoWo = 1
    _epilogue(d, None, None, None, False, accumulate)
    if LINEAR_BWD_SPLIT or not accumulate:
        ws = splitk_ws(dy.device)  # the fixed-order slab epilogue adds into `out` when accumulating
        d.ws, d.ws_floats, d.splits = dptr(ws), ws.numel(), 0
    else:
        d.splits = 1
    gemm(d)
    return out


WGRAD_EXACT_X = os.environ.get("IVLN_WGRAD_EXACT_X", "1") != "0"  # A/B: 0 = the one-hot first layer's weight gradient stages x as three pieces too


def conv2d_bwd_weight(dy, x, KH, KW, stride=1, pad=0, x_exact_bf16=False):
    """dW (Cout, Cin, KH, KW) = sum over pixels dy[co][p] * im2col(x)[(ci,kh,kw)][p].
    x_exact_bf16: the caller's promise that every value of x is exact in bf16 (one-hot map features): the split-bf16 kernel
    stages x as one piece (ivln_gemm_desc.split_ok = 2; a broken promise gives NaNs, not a wrong gradient)."""
    N, Cout, Ho, Wo = dy.shape
    _, Cin, H, W = x.shape
    out = torch.empty((Cout, Cin, KH, KW), dtype=torch.float32, device=dy.device)
    d = GemmDesc()
    d.A, d.B, d.D = dptr(dy), dptr(x), dptr(out)
    d.M, d.N, d.K = Cout, Cin * KH * KW, N * Ho * Wo
    d.amode, d.bmode, d.dmode = A_NCHW_P, B_IM2COL_T, D_DENSE
    d.Cin, d.Hin, d.Win, d.Hout, d.Wout = Cin, H, W, Ho, Wo
    d.stride, d.pad, d.dil = stride, pad, 1
    d.HoWo = Ho * Wo
    d.sDm, d.sDn = Cin * KH * KW, 1
    koff, kpos = conv_tables(Cin, KH, KW, H, W, 1, x.device)
    d.koff, d.kpos = dptr(koff), dptr(kpos)
    _epilogue(d, None, None, None, False)
    ws = splitk_ws(dy.device)
    d.ws, d.ws_floats, d.splits = dptr(ws), ws.numel(), 0
    d.split_ok = int(SPLIT_BF16 and SPLIT_BF16_WGRAD) * (2 if (x_exact_bf16 and WGRAD_EXACT_X) else 1)
    gemm(d)
    return out
