"""Host side of the persistent depth encoder (csrc/depth_net.hip, `ivln_depth_net_f32`): turns a `ResNetEncoder`
(habitat-lab's GroupNorm ResNet-50 + compression, models/encoders/resnet_encoders.py:31-43, 95) into

  * an OP TABLE (`ivln_depthnet_op`, include/ivln_hip.h): one entry per conv layer with its tiling over the 32
    workgroups x 8 waves of an image's cluster, where its input comes from and how it is transformed on load (GroupNorm
    from statistics partials, downsample-branch GroupNorm or identity add, ReLU, MaxPool, the input's avg-pool), where
    the raw output and its statistics partials go;
  * the PACKED WEIGHTS: for every (16- or 8-row output-channel tile, K range of a wave) the k-steps in the order the
    MFMA consumes them, four k-steps per lane and 16-byte load:
        A[co][k-step s][slot kq]  =  W[co][4 q + kq][tap t]   with s = q * ks*ks + t        (ks = 1 | 3)
                                  =  W[co][0][tap 4 s + kq]   (0 for taps >= 49)             (the one-channel 7x7 stem)
        blob[((tile * KWT + kwt) * CPK + chunk) * (64 | 32) * 4 + lane_entry * 4 + u] = A[tile * M + i][kbeg(kwt) + 4 chunk + u][kq]
        lane_entry = lane (i = lane & 15, kq = lane >> 4) for M = 16, i + 8 kq for M = 8; k-steps past a range are zero;
  * the PARAMETER BLOB (GroupNorm gammas / betas) and the per-image ARENA layout (raw conv outputs, identity
    activations, statistics partials).

The packing and the wiring are testable without a GPU: tests/depth_net_emulator.py replays a program on the CPU with torch
from the PACKED weights and through the statistics-partial layout (tests/test_depth_net_program.py).  That emulator is
test infrastructure and lives under tests/; `DepthNetPlan.run` only ever launches the HIP kernel."""
import ctypes as C
import math

import torch

I32 = C.c_int
_FIELDS = ["kind", "Cin", "Cout", "ks", "stride", "pad", "Hin", "Win", "wout_shift", "M", "WCT", "WPT", "P", "KW", "kwg",
           "n_ctg", "n_ptg", "ksteps", "cs", "wp", "src_off", "nslab", "slab_stride", "st_off", "st_parts", "gamma_off",
           "beta_off", "src2_off", "st2_off", "st2_parts", "gamma2_off", "beta2_off", "res_off", "relu", "pool", "avg_in",
           "act_out_off", "dst_off", "dst_slab_stride", "st_out_off", "st_out_parts", "w_off", "barrier_before"]


class DepthNetOp(C.Structure):
    _fields_ = [(f, I32) for f in _FIELDS]


NWAVES, CLUSTER = 8, 32
ST_FLOATS = 16 * 32 * 4  # statistics partials of one tensor: [16 groups][<= 32 parts][4] = (count, mean, M2, -)


def _pow2_ge(x):
    return 1 << max(0, math.ceil(math.log2(max(1, x))))


def choose_tiling(Cout, Hout, Wout, ksteps, allow_kwg=False):
    """-> dict(M, WCT, WPT, P, KW, kwg, n_ctg, n_ptg).  A task (workgroup) = WCT channel tiles x (WPT * P) pixel tiles of
    16 whole output rows' worth; tasks <= 32; the 8 waves = WCT x WPT x KW."""
    npt = Hout * Wout // 16
    grp = max(1, Wout // 16)               # pixel tiles per task: whole output rows
    while npt // grp > CLUSTER:
        grp *= 2
    n_ptg = npt // grp
    P = 2 if grp >= 2 else 1
    WPT = grp // P
    assert WPT in (1, 2, 4, 8) and NWAVES % WPT == 0
    rw = NWAVES // WPT
    max_ctg = CLUSTER // n_ptg
    M = 16
    nct = Cout // M
    if nct * n_ptg <= CLUSTER // 2 and not allow_kwg:  # half the cluster idle with 16-row tiles: 8-row tiles (half-filled MFMA rows)
        M, nct = 8, Cout // 8
    WCT = min(rw, _pow2_ge(math.ceil(nct / max_ctg)))
    n_ctg = math.ceil(nct / WCT)
    assert n_ctg * WCT == nct, (Cout, M, WCT)
    KW = rw // WCT
    kwg = 1
    if allow_kwg and n_ctg * n_ptg < CLUSTER:
        kwg = CLUSTER // (n_ctg * n_ptg)
    assert n_ctg * n_ptg * kwg <= CLUSTER
    return dict(M=M, WCT=WCT, WPT=WPT, P=P, KW=KW, kwg=kwg, n_ctg=n_ctg, n_ptg=n_ptg)


def _k_ranges(ksteps, KWT):
    per = (ksteps + KWT - 1) // KWT
    cpk = (per + 3) // 4
    return per, cpk, [(min(k * per, ksteps), min(k * per + per, ksteps)) for k in range(KWT)]


def weight_matrix(w):
    """(Cout, Cin, ks, ks) -> A (Cout, ksteps, 4) in MFMA k order (module docstring)."""
    Cout, Cin, ks, _ = w.shape
    KK = ks * ks
    if ks == 7:
        assert Cin == 1
        a = torch.zeros(Cout, 52, dtype=w.dtype)
        a[:, :49] = w.reshape(Cout, 49)
        return a.view(Cout, 13, 4)
    assert Cin % 4 == 0
    # A[co][q*KK + t][kq] = W[co][4q + kq][t]
    return w.reshape(Cout, Cin // 4, 4, KK).permute(0, 1, 3, 2).reshape(Cout, (Cin // 4) * KK, 4).contiguous()


def pack_weights(w, M, KWT):
    """-> flat float32 tensor in the per-lane order of csrc/depth_net.hip (module docstring)."""
    A = weight_matrix(w.detach().float().cpu())
    Cout, ksteps, _ = A.shape
    per, cpk, ranges = _k_ranges(ksteps, KWT)
    nct = Cout // M
    ent = 64 if M == 16 else 32
    out = torch.zeros(nct, KWT, cpk, ent, 4, dtype=torch.float32)
    for kwt, (kb, ke) in enumerate(ranges):
        n = ke - kb
        if n <= 0:
            continue
        blk = torch.zeros(Cout, cpk * 4, 4)
        blk[:, :n] = A[:, kb:ke]
        # blk[co][4 chunk + u][kq] -> out[tile][kwt][chunk][i + (16|8) kq][u]
        v = blk.view(nct, M, cpk, 4, 4)          # tile, i, chunk, u, kq
        v = v.permute(0, 2, 4, 1, 3)             # tile, chunk, kq, i, u
        out[:, kwt] = v.reshape(nct, cpk, 4 * M, 4)[:, :, :ent]
    return out.reshape(-1)


class Program:
    """ops (list of dict), weight blob, parameter blob, arena layout."""

    def __init__(self):
        self.ops, self.wchunks, self.pchunks, self.names = [], [], [], []  # names: the conv module each op runs (state_dict path)
        self.w_floats = self.p_floats = 0
        self.arena = 0
        self.flops_per_image = 0

    def alloc(self, n):
        off = self.arena
        self.arena += (n + 63) // 64 * 64
        return off

    def add_params(self, t):
        off = self.p_floats
        self.pchunks.append(t.detach().float().cpu().reshape(-1))
        self.p_floats += t.numel()
        return off

    def add_weights(self, blob):
        off = self.w_floats
        self.wchunks.append(blob)
        self.w_floats += blob.numel()
        return off

    def ctypes_ops(self):
        arr = (DepthNetOp * len(self.ops))()
        for i, o in enumerate(self.ops):
            for f in _FIELDS:
                setattr(arr[i], f, int(o[f]))
        return arr


def _conv_op(prog, w, stride, pad, Hin, Win, src, barrier, allow_kwg=False, pool=False, avg_in=False, name=""):
    """One conv op reading `src` = dict(off, nslab, slab_stride, st_off, st_parts, gn, x2 (off, st_off, st_parts, gn) | None,
    res_off | None, relu, act_out_off | None).  Returns (op dict, out descriptor)."""
    Cout, Cin, ks, _ = w.shape
    Hout, Wout = (Hin + 2 * pad - ks) // stride + 1, (Win + 2 * pad - ks) // stride + 1
    assert Hout == Wout and Wout & (Wout - 1) == 0 and Win % 4 == 0
    ksteps = 13 if ks == 7 else (Cin // 4) * ks * ks
    t = choose_tiling(Cout, Hout, Wout, ksteps, allow_kwg)
    KWT = t["KW"] * t["kwg"]
    PG = 16 * t["WPT"] * t["P"]
    rows_out = PG // Wout
    assert rows_out >= 1 and rows_out * Wout == PG
    sub = ks == 1 and stride == 2  # the kernel stages the sub-sampled map of a 1x1 stride-2 conv
    Rs = (rows_out - 1) * (1 if sub else stride) + ks
    wp = Wout if sub else Win + 2 * pad
    cs = Rs * wp
    if ks != 7:
        if ks == 3 and stride == 2:
            cs = (cs + 3) // 4 * 4  # (the largest tiles of the program - 256 channels x 9 x 10 at layer 4: bank spread given up for 18 KB)
        else:
            while cs % 32 != 16:  # k slots of an MFMA step (4 channels) land on different LDS banks
                cs += 1
    rows_t = t["WCT"] * t["M"]
    cpo = Cout // 16
    st_out_parts = t["n_ptg"] * max(1, cpo // rows_t) if t["kwg"] == 1 else 0
    assert st_out_parts <= 32
    dst = prog.alloc(Cout * Hout * Wout * t["kwg"])
    st_out = prog.alloc(ST_FLOATS) if st_out_parts else 0
    op = dict(kind=0, Cin=Cin, Cout=Cout, ks=ks, stride=stride, pad=pad, Hin=Hin, Win=Win, wout_shift=int(math.log2(Wout)),
              ksteps=ksteps, cs=cs, wp=wp, relu=int(src.get("relu", 0)), pool=int(pool), avg_in=int(avg_in),
              src_off=src["off"], nslab=src.get("nslab", 1), slab_stride=src.get("slab_stride", 0),
              st_off=src.get("st_off", 0), st_parts=src.get("st_parts", 0), gamma_off=0, beta_off=0,
              src2_off=-1, st2_off=0, st2_parts=0, gamma2_off=0, beta2_off=0, res_off=-1,
              act_out_off=-1 if src.get("act_out_off") is None else src["act_out_off"],
              dst_off=dst, dst_slab_stride=Cout * Hout * Wout, st_out_off=st_out, st_out_parts=st_out_parts,
              w_off=prog.add_weights(pack_weights(w, t["M"], KWT)), barrier_before=int(barrier), **t)
    if src.get("gn") is not None:
        op["gamma_off"], op["beta_off"] = src["gn"]
    if src.get("x2") is not None:
        x2 = src["x2"]
        op.update(src2_off=x2["off"], st2_off=x2["st_off"], st2_parts=x2["st_parts"], gamma2_off=x2["gn"][0], beta2_off=x2["gn"][1])
    if src.get("res_off") is not None:
        op["res_off"] = src["res_off"]
    prog.ops.append(op)
    prog.names.append(name)
    prog.flops_per_image += 2 * Cout * Hout * Wout * Cin * ks * ks
    out = dict(off=dst, st_off=st_out, st_parts=st_out_parts, C=Cout, H=Hout, W=Wout, nslab=t["kwg"], slab_stride=Cout * Hout * Wout)
    return op, out


_UNSUPPORTED = None  # weak set of encoders whose architecture the persistent kernel does not cover (negative cache)


def supported(encoder):
    """True when `encoder` is the architecture csrc/depth_net.hip is written for - the DD-PPO default the reference
    builds (resnet_encoders.py:31-43): one-channel 7x7 stride-2 stem, GroupNorm with 16 groups, bottlenecks [3, 4, 6, 3]
    on 32 base planes (every tiling of `choose_tiling` and the kernel's statistics layout assume 16 groups), a 3x3
    compression conv with GroupNorm(1).  Anything else (`resnet_baseplanes` = 64 gives 32 groups) runs the launch chain /
    the conv + GroupNorm pairs as before; the verdict is cached per encoder object."""
    global _UNSUPPORTED
    import weakref

    if _UNSUPPORTED is None:
        _UNSUPPORTED = weakref.WeakSet()
    if encoder in _UNSUPPORTED:
        return False
    ok = getattr(encoder, "_depth_net_ok", None)
    if ok is None:
        try:
            bb = encoder.backbone
            c1, g1 = bb.conv1[0], bb.conv1[1]
            layers = (bb.layer1, bb.layer2, bb.layer3, bb.layer4)
            ok = (c1.in_channels == 1 and c1.kernel_size == (7, 7) and c1.stride == (2, 2) and c1.out_channels == 32
                  and g1.num_groups == 16 and [len(l) for l in layers] == [3, 4, 6, 3])
            for li, layer in enumerate(layers):
                for bi, blk in enumerate(layer):
                    c = blk.convs
                    planes = 32 * 2 ** li
                    ok = ok and len(c) == 8 and c[0].kernel_size == (1, 1) and c[3].kernel_size == (3, 3) and c[6].kernel_size == (1, 1)
                    ok = ok and c[0].out_channels == planes and c[3].out_channels == planes and c[6].out_channels == 4 * planes
                    ok = ok and all(c[i].num_groups == 16 for i in (1, 4, 7)) and all(c[i].bias is None for i in (0, 3, 6))
                    ok = ok and (blk.downsample is None or (blk.downsample[0].kernel_size == (1, 1) and blk.downsample[1].num_groups == 16))
                    ok = ok and (blk.downsample is not None) == (bi == 0)
            comp, gcomp = encoder.compression[0], encoder.compression[1]
            ok = bool(ok and comp.kernel_size == (3, 3) and comp.in_channels == 1024 and gcomp.num_groups == 1)
        except (AttributeError, IndexError, TypeError):
            ok = False
        if not ok:
            _UNSUPPORTED.add(encoder)
    return bool(ok)


def build_program(encoder):
    """`encoder`: ivln_ce_amd.encoders.ResNetEncoder (depth-only).  One image's program; every image of a batch runs it
    on its own arena."""
    bb = encoder.backbone
    prog = Program()
    gn_of = lambda gn: (prog.add_params(gn.weight), prog.add_params(gn.bias))  # noqa: E731
    H0 = W0 = None
    # --- stem: avg_pool2d(2) on load -> 7x7 stride 2 ---
    c1, g1 = bb.conv1[0], bb.conv1[1]
    assert c1.in_channels == 1 and c1.kernel_size == (7, 7) and g1.num_groups == 16
    Hin = Win = 128
    _, x = _conv_op(prog, c1.weight, 2, 3, Hin, Win, dict(off=0), barrier=False, avg_in=True, name="backbone.conv1.0")
    stem_gn = gn_of(g1)
    # the stem's GroupNorm + ReLU + MaxPool(3, 2, 1) happen on load in the first block's convs
    cur = dict(kind="pool", x=x, gn=stem_gn)
    H, W = x["H"] // 2, x["W"] // 2
    # identity activations ping-pong between two slots sized for the largest block output of the model (layer 1's)
    act_floats = max(b.convs[6].out_channels * (x["H"] // 2 // max(1, 2 ** li)) ** 2
                     for li, layer in enumerate((bb.layer1, bb.layer2, bb.layer3, bb.layer4)) for b in layer)
    act_slots = [prog.alloc(act_floats), prog.alloc(act_floats)]
    act_i = 0
    identity_off = None
    blocks = [b for layer in (bb.layer1, bb.layer2, bb.layer3, bb.layer4) for b in layer]
    bnames = [f"backbone.layer{li + 1}.{bi}" for li, layer in enumerate((bb.layer1, bb.layer2, bb.layer3, bb.layer4)) for bi in range(len(layer))]

    def tail_src(cur, want_act):
        """How the consumer of a block output reads it."""
        nonlocal act_i
        if cur["kind"] == "pool":
            return dict(off=cur["x"]["off"], st_off=cur["x"]["st_off"], st_parts=cur["x"]["st_parts"], gn=cur["gn"], relu=1), True
        s = dict(off=cur["x3"]["off"], st_off=cur["x3"]["st_off"], st_parts=cur["x3"]["st_parts"], gn=cur["gn3"], relu=1)
        if cur.get("xds") is not None:
            s["x2"] = dict(off=cur["xds"]["off"], st_off=cur["xds"]["st_off"], st_parts=cur["xds"]["st_parts"], gn=cur["gnds"])
        else:
            s["res_off"] = cur["identity"]
        if want_act:
            s["act_out_off"] = act_slots[act_i]
            act_i ^= 1
        return s, False

    for bi, blk in enumerate(blocks):
        c = blk.convs
        has_ds = blk.downsample is not None
        src, pool = tail_src(cur, want_act=not has_ds)
        _, x1 = _conv_op(prog, c[0].weight, 1, 0, H, W, src, barrier=True, pool=pool, name=bnames[bi] + ".convs.0")
        identity = src.get("act_out_off")
        xds = gnds = None
        if has_ds:
            src_ds = dict(src)
            src_ds.pop("act_out_off", None)
            _, xds = _conv_op(prog, blk.downsample[0].weight, blk.stride, 0, H, W, src_ds, barrier=False, pool=pool,
                              name=bnames[bi] + ".downsample.0")
            gnds = gn_of(blk.downsample[1])
        _, x2 = _conv_op(prog, c[3].weight, blk.stride, 1, H, W,
                         dict(off=x1["off"], st_off=x1["st_off"], st_parts=x1["st_parts"], gn=gn_of(c[1]), relu=1), barrier=True,
                         name=bnames[bi] + ".convs.3")
        H, W = x2["H"], x2["W"]
        _, x3 = _conv_op(prog, c[6].weight, 1, 0, H, W,
                         dict(off=x2["off"], st_off=x2["st_off"], st_parts=x2["st_parts"], gn=gn_of(c[4]), relu=1), barrier=True,
                         name=bnames[bi] + ".convs.6")
        cur = dict(kind="block", x3=x3, gn3=gn_of(c[7]), xds=xds, gnds=gnds, identity=identity)
    # --- compression: 3x3, K split over workgroups (slabs), then GroupNorm(1) + ReLU ---
    comp, gcomp = encoder.compression[0], encoder.compression[1]
    assert gcomp.num_groups == 1
    src, _ = tail_src(cur, want_act=False)
    _, xc = _conv_op(prog, comp.weight, 1, 1, H, W, src, barrier=True, allow_kwg=True, name="compression.0")
    fin = {f: 0 for f in _FIELDS}
    fin.update(kind=1, Cin=xc["C"], Hin=xc["H"], Win=xc["W"], src_off=xc["off"], nslab=xc["nslab"], slab_stride=xc["slab_stride"],
               gamma_off=prog.add_params(gcomp.weight), beta_off=prog.add_params(gcomp.bias), barrier_before=1, src2_off=-1, res_off=-1,
               act_out_off=-1, WCT=1, WPT=1, P=1, KW=8, kwg=1, n_ctg=1, n_ptg=1, M=16)
    prog.ops.append(fin)
    prog.names.append("compression.1")
    prog.out_shape = (xc["C"], xc["H"], xc["W"])
    prog.eps = float(g1.eps)
    return prog


# ------------------------------------------------------------------------------------------------
# device plan
# ------------------------------------------------------------------------------------------------
class DepthNetPlan:
    """Device-side state of one encoder: op table, packed weights and parameters (shared by every launch), and PER STREAM an
    arena for up to 8 images + the cluster sync words.  Two launches that may overlap on the device - the side-stream graph
    of a split replay and an eager act() on the main stream, two runners captured on one policy - never share an arena or
    arrival counters; launches on ONE stream are ordered by the stream (the contract the workspaces of ops.py follow)."""

    MAX_IMAGES = 8

    def __init__(self, encoder, device):
        from . import ops
        from ._lib import lib

        self.prog = build_program(encoder)
        self.device = device
        self._ops_host = self.prog.ctypes_ops()
        raw = bytes(self._ops_host)
        self.ops_dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
        self.weights = torch.cat(self.prog.wchunks).to(device)
        self.params = torch.cat(self.prog.pchunks).to(device)
        self.arena_stride = (self.prog.arena + 255) // 256 * 256
        self._per_stream = {}  # stream handle -> (arena, sync words)
        # a pinned host word every sync workspace of this plan points to: a launch that times out raises it, and the host reads
        # it - a plain memory read - right after the stream synchronisation its loops perform anyway (`failed`)
        self.host_flag = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.disabled = False  # set by `recover`: the rest of the run uses the launch chain
        self.stamp = self.stamp_of(encoder)
        L = lib()
        vp, i64 = C.c_void_p, C.c_int64
        L.ivln_depth_net_f32.argtypes = [vp, C.POINTER(DepthNetOp), I32, vp, vp, vp, i64, vp, i64, vp, i64, I32, C.c_float, vp, vp]
        L.ivln_depth_net_status.argtypes = [vp, vp]
        L.ivln_depth_net_reset.argtypes = [vp, vp]
        L.ivln_host_device_ptr.argtypes = [vp, C.POINTER(vp)]
        self._L, self._ops = L, ops
        dp = vp()
        from ._lib import check
        check(L.ivln_host_device_ptr(self.host_flag.data_ptr(), C.byref(dp)), "ivln_host_device_ptr")
        self._host_flag_dev = int(dp.value)

    @staticmethod
    def stamp_of(encoder):
        """Changes when the encoder's weights may have changed: tensor versions / addresses, and - only when some of its
        parameters train (FlatAdam then updates them through raw pointers without touching `_version`) - ops.WEIGHT_EPOCH.
        The DD-PPO depth encoder is frozen in every reference configuration (resnet_encoders.py:45-46)."""
        from . import ops

        ps = list(encoder.parameters())
        epoch = ops.WEIGHT_EPOCH if any(p.requires_grad for p in ps) else 0
        return (epoch,) + tuple((p._version, p.data_ptr()) for p in ps)

    def refresh(self, encoder):
        """New weights, same architecture: repack into the SAME device buffers (a captured graph keeps raw pointers to
        them; the op table does not depend on the weights' values).  The copies are ordered behind every launch that may
        still read the old weights (the device is drained first: this is a rare, host-heavy path - a frozen encoder, the
        only kind the reference configures, never comes here) and in front of the next launch on any stream."""
        prog = build_program(encoder)
        assert prog.w_floats == self.weights.numel() and prog.p_floats == self.params.numel()
        torch.cuda.synchronize(self.device)
        self.weights.copy_(torch.cat(prog.wchunks))
        self.params.copy_(torch.cat(prog.pchunks))
        torch.cuda.synchronize(self.device)
        self.stamp = self.stamp_of(encoder)

    def stream_state(self, create=True):
        """(arena, sync words) of the CURRENT stream; None when it does not exist yet and may not be created (a capture in
        progress: the warm-up step on the capturing stream creates it)."""
        key = self._ops.stream_ptr()
        st = self._per_stream.get(key)
        if st is None and create and not torch.cuda.is_current_stream_capturing():
            sync = torch.zeros(512, dtype=torch.int32)
            lo, hi = self._host_flag_dev & 0xFFFFFFFF, self._host_flag_dev >> 32
            sync[258], sync[259] = (lo - (1 << 32) if lo >= 1 << 31 else lo), (hi - (1 << 32) if hi >= 1 << 31 else hi)
            st = self._per_stream[key] = (
                torch.zeros(self.MAX_IMAGES * self.arena_stride, dtype=torch.float32, device=self.device), sync.to(self.device))
        return st

    def failed(self):
        """True when a launch of this plan timed out (a cluster barrier's bounded spin: some workgroup never became
        resident).  A host memory read - meaningful after a synchronisation of the stream(s) the launches ran on."""
        return int(self.host_flag[0]) != 0

    def recover(self):
        """After `failed`: clear every workspace's counters and error word (the device is drained first: the launch that
        failed has wound down, the ones queued behind it returned at entry) and retire the plan - `plan_for` answers None
        from now on, i.e. the encoder runs the launch chain for the rest of the run."""
        torch.cuda.synchronize(self.device)
        from ._lib import check

        for _, sync in list(self._per_stream.values()):
            check(self._L.ivln_depth_net_reset(self._ops.dptr(sync), self._ops.stream_ptr()), "ivln_depth_net_reset")
        torch.cuda.synchronize(self.device)
        self.host_flag.zero_()
        self.disabled = True

    def run(self, depth, out, out_img_stride):
        """depth (B, H, W, 1) float32 contiguous on the device -> out[b * out_img_stride + ...] (C, h, w) per image.
        Returns False when the library declines (IVLN_E_UNSUPPORTED: not all workgroups resident)."""
        from ._lib import IVLN_E_UNSUPPORTED, check

        ops = self._ops
        B, H, W, _ = depth.shape
        assert B <= self.MAX_IMAGES and H == 256 and W == 256
        st = self.stream_state()
        if st is None:
            return False
        arena, sync = st
        rc = self._L.ivln_depth_net_f32(ops.dptr(self.ops_dev), self._ops_host, len(self.prog.ops), ops.dptr(self.weights),
                                        ops.dptr(self.params), ops.dptr(depth), H * W, ops.dptr(arena), self.arena_stride,
                                        out.data_ptr(), out_img_stride, B, self.prog.eps, ops.dptr(sync), ops.stream_ptr())  # (`out`: a channel slice of a wider buffer)
        if rc == IVLN_E_UNSUPPORTED:
            return False
        check(rc, "ivln_depth_net_f32")
        return True

    def check_status(self):
        from ._lib import check

        for _, sync in list(self._per_stream.values()):
            check(self._L.ivln_depth_net_status(self._ops.dptr(sync), self._ops.stream_ptr()),
                  "ivln_depth_net_status (a cluster barrier of the persistent depth encoder timed out)")


_PLANS = {}


def ops_weight_flags(encoder):
    return tuple(p.requires_grad for p in encoder.parameters())


def plan_for(encoder, device):
    """The encoder's plan on `device`, rebuilt when its parameters change.  None while a graph is being captured and no
    plan exists yet (building one allocates and uploads: the warm-up step before a capture creates it).  Plans of encoders
    that no longer exist are dropped (each holds ~100 MB: arena for 8 images + packed weights)."""
    import weakref

    if not supported(encoder):
        return None
    # A trainable encoder changes after every optimizer step: each change would cost a host-side repack of 23 M weights
    # (hundreds of ms) - it runs the launch chain.  (The reference freezes the DD-PPO encoder, resnet_encoders.py:45-46;
    # checked once per encoder object and requires_grad pattern.)
    rg = getattr(encoder, "_depth_net_rg", None)
    if rg is None or rg[0] != ops_weight_flags(encoder):
        flags = ops_weight_flags(encoder)
        encoder._depth_net_rg = rg = (flags, any(flags))
    if rg[1]:
        return None
    for k in [k for k, (ref, _) in _PLANS.items() if ref() is None]:
        del _PLANS[k]
    key = (id(encoder), str(device))
    ent = _PLANS.get(key)
    stamp = DepthNetPlan.stamp_of(encoder)
    if ent is not None and ent[0]() is encoder and ent[1].disabled:
        return None
    if ent is not None and ent[0]() is encoder and ent[1].stamp != stamp:
        if torch.cuda.is_current_stream_capturing():
            return None
        ent[1].refresh(encoder)
    elif ent is None or ent[0]() is not encoder:
        if torch.cuda.is_current_stream_capturing():
            return None
        ent = _PLANS[key] = (weakref.ref(encoder), DepthNetPlan(encoder, device))
    return ent[1]


def check_all():
    """Raises when a cluster barrier of any plan's launches timed out (sticky word); synchronises the current stream."""
    for _, plan in list(_PLANS.values()):
        plan.check_status()


def armed():
    """A persistent launch may be part of the steps being run: some plan exists and has not been retired.  (Cheap: what the
    loops ask before they spend a stream synchronisation on `any_failed`.)"""
    return any(not plan.disabled for _, plan in _PLANS.values())


def any_failed():
    """A persistent launch of some plan timed out (host-visible flags: no device access; call after the step's stream
    synchronisation)."""
    return any(plan.failed() for _, plan in _PLANS.values())


def recover_all():
    """Clears and retires every plan that failed; returns how many."""
    import logging

    n = 0
    for _, plan in list(_PLANS.values()):
        if plan.failed():
            plan.recover()
            n += 1
    if n:
        logging.getLogger("ivln_ce_amd").warning(
            "persistent depth encoder: a cluster barrier timed out (the launch's workgroups were not all resident); the step is "
            "computed again on the launch chain and the persistent form stays off for the rest of the run")
    return n
