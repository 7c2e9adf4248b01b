"""HIP backward of the MapCMA policy and the DAgger update step.

`MapCMAForwardFn` wraps the whole net forward (policy.MapCMANet.forward_hip) in ONE
torch.autograd.Function: autograd is only the plumbing that hands us d(features) and accumulates
the parameter gradients we return; every gradient is computed by hand-written HIP kernels
(csrc/train_ops.hip for the element/recurrence parts, csrc/gemm_conv.hip for every dX / dW GEMM).
So the reference's `loss.backward()` (ivlnce_baselines/common/base_il_trainer.py:211) works on
this policy unchanged, and `update_agent` below is the all-HIP version of `_update_agent`
(:173-219) with fused inflection-weighted CE, flat-bucket Adam and one RCCL all-reduce.
"""
import os
from typing import Dict, List

import torch

from . import ops


# The instruction bi-LSTM (forward 0.3 ms, BPTT 0.6 ms at T*N = 512 rows) is a latency-bound recurrence that
# leaves the chip idle, and it is independent of the map CNN's convolutions (3.5 / 5.5 ms, MFMA-bound): in a
# training pass it runs on a side stream next to them.  A/B switch for measurements and tests.
OVERLAP_INSTRUCTION = True
# (The map CNN's four weight gradients on a second side stream beside the sequential dgrad chain: built and measured in
#  round 5 - 9.40 ms with, 9.36 without: both kernels fill the chip with 256-register workgroups - and removed in round 6.)
_side = {}


def side_stream(device, role="txt"):
    key = (str(device), role)
    if key not in _side:
        _side[key] = torch.cuda.Stream(device)
    return _side[key]


def share_with_stream(obj, stream):
    """Tensors allocated on one stream and read on another: tell the caching allocator (record_stream)."""
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            share_with_stream(v, stream)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            share_with_stream(v, stream)


# ------------------------------------------------------------------------------------------------
# small autograd wrappers (plumbing only)
# ------------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """nn.Linear forward/backward on HIP (action head, progress monitor)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = x.contiguous()
        ctx.save_for_backward(x, w)
        return ops.linear(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = ops.linear_bwd_input(dy, w) if ctx.needs_input_grad[0] else None
        dw = ops.linear_bwd_weight(dy, x) if ctx.needs_input_grad[1] else None
        db = ops.colsum(dy) if ctx.needs_input_grad[2] else None
        return dx, dw, db


class _PMLossFn(torch.autograd.Function):
    """L[j][i] = (tanh(pre[i]) - progress[j])^2 (quirk Q7 broadcast, map_cma_policy.py:355-361)."""

    @staticmethod
    def forward(ctx, pre, progress):
        pre = pre.contiguous()
        progress = progress.to(torch.float32).reshape(-1).contiguous()
        hat, Lm = ops.pm_loss_fwd(pre, progress)
        ctx.save_for_backward(hat, progress)
        return Lm

    @staticmethod
    def backward(ctx, dL):
        hat, progress = ctx.saved_tensors
        return ops.pm_loss_bwd(dL.contiguous(), hat, progress), None


class _PMMaskedMeanFn(torch.autograd.Function):
    """alpha * mean(L[:, mask]) of the same matrix in one workgroup, without writing the matrix (what
    AuxLosses.reduce makes of the term: aux_losses.py:22-29).  The upstream gradient stays a device scalar."""

    @staticmethod
    def forward(ctx, pre, progress, mask, alpha):
        out2, hat, dsum = ops.pm_masked_mean_fwd(pre.contiguous(), progress, mask)
        ctx.save_for_backward(hat, dsum, mask, out2)
        ctx.alpha = float(alpha)
        return out2[0] * ctx.alpha if ctx.alpha != 1.0 else out2[0].clone()

    @staticmethod
    def backward(ctx, g):
        hat, dsum, mask, out2 = ctx.saved_tensors
        g = g.to(torch.float32).reshape(1).contiguous()
        return ops.pm_masked_mean_bwd(g, hat, dsum, mask, out2, ctx.alpha), None, None, None


class PMLossTerm:
    """The progress monitor's loss term as the registry holds it: lazily the reference's (TN, TN) matrix
    (`values`, for `AuxLosses.get_loss`), and - what the update actually asks for - its masked mean computed
    directly (`masked_mean`), with no matrix, no masked_select and no host synchronisation."""

    def __init__(self, pre, progress):
        self.pre = pre
        self.progress = progress.to(torch.float32).reshape(-1).contiguous()

    @property
    def values(self):
        return _PMLossFn.apply(self.pre, self.progress)

    def masked_mean(self, mask, alpha):
        mask = mask.reshape(-1)
        if mask.numel() != self.pre.numel() or not self.pre.is_cuda:
            return alpha * torch.masked_select(self.values, mask).mean()
        return _PMMaskedMeanFn.apply(self.pre, self.progress, mask.to(torch.uint8).contiguous(), alpha)


def progress_monitor_loss(net, feats, progress):
    pm = net.progress_monitor
    pre = LinearFn.apply(feats, pm.weight, pm.bias)  # (rows, 1)
    return PMLossTerm(pre.reshape(-1), progress)


# ------------------------------------------------------------------------------------------------
# whole-net backward
# ------------------------------------------------------------------------------------------------
def _gru_backward(enc, s, d_out, G):
    """BPTT of the masked GRU state encoder; returns d(x) (rows, I)."""
    rnn = enc.rnn
    T, N = s["T"], s["N"]
    H = rnn.hidden_size
    rows = T * N
    dev = d_out.device
    out, h0, masks, x_in = s["out"], s["h0"], s["masks"], s["x"]
    whh_t = ops.transpose(rnn.weight_hh_l0)  # (H, 3H): dh_prev = dgh . W_hh as a skinny linear
    dgi = torch.empty((rows, 3 * H), dtype=torch.float32, device=dev)
    dgh = torch.empty((rows, 3 * H), dtype=torch.float32, device=dev)
    hp = torch.empty((rows, H), dtype=torch.float32, device=dev)
    dhz = torch.empty((N, H), dtype=torch.float32, device=dev)
    # step T-1: element part alone; every earlier step t-1 rides in the launch that finishes step t - the whole chain
    # enqueued by one C-ABI call (ivln_cma_seq_bwd_f32)
    ops.gru_seq_bwd(d_out, s["r"], s["z"], s["n"], s["ghn"], out, h0, masks, whh_t, T, N, dgi, dgh, hp, dhz)
    G[rnn.weight_ih_l0] = ops.linear_bwd_weight(dgi, x_in)
    G[rnn.bias_ih_l0] = _colsum(dgi)
    G[rnn.weight_hh_l0] = ops.linear_bwd_weight(dgh, hp)
    G[rnn.bias_hh_l0] = _colsum(dgh)
    return ops.linear_bwd_input(dgi, rnn.weight_ih_l0)


def _conv1d_backward(conv, d_out4, x4, d_in_residual4, G):
    """Conv1d(k=1) viewed as a 1x1 conv over (rows, C, 1, P): returns d(x) (+ residual)."""
    O, Cc = conv.weight.shape[0], conv.weight.shape[1]
    w_t = ops.transpose(conv.weight.view(O, Cc))  # (C, O)
    G[conv.weight] = ops.conv2d_bwd_weight(d_out4, x4, 1, 1).view(O, Cc, 1)
    G[conv.bias] = ops.nchw_chansum(d_out4)
    return ops.conv2d(d_out4, w_t.view(Cc, O, 1, 1), residual=d_in_residual4)


def instruction_backward(ie, st, d_txt, rows, L, G):
    """Bidirectional-LSTM BPTT of the instruction encoder (+ the embedding table when it trains)."""
    rnn = ie.encoder_rnn
    dgx_f, dgx_r, hp_f, hp_r = ops.lstm_bidir_bwd(d_txt.contiguous(), st["out"], st["gates"], st["cs"],
                                                  rnn.weight_hh_l0, rnn.weight_hh_l0_reverse, st["lengths"], rows,
                                                  L, rnn.hidden_size)
    emb_x = st["emb"]
    G[rnn.weight_ih_l0] = ops.linear_bwd_weight(dgx_f, emb_x)
    G[rnn.weight_ih_l0_reverse] = ops.linear_bwd_weight(dgx_r, emb_x)
    G[rnn.weight_hh_l0] = ops.linear_bwd_weight(dgx_f, hp_f)
    G[rnn.weight_hh_l0_reverse] = ops.linear_bwd_weight(dgx_r, hp_r)
    bf, br = _colsum(dgx_f), _colsum(dgx_r)
    G[rnn.bias_ih_l0], G[rnn.bias_hh_l0] = bf, bf
    G[rnn.bias_ih_l0_reverse], G[rnn.bias_hh_l0_reverse] = br, br
    if ie.embedding_layer.weight.requires_grad:
        d_emb = ops.linear_bwd_input(dgx_f, rnn.weight_ih_l0)
        ops.linear_bwd_input(dgx_r, rnn.weight_ih_l0_reverse, out=d_emb, accumulate=True)
        g = torch.zeros_like(ie.embedding_layer.weight)
        ops.embedding_scatter_add(st["tokens"].reshape(-1), d_emb, g, ie.embedding_layer.padding_idx)
        G[ie.embedding_layer.weight] = g


_ZEROS = {}


def _zeros_like_cached(p):
    """A shared, never-written zero tensor of p's shape (exact-zero gradients: one fill per process instead of per update)."""
    key = (tuple(p.shape), str(p.device))
    z = _ZEROS.get(key)
    if z is None:
        z = _ZEROS[key] = torch.zeros_like(p)
    return z


_CS = None  # the running backward pass's queue of deferred column sums (bias gradients), see ops.ColsumQueue


def _colsum(x):
    return _CS.add(x) if _CS is not None else ops.colsum(x)


def net_backward(net, S: Dict, d_feats: torch.Tensor) -> Dict:
    """Gradients of every parameter of MapCMANet given d(loss)/d(features).  S = saves of forward_hip."""
    global _CS
    _CS = ops.ColsumQueue()  # bias gradients: queued here, computed by two launches at the end (ops.ColsumQueue)
    try:
        G = _net_backward(net, S, d_feats)
        _CS.flush()
        return G
    finally:
        _CS = None


def _net_backward(net, S: Dict, d_feats: torch.Tensor) -> Dict:
    G: Dict = {}
    rows, L, P = S["rows"], S["L"], S["P"]
    H = net._hidden_size
    h2 = H // 2
    scale = net._scale_f
    o_txt, o_dep, o_map, o_prev = S["offs"]
    x2, state_in = S["x2"], S["state_in"]
    dev = d_feats.device
    d_out_dep = net.depth_linear[1].out_features
    m_out = net.map_linear[1].out_features

    # ---- second GRU and its input compression ---------------------------------------------------
    d_c2 = _gru_backward(net.second_state_encoder, S["g2"], d_feats.contiguous(), G)
    d_pre = ops.relu_bwd(d_c2, S["c2"])
    sc = net.second_state_compress[0]
    G[sc.weight] = ops.linear_bwd_weight(d_pre, x2)
    G[sc.bias] = _colsum(d_pre)
    dx2 = ops.linear_bwd_input(d_pre, sc.weight)  # (rows, 1184): [state | text | dep' | map' | prev]

    # ---- depth / map attentions keyed by the attended text --------------------------------------
    dkv, mkv = S["dkv"], S["mkv"]
    d_dkv = torch.empty_like(dkv)
    d_mkv = torch.empty_like(mkv)
    dq2_d = torch.empty((rows, h2), dtype=torch.float32, device=dev)
    dq2_m = torch.empty((rows, h2), dtype=torch.float32, device=dev)
    ops.attn_bwd(dx2[:, o_dep:o_dep + d_out_dep], S["a_dep"], S["q2"], dkv[:, :h2], dkv[:, h2:], scale, dq2_d,
                 d_dkv[:, :h2], d_dkv[:, h2:])
    ops.attn_bwd(dx2[:, o_map:o_map + m_out], S["a_map"], S["q2"], mkv[:, :h2], mkv[:, h2:], scale, dq2_m,
                 d_mkv[:, :h2], d_mkv[:, h2:])
    dq2 = ops.add2d(dq2_d, dq2_m)
    text = x2[:, o_txt:o_txt + 256]
    G[net.text_q.weight] = ops.linear_bwd_weight(dq2, text)
    G[net.text_q.bias] = _colsum(dq2)
    d_text = dx2[:, o_txt:o_txt + 256]
    ops.linear_bwd_input(dq2, net.text_q.weight, out=d_text, accumulate=True)

    # ---- text attention (v = the LSTM outputs themselves, k = text_k(outputs)) -------------------
    txt, tk, inv = S["txt_out"], S["tk"], S.get("inv")
    U = txt.shape[0]  # == rows unless the batch came with its unique instruction rows (policy.forward_hip)
    dq1 = torch.empty((rows, h2), dtype=torch.float32, device=dev)
    d_tk = torch.empty((rows, h2, L), dtype=torch.float32, device=dev)
    d_txt = torch.empty((rows, txt.shape[1], L), dtype=torch.float32, device=dev)
    ops.attn_bwd(d_text, S["a_txt"], S["q1"], tk.view(U, h2, L), txt, scale, dq1, d_tk, d_txt, row_index=inv)
    if inv is not None:  # fold the per-row gradients onto the U shared instruction encodings (fixed row order)
        d_tk, d_txt = ops.index_sum(d_tk, inv, U), ops.index_sum(d_txt, inv, U)
    d_txt = _conv1d_backward(net.text_k, d_tk.view(U, h2, 1, L), txt.view(U, -1, 1, L),
                             d_txt.view(U, -1, 1, L), G).view(U, -1, L)
    G_txt, side, main = None, None, torch.cuda.current_stream()
    if OVERLAP_INSTRUCTION and not torch.cuda.is_current_stream_capturing():
        # d(instruction features) is final here: the bi-LSTM BPTT runs beside the GRU / map-CNN backward below
        side, G_txt = side_stream(dev), {}
        side.wait_stream(main)
        share_with_stream((d_txt, S["txt"]), side)
        with torch.cuda.stream(side):
            instruction_backward(net.instruction_encoder, S["txt"], d_txt, U, L, G_txt)
    state = x2[:, :H]
    G[net.state_q.weight] = ops.linear_bwd_weight(dq1, state)
    G[net.state_q.bias] = _colsum(dq1)
    d_state = dx2[:, :H]
    ops.linear_bwd_input(dq1, net.state_q.weight, out=d_state, accumulate=True)

    # ---- first GRU ----------------------------------------------------------------------------------
    d_state_in = _gru_backward(net.state_encoder, S["g1"], d_state, G)  # (rows, 416)
    emb = net.prev_action_embedding
    G[emb.weight] = ops.prev_action_embed_bwd(S["prev_actions"], S["masks"], d_state_in[:, d_out_dep + m_out:],
                                              dx2[:, o_prev:], emb.num_embeddings)

    # ---- depth branch: depth_linear + dep_kv -> spatial embedding (visual encoder is frozen) --------
    dep, mp = S["dep"], S["mp"]
    Cd, Cm = dep.shape[1], mp.shape[1]
    dl, ml = net.depth_linear[1], net.map_linear[1]
    d_pre_d = ops.relu_bwd(d_state_in[:, :d_out_dep], state_in[:, :d_out_dep])
    G[dl.weight] = ops.linear_bwd_weight(d_pre_d, dep.view(rows, -1))
    G[dl.bias] = _colsum(d_pre_d)
    d_dep = ops.linear_bwd_input(d_pre_d, dl.weight)  # (rows, 192*16)
    d_dep = _conv1d_backward(net.dep_kv, d_dkv.view(rows, -1, 1, P), dep.view(rows, Cd, 1, P),
                             d_dep.view(rows, Cd, 1, P), G)
    se = net.depth_encoder.spatial_embeddings
    c_vis = Cd - se.embedding_dim
    G[se.weight] = _colsum(d_dep.view(rows, -1)[:, c_vis * P:]).view_as(se.weight)

    # ---- map branch: map_linear + map_kv -> map CNN -----------------------------------------------
    d_pre_m = ops.relu_bwd(d_state_in[:, d_out_dep:d_out_dep + m_out], state_in[:, d_out_dep:d_out_dep + m_out])
    G[ml.weight] = ops.linear_bwd_weight(d_pre_m, mp.view(rows, -1))
    G[ml.bias] = _colsum(d_pre_m)
    d_mp = ops.linear_bwd_input(d_pre_m, ml.weight)
    d_mp = _conv1d_backward(net.map_kv, d_mkv.view(rows, -1, 1, P), mp.view(rows, Cm, 1, P),
                            d_mp.view(rows, Cm, 1, P), G)
    if any(p.requires_grad for p in net.map_encoder.parameters()):
        d = d_mp.view(mp.shape)
        blocks = list(net.map_encoder.cnn)
        for i in range(len(blocks) - 1, -1, -1):
            conv, bn = blocks[i].conv[0], blocks[i].conv[1]
            s = S["map"][i]
            if s["train"]:
                mean, rstd = s["mean"], s["rstd"]
            else:
                mean, rstd = bn.running_mean, torch.rsqrt(bn.running_var + bn.eps)
            dy, dgamma, dbeta = ops.cbra_bwd(d.contiguous(), s["y"], s["scale"], s["shift"], mean, rstd, s["train"])
            G[bn.weight], G[bn.bias] = dgamma, dbeta
            # (layer 0 reads ops.map_features' output: zeros and ones - encoders.SemanticMapEncoder.generate_map_features)
            G[conv.weight] = ops.conv2d_bwd_weight(dy, s["x"], 7, 7, 1, 3, x_exact_bf16=(i == 0))
            if s["train"]:
                # train-mode BatchNorm: sum_{n,h,w} dy = gamma*rstd*(S1 - S1 - S2*sum(xhat)/M) and sum(xhat) = 0,
                # so the gradient of a conv bias feeding it is identically zero (autograd in the reference
                # returns ~1e-9 rounding noise there); no pass over the 270 MB dy tensor
                G[conv.bias] = _zeros_like_cached(conv.bias)
            else:
                G[conv.bias] = ops.nchw_chansum(dy)
            if i > 0:
                d = ops.conv2d(dy, ops.weight_flip_transpose(conv.weight), pad=3, weight_is_temp=True)

    if G_txt is None:
        instruction_backward(net.instruction_encoder, S["txt"], d_txt, U, L, G)
    else:
        main.wait_stream(side)
        share_with_stream(G_txt, main)
        G.update(G_txt)
    return G


def all_params(net):
    """list(net.parameters()), cached per module.  The cache is keyed by the identity of every sub-module's `_parameters`
    / `_modules` entries' count and of the Parameter objects themselves (a cheap walk over ~40 modules, no tensor work):
    replacing a sub-module or re-assigning a Parameter after the first forward (e.g. swapping in a pretrained encoder)
    rebuilds the list instead of silently training the old objects.  Note for users of autograd hooks: with FlatAdam the
    HIP backward adds gradients straight into the existing `.grad` views (MapCMAForwardFn.backward) and reports None
    to autograd for those parameters, so per-parameter hooks / torch.autograd.grad do not see them."""
    stamp = tuple(id(p) for m in net.modules() for p in m._parameters.values())
    ent = net.__dict__.get("_all_params_cache")
    if ent is None or ent[0] != stamp:
        ent = net.__dict__["_all_params_cache"] = (stamp, list(net.parameters()))
    return ent[1]


class MapCMAForwardFn(torch.autograd.Function):
    """`holder` = (net, *forward_hip arguments); the net's `backward_hip` (default: net_backward) turns
    d(features) into the parameter gradients."""

    @staticmethod
    def forward(ctx, holder, *params):
        net, args = holder[0], holder[1:]
        save: Dict = {}
        with torch.no_grad():
            feats, rnn_out = net.forward_hip(*args, save=save)
        ctx.net, ctx.saves, ctx.params = net, save, params
        ctx.mark_non_differentiable(rnn_out)
        return feats, rnn_out

    @staticmethod
    def backward(ctx, d_feats, _d_rnn):
        with torch.no_grad():
            G = getattr(ctx.net, "backward_hip", None)
            G = (G or (lambda S, d: net_backward(ctx.net, S, d)))(ctx.saves, d_feats)
        # Parameters whose .grad already exists as a contiguous device tensor (FlatAdam: views of the flat bucket) get
        # their gradient ADDED there by one multi-tensor launch and report None to autograd (a None gradient leaves
        # .grad alone); autograd's own accumulation is one elementwise launch per parameter.  Anything else - no
        # .grad yet, a foreign layout - goes back through autograd as before.
        grads: List = []
        direct = []
        for p, need in zip(ctx.params, ctx.needs_input_grad[1:]):
            g = G.get(p) if need else None
            if (g is not None and p.grad is not None and p.grad.is_cuda and p.grad.is_contiguous()
                    and g.is_contiguous() and g.dtype == torch.float32 and p.grad.dtype == torch.float32):
                direct.append((g, p.grad))
                grads.append(None)
            else:
                grads.append(g.view_as(p) if g is not None else None)
        if direct:
            ops.add_multi(direct)
        ctx.saves = None
        return (None, *grads)

    @staticmethod
    def run(net, *args):
        # (Module.parameters() walks the module tree with de-duplication: 0.4 ms of host time per update, paid while the
        #  GPU idles behind the previous update's .item(); the tree is fixed after construction, only the flags can change)
        params = [p for p in all_params(net) if p.requires_grad]
        # Every trainable parameter already owns a contiguous fp32 .grad on the device (FlatAdam's flat bucket): the
        # gradients are added there by backward itself, so autograd only has to CALL it - one anchor tensor goes through
        # Function.apply instead of ~270 parameters (0.2 ms of host time per update in front of the first kernel).
        if DIRECT_GRADS and all(p.grad is not None and p.grad.is_cuda and p.grad.is_contiguous() and p.grad.dtype == torch.float32
                                for p in params):
            anchor = net.__dict__.get("_grad_anchor")
            if anchor is None or anchor.device != params[0].device:
                anchor = net.__dict__["_grad_anchor"] = torch.zeros(1, device=params[0].device, requires_grad=True)
            return MapCMAAnchoredFn.apply((net, *args), params, anchor)
        return MapCMAForwardFn.apply((net, *args), *params)


DIRECT_GRADS = True  # (False: parameters through Function.apply as before round 3)


class MapCMAAnchoredFn(torch.autograd.Function):
    """MapCMAForwardFn for the case where every parameter gradient is accumulated by backward itself (see `run`)."""

    @staticmethod
    def forward(ctx, holder, params, anchor):
        net, args = holder[0], holder[1:]
        save: Dict = {}
        with torch.no_grad():
            feats, rnn_out = net.forward_hip(*args, save=save)
        ctx.net, ctx.saves, ctx.params = net, save, params
        ctx.mark_non_differentiable(rnn_out)
        return feats, rnn_out

    @staticmethod
    def backward(ctx, d_feats, _d_rnn):
        with torch.no_grad():
            G = getattr(ctx.net, "backward_hip", None)
            G = (G or (lambda S, d: net_backward(ctx.net, S, d)))(ctx.saves, d_feats)
            direct = []
            for p in ctx.params:
                g = G.get(p)
                if g is None:
                    continue
                if g.is_contiguous() and g.dtype == torch.float32:
                    direct.append((g, p.grad))
                else:
                    p.grad.add_(g.view_as(p))
            if direct:
                ops.add_multi(direct)
        ctx.saves = None
        return None, None, None
