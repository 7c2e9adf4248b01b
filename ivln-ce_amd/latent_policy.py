"""Latent-CMA baseline on HIP: `LatentCMAPolicy` / `LatentCMANet`
(ivlnce_baselines/models/latent_cma_policy.py:28-497) and the `TorchVisionResNet50` RGB encoder
(models/encoders/resnet_encoders.py:118-229) under the same registry name and state_dict keys.

Inference (`act`, `act_iterative`, tour-memory variants) and training (`build_distribution` under autograd:
sequence mode and the tour-memory variant the reference unrolls) run on the HIP kernels of this package: the
RGB ResNet-50 is the BN-folded bottleneck stack already used by RedNet (MFMA convs with fused scale/shift/
residual/ReLU epilogues), the head reuses the MapCMA kernels (instruction bi-LSTM, DD-PPO depth ResNet,
GRU steps and BPTT, attention forward/backward); `backward_hip` is the hand-written backward.
"""
from typing import Tuple

import numpy as np
import torch
import torch.nn as nn
from torch import Tensor

from . import ops
from .aux_losses import AuxLosses
from .encoders import InstructionEncoder, VlnResnetDepthEncoder, build_rnn_state_encoder
from .policy import ILPolicy, Net
from .rednet import Bottleneck, _Folded
from .registry import baseline_registry


class TorchVisionResNet50(nn.Module):
    """resnet_encoders.py:118-229 with `spatial_output=True` (the only way LatentCMANet builds it): torchvision
    ResNet-50 children [conv1, bn1, relu, maxpool, layer1..4] as `cnn` (avgpool -> adaptive 4x4), plus a 64-d
    learned embedding of the 16 positions.  Output (B, 2048 + 64, 4, 4)."""

    def __init__(self, output_size: int, normalize_visual_inputs: bool = False, trainable: bool = False,
                 spatial_output: bool = True, single_spatial_filter: bool = True) -> None:
        super().__init__()
        assert spatial_output, "LatentCMANet uses the spatial output"
        assert not normalize_visual_inputs, "the reference never enables ImageNet normalisation here"
        self.normalize_visual_inputs = normalize_visual_inputs
        self.spatial_output = spatial_output
        self.resnet_layer_size = 2048
        inplanes = [64]

        def make_layer(planes, blocks, stride=1):
            downsample = None
            if stride != 1 or inplanes[0] != planes * 4:
                downsample = nn.Sequential(nn.Conv2d(inplanes[0], planes * 4, 1, stride, bias=False),
                                           nn.BatchNorm2d(planes * 4))
            layers = [Bottleneck(inplanes[0], planes, stride, downsample)]
            inplanes[0] = planes * 4
            layers += [Bottleneck(inplanes[0], planes) for _ in range(1, blocks)]
            return nn.Sequential(*layers)

        self.cnn = _ResNet50Body(
            nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2, 1),
            make_layer(64, 3), make_layer(128, 4, 2), make_layer(256, 6, 2), make_layer(512, 3, 2),
        )
        for param in self.cnn.parameters():
            param.requires_grad_(trainable)
        self.cnn.train(trainable)
        self.spatial_embeddings = nn.Embedding(4 * 4, 64)
        self.cnn.extra_channels = self.spatial_embeddings.embedding_dim
        self.output_shape = (self.resnet_layer_size + self.spatial_embeddings.embedding_dim, 4, 4)

    @property
    def is_blind(self):
        return False

    def forward(self, observations) -> Tensor:
        c, E = self.resnet_layer_size, self.spatial_embeddings.embedding_dim
        if "rgb_features" in observations:
            feats = observations["rgb_features"].to(torch.float32).contiguous()
            B = feats.shape[0]
            out = torch.empty((B, c + E, 4, 4), dtype=torch.float32, device=feats.device)
            ops.copy2d(feats.view(B, -1), out.view(B, -1), B, c * 16)
        else:
            rgb = observations["rgb"]
            if not rgb.is_cuda:
                raise RuntimeError("HIP hot path needs GPU tensors (no CPU fallback)")
            B = rgb.shape[0]
            self.cnn(rgb)  # a module call, so the trainers' forward hook on `.cnn` sees the (B,2048,4,4) features
            out = self.cnn.full_output
        # the (16, 64) table viewed as (1, 64, 4, 4): a reshape of row-major memory, not a transpose
        ops.copy2d(self.spatial_embeddings.weight.view(1, -1), out.view(B, -1)[:, c * 16:], B, E * 16, broadcast_rows=True)
        return out


class _ResNet50Body(nn.Sequential):
    """torchvision ResNet-50 children [conv1, bn1, relu, maxpool, layer1..4] (+ the 4x4 spatial average pool the
    reference swaps in for `avgpool`, resnet_encoders.py:153-163) as parameter holders with a HIP forward.  The
    result is written into the first 2048 channels of a (B, 2048 + extra, 4, 4) buffer - the caller appends the
    spatial embedding without a concat copy - and returned as that view, which is what the reference's
    `rgb_encoder.cnn` forward hook caches as `rgb_features` (dagger_trainer.py:306-314)."""

    extra_channels = 0

    def forward(self, rgb: Tensor) -> Tensor:
        if getattr(self, "_epoch", -1) != ops.WEIGHT_EPOCH:  # folded BatchNorms follow parameter updates / loads
            self._folded, self._epoch = _Folded(), ops.WEIGHT_EPOCH
        f = self._folded
        x = ops.rgb_to_nchw(rgb.to(torch.uint8).contiguous(), 255.0)
        s, b = f.bn(self[1])
        x = ops.conv2d(x, self[0].weight, stride=2, pad=3, scale=s, shift=b, relu=True)
        x = ops.pool2d(x, 3, 2, 1, "max")
        for layer in (self[4], self[5], self[6], self[7]):
            for blk in layer:
                x = blk.forward_hip(x, f)
        c = x.shape[1]
        out = torch.empty((x.shape[0], c + self.extra_channels, 4, 4), dtype=torch.float32, device=x.device)
        ops.adaptive_avgpool2d(x, 4, 4, out=out, out_ctot=c + self.extra_channels)
        self.full_output = out
        return out[:, :c]


class LatentCMANet(Net):
    """latent_cma_policy.py:195-497 (inference): instruction bi-LSTM, DD-PPO depth ResNet, torchvision RGB
    ResNet-50, two GRU state encoders, three attentions, optional cross-episode tour memory."""

    def __init__(self, observation_space, model_config, num_actions):
        super().__init__()
        self.model_config = model_config
        model_config.defrost()
        model_config.INSTRUCTION_ENCODER.final_state_only = False
        model_config.freeze()
        self.instruction_encoder = InstructionEncoder(model_config.INSTRUCTION_ENCODER)
        assert model_config.DEPTH_ENCODER.cnn_type in ["VlnResnetDepthEncoder"], \
            "DEPTH_ENCODER.cnn_type must be VlnResnetDepthEncoder"
        self.depth_encoder = VlnResnetDepthEncoder(
            observation_space, output_size=model_config.DEPTH_ENCODER.output_size,
            checkpoint=model_config.DEPTH_ENCODER.ddppo_checkpoint, backbone=model_config.DEPTH_ENCODER.backbone,
            spatial_output=True,
        )
        assert model_config.RGB_ENCODER.cnn_type in ["TorchVisionResNet50"], \
            "RGB_ENCODER.cnn_type must be TorchVisionResNet50"
        self.rgb_encoder = TorchVisionResNet50(output_size=model_config.RGB_ENCODER.output_size, spatial_output=True)
        self.prev_action_embedding = nn.Embedding(num_actions + 1, 32)
        hidden_size = model_config.STATE_ENCODER.hidden_size
        self._hidden_size = hidden_size
        self.rgb_linear = nn.Sequential(
            nn.AdaptiveAvgPool1d(1), nn.Flatten(),
            nn.Linear(self.rgb_encoder.output_shape[0], model_config.RGB_ENCODER.output_size), nn.ReLU(True),
        )
        self.depth_linear = nn.Sequential(
            nn.Flatten(), nn.Linear(int(np.prod(self.depth_encoder.output_shape)), model_config.DEPTH_ENCODER.output_size),
            nn.ReLU(True),
        )
        rnn_input_size = (model_config.DEPTH_ENCODER.output_size + model_config.RGB_ENCODER.output_size
                          + self.prev_action_embedding.embedding_dim)
        if model_config.tour_memory_variant:
            rnn_input_size += hidden_size
        self.state_encoder = build_rnn_state_encoder(input_size=rnn_input_size, hidden_size=hidden_size,
                                                     rnn_type=model_config.STATE_ENCODER.rnn_type, num_layers=1)
        self._output_size = (hidden_size + model_config.RGB_ENCODER.output_size + model_config.DEPTH_ENCODER.output_size
                             + self.instruction_encoder.output_size)
        self.rgb_kv = nn.Conv1d(self.rgb_encoder.output_shape[0], hidden_size // 2 + model_config.RGB_ENCODER.output_size, 1)
        self.depth_kv = nn.Conv1d(self.depth_encoder.output_shape[0],
                                  hidden_size // 2 + model_config.DEPTH_ENCODER.output_size, 1)
        self.state_q = nn.Linear(hidden_size, hidden_size // 2)
        self.text_k = nn.Conv1d(self.instruction_encoder.output_size, hidden_size // 2, 1)
        self.text_q = nn.Linear(self.instruction_encoder.output_size, hidden_size // 2)
        self.register_buffer("_scale", torch.tensor(1.0 / ((hidden_size // 2) ** 0.5)))
        self._scale_f = float(1.0 / ((hidden_size // 2) ** 0.5))
        self.second_state_compress = nn.Sequential(
            nn.Linear(self._output_size + self.prev_action_embedding.embedding_dim, hidden_size), nn.ReLU(True))
        self.second_state_encoder = build_rnn_state_encoder(input_size=hidden_size, hidden_size=hidden_size,
                                                            rnn_type=model_config.STATE_ENCODER.rnn_type, num_layers=1)
        assert (not model_config.memory_at_end) or model_config.tour_memory_variant, \
            "`memory_at_end` requires `tour_memory_variant`."
        if model_config.memory_at_end:
            self.out_layer = nn.Sequential(nn.Linear(hidden_size * 2, hidden_size), nn.ReLU(True))
        self._output_size = hidden_size
        self.progress_monitor = nn.Linear(self.output_size, 1)
        if model_config.PROGRESS_MONITOR.use:
            nn.init.kaiming_normal_(self.progress_monitor.weight, nonlinearity="tanh")
            nn.init.constant_(self.progress_monitor.bias, 0)
        self.train()

    @property
    def output_size(self):
        return self._output_size

    @property
    def is_blind(self):
        return self.rgb_encoder.is_blind or self.depth_encoder.is_blind

    @property
    def num_recurrent_layers(self):
        return (self.state_encoder.num_recurrent_layers + self.second_state_encoder.num_recurrent_layers
                + int(self.model_config.tour_memory_variant))

    def forward(self, observations, rnn_states, prev_actions, action_masks, episode_masks=None, tour_masks=None):
        """Same signature/return as latent_cma_policy.py:375-497.  rows == envs: one step; rows == T*envs: a
        time-major trajectory batch (every mode, including the tour-memory variant the reference unrolls in Python).
        Under autograd the whole net is one Function whose backward is `backward_hip`."""
        from .train import MapCMAForwardFn, all_params

        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in all_params(self))
        if needs_grad:

            feats, rnn_out = MapCMAForwardFn.run(self, observations, rnn_states, prev_actions, action_masks,
                                                 episode_masks, tour_masks)
        else:
            feats, rnn_out = self.forward_hip(observations, rnn_states, prev_actions, action_masks, episode_masks,
                                              tour_masks)
        if self.model_config.PROGRESS_MONITOR.use and AuxLosses.is_active():
            from .train import progress_monitor_loss

            loss = progress_monitor_loss(self, feats, observations["progress"])
            AuxLosses.register_loss("progress_monitor", loss, self.model_config.PROGRESS_MONITOR.alpha)
        return feats, rnn_out

    def forward_hip(self, observations, rnn_states, prev_actions, action_masks, episode_masks=None, tour_masks=None,
                    save=None):
        mc = self.model_config
        if mc.disable_tour_memory:
            tour_masks = None
        if episode_masks is None:
            episode_masks = action_masks
        if tour_masks is None:
            tour_masks = episode_masks
        dev = rnn_states.device
        H, h2 = self._hidden_size, self._hidden_size // 2
        N = rnn_states.shape[0]
        variant = bool(mc.tour_memory_variant)
        rnn_states = rnn_states.detach().to(torch.float32).contiguous()
        act_u8 = action_masks.reshape(-1).to(torch.uint8).contiguous()
        ep_u8 = episode_masks.reshape(-1).to(torch.uint8).contiguous()
        rows = act_u8.shape[0]
        T = rows // N
        if rows != T * N:
            raise ValueError(f"{rows} rows are not a multiple of {N} recurrent states")
        prev_actions = prev_actions.reshape(-1).long().contiguous()

        s_txt = {} if save is not None else None
        from . import train as _train

        # training pass: the instruction bi-LSTM (latency-bound) runs on a side stream beside the encoders' feature
        # plumbing and the 2112-channel k/v projection below (train.OVERLAP_INSTRUCTION)
        overlap = (save is not None and _train.OVERLAP_INSTRUCTION and not mc.ablate_instruction
                   and not torch.cuda.is_current_stream_capturing())
        if overlap:
            cur, st = torch.cuda.current_stream(), _train.side_stream(dev)
            st.wait_stream(cur)
            _train.share_with_stream(observations.get("instruction"), st)
            with torch.cuda.stream(st):
                txt, lengths = self.instruction_encoder(observations, s_txt)
                tk = ops.conv2d(txt.view(txt.shape[0], -1, 1, txt.shape[2]), self.text_k.weight.view(h2, -1, 1, 1),
                                shift=self.text_k.bias, splitk=False)
        else:
            txt, lengths = self.instruction_encoder(observations, s_txt)  # (rows, 256, L), zero beyond each length
            tk = None
        dep = self.depth_encoder(observations)                         # (rows, 192, 4, 4)
        rgb = self.rgb_encoder(observations)                           # (rows, 2112, 4, 4)
        if mc.ablate_instruction:
            txt = torch.zeros_like(txt)
        if mc.ablate_depth:
            dep = torch.zeros_like(dep)
        if mc.ablate_rgb:
            rgb = torch.zeros_like(rgb)
        L = txt.shape[2]
        P = dep.shape[2] * dep.shape[3]
        Cr, Cd = rgb.shape[1], dep.shape[1]
        r_out, d_out = self.rgb_linear[2].out_features, self.depth_linear[1].out_features
        E = self.prev_action_embedding.embedding_dim

        # state_in = [rgb_in | depth_in | prev (| tour memory)];  x2 = [state | text | rgb' | depth' | prev]
        base = r_out + d_out + E
        state_in = torch.empty((rows, base + (H if variant else 0)), dtype=torch.float32, device=dev)
        x2 = torch.empty((rows, H + 256 + r_out + d_out + E), dtype=torch.float32, device=dev)
        o_txt, o_rgb, o_dep, o_prev = H, H + 256, H + 256 + r_out, H + 256 + r_out + d_out
        ops.prev_action_embed(prev_actions, act_u8, self.prev_action_embedding.weight, state_in[:, r_out + d_out:base],
                              x2[:, o_prev:])
        rgb_mean = ops.pool2d(rgb, 4, 4, 0, "avg").view(rows, Cr)  # AdaptiveAvgPool1d(1) over the 16 positions
        ops.linear(rgb_mean, self.rgb_linear[2].weight, self.rgb_linear[2].bias, relu=True, out=state_in[:, :r_out])
        ops.linear(dep.view(rows, -1), self.depth_linear[1].weight, self.depth_linear[1].bias, relu=True,
                   out=state_in[:, r_out:r_out + d_out])

        rnn_out = rnn_states.clone()
        state = x2[:, :H]
        s_g1 = {} if save is not None else None
        s_g2 = {} if save is not None else None
        mem_in = None
        if variant:
            # the tour-long slot: cleared where a tour starts, fed to the first GRU, then max-pooled with that
            # GRU's new state (latent_cma_policy.py:395-399,422-425,433-439).  It carries no gradient in the
            # reference (updated under no_grad from a detached clone), so it is a constant input here too; the
            # first GRU reads its own previous output through it, hence the per-step loop for this one layer.
            rnn = self.state_encoder.rnn
            tour_u8 = tour_masks.reshape(-1).to(torch.uint8).contiguous()
            mem_in = torch.empty((rows, H), dtype=torch.float32, device=dev)
            saves = None
            if save is not None:
                saves = tuple(torch.empty((rows, H), dtype=torch.float32, device=dev) for _ in range(4))
                s_g1.update(r=saves[0], z=saves[1], n=saves[2], ghn=saves[3], T=T, N=N, x=state_in,
                            h0=rnn_states[:, 0], masks=ep_u8, out=state)
            for t in range(T):
                sl, sp = slice(t * N, (t + 1) * N), slice((t - 1) * N, t * N)
                # memory entering step t = tour mask * max(memory entering t-1, first GRU's state at t-1)
                ops.tour_memory(rnn_states[:, 2] if t == 0 else mem_in[sp], None if t == 0 else state[sp], tour_u8[sl],
                                mem_in[sl], state_in[sl, base:])
                h_in = rnn_states[:, 0] if t == 0 else state[sp]
                ops.gru_step(state_in[sl], None, h_in, ep_u8[sl], rnn.weight_ih_l0, rnn.weight_hh_l0, rnn.bias_ih_l0,
                             rnn.bias_hh_l0, state[sl], rnn_out[:, 0] if t == T - 1 else None,
                             tuple(x[sl] for x in saves) if saves is not None else None)
            ops.tour_memory(mem_in[(T - 1) * N:], state[(T - 1) * N:], None, rnn_out[:, 2])
        else:
            self.state_encoder(state_in, rnn_states[:, 0], ep_u8, state, rnn_out[:, 0], s_g1)

        # the key/value projections of the visual features do not depend on the recurrent state: issued before
        # the join with the instruction branch
        rkv = ops.conv2d(rgb.view(rows, Cr, 1, P), self.rgb_kv.weight.view(-1, Cr, 1, 1), shift=self.rgb_kv.bias).view(rows, -1, P)
        dkv = ops.conv2d(dep.view(rows, Cd, 1, P), self.depth_kv.weight.view(-1, Cd, 1, 1), shift=self.depth_kv.bias).view(rows, -1, P)
        if overlap:
            cur.wait_stream(st)
            _train.share_with_stream((txt, lengths, tk, s_txt), cur)
        q1 = ops.linear(state, self.state_q.weight, self.state_q.bias)
        if tk is None:
            tk = ops.conv2d(txt.view(rows, -1, 1, L), self.text_k.weight.view(h2, -1, 1, 1), shift=self.text_k.bias, splitk=False)
        text = x2[:, o_txt:o_txt + 256]
        a_txt = torch.empty((rows, L), dtype=torch.float32, device=dev) if save is not None else None
        ops.attn(q1, tk.view(rows, h2, L), txt, lengths, self._scale_f, text, a_txt)
        q2 = ops.linear(text, self.text_q.weight, self.text_q.bias)
        a_rgb = a_dep = None
        if save is None and P <= 32:
            ops.attn_small2(q2, rkv[:, :h2], rkv[:, h2:], x2[:, o_rgb:o_rgb + r_out], dkv[:, :h2], dkv[:, h2:],
                            x2[:, o_dep:o_dep + d_out], self._scale_f)
        else:
            a_rgb = torch.empty((rows, P), dtype=torch.float32, device=dev) if save is not None else None
            a_dep = torch.empty((rows, P), dtype=torch.float32, device=dev) if save is not None else None
            ops.attn(q2, rkv[:, :h2], rkv[:, h2:], None, self._scale_f, x2[:, o_rgb:o_rgb + r_out], a_rgb)
            ops.attn(q2, dkv[:, :h2], dkv[:, h2:], None, self._scale_f, x2[:, o_dep:o_dep + d_out], a_dep)
        sc = self.second_state_compress[0]
        c2 = ops.linear(x2, sc.weight, sc.bias, relu=True)
        g2_out = torch.empty((rows, H), dtype=torch.float32, device=dev)
        self.second_state_encoder(c2, rnn_states[:, 1], ep_u8, g2_out, rnn_out[:, 1], s_g2)
        feats, cat = g2_out, None
        if mc.memory_at_end:
            ol = self.out_layer[0]
            cat = torch.cat([g2_out, mem_in], dim=1)
            feats = ops.linear(cat, ol.weight, ol.bias, relu=True)
        if save is not None:
            save.update(txt=s_txt, g1=s_g1, g2=s_g2, dep=dep, rgb=rgb, rgb_mean=rgb_mean, txt_out=txt, state_in=state_in,
                        x2=x2, q1=q1, tk=tk, a_txt=a_txt, rkv=rkv, dkv=dkv, q2=q2, a_rgb=a_rgb, a_dep=a_dep, c2=c2,
                        cat=cat, feats=feats, rows=rows, N=N, L=L, P=P, offs=(o_txt, o_rgb, o_dep, o_prev),
                        act_masks=act_u8, prev_actions=prev_actions)
        return feats, rnn_out

    def backward_hip(self, S, d_feats):
        """Gradients of every trainable parameter given d(loss)/d(features): the hand-written HIP backward of
        `forward_hip` (the frozen RGB / depth ResNets are not traversed; their learned spatial embeddings are)."""
        from . import train as _train
        from .train import _conv1d_backward, _gru_backward, instruction_backward

        G = {}
        mc = self.model_config
        rows, L, P = S["rows"], S["L"], S["P"]
        H, h2 = self._hidden_size, self._hidden_size // 2
        scale = self._scale_f
        o_txt, o_rgb, o_dep, o_prev = S["offs"]
        x2, state_in = S["x2"], S["state_in"]
        dev = d_feats.device
        r_out, d_out = self.rgb_linear[2].out_features, self.depth_linear[1].out_features
        E = self.prev_action_embedding.embedding_dim
        d_feats = d_feats.contiguous()

        if mc.memory_at_end:  # out_layer over [second GRU output | tour memory]; the memory half is a constant
            ol = self.out_layer[0]
            d_pre = ops.relu_bwd(d_feats, S["feats"])
            G[ol.weight] = ops.linear_bwd_weight(d_pre, S["cat"])
            G[ol.bias] = ops.colsum(d_pre)
            d_feats = ops.linear_bwd_input(d_pre, ol.weight)[:, :H]

        d_c2 = _gru_backward(self.second_state_encoder, S["g2"], d_feats, G)
        d_pre = ops.relu_bwd(d_c2, S["c2"])
        sc = self.second_state_compress[0]
        G[sc.weight] = ops.linear_bwd_weight(d_pre, x2)
        G[sc.bias] = ops.colsum(d_pre)
        dx2 = ops.linear_bwd_input(d_pre, sc.weight)  # [state | text | rgb' | depth' | prev]

        rkv, dkv = S["rkv"], S["dkv"]
        d_rkv, d_dkv = torch.empty_like(rkv), torch.empty_like(dkv)
        dq2_r = torch.empty((rows, h2), dtype=torch.float32, device=dev)
        dq2_d = torch.empty((rows, h2), dtype=torch.float32, device=dev)
        ops.attn_bwd(dx2[:, o_rgb:o_rgb + r_out], S["a_rgb"], S["q2"], rkv[:, :h2], rkv[:, h2:], scale, dq2_r,
                     d_rkv[:, :h2], d_rkv[:, h2:])
        ops.attn_bwd(dx2[:, o_dep:o_dep + d_out], S["a_dep"], S["q2"], dkv[:, :h2], dkv[:, h2:], scale, dq2_d,
                     d_dkv[:, :h2], d_dkv[:, h2:])
        dq2 = ops.add2d(dq2_r, dq2_d)
        text = x2[:, o_txt:o_txt + 256]
        G[self.text_q.weight] = ops.linear_bwd_weight(dq2, text)
        G[self.text_q.bias] = ops.colsum(dq2)
        d_text = dx2[:, o_txt:o_txt + 256]
        ops.linear_bwd_input(dq2, self.text_q.weight, out=d_text, accumulate=True)

        txt, tk = S["txt_out"], S["tk"]
        dq1 = torch.empty((rows, h2), dtype=torch.float32, device=dev)
        d_tk = torch.empty((rows, h2, L), dtype=torch.float32, device=dev)
        d_txt = torch.empty_like(txt)
        ops.attn_bwd(d_text, S["a_txt"], S["q1"], tk.view(rows, h2, L), txt, scale, dq1, d_tk, d_txt)
        d_txt = _conv1d_backward(self.text_k, d_tk.view(rows, h2, 1, L), txt.view(rows, -1, 1, L),
                                 d_txt.view(rows, -1, 1, L), G).view(rows, -1, L)
        G_txt, side, main = None, None, torch.cuda.current_stream()
        if _train.OVERLAP_INSTRUCTION and not torch.cuda.is_current_stream_capturing():
            side, G_txt = _train.side_stream(dev), {}
            side.wait_stream(main)
            _train.share_with_stream((d_txt, S["txt"]), side)
            with torch.cuda.stream(side):
                instruction_backward(self.instruction_encoder, S["txt"], d_txt, rows, L, G_txt)
        state = x2[:, :H]
        G[self.state_q.weight] = ops.linear_bwd_weight(dq1, state)
        G[self.state_q.bias] = ops.colsum(dq1)
        d_state = dx2[:, :H]
        ops.linear_bwd_input(dq1, self.state_q.weight, out=d_state, accumulate=True)

        d_state_in = _gru_backward(self.state_encoder, S["g1"], d_state, G)  # (rows, 288 [+ 512 memory: unused])
        emb = self.prev_action_embedding
        G[emb.weight] = ops.prev_action_embed_bwd(S["prev_actions"], S["act_masks"],
                                                  d_state_in[:, r_out + d_out:r_out + d_out + E], dx2[:, o_prev:],
                                                  emb.num_embeddings)

        # ---- rgb branch: mean over positions -> rgb_linear; rgb_kv.  Only the 64 learned spatial-embedding
        # channels of the (frozen) encoder output need an input gradient.
        rgb, dep = S["rgb"], S["dep"]
        Cr, Cd = rgb.shape[1], dep.shape[1]
        rl, dl = self.rgb_linear[2], self.depth_linear[1]
        se_r, se_d = self.rgb_encoder.spatial_embeddings, self.depth_encoder.spatial_embeddings
        Er, Ed = se_r.embedding_dim, se_d.embedding_dim
        d_pre_r = ops.relu_bwd(d_state_in[:, :r_out], state_in[:, :r_out])
        G[rl.weight] = ops.linear_bwd_weight(d_pre_r, S["rgb_mean"])
        G[rl.bias] = ops.colsum(d_pre_r)
        kvw = self.rgb_kv.weight.view(-1, Cr)
        G[self.rgb_kv.weight] = ops.conv2d_bwd_weight(d_rkv.view(rows, -1, 1, P), rgb.view(rows, Cr, 1, P), 1, 1).view_as(
            self.rgb_kv.weight)
        G[self.rgb_kv.bias] = ops.nchw_chansum(d_rkv.view(rows, -1, 1, P))
        w_se = ops.transpose(kvw[:, Cr - Er:].contiguous())  # (Er, O)
        d_se = ops.conv2d(d_rkv.view(rows, -1, 1, P), w_se.view(Er, -1, 1, 1), weight_is_temp=True).view(rows, Er, P)
        g_se = ops.colsum(d_se.view(rows, -1)).view(Er, P)
        d_mean = ops.colsum(ops.linear_bwd_input(d_pre_r, rl.weight[:, Cr - Er:].contiguous()))  # (Er,)
        G[se_r.weight] = (g_se + d_mean.view(Er, 1) / P).reshape(se_r.weight.shape)

        # ---- depth branch: depth_linear + depth_kv -> spatial embedding
        d_pre_d = ops.relu_bwd(d_state_in[:, r_out:r_out + d_out], state_in[:, r_out:r_out + d_out])
        G[dl.weight] = ops.linear_bwd_weight(d_pre_d, dep.view(rows, -1))
        G[dl.bias] = ops.colsum(d_pre_d)
        d_dep = ops.linear_bwd_input(d_pre_d, dl.weight)
        d_dep = _conv1d_backward(self.depth_kv, d_dkv.view(rows, -1, 1, P), dep.view(rows, Cd, 1, P),
                                 d_dep.view(rows, Cd, 1, P), G)
        G[se_d.weight] = ops.colsum(d_dep.view(rows, -1)[:, (Cd - Ed) * P:]).view_as(se_d.weight)

        if G_txt is None:
            instruction_backward(self.instruction_encoder, S["txt"], d_txt, rows, L, G)
        else:
            main.wait_stream(side)
            _train.share_with_stream(G_txt, main)
            G.update(G_txt)
        return G


@baseline_registry.register_policy
class LatentCMAPolicy(ILPolicy):
    """latent_cma_policy.py:28-193: memory options - `tour_memory`: the RNN states reset with the TOUR, not the
    episode; `tour_memory_variant`: episodic states plus a third, tour-long memory slot."""

    def __init__(self, observation_space, action_space, model_config):
        self.tour_memory = model_config.tour_memory
        self.tour_memory_variant = model_config.tour_memory_variant
        self.train_unrolled = model_config.train_unrolled
        super().__init__(
            LatentCMANet(observation_space=observation_space, model_config=model_config, num_actions=action_space.n),
            action_space.n,
        )

    def act_iterative(self, observations, rnn_hidden_states, prev_actions, agent_episode_not_done_masks,
                      sim_episode_not_done_masks, tour_not_done_masks, action_masks, deterministic=False):
        if self.tour_memory_variant:
            episode_masks, tour_masks = agent_episode_not_done_masks, tour_not_done_masks
        else:
            episode_masks = tour_not_done_masks if self.tour_memory else None
            tour_masks = None
        features, rnn_hidden_states = self.net(
            observations, rnn_hidden_states, prev_actions, action_masks=agent_episode_not_done_masks,
            episode_masks=episode_masks, tour_masks=tour_masks,
        )
        return self._act(features, deterministic, observations), rnn_hidden_states

    def build_distribution(self, observations, rnn_hidden_states, prev_actions, agent_episode_not_done_masks,
                           tour_not_done_masks=None) -> Tuple:
        features, rnn_hidden_states = self.build_features(observations, rnn_hidden_states, prev_actions,
                                                          agent_episode_not_done_masks, tour_not_done_masks)
        return self.action_distribution(features), rnn_hidden_states

    def build_features(self, observations, rnn_hidden_states, prev_actions, agent_episode_not_done_masks,
                       tour_not_done_masks=None) -> Tuple:
        """latent_cma_policy.py:124-179.  The reference unrolls the time axis in Python for the tour-memory
        variant (`_view_sequential_inputs`); the HIP net takes the time-major batch directly in every mode - the
        masks select which memory resets where."""
        if tour_not_done_masks is None:
            tour_not_done_masks = agent_episode_not_done_masks.clone()
        if self.tour_memory_variant or self.train_unrolled:
            episode_masks, tour_masks = agent_episode_not_done_masks, tour_not_done_masks
        else:
            episode_masks = tour_not_done_masks if self.tour_memory else None
            tour_masks = None
        return self.net(observations, rnn_hidden_states, prev_actions, action_masks=agent_episode_not_done_masks,
                        episode_masks=episode_masks, tour_masks=tour_masks)

    @classmethod
    def from_config(cls, config, observation_space, action_space):
        config.defrost()
        config.MODEL.TORCH_GPU_ID = config.TORCH_GPU_ID
        config.freeze()
        return cls(observation_space=observation_space, action_space=action_space, model_config=config.MODEL)
