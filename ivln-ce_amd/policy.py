"""MapCMA policy plugin: same registry name, constructor, `act` / `act_iterative` /
`build_distribution` surface and `state_dict()` keys as the reference
(ivlnce_baselines/models/map_cma_policy.py:28-368, models/policy.py:12-83,
common/utils.py:149-185), forward AND backward on HIP kernels.

torch.nn modules hold parameters only.  The whole net forward is one torch.autograd.Function whose
backward runs the hand-written HIP backward kernels (train.py), so `loss.backward()` in a trainer
(base_il_trainer.py:211) works unchanged.
"""
from typing import Tuple

import numpy as np
import torch
import torch.nn as nn
from torch import Tensor

from . import ops
from .aux_losses import AuxLosses
from .encoders import InstructionEncoder, SemanticMapEncoder, VlnResnetDepthEncoder, build_rnn_state_encoder
from .registry import baseline_registry

try:  # pragma: no cover
    from habitat_baselines.rl.ppo.policy import Net, Policy  # type: ignore
except Exception:  # noqa: BLE001

    class Net(nn.Module):
        pass

    class Policy(nn.Module):
        pass


class CustomFixedCategorical(torch.distributions.Categorical):
    """common/utils.py:149-169."""

    def sample(self, sample_shape=torch.Size()):  # noqa: B008
        return super().sample(sample_shape).unsqueeze(-1)

    def log_prob(self, actions: Tensor) -> Tensor:
        return super().log_prob(actions.squeeze(-1)).view(actions.size(0), -1).sum(-1).unsqueeze(-1)

    def mode(self):
        return self.probs.argmax(dim=-1, keepdim=True)


class CategoricalNet(nn.Module):
    """common/utils.py:172-185; the 512->4 linear runs on the HIP skinny/MFMA linear."""

    def __init__(self, num_inputs: int, num_outputs: int) -> None:
        super().__init__()
        self.linear = nn.Linear(num_inputs, num_outputs)
        self.num_outputs = num_outputs
        nn.init.orthogonal_(self.linear.weight, gain=0.01)
        nn.init.constant_(self.linear.bias, 0)

    def raw_logits(self, x: Tensor) -> Tensor:
        if x.shape[-1] != self.num_outputs:
            if x.requires_grad or (torch.is_grad_enabled() and self.linear.weight.requires_grad):
                from .train import LinearFn

                x = LinearFn.apply(x, self.linear.weight, self.linear.bias)
            else:
                x = ops.linear(x.contiguous(), self.linear.weight, self.linear.bias)
        return x

    def forward(self, x: Tensor) -> CustomFixedCategorical:
        return CustomFixedCategorical(logits=self.raw_logits(x))


class MapCMANet(Net):
    """Cross-modal attention network: instruction bi-LSTM, DD-PPO depth ResNet, semantic-map CNN,
    two GRU state encoders and three attentions (map_cma_policy.py:103-368)."""

    def __init__(self, observation_space, config, num_actions):
        super().__init__()
        model_config = config.MODEL
        self.model_config = model_config
        model_config.defrost()
        model_config.INSTRUCTION_ENCODER.final_state_only = False
        model_config.freeze()

        assert model_config.SEMANTIC_MAP_ENCODER.classname in [
            "SemanticMapEncoder"
        ], "SEMANTIC_MAP_ENCODER.classname must be SemanticMapEncoder"
        sm = model_config.SEMANTIC_MAP_ENCODER
        self.map_encoder = SemanticMapEncoder(
            observation_space, sm.num_semantic_classes, sm.channels, sm.last_ch_mult, sm.trainable, sm.from_pretrained,
            sm.checkpoint,
        )
        self.instruction_encoder = InstructionEncoder(model_config.INSTRUCTION_ENCODER)
        assert model_config.DEPTH_ENCODER.cnn_type in [
            "VlnResnetDepthEncoder"
        ], "DEPTH_ENCODER.cnn_type must be VlnResnetDepthEncoder"
        self.depth_encoder = VlnResnetDepthEncoder(
            observation_space,
            output_size=model_config.DEPTH_ENCODER.output_size,
            checkpoint=model_config.DEPTH_ENCODER.ddppo_checkpoint,
            backbone=model_config.DEPTH_ENCODER.backbone,
            spatial_output=True,
        )
        self.prev_action_embedding = nn.Embedding(num_actions + 1, 32)
        hidden_size = model_config.STATE_ENCODER.hidden_size
        self._hidden_size = hidden_size
        self.depth_linear = nn.Sequential(
            nn.Flatten(),
            nn.Linear(int(np.prod(self.depth_encoder.output_shape)), model_config.DEPTH_ENCODER.output_size),
            nn.ReLU(True),
        )
        self.map_linear = nn.Sequential(
            nn.Flatten(),
            nn.Linear(int(np.prod(self.map_encoder.output_shape)), model_config.SEMANTIC_MAP_ENCODER.output_size),
            nn.ReLU(True),
        )
        rnn_input_size = (
            model_config.DEPTH_ENCODER.output_size
            + model_config.SEMANTIC_MAP_ENCODER.output_size
            + self.prev_action_embedding.embedding_dim
        )
        self.state_encoder = build_rnn_state_encoder(
            input_size=rnn_input_size, hidden_size=hidden_size, rnn_type=model_config.STATE_ENCODER.rnn_type, num_layers=1
        )
        self._output_size = (
            hidden_size
            + model_config.DEPTH_ENCODER.output_size
            + self.instruction_encoder.output_size
            + model_config.SEMANTIC_MAP_ENCODER.output_size
        )
        self.dep_kv = nn.Conv1d(
            self.depth_encoder.output_shape[0], hidden_size // 2 + model_config.DEPTH_ENCODER.output_size, 1
        )
        self.map_kv = nn.Conv1d(
            self.map_encoder.output_shape[0], hidden_size // 2 + model_config.SEMANTIC_MAP_ENCODER.output_size, 1
        )
        self.state_q = nn.Linear(hidden_size, hidden_size // 2)
        self.text_k = nn.Conv1d(self.instruction_encoder.output_size, hidden_size // 2, 1)
        self.text_q = nn.Linear(self.instruction_encoder.output_size, hidden_size // 2)
        self.register_buffer("_scale", torch.tensor(1.0 / ((hidden_size // 2) ** 0.5)))
        self.second_state_compress = nn.Sequential(
            nn.Linear(self._output_size + self.prev_action_embedding.embedding_dim, self._hidden_size), nn.ReLU(True)
        )
        self.second_state_encoder = build_rnn_state_encoder(
            input_size=self._hidden_size, hidden_size=self._hidden_size,
            rnn_type=model_config.STATE_ENCODER.rnn_type, num_layers=1,
        )
        self._output_size = model_config.STATE_ENCODER.hidden_size
        self.progress_monitor = nn.Linear(self.output_size, 1)
        self._init_layers()
        self.train()
        if not model_config.SEMANTIC_MAP_ENCODER.trainable:
            self.map_encoder.eval()
        self._scale_f = float(1.0 / ((hidden_size // 2) ** 0.5))

    @property
    def output_size(self):
        return self._output_size

    @property
    def is_blind(self):
        return self.depth_encoder.is_blind

    @property
    def num_recurrent_layers(self):
        return self.state_encoder.num_recurrent_layers + self.second_state_encoder.num_recurrent_layers

    def _cma_fold_weights(self):
        """Instruction-side folds of the fused rollout head (csrc/cma_step.hip), cached until the weights change:
        rows 0..H-1 = W_q^T W_k, row H = b_q^T W_k (shift: the same rows applied to b_k), then W_tq / b_tq - one
        (H + 1 + h2) x 256 weight for a single 1x1 conv over the instruction encoder's output."""
        ps = (self.state_q.weight, self.state_q.bias, self.text_k.weight, self.text_k.bias, self.text_q.weight,
              self.text_q.bias)
        key = (ops.WEIGHT_EPOCH,) + tuple(p._version for p in ps) + tuple(p.data_ptr() for p in ps)
        if getattr(self, "_cma_fold_key", None) != key:
            with torch.no_grad():
                wq, bq, wk, bk, wtq, btq = [p.detach() for p in ps]
                a = torch.cat([ops.transpose(wq.contiguous()), bq.view(1, -1)], 0).contiguous()  # (H+1, h2)
                wk2 = torch.cat([wk.view(wk.shape[0], -1), bk.view(-1, 1)], 1).contiguous()       # (h2, Ct + 1)
                y = ops.linear_gemm(a, ops.transpose(wk2))                                        # (H+1, Ct + 1)
                ct = wk.shape[1]
                wf = torch.cat([y[:, :ct], wtq.view(wtq.shape[0], -1)], 0).contiguous()
                bf = torch.cat([y[:, ct], btq], 0).contiguous()
            self._cma_fold_key, self._cma_fold = key, (wf, bf)
        return self._cma_fold

    def prepare_capture(self, example_obs):
        """Create everything the rollout step caches lazily BEFORE a stream capture (graphed.py): a buffer born inside
        a capture lives in that graph's private pool and must not survive in a process-wide cache."""
        if ops.CMA_STEP_MODE < 0 or "instruction" not in example_obs:
            return
        rows, L = example_obs["instruction"].shape[0], example_obs["instruction"].shape[1]
        P = self.depth_encoder.output_shape[1] * self.depth_encoder.output_shape[2]
        if P > 16 or L > 512:
            return  # outside the fused head's envelope: forward_hip takes the unfused chain
        self._cma_fold_weights()
        ops.cma_step_ws(rows, L, P, self._hidden_size, example_obs["instruction"].device)

    def _init_layers(self):
        if self.model_config.PROGRESS_MONITOR.use:
            nn.init.kaiming_normal_(self.progress_monitor.weight, nonlinearity="tanh")
            nn.init.constant_(self.progress_monitor.bias, 0)

    # ------------------------------------------------------------------------------------------
    def forward_hip(self, observations, rnn_states, prev_actions, action_masks, save=None):
        """HIP forward of map_cma_policy.py:276-353.  Returns (features (rows,512), rnn_states_out
        (N,2,512)).  `save` (dict) collects what the HIP backward needs."""
        mc = self.model_config
        dev = rnn_states.device
        H = self._hidden_size
        h2 = H // 2
        N = rnn_states.shape[0]
        rnn_states = rnn_states.to(torch.float32).contiguous()
        masks_u8 = action_masks.reshape(-1).to(torch.uint8).contiguous()
        rows = masks_u8.shape[0]
        prev_actions = prev_actions.reshape(-1).long().contiguous()

        s_txt = {} if save is not None else None
        s_map = [] if save is not None else None
        d_out = self.depth_linear[1].out_features
        m_out = self.map_linear[1].out_features
        E = self.prev_action_embedding.embedding_dim
        # state_in = [dep_in | map_in | prev]; x2 = [state | text | dep' | map' | prev]
        x2w = H + self.instruction_encoder.output_size + d_out + m_out + E
        persist = getattr(self, "_persist", None) if save is None else None  # graphed.py split mode
        if persist is not None and persist["state_in"].shape[0] == rows:
            state_in, x2 = persist["state_in"], persist["x2"]
        else:
            state_in = torch.empty((rows, d_out + m_out + E), dtype=torch.float32, device=dev)
            x2 = torch.empty((rows, x2w), dtype=torch.float32, device=dev)
        dl, ml = self.depth_linear[1], self.map_linear[1]
        o_txt, o_dep, o_map, o_prev = H, H + 256, H + 256 + d_out, H + 256 + d_out + m_out

        # The fused recurrent head (ivln_cma_step_fwd) wants the instruction branch to emit the folded [Mq | TQb] operand
        # instead of text_k, so its eligibility is decided HERE, from shapes known before any branch runs (the encoders'
        # static output shapes, the token axis, the widths' divisibility rules of csrc/cma_step.hip) - anything
        # outside the kernel's envelope takes the unfused chain with a real text_k.
        P_static = self.depth_encoder.output_shape[1] * self.depth_encoder.output_shape[2]
        L_static = observations["instruction"].shape[-1] if "instruction" in observations else 0
        fused_head = (save is None and ops.CMA_STEP_MODE >= 0
                      and not (mc.ablate_instruction or mc.ablate_depth or mc.ablate_map)
                      and P_static <= 16 and 0 < L_static <= 512
                      and self.map_encoder.output_shape[1] * self.map_encoder.output_shape[2] == P_static
                      and H % 64 == 0 and h2 % 64 == 0 and d_out % 16 == 0 and m_out % 16 == 0
                      and self.instruction_encoder.output_size % 64 == 0)

        # update batches: the instruction of a trajectory is the same at every timestep, so the loader hands over the
        # UNIQUE token rows and each row's index into them (trainers.PrefetchLoader); the encoder, text_k and their
        # backward then run on U sequences instead of T*N (8-9 instead of 512 at the benched shape)
        inv = None
        if save is not None and "instruction_index" in observations and "instruction_unique" in observations:
            inv = observations["instruction_index"].reshape(-1).to(torch.int32).contiguous()

        def _txt_branch(sv):
            src = observations if inv is None else {"instruction": observations["instruction_unique"]}
            t, ln = self.instruction_encoder(src, sv)  # (rows | U, 256, L)
            if mc.ablate_instruction:
                t = torch.zeros_like(t)
            r_, L_ = t.shape[0], t.shape[2]
            if fused_head:
                # rollout: text_k, state_q and text_q folded over the instruction (csrc/cma_step.hip): ONE 1x1 conv
                # yields Mq (H+1 channels) and TQb (h2 channels) for the fused head, no text_k tensor is made
                wf, bf = self._cma_fold_weights()
                cache = self.instruction_encoder.last_cache if sv is None else None
                if cache is not None and getattr(cache, "fold_behind", False):
                    # an earlier cached call re-encoded rows without folding them (the unfused head, or a capture that had
                    # to drop the cache): `dirty` only names THIS call's rows, so every row is folded again
                    if torch.cuda.is_current_stream_capturing():
                        cache = None
                    else:
                        cache.dirty.fill_(1)
                        cache.fold_behind = False
                if cache is not None:
                    # per-episode cache (encoders.InstructionEncoder.step_cache): the folded operands live in the cache's
                    # persistent buffer and only the rows the encoder just re-encoded (cache.dirty) are recomputed
                    fkey = getattr(self, "_cma_fold_key", None)
                    if cache.fold is None or cache.fold_key != fkey:
                        if torch.cuda.is_current_stream_capturing():
                            cache.fold_behind = True  # (the replays advance the cache's tokens without its fold)
                            cache = None  # (cannot be created / invalidated inside a capture: plain conv below)
                        else:
                            if cache.fold is None:
                                cache.fold = torch.zeros((r_, wf.shape[0], 1, L_), dtype=torch.float32, device=t.device)
                            cache.fold_key = fkey
                            cache.dirty.fill_(1)  # (folded for other weights, or never: every row's operands are made now)
                if cache is not None:
                    ops.conv2d(t.view(r_, -1, 1, L_), wf.view(wf.shape[0], -1, 1, 1), shift=bf, splitk=False, out=cache.fold,
                               run_flags=cache.dirty)
                    return t, ln, cache.fold.view(r_, -1, L_)
                # (no cache: every row is "dirty" - the same kernel as the cached call, so that the two agree bit for bit
                #  whatever the tile heuristics make of a plain 1x1 conv of this shape)
                fold = ops.conv2d(t.view(r_, -1, 1, L_), wf.view(wf.shape[0], -1, 1, 1), shift=bf, splitk=False,
                                  run_flags=ops.all_rows_flags(r_, t.device))
                return t, ln, fold.view(r_, -1, L_)
            if sv is None and self.instruction_encoder.last_cache is not None:
                self.instruction_encoder.last_cache.fold_behind = True  # (cached encoding, no fold made for its dirty rows)
            tk_ = ops.conv2d(t.view(r_, -1, 1, L_), self.text_k.weight.view(h2, -1, 1, 1), shift=self.text_k.bias,
                             splitk=False)
            return t, ln, tk_

        def _map_branch(sv):
            m = self.map_encoder(observations, sv)  # (rows,128,4,4)
            if mc.ablate_map:
                m = torch.zeros_like(m)
            r_, Cm_, P_ = m.shape[0], m.shape[1], m.shape[2] * m.shape[3]
            # k/v projection + map_linear: one launch for rollout batches, else the MFMA GEMM + the skinny linear
            kv = ops.kv_linear(m, self.map_kv.weight, self.map_kv.bias, ml.weight, ml.bias, state_in[:, d_out:d_out + m_out])
            if kv is None:
                kv = ops.conv2d(m.view(r_, Cm_, 1, P_), self.map_kv.weight.view(-1, Cm_, 1, 1), shift=self.map_kv.bias,
                                splitk=False)
                ops.linear(m.view(r_, -1), ml.weight, ml.bias, relu=True, out=state_in[:, d_out:d_out + m_out])
            return m, kv

        def _dep_branch():
            d = self.depth_encoder(observations)  # (rows,192,4,4)
            if mc.ablate_depth:
                d = torch.zeros_like(d)
            r_, Cd_, P_ = d.shape[0], d.shape[1], d.shape[2] * d.shape[3]
            kv = ops.kv_linear(d, self.dep_kv.weight, self.dep_kv.bias, dl.weight, dl.bias, state_in[:, :d_out])
            if kv is None:
                kv = ops.conv2d(d.view(r_, Cd_, 1, P_), self.dep_kv.weight.view(-1, Cd_, 1, 1), shift=self.dep_kv.bias,
                                splitk=False)
                ops.linear(d.view(r_, -1), dl.weight, dl.bias, relu=True, out=state_in[:, :d_out])
            return d, kv

        # The key/value projections depend only on their own encoder, so they run inside the branches.
        side = getattr(self, "_side_streams", None) if save is None else None
        stage = getattr(self, "_stage", None) if save is None else None
        if stage == "dep":
            # graphed.py, split mode: depth ResNet + its k/v projection + depth_linear as one graph on a side
            # stream (they depend on nothing but the new depth image); results land in the persistent buffers
            if getattr(self, "_txt_with_dep", False) == "first":
                # 8 images: the persistent depth encoder leaves no XCD for the bi-LSTM, which then runs in front of it
                self._stash_txt = _txt_branch(None)
            self._stash_dep = _dep_branch()
            if getattr(self, "_txt_with_dep", False) is True:
                # predicted semantics: the main stream is RedNet's critical path, the instruction encoder moves here
                self._stash_txt = _txt_branch(None)
            return None, None
        if stage == "pre":
            # ... while the instruction and map branches (and the previous-action embedding) run as a second
            # graph on the main stream
            # (order: when the persistent depth encoder is the neighbour, the bi-LSTM - 340 registers per SIMD lane - cannot
            #  be scheduled before that launch ends; last in this graph it delays nothing else - graphed.py sets _txt_last)
            last = getattr(self, "_txt_last", False) and not getattr(self, "_txt_with_dep", False)
            if not last:
                txt, lengths, tk = self._stash_txt if getattr(self, "_txt_with_dep", False) else _txt_branch(None)
            mp, mkv = _map_branch(None)
            if last:
                txt, lengths, tk = _txt_branch(None)
            ops.prev_action_embed(prev_actions, masks_u8, self.prev_action_embedding.weight,
                                  state_in[:, d_out + m_out:], x2[:, o_prev:])
            self._stash = dict(txt=(txt, lengths, tk), map=(mp, mkv))
            return None, None
        if stage == "post":
            st = self._stash
            txt, lengths, tk = st["txt"]
            mp, mkv = st["map"]
            dep, dkv = self._stash_dep
        elif side is None:
            from . import train as _train

            overlap = save is not None and _train.OVERLAP_INSTRUCTION and not torch.cuda.is_current_stream_capturing()
            if overlap:  # training pass: the instruction bi-LSTM runs beside the map CNN (train.py)
                cur, st = torch.cuda.current_stream(), _train.side_stream(dev)
                st.wait_stream(cur)
                _train.share_with_stream((observations.get("instruction"), observations.get("instruction_unique")), st)
                with torch.cuda.stream(st):
                    txt, lengths, tk = _txt_branch(s_txt)
            else:
                txt, lengths, tk = _txt_branch(s_txt)
            dep, dkv = _dep_branch()
            mp, mkv = _map_branch(s_map)
            if overlap:
                cur.wait_stream(st)
                _train.share_with_stream((txt, lengths, tk, s_txt), cur)
        else:
            # three independent, latency-bound branches on forked streams (graphed.py): depth ResNet
            # (critical path, submitted first) || instruction bi-LSTM || (mapper ->) map CNN
            cur = torch.cuda.current_stream()
            st_txt, st_map = side
            st_txt.wait_stream(cur)
            st_map.wait_stream(cur)
            dep, dkv = _dep_branch()
            with torch.cuda.stream(st_txt):
                txt, lengths, tk = _txt_branch(None)
            with torch.cuda.stream(st_map):
                mp, mkv = _map_branch(None)
            cur.wait_stream(st_txt)
            cur.wait_stream(st_map)
        L = txt.shape[2]
        Cd, Cm = dep.shape[1], mp.shape[1]
        P = dep.shape[2] * dep.shape[3]
        if stage != "post":
            ops.prev_action_embed(prev_actions, masks_u8, self.prev_action_embedding.weight,
                                  state_in[:, d_out + m_out:], x2[:, o_prev:])

        rnn_out = getattr(self, "_rnn_out_buffer", None) if save is None else None  # graphed.py: persistent buffer
        if rnn_out is None:
            rnn_out = torch.empty_like(rnn_states)
        if fused_head and P <= 16 and L <= 512 and tk.shape[1] == H + 1 + h2:
            # GRU-1 -> text attention -> depth / map attention -> compress -> GRU-2 as one C-ABI call (five phase kernels,
            # csrc/cma_step.hip); `tk` holds the folded [Mq | TQb] operand here
            feats = torch.empty((rows, H), dtype=torch.float32, device=dev)
            g1, g2, sc = self.state_encoder.rnn, self.second_state_encoder.rnn, self.second_state_compress[0]
            d = ops.CmaStepDesc()
            d.rows, d.L, d.P, d.H, d.Hq, d.Ct, d.d_out, d.m_out, d.E, d.x2w = rows, L, P, H, h2, txt.shape[1], d_out, m_out, E, x2w
            d.state_in, d.h_in, d.ld_h, d.mask = ops.dptr(state_in), ops.dptr(rnn_states), rnn_states.stride(0), ops.dptr(masks_u8)
            d.w_ih1, d.w_hh1, d.b_ih1, d.b_hh1 = (ops.dptr(g1.weight_ih_l0), ops.dptr(g1.weight_hh_l0),
                                                  ops.dptr(g1.bias_ih_l0), ops.dptr(g1.bias_hh_l0))
            d.Mq, d.Mq_img = ops._p(tk), tk.stride(0)
            d.TQb, d.TQb_img = ops._p(tk[:, H + 1:]), tk.stride(0)
            d.lengths, d.txt = ops.dptr(lengths), ops.dptr(txt)
            d.dkv, d.mkv, d.scale = ops.dptr(dkv), ops.dptr(mkv), self._scale_f
            d.w_c, d.b_c = ops.dptr(sc.weight), ops.dptr(sc.bias)
            d.w_ih2, d.w_hh2, d.b_ih2, d.b_hh2 = (ops.dptr(g2.weight_ih_l0), ops.dptr(g2.weight_hh_l0),
                                                  ops.dptr(g2.bias_ih_l0), ops.dptr(g2.bias_hh_l0))
            d.x2, d.h_out, d.ld_ho, d.feats = ops.dptr(x2), ops.dptr(rnn_out), rnn_out.stride(0), ops.dptr(feats)
            ws = ops.cma_step_ws(rows, L, P, H, dev)
            d.ws = ops.dptr(ws)
            ops.cma_step(d)
            return feats, rnn_out
        s_g1 = {} if save is not None else None
        s_g2 = {} if save is not None else None
        state = x2[:, :H]
        self.state_encoder(state_in, rnn_states[:, 0], masks_u8, state, rnn_out[:, 0], s_g1)

        q1 = ops.linear(state, self.state_q.weight, self.state_q.bias)
        a_txt = torch.empty((rows, L), dtype=torch.float32, device=dev) if save is not None else None
        text = x2[:, o_txt:o_txt + 256]
        ops.attn(q1, tk.view(tk.shape[0], h2, L), txt, lengths, self._scale_f, text, a_txt, row_index=inv)

        dkv, mkv = dkv.view(rows, -1, P), mkv.view(rows, -1, P)
        q2 = ops.linear(text, self.text_q.weight, self.text_q.bias)
        a_dep = torch.empty((rows, P), dtype=torch.float32, device=dev) if save is not None else None
        a_map = torch.empty((rows, P), dtype=torch.float32, device=dev) if save is not None else None
        if save is None and P <= 32:
            # rollout head: both short-axis attentions (they share the query) in one launch
            ops.attn_small2(q2, dkv[:, :h2], dkv[:, h2:], x2[:, o_dep:o_dep + d_out], mkv[:, :h2], mkv[:, h2:],
                            x2[:, o_map:o_map + m_out], self._scale_f)
        else:
            ops.attn(q2, dkv[:, :h2], dkv[:, h2:], None, self._scale_f, x2[:, o_dep:o_dep + d_out], a_dep)
            ops.attn(q2, mkv[:, :h2], mkv[:, h2:], None, self._scale_f, x2[:, o_map:o_map + m_out], a_map)

        sc = self.second_state_compress[0]
        c2 = ops.linear(x2, sc.weight, sc.bias, relu=True)
        feats = torch.empty((rows, H), dtype=torch.float32, device=dev)
        self.second_state_encoder(c2, rnn_states[:, 1], masks_u8, feats, rnn_out[:, 1], s_g2)

        if save is not None:
            save.update(
                txt=s_txt, map=s_map, g1=s_g1, g2=s_g2, dep=dep, mp=mp, txt_out=txt, lengths=lengths,
                state_in=state_in, x2=x2, q1=q1, tk=tk, a_txt=a_txt, dkv=dkv, mkv=mkv, q2=q2, a_dep=a_dep,
                a_map=a_map, c2=c2, feats=feats, rows=rows, N=N, L=L, P=P, offs=(o_txt, o_dep, o_map, o_prev),
                masks=masks_u8, prev_actions=prev_actions, depth_from_features="depth_features" in observations,
                inv=inv,
            )
        return feats, rnn_out

    def forward(self, observations, rnn_states, prev_actions, action_masks, episode_masks=None, tour_masks=None):
        """Same signature/return as the reference forward (map_cma_policy.py:276-368); the MapCMA net
        only consumes `action_masks` (episode/tour masks default to it, :284-287)."""
        from .train import MapCMAForwardFn, all_params

        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in all_params(self))
        if needs_grad:

            feats, rnn_out = MapCMAForwardFn.run(self, observations, rnn_states, prev_actions, action_masks)
        else:
            feats, rnn_out = self.forward_hip(observations, rnn_states, prev_actions, action_masks)
        if self.model_config.PROGRESS_MONITOR.use and AuxLosses.is_active():
            from .train import progress_monitor_loss

            loss = progress_monitor_loss(self, feats, observations["progress"])
            AuxLosses.register_loss("progress_monitor", loss, self.model_config.PROGRESS_MONITOR.alpha)
        return feats, rnn_out


class ILPolicy(Policy):
    """models/policy.py:12-83."""

    def __init__(self, net, dim_actions):
        nn.Module.__init__(self)
        self.net = net
        self.dim_actions = dim_actions
        self.action_distribution = CategoricalNet(self.net.output_size, self.dim_actions)

    def forward(self, *x):
        raise NotImplementedError

    # keys a collection loop adds to the observation dict to have the sampled action drawn (and beta-mixed) on the
    # device from host-supplied uniforms; without them `sample()` is torch's own Categorical draw
    U_SAMPLE, U_BETA = "_u_sample", "_u_beta"
    collect_mix = None  # {"beta": float, "expert_uuid": str} while a DAgger collection mixes in-kernel

    def _act(self, features, deterministic, observations=None):
        lin = self.action_distribution.linear
        small = features.is_cuda and lin.out_features <= 8 and features.stride(-1) == 1
        out = getattr(self, "_action_out_buffer", None)
        if deterministic:  # distribution.mode() == argmax of probs == argmax of logits: head + argmax, one launch
            if small:
                return ops.linear_argmax(features, lin.weight, lin.bias, out=out)
            logits = self.action_distribution.raw_logits(features)
            return ops.argmax_rows(logits.contiguous(), out=out)
        if small and observations is not None and self.U_SAMPLE in observations:
            # inverse-CDF draw from the caller's uniforms (+ expert mixing / the -1 rule of a DAgger collection) in the
            # head's own launch: a pure function of the step's inputs, so the sampled step replays as a hipGraph
            mix = self.collect_mix
            u_beta = observations.get(self.U_BETA) if mix else None
            expert = observations[mix["expert_uuid"]].view(-1) if mix else None
            return ops.linear_sample(features, lin.weight, lin.bias, observations[self.U_SAMPLE].view(-1),
                                     None if u_beta is None else u_beta.view(-1), mix["beta"] if mix else 0.0, expert,
                                     out=out)
        logits = self.action_distribution.raw_logits(features)
        return CustomFixedCategorical(logits=logits).sample()

    def act(self, observations, rnn_states, prev_actions, masks, deterministic=False):
        features, rnn_states = self.net(observations, rnn_states, prev_actions, masks)
        return self._act(features, deterministic, observations), rnn_states

    def act_iterative(self, observations, rnn_hidden_states, prev_actions, agent_episode_not_done_masks,
                      sim_episode_not_done_masks, tour_not_done_masks, action_masks, deterministic=False):
        return self.act(observations, rnn_hidden_states, prev_actions, agent_episode_not_done_masks,
                        deterministic=deterministic)

    def get_value(self, *args, **kwargs):
        raise NotImplementedError

    def evaluate_actions(self, *args, **kwargs):
        raise NotImplementedError

    def build_distribution(self, observations, rnn_states, prev_actions, masks) -> CustomFixedCategorical:
        features, rnn_states = self.net(observations, rnn_states, prev_actions, masks)
        return self.action_distribution(features)


@baseline_registry.register_policy
class MapCMAPolicy(ILPolicy):
    def __init__(self, observation_space, action_space, config):
        super().__init__(
            MapCMANet(observation_space=observation_space, config=config, num_actions=action_space.n),
            action_space.n,
        )

    def act_iterative(self, observations, rnn_hidden_states, prev_actions, agent_episode_not_done_masks,
                      sim_episode_not_done_masks, tour_not_done_masks, action_masks, deterministic=False):
        # quirk Q11: only the agent-episode mask reaches the net (map_cma_policy.py:56-63)
        features, rnn_hidden_states = self.net(
            observations, rnn_hidden_states, prev_actions, action_masks=agent_episode_not_done_masks,
            episode_masks=None, tour_masks=None,
        )
        return self._act(features, deterministic, observations), rnn_hidden_states

    def build_features(self, observations, rnn_hidden_states, prev_actions, agent_episode_not_done_masks,
                       tour_not_done_masks=None) -> Tuple[Tensor, Tensor]:
        """The net half of `build_distribution` (the fused CE kernel of the HIP update takes raw logits)."""
        return self.net(observations, rnn_hidden_states, prev_actions, action_masks=agent_episode_not_done_masks)

    def build_distribution(self, observations, rnn_hidden_states, prev_actions, agent_episode_not_done_masks,
                           tour_not_done_masks=None) -> Tuple[CustomFixedCategorical, Tensor]:
        features, rnn_hidden_states = self.build_features(observations, rnn_hidden_states, prev_actions,
                                                          agent_episode_not_done_masks, tour_not_done_masks)
        return self.action_distribution(features), rnn_hidden_states

    @classmethod
    def from_config(cls, config, observation_space, action_space):
        return cls(observation_space=observation_space, action_space=action_space, config=config)
