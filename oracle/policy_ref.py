"""TEST INFRASTRUCTURE ONLY - plain-PyTorch (CPU, fp32) restatement of the MapCMA policy.
Never imported by the product path (ivln-ce_amd/).  Used by tests as the float oracle for the HIP
kernels and by bench.py as the `cpu_baseline` ("port") leg.

Follows (paths relative to /root/reference):
  MapCMANet / MapCMAPolicy   ivlnce_baselines/models/map_cma_policy.py:28-368
  ILPolicy                   ivlnce_baselines/models/policy.py:12-83
  SemanticMapEncoder / CBRA  ivlnce_baselines/models/encoders/map_encoder.py:8-97
  InstructionEncoder         ivlnce_baselines/models/encoders/instruction_encoder.py:11-94
  VlnResnetDepthEncoder      ivlnce_baselines/models/encoders/resnet_encoders.py:17-115
  CategoricalNet             ivlnce_baselines/common/utils.py:149-185
  AuxLosses                  ivlnce_baselines/common/aux_losses.py:4-44
  _update_agent loss         ivlnce_baselines/common/base_il_trainer.py:173-219
and the habitat-lab pieces restated in oracle/habitat_ext_ref.py (parity unpinned there).

Module/parameter names mirror the reference so `state_dict()` keys are identical
(SURVEY.md section 8b); pinned by tests/test_oracle_policy.py against
tests/golden/policy_*.npz, produced by the reference's own MapCMAPolicy
(tests/golden/gen_policy_golden.py) under the shared deterministic initialiser
tests/golden/det_init.py.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import habitat_ext_ref as ext


class _CBRA(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Sequential(
            nn.Conv2d(cin, cout, kernel_size=7, padding=3), nn.BatchNorm2d(cout), nn.ReLU(inplace=True), nn.AvgPool2d(2)
        )

    def forward(self, x):
        return self.conv(x)


class SemanticMapEncoderRef(nn.Module):
    def __init__(self, num_classes=13, ch=32, last_ch_mult=4, map_hw=(64, 64)):
        super().__init__()
        self.num_classes = num_classes
        self.cnn = nn.Sequential(_CBRA(num_classes + 1, ch), _CBRA(ch, 2 * ch), _CBRA(2 * ch, 4 * ch), _CBRA(4 * ch, ch * last_ch_mult))
        self.output_shape = (ch * last_ch_mult, map_hw[0] // 16, map_hw[1] // 16)

    def features(self, obs):
        occ = obs["occupancy_map"].unsqueeze(1)
        sem = F.one_hot(obs["semantic_map"].long(), self.num_classes).permute(0, 3, 1, 2)
        # (the reference casts to torch.float; tests that run this oracle in float64 - the exact value both fp32
        #  implementations approximate - change the default dtype around the call)
        return torch.cat((occ, sem), 1).to(torch.get_default_dtype())

    def forward(self, obs):
        return self.cnn(self.features(obs))


class InstructionEncoderRef(nn.Module):
    def __init__(self, vocab=2504, emb=50, hidden=128):
        super().__init__()
        self.encoder_rnn = nn.LSTM(input_size=emb, hidden_size=hidden, bidirectional=True)
        self.embedding_layer = nn.Embedding(vocab, emb, padding_idx=0)
        self.output_size = 2 * hidden

    def forward(self, obs):
        x = self.embedding_layer(obs["instruction"].long())
        lengths = ((x != 0.0).long().sum(dim=2) != 0).long().sum(dim=1).cpu()
        packed = nn.utils.rnn.pack_padded_sequence(x, lengths, batch_first=True, enforce_sorted=False)
        out, _ = self.encoder_rnn(packed)
        return nn.utils.rnn.pad_packed_sequence(out, batch_first=True)[0].permute(0, 2, 1)


class _Space:
    def __init__(self, shape):
        self.shape = shape


class _Spaces:
    def __init__(self, d):
        self.spaces = d


class DepthEncoderRef(nn.Module):
    def __init__(self, depth_hw=(256, 256)):
        super().__init__()
        self.visual_encoder = ext.ResNetEncoder(
            _Spaces({"depth": _Space((depth_hw[0], depth_hw[1], 1))}),
            baseplanes=32, ngroups=16, make_backbone=ext.resnet50, normalize_visual_inputs=False,
        )
        for p in self.visual_encoder.parameters():
            p.requires_grad_(False)
        c, h, w = self.visual_encoder.output_shape
        self.spatial_embeddings = nn.Embedding(h * w, 64)
        self.output_shape = (c + 64, h, w)

    def forward(self, obs):
        x = obs["depth_features"] if "depth_features" in obs else self.visual_encoder(obs)
        b, c, h, w = x.size()
        sp = self.spatial_embeddings(torch.arange(0, self.spatial_embeddings.num_embeddings, device=x.device))
        sp = sp.view(1, -1, h, w).expand(b, self.spatial_embeddings.embedding_dim, h, w)
        return torch.cat([x, sp], dim=1)


class MapCMANetRef(nn.Module):
    def __init__(self, num_actions=4, hidden=512, depth_out=128, map_out=256, use_pm=False, pm_alpha=1.0,
                 depth_hw=(256, 256), map_hw=(64, 64)):
        super().__init__()
        self.use_pm, self.pm_alpha = use_pm, pm_alpha
        self.map_encoder = SemanticMapEncoderRef(map_hw=map_hw)
        self.instruction_encoder = InstructionEncoderRef()
        self.depth_encoder = DepthEncoderRef(depth_hw)
        self.prev_action_embedding = nn.Embedding(num_actions + 1, 32)
        self._hidden_size = hidden
        dshape, mshape = self.depth_encoder.output_shape, self.map_encoder.output_shape
        self.depth_linear = nn.Sequential(nn.Flatten(), nn.Linear(dshape[0] * dshape[1] * dshape[2], depth_out), nn.ReLU(True))
        self.map_linear = nn.Sequential(nn.Flatten(), nn.Linear(mshape[0] * mshape[1] * mshape[2], map_out), nn.ReLU(True))
        self.state_encoder = ext.build_rnn_state_encoder(depth_out + map_out + 32, hidden, "GRU", 1)
        self.dep_kv = nn.Conv1d(dshape[0], hidden // 2 + depth_out, 1)
        self.map_kv = nn.Conv1d(mshape[0], hidden // 2 + map_out, 1)
        self.state_q = nn.Linear(hidden, hidden // 2)
        self.text_k = nn.Conv1d(256, hidden // 2, 1)
        self.text_q = nn.Linear(256, hidden // 2)
        self.register_buffer("_scale", torch.tensor(1.0 / ((hidden // 2) ** 0.5)))
        self.second_state_compress = nn.Sequential(nn.Linear(hidden + depth_out + 256 + map_out + 32, hidden), nn.ReLU(True))
        self.second_state_encoder = ext.build_rnn_state_encoder(hidden, hidden, "GRU", 1)
        self.output_size = hidden
        self.progress_monitor = nn.Linear(hidden, 1)
        self.num_recurrent_layers = 2
        self.aux = {}
        self.train()

    def _attn(self, q, k, v, mask=None):
        logits = torch.einsum("nc, nci -> ni", q, k)
        if mask is not None:
            logits = logits - mask.float() * 1e8
        return torch.einsum("ni, nci -> nc", F.softmax(logits * self._scale, dim=1), v)

    def forward(self, obs, rnn_states, prev_actions, masks, want_aux=False):
        txt = self.instruction_encoder(obs)
        dep = torch.flatten(self.depth_encoder(obs), 2)
        mp = torch.flatten(self.map_encoder(obs), 2)
        pa = self.prev_action_embedding(((prev_actions.float() + 1) * masks).long().view(-1))
        state_in = torch.cat([self.depth_linear(dep), self.map_linear(mp), pa], dim=1)
        out_states = rnn_states.detach().clone()
        state, out_states[:, 0:1] = self.state_encoder(state_in, rnn_states[:, 0:1], masks)
        txt_mask = (txt == 0.0).all(dim=1)
        text = self._attn(self.state_q(state), self.text_k(txt), txt, txt_mask)
        h2 = self._hidden_size // 2
        dep_k, dep_v = torch.split(self.dep_kv(dep), h2, dim=1)
        map_k, map_v = torch.split(self.map_kv(mp), h2, dim=1)
        tq = self.text_q(text)
        x = torch.cat([state, text, self._attn(tq, dep_k, dep_v), self._attn(tq, map_k, map_v), pa], dim=1)
        x = self.second_state_compress(x)
        x, out_states[:, 1:2] = self.second_state_encoder(x, rnn_states[:, 1:2], masks)
        self.aux = {}
        if self.use_pm and want_aux:
            hat = torch.tanh(self.progress_monitor(x))
            # quirk Q7: (TN,) vs (TN,1) broadcast -> (TN,TN) loss matrix (map_cma_policy.py:356-361)
            self.aux["progress_monitor"] = (F.mse_loss(hat.squeeze(1), obs["progress"], reduction="none"), self.pm_alpha)
        return x, out_states


class _CategoricalNetRef(nn.Module):
    def __init__(self, nin, nout):
        super().__init__()
        self.linear = nn.Linear(nin, nout)
        nn.init.orthogonal_(self.linear.weight, gain=0.01)
        nn.init.constant_(self.linear.bias, 0)

    def forward(self, x):
        return self.linear(x)


class MapCMAPolicyRef(nn.Module):
    def __init__(self, num_actions=4, **kw):
        super().__init__()
        self.net = MapCMANetRef(num_actions=num_actions, **kw)
        self.action_distribution = _CategoricalNetRef(self.net.output_size, num_actions)

    def logits(self, obs, rnn_states, prev_actions, masks, want_aux=False):
        feats, states = self.net(obs, rnn_states, prev_actions, masks, want_aux)
        return self.action_distribution(feats), states, feats

    def act(self, obs, rnn_states, prev_actions, masks):
        logits, states, _ = self.logits(obs, rnn_states, prev_actions, masks)
        return logits.argmax(-1, keepdim=True), states, logits

    def update_loss(self, obs, prev_actions, not_done_masks, corrected_actions, weights):
        """base_il_trainer.py:173-219 up to (and excluding) backward()."""
        T, N = corrected_actions.size()
        h0 = torch.zeros(N, 2, self.net._hidden_size)
        logits, _, _ = self.logits(obs, h0, prev_actions, not_done_masks, want_aux=True)
        logits = logits.view(T, N, -1)
        ce = F.cross_entropy(logits.permute(0, 2, 1), corrected_actions, reduction="none")
        action_loss = ((weights * ce).sum(0) / weights.sum(0)).mean()
        aux = 0.0
        aux_mask = (weights > 0).view(-1)
        for loss, alpha in self.net.aux.values():
            aux = aux + alpha * torch.masked_select(loss, aux_mask).mean()
        return action_loss + aux, action_loss, aux, logits
