"""Encoders of the MapCMA policy with the reference's module / parameter names, forward on HIP.

torch.nn modules are used ONLY as parameter containers (identical `state_dict()` keys, init,
`.to()/.train()/.eval()` semantics); every forward below calls the HIP kernels through ops.py.

  SemanticMapEncoder / CBRA   ivlnce_baselines/models/encoders/map_encoder.py:8-97
  InstructionEncoder          ivlnce_baselines/models/encoders/instruction_encoder.py:11-94
  VlnResnetDepthEncoder       ivlnce_baselines/models/encoders/resnet_encoders.py:17-115
  ResNetEncoder / resnet50    habitat-lab v0.1.7 ddppo policy (un-vendored)
  RNNStateEncoder             habitat-lab v0.1.7 rnn_state_encoder (Appendix A.2)
"""
import gzip
import json
from typing import Optional

import torch
import torch.nn as nn

from . import depth_net, ops


# ------------------------------------------------------------------------------------------------
# Semantic map encoder
# ------------------------------------------------------------------------------------------------
class CBRA(nn.Module):
    """Conv(7x7 same) -> BatchNorm -> ReLU -> AvgPool(2)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Sequential(
            nn.Conv2d(in_channels, out_channels, kernel_size=7, padding=3),
            nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True),
            nn.AvgPool2d(2),
        )

    def forward_hip(self, x, save=None):
        conv, bn = self.conv[0], self.conv[1]
        # train mode: the conv's epilogue leaves per-tile {count, mean, M2} of what it stores, so the BatchNorm statistics
        # need no pass of their own over y (two reads of the largest tensors of the update)
        stats = [] if (bn.training and ops.CONV_STATS) else None
        y = ops.conv2d(x, conv.weight, stride=1, pad=3, shift=conv.bias, stats=stats)
        C = bn.num_features
        scale = torch.empty(C, dtype=torch.float32, device=x.device)
        shift = torch.empty(C, dtype=torch.float32, device=x.device)
        if bn.training:
            sm = sr = None
            if save is not None:
                sm = torch.empty(C, dtype=torch.float32, device=x.device)
                sr = torch.empty(C, dtype=torch.float32, device=x.device)
            if stats:
                ops.bn_stats_from_partials(stats[0][0], stats[0][1], bn, scale, shift, sm, sr)
            else:  # (the launch went another way - split K, scalar-gather GEMM: statistics from y)
                ops.bn_train_stats(y, bn, scale, shift, sm, sr)
            if bn.num_batches_tracked is not None:
                bn.num_batches_tracked += 1
            if save is not None:
                save.append(dict(x=x, y=y, scale=scale, shift=shift, mean=sm, rstd=sr, train=True))
        else:
            ops.bn_fold(bn, scale, shift)
            if save is not None:
                save.append(dict(x=x, y=y, scale=scale, shift=shift, train=False))
        return ops.scale_shift_relu_avgpool2(y, scale, shift)

    def _folded(self):
        """Eval-mode BatchNorm folded with the conv bias, cached until the weights change."""
        conv, bn = self.conv[0], self.conv[1]
        key = (ops.WEIGHT_EPOCH, bn.weight._version, bn.bias._version, bn.running_mean._version,
               bn.running_var._version, conv.bias._version, bn.weight.data_ptr())
        if getattr(self, "_fold_key", None) != key:
            C = bn.num_features
            scale = torch.empty(C, dtype=torch.float32, device=bn.weight.device)
            shift = torch.empty(C, dtype=torch.float32, device=bn.weight.device)
            ops.bn_fold(bn, scale, shift, conv.bias)
            self._fold_key, self._fold = key, (scale, shift)
        return self._fold

    def forward_infer(self, x):
        """Rollout path (BatchNorm in eval mode, nothing saved): conv leaves raw split-K slabs, one
        kernel reduces them + folded BN + ReLU + AvgPool: 2 launches per block."""
        scale, shift = self._folded()
        return ops.scale_shift_relu_avgpool2(ops.conv2d(x, self.conv[0].weight, pad=3, defer=True), scale, shift)


class SemanticMapEncoder(nn.Module):
    def __init__(self, observation_space, num_semantic_classes: int = 13, ch: int = 32, last_ch_mult: int = 8,
                 trainable: bool = True, from_pretrained: bool = False, checkpoint: Optional[str] = None):
        super().__init__()
        for k in ["occupancy_map", "semantic_map"]:
            if k not in observation_space.spaces:
                raise ValueError(f"key `{k}` expected in observation space.")
        self._map_dimensions = observation_space.spaces["occupancy_map"].shape
        self._num_semantic_classes = num_semantic_classes
        self.last_ch_mult = last_ch_mult
        self._ch = ch
        self.cnn = nn.Sequential(CBRA(14, ch), CBRA(ch, ch * 2), CBRA(ch * 2, ch * 4), CBRA(ch * 4, ch * last_ch_mult))
        if from_pretrained:
            ckpt = torch.load(checkpoint, map_location="cpu")["state_dict"]
            prefix = "encoder.cnn."
            self.cnn.load_state_dict({k[len(prefix):]: v for k, v in ckpt.items() if k.startswith(prefix)})
        for param in self.cnn.parameters():
            param.requires_grad_(trainable)
        if not trainable:
            self.eval()

    @property
    def output_shape(self):
        return (self._ch * self.last_ch_mult, self._map_dimensions[0] // 16, self._map_dimensions[1] // 16)

    def generate_map_features(self, observations):
        occ = observations["occupancy_map"].to(torch.uint8).contiguous()
        sem = observations["semantic_map"].to(torch.uint8).contiguous()
        return ops.map_features(occ, sem, self._num_semantic_classes)

    def forward(self, observations, save=None):
        for k in ["occupancy_map", "semantic_map"]:
            if k not in observations:
                raise ValueError(f"Observation `{k}` is missing.")
        x = self.generate_map_features(observations)
        for blk in self.cnn:
            if save is None and not blk.conv[1].training:
                x = blk.forward_infer(x)
            else:
                x = blk.forward_hip(x, save)
        return x


# ------------------------------------------------------------------------------------------------
# Instruction encoder
# ------------------------------------------------------------------------------------------------
class InstructionEncoder(nn.Module):
    def __init__(self, config) -> None:
        super().__init__()
        self.config = config
        assert config.rnn_type == "LSTM" and config.bidirectional, "MapCMA uses a bidirectional LSTM"
        self.encoder_rnn = nn.LSTM(input_size=config.embedding_size, hidden_size=config.hidden_size, bidirectional=True)
        if config.sensor_uuid == "instruction":
            if config.use_pretrained_embeddings:
                self.embedding_layer = nn.Embedding.from_pretrained(
                    embeddings=self._load_embeddings(), freeze=not config.fine_tune_embeddings
                )
            else:
                self.embedding_layer = nn.Embedding(
                    num_embeddings=config.vocab_size, embedding_dim=config.embedding_size, padding_idx=0
                )

    @property
    def output_size(self):
        return self.config.hidden_size * 2

    def _load_embeddings(self):
        with gzip.open(self.config.embedding_file, "rt") as f:
            return torch.tensor(json.load(f))

    def _gate_table(self):
        """(table (V, 8H), row_nonzero u8 (V)) of ivln_embed_gates_f32, cached until a weight changes; None while a
        stream capture is running and no valid table exists (the caller then runs the unfolded launches)."""
        rnn, E = self.encoder_rnn, self.embedding_layer.weight
        ps = (E, rnn.weight_ih_l0, rnn.weight_ih_l0_reverse, rnn.bias_ih_l0, rnn.bias_ih_l0_reverse)
        key = (ops.WEIGHT_EPOCH,) + tuple(p._version for p in ps) + tuple(p.data_ptr() for p in ps)
        c = self.__dict__.get("_gate_cache")
        if c is None or c[0] != key:
            if torch.cuda.is_current_stream_capturing():
                return None
            with torch.no_grad():
                W = torch.cat([rnn.weight_ih_l0, rnn.weight_ih_l0_reverse]).contiguous()
                b = torch.cat([rnn.bias_ih_l0, rnn.bias_ih_l0_reverse]).contiguous()
                table = ops.linear_gemm(E.detach().contiguous(), W, b)
                nz = (E.detach() != 0).any(dim=1).to(torch.uint8).contiguous()
            c = self.__dict__["_gate_cache"] = (key, (table, nz))
        return c[1]

    def step_cache(self, rows, L, device):
        """The per-episode cache of a rollout batch shape (ops.InstructionStepCache), or None: switched off, a capture in
        progress that would have to create or invalidate it (the warm-up steps before a capture do that), no folded table.
        A cache made for other weights (LSTM / embedding versions, ops.WEIGHT_EPOCH) is invalidated, not rebuilt: captured
        graphs hold its buffers."""
        if not ops.CACHE_INSTRUCTION:
            return None
        gate_key = (self.__dict__.get("_gate_cache") or (None,))[0]
        rnn = self.encoder_rnn
        ps = (rnn.weight_hh_l0, rnn.weight_hh_l0_reverse, rnn.bias_hh_l0, rnn.bias_hh_l0_reverse)
        key = (gate_key,) + tuple(p._version for p in ps) + tuple(p.data_ptr() for p in ps)
        caches = self.__dict__.setdefault("_step_caches", {})
        c = caches.get((rows, L, str(device)))
        capturing = torch.cuda.is_current_stream_capturing()
        if c is None:
            if capturing:
                return None
            if len(caches) >= 4:  # (batch shapes come and go as envs pause: keep the table small)
                caches.pop(next(iter(caches)))
            c = caches[(rows, L, str(device))] = ops.InstructionStepCache(rows, L, 4 * rnn.hidden_size, rnn.hidden_size, device, key)
        elif c.key != key:
            if capturing:
                return None
            c.invalidate()
            c.key = key
        return c

    def forward(self, observations, save=None):
        """(B, L) tokens -> (B, 2H, L) channel-major outputs, zero for t >= length; also returns
        lengths (device int32).  The reference returns (B, 2H, Lmax); the extra columns here are
        exactly the masked (zero-weight) attention positions.
        Rollout steps (no `save`, no autograd): the encoding is cached per row and recomputed only where the tokens
        changed (`step_cache`; the decision is taken on the device, ops.embed_gates) - `self.last_cache` then names the
        cache whose `dirty` flags belong to this call, else it is None."""
        tokens = observations["instruction"].long().contiguous()
        B, L = tokens.shape
        rnn = self.encoder_rnn
        H = rnn.hidden_size
        fold = self._gate_table() if (save is None and ops.FOLD_INSTRUCTION_GATES) else None
        self.last_cache = None
        if fold is not None and tokens.is_cuda and not torch.is_grad_enabled():
            cache = self.step_cache(B, L, tokens.device)
            if cache is not None:
                gx_f, gx_r, lengths = ops.embed_gates(tokens, *fold, cache=cache)
                out, _, _ = ops.lstm_bidir(gx_f, gx_r, rnn.weight_hh_l0, rnn.weight_hh_l0_reverse, rnn.bias_hh_l0,
                                           rnn.bias_hh_l0_reverse, lengths, B, L, H, spare=getattr(self, "lstm_spare", 1),
                                           ticket=getattr(self, "lstm_ticket", None), cache=cache)
                self.last_cache = cache
                return out, lengths
        if fold is not None:  # inference: embedding lookup + both W_ih projections = one lookup in a folded table
            emb = None
            gx_f, gx_r, lengths = ops.embed_gates(tokens, *fold)
        else:
            emb, lengths = ops.embed_lengths(tokens, self.embedding_layer.weight)
            gx_f = ops.linear_gemm(emb, rnn.weight_ih_l0, rnn.bias_ih_l0)
            gx_r = ops.linear_gemm(emb, rnn.weight_ih_l0_reverse, rnn.bias_ih_l0_reverse)
        out, gates, cs = ops.lstm_bidir(
            gx_f, gx_r, rnn.weight_hh_l0, rnn.weight_hh_l0_reverse, rnn.bias_hh_l0, rnn.bias_hh_l0_reverse, lengths,
            B, L, H, save=save is not None, spare=getattr(self, "lstm_spare", 1), ticket=getattr(self, "lstm_ticket", None),
        )
        if save is not None:
            save.update(emb=emb, lengths=lengths, gates=gates, cs=cs, out=out, tokens=tokens)
        return out, lengths


# ------------------------------------------------------------------------------------------------
# DD-PPO depth ResNet50 (GroupNorm), parameter tree named like habitat-lab's
# ------------------------------------------------------------------------------------------------
def _conv(cin, cout, k, stride=1, pad=0):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=pad, bias=False)


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, ngroups, stride=1, downsample=None):
        super().__init__()
        self.convs = nn.Sequential(
            _conv(inplanes, planes, 1), nn.GroupNorm(ngroups, planes), nn.ReLU(True),
            _conv(planes, planes, 3, stride, 1), nn.GroupNorm(ngroups, planes), nn.ReLU(True),
            _conv(planes, planes * 4, 1), nn.GroupNorm(ngroups, planes * 4),
        )
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward_hip(self, x):
        """conv -> GroupNorm(+ReLU) pairs: the conv leaves raw (split-K) slabs in the workspace and the
        GroupNorm kernel reduces + normalises them, 2 launches per pair."""
        c = self.convs
        y, ds = self.body_hip(x)
        if ds is not None:
            gn = self.downsample[1]
            return ops.groupnorm(y, c[7].weight, c[7].bias, c[7].num_groups, c[7].eps, relu=True, x2=ds,
                                 gamma2=gn.weight, beta2=gn.bias)
        return ops.groupnorm(y, c[7].weight, c[7].bias, c[7].num_groups, c[7].eps, relu=True, residual=x)

    def body_hip(self, x):
        """Everything but the block's last GroupNorm: -> (raw conv3 output, raw downsample conv output | None), both
        left as split-K slabs in the two workspaces."""
        c = self.convs
        ds = None
        if self.downsample is not None:
            # the downsample conv leaves its slabs in the second workspace; its GroupNorm is folded into the
            # block's last GroupNorm launch (same channels, same groups)
            gn = self.downsample[1]
            assert gn.num_groups == c[7].num_groups and gn.eps == c[7].eps
            ds = ops.conv2d(x, self.downsample[0].weight, stride=self.stride, defer=True, ws_slot=1)
        y = ops.conv2d(x, c[0].weight, defer=True)
        y = ops.groupnorm(y, c[1].weight, c[1].bias, c[1].num_groups, c[1].eps, relu=True)
        y = ops.conv2d(y, c[3].weight, stride=self.stride, pad=1, defer=True)
        y = ops.groupnorm(y, c[4].weight, c[4].bias, c[4].num_groups, c[4].eps, relu=True)
        return ops.conv2d(y, c[6].weight, defer=True), ds


class _ResNet50GN(nn.Module):
    def __init__(self, in_channels, base_planes, ngroups):
        super().__init__()
        self.conv1 = nn.Sequential(
            nn.Conv2d(in_channels, base_planes, kernel_size=7, stride=2, padding=3, bias=False),
            nn.GroupNorm(ngroups, base_planes), nn.ReLU(True),
        )
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.inplanes = base_planes
        self.layer1 = self._make_layer(ngroups, base_planes, 3)
        self.layer2 = self._make_layer(ngroups, base_planes * 2, 4, stride=2)
        self.layer3 = self._make_layer(ngroups, base_planes * 4, 6, stride=2)
        self.layer4 = self._make_layer(ngroups, base_planes * 8, 3, stride=2)
        self.final_channels = self.inplanes
        self.final_spatial_compress = 1.0 / (2 ** 5)

    def _make_layer(self, ngroups, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(_conv(self.inplanes, planes * 4, 1, stride), nn.GroupNorm(ngroups, planes * 4))
        layers = [_Bottleneck(self.inplanes, planes, ngroups, stride, downsample)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(_Bottleneck(self.inplanes, planes, ngroups))
        return nn.Sequential(*layers)

    def forward_chain(self, x, tail):
        """The whole backbone as a chain of ops.gn_conv launches: every GroupNorm kernel also computes its slice of
        the NEXT conv (csrc/gn_conv.hip), so a bottleneck is 3 launches and no conv launch remains after the stem.
        `tail` = (weight, stride, pad) of the conv that follows the backbone (the encoder's compression conv).
        Returns that conv's raw output (a `Deferred`), or None when a shape is outside the kernel's envelope."""
        blocks = [b for layer in (self.layer1, self.layer2, self.layer3, self.layer4) for b in layer]

        def feeds(blk):  # what the kernel that produces `blk`'s input has to emit for it
            ds = None if blk.downsample is None else (blk.downsample[0].weight, blk.stride)
            return dict(conv_a=(blk.convs[0].weight, 1, 0), conv_b=ds, want_act=ds is None)

        c, gn = self.conv1[0], self.conv1[1]
        y = ops.conv2d(x, c.weight, stride=2, pad=3, defer=True)
        first = ops.CHAIN_FROM_BLOCK if ops.CHAIN_FROM_BLOCK >= 0 else (3 if x.shape[0] <= 5 else 7)
        first = min(first, len(blocks))  # blocks before it run as conv + GroupNorm pairs
        if first == 0:
            r = ops.gn_conv(y, gn, relu=True, pool=True, **feeds(blocks[0]))
        else:
            r0 = ops.gn_conv(y, gn, relu=True, pool=True, want_act=True)  # stem GroupNorm + ReLU + MaxPool in one launch
            act = r0[0] if r0 is not None else \
                ops.pool2d(ops.groupnorm(y, gn.weight, gn.bias, gn.num_groups, gn.eps, relu=True), 3, 2, 1, "max")
            nxt = feeds(blocks[first]) if first < len(blocks) else dict(conv_a=tail)
            # the leading blocks as ivln_nconv_f32 launches (layer 1, or more: NCONV_BLOCKS), then pairs up to `first`
            nb_auto = ops.NCONV_BLOCKS if ops.NCONV_BLOCKS >= 0 else (3 if x.shape[0] <= 5 else 7)
            done = min(first, nb_auto) if ops.NCONV_FRONT else 0
            r = self._front_blocks_nconv(blocks[:done], act, nxt if done == first else None) if done else None
            if r is not None and done < first:
                act, r = r, None  # (the run ended before the chain starts: `r` is the activated output of its last block)
            elif r is None:
                done = 0
            if r is None:
                for blk in blocks[done:first - 1]:
                    act = blk.forward_hip(act)
                # the last pairwise block hands over: its final GroupNorm is the chain's first launch
                blk = blocks[first - 1]
                y3, ds = blk.body_hip(act)
                if ds is not None:
                    r = ops.gn_conv(y3, blk.convs[7], x2=ds, gn2=blk.downsample[1], **nxt)
                else:
                    r = ops.gn_conv(y3, blk.convs[7], residual=act, **nxt)
        for i, blk in enumerate(blocks):
            if i < first:
                continue
            if r is None:
                return None
            act, ya, yb = r
            c = blk.convs
            r = ops.gn_conv(ya, c[1], conv_a=(c[3].weight, blk.stride, 1))
            if r is None:
                return None
            nxt = feeds(blocks[i + 1]) if i + 1 < len(blocks) else dict(conv_a=tail)
            tail_in = dict(x2=yb, gn2=blk.downsample[1]) if blk.downsample is not None else dict(residual=act)
            r2 = None
            if i >= ops.CHAIN_PAIR_FROM_BLOCK:  # GN2 -> conv3 -> GN3 tail -> next conv1 in one launch
                r2 = ops.gn_conv(None, c[7], front=(r[1], c[4], c[6].weight), **tail_in, **nxt)
            if r2 is None:
                r = ops.gn_conv(r[1], c[4], conv_a=(c[6].weight, 1, 0))
                if r is None:
                    return None
                r2 = ops.gn_conv(r[1], c[7], **tail_in, **nxt)
            r = r2
        return None if r is None else r[1]

    @staticmethod
    def _front_blocks_nconv(blocks, act, nxt):
        """The bottlenecks BEFORE the gn_conv chain (large maps: layer 1) as one ivln_nconv_f32 launch per conv layer:
        every conv normalises its input on load from the statistics partials its producer left and emits those of its
        own output - complete tensors, no slabs, no GroupNorm launches.  `act`: the activated input of the first block;
        `nxt`: what the chain's first launch has to emit.  Returns that launch's result, or None (a stride or a shape
        outside the kernel's envelope: the caller runs the conv + GroupNorm pairs).  With `nxt` None the run ends before
        the chain starts: the activated output of the last block is returned instead."""
        identity, x1, xds = act, None, None
        for i, blk in enumerate(blocks):
            c = blk.convs
            G1, G3 = c[1].num_groups, c[7].num_groups
            if i == 0:  # first block: its convs read the activated tensor as it is
                ds = None if blk.downsample is None else (blk.downsample[0].weight, blk.downsample[1].num_groups, blk.stride)
                r = ops.nconv(act, None, relu=False, conv_a=(c[0].weight, G1), conv_b=ds)
                if r is None:
                    return None
                x1, xds = r[1], r[2]
            r = ops.nconv(x1, c[1], conv_a=(c[3].weight, c[4].num_groups, blk.stride))  # GroupNorm 1 + ReLU on load -> 3x3
            if r is None:
                return None
            r = ops.nconv(r[1], c[4], conv_a=(c[6].weight, G3))                  # GroupNorm 2 + ReLU on load -> 1x1
            if r is None:
                return None
            x3 = r[1]
            tail = dict(x2=xds, gn2=blk.downsample[1]) if blk.downsample is not None else dict(residual=identity)
            if i + 1 < len(blocks):  # the block's tail is built on load by the next block's first conv(s)
                nb = blocks[i + 1]
                ds = None if nb.downsample is None else (nb.downsample[0].weight, nb.downsample[1].num_groups, nb.stride)
                r = ops.nconv(x3, c[7], conv_a=(nb.convs[0].weight, nb.convs[1].num_groups), conv_b=ds,
                              want_act=nb.downsample is None, **tail)
                if r is None:
                    return None
                identity, x1, xds = r[0], r[1], r[2]
            elif nxt is not None:  # hand over to the chain: its first launch takes the raw tensors as one-slab inputs
                if blk.downsample is not None:
                    return ops.gn_conv(x3.deferred(), c[7], x2=xds.deferred(), gn2=blk.downsample[1], **nxt)
                return ops.gn_conv(x3.deferred(), c[7], residual=identity, **nxt)
            else:  # hand over to conv + GroupNorm pairs: materialise the block's output
                if blk.downsample is not None:
                    gd = blk.downsample[1]
                    return ops.groupnorm(x3.deferred(), c[7].weight, c[7].bias, G3, c[7].eps, relu=True, x2=xds.deferred(),
                                         gamma2=gd.weight, beta2=gd.bias)
                return ops.groupnorm(x3.deferred(), c[7].weight, c[7].bias, G3, c[7].eps, relu=True, residual=identity)
        return None

    def forward_hip(self, x):
        c, gn = self.conv1[0], self.conv1[1]
        y = ops.conv2d(x, c.weight, stride=2, pad=3, defer=True)
        y = ops.groupnorm(y, gn.weight, gn.bias, gn.num_groups, gn.eps, relu=True)
        y = ops.pool2d(y, 3, 2, 1, "max")
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                y = blk.forward_hip(y)
        return y


class ResNetEncoder(nn.Module):
    """habitat-lab ResNetEncoder for a depth-only observation space: avg_pool2d(2) -> GN-ResNet50 ->
    3x3 compression conv + GroupNorm(1) + ReLU -> (B,128,4,4)."""

    def __init__(self, depth_shape, baseplanes=32, ngroups=16):
        super().__init__()
        spatial_size = depth_shape[0] // 2
        self._n_input_depth = depth_shape[2]
        self.running_mean_and_var = nn.Sequential()
        self.backbone = _ResNet50GN(self._n_input_depth, baseplanes, ngroups)
        final_spatial = int(spatial_size * self.backbone.final_spatial_compress)
        num_compression_channels = int(round(2048 / (final_spatial ** 2)))
        self.compression = nn.Sequential(
            nn.Conv2d(self.backbone.final_channels, num_compression_channels, kernel_size=3, padding=1, bias=False),
            nn.GroupNorm(1, num_compression_channels), nn.ReLU(True),
        )
        self.output_shape = (num_compression_channels, final_spatial, final_spatial)
        for layer in self.modules():
            if isinstance(layer, (nn.Conv2d, nn.Linear)):
                nn.init.kaiming_normal_(layer.weight, nn.init.calculate_gain("relu"))
                if layer.bias is not None:
                    nn.init.constant_(layer.bias, val=0)

    @property
    def is_blind(self):
        return self._n_input_depth == 0

    def forward(self, observations, out=None, out_ctot=0):
        """observations["depth"]: (B,H,W,1) f32.  `out`: optional (B,128,4,4) channel slice of a wider
        NCHW buffer with `out_ctot` channels (the policy passes its depth+spatial-embedding buffer)."""
        depth = observations["depth"].to(torch.float32).contiguous()
        B, H, W, Cd = depth.shape
        assert Cd == 1, "depth-only encoder"
        c, gn = self.compression[0], self.compression[1]
        hw = self.output_shape[1] * self.output_shape[2]
        want = ops.DEPTH_NET == 2 or (ops.DEPTH_NET == 1 and (not getattr(self, "beside_other_work", False)
                                                                or B >= ops.DEPTH_NET_SPLIT_MIN))
        if (want and B <= depth_net.DepthNetPlan.MAX_IMAGES and (H, W) == (256, 256) and not torch.is_grad_enabled()
                and getattr(self, "latency_bound", True) and not getattr(self, "no_persistent", False)):
            # rollout batches: the whole encoder as ONE persistent launch, a cluster of 32 workgroups per image
            # (csrc/depth_net.hip); declined (False) when the device cannot keep all its workgroups resident
            plan = depth_net.plan_for(self, depth.device)
            dst = out if out is not None else torch.empty((B,) + tuple(self.output_shape), dtype=torch.float32, device=depth.device)
            if plan is not None and plan.run(depth, dst, out_ctot * hw if out is not None else self.output_shape[0] * hw):
                return dst
        x = ops.pool2d(depth.view(B, 1, H, W), 2, 2, 0, "avg")  # F.avg_pool2d(x, 2)
        chain = ops.CHAIN_GN_CONV and B <= ops.CHAIN_MAX_IMAGES and getattr(self, "latency_bound", True)
        y = self.backbone.forward_chain(x, (c.weight, 1, 1)) if chain else None
        if y is None:
            x = self.backbone.forward_hip(x)
            y = ops.conv2d(x, c.weight, pad=1, defer=True)
        if out is None:
            return ops.groupnorm(y, gn.weight, gn.bias, gn.num_groups, gn.eps, relu=True)
        return ops.groupnorm(y, gn.weight, gn.bias, gn.num_groups, gn.eps, relu=True, out=out,
                             y_img_stride=out_ctot * hw)


class VlnResnetDepthEncoder(nn.Module):
    def __init__(self, observation_space, output_size: int = 128, checkpoint: str = "NONE", backbone: str = "resnet50",
                 resnet_baseplanes: int = 32, normalize_visual_inputs: bool = False, trainable: bool = False,
                 spatial_output: bool = False) -> None:
        super().__init__()
        assert backbone == "resnet50" and not normalize_visual_inputs and spatial_output
        depth_shape = observation_space.spaces["depth"].shape
        if len(depth_shape) == 4:  # single_frame_box_shape (common/utils.py:38-48)
            depth_shape = depth_shape[1:]
        self.visual_encoder = ResNetEncoder(depth_shape, baseplanes=resnet_baseplanes, ngroups=resnet_baseplanes // 2)
        for param in self.visual_encoder.parameters():
            param.requires_grad_(trainable)
        if checkpoint != "NONE":
            ddppo_weights = torch.load(checkpoint, map_location="cpu")
            weights_dict = {}
            for k, v in ddppo_weights["state_dict"].items():  # resnet_encoders.py:48-61
                split_layer_name = k.split(".")[2:]
                if split_layer_name[0] != "visual_encoder":
                    continue
                weights_dict[".".join(split_layer_name[1:])] = v
            del ddppo_weights
            self.visual_encoder.load_state_dict(weights_dict, strict=True)
        self.spatial_output = spatial_output
        c, h, w = self.visual_encoder.output_shape
        self.spatial_embeddings = nn.Embedding(h * w, 64)
        self.output_shape = (c + self.spatial_embeddings.embedding_dim, h, w)

    @property
    def is_blind(self):
        return self.visual_encoder.is_blind

    def forward(self, observations):
        """-> (B, 192, 4, 4): visual features ++ the learned spatial embedding (a raw `.view` of the
        (16,64) table as (64,4,4), resnet_encoders.py:97-113)."""
        c, h, w = self.visual_encoder.output_shape
        E = self.spatial_embeddings.embedding_dim
        if "depth_features" in observations:
            feats = observations["depth_features"].to(torch.float32).contiguous()
            B = feats.shape[0]
            out = torch.empty((B, c + E, h, w), dtype=torch.float32, device=feats.device)
            ops.copy2d(feats.view(B, -1), out.view(B, -1), B, c * h * w)
        else:
            B = observations["depth"].shape[0]
            dev = observations["depth"].device
            sw = self.spatial_embeddings.weight
            if not torch.is_grad_enabled():
                # inference: the embedding half of the output is constant between weight updates - keep one output
                # buffer per batch size with that half filled once instead of one copy launch per step.  (Created
                # outside graph capture only: a buffer from a graph's private pool must not outlive that graph.)
                # Buffers are never freed or reallocated (a captured graph keeps raw pointers to them); a weight
                # update refills the embedding half in place.
                stamp = (ops.WEIGHT_EPOCH, sw._version, sw.data_ptr())
                cache = self.__dict__.setdefault("_out_cache", {})
                ent = cache.get((B, str(dev)))
                if (ent is None or ent[0] != stamp) and not torch.cuda.is_current_stream_capturing():
                    buf = ent[1] if ent is not None else torch.empty((B, c + E, h, w), dtype=torch.float32, device=dev)
                    ops.copy2d(sw.view(1, -1), buf.view(B, -1)[:, c * h * w:], B, E * h * w, broadcast_rows=True)
                    ent = cache[(B, str(dev))] = (stamp, buf)
                if ent is not None and ent[0] == stamp:
                    self.visual_encoder(observations, out=ent[1][:, :c], out_ctot=c + E)
                    return ent[1]
            out = torch.empty((B, c + E, h, w), dtype=torch.float32, device=dev)
            self.visual_encoder(observations, out=out[:, :c], out_ctot=c + E)
        ops.copy2d(self.spatial_embeddings.weight.view(1, -1), out.view(B, -1)[:, c * h * w:], B, E * h * w,
                   broadcast_rows=True)
        return out


# ------------------------------------------------------------------------------------------------
# RNN state encoder (masked GRU)
# ------------------------------------------------------------------------------------------------
class RNNStateEncoder(nn.Module):
    def __init__(self, input_size: int, hidden_size: int, num_layers: int = 1):
        super().__init__()
        assert num_layers == 1
        self.num_recurrent_layers = num_layers
        self.rnn = nn.GRU(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers)
        for name, param in self.rnn.named_parameters():
            if "weight" in name:
                nn.init.orthogonal_(param)
            elif "bias" in name:
                nn.init.constant_(param, 0)

    def forward(self, x, h0, masks_u8, out, state_out, save=None):
        """x (rows,I); h0 (N,H) view (row-strided) of the incoming state; masks u8 (rows); `out`
        (rows,H) row-strided destination of the per-step outputs; state_out (N,H) view for the final
        hidden state.  rows == N -> single step, else time-major sequence of T = rows/N steps."""
        rnn = self.rnn
        rows, N = x.shape[0], h0.shape[0]
        H = rnn.hidden_size
        if rows == N:
            saves = None
            if save is not None:
                saves = tuple(torch.empty((rows, H), dtype=torch.float32, device=x.device) for _ in range(4))
                save.update(r=saves[0], z=saves[1], n=saves[2], ghn=saves[3], T=1, N=N, x=x, h0=h0, masks=masks_u8, out=out)
            ops.gru_step(x, None, h0, masks_u8, rnn.weight_ih_l0, rnn.weight_hh_l0, rnn.bias_ih_l0, rnn.bias_hh_l0,
                         out, state_out, saves)
            return out
        T = rows // N
        gi = ops.linear_gemm(x, rnn.weight_ih_l0, rnn.bias_ih_l0)
        saves = None
        if save is not None:
            saves = tuple(torch.empty((rows, H), dtype=torch.float32, device=x.device) for _ in range(4))
            save.update(r=saves[0], z=saves[1], n=saves[2], ghn=saves[3], T=T, N=N, x=x, h0=h0, masks=masks_u8, out=out)
        ops.gru_seq(gi, h0, masks_u8, rnn.weight_hh_l0, rnn.bias_hh_l0, out, state_out, T, N, saves)
        return out


def build_rnn_state_encoder(input_size: int, hidden_size: int, rnn_type: str = "GRU", num_layers: int = 1):
    assert rnn_type.lower() == "gru", "MapCMA configs use STATE_ENCODER.rnn_type GRU"
    return RNNStateEncoder(input_size, hidden_size, num_layers)
