"""TEST INFRASTRUCTURE ONLY - CPU (torch fp32) restatement of the un-vendored habitat-lab pieces
the MapCMA hot path calls.  Never imported by the product path (ivln-ce_amd/).

habitat-lab is an EMPTY, un-vendored submodule of the reference (/root/reference/.gitmodules:1-3;
README.md:15 names tag v0.1.7).  The arithmetic below restates, from the published v0.1.7 sources,

  * habitat_baselines.rl.ddppo.policy.resnet        (GroupNorm ResNet50, baseplanes/ngroups args)
  * habitat_baselines.rl.ddppo.policy.resnet_policy (ResNetEncoder: avg_pool2d(2) -> backbone ->
                                                     3x3 compression conv + GroupNorm(1) + ReLU)
  * habitat_baselines.rl.models.rnn_state_encoder   (build_rnn_state_encoder / masked GRU)

anchored on the reference's own call sites:
  ivlnce_baselines/models/encoders/resnet_encoders.py:31-43  (baseplanes=32, ngroups=16, resnet50)
  ivlnce_baselines/models/map_cma_policy.py:180-185,226-231,314-318,346-353 (state encoders)
and on the structural pins it implies: state-dict key names
`visual_encoder.backbone.layer{1-4}.{i}.convs.{0,1,3,4,6,7}`, `compression.{0,1}`,
`state_encoder.rnn.weight_ih_l0` (1536,416), and output_shape == (128,4,4) (depth_linear
in-features 3072, map_cma_policy.py:156-163).

PARITY UNPINNED at this boundary: the reference holds no tests or golden vectors for these
modules and habitat-lab cannot be imported here; this restatement IS the definition the HIP
kernels are checked against.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def conv3x3(in_planes, out_planes, stride=1, groups=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False, groups=groups)


def conv1x1(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=False)


def _build_bottleneck_branch(inplanes, planes, ngroups, stride, expansion, groups=1):
    return nn.Sequential(
        conv1x1(inplanes, planes),
        nn.GroupNorm(ngroups, planes),
        nn.ReLU(True),
        conv3x3(planes, planes, stride, groups=groups),
        nn.GroupNorm(ngroups, planes),
        nn.ReLU(True),
        conv1x1(planes, planes * expansion),
        nn.GroupNorm(ngroups, planes * expansion),
    )


class Bottleneck(nn.Module):
    expansion = 4
    resneXt = False

    def __init__(self, inplanes, planes, ngroups, stride=1, downsample=None, cardinality=1):
        super().__init__()
        self.convs = _build_bottleneck_branch(inplanes, planes, ngroups, stride, self.expansion, groups=1)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.convs(x)
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class ResNet(nn.Module):
    def __init__(self, in_channels, base_planes, ngroups, block, layers, cardinality=1):
        super().__init__()
        self.conv1 = nn.Sequential(
            nn.Conv2d(in_channels, base_planes, kernel_size=7, stride=2, padding=3, bias=False),
            nn.GroupNorm(ngroups, base_planes),
            nn.ReLU(True),
        )
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.cardinality = cardinality
        self.inplanes = base_planes
        if block.resneXt:
            base_planes *= 2
        self.layer1 = self._make_layer(block, ngroups, base_planes, layers[0])
        self.layer2 = self._make_layer(block, ngroups, base_planes * 2, layers[1], stride=2)
        self.layer3 = self._make_layer(block, ngroups, base_planes * 2 * 2, layers[2], stride=2)
        self.layer4 = self._make_layer(block, ngroups, base_planes * 2 * 2 * 2, layers[3], stride=2)
        self.final_channels = self.inplanes
        self.final_spatial_compress = 1.0 / (2 ** 5)

    def _make_layer(self, block, ngroups, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                conv1x1(self.inplanes, planes * block.expansion, stride),
                nn.GroupNorm(ngroups, planes * block.expansion),
            )
        layers = [block(self.inplanes, planes, ngroups, stride, downsample, cardinality=self.cardinality)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, ngroups))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.conv1(x)
        x = self.maxpool(x)
        x = self.layer1(x)
        x = self.layer2(x)
        x = self.layer3(x)
        x = self.layer4(x)
        return x


def resnet50(in_channels, base_planes, ngroups):
    return ResNet(in_channels, base_planes, ngroups, Bottleneck, [3, 4, 6, 3])


class ResNetEncoder(nn.Module):
    def __init__(
        self,
        observation_space,
        baseplanes=32,
        ngroups=32,
        spatial_size=128,
        make_backbone=None,
        normalize_visual_inputs=False,
    ):
        super().__init__()
        spaces = observation_space.spaces
        if "rgb" in spaces:
            self._n_input_rgb = spaces["rgb"].shape[2]
            spatial_size = spaces["rgb"].shape[0] // 2
        else:
            self._n_input_rgb = 0
        if "depth" in spaces:
            self._n_input_depth = spaces["depth"].shape[2]
            spatial_size = spaces["depth"].shape[0] // 2
        else:
            self._n_input_depth = 0
        assert not normalize_visual_inputs, "MapCMA passes normalize_visual_inputs=False"
        self.running_mean_and_var = nn.Sequential()
        input_channels = self._n_input_depth + self._n_input_rgb
        self.backbone = make_backbone(input_channels, baseplanes, ngroups)
        final_spatial = int(spatial_size * self.backbone.final_spatial_compress)
        after_compression_flat_size = 2048
        num_compression_channels = int(round(after_compression_flat_size / (final_spatial ** 2)))
        self.compression = nn.Sequential(
            nn.Conv2d(self.backbone.final_channels, num_compression_channels, kernel_size=3, padding=1, bias=False),
            nn.GroupNorm(1, num_compression_channels),
            nn.ReLU(True),
        )
        self.output_shape = (num_compression_channels, final_spatial, final_spatial)
        self.layer_init()

    def layer_init(self):
        for layer in self.modules():
            if isinstance(layer, (nn.Conv2d, nn.Linear)):
                nn.init.kaiming_normal_(layer.weight, nn.init.calculate_gain("relu"))
                if layer.bias is not None:
                    nn.init.constant_(layer.bias, val=0)

    @property
    def is_blind(self):
        return self._n_input_rgb + self._n_input_depth == 0

    def forward(self, observations):
        cnn_input = []
        if self._n_input_rgb > 0:
            rgb = observations["rgb"].permute(0, 3, 1, 2) / 255.0
            cnn_input.append(rgb)
        if self._n_input_depth > 0:
            cnn_input.append(observations["depth"].permute(0, 3, 1, 2))
        x = torch.cat(cnn_input, dim=1)
        x = F.avg_pool2d(x, 2)
        x = self.running_mean_and_var(x)
        x = self.backbone(x)
        x = self.compression(x)
        return x


class RNNStateEncoder(nn.Module):
    """Masked single-layer GRU.  hidden (N, L, H) batch-first; x is (N, F) for one step or
    time-major (T*N, F) for a sequence; hidden is zeroed where masks == 0."""

    def __init__(self, input_size, hidden_size, num_layers=1):
        super().__init__()
        self.num_recurrent_layers = num_layers
        self.rnn = nn.GRU(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers)
        for name, param in self.rnn.named_parameters():
            if "weight" in name:
                nn.init.orthogonal_(param)
            elif "bias" in name:
                nn.init.constant_(param, 0)

    def forward(self, x, hidden_states, masks):
        hidden_states = hidden_states.permute(1, 0, 2).contiguous()  # (L, N, H)
        n = hidden_states.size(1)
        if x.size(0) == n:
            hidden_states = hidden_states * masks.view(1, -1, 1).to(hidden_states.dtype)
            x, hidden_states = self.rnn(x.unsqueeze(0), hidden_states)
            x = x.squeeze(0)
        else:
            t = x.size(0) // n
            x = x.view(t, n, x.size(1))
            masks = masks.view(t, n).to(hidden_states.dtype)
            outs = []
            for i in range(t):
                hidden_states = hidden_states * masks[i].view(1, -1, 1)
                o, hidden_states = self.rnn(x[i : i + 1], hidden_states)
                outs.append(o)
            x = torch.cat(outs, dim=0).view(t * n, -1)
        return x, hidden_states.permute(1, 0, 2)


def build_rnn_state_encoder(input_size, hidden_size, rnn_type="GRU", num_layers=1):
    assert rnn_type.lower() == "gru", "MapCMA configs use GRU (config/default.py:156)"
    return RNNStateEncoder(input_size, hidden_size, num_layers)


def scatter_max(src, index):
    """torch_scatter 2.0.9 `scatter_max(src, index)` for 1-D src (call site
    ivlnce_baselines/common/mapping_module/mapper.py:471-474), CPU semantics: out size
    index.max()+1, first index attaining the max wins, empty slots: out 0 / arg = src.numel().
    PARITY UNPINNED for ties (package absent)."""
    n = int(index.max().item()) + 1 if index.numel() else 0
    P = src.numel()
    big = torch.full((n,), float("-inf"), dtype=src.dtype)
    out = big.scatter_reduce(0, index, src, reduce="amax", include_self=True)
    is_max = src == out[index]
    cand = torch.where(is_max, torch.arange(P), torch.full((P,), P))
    arg = torch.full((n,), P, dtype=torch.long).scatter_reduce(0, index, cand, reduce="amin", include_self=True)
    out = torch.where(arg == P, torch.zeros_like(out), out)
    return out, arg
