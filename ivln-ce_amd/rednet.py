"""RedNet RGB-D semantic segmentation (pred-semantics mapper) on HIP.

Same module / parameter names as the reference (ivlnce_baselines/common/mapping_module/rednet.py:7-358)
so `rednet_mp3d_best_model.pkl["model_state"]` loads unchanged; `PredictSemantics` mirrors
mapper.py:703-800.  Inference only (the reference freezes it and runs it under no_grad in eval
mode): every BatchNorm is folded into the producing conv's fused scale/shift epilogue, residual
adds and ReLU run in the same GEMM epilogue, transposed convs use the transposed-gather operand
mode of the MFMA implicit-GEMM kernel.
"""
import os

import torch
import torch.nn as nn

from . import ops


# RGB + depth encoders as one image-grouped pass (A/B switch: IVLN_REDNET_NO_GROUP=1 runs them as two chains)
GROUP_ENCODERS = True  # (False: the two encoders as separate launch chains - what tests compare the stacked form with)


def _conv3x3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=False)


class _Folded:
    """Cache of folded-BN scale/shift and re-laid-out transposed-conv weights (frozen network)."""

    def __init__(self):
        self.c = {}

    def bn(self, bn: nn.BatchNorm2d):
        k = id(bn)
        if k not in self.c:
            C = bn.num_features
            scale = torch.empty(C, dtype=torch.float32, device=bn.weight.device)
            shift = torch.empty(C, dtype=torch.float32, device=bn.weight.device)
            ops.bn_fold(bn, scale, shift)
            self.c[k] = (scale, shift)
        return self.c[k]

    def pair(self, conv_a, bn_a, conv_b, bn_b):
        """Twin layers of the RGB and depth encoders as ONE image-grouped conv: weights stacked (2, Cout, Cin, k, k),
        folded BatchNorm scale / shift stacked (2*Cout) - set g serves images [g*B, (g+1)*B) (ivln_gemm_desc.grp_imgs)."""
        k = ("pair", id(conv_a))
        if k not in self.c:
            (sa, ba), (sb, bb) = self.bn(bn_a), self.bn(bn_b)
            w = torch.stack([conv_a.weight.detach(), conv_b.weight.detach()]).contiguous()
            self.c[k] = (w, torch.cat([sa, sb]).contiguous(), torch.cat([ba, bb]).contiguous())
        return self.c[k]

    def classes(self, convt: nn.ConvTranspose2d):
        """Output-parity decomposition of a stride-2 transposed conv (None when not applicable)."""
        k = ("cls", id(convt))
        if k not in self.c:
            ok = convt.stride == (2, 2) and convt.kernel_size[0] == convt.kernel_size[1] and \
                2 - 2 * convt.padding[0] + convt.kernel_size[0] - 2 + convt.output_padding[0] == 2
            cls = ops.convt_s2_classes(convt.weight.detach(), convt.padding[0]) if ok else None
            self.c[k] = None if cls is None else (cls, ops.convt_s2_stack(cls))
        return self.c[k]

    def wt(self, convt: nn.ConvTranspose2d):
        k = id(convt)
        if k not in self.c:
            self.c[k] = convt.weight.detach().permute(1, 0, 2, 3).contiguous()  # (Cin,Cout,k,k) -> (Cout,Cin,k,k)
        return self.c[k]


def _convt(x, convt: nn.ConvTranspose2d, f: _Folded, **epi):
    """Stride-2 upsampling convs run as four output-parity sub-convolutions; anything else through the
    transposed-gather operand mode."""
    cls = f.classes(convt)
    if cls is not None:
        return ops.conv_transpose2d_s2(x, cls[0], stacked=cls[1], **epi)
    return ops.conv_transpose2d(x, f.wt(convt), convt.stride[0], convt.padding[0], convt.output_padding[0], **epi)


# Called with a stage name ("stem", "layer1" .. "layer4", "deconv1" .. "deconv4") when the launches of that stage are all
# enqueued: graphed.GraphedRollout cuts its capture of the step there, so that the policy's depth encoder (another graph, on
# the side stream) starts beside RedNet's pixel-starved stages instead of its chip-filling first ones.  None = no-op.
STAGE_HOOK = None


def _stage_done(name):
    if STAGE_HOOK is not None:
        STAGE_HOOK(name)


SKIP_ADD_FUSED = os.environ.get("IVLN_REDNET_SKIP_ADD", "1") != "0"  # A/B: 0 = the decoder's skip adds as launches of their own


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward_hip(self, x, f: _Folded):
        residual = x
        if self.downsample is not None:
            s, b = f.bn(self.downsample[1])
            residual = ops.conv2d(x, self.downsample[0].weight, stride=self.stride, scale=s, shift=b)
        s, b = f.bn(self.bn1)
        y = ops.conv2d(x, self.conv1.weight, scale=s, shift=b, relu=True)
        s, b = f.bn(self.bn2)
        s3, b3 = f.bn(self.bn3)
        if self.stride == 1:
            out = ops.conv3x3_then_1x1(y, self.conv2.weight, s, b, self.conv3.weight, s3, b3, residual)
            if out is not None:
                return out
        y = ops.conv2d(y, self.conv2.weight, stride=self.stride, pad=1, scale=s, shift=b, relu=True)
        return ops.conv2d(y, self.conv3.weight, scale=s3, shift=b3, residual=residual, relu=True)


def _conv_pair(x2, w2, s2, b2, stride=1, pad=0, residual=None, relu=False):
    """Image-grouped conv of the stacked [RGB ; depth] batch; tiny feature maps (fewer than 32 pixels per weight
    set: an output tile would straddle the sets) run as two plain convs into the halves of one output."""
    N, _, H, W = x2.shape
    G, Cout, _, KH, KW = w2.shape
    B = N // G
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    if (B * Ho * Wo) % 32 == 0:
        return ops.conv2d(x2, w2, stride=stride, pad=pad, scale=s2, shift=b2, residual=residual, relu=relu)
    out = torch.empty((N, Cout, Ho, Wo), dtype=torch.float32, device=x2.device)
    for g in range(G):
        sl = slice(g * B, (g + 1) * B)
        ops.conv2d(x2[sl], w2[g], stride=stride, pad=pad, scale=s2[g * Cout:(g + 1) * Cout], shift=b2[g * Cout:(g + 1) * Cout],
                   residual=residual[sl] if residual is not None else None, relu=relu, out=out[sl])
    return out


def _bottleneck_pair(blk: Bottleneck, blk_d: Bottleneck, x2, f: _Folded):
    """The same bottleneck of the RGB encoder (first half of the stacked batch) and of the depth encoder (second
    half) in one pass: 3-4 launches instead of 6-8, every launch with twice the output tiles."""
    residual = x2
    if blk.downsample is not None:
        w, s, b = f.pair(blk.downsample[0], blk.downsample[1], blk_d.downsample[0], blk_d.downsample[1])
        residual = _conv_pair(x2, w, s, b, stride=blk.stride)
    w, s, b = f.pair(blk.conv1, blk.bn1, blk_d.conv1, blk_d.bn1)
    y = _conv_pair(x2, w, s, b, relu=True)
    w, s, b = f.pair(blk.conv2, blk.bn2, blk_d.conv2, blk_d.bn2)
    w3, s3, b3 = f.pair(blk.conv3, blk.bn3, blk_d.conv3, blk_d.bn3)
    if blk.stride == 1:  # layers 1-2: conv2 + conv3 (+ residual + ReLU) as one launch, the narrow tensor stays on the CU
        out = ops.conv3x3_then_1x1(y, w, s, b, w3, s3, b3, residual)
        if out is not None:
            return out
    y = _conv_pair(y, w, s, b, stride=blk.stride, pad=1, relu=True)
    return _conv_pair(y, w3, s3, b3, residual=residual, relu=True)


class TransBasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, upsample=None, **kwargs):
        super().__init__()
        self.conv1 = _conv3x3(inplanes, inplanes)
        self.bn1 = nn.BatchNorm2d(inplanes)
        self.relu = nn.ReLU(inplace=True)
        if upsample is not None and stride != 1:
            self.conv2 = nn.ConvTranspose2d(inplanes, planes, kernel_size=3, stride=stride, padding=1,
                                            output_padding=1, bias=False)
        else:
            self.conv2 = _conv3x3(inplanes, planes, stride)
        self.bn2 = nn.BatchNorm2d(planes)
        self.upsample = upsample
        self.stride = stride

    def forward_hip(self, x, f: _Folded):
        residual = x
        if self.upsample is not None:
            s, b = f.bn(self.upsample[1])
            up = self.upsample[0]
            if isinstance(up, nn.ConvTranspose2d):
                residual = _convt(x, up, f, scale=s, shift=b)
            else:
                residual = ops.conv2d(x, up.weight, scale=s, shift=b)
        s, b = f.bn(self.bn1)
        y = ops.conv2d(x, self.conv1.weight, pad=1, scale=s, shift=b, relu=True)
        s, b = f.bn(self.bn2)
        if isinstance(self.conv2, nn.ConvTranspose2d):
            return _convt(y, self.conv2, f, scale=s, shift=b, residual=residual, relu=True)
        return ops.conv2d(y, self.conv2.weight, stride=self.stride, pad=1, scale=s, shift=b, residual=residual,
                          relu=True)


class RedNet(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        num_classes = cfg["n_classes"]
        block, transblock, layers = Bottleneck, TransBasicBlock, [3, 4, 6, 3]
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.inplanes = 64
        self.conv1_d = nn.Conv2d(1, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1_d = nn.BatchNorm2d(64)
        self.layer1_d = self._make_layer(block, 64, layers[0])
        self.layer2_d = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3_d = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4_d = self._make_layer(block, 512, layers[3], stride=2)
        self.inplanes = 512
        self.deconv1 = self._make_transpose(transblock, 256, 6, stride=2)
        self.deconv2 = self._make_transpose(transblock, 128, 4, stride=2)
        self.deconv3 = self._make_transpose(transblock, 64, 3, stride=2)
        self.deconv4 = self._make_transpose(transblock, 64, 3, stride=2)
        self.agant0 = self._make_agant_layer(64, 64)
        self.agant1 = self._make_agant_layer(64 * 4, 64)
        self.agant2 = self._make_agant_layer(128 * 4, 128)
        self.agant3 = self._make_agant_layer(256 * 4, 256)
        self.agant4 = self._make_agant_layer(512 * 4, 512)
        self.inplanes = 64
        self.final_conv = self._make_transpose(transblock, 64, 3)
        self.final_deconv_custom = nn.ConvTranspose2d(self.inplanes, num_classes, kernel_size=2, stride=2, padding=0,
                                                      bias=True)
        # training-only auxiliary heads (kept so the released state_dict loads strictly)
        self.out5_conv_custom = nn.Conv2d(256, num_classes, kernel_size=1, stride=1, bias=True)
        self.out4_conv_custom = nn.Conv2d(128, num_classes, kernel_size=1, stride=1, bias=True)
        self.out3_conv_custom = nn.Conv2d(64, num_classes, kernel_size=1, stride=1, bias=True)
        self.out2_conv_custom = nn.Conv2d(64, num_classes, kernel_size=1, stride=1, bias=True)
        self._folded = _Folded()

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion),
            )
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def _make_transpose(self, block, planes, blocks, stride=1):
        upsample = None
        if stride != 1:
            upsample = nn.Sequential(
                nn.ConvTranspose2d(self.inplanes, planes, kernel_size=2, stride=stride, padding=0, bias=False),
                nn.BatchNorm2d(planes),
            )
        elif self.inplanes != planes:
            upsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes, kernel_size=1, stride=stride, bias=False), nn.BatchNorm2d(planes)
            )
        layers = [block(self.inplanes, self.inplanes) for _ in range(1, blocks)]
        layers.append(block(self.inplanes, planes, stride, upsample))
        self.inplanes = planes
        return nn.Sequential(*layers)

    def _make_agant_layer(self, inplanes, planes):
        return nn.Sequential(
            nn.Conv2d(inplanes, planes, kernel_size=1, stride=1, padding=0, bias=False),
            nn.BatchNorm2d(planes), nn.ReLU(inplace=True),
        )

    def invalidate_folded(self):
        """Call after changing weights (load_state_dict / .to()) so BN folds are recomputed."""
        self._folded = _Folded()

    def _seq(self, seq, x):
        for blk in seq:
            x = blk.forward_hip(x, self._folded)
        return x

    def _agant(self, layer, x):
        s, b = self._folded.bn(layer[1])
        return ops.conv2d(x, layer[0].weight, scale=s, shift=b, relu=True)

    def _skip_add(self, up, layer, fuse):
        """up + agant(fuse) (rednet.py:244-263): the 1x1 skip conv with the decoder tensor added behind its ReLU in the same
        launch where the library has that epilogue, else conv + add."""
        s, b = self._folded.bn(layer[1])
        out = ops.conv2d(fuse, layer[0].weight, scale=s, shift=b, residual=up, relu=True, residual_after_relu=True) if SKIP_ADD_FUSED else None
        return out if out is not None else ops.add(up, self._agant(layer, fuse))

    def forward(self, rgb, depth):
        """rgb (B,3,H,W) normalised, depth (B,1,H,W) normalised -> scores (B,classes,H,W)
        (rednet.py:190-269, eval path)."""
        assert not self.training, "HIP RedNet is inference-only (the reference freezes it, mapper.py:751-752)"
        f = self._folded
        if GROUP_ENCODERS:
            # The two encoders are the same ResNet-50 with different weights, coupled only by the five fusion adds
            # (rednet.py:190-222): stacked on the image axis - [RGB branch ; depth branch] - every layer is ONE
            # image-grouped launch.  A fusion add is done in place on the first half, which then IS the RGB branch's
            # next input (and the decoder's skip tensor).
            B = rgb.shape[0]
            S = torch.empty((2 * B, 64, rgb.shape[2] // 2, rgb.shape[3] // 2), dtype=torch.float32, device=rgb.device)
            s, b = f.bn(self.bn1)
            sd, bd = f.bn(self.bn1_d)
            ops.conv2d(depth, self.conv1_d.weight, stride=2, pad=3, scale=sd, shift=bd, relu=True, out=S[B:])
            # fuse0 = relu(bn1(conv1(rgb))) + the depth stem's output (rednet.py:196): added behind the RGB stem's ReLU in its own
            # epilogue where the library has the stem kernel (256 x 256 inputs), else conv + add
            fuse0 = ops.conv2d(rgb, self.conv1.weight, stride=2, pad=3, scale=s, shift=b, relu=True, out=S[:B], residual=S[B:],
                               residual_after_relu=True) if ops.BF3_STEM else None
            if fuse0 is None:
                ops.conv2d(rgb, self.conv1.weight, stride=2, pad=3, scale=s, shift=b, relu=True, out=S[:B])
                fuse0 = ops.add(S[:B], S[B:], out=S[:B])
            P = ops.pool2d(S, 3, 2, 1, "max")
            fuses = []
            _stage_done("stem")
            for li, (la, lb) in enumerate(((self.layer1, self.layer1_d), (self.layer2, self.layer2_d),
                                           (self.layer3, self.layer3_d), (self.layer4, self.layer4_d))):
                for blk, blk_d in zip(la, lb):
                    P = _bottleneck_pair(blk, blk_d, P, f)
                fuses.append(ops.add(P[:B], P[B:], out=P[:B]))
                _stage_done(f"layer{li + 1}")
            fuse1, fuse2, fuse3, fuse4 = fuses
        else:
            s, b = f.bn(self.bn1)
            x = ops.conv2d(rgb, self.conv1.weight, stride=2, pad=3, scale=s, shift=b, relu=True)
            s, b = f.bn(self.bn1_d)
            d = ops.conv2d(depth, self.conv1_d.weight, stride=2, pad=3, scale=s, shift=b, relu=True)
            fuse0 = ops.add(x, d)
            x = ops.pool2d(fuse0, 3, 2, 1, "max")
            d = ops.pool2d(d, 3, 2, 1, "max")
            x, d = self._seq(self.layer1, x), self._seq(self.layer1_d, d)
            fuse1 = ops.add(x, d)
            x, d = self._seq(self.layer2, fuse1), self._seq(self.layer2_d, d)
            fuse2 = ops.add(x, d)
            x, d = self._seq(self.layer3, fuse2), self._seq(self.layer3_d, d)
            fuse3 = ops.add(x, d)
            x, d = self._seq(self.layer4, fuse3), self._seq(self.layer4_d, d)
            fuse4 = ops.add(x, d)
        x = self._agant(self.agant4, fuse4)
        for di, (seq, ag, fuse) in enumerate(((self.deconv1, self.agant3, fuse3), (self.deconv2, self.agant2, fuse2),
                                              (self.deconv3, self.agant1, fuse1), (self.deconv4, self.agant0, fuse0))):
            x = self._skip_add(self._seq(seq, x), ag, fuse)
            _stage_done(f"deconv{di + 1}")
        x = self._seq(self.final_conv, x)
        fd = self.final_deconv_custom
        return _convt(x, fd, f, shift=fd.bias)


class PredictSemantics:
    """mapper.py:703-800: labels = argmax_c RedNet(normalised rgb, normalised depth) as u8."""

    CFG = {
        "arch": "rednet", "resnet_pretrained": False, "finetune": True, "SUNRGBD_pretrained_weights": "",
        "n_classes": 13, "upsample_prediction": True, "load_model": "data/rednet_mp3d_best_model.pkl",
    }

    def __init__(self, device, model: RedNet = None, load_weights: bool = True):
        self.device = device
        self.model = model
        self.load_weights = load_weights

    def setup(self):
        if self.model is None:
            self.model = RedNet(self.CFG)
            path = self.CFG["load_model"]
            if self.load_weights and os.path.exists(path):
                state = torch.load(path, map_location="cpu")["model_state"]
                if next(iter(state)).split(".")[0] == "module":  # convert_weights_cuda_cpu, mapper.py:767-779
                    state = {".".join(k.split(".")[1:]): v for k, v in state.items()}
                self.model.load_state_dict(state)
            self.model = self.model.to(self.device).eval()
            for p in self.model.parameters():
                p.requires_grad = False

    @torch.no_grad()
    def scores(self, observations):
        if observations.get("rgb", None) is None:
            raise Exception("RGB Sensor not in use")  # mapper.py:783-784
        self.setup()
        depth = observations["depth"].to(torch.float32).contiguous()  # (B,H,W,1)
        B, H, W, _ = depth.shape
        rgb = ops.rgb_resize_normalize(observations["rgb"].to(torch.uint8).contiguous(), H, W)
        dn = ops.affine(depth.view(B, 1, H, W), 0.213, 0.285)
        return self.model(rgb, dn)

    # -- the forward as ONE C call ---------------------------------------------------------------------------------
    # (ivln_rednet_fwd).  The first step of a given batch shape runs the layer walk above with the recorder on: every
    # launch lands in a packed table with its pointers resolved (weights, folded BN, activation buffers, split-K
    # workspaces - all kept alive by the plan); later steps hand that table to the library with this step's frames.
    # What it removes is the Python between ~170 launches (~1.5 ms of host time per forward), which
    # only an eager rollout pays - a captured step replays the launches without host code anyway, so capture takes the
    # layer walk.  A/B: IVLN_REDNET_PLAN=0.
    USE_PLAN = os.environ.get("IVLN_REDNET_PLAN", "1") != "0"

    def _plan_key(self, rgb, depth):
        return (tuple(rgb.shape), tuple(depth.shape), str(depth.device), ops.stream_ptr(),
                ops.WEIGHT_EPOCH, ops.TILE_OVERRIDE)

    def __call__(self, observations):
        if not self.USE_PLAN or torch.cuda.is_current_stream_capturing():
            return ops.argmax_channels_u8(self.scores(observations))
        if observations.get("rgb", None) is None:
            raise Exception("RGB Sensor not in use")  # mapper.py:783-784
        self.setup()
        depth = observations["depth"].to(torch.float32).contiguous()
        rgb = observations["rgb"].to(torch.uint8).contiguous()
        B, H, W, _ = depth.shape
        labels = torch.empty((B, 1, H, W), dtype=torch.uint8, device=depth.device)
        plans = self.__dict__.setdefault("_plans", {})
        key = self._plan_key(rgb, depth)
        plan = plans.get(key)
        if plan is None:
            with ops.recording() as rec, torch.no_grad():
                first = ops.argmax_channels_u8(self.scores({"rgb": rgb, "depth": depth}))
            if len(plans) > 8:
                plans.clear()
            plans[key] = (rec.table(), len(rec.ops), rec.keep)
            return first
        ops.rednet_fwd(plan[0], plan[1], rgb, depth, labels)
        return labels
