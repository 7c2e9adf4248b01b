"""TEST INFRASTRUCTURE ONLY - plain-PyTorch (CPU fp32) restatement of RedNet and PredictSemantics
(/root/reference/ivlnce_baselines/common/mapping_module/rednet.py:7-358, mapper.py:665-800); same
state_dict keys.  Pinned by tests/test_oracle_rednet.py against tests/golden/rednet.npz, produced
by the reference's own RedNet (tests/golden/gen_rednet_golden.py).  Never imported by the product."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _Bottleneck(nn.Module):
    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1, self.bn1 = nn.Conv2d(cin, planes, 1, bias=False), nn.BatchNorm2d(planes)
        self.conv2, self.bn2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False), nn.BatchNorm2d(planes)
        self.conv3, self.bn3 = nn.Conv2d(planes, planes * 4, 1, bias=False), nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        r = x if self.downsample is None else self.downsample(x)
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        return F.relu(self.bn3(self.conv3(y)) + r)


class _TransBlock(nn.Module):
    def __init__(self, cin, planes, stride=1, upsample=None):
        super().__init__()
        self.conv1, self.bn1 = nn.Conv2d(cin, cin, 3, 1, 1, bias=False), nn.BatchNorm2d(cin)
        if upsample is not None and stride != 1:
            self.conv2 = nn.ConvTranspose2d(cin, planes, 3, stride, 1, 1, bias=False)
        else:
            self.conv2 = nn.Conv2d(cin, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.upsample = upsample

    def forward(self, x):
        r = x if self.upsample is None else self.upsample(x)
        y = F.relu(self.bn1(self.conv1(x)))
        return F.relu(self.bn2(self.conv2(y)) + r)


class RedNetRef(nn.Module):
    def __init__(self, num_classes=13):
        super().__init__()
        L = [3, 4, 6, 3]
        self.inplanes = 64
        self.conv1, self.bn1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64)
        self.layer1, self.layer2 = self._layer(64, L[0]), self._layer(128, L[1], 2)
        self.layer3, self.layer4 = self._layer(256, L[2], 2), self._layer(512, L[3], 2)
        self.inplanes = 64
        self.conv1_d, self.bn1_d = nn.Conv2d(1, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64)
        self.layer1_d, self.layer2_d = self._layer(64, L[0]), self._layer(128, L[1], 2)
        self.layer3_d, self.layer4_d = self._layer(256, L[2], 2), self._layer(512, L[3], 2)
        self.inplanes = 512
        self.deconv1, self.deconv2 = self._trans(256, 6, 2), self._trans(128, 4, 2)
        self.deconv3, self.deconv4 = self._trans(64, 3, 2), self._trans(64, 3, 2)
        self.agant0, self.agant1, self.agant2 = self._agant(64, 64), self._agant(256, 64), self._agant(512, 128)
        self.agant3, self.agant4 = self._agant(1024, 256), self._agant(2048, 512)
        self.inplanes = 64
        self.final_conv = self._trans(64, 3)
        self.final_deconv_custom = nn.ConvTranspose2d(64, num_classes, 2, 2, 0, bias=True)
        self.out5_conv_custom = nn.Conv2d(256, num_classes, 1)
        self.out4_conv_custom = nn.Conv2d(128, num_classes, 1)
        self.out3_conv_custom = nn.Conv2d(64, num_classes, 1)
        self.out2_conv_custom = nn.Conv2d(64, num_classes, 1)

    def _layer(self, planes, blocks, stride=1):
        ds = None
        if stride != 1 or self.inplanes != planes * 4:
            ds = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))
        layers = [_Bottleneck(self.inplanes, planes, stride, ds)]
        self.inplanes = planes * 4
        layers += [_Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def _trans(self, planes, blocks, stride=1):
        up = None
        if stride != 1:
            up = nn.Sequential(nn.ConvTranspose2d(self.inplanes, planes, 2, stride, 0, bias=False), nn.BatchNorm2d(planes))
        elif self.inplanes != planes:
            up = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
        layers = [_TransBlock(self.inplanes, self.inplanes) for _ in range(1, blocks)]
        layers.append(_TransBlock(self.inplanes, planes, stride, up))
        self.inplanes = planes
        return nn.Sequential(*layers)

    @staticmethod
    def _agant(cin, planes):
        return nn.Sequential(nn.Conv2d(cin, planes, 1, bias=False), nn.BatchNorm2d(planes), nn.ReLU(inplace=True))

    def forward(self, rgb, depth):
        x = F.relu(self.bn1(self.conv1(rgb)))
        d = F.relu(self.bn1_d(self.conv1_d(depth)))
        f0 = x + d
        x, d = F.max_pool2d(f0, 3, 2, 1), F.max_pool2d(d, 3, 2, 1)
        x, d = self.layer1(x), self.layer1_d(d)
        f1 = x + d
        x, d = self.layer2(f1), self.layer2_d(d)
        f2 = x + d
        x, d = self.layer3(f2), self.layer3_d(d)
        f3 = x + d
        x, d = self.layer4(f3), self.layer4_d(d)
        x = self.agant4(x + d)
        x = self.deconv1(x) + self.agant3(f3)
        x = self.deconv2(x) + self.agant2(f2)
        x = self.deconv3(x) + self.agant1(f1)
        x = self.deconv4(x) + self.agant0(f0)
        return self.final_deconv_custom(self.final_conv(x))


def predict_semantics_ref(net, rgb_u8_nhwc, depth_nhwc):
    """mapper.py:781-800 (rgb/255 -> bilinear to depth size -> ImageNet normalise; depth normalise)."""
    H, W = depth_nhwc.shape[1], depth_nhwc.shape[2]
    rgb = F.interpolate(rgb_u8_nhwc.permute(0, 3, 1, 2).float() / 255.0, size=(H, W), mode="bilinear")
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    rgb = (rgb - mean) / std
    dep = (depth_nhwc.permute(0, 3, 1, 2) - 0.213) / 0.285
    with torch.no_grad():
        scores = net(rgb, dep)
    return scores, scores.argmax(1, keepdim=True).to(torch.uint8), rgb
