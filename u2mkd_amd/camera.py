"""SwiftNet-18 camera branch (row a13; core/models/image_branch/swiftnet.py:20-50, 114-341).

Plain torch.nn: the dense 2D convolutions ride PyTorch-ROCm / MIOpen as the north star
prescribes.  ResNet-18 encoder whose 7x7 stem has stride 1 (feature
maps are H/2 after the stem), pre-activation lateral skips, a spatial pyramid pooling
bottleneck (grids 8/4/2/1 scaled by the aspect ratio, BN momentum 0.012) and a 3-level
up-sampling decoder.  Module / parameter names equal the reference's, so its ImageNet and
U2MKD checkpoints load unchanged."""
import torch
import torch.nn.functional as F
from torch import nn

__all__ = ['SwiftNetRes18', 'SwiftNetResNet', 'BNReluConv']


def _up(x, size):
    return F.interpolate(x, size, mode='bilinear', align_corners=True)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        out = self.bn2(self.conv2(self.relu(self.bn1(self.conv1(x)))))
        out = out + (x if self.downsample is None else self.downsample(x))
        return self.relu(out), out          # (activated, pre-activation skip)


class BNReluConv(nn.Sequential):
    def __init__(self, cin, cout, k=3, bn_momentum=0.1):
        super().__init__()
        self.add_module('norm', nn.BatchNorm2d(cin, momentum=bn_momentum))
        self.add_module('relu', nn.ReLU(inplace=True))
        self.add_module('conv', nn.Conv2d(cin, cout, kernel_size=k, padding=k // 2, bias=False))


class SpatialPyramidPooling(nn.Module):
    def __init__(self, cin, num_levels, bt_size, level_size, out_size, grids, bn_momentum):
        super().__init__()
        self.grids = grids
        self.spp = nn.Sequential()
        self.spp.add_module('spp_bn', BNReluConv(cin, bt_size, k=1, bn_momentum=bn_momentum))
        final = bt_size
        for i in range(num_levels):
            final += level_size
            self.spp.add_module('spp' + str(i), BNReluConv(bt_size, level_size, k=1, bn_momentum=bn_momentum))
        self.spp.add_module('spp_fuse', BNReluConv(final, out_size, k=1, bn_momentum=bn_momentum))

    def forward(self, x):
        h, w = x.shape[2:4]
        ar = w / h
        x = self.spp[0](x)
        levels = [x]
        for i in range(1, len(self.spp) - 1):
            grid = (self.grids[i - 1], max(1, round(ar * self.grids[i - 1])))
            levels.append(_up(self.spp[i](F.adaptive_avg_pool2d(x, grid)), (h, w)))
        return self.spp[-1](torch.cat(levels, 1))


class _Upsample(nn.Module):
    def __init__(self, cin, cskip, cout, k=3):
        super().__init__()
        self.bottleneck = BNReluConv(cskip, cin, k=1)
        self.blend_conv = BNReluConv(cin, cout, k=k)

    def forward(self, x, skip):
        skip = self.bottleneck(skip)
        return self.blend_conv(_up(x, skip.shape[2:4]) + skip)


class SwiftNetResNet(nn.Module):
    def __init__(self, layers=(2, 2, 2, 2), num_features=(128, 128, 128), spp_grids=(8, 4, 2, 1)):
        super().__init__()
        self.inplanes = 64
        self.img_cs = [64, 64, 128, 256, num_features[0]]
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=1, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        skips = []
        self.layer1 = self._make_layer(64, layers[0])
        skips.append(self.inplanes)
        self.layer2 = self._make_layer(128, layers[1], stride=2)
        skips.append(self.inplanes)
        self.layer3 = self._make_layer(256, layers[2], stride=2)
        skips.append(self.inplanes)
        self.layer4 = self._make_layer(512, layers[3], stride=2)
        spp_size = num_features[0]
        self.spp = SpatialPyramidPooling(self.inplanes, 3, bt_size=spp_size, level_size=spp_size // 3,
                                         out_size=num_features[0], grids=spp_grids, bn_momentum=0.024 / 2)
        ups = [_Upsample(num_features[1], skips[0], num_features[2]),
               _Upsample(num_features[0], skips[1], num_features[1]),
               _Upsample(num_features[0], skips[2], num_features[0])]
        self.upsample = nn.ModuleList(list(reversed(ups)))
        self.num_features = num_features[-1]
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes, kernel_size=1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes))
        layers = [BasicBlock(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes
        layers += [BasicBlock(planes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    @staticmethod
    def forward_resblock(x, layers):
        skip = None
        for layer in layers:
            x, skip = layer(x)
        return x, skip

    def forward_stem(self, image):
        return self.maxpool(self.relu(self.bn1(self.conv1(image))))

    def forward_down(self, image):
        x = self.forward_stem(image)
        feats = []
        for layer in (self.layer1, self.layer2, self.layer3):
            x, skip = self.forward_resblock(x, layer)
            feats.append(skip)
        x, skip = self.forward_resblock(x, self.layer4)
        feats.append(self.spp(skip))
        return feats

    def forward_up(self, features, im_size=None):
        features = features[::-1]
        x = features[0]
        for skip, up in zip(features[1:], self.upsample):
            x = up(x, skip)
        return _up(x, im_size) if im_size is not None else x

    def forward(self, image, im_size=None):
        return self.forward_up(self.forward_down(image), im_size=im_size)


def SwiftNetRes18(num_feature=(128, 128, 128), pretrained_path=None):
    model = SwiftNetResNet((2, 2, 2, 2), num_feature)
    if pretrained_path is not None:
        model.load_state_dict(torch.load(pretrained_path, map_location='cpu'), strict=False)
    return model
