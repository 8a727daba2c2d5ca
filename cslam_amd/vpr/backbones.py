"""Convolutional backbones as plain torch modules with torchvision's module/parameter names
(conv1, bn1, layer1.0.conv1, downsample.0, ..., features.N), cut where the reference cuts them
(network.py:58-97: ResNet without avgpool/fc; netvlad.py:150-159: VGG16 features without the
last ReLU and max-pool).  torchvision is not available in this image and is not needed."""
import torch.nn as nn


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
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + identity)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + identity)


_RESNETS = {"ResNet18": (BasicBlock, [2, 2, 2, 2]), "ResNet50": (Bottleneck, [3, 4, 6, 3]),
            "ResNet101": (Bottleneck, [3, 4, 23, 3]), "ResNet152": (Bottleneck, [3, 8, 36, 3])}


def resnet_trunk(name):
    """Sequential(conv1, bn1, relu, maxpool, layer1..layer4) — `list(resnet.children())[:-2]` of
    the torchvision model (network.py:69), so state-dict keys read `0.weight`, `4.0.conv1.weight`,
    ...  Returns (module, channels of the last conv)."""
    block, layers = _RESNETS[name]
    inplanes = 64

    def make_layer(planes, blocks, stride):
        nonlocal inplanes
        downsample = None
        if stride != 1 or inplanes != planes * block.expansion:
            downsample = nn.Sequential(nn.Conv2d(inplanes, planes * block.expansion, 1, stride, bias=False),
                                       nn.BatchNorm2d(planes * block.expansion))
        mods = [block(inplanes, planes, stride, downsample)]
        inplanes = planes * block.expansion
        mods += [block(inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*mods)

    trunk = nn.Sequential(nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
                          nn.MaxPool2d(3, 2, 1), make_layer(64, layers[0], 1), make_layer(128, layers[1], 2),
                          make_layer(256, layers[2], 2), make_layer(512, layers[3], 2))
    return trunk, 512 * block.expansion


def vgg16_features(cut_last=2):
    """torchvision's vgg16().features without its last `cut_last` modules."""
    cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]
    mods, cin = [], 3
    for v in cfg:
        if v == "M":
            mods.append(nn.MaxPool2d(2, 2))
        else:
            mods += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return nn.Sequential(*mods[:len(mods) - cut_last]), 512
