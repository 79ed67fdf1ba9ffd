"""Image branch: ResNet-34 trunk truncated after layer2.

Interface parity with the reference (model/Img_Encoder.py:9-18, model/resnet.py:118-224):
`ImageEncoder().backbone` is a torchvision-layout ResNet-34 (parameter names conv1, bn1,
layer{1..4}.{i}.conv{1,2} / bn{1,2} / downsample.{0,1}, fc), so reference checkpoints -- which store
all 218 backbone tensors although only conv1..layer2 are ever executed (resnet.py:205-216) -- load
with strict=True.  `forward` returns the stride-8 map [B,128,H/8,W/8].

Unlike the reference, construction never touches the network (resnet.py:222 downloads ImageNet
weights that the checkpoint then overwrites).
"""
import torch.nn as nn
import torch.nn.functional as F


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return F.relu(y + (x if self.downsample is None else self.downsample(x)))


class ResNet(nn.Module):
    """torchvision ResNet skeleton; `forward` stops after layer2 (reference resnet.py:195-216)."""

    def __init__(self, in_channels=3, layers=(3, 4, 6, 3), num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._stage(64, layers[0], 1)
        self.layer2 = self._stage(128, layers[1], 2)
        self.layer3 = self._stage(256, layers[2], 2)      # stored in checkpoints, never executed
        self.layer4 = self._stage(512, layers[3], 2)      # "
        self.fc = nn.Linear(512, num_classes)             # "
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')

    def _stage(self, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, stride, bias=False),
                                 nn.BatchNorm2d(planes))
        mods = [BasicBlock(self.inplanes, planes, stride, down)]
        self.inplanes = planes
        mods += [BasicBlock(planes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*mods)

    def forward(self, x):
        x = self.maxpool(F.relu(self.bn1(self.conv1(x))))
        return self.layer2(self.layer1(x))


def resnet34(in_channels=3, pretrained=False, progress=True, **kwargs):
    """`pretrained` is accepted for signature parity and ignored: weights come from the IMFNet
    checkpoint's state_dict (scripts/generate_desc.py:174)."""
    return ResNet(in_channels, (3, 4, 6, 3), **kwargs)


class ImageEncoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.backbone = resnet34(in_channels=3, pretrained=False)

    def forward(self, x):
        return self.backbone(x)
