"""Oracle: pure torch.nn CPU restatement of the U-Net the reference instantiates.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED for this file: the
network lives in the un-vendored dependency `segmentation_models_pytorch`
(reference call sites: d3f/train_denoiser/lit_module.py:46-52,
d3f/train_deep_fake/lit_module.py:53-59); it is restated here from the published
structure of smp.Unet(encoder_name="resnet34") + torchvision ResNet-34
(SURVEY.md Appendix A.1).  Cross-checks available offline: 24 436 659 parameters,
output shape == input shape, H,W % 32 == 0 required, smp state_dict key names.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class BasicBlock(nn.Module):
    """torchvision.models.resnet.BasicBlock (expansion 1)."""

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(
                nn.Conv2d(inplanes, planes, 1, stride, bias=False),
                nn.BatchNorm2d(planes),
            )

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class ResNet34Encoder(nn.Module):
    """smp ResNetEncoder(resnet34, depth=5): returns 6 feature maps."""

    def __init__(self, in_channels=3):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cfg = [(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]
        inplanes = 64
        for li, (planes, blocks, stride) in enumerate(cfg, start=1):
            layers = [BasicBlock(inplanes, planes, stride)]
            inplanes = planes
            layers += [BasicBlock(planes, planes) for _ in range(1, blocks)]
            setattr(self, f"layer{li}", nn.Sequential(*layers))
        # torchvision init
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        f0 = x
        f1 = self.relu(self.bn1(self.conv1(x)))
        f2 = self.layer1(self.maxpool(f1))
        f3 = self.layer2(f2)
        f4 = self.layer3(f3)
        f5 = self.layer4(f4)
        return [f0, f1, f2, f3, f4, f5]


class Conv2dReLU(nn.Sequential):
    """smp.base.modules.Conv2dReLU(use_batchnorm=True): (conv, bn, relu)."""

    def __init__(self, cin, cout):
        super().__init__(
            nn.Conv2d(cin, cout, 3, padding=1, bias=False),
            nn.BatchNorm2d(cout),
            nn.ReLU(inplace=True),
        )


class DecoderBlock(nn.Module):
    def __init__(self, cin, cskip, cout):
        super().__init__()
        self.conv1 = Conv2dReLU(cin + cskip, cout)
        self.attention1 = nn.Identity()
        self.conv2 = Conv2dReLU(cout, cout)
        self.attention2 = nn.Identity()

    def forward(self, x, skip=None):
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        if skip is not None:
            x = torch.cat([x, skip], dim=1)
        x = self.conv1(x)
        return self.conv2(x)


class UnetDecoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.center = nn.Identity()
        spec = [(512, 256, 256), (256, 128, 128), (128, 64, 64), (64, 64, 32), (32, 0, 16)]
        self.blocks = nn.ModuleList([DecoderBlock(*s) for s in spec])
        # smp.base.initialization.initialize_decoder
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, mode="fan_in", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, *features):
        features = features[1:][::-1]
        x = self.center(features[0])
        skips = features[1:]
        for i, blk in enumerate(self.blocks):
            x = blk(x, skips[i] if i < len(skips) else None)
        return x


class Unet(nn.Module):
    """Unet(encoder_name, encoder_weights, in_channels, classes, activation) -- the
    constructor signature the reference uses (train_denoiser/lit_module.py:46-52)."""

    def __init__(self, encoder_name="resnet34", encoder_weights=None, in_channels=3,
                 classes=3, activation=None):
        super().__init__()
        if encoder_name != "resnet34":
            raise KeyError(f"Wrong encoder name `{encoder_name}`, supported encoders: ['resnet34']")
        if encoder_weights is not None:
            raise KeyError("encoder_weights must be None (no pretrained weights offline)")
        if activation is not None:
            raise ValueError("activation must be None")
        self.encoder = ResNet34Encoder(in_channels)
        self.decoder = UnetDecoder()
        self.segmentation_head = nn.Sequential(
            nn.Conv2d(16, classes, 3, padding=1), nn.Identity(), nn.Identity())
        nn.init.xavier_uniform_(self.segmentation_head[0].weight)
        nn.init.constant_(self.segmentation_head[0].bias, 0)

    def forward(self, x):
        h, w = x.shape[-2:]
        if h % 32 != 0 or w % 32 != 0:
            raise RuntimeError(
                f"Wrong input shape height={h}, width={w}. Expected image height and "
                f"width divisible by 32.")
        return self.segmentation_head(self.decoder(*self.encoder(x)))


def count_parameters(m):
    return sum(p.numel() for p in m.parameters())
