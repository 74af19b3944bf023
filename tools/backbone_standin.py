"""HARNESS ONLY -- a stock-PyTorch stand-in for the reference's image branch (ResNet-50 + FPN, configs/
r50_nuimg_704x256.py:31-45), random-init, fp16 autocast, channels-last.  torchvision / mmdet are not in the image, so
the architecture is restated with plain torch.nn (convolutions run on MIOpen).  It exists to produce FPN features for
the LABELLED detector-level samples/s of `bench.py --detector`; it is not part of
the product and nothing in sparsebev_amd imports it."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class Bottleneck(nn.Module):
    def __init__(self, cin, mid, stride):
        super().__init__()
        cout = mid * 4
        self.conv1, self.bn1 = nn.Conv2d(cin, mid, 1, bias=False), nn.BatchNorm2d(mid)
        self.conv2, self.bn2 = nn.Conv2d(mid, mid, 3, stride, 1, bias=False), nn.BatchNorm2d(mid)    # style='pytorch': stride on the 3x3
        self.conv3, self.bn3 = nn.Conv2d(mid, cout, 1, bias=False), nn.BatchNorm2d(cout)
        self.down = None
        if stride != 1 or cin != cout:
            self.down = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.down is None else self.down(x)
        x = F.relu(self.bn1(self.conv1(x)), inplace=True)
        x = F.relu(self.bn2(self.conv2(x)), inplace=True)
        return F.relu(self.bn3(self.conv3(x)) + idt, inplace=True)


class ResNet50FPN(nn.Module):
    def __init__(self, out_channels=256, num_outs=4):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2, 1))
        stages, cin = [], 64
        for mid, n, stride in ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)):
            blocks = [Bottleneck(cin, mid, stride)] + [Bottleneck(mid * 4, mid, 1) for _ in range(n - 1)]
            stages.append(nn.Sequential(*blocks))
            cin = mid * 4
        self.stages = nn.ModuleList(stages)
        chans = [256, 512, 1024, 2048]
        self.lateral = nn.ModuleList(nn.Conv2d(c, out_channels, 1) for c in chans)
        self.fpn = nn.ModuleList(nn.Conv2d(out_channels, out_channels, 3, padding=1) for _ in chans)
        self.num_outs = num_outs
        self.register_buffer('mean', torch.tensor([123.675, 116.280, 103.530]).view(1, 3, 1, 1))
        self.register_buffer('std', torch.tensor([58.395, 57.120, 57.375]).view(1, 3, 1, 1))

    def forward(self, img):
        """img [n, 3, H, W] uint8-range float -> list of num_outs feature maps [n, 256, H/4.., W/4..]."""
        x = ((img - self.mean) / self.std).contiguous(memory_format=torch.channels_last)
        x = self.stem(x)
        outs = []
        for st in self.stages:
            x = st(x)
            outs.append(x)
        lat = [l(o) for l, o in zip(self.lateral, outs)]
        for i in range(len(lat) - 1, 0, -1):
            lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[-2:], mode='nearest')
        feats = [f(l) for f, l in zip(self.fpn, lat)]
        for _ in range(self.num_outs - len(feats)):                      # extra levels by stride-2 subsampling (mmdet FPN default)
            feats.append(F.max_pool2d(feats[-1], 1, stride=2))
        return feats[:self.num_outs]


def build(device, num_outs=4):
    torch.manual_seed(0)
    m = ResNet50FPN(num_outs=num_outs).to(device).eval().to(memory_format=torch.channels_last)
    return m
