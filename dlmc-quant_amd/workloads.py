"""Shape generators for the benchmark configurations (BASELINE.json `configs`).

The reference takes ResNet-18/50 from torchvision/timm (model/classification/__init__.py:2,4), which are
not installed here, and RepVGG-A1 from model/classification/repvgg.py:205-207.  These are independent
minimal definitions of the same public architectures - only the layer shapes matter to the fake-quantize
path - with random-init weights (Kaiming normal, as cifarresnet.py:46-48 initialises convs).

`layer_table(model, x)` lists every quantisable layer with its input-activation and weight element
counts; tests pin the totals to SURVEY.md section 8(d):
    ResNet-18 : 21 layers, 2 183 168 input elems/img, 11 678 912 weight elems
    ResNet-50 : 54 layers, 10 664 448 input elems/img, 25 502 912 weight elems (4.089 GMAC/img)
    RepVGG-A1 (deploy): 23 layers, 2 459 904 input elems/img, 12 783 296 weight elems
"""
import torch
from torch import nn


def _conv(cin, cout, k, stride=1, groups=1):
    return nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, groups=groups, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, width, stride):
        super().__init__()
        self.conv1 = _conv(cin, width, 3, stride)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = _conv(width, width, 3)
        self.bn2 = nn.BatchNorm2d(width)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1 or cin != width:
            self.downsample = nn.Sequential(_conv(cin, width, 1, stride), nn.BatchNorm2d(width))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + idt)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, width, stride):
        super().__init__()
        cout = width * 4
        self.conv1 = _conv(cin, width, 1)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = _conv(width, width, 3, stride)   # stride on the 3x3 (torchvision "v1.5")
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = _conv(width, cout, 1)
        self.bn3 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(_conv(cin, cout, 1, stride), nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + idt)


class ResNet(nn.Module):
    def __init__(self, block, depths, num_classes=1000):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        cin, stages = 64, []
        for i, (width, depth) in enumerate(zip((64, 128, 256, 512), depths)):
            blocks = []
            for j in range(depth):
                blocks.append(block(cin, width, 2 if (j == 0 and i > 0) else 1))
                cin = width * block.expansion
            stages.append(nn.Sequential(*blocks))
        self.layer1, self.layer2, self.layer3, self.layer4 = stages
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(cin, num_classes)
        _init(self)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


class RepVGGDeployBlock(nn.Module):
    """A RepVGG block after `switch_to_deploy` (repvgg.py:132-147): one 3x3 conv with bias + ReLU."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.rbr_reparam = nn.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=True)
        self.nonlinearity = nn.ReLU()

    def forward(self, x):
        return self.nonlinearity(self.rbr_reparam(x))


class RepVGGTrainBlock(nn.Module):
    """A RepVGG block in its multi-branch training form (3x3+BN, 1x1+BN, identity BN), with the reference's
    attribute names (repvgg.py:22-60) so `dlmc.utils.reparam` converts it."""

    def __init__(self, cin, cout, stride, groups=1):
        super().__init__()
        self.in_channels, self.groups = cin, groups
        self.rbr_identity = nn.BatchNorm2d(cin) if (cin == cout and stride == 1) else None

        def conv_bn(k, pad):
            seq = nn.Sequential()
            seq.add_module("conv", nn.Conv2d(cin, cout, k, stride=stride, padding=pad, groups=groups, bias=False))
            seq.add_module("bn", nn.BatchNorm2d(cout))
            return seq
        self.rbr_dense = conv_bn(3, 1)
        self.rbr_1x1 = conv_bn(1, 0)
        self.nonlinearity = nn.ReLU()

    def forward(self, x):
        if hasattr(self, "rbr_reparam"):
            return self.nonlinearity(self.rbr_reparam(x))
        idt = 0 if self.rbr_identity is None else self.rbr_identity(x)
        return self.nonlinearity(self.rbr_dense(x) + self.rbr_1x1(x) + idt)


class RepVGGDeploy(nn.Module):
    def __init__(self, num_blocks, widths, num_classes=1000):
        super().__init__()
        self.stage0 = RepVGGDeployBlock(3, widths[0], 2)
        cin, stages = widths[0], []
        for n, w in zip(num_blocks, widths[1:]):
            stages.append(nn.Sequential(*[RepVGGDeployBlock(cin if j == 0 else w, w, 2 if j == 0 else 1)
                                          for j in range(n)]))
            cin = w
        self.stage1, self.stage2, self.stage3, self.stage4 = stages
        self.gap = nn.AdaptiveAvgPool2d(1)
        self.linear = nn.Linear(cin, num_classes)
        _init(self)

    def forward(self, x):
        x = self.stage4(self.stage3(self.stage2(self.stage1(self.stage0(x)))))
        return self.linear(torch.flatten(self.gap(x), 1))


class MobileOneDeployBlock(nn.Module):
    """A re-parameterised MobileOne unit: depthwise 3x3 (+bias, ReLU) then pointwise 1x1 (+bias, ReLU)."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.dw = nn.Conv2d(cin, cin, 3, stride=stride, padding=1, groups=cin, bias=True)
        self.pw = nn.Conv2d(cin, cout, 1, bias=True)
        self.act = nn.ReLU()

    def forward(self, x):
        return self.act(self.pw(self.act(self.dw(x))))


class MobileOneDeploy(nn.Module):
    """MobileOne in inference form (public architecture, Vasu et al. 2022; NOT in the reference - BASELINE
    config 5 only borrows its layer shapes).  S1: width multipliers (1.5, 1.5, 2.0, 2.5), blocks (2, 8, 10, 1)."""

    def __init__(self, widths=(1.5, 1.5, 2.0, 2.5), blocks=(2, 8, 10, 1), num_classes=1000):
        super().__init__()
        cin = min(64, int(64 * widths[0]))
        self.stage0 = nn.Sequential(nn.Conv2d(3, cin, 3, stride=2, padding=1, bias=True), nn.ReLU())
        stages = []
        for base, mult, n in zip((64, 128, 256, 512), widths, blocks):
            cout = int(base * mult)
            stages.append(nn.Sequential(*[MobileOneDeployBlock(cin if j == 0 else cout, cout, 2 if j == 0 else 1)
                                          for j in range(n)]))
            cin = cout
        self.stage1, self.stage2, self.stage3, self.stage4 = stages
        self.gap = nn.AdaptiveAvgPool2d(1)
        self.linear = nn.Linear(cin, num_classes)
        _init(self)

    def forward(self, x):
        x = self.stage4(self.stage3(self.stage2(self.stage1(self.stage0(x)))))
        return self.linear(torch.flatten(self.gap(x), 1))


def _init(model):
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_in", nonlinearity="relu")
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.Linear):
            nn.init.kaiming_normal_(m.weight, mode="fan_in", nonlinearity="relu")
            nn.init.zeros_(m.bias)


def resnet18(num_classes=1000):
    return ResNet(BasicBlock, (2, 2, 2, 2), num_classes)


def resnet50(num_classes=1000):
    return ResNet(Bottleneck, (3, 4, 6, 3), num_classes)


def repvgg_a1_deploy(num_classes=1000):
    """RepVGG-A1: num_blocks [2,4,14,1], width multipliers [1,1,1,2.5] (repvgg.py:205-207)."""
    return RepVGGDeploy((2, 4, 14, 1), (64, 64, 128, 256, 1280), num_classes)


def mobileone_s1_deploy(num_classes=1000):
    return MobileOneDeploy(num_classes=num_classes)


MODELS = {"resnet18": resnet18, "resnet50": resnet50, "repvgg_a1": repvgg_a1_deploy, "mobileone_s1": mobileone_s1_deploy}


def layer_table(model, x):
    """[(name, kind, input_shape, weight_shape, macs)] of every Conv2d / Linear, by a hooked forward."""
    rows, hooks = [], []

    def hook(name):
        def fn(mod, inp, out):
            w = mod.weight
            macs = out.numel() // out.shape[0] * (w.numel() // w.shape[0])
            rows.append((name, type(mod).__name__, tuple(inp[0].shape), tuple(w.shape), macs))
        return fn
    for name, m in model.named_modules():
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            hooks.append(m.register_forward_hook(hook(name)))
    was = model.training
    model.eval()
    with torch.no_grad():
        model(x)
    model.train(was)
    for h in hooks:
        h.remove()
    return rows


def table_totals(rows):
    """(layers, input elems per image, weight elems, MACs per image)."""
    n = rows[0][2][0]
    act = sum(int(torch.tensor(r[2]).prod()) for r in rows) // n
    wt = sum(int(torch.tensor(r[3]).prod()) for r in rows)
    return len(rows), act, wt, sum(r[4] for r in rows)
