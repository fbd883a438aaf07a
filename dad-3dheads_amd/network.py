"""DAD-3DNet regressor declared for PyTorch-ROCm: the step directly in front of the decode hot path.

Architecture as the reference builds it (structure only; `pytorchcv`, `torchvision` and the trained TorchScript
checkpoint are absent here, so the ResNet-50 is hand-declared and weights are random-initialised -- throughput and
plumbing, not accuracy):

    encoder   ResNet-50 stem + 4 stages                       model_training/model/encoders.py:42-48 (StagedEncoder)
    bifpn     3 lateral 1x1 convs, P6/P7, two BiFPN blocks    model_training/model/bifpn.py:134-163, 76-131
    heatmap   3x3 conv, 68 channels, on the finest level      model_training/model/flame_regression.py:14-25
    fusion    1x1 conv over [stage3, sigmoid(heatmap), P5]    flame_regression.py:28-43, then ResNet stage 4 (:91-93)
    heads     avg-pool -> 512 -> {403 tanh*3, 10, 136 relu}   flame_regression.py:46-59, 94-99

Module NAMES differ from the reference's `FlameRegression` (`fusion` / `heatmap` / `bifpn.lateral` here,
`fusion_layer.conv1x1` / `head.heatmap` / `bifpn.p3..p5` there): `convert_reference_state_dict` renames a reference
(Lightning) state dict, `DAD3DNet.load_reference_state_dict` loads it strictly. tests/test_network_reference.py runs the
reference's own FlameRegression / BiFPN / heads live, moves its weights across and compares the outputs. The trained
model is otherwise used through its TorchScript file (`FaceMeshPredictor(config)` -> `torch.jit.load`, predictor.py:72),
exactly like the reference's predictor does.

`forward` returns the reference's output dict (`OUTPUT_LANDMARKS_HEATMAP`, `OUTPUT_3DMM_PARAMS` [B,413],
`OUTPUT_2D_LANDMARKS` [B,68,2]: flame_regression.py:100-104).
Inference runs channels-last with bf16 WEIGHTS (converted once, no autocast; MIOpen / hipBLASLt underneath); the 413
parameters are cast to fp32 on the way
out because the decode computes in fp32. PyTorch is plumbing here; the elementwise passes it would leave between the
convolutions (bias, residual, ReLU; the BiFPN's weighted resize-and-sum) go through csrc/cnn_glue.hip when serving.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from . import _glue

# keys of the network's output dict: the VALUES model_training/data/config.py:16-23 gives these names (each constant's
# value is its own name) -- what the reference's TorchScript checkpoint returns and predictor.py:103-109 looks up
OUTPUT_3DMM_PARAMS = "OUTPUT_3DMM_PARAMS"
OUTPUT_2D_LANDMARKS = "OUTPUT_2D_LANDMARKS"
OUTPUT_LANDMARKS_HEATMAP = "OUTPUT_LANDMARKS_HEATMAP"


def _conv_bn(cin: int, cout: int, k: int, stride: int = 1, relu: bool = True) -> nn.Sequential:
    layers: List[nn.Module] = [nn.Conv2d(cin, cout, k, stride, k // 2, bias=False), nn.BatchNorm2d(cout)]
    if relu:
        layers.append(nn.ReLU(inplace=True))
    return nn.Sequential(*layers)


class Bottleneck(nn.Module):
    """ResNet bottleneck: 1x1 -> 3x3 -> 1x1 x4, projection shortcut when the shape changes. The reference's backbone is
    pytorchcv's `resnet50` (encoders.py:5,52; config/model/resnet_regression.yaml:3), which is built with
    `conv1_stride=True`: the stride sits on the FIRST 1x1 convolution (the original v1 placement), not on the 3x3 of
    torchvision's v1.5 -- that is what is declared here (`stride_on_first=False` gives the v1.5 variant, `resnet50b`)."""

    def __init__(self, cin: int, width: int, stride: int, stride_on_first: bool = True):
        super().__init__()
        cout = 4 * width
        s1, s3 = (stride, 1) if stride_on_first else (1, stride)
        self.body = nn.Sequential(_conv_bn(cin, width, 1, s1), _conv_bn(width, width, 3, s3), _conv_bn(width, cout, 1, relu=False))
        self.shortcut = None if (stride == 1 and cin == cout) else _conv_bn(cin, cout, 1, stride, relu=False)

    def forward(self, x: Tensor) -> Tensor:
        skip = x if self.shortcut is None else self.shortcut(x)
        last = self.body[2][0]
        if isinstance(last, ConvBiasAct):  # serving form: bias + identity + ReLU in one pass behind the last convolution
            return last(self.body[1](self.body[0](x)), z=skip, relu=True)
        return F.relu(self.body(x) + skip, inplace=True)


def _stage(cin: int, width: int, blocks: int, stride: int) -> nn.Sequential:
    return nn.Sequential(*[Bottleneck(cin if i == 0 else 4 * width, width, stride if i == 0 else 1) for i in range(blocks)])


class ResNet50Stages(nn.Module):
    """`stages[0..4]` = stem, 256@1/4, 512@1/8, 1024@1/16, 2048@1/32 -- the split the reference's forward walks."""

    channels = {"layer0": 2048, "layer1": 1024, "layer2": 512, "layer3": 256, "layer4": 64}  # model/backbone.yaml

    def __init__(self):
        super().__init__()
        stem = nn.Sequential(_conv_bn(3, 64, 7, 2), nn.MaxPool2d(3, 2, 1))
        self.stages = nn.ModuleList([stem, _stage(64, 64, 3, 1), _stage(256, 128, 4, 2), _stage(512, 256, 6, 2), _stage(1024, 512, 3, 2)])


class SeparableBlock(nn.Module):
    """depthwise 1x1 -> pointwise 1x1 -> BN(momentum .9997, eps 4e-5) -> ReLU (bifpn.py:11-43; kernel size 1 as there)."""

    def __init__(self, ch: int):
        super().__init__()
        self.depthwise = nn.Conv2d(ch, ch, 1, groups=ch, bias=False)
        self.pointwise = nn.Conv2d(ch, ch, 1, bias=False)
        self.bn = nn.BatchNorm2d(ch, momentum=0.9997, eps=4e-5)

    def forward(self, x: Tensor) -> Tensor:
        if isinstance(self.pointwise, ConvBiasAct):  # serving form: depthwise scale and BatchNorm folded in, ReLU fused behind
            return self.pointwise(x)
        return F.relu(self.bn(self.pointwise(self.depthwise(x))))


def _resize_to(x: Tensor, ref: Tensor) -> Tensor:
    return F.interpolate(x, size=ref.shape[2:])  # nearest, like bifpn.py:105-125


class BiFPNBlock(nn.Module):
    """One top-down + bottom-up pass over five levels with ReLU-normalised fusion weights (+ eps AFTER the division,
    as the reference writes it, bifpn.py:98-101)."""

    def __init__(self, ch: int, eps: float = 1e-4):
        super().__init__()
        self.eps = eps
        self.td = nn.ModuleList([SeparableBlock(ch) for _ in range(4)])   # P6, P5, P4, P3 (top-down order)
        self.out = nn.ModuleList([SeparableBlock(ch) for _ in range(4)])  # P4, P5, P6, P7 (bottom-up order)
        self.w_td = nn.Parameter(torch.ones(2, 4))
        self.w_out = nn.Parameter(torch.ones(3, 4))
        self.frozen = None  # (a, b) as nested lists of floats once fold_batchnorm() has frozen the block

    def fusion_weights(self):
        a = F.relu(self.w_td)
        a = a / a.sum(0) + self.eps
        b = F.relu(self.w_out)
        b = b / b.sum(0) + self.eps
        return a, b

    def forward(self, levels: Sequence[Tensor]) -> List[Tensor]:
        p3, p4, p5, p6, p7 = levels
        if self.frozen is not None and self.__dict__.get("glue") and _glue.supported(p3, p4, p5, p6, p7):
            # serving form on the GPU: a node's weighted sum WITH its nearest-neighbour resize is one streaming kernel
            # (csrc/cnn_glue.hip) instead of a scale, a materialised F.interpolate and one or two adds
            a, b = self.frozen
            t6 = self.td[0](_glue.resize_sum((a[0][0], a[1][0]), (p6, p7), p6.shape[2:]))
            t5 = self.td[1](_glue.resize_sum((a[0][1], a[1][1]), (p5, t6), p5.shape[2:]))
            t4 = self.td[2](_glue.resize_sum((a[0][2], a[1][2]), (p4, t5), p4.shape[2:]))
            o3 = self.td[3](_glue.resize_sum((a[0][3], a[1][3]), (p3, t4), p3.shape[2:]))
            o4 = self.out[0](_glue.resize_sum((b[0][0], b[1][0], b[2][0]), (p4, t4, o3), p4.shape[2:]))
            o5 = self.out[1](_glue.resize_sum((b[0][1], b[1][1], b[2][1]), (p5, t5, o4), p5.shape[2:]))
            o6 = self.out[2](_glue.resize_sum((b[0][2], b[1][2], b[2][2]), (p6, t6, o5), p6.shape[2:]))
            o7 = self.out[3](_glue.resize_sum((b[0][3], b[1][3], b[2][3]), (p7, p7, o6), p7.shape[2:]))
            return [o3, o4, o5, o6, o7]
        if self.frozen is not None:  # inference: the weights are python floats, a weighted sum is two passes, not three
            a, b = self.frozen
            fuse2 = lambda w0, x0, w1, x1: torch.add(w0 * x0, x1, alpha=w1)  # noqa: E731
            fuse3 = lambda w0, x0, w1, x1, w2, x2: torch.add(torch.add(w0 * x0, x1, alpha=w1), x2, alpha=w2)  # noqa: E731
        else:
            ta, tb = self.fusion_weights()
            a, b = [[ta[i, j] for j in range(4)] for i in range(2)], [[tb[i, j] for j in range(4)] for i in range(3)]
            fuse2 = lambda w0, x0, w1, x1: w0 * x0 + w1 * x1  # noqa: E731
            fuse3 = lambda w0, x0, w1, x1, w2, x2: w0 * x0 + w1 * x1 + w2 * x2  # noqa: E731
        t6 = self.td[0](fuse2(a[0][0], p6, a[1][0], _resize_to(p7, p6)))
        t5 = self.td[1](fuse2(a[0][1], p5, a[1][1], _resize_to(t6, p5)))
        t4 = self.td[2](fuse2(a[0][2], p4, a[1][2], _resize_to(t5, p4)))
        o3 = self.td[3](fuse2(a[0][3], p3, a[1][3], _resize_to(t4, p3)))
        o4 = self.out[0](fuse3(b[0][0], p4, b[1][0], t4, b[2][0], _resize_to(o3, p4)))
        o5 = self.out[1](fuse3(b[0][1], p5, b[1][1], t5, b[2][1], _resize_to(o4, p5)))
        o6 = self.out[2](fuse3(b[0][2], p6, b[1][2], t6, b[2][2], _resize_to(o5, p6)))
        o7 = self.out[3](fuse3(b[0][3], p7, b[1][3], p7, b[2][3], _resize_to(o6, p7)))
        return [o3, o4, o5, o6, o7]


class BiFPN(nn.Module):
    def __init__(self, in_channels: Sequence[int], ch: int, num_layers: int = 2):
        super().__init__()
        c2, c3, c4 = in_channels
        self.lateral = nn.ModuleList([nn.Conv2d(c2, ch, 1), nn.Conv2d(c3, ch, 1), nn.Conv2d(c4, ch, 1)])
        self.p6 = nn.Conv2d(c4, ch, 3, 2, 1)
        self.p7 = nn.Sequential(nn.Conv2d(ch, ch, 3, 2, 1), nn.BatchNorm2d(ch, momentum=0.9997, eps=4e-5), nn.ReLU())
        self.blocks = nn.ModuleList([BiFPNBlock(ch) for _ in range(num_layers)])

    def forward(self, feats: Sequence[Tensor]) -> List[Tensor]:
        c2, c3, c4 = feats
        p6 = self.p6(c4)
        levels = [self.lateral[0](c2), self.lateral[1](c3), self.lateral[2](c4), p6, self.p7(p6)]
        for blk in self.blocks:
            levels = blk(levels)
        return levels


class RegressionHead(nn.Module):
    def __init__(self, cin: int, cout: int, hidden: int = 512, dropout: float = 0.3):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(cin, hidden), nn.ReLU(inplace=True), nn.Dropout(dropout), nn.Linear(hidden, cout))

    def forward(self, fmap: Tensor) -> Tensor:
        return self.mlp(fmap.mean(dim=(2, 3)))


class DAD3DNet(nn.Module):
    """FlameRegression (flame_regression.py:62-105) with the resnet50 config of config/model/resnet_regression.yaml."""

    def __init__(self, num_landmarks: int = 68, num_filters: int = 256, limit_value: float = 3.0, seed: int = 0):
        super().__init__()
        gen_state = torch.random.get_rng_state()
        torch.manual_seed(seed)
        try:
            self.encoder = ResNet50Stages()
            ch = ResNet50Stages.channels
            self.bifpn = BiFPN([ch["layer3"], ch["layer2"], ch["layer1"]], num_filters)
            self.heatmap = nn.Conv2d(num_filters, num_landmarks, 3, padding=1)
            nn.init.zeros_(self.heatmap.bias)
            self.fusion = nn.Conv2d(num_filters + num_landmarks + ch["layer1"], ch["layer1"], 1)
            self.shape = RegressionHead(ch["layer0"], 403)   # 300 shape + 100 expression + 3 jaw
            self.pose = RegressionHead(ch["layer0"], 10)     # 6-DoF rotation, translation, scale
            self.landmarks = RegressionHead(ch["layer0"], 2 * num_landmarks)
        finally:
            torch.random.set_rng_state(gen_state)
        self.limit_value = limit_value

    def load_reference_state_dict(self, state: Dict[str, Tensor], strict: bool = True):
        """Load the weights of a reference `FlameRegression` (its `state_dict()` or the `state_dict` entry of a
        Lightning checkpoint) into this declaration; to be called BEFORE `InferenceNet` folds the BatchNorms."""
        return self.load_state_dict(convert_reference_state_dict(state), strict=strict)

    def forward(self, x: Tensor) -> Dict[str, Tensor]:
        stages = self.encoder.stages
        feats = []
        for st in stages[:4]:
            x = st(x)
            feats.append(x)
        levels = self.bifpn(feats[1:])
        heatmap = self.heatmap(levels[0])
        hm = F.interpolate(heatmap, size=x.shape[2:], mode="bilinear", align_corners=True).sigmoid()
        fused = self.fusion(torch.cat([x, hm, levels[2]], dim=1)) * x
        top = stages[4](fused)
        shape = torch.tanh(self.shape(top)) * self.limit_value
        lmk = F.relu(self.landmarks(top)).reshape(x.shape[0], -1, 2)
        return {OUTPUT_LANDMARKS_HEATMAP: heatmap, OUTPUT_3DMM_PARAMS: torch.cat([shape, self.pose(top)], dim=1),
                OUTPUT_2D_LANDMARKS: lmk}


_TD_SLOT = {"p6_td": 0, "p5_td": 1, "p4_td": 2, "p3_td": 3}      # BiFPNBlock.td, top-down order (bifpn.py:84-87)
_OUT_SLOT = {"p4_out": 0, "p5_out": 1, "p6_out": 2, "p7_out": 3}  # BiFPNBlock.out, bottom-up order (bifpn.py:89-92)


def convert_reference_state_dict(state: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """Key names of the reference's `FlameRegression.state_dict()` -> the names of `DAD3DNet`. Accepts the optional
    `model.` prefix a Lightning checkpoint puts in front (model_training/model/utils.py:20-27 strips it the same way).

        encoder.model.init_block.conv.{conv,bn}                 -> encoder.stages.0.0.{0,1}      (pytorchcv's names:
        encoder.model.stage{s}.unit{u}.body.conv{c}.{conv,bn}   -> encoder.stages.{s}.{u-1}.body.{c-1}.{0,1}   encoders.py:22,
        encoder.model.stage{s}.unit{u}.identity_conv.{conv,bn}  -> encoder.stages.{s}.{u-1}.shortcut.{0,1}     45-47)
        bifpn.p{3,4,5}  -> bifpn.lateral.{0,1,2}     bifpn.p7.{conv,bn} -> bifpn.p7.{0,1}        (bifpn.py:137-146)
        bifpn.bifpn.{i}.{p6_td..p3_td, p4_out..p7_out, w1, w2} -> bifpn.blocks.{i}.{td.k, out.k, w_td, w_out}
        head.heatmap -> heatmap    fusion_layer.conv1x1 -> fusion    {shape,pose,landmarks}.logit_image -> .mlp
    """
    import re

    out: Dict[str, Tensor] = {}
    for key, value in state.items():
        k = key[len("model."):] if key.startswith("model.") else key
        if k.startswith("encoder.model."):
            k = k[len("encoder.model."):]
            k = re.sub(r"^init_block\.conv\.conv\.", "encoder.stages.0.0.0.", k)
            k = re.sub(r"^init_block\.conv\.bn\.", "encoder.stages.0.0.1.", k)
            m = re.match(r"^stage(\d+)\.unit(\d+)\.(body\.conv(\d)|identity_conv)\.(conv|bn)\.(.+)$", k)
            if m:
                where = f"body.{int(m.group(4)) - 1}" if m.group(4) else "shortcut"
                k = f"encoder.stages.{m.group(1)}.{int(m.group(2)) - 1}.{where}.{0 if m.group(5) == 'conv' else 1}.{m.group(6)}"
            elif not k.startswith("encoder.stages."):
                raise KeyError(f"unexpected encoder entry in the reference state dict: {key}")
        else:
            k = re.sub(r"^bifpn\.p([345])\.", lambda m: f"bifpn.lateral.{int(m.group(1)) - 3}.", k)
            k = re.sub(r"^bifpn\.p7\.conv\.", "bifpn.p7.0.", k)
            k = re.sub(r"^bifpn\.p7\.bn\.", "bifpn.p7.1.", k)
            m = re.match(r"^bifpn\.bifpn\.(\d+)\.(\w+?)(\..+)?$", k)
            if m:
                name, rest = m.group(2), m.group(3) or ""
                if name in _TD_SLOT:
                    name = f"td.{_TD_SLOT[name]}"
                elif name in _OUT_SLOT:
                    name = f"out.{_OUT_SLOT[name]}"
                else:
                    name = {"w1": "w_td", "w2": "w_out"}[name]
                k = f"bifpn.blocks.{m.group(1)}.{name}{rest}"
            k = re.sub(r"^head\.heatmap\.", "heatmap.", k)
            k = re.sub(r"^fusion_layer\.conv1x1\.", "fusion.", k)
            k = re.sub(r"^(shape|pose|landmarks)\.logit_image\.", r"\1.mlp.", k)
        out[k] = value
    return out


@torch.no_grad()
def _fold(conv: nn.Conv2d, bn: nn.BatchNorm2d) -> nn.Conv2d:
    """conv followed by an eval-mode BatchNorm == one conv: w' = w * g / sqrt(var + eps), b' = (b - mean) * g / sqrt(..) + beta."""
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    out = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, conv.dilation,
                    conv.groups, bias=True).to(conv.weight.device, conv.weight.dtype)
    out.weight.copy_(conv.weight * scale.view(-1, 1, 1, 1))
    bias = conv.bias if conv.bias is not None else torch.zeros_like(bn.running_mean)
    out.bias.copy_((bias - bn.running_mean) * scale + bn.bias)
    return out


def fold_batchnorm(module: nn.Module) -> nn.Module:
    """Inference-only rewrite, in place: every Conv2d directly followed by a BatchNorm2d (the ResNet blocks, P7, the
    BiFPN separable blocks) becomes one convolution -- 60-odd memory-bound normalisation passes over the activations
    disappear. Outputs change by fp32 rounding only."""
    for child in module.children():
        fold_batchnorm(child)
    if isinstance(module, nn.Sequential):
        mods = list(module.children())
        i = 0
        while i + 1 < len(mods):
            if isinstance(mods[i], nn.Conv2d) and isinstance(mods[i + 1], nn.BatchNorm2d):
                module[i] = _fold(mods[i], mods[i + 1])
                module[i + 1] = nn.Identity()
                i += 2
            else:
                i += 1
    elif isinstance(module, SeparableBlock) and isinstance(module.bn, nn.BatchNorm2d):
        module.pointwise = _fold(module.pointwise, module.bn)
        module.bn = nn.Identity()
        if isinstance(module.depthwise, nn.Conv2d) and module.depthwise.kernel_size == (1, 1) and module.depthwise.bias is None:
            # a 1x1 depthwise convolution is a per-channel scale of the pointwise convolution's input: fold it in
            with torch.no_grad():
                module.pointwise.weight.mul_(module.depthwise.weight.view(1, -1, 1, 1))
            module.depthwise = nn.Identity()
    elif isinstance(module, BiFPNBlock):
        with torch.no_grad():
            a, b = module.fusion_weights()
        module.frozen = (a.tolist(), b.tolist())
    return module


class ConvBiasAct(nn.Module):
    """Serving form of `conv -> (+ bias) -> (+ residual) -> ReLU`: the convolution runs without its bias, then ONE in-place
    streaming pass (csrc/cnn_glue.hip `dad3d_nhwc_bias_act`) adds the bias -- the folded BatchNorm's shift -- and the identity
    and clamps. The framework's own sequence was a broadcast add, a second add and a clamp: three memory-bound launches per
    bottleneck, two behind every other convolution. CPU tensors and shapes the kernel does not take use the plain ops."""

    def __init__(self, conv: nn.Conv2d, relu: bool):
        super().__init__()
        self.bias = nn.Parameter(conv.bias.detach().clone(), requires_grad=False)
        conv.bias = None
        self.conv = conv
        self.relu = relu

    def forward(self, x: Tensor, z: Tensor = None, relu: bool = None) -> Tensor:
        relu = self.relu if relu is None else relu
        y = self.conv(x)
        if _glue.supported(y, z) and self.bias.dtype == y.dtype:
            return _glue.bias_act_(y, self.bias, z, relu)
        y = y + self.bias.view(1, -1, 1, 1)
        if z is not None:
            y = y + z
        return F.relu(y) if relu else y


def fuse_glue(module: nn.Module) -> nn.Module:
    """Inference-only rewrite, in place, AFTER fold_batchnorm: every biased Conv2d of a Sequential becomes a ConvBiasAct that
    also takes the ReLU behind it; folded separable blocks and the BiFPN's fusion nodes switch to their one-pass forms."""
    for child in module.children():
        fuse_glue(child)
    if isinstance(module, nn.Sequential):
        mods = list(module.children())
        for i, m in enumerate(mods):
            if isinstance(m, nn.Conv2d) and m.bias is not None:
                j = i + 1
                while j < len(mods) and isinstance(mods[j], nn.Identity):
                    j += 1
                relu = j < len(mods) and isinstance(mods[j], nn.ReLU)
                module[i] = ConvBiasAct(m, relu)
                if relu:
                    module[j] = nn.Identity()
                    mods[j] = module[j]
    if isinstance(module, Bottleneck) and module.shortcut is not None and isinstance(module.body[2][0], ConvBiasAct) \
            and isinstance(module.shortcut[0], ConvBiasAct):
        # the projection shortcut's shift joins the last convolution's: the shortcut needs no pass of its own
        with torch.no_grad():
            module.body[2][0].bias.add_(module.shortcut[0].bias)
        module.shortcut[0] = module.shortcut[0].conv  # bias already None
    if isinstance(module, SeparableBlock) and isinstance(module.depthwise, nn.Identity) and isinstance(module.bn, nn.Identity) \
            and isinstance(module.pointwise, nn.Conv2d) and module.pointwise.bias is not None:
        module.pointwise = ConvBiasAct(module.pointwise, True)
    if isinstance(module, BiFPNBlock) and module.frozen is not None:
        module.__dict__["glue"] = True
    return module


class InferenceNet(nn.Module):
    """`DAD3DNet` frozen for serving: eval mode, BatchNorm folded into the convolutions, channels-last, reduced-precision
    autocast; fp32 parameters out. `tune=True` lets MIOpen time its kernels per layer shape on the first call (+36 % at
    batch 64, first call ~17 s)."""

    def __init__(self, net: nn.Module, dtype: torch.dtype = torch.bfloat16, fold_bn: bool = True, tune: bool = False,
                 glue: bool = True):
        super().__init__()
        net = net.eval()
        if fold_bn:
            fold_batchnorm(net)
            if glue:
                fuse_glue(net)
        # Reduced precision by converting the WEIGHTS once, not by autocast: under autocast every forward re-cast every fp32
        # weight (175 copy kernels per batch) and the BiFPN's resize / weighted sums ran in fp32 between casts
        # (profiles/r03_kernel_log.md section 5: 30 % of the forward's kernel time). Measured and NOT done: convolution + bias + ReLU and
        # convolution + residual + ReLU through torch.miopen_convolution_relu / _add_relu -- for channels-last bf16 MIOpen's
        # fusion plans fall back to its naive reference convolution (860 ms per batch of 64 instead of 7).
        self.net = net.to(dtype).to(memory_format=torch.channels_last) if dtype != torch.float32 else net.to(memory_format=torch.channels_last)
        self.dtype = dtype
        if tune:
            torch.backends.cudnn.benchmark = True  # MIOpen find mode on ROCm

    @torch.no_grad()
    def forward(self, x: Tensor) -> Dict[str, Tensor]:
        x = x.to(self.dtype).contiguous(memory_format=torch.channels_last)
        out = self.net(x)
        return {k: v.float() for k, v in out.items()}


class GraphedNet(nn.Module):
    """Replays a frozen network from a hipGraph, one graph per input shape: at batch 1 the ~200 kernels of DAD-3DNet are
    launch-bound (5.5 ms eager), the captured graph removes the per-kernel launch cost. Inputs are copied into the
    graph's static buffer, outputs are cloned out of it."""

    def __init__(self, net: nn.Module, warmup: int = 3):
        super().__init__()
        self.net = net
        self.warmup = warmup
        self._graphs: Dict[tuple, tuple] = {}

    @torch.no_grad()
    def forward(self, x: Tensor) -> Dict[str, Tensor]:
        key = (tuple(x.shape), x.dtype, x.device)
        entry = self._graphs.get(key)
        if entry is None:
            static_x = x.clone()
            side = torch.cuda.Stream(x.device)
            side.wait_stream(torch.cuda.current_stream(x.device))
            with torch.cuda.stream(side):
                for _ in range(self.warmup):  # MIOpen picks its kernels and allocates workspaces outside the capture
                    self.net(static_x)
            torch.cuda.current_stream(x.device).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = self.net(static_x)
            entry = (graph, static_x, static_out)
            self._graphs[key] = entry
        graph, static_x, static_out = entry
        static_x.copy_(x)
        graph.replay()
        return {k: v.clone() for k, v in static_out.items()}
