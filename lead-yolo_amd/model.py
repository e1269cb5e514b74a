"""yaml -> Model plugin surface of the detector (the reference's models/yolo.py:173-362, 397-492),
rebuilt around the HIP modules.

Same yaml schema ([from, repeats, "ModuleName", args]), same width/depth scaling rules
(`make_divisible(c2*gw, 8)`, `max(round(n*gd), 1)`), same routing (`m.f`, `save`), same attributes
(`.model .save .stride .names .yaml .inplace`), `fuse()`, `_apply()`.  Differences, by design:
  * module names resolve through an explicit registry instead of eval() on module globals;
  * strides come from shape inference rather than a 256x256 CPU probe forward (there is no CPU
    compute path in this package);
  * with `fuse_graph=True` (default) Upsample/Concat hand a lazy view to the consumer GEMM.
"""
import math
from copy import deepcopy
from pathlib import Path

import torch
import torch.nn as nn

from . import grad
from . import modules as M

REGISTRY = {
    "PatchEmbed_FasterNet": M.PatchEmbed_FasterNet, "PatchMerging_FasterNet": M.PatchMerging_FasterNet,
    "BasicStage": M.BasicStage, "RFCBAMConv": M.RFCBAMConv, "C3_CA": M.C3_CA, "Conv": M.Conv, "SPPF": M.SPPF,
    "Concat": M.Concat, "nn.Upsample": M.Upsample, "Detect": M.Detect,
}
_CHANNEL_KINDS = {M.Conv, M.SPPF, M.C3_CA, M.RFCBAMConv, M.BasicStage, M.PatchEmbed_FasterNet, M.PatchMerging_FasterNet}
DEFAULT_CFG = str(Path(__file__).resolve().parent / "cfg" / "LEAD-YOLO.yaml")
# (depth_multiple, width_multiple) by analogy with models/yolov5{n,s,l}.yaml.  The m / x widths (BasicStage dims 32/64/120/240 and
# 56/104/200/400) are not among the widths the fused MLPBlock kernel is built for, so those scales are not offered.
SCALES = {"n": (0.33, 0.25), "s": (0.33, 0.50), "l": (1.0, 1.0)}


def make_divisible(x, divisor):
    """reference utils/general.py:669-673"""
    if isinstance(divisor, torch.Tensor):
        divisor = int(divisor.max())
    return math.ceil(x / divisor) * divisor


def load_cfg(cfg=DEFAULT_CFG, scale=None):
    if isinstance(cfg, dict):
        d = deepcopy(cfg)
    else:
        import yaml
        with open(cfg, encoding="ascii", errors="ignore") as f:
            d = yaml.safe_load(f)
    if scale is not None:
        if scale not in SCALES:
            raise NotImplementedError(f"scale {scale!r} is not built (available: {sorted(SCALES)}): its FasterNet stage widths have no "
                                      "fused MLPBlock kernel (csrc/ly_mlpblock.hpp is instantiated for C in {16, 24, 40, 80, 160, 320})")
        d["depth_multiple"], d["width_multiple"] = SCALES[scale]
    return d


def _resolve_arg(a, nc, anchors):
    if not isinstance(a, str):
        return a
    return {"nc": nc, "anchors": anchors, "None": None, "True": True, "False": False}.get(a, a)


def parse_model(d, ch):
    """reference models/yolo.py:397-492.  Returns (nn.Sequential, save, stride_factors)."""
    anchors, nc, gd, gw = d["anchors"], d["nc"], d["depth_multiple"], d["width_multiple"]
    act = d.get("activation")
    if act:
        acts = {"nn.SiLU()": nn.SiLU, "nn.ReLU()": nn.ReLU, "nn.Identity()": nn.Identity}
        if act not in acts:
            raise NotImplementedError(f"activation override {act!r} is not supported by the HIP epilogues")
        M.Conv.default_act = acts[act]()
    na = (len(anchors[0]) // 2) if isinstance(anchors, list) else anchors
    no = na * (nc + 5)
    layers, save, c2 = [], [], ch[-1]
    down = []                                   # cumulative downsampling factor of each layer's output
    for i, (f, n, name, args) in enumerate(d["backbone"] + d["head"]):
        if name not in REGISTRY:
            raise NotImplementedError(f"module {name!r} (layer {i}) is outside the LEAD-YOLO hot path built here; "
                                      f"available: {sorted(REGISTRY)}")
        m = REGISTRY[name]
        args = [_resolve_arg(a, nc, anchors) for a in args]
        n = n_ = max(round(n * gd), 1) if n > 1 else n
        d_in = (down[f] if isinstance(f, int) else down[f[0]]) if i > 0 else 1.0
        d_out = d_in
        if m in _CHANNEL_KINDS:
            c1, c2 = ch[f], args[0]
            if c2 != no:
                c2 = make_divisible(c2 * gw, 8)
            args = [c1, c2, *args[1:]]
            if m is M.C3_CA:
                args.insert(2, n)
                n = 1
            elif m is M.BasicStage:
                args.pop(1)
            if m in (M.PatchEmbed_FasterNet, M.PatchMerging_FasterNet):
                d_out = d_in * args[3]
            elif m is M.RFCBAMConv:
                d_out = d_in * (args[3] if len(args) > 3 else 1)
            elif m is M.Conv:
                d_out = d_in * (args[3] if len(args) > 3 else 1)
        elif m is M.Concat:
            c2 = sum(ch[x] for x in f)
        elif m is M.Detect:
            args.append([ch[x] for x in f])
            if isinstance(args[1], int):
                args[1] = [list(range(args[1] * 2))] * len(f)
        elif m is M.Upsample:
            c2 = ch[f]
            d_out = d_in / float(args[1])
        else:
            c2 = ch[f]
        m_ = nn.Sequential(*(m(*args) for _ in range(n))) if n > 1 else m(*args)
        if n > 1:
            d_out = d_in * ((d_out / d_in) ** n)
        t = name
        np_ = sum(x.numel() for x in m_.parameters())
        m_.i, m_.f, m_.type, m_.np = i, f, t, np_
        save.extend(x % i for x in ([f] if isinstance(f, int) else f) if x != -1)
        layers.append(m_)
        down.append(d_out)
        if i == 0:
            ch = []
        ch.append(c2)
    return nn.Sequential(*layers), sorted(save), down


def initialize_weights(model):
    """reference utils/torch_utils.py:212-221: BN eps/momentum policy, in-place activations."""
    for m in model.modules():
        t = type(m)
        if t is nn.BatchNorm2d:
            m.eps = 1e-3
            m.momentum = 0.03
        elif t in (nn.Hardswish, nn.LeakyReLU, nn.ReLU, nn.ReLU6, nn.SiLU):
            m.inplace = True


def check_anchor_order(m):
    """reference utils/autoanchor.py:19-27"""
    a = m.anchors.prod(-1).mean(-1).view(-1)
    da = a[-1] - a[0]
    ds = m.stride[-1] - m.stride[0]
    if da and (da.sign() != ds.sign()):
        m.anchors[:] = m.anchors.flip(0)


def fold_bn(conv, bn):
    """Fold eval-mode BatchNorm into the preceding convolution's weight/bias, in place on `conv`
    (what reference utils/torch_utils.py:248-269 computes)."""
    with torch.no_grad():
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        w = conv.weight * scale.view(-1, 1, 1, 1)
        b0 = conv.bias if conv.bias is not None else torch.zeros_like(bn.running_mean)
        b = (b0 - bn.running_mean) * scale + bn.bias
    conv.weight = nn.Parameter(w, requires_grad=False)
    conv.bias = nn.Parameter(b, requires_grad=False)
    return conv


class DetectionModel(nn.Module):
    def __init__(self, cfg=DEFAULT_CFG, ch=3, nc=None, anchors=None, fuse_graph=True):
        super().__init__()
        if isinstance(cfg, dict):
            self.yaml = deepcopy(cfg)
        else:
            self.yaml_file = Path(cfg).name
            self.yaml = load_cfg(cfg)
        ch = self.yaml["ch"] = self.yaml.get("ch", ch)
        if nc and nc != self.yaml["nc"]:
            self.yaml["nc"] = nc
        if anchors:
            self.yaml["anchors"] = round(anchors)
        self.model, self.save, down = parse_model(deepcopy(self.yaml), ch=[ch])
        self.names = [str(i) for i in range(self.yaml["nc"])]
        self.inplace = self.yaml.get("inplace", True)
        m = self.model[-1]
        if isinstance(m, M.Detect):
            m.inplace = self.inplace
            # reference probes a 256x256 forward (models/yolo.py:289); s / x.shape[-2] == the layer's
            # cumulative downsampling factor, obtained here by shape inference
            m.stride = torch.tensor([float(down[j]) for j in m.f])
            check_anchor_order(m)
            m.anchors /= m.stride.view(-1, 1, 1)
            self.stride = m.stride
            self._initialize_biases()
        initialize_weights(self)
        self.set_fuse_graph(fuse_graph)

    @property
    def u8_input(self):
        """the first layer is the HIP patch embedding on an NCHW image: a uint8 batch can be passed as it is (pixel / 255 is folded
        into the patch gather), which is what train.forward_backward does"""
        first = self.model[0]
        return isinstance(first, M.PatchEmbed_FasterNet) and first.k == 4 and first.cin % 4 != 0

    def set_fuse_graph(self, on):
        for mod in self.modules():
            if isinstance(mod, (M.Upsample, M.Concat)):
                mod.lazy = bool(on)

    def forward(self, x, augment=False, profile=False, visualize=False):
        if augment or profile or visualize:
            raise NotImplementedError("augmented inference / per-layer profiling / feature visualisation are host-side tools "
                                      "outside the hot path")
        return self._forward_once(x)

    def _forward_once(self, x):
        y = []
        det = self.model[-1]
        # Inference under hipGraph capture: a Detect level is launched on the auxiliary stream as soon as its feature map
        # exists, so the small, latency-bound head kernels of P3 / P4 overlap the rest of the neck (the last level runs in
        # Detect.forward).  In eager mode everything stays on one stream (fork/join events would cost more host time than they save).
        early = None
        if isinstance(det, M.Detect) and not self.training and not torch.is_grad_enabled() and isinstance(x, torch.Tensor) and x.is_cuda \
                and M._overlap() \
                and x.shape[2] % 32 == 0 and x.shape[3] % 32 == 0 and det.stride is not None and isinstance(det.f, (list, tuple)):
            hw = [(int(x.shape[2] // s), int(x.shape[3] // s)) for s in det._strides()]
            early = det.begin(x.shape[0], hw, x.device)
        # training on the HIP path: a layer output with exactly two consumers goes to them as the two aliases of grad.fork, so that the
        # sum of their gradients can be a GEMM's store (grad.Fork) instead of autograd's extra pass
        forks = self._two_consumer_layers() if (self.training and torch.is_grad_enabled() and grad.FORK_SUM and isinstance(x, torch.Tensor)
                                                and x.is_cuda) else ()

        def take(j):
            v = y[j]
            return v.pop(0) if isinstance(v, list) else v
        for m in self.model:
            if m.f != -1:
                x = take(m.f) if isinstance(m.f, int) else [(take(m.i - 1) if isinstance(y[m.i - 1], list) else x) if j == -1 else take(j) for j in m.f]
            elif m.i > 0 and isinstance(y[m.i - 1], list):
                x = take(m.i - 1)
            if m is det and early is not None:
                det._early = early
            x = m(x)
            if m.i in forks and isinstance(x, torch.Tensor) and x.dim() == 4 and x.requires_grad and x.dtype in (torch.bfloat16, torch.float32):
                y.append(list(grad.fork(x)))
            else:
                y.append(x if m.i in self.save else None)
            if early is not None and m is not det and m.i in det.f[:-1] and isinstance(x, torch.Tensor):
                det.level(early, det.f.index(m.i), x, side=True)
        return x

    def _two_consumer_layers(self):
        """indices of the layers whose output is read by exactly two later layers (models/yolo.py:179-195 routing: m.f)"""
        plan = getattr(self, "_fork_plan", None)
        if plan is None:
            count = {}
            for m in self.model:
                for j in ([m.f] if isinstance(m.f, int) else list(m.f)):
                    j = m.i - 1 if j == -1 else (j if j >= 0 else m.i + j)
                    count[j] = count.get(j, 0) + 1
            plan = self._fork_plan = frozenset(j for j, c in count.items() if c == 2 and j >= 0)
        return plan

    def _initialize_biases(self, cf=None):
        """reference models/yolo.py:352-359"""
        m = self.model[-1]
        for mi, s in zip(m.m, m.stride):
            b = mi.bias.view(m.na, -1)
            b.data[:, 4] += math.log(8 / (640 / s) ** 2)
            b.data[:, 5:5 + m.nc] += math.log(0.6 / (m.nc - 0.99999)) if cf is None else torch.log(cf / cf.sum())
            mi.bias = torch.nn.Parameter(b.view(-1), requires_grad=True)

    def fuse(self):
        """reference models/yolo.py:213-233: fold BN into Conv / PatchEmbed / PatchMerging."""
        for m in self.model.modules():
            if isinstance(m, M.Conv) and hasattr(m, "bn"):
                fold_bn(m.conv, m.bn)
                delattr(m, "bn")
            if type(m) is M.PatchEmbed_FasterNet and isinstance(getattr(m, "norm", None), nn.BatchNorm2d):
                fold_bn(m.proj, m.norm)
                delattr(m, "norm")
            if type(m) is M.PatchMerging_FasterNet and isinstance(getattr(m, "norm", None), nn.BatchNorm2d):
                fold_bn(m.reduction, m.norm)
                delattr(m, "norm")
        return self

    def _apply(self, fn):
        self = super()._apply(fn)
        m = self.model[-1]
        if isinstance(m, M.Detect):
            m.stride = fn(m.stride)
            m.grid = list(map(fn, m.grid))
            if isinstance(m.anchor_grid, list):
                m.anchor_grid = list(map(fn, m.anchor_grid))
        return self


Model = DetectionModel
