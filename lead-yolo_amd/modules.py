"""Drop-in torch.nn.Module replacements for the LEAD-YOLO hot-path blocks, backed by the gfx950 HIP
library (csrc/ -> libleadyolo_hip.so through capi.py).

Same class names, positional constructor signatures and state_dict keys/shapes as the reference
(SURVEY.md §8b), so weights transfer by key and the classes can be injected into the reference's
`models.yolo` globals (inject.py):
  PatchEmbed_FasterNet / PatchMerging_FasterNet / BasicStage   reference models/common.py:1411-1561
  RFCBAMConv                                                    reference models/rfa.py:77-129
  C3_CA (+ CoordAtt, CA_Bottleneck, Conv)                       reference models/common.py:1583-1637,1890-1910

nn.Conv2d / nn.BatchNorm2d / nn.Linear sub-modules are PARAMETER HOLDERS ONLY (they give the
reference's key names and let `initialize_weights` / `fuse()` treat them as usual); their own
forward is never called.  All arithmetic runs in HIP kernels on NHWC (channels_last) activations.
There is no CPU path: calling forward without the GPU library raises.
"""
import ctypes

import torch
import torch.nn as nn

from . import capi, pack

BN_EPS = 1e-3
BN_MOMENTUM = 0.03


def _require_cuda(x, who):
    if not x.is_cuda:
        raise RuntimeError(f"{who}: the HIP path needs a CUDA/ROCm tensor (got {x.device}); there is no CPU fallback")
    if x.dtype != torch.float32:
        raise NotImplementedError(f"{who}: only float32 activations are built so far (got {x.dtype})")


def nhwc(x):
    """Logical NCHW tensor with channels_last (NHWC) physical layout; zero-copy when already so."""
    return x.contiguous(memory_format=torch.channels_last)


def empty_nhwc(n, c, h, w, like):
    return torch.empty((n, c, h, w), dtype=like.dtype, device=like.device, memory_format=torch.channels_last)


class _Prepared:
    """Cache of kernel-ready (packed / folded) parameters, rebuilt when any source tensor changes."""

    def __init__(self):
        self.key = None
        self.val = None

    def get(self, key, build):
        if key != self.key:
            with torch.no_grad():
                self.val = build()
            self.key = key
        return self.val


# --------------------------------------------------------------------------------------------------
# FasterNet
# --------------------------------------------------------------------------------------------------
class Partial_conv3(nn.Module):
    """Parameter holder for the partial 3x3 convolution (first dim//n_div channels)."""

    def __init__(self, dim, n_div, forward="split_cat"):
        super().__init__()
        self.dim_conv3 = dim // n_div
        self.dim_untouched = dim - self.dim_conv3
        self.partial_conv3 = nn.Conv2d(self.dim_conv3, self.dim_conv3, 3, 1, 1, bias=False)


class MLPBlock(nn.Module):
    """x + W2 . relu(BN(W1 . [pconv3x3(x[:, :C/4]) | x[:, C/4:]]))  as ONE fused HIP kernel."""

    def __init__(self, dim, n_div=4, mlp_ratio=2, drop_path=0.0, layer_scale_init_value=0, act_layer=nn.ReLU,
                 norm_layer=nn.BatchNorm2d, pconv_fw_type="split_cat"):
        super().__init__()
        if n_div != 4 or mlp_ratio != 2 or layer_scale_init_value > 0 or drop_path > 0 \
                or norm_layer is not nn.BatchNorm2d or act_layer is not nn.ReLU:
            raise NotImplementedError("HIP MLPBlock is built for n_div=4, mlp_ratio=2, BatchNorm2d+ReLU, no layer-scale/drop-path "
                                      "(the only configuration LEAD-YOLO instantiates)")
        self.dim = dim
        hidden = int(dim * mlp_ratio)
        self.mlp = nn.Sequential(nn.Conv2d(dim, hidden, 1, bias=False), norm_layer(hidden), act_layer(),
                                 nn.Conv2d(hidden, dim, 1, bias=False))
        self.spatial_mixing = Partial_conv3(dim, n_div, pconv_fw_type)
        self._prep = _Prepared()

    def _packed(self):
        wp_, w1_, bn, w2_ = self.spatial_mixing.partial_conv3.weight, self.mlp[0].weight, self.mlp[1], self.mlp[3].weight
        key = pack.versions(wp_, w1_, w2_, bn.weight, bn.bias, bn.running_mean, bn.running_var) + (bn.eps,)

        def build():
            c = self.dim
            wp = pack.frag_pack(pack.conv_taps_matrix(wp_.detach(), 4))
            w1 = pack.frag_pack(w1_.detach().view(2 * c, c))
            w2 = pack.frag_pack(w2_.detach().view(c, 2 * c))
            sc, sh = pack.bn_scale_shift(bn)
            return wp, w1, w2, sc, sh
        return self._prep.get(key, build)

    def forward(self, x):
        _require_cuda(x, "MLPBlock")
        if self.training:
            raise NotImplementedError("MLPBlock: train-mode (batch-statistics) HIP path is not built yet")
        x = nhwc(x)
        n, c, h, w = x.shape
        wp, w1, w2, sc, sh = self._packed()
        y = empty_nhwc(n, c, h, w, x)
        L = capi.lib()
        capi.check(L.ly_mlpblock_fwd(capi.ptr(x), capi.ptr(y), n, h, w, c, capi.ptr(wp), capi.ptr(w1), capi.ptr(w2),
                                     capi.ptr(sc), capi.ptr(sh), capi.stream_ptr()), "ly_mlpblock_fwd")
        return y


class BasicStage(nn.Module):
    def __init__(self, dim, depth=1, n_div=4, mlp_ratio=2, layer_scale_init_value=0, norm_layer=nn.BatchNorm2d,
                 act_layer=nn.ReLU, pconv_fw_type="split_cat"):
        super().__init__()
        self.blocks = nn.Sequential(*[MLPBlock(dim, n_div, mlp_ratio, 0.0, layer_scale_init_value, act_layer, norm_layer,
                                               pconv_fw_type) for _ in range(depth)])

    def forward(self, x):
        for blk in self.blocks:
            x = blk(x)
        return x
