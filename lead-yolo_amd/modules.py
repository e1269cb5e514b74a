"""Drop-in torch.nn.Module replacements for the LEAD-YOLO hot-path blocks, backed by the gfx950 HIP
library (csrc/ -> libleadyolo_hip.so through capi.py / ops.py).

Same class names, positional constructor signatures and state_dict keys/shapes as the reference
(SURVEY.md §8b), so weights transfer by key and the classes can be injected into the reference's
`models.yolo` globals (inject.py):
  PatchEmbed_FasterNet / PatchMerging_FasterNet / BasicStage   reference models/common.py:1411-1561
  RFCBAMConv (+ SE)                                             reference models/rfa.py:77-129
  C3_CA (+ CoordAtt, CA_Bottleneck, Conv)                       reference models/common.py:1583-1637,1890-1910
  SPPF, Concat, Detect (graph remainder)                        reference models/common.py:348-366,531-538; models/yolo.py:39-153

nn.Conv2d / nn.BatchNorm2d / nn.Linear sub-modules are PARAMETER HOLDERS ONLY (they give the
reference's key names and let `initialize_weights` / `fuse()` treat them as usual); their own
forward is never called.  All arithmetic of the hot path runs in HIP kernels on NHWC (channels_last)
activations.  There is no CPU path: calling forward without the GPU library raises.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, pack
from .ops import ACT_NONE, ACT_RELU, ACT_SILU

BN_EPS = 1e-3
BN_MOMENTUM = 0.03


class _Prepared:
    """Cache of kernel-ready (packed / folded) parameters, rebuilt when any source tensor changes.  `variant` separates the
    forms one parameter set is prepared in (weight packs with 2 planes for fp32 storage, 1 plane for bf16 storage)."""

    def __init__(self):
        self.slots = {}

    def get(self, key, build, variant=0):
        ent = self.slots.get(variant)
        # under hipGraph capture of a training step the weights change between replays: the packing launches must be part of
        # the captured work every time, whatever the version counters say
        if ent is None or ent[0] != key or (torch.is_grad_enabled() and torch.cuda.is_current_stream_capturing()):
            with torch.no_grad():
                ent = (key, build())
            self.slots[variant] = ent
        return ent[1]

    def __deepcopy__(self, memo):
        return _Prepared()

    def __getstate__(self):
        return {}

    def __setstate__(self, state):
        self.slots = {}


def _edge(fn):
    """forward wrapper applying the dtype policy of ops.edge_in / edge_out around a module's body, and running the body with
    autocast OFF: inside a module every dtype is explicit (storage dtype activations, fp32 tables), torch helper ops on pooled
    vectors must not be re-cast by an enclosing torch.autocast region."""
    def forward(self, x):
        if isinstance(x, torch.Tensor) and not (SHAPE_PROBE and not x.is_cuda):
            x, back = ops.edge_in(x, type(self).__name__, self)
        else:
            back = None
        with torch.autocast("cuda", enabled=False):
            y = fn(self, x)
        return ops.edge_out(y, back)
    forward.__name__ = fn.__name__
    forward.__doc__ = fn.__doc__
    return forward


# Shape-probe mode: the reference's DetectionModel.__init__ derives strides from the SHAPES of a
# 256x256 CPU forward (models/yolo.py:289).  inject.patch() switches this flag on only for the duration
# of that constructor; modules then return correctly-shaped zero tensors for CPU inputs.  No arithmetic
# is emulated and the flag is off everywhere else, so a CPU tensor still raises in normal use.
SHAPE_PROBE = False


def _probe(x, shape):
    if SHAPE_PROBE and isinstance(x, torch.Tensor) and not x.is_cuda:
        return x.new_zeros(shape)
    return None


def _grad_mode(mod):
    """Training step: train-mode module called with autograd recording -> the autograd.Function path (grad.py)."""
    return mod.training and torch.is_grad_enabled()


_SIDE_STREAMS = {}


FORK_BRANCHES = True     # graph.GraphedForward clears this while it captures several sub-batch streams (no nested forks)


def _overlap():
    """fork independent latency-bound branches onto the auxiliary stream?  Only under hipGraph capture."""
    return FORK_BRANCHES and torch.cuda.is_current_stream_capturing()



def _side_stream(device):
    """the auxiliary stream paired with the CURRENT stream (one per stream, so that several capture streams can each fork
    their own independent branches: SE attention, early Detect heads)"""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


def _bn_tensors(bn):
    return (bn.weight, bn.bias, bn.running_mean, bn.running_var)


def _act_code(act):
    if act is None or isinstance(act, nn.Identity):
        return ACT_NONE
    if isinstance(act, nn.SiLU):
        return ACT_SILU
    if isinstance(act, nn.ReLU):
        return ACT_RELU
    raise NotImplementedError(f"activation {type(act).__name__} is not built into the HIP epilogues (SiLU / ReLU / identity)")


class Lazy:
    """A not-yet-materialised activation a consumer GEMM can absorb: channel concat of up to two row
    sources, the first optionally at half resolution (nearest 2x upsample) or gated by CoordAtt."""

    def __init__(self, shape, a0, lda0, k0, a1=None, lda1=0, up=False, gate=None, keep=()):
        self.shape = shape          # logical (n, c, h, w)
        self.a0, self.lda0, self.k0 = a0, lda0, k0
        self.a1, self.lda1 = a1, lda1
        self.up = up
        self.gate = gate            # (a_h, a_w) or None
        self.keep = keep            # tensors kept alive

    @staticmethod
    def of(x):
        if isinstance(x, Lazy):
            return x
        t, ld = ops.rows(x)
        n, c, h, w = t.shape
        return Lazy((n, c, h, w), t, ld, c, keep=(t,))

    def materialize(self):
        n, c, h, w = self.shape
        if self.gate is not None:
            assert self.a1 is None and not self.up
            return ops.coordatt_gate(self.a0, self.lda0, n, h, w, c, self.gate[0], self.gate[1])
        parts = []
        a0 = self.keep[0]
        if self.up:
            a0 = F.interpolate(a0, scale_factor=2, mode="nearest")
        parts.append(a0)
        if self.a1 is not None:
            parts.append(self.keep[1])
        return parts[0] if len(parts) == 1 else torch.cat(parts, 1)


def _run_pointwise(src, wp, n_out, e_scale, e_shift, act, out=None, ldo=None, **extra):
    """GEMM over a Lazy / tensor source -> NHWC tensor [n, n_out, h, w] (or into `out` rows)."""
    L = Lazy.of(src)
    n, c, h, w = L.shape
    if extra.get("stats") is not None:
        res, out_t, ldo = None, None, n_out                     # statistics pass: nothing is stored
    elif out is None:
        res = ops.empty_nhwc(n, n_out, h, w, L.a0)
        out_t, ldo = res, n_out
    else:
        res, out_t = None, out
    kw = dict(M=n * h * w, H=h, W=w, K=c, N=n_out, a0=L.a0, lda0=L.lda0, k0=L.k0, a1=L.a1, lda1=L.lda1, wp=wp, out=out_t,
              ldo=ldo, e_scale=e_scale, e_shift=e_shift, act=act, gather=ops.GATHER_UP2 if L.up else ops.GATHER_ROWS)
    if L.gate is not None:
        kw.update(pro=ops.PRO_GATE, g_h=L.gate[0], g_w=L.gate[1])
    kw.update(extra)
    ops.gemm(**kw)
    return res


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
        self.partial_conv3.weight._ly_tap_major = True


class MLPBlock(nn.Module):
    """x + W2 . relu(BN(W1 . [pconv3x3(x[:, :C/4]) | x[:, C/4:]]))  as ONE fused HIP kernel."""

    def __init__(self, dim, n_div=4, mlp_ratio=2, drop_path=0.0, layer_scale_init_value=0, act_layer=nn.ReLU,
                 norm_layer=nn.BatchNorm2d, pconv_fw_type="split_cat"):
        super().__init__()
        if n_div != 4 or mlp_ratio != 2 or layer_scale_init_value > 0 or drop_path > 0 \
                or norm_layer is not nn.BatchNorm2d or act_layer is not nn.ReLU:
            raise NotImplementedError("HIP MLPBlock is built for n_div=4, mlp_ratio=2, BatchNorm2d+ReLU, no layer-scale/drop-path "
                                      "(the only configuration LEAD-YOLO instantiates)")
        if dim % 16 != 0 and dim not in ops.MLP_WIDTHS:
            raise NotImplementedError(f"HIP MLPBlock: the fused kernel is built for dim in {sorted(ops.MLP_WIDTHS)} (lead-yolo n / s / l), the "
                                      f"composed path (partial 3x3 + two 1x1 contractions) for dim % 16 == 0; got dim={dim}")
        self.dim = dim
        hidden = int(dim * mlp_ratio)
        self.mlp = nn.Sequential(nn.Conv2d(dim, hidden, 1, bias=False), norm_layer(hidden), act_layer(),
                                 nn.Conv2d(hidden, dim, 1, bias=False))
        self.spatial_mixing = Partial_conv3(dim, n_div, pconv_fw_type)
        self._prep = _Prepared()
        self._prep_bn = _Prepared()

    def _weights(self, planes=2):
        wp_, w1_, w2_ = self.spatial_mixing.partial_conv3.weight, self.mlp[0].weight, self.mlp[3].weight
        key = pack.versions(wp_, w1_, w2_)

        def build():
            c = self.dim
            htp = (2 * c // 16 + 1) // 2 * 2
            cq = c // 4
            cqp = (cq + 3) // 4 * 4
            return (pack.packed(pack.src_taps(wp_, cqp), 9 * cqp, planes),
                    pack.packed(pack.src_matrix(w1_, 2 * c, c), c, planes, rows_to=16 * htp),
                    pack.packed(pack.src_matrix(w2_, c, 2 * c), 2 * c, planes))
        return self._prep.get(key, build, planes)

    def _bn_eval(self):
        bn = self.mlp[1]
        key = pack.versions(*_bn_tensors(bn)) + (bn.eps,)
        htp = (2 * self.dim // 16 + 1) // 2 * 2

        def build():
            sc, sh = pack.bn_scale_shift(bn)
            return pack.pad_to(sc, 16 * htp), pack.pad_to(sh, 16 * htp)
        return self._prep_bn.get(key, build)

    def _forward_composed(self, x):
        """Any width with dim % 16 == 0 (models/common.py:1432-1437, 1478-1482 take any dim; the fused kernel is instantiated for the widths of
        lead-yolo n / s / l): the same arithmetic as three HIP contraction units — the partial 3x3 convolution on the first dim/4 channels (read
        in place as a channel slice), 1x1 expand + BatchNorm + ReLU, 1x1 project — with the units' own autograd nodes in training.  The channel
        concatenation and the residual add between them are stock ATen device ops.  bf16 storage needs dim % 32 == 0 (16-byte channel slices)."""
        from . import grad
        x = ops.nhwc(x)
        c, cq = self.dim, self.dim // 4
        if cq % ops.vw_of(x) != 0:
            raise NotImplementedError(f"HIP MLPBlock (composed path): dim/4 = {cq} channels are not a whole number of {ops.vw_of(x)}-channel vectors "
                                      f"in {x.dtype} storage")
        planes = ops.planes_of(x)
        wpc_, w1_, w2_ = self.spatial_mixing.partial_conv3.weight, self.mlp[0].weight, self.mlp[3].weight
        bn = self.mlp[1]

        def build():
            cip = (cq + 31) // 32 * 32
            return (pack.packed(pack.src_taps(wpc_, cip), 9 * cip, planes), pack.packed(pack.src_matrix(w1_, 2 * c, c), c, planes),
                    pack.packed(pack.src_matrix(w2_, c, 2 * c), 2 * c, planes))
        wpc, w1p, w2p = self._prep.get(pack.versions(wpc_, w1_, w2_) + ("composed",), build, planes)
        pc = grad.conv_bn_act(grad.ConvSpec("c3", cq), wpc, x[:, :cq], None, wpc_, None, None)
        z = torch.cat((pc, x[:, cq:]), 1).contiguous(memory_format=torch.channels_last)
        hid = grad.conv_bn_act(grad.ConvSpec("pw", 2 * c, ACT_RELU, bn, self.training), w1p, z, None, w1_, None, bn)
        return x + grad.conv_bn_act(grad.ConvSpec("pw", c), w2p, hid, None, w2_, None, None)

    @_edge
    def forward(self, x):
        pr = _probe(x, x.shape)
        if pr is not None:
            return pr
        if self.dim not in ops.MLP_WIDTHS:
            return self._forward_composed(x)
        if _grad_mode(self):
            from . import grad
            bn = self.mlp[1]
            return grad.MlpBlockFn.apply(self, x, self.spatial_mixing.partial_conv3.weight, self.mlp[0].weight, bn.weight, bn.bias,
                                         self.mlp[3].weight)
        x = ops.nhwc(x)
        n, c, h, w = x.shape
        wp, w1, w2 = self._weights(ops.planes_of(x))
        if self.training:
            htp = (2 * c // 16 + 1) // 2 * 2
            stats = ops.new_stats(16 * htp, x.device)
            ops.mlpblock(x, None, n, h, w, c, wp, w1, w2, None, None, stats=stats)         # statistics pass (no store)
            sc, sh = ops.bn_finalize(self.mlp[1], stats, 16 * htp, n * h * w, n=2 * c, pad_to=16 * htp)
        else:
            sc, sh = self._bn_eval()
        y = ops.empty_nhwc(n, c, h, w, x)
        ops.mlpblock(x, y, n, h, w, c, wp, w1, w2, sc, sh)
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


class _PatchConv(nn.Module):
    """k x k stride-k convolution (no bias) + BatchNorm as one gather-GEMM."""
    _conv_name = "proj"
    stride_factor = None

    def _setup(self, cin, cout, k, s, norm_layer):
        if k != s:
            raise NotImplementedError("HIP patch convolution is built for kernel_size == stride (non-overlapping patches)")
        setattr(self, self._conv_name, nn.Conv2d(cin, cout, kernel_size=k, stride=s, bias=False))
        self.norm = norm_layer(cout) if norm_layer is not None else nn.Identity()
        self.k, self.cin, self.cout = k, cin, cout
        self._prep = _Prepared()
        self._prep_bn = _Prepared()

    def _weights(self, nchw, planes=2):
        conv = getattr(self, self._conv_name)
        key = pack.versions(conv.weight) + (nchw,)

        def build():
            w = conv.weight
            co, ci, kh, kw = w.shape
            if nchw:                                              # columns in the weight's own (c, ky, kx) order
                return pack.packed(pack.src_matrix(w, co, ci * kh * kw), ci * kh * kw, planes)
            return pack.packed(pack.src_taps(w, ci), kh * kw * ci, planes)      # columns (ky, kx, c): the NHWC patch gather's order
        return self._prep.get(key, build, planes)

    def _affine_eval(self):
        conv = getattr(self, self._conv_name)
        bn = getattr(self, "norm", None)
        has_bn = isinstance(bn, nn.BatchNorm2d)
        key = pack.versions(conv.bias, *(_bn_tensors(bn) if has_bn else ())) + (bn.eps if has_bn else 0,)

        def build():
            if has_bn:
                return pack.bn_scale_shift(bn, conv.bias)
            return None, (conv.bias.detach().float().contiguous() if conv.bias is not None else None)
        return self._prep_bn.get(key, build)

    def forward(self, x):
        pr = _probe(x, (x.shape[0], self.cout, x.shape[2] // self.k, x.shape[3] // self.k))
        if pr is not None:
            return pr
        n, c, h, w = x.shape
        nchw = c % 4 != 0                           # an image (3 channels): read as fp32 NCHW whatever the compute dtype is
        if nchw:
            # dtype policy by hand: the image itself stays fp32 (the gather kernel converts), only the OUTPUT takes the policy's dtype
            if not x.is_cuda:
                raise RuntimeError(f"{type(self).__name__}: the HIP path needs a CUDA/ROCm tensor (got {x.device}); there is no CPU fallback")
            if x.dtype == torch.uint8:
                # a uint8 image = pixel values to be scaled by 1/255, what the training loop does before the model (train.py:309
                # `imgs.float() / 255`): the gather kernel does it while loading, so the fp32 copy of the batch never exists
                want = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else torch.float32
            else:
                want = torch.get_autocast_dtype("cuda") if (torch.is_autocast_enabled("cuda") and x.dtype == torch.float32) else x.dtype
            odt = torch.float32 if want == torch.float32 else torch.bfloat16
            back = torch.float16 if want == torch.float16 else None
            if x.dtype != torch.uint8:
                # a bf16 / fp16 image of a no-grad bf16 forward is gathered as it is (LY_GATHER_PATCH_NCHW_BF16 / _F16); everything else as fp32
                half_img = x.dtype in (torch.bfloat16, torch.float16) and odt == torch.bfloat16 and not _grad_mode(self) and self.k == 4 and x.shape[3] % 4 == 0
                if not half_img:
                    x = x.float()
                    ops.require_cuda(x, type(self).__name__, self)
        else:
            x, back = ops.edge_in(x, type(self).__name__, self)
            odt = x.dtype
        with torch.autocast("cuda", enabled=False):
            return ops.edge_out(self._fwd(x, nchw, odt), back)

    def _fwd(self, x, nchw, odt):
        n, c, h, w = x.shape
        k = self.k
        ho, wo = h // k, w // k
        planes = 2 if nchw else ops.planes_of(odt)       # the image gather contracts in bf16x3 (fp32 source) whatever the output dtype
        if not nchw and c % ops.vw_of(odt) != 0:
            raise NotImplementedError(f"{type(self).__name__}: {odt} patch gathers need channels % {ops.vw_of(odt)} == 0 (got {c})")
        if _grad_mode(self) and isinstance(getattr(self, "norm", None), nn.BatchNorm2d):
            from . import grad
            if nchw and (k != 4 or w % 4 != 0):
                raise NotImplementedError("HIP patch embedding of an NCHW image needs patch_size 4 and W % 4 == 0")
            conv = getattr(self, self._conv_name)
            spec = grad.ConvSpec("patch", self.cout, ACT_NONE, self.norm, True, k=k, nchw=nchw, out_dtype=odt)
            return grad.conv_bn_act(spec, self._weights(nchw, planes), x, None, conv.weight, conv.bias, self.norm)
        if c % 4 == 0:
            xr, ld = ops.rows(x)
            if ld != c:
                xr, ld = ops.nhwc(x.contiguous()), c
            kw = dict(M=n * ho * wo, H=ho, W=wo, K=k * k * c, N=self.cout, a0=xr, lda0=c, k0=k * k * c, wp=self._weights(False, planes),
                      ldo=self.cout, gather=ops.GATHER_PATCH, Hin=h, Win=w, Cin=c, ks=k, pk=k * c, dtype=odt)
        else:
            if k != 4 or w % 4 != 0:
                raise NotImplementedError("HIP patch embedding of an NCHW image needs patch_size 4 and W % 4 == 0")
            xr = x.contiguous()                     # NCHW image
            kw = dict(M=n * ho * wo, H=ho, W=wo, K=16 * c, N=self.cout, a0=xr, lda0=0, k0=16 * c, wp=self._weights(True, planes),
                      ldo=self.cout, gather=ops.GATHER_PATCH_NCHW, Hin=h, Win=w, Cin=c, ks=4, pk=0, dtype=odt)
        bn = getattr(self, "norm", None)
        if self.training and isinstance(bn, nn.BatchNorm2d):
            conv = getattr(self, self._conv_name)
            bias = conv.bias.detach().float().contiguous() if conv.bias is not None else None
            stats = ops.new_stats(self.cout, x.device)
            ops.gemm(out=None, e_scale=None, e_shift=bias, stats=stats, **kw)                  # statistics pass
            sc, sh = ops.bn_finalize(bn, stats, self.cout, n * ho * wo, bias=bias)
        else:
            sc, sh = self._affine_eval()
        out = ops.empty_nhwc(n, self.cout, ho, wo, x, dtype=odt)
        ops.gemm(out=out, e_scale=sc, e_shift=sh, **kw)
        return out

    def fuseforward(self, x):
        return self.forward(x)


class PatchEmbed_FasterNet(_PatchConv):
    _conv_name = "proj"

    def __init__(self, in_chans, embed_dim, patch_size, patch_stride, norm_layer=nn.BatchNorm2d):
        super().__init__()
        self._setup(in_chans, embed_dim, patch_size, patch_stride, norm_layer)


class PatchMerging_FasterNet(_PatchConv):
    _conv_name = "reduction"

    def __init__(self, dim, out_dim, k, patch_stride2, norm_layer=nn.BatchNorm2d):
        super().__init__()
        self._setup(dim, out_dim, k, patch_stride2, norm_layer)
        self.reduction.weight._ly_tap_major = True


# --------------------------------------------------------------------------------------------------
# effective Conv (conv + BN + SiLU), k in {1, 3}, stride 1
# --------------------------------------------------------------------------------------------------
def autopad(k, p=None, d=1):
    if d > 1:
        k = d * (k - 1) + 1 if isinstance(k, int) else [d * (x - 1) + 1 for x in k]
    if p is None:
        p = k // 2 if isinstance(k, int) else [x // 2 for x in k]
    return p


class Conv(nn.Module):
    default_act = nn.SiLU()

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, d=1, act=True):
        super().__init__()
        if g != 1 or d != 1 or k not in (1, 3) or autopad(k, p, d) != k // 2 or s not in (1, 2) or (s == 2 and k != 3):
            raise NotImplementedError(f"HIP Conv is built for k in (1, 3), stride 1 (k = 3 also stride 2, inference), groups 1, 'same' padding "
                                      f"(got k={k} s={s} g={g} d={d}); grouped / depthwise Conv (DWConv) is not used by LEAD-YOLO and not built")
        self.s = s
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p, d), groups=g, dilation=d, bias=False)
        if k == 3:
            self.conv.weight._ly_tap_major = True       # optim.FusedSGD may keep this weight's gradient tap-major (what ly_wgrad writes fastest)
        self.bn = nn.BatchNorm2d(c2)
        self.act = self.default_act if act is True else act if isinstance(act, nn.Module) else nn.Identity()
        self.k, self.c1, self.c2 = k, c1, c2
        self._prep = _Prepared()
        self._prep_bn = _Prepared()

    def weights(self, planes=2):
        conv = self.conv
        key = pack.versions(conv.weight)

        def build():
            w = conv.weight
            if self.k == 1:
                return pack.packed(pack.src_matrix(w, self.c2, self.c1), self.c1, planes)
            cip = (self.c1 + 31) // 32 * 32
            return pack.packed(pack.src_taps(w, cip), 9 * cip, planes)
        return self._prep.get(key, build, planes)

    def affine_eval(self):
        conv, bn = self.conv, getattr(self, "bn", None)
        key = pack.versions(conv.bias, *(_bn_tensors(bn) if bn is not None else ())) + (bn.eps if bn is not None else 0,)

        def build():
            if bn is not None:
                return pack.bn_scale_shift(bn, conv.bias)
            return None, (conv.bias.detach().float().contiguous() if conv.bias is not None else None)
        return self._prep_bn.get(key, build)

    def _run(self, x, sc, sh, act, stats=None):
        wp = self.weights(ops.planes_of(x.a0 if isinstance(x, Lazy) else x))
        if self.k == 1:
            return _run_pointwise(x, wp, self.c2, sc, sh, act, stats=stats)
        xr, ld = ops.rows(x)
        n, c, h, w = xr.shape
        out = None if stats is not None else ops.empty_nhwc(n, self.c2, h, w, xr)
        ops.conv3x3(M=n * h * w, H=h, W=w, Cin=c, N=self.c2, x=xr, ldx=ld, wp=wp, out=out, ldo=self.c2, e_scale=sc, e_shift=sh, act=act,
                    stats=stats)
        return out

    @_edge
    def forward(self, x):
        s_ = getattr(self, "s", 1)
        pr = _probe(x, (x.shape[0], self.c2, (x.shape[2] - 1) // s_ + 1, (x.shape[3] - 1) // s_ + 1)) if isinstance(x, torch.Tensor) else None
        if pr is not None:
            return pr
        if self.k == 3 and isinstance(x, Lazy):
            x = x.materialize()
        act = _act_code(self.act)
        bn = getattr(self, "bn", None)
        if getattr(self, "s", 1) == 2:
            # k = 3, stride 2, pad 1 (models/common.py:1890-1910 with s = 2; not instantiated by LEAD-YOLO.yaml): output (oy, ox) of the
            # strided convolution IS output (2 oy, 2 ox) of the stride-1 one, so the stride-1 kernel runs and every second row / column
            # is kept.  Inference only: train-mode batch statistics would have to be taken over the kept pixels.
            if self.training:
                raise NotImplementedError("HIP Conv with stride 2 is built for inference (eval mode) only")
            sc, sh = self.affine_eval()
            return self._run(x, sc, sh, act)[:, :, ::2, ::2].contiguous(memory_format=torch.channels_last)
        if _grad_mode(self) and bn is not None:
            from . import grad
            x0, x1, up = x, None, False
            if isinstance(x, Lazy):
                if x.gate is not None:
                    x0 = x.materialize()
                else:
                    x0, x1, up = x.keep[0], (x.keep[1] if x.a1 is not None else None), x.up
            spec = grad.ConvSpec("pw" if self.k == 1 else "c3", self.c2, act, bn, True, up=up)
            return grad.conv_bn_act(spec, self.weights(ops.planes_of(x0)), x0, x1, self.conv.weight, self.conv.bias, bn)
        if self.training and bn is not None:
            L = Lazy.of(x)
            n, _, h, w = L.shape
            bias = self.conv.bias.detach().float().contiguous() if self.conv.bias is not None else None
            stats = ops.new_stats(self.c2, L.a0.device)
            self._run(x, None, bias, ACT_NONE, stats=stats)                                   # statistics pass
            sc, sh = ops.bn_finalize(bn, stats, self.c2, n * h * w, bias=bias)
        else:
            sc, sh = self.affine_eval()
        return self._run(x, sc, sh, act)

    def forward_fuse(self, x):
        return self.forward(x)


# --------------------------------------------------------------------------------------------------
# RFCBAMConv
# --------------------------------------------------------------------------------------------------
class SE(nn.Module):
    def __init__(self, in_channel, ratio=16):
        super().__init__()
        self.gap = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Sequential(nn.Linear(in_channel, ratio, bias=False), nn.ReLU(), nn.Linear(ratio, in_channel, bias=False),
                                nn.Sigmoid())
        self.ratio = ratio

    def attention(self, xr, ld, n, hw, c):
        wa, wb = self.fc[0].weight.detach(), self.fc[2].weight.detach()
        return ops.se_attention(xr, ld, n, hw, c, wa.contiguous(), wb.contiguous(), self.ratio)

    def forward(self, x):
        x, _ = ops.edge_in(x, "SE", self)
        xr, ld = ops.rows(x)
        n, c, h, w = xr.shape
        return self.attention(xr, ld, n, h * w, c).view(n, c, 1, 1)


class RFCBAMConv(nn.Module):
    def __init__(self, in_channel, out_channel, kernel_size=3, stride=1, dilation=1):
        super().__init__()
        k = kernel_size
        if k not in (1, 3):
            raise NotImplementedError("HIP RFCBAMConv is built for kernel_size 1 and 3")
        if k == 1 and stride != 1:
            raise NotImplementedError("HIP RFCBAMConv with kernel_size 1 is built for stride 1")
        if k == 3 and in_channel % 16 != 0:
            raise NotImplementedError("HIP RFCBAMConv with kernel_size 3 needs in_channel % 16 == 0")
        self.kernel_size, self.stride = k, stride
        self.c, self.o = in_channel, out_channel
        self.generate = nn.Sequential(nn.Conv2d(in_channel, in_channel * k * k, k, padding=k // 2, stride=stride, groups=in_channel,
                                                bias=False), nn.BatchNorm2d(in_channel * k * k), nn.ReLU())
        self.get_weight = nn.Sequential(nn.Conv2d(2, 1, kernel_size=3, padding=1, bias=False), nn.Sigmoid())
        self.se = SE(in_channel)
        self.conv = nn.Sequential(nn.Conv2d(in_channel, out_channel, k, stride=k), nn.BatchNorm2d(out_channel), nn.ReLU())
        if k > 1:
            self.conv[0].weight._ly_tap_major = True    # optim.FusedSGD may keep this gradient [o][kh][kw][c]: what the k = 3 backward produces
        self._prep = _Prepared()

    def _packed(self, planes=2):
        gw, gbn, cw, cbn = self.generate[0].weight, self.generate[1], self.conv[0], self.conv[1]
        key = pack.versions(gw, *_bn_tensors(gbn), cw.weight, cw.bias, *_bn_tensors(cbn), self.get_weight[0].weight) + (gbn.eps, cbn.eps, RF3M)

        def build():
            k, c, o = self.kernel_size, self.c, self.o
            gs, gb = pack.bn_scale_shift(gbn)                                 # [c*k*k]
            w18 = self.get_weight[0].weight.detach().float().reshape(18).contiguous()
            es, eb = pack.bn_scale_shift(cbn, cw.bias)
            if k == 1:
                a1 = (gw.detach().float().view(c) * gs).contiguous()
                return dict(a1=a1, b1=gb, w18=w18, wp=pack.packed(pack.src_matrix(cw.weight, o, c), c, planes), es=es, eb=eb)
            wq_stats = pack.rfcbam_gen_weights(gw, gs, gb, 32, False)           # stats kernel: 32-ch chunks, c0 + w + 4j
            wq_main = pack.rfcbam_gen_weights(gw, gs, gb, 16, True)             # main kernel: 16-ch chunks, c0 + 4w + j
            # conv.0.weight [o, c, 3, 3] read as [o][c/16 chunks][10 tap slots (9 real)][16 channels]: k = chunk*160 + t*16 + ch
            wsrc = pack.Src(cw.weight, o, srb=c * 9, nb=10, vb=9, nc=16, sa=144, sb=1, sc=9)
            d = dict(wq_stats=wq_stats, wq_main=wq_main, w18=w18, wp=pack.packed(wsrc, (c // 16) * 160, planes), es=es, eb=eb)
            if ops.rf3c_ok(c, self.stride):
                d["wq_c"] = pack.rfcbam_gen_weights_c(gw, gs, gb)
                d["wp_c"] = self._wp_c(planes)
            if planes == 1 and RF3M and ops.rf3m_ok(torch.bfloat16, c, o, self.stride):     # (stride 1 / 2 only: other strides keep the lane = pixel kernels)
                d["wm_stats"] = pack.rf3m_stream(gw, gs, gb, pool_stride=self.stride)          # csrc/ly_rf3m.hip: generate on the matrix cores
                d["wm"] = pack.rf3m_stream(gw, gs, gb, cw.weight, 4 if o % 128 == 0 else 2)
            return d
        return self._prep.get(key, build, planes)

    def _wp_c(self, planes):
        """conv.0.weight [o, c, 3, 3] read as [o][c/32 chunks][9 taps][32 channels] (k = chunk*288 + t*32 + ch): the operand of ly_rf3c_fwd"""
        cw, c = self.conv[0], self.c
        return pack.packed(pack.Src(cw.weight, self.o, srb=c * 9, nb=9, nc=32, sa=288, sb=1, sc=9), 9 * c, planes)

    def _packed_train(self, planes=2):
        """what the training node (grad.RfcbamFn) needs of `_packed`: the conv weight image and get_weight's taps.  Keyed on those two
        parameters only — `_packed` also folds the BatchNorm RUNNING statistics, which change every training step, and rebuilding its
        folded generate weights there cost ~25 tiny launches per module and step for images the training path never reads."""
        cw = self.conv[0]
        key = pack.versions(cw.weight, self.get_weight[0].weight) + (RF3C,)

        def build():
            k, c, o = self.kernel_size, self.c, self.o
            w18 = self.get_weight[0].weight.detach().float().reshape(18).contiguous()
            if k == 1:
                return dict(w18=w18, wp=pack.packed(pack.src_matrix(cw.weight, o, c), c, planes))
            if ops.rf3c_ok(c, self.stride) and RF3C:
                return dict(w18=w18, wp_c=self._wp_c(planes))
            wsrc = pack.Src(cw.weight, o, srb=c * 9, nb=10, vb=9, nc=16, sa=144, sb=1, sc=9)
            return dict(w18=w18, wp=pack.packed(wsrc, (c // 16) * 160, planes))
        if not hasattr(self, "_prep_train"):
            self._prep_train = _Prepared()
        return self._prep_train.get(key, build, planes)

    @_edge
    def forward(self, x):
        if isinstance(x, Lazy):
            x = x.materialize()
        k_, s_ = self.kernel_size, self.stride
        pr = _probe(x, (x.shape[0], self.o, (x.shape[2] + 2 * (k_ // 2) - k_) // s_ + 1, (x.shape[3] + 2 * (k_ // 2) - k_) // s_ + 1))
        if pr is not None:
            return pr
        if _grad_mode(self):
            from . import grad
            return grad.rfcbam_train(self, x)
        xr, ld = ops.rows(x)
        n, c, h, w = xr.shape
        k, s = self.kernel_size, self.stride
        P = self._packed(ops.planes_of(xr))
        wa, wb = self.se.fc[0].weight.detach().float().contiguous(), self.se.fc[2].weight.detach().float().contiguous()
        # k = 1, three launches: (1) ONE pass over x leaves the [max, mean] statistics map AND the partial sums of SE's global average
        # pool, (2) SE's linears and get_weight's 3x3 conv on the small maps, (3) the contraction.  k = 3, four: the pooling partials
        # are their own pass.  (The reference reads x for the pool, again for `generate`, and walks the 9x tensor ~13 times.)
        if k == 1:
            a1, b1, es, eb = P["a1"], P["b1"], P["es"], P["eb"]
            if self.training:
                # generate = per-channel scale g_c followed by BatchNorm over (n, h, w): its batch statistics follow
                # from the per-channel moments of x:  mean = g*E[x],  E[a^2] = g^2 * E[x^2]
                gwv = self.generate[0].weight.detach().float().view(c)
                mom = ops.chan_moments(xr, ld, n * h * w, c)
                gs, gb = ops.bn_batch_affine(self.generate[1], gwv * mom[:c], gwv * gwv * mom[c:], n * h * w)
                a1, b1 = (gwv * gs).contiguous(), gb
            mm, part = ops.rfcbam_stats(xr, ld, n, h, w, c, 1, 1, a1=a1, b1=b1, gap=True)
            ca, rfa = ops.rfcbam_mid(part, h * w, wa, wb, self.se.ratio, mm, P["w18"])
            kw = dict(M=n * h * w, H=h, W=w, K=c, N=self.o, a0=xr, lda0=ld, k0=c, wp=P["wp"], ldo=self.o, pro=ops.PRO_AFFINE_RELU_CA,
                      p_scale=a1, p_shift=b1, p_ca=ca, rowscale=rfa)
            if self.training:
                bias = self.conv[0].bias.detach().float().contiguous()
                stats = ops.new_stats(self.o, xr.device)
                ops.gemm(out=None, e_scale=None, e_shift=bias, stats=stats, **kw)               # conv.1 BatchNorm statistics pass
                es, eb = ops.bn_finalize(self.conv[1], stats, self.o, n * h * w, bias=bias)
            out = ops.empty_nhwc(n, self.o, h, w, xr)
            ops.gemm(out=out, e_scale=es, e_shift=eb, act=ACT_RELU, **kw)
            return out
        ho, wo = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
        if RF3M and not self.training and ops.rf3m_ok(xr, c, self.o, s, n, ho, wo):
            return self._forward3_m(xr, ld, n, c, h, w, ho, wo, s, P, wa, wb)
        # (fp32 storage with more than 128 output channels: the lane = pixel kernels measure faster — layer 20 at bs=64: 215 vs 289 us —
        # the two-plane operand tile of the lane = channel kernel leaves one block per CU there)
        if ops.rf3c_ok(c, s) and RF3C and not (xr.dtype == torch.float32 and self.o > 128):
            return self._forward3_c(xr, ld, n, c, h, w, ho, wo, s, P, wa, wb)
        th, tw = ops.pick_tile(ho, wo)
        wq_stats, wq_main, es, eb = P["wq_stats"], P["wq_main"], P["es"], P["eb"]
        if self.training:
            gw = self.generate[0].weight
            s1, s2, cnt = ops.rfcbam_generate_stats(xr, ld, n, h, w, c, s, gw)                # generate.1 batch statistics
            gs, gb = ops.bn_batch_affine(self.generate[1], s1, s2, cnt)
            wq_stats = pack.rfcbam_gen_weights(gw, gs, gb, 32, False)
            wq_main = pack.rfcbam_gen_weights(gw, gs, gb, 16, True)
        part = ops.colsum(xr, ld, n, h * w, c)                                                 # k = 3: pooling partials as their own pass
        mm = ops.rfcbam_stats(xr, ld, n, h, w, c, 3, s, wg=wq_stats, th=th, tw=tw)
        ca, rfa = ops.rfcbam_mid(part, h * w, wa, wb, self.se.ratio, mm, P["w18"])
        kw = dict(n=n, h=h, w=w, c=c, ho=ho, wo=wo, N=self.o, s=s, th=th, tw=tw, x=xr, ldx=ld, wg=wq_main, ca=ca, rfa=rfa, wp=P["wp"],
                  ldo=self.o)
        if self.training:
            bias = self.conv[0].bias.detach().float().contiguous()
            stats = ops.new_stats(self.o, xr.device)
            ops.rfcbam3(out=None, e_scale=torch.ones_like(bias), e_shift=bias, stats=stats, **kw)   # conv.1 statistics pass
            es, eb = ops.bn_finalize(self.conv[1], stats, self.o, n * ho * wo, bias=bias)
        out = ops.empty_nhwc(n, self.o, ho, wo, xr)
        ops.rfcbam3(out=out, e_scale=es, e_shift=eb, **kw)
        return out


    def _forward3_c(self, xr, ld, n, c, h, w, ho, wo, s, P, wa, wb):
        """k = 3 on the lane = channel kernels (csrc/ly_rf3c.hip), three launches, x read twice: (1) [max, mean] map + SE pooling partials,
        (2) SE linears + get_weight's conv on the small maps, (3) regenerate + contraction."""
        th, tw = ops.pick_tile_c(ho, wo, s)
        wq, es, eb = P["wq_c"], P["es"], P["eb"]
        if self.training:
            gw = self.generate[0].weight
            s1, s2, cnt = ops.rfcbam_generate_stats(xr, ld, n, h, w, c, s, gw)                # generate.1 batch statistics
            gs, gb = ops.bn_batch_affine(self.generate[1], s1, s2, cnt)
            wq = pack.rfcbam_gen_weights_c(gw, gs, gb)
        mm, part = ops.rf3c_stats(xr, ld, n, h, w, c, s, wq, th, tw)
        ca, rfa = ops.rfcbam_mid(part, h * w, wa, wb, self.se.ratio, mm, P["w18"])
        kw = dict(n=n, h=h, w=w, c=c, ho=ho, wo=wo, N=self.o, s=s, th=th, tw=tw, x=xr, ldx=ld, wq=wq, ca=ca, rfa=rfa, wp=P["wp_c"], ldo=self.o)
        if self.training:
            bias = self.conv[0].bias.detach().float().contiguous()
            stats = ops.new_stats(self.o, xr.device)
            ops.rf3c_fwd(out=None, e_scale=ops.ones_f32(bias.numel(), bias.device), e_shift=bias, stats=stats, **kw)   # conv.1 statistics pass
            es, eb = ops.bn_finalize(self.conv[1], stats, self.o, n * ho * wo, bias=bias)
        out = ops.empty_nhwc(n, self.o, ho, wo, xr)
        ops.rf3c_fwd(out=out, e_scale=es, e_shift=eb, **kw)
        return out


    def _forward3_m(self, xr, ld, n, c, h, w, ho, wo, s, P, wa, wb):
        """k = 3, bf16, inference: `generate` as block-diagonal MFMA products whose accumulators feed the main contraction in registers
        (csrc/ly_rf3m.hip) — (1) [max, mean] map + SE pooling partials, (2) SE linears + get_weight's conv, (3) generate + contraction"""
        th, tw = ops.pick_tile_m(ho, wo, s)
        mm, part = ops.rf3m_stats(xr, ld, n, h, w, c, s, P["wm_stats"], th, tw)
        ca, rfa = ops.rfcbam_mid(part, h * w, wa, wb, self.se.ratio, mm, P["w18"])
        out = ops.empty_nhwc(n, self.o, ho, wo, xr)
        ops.rf3m_fwd(n=n, h=h, w=w, c=c, ho=ho, wo=wo, N=self.o, s=s, th=th, tw=tw, x=xr, ldx=ld, ca=ca, rfa=rfa, wp=P["wm"], e_scale=P["es"],
                     e_shift=P["eb"], out=out, ldo=self.o)
        return out


RF3M = True        # tools / tests: False keeps the lane = channel kernels for the bf16 inference forward of k = 3
RF3C = True        # tools / tests: False runs the first-generation k=3 kernels (lane = pixel) for A/B comparisons


# --------------------------------------------------------------------------------------------------
# C3_CA
# --------------------------------------------------------------------------------------------------
class h_sigmoid(nn.Module):
    def __init__(self, inplace=True):
        super().__init__()
        self.relu = nn.ReLU6(inplace=inplace)


class h_swish(nn.Module):
    def __init__(self, inplace=True):
        super().__init__()
        self.sigmoid = h_sigmoid(inplace=inplace)


class CoordAtt(nn.Module):
    def __init__(self, inp, oup, reduction=32):
        super().__init__()
        if inp != oup:
            raise NotImplementedError("HIP CoordAtt gates its own input: inp must equal oup")
        self.pool_h = nn.AdaptiveAvgPool2d((None, 1))
        self.pool_w = nn.AdaptiveAvgPool2d((1, None))
        mip = max(8, inp // reduction)
        self.conv1 = nn.Conv2d(inp, mip, kernel_size=1, stride=1, padding=0)
        self.bn1 = nn.BatchNorm2d(mip)
        self.act = h_swish()
        self.conv_h = nn.Conv2d(mip, oup, kernel_size=1, stride=1, padding=0)
        self.conv_w = nn.Conv2d(mip, oup, kernel_size=1, stride=1, padding=0)
        self.mip, self.c = mip, inp
        self._prep = _Prepared()
        self._prep_bn = _Prepared()
        self.trainable_on_hip = mip in (8, 16) and inp <= 512      # widths ly_coordatt_mlp_bwd is instantiated for (grad.coordatt_train)

    def train(self, mode=True):
        """Eval runs at any width.  Training is built for mip = max(8, c // 32) in (8, 16) and c <= 512 (every CoordAtt of lead-yolo-n / s / l);
        a custom width (c_ = 384 -> mip 12, c_ = 640 -> mip 20) is refused HERE — when the model is put into training mode with parameters that
        ask for gradients — not in the middle of the first training forward."""
        if mode and not self.trainable_on_hip and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError(f"CoordAtt(c={self.c}, mip={self.mip}): the HIP training path is built for mip in (8, 16) and c <= 512; this width "
                                      "runs in eval mode only (freeze the module's parameters or pick a width with c // 32 in {<= 8, 16})")
        return super().train(mode)

    def _weights(self):
        key = pack.versions(self.conv1.weight, self.conv1.bias, self.conv_h.weight, self.conv_h.bias, self.conv_w.weight, self.conv_w.bias)

        def build():
            f = lambda p: p.detach().float().reshape(p.shape[0], -1).contiguous()
            return (f(self.conv1.weight), self.conv1.bias.detach().float().contiguous(), f(self.conv_h.weight),
                    self.conv_h.bias.detach().float().contiguous(), f(self.conv_w.weight), self.conv_w.bias.detach().float().contiguous())
        return self._prep.get(key, build)

    def _conv1_eval(self):
        key = pack.versions(self.conv1.weight, self.conv1.bias, *_bn_tensors(self.bn1)) + (self.bn1.eps,)

        def build():
            s, t = pack.bn_scale_shift(self.bn1, self.conv1.bias)
            return (self.conv1.weight.detach().float().view(self.mip, self.c) * s.view(-1, 1)).contiguous(), t
        return self._prep_bn.get(key, build)

    def attention(self, xr, ld, n, h, w, c):
        w1raw, b1raw, wh, bh, ww, bw = self._weights()
        pool = ops.pool_hw(xr, ld, n, h, w, c)
        if self.training:
            st = ops.coordatt_conv1_stats(pool, n * (h + w), c, self.mip, w1raw, b1raw).float()       # bn1 batch statistics (double accumulators)
            sc, sh = ops.bn_batch_affine(self.bn1, st[:self.mip], st[self.mip:], n * (h + w))
            w1, b1 = (w1raw * sc.view(-1, 1)).contiguous(), (b1raw * sc + sh).contiguous()
        else:
            w1, b1 = self._conv1_eval()
        return ops.coordatt_mlp(pool, n, h, w, c, self.mip, w1, b1, wh, bh, ww, bw)

    @_edge
    def forward(self, x):
        pr = _probe(x, x.shape)
        if pr is not None:
            return pr
        if _grad_mode(self):
            from . import grad
            return grad.coordatt_train(self, x)
        xr, ld = ops.rows(x)
        n, c, h, w = xr.shape
        a_h, a_w = self.attention(xr, ld, n, h, w, c)
        return ops.coordatt_gate(xr, ld, n, h, w, c, a_h, a_w)


class CA_Bottleneck(nn.Module):
    def __init__(self, c1, c2, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_, c2, 3, 1, g=g)
        self.ca = CoordAtt(c2, c2, 32)
        self.add = shortcut and c1 == c2

    def forward_lazy(self, x):
        """Returns a Lazy (gated, un-materialised) output when there is no residual, else a tensor."""
        if _grad_mode(self):
            xin = x.materialize() if isinstance(x, Lazy) else x
            y = self.ca(self.cv2(self.cv1(xin)))
            return xin + y if self.add else y
        src = Lazy.of(x)
        t1 = self.cv1(src)
        t2 = self.cv2(t1)
        n, c, h, w = t2.shape
        a_h, a_w = self.ca.attention(t2, c, n, h, w, c)
        if self.add:
            if src.gate is not None or src.a1 is not None or src.up:
                rr, ldr = ops.rows(src.materialize())
            else:
                rr, ldr = src.a0, src.lda0
            return ops.coordatt_gate(t2, c, n, h, w, c, a_h, a_w, rr, ldr)
        return Lazy((n, c, h, w), t2, c, c, gate=(a_h, a_w), keep=(t2, a_h, a_w))

    @_edge
    def forward(self, x):
        pr = _probe(x, (x.shape[0], self.cv2.c2, x.shape[2], x.shape[3])) if isinstance(x, torch.Tensor) else None
        if pr is not None:
            return pr
        y = self.forward_lazy(x)
        return y.materialize() if isinstance(y, Lazy) else y


class C3_CA(nn.Module):
    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c1, c_, 1, 1)
        self.cv3 = Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(*(CA_Bottleneck(c_, c_, shortcut, g, e=1.0) for _ in range(n)))
        self.c_, self.c2 = c_, c2
        self._prep = _Prepared()
        self._prep_bn = _Prepared()
        self.cv1.conv.weight._ly_grad_pair = self.cv2.conv.weight      # optim.FusedSGD may keep the two gradients adjacent (one wgrad launch)

    def _weights12(self, planes=2):
        """cv1 and cv2 read the same input: one GEMM with stacked weights writes [cv1 | cv2] side by side,
        which is also exactly where the later concat wants cv2's output."""
        p1, p2 = self.cv1, self.cv2
        key = pack.versions(p1.conv.weight, p2.conv.weight)
        c1 = p1.conv.weight.shape[1]

        def build():
            if self.c_ % 16 == 0:
                return pack.packed([pack.src_matrix(p1.conv.weight, self.c_, c1), pack.src_matrix(p2.conv.weight, self.c_, c1)], c1, planes)
            return pack.frag_pack3(torch.cat((p1.conv.weight.detach().view(self.c_, -1), p2.conv.weight.detach().view(self.c_, -1)), 0), planes=planes)
        return self._prep.get(key, build, planes)

    def _cv12_train(self, x):
        """cv1(x), cv2(x) under autograd: ONE node over the shared input (grad.ConvBnActPair) when both are 1x1 conv -> train-mode BatchNorm
        units of a vector-friendly width, else the two Conv modules on their own"""
        p1, p2 = self.cv1, self.cv2
        b1, b2 = getattr(p1, "bn", None), getattr(p2, "bn", None)
        gated = isinstance(x, Lazy) and x.gate is not None
        if (b1 is None or b2 is None or not (b1.training and b2.training) or p1.k != 1 or p2.k != 1 or p1.conv.bias is not None
                or p2.conv.bias is not None or gated or self.c_ % 16 or b1.weight is None or b2.weight is None):
            return self.cv1(x), self.cv2(x)
        from . import grad
        if pack.src_matrix_kcat_t(p1.conv.weight, p2.conv.weight) is None:
            return self.cv1(x), self.cv2(x)
        x0, x1, up = x, None, False
        if isinstance(x, Lazy):
            x0, x1, up = x.keep[0], (x.keep[1] if x.a1 is not None else None), x.up
        return grad.conv_bn_act_pair(_act_code(p1.act), up, self._weights12(ops.planes_of(x0)), x0, x1, p1.conv, b1, p2.conv, b2)

    def _affine12_eval(self):
        p1, p2 = self.cv1, self.cv2
        b1, b2 = getattr(p1, "bn", None), getattr(p2, "bn", None)
        key = pack.versions(p1.conv.bias, p2.conv.bias, *(_bn_tensors(b1) if b1 is not None else ()),
                            *(_bn_tensors(b2) if b2 is not None else ()))

        def build():
            parts = []
            for p, b in ((p1, b1), (p2, b2)):
                if b is not None:
                    parts.append(pack.bn_scale_shift(b, p.conv.bias))
                else:
                    parts.append((torch.ones(self.c_, device=p.conv.weight.device), p.conv.bias.detach().float()))
            return torch.cat((parts[0][0], parts[1][0])).contiguous(), torch.cat((parts[0][1], parts[1][1])).contiguous()
        return self._prep_bn.get(key, build)

    @_edge
    def forward(self, x):
        pr = _probe(x, (x.shape[0], self.c2, x.shape[2], x.shape[3])) if isinstance(x, torch.Tensor) else None
        if pr is not None:
            return pr
        src = Lazy.of(x)
        n, c, h, w = src.shape
        c_ = self.c_
        if _act_code(self.cv1.act) != _act_code(self.cv2.act):
            raise NotImplementedError("C3_CA: cv1 and cv2 must share one activation")
        if _grad_mode(self):
            a, b = self._cv12_train(x)
            for blk in self.m:
                a = blk.forward_lazy(a)
            n_, ca_, h_, w_ = a.shape
            ta, lda = ops.rows(a)
            tb, ldb = ops.rows(b)
            return self.cv3(Lazy((n_, ca_ + b.shape[1], h_, w_), ta, lda, ca_, a1=tb, lda1=ldb, keep=(ta, tb)))
        wp = self._weights12(ops.planes_of(src.a0))
        b1, b2 = getattr(self.cv1, "bn", None), getattr(self.cv2, "bn", None)
        if self.training and b1 is not None and b2 is not None:
            stats = ops.new_stats(2 * c_, src.a0.device)
            _run_pointwise(src, wp, 2 * c_, None, None, ACT_NONE, stats=stats)                  # statistics pass, both halves
            sa, ta = ops.bn_finalize(b1, stats, 2 * c_, n * h * w, n=c_, c_off=0)
            sb, tb = ops.bn_finalize(b2, stats, 2 * c_, n * h * w, n=c_, c_off=c_)
            sc, sh = torch.cat((sa, sb)).contiguous(), torch.cat((ta, tb)).contiguous()
        else:
            sc, sh = self._affine12_eval()
        ycat = ops.empty_nhwc(n, 2 * c_, h, w, src.a0)
        _run_pointwise(src, wp, 2 * c_, sc, sh, _act_code(self.cv1.act), out=ycat, ldo=2 * c_)
        cur = Lazy((n, c_, h, w), ycat, 2 * c_, c_, keep=(ycat,))
        for blk in self.m:
            cur = Lazy.of(blk.forward_lazy(cur))
        right = ycat[:, c_:]
        both = Lazy((n, 2 * c_, h, w), cur.a0, cur.lda0, c_, a1=right, lda1=2 * c_, gate=cur.gate, keep=cur.keep + (ycat,))
        return self.cv3(both)


# --------------------------------------------------------------------------------------------------
# graph remainder
# --------------------------------------------------------------------------------------------------
class SPPF(nn.Module):
    def __init__(self, c1, c2, k=5):
        super().__init__()
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * 4, c2, 1, 1)
        self.m = nn.MaxPool2d(kernel_size=k, stride=1, padding=k // 2)

    @_edge
    def forward(self, x):
        y = self.cv1(x)
        if SHAPE_PROBE and not y.is_cuda:
            return y.new_zeros((y.shape[0], self.cv2.c2, y.shape[2], y.shape[3]))
        n, c_, h, w = y.shape
        k = self.m.kernel_size
        if _grad_mode(self):
            from . import grad
            return self.cv2(grad.SppfPool.apply(y, k))
        if ops.sppf_pool_fits(h, w):
            yr, ld = ops.rows(y)
            buf = ops.empty_nhwc(n, 4 * c_, h, w, yr)
            ops.sppf_pool(yr, ld, n, h, w, c_, k, buf, 4 * c_)          # [y | m(y) | m(m(y)) | m(m(m(y)))] in one pass
            return self.cv2(buf)
        y1 = self.m(y)                                                    # large maps: stock GPU max-pool
        y2 = self.m(y1)
        return self.cv2(torch.cat((y, y1, y2, self.m(y2)), 1))


class Upsample(nn.Upsample):
    """nn.Upsample; with `lazy` set (by Model) a nearest 2x upsample is not executed but handed to the
    consumer GEMM, which reads source row (n, h/2, w/2) instead (ly_gemm_fwd LY_GATHER_UP2)."""
    lazy = False

    def forward(self, x):
        if self.lazy and not isinstance(x, Lazy) and x.dim() == 4 and x.is_cuda and self.mode == "nearest" \
                and self.size is None and float(self.scale_factor) == 2.0 and x.shape[1] % ops.vw_of(x) == 0 \
                and x.dtype in (torch.float32, torch.bfloat16):
            t, ld = ops.rows(x)
            n, c, h, w = t.shape
            return Lazy((n, c, 2 * h, 2 * w), t, ld, c, up=True, keep=(t,))
        if isinstance(x, Lazy):
            x = x.materialize()
        return super().forward(x)


class Concat(nn.Module):
    """torch.cat; with `lazy` set (by Model) a two-way channel concat becomes a two-source Lazy that
    the consumer GEMM reads in place (no copy)."""
    lazy = False

    def __init__(self, dimension=1):
        super().__init__()
        self.d = dimension

    def forward(self, x):
        if self.lazy and self.d == 1 and len(x) == 2 and not isinstance(x[1], Lazy) and x[1].is_cuda:
            a = Lazy.of(x[0])
            vw = ops.vw_of(x[1])
            if a.a1 is None and a.gate is None and a.shape[1] % vw == 0 and x[1].shape[1] % vw == 0 and a.a0.dtype == x[1].dtype \
                    and x[1].dtype in (torch.float32, torch.bfloat16) and tuple(a.shape[2:]) == tuple(x[1].shape[2:]):
                t1, ld1 = ops.rows(x[1])
                n, c0, h, w = a.shape
                return Lazy((n, c0 + t1.shape[1], h, w), a.a0, a.lda0, c0, a1=t1, lda1=ld1, up=a.up, keep=(a.keep[0], t1))
        x = [t.materialize() if isinstance(t, Lazy) else t for t in x]
        return torch.cat(x, self.d)


FUSED_DETECT_LEVEL = True      # development switch: False sends Detect levels through the GEMM + ly_detect_tail pair


class Detect(nn.Module):
    stride = None
    dynamic = False
    export = False

    def __init__(self, nc=80, anchors=(), ch=(), inplace=True):
        super().__init__()
        self.nc = nc
        self.no = nc + 5
        self.nl = len(anchors)
        self.na = len(anchors[0]) // 2
        self.grid = [torch.empty(0) for _ in range(self.nl)]
        self.anchor_grid = [torch.empty(0) for _ in range(self.nl)]
        self.register_buffer("anchors", torch.tensor(anchors).float().view(self.nl, -1, 2))
        self.m = nn.ModuleList(nn.Conv2d(x, self.no * self.na, 1) for x in ch)
        self.inplace = inplace
        self._prep = [_Prepared() for _ in ch]

    def _packed(self, i, planes):
        conv = self.m[i]
        key = pack.versions(conv.weight, conv.bias)
        return self._prep[i].get(key, lambda: (pack.packed(pack.src_matrix(conv.weight, conv.out_channels, conv.in_channels), conv.in_channels, planes),
                                               conv.bias.detach().float().contiguous()), planes)

    def _packed_nat(self, i, planes):
        """the head weight as two 16-row tiles in natural k order (ly_detect_level) + fp32 bias"""
        conv = self.m[i]
        key = pack.versions(conv.weight, conv.bias)
        return self._prep[i].get(key, lambda: (pack.frag_pack_nat(conv.weight.detach().float().view(conv.out_channels, conv.in_channels), planes),
                                               conv.bias.detach().float().contiguous()), ("nat", planes))

    def _fused_level_input(self, i, xi):
        """the feature map as dense rows when the one-launch level kernel takes it (plain tensor, shape built), else None"""
        if isinstance(xi, Lazy) or not isinstance(xi, torch.Tensor) or xi.dim() != 4 or not FUSED_DETECT_LEVEL:
            return None
        conv = self.m[i]
        if conv.bias is None or not ops.detect_level_ok(conv.in_channels, self.na, self.no, xi.dtype):
            return None
        rows, ld = ops.rows(xi)
        return (rows, ld) if ld % ops.vw_of(xi.dtype) == 0 and rows.data_ptr() % 16 == 0 else None

    def _head(self, i, x):
        conv = self.m[i]
        L = Lazy.of(x)
        wp, b = self._packed(i, ops.planes_of(L.a0))
        ldo = (conv.out_channels + 3) // 4 * 4
        n, c, h, w = L.shape
        buf = torch.empty((n, h, w, ldo), dtype=L.a0.dtype, device=L.a0.device)
        _run_pointwise(L, wp, conv.out_channels, None, b, ACT_NONE, out=buf, ldo=ldo)
        return buf, ldo                                         # [n, h, w, ldo] rows, first na*no columns valid

    def _strides(self):
        key = (self.stride.data_ptr(), self.stride._version)
        if getattr(self, "_stride_key", None) != key:           # python floats cached: no device sync per forward
            self._stride_f = [float(v) for v in self.stride.detach().cpu()]
            self._stride_key = key
        return self._stride_f

    def forward(self, x):
        x = [ops.edge_in(t, "Detect", self)[0] if isinstance(t, torch.Tensor) and not (SHAPE_PROBE and not t.is_cuda) else t for t in x]
        with torch.autocast("cuda", enabled=False):
            return self._fwd(x)

    def _fwd(self, x):
        if _grad_mode(self):
            from . import grad
            for i in range(self.nl):
                t = x[i].materialize() if isinstance(x[i], Lazy) else x[i]
                conv = self.m[i]
                wp, _ = self._packed(i, ops.planes_of(t))
                # raw maps go to the loss as fp32 whatever the storage dtype (the loss is fp32, as under the reference's autocast):
                # the permuting copy and the conversion are one pass, and so is their adjoint (grad.DetectHeadFn)
                x[i] = grad.detect_head(self, i, wp, t, conv.weight, conv.bias)
            return x
        st = getattr(self, "_early", None)                      # levels already launched by Model._forward_once (side stream)
        self._early = None
        if st is None:
            shapes = [Lazy.of(t).shape for t in x]
            st = self.begin(shapes[0][0], [s_[2:] for s_ in shapes], Lazy.of(x[0]).a0.device)
        for i in range(self.nl):
            if st["p"][i] is None:
                self.level(st, i, x[i])
        if st["forked"]:
            torch.cuda.current_stream().wait_stream(_side_stream(st["device"]))
        out, z = st["p"], st["z"]
        return out if self.training else (z,) if self.export else (z, out)

    # ---- per-level API (lets the model launch a level as soon as its feature map exists) -----------------
    def begin(self, bs, hw, device):
        """allocate the outputs for feature maps of sizes hw[i] = (ny, nx); returns the state `level` fills"""
        decode = not self.training
        rows = [self.na * ny * nx for ny, nx in hw]
        ext = getattr(self, "_out", None)                       # caller-provided output storage (graph.GraphedForward: batch slices)
        if ext is not None and decode:
            z = ext["z"]
            assert tuple(z.shape) == (bs, sum(rows), self.no) and z.is_contiguous()
        else:
            z = torch.empty((bs, sum(rows), self.no), dtype=torch.float32, device=device) if decode else None
        offs = [sum(rows[:i]) for i in range(self.nl)]
        return dict(z=z, zrows=sum(rows), offs=offs, hw=[tuple(v) for v in hw], p=[None] * self.nl, bs=bs, device=device, forked=False)

    def level(self, st, i, xi, side=False):
        """head conv + decode of level i; with side=True on the auxiliary stream (joined in forward)"""
        main = torch.cuda.current_stream()
        ctx = torch.cuda.stream(_side_stream(st["device"])) if side else None
        if side:
            _side_stream(st["device"]).wait_stream(main)
            st["forked"] = True
            ctx.__enter__()
        try:
            ny, nx = st["hw"][i]
            ext = getattr(self, "_out", None)
            p = ext["p"][i] if ext is not None else torch.empty((st["bs"], self.na, ny, nx, self.no), dtype=torch.float32, device=st["device"])
            fused = self._fused_level_input(i, xi)
            if fused is not None:                           # head convolution + decode in one launch (csrc/ly_detect.hip)
                wp, b = self._packed_nat(i, ops.planes_of(xi))
                ops.detect_level(fused[0], fused[1], st["bs"], ny, nx, self.m[i].in_channels, wp, b, self.na, self.no, self.anchors[i],
                                 self._strides()[i], p, st["z"], st["zrows"], st["offs"][i])
            else:
                buf, ldo = self._head(i, xi)
                ops.detect_tail(buf, ldo, st["bs"], ny, nx, self.na, self.no, self.anchors[i], self._strides()[i], p, st["z"], st["zrows"],
                                st["offs"][i])
            if side:
                p.record_stream(main)
            st["p"][i] = p
        finally:
            if side:
                ctx.__exit__(None, None, None)

    def _make_grid(self, nx=20, ny=20, i=0):
        d, t = self.anchors[i].device, self.anchors[i].dtype
        shape = 1, self.na, ny, nx, 2
        ys, xs = torch.arange(ny, device=d, dtype=t), torch.arange(nx, device=d, dtype=t)
        yv, xv = torch.meshgrid(ys, xs, indexing="ij")
        grid = torch.stack((xv, yv), 2).expand(shape) - 0.5
        anchor_grid = (self.anchors[i] * self.stride[i]).view((1, self.na, 1, 1, 2)).expand(shape)
        return grid, anchor_grid
