"""ORACLE — CPU restatement of the LEAD-YOLO hot path (TEST INFRASTRUCTURE, not product code).

A purely *functional* torch-CPU fp32 restatement of the reference's detector forward path, its yaml
graph rules, its target assignment / loss and its optimiser step.  Every function takes a flat
``state`` dict (name -> tensor, the reference's own state_dict keys) and a key ``prefix`` — there are
no nn.Module classes here on purpose: this file restates the *arithmetic*, the product
(lead-yolo_amd/) re-implements it in HIP, and tests compare the two.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Parity pinning: the reference has no tests or golden vectors of its own for this path,
so this oracle is pinned against outputs of the reference itself, imported in the build container by
oracle/gen_golden.py (fixtures under tests/golden/, regenerable with that script while /root/reference is
present) and checked against them by tests/test_oracle_golden.py.

Reference citations are `file:line` into the upstream checkout (qingqing-zijin/LEAD-YOLO @ 2024-12-20).
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-3        # utils/torch_utils.py:218  (initialize_weights rewrites every BatchNorm2d)
BN_MOMENTUM = 0.03   # utils/torch_utils.py:219


# --------------------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------------------
def batchnorm(state, prefix, x, training=False, eps=BN_EPS, momentum=BN_MOMENTUM):
    """nn.BatchNorm2d semantics through the same ATen primitive the reference reaches
    (torch.nn.functional.batch_norm).  eval: running stats.  train: biased batch variance for the
    normalisation, unbiased for the running_var update; running stats in `state` updated in place.
    (A hand-composed mean/var formulation is mathematically identical but its fp32 autograd is ~1e-2
    less accurate than the fused backward, so the primitive is used on purpose.)"""
    w, b = state[prefix + "weight"], state[prefix + "bias"]
    rm, rv = state[prefix + "running_mean"], state[prefix + "running_var"]
    if training:
        key = prefix + "num_batches_tracked"
        if key in state:
            state[key] += 1
    return F.batch_norm(x, rm, rv, w, b, training, momentum, eps)


def silu(x):
    return x * torch.sigmoid(x)


def h_swish(x):
    """models/common.py:1565-1580:  x * relu6(x + 3) / 6"""
    return x * (torch.clamp(x + 3.0, 0.0, 6.0) / 6.0)


def conv_bn_silu(state, prefix, x, k=1, s=1, training=False):
    """Effective `Conv` (models/common.py:1890-1910): conv(no bias, pad k//2) -> BN -> SiLU.
    After fuse() the `bn.*` keys are gone and `conv.bias` exists (forward_fuse)."""
    w = state[prefix + "conv.weight"]
    bias = state.get(prefix + "conv.bias")
    y = F.conv2d(x, w, bias, stride=s, padding=k // 2)
    if prefix + "bn.weight" in state:
        y = batchnorm(state, prefix + "bn.", y, training)
    return silu(y)


# --------------------------------------------------------------------------------------------------
# A — FasterNet
# --------------------------------------------------------------------------------------------------
def patch_conv(state, prefix, x, k, conv_name, training=False):
    """PatchEmbed_FasterNet / PatchMerging_FasterNet (models/common.py:1528-1561):
    non-overlapping k x k stride-k conv (no bias) then BN; fused form has conv bias and no `norm.*`."""
    w = state[prefix + conv_name + ".weight"]
    bias = state.get(prefix + conv_name + ".bias")
    y = F.conv2d(x, w, bias, stride=k)
    if prefix + "norm.weight" in state:
        y = batchnorm(state, prefix + "norm.", y, training)
    return y


def partial_conv3(state, prefix, x):
    """Partial_conv3.forward_split_cat (models/common.py:1432-1437): 3x3/s1/p1 conv without bias on
    the first C//4 channels; the remaining channels pass through."""
    w = state[prefix + "partial_conv3.weight"]
    cq = w.shape[0]
    head = F.conv2d(x[:, :cq], w, None, stride=1, padding=1)
    return torch.cat((head, x[:, cq:]), 1)


def mlp_block(state, prefix, x, training=False):
    """MLPBlock.forward (models/common.py:1478-1482): x + W2 . relu(BN(W1 . pconv(x)))."""
    y = partial_conv3(state, prefix + "spatial_mixing.", x)
    y = F.conv2d(y, state[prefix + "mlp.0.weight"])
    y = batchnorm(state, prefix + "mlp.1.", y, training)
    y = F.relu(y)
    y = F.conv2d(y, state[prefix + "mlp.3.weight"])
    return x + y


def basic_stage(state, prefix, x, training=False):
    """BasicStage.forward (models/common.py:1523-1525): `depth` MLPBlocks in sequence."""
    d = 0
    while f"{prefix}blocks.{d}.mlp.0.weight" in state:
        x = mlp_block(state, f"{prefix}blocks.{d}.", x, training)
        d += 1
    assert d > 0, f"no blocks under {prefix}"
    return x


# --------------------------------------------------------------------------------------------------
# B — RFCBAMConv
# --------------------------------------------------------------------------------------------------
def se_attention(state, prefix, x):
    """rfa.SE.forward (models/rfa.py:88-92): sigmoid(Wb . relu(Wa . GAP(x))), no biases."""
    g = x.mean((2, 3))
    h = F.relu(g @ state[prefix + "fc.0.weight"].t())
    return torch.sigmoid(h @ state[prefix + "fc.2.weight"].t())          # [b, c]


def rfcbam(state, prefix, x, k, s, training=False, return_intermediates=False):
    """RFCBAMConv.forward (models/rfa.py:113-129)."""
    b, c, _, _ = x.shape
    ca = se_attention(state, prefix + "se.", x)                           # [b, c]
    g = F.conv2d(x, state[prefix + "generate.0.weight"], None, stride=s, padding=k // 2, groups=c)
    g = F.relu(batchnorm(state, prefix + "generate.1.", g, training))    # [b, c*k*k, h, w]
    h, w = g.shape[2:]
    # channel c*k*k + n1*k + n2  ->  expanded pixel (h*k + n1, w*k + n2)   (rfa.py:121-122)
    G = g.view(b, c, k, k, h, w).permute(0, 1, 4, 2, 5, 3).reshape(b, c, h * k, w * k)
    mm = torch.cat((G.max(1, keepdim=True)[0], G.mean(1, keepdim=True)), 1)   # un-attended G
    rfa = torch.sigmoid(F.conv2d(mm, state[prefix + "get_weight.0.weight"], None, padding=1))
    y = F.conv2d(G * ca.view(b, c, 1, 1) * rfa, state[prefix + "conv.0.weight"],
                 state[prefix + "conv.0.bias"], stride=k)
    y = F.relu(batchnorm(state, prefix + "conv.1.", y, training))
    if return_intermediates:
        return y, {"ca": ca, "mm": mm, "rfa": rfa}
    return y


# --------------------------------------------------------------------------------------------------
# C — C3_CA
# --------------------------------------------------------------------------------------------------
def coord_att(state, prefix, x, training=False, return_intermediates=False):
    """CoordAtt.forward (models/common.py:1595-1609)."""
    n, c, h, w = x.shape
    ph = x.mean(3, keepdim=True)                        # [n,c,h,1]
    pw = x.mean(2, keepdim=True).permute(0, 1, 3, 2)    # [n,c,w,1]
    y = torch.cat((ph, pw), 2)
    y = F.conv2d(y, state[prefix + "conv1.weight"], state[prefix + "conv1.bias"])
    y = h_swish(batchnorm(state, prefix + "bn1.", y, training))
    yh, yw = y[:, :, :h], y[:, :, h:].permute(0, 1, 3, 2)
    a_h = torch.sigmoid(F.conv2d(yh, state[prefix + "conv_h.weight"], state[prefix + "conv_h.bias"]))
    a_w = torch.sigmoid(F.conv2d(yw, state[prefix + "conv_w.weight"], state[prefix + "conv_w.bias"]))
    out = x * a_w * a_h
    if return_intermediates:
        return out, {"a_h": a_h, "a_w": a_w}
    return out


def ca_bottleneck(state, prefix, x, shortcut, training=False):
    """CA_Bottleneck.forward (models/common.py:1622-1623); e=1.0 so c_ == c2; residual only when
    `shortcut` and c1 == c2 (LEAD-YOLO.yaml passes False)."""
    y = conv_bn_silu(state, prefix + "cv1.", x, 1, 1, training)
    y = conv_bn_silu(state, prefix + "cv2.", y, 3, 1, training)
    y = coord_att(state, prefix + "ca.", y, training)
    c1 = state[prefix + "cv1.conv.weight"].shape[1]
    c2 = state[prefix + "cv2.conv.weight"].shape[0]
    return x + y if (shortcut and c1 == c2) else y


def c3_ca(state, prefix, x, shortcut=True, training=False):
    """C3_CA.forward (models/common.py:1636-1637): cv3(cat(m(cv1(x)), cv2(x)))."""
    a = conv_bn_silu(state, prefix + "cv1.", x, 1, 1, training)
    j = 0
    while f"{prefix}m.{j}.cv1.conv.weight" in state:
        a = ca_bottleneck(state, f"{prefix}m.{j}.", a, shortcut, training)
        j += 1
    b = conv_bn_silu(state, prefix + "cv2.", x, 1, 1, training)
    return conv_bn_silu(state, prefix + "cv3.", torch.cat((a, b), 1), 1, 1, training)


# --------------------------------------------------------------------------------------------------
# graph remainder: SPPF, Detect
# --------------------------------------------------------------------------------------------------
def sppf(state, prefix, x, k=5, training=False):
    """SPPF.forward (models/common.py:348-366)."""
    x = conv_bn_silu(state, prefix + "cv1.", x, 1, 1, training)
    y1 = F.max_pool2d(x, k, 1, k // 2)
    y2 = F.max_pool2d(y1, k, 1, k // 2)
    y3 = F.max_pool2d(y2, k, 1, k // 2)
    return conv_bn_silu(state, prefix + "cv2.", torch.cat((x, y1, y2, y3), 1), 1, 1, training)


def make_grid(anchors_i, stride_i, nx, ny, na):
    """Detect._make_grid (models/yolo.py:132-153): grid = meshgrid - 0.5; anchor_grid = anchors*stride."""
    ys = torch.arange(ny, dtype=anchors_i.dtype)
    xs = torch.arange(nx, dtype=anchors_i.dtype)
    yv, xv = torch.meshgrid(ys, xs, indexing="ij")
    grid = torch.stack((xv, yv), 2).expand(1, na, ny, nx, 2) - 0.5
    anchor_grid = (anchors_i * stride_i).view(1, na, 1, 1, 2).expand(1, na, ny, nx, 2)
    return grid, anchor_grid


def detect(state, prefix, xs, stride, nc, training=False):
    """Detect.forward (models/yolo.py:84-125).  `state[prefix+'anchors']` is already divided by stride
    (models/yolo.py:291).  train -> list of [bs,na,ny,nx,no]; eval -> (cat(z,1), list)."""
    anchors = state[prefix + "anchors"]
    nl, na = anchors.shape[0], anchors.shape[1]
    no = nc + 5
    outs, z = [], []
    for i in range(nl):
        y = F.conv2d(xs[i], state[f"{prefix}m.{i}.weight"], state[f"{prefix}m.{i}.bias"])
        bs, _, ny, nx = y.shape
        y = y.view(bs, na, no, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
        outs.append(y)
        if not training:
            grid, anchor_grid = make_grid(anchors[i], stride[i], nx, ny, na)
            sg = torch.sigmoid(y)
            xy = (sg[..., 0:2] * 2 + grid) * stride[i]
            wh = (sg[..., 2:4] * 2) ** 2 * anchor_grid
            z.append(torch.cat((xy, wh, sg[..., 4:]), 4).view(bs, na * nx * ny, no))
    return outs if training else (torch.cat(z, 1), outs)


# --------------------------------------------------------------------------------------------------
# P — yaml graph rules
# --------------------------------------------------------------------------------------------------
def make_divisible(x, divisor=8):
    """utils/general.py:669-673: ceil to a multiple of divisor."""
    return math.ceil(x / divisor) * divisor


_CHANNEL_KINDS = {"Conv", "SPPF", "C3_CA", "RFCBAMConv", "BasicStage", "PatchEmbed_FasterNet",
                  "PatchMerging_FasterNet"}


def parse_graph(cfg, ch=3):
    """Restates parse_model (models/yolo.py:397-492) for the module kinds LEAD-YOLO.yaml uses.
    Returns (layers, save): layers[i] = dict(i, f, n, kind, args, c2, repeated)."""
    anchors, nc, gd, gw = cfg["anchors"], cfg["nc"], cfg["depth_multiple"], cfg["width_multiple"]
    na = len(anchors[0]) // 2 if isinstance(anchors, list) else anchors
    no = na * (nc + 5)
    chans, layers, save = [ch], [], []
    c2 = ch
    for i, (f, n, kind, args) in enumerate(cfg["backbone"] + cfg["head"]):
        args = [nc if a == "nc" else anchors if a == "anchors" else (None if a == "None" else a) for a in args]
        n_ = max(round(n * gd), 1) if n > 1 else n          # models/yolo.py:432
        n = n_
        if kind in _CHANNEL_KINDS:
            c1, c2 = chans[f], args[0]
            if c2 != no:
                c2 = make_divisible(c2 * gw, 8)             # models/yolo.py:451
            args = [c1, c2, *args[1:]]
            if kind == "C3_CA":                             # models/yolo.py:454-456
                args.insert(2, n)
                n = 1
            elif kind == "BasicStage":                      # models/yolo.py:457-458
                args.pop(1)
        elif kind == "Concat":
            c2 = sum(chans[x] for x in f)
        elif kind == "Detect":
            args.append([chans[x] for x in f])
        else:                                               # nn.Upsample
            c2 = chans[f]
        layers.append(dict(i=i, f=f, n=n_, kind=kind, args=args, c2=c2, repeated=n))
        save.extend(x % i for x in ([f] if isinstance(f, int) else f) if x != -1)
        if i == 0:
            chans = []
        chans.append(c2)
    return layers, sorted(save)


def model_forward(state, cfg, x, stride, training=False, ch=3, upto=None, collect=None):
    """DetectionModel._forward_once (models/yolo.py:179-195) over the functional layers above."""
    layers, save = parse_graph(cfg, ch)
    ys = []
    for L in layers:
        f = L["f"]
        if f != -1:
            x = ys[f] if isinstance(f, int) else [x if j == -1 else ys[j] for j in f]
        kind, args, i = L["kind"], L["args"], L["i"]

        def run(pfx, x):
            if kind == "PatchEmbed_FasterNet":
                return patch_conv(state, pfx, x, args[2], "proj", training)
            if kind == "PatchMerging_FasterNet":
                return patch_conv(state, pfx, x, args[2], "reduction", training)
            if kind == "BasicStage":
                return basic_stage(state, pfx, x, training)
            if kind == "SPPF":
                return sppf(state, pfx, x, args[2], training)
            if kind == "RFCBAMConv":
                return rfcbam(state, pfx, x, args[2], args[3], training)
            if kind == "C3_CA":
                return c3_ca(state, pfx, x, args[3], training)
            if kind == "Conv":
                ck = args[2] if len(args) > 2 else 1
                cs = args[3] if len(args) > 3 else 1
                return conv_bn_silu(state, pfx, x, ck, cs, training)
            if kind == "nn.Upsample":
                return F.interpolate(x, scale_factor=args[1], mode=args[2])
            if kind == "Concat":
                return torch.cat(x, args[0])
            if kind == "Detect":
                return detect(state, pfx, list(x), stride, args[0], training)
            raise NotImplementedError(kind)

        if L["repeated"] > 1:
            for r in range(L["repeated"]):
                x = run(f"model.{i}.{r}.", x)
        else:
            x = run(f"model.{i}.", x)
        if collect is not None:
            collect[i] = x
        ys.append(x if i in save else None)
        if upto is not None and i == upto:
            return x
    return x


# --------------------------------------------------------------------------------------------------
# L — target assignment and loss
# --------------------------------------------------------------------------------------------------
def build_targets(preds_shapes, targets, anchors, anchor_t=4.0):
    """ComputeLoss.build_targets (utils/loss.py:194-268).  preds_shapes[i] = (bs, na, ny, nx, no);
    targets f32[n,6] = (image, class, x, y, w, h) normalised; anchors [nl, na, 2] in grid units.
    Returns tcls, tbox, indices(b, a, gj, gi as int64), anch."""
    na, nt = anchors.shape[1], targets.shape[0]
    tcls, tbox, indices, anch = [], [], [], []
    gain = torch.ones(7)
    ai = torch.arange(na).float().view(na, 1).repeat(1, nt)
    tg = torch.cat((targets.repeat(na, 1, 1), ai[..., None]), 2)
    g = 0.5
    off = torch.tensor([[0, 0], [1, 0], [0, 1], [-1, 0], [0, -1]]).float() * g
    for i in range(anchors.shape[0]):
        shape = preds_shapes[i]
        gain[2:6] = torch.tensor(shape)[[3, 2, 3, 2]].float()
        t = tg * gain
        if nt:
            r = t[..., 4:6] / anchors[i][:, None]
            keep = torch.max(r, 1 / r).max(2)[0] < anchor_t
            t = t[keep]
            gxy = t[:, 2:4]
            gxi = gain[[2, 3]] - gxy
            j, k = ((gxy % 1 < g) & (gxy > 1)).T
            l, m = ((gxi % 1 < g) & (gxi > 1)).T
            sel = torch.stack((torch.ones_like(j), j, k, l, m))
            t = t.repeat((5, 1, 1))[sel]
            offsets = (torch.zeros_like(gxy)[None] + off[:, None])[sel]
        else:
            t = tg[0]
            offsets = 0
        bc, gxy, gwh, a = t.chunk(4, 1)
        a, (b, c) = a.long().view(-1), bc.long().T
        gij = (gxy - offsets).long()
        gi, gj = gij.T
        indices.append((b, a, gj.clamp(0, shape[2] - 1), gi.clamp(0, shape[3] - 1)))
        tbox.append(torch.cat((gxy - gij, gwh), 1))
        anch.append(anchors[i][a])
        tcls.append(c)
    return tcls, tbox, indices, anch


def eiou(box1, box2, eps=1e-7):
    """bbox_iou(xywh=True, EIoU=True, alpha=1) (utils/metrics.py:293-354); note the double +eps on
    the union (`:323`, `:328`)."""
    (x1, y1, w1, h1), (x2, y2, w2, h2) = box1.chunk(4, -1), box2.chunk(4, -1)
    b1x1, b1x2, b1y1, b1y2 = x1 - w1 / 2, x1 + w1 / 2, y1 - h1 / 2, y1 + h1 / 2
    b2x1, b2x2, b2y1, b2y2 = x2 - w2 / 2, x2 + w2 / 2, y2 - h2 / 2, y2 + h2 / 2
    inter = (b1x2.minimum(b2x2) - b1x1.maximum(b2x1)).clamp(0) * (b1y2.minimum(b2y2) - b1y1.maximum(b2y1)).clamp(0)
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / (union + eps)
    cw = b1x2.maximum(b2x2) - b1x1.minimum(b2x1)
    ch = b1y2.maximum(b2y2) - b1y1.minimum(b2y1)
    c2 = (cw ** 2 + ch ** 2) + eps
    rho2 = ((b2x1 + b2x2 - b1x1 - b1x2) ** 2 + (b2y1 + b2y2 - b1y1 - b1y2) ** 2) / 4
    rho_w2 = ((b2x2 - b2x1) - (b1x2 - b1x1)) ** 2
    rho_h2 = ((b2y2 - b2y1) - (b1y2 - b1y1)) ** 2
    return iou - (rho2 / c2 + rho_w2 / (cw ** 2 + eps) + rho_h2 / (ch ** 2 + eps))


DEFAULT_HYP = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0,
                   label_smoothing=0.0)   # data/hyps/hyp.scratch-low.yaml


def compute_loss(preds, targets, anchors, nc, hyp=None):
    """ComputeLoss.__call__ (utils/loss.py:121-191) with EIoU box loss, gr=1, no focal, no autobalance.
    Returns (loss*bs, [lbox, lobj, lcls])."""
    hyp = dict(DEFAULT_HYP, **(hyp or {}))
    nl = len(preds)
    balance = {3: [4.0, 1.0, 0.4]}.get(nl, [4.0, 1.0, 0.25, 0.06, 0.02])
    lcls, lbox, lobj = torch.zeros(1), torch.zeros(1), torch.zeros(1)
    tcls, tbox, indices, anch = build_targets([p.shape for p in preds], targets, anchors, hyp["anchor_t"])
    cp, cn = 1.0 - 0.5 * hyp["label_smoothing"], 0.5 * hyp["label_smoothing"]
    for i, pi in enumerate(preds):
        b, a, gj, gi = indices[i]
        tobj = torch.zeros(pi.shape[:4], dtype=pi.dtype)
        n = b.shape[0]
        if n:
            sel = pi[b, a, gj, gi]
            pxy = sel[:, 0:2].sigmoid() * 2 - 0.5
            pwh = (sel[:, 2:4].sigmoid() * 2) ** 2 * anch[i]
            iou = eiou(torch.cat((pxy, pwh), 1), tbox[i]).squeeze()
            lbox = lbox + (1.0 - iou).mean()
            tobj[b, a, gj, gi] = iou.detach().clamp(0).type(tobj.dtype)
            if nc > 1:
                pcls = sel[:, 5:]
                t = torch.full_like(pcls, cn)
                t[range(n), tcls[i]] = cp
                lcls = lcls + F.binary_cross_entropy_with_logits(pcls, t, pos_weight=torch.tensor([hyp["cls_pw"]]))
        obji = F.binary_cross_entropy_with_logits(pi[..., 4], tobj, pos_weight=torch.tensor([hyp["obj_pw"]]))
        lobj = lobj + obji * balance[i]
    lbox = lbox * hyp["box"]
    lobj = lobj * hyp["obj"]
    lcls = lcls * hyp["cls"]
    bs = preds[0].shape[0]
    return (lbox + lobj + lcls) * bs, torch.cat((lbox, lobj, lcls)).detach()


# --------------------------------------------------------------------------------------------------
# T — optimiser step
# --------------------------------------------------------------------------------------------------
def param_groups(state_keys_shapes):
    """smart_optimizer grouping (utils/torch_utils.py:318-346): g2 = every `.bias`; g1 = BatchNorm
    weights (1-D `.weight` of a module that also has running stats); g0 = remaining `.weight`s (decay)."""
    keys = set(state_keys_shapes)
    groups = {"decay": [], "bn": [], "bias": []}
    for k in state_keys_shapes:
        if k.endswith(".bias"):
            groups["bias"].append(k)
        elif k.endswith(".weight"):
            if k[:-len("weight")] + "running_mean" in keys:
                groups["bn"].append(k)
            else:
                groups["decay"].append(k)
    return groups


def sgd_nesterov_step(params, grads, bufs, lr, momentum=0.937, weight_decay=0.0):
    """torch.optim.SGD(nesterov=True) single-tensor update, in place.  bufs[k] is None on first step."""
    for k, p in params.items():
        g = grads[k]
        if weight_decay:
            g = g + weight_decay * p
        if bufs.get(k) is None:
            bufs[k] = g.clone()
        else:
            bufs[k].mul_(momentum).add_(g)
        p.sub_(lr * (g + momentum * bufs[k]))
