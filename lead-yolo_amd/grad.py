"""Training-step backward of the HIP path: torch.autograd.Function wrappers whose
forward is the same HIP forward the inference path runs and whose backward is built from the HIP
backward blocks in csrc/ly_backward.hip + the forward contraction kernels with transposed weights.

What autograd would derive for the reference's modules (train.py:324 `scaler.scale(loss).backward()`):
    Conv2d -> BatchNorm2d(train) -> SiLU/ReLU          models/common.py:1890-1910, 1537-1561
    MLPBlock (pconv -> 1x1 -> BN -> ReLU -> 1x1, +x)   models/common.py:1432-1437, 1478-1482

Scheme for every conv -> BN -> act unit (v = a*u + b with the BATCH statistics, y = act(v)):
    1. recompute u with the forward kernel (identity epilogue)  — nothing but the unit's input was saved
    2. ly_bnact_bwd_reduce:  s1 = sum dv, s2 = sum dv*u  (dv = dy*act'(v))  ->  dgamma, dbeta
    3. ly_bnact_bwd_apply :  du = a*(dv - mean(dv) - xhat*mean(dv*xhat))  written as alpha*dv + kappa + lambda*u
    4. dgrad: forward kernel on du with transposed weights;  wgrad: ly_wgrad (pixel contraction)
Only [C]-sized vectors are handled with torch ops (coefficients of step 3, parameter-gradient reshapes).
"""
import ctypes

import os

import torch

from . import ops, pack
from .ops import ACT_NONE, ACT_RELU, ACT_SILU  # noqa: F401


def _rows_dense(t):
    """NHWC-dense view of a logical [n, c, h, w] tensor (copy only if needed)."""
    t = ops.nhwc(t)
    if t.stride(1) != 1:
        t = t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    return t




# ---- tensors with two consumers: the sum of the two gradients as a store ----------------------------------------------------------------
# The skip connections of models/LEAD-YOLO.yaml (backbone stage -> next stage + neck concat; neck map -> upsample + later concat; neck output
# -> Detect level + next RFCBAMConv) hand one tensor to two consumers; autograd sums the two input gradients with one more
# read-read-write pass per tensor (six per step, 137 us at bs = 64).  `fork` gives each consumer its own alias; a consumer whose data
# gradient is a GEMM does not launch it in its own backward but leaves it with the tensor's slot, and Fork.backward — which runs when BOTH
# consumers are done — launches it with the other consumer's gradient as LyGemmParams.eadd: the sum is the GEMM's store.
FORK_SUM = True                 # a module constant (tests monkeypatch it)


class _Slot:
    __slots__ = ("deferred",)

    def __init__(self):
        self.deferred = []


class Fork(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, slot):
        ctx.slot = slot
        ctx.set_materialize_grads(False)
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, d1, d2):
        total = d1 if d2 is None else d2 if d1 is None else d1 + d2
        deferred, ctx.slot.deferred = ctx.slot.deferred, []
        with torch.no_grad():
            for produce in deferred:
                total = produce(total)
        return total, None


def fork(x):
    """(x, x) as two aliases whose gradients meet in Fork.backward; each carries the slot a GEMM-producing consumer defers its launch to"""
    slot = _Slot()
    a, b = Fork.apply(x, slot)
    a._ly_slot = b._ly_slot = slot
    return a, b


def _slot_of(x):
    return getattr(x, "_ly_slot", None) if (FORK_SUM and x is not None) else None


def _gemm_dx(slot, n, c, h, w, like, **kw):
    """dx [n, c, h, w] = ops.gemm(out=dx, **kw) — now, or (slot) when the tensor's other gradient exists, added to it in the store"""
    def produce(eadd):
        out = ops.empty_nhwc(n, c, h, w, like)
        extra, late = {}, None
        if eadd is not None:
            if eadd.dtype == out.dtype and tuple(eadd.shape) == tuple(out.shape) and kw["N"] % 4 == 0:
                e, lde = ops.rows(eadd)
                extra = dict(eadd=e, ldeadd=lde)
            else:
                late = eadd
        ops.gemm(out=out, **kw, **extra)
        return out if late is None else out + late.to(out.dtype)
    if slot is None:
        return produce(None)
    slot.deferred.append(produce)
    return None


class ConvSpec:
    """Static description of one conv(+bias) -> [BN] -> act unit."""

    def __init__(self, kind, cout, act=ACT_NONE, bn=None, bn_train=False, k=1, nchw=False, up=False, out_dtype=None):
        self.kind = kind            # "pw" (1x1, one or two row sources), "c3" (3x3 s1 p1), "patch" (k = s)
        self.cout, self.act, self.bn, self.bn_train = cout, act, bn, bn_train
        self.k, self.nchw, self.up = k, nchw, up
        self.out_dtype = out_dtype  # storage dtype of the output when it differs from the input's (fp32 NCHW image -> bf16 map)




def _conv_forward(spec, x0, x1, wp, e_scale, e_shift, act, stats=None, out=None, store=False):
    """Runs the forward contraction of `spec`; returns the output tensor (None for a pure statistics pass; with store=True a
    statistics pass also stores its pre-BN value, in the same launch)."""
    co = spec.cout
    if spec.kind == "pw":
        t0, ld0 = ops.rows(x0)
        n, c0, h, w = t0.shape
        if spec.up:
            h, w = 2 * h, 2 * w
        kw = dict(a0=t0, lda0=ld0, k0=c0, gather=ops.GATHER_UP2 if spec.up else ops.GATHER_ROWS)
        k = c0
        if x1 is not None:
            t1, ld1 = ops.rows(x1)
            kw.update(a1=t1, lda1=ld1)
            k += t1.shape[1]
        if (stats is None or store) and out is None:
            out = ops.empty_nhwc(n, co, h, w, t0)
        ops.gemm(M=n * h * w, H=h, W=w, K=k, N=co, wp=wp, out=out, ldo=co, e_scale=e_scale, e_shift=e_shift, act=act, stats=stats, **kw)
        return out
    if spec.kind == "c3":
        t0, ld0 = ops.rows(x0)
        n, c, h, w = t0.shape
        if (stats is None or store) and out is None:
            out = ops.empty_nhwc(n, co, h, w, t0)
        ops.conv3x3(M=n * h * w, H=h, W=w, Cin=c, N=co, x=t0, ldx=ld0, wp=wp, out=out, ldo=co, e_scale=e_scale, e_shift=e_shift, act=act,
                    stats=stats)
        return out
    n, c, h, w = x0.shape
    k = spec.k
    ho, wo = h // k, w // k
    if spec.nchw:
        xr = x0.contiguous()
        kw = dict(K=16 * c, a0=xr, lda0=0, k0=16 * c, gather=ops.GATHER_PATCH_NCHW, Hin=h, Win=w, Cin=c, ks=4, pk=0)
    else:
        xr = _rows_dense(x0)
        kw = dict(K=k * k * c, a0=xr, lda0=c, k0=k * k * c, gather=ops.GATHER_PATCH, Hin=h, Win=w, Cin=c, ks=k, pk=k * c)
    odt = spec.out_dtype or xr.dtype
    if (stats is None or store) and out is None:
        out = ops.empty_nhwc(n, co, ho, wo, xr, dtype=odt)
    ops.gemm(M=n * ho * wo, H=ho, W=wo, N=co, wp=wp, out=out, ldo=co, e_scale=e_scale, e_shift=e_shift, act=act, stats=stats, dtype=odt, **kw)
    return out




def affine_backward(dy, u, a, b, act, mean, invstd, train, inplace=True, gamma=None, beta=None, lddy=None, out=None):
    """du and (dgamma, dbeta) for v = a*u + b, y = act(v); dy/u are NHWC-dense [n, c, h, w].  du is written over u unless
    inplace=False (u is a tensor saved for backward) or into `out` (e.g. dy itself).  gamma / beta: the BatchNorm parameters — when a gradient sink holds their
    storage (ops.GradSink) dgamma / dbeta are added there by the coefficient kernel and returned as None."""
    n, c, h, w = u.shape
    rows = n * h * w
    lddy = c if lddy is None else lddy          # dy may be a channel slice of a wider gradient (row stride lddy), read in place
    sums = ops.bnact_bwd_reduce(dy, lddy, u, c, rows, c, a, b, act)
    tg, tb = ops.grad_target(gamma), ops.grad_target(beta)
    direct = tg is not None and tb is not None and tg.numel() == c and tb.numel() == c
    dgamma, dbeta, alpha, kappa, lam = ops.bn_bwd_coeffs(sums, c, rows, a, mean, invstd, train, dgamma=tg if direct else None,
                                                         dbeta=tb if direct else None)
    if direct:
        ops.grad_done(gamma)
        ops.grad_done(beta)
    du = out if out is not None else u if inplace else torch.empty_like(u)     # out: e.g. dy itself (elementwise: same index read and written)
    ops.bnact_bwd_apply(dy, lddy, u, c, rows, c, a, b, act, alpha, kappa, lam, du, c)
    return du, dgamma, dbeta


def conv_wgrad(spec, du, x0, x1, weight):
    """Weight gradient, written in the weight's OWN layout [cout][cin][kh][kw] (ly_wgrad's dw_ts / dw_cs): straight into the
    parameter's persistent gradient storage when a sink holds it (returns None: autograd has nothing to accumulate), else into a
    fresh zeroed tensor of the weight's shape.  du: NHWC-dense [n, cout, ho, wo]; cout may be weight.shape[0] zero-padded to a
    multiple of the vector width (Detect heads: 18 -> 20 / 24) — rows beyond the weight's are dropped (n_valid)."""
    n, co, ho, wo = du.shape
    m = n * ho * wo
    nv = weight.shape[0]
    tgt = ops.grad_target(weight)
    dw = tgt if tgt is not None else torch.zeros(weight.shape, dtype=torch.float32, device=du.device)
    if spec.kind == "pw":
        t0, ld0 = ops.rows(x0)
        c0 = t0.shape[1]
        k = c0 + (x1.shape[1] if x1 is not None else 0)
        probs = [dict(M=m, H=ho, W=wo, N=co, du=du, lddu=co, x=t0, ldx=ld0, Hin=t0.shape[2], Win=t0.shape[3], Cin=c0, dw=dw, lddw=k,
                      up2=spec.up, n_valid=nv)]
        if x1 is not None:
            t1, ld1 = ops.rows(x1)
            probs.append(dict(M=m, H=ho, W=wo, N=co, du=du, lddu=co, x=t1, ldx=ld1, Hin=ho, Win=wo, Cin=t1.shape[1], dw=dw, lddw=k, dw_off=c0,
                              n_valid=nv))
        ops.wgrad_group(probs)               # two row sources (a concat read in place): one launch when both are plain rows
    elif spec.kind == "c3":
        t0, ld0 = ops.rows(x0)
        c = t0.shape[1]
        # tap-major gradient storage (optim.FusedSGD, [cout][tap][cin]): packed rows, contiguous atomics; the weight's own layout
        # [cout][cin][tap] otherwise (lanes 36 bytes apart: ~3x slower atomics)
        ts, cs = (c, 1) if _tap_major(dw) else (1, 9)
        ops.wgrad(M=m, H=ho, W=wo, N=co, du=du, lddu=co, x=t0, ldx=ld0, Hin=ho, Win=wo, Cin=c, dw=dw, lddw=9 * c, ks=3, stride=1, pad=1,
                  dw_ts=ts, dw_cs=cs, n_valid=nv)
    else:
        _, c, h, w = x0.shape
        k = spec.k
        if spec.nchw and h == ho * k and w == wo * k and (k * k * c) % 4 == 0 and co % 4 == 0:
            # PatchEmbed on the NCHW image: one space-to-depth copy turns the k x k patch gather into plain rows [M, c*k*k] (the
            # weight's own (c, ky, kx) column order), which the tiled kernel takes; the per-lane NCHW gather kernel was 3x slower.
            # A uint8 image (pixel / 255 folded into the forward gather) is contracted as the integers it holds — exact in bf16 —
            # and the 1/255 is applied to the small weight gradient instead of to the batch.
            u8 = x0.dtype == torch.uint8
            if u8 and k == 4 and co == nv and dw.is_contiguous() and ops.patch4_wgrad_u8_ok(x0, du, co):
                # round 6: straight from the image — no space-to-depth rows (157 MB written and read back at bs = 64), no scratch, no add
                ops.patch4_wgrad_u8(x0, du, co, co, dw, 1.0 / 255.0)
                if tgt is not None:
                    ops.grad_done(weight)
                    return None
                return dw
            if u8 and k == 4 and x0.is_contiguous():
                xr = ops.patch4_rows_u8(x0, du.dtype)                # one pass (was a permuted copy + a cast)
            else:
                xr = x0.reshape(n, c, ho, k, wo, k).permute(0, 2, 4, 1, 3, 5).reshape(m, c * k * k).to(du.dtype)
            acc = ops.zeros_f32(weight.numel(), du.device).view(weight.shape) if u8 else dw
            ops.wgrad(M=m, H=ho, W=wo, N=co, du=du, lddu=co, x=xr, ldx=k * k * c, Hin=ho, Win=wo, Cin=k * k * c, dw=acc, lddw=k * k * c, n_valid=nv)
            if u8:
                dw.add_(acc, alpha=1.0 / 255.0)
        elif spec.nchw:
            if x0.dtype == torch.uint8:
                raise NotImplementedError("uint8 image: the weight gradient of the patch embedding needs H, W multiples of the patch size")
            xr = x0.contiguous().to(du.dtype)
            ops.wgrad(M=m, H=ho, W=wo, N=co, du=du, lddu=co, x=xr, ldx=0, Hin=h, Win=w, Cin=c, dw=dw, lddw=k * k * c, ks=k, stride=k, nchw=True,
                      dw_ts=1, dw_cs=k * k, n_valid=nv)
        else:
            xr = _rows_dense(x0)
            ts, cs = (c, 1) if _tap_major(dw) else (1, k * k)
            ops.wgrad(M=m, H=ho, W=wo, N=co, du=du, lddu=co, x=xr, ldx=c, Hin=h, Win=w, Cin=c, dw=dw, lddw=k * k * c, ks=k, stride=k,
                      dw_ts=ts, dw_cs=cs, n_valid=nv)
    if tgt is not None:
        ops.grad_done(weight)
        return None
    return dw


def _tap_major_rows(g):
    """[co, kh*kw*ci] row view of the [co][kh][kw][ci] storage behind a tap-major gradient target (see _tap_major), else None"""
    if g is None or not _tap_major(g):
        return None
    co, ci, kh, kw = g.shape
    rows = g.permute(0, 2, 3, 1)
    if not rows.is_contiguous():                 # (reshape would hand out a copy: the gradient written into it would be lost)
        return None
    return rows.view(co, kh * kw * ci)


def _tap_major(g):
    """gradient tensor of a [co, ci, kh, kw] weight whose storage is [co][kh][kw][ci] (see optim.FusedSGD)"""
    return g.dim() == 4 and not g.is_contiguous() and g.stride(1) == 1 and g.stride(3) == g.shape[1]


def conv_dgrad(spec, du, weight, x0, x1, need0, need1, slots=(None, None)):
    """Input gradients (dx0, dx1) of the contraction; du NHWC-dense [n, cout, ho, wo] with cout % 4 == 0 columns valid.  slots: the
    Fork slot of x0 (see `fork`): the single-source 1x1 and the patch data gradients then wait for the other consumer's gradient."""
    n, co, ho, wo = du.shape
    m = n * ho * wo
    with torch.no_grad():
        if spec.kind == "pw":
            nw, kin = weight.shape[0], weight.numel() // weight.shape[0]
            # W^T [kin, co] read in place from the parameter (columns zero-padded to du's channel count)
            wt = pack.packed(pack.src_matrix(weight, kin, nw, sr=1, sk=kin, k_pad=co), co, ops.planes_of(du))
            if x1 is None and not spec.up and slots[0] is not None and kin % 4 == 0:
                return _gemm_dx(slots[0], n, kin, ho, wo, du, M=m, H=ho, W=wo, K=co, N=kin, a0=du, lda0=co, k0=co, wp=wt, ldo=kin), None
            d = ops.empty_nhwc(n, kin, ho, wo, du)
            ops.gemm(M=m, H=ho, W=wo, K=co, N=kin, a0=du, lda0=co, k0=co, wp=wt, out=d, ldo=kin)
            if x1 is None:
                return (ops.up2_bwd(d, kin, n, ho // 2, wo // 2, kin) if spec.up else d), None
            c0 = x0.shape[1]
            d0 = d[:, :c0]
            if spec.up and need0:
                d0 = ops.up2_bwd(d, kin, n, ho // 2, wo // 2, c0)
            return (d0 if need0 else None), (d[:, c0:] if need1 else None)
        if spec.kind == "c3":
            cop = (weight.shape[0] + 31) // 32 * 32
            wt = pack.packed(pack.src_taps(weight, cop, transposed_flipped=True), 9 * cop, ops.planes_of(du))
            cin = weight.shape[1]
            d = ops.empty_nhwc(n, cin, ho, wo, du)
            ops.conv3x3(M=m, H=ho, W=wo, Cin=co, N=cin, x=du, ldx=co, wp=wt, out=d, ldo=cin)
            return d, None
        if spec.nchw:
            raise NotImplementedError("gradient with respect to the NCHW input image is not built (never needed: it is the data)")
        _, c, h, w = x0.shape
        k = spec.k
        # rows (ky, kx, c), columns co: the transposed patch matrix, read in place
        wt = pack.packed(pack.Src(weight, k * k * c, nrb=c, sra=1, srb=k * k, nc=co, sc=c * k * k), co, ops.planes_of(du))
        if h == ho * k and w == wo * k and c % 4 == 0:
            # round 6: the adjoint of the patch gather is the GEMM's store (LyGemmParams.scat_ks) — no [M][k*k*c] intermediate, no ly_unpatch pass
            return _gemm_dx(slots[0], n, c, h, w, du, M=m, H=ho, W=wo, K=co, N=k * k * c, a0=du, lda0=co, k0=co, wp=wt, ldo=c, scat_ks=k,
                            scat_c=c), None
        g = torch.empty((m, k * k * c), dtype=du.dtype, device=du.device)
        ops.gemm(M=m, H=ho, W=wo, K=co, N=k * k * c, a0=du, lda0=co, k0=co, wp=wt, out=g, ldo=k * k * c)
        return ops.unpatch(g, n, ho, wo, c, k, h, w), None


def _pad_cols(w2d, k):
    """[R, k0] -> [R, k] zero padded on the right (k >= k0)."""
    if w2d.shape[1] == k:
        return w2d
    out = torch.zeros(w2d.shape[0], k, dtype=w2d.dtype, device=w2d.device)
    out[:, :w2d.shape[1]] = w2d
    return out


class ConvBnAct(torch.autograd.Function):
    """y = act(BN(conv(x0 [| x1]) + bias)).  tensors: x0, x1|None, weight, bias|None, gamma|None, beta|None."""

    @staticmethod
    def forward(ctx, spec, wp, x0, x1, weight, bias, gamma, beta):
        co = spec.cout
        dev = x0.device
        ctx.slots = (_slot_of(x0), _slot_of(x1))
        bias_f = bias.detach().float().contiguous() if bias is not None else None
        mean = invstd = None
        if spec.bn is not None:
            if spec.bn_train and co % 4 == 0:
                # ONE contraction: statistics and the pre-BN value u in the same launch; y = act(a*u + b) is an elementwise
                # pass, and u is kept for the backward (no recompute there)
                stats = ops.new_stats(co, dev)
                u = _conv_forward(spec, x0, x1, wp, None, bias_f, ACT_NONE, stats=stats, store=True)
                rows = u.shape[0] * u.shape[2] * u.shape[3]
                a, b, mean, invstd = ops.bn_finalize(spec.bn, stats, co, rows, want_stats=True)
                y = torch.empty_like(u)
                ops.bnact_fwd(u, co, rows, co, a, b, spec.act, y, co)
                ctx.spec, ctx.wp = spec, wp
                ctx.has = (x1 is not None, bias is not None)
                ctx.params = (weight, gamma, beta)
                ctx.save_for_backward(x0, x1, weight, bias_f, a, b, mean, invstd, u)
                return y
            elif spec.bn_train:
                stats = ops.new_stats(co, dev)
                _conv_forward(spec, x0, x1, wp, None, bias_f, ACT_NONE, stats=stats)
                y0 = _out_shape(spec, x0)
                a, b, mean, invstd = ops.bn_finalize(spec.bn, stats, co, y0[0] * y0[2] * y0[3], want_stats=True)
            else:
                bn = spec.bn
                invstd = torch.rsqrt(bn.running_var.detach().float() + bn.eps)
                mean = bn.running_mean.detach().float()
                a = (gamma.detach().float() * invstd).contiguous()
                b = (beta.detach().float() - mean * a).contiguous()
            sh = b if bias_f is None else (b + bias_f * a).contiguous()
            y = _conv_forward(spec, x0, x1, wp, a, sh, spec.act)
        else:
            a = b = None
            y = _conv_forward(spec, x0, x1, wp, None, bias_f, spec.act)
        ctx.spec, ctx.wp = spec, wp
        ctx.has = (x1 is not None, bias is not None)
        ctx.params = (weight, gamma, beta)
        ctx.save_for_backward(x0, x1, weight, bias_f, a, b, mean, invstd, None)
        return y

    @staticmethod
    def backward(ctx, dy):
        spec, wp = ctx.spec, ctx.wp
        x0, x1, weight, bias_f, a, b, mean, invstd, u_saved = ctx.saved_tensors
        w_param, g_param, b_param = ctx.params          # the Parameter objects themselves (gradient-sink lookup)
        co = spec.cout
        need = ctx.needs_input_grad          # (spec, wp, x0, x1, weight, bias, gamma, beta)
        with torch.no_grad():
            odt = spec.out_dtype or x0.dtype
            dy = dy if dy.dtype == odt else dy.to(odt)
            lddy = None
            if spec.bn is not None or spec.act != ACT_NONE:
                dy, lddy = ops.rows(dy)              # the BN / activation backward reads a channel slice of a wider gradient in place
            else:
                dy = _rows_dense(dy)
            dgamma = dbeta = dbias = None
            if spec.bn is not None or spec.act != ACT_NONE:
                # pre-BN value: kept by the forward (train-mode BN) or recomputed (conv + bias)
                u = u_saved if u_saved is not None else _conv_forward(spec, x0, x1, wp, None, bias_f, ACT_NONE)
                if spec.bn is None:
                    a = torch.ones(co, dtype=torch.float32, device=dy.device)
                    b = torch.zeros_like(a)
                    mean, invstd = b, a
                du, dgamma, dbeta = affine_backward(dy, u, a, b, spec.act, mean, invstd, spec.bn is not None and spec.bn_train,
                                                    inplace=u_saved is None, gamma=g_param if spec.bn is not None else None,
                                                    beta=b_param if spec.bn is not None else None, lddy=lddy)
                if bias_f is not None:
                    if spec.bn is None:
                        dbias, dgamma, dbeta = dbeta, None, None
                    elif spec.bn_train:
                        dbias = torch.zeros_like(bias_f)                               # BN removes the batch mean: d/dbias = 0
                    else:
                        dbias = a * (dbeta if dbeta is not None else b_param.grad)
                elif spec.bn is None:
                    dgamma = dbeta = None
            else:
                du = dy
                if bias_f is not None:
                    n, _, h, w = dy.shape
                    dbias = _chan_sum(dy, co)
            vw = ops.vw_of(du)
            cq = (co + vw - 1) // vw * vw
            if cq != co:                                                               # e.g. Detect heads (18 channels)
                n, _, h, w = du.shape
                pad = torch.zeros((n, cq, h, w), dtype=du.dtype, device=du.device).contiguous(memory_format=torch.channels_last)
                pad[:, :co] = du
                du_d = pad
            else:
                du_d = du
            dw = conv_wgrad(spec, du_d if spec.kind == "pw" else du, x0, x1, w_param) if need[4] else None
            dx0 = dx1 = None
            if need[2] or need[3]:
                dspec = spec
                if cq != co:
                    dspec = ConvSpec(spec.kind, cq, k=spec.k, nchw=spec.nchw, up=spec.up)
                dx0, dx1 = conv_dgrad(dspec, du_d, weight, x0, x1, need[2], need[3], ctx.slots)
        return None, None, dx0, dx1, dw, dbias, dgamma, dbeta


def _chan_sum(t, c):
    """per-channel sum over pixels of an NHWC-dense tensor (c may be any size) -> fp32."""
    n, _, h, w = t.shape
    if c % 4 == 0 and c <= 1024:
        return ops.chan_moments(t, c, n * h * w, c)[:c]
    return t.sum((0, 2, 3), dtype=torch.float32)


def _out_shape(spec, x0):
    n, c, h, w = x0.shape
    if spec.kind == "patch":
        return n, spec.cout, h // spec.k, w // spec.k
    if spec.up:
        return n, spec.cout, 2 * h, 2 * w
    return n, spec.cout, h, w


def conv_bn_act(spec, wp, x0, x1, weight, bias, bn):
    gamma = bn.weight if bn is not None else None
    beta = bn.bias if bn is not None else None
    return ConvBnAct.apply(spec, wp, x0, x1, weight, bias, gamma, beta)


class ConvBnActPair(torch.autograd.Function):
    """Two 1x1 conv -> BatchNorm(train) -> act units over the SAME input (C3_CA's cv1 and cv2, models/common.py:1630-1636): one contraction
    with the stacked weights writes [u1 | u2] and its batch statistics, one elementwise pass applies both BatchNorms; the two outputs
    are the channel halves of one buffer.  Backward: each half's BN / activation backward writes its half of one du, the data gradient
    is ONE contraction with [W1^T | W2^T] — as two separate nodes the input was read twice, its gradient
    produced twice (twice through the 2x-upsample adjoint when the input is lazily upsampled) and summed by autograd with one more pass
    per source."""

    @staticmethod
    def forward(ctx, spec, wp, bn1, bn2, x0, x1, w1, w2, g1, be1, g2, be2):
        co = spec.cout
        c_ = co // 2
        stats = ops.new_stats(co, x0.device)
        u = _conv_forward(spec, x0, x1, wp, None, None, ACT_NONE, stats=stats, store=True)
        rows = u.shape[0] * u.shape[2] * u.shape[3]
        v = torch.empty(4, co, dtype=torch.float32, device=x0.device)          # scale, shift, mean, invstd of both halves
        ops.bn_finalize_pair(bn1, bn2, stats, c_, rows, v)                      # one launch for the two BatchNorms
        y = torch.empty_like(u)
        ops.bnact_fwd(u, co, rows, co, v[0], v[1], spec.act, y, co)
        ctx.spec = spec
        ctx.wt_src = pack.src_matrix_kcat_t(w1, w2)
        ctx.slots = (_slot_of(x0), _slot_of(x1))
        # one data-gradient contraction per source (instead of one over both) when the sources' gradients differ in resolution (a lazily
        # upsampled x0) or in destination (a source shared with another consumer: its gradient is added to that one's in the store, `fork`)
        ctx.split = spec.up or (x1 is not None and (ctx.slots[0] is not None or ctx.slots[1] is not None))
        if ctx.split:
            c0 = x0.shape[1]
            ctx.wt_src = (pack.src_matrix_kcat_t(w1, w2, 0, c0), pack.src_matrix_kcat_t(w1, w2, c0, None) if x1 is not None else None)
        ctx.params = ((w1, g1, be1), (w2, g2, be2))
        ctx.save_for_backward(x0, x1, w1, w2, v, u)
        return y[:, :c_], y[:, c_:]

    @staticmethod
    def backward(ctx, dy1, dy2):
        spec = ctx.spec
        x0, x1, w1, w2, v, u = ctx.saved_tensors
        co = spec.cout
        c_ = co // 2
        n, _, ho, wo = u.shape
        rows = n * ho * wo
        need = ctx.needs_input_grad          # (spec, wp, bn1, bn2, x0, x1, w1, w2, g1, be1, g2, be2)
        out = [None] * 12
        with torch.no_grad():
            du = torch.empty_like(u)
            t0, ld0 = ops.rows(x0)
            c0 = t0.shape[1]
            t1, ld1 = ops.rows(x1) if x1 is not None else (None, 0)
            kin = c0 + (t1.shape[1] if t1 is not None else 0)
            # the two weight gradients as ONE stacked [2c_, kin] matrix when the sink keeps them adjacent (optim.FusedSGD does)
            tw = [ops.grad_target(ctx.params[i][0]) for i in range(2)]
            stacked = (need[6] and need[7] and tw[0] is not None and tw[1] is not None and tw[0].is_contiguous() and tw[1].is_contiguous()
                       and tw[1].data_ptr() == tw[0].data_ptr() + 4 * tw[0].numel())
            dys = []
            for dy in (dy1, dy2):
                if dy is None:
                    dy = torch.zeros((n, c_, ho, wo), dtype=u.dtype, device=u.device).contiguous(memory_format=torch.channels_last)
                dys.append(ops.rows(dy if dy.dtype == u.dtype else dy.to(u.dtype)))
            # both units' BatchNorm / activation backward in ONE reduce pass and ONE apply pass over the stacked tensor (were two half-width
            # launches each): the coefficient launches stay per unit (their targets are two parameters' gradient storages)
            (dya, lda), (dyb, ldb) = dys
            sums2 = ops.bnact_bwd_reduce_pair(dya, lda, dyb, ldb, c_, u, co, rows, co, v[0], v[1], spec.act)
            coef = torch.empty(3, co, dtype=torch.float32, device=u.device)
            # the coefficient kernel of both units in ONE launch; the gradients of gamma / beta are added into the sink's storage where
            # there is one, else into fresh zeros handed to autograd
            tgt, fresh = [], None
            for i in range(2):
                w_p, g_p, b_p = ctx.params[i]
                tg, tb = ops.grad_target(g_p), ops.grad_target(b_p)
                if tg is not None and tb is not None and tg.is_contiguous() and tb.is_contiguous():
                    tgt.append((tg, tb))
                else:
                    fresh = torch.zeros(2, 2, c_, dtype=torch.float32, device=u.device) if fresh is None else fresh
                    tgt.append((fresh[i, 0], fresh[i, 1]))
                    out[8 + 2 * i], out[9 + 2 * i] = fresh[i, 0], fresh[i, 1]
            ops.bn_bwd_coeffs_pair(sums2[0], sums2[1], c_, rows, v, tgt, coef)
            for i in range(2):
                if out[8 + 2 * i] is None:
                    ops.grad_done(ctx.params[i][1])
                    ops.grad_done(ctx.params[i][2])
            ops.bnact_bwd_apply_pair(dya, lda, dyb, ldb, c_, u, co, rows, co, v[0], v[1], spec.act, coef[0], coef[1], coef[2], du, co)
            # A lazily 2x-upsampled source: the adjoint of the upsample (sum over each 2x2 block) commutes with the 1x1 convolution, so it is
            # applied ONCE to du (co channels) and that source's weight and data gradients are plain contractions at a quarter of the rows —
            # not an upsample-gathering weight gradient at full resolution plus a full-resolution data gradient folded afterwards.
            hq, wq, rq = ho // 2, wo // 2, rows // 4
            dup = ops.up2_bwd(du, co, n, hq, wq, co) if spec.up else None
            p0 = (dict(M=rq, H=hq, W=wq, du=dup, lddu=co, x=t0, ldx=ld0, Hin=hq, Win=wq, Cin=c0, lddw=kin) if spec.up else
                  dict(M=rows, H=ho, W=wo, du=du, lddu=co, x=t0, ldx=ld0, Hin=t0.shape[2], Win=t0.shape[3], Cin=c0, lddw=kin))
            for i in range(2):
                off = i * c_
                w_p = ctx.params[i][0]
                if need[6 + i] and not stacked:
                    tgt = tw[i]
                    dw = tgt if tgt is not None else torch.zeros(w_p.shape, dtype=torch.float32, device=u.device)
                    ops.wgrad(N=c_, du_off=off, dw=dw, **p0)
                    if t1 is not None:
                        ops.wgrad(M=rows, H=ho, W=wo, N=c_, du=du, lddu=co, du_off=off, x=t1, ldx=ld1, Hin=ho, Win=wo, Cin=t1.shape[1], dw=dw,
                                  lddw=kin, dw_off=c0)
                    if tgt is not None:
                        ops.grad_done(w_p)
                    else:
                        out[6 + i] = dw
            if stacked:
                probs = [dict(N=co, dw=tw[0], **p0)]
                if t1 is not None:
                    probs.append(dict(M=rows, H=ho, W=wo, N=co, du=du, lddu=co, x=t1, ldx=ld1, Hin=ho, Win=wo, Cin=t1.shape[1], dw=tw[0], lddw=kin,
                                      dw_off=c0))
                ops.wgrad_group(probs)
                ops.grad_done(ctx.params[0][0])
                ops.grad_done(ctx.params[1][0])
            s0, s1 = ctx.slots
            if ctx.split:
                pl = ops.planes_of(du)
                if need[4]:
                    a, m0, h0, w0 = (dup, rq, hq, wq) if spec.up else (du, rows, ho, wo)
                    out[4] = _gemm_dx(s0 if c0 % 4 == 0 else None, n, c0, h0, w0, du, M=m0, H=h0, W=w0, K=co, N=c0, a0=a, lda0=co, k0=co,
                                      wp=pack.packed(ctx.wt_src[0], co, pl), ldo=c0)
                if need[5] and x1 is not None:
                    c1 = kin - c0
                    out[5] = _gemm_dx(s1 if c1 % 4 == 0 else None, n, c1, ho, wo, du, M=rows, H=ho, W=wo, K=co, N=c1, a0=du, lda0=co, k0=co,
                                      wp=pack.packed(ctx.wt_src[1], co, pl), ldo=c1)
            elif x1 is None and s0 is not None and need[4] and kin % 4 == 0:
                out[4] = _gemm_dx(s0, n, kin, ho, wo, du, M=rows, H=ho, W=wo, K=co, N=kin, a0=du, lda0=co, k0=co,
                                  wp=pack.packed(ctx.wt_src, co, ops.planes_of(du)), ldo=kin)
            elif need[4] or need[5]:
                pl = ops.planes_of(du)
                d = ops.empty_nhwc(n, kin, ho, wo, du)
                # [W1^T | W2^T] ([kin, 2c_]) read in place from the two parameters: ONE contraction over both halves of du
                wt = pack.packed(ctx.wt_src, co, pl)
                ops.gemm(M=rows, H=ho, W=wo, K=co, N=kin, a0=du, lda0=co, k0=co, wp=wt, out=d, ldo=kin)
                if x1 is None:
                    out[4] = d
                else:
                    if need[4]:
                        out[4] = d[:, :c0]
                    if need[5]:
                        out[5] = d[:, c0:]
        return tuple(out)


def conv_bn_act_pair(act, up, wp, x0, x1, conv1, bn1, conv2, bn2):
    """cv1(x), cv2(x) of two Conv modules sharing their input, as one node (see ConvBnActPair)"""
    spec = ConvSpec("pw", conv1.weight.shape[0] + conv2.weight.shape[0], act, bn1, True, up=up)
    return ConvBnActPair.apply(spec, wp, bn1, bn2, x0, x1, conv1.weight, conv2.weight, bn1.weight, bn1.bias, bn2.weight, bn2.bias)


FUSED_DETECT_LEVEL = True      # development switch: False = head GEMM + ly_detect_tail in the training forward (measured 25-50 us per step slower)


class DetectHeadFn(torch.autograd.Function):
    """One Detect level in training (models/yolo.py:84-88): p = (conv1x1(x) + bias).view(bs, na, no, ny, nx).permute(0, 1, 3, 4, 2) as the
    fp32 raw map the loss reads.  forward = the head contraction + ly_detect_tail (permute + conversion, one pass); backward = ONE kernel
    from dp to the zero-padded rows the dgrad / wgrad contractions take, with the bias gradient summed on the way (autograd's chain for
    the same thing: cast, two layout copies, zero-padded copy, a reduction — 8 launches per level)."""

    MAXW, MAXLD = 160, 32           # ly_detect_head_bwd's LDS tile

    @staticmethod
    def forward(ctx, det, i, wp, x, weight, bias):
        co = weight.shape[0]
        spec = ConvSpec("pw", co)
        bias_f = bias.detach().float().contiguous()
        bs, cin, ny, nx = x.shape
        p = torch.empty((bs, det.na, ny, nx, det.no), dtype=torch.float32, device=x.device)
        t0, ld = ops.rows(x)
        if FUSED_DETECT_LEVEL and ops.detect_level_ok(cin, det.na, det.no, t0.dtype) and ld % ops.vw_of(t0) == 0 and t0.data_ptr() % 16 == 0:
            # head contraction + permute in one launch (csrc/ly_detect.hip), on the step's packed weights as they are
            ops.detect_level(t0, ld, bs, ny, nx, cin, wp, bias_f, det.na, det.no, det.anchors[i], 1.0, p, None, 0, 0, nat=False)
        else:
            y = _conv_forward(spec, x, None, wp, None, bias_f, ACT_NONE)
            ops.detect_tail(y, co, bs, ny, nx, det.na, det.no, det.anchors[i], 1.0, p, None, 0, 0)
        ctx.spec, ctx.geom = spec, (bs, ny, nx, det.na, det.no)
        ctx.slots = (_slot_of(x), None)
        ctx.params = (weight, bias)
        ctx.save_for_backward(x, weight)
        return p

    @staticmethod
    def backward(ctx, dp):
        x, weight = ctx.saved_tensors
        w_param, b_param = ctx.params
        bs, ny, nx, na, no = ctx.geom
        co = na * no
        need = ctx.needs_input_grad          # (det, i, wp, x, weight, bias)
        with torch.no_grad():
            t0, _ = ops.rows(x)
            vw = ops.vw_of(t0)
            cq = (co + vw - 1) // vw * vw
            dp = dp.float().contiguous()
            tb = ops.grad_target(b_param)
            tb = tb if tb is not None and tb.is_contiguous() and tb.numel() == co else None
            defer = tb is not None and ops.small_grads_ok()       # float64 scratch, rounded into the sink when the backward pass ends
            fresh64 = tb is None and ops.DETERMINISTIC_SMALL_GRADS      # no sink: float64 scratch, rounded right away (returned to autograd)
            dbias = ops.small_grad_scratch(tb, b_param) if defer else tb if tb is not None else \
                ops.zeros_f64(co, dp.device) if fresh64 else torch.zeros(co, dtype=torch.float32, device=dp.device)
            du = torch.empty((bs, ny, nx, cq), dtype=t0.dtype, device=dp.device)
            ops.detect_head_bwd(dp, bs, ny, nx, na, no, du, cq, dbias)
            if fresh64:
                dbias = ops.f64_round([dbias], [(co,)])[0]
            if tb is not None and not defer:
                ops.grad_done(b_param)
            du_d = du.permute(0, 3, 1, 2)
            dw = conv_wgrad(ctx.spec, du_d, x, None, w_param) if need[4] else None
            dx = None
            if need[3]:
                dx, _ = conv_dgrad(ConvSpec("pw", cq), du_d, weight, x, None, True, False, ctx.slots)
        return None, None, None, dx, dw, (None if tb is not None else dbias)


def detect_head(det, i, wp, x, weight, bias):
    """Detect level i in training -> fp32 raw map [bs, na, ny, nx, no]; the fused node when the map fits its kernel, else the generic
    conv node + autograd's permute."""
    co = weight.shape[0]
    n, _, h, w = x.shape
    if w <= DetectHeadFn.MAXW and (co + 7) // 8 * 8 <= DetectHeadFn.MAXLD and bias is not None:
        return DetectHeadFn.apply(det, i, wp, x, weight, bias)
    y = conv_bn_act(ConvSpec("pw", co), wp, x, None, weight, bias, None)
    p = torch.empty((n, det.na, h, w, det.no), dtype=torch.float32, device=y.device)
    p.copy_(y.view(n, det.na, det.no, h, w).permute(0, 1, 3, 4, 2))
    return p


# --------------------------------------------------------------------------------------------------
# MLPBlock:  y = x + W2 . relu(BN(W1 . z)),  z = [pconv3x3(x[:, :C/4]) | x[:, C/4:]]
# forward  = the fused kernel (statistics pass + normal pass), only x is saved
# backward = recompute z, u1 = W1 z, h = relu(BN u1) with the contraction kernels, then the unit scheme above
# --------------------------------------------------------------------------------------------------
def _ceil4(v):
    return (v + 3) // 4 * 4


class MlpBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x, wpc, w1, gamma, beta, w2):
        x = _rows_dense(x)
        n, c, h, w = x.shape
        pk_p, pk_1, pk_2 = mod._weights(ops.planes_of(x))
        htp = (2 * c // 16 + 1) // 2 * 2
        stats = ops.new_stats(16 * htp, x.device)
        ops.mlpblock(x, None, n, h, w, c, pk_p, pk_1, pk_2, None, None, stats=stats)
        a, b, mean, invstd = ops.bn_finalize(mod.mlp[1], stats, 16 * htp, n * h * w, n=2 * c, pad_to=16 * htp, want_stats=True)
        y = ops.empty_nhwc(n, c, h, w, x)
        ops.mlpblock(x, y, n, h, w, c, pk_p, pk_1, pk_2, a, b)
        a, b = a[:2 * c], b[:2 * c]
        ctx.mod = mod
        ctx.params = (wpc, w1, gamma, beta, w2)
        ctx.save_for_backward(x, wpc, w1, w2, a, b, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wpc, w1, w2, a, b, mean, invstd = ctx.saved_tensors
        p_wpc, p_w1, p_gamma, p_beta, p_w2 = ctx.params
        n, c, h, w = x.shape
        m = n * h * w
        c4 = c // 4
        c4p = _ceil4(c4)
        with torch.no_grad():
            dy = _rows_dense(dy if dy.dtype == x.dtype else dy.to(x.dtype))
            pl = ops.planes_of(x)
            if ops.mlpblock_bwd_ok(x, c):
                return MlpBlockFn._backward_fused(ctx, dy)
            c4q = (c4 + 31) // 32 * 32                 # the 3x3 kernel's tap matrices pad the channels of a tap to 32
            # recompute z, u1, h
            # z = [pconv(x[:c4]) | x[c4:]]: one pass of the persistent kernel where it is built, else a copy + the 3x3 kernel on the slice
            z = torch.empty_like(x)
            if not ops.mlpblock_pconv(x, z, n, h, w, c, pack.packed(pack.src_taps(p_wpc, c4p), 9 * c4p, pl)):
                z.copy_(x)
                ops.conv3x3(M=m, H=h, W=w, Cin=c4p, N=c4, x=x, ldx=c, wp=pack.packed(pack.src_taps(p_wpc, c4q), 9 * c4q, pl), out=z, ldo=c)
            htp = (2 * c // 16 + 1) // 2 * 2
            pk1 = pack.packed(pack.src_matrix(p_w1, 2 * c, c), c, pl, rows_to=16 * htp)          # the forward's own image
            u1 = ops.empty_nhwc(n, 2 * c, h, w, x)
            ops.gemm(M=m, H=h, W=w, K=c, N=2 * c, a0=z, lda0=c, k0=c, wp=pk1, out=u1, ldo=2 * c)
            # The hidden tensor relu(BN(u1)): for the narrow blocks (C <= 40: the 64-wide wgrad tiles) it is never rebuilt — the second 1x1's
            # weight gradient reads u1 through ly_wgrad's x prologue (one GEMM over the largest maps less: 60 / 30 us at 160 / 80 px).  In the
            # 128 x 128 tile the prologue's conversions sit between the two barriers of every step: +40 us per launch for a 17 us GEMM.
            pro = c <= 40
            if not pro:
                hid = ops.empty_nhwc(n, 2 * c, h, w, x)
                ops.gemm(M=m, H=h, W=w, K=c, N=2 * c, a0=z, lda0=c, k0=c, wp=pk1, out=hid, ldo=2 * c, e_scale=a, e_shift=b, act=ACT_RELU)
            # second 1x1
            dh = ops.empty_nhwc(n, 2 * c, h, w, x)
            ops.gemm(M=m, H=h, W=w, K=c, N=2 * c, a0=dy, lda0=c, k0=c, wp=pack.packed(pack.src_matrix(p_w2, 2 * c, c, sr=1, sk=2 * c), c, pl), out=dh, ldo=2 * c)
            def sink(p):
                """(gradient tensor to add into, whether it is the parameter's own storage)"""
                t = ops.grad_target(p)
                return (t, True) if t is not None else (torch.zeros(p.shape, dtype=torch.float32, device=x.device), False)
            dw2, d2 = sink(p_w2)
            wg2 = dict(M=m, H=h, W=w, N=c, du=dy, lddu=c, x=u1 if pro else hid, ldx=2 * c, Hin=h, Win=w, Cin=2 * c, dw=dw2, lddw=2 * c,
                       x_scale=a if pro else None, x_shift=b if pro else None)
            # BN + ReLU (with the prologue du1 is written over dh: u1 is still read by the weight gradient above, launched with the first
            # 1x1's below)
            du1, dgamma, dbeta = affine_backward(dh, u1, a, b, ACT_RELU, mean, invstd, True, gamma=p_gamma, beta=p_beta, out=dh if pro else None)
            # first 1x1
            g = ops.empty_nhwc(n, c, h, w, x)
            ops.gemm(M=m, H=h, W=w, K=2 * c, N=c, a0=du1, lda0=2 * c, k0=2 * c, wp=pack.packed(pack.src_matrix(p_w1, c, 2 * c, sr=1, sk=c), 2 * c, pl), out=g, ldo=c)
            dw1, d1 = sink(p_w1)
            # both 1x1 weight gradients in one launch (they share the launch's blocks: longer pixel chunks, half the tile flushes per pixel)
            ops.wgrad_group([wg2, dict(M=m, H=h, W=w, N=2 * c, du=du1, lddu=2 * c, x=z, ldx=c, Hin=h, Win=w, Cin=c, dw=dw1, lddw=c)])
            # partial 3x3 conv on the first C/4 channels
            # channel counts padded to multiples of 4 (C/4 = 6, 10): the tiled wgrad then applies; the extra rows / columns
            # (gradients of, and against, the neighbouring untouched channels) are computed and discarded
            # written in the weight's own [c4][c4][3][3] layout; the padded rows / channels are masked off by n_valid / c_valid
            dwp, dp = sink(p_wpc)
            ts, cs = (c4, 1) if _tap_major(dwp) else (1, 9)
            ops.wgrad(M=m, H=h, W=w, N=c4p, du=g, lddu=c, x=x, ldx=c, Hin=h, Win=w, Cin=c4p, dw=dwp, lddw=9 * c4, ks=3, stride=1, pad=1,
                      dw_ts=ts, dw_cs=cs, n_valid=c4, c_valid=c4)
            dx = torch.empty_like(dy)                      # dy + g, the partial conv's channels dy + t
            gz = torch.empty_like(g)
            if ops.mlpblock_bwd_dx_ok(x, c, w):
                # C >= 160 (the fused two-pass backward is not built there): the tail still runs as one launch (csrc/ly_mlpblock_bwd.hpp)
                dx, _ = ops.mlpblock_bwd_dx(g, dy, x, n, h, w, c, pack.packed(pack.src_taps(p_wpc, c4p, transposed_flipped=True), 9 * c4p, pl))
            elif ops.mlpblock_pconv(g, gz, n, h, w, c, pack.packed(pack.src_taps(p_wpc, c4p, transposed_flipped=True), 9 * c4p, pl)):
                # gz = [t | g[c4:]] straight from the persistent kernel with the transposed-flipped taps: dx = dy + gz
                _lib().check(_lib().lib().ly_mlp_dx(_lib().ptr(dy), _lib().ptr(gz), _lib().ptr(gz), c, m, c, 4, _lib().ptr(dx), _lib().dtype_code(dy),
                                                    _lib().stream_ptr()), "ly_mlp_dx")
            else:
                t = ops.empty_nhwc(n, c4p, h, w, x)
                wt = pack.packed(pack.src_taps(p_wpc, c4q, transposed_flipped=True), 9 * c4q, pl)
                ops.conv3x3(M=m, H=h, W=w, Cin=c4p, N=c4, x=g, ldx=c, wp=wt, out=t, ldo=c4p)
                _lib().check(_lib().lib().ly_mlp_dx(_lib().ptr(dy), _lib().ptr(g), _lib().ptr(t), c4p, m, c, c4, _lib().ptr(dx), _lib().dtype_code(dy),
                                                    _lib().stream_ptr()), "ly_mlp_dx")
            for prm, direct in ((p_wpc, dp), (p_w1, d1), (p_w2, d2)):
                if direct:
                    ops.grad_done(prm)
        return (None, dx, None if dp else dwp, None if d1 else dw1, dgamma, dbeta, None if d2 else dw2)


def _mlpblock_backward_fused(ctx, dy):
    """MLPBlock backward on the fused kernels (csrc/ly_mlpblock_bwd.hpp): two passes over (x, dy) with the 2C-wide hidden tensors on chip —
    pass 1 the BatchNorm sums, pass 2 g = d/dz and both 1x1 weight gradients — then the partial conv's two gradients on g.
    dy: dense NHWC, the storage dtype."""
    x, wpc, w1, w2, a, b, mean, invstd = ctx.saved_tensors
    p_wpc, p_w1, p_gamma, p_beta, p_w2 = ctx.params
    n, c, h, w = x.shape
    m = n * h * w
    c4 = c // 4
    c4p = _ceil4(c4)
    pl = ops.planes_of(x)
    htp = (2 * c // 16 + 1) // 2 * 2
    pk_p, pk_1, _ = ctx.mod._weights(pl)
    pk_2t = pack.packed(pack.src_matrix(p_w2, 2 * c, c, sr=1, sk=2 * c), c, pl, rows_to=16 * htp)
    pk_1t = pack.packed(pack.src_matrix(p_w1, c, 2 * c, sr=1, sk=c), 2 * c, pl)
    # pass 1: s1 = sum dv, s2 = sum dv u
    sums = ops.new_stats(2 * c, x.device)
    ops.mlpblock_bwd(x, dy, n, h, w, c, pk_p, pk_1, pk_2t, pk_1t, a, b, stats=sums)
    tg, tb = ops.grad_target(p_gamma), ops.grad_target(p_beta)
    direct = tg is not None and tb is not None and tg.numel() == 2 * c and tb.numel() == 2 * c
    dgamma, dbeta, alpha, kappa, lam = ops.bn_bwd_coeffs(sums, 2 * c, m, a, mean, invstd, True, dgamma=tg if direct else None,
                                                         dbeta=tb if direct else None)
    if direct:
        ops.grad_done(p_gamma)
        ops.grad_done(p_beta)

    def sink(p):
        t = ops.grad_target(p)
        return (t, True) if t is not None and t.is_contiguous() else (torch.zeros(p.shape, dtype=torch.float32, device=x.device), False)
    dw1, d1 = sink(p_w1)
    dw2, d2 = sink(p_w2)
    # pass 2: g, dW1, dW2
    g = ops.empty_nhwc(n, c, h, w, x)
    ops.mlpblock_bwd(x, dy, n, h, w, c, pk_p, pk_1, pk_2t, pk_1t, a, b, g=g, alpha=alpha, kappa=kappa, lam=lam, dw1=dw1, dw2=dw2)
    # partial 3x3 conv on the first C/4 channels: weight gradient from (g, x), data gradient folded into dx = dy + [conv^T(g[:C/4]) | g[C/4:]]
    tgt = ops.grad_target(p_wpc)
    dwp, dp = (tgt, True) if tgt is not None else (torch.zeros(p_wpc.shape, dtype=torch.float32, device=x.device), False)
    ts, cs = (c4, 1) if _tap_major(dwp) else (1, 9)
    wt = pack.packed(pack.src_taps(p_wpc, c4p, transposed_flipped=True), 9 * c4p, pl)
    # the tail in ONE launch: dx = dy + [conv^T(g[:C/4]) | g[C/4:]] and the partial conv's weight gradient out of the same patches
    dx, wdone = ops.mlpblock_bwd_dx(g, dy, x, n, h, w, c, wt, dwp=dwp, lddw=9 * c4, dw_ts=ts, dw_cs=cs)
    if not wdone:
        ops.wgrad(M=m, H=h, W=w, N=c4p, du=g, lddu=c, x=x, ldx=c, Hin=h, Win=w, Cin=c4p, dw=dwp, lddw=9 * c4, ks=3, stride=1, pad=1,
                  dw_ts=ts, dw_cs=cs, n_valid=c4, c_valid=c4)
    for prm, direct_ in ((p_wpc, dp), (p_w1, d1), (p_w2, d2)):
        if direct_:
            ops.grad_done(prm)
    return (None, dx, None if dp else dwp, None if d1 else dw1, dgamma, dbeta, None if d2 else dw2)


MlpBlockFn._backward_fused = staticmethod(_mlpblock_backward_fused)


# --------------------------------------------------------------------------------------------------
# CoordAtt: ONE autograd node, every step a HIP kernel in both directions (pools, the [n, h+w, c] -> [n, h+w, mip] -> a_h, a_w MLP with
# bn1's batch statistics, the gate).
# --------------------------------------------------------------------------------------------------
class CoordAttFn(torch.autograd.Function):
    """The whole CoordAtt (models/common.py:1595-1609) as ONE autograd node: pools -> conv1 -> bn1 (batch statistics) -> h_swish ->
    conv_h / conv_w -> sigmoid -> gate.  Forward = the inference kernels plus the statistics pass (6 launches); backward = gate
    backward, the two-launch MLP backward (csrc/ly_attention.hip, ly_coordatt_mlp_bwd) and the pool gradient added into dx in place.
    The same arithmetic as torch ops under autograd was ~60 launches of rocBLAS GEMMs on [n*(h+w), 8] matrices, native BatchNorm and
    elementwise kernels per block."""

    @staticmethod
    def forward(ctx, mod, x, w1, b1, gamma, beta, wh, bh, ww, bw):
        t, ld = ops.rows(x)
        n, c, h, w = t.shape
        mip = mod.mip
        W1, Wh, Ww = (p.detach().reshape(p.shape[0], -1) for p in (w1, wh, ww))
        pool = ops.pool_hw(t, ld, n, h, w, c)
        st = ops.coordatt_conv1_stats(pool, n * (h + w), c, mip, W1, b1)
        sc, sh, mean, invstd = ops.bn_finalize(mod.bn1, st.view(1, -1), mip, n * (h + w), want_stats=True)
        a_h, a_w = ops.coordatt_mlp(pool, n, h, w, c, mip, W1, b1.detach(), Wh, bh.detach(), Ww, bw.detach(), sc=sc, sh=sh)
        out = ops.coordatt_gate(t, ld, n, h, w, c, a_h, a_w)
        ctx.save_for_backward(t, pool, a_h, a_w, mean, invstd)
        ctx.params = (w1, b1, gamma, beta, wh, bh, ww, bw)
        ctx.mip = mip
        return out

    @staticmethod
    def backward(ctx, dout):
        t, pool, a_h, a_w, mean, invstd = ctx.saved_tensors
        xr, ld = ops.rows(t)
        n, c, h, w = xr.shape
        w1, b1, gamma, beta, wh, bh, ww, bw = ctx.params
        dr, ldd = ops.rows(dout if dout.dtype == xr.dtype else dout.to(xr.dtype))          # a channel slice of a wider gradient: read in place
        dx, da_h, da_w = ops.coordatt_gate_bwd(dr, xr, ld, n, h, w, c, a_h, a_w, ldd=ldd)
        targets, ret = [], []
        for i, p in enumerate(ctx.params):                    # accumulate into the persistent gradient storage where there is one
            if i == 1:                                        # conv1.bias: bn1 removes the batch mean, d/dbias == 0 (nothing to add)
                sunk = p.requires_grad and ops.grad_target(p) is not None
                ret.append(None if (sunk or not p.requires_grad) else torch.zeros_like(p))
                continue
            tgt = ops.grad_target(p) if p.requires_grad else None
            fresh = tgt is None
            if fresh:
                tgt = torch.zeros(p.shape, dtype=torch.float32, device=xr.device)
            targets.append(tgt)
            ret.append(tgt if (fresh and p.requires_grad) else None)
        # every target in the sink: accumulate in float64 scratches, rounded into the sink when the backward pass ends (ops.small_grad_scratch)
        defer = ops.small_grads_ok() and all(r is None for r in ret) and all(t.is_contiguous() for t in targets)
        no_sink = all(q is None or not q.requires_grad or ops.grad_target(q) is None for i, q in enumerate(ctx.params) if i != 1)
        fresh64 = (not defer) and ops.DETERMINISTIC_SMALL_GRADS and no_sink
        if defer:
            targets = [ops.small_grad_scratch(t, q) for t, q in zip(targets, [q for i, q in enumerate(ctx.params) if i != 1])]
        elif fresh64:                                         # no sink anywhere: float64 scratches, rounded right after the launch
            shapes = [t.shape for t in targets]
            targets = [ops.zeros_f64(t.numel(), xr.device) for t in targets]
        W1, Wh, Ww = (p.detach().reshape(p.shape[0], -1) for p in (w1, wh, ww))
        dpool = ops.coordatt_mlp_bwd(pool, n, h, w, c, ctx.mip, W1, b1.detach(), mean, invstd, gamma.detach(), beta.detach(), Wh, Ww, a_h, a_w,
                                     da_h, da_w, targets)
        if fresh64:
            rounded, k, out = ops.f64_round(targets, shapes), 0, []
            for i, r in enumerate(ret):
                if i == 1:
                    out.append(r)
                    continue
                out.append(rounded[k] if r is not None else None)
                k += 1
            ret = out
        ops.pool_hw_bwd(dpool, n, h, w, c, into=dx)
        for i, (p, r) in enumerate(zip(ctx.params, ret)):
            if r is None and p.requires_grad and not (defer and i != 1):
                ops.grad_done(p)
        return (None, dx, *ret)


def coordatt_train(mod, x):
    """CoordAtt.forward in training (models/common.py:1595-1609)."""
    n, c, h, w = x.shape
    params = (mod.conv1.weight, mod.conv1.bias, mod.bn1.weight, mod.bn1.bias, mod.conv_h.weight, mod.conv_h.bias, mod.conv_w.weight, mod.conv_w.bias)
    if mod.mip in (8, 16) and c <= 512 and all(p is not None and p.dtype == torch.float32 for p in params):
        return CoordAttFn.apply(mod, x, *params)
    raise NotImplementedError(f"CoordAtt training is built for mip = max(8, c // 32) in (8, 16) and c <= 512 with float32 parameters (got mip={mod.mip}, "
                              f"c={c}): ly_coordatt_mlp_bwd has no instantiation for other widths, and there is no ATen fallback on the hot path")


# --------------------------------------------------------------------------------------------------
# SPPF pooling: [y | m(y) | m(m(y)) | m(m(m(y)))]
# --------------------------------------------------------------------------------------------------
class SppfPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, k):
        t, ld = ops.rows(y)
        n, c, h, w = t.shape
        if not ops.sppf_pool_fits(h, w):
            raise NotImplementedError("SPPF training path needs the map to fit the LDS pooling kernel")
        buf = ops.empty_nhwc(n, 4 * c, h, w, t)
        ops.sppf_pool(t, ld, n, h, w, c, k, buf, 4 * c)
        ctx.save_for_backward(buf)
        ctx.k = k
        return buf

    @staticmethod
    def backward(ctx, d):
        buf, = ctx.saved_tensors
        n, c4, h, w = buf.shape
        c = c4 // 4
        k = ctx.k
        if k != 5 or c % 4 != 0:
            acc = _rows_dense(d).float() if d.dtype != torch.float32 else _rows_dense(d).clone()   # fp32 accumulator (float atomics)
            for j in (2, 1, 0):                           # m(y_j) = y_{j+1}: add its gradient into slot j
                ops.maxpool_bwd(buf, j * c, c4, acc, (j + 1) * c, c4, n, h, w, c, k, acc, j * c, c4)
            return acc[:, :c].to(buf.dtype), None
        # gather formulation (no atomics): the routing of all three pools in one launch, then level by level
        L = _lib()
        st = L.stream_ptr()
        dd = _rows_dense(d)
        if dd.dtype == buf.dtype:
            # small maps (SPPF sits at stride 32): the three levels in ONE launch, map and routing in LDS (csrc/ly_backward.hip ly_sppf_bwd)
            out = ops.empty_nhwc(n, c, h, w, buf)
            rc = L.lib().ly_sppf_bwd(L.ptr(buf), c4, L.ptr(dd), c4, n, h, w, c, k, L.ptr(out), c, L.dtype_code(buf), st)
            if rc == 0:
                return out, None
            if rc != 1:
                L.check(rc, "ly_sppf_bwd")
        at = lambda t, off: ctypes.c_void_p(t.data_ptr() + t.element_size() * off)
        arg = torch.empty((n * h * w, 3 * c), dtype=torch.uint8, device=buf.device)
        L.check(L.lib().ly_maxpool_arg(L.ptr(buf), c4, n, h, w, 3 * c, k, L.ptr(arg), 3 * c, L.dtype_code(buf), st), "ly_maxpool_arg")
        up = dd[:, 3 * c:].float().contiguous(memory_format=torch.channels_last)            # total gradient of y3 = its direct gradient
        tmp = [torch.empty((n * h * w, c), dtype=torch.float32, device=buf.device) for _ in range(2)]
        out = ops.empty_nhwc(n, c, h, w, buf)
        up_p, up_ld = ops.rows(up)
        for j in (2, 1, 0):
            dst = out if j == 0 else tmp[j & 1]
            L.check(L.lib().ly_maxpool_gather(at(arg, j * c), 3 * c, L.ptr(up_p) if j == 2 else L.ptr(tmp[(j + 1) & 1]), c, at(dd, j * c), c4, L.dtype_code(dd),
                                              n, h, w, c, k, L.ptr(dst), c, L.dtype_code(dst), st), "ly_maxpool_gather")
        return out, None


# --------------------------------------------------------------------------------------------------
# RFCBAMConv (models/rfa.py:113-129).  forward = the fused inference kernels with batch statistics;
# backward = csrc/ly_rfcbam_bwd.hip.  The SE vector `ca` [n, c] is an INPUT (computed by torch ops on pooled
# vectors under autograd, see rfcbam_train), its gradient is returned.
# --------------------------------------------------------------------------------------------------
def _lib():
    from . import capi
    return capi


DEBUG_TAP = None        # tools/: callable(name, tensor) observing backward intermediates


def _tap(name, t):
    if DEBUG_TAP is not None:
        DEBUG_TAP(name, t)


class RfcbamFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x, se_wa, se_wb, gen_w, gen_gamma, gen_beta, getw, conv_w, conv_b, out_gamma, out_beta):
        xr, ld = ops.rows(x)
        n, c, h, w = xr.shape
        k, s, o = mod.kernel_size, mod.stride, mod.o
        ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
        P = mod._packed_train(ops.planes_of(xr))
        if se_wa.dtype != torch.float32:
            raise NotImplementedError("RFCBAMConv training needs float32 parameters (train under autocast, fp32 master weights)")
        from . import modules as _m
        rc = k == 3 and ops.rf3c_ok(c, s) and _m.RF3C             # lane = channel kernels (csrc/ly_rf3c.hip): the pooling partials come out of the statistics pass
        if not rc:
            # SE (models/rfa.py:88-92): pooling partials + the two linears, two launches; the partials are kept for the backward
            ca, se_part = ops.se_attention(xr, ld, n, h * w, c, se_wa.detach(), se_wb.detach(), se_wa.shape[0], want_part=True)
        bias = conv_b.detach().float().contiguous()
        if gen_w.dtype != torch.float32:
            raise NotImplementedError("RFCBAMConv training needs float32 parameters (train under autocast, fp32 master weights)")
        G = ops.rfcbam_gen_prepare(xr, ld, n, h, w, c, k, s, gen_w, mod.generate[1])          # generate BatchNorm: moments + ONE launch
        gs, gb, gmean, ginv = G["gs"], G["gb"], G["gmean"], G["ginv"]
        if k == 1:
            a1 = G["a1"]
            mm = ops.rfcbam_stats(xr, ld, n, h, w, c, 1, 1, a1=a1, b1=gb)
            rfa = ops.rfa_map(mm, P["w18"])
            kw = dict(M=n * h * w, H=h, W=w, K=c, N=o, a0=xr, lda0=ld, k0=c, wp=P["wp"], ldo=o, pro=ops.PRO_AFFINE_RELU_CA,
                      p_scale=a1, p_shift=gb, p_ca=ca, rowscale=rfa)
            # ONE contraction: statistics and the pre-BN value u (bias included) in the same launch; y = relu(es*u + t) is an elementwise
            # pass and u is kept for the backward (was: statistics pass + main pass + a recompute in backward)
            stats = ops.new_stats(o, xr.device)
            u = ops.empty_nhwc(n, o, h, w, xr)
            ops.gemm(out=u, e_scale=None, e_shift=bias, stats=stats, **kw)
            es, t, omean, oinv = ops.bn_finalize(mod.conv[1], stats, o, n * h * w, want_stats=True)
            out = torch.empty_like(u)
            ops.bnact_fwd(u, o, n * h * w, o, es, t, ACT_RELU, out, o)
            ctx.fwd = dict(kw=kw)
        elif rc:
            # three launches, x read twice: statistics + SE pooling partials | SE linears + get_weight conv | regenerate + contraction.  The RAW
            # generate image (u = w.x, v = a*u + b) is what the backward re-evaluates bit for bit
            th, tw = ops.pick_tile_c(ho, wo, s)
            mm, se_part = ops.rf3c_stats(xr, ld, n, h, w, c, s, G["wq_c"], th, tw, raw=True)
            ca, rfa = ops.rfcbam_mid(se_part, h * w, se_wa.detach(), se_wb.detach(), se_wa.shape[0], mm, P["w18"])
            kw = dict(n=n, h=h, w=w, c=c, ho=ho, wo=wo, N=o, s=s, th=th, tw=tw, x=xr, ldx=ld, wq=G["wq_c"], ca=ca, rfa=rfa, wp=P["wp_c"], ldo=o, raw=True)
            stats = ops.new_stats(o, xr.device)
            u = ops.empty_nhwc(n, o, ho, wo, xr)
            ops.rf3c_fwd(out=u, e_scale=ops.ones_f32(bias.numel(), bias.device), e_shift=bias, stats=stats, **kw)
            es, t, omean, oinv = ops.bn_finalize(mod.conv[1], stats, o, n * ho * wo, want_stats=True)
            out = torch.empty_like(u)
            ops.bnact_fwd(u, o, n * ho * wo, o, es, t, ACT_RELU, out, o)
            ctx.fwd = dict(kw=kw)
            ctx.rc = dict(th=th, tw=tw, wq=G["wq_c"])
        else:
            th, tw = ops.pick_tile(ho, wo)
            wq_stats, wq_main = G["wq_stats"], G["wq_main"]
            mm = ops.rfcbam_stats(xr, ld, n, h, w, c, 3, s, wg=wq_stats, th=th, tw=tw)
            rfa = ops.rfa_map(mm, P["w18"])
            kw = dict(n=n, h=h, w=w, c=c, ho=ho, wo=wo, N=o, s=s, th=th, tw=tw, x=xr, ldx=ld, wg=wq_main, ca=ca, rfa=rfa, wp=P["wp"], ldo=o)
            stats = ops.new_stats(o, xr.device)
            u = ops.empty_nhwc(n, o, ho, wo, xr)
            ops.rfcbam3(out=u, e_scale=torch.ones_like(bias), e_shift=bias, stats=stats, **kw)      # one contraction (see k = 1)
            es, t, omean, oinv = ops.bn_finalize(mod.conv[1], stats, o, n * ho * wo, want_stats=True)
            out = torch.empty_like(u)
            ops.bnact_fwd(u, o, n * ho * wo, o, es, t, ACT_RELU, out, o)
            ctx.fwd = dict(kw=kw)
        if not rc:
            ctx.rc = None
        ctx.geom = (n, c, h, w, k, s, o, ho, wo, ld)
        ctx.conv_w_param = conv_w
        ctx.se_params = (se_wa, se_wb)
        ctx.conv_b_param = conv_b
        ctx.out_bn_params = (out_gamma, out_beta)
        ctx.getw_param = getw
        ctx.gen_w_param = gen_w
        ctx.gen_bn_params = (gen_gamma, gen_beta)
        ctx.save_for_backward(xr, ca, gen_w, getw, conv_w, bias, G["ag"], G["bg"], G["gmean_tc"], G["ginv_tc"], es, t, omean, oinv, mm, rfa, se_part, u)
        return out

    @staticmethod
    def backward(ctx, dy):
        xr, ca, gen_w, getw, conv_w, bias, ag, bg, gmean_tc, ginv_tc, es, t, omean, oinv, mm, rfa, se_part, u = ctx.saved_tensors
        n, c, h, w, k, s, o, ho, wo, ld = ctx.geom
        kk = k * k
        mo = n * ho * wo
        dev = xr.device
        L = _lib()
        st = L.stream_ptr()
        p = L.ptr
        with torch.no_grad():
            dt = xr.dtype
            pl = ops.planes_of(xr)
            code = L.dtype_code(xr)
            dy = _rows_dense(dy if dy.dtype == dt else dy.to(dt))
            # 1-2. output conv: the pre-BN value u (bias included) was kept by the forward; BN + ReLU backward (du is written over a copy
            # of u: saved tensors must stay intact for a second backward through the graph)
            _tap("rf.dy", dy); _tap("rf.u", u)
            du, dgo, dbo = affine_backward(dy, u, es, t, ACT_RELU, omean, oinv, True, inplace=False, gamma=ctx.out_bn_params[0],
                                           beta=ctx.out_bn_params[1])       # (straight into the sink when one holds them: returns None, None)
            _tap("rf.du", du)
            # (o = 256 — layer 20 — measured SLOWER on the recompute passes than on the streamed 9x tensors: 1.08 vs 0.73 ms of kernels at bs = 64;
            # layer 17, o = 128: 0.95 vs 1.47 ms.  The wider layer stays on the first-generation backward.)
            if ctx.rc is not None and dt == torch.bfloat16 and s == 2 and o in RC_BWD_WIDTHS and RC_BWD:
                return RfcbamFn._backward_rc(ctx, du, dgo, dbo)
            if k == 1 and RF1_BWD and c % ops.vw_of(xr) == 0 and c // ops.vw_of(xr) <= 64 and ld % ops.vw_of(xr) == 0:
                return RfcbamFn._backward_k1(ctx, du, dgo, dbo)
            # 3. dcd [mo][t][c]
            # Wc^T with rows (t, c): conv.0.weight [o, c, kh, kw] read in place
            dcd = torch.empty((mo, kk * c), dtype=dt, device=dev)
            wct = pack.packed(pack.Src(ctx.conv_w_param, kk * c, nrb=c, sra=1, srb=kk, nc=o, sc=c * kk), o, pl)
            ops.gemm(M=mo, H=ho, W=wo, K=o, N=kk * c, a0=du, lda0=o, k0=o, wp=wct, out=dcd, ldo=kk * c)
            _tap("rf.dcd", dcd)
            # 4. ug
            wg = gen_w.detach().float().reshape(c * kk, kk).contiguous()
            ug = torch.empty((mo, kk * c), dtype=dt, device=dev)
            es9 = xr.element_size() * mo * kk * c                 # bytes of one expanded tensor
            with ops._Timed(f"ly_rf_generate_kernel<{ops._tname(xr)}, {k}>", 0.0, xr.element_size() * n * h * w * c + es9, valu_flops=2.0 * mo * kk * kk * c):
                L.check(L.lib().ly_rf_generate(p(xr), ld, n, h, w, c, k, s, p(wg), p(ug), code, st), "ly_rf_generate")
            # 5. cd, d_rfa, gmax, d_ca
            cd = torch.empty((mo, kk * c), dtype=dt, device=dev)
            zz = ops.zeros_f32(2 * rfa.numel() + ca.numel(), dev)
            d_rfa, gmax, d_ca = zz[:rfa.numel()].view_as(rfa), zz[rfa.numel():2 * rfa.numel()].view_as(rfa), zz[2 * rfa.numel():].view_as(ca)
            vw = ops.vw_of(xr)
            rf3s = k == 3 and RF3S_BWD and c % vw == 0 and c // vw <= 64          # 16-bytes-per-lane passes (csrc/ly_rf1_bwd.hip: ly_rf3s_bwd)
            if rf3s:
                d_ca64 = ops.zeros_f64(ca.numel(), dev)                            # double accumulators: d_ca feeds dx through SE's backward
                P3 = L.LyRf1BwdParams(n, ho * wo, c, p(ug), c, p(dcd), None, p(ag), p(bg), p(ca), p(rfa), p(cd), p(d_rfa), p(gmax), p(d_ca64),
                                      p(gmax), None, None, None, None, None, None, 0.0, None, c, None, code)
                with ops._Timed(f"ly_rf3s_bwd_kernel<{ops._tname(xr)}, 0>", 8.0 * mo * kk * c, 3.0 * es9):
                    L.check(L.lib().ly_rf3s_bwd(ctypes.byref(P3), ho, wo, 0, st), "ly_rf3s_bwd 0")
                d_ca = d_ca64                                            # read as doubles by ly_se_bwd
            else:
                with ops._Timed(f"ly_rf_bwd_attn_kernel<{ops._tname(xr)}, {k}>", 8.0 * mo * kk * c, 3.0 * es9):
                    L.check(L.lib().ly_rf_bwd_attn(n, h, w, c, k, s, p(ug), p(dcd), p(ag), p(bg), p(ca), p(rfa), p(cd), p(d_rfa), p(gmax), p(d_ca), code, st),
                            "ly_rf_bwd_attn")
            _tap("rf.ug", ug); _tap("rf.cd", cd); _tap("rf.d_rfa", d_rfa); _tap("rf.gmax", gmax); _tap("rf.d_ca", d_ca)
            _tap("rf.rfa", rfa); _tap("rf.ca", ca); _tap("rf.ag", ag); _tap("rf.bg", bg)
            # 6. conv weight gradient
            tgt = ops.grad_target(ctx.conv_w_param)            # k = 1: the weight's own layout is what ly_wgrad writes
            if tgt is not None and kk == 1:
                ops.wgrad(M=mo, H=ho, W=wo, N=o, du=du, lddu=o, x=cd, ldx=c, Hin=ho, Win=wo, Cin=c, dw=tgt, lddw=c)
                ops.grad_done(ctx.conv_w_param)
                dwc = None
            elif _tap_major_rows(tgt) is not None:                  # k = 3 with a tap-major sink: [o][t][c] is the storage's own order
                ops.wgrad(M=mo, H=ho, W=wo, N=o, du=du, lddu=o, x=cd, ldx=kk * c, Hin=ho, Win=wo, Cin=kk * c, dw=_tap_major_rows(tgt), lddw=kk * c)
                ops.grad_done(ctx.conv_w_param)
                dwc = None
            else:
                dwc = torch.zeros(o, kk * c, dtype=torch.float32, device=dev)      # handed to autograd (k = 1: as a view): not from the pool
                ops.wgrad(M=mo, H=ho, W=wo, N=o, du=du, lddu=o, x=cd, ldx=kk * c, Hin=ho, Win=wo, Cin=kk * c, dw=dwc, lddw=kk * c)
                dwc = dwc.view(o, kk, c).permute(0, 2, 1).reshape(conv_w.shape)
            # 7. get_weight + sigmoid
            w18 = getw.detach().float().reshape(18).contiguous()
            d_mm = torch.empty_like(mm)
            t18 = ops.grad_target(ctx.getw_param)
            t18 = t18 if t18 is not None and t18.is_contiguous() else None
            d18 = t18 is not None and ops.small_grads_ok()          # deferred: a float64 scratch, rounded into the sink when the backward pass ends
            f18 = t18 is None and ops.DETERMINISTIC_SMALL_GRADS          # no sink: float64 scratch, rounded right away (returned to autograd)
            dw18 = ops.small_grad_scratch(t18, ctx.getw_param) if d18 else t18.view(-1) if t18 is not None else \
                ops.zeros_f64(18, dev) if f18 else torch.zeros(18, dtype=torch.float32, device=dev)
            L.check(L.lib().ly_rfa_bwd(p(d_rfa), p(rfa), p(mm), p(w18), n, k * ho, k * wo, p(d_mm), p(dw18), int(d18 or f18), st), "ly_rfa_bwd")
            if f18:
                dw18 = ops.f64_round([dw18], [(18,)])[0]
            if t18 is not None and not d18:
                ops.grad_done(ctx.getw_param)
            # 8. through max/mean, ca, rfa and ReLU; generate-BN sums
            if rf3s:
                sums = ops.new_stats(kk * c, dev)                                   # striped [STRIPES][2][9][c] doubles
                P3.d_mm, P3.sums = p(d_mm), p(sums)
                with ops._Timed(f"ly_rf3s_bwd_kernel<{ops._tname(xr)}, 1>", 8.0 * mo * kk * c, 3.0 * es9):
                    L.check(L.lib().ly_rf3s_bwd(ctypes.byref(P3), ho, wo, 1, st), "ly_rf3s_bwd 1")
            else:
                sums = ops.zeros_f32(2 * kk * c, dev)
                with ops._Timed(f"ly_rf_bwd_relu_kernel<{ops._tname(xr)}, {k}>", 8.0 * mo * kk * c, 3.0 * es9):
                    L.check(L.lib().ly_rf_bwd_relu(n, h, w, c, k, s, p(ug), p(dcd), p(ag), p(bg), p(ca), p(rfa), p(gmax), p(d_mm), p(sums), code, st),
                            "ly_rf_bwd_relu")
            _tap("rf.d_mm", d_mm); _tap("rf.dv", dcd); _tap("rf.sums", sums)
            # 9. generate BatchNorm coefficients ([t][c] order)
            tgg, tgb = (ops.grad_target(q) for q in ctx.gen_bn_params)
            bn_direct = all(q is not None and q.numel() == kk * c and q.is_contiguous() for q in (tgg, tgb))
            dgg_tc, dbg_tc, alpha, kappa, lam = ops.bn_bwd_coeffs(sums, kk * c, mo, ag, gmean_tc, ginv_tc, True, dgamma=tgg if bn_direct else None,
                                                                  dbeta=tgb if bn_direct else None, transpose=(kk, c))
            if bn_direct:
                ops.grad_done(ctx.gen_bn_params[0])
                ops.grad_done(ctx.gen_bn_params[1])
            ct = lambda v: None if v is None else v.view(kk, c).t().contiguous().view(-1)
            # 10. dug, generate weight gradient
            part_rows = 512
            dwg = torch.zeros(part_rows, c * kk, kk, dtype=torch.float32, device=dev)         # per-block partial sums, summed below
            with ops._Timed(f"ly_rf_bwd_gen_kernel<{ops._tname(xr)}, {k}>", 0.0, 3.0 * es9, valu_flops=2.0 * mo * kk * kk * c):
                L.check(L.lib().ly_rf_bwd_gen(p(xr), ld, n, h, w, c, k, s, p(ug), p(dcd), p(alpha), p(kappa), p(lam), p(dwg), part_rows, code, st),
                        "ly_rf_bwd_gen")
            _tap("rf.coef", alpha); _tap("rf.dwg", dwg)
            # 11. dx
            # SE backward: parameter gradients, and d/d(mean x) which the dx kernel spreads over the pixels while it writes dx
            se_wa, se_wb = ctx.se_params
            ta, tb = ops.grad_target(se_wa), ops.grad_target(se_wb)
            se_direct = ta is not None and tb is not None
            dwa = ta if se_direct else torch.zeros(se_wa.shape, dtype=torch.float32, device=dev)
            dwb = tb if se_direct else torch.zeros(se_wb.shape, dtype=torch.float32, device=dev)
            dgap = ops.se_bwd(se_part, n, h * w, c, se_wa.detach(), se_wb.detach(), se_wa.shape[0], ca, d_ca, dwa, dwb)
            if se_direct:
                ops.grad_done(se_wa)
                ops.grad_done(se_wb)
            dx = None
            if ctx.needs_input_grad[1]:
                dx = ops.empty_nhwc(n, c, h, w, xr)
                with ops._Timed(f"ly_rf_bwd_dx_kernel<{ops._tname(xr)}, {k}>", 0.0, es9 + xr.element_size() * n * h * w * c, valu_flops=2.0 * mo * kk * kk * c):
                    L.check(L.lib().ly_rf_bwd_dx(n, h, w, c, k, s, p(dcd), p(wg), p(dx), c, p(dgap), 1.0 / (h * w), code, st), "ly_rf_bwd_dx")
            dbias = None if ops.grad_target(ctx.conv_b_param) is not None else torch.zeros_like(bias)      # BN removes the batch mean: d/dbias = 0
            tgw = ops.grad_target(ctx.gen_w_param)
            if tgw is not None and tgw.is_contiguous() and tgw.numel() == dwg[0].numel():
                ops.sum_rows(dwg, out=tgw.view(-1), accumulate=True)        # the partial rows of d(generate.0.weight) added straight into the sink
                ops.grad_done(ctx.gen_w_param)
                dgw = None
            else:
                dgw = ops.sum_rows(dwg).view(gen_w.shape)
        return (None, dx, None if se_direct else dwa, None if se_direct else dwb, dgw, ct(dgg_tc), ct(dbg_tc),
                (None if t18 is not None else dw18.view(getw.shape)), dwc, dbias, dgo, dbo)


def _rfcbam_backward_rc(ctx, du, dgo, dbo):
    """RFCBAMConv k=3 backward on csrc/ly_rf3c_bwd.hip: no 9x-sized tensor in HBM — three recompute passes + the conv weight gradient"""
    xr, ca, gen_w, getw, conv_w, bias, ag, bg, gmean_tc, ginv_tc, es, t, omean, oinv, mm, rfa, se_part, u = ctx.saved_tensors
    n, c, h, w, k, s, o, ho, wo, ld = ctx.geom
    dev = xr.device
    L = _lib()
    st = L.stream_ptr()
    p = L.ptr
    mo, nch = n * ho * wo, c // 32
    th, tw, wq = ctx.rc["th"], ctx.rc["tw"], ctx.rc["wq"]
    wct = pack.packed(pack.Src(ctx.conv_w_param, 9 * c, nrb=c, sra=1, srb=9, nc=o, sc=c * 9), o, 1)
    npos = n * 9 * ho * wo
    d_rfa_part = torch.empty((nch, npos), dtype=torch.float32, device=dev)
    d_ca = torch.empty((n, c), dtype=torch.float32, device=dev)
    sums = torch.empty((n, 18 * c), dtype=torch.float32, device=dev)
    dwg = torch.empty((n, c * 81), dtype=torch.float32, device=dev)
    nog = -(-o // 128)
    ng = max(1, min(n, 256 // (nch * nog)))
    dwc_part = torch.empty((ng, o, 9, c), dtype=torch.float32, device=dev)
    need_dx = ctx.needs_input_grad[1]
    dx = ops.empty_nhwc(n, c, h, w, xr)
    P = L.LyRf3cBwdParams(n, h, w, c, ho, wo, o, s, th, tw, p(xr), ld, p(du), o, p(wq), p(wct), p(ca), p(rfa), p(mm), None, None,
                          p(d_rfa_part), p(d_ca), p(sums), p(dwg), p(dx), c, None, 1.0 / (h * w), p(dwc_part), ng, L.dtype_code(xr))
    es9 = 2.0 * mo * 9 * c
    xb = xr.element_size() * (n * h * w * c + mo * o)
    tn = ops._tname(xr)
    # A: d_rfa (one slab per channel chunk), d_ca
    with ops._Timed(f"ly_rf3c_bwd_kernel<0, {o // 32}>", 2.0 * mo * 9 * c * o, xb, valu_flops=2.0 * mo * c * 81):
        L.check(L.lib().ly_rf3c_bwd(ctypes.byref(P), 0, st), "ly_rf3c_bwd A")
    # conv weight gradient (independent of the chain below)
    with ops._Timed("ly_rf3c_wgrad_kernel", 2.0 * mo * 9 * c * o, xb, valu_flops=2.0 * mo * c * 81):
        L.check(L.lib().ly_rf3c_wgrad(ctypes.byref(P), st), "ly_rf3c_wgrad")
    trows = _tap_major_rows(ops.grad_target(ctx.conv_w_param))
    if trows is not None:                                       # tap-major sink: the fold of the partials adds straight into the gradient storage
        ops.sum_rows(dwc_part.view(ng, o, 9 * c), out=trows, accumulate=True)
        ops.grad_done(ctx.conv_w_param)
        dwc = None
    else:
        dwc = ops.sum_rows(dwc_part) if ng > 1 else dwc_part[0]
        dwc = dwc.permute(0, 2, 1).reshape(conv_w.shape)
    # get_weight + sigmoid
    d_rfa = ops.sum_rows(d_rfa_part).view_as(rfa) if nch > 1 else d_rfa_part[0].view_as(rfa)
    w18 = getw.detach().float().reshape(18).contiguous()
    d_mm = torch.empty_like(mm)
    t18 = ops.grad_target(ctx.getw_param)
    t18 = t18 if t18 is not None and t18.is_contiguous() else None
    d18 = t18 is not None and ops.small_grads_ok()          # deferred: a float64 scratch, rounded into the sink when the backward pass ends
    f18 = t18 is None and ops.DETERMINISTIC_SMALL_GRADS          # no sink: float64 scratch, rounded right away (returned to autograd)
    dw18 = ops.small_grad_scratch(t18, ctx.getw_param) if d18 else t18.view(-1) if t18 is not None else \
        ops.zeros_f64(18, dev) if f18 else torch.zeros(18, dtype=torch.float32, device=dev)
    L.check(L.lib().ly_rfa_bwd(p(d_rfa), p(rfa), p(mm), p(w18), n, 3 * ho, 3 * wo, p(d_mm), p(dw18), int(d18 or f18), st), "ly_rfa_bwd")
    if f18:
        dw18 = ops.f64_round([dw18], [(18,)])[0]
    if t18 is not None and not d18:
        ops.grad_done(ctx.getw_param)
    P.d_mm = p(d_mm)
    # B: BatchNorm sums, one stripe per image
    with ops._Timed(f"ly_rf3c_bwd_kernel<1, {o // 32}>", 2.0 * mo * 9 * c * o, xb, valu_flops=2.0 * mo * c * 81):
        L.check(L.lib().ly_rf3c_bwd(ctypes.byref(P), 1, st), "ly_rf3c_bwd B")
    tgg, tgb = (ops.grad_target(q) for q in ctx.gen_bn_params)
    bn_direct = all(q is not None and q.numel() == 9 * c and q.is_contiguous() for q in (tgg, tgb))
    dgg_tc, dbg_tc, alpha, kappa, lam = ops.bn_bwd_coeffs(sums, 9 * c, mo, ag, gmean_tc, ginv_tc, True, dgamma=tgg if bn_direct else None,
                                                          dbeta=tgb if bn_direct else None, transpose=(9, c))
    if bn_direct:
        ops.grad_done(ctx.gen_bn_params[0])
        ops.grad_done(ctx.gen_bn_params[1])
    P.coef = p(alpha)                          # alpha, kappa, lambda are rows 2..4 of one [5, 9c] tensor
    assert kappa.data_ptr() == alpha.data_ptr() + 4 * 9 * c and lam.data_ptr() == alpha.data_ptr() + 8 * 9 * c
    ct = lambda v: None if v is None else v.view(9, c).t().contiguous().view(-1)
    # SE backward: parameter gradients, and d/d(mean x) which pass C adds while it writes dx
    se_wa, se_wb = ctx.se_params
    ta, tb = ops.grad_target(se_wa), ops.grad_target(se_wb)
    se_direct = ta is not None and tb is not None
    dwa = ta if se_direct else torch.zeros(se_wa.shape, dtype=torch.float32, device=dev)
    dwb = tb if se_direct else torch.zeros(se_wb.shape, dtype=torch.float32, device=dev)
    dgap = ops.se_bwd(se_part, n, h * w, c, se_wa.detach(), se_wb.detach(), se_wa.shape[0], ca, d_ca, dwa, dwb)
    if se_direct:
        ops.grad_done(se_wa)
        ops.grad_done(se_wb)
    P.dgap = p(dgap)
    P.TH, P.TW = ops.pick_tile_bwd_dx(ho, wo, o)          # pass C walks its pixel pairs in four colours: its own tile choice
    # C: generate weight gradient rows + dx
    # pass C: regenerate (81 MAC) + generate weight gradient (81) + dx (81) per (output pixel, channel) on the VALU; dG = W^T du on the MFMAs
    with ops._Timed(f"ly_rf3c_bwd_kernel<2, {o // 32}>", 2.0 * mo * 9 * c * o, xb + xr.element_size() * n * h * w * c, valu_flops=2.0 * mo * c * 243):
        L.check(L.lib().ly_rf3c_bwd(ctypes.byref(P), 2, st), "ly_rf3c_bwd C")
    dbias = None if ops.grad_target(ctx.conv_b_param) is not None else torch.zeros_like(bias)      # BN removes the batch mean: d/dbias = 0
    tgw = ops.grad_target(ctx.gen_w_param)
    if tgw is not None and tgw.is_contiguous() and tgw.numel() == dwg.shape[1]:
        ops.sum_rows(dwg, out=tgw.view(-1), accumulate=True)            # the image rows of d(generate.0.weight) added straight into the sink
        ops.grad_done(ctx.gen_w_param)
        dgw = None
    else:
        dgw = ops.sum_rows(dwg).view(gen_w.shape)
    return (None, dx if need_dx else None, None if se_direct else dwa, None if se_direct else dwb, dgw, ct(dgg_tc), ct(dbg_tc),
            (None if t18 is not None else dw18.view(getw.shape)), dwc, dbias, dgo, dbo)


RfcbamFn._backward_rc = staticmethod(_rfcbam_backward_rc)


def _rfcbam_backward_k1(ctx, du, dgo, dbo):
    """RFCBAMConv kernel_size 1 backward on the fused recompute passes (csrc/ly_rf1_bwd.hip): G = relu(bn(gw*x)) is recomputed from x in
    every pass, nothing the size of the input is stored except cd (the conv weight gradient's operand) and dx.  du: gradient of the conv
    output (after the output BatchNorm / ReLU backward)."""
    xr, ca, gen_w, getw, conv_w, bias, ag, bg, gmean_tc, ginv_tc, es, t, omean, oinv, mm, rfa, se_part, u = ctx.saved_tensors
    n, c, h, w, k, s, o, ho, wo, ld = ctx.geom
    dev, dt = xr.device, xr.dtype
    L = _lib()
    st, p = L.stream_ptr(), L.ptr
    mo = n * h * w
    pl = ops.planes_of(xr)
    # d(loss)/d(conv input): du . Wc^T, Wc^T read in place from conv.0.weight
    dcd = torch.empty((mo, c), dtype=dt, device=dev)
    wct = pack.packed(pack.Src(ctx.conv_w_param, c, nrb=c, sra=1, srb=1, nc=o, sc=c), o, pl)
    ops.gemm(M=mo, H=h, W=w, K=o, N=c, a0=du, lda0=o, k0=o, wp=wct, out=dcd, ldo=c)
    gw = gen_w.detach().float().reshape(c).contiguous()
    cd = torch.empty((mo, c), dtype=dt, device=dev)
    d_ca64 = ops.zeros_f64(ca.numel(), dev)                 # double accumulators: d_ca feeds dx through SE's backward
    d_rfa = torch.empty(rfa.numel(), dtype=torch.float32, device=dev)
    gmax = torch.empty(rfa.numel(), dtype=torch.float32, device=dev)
    dx = ops.empty_nhwc(n, c, h, w, xr) if ctx.needs_input_grad[1] else ops.empty_nhwc(n, c, h, w, xr)
    tgw = ops.grad_target(ctx.gen_w_param) if getattr(ctx, "gen_w_param", None) is not None else None
    tgw = tgw if tgw is not None and tgw.is_contiguous() else None
    dgw_defer = tgw is not None and ops.small_grads_ok()
    dgw_fresh64 = tgw is None and ops.DETERMINISTIC_SMALL_GRADS         # no sink: float64 scratch, rounded after pass C (returned to autograd)
    dgw = ops.small_grad_scratch(tgw, ctx.gen_w_param) if dgw_defer else tgw.view(-1) if tgw is not None else \
        ops.zeros_f64(c, dev) if dgw_fresh64 else torch.zeros(c, dtype=torch.float32, device=dev)
    P = L.LyRf1BwdParams(n, h * w, c, p(xr), ld, p(dcd), p(gw), p(ag), p(bg), p(ca), p(rfa), p(cd), p(d_rfa), p(gmax), p(d_ca64),
                         p(gmax), None, None, None, None, None, None, 1.0 / (h * w), p(dx), c, p(dgw), L.dtype_code(xr), int(dgw_defer or dgw_fresh64))
    es1 = xr.element_size() * mo * c
    tn = ops._tname(xr)
    with ops._Timed(f"ly_rf1_bwd_kernel<{tn}, 0>", 6.0 * mo * c, 3.0 * es1):
        L.check(L.lib().ly_rf1_bwd(ctypes.byref(P), 0, st), "ly_rf1_bwd A")
    d_ca = d_ca64                                            # read as doubles by ly_se_bwd
    # conv weight gradient from cd
    tgt = ops.grad_target(ctx.conv_w_param)
    if tgt is not None:
        ops.wgrad(M=mo, H=h, W=w, N=o, du=du, lddu=o, x=cd, ldx=c, Hin=h, Win=w, Cin=c, dw=tgt, lddw=c)
        ops.grad_done(ctx.conv_w_param)
        dwc = None
    else:
        dwc = torch.zeros(o, c, dtype=torch.float32, device=dev)
        ops.wgrad(M=mo, H=h, W=w, N=o, du=du, lddu=o, x=cd, ldx=c, Hin=h, Win=w, Cin=c, dw=dwc, lddw=c)
        dwc = dwc.view(conv_w.shape)
    # get_weight + sigmoid
    w18 = getw.detach().float().reshape(18).contiguous()
    d_mm = torch.empty_like(mm)
    t18 = ops.grad_target(ctx.getw_param)
    t18 = t18 if t18 is not None and t18.is_contiguous() else None
    d18 = t18 is not None and ops.small_grads_ok()          # deferred: a float64 scratch, rounded into the sink when the backward pass ends
    f18 = t18 is None and ops.DETERMINISTIC_SMALL_GRADS          # no sink: float64 scratch, rounded right away (returned to autograd)
    dw18 = ops.small_grad_scratch(t18, ctx.getw_param) if d18 else t18.view(-1) if t18 is not None else \
        ops.zeros_f64(18, dev) if f18 else torch.zeros(18, dtype=torch.float32, device=dev)
    L.check(L.lib().ly_rfa_bwd(p(d_rfa), p(rfa), p(mm), p(w18), n, h, w, p(d_mm), p(dw18), int(d18 or f18), st), "ly_rfa_bwd")
    if f18:
        dw18 = ops.f64_round([dw18], [(18,)])[0]
    if t18 is not None and not d18:
        ops.grad_done(ctx.getw_param)
    # BatchNorm sums of the generate BatchNorm (double accumulators)
    sums = ops.new_stats(c, dev)
    P.d_mm, P.sums = p(d_mm), p(sums)
    with ops._Timed(f"ly_rf1_bwd_kernel<{tn}, 1>", 8.0 * mo * c, 2.0 * es1):
        L.check(L.lib().ly_rf1_bwd(ctypes.byref(P), 1, st), "ly_rf1_bwd B")
    tgg, tgb = (ops.grad_target(q) for q in ctx.gen_bn_params)
    bn_direct = tgg is not None and tgb is not None and tgg.numel() == c and tgb.numel() == c and tgg.is_contiguous() and tgb.is_contiguous()
    dgg, dbg, alpha, kappa, lam = ops.bn_bwd_coeffs(sums, c, mo, ag, gmean_tc, ginv_tc, True, dgamma=tgg if bn_direct else None,
                                                    dbeta=tgb if bn_direct else None)
    if bn_direct:
        ops.grad_done(ctx.gen_bn_params[0])
        ops.grad_done(ctx.gen_bn_params[1])
    # SE backward: parameter gradients, and d/d(mean x), which pass C adds while it writes dx
    se_wa, se_wb = ctx.se_params
    ta, tb = ops.grad_target(se_wa), ops.grad_target(se_wb)
    se_direct = ta is not None and tb is not None
    dwa = ta if se_direct else torch.zeros(se_wa.shape, dtype=torch.float32, device=dev)
    dwb = tb if se_direct else torch.zeros(se_wb.shape, dtype=torch.float32, device=dev)
    dgap = ops.se_bwd(se_part, n, h * w, c, se_wa.detach(), se_wb.detach(), se_wa.shape[0], ca, d_ca, dwa, dwb)
    if se_direct:
        ops.grad_done(se_wa)
        ops.grad_done(se_wb)
    P.alpha, P.kappa, P.lambda_, P.dgap = p(alpha), p(kappa), p(lam), p(dgap)
    with ops._Timed(f"ly_rf1_bwd_kernel<{tn}, 2>", 10.0 * mo * c, 3.0 * es1):
        L.check(L.lib().ly_rf1_bwd(ctypes.byref(P), 2, st), "ly_rf1_bwd C")
    if dgw_fresh64:
        dgw = ops.f64_round([dgw], [(c,)])[0]
    if tgw is not None and not dgw_defer:
        ops.grad_done(ctx.gen_w_param)
    dbias = None if ops.grad_target(ctx.conv_b_param) is not None else torch.zeros_like(bias)      # BN removes the batch mean: d/dbias = 0
    return (None, dx if ctx.needs_input_grad[1] else None, None if se_direct else dwa, None if se_direct else dwb,
            None if tgw is not None else dgw.view(gen_w.shape), dgg, dbg, (None if t18 is not None else dw18.view(getw.shape)), dwc, dbias, dgo, dbo)


RfcbamFn._backward_k1 = staticmethod(_rfcbam_backward_k1)
RF1_BWD = True         # tools: False keeps the first-generation k = 1 backward
RF3S_BWD = True        # tools: False keeps the thread = channel attention / ReLU passes of the streamed k = 3 backward
RC_BWD = True          # tools: False keeps the first-generation backward behind the lane = channel forward
RC_BWD_WIDTHS = (64, 128)           # output widths on the recompute passes (see RfcbamFn.backward); a module constant: tools / tests monkeypatch it
def rfcbam_train(mod, x):
    """RFCBAMConv.forward in training: one autograd node (SE, generate BatchNorm, attention maps, contraction)."""
    g, cv = mod.generate, mod.conv
    return RfcbamFn.apply(mod, x, mod.se.fc[0].weight, mod.se.fc[2].weight, g[0].weight, g[1].weight, g[1].bias, mod.get_weight[0].weight, cv[0].weight, cv[0].bias,
                          cv[1].weight, cv[1].bias)
