"""Thin host wrappers over the C-ABI (capi.py): torch supplies device memory and the current stream,
every arithmetic step is a HIP kernel in libleadyolo_hip.so.  Activations are logical NCHW tensors
with channels_last (NHWC) storage; a channel slice of such a tensor (a slot of a concat buffer) is
addressed as rows of stride `ld` without copying.

Storage dtype: every wrapper takes it from its activation argument — float32 (bf16x3 products) or bfloat16 (plain bf16
products, BASELINE configs[2]-[4]); outputs have the input's dtype, statistics / attention tables / weight gradients
are float32.  `edge_in` is the dtype policy at a module's edge (autocast, .half())."""
import ctypes
import os

import torch

from . import capi
from .capi import (ACT_NONE, ACT_RELU, ACT_SILU, GATHER_PATCH, GATHER_PATCH_NCHW, GATHER_PATCH_NCHW_BF16, GATHER_PATCH_NCHW_F16, GATHER_PATCH_NCHW_U8, GATHER_ROWS, GATHER_UP2,  # noqa: F401
                   PRO_AFFINE_RELU_CA, PRO_GATE, PRO_NONE)


# ---- optional per-launch timing (bench.py / tools): PROFILE = [] turns it on -----------------------
PROFILE = None


class _Timed:
    """Brackets one C-ABI call with events on the launch stream and records (kernel name as rocprofv3
    prints it, algorithmic MFMA flops, algorithmic bytes, algorithmic VALU flops).  `flops` = 2*MAC of the contractions that run on
    the matrix cores; `valu_flops` = 2*MAC of work that is vector arithmetic by construction (the RFCBAM `generate` regeneration:
    81 MAC per (output pixel, channel) and pass; the two are priced against different peaks)."""

    def __init__(self, name, flops, nbytes, valu_flops=0.0):
        self.rec = None
        if PROFILE is not None:
            self.rec = [name, float(flops), float(nbytes), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), float(valu_flops)]

    def __enter__(self):
        if self.rec is not None:
            self.rec[3].record()

    def __exit__(self, *a):
        if self.rec is not None:
            self.rec[4].record()
            PROFILE.append(self.rec)


def gemm_config(n_out):
    """mirror of the tile heuristic in csrc/ly_gemm.hip (ly_gemm_fwd)"""
    return (4, 2, 4) if n_out > 64 else (4, 1, 4) if n_out > 32 else (2, 2, 1)


_CONV_TILE_CACHE = {}


def pick_conv_tile(h, w, max_px=128, max_frame=192):
    """TH x TW patch (<= 128 pixels, frame (TH+2)(TW+2) <= 192) covering an h x w map with the fewest
    MFMA pixel tiles, ties broken by the smaller halo frame."""
    key = (h, w)
    if key not in _CONV_TILE_CACHE:
        best = None
        for tw in range(1, min(w, max_px) + 1):
            for th in range(1, min(h, max_px // tw) + 1):
                if (th + 2) * (tw + 2) > max_frame:
                    continue
                blocks = -(-h // th) * -(-w // tw)
                work = blocks * (-(-(th * tw) // 16))               # MFMA pixel tiles issued
                halo = blocks * (th + 2) * (tw + 2)                 # frame positions staged
                score = (work * 16 + 0.35 * halo, blocks)
                if best is None or score < best[0]:
                    best = (score, th, tw)
        _CONV_TILE_CACHE[key] = (best[1], best[2])
    return _CONV_TILE_CACHE[key]


MLP_WIDTHS = (16, 24, 40, 80, 160, 320)     # C values ly_mlpblock_fwd is instantiated for (include/lead_yolo_hip.h)


def mlp_config(c, m, w, bf16=False):
    """(C, NT, HT, T2D) of the kernel ly_mlpblock_fwd launches -- mirrors dispatch_nt in csrc/ly_mlpblock.hip"""
    ht = {16: 2, 24: 4, 40: 2, 80: 2, 160: 4, 320: 4}[c]
    ntmax = {16: 4, 24: 4, 40: 4, 80: 2, 160: 2, 320: 1}[c]
    if c >= 80:
        if c == 80 and bf16 and m >= 384 * 256:
            return c, 4, ht, "false"
        return c, (2 if ntmax >= 2 and m >= 200 * 128 else 1), ht, "false"
    if w % 16 == 0 and w >= 64 and ntmax >= 2:
        return c, 2, ht, "true"
    nt = 4 if (ntmax >= 4 and m >= 256 * 1024) else 2 if (ntmax >= 2 and m >= 128 * 512) else 1
    return c, nt, ht, "false"


def require_cuda(x, who, mod=None):
    if mod is not None and not mod.training and torch.is_grad_enabled() and x.requires_grad:
        raise NotImplementedError(f"{who}: backward is built for train mode (batch-statistics BatchNorm); an eval-mode module would return "
                                  "tensors detached from autograd — call .train(), or run inference under torch.no_grad()")
    if not x.is_cuda:
        raise RuntimeError(f"{who}: the HIP path needs a CUDA/ROCm tensor (got {x.device}); there is no CPU fallback")
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise NotImplementedError(f"{who}: activations must be float32 or bfloat16 here (got {x.dtype}); float16 is converted at the "
                                  "module edge (ops.edge_in)")


def edge_in(x, who, mod=None):
    """Dtype policy at a module's edge (SURVEY §8b call contract: fp32, or fp16 under `autocast` / `.half()`, train.py:229,316).
      * under torch.autocast('cuda', dtype) a float32 input is cast to the autocast dtype, as autocast does for convolutions;
      * bfloat16 runs the bf16 kernels; float32 (outside autocast) the fp32-storage kernels;
      * float16 (the reference's AMP dtype) has no kernels of its own: it is computed as bfloat16 (same 16-bit storage, wider
        exponent, fp32 accumulation) and the module's output is cast back to float16, so neighbours still see fp16.
    Returns (tensor to compute on, dtype to cast the output back to or None)."""
    if not isinstance(x, torch.Tensor):
        return x, None
    if not x.is_cuda:
        raise RuntimeError(f"{who}: the HIP path needs a CUDA/ROCm tensor (got {x.device}); there is no CPU fallback")
    want = x.dtype
    if torch.is_autocast_enabled("cuda") and x.dtype == torch.float32:
        want = torch.get_autocast_dtype("cuda")
    if want not in (torch.float32, torch.bfloat16, torch.float16):
        raise NotImplementedError(f"{who}: unsupported activation dtype {x.dtype}")
    compute = torch.bfloat16 if want == torch.float16 else want
    back = torch.float16 if (want == torch.float16 and x.dtype != torch.bfloat16) else None
    if x.dtype != compute:
        x = x.to(compute)
    require_cuda(x, who, mod)
    return x, back


def edge_out(y, back):
    if back is None:
        return y
    if isinstance(y, torch.Tensor):
        return y.to(back)
    if isinstance(y, (list, tuple)):
        return type(y)(edge_out(t, back) for t in y)
    return y


def planes_of(t):
    """operand planes of the weight packs a kernel call on tensor / dtype `t` needs: 2 (bf16x3, fp32 storage) or 1 (bf16 storage)"""
    d = t if isinstance(t, torch.dtype) else t.dtype
    return 1 if d == torch.bfloat16 else 2


def vw_of(t):
    """elements of one 16-byte vector: channel counts / leading dimensions of contraction sources must be multiples of it"""
    d = t if isinstance(t, torch.dtype) else t.dtype
    return 8 if d == torch.bfloat16 else 4


def nhwc(x):
    return x.contiguous(memory_format=torch.channels_last)


def empty_nhwc(n, c, h, w, like, dtype=None):
    return torch.empty((n, c, h, w), dtype=dtype or like.dtype, device=like.device, memory_format=torch.channels_last)


def rows(x):
    """(tensor, ld): x as an [n*h*w, c] row matrix with row stride ld floats, zero-copy when x is a
    channels_last tensor or a channel slice of one; otherwise a channels_last copy is made."""
    n, c, h, w = x.shape
    if x.is_contiguous(memory_format=torch.channels_last) and x.stride(1) == 1:
        return x, c
    sn, sc, sh, sw = x.stride()
    if sc == 1 and sw >= c and sw % vw_of(x) == 0 and (h == 1 or sh == w * sw) and (n == 1 or sn == h * w * sw) and w > 1 \
            and x.data_ptr() % 16 == 0:
        return x, sw
    x = nhwc(x)
    if x.stride(1) != 1:        # degenerate shapes where torch reports ambiguous strides
        x = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    return x, c


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def gemm(*, M, H, W, K, N, a0, lda0, k0, wp, out, ldo, a1=None, lda1=0, gather=GATHER_ROWS, Hin=0, Win=0, Cin=0, ks=0, pk=0,
         pro=PRO_NONE, g_h=None, g_w=None, res=None, ldres=0, p_scale=None, p_shift=None, p_ca=None, e_scale=None,
         e_shift=None, rowscale=None, act=ACT_NONE, stats=None, dtype=None, scat_ks=0, scat_c=0, eadd=None, ldeadd=0):
    # element type of the call: the output's (statistics passes: the source's); the NCHW image gather always READS fp32 — or uint8
    # (pixel / 255 on load: the training loop's `imgs.float() / 255` folded into the gather)
    # or a 16-bit image as it is (the `im.half()` batch of a reduced-precision forward; LY_BF16 calls)
    if gather == GATHER_PATCH_NCHW:
        gather = {torch.uint8: GATHER_PATCH_NCHW_U8, torch.bfloat16: GATHER_PATCH_NCHW_BF16, torch.float16: GATHER_PATCH_NCHW_F16}.get(a0.dtype, gather)
    image = gather in (GATHER_PATCH_NCHW, GATHER_PATCH_NCHW_U8, GATHER_PATCH_NCHW_BF16, GATHER_PATCH_NCHW_F16)
    dt = (out if out is not None else a0).dtype if not image or out is not None else torch.float32
    if dtype is not None:
        dt = dtype
    code = capi.dtype_code(dt)
    P = capi.LyGemmParams(M, H, W, K, N, _p(a0), lda0, k0, _p(a1), lda1, gather, Hin, Win, Cin, ks, pk, pro, _p(g_h), _p(g_w),
                          _p(res), ldres, _p(p_scale), _p(p_shift), _p(p_ca), _p(wp), _p(e_scale), _p(e_shift), _p(rowscale),
                          act, _p(out), ldo, _p(stats), code, scat_ks, scat_c, _p(eadd), ldeadd)
    if eadd is not None and eadd.dtype != out.dtype:
        raise ValueError("gemm: eadd must have the output's dtype")
    nt, mt, wc = gemm_config(N)
    ti = {GATHER_PATCH_NCHW_U8: "unsigned char", GATHER_PATCH_NCHW_BF16: "ly_bf16img", GATHER_PATCH_NCHW_F16: "ly_f16img"}.get(
        gather, "float" if (code == 0 or image) else "__bf16")
    to = "float" if code == 0 else "__bf16"
    es_i, es_o = {"float": 4, "__bf16": 2, "unsigned char": 1, "ly_bf16img": 2, "ly_f16img": 2}[ti], (4 if to == "float" else 2)
    kgather = GATHER_PATCH_NCHW if image else gather        # the kernel template's gather code (the uint8 image differs by TI only)
    # the variant launch_gemm_v (csrc/ly_gemm.hpp) picks: resident weights for K in one / two chunks + the branch-free epilogue
    nchunk = -(-K // (128 if ti == "__bf16" else 64))
    nch = 0
    if N % 4 == 0 and ldo % 4 == 0 and out is not None:
        ok1 = wc == 4 or pro == 0
        ok2 = wc == 4 and pro != PRO_GATE
        if stats is not None:
            ok1, ok2 = ok1 and pro != PRO_GATE, ok2 and pro == 0
        nch = 1 if (nchunk == 1 and ok1) else 2 if (nchunk == 2 and ok2) else 0
    fast = N % 4 == 0 and ldo % 4 == 0 and out is not None
    ep = (2 if stats is not None else 1) if (nch or (fast and pro == 0)) else 1 if (fast and pro == PRO_GATE and stats is None) else 0
    if scat_ks or eadd is not None:                         # the scatter / add-before-store epilogues (launch_gemm_v)
        ep = 4 if eadd is not None else 3
    name = f"ly_gemm_kernel_d2<{ti}, {to}, {nt}, {mt}, {wc}, {kgather}, {pro}, {nch}, {ep}>"
    if (image and Cin == 3 and K == 48 and 20 <= N <= 80 and N % 4 == 0 and (out is None or ldo % 4 == 0) and Hin == 4 * H and Win == 4 * W
            and (code != 0 or gather in (GATHER_PATCH_NCHW, GATHER_PATCH_NCHW_U8))):
        # ly_patch4_try (csrc/ly_patch4.hip): PatchEmbed on an RGB image has its own LDS-free kernel
        name = f"ly_patch4_kernel<{ti}, {to}, {max(2, -(-N // 16))}, {'true' if stats is not None else 'false'}>"
    with _Timed(name, 2.0 * M * K * N, es_i * M * K + (es_o * M * N if out is not None else 0) + 4.0 * N * K):
        capi.check(capi.lib().ly_gemm_fwd(ctypes.byref(P), capi.stream_ptr()), "ly_gemm_fwd")


def mlpblock_pconv(x, z, n, h, w, c, wp):
    """z = [conv3x3(x[:, :c/4]; wp) | x[:, c/4:]] in one pass (persistent MLPBlock kernel, partial conv only); x, z dense NHWC [n, c, h, w].
    Returns False when the kernel is not built for this shape (the caller then copies and calls conv3x3)."""
    with _Timed(f"ly_mlpblock_persist_kernel<{_tname(x)}, {c}, pconv>", 18.0 * n * h * w * (c // 4) ** 2, 2.0 * x.element_size() * n * h * w * c):
        rc = capi.lib().ly_mlpblock_pconv(_p(x), _p(z), n, h, w, c, _p(wp), capi.dtype_code(x), capi.stream_ptr())
    if rc < 0:
        capi.check(rc, "ly_mlpblock_pconv")
    return rc == 0


def conv3x3(*, M, H, W, Cin, N, x, ldx, wp, out, ldo, e_scale=None, e_shift=None, act=ACT_NONE, stats=None):
    th, tw = pick_conv_tile(H, W)
    code = capi.dtype_code(x)
    P = capi.LyConv3Params(M, H, W, Cin, N, th, tw, _p(x), ldx, _p(wp), _p(e_scale), _p(e_shift), act, _p(out), ldo, _p(stats), code)
    mt, wc = (2, 4) if N > 64 else (2, 2)
    ntv = -(-(th * tw) // 16)                               # active pixel tiles of a patch: compile-time in the bf16 instantiations (conv3_dispatch)
    nta = 0 if code == 0 else (ntv if (wc == 4 and ntv in (5, 8)) else 4 if (wc == 2 and ntv == 8) else 0)
    name = f"ly_conv3x3_kernel<{_tname(x)}, {mt}, {wc}, {nta}>"
    if code != 0 and N > 64 and (M // (H * W)) * -(-W // tw) * -(-H // th) * -(-N // 128) < 512:
        name = "ly_conv3x3_lat_kernel<__bf16, 2, 2>"        # conv3_dispatch (csrc/ly_conv3x3.hip): grids under two blocks per CU take the latency form
    with _Timed(name, 2.0 * M * 9 * Cin * N, x.element_size() * M * (Cin + N) + 4.0 * 9 * Cin * N):
        capi.check(capi.lib().ly_conv3x3_fwd(ctypes.byref(P), capi.stream_ptr()), "ly_conv3x3_fwd")


def _tname(t):
    return "float" if t.dtype == torch.float32 else "__bf16"


def pool_hw(x, ldx, n, h, w, c):
    pool = torch.empty((n, h + w, c), dtype=torch.float32, device=x.device)
    with _Timed(f"ly_pool_hw_kernel<{_tname(x)}>", 1.0 * n * h * w * c, x.element_size() * n * h * w * c):
        capi.check(capi.lib().ly_pool_hw(_p(x), ldx, n, h, w, c, _p(pool), capi.dtype_code(x), capi.stream_ptr()), "ly_pool_hw")
    return pool


def coordatt_mlp(pool, n, h, w, c, mip, w1, b1, wh, bh, ww, bw, sc=None, sh=None):
    """sc/sh: bn1 as a per-channel affine applied after the raw conv1 (train mode); None = already folded into w1/b1"""
    a_h = torch.empty((n, h, c), dtype=torch.float32, device=pool.device)
    a_w = torch.empty((n, w, c), dtype=torch.float32, device=pool.device)
    capi.check(capi.lib().ly_coordatt_mlp(_p(pool), n, h, w, c, mip, _p(w1), _p(b1), _p(sc), _p(sh), _p(wh), _p(bh), _p(ww), _p(bw), _p(a_h),
                                          _p(a_w), capi.stream_ptr()), "ly_coordatt_mlp")
    return a_h, a_w


def coordatt_mlp_bwd(pool, n, h, w, c, mip, w1, b1, mean, invstd, gamma, beta, wh, ww, a_h, a_w, da_h, da_w, targets):
    """targets: (dw1, dgamma, dbeta, dwh, dbh, dww, dbw) buffers that are ADDED to — all fp32, or all float64 scratches (small_grad_scratch);
    returns dpool [n, h+w, c]"""
    r = n * (h + w)
    sums = zeros_f64(32 * 2 * mip, pool.device)                        # striped (sum dy1, sum dy1*xh), double accumulators
    ws = torch.empty(r * 3 * mip, dtype=torch.float32, device=pool.device)
    dpool = torch.empty((n, h + w, c), dtype=torch.float32, device=pool.device)
    with _Timed("ly_coordatt_mlp_bwd1_kernel + bwd2", 8.0 * r * c * mip, 4.0 * 4 * r * c):
        capi.check(capi.lib().ly_coordatt_mlp_bwd(_p(pool), n, h, w, c, mip, _p(w1), _p(b1), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(wh), _p(ww),
                                                  _p(a_h), _p(a_w), _p(da_h), _p(da_w), _p(ws), _p(sums), _p(dpool),
                                                  *[_p(t) for t in targets], int(targets[0].dtype == torch.float64), capi.stream_ptr()),
                   "ly_coordatt_mlp_bwd")
    return dpool


def coordatt_gate(x, ldx, n, h, w, c, a_h, a_w, res=None, ldres=0):
    out = empty_nhwc(n, c, h, w, x)
    with _Timed(f"ly_gate_kernel<{_tname(x)}>", 2.0 * n * h * w * c, 2.0 * x.element_size() * n * h * w * c):
        capi.check(capi.lib().ly_coordatt_gate(_p(x), ldx, n, h, w, c, _p(a_h), _p(a_w), _p(res), ldres, _p(out), c, capi.dtype_code(x),
                                               capi.stream_ptr()), "ly_coordatt_gate")
    return out


def se_attention(x, ldx, n, hw, c, wa, wb, r, want_part=False):
    slices = se_slices(hw)                       # short per-block row loops: the pass is latency-bound, not bandwidth-bound
    part = torch.empty((n, slices, c), dtype=torch.float32, device=x.device)
    ca = torch.empty((n, c), dtype=torch.float32, device=x.device)
    with _Timed(f"ly_colsum_kernel<{_tname(x)}> + ly_se_mlp_kernel", 1.0 * n * hw * c, x.element_size() * n * hw * c):
        capi.check(capi.lib().ly_se_fwd(_p(x), ldx, n, hw, c, _p(wa), _p(wb), r, _p(part), slices, _p(ca), capi.dtype_code(x), capi.stream_ptr()),
                   "ly_se_fwd")
    return (ca, part) if want_part else ca


def se_bwd(part, n, hw, c, wa, wb, r, ca, d_ca, dwa, dwb):
    """SE backward: dwa [r, c] / dwb [c, r] are ADDED to; returns dgap [n, c] = d/d(mean x).  d_ca: float32, or the float64 accumulators
    the RFCBAM backward kernels sum it in (read as they are)"""
    if d_ca.dtype not in (torch.float32, torch.float64) or d_ca.numel() != n * c or not d_ca.is_contiguous():
        raise ValueError("se_bwd: d_ca must be a contiguous float32 / float64 [n, c]")
    dgap = torch.empty((n, c), dtype=torch.float32, device=part.device)
    ws = torch.empty(n * (2 * c + 2 * r), dtype=torch.float32, device=part.device)
    capi.check(capi.lib().ly_se_bwd(_p(part), part.shape[1], n, hw, c, _p(wa), _p(wb), r, _p(ca), _p(d_ca), int(d_ca.dtype == torch.float64), _p(dwa), _p(dwb), _p(dgap), _p(ws),
                                    capi.stream_ptr()), "ly_se_bwd")
    return dgap


def se_slices(hw):
    """pixel slices of the SE pooling partials: a function of the MAP only, so results do not change with the batch split"""
    return max(1, min(hw // 64, 128))


def colsum(x, ldx, n, hw, c):
    """SE pooling partials part[n][slice][c] (one pass over x)"""
    slices = se_slices(hw)
    part = torch.empty((n, slices, c), dtype=torch.float32, device=x.device)
    with _Timed(f"ly_colsum_kernel<{_tname(x)}>", 1.0 * n * hw * c, x.element_size() * n * hw * c):
        capi.check(capi.lib().ly_colsum(_p(x), ldx, n, hw, c, _p(part), slices, capi.dtype_code(x), capi.stream_ptr()), "ly_colsum")
    return part


def pick_tile(ho, wo):
    """TH x TW (<= 64 output pixels, one per lane) minimising idle lanes over the Ho x Wo map."""
    best = None
    for nct in range(1, 9):
        tw = -(-wo // nct)
        if tw > 64:
            continue
        th = max(1, 64 // tw)
        th = min(th, ho)
        nrt = -(-ho // th)
        eff = (ho * wo) / (nct * nrt * 64.0)
        if best is None or eff > best[0] + 1e-9:
            best = (eff, th, tw)
    if best is None:
        return 1, 64
    return best[1], best[2]


def rf3c_ok(c, s):
    """the lane = channel RFCBAMConv k=3 kernels (csrc/ly_rf3c.hip) cover C % 32 == 0 at stride 1 / 2"""
    return c % 32 == 0 and s in (1, 2)


RF3M_MIN_UNITS = 1500      # wave tiles x output-channel blocks below which the lane = channel kernels are faster (measured, MI355X, module time in us,
                           # rf3c / rf3m: layer 17 bs 16: 79 / 89, bs 32: 135 / 101, bs 64: 227 / 185; layer 20 bs 32: 110 / 144, bs 64: 165 / 155 — a block of
                           # ly_rf3m walks ALL input channels of its 256 pixels serially, ~45 us however small the launch)


CONCURRENT_PARTS = 1       # graph.GraphedForward runs the batch as this many sub-batches side by side: size-based dispatch decisions look at the
                           # whole batch, so that a sub-batched replay takes the kernels (and returns the bits) of the eager forward of the full batch


def rf3m_ok(x, c, o, s, n=None, ho=None, wo=None):
    """`generate` on the matrix cores (csrc/ly_rf3m.hip): bf16 storage, C % 32 == 0, O % 64 == 0, stride 1 / 2 — and, when the problem size
    is given, enough wave tiles to fill the chip (RF3M_MIN_UNITS)"""
    if not ((x if isinstance(x, torch.dtype) else x.dtype) == torch.bfloat16 and c % 32 == 0 and o % 64 == 0 and s in (1, 2)):
        return False
    if n is None:
        return True
    th, tw = pick_tile_m(ho, wo, s)
    return n * CONCURRENT_PARTS * -(-ho // th) * -(-wo // tw) * (o // (128 if o % 128 == 0 else 64)) >= RF3M_MIN_UNITS


def pick_tile_m(ho, wo, s):
    """TH x TW wave tile of csrc/ly_rf3m.hip: at most 32 output pixels reading at most 160 input positions; fewest tiles, then smallest halo"""
    best = None
    for tw in range(1, 33):
        for th in range(1, 32 // tw + 1):
            pos = (s * (th - 1) + 3) * (s * (tw - 1) + 3)
            if pos > 160 or th > ho or tw > max(wo, 1) * 2:
                continue
            key = (-(-ho // th) * -(-wo // tw), pos)
            if best is None or key < best[0]:
                best = (key, th, tw)
    return best[1], best[2]


def rf3m_stats(x, ldx, n, h, w, c, s, wst, th, tw, gap=True):
    """ONE pass over x (csrc/ly_rf3m.hip): (mm [n, 3ho, 3wo, 2], part [n, tiles, c])"""
    ho, wo = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
    mm = torch.empty((n, 3 * ho, 3 * wo, 2), dtype=torch.float32, device=x.device)
    tiles = -(-ho // th) * -(-wo // tw)
    part = torch.empty((n, tiles, c), dtype=torch.float32, device=x.device) if gap else None
    # generate on the MFMAs: 3 + 3 products of 32 x 32 x 16 per 4-channel group and 32 pixels (3/4 of them multiply structural zeros)
    with _Timed("ly_rf3m_stats_kernel", 2.0 * n * ho * wo * (c // 4) * 6 * 32 * 16, x.element_size() * n * h * w * c + 4.0 * 18 * n * ho * wo):
        capi.check(capi.lib().ly_rf3m_stats(_p(x), ldx, n, h, w, c, s, _p(wst), th, tw, _p(mm), _p(part), tiles, capi.stream_ptr()), "ly_rf3m_stats")
    return mm, part


def rf3m_fwd(*, n, h, w, c, ho, wo, N, s, th, tw, x, ldx, ca, rfa, wp, e_scale, e_shift, out, ldo):
    P = capi.LyRfcbam3Params(n, h, w, c, ho, wo, N, s, th, tw, _p(x), ldx, None, _p(ca), _p(rfa), _p(wp), _p(e_scale),
                             _p(e_shift), _p(out), ldo, None, 0, capi.dtype_code(x))
    mo = n * ho * wo
    mt = 4 if N % 128 == 0 else 2
    with _Timed(f"ly_rf3m_fwd_kernel<{mt}>", 2.0 * mo * 9 * c * N + 2.0 * mo * (N // (32 * mt)) * (c // 4) * 6 * 32 * 16,
                x.element_size() * (n * h * w * c + mo * N) + 2.0 * 9 * c * N):
        capi.check(capi.lib().ly_rf3m_fwd(ctypes.byref(P), capi.stream_ptr()), "ly_rf3m_fwd")


def pick_tile_c(ho, wo, s):
    """TH x TW tile of the lane = channel kernels: TW even, TH*TW <= 64, at most 320 input positions; fewest tiles, then smallest halo"""
    best = None
    for tw in range(2, 65, 2):
        th = min(64 // tw, ho)
        if th < 1:
            continue
        pos = (s * (th - 1) + 3) * (s * (tw - 1) + 3)
        if pos > 320:
            continue
        tiles = -(-ho // th) * -(-wo // tw)
        key = (tiles, pos)
        if best is None or key < best[0]:
            best = (key, th, tw)
    return best[1], best[2]


def pick_tile_bwd_dx(ho, wo, o):
    """tile of the recompute backward's dx pass (csrc/ly_rf3c_bwd.hip pass C): its pixel pairs are walked in four colours (row parity x pair-column
    parity) of ceil(count / 8) iterations each — fewest iteration slots over the map among the tiles whose LDS footprint fits, then smallest halo"""
    best = None
    for tw in range(2, 65, 2):
        th = min(64 // tw, ho)
        ih, iw = 2 * (th - 1) + 3, 2 * (tw - 1) + 3
        pos = ih * iw
        nct = -(-wo // tw)
        lds = 2 * pos * 128 + 288 * 128 + 64 * (2 * o + 16) + 32 * 9 * 8 * 4 + 256 + (ih + 2 * tw * nct + 2 + 27) * 128     # mirrors rb_launch
        if th < 1 or pos > 320 or lds > 160 * 1024:
            continue
        its = sum(-(-(((th + 1 - ry) // 2) * ((tw // 2 + 1 - rx) // 2)) // 8) for ry in (0, 1) for rx in (0, 1))
        key = (-(-ho // th) * nct * its, pos)
        if best is None or key < best[0]:
            best = (key, th, tw)
    return best[1], best[2]


def rf3c_stats(x, ldx, n, h, w, c, s, wq, th, tw, gap=True, raw=False):
    """ONE pass over x: (mm [n, 3ho, 3wo, 2], part [n, tiles, c]) — the [max, mean] map of relu(bn(generate(x))) and SE's pooling partials"""
    ho, wo = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
    mm = torch.empty((n, 3 * ho, 3 * wo, 2), dtype=torch.float32, device=x.device)
    tiles = -(-ho // th) * -(-wo // tw)
    part = torch.empty((n, tiles, c), dtype=torch.float32, device=x.device) if gap else None
    with _Timed(f"ly_rf3c_stats_kernel<{_tname(x)}, {s}, {'true' if raw else 'false'}>", 0.0, x.element_size() * n * h * w * c + 4.0 * 18 * n * ho * wo, valu_flops=2.0 * n * ho * wo * c * 81):
        capi.check(capi.lib().ly_rf3c_stats(_p(x), ldx, n, h, w, c, s, _p(wq), int(raw), th, tw, _p(mm), _p(part), tiles, capi.dtype_code(x),
                                            capi.stream_ptr()), "ly_rf3c_stats")
    return mm, part


def rf3c_fwd(*, n, h, w, c, ho, wo, N, s, th, tw, x, ldx, wq, ca, rfa, wp, e_scale, e_shift, out, ldo, stats=None, linear=False, raw=False):
    P = capi.LyRfcbam3Params(n, h, w, c, ho, wo, N, s, th, tw, _p(x), ldx, None, _p(ca), _p(rfa), _p(wp), _p(e_scale),
                             _p(e_shift), _p(out), ldo, _p(stats), int(linear), capi.dtype_code(x))
    mo = n * ho * wo
    bf = x.dtype == torch.bfloat16
    cfg = "2, 8" if (N > 128 and bf) else ("2, 4" if N > 64 else "1, 4")         # mirrors rc_dispatch_fwd
    with _Timed(f"ly_rf3c_fwd_kernel<{_tname(x)}, {cfg}, {s}, {'true' if raw else 'false'}>", 2.0 * mo * 9 * c * N,
                x.element_size() * (n * h * w * c + mo * N) + 4.0 * 9 * c * N, valu_flops=2.0 * mo * 81 * c):
        capi.check(capi.lib().ly_rf3c_fwd(ctypes.byref(P), _p(wq), int(raw), capi.stream_ptr()), "ly_rf3c_fwd")


PRE1_PIXELS = 64           # pixels per block of the k = 1 statistics + pooling pass (ly_rfcbam_pre1v: 16 pixels per trip; measured 16 / 32 / 64 / 128:
                           # 19.5 / 16.9 / 16.7 / 16.2 us at 256 x 40 x 40 x 64, 10.5 / 10.4 / 10.4 / 11.1 at 160 x 20 x 20 x 64)


def rfcbam_stats(x, ldx, n, h, w, c, k, s, wg=None, a1=None, b1=None, th=1, tw=64, gap=False):
    """[max, mean] map of relu(bn(generate(x))); gap=True: the same pass also leaves the SE pooling partials -> (mm, part)"""
    ho, wo = ((h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1)
    mm = torch.empty((n, k * ho, k * wo, 2), dtype=torch.float32, device=x.device)
    part, slices = None, 0
    if gap:
        if k == 3:
            raise NotImplementedError("the fused statistics + pooling pass is built for k = 1 (for k = 3 the extra per-chunk LDS pass cost more "
                                      "than the separate ly_colsum launch it saved: measured 44 -> 69 us)")
        slices = max(1, min(h * w // PRE1_PIXELS, 128))    # a function of the map only: results do not change with the batch split
        part = torch.empty((n, slices, c), dtype=torch.float32, device=x.device)
    with _Timed((f"ly_rfcbam_pre1_kernel<{_tname(x)}>" if (gap and k == 1) else f"ly_rfcbam_stats{k}_kernel<{_tname(x)}>"), 0.0,
                x.element_size() * n * h * w * c + 4.0 * 2 * k * k * n * ho * wo, valu_flops=2.0 * n * ho * wo * c * (81 if k == 3 else 1)):
        capi.check(capi.lib().ly_rfcbam_stats(_p(x), ldx, n, h, w, c, k, s, _p(wg), _p(a1), _p(b1), th, tw, _p(mm), _p(part), slices,
                                              capi.dtype_code(x), capi.stream_ptr()), "ly_rfcbam_stats")
    return (mm, part) if gap else mm


def rfcbam_mid(part, hw, wa, wb, r, mm, w18):
    """SE's linears (from the pooling partials) + get_weight's 3x3 conv on the [max, mean] map, one launch -> (ca, rfa)"""
    n, slices, c = part.shape
    _, hk, wk, _ = mm.shape
    ca = torch.empty((n, c), dtype=torch.float32, device=part.device)
    rfa = torch.empty((n, hk, wk), dtype=torch.float32, device=mm.device)
    with _Timed("ly_rfcbam_mid_kernel", 36.0 * n * hk * wk, 12.0 * n * hk * wk + 4.0 * n * slices * c):
        capi.check(capi.lib().ly_rfcbam_mid(_p(part), slices, c, hw, _p(wa), _p(wb), r, _p(ca), n, _p(mm), hk, wk, _p(w18), _p(rfa), capi.stream_ptr()),
                   "ly_rfcbam_mid")
    return ca, rfa


def rfa_map(mm, w18):
    n, hk, wk, _ = mm.shape
    rfa = torch.empty((n, hk, wk), dtype=torch.float32, device=mm.device)
    with _Timed("ly_rfa_map_kernel", 36.0 * n * hk * wk, 12.0 * n * hk * wk):
        capi.check(capi.lib().ly_rfa_map(_p(mm), n, hk, wk, _p(w18), _p(rfa), capi.stream_ptr()), "ly_rfa_map")
    return rfa


def rfcbam3(*, n, h, w, c, ho, wo, N, s, th, tw, x, ldx, wg, ca, rfa, wp, e_scale, e_shift, out, ldo, stats=None, linear=False):
    P = capi.LyRfcbam3Params(n, h, w, c, ho, wo, N, s, th, tw, _p(x), ldx, _p(wg), _p(ca), _p(rfa), _p(wp), _p(e_scale),
                             _p(e_shift), _p(out), ldo, _p(stats), int(linear), capi.dtype_code(x))
    mt = 4 if N > 128 else (2 if N > 64 else 1)            # mirrors ly_rfcbam3_fwd
    mo = n * ho * wo
    sw = mt == 4 and n * -(-ho // th) * -(-wo // tw) * -(-N // 256) > 256    # mirrors launch_rf3: weights via the scalar cache
    with _Timed(f"ly_rfcbam3{'_sw' if sw else ''}_kernel<{_tname(x)}, {mt}>", 2.0 * mo * 9 * c * N,
                x.element_size() * (n * h * w * c + mo * N) + 4.0 * 9 * c * N, valu_flops=2.0 * mo * 81 * c):
        capi.check(capi.lib().ly_rfcbam3_fwd(ctypes.byref(P), capi.stream_ptr()), "ly_rfcbam3_fwd")


def sppf_pool_fits(h, w):
    return 2 * h * w * 4 * 4 <= 160 * 1024


def sppf_pool(x, ldx, n, h, w, c, k, out, ldo):
    with _Timed(f"ly_sppf_pool_kernel<{_tname(x)}>", 0.0, 5.0 * x.element_size() * n * h * w * c):
        capi.check(capi.lib().ly_sppf_pool(_p(x), ldx, n, h, w, c, k, _p(out), ldo, capi.dtype_code(x), capi.stream_ptr()), "ly_sppf_pool")


def detect_tail(y, ldy, n, h, w, na, no, anchors, stride, p, z, zrows, zoff):
    capi.check(capi.lib().ly_detect_tail(_p(y), ldy, n, h, w, na, no, _p(anchors), float(stride), _p(p), _p(z), zrows, zoff,
                                         capi.dtype_code(y), capi.stream_ptr()), "ly_detect_tail")


def detect_level_ok(k, na, no, dtype):
    return dtype in (torch.float32, torch.bfloat16) and bool(capi.lib().ly_detect_level_ok(k, na, no, capi.dtype_code(dtype)))


def detect_level(x, ldx, n, h, w, k, wp, bias, na, no, anchors, stride, p, z, zrows, zoff, nat=True):
    """head 1x1 convolution + decode of one Detect level in ONE launch (ly_detect_level): x rows [n*h*w, k] -> p [n, na, h, w, no], z rows
    (z None: raw maps only).  nat: wp from pack.frag_pack_nat, else the frag_pack3 layout (>= 32 rows)"""
    es = x.element_size()
    with _Timed(f"ly_detect_level_kernel<{_tname(x)}, {k // 32}>", 2.0 * n * h * w * k * 32, es * n * h * w * k + 8.0 * n * h * w * na * no):
        capi.check(capi.lib().ly_detect_level(_p(x), ldx, n, h, w, k, _p(wp), int(nat), _p(bias), na, no, _p(anchors), float(stride), _p(p), _p(z), zrows, zoff,
                                              capi.dtype_code(x), capi.stream_ptr()), "ly_detect_level")


def detect_head_bwd(dp, n, h, w, na, no, du, ldu, dbias):
    """dp fp32 [n, na, h, w, no] -> du rows [n*h*w, ldu] (columns >= na*no zero), dbias[na*no] += column sums (dbias fp32, or a float64
    scratch of small_grad_scratch)"""
    capi.check(capi.lib().ly_detect_head_bwd(_p(dp), n, h, w, na, no, _p(du), ldu, _p(dbias), int(dbias.dtype == torch.float64), capi.dtype_code(du),
                                             capi.stream_ptr()), "ly_detect_head_bwd")


# ---- small parameter-gradient reductions, reproducibly -------------------------------------------------------------------------------
# A few kernels add a handful of values from many blocks into a parameter gradient (Detect bias, get_weight's taps, the k = 1 generate weight,
# CoordAtt's MLP).  With float atomics those gradients moved by ~1e-6 from run to run — the only part of a training step that was not the same
# bits every time.  Inside a backward pass they accumulate into zeroed DOUBLE scratches instead (the arrival order then only moves the 53rd
# bit), and ONE ly_f64_add launch, queued on the autograd engine for the end of the pass, rounds every sum into its fp32 `.grad` storage.
class _SmallGrads:
    pending = []            # (scratch, flat fp32 target, parameter)
    task = -1               # autograd graph-task id the pending entries (and the queued end-of-pass callback) belong to
    announced = set()       # id(parameter) of the entries that already left in this pass (early flush under a gradient listener)
    nodes = {}              # id(scratch) -> id of the autograd node that took it


DETERMINISTIC_SMALL_GRADS = True


def _graph_task():
    task = getattr(torch._C, "_current_graph_task_id", None)
    return task() if task is not None else -1


def small_grads_ok():
    """inside a backward pass of the autograd engine (where the end-of-pass callback can be queued)?"""
    return DETERMINISTIC_SMALL_GRADS and _graph_task() != -1


def small_grads_reset():
    """drop whatever an aborted backward pass left behind (the engine discards a graph task's final callbacks when a backward raises:
    the entries would otherwise wait for a flush that never comes)"""
    _SmallGrads.pending, _SmallGrads.task, _SmallGrads.announced, _SmallGrads.nodes = [], -1, set(), {}
    _WgradQueue.items, _WgradQueue.notify, _WgradQueue.task = [], [], -1


def small_grad_scratch(target, param):
    """zeroed float64 stand-in for the fp32 gradient storage `target` (contiguous) of `param`; added into it — and `grad_done(param)` called —
    when the running backward pass ends.  The pending list is keyed on the engine's graph-task id: entries of ANOTHER pass are leftovers of a
    backward that raised after taking a scratch (its callback was dropped with it) — they are discarded and the callback queued anew.  A
    parameter deferred twice in one pass (a module applied twice, shared weights) reuses ONE scratch: ly_f64_add's table then holds one
    entry per destination (two entries with the same target would race)."""
    task = _graph_task()
    if task != _SmallGrads.task:
        _SmallGrads.pending, _SmallGrads.task, _SmallGrads.announced, _SmallGrads.nodes = [], task, set(), {}
        torch.autograd.Variable._execution_engine.queue_callback(flush_small_grads)
    tgt = target.view(-1)
    for scr, t, prm in _SmallGrads.pending:
        if t.data_ptr() == tgt.data_ptr() and t.numel() == tgt.numel():
            return scr
    node = _autograd_node()
    if GRAD_LISTENERS and SMALL_GRADS_EARLY_FLUSH and node is not None:
        # under a gradient listener (ddp.GradReducer) a waiting sum holds back its whole bucket's exchange until the backward pass ends
        # (tools/dp_overlap_probe.py: 5.4 of 12.5 MB left at 99-100 % of the backward).  Entries taken by EARLIER autograd nodes are complete
        # — their kernels are in the stream — and leave as soon as a handful has collected; entries of the running node stay (it may not
        # have launched the kernel that fills them yet).
        if param is not None and id(param) in _SmallGrads.announced:
            raise RuntimeError("small_grad_scratch: a parameter whose deferred gradient was already announced to the gradient listeners "
                               "receives another contribution in the same backward pass (shared weights under a GradReducer): set "
                               "ops.SMALL_GRADS_EARLY_FLUSH = False")
        old = [e for e in _SmallGrads.pending if _SmallGrads.nodes.get(id(e[0])) != node]
        if len(old) >= SMALL_GRADS_FLUSH_MIN:
            _SmallGrads.pending = [e for e in _SmallGrads.pending if _SmallGrads.nodes.get(id(e[0])) == node]
            _flush_small_items(old)
    scr = zeros_f64(target.numel(), target.device)
    if param is not None and not any(prm is param for _, _, prm in _SmallGrads.pending):
        for fn in GRAD_DEFER_LISTENERS:
            fn(param)
    _SmallGrads.pending.append((scr, tgt, param))
    _SmallGrads.nodes[id(scr)] = node
    return scr


SMALL_GRADS_EARLY_FLUSH = True
SMALL_GRADS_FLUSH_MIN = 6


def _autograd_node():
    fn = getattr(torch._C, "_current_autograd_node", None)
    n = fn() if fn is not None else None
    return id(n) if n is not None else None


def f64_round(scratches, shapes):
    """fp32 tensors holding the rounded sums of float64 scratches (the route without a gradient sink: the values go back to autograd) — ONE
    ly_f64_add launch into one zeroed buffer"""
    assert 0 < len(scratches) <= capi.F64_ADD_MAX
    sizes = [s_.numel() for s_ in scratches]
    flat = torch.zeros(sum(sizes), dtype=torch.float32, device=scratches[0].device)
    t = capi.LyF64AddTable()
    outs, off = [], 0
    for j, (scr, n, shp) in enumerate(zip(scratches, sizes, shapes)):
        t.src[j], t.dst[j], t.n[j] = scr.data_ptr(), flat.data_ptr() + 4 * off, n
        outs.append(flat[off:off + n].view(shp))
        off += n
    t.count = len(scratches)
    capi.check(capi.lib().ly_f64_add(ctypes.byref(t), capi.stream_ptr()), "ly_f64_add")
    return outs


def flush_small_grads():
    items, _SmallGrads.pending, _SmallGrads.task = _SmallGrads.pending, [], -1
    _flush_small_items(items)
    _SmallGrads.announced, _SmallGrads.nodes = set(), {}


def _flush_small_items(items):
    for i in range(0, len(items), capi.F64_ADD_MAX):
        chunk = items[i:i + capi.F64_ADD_MAX]
        t = capi.LyF64AddTable()
        for j, (scr, tgt, _) in enumerate(chunk):
            t.src[j], t.dst[j], t.n[j] = scr.data_ptr(), tgt.data_ptr(), tgt.numel()
        t.count = len(chunk)
        capi.check(capi.lib().ly_f64_add(ctypes.byref(t), capi.stream_ptr()), "ly_f64_add")
    done = set()
    for _, _, prm in items:
        if prm is not None and id(prm) not in done:
            done.add(id(prm))
            _SmallGrads.announced.add(id(prm))
            grad_done(prm)


def mlpblock(x, y, n, h, w, c, wp, w1, w2, sc, sh, stats=None):
    m = n * h * w
    cc, nt, ht, t2d = mlp_config(c, m, w, x.dtype == torch.bfloat16)
    kind = "_ring" if c >= 80 else "_occ4" if (c <= 24 and t2d == "true") else ""
    name = f"ly_mlpblock_fwd{kind}_kernel<{_tname(x)}, {cc}, {nt}, {ht}, {t2d}, {'true' if stats is not None else 'false'}>"
    if c == 80 and x.dtype == torch.bfloat16 and m >= 128 * 256 and (256 + 2 * w + 2) * 5 <= 2048:
        name = f"ly_mlpblock_res_kernel<__bf16, 80, 2, 2, {'true' if stats is not None else 'false'}, 4>"      # ly_mlp_dispatch_80 (csrc/ly_mlpblock_b.hip)
    elif c < 80 and w % 16 == 0 and w >= 32 and n * -(-h // 8) * (w // 16) >= 1024 and (x.dtype == torch.bfloat16 or c < 40):
        name = f"ly_mlpblock_persist_kernel<{_tname(x)}, {c}, 2, {ht}, {1 if stats is not None else 0}, 1>"      # dispatch_nt_t: the persistent patch walk
    with _Timed(name, 2.0 * m * (9 * (c // 4) ** 2 + 4 * c * c),
                x.element_size() * (1 if stats is not None else 2) * m * c + 4.0 * (9 * (c // 4) ** 2 + 4 * c * c)):
        capi.check(capi.lib().ly_mlpblock_fwd(_p(x), _p(y), n, h, w, c, _p(wp), _p(w1), _p(w2), _p(sc), _p(sh), _p(stats), capi.dtype_code(x),
                                              capi.stream_ptr()), "ly_mlpblock_fwd")


def mlpblock_bwd_ok(x, c):
    """the fused MLPBlock backward (csrc/ly_mlpblock_bwd.hpp) is built for bf16 storage and C in 16 / 24 / 40 / 80"""
    return MLP_BWD_FUSED and x.dtype == torch.bfloat16 and bool(capi.lib().ly_mlpblock_bwd_ok(c, capi.dtype_code(x)))


MLP_BWD_FUSED = True       # tools / tests: False keeps the unfused backward (eleven launches over 2C-wide tensors)


def _mlpblock_bwd_slab(c, device):
    """the slab the fused backward's blocks park their weight-gradient tiles in: written by the pass and read back by its fold launch.  Taken
    from the caching allocator PER CALL (round 6; it was one persistent buffer per (device, C) shared by every MLPBlock of that width, safe
    only while all of them ran on one stream in order — ADVICE r5): the allocator orders reuse by stream, and inside a hipGraph capture the
    block comes from the graph's private pool, whose address the replay keeps."""
    return torch.empty(int(capi.lib().ly_mlpblock_bwd_slab_floats(c)), dtype=torch.float32, device=device)


def mlpblock_bwd(x, dy, n, h, w, c, wp, w1, w2t, w1t, a, b, *, stats=None, g=None, alpha=None, kappa=None, lam=None, dw1=None, dw2=None):
    """one pass of the fused MLPBlock backward: stats given = pass 1 (BatchNorm sums), else pass 2 (g, dw1 +=, dw2 +=)"""
    m = n * h * w
    p1 = stats is not None
    slab = None if p1 else _mlpblock_bwd_slab(c, x.device)
    hid = 2 * c
    flops = 2.0 * m * (9 * (c // 4) ** 2 + (2 if p1 else 5) * c * hid)
    with _Timed(f"ly_mlpblock_bwd_kernel<{c}, pass {1 if p1 else 2}>", flops, x.element_size() * (2 if p1 else 3) * m * c + 4.0 * (9 * (c // 4) ** 2 + 2 * c * hid)):
        capi.check(capi.lib().ly_mlpblock_bwd(_p(x), _p(dy), _p(g), n, h, w, c, _p(wp), _p(w1), _p(w2t), _p(w1t), _p(a), _p(b), _p(alpha), _p(kappa), _p(lam),
                                              _p(stats), _p(slab), slab.numel() if slab is not None else 0, _p(dw1), _p(dw2), 1 if p1 else 2,
                                              capi.dtype_code(x), capi.stream_ptr()), "ly_mlpblock_bwd")


def mlpblock_bwd_dx_ok(x, c, w):
    """ly_mlpblock_bwd_dx is built for this (storage, C) and its tile fits LDS at map width w"""
    return MLP_BWD_FUSED and x.dtype == torch.bfloat16 and bool(capi.lib().ly_mlpblock_bwd_dx_ok(c, w, capi.dtype_code(x)))


def mlpblock_bwd_dx(g, dy, x, n, h, w, c, wpt, dwp=None, lddw=0, dw_ts=0, dw_cs=0):
    """dx = dy + [pconv^T(g[:, :c/4]) | g[:, c/4:]] (+ dwp += the partial conv's weight gradient where the kernel builds it in): returns
    (dx, whether dwp was done)"""
    dx = torch.empty_like(dy)
    m = n * h * w
    slab = _mlpblock_bwd_slab(c, x.device) if (dwp is not None and capi.lib().ly_mlpblock_bwd_slab_floats(c) > 0) else None
    with _Timed(f"ly_mlpblock_bwd_dx_kernel<{c}>", 2.0 * m * 18 * (c // 4) ** 2, x.element_size() * m * (3 * c + c // 4) + 4.0 * 9 * (c // 4) ** 2):
        rc = capi.lib().ly_mlpblock_bwd_dx(_p(g), _p(dy), _p(x), _p(dx), n, h, w, c, _p(wpt), _p(slab), slab.numel() if slab is not None else 0,
                                           _p(dwp if slab is not None else None), lddw, dw_ts, dw_cs, capi.dtype_code(x), capi.stream_ptr())
    if rc < 0:
        capi.check(rc, "ly_mlpblock_bwd_dx")
    return dx, rc == 0


def chan_moments(x, ldx, rows, c, f64=False, striped=False):
    """per-channel (sum x, sum x^2) over the rows of an [rows, c] matrix -> [2c]; accumulated in double stripes, folded in index order
    (float32 result unless f64: ly_rfcbam_gen_prepare takes the doubles; striped: the [STRIPES][2c] accumulators as they are — that kernel
    folds them itself)"""
    mom = new_stats(c, x.device)
    with _Timed(f"ly_chan_moments_kernel<{_tname(x)}>", 3.0 * rows * c, x.element_size() * rows * c):
        capi.check(capi.lib().ly_chan_moments(_p(x), ldx, rows, c, _p(mom), capi.dtype_code(x), capi.stream_ptr()), "ly_chan_moments")
    if striped:
        return mom
    out = torch.empty(2 * c, dtype=torch.float64, device=x.device)
    capi.check(capi.lib().ly_sum_rows_f64(_p(mom), mom.shape[0], 2 * c, _p(out), capi.stream_ptr()), "ly_sum_rows_f64")
    return out if f64 else out.float()


STRIPES = capi.STATS_STRIPES


class _StatsPool:
    """One zero fill per training step instead of one per BatchNorm pass (74 launches at lead-yolo-s): between
    `stats_pool_begin` and `stats_pool_end` (train.train_step) `new_stats` hands out slices of one buffer that was zeroed at
    `begin`.  Safe because a statistics accumulator is transient — written by one pass, read by the `ly_bn_finalize` /
    `ly_bn_bwd_coeffs` launch right after it on the same stream, never saved — and the next `begin` (same stream) is ordered
    after every reader.  Outside a begin/end pair, or when the buffer is too small (first step, a larger model), `new_stats`
    falls back to its own `torch.zeros`; the demand seen in a step sizes the buffer of the next."""
    buf = None
    off = 0
    used = 0
    need = 0
    active = False


_POOL = _StatsPool()


def stats_pool_begin(device):
    small_grads_reset()                      # (a step starts outside any backward pass: whatever is pending belongs to an aborted one)
    p = _POOL
    if p.buf is None or p.buf.device != device or p.buf.numel() < p.need:
        p.buf = torch.zeros(max(p.need, 1), dtype=torch.float32, device=device) if p.need else None
    elif p.buf is not None:
        p.buf.zero_()
    p.off = p.used = 0
    p.active = True


def stats_pool_end():
    p = _POOL
    p.active = False
    p.need = max(p.need, p.used)


def new_sums(nch, device):
    """zeroed striped FLOAT accumulator [STRIPES][2*nch] of a backward reduction (ly_bnact_bwd_reduce, ly_rf*_bwd -> ly_bn_bwd_coeffs)"""
    p = _POOL
    n = STRIPES * 2 * nch
    if p.active:
        span = (n + 63) // 64 * 64                  # 256-byte aligned slices
        p.used += span
        if p.buf is not None and p.buf.device == device and p.off + span <= p.buf.numel():
            v = p.buf[p.off:p.off + n].view(STRIPES, 2 * nch)
            p.off += span
            return v
    return torch.zeros(STRIPES, 2 * nch, dtype=torch.float32, device=device)


_ONES = {}


def ones_f32(numel, device):
    """a shared READ-ONLY vector of ones (an identity epilogue scale): allocated once per (size, device), no fill launch per step"""
    t = _ONES.get((numel, device))
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            return torch.ones(numel, dtype=torch.float32, device=device)
        t = _ONES[(numel, device)] = torch.ones(numel, dtype=torch.float32, device=device)
    return t


def zeros_f64(numel, device):
    """zeroed float64 scratch out of the step's zero pool (an 8-byte view of 2 * numel pool floats; pool slices are 256-byte aligned)"""
    return zeros_f32(2 * numel, device).view(torch.float64)


def new_stats(nch, device):
    """zeroed striped DOUBLE accumulator for a forward statistics pass over `nch` channels: [STRIPES][2*nch] float64 (ly_stats_flush of
    csrc/ly_common.hpp adds the waves' fp32 partial sums as doubles: the batch statistics, and with them every ReLU / arg-max decision of
    a training step, are reproducible from run to run; ly_bn_finalize reads it with stats_f64 = 1)"""
    return zeros_f64(STRIPES * 2 * nch, device).view(STRIPES, 2 * nch)


def zeros_f32(numel, device):
    """zeroed fp32 scratch (transient accumulators of one launch sequence: never saved, never a parameter gradient) from the step's
    pool, so that a training step issues one fill for all of them"""
    p = _POOL
    if p.active:
        span = (numel + 63) // 64 * 64
        p.used += span
        if p.buf is not None and p.buf.device == device and p.off + span <= p.buf.numel():
            v = p.buf[p.off:p.off + numel]
            p.off += span
            return v
    return torch.zeros(numel, dtype=torch.float32, device=device)


def bn_finalize(bn, stats, nch, count, n=None, c_off=0, bias=None, pad_to=0, want_stats=False, into=None):
    """Train-mode BatchNorm2d from the striped sums of a statistics pass, ONE launch (ly_bn_finalize): returns
    (scale, shift) of y = x*scale + shift [+ (mean, invstd)], updates running_mean/var/num_batches_tracked in place.
    into = (scale, shift, mean, invstd): contiguous fp32 [n] destinations (slices of a wider vector: two BatchNorms over one stacked output)."""
    n = nch if n is None else n
    dev = stats.device
    size = max(n, pad_to)
    if into is not None:
        scale, shift, mean, invstd = into
        want_stats = True
    else:
        if size > n:                 # zero-padded tails (MLPBlock's hidden tiles): out of the step's zero pool, no fill launches
            scale, shift = zeros_f32(size, dev), zeros_f32(size, dev)
        else:
            scale, shift = torch.empty(size, dtype=torch.float32, device=dev), torch.empty(size, dtype=torch.float32, device=dev)
        mean = torch.empty(n, dtype=torch.float32, device=dev) if want_stats else None
        invstd = torch.empty(n, dtype=torch.float32, device=dev) if want_stats else None
    track = bn.track_running_stats and bn.running_mean is not None
    if bn.weight is not None and bn.weight.dtype != torch.float32:
        raise NotImplementedError("train-mode BatchNorm needs float32 parameters and buffers: train under torch.autocast (fp32 master "
                                  "weights, as the reference does), not with a .half()/.bfloat16() model")
    if track and bn.momentum is None:
        raise NotImplementedError("BatchNorm with momentum=None (cumulative average) is not built")
    if track:
        from . import pack
        pack.touch()                     # running statistics are written by the kernel: eval-mode folded caches must refresh
    if stats.dtype not in (torch.float32, torch.float64):
        raise TypeError("bn_finalize: statistics must be float32 or float64")
    capi.check(capi.lib().ly_bn_finalize(_p(stats), int(stats.dtype == torch.float64), stats.shape[0], nch, c_off, n, float(count), _p(bn.weight), _p(bn.bias), _p(bias), float(bn.eps),
                                         float(bn.momentum or 0.0), _p(bn.running_mean if track else None), _p(bn.running_var if track else None),
                                         _p(bn.num_batches_tracked if track else None), _p(scale), _p(shift), _p(mean), _p(invstd),
                                         capi.stream_ptr()), "ly_bn_finalize")
    return (scale, shift, mean, invstd) if want_stats else (scale, shift)


def _bn_train_args(bn):
    """(gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked) of a train-mode BatchNorm for the finalize kernels"""
    track = bn.track_running_stats and bn.running_mean is not None
    if bn.weight is not None and bn.weight.dtype != torch.float32:
        raise NotImplementedError("train-mode BatchNorm needs float32 parameters and buffers: train under torch.autocast (fp32 master "
                                  "weights, as the reference does), not with a .half()/.bfloat16() model")
    if track and bn.momentum is None:
        raise NotImplementedError("BatchNorm with momentum=None (cumulative average) is not built")
    if track:
        from . import pack
        pack.touch()                     # running statistics are written by the kernel: eval-mode folded caches must refresh
    return (_p(bn.weight), _p(bn.bias), float(bn.eps), float(bn.momentum or 0.0), _p(bn.running_mean if track else None),
            _p(bn.running_var if track else None), _p(bn.num_batches_tracked if track else None))


def bn_finalize_pair(bn1, bn2, stats, c_half, count, v):
    """bn_finalize of two BatchNorms over one stacked [2 c_half]-channel statistics array in ONE launch; v: fp32 [4, 2 c_half] that
    receives scale, shift, mean, invstd"""
    if stats.dtype not in (torch.float32, torch.float64) or v.dtype != torch.float32 or not v.is_contiguous() or tuple(v.shape) != (4, 2 * c_half):
        raise TypeError("bn_finalize_pair: float32 / float64 statistics and a contiguous float32 [4, 2 c_half] result expected")
    capi.check(capi.lib().ly_bn_finalize_pair(_p(stats), int(stats.dtype == torch.float64), stats.shape[0], c_half, float(count), *_bn_train_args(bn1),
                                              *_bn_train_args(bn2), _p(v[0]), _p(v[1]), _p(v[2]), _p(v[3]), capi.stream_ptr()), "ly_bn_finalize_pair")


def bn_bwd_coeffs_pair(sums0, sums1, c_half, count, v, targets, coef):
    """bn_bwd_coeffs (train) of the same two units in ONE launch.  v: [4, 2 c_half] of bn_finalize_pair; targets = ((dgamma0, dbeta0),
    (dgamma1, dbeta1)): fp32 [c_half] tensors the kernel ADDS into; coef: fp32 [3, 2 c_half] that receives alpha, kappa, lambda"""
    if sums0.dtype != sums1.dtype or sums0.shape != sums1.shape or sums0.dim() != 2:
        raise TypeError("bn_bwd_coeffs_pair: two striped sum arrays of one shape expected")
    (g0, b0), (g1, b1) = targets
    capi.check(capi.lib().ly_bn_bwd_coeffs_pair(_p(sums0), _p(sums1), int(sums0.dtype == torch.float64), sums0.shape[0], c_half, float(count), _p(v[0]), _p(v[2]),
                                                _p(v[3]), _p(g0), _p(b0), _p(g1), _p(b1), _p(coef[0]), _p(coef[1]), _p(coef[2]), capi.stream_ptr()),
               "ly_bn_bwd_coeffs_pair")


def bn_bwd_coeffs(sums, n, count, a, mean, invstd, train, dgamma=None, dbeta=None, transpose=None, into=None):
    """(dgamma, dbeta, alpha, kappa, lambda) from (striped) sums [.., 2n] of ly_bnact_bwd_reduce, ONE launch.  dgamma / dbeta given:
    the kernel ADDS into them (a parameter's persistent gradient storage, see GradSink) and None is returned in their place.
    transpose = (A, B): the sums' channels are in [A][B] order, dgamma / dbeta in [B][A] (generate BatchNorm of RFCBAMConv k = 3)."""
    dev = sums.device
    stripes = sums.shape[0] if sums.dim() == 2 else 1
    # into = (alpha, kappa, lambda): contiguous fp32 [n] destinations (slices of wider vectors: two units side by side)
    out = torch.empty(5 if into is None else 2, n, dtype=torch.float32, device=dev)
    direct = dgamma is not None and dbeta is not None
    if not direct:
        out[:2].zero_()
    capi.check(capi.lib().ly_bn_bwd_coeffs(_p(sums), int(sums.dtype == torch.float64), stripes, n, float(count), _p(a), _p(mean), _p(invstd), int(train), _p(dgamma if direct else out[0]),
                                           _p(dbeta if direct else out[1]), *[_p(t) for t in (into if into is not None else (out[2], out[3], out[4]))],
                                           *((transpose if direct else None) or (0, 0)), capi.stream_ptr()), "ly_bn_bwd_coeffs")
    al, ka, la = into if into is not None else (out[2], out[3], out[4])
    return (None if direct else out[0]), (None if direct else out[1]), al, ka, la


class GradSink:
    """Where the backward kernels put parameter gradients.  Default (no sink installed): fresh tensors handed to autograd, which
    accumulates them into `.grad` — one extra launch per parameter and step, plus the zero fill of every fresh tensor.  With a sink
    (optim.FusedSGD installs one: its persistent, step-zeroed gradient storage) ly_wgrad / ly_bn_bwd_coeffs ADD straight into
    `param.grad` and the autograd function returns None for that parameter; `done(param)` then stands in for the post-accumulate
    hook (ddp.GradReducer launches a bucket's all-reduce from it)."""

    def __init__(self):
        self.targets = {}          # id(param) -> gradient tensor (same shape as the parameter, fp32, contiguous)
        self.writes = 0            # bumped whenever a backward kernel is handed a target (or a captured backward is replayed): the
                                   # optimiser's zero_grad() compares it with the value at its last step() to know whether the storage is dirty

    def is_target(self, t):
        """t is (a view from the start of) one of the sink's persistent gradient tensors"""
        if getattr(self, "_ptr_src", None) is not self.targets or len(self.targets) != self._nptr:      # (the optimiser assigns a whole new dict)
            self._ptrs, self._nptr, self._ptr_src = {v.data_ptr() for v in self.targets.values()}, len(self.targets), self.targets
        return t.data_ptr() in self._ptrs

    def target(self, p):
        if p is None or not torch.is_tensor(p):
            return None
        t = self.targets.get(id(p))
        if t is not None and p.grad is not t:
            return None                         # someone re-assigned .grad: fall back to autograd for this one
        if t is not None:
            self.writes += 1
        return t



SINK = None
GRAD_DEFER_LISTENERS = []          # callables(param): told that a directly-written gradient will be announced at the end of the backward pass
GRAD_LISTENERS = []                # callables(param): told when a directly-written gradient is complete (ddp.GradReducer)


def grad_target(p):
    return SINK.target(p) if SINK is not None else None


def grad_done(p):
    """the directly-written gradient of parameter p is complete AND its launches are in the stream.  A weight gradient that still waits in the
    deferral queue (_WgradQueue) is announced when its group has been launched instead; the listeners are told now that it comes later."""
    if GRAD_LISTENERS and _WgradQueue.items and SINK is not None:
        t = SINK.targets.get(id(p))
        # a waiting problem covers the address RANGE [dw, dw + N * lddw): the stacked cv1 / cv2 problem of a ConvBnActPair (N = 2 c_) also writes
        # its partner's storage, which starts 4 * numel bytes behind dw — a pointer-equality match announced that partner early
        if t is not None and any(lo <= t.data_ptr() < hi for lo, hi in map(_wgrad_span, _WgradQueue.items)):
            if all(prm is not p for prm, _ in _WgradQueue.notify):
                _WgradQueue.notify.append((p, t.data_ptr()))
                for fn in GRAD_DEFER_LISTENERS:
                    fn(p)
            return
    for fn in GRAD_LISTENERS:
        fn(p)


def coordatt_conv1_stats(pool, positions, c, mip, w1, b1):
    st = zeros_f64(2 * mip, pool.device)
    capi.check(capi.lib().ly_coordatt_conv1_stats(_p(pool), positions, c, mip, _p(w1), _p(b1), _p(st), capi.stream_ptr()),
               "ly_coordatt_conv1_stats")
    return st


def bn_batch_affine(bn, s1, s2, count):
    """Train-mode BatchNorm from per-channel sums: returns (scale, shift) of y = x*scale + shift with the
    BATCH statistics (biased variance), and updates running_mean / running_var (unbiased, momentum) and
    num_batches_tracked in place exactly as nn.BatchNorm2d does.  Device tensors only, no host sync."""
    return bn_batch_stats(bn, s1, s2, count)[:2]


_TRIU = None


def rfcbam_generate_stats(x, ldx, n, h, w, c, s, gen_w):
    """Batch statistics of the k=3 `generate` BatchNorm input: returns (sum a, sum a^2) per generate channel
    (c*9 + t) and the sample count, from the per-channel tap moments (see ly_rfcbam_tap_moments)."""
    global _TRIU
    mom = torch.zeros(54, c, dtype=torch.float64, device=x.device)
    with _Timed(f"ly_rfcbam_tap_moments_kernel<{_tname(x)}>", 108.0 * n * h * w * c / (s * s), x.element_size() * n * h * w * c):
        capi.check(capi.lib().ly_rfcbam_tap_moments(_p(x), ldx, n, h, w, c, s, _p(mom), capi.dtype_code(x), capi.stream_ptr()), "ly_rfcbam_tap_moments")
    mom = mom.float()
    if _TRIU is None or _TRIU[0].device != x.device:
        iu = torch.triu_indices(9, 9, device=x.device)
        _TRIU = (iu[0], iu[1])
    m1 = mom[:9].t()                                            # [c, 9]
    M = torch.zeros(c, 9, 9, dtype=torch.float32, device=x.device)
    M[:, _TRIU[0], _TRIU[1]] = mom[9:].t()
    M = M + M.transpose(1, 2) - torch.diag_embed(torch.diagonal(M, dim1=1, dim2=2))
    wv = gen_w.detach().float().view(c, 9, 9)                  # [c, t, u]
    s1 = torch.einsum("ctu,cu->ct", wv, m1)
    s2 = torch.einsum("ctu,cuv,ctv->ct", wv, M, wv)
    ho, wo = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
    return s1.reshape(-1), s2.reshape(-1), n * ho * wo


def rfcbam_gen_prepare(x, ldx, n, h, w, c, k, s, gen_w, bn):
    """Train-mode `generate` BatchNorm of RFCBAMConv in two launches: the moment kernel over x, then ly_rfcbam_gen_prepare (batch
    statistics, running-stat update, scale / shift in both index orders and the folded weights in the kernels' LDS orders).
    Returns dict(gs, gb, gmean, ginv [c*kk + t order], ag, bg, gmean_tc, ginv_tc [t*c + c order], a1 (k=1), wq_stats, wq_main (k=3))."""
    kk = k * k
    dev = x.device
    if k == 3:
        mom = zeros_f64(54 * c, dev)
        with _Timed(f"ly_rfcbam_tap_moments_kernel<{_tname(x)}>", 108.0 * n * h * w * c / (s * s), x.element_size() * n * h * w * c):
            capi.check(capi.lib().ly_rfcbam_tap_moments(_p(x), ldx, n, h, w, c, s, _p(mom), capi.dtype_code(x), capi.stream_ptr()), "ly_rfcbam_tap_moments")
        ho, wo = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
        count = n * ho * wo
    else:
        mom = chan_moments(x, ldx, n * h * w, c, striped=True)
        count = n * h * w
    g = c * kk
    out8 = torch.empty(8, g, dtype=torch.float32, device=dev)
    a1 = torch.empty(c, dtype=torch.float32, device=dev) if k == 1 else None
    cps, cpm = (c + 31) // 32 * 32, (c + 15) // 16 * 16
    wqs = torch.empty(cps * 90, dtype=torch.float32, device=dev) if k == 3 else None
    wqm = torch.empty(cpm * 90, dtype=torch.float32, device=dev) if k == 3 else None
    wqc = torch.empty(c * 100, dtype=torch.float32, device=dev) if (k == 3 and c % 32 == 0) else None
    if bn.weight.dtype != torch.float32:
        raise NotImplementedError("train-mode BatchNorm needs float32 parameters and buffers")
    track = bn.track_running_stats and bn.running_mean is not None
    capi.check(capi.lib().ly_rfcbam_gen_prepare(_p(mom), c, k, _p(gen_w.detach()), _p(bn.weight.detach()), _p(bn.bias.detach()), float(bn.eps),
                                                float(bn.momentum or 0.0), float(count), _p(bn.running_mean if track else None),
                                                _p(bn.running_var if track else None), _p(bn.num_batches_tracked if track else None), _p(out8),
                                                _p(a1), _p(wqs), _p(wqm), _p(wqc), mom.shape[0] if k == 1 else 1, capi.stream_ptr()), "ly_rfcbam_gen_prepare")
    from . import pack
    pack.touch()                                   # running statistics written behind torch's version counters
    return dict(gs=out8[0], gb=out8[1], gmean=out8[2], ginv=out8[3], ag=out8[4], bg=out8[5], gmean_tc=out8[6], ginv_tc=out8[7], a1=a1,
                wq_stats=wqs, wq_main=wqm, wq_c=wqc)


# ---- backward building blocks (training step) --------------------------------------------------------
def bn_batch_stats(bn, s1, s2, count):
    """As bn_batch_affine, additionally returning the batch mean and 1/sqrt(var + eps) the backward needs."""
    with torch.no_grad():
        mean = s1 / count
        var = (s2 / count - mean * mean).clamp_(min=0.0)
        invstd = torch.rsqrt(var + bn.eps)
        scale = bn.weight.detach().float() * invstd
        shift = bn.bias.detach().float() - mean * scale
        if bn.track_running_stats and bn.running_mean is not None:
            mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked + 1)
            bn.running_mean.mul_(1.0 - mom).add_(mean, alpha=mom)
            bn.running_var.mul_(1.0 - mom).add_(var * (count / max(count - 1, 1)), alpha=mom)
            bn.num_batches_tracked += 1
    return scale.contiguous(), shift.contiguous(), mean, invstd


def bnact_fwd(u, ldu, rows, c, a, b, act, y, ldy):
    with _Timed(f"ly_bnact_fwd_kernel<{_tname(u)}, {act}>", 4.0 * rows * c, 2.0 * u.element_size() * rows * c):
        capi.check(capi.lib().ly_bnact_fwd(_p(u), ldu, rows, c, _p(a), _p(b), act, _p(y), ldy, capi.dtype_code(u), capi.stream_ptr()), "ly_bnact_fwd")


def bnact_bwd_reduce(dy, lddy, u, ldu, rows, c, a, b, act):
    sums = new_stats(c, u.device)                 # double accumulators: the coefficients of du, and with them the whole activation-gradient chain, are reproducible
    with _Timed(f"ly_bnact_bwd_reduce_kernel<{_tname(u)}, {act}>", 6.0 * rows * c, 2.0 * u.element_size() * rows * c):
        capi.check(capi.lib().ly_bnact_bwd_reduce(_p(dy), lddy, _p(u), ldu, rows, c, _p(a), _p(b), act, _p(sums), capi.dtype_code(u),
                                                  capi.stream_ptr()), "ly_bnact_bwd_reduce")
    return sums


def bnact_bwd_apply(dy, lddy, u, ldu, rows, c, a, b, act, alpha, kappa, lam, du, lddu):
    with _Timed(f"ly_bnact_bwd_apply_kernel<{_tname(u)}, {act}>", 8.0 * rows * c, 3.0 * u.element_size() * rows * c):
        capi.check(capi.lib().ly_bnact_bwd_apply(_p(dy), lddy, _p(u), ldu, rows, c, _p(a), _p(b), act, _p(alpha), _p(kappa), _p(lam),
                                                 _p(du), lddu, capi.dtype_code(u), capi.stream_ptr()), "ly_bnact_bwd_apply")


def bnact_bwd_reduce_pair(dy1, lddy1, dy2, lddy2, csplit, u, ldu, rows, c, a, b, act):
    """both units of a stacked pair in one pass -> (sums1, sums2)"""
    s1, s2 = new_stats(csplit, u.device), new_stats(c - csplit, u.device)
    with _Timed(f"ly_bnact_bwd_reduce_kernel<{_tname(u)}, {act}>", 6.0 * rows * c, 2.0 * u.element_size() * rows * c):
        capi.check(capi.lib().ly_bnact_bwd_reduce_pair(_p(dy1), lddy1, _p(dy2), lddy2, csplit, _p(u), ldu, rows, c, _p(a), _p(b), act, _p(s1), _p(s2),
                                                       capi.dtype_code(u), capi.stream_ptr()), "ly_bnact_bwd_reduce_pair")
    return s1, s2


def bnact_bwd_apply_pair(dy1, lddy1, dy2, lddy2, csplit, u, ldu, rows, c, a, b, act, alpha, kappa, lam, du, lddu):
    with _Timed(f"ly_bnact_bwd_apply_kernel<{_tname(u)}, {act}>", 8.0 * rows * c, 3.0 * u.element_size() * rows * c):
        capi.check(capi.lib().ly_bnact_bwd_apply_pair(_p(dy1), lddy1, _p(dy2), lddy2, csplit, _p(u), ldu, rows, c, _p(a), _p(b), act, _p(alpha), _p(kappa),
                                                      _p(lam), _p(du), lddu, capi.dtype_code(u), capi.stream_ptr()), "ly_bnact_bwd_apply_pair")


def wgrad(*, M, H, W, N, du, lddu, x, ldx, Hin, Win, Cin, dw, lddw, ks=1, stride=1, pad=0, nchw=False, up2=False, du_off=0, x_off=0,
          dw_off=0, dw_ts=None, dw_cs=1, n_valid=None, c_valid=None, x_scale=None, x_shift=None, _now=False):
    """dw[n][tap*dw_ts + c*dw_cs] += sum_pixels du[p][n] * x[src(p, tap)][c] for n < n_valid, c < c_valid; *_off are element offsets
    into the tensors.  Defaults: packed rows (dw_ts = Cin, dw_cs = 1), everything valid.  x_scale / x_shift (fp32 [Cin], plain-row 1x1
    problems): x is read as max(x*x_scale + x_shift, 0)."""
    q = dict(M=M, H=H, W=W, N=N, du=du, lddu=lddu, x=x, ldx=ldx, Hin=Hin, Win=Win, Cin=Cin, dw=dw, lddw=lddw, ks=ks, stride=stride,
             pad=pad, nchw=nchw, up2=up2, du_off=du_off, x_off=x_off, dw_off=dw_off, dw_ts=dw_ts, dw_cs=dw_cs, n_valid=n_valid,
             c_valid=c_valid, x_scale=x_scale, x_shift=x_shift)
    if not _now and _wgrad_deferrable(q):
        return _wgrad_enqueue(q)
    P = _wgrad_params(**q)
    with _Timed(wgrad_kernel_name(_tname(x), N, ks * ks * Cin, ks == 1 and stride == 1 and pad == 0 and not nchw and not up2,
                                  (not nchw) and N % 4 == 0 and Cin % 4 == 0 and lddu % 4 == 0 and ldx % 4 == 0, x_scale is not None,
                                  _wgrad_octets(x, du, N, Cin, lddu, ldx, du_off, x_off)),
                2.0 * M * N * ks * ks * Cin, x.element_size() * M * (N + Cin) + 4.0 * N * ks * ks * Cin):
        capi.check(capi.lib().ly_wgrad(ctypes.byref(P), capi.stream_ptr()), "ly_wgrad")


def _wgrad_params(*, M, H, W, N, du, lddu, x, ldx, Hin, Win, Cin, dw, lddw, ks=1, stride=1, pad=0, nchw=False, up2=False, du_off=0, x_off=0,
                  dw_off=0, dw_ts=None, dw_cs=1, n_valid=None, c_valid=None, x_scale=None, x_shift=None):
    def at(t, off):
        return ctypes.c_void_p(t.data_ptr() + t.element_size() * off)
    if du.dtype != x.dtype:
        raise capi.HipLibraryError(f"wgrad: du ({du.dtype}) and x ({x.dtype}) must share one storage dtype")
    if (x_scale is None) != (x_shift is None):
        raise ValueError("wgrad: x_scale and x_shift come together")
    if x_scale is not None and (x_scale.dtype != torch.float32 or x_shift.dtype != torch.float32 or x_scale.numel() < Cin or x_shift.numel() < Cin):
        raise ValueError("wgrad: x_scale / x_shift must be float32 vectors of at least Cin elements")
    ws = wgrad_workspace(x.device)
    return capi.LyWgradParams(M, H, W, N, at(du, du_off), lddu, at(x, x_off), ldx, Hin, Win, Cin, ks, stride, pad, int(nchw), int(up2),
                              at(dw, dw_off), lddw, capi.dtype_code(x), Cin if dw_ts is None else dw_ts, dw_cs, N if n_valid is None else n_valid,
                              Cin if c_valid is None else c_valid, _p(x_scale), _p(x_shift), _p(ws), ws.numel() if ws is not None else 0)


WGRAD_WS_FLOATS = 24 << 20          # 96 MB of fp32 scratch per device: the partial tiles of one weight-gradient launch (largest: 38 MB)
_WGRAD_WS = {}


def wgrad_workspace(device):
    """Per-device scratch the weight-gradient kernels park their per-chunk partial tiles in (LyWgradParams.ws).  Allocated ONCE (a captured
    hipGraph keeps its address; every launch overwrites what it reads back in the same launch sequence, stream-ordered), never zeroed.
    ONE buffer per device: weight-gradient launches of a device are expected on one stream at a time (the training step's); two training
    loops on different streams of the same device in one process must set WGRAD_WS_FLOATS = 0 (atomic accumulation) for one of them."""
    if WGRAD_WS_FLOATS <= 0:
        return None
    ws = _WGRAD_WS.get(device)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            return None                               # (a first use inside a capture: this launch keeps the atomic path)
        ws = _WGRAD_WS[device] = torch.empty(WGRAD_WS_FLOATS, dtype=torch.float32, device=device)
    return ws


def _wgrad_groupable(q, dtype):
    return (q.get("ks", 1) == 1 and q.get("stride", 1) == 1 and q.get("pad", 0) == 0 and not q.get("nchw") and not q.get("up2")
            and q["N"] > 64 and q["N"] % 4 == 0 and q["Cin"] % 4 == 0 and q["lddu"] % 4 == 0 and q["ldx"] % 4 == 0
            and q["Hin"] == q["H"] and q["Win"] == q["W"] and q["x"].dtype == dtype)


class _WgradQueue:
    """Weight gradients are leaves of the backward pass: a plain-row problem of the 128 x 128 tile class whose destination is the gradient
    sink's persistent storage need not launch where the backward reaches it.  Such problems wait here (operands kept alive) and leave four
    at a time as ONE grouped launch + ONE fold (ly_wgrad_group) — the small-map problems (25600 pixels: ~30 us each alone, a few blocks per CU
    walking long pixel runs) share the chip — and the rest leaves when the pass ends (engine callback).  Program order fixes the grouping:
    the step stays bit-reproducible and a captured step replays the same launches.  `grad_done(param)` of a gradient that still waits here
    is passed on to the listeners (ddp.GradReducer) when its group has been launched (`notify`).  Narrow problems (Cin <= 64, or a column slice of a wider gradient) stay
    out: the group runs every problem on the 128 x 128 tile, and with the two K = 40 / 80 concat-slice problems of the neck in the groups
    the step went from 10.01 to 10.21 ms."""
    items = []
    notify = []         # (param, gradient data_ptr): grad_done arrived while the gradient's problem was waiting
    task = -1


WGRAD_DEFER = True                  # a module constant (tests monkeypatch it): not reachable from the environment
WGRAD_GROUP_MAX = 4
WGRAD_DEFER_MAX_MACS = 4 << 30      # M * N * K: a larger problem fills the chip alone, and the grouped kernel costs ~15 % more per unit of work than the
                                    # single-problem one (lead-yolo-l bs=16 1280x1280 with all its plain-row problems in groups: 33.96 -> 34.96 ms per step;
                                    # lead-yolo-s: every deferred problem is below 3.4 G)


def _wgrad_deferrable(q):
    return (WGRAD_DEFER and SINK is not None and q["x"].dtype == torch.bfloat16 and _wgrad_groupable(q, torch.bfloat16)
            and q["Cin"] > 64 and q["M"] * q["N"] * q["Cin"] <= WGRAD_DEFER_MAX_MACS and q.get("dw_off", 0) == 0 and SINK.is_target(q["dw"]) and _graph_task() >= 0)


def _wgrad_span(q):
    """byte range of the gradient storage a queued problem adds into"""
    lo = q["dw"].data_ptr() + 4 * q.get("dw_off", 0)
    return lo, lo + 4 * q["N"] * q["lddw"]


def _wgrad_enqueue(q):
    """A queued problem keeps REFERENCES to du / x / x_scale / x_shift and reads them when its group launches (after up to three more problems
    have collected, or when the backward pass ends): the caller must not write those tensors in place after ops.wgrad returns.  Their torch
    version counters are recorded here and checked at the launch; two problems adding into overlapping gradient storage never wait together
    (the group flushes them through one fold: the second would be lost / the order undefined)."""
    task = _graph_task()
    if _WgradQueue.task != task:
        _WgradQueue.items, _WgradQueue.notify, _WgradQueue.task = [], [], task    # (leftovers of a backward pass that raised are dropped with it)
        torch.autograd.Variable._execution_engine.queue_callback(wgrad_flush)
    lo, hi = _wgrad_span(q)
    if any(l2 < hi and lo < h2 for l2, h2 in map(_wgrad_span, _WgradQueue.items)):
        wgrad_flush(final=False)               # shared weights: the earlier problem leaves before this one is queued
    q["_versions"] = tuple(t._version for t in (q["du"], q["x"], q.get("x_scale"), q.get("x_shift")) if t is not None)
    _WgradQueue.items.append(q)
    # under a gradient listener (ddp.GradReducer) a waiting gradient holds back its whole bucket's exchange: leave in pairs there
    # (tools/dp_overlap_probe.py: with fours a 1.3 MB bucket was released at 77 % of the backward instead of 33 %)
    if len(_WgradQueue.items) >= (2 if GRAD_LISTENERS else WGRAD_GROUP_MAX):
        wgrad_flush(final=False)


def wgrad_flush(final=True):
    """launch what waits in the deferral queue (end of the backward pass, or four problems collected)"""
    items, _WgradQueue.items = _WgradQueue.items, []
    if final:
        _WgradQueue.task = -1
    for q in items:
        now = tuple(t._version for t in (q["du"], q["x"], q.get("x_scale"), q.get("x_shift")) if t is not None)
        if now != q.pop("_versions", now):
            raise RuntimeError("ops.wgrad: an operand of a deferred weight gradient was written in place before its launch (du / x / x_scale / "
                               "x_shift must stay untouched until the group leaves: see _wgrad_enqueue)")
    if len(items) == 1:
        wgrad(_now=True, **items[0])
    elif items:
        wgrad_group(items, _now=True)
    if _WgradQueue.notify:
        spans = [_wgrad_span(q) for q in _WgradQueue.items]
        waits = lambda ptr: any(lo <= ptr < hi for lo, hi in spans)
        ready = [prm for prm, ptr in _WgradQueue.notify if not waits(ptr)]
        _WgradQueue.notify = [(prm, ptr) for prm, ptr in _WgradQueue.notify if waits(ptr)]
        for prm in ready:
            for fn in GRAD_LISTENERS:
                fn(prm)


def wgrad_group(problems, _now=False):
    """Several independent weight gradients (each a dict of `wgrad` arguments) in ONE launch when all are plain-row 1x1 problems of the
    128 x 128 tile class (ly_wgrad_group: N > 64, vector-friendly widths, no gather, at most 4); otherwise one `wgrad` each."""
    if not _now and all(_wgrad_deferrable(q) for q in problems):
        for q in problems:
            _wgrad_enqueue(q)
        return
    if not (2 <= len(problems) <= 4) or not all(_wgrad_groupable(q, problems[0]["x"].dtype) for q in problems):
        for q in problems:
            wgrad(_now=_now, **q)
        return
    arr = (capi.LyWgradParams * len(problems))(*[_wgrad_params(**q) for q in problems])
    x0 = problems[0]["x"]
    px = 128 if x0.dtype == torch.bfloat16 else 64
    anyp = any(q.get("x_scale") is not None for q in problems)
    gslab = wgrad_workspace(x0.device) is not None                               # slab flush + ly_wgrad_combine_group_kernel (wgrad_group_launch)
    octs = gslab and all(_wgrad_octets(q["x"], q["du"], q["N"], q["Cin"], q["lddu"], q["ldx"], q.get("du_off", 0), q.get("x_off", 0)) for q in problems)
    with _Timed(f"ly_wgrad_tiled_group_kernel<{_tname(x0)}, 128, 128, {px}, true, {'true' if anyp else 'false'}, {'true' if gslab else 'false'}, "
                f"{'true' if octs else 'false'}>", sum(2.0 * q["M"] * q["N"] * q["Cin"] for q in problems),
                sum(x0.element_size() * q["M"] * (q["N"] + q["Cin"]) + 4.0 * q["N"] * q["Cin"] for q in problems)):
        capi.check(capi.lib().ly_wgrad_group(arr, len(problems), capi.stream_ptr()), "ly_wgrad_group")


def _wgrad_octets(x, du, n, cin, lddu, ldx, du_off=0, x_off=0):
    """mirror of wgrad_octets (csrc/ly_backward.hip): the 16-byte staging form applies — bf16 storage, widths / strides / addresses in whole
    8-channel vectors"""
    return (x.dtype == torch.bfloat16 and n % 8 == 0 and cin % 8 == 0 and lddu % 8 == 0 and ldx % 8 == 0
            and (du.data_ptr() + 2 * du_off) % 16 == 0 and (x.data_ptr() + 2 * x_off) % 16 == 0)


def wgrad_kernel_name(t, n, ktot, rows, tiled, pro=False, octets=False):
    """mirror of the tile dispatch in csrc/ly_backward.hip (wgrad_dispatch): the kernel name rocprofv3 prints"""
    r = "true" if rows else "false"
    if not tiled:
        return f"ly_wgrad_kernel<{t}, {r}>"
    px = 128 if t == "__bf16" else 64             # one 272-byte LDS row per channel and step in both dtypes, single buffer
    if n <= 64 and ktot <= 64:
        bn, bk = 64, 64
    elif n <= 32:
        bn, bk = 32, 128
    elif n <= 64:
        bn, bk = 64, 128
    else:
        bn, bk = 128, 128
    return f"ly_wgrad_tiled_kernel<{t}, {bn}, {bk}, {px}, {r}, {'true' if pro and rows else 'false'}, {'true' if octets else 'false'}>"


def sum_rows(t, out=None, accumulate=False):
    """sum over dim 0 of a contiguous fp32 tensor [R, ...] -> [...] (ly_sum_rows: fixed order, no atomics, no ATen reduction scratch)"""
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError("sum_rows: contiguous float32 tensor expected")
    r = t.shape[0]
    c = t.numel() // r
    if out is None:
        out = torch.empty(t.shape[1:], dtype=torch.float32, device=t.device)
    capi.check(capi.lib().ly_sum_rows(_p(t), r, c, c, _p(out), int(accumulate), capi.stream_ptr()), "ly_sum_rows")
    return out


def up2_bwd(d, ldd, n, hs, ws, c):
    out = empty_nhwc(n, c, hs, ws, d)
    capi.check(capi.lib().ly_up2_bwd(_p(d), ldd, n, hs, ws, c, _p(out), c, capi.dtype_code(d), capi.stream_ptr()), "ly_up2_bwd")
    return out


def unpatch(g, n, ho, wo, c, ks, h, w):
    """g [n*ho*wo, ks*ks*c] -> dx [n, c, h, w] (NHWC storage); rows/cols beyond ks*ho / ks*wo stay zero."""
    exact = (h == ho * ks and w == wo * ks)
    if not exact:
        raise NotImplementedError("patch-gather backward needs H, W divisible by the patch size")
    dx = empty_nhwc(n, c, h, w, g)
    capi.check(capi.lib().ly_unpatch(_p(g), n, ho, wo, c, ks, _p(dx), capi.dtype_code(g), capi.stream_ptr()), "ly_unpatch")
    return dx


def patch4_rows_u8(img, dtype):
    """uint8 NCHW image [n, c, h, w] -> rows [n*h/4*w/4, 16*c] of `dtype` holding the integer pixel values (PatchEmbed weight gradient)"""
    n, c, h, w = img.shape
    if img.dtype != torch.uint8 or not img.is_contiguous() or h % 4 or w % 4:
        raise ValueError("patch4_rows_u8: contiguous uint8 NCHW image with H, W multiples of 4 expected")
    rows = torch.empty((n * (h // 4) * (w // 4), 16 * c), dtype=dtype, device=img.device)
    capi.check(capi.lib().ly_patch4_rows_u8(_p(img), n, c, h, w, _p(rows), capi.dtype_code(rows), capi.stream_ptr()), "ly_patch4_rows_u8")
    return rows


def patch4_wgrad_u8_ok(img, du, n_out):
    """ly_patch4_wgrad_u8 applies: contiguous uint8 RGB batch with H, W multiples of 4, bf16 gradient rows of a vector-friendly width, and the
    weight-gradient workspace of the device holds the per-block partials"""
    n, c, h, w = img.shape
    ws = wgrad_workspace(img.device)
    return (img.dtype == torch.uint8 and img.is_contiguous() and c == 3 and h % 4 == 0 and w % 4 == 0 and du.dtype == torch.bfloat16
            and n_out % 8 == 0 and 8 <= n_out <= 64 and du.data_ptr() % 16 == 0 and ws is not None and ws.numel() >= 1536 * n_out * 48
            and n * (h // 4) * (w // 4) < (1 << 24))


def patch4_wgrad_u8(img, du, lddu, n_out, dw, scale):
    """dw [n_out][48] (fp32) += scale * sum over output pixels of du (x) the 4 x 4 patches of the uint8 image (csrc/ly_patch4.hip): the
    PatchEmbed weight gradient without the space-to-depth rows"""
    n, c, h, w = img.shape
    m = n * (h // 4) * (w // 4)
    ws = wgrad_workspace(img.device)
    with _Timed("ly_patch4_wgrad_u8_kernel", 2.0 * m * n_out * 48, 1.0 * n * c * h * w + 2.0 * m * n_out + 4.0 * n_out * 48):
        capi.check(capi.lib().ly_patch4_wgrad_u8(_p(img), n, c, h, w, _p(du), lddu, n_out, float(scale), _p(ws), ws.numel(), _p(dw), capi.dtype_code(du),
                                                 capi.stream_ptr()), "ly_patch4_wgrad_u8")


def coordatt_gate_bwd(dout, x, ldx, n, h, w, c, a_h, a_w, ldd=None):
    """dout: rows of the incoming gradient with row stride ldd (default c: dense) — a channel slice of a wider gradient is read in place"""
    ldd = c if ldd is None else ldd
    dx = empty_nhwc(n, c, h, w, x)
    # partial sums per column slab (da_h) / row band (da_w), plain stores, folded in index order: no atomics (csrc/ly_backward.hip)
    groups = 256 // (c // 4)
    slabs, bands = -(-w // (groups * 8)), -(-h // 8)
    ph = torch.empty((slabs, n * h * c), dtype=torch.float32, device=dout.device)
    pw = torch.empty((bands, n * w * c), dtype=torch.float32, device=dout.device)
    with _Timed(f"ly_coordatt_gate_bwd_kernel<{_tname(x)}>", 6.0 * n * h * w * c, 3.0 * x.element_size() * n * h * w * c):
        capi.check(capi.lib().ly_coordatt_gate_bwd(_p(dout), ldd, _p(x), ldx, n, h, w, c, _p(a_h), _p(a_w), _p(dx), c, _p(ph), _p(pw), bands, slabs,
                                                   capi.dtype_code(x), capi.stream_ptr()), "ly_coordatt_gate_bwd")
    da_h = (sum_rows(ph) if slabs > 1 else ph[0]).view(n, h, c)
    da_w = (sum_rows(pw) if bands > 1 else pw[0]).view(n, w, c)
    return dx, da_h, da_w


def pool_hw_bwd(gp, n, h, w, c, dtype=torch.float32, into=None):
    """into: a dense NHWC gradient the pool gradient is ADDED to in place (one pass instead of write + add)"""
    dx = into if into is not None else empty_nhwc(n, c, h, w, gp, dtype=dtype)
    capi.check(capi.lib().ly_pool_hw_bwd(_p(gp), n, h, w, c, _p(dx), c, 1 if into is not None else 0, capi.dtype_code(dx), capi.stream_ptr()),
               "ly_pool_hw_bwd")
    return dx


def maxpool_bwd(x, x_off, ldx, dy, dy_off, lddy, n, h, w, c, k, dx, dx_off, lddx):
    """x: the pooled map (storage dtype); dy / dx: ALWAYS float32 (the gradient is accumulated with float atomics)"""
    at = lambda t, off: ctypes.c_void_p(t.data_ptr() + t.element_size() * off)
    assert dy.dtype == torch.float32 and dx.dtype == torch.float32
    capi.check(capi.lib().ly_maxpool_bwd(at(x, x_off), ldx, at(dy, dy_off), lddy, n, h, w, c, k, at(dx, dx_off), lddx, capi.dtype_code(x),
                                         capi.stream_ptr()), "ly_maxpool_bwd")
