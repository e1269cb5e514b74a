"""Host-side weight packing into the fragment order the gfx950 kernels read (see csrc/ly_common.cuh).

`frag_pack(W[R, K])` -> flat fp32 [T*S*256]:  out[((t*S + s)*64 + lane)*4 + j] =
W[16t + (lane & 15)][16s + 4*(lane >> 4) + j], zero padded to 16-multiples.  One wave-wide weight
fragment is then one contiguous 1 KiB read."""
import torch


def _ceil(a, b):
    return (a + b - 1) // b


def frag_pack(w2d):
    r, k = w2d.shape
    t, s = _ceil(r, 16), _ceil(k, 16)
    wp = torch.zeros(t * 16, s * 16, dtype=torch.float32, device=w2d.device)
    wp[:r, :k] = w2d.float()
    # [t, i, s, q, j] -> [t, s, q, i, j]; lane = q*16 + i
    return wp.view(t, 16, s, 4, 4).permute(0, 2, 3, 1, 4).contiguous().view(-1)


def conv_taps_matrix(w4d, cin_pad_to=4):
    """[co, ci, kh, kw] -> [co, kh*kw*ci_p] with k = tap*ci_p + ci (tap = ky*kw + kx), ci zero-padded."""
    co, ci, kh, kw = w4d.shape
    cip = _ceil(ci, cin_pad_to) * cin_pad_to
    if cip == ci:                                     # nothing to pad: one permuting copy, no fill
        return w4d.permute(0, 2, 3, 1).float().reshape(co, kh * kw * ci)
    m = torch.zeros(co, kh, kw, cip, dtype=torch.float32, device=w4d.device)
    m[..., :ci] = w4d.permute(0, 2, 3, 1).float()
    return m.view(co, kh * kw * cip)


def bn_scale_shift(bn, conv_bias=None):
    """Eval-mode BatchNorm as y = x*scale + shift (running stats), optionally absorbing a conv bias."""
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    shift = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
    if conv_bias is not None:
        shift = shift + conv_bias.detach().float() * scale
    return scale.contiguous(), shift.contiguous()


EPOCH = 0


def touch():
    """Parameters or BatchNorm statistics were just written by a HIP kernel through raw pointers (optim.FusedSGD, ly_bn_finalize, a
    replayed hipGraph): torch's per-tensor version counters did not move, so every cache of packed / folded parameters keyed by
    `versions()` is invalidated through this global epoch instead."""
    global EPOCH
    EPOCH += 1


def versions(*tensors):
    return tuple((t.data_ptr(), t._version) if t is not None else None for t in tensors) + (EPOCH,)


def frag_pack3(w2d, rows_to=0, planes=2):
    """MFMA operand packing (csrc/ly_tile.cuh): W[R, K] fp32 -> int16 tensor [T, S, planes, 64, 8]; planes = 2: hi = bf16(W),
    lo = bf16(W - hi) (the bf16x3 operand of the fp32-storage kernels), planes = 1: hi only (bf16-storage kernels).
    lane = q*16 + i holds row 16t + i and k = 32s + 16*(j >> 2) + 4q + (j & 3), j = 0..7.  Zero padded to 16 x 32 multiples."""
    r, k = w2d.shape
    t, s = _ceil(max(r, rows_to), 16), _ceil(k, 32)
    if w2d.is_cuda:                                   # one HIP launch (ly_frag_pack3); strides cover transposed views
        from . import capi
        if w2d.dtype != torch.float32:
            w2d = w2d.float()
        out = torch.empty((t, s, planes, 64, 8), dtype=torch.int16, device=w2d.device)
        capi.check(capi.lib().ly_frag_pack3(capi.ptr(w2d), r, k, w2d.stride(0), w2d.stride(1), rows_to, planes, capi.ptr(out), capi.stream_ptr()),
                   "ly_frag_pack3")
        return out
    wp = torch.zeros(t * 16, s * 32, dtype=torch.float32, device=w2d.device)
    wp[:r, :k] = w2d.float()
    hi = wp.to(torch.bfloat16)
    lo = (wp - hi.float()).to(torch.bfloat16)
    pl = torch.stack((hi, lo), 0)[:planes]                              # [planes, T*16, S*32]
    # [p, t, i, s, jh, q, jl] -> [t, s, p, q, i, jh, jl]
    v = pl.view(planes, t, 16, s, 2, 4, 4).permute(1, 3, 0, 5, 2, 4, 6).contiguous()
    return v.view(t, s, planes, 64, 8).view(torch.int16)


def pad_to(v, n):
    out = torch.zeros(n, dtype=torch.float32, device=v.device)
    out[:v.numel()] = v
    return out


def rfcbam_gen_weights(gen_w, scale, shift, chunk, per_wave_contiguous):
    """Folded depthwise 'generate' weights of RFCBAMConv (k=3) in the order the kernels read them from
    LDS: [C_pad/chunk][4 waves][9 taps t][chunk/8 channel pairs][20] where the 20 floats are the 9
    interleaved pairs (W'_a[t][u], W'_b[t][u]) followed by (b'_a[t], b'_b[t]) — five aligned float4 that
    feed v_pk_fma_f32 (two channels per instruction).
    gen_w: generate.0.weight [C*9, 1, 3, 3]; scale/shift: folded generate.1 BN, [C*9].
    Channel of (chunk q, wave w, slot j): q*chunk + (4*w + j if per_wave_contiguous else w + 4*j); pair p =
    slots (2p, 2p+1).  Channels are zero padded to a multiple of `chunk` (zero weights => G = 0, neutral
    for the max/mean of ReLU outputs)."""
    c = gen_w.shape[0] // 9
    wt = torch.cat(((gen_w.detach().float().view(c, 9, 9) * scale.view(c, 9, 1)), shift.view(c, 9, 1)), 2)   # [c][t][10]
    cp = _ceil(c, chunk) * chunk
    full = torch.zeros(cp, 9, 10, dtype=torch.float32, device=gen_w.device)
    full[:c] = wt
    per = chunk // 4
    if per_wave_contiguous:
        v = full.view(cp // chunk, 4, per, 9, 10)                          # [q][w][j][t][10]
    else:
        v = full.view(cp // chunk, per, 4, 9, 10).permute(0, 2, 1, 3, 4)   # [q][j][w] -> [q][w][j]
    v = v.reshape(cp // chunk, 4, per // 2, 2, 9, 10)                      # [q][w][p][ab][t][10]
    v = v.permute(0, 1, 4, 2, 5, 3).contiguous()                           # [q][w][t][p][10][ab]
    return v.view(-1)
