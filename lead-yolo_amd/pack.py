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


def versions(*tensors):
    return tuple((t.data_ptr(), t._version) if t is not None else None for t in tensors)


def frag_pack3(w2d, rows_to=0):
    """bf16x3 operand packing (csrc/ly_tile.cuh): W[R, K] fp32 -> int16 tensor [T, S, 2, 64, 8]
    (planes hi = bf16(W), lo = bf16(W - hi)); lane = q*16 + i holds row 16t + i and
    k = 32s + 16*(j >> 2) + 4q + (j & 3), j = 0..7.  Zero padded to 16 x 32 multiples."""
    r, k = w2d.shape
    t, s = _ceil(max(r, rows_to), 16), _ceil(k, 32)
    wp = torch.zeros(t * 16, s * 32, dtype=torch.float32, device=w2d.device)
    wp[:r, :k] = w2d.float()
    hi = wp.to(torch.bfloat16)
    lo = (wp - hi.float()).to(torch.bfloat16)
    planes = torch.stack((hi, lo), 0)                                   # [2, T*16, S*32]
    # [p, t, i, s, jh, q, jl] -> [t, s, p, q, i, jh, jl]
    v = planes.view(2, t, 16, s, 2, 4, 4).permute(1, 3, 0, 5, 2, 4, 6).contiguous()
    return v.view(t, s, 2, 64, 8).view(torch.int16)


def pad_to(v, n):
    out = torch.zeros(n, dtype=torch.float32, device=v.device)
    out[:v.numel()] = v
    return out
