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
