"""Host-side weight packing into the fragment order the gfx950 kernels read (see csrc/ly_common.hpp).

`frag_pack(W[R, K])` -> flat fp32 [T*S*256]:  out[((t*S + s)*64 + lane)*4 + j] =
W[16t + (lane & 15)][16s + 4*(lane >> 4) + j], zero padded to 16-multiples.  One wave-wide weight
fragment is then one contiguous 1 KiB read."""
import weakref

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
    """MFMA operand packing (csrc/ly_tile.hpp): W[R, K] fp32 -> int16 tensor [T, S, planes, 64, 8]; planes = 2: hi = bf16(W),
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


def frag_pack_nat(w2d, planes=2):
    """MFMA A-operand packing in NATURAL k order (csrc/ly_detect.hip: the B operand is loaded straight from global memory, 8 consecutive
    channels per lane): W[R, K] fp32 -> int16 [T, S, planes, 64, 8], lane = g*16 + i holds row 16t + i and k = 32s + 8g + j, j = 0..7."""
    r, k = w2d.shape
    t, s = _ceil(r, 16), _ceil(k, 32)
    wp = torch.zeros(t * 16, s * 32, dtype=torch.float32, device=w2d.device)
    wp[:r, :k] = w2d.float()
    hi = wp.to(torch.bfloat16)
    lo = (wp - hi.float()).to(torch.bfloat16)
    pl = torch.stack((hi, lo), 0)[:planes]                              # [planes, T*16, S*32]
    # [p, t, i, s, g, j] -> [t, s, p, g, i, j]
    v = pl.view(planes, t, 16, s, 4, 8).permute(1, 3, 0, 4, 2, 5).contiguous()
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


def rfcbam_gen_weights_c(gen_w, scale, shift, raw=False):
    """Depthwise 'generate' weights of RFCBAMConv (k=3) in LANE = CHANNEL order (csrc/ly_rf3c.hpp rc_load_w): per channel 100 floats —
    w[t][u] at t*9 + u, b[t] at 81 + t, a[t] at 90 + t, one of padding — stored as float [C/32][25][32][4] so that a half wave's 16-byte
    loads are contiguous.  Folded (inference): w = weight * scale, b = shift;  raw (training): w = weight, a = scale, b = shift."""
    c = gen_w.shape[0] // 9
    out = torch.zeros(c, 100, dtype=torch.float32, device=gen_w.device)
    wv = gen_w.detach().float().view(c, 9, 9)
    out[:, :81] = (wv if raw else wv * scale.view(c, 9, 1)).reshape(c, 81)
    out[:, 81:90] = shift.view(c, 9)
    out[:, 90:99] = scale.view(c, 9) if raw else 1.0
    return out.view(c // 32, 32, 25, 4).permute(0, 2, 1, 3).contiguous().view(-1)


def rf3m_stream(gen_w, scale, shift, conv_w=None, mt=4, pool_stride=None):
    """Weight stream of csrc/ly_rf3m.hip (RFCBAMConv k=3, `generate` on the matrix cores; models/rfa.py:101-106, 110, 121-128): every A fragment
    of v_mfma_f32_32x32x16_bf16 (64 lanes x 8 bf16 = 1 KiB; lane = (row r = lane & 31, half h = lane >> 5), element e <-> k = 8h + e) in the
    order the kernel consumes them, bf16.

    Per 16-channel chunk, 4 units (4-channel groups j):
      3 generate fragments of the (channel 4j + r//8, tap r%8) row tile: k-step st, patch slot u = 4st + 2h + e//4, channel slot e%4 —
        the value is generate.0.weight[c, t, u] * bn_scale[c, t] where the channel slot is the row's channel, else 0; slots 9, 10, 11 carry the
        BatchNorm shift split hi + lo + lolo (their B operand is 1.0);
      3 generate fragments of the chunk's tap-8 row tile (row r < 16 = channel r of the chunk), non-zero only for the unit's 4 channels;
      [conv_w given] 2 x mt main fragments: conv.0.weight[o, c, t] with o = 32 (g mt + m) + r and (c, t) = the generated row
        16 s2 + 8 (e//4) + 4h + e%4 of the unit's tile (the accumulator layout the generate product leaves in registers);
    then [conv_w given] mt main fragments for the tap-8 tile (one k-step: rows 8 (e//4) + 4h + e%4 = channels of the chunk).
    conv_w None: the statistics stream (generate fragments only).  With pool_stride = s (the statistics stream) rows 16 .. 31 of the tap-8
    tile — unused by the generate — carry the SE POOLING: row 16 + r (channel r of the chunk) holds 1.0 at the patch slots of the input
    positions an output pixel OWNS ((s oy + dy, s ox + dx), dy, dx < s: slots 4, 5, 7, 8 at stride 2, slot 4 at stride 1), so the product
    leaves, per pixel, the sum of its own inputs per channel (models/rfa.py:90's average pool, before the division).
    Returns bf16 [gy][chunks][fragments][64][8] (gy = O / (32 mt) blocks of output channels; 1 for the statistics stream)."""
    dev = gen_w.device
    c_in = gen_w.shape[0] // 9
    nch = c_in // 16
    wx = torch.zeros(c_in, 9, 12, dtype=torch.float32, device=dev)
    wx[:, :, :9] = gen_w.detach().float().view(c_in, 9, 9) * scale.view(c_in, 9, 1)
    b = shift.detach().float().view(c_in, 9)
    bh = b.bfloat16().float()
    bl = (b - bh).bfloat16().float()
    wx[:, :, 9], wx[:, :, 10], wx[:, :, 11] = bh, bl, (b - bh - bl).bfloat16().float()
    ar = lambda n: torch.arange(n, device=dev)
    lane = ar(64)
    R, H, E = (lane & 31).view(1, 1, 1, 64, 1), (lane >> 5).view(1, 1, 1, 64, 1), ar(8).view(1, 1, 1, 1, 8)
    CH, J, ST = ar(nch).view(nch, 1, 1, 1, 1), ar(4).view(1, 4, 1, 1, 1), ar(3).view(1, 1, 3, 1, 1)
    us, cs = 4 * ST + 2 * H + (E >> 2), E & 3
    zero = torch.zeros((), dtype=torch.float32, device=dev)
    gen = torch.where(cs == (R >> 3), wx[16 * CH + 4 * J + (R >> 3), R & 7, us], zero)                      # [nch, 4, 3, 64, 8]
    gen8 = torch.where((R < 16) & ((R >> 2) == J) & (cs == (R & 3)), wx[16 * CH + (R & 15) + 0 * J, 8, us], zero)
    if pool_stride is not None:
        if conv_w is not None or pool_stride not in (1, 2):
            raise ValueError("rf3m_stream: pooling rows belong to the statistics stream, stride 1 or 2")
        owned = torch.zeros(12, dtype=torch.bool, device=dev)
        owned[[4, 5, 7, 8] if pool_stride == 2 else [4]] = True
        Rp = R - 16
        one = torch.ones((), dtype=torch.float32, device=dev)
        pool = torch.where((R >= 16) & ((Rp >> 2) == J) & (cs == (Rp & 3)) & owned[us] & (CH >= 0), one, zero)
        gen8 = gen8 + pool
    genf = torch.cat((gen, gen8), 2)                                                                        # [nch, 4, 6, 64, 8]
    if conv_w is None:
        return genf.reshape(1, nch, 24, 64, 8).to(torch.bfloat16).contiguous()
    o_ch = conv_w.shape[0]
    gy = o_ch // (32 * mt)
    wc = conv_w.detach().float().view(o_ch, c_in, 9)
    S2, M = ar(2).view(1, 1, 1, 2, 1, 1, 1), ar(mt).view(1, 1, 1, 1, mt, 1, 1)
    G, CH7, J7 = ar(gy).view(gy, 1, 1, 1, 1, 1, 1), ar(nch).view(1, nch, 1, 1, 1, 1, 1), ar(4).view(1, 1, 4, 1, 1, 1, 1)
    R7, H7, E7 = (lane & 31).view(1, 1, 1, 1, 1, 64, 1), (lane >> 5).view(1, 1, 1, 1, 1, 64, 1), ar(8).view(1, 1, 1, 1, 1, 1, 8)
    row = 16 * S2 + 8 * (E7 >> 2) + 4 * H7 + (E7 & 3)
    o_idx = 32 * (G * mt + M) + R7
    main = wc[o_idx, 16 * CH7 + 4 * J7 + (row >> 3), row & 7]                                               # [gy, nch, 4, 2, mt, 64, 8]
    row8 = 8 * (E7 >> 2) + 4 * H7 + (E7 & 3)                                                                # rows 0 .. 15 of the tap-8 tile
    main8 = wc[o_idx[:, :, :1, :1], 16 * CH7 + row8, 8]                                                     # [gy, nch, 1, 1, mt, 64, 8]
    units = torch.cat((genf.view(1, nch, 4, 6, 64, 8).expand(gy, -1, -1, -1, -1, -1), main.reshape(gy, nch, 4, 2 * mt, 64, 8)), 3)
    out = torch.cat((units.reshape(gy, nch, 4 * (6 + 2 * mt), 64, 8), main8.reshape(gy, nch, mt, 64, 8)), 2)
    return out.to(torch.bfloat16).contiguous()


# --------------------------------------------------------------------------------------------------
# Batched packing (csrc/ly_backward.hip ly_pack_table): every packed weight image of the model — forward, transposed for dgrad,
# tap-flipped, concatenated — is described ONCE by how it reads the fp32 parameter in place, and all of them are refreshed by a
# single launch per optimisation step instead of 113-144 ly_frag_pack3 launches plus the permute / pad / cat temporaries feeding
# them.  `packed(parts, K, planes)` returns the (persistent) packed tensor; it is refreshed lazily, when any source parameter
# changed since the last refresh (torch version counters, or W_EPOCH for writes through raw pointers: optim.FusedSGD, graph replays).
# --------------------------------------------------------------------------------------------------
W_EPOCH = 0


def touch_weights():
    """weights were written through raw pointers (fused optimiser step, replayed graph): packed images must be refreshed"""
    global W_EPOCH
    W_EPOCH += 1
    touch()


_SHADOW = {}


def master(param):
    """the contiguous float32 tensor the packer reads: the parameter itself, or — for `.half()` / `.bfloat16()` models (inference) — a
    cached float32 copy, rebuilt when the parameter changes"""
    if param.dtype == torch.float32 and param.is_contiguous():
        return param
    ent = _SHADOW.get(id(param))
    if ent is None or ent[0]() is not param or ent[1] != param._version:
        sh = param.detach().float().contiguous()
        ref = weakref.ref(param, lambda _r, k=id(param): _SHADOW.pop(k, None))
        _SHADOW[id(param)] = ent = (ref, param._version, sh)
    return ent[2]


class Src:
    """rows x K matrix read in place from parameter `param` (contiguous fp32):  row r = ra*nrb + rb, column k = (a*nb + b)*nc + c,
    element = param.flat[base + ra*sra + rb*srb + a*sa + b*sb + c*sc] for r < rows, b < vb, c < vc (else 0)."""

    def __init__(self, param, rows, base=0, nrb=None, sra=0, srb=0, nb=1, nc=1, vb=None, vc=None, sa=0, sb=0, sc=1):
        param = master(param)
        self._ref = weakref.ref(param)             # the plan must not keep dead models' parameters alive
        self.ptr, self.device = param.data_ptr(), param.device
        self.ok = param.is_cuda and param.dtype == torch.float32 and param.is_contiguous()
        self.rows, self.base = rows, base
        self.nrb = rows if nrb is None else nrb
        self.sra, self.srb = sra, srb
        self.nb, self.nc = nb, nc
        self.vb, self.vc = (nb if vb is None else vb), (nc if vc is None else vc)
        self.sa, self.sb, self.sc = sa, sb, sc
        self._ref2 = None                          # a second parameter reached through the (a, b) strides (src_matrix_kcat)

    @property
    def param(self):
        p = self._ref()
        if self._ref2 is None or p is None:
            return p
        q = self._ref2()
        return p if q is not None and self._pair_ok(p, q) else None

    def _pair_ok(self, p, q):
        return (q.data_ptr() - p.data_ptr()) == 4 * self.sb and q.is_cuda and q.dtype == torch.float32 and q.is_contiguous()

    def version(self):
        p = self._ref()
        if p is None:
            return -1
        if self._ref2 is None:
            return p._version
        q = self._ref2()
        return (p._version, q._version if q is not None else -1)

    def sig(self):
        return (self.ptr, self.rows, self.base, self.nrb, self.sra, self.srb, self.nb, self.nc, self.vb, self.vc, self.sa, self.sb, self.sc)


def src_matrix(param, rows, cols, sr=None, sk=1, k_pad=None):
    """W[r][k] = param.flat[r*sr + k*sk] (a 2-D view of the parameter: plain or transposed), columns zero-padded to k_pad"""
    return Src(param, rows, srb=cols * sk if sr is None else sr, nc=(k_pad or cols), vc=cols, sc=sk)


def src_matrix_kcat_t(p1, p2, r0=0, r1=None):
    """[kin, 2*c_] matrix [W1^T | W2^T] of two [c_, kin] weights (the data-gradient operand of two 1x1 convolutions over one input, C3_CA's
    cv1 / cv2), read in place from BOTH parameters: the column-block stride is the distance between the two allocations.  None when that
    is not expressible (different devices / dtypes, a `.half()` shadow copy).  r0 / r1: rows [r0, r1) of it only (one source of a
    concatenated input)."""
    m1, m2 = master(p1), master(p2)
    if m1 is not p1 or m2 is not p2 or p1.shape != p2.shape or p1.device != p2.device or (m2.data_ptr() - m1.data_ptr()) % 4:
        return None
    c_, kin = p1.shape[0], p1.numel() // p1.shape[0]
    r1 = kin if r1 is None else r1
    s = Src(p1, r1 - r0, base=r0, srb=1, nb=2, nc=c_, sb=(m2.data_ptr() - m1.data_ptr()) // 4, sc=kin)
    s._ref2 = weakref.ref(p2)
    s.ok = s.ok and p2.is_cuda and p2.dtype == torch.float32 and p2.is_contiguous()
    return s


def src_taps(w, cip, transposed_flipped=False):
    """pack.conv_taps_matrix(w, cip) of w [co, ci, kh, kw]: k = tap*cip + c; transposed_flipped: of w.permute(1,0,2,3).flip(2,3)
    (the operand of the stride-1 convolution's data gradient)"""
    co, ci, kh, kw = w.shape
    t = kh * kw
    if not transposed_flipped:
        return Src(w, co, srb=ci * t, nb=t, nc=cip, vc=ci, sb=1, sc=t)
    return Src(w, ci, base=t - 1, srb=t, nb=t, nc=cip, vc=co, sb=-1, sc=ci * t)


class _Plan:
    def __init__(self):
        self.entries = {}        # key -> dict(out, parts, K, planes, versions)
        self.table = None
        self.w_epoch = -1
        self.trace = None        # set(): keys requested while a step is being traced (train.GraphedTrainStep's last warm-up step)
        self.capture_table = None  # (tab, blk, n): the private table a capture launches instead of the global one

    def private_table(self, keys):
        """A descriptor table over `keys` only, for a hipGraph capture: the GLOBAL table is rebuilt (and its device tensors freed) whenever any
        model in the process registers a new image or dies — a captured ly_pack_table launch that addressed it would then read freed memory
        (seen as an intermittent GPU memory fault: a replayed training graph after an eval forward / a garbage-collected model of an earlier
        test).  The caller keeps the returned tensors (and the images they address) alive as long as the graph."""
        keys = [k for k in keys if k in self.entries]
        if not keys:
            return None, []
        return self._build_table(keys), [self.entries[k]["out"] for k in keys]

    def _descs(self, e, blk0):
        from . import capi
        out, K, planes = e["out"], e["K"], e["planes"]
        S = _ceil(K, 32)
        descs, t0 = [], 0
        for part in e["parts"]:
            T = _ceil(part.rows, 16) if part is not e["parts"][-1] else out.shape[0] - t0
            blocks = _ceil(T * S * 64, 256)
            descs.append((capi.LyPackDesc(part.ptr + 4 * part.base, out.data_ptr(), part.rows, K, planes, S, t0, T, part.nrb, part.nb, part.nc,
                                          part.vb, part.vc, part.sra, part.srb, part.sa, part.sb, part.sc, blk0), blocks))
            blk0 += blocks
            t0 += T
        return descs, blk0

    def _build_table(self, keys):
        from . import capi
        descs, blk0 = [], 0
        for k in keys:
            d, blk0 = self._descs(self.entries[k], blk0)
            descs += d
        arr = (capi.LyPackDesc * len(descs))(*[d for d, _ in descs])
        dev = self.entries[keys[0]]["out"].device
        tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone().to(dev)
        blk = torch.tensor([i for i, (_, nb) in enumerate(descs) for _ in range(nb)], dtype=torch.int32, device=dev)
        return tab, blk, blk0

    def _launch(self, table):
        from . import capi
        tab, blk, n = table
        capi.check(capi.lib().ly_pack_table(capi.ptr(tab), capi.ptr(blk), n, capi.stream_ptr()), "ly_pack_table")

    def get(self, parts, K, planes, rows_to=0):
        key = (tuple(p.sig() for p in parts), K, planes, rows_to)
        if self.trace is not None:
            self.trace.add(key)
        e = self.entries.get(key)
        capturing = torch.cuda.is_current_stream_capturing()
        if e is None:
            if capturing:
                raise RuntimeError("pack: a new packed-weight layout was requested during hipGraph capture; run one eager step first")
            dev = parts[0].device
            for p in parts[:-1]:
                if p.rows % 16:
                    raise ValueError("pack: stacked parts must have a multiple of 16 rows")
            rows = sum(p.rows for p in parts)
            out = torch.empty((_ceil(max(rows, rows_to), 16), _ceil(K, 32), planes, 64, 8), dtype=torch.int16, device=dev)
            e = dict(out=out, parts=parts, K=K, planes=planes, versions=None)
            self.entries[key] = e
            self.table = None
            self._launch(self._build_table([key]))                # this image now; it joins the batched refresh from the next change on
            e["versions"] = tuple(p.version() for p in parts)
            return out
        # (inside a captured training step the first request finds the plan stale — an optimiser step always precedes the capture —
        # so exactly one refresh launch is captured, ahead of the forward)
        if any(old.param is not new.param for old, new in zip(e["parts"], parts)):
            # same address, same layout, but ANOTHER tensor (the allocator handed a freed parameter's storage to a new model): the
            # descriptors (raw pointers) still hold, the contents do not
            e["parts"] = parts
            e["versions"] = None
        if self.w_epoch != W_EPOCH or e["versions"] != tuple(p.version() for p in parts):
            self.refresh(capturing)
        return e["out"]

    def refresh(self, capturing=False):
        """ONE launch over every registered image"""
        if not capturing:
            dead = [k for k, e in self.entries.items() if any(p.param is None for p in e["parts"])]
            if dead:
                for k in dead:
                    del self.entries[k]
                self.table = None
        if capturing and self.capture_table is not None:
            self._launch(self.capture_table)               # the capture's own table (its owner pins it): only the traced model's images
            for e in self.entries.values():
                e["versions"] = tuple(p.version() for p in e["parts"])
            self.w_epoch = W_EPOCH
            return
        if self.table is None:
            if capturing:
                raise RuntimeError("pack: the descriptor table changed during hipGraph capture")
            by_dev = {}
            for k, e in self.entries.items():
                by_dev.setdefault(e["out"].device, []).append(k)
            self.table = {dev: self._build_table(keys) for dev, keys in by_dev.items()}
        cur = torch.cuda.current_device()
        for dev, t in self.table.items():
            if dev.index == cur or len(self.table) == 1:
                self._launch(t)
        for e in self.entries.values():
            e["versions"] = tuple(p.version() for p in e["parts"])
        self.w_epoch = W_EPOCH


PLAN = _Plan()


def packed(parts, K, planes=2, rows_to=0):
    """packed fragment image [T, S, planes, 64, 8] (int16 view of bf16) of the matrix described by `parts` (Src or list of Src stacked by
    rows), refreshed together with every other registered image by one ly_pack_table launch whenever a source parameter changed"""
    if isinstance(parts, Src):
        parts = [parts]
    for p in parts:
        if not p.ok:
            raise ValueError("pack.packed: sources must be contiguous float32 CUDA parameters")
    return PLAN.get(list(parts), K, planes, rows_to)
