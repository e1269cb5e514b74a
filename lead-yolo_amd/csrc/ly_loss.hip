// Detection loss of one pyramid level, forward AND gradient in three launches (reference utils/loss.py:121-268
// `ComputeLoss.__call__` / `build_targets`, utils/metrics.py:293-354 `bbox_iou(EIoU=True)`); any nc (class BCE with label smoothing and
// cls_pw when nc > 1, utils/loss.py:168-173), obj_pw, no focal loss.
//
// The reference runs ~150 tiny ops forward and ~250 backward per step here (anchor matching with boolean indexing — a
// host sync per level —, EIoU, two BCEs): 3.4 ms + 4.2 ms of pure launch latency at bs=64, a fifth of the training step.
//
//   ly_loss_match   one thread per CANDIDATE (offset k in 0..4, anchor a, target t), order index = k*na*nt + a*nt + t — exactly
//                   the order the reference's `t.repeat((5,1,1))[sel]` produces.  Valid candidates (anchor ratio test, offset
//                   rule) compute their cell, the target box, the predicted box and EIoU WITH its gradient (forward-mode dual
//                   numbers over the 4 box parameters), add (1 - eiou) and 1 to the level's accumulators, and enter the
//                   per-cell "last writer" election (atomicMax of the order index: the reference's `tobj[b,a,gj,gi] = iou`
//                   assigns sequentially on the CPU, so the highest order index wins; on CUDA the reference itself is unordered).
//   ly_loss_apply   per valid candidate: box gradient * box*bs/count into dpred (atomic: several candidates can share a cell),
//                   and the elected candidate writes tobj = max(iou, 0).
//   ly_loss_obj     per cell: BCE-with-logits against tobj, its gradient into dpred[..., 4], level sum by block reduction.
// ly_loss_finish combines the level accumulators into (loss, lbox, lobj, lcls) exactly as the reference does (mean per level,
// balance [4, 1, 0.4], * hyp gains, * batch size).
#include "ly_common.hpp"

struct D4 {                        // value + derivatives w.r.t. (px, py, pw, ph)
  float v, d[4];
};
__device__ __forceinline__ D4 d4c(float v) { return D4{v, {0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ D4 d4v(float v, int i) { D4 r = d4c(v); r.d[i] = 1.f; return r; }
__device__ __forceinline__ D4 operator+(D4 a, D4 b) { D4 r; r.v = a.v + b.v; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
__device__ __forceinline__ D4 operator-(D4 a, D4 b) { D4 r; r.v = a.v - b.v; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
__device__ __forceinline__ D4 operator*(D4 a, D4 b) { D4 r; r.v = a.v * b.v; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
__device__ __forceinline__ D4 operator/(D4 a, D4 b) {
  D4 r; const float ib = 1.f / b.v; r.v = a.v * ib;
  for (int i = 0; i < 4; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib;
  return r;
}
__device__ __forceinline__ D4 operator*(D4 a, float s) { D4 r; r.v = a.v * s; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * s; return r; }
__device__ __forceinline__ D4 operator+(D4 a, float s) { a.v += s; return a; }
__device__ __forceinline__ D4 d4min(D4 a, float b) { return a.v <= b ? a : d4c(b); }       // torch.minimum: gradient to the smaller (ties: first)
__device__ __forceinline__ D4 d4max(D4 a, float b) { return a.v >= b ? a : d4c(b); }
__device__ __forceinline__ D4 d4clamp0(D4 a) { return a.v > 0.f ? a : d4c(0.f); }         // clamp(0): zero gradient at and below 0

// EIoU of the predicted box (px, py, pw, ph: dual) against the target box, as bbox_eiou in loss.py (reference metrics.py:293-354)
__device__ __forceinline__ D4 ly_eiou(D4 px, D4 py, D4 pw, D4 ph, float tx, float ty, float tw, float th) {
  const float eps = 1e-7f;
  const D4 ax1 = px - pw * 0.5f, ax2 = px + pw * 0.5f, ay1 = py - ph * 0.5f, ay2 = py + ph * 0.5f;
  const float bx1 = tx - tw * 0.5f, bx2 = tx + tw * 0.5f, by1 = ty - th * 0.5f, by2 = ty + th * 0.5f;
  const D4 iw = d4clamp0(d4min(ax2, bx2) - d4max(ax1, bx1)), ih = d4clamp0(d4min(ay2, by2) - d4max(ay1, by1));
  const D4 inter = iw * ih;
  const D4 uni = pw * ph + (tw * th) - inter + eps;
  const D4 iou = inter / (uni + eps);
  const D4 cw = d4max(ax2, bx2) - d4min(ax1, bx1), ch = d4max(ay2, by2) - d4min(ay1, by1);
  const D4 c2 = cw * cw + ch * ch + eps;
  const D4 sx = d4c(bx1 + bx2) - ax1 - ax2, sy = d4c(by1 + by2) - ay1 - ay2;
  const D4 rho2 = (sx * sx + sy * sy) * 0.25f;
  const D4 dw = d4c(bx2 - bx1) - (ax2 - ax1), dh = d4c(by2 - by1) - (ay2 - ay1);
  return iou - (rho2 / c2 + (dw * dw) / (cw * cw + eps) + (dh * dh) / (ch * ch + eps));
}

struct LyLossLevel {
  const float* p;        // [bs][na][ny][nx][no]
  float* dp;             // same shape, zeroed by the caller
  const float* anchors;  // [na][2], grid units
  const float* targets;  // [nt][6]: image, class, x, y, w, h (normalised)
  int bs, na, ny, nx, no;
  long nt;
  float anchor_t;
  // workspaces (per level)
  float* tobj;           // [cells], zeroed
  int* winner;           // [cells], filled with -1
  long* cand_cell;       // [5*na*nt], -1 = invalid
  float* cand;           // [5*na*nt][5]: iou, d(1-eiou)/d(raw0..3)
  float* acc;            // [8]: sum(1-eiou), count, sum BCE obj, number of rejected target rows, sum BCE cls, 3 unused   (zeroed)
  float* tbox;           // optional [5*na*nt][4]: (gx - gi, gy - gj, gw, gh) of every valid candidate (build_targets' tbox)
  float cp, cn, cls_pw, obj_pw;      // class targets (label smoothing: 1 - eps/2, eps/2), positive weights of the two BCEs
};

// BCE-with-logits with a positive weight (torch.nn.BCEWithLogitsLoss(pos_weight = pw)): value and d/dx
__device__ __forceinline__ float ly_bce(float x, float z, float pw) {
  const float sp = log1pf(__expf(-fabsf(x))) + fmaxf(-x, 0.f);         // softplus(-x) = -log sigmoid(x)
  return (1.f - z) * x + (1.f + (pw - 1.f) * z) * sp;
}
__device__ __forceinline__ float ly_bce_grad(float x, float z, float pw) {
  const float sg = ly_sigmoid(x);
  return (1.f - z) * sg - pw * z * (1.f - sg);
}

__global__ __launch_bounds__(LY_THREADS) void ly_loss_match_kernel(const LyLossLevel L) {
  const long ncand = 5L * L.na * L.nt;
  const long idx = (long)blockIdx.x * LY_THREADS + threadIdx.x;
  if (idx >= ncand) return;
  const int k = (int)(idx / ((long)L.na * L.nt));
  const long rem = idx - (long)k * L.na * L.nt;
  const int a = (int)(rem / L.nt);
  const long t = rem - (long)a * L.nt;
  const float* tg = L.targets + t * 6;
  const float gx = __fmul_rn(tg[2], (float)L.nx), gy = __fmul_rn(tg[3], (float)L.ny), gw = __fmul_rn(tg[4], (float)L.nx), gh = __fmul_rn(tg[5], (float)L.ny);
  const float aw = L.anchors[2 * a], ah = L.anchors[2 * a + 1];
  const float rw = gw / aw, rh = gh / ah;
  // A target row whose image index is outside [0, bs) (a last partial batch, per-rank slices that kept global indices) or that
  // holds a NaN would index p / dp / winner out of bounds; the torch formulation raises IndexError there.  Such rows are
  // rejected and counted: ly_loss_finish turns a non-zero count into a NaN loss (loud, and without a host sync).
  const int b = (int)tg[0];
  const bool padding = tg[0] == -1.f;                     // a row with image index exactly -1 is PADDING (fixed-shape target
                                                          // buffers of a captured step): ignored silently
  const bool sane = tg[0] >= 0.f && b < L.bs && gx == gx && gy == gy && gw == gw && gh == gh;
  if (!sane && !padding && k == 0 && a == 0) atomicAdd(L.acc + 3, 1.f);
  bool ok = sane && fmaxf(fmaxf(rw, 1.f / rw), fmaxf(rh, 1.f / rh)) < L.anchor_t;
  const float g = 0.5f;
  float ox = 0.f, oy = 0.f;
  if (k == 1) { ok = ok && (fmodf(gx, 1.f) < g && gx > 1.f); ox = g; }
  else if (k == 2) { ok = ok && (fmodf(gy, 1.f) < g && gy > 1.f); oy = g; }
  else if (k == 3) { const float ix = L.nx - gx; ok = ok && (fmodf(ix, 1.f) < g && ix > 1.f); ox = -g; }
  else if (k == 4) { const float iy = L.ny - gy; ok = ok && (fmodf(iy, 1.f) < g && iy > 1.f); oy = -g; }
  long cell = -1;
  if (ok) {
    const int gi_raw = (int)(gx - ox), gj_raw = (int)(gy - oy);           // .long(): truncation
    const int gi = gi_raw < 0 ? 0 : gi_raw > L.nx - 1 ? L.nx - 1 : gi_raw;
    const int gj = gj_raw < 0 ? 0 : gj_raw > L.ny - 1 ? L.ny - 1 : gj_raw;
    cell = (((long)b * L.na + a) * L.ny + gj) * L.nx + gi;
    const float tx = __fsub_rn(gx, (float)gi_raw), ty = __fsub_rn(gy, (float)gj_raw);   // tbox uses the UNclamped cell; no FMA contraction with gx = x*nx (bit-exact vs torch)
    if (L.tbox) { float* tb = L.tbox + idx * 4; tb[0] = tx; tb[1] = ty; tb[2] = gw; tb[3] = gh; }
    const float* pr = L.p + cell * L.no;
    const float s0 = ly_sigmoid(pr[0]), s1 = ly_sigmoid(pr[1]), s2 = ly_sigmoid(pr[2]), s3 = ly_sigmoid(pr[3]);
    const float pxv = s0 * 2.f - 0.5f, pyv = s1 * 2.f - 0.5f, pwv = (s2 * 2.f) * (s2 * 2.f) * aw, phv = (s3 * 2.f) * (s3 * 2.f) * ah;
    const D4 e = ly_eiou(d4v(pxv, 0), d4v(pyv, 1), d4v(pwv, 2), d4v(phv, 3), tx, ty, gw, gh);
    float* c = L.cand + idx * 5;
    c[0] = e.v;                                                            // (the reference's `iou` here is the EIoU value)
    // d(1 - eiou)/d raw = -d eiou/d box * d box/d raw
    c[1] = -e.d[0] * 2.f * s0 * (1.f - s0);
    c[2] = -e.d[1] * 2.f * s1 * (1.f - s1);
    c[3] = -e.d[2] * 8.f * s2 * s2 * (1.f - s2) * aw;
    c[4] = -e.d[3] * 8.f * s3 * s3 * (1.f - s3) * ah;
    atomicAdd(L.acc + 0, 1.f - e.v);
    atomicAdd(L.acc + 1, 1.f);
    atomicMax(L.winner + cell, (int)idx);
    const int nc = L.no - 5;
    if (nc > 1) {                                                          // class BCE of this matched row (utils/loss.py:168-173)
      const int cls = (int)tg[1];
      float sc = 0.f;
      for (int q = 0; q < nc; ++q) sc += ly_bce(pr[5 + q], q == cls ? L.cp : L.cn, L.cls_pw);
      atomicAdd(L.acc + 4, sc);
    }
  }
  L.cand_cell[idx] = cell;
}

__global__ __launch_bounds__(LY_THREADS) void ly_loss_apply_kernel(const LyLossLevel L, float box_gain, float cls_gain) {
  const long ncand = 5L * L.na * L.nt;
  const long idx = (long)blockIdx.x * LY_THREADS + threadIdx.x;
  if (idx >= ncand) return;
  const long cell = L.cand_cell[idx];
  if (cell < 0) return;
  const float* c = L.cand + idx * 5;
  const float scale = box_gain * (float)L.bs / L.acc[1];                   // d/d(1-eiou) of  box * mean(1-eiou) * bs
  float* d = L.dp + cell * L.no;
#pragma unroll
  for (int r = 0; r < 4; ++r) atomicAdd(d + r, scale * c[1 + r]);
  if (L.winner[cell] == (int)idx) L.tobj[cell] = fmaxf(c[0], 0.f);
  const int nc = L.no - 5;
  if (nc > 1) {                                                            // d/dx of  cls * mean over (rows, classes) of BCE * bs
    const long t = idx % L.nt;
    const int cls = (int)L.targets[t * 6 + 1];
    const float cs = cls_gain * (float)L.bs / (L.acc[1] * (float)nc);
    const float* pr = L.p + cell * L.no;
    for (int q = 0; q < nc; ++q) atomicAdd(d + 5 + q, cs * ly_bce_grad(pr[5 + q], q == cls ? L.cp : L.cn, L.cls_pw));
  }
}

__global__ __launch_bounds__(LY_THREADS) void ly_loss_obj_kernel(const LyLossLevel L, float obj_scale) {
  __shared__ float red[4];
  const long cells = (long)L.bs * L.na * L.ny * L.nx;
  float s = 0.f;
  for (long i = (long)blockIdx.x * LY_THREADS + threadIdx.x; i < cells; i += (long)gridDim.x * LY_THREADS) {
    const float x = L.p[i * L.no + 4], z = L.tobj[i];
    s += ly_bce(x, z, L.obj_pw);
    L.dp[i * L.no + 4] = ly_bce_grad(x, z, L.obj_pw) * obj_scale;          // obj_scale = obj * balance * bs / cells
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(L.acc + 2, red[0] + red[1] + red[2] + red[3]);
}

extern "C" int ly_loss_level(const float* p, float* dp, const float* anchors, const float* targets, int bs, int na, int ny, int nx, int no, long nt,
                             float anchor_t, float box_gain, float obj_gain, float balance, float* tobj, int* winner, long* cand_cell, float* cand,
                             float* acc, float* tbox, int match_only, float cls_gain, float cp, float cn, float cls_pw, float obj_pw, void* stream) {
  LY_CHECK(p && (dp || match_only) && anchors && tobj && winner && acc && (nt == 0 || (targets && cand_cell && cand)), "loss_level: null pointer");
  LY_CHECK(bs > 0 && na > 0 && ny > 0 && nx > 0 && no >= 5 && nt >= 0, "loss_level: bad sizes");
  const long cells = (long)bs * na * ny * nx;
  LY_CHECK(5L * na * nt < (1L << 31) && cells < (1L << 40), "loss_level: too many candidates");
  LyLossLevel L{p, dp, anchors, targets, bs, na, ny, nx, no, nt, anchor_t, tobj, winner, cand_cell, cand, acc, tbox, cp, cn, cls_pw, obj_pw};
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long ncand = 5L * na * nt;
  if (ncand > 0) {
    const unsigned blocks = (unsigned)((ncand + LY_THREADS - 1) / LY_THREADS);
    hipLaunchKernelGGL(ly_loss_match_kernel, dim3(blocks), dim3(LY_THREADS), 0, st, L);
    if (!match_only) hipLaunchKernelGGL(ly_loss_apply_kernel, dim3(blocks), dim3(LY_THREADS), 0, st, L, box_gain, cls_gain);
  }
  if (match_only) {            // target assignment only (ComputeLoss.build_targets): cand_cell / tbox are the result
    LY_LAUNCH_CHECK();
    return 0;
  }
  long ob = (cells + LY_THREADS * 8L - 1) / (LY_THREADS * 8L);
  ob = ob < 1 ? 1 : ob > 2048 ? 2048 : ob;
  hipLaunchKernelGGL(ly_loss_obj_kernel, dim3((unsigned)ob), dim3(LY_THREADS), 0, st, L, obj_gain * balance * (float)bs / (float)cells);
  LY_LAUNCH_CHECK();
  return 0;
}

// out[0] = total loss, out[1..3] = (lbox, lobj, lcls) as the reference returns them; acc = [nl][8] level accumulators
__global__ void ly_loss_finish_kernel(const float* __restrict__ acc, int nl, const float* __restrict__ cells, const float* __restrict__ balance,
                                      float box_gain, float obj_gain, float cls_gain, int nc, float bs, float* __restrict__ out) {
  if (threadIdx.x != 0) return;
  float lbox = 0.f, lobj = 0.f, lcls = 0.f, bad = 0.f;
  for (int i = 0; i < nl; ++i) {
    const float* a = acc + 8 * i;
    bad += a[3];
    if (a[1] > 0.f) {
      lbox += a[0] / a[1];
      if (nc > 1) lcls += a[4] / (a[1] * (float)nc);
    }
    lobj += a[2] / cells[i] * balance[i];
  }
  lbox *= box_gain;
  lobj *= obj_gain;
  lcls *= cls_gain;
  out[0] = bad > 0.f ? __builtin_nanf("") : (lbox + lobj + lcls) * bs;      // rejected target rows (image index outside the batch / NaN): fail loudly
  out[1] = lbox;
  out[2] = lobj;
  out[3] = lcls;
}

extern "C" int ly_loss_finish(const float* acc, int nl, const float* cells, const float* balance, float box_gain, float obj_gain, float cls_gain,
                              int nc, int bs, float* out, void* stream) {
  LY_CHECK(acc && cells && balance && out && nl > 0, "loss_finish: bad arguments");
  hipLaunchKernelGGL(ly_loss_finish_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), acc, nl, cells, balance, box_gain, obj_gain,
                     cls_gain, nc, (float)bs, out);
  LY_LAUNCH_CHECK();
  return 0;
}
