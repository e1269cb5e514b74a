// Fused FasterNet MLPBlock forward (eval / folded-BN form), fp32, gfx950.
//
//   y = x + W2 . relu( s * (W1 . [ pconv3x3(x[:, :C/4]) | x[:, C/4:] ]) + b )
//
// Replaces Partial_conv3.forward_split_cat + MLPBlock.forward (reference models/common.py:1432-1437,
// 1478-1482): the split/cat copies, the 2C-wide hidden tensor and the BN/ReLU passes never touch HBM.
// HBM traffic = x once + y once (+ a one-pixel-row halo of the first C/4 channels, + the residual
// re-read which is an L2 hit on lines this block just fetched).
//
// Block = 256 threads (4 waves) owns BP = 64*NT consecutive pixels of the flattened N*H*W index
// (NHWC rows, so its input tile is one contiguous span of memory).  Each wave owns 16*NT pixels and
// carries them through all three contractions:
//   1. partial 3x3 conv as an implicit GEMM over K = 9 * ceil4(C/4), operands gathered from a halo
//      copy (ps) of the first C/4 channels with per-tap border masks; result overwrites channels
//      [0, C/4) of the wave's own rows in the LDS tile xs (the "concat" is a no-op).
//   2. hidden = relu(bn(W1 . xs_row)): HT hidden tiles at a time.
//   3. out += W2[:, hidden tile] . hidden   -- the MFMA D tile of step 2 IS the B operand (see
//      ly_common.cuh), so the hidden activations live only in registers.
// Weights are frag-packed and read straight from global/L2 (they are shared by every block).
#include "ly_common.cuh"

template <int C, int NT, int HT>
__global__ __launch_bounds__(LY_THREADS) void ly_mlpblock_fwd_kernel(
    const float* __restrict__ x, float* __restrict__ y, long M, int H, int W,
    const f32x4* __restrict__ wp, const f32x4* __restrict__ w1, const f32x4* __restrict__ w2,
    const float* __restrict__ bn_scale, const float* __restrict__ bn_shift) {
  constexpr int CQ = C / 4;
  constexpr int CQP = (CQ + 3) / 4 * 4;
  constexpr int G = CQP / 4;
  constexpr int SP = (9 * G + 3) / 4;
  constexpr int PT = (CQ + 15) / 16;
  constexpr int C16 = (C + 15) / 16;
  constexpr int XV = C16 * 4;
  constexpr int LDX = C16 * 16 + 4;
  constexpr int LDP = CQP + 4;
  constexpr int HTILES = 2 * C / 16;
  constexpr int BP = 64 * NT;
  static_assert(HTILES % HT == 0, "hidden tiles must split evenly into chunks");
  static_assert(C % 8 == 0, "C must be a multiple of 8");

  extern __shared__ f32x4 ly_smem4[];
  float* xs = reinterpret_cast<float*>(ly_smem4);
  float* ps = xs + BP * LDX;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  const long p0 = (long)blockIdx.x * BP;
  const int BPH = BP + 2 * W + 2;
  const f32x4 zero = ly_zero4();

  for (int idx = tid; idx < BP * XV; idx += LY_THREADS) {
    int pix = idx / XV, c4 = idx - pix * XV;
    long gp = p0 + pix;
    f32x4 v = zero;
    if (gp < M && c4 * 4 < C) v = ly_ldg4(x + gp * C + c4 * 4);
    *reinterpret_cast<f32x4*>(xs + pix * LDX + c4 * 4) = v;
  }
  for (int idx = tid; idx < BPH * G; idx += LY_THREADS) {
    int hp = idx / G, c4 = idx - hp * G;
    long gp = p0 - W - 1 + hp;
    f32x4 v = zero;
    if (gp >= 0 && gp < M) v = ly_ldg4(x + gp * C + c4 * 4);
    *reinterpret_cast<f32x4*>(ps + hp * LDP + c4 * 4) = v;
  }
  __syncthreads();

  const int pixbase = wave * (16 * NT);

  // ---- 1. partial 3x3 conv -------------------------------------------------------------------
  {
    uint32_t tmask[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      long gp = p0 + pixbase + 16 * n + li;
      int w_ = (int)(gp % W);
      int h_ = (int)((gp / W) % H);
      tmask[n] = ly_tapmask(h_, w_, H, W, gp < M);
    }
    f32x4 accp[PT][NT];
#pragma unroll
    for (int t = 0; t < PT; ++t)
#pragma unroll
      for (int n = 0; n < NT; ++n) accp[t][n] = zero;

#pragma unroll
    for (int s = 0; s < SP; ++s) {
      const int g = 4 * s + lq;
      const bool gv = g < 9 * G;
      const int tap = gv ? g / G : 0;
      const int cq4 = gv ? g - tap * G : 0;
      const int ty = tap / 3, tx = tap - 3 * ty;
      const int off = (ty * W + tx) * LDP + cq4 * 4;
      f32x4 xf[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        f32x4 v = *reinterpret_cast<const f32x4*>(ps + (pixbase + 16 * n + li) * LDP + off);
        bool ok = gv && ((tmask[n] >> tap) & 1u);
        xf[n] = ok ? v : zero;
      }
#pragma unroll
      for (int t = 0; t < PT; ++t) {
        f32x4 wf = wp[(t * SP + s) * 64 + lane];
#pragma unroll
        for (int n = 0; n < NT; ++n) accp[t][n] = ly_mfma4(wf, xf[n], accp[t][n]);
      }
    }
#pragma unroll
    for (int t = 0; t < PT; ++t)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int c = 16 * t + 4 * lq + r;
          if (c < CQ) xs[(pixbase + 16 * n + li) * LDX + c] = accp[t][n][r];
        }
  }

  // ---- 2 + 3. expand -> BN -> ReLU -> project, hidden kept in registers ---------------------------
  f32x4 acco[C16][NT];
#pragma unroll
  for (int t = 0; t < C16; ++t)
#pragma unroll
    for (int n = 0; n < NT; ++n) acco[t][n] = zero;

#pragma unroll 1
  for (int hc = 0; hc < HTILES / HT; ++hc) {
    f32x4 acch[HT][NT];
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
      for (int n = 0; n < NT; ++n) acch[t][n] = zero;
#pragma unroll
    for (int s = 0; s < C16; ++s) {
      f32x4 xf[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n)
        xf[n] = *reinterpret_cast<const f32x4*>(xs + (pixbase + 16 * n + li) * LDX + 16 * s + 4 * lq);
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        f32x4 wf = w1[((hc * HT + t) * C16 + s) * 64 + lane];
#pragma unroll
        for (int n = 0; n < NT; ++n) acch[t][n] = ly_mfma4(wf, xf[n], acch[t][n]);
      }
    }
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      const int ch = (hc * HT + t) * 16 + 4 * lq;
      const f32x4 sc = ly_ldg4(bn_scale + ch), sh = ly_ldg4(bn_shift + ch);
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) acch[t][n][r] = fmaxf(acch[t][n][r] * sc[r] + sh[r], 0.f);
    }
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
      for (int ct = 0; ct < C16; ++ct) {
        f32x4 wf = w2[(ct * HTILES + hc * HT + t) * 64 + lane];
#pragma unroll
        for (int n = 0; n < NT; ++n) acco[ct][n] = ly_mfma4(wf, acch[t][n], acco[ct][n]);
      }
  }

  // ---- epilogue: residual + store ------------------------------------------------------------
#pragma unroll
  for (int ct = 0; ct < C16; ++ct)
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int c = 16 * ct + 4 * lq;
      const long gp = p0 + pixbase + 16 * n + li;
      if (c < C && gp < M) {
        f32x4 r = ly_ldg4(x + gp * C + c);
        ly_stg4(y + gp * C + c, acco[ct][n] + r);
      }
    }
}

template <int C, int NT, int HT>
static int launch_mlp(const float* x, float* y, long M, int H, int W, const float* wp, const float* w1,
                      const float* w2, const float* s, const float* b, hipStream_t st) {
  constexpr int CQ = C / 4, CQP = (CQ + 3) / 4 * 4, C16 = (C + 15) / 16;
  constexpr int LDX = C16 * 16 + 4, LDP = CQP + 4, BP = 64 * NT;
  size_t lds = sizeof(float) * ((size_t)BP * LDX + (size_t)(BP + 2 * W + 2) * LDP);
  LY_CHECK(lds <= 160 * 1024, "mlpblock: tile needs %zu B of LDS (C=%d W=%d)", lds, C, W);
  auto k = ly_mlpblock_fwd_kernel<C, NT, HT>;
  static size_t configured = 0;
  if (lds > configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    LY_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    configured = 160 * 1024;
  }
  long blocks = (M + BP - 1) / BP;
  hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(LY_THREADS), lds, st, x, y, M, H, W,
                     reinterpret_cast<const f32x4*>(wp), reinterpret_cast<const f32x4*>(w1),
                     reinterpret_cast<const f32x4*>(w2), s, b);
  LY_LAUNCH_CHECK();
  return 0;
}

// pick pixel tiles per wave so that the grid still covers the chip (256 CUs) where M is small
template <int C, int HT, int NTMAX>
static int dispatch_nt(const float* x, float* y, long M, int H, int W, const float* wp, const float* w1,
                       const float* w2, const float* s, const float* b, hipStream_t st) {
  if (NTMAX >= 4 && M >= 4L * 64 * 4 * 256) return launch_mlp<C, (NTMAX >= 4 ? 4 : NTMAX), HT>(x, y, M, H, W, wp, w1, w2, s, b, st);
  if (NTMAX >= 2 && M >= 2L * 64 * 2 * 256) return launch_mlp<C, (NTMAX >= 2 ? 2 : NTMAX), HT>(x, y, M, H, W, wp, w1, w2, s, b, st);
  return launch_mlp<C, 1, HT>(x, y, M, H, W, wp, w1, w2, s, b, st);
}

extern "C" int ly_mlpblock_fwd(const float* x, float* y, int n_img, int H, int W, int C, const float* wp,
                               const float* w1, const float* w2, const float* bn_scale, const float* bn_shift,
                               void* stream) {
  LY_CHECK(x && y && wp && w1 && w2 && bn_scale && bn_shift, "mlpblock: null pointer");
  LY_CHECK(x != y, "mlpblock: in-place call is not supported (neighbouring tiles read halo rows)");
  LY_CHECK(n_img > 0 && H > 0 && W > 0, "mlpblock: bad shape %d x %d x %d", n_img, H, W);
  long M = (long)n_img * H * W;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  switch (C) {
    case 16:  return dispatch_nt<16, 2, 4>(x, y, M, H, W, wp, w1, w2, bn_scale, bn_shift, st);
    case 24:  return dispatch_nt<24, 3, 4>(x, y, M, H, W, wp, w1, w2, bn_scale, bn_shift, st);
    case 40:  return dispatch_nt<40, 5, 4>(x, y, M, H, W, wp, w1, w2, bn_scale, bn_shift, st);
    case 80:  return dispatch_nt<80, 5, 2>(x, y, M, H, W, wp, w1, w2, bn_scale, bn_shift, st);
    case 160: return dispatch_nt<160, 5, 2>(x, y, M, H, W, wp, w1, w2, bn_scale, bn_shift, st);
    case 320: return dispatch_nt<320, 4, 1>(x, y, M, H, W, wp, w1, w2, bn_scale, bn_shift, st);
    default:
      ly_set_error("mlpblock: unsupported channel count C=%d (built for 16/24/40/80/160/320)", C);
      return -1;
  }
}

// sizes (in floats) of the three frag-packed weight buffers for a given C
extern "C" int ly_mlpblock_pack_sizes(int C, long* n_wp, long* n_w1, long* n_w2) {
  int CQ = C / 4, CQP = (CQ + 3) / 4 * 4, G = CQP / 4, SP = (9 * G + 3) / 4, PT = (CQ + 15) / 16;
  int C16 = (C + 15) / 16, HT = 2 * C / 16;
  *n_wp = (long)PT * SP * 256;
  *n_w1 = (long)HT * C16 * 256;
  *n_w2 = (long)C16 * HT * 256;
  return 0;
}
