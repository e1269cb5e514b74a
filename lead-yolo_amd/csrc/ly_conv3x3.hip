// 3x3 / stride 1 / pad 1 convolution as an implicit GEMM + folded BN + activation, fp32, gfx950.
//
//   out[m, n] = act( scale[n] * sum_{tap, c} X[pix(m) + tap][c] * W[n][c][tap] + shift[n] )
//
// Replaces CA_Bottleneck.cv2 = Conv(c_, c_, 3, 1) (reference models/common.py:1617,1890-1910): no
// im2col buffer exists anywhere; the block stages a halo run of the NHWC input (BP + 2W + 2
// consecutive pixels x 32 input channels) in LDS and every MFMA B-operand is a 16-byte read at
// (pixel + tap offset), masked at the image borders.  K order = (tap, cin); weights bf16x3 frag-packed
// from pack.conv_taps_matrix(w, 32).
#include "ly_tile.cuh"
#include "ly_params.h"

#define LY_CC 32
#define LY_RSH (2 * LY_CC + 16)   // bytes per halo row, per plane

template <int NT, int MT, int WC>
__global__ __launch_bounds__(LY_THREADS) void ly_conv3x3_kernel(const LyConv3Params P, const int gy, const int nblocks) {
  constexpr int WP = 4 / WC;
  constexpr int BP = 16 * NT * WP;
  extern __shared__ f32x4 ly_smem4[];
  char* hs_hi = reinterpret_cast<char*>(ly_smem4);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  const int wc = wave % WC, wp_ = wave / WC;
  const int lid = ly_xcd_remap(blockIdx.x, nblocks);
  const int by = lid % gy;
  const long p0 = (long)(lid / gy) * BP;
  const int W = P.W, H = P.H;
  const int BPH = BP + 2 * W + 2;
  char* hs_lo = hs_hi + BPH * LY_RSH;
  const f32x4 zero = ly_zero4();
  const bf16x8 z8 = __builtin_bit_cast(bf16x8, make_uint4(0u, 0u, 0u, 0u));
  const int C32 = (P.Cin + 31) >> 5;       // k-steps per tap
  const int S = 9 * C32;
  const int T = (P.N + 15) >> 4;
  const int pixgrp = wp_ * (16 * NT);

  uint32_t tmask[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    long gp = p0 + pixgrp + 16 * n + li;
    int h_, w_;
    ly_pix_hw(gp, H, W, h_, w_);
    tmask[n] = ly_tapmask(h_, w_, H, W, gp < P.M);
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[t][n] = zero;
  int tile[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    int tt = (by * WC + wc) * MT + t;
    tile[t] = tt < T ? tt : T - 1;
  }
  const uint4* wpk = reinterpret_cast<const uint4*>(P.wp);

  for (int c0 = 0; c0 < P.Cin; c0 += LY_CC) {
    __syncthreads();
    ly_stage_f4<8>(BPH * (LY_CC / 4), tid, P.x,
        [&](int idx) -> const float* {
          const int hp = idx / (LY_CC / 4), c4 = idx - hp * (LY_CC / 4);
          const long gp = p0 - W - 1 + hp;
          const int c = c0 + 4 * c4;
          return (gp >= 0 && gp < P.M && c < P.Cin) ? P.x + gp * P.ldx + c : nullptr;
        },
        [&](int idx, f32x4 v) {
          const int hp = idx / (LY_CC / 4), c4 = idx - hp * (LY_CC / 4);
          ly_lds_put4(hs_hi, hs_lo, hp * LY_RSH, 4 * c4, v);
        });
    __syncthreads();
    const int sc0 = c0 >> 5;
#pragma unroll 3
    for (int tap = 0; tap < 9; ++tap) {
      const int ty = tap / 3, tx = tap - 3 * ty;
      const int off = (ty * W + tx) * LY_RSH;
      bf16x8 xh[NT], xl[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int rb = (pixgrp + 16 * n + li) * LY_RSH + off;
        const bool ok = (tmask[n] >> tap) & 1u;
        const bf16x8 a = ly_lds_frag(hs_hi, rb, 0, lq), b = ly_lds_frag(hs_lo, rb, 0, lq);
        xh[n] = ok ? a : z8;
        xl[n] = ok ? b : z8;
      }
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const LyWFrag wf = ly_wfrag(wpk, (long)tile[t] * S + tap * C32 + sc0, lane);
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[t][n] = ly_mfma3(wf.hi, wf.lo, xh[n], xl[n], acc[t][n]);
      }
    }
  }

#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int tt = (by * WC + wc) * MT + t;
    const int c = 16 * tt + 4 * lq;
    if (tt >= T || c >= P.N) continue;
    float sc[4], sh[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = c + r < P.N;
      sc[r] = (ok && P.e_scale) ? P.e_scale[c + r] : 1.f;
      sh[r] = (ok && P.e_shift) ? P.e_shift[c + r] : 0.f;
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const long gp = p0 + pixgrp + 16 * n + li;
      if (gp >= P.M) continue;
      f32x4 u;
#pragma unroll
      for (int r = 0; r < 4; ++r) u[r] = acc[t][n][r] * sc[r] + sh[r];
      const f32x4 v = ly_act4(u, P.act);
      float* o = P.out + gp * P.ldo + c;
      if ((P.ldo & 3) == 0 && c + 3 < P.N) {
        ly_stg4(o, v);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (c + r < P.N) o[r] = v[r];
      }
    }
  }
}

template <int NT, int MT, int WC>
static int launch_conv3(const LyConv3Params& P, hipStream_t st) {
  constexpr int BP = 16 * NT * (4 / WC);
  constexpr int BN = 16 * MT * WC;
  long gx = (P.M + BP - 1) / BP;
  int gy = (P.N + BN - 1) / BN;
  long nb = gx * gy;
  LY_CHECK(nb < (1L << 31), "conv3x3: grid too large");
  size_t lds = 2 * (size_t)(BP + 2 * P.W + 2) * LY_RSH;
  LY_CHECK(lds <= 160 * 1024, "conv3x3: halo tile needs %zu B of LDS (W=%d)", lds, P.W);
  auto k = ly_conv3x3_kernel<NT, MT, WC>;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    LY_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    configured = true;
  }
  hipLaunchKernelGGL(k, dim3((unsigned)nb), dim3(LY_THREADS), lds, st, P, gy, (int)nb);
  LY_LAUNCH_CHECK();
  return 0;
}

extern "C" int ly_conv3x3_fwd(const LyConv3Params* p, void* stream) {
  LY_CHECK(p, "conv3x3: null params");
  const LyConv3Params& P = *p;
  LY_CHECK(P.x && P.wp && P.out, "conv3x3: null pointer");
  LY_CHECK(P.M > 0 && P.H > 0 && P.W > 0 && P.Cin > 0 && P.N > 0, "conv3x3: bad sizes");
  LY_CHECK((P.Cin & 3) == 0 && (P.ldx & 3) == 0, "conv3x3: Cin=%d / ldx=%d must be multiples of 4", P.Cin, P.ldx);
  LY_CHECK(P.M % ((long)P.H * P.W) == 0, "conv3x3: M is not a whole number of images");
  LY_CHECK(P.M < (1L << 24), "conv3x3: M=%ld pixels exceeds the 2^24 limit of the fast index path", P.M);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool big = P.M >= 128L * 384;
  if (P.N > 128) return big ? launch_conv3<8, 4, 4>(P, st) : launch_conv3<4, 4, 4>(P, st);   // x 256 ch
  if (P.N > 64) return big ? launch_conv3<8, 2, 4>(P, st) : launch_conv3<4, 2, 4>(P, st);    // x 128 ch
  if (P.N > 32) return big ? launch_conv3<8, 1, 4>(P, st) : launch_conv3<4, 1, 4>(P, st);    // x 64 ch
  return launch_conv3<2, 2, 1>(P, st);                                                        // 128 px x 32 ch
}
