// Pointwise-convolution GEMM with fused gather/prologue and epilogue, fp32, gfx950.
//
//   out[m, n] = act( rowscale[m] * scale[n] * sum_k A'[m, k] * W[n, k] + shift[n] )
//
// m runs over the flattened NHWC pixels, so A' is the activation matrix itself (no im2col).  A' is
// assembled while the tile is staged into LDS, which is where the reference's separate passes go:
//   * two row sources (a0 | a1)            -> torch.cat of two feature maps never materialised
//   * a0 at half resolution (up0)          -> nn.Upsample(2, 'nearest') folded into the load
//   * gate: a0 * a_w[n,w,:] * a_h[n,h,:]   -> CoordAtt's `identity * a_w * a_h` (models/common.py:1608)
//   * affine+relu+ca: relu(x*s+b)*ca[n,:]  -> RFCBAMConv k=1 generate/BN/ReLU and SE scaling
//                                             (models/rfa.py:101-106,124)
//   * patch gather (k x k stride k)        -> PatchEmbed/PatchMerging_FasterNet (models/common.py:1528-1561)
// and the epilogue covers BN (folded scale/shift), conv bias, the per-pixel receptive-field weight
// (rowscale) and ReLU / SiLU.  `out` may point into a wider buffer (ldo, pre-offset pointer) so a
// producer can write straight into its slot of a later concat.
//
// Block = 4 waves; WC waves split the output channels (distinct weight fragments per wave: no
// redundant weight traffic), 4/WC waves split the pixels.  Wave tile = NT pixel tiles x MT channel
// tiles of 16x16 (f32 MFMA 16x16x4, see ly_common.cuh).  K is consumed in LDS chunks of 64.
#include "ly_common.cuh"
#include "ly_params.h"

#define LY_BK 64
#define LY_LDX (LY_BK + 4)

template <int NT, int MT, int WC>
__global__ __launch_bounds__(LY_THREADS) void ly_gemm_kernel(const LyGemmParams P, const int gy, const int nblocks) {
  constexpr int WP = 4 / WC;
  constexpr int BP = 16 * NT * WP;
  __shared__ f32x4 xs4[BP * LY_LDX / 4];
  __shared__ int r_n[BP], r_h[BP], r_w[BP];
  __shared__ long r_row0[BP];
  float* xs = reinterpret_cast<float*>(xs4);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  const int wc = wave % WC, wp_ = wave / WC;
  const int lid = ly_xcd_remap(blockIdx.x, nblocks);
  const int by = lid % gy;
  const long p0 = (long)(lid / gy) * BP;
  const int HW = P.H * P.W;
  const f32x4 zero = ly_zero4();
  const int S = (P.K + 15) >> 4;
  const int T = (P.N + 15) >> 4;

  for (int pix = tid; pix < BP; pix += LY_THREADS) {
    long gp = p0 + pix;
    int n = -1, h = 0, w = 0;
    long row0 = 0;
    if (gp < P.M) {
      n = (int)(gp / HW);
      int rem = (int)(gp - (long)n * HW);
      h = rem / P.W;
      w = rem - h * P.W;
      if (P.gather == LY_GATHER_UP2)
        row0 = ((long)n * (P.H >> 1) + (h >> 1)) * (P.W >> 1) + (w >> 1);
      else if (P.gather == LY_GATHER_PATCH)
        row0 = (((long)n * P.Hin + (long)h * P.ks) * P.Win + (long)w * P.ks);       // first input pixel of the patch
      else if (P.gather == LY_GATHER_PATCH_NCHW)
        row0 = ((long)n * P.Cin * P.Hin + (long)h * P.ks) * P.Win + (long)w * P.ks; // element offset of (n, c=0, ks*h, ks*w)
      else
        row0 = gp;
    }
    r_n[pix] = n; r_h[pix] = h; r_w[pix] = w; r_row0[pix] = row0;
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
  const f32x4* wpk = reinterpret_cast<const f32x4*>(P.wp);
  const int pixgrp = wp_ * (16 * NT);

  for (int kc = 0; kc < P.K; kc += LY_BK) {
    __syncthreads();
    // ---- stage A'[BP x 64] ----------------------------------------------------------------------
    for (int idx = tid; idx < BP * (LY_BK / 4); idx += LY_THREADS) {
      const int pix = idx / (LY_BK / 4), k4 = idx - pix * (LY_BK / 4);
      const int kk = kc + 4 * k4;
      const int n = r_n[pix];
      f32x4 v = zero;
      if (n >= 0 && kk < P.K) {
        const long row0 = r_row0[pix];
        if (P.gather == LY_GATHER_PATCH) {
          const int seg = kk / P.pk, within = kk - seg * P.pk;
          v = ly_ldg4(P.a0 + (row0 + (long)seg * P.Win) * P.lda0 + within);
        } else if (P.gather == LY_GATHER_PATCH_NCHW) {
          const int c = kk / (P.ks * P.ks), ky = (kk - c * P.ks * P.ks) / P.ks;   // ks == 4: one float4 = one (c, ky) row
          v = ly_ldg4(P.a0 + row0 + ((long)c * P.Hin + ky) * P.Win);
        } else if (kk < P.k0) {
          v = ly_ldg4(P.a0 + row0 * P.lda0 + kk);
          if (P.pro == LY_PRO_GATE) {
            const f32x4 gw = ly_ldg4(P.g_w + ((long)n * P.W + r_w[pix]) * P.k0 + kk);
            const f32x4 gh = ly_ldg4(P.g_h + ((long)n * P.H + r_h[pix]) * P.k0 + kk);
            v = v * gw * gh;
            if (P.res) v = v + ly_ldg4(P.res + (p0 + pix) * P.ldres + kk);
          }
        } else {
          v = ly_ldg4(P.a1 + (p0 + pix) * P.lda1 + (kk - P.k0));
        }
        if (P.pro == LY_PRO_AFFINE_RELU_CA) {
          const f32x4 s = ly_ldg4(P.p_scale + kk), b = ly_ldg4(P.p_shift + kk);
          const f32x4 ca = ly_ldg4(P.p_ca + (long)n * P.K + kk);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r] * s[r] + b[r], 0.f) * ca[r];
        }
      }
      *reinterpret_cast<f32x4*>(xs + pix * LY_LDX + 4 * k4) = v;
    }
    __syncthreads();
    // ---- contract -------------------------------------------------------------------------------
    const int s0 = kc >> 4;
    const int ns = (S - s0) < (LY_BK / 16) ? (S - s0) : (LY_BK / 16);
    for (int s = 0; s < ns; ++s) {
      f32x4 xf[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n)
        xf[n] = *reinterpret_cast<const f32x4*>(xs + (pixgrp + 16 * n + li) * LY_LDX + 16 * s + 4 * lq);
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const f32x4 wf = wpk[((long)tile[t] * S + s0 + s) * 64 + lane];
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[t][n] = ly_mfma4(wf, xf[n], acc[t][n]);
      }
    }
  }

  // ---- epilogue ---------------------------------------------------------------------------------
  const bool vec_ok = (P.ldo & 3) == 0;
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
      const float rs = P.rowscale ? P.rowscale[gp] : 1.f;
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float u = acc[t][n][r] * rs * sc[r] + sh[r];
        v[r] = P.act == LY_ACT_RELU ? ly_relu(u) : (P.act == LY_ACT_SILU ? ly_silu(u) : u);
      }
      float* o = P.out + gp * P.ldo + c;
      if (vec_ok && c + 3 < P.N) {
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
static int launch_gemm(const LyGemmParams& P, hipStream_t st) {
  constexpr int BP = 16 * NT * (4 / WC);
  constexpr int BN = 16 * MT * WC;
  long gx = (P.M + BP - 1) / BP;
  int gy = (P.N + BN - 1) / BN;
  long nb = gx * gy;
  LY_CHECK(nb < (1L << 31), "gemm: grid too large");
  hipLaunchKernelGGL((ly_gemm_kernel<NT, MT, WC>), dim3((unsigned)nb), dim3(LY_THREADS), 0, st, P, gy, (int)nb);
  LY_LAUNCH_CHECK();
  return 0;
}

extern "C" int ly_gemm_fwd(const LyGemmParams* p, void* stream) {
  LY_CHECK(p, "gemm: null params");
  const LyGemmParams& P = *p;
  LY_CHECK(P.a0 && P.wp && P.out, "gemm: null a0/wp/out");
  LY_CHECK(P.M > 0 && P.K > 0 && P.N > 0 && P.H > 0 && P.W > 0, "gemm: bad sizes M=%ld K=%d N=%d", P.M, P.K, P.N);
  LY_CHECK((P.K & 3) == 0, "gemm: K=%d must be a multiple of 4", P.K);
  if (P.gather == LY_GATHER_ROWS || P.gather == LY_GATHER_UP2) {
    LY_CHECK((P.lda0 & 3) == 0 && (P.k0 & 3) == 0 && P.k0 <= P.K && P.k0 > 0, "gemm: lda0=%d k0=%d must be multiples of 4", P.lda0, P.k0);
    LY_CHECK(P.k0 == P.K || (P.a1 && (P.lda1 & 3) == 0), "gemm: second source missing or misaligned");
    if (P.gather == LY_GATHER_UP2) LY_CHECK((P.H & 1) == 0 && (P.W & 1) == 0, "gemm: upsample source needs even H, W");
  } else if (P.gather == LY_GATHER_PATCH) {
    LY_CHECK(P.ks > 0 && P.pk == P.ks * P.lda0 && (P.lda0 & 3) == 0 && P.K == P.ks * P.pk, "gemm: patch gather misconfigured");
    LY_CHECK(P.Hin >= P.H * P.ks && P.Win >= P.W * P.ks, "gemm: patch gather input too small");
  } else if (P.gather == LY_GATHER_PATCH_NCHW) {
    LY_CHECK(P.ks == 4 && (P.Win & 3) == 0 && P.K == P.Cin * 16, "gemm: NCHW patch gather needs ks=4, Win%%4==0");
    LY_CHECK(P.Hin >= P.H * P.ks && P.Win >= P.W * P.ks, "gemm: patch gather input too small");
  } else {
    LY_CHECK(false, "gemm: unknown gather mode %d", P.gather);
  }
  if (P.pro == LY_PRO_GATE) LY_CHECK(P.g_h && P.g_w, "gemm: gate prologue needs g_h/g_w");
  if (P.pro == LY_PRO_AFFINE_RELU_CA) LY_CHECK(P.p_scale && P.p_shift && P.p_ca, "gemm: affine prologue needs scale/shift/ca");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long M = P.M;
  if (P.N > 128) {
    if (M >= 64L * 512) return launch_gemm<4, 4, 4>(P, st);   // 64 px x 256 ch
    return launch_gemm<2, 4, 4>(P, st);                        // 32 px x 256 ch
  }
  if (P.N > 64) {
    if (M >= 64L * 512) return launch_gemm<4, 2, 4>(P, st);   // 64 px x 128 ch
    return launch_gemm<2, 2, 4>(P, st);                        // 32 px x 128 ch
  }
  if (P.N > 32) {
    if (M >= 64L * 512) return launch_gemm<4, 2, 2>(P, st);   // 128 px x 64 ch
    return launch_gemm<2, 2, 2>(P, st);                        // 64 px x 64 ch
  }
  if (M >= 128L * 512) return launch_gemm<2, 2, 1>(P, st);    // 128 px x 32 ch
  return launch_gemm<1, 2, 1>(P, st);                          // 64 px x 32 ch
}
