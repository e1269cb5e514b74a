// Detect level in ONE launch (eval): the head's 1x1 convolution to na*no channels + the decode of models/yolo.py:95-120   (gfx950 only)
//   p[n, a, h, w, o] = (W x[n, h, w, :] + b)[a*no + o]                              (the raw map the reference returns beside z)
//   z[n, zoff + (a*H + h)*W + w, o] = xy: (2 sig - 0.5 + grid) * stride, wh: (2 sig)^2 * anchor * stride, rest: sig
// The generic route was two launches per level — a GEMM to an [M, 20]-column buffer in the storage type (N = 18 is no multiple of 4: its
// slow epilogue) and ly_detect_tail re-reading it — ~17 us per level in the serving forward for a few MB.  Here a wave owns 16 pixels: its B
// operand is read straight from the feature map (lane (pixel, g) loads the 8 channels 32 s + 8 g .. of its pixel per k-step: 64 contiguous bytes
// per pixel over the four g), the two weight tiles (32 rows, natural k order: pack.frag_pack_nat) sit in LDS, and sigmoid / grid / anchor
// arithmetic runs on the fp32 accumulators — the raw map is no longer rounded to bf16 on its way to p.
#include "ly_common.hpp"
#include "ly_tile.hpp"

#define LY_DET_LD 36            // floats per pixel row of a wave's output tile in LDS (32 channels + pad, rows 16-byte aligned)

template <typename T, int S, bool NAT>
__global__ __launch_bounds__(LY_THREADS) void ly_detect_level_kernel(const T* __restrict__ x, const int ldx, const long M, const int H, const int W,
                                                                      const uint4* __restrict__ wp, const float* __restrict__ bias, const int na,
                                                                      const int no, const float* __restrict__ anchors, const float stride,
                                                                      float* __restrict__ p, float* __restrict__ z, const long zrows, const long zoff,
                                                                      const int ntiles, const int tpw) {
  constexpr int PL = LyT<T>::PL;
  extern __shared__ uint4 ly_det_w[];                       // [2 tiles][S][PL][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * S * PL * 64; i += LY_THREADS) ly_det_w[i] = wp[i];
  __syncthreads();
  const int px = lane & 15, g = lane >> 4;
  const int co = na * no, HW = H * W;
  const float invHW = 1.f / (float)HW, invW = 1.f / (float)W;
  float cb[2][4];                                            // bias of the lane's channels c = 16 t + 4 g + r
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = 16 * t + 4 * g + r;
      cb[t][r] = c < co ? bias[c] : 0.f;
    }
  float* const ot = reinterpret_cast<float*>(ly_det_w + 2 * S * PL * 64) + wave * (16 * LY_DET_LD);
  const int tile0 = (blockIdx.x * (LY_THREADS / 64) + wave) * tpw;
  for (int i = 0; i < tpw; ++i) {
    const int tile = tile0 + i;
    if (tile >= ntiles) break;
    const long m = (long)tile * 16 + px;
    const bool live = m < M;
    // NAT: the lane's 8 channels of a k-step are consecutive (32 s + 8 g ..: one 16-byte load in bf16); else the k order of pack.frag_pack3
    // (32 s + 4 g + 0..3 and 32 s + 16 + 4 g + 0..3: the training step's packed weights are used as they are)
    const T* row = x + (live ? m : M - 1) * ldx + (NAT ? 8 : 4) * g;
    constexpr int SECOND = NAT ? 4 : 16;
    bf16x8 xh[S], xl[S];
    if constexpr (PL == 1) {
      ly_u32x2 ra[S], rb[S];
#pragma unroll
      for (int s = 0; s < S; ++s) {
        ra[s] = *reinterpret_cast<const ly_u32x2*>(row + 32 * s);
        rb[s] = *reinterpret_cast<const ly_u32x2*>(row + 32 * s + SECOND);
      }
#pragma unroll
      for (int s = 0; s < S; ++s) {
        xh[s] = ly_cat8(__builtin_bit_cast(bf16x4, ra[s]), __builtin_bit_cast(bf16x4, rb[s]));
        xl[s] = xh[s];
      }
    } else {
      f32x4 ra[S], rb[S];
#pragma unroll
      for (int s = 0; s < S; ++s) {
        ra[s] = *reinterpret_cast<const f32x4*>(row + 32 * s);
        rb[s] = *reinterpret_cast<const f32x4*>(row + 32 * s + SECOND);
      }
#pragma unroll
      for (int s = 0; s < S; ++s) {
        bf16x4 h0, l0, h1, l1;
        ly_split4(ra[s], h0, l0);
        ly_split4(rb[s], h1, l1);
        xh[s] = ly_cat8(h0, h1);
        xl[s] = ly_cat8(l0, l1);
      }
    }
    f32x4 acc[2] = {ly_zero4(), ly_zero4()};
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x8 wh = __builtin_bit_cast(bf16x8, ly_det_w[((t * S + s) * PL + 0) * 64 + lane]);
        if constexpr (PL == 2) {
          const bf16x8 wl = __builtin_bit_cast(bf16x8, ly_det_w[((t * S + s) * PL + 1) * 64 + lane]);
          acc[t] = ly_mfma3(wh, wl, xh[s], xl[s], acc[t]);
        } else {
          acc[t] = ly_mfma_bf16(wh, xh[s], acc[t]);
        }
      }
    // accumulators (+ bias) -> the wave's [16 px][36] fp32 LDS tile -> lanes walk the tile's outputs in MEMORY order (anchor, pixel, o): p and z
    // of 16 consecutive pixels are na runs of 16 no contiguous floats each, so every store instruction writes consecutive addresses
    // (written from the accumulator layout — lane = (pixel, 4 channels) — the same bytes took 16 scattered 4-byte stores per lane: 19.6 us at
    // 80 x 80 x 128, bs=16).
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[t][r] + cb[t][r];
      *reinterpret_cast<f32x4*>(ot + px * LY_DET_LD + 16 * t + 4 * g) = v;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0): the tile is written (one wave: no block barrier)
    const long m0 = (long)tile * 16;
    const int run = 16 * no, total = na * run;
    for (int e = lane; e < total; e += 64) {
      const int a = e / run, q = e - a * run;
      const int pl = q / no, o = q - pl * no;
      const long mm = m0 + pl;
      if (mm >= M) continue;
      const int n = ly_fdiv((int)mm, HW, invHW);
      const int hw = (int)mm - n * HW;
      const float v = ot[pl * LY_DET_LD + a * no + o];
      p[(((long)n * na + a) * HW + hw) * no + o] = v;
      if (z) {
        const int h = ly_fdiv(hw, W, invW);
        const int w = hw - h * W;
        const float sg = ly_sigmoid(v);
        float r = sg;
        if (o == 0) r = (sg * 2.f + ((float)w - 0.5f)) * stride;
        else if (o == 1) r = (sg * 2.f + ((float)h - 0.5f)) * stride;
        else if (o == 2 || o == 3) { const float d = sg * 2.f; r = d * d * (anchors[a * 2 + (o - 2)] * stride); }
        z[((long)n * zrows + zoff + (long)a * HW + hw) * no + o] = r;
      }
    }
    __builtin_amdgcn_wave_barrier();                         // (the next tile overwrites the LDS tile)
  }
}

template <typename T, int S>
static void detect_level_launch(const int nat, const void* x, int ldx, long M, int H, int W, const void* wp, const float* bias, int na, int no, const float* anchors,
                                float stride, float* p, float* z, long zrows, long zoff, hipStream_t st) {
  const int ntiles = (int)((M + 15) / 16);
  // tiles per wave: one until every SIMD holds ~4 waves, then more (the 2 x K weight rows a block stages are amortised over 4 tpw tiles).
  // Measured at bs=16 (one round of blocks): every further tile per wave adds ~5 us — a tile is one serial chain of memory round trips.
  int tpw = ntiles / (256 * 4 * 4);
  tpw = tpw < 1 ? 1 : tpw > 8 ? 8 : tpw;
  const int per_block = (LY_THREADS / 64) * tpw;
  const size_t lds = (size_t)2 * S * LyT<T>::PL * 64 * sizeof(uint4) + (LY_THREADS / 64) * 16 * LY_DET_LD * sizeof(float);
  const dim3 grid((unsigned)((ntiles + per_block - 1) / per_block));
  if (nat)
    hipLaunchKernelGGL((ly_detect_level_kernel<T, S, true>), grid, dim3(LY_THREADS), lds, st, reinterpret_cast<const T*>(x), ldx, M, H, W,
                       reinterpret_cast<const uint4*>(wp), bias, na, no, anchors, stride, p, z, zrows, zoff, ntiles, tpw);
  else
    hipLaunchKernelGGL((ly_detect_level_kernel<T, S, false>), grid, dim3(LY_THREADS), lds, st, reinterpret_cast<const T*>(x), ldx, M, H, W,
                       reinterpret_cast<const uint4*>(wp), bias, na, no, anchors, stride, p, z, zrows, zoff, ntiles, tpw);
}

extern "C" int ly_detect_level(const void* x, int ldx, int n_img, int H, int W, int K, const void* wp, int nat, const float* bias, int na, int no,
                               const float* anchors, float stride, float* p, float* z, long zrows, long zoff, int dtype, void* stream) {
  LY_CHECK_DTYPE(dtype, "detect_level");
  LY_CHECK(x && wp && bias && anchors && p && n_img > 0 && H > 0 && W > 0, "detect_level: null pointer / bad sizes");
  LY_CHECK(na * no <= 32 && na > 0 && no > 0, "detect_level: na*no = %d exceeds the 32 output rows of the kernel", na * no);
  const int vw = dtype == LY_BF16 ? 8 : 4;
  LY_CHECK((K & 31) == 0 && ldx >= K && (ldx % vw) == 0 && ((uintptr_t)x & 15) == 0, "detect_level: K = %d must be a multiple of 32, rows 16-byte aligned", K);
  const long M = (long)n_img * H * W;
  LY_CHECK(M < (1L << 24), "detect_level: too many pixels");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int S = K / 32;
#define LY_DET(S_) LY_WITH_T(dtype, (detect_level_launch<T, S_>(nat, x, ldx, M, H, W, wp, bias, na, no, anchors, stride, p, z, zrows, zoff, st)))
  if (S == 2) LY_DET(2);
  else if (S == 4) LY_DET(4);
  else if (S == 8) LY_DET(8);
  else if (S == 16 && dtype == LY_BF16) detect_level_launch<__bf16, 16>(nat, x, ldx, M, H, W, wp, bias, na, no, anchors, stride, p, z, zrows, zoff, st);
  else { ly_set_error("detect_level: K = %d is not built (64, 128, 256; 512 in bf16)", K); return -1; }
#undef LY_DET
  LY_LAUNCH_CHECK();
  return 0;
}

extern "C" int ly_detect_level_ok(int K, int na, int no, int dtype) {
  const int S = K / 32;
  return (K % 32 == 0) && na * no <= 32 && (S == 2 || S == 4 || S == 8 || (S == 16 && dtype == LY_BF16));
}
