// PatchEmbed on an RGB image: Conv2d(3, N, 4, 4) [+ BatchNorm] as 4 x 4 stride-4 patches of the NCHW batch -> [pixels, N]   (gfx950 only)
// (models/common.py:1537-1550 PatchEmbed_FasterNet; the first launch of every forward, the only one that reads the image).
//
// The generic contraction (ly_gemm_kernel_d2, LY_GATHER_PATCH_NCHW*) stages 64-pixel tiles through LDS behind two barriers per tile; on this
// shape (K = 48, N = 24 .. 80) it took the same ~46 us per 32 images whether the image was uint8, bf16 or fp32 — paced by the tile loop, not by
// bytes.  Here nothing is shared between waves and nothing goes through LDS: with the k order of the packed weights (ly_tile.hpp: lane l of a
// k-step holds k = 4 (l >> 4) + {0..3} and 16 + 4 (l >> 4) + {0..3}) the B operand of `v_mfma_f32_16x16x32_bf16` for 16 output pixels is, per
// lane, the 4-pixel row segment (channel c, row 4 h + (l >> 4)) of pixel l & 15 for c = 0, 1 (k-step 0) and c = 2 (k-step 1, upper half zero)
// — three vector loads per lane and pixel tile, straight from the image into the operand registers; 16 lanes read 16 adjacent segments.
// A wave keeps all weight fragments, walks pixel tiles U at a time and has the next U tiles' loads in flight while it contracts.
//   fp32 / fp16 image: x = hi + lo (bf16x3 products, as every LY_F32 source); bf16 image: the values as they are; uint8 image: the integers
//   0 .. 255 are exact in bf16 and 1/255 is folded into the epilogue scale.  Weights always hi + lo.
#include "ly_common.hpp"
#include "ly_tile.hpp"
#include "ly_gemm.hpp"

#define LY_P4_U 4

template <typename TI> struct LyP4;
template <> struct LyP4<float> { static constexpr bool SPLIT = true; };
template <> struct LyP4<ly_f16img> { static constexpr bool SPLIT = true; };
template <> struct LyP4<ly_bf16img> { static constexpr bool SPLIT = false; };
template <> struct LyP4<unsigned char> { static constexpr bool SPLIT = false; };

// raw 4-pixel vector -> bf16 operand halves
__device__ __forceinline__ void ly_p4_cvt(const f32x4 r, bf16x4& hi, bf16x4& lo) { ly_split4(r, hi, lo); }
__device__ __forceinline__ void ly_p4_cvt(const ly_h4raw r, bf16x4& hi, bf16x4& lo) {
  ly_split4(__builtin_convertvector(__builtin_bit_cast(ly_f16x4, r.r), f32x4), hi, lo);
}
__device__ __forceinline__ void ly_p4_cvt(const ly_u32x2 r, bf16x4& hi, bf16x4&) { hi = __builtin_bit_cast(bf16x4, r); }
__device__ __forceinline__ void ly_p4_cvt(const unsigned r, bf16x4& hi, bf16x4&) {
  const f32x4 v = (f32x4){(float)(r & 255u), (float)((r >> 8) & 255u), (float)((r >> 16) & 255u), (float)(r >> 24)};
  hi = __builtin_convertvector(v, bf16x4);
}

template <typename TI, typename TO, int MT, bool STATS>
__global__ __launch_bounds__(LY_THREADS) void ly_patch4_kernel(const LyGemmParams P, const int ntiles, const float in_scale) {
  using RV = typename LyT<TI>::RV;
  constexpr bool SPLIT = LyP4<TI>::SPLIT;
  constexpr int U = LY_P4_U;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = lane & 15, g = lane >> 4;
  const int nwaves = gridDim.x * (LY_THREADS / 64);
  const int w0 = blockIdx.x * (LY_THREADS / 64) + wave;
  const TI* const img = reinterpret_cast<const TI*>(P.a0);
  TO* const out = reinterpret_cast<TO*>(P.out);
  const int HW = P.H * P.W;
  const float invHW = 1.f / (float)HW, invW = 1.f / (float)P.W;
  const long plane = (long)P.Hin * P.Win;

  // weight fragments [t][s][plane]: resident
  bf16x8 wh[MT][2], wl[MT][2];
  {
    const uint4* wpk = reinterpret_cast<const uint4*>(P.wp);
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        wh[t][s] = __builtin_bit_cast(bf16x8, wpk[((t * 2 + s) * 2 + 0) * 64 + lane]);
        wl[t][s] = __builtin_bit_cast(bf16x8, wpk[((t * 2 + s) * 2 + 1) * 64 + lane]);
      }
  }
  float esc[MT][4], esh[MT][4];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = 16 * t + 4 * g + r;
      const bool ok = c < P.N;
      esc[t][r] = ((ok && P.e_scale) ? P.e_scale[c] : 1.f) * in_scale;
      esh[t][r] = (ok && P.e_shift) ? P.e_shift[c] : 0.f;
    }
  f32x4 sum1[STATS ? MT : 1], sum2[STATS ? MT : 1];
  if constexpr (STATS) {
#pragma unroll
    for (int t = 0; t < MT; ++t) { sum1[t] = ly_zero4(); sum2[t] = ly_zero4(); }
  }
  const int act = P.act;

  RV xv[2][U][3];
  auto issue = [&](auto bC, const int tile0) {
    constexpr int b = decltype(bC)::value;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int tile = tile0 + u * nwaves;
      long m = (long)tile * 16 + p;
      if (!(tile < ntiles && m < P.M)) m = 0;                 // surplus lanes / tiles re-read pixel 0 (never stored)
      const int n = ly_fdiv((int)m, HW, invHW);
      const int rem = (int)m - n * HW;
      const int h = ly_fdiv(rem, P.W, invW);
      const int w = rem - h * P.W;
      const TI* src = img + ((long)n * 3 * P.Hin + 4 * h + g) * P.Win + 4 * w;
#pragma unroll
      for (int c = 0; c < 3; ++c) xv[b][u][c] = ly_ldrv<TI>(src + c * plane);
    }
  };
  auto compute = [&](auto bC, const int tile0) {
    constexpr int b = decltype(bC)::value;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int tile = tile0 + u * nwaves;
      bf16x4 h0, h1, h2, l0, l1, l2;
      const bf16x4 z = __builtin_bit_cast(bf16x4, (ly_u32x2){0u, 0u});
      l0 = z; l1 = z; l2 = z;
      ly_p4_cvt(xv[b][u][0], h0, l0);
      ly_p4_cvt(xv[b][u][1], h1, l1);
      ly_p4_cvt(xv[b][u][2], h2, l2);
      const bf16x8 xh0 = ly_cat8(h0, h1), xh1 = ly_cat8(h2, z);
      const bf16x8 xl0 = ly_cat8(l0, l1), xl1 = ly_cat8(l2, z);
      const long m = (long)tile * 16 + p;
      const bool live = tile < ntiles && m < P.M;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        f32x4 acc = ly_zero4();
        if constexpr (SPLIT) {
          acc = ly_mfma_bf16(wh[t][0], xl0, acc);
          acc = ly_mfma_bf16(wh[t][1], xl1, acc);
        }
        acc = ly_mfma_bf16(wl[t][0], xh0, acc);
        acc = ly_mfma_bf16(wl[t][1], xh1, acc);
        acc = ly_mfma_bf16(wh[t][0], xh0, acc);
        acc = ly_mfma_bf16(wh[t][1], xh1, acc);
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[r] * esc[t][r] + esh[t][r];
        if constexpr (STATS) {
          if (live) { sum1[t] += v; sum2[t] += v * v; }
        }
        if (out && live && 16 * t + 4 * g < P.N) ly_st4<TO>(out + m * P.ldo + 16 * t + 4 * g, ly_act4(v, act));
      }
    }
  };

  const int step = U * nwaves;
  int tile0 = w0;
  if (tile0 >= ntiles) return;
  issue(LyIc<0>(), tile0);
  while (true) {
    const int next = tile0 + step;
    issue(LyIc<1>(), next);
    compute(LyIc<0>(), tile0);
    if (next >= ntiles) break;
    tile0 = next + step;
    issue(LyIc<0>(), tile0);
    compute(LyIc<1>(), next);
    if (tile0 >= ntiles) break;
  }
  if constexpr (STATS) {
#pragma unroll
    for (int t = 0; t < MT; ++t)
      if (16 * t + 4 * g < P.N) ly_stats_flush(P.stats, P.N, 16 * t + 4 * g, sum1[t], sum2[t]);
  }
}

template <typename TI, typename TO, int MT>
static void patch4_launch_mt(const LyGemmParams& P, int ntiles, float in_scale, hipStream_t st) {
  // a wave walks ~2 U-groups at least; 8 blocks of four waves per CU at most
  long blocks = ((long)ntiles + 4 * LY_P4_U * 2 - 1) / (4 * LY_P4_U * 2);
  blocks = blocks < 1 ? 1 : blocks > 2048 ? 2048 : blocks;
  if (P.stats) hipLaunchKernelGGL((ly_patch4_kernel<TI, TO, MT, true>), dim3((unsigned)blocks), dim3(LY_THREADS), 0, st, P, ntiles, in_scale);
  else hipLaunchKernelGGL((ly_patch4_kernel<TI, TO, MT, false>), dim3((unsigned)blocks), dim3(LY_THREADS), 0, st, P, ntiles, in_scale);
}

template <typename TI, typename TO>
static void patch4_launch(const LyGemmParams& P, float in_scale, hipStream_t st) {
  const int ntiles = (int)((P.M + 15) / 16);
  const int mt = (P.N + 15) / 16;
  if (mt <= 2) patch4_launch_mt<TI, TO, 2>(P, ntiles, in_scale, st);
  else if (mt == 3) patch4_launch_mt<TI, TO, 3>(P, ntiles, in_scale, st);
  else if (mt == 4) patch4_launch_mt<TI, TO, 4>(P, ntiles, in_scale, st);
  else patch4_launch_mt<TI, TO, 5>(P, ntiles, in_scale, st);
}

// 1 = launched; 0 = not this kernel's shape (the caller goes on to the generic contraction)
int ly_patch4_try(const LyGemmParams& P, hipStream_t st) {
  const bool image = P.gather == LY_GATHER_PATCH_NCHW || P.gather == LY_GATHER_PATCH_NCHW_U8 || P.gather == LY_GATHER_PATCH_NCHW_BF16 ||
                     P.gather == LY_GATHER_PATCH_NCHW_F16;
  if (!image || P.Cin != 3 || P.K != 48 || P.N > 80 || P.N < 20 || (P.N & 3) != 0 || (P.out && (P.ldo & 3) != 0) || P.pro != LY_PRO_NONE || P.rowscale || P.res) return 0;
  if (P.Hin != 4 * P.H || P.Win != 4 * P.W) return 0;
  if (P.dtype == LY_BF16) {
    if (P.gather == LY_GATHER_PATCH_NCHW) patch4_launch<float, __bf16>(P, 1.f, st);
    else if (P.gather == LY_GATHER_PATCH_NCHW_U8) patch4_launch<unsigned char, __bf16>(P, 1.f / 255.f, st);
    else if (P.gather == LY_GATHER_PATCH_NCHW_BF16) patch4_launch<ly_bf16img, __bf16>(P, 1.f, st);
    else patch4_launch<ly_f16img, __bf16>(P, 1.f, st);
  } else {
    if (P.gather == LY_GATHER_PATCH_NCHW) patch4_launch<float, float>(P, 1.f, st);
    else if (P.gather == LY_GATHER_PATCH_NCHW_U8) patch4_launch<unsigned char, float>(P, 1.f / 255.f, st);
    else return 0;
  }
  return 1;
}

// -------------------------------------------------------------------------------------------------
// PatchEmbed's WEIGHT GRADIENT straight from the uint8 image (round 6; models/common.py:1537-1550 under autograd, train.py:327):
//     dW[n][(c, ky, kx)] += scale * sum_m du[m][n] * img[n_img(m)][c][4 ho(m) + ky][4 wo(m) + kx]
// Before: ly_patch4_rows_u8 wrote the space-to-depth rows [M][48] in the storage type (157 MB at bs = 64, 640 x 640) and ly_wgrad read
// them back beside du — 57 + 84 us for 157 MB of inputs that matter.  Here a block walks units of 128 consecutive output pixels: the unit's
// du rows (16-byte loads) and its twelve (channel, ky) image row pieces (4-byte loads: the four kx of a patch) are written TRANSPOSED into
// LDS — duT[channel][pixel], rowsT[(c, ky, kx)][pixel], the pixel values as the integers they are (exact in bf16) — so that both MFMA operands
// of the contraction over pixels are plain 8-byte row reads (k-set of ly_tile.hpp: pixels 32 s + 4 q + {0..3} and + 16); wave w contracts
// k-step w of the unit.  The waves' accumulators meet in LDS when the walk ends; the block stores ONE [N][48] partial (times `scale`,
// 1/255) and ly_sum_rows folds the partials in block order: no atomics, the same bits in every run.
// -------------------------------------------------------------------------------------------------
#define LY_P4W_PX 128
template <int MTN>
__global__ __launch_bounds__(LY_THREADS) void ly_patch4_wgrad_u8_kernel(const unsigned char* __restrict__ img, const int H, const int W, const long M,
                                                                        const __bf16* __restrict__ du, const int lddu, const int N, const float scale,
                                                                        float* __restrict__ slab) {
  constexpr int KE = 48, NTL = MTN * 3;
  constexpr int RS = (LY_P4W_PX + 8) * 2;                      // bytes per LDS row: 272 (68 dwords: a 32-lane fragment read covers the 64 banks once)
  constexpr int TILE_B = (MTN * 16 + KE) * RS, RED_B = 4 * NTL * 1024;
  __shared__ __attribute__((aligned(16))) char lds[TILE_B > RED_B ? TILE_B : RED_B];
  char* const dut = lds;
  char* const rwt = lds + MTN * 16 * RS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int Wo = W >> 2, Ho = H >> 2;
  const int HWo = Ho * Wo;
  const float invHWo = 1.f / (float)HWo, invWo = 1.f / (float)Wo;
  const long units = (M + LY_P4W_PX - 1) / LY_P4W_PX;
  const int NV8 = N >> 3;                                      // 16-byte vectors per du row
  // rows N .. 16 MTN - 1 of duT are never written: zero once
  for (int i = tid; i < (MTN * 16 - N) * (RS / 4); i += LY_THREADS) reinterpret_cast<unsigned*>(dut + N * RS)[i] = 0u;

  // staging plan.  image: item = tid + 256 e -> row r = (c, ky) = tid / 128 + 2 e, pixel j = tid % 128 (one pixel per thread: one index
  // decomposition per unit); du: item -> (pixel, vector)
  const int j = tid & (LY_P4W_PX - 1), r0 = tid >> 7;
  unsigned iv[6];
  constexpr int ND = MTN;                                    // 16-byte du items per thread: 128 pixels x (N / 8 <= 2 MTN) vectors over 256 threads
  ly_u32x4 dv[ND];
  auto prefetch = [&](const long unit) {
    const long m = unit * LY_P4W_PX + j;
    const bool ok = m < M;
    const int mc = (int)(ok ? m : M - 1);
    const int n = ly_fdiv(mc, HWo, invHWo);
    const int rem = mc - n * HWo;
    const int ho = ly_fdiv(rem, Wo, invWo), wo = rem - ho * Wo;
    const unsigned char* const p0 = img + ((long)n * 3 * H + 4 * ho) * W + 4 * wo;
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      const int r = r0 + 2 * e, c = r >> 2, ky = r & 3;
      iv[e] = *reinterpret_cast<const unsigned*>(p0 + ((long)c * H + ky) * W);
    }
#pragma unroll
    for (int e = 0; e < ND; ++e) {
      const int item = tid + LY_THREADS * e;
      const int px = item / NV8, v = item - px * NV8;
      const long mm = unit * LY_P4W_PX + px;
      const bool okd = px < LY_P4W_PX && mm < M;
      dv[e] = *reinterpret_cast<const ly_u32x4*>(du + (okd ? mm * lddu + 8 * v : 0));
    }
  };
  auto commit = [&](const long unit) {
    const bool ok = unit * LY_P4W_PX + j < M;
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      const int r = r0 + 2 * e;
      const unsigned v = ok ? iv[e] : 0u;
      __bf16* const d = reinterpret_cast<__bf16*>(rwt + (4 * r) * RS) + j;
      d[0] = (__bf16)(float)(v & 255u);
      d[RS / 2] = (__bf16)(float)((v >> 8) & 255u);
      d[2 * (RS / 2)] = (__bf16)(float)((v >> 16) & 255u);
      d[3 * (RS / 2)] = (__bf16)(float)(v >> 24);
    }
#pragma unroll
    for (int e = 0; e < ND; ++e) {
      const int item = tid + LY_THREADS * e;
      const int px = item / NV8, v = item - px * NV8;
      if (px < LY_P4W_PX) {
        const bool okd = unit * LY_P4W_PX + px < M;
        unsigned short* const d = reinterpret_cast<unsigned short*>(dut + (8 * v) * RS) + px;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned w2 = okd ? dv[e][k] : 0u;
          d[(2 * k) * (RS / 2)] = (unsigned short)(w2 & 0xffffu);
          d[(2 * k + 1) * (RS / 2)] = (unsigned short)(w2 >> 16);
        }
      }
    }
  };

  f32x4 acc[MTN][3];
#pragma unroll
  for (int t = 0; t < MTN; ++t)
#pragma unroll
    for (int n = 0; n < 3; ++n) acc[t][n] = ly_zero4();
  long unit = blockIdx.x;
  if (unit < units) prefetch(unit);
  const int koff = 2 * (32 * wave + 4 * lq);                   // the wave's k-step, the lane's first pixel group (bytes)
  for (; unit < units; unit += gridDim.x) {
    __syncthreads();                                           // the previous unit's fragment reads are done
    commit(unit);
    const long nxt = unit + gridDim.x < units ? unit + gridDim.x : unit;      // (the last unit re-requests itself: straight-line loads)
    prefetch(nxt);
    __syncthreads();
    bf16x8 a[MTN], b[3];
#pragma unroll
    for (int t = 0; t < MTN; ++t) {
      const char* p = dut + (16 * t + li) * RS + koff;
      a[t] = ly_cat8(*reinterpret_cast<const bf16x4*>(p), *reinterpret_cast<const bf16x4*>(p + 32));
    }
#pragma unroll
    for (int n = 0; n < 3; ++n) {
      const char* p = rwt + (16 * n + li) * RS + koff;
      b[n] = ly_cat8(*reinterpret_cast<const bf16x4*>(p), *reinterpret_cast<const bf16x4*>(p + 32));
    }
#pragma unroll
    for (int t = 0; t < MTN; ++t)
#pragma unroll
      for (int n = 0; n < 3; ++n) acc[t][n] = ly_mfma_bf16(a[t], b[n], acc[t][n]);
  }
  // the four waves' partial tiles meet in LDS; D: lane (i = column l & 15, q) holds rows 4 q + r
  __syncthreads();
  float* const red = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int t = 0; t < MTN; ++t)
#pragma unroll
    for (int n = 0; n < 3; ++n) *reinterpret_cast<f32x4*>(red + ((wave * NTL + t * 3 + n) * 64 + lane) * 4) = acc[t][n];
  __syncthreads();
  float* const out = slab + (long)blockIdx.x * N * KE;
  for (int o = tid; o < NTL * 256; o += LY_THREADS) {
    const float s = (red[o] + red[NTL * 256 + o]) + (red[2 * NTL * 256 + o] + red[3 * NTL * 256 + o]);
    const int tile = o >> 8, l = (o >> 2) & 63, rr = o & 3;
    const int row = 16 * (tile / 3) + 4 * (l >> 4) + rr, col = 16 * (tile % 3) + (l & 15);
    if (row < N) out[row * KE + col] = s * scale;
  }
}

extern "C" int ly_patch4_wgrad_u8(const unsigned char* img, int n_img, int C, int H, int W, const void* du, int lddu, int N, float scale, float* slab,
                                  long slab_floats, float* dw, int dtype, void* stream) {
  LY_CHECK(img && du && slab && dw && n_img > 0 && H > 0 && W > 0, "patch4_wgrad_u8: bad arguments");
  LY_CHECK(C == 3 && (H & 3) == 0 && (W & 3) == 0, "patch4_wgrad_u8: built for RGB images with H, W multiples of 4 (C=%d H=%d W=%d)", C, H, W);
  LY_CHECK(dtype == LY_BF16, "patch4_wgrad_u8: bf16 gradients only (fp32 storage keeps ly_patch4_rows_u8 + ly_wgrad)");
  LY_CHECK(N >= 8 && N <= 64 && (N & 7) == 0 && (lddu & 7) == 0 && ((uintptr_t)du & 15) == 0 && ((uintptr_t)img & 3) == 0,
           "patch4_wgrad_u8: N=%d must be a multiple of 8 up to 64, du rows 16-byte aligned", N);
  const long M = (long)n_img * (H >> 2) * (W >> 2);
  LY_CHECK(M < (1L << 24), "patch4_wgrad_u8: M=%ld pixels exceeds the 2^24 limit of the fast index path", M);
  const long units = (M + LY_P4W_PX - 1) / LY_P4W_PX;
  long blocks = units < 1536 ? units : 1536;                   // six 22-24 KB blocks per CU
  LY_CHECK(blocks * N * 48 <= slab_floats, "patch4_wgrad_u8: the slab holds %ld floats, %ld needed", slab_floats, blocks * N * 48);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const __bf16* d = reinterpret_cast<const __bf16*>(du);
  const int mtn = (N + 15) / 16;
  if (mtn == 1) hipLaunchKernelGGL((ly_patch4_wgrad_u8_kernel<1>), dim3((unsigned)blocks), dim3(LY_THREADS), 0, st, img, H, W, M, d, lddu, N, scale, slab);
  else if (mtn == 2) hipLaunchKernelGGL((ly_patch4_wgrad_u8_kernel<2>), dim3((unsigned)blocks), dim3(LY_THREADS), 0, st, img, H, W, M, d, lddu, N, scale, slab);
  else if (mtn == 3) hipLaunchKernelGGL((ly_patch4_wgrad_u8_kernel<3>), dim3((unsigned)blocks), dim3(LY_THREADS), 0, st, img, H, W, M, d, lddu, N, scale, slab);
  else hipLaunchKernelGGL((ly_patch4_wgrad_u8_kernel<4>), dim3((unsigned)blocks), dim3(LY_THREADS), 0, st, img, H, W, M, d, lddu, N, scale, slab);
  LY_LAUNCH_CHECK();
  return ly_sum_rows(slab, blocks, (long)N * 48, (long)N * 48, dw, 1, stream);
}
