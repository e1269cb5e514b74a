// Small reduction / gating kernels of the two attention blocks (gfx950; activations T = float / __bf16, tables fp32).  All are
// bandwidth-trivial next to the convolutions; they exist so that the big tensors are read a minimal
// number of times and the 9x-expanded RFCBAM tensor is never written.
//
//   ly_pool_hw        CoordAtt pool_h / pool_w               (models/common.py:1598-1599)
//   ly_coordatt_mlp   conv1+bn1+h_swish, conv_h/conv_w+sigmoid (models/common.py:1600-1607)
//   ly_colsum         spatial sum per (image, channel)       (SE gap, models/rfa.py:90)
//   ly_se_mlp         sigmoid(Wb relu(Wa mean))              (models/rfa.py:78-92)
//   ly_rfcbam_stats   max_c / mean_c of relu(bn(generate(x))) on the expanded grid (models/rfa.py:115-126)
//   ly_rfa_map        sigmoid(conv3x3([max, mean]))          (models/rfa.py:107,127)
#include <float.h>

#include "ly_tile.hpp"
#include "ly_params.h"


// ---------------------------------------------------------------------------------------------------
// generic strided reduction:  out[o, c] = scale * sum_{j < L} x[base(o) + j*stride + c]
// one block per output row o; threads = (c4, j-lane)
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(LY_THREADS) void ly_pool_hw_kernel(const T* __restrict__ x, int ldx, int H, int W, int C,
                                                                 float* __restrict__ pool) {
  // block b -> (n, pos); pos < H: mean over w of row pos; else mean over h of column pos-H
  __shared__ f32x4 red[LY_THREADS];
  const int L = H + W;
  const int bid = ly_xcd_remap((int)blockIdx.x, (int)gridDim.x);     // an image's H row blocks and W column blocks read the same cache lines: one XCD's L2
  const int n = bid / L, pos = bid - n * L;
  const int nc4 = C >> 2;
  const int tid = threadIdx.x;
  const int groups = LY_THREADS / nc4;
  const bool row = pos < H;
  const int len = row ? W : H;
  const long base = row ? ((long)n * H + pos) * W : ((long)n * H) * W + (pos - H);
  const long step = row ? 1 : W;
  const int c4 = tid % nc4, j0 = tid / nc4;       // C <= 1024 so nc4 <= 256
  f32x4 s = ly_zero4();
  if (j0 < groups) {
#pragma unroll 4
    for (int j = j0; j < len; j += groups) s += ly_ld4<T>(x + (base + j * step) * ldx + 4 * c4);
  }
  red[tid] = s;
  __syncthreads();
  if (j0 == 0) {
    for (int g = 1; g < groups; ++g) s += red[g * nc4 + c4];
    const float inv = 1.f / (float)len;
    ly_stg4(pool + ((long)n * L + pos) * C + 4 * c4, s * inv);
  }
}

extern "C" int ly_pool_hw(const void* x, int ldx, int n_img, int H, int W, int C, float* pool, int dtype, void* stream) {
  LY_CHECK_DTYPE(dtype, "pool_hw");
  LY_CHECK(x && pool && (C & 3) == 0 && (ldx & 3) == 0 && C <= 1024, "pool_hw: bad arguments (C=%d ldx=%d)", C, ldx);
  LY_WITH_T(dtype, hipLaunchKernelGGL(ly_pool_hw_kernel<T>, dim3(n_img * (H + W)), dim3(LY_THREADS), 0, reinterpret_cast<hipStream_t>(stream),
                                      reinterpret_cast<const T*>(x), ldx, H, W, C, pool));
  LY_LAUNCH_CHECK();
  return 0;
}

// y = hswish(W1' pool + b1')  (BN folded),  a = sigmoid(Wx y + bx);  one block per (n, pos)
__global__ __launch_bounds__(LY_THREADS) void ly_coordatt_mlp_kernel(const float* __restrict__ pool, int H, int W, int C, int mip,
                                                                      const float* __restrict__ w1, const float* __restrict__ b1,
                                                                      const float* __restrict__ sc, const float* __restrict__ sh,
                                                                      const float* __restrict__ wh, const float* __restrict__ bh,
                                                                      const float* __restrict__ ww, const float* __restrict__ bw,
                                                                      float* __restrict__ a_h, float* __restrict__ a_w) {
  __shared__ float ys[64];
  const int L = H + W;
  const int n = blockIdx.x / L, pos = blockIdx.x - n * L;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* p = pool + ((long)n * L + pos) * C;
  for (int m = wave; m < mip; m += 4) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += w1[m * C + c] * p[c];
    s = ly_group_sum(s, 64);
    if (lane == 0) ys[m] = ly_hswish(sc ? (s + b1[m]) * sc[m] + sh[m] : s + b1[m]);
  }
  __syncthreads();
  const bool isrow = pos < H;
  const float* wx = isrow ? wh : ww;
  const float* bx = isrow ? bh : bw;
  float* out = isrow ? a_h + ((long)n * H + pos) * C : a_w + ((long)n * W + (pos - H)) * C;
  for (int c = tid; c < C; c += LY_THREADS) {
    float s = bx[c];
    for (int m = 0; m < mip; ++m) s += wx[c * mip + m] * ys[m];
    out[c] = ly_sigmoid(s);
  }
}

extern "C" int ly_coordatt_mlp(const float* pool, int n_img, int H, int W, int C, int mip, const float* w1, const float* b1,
                               const float* sc, const float* sh, const float* wh, const float* bh, const float* ww, const float* bw, float* a_h, float* a_w,
                               void* stream) {
  LY_CHECK(pool && w1 && b1 && wh && bh && ww && bw && a_h && a_w && (!sc == !sh), "coordatt_mlp: null pointer");
  LY_CHECK(mip > 0 && mip <= 64, "coordatt_mlp: mip=%d out of range", mip);
  hipLaunchKernelGGL(ly_coordatt_mlp_kernel, dim3(n_img * (H + W)), dim3(LY_THREADS), 0, reinterpret_cast<hipStream_t>(stream),
                     pool, H, W, C, mip, w1, b1, sc, sh, wh, bh, ww, bw, a_h, a_w);
  LY_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// SE: partial spatial sums then the two tiny linears
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(LY_THREADS) void ly_colsum_kernel(const T* __restrict__ x, int ldx, int P, int C, int slices,
                                                                float* __restrict__ part) {
  __shared__ f32x4 red[LY_THREADS];
  const int n = blockIdx.x / slices, sl = blockIdx.x - n * slices;
  const int nc4 = C >> 2, tid = threadIdx.x;
  const int per = (P + slices - 1) / slices;
  const int j_lo = sl * per, j_hi = (j_lo + per) < P ? (j_lo + per) : P;
  const int groups = LY_THREADS / nc4;              // C <= 1024
  const int c4 = tid % nc4, j0 = tid / nc4;
  f32x4 s = ly_zero4();
  if (j0 < groups) {
#pragma unroll 4
    for (int j = j_lo + j0; j < j_hi; j += groups) s += ly_ld4<T>(x + ((long)n * P + j) * ldx + 4 * c4);
  }
  red[tid] = s;
  __syncthreads();
  if (j0 == 0) {
    for (int g = 1; g < groups; ++g) s += red[g * nc4 + c4];
    ly_stg4(part + ((long)n * slices + sl) * C + 4 * c4, s);
  }
}

// LDS floats of the SE step: g[C] + hid[R] + red[LY_THREADS][4], plus — when they fit 48 KiB — both weight matrices (2 R C): staged while the
// pooling partials are on their way, so that the step is ONE global round trip instead of three dependent ones (partials, fc1 rows, fc2 rows:
// 12.6 us per launch of ly_rfcbam_mid was this chain, not the get_weight map that runs beside it)
__host__ __device__ __forceinline__ size_t ly_se_lds_floats(int C, int R) {
  const size_t base = (size_t)((C + R + 3) & ~3) + 4 * LY_THREADS;
  const size_t w = 2 * (size_t)R * C;
  return (C & 3) == 0 && w * 4 <= 48 * 1024 ? base + w : base;
}

__device__ __forceinline__ void ly_se_mlp_body(float* sm, const int n, const float* __restrict__ part, int slices, int C, float inv_hw,
                                               const float* __restrict__ wa, const float* __restrict__ wb, int R, float* __restrict__ ca) {
  float* g = sm;                     // g[C] + hid[R] + red[LY_THREADS][4] (+ wa[R][C] + wb[C][R])
  float* hid = sm + C;
  f32x4* red = reinterpret_cast<f32x4*>(sm + ((C + R + 3) & ~3));
  const size_t base = (size_t)((C + R + 3) & ~3) + 4 * LY_THREADS;
  const bool staged = ly_se_lds_floats(C, R) > base;
  float* was = sm + base;
  float* wbs = was + (size_t)R * C;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // slice partials -> channel means.  All 256 threads load: thread = (slice group, 4 channels), so the up to 128 slices
  // are a handful of independent float4 loads per thread instead of a 128-long chain per channel; groups meet in LDS.
  const int nq = C >> 2;                 // channel quads: C % 4 == 0 and C <= 1024, so nq <= 256
  const int ng = LY_THREADS / nq;        // slice groups
  const int q = tid % nq, sg = tid / nq;
  f32x4 s4 = ly_zero4();
  if (sg < ng) {
    int sl = sg;
    for (; sl + 3 * ng < slices; sl += 4 * ng) {         // four loads in flight per trip
      const f32x4 a0 = ly_ldg4(part + ((long)n * slices + sl) * C + 4 * q), a1 = ly_ldg4(part + ((long)n * slices + sl + ng) * C + 4 * q);
      const f32x4 a2 = ly_ldg4(part + ((long)n * slices + sl + 2 * ng) * C + 4 * q), a3 = ly_ldg4(part + ((long)n * slices + sl + 3 * ng) * C + 4 * q);
      s4 += a0; s4 += a1; s4 += a2; s4 += a3;
    }
    for (; sl < slices; sl += ng) s4 += ly_ldg4(part + ((long)n * slices + sl) * C + 4 * q);
  }
  if (staged) {
    const int nw4 = (R * C) >> 2;                        // float4 items per matrix (C % 4 == 0)
    for (int i = tid; i < nw4; i += 4 * LY_THREADS) {
      f32x4 va[4], vb[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int j = i + k * LY_THREADS < nw4 ? i + k * LY_THREADS : i;
        va[k] = ly_ldg4(wa + 4 * j);
        vb[k] = ly_ldg4(wb + 4 * j);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (i + k * LY_THREADS < nw4) {
          *reinterpret_cast<f32x4*>(was + 4 * (i + k * LY_THREADS)) = va[k];
          *reinterpret_cast<f32x4*>(wbs + 4 * (i + k * LY_THREADS)) = vb[k];
        }
    }
  }
  red[tid] = s4;
  __syncthreads();
  if (sg == 0) {
    for (int k = 1; k < ng; ++k) s4 += red[tid + k * nq];
    g[4 * q] = s4[0] * inv_hw; g[4 * q + 1] = s4[1] * inv_hw; g[4 * q + 2] = s4[2] * inv_hw; g[4 * q + 3] = s4[3] * inv_hw;
  }
  __syncthreads();
  const float* wap = staged ? was : wa;
  const float* wbp = staged ? wbs : wb;
  for (int r = wave; r < R; r += 4) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += wap[r * C + c] * g[c];
    s = ly_group_sum(s, 64);
    if (lane == 0) hid[r] = fmaxf(s, 0.f);
  }
  __syncthreads();
  for (int c = tid; c < C; c += LY_THREADS) {
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += wbp[c * R + r] * hid[r];
    ca[(long)n * C + c] = ly_sigmoid(s);
  }
}

__global__ __launch_bounds__(LY_THREADS) void ly_se_mlp_kernel(const float* __restrict__ part, int slices, int C, float inv_hw,
                                                                const float* __restrict__ wa, const float* __restrict__ wb, int R,
                                                                float* __restrict__ ca) {
  extern __shared__ float sm[];
  ly_se_mlp_body(sm, blockIdx.x, part, slices, C, inv_hw, wa, wb, R, ca);
}

// SE pooling partials alone: part[n][slice][C] = sum of x over the slice's pixels (for ly_rfcbam_mid)
extern "C" int ly_colsum(const void* x, int ldx, int n_img, int HW, int C, float* part, int slices, int dtype, void* stream) {
  LY_CHECK_DTYPE(dtype, "colsum");
  LY_CHECK(x && part && (C & 3) == 0 && (ldx & 3) == 0 && C <= 1024 && slices > 0, "colsum: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  LY_WITH_T(dtype, hipLaunchKernelGGL(ly_colsum_kernel<T>, dim3(n_img * slices), dim3(LY_THREADS), 0, st, reinterpret_cast<const T*>(x), ldx, HW, C, slices, part));
  LY_LAUNCH_CHECK();
  return 0;
}

extern "C" int ly_se_fwd(const void* x, int ldx, int n_img, int HW, int C, const float* wa, const float* wb, int R, float* part,
                         int slices, float* ca, int dtype, void* stream) {
  LY_CHECK_DTYPE(dtype, "se");
  LY_CHECK(x && wa && wb && part && ca, "se: null pointer");
  LY_CHECK((C & 3) == 0 && (ldx & 3) == 0 && C <= 1024 && slices > 0 && R > 0 && R <= 256, "se: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  LY_WITH_T(dtype, hipLaunchKernelGGL(ly_colsum_kernel<T>, dim3(n_img * slices), dim3(LY_THREADS), 0, st, reinterpret_cast<const T*>(x), ldx, HW, C, slices, part));
  LY_LAUNCH_CHECK();
  hipLaunchKernelGGL(ly_se_mlp_kernel, dim3(n_img), dim3(LY_THREADS), sizeof(float) * ly_se_lds_floats(C, R), st, part, slices, C, 1.f / (float)HW,
                     wa, wb, R, ca);
  LY_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// RFCBAM statistics, k = 1:  g = relu(x*a + b);  mm[pix] = (max_c g, mean_c g).  One wave per pixel.
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(LY_THREADS) void ly_rfcbam_stats1_kernel(const T* __restrict__ x, int ldx, long M, int C,
                                                                       const float* __restrict__ a, const float* __restrict__ b,
                                                                       float* __restrict__ mm) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * 4;
  const int nc4 = C >> 2;
  const float inv = 1.f / (float)C;
  for (long p = wave; p < M; p += nwaves) {
    float mx = -FLT_MAX, sm = 0.f;
    for (int c4 = lane; c4 < nc4; c4 += 64) {
      const f32x4 v = ly_ld4<T>(x + p * ldx + 4 * c4), s = ly_ldg4(a + 4 * c4), t = ly_ldg4(b + 4 * c4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float g = fmaxf(v[r] * s[r] + t[r], 0.f);
        mx = fmaxf(mx, g);
        sm += g;
      }
    }
    mx = ly_group_max(mx, 64);                              // (in-row steps as DPP moves: ly_common.hpp)
    sm = ly_group_sum(sm, 64);
    if (lane == 0) {
      mm[2 * p] = mx;
      mm[2 * p + 1] = sm * inv;
    }
  }
}

// k = 1 statistics AND the SE pooling in one pass over x: block = (image, slice of its pixels), wave = one pixel at a time, lane =
// channel quads lane, lane + 64, ...  The per-pixel [max, mean] needs a wave reduction (as above); the per-channel sums of the SE
// global average pool stay in the lanes' registers over the block's pixels and leave as one partial row per block (no atomics,
// no zero fill): part[n][slice][C], summed by the SE step (ly_rfcbam_mid / ly_se_mlp).
#define LY_PRE1_NS 3                 // channel-quad slots per lane: C <= 768
template <typename T>
__global__ __launch_bounds__(LY_THREADS) void ly_rfcbam_pre1_kernel(const T* __restrict__ x, int ldx, int HW, int C, int slices,
                                                                    const float* __restrict__ a, const float* __restrict__ b,
                                                                    float* __restrict__ mm, float* __restrict__ part) {
  __shared__ f32x4 red[3][LY_PRE1_NS * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x / slices, sl = blockIdx.x - n * slices;
  const int per = (HW + slices - 1) / slices;
  const int j_lo = sl * per, j_hi = (j_lo + per) < HW ? (j_lo + per) : HW;
  const int nc4 = C >> 2;
  const float inv = 1.f / (float)C;
  f32x4 sa[LY_PRE1_NS], sb[LY_PRE1_NS], gs[LY_PRE1_NS];
#pragma unroll
  for (int q = 0; q < LY_PRE1_NS; ++q) {
    const int c4 = lane + 64 * q;
    const bool ok = c4 < nc4;
    sa[q] = ok ? ly_ldg4(a + 4 * c4) : ly_zero4();
    sb[q] = ok ? ly_ldg4(b + 4 * c4) : ly_zero4();
    gs[q] = ly_zero4();
  }
  for (int j = j_lo + wave; j < j_hi; j += 4) {
    const long p = (long)n * HW + j;
    float mx = 0.f, sm = 0.f;                  // relu(.) >= 0: zero is the identity of the channel max (absent lanes contribute 0)
#pragma unroll
    for (int q = 0; q < LY_PRE1_NS; ++q) {
      const int c4 = lane + 64 * q;
      if (c4 < nc4) {
        const f32x4 v = ly_ld4<T>(x + p * ldx + 4 * c4);
        gs[q] += v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float g = fmaxf(v[r] * sa[q][r] + sb[q][r], 0.f);
          mx = fmaxf(mx, g);
          sm += g;
        }
      }
    }
    mx = ly_group_max(mx, 64);                              // (in-row steps as DPP moves: ly_common.hpp)
    sm = ly_group_sum(sm, 64);
    if (lane == 0) {
      mm[2 * p] = mx;
      mm[2 * p + 1] = sm * inv;
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int q = 0; q < LY_PRE1_NS; ++q) red[wave - 1][q * 64 + lane] = gs[q];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int q = 0; q < LY_PRE1_NS; ++q) {
      const int c4 = lane + 64 * q;
      if (c4 < nc4) ly_stg4(part + ((long)n * slices + sl) * C + 4 * c4, gs[q] + red[0][q * 64 + lane] + red[1][q * 64 + lane] + red[2][q * 64 + lane]);
    }
  }
}

// The same pass with SEVERAL pixels per wave (round 6).  Above, a wave takes one pixel per trip: one 8-byte load per lane in flight and two
// 64-lane reductions per pixel (two of their six steps go through the LDS pipe) — a chain of dependent latencies per pixel, 16-24 us per launch
// where the bytes need 5-10.  Here LPP = 4 .. 16 lanes share a pixel (64 / LPP pixels per wave and trip), a lane holds VPL 16-byte vectors of
// its pixel — all loads of a trip issued together — and the per-pixel [max, mean] is a reduction over LPP <= 16 lanes: DPP moves only.  The SE
// pooling sums stay per lane over the block's pixels and meet once, in LDS.
template <typename T, int VPL>                  // 16-byte vectors per lane: C / VW <= 16 * VPL (vector ln + 16 v of the pixel; absent ones are masked)
__global__ __launch_bounds__(LY_THREADS) void ly_rfcbam_pre1v_kernel(const T* __restrict__ x, int ldx, int HW, int C, int slices,
                                                                     const float* __restrict__ a, const float* __restrict__ b,
                                                                     float* __restrict__ mm, float* __restrict__ part) {
  constexpr int LPP = 16, VW = LyT<T>::VW, NQ = VW / 4, PPW = 64 / LPP, NPG = 4 * PPW;        // pixels per wave and trip, pixel groups per block
  const int nv = C / VW;
  using RV = typename LyT<T>::RV;
  extern __shared__ f32x4 pre1v_sm[];                  // [NPG][C / 4]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ln = lane % LPP, pg = wave * PPW + lane / LPP;
  const int n = blockIdx.x / slices, sl = blockIdx.x - n * slices;
  const int per = (HW + slices - 1) / slices;
  const int j_lo = sl * per, j_hi = (j_lo + per) < HW ? (j_lo + per) : HW;
  const float inv = 1.f / (float)C;
  f32x4 sa[VPL][NQ], sb[VPL][NQ], gs[VPL][NQ];
#pragma unroll
  for (int v = 0; v < VPL; ++v)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const bool ok = ln + LPP * v < nv;
      const int c = ok ? VW * (ln + LPP * v) + 4 * q : 0;
      sa[v][q] = ok ? ly_ldg4(a + c) : ly_zero4(); sb[v][q] = ok ? ly_ldg4(b + c) : ly_zero4(); gs[v][q] = ly_zero4();      // absent vectors: relu(0 * x + 0) = 0
    }
  for (int j0 = j_lo; j0 < j_hi; j0 += NPG) {
    const int j = j0 + pg;
    const bool live = j < j_hi;
    const long p = (long)n * HW + (live ? j : j_hi - 1);
    RV xv[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) xv[v] = ly_ldrv<T>(x + p * ldx + (ln + LPP * v < nv ? VW * (ln + LPP * v) : 0));      // (clamped: straight-line loads)
    float mx = 0.f, sm = 0.f;                  // relu(.) >= 0: zero is the identity of the channel max
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      f32x4 xq[NQ];
      ly_rv_unpack(xv[v], xq);
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        if (live) gs[v][q] += xq[q];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float g = fmaxf(xq[q][r] * sa[v][q][r] + sb[v][q][r], 0.f);
          mx = fmaxf(mx, g);
          sm += g;
        }
      }
    }
    mx = ly_group_max(mx, LPP);
    sm = ly_group_sum(sm, LPP);
    if (ln == 0 && live) *reinterpret_cast<f32x2*>(mm + 2 * p) = (f32x2){mx, sm * inv};
  }
  const int nc4 = C >> 2;
#pragma unroll
  for (int v = 0; v < VPL; ++v)
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      if (ln + LPP * v < nv) pre1v_sm[pg * nc4 + (VW * (ln + LPP * v) >> 2) + q] = gs[v][q];
  __syncthreads();
  for (int c4 = threadIdx.x; c4 < nc4; c4 += LY_THREADS) {
    f32x4 t = pre1v_sm[c4];
    for (int g = 1; g < NPG; ++g) t += pre1v_sm[g * nc4 + c4];
    ly_stg4(part + ((long)n * slices + sl) * C + 4 * c4, t);
  }
}

template <typename T>
static bool pre1v_launch(const T* x, int ldx, int n_img, int HW, int C, int slices, const float* a1, const float* b1, float* mm, float* part, hipStream_t st) {
  constexpr int VW = LyT<T>::VW;
  if (C % VW || ldx % VW) return false;
  const int nv = C / VW, vpl = (nv + 15) / 16;
  const size_t lds = (size_t)16 * C * sizeof(float);            // 16 pixel groups per block
#define LY_PV(V_) case V_: hipLaunchKernelGGL((ly_rfcbam_pre1v_kernel<T, V_>), dim3((unsigned)(n_img * slices)), dim3(LY_THREADS), lds, st, x, ldx, HW, C, slices, a1, b1, mm, part); return true
  switch (vpl) { LY_PV(1); LY_PV(2); LY_PV(3); LY_PV(4); default: break; }
#undef LY_PV
  return false;
}

// ---------------------------------------------------------------------------------------------------
// RFCBAM statistics, k = 3 (stride s, pad 1).  lane = output pixel of a TH x TW tile, the 4 waves
// split the channels (channel c0 + wave + 4j of each 32-channel chunk); depthwise weights are wave-
// uniform scalar loads from wq[C32/32][4 waves][9 t][8 j][10] = 9 folded weights W'[t][u] + folded
// bias b'[t] per (t, channel), zero padded to a multiple of 32 channels (pack.rfcbam_stats_weights).
// ---------------------------------------------------------------------------------------------------
#define LY_SCC 32
#define LY_ST3_NV 10              // float4 staging items per thread per chunk: IH*IW*8 <= 10*256
#define LY_ST3_WF (9 * 4 * 20)    // floats of folded weights per wave per chunk: [9 t][4 channel pairs][9 x (w_a, w_b) + (b_a, b_b)]
template <typename T>
__global__ __launch_bounds__(LY_THREADS) void ly_rfcbam_stats3_kernel(const T* __restrict__ x, int ldx, int H, int W, int C,
                                                                       int Ho, int Wo, int s, int TH, int TW, int nct, int nrt,
                                                                       const float* __restrict__ wg, float* __restrict__ mm, float* __restrict__ part) {
  extern __shared__ float lds[];
  const int IH = s * (TH - 1) + 3, IW = s * (TW - 1) + 3;
  float* wsm = lds;                                  // [4 waves][LY_ST3_WF]: this chunk's folded weights (16-B aligned)
  float* xs = lds + 4 * LY_ST3_WF;                   // [IH*IW][LY_SCC + 1]
  float* red = xs + IH * IW * (LY_SCC + 1);          // [4][18][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int b = blockIdx.x;
  const int ct = b % nct; b /= nct;
  const int rt = b % nrt;
  const int n = b / nrt;
  const int oy0 = rt * TH, ox0 = ct * TW;
  const int ly = lane / TW, lx = lane - ly * TW;
  const int oy = oy0 + ly, ox = ox0 + lx;
  const bool active = ly < TH && oy < Ho && ox < Wo;
  const int iy0 = s * oy0 - 1, ix0 = s * ox0 - 1;

  // staging plan of this thread (independent of the channel chunk): global element offset (or -1) and LDS slot
  const int items = IH * IW * (LY_SCC / 4);
  long soff[LY_ST3_NV];
  int doff[LY_ST3_NV];
#pragma unroll
  for (int e = 0; e < LY_ST3_NV; ++e) {
    const int idx = tid + e * LY_THREADS;
    long so = -1;
    int dd = -1;
    if (idx < items) {
      const int ip = idx >> 3, c4 = idx & 7;
      const int r = ip / IW, q = ip - r * IW;
      const int iy = iy0 + r, ix = ix0 + q;
      dd = ip * (LY_SCC + 1) + 4 * c4;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) so = (((long)n * H + iy) * W + ix) * ldx + 4 * c4;
    }
    soff[e] = so; doff[e] = dd;
  }
  // x addresses of the 9 taps of this lane's pixel (LDS float index, channel 0)
  int xa[9];
#pragma unroll
  for (int u = 0; u < 9; ++u) xa[u] = ((s * ly + u / 3) * IW + (s * lx + u % 3)) * (LY_SCC + 1);

  float mx[9], sm[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) { mx[t] = 0.f; sm[t] = 0.f; }      // G = relu(.) >= 0, so 0 is the identity of the channel max

  // the input tile (pv) and the folded weights (wv) of chunk c+1 are requested before chunk c is regenerated: no load is waited
  // for right after it is issued
  typename LyT<T>::R4 pv[LY_ST3_NV];
  auto prefetch = [&](int c0) {
#pragma unroll
    for (int e = 0; e < LY_ST3_NV; ++e) {
      const bool ok = soff[e] >= 0 && c0 + 4 * ((tid + e * LY_THREADS) & 7) < C;
      pv[e] = ly_ldr4<T>(ok ? x + soff[e] + c0 : x);
    }
  };
  constexpr int WV = (4 * LY_ST3_WF / 4 + LY_THREADS - 1) / LY_THREADS;
  f32x4 wv[WV];
  auto wprefetch = [&](int chunk) {
    const float* wsrc = wg + (long)chunk * (4 * LY_ST3_WF);
#pragma unroll
    for (int e = 0; e < WV; ++e) {
      const int i = tid + e * LY_THREADS;
      wv[e] = ly_ldg4(wsrc + 4 * (i < 4 * LY_ST3_WF / 4 ? i : 0));
    }
  };
  wprefetch(0);
  prefetch(0);

  for (int c0 = 0; c0 < C; c0 += LY_SCC) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < WV; ++e) {
      const int i = tid + e * LY_THREADS;
      if (i < 4 * LY_ST3_WF / 4) reinterpret_cast<f32x4*>(wsm)[i] = wv[e];
    }
    const bool more = c0 + LY_SCC < C;
    wprefetch(more ? c0 / LY_SCC + 1 : 0);
#pragma unroll
    for (int e = 0; e < LY_ST3_NV; ++e) {
      if (doff[e] >= 0) {
        const bool ok = soff[e] >= 0 && c0 + 4 * ((tid + e * LY_THREADS) & 7) < C;
        const f32x4 pf = ly_r4_f32(pv[e]);
        float* d = xs + doff[e];
        d[0] = ok ? pf[0] : 0.f; d[1] = ok ? pf[1] : 0.f; d[2] = ok ? pf[2] : 0.f; d[3] = ok ? pf[3] : 0.f;
      }
    }
    prefetch(more ? c0 + LY_SCC : 0);      // unconditional: a load under a run-time branch makes every later wait conservative
    __syncthreads();
    if (part) {
      // SE global-average-pool partial of this tile (models/rfa.py:90) from the staged chunk: the block OWNS input rows
      // [s*oy0, s*(oy0+TH)) x cols [s*ox0, s*(ox0+TW)) (every input pixel belongs to exactly one tile), thread = (channel, 1 of 8
      // pixel stripes); the stripes meet in LDS (red is free until the end) and leave as one row of part[n][tile][C]
      const int ch = tid & 31, stripe = tid >> 5;
      const int rows_own = min(s * TH, H - s * oy0), cols_own = min(s * TW, W - s * ox0);
      float acc = 0.f;
      for (int q = stripe; q < rows_own * cols_own; q += 8) {
        const int r = q / cols_own, cq = q - r * cols_own;
        acc += xs[((r + 1) * IW + cq + 1) * (LY_SCC + 1) + ch];
      }
      red[stripe * 32 + ch] = acc;
      __syncthreads();
      if (tid < 32 && c0 + tid < C) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += red[g * 32 + tid];
        part[((long)n * (nrt * nct) + rt * nct + ct) * C + c0 + tid] = t;
      }
      __syncthreads();
    }
    {
      // inputs of this wave's 8 channels (c0 + wave + 4j) as 4 packed pairs (j = 2p, 2p+1), then the folded
      // weights [t][p][9 x (w_a, w_b), (b_a, b_b)] read as wave-uniform LDS broadcasts: v_pk_fma_f32 does two
      // channels per instruction; two pairs at a time so that two independent accumulation chains interleave.
      f32x2 xv[4][9];
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int u = 0; u < 9; ++u) {
          const float a = xs[xa[u] + wave + 8 * p], c = xs[xa[u] + wave + 8 * p + 4];
          xv[p][u] = active ? (f32x2){a, c} : (f32x2){0.f, 0.f};
        }
      const f32x4* wq = reinterpret_cast<const f32x4*>(wsm + wave * LY_ST3_WF);
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
          f32x4 q0[5], q1[5];
#pragma unroll
          for (int i = 0; i < 5; ++i) {
            q0[i] = wq[(t * 4 + 2 * ph) * 5 + i];
            q1[i] = wq[(t * 4 + 2 * ph + 1) * 5 + i];
          }
          f32x2 a0 = {q0[4][2], q0[4][3]}, a1 = {q1[4][2], q1[4][3]};
#pragma unroll
          for (int u = 0; u < 9; ++u) {
            a0 += xv[2 * ph][u] * (f32x2){q0[u >> 1][2 * (u & 1)], q0[u >> 1][2 * (u & 1) + 1]};
            a1 += xv[2 * ph + 1][u] * (f32x2){q1[u >> 1][2 * (u & 1)], q1[u >> 1][2 * (u & 1) + 1]};
          }
          const float g0 = fmaxf(a0[0], 0.f), g1 = fmaxf(a0[1], 0.f), g2 = fmaxf(a1[0], 0.f), g3 = fmaxf(a1[1], 0.f);
          mx[t] = fmaxf(fmaxf(mx[t], fmaxf(g0, g1)), fmaxf(g2, g3));
          sm[t] += (g0 + g1) + (g2 + g3);
        }
    }
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    red[(wave * 18 + t) * 64 + lane] = mx[t];
    red[(wave * 18 + 9 + t) * 64 + lane] = sm[t];
  }
  __syncthreads();
  if (wave == 0 && active) {
    const float inv = 1.f / (float)C;
    const int HK = 3 * Ho, WK = 3 * Wo;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float m = red[t * 64 + lane], a = red[(9 + t) * 64 + lane];
      for (int w2 = 1; w2 < 4; ++w2) {
        m = fmaxf(m, red[(w2 * 18 + t) * 64 + lane]);
        a += red[(w2 * 18 + 9 + t) * 64 + lane];
      }
      const long o = (((long)n * HK + 3 * oy + t / 3) * WK + 3 * ox + t % 3) * 2;
      mm[o] = m;
      mm[o + 1] = a * inv;
    }
  }
}

extern "C" int ly_rfcbam_stats(const void* x, int ldx, int n_img, int H, int W, int C, int k, int s, const float* wg,
                               const float* a1, const float* b1, int TH, int TW, float* mm, float* part, int slices, int dtype, void* stream) {
  LY_CHECK_DTYPE(dtype, "rfcbam_stats");
  LY_CHECK(x && mm && (C & 3) == 0 && (ldx & 3) == 0, "rfcbam_stats: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (k == 1) {
    LY_CHECK(s == 1 && a1 && b1, "rfcbam_stats: k=1 needs stride 1 and folded scale/shift");
    long M = (long)n_img * H * W;
    if (part) {           // statistics + SE pooling partials in one pass
      LY_CHECK(slices > 0 && C <= 256 * LY_PRE1_NS, "rfcbam_stats: fused SE pooling needs slices > 0 and C <= %d", 256 * LY_PRE1_NS);
      bool done = false;
      LY_WITH_T(dtype, done = pre1v_launch<T>(reinterpret_cast<const T*>(x), ldx, n_img, H * W, C, slices, a1, b1, mm, part, st));      // several pixels per wave
      if (!done)
        LY_WITH_T(dtype, hipLaunchKernelGGL(ly_rfcbam_pre1_kernel<T>, dim3((unsigned)(n_img * slices)), dim3(LY_THREADS), 0, st, reinterpret_cast<const T*>(x), ldx,
                                            H * W, C, slices, a1, b1, mm, part));
      LY_LAUNCH_CHECK();
      return 0;
    }
    long blocks = (M + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    LY_WITH_T(dtype, hipLaunchKernelGGL(ly_rfcbam_stats1_kernel<T>, dim3((unsigned)blocks), dim3(LY_THREADS), 0, st, reinterpret_cast<const T*>(x), ldx, M, C, a1, b1, mm));
    LY_LAUNCH_CHECK();
    return 0;
  }
  LY_CHECK(k == 3 && s >= 1 && wg, "rfcbam_stats: only k in {1,3} is built");
  LY_CHECK(TW >= 1 && TH >= 1 && TH * TW <= 64, "rfcbam_stats: tile %dx%d does not fit a wave", TH, TW);
  const int Ho = (H + 2 - 3) / s + 1, Wo = (W + 2 - 3) / s + 1;
  const int nct = (Wo + TW - 1) / TW, nrt = (Ho + TH - 1) / TH;
  LY_CHECK(!part || slices == nct * nrt, "rfcbam_stats: k=3 fused SE pooling writes one partial row per tile: slices must be %d", nct * nrt);
  const int IH = s * (TH - 1) + 3, IW = s * (TW - 1) + 3;
  size_t lds = sizeof(float) * ((size_t)4 * LY_ST3_WF + (size_t)IH * IW * (LY_SCC + 1) + 4 * 18 * 64);
  LY_CHECK(IH * IW * (LY_SCC / 4) <= LY_ST3_NV * LY_THREADS, "rfcbam_stats: input tile %dx%d exceeds the staging capacity", IH, IW);
  LY_CHECK(lds <= 160 * 1024, "rfcbam_stats: tile needs %zu B LDS", lds);
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ly_rfcbam_stats3_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    LY_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(ly_rfcbam_stats3_kernel<__bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    LY_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    configured = true;
  }
  LY_WITH_T(dtype, hipLaunchKernelGGL(ly_rfcbam_stats3_kernel<T>, dim3(n_img * nrt * nct), dim3(LY_THREADS), lds, st, reinterpret_cast<const T*>(x), ldx, H, W, C,
                                      Ho, Wo, s, TH, TW, nct, nrt, wg, mm, part));
  LY_LAUNCH_CHECK();
  return 0;
}

// rfa[n, y, x] = sigmoid( sum_{ch, dy, dx} w[ch][dy][dx] * mm[n, y+dy-1, x+dx-1, ch] )
// A thread owns FOUR horizontally adjacent outputs: 3 rows x 6 columns of [max, mean] pairs = 18 eight-byte loads (clamped addresses, all
// issued before the first use) for 72 multiply-adds — one output per thread was 18 guarded loads for 18 (12.6 us per launch at 120 x 120 x 64,
// all of it L2 round trips).  Items: i -> (n, y, x0 = 4 * (i % ceil(WK / 4))).
__device__ __forceinline__ void ly_rfa_map_body(const long i, const float* __restrict__ mm, int HK, int WK, long total4,
                                                const float* __restrict__ w, float* __restrict__ rfa) {
  if (i >= total4) return;
  const int wq = (WK + 3) >> 2;
  const int x0 = 4 * (int)(i % wq);
  const int yk = (int)((i / wq) % HK);
  const long n = i / ((long)wq * HK);
  f32x2 v[3][6];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int yy = yk + dy - 1;
    const int yc = yy < 0 ? 0 : (yy >= HK ? HK - 1 : yy);
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const int xx = x0 + c - 1;
      const int xc = xx < 0 ? 0 : (xx >= WK ? WK - 1 : xx);
      v[dy][c] = *reinterpret_cast<const f32x2*>(mm + ((n * HK + yc) * WK + xc) * 2);
    }
  }
  float wv[18];
#pragma unroll
  for (int k = 0; k < 18; ++k) wv[k] = w[k];
  float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int yy = yk + dy - 1;
    const bool yok = yy >= 0 && yy < HK;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const int xx = x0 + c - 1;
      const bool ok = yok && xx >= 0 && xx < WK;
      const f32x2 p = ok ? v[dy][c] : (f32x2){0.f, 0.f};
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int dx = c - o;                            // output x0 + o reads column x0 + o + dx - 1 = x0 + c - 1
        if (dx >= 0 && dx < 3) s[o] += wv[dy * 3 + dx] * p[0] + wv[9 + dy * 3 + dx] * p[1];
      }
    }
  }
  float* out = rfa + (n * HK + yk) * WK + x0;
  if (x0 + 3 < WK && (WK & 3) == 0) {
    *reinterpret_cast<f32x4*>(out) = (f32x4){ly_sigmoid(s[0]), ly_sigmoid(s[1]), ly_sigmoid(s[2]), ly_sigmoid(s[3])};
  } else {
#pragma unroll
    for (int o = 0; o < 4; ++o)
      if (x0 + o < WK) out[o] = ly_sigmoid(s[o]);
  }
}

__global__ __launch_bounds__(LY_THREADS) void ly_rfa_map_kernel(const float* __restrict__ mm, int HK, int WK, long total4,
                                                                 const float* __restrict__ w, float* __restrict__ rfa) {
  ly_rfa_map_body((long)blockIdx.x * LY_THREADS + threadIdx.x, mm, HK, WK, total4, w, rfa);
}

// The two small dependent steps between the statistics pass and the main contraction of RFCBAMConv as ONE launch: blocks
// [0, n_img) finish SE (slice partials -> mean -> fc -> sigmoid, models/rfa.py:88-92), the remaining blocks evaluate
// get_weight (3x3 conv on the [max, mean] map + sigmoid, models/rfa.py:107,127).  They are independent of each other.
__global__ __launch_bounds__(LY_THREADS) void ly_rfcbam_mid_kernel(const float* __restrict__ part, int slices, int C, float inv_hw,
                                                                   const float* __restrict__ wa, const float* __restrict__ wb, int R,
                                                                   float* __restrict__ ca, int n_img, const float* __restrict__ mm, int HK, int WK,
                                                                   long total4, const float* __restrict__ w18, float* __restrict__ rfa) {
  extern __shared__ float sm[];
  if ((int)blockIdx.x < n_img) ly_se_mlp_body(sm, blockIdx.x, part, slices, C, inv_hw, wa, wb, R, ca);
  else ly_rfa_map_body((long)(blockIdx.x - n_img) * LY_THREADS + threadIdx.x, mm, HK, WK, total4, w18, rfa);
}

extern "C" int ly_rfcbam_mid(const float* part, int slices, int C, int HW, const float* wa, const float* wb, int R, float* ca, int n_img,
                             const float* mm, int HK, int WK, const float* w18, float* rfa, void* stream) {
  LY_CHECK(part && wa && wb && ca && mm && w18 && rfa, "rfcbam_mid: null pointer");
  LY_CHECK((C & 3) == 0 && C <= 1024 && slices > 0 && R > 0 && R <= 256 && n_img > 0, "rfcbam_mid: bad arguments");
  const long total4 = (long)n_img * HK * ((WK + 3) / 4);
  const unsigned blocks = (unsigned)(n_img + (total4 + LY_THREADS - 1) / LY_THREADS);
  hipLaunchKernelGGL(ly_rfcbam_mid_kernel, dim3(blocks), dim3(LY_THREADS), sizeof(float) * ly_se_lds_floats(C, R),
                     reinterpret_cast<hipStream_t>(stream), part, slices, C, 1.f / (float)HW, wa, wb, R, ca, n_img, mm, HK, WK, total4, w18, rfa);
  LY_LAUNCH_CHECK();
  return 0;
}

extern "C" int ly_rfa_map(const float* mm, int n_img, int HK, int WK, const float* w, float* rfa, void* stream) {
  LY_CHECK(mm && w && rfa, "rfa_map: null pointer");
  const long total4 = (long)n_img * HK * ((WK + 3) / 4);
  hipLaunchKernelGGL(ly_rfa_map_kernel, dim3((unsigned)((total4 + LY_THREADS - 1) / LY_THREADS)), dim3(LY_THREADS), 0,
                     reinterpret_cast<hipStream_t>(stream), mm, HK, WK, total4, w, rfa);
  LY_LAUNCH_CHECK();
  return 0;
}

// standalone CoordAtt gating: out = x * a_w[n,w,:] * a_h[n,h,:] (+ res)   (models/common.py:1608, 1623)
// One block per (image, band of 8 rows, slab of columns); a thread owns a 16-byte channel vector of up to 8 columns: their a_w factors are
// loaded once, a row's a_h factor once per row, and each row's x (+ res) vectors are all requested before the first use — per 16 bytes of the
// map one load and one store.  (The first version — thread = (pixel, 4 channels), three 64-bit divisions and two fp32 table loads per 8 bytes
// of x — ran at 2.4 TB/s.)
#define LY_GT_RB 8
#define LY_GT_MAXW 8
template <typename T>
__global__ __launch_bounds__(LY_THREADS) void ly_gate_kernel(const T* __restrict__ x, int ldx, int H, int W, int C,
                                                              const float* __restrict__ a_h, const float* __restrict__ a_w,
                                                              const T* __restrict__ res, int ldres, T* __restrict__ out, int ldo, int bands, int slabs, int cpt) {
  constexpr int VW = LyT<T>::VW, NQ = VW / 4;
  using RV = typename LyT<T>::RV;
  const int ncv = C / VW, tid = threadIdx.x;
  const int groups = LY_THREADS / ncv;
  const int cv = tid % ncv, g0 = tid / ncv;
  if (g0 >= groups) return;
  const int slab = blockIdx.x % slabs;
  const int bb = blockIdx.x / slabs;
  const long n = bb / bands;
  const int band = bb - (int)n * bands;
  const int w0 = slab * groups * cpt;
  const int h_lo = band * LY_GT_RB, h_hi = h_lo + LY_GT_RB < H ? h_lo + LY_GT_RB : H;
  const int c = VW * cv;
  f32x4 aw[LY_GT_MAXW][NQ];
  bool okw[LY_GT_MAXW];
  int wcl[LY_GT_MAXW];
#pragma unroll
  for (int i = 0; i < LY_GT_MAXW; ++i) {
    const int w = w0 + g0 + i * groups;
    okw[i] = i < cpt && w < W;
    wcl[i] = okw[i] ? w : (w0 < W ? w0 : 0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) aw[i][q] = ly_ldg4(a_w + (n * W + wcl[i]) * C + c + 4 * q);
  }
  for (int h = h_lo; h < h_hi; ++h) {
    const long nh = n * H + h;
    f32x4 ah[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) ah[q] = ly_ldg4(a_h + nh * C + c + 4 * q);
    RV xr[LY_GT_MAXW], rr[LY_GT_MAXW];
#pragma unroll
    for (int i = 0; i < LY_GT_MAXW; ++i) {
      if (i >= cpt) break;                                                     // (uniform: columns per thread of this launch)
      const long row = nh * W + wcl[i];
      xr[i] = ly_ldrv<T>(x + row * ldx + c);
      if (res) rr[i] = ly_ldrv<T>(res + row * ldres + c);
    }
#pragma unroll
    for (int i = 0; i < LY_GT_MAXW; ++i) {
      if (i >= cpt) break;
      f32x4 v[NQ], r[NQ];
      ly_rv_unpack(xr[i], v);
      if (res) ly_rv_unpack(rr[i], r);
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        v[q] = v[q] * aw[i][q] * ah[q];
        if (res) v[q] += r[q];
      }
      if (okw[i]) *reinterpret_cast<RV*>(out + (nh * W + wcl[i]) * ldo + c) = ly_rv_pack(v, (RV*)nullptr);
    }
  }
}

extern "C" int ly_coordatt_gate(const void* x, int ldx, int n_img, int H, int W, int C, const float* a_h, const float* a_w,
                                const void* res, int ldres, void* out, int ldo, int dtype, void* stream) {
  LY_CHECK_DTYPE(dtype, "gate");
  const int vw = dtype == LY_BF16 ? 8 : 4;
  LY_CHECK(x && a_h && a_w && out && (C % vw) == 0 && (ldx % vw) == 0 && (ldo % vw) == 0 && (ldres % vw) == 0 && C / vw <= LY_THREADS,
           "gate: C = %d and the row strides must be multiples of %d (16-byte channel vectors)", C, vw);
  const int groups = LY_THREADS / (C / vw);
  int cpt = (W + groups - 1) / groups;
  if (cpt > LY_GT_MAXW) cpt = LY_GT_MAXW;
  const int slabs = (W + groups * cpt - 1) / (groups * cpt);
  const int bands = (H + LY_GT_RB - 1) / LY_GT_RB;
  LY_WITH_T(dtype, hipLaunchKernelGGL(ly_gate_kernel<T>, dim3((unsigned)((long)n_img * bands * slabs)), dim3(LY_THREADS), 0, reinterpret_cast<hipStream_t>(stream),
                                      reinterpret_cast<const T*>(x), ldx, H, W, C, a_h, a_w, reinterpret_cast<const T*>(res), ldres, reinterpret_cast<T*>(out), ldo,
                                      bands, slabs, cpt));
  LY_LAUNCH_CHECK();
  return 0;
}


// ---------------------------------------------------------------------------------------------------
// SPPF pooling: out[n, p, 0:C] = x, [C:2C] = m5(x), [2C:3C] = m5(m5(x)), [3C:4C] = m5(m5(m5(x)))
// (k x k, stride 1, pad k//2, -inf padding; reference models/common.py:348-366).  Chained k-max pools
// equal single (2k-1)- and (3k-2)-wide max windows, computed separably from one LDS copy of the map.
// One block per (image, 16-channel group); the map (H*W <= 4096) lives in LDS.
// ---------------------------------------------------------------------------------------------------
#define LY_SP_CG 4     // channels per block: n_img * C/4 blocks keep the whole chip busy on the small P5 map
template <typename T>
__global__ __launch_bounds__(LY_THREADS) void ly_sppf_pool_kernel(const T* __restrict__ x, int ldx, int H, int W, int C, int k,
                                                                  T* __restrict__ out, int ldo) {
  extern __shared__ f32x4 sp4[];               // a[HW], b[HW] as float4 (4 channels per position)
  const int HW = H * W;
  f32x4* a = sp4;
  f32x4* b = sp4 + HW;
  const int groups = C / LY_SP_CG;
  const int bid = ly_xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int n = bid / groups, g = bid - n * groups;
  const int c0 = g * LY_SP_CG;
  const int tid = threadIdx.x, r = k / 2;
  const f32x4 ninf = (f32x4){-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int p = tid; p < HW; p += LY_THREADS) {
    const f32x4 v = ly_ld4<T>(x + ((long)n * HW + p) * ldx + c0);
    a[p] = v;
    ly_st4<T>(out + ((long)n * HW + p) * ldo + c0, v);
  }
  __syncthreads();
  for (int level = 1; level <= 3; ++level) {
    // b = row-max of a, then a = col-max of b  (one k x k max pool of the previous level)
    for (int p = tid; p < HW; p += LY_THREADS) {
      const int h = p / W, w = p - h * W;
      f32x4 m = ninf;
      for (int d = -r; d <= r; ++d) {
        const int ww = w + d;
        if (ww >= 0 && ww < W) {
          const f32x4 v = a[h * W + ww];
#pragma unroll
          for (int q = 0; q < 4; ++q) m[q] = fmaxf(m[q], v[q]);
        }
      }
      b[p] = m;
    }
    __syncthreads();
    for (int p = tid; p < HW; p += LY_THREADS) {
      const int h = p / W, w = p - h * W;
      f32x4 m = ninf;
      for (int d = -r; d <= r; ++d) {
        const int hh = h + d;
        if (hh >= 0 && hh < H) {
          const f32x4 v = b[hh * W + w];
#pragma unroll
          for (int q = 0; q < 4; ++q) m[q] = fmaxf(m[q], v[q]);
        }
      }
      a[p] = m;
      ly_st4<T>(out + ((long)n * HW + p) * ldo + level * C + c0, m);
    }
    __syncthreads();
  }
}

extern "C" int ly_sppf_pool(const void* x, int ldx, int n_img, int H, int W, int C, int k, void* out, int ldo, int dtype, void* stream) {
  LY_CHECK_DTYPE(dtype, "sppf_pool");
  LY_CHECK(x && out && k >= 1 && (k & 1) && (C & 3) == 0 && (ldx & 3) == 0 && (ldo & 3) == 0, "sppf_pool: bad arguments");
  size_t lds = sizeof(float) * 2 * (size_t)H * W * 4;
  LY_CHECK(lds <= 160 * 1024, "sppf_pool: %dx%d map does not fit LDS (%zu B)", H, W, lds);
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ly_sppf_pool_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    LY_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(ly_sppf_pool_kernel<__bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    LY_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    configured = true;
  }
  LY_WITH_T(dtype, hipLaunchKernelGGL(ly_sppf_pool_kernel<T>, dim3(n_img * (C / LY_SP_CG)), dim3(LY_THREADS), lds, reinterpret_cast<hipStream_t>(stream),
                                      reinterpret_cast<const T*>(x), ldx, H, W, C, k, reinterpret_cast<T*>(out), ldo));
  LY_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// Detect tail (reference models/yolo.py:95-120): from the head GEMM output y[n, h, w, na*no] (row stride
// ldy) write the raw map p[n, a, h, w, o] and, in eval mode, the decoded rows z[n, zoff + (a*H + h)*W + w, o]:
//   xy = (2*sig - 0.5 + grid) * stride,  wh = (2*sig)^2 * anchor*stride,  rest = sig.
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(LY_THREADS) void ly_detect_tail_kernel(const T* __restrict__ y, int ldy, long total, int H, int W, int na, int no,
                                                                    const float* __restrict__ anchors /* [na,2] grid units */, float stride,
                                                                    float* __restrict__ p, float* __restrict__ z, long zrows, long zoff) {
  const long i = (long)blockIdx.x * LY_THREADS + threadIdx.x;      // over n*na*H*W*no, output order
  if (i >= total) return;
  const int o = (int)(i % no);
  long t = i / no;
  const int w = (int)(t % W); t /= W;
  const int h = (int)(t % H); t /= H;
  const int a = (int)(t % na);
  const long n = t / na;
  const float v = ly_ld1<T>(y + ((n * H + h) * W + w) * ldy + a * no + o);
  p[i] = v;
  if (z) {
    const float s = ly_sigmoid(v);
    float r = s;
    if (o == 0) r = (s * 2.f + ((float)w - 0.5f)) * stride;
    else if (o == 1) r = (s * 2.f + ((float)h - 0.5f)) * stride;
    else if (o == 2 || o == 3) { const float q = s * 2.f; r = q * q * (anchors[a * 2 + (o - 2)] * stride); }
    z[(n * zrows + zoff + ((long)a * H + h) * W + w) * no + o] = r;
  }
}

extern "C" int ly_detect_tail(const void* y, int ldy, int n_img, int H, int W, int na, int no, const float* anchors, float stride,
                              float* p, float* z, long zrows, long zoff, int dtype, void* stream) {
  LY_CHECK_DTYPE(dtype, "detect_tail");
  LY_CHECK(y && p && anchors, "detect_tail: null pointer");
  const long total = (long)n_img * na * H * W * no;
  LY_WITH_T(dtype, hipLaunchKernelGGL(ly_detect_tail_kernel<T>, dim3((unsigned)((total + LY_THREADS - 1) / LY_THREADS)), dim3(LY_THREADS), 0,
                                      reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const T*>(y), ldy, total, H, W, na, no, anchors, stride, p, z, zrows, zoff));
  LY_LAUNCH_CHECK();
  return 0;
}

// Adjoint of the raw-map permute above for the training step (models/yolo.py:88 `x[i].view(bs, na, no, ny, nx).permute(0, 1, 3, 4, 2)`):
// dp fp32 [n, na, H, W, no] -> du rows [n*H*W][ldu] of T (columns a*no + o, columns >= na*no zero: the operand of the head's dgrad /
// wgrad contractions), and dbias[a*no + o] += sum over pixels.  One launch instead of autograd's cast + two layout copies + zero-padded
// copy + a reduction.  A block walks (image, row) pairs: the na runs of W*no contiguous floats are staged in LDS, written back as
// whole rows, and the column sums are kept in registers until one atomic per column and block.
#define LY_DH_MAXW 160
#define LY_DH_MAXLD 32
#define LY_DH_TILE 12288                                   // floats of the LDS tile (48 KB): RB image rows of W pixels x (ldu + 1); >= MAXW * (MAXLD + 1)
// RB image rows per trip (as many as fit the tile, at most 8): one trip's loads are all in flight together — with one row per trip a block
// of the 80 x 80 level paid a dependent global -> LDS -> global round trip and two barriers per 5.6 KB — and the rows leave as 8-byte
// vectors (were 2-byte element stores).
template <typename T>
__global__ __launch_bounds__(LY_THREADS) void ly_detect_head_bwd_kernel(const float* __restrict__ dp, int n_img, int H, int W, int na, int no,
                                                                        T* __restrict__ du, int ldu, float* __restrict__ dbias, const int f64, const int RB) {
  __shared__ float tile[LY_DH_TILE];
  __shared__ float red[LY_THREADS];
  const int tid = threadIdx.x, co = na * no, LT = ldu + 1;
  const int col = tid & 31, part = tid >> 5;              // column sums: 8 row classes x 32 columns
  float bsum = 0.f;
  for (int i = tid; i < RB * W * LT; i += LY_THREADS) tile[i] = 0.f;      // pad columns stay zero
  const long rows_total = (long)n_img * H;
  const int run = W * no;
  const int q4 = ldu >> 2;                                // 4-element groups of an output row (ldu % 4 == 0: checked by the launcher)
  for (long r0 = (long)blockIdx.x * RB; r0 < rows_total; r0 += (long)gridDim.x * RB) {
    const int nr = rows_total - r0 < RB ? (int)(rows_total - r0) : RB;
    __syncthreads();
    // one CELL (image row, anchor, pixel: `no` contiguous floats) per thread and step: its index arithmetic is done once, not per value (the
    // per-value form spent ~30 instructions of divisions on each 4-byte load: 43 us for the 80 x 80 level's 49 MB); no even: 8-byte loads
    const int cells = nr * na * W;
    for (int e = tid; e < cells; e += LY_THREADS) {
      const int rr = e / (na * W), e1 = e - rr * (na * W);
      const int a = e1 / W, w = e1 - a * W;
      const long r = r0 + rr;
      const long n = r / H;
      const int h = (int)(r - n * H);
      const float* const src = dp + ((((n * na + a) * H + h) * (long)W) + w) * no;
      float* const dst = tile + (rr * W + w) * LT + a * no;
      if ((no & 1) == 0) {
        for (int o = 0; o < no; o += 2) {
          const float2 v = *reinterpret_cast<const float2*>(src + o);
          dst[o] = v.x;
          dst[o + 1] = v.y;
        }
      } else {
        for (int o = 0; o < no; ++o) dst[o] = src[o];
      }
    }
    __syncthreads();
    T* const out = du + r0 * (long)W * ldu;               // the nr rows are contiguous in du
    for (int e = tid; e < nr * W * q4; e += LY_THREADS) {
      const int px = e / q4, g = e - px * q4;
      const float* t = tile + px * LT + 4 * g;
      ly_st4<T>(out + (long)px * ldu + 4 * g, (f32x4){t[0], t[1], t[2], t[3]});
    }
    if (col < co)
      for (int px = part; px < nr * W; px += LY_THREADS / 32) bsum += tile[px * LT + col];
  }
  red[tid] = bsum;
  __syncthreads();
  if (tid < 32 && tid < co) {
    float s = 0.f;
    for (int g = 0; g < LY_THREADS / 32; ++g) s += red[g * 32 + tid];
    ly_gacc(dbias, tid, s, f64);
  }
}

extern "C" int ly_detect_head_bwd(const float* dp, int n_img, int H, int W, int na, int no, void* du, int ldu, float* dbias, int dbias_f64, int dtype,
                                  void* stream) {
  LY_CHECK_DTYPE(dtype, "detect_head_bwd");
  LY_CHECK(dp && du && dbias && n_img > 0 && H > 0 && W > 0 && na > 0 && no > 0 && ((uintptr_t)dp & 7) == 0, "detect_head_bwd: bad arguments");
  LY_CHECK(W <= LY_DH_MAXW && ldu <= LY_DH_MAXLD && na * no <= ldu && (ldu & 3) == 0 && ((uintptr_t)du & 15) == 0,
           "detect_head_bwd: W=%d (max %d) / ldu=%d (max %d, >= na*no=%d, a multiple of 4) out of range", W, LY_DH_MAXW, ldu, LY_DH_MAXLD, na * no);
  int RB = LY_DH_TILE / (W * (ldu + 1));
  if (RB > 8) RB = 8;
  long blocks = ((long)n_img * H + RB - 1) / RB;
  if (blocks > 1024) blocks = 1024;
  LY_WITH_T(dtype, hipLaunchKernelGGL(ly_detect_head_bwd_kernel<T>, dim3((unsigned)blocks), dim3(LY_THREADS), 0, reinterpret_cast<hipStream_t>(stream), dp,
                                      n_img, H, W, na, no, reinterpret_cast<T*>(du), ldu, dbias, dbias_f64, RB));
  LY_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// Train-mode BatchNorm support kernels (statistics passes; normalisation itself is folded into the
// consumers' scale/shift exactly like the eval path)
// ---------------------------------------------------------------------------------------------------
// per-channel sum / sum of squares over all rows of an [rows, C] matrix: mom[c] += sum x, mom[C + c] += sum x^2
template <typename T>
__global__ __launch_bounds__(LY_THREADS) void ly_chan_moments_kernel(const T* __restrict__ x, int ldx, long rows, int C,
                                                                      double* __restrict__ mom) {
  __shared__ f32x4 red1[LY_THREADS], red2[LY_THREADS];
  const int nc4 = C >> 2, tid = threadIdx.x;
  const int groups = LY_THREADS / nc4;
  const int c4 = tid % nc4, j0 = tid / nc4;
  f32x4 s1 = ly_zero4(), s2 = ly_zero4();
  if (j0 < groups) {
    // four rows per trip, all four loads issued before the first use (one row per trip paid a round trip per row: 21 us for the 8 MB of
    // RFCBAMConv layer 9's input, on 67 blocks); rows past the end re-read the last row and are masked
    const long step = (long)gridDim.x * groups;
    for (long r = (long)blockIdx.x * groups + j0; r < rows; r += 4 * step) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long rr = r + u * step;
        v[u] = ly_ld4<T>(x + (rr < rows ? rr : rows - 1) * ldx + 4 * c4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const f32x4 w = r + u * step < rows ? v[u] : ly_zero4();
        s1 += w;
        s2 += w * w;
      }
    }
  }
  red1[tid] = s1; red2[tid] = s2;
  __syncthreads();
  if (j0 == 0) {
    for (int g = 1; g < groups; ++g) { s1 += red1[g * nc4 + c4]; s2 += red2[g * nc4 + c4]; }
    double* m = mom + (size_t)(blockIdx.x & (LY_STATS_STRIPES - 1)) * 2 * C;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      atomicAdd(m + 4 * c4 + r, (double)s1[r]);
      atomicAdd(m + C + 4 * c4 + r, (double)s2[r]);
    }
  }
}

extern "C" int ly_chan_moments(const void* x, int ldx, long rows, int C, double* mom, int dtype, void* stream) {
  LY_CHECK_DTYPE(dtype, "chan_moments");
  LY_CHECK(x && mom && (C & 3) == 0 && (ldx & 3) == 0 && C <= 1024 && rows > 0, "chan_moments: bad arguments");
  const int groups = LY_THREADS / (C >> 2);
  long blocks = (rows + groups * 16L - 1) / (groups * 16L);    // ~16 rows per row group: four trips of four rows in flight
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  LY_WITH_T(dtype, hipLaunchKernelGGL(ly_chan_moments_kernel<T>, dim3((unsigned)blocks), dim3(LY_THREADS), 0, reinterpret_cast<hipStream_t>(stream),
                                      reinterpret_cast<const T*>(x), ldx, rows, C, mom));
  LY_LAUNCH_CHECK();
  return 0;
}

// CoordAtt bn1 statistics: y1[n, pos, m] = w1[m, :] . pool[n, pos, :] + b1[m];  stats[m] += y1, stats[mip + m] += y1^2
// One WAVE per position: lanes stride the channels and carry all `mip` outputs (MIPMAX accumulators), one shuffle tree per position,
// lanes m < mip keep the running sums; the four waves of a block meet in LDS and the block adds ONCE per output.  (The first version
// gave each wave two outputs and walked the block's positions serially per output: 113 us per launch.)
template <int MIPMAX>
__global__ __launch_bounds__(LY_THREADS) void ly_coordatt_conv1_stats_kernel(const float* __restrict__ pool, long positions, int C, int mip,
                                                                              const float* __restrict__ w1, const float* __restrict__ b1,
                                                                              double* __restrict__ stats) {
  __shared__ float red[4][2 * MIPMAX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float a1 = 0.f, a2 = 0.f;
  // the lane that accumulates output m: lane m on the general path; on the fast path the reduce-scatter leaves output m in lane m * (64 / MIPMAX)
  constexpr bool FASTP = MIPMAX <= 16;
  constexpr int LSTEP = FASTP ? 64 / MIPMAX : 1;
  const bool fast = FASTP && C <= 256;
  const int mine = fast ? lane / LSTEP : lane;
  const bool owner = (fast ? (lane % LSTEP) == 0 : true) && mine < mip;
  const float bm = owner ? b1[mine] : 0.f;
  const long nw = (long)gridDim.x * 4;
  if constexpr (FASTP) if (fast) {
    // the lane's weights stay in registers over all positions, and two positions per trip have all their loads issued before the first use
    // (the one-position loop re-read w1 and paid a dependent round trip per position: 23 us for 2.6 MB)
    constexpr int KC = 4;
    float wreg[KC][MIPMAX];
#pragma unroll
    for (int k = 0; k < KC; ++k)
#pragma unroll
      for (int m = 0; m < MIPMAX; ++m) wreg[k][m] = (m < mip && lane + 64 * k < C) ? w1[m * C + lane + 64 * k] : 0.f;
    constexpr int UP = 2;
    for (long pos0 = (long)blockIdx.x * 4 + wave; pos0 < positions; pos0 += UP * nw) {
      float pv[UP][KC];
#pragma unroll
      for (int u = 0; u < UP; ++u) {
        const long pos = pos0 + u * nw < positions ? pos0 + u * nw : positions - 1;
#pragma unroll
        for (int k = 0; k < KC; ++k) pv[u][k] = lane + 64 * k < C ? pool[pos * C + lane + 64 * k] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < UP; ++u) {
        const bool live = pos0 + u * nw < positions;
        float y[MIPMAX];
#pragma unroll
        for (int m = 0; m < MIPMAX; ++m) {
          y[m] = 0.f;
#pragma unroll
          for (int k = 0; k < KC; ++k) y[m] += wreg[k][m] * pv[u][k];
        }
        ly_reduce_scatter<MIPMAX>(y, lane);                 // MIPMAX - 1 + log2(64 / MIPMAX) cross-lane moves instead of 6 MIPMAX
        if (owner && live) {
          const float v = y[0] + bm;
          a1 += v;
          a2 += v * v;
        }
      }
    }
  }
  if (!fast) {
  for (long pos = (long)blockIdx.x * 4 + wave; pos < positions; pos += nw) {
    const float* p = pool + pos * C;
    float y[MIPMAX];
#pragma unroll
    for (int m = 0; m < MIPMAX; ++m) y[m] = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float pv = p[c];
#pragma unroll
      for (int m = 0; m < MIPMAX; ++m)
        if (m < mip) y[m] += w1[m * C + c] * pv;
    }
#pragma unroll
    for (int m = 0; m < MIPMAX; ++m) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) y[m] += __shfl_xor(y[m], o);
      if (lane == m) {
        const float v = y[m] + bm;
        a1 += v;
        a2 += v * v;
      }
    }
  }
  }
  if (owner) { red[wave][mine] = a1; red[wave][MIPMAX + mine] = a2; }
  __syncthreads();
  if (wave == 0 && lane < mip) {
    atomicAdd(stats + lane, (double)((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane])));
    atomicAdd(stats + mip + lane, (double)((red[0][MIPMAX + lane] + red[1][MIPMAX + lane]) + (red[2][MIPMAX + lane] + red[3][MIPMAX + lane])));
  }
}

extern "C" int ly_coordatt_conv1_stats(const float* pool, long positions, int C, int mip, const float* w1, const float* b1,
                                       double* stats, void* stream) {
  LY_CHECK(pool && w1 && b1 && stats && positions > 0 && mip > 0, "coordatt_conv1_stats: bad arguments");
  LY_CHECK(mip <= 64, "coordatt_conv1_stats: mip=%d out of range", mip);
  long blocks = (positions + 3) / 4;
  if (blocks > 256) blocks = 256;                          // one atomic per (block, output): keep the adds per address few
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (mip <= 8) hipLaunchKernelGGL(ly_coordatt_conv1_stats_kernel<8>, dim3((unsigned)blocks), dim3(LY_THREADS), 0, st, pool, positions, C, mip, w1, b1, stats);
  else if (mip <= 16) hipLaunchKernelGGL(ly_coordatt_conv1_stats_kernel<16>, dim3((unsigned)blocks), dim3(LY_THREADS), 0, st, pool, positions, C, mip, w1, b1, stats);
  else hipLaunchKernelGGL(ly_coordatt_conv1_stats_kernel<64>, dim3((unsigned)blocks), dim3(LY_THREADS), 0, st, pool, positions, C, mip, w1, b1, stats);
  LY_LAUNCH_CHECK();
  return 0;
}

// RFCBAMConv k=3 `generate` BatchNorm statistics (train mode).  The pre-BN value of channel c, tap-output t at
// an output pixel is a_t = sum_u w[c,t,u] * x_u with x_u the 9 (zero padded, stride s) input taps of that pixel,
// so  sum a_t = w_t . m   and   sum a_t^2 = w_t^T M w_t  with  m[u] = sum_pixels x_u,  M[u][v] = sum_pixels x_u x_v.
// This kernel accumulates the 9 + 45 moments per channel: thread = channel (coalesced NHWC reads), blocks split
// the output pixels, 54 vectorised atomics per thread at the end.  mom layout: [54][C].
template <typename T>
__global__ __launch_bounds__(LY_THREADS) void ly_rfcbam_tap_moments_kernel(const T* __restrict__ x, int ldx, int n_img, int H, int W,
                                                                            int C, int Ho, int Wo, int s, double* __restrict__ mom) {
  const int cb = C < LY_THREADS ? C : LY_THREADS;          // channels handled per pass by this block
  const int subs = LY_THREADS / cb;
  const int tid = threadIdx.x;
  const int cl = tid % cb, sub = tid / cb;
  const long npix = (long)n_img * Ho * Wo;
  for (int c0 = 0; c0 < C; c0 += cb) {
    const int c = c0 + cl;
    float m1[9], m2[45];
#pragma unroll
    for (int i = 0; i < 9; ++i) m1[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 45; ++i) m2[i] = 0.f;
    if (sub < subs && c < C) {
      // two output pixels per trip, their 18 taps loaded unconditionally from clamped coordinates before the first use (`ok ? load : 0` is a
      // branch around a load: every wait after it turns conservative and the taps arrive one round trip at a time); masks in the arithmetic
      const long stride = (long)gridDim.x * subs;
      for (long p0 = (long)blockIdx.x * subs + sub; p0 < npix; p0 += 2 * stride) {
        float xv[2][9];
        bool okk[2][9];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const long p = p0 + j * stride;
          const bool live = p < npix;
          const long pc = live ? p : npix - 1;
          const int ox = (int)(pc % Wo);
          const long q = pc / Wo;
          const int oy = (int)(q % Ho);
          const long n = q / Ho;
#pragma unroll
          for (int u = 0; u < 9; ++u) {
            const int iy = s * oy + u / 3 - 1, ix = s * ox + u % 3 - 1;
            okk[j][u] = live && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const int cy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy), cx = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
            xv[j][u] = ly_ld1<T>(x + ((n * H + cy) * W + cx) * ldx + c);
          }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
          for (int u = 0; u < 9; ++u) xv[j][u] = okk[j][u] ? xv[j][u] : 0.f;
          int k = 0;
#pragma unroll
          for (int u = 0; u < 9; ++u) {
            m1[u] += xv[j][u];
#pragma unroll
            for (int v = u; v < 9; ++v) m2[k++] += xv[j][u] * xv[j][v];
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 9; ++i) atomicAdd(mom + (long)i * C + c, (double)m1[i]);
#pragma unroll
      for (int i = 0; i < 45; ++i) atomicAdd(mom + (long)(9 + i) * C + c, (double)m2[i]);
    }
  }
}

// Stride-2 form through an LDS tile (the two k = 3 RFCBAMConv layers of the detector): a block walks 8 x 8 tiles of output pixels; the 17 x 17 input
// pixels x 64 channels of a tile are staged ONCE with 16-byte loads, lane = channel then reads its nine taps from LDS.  The 54 moments stay in
// registers over all tiles of the block and are reduced over its four waves before ONE atomic per moment, channel and block — the per-pixel
// kernel above issues eighteen 2-byte loads per trip and 54 atomics per thread (91 us per launch at bs=64; fewer blocks starve it of loads in
// flight, more blocks drown it in atomics).
#define LY_TM_T 8
template <typename T>
__global__ __launch_bounds__(LY_THREADS) void ly_rfcbam_tap_moments_s2t_kernel(const T* __restrict__ x, int ldx, int n_img, int H, int W, int C, int Ho, int Wo,
                                                                                int tiles_y, int tiles_x, double* __restrict__ mom) {
  constexpr int TM = LY_TM_T, TW = 2 * TM + 1, NPX = TW * TW;
  constexpr int VE = 16 / (int)sizeof(T), VPR = 64 / VE;
  extern __shared__ f32x4 ly_tm_smem[];
  T* const tile = reinterpret_cast<T*>(ly_tm_smem);              // [NPX][64]
  float* const red = reinterpret_cast<float*>(ly_tm_smem);       // reused after the walk: [4][27][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = blockIdx.y * 64;
  const int c = c0 + lane;
  const bool cok = c < C;
  // two output pixels per trip as the halves of float2 accumulators: the 54 multiply-adds per pixel become 54 packed ones per PAIR
  // (v_pk_fma_f32) — the kernel is bound by exactly these
  typedef float ly_f2 __attribute__((ext_vector_type(2)));
  ly_f2 a1[9], a2[45];
#pragma unroll
  for (int i = 0; i < 9; ++i) a1[i] = (ly_f2){0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 45; ++i) a2[i] = (ly_f2){0.f, 0.f};
  const long ntiles = (long)n_img * tiles_y * tiles_x;
  for (long tix = ly_xcd_remap((int)blockIdx.x, (int)gridDim.x); tix < ntiles; tix += gridDim.x) {
    const int tx = (int)(tix % tiles_x);
    const long q = tix / tiles_x;
    const int ty = (int)(q % tiles_y);
    const long n = q / tiles_y;
    const int oy0 = ty * TM, ox0 = tx * TM;
    __syncthreads();                                             // the previous tile has been consumed
    for (int i = tid; i < NPX * VPR; i += LY_THREADS) {
      const int px = i / VPR, v = i - px * VPR;
      const int iy = 2 * oy0 - 1 + px / TW, ix = 2 * ox0 - 1 + px % TW;
      const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W && c0 + v * VE < C;
      const int cy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy), cx = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
      ly_u32x4 raw = *reinterpret_cast<const ly_u32x4*>(x + ((n * H + cy) * W + cx) * (long)ldx + (c0 + v * VE < C ? c0 + v * VE : 0));
      if (!ok) raw = (ly_u32x4){0u, 0u, 0u, 0u};
      *reinterpret_cast<ly_u32x4*>(tile + (long)px * 64 + v * VE) = raw;
    }
    __syncthreads();
    for (int p = wave; p < TM * TM; p += 8) {                   // pixels p and p + 4 (TM*TM is a multiple of 8)
      const int pa = p, pb = p + 4;
      const int ya = pa / TM, xa = pa - ya * TM, yb = pb / TM, xb = pb - yb * TM;
      const bool la = oy0 + ya < Ho && ox0 + xa < Wo, lb = oy0 + yb < Ho && ox0 + xb < Wo;
      ly_f2 xv[9];
#pragma unroll
      for (int u = 0; u < 9; ++u) {
        const float va = (float)tile[((2 * ya + u / 3) * TW + 2 * xa + u % 3) * 64 + lane];
        const float vb = (float)tile[((2 * yb + u / 3) * TW + 2 * xb + u % 3) * 64 + lane];
        xv[u] = (ly_f2){la ? va : 0.f, lb ? vb : 0.f};
      }
      int k = 0;
#pragma unroll
      for (int u = 0; u < 9; ++u) {
        a1[u] += xv[u];
#pragma unroll
        for (int v = u; v < 9; ++v) a2[k++] += xv[u] * xv[v];
      }
    }
  }
  float m1[9], m2[45];
#pragma unroll
  for (int i = 0; i < 9; ++i) m1[i] = a1[i][0] + a1[i][1];
#pragma unroll
  for (int i = 0; i < 45; ++i) m2[i] = a2[i][0] + a2[i][1];
  // block reduction over the four waves, two rounds of 27 moments through the tile's LDS, then one atomic per (moment, channel)
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 27; ++i) {
      const int e = half * 27 + i;
      red[(wave * 27 + i) * 64 + lane] = e < 9 ? m1[e < 9 ? e : 0] : m2[e >= 9 ? e - 9 : 0];
    }
    __syncthreads();
    for (int j = tid; j < 27 * 64; j += LY_THREADS) {
      const int i = j >> 6, l = j & 63;
      if (c0 + l < C) atomicAdd(mom + (long)(half * 27 + i) * C + c0 + l, (double)((red[j] + red[27 * 64 + j]) + (red[2 * 27 * 64 + j] + red[3 * 27 * 64 + j])));
    }
  }
  (void)cok;
}

extern "C" int ly_rfcbam_tap_moments(const void* x, int ldx, int n_img, int H, int W, int C, int s, double* mom, int dtype, void* stream) {
  LY_CHECK_DTYPE(dtype, "rfcbam_tap_moments");
  LY_CHECK(x && mom && s >= 1 && C > 0, "rfcbam_tap_moments: bad arguments");
  const int Ho = (H + 2 - 3) / s + 1, Wo = (W + 2 - 3) / s + 1;
  long npix = (long)n_img * Ho * Wo;
  const int esz = dtype == LY_BF16 ? 2 : 4;
  if (s == 2 && C % (16 / esz) == 0 && ldx % (16 / esz) == 0 && ((uintptr_t)x & 15) == 0) {
    const int tiles_y = (Ho + LY_TM_T - 1) / LY_TM_T, tiles_x = (Wo + LY_TM_T - 1) / LY_TM_T;
    const int groups = (C + 63) / 64;
    long nb = (long)n_img * tiles_y * tiles_x;
    const long cap = 768 / groups > 0 ? 768 / groups : 1;               // ~768 blocks: 3456 atomics each
    if (nb > cap) nb = cap;
    constexpr int NPX = (2 * LY_TM_T + 1) * (2 * LY_TM_T + 1);
    size_t lds = (size_t)NPX * 64 * esz;
    if (lds < 4 * 27 * 64 * sizeof(float)) lds = 4 * 27 * 64 * sizeof(float);
    LY_WITH_T(dtype, {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ly_rfcbam_tap_moments_s2t_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(ly_rfcbam_tap_moments_s2t_kernel<T>, dim3((unsigned)nb, (unsigned)groups), dim3(LY_THREADS), lds, reinterpret_cast<hipStream_t>(stream),
                         reinterpret_cast<const T*>(x), ldx, n_img, H, W, C, Ho, Wo, tiles_y, tiles_x, mom);
    });
    LY_LAUNCH_CHECK();
    return 0;
  }
  long blocks = npix / 32 + 1;
  if (blocks > 1024) blocks = 1024;
  LY_WITH_T(dtype, hipLaunchKernelGGL(ly_rfcbam_tap_moments_kernel<T>, dim3((unsigned)blocks), dim3(LY_THREADS), 0, reinterpret_cast<hipStream_t>(stream),
                                      reinterpret_cast<const T*>(x), ldx, n_img, H, W, C, Ho, Wo, s, mom));
  LY_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// CoordAtt vector MLP, training backward (models/common.py:1600-1607 under autograd):
//   y0 = W1 pool + b1;  xh = (y0 - mean) * invstd;  y1 = gamma*xh + beta;  y2 = h_swish(y1);  a = sigmoid(Wx y2 + bx)
// with Wx = conv_h for the H row positions of an image and conv_w for its W column positions.  Two launches, because
// BatchNorm's backward needs sum(dy1) and sum(dy1*xh) over ALL positions before any dy0 exists:
//   bwd1 (one WAVE per position, lanes stride the channels, both length-C reductions are shuffle trees, nothing synchronises):
//         dz = da*a*(1-a) -> dpool buffer;  dy1 = (Wx^T dz) * h_swish'(y1);  (dy1, xh, y2) -> ws;  sums += (dy1, dy1*xh)  (striped)
//   bwd2 (lane = channel, block = 64 channels x a chunk of positions, waves = row phases):
//         dy0 = gamma*invstd*(dy1 - S1/R - xh*S2/R);  dpool = W1^T dy0 (over dz, in place);  dW1 += dy0 (x) pool;
//         dWx += dz (x) y2, dbx += dz;  dgamma += S2, dbeta += S1.   (db1 = sum dy0 is identically zero: BatchNorm removes the mean.)
// Every parameter gradient is summed in registers, then over the block's waves through LDS, and ADDED to its target once per block
// (a few dozen adds per address: same-address float atomics serialise, ~0.05 us each).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float ly_hswish_grad(float x) { return x <= -3.f ? 0.f : (x >= 3.f ? 1.f : (2.f * x + 3.f) * (1.f / 6.f)); }
#define LY_CA_STRIPES 32

template <int MIP, int NS>
__global__ __launch_bounds__(LY_THREADS) void ly_coordatt_mlp_bwd1_kernel(
    const float* __restrict__ pool, int n_img, int H, int W, int C, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ wh, const float* __restrict__ ww, const float* __restrict__ a_h, const float* __restrict__ a_w,
    const float* __restrict__ da_h, const float* __restrict__ da_w, float* __restrict__ ws, double* __restrict__ sums, float* __restrict__ dzb) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int L = H + W;
  const long R = (long)n_img * L;
  float s1 = 0.f, s2 = 0.f;                                // the lane that owns output m: sum dy1[m], sum dy1[m]*xh[m]
  static_assert(MIP == 8 || MIP == 16, "ly_coordatt_mlp_bwd1: MIP = 8 or 16");
  constexpr int LSTEP = 64 / MIP;
  const int mine = lane / LSTEP;
  const bool owner = (lane % LSTEP) == 0;
  const float b1m = b1[mine], meanm = mean[mine], invm = invstd[mine], gm = gamma[mine], bem = beta[mine];
  const long nw = (long)gridDim.x * 4;
  for (long r = (long)blockIdx.x * 4 + wave; r < R; r += nw) {
    const long n = r / L;
    const int pos = (int)(r - n * L);
    const bool isrow = pos < H;
    const long q = isrow ? n * H + pos : n * W + (pos - H);
    const float* wx = isrow ? wh : ww;
    const float* am = (isrow ? a_h : a_w) + q * C;
    const float* dam = (isrow ? da_h : da_w) + q * C;
    float y[MIP], d[MIP];
#pragma unroll
    for (int m = 0; m < MIP; ++m) { y[m] = 0.f; d[m] = 0.f; }
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
      const int c = lane + 64 * sl;
      if (c < C) {
        const float pv = pool[r * C + c];
        const float av = am[c];
        const float dz = dam[c] * av * (1.f - av);
        dzb[r * C + c] = dz;
#pragma unroll
        for (int m = 0; m < MIP; ++m) {
          y[m] += w1[m * C + c] * pv;
          d[m] += dz * wx[c * MIP + m];
        }
      }
    }
    // both length-C reductions as reduce-scatters (ly_common.hpp): 2 (MIP - 1 + log2(64 / MIP)) cross-lane moves per position instead of 12 MIP;
    // output m ends up in lane m * (64 / MIP), which keeps its running sums
    ly_reduce_scatter<MIP>(y, lane);
    ly_reduce_scatter<MIP>(d, lane);
    if (owner) {
      const float xh = (y[0] + b1m - meanm) * invm;
      const float y1 = gm * xh + bem;
      const float g = d[0] * ly_hswish_grad(y1);
      s1 += g;
      s2 += g * xh;
      ws[r * 3 * MIP + mine] = g;
      ws[r * 3 * MIP + MIP + mine] = xh;
      ws[r * 3 * MIP + 2 * MIP + mine] = ly_hswish(y1);
    }
  }
  if (owner) {
    double* st = sums + ((blockIdx.x * 4 + wave) & (LY_CA_STRIPES - 1)) * 2 * MIP;      // double accumulators: see ly_stats_flush (ly_common.hpp)
    atomicAdd(st + mine, (double)s1);
    atomicAdd(st + MIP + mine, (double)s2);
  }
}

template <int MIP>
__global__ __launch_bounds__(LY_THREADS) void ly_coordatt_mlp_bwd2_kernel(
    const float* __restrict__ pool, int n_img, int H, int W, int C, long rows_per_block, const float* __restrict__ w1,
    const float* __restrict__ gamma, const float* __restrict__ invstd, const float* __restrict__ ws, const double* __restrict__ sums,
    float* __restrict__ dpool /* in: dz */, float* __restrict__ dw1, float* __restrict__ dgamma, float* __restrict__ dbeta,
    float* __restrict__ dwh, float* __restrict__ dbh, float* __restrict__ dww, float* __restrict__ dbw, const int f64) {
  constexpr int NA = 3 * MIP + 2;                          // accumulators per channel: dW1[m], dWh[m], dWw[m], dbh, dbw
  __shared__ float red[4 * 64 * (3 * MIP + 2)];
  __shared__ float ssum[2 * MIP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int L = H + W;
  const long R = (long)n_img * L;
  const int c = blockIdx.x * 64 + lane;
  const bool cok = c < C;
  if (tid < 2 * MIP) {
    double v = 0.0;
    for (int st = 0; st < LY_CA_STRIPES; ++st) v += sums[st * 2 * MIP + tid];
    ssum[tid] = (float)v;
  }
  __syncthreads();
  const float invR = 1.f / (float)R;
  float k0[MIP], m1[MIP], m2[MIP], w1r[MIP];
  float acc[NA];
#pragma unroll
  for (int m = 0; m < MIP; ++m) {
    k0[m] = gamma[m] * invstd[m];
    m1[m] = ssum[m] * invR;
    m2[m] = ssum[MIP + m] * invR;
    w1r[m] = cok ? w1[m * C + c] : 0.f;
  }
#pragma unroll
  for (int e = 0; e < NA; ++e) acc[e] = 0.f;
  const long r_lo = (long)blockIdx.y * rows_per_block;
  const long r_hi = r_lo + rows_per_block < R ? r_lo + rows_per_block : R;
  // four rows per trip, every load of the trip issued before the first use (the one-row loop paid a dependent round trip per row: 47 us
  // for 2.6 MB of pooled vectors); rows past the block's range re-read its last row and are masked
  constexpr int UR = 4;
  for (long r0 = r_lo + wave; r0 < r_hi; r0 += 4 * UR) {
    float pv[UR], dz[UR], wv[UR][3 * MIP];
    bool live[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const long rr = r0 + 4 * u;
      live[u] = rr < r_hi;
      const long r = live[u] ? rr : r_hi - 1;
      const float* wr = ws + r * 3 * MIP;                  // wave-uniform row: (dy1, xh, y2)
#pragma unroll
      for (int m = 0; m < 3 * MIP; ++m) wv[u][m] = wr[m];
      pv[u] = cok ? pool[r * C + c] : 0.f;
      dz[u] = cok ? dpool[r * C + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const long r = r0 + 4 * u;
      const int pos = (int)(r % L);
      const bool isrow = pos < H;
      const float dzu = live[u] ? dz[u] : 0.f, pvu = live[u] ? pv[u] : 0.f;
      float g = 0.f;
#pragma unroll
      for (int m = 0; m < MIP; ++m) {
        const float dy0 = live[u] ? k0[m] * (wv[u][m] - m1[m] - wv[u][MIP + m] * m2[m]) : 0.f;
        const float t = dzu * wv[u][2 * MIP + m];
        g += dy0 * w1r[m];
        acc[m] += dy0 * pvu;
        acc[MIP + m] += isrow ? t : 0.f;
        acc[2 * MIP + m] += isrow ? 0.f : t;
      }
      acc[3 * MIP] += isrow ? dzu : 0.f;
      acc[3 * MIP + 1] += isrow ? 0.f : dzu;
      if (cok && live[u]) dpool[r * C + c] = g;
    }
  }
#pragma unroll
  for (int e = 0; e < NA; ++e) red[(wave * 64 + lane) * NA + e] = acc[e];
  __syncthreads();
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid < MIP) {
    ly_gacc(dgamma, tid, ssum[MIP + tid], f64);
    ly_gacc(dbeta, tid, ssum[tid], f64);
  }
  for (int e = tid; e < 64 * NA; e += LY_THREADS) {
    const int cl = e / NA, k = e - cl * NA;
    const int cc = blockIdx.x * 64 + cl;
    if (cc >= C) continue;
    const float v = red[e] + red[64 * NA + e] + red[2 * 64 * NA + e] + red[3 * 64 * NA + e];
    if (k < MIP) ly_gacc(dw1, k * C + cc, v, f64);
    else if (k < 2 * MIP) ly_gacc(dwh, cc * MIP + (k - MIP), v, f64);
    else if (k < 3 * MIP) ly_gacc(dww, cc * MIP + (k - 2 * MIP), v, f64);
    else if (k == 3 * MIP) ly_gacc(dbh, cc, v, f64);
    else ly_gacc(dbw, cc, v, f64);
  }
}

template <int MIP, int NS>
static void launch_coordatt_bwd(hipStream_t st, const float* pool, int n_img, int H, int W, int C, const float* w1, const float* b1, const float* mean,
                                const float* invstd, const float* gamma, const float* beta, const float* wh, const float* ww, const float* a_h,
                                const float* a_w, const float* da_h, const float* da_w, float* ws, double* sums, float* dpool, float* dw1,
                                float* dgamma, float* dbeta, float* dwh, float* dbh, float* dww, float* dbw, int f64) {
  const long R = (long)n_img * (H + W);
  long b1n = (R + 3) / 4;
  if (b1n > 1024) b1n = 1024;
  hipLaunchKernelGGL((ly_coordatt_mlp_bwd1_kernel<MIP, NS>), dim3((unsigned)b1n), dim3(LY_THREADS), 0, st, pool, n_img, H, W, C, w1, b1, mean, invstd, gamma,
                     beta, wh, ww, a_h, a_w, da_h, da_w, ws, sums, dpool);
  const int groups = (C + 63) / 64;
  long chunks = (96 + groups - 1) / groups;                // ~96 blocks; each target address then receives `chunks` same-address adds
  if (chunks > (R + 31) / 32) chunks = (R + 31) / 32;
  if (chunks < 1) chunks = 1;
  const long rpb = (R + chunks - 1) / chunks;
  chunks = (R + rpb - 1) / rpb;
  hipLaunchKernelGGL((ly_coordatt_mlp_bwd2_kernel<MIP>), dim3((unsigned)groups, (unsigned)chunks), dim3(LY_THREADS), 0, st, pool, n_img, H, W, C, rpb, w1,
                     gamma, invstd, ws, sums, dpool, dw1, dgamma, dbeta, dwh, dbh, dww, dbw, f64);
}

extern "C" int ly_coordatt_mlp_bwd(const float* pool, int n_img, int H, int W, int C, int mip, const float* w1, const float* b1,
                                   const float* mean, const float* invstd, const float* gamma, const float* beta, const float* wh,
                                   const float* ww, const float* a_h, const float* a_w, const float* da_h, const float* da_w, float* ws,
                                   double* sums, float* dpool, float* dw1, float* dgamma, float* dbeta, float* dwh, float* dbh,
                                   float* dww, float* dbw, int grads_f64, void* stream) {
  LY_CHECK(pool && w1 && b1 && mean && invstd && gamma && beta && wh && ww && a_h && a_w && da_h && da_w && ws && sums && dpool && dw1 &&
               dgamma && dbeta && dwh && dbh && dww && dbw, "coordatt_mlp_bwd: null pointer");
  LY_CHECK((mip == 8 || mip == 16) && C > 0 && C <= 512 && n_img > 0 && H > 0 && W > 0, "coordatt_mlp_bwd: built for mip 8 / 16 and C <= 512 (mip=%d C=%d)", mip, C);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define LY_CA_BWD(MIP, NS) launch_coordatt_bwd<MIP, NS>(st, pool, n_img, H, W, C, w1, b1, mean, invstd, gamma, beta, wh, ww, a_h, a_w, da_h, da_w, ws, sums, \
                                                        dpool, dw1, dgamma, dbeta, dwh, dbh, dww, dbw, grads_f64)
  const int ns = (C + 63) / 64;
  if (mip == 8) {
    if (ns <= 1) LY_CA_BWD(8, 1); else if (ns <= 2) LY_CA_BWD(8, 2); else if (ns <= 4) LY_CA_BWD(8, 4); else LY_CA_BWD(8, 8);
  } else {
    if (ns <= 4) LY_CA_BWD(16, 4); else LY_CA_BWD(16, 8);
  }
#undef LY_CA_BWD
  LY_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// RFCBAMConv `generate` BatchNorm in training, everything between the moment kernel and the contraction in ONE launch
// (models/rfa.py:101-106 generate = depthwise conv -> BatchNorm -> ReLU, training=True):
//   from the tap moments of x (k = 3: ly_rfcbam_tap_moments, 9 first + 45 second moments per channel; k = 1: ly_chan_moments, sum x and
//   sum x^2) and the depthwise weights, per generate channel g = c*KK + t:
//     sum a = w_t . m,  sum a^2 = w_t^T M w_t   ->  batch mean / var  ->  scale = gamma*invstd, shift = beta - mean*scale,
//     running_mean / running_var (momentum, unbiased) and num_batches_tracked updated as nn.BatchNorm2d does;
//   outputs: scale, shift, mean, invstd in [c*KK + t] order, the same four in [t*C + c] order (what the backward kernels index),
//   and — k = 3 — the folded weights (w*scale | shift) in the two LDS orders of the statistics / contraction kernels
//   (pack.rfcbam_gen_weights: [C_pad/chunk][4 waves][9 t][chunk/8 pairs][10][2]); k = 1: a1 = w*scale.
// Replaces ~40 tiny torch launches per module and step (einsum / index_put / cat / permute / BatchNorm arithmetic).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ long ly_gw_index(int ch, int t, int e, int chunk, bool contiguous) {
  const int per = chunk >> 2;
  const int q = ch / chunk, r = ch - q * chunk;
  const int w = contiguous ? r / per : r & 3;
  const int j = contiguous ? r - w * per : r >> 2;
  const int p = j >> 1, ab = j & 1;
  return (((((long)q * 4 + w) * 9 + t) * (per >> 1) + p) * 10 + e) * 2 + ab;
}

__global__ __launch_bounds__(LY_THREADS) void ly_rfcbam_gen_prepare_kernel(
    const double* __restrict__ mom, int C, int KK, const float* __restrict__ gen_w, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, float momentum, double count, float* __restrict__ running_mean,
    float* __restrict__ running_var, long* __restrict__ nbt, float* __restrict__ out8 /* [8][C*KK] */, float* __restrict__ a1,
    float* __restrict__ wq_stats, int cp_stats, float* __restrict__ wq_main, int cp_main, float* __restrict__ wq_c, const int mom_stripes) {
  const int G = C * KK;
  const int cpm = cp_stats > cp_main ? cp_stats : cp_main;
  const int total = (KK == 9 ? (cpm > C ? cpm : C) : C) * KK;
  for (int g = blockIdx.x * LY_THREADS + threadIdx.x; g < total; g += gridDim.x * LY_THREADS) {
    const int c = g / KK, t = g - c * KK;
    if (c >= C) {                                          // padded channels of the packed images: zero weights
      for (int e = 0; e < 10; ++e) {
        if (c < cp_stats) wq_stats[ly_gw_index(c, t, e, 32, false)] = 0.f;
        if (c < cp_main) wq_main[ly_gw_index(c, t, e, 16, true)] = 0.f;
      }
      continue;
    }
    double s1 = 0.0, s2 = 0.0;
    float w[9];
    if (KK == 9) {
#pragma unroll
      for (int u = 0; u < 9; ++u) w[u] = gen_w[(long)g * 9 + u];
#pragma unroll
      for (int u = 0; u < 9; ++u) {
        s1 += (double)w[u] * mom[u * C + c];
#pragma unroll
        for (int v = u; v < 9; ++v) {
          const int tri = u * 9 - (u * (u - 1)) / 2 + (v - u);                 // torch.triu_indices(9, 9) order
          const double m = mom[(9 + tri) * C + c];
          s2 += (u == v ? 1.0 : 2.0) * (double)w[u] * (double)w[v] * m;
        }
      }
    } else {
      w[0] = gen_w[g];
      // k = 1: the moments as ly_chan_moments left them — mom_stripes copies of [2C] doubles, folded here in index order (was a launch of its own)
      double m1 = 0.0, m2 = 0.0;
      for (int st = 0; st < mom_stripes; ++st) {
        m1 += mom[(size_t)st * 2 * C + c];
        m2 += mom[(size_t)st * 2 * C + C + c];
      }
      s1 = (double)w[0] * m1;
      s2 = (double)w[0] * (double)w[0] * m2;
    }
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[g] * invstd;
    const float sh = beta[g] - (float)mean * sc;
    if (running_mean) {
      running_mean[g] = (1.f - momentum) * running_mean[g] + momentum * (float)mean;
      running_var[g] = (1.f - momentum) * running_var[g] + momentum * (float)(var * (count / (count > 1.0 ? count - 1.0 : 1.0)));
    }
    const int gt = t * C + c;
    out8[g] = sc;            out8[G + g] = sh;             out8[2 * G + g] = (float)mean;  out8[3 * G + g] = invstd;
    out8[4 * G + gt] = sc;   out8[5 * G + gt] = sh;        out8[6 * G + gt] = (float)mean; out8[7 * G + gt] = invstd;
    if (KK == 9) {
#pragma unroll
      for (int e = 0; e < 10; ++e) {
        const float v = e < 9 ? w[e] * sc : sh;
        wq_stats[ly_gw_index(c, t, e, 32, false)] = v;
        wq_main[ly_gw_index(c, t, e, 16, true)] = v;
        if (wq_c) {                                                            // lane = channel order, RAW form (ly_rf3c.hpp rc_load_w)
          const int i = e < 9 ? t * 9 + e : 81 + t;
          wq_c[(((long)(c >> 5) * 25 + (i >> 2)) * 32 + (c & 31)) * 4 + (i & 3)] = e < 9 ? w[e] : sh;
          if (e == 9) wq_c[(((long)(c >> 5) * 25 + ((90 + t) >> 2)) * 32 + (c & 31)) * 4 + ((90 + t) & 3)] = sc;
        }
      }
    } else {
      a1[g] = w[0] * sc;
    }
  }
  if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;
}

extern "C" int ly_rfcbam_gen_prepare(const double* mom, int C, int k, const float* gen_w, const float* gamma, const float* beta, float eps,
                                     float momentum, double count, float* running_mean, float* running_var, long* nbt, float* out8,
                                     float* a1, float* wq_stats, float* wq_main, float* wq_c, int mom_stripes, void* stream) {
  LY_CHECK(mom && gen_w && gamma && beta && out8 && C > 0 && (k == 1 || k == 3) && count > 0, "rfcbam_gen_prepare: bad arguments");
  LY_CHECK(mom_stripes >= 1 && (k == 1 || mom_stripes == 1), "rfcbam_gen_prepare: striped moments are read for k = 1 only (mom_stripes=%d)", mom_stripes);
  LY_CHECK(k == 1 ? a1 != nullptr : (wq_stats && wq_main), "rfcbam_gen_prepare: missing output for k=%d", k);
  LY_CHECK(!running_mean == !running_var, "rfcbam_gen_prepare: running_mean and running_var go together");
  const int KK = k * k;
  const int cp_s = (C + 31) / 32 * 32, cp_m = (C + 15) / 16 * 16;
  const int total = (k == 3 ? cp_s : C) * KK;
  hipLaunchKernelGGL(ly_rfcbam_gen_prepare_kernel, dim3((unsigned)((total + LY_THREADS - 1) / LY_THREADS)), dim3(LY_THREADS), 0,
                     reinterpret_cast<hipStream_t>(stream), mom, C, KK, gen_w, gamma, beta, eps, momentum, count, running_mean, running_var, nbt,
                     out8, a1, wq_stats, k == 3 ? cp_s : 0, wq_main, k == 3 ? cp_m : 0, k == 3 ? wq_c : nullptr, mom_stripes);
  LY_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// SE backward in training (models/rfa.py:88-92 under autograd), one block per image:
//   gap = mean x (from the forward's slice partials);  h = relu(Wa gap);  ca = sigmoid(Wb h)  (ca given)
//   dz = d_ca*ca*(1-ca);  dWb[c,r] += dz[c]*h[r];  dh = (Wb^T dz) * (h > 0);  dWa[r,c] += dh[r]*gap[c];  dgap[n,c] = Wa^T dh  (written)
// dgap / HW is what every pixel of x receives from the pooling: ly_rf_bwd_dx adds it while it writes dx.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LY_THREADS) void ly_se_bwd_kernel(const float* __restrict__ part, int slices, int C, float inv_hw,
                                                                const float* __restrict__ wa, const float* __restrict__ wb, int R,
                                                                const float* __restrict__ ca, const void* __restrict__ d_ca_v, const int d_ca_f64,
                                                                float* __restrict__ ws, float* __restrict__ dgap) {
  extern __shared__ float sm[];
  float* g = sm;                     // g[C] | dz[C] | hid[R] | dh[R] | red
  float* dz = sm + C;
  float* hid = sm + 2 * C;
  float* dh = hid + R;
  f32x4* red = reinterpret_cast<f32x4*>(sm + ((2 * C + 2 * R + 3) & ~3));
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nq = C >> 2, ng = LY_THREADS / nq;
  const int q = tid % nq, sg = tid / nq;
  f32x4 s4 = ly_zero4();
  if (sg < ng)
    for (int sl = sg; sl < slices; sl += ng) s4 += ly_ldg4(part + ((long)n * slices + sl) * C + 4 * q);
  red[tid] = s4;
  __syncthreads();
  if (sg == 0) {
    for (int k = 1; k < ng; ++k) s4 += red[tid + k * nq];
#pragma unroll
    for (int e = 0; e < 4; ++e) g[4 * q + e] = s4[e] * inv_hw;
  }
  for (int c = tid; c < C; c += LY_THREADS) {
    const float a = ca[(long)n * C + c];
    // d_ca as the double accumulators ly_rf1_bwd / ly_rf3c_bwd left it in (d_ca_f64), no conversion pass in between
    const float dca = d_ca_f64 ? (float)reinterpret_cast<const double*>(d_ca_v)[(long)n * C + c] : reinterpret_cast<const float*>(d_ca_v)[(long)n * C + c];
    dz[c] = dca * a * (1.f - a);
  }
  __syncthreads();
  for (int r = wave; r < R; r += 4) {
    float s = 0.f, d = 0.f;
    for (int c = lane; c < C; c += 64) {
      s += wa[r * C + c] * g[c];
      d += wb[c * R + r] * dz[c];
    }
    s = ly_group_sum(s, 64);
    d = ly_group_sum(d, 64);
    if (lane == 0) {
      hid[r] = fmaxf(s, 0.f);
      dh[r] = s > 0.f ? d : 0.f;
    }
  }
  __syncthreads();
  // the per-image factors of the two weight gradients go to the workspace ws[n][g | dz | hid | dh]; ly_se_bwd_wsum_kernel sums the outer
  // products over the images, one thread per weight (2*C*R float atomics per image here made the launch 38 us at bs=64)
  float* const wsn = ws + (long)n * (2 * C + 2 * R);
  for (int c = tid; c < C; c += LY_THREADS) {
    float dg = 0.f;
    for (int r = 0; r < R; ++r) dg += dh[r] * wa[r * C + c];
    dgap[(long)n * C + c] = dg;
    wsn[c] = g[c];
    wsn[C + c] = dz[c];
  }
  for (int r = tid; r < R; r += LY_THREADS) {
    wsn[2 * C + r] = hid[r];
    wsn[2 * C + R + r] = dh[r];
  }
}

// dWb[c][r] += sum_n dz[n][c]*hid[n][r],  dWa[r][c] += sum_n dh[n][r]*g[n][c]   (thread = (r, c), c fastest: deterministic, no atomics)
__global__ __launch_bounds__(LY_THREADS) void ly_se_bwd_wsum_kernel(const float* __restrict__ ws, int n_img, int C, int R, float* __restrict__ dwa,
                                                                     float* __restrict__ dwb) {
  const int i = blockIdx.x * LY_THREADS + threadIdx.x;
  if (i >= C * R) return;
  const int r = i / C, c = i - r * C;
  const int st = 2 * C + 2 * R;
  // eight images per trip, all 32 loads issued before the first use (one image per trip was 64 dependent L2 round trips: 19 us for 80 KB);
  // the sums keep the image order
  float sa = 0.f, sb = 0.f;
  int n = 0;
  for (; n + 8 <= n_img; n += 8) {
    float a0[8], a1[8], b0[8], b1[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float* w = ws + (long)(n + k) * st;
      b0[k] = w[C + c]; b1[k] = w[2 * C + r]; a0[k] = w[2 * C + R + r]; a1[k] = w[c];
    }
    __builtin_amdgcn_sched_barrier(0);               // (left alone the scheduler sinks the loads back between the FMAs: four per round trip again)
#pragma unroll
    for (int k = 0; k < 8; ++k) { sb += b0[k] * b1[k]; sa += a0[k] * a1[k]; }
  }
  for (; n < n_img; ++n) {
    const float* w = ws + (long)n * st;
    sb += w[C + c] * w[2 * C + r];
    sa += w[2 * C + R + r] * w[c];
  }
  dwb[c * R + r] += sb;
  dwa[r * C + c] += sa;
}

extern "C" int ly_se_bwd(const float* part, int slices, int n_img, int HW, int C, const float* wa, const float* wb, int R, const float* ca,
                         const void* d_ca, int d_ca_f64, float* dwa, float* dwb, float* dgap, float* ws, void* stream) {
  LY_CHECK(part && wa && wb && ca && d_ca && dwa && dwb && dgap && ws, "se_bwd: null pointer");
  LY_CHECK((C & 3) == 0 && C <= 1024 && slices > 0 && R > 0 && R <= 256 && n_img > 0 && HW > 0, "se_bwd: bad arguments");
  hipLaunchKernelGGL(ly_se_bwd_kernel, dim3(n_img), dim3(LY_THREADS), sizeof(float) * (((2 * C + 2 * R + 3) & ~3) + 4 * LY_THREADS),
                     reinterpret_cast<hipStream_t>(stream), part, slices, C, 1.f / (float)HW, wa, wb, R, ca, d_ca, d_ca_f64, ws, dgap);
  LY_LAUNCH_CHECK();
  hipLaunchKernelGGL(ly_se_bwd_wsum_kernel, dim3((unsigned)((C * R + LY_THREADS - 1) / LY_THREADS)), dim3(LY_THREADS), 0,
                     reinterpret_cast<hipStream_t>(stream), ws, n_img, C, R, dwa, dwb);
  LY_LAUNCH_CHECK();
  return 0;
}
