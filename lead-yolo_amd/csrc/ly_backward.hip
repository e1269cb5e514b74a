// Backward building blocks of the training step, gfx950; activations / activation gradients T = float or
// __bf16, weight gradients, BatchNorm sums and attention-table gradients fp32.
//
//   ly_bnact_bwd_reduce / ly_bnact_bwd_apply   BatchNorm(train) + activation backward on an [rows, C] matrix
//   ly_wgrad                                   weight gradient: contraction over PIXELS with the forward's gather
//   ly_up2_bwd, ly_unpatch2                    adjoints of the nearest-2x read and of the k=s=2 patch gather
//
// Data gradients (dgrad) of the 1x1 / 3x3 convolutions reuse the forward contraction kernels
// (ly_gemm_fwd / ly_conv3x3_fwd) with transposed, frag-packed weights: the adjoint of a stride-1
// convolution is a stride-1 convolution.
//
// Replaces what autograd derives for Conv2d -> BatchNorm2d -> SiLU/ReLU in the reference's
// `scaler.scale(loss).backward()` (train.py:324) over models/common.py:1890-1910 (Conv),
// :1478-1482 (MLPBlock), :1537-1561 (patch layers).
#include "ly_tile.hpp"
#include "ly_params.h"

// -------------------------------------------------------------------------------------------------
// BN(train)+activation backward.  Forward:  v = a[c]*u + b[c],  y = act(v).
//   reduce:  s1[c] = sum_r dv,  s2[c] = sum_r dv*u          with dv = dy * act'(v)
//   apply :  du = alpha[c]*dv + kappa[c] + lambda[c]*u      (coefficients built on the host from s1, s2:
//            the usual  gamma*invstd*(dv - mean(dv) - xhat*mean(dv*xhat))  written as an affine map of (dv, u))
// -------------------------------------------------------------------------------------------------
template <int ACT>
__device__ __forceinline__ f32x4 ly_dact4(const f32x4 v, const f32x4 dy) {
  f32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (ACT == LY_ACT_RELU) {
      r[i] = v[i] > 0.f ? dy[i] : 0.f;
    } else if (ACT == LY_ACT_SILU) {
      const float s = ly_sigmoid(v[i]);
      r[i] = dy[i] * s * (1.f + v[i] * (1.f - s));
    } else {
      r[i] = dy[i];
    }
  }
  return r;
}

// Row ranges per XCD for the streaming BatchNorm / activation passes: XCD x (= blockIdx.x % 8, round-robin dispatch) walks the x-th eighth of the
// items with its own blocks, so that a row is written / read by the XCD whose L2 the neighbouring tile kernels (ly_xcd_remap: contiguous
// tile ranges per XCD) use for the same rows.  (LY_XCD_STREAM=0 at build time: the plain grid-stride order.)
#ifndef LY_XCD_STREAM
#define LY_XCD_STREAM 1
#endif
#if LY_XCD_STREAM
#define LY_XCD_RANGE(total)                                                                                             \
  const bool xr_on = gridDim.x >= 8;                          /* fewer blocks than XCDs: an eighth would have no block */ \
  const long xr_x = blockIdx.x & 7, xr_slot = blockIdx.x >> 3, xr_n = ((long)gridDim.x + 7 - xr_x) >> 3;                  \
  const long xr_per = (((total) + 7) / 8 + LY_THREADS - 1) / LY_THREADS * LY_THREADS;                                      \
  const long xr_lo = xr_x * xr_per, xr_hi = xr_lo + xr_per < (total) ? xr_lo + xr_per : (total);                           \
  const long xr_begin = xr_on ? xr_lo + xr_slot * LY_THREADS + threadIdx.x : (long)blockIdx.x * LY_THREADS + threadIdx.x; \
  const long xr_end = xr_on ? xr_hi : (total), xr_step = (xr_on ? xr_n : (long)gridDim.x) * LY_THREADS
#else
#define LY_XCD_RANGE(total)                                                                                             \
  const long xr_begin = (long)blockIdx.x * LY_THREADS + threadIdx.x, xr_end = (total), xr_step = (long)gridDim.x * LY_THREADS
#endif

// The same split over ROWS for the kernels whose threads keep a fixed channel vector and walk rows (`groups` rows per block and trip): XCD x's
// blocks (slot s of n) take rows lo + s * groups + j, stepping by n * groups inside [lo, hi) = the x-th eighth of the rows.
#define LY_EW_UR 4
#define LY_XCD_ROWS(rows_, groups_)                                                                                     \
  const bool xw_on = gridDim.x >= 8;                                                                                      \
  const long xw_x = blockIdx.x & 7, xw_slot = blockIdx.x >> 3, xw_n = ((long)gridDim.x + 7 - xw_x) >> 3;                  \
  const long xw_per = ((rows_) + 7) / 8;                                                                                  \
  const long xw_lo = xw_on ? xw_x * xw_per : 0, xw_end = xw_on ? (xw_lo + xw_per < (rows_) ? xw_lo + xw_per : (rows_)) : (rows_); \
  const long xw_begin = xw_lo + (xw_on ? xw_slot : (long)blockIdx.x) * (groups_);                                         \
  const long xw_step = (xw_on ? xw_n : (long)gridDim.x) * (groups_)

template <typename T, int ACT>
__global__ __launch_bounds__(LY_THREADS) void ly_bnact_bwd_reduce_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ u,
                                                                          int ldu, long rows, int C, const float* __restrict__ a,
                                                                          const float* __restrict__ b, double* __restrict__ sums,
                                                                          const T* __restrict__ dy2, int lddy2, int csplit, double* __restrict__ sums2) {
  // (pair form, csplit < C: channels >= csplit take their gradient from dy2 — column c - csplit — and their sums go to sums2: the two
  // BatchNorms over one stacked pre-activation tensor, C3_CA's cv1 | cv2, in ONE pass; csplit == C: one unit)
  // thread = (channel quad, row lane).  Four rows per trip with all eight loads issued before the first use, in straight-line code
  // (a load under a run-time branch makes every later s_waitcnt conservative: the first unrolled version, which still chose the vector
  // width at run time, was SLOWER than one row per trip).  Rows past the end re-read the last row and are masked.
  __shared__ f32x4 red1[LY_THREADS], red2[LY_THREADS];
  using R4 = typename LyT<T>::R4;
  const int ncv = C >> 2, tid = threadIdx.x;
  const int groups = LY_THREADS / ncv;
  const int cv = tid % ncv, j0 = tid / ncv;
  f32x4 s1 = ly_zero4(), s2 = ly_zero4();
  const bool second = 4 * cv >= csplit;
  if (j0 < groups) {
    const f32x4 av = ly_ldg4(a + 4 * cv), bv = ly_ldg4(b + 4 * cv);
    constexpr int UR = 4;
    // round 6: rows walked in per-XCD ranges, as the forward / apply passes do (LY_XCD_ROWS: XCD x's blocks take the x-th eighth of the rows) —
    // dy was just written by a tile kernel whose tiles of these rows ran on the same XCD (ly_xcd_remap): with the plain grid-stride order every
    // row was fetched past the reading block's L2
    LY_XCD_ROWS(rows, groups);
    const long stride = xw_step;
    const T* const dyb = second ? dy2 + (4 * cv - csplit) : dy + 4 * cv;      // (a selected base pointer: no load under a branch)
    const long ldd = second ? lddy2 : lddy;
    for (long r0 = xw_begin + j0; r0 < xw_end; r0 += UR * stride) {
      R4 qu[UR], qg[UR];
#pragma unroll
      for (int k = 0; k < UR; ++k) {
        const long r = r0 + k * stride < xw_end ? r0 + k * stride : xw_end - 1;
        qu[k] = ly_ldr4<T>(u + r * ldu + 4 * cv);
        qg[k] = ly_ldr4<T>(dyb + r * ldd);
      }
#pragma unroll
      for (int k = 0; k < UR; ++k) {
        const f32x4 uu = ly_r4_f32(qu[k]);
        f32x4 dv = ly_dact4<ACT>(av * uu + bv, ly_r4_f32(qg[k]));
        if (!(r0 + k * stride < xw_end)) dv = ly_zero4();
        s1 += dv;
        s2 += dv * uu;
      }
    }
  }
  red1[tid] = s1;
  red2[tid] = s2;
  __syncthreads();
  if (j0 == 0) {
    const int ch = second ? C - csplit : csplit, lc = 4 * cv - (second ? csplit : 0);        // channels of this unit, the thread's first one in it
    double* sm = (second ? sums2 : sums) + (size_t)(blockIdx.x & (LY_STATS_STRIPES - 1)) * 2 * ch;      // double accumulators: see ly_stats_flush (ly_common.hpp)
    for (int g = 1; g < groups; ++g) { s1 += red1[g * ncv + cv]; s2 += red2[g * ncv + cv]; }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      atomicAdd(sm + lc + r, (double)s1[r]);
      atomicAdd(sm + ch + lc + r, (double)s2[r]);
    }
  }
}

template <typename T, int ACT>
__global__ __launch_bounds__(LY_THREADS) void ly_bnact_bwd_apply_kernel(const T* __restrict__ dy, int lddy, const T* u, int ldu,
                                                                         long rows, int C, const float* __restrict__ a,
                                                                         const float* __restrict__ b, const float* __restrict__ alpha,
                                                                         const float* __restrict__ kappa, const float* __restrict__ lambda,
                                                                         T* du, int lddu, const T* __restrict__ dy2, int lddy2, int csplit) {
  // 16-byte accesses in both dtypes (4 fp32 / 8 bf16 channels per thread) when C allows, else 4 channels
  // (pair form, csplit < C: channels >= csplit read their gradient from dy2, column c - csplit; csplit a multiple of the vector width)
  // Round 6: thread = (channel vector, row lane), as in the reduce pass — the five coefficient vectors of the thread's channels are loaded ONCE,
  // rows advance by a pointer stride, LY_EW_UR rows are in flight per trip.  (The first form walked a flat item index: a 64-bit division per
  // item and ten coefficient loads per trip were half of the loop's instructions — ~400 issue cycles of index arithmetic per 8 elements.)
  constexpr int VW = LyT<T>::VW, NQ = VW / 4;
  using RV = typename LyT<T>::RV;
  if ((C % VW) == 0 && (ldu % VW) == 0 && (lddy % VW) == 0 && (lddu % VW) == 0 && (lddy2 % VW) == 0 && (csplit % VW) == 0 && C / VW <= LY_THREADS) {
    const int ncv = C / VW, groups = LY_THREADS / ncv;
    const int cv = threadIdx.x % ncv, j0 = threadIdx.x / ncv;
    if (j0 >= groups) return;
    const int c = VW * cv;
    f32x4 ca[NQ], cb[NQ], cal[NQ], cka[NQ], cla[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      ca[q] = ly_ldg4(a + c + 4 * q); cb[q] = ly_ldg4(b + c + 4 * q);
      cal[q] = ly_ldg4(alpha + c + 4 * q); cka[q] = ly_ldg4(kappa + c + 4 * q); cla[q] = ly_ldg4(lambda + c + 4 * q);
    }
    LY_XCD_ROWS(rows, groups);
    const T* const gb = c >= csplit ? dy2 + (c - csplit) : dy + c;        // (a selected base pointer: no load under a branch)
    const long ldg = c >= csplit ? lddy2 : lddy;
    for (long r0 = xw_begin + j0; r0 < xw_end; r0 += LY_EW_UR * xw_step) {
      RV qu[LY_EW_UR], qg[LY_EW_UR];
#pragma unroll
      for (int k = 0; k < LY_EW_UR; ++k) {
        const long r = r0 + k * xw_step < xw_end ? r0 + k * xw_step : xw_end - 1;       // rows past the end re-read the last row (straight-line loads)
        qu[k] = ly_ldrv<T>(u + r * ldu + c);
        qg[k] = ly_ldrv<T>(gb + r * ldg);
      }
#pragma unroll
      for (int k = 0; k < LY_EW_UR; ++k) {
        f32x4 uu[NQ], g[NQ];
        ly_rv_unpack(qu[k], uu);
        ly_rv_unpack(qg[k], g);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const f32x4 dv = ly_dact4<ACT>(ca[q] * uu[q] + cb[q], g[q]);
          uu[q] = cal[q] * dv + cka[q] + cla[q] * uu[q];
        }
        const long r = r0 + k * xw_step;
        if (r < xw_end) *reinterpret_cast<RV*>(du + r * lddu + c) = ly_rv_pack(uu, (RV*)nullptr);
      }
    }
    return;
  }
  const int nc4 = C >> 2;
  const long total = rows * nc4;
  LY_XCD_RANGE(total);
  for (long i = xr_begin; i < xr_end; i += xr_step) {
    const long r = i / nc4;
    const int c = 4 * (int)(i - r * nc4);
    const f32x4 uu = ly_ld4<T>(u + r * ldu + c);
    const f32x4 g = ly_ld4<T>(c >= csplit ? dy2 + r * lddy2 + (c - csplit) : dy + r * lddy + c);
    const f32x4 dv = ly_dact4<ACT>(ly_ldg4(a + c) * uu + ly_ldg4(b + c), g);
    ly_st4<T>(du + r * lddu + c, ly_ldg4(alpha + c) * dv + ly_ldg4(kappa + c) + ly_ldg4(lambda + c) * uu);
  }
}

// y = act(a[c]*u + b[c]) over an [rows, C] matrix: the second half of a train-mode conv -> BN -> act unit whose
// contraction pass stored the pre-BN value u and accumulated its statistics in the same launch
template <typename T, int ACT>
__global__ __launch_bounds__(LY_THREADS) void ly_bnact_fwd_kernel(const T* __restrict__ u, int ldu, long rows, int C, const float* __restrict__ a,
                                                                   const float* __restrict__ b, T* __restrict__ y, int ldy) {
  constexpr int VW = LyT<T>::VW, NQ = VW / 4;
  using RV = typename LyT<T>::RV;
  if ((C % VW) == 0 && (ldu % VW) == 0 && (ldy % VW) == 0 && C / VW <= LY_THREADS) {
    // thread = (channel vector, row lane): scale / shift of the thread's channels in registers, LY_EW_UR rows in flight (see ly_bnact_bwd_apply_kernel)
    const int ncv = C / VW, groups = LY_THREADS / ncv;
    const int cv = threadIdx.x % ncv, j0 = threadIdx.x / ncv;
    if (j0 >= groups) return;
    const int c = VW * cv;
    f32x4 ca[NQ], cb[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) { ca[q] = ly_ldg4(a + c + 4 * q); cb[q] = ly_ldg4(b + c + 4 * q); }
    LY_XCD_ROWS(rows, groups);
    for (long r0 = xw_begin + j0; r0 < xw_end; r0 += LY_EW_UR * xw_step) {
      RV qu[LY_EW_UR];
#pragma unroll
      for (int k = 0; k < LY_EW_UR; ++k) {
        const long r = r0 + k * xw_step < xw_end ? r0 + k * xw_step : xw_end - 1;
        qu[k] = ly_ldrv<T>(u + r * ldu + c);
      }
#pragma unroll
      for (int k = 0; k < LY_EW_UR; ++k) {
        f32x4 uu[NQ];
        ly_rv_unpack(qu[k], uu);
#pragma unroll
        for (int q = 0; q < NQ; ++q) uu[q] = ly_act4(ca[q] * uu[q] + cb[q], ACT);
        const long r = r0 + k * xw_step;
        if (r < xw_end) *reinterpret_cast<RV*>(y + r * ldy + c) = ly_rv_pack(uu, (RV*)nullptr);
      }
    }
    return;
  }
  const int nc4 = C >> 2;
  const long total = rows * nc4;
  LY_XCD_RANGE(total);
  for (long i = xr_begin; i < xr_end; i += xr_step) {
    const long r = i / nc4;
    const int c = 4 * (int)(i - r * nc4);
    const f32x4 v = ly_ldg4(a + c) * ly_ld4<T>(u + r * ldu + c) + ly_ldg4(b + c);
    ly_st4<T>(y + r * ldy + c, ly_act4(v, ACT));
  }
}

static long ly_ew_blocks(long items) {
  long b = (items + LY_THREADS * 4L - 1) / (LY_THREADS * 4L);
  return b < 1 ? 1 : b > 4096 ? 4096 : b;
}
// grid of the row-walking elementwise kernels (thread = channel vector x row lane, LY_EW_UR rows per trip): a whole number of trips per block
// (at most 8 x 256 blocks resident per XCD eighth), a multiple of 8 so that every XCD range has its blocks
static long ly_ew_row_blocks(long rows, int C, int vw) {
  const int ncv = C / vw;
  if (ncv <= 0 || ncv > LY_THREADS) return ly_ew_blocks(rows * (C >> 2));
  const long groups = LY_THREADS / ncv;
  const long trips = ((rows + 7) / 8 + groups * LY_EW_UR - 1) / (groups * LY_EW_UR);        // per XCD eighth
  const long per = (trips + 255) / 256;                                                       // trips per block
  const long bx = (trips + per - 1) / per;                                                    // blocks per eighth
  return 8 * (bx < 1 ? 1 : bx);
}

extern "C" int ly_bnact_fwd(const void* u_, int ldu, long rows, int C, const float* a, const float* b, int act, void* y_, int ldy, int dtype, void* stream) {
  LY_CHECK_DTYPE(dtype, "bnact_fwd");
  LY_CHECK(u_ && a && b && y_ && rows > 0, "bnact_fwd: null pointer");
  LY_CHECK((C & 3) == 0 && C > 0 && (ldu & 3) == 0 && (ldy & 3) == 0, "bnact_fwd: C=%d / ld must be multiples of 4", C);
  const int vw_ = dtype == LY_BF16 ? 8 : 4;
  const bool vec_ = (C % vw_) == 0 && (ldu % vw_) == 0 && (ldy % vw_) == 0 && C / vw_ <= LY_THREADS;
  const long blocks = vec_ ? ly_ew_row_blocks(rows, C, vw_) : ly_ew_blocks(rows * (C >> 2));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  LY_WITH_T(dtype, {
    const T* u = reinterpret_cast<const T*>(u_);
    T* y = reinterpret_cast<T*>(y_);
    if (act == LY_ACT_SILU) hipLaunchKernelGGL((ly_bnact_fwd_kernel<T, LY_ACT_SILU>), dim3((unsigned)blocks), dim3(LY_THREADS), 0, st, u, ldu, rows, C, a, b, y, ldy);
    else if (act == LY_ACT_RELU) hipLaunchKernelGGL((ly_bnact_fwd_kernel<T, LY_ACT_RELU>), dim3((unsigned)blocks), dim3(LY_THREADS), 0, st, u, ldu, rows, C, a, b, y, ldy);
    else hipLaunchKernelGGL((ly_bnact_fwd_kernel<T, LY_ACT_NONE>), dim3((unsigned)blocks), dim3(LY_THREADS), 0, st, u, ldu, rows, C, a, b, y, ldy);
  });
  LY_LAUNCH_CHECK();
  return 0;
}

static int bnact_bwd_reduce_launch(const void* dy_, int lddy, const void* dy2_, int lddy2, int csplit, const void* u_, int ldu, long rows, int C, const float* a,
                                   const float* b, int act, double* sums, double* sums2, int dtype, void* stream) {
  // channels per thread: 4 in both dtypes.  (8 bf16 channels = 16-byte accesses measured SLOWER here, 24.6 -> 29.0 us per launch:
  // half as many threads share a row, and this pass lives on loads in flight; the elementwise apply / forward passes gain, 21 -> 18.6
  // and 14.6 -> 12.6 us, and use 16-byte accesses.)
  const int vw = 4;
  const int groups = LY_THREADS / (C / vw);
#ifndef LY_RED_ROWS
#define LY_RED_ROWS 32            // rows per row lane and block.  Measured round 6 (bnact family per step): 16 -> 1.83, 32 -> 1.78, 64 -> 1.96, 128 -> 2.31 ms
#endif
  long blocks = (rows + groups * (long)LY_RED_ROWS - 1) / (groups * (long)LY_RED_ROWS);
  blocks = blocks < 1 ? 1 : blocks > 2048 ? 2048 : blocks;
  if (blocks >= 8) blocks = (blocks + 7) / 8 * 8;           // every XCD eighth of the rows has its blocks (LY_XCD_ROWS)
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define LY_RED(A) hipLaunchKernelGGL((ly_bnact_bwd_reduce_kernel<T, A>), dim3((unsigned)blocks), dim3(LY_THREADS), 0, st, dy, lddy, u, ldu, rows, C, a, b, sums, dy2, lddy2, csplit, sums2)
  LY_WITH_T(dtype, {
    const T* dy = reinterpret_cast<const T*>(dy_);
    const T* dy2 = reinterpret_cast<const T*>(dy2_);
    const T* u = reinterpret_cast<const T*>(u_);
    if (act == LY_ACT_SILU) LY_RED(LY_ACT_SILU);
    else if (act == LY_ACT_RELU) LY_RED(LY_ACT_RELU);
    else LY_RED(LY_ACT_NONE);
  });
#undef LY_RED
  LY_LAUNCH_CHECK();
  return 0;
}

extern "C" int ly_bnact_bwd_reduce(const void* dy_, int lddy, const void* u_, int ldu, long rows, int C, const float* a, const float* b,
                                   int act, double* sums, int dtype, void* stream) {
  LY_CHECK_DTYPE(dtype, "bnact_bwd_reduce");
  LY_CHECK(dy_ && u_ && a && b && sums && rows > 0, "bnact_bwd_reduce: null pointer");
  LY_CHECK((C & 3) == 0 && C > 0 && C <= 1024 && (lddy & 3) == 0 && (ldu & 3) == 0, "bnact_bwd_reduce: C=%d / ld must be multiples of 4", C);
  return bnact_bwd_reduce_launch(dy_, lddy, dy_, lddy, C, u_, ldu, rows, C, a, b, act, sums, sums, dtype, stream);
}

// Two conv -> BN(train) -> act units over ONE stacked pre-activation tensor u [rows, C] (channels [0, csplit) and [csplit, C): C3_CA's cv1 | cv2,
// models/common.py:1630-1636) in one pass: unit 1's gradient dy1 [rows, csplit], unit 2's dy2 [rows, C - csplit]; sums1 / sums2 as `sums` of
// ly_bnact_bwd_reduce for csplit / C - csplit channels.
extern "C" int ly_bnact_bwd_reduce_pair(const void* dy1, int lddy1, const void* dy2, int lddy2, int csplit, const void* u_, int ldu, long rows, int C,
                                        const float* a, const float* b, int act, double* sums1, double* sums2, int dtype, void* stream) {
  LY_CHECK_DTYPE(dtype, "bnact_bwd_reduce_pair");
  LY_CHECK(dy1 && dy2 && u_ && a && b && sums1 && sums2 && rows > 0, "bnact_bwd_reduce_pair: null pointer");
  LY_CHECK((C & 3) == 0 && C > 0 && C <= 1024 && csplit > 0 && csplit < C && (csplit & 3) == 0 && (lddy1 & 3) == 0 && (lddy2 & 3) == 0 && (ldu & 3) == 0,
           "bnact_bwd_reduce_pair: C=%d csplit=%d / ld must be multiples of 4", C, csplit);
  return bnact_bwd_reduce_launch(dy1, lddy1, dy2, lddy2, csplit, u_, ldu, rows, C, a, b, act, sums1, sums2, dtype, stream);
}

static int bnact_bwd_apply_launch(const void* dy_, int lddy, const void* dy2_, int lddy2, int csplit, const void* u_, int ldu, long rows, int C, const float* a,
                                  const float* b, int act, const float* alpha, const float* kappa, const float* lambda, void* du_, int lddu, int dtype, void* stream) {
  const int vw_ = dtype == LY_BF16 ? 8 : 4;
  const bool vec_ = (C % vw_) == 0 && (ldu % vw_) == 0 && (lddy % vw_) == 0 && (lddu % vw_) == 0 && (lddy2 % vw_) == 0 && (csplit % vw_) == 0 && C / vw_ <= LY_THREADS;
  const long blocks = vec_ ? ly_ew_row_blocks(rows, C, vw_) : ly_ew_blocks(rows * (C >> 2));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define LY_APP(A) hipLaunchKernelGGL((ly_bnact_bwd_apply_kernel<T, A>), dim3((unsigned)blocks), dim3(LY_THREADS), 0, st, dy, lddy, u, ldu, rows, C, a, b, alpha, kappa, lambda, du, lddu, dy2, lddy2, csplit)
  LY_WITH_T(dtype, {
    const T* dy = reinterpret_cast<const T*>(dy_);
    const T* dy2 = reinterpret_cast<const T*>(dy2_);
    const T* u = reinterpret_cast<const T*>(u_);
    T* du = reinterpret_cast<T*>(du_);
    if (act == LY_ACT_SILU) LY_APP(LY_ACT_SILU);
    else if (act == LY_ACT_RELU) LY_APP(LY_ACT_RELU);
    else LY_APP(LY_ACT_NONE);
  });
#undef LY_APP
  LY_LAUNCH_CHECK();
  return 0;
}

extern "C" int ly_bnact_bwd_apply(const void* dy_, int lddy, const void* u_, int ldu, long rows, int C, const float* a, const float* b,
                                  int act, const float* alpha, const float* kappa, const float* lambda, void* du_, int lddu, int dtype, void* stream) {
  LY_CHECK_DTYPE(dtype, "bnact_bwd_apply");
  LY_CHECK(dy_ && u_ && a && b && alpha && kappa && lambda && du_ && rows > 0, "bnact_bwd_apply: null pointer");
  LY_CHECK((C & 3) == 0 && C > 0 && (lddy & 3) == 0 && (ldu & 3) == 0 && (lddu & 3) == 0, "bnact_bwd_apply: C=%d / ld must be multiples of 4", C);
  return bnact_bwd_apply_launch(dy_, lddy, dy_, lddy, C, u_, ldu, rows, C, a, b, act, alpha, kappa, lambda, du_, lddu, dtype, stream);
}

// the pair form (see ly_bnact_bwd_reduce_pair): du [rows, C] from dy1 | dy2, coefficient vectors of C entries (both units' side by side)
extern "C" int ly_bnact_bwd_apply_pair(const void* dy1, int lddy1, const void* dy2, int lddy2, int csplit, const void* u_, int ldu, long rows, int C,
                                       const float* a, const float* b, int act, const float* alpha, const float* kappa, const float* lambda, void* du_,
                                       int lddu, int dtype, void* stream) {
  LY_CHECK_DTYPE(dtype, "bnact_bwd_apply_pair");
  LY_CHECK(dy1 && dy2 && u_ && a && b && alpha && kappa && lambda && du_ && rows > 0, "bnact_bwd_apply_pair: null pointer");
  LY_CHECK((C & 3) == 0 && C > 0 && csplit > 0 && csplit < C && (csplit & 3) == 0 && (lddy1 & 3) == 0 && (lddy2 & 3) == 0 && (ldu & 3) == 0 && (lddu & 3) == 0,
           "bnact_bwd_apply_pair: C=%d csplit=%d / ld must be multiples of 4", C, csplit);
  return bnact_bwd_apply_launch(dy1, lddy1, dy2, lddy2, csplit, u_, ldu, rows, C, a, b, act, alpha, kappa, lambda, du_, lddu, dtype, stream);
}

// -------------------------------------------------------------------------------------------------
// Weight gradient:  dw[n][tap*Cin + c] += sum_{p in pixels} du[p][n] * X[src(p, tap)][c]
//
// The contraction index is the PIXEL, which is the slow index of both operands in memory (NHWC rows), so
// the MFMA fragments are gathered straight from global memory with per-lane dword loads: lane (i, q) of a
// 16x16x32 step reads channel (tile*16 + i) of the 8 pixels p0 + 8q + j.  At fixed j the 16 lanes of a
// quarter-wave read 64 contiguous bytes of one pixel row, so every load instruction moves four full 64-byte
// segments.  The same per-lane addressing makes the forward's gathers (3x3 taps with zero padding, k=s patch
// gathers of an NHWC map or an NCHW image, the nearest-2x upsampled read) a pure address computation.
// bf16x3 products, fp32 accumulation.  Block = 64 x 64 output tile (wave = 32 x 32), grid.y splits the pixels;
// partial sums are added to dw with float atomics (dw is zeroed by the caller).
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ly_split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
  bf16x4 h0, l0, h1, l1;
  ly_split4((f32x4){v[0], v[1], v[2], v[3]}, h0, l0);
  ly_split4((f32x4){v[4], v[5], v[6], v[7]}, h1, l1);
  hi = ly_cat8(h0, h1);
  lo = ly_cat8(l0, l1);
}

template <typename T, bool ROWS>
__global__ __launch_bounds__(LY_THREADS) void ly_wgrad_kernel(const LyWgradParams P, const int tiles_k, const long chunk_px) {
  constexpr int PL = LyT<T>::PL;
  const T* const du = reinterpret_cast<const T*>(P.du);
  const T* const xin = reinterpret_cast<const T*>(P.x);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int tk = blockIdx.x % tiles_k, tn = blockIdx.x / tiles_k;
  const int n_base = tn * 64 + (wave & 1) * 32, k_base = tk * 64 + (wave >> 1) * 32;
  const long p_begin = (long)blockIdx.y * chunk_px;
  const long p_end = p_begin + chunk_px < P.M ? p_begin + chunk_px : P.M;
  const int Ktot = P.ks * P.ks * P.Cin;

  int arow[2], bcol[2], ky[2], kx[2], cc[2];
  bool aok[2], bok[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int r = n_base + 16 * t + li;
    aok[t] = r < P.N;
    arow[t] = aok[t] ? r : 0;
    const int c = k_base + 16 * t + li;
    bok[t] = c < Ktot;
    bcol[t] = bok[t] ? c : 0;
    const int tap = bcol[t] / P.Cin;
    cc[t] = bcol[t] - tap * P.Cin;
    ky[t] = tap / P.ks;
    kx[t] = tap - ky[t] * P.ks;
  }
  const int Hv = P.up2 ? 2 * P.Hin : P.Hin, Wv = P.up2 ? 2 * P.Win : P.Win;
  const float invW = 1.f / (float)P.W, invH = 1.f / (float)P.H;

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = ly_zero4();

  for (long p0 = p_begin; p0 < p_end; p0 += 32) {
    float av[2][8], bv[2][8];
    const long pf = p0 + 8 * lq;
    int n_i = 0, ho = 0, wo = 0;
    if (!ROWS) {
      const int g = (int)(pf < P.M ? pf : P.M - 1);
      const int row = ly_fdiv(g, P.W, invW);
      wo = g - row * P.W;
      n_i = ly_fdiv(row, P.H, invH);
      ho = row - n_i * P.H;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long p = pf + j;
      const bool pok = p < p_end;
      const long pc = pok ? p : p_end - 1;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float v = ly_ld1<T>(du + pc * P.lddu + arow[t]);
        av[t][j] = (pok && aok[t]) ? v : 0.f;
      }
      if (ROWS) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const float v = ly_ld1<T>(xin + pc * P.ldx + bcol[t]);
          bv[t][j] = (pok && bok[t]) ? v : 0.f;
        }
      } else {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          int hi = ho * P.stride + ky[t] - P.pad, wi = wo * P.stride + kx[t] - P.pad;
          const bool ok = pok && bok[t] && hi >= 0 && hi < Hv && wi >= 0 && wi < Wv;
          if (P.up2) { hi >>= 1; wi >>= 1; }
          long off = P.nchw ? (((long)n_i * P.Cin + cc[t]) * P.Hin + hi) * P.Win + wi
                            : (((long)n_i * P.Hin + hi) * P.Win + wi) * P.ldx + cc[t];
          const float v = ly_ld1<T>(xin + (ok ? off : 0));
          bv[t][j] = ok ? v : 0.f;
        }
        if (++wo == P.W) { wo = 0; if (++ho == P.H) { ho = 0; ++n_i; } }
      }
    }
    bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      ly_split8(av[t], ah[t], al[t]);
      ly_split8(bv[t], bh[t], bl[t]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = ly_mfmapp<PL>(ah[i], al[i], bh[j], bl[j], acc[i][j]);
  }

  // D: lane (i = column, q) holds rows 4q + r
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = k_base + 16 * j + li;
      if (col >= Ktot) continue;
      const int tap = col / P.Cin, cch = col - tap * P.Cin;
      if (cch >= P.c_valid) continue;
      const long cidx = (long)tap * P.dw_ts + (long)cch * P.dw_cs;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = n_base + 16 * i + 4 * lq + r;
        if (row < P.n_valid) atomicAdd(P.dw + (long)row * P.lddw + cidx, acc[i][j][r]);
      }
    }
}


// -------------------------------------------------------------------------------------------------
// Tiled weight gradient (the fast path: N, Cin, lddu, ldx multiples of 4, NHWC input).
// Block = BN x BK output tile, 4 waves as 2 x 2, wave tile (BN/2) x (BK/2) = 64 accumulator registers.
// Per step of P pixels (32, or 64 for skinny outputs) the block stages the [P px][BN] slab of du and the gathered [P px][BK]
// slab of x ONCE:
// a thread owns one channel quad x one group of 8 consecutive pixels, i.e. 8 coalesced float4 row loads (a wave reads
// 512 contiguous bytes per pixel row), transposes them in registers and writes, per channel, the 8 pixels as ONE
// 16-byte bf16 vector into the hi / lo planes  plane[channel][P px].  An MFMA fragment (lane (i, q):
// channel i, pixels 8q..8q+7) is then a single conflict-free ds_read_b128, for both operands.  Global loads of step
// s+1 are in flight while step s is contracted (register prefetch, two LDS buffers, one barrier per step).
// Re-reads drop from (N/64 + K/64) to (N/BN + K/BK) passes over the two tensors.
// -------------------------------------------------------------------------------------------------
// SLAB: 0 = atomic flush only, 1 = slab when the pointer is given (run-time), 2 = slab always (the grouped launch's instantiation: a second flush
// path compiled into it de-pipelined its pixel loop)
template <typename T, int BN, int BK, int P, bool ROWS, bool PRO, int SLAB = 1, bool OCT = false>
__device__ __forceinline__ void ly_wgrad_tiled_body(const LyWgradParams& Q, const int tile_idx, const int chunk_idx, const int tiles_k, const long chunk_px,
                                                    float* __restrict__ const slab, const int tiles) {
  using R4 = typename LyT<T>::R4;
  constexpr int PL = LyT<T>::PL;
  const T* const du = reinterpret_cast<const T*>(Q.du);
  const T* const xin = reinterpret_cast<const T*>(Q.x);
  constexpr int PG = P / 8;                             // 8-pixel groups per step
  constexpr int TASKS = (BN / 4 + BK / 4) * PG;         // (channel quad, pixel group) pairs per step
  constexpr int TPT = (TASKS + LY_THREADS - 1) / LY_THREADS;
  constexpr int NI = BN / 32, NJ = BK / 32;             // MFMA tiles per wave along n / k
  // LDS image of a step: [plane hi | lo][pixel group g of 8][row slot][8 px bf16 = 16 bytes].  A fragment read (ds_read_b128: lane (i, q) takes
  // row i, pixel group 4 ks + q) walks 16 consecutive row slots per pixel group, and a commit (4 channel rows x 16 bytes per thread, threads on
  // consecutive channel quads) must not put rows 4 cq + e of neighbouring threads 64 bytes apart — so row r sits at slot (r & 3) (NR/4 + 4) +
  // (r >> 2): for a fixed e consecutive threads write consecutive slots, and rows 0 .. 15 of a fragment land on 16 different slots modulo 16
  // ((r & 3) * 4 + (r >> 2): a 4 x 4 transpose).  Both conflict-free; the round-5 image ([row][P px] with 16 bytes of row padding) read with one
  // 2-way slot per lane group and wrote 4-way: SQ_LDS_BANK_CONFLICT = 24 % of the grouped kernel's cycles (profiles/r05_train_bf16_pmc_wait.txt).
  constexpr int NR = BN + BK;
  constexpr int PLANE = (NR + 16) * 16;                 // bytes per pixel group
  constexpr int BUF = PL * PG * PLANE;
  // OC (round 6; bf16 storage, plain rows, every width / stride a multiple of 8): a staging task is a channel OCTET x 8 pixels = eight 16-byte
  // row loads (a wave reads four 256-byte row pieces per instruction) instead of a quad x 8 pixels = eight 8-byte loads: half the vector-memory
  // instructions per step, and 16-byte accesses (the 8-byte form streams at 0.54-0.70 of their rate: MI355X guide, HBM section; the kernels sat
  // at 2.6-3.5 TB/s).  The 8 x 8 transpose is 32 v_perm_b32 per task (the quads' 4 x 8: 16) — the same count per byte.  Row r then sits at slot
  // (r & 7) (NR/8 + 2) + (r >> 3): for a fixed channel-in-octet consecutive threads write consecutive slots, and rows 0 .. 15 of a fragment land
  // on the 16 slots (r & 7) * 2 + (r >> 3) modulo 16 (NR/8 + 2 = 2 mod 16 for NR = 64, 96, 128, 160, 192, 256).
  constexpr bool OC = OCT && LyT<T>::BF;                // (gathered inputs too: an octet lies inside one tap when Cin is a multiple of 8)
  static_assert(!OC || ((NR / 8 + 2) % 16 == 2 || (NR / 8 + 2) % 16 == 10 || (NR / 8 + 2) % 16 == 6 || (NR / 8 + 2) % 16 == 14), "octet row map: fragment rows must hit 16 slots");
  auto rowoff = [](int r) -> int { return OC ? ((r & 7) * (NR / 8 + 2) + (r >> 3)) * 16 : ((r & 3) * (NR / 4 + 4) + (r >> 2)) * 16; };
  constexpr int ISTEP = OC ? 32 : 64;                   // bytes between the row slots of rows r and r + 16
  constexpr int TASKS8 = (BN / 8 + BK / 8) * PG, TPT8 = (TASKS8 + LY_THREADS - 1) / LY_THREADS;
  extern __shared__ f32x4 ly_smem4[];
  char* const lds = reinterpret_cast<char*>(ly_smem4);
  constexpr bool SB = PL * P >= 128;       // 128 bf16 / 64 fp32 pixels per step: ONE LDS buffer (see the loop below)
  float* const sab = reinterpret_cast<float*>(lds + (SB ? 1 : 2) * BUF);      // [scale BK | shift BK] of the optional x prologue
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int tk = tile_idx % tiles_k, tn = tile_idx / tiles_k;
  const int n0 = tn * BN, k0 = tk * BK;
  const long p_begin = (long)chunk_idx * chunk_px;
  const long p_end = p_begin + chunk_px < Q.M ? p_begin + chunk_px : Q.M;
  const int Ktot = Q.ks * Q.ks * Q.Cin;
  // x prologue max(x*scale + shift, 0) (LyWgradParams.x_scale): the block's BK columns of both vectors staged once; applied while the x slab
  // is transposed into LDS (masked pixels meet zero rows of du, so relu(shift) there is harmless; masked columns get scale = shift = 0)
  // PRO: the instantiation that can; a group may mix problems with and without vectors (pro is uniform over the block).  Straight-line on
  // purpose — a load under a branch makes every later s_waitcnt conservative (the first version, `if (pro) { loads }`, doubled the kernel's
  // time): all threads load (a problem without vectors reads the head of x as a harmless stand-in), the select happens on values.
  const bool pro = ROWS && PRO && Q.x_scale != nullptr;
  if constexpr (ROWS && PRO) {
    const float* const sp = pro ? Q.x_scale : reinterpret_cast<const float*>(Q.x);      // (4*Cin bytes of x exist: M >= 2 rows)
    const float* const hp = pro ? Q.x_shift : reinterpret_cast<const float*>(Q.x);
    const int i = tid & (BK - 1);
    const int k = k0 + i < Ktot ? k0 + i : 0;
    const float sv = sp[k], hv = hp[k];
    sab[i] = k0 + i < Ktot ? sv : 0.f;
    sab[BK + i] = k0 + i < Ktot ? hv : 0.f;
    __syncthreads();
  }

  // ---- per-task constants ---------------------------------------------------------------------
  bool t_isA[TPT], t_ok[TPT];
  int t_row[TPT], t_g[TPT], t_c[TPT], t_ky[TPT], t_kx[TPT];
#pragma unroll
  for (int u = 0; u < TPT; ++u) {
    const int t = tid + u * LY_THREADS;
    const bool live = t < TASKS;
    t_isA[u] = t < (BN / 4) * PG;
    const int tt = t_isA[u] ? t : t - (BN / 4) * PG;
    const int quads = t_isA[u] ? BN / 4 : BK / 4;
    const int cq = tt % quads;
    t_g[u] = (tt / quads) % PG;
    t_row[u] = (t_isA[u] ? 0 : BN) + 4 * cq;            // first of the 4 LDS rows this task writes
    if (t_isA[u]) {
      t_c[u] = n0 + 4 * cq;
      t_ok[u] = live && t_c[u] < Q.N;
      t_ky[u] = t_kx[u] = 0;
    } else {
      const int col = k0 + 4 * cq;
      t_ok[u] = live && col < Ktot;
      const int cc = t_ok[u] ? col : 0;
      const int tap = cc / Q.Cin;
      t_c[u] = cc - tap * Q.Cin;
      t_ky[u] = tap / Q.ks;
      t_kx[u] = tap - t_ky[u] * Q.ks;
    }
    if (!live) t_row[u] = -1;
  }
  const int Hv = Q.up2 ? 2 * Q.Hin : Q.Hin, Wv = Q.up2 ? 2 * Q.Win : Q.Win;
  const float invW = 1.f / (float)Q.W, invH = 1.f / (float)Q.H;

  // DEFER (the grouped slab instantiation): the loads of a step are pure — masked pixels are zeroed when the registers are committed to LDS,
  // not right behind each load.  With the select behind the load hipcc's scheduler, in THIS instantiation, sank every load next to its
  // select: one `s_waitcnt vmcnt(0)` per load, the pixel loop ran at one memory round trip per 8 bytes (53 -> 108 us per launch).
  constexpr bool DEFER = SLAB == 2 && ROWS;      // (the same change in the single-problem instantiations: 32.6 -> 37.7 us — only where the scheduler had gone wrong)
  long pre_p0 = 0;
  R4 pre[TPT][8];
  auto prefetch = [&](long p0) {
    pre_p0 = p0;
#pragma unroll
    for (int u = 0; u < TPT; ++u) {
      const long pf = p0 + 8 * t_g[u];
      int n_i = 0, ho = 0, wo = 0;
      if (!ROWS && !t_isA[u]) {
        const int g = (int)(pf < Q.M ? pf : Q.M - 1);
        const int row = ly_fdiv(g, Q.W, invW);
        wo = g - row * Q.W;
        n_i = ly_fdiv(row, Q.H, invH);
        ho = row - n_i * Q.H;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const long p = pf + j;
        bool ok = t_ok[u] && p < p_end;
        const T* src;
        if (t_isA[u]) {
          src = du + (ok ? p : p_begin) * Q.lddu + t_c[u];
        } else if (ROWS) {
          src = xin + (ok ? p : p_begin) * Q.ldx + t_c[u];
        } else {
          int hi = ho * Q.stride + t_ky[u] - Q.pad, wi = wo * Q.stride + t_kx[u] - Q.pad;
          ok = ok && hi >= 0 && hi < Hv && wi >= 0 && wi < Wv;
          if (Q.up2) { hi >>= 1; wi >>= 1; }
          src = xin + (ok ? (((long)n_i * Q.Hin + hi) * Q.Win + wi) * Q.ldx : 0) + t_c[u];
          if (++wo == Q.W) { wo = 0; if (++ho == Q.H) { ho = 0; ++n_i; } }
        }
        R4 v = ly_ldr4<T>(src);
        if constexpr (!DEFER) {
          if (!ok) ly_zero_raw(v);
        }
        pre[u][j] = v;
      }
    }
  };
  auto commit = [&](int buf) {
    char* base = lds + buf * BUF;
#pragma unroll
    for (int u = 0; u < TPT; ++u) {
      if (t_row[u] < 0) continue;
      if constexpr (DEFER) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (!(t_ok[u] && pre_p0 + 8 * t_g[u] + j < p_end)) ly_zero_raw(pre[u][j]);
      }
      const bool pb = pro && !t_isA[u];
      f32x4 sa = ly_zero4(), sh = ly_zero4();
      if constexpr (ROWS && PRO) {
        const int lc = t_isA[u] ? 0 : t_row[u] - BN;
        sa = *reinterpret_cast<const f32x4*>(sab + lc);
        sh = *reinterpret_cast<const f32x4*>(sab + BK + lc);
      }
      if constexpr (PL == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v[8] = {pre[u][0][e], pre[u][1][e], pre[u][2][e], pre[u][3][e], pre[u][4][e], pre[u][5][e], pre[u][6][e], pre[u][7][e]};
          if constexpr (ROWS && PRO) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = pb ? fmaxf(v[j] * sa[e] + sh[e], 0.f) : v[j];
          }
          bf16x8 hi, lo;
          ly_split8(v, hi, lo);
          char* d = base + t_g[u] * PLANE + t_row[u] * 4 + e * ((NR / 4 + 4) * 16);      // rowoff(t_row + e), t_row a multiple of 4
          *reinterpret_cast<bf16x8*>(d) = hi;
          *reinterpret_cast<bf16x8*>(d + PG * PLANE) = lo;
        }
      } else {
        // bf16 storage: a pure 8 x 4 transpose of 16-bit values (channel e of the 8 pixels becomes one 16-byte row piece)
        bf16x4 q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = __builtin_bit_cast(bf16x4, pre[u][j]);
        if constexpr (ROWS && PRO) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            f32x4 v = ly_cvt4(q[j]) * sa + sh;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            const bf16x4 r = ly_cvtb4(v);
            q[j] = pb ? r : q[j];
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bf16x8 row = {q[0][e], q[1][e], q[2][e], q[3][e], q[4][e], q[5][e], q[6][e], q[7][e]};
          *reinterpret_cast<bf16x8*>(base + t_g[u] * PLANE + t_row[u] * 4 + e * ((NR / 4 + 4) * 16)) = row;      // rowoff(t_row + e), t_row a multiple of 4
        }
      }
    }
  };

  // ---- the octet form of the same two steps -------------------------------------------------------
  bool o_isA[OC ? TPT8 : 1], o_ok[OC ? TPT8 : 1];
  int o_row[OC ? TPT8 : 1], o_g[OC ? TPT8 : 1], o_c[OC ? TPT8 : 1], o_ky[OC ? TPT8 : 1], o_kx[OC ? TPT8 : 1];
  ly_u32x4 pre8[OC ? TPT8 : 1][OC ? 8 : 1];
  if constexpr (OC) {
#pragma unroll
    for (int u = 0; u < TPT8; ++u) {
      const int t = tid + u * LY_THREADS;
      const bool live = t < TASKS8;
      o_isA[u] = t < (BN / 8) * PG;
      const int tt = o_isA[u] ? t : t - (BN / 8) * PG;
      const int octs = o_isA[u] ? BN / 8 : BK / 8;
      const int co = tt % octs;
      o_g[u] = (tt / octs) % PG;
      o_row[u] = live ? (o_isA[u] ? 0 : BN) + 8 * co : -1;
      o_c[u] = (o_isA[u] ? n0 : k0) + 8 * co;
      o_ok[u] = live && o_c[u] < (o_isA[u] ? Q.N : Ktot);
      o_ky[u] = o_kx[u] = 0;
      if (!ROWS && !o_isA[u]) {                              // gathered input: column -> (tap, channel inside the tap)
        const int cc = o_ok[u] ? o_c[u] : 0;
        const int tap = cc / Q.Cin;
        o_c[u] = cc - tap * Q.Cin;
        o_ky[u] = tap / Q.ks;
        o_kx[u] = tap - o_ky[u] * Q.ks;
      }
    }
  }
  auto prefetch8 = [&](long p0) {
    pre_p0 = p0;
#pragma unroll
    for (int u = 0; u < (OC ? TPT8 : 0); ++u) {
      const long pf = p0 + 8 * o_g[u];
      int n_i = 0, ho = 0, wo = 0;
      if (!ROWS && !o_isA[u]) {
        const int g = (int)(pf < Q.M ? pf : Q.M - 1);
        const int row = ly_fdiv(g, Q.W, invW);
        wo = g - row * Q.W;
        n_i = ly_fdiv(row, Q.H, invH);
        ho = row - n_i * Q.H;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const long p = pf + j;
        bool ok = o_ok[u] && p < p_end;
        const T* src;
        if (o_isA[u]) {
          src = du + (ok ? p : p_begin) * Q.lddu + (ok ? o_c[u] : 0);
        } else if (ROWS) {
          src = xin + (ok ? p : p_begin) * Q.ldx + (ok ? o_c[u] : 0);
        } else {
          int hi = ho * Q.stride + o_ky[u] - Q.pad, wi = wo * Q.stride + o_kx[u] - Q.pad;
          ok = ok && hi >= 0 && hi < Hv && wi >= 0 && wi < Wv;
          if (Q.up2) { hi >>= 1; wi >>= 1; }
          src = xin + (ok ? (((long)n_i * Q.Hin + hi) * Q.Win + wi) * Q.ldx + o_c[u] : 0);
          if (++wo == Q.W) { wo = 0; if (++ho == Q.H) { ho = 0; ++n_i; } }
        }
        ly_u32x4 v = *reinterpret_cast<const ly_u32x4*>(src);
        if constexpr (!DEFER) {
          if (!ok) v = (ly_u32x4){0u, 0u, 0u, 0u};
        }
        pre8[u][j] = v;
      }
    }
  };
  auto commit8 = [&](int buf) {
    char* base = lds + buf * BUF;
#pragma unroll
    for (int u = 0; u < (OC ? TPT8 : 0); ++u) {
      if (o_row[u] < 0) continue;
      if constexpr (DEFER) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (!(o_ok[u] && pre_p0 + 8 * o_g[u] + j < p_end)) pre8[u][j] = (ly_u32x4){0u, 0u, 0u, 0u};
      }
      if constexpr (PRO) {
        if (pro && !o_isA[u]) {
          const int lc = o_row[u] - BN;
          const f32x4 sa0 = *reinterpret_cast<const f32x4*>(sab + lc), sa1 = *reinterpret_cast<const f32x4*>(sab + lc + 4);
          const f32x4 sh0 = *reinterpret_cast<const f32x4*>(sab + BK + lc), sh1 = *reinterpret_cast<const f32x4*>(sab + BK + lc + 4);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            f32x4 q[2];
            ly_rv_unpack(pre8[u][j], q);
            q[0] = q[0] * sa0 + sh0;
            q[1] = q[1] * sa1 + sh1;
#pragma unroll
            for (int e = 0; e < 4; ++e) { q[0][e] = fmaxf(q[0][e], 0.f); q[1][e] = fmaxf(q[1][e], 0.f); }
            pre8[u][j] = ly_rv_pack(q, (ly_u32x4*)nullptr);
          }
        }
      }
      // 8 pixels x 8 channels of 16 bits -> channel e: its 8 pixels as one 16-byte row piece (dword k = pixels 2k, 2k + 1)
      char* const d0 = base + o_g[u] * PLANE + (o_row[u] >> 3) * 16;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        ly_u32x4 row;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          row[k] = __builtin_amdgcn_perm(pre8[u][2 * k + 1][e >> 1], pre8[u][2 * k][e >> 1], (e & 1) ? 0x07060302u : 0x05040100u);
        *reinterpret_cast<ly_u32x4*>(d0 + e * ((NR / 8 + 2) * 16)) = row;            // rowoff(o_row + e), o_row a multiple of 8
      }
    }
  };
  auto prefetch_any = [&](long p0) { if constexpr (OC) prefetch8(p0); else prefetch(p0); };
  auto commit_any = [&](int buf) { if constexpr (OC) commit8(buf); else commit(buf); };

  f32x4 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = ly_zero4();
  const int wn = (wave & 1) * (BN / 2), wk = BN + (wave >> 1) * (BK / 2);
  // the lane's fragment addresses: everything else is an immediate (wn, wk multiples of 16: rows r + 16 m sit m * ISTEP bytes further)
  const int rd_a = lq * PLANE + rowoff(li) + (wn / 16) * ISTEP, rd_b = lq * PLANE + rowoff(li) + (wk / 16) * ISTEP;

  // SB: ONE LDS buffer (two would leave one block per CU); the loads of the next step are in flight during the contraction, the buffer is
  prefetch_any(p_begin);                   // rewritten between two barriers
  commit_any(0);
  __syncthreads();
  int buf = 0;
  for (long p0 = p_begin; p0 < p_end; p0 += P) {
    const bool more = p0 + P < p_end;
    if (more) prefetch_any(p0 + P);
    const char* base = lds + buf * BUF;
#pragma unroll
    for (int ks = 0; ks < P / 32; ++ks) {
      bf16x8 ah[NI], al[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const char* r = base + rd_a + ks * (4 * PLANE) + i * ISTEP;     // row wn + 16 i + li: wn and 16 i are multiples of 16 (4 / 2 slots of 16 bytes per 16 rows)
        ah[i] = *reinterpret_cast<const bf16x8*>(r);
        al[i] = PL == 2 ? *reinterpret_cast<const bf16x8*>(r + (PL - 1) * PG * PLANE) : ah[i];
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const char* r = base + rd_b + ks * (4 * PLANE) + j * ISTEP;
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(r);
        const bf16x8 bl = PL == 2 ? *reinterpret_cast<const bf16x8*>(r + (PL - 1) * PG * PLANE) : bh;
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i][j] = ly_mfmapp<PL>(ah[i], al[i], bh, bl, acc[i][j]);
      }
    }
    if constexpr (SB) {
      __syncthreads();
      if (more) commit_any(0);
      __syncthreads();
    } else {
      if (more) commit_any(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }

  if (SLAB == 2 || (SLAB == 1 && slab)) {
    // the chunk's tile [BN][BK] into its own slab with plain stores (16 lanes = 64 contiguous bytes); ly_wgrad_combine folds the chunks in a
    // fixed order: no float atomics (the atomic flush of 512 x 64 KB was a quarter of a launch), bit-reproducible dw
    float* const sl = slab + ((long)chunk_idx * tiles + tile_idx) * (BN * BK);
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int lc = (wave >> 1) * (BK / 2) + 16 * j + li;
        const int col = k0 + lc;
        if (col >= Ktot) continue;
        const int tap = col / Q.Cin, cch = col - tap * Q.Cin;
        if (cch >= Q.c_valid) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int lr = wn + 16 * i + 4 * lq + r;
          if (n0 + lr < Q.n_valid) sl[lr * BK + lc] = acc[i][j][r];
        }
      }
    return;
  }
  if constexpr (SLAB != 2) {
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = k0 + (wave >> 1) * (BK / 2) + 16 * j + li;
      if (col >= Ktot) continue;
      const int tap = col / Q.Cin, cch = col - tap * Q.Cin;
      if (cch >= Q.c_valid) continue;
      const long cidx = (long)tap * Q.dw_ts + (long)cch * Q.dw_cs;     // (tap, channel) -> position inside a dw row: the weight's own layout
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = n0 + wn + 16 * i + 4 * lq + r;
        if (row < Q.n_valid) atomicAdd(Q.dw + (long)row * Q.lddw + cidx, acc[i][j][r]);
      }
    }
  }
}

// dw[n][tap][c] += sum over chunks of slab[chunk][tile][row][col] in a fixed order, one writer per element (the second launch of a tiled
// weight gradient whose chunks left their tiles in LyWgradParams.ws).  Block = 64 slab columns x 16 chunk lanes.
__device__ __forceinline__ void ly_wgrad_combine_body(const LyWgradParams& P, const float* __restrict__ slab, const int chunks, const int tiles_k,
                                                      const int tiles, const int BN, const int BK, const long blk, const int rls) {
  // rls row lanes (block = 64 columns x rls lanes, 64 rls threads) walk the chunks: 16 for the single-problem launches (hundreds of chunks
  // per element), 4 when a group's problems have a few dozen chunks each (1024 threads with two loads apiece were all overhead: 13 us)
  __shared__ float red[16][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const long E = (long)tiles * BN * BK;
  const long e = blk * 64 + cl;
  const int tile = (int)(e / (BN * BK));
  const int rem = (int)(e - (long)tile * (BN * BK));
  const int lr = rem / BK, lc = rem - lr * BK;
  const int tk = tile % tiles_k, tn = tile / tiles_k;
  const int row = tn * BN + lr, col = tk * BK + lc;
  const int tap = col / P.Cin, cch = col - tap * P.Cin;
  const bool live = e < E && row < P.n_valid && col < P.ks * P.ks * P.Cin && cch < P.c_valid;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (live) {
    // four loads in flight per lane (the launch is a handful of dependent memory round trips: 8.6 -> ~6 us with two more in flight);
    // the summation order is fixed, whatever the block count of the producing launch
    const float* p = slab + e;
    int c = rl;
    for (; c + 7 * rls < chunks; c += 8 * rls) {        // eight loads fenced ahead of the adds (the scheduler sinks them back otherwise)
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = p[(long)(c + rls * k) * E];
      __builtin_amdgcn_sched_barrier(0);
      a0 += v[0]; a1 += v[1]; a2 += v[2]; a3 += v[3];
      a0 += v[4]; a1 += v[5]; a2 += v[6]; a3 += v[7];
    }
    for (; c + 3 * rls < chunks; c += 4 * rls) {
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = p[(long)(c + rls * k) * E];
      __builtin_amdgcn_sched_barrier(0);
      a0 += v[0]; a1 += v[1]; a2 += v[2]; a3 += v[3];
    }
    if (c < chunks) a0 += p[(long)c * E];
    if (c + rls < chunks) a1 += p[(long)(c + rls) * E];
    if (c + 2 * rls < chunks) a2 += p[(long)(c + 2 * rls) * E];
  }
  red[rl][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (rl == 0 && live) {
    float sacc = 0.f;
    for (int i = 0; i < rls; ++i) sacc += red[i][cl];
    float* d = P.dw + (long)row * P.lddw + (long)tap * P.dw_ts + (long)cch * P.dw_cs;
    *d += sacc;
  }
}
__global__ __launch_bounds__(1024) void ly_wgrad_combine_kernel(const LyWgradParams P, const float* __restrict__ slab, const int chunks, const int tiles_k,
                                                               const int tiles, const int BN, const int BK, const int rls) {
  ly_wgrad_combine_body(P, slab, chunks, tiles_k, tiles, BN, BK, (long)blockIdx.x, rls);
}
static void wgrad_combine_launch(const LyWgradParams& P, const float* slab, long chunks, int tiles_k, long tiles, int BN, int BK, hipStream_t st) {
  const long E = tiles * BN * BK;
  const int rls = chunks <= 192 ? 4 : 16;                 // row lanes: see ly_wgrad_combine_body
  hipLaunchKernelGGL(ly_wgrad_combine_kernel, dim3((unsigned)((E + 63) / 64)), dim3(64 * rls), 0, st, P, slab, (int)chunks, tiles_k, (int)tiles, BN, BK, rls);
}

template <typename T, int BN, int BK, int P, bool ROWS, bool PRO = false, bool OCT = false>
__global__ __launch_bounds__(LY_THREADS) __attribute__((amdgpu_waves_per_eu(BN + BK > 160 ? 2 : 3))) void ly_wgrad_tiled_kernel(const LyWgradParams P_, const int tiles_k, const long chunk_px, float* const slab) {
  // (no XCD-aware re-ordering here: tile / chunk computed from the remapped id with a run-time division made hipcc treat them as per-lane values —
  // the 128 x 128 rows form took 3x as long, 45 -> 146 us per launch; forced back to scalars (readfirstlane) it fetched a third less from HBM
  // but still ran 45 -> 50 us, the other forms +-1 us.  The grouped kernel, ly_wgrad3 and the other tile kernels keep the remap.)
  ly_wgrad_tiled_body<T, BN, BK, P, ROWS, PRO, 1, OCT>(P_, (int)blockIdx.x, (int)blockIdx.y, tiles_k, chunk_px, slab, (int)gridDim.x);
}

// Several independent weight gradients of ONE tile class in one launch (ly_wgrad_group): the problems share the ~512 blocks, so each block
// walks a longer pixel chunk — the fixed cost of a block (first-load latency, the atomic flush of its 64 KB tile) is paid half / a third as
// often per pixel, and the launch fills the chip where a lone small problem leaves a ragged second wave.
#define LY_WGRAD_GROUP_MAX 4
struct LyWgradGroupArgs {
  LyWgradParams p[LY_WGRAD_GROUP_MAX];
  long chunk_px[LY_WGRAD_GROUP_MAX];
  float* slab[LY_WGRAD_GROUP_MAX];       // per problem: [chunks][tiles][BN * BK] partial tiles (the slab form), else unused
  int tiles_k[LY_WGRAD_GROUP_MAX], tiles[LY_WGRAD_GROUP_MAX], blk0[LY_WGRAD_GROUP_MAX + 1];
  int chunks[LY_WGRAD_GROUP_MAX], cblk0[LY_WGRAD_GROUP_MAX + 1];      // combine launch: chunks per problem, first combine block per problem
  int n;
};
template <typename T, int BN, int BK, int P, bool ROWS, bool PRO = false, bool GSLAB = false, bool OCT = false>
__global__ __launch_bounds__(LY_THREADS) __attribute__((amdgpu_waves_per_eu(BN + BK > 160 ? 2 : 3))) void ly_wgrad_tiled_group_kernel(const LyWgradGroupArgs G) {
  const int bid = ly_xcd_remap((int)blockIdx.x, (int)gridDim.x);            // (see ly_wgrad_tiled_kernel)
  int g = 0;
#pragma unroll
  for (int i = 1; i < LY_WGRAD_GROUP_MAX; ++i)
    if (i < G.n && bid >= G.blk0[i]) g = i;
  g = __builtin_amdgcn_readfirstlane(g);
  const int local = bid - G.blk0[g];
  const int tiles = G.tiles[g];
  // (no slab path here: with it compiled in, every wait of this instantiation's pixel loop became `s_waitcnt vmcnt(0)` — 53 -> 107 us per launch;
  // a grouped launch keeps the atomic flush, whose cost the longer pixel runs per block already halve)
  // GSLAB: every block leaves its tile in the problem's slab (plain stores) and ly_wgrad_combine_group_kernel folds the chunks in index order —
  // no float atomics: the grouped weight gradients are the same bits in every run
  ly_wgrad_tiled_body<T, BN, BK, P, ROWS, PRO, GSLAB ? 2 : 0, OCT>(G.p[g], local % tiles, local / tiles, G.tiles_k[g], G.chunk_px[g], GSLAB ? G.slab[g] : nullptr, tiles);
}
// the combine launches of a group's problems as ONE launch
__global__ __launch_bounds__(1024) void ly_wgrad_combine_group_kernel(const LyWgradGroupArgs G, const int BN, const int BK, const int rls) {
  int g = 0;
#pragma unroll
  for (int i = 1; i < LY_WGRAD_GROUP_MAX; ++i)
    if (i < G.n && (int)blockIdx.x >= G.cblk0[i]) g = i;
  g = __builtin_amdgcn_readfirstlane(g);
  ly_wgrad_combine_body(G.p[g], G.slab[g], G.chunks[g], G.tiles_k[g], G.tiles[g], BN, BK, (long)((int)blockIdx.x - G.cblk0[g]), rls);
}

// blocks a weight-gradient launch aims for (development builds, `make DEVEL=1`, read LY_WG_BLOCKS / LY_WG_GROUP_BLOCKS)
static long wg_target_blocks(bool group) {
#ifdef LY_DEVEL
  static long t[2] = {0, 0};
  if (!t[group]) {
    const char* e = getenv(group ? "LY_WG_GROUP_BLOCKS" : "LY_WG_BLOCKS");
    t[group] = e && atol(e) > 0 ? atol(e) : 512;
  }
  return t[group];
#else
  (void)group;
  return 512;
#endif
}

// the octet staging of ly_wgrad_tiled_body applies: bf16 storage, both operands' widths, strides and addresses in whole 16-byte vectors
template <typename T>
static bool wgrad_octets(const LyWgradParams& Q) {
  if constexpr (!LyT<T>::BF) return false;
#ifdef LY_DEVEL
  static int off = -1;
  if (off < 0) { const char* e = getenv("LY_WG_NO_OCT"); off = e && atoi(e) ? 1 : 0; }
  if (off) return false;
#endif
  return (Q.N & 7) == 0 && (Q.Cin & 7) == 0 && (Q.lddu & 7) == 0 && (Q.ldx & 7) == 0 && ((uintptr_t)Q.du & 15) == 0 && ((uintptr_t)Q.x & 15) == 0;
}

template <typename T, int BN, int BK, int P>
static int launch_wgrad_tiled(const LyWgradParams& Q, bool rows, hipStream_t st) {
  const int Ktot = Q.ks * Q.ks * Q.Cin;
  const int tiles_n = (Q.N + BN - 1) / BN, tiles_k = (Ktot + BK - 1) / BK;
  const long tiles = (long)tiles_n * tiles_k;
  // ~512 blocks: every chunk adds its whole tile to dw with float atomics (a quarter of all wgrad time at 1024 blocks; 256 blocks
  // leave CUs idle: 3.80 / 3.56 / 4.36 ms per bs=64 bf16 step for 1024 / 512 / 256)
  long chunks = (wg_target_blocks(false) + tiles - 1) / tiles;
  const long max_chunks = (Q.M + 511) / 512;
  if (chunks > max_chunks) chunks = max_chunks;
  if (chunks < 1) chunks = 1;
  long chunk_px = (Q.M + chunks - 1) / chunks;
  chunk_px = (chunk_px + P - 1) / P * P;
  chunks = (Q.M + chunk_px - 1) / chunk_px;
  LY_CHECK(chunks < 65536, "wgrad: too many pixel chunks");
  const size_t lds = (LyT<T>::PL * P >= 128 ? 1 : 2) * (size_t)LyT<T>::PL * (P / 8) * (BN + BK + 16) * 16 + 2 * BK * sizeof(float);
  const dim3 grid((unsigned)tiles, (unsigned)chunks);
  const long need = chunks * tiles * (long)(BN * BK);
  float* const slab = (chunks > 1 && Q.ws && need <= Q.ws_floats) ? Q.ws : nullptr;
  bool oct_done = false;
  if constexpr (LyT<T>::BF) {
   if (rows && wgrad_octets<T>(Q)) {
    // bf16 rows with every width a multiple of 8: 16-byte staging loads (the octet form of ly_wgrad_tiled_body)
    oct_done = true;
    if (Q.x_scale) {
      static bool attr = false;
      if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ly_wgrad_tiled_kernel<T, BN, BK, P, true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
      hipLaunchKernelGGL((ly_wgrad_tiled_kernel<T, BN, BK, P, true, true, true>), grid, dim3(LY_THREADS), lds, st, Q, tiles_k, chunk_px, slab);
    } else {
      static bool attr = false;
      if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ly_wgrad_tiled_kernel<T, BN, BK, P, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
      hipLaunchKernelGGL((ly_wgrad_tiled_kernel<T, BN, BK, P, true, false, true>), grid, dim3(LY_THREADS), lds, st, Q, tiles_k, chunk_px, slab);
    }
   }
  }
  if constexpr (LyT<T>::BF) {
    if (!oct_done && !rows && wgrad_octets<T>(Q)) {
      oct_done = true;
      static bool attr = false;
      if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ly_wgrad_tiled_kernel<T, BN, BK, P, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
      hipLaunchKernelGGL((ly_wgrad_tiled_kernel<T, BN, BK, P, false, false, true>), grid, dim3(LY_THREADS), lds, st, Q, tiles_k, chunk_px, slab);
    }
  }
  if (oct_done) {
  } else if (rows && Q.x_scale) {
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ly_wgrad_tiled_kernel<T, BN, BK, P, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    hipLaunchKernelGGL((ly_wgrad_tiled_kernel<T, BN, BK, P, true, true>), grid, dim3(LY_THREADS), lds, st, Q, tiles_k, chunk_px, slab);
  } else if (rows) {
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ly_wgrad_tiled_kernel<T, BN, BK, P, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    hipLaunchKernelGGL((ly_wgrad_tiled_kernel<T, BN, BK, P, true>), grid, dim3(LY_THREADS), lds, st, Q, tiles_k, chunk_px, slab);
  } else {
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ly_wgrad_tiled_kernel<T, BN, BK, P, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    hipLaunchKernelGGL((ly_wgrad_tiled_kernel<T, BN, BK, P, false>), grid, dim3(LY_THREADS), lds, st, Q, tiles_k, chunk_px, slab);
  }
  if (slab) wgrad_combine_launch(Q, slab, chunks, tiles_k, tiles, BN, BK, st);
  LY_LAUNCH_CHECK();
  return 0;
}

template <typename T>
static int wgrad_dispatch(const LyWgradParams& P, void* stream);
bool ly_wgrad3_ok(const LyWgradParams& P);                         // ly_wgrad3.hip
int ly_wgrad3_launch(const LyWgradParams& P, hipStream_t st);
static int g_ly_wgrad3 = 1;
extern "C" int ly_tune_wgrad3(int on) {                            // development switch: 0 sends 3x3 problems back to the generic tiled kernel
  const int was = g_ly_wgrad3;
  g_ly_wgrad3 = on;
  return was;
}

extern "C" int ly_wgrad(const LyWgradParams* p, void* stream) {
  LY_CHECK(p, "wgrad: null params");
  LY_CHECK_DTYPE(p->dtype, "wgrad");
  return p->dtype == LY_BF16 ? wgrad_dispatch<__bf16>(*p, stream) : wgrad_dispatch<float>(*p, stream);
}

// true when P takes the plain-rows 128 x 128 tile of the fast path (the class ly_wgrad_group batches)
template <typename T>
static bool wgrad_is_rows128(const LyWgradParams& P) {
  if (!(P.du && P.x && P.dw) || !(P.M > 0 && P.H > 0 && P.W > 0 && P.N > 64 && P.Cin > 0) || P.M >= (1L << 24) || P.M % ((long)P.H * P.W) != 0) return false;
  if (!(P.ks == 1 && P.stride == 1 && P.pad == 0 && !P.nchw && !P.up2 && P.Hin == P.H && P.Win == P.W)) return false;
  if (!(P.n_valid > 0 && P.n_valid <= P.N && P.c_valid > 0 && P.c_valid <= P.Cin && P.dw_ts > 0 && P.dw_cs > 0)) return false;
  if ((long)P.lddw < (long)(P.c_valid - 1) * P.dw_cs + 1) return false;
  return (P.N & 3) == 0 && (P.Cin & 3) == 0 && (P.lddu & 3) == 0 && (P.ldx & 3) == 0 && ((uintptr_t)P.du & (4 * sizeof(T) - 1)) == 0 &&
         ((uintptr_t)P.x & (4 * sizeof(T) - 1)) == 0;
}

template <typename T>
static int wgrad_group_launch(const LyWgradParams* arr, int n, hipStream_t st) {
  constexpr int PX = LyT<T>::BF ? 128 : 64, BN = 128, BK = 128;
  LyWgradGroupArgs G;
  long tiles_total = 0;
  for (int g = 0; g < n; ++g) {
    G.p[g] = arr[g];
    G.tiles_k[g] = (arr[g].Cin + BK - 1) / BK;
    G.tiles[g] = ((arr[g].N + BN - 1) / BN) * G.tiles_k[g];
    tiles_total += G.tiles[g];
  }
  G.n = n;
  G.blk0[0] = 0;
  for (int g = 0; g < n; ++g) {
    long chunks = (wg_target_blocks(true) + tiles_total - 1) / tiles_total;               // the group shares the ~512 blocks (launch_wgrad_tiled's policy)
    const long max_chunks = (arr[g].M + 511) / 512;
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks < 1) chunks = 1;
    long chunk_px = (arr[g].M + chunks - 1) / chunks;
    chunk_px = (chunk_px + PX - 1) / PX * PX;
    chunks = (arr[g].M + chunk_px - 1) / chunk_px;
    G.chunk_px[g] = chunk_px;
    G.chunks[g] = (int)chunks;
    G.blk0[g + 1] = G.blk0[g] + (int)(chunks * G.tiles[g]);
  }
  for (int g = n; g < LY_WGRAD_GROUP_MAX; ++g) { G.tiles_k[g] = G.tiles[g] = 1; G.chunk_px[g] = PX; G.blk0[g + 1] = G.blk0[n]; G.p[g] = arr[0]; G.chunks[g] = 0; G.slab[g] = nullptr; }
  // slab form when the caller's scratch (LyWgradParams.ws of the first problem) holds every block's tile
  bool gslab = arr[0].ws != nullptr && (long)G.blk0[n] * (BN * BK) <= arr[0].ws_floats;
  G.cblk0[0] = 0;
  for (int g = 0; g < n; ++g) {
    G.slab[g] = gslab ? arr[0].ws + (long)G.blk0[g] * (BN * BK) : nullptr;
    G.cblk0[g + 1] = G.cblk0[g] + (int)(((long)G.tiles[g] * BN * BK + 63) / 64);
  }
  for (int g = n; g < LY_WGRAD_GROUP_MAX; ++g) G.cblk0[g + 1] = G.cblk0[n];
  const size_t lds = (LyT<T>::PL * PX >= 128 ? 1 : 2) * (size_t)LyT<T>::PL * (PX / 8) * (BN + BK + 16) * 16 + 2 * BK * sizeof(float);
  bool any_pro = false;
  for (int g = 0; g < n; ++g) any_pro = any_pro || arr[g].x_scale != nullptr;
  bool oct = true;
  for (int g = 0; g < n; ++g) oct = oct && wgrad_octets<T>(arr[g]);
  if (gslab) {
    bool launched = false;
    if constexpr (LyT<T>::BF) {
      if (oct && any_pro) {
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ly_wgrad_tiled_group_kernel<T, BN, BK, PX, true, true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
        hipLaunchKernelGGL((ly_wgrad_tiled_group_kernel<T, BN, BK, PX, true, true, true, true>), dim3((unsigned)G.blk0[n]), dim3(LY_THREADS), lds, st, G);
        launched = true;
      } else if (oct) {
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ly_wgrad_tiled_group_kernel<T, BN, BK, PX, true, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
        hipLaunchKernelGGL((ly_wgrad_tiled_group_kernel<T, BN, BK, PX, true, false, true, true>), dim3((unsigned)G.blk0[n]), dim3(LY_THREADS), lds, st, G);
        launched = true;
      }
    }
    if (launched) {
    } else if (any_pro) {
      static bool attr = false;
      if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ly_wgrad_tiled_group_kernel<T, BN, BK, PX, true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
      hipLaunchKernelGGL((ly_wgrad_tiled_group_kernel<T, BN, BK, PX, true, true, true>), dim3((unsigned)G.blk0[n]), dim3(LY_THREADS), lds, st, G);
    } else {
      static bool attr = false;
      if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ly_wgrad_tiled_group_kernel<T, BN, BK, PX, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
      hipLaunchKernelGGL((ly_wgrad_tiled_group_kernel<T, BN, BK, PX, true, false, true>), dim3((unsigned)G.blk0[n]), dim3(LY_THREADS), lds, st, G);
    }
    int cmax = 0;
    for (int g = 0; g < n; ++g) cmax = G.chunks[g] > cmax ? G.chunks[g] : cmax;
    const int rls = cmax <= 64 ? 4 : 16;
    hipLaunchKernelGGL(ly_wgrad_combine_group_kernel, dim3((unsigned)G.cblk0[n]), dim3(64 * rls), 0, st, G, BN, BK, rls);
    LY_LAUNCH_CHECK();
    return 0;
  }
  if (any_pro) {
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ly_wgrad_tiled_group_kernel<T, BN, BK, PX, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    hipLaunchKernelGGL((ly_wgrad_tiled_group_kernel<T, BN, BK, PX, true, true>), dim3((unsigned)G.blk0[n]), dim3(LY_THREADS), lds, st, G);
  } else {
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ly_wgrad_tiled_group_kernel<T, BN, BK, PX, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    hipLaunchKernelGGL((ly_wgrad_tiled_group_kernel<T, BN, BK, PX, true>), dim3((unsigned)G.blk0[n]), dim3(LY_THREADS), lds, st, G);
  }
  LY_LAUNCH_CHECK();
  return 0;
}

extern "C" int ly_wgrad_group(const LyWgradParams* arr, int n, void* stream) {
  LY_CHECK(arr && n > 0, "wgrad_group: bad arguments");
  bool same = n >= 2 && n <= LY_WGRAD_GROUP_MAX;
  for (int g = 0; g < n && same; ++g) {
    LY_CHECK_DTYPE(arr[g].dtype, "wgrad_group");
    same = arr[g].dtype == arr[0].dtype && (arr[g].dtype == LY_BF16 ? wgrad_is_rows128<__bf16>(arr[g]) : wgrad_is_rows128<float>(arr[g]));
  }
  if (!same) {                                                        // anything else: one launch per problem, as ly_wgrad would
    for (int g = 0; g < n; ++g) {
      const int rc = ly_wgrad(arr + g, stream);
      if (rc) return rc;
    }
    return 0;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  return arr[0].dtype == LY_BF16 ? wgrad_group_launch<__bf16>(arr, n, st) : wgrad_group_launch<float>(arr, n, st);
}

template <typename T>
static int wgrad_dispatch(const LyWgradParams& P, void* stream) {
  LY_CHECK(P.du && P.x && P.dw, "wgrad: null pointer");
  LY_CHECK(P.M > 0 && P.H > 0 && P.W > 0 && P.N > 0 && P.Cin > 0 && P.ks > 0 && P.stride > 0, "wgrad: bad sizes");
  LY_CHECK(P.M < (1L << 24), "wgrad: M=%ld pixels exceeds the 2^24 limit of the fast index path", P.M);
  LY_CHECK(P.M % ((long)P.H * P.W) == 0, "wgrad: M is not a whole number of images");
  const int Ktot = P.ks * P.ks * P.Cin;
  LY_CHECK(P.n_valid > 0 && P.n_valid <= P.N && P.c_valid > 0 && P.c_valid <= P.Cin && P.dw_ts > 0 && P.dw_cs > 0, "wgrad: bad dw layout");
  LY_CHECK((long)P.lddw >= (long)(P.ks * P.ks - 1) * P.dw_ts + (long)(P.c_valid - 1) * P.dw_cs + 1, "wgrad: lddw=%d does not cover a dw row", P.lddw);
  const bool rows = P.ks == 1 && P.stride == 1 && P.pad == 0 && !P.nchw && !P.up2;
  if (rows) LY_CHECK(P.Hin == P.H && P.Win == P.W, "wgrad: 1x1 gather needs Hin == H, Win == W");
  LY_CHECK((P.x_scale == nullptr) == (P.x_shift == nullptr), "wgrad: x_scale and x_shift come together");
  const bool tiled_ok = !P.nchw && (P.N & 3) == 0 && (P.Cin & 3) == 0 && (P.lddu & 3) == 0 && (P.ldx & 3) == 0 && ((uintptr_t)P.du & (4 * sizeof(T) - 1)) == 0 &&
                        ((uintptr_t)P.x & (4 * sizeof(T) - 1)) == 0;
  if (P.x_scale) LY_CHECK(rows && tiled_ok, "wgrad: the x prologue is built for plain-row 1x1 problems with vector-friendly widths");
  if (g_ly_wgrad3 && ly_wgrad3_ok(P)) return ly_wgrad3_launch(P, reinterpret_cast<hipStream_t>(stream));      // 3x3 / stride 1, bf16: halo-tile kernel (ly_wgrad3.hip)
  if (!P.nchw && (P.N & 3) == 0 && (P.Cin & 3) == 0 && (P.lddu & 3) == 0 && (P.ldx & 3) == 0 && ((uintptr_t)P.du & (4 * sizeof(T) - 1)) == 0 && ((uintptr_t)P.x & (4 * sizeof(T) - 1)) == 0) {
    hipStream_t st2 = reinterpret_cast<hipStream_t>(stream);
    // One step of a block is one memory round trip, so what matters for the skinny shapes is how many blocks a CU holds and how many
    // bytes each has in flight: the 64 x 256 tile's 92 KB of LDS meant ONE block (N=8 K=72 M=1.6M: 690 -> 280 us, N=64 K=576: 538 -> 342 us
    // with the tiles below; all wgrad launches of a bs=64 step 7.8 -> 6.5 ms in round 1).
    // pixels per step: 128 (bf16) / 64 (fp32) = the same 272-byte LDS rows in both, held in ONE buffer (the rows of the next step are in
    // flight in registers during the contraction and the buffer is rewritten between two barriers).  Against the double-buffered
    // 64 / 32-pixel step this doubles the bytes in flight per block at the same blocks per CU: all wgrad launches of a bs=64 bf16 step
    // 3.56 -> 3.28 ms.
    constexpr int PX = LyT<T>::BF ? 128 : 64;
    if (P.N <= 64 && Ktot <= 64) return launch_wgrad_tiled<T, 64, 64, PX>(P, rows, st2);
    if (P.N <= 32) return launch_wgrad_tiled<T, 32, 128, PX>(P, rows, st2);
    if (P.N <= 64) return launch_wgrad_tiled<T, 64, 128, PX>(P, rows, st2);
    return launch_wgrad_tiled<T, 128, 128, PX>(P, rows, st2);
  }
  const int tiles_n = (P.N + 63) / 64, tiles_k = (Ktot + 63) / 64;
  const long tiles = (long)tiles_n * tiles_k;
  long chunks = (2048 + tiles - 1) / tiles;
  const long max_chunks = (P.M + 255) / 256;
  if (chunks > max_chunks) chunks = max_chunks;
  if (chunks < 1) chunks = 1;
  long chunk_px = (P.M + chunks - 1) / chunks;
  chunk_px = (chunk_px + 31) / 32 * 32;
  chunks = (P.M + chunk_px - 1) / chunk_px;
  LY_CHECK(chunks < 65536, "wgrad: too many pixel chunks");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)tiles, (unsigned)chunks);
  if (rows) hipLaunchKernelGGL((ly_wgrad_kernel<T, true>), grid, dim3(LY_THREADS), 0, st, P, tiles_k, chunk_px);
  else hipLaunchKernelGGL((ly_wgrad_kernel<T, false>), grid, dim3(LY_THREADS), 0, st, P, tiles_k, chunk_px);
  LY_LAUNCH_CHECK();
  return 0;
}

// -------------------------------------------------------------------------------------------------
// Adjoint of the nearest-2x upsampled read (nn.Upsample(2,'nearest'), models/LEAD-YOLO.yaml neck):
//   dsrc[n, h, w, :] = sum of the four dst pixels (2h+{0,1}, 2w+{0,1})
// -------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(LY_THREADS) void ly_up2_bwd_kernel(const T* __restrict__ d, int ldd, int n_img, int Hs, int Ws, int C,
                                                                T* __restrict__ o, int ldo) {
  const int nc4 = C >> 2;
  const long total = (long)n_img * Hs * Ws * nc4;
  for (long i = (long)blockIdx.x * LY_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * LY_THREADS) {
    const long pix = i / nc4;
    const int c = 4 * (int)(i - pix * nc4);
    const long row = pix / Ws;
    const int w = (int)(pix - row * Ws);
    const long n = row / Hs;
    const int h = (int)(row - n * Hs);
    const T* s = d + (((n * 2 * Hs + 2 * h) * 2L * Ws) + 2 * w) * ldd + c;
    const f32x4 v = ly_ld4<T>(s) + ly_ld4<T>(s + ldd) + ly_ld4<T>(s + 2L * Ws * ldd) + ly_ld4<T>(s + 2L * Ws * ldd + ldd);
    ly_st4<T>(o + pix * ldo + c, v);
  }
}

extern "C" int ly_up2_bwd(const void* d, int ldd, int n_img, int Hs, int Ws, int C, void* out, int ldo, int dtype, void* stream) {
  LY_CHECK_DTYPE(dtype, "up2_bwd");
  LY_CHECK(d && out && n_img > 0 && Hs > 0 && Ws > 0 && (C & 3) == 0 && (ldd & 3) == 0 && (ldo & 3) == 0, "up2_bwd: bad arguments");
  LY_WITH_T(dtype, hipLaunchKernelGGL(ly_up2_bwd_kernel<T>, dim3((unsigned)ly_ew_blocks((long)n_img * Hs * Ws * (C >> 2))), dim3(LY_THREADS), 0,
                                      reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const T*>(d), ldd, n_img, Hs, Ws, C, reinterpret_cast<T*>(out), ldo));
  LY_LAUNCH_CHECK();
  return 0;
}

// -------------------------------------------------------------------------------------------------
// Adjoint of the k = s = 2 patch gather (PatchMerging_FasterNet, models/common.py:1555-1561): the dgrad GEMM
// produces g[m = (n, ho, wo)][(ky, kx, c)]; scatter it to dx[n, 2ho+ky, 2wo+kx, c] (every input pixel belongs
// to exactly one patch).
// -------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(LY_THREADS) void ly_unpatch_kernel(const T* __restrict__ g, int n_img, int Ho, int Wo, int C, int ks,
                                                                T* __restrict__ dx) {
  const int nc4 = C >> 2;
  const int kc = ks * ks * nc4;
  const long total = (long)n_img * Ho * Wo * kc;
  for (long i = (long)blockIdx.x * LY_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * LY_THREADS) {
    const long m = i / kc;
    int r = (int)(i - m * kc);
    const int tap = r / nc4;
    const int c = 4 * (r - tap * nc4);
    const int ky = tap / ks, kx = tap - ky * ks;
    const long row = m / Wo;
    const int wo = (int)(m - row * Wo);
    const long n = row / Ho;
    const int ho = (int)(row - n * Ho);
    const long dst = ((n * Ho * ks + (long)ho * ks + ky) * ((long)Wo * ks) + (long)wo * ks + kx) * C + c;
    ly_st4<T>(dx + dst, ly_ld4<T>(g + m * ((long)ks * ks * C) + (long)tap * C + c));
  }
}

// -------------------------------------------------------------------------------------------------
// dst[c] (+)= sum_r src[r][c]: the partial-row buffers of the backward (per-block generate weight gradients, per-group conv weight
// gradients, per-chunk d_rfa slabs, moment slices) folded deterministically — fixed order, no atomics.  Block = 64 columns x 16 row lanes;
// a row lane walks rows lane, lane + 16, ... with four loads in flight.  (Was `tensor.sum(0)`: ATen's multi-block reduction keeps scratch
// state of its own, and a captured instance returned wrong sums once the same reduction had also run eagerly in the process.)
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void ly_sum_rows_kernel(const float* __restrict__ src, const long R, const long C, const long ld,
                                                          float* __restrict__ dst, const int accumulate, const int rls) {
  __shared__ float red[16][64];                            // rls row lanes (4 for R <= 192, else 16: see ly_wgrad_combine_body)
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const long c = (long)blockIdx.x * 64 + cl;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < C) {
    const float* p = src + c;
    long r = rl;
    for (; r + 7 * rls < R; r += 8 * rls) {             // eight loads fenced ahead of the adds
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = p[(r + rls * k) * ld];
      __builtin_amdgcn_sched_barrier(0);
      a0 += v[0]; a1 += v[1]; a2 += v[2]; a3 += v[3];
      a0 += v[4]; a1 += v[5]; a2 += v[6]; a3 += v[7];
    }
    for (; r + 3 * rls < R; r += 4 * rls) {
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = p[(r + rls * k) * ld];
      __builtin_amdgcn_sched_barrier(0);
      a0 += v[0]; a1 += v[1]; a2 += v[2]; a3 += v[3];
    }
    for (; r < R; r += rls) a0 += p[r * ld];
  }
  red[rl][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (rl == 0 && c < C) {
    float s = 0.f;
    for (int i = 0; i < rls; ++i) s += red[i][cl];
    dst[c] = accumulate ? dst[c] + s : s;
  }
}

extern "C" int ly_sum_rows(const float* src, long R, long C, long ld, float* dst, int accumulate, void* stream) {
  LY_CHECK(src && dst && R > 0 && C > 0 && ld >= C, "sum_rows: bad arguments");
  LY_CHECK((C + 63) / 64 < (1L << 31), "sum_rows: too many columns");
  const int rls = R <= 192 ? 4 : 16;
  hipLaunchKernelGGL(ly_sum_rows_kernel, dim3((unsigned)((C + 63) / 64)), dim3(64 * rls), 0, reinterpret_cast<hipStream_t>(stream), src, R, C, ld, dst,
                     accumulate, rls);
  LY_LAUNCH_CHECK();
  return 0;
}

// column sums of a small [R][C] matrix of doubles (the stripes of a double statistics accumulator), rows added in index order
__global__ __launch_bounds__(LY_THREADS) void ly_sum_rows_f64_kernel(const double* __restrict__ src, const int R, const int C, double* __restrict__ dst) {
  const int c = blockIdx.x * LY_THREADS + threadIdx.x;
  if (c >= C) return;
  double a = 0.0;
  for (int r = 0; r < R; ++r) a += src[(long)r * C + c];
  dst[c] = a;
}
// dst[i] += (float)src[i] for up to LY_F64_ADD_MAX small vectors in one launch (the double scratches of ly_gacc, ly_common.hpp)
__global__ __launch_bounds__(LY_THREADS) void ly_f64_add_kernel(const LyF64AddTable T) {
  const int e = blockIdx.y;
  const int i = blockIdx.x * LY_THREADS + threadIdx.x;
  if (e < T.count && i < T.n[e]) T.dst[e][i] += (float)T.src[e][i];
}
extern "C" int ly_f64_add(const LyF64AddTable* t, void* stream) {
  LY_CHECK(t && t->count > 0 && t->count <= LY_F64_ADD_MAX, "f64_add: bad table");
  int nmax = 0;
  for (int e = 0; e < t->count; ++e) {
    LY_CHECK(t->src[e] && t->dst[e] && t->n[e] > 0, "f64_add: entry %d is empty", e);
    nmax = t->n[e] > nmax ? t->n[e] : nmax;
  }
  hipLaunchKernelGGL(ly_f64_add_kernel, dim3((unsigned)((nmax + LY_THREADS - 1) / LY_THREADS), (unsigned)t->count), dim3(LY_THREADS), 0,
                     reinterpret_cast<hipStream_t>(stream), *t);
  LY_LAUNCH_CHECK();
  return 0;
}

extern "C" int ly_sum_rows_f64(const double* src, int R, int C, double* dst, void* stream) {
  LY_CHECK(src && dst && R > 0 && C > 0, "sum_rows_f64: bad arguments");
  hipLaunchKernelGGL(ly_sum_rows_f64_kernel, dim3((unsigned)((C + LY_THREADS - 1) / LY_THREADS)), dim3(LY_THREADS), 0, reinterpret_cast<hipStream_t>(stream), src,
                     R, C, dst);
  LY_LAUNCH_CHECK();
  return 0;
}

extern "C" int ly_unpatch(const void* g, int n_img, int Ho, int Wo, int C, int ks, void* dx, int dtype, void* stream) {
  LY_CHECK_DTYPE(dtype, "unpatch");
  LY_CHECK(g && dx && n_img > 0 && Ho > 0 && Wo > 0 && ks > 0 && (C & 3) == 0, "unpatch: bad arguments");
  LY_WITH_T(dtype, hipLaunchKernelGGL(ly_unpatch_kernel<T>, dim3((unsigned)ly_ew_blocks((long)n_img * Ho * Wo * ks * ks * (C >> 2))), dim3(LY_THREADS), 0,
                                      reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const T*>(g), n_img, Ho, Wo, C, ks, reinterpret_cast<T*>(dx)));
  LY_LAUNCH_CHECK();
  return 0;
}

// -------------------------------------------------------------------------------------------------
// Space-to-depth of the uint8 NCHW image for PatchEmbed's weight gradient (models/common.py:1537-1550, k = s = 4): rows[m][(c, ky, kx)]
// = the integer pixel values as T (exact in bf16), so that ly_wgrad contracts plain rows; the caller applies 1/255 to the small dw.
// Block = 64 patches of one patch row: coalesced 4-byte loads (the 4 kx of a patch) staged through LDS, 16-byte row stores
// (was: a permuted torch copy + a cast, 250 us at bs=64 640x640; one pass at the copy rate now).
// -------------------------------------------------------------------------------------------------
#define LY_P4_WO 64
template <typename T>
__global__ __launch_bounds__(LY_THREADS) void ly_patch4_rows_u8_kernel(const unsigned char* __restrict__ img, int C, int H, int W, T* __restrict__ rows) {
  extern __shared__ f32x4 ly_p4_smem[];
  T* const tile = reinterpret_cast<T*>(ly_p4_smem);
  const int Wo = W >> 2, Ho = H >> 2;
  const int K = 16 * C, KP = K + 8;                       // padded LDS row
  const int wo0 = blockIdx.x * LY_P4_WO;
  const int n = blockIdx.y / Ho, ho = blockIdx.y - n * Ho;
  const int nw = Wo - wo0 < LY_P4_WO ? Wo - wo0 : LY_P4_WO;
  for (int i = threadIdx.x; i < 4 * C * LY_P4_WO; i += LY_THREADS) {
    const int wl = i & (LY_P4_WO - 1), r = i / LY_P4_WO;  // r = c * 4 + ky
    const int c = r >> 2, ky = r & 3;
    const int wc = wl < nw ? wl : nw - 1;
    const unsigned v = *reinterpret_cast<const unsigned*>(img + (((long)n * C + c) * H + 4 * ho + ky) * W + 4 * (wo0 + wc));
    T* d = tile + wl * KP + 4 * r;
    d[0] = (T)(float)(v & 255u); d[1] = (T)(float)((v >> 8) & 255u); d[2] = (T)(float)((v >> 16) & 255u); d[3] = (T)(float)(v >> 24);
  }
  __syncthreads();
  constexpr int VE = 16 / sizeof(T);                       // elements per 16-byte store
  const int vpr = K / VE;
  T* const out = rows + ((long)blockIdx.y * Wo + wo0) * K;
  for (int j = threadIdx.x; j < nw * vpr; j += LY_THREADS) {
    const int wl = j / vpr, q = j - wl * vpr;
    *reinterpret_cast<f32x4*>(out + (long)wl * K + q * VE) = *reinterpret_cast<const f32x4*>(tile + wl * KP + q * VE);
  }
}

extern "C" int ly_patch4_rows_u8(const unsigned char* img, int n_img, int C, int H, int W, void* rows, int dtype, void* stream) {
  LY_CHECK_DTYPE(dtype, "patch4_rows_u8");
  LY_CHECK(img && rows && n_img > 0 && C > 0 && C <= 16 && H > 0 && W > 0 && (H & 3) == 0 && (W & 3) == 0, "patch4_rows_u8: bad arguments");
  LY_CHECK(((uintptr_t)img & 3) == 0 && ((uintptr_t)rows & 15) == 0, "patch4_rows_u8: unaligned pointer");
  LY_CHECK((long)n_img * (H >> 2) < 65536, "patch4_rows_u8: too many patch rows");
  const dim3 grid((unsigned)(((W >> 2) + LY_P4_WO - 1) / LY_P4_WO), (unsigned)(n_img * (H >> 2)));
  LY_WITH_T(dtype, hipLaunchKernelGGL(ly_patch4_rows_u8_kernel<T>, grid, dim3(LY_THREADS), (size_t)LY_P4_WO * (16 * C + 8) * sizeof(T),
                                      reinterpret_cast<hipStream_t>(stream), img, C, H, W, reinterpret_cast<T*>(rows)));
  LY_LAUNCH_CHECK();
  return 0;
}

// -------------------------------------------------------------------------------------------------
// CoordAtt (models/common.py:1595-1609) backward pieces.
//   gate:  out = x * a_h[n,h,:] * a_w[n,w,:]
//          dx = dout*a_h*a_w,  da_h[n,h,c] = sum_w dout*x*a_w,  da_w[n,w,c] = sum_h dout*x*a_h
//   pools: pool[n, 0:H, c] = mean_w x,  pool[n, H:H+W, c] = mean_h x
//          dx[n,h,w,c] = gp[n,h,c]/W + gp[n,H+w,c]/H
// One block per (image, row h): da_h is reduced in the block, da_w (zeroed by the caller) by float atomics.
// -------------------------------------------------------------------------------------------------
// One block per (image, band of RB rows): da_h is reduced per row in the block; the da_w contributions of the band's rows are
// summed in registers (a thread owns a fixed set of (w, channel quad) pairs) and added with ONE float atomic per element and
// band instead of one per element and row.
#define LY_CAG_RB 8
#define LY_CAG_MAXW 8        // (w, c4) pairs per thread: ceil(W / groups) <= 8
template <typename T>
__global__ __launch_bounds__(LY_THREADS) void ly_coordatt_gate_bwd_kernel(const T* __restrict__ dout, int ldd, const T* __restrict__ x,
                                                                           int ldx, int H, int W, int C, const float* __restrict__ a_h,
                                                                           const float* __restrict__ a_w, T* __restrict__ dx, int lddx,
                                                                           float* __restrict__ da_h, float* __restrict__ da_w, int bands, int slabs, long n_img) {
  __shared__ f32x4 red[LY_THREADS];
  const int nc4 = C >> 2, tid = threadIdx.x;
  const int groups = LY_THREADS / nc4;
  const int c4 = tid % nc4, g0 = tid / nc4;
  const int slab = blockIdx.x % slabs;                       // column slab of groups * LY_CAG_MAXW columns
  const int bb = blockIdx.x / slabs;
  const long n = bb / bands;
  const int band = bb - (int)n * bands;
  const int w0 = slab * groups * LY_CAG_MAXW;
  const int h_lo = band * LY_CAG_RB, h_hi = h_lo + LY_CAG_RB < H ? h_lo + LY_CAG_RB : H;
  f32x4 accw[LY_CAG_MAXW], aw[LY_CAG_MAXW];
  bool okw[LY_CAG_MAXW];
  int wcl[LY_CAG_MAXW];
  // the thread's columns are the same for every row of the band: their a_w factors are loaded once; per row all 16 row loads are issued from
  // clamped addresses before the first use (they sat under a per-column branch: one exposed round trip per column)
  const bool gok = g0 < groups;
#pragma unroll
  for (int i = 0; i < LY_CAG_MAXW; ++i) {
    accw[i] = ly_zero4();
    const int w = w0 + g0 + i * groups;
    okw[i] = gok && w < W;
    wcl[i] = okw[i] ? w : (w0 < W ? w0 : 0);
    aw[i] = ly_ldg4(a_w + (n * W + wcl[i]) * C + 4 * (gok ? c4 : 0));
  }
  using R4 = typename LyT<T>::R4;
  for (int h = h_lo; h < h_hi; ++h) {
    const long nh = n * H + h;
    f32x4 sh = ly_zero4();
    {
      const int c4c = gok ? c4 : 0;
      const f32x4 ah = ly_ldg4(a_h + nh * C + 4 * c4c);
      R4 dr[LY_CAG_MAXW], xr[LY_CAG_MAXW];
#pragma unroll
      for (int i = 0; i < LY_CAG_MAXW; ++i) {
        const long row = nh * W + wcl[i];
        dr[i] = ly_ldr4<T>(dout + row * ldd + 4 * c4c);
        xr[i] = ly_ldr4<T>(x + row * ldx + 4 * c4c);
      }
#pragma unroll
      for (int i = 0; i < LY_CAG_MAXW; ++i) {
        if (okw[i]) {
          const long row = nh * W + wcl[i];
          const f32x4 d = ly_r4_f32(dr[i]), xv = ly_r4_f32(xr[i]);
          ly_st4<T>(dx + row * lddx + 4 * c4, d * ah * aw[i]);
          const f32x4 t = d * xv;
          sh += t * aw[i];
          accw[i] += t * ah;
        }
      }
    }
    __syncthreads();
    red[tid] = sh;
    __syncthreads();
    if (g0 == 0) {
      for (int g = 1; g < groups; ++g) sh += red[g * nc4 + c4];
      // one PARTIAL per row, slab and channel, stored (not added): da_h_part[slab][n][h][c].  (Float atomics into one [n][h][c] array
      // summed in arrival order and seeded run-to-run differences of dx; the caller folds the slabs in index order: ly_sum_rows)
      ly_stg4(da_h + ((long)slab * n_img * H + nh) * C + 4 * c4, sh);
    }
  }
  if (g0 < groups) {
#pragma unroll
    for (int i = 0; i < LY_CAG_MAXW; ++i) {
      const int w = w0 + g0 + i * groups;
      if (w < W) {
        ly_stg4(da_w + ((long)band * n_img * W + n * W + w) * C + 4 * c4, accw[i]);      // da_w_part[band][n][w][c]
      }
    }
  }
}

extern "C" int ly_coordatt_gate_bwd(const void* dout, int ldd, const void* x, int ldx, int n_img, int H, int W, int C, const float* a_h,
                                    const float* a_w, void* dx, int lddx, float* da_h, float* da_w, int bands_in, int slabs_in, int dtype, void* stream) {
  LY_CHECK_DTYPE(dtype, "coordatt_gate_bwd");
  LY_CHECK(dout && x && a_h && a_w && dx && da_h && da_w, "coordatt_gate_bwd: null pointer");
  LY_CHECK((C & 3) == 0 && C <= 1024 && (ldd & 3) == 0 && (ldx & 3) == 0 && (lddx & 3) == 0, "coordatt_gate_bwd: C / ld must be multiples of 4");
  const int groups = LY_THREADS / (C >> 2);
  const int slabs = (W + groups * LY_CAG_MAXW - 1) / (groups * LY_CAG_MAXW);
  const int bands = (H + LY_CAG_RB - 1) / LY_CAG_RB;
  LY_CHECK(bands_in == bands && slabs_in == slabs, "coordatt_gate_bwd: the partial buffers are da_h [%d slabs][n][H][C] and da_w [%d bands][n][W][C] (got %d, %d)",
           slabs, bands, slabs_in, bands_in);
  LY_WITH_T(dtype, hipLaunchKernelGGL(ly_coordatt_gate_bwd_kernel<T>, dim3((unsigned)(n_img * bands * slabs)), dim3(LY_THREADS), 0, reinterpret_cast<hipStream_t>(stream),
                                      reinterpret_cast<const T*>(dout), ldd, reinterpret_cast<const T*>(x), ldx, H, W, C, a_h, a_w, reinterpret_cast<T*>(dx), lddx, da_h, da_w,
                                      bands, slabs, (long)n_img));
  LY_LAUNCH_CHECK();
  return 0;
}

template <typename T, bool ACC>
__global__ __launch_bounds__(LY_THREADS) void ly_pool_hw_bwd_kernel(const float* __restrict__ gp, int n_img, int H, int W, int C,
                                                                    T* __restrict__ dx, int lddx) {
  // thread = (16-byte channel vector, column lane): a block walks (image, row) pairs, the row's own pool gradient is loaded once per pair and
  // columns advance by a stride — no per-item index arithmetic (the flat form below pays three 64-bit divisions per 8 bytes)
  constexpr int VW = LyT<T>::VW, NQ = VW / 4;
  using RV = typename LyT<T>::RV;
  if ((C % VW) == 0 && (lddx % VW) == 0 && C / VW <= LY_THREADS) {
    const int ncv = C / VW, groups = LY_THREADS / ncv;
    const int cv = threadIdx.x % ncv, wl = threadIdx.x / ncv;
    if (wl >= groups) return;
    const int c = VW * cv;
    const float iw = 1.f / (float)W, ih = 1.f / (float)H;
    const long rows = (long)n_img * H;
    for (long nh = blockIdx.x; nh < rows; nh += gridDim.x) {
      const long n = nh / H;
      const int h = (int)(nh - n * H);
      const float* const ga = gp + (n * (H + W) + h) * C + c;
      const float* const gb = gp + (n * (H + W) + H) * C + c;
      f32x4 a[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) a[q] = ly_ldg4(ga + 4 * q) * iw;
      T* const drow = dx + nh * (long)W * lddx + c;
      for (int w = wl; w < W; w += groups) {
        f32x4 v[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) v[q] = a[q] + ly_ldg4(gb + (long)w * C + 4 * q) * ih;
        if (ACC) {
          f32x4 o[NQ];
          ly_rv_unpack(ly_ldrv<T>(drow + (long)w * lddx), o);
#pragma unroll
          for (int q = 0; q < NQ; ++q) v[q] += o[q];
        }
        *reinterpret_cast<RV*>(drow + (long)w * lddx) = ly_rv_pack(v, (RV*)nullptr);
      }
    }
    return;
  }
  const int nc4 = C >> 2;
  const long total = (long)n_img * H * W * nc4;
  const float iw = 1.f / (float)W, ih = 1.f / (float)H;
  for (long i = (long)blockIdx.x * LY_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * LY_THREADS) {
    const long pix = i / nc4;
    const int c = 4 * (int)(i - pix * nc4);
    const long row = pix / W;
    const int w = (int)(pix - row * W);
    const long n = row / H;
    const int h = (int)(row - n * H);
    const f32x4 a = ly_ldg4(gp + (n * (H + W) + h) * C + c), b = ly_ldg4(gp + (n * (H + W) + H + w) * C + c);
    f32x4 v = a * iw + b * ih;
    if (ACC) v += ly_ld4<T>(dx + pix * lddx + c);
    ly_st4<T>(dx + pix * lddx + c, v);
  }
}

extern "C" int ly_pool_hw_bwd(const float* gp, int n_img, int H, int W, int C, void* dx, int lddx, int accumulate, int dtype, void* stream) {
  LY_CHECK_DTYPE(dtype, "pool_hw_bwd");
  LY_CHECK(gp && dx && n_img > 0 && H > 0 && W > 0 && (C & 3) == 0 && (lddx & 3) == 0, "pool_hw_bwd: bad arguments");
  long nb = ly_ew_blocks((long)n_img * H * W * (C >> 2));
  if (nb > (long)n_img * H) nb = (long)n_img * H;              // (the row-walking form: a block per (image, row) pair at most)
  const dim3 grid((unsigned)nb);
  if (accumulate) {
    LY_WITH_T(dtype, hipLaunchKernelGGL((ly_pool_hw_bwd_kernel<T, true>), grid, dim3(LY_THREADS), 0, reinterpret_cast<hipStream_t>(stream), gp, n_img, H, W, C,
                                        reinterpret_cast<T*>(dx), lddx));
  } else {
    LY_WITH_T(dtype, hipLaunchKernelGGL((ly_pool_hw_bwd_kernel<T, false>), grid, dim3(LY_THREADS), 0, reinterpret_cast<hipStream_t>(stream), gp, n_img, H, W, C,
                                        reinterpret_cast<T*>(dx), lddx));
  }
  LY_LAUNCH_CHECK();
  return 0;
}

// -------------------------------------------------------------------------------------------------
// k x k / stride 1 / pad k//2 max-pool backward (SPPF, models/common.py:348-366): for every output position the
// argmax of its window (first maximum in row-major scan order, as ATen's max_pool2d) is recomputed from x and
// dy is added there: dx[argmax] += dy.  dx is accumulated into (float atomics), so chained pools can add into
// the gradient slots of the concat buffer in place.
// -------------------------------------------------------------------------------------------------
template <typename T, int K>
__global__ __launch_bounds__(LY_THREADS) void ly_maxpool_bwd_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ dy, int lddy,
                                                                    int n_img, int H, int W, int C, int k_rt, float* __restrict__ dx, int lddx) {
  // K > 0: window size known at compile time — the K*K loads of a position are all issued before the first compare, from clamped
  // coordinates (a `continue` around a load is a branch around a load: every later s_waitcnt turns conservative and the window is
  // walked one round trip at a time); out-of-range taps are masked in the compare.  K == 0: run-time window (any odd k).
  const int nc4 = C >> 2;
  const int k = K > 0 ? K : k_rt, r = k >> 1;
  const long total = (long)n_img * H * W * nc4;
  for (long i = (long)blockIdx.x * LY_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * LY_THREADS) {
    const long pix = i / nc4;
    const int c = 4 * (int)(i - pix * nc4);
    const long row = pix / W;
    const int w = (int)(pix - row * W);
    const long n = row / H;
    const int h = (int)(row - n * H);
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    long arg[4] = {-1, -1, -1, -1};
    if constexpr (K > 0) {
      typename LyT<T>::R4 raw[K * K];
#pragma unroll
      for (int dyy = 0; dyy < K; ++dyy)
#pragma unroll
        for (int dxx = 0; dxx < K; ++dxx) {
          int yy = h - r + dyy, xx = w - r + dxx;
          yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
          xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
          raw[dyy * K + dxx] = ly_ldr4<T>(x + ((n * H + yy) * W + xx) * ldx + c);
        }
#pragma unroll
      for (int dyy = 0; dyy < K; ++dyy)
#pragma unroll
        for (int dxx = 0; dxx < K; ++dxx) {
          const int yy = h - r + dyy, xx = w - r + dxx;
          const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
          const long q = (n * H + yy) * W + xx;
          const f32x4 v = ly_r4_f32(raw[dyy * K + dxx]);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (ok && (v[e] > best[e] || arg[e] < 0)) { best[e] = v[e]; arg[e] = q; }
        }
    } else {
      for (int yy = h - r; yy <= h + r; ++yy) {
        if (yy < 0 || yy >= H) continue;
        for (int xx = w - r; xx <= w + r; ++xx) {
          if (xx < 0 || xx >= W) continue;
          const long q = (n * H + yy) * W + xx;
          const f32x4 v = ly_ld4<T>(x + q * ldx + c);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (v[e] > best[e] || arg[e] < 0) { best[e] = v[e]; arg[e] = q; }
        }
      }
    }
    const f32x4 g = ly_ldg4(dy + pix * lddy + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicAdd(dx + arg[e] * lddx + c + e, g[e]);
  }
}

extern "C" int ly_maxpool_bwd(const void* x, int ldx, const float* dy, int lddy, int n_img, int H, int W, int C, int k, float* dx, int lddx,
                              int dtype, void* stream) {
  LY_CHECK_DTYPE(dtype, "maxpool_bwd");
  LY_CHECK(x && dy && dx && n_img > 0 && H > 0 && W > 0 && (k & 1) == 1, "maxpool_bwd: bad arguments");
  LY_CHECK((C & 3) == 0 && (ldx & 3) == 0 && (lddy & 3) == 0, "maxpool_bwd: C / ld must be multiples of 4");
  const dim3 grid((unsigned)ly_ew_blocks((long)n_img * H * W * (C >> 2)));
  if (k == 5) {                                            // SPPF(k=5): the one size LEAD-YOLO uses
    LY_WITH_T(dtype, hipLaunchKernelGGL((ly_maxpool_bwd_kernel<T, 5>), grid, dim3(LY_THREADS), 0, reinterpret_cast<hipStream_t>(stream),
                                        reinterpret_cast<const T*>(x), ldx, dy, lddy, n_img, H, W, C, k, dx, lddx));
  } else {
    LY_WITH_T(dtype, hipLaunchKernelGGL((ly_maxpool_bwd_kernel<T, 0>), grid, dim3(LY_THREADS), 0, reinterpret_cast<hipStream_t>(stream),
                                        reinterpret_cast<const T*>(x), ldx, dy, lddy, n_img, H, W, C, k, dx, lddx));
  }
  LY_LAUNCH_CHECK();
  return 0;
}


// -------------------------------------------------------------------------------------------------
// Per-channel vector work of BatchNorm, one launch each (replaces ~10 tiny elementwise launches):
//   ly_bn_finalize   striped (sum, sum^2) accumulators -> batch mean / invstd, y = x*scale + shift, running-stat update
//   ly_bn_bwd_coeffs striped (sum dv, sum dv*u)        -> dgamma, dbeta and the affine map du = alpha*dv + kappa + lambda*u
// Sums over stripes and the mean / variance arithmetic are done in double (E[x^2] - E[x]^2 cancels badly in fp32).
// -------------------------------------------------------------------------------------------------
// One thread per channel: its stripes are 2 x `stripes` independent loads (all in flight together, coalesced over the channels of a wave) folded
// in stripe order, and every per-channel parameter is requested before the fold — one memory round trip per launch.  (The first form — a
// 32-lane group per channel and two double shuffle trees, parameters loaded after them — was a chain of ~25 dependent LDS-pipe and memory
// latencies: 4.4 us per launch, 79 launches per step.)
#define LY_BNV_THREADS 64
template <typename TS>
__device__ __forceinline__ void ly_fold_stripes(const TS* __restrict__ p, const int stripes, const size_t stride, const int second, double& s1, double& s2) {
  s1 = 0.0;
  s2 = 0.0;
  int q = 0;
  if (stripes == LY_STATS_STRIPES) {
    // the usual case: all 2 x 32 loads of the channel in flight together, folded in stripe order (in batches of eight the fold was four
    // dependent memory round trips: the atomics' results come from the memory side)
    TS a[LY_STATS_STRIPES], b[LY_STATS_STRIPES];
#pragma unroll
    for (int k = 0; k < LY_STATS_STRIPES; ++k) {
      a[k] = p[(size_t)k * stride];
      b[k] = p[(size_t)k * stride + second];
    }
#pragma unroll
    for (int k = 0; k < LY_STATS_STRIPES; ++k) {
      s1 += (double)a[k];
      s2 += (double)b[k];
    }
    return;
  }
  for (; q + 8 <= stripes; q += 8) {
    TS a[8], b[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      a[k] = p[(size_t)(q + k) * stride];
      b[k] = p[(size_t)(q + k) * stride + second];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      s1 += (double)a[k];
      s2 += (double)b[k];
    }
  }
  for (; q < stripes; ++q) {
    s1 += (double)p[(size_t)q * stride];
    s2 += (double)p[(size_t)q * stride + second];
  }
}

template <typename TS>
__global__ __launch_bounds__(LY_BNV_THREADS) void ly_bn_finalize_kernel(const TS* __restrict__ stats, int stripes, int nch, int c_off, int N,
                                                                    double count, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                    const float* __restrict__ bias, float eps, float momentum, float* running_mean,
                                                                    float* running_var, long* nbt, float* __restrict__ scale, float* __restrict__ shift,
                                                                    float* __restrict__ mean, float* __restrict__ invstd) {
  const int c = blockIdx.x * LY_BNV_THREADS + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
  if (c >= N) return;
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f, bi = bias ? bias[c] : 0.f;
  const float rm = running_mean ? running_mean[c] : 0.f, rv = running_var ? running_var[c] : 0.f;
  double s1, s2;
  ly_fold_stripes(stats + c_off + c, stripes, (size_t)2 * nch, nch, s1, s2);
  const double m = s1 / count;
  double var = s2 / count - m * m;
  var = var > 0.0 ? var : 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = g * is;
  float sh = b - (float)m * sc;
  if (bias) sh += bi * sc;
  scale[c] = sc;
  shift[c] = sh;
  if (mean) mean[c] = (float)m;
  if (invstd) invstd[c] = is;
  if (running_mean) running_mean[c] = (1.f - momentum) * rm + momentum * (float)m;
  if (running_var) running_var[c] = (1.f - momentum) * rv + momentum * (float)(var * (count / (count > 1.0 ? count - 1.0 : 1.0)));
}

extern "C" int ly_bn_finalize(const void* stats, int stats_f64, int stripes, int nch, int c_off, int N, double count, const float* gamma, const float* beta,
                              const float* bias, float eps, float momentum, float* running_mean, float* running_var, long* nbt, float* scale,
                              float* shift, float* mean, float* invstd, void* stream) {
  LY_CHECK(stats && scale && shift && stripes > 0 && N > 0 && c_off >= 0 && c_off + N <= nch && count > 0, "bn_finalize: bad arguments");
  if (stats_f64)
    hipLaunchKernelGGL(ly_bn_finalize_kernel<double>, dim3((N + LY_BNV_THREADS - 1) / LY_BNV_THREADS), dim3(LY_BNV_THREADS), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const double*>(stats), stripes, nch, c_off, N, count, gamma, beta, bias, eps, momentum, running_mean, running_var, nbt,
                       scale, shift, mean, invstd);
  else
    hipLaunchKernelGGL(ly_bn_finalize_kernel<float>, dim3((N + LY_BNV_THREADS - 1) / LY_BNV_THREADS), dim3(LY_BNV_THREADS), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const float*>(stats), stripes, nch, c_off, N, count, gamma, beta, bias, eps, momentum, running_mean, running_var, nbt,
                       scale, shift, mean, invstd);
  LY_LAUNCH_CHECK();
  return 0;
}

// Two BatchNorms over one stacked output (ConvBnActPair: C3_CA's cv1 | cv2, channels [0, c_half) and [c_half, 2 c_half) of one
// statistics array) in ONE launch: every one of these coefficient kernels is a dependent ~4 us launch on the step's critical path.
struct LyBnSide {
  const float* gamma; const float* beta; float* running_mean; float* running_var; long* nbt; float eps, momentum;
};
template <typename TS>
__global__ __launch_bounds__(LY_BNV_THREADS) void ly_bn_finalize_pair_kernel(const TS* __restrict__ stats, int stripes, int c_half, double count, const LyBnSide u0,
                                                                         const LyBnSide u1, float* __restrict__ scale, float* __restrict__ shift,
                                                                         float* __restrict__ mean, float* __restrict__ invstd) {
  const int c = blockIdx.x * LY_BNV_THREADS + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x < 2) {
    long* nbt = threadIdx.x ? u1.nbt : u0.nbt;
    if (nbt) *nbt += 1;
  }
  const int nch = 2 * c_half;
  if (c >= nch) return;
  const bool hi = c >= c_half;
  const LyBnSide& U = hi ? u1 : u0;
  const int cl = hi ? c - c_half : c;
  const float g = U.gamma ? U.gamma[cl] : 1.f, b = U.beta ? U.beta[cl] : 0.f;
  const float rm = U.running_mean ? U.running_mean[cl] : 0.f, rv = U.running_var ? U.running_var[cl] : 0.f;
  double s1, s2;
  ly_fold_stripes(stats + c, stripes, (size_t)2 * nch, nch, s1, s2);
  const double m = s1 / count;
  double var = s2 / count - m * m;
  var = var > 0.0 ? var : 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)U.eps));
  const float sc = g * is;
  scale[c] = sc;
  shift[c] = b - (float)m * sc;
  mean[c] = (float)m;
  invstd[c] = is;
  if (U.running_mean) U.running_mean[cl] = (1.f - U.momentum) * rm + U.momentum * (float)m;
  if (U.running_var) U.running_var[cl] = (1.f - U.momentum) * rv + U.momentum * (float)(var * (count / (count > 1.0 ? count - 1.0 : 1.0)));
}

extern "C" int ly_bn_finalize_pair(const void* stats, int stats_f64, int stripes, int c_half, double count, const float* gamma0, const float* beta0, float eps0,
                                   float momentum0, float* running_mean0, float* running_var0, long* nbt0, const float* gamma1, const float* beta1,
                                   float eps1, float momentum1, float* running_mean1, float* running_var1, long* nbt1, float* scale, float* shift,
                                   float* mean, float* invstd, void* stream) {
  LY_CHECK(stats && scale && shift && mean && invstd && stripes > 0 && c_half > 0 && count > 0, "bn_finalize_pair: bad arguments");
  const LyBnSide u0 = {gamma0, beta0, running_mean0, running_var0, nbt0, eps0, momentum0}, u1 = {gamma1, beta1, running_mean1, running_var1, nbt1, eps1, momentum1};
  const dim3 grid((2 * c_half + LY_BNV_THREADS - 1) / LY_BNV_THREADS);
  if (stats_f64)
    hipLaunchKernelGGL(ly_bn_finalize_pair_kernel<double>, grid, dim3(LY_BNV_THREADS), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const double*>(stats),
                       stripes, c_half, count, u0, u1, scale, shift, mean, invstd);
  else
    hipLaunchKernelGGL(ly_bn_finalize_pair_kernel<float>, grid, dim3(LY_BNV_THREADS), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const float*>(stats),
                       stripes, c_half, count, u0, u1, scale, shift, mean, invstd);
  LY_LAUNCH_CHECK();
  return 0;
}

// the backward coefficients of the same two units: sums0 / sums1 are the two striped arrays of ly_bnact_bwd_reduce_pair ([stripes][2 c_half]
// each), a / mean / invstd and alpha / kappa / lambda the stacked [2 c_half] vectors, dgamma / dbeta per unit (ACCUMULATED, as below)
template <typename TS>
__global__ __launch_bounds__(LY_BNV_THREADS) void ly_bn_bwd_coeffs_pair_kernel(const TS* __restrict__ sums0, const TS* __restrict__ sums1, int stripes, int c_half,
                                                                           double count, const float* __restrict__ a, const float* __restrict__ mean,
                                                                           const float* __restrict__ invstd, float* __restrict__ dgamma0,
                                                                           float* __restrict__ dbeta0, float* __restrict__ dgamma1, float* __restrict__ dbeta1,
                                                                           float* __restrict__ alpha, float* __restrict__ kappa, float* __restrict__ lambda) {
  const int c = blockIdx.x * LY_BNV_THREADS + threadIdx.x;
  if (c >= 2 * c_half) return;
  const bool hi = c >= c_half;
  const int cl = hi ? c - c_half : c;
  float* const dgamma = hi ? dgamma1 : dgamma0;
  float* const dbeta = hi ? dbeta1 : dbeta0;
  const double mu = mean[c], is = invstd[c], av = a[c];
  const float dg0 = dgamma[cl], db0 = dbeta[cl];
  double s1, s2;
  ly_fold_stripes((hi ? sums1 : sums0) + cl, stripes, (size_t)2 * c_half, c_half, s1, s2);
  const double dg = (s2 - mu * s1) * is;
  dgamma[cl] = dg0 + (float)dg;
  dbeta[cl] = db0 + (float)s1;
  alpha[c] = (float)av;
  const double lam = -av * dg * is / count;
  lambda[c] = (float)lam;
  kappa[c] = (float)(-av * s1 / count - lam * mu);
}

extern "C" int ly_bn_bwd_coeffs_pair(const void* sums0, const void* sums1, int sums_f64, int stripes, int c_half, double count, const float* a, const float* mean,
                                     const float* invstd, float* dgamma0, float* dbeta0, float* dgamma1, float* dbeta1, float* alpha, float* kappa,
                                     float* lambda, void* stream) {
  LY_CHECK(sums0 && sums1 && a && mean && invstd && dgamma0 && dbeta0 && dgamma1 && dbeta1 && alpha && kappa && lambda && c_half > 0 && count > 0 && stripes > 0,
           "bn_bwd_coeffs_pair: bad arguments");
  const dim3 grid((2 * c_half + LY_BNV_THREADS - 1) / LY_BNV_THREADS);
  if (sums_f64)
    hipLaunchKernelGGL(ly_bn_bwd_coeffs_pair_kernel<double>, grid, dim3(LY_BNV_THREADS), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const double*>(sums0),
                       reinterpret_cast<const double*>(sums1), stripes, c_half, count, a, mean, invstd, dgamma0, dbeta0, dgamma1, dbeta1, alpha, kappa, lambda);
  else
    hipLaunchKernelGGL(ly_bn_bwd_coeffs_pair_kernel<float>, grid, dim3(LY_BNV_THREADS), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const float*>(sums0),
                       reinterpret_cast<const float*>(sums1), stripes, c_half, count, a, mean, invstd, dgamma0, dbeta0, dgamma1, dbeta1, alpha, kappa, lambda);
  LY_LAUNCH_CHECK();
  return 0;
}

template <typename TS>
__global__ __launch_bounds__(LY_BNV_THREADS) void ly_bn_bwd_coeffs_kernel(const TS* __restrict__ sums, int stripes, int N, double count,
                                                                      const float* __restrict__ a, const float* __restrict__ mean,
                                                                      const float* __restrict__ invstd, int train, float* __restrict__ dgamma,
                                                                      float* __restrict__ dbeta, float* __restrict__ alpha, float* __restrict__ kappa,
                                                                      float* __restrict__ lambda, const int tr_a, const int tr_b) {
  const int c = blockIdx.x * LY_BNV_THREADS + threadIdx.x;
  if (c >= N) return;
  // tr_a > 0: the sums are in [tr_a][tr_b] order (RFCBAMConv's generate BatchNorm: [tap][channel]) and dgamma / dbeta are the parameter's own
  // [tr_b][tr_a] storage — the transpose rides on the accumulation instead of two copies + two adds per layer
  const int cd = tr_a > 0 ? (c % tr_b) * tr_a + c / tr_b : c;
  const double mu = mean[c], is = invstd[c], av = a[c];
  const float dg0 = dgamma[cd], db0 = dbeta[cd];
  double s1, s2;
  ly_fold_stripes(sums + c, stripes, (size_t)2 * N, N, s1, s2);
  const double dg = (s2 - mu * s1) * is;
  dgamma[cd] = dg0 + (float)dg;    // ACCUMULATED: the targets may be the parameters' persistent .grad storage (zeroed by the optimiser step)
  dbeta[cd] = db0 + (float)s1;
  alpha[c] = (float)av;
  if (train) {
    const double lam = -av * dg * is / count;
    lambda[c] = (float)lam;
    kappa[c] = (float)(-av * s1 / count - lam * mu);
  } else {
    lambda[c] = 0.f;
    kappa[c] = 0.f;
  }
}

extern "C" int ly_bn_bwd_coeffs(const void* sums, int sums_f64, int stripes, int N, double count, const float* a, const float* mean, const float* invstd,
                                int train, float* dgamma, float* dbeta, float* alpha, float* kappa, float* lambda, int tr_a, int tr_b, void* stream) {
  LY_CHECK(sums && a && mean && invstd && dgamma && dbeta && alpha && kappa && lambda && N > 0 && count > 0, "bn_bwd_coeffs: bad arguments");
  LY_CHECK(tr_a == 0 || (tr_a > 0 && tr_b > 0 && tr_a * tr_b == N), "bn_bwd_coeffs: transposed targets need tr_a * tr_b == N");
  if (sums_f64)
    hipLaunchKernelGGL(ly_bn_bwd_coeffs_kernel<double>, dim3((N + LY_BNV_THREADS - 1) / LY_BNV_THREADS), dim3(LY_BNV_THREADS), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const double*>(sums), stripes, N, count, a, mean, invstd, train, dgamma, dbeta, alpha, kappa, lambda, tr_a, tr_b);
  else
    hipLaunchKernelGGL(ly_bn_bwd_coeffs_kernel<float>, dim3((N + LY_BNV_THREADS - 1) / LY_BNV_THREADS), dim3(LY_BNV_THREADS), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const float*>(sums), stripes, N, count, a, mean, invstd, train, dgamma, dbeta, alpha, kappa, lambda, tr_a, tr_b);
  LY_LAUNCH_CHECK();
  return 0;
}

// -------------------------------------------------------------------------------------------------
// bf16x3 fragment packing on the device (pack.frag_pack3 as ONE launch; weights change every optimiser step):
//   out[((t*S + s)*2 + plane)*64 + lane][j] = plane(W[16t + (lane&15)][32s + 16(j>>2) + 4(lane>>4) + (j&3)])
// W[r][k] is read as w[r*ld_r + k*ld_k], so a transposed view is packed without materialising it.
// -------------------------------------------------------------------------------------------------
template <int PL>
__global__ __launch_bounds__(LY_THREADS) void ly_frag_pack3_kernel(const float* __restrict__ w, int R, int K, long ld_r, long ld_k, int T, int S,
                                                                   uint4* __restrict__ out) {
  const long total = (long)T * S * 64;
  for (long i = (long)blockIdx.x * LY_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * LY_THREADS) {
    const int lane = (int)(i & 63);
    const long ts = i >> 6;
    const int s = (int)(ts % S), t = (int)(ts / S);
    const int row = 16 * t + (lane & 15), q = lane >> 4;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 32 * s + 16 * (j >> 2) + 4 * q + (j & 3);
      v[j] = (row < R && k < K) ? w[row * ld_r + k * ld_k] : 0.f;
    }
    bf16x8 hi, lo;
    ly_split8(v, hi, lo);
    out[(ts * PL) * 64 + lane] = __builtin_bit_cast(uint4, hi);
    if constexpr (PL == 2) out[(ts * PL + 1) * 64 + lane] = __builtin_bit_cast(uint4, lo);
  }
}

extern "C" int ly_frag_pack3(const float* w, int R, int K, long ld_r, long ld_k, int rows_to, int planes, void* out, void* stream) {
  LY_CHECK(w && out && R > 0 && K > 0 && (planes == 1 || planes == 2), "frag_pack3: bad arguments");
  const int rr = R > rows_to ? R : rows_to;
  const int T = (rr + 15) / 16, S = (K + 31) / 32;
  if (planes == 2)
    hipLaunchKernelGGL(ly_frag_pack3_kernel<2>, dim3((unsigned)ly_ew_blocks((long)T * S * 64)), dim3(LY_THREADS), 0, reinterpret_cast<hipStream_t>(stream), w,
                       R, K, ld_r, ld_k, T, S, reinterpret_cast<uint4*>(out));
  else
    hipLaunchKernelGGL(ly_frag_pack3_kernel<1>, dim3((unsigned)ly_ew_blocks((long)T * S * 64)), dim3(LY_THREADS), 0, reinterpret_cast<hipStream_t>(stream), w,
                       R, K, ld_r, ld_k, T, S, reinterpret_cast<uint4*>(out));
  LY_LAUNCH_CHECK();
  return 0;
}


// -------------------------------------------------------------------------------------------------
// Batched packing: every packed weight matrix the model needs (forward, transposed for dgrad, tap-flipped, ...) refreshed by ONE
// launch over a device-resident table of descriptors (113-144 ly_frag_pack3 launches per training step before).  A descriptor
// reads the fp32 parameter IN PLACE through an index map, so none of the permuted / padded / concatenated temporaries the
// per-matrix path built on the host exist any more:
//   packed row r = ra*nrb + rb, column k = (a*nb + b)*nc + c;   valid iff r < r_valid, b < vb, c < vc
//   element = src[ra*sra + rb*srb + a*sa + b*sb + c*sc]                (strides may be negative: tap flips)
// and writes row tiles [t0, t0 + T) of a [Ttot][S][planes][64][8] bf16 fragment image (zeros where invalid).
// -------------------------------------------------------------------------------------------------
template <int PL>
__device__ __forceinline__ void ly_pack_one(const LyPackDesc& d, long i) {
  const int lane = (int)(i & 63);
  const long ts = i >> 6;
  const int s = (int)(ts % d.S), t = (int)(ts / d.S);
  const int row = 16 * t + (lane & 15), q = lane >> 4;
  const bool rok = row < d.r_valid;
  const int ra = row / d.nrb, rb = row - ra * d.nrb;
  const long roff = (long)ra * d.sra + (long)rb * d.srb;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 32 * s + 16 * (j >> 2) + 4 * q + (j & 3);
    const int c = k % d.nc, ab = k / d.nc;
    const int b = ab % d.nb, a = ab / d.nb;
    const bool ok = rok && k < d.K && b < d.vb && c < d.vc;
    v[j] = ok ? d.src[roff + (long)a * d.sa + (long)b * d.sb + (long)c * d.sc] : 0.f;
  }
  bf16x8 hi, lo;
  ly_split8(v, hi, lo);
  uint4* out = reinterpret_cast<uint4*>(d.dst);
  const long fs = ((long)(d.t0 + t) * d.S + s) * PL;
  out[fs * 64 + lane] = __builtin_bit_cast(uint4, hi);
  if constexpr (PL == 2) out[(fs + 1) * 64 + lane] = __builtin_bit_cast(uint4, lo);
}

__global__ __launch_bounds__(LY_THREADS) void ly_pack_table_kernel(const LyPackDesc* __restrict__ tab, const int* __restrict__ blk_desc) {
  const LyPackDesc d = tab[blk_desc[blockIdx.x]];
  const long i = ((long)blockIdx.x - d.blk0) * LY_THREADS + threadIdx.x;
  if (i >= (long)d.T * d.S * 64) return;
  if (d.planes == 2) ly_pack_one<2>(d, i);
  else ly_pack_one<1>(d, i);
}

extern "C" int ly_pack_table(const LyPackDesc* table, const int* blk_desc, int n_blocks, void* stream) {
  LY_CHECK(table && blk_desc && n_blocks > 0, "pack_table: bad arguments");
  hipLaunchKernelGGL(ly_pack_table_kernel, dim3((unsigned)n_blocks), dim3(LY_THREADS), 0, reinterpret_cast<hipStream_t>(stream), table, blk_desc);
  LY_LAUNCH_CHECK();
  return 0;
}

// -------------------------------------------------------------------------------------------------
// MLPBlock backward, last step: dx = dy + g, except the first c4 channels (the partial 3x3 conv's) which take dy + t
// (t [rows, ldt]: the data gradient of the partial conv).  One pass instead of two adds and a strided copy.
// -------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(LY_THREADS) void ly_mlp_dx_kernel(const T* __restrict__ dy, const T* __restrict__ g, const T* __restrict__ t, int ldt,
                                                                long rows, int C, int c4, T* __restrict__ dx) {
  const int nq = C >> 2;
  const long total = rows * nq;
  for (long i = (long)blockIdx.x * LY_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * LY_THREADS) {
    const long r = i / nq;
    const int c = 4 * (int)(i - r * nq);
    const f32x4 a = ly_ld4<T>(dy + r * C + c);
    f32x4 b = ly_ld4<T>(g + r * C + c);
    if (c < c4) {                                           // (c4 need not be a multiple of 4: per-channel select in the boundary quad)
      const f32x4 tv = ly_ld4<T>(t + r * ldt + c);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c + e < c4) b[e] = tv[e];
    }
    ly_st4<T>(dx + r * C + c, a + b);
  }
}

extern "C" int ly_mlp_dx(const void* dy, const void* g, const void* t, int ldt, long rows, int C, int c4, void* dx, int dtype, void* stream) {
  LY_CHECK_DTYPE(dtype, "mlp_dx");
  LY_CHECK(dy && g && t && dx && rows > 0 && (C & 3) == 0 && (ldt & 3) == 0 && c4 > 0 && c4 <= C && ((c4 + 3) & ~3) <= ldt, "mlp_dx: bad arguments");
  LY_WITH_T(dtype, hipLaunchKernelGGL(ly_mlp_dx_kernel<T>, dim3((unsigned)ly_ew_blocks(rows * (C >> 2))), dim3(LY_THREADS), 0, reinterpret_cast<hipStream_t>(stream),
                                      reinterpret_cast<const T*>(dy), reinterpret_cast<const T*>(g), reinterpret_cast<const T*>(t), ldt, rows, C, c4,
                                      reinterpret_cast<T*>(dx)));
  LY_LAUNCH_CHECK();
  return 0;
}

// -------------------------------------------------------------------------------------------------
// SPPF backward without atomics (models/common.py:348-366, three chained k x k / s1 max-pools): the routing of every window — the tap
// index of its first maximum in row-major scan order, ATen's rule — is computed ONCE for all levels (ly_maxpool_arg over the first 3c
// channels of the [y | m(y) | m(m(y)) | m(m(m(y)))] buffer), then each level is a GATHER:
//   dtot[p] = d_own[p] + sum over the k*k windows q that contain p of [arg(q) == tap of p in q] * d_up[q]
// Deterministic, every output written once (the scatter version added with float atomics: 98 us per level).
// -------------------------------------------------------------------------------------------------
template <typename T, int K>
__global__ __launch_bounds__(LY_THREADS) void ly_maxpool_arg_kernel(const T* __restrict__ x, int ldx, int n_img, int H, int W, int C,
                                                                    unsigned char* __restrict__ arg, int lda) {
  constexpr int r = K >> 1;
  const int nc4 = C >> 2;
  const long total = (long)n_img * H * W * nc4;
  for (long i = (long)blockIdx.x * LY_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * LY_THREADS) {
    const long pix = i / nc4;
    const int c = 4 * (int)(i - pix * nc4);
    const long row = pix / W;
    const int w = (int)(pix - row * W);
    const long n = row / H;
    const int h = (int)(row - n * H);
    typename LyT<T>::R4 raw[K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) {
      int yy = h - r + t / K, xx = w - r + t % K;
      yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
      xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
      raw[t] = ly_ldr4<T>(x + ((n * H + yy) * W + xx) * ldx + c);
    }
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int a[4] = {-1, -1, -1, -1};
#pragma unroll
    for (int t = 0; t < K * K; ++t) {
      const int yy = h - r + t / K, xx = w - r + t % K;
      const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
      const f32x4 v = ly_r4_f32(raw[t]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (ok && (v[e] > best[e] || a[e] < 0)) { best[e] = v[e]; a[e] = t; }
    }
    *reinterpret_cast<unsigned*>(arg + pix * lda + c) = (unsigned)a[0] | ((unsigned)a[1] << 8) | ((unsigned)a[2] << 16) | ((unsigned)a[3] << 24);
  }
}

template <typename TD, typename TO, int K>
__global__ __launch_bounds__(LY_THREADS) void ly_maxpool_gather_kernel(const unsigned char* __restrict__ arg, int lda, const float* __restrict__ d_up, int ldu,
                                                                       const TD* __restrict__ d_own, int ldd, int n_img, int H, int W, int C,
                                                                       TO* __restrict__ out, int ldo) {
  constexpr int r = K >> 1;
  const int nc4 = C >> 2;
  const long total = (long)n_img * H * W * nc4;
  for (long i = (long)blockIdx.x * LY_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * LY_THREADS) {
    const long pix = i / nc4;
    const int c = 4 * (int)(i - pix * nc4);
    const long row = pix / W;
    const int w = (int)(pix - row * W);
    const long n = row / H;
    const int h = (int)(row - n * H);
    unsigned av[K * K];
    f32x4 dv[K * K];
    // window q = (h - (ty - r), w - (tx - r)) sees this pixel as its tap t = ty*K + tx; all loads first, from clamped coordinates
#pragma unroll
    for (int t = 0; t < K * K; ++t) {
      int qy = h - (t / K - r), qx = w - (t % K - r);
      qy = qy < 0 ? 0 : (qy >= H ? H - 1 : qy);
      qx = qx < 0 ? 0 : (qx >= W ? W - 1 : qx);
      const long q = (n * H + qy) * W + qx;
      av[t] = *reinterpret_cast<const unsigned*>(arg + q * lda + c);
      dv[t] = ly_ldg4(d_up + q * ldu + c);
    }
    f32x4 s = ly_ld4<TD>(d_own + pix * ldd + c);
#pragma unroll
    for (int t = 0; t < K * K; ++t) {
      const int qy = h - (t / K - r), qx = w - (t % K - r);
      const bool ok = qy >= 0 && qy < H && qx >= 0 && qx < W;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (ok && ((av[t] >> (8 * e)) & 255u) == (unsigned)t) s[e] += dv[t][e];
    }
    ly_st4<TO>(out + pix * ldo + c, s);
  }
}

// SPPF backward in ONE launch for small maps (models/common.py:348-366 at 20 x 20 / 40 x 40): a block owns one image x 8 channels and keeps the
// whole map in LDS — per level j = 2, 1, 0 it stages y_j, computes the window argmax of every pixel (the rule of ly_maxpool_arg), and gathers
// t_j = d_j + sum over the windows routed to the pixel of t_{j+1} (the order of ly_maxpool_gather: bit-identical sums); t_3 = d_3.  The three
// per-level launches read every routing byte and gradient 25 times through L2 (180 us for a 4 MB map at bs=64); here they come from LDS.
#define LY_SPPF_CG 8
template <typename T, int K>
__global__ __launch_bounds__(LY_THREADS) void ly_sppf_bwd_kernel(const T* __restrict__ buf, int ldb, const T* __restrict__ d, int ldd, int H, int W, int c,
                                                                 T* __restrict__ out, int ldo) {
  constexpr int r = K >> 1, CG = LY_SPPF_CG, NQ = CG / 4;
  extern __shared__ f32x4 ly_sppf_smem[];
  const int HW = H * W;
  f32x4* const GA = ly_sppf_smem;                                  // [HW][NQ] running gradient t_{j+1}
  f32x4* const GB = GA + HW * NQ;                                  // [HW][NQ] t_j being built
  f32x4* const Y = GB + HW * NQ;                                   // [HW][NQ] y_j as fp32 (exact for bf16 / fp32 inputs)
  unsigned* const A = reinterpret_cast<unsigned*>(Y + HW * NQ);    // [HW][NQ] four routing bytes
  const int groups = c / CG;
  const int bid = ly_xcd_remap((int)blockIdx.x, (int)gridDim.x);            // the channel groups of an image read the same cache lines: one XCD's L2
  const long n = bid / groups;
  const int c0 = (int)(bid - n * groups) * CG;
  const int tid = threadIdx.x;
  const T* const bn = buf + n * HW * (long)ldb;
  const T* const dn = d + n * HW * (long)ldd;
  for (int i = tid; i < HW * NQ; i += LY_THREADS) {
    const int pix = i / NQ, q = i - pix * NQ;
    GA[i] = ly_ld4<T>(dn + (long)pix * ldd + 3 * c + c0 + 4 * q);
  }
  f32x4* gin = GA;
  f32x4* gout = GB;
  for (int j = 2; j >= 0; --j) {
    // y_j and d_j of the level are requested together (d_j waits in gout, which the gather below overwrites item by item): one memory
    // round trip per level instead of two
    for (int i = tid; i < HW * NQ; i += LY_THREADS) {
      const int pix = i / NQ, q = i - pix * NQ;
      const f32x4 yv = ly_ld4<T>(bn + (long)pix * ldb + j * c + c0 + 4 * q);
      const f32x4 dv = ly_ld4<T>(dn + (long)pix * ldd + j * c + c0 + 4 * q);
      Y[i] = yv;
      gout[i] = dv;
    }
    __syncthreads();
    for (int i = tid; i < HW * NQ; i += LY_THREADS) {
      const int pix = i / NQ, q = i - pix * NQ;
      const int h = pix / W, w = pix - h * W;
      f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      int a[4] = {-1, -1, -1, -1};
#pragma unroll
      for (int t = 0; t < K * K; ++t) {
        const int yy = h - r + t / K, xx = w - r + t % K;
        const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
        const f32x4 v = Y[(ok ? yy * W + xx : pix) * NQ + q];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (ok && (v[e] > best[e] || a[e] < 0)) { best[e] = v[e]; a[e] = t; }
      }
      A[i] = (unsigned)a[0] | ((unsigned)a[1] << 8) | ((unsigned)a[2] << 16) | ((unsigned)a[3] << 24);
    }
    __syncthreads();
    for (int i = tid; i < HW * NQ; i += LY_THREADS) {
      const int pix = i / NQ, q = i - pix * NQ;
      const int h = pix / W, w = pix - h * W;
      f32x4 sacc = gout[i];
#pragma unroll
      for (int t = 0; t < K * K; ++t) {
        // window qp = (h - (ty - r), w - (tx - r)) sees this pixel as its tap t
        const int qy = h - (t / K - r), qx = w - (t % K - r);
        const bool ok = qy >= 0 && qy < H && qx >= 0 && qx < W;
        const int qp = ok ? qy * W + qx : pix;
        const unsigned av = A[qp * NQ + q];
        const f32x4 dv = gin[qp * NQ + q];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (ok && ((av >> (8 * e)) & 255u) == (unsigned)t) sacc[e] += dv[e];
      }
      if (j == 0) ly_st4<T>(out + (n * HW + pix) * (long)ldo + c0 + 4 * q, sacc);
      else gout[i] = sacc;
    }
    __syncthreads();
    f32x4* const tmp = gin; gin = gout; gout = tmp;
  }
}

// returns 1 (nothing launched) when the map does not fit the fused kernel: the caller then uses ly_maxpool_arg / ly_maxpool_gather
extern "C" int ly_sppf_bwd(const void* buf, int ldb, const void* d, int ldd, int n_img, int H, int W, int c, int k, void* out, int ldo, int dtype, void* stream) {
  LY_CHECK_DTYPE(dtype, "sppf_bwd");
  LY_CHECK(buf && d && out && n_img > 0 && H > 0 && W > 0 && c > 0, "sppf_bwd: bad arguments");
  const size_t lds = (size_t)H * W * (LY_SPPF_CG / 4) * (3 * sizeof(f32x4) + sizeof(unsigned));
  if (k != 5 || c % LY_SPPF_CG || (ldb & 3) || (ldd & 3) || (ldo & 3) || lds > 150 * 1024 || (long)n_img * (c / LY_SPPF_CG) > 2000000000L) return 1;
  const dim3 grid((unsigned)(n_img * (c / LY_SPPF_CG)));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  LY_WITH_T(dtype, {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ly_sppf_bwd_kernel<T, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((ly_sppf_bwd_kernel<T, 5>), grid, dim3(LY_THREADS), lds, st, reinterpret_cast<const T*>(buf), ldb, reinterpret_cast<const T*>(d), ldd, H, W, c,
                       reinterpret_cast<T*>(out), ldo);
  });
  LY_LAUNCH_CHECK();
  return 0;
}

extern "C" int ly_maxpool_arg(const void* x, int ldx, int n_img, int H, int W, int C, int k, unsigned char* arg, int lda, int dtype, void* stream) {
  LY_CHECK_DTYPE(dtype, "maxpool_arg");
  LY_CHECK(x && arg && n_img > 0 && H > 0 && W > 0 && k == 5, "maxpool_arg: bad arguments (built for k = 5, SPPF)");
  LY_CHECK((C & 3) == 0 && (ldx & 3) == 0 && (lda & 3) == 0, "maxpool_arg: C / ld must be multiples of 4");
  LY_WITH_T(dtype, hipLaunchKernelGGL((ly_maxpool_arg_kernel<T, 5>), dim3((unsigned)ly_ew_blocks((long)n_img * H * W * (C >> 2))), dim3(LY_THREADS), 0,
                                      reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const T*>(x), ldx, n_img, H, W, C, arg, lda));
  LY_LAUNCH_CHECK();
  return 0;
}

extern "C" int ly_maxpool_gather(const unsigned char* arg, int lda, const float* d_up, int ldu, const void* d_own, int ldd, int own_dtype, int n_img, int H,
                                 int W, int C, int k, void* out, int ldo, int out_dtype, void* stream) {
  LY_CHECK_DTYPE(own_dtype, "maxpool_gather");
  LY_CHECK_DTYPE(out_dtype, "maxpool_gather");
  LY_CHECK(arg && d_up && d_own && out && n_img > 0 && H > 0 && W > 0 && k == 5, "maxpool_gather: bad arguments (built for k = 5, SPPF)");
  LY_CHECK((C & 3) == 0 && (lda & 3) == 0 && (ldu & 3) == 0 && (ldd & 3) == 0 && (ldo & 3) == 0, "maxpool_gather: C / ld must be multiples of 4");
  const dim3 grid((unsigned)ly_ew_blocks((long)n_img * H * W * (C >> 2)));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define LY_MPG(TD, TO) hipLaunchKernelGGL((ly_maxpool_gather_kernel<TD, TO, 5>), grid, dim3(LY_THREADS), 0, st, arg, lda, d_up, ldu, reinterpret_cast<const TD*>(d_own), ldd, \
                                          n_img, H, W, C, reinterpret_cast<TO*>(out), ldo)
  if (own_dtype == LY_BF16 && out_dtype == LY_BF16) LY_MPG(__bf16, __bf16);
  else if (own_dtype == LY_BF16) LY_MPG(__bf16, float);
  else if (out_dtype == LY_BF16) LY_MPG(float, __bf16);
  else LY_MPG(float, float);
#undef LY_MPG
  LY_LAUNCH_CHECK();
  return 0;
}
