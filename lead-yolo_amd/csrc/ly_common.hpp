// Common device helpers for the LEAD-YOLO gfx950 kernels.
//
// Tile convention used by every contraction in this library (fp32 path):
//   v_mfma_f32_16x16x4_f32,  D[row][col] += sum_k A[row][k] * B[k][col]
//   A operand  = WEIGHTS      (row  = output channel),  lane l supplies A[row = l&15][k = l>>4]
//   B operand  = ACTIVATIONS  (col  = pixel),           lane l supplies B[k = l>>4][col = l&15]
//   D          : lane l holds D[row = 4*(l>>4) + r][col = l&15], r = 0..3
//   => each lane ends up with 4 CONSECUTIVE output channels of ONE pixel: NHWC float4 stores.
//
// One "k-step" covers 16 values of K with four MFMAs; lane (i = l&15, q = l>>4) feeds
//   k = 16*s + 4*q + j   in MFMA j (j = 0..3)
// so both operands are fetched as ONE 16-byte vector per lane per step (K-contiguous layouts:
// NHWC activations, [cout][k] weights).  The summation order inside a step is permuted relative to
// natural k order, which is irrelevant to the result's definition (a sum) and is applied to both
// operands identically.
//
// Weights are pre-packed ("frag-packed") so that one wave-wide fragment is one contiguous 1 KiB
// read:  Wp[(t*S + s)*64 + lane] (float4) = W[16t + (lane&15)][16s + 4(lane>>4) + 0..3], zero padded.
//
// A D tile of one contraction is directly the B operand of the next contraction over those channels
// (register j of lane (i,q) is channel 16t + 4q + j of pixel i): no LDS round trip between the two
// 1x1 convolutions of an MLP block.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LY_WAVE 64
#define LY_THREADS 256

__device__ __forceinline__ f32x4 ly_mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// acc += Wfrag(4 k-values) x Xfrag(4 k-values)
__device__ __forceinline__ f32x4 ly_mfma4(const f32x4 w, const f32x4 x, f32x4 acc) {
  acc = ly_mfma(w[0], x[0], acc);
  acc = ly_mfma(w[1], x[1], acc);
  acc = ly_mfma(w[2], x[2], acc);
  acc = ly_mfma(w[3], x[3], acc);
  return acc;
}

// Read-only, wave-uniform data (weights indexed by a wave-uniform expression): a constant-address-space
// view makes the compiler use the scalar cache (s_load_dwordxN) instead of per-lane vector loads.
typedef const float __attribute__((address_space(4))) ly_cfloat;
__device__ __forceinline__ const ly_cfloat* ly_const(const float* p) { return (const ly_cfloat*)p; }

__device__ __forceinline__ f32x4 ly_zero4() { return (f32x4){0.f, 0.f, 0.f, 0.f}; }

__device__ __forceinline__ f32x4 ly_ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void ly_stg4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// v_exp_f32 + v_rcp_f32 (each <= 1 ulp): ~3 instructions instead of the ~15 of an IEEE division
__device__ __forceinline__ float ly_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float ly_silu(float x) { return x * ly_sigmoid(x); }

// activation applied to a whole float4 with ONE uniform switch (no per-element branching)
__device__ __forceinline__ f32x4 ly_act4(f32x4 u, int act) {
  f32x4 v;
  if (act == 2) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = ly_silu(u[r]);
  } else if (act == 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = fmaxf(u[r], 0.f);
  } else {
    v = u;
  }
  return v;
}
__device__ __forceinline__ float ly_relu(float x) { return fmaxf(x, 0.f); }
__device__ __forceinline__ float ly_hswish(float x) { return x * fminf(fmaxf(x + 3.f, 0.f), 6.f) * (1.f / 6.f); }

enum { LY_ACT_NONE = 0, LY_ACT_RELU = 1, LY_ACT_SILU = 2 };  // == LY_ACT_*_ of the public header

template <int ACT>
__device__ __forceinline__ float ly_act(float x) {
  if (ACT == LY_ACT_RELU) return ly_relu(x);
  if (ACT == LY_ACT_SILU) return ly_silu(x);
  return x;
}

// exact floor(x / d) for 0 <= x < 2^24 via a float reciprocal and one fix-up step
__device__ __forceinline__ int ly_fdiv(int x, int d, float inv) {
  int q = (int)((float)x * inv);
  int r = x - q * d;
  if (r < 0) { --q; r += d; }
  if (r >= d) ++q;
  return q;
}

// (h, w) of flattened pixel gp (< 2^24) in an image of H x W, without 64-bit division
__device__ __forceinline__ void ly_pix_hw(long gp, int H, int W, int& h, int& w) {
  const int g = (int)gp;
  const int row = ly_fdiv(g, W, 1.f / (float)W);     // n*H + h
  w = g - row * W;
  h = row - ly_fdiv(row, H, 1.f / (float)H) * H;
}

// 9-bit validity mask of the 3x3 neighbourhood (pad 1) of pixel (h, w) in an H x W image.
// bit (ty*3 + tx) set  <=>  (h + ty - 1, w + tx - 1) is inside the image.
__device__ __forceinline__ uint32_t ly_tapmask(int h, int w, int H, int W, bool pixel_valid) {
  if (!pixel_valid) return 0u;
  uint32_t rows = (h > 0 ? 1u : 0u) | 2u | (h < H - 1 ? 4u : 0u);
  uint32_t cols = (w > 0 ? 1u : 0u) | 2u | (w < W - 1 ? 4u : 0u);
  uint32_t m = 0;
  if (rows & 1u) m |= cols;
  if (rows & 2u) m |= cols << 3;
  if (rows & 4u) m |= cols << 6;
  return m;
}

// XCD-aware block remap (8 XCDs, round-robin dispatch): gives each XCD a contiguous chunk of the
// logical grid so neighbouring tiles share that XCD's L2.  Bijective for any n.
__device__ __forceinline__ int ly_xcd_remap(int bid, int n) {
  const int nx = 8;
  int q = n / nx, r = n % nx;
  int xcd = bid % nx, slot = bid / nx;
  // XCD x owns q (+1 if x < r) logical blocks, laid out consecutively
  int base = xcd * q + (xcd < r ? xcd : r);
  return base + slot;
}

// L2 warm-up: the whole grid touches every 128-byte line of a read-only parameter block once, right at
// kernel start and in parallel, so that the dependent weight-fragment fetches of the contraction loops
// hit L2 instead of chasing HBM misses one after another (weights are evicted between layers by the
// activation stream).  The loaded values feed a never-true store so the loads cannot be elided.
__device__ __forceinline__ void ly_l2_warm(const void* base, long bytes, float* sink) {
  const long lines = bytes >> 7;
  const float* p = reinterpret_cast<const float*>(base);
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < lines; i += (long)gridDim.x * blockDim.x) acc += p[i * 32];
  if (acc == 1.2345678e-30f) *sink = acc;
}

// All-reduce (sum / max) over aligned groups of G lanes (G a power of two, 2 .. 64): every lane ends with its group's result.  The four
// steps inside a 16-lane row are DPP moves on the vector ALU (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: after the first two a quad
// is uniform, so mirroring — which pairs lanes of different quads / different 8-lane halves — reduces like xor 4 / xor 8 would); only rows
// meet through the LDS pipe (ds_bpermute: ~100 cycles of latency per step where a DPP add is one ALU instruction).
template <int CTRL> __device__ __forceinline__ float ly_dpp_mov(const float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// the value of lane l ^ M: inside a 16-lane row (M = 1, 2, 4, 8) as DPP moves — quad_perm for 1 and 2, a row rotation by 8 for 8, and for 4 a
// shift left by 4 into the lanes with bit 2 clear plus a shift right by 4 into the others (bank masks) — across rows through the LDS pipe
template <int M> __device__ __forceinline__ float ly_lane_xor(const float v) {
  if constexpr (M == 1) return ly_dpp_mov<0xB1>(v);
  else if constexpr (M == 2) return ly_dpp_mov<0x4E>(v);
  else if constexpr (M == 8) return ly_dpp_mov<0x128>(v);                // row_ror:8
  else if constexpr (M == 4) {
    const int b = __builtin_bit_cast(int, v);
    int r = __builtin_amdgcn_update_dpp(0, b, 0x104, 0xf, 0x5, false);       // row_shl:4 -> lanes 0-3, 8-11 of a row take lane + 4
    r = __builtin_amdgcn_update_dpp(r, b, 0x114, 0xf, 0xA, false);           // row_shr:4 -> lanes 4-7, 12-15 take lane - 4
    return __builtin_bit_cast(float, r);
  } else return __shfl_xor(v, M);
}
__device__ __forceinline__ float ly_group_sum(float v, const int G) {
  if (G >= 2) v += ly_dpp_mov<0xB1>(v);
  if (G >= 4) v += ly_dpp_mov<0x4E>(v);
  if (G >= 8) v += ly_dpp_mov<0x141>(v);
  if (G >= 16) v += ly_dpp_mov<0x140>(v);
  if (G >= 32) v += __shfl_xor(v, 16);
  if (G >= 64) v += __shfl_xor(v, 32);
  return v;
}
__device__ __forceinline__ float ly_group_max(float v, const int G) {
  if (G >= 2) v = fmaxf(v, ly_dpp_mov<0xB1>(v));
  if (G >= 4) v = fmaxf(v, ly_dpp_mov<0x4E>(v));
  if (G >= 8) v = fmaxf(v, ly_dpp_mov<0x141>(v));
  if (G >= 16) v = fmaxf(v, ly_dpp_mov<0x140>(v));
  if (G >= 32) v = fmaxf(v, __shfl_xor(v, 16));
  if (G >= 64) v = fmaxf(v, __shfl_xor(v, 32));
  return v;
}

// Small parameter-gradient reductions (a handful of values that many blocks add to: Detect's bias, get_weight's 18 taps, the k = 1 generate
// weights, CoordAtt's tiny MLP).  f64 != 0: the target is a zeroed DOUBLE scratch of the same shape — the arrival order of the atomics then only
// moves the 53rd bit, and ly_f64_add rounds the sums into the parameters' fp32 gradient storage after the backward pass: the same bits in
// every run (as the BatchNorm statistics, ly_stats_flush below).  f64 == 0: float atomics straight into the target.
__device__ __forceinline__ void ly_gacc(float* __restrict__ p, const long i, const float v, const int f64) {
  if (f64) atomicAdd(reinterpret_cast<double*>(p) + i, (double)v);
  else atomicAdd(p + i, v);
}

// Batch-statistics pass support: a lane holds 4 consecutive channels (c .. c+3) of some pixels; sum the
// two 4-vectors over the 16 lanes that share lq (lanes differing in l&15) and let lane l&15 == 0 add them
// to stats[c + r] (sum) and stats[nch + c + r] (sum of squares).
// The accumulator is STRIPED: LY_STATS_STRIPES copies of the [2*nch] array, block b adds into copy
// b % LY_STATS_STRIPES (thousands of blocks adding to the same few addresses serialise in L2 otherwise);
// ly_bn_finalize sums the copies in double precision.
// The accumulators are DOUBLES (round 4): the wave's partial sums are computed in a fixed order, and what the arrival order of the
// atomics can still change is the 53rd bit of a stripe — after ly_bn_finalize rounds mean / variance to fp32 the batch statistics of two
// runs are the same bits (a summation-order effect survives the rounding with probability ~2^-29 per value).  With float accumulators
// the 1e-7 noise of the statistics flipped ReLU / arg-max decisions downstream and two training runs differed by 1e-3 .. 5e-2 in whole
// gradients; forward and routing decisions of a training step are now reproducible.
#define LY_STATS_STRIPES 32
__device__ __forceinline__ void ly_stats_flush(double* __restrict__ stats, int nch, int c, f32x4 s1, f32x4 s2) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    s1[r] = ly_group_sum(s1[r], 16);
    s2[r] = ly_group_sum(s2[r], 16);
  }
  if ((threadIdx.x & 15) == 0) {
    double* st = stats + (size_t)(blockIdx.x & (LY_STATS_STRIPES - 1)) * 2 * nch;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (c + r < nch) {
        atomicAdd(st + c + r, (double)s1[r]);
        atomicAdd(st + nch + c + r, (double)s2[r]);
      }
  }
}

// Sums of N values (N = 4, 8, 16) over the 64 lanes of a wave with N - 1 + log2(64 / N) cross-lane moves instead of 6 N: at each of the first
// log2 N butterfly steps a lane keeps one half of its values and hands the other half to its partner (reduce-scatter), the remaining steps
// are a plain tree on one value.  A cross-lane move is an LDS-pipe instruction (ds_bpermute / ds_swizzle): in the small vector kernels trees
// of them, not the arithmetic, set the time (ly_rf3m_stats: 40 -> 9 moves per chunk = 74 -> 65 us).  Afterwards v[0] of lane l holds the
// full sum of value l >> (6 - log2 N); the lanes that share those top bits hold copies.
template <int CNT, int MASK, int N>
__device__ __forceinline__ void ly_rs_step(float (&v)[N], const int lane) {
  if constexpr (CNT > 1) {
    constexpr int HALF = CNT / 2;
    const bool up = (lane & MASK) != 0;
#pragma unroll
    for (int k = 0; k < HALF; ++k) {
      const float keep = up ? v[k + HALF] : v[k], send = up ? v[k] : v[k + HALF];
      v[k] = keep + ly_lane_xor<MASK>(send);
    }
    ly_rs_step<HALF, MASK / 2, N>(v, lane);
  } else if constexpr (MASK > 0) {
    v[0] += ly_lane_xor<MASK>(v[0]);
    ly_rs_step<1, MASK / 2, N>(v, lane);
  }
}
template <int N>
__device__ __forceinline__ void ly_reduce_scatter(float (&v)[N], const int lane) {
  static_assert(N == 4 || N == 8 || N == 16, "ly_reduce_scatter: N = 4, 8 or 16");
  ly_rs_step<N, 32, N>(v, lane);
}

extern "C" void ly_set_error(const char* fmt, ...);

#define LY_CHECK(cond, ...)                \
  do {                                     \
    if (!(cond)) {                         \
      ly_set_error(__VA_ARGS__);           \
      return -1;                           \
    }                                      \
  } while (0)

// hipFuncSetAttribute is per DEVICE: a launcher's "already configured" memo is a bitmask over device ordinals (one process may drive
// several GPUs: tests, nn.DataParallel-style use), not a process-wide flag
struct LyDevOnce {
  unsigned long long mask = 0ull;
  bool need() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d > 63) return true;
    if (mask >> d & 1ull) return false;
    mask |= 1ull << d;
    return true;
  }
};

#define LY_LAUNCH_CHECK()                                         \
  do {                                                            \
    hipError_t e_ = hipGetLastError();                            \
    if (e_ != hipSuccess) {                                       \
      ly_set_error("HIP launch failed: %s", hipGetErrorString(e_)); \
      return -2;                                                  \
    }                                                             \
  } while (0)
