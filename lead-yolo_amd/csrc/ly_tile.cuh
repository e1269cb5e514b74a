// bf16x3 contraction core (gfx950): fp32-grade products on the bf16 matrix pipe.
//
// Every fp32 operand x is split as x = hi + lo (+ O(2^-16 |x|)), hi = bf16(x), lo = bf16(x - hi), and
//      w * x  ~=  w_hi*x_hi + w_hi*x_lo + w_lo*x_hi          (w_lo*x_lo ~ 2^-16 relative is dropped)
// is accumulated in fp32 by three v_mfma_f32_16x16x32_bf16 per 32 values of K.  Relative error per
// product ~2^-16 (vs 2^-8 for plain bf16), i.e. far inside the 1e-3 parity budget, at 16/3 = 5.3x the
// rate of the exact v_mfma_f32_16x16x4_f32 pipe (MI355X_MICROARCH.md: f32 MFMA = 1/16 of bf16).
//
// Tile convention (both operands K-contiguous, same as the fp32 core in ly_common.cuh):
//   A operand = WEIGHTS (row = output channel l&15), B operand = ACTIVATIONS (col = pixel l&15),
//   D: lane l holds rows 4*(l>>4) + r of column l&15  -> 4 consecutive output channels of one pixel.
//   k-set of lane (i, q = l>>4) in k-step s (32 values):  k = 32s + 16*(j>>2) + 4q + (j&3), j = 0..7
//   i.e. two groups of 4 consecutive k.  With this permutation
//     * an LDS-resident activation row is read as two 8-byte pieces (ds_read_b64, conflict-free with
//       the row stride below), and
//     * two fp32 D tiles (hidden channels 32u .. 32u+31) ARE the B operand of k-step u of the next
//       contraction after an in-register split: lane (i,q) register r of tile 2u+h is k = 32u+16h+4q+r.
//   Weights are frag-packed on the host (pack.frag_pack3): uint4 wpk[((t*S + s)*2 + plane)*64 + lane].
//
// LDS activation image: two planes (hi, lo) of [rows][KP] bf16, KP = ceil32(K), row stride
// RS = 2*KP + 16 bytes (RS/16 odd => the 2 x b64 fragment reads of a wave hit 32 distinct 8-byte slots).
#pragma once
#include "ly_common.cuh"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void ly_split4(const f32x4 v, bf16x4& hi, bf16x4& lo) {
  hi = __builtin_convertvector(v, bf16x4);
  const f32x4 back = __builtin_convertvector(hi, f32x4);
  lo = __builtin_convertvector(v - back, bf16x4);
}

__device__ __forceinline__ bf16x8 ly_cat8(const bf16x4 a, const bf16x4 b) {
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ f32x4 ly_mfma_bf16(const bf16x8 a, const bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// acc += W x X over 32 k-values, 3-term split (small terms first)
__device__ __forceinline__ f32x4 ly_mfma3(const bf16x8 whi, const bf16x8 wlo, const bf16x8 xhi, const bf16x8 xlo, f32x4 acc) {
  acc = ly_mfma_bf16(whi, xlo, acc);
  acc = ly_mfma_bf16(wlo, xhi, acc);
  acc = ly_mfma_bf16(whi, xhi, acc);
  return acc;
}

__host__ __device__ constexpr int ly_kp(int k) { return (k + 31) / 32 * 32; }
__host__ __device__ constexpr int ly_rs(int kp) { return 2 * kp + 16; }      // bytes

// store 4 consecutive channels (c % 4 == 0) of one row into both planes
__device__ __forceinline__ void ly_lds_put4(char* hi_plane, char* lo_plane, int row_byte, int c, const f32x4 v) {
  bf16x4 h, l;
  ly_split4(v, h, l);
  *reinterpret_cast<bf16x4*>(hi_plane + row_byte + 2 * c) = h;
  *reinterpret_cast<bf16x4*>(lo_plane + row_byte + 2 * c) = l;
}

// B-operand fragment of k-step s for the row at `row_byte`
__device__ __forceinline__ bf16x8 ly_lds_frag(const char* plane, int row_byte, int s, int lq) {
  const char* p = plane + row_byte + 2 * (32 * s + 4 * lq);
  const bf16x4 a = *reinterpret_cast<const bf16x4*>(p);
  const bf16x4 b = *reinterpret_cast<const bf16x4*>(p + 32);
  return ly_cat8(a, b);
}

struct LyWFrag {
  bf16x8 hi, lo;
};

__device__ __forceinline__ LyWFrag ly_wfrag(const uint4* __restrict__ wpk, long tile_step, int lane) {
  LyWFrag f;
  const uint4 a = wpk[(tile_step * 2) * 64 + lane];
  const uint4 b = wpk[(tile_step * 2 + 1) * 64 + lane];
  f.hi = __builtin_bit_cast(bf16x8, a);
  f.lo = __builtin_bit_cast(bf16x8, b);
  return f;
}


// Cooperative staging of `total` float4 items by the whole block, U loads in flight per thread.
// src(idx) returns the global address of item idx or nullptr (-> zeros); dst(idx, v) consumes it.
// All U loads of a batch are issued back to back from clamped addresses (no branch around a load),
// so a thread pays one memory latency per batch instead of one per item.
template <int U, class SrcFn, class DstFn>
__device__ __forceinline__ void ly_stage_f4(const int total, const int tid, const float* safe, SrcFn src, DstFn dst) {
  for (int base = tid; base < total; base += LY_THREADS * U) {
    f32x4 v[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * LY_THREADS;
      const float* p = idx < total ? src(idx) : nullptr;
      ok[u] = p != nullptr;
      v[u] = ly_ldg4(ok[u] ? p : safe);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * LY_THREADS;
      if (idx < total) dst(idx, ok[u] ? v[u] : ly_zero4());
    }
  }
}
