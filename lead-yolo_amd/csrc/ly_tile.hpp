// bf16x3 contraction core (gfx950): fp32-grade products on the bf16 matrix pipe.
//
// Every fp32 operand x is split as x = hi + lo (+ O(2^-16 |x|)), hi = bf16(x), lo = bf16(x - hi), and
//      w * x  ~=  w_hi*x_hi + w_hi*x_lo + w_lo*x_hi          (w_lo*x_lo ~ 2^-16 relative is dropped)
// is accumulated in fp32 by three v_mfma_f32_16x16x32_bf16 per 32 values of K.  Relative error per
// product ~2^-16 (vs 2^-8 for plain bf16), i.e. far inside the 1e-3 parity budget, at 16/3 = 5.3x the
// rate of the exact v_mfma_f32_16x16x4_f32 pipe (MI355X_MICROARCH.md: f32 MFMA = 1/16 of bf16).
//
// Tile convention (both operands K-contiguous, same as the fp32 core in ly_common.hpp):
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
#include "ly_common.hpp"
#include "ly_params.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// -------------------------------------------------------------------------------------------------
// Storage dtypes.  Every kernel is a template over the activation element type T:
//   float  : fp32 storage, bf16x3 products (two planes hi/lo of both operands, 3 MFMAs per k-step)
//   __bf16 : bf16 storage (BASELINE configs[2]-[4]), plain bf16 products (ONE plane, 1 MFMA per k-step)
// Accumulation, BatchNorm statistics, attention tables and parameters are fp32 in both.
// LyT<T>::PL = operand planes, VW = elements of one 16-byte vector, R4 / RV = raw 4-element / 16-byte register images.
// -------------------------------------------------------------------------------------------------
// dtype codes LY_F32 / LY_BF16: include/lead_yolo_hip.h
typedef unsigned int ly_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int ly_u32x4 __attribute__((ext_vector_type(4)));
template <typename T> struct LyT;
template <> struct LyT<float> {
  static constexpr int PL = 2, VW = 4;
  static constexpr bool BF = false;
  typedef f32x4 R4;
  typedef f32x4 RV;
};
template <> struct LyT<__bf16> {
  static constexpr int PL = 1, VW = 8;
  static constexpr bool BF = true;
  typedef ly_u32x2 R4;
  typedef ly_u32x4 RV;
};
// uint8 IMAGE source (LY_GATHER_PATCH_NCHW_U8): 4 pixels per 4-byte vector, value/255 on load, contracted like an fp32 source
template <> struct LyT<unsigned char> {
  static constexpr int PL = 2, VW = 4;
  static constexpr bool BF = false;
  typedef unsigned R4;
  typedef unsigned RV;
};

// 16-bit IMAGE sources (LY_GATHER_PATCH_NCHW_BF16 / _F16: the `im.half()` batch of a reduced-precision forward, val.py:207): 4 pixels per
// 8-byte vector, widened on the way into LDS and contracted like an fp32 source — the fp32 copy of the batch never exists
struct ly_bf16img { unsigned short v; };
struct ly_f16img { unsigned short v; };
struct ly_h4raw { ly_u32x2 r; };
template <> struct LyT<ly_bf16img> {
  static constexpr int PL = 2, VW = 4;
  static constexpr bool BF = false;
  typedef ly_u32x2 R4;
  typedef ly_u32x2 RV;
};
template <> struct LyT<ly_f16img> {
  static constexpr int PL = 2, VW = 4;
  static constexpr bool BF = false;
  typedef ly_h4raw R4;
  typedef ly_h4raw RV;
};

// runs `stmt` with T bound to the element type selected by the C ABI's dtype code
#define LY_WITH_T(dtype, ...)                                 \
  do {                                                        \
    if ((dtype) == LY_BF16) { using T = __bf16; __VA_ARGS__; } \
    else { using T = float; __VA_ARGS__; }                    \
  } while (0)
#define LY_CHECK_DTYPE(dtype, who) LY_CHECK((dtype) == LY_F32 || (dtype) == LY_BF16, who ": unknown dtype %d", (dtype))

__device__ __forceinline__ f32x4 ly_cvt4(const bf16x4 h) { return __builtin_convertvector(h, f32x4); }
__device__ __forceinline__ bf16x4 ly_cvtb4(const f32x4 v) { return __builtin_convertvector(v, bf16x4); }

// 4 consecutive elements <-> fp32 registers (fp32: one 16-byte access; bf16: one 8-byte access + conversion)
template <typename T> __device__ __forceinline__ f32x4 ly_ld4(const T* p);
template <> __device__ __forceinline__ f32x4 ly_ld4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 ly_ld4<__bf16>(const __bf16* p) { return ly_cvt4(*reinterpret_cast<const bf16x4*>(p)); }
template <typename T> __device__ __forceinline__ void ly_st4(T* p, const f32x4 v);
template <> __device__ __forceinline__ void ly_st4<float>(float* p, const f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void ly_st4<__bf16>(__bf16* p, const f32x4 v) { *reinterpret_cast<bf16x4*>(p) = ly_cvtb4(v); }
template <typename T> __device__ __forceinline__ float ly_ld1(const T* p) { return (float)*p; }
template <typename T> __device__ __forceinline__ void ly_st1(T* p, float v) { *p = (T)v; }

// raw (unconverted) register images: what a prefetch holds while the loads are in flight
template <typename T> __device__ __forceinline__ typename LyT<T>::R4 ly_ldr4(const T* p) { return *reinterpret_cast<const typename LyT<T>::R4*>(p); }
template <typename T> __device__ __forceinline__ typename LyT<T>::RV ly_ldrv(const T* p) { return *reinterpret_cast<const typename LyT<T>::RV*>(p); }
__device__ __forceinline__ f32x4 ly_r4_f32(const f32x4 r) { return r; }
__device__ __forceinline__ f32x4 ly_r4_f32(const ly_u32x2 r) { return ly_cvt4(__builtin_bit_cast(bf16x4, r)); }
__device__ __forceinline__ void ly_zero_raw(f32x4& r) { r = (f32x4){0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ void ly_zero_raw(ly_u32x2& r) { r = (ly_u32x2){0u, 0u}; }
__device__ __forceinline__ void ly_zero_raw(ly_u32x4& r) { r = (ly_u32x4){0u, 0u, 0u, 0u}; }
__device__ __forceinline__ void ly_zero_raw(unsigned& r) { r = 0u; }
__device__ __forceinline__ void ly_zero_raw(ly_h4raw& r) { r.r = (ly_u32x2){0u, 0u}; }
__device__ __forceinline__ f32x4 ly_u8x4_f32(unsigned r) {          // 4 image bytes -> pixel values / 255 (train.py:309 `imgs.float() / 255`)
  return (f32x4){(float)(r & 255u), (float)((r >> 8) & 255u), (float)((r >> 16) & 255u), (float)(r >> 24)} * (1.f / 255.f);
}
// 16-byte raw vector <-> VW/4 fp32 quads
__device__ __forceinline__ void ly_rv_unpack(const f32x4 r, f32x4 (&q)[1]) { q[0] = r; }
__device__ __forceinline__ void ly_rv_unpack(const ly_u32x4 r, f32x4 (&q)[2]) {
  const bf16x8 b = __builtin_bit_cast(bf16x8, r);
  q[0] = ly_cvt4(__builtin_shufflevector(b, b, 0, 1, 2, 3));
  q[1] = ly_cvt4(__builtin_shufflevector(b, b, 4, 5, 6, 7));
}
__device__ __forceinline__ f32x4 ly_rv_pack(const f32x4 (&q)[1], f32x4*) { return q[0]; }
__device__ __forceinline__ ly_u32x4 ly_rv_pack(const f32x4 (&q)[2], ly_u32x4*) {
  const bf16x4 a = ly_cvtb4(q[0]), b = ly_cvtb4(q[1]);
  return __builtin_bit_cast(ly_u32x4, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

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

// raw register images into the LDS operand image: fp32 is split into the two planes, bf16 is stored as it is (one plane)
__device__ __forceinline__ void ly_lds_put_r4(char* hi_plane, char* lo_plane, int row_byte, int c, const f32x4 v) { ly_lds_put4(hi_plane, lo_plane, row_byte, c, v); }
__device__ __forceinline__ void ly_lds_put_r4(char* hi_plane, char*, int row_byte, int c, const ly_u32x2 v) {
  *reinterpret_cast<ly_u32x2*>(hi_plane + row_byte + 2 * c) = v;
}
__device__ __forceinline__ void ly_lds_put_rv(char* hi_plane, char* lo_plane, int row_byte, int c, const f32x4 v) { ly_lds_put4(hi_plane, lo_plane, row_byte, c, v); }
__device__ __forceinline__ void ly_lds_put_rv(char* hi_plane, char*, int row_byte, int c, const ly_u32x4 v) {
  *reinterpret_cast<ly_u32x4*>(hi_plane + row_byte + 2 * c) = v;
}
__device__ __forceinline__ void ly_lds_put_rv(char* hi_plane, char* lo_plane, int row_byte, int c, const unsigned v) { ly_lds_put4(hi_plane, lo_plane, row_byte, c, ly_u8x4_f32(v)); }
__device__ __forceinline__ void ly_lds_put_rv(char* hi_plane, char* lo_plane, int row_byte, int c, const ly_u32x2 v) {      // bf16 image
  ly_lds_put4(hi_plane, lo_plane, row_byte, c, ly_cvt4(__builtin_bit_cast(bf16x4, v)));
}
typedef _Float16 ly_f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ly_lds_put_rv(char* hi_plane, char* lo_plane, int row_byte, int c, const ly_h4raw v) {      // fp16 image
  ly_lds_put4(hi_plane, lo_plane, row_byte, c, __builtin_convertvector(__builtin_bit_cast(ly_f16x4, v.r), f32x4));
}
// fp32 quad of channels c..c+3 into the operand image of a PL-plane kernel
template <int PL>
__device__ __forceinline__ void ly_lds_put_f32(char* hi_plane, char* lo_plane, int row_byte, int c, const f32x4 v) {
  if constexpr (PL == 2) ly_lds_put4(hi_plane, lo_plane, row_byte, c, v);
  else *reinterpret_cast<bf16x4*>(hi_plane + row_byte + 2 * c) = ly_cvtb4(v);
}

// B-operand fragment of k-step s for the row at `row_byte`
__device__ __forceinline__ bf16x8 ly_lds_frag(const char* plane, int row_byte, int s, int lq) {
  const char* p = plane + row_byte + 2 * (32 * s + 4 * lq);
  const bf16x4 a = *reinterpret_cast<const bf16x4*>(p);
  const bf16x4 b = *reinterpret_cast<const bf16x4*>(p + 32);
  return ly_cat8(a, b);
}

// -------------------------------------------------------------------------------------------------
// Lane-group LDS operand image (round 6).  The image above keeps the lane's two 4-channel groups of a k-step 32 bytes apart in ONE row, meant to
// be read as two ds_read_b64 (64 banks, conflict-free with RS/16 odd).  hipcc's load/store optimiser fuses such pairs — and the reads of
// neighbouring k-steps and pixel tiles — into ds_read2_b64, which the LDS serves in 16-lane groups over 32 banks at HALF the rate (MI355X
// guide, LDS table: 8 cycles per 16 bytes against 4 for one ds_read_b128), and for which RS/8 even is a 2-way conflict on every access: 16
// LDS cycles per fragment where 4 were designed (SQ_LDS_BANK_CONFLICT = 25 % of ly_conv3x3's cycles, 2x SQ_ACTIVE_INST_LDS in the 1x1 GEMM:
// profiles/r05_train_bf16_pmc_wait.txt — every ds_read of those kernels was a ds_read2_b64).  Two ds_read_b64 kept apart by construction
// (planes of 16 channels, > 2040 bytes between any two reads of a lane) were measured as well: conflicts gone, GEMM -2 %, but two reads and
// two addresses per fragment pushed ly_conv3x3 at three waves per SIMD into spilling (0.46 -> 0.58 ms per step).
// Here the lane's 8 k-values of a k-step are 16 CONTIGUOUS bytes and each lane group q = lane >> 4 has a plane of its own:
//      byte offset of channel c of row r = ((c & 15) >> 2) * ps + r * rsq + 16 * (c >> 5) + 8 * ((c >> 4) & 1) + 2 * (c & 3)
//   * a fragment (k-step s) is ONE ds_read_b128 at q * ps + r * rsq + 16 s: nothing to fuse, one address, 4 LDS cycles;
//   * rsq = 16 x odd and ps = 0 (mod 256): a ds_read_b128 is served in four groups of 16 lanes, each holding all 16 rows l & 15 of the fragment
//     (8 from lane group 2g, 8 from 2g + 1) — 16 consecutive rows x an odd number of 16-byte slots = the 16 slots of the 256-byte bank row once;
//   * staging writes: 4 channels = 8 bytes; an 8-channel vector (c % 8 == 0) is two 8-byte pieces in planes q and q + 1 (ds_write_b64; the
//     two planes a 16-lane store group touches alias in the banks: 2-way, on a path that runs once per tile, not once per MFMA).
// The lane's k-set is unchanged (k = 32s + 16(j>>2) + 4q + (j&3)): weights, accumulator chaining and every other operand order stay as they are.
// -------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int ly_qrs(int ks) { return 16 * ((ks & 1) ? ks + 2 : ks + 1); }     // bytes per row of a plane holding ks k-steps
__host__ __device__ constexpr int ly_qps(int bytes) { return (bytes + 255) / 256 * 256; }           // plane stride for a plane of `bytes`
__device__ __forceinline__ int ly_img_off(const int row_byte, const int ps, const int c) {
  return ((c & 15) >> 2) * ps + row_byte + 16 * (c >> 5) + 8 * ((c >> 4) & 1) + 2 * (c & 3);
}
// B-operand fragment of k-step s for the row at row_byte = r * rsq
__device__ __forceinline__ bf16x8 ly_img_frag(const char* img, const int row_byte, const int ps, const int s, const int lq) {
  return *reinterpret_cast<const bf16x8*>(img + lq * ps + row_byte + 16 * s);
}
// 4 consecutive channels (c % 4 == 0) as 8 bytes of bf16
__device__ __forceinline__ void ly_img_put4b(char* img, const int row_byte, const int ps, const int c, const ly_u32x2 v) {
  *reinterpret_cast<ly_u32x2*>(img + ly_img_off(row_byte, ps, c)) = v;
}
// 16-byte vector of VW consecutive elements starting at channel c (c % VW == 0) into the image: bf16 storage one image, fp32 split in two
__device__ __forceinline__ void ly_img_put_rv(char* hi_img, char*, const int row_byte, const int ps, const int c, const ly_u32x4 v) {
  const int o = ly_img_off(row_byte, ps, c);
  *reinterpret_cast<ly_u32x2*>(hi_img + o) = (ly_u32x2){v[0], v[1]};
  *reinterpret_cast<ly_u32x2*>(hi_img + o + ps) = (ly_u32x2){v[2], v[3]};            // channels c + 4 .. c + 7: the next lane group's plane, same place
}
__device__ __forceinline__ void ly_img_put_rv(char* hi_img, char* lo_img, const int row_byte, const int ps, const int c, const f32x4 v) {
  bf16x4 h, l;
  ly_split4(v, h, l);
  const int o = ly_img_off(row_byte, ps, c);
  *reinterpret_cast<bf16x4*>(hi_img + o) = h;
  *reinterpret_cast<bf16x4*>(lo_img + o) = l;
}
__device__ __forceinline__ void ly_img_put_rv(char* hi_img, char* lo_img, const int row_byte, const int ps, const int c, const unsigned v) {
  ly_img_put_rv(hi_img, lo_img, row_byte, ps, c, ly_u8x4_f32(v));
}
__device__ __forceinline__ void ly_img_put_rv(char* hi_img, char* lo_img, const int row_byte, const int ps, const int c, const ly_u32x2 v) {      // bf16 image
  ly_img_put_rv(hi_img, lo_img, row_byte, ps, c, ly_cvt4(__builtin_bit_cast(bf16x4, v)));
}
__device__ __forceinline__ void ly_img_put_rv(char* hi_img, char* lo_img, const int row_byte, const int ps, const int c, const ly_h4raw v) {      // fp16 image
  ly_img_put_rv(hi_img, lo_img, row_byte, ps, c, __builtin_convertvector(__builtin_bit_cast(ly_f16x4, v.r), f32x4));
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


// PL-plane weight fragment / contraction step (PL = 2: bf16x3, PL = 1: plain bf16); packed by pack.frag_pack3(planes=PL):
// uint4 wpk[((t*S + s)*PL + plane)*64 + lane]
template <int PL> struct LyWF;
template <> struct LyWF<2> { bf16x8 hi, lo; };
template <> struct LyWF<1> { bf16x8 hi; };
template <int PL>
__device__ __forceinline__ LyWF<PL> ly_wfragp(const uint4* __restrict__ wpk, long tile_step, int lane) {
  LyWF<PL> f;
  f.hi = __builtin_bit_cast(bf16x8, wpk[(tile_step * PL) * 64 + lane]);
  if constexpr (PL == 2) f.lo = __builtin_bit_cast(bf16x8, wpk[(tile_step * PL + 1) * 64 + lane]);
  return f;
}
template <int PL>
__device__ __forceinline__ f32x4 ly_mfmap(const LyWF<PL>& w, const bf16x8 xhi, const bf16x8 xlo, f32x4 acc) {
  if constexpr (PL == 2) return ly_mfma3(w.hi, w.lo, xhi, xlo, acc);
  else return ly_mfma_bf16(w.hi, xhi, acc);
}
// operand-by-operand form (wgrad: both operands come from LDS)
template <int PL>
__device__ __forceinline__ f32x4 ly_mfmapp(const bf16x8 ahi, const bf16x8 alo, const bf16x8 bhi, const bf16x8 blo, f32x4 acc) {
  if constexpr (PL == 2) return ly_mfma3(ahi, alo, bhi, blo, acc);
  else return ly_mfma_bf16(ahi, bhi, acc);
}

// Cooperative staging of `total` raw items (register image R: 16-byte vector or 4-element group) by the whole block, U loads
// in flight per thread.  src(idx) returns the global address of item idx or nullptr (-> zeros); dst(idx, v) consumes it.
// All U loads of a batch are issued back to back from clamped addresses (no branch around a load),
// so a thread pays one memory latency per batch instead of one per item.
template <int U, typename R, class SrcFn, class DstFn>
__device__ __forceinline__ void ly_stage_raw(const int total, const int tid, const void* safe, SrcFn src, DstFn dst) {
  for (int base = tid; base < total; base += LY_THREADS * U) {
    R v[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * LY_THREADS;
      const void* p = idx < total ? src(idx) : nullptr;
      ok[u] = p != nullptr;
      v[u] = *reinterpret_cast<const R*>(ok[u] ? p : safe);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * LY_THREADS;
      if (!ok[u]) ly_zero_raw(v[u]);
      if (idx < total) dst(idx, v[u]);
    }
  }
}
