// RFCBAMConv kernel_size 3, "lane = channel" core (reference models/rfa.py:101-129), gfx950.
//
// The depthwise `generate` conv produces, per output pixel p and input channel c, nine values (one per tap t)
//      v[p, t, c] = sum_u Wd[c, t, u] * x[patch(p, u), c]              (81 MAC per pixel and channel)
// that the reference materialises as a 9x expanded tensor.  Every kernel of this family REGENERATES them on chip.  The first
// generation of kernels (ly_rfcbam3.hip, ly_attention.hip) mapped lanes to pixels, which makes the 81 weights of a channel
// wave-uniform: they came out of LDS as broadcast reads and the LDS pipe, not the VALU, set the time.  Here
//      lane = channel (32 channels x 2 pixel streams per wave),
// so the 81 weights (+ 9 folded biases) of the lane's channel sit in 92 registers for a whole 32-channel chunk, the x patch is
// read from an fp32 LDS tile with conflict-free per-lane reads, two horizontally adjacent output pixels are computed together (one
// weight register feeds both), and whatever has to change hands
// between the channel-parallel VALU phase and a pixel-parallel phase (MFMA operand, channel reductions) goes through ONE
// K-major LDS tile  [k = t*32 + c][pixel]  whose pixel pairs are stored as one dword / qword per lane.
//
// unit of work: (tile of <= 64 output pixels, TH x TW with TW even) x (chunk of 32 channels).
#pragma once
#include "ly_tile.hpp"

#define RC_CB 32                 // channels per chunk
#define RC_TP 64                 // output-pixel slots per tile
#define RC_KR (9 * RC_CB)        // operand rows of a chunk: k = t*32 + c
#define RC_WQ 100                // floats per channel in the generate-weight image: w[t][u] (81), b[t] (9), a[t] (9), 1 pad  (25 x 16 bytes)
#define RC_WQ_FOLD 92             // ... of which the folded (inference) form reads the first 92: v = b + sum w x
#define RC_MAXPOS 320            // input positions of a tile (IH*IW) the staging registers are sized for (8x8 / 4x16 tiles at stride 2)

typedef short ly_s16x4 __attribute__((ext_vector_type(4)));

// acc + x * w[sel] for the two pixels of a pair (x) and ONE weight (element sel of the register pair w).
// These were hand-written v_pk_fma_f32 with the weight broadcast by op_sel / op_sel_hi (one packed instruction per weight and pixel pair).
// Alone on a SIMD they are exact; MEASURED: whenever another wave of the SIMD issues MFMAs (a second block of the same kernel in its
// contraction phase, a GEMM of another stream) their results come back slightly wrong and different from run to run — 28 of 30 runs of the
// statistics pass beside a bf16 matmul stream, 0 of 30 with the plain FMAs below; padding the LDS allocations apart changed nothing, one
// block per CU hid it.  The compiler's v_fmac_f32 cost nothing at two waves per SIMD (L17 forward 225 vs 223 us), so the asm is gone.
// RC_ASM_FMA (defined by ly_rf3c_bwd.hip only) selects the packed asm form: those kernels hold 180-250 live registers and the opaque
// instructions keep the compiler's scheduler from spilling 170-310 of them; they run ONE wave per SIMD and own their CU (whole-LDS launch),
// so no other wave can issue an MFMA beside them.
#ifndef RC_ASM_FMA
__device__ __forceinline__ f32x2 rc_pkfma(const f32x2 x, const f32x2 w, const f32x2 acc, const int sel) {
  const float ww = w[sel];
  return (f32x2){__builtin_fmaf(x[0], ww, acc[0]), __builtin_fmaf(x[1], ww, acc[1])};
}
// x * w[sel] + b[selb]  (first term of a chain)
__device__ __forceinline__ f32x2 rc_pkfma_b(const f32x2 x, const f32x2 w, const f32x2 b, const int sel, const int selb) {
  const float ww = w[sel], bb = b[selb];
  return (f32x2){__builtin_fmaf(x[0], ww, bb), __builtin_fmaf(x[1], ww, bb)};
}
// x * w[sel]
__device__ __forceinline__ f32x2 rc_pkmul(const f32x2 x, const f32x2 w, const int sel) {
  const float ww = w[sel];
  return (f32x2){x[0] * ww, x[1] * ww};
}
#else
__device__ __forceinline__ f32x2 rc_pkfma(const f32x2 x, const f32x2 w, f32x2 acc, const int sel) {
  if (sel) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(x), "v"(w));
  else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(x), "v"(w));
  return acc;
}
__device__ __forceinline__ f32x2 rc_pkfma_b(const f32x2 x, const f32x2 w, const f32x2 b, const int sel, const int selb) {
  f32x2 r;
  if (sel) {
    if (selb) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(x), "v"(w), "v"(b));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(x), "v"(w), "v"(b));
  } else {
    if (selb) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(x), "v"(w), "v"(b));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(x), "v"(w), "v"(b));
  }
  return r;
}
__device__ __forceinline__ f32x2 rc_pkmul(const f32x2 x, const f32x2 w, const int sel) {
  f32x2 r;
  if (sel) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(x), "v"(w));
  else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(w));
  return r;
}
#endif

// the lane's generate weights.  Image: float wq[C/32 chunks][25][32 channels][4] — element i = 4*q + e of channel c at
// ((chunk*25 + q)*32 + (c & 31))*4 + e — so that the 32 lanes of a half wave read 512 contiguous bytes per load (a per-channel
// [C][100] image costs one cache line per LANE and load: measured, the kernels were bound by the texture-address path).
// Elements: w[t][u] at t*9 + u, b[t] at 81 + t, a[t] at 90 + t.  Two forms of the same image:
//   folded (inference):  w = generate.0.weight * bn_scale, b = bn_shift, a unused        v = b + sum_u w x                (RAW = false)
//   raw (training):      w = generate.0.weight, a = bn_scale, b = bn_shift              u = sum_u w x,  v = a*u + b       (RAW = true)
// The raw form is what forward AND backward of a training step evaluate, instruction for instruction: the backward recomputes G and
// compares it with the forward's channel maximum for equality, and needs the pre-BatchNorm value u for the BatchNorm gradient.
// Element i is pair i>>1, dword i&1 of the register image.
struct RcW {
  f32x2 p[RC_WQ / 2];
};
template <bool RAW>
__device__ __forceinline__ void rc_load_w(RcW& w, const float* __restrict__ wq, int c0, int c) {
  const float* src = wq + ((long)(c0 >> 5) * (RC_WQ / 4) * RC_CB + c) * 4;
#pragma unroll
  for (int i = 0; i < (RAW ? RC_WQ : RC_WQ_FOLD) / 4; ++i) {
    const f32x4 v = ly_ldg4(src + i * (RC_CB * 4));
    w.p[2 * i] = (f32x2){v[0], v[1]};
    w.p[2 * i + 1] = (f32x2){v[2], v[3]};
  }
}
__device__ __forceinline__ float rc_relu(float v) { return fmaxf(v, 0.f); }

// folded: v[t] (pixel pair) = b[t] + sum_u w[t][u] * x[u];  raw: u[t] = sum_u w[t][u] * x[u].  Nine independent chains, interleaved so
// that no packed FMA waits for its predecessor
template <bool RAW>
__device__ __forceinline__ void rc_generate(const RcW& w, const f32x2 (&x)[9], f32x2 (&a)[9]) {
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    if constexpr (!RAW) a[t] = rc_pkfma_b(x[0], w.p[(t * 9) >> 1], w.p[(81 + t) >> 1], (t * 9) & 1, (81 + t) & 1);
    else a[t] = rc_pkmul(x[0], w.p[(t * 9) >> 1], (t * 9) & 1);
  }
#pragma unroll
  for (int u = 1; u < 9; ++u)
#pragma unroll
    for (int t = 0; t < 9; ++t) a[t] = rc_pkfma(x[u], w.p[(t * 9 + u) >> 1], a[t], (t * 9 + u) & 1);
}
// raw form: v[t] = a[t] * u[t] + b[t]
__device__ __forceinline__ void rc_affine(const RcW& w, const f32x2 (&u)[9], f32x2 (&v)[9]) {
#pragma unroll
  for (int t = 0; t < 9; ++t) v[t] = rc_pkfma_b(u[t], w.p[(90 + t) >> 1], w.p[(81 + t) >> 1], (90 + t) & 1, (81 + t) & 1);
}
// the BatchNorm output v of `generate` for a pixel pair, either form
template <bool RAW>
__device__ __forceinline__ void rc_gen_bn(const RcW& w, const f32x2 (&x)[9], f32x2 (&v)[9]) {
  if constexpr (RAW) {
    f32x2 u[9];
    rc_generate<true>(w, x, u);
    rc_affine(w, u, v);
  } else {
    rc_generate<false>(w, x, v);
  }
}

// tile geometry shared by the kernels of the family
struct RcGeom {
  int s, TH, TW, IH, IW, NPX;
};
__device__ __forceinline__ RcGeom rc_geom(int s, int TH, int TW) {
  RcGeom g;
  g.s = s; g.TH = TH; g.TW = TW;
  g.IH = s * (TH - 1) + 3; g.IW = s * (TW - 1) + 3;
  g.NPX = TH * TW;
  return g;
}

// ---- x tile: fp32 [IH*IW][32] -----------------------------------------------------------------------
// staging plan of one thread: items (input position, VW-channel group), NV per thread
template <typename T, int NTHR = LY_THREADS> struct RcStage {
  static constexpr int VW = LyT<T>::VW;
  static constexpr int GP = RC_CB / VW;                                        // channel groups per position: 8 (fp32) / 4 (bf16)
  static constexpr int NV = (RC_MAXPOS * GP + NTHR - 1) / NTHR;                // 256 threads: 10 / 5
  int soff[NV];        // element offset of the item in x (channel c0 = 0), -1: outside the image / no item
  int doff[NV];        // float index in the LDS tile, -1: no item
  typename LyT<T>::RV pv[NV];
};
template <typename T, int NTHR>
__device__ __forceinline__ void rc_stage_plan(RcStage<T, NTHR>& S, const RcGeom& g, int tid, int n, int H, int W, int ldx, int iy0, int ix0) {
  constexpr int GP = RcStage<T, NTHR>::GP, VW = RcStage<T, NTHR>::VW;
  const int items = g.IH * g.IW * GP;
#pragma unroll
  for (int e = 0; e < RcStage<T, NTHR>::NV; ++e) {
    const int idx = tid + e * NTHR;
    int so = -1, dd = -1;
    if (idx < items) {
      const int ip = idx / GP, cv = idx - ip * GP;
      const int r = ip / g.IW, q = ip - r * g.IW;
      const int iy = iy0 + r, ix = ix0 + q;
      dd = ip * RC_CB + VW * cv;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) so = (int)((((long)n * H + iy) * W + ix) * ldx + VW * cv);
    }
    S.soff[e] = so; S.doff[e] = dd;
  }
}
// per-thread, tile-independent part of the plan for kernels that walk many tiles (no divisions per tile): item -> (row, column) of the input
// tile and its element offset relative to the tile's first position
template <typename T, int NTHR = LY_THREADS> struct RcStageFix {
  int rq[RcStage<T, NTHR>::NV];       // (r << 16) | q, -1: no item
  int rel[RcStage<T, NTHR>::NV];      // (r*W + q)*ldx + VW*cv
};
template <typename T, int NTHR>
__device__ __forceinline__ void rc_stage_fix(RcStageFix<T, NTHR>& F, RcStage<T, NTHR>& S, const RcGeom& g, int tid, int W, int ldx) {
  constexpr int GP = RcStage<T, NTHR>::GP, VW = RcStage<T, NTHR>::VW;
  const int items = g.IH * g.IW * GP;
#pragma unroll
  for (int e = 0; e < RcStage<T, NTHR>::NV; ++e) {
    const int idx = tid + e * NTHR;
    F.rq[e] = -1; F.rel[e] = 0; S.doff[e] = -1;
    if (idx < items) {
      const int ip = idx / GP, cv = idx - ip * GP;
      const int r = ip / g.IW, q = ip - r * g.IW;
      F.rq[e] = (r << 16) | q;
      F.rel[e] = (r * W + q) * ldx + VW * cv;
      S.doff[e] = ip * RC_CB + VW * cv;
    }
  }
}
// retarget the plan at the tile whose first input position is (iy0, ix0) of image n; base = ((n*H + iy0)*W + ix0)*ldx (may be negative)
template <typename T, int NTHR>
__device__ __forceinline__ void rc_stage_retarget(RcStage<T, NTHR>& S, const RcStageFix<T, NTHR>& F, int base, int iy0, int ix0, int H, int W) {
#pragma unroll
  for (int e = 0; e < RcStage<T, NTHR>::NV; ++e) {
    const int iy = iy0 + (F.rq[e] >> 16), ix = ix0 + (F.rq[e] & 0xffff);
    const bool ok = F.rq[e] >= 0 && iy >= 0 && iy < H && ix >= 0 && ix < W;
    S.soff[e] = ok ? base + F.rel[e] : -1;
  }
}
template <typename T, int NTHR>
__device__ __forceinline__ void rc_stage_load(RcStage<T, NTHR>& S, const T* __restrict__ x, int c0) {
#pragma unroll
  for (int e = 0; e < RcStage<T, NTHR>::NV; ++e) S.pv[e] = ly_ldrv<T>(S.soff[e] >= 0 ? x + S.soff[e] + c0 : x);      // clamped address, never a branch around a load
}
template <typename T, int NTHR>
__device__ __forceinline__ void rc_stage_store(const RcStage<T, NTHR>& S, float* __restrict__ xs) {
#pragma unroll
  for (int e = 0; e < RcStage<T, NTHR>::NV; ++e)
    if (S.doff[e] >= 0) {
      f32x4 q[RcStage<T, NTHR>::VW / 4];
      ly_rv_unpack(S.pv[e], q);
      const bool ok = S.soff[e] >= 0;
#pragma unroll
      for (int i = 0; i < RcStage<T, NTHR>::VW / 4; ++i) *reinterpret_cast<f32x4*>(xs + S.doff[e] + 4 * i) = ok ? q[i] : ly_zero4();
    }
}

// the 9 patch values of a pixel pair: xp = the lane's address of the pair's first patch element (xs + pos0*32 + c, pos0 = (S*ly)*IW + S*lx),
// row = IW*32 floats; everything else is an immediate offset
template <int S>
__device__ __forceinline__ void rc_patch(const float* __restrict__ xp, int row, f32x2 (&x)[9]) {
#pragma unroll
  for (int uy = 0; uy < 3; ++uy) {
    const float* r = xp + uy * row;
#pragma unroll
    for (int ux = 0; ux < 3; ++ux) x[uy * 3 + ux] = (f32x2){r[ux * RC_CB], r[(ux + S) * RC_CB]};
  }
}
// tile position of pixel px (0 for pixel slots past the tile)
__device__ __forceinline__ int rc_pos0(const RcGeom& g, int px) {
  const int pxc = px < g.NPX ? px : 0;
  const int ly = pxc / g.TW, lx = pxc - ly * g.TW;
  return (g.s * ly) * g.IW + g.s * lx;
}

// ---- K-major bf16 operand tile [288][64 px], 128-byte rows, 8-byte chunks XOR-swizzled by the row -----------------
//   chunk' = chunk ^ sw(k),  sw(k) = 4*((k>>1)&3) ^ ((k>>3)&3)
// writes (32 lanes = 32 consecutive rows, same pixel pair): 16 chunk positions x 2 lanes: 2-way, free for ds_write_b32;
// transposed reads (ds_read_b64_tr_b16, a half wave = 8 consecutive rows x 4 chunks): all 64 banks, conflict-free.
__device__ __forceinline__ int rc_sw(int k) { return (((k >> 1) & 3) << 2) ^ ((k >> 3) & 3); }
__device__ __forceinline__ int rc_goff(int k, int px0) { return k * 128 + ((((px0 >> 2) ^ rc_sw(k)) << 3) | ((px0 & 2) << 1)); }

__device__ __forceinline__ unsigned rc_pack2(const f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }
__device__ __forceinline__ f32x2 rc_unpack2(const unsigned u) { return (f32x2){__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)}; }

// B-operand fragment (k-step = 32 rows starting at row k32, pixel tile j of 16) for the lane: two transposed reads
struct RcTr {
  int b0, b1, s0, s1, p;      // byte offsets of the lane's block row in the two 4-row blocks, their swizzles, its 8-byte chunk
};
__device__ __forceinline__ RcTr rc_tr_plan(int lane) {
  const int li = lane & 15, q = lane >> 4;
  const int r0 = 4 * q + (li >> 2), r1 = 16 + r0;
  RcTr t;
  t.b0 = r0 * 128; t.b1 = r1 * 128;
  t.s0 = rc_sw(r0); t.s1 = rc_sw(r1);
  t.p = li & 3;
  return t;
}
__device__ __forceinline__ bf16x8 rc_tr_frag(const char* __restrict__ plane, const RcTr& t, int k32, int j) {
  typedef __attribute__((address_space(3))) ly_s16x4 lds_s16x4;
  const char* base = plane + k32 * 128;
  const ly_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + t.b0 + (((4 * j + t.p) ^ t.s0) << 3)));
  const ly_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + t.b1 + (((4 * j + t.p) ^ t.s1) << 3)));
  return ly_cat8(__builtin_bit_cast(bf16x4, a), __builtin_bit_cast(bf16x4, b));
}

// argument checks shared by the launchers of the family
static inline int rc_check_tile(const char* who, int C, int s, int TH, int TW, int ldx, const void* x) {
  LY_CHECK(C > 0 && (C % RC_CB) == 0, "%s: C=%d must be a multiple of %d", who, C, RC_CB);
  LY_CHECK(s == 1 || s == 2, "%s: stride %d is not built (1 or 2)", who, s);
  LY_CHECK(TH >= 1 && TW >= 2 && (TW & 1) == 0 && TH * TW <= RC_TP, "%s: bad tile %dx%d (TW even, TH*TW <= %d)", who, TH, TW, RC_TP);
  const int IH = s * (TH - 1) + 3, IW = s * (TW - 1) + 3;
  LY_CHECK(IH * IW <= RC_MAXPOS, "%s: the %dx%d tile reads %d input positions (max %d)", who, TH, TW, IH * IW, RC_MAXPOS);
  LY_CHECK((ldx & 7) == 0 && ((uintptr_t)x & 15) == 0, "%s: x must be 16-byte aligned with a row stride that is a multiple of 8", who);
  return 0;
}
