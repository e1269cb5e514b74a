// 3x3 / stride 1 / pad 1 convolution as an implicit GEMM + folded BN + activation, gfx950; storage dtype T = float (bf16x3
// products) or __bf16 (plain bf16 products), fp32 accumulation (ly_tile.hpp).
//
//   out[m, n] = act( scale[n] * sum_{tap, c} X[pix(m) + tap][c] * W[n][c][tap] + shift[n] )
//
// Replaces CA_Bottleneck.cv2 = Conv(c_, c_, 3, 1) (reference models/common.py:1617,1890-1910): no
// im2col buffer exists anywhere.  A block owns a TH x TW patch of one image (<= 128 pixels, chosen on
// the host to tile H x W with little waste).  Per 32-input-channel chunk it stages the (TH+2) x (TW+2)
// frame of the NHWC input as bf16 hi/lo planes in LDS with out-of-image positions written as ZEROS, so
// every MFMA B operand is a plain 2 x 8-byte LDS read at (pixel + tap offset): no border masks, tap
// offsets are wave-uniform.  K order = (tap, cin); weights bf16x3 frag-packed from
// pack.conv_taps_matrix(w, 32).
//
// Pipeline: the raw fp32 frame of chunk c+1 is loaded into registers before the 9-tap contraction of
// chunk c is issued and committed to LDS after it; weight fragments are fetched one tap ahead.
// The 4 waves split the output channels (WC = 4): each weight fragment is read by exactly one wave.
#include <cstdlib>
#include "ly_tile.hpp"
#include "ly_params.h"

#define LY_C3_NT 8                // max pixel tiles (128 pixels) per block
#define LY_C3_NV 6                // 16-byte vectors per thread per chunk: (TH+2)(TW+2)*8 <= 6*256  =>  frame <= 192 positions

// T = float: 32-channel chunks (8 float4 per frame position), two operand planes; T = __bf16: 64-channel chunks (8 x 16 bytes
// per position, the same bytes in flight per thread), one plane, two k-steps per tap.
// NTA > 0 (round 6): the patch has exactly NTA active pixel tiles per wave (80 x 80 maps: 8 / 4, 40 x 40 and 20 x 20 maps: 5) — the k-step's NTA operand
// fragments are read as ONE batch ahead of its MFMAs.  With the run-time guard `if (tile < ntv)` around every tile hipcc keeps each tile in a block
// of its own: read, s_waitcnt lgkmcnt(0), MT MFMAs — 136 full LDS round trips per chunk (PMC r05: 66 % of the wave cycles in s_waitcnt).
#ifndef LY_C3_NTA_WPE
#define LY_C3_NTA_WPE 3
#endif
template <typename T, int MT, int WC, int NTA = 0>
__global__ __launch_bounds__(LY_THREADS) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 2 ? (NTA > 0 ? LY_C3_NTA_WPE : 3) : 2))) void ly_conv3x3_kernel(const LyConv3Params P, const int gy, const int tiles_x, const int tiles_y, const int rp, const int ps) {
  using TR = LyT<T>;
  using RV = typename TR::RV;
  constexpr int VW = TR::VW, PL = TR::PL;
  constexpr int LY_CC = 8 * VW;                 // channels per chunk
  constexpr int KS = LY_CC / 32;                // k-steps per tap and chunk
  constexpr int LY_RSH = ly_qrs(KS);            // bytes per frame position in a plane of the lane-group operand image (ly_tile.hpp); ps = plane stride
  extern __shared__ f32x4 ly_smem4[];
  const int TH = P.TH, TW = P.TW, FW = TW + 2;
  const int frame = (TH + 2) * FW;
  char* hs_hi = reinterpret_cast<char*>(ly_smem4);
  char* hs_lo = hs_hi + (PL - 1) * 4 * ps;
  const T* const x = reinterpret_cast<const T*>(P.x);
  T* const out = reinterpret_cast<T*>(P.out);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  constexpr int WP = 4 / WC;                 // waves along pixels
  constexpr int NTW = LY_C3_NT / WP;         // pixel tiles per wave
  const int wc = wave % WC, wp = wave / WC;
  int b = ly_xcd_remap((int)blockIdx.x, (int)gridDim.x);   // neighbouring tiles (shared halo rows / columns) on one XCD's L2
  const int by = b % gy; b /= gy;
  const int tx = b % tiles_x; b /= tiles_x;
  const int ty = b % tiles_y;
  const int n_img = b / tiles_y;
  const int h0 = ty * TH, w0 = tx * TW;
  const int npx = TH * TW;
  const int ntv = (npx + 15) >> 4;                         // pixel tiles actually used (wave-uniform)
  const f32x4 zero = ly_zero4();
  const int C32 = (P.Cin + 31) >> 5;                       // k-steps per tap (weight layout: conv_taps_matrix(w, 32))
  const int NCH = (P.Cin + LY_CC - 1) / LY_CC;             // channel chunks
  const int S = 9 * C32;
  const int Tt = (P.N + 15) >> 4;
  const long img0 = (long)n_img * P.H * P.W;

  // per-lane pixel -> byte offset of its (0,0) tap in the frame, and its output row (or -1)
  int hb[NTW];
  long orow[NTW];
#pragma unroll
  for (int n = 0; n < NTW; ++n) {
    const int p = 16 * (wp * NTW + n) + li;
    const int pp = p < npx ? p : 0;
    const int r = pp / TW, c = pp - r * TW;
    hb[n] = r * rp + c * LY_RSH;
    const bool ok = p < npx && h0 + r < P.H && w0 + c < P.W;
    orow[n] = ok ? img0 + (long)(h0 + r) * P.W + (w0 + c) : -1;
  }

  // ---- staging: this thread's frame items (position, 16-byte channel group) --------------------------------
  const int items = frame * 8;
  int src[LY_C3_NV];                                        // element offset of the item's first channel of chunk 0, or -1 (31 bits: checked by the launcher)
  int dst[LY_C3_NV];                                        // byte offset of the item's position in the LDS frame (row pitch rp: see conv3_row_pitch)
#pragma unroll
  for (int e = 0; e < LY_C3_NV; ++e) {
    const int idx = tid + e * LY_THREADS;
    int off = -1;
    int dd = 0;
    if (idx < items) {
      const int pos = idx >> 3, c4 = idx & 7;
      const int fr = pos / FW, fc = pos - fr * FW;
      const int hh = h0 - 1 + fr, ww = w0 - 1 + fc;
      dd = fr * rp + fc * LY_RSH;
      if (hh >= 0 && hh < P.H && ww >= 0 && ww < P.W) off = (int)((img0 + (long)hh * P.W + ww) * P.ldx + VW * c4);
    }
    src[e] = off; dst[e] = dd;
  }
  RV pv[LY_C3_NV];
  // a vector is loaded when its FIRST channel is inside Cin: the channels beyond Cin it may carry (Cin % VW != 0: the partial
  // conv gradients of the MLPBlock backward) meet zero-padded weights, and ldx >= roundup(Cin, VW) is checked by the launcher
  auto prefetch = [&](int c0) {
#pragma unroll
    for (int e = 0; e < LY_C3_NV; ++e) {
      const int c4 = (tid + e * LY_THREADS) & 7;
      const bool ok = src[e] >= 0 && c0 + VW * c4 < P.Cin;
      pv[e] = ly_ldrv<T>(ok ? x + src[e] + c0 : x);      // clamped address, zero selected at commit
    }
  };
  auto commit = [&](int c0) {
#pragma unroll
    for (int e = 0; e < LY_C3_NV; ++e) {
      const int idx = tid + e * LY_THREADS;
      const int c4 = idx & 7;
      const bool ok = src[e] >= 0 && c0 + VW * c4 < P.Cin;
      RV v = pv[e];
      if (!ok) ly_zero_raw(v);
      if (idx < items) ly_img_put_rv(hs_hi, hs_lo, dst[e], ps, VW * c4, v);
    }
  };

  f32x4 acc[MT][NTW];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int n = 0; n < NTW; ++n) acc[t][n] = zero;
  long wbase[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int tt = (by * WC + wc) * MT + t;
    wbase[t] = (long)(tt < Tt ? tt : Tt - 1) * S;
  }
  const uint4* wpk = reinterpret_cast<const uint4*>(P.wp);
  ly_l2_warm(P.wp, (long)Tt * S * PL * 1024, P.stats ? reinterpret_cast<float*>(P.stats) : reinterpret_cast<float*>(P.out));

  // weight fragment of (tap, k-step ks of chunk cc); a ragged last chunk (Cin % 64 == 32) clamps the absent second step to the
  // first: its activations are staged as zeros
  auto wstep = [&](int tap, int cc, int ks) -> long {
    int st = KS * cc + ks;
    if (st >= C32) st = C32 - 1;
    return (long)tap * C32 + st;
  };
  LyWF<PL> wcur[MT], wnxt[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) wcur[t] = ly_wfragp<PL>(wpk, wbase[t], lane);      // (tap 0, chunk 0, step 0)

  prefetch(0);
  commit(0);
  __syncthreads();

  for (int cc = 0; cc < NCH; ++cc) {
    const bool more = cc + 1 < NCH;
    // unconditional (the last chunk re-requests itself): a load under a branch makes the compiler drain the whole queue
    // (s_waitcnt vmcnt(0)) after every tap, i.e. wait for this prefetch at tap 0 instead of hiding it behind nine taps of MFMAs
    prefetch((more ? cc + 1 : cc) * LY_CC);
#pragma unroll(MT * NTW <= 8 ? 1 : 9)
    for (int tap = 0; tap < 9; ++tap) {
      const int toff = (tap / 3) * rp + (tap % 3) * LY_RSH;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        // next fragment: next k-step of this tap, else next tap of this chunk, else tap 0 of the next chunk (clamped at the very end)
        {
          const bool last = ks == KS - 1;
          const int nk = last ? 0 : ks + 1;
          const int nt = last ? (tap < 8 ? tap + 1 : 0) : tap;
          const int nc = (last && tap == 8) ? (more ? cc + 1 : cc) : cc;
#pragma unroll
          for (int t = 0; t < MT; ++t) wnxt[t] = ly_wfragp<PL>(wpk, wbase[t] + wstep(nt, nc, nk), lane);
        }
        if constexpr (NTA > 0) {
          bf16x8 xh[NTA], xl[NTA];
#pragma unroll
          for (int n = 0; n < NTA; ++n) {
            xh[n] = ly_img_frag(hs_hi, hb[n] + toff, ps, ks, lq);
            xl[n] = xh[n];
            if constexpr (PL == 2) xl[n] = ly_img_frag(hs_lo, hb[n] + toff, ps, ks, lq);
          }
#pragma unroll
          for (int n = 0; n < NTA; ++n)
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t][n] = ly_mfmap<PL>(wcur[t], xh[n], xl[n], acc[t][n]);
        } else {
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
          if (wp * NTW + n < ntv) {
            const bf16x8 xh = ly_img_frag(hs_hi, hb[n] + toff, ps, ks, lq);
            bf16x8 xl = xh;
            if constexpr (PL == 2) xl = ly_img_frag(hs_lo, hb[n] + toff, ps, ks, lq);
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t][n] = ly_mfmap<PL>(wcur[t], xh, xl, acc[t][n]);
          }
        }
        }
#pragma unroll
        for (int t = 0; t < MT; ++t) wcur[t] = wnxt[t];
      }
    }
    if (more) {
      __syncthreads();            // every wave is done reading the frame of chunk cc
      commit((cc + 1) * LY_CC);
      __syncthreads();
    }
  }

  // ---- epilogue ---------------------------------------------------------------------------------
  const int act = P.act;
  const bool vec_ok = (P.ldo & 3) == 0;
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int tt = (by * WC + wc) * MT + t;
    const int c = 16 * tt + 4 * lq;
    if (tt >= Tt || c >= P.N) continue;
    float sc[4], sh[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = c + r < P.N;
      sc[r] = (ok && P.e_scale) ? P.e_scale[c + r] : 1.f;
      sh[r] = (ok && P.e_shift) ? P.e_shift[c + r] : 0.f;
    }
    f32x4 s1 = zero, s2 = zero;
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
      if (orow[n] < 0) continue;
      f32x4 u;
#pragma unroll
      for (int r = 0; r < 4; ++r) u[r] = acc[t][n][r] * sc[r] + sh[r];
      if (P.stats) {
        s1 += u;
        s2 += u * u;
        if (!out) continue;                                // pure statistics pass; with `out` the value is stored as well
      }
      const f32x4 v = ly_act4(u, act);
      T* o = out + orow[n] * P.ldo + c;
      if (vec_ok && c + 3 < P.N) {
        ly_st4<T>(o, v);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (c + r < P.N) ly_st1<T>(o + r, v[r]);
      }
    }
    if (P.stats) ly_stats_flush(P.stats, P.N, c, s1, s2);
  }
}

#ifndef LY_C3_PD
#define LY_C3_PD 3                // weight-fragment ring depth of the latency form
#endif
template <int V>
struct LyC3Ic { static constexpr int value = V; };

// The LATENCY form (bf16 storage; conv3_dispatch takes it for grids under two blocks per CU — the serving forward, where a CU holds about one
// block and nothing hides a wave's waits).  In the throughput form above every (LDS fragment read -> its two MFMAs) pair sits behind its own
// run-time "tile in use" branch and weight fragments arrive one step ahead: fine with three waves per SIMD, but alone a wave ran ~7x longer
// than its MFMAs.  Here the wave's tile count NC is a compile-time constant (switch over 0..NTW), so the NC reads of a step are issued
// together; weight fragments travel two steps ahead in a ring of three register sets; 223 / 164 registers = two waves per SIMD.
// bs=16, 64 channels per block (<2,2>): 40 x 40 x 128: 25.5 -> 19.8 us, 20 x 20 x 256: 32.5 -> 22.5 us.  With every CU holding three blocks
// (bs=64) the throughput form stays ahead (55 vs 55-69 us), hence the dispatch by grid size.
template <typename T, int MT, int WC>
__global__ __launch_bounds__(LY_THREADS) __attribute__((amdgpu_waves_per_eu(2))) void ly_conv3x3_lat_kernel(const LyConv3Params P, const int gy, const int tiles_x, const int tiles_y, const int rp, const int ps) {
  using TR = LyT<T>;
  using RV = typename TR::RV;
  constexpr int VW = TR::VW, PL = TR::PL;
  constexpr int LY_CC = 8 * VW;                 // channels per chunk
  constexpr int KS = LY_CC / 32;                // k-steps per tap and chunk
  constexpr int LY_RSH = ly_qrs(KS);            // bytes per frame position in a plane of the lane-group operand image (ly_tile.hpp); ps = plane stride
  extern __shared__ f32x4 ly_smem4[];
  const int TH = P.TH, TW = P.TW, FW = TW + 2;
  const int frame = (TH + 2) * FW;
  char* hs_hi = reinterpret_cast<char*>(ly_smem4);
  char* hs_lo = hs_hi + (PL - 1) * 4 * ps;
  const T* const x = reinterpret_cast<const T*>(P.x);
  T* const out = reinterpret_cast<T*>(P.out);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  constexpr int WP = 4 / WC;                 // waves along pixels
  constexpr int NTW = LY_C3_NT / WP;         // pixel tiles per wave
  const int wc = wave % WC, wp = wave / WC;
  int b = ly_xcd_remap((int)blockIdx.x, (int)gridDim.x);   // neighbouring tiles (shared halo rows / columns) on one XCD's L2
  const int by = b % gy; b /= gy;
  const int tx = b % tiles_x; b /= tiles_x;
  const int ty = b % tiles_y;
  const int n_img = b / tiles_y;
  const int h0 = ty * TH, w0 = tx * TW;
  const int npx = TH * TW;
  const int ntv = (npx + 15) >> 4;                         // pixel tiles actually used (wave-uniform)
  const f32x4 zero = ly_zero4();
  const int C32 = (P.Cin + 31) >> 5;                       // k-steps per tap (weight layout: conv_taps_matrix(w, 32))
  const int NCH = (P.Cin + LY_CC - 1) / LY_CC;             // channel chunks
  const int S = 9 * C32;
  const int Tt = (P.N + 15) >> 4;
  const long img0 = (long)n_img * P.H * P.W;

  // per-lane pixel -> byte offset of its (0,0) tap in the frame, and its output row (or -1)
  int hb[NTW];
#pragma unroll
  for (int n = 0; n < NTW; ++n) {
    const int p = 16 * (wp * NTW + n) + li;
    const int pp = p < npx ? p : 0;
    const int r = pp / TW, c = pp - r * TW;
    hb[n] = r * rp + c * LY_RSH;
  }
  // output row of the lane's pixel in tile n, or -1 (LAT: recomputed in the epilogue — eight 64-bit values held through the loop cost 16 registers)
  auto out_row = [&](int n) -> long {
    const int p = 16 * (wp * NTW + n) + li;
    const int pp = p < npx ? p : 0;
    const int r = pp / TW, c = pp - r * TW;
    const bool ok = p < npx && h0 + r < P.H && w0 + c < P.W;
    return ok ? img0 + (long)(h0 + r) * P.W + (w0 + c) : -1;
  };

  // ---- staging: this thread's frame items (position, 16-byte channel group) --------------------------------
  const int items = frame * 8;
  int src[LY_C3_NV];                                        // element offset of the item's first channel of chunk 0, or -1 (31 bits: checked by the launcher)
  int dst[LY_C3_NV];                                        // byte offset of the item's position in the LDS frame (row pitch rp: see conv3_row_pitch)
#pragma unroll
  for (int e = 0; e < LY_C3_NV; ++e) {
    const int idx = tid + e * LY_THREADS;
    int off = -1;
    int dd = 0;
    if (idx < items) {
      const int pos = idx >> 3, c4 = idx & 7;
      const int fr = pos / FW, fc = pos - fr * FW;
      const int hh = h0 - 1 + fr, ww = w0 - 1 + fc;
      dd = fr * rp + fc * LY_RSH;
      if (hh >= 0 && hh < P.H && ww >= 0 && ww < P.W) off = (int)((img0 + (long)hh * P.W + ww) * P.ldx + VW * c4);
    }
    src[e] = off; dst[e] = dd;
  }
  RV pv[LY_C3_NV];
  // a vector is loaded when its FIRST channel is inside Cin: the channels beyond Cin it may carry (Cin % VW != 0: the partial
  // conv gradients of the MLPBlock backward) meet zero-padded weights, and ldx >= roundup(Cin, VW) is checked by the launcher
  auto prefetch = [&](int c0) {
#pragma unroll
    for (int e = 0; e < LY_C3_NV; ++e) {
      const int c4 = (tid + e * LY_THREADS) & 7;
      const bool ok = src[e] >= 0 && c0 + VW * c4 < P.Cin;
      pv[e] = ly_ldrv<T>(ok ? x + src[e] + c0 : x);      // clamped address, zero selected at commit
    }
  };
  auto commit = [&](int c0) {
#pragma unroll
    for (int e = 0; e < LY_C3_NV; ++e) {
      const int idx = tid + e * LY_THREADS;
      const int c4 = idx & 7;
      const bool ok = src[e] >= 0 && c0 + VW * c4 < P.Cin;
      RV v = pv[e];
      if (!ok) ly_zero_raw(v);
      if (idx < items) ly_img_put_rv(hs_hi, hs_lo, dst[e], ps, VW * c4, v);
    }
  };

  f32x4 acc[MT][NTW];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int n = 0; n < NTW; ++n) acc[t][n] = zero;
  long wbase[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int tt = (by * WC + wc) * MT + t;
    wbase[t] = (long)(tt < Tt ? tt : Tt - 1) * S;
  }
  const uint4* wpk = reinterpret_cast<const uint4*>(P.wp);
  ly_l2_warm(P.wp, (long)Tt * S * PL * 1024, P.stats ? reinterpret_cast<float*>(P.stats) : reinterpret_cast<float*>(P.out));

  // weight fragment of (tap, k-step ks of chunk cc); a ragged last chunk (Cin % 64 == 32) clamps the absent second step to the
  // first: its activations are staged as zeros
  auto wstep = [&](int tap, int cc, int ks) -> long {
    int st = KS * cc + ks;
    if (st >= C32) st = C32 - 1;
    return (long)tap * C32 + st;
  };
  // Weight fragments travel PD - 1 steps ahead of their use in a ring of PD register sets (a step = one (tap, k-step): MT fragments, MT x NTW
  // MFMAs = 160-256 cycles).  One step ahead (the first form) left a wave waiting on L2 every step whenever fewer than ~4 waves shared the
  // SIMD: in the serving forward (320 blocks on 256 CUs) a block ran 10x longer than its MFMAs.  PD divides the 9 KS steps of a chunk, so the
  // slot of a step is a compile-time constant of the unrolled body.
  static_assert(PL == 1, "bf16 storage only");
  constexpr int PD = LY_C3_PD;
  constexpr int SPC = 9 * KS;                               // steps per chunk
  static_assert(SPC % PD == 0, "ring depth must divide the steps of a chunk");
  LyWF<PL> wq[PD][MT];
  // fragment index of step j (0 .. SPC + PD - 2) counted from the start of chunk cc; past the last chunk it clamps (never used)
  auto wstep_at = [&](int cc, int j, bool more) -> long {
    int c2 = cc;
    if (j >= SPC) { j -= SPC; c2 = more ? cc + 1 : cc; }
    return wstep(j / KS, c2, j % KS);
  };
  {
#pragma unroll
    for (int j = 0; j < PD - 1; ++j)
#pragma unroll
      for (int t = 0; t < MT; ++t) wq[j][t] = ly_wfragp<PL>(wpk, wbase[t] + wstep_at(0, j, NCH > 1), lane);
  }

  prefetch(0);
  commit(0);
  __syncthreads();

  // The contraction with the wave's pixel-tile count NC as a compile-time constant (dispatched below).  With a run-time test per tile every
  // (LDS fragment read -> its MFMAs) pair sat behind its own branch: the read's latency was exposed once per tile, hidden only by other waves —
  // and the serving forward runs these blocks one per CU.  Unrolled, the NC reads of a step are issued together ahead of its MFMAs.
  auto contract = [&](auto ncC) {
    constexpr int NC = decltype(ncC)::value;
    for (int cc = 0; cc < NCH; ++cc) {
      const bool more = cc + 1 < NCH;
      // unconditional (the last chunk re-requests itself): a load under a branch makes the compiler drain the whole queue
      // (s_waitcnt vmcnt(0)) after every tap, i.e. wait for this prefetch at tap 0 instead of hiding it behind nine taps of MFMAs
      prefetch((more ? cc + 1 : cc) * LY_CC);
      {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          int toff = (tap / 3) * rp + (tap % 3) * LY_RSH;
          asm volatile("" : "+s"(toff));                   // opaque per chunk: the 9 x NC fragment addresses are not hoisted out of the chunk loop (72 registers)
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const int j = tap * KS + ks;
            int lane_j = lane;
            asm volatile("" : "+v"(lane_j));               // the fragment addresses of a step are formed in the step (formed early they spill)
#pragma unroll
            for (int t = 0; t < MT; ++t) wq[(j + PD - 1) % PD][t] = ly_wfragp<PL>(wpk, wbase[t] + wstep_at(cc, j + PD - 1, more), lane_j);
#pragma unroll
            for (int n = 0; n < NC; ++n) {
              {
                const bf16x8 xh = ly_img_frag(hs_hi, hb[n] + toff, ps, ks, lq);
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[t][n] = ly_mfmap<PL>(wq[j % PD][t], xh, xh, acc[t][n]);
              }
            }
            __builtin_amdgcn_sched_barrier(0);             // a step's reads stay in the step
          }
        }
      }
      if (more) {
        __syncthreads();            // every wave is done reading the frame of chunk cc
        commit((cc + 1) * LY_CC);
        __syncthreads();
      }
    }
  };
  {
    int nw = ntv - wp * NTW;
    nw = nw < 0 ? 0 : nw > NTW ? NTW : nw;
    switch (nw) {
      case 1: contract(LyC3Ic<1>()); break;
      case 2: contract(LyC3Ic<2>()); break;
      case 3: contract(LyC3Ic<3>()); break;
      case 4: contract(LyC3Ic<(NTW >= 4 ? 4 : NTW)>()); break;
      case 5: contract(LyC3Ic<(NTW >= 5 ? 5 : NTW)>()); break;
      case 6: contract(LyC3Ic<(NTW >= 6 ? 6 : NTW)>()); break;
      case 7: contract(LyC3Ic<(NTW >= 7 ? 7 : NTW)>()); break;
      case 8: contract(LyC3Ic<(NTW >= 8 ? 8 : NTW)>()); break;
      default: contract(LyC3Ic<0>()); break;                  // (a wave without pixels still stages frames and meets the barriers)
    }
  }

  // ---- epilogue ---------------------------------------------------------------------------------
  const int act = P.act;
  const bool vec_ok = (P.ldo & 3) == 0;
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int tt = (by * WC + wc) * MT + t;
    const int c = 16 * tt + 4 * lq;
    if (tt >= Tt || c >= P.N) continue;
    float sc[4], sh[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = c + r < P.N;
      sc[r] = (ok && P.e_scale) ? P.e_scale[c + r] : 1.f;
      sh[r] = (ok && P.e_shift) ? P.e_shift[c + r] : 0.f;
    }
    f32x4 s1 = zero, s2 = zero;
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
      const long orow_n = out_row(n);
      if (orow_n < 0) continue;
      f32x4 u;
#pragma unroll
      for (int r = 0; r < 4; ++r) u[r] = acc[t][n][r] * sc[r] + sh[r];
      if (P.stats) {
        s1 += u;
        s2 += u * u;
        if (!out) continue;                                // pure statistics pass; with `out` the value is stored as well
      }
      const f32x4 v = ly_act4(u, act);
      T* o = out + orow_n * P.ldo + c;
      if (vec_ok && c + 3 < P.N) {
        ly_st4<T>(o, v);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (c + r < P.N) ly_st1<T>(o + r, v[r]);
      }
    }
    if (P.stats) ly_stats_flush(P.stats, P.N, c, s1, s2);
  }
}

// Row pitch (bytes) of the LDS frame.  A B-operand read is one ds_read_b128 per lane (lane-group operand image, ly_tile.hpp), served in groups
// of 16 lanes that hold the 16 pixels of an MFMA tile: conflict-free iff the 16 pixel addresses cover 16 different 16-byte slots of the
// 256-byte bank row.
// Pixels of one frame row do (consecutive positions are an odd number of slots apart), but the patches the host picks are 8 or 4
// pixels wide (pick_conv_tile: 16 x 8 at 80 x 80, 10 x 8 at 40 x 40, 20 x 4 at 20 x 20), so a tile spans 2 or 4 frame rows, and with the
// dense pitch (TW + 2) * RSH two of them meet on the same slots.
// The pitch is padded (at most 240 bytes per frame row) to the first value for which every tile of the patch reads conflict-free.
static int conv3_row_pitch(int TH, int TW, int rsh) {
  const int FW = TW + 2, npx = TH * TW;
  int best = FW * rsh, best_w = 1 << 30;
  for (int rp = FW * rsh; rp < FW * rsh + 256; rp += 16) {
    int worst = 0;
    for (int n = 0; 16 * n < npx; ++n) {
      int cnt[16] = {0};
      for (int li = 0; li < 16; ++li) {
        const int p = 16 * n + li, pp = p < npx ? p : 0;
        const int c = ++cnt[(((pp / TW) * rp + (pp % TW) * rsh) >> 4) & 15];
        if (p < npx && c > worst) worst = c;
      }
    }
    if (worst < best_w) { best_w = worst; best = rp; }
    if (worst <= 1) break;
  }
  return best;
}

template <typename T, int MT, int WC, bool LAT = false, int NTA = 0>
static int launch_conv3(const LyConv3Params& P, hipStream_t st) {
  const int tiles_x = (P.W + P.TW - 1) / P.TW, tiles_y = (P.H + P.TH - 1) / P.TH;
  const int gy = (P.N + 16 * MT * WC - 1) / (16 * MT * WC);
  const long n_img = P.M / ((long)P.H * P.W);
  long nb = n_img * tiles_x * tiles_y * gy;
  LY_CHECK(nb < (1L << 31), "conv3x3: grid too large");
  const int rp = conv3_row_pitch(P.TH, P.TW, ly_qrs(8 * LyT<T>::VW / 32));
  const int ps = ly_qps((P.TH + 2) * rp);
  size_t lds = LyT<T>::PL * (size_t)4 * ps;
  void (*k)(const LyConv3Params, const int, const int, const int, const int, const int) = ly_conv3x3_kernel<T, MT, WC, LAT ? 0 : NTA>;
  if constexpr (LAT) k = ly_conv3x3_lat_kernel<T, MT, WC>;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    LY_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    configured = true;
  }
  hipLaunchKernelGGL(k, dim3((unsigned)nb), dim3(LY_THREADS), lds, st, P, gy, tiles_x, tiles_y, rp, ps);
  LY_LAUNCH_CHECK();
  return 0;
}

template <typename T>
static int conv3_dispatch(const LyConv3Params& P, hipStream_t st) {
  // measured on MI355X: 32 channels per wave is the sweet spot at every LEAD-YOLO shape
#ifdef LY_DEVEL
  static int mode = -1;                                   // development builds (make DEVEL=1) read LY_C3_MODE: 1 = <2,4>, 2 = <2,4,LAT>, 3 = <2,2>, 4 = <2,2,LAT>
  if (mode < 0) { const char* e = getenv("LY_C3_MODE"); mode = e ? atoi(e) : 0; }
#else
  constexpr int mode = 0;
#endif
  if constexpr (LyT<T>::BF) {
    const long tiles = (P.M / ((long)P.H * P.W)) * ((P.W + P.TW - 1) / P.TW) * ((P.H + P.TH - 1) / P.TH);
    const bool small = tiles * ((P.N + 127) / 128) < 2 * 256;            // fewer than two blocks per CU
    if (mode == 2) return launch_conv3<T, 2, 4, true>(P, st);
    if (mode == 4 || (mode == 0 && P.N > 64 && small)) return launch_conv3<T, 2, 2, true>(P, st);   // 64 channels per block: twice the blocks
  }
  const int ntv = (P.TH * P.TW + 15) >> 4;                // active pixel tiles of a patch
  if (mode == 3) return launch_conv3<T, 2, 2>(P, st);
  if (mode == 1 || P.N > 64) {                            // 4 waves x 32 ch (128 ch per block), all 8 pixel tiles each
    if constexpr (LyT<T>::BF) {
      if (mode == 0 && ntv == 5) return launch_conv3<T, 2, 4, false, 5>(P, st);
      if (mode == 0 && ntv == 8) return launch_conv3<T, 2, 4, false, 8>(P, st);
    }
    return launch_conv3<T, 2, 4>(P, st);
  }
  if constexpr (LyT<T>::BF) {
    if (mode == 0 && ntv == 8) return launch_conv3<T, 2, 2, false, 4>(P, st);
  }
  return launch_conv3<T, 2, 2>(P, st);                    // 2 x 32 ch, 2 pixel groups of 4 tiles
}

extern "C" int ly_conv3x3_fwd(const LyConv3Params* p, void* stream) {
  LY_CHECK(p, "conv3x3: null params");
  const LyConv3Params& P = *p;
  LY_CHECK(P.dtype == LY_F32 || P.dtype == LY_BF16, "conv3x3: unknown dtype %d", P.dtype);
  const int vw = P.dtype == LY_BF16 ? 8 : 4;
  LY_CHECK(P.x && P.wp && (P.out || P.stats), "conv3x3: null pointer");
  LY_CHECK(P.M > 0 && P.H > 0 && P.W > 0 && P.Cin > 0 && P.N > 0, "conv3x3: bad sizes");
  LY_CHECK((P.Cin & 3) == 0 && P.ldx % vw == 0 && P.ldx >= (P.Cin + vw - 1) / vw * vw && ((uintptr_t)P.x & 15) == 0,
           "conv3x3: Cin=%d must be a multiple of 4, ldx=%d a multiple of %d covering Cin rounded up to it", P.Cin, P.ldx, vw);
  LY_CHECK(P.M % ((long)P.H * P.W) == 0, "conv3x3: M is not a whole number of images");
  LY_CHECK(P.TH >= 1 && P.TW >= 1 && P.TH * P.TW <= 16 * LY_C3_NT, "conv3x3: patch %dx%d exceeds %d pixels", P.TH, P.TW, 16 * LY_C3_NT);
  LY_CHECK((P.TH + 2) * (P.TW + 2) * 8 <= LY_C3_NV * LY_THREADS, "conv3x3: frame of patch %dx%d exceeds the staging capacity", P.TH, P.TW);
  LY_CHECK(P.M * (long)P.ldx < (1L << 31), "conv3x3: input exceeds the 31-bit offsets of the staging plan");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  return P.dtype == LY_BF16 ? conv3_dispatch<__bf16>(P, st) : conv3_dispatch<float>(P, st);
}
