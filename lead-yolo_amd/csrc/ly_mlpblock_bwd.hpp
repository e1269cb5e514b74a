#pragma once
// Fused FasterNet MLPBlock BACKWARD (training), gfx950, bf16 storage, fp32 accumulation.
//
//   forward (models/common.py:1432-1437, 1478-1482):  z = [pconv3x3(x[:, :C/4]) | x[:, C/4:]],  u = W1 z,  v = a u + b (BatchNorm with the
//   BATCH statistics),  h = relu(v),  y = x + W2 h.      What autograd derives for it (train.py:327 `scaler.scale(loss).backward()`):
//        dh = W2^T dy,   dv = dh [v > 0],   s1 = sum dv,  s2 = sum dv u   (-> dgamma, dbeta and the coefficients of du)
//        du = alpha dv + kappa + lambda u,   g = W1^T du,   dW1 = sum du (x) z,   dW2 = sum dy (x) h,   dx = dy + [pconv^T(g[:C/4]) | g[C/4:]]
//
// The unfused path (round 2-4) ran this as eleven launches over 2C-wide tensors in HBM — z, u, relu(bn(u)), dh, the reduce pass, the apply
// pass, the g GEMM, two weight-gradient launches + their combines — 780 us for the 160 x 160 x 24 stage at bs = 64 where the fused forward
// takes 85.  Here the hidden tensors never leave the chip, exactly as in ly_mlpblock_fwd: TWO passes over (x, dy), split by the one global
// dependency (the BatchNorm sums):
//   PASS 1  per pixel tile: z (partial conv from the LDS halo image), u = W1 z and dh = W2^T dy as two MFMA chains with the SAME accumulator
//           layout (lane = pixel, 4 hidden channels), dv and the two sums in registers; one double atomic per channel and block at the end.
//   PASS 2  the same up to dv; du and h in registers; g += W1^T du straight from the accumulators (two fp32 tiles ARE the B operand of the
//           next k-step, ly_tile.hpp); du and h also go — as bf16, 8 bytes per lane — into two LDS tiles [pixel][hidden], from which
//           dW1 += du^T z and dW2^T += h^T dy are contracted through transposed reads (ds_read_b64_tr_b16: the contraction index is the
//           pixel).  Weight-gradient accumulators stay in registers over the block's whole tile walk; the block writes ONE slab,
//           ly_mlpblock_bwd_combine folds the slabs in block order (bit-reproducible).
//           DWX = false (C <= 40): a wave contracts ALL hidden tiles over its OWN 32 pixels — du / h rows are wave-private, no barrier,
//             2 * HTR * C16 accumulator tiles per wave; the four waves meet in LDS when the block ends.
//           DWX = true (C = 80): 400 accumulator registers per wave would not fit — the hidden tiles go through LDS in rounds of four, after
//             a barrier wave w contracts tile 4 r + w over ALL the block's pixels and owns its slab entries outright.
// HBM traffic: pass 1 reads x, dy; pass 2 reads x, dy and writes g — against ~17 map-sized reads / writes before.
//
// Tiling: T2D = 4*NT x 16 pixel patches of ANY map width (columns past W are masked: W = 40 wastes 17 % of the third patch column), halo frame
// with zeros staged for out-of-image taps, the next patch's raw pixels in flight (registers) during the arithmetic; flattened runs of 64*NT
// pixels where patches would waste more than a quarter of their columns.  All weight fragments of the four contractions are resident in
// LDS (15 / 37 / 97 KB at C = 24 / 40 / 80), copied once per block.
#include "ly_mlpblock.hpp"

typedef short mb_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x4 mb_tr(const char* p) {
  typedef __attribute__((address_space(3))) mb_s16x4 lds_s16x4;
  return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p));
}

struct LyMlpBwdArgs {
  const __bf16* x;
  const __bf16* dy;
  __bf16* g;                     // pass 2: d(loss)/dz, dense [M, C]
  long M;
  int H, W, n_img, ntiles;
  const uint4 *wp, *w1, *w2t, *w1t;
  const float *a, *b;            // hidden BatchNorm as v = a u + b (batch statistics), 2C entries
  const float *alpha, *kappa, *lambda;   // pass 2: du = alpha dv + kappa + lambda u
  double* stats;                 // pass 1: striped [LY_STATS_STRIPES][2 * 2C]: sum dv | sum dv u
  float* slab;                   // pass 2: [gridDim.x][MlpBwdGeom::SLAB] raw accumulator tiles
};

template <int C, int PASS = 2>
struct MlpBwdGeom {
  using Gm = MlpGeom<C>;
  static constexpr int HTR = 2 * C / 16;                   // real hidden tiles (2C is a multiple of 16)
  static constexpr bool DWX = C >= 80;
  static constexpr int NT = DWX ? 1 : 2;                   // pixel tiles of 16 per wave.  (C = 80 pass 1 at two tiles — it has the LDS room — was measured
                                                           // twice: 47 spilled registers, 45 -> 50 us; operands k-step by k-step, 21 spilled: 44 -> 54 us)
  static constexpr int BP = 64 * NT;
  static constexpr int DT = DWX ? 4 : Gm::HTP;             // hidden tiles per du / h round (DWX: one per wave)
  static constexpr int NRD = (Gm::HTP + DT - 1) / DT;      // rounds
  static constexpr int RSD = 2 * 16 * DT + 16;             // du / h tile row stride (bytes)
  static constexpr int NACC = HTR * Gm::C16;               // accumulator tiles per weight gradient
  static constexpr int SLAB = 2 * NACC * 256;              // floats per block: [dW1 | dW2^T][t][ct][lane][4]
  static constexpr int NFP = Gm::PT * Gm::SP, NF1 = Gm::HTP * Gm::S1, NF1T = Gm::C16 * Gm::S2;
  static constexpr int NFW = NFP + 2 * NF1 + NF1T;         // fragments of Wp, W1, W2^T, W1^T
};

// PD = patches in flight ahead of the one being computed (T2D, one or two register sets).  MEASURED: two sets change nothing (C = 24 pass 1
// 80.7 -> 80.7 us, C = 40 64 -> 68 us at bs = 64) — a patch's arithmetic already covers the HBM round trip; the passes are paced by instruction
// issue and the LDS pipe (PMC: a third of the wave cycles in LDS-issue stalls), not by loads in flight.  PD = 1 everywhere.
// (C = 40 pass 1 pinned to two waves per SIMD — 256 registers, 17 spilled — was measured: module 273 -> 286 us)
template <int C, int HT, bool T2D, int PASS, int PD>
__device__ __forceinline__ void ly_mlp_bwd_body(const LyMlpBwdArgs& P) {
  using Gm = MlpGeom<C>;
  using Bg = MlpBwdGeom<C, PASS>;
  using T = __bf16;
  typedef ly_u32x4 RV;
  typedef ly_u32x2 R4;
  constexpr int VW = 8;
  constexpr int CQ = Gm::CQ, G = Gm::G, SP = Gm::SP, PT = Gm::PT, C16 = Gm::C16, KP = Gm::KP, S1 = Gm::S1;
  constexpr int RS = Gm::RS, RSP = Gm::RSP, HTP = Gm::HTP, S2 = Gm::S2;
  constexpr int HTR = Bg::HTR, NT = Bg::NT, BP = Bg::BP, RSD = Bg::RSD, DT = Bg::DT, NRD = Bg::NRD, TH = 4 * NT;
  constexpr bool DWX = Bg::DWX;
  constexpr int NFP = Bg::NFP, NF1 = Bg::NF1, NF1T = Bg::NF1T;
  static_assert(HTP % HT == 0 && HT % 2 == 0 && DT % HT == 0 && C % VW == 0 && (DWX || NT % 2 == 0), "geometry");
  const int H = P.H, W = P.W;
  const long M = P.M;
  const int BPH = T2D ? (TH + 2) * 18 : BP + 2 * W + 2;

  extern __shared__ f32x4 ly_smem4[];
  char* const wl = reinterpret_cast<char*>(ly_smem4);
  float* const cfl = reinterpret_cast<float*>(wl + (NFP + 2 * NF1 + (PASS == 2 ? NF1T : 0)) * 1024);      // a | b | alpha | kappa | lambda, 2C floats each
  constexpr int NCF = PASS == 2 ? 5 : 2;
  char* const xs = reinterpret_cast<char*>(cfl) + NCF * 2 * C * 4;
  char* const dys = xs + BP * RS;
  char* const ps = dys + BP * RS;
  char* const dus = ps + (BPH * RSP + 15) / 16 * 16;       // pass 2 only
  char* const hs = dus + BP * RSD;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  const f32x4 zero = ly_zero4();
  const T* const x = P.x;
  const T* const dy = P.dy;

  // ---- weights -> LDS, once ---------------------------------------------------------------------------------------------------------
  for (int i = tid; i < NFP * 64; i += LY_THREADS) reinterpret_cast<uint4*>(wl)[i] = P.wp[i];
  for (int i = tid; i < NF1 * 64; i += LY_THREADS) reinterpret_cast<uint4*>(wl)[NFP * 64 + i] = P.w1[i];
  for (int i = tid; i < NF1 * 64; i += LY_THREADS) reinterpret_cast<uint4*>(wl)[(NFP + NF1) * 64 + i] = P.w2t[i];
  if constexpr (PASS == 2)
    for (int i = tid; i < NF1T * 64; i += LY_THREADS) reinterpret_cast<uint4*>(wl)[(NFP + 2 * NF1) * 64 + i] = P.w1t[i];
  for (int i = tid; i < 2 * C; i += LY_THREADS) {
    cfl[i] = P.a[i];
    cfl[2 * C + i] = P.b[i];
    if constexpr (PASS == 2) {
      cfl[4 * C + i] = P.alpha[i];
      cfl[6 * C + i] = P.kappa[i];
      cfl[8 * C + i] = P.lambda[i];
    }
  }
  auto wlds = [&](int fi) -> bf16x8 { return *reinterpret_cast<const bf16x8*>(wl + (fi * 64 + lane) * 16); };
  auto coef = [&](int k, int ch) -> f32x4 { return *reinterpret_cast<const f32x4*>(cfl + k * 2 * C + ch); };

  // ---- tile geometry ------------------------------------------------------------------------------------------------------------
  const int tw = T2D ? ((W + 15) >> 4) : 1, th = T2D ? (H + TH - 1) / TH : 1;
  long img0 = 0, p0 = 0;
  int h0 = 0, w0 = 0;
  auto decode = [&](int tile, long& i0, int& hh0, int& ww0, long& q0) {
    if constexpr (T2D) {
      int b = tile;
      const int tx = b % tw; b /= tw;
      const int ty = b % th;
      i0 = (long)(b / th) * H * W;
      hh0 = ty * TH; ww0 = tx * 16;
      q0 = 0;
    } else {
      i0 = 0; hh0 = 0; ww0 = 0;
      q0 = (long)tile * BP;
    }
  };

  // ---- staging: T2D = register prefetch of the next patch (x tile, x halo, dy tile), committed after the arithmetic ----------------
  constexpr int TVN = BP * (KP / VW), NVT = T2D ? (TVN + LY_THREADS - 1) / LY_THREADS : 1;
  constexpr int HVN = T2D ? (TH + 2) * 18 * G : 1, NVH = T2D ? (HVN + LY_THREADS - 1) / LY_THREADS : 1;
  struct PSet {
    RV tv[NVT], dv_[NVT];
    R4 hv[NVH];
    bool tok[NVT], hok[NVH];
  };
  PSet Q0, Q1;
  auto issue = [&](PSet& S, int tile) {
    RV (&tv)[NVT] = S.tv; RV (&dv_)[NVT] = S.dv_; R4 (&hv)[NVH] = S.hv; bool (&tok)[NVT] = S.tok; bool (&hok)[NVH] = S.hok;
    long i0, q0; int hh0, ww0;
    decode(tile, i0, hh0, ww0, q0);
#pragma unroll
    for (int e = 0; e < NVT; ++e) {
      const int idx = tid + e * LY_THREADS;
      const int pix = idx / (KP / VW), c4 = idx - pix * (KP / VW);
      const int r = pix >> 4, cx = pix & 15;
      tok[e] = idx < TVN && hh0 + r < H && ww0 + cx < W && c4 * VW < C;
      const long off = tok[e] ? (i0 + (long)(hh0 + r) * W + ww0 + cx) * C + c4 * VW : 0;
      tv[e] = ly_ldrv<T>(x + off);
      dv_[e] = ly_ldrv<T>(dy + off);
    }
#pragma unroll
    for (int e = 0; e < NVH; ++e) {
      const int idx = tid + e * LY_THREADS;
      const int hp = idx / G, c4 = idx - hp * G;
      const int hr = hp / 18, hc = hp - hr * 18;
      const int hh = hh0 - 1 + hr, ww = ww0 - 1 + hc;
      hok[e] = idx < HVN && hh >= 0 && hh < H && ww >= 0 && ww < W;
      hv[e] = ly_ldr4<T>(hok[e] ? x + (i0 + (long)hh * W + ww) * C + c4 * 4 : x);
    }
  };
  auto commit = [&](const PSet& S) {
    const RV (&tv)[NVT] = S.tv; const RV (&dv_)[NVT] = S.dv_; const R4 (&hv)[NVH] = S.hv; const bool (&tok)[NVT] = S.tok; const bool (&hok)[NVH] = S.hok;
#pragma unroll
    for (int e = 0; e < NVT; ++e) {
      const int idx = tid + e * LY_THREADS;
      const int pix = idx / (KP / VW), c4 = idx - pix * (KP / VW);
      RV v = tv[e], d = dv_[e];
      if (!tok[e]) { ly_zero_raw(v); ly_zero_raw(d); }
      if (idx < TVN) {
        *reinterpret_cast<RV*>(xs + pix * RS + 2 * VW * c4) = v;
        *reinterpret_cast<RV*>(dys + pix * RS + 2 * VW * c4) = d;
      }
    }
#pragma unroll
    for (int e = 0; e < NVH; ++e) {
      const int idx = tid + e * LY_THREADS;
      const int hp = idx / G, c4 = idx - hp * G;
      R4 v = hv[e];
      if (!hok[e]) ly_zero_raw(v);
      if (idx < HVN) *reinterpret_cast<R4*>(ps + hp * RSP + 8 * c4) = v;
    }
  };
  // flattened runs: direct staging (the halo length depends on W)
  auto stage_flat = [&](int tile) {
    const long q0 = (long)tile * BP;
    ly_stage_raw<4, RV>(BP * (KP / VW), tid, x,
        [&](int idx) -> const void* {
          const int pix = idx / (KP / VW), c4 = idx - pix * (KP / VW);
          const long gp = q0 + pix;
          return (gp < M && c4 * VW < C) ? x + gp * C + c4 * VW : nullptr;
        },
        [&](int idx, RV v) {
          const int pix = idx / (KP / VW), c4 = idx - pix * (KP / VW);
          *reinterpret_cast<RV*>(xs + pix * RS + 2 * VW * c4) = v;
        });
    ly_stage_raw<4, RV>(BP * (KP / VW), tid, dy,
        [&](int idx) -> const void* {
          const int pix = idx / (KP / VW), c4 = idx - pix * (KP / VW);
          const long gp = q0 + pix;
          return (gp < M && c4 * VW < C) ? dy + gp * C + c4 * VW : nullptr;
        },
        [&](int idx, RV v) {
          const int pix = idx / (KP / VW), c4 = idx - pix * (KP / VW);
          *reinterpret_cast<RV*>(dys + pix * RS + 2 * VW * c4) = v;
        });
    ly_stage_raw<4, R4>(BPH * G, tid, x,
        [&](int idx) -> const void* {
          const int hp = idx / G, c4 = idx - hp * G;
          const long gp = q0 - W - 1 + hp;
          return (gp >= 0 && gp < M) ? x + gp * C + c4 * 4 : nullptr;
        },
        [&](int idx, R4 v) {
          const int hp = idx / G, c4 = idx - hp * G;
          *reinterpret_cast<R4*>(ps + hp * RSP + 8 * c4) = v;
        });
  };

  // ---- block-lifetime accumulators ---------------------------------------------------------------------------------------------------
  constexpr int NOWN = DWX ? NRD : HTR;                    // hidden tiles whose weight gradients this wave accumulates
  constexpr int NAW = NOWN * C16;
  f32x4 st1[PASS == 1 ? HTR : 1], st2[PASS == 1 ? HTR : 1];
  f32x4 aw1[PASS == 2 ? NAW : 1], aw2[PASS == 2 ? NAW : 1];
  if constexpr (PASS == 1) {
#pragma unroll
    for (int t = 0; t < HTR; ++t) { st1[t] = zero; st2[t] = zero; }
  } else {
#pragma unroll
    for (int i = 0; i < NAW; ++i) { aw1[i] = zero; aw2[i] = zero; }
  }

  const int g_ = (int)gridDim.x;
  const int bid = ly_xcd_remap((int)blockIdx.x, (int)gridDim.x);            // blocks of one XCD walk neighbouring patches (shared halo rows in its L2)
  const int tlast = P.ntiles > bid ? bid + (P.ntiles - 1 - bid) / g_ * g_ : 0;      // the block's last patch
  auto clampt = [&](int t) -> int { return t < P.ntiles ? t : tlast; };     // (past the end: the last patch is re-requested — straight-line loads)
  if constexpr (T2D) {
    if (bid < P.ntiles) {
      issue(Q0, bid);
      commit(Q0);
      if constexpr (PD == 2) issue(Q1, clampt(bid + g_));
    }
  }
  __syncthreads();                                           // weights (and the first patch) are in LDS
  const int pixbase = wave * (16 * NT);
  const bf16x4 z4 = __builtin_bit_cast(bf16x4, make_uint2(0u, 0u));
  const int r0 = 4 * lq + (li >> 2), c8 = 8 * (li & 3);     // transposed-read addressing (the k-set of ly_tile.hpp)

  // one patch: `tile` is in LDS; Si receives the patch PD ahead, Sc (the patch after `tile`) is committed when the arithmetic is done
  auto step = [&](const int tile, PSet& Si, PSet& Sc) {
    if constexpr (T2D) {
      issue(Si, clampt(tile + PD * g_));
    } else {
      stage_flat(tile);
      __syncthreads();
    }
    decode(tile, img0, h0, w0, p0);
    auto gpix = [&](int pix) -> long {
      if constexpr (T2D) {
        const int r = pix >> 4, cx = pix & 15;
        return (h0 + r < H && w0 + cx < W) ? img0 + (long)(h0 + r) * W + w0 + cx : -1;
      } else {
        const long gp = p0 + pix;
        return gp < M ? gp : -1;
      }
    };

    // ---- 1. z = partial 3x3 conv into the tile (the forward's code) ---------------------------------------------------------------
    {
      uint32_t tmask[NT];
      int pbase[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int pix = pixbase + 16 * n + li;
        if constexpr (T2D) {
          tmask[n] = 0x1ffu;
          pbase[n] = ((pix >> 4) * 18 + (pix & 15)) * RSP;
        } else {
          const long gp = p0 + pix;
          int h_, w_;
          ly_pix_hw(gp < M ? gp : 0, H, W, h_, w_);
          tmask[n] = ly_tapmask(h_, w_, H, W, gp < M);
          pbase[n] = pix * RSP;
        }
      }
      const int rowpitch = T2D ? 18 : W;
      f32x4 accp[PT][NT];
#pragma unroll
      for (int t = 0; t < PT; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) accp[t][n] = zero;
      // every operand of the SP k-steps is requested before the first MFMA (left alone hipcc reads each fragment right before its use: one
      // exposed LDS round trip per MFMA at one wave per SIMD — 69 lgkmcnt(0) waits per tile in the C = 80 listing)
      // (SB k-steps per batch: all SP at once unless the operand set would exceed 72 registers — no built configuration does)
      constexpr int SB = SP * (NT + PT) * 4 <= 72 ? SP : 3;
#pragma unroll
      for (int s0 = 0; s0 < SP; s0 += SB) {
        bf16x8 xh[SB][NT], wpf[SB][PT];
#pragma unroll
        for (int sb = 0; sb < SB; ++sb) {
          const int s = s0 + sb;
          if (s >= SP) continue;
          int off[2], tap[2];
          bool gv[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int gq = 8 * s + 4 * h + lq;
            gv[h] = gq < 9 * G;
            tap[h] = gv[h] ? gq / G : 0;
            const int cq4 = gv[h] ? gq - tap[h] * G : 0;
            const int ty = tap[h] / 3, tx = tap[h] - 3 * ty;
            off[h] = (ty * rowpitch + tx) * RSP + 8 * cq4;
          }
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            bf16x4 ph[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const bool ok = gv[h] && (T2D || ((tmask[n] >> tap[h]) & 1u));
              const bf16x4 a = *reinterpret_cast<const bf16x4*>(ps + pbase[n] + off[h]);
              ph[h] = ok ? a : z4;
            }
            xh[sb][n] = ly_cat8(ph[0], ph[1]);
          }
#pragma unroll
          for (int t = 0; t < PT; ++t) wpf[sb][t] = wlds(t * SP + s);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sb = 0; sb < SB; ++sb) {
          if (s0 + sb >= SP) continue;
#pragma unroll
          for (int t = 0; t < PT; ++t)
#pragma unroll
            for (int n = 0; n < NT; ++n) accp[t][n] = ly_mfma_bf16(wpf[sb][t], xh[sb][n], accp[t][n]);
        }
        if (s0 + SB < SP) __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int t = 0; t < PT; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const bf16x4 h = ly_cvtb4(accp[t][n]);
          const int c = 16 * t + 4 * lq;
          const int rb = (pixbase + 16 * n + li) * RS + 2 * c;
          if (c < CQ) *reinterpret_cast<bf16x2*>(xs + rb) = __builtin_shufflevector(h, h, 0, 1);
          if (c + 2 < CQ) *reinterpret_cast<bf16x2*>(xs + rb + 4) = __builtin_shufflevector(h, h, 2, 3);
        }
    }

    // ---- 2. hidden chunks: u = W1 z, dh = W2^T dy, dv / du / h in registers, g += W1^T du; du / h -> LDS, dW per round -------------
    bool okp[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) okp[n] = gpix(pixbase + 16 * n + li) >= 0;
    f32x4 acco[PASS == 2 ? C16 : 1][NT];
    if constexpr (PASS == 2) {
#pragma unroll
      for (int t = 0; t < C16; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) acco[t][n] = zero;
    }
#pragma unroll
    for (int rd = 0; rd < NRD; ++rd) {
#pragma unroll
      for (int hq = 0; hq < DT / HT; ++hq) {
        const int hc = rd * (DT / HT) + hq;                 // hidden chunk of HT tiles
        if (hc * HT >= HTR) continue;
        f32x4 au[HT][NT], ad[HT][NT];
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
          for (int n = 0; n < NT; ++n) { au[t][n] = zero; ad[t][n] = zero; }
        // all operands of the chunk's two contractions (and, pass 2, of its share of g = W1^T du) in flight together, one wait
        // (KB k-steps per batch: all S1 unless the operand set would exceed 72 registers — no built configuration does; C = 80 at two pixel
        // tiles per wave would)
        constexpr int KB = S1 * (NT + HT) * 8 <= 72 ? S1 : 1;
        bf16x8 wtf[PASS == 2 ? HT / 2 : 1][PASS == 2 ? C16 : 1];
#pragma unroll
        for (int s0 = 0; s0 < S1; s0 += KB) {
          bf16x8 xb[KB][NT], db[KB][NT], w1f[KB][HT], w2f[KB][HT];
#pragma unroll
          for (int sb = 0; sb < KB; ++sb) {
            const int s = s0 + sb;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
              xb[sb][n] = ly_lds_frag(xs, (pixbase + 16 * n + li) * RS, s, lq);
              db[sb][n] = ly_lds_frag(dys, (pixbase + 16 * n + li) * RS, s, lq);
            }
#pragma unroll
            for (int t = 0; t < HT; ++t) {
              if (hc * HT + t >= HTR) continue;              // padded hidden tile: nothing to contract
              w1f[sb][t] = wlds(NFP + (hc * HT + t) * S1 + s);
              w2f[sb][t] = wlds(NFP + NF1 + (hc * HT + t) * S1 + s);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int sb = 0; sb < KB; ++sb)
#pragma unroll
            for (int t = 0; t < HT; ++t) {
              if (hc * HT + t >= HTR) continue;
#pragma unroll
              for (int n = 0; n < NT; ++n) {
                au[t][n] = ly_mfma_bf16(w1f[sb][t], xb[sb][n], au[t][n]);
                ad[t][n] = ly_mfma_bf16(w2f[sb][t], db[sb][n], ad[t][n]);
              }
            }
          if (s0 + KB < S1) __builtin_amdgcn_sched_barrier(0);
        }
        // W1^T fragments of g += W1^T du and the BatchNorm coefficients: read behind the MFMAs (not live beside the operand sets)
        if constexpr (PASS == 2) {
#pragma unroll
          for (int u2 = 0; u2 < HT / 2; ++u2) {
            if (hc * HT + 2 * u2 >= HTR) continue;
#pragma unroll
            for (int ct = 0; ct < C16; ++ct) wtf[u2][ct] = wlds(NFP + 2 * NF1 + ct * S2 + hc * (HT / 2) + u2);
          }
        }
        f32x4 cfa[HT], cfb[HT], cal[PASS == 2 ? HT : 1], cka[PASS == 2 ? HT : 1], cla[PASS == 2 ? HT : 1];
#pragma unroll
        for (int t = 0; t < HT; ++t) {
          const int tg = hc * HT + t;
          if (tg >= HTR) continue;
          const int ch = tg * 16 + 4 * lq;
          cfa[t] = coef(0, ch); cfb[t] = coef(1, ch);
          if constexpr (PASS == 2) { cal[t] = coef(2, ch); cka[t] = coef(3, ch); cla[t] = coef(4, ch); }
        }
        bf16x4 dub[HT][NT], hb[HT][NT];
#pragma unroll
        for (int t = 0; t < HT; ++t) {
          const int tg = hc * HT + t;
          if (tg >= HTR) {
#pragma unroll
            for (int n = 0; n < NT; ++n) { dub[t][n] = z4; hb[t][n] = z4; }
            continue;
          }
          const f32x4 a4 = cfa[t], b4 = cfb[t];
          f32x4 al4 = zero, ka4 = zero, la4 = zero;
          if constexpr (PASS == 2) { al4 = cal[t]; ka4 = cka[t]; la4 = cla[t]; }
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            f32x4 du4, h4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float u = au[t][n][r];
              const float v = u * a4[r] + b4[r];
              const bool pos = v > 0.f;
              const float dvv = pos ? ad[t][n][r] : 0.f;           // (dy is staged as zeros for pixels outside the map: dv = 0 there)
              if constexpr (PASS == 1) {
                st1[tg][r] += dvv;
                st2[tg][r] += dvv * u;
              } else {
                du4[r] = okp[n] ? al4[r] * dvv + ka4[r] + la4[r] * u : 0.f;
                h4[r] = pos ? v : 0.f;
              }
            }
            if constexpr (PASS == 2) {
              dub[t][n] = ly_cvtb4(du4);
              hb[t][n] = ly_cvtb4(h4);
            }
          }
        }
        if constexpr (PASS == 2) {
#pragma unroll
          for (int u2 = 0; u2 < HT / 2; ++u2) {
            if (hc * HT + 2 * u2 >= HTR) continue;
            bf16x8 kb[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) kb[n] = ly_cat8(dub[2 * u2][n], dub[2 * u2 + 1][n]);
#pragma unroll
            for (int ct = 0; ct < C16; ++ct)
#pragma unroll
              for (int n = 0; n < NT; ++n) acco[ct][n] = ly_mfma_bf16(wtf[u2][ct], kb[n], acco[ct][n]);
          }
          // du, h of the wave's own pixels -> its rows of the [pixel][hidden] tiles of this round (8 bytes per lane and tile)
#pragma unroll
          for (int t = 0; t < HT; ++t) {
            const int tg = hc * HT + t;
            if (tg >= HTR) continue;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
              const int ob = (pixbase + 16 * n + li) * RSD + 32 * (tg - rd * DT) + 8 * lq;
              *reinterpret_cast<bf16x4*>(dus + ob) = dub[t][n];
              *reinterpret_cast<bf16x4*>(hs + ob) = hb[t][n];
            }
          }
        }
      }
      if constexpr (PASS == 2) {
        // ---- 3. dW1 += du^T z, dW2^T += h^T dy (contraction index = pixel: transposed reads) ---------------------------------------
        if constexpr (!DWX) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave's own LDS stores above are visible to its transposed reads below
#pragma unroll
          for (int ks = 0; ks < NT / 2; ++ks) {
            const int row = pixbase + 32 * ks + r0;
            bf16x8 bz[C16], bd[C16];
#pragma unroll
            for (int ct = 0; ct < C16; ++ct) {
              bz[ct] = ly_cat8(mb_tr(xs + row * RS + 32 * ct + c8), mb_tr(xs + (row + 16) * RS + 32 * ct + c8));
              bd[ct] = ly_cat8(mb_tr(dys + row * RS + 32 * ct + c8), mb_tr(dys + (row + 16) * RS + 32 * ct + c8));
            }
            bf16x8 a1[HTR], a2[HTR];
#pragma unroll
            for (int t = 0; t < HTR; ++t) {
              a1[t] = ly_cat8(mb_tr(dus + row * RSD + 32 * t + c8), mb_tr(dus + (row + 16) * RSD + 32 * t + c8));
              a2[t] = ly_cat8(mb_tr(hs + row * RSD + 32 * t + c8), mb_tr(hs + (row + 16) * RSD + 32 * t + c8));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < HTR; ++t)
#pragma unroll
              for (int ct = 0; ct < C16; ++ct) {
                aw1[t * C16 + ct] = ly_mfma_bf16(a1[t], bz[ct], aw1[t * C16 + ct]);
                aw2[t * C16 + ct] = ly_mfma_bf16(a2[t], bd[ct], aw2[t * C16 + ct]);
              }
          }
        } else {
          __syncthreads();                                         // every wave's du / h rows of this round (and its z rows) are in LDS
          if (rd * DT + wave < HTR) {                              // wave-uniform: the transposed reads below run with all lanes active
#pragma unroll
            for (int ks = 0; ks < BP / 32; ++ks) {
              const int row = 32 * ks + r0;
              const bf16x8 a1 = ly_cat8(mb_tr(dus + row * RSD + 32 * wave + c8), mb_tr(dus + (row + 16) * RSD + 32 * wave + c8));
              const bf16x8 a2 = ly_cat8(mb_tr(hs + row * RSD + 32 * wave + c8), mb_tr(hs + (row + 16) * RSD + 32 * wave + c8));
              bf16x8 bz[C16], bd[C16];
#pragma unroll
              for (int ct = 0; ct < C16; ++ct) {
                bz[ct] = ly_cat8(mb_tr(xs + row * RS + 32 * ct + c8), mb_tr(xs + (row + 16) * RS + 32 * ct + c8));
                bd[ct] = ly_cat8(mb_tr(dys + row * RS + 32 * ct + c8), mb_tr(dys + (row + 16) * RS + 32 * ct + c8));
              }
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int ct = 0; ct < C16; ++ct) {
                aw1[rd * C16 + ct] = ly_mfma_bf16(a1, bz[ct], aw1[rd * C16 + ct]);
                aw2[rd * C16 + ct] = ly_mfma_bf16(a2, bd[ct], aw2[rd * C16 + ct]);
              }
            }
          }
          if (rd + 1 < NRD) __syncthreads();                       // the next round overwrites the du / h tiles
        }
      }
    }

    if constexpr (PASS == 2) {
#pragma unroll
      for (int ct = 0; ct < C16; ++ct)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int c = 16 * ct + 4 * lq;
          const long gp = gpix(pixbase + 16 * n + li);
          if (c < C && gp >= 0) ly_st4<T>(P.g + gp * C + c, acco[ct][n]);
        }
    }
    __syncthreads();                                               // every wave is done with the halo image / the tiles
    if constexpr (T2D) {
      commit(Sc);
      __syncthreads();
    }
  };
  if constexpr (PD == 2) {
    for (int tile = bid; tile < P.ntiles; tile += 2 * g_) {
      step(tile, Q0, Q1);
      if (tile + g_ < P.ntiles) step(tile + g_, Q1, Q0);
    }
  } else {
    for (int tile = bid; tile < P.ntiles; tile += g_) step(tile, Q0, Q0);
  }

  if constexpr (PASS == 1) {
#pragma unroll
    for (int t = 0; t < HTR; ++t) ly_stats_flush(P.stats, 2 * C, t * 16 + 4 * lq, st1[t], st2[t]);
  } else {
    f32x4* const out = reinterpret_cast<f32x4*>(P.slab + (size_t)blockIdx.x * Bg::SLAB);
    if constexpr (DWX) {
      // every (hidden tile, channel tile) entry has ONE owner wave: straight to the slab
#pragma unroll
      for (int rd = 0; rd < NRD; ++rd) {
        const int t = rd * DT + wave;
        if (t < HTR) {
#pragma unroll
          for (int ct = 0; ct < C16; ++ct) {
            out[(t * C16 + ct) * 64 + lane] = aw1[rd * C16 + ct];
            out[(Bg::NACC + t * C16 + ct) * 64 + lane] = aw2[rd * C16 + ct];
          }
        }
      }
    } else {
      // the four waves' accumulator tiles meet in LDS in wave order (fixed summation order), the block writes ONE slab
      float* const red = reinterpret_cast<float*>(xs);            // (tiles are dead: every wave passed the loop's last barrier)
      constexpr int RED_FLOATS = Bg::SLAB;
      static_assert(DWX || RED_FLOATS * 4 <= 2 * BP * RS + 2 * BP * RSD, "reduction scratch must fit the tile area");
      for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
          for (int i = 0; i < NAW; ++i) {
            f32x4* const p1 = reinterpret_cast<f32x4*>(red) + i * 64 + lane;
            f32x4* const p2 = reinterpret_cast<f32x4*>(red) + (Bg::NACC + i) * 64 + lane;
            if (w == 0) { *p1 = aw1[i]; *p2 = aw2[i]; }
            else { *p1 = *p1 + aw1[i]; *p2 = *p2 + aw2[i]; }
          }
        }
        __syncthreads();
      }
      for (int i = tid; i < RED_FLOATS / 4; i += LY_THREADS) out[i] = reinterpret_cast<const f32x4*>(red)[i];
    }
  }
}

#ifndef LY_MLPB_WAVES
#define LY_MLPB_WAVES(C, PASS) (((C) == 24 && (PASS) == 1) ? 3 : 1)
#endif
template <int C, int HT, bool T2D, int PASS, int PD>
__global__ __launch_bounds__(LY_THREADS) __attribute__((amdgpu_waves_per_eu(LY_MLPB_WAVES(C, PASS)))) void ly_mlpblock_bwd_kernel(const LyMlpBwdArgs P) {
  ly_mlp_bwd_body<C, HT, T2D, PASS, PD>(P);
}

// Fold of the blocks' slabs in a FIXED order (bit-reproducible): a block of 1024 threads owns EL consecutive float4 entries of the slab and
// 1024 / EL row lanes; row lane r adds the slabs r, r + RL, r + 2 RL ... in index order, the lanes meet in a binary tree over LDS.  EL is
// chosen by the launch so that the grid has ~100+ blocks (C = 24: 576 entries — at EL = 64 the fold ran on 9 CUs: 18.6 us).
template <int EL>
__device__ __forceinline__ bool ly_mlp_slab_fold(const float* __restrict__ slab, const int nblk, const int entries, int& e, f32x4& s) {
  constexpr int RL = 1024 / EL;
  __shared__ f32x4 red[1024];
  const int cl = threadIdx.x % EL, rl = threadIdx.x / EL;
  e = blockIdx.x * EL + cl;
  const bool live = e < entries;
  f32x4 acc0 = ly_zero4(), acc1 = ly_zero4();
  if (live) {
    const f32x4* p = reinterpret_cast<const f32x4*>(slab) + e;
    int b = rl;
    for (; b + RL < nblk; b += 2 * RL) {
      const f32x4 v0 = p[(size_t)b * entries], v1 = p[(size_t)(b + RL) * entries];
      acc0 += v0; acc1 += v1;
    }
    if (b < nblk) acc0 += p[(size_t)b * entries];
  }
  red[rl * EL + cl] = acc0 + acc1;
  __syncthreads();
#pragma unroll
  for (int h = RL / 2; h > 0; h >>= 1) {
    if (rl < h) red[rl * EL + cl] += red[(rl + h) * EL + cl];
    __syncthreads();
  }
  s = red[cl];
  return live && rl == 0;
}

// dw1[hid][c] += sum_b slab[b][0][t][ct][lane][r],  dw2[c][hid] += sum_b slab[b][1][...]   with hid = 16 t + 4 (lane >> 4) + r, c = 16 ct + (lane & 15)
template <int C, int EL>
__global__ __launch_bounds__(1024) void ly_mlpblock_bwd_combine_kernel(const float* __restrict__ slab, const int nblk, float* __restrict__ dw1,
                                                                      float* __restrict__ dw2) {
  using Bg = MlpBwdGeom<C>;
  constexpr int C16 = MlpGeom<C>::C16;
  int e;
  f32x4 s;
  if (!ly_mlp_slab_fold<EL>(slab, nblk, Bg::SLAB / 4, e, s)) return;
  const int which = e / (Bg::NACC * 64), rem = e - which * (Bg::NACC * 64);
  const int tile = rem >> 6, lane = rem & 63;
  const int t = tile / C16, ct = tile - t * C16;
  const int c = 16 * ct + (lane & 15);
  if (c < C) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int hid = 16 * t + 4 * (lane >> 4) + r;
      float* const d = which == 0 ? dw1 + (size_t)hid * C + c : dw2 + (size_t)c * (2 * C) + hid;
      *d += s[r];
    }
  }
}

template <int C, int HT, bool T2D, int PASS>
static int launch_mlp_bwd(LyMlpBwdArgs P, long slab_floats, int* blocks_out, hipStream_t st) {
  constexpr int PD = 1;
  using Gm = MlpGeom<C>;
  using Bg = MlpBwdGeom<C, PASS>;
  constexpr int BP = Bg::BP, NT = Bg::NT;
  const long halo = T2D ? (4 * NT + 2) * 18 : BP + 2 * P.W + 2;
  const size_t lds = (size_t)(Bg::NFP + 2 * Bg::NF1 + (PASS == 2 ? Bg::NF1T : 0)) * 1024 + (PASS == 2 ? 5 : 2) * 2 * C * 4 + 2 * (size_t)BP * Gm::RS +
                     ((size_t)halo * Gm::RSP + 15) / 16 * 16 + (PASS == 2 ? 2 * (size_t)BP * Bg::RSD : 0);
  LY_CHECK(lds <= 160 * 1024, "mlpblock_bwd: tile needs %zu B of LDS (C=%d W=%d)", lds, C, P.W);
  auto k = ly_mlpblock_bwd_kernel<C, HT, T2D, PASS, PD>;
  static LyDevOnce once;
  static int per_cu = 0;
  static size_t lds_q = 0;
  if (once.need() || lds_q != lds) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    LY_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    int nb = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(k), LY_THREADS, lds);
    LY_CHECK(e == hipSuccess, "hipOccupancyMaxActiveBlocksPerMultiprocessor: %s", hipGetErrorString(e));
    per_cu = nb < 1 ? 1 : (nb > 8 ? 8 : nb);
    lds_q = lds;
  }
  const long ntiles = T2D ? (long)P.n_img * ((P.H + 4 * NT - 1) / (4 * NT)) * ((P.W + 15) / 16) : (P.M + BP - 1) / BP;
  LY_CHECK(ntiles < (1L << 30), "mlpblock_bwd: too many tiles");
  long blocks = 256L * per_cu;
  if (blocks > ntiles) blocks = ntiles;
  if (PASS == 2) {
    LY_CHECK(P.slab && blocks * (long)Bg::SLAB <= slab_floats, "mlpblock_bwd: the slab workspace holds %ld floats, %ld needed", slab_floats,
             blocks * (long)Bg::SLAB);
  }
  P.ntiles = (int)ntiles;
  hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(LY_THREADS), lds, st, P);
  LY_LAUNCH_CHECK();
  if (blocks_out) *blocks_out = (int)blocks;
  return 0;
}

template <int C>
static int launch_mlp_bwd_combine(const float* slab, int nblk, float* dw1, float* dw2, hipStream_t st) {
  using Bg = MlpBwdGeom<C>;
  constexpr int EL = Bg::SLAB / 4 >= 6400 ? 64 : 16;       // entries per block: >= 48 blocks in every case
  hipLaunchKernelGGL((ly_mlpblock_bwd_combine_kernel<C, EL>), dim3((Bg::SLAB / 4 + EL - 1) / EL), dim3(1024), 0, st, slab, nblk, dw1, dw2);
  LY_LAUNCH_CHECK();
  return 0;
}

// patches (with the next patch prefetched) unless more than a quarter of the 16-wide patch columns would lie outside the map
template <int C, int HT, int PASS>
static int dispatch_mlp_bwd(LyMlpBwdArgs P, long slab_floats, int* blocks_out, hipStream_t st) {
  const int wp16 = (P.W + 15) / 16 * 16;
  if (P.W >= 12 && 4 * (wp16 - P.W) <= wp16) return launch_mlp_bwd<C, HT, true, PASS>(P, slab_floats, blocks_out, st);
  return launch_mlp_bwd<C, HT, false, PASS>(P, slab_floats, blocks_out, st);
}

template <int C, int HT>
static int mlp_bwd_pass(LyMlpBwdArgs P, int pass, long slab_floats, float* dw1, float* dw2, hipStream_t st) {
  if (pass == 1) return dispatch_mlp_bwd<C, HT, 1>(P, 0, nullptr, st);
  int blocks = 0;
  const int rc = dispatch_mlp_bwd<C, HT, 2>(P, slab_floats, &blocks, st);
  if (rc) return rc;
  return launch_mlp_bwd_combine<C>(P.slab, blocks, dw1, dw2, st);
}

int ly_mlp_bwd_pass_80(LyMlpBwdArgs P, int pass, long slab_floats, float* dw1, float* dw2, hipStream_t st);

// -------------------------------------------------------------------------------------------------------------------------------------------
// The tail of the MLPBlock backward in ONE launch (was: partial-conv weight gradient through the generic tiled kernel — 84 us at 160 x 160 x 24,
// 0.6 TB/s: a 72-wide K of which a quarter is real —, the partial conv's data gradient as a map-sized pass, and a third pass for the residual):
//      dx = dy + [ pconv^T(g[:, :C/4]) | g[:, C/4:] ]                 (Partial_conv3.forward_split_cat + the residual of MLPBlock.forward under autograd)
//      dWp[co][tap][ci] += sum_p g[p][co] x[p + tap][ci]               (partial_conv3.weight)
// Per patch: the g tile, and the halo frames of the first C/4 channels of BOTH g and x, go to LDS (the next patch's pixels and dy are in flight in
// registers meanwhile); the transposed-flipped 3x3 runs as in the forward and overwrites the tile's first C/4 channels; every thread then adds dy
// to its 16-byte pieces of the tile and stores dx.  The weight gradient contracts over the wave's own pixels with both operands read TRANSPOSED
// out of the halo frames (a tap is a constant offset of the x frame; out-of-image positions are staged zeros): 9 accumulator tiles per wave for
// C/4 <= 16, kept over the block's patch walk, folded over the waves in LDS and over the blocks by ly_mlpblock_bwd_dx_combine (fixed order).
// WG = false (flattened runs / C/4 > 32): dx only — the caller launches ly_wgrad for dWp.
// -------------------------------------------------------------------------------------------------------------------------------------------
struct LyMlpDxArgs {
  const __bf16* g;
  const __bf16* dy;
  const __bf16* x;
  __bf16* dx;
  long M;
  int H, W, n_img, ntiles;
  const uint4* wpt;              // transposed-flipped taps, frag-packed like wp
  float* slab;                   // [gridDim.x][9 * PT * PT * 256]
};

template <int C, bool T2D, bool WG>
__device__ __forceinline__ void ly_mlp_bwd_dx_body(const LyMlpDxArgs& P) {
  using Gm = MlpGeom<C>;
  using T = __bf16;
  typedef ly_u32x4 RV;
  typedef ly_u32x2 R4;
  constexpr int VW = 8, NT = 2, BP = 64 * NT, TH = 4 * NT;
  constexpr int CQ = Gm::CQ, G = Gm::G, SP = Gm::SP, PT = Gm::PT, KP = Gm::KP, RS = Gm::RS, RSP = Gm::RSP;
  constexpr int NFP = PT * SP;
  static_assert(!WG || T2D, "in-kernel weight gradient: 2-D patches only");
  const int H = P.H, W = P.W;
  const long M = P.M;
  const int BPH = T2D ? (TH + 2) * 18 : BP + 2 * W + 2;
  const int PSB = (BPH * RSP + 15 + 64) / 16 * 16;          // bytes of one halo frame (+ slack: transposed reads of the last rows run past CQP channels)

  extern __shared__ f32x4 ly_smem4[];
  char* const wl = reinterpret_cast<char*>(ly_smem4);
  char* const xs = wl + NFP * 1024;
  char* const psg = xs + BP * RS;
  char* const psx = psg + PSB;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  const f32x4 zero = ly_zero4();
  for (int i = tid; i < NFP * 64; i += LY_THREADS) reinterpret_cast<uint4*>(wl)[i] = P.wpt[i];
  auto wlds = [&](int fi) -> bf16x8 { return *reinterpret_cast<const bf16x8*>(wl + (fi * 64 + lane) * 16); };
  if constexpr (WG) {                                       // slack rows past the frames: finite values for the transposed reads
    for (int i = tid; i < 64 / 4; i += LY_THREADS) {
      reinterpret_cast<float*>(psg + PSB - 64)[i] = 0.f;
      reinterpret_cast<float*>(psx + PSB - 64)[i] = 0.f;
    }
  }

  const int tw = T2D ? ((W + 15) >> 4) : 1, th = T2D ? (H + TH - 1) / TH : 1;
  long img0 = 0, p0 = 0;
  int h0 = 0, w0 = 0;
  auto decode = [&](int tile, long& i0, int& hh0, int& ww0, long& q0) {
    if constexpr (T2D) {
      int b = tile;
      const int tx = b % tw; b /= tw;
      const int ty = b % th;
      i0 = (long)(b / th) * H * W;
      hh0 = ty * TH; ww0 = tx * 16;
      q0 = 0;
    } else {
      i0 = 0; hh0 = 0; ww0 = 0;
      q0 = (long)tile * BP;
    }
  };
  // global pixel index of tile-local pixel `pix` of tile (i0, hh0, ww0, q0), or -1
  auto gpix_of = [&](int pix, long i0, int hh0, int ww0, long q0) -> long {
    if constexpr (T2D) {
      const int r = pix >> 4, cx = pix & 15;
      return (hh0 + r < H && ww0 + cx < W) ? i0 + (long)(hh0 + r) * W + ww0 + cx : -1;
    } else {
      const long gp = q0 + pix;
      return gp < M ? gp : -1;
    }
  };

  constexpr int TVN = BP * (KP / VW), NVT = (TVN + LY_THREADS - 1) / LY_THREADS;
  constexpr int HVN = T2D ? (TH + 2) * 18 * G : 1, NVH = T2D ? (HVN + LY_THREADS - 1) / LY_THREADS : 1;
  RV tv[NVT], dv_[NVT];
  R4 hg[NVH], hx[WG ? NVH : 1];
  bool tok[NVT], hok[NVH];
  long toff[NVT];                                           // element offset of the thread's pieces in g / dy / dx (the patch being computed)
  auto issue = [&](int tile, bool keep_off) {
    long i0, q0; int hh0, ww0;
    decode(tile, i0, hh0, ww0, q0);
#pragma unroll
    for (int e = 0; e < NVT; ++e) {
      const int idx = tid + e * LY_THREADS;
      const int pix = idx / (KP / VW), c4 = idx - pix * (KP / VW);
      const long gp = gpix_of(pix, i0, hh0, ww0, q0);
      tok[e] = idx < TVN && gp >= 0 && c4 * VW < C;
      const long off = tok[e] ? gp * C + c4 * VW : 0;
      if (keep_off) toff[e] = tok[e] ? off : -1;
      tv[e] = ly_ldrv<T>(P.g + off);
      dv_[e] = ly_ldrv<T>(P.dy + off);
    }
    if constexpr (T2D) {
#pragma unroll
      for (int e = 0; e < NVH; ++e) {
        const int idx = tid + e * LY_THREADS;
        const int hp = idx / G, c4 = idx - hp * G;
        const int hr = hp / 18, hc = hp - hr * 18;
        const int hh = hh0 - 1 + hr, ww = ww0 - 1 + hc;
        hok[e] = idx < HVN && hh >= 0 && hh < H && ww >= 0 && ww < W;
        const long off = hok[e] ? (i0 + (long)hh * W + ww) * C + c4 * 4 : 0;
        hg[e] = ly_ldr4<T>(P.g + off);
        if constexpr (WG) hx[e] = ly_ldr4<T>(P.x + off);
      }
    }
  };
  RV dyk[NVT];                                              // dy pieces of the patch being computed (the prefetch registers are re-used for the next)
  long koff[NVT];
  auto commit = [&]() {
#pragma unroll
    for (int e = 0; e < NVT; ++e) {
      const int idx = tid + e * LY_THREADS;
      const int pix = idx / (KP / VW), c4 = idx - pix * (KP / VW);
      RV v = tv[e];
      if (!tok[e]) ly_zero_raw(v);
      if (idx < TVN) *reinterpret_cast<RV*>(xs + pix * RS + 2 * VW * c4) = v;
      dyk[e] = dv_[e];
      koff[e] = toff[e];
    }
    if constexpr (T2D) {
#pragma unroll
      for (int e = 0; e < NVH; ++e) {
        const int idx = tid + e * LY_THREADS;
        const int hp = idx / G, c4 = idx - hp * G;
        R4 v = hg[e];
        if (!hok[e]) ly_zero_raw(v);
        if (idx < HVN) *reinterpret_cast<R4*>(psg + hp * RSP + 8 * c4) = v;
        if constexpr (WG) {
          R4 u = hx[e];
          if (!hok[e]) ly_zero_raw(u);
          if (idx < HVN) *reinterpret_cast<R4*>(psx + hp * RSP + 8 * c4) = u;
        }
      }
    }
  };
  auto stage_flat_halo = [&](int tile) {
    const long q0 = (long)tile * BP;
    ly_stage_raw<4, R4>(BPH * G, tid, P.g,
        [&](int idx) -> const void* {
          const int hp = idx / G, c4 = idx - hp * G;
          const long gp = q0 - W - 1 + hp;
          return (gp >= 0 && gp < M) ? P.g + gp * C + c4 * 4 : nullptr;
        },
        [&](int idx, R4 v) {
          const int hp = idx / G, c4 = idx - hp * G;
          *reinterpret_cast<R4*>(psg + hp * RSP + 8 * c4) = v;
        });
  };

  constexpr int NAW = WG ? 9 * PT * PT : 1;
  f32x4 aw[NAW];
#pragma unroll
  for (int i = 0; i < NAW; ++i) aw[i] = zero;

  int tile = ly_xcd_remap((int)blockIdx.x, (int)gridDim.x);
  if (tile < P.ntiles) {
    issue(tile, true);
    commit();
    if constexpr (!T2D) stage_flat_halo(tile);
  }
  __syncthreads();
  const int pixbase = wave * (16 * NT);
  const bf16x4 z4 = __builtin_bit_cast(bf16x4, make_uint2(0u, 0u));
  const int r0 = 4 * lq + (li >> 2), c8 = 8 * (li & 3);

  for (; tile < P.ntiles; tile += gridDim.x) {
    const int nxt = tile + (int)gridDim.x < P.ntiles ? tile + (int)gridDim.x : tile;
    issue(nxt, true);
    decode(tile, img0, h0, w0, p0);

    // ---- weight gradient of the partial conv: both operands transposed out of the halo frames (before the conv overwrites nothing of them) ----
    if constexpr (WG) {
#pragma unroll
      for (int ks = 0; ks < NT / 2; ++ks) {
        int hb[2];                                         // frame offset (tap 0, 0) of the lane's two fetched pixels
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int pix = pixbase + 32 * ks + 16 * e + r0;
          hb[e] = ((pix >> 4) * 18 + (pix & 15)) * RSP + c8;
        }
        bf16x8 ag[PT], bx[9][PT];
#pragma unroll
        for (int t = 0; t < PT; ++t)                       // centre tap of the g frame = g at the pixel itself
          ag[t] = ly_cat8(mb_tr(psg + hb[0] + 19 * RSP + 32 * t), mb_tr(psg + hb[1] + 19 * RSP + 32 * t));
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
          for (int t = 0; t < PT; ++t) {
            const int to = ((tap / 3) * 18 + tap % 3) * RSP + 32 * t;
            bx[tap][t] = ly_cat8(mb_tr(psx + hb[0] + to), mb_tr(psx + hb[1] + to));
          }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
          for (int to = 0; to < PT; ++to)
#pragma unroll
            for (int ti = 0; ti < PT; ++ti) aw[(tap * PT + to) * PT + ti] = ly_mfma_bf16(ag[to], bx[tap][ti], aw[(tap * PT + to) * PT + ti]);
      }
    }

    // ---- pconv^T(g[:C/4]) into the tile's first C/4 channels (the forward's partial conv with the transposed-flipped taps) ----
    {
      uint32_t tmask[NT];
      int pbase[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int pix = pixbase + 16 * n + li;
        if constexpr (T2D) {
          tmask[n] = 0x1ffu;
          pbase[n] = ((pix >> 4) * 18 + (pix & 15)) * RSP;
        } else {
          const long gp = p0 + pix;
          int h_, w_;
          ly_pix_hw(gp < M ? gp : 0, H, W, h_, w_);
          tmask[n] = ly_tapmask(h_, w_, H, W, gp < M);
          pbase[n] = pix * RSP;
        }
      }
      const int rowpitch = T2D ? 18 : W;
      f32x4 accp[PT][NT];
#pragma unroll
      for (int t = 0; t < PT; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) accp[t][n] = zero;
      // operand reads in batches of SB k-steps ahead of their MFMAs (all SP at once for C <= 80; C = 160: 12 k-steps x 5 fragments would be
      // 240 registers — 32 spilled)
      constexpr int SB = SP <= 6 ? SP : 4;
#pragma unroll
      for (int s0 = 0; s0 < SP; s0 += SB) {
        bf16x8 xh[SB][NT], wpf[SB][PT];
#pragma unroll
        for (int sb = 0; sb < SB; ++sb) {
          const int s = s0 + sb;
          if (s >= SP) continue;
          int off[2], tap[2];
          bool gv[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int gq = 8 * s + 4 * h + lq;
            gv[h] = gq < 9 * G;
            tap[h] = gv[h] ? gq / G : 0;
            const int cq4 = gv[h] ? gq - tap[h] * G : 0;
            const int ty = tap[h] / 3, tx = tap[h] - 3 * ty;
            off[h] = (ty * rowpitch + tx) * RSP + 8 * cq4;
          }
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            bf16x4 ph[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const bool ok = gv[h] && (T2D || ((tmask[n] >> tap[h]) & 1u));
              const bf16x4 a = *reinterpret_cast<const bf16x4*>(psg + pbase[n] + off[h]);
              ph[h] = ok ? a : z4;
            }
            xh[sb][n] = ly_cat8(ph[0], ph[1]);
          }
#pragma unroll
          for (int t = 0; t < PT; ++t) wpf[sb][t] = wlds(t * SP + s);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sb = 0; sb < SB; ++sb) {
          if (s0 + sb >= SP) continue;
#pragma unroll
          for (int t = 0; t < PT; ++t)
#pragma unroll
            for (int n = 0; n < NT; ++n) accp[t][n] = ly_mfma_bf16(wpf[sb][t], xh[sb][n], accp[t][n]);
        }
        if (s0 + SB < SP) __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int t = 0; t < PT; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const bf16x4 h = ly_cvtb4(accp[t][n]);
          const int c = 16 * t + 4 * lq;
          const int rb = (pixbase + 16 * n + li) * RS + 2 * c;
          if (c < CQ) *reinterpret_cast<bf16x2*>(xs + rb) = __builtin_shufflevector(h, h, 0, 1);
          if (c + 2 < CQ) *reinterpret_cast<bf16x2*>(xs + rb + 4) = __builtin_shufflevector(h, h, 2, 3);
        }
    }
    __syncthreads();                                         // the tile holds [pconv^T(g_p) | g_r] for every wave's rows

    // ---- dx = dy + tile: every thread its own 16-byte pieces (the mapping of the staging) ----
#pragma unroll
    for (int e = 0; e < NVT; ++e) {
      const int idx = tid + e * LY_THREADS;
      const int pix = idx / (KP / VW), c4 = idx - pix * (KP / VW);
      if (idx < TVN && koff[e] >= 0) {
        const RV zt = *reinterpret_cast<const RV*>(xs + pix * RS + 2 * VW * c4);
        f32x4 a[2], b[2];
        ly_rv_unpack(zt, a);
        ly_rv_unpack(dyk[e], b);
        a[0] += b[0]; a[1] += b[1];
        *reinterpret_cast<RV*>(P.dx + koff[e]) = ly_rv_pack(a, (RV*)nullptr);
      }
    }
    __syncthreads();                                         // every thread has read the tile and the frames
    commit();
    if constexpr (!T2D) stage_flat_halo(nxt);
    __syncthreads();
  }

  if constexpr (WG) {
    float* const red = reinterpret_cast<float*>(xs);
    static_assert(!WG || NAW * 1024 <= BP * RS + 2 * 180 * RSP, "reduction scratch: the tile and the two frames");
    for (int w = 0; w < 4; ++w) {
      if (wave == w) {
#pragma unroll
        for (int i = 0; i < NAW; ++i) {
          f32x4* const p = reinterpret_cast<f32x4*>(red) + i * 64 + lane;
          if (w == 0) *p = aw[i];
          else *p = *p + aw[i];
        }
      }
      __syncthreads();
    }
    f32x4* const out = reinterpret_cast<f32x4*>(P.slab + (size_t)blockIdx.x * (NAW * 256));
    for (int i = tid; i < NAW * 64; i += LY_THREADS) out[i] = reinterpret_cast<const f32x4*>(red)[i];
  }
}

template <int C, bool T2D, bool WG>
__global__ __launch_bounds__(LY_THREADS) void ly_mlpblock_bwd_dx_kernel(const LyMlpDxArgs P) {
  ly_mlp_bwd_dx_body<C, T2D, WG>(P);
}

// dwp[co * lddw + tap * ts + ci * cs] += sum_b slab[b][tap][to][ti][lane][r],  co = 16 to + 4 (lane >> 4) + r, ci = 16 ti + (lane & 15)   (fixed order)
template <int C, int EL>
__global__ __launch_bounds__(1024) void ly_mlpblock_bwd_dx_combine_kernel(const float* __restrict__ slab, const int nblk, float* __restrict__ dwp,
                                                                         const int lddw, const int ts, const int cs) {
  constexpr int PT = MlpGeom<C>::PT, CQ = MlpGeom<C>::CQ, NE = 9 * PT * PT * 64;
  int e;
  f32x4 s;
  if (!ly_mlp_slab_fold<EL>(slab, nblk, NE, e, s)) return;
  const int tile = e >> 6, lane = e & 63;
  const int tap = tile / (PT * PT), to = (tile / PT) % PT, ti = tile % PT;
  const int ci = 16 * ti + (lane & 15);
  if (ci < CQ) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = 16 * to + 4 * (lane >> 4) + r;
      if (co < CQ) dwp[(size_t)co * lddw + (size_t)tap * ts + (size_t)ci * cs] += s[r];
    }
  }
}

// LDS bytes of the tail kernel: transposed-flipped tap fragments + the [128 px][C] tile + the g and x halo frames
template <int C>
static size_t mlp_bwd_dx_lds(bool t2d, int W) {
  using Gm = MlpGeom<C>;
  const long halo = t2d ? 10 * 18 : 128 + 2 * W + 2;
  const size_t psb = ((size_t)halo * Gm::RSP + 15 + 64) / 16 * 16;
  return (size_t)Gm::PT * Gm::SP * 1024 + (size_t)128 * Gm::RS + 2 * psb;
}
static inline bool mlp_bwd_dx_patches(int W) {
  const int wp16 = (W + 15) / 16 * 16;
  return W >= 12 && 4 * (wp16 - W) <= wp16;
}
template <int C>
static bool mlp_bwd_dx_fits(int W) { return mlp_bwd_dx_lds<C>(mlp_bwd_dx_patches(W), W) <= 160 * 1024; }

template <int C, bool T2D, bool WG>
static int launch_mlp_bwd_dx(LyMlpDxArgs P, long slab_floats, float* dwp, int lddw, int ts, int cs, hipStream_t st) {
  using Gm = MlpGeom<C>;
  constexpr int BP = 128, PT = Gm::PT;
  const size_t lds = mlp_bwd_dx_lds<C>(T2D, P.W);
  LY_CHECK(lds <= 160 * 1024, "mlpblock_bwd_dx: tile needs %zu B of LDS (C=%d W=%d)", lds, C, P.W);
  auto k = ly_mlpblock_bwd_dx_kernel<C, T2D, WG>;
  static LyDevOnce once;
  static int per_cu = 0;
  static size_t lds_q = 0;
  if (once.need() || lds_q != lds) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    LY_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    int nb = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(k), LY_THREADS, lds);
    LY_CHECK(e == hipSuccess, "hipOccupancyMaxActiveBlocksPerMultiprocessor: %s", hipGetErrorString(e));
    per_cu = nb < 1 ? 1 : (nb > 4 ? 4 : nb);
    lds_q = lds;
  }
  const long ntiles = T2D ? (long)P.n_img * ((P.H + 7) / 8) * ((P.W + 15) / 16) : (P.M + BP - 1) / BP;
  LY_CHECK(ntiles < (1L << 30), "mlpblock_bwd_dx: too many tiles");
  long blocks = 256L * per_cu;
  if (blocks > ntiles) blocks = ntiles;
  constexpr int SL = 9 * PT * PT * 256;
  if (WG) LY_CHECK(P.slab && dwp && blocks * (long)SL <= slab_floats, "mlpblock_bwd_dx: the slab workspace holds %ld floats, %ld needed", slab_floats, blocks * (long)SL);
  P.ntiles = (int)ntiles;
  hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(LY_THREADS), lds, st, P);
  if (WG) {
    constexpr int NE = 9 * PT * PT * 64, EL = NE >= 2304 ? 32 : 8;      // 72 blocks in both cases
    hipLaunchKernelGGL((ly_mlpblock_bwd_dx_combine_kernel<C, EL>), dim3((NE + EL - 1) / EL), dim3(1024), 0, st, P.slab, (int)blocks, dwp, lddw, ts, cs);
  }
  LY_LAUNCH_CHECK();
  return WG ? 0 : 1;
}

// returns 0: dx and dwp done; 1: dx done, the partial conv's weight gradient is left to the caller (ly_wgrad); < 0: error
template <int C>
static int dispatch_mlp_bwd_dx(LyMlpDxArgs P, long slab_floats, float* dwp, int lddw, int ts, int cs, hipStream_t st) {
  const bool patches = mlp_bwd_dx_patches(P.W);
  if constexpr (MlpGeom<C>::PT <= 2) {
    if (patches && dwp) return launch_mlp_bwd_dx<C, true, true>(P, slab_floats, dwp, lddw, ts, cs, st);
  }
  if (patches) return launch_mlp_bwd_dx<C, true, false>(P, 0, nullptr, 0, 0, 0, st);
  return launch_mlp_bwd_dx<C, false, false>(P, 0, nullptr, 0, 0, 0, st);
}

int ly_mlp_bwd_dx_80(LyMlpDxArgs P, long slab_floats, float* dwp, int lddw, int ts, int cs, hipStream_t st);
int ly_mlp_bwd_dx_160(LyMlpDxArgs P, long slab_floats, float* dwp, int lddw, int ts, int cs, hipStream_t st);
bool ly_mlp_bwd_dx_fits_wide(int C, int W);
