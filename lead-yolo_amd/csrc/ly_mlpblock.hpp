#pragma once
// Fused FasterNet MLPBlock forward (eval / folded-BN form), gfx950; storage dtype T = float (bf16x3 products, two operand
// planes) or __bf16 (plain bf16 products, one plane), fp32 accumulation (ly_tile.hpp).
//
//   y = x + W2 . relu( s * (W1 . [ pconv3x3(x[:, :C/4]) | x[:, C/4:] ]) + b )
//
// Replaces Partial_conv3.forward_split_cat + MLPBlock.forward (reference models/common.py:1432-1437,
// 1478-1482): the split/cat copies, the 2C-wide hidden tensor and the BN/ReLU passes never touch HBM.
// HBM traffic = x once + y once (+ a one-pixel-row halo of the first C/4 channels); the residual is
// taken from the LDS copy of the tile.
//
// Block = 256 threads (4 waves) owns BP = 64*NT consecutive pixels of the flattened N*H*W index
// (NHWC rows, so its input tile is one contiguous span of memory).  The tile is split once into
// bf16 hi/lo planes in LDS (ly_tile.hpp).  Each wave owns 16*NT pixels and carries them through all
// three contractions:
//   1. partial 3x3 conv as an implicit GEMM over K = 9 * ceil4(C/4): operands are 8-byte gathers from
//      a halo image (ps) of the first C/4 channels with per-tap border masks; the result overwrites
//      channels [0, C/4) of the wave's own rows of the tile (the "concat" is a no-op).
//   2. hidden = relu(bn(W1 . row)), HT hidden tiles at a time, kept in registers.
//   3. out += W2[:, hidden pair] . hidden -- two fp32 D tiles of step 2, split in registers, ARE the
//      B operand of one k-step (ly_tile.hpp), so the hidden activations never leave the register file.
#include "ly_tile.hpp"

#ifndef LY_MLP_PD
#define LY_MLP_PD 1              // patches in flight per block of the persistent walk (register sets); measured round 6: 2 and 3 are slower (C = 24: 54.6 -> 62.5 -> 63.5 us): the walk is issue-bound, not latency-bound
#endif

template <int C>
struct MlpGeom {
  static constexpr int CQ = C / 4;
  static constexpr int CQP = (CQ + 3) / 4 * 4;
  static constexpr int G = CQP / 4;                 // 4-channel groups per tap
  static constexpr int SP = (9 * G + 7) / 8;        // pconv k-steps (8 groups each)
  static constexpr int PT = (CQ + 15) / 16;         // pconv output tiles
  static constexpr int C16 = (C + 15) / 16;         // output tiles
  static constexpr int KP = (C + 31) / 32 * 32;
  static constexpr int S1 = KP / 32;                // GEMM1 k-steps
  static constexpr int RS = 2 * KP + 16;            // xs row stride (bytes)
  static constexpr int RSP = 2 * CQP + 8;           // ps row stride (bytes)
  static constexpr int HTP = (2 * C / 16 + 1) / 2 * 2;   // hidden tiles, padded to even
  static constexpr int S2 = HTP / 2;                // GEMM2 k-steps
};

// Weight fragments are NOT staged in LDS (the three packed arrays of one block are up to 470 KB); every wave streams them from
// L2 in the fixed order it consumes them.  Left to the compiler each 2 KB fragment is loaded and waited for right before its
// three MFMAs (vmcnt(0) after every pair of loads: ~0.3 us of exposed L2 latency per fragment, 236 fragments at C=160), so the
// stream is software-pipelined by hand: a ring of D fragments is kept in flight, slot g % D is refilled with fragment g + D
// as soon as fragment g's MFMAs are issued.  All loops are fully unrolled, so g and the slot index are compile-time constants.
// D = 0: no ring (loads where they are used; fewer registers, more co-resident waves -- the better trade for the narrow,
// memory-heavy stages).
template <typename T, int C, int NT, int HT, bool T2D, bool STATS, int D, bool ZONLY = false>
__device__ __forceinline__ void ly_mlpblock_body(
    const T* __restrict__ x, T* __restrict__ y, long M, int H, int W,
    const uint4* __restrict__ wp, const uint4* __restrict__ w1, const uint4* __restrict__ w2,
    const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, double* __restrict__ stats) {
  using Gm = MlpGeom<C>;
  using TR = LyT<T>;
  constexpr int PL = TR::PL, VW = TR::VW;
  constexpr int CQ = Gm::CQ, G = Gm::G, SP = Gm::SP, PT = Gm::PT, C16 = Gm::C16, KP = Gm::KP, S1 = Gm::S1;
  constexpr int RS = Gm::RS, RSP = Gm::RSP, HTP = Gm::HTP, S2 = Gm::S2;
  constexpr int BP = 64 * NT;
  static_assert(HTP % HT == 0 && HT % 2 == 0, "hidden tiles must split evenly into even chunks");
  static_assert(C % 8 == 0, "C must be a multiple of 8");

  // Two tilings of the pixel index space:
  //  * T2D (W % 16 == 0): a block owns a TH x 16 patch (TH = 4*NT rows, one MFMA pixel tile per row);
  //    the halo is the (TH+2) x 18 frame with out-of-image positions staged as zeros, so the partial
  //    conv needs no border masks and tap offsets are compile-time constants.
  //  * flattened run: BP consecutive pixels of the N*H*W index (any W, tiles may span images); halo =
  //    the run extended by W+1 pixels on both sides, border taps masked per lane.
  constexpr int TH = 4 * NT;
  extern __shared__ f32x4 ly_smem4[];
  char* xs_hi = reinterpret_cast<char*>(ly_smem4);
  char* xs_lo = xs_hi + (PL - 1) * BP * RS;               // PL == 1: the "lo" pointers alias hi and are never used
  const int BPH = T2D ? (TH + 2) * 18 : BP + 2 * W + 2;
  char* ps_hi = xs_hi + PL * BP * RS;
  char* ps_lo = ps_hi + (PL - 1) * BPH * RSP;
  // statistics pass: the block's per-channel sums meet in LDS (one [2][HTP*16] float slice PER WAVE behind the halo planes: every
  // address has one owner lane, plain read-add-write in program order — LDS float atomics from four waves summed in arrival order and
  // were one of the sources of run-to-run noise) and leave, the four slices added in a fixed order, with ONE global (double) atomic per
  // channel and block — the four waves of a block used to flush every hidden tile on their own (C = 80: 1280 global atomics per block
  // inside the chunk loop; the statistics pass took 47 us where the whole forward takes 21)
  float* const sacc = reinterpret_cast<float*>(ps_hi + PL * BPH * RSP);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  const f32x4 zero = ly_zero4();
  if constexpr (STATS) {
    for (int i = tid; i < 4 * 2 * HTP * 16; i += LY_THREADS) sacc[i] = 0.f;      // (the staging barrier below orders this before the first add)
  }

  // the fragment stream: partial conv (k-step major), then per hidden chunk GEMM1 (k-step major) and, unless this is the
  // statistics pass, GEMM2 (hidden pair major)
  constexpr int FP = SP * PT, F1 = S1 * HT, F2 = STATS ? 0 : (HT / 2) * C16, FQ = F1 + F2, NFRAG = FP + (HTP / HT) * FQ;
  auto wseq = [&](int g) -> LyWF<PL> {
    if (g < FP) return ly_wfragp<PL>(wp, (g % PT) * SP + g / PT, lane);
    g -= FP;
    const int chunk = g / FQ, r = g - chunk * FQ;
    if (r < F1) return ly_wfragp<PL>(w1, (chunk * HT + r % HT) * S1 + r / HT, lane);
    const int r2 = r - F1;
    return ly_wfragp<PL>(w2, (r2 % C16) * S2 + chunk * (HT / 2) + r2 / C16, lane);
  };
  LyWF<PL> ring[D > 0 ? D : 1];
#pragma unroll
  for (int g = 0; g < D; ++g)
    if (g < NFRAG) ring[g] = wseq(g);
  int g = 0;                   // fragments consumed so far (a constant at every use after unrolling)
  auto wnext = [&]() -> LyWF<PL> {
    if constexpr (D == 0) return wseq(g);
    else return ring[g % D];
  };
  auto wrefill = [&]() {
    if constexpr (D > 0) {
      if (g + D < NFRAG) ring[g % D] = wseq(g + D);
      __builtin_amdgcn_sched_barrier(0x786);   // neither loads nor MFMAs may move across: the refills stay D fragments ahead
    }
    ++g;
  };
  if (C >= 80) {     // large weight sets, few pixels: warm L2 with all three packed weight arrays
    float* const sink = stats ? reinterpret_cast<float*>(stats) : reinterpret_cast<float*>(y);
    if constexpr (!ZONLY) {                                  // (partial conv only: w1 / w2 are not passed)
      ly_l2_warm(w1, (long)HTP * S1 * PL * 1024, sink);
      ly_l2_warm(w2, (long)C16 * S2 * PL * 1024, sink);
    }
    ly_l2_warm(wp, (long)PT * SP * PL * 1024, sink);
  }
  long p0 = 0;                 // flattened: first pixel of the run
  long img0 = 0;               // T2D: pixel index of (n, 0, 0)
  int h0 = 0, w0 = 0;          // T2D: patch origin
  if (T2D) {
    const int tw = W >> 4, th = (H + TH - 1) / TH;
    int b = ly_xcd_remap((int)blockIdx.x, (int)gridDim.x);            // neighbouring patches (shared halo) on one XCD's L2
    const int tx = b % tw; b /= tw;
    const int ty = b % th;
    img0 = (long)(b / th) * H * W;
    h0 = ty * TH; w0 = tx * 16;
  } else {
    p0 = (long)ly_xcd_remap((int)blockIdx.x, (int)gridDim.x) * BP;
  }
  // global pixel index of tile-local pixel `pix` (or -1)
  auto gpix = [&](int pix) -> long {
    if (T2D) {
      const int r = pix >> 4;
      return (h0 + r < H) ? img0 + (long)(h0 + r) * W + w0 + (pix & 15) : -1;
    }
    const long gp = p0 + pix;
    return gp < M ? gp : -1;
  };

  static_assert(C % VW == 0, "a pixel row must be a whole number of 16-byte vectors");
  ly_stage_raw<8, typename TR::RV>(BP * (KP / VW), tid, x,
      [&](int idx) -> const void* {
        const int pix = idx / (KP / VW), c4 = idx - pix * (KP / VW);
        const long gp = gpix(pix);
        return (gp >= 0 && c4 * VW < C) ? x + gp * C + c4 * VW : nullptr;
      },
      [&](int idx, typename TR::RV v) {
        const int pix = idx / (KP / VW), c4 = idx - pix * (KP / VW);
        ly_lds_put_rv(xs_hi, xs_lo, pix * RS, VW * c4, v);
      });
  ly_stage_raw<4, typename TR::R4>(BPH * G, tid, x,
      [&](int idx) -> const void* {
        const int hp = idx / G, c4 = idx - hp * G;
        long gp;
        if (T2D) {
          const int hr = hp / 18, hc = hp - hr * 18;
          const int hh = h0 - 1 + hr, ww = w0 - 1 + hc;
          gp = (hh >= 0 && hh < H && ww >= 0 && ww < W) ? img0 + (long)hh * W + ww : -1;
        } else {
          gp = p0 - W - 1 + hp;
          if (gp >= M) gp = -1;
        }
        return gp >= 0 ? x + gp * C + c4 * 4 : nullptr;
      },
      [&](int idx, typename TR::R4 v) {
        const int hp = idx / G, c4 = idx - hp * G;
        ly_lds_put_r4(ps_hi, ps_lo, hp * RSP, 4 * c4, v);
      });
  __syncthreads();

  const int pixbase = wave * (16 * NT);
  const bf16x4 z4 = __builtin_bit_cast(bf16x4, make_uint2(0u, 0u));

  // ---- 1. partial 3x3 conv -------------------------------------------------------------------
  {
    uint32_t tmask[NT];
    int pbase[NT];             // byte offset of the (ty=0, tx=0) tap of this lane's pixel in the halo image
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int pix = pixbase + 16 * n + li;
      if (T2D) {
        tmask[n] = 0x1ffu;                                   // zeros are staged for out-of-image taps
        pbase[n] = ((pix >> 4) * 18 + (pix & 15)) * RSP;
      } else {
        const long gp = p0 + pix;
        int h_, w_;
        ly_pix_hw(gp, H, W, h_, w_);
        tmask[n] = ly_tapmask(h_, w_, H, W, gp < M);
        pbase[n] = pix * RSP;
      }
    }
    const int rowpitch = T2D ? 18 : W;                       // halo-image pixels per image row
    f32x4 accp[PT][NT];
#pragma unroll
    for (int t = 0; t < PT; ++t)
#pragma unroll
      for (int n = 0; n < NT; ++n) accp[t][n] = zero;

#pragma unroll
    for (int s = 0; s < SP; ++s) {
      int off[2], tap[2];
      bool gv[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int g = 8 * s + 4 * h + lq;
        gv[h] = g < 9 * G;
        tap[h] = gv[h] ? g / G : 0;
        const int cq4 = gv[h] ? g - tap[h] * G : 0;
        const int ty = tap[h] / 3, tx = tap[h] - 3 * ty;
        off[h] = (ty * rowpitch + tx) * RSP + 8 * cq4;
      }
      bf16x8 xh[NT], xl[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int rb = pbase[n];
        bf16x4 ph[2], pl[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bool ok = gv[h] && (T2D || ((tmask[n] >> tap[h]) & 1u));
          const bf16x4 a = *reinterpret_cast<const bf16x4*>(ps_hi + rb + off[h]);
          ph[h] = ok ? a : z4;
          if constexpr (PL == 2) {
            const bf16x4 b = *reinterpret_cast<const bf16x4*>(ps_lo + rb + off[h]);
            pl[h] = ok ? b : z4;
          } else {
            pl[h] = z4;
          }
        }
        xh[n] = ly_cat8(ph[0], ph[1]);
        xl[n] = ly_cat8(pl[0], pl[1]);
      }
#pragma unroll
      for (int t = 0; t < PT; ++t) {
        const LyWF<PL> wf = wnext();
#pragma unroll
        for (int n = 0; n < NT; ++n) accp[t][n] = ly_mfmap<PL>(wf, xh[n], xl[n], accp[t][n]);
        wrefill();
      }
    }
#pragma unroll
    for (int t = 0; t < PT; ++t)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        bf16x4 h, l;
        ly_split4(accp[t][n], h, l);
        const int c = 16 * t + 4 * lq;
        const int rb = (pixbase + 16 * n + li) * RS + 2 * c;
        if (c < CQ) {            // CQ is even: channels (c, c+1) are valid together
          *reinterpret_cast<bf16x2*>(xs_hi + rb) = __builtin_shufflevector(h, h, 0, 1);
          if constexpr (PL == 2) *reinterpret_cast<bf16x2*>(xs_lo + rb) = __builtin_shufflevector(l, l, 0, 1);
        }
        if (c + 2 < CQ) {
          *reinterpret_cast<bf16x2*>(xs_hi + rb + 4) = __builtin_shufflevector(h, h, 2, 3);
          if constexpr (PL == 2) *reinterpret_cast<bf16x2*>(xs_lo + rb + 4) = __builtin_shufflevector(l, l, 2, 3);
        }
      }
  }

  if constexpr (ZONLY) {
    // partial conv only (see the persistent kernel's MODE 2): the tile holds z, the wave stores its own rows
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int pix = pixbase + 16 * n + li;
      const long gp = gpix(pix);
      if (gp >= 0) {
        if constexpr (PL == 1) {
          for (int v = lq; v < C / 8; v += 4)
            *reinterpret_cast<uint4*>(reinterpret_cast<char*>(y) + (gp * C + 8 * v) * 2) = *reinterpret_cast<const uint4*>(xs_hi + pix * RS + 16 * v);
        } else {
          for (int v = lq; v < C / 4; v += 4) {
            const f32x4 r = ly_cvt4(*reinterpret_cast<const bf16x4*>(xs_hi + pix * RS + 8 * v)) + ly_cvt4(*reinterpret_cast<const bf16x4*>(xs_lo + pix * RS + 8 * v));
            ly_st4<T>(y + gp * C + 4 * v, r);
          }
        }
      }
    }
    return;
  }
  // ---- 2 + 3. expand -> BN -> ReLU -> project, hidden kept in registers ---------------------------
  f32x4 acco[C16][NT];
#pragma unroll
  for (int t = 0; t < C16; ++t)
#pragma unroll
    for (int n = 0; n < NT; ++n) acco[t][n] = zero;

#pragma unroll
  for (int hc = 0; hc < HTP / HT; ++hc) {
    f32x4 acch[HT][NT];
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
      for (int n = 0; n < NT; ++n) acch[t][n] = zero;
#pragma unroll
    for (int s = 0; s < S1; ++s) {
      bf16x8 xh[NT], xl[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int rb = (pixbase + 16 * n + li) * RS;
        xh[n] = ly_lds_frag(xs_hi, rb, s, lq);
        if constexpr (PL == 2) xl[n] = ly_lds_frag(xs_lo, rb, s, lq);
        else xl[n] = xh[n];
      }
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        const LyWF<PL> wf = wnext();
#pragma unroll
        for (int n = 0; n < NT; ++n) acch[t][n] = ly_mfmap<PL>(wf, xh[n], xl[n], acch[t][n]);
        wrefill();
      }
    }
    if (STATS) {
      // statistics pass of train-mode BatchNorm: sum / sum of squares of the pre-BN hidden activations
      // over the valid pixels of this block; nothing else is computed or stored
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        f32x4 s1 = zero, s2 = zero;
#pragma unroll
        for (int n = 0; n < NT; ++n)
          if (gpix(pixbase + 16 * n + li) >= 0) {
            s1 += acch[t][n];
            s2 += acch[t][n] * acch[t][n];
          }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s1[r] = ly_group_sum(s1[r], 16);
          s2[r] = ly_group_sum(s2[r], 16);
        }
        if (li == 0) {
          const int ch = (hc * HT + t) * 16 + 4 * lq;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* const sw = sacc + wave * (2 * HTP * 16);
            sw[ch + r] += s1[r];
            sw[HTP * 16 + ch + r] += s2[r];
          }
        }
      }
      continue;
    }
    bf16x4 hh[HT][NT], hl[HT][NT];
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      const int ch = (hc * HT + t) * 16 + 4 * lq;
      const f32x4 sc = ly_ldg4(bn_scale + ch), sh = ly_ldg4(bn_shift + ch);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(acch[t][n][r] * sc[r] + sh[r], 0.f);
        if constexpr (PL == 2) ly_split4(v, hh[t][n], hl[t][n]);
        else { hh[t][n] = ly_cvtb4(v); hl[t][n] = hh[t][n]; }
      }
    }
#pragma unroll
    for (int u = 0; u < HT / 2; ++u) {
      bf16x8 xh[NT], xl[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        xh[n] = ly_cat8(hh[2 * u][n], hh[2 * u + 1][n]);
        xl[n] = ly_cat8(hl[2 * u][n], hl[2 * u + 1][n]);
      }
#pragma unroll
      for (int ct = 0; ct < C16; ++ct) {
        const LyWF<PL> wf = wnext();
#pragma unroll
        for (int n = 0; n < NT; ++n) acco[ct][n] = ly_mfmap<PL>(wf, xh[n], xl[n], acco[ct][n]);
        wrefill();
      }
    }
  }

  if (STATS) {
    __syncthreads();
    double* const st = stats + (size_t)(blockIdx.x & (LY_STATS_STRIPES - 1)) * 2 * (HTP * 16);
    constexpr int SL = 2 * HTP * 16;
    for (int i = tid; i < SL; i += LY_THREADS) atomicAdd(st + i, (double)((sacc[i] + sacc[SL + i]) + (sacc[2 * SL + i] + sacc[3 * SL + i])));
    return;
  }
  // ---- epilogue: residual + store ------------------------------------------------------------
  // The residual x is rebuilt from the bf16 hi/lo planes already in LDS (|err| <= 2^-17 |x|) instead
  // of re-reading global memory: channels < CQP from the halo image's centre tap (the tile's own
  // copy of those channels was overwritten by the partial conv), the rest from the tile.
#pragma unroll
  for (int ct = 0; ct < C16; ++ct)
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int c = 16 * ct + 4 * lq;
      const int pix = pixbase + 16 * n + li;
      const long gp = gpix(pix);
      if (c < C && gp >= 0) {
        bf16x4 rh, rl;
        if (c < Gm::CQP) {
          const int rb = (T2D ? (((pix >> 4) + 1) * 18 + (pix & 15) + 1) : (pix + W + 1)) * RSP + 2 * c;
          rh = *reinterpret_cast<const bf16x4*>(ps_hi + rb);
          if constexpr (PL == 2) rl = *reinterpret_cast<const bf16x4*>(ps_lo + rb);
        } else {
          const int rb = pix * RS + 2 * c;
          rh = *reinterpret_cast<const bf16x4*>(xs_hi + rb);
          if constexpr (PL == 2) rl = *reinterpret_cast<const bf16x4*>(xs_lo + rb);
        }
        f32x4 r = ly_cvt4(rh);
        if constexpr (PL == 2) r += ly_cvt4(rl);
        ly_st4<T>(y + gp * C + c, acco[ct][n] + r);
      }
    }
}

// -------------------------------------------------------------------------------------------------
// Persistent form of the same block for the narrow, memory-heavy stages (C <= 40; 8 x 16 / 16 x 16 pixel patches):
//   * a block loops over patches (grid = what is resident at once); ALL weight fragments of the three contractions are copied to
//     LDS once per block (15-78 KB) instead of being fetched from L2 by every wave of every patch, one dependent round trip each;
//   * the raw pixels of patch i+1 (tile + halo) are requested before patch i is computed and committed to the other LDS buffer
//     after it, so the HBM latency of a patch hides behind the arithmetic of the previous one.
// The arithmetic per pixel is the block above verbatim (same operand order: results are bit-identical).
// -------------------------------------------------------------------------------------------------
// MODE 0: the whole block (y = x' + mlp(x')), 1: statistics pass of the hidden BatchNorm, 2: the partial 3x3 conv alone — y = z =
// [pconv(x[:C/4]) | x[C/4:]], what the training backward needs twice (z for the recomputed hidden tensor, and, with the transposed-
// flipped taps on the gradient g, [d/dx of the conv | g[C/4:]]): one read + one write of the map instead of a clone and a 3x3 launch
// whose 32/64-channel K chunks are 90 % padding at C/4 = 6.
template <typename T, int C, int NT, int HT, int MODE, int PD>
__device__ __forceinline__ void ly_mlpblock_persist_body(
    const T* __restrict__ x, T* __restrict__ y, long M, int H, int W, int n_img,
    const uint4* __restrict__ wp, const uint4* __restrict__ w1, const uint4* __restrict__ w2,
    const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, double* __restrict__ stats) {
  using Gm = MlpGeom<C>;
  using TR = LyT<T>;
  using RV = typename TR::RV;
  using R4 = typename TR::R4;
  constexpr int PL = TR::PL, VW = TR::VW;
  constexpr int CQ = Gm::CQ, G = Gm::G, SP = Gm::SP, PT = Gm::PT, C16 = Gm::C16, KP = Gm::KP, S1 = Gm::S1;
  constexpr int RS = Gm::RS, RSP = Gm::RSP, HTP = Gm::HTP, S2 = Gm::S2;
  constexpr int BP = 64 * NT, TH = 4 * NT, BPH = (TH + 2) * 18;
  constexpr bool T2D = true, STATS = MODE == 1, ZONLY = MODE == 2;
  constexpr int NFP = PT * SP, NF1 = HTP * S1, NF2 = C16 * S2, NFW = NFP + NF1 + NF2;
  constexpr int WBYTES = NFW * PL * 1024, XB = PL * BP * RS, PB = PL * BPH * RSP, BUFB = (XB + PB + 15) / 16 * 16;
  static_assert(HTP % HT == 0 && HT % 2 == 0 && C % VW == 0, "geometry");
  extern __shared__ f32x4 ly_smem4[];
  char* const wl = reinterpret_cast<char*>(ly_smem4);
  char* const bufs = wl + WBYTES;
  const long p0 = 0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  const f32x4 zero = ly_zero4();

  // ---- weights -> LDS, once -------------------------------------------------------------------
  for (int i = tid; i < NFP * PL * 64; i += LY_THREADS) reinterpret_cast<uint4*>(wl)[i] = wp[i];
  if constexpr (!ZONLY) {
    for (int i = tid; i < NF1 * PL * 64; i += LY_THREADS) reinterpret_cast<uint4*>(wl)[NFP * PL * 64 + i] = w1[i];
    for (int i = tid; i < NF2 * PL * 64; i += LY_THREADS) reinterpret_cast<uint4*>(wl)[(NFP + NF1) * PL * 64 + i] = w2[i];
  }
  auto wlds = [&](int fi) -> LyWF<PL> {
    LyWF<PL> f;
    f.hi = *reinterpret_cast<const bf16x8*>(wl + ((fi * PL) * 64 + lane) * 16);
    if constexpr (PL == 2) f.lo = *reinterpret_cast<const bf16x8*>(wl + ((fi * PL + 1) * 64 + lane) * 16);
    return f;
  };
  constexpr int FP = SP * PT, F1 = S1 * HT, F2 = (HT / 2) * C16, FQ = F1 + F2;
  auto wseq = [&](int g) -> LyWF<PL> {
    if (g < FP) return wlds((g % PT) * SP + g / PT);
    g -= FP;
    const int chunk = g / FQ, r = g - chunk * FQ;
    if (r < F1) return wlds(NFP + (chunk * HT + r % HT) * S1 + r / HT);
    const int r2 = r - F1;
    return wlds(NFP + NF1 + (r2 % C16) * S2 + chunk * (HT / 2) + r2 / C16);
  };

  // ---- patch walk ------------------------------------------------------------------------------
  const int tw = W >> 4, th = (H + TH - 1) / TH;
  const int ntiles = n_img * th * tw;
  long img0 = 0;
  int h0 = 0, w0 = 0;
  auto decode = [&](int tile, long& i0, int& hh0, int& ww0) {
    int b = tile;
    const int tx = b % tw; b /= tw;
    const int ty = b % th;
    i0 = (long)(b / th) * H * W;
    hh0 = ty * TH; ww0 = tx * 16;
  };
  constexpr int TVN = BP * (KP / VW), NVT = (TVN + LY_THREADS - 1) / LY_THREADS;
  constexpr int HVN = BPH * G, NVH = (HVN + LY_THREADS - 1) / LY_THREADS;
  // PD register sets: set k holds the raw pixels of the patch PD walk steps ahead of the one being computed (round 6: with ONE patch in
  // flight per block a walk step was one HBM round trip, ~2 us for 0.6 us of arithmetic: 2.7 / 1.8 TB/s at C = 24 / 40 with 3 / 2 blocks per CU)
  RV tv[PD][NVT];
  R4 hv[PD][NVH];
  bool tok[PD][NVT], hok[PD][NVH];
  auto issue = [&](const int k, int tile) {
    long i0; int hh0, ww0;
    decode(tile, i0, hh0, ww0);
#pragma unroll
    for (int e = 0; e < NVT; ++e) {
      const int idx = tid + e * LY_THREADS;
      const int pix = idx / (KP / VW), c4 = idx - pix * (KP / VW);
      const int r = pix >> 4;
      tok[k][e] = idx < TVN && hh0 + r < H && c4 * VW < C;
      tv[k][e] = ly_ldrv<T>(tok[k][e] ? x + (i0 + (long)(hh0 + r) * W + ww0 + (pix & 15)) * C + c4 * VW : x);
    }
#pragma unroll
    for (int e = 0; e < NVH; ++e) {
      const int idx = tid + e * LY_THREADS;
      const int hp = idx / G, c4 = idx - hp * G;
      const int hr = hp / 18, hc = hp - hr * 18;
      const int hh = hh0 - 1 + hr, ww = ww0 - 1 + hc;
      hok[k][e] = idx < HVN && hh >= 0 && hh < H && ww >= 0 && ww < W;
      hv[k][e] = ly_ldr4<T>(hok[k][e] ? x + (i0 + (long)hh * W + ww) * C + c4 * 4 : x);
    }
  };
  auto commit = [&](const int k, int buf) {
    char* xh_ = bufs + buf * BUFB;
    char* xl_ = xh_ + (PL - 1) * BP * RS;
    char* ph_ = xh_ + XB;
    char* pl_ = ph_ + (PL - 1) * BPH * RSP;
#pragma unroll
    for (int e = 0; e < NVT; ++e) {
      const int idx = tid + e * LY_THREADS;
      const int pix = idx / (KP / VW), c4 = idx - pix * (KP / VW);
      RV v = tv[k][e];
      if (!tok[k][e]) ly_zero_raw(v);
      if (idx < TVN) ly_lds_put_rv(xh_, xl_, pix * RS, VW * c4, v);
    }
#pragma unroll
    for (int e = 0; e < NVH; ++e) {
      const int idx = tid + e * LY_THREADS;
      const int hp = idx / G, c4 = idx - hp * G;
      R4 v = hv[k][e];
      if (!hok[k][e]) ly_zero_raw(v);
      if (idx < HVN) ly_lds_put_r4(ph_, pl_, hp * RSP, 4 * c4, v);
    }
  };

  // statistics pass: the sums of a block's patches stay in registers until the walk ends (one flush per block)
  f32x4 st1[STATS ? HTP : 1], st2[STATS ? HTP : 1];
  if constexpr (STATS) {
#pragma unroll
    for (int t = 0; t < HTP; ++t) { st1[t] = zero; st2[t] = zero; }
  }
  const int tile0 = ly_xcd_remap((int)blockIdx.x, (int)gridDim.x);       // blocks of one XCD walk neighbouring patches (shared halo rows in its L2)
  if (tile0 >= ntiles) return;
  const int GS = (int)gridDim.x;
  issue(0, tile0);
  commit(0, 0);
#pragma unroll
  for (int k = 0; k < PD; ++k) issue(k, tile0 + (k + 1) * GS < ntiles ? tile0 + (k + 1) * GS : tile0);       // (a patch past the end re-requests a valid one: straight-line loads)
  __syncthreads();                                         // weights and the first patch are in LDS
  int buf = 0;
  for (int base = tile0; base < ntiles; base += PD * GS) {
#pragma unroll
  for (int pk = 0; pk < PD; ++pk) {
    const int tile = base + pk * GS;
    if (tile >= ntiles) break;
    // set pk holds patch tile + GS (requested PD walk steps ago): into the other LDS buffer while this patch is computed, and the set goes out
    // again for patch tile + (PD + 1) GS
    commit(pk, buf ^ 1);
    issue(pk, tile + (PD + 1) * GS < ntiles ? tile + (PD + 1) * GS : tile);
    decode(tile, img0, h0, w0);
    char* const xs_hi = bufs + buf * BUFB;
    char* const xs_lo = xs_hi + (PL - 1) * BP * RS;
    char* const ps_hi = xs_hi + XB;
    char* const ps_lo = ps_hi + (PL - 1) * BPH * RSP;
    auto gpix = [&](int pix) -> long {
      const int r = pix >> 4;
      return (h0 + r < H) ? img0 + (long)(h0 + r) * W + w0 + (pix & 15) : -1;
    };
    int g = 0;
    auto wnext = [&]() -> LyWF<PL> { return wseq(g); };
    auto wrefill = [&]() { ++g; };
    const int pixbase = wave * (16 * NT);
    const bf16x4 z4 = __builtin_bit_cast(bf16x4, make_uint2(0u, 0u));

    // ---- 1. partial 3x3 conv -------------------------------------------------------------------
    {
      uint32_t tmask[NT];
      int pbase[NT];             // byte offset of the (ty=0, tx=0) tap of this lane's pixel in the halo image
  #pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int pix = pixbase + 16 * n + li;
        if (T2D) {
          tmask[n] = 0x1ffu;                                   // zeros are staged for out-of-image taps
          pbase[n] = ((pix >> 4) * 18 + (pix & 15)) * RSP;
        } else {
          const long gp = p0 + pix;
          int h_, w_;
          ly_pix_hw(gp, H, W, h_, w_);
          tmask[n] = ly_tapmask(h_, w_, H, W, gp < M);
          pbase[n] = pix * RSP;
        }
      }
      const int rowpitch = T2D ? 18 : W;                       // halo-image pixels per image row
      f32x4 accp[PT][NT];
  #pragma unroll
      for (int t = 0; t < PT; ++t)
  #pragma unroll
        for (int n = 0; n < NT; ++n) accp[t][n] = zero;

  #pragma unroll
      for (int s = 0; s < SP; ++s) {
        int off[2], tap[2];
        bool gv[2];
  #pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int g = 8 * s + 4 * h + lq;
          gv[h] = g < 9 * G;
          tap[h] = gv[h] ? g / G : 0;
          const int cq4 = gv[h] ? g - tap[h] * G : 0;
          const int ty = tap[h] / 3, tx = tap[h] - 3 * ty;
          off[h] = (ty * rowpitch + tx) * RSP + 8 * cq4;
        }
        bf16x8 xh[NT], xl[NT];
  #pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int rb = pbase[n];
          bf16x4 ph[2], pl[2];
  #pragma unroll
          for (int h = 0; h < 2; ++h) {
            const bool ok = gv[h] && (T2D || ((tmask[n] >> tap[h]) & 1u));
            const bf16x4 a = *reinterpret_cast<const bf16x4*>(ps_hi + rb + off[h]);
            ph[h] = ok ? a : z4;
            if constexpr (PL == 2) {
              const bf16x4 b = *reinterpret_cast<const bf16x4*>(ps_lo + rb + off[h]);
              pl[h] = ok ? b : z4;
            } else {
              pl[h] = z4;
            }
          }
          xh[n] = ly_cat8(ph[0], ph[1]);
          xl[n] = ly_cat8(pl[0], pl[1]);
        }
  #pragma unroll
        for (int t = 0; t < PT; ++t) {
          const LyWF<PL> wf = wnext();
  #pragma unroll
          for (int n = 0; n < NT; ++n) accp[t][n] = ly_mfmap<PL>(wf, xh[n], xl[n], accp[t][n]);
          wrefill();
        }
      }
  #pragma unroll
      for (int t = 0; t < PT; ++t)
  #pragma unroll
        for (int n = 0; n < NT; ++n) {
          bf16x4 h, l;
          ly_split4(accp[t][n], h, l);
          const int c = 16 * t + 4 * lq;
          const int rb = (pixbase + 16 * n + li) * RS + 2 * c;
          if (c < CQ) {            // CQ is even: channels (c, c+1) are valid together
            *reinterpret_cast<bf16x2*>(xs_hi + rb) = __builtin_shufflevector(h, h, 0, 1);
            if constexpr (PL == 2) *reinterpret_cast<bf16x2*>(xs_lo + rb) = __builtin_shufflevector(l, l, 0, 1);
          }
          if (c + 2 < CQ) {
            *reinterpret_cast<bf16x2*>(xs_hi + rb + 4) = __builtin_shufflevector(h, h, 2, 3);
            if constexpr (PL == 2) *reinterpret_cast<bf16x2*>(xs_lo + rb + 4) = __builtin_shufflevector(l, l, 2, 3);
          }
        }
    }

    if constexpr (ZONLY) {
      // the tile now holds z: the wave stores its own rows (16-byte vectors, lanes of a row split the channels)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int pix = pixbase + 16 * n + li;
        const long gp = gpix(pix);
        if (gp >= 0) {
          if constexpr (PL == 1) {
            for (int v = lq; v < C / 8; v += 4)
              *reinterpret_cast<uint4*>(reinterpret_cast<char*>(y) + (gp * C + 8 * v) * 2) = *reinterpret_cast<const uint4*>(xs_hi + pix * RS + 16 * v);
          } else {
            for (int v = lq; v < C / 4; v += 4) {
              const f32x4 r = ly_cvt4(*reinterpret_cast<const bf16x4*>(xs_hi + pix * RS + 8 * v)) + ly_cvt4(*reinterpret_cast<const bf16x4*>(xs_lo + pix * RS + 8 * v));
              ly_st4<T>(y + gp * C + 4 * v, r);
            }
          }
        }
      }
      __syncthreads();
      buf ^= 1;
      continue;
    }
    // ---- 2 + 3. expand -> BN -> ReLU -> project, hidden kept in registers ---------------------------
    f32x4 acco[C16][NT];
  #pragma unroll
    for (int t = 0; t < C16; ++t)
  #pragma unroll
      for (int n = 0; n < NT; ++n) acco[t][n] = zero;

  #pragma unroll
    for (int hc = 0; hc < HTP / HT; ++hc) {
      f32x4 acch[HT][NT];
  #pragma unroll
      for (int t = 0; t < HT; ++t)
  #pragma unroll
        for (int n = 0; n < NT; ++n) acch[t][n] = zero;
  #pragma unroll
      for (int s = 0; s < S1; ++s) {
        bf16x8 xh[NT], xl[NT];
  #pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int rb = (pixbase + 16 * n + li) * RS;
          xh[n] = ly_lds_frag(xs_hi, rb, s, lq);
          if constexpr (PL == 2) xl[n] = ly_lds_frag(xs_lo, rb, s, lq);
          else xl[n] = xh[n];
        }
  #pragma unroll
        for (int t = 0; t < HT; ++t) {
          const LyWF<PL> wf = wnext();
  #pragma unroll
          for (int n = 0; n < NT; ++n) acch[t][n] = ly_mfmap<PL>(wf, xh[n], xl[n], acch[t][n]);
          wrefill();
        }
      }
      if constexpr (STATS) {
        // statistics pass of train-mode BatchNorm: sum / sum of squares of the pre-BN hidden activations over the valid pixels
  #pragma unroll
        for (int t = 0; t < HT; ++t)
  #pragma unroll
          for (int n = 0; n < NT; ++n)
            if (gpix(pixbase + 16 * n + li) >= 0) {
              st1[hc * HT + t] += acch[t][n];
              st2[hc * HT + t] += acch[t][n] * acch[t][n];
            }
        continue;
      }
      bf16x4 hh[HT][NT], hl[HT][NT];
  #pragma unroll
      for (int t = 0; t < HT; ++t) {
        const int ch = (hc * HT + t) * 16 + 4 * lq;
        const f32x4 sc = ly_ldg4(bn_scale + ch), sh = ly_ldg4(bn_shift + ch);
  #pragma unroll
        for (int n = 0; n < NT; ++n) {
          f32x4 v;
  #pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(acch[t][n][r] * sc[r] + sh[r], 0.f);
          if constexpr (PL == 2) ly_split4(v, hh[t][n], hl[t][n]);
          else { hh[t][n] = ly_cvtb4(v); hl[t][n] = hh[t][n]; }
        }
      }
  #pragma unroll
      for (int u = 0; u < HT / 2; ++u) {
        bf16x8 xh[NT], xl[NT];
  #pragma unroll
        for (int n = 0; n < NT; ++n) {
          xh[n] = ly_cat8(hh[2 * u][n], hh[2 * u + 1][n]);
          xl[n] = ly_cat8(hl[2 * u][n], hl[2 * u + 1][n]);
        }
  #pragma unroll
        for (int ct = 0; ct < C16; ++ct) {
          const LyWF<PL> wf = wnext();
  #pragma unroll
          for (int n = 0; n < NT; ++n) acco[ct][n] = ly_mfmap<PL>(wf, xh[n], xl[n], acco[ct][n]);
          wrefill();
        }
      }
    }

    if constexpr (!STATS) {
    // ---- epilogue: residual + store ------------------------------------------------------------
    // The residual x is rebuilt from the bf16 hi/lo planes already in LDS (|err| <= 2^-17 |x|) instead
    // of re-reading global memory: channels < CQP from the halo image's centre tap (the tile's own
    // copy of those channels was overwritten by the partial conv), the rest from the tile.
  #pragma unroll
    for (int ct = 0; ct < C16; ++ct)
  #pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int c = 16 * ct + 4 * lq;
        const int pix = pixbase + 16 * n + li;
        const long gp = gpix(pix);
        if (c < C && gp >= 0) {
          bf16x4 rh, rl;
          if (c < Gm::CQP) {
            const int rb = (T2D ? (((pix >> 4) + 1) * 18 + (pix & 15) + 1) : (pix + W + 1)) * RSP + 2 * c;
            rh = *reinterpret_cast<const bf16x4*>(ps_hi + rb);
            if constexpr (PL == 2) rl = *reinterpret_cast<const bf16x4*>(ps_lo + rb);
          } else {
            const int rb = pix * RS + 2 * c;
            rh = *reinterpret_cast<const bf16x4*>(xs_hi + rb);
            if constexpr (PL == 2) rl = *reinterpret_cast<const bf16x4*>(xs_lo + rb);
          }
          f32x4 r = ly_cvt4(rh);
          if constexpr (PL == 2) r += ly_cvt4(rl);
          ly_st4<T>(y + gp * C + c, acco[ct][n] + r);
        }
      }
    }
    __syncthreads();
    buf ^= 1;
  }
  }
  if constexpr (STATS) {
#pragma unroll
    for (int t = 0; t < HTP; ++t) ly_stats_flush(stats, HTP * 16, t * 16 + 4 * lq, st1[t], st2[t]);
  }
}

template <typename T, int C, int NT, int HT, int MODE, int PD>
__global__ __launch_bounds__(LY_THREADS) void ly_mlpblock_persist_kernel(
    const T* __restrict__ x, T* __restrict__ y, long M, int H, int W, int n_img,
    const uint4* __restrict__ wp, const uint4* __restrict__ w1, const uint4* __restrict__ w2,
    const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, double* __restrict__ stats) {
  ly_mlpblock_persist_body<T, C, NT, HT, MODE, PD>(x, y, M, H, W, n_img, wp, w1, w2, bn_scale, bn_shift, stats);
}

template <typename T, int C, int NT, int HT, bool T2D, bool STATS>
__global__ __launch_bounds__(LY_THREADS) void ly_mlpblock_fwd_kernel(
    const T* __restrict__ x, T* __restrict__ y, long M, int H, int W,
    const uint4* __restrict__ wp, const uint4* __restrict__ w1, const uint4* __restrict__ w2,
    const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, double* __restrict__ stats) {
  ly_mlpblock_body<T, C, NT, HT, T2D, STATS, 0>(x, y, M, H, W, wp, w1, w2, bn_scale, bn_shift, stats);
}

// the narrowest stages (C <= 24, 8 x 16 patches) are pure HBM streams: four resident waves per SIMD instead of three
// (116 instead of 136 registers, no spills) is +6 % (64.8 -> 60.8 us at 160 x 160 x 32); C = 40 loses 8 % with it
template <typename T, int C, int NT, int HT, bool T2D, bool STATS>
__global__ __launch_bounds__(LY_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void ly_mlpblock_fwd_occ4_kernel(
    const T* __restrict__ x, T* __restrict__ y, long M, int H, int W,
    const uint4* __restrict__ wp, const uint4* __restrict__ w1, const uint4* __restrict__ w2,
    const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, double* __restrict__ stats) {
  ly_mlpblock_body<T, C, NT, HT, T2D, STATS, 0>(x, y, M, H, W, wp, w1, w2, bn_scale, bn_shift, stats);
}

template <typename T, int C, int NT, int HT, bool T2D, bool STATS>
__global__ __launch_bounds__(LY_THREADS) __attribute__((amdgpu_waves_per_eu(1, 2))) void ly_mlpblock_fwd_ring_kernel(
    const T* __restrict__ x, T* __restrict__ y, long M, int H, int W,
    const uint4* __restrict__ wp, const uint4* __restrict__ w1, const uint4* __restrict__ w2,
    const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, double* __restrict__ stats) {
  ly_mlpblock_body<T, C, NT, HT, T2D, STATS, 8>(x, y, M, H, W, wp, w1, w2, bn_scale, bn_shift, stats);
}

template <typename T, int C, int NT, int HT, bool T2D>
__global__ __launch_bounds__(LY_THREADS) void ly_mlpblock_pconv1_kernel(const T* __restrict__ x, T* __restrict__ y, long M, int H, int W, const uint4* __restrict__ wp) {
  ly_mlpblock_body<T, C, NT, HT, T2D, false, 0, true>(x, y, M, H, W, wp, wp, wp, nullptr, nullptr, nullptr);
}

template <typename T, int C, int NT, int HT, bool T2D>
static int launch_mlp_pconv1(const T* x, T* y, long M, int n_img, int H, int W, const void* wp, hipStream_t st) {
  using Gm = MlpGeom<C>;
  constexpr int BP = 64 * NT;
  const long halo = T2D ? (4 * NT + 2) * 18 : BP + 2 * W + 2;
  size_t lds = LyT<T>::PL * ((size_t)BP * Gm::RS + (size_t)halo * Gm::RSP);
  LY_CHECK(lds <= 160 * 1024, "mlpblock_pconv: tile needs %zu B of LDS (C=%d W=%d)", lds, C, W);
  auto k = ly_mlpblock_pconv1_kernel<T, C, NT, HT, T2D>;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    LY_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    configured = true;
  }
  long blocks = T2D ? (long)n_img * ((H + 4 * NT - 1) / (4 * NT)) * (W / 16) : (M + BP - 1) / BP;
  hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(LY_THREADS), lds, st, x, y, M, H, W, reinterpret_cast<const uint4*>(wp));
  LY_LAUNCH_CHECK();
  return 0;
}

// The instantiations are split over translation units (ly_mlpblock.hip: C = 16/24/40 + the C ABI, ly_mlpblock_b.hip:
// C = 80/160, ly_mlpblock_c.hip: C = 320) only to keep the in-tree build short.
#define LY_MLP_ARGS const void* x, void* y, long M, int n_img, int H, int W, const void* wp, const void* w1, const void* w2, \
                    const float* s, const float* b, double* stats, int dtype, hipStream_t st
int ly_mlp_dispatch_80(LY_MLP_ARGS);
int ly_mlp_dispatch_160(LY_MLP_ARGS);
int ly_mlp_dispatch_320(LY_MLP_ARGS);
int ly_mlp_pconv_80(const void* x, void* y, long M, int n_img, int H, int W, const void* wp, int dtype, hipStream_t st);
int ly_mlp_pconv_160(const void* x, void* y, long M, int n_img, int H, int W, const void* wp, int dtype, hipStream_t st);
int ly_mlp_pconv_320(const void* x, void* y, long M, int n_img, int H, int W, const void* wp, int dtype, hipStream_t st);

template <typename T, int C, int NT, int HT, bool T2D, bool STATS, bool RING>
static int launch_mlp_k(const T* x, T* y, long M, int n_img, int H, int W, const void* wp, const void* w1, const void* w2,
                        const float* s, const float* b, double* stats, hipStream_t st) {
  using Gm = MlpGeom<C>;
  constexpr int BP = 64 * NT;
  const long halo = T2D ? (4 * NT + 2) * 18 : BP + 2 * W + 2;
  size_t lds = LyT<T>::PL * ((size_t)BP * Gm::RS + (size_t)halo * Gm::RSP) + (STATS ? 4 * 2 * Gm::HTP * 16 * sizeof(float) : 0);
  LY_CHECK(lds <= 160 * 1024, "mlpblock: tile needs %zu B of LDS (C=%d W=%d)", lds, C, W);
  void (*k)(const T*, T*, long, int, int, const uint4*, const uint4*, const uint4*, const float*, const float*, double*);
  if constexpr (RING) k = ly_mlpblock_fwd_ring_kernel<T, C, NT, HT, T2D, STATS>;
  else if constexpr (C <= 24 && T2D) k = ly_mlpblock_fwd_occ4_kernel<T, C, NT, HT, T2D, STATS>;
  else k = ly_mlpblock_fwd_kernel<T, C, NT, HT, T2D, STATS>;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    LY_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    configured = true;
  }
  long blocks = T2D ? (long)n_img * ((H + 4 * NT - 1) / (4 * NT)) * (W / 16) : (M + BP - 1) / BP;
  hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(LY_THREADS), lds, st, x, y, M, H, W, reinterpret_cast<const uint4*>(wp),
                     reinterpret_cast<const uint4*>(w1), reinterpret_cast<const uint4*>(w2), s, b, stats);
  LY_LAUNCH_CHECK();
  return 0;
}

template <typename T, int C, int NT, int HT, int MODE>
static int launch_mlp_persist(const T* x, T* y, long M, int n_img, int H, int W, const void* wp, const void* w1, const void* w2,
                              const float* s, const float* b, double* stats, hipStream_t st) {
  using Gm = MlpGeom<C>;
  constexpr int PL = LyT<T>::PL, BP = 64 * NT, BPH = (4 * NT + 2) * 18;
  constexpr int NFW = Gm::PT * Gm::SP + Gm::HTP * Gm::S1 + Gm::C16 * Gm::S2;
  constexpr size_t lds = (size_t)NFW * PL * 1024 + 2 * (((size_t)PL * BP * Gm::RS + (size_t)PL * BPH * Gm::RSP + 15) / 16 * 16);
  static_assert(lds <= 160 * 1024, "persistent MLPBlock: weights + two patch buffers must fit LDS");
  constexpr int PD = LY_MLP_PD;
  auto k = ly_mlpblock_persist_kernel<T, C, NT, HT, MODE, PD>;
  static int per_cu = 0;
  if (per_cu == 0) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    LY_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    int nb = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(k), LY_THREADS, lds);
    LY_CHECK(e == hipSuccess, "hipOccupancyMaxActiveBlocksPerMultiprocessor: %s", hipGetErrorString(e));
    per_cu = nb < 1 ? 1 : (nb > 8 ? 8 : nb);
  }
  const long ntiles = (long)n_img * ((H + 4 * NT - 1) / (4 * NT)) * (W / 16);
  long blocks = 256L * per_cu;
  if (blocks > ntiles) blocks = ntiles;
  hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(LY_THREADS), lds, st, x, y, M, H, W, n_img, reinterpret_cast<const uint4*>(wp),
                     reinterpret_cast<const uint4*>(w1), reinterpret_cast<const uint4*>(w2), s, b, stats);
  LY_LAUNCH_CHECK();
  return 0;
}

template <typename T, int C, int NT, int HT, bool T2D>
static int launch_mlp(const T* x, T* y, long M, int n_img, int H, int W, const void* wp, const void* w1, const void* w2,
                      const float* s, const float* b, double* stats, hipStream_t st) {
  constexpr bool RING = C >= 80;
  if (stats) return launch_mlp_k<T, C, NT, HT, T2D, true, RING>(x, y, M, n_img, H, W, wp, w1, w2, s, b, stats, st);
  return launch_mlp_k<T, C, NT, HT, T2D, false, RING>(x, y, M, n_img, H, W, wp, w1, w2, s, b, stats, st);
}

// Tiling policy (kernel time at bs=32 / 64):
//  * C < 80 (wide maps, few weights; HBM-heavy): 2-D patches (8 x 16 px per block: small halo, 4+ co-resident blocks per CU)
//    where the map is wide and a multiple of 16; flattened runs otherwise, with as many pixel tiles per wave as still fill
//    the chip.  No fragment ring: it costs the co-residency these stages live on (C=24: 64 -> 69 us with it).
//  * C >= 80 (small maps, 134-470 KB of weights per block; bound by streaming the fragments from L2): ring kernel, and two
//    pixel tiles per wave -- half the fragment traffic per pixel -- as soon as that still leaves >= 200 blocks
//    (C=80 @ 40x40x32: 26.5 -> 20.7 us; C=160 @ 20x20: 22.6 us with one tile at bs=32, 38.4 -> 28.8 us with two at bs=64).
template <typename T, int C, int HT, int NTMAX>
static int dispatch_nt_t(const T* x, T* y, long M, int n_img, int H, int W, const void* wp, const void* w1, const void* w2,
                         const float* s, const float* b, double* stats, hipStream_t st) {
  constexpr int NT2 = NTMAX >= 2 ? 2 : NTMAX, NT4 = NTMAX >= 4 ? 4 : NTMAX;
  if (C >= 80) {
    // bf16, C = 80: four pixel tiles per wave (a quarter of the fragment stream per pixel; 236 registers, two waves per SIMD) once that
    // still leaves 384 blocks: 40 x 40 x 64 27.6 -> 25.0 us per block of the stage; at 200 blocks (bs = 32) the forward lost 1 %
    if constexpr (NTMAX >= 4 && LyT<T>::BF) {
      if (M >= 384L * 256) return launch_mlp<T, C, NT4, HT, false>(x, y, M, n_img, H, W, wp, w1, w2, s, b, stats, st);
    }
    if (NTMAX >= 2 && M >= 200L * 128) return launch_mlp<T, C, NT2, HT, false>(x, y, M, n_img, H, W, wp, w1, w2, s, b, stats, st);
    return launch_mlp<T, C, 1, HT, false>(x, y, M, n_img, H, W, wp, w1, w2, s, b, stats, st);
  }
  if constexpr (C < 80) {
    // persistent patch walk with the weights in LDS: enough patches for every resident block to amortise the weight copy
    // (fp32 storage at C = 40 needs 148 KB of LDS and 242 registers: one block per CU, measured 63 -> 79 us — it keeps the one-shot kernel)
    if ((W & 15) == 0 && W >= 32 && (long)n_img * ((H + 7) / 8) * (W / 16) >= 1024 && (LyT<T>::BF || C < 40))
      return stats ? launch_mlp_persist<T, C, 2, HT, 1>(x, y, M, n_img, H, W, wp, w1, w2, s, b, stats, st)
                   : launch_mlp_persist<T, C, 2, HT, 0>(x, y, M, n_img, H, W, wp, w1, w2, s, b, stats, st);
  }
  if ((W & 15) == 0 && W >= 64 && NTMAX >= 2) return launch_mlp<T, C, 2, HT, true>(x, y, M, n_img, H, W, wp, w1, w2, s, b, stats, st);
  if (NTMAX >= 4 && M >= 256L * 1024) return launch_mlp<T, C, NT4, HT, false>(x, y, M, n_img, H, W, wp, w1, w2, s, b, stats, st);
  if (NTMAX >= 2 && M >= 128L * 512) return launch_mlp<T, C, NT2, HT, false>(x, y, M, n_img, H, W, wp, w1, w2, s, b, stats, st);
  return launch_mlp<T, C, 1, HT, false>(x, y, M, n_img, H, W, wp, w1, w2, s, b, stats, st);
}

// the partial conv alone (persist MODE 2) where the persistent kernel applies; returns 1 = "not built for this shape" (the caller then
// uses a clone + ly_conv3x3_fwd)
template <int C, int HT>
static int dispatch_pconv(const void* x, void* y, long M, int n_img, int H, int W, const void* wp, int dtype, hipStream_t st) {
  if constexpr (C < 80) {
    const bool shape_ok = (W & 15) == 0 && W >= 32 && (long)n_img * ((H + 7) / 8) * (W / 16) >= 1024;
    if (shape_ok && dtype == LY_BF16)
      return launch_mlp_persist<__bf16, C, 2, HT, 2>(reinterpret_cast<const __bf16*>(x), reinterpret_cast<__bf16*>(y), M, n_img, H, W, wp, wp, wp, nullptr, nullptr, nullptr, st);
    if constexpr (C < 40) {
      if (shape_ok && dtype == LY_F32)
        return launch_mlp_persist<float, C, 2, HT, 2>(reinterpret_cast<const float*>(x), reinterpret_cast<float*>(y), M, n_img, H, W, wp, wp, wp, nullptr, nullptr, nullptr, st);
    }
  }
  // any other map: the one-shot kernel on flattened 64-pixel runs, stopped after the partial conv
  if (dtype == LY_BF16) return launch_mlp_pconv1<__bf16, C, 1, HT, false>(reinterpret_cast<const __bf16*>(x), reinterpret_cast<__bf16*>(y), M, n_img, H, W, wp, st);
  return launch_mlp_pconv1<float, C, 1, HT, false>(reinterpret_cast<const float*>(x), reinterpret_cast<float*>(y), M, n_img, H, W, wp, st);
}

template <int C, int HT, int NTMAX>
static int dispatch_nt(LY_MLP_ARGS) {
  if (dtype == LY_BF16)
    return dispatch_nt_t<__bf16, C, HT, NTMAX>(reinterpret_cast<const __bf16*>(x), reinterpret_cast<__bf16*>(y), M, n_img, H, W, wp, w1, w2, s, b, stats, st);
  return dispatch_nt_t<float, C, HT, NTMAX>(reinterpret_cast<const float*>(x), reinterpret_cast<float*>(y), M, n_img, H, W, wp, w1, w2, s, b, stats, st);
}
