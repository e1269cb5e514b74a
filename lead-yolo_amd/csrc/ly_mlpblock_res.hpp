#pragma once
// Fused FasterNet MLPBlock forward, RESIDENT-WEIGHTS form for the mid-width stages (C = 80; bf16 storage), gfx950, round 6.
// Reference: Partial_conv3.forward_split_cat + MLPBlock.forward (models/common.py:1432-1437, 1478-1482).
//
// Why: the one-shot kernels (ly_mlpblock.hpp) give every wave its own copy of the fragment stream: at C = 80 a 256-pixel block pulls
// 4 x 67 KiB of packed weights from L2 for 40 KiB of pixels (bs = 64, 40 x 40: 400 blocks x 268 KiB = 107 MB over the L2 -> CU path against
// 16 MB of activations; 19-25 us per block of the stage for 2.8 us of MFMA issue).  Here ONE copy of all fragments (67 KiB) sits in LDS for
// the whole launch and 8 waves (two per SIMD: one wave's LDS round trips run under the other's MFMAs) read it — the L2 -> CU weight traffic
// of a launch is 256 x 67 KiB.  A block walks a contiguous range of the flattened N*H*W index in runs of 256 pixels (32 per wave: every
// fragment read from LDS feeds two MFMAs); the raw pixels of the next run are requested before the current one is computed.
// The arithmetic per pixel is ly_mlpblock_body's, operand for operand: results are bit-identical to the one-shot kernels.
#include "ly_mlpblock.hpp"

#define LY_RES_WAVES 8
#define LY_RES_THREADS (64 * LY_RES_WAVES)
#define LY_RES_NVH 4              // halo staging items (4 channels of one halo pixel) per thread: 2048 >= (256 + 2 W + 2) * G  ->  W <= 75 at C = 80

template <typename T, int C, int NT, int HT, bool STATS, int D>
__global__ __launch_bounds__(LY_RES_THREADS) void ly_mlpblock_res_kernel(
    const T* __restrict__ x, T* __restrict__ y, const long M, const int H, const int W, const int per_block,
    const uint4* __restrict__ wp, const uint4* __restrict__ w1, const uint4* __restrict__ w2,
    const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, double* __restrict__ stats) {
  using Gm = MlpGeom<C>;
  using TR = LyT<T>;
  using RV = typename TR::RV;
  using R4 = typename TR::R4;
  constexpr int PL = TR::PL, VW = TR::VW;
  static_assert(PL == 1, "resident-weights MLPBlock: bf16 storage only (fp32 needs two weight planes: 134 KiB at C = 80)");
  constexpr int CQ = Gm::CQ, G = Gm::G, SP = Gm::SP, PT = Gm::PT, C16 = Gm::C16, KP = Gm::KP, S1 = Gm::S1;
  constexpr int RS = Gm::RS, RSP = Gm::RSP, HTP = Gm::HTP, S2 = Gm::S2;
  constexpr int BP = LY_RES_WAVES * 16 * NT;
  constexpr int NFP = PT * SP, NF1 = HTP * S1, NF2 = C16 * S2, NFW = NFP + NF1 + NF2;
  constexpr int NWV = (NFW * 64 + LY_RES_THREADS - 1) / LY_RES_THREADS;        // 16-byte weight items per thread
  constexpr int WBYTES = NWV * LY_RES_THREADS * 16;
  constexpr int CV = C / VW;                                  // real 16-byte vectors per pixel row (the K padding up to KP stays zero)
  static_assert(HTP % HT == 0 && HT % 2 == 0 && C % VW == 0, "geometry");
  extern __shared__ f32x4 ly_smem4[];
  char* const wl = reinterpret_cast<char*>(ly_smem4);
  char* const xs_hi = wl + WBYTES;
  char* const ps_hi = xs_hi + BP * RS;
  const int BPH = BP + 2 * W + 2;
  float* const sacc = reinterpret_cast<float*>(ps_hi + (BPH * RSP + 15) / 16 * 16);       // STATS: [waves][2][HTP*16]; else BatchNorm scale | shift [2][HTP*16]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const f32x4 zero = ly_zero4();

  const long pb = (long)ly_xcd_remap((int)blockIdx.x, (int)gridDim.x) * per_block;          // this block's pixels: [pb, pe)
  const long pe = pb + per_block < M ? pb + per_block : M;
  if (pb >= M) return;

  // ---- staging plan: the run's pixels (tv: one 16-byte vector each) and its halo of the first C/4 channels (hv: 4 channels each) ----
  constexpr int TVN = BP * CV, NVT = (TVN + LY_RES_THREADS - 1) / LY_RES_THREADS;
  constexpr int NVH = LY_RES_NVH;
  const int HVN = BPH * G;
  RV tv[NVT];
  R4 hv[NVH];
  // the run-independent part of the plan: pixel / halo-pixel index, element offset from the run's first pixel (halo: from the halo's first
  // pixel), LDS byte offset; -1: no item.  Per run only the bounds checks and one scalar base remain
  int tpix[NVT], toff[NVT], tdst[NVT], hpx[NVH], hoff[NVH], hdst[NVH];
#pragma unroll
  for (int e = 0; e < NVT; ++e) {
    const int idx = tid + e * LY_RES_THREADS;
    const int pix = idx / CV, c4 = idx - pix * CV;
    tpix[e] = idx < TVN ? pix : (1 << 30);
    toff[e] = pix * C + c4 * VW;
    tdst[e] = pix * RS + 2 * VW * c4;
  }
#pragma unroll
  for (int e = 0; e < NVH; ++e) {
    const int idx = tid + e * LY_RES_THREADS;
    const int hp = idx / G, c4 = idx - hp * G;
    hpx[e] = idx < HVN ? hp : (1 << 30);
    hoff[e] = hp * C + c4 * 4;
    hdst[e] = hp * RSP + 8 * c4;
  }
  const int halo0 = W + 1;                                   // halo pixel hp is global pixel p0 - halo0 + hp
  auto issue = [&](const long p0) {
    const T* const xb = x + p0 * C;                          // p0 < M: a valid address for the clamped loads
    const long lim = M - p0;
#pragma unroll
    for (int e = 0; e < NVT; ++e) tv[e] = ly_ldrv<T>(xb + (tpix[e] < lim ? toff[e] : 0));
    const long lo = halo0 - p0, hi = lim + halo0;            // valid halo pixels: lo <= hp < hi
    const T* const xh = xb - (long)halo0 * C;
#pragma unroll
    for (int e = 0; e < NVH; ++e) hv[e] = ly_ldr4<T>((hpx[e] >= lo && hpx[e] < hi) ? xh + hoff[e] : xb);
  };
  auto commit = [&](const long p0) {
    const long lim = M - p0;
#pragma unroll
    for (int e = 0; e < NVT; ++e) {
      RV v = tv[e];
      if (tpix[e] >= lim) ly_zero_raw(v);
      if (tpix[e] < BP) *reinterpret_cast<RV*>(xs_hi + tdst[e]) = v;
    }
    const long lo = halo0 - p0, hi = lim + halo0;
#pragma unroll
    for (int e = 0; e < NVH; ++e) {
      R4 v = hv[e];
      if (hpx[e] < lo || hpx[e] >= hi) ly_zero_raw(v);
      if (hpx[e] < BPH) *reinterpret_cast<R4*>(ps_hi + hdst[e]) = v;
    }
  };
  issue(pb);

  // ---- weights -> LDS, once; the K padding of the pixel rows (channels C .. KP-1) -> zero, once --------------------------------
  // (every load of a thread is issued before its first store: one round trip for the whole copy, not one per 8 KiB — as `for (i = tid; ...)
  // wl[i] = w[i]` loops the copy was nine dependent round trips in the prologue of every block)
  {
    constexpr int NW4 = NFW * 64;
    uint4 wv[NWV];
#pragma unroll
    for (int e = 0; e < NWV; ++e) {
      const int i = tid + e * LY_RES_THREADS;
      const uint4* src = i < NFP * 64 ? wp + i : (i < (NFP + NF1) * 64 ? w1 + (i - NFP * 64) : w2 + (i - (NFP + NF1) * 64));
      wv[e] = *((i < (STATS ? (NFP + NF1) * 64 : NW4)) ? src : wp);
    }
#pragma unroll
    for (int e = 0; e < NWV; ++e) reinterpret_cast<uint4*>(wl)[tid + e * LY_RES_THREADS] = wv[e];       // (the image is padded to NWV * 512 entries)
  }
  if constexpr (KP > C) {
    constexpr int PV = (KP - C) / VW;
    for (int i = tid; i < BP * PV; i += LY_RES_THREADS) {
      const int pix = i / PV, c4 = CV + (i - pix * PV);
      *reinterpret_cast<ly_u32x4*>(xs_hi + pix * RS + 2 * VW * c4) = (ly_u32x4){0u, 0u, 0u, 0u};
    }
  }
  if constexpr (STATS) {
    for (int i = tid; i < LY_RES_WAVES * 2 * HTP * 16; i += LY_RES_THREADS) sacc[i] = 0.f;
  } else {
    // the hidden BatchNorm's folded scale / shift: read per hidden tile from LDS (as loop-invariant global loads hipcc keeps all 2 x HTP x 4 of
    // them in registers across the walk: 80 registers at C = 80, 58 spilled at two waves per SIMD)
    for (int i = tid; i < 2 * HTP * 16; i += LY_RES_THREADS) sacc[i] = i < HTP * 16 ? bn_scale[i] : bn_shift[i - HTP * 16];
  }
  const char* const wlane = wl + lane * 16;
  auto wlds = [&](const int fi) -> LyWF<PL> {
    LyWF<PL> f;
    f.hi = *reinterpret_cast<const bf16x8*>(wlane + fi * 1024);
    return f;
  };
  constexpr int FP = SP * PT, F1 = S1 * HT, F2 = STATS ? 0 : (HT / 2) * C16, FQ = F1 + F2, NFRAG = FP + (HTP / HT) * FQ;
  auto wseq = [&](int g) -> LyWF<PL> {
    if (g < FP) return wlds((g % PT) * SP + g / PT);
    g -= FP;
    const int chunk = g / FQ, r = g - chunk * FQ;
    if (r < F1) return wlds(NFP + (chunk * HT + r % HT) * S1 + r / HT);
    const int r2 = r - F1;
    return wlds(NFP + NF1 + (r2 % C16) * S2 + chunk * (HT / 2) + r2 / C16);
  };

  const int pixbase = wave * (16 * NT);
  const bf16x4 z4 = __builtin_bit_cast(bf16x4, make_uint2(0u, 0u));

  commit(pb);
  __syncthreads();                                         // weights, coefficients and the first run are in LDS
  for (long p0 = pb; p0 < pe; p0 += BP) {
    auto pvalid = [&](const int pix) -> bool { return p0 + pix < pe; };
    // Every LDS operand of the run is requested HERE, ahead of the first MFMA (72 registers): hipcc otherwise issues each k-step's reads
    // right before their MFMAs — the wave waits out an LDS round trip per k-step, and because lgkmcnt counts in order that wait also drains
    // the fragment ring (measured: 50 % of the wave cycles in s_waitcnt, the MFMA pipe 21 % busy, ring or no ring the same 18-19 us).
    //  * xf: the pixel rows as B operands of the expand contraction, all S1 k-steps — read ONCE per run, not once per hidden chunk;
    //  * pxo: the partial conv's operands (halo image), all SP k-steps, masked after they arrive.
    bf16x8 xf[S1][NT];
#pragma unroll
    for (int s_ = 0; s_ < S1; ++s_)
#pragma unroll
      for (int n = 0; n < NT; ++n) xf[s_][n] = ly_lds_frag(xs_hi, (pixbase + 16 * n + li) * RS, s_, lq);
    uint32_t tmask[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const long gp = p0 + pixbase + 16 * n + li;
      int h_, w_;
      ly_pix_hw(gp, H, W, h_, w_);
      tmask[n] = ly_tapmask(h_, w_, H, W, gp < M);
    }
    bf16x4 pxo[SP][NT][2];
#pragma unroll
    for (int s_ = 0; s_ < SP; ++s_)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int gg = 8 * s_ + 4 * h + lq;
        const bool gv = gg < 9 * G;
        const int tap = gv ? gg / G : 0;
        const int cq4 = gv ? gg - tap * G : 0;
        const int ty = tap / 3, tx = tap - 3 * ty;
        const int off = (ty * W + tx) * RSP + 8 * cq4;
#pragma unroll
        for (int n = 0; n < NT; ++n) pxo[s_][n][h] = *reinterpret_cast<const bf16x4*>(ps_hi + (pixbase + 16 * n + li) * RSP + off);
      }
    // the run's operands are in registers: the tile and the halo image are free for the next run, whose pixels are requested now and
    // committed behind this run's MFMAs — BEFORE this run's output stores are issued: s_waitcnt vmcnt counts in order, and hipcc sizes the
    // wait of a load that crosses the loop edge as if nothing younger were pending, so a commit at the top of the next trip also waited for
    // the previous trip's ten output stores to be acknowledged (~2 us per run)
    const bool more = p0 + BP < pe;
    __syncthreads();
    issue(more ? p0 + BP : p0);                            // (the last run re-requests itself: straight-line loads)
    // the fragment ring: D fragments of the LDS stream in registers ahead of their MFMAs
    LyWF<PL> ring[D > 0 ? D : 1];
#pragma unroll
    for (int q = 0; q < D; ++q)
      if (q < NFRAG) ring[q] = wseq(q);
    __builtin_amdgcn_sched_barrier(0x476);
    int g = 0;
    auto wnext = [&]() -> LyWF<PL> {
      if constexpr (D == 0) return wseq(g);
      else return ring[g % D];
    };
    auto wrefill = [&]() {
      if constexpr (D > 0) {
        if (g + D < NFRAG) ring[g % D] = wseq(g + D);
        __builtin_amdgcn_sched_barrier(0x476);             // LDS reads and MFMAs stay in program order: the refills stay D fragments ahead
      }
      ++g;
    };

    // ---- 1. partial 3x3 conv -------------------------------------------------------------------
    bf16x8 x0[NT];                                         // k-step 0 of the pixel rows as loaded (the residual's channels 0 .. 31)
    {
      f32x4 accp[PT][NT];
#pragma unroll
      for (int t = 0; t < PT; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) accp[t][n] = zero;
#pragma unroll
      for (int s_ = 0; s_ < SP; ++s_) {
        bf16x8 xh[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          bf16x4 ph[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int gg = 8 * s_ + 4 * h + lq;
            const int tap = gg < 9 * G ? gg / G : 0;
            const bool ok = gg < 9 * G && ((tmask[n] >> tap) & 1u);
            ph[h] = ok ? pxo[s_][n][h] : z4;
          }
          xh[n] = ly_cat8(ph[0], ph[1]);
        }
#pragma unroll
        for (int t = 0; t < PT; ++t) {
          const LyWF<PL> wf = wnext();
#pragma unroll
          for (int n = 0; n < NT; ++n) accp[t][n] = ly_mfmap<PL>(wf, xh[n], xh[n], accp[t][n]);
          wrefill();
        }
      }
      // z = [pconv | x[C/4:]] joins the first k-step of the expand contraction IN REGISTERS: the lane's accumulator rows 16 t + 4 lq + r are
      // its operand elements k = 16 t + 4 lq + r of k-step 0 (ly_tile.hpp), so nothing is written back to the tile
      static_assert(CQ % 4 == 0 && PT <= 2 && CQ <= 32, "partial conv output must sit inside k-step 0 in whole 4-channel groups");
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        x0[n] = xf[0][n];
        const bf16x4 o0 = __builtin_shufflevector(xf[0][n], xf[0][n], 0, 1, 2, 3), o1 = __builtin_shufflevector(xf[0][n], xf[0][n], 4, 5, 6, 7);
        const bf16x4 z0 = ly_cvtb4(accp[0][n]), z1 = ly_cvtb4(accp[PT - 1][n]);
        const bf16x4 a0 = 4 * lq < CQ ? z0 : o0;
        const bf16x4 a1 = (PT == 2 && 16 + 4 * lq < CQ) ? z1 : o1;
        xf[0][n] = ly_cat8(a0, a1);
      }
    }

    // ---- 2 + 3. expand -> BN -> ReLU -> project, hidden kept in registers ---------------------------
    f32x4 acco[C16][NT];
#pragma unroll
    for (int t = 0; t < C16; ++t)
#pragma unroll
      for (int n = 0; n < NT; ++n) acco[t][n] = zero;

#pragma unroll
    for (int hc = 0; hc < HTP / HT; ++hc) {
      // the chunk's BatchNorm coefficients: requested before its MFMAs (they are older than the ring reads in flight when they are used)
      f32x4 bsc[STATS ? 1 : HT], bsh[STATS ? 1 : HT];
      if constexpr (!STATS) {
#pragma unroll
        for (int t = 0; t < HT; ++t) {
          const int ch = (hc * HT + t) * 16 + 4 * lq;
          bsc[t] = *reinterpret_cast<const f32x4*>(sacc + ch);
          bsh[t] = *reinterpret_cast<const f32x4*>(sacc + HTP * 16 + ch);
        }
        __builtin_amdgcn_sched_barrier(0x476);
      }
      f32x4 acch[HT][NT];
#pragma unroll
      for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) acch[t][n] = zero;
#pragma unroll
      for (int s_ = 0; s_ < S1; ++s_) {
#pragma unroll
        for (int t = 0; t < HT; ++t) {
          const LyWF<PL> wf = wnext();
#pragma unroll
          for (int n = 0; n < NT; ++n) acch[t][n] = ly_mfmap<PL>(wf, xf[s_][n], xf[s_][n], acch[t][n]);
          wrefill();
        }
      }
      if constexpr (STATS) {
        // statistics pass of train-mode BatchNorm: sum / sum of squares of the pre-BN hidden activations over the block's valid pixels,
        // per wave in LDS (every address has one owner lane: plain read-add-write in program order), flushed once at the end of the walk
#pragma unroll
        for (int t = 0; t < HT; ++t) {
          f32x4 s1 = zero, s2 = zero;
#pragma unroll
          for (int n = 0; n < NT; ++n)
            if (pvalid(pixbase + 16 * n + li)) {
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
            float* const sw = sacc + wave * (2 * HTP * 16);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              sw[ch + r] += s1[r];
              sw[HTP * 16 + ch + r] += s2[r];
            }
          }
        }
        continue;
      }
      bf16x4 hh[HT][NT];
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        const f32x4 sc = bsc[t], sh = bsh[t];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(acch[t][n][r] * sc[r] + sh[r], 0.f);
          hh[t][n] = ly_cvtb4(v);
        }
      }
#pragma unroll
      for (int u = 0; u < HT / 2; ++u) {
        bf16x8 xh[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) xh[n] = ly_cat8(hh[2 * u][n], hh[2 * u + 1][n]);
#pragma unroll
        for (int ct = 0; ct < C16; ++ct) {
          const LyWF<PL> wf = wnext();
#pragma unroll
          for (int n = 0; n < NT; ++n) acco[ct][n] = ly_mfmap<PL>(wf, xh[n], xh[n], acco[ct][n]);
          wrefill();
        }
      }
    }

    // the next run into the tile / halo image (every wave passed the barrier above with its operands in registers)
    if (more) commit(p0 + BP);
    if constexpr (!STATS) {
      // ---- epilogue: residual + store.  The residual x is the operand image the lane already holds: output tile ct = channels
      // 16 ct + 4 lq + r = half (ct & 1) of k-step ct / 2 (bf16 storage: exact)
#pragma unroll
      for (int ct = 0; ct < C16; ++ct)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int c = 16 * ct + 4 * lq;
          const int pix = pixbase + 16 * n + li;
          if (c < C && pvalid(pix)) {
            const bf16x8 src = ct < 2 ? x0[n] : xf[ct / 2][n];
            const bf16x4 rh = (ct & 1) ? __builtin_shufflevector(src, src, 4, 5, 6, 7) : __builtin_shufflevector(src, src, 0, 1, 2, 3);
            ly_st4<T>(y + (p0 + pix) * C + c, acco[ct][n] + ly_cvt4(rh));
          }
        }
    }
    __syncthreads();                                       // the next run is visible
  }
  if constexpr (STATS) {
    __syncthreads();
    double* const st = stats + (size_t)(blockIdx.x & (LY_STATS_STRIPES - 1)) * 2 * (HTP * 16);
    constexpr int SL = 2 * HTP * 16;
    for (int i = tid; i < SL; i += LY_RES_THREADS) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < LY_RES_WAVES; ++w) s += sacc[w * SL + i];
      atomicAdd(st + i, (double)s);
    }
  }
}

// launch: returns 1 when the shape is outside what the kernel is built for (the caller then takes the one-shot kernels)
template <typename T, int C, int NT, int HT, int D>
static int launch_mlp_res(const T* x, T* y, long M, int H, int W, const void* wp, const void* w1, const void* w2,
                          const float* s, const float* b, double* stats, hipStream_t st) {
  using Gm = MlpGeom<C>;
  constexpr int BP = LY_RES_WAVES * 16 * NT;
  constexpr int NFW = Gm::PT * Gm::SP + Gm::HTP * Gm::S1 + Gm::C16 * Gm::S2;
  const long BPH = BP + 2L * W + 2;
  if (BPH * Gm::G > (long)LY_RES_NVH * LY_RES_THREADS) return 1;
  const size_t lds = (size_t)((NFW * 64 + LY_RES_THREADS - 1) / LY_RES_THREADS) * LY_RES_THREADS * 16 + (size_t)BP * Gm::RS + ((size_t)BPH * Gm::RSP + 15) / 16 * 16 +
                     (stats ? (size_t)LY_RES_WAVES : (size_t)1) * 2 * Gm::HTP * 16 * sizeof(float);
  if (lds > 160 * 1024) return 1;
  // one block per CU; every block the same number of pixels (a multiple of 16), as few runs as that allows
  long blocks = (M + BP - 1) / BP;
  if (blocks > 256) blocks = 256;
  long per_block = ((M + blocks - 1) / blocks + 15) / 16 * 16;
  blocks = (M + per_block - 1) / per_block;
  void (*k)(const T*, T*, long, int, int, int, const uint4*, const uint4*, const uint4*, const float*, const float*, double*);
  if (stats) k = ly_mlpblock_res_kernel<T, C, NT, HT, true, D>;
  else k = ly_mlpblock_res_kernel<T, C, NT, HT, false, D>;
  static bool configured[2] = {false, false};
  if (!configured[stats ? 1 : 0]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    LY_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    configured[stats ? 1 : 0] = true;
  }
  hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(LY_RES_THREADS), lds, st, x, y, M, H, W, (int)per_block, reinterpret_cast<const uint4*>(wp),
                     reinterpret_cast<const uint4*>(w1), reinterpret_cast<const uint4*>(w2), s, b, stats);
  LY_LAUNCH_CHECK();
  return 0;
}
