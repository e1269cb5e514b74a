// Pointwise-convolution GEMM with fused gather/prologue and epilogue, fp32 I/O, bf16x3 math, gfx950.
//
//   out[m, n] = act( rowscale[m] * scale[n] * sum_k A'[m, k] * W[n, k] + shift[n] )
//
// m runs over the flattened NHWC pixels, so A' is the activation matrix itself (no im2col).  A' is
// assembled while the tile is staged into LDS, which is where the reference's separate passes go:
//   * two row sources (a0 | a1)            -> torch.cat of two feature maps never materialised
//   * a0 at half resolution (up0)          -> nn.Upsample(2, 'nearest') folded into the load
//   * gate: a0 * a_w[n,w,:] * a_h[n,h,:]   -> CoordAtt's `identity * a_w * a_h` (models/common.py:1608)
//   * affine+relu+ca: relu(x*s+b)*ca[n,:]  -> RFCBAMConv k=1 generate/BN/ReLU and SE scaling
//                                             (models/rfa.py:101-106,124)
//   * patch gather (k x k stride k)        -> PatchEmbed/PatchMerging_FasterNet (models/common.py:1528-1561)
// and the epilogue covers BN (folded scale/shift), conv bias, the per-pixel receptive-field weight
// (rowscale) and ReLU / SiLU.  `out` may point into a wider buffer (ldo, pre-offset pointer) so a
// producer can write straight into its slot of a later concat.
//
// Block = 4 waves; WC waves split the output channels (distinct weight fragments per wave: no
// redundant weight traffic), 4/WC waves split the pixels.  Wave tile = NT pixel tiles x MT channel
// tiles of 16x16 (bf16x3 on v_mfma_f32_16x16x32_bf16, see ly_tile.cuh).
//
// Persistent pipeline: a block owns a strided set of pixel tiles and walks the work items
// (tile, K-chunk of 64).  The raw fp32 values of item i+1 are loaded into registers BEFORE the MFMAs
// of item i are issued and are transformed/split/written to the other LDS buffer after them (one
// barrier per item), so global loads are in flight during every contraction, also across tile
// boundaries; weight fragments are fetched one k-step ahead.
#include "ly_tile.cuh"
#include "ly_params.h"

// K chunk per pipeline stage: 64; a 128-wide stage (twice the bytes in flight per thread) is kept as a tuning
// variant (ly_debug_set_gemm_bk) — on MI355X it needs 272 registers, drops to one wave per SIMD and measures
// 15-30 % slower on every LEAD-YOLO shape.  LDS row = 2*BK + 16 bytes per plane

#ifndef LY_GEMM_MINW
#define LY_GEMM_MINW 1
#endif
template <int NT, int MT, int WC, int GATHER, int PRO, bool DBG, int LY_BK>
__device__ __forceinline__ void ly_gemm_body(const LyGemmParams& P, const int gy, const int nslots, const int gx, const int dbg_arg) {
  const int dbg = DBG ? dbg_arg : 0;                      // production instantiation: no ablation branches inside the loop
  constexpr int WP = 4 / WC;
  constexpr int BP = 16 * NT * WP;
  constexpr int LY_RSX = 2 * LY_BK + 16;
  constexpr int KQ = LY_BK / 4;                          // float4 columns per chunk
  constexpr int RSTEP = LY_THREADS / KQ;                 // pixel rows covered by one pass of the block
  constexpr int SPC = LY_BK / 32;                        // k-steps per chunk
  constexpr int NV = BP * (LY_BK / 4) / LY_THREADS;      // float4 per thread per chunk; thread's pixels: tid/KQ + RSTEP*e
  static_assert(NV >= 1 && BP * (LY_BK / 4) % LY_THREADS == 0, "tile must divide evenly over the block");
  constexpr int PLANE = BP * LY_RSX;
  extern __shared__ f32x4 ly_smem4[];                     // [buf][plane][BP][RSX]
  char* xs = reinterpret_cast<char*>(ly_smem4);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  const int wc = wave % WC, wp_ = wave / WC;
  const int lid = ly_xcd_remap(blockIdx.x, gy * nslots);
  const int by = lid % gy;
  const int slot = lid / gy;
  const int HW = P.H * P.W;
  const float invHW = 1.f / (float)HW, invW = 1.f / (float)P.W;
  const f32x4 zero = ly_zero4();
  const int S = (P.K + 31) >> 5;
  const int T = (P.N + 15) >> 4;
  const int nchunk = (P.K + LY_BK - 1) / LY_BK;
  constexpr bool need_nhw = GATHER != LY_GATHER_ROWS || PRO != LY_PRO_NONE;
  const int k4 = tid % KQ;                                // this thread's float4 column inside a chunk
  const int prow = tid / KQ;                              // its first pixel row; others at +RSTEP*e

  // per-thread description of the NV pixels it stages for the tile being prefetched
  long t_row0[NV];
  int t_n[NV], t_hw[NV];
  auto setup = [&](long p0) {
#pragma unroll
    for (int e = 0; e < NV; ++e) {
      const long gp = p0 + prow + RSTEP * e;
      int n = -1, h = 0, w = 0;
      long row0 = gp;
      if (gp < P.M) {
        n = 0;
        if (need_nhw) {
          n = ly_fdiv((int)gp, HW, invHW);
          const int rem = (int)gp - n * HW;
          h = ly_fdiv(rem, P.W, invW);
          w = rem - h * P.W;
          if (GATHER == LY_GATHER_UP2)
            row0 = ((long)n * (P.H >> 1) + (h >> 1)) * (P.W >> 1) + (w >> 1);
          else if (GATHER == LY_GATHER_PATCH)
            row0 = (((long)n * P.Hin + (long)h * P.ks) * P.Win + (long)w * P.ks);       // first input pixel of the patch
          else if (GATHER == LY_GATHER_PATCH_NCHW)
            row0 = ((long)n * P.Cin * P.Hin + (long)h * P.ks) * P.Win + (long)w * P.ks; // element offset of (n, 0, ks*h, ks*w)
        }
      }
      t_row0[e] = row0; t_n[e] = n; t_hw[e] = (h << 16) | w;
    }
  };

  f32x4 pv[NV];
  f32x4 pgw[PRO == LY_PRO_GATE ? NV : 1], pgh[PRO == LY_PRO_GATE ? NV : 1];
  auto prefetch = [&](long p0, int kc) {
    // branch-free: every lane issues its NV loads back to back (clamped address), zeros are selected afterwards
    const int kk = kc + 4 * k4;
    const bool kok = kk < P.K;
    long koff;                                             // element offset contributed by the k position
    const float* src = P.a0;
    long rowmul = P.lda0;
    bool second = false;
    if (GATHER == LY_GATHER_PATCH) {
      const int seg = kk / P.pk, within = kk - seg * P.pk;
      koff = (long)seg * P.Win * P.lda0 + within;
    } else if (GATHER == LY_GATHER_PATCH_NCHW) {
      const int c = kk >> 4, ky = (kk >> 2) & 3;           // ks == 4: one float4 = one (c, ky) input row segment
      koff = ((long)c * P.Hin + ky) * P.Win;
      rowmul = 1;
    } else {
      second = kk >= P.k0;
      koff = second ? kk - P.k0 : kk;
      if (second) { src = P.a1; rowmul = P.lda1; }
    }
#pragma unroll
    for (int e = 0; e < NV; ++e) {
      const bool ok = kok && t_n[e] >= 0;
      const long row = second ? (p0 + prow + RSTEP * e) : t_row0[e];
      const float* ptr = ok ? src + row * rowmul + koff : P.a0;
      pv[e] = ly_ldg4(ptr);
    }
    // NOTE: out-of-range lanes are zeroed in commit(), NOT here: touching pv[] now would force an
    // s_waitcnt on the loads just issued and serialise them with the MFMAs they are meant to overlap.
    if (PRO == LY_PRO_GATE) {          // CoordAtt factors of the same items travel with them (L2-resident tables)
      const bool gk = kk < P.k0;
#pragma unroll
      for (int e = 0; e < NV; ++e) {
        const bool ok = gk && t_n[e] >= 0;
        const int n = ok ? t_n[e] : 0, h = ok ? (t_hw[e] >> 16) : 0, w = ok ? (t_hw[e] & 0xffff) : 0;
        const int kq = ok ? kk : 0;
        pgw[e] = ly_ldg4(P.g_w + ((long)n * P.W + w) * P.k0 + kq);
        pgh[e] = ly_ldg4(P.g_h + ((long)n * P.H + h) * P.k0 + kq);
      }
    }
  };
  auto commit = [&](long p0, int kc, int buf) {
    char* hi = xs + buf * 2 * PLANE;
    char* lo = hi + PLANE;
    const int kk = kc + 4 * k4;
#pragma unroll
    for (int e = 0; e < NV; ++e)
      if (!(kk < P.K && t_n[e] >= 0)) pv[e] = zero;
    if (PRO == LY_PRO_GATE) {
      const bool kok = kk < P.k0;
      f32x4 rr[NV];
#pragma unroll
      for (int e = 0; e < NV; ++e) {
        const bool ok = kok && t_n[e] >= 0;
        rr[e] = (P.res && ok) ? ly_ldg4(P.res + (p0 + prow + RSTEP * e) * P.ldres + kk) : zero;
      }
#pragma unroll
      for (int e = 0; e < NV; ++e)
        if (kok && t_n[e] >= 0) pv[e] = pv[e] * pgw[e] * pgh[e] + rr[e];
    } else if (PRO == LY_PRO_AFFINE_RELU_CA) {
      const bool kok = kk < P.K;
      const int kq = kok ? kk : 0;
      const f32x4 sa = ly_ldg4(P.p_scale + kq), sb = ly_ldg4(P.p_shift + kq);
      f32x4 ca[NV];
#pragma unroll
      for (int e = 0; e < NV; ++e) ca[e] = ly_ldg4(P.p_ca + (long)(t_n[e] >= 0 ? t_n[e] : 0) * P.K + kq);
#pragma unroll
      for (int e = 0; e < NV; ++e)
        if (kok && t_n[e] >= 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) pv[e][r] = fmaxf(pv[e][r] * sa[r] + sb[r], 0.f) * ca[e][r];
        }
    }
#pragma unroll
    for (int e = 0; e < NV; ++e) ly_lds_put4(hi, lo, (prow + RSTEP * e) * LY_RSX, 4 * k4, pv[e]);
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[t][n] = zero;

  long wbase[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    int tt = (by * WC + wc) * MT + t;
    wbase[t] = (long)(tt < T ? tt : T - 1) * S;
  }
  const uint4* wpk = reinterpret_cast<const uint4*>(P.wp);
  const int pixgrp = wp_ * (16 * NT);
  const bool vec_ok = (P.ldo & 3) == 0;
  const int act = P.act;
  float* const stats = P.stats;                           // non-NULL: batch-statistics pass (no store)
  float rsv[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) rsv[n] = 1.f;
  float esc[MT][4], esh[MT][4];                            // epilogue scale/shift: fetched once, not per tile
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int c = 16 * ((by * WC + wc) * MT + t) + 4 * lq;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = c + r < P.N;
      esc[t][r] = (ok && P.e_scale) ? P.e_scale[c + r] : 1.f;
      esh[t][r] = (ok && P.e_shift) ? P.e_shift[c + r] : 0.f;
    }
  }

  if (slot >= gx) return;                                  // (host never launches such blocks)
  ly_l2_warm(P.wp, (long)T * S * 2048, P.stats ? P.stats : P.out);
  LyWFrag wcur[MT], wnxt[SPC - 1][MT];                   // wnxt[j]: weights of step j+1 of the chunk; slot 0 is reused for the
                                                          // first step of the NEXT item once step 0 has consumed it
#pragma unroll
  for (int t = 0; t < MT; ++t) wcur[t] = ly_wfrag(wpk, wbase[t], lane);

  // work items = (pixel tile, K chunk); item i+1 is always in flight while item i is contracted
  long p0 = (long)slot * BP;                               // tile being computed
  long pn = p0;                                            // tile being prefetched
  int cn = 0;                                              // chunk being prefetched
  setup(pn);
  prefetch(pn, 0);
  commit(pn, 0, 0);
  __syncthreads();
  int buf = 0;
  int pt = slot;
  while (true) {
    for (int c = 0; c < nchunk; ++c) {
      // advance the prefetch cursor to the item after (pt, c)
      bool more = true;
      if (cn + 1 < nchunk) {
        ++cn;
      } else if (pt + nslots < gx && c == nchunk - 1) {
        cn = 0;
        pn = (long)(pt + nslots) * BP;
        setup(pn);
      } else if (c == nchunk - 1) {
        more = false;                                      // last item: re-stage chunk 0 of the same tile (harmless), so that the
        cn = 0;                                            // loop body is STRAIGHT-LINE: the compiler then counts outstanding loads
      }                                                    // exactly and the prefetch really overlaps the contraction
      // weights of the chunk's later k-steps first (older in the in-order vmcnt queue than the activation prefetch)
      if (!(dbg & 16)) {
#pragma unroll
        for (int j = 1; j < SPC; ++j) {
          const int gj = SPC * c + j < S ? SPC * c + j : 0;
#pragma unroll
          for (int t = 0; t < MT; ++t) wnxt[j - 1][t] = ly_wfrag(wpk, wbase[t] + gj, lane);
        }
      }
      if (PRO == LY_PRO_AFFINE_RELU_CA) {
        // per-pixel row scale of the tile being finished: issued BEFORE the next item's prefetch so the
        // epilogue's wait for it does not also wait for that prefetch (vmcnt retires in order)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const long gp = p0 + pixgrp + 16 * n + li;
          rsv[n] = P.rowscale[gp < P.M ? gp : 0];
        }
      }
      if (!(dbg & 4)) prefetch(pn, cn * LY_BK);
      const char* hi = xs + buf * 2 * PLANE;
      const char* lo = hi + PLANE;
#pragma unroll
      for (int s = 0; s < LY_BK / 32; ++s) {
        if (!(dbg & 2)) {
          if (s == SPC - 1) {                              // weights of the next item's first step (absent steps of a ragged
            const int gn = SPC * (c + 1) < S ? SPC * (c + 1) : 0;   // last chunk contract LDS zeros with clamped weights: no branch)
#pragma unroll
            for (int t = 0; t < MT; ++t) wnxt[0][t] = ly_wfrag(wpk, wbase[t] + gn, lane);
          }
          bf16x8 xh[NT], xl[NT];
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const int rb = (pixgrp + 16 * n + li) * LY_RSX;
            xh[n] = ly_lds_frag(hi, rb, s, lq);
            xl[n] = ly_lds_frag(lo, rb, s, lq);
          }
#pragma unroll
          for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[t][n] = ly_mfma3(wcur[t].hi, wcur[t].lo, xh[n], xl[n], acc[t][n]);
#pragma unroll
          for (int t = 0; t < MT; ++t) wcur[t] = wnxt[s < SPC - 1 ? s : 0][t];
        }
      }
      if (!(dbg & 1)) commit(pn, cn * LY_BK, buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
    // ---- epilogue of tile pt -----------------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int tt = (by * WC + wc) * MT + t;
      const int c = 16 * tt + 4 * lq;
      if (tt < T && c < P.N && !(dbg & 32)) {
        f32x4 st1 = zero, st2 = zero;              // statistics pass only: live just inside the epilogue (no register cost in the main loop)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const long gp = p0 + pixgrp + 16 * n + li;
          if (gp < P.M) {
            const float rs = PRO == LY_PRO_AFFINE_RELU_CA ? rsv[n] : 1.f;
            f32x4 u;
#pragma unroll
            for (int r = 0; r < 4; ++r) u[r] = acc[t][n][r] * rs * esc[t][r] + esh[t][r];
            if (stats) {                                   // pre-activation value is what BatchNorm normalises
              st1 += u;
              st2 += u * u;
              if (!P.out) continue;                        // pure statistics pass; with `out` the value is stored as well
            }
            const f32x4 v = ly_act4(u, act);
            float* o = P.out + gp * P.ldo + c;
            if (dbg & 8) {
            } else if (vec_ok && c + 3 < P.N) {
              ly_stg4(o, v);
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (c + r < P.N) o[r] = v[r];
            }
          }
        }
        if (stats) ly_stats_flush(stats, P.N, c, st1, st2);
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[t][n] = zero;
    }
    pt += nslots;
    if (pt >= gx) break;
    p0 = (long)pt * BP;
  }
}

// -------------------------------------------------------------------------------------------------
// TWO-DEEP variant of the pipeline above.  A K chunk contracts in ~770 MFMA cycles but its loads need 1-2 us to
// arrive, so with one chunk in flight per block every chunk waits for memory (measured: 12.2 us per 64 px x 128 ch x
// K=256 item, 1.3 us of it matrix work; chip-wide ~3.7 TB/s = bytes in flight / latency, independent of tile shape,
// occupancy and weight traffic — tools/gemm_sweep.py, tools/gemm_ablate*.py).  Here the raw values of items i+1 AND
// i+2 are in flight while item i is contracted: two register sets, used alternately (the item loop is unrolled by
// two so the set index is static); item i+2 is issued into the set that item i vacated when it was committed.
// -------------------------------------------------------------------------------------------------
template <int V>
struct LyIc { static constexpr int value = V; };

template <int NT, int MT, int WC, int GATHER, int PRO>
__device__ __forceinline__ void ly_gemm_body2(const LyGemmParams& P, const int gy, const int nslots, const int gx) {
  constexpr int LY_BK = 64;
  constexpr int WP = 4 / WC;
  constexpr int BP = 16 * NT * WP;
  constexpr int LY_RSX = 2 * LY_BK + 16;
  constexpr int KQ = LY_BK / 4;
  constexpr int RSTEP = LY_THREADS / KQ;
  constexpr int SPC = LY_BK / 32;
  constexpr int NV = BP * (LY_BK / 4) / LY_THREADS;
  constexpr int PLANE = BP * LY_RSX;
  extern __shared__ f32x4 ly_smem4[];
  char* xs = reinterpret_cast<char*>(ly_smem4);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  const int wc = wave % WC, wp_ = wave / WC;
  const int lid = ly_xcd_remap(blockIdx.x, gy * nslots);
  const int by = lid % gy;
  const int slot = lid / gy;
  const int HW = P.H * P.W;
  const float invHW = 1.f / (float)HW, invW = 1.f / (float)P.W;
  const f32x4 zero = ly_zero4();
  const int S = (P.K + 31) >> 5;
  const int T = (P.N + 15) >> 4;
  const int nchunk = (P.K + LY_BK - 1) / LY_BK;
  constexpr bool need_nhw = GATHER != LY_GATHER_ROWS || PRO != LY_PRO_NONE;
  // thread -> (float4 column k4, first pixel row prow).  NHWC sources: consecutive lanes take consecutive k (one pixel row is
  // K-contiguous).  NCHW image patches: consecutive k are different (channel, ky) planes, megabytes apart, while consecutive
  // output pixels of one plane row ARE contiguous (16 B each) — so there consecutive lanes take consecutive pixels.
  static_assert(KQ == RSTEP, "the NCHW lane mapping swaps the two 16-way indices");
  const int k4 = GATHER == LY_GATHER_PATCH_NCHW ? tid / RSTEP : tid % KQ;
  const int prow = GATHER == LY_GATHER_PATCH_NCHW ? tid % RSTEP : tid / KQ;
  if (slot >= gx) return;

  // ---- staging state, one copy per register set ---------------------------------------------------
  f32x4 pv[2][NV];
  long t_row0[2][NV];
  int t_n[2][NV], t_hw[2][NV];
  long s_p[2];                                             // tile start and K offset of the item held by the set
  int s_kc[2];
  int cur_pt = slot, cur_c = 0;                            // issue cursor (saturates at the slot's last item)

  auto issue = [&](auto sC) {
    constexpr int s = decltype(sC)::value;
    const long p0 = (long)cur_pt * BP;
    const int kc = cur_c * LY_BK;
    s_p[s] = p0; s_kc[s] = kc;
    if (cur_c == 0 || !need_nhw) {                         // first chunk of a tile: describe its rows (else: same tile as the
#pragma unroll                                             // other set, which holds the previous chunk)
      for (int e = 0; e < NV; ++e) {
        const long gp = p0 + prow + RSTEP * e;
        int n = -1, h = 0, w = 0;
        long row0 = gp;
        if (gp < P.M) {
          n = 0;
          if (need_nhw) {
            n = ly_fdiv((int)gp, HW, invHW);
            const int rem = (int)gp - n * HW;
            h = ly_fdiv(rem, P.W, invW);
            w = rem - h * P.W;
            if (GATHER == LY_GATHER_UP2)
              row0 = ((long)n * (P.H >> 1) + (h >> 1)) * (P.W >> 1) + (w >> 1);
            else if (GATHER == LY_GATHER_PATCH)
              row0 = (((long)n * P.Hin + (long)h * P.ks) * P.Win + (long)w * P.ks);
            else if (GATHER == LY_GATHER_PATCH_NCHW)
              row0 = ((long)n * P.Cin * P.Hin + (long)h * P.ks) * P.Win + (long)w * P.ks;
          }
        }
        t_row0[s][e] = row0; t_n[s][e] = n; t_hw[s][e] = (h << 16) | w;
      }
    } else {
#pragma unroll
      for (int e = 0; e < NV; ++e) { t_row0[s][e] = t_row0[1 - s][e]; t_n[s][e] = t_n[1 - s][e]; t_hw[s][e] = t_hw[1 - s][e]; }
    }
    const int kk = kc + 4 * k4;
    const bool kok = kk < P.K;
    long koff;
    const float* src = P.a0;
    long rowmul = P.lda0;
    bool second = false;
    if (GATHER == LY_GATHER_PATCH) {
      const int seg = kk / P.pk, within = kk - seg * P.pk;
      koff = (long)seg * P.Win * P.lda0 + within;
    } else if (GATHER == LY_GATHER_PATCH_NCHW) {
      const int c = kk >> 4, ky = (kk >> 2) & 3;
      koff = ((long)c * P.Hin + ky) * P.Win;
      rowmul = 1;
    } else {
      second = kk >= P.k0;
      koff = second ? kk - P.k0 : kk;
      if (second) { src = P.a1; rowmul = P.lda1; }
    }
#pragma unroll
    for (int e = 0; e < NV; ++e) {
      const bool ok = kok && t_n[s][e] >= 0;
      const long row = second ? (p0 + prow + RSTEP * e) : t_row0[s][e];
      pv[s][e] = ly_ldg4(ok ? src + row * rowmul + koff : P.a0);
    }
    // advance the cursor; past the last item it stays there (the surplus issues re-read it, harmlessly, so that every
    // pass through the loop issues the same loads and the compiler's vmcnt bookkeeping is exact)
    if (cur_c + 1 < nchunk) ++cur_c;
    else if (cur_pt + nslots < gx) { cur_pt += nslots; cur_c = 0; }
  };

  auto commit = [&](auto sC, int buf) {
    constexpr int s = decltype(sC)::value;
    char* hi = xs + buf * 2 * PLANE;
    char* lo = hi + PLANE;
    const long p0 = s_p[s];
    const int kk = s_kc[s] + 4 * k4;
    f32x4 v[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) v[e] = (kk < P.K && t_n[s][e] >= 0) ? pv[s][e] : zero;
    if (PRO == LY_PRO_GATE) {
      // CoordAtt factors: small L2-resident tables, fetched here (two more register sets of them do not fit)
      const bool kok = kk < P.k0;
      f32x4 gw[NV], gh[NV], rr[NV];
#pragma unroll
      for (int e = 0; e < NV; ++e) {
        const bool ok = kok && t_n[s][e] >= 0;
        const int n = ok ? t_n[s][e] : 0, h = ok ? (t_hw[s][e] >> 16) : 0, w = ok ? (t_hw[s][e] & 0xffff) : 0;
        const int kq = ok ? kk : 0;
        gw[e] = ly_ldg4(P.g_w + ((long)n * P.W + w) * P.k0 + kq);
        gh[e] = ly_ldg4(P.g_h + ((long)n * P.H + h) * P.k0 + kq);
        rr[e] = (P.res && ok) ? ly_ldg4(P.res + (p0 + prow + RSTEP * e) * P.ldres + kk) : zero;
      }
#pragma unroll
      for (int e = 0; e < NV; ++e)
        if (kok && t_n[s][e] >= 0) v[e] = v[e] * gw[e] * gh[e] + rr[e];
    } else if (PRO == LY_PRO_AFFINE_RELU_CA) {
      const bool kok = kk < P.K;
      const int kq = kok ? kk : 0;
      const f32x4 sa = ly_ldg4(P.p_scale + kq), sb = ly_ldg4(P.p_shift + kq);
      f32x4 ca[NV];
#pragma unroll
      for (int e = 0; e < NV; ++e) ca[e] = ly_ldg4(P.p_ca + (long)(t_n[s][e] >= 0 ? t_n[s][e] : 0) * P.K + kq);
#pragma unroll
      for (int e = 0; e < NV; ++e)
        if (kok && t_n[s][e] >= 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[e][r] = fmaxf(v[e][r] * sa[r] + sb[r], 0.f) * ca[e][r];
        }
    }
#pragma unroll
    for (int e = 0; e < NV; ++e) ly_lds_put4(hi, lo, (prow + RSTEP * e) * LY_RSX, 4 * k4, v[e]);
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[t][n] = zero;
  long wbase[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    int tt = (by * WC + wc) * MT + t;
    wbase[t] = (long)(tt < T ? tt : T - 1) * S;
  }
  const uint4* wpk = reinterpret_cast<const uint4*>(P.wp);
  const int pixgrp = wp_ * (16 * NT);
  const bool vec_ok = (P.ldo & 3) == 0;
  const int act = P.act;
  float* const stats = P.stats;
  float rsv[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) rsv[n] = 1.f;
  float esc[MT][4], esh[MT][4];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int c = 16 * ((by * WC + wc) * MT + t) + 4 * lq;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = c + r < P.N;
      esc[t][r] = (ok && P.e_scale) ? P.e_scale[c + r] : 1.f;
      esh[t][r] = (ok && P.e_shift) ? P.e_shift[c + r] : 0.f;
    }
  }
  ly_l2_warm(P.wp, (long)T * S * 2048, P.stats ? P.stats : P.out);
  LyWFrag wcur[MT], wnxt[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) wcur[t] = ly_wfrag(wpk, wbase[t], lane);

  int pt = slot, c = 0, buf = 0;                           // item being contracted
  long p0 = (long)slot * BP;
  issue(LyIc<0>());
  issue(LyIc<1>());
  commit(LyIc<0>(), 0);
  __syncthreads();

  // one item: weights, re-issue the vacated set two items ahead, contract, commit the next item, barrier, epilogue at tile end
  auto item = [&](auto sC) -> bool {
    constexpr int s = decltype(sC)::value;                 // set that held THIS item (already committed): free
    {
      const int g1 = 2 * c + 1 < S ? 2 * c + 1 : 0;        // second k-step's weights first (older than the prefetch in the queue)
#pragma unroll
      for (int t = 0; t < MT; ++t) wnxt[t] = ly_wfrag(wpk, wbase[t] + g1, lane);
    }
    if (PRO == LY_PRO_AFFINE_RELU_CA) {
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const long gp = p0 + pixgrp + 16 * n + li;
        rsv[n] = P.rowscale[gp < P.M ? gp : 0];
      }
    }
    issue(sC);                                             // item i+2
    const char* hi = xs + buf * 2 * PLANE;
    const char* lo = hi + PLANE;
#pragma unroll
    for (int st = 0; st < SPC; ++st) {
      if (st == SPC - 1) {                                 // weights of the next item's first step
        const int gn = SPC * (c + 1) < S ? SPC * (c + 1) : 0;
#pragma unroll
        for (int t = 0; t < MT; ++t) wnxt[t] = ly_wfrag(wpk, wbase[t] + gn, lane);
      }
      bf16x8 xh[NT], xl[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int rb = (pixgrp + 16 * n + li) * LY_RSX;
        xh[n] = ly_lds_frag(hi, rb, st, lq);
        xl[n] = ly_lds_frag(lo, rb, st, lq);
      }
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[t][n] = ly_mfma3(wcur[t].hi, wcur[t].lo, xh[n], xl[n], acc[t][n]);
#pragma unroll
      for (int t = 0; t < MT; ++t) wcur[t] = wnxt[t];
    }
    commit(LyIc<1 - s>(), buf ^ 1);                        // item i+1
    __syncthreads();
    buf ^= 1;
    if (c + 1 < nchunk) { ++c; return true; }
    // ---- epilogue of tile pt ---------------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int tt = (by * WC + wc) * MT + t;
      const int cc = 16 * tt + 4 * lq;
      if (tt < T && cc < P.N) {
        f32x4 st1 = zero, st2 = zero;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const long gp = p0 + pixgrp + 16 * n + li;
          if (gp < P.M) {
            const float rs = PRO == LY_PRO_AFFINE_RELU_CA ? rsv[n] : 1.f;
            f32x4 u;
#pragma unroll
            for (int r = 0; r < 4; ++r) u[r] = acc[t][n][r] * rs * esc[t][r] + esh[t][r];
            if (stats) {
              st1 += u;
              st2 += u * u;
              if (!P.out) continue;                        // pure statistics pass; with `out` the value is stored as well
            }
            const f32x4 v = ly_act4(u, act);
            float* o = P.out + gp * P.ldo + cc;
            if (vec_ok && cc + 3 < P.N) {
              ly_stg4(o, v);
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (cc + r < P.N) o[r] = v[r];
            }
          }
        }
        if (stats) ly_stats_flush(stats, P.N, cc, st1, st2);
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[t][n] = zero;
    }
    pt += nslots;
    if (pt >= gx) return false;
    p0 = (long)pt * BP;
    c = 0;
    return true;
  };
  while (true) {
    if (!item(LyIc<0>())) break;
    if (!item(LyIc<1>())) break;
  }
}

// same launch contract as ly_gemm_kernel; selected by launch_gemm_bk when the two-deep pipeline applies
template <int NT, int MT, int WC, int GATHER, int PRO>
__global__ __launch_bounds__(LY_THREADS, LY_GEMM_MINW) void ly_gemm_kernel_d2(const LyGemmParams P, const int gy, const int nslots, const int gx, const int dbg) {
  ly_gemm_body2<NT, MT, WC, GATHER, PRO>(P, gy, nslots, gx);
}

template <int NT, int MT, int WC, int GATHER, int PRO>
__global__ __launch_bounds__(LY_THREADS, LY_GEMM_MINW) void ly_gemm_kernel(const LyGemmParams P, const int gy, const int nslots, const int gx, const int dbg) {
  ly_gemm_body<NT, MT, WC, GATHER, PRO, false, 64>(P, gy, nslots, gx, 0);
}
// 128-wide K stage (K >= 128)
template <int NT, int MT, int WC, int GATHER, int PRO>
__global__ __launch_bounds__(LY_THREADS, LY_GEMM_MINW) void ly_gemm_kernel_k128(const LyGemmParams P, const int gy, const int nslots, const int gx, const int dbg) {
  ly_gemm_body<NT, MT, WC, GATHER, PRO, false, 128>(P, gy, nslots, gx, 0);
}
// same body with the ablation switches of ly_debug_set_gemm compiled in (tools/ only)
template <int NT, int MT, int WC, int GATHER, int PRO>
__global__ __launch_bounds__(LY_THREADS, LY_GEMM_MINW) void ly_gemm_kernel_dbg(const LyGemmParams P, const int gy, const int nslots, const int gx, const int dbg) {
  ly_gemm_body<NT, MT, WC, GATHER, PRO, true, 64>(P, gy, nslots, gx, dbg);
}

static int g_gemm_dbg = 0;      // ablation aid: 1 skip commit (split + LDS write), 2 skip MFMA, 4 skip prefetch loads, 8 skip stores
extern "C" int ly_debug_set_gemm(int v) { g_gemm_dbg = v; return 0; }
static int g_gemm_cfg = 0;      // 0 = heuristic; otherwise forced NT*100 + MT*10 + WC (tuning aid)
extern "C" int ly_debug_set_gemm_cfg(int cfg) { g_gemm_cfg = cfg; return 0; }

static int g_gemm_d2 = 1;       // 1 = two-deep prefetch kernel (ly_gemm_kernel_d2), 0 = one-deep (A/B aid)
extern "C" int ly_debug_set_gemm_d2(int v) { g_gemm_d2 = v; return 0; }
static int g_gemm_bk = 0;       // 0 = default (64), 128 = 128-wide K stage (tuning aid: measured slower, its 272 registers leave one wave per SIMD)
extern "C" int ly_debug_set_gemm_bk(int v) { g_gemm_bk = v; return 0; }

template <int NT, int MT, int WC, int GATHER, int PRO>
static int launch_gemm_d2(const LyGemmParams& P, hipStream_t st) {
  constexpr int BP = 16 * NT * (4 / WC);
  constexpr int BN = 16 * MT * WC;
  constexpr size_t lds = 4 * (size_t)BP * (2 * 64 + 16);
  long gx = (P.M + BP - 1) / BP;
  int gy = (P.N + BN - 1) / BN;
  LY_CHECK(gx < (1L << 30), "gemm: too many pixel tiles");
  auto k = ly_gemm_kernel_d2<NT, MT, WC, GATHER, PRO>;
  static int per_cu = 0;
  if (per_cu == 0) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    LY_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    int nb = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(k), LY_THREADS, lds);
    LY_CHECK(e == hipSuccess, "hipOccupancyMaxActiveBlocksPerMultiprocessor: %s", hipGetErrorString(e));
    per_cu = nb < 1 ? 1 : (nb > 8 ? 8 : nb);
  }
  long nslots = (256L * per_cu) / gy;
  if (nslots < 1) nslots = 1;
  if (nslots > gx) nslots = gx;
  hipLaunchKernelGGL(k, dim3((unsigned)(nslots * gy)), dim3(LY_THREADS), lds, st, P, gy, (int)nslots, (int)gx, 0);
  LY_LAUNCH_CHECK();
  return 0;
}

template <int NT, int MT, int WC, int GATHER, int PRO, int BK>
static int launch_gemm_bk(const LyGemmParams& P, hipStream_t st) {
  constexpr int BP = 16 * NT * (4 / WC);
  constexpr int BN = 16 * MT * WC;
  constexpr size_t lds = 4 * (size_t)BP * (2 * BK + 16);
  long gx = (P.M + BP - 1) / BP;
  int gy = (P.N + BN - 1) / BN;
  LY_CHECK(gx < (1L << 30), "gemm: too many pixel tiles");
  // the A/B variants (one-deep, 128-wide K stage, ablation switches) are built for the three tile shapes the heuristic
  // picks; a tile shape forced through ly_debug_set_gemm_cfg always runs the production (two-deep) kernel
  constexpr bool variants = (NT == 4 && MT == 2 && WC == 4) || (NT == 4 && MT == 1 && WC == 4) || (NT == 2 && MT == 2 && WC == 1);
  if (!variants || (BK == 64 && g_gemm_d2 && !g_gemm_dbg)) return launch_gemm_d2<NT, MT, WC, GATHER, PRO>(P, st);
  if constexpr (variants) {
  auto k = BK == 128 ? ly_gemm_kernel_k128<NT, MT, WC, GATHER, PRO> : ly_gemm_kernel<NT, MT, WC, GATHER, PRO>;
  static int per_cu = 0;            // co-resident blocks per CU (registers + LDS), measured once per instantiation
  if (per_cu == 0) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    LY_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    int nb = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(k), LY_THREADS, lds);
    LY_CHECK(e == hipSuccess, "hipOccupancyMaxActiveBlocksPerMultiprocessor: %s", hipGetErrorString(e));
    per_cu = nb < 1 ? 1 : (nb > 8 ? 8 : nb);
  }
  // persistent grid = exactly the blocks that can be resident at once (a larger grid would run in rounds)
  long nslots = (256L * per_cu) / gy;
  if (nslots < 1) nslots = 1;
  if (nslots > gx) nslots = gx;
  if (g_gemm_dbg && BK == 64) {
    auto kd = ly_gemm_kernel_dbg<NT, MT, WC, GATHER, PRO>;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)); attr = true; }
    hipLaunchKernelGGL(kd, dim3((unsigned)(nslots * gy)), dim3(LY_THREADS), lds, st, P, gy, (int)nslots, (int)gx, g_gemm_dbg);
  } else {
    hipLaunchKernelGGL(k, dim3((unsigned)(nslots * gy)), dim3(LY_THREADS), lds, st, P, gy, (int)nslots, (int)gx, 0);
  }
  LY_LAUNCH_CHECK();
  }
  return 0;
}

template <int NT, int MT, int WC, int GATHER, int PRO>
static int launch_gemm_mode(const LyGemmParams& P, hipStream_t st) {
  constexpr int BP = 16 * NT * (4 / WC);
  constexpr bool variants = (NT == 4 && MT == 2 && WC == 4) || (NT == 4 && MT == 1 && WC == 4) || (NT == 2 && MT == 2 && WC == 1);
  constexpr bool can128 = variants && BP * 32 % LY_THREADS == 0 && BP * 32 / LY_THREADS >= 1 && GATHER != LY_GATHER_PATCH_NCHW;
  const bool want128 = g_gemm_bk == 128;
  if constexpr (can128) {
    if (want128) return launch_gemm_bk<NT, MT, WC, GATHER, PRO, 128>(P, st);
  }
  return launch_gemm_bk<NT, MT, WC, GATHER, PRO, 64>(P, st);
}

template <int NT, int MT, int WC>
static int launch_gemm(const LyGemmParams& P, hipStream_t st) {
  if (P.gather == LY_GATHER_PATCH) return launch_gemm_mode<NT, MT, WC, LY_GATHER_PATCH, LY_PRO_NONE>(P, st);
  if (P.gather == LY_GATHER_PATCH_NCHW) return launch_gemm_mode<NT, MT, WC, LY_GATHER_PATCH_NCHW, LY_PRO_NONE>(P, st);
  if (P.gather == LY_GATHER_UP2) {
    if (P.pro == LY_PRO_NONE) return launch_gemm_mode<NT, MT, WC, LY_GATHER_UP2, LY_PRO_NONE>(P, st);
    ly_set_error("gemm: upsampled source with a prologue is not built");
    return -1;
  }
  if (P.pro == LY_PRO_GATE) return launch_gemm_mode<NT, MT, WC, LY_GATHER_ROWS, LY_PRO_GATE>(P, st);
  if (P.pro == LY_PRO_AFFINE_RELU_CA) return launch_gemm_mode<NT, MT, WC, LY_GATHER_ROWS, LY_PRO_AFFINE_RELU_CA>(P, st);
  return launch_gemm_mode<NT, MT, WC, LY_GATHER_ROWS, LY_PRO_NONE>(P, st);
}

extern "C" int ly_gemm_fwd(const LyGemmParams* p, void* stream) {
  LY_CHECK(p, "gemm: null params");
  const LyGemmParams& P = *p;
  LY_CHECK(P.a0 && P.wp && (P.out || P.stats), "gemm: null a0/wp/out");
  LY_CHECK(P.M > 0 && P.K > 0 && P.N > 0 && P.H > 0 && P.W > 0, "gemm: bad sizes M=%ld K=%d N=%d", P.M, P.K, P.N);
  LY_CHECK((P.K & 3) == 0, "gemm: K=%d must be a multiple of 4", P.K);
  if (P.gather == LY_GATHER_ROWS || P.gather == LY_GATHER_UP2) {
    LY_CHECK((P.lda0 & 3) == 0 && (P.k0 & 3) == 0 && P.k0 <= P.K && P.k0 > 0, "gemm: lda0=%d k0=%d must be multiples of 4", P.lda0, P.k0);
    LY_CHECK(P.k0 == P.K || (P.a1 && (P.lda1 & 3) == 0), "gemm: second source missing or misaligned");
    if (P.gather == LY_GATHER_UP2) LY_CHECK((P.H & 1) == 0 && (P.W & 1) == 0, "gemm: upsample source needs even H, W");
  } else if (P.gather == LY_GATHER_PATCH) {
    LY_CHECK(P.ks > 0 && P.pk == P.ks * P.lda0 && (P.lda0 & 3) == 0 && P.K == P.ks * P.pk, "gemm: patch gather misconfigured");
    LY_CHECK(P.Hin >= P.H * P.ks && P.Win >= P.W * P.ks, "gemm: patch gather input too small");
  } else if (P.gather == LY_GATHER_PATCH_NCHW) {
    LY_CHECK(P.ks == 4 && (P.Win & 3) == 0 && P.K == P.Cin * 16, "gemm: NCHW patch gather needs ks=4, Win%%4==0");
    LY_CHECK(P.Hin >= P.H * P.ks && P.Win >= P.W * P.ks, "gemm: patch gather input too small");
  } else {
    LY_CHECK(false, "gemm: unknown gather mode %d", P.gather);
  }
  if (P.pro == LY_PRO_GATE) LY_CHECK(P.g_h && P.g_w, "gemm: gate prologue needs g_h/g_w");
  if (P.pro == LY_PRO_AFFINE_RELU_CA) LY_CHECK(P.p_scale && P.p_shift && P.p_ca && P.rowscale, "gemm: affine prologue needs scale/shift/ca and rowscale");
  else LY_CHECK(!P.rowscale, "gemm: rowscale is only built together with the affine (RFCBAM k=1) prologue");
  LY_CHECK(P.M < (1L << 24), "gemm: M=%ld pixels exceeds the 2^24 limit of the fast index path", P.M);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  switch (g_gemm_cfg) {
    case 844: return launch_gemm<8, 4, 4>(P, st);
    case 824: return launch_gemm<8, 2, 4>(P, st);
    case 814: return launch_gemm<8, 1, 4>(P, st);
    case 444: return launch_gemm<4, 4, 4>(P, st);
    case 424: return launch_gemm<4, 2, 4>(P, st);
    case 414: return launch_gemm<4, 1, 4>(P, st);
    case 422: return launch_gemm<4, 2, 2>(P, st);
    case 442: return launch_gemm<4, 4, 2>(P, st);
    case 242: return launch_gemm<2, 4, 2>(P, st);
    case 221: return launch_gemm<2, 2, 1>(P, st);
    case 121: return launch_gemm<1, 2, 1>(P, st);
    default: break;
  }
  // measured on MI355X (tools_gcfg.py): 64-pixel tiles with 3 co-resident blocks per CU beat the
  // 128-pixel tiles (1 wave/SIMD) on every LEAD-YOLO shape
  if (P.N > 64) return launch_gemm<4, 2, 4>(P, st);    // 64 px x 128 ch per block
  if (P.N > 32) return launch_gemm<4, 1, 4>(P, st);    // 64 px x 64 ch
  return launch_gemm<2, 2, 1>(P, st);                  // 128 px x 32 ch
}
