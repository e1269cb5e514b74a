#pragma once
// Pointwise-convolution GEMM with fused gather/prologue and epilogue, gfx950; storage dtype T = float (bf16x3 products) or
// __bf16 (plain bf16 products), fp32 accumulation in both (ly_tile.hpp).
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
// tiles of 16x16 on v_mfma_f32_16x16x32_bf16.
//
// Persistent two-deep pipeline: a block owns a strided set of pixel tiles and walks the work items (tile, K chunk).  A K
// chunk is 16 vectors of 16 bytes per pixel row: BK = 64 fp32 or 128 bf16 elements — the same bytes in flight per thread
// in both dtypes.  A chunk contracts in a few hundred MFMA cycles but its loads need 1-2 us to arrive, so the raw values
// of items i+1 AND i+2 are in flight while item i is contracted: two register sets, used alternately (the item loop is
// unrolled by two so the set index is static); item i+2 is issued into the set item i vacated when it was committed to
// LDS.  The item body is straight-line code (surplus prefetches re-read the last item, ragged K contracts LDS zeros):
// the compiler's s_waitcnt bookkeeping is exact only then.  Weight fragments of the chunk's later k-steps are requested
// BEFORE the activation prefetch (vmcnt retires in order), the first step of the next item during the last step.
#include "ly_tile.hpp"
#include "ly_params.h"
#ifndef LY_GEMM_DEPTH
#define LY_GEMM_DEPTH 3
#endif

static __device__ float ly_gemm_trash[64 * 4];      // where the stores of rows past M land (never read)

// three waves per SIMD where 168 registers suffice: bf16 rows, one resident chunk (the variant most 1x1 launches of a step take)
#ifndef LY_GEMM_W3
#define LY_GEMM_W3(TI, PRO, NCH, MT) (sizeof(TI) == 2 && (PRO) == LY_PRO_NONE && (NCH) == 1 && GATHER == LY_GATHER_ROWS && FAST == 1 && LY_GEMM_W3_ON)
#ifndef LY_GEMM_W3_ON
#define LY_GEMM_W3_ON 1
#endif
#ifndef LY_GEMM_W3_DEPTH
#define LY_GEMM_W3_DEPTH 2
#endif
#endif

template <int V>
struct LyIc { static constexpr int value = V; };

// TI: element type of the sources (fp32 image for LY_GATHER_PATCH_NCHW whatever the output is), TO: of res / out
template <typename TI, typename TO, int NT, int MT, int WC, int GATHER, int PRO, int D, int NCH, int FAST>
__device__ __forceinline__ void ly_gemm_body2(const LyGemmParams& P, const int gy, const int nslots, const int gx) {
  using TR = LyT<TI>;
  using RV = typename TR::RV;
  constexpr int VW = TR::VW, PL = TR::PL, NQ = VW / 4;
  constexpr int LY_BK = 16 * VW;
  constexpr int WP = 4 / WC;
  constexpr int BP = 16 * NT * WP;
  constexpr int LY_RSQ = ly_qrs(LY_BK / 32), LY_PS = ly_qps(BP * LY_RSQ);      // lane-group operand image (ly_tile.hpp): 4 planes of BP rows
  constexpr int KQ = 16;                                  // vector columns per chunk
  constexpr int RSTEP = LY_THREADS / KQ;
  constexpr int SPC = LY_BK / 32;                         // k-steps per chunk: 2 (fp32) / 4 (bf16)
  constexpr int NV = BP * KQ / LY_THREADS;
  constexpr int PLANE = 4 * LY_PS;
  static_assert(NV >= 1 && BP * KQ % LY_THREADS == 0, "tile must divide evenly over the block");
  extern __shared__ f32x4 ly_smem4[];
  char* xs = reinterpret_cast<char*>(ly_smem4);           // [buf][hi | lo][lane group][BP][RSQ]

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
  // thread -> (vector column k4, first pixel row prow).  NHWC sources: consecutive lanes take consecutive k (one pixel row is
  // K-contiguous).  NCHW image patches: consecutive k are different (channel, ky) planes, megabytes apart, while consecutive
  // output pixels of one plane row ARE contiguous (16 B each) — so there consecutive lanes take consecutive pixels.
  static_assert(KQ == RSTEP, "the NCHW lane mapping swaps the two 16-way indices");
  const int k4 = GATHER == LY_GATHER_PATCH_NCHW ? tid / RSTEP : tid % KQ;
  const int prow = GATHER == LY_GATHER_PATCH_NCHW ? tid % RSTEP : tid / KQ;
  if (slot >= gx) return;
  const TI* const a0 = reinterpret_cast<const TI*>(P.a0);
  const TI* const a1 = reinterpret_cast<const TI*>(P.a1);
  const TO* const res = reinterpret_cast<const TO*>(P.res);
  TO* const out = reinterpret_cast<TO*>(P.out);

  // ---- staging state, one copy per register set ---------------------------------------------------
  RV pv[D][NV];
  long t_row0[D][NV];
  int t_n[D][NV], t_hw[D][NV];
  long s_p[D];                                             // tile start and K offset of the item held by the set
  int s_kc[D];
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
      for (int e = 0; e < NV; ++e) { t_row0[s][e] = t_row0[(s + D - 1) % D][e]; t_n[s][e] = t_n[(s + D - 1) % D][e]; t_hw[s][e] = t_hw[(s + D - 1) % D][e]; }
    }
    const int kk = kc + VW * k4;
    const bool kok = kk < P.K;
    long koff;
    const TI* src = a0;
    long rowmul = P.lda0;
    bool second = false;
    if (GATHER == LY_GATHER_PATCH) {
      const int seg = kk / P.pk, within = kk - seg * P.pk;
      koff = (long)seg * P.Win * P.lda0 + within;
    } else if (GATHER == LY_GATHER_PATCH_NCHW) {
      const int c = kk >> 4, ky = (kk >> 2) & 3;           // ks == 4, fp32 / uint8 image: one 4-pixel vector = one (c, ky) input row segment
      koff = ((long)c * P.Hin + ky) * P.Win;
      rowmul = 1;
    } else {
      second = kk >= P.k0;
      koff = second ? kk - P.k0 : kk;
      if (second) { src = a1; rowmul = P.lda1; }
    }
#pragma unroll
    for (int e = 0; e < NV; ++e) {
      const bool ok = kok && t_n[s][e] >= 0;
      const long row = second ? (p0 + prow + RSTEP * e) : t_row0[s][e];
      pv[s][e] = ly_ldrv<TI>(ok ? src + row * rowmul + koff : a0);
    }
    // advance the cursor; past the last item it stays there (the surplus issues re-read it, harmlessly, so that every
    // pass through the loop issues the same loads and the compiler's vmcnt bookkeeping is exact)
    if (cur_c + 1 < nchunk) ++cur_c;
    else if (cur_pt + nslots < gx) { cur_pt += nslots; cur_c = 0; }
  };

  auto commit = [&](auto sC, int buf) {
    constexpr int s = decltype(sC)::value;
    char* hi = xs + buf * PL * PLANE;
    char* lo = hi + (PL - 1) * PLANE;
    const long p0 = s_p[s];
    const int kk = s_kc[s] + VW * k4;
    RV v[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) {
      v[e] = pv[s][e];
      if (!(kk < P.K && t_n[s][e] >= 0)) ly_zero_raw(v[e]);
    }
    if constexpr (PRO == LY_PRO_GATE) {
      // CoordAtt factors: small L2-resident fp32 tables, fetched here (two more register sets of them do not fit)
      const bool kok = kk < P.k0;
      f32x4 gw[NV][NQ], gh[NV][NQ];
      RV rr[NV];
#pragma unroll
      for (int e = 0; e < NV; ++e) {
        const bool ok = kok && t_n[s][e] >= 0;
        const int n = ok ? t_n[s][e] : 0, h = ok ? (t_hw[s][e] >> 16) : 0, w = ok ? (t_hw[s][e] & 0xffff) : 0;
        const int kq = ok ? kk : 0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          gw[e][q] = ly_ldg4(P.g_w + ((long)n * P.W + w) * P.k0 + kq + 4 * q);
          gh[e][q] = ly_ldg4(P.g_h + ((long)n * P.H + h) * P.k0 + kq + 4 * q);
        }
        if (res && ok) rr[e] = ly_ldrv<TO>(res + (p0 + prow + RSTEP * e) * P.ldres + kk);
        else ly_zero_raw(rr[e]);
      }
#pragma unroll
      for (int e = 0; e < NV; ++e)
        if (kok && t_n[s][e] >= 0) {
          f32x4 x[NQ], r[NQ];
          ly_rv_unpack(v[e], x);
          ly_rv_unpack(rr[e], r);
#pragma unroll
          for (int q = 0; q < NQ; ++q) x[q] = x[q] * gw[e][q] * gh[e][q] + r[q];
          v[e] = ly_rv_pack(x, (RV*)nullptr);
        }
    } else if constexpr (PRO == LY_PRO_AFFINE_RELU_CA) {
      const bool kok = kk < P.K;
      const int kq = kok ? kk : 0;
      f32x4 sa[NQ], sb[NQ], ca[NV][NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) { sa[q] = ly_ldg4(P.p_scale + kq + 4 * q); sb[q] = ly_ldg4(P.p_shift + kq + 4 * q); }
#pragma unroll
      for (int e = 0; e < NV; ++e)
#pragma unroll
        for (int q = 0; q < NQ; ++q) ca[e][q] = ly_ldg4(P.p_ca + (long)(t_n[s][e] >= 0 ? t_n[s][e] : 0) * P.K + kq + 4 * q);
#pragma unroll
      for (int e = 0; e < NV; ++e)
        if (kok && t_n[s][e] >= 0) {
          f32x4 x[NQ];
          ly_rv_unpack(v[e], x);
#pragma unroll
          for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) x[q][r] = fmaxf(x[q][r] * sa[q][r] + sb[q][r], 0.f) * ca[e][q][r];
          v[e] = ly_rv_pack(x, (RV*)nullptr);
        }
    }
#pragma unroll
    for (int e = 0; e < NV; ++e) ly_img_put_rv(hi, lo, (prow + RSTEP * e) * LY_RSQ, LY_PS, VW * k4, v[e]);
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
  double* const stats = P.stats;
  float rsv[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) rsv[n] = 1.f;
  f32x4 sum1[FAST == 2 ? MT : 1], sum2[FAST == 2 ? MT : 1];
  if constexpr (FAST == 2) {
#pragma unroll
    for (int t = 0; t < MT; ++t) { sum1[t] = zero; sum2[t] = zero; }
  }
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
  ly_l2_warm(P.wp, (long)T * S * PL * 1024, P.stats ? reinterpret_cast<float*>(P.stats) : reinterpret_cast<float*>(P.out));
  // NCH == 0: weights streamed, wq[j] = k-step j of the item being contracted.  NCH > 0 (K is exactly NCH chunks): ALL weight
  // fragments of the block's channel slice stay in registers.  That is not about the L2 traffic: vmcnt retires IN ORDER, so a
  // load that is consumed in the item that issued it (a streamed weight fragment) makes its s_waitcnt drain every older load —
  // the activation prefetches of the next items — and the pipeline is never more than one item deep, whatever D says.
  // NCH == 0 without a prologue (WDB): the weights of the NEXT item are requested at the top of an item, before the activation
  // prefetch, into the other of two fragment sets — waiting for them an item later then only retires loads that are needed by then anyway.
  constexpr bool WDB = NCH == 0 && PRO == LY_PRO_NONE && D == 2;
  constexpr int NW = NCH > 0 ? NCH : (WDB ? 2 : 1);
  LyWF<PL> wq[NW][SPC][MT];
  if constexpr (NCH > 0) {
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc)
#pragma unroll
      for (int j = 0; j < SPC; ++j) {
        const int gj = SPC * cc + j < S ? SPC * cc + j : 0;
#pragma unroll
        for (int t = 0; t < MT; ++t) wq[cc][j][t] = ly_wfragp<PL>(wpk, wbase[t] + gj, lane);
      }
  } else if constexpr (WDB) {
#pragma unroll
    for (int j = 0; j < SPC; ++j) {
      const int gj = j < S ? j : 0;
#pragma unroll
      for (int t = 0; t < MT; ++t) wq[0][j][t] = ly_wfragp<PL>(wpk, wbase[t] + gj, lane);
    }
  } else {
#pragma unroll
    for (int t = 0; t < MT; ++t) wq[0][0][t] = ly_wfragp<PL>(wpk, wbase[t], lane);
  }

  int pt = slot, c = 0, buf = 0;                           // item being contracted
  long p0 = (long)slot * BP;
  issue(LyIc<0>());
  issue(LyIc<1>());
  if constexpr (D > 2) issue(LyIc<2 % D>());
  commit(LyIc<0>(), 0);
  __syncthreads();

  // one item: weights, re-issue the vacated set two items ahead, contract, commit the next item, barrier, epilogue at tile end
  auto item = [&](auto sC, auto cC) -> bool {
    constexpr int s = decltype(sC)::value;                 // set that held THIS item (already committed): free
    constexpr int CS = NCH > 0 ? decltype(cC)::value : (WDB ? s : 0);  // resident weights: the chunk index is static
    if constexpr (WDB) {
      const int cn = c + 1 < nchunk ? c + 1 : 0;           // chunk of the next item (a new tile starts at chunk 0)
#pragma unroll
      for (int j = 0; j < SPC; ++j) {
        const int gj = SPC * cn + j < S ? SPC * cn + j : 0;
#pragma unroll
        for (int t = 0; t < MT; ++t) wq[1 - s][j][t] = ly_wfragp<PL>(wpk, wbase[t] + gj, lane);
      }
    } else if constexpr (NCH == 0) {
#pragma unroll
      for (int j = 1; j < SPC; ++j) {                      // later k-steps' weights first (older than the prefetch in the queue)
        const int gj = SPC * c + j < S ? SPC * c + j : 0;
#pragma unroll
        for (int t = 0; t < MT; ++t) wq[0][j][t] = ly_wfragp<PL>(wpk, wbase[t] + gj, lane);
      }
    }
    if (PRO == LY_PRO_AFFINE_RELU_CA) {
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const long gp = p0 + pixgrp + 16 * n + li;
        rsv[n] = P.rowscale[gp < P.M ? gp : 0];
      }
    }
    issue(sC);                                             // item i+2
    const char* hi = xs + buf * PL * PLANE;
    const char* lo = hi + (PL - 1) * PLANE;
#pragma unroll
    for (int st = 0; st < SPC; ++st) {
      bf16x8 xh[NT], xl[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int rb = (pixgrp + 16 * n + li) * LY_RSQ;
        xh[n] = ly_img_frag(hi, rb, LY_PS, st, lq);
        if constexpr (PL == 2) xl[n] = ly_img_frag(lo, rb, LY_PS, st, lq);
        else xl[n] = xh[n];
      }
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[t][n] = ly_mfmap<PL>(wq[CS][st][t], xh[n], xl[n], acc[t][n]);
      if constexpr (NCH == 0 && !WDB) {
        if (st == SPC - 1) {                               // weights of the next item's first step into the slot step 0 vacated
          const int gn = SPC * (c + 1) < S ? SPC * (c + 1) : 0; // (absent steps of a ragged last chunk contract LDS zeros with clamped weights: no branch)
#pragma unroll
          for (int t = 0; t < MT; ++t) wq[0][0][t] = ly_wfragp<PL>(wpk, wbase[t] + gn, lane);
        }
      }
    }
    commit(LyIc<(s + 1) % D>(), buf ^ 1);                  // item i+1
    __syncthreads();
    buf ^= 1;
    if constexpr (NCH > 0) {
      if (CS + 1 < NCH) return true;
    } else {
      if (c + 1 < nchunk) { ++c; return true; }
    }
    // ---- epilogue of tile pt ---------------------------------------------------------------------
    if constexpr (FAST != 0) {
      // N and ldo are multiples of 4, so a lane's 4 channels are all valid or all outside N: ONE path, every lane stores (rows
      // past M and channel quads past N go to a scratch line).  A store under a branch — even an exec-skip around a masked store — leaves the compiler
      // two vmcnt histories to merge, and the next item's wait for its prefetch then also waits for these stores to be
      // acknowledged: a full memory round trip at every tile end.
      // scat_ks > 0: the store IS the adjoint of a k = s patch gather (LyGemmParams.scat_ks): row m = (n, h, w), column group (ky, kx) -> the
      // pixel (ks h + ky, ks w + kx) of the ks-times larger map.  Only addresses change; the store itself stays unconditional.
      // (FAST == 3: an instantiation of its own — as a run-time switch its address registers cost the resident variants spills)
      long srow[FAST >= 3 ? NT : 1];
      if constexpr (FAST >= 3) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const long gp = p0 + pixgrp + 16 * n + li;
          const int g = (int)(gp < P.M ? gp : P.M - 1);
          const int ni = ly_fdiv(g, HW, invHW);
          const int rem = g - ni * HW;
          const int h = ly_fdiv(rem, P.W, invW), w = rem - h * P.W;
          srow[n] = ((long)ni * (P.scat_ks * P.H) + P.scat_ks * h) * (P.scat_ks * P.W) + P.scat_ks * w;
        }
      }
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const int cc = 16 * ((by * WC + wc) * MT + t) + 4 * lq;
        long soff = 0;
        int scol = cc;
        if constexpr (FAST >= 3) {
          const int grp = cc / P.scat_c;
          scol = cc - grp * P.scat_c;
          const int ky = grp / P.scat_ks, kx = grp - ky * P.scat_ks;
          soff = (long)ky * (P.scat_ks * P.W) + kx;
        }
        // FAST == 4: + eadd (LyGemmParams.eadd: the gradient another consumer of the same tensor already produced) — the tile's NT loads
        // leave together before the first value is needed; lanes outside the tile read the scratch line they store to
        f32x4 ev[FAST == 4 ? NT : 1];
        if constexpr (FAST == 4) {
          const TO* const eadd = reinterpret_cast<const TO*>(P.eadd);
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const long gp = p0 + pixgrp + 16 * n + li;
            const bool ok = gp < P.M && cc < P.N;
            ev[n] = ly_ld4<TO>(ok ? eadd + (srow[n] + soff) * P.ldeadd + scol : reinterpret_cast<const TO*>(ly_gemm_trash) + 4 * lane);
          }
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const long gp = p0 + pixgrp + 16 * n + li;
          const float rs = PRO == LY_PRO_AFFINE_RELU_CA ? rsv[n] : 1.f;
          f32x4 u;
#pragma unroll
          for (int r = 0; r < 4; ++r) u[r] = acc[t][n][r] * rs * esc[t][r] + esh[t][r];
          const bool ok = gp < P.M && cc < P.N;
          if constexpr (FAST == 2) {                       // BatchNorm sums stay in registers until the block has walked all its tiles
            const f32x4 um = ok ? u : zero;
            sum1[t] += um;
            sum2[t] += um * um;
          }
          f32x4 v = ly_act4(u, act);
          if constexpr (FAST == 4) v += ev[n];
          long orow = gp;
          if constexpr (FAST >= 3) orow = srow[n] + soff;
          TO* o = ok ? out + orow * P.ldo + scol : reinterpret_cast<TO*>(ly_gemm_trash) + 4 * lane;
          ly_st4<TO>(o, v);
          acc[t][n] = zero;
        }
      }
      pt += nslots;
      if (pt >= gx) return false;
      p0 = (long)pt * BP;
      c = 0;
      return true;
    }
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
            if (stats) {                                   // pre-activation value (fp32, before any rounding) is what BatchNorm normalises
              st1 += u;
              st2 += u * u;
              if (!out) continue;                          // pure statistics pass; with `out` the value is stored as well
            }
            const f32x4 v = ly_act4(u, act);
            TO* o = out + gp * P.ldo + cc;
            if (vec_ok && cc + 3 < P.N) {
              ly_st4<TO>(o, v);
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (cc + r < P.N) ly_st1<TO>(o + r, v[r]);
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
  // the item loop is unrolled so that the register-set index (mod D) and, with resident weights, the chunk index (mod NCH) are static
  constexpr int NC1 = NCH > 0 ? NCH : 1;
  constexpr int U = (D % NC1 == 0) ? D : D * NC1;
  static_assert(U <= 6, "unroll");
#define LY_GEMM_ITEM(u) if constexpr (u < U) { if (!item(LyIc<(u) % D>(), LyIc<(u) % NC1>())) break; }
  while (true) {
    LY_GEMM_ITEM(0) LY_GEMM_ITEM(1) LY_GEMM_ITEM(2) LY_GEMM_ITEM(3) LY_GEMM_ITEM(4) LY_GEMM_ITEM(5)
  }
#undef LY_GEMM_ITEM
  if constexpr (FAST == 2) {
#pragma unroll
    for (int t = 0; t < MT; ++t) ly_stats_flush(stats, P.N, 16 * ((by * WC + wc) * MT + t) + 4 * lq, sum1[t], sum2[t]);
  }
}

// Two waves per SIMD (8 per CU) are what hides the 1-2 us of a load round trip behind another wave's MFMAs: the register
// allocator is told to fit 256 unified registers (left alone it takes up to ~290 for the gather variants and halves the occupancy:
// UP2 at 80x80x32 57 -> 78 us).
template <typename TI, typename TO, int NT, int MT, int WC, int GATHER, int PRO, int NCH = 0, int FAST = 0>
__global__ __launch_bounds__(LY_THREADS) __attribute__((amdgpu_waves_per_eu(LY_GEMM_W3(TI, PRO, NCH, MT) ? 3 : 2))) void ly_gemm_kernel_d2(const LyGemmParams P, const int gy, const int nslots, const int gx) {
  // with resident weights nothing in the loop is consumed in the item that loaded it: a third item in flight then really is in flight
  // (the third register set fits only next to bf16 rows without a prologue; everything else keeps two)
  constexpr int D = LY_GEMM_W3(TI, PRO, NCH, MT) ? LY_GEMM_W3_DEPTH : (NCH > 0 && PRO == LY_PRO_NONE && sizeof(TI) == 2) ? LY_GEMM_DEPTH : 2;
  ly_gemm_body2<TI, TO, NT, MT, WC, GATHER, PRO, D, NCH, FAST>(P, gy, nslots, gx);
}

template <typename TI, typename TO, int NT, int MT, int WC, int GATHER, int PRO, int NCH = 0, int FAST = 0>
static int launch_gemm_d2(const LyGemmParams& P, hipStream_t st) {
  constexpr int BP = 16 * NT * (4 / WC);
  constexpr int BN = 16 * MT * WC;
  constexpr int BK = 16 * LyT<TI>::VW;
  constexpr size_t lds = 2 * LyT<TI>::PL * (size_t)4 * ly_qps(BP * ly_qrs(BK / 32));
  long gx = (P.M + BP - 1) / BP;
  int gy = (P.N + BN - 1) / BN;
  LY_CHECK(gx < (1L << 30), "gemm: too many pixel tiles");
  auto k = ly_gemm_kernel_d2<TI, TO, NT, MT, WC, GATHER, PRO, NCH, FAST>;
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
  hipLaunchKernelGGL(k, dim3((unsigned)(nslots * gy)), dim3(LY_THREADS), lds, st, P, gy, (int)nslots, (int)gx);
  LY_LAUNCH_CHECK();
  return 0;
}

// Variant choice per call.  K in one or two chunks: weights resident (NCH) and, when every channel tile is full and the stores can be
// vectors, the branch-free epilogue (FAST 1; 2 = with the BatchNorm sums of the training forward kept in registers; 3 = FAST 1 with the
// scatter store of LyGemmParams.scat_ks; 4 = 3 + LyGemmParams.eadd added before the store).
template <typename TI, typename TO, int NT, int MT, int WC, int GATHER, int PRO>
static int launch_gemm_v(const LyGemmParams& P, hipStream_t st) {
  // which of the variants fit 256 registers without spilling into the loop (checked in the .s of every instantiation)
  constexpr bool ok1 = WC == 4 || PRO == LY_PRO_NONE;                  // one chunk resident
  constexpr bool ok2 = WC == 4 && PRO != LY_PRO_GATE;                  // two chunks resident
  constexpr bool oks1 = ok1 && PRO != LY_PRO_GATE;                     // ... with the BatchNorm sums in registers as well
  constexpr bool oks2 = ok2 && PRO == LY_PRO_NONE;
  const int nchunk = (P.K + 16 * LyT<TI>::VW - 1) / (16 * LyT<TI>::VW);
  const bool fast = (P.N & 3) == 0 && (P.ldo & 3) == 0 && P.out;
  if (P.scat_ks) {                                          // scatter store (ly_gemm_fwd checked: plain rows, no prologue, no statistics, fast widths)
    if constexpr (GATHER == LY_GATHER_ROWS && PRO == LY_PRO_NONE) {
      if (P.eadd) {
        if constexpr (ok1) { if (nchunk == 1) return launch_gemm_d2<TI, TO, NT, MT, WC, GATHER, PRO, 1, 4>(P, st); }
        if constexpr (ok2) { if (nchunk == 2) return launch_gemm_d2<TI, TO, NT, MT, WC, GATHER, PRO, 2, 4>(P, st); }
        return launch_gemm_d2<TI, TO, NT, MT, WC, GATHER, PRO, 0, 4>(P, st);
      }
      if constexpr (ok1) { if (nchunk == 1) return launch_gemm_d2<TI, TO, NT, MT, WC, GATHER, PRO, 1, 3>(P, st); }
      if constexpr (ok2) { if (nchunk == 2) return launch_gemm_d2<TI, TO, NT, MT, WC, GATHER, PRO, 2, 3>(P, st); }
      return launch_gemm_d2<TI, TO, NT, MT, WC, GATHER, PRO, 0, 3>(P, st);
    } else {
      ly_set_error("gemm: the scatter store is built for plain-row sources without a prologue");
      return -1;
    }
  }
  if (fast && !P.stats) {
    if constexpr (ok1) { if (nchunk == 1) return launch_gemm_d2<TI, TO, NT, MT, WC, GATHER, PRO, 1, 1>(P, st); }
    if constexpr (ok2) { if (nchunk == 2) return launch_gemm_d2<TI, TO, NT, MT, WC, GATHER, PRO, 2, 1>(P, st); }
  } else if (fast) {
    if constexpr (oks1) { if (nchunk == 1) return launch_gemm_d2<TI, TO, NT, MT, WC, GATHER, PRO, 1, 2>(P, st); }
    if constexpr (oks2) { if (nchunk == 2) return launch_gemm_d2<TI, TO, NT, MT, WC, GATHER, PRO, 2, 2>(P, st); }
  }
  if constexpr (PRO == LY_PRO_GATE) {                      // CoordAtt-gate GEMM with K in more than one chunk (eval C3_CA.cv3): the branch-free epilogue
    if (fast && !P.stats) return launch_gemm_d2<TI, TO, NT, MT, WC, GATHER, PRO, 0, 1>(P, st);
  }
  if constexpr (PRO == LY_PRO_NONE) {                      // long K: weights streamed one item ahead, still the branch-free epilogue
    if (fast) return P.stats ? launch_gemm_d2<TI, TO, NT, MT, WC, GATHER, PRO, 0, 2>(P, st) : launch_gemm_d2<TI, TO, NT, MT, WC, GATHER, PRO, 0, 1>(P, st);
  }
  return launch_gemm_d2<TI, TO, NT, MT, WC, GATHER, PRO>(P, st);
}

template <typename T, int NT, int MT, int WC>
static int launch_gemm(const LyGemmParams& P, hipStream_t st) {
  if (P.gather == LY_GATHER_PATCH) return launch_gemm_v<T, T, NT, MT, WC, LY_GATHER_PATCH, LY_PRO_NONE>(P, st);
  if (P.gather == LY_GATHER_PATCH_NCHW) return launch_gemm_v<float, T, NT, MT, WC, LY_GATHER_PATCH_NCHW, LY_PRO_NONE>(P, st);
  if (P.gather == LY_GATHER_PATCH_NCHW_U8) return launch_gemm_v<unsigned char, T, NT, MT, WC, LY_GATHER_PATCH_NCHW, LY_PRO_NONE>(P, st);
  if (P.gather == LY_GATHER_PATCH_NCHW_BF16 || P.gather == LY_GATHER_PATCH_NCHW_F16) {
    if constexpr (LyT<T>::BF) {
      if (P.gather == LY_GATHER_PATCH_NCHW_BF16) return launch_gemm_v<ly_bf16img, T, NT, MT, WC, LY_GATHER_PATCH_NCHW, LY_PRO_NONE>(P, st);
      return launch_gemm_v<ly_f16img, T, NT, MT, WC, LY_GATHER_PATCH_NCHW, LY_PRO_NONE>(P, st);
    } else {
      ly_set_error("gemm: a 16-bit image source is built for LY_BF16 calls only");
      return -1;
    }
  }
  if (P.gather == LY_GATHER_UP2) {
    if (P.pro == LY_PRO_NONE) return launch_gemm_v<T, T, NT, MT, WC, LY_GATHER_UP2, LY_PRO_NONE>(P, st);
    ly_set_error("gemm: upsampled source with a prologue is not built");
    return -1;
  }
  if (P.pro == LY_PRO_GATE) return launch_gemm_v<T, T, NT, MT, WC, LY_GATHER_ROWS, LY_PRO_GATE>(P, st);
  if (P.pro == LY_PRO_AFFINE_RELU_CA) return launch_gemm_v<T, T, NT, MT, WC, LY_GATHER_ROWS, LY_PRO_AFFINE_RELU_CA>(P, st);
  return launch_gemm_v<T, T, NT, MT, WC, LY_GATHER_ROWS, LY_PRO_NONE>(P, st);
}

// Tile policy, measured on MI355X at every LEAD-YOLO shape: 64-pixel tiles with 3 co-resident blocks per CU beat 128-pixel
// tiles (one wave per SIMD); the narrow-output tile trades channels for pixels.  (bf16 storage, whole train step: 64 x 64 tiles for
// N > 64 — four waves per SIMD instead of two — 19.95 vs 19.79 ms: no gain, the 64 x 128 tile stays.)
template <typename T>
static int ly_gemm_dispatch(const LyGemmParams& P, hipStream_t st) {
  if (P.N > 64) return launch_gemm<T, 4, 2, 4>(P, st);    // 64 px x 128 ch per block
  if (P.N > 32) return launch_gemm<T, 4, 1, 4>(P, st);    // 64 px x 64 ch
  // (the CoordAtt-gate prologue on the 128 px x 32 ch tile compiles to 256 registers with 130-320 spilled — tools/kernel_regs.py; lead-yolo-n's
  // 32-channel C3_CA.cv3 takes the 64 x 64 tile with half its channel tiles idle instead)
  if (P.pro == LY_PRO_GATE) return launch_gemm<T, 4, 1, 4>(P, st);
  return launch_gemm<T, 2, 2, 1>(P, st);                  // 128 px x 32 ch
}
int ly_gemm_dispatch_f32(const LyGemmParams& P, hipStream_t st);
int ly_gemm_dispatch_bf16(const LyGemmParams& P, hipStream_t st);
int ly_patch4_try(const LyGemmParams& P, hipStream_t st);      // ly_patch4.hip: 1 = launched
