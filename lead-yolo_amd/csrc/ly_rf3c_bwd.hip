// RFCBAMConv kernel_size 3 backward on the lane = channel core (ly_rf3c.hpp): reference models/rfa.py:113-129 under autograd.
//
// The first-generation backward (ly_rfcbam_bwd.hip) materialises the 9x expanded tensors ug, dcd, cd, dv in HBM and streams them
// thirteen times (236 MB each at layer 17, bs = 64).  Here NOTHING 9x-sized exists in HBM: every pass re-derives what it needs on chip
// from x (the saved input) and du (the gradient of the conv's pre-BatchNorm output), per unit = (64-pixel tile, 32-channel chunk):
//     dcd[p, t, c] = sum_o Wc[o, c, t] * du[p, o]          on the MFMAs (du tile x Wc^T fragments) into a K-major LDS tile
//     u, v, G      = generate / BatchNorm / ReLU            on the VALU, lane = channel, bit-identical to the training forward
// Three global dependencies separate the passes (the channel sums behind rfa's gradient, the batch sums of the generate BatchNorm):
//   pass A: d_rfa[p, t] = sum_c dcd*G*ca (one slab per channel chunk, summed by ly_rf3c_rfa_bwd),  d_ca[n, c] = sum_{p,t} dcd*G*rfa
//   (ly_rf3c_rfa_bwd: d_mm and d(get_weight) from d_rfa through sigmoid + 3x3 conv)
//   pass B: dv = [G>0] * (dcd*ca*rfa + d_mean/C + [G == max_c G] * d_max);  BatchNorm sums s1 = sum dv, s2 = sum dv*u (one stripe per image)
//   (ly_bn_bwd_coeffs: alpha, kappa, lambda)
//   pass C: du_g = alpha*dv + kappa + lambda*u;  d(generate weight)[c][t][u'] = sum_p du_g[t]*x_u' (one row per image);
//           dx[q] = sum_{(p, u') reading q} sum_t w[t][u']*du_g[p][t]  accumulated in an fp32 LDS tile whose seam row / column is CARRIED to the
//           neighbouring tile on chip: a block owns one (image, chunk) and walks the image's tiles in raster order, so every dx element
//           is completed by one block and written once, final (+ the SE term d/d(mean x)).
//   ly_rf3c_wgrad: d(conv.0.weight)[o, c, t] = sum_p du[p, o] * (G*ca*rfa)[p, t, c]  (G' regenerated as in the forward, contraction over pixels)
// bf16 storage, stride 2, C % 32 == 0, O in {64, 128, 256}; everything else stays on the first-generation kernels.
#define RC_ASM_FMA          // see ly_rf3c.hpp: safe here (one wave per SIMD, the block owns its CU)
#include "ly_rf3c.hpp"
#include "ly_params.h"
#include <stdlib.h>

enum { RB_A = 0, RB_B = 1, RB_C = 2 };

#ifdef RC_PHASE_PROF                    // development: cycles per phase of pass C, block 0 / wave 0 (make CXXFLAGS+=-DRC_PHASE_PROF; tools read ly_rf3c_prof)
__device__ unsigned long long rc_prof[8];
// (accumulated in scalar registers and written once at the end: a global read-modify-write per mark would put an `s_waitcnt vmcnt(0)` there
// and drain the very loads whose latency is being looked at)
#ifndef RC_PROF_MODE
#define RC_PROF_MODE RB_C
#endif
#define RC_T(i) do { if (MODE == RC_PROF_MODE) { const unsigned long long t_ = __builtin_readcyclecounter(); rc_acc[i] += t_ - rc_t0; rc_t0 = t_; } } while (0)
extern "C" int ly_rf3c_prof(unsigned long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(rc_prof), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rc_prof), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#else
#define RC_T(i) do { } while (0)
#endif

template <int MODE, int KS>
__global__ __launch_bounds__(LY_THREADS) void ly_rf3c_bwd_kernel(const LyRf3cBwdParams P, const int nct, const int nrt) {
  typedef __bf16 T;
  constexpr int S = 2;
  constexpr int O = 32 * KS;
  constexpr int RSD = 2 * O + 16;                       // bytes per du-tile row: RSD/16 odd => the b64 fragment reads are conflict-free (ly_tile.hpp)
  constexpr int NDU = KS;                               // 16-byte du items per thread: 64 px * (O/8) / 256
  constexpr int TABW = MODE == RB_A ? 2 : 8;            // floats per (pixel pair, tap): A: rfa | B, C: rfa, max_c G, d_max, d_mean/C  (x 2 pixels)
  extern __shared__ f32x4 rc_smem4[];
  const RcGeom g = rc_geom(S, P.TH, P.TW);
  const int IHW = g.IH * g.IW;
  float* xs = reinterpret_cast<float*>(rc_smem4);                     // [IHW][32] fp32
  char* dt = reinterpret_cast<char*>(xs + IHW * RC_CB);              // [288][128 B] bf16, K-major, swizzled
  char* dus = dt + RC_KR * 128;                                      // [64][RSD]
  float* tab = reinterpret_cast<float*>(dus + RC_TP * RSD);          // [32 pairs][9][TABW]
  float* act = tab + 32 * 9 * TABW;                                  // [32 pairs][2]: 1 for pixels inside the map (pass C)
  float* dxs = act + 64;                                             // pass C: [IHW][32] fp32 | right carry [IH][32] | bottom carry [2*TW*nct + 2][32]
  float* rcar = dxs + IHW * RC_CB;
  float* bcar = rcar + g.IH * RC_CB;
  float* cfs = bcar + (S * P.TW * nct + 2) * RC_CB;                  // pass C: [27][32] alpha, kappa, lambda of the chunk's channels
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, c = lane & 31;
  const int li = lane & 15, lq = lane >> 4;
  const int NCH = P.C / RC_CB;
  const int bid = ly_xcd_remap((int)blockIdx.x, (int)gridDim.x);            // the channel chunks of an image (same du tiles) on one XCD's L2
  const int chunk = bid % NCH, n = bid / NCH;
  const int c0 = chunk * RC_CB;
  const T* const x = reinterpret_cast<const T*>(P.x);
  const T* const du = reinterpret_cast<const T*>(P.du);
  const int HK = 3 * P.Ho, WK = 3 * P.Wo;
  const float invC = 1.f / (float)P.C;
  // dx output pass (pass C): IW*4 threads per tile row, dxo_nrs rows at a time
  const int dxo_per = g.IW * 4, dxo_nrs = LY_THREADS / dxo_per > 0 ? LY_THREADS / dxo_per : 1;
  const int dxo_rsub = tid / dxo_per, dxo_q = (tid - dxo_rsub * dxo_per) >> 2;

  constexpr bool ROLLED = MODE == RB_C || KS > 4;           // the pair loop of the VALU phase is not unrolled (register file)
  // ---- per-lane constants of the VALU phase ----------------------------------------------------------
  const int stream = wave * 2 + half;                       // pixel pairs px0 = 8*stream + 2*j, j < 4
  const int row = g.IW * RC_CB;
  const int csw = rc_sw(c);
  const float* xp[4];
  int goff[4], pos0[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int px0 = 8 * stream + 2 * j;
    pos0[j] = rc_pos0(g, px0);
    xp[j] = xs + pos0[j] * RC_CB + c;
    goff[j] = c * 128 + ((((px0 >> 2) ^ csw) << 3) | ((px0 & 2) << 1));
  }
  RcW w;
  rc_load_w<true>(w, P.wq, c0, c);
  const float cav = P.ca[(long)n * P.C + c0 + c];
  const f32x2 cav2 = {cav, cav};

  // pass state
  f32x2 dca2 = {0.f, 0.f};                                  // A
  f32x2 s1[9], s2[9];                                       // B
  float dwd[MODE == RB_C ? 81 : 1];                         // C: scalar accumulators (81 pairs + 100 weight registers + the rest exceed the file)
  if constexpr (MODE == RB_B) {
#pragma unroll
    for (int t = 0; t < 9; ++t) { s1[t] = (f32x2){0.f, 0.f}; s2[t] = (f32x2){0.f, 0.f}; }
  }
  if constexpr (MODE == RB_C) {
#pragma unroll
    for (int i = 0; i < 81; ++i) dwd[i] = 0.f;
    // the BatchNorm-backward coefficients of the lane's channel stay in LDS (27 more registers on top of 81 + 100 did not fit): a per-lane,
    // conflict-free read per use
    for (int i = tid; i < 27 * RC_CB; i += LY_THREADS) cfs[i] = P.coef[(long)(i / RC_CB) * P.C + c0 + (i % RC_CB)];
    // dx tile and carries start at zero
    for (int i = tid; i < (IHW + g.IH + S * P.TW * nct + 2) * RC_CB; i += LY_THREADS) dxs[i] = 0.f;
  }

  // ---- staging: x tile, du tile, per-pixel tables of the NEXT tile are requested while the current one is processed --------------
  // The per-thread item -> (row, column) maps are computed ONCE (the integer divisions of a per-tile plan were ~2 k cycles of every unit).
  RcStage<T> St;
  RcStageFix<T> Sf;
  rc_stage_fix(Sf, St, g, tid, P.W, P.ldx);
  ly_u32x4 dpv[NDU];
  bool dok[NDU];
  int dyx[NDU];                                        // du item e: pixel (ly << 16 | lx), -1: pixel slot past the tile
#pragma unroll
  for (int e = 0; e < NDU; ++e) {
    const int px = (tid + e * LY_THREADS) / (O / 8);
    const int ly = px / P.TW, lx = px - ly * P.TW;
    dyx[e] = px < g.NPX ? (ly << 16) | lx : -1;
  }
  float tv[3][MODE == RB_A ? 1 : 4];
  bool tok[3];
  int tyx[3], trel[3];                                 // table item e: pixel (ly << 16 | lx) (-1: none), map offset (3*ly + ty)*WK + 3*lx + tx
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    const int idx = tid + e * LY_THREADS;
    const int px = idx / 9, t = idx - px * 9;
    const int ly = px / P.TW, lx = px - ly * P.TW;
    tyx[e] = (idx < RC_TP * 9 && px < g.NPX) ? (ly << 16) | lx : -1;
    trel[e] = (3 * ly + t / 3) * WK + 3 * lx + t % 3;
  }
  auto issue = [&](int tt) {
    const int ct = tt % nct, rt = tt / nct;
    const int oy0 = rt * P.TH, ox0 = ct * P.TW;
    rc_stage_retarget(St, Sf, ((n * P.H + S * oy0 - 1) * P.W + S * ox0 - 1) * P.ldx, S * oy0 - 1, S * ox0 - 1, P.H, P.W);
    rc_stage_load(St, x, c0);
    const int mbase = (n * P.Ho + oy0) * P.Wo + ox0;
#pragma unroll
    for (int e = 0; e < NDU; ++e) {
      const int ly = dyx[e] >> 16, lx = dyx[e] & 0xffff;
      dok[e] = dyx[e] >= 0 && oy0 + ly < P.Ho && ox0 + lx < P.Wo;
      const int m = dok[e] ? mbase + ly * P.Wo + lx : 0;
      dpv[e] = *reinterpret_cast<const ly_u32x4*>(du + (long)m * P.lddu + 8 * ((tid + e * LY_THREADS) % (O / 8)));
    }
    const int pbase = (n * HK + 3 * oy0) * WK + 3 * ox0;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      tok[e] = tyx[e] >= 0 && oy0 + (tyx[e] >> 16) < P.Ho && ox0 + (tyx[e] & 0xffff) < P.Wo;
      const int pos = tok[e] ? pbase + trel[e] : 0;
      tv[e][0] = P.rfa[pos];
      if constexpr (MODE != RB_A) {
        tv[e][1] = P.mm[2 * pos];
        const f32x2 d = *reinterpret_cast<const f32x2*>(P.d_mm + 2 * pos);
        tv[e][2] = d[0];
        tv[e][3] = d[1];                                   // (scaled by 1/C at commit: nothing here may wait for a load)
      }
    }
  };
  auto commit = [&]() {
    rc_stage_store(St, xs);
#pragma unroll
    for (int e = 0; e < NDU; ++e) {
      const int idx = tid + e * LY_THREADS;
      const int px = idx / (O / 8), v8 = idx - px * (O / 8);
      *reinterpret_cast<ly_u32x4*>(dus + px * RSD + 16 * v8) = dok[e] ? dpv[e] : (ly_u32x4){0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      const int idx = tid + e * LY_THREADS;
      const int px = idx / 9, t = idx - px * 9;
      if (idx < RC_TP * 9) {
        float* d = tab + ((px >> 1) * 9 + t) * TABW + (px & 1);
#pragma unroll
        for (int q = 0; q < (MODE == RB_A ? 1 : 4); ++q) d[2 * q] = tok[e] ? (q == 3 ? tv[e][q] * invC : tv[e][q]) : 0.f;
        if (MODE == RB_C && t == 0) act[px] = tok[e] ? 1.f : 0.f;
      }
    }
  };

  // dcd contraction: wave w owns the 16-column tiles e_i = w + 4i (i < 5; the fifth exists for waves 0 and 1 only) of the 18.  Its Wc^T
  // fragments are the same for every unit (the chunk is fixed): a register ring runs one k-step (5 fragments) ahead, ACROSS units.
  constexpr int NE = 5;
  int ftile[NE];
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int e = wave + 4 * i < 18 ? wave + 4 * i : wave;            // (waves 2, 3: the fifth tile repeats the first, its product is dropped)
    ftile[i] = (((e >> 1) * (P.C / 16) + chunk * 2 + (e & 1)) * KS) * 64 + lane;
  }
  const uint4* wpk = reinterpret_cast<const uint4*>(P.wct);
  bf16x8 ring[NE];
#pragma unroll
  for (int i = 0; i < NE; ++i) ring[i] = __builtin_bit_cast(bf16x8, wpk[ftile[i]]);
  const int ntile = nct * nrt;
  issue(0);
#ifdef RC_PHASE_PROF
  unsigned long long rc_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long rc_t0 = __builtin_readcyclecounter();
#endif
  for (int tt = 0; tt < ntile; ++tt) {
    const int ct = tt % nct, rt = tt / nct;
    const int oy0 = rt * P.TH, ox0 = ct * P.TW;
    __syncthreads();                                   // the previous tile is done with every LDS region
    RC_T(0);
    commit();
    RC_T(5);
    issue(tt + 1 < ntile ? tt + 1 : tt);               // unconditional (the last tile re-requests itself): no load under a branch
    RC_T(6);
    __syncthreads();
    RC_T(1);

    // ---- dcd tile on the MFMAs: D[px][(t, c)] = du[px][:] . Wc^T[:, (t, c)] ----
    // per k-step the four du fragments are read ONCE and meet the wave's five column tiles (20 accumulator quads); the fragment used last is
    // replaced by the one of the next k-step (of the next unit after the last) right after its MFMAs
    // (O = 256: the five column tiles go in two groups of 3 + 2 — 20 accumulator quads on top of 8 du items in flight did not fit the file)
    {
      constexpr int NG = KS > 4 ? 2 : 1, GM = KS > 4 ? 3 : NE;
#pragma unroll
      for (int grp = 0; grp < NG; ++grp) {
        const int i0 = grp * GM;
        f32x4 acc[GM][4];
#pragma unroll
        for (int i = 0; i < GM; ++i)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) acc[i][mt] = ly_zero4();
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          bf16x8 af[4];
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) af[mt] = ly_lds_frag(dus, (16 * mt + li) * RSD, s, lq);
#pragma unroll
          for (int i = 0; i < GM; ++i) {
            if (i0 + i < NE) {
#pragma unroll
              for (int mt = 0; mt < 4; ++mt) acc[i][mt] = ly_mfma_bf16(af[mt], ring[i0 + i], acc[i][mt]);
              ring[i0 + i] = __builtin_bit_cast(bf16x8, wpk[ftile[i0 + i] + ((s + 1 < KS ? s + 1 : 0) * 64)]);
              __builtin_amdgcn_sched_barrier(0x786);      // neither loads nor MFMAs may move across: the refills stay a k-step ahead
            }
          }
        }
#pragma unroll
        for (int i = 0; i < GM; ++i) {
          const int e = wave + 4 * (i0 + i);
          if (i0 + i < NE && e < 18) {
            const int kk = 16 * (e & 1) + li;
            char* drow = dt + ((e >> 1) * RC_CB + kk) * 128;
            const int sw = rc_sw(kk);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<bf16x4*>(drow + (((4 * mt + lq) ^ sw) << 3)) = ly_cvtb4(acc[i][mt]);
          }
        }
      }
    }
    __syncthreads();
    RC_T(2);

    // ---- VALU phase, lane = channel -------------------------------------------------------------------
    // (ROLLED: pass C, and every pass at O = 256 — there the unrolled body's registers spilled.)
    // Pass C keeps 81 + 100 registers across the loop: its pair iterations are NOT unrolled and the per-pair addresses are recomputed.  Its pairs
    // are also walked in FOUR COLOURS (output-row parity x pair-column parity): the 3x5 input patches of two pairs of one colour never overlap,
    // so the dx tile is updated with plain read-add-write and a barrier between colours.  (ds_add_f32 is no alternative: measured ~640 cycles
    // per wave instruction — the LDS float atomic is serialised over the lanes — and pass C sat 50 % in SQ_WAIT_INST_LDS.)
#pragma unroll 1
    for (int col = 0; col < (MODE == RB_C ? 4 : 1); ++col) {
    int nit = 4, npx = 1, cnt = 0;
    const int ry = col >> 1, rx = col & 1;
    if constexpr (MODE == RB_C) {
      npx = ((P.TW >> 1) + 1 - rx) >> 1;
      cnt = ((P.TH + 1 - ry) >> 1) * npx;
      nit = (cnt + 7) >> 3;
      if (col > 0) __syncthreads();
    }
#pragma unroll(ROLLED ? 1 : 4)
    for (int j = 0; j < nit; ++j) {
      f32x2 xv[9], u[9], v[9];
      const float* xpj;
      int goffj, pos0j, pair;
      if constexpr (ROLLED) {
        int px0 = 8 * stream + 2 * j;
        if constexpr (MODE == RB_C) {
          const int idx = j * 8 + stream;
          if (idx >= cnt) continue;
          const int ia = idx / npx, ib = idx - ia * npx;
          px0 = (2 * ia + ry) * P.TW + 2 * (2 * ib + rx);
        }
        pos0j = rc_pos0(g, px0);
        xpj = xs + pos0j * RC_CB + c;
        goffj = c * 128 + ((((px0 >> 2) ^ csw) << 3) | ((px0 & 2) << 1));
        pair = px0 >> 1;
      } else {
        pos0j = pos0[j]; xpj = xp[j]; goffj = goff[j];
        pair = 4 * stream + j;
      }
      rc_patch<S>(xpj, row, xv);
      rc_generate<true>(w, xv, u);
      rc_affine(w, u, v);
      if constexpr (MODE == RB_A) {
        const f32x2* rfp = reinterpret_cast<const f32x2*>(tab) + pair * 9;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          unsigned* slot = reinterpret_cast<unsigned*>(dt + t * (RC_CB * 128) + goffj);
          const f32x2 q = rc_unpack2(*slot) * (f32x2){rc_relu(v[t][0]), rc_relu(v[t][1])};
          dca2 += q * rfp[t];
          *slot = rc_pack2(q * cav2);                    // z = dcd*G*ca: summed over the channels below
        }
      } else {
        const f32x4* tb = reinterpret_cast<const f32x4*>(tab) + pair * 18;
        f32x2 dv[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const f32x4 ta = tb[2 * t], tc = tb[2 * t + 1];          // (rfa0, rfa1, gmax0, gmax1), (dmax0, dmax1, dmean0/C, dmean1/C)
          const f32x2 dc = rc_unpack2(*reinterpret_cast<const unsigned*>(dt + t * (RC_CB * 128) + goffj));
          const f32x2 G = {rc_relu(v[t][0]), rc_relu(v[t][1])};
          f32x2 dG = dc * ((f32x2){ta[0], ta[1]} * cav2) + (f32x2){tc[2], tc[3]};
          dG[0] += G[0] == ta[2] ? tc[0] : 0.f;
          dG[1] += G[1] == ta[3] ? tc[1] : 0.f;
          const f32x2 dvv = {G[0] > 0.f ? dG[0] : 0.f, G[1] > 0.f ? dG[1] : 0.f};
          if constexpr (MODE == RB_B) {
            s1[t] += dvv;
            s2[t] += dvv * u[t];
          } else {
            // du_g = alpha*dv + kappa + lambda*u, zero for pixel slots outside the map
            const float al = cfs[t * RC_CB + c], ka = cfs[(9 + t) * RC_CB + c], la = cfs[(18 + t) * RC_CB + c];
            const f32x2 d = dvv * (f32x2){al, al} + (f32x2){ka, ka} + u[t] * (f32x2){la, la};
            dv[t] = d * *reinterpret_cast<const f32x2*>(act + 2 * pair);
          }
        }
        if constexpr (MODE == RB_C) {
          // generate weight gradient: both pixels of the pair into the halves of 81 accumulators
#pragma unroll
          for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int uu = 0; uu < 9; ++uu) dwd[t * 9 + uu] += dv[t][0] * xv[uu][0] + dv[t][1] * xv[uu][1];
          // data gradient of the patch: dxc[u'] = sum_t w[t][u'] * du_g[t]
          f32x2 dxc[9];
#pragma unroll
          for (int uu = 0; uu < 9; ++uu) dxc[uu] = rc_pkmul(dv[0], w.p[uu >> 1], uu & 1);
#pragma unroll
          for (int t = 1; t < 9; ++t)
#pragma unroll
            for (int uu = 0; uu < 9; ++uu) dxc[uu] = rc_pkfma(dv[t], w.p[(t * 9 + uu) >> 1], dxc[uu], (t * 9 + uu) & 1);
          // the pair's 3 x 5 input patch (the middle column belongs to both pixels): read-add-write, no other stream touches it in this colour
          // ALL fifteen reads first, then the adds, then the writes: written as fifteen `+=` the compiler must keep every later read behind
          // the earlier writes (they may alias for all it knows) — a chain of 14 exposed LDS round trips per pair at one wave per SIMD
          float* dp = dxs + pos0j * RC_CB + c;
          float old[15];
#pragma unroll
          for (int uy = 0; uy < 3; ++uy)
#pragma unroll
            for (int k = 0; k < 5; ++k) old[5 * uy + k] = dp[uy * row + k * RC_CB];
#pragma unroll
          for (int uy = 0; uy < 3; ++uy) {
            float* dr = dp + uy * row;
            const float e0 = dxc[uy * 3][0], e1 = dxc[uy * 3 + 1][0], e2 = dxc[uy * 3 + 2][0] + dxc[uy * 3][1], e3 = dxc[uy * 3 + 1][1], e4 = dxc[uy * 3 + 2][1];
            dr[0] = old[5 * uy] + e0; dr[RC_CB] = old[5 * uy + 1] + e1; dr[2 * RC_CB] = old[5 * uy + 2] + e2;
            dr[3 * RC_CB] = old[5 * uy + 3] + e3; dr[4 * RC_CB] = old[5 * uy + 4] + e4;
          }
        }
      }
    }
    }

    if constexpr (MODE == RB_A) {
      // ---- d_rfa of this chunk: lane = pixel sums z over the chunk's 32 channels, wave w takes taps w, w+4, (w+8) ----
      __syncthreads();
      RC_T(3);
      // thread = (tap t, 4-pixel chunk pq): 32 independent 8-byte reads (the chunk of row t*32 + cc), four running sums.  (Until late round 3
      // a lane owned ONE pixel and walked 32 rows x 3 taps with 2-byte reads: 96 reads per lane, a quarter of pass A by the phase counters.)
      if (tid < 9 * 16) {
        const int t = tid >> 4, pq = tid & 15;
        const char* base = dt + t * (RC_CB * 128);
        f32x4 sum = ly_zero4();
#pragma unroll
        for (int cc = 0; cc < RC_CB; ++cc) {
          const uint2 hv = *reinterpret_cast<const uint2*>(base + cc * 128 + ((pq ^ rc_sw(cc)) << 3));
          sum[0] += __builtin_bit_cast(float, hv.x << 16);
          sum[1] += __builtin_bit_cast(float, hv.x & 0xffff0000u);
          sum[2] += __builtin_bit_cast(float, hv.y << 16);
          sum[3] += __builtin_bit_cast(float, hv.y & 0xffff0000u);
        }
        const long pbase = (long)chunk * ((long)P.n_img * HK * WK) + ((long)n * HK + t / 3) * WK + t % 3;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int px = 4 * pq + k;
          const int ly = px / P.TW, lx = px - ly * P.TW;
          const int oy = oy0 + ly, ox = ox0 + lx;
          if (px < g.NPX && oy < P.Ho && ox < P.Wo) P.d_rfa_part[pbase + (long)(3 * oy) * WK + 3 * ox] = sum[k];
        }
      }
      RC_T(4);
    }
    if constexpr (MODE == RB_B) RC_T(3);
    if constexpr (MODE == RB_C) {
      // ---- dx: rows / columns this tile completes leave as T (+ the SE term); its last row / column is carried to the neighbours ----
      __syncthreads();
    RC_T(3);
      const int iy0 = S * oy0 - 1, ix0 = S * ox0 - 1;
      const int RL = g.IH - 1, CL = g.IW - 1;            // the seam row / column (shared with the tile below / to the right)
      const bool lastr = rt == nrt - 1, lastc = ct == nct - 1;
      T* const dxo = reinterpret_cast<T*>(P.dx);
      // thread = (row group, column q, 8-channel group v8): IW*4 threads per row, LY_THREADS / (IW*4) rows at a time (the division is done
      // once per kernel).  (Until late round 3 only the IW*4 = 68 threads of ONE row group worked here, each walking all 17 rows: this
      // pass was 19.5 % of the kernel by the phase counters, RC_PHASE_PROF.)
      {
        const int v8 = tid & 3;
        f32x4 dga = ly_zero4(), dgb = ly_zero4();              // the SE term of the thread's 8 channels
        if (P.dgap) {
          const float* dg = P.dgap + (long)n * P.C + c0 + 8 * v8;
          dga = *reinterpret_cast<const f32x4*>(dg) * P.dgap_scale;
          dgb = *reinterpret_cast<const f32x4*>(dg + 4) * P.dgap_scale;
        }
        if (dxo_rsub < dxo_nrs) {
          const int q = dxo_q;
          const int ix = ix0 + q;
          const bool colfin = (q < CL || lastc) && ix >= 0 && ix < P.W;
#pragma unroll 1                       // (two rows in flight: 5 spilled registers, 367 -> 377 us)
          for (int r = dxo_rsub; r < g.IH; r += dxo_nrs) {
            const int iy = iy0 + r;
            const bool fin = colfin && (r < RL || lastr) && iy >= 0 && iy < P.H;
            float* src = dxs + (r * g.IW + q) * RC_CB + 8 * v8;
            f32x4 a = *reinterpret_cast<f32x4*>(src), b = *reinterpret_cast<f32x4*>(src + 4);
            *reinterpret_cast<f32x4*>(src) = ly_zero4();            // the next tile starts from zero (+ the carries, below)
            *reinterpret_cast<f32x4*>(src + 4) = ly_zero4();
            if (fin) {
              const f32x4 qq[2] = {a + dga, b + dgb};
              const int off = ((n * P.H + iy) * P.W + ix) * P.lddx + c0 + 8 * v8;       // < 2^31 elements (checked by the launcher)
              *reinterpret_cast<ly_u32x4*>(dxo + off) = ly_rv_pack(qq, (ly_u32x4*)nullptr);
            }
            // carries: the seam column (rows above the seam row; all rows in the last tile row) goes right, the seam row (all columns) goes
            // down, ADDED to what is there at its first column: that element also received the left neighbour's corner
            if (q == CL && (r < RL || lastr) && !lastc) {
              *reinterpret_cast<f32x4*>(rcar + r * RC_CB + 8 * v8) = a;
              *reinterpret_cast<f32x4*>(rcar + r * RC_CB + 8 * v8 + 4) = b;
            }
            if (r == RL && !lastr) {
              float* bc = bcar + (ix + 1) * RC_CB + 8 * v8;
              if (q == 0 && ct > 0) {
                a += *reinterpret_cast<f32x4*>(bc);
                b += *reinterpret_cast<f32x4*>(bc + 4);
              }
              *reinterpret_cast<f32x4*>(bc) = a;
              *reinterpret_cast<f32x4*>(bc + 4) = b;
            }
          }
        }
      }
      __syncthreads();
    RC_T(4);
      // next tile (raster order): column 0 (rows above the seam row) from the right carry, row 0 from the bottom carry of the tile row above
      // (column 0 of row 0 came in through the right carry when there is a left neighbour); everything else was zeroed above
      {
        const int tn = tt + 1;
        const int ctn = tn % nct, rtn = tn / nct;
        const bool lastrn = rtn == nrt - 1;
        const int ix0n = S * (ctn * P.TW) - 1;
        const int v4 = tid & 7;
        if (ctn > 0)
          for (int r = tid >> 3; r < g.IH; r += LY_THREADS / 8)
            if (r < RL || lastrn) *reinterpret_cast<f32x4*>(dxs + (r * g.IW) * RC_CB + 4 * v4) = *reinterpret_cast<f32x4*>(rcar + r * RC_CB + 4 * v4);
        if (rtn > 0)
          for (int q = (tid >> 3) + (ctn > 0 ? 1 : 0); q < g.IW; q += LY_THREADS / 8)
            *reinterpret_cast<f32x4*>(dxs + q * RC_CB + 4 * v4) = *reinterpret_cast<f32x4*>(bcar + (ix0n + q + 1) * RC_CB + 4 * v4);
      }
    }
  }

#ifdef RC_PHASE_PROF
  if (MODE == RC_PROF_MODE && blockIdx.x == 0 && threadIdx.x == 0)
    for (int i = 0; i < 8; ++i) rc_prof[i] += rc_acc[i];
#endif
  // ---- flush the per-image accumulators ---------------------------------------------------------------
  if constexpr (MODE == RB_A) {
    float d = dca2[0] + dca2[1];
    d += __shfl_xor(d, 32);
    __syncthreads();
    float* red = reinterpret_cast<float*>(dt);
    if (half == 0) red[wave * RC_CB + c] = d;
    __syncthreads();
    if (tid < RC_CB) P.d_ca[(long)n * P.C + c0 + tid] = (red[tid] + red[RC_CB + tid]) + (red[2 * RC_CB + tid] + red[3 * RC_CB + tid]);
  }
  if constexpr (MODE == RB_B) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(dt);          // [4 waves][18][32]
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float a = s1[t][0] + s1[t][1], b = s2[t][0] + s2[t][1];
      a += __shfl_xor(a, 32);
      b += __shfl_xor(b, 32);
      if (half == 0) {
        red[(wave * 18 + t) * RC_CB + c] = a;
        red[(wave * 18 + 9 + t) * RC_CB + c] = b;
      }
    }
    __syncthreads();
    // sums stripe of this image: [2][9*C] in [t*C + c] order
    for (int i = tid; i < 18 * RC_CB; i += LY_THREADS) {
      const int q = i / RC_CB, cc = i - q * RC_CB;
      const float sv = (red[i] + red[18 * RC_CB + i]) + (red[2 * 18 * RC_CB + i] + red[3 * 18 * RC_CB + i]);
      P.sums[(long)n * (18 * P.C) + (long)(q / 9) * (9 * P.C) + (q % 9) * P.C + c0 + cc] = sv;
    }
  }
  if constexpr (MODE == RB_C) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(xs);          // [4 waves][81][32] = 41 KB over xs + dt
#pragma unroll
    for (int i = 0; i < 81; ++i) {
      float a = dwd[i];
      a += __shfl_xor(a, 32);
      if (half == 0) red[(wave * 81 + i) * RC_CB + c] = a;
    }
    __syncthreads();
    // dwg row of this image: generate.0.weight layout [c*9 + t][u']
    for (int i = tid; i < 81 * RC_CB; i += LY_THREADS) {
      const int cc = i / 81, e = i - cc * 81;
      const int k = e * RC_CB + cc;
      P.dwg[(long)n * (P.C * 81) + (long)(c0 + cc) * 81 + e] = (red[k] + red[81 * RC_CB + k]) + (red[2 * 81 * RC_CB + k] + red[3 * 81 * RC_CB + k]);
    }
  }
}

template <int MODE, int KS>
static int rb_launch(const LyRf3cBwdParams& P, hipStream_t st) {
  const int nct = (P.Wo + P.TW - 1) / P.TW, nrt = (P.Ho + P.TH - 1) / P.TH;
  const int IH = 2 * (P.TH - 1) + 3, IW = 2 * (P.TW - 1) + 3;
  constexpr int O = 32 * KS;
  size_t lds = (size_t)IH * IW * RC_CB * 4 + RC_KR * 128 + RC_TP * (2 * O + 16) + (size_t)32 * 9 * (MODE == RB_A ? 2 : 8) * 4 + 64 * 4;
  if (MODE == RB_C) lds += ((size_t)IH * IW + IH + 2 * P.TW * nct + 2 + 27) * RC_CB * 4;
  LY_CHECK(lds <= 160 * 1024, "rf3c_bwd: pass %d needs %zu B LDS for this shape (tile %dx%d, W = %d, O = %d)", MODE, lds, P.TH, P.TW, P.W, O);
  auto k = ly_rf3c_bwd_kernel<MODE, KS>;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    LY_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    configured = true;
  }
  // (launched with the whole LDS of a CU: a block of this family must not share its CU with any other block, see ly_rf3c.hip)
  hipLaunchKernelGGL(k, dim3((unsigned)(P.n_img * (P.C / RC_CB))), dim3(LY_THREADS), (size_t)(160 * 1024), st, P, nct, nrt);
  LY_LAUNCH_CHECK();
  return 0;
}

template <int MODE>
static int rb_dispatch(const LyRf3cBwdParams& P, hipStream_t st) {
  if (P.O == 64) return rb_launch<MODE, 2>(P, st);
  if (P.O == 128) return rb_launch<MODE, 4>(P, st);
  return rb_launch<MODE, 8>(P, st);
}

extern "C" int ly_rf3c_bwd(const LyRf3cBwdParams* p, int pass, void* stream) {
  LY_CHECK(p, "rf3c_bwd: null params");
  const LyRf3cBwdParams& P = *p;
  LY_CHECK(P.dtype == LY_BF16, "rf3c_bwd: built for bf16 storage (dtype %d)", P.dtype);
  LY_CHECK(P.s == 2 && (P.O == 64 || P.O == 128 || P.O == 256), "rf3c_bwd: built for stride 2 and 64 / 128 / 256 output channels (s = %d, O = %d)", P.s, P.O);
  if (rc_check_tile("rf3c_bwd", P.C, P.s, P.TH, P.TW, P.ldx, P.x)) return -1;
  LY_CHECK(P.x && P.du && P.wq && P.wct && P.ca && P.rfa && (P.lddu & 7) == 0 && ((uintptr_t)P.du & 15) == 0, "rf3c_bwd: null / misaligned pointer");
  LY_CHECK((long)P.n_img * P.H * P.W * P.ldx < (1L << 31), "rf3c_bwd: input exceeds the 31-bit offsets of the staging plan");
  LY_CHECK(pass != RB_C || (long)P.n_img * P.H * P.W * P.lddx < (1L << 31), "rf3c_bwd: dx exceeds 31-bit offsets");
  LY_CHECK(P.Ho == (P.H + 2 - 3) / P.s + 1 && P.Wo == (P.W + 2 - 3) / P.s + 1, "rf3c_bwd: inconsistent output size");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (pass == RB_A) {
    LY_CHECK(P.d_rfa_part && P.d_ca, "rf3c_bwd: pass A needs d_rfa_part and d_ca");
    return rb_dispatch<RB_A>(P, st);
  }
  if (pass == RB_B) {
    LY_CHECK(P.mm && P.d_mm && P.sums, "rf3c_bwd: pass B needs mm, d_mm and sums");
    return rb_dispatch<RB_B>(P, st);
  }
  LY_CHECK(pass == RB_C && P.mm && P.d_mm && P.coef && P.dwg && P.dx && (P.lddx & 7) == 0 && ((uintptr_t)P.dx & 15) == 0, "rf3c_bwd: pass C needs mm, d_mm, coef, dwg and an aligned dx");
  return rb_dispatch<RB_C>(P, st);
}

// ---------------------------------------------------------------------------------------------------
// conv.0.weight gradient:  dWc[o, (t, c)] = sum_p du[p, o] * cd[p, (t, c)],  cd = G*ca*rfa regenerated exactly as the training forward does.
// Block = (image group, 128-row output group, 32-channel chunk): 128 x 288 fp32 accumulators (144 registers per lane) live over all the
// block's pixels; per 64-pixel tile the VALU writes the K-major cd tile (pixels contiguous = the contraction index: plain 8-byte operand
// reads), du^T comes out of the pixel-major du tile by transposed reads (ds_read_b64_tr_b16), two k-steps of 32 pixels.
// The accumulators leave as ONE slab per image group (plain stores, summed by the caller): no float atomics, deterministic.
// LDS: x tile fp32 | cd tile bf16 [288][128 B] | du tile [64 px][288 B] | rfa pairs [32][9][2]
// ---------------------------------------------------------------------------------------------------
#define RW_OG 128                      // output channels per block
#define RW_RS (2 * RW_OG + 32)         // du-tile row bytes: 8 (mod 64) dwords => the transposed reads of 8 pixel rows x 32 bytes hit all 64 banks

__global__ __launch_bounds__(LY_THREADS) void ly_rf3c_wgrad_kernel(const LyRf3cBwdParams P, const int nct, const int nrt, const int nog) {
  typedef __bf16 T;
  constexpr int S = 2;
  extern __shared__ f32x4 rc_smem4[];
  const RcGeom g = rc_geom(S, P.TH, P.TW);
  const int IHW = g.IH * g.IW;
  float* xs = reinterpret_cast<float*>(rc_smem4);
  char* cdt = reinterpret_cast<char*>(xs + IHW * RC_CB);
  char* dus = cdt + RC_KR * 128;
  float* rfs = reinterpret_cast<float*>(dus + RC_TP * RW_RS);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, c = lane & 31;
  const int li = lane & 15, lq = lane >> 4;
  const int NCH = P.C / RC_CB;
  int b = ly_xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int chunk = b % NCH; b /= NCH;
  const int og = b % nog;
  const int ig = b / nog;
  const int c0 = chunk * RC_CB, o0 = og * RW_OG;
  const int ovalid = P.O - o0 < RW_OG ? P.O - o0 : RW_OG;         // 64 or 128
  const T* const x = reinterpret_cast<const T*>(P.x);
  const T* const du = reinterpret_cast<const T*>(P.du);
  const int HK = 3 * P.Ho, WK = 3 * P.Wo;

  const int stream = wave * 2 + half;
  const int row = g.IW * RC_CB;
  const int csw = rc_sw(c);
  const float* xp[4];
  int goff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int px0 = 8 * stream + 2 * j;
    xp[j] = xs + rc_pos0(g, px0) * RC_CB + c;
    goff[j] = c * 128 + ((((px0 >> 2) ^ csw) << 3) | ((px0 & 2) << 1));
  }
  RcW w;
  rc_load_w<true>(w, P.wq, c0, c);

  f32x4 acc[2][18];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int e = 0; e < 18; ++e) acc[m][e] = ly_zero4();

  // operand addressing (per lane, tile independent)
  //   A = du^T: lane supplies pixel row 32s + 16hh + 4*lq + (li >> 2), 8-byte chunk (li & 3) of the 16 output channels of m-tile mt
  const int arow = (4 * lq + (li >> 2)) * RW_RS + (li & 3) * 8;
  //   B = cd: lane reads row k = 16e + li, pixel chunk 8s + 4hh + lq; the swizzle depends on k & 31 = 16*(e & 1) + li
  const int bsw0 = rc_sw(li), bsw1 = rc_sw(16 + li);

  RcStage<T> St;
  ly_u32x4 dpv[4];
  bool dok[4];
  float tv[3];
  bool tok[3];
  auto issue = [&](int n, int tt) {
    const int ct = tt % nct, rt = tt / nct;
    const int oy0 = rt * P.TH, ox0 = ct * P.TW;
    rc_stage_plan(St, g, tid, n, P.H, P.W, P.ldx, S * oy0 - 1, S * ox0 - 1);
    rc_stage_load(St, x, c0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + e * LY_THREADS;
      const int px = idx >> 4, v8 = idx & 15;
      const int ly = px / P.TW, lx = px - ly * P.TW;
      const int oy = oy0 + ly, ox = ox0 + lx;
      dok[e] = px < g.NPX && oy < P.Ho && ox < P.Wo && 8 * v8 < ovalid;
      const long m = dok[e] ? ((long)n * P.Ho + oy) * P.Wo + ox : 0;
      dpv[e] = *reinterpret_cast<const ly_u32x4*>(du + m * P.lddu + o0 + (dok[e] ? 8 * v8 : 0));
    }
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      const int idx = tid + e * LY_THREADS;
      const int px = idx / 9, t = idx - px * 9;
      const int ly = px / P.TW, lx = px - ly * P.TW;
      const int oy = oy0 + ly, ox = ox0 + lx;
      tok[e] = idx < RC_TP * 9 && px < g.NPX && oy < P.Ho && ox < P.Wo;
      tv[e] = P.rfa[tok[e] ? ((long)n * HK + 3 * oy + t / 3) * WK + 3 * ox + t % 3 : 0];
    }
  };
  auto commit = [&]() {
    rc_stage_store(St, xs);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + e * LY_THREADS;
      const int px = idx >> 4, v8 = idx & 15;
      *reinterpret_cast<ly_u32x4*>(dus + px * RW_RS + 16 * v8) = dok[e] ? dpv[e] : (ly_u32x4){0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      const int idx = tid + e * LY_THREADS;
      const int px = idx / 9, t = idx - px * 9;
      if (idx < RC_TP * 9) rfs[((px >> 1) * 9 + t) * 2 + (px & 1)] = tok[e] ? tv[e] : 0.f;
    }
  };

  typedef __attribute__((address_space(3))) ly_s16x4 lds_s16x4;
  const int ntile = nct * nrt;
  int n = ig;
  if (n < P.n_img) issue(n, 0);
  for (; n < P.n_img; n += P.ng) {
    const float cav = P.ca[(long)n * P.C + c0 + c];
    const f32x2 cav2 = {cav, cav};
    for (int tt = 0; tt < ntile; ++tt) {
      __syncthreads();
      commit();
      {
        const bool lastt = tt + 1 == ntile;
        const int nn = lastt ? (n + P.ng < P.n_img ? n + P.ng : n) : n;
        issue(nn, lastt ? (n + P.ng < P.n_img ? 0 : tt) : tt + 1);
      }
      __syncthreads();
      // ---- cd tile: G * ca * rfa, exactly the training forward's operand ----
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x2 xv[9], a[9];
        rc_patch<S>(xp[j], row, xv);
        rc_gen_bn<true>(w, xv, a);
        const f32x2* rfp = reinterpret_cast<const f32x2*>(rfs) + (4 * stream + j) * 9;
#pragma unroll
        for (int t = 0; t < 9; ++t)
          *reinterpret_cast<unsigned*>(cdt + t * (RC_CB * 128) + goff[j]) = rc_pack2((f32x2){rc_relu(a[t][0]), rc_relu(a[t][1])} * (rfp[t] * cav2));
      }
      __syncthreads();
      // ---- dWc += du^T . cd over the tile's 64 pixels: wave w owns output rows 32w .. 32w+31 ----
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 af[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const char* ab = dus + (32 * s2) * RW_RS + arow + (16 * (2 * wave + m)) * 2;
          const ly_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ab));
          const ly_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ab + 16 * RW_RS));
          af[m] = ly_cat8(__builtin_bit_cast(bf16x4, lo), __builtin_bit_cast(bf16x4, hi));
        }
#pragma unroll
        for (int e = 0; e < 18; ++e) {
          const char* br = cdt + (16 * e + li) * 128;
          const int sw = (e & 1) ? bsw1 : bsw0;
          const bf16x4 b0 = *reinterpret_cast<const bf16x4*>(br + (((8 * s2 + lq) ^ sw) << 3));
          const bf16x4 b1 = *reinterpret_cast<const bf16x4*>(br + (((8 * s2 + 4 + lq) ^ sw) << 3));
          const bf16x8 bf = ly_cat8(b0, b1);
#pragma unroll
          for (int m = 0; m < 2; ++m) acc[m][e] = ly_mfma_bf16(af[m], bf, acc[m][e]);
        }
      }
    }
  }
  // ---- one slab per image group: dwc_part[ig][o][t][c] ----
  float* slab = P.dwc_part + (long)ig * P.O * 9 * P.C;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int e = 0; e < 18; ++e) {
      const int t = e >> 1, cc = c0 + 16 * (e & 1) + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ol = 16 * (2 * wave + m) + 4 * lq + r;
        if (ol < ovalid) slab[((long)(o0 + ol) * 9 + t) * P.C + cc] = acc[m][e][r];
      }
    }
}

extern "C" int ly_rf3c_wgrad(const LyRf3cBwdParams* p, void* stream) {
  LY_CHECK(p, "rf3c_wgrad: null params");
  const LyRf3cBwdParams& P = *p;
  LY_CHECK(P.dtype == LY_BF16, "rf3c_wgrad: built for bf16 storage (dtype %d)", P.dtype);
  LY_CHECK(P.s == 2 && (P.O % 64) == 0, "rf3c_wgrad: built for stride 2 and output channels in multiples of 64 (s = %d, O = %d)", P.s, P.O);
  if (rc_check_tile("rf3c_wgrad", P.C, P.s, P.TH, P.TW, P.ldx, P.x)) return -1;
  LY_CHECK(P.x && P.du && P.wq && P.ca && P.rfa && P.dwc_part && P.ng > 0 && (P.lddu & 7) == 0 && ((uintptr_t)P.du & 15) == 0, "rf3c_wgrad: null / misaligned pointer");
  LY_CHECK((long)P.n_img * P.H * P.W * P.ldx < (1L << 31), "rf3c_wgrad: input exceeds the 31-bit offsets of the staging plan");
  const int nct = (P.Wo + P.TW - 1) / P.TW, nrt = (P.Ho + P.TH - 1) / P.TH;
  const int nog = (P.O + RW_OG - 1) / RW_OG;
  const int IH = 2 * (P.TH - 1) + 3, IW = 2 * (P.TW - 1) + 3;
  const size_t lds = (size_t)IH * IW * RC_CB * 4 + RC_KR * 128 + RC_TP * RW_RS + RC_TP * 9 * 4;
  LY_CHECK(lds <= 160 * 1024, "rf3c_wgrad: tile needs %zu B LDS", lds);
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ly_rf3c_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    LY_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    configured = true;
  }
  const int ng = P.ng < P.n_img ? P.ng : P.n_img;
  LY_CHECK(ng == P.ng, "rf3c_wgrad: ng = %d exceeds the %d images", P.ng, P.n_img);
  hipLaunchKernelGGL(ly_rf3c_wgrad_kernel, dim3((unsigned)(ng * nog * (P.C / RC_CB))), dim3(LY_THREADS), (size_t)(160 * 1024), reinterpret_cast<hipStream_t>(stream), P, nct,
                     nrt, nog);
  LY_LAUNCH_CHECK();
  return 0;
}
