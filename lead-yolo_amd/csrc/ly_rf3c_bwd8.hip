// RFCBAMConv kernel_size 3 backward, passes B and C of ly_rf3c_bwd.hip on EIGHT waves with the nine taps split over the half waves
// (reference models/rfa.py:113-129 under autograd; the pass structure, LDS tiles and carries are those of ly_rf3c_bwd.hip — read its header first).
//
// The four-wave kernels keep, per lane (= channel), the channel's 81 generate weights + 18 BatchNorm terms and — pass C — 81 weight-gradient
// accumulators: 180-250 live registers, ONE wave per SIMD, every LDS round trip of the pair loop exposed (SQ_WAIT_ANY 37 %, 50 lgkmcnt waits per
// pair iteration), and hand-written v_pk_fma_f32 (RC_ASM_FMA) to keep hipcc from spilling — an instruction form that returns wrong results beside
// another wave's MFMAs, guarded only by "one wave per SIMD, the block owns its CU".  Everything in passes B / C is independent PER TAP (G, dv,
// du_g, the generate weight gradient; only dx sums over taps), so here the two half waves of a channel split the TAPS (5 + 4) of the SAME pixel
// pair: 45 weights + 10 BatchNorm terms + 45 accumulators per lane, plain compiler FMAs (no asm), the kernel fits 256 registers and TWO waves per
// SIMD (512-thread blocks) hide each other's LDS latency.  The dx partial sums of the two halves meet through v_permlane32_swap (a VALU move);
// the pair's 3 x 5 patch of the dx tile is updated 8 + 7 elements per half.
//
// The generate weight gradient leaves pass C altogether.  d(generate.0.weight)[c][t][u'] = sum_p du_g[p,t] x[p,u'] with du_g = alpha dv + kappa +
// lambda u and u = w[t] . x is LINEAR in three sums:   alpha_t * A[t][u'] + kappa_t * m[u'] + lambda_t * (w[t] . M)[u'],   where
// A[t][u'] = sum_p dv[p,t] x[p,u'] does not need the BatchNorm coefficients — pass B accumulates it next to its two BatchNorm sums — and
// m = sum_p x[p, .], M = sum_p x[p, .] x[p, .]^T are the 9 + 45 tap moments the FORWARD already took for the generate BatchNorm's batch statistics
// (ly_rfcbam_tap_moments).  ly_rf3c_dwg_finish combines them per (channel, tap, u'): 81 fewer accumulators and 81 fewer MACs per (pixel,
// channel) in pass C, whose per-lane state is then the 55 weight terms alone.
#include "ly_rf3c.hpp"
#include "ly_params.h"

#define R8_THREADS 512
#define R8_NW 8
#define R8_NT 5                  // tap slots per half wave: taps 5h .. 5h + 4 (slot 4 of the upper half is empty)
#ifndef R8_NA
#define R8_NA 4                  // the A sums of slots [0, R8_NA) are accumulated by pass B, of slots [R8_NA, 5) by pass C (zero spills in both at 36 + 9 accumulators; 45 + 0 spills pass B, 27 + 18 pass B at O = 256)
#endif
enum { R8_B = 1, R8_C = 2 };

// v[i] += the partner half wave's v[i] (lane ^ 32), nine values at a time: v_permlane32_swap a, b exchanges a[32:63] with b[0:31], so from two
// copies of a value a' = (lower, lower), b' = (upper, upper) and a' + b' is the sum in every lane — a VALU move, not an LDS-pipe shuffle.
// Written as asm with its own wait states: (1) ROCm 7.2's __builtin_amdgcn_permlane32_swap hands back the FIRST register for both elements of its
// result (seen in the ISA: the second result is stored from v1 as well), which loses the upper half; (2) issued right behind the VALU write of
// its operands the instruction returned (lower, upper) in a — the swap half of the hazard — so the block opens and closes with s_nop 4
// (measured with tools' swap probe: without them a' = a, with them the values above).
__device__ __forceinline__ void r8_sum32x9(float (&v)[9]) {
  unsigned a[9], b[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) { a[i] = __builtin_bit_cast(unsigned, v[i]); b[i] = a[i]; }
  asm volatile("s_nop 4\n\t"
               "v_permlane32_swap_b32_e32 %0, %9\n\tv_permlane32_swap_b32_e32 %1, %10\n\tv_permlane32_swap_b32_e32 %2, %11\n\t"
               "v_permlane32_swap_b32_e32 %3, %12\n\tv_permlane32_swap_b32_e32 %4, %13\n\tv_permlane32_swap_b32_e32 %5, %14\n\t"
               "v_permlane32_swap_b32_e32 %6, %15\n\tv_permlane32_swap_b32_e32 %7, %16\n\tv_permlane32_swap_b32_e32 %8, %17\n\t"
               "s_nop 4"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),
                 "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]), "+v"(b[8]));
#pragma unroll
  for (int i = 0; i < 9; ++i) v[i] = __builtin_bit_cast(float, a[i]) + __builtin_bit_cast(float, b[i]);
}

template <int MODE, int KS>
__global__ __launch_bounds__(R8_THREADS) void ly_rf3c_bwd8_kernel(const LyRf3cBwdParams P, const int nct, const int nrt) {
  typedef __bf16 T;
  constexpr int S = 2;
  constexpr int O = 32 * KS;
  constexpr int RSD = 2 * O + 16;                       // bytes per du-tile row: RSD/16 odd => the b64 fragment reads are conflict-free (ly_tile.hpp)
  constexpr int NDU = (64 * (O / 8) + R8_THREADS - 1) / R8_THREADS;      // 16-byte du items per thread
  constexpr int TABW = 8;                               // floats per (pixel pair, tap): rfa, max_c G, d_max, d_mean/C  (x 2 pixels)
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
  const int chunk = blockIdx.x % NCH, n = blockIdx.x / NCH;
  const int c0 = chunk * RC_CB;
  const T* const x = reinterpret_cast<const T*>(P.x);
  const T* const du = reinterpret_cast<const T*>(P.du);
  const int HK = 3 * P.Ho, WK = 3 * P.Wo;
  const float invC = 1.f / (float)P.C;
  // dx output pass (pass C): IW*4 threads per tile row, dxo_nrs rows at a time
  const int dxo_per = g.IW * 4, dxo_nrs = R8_THREADS / dxo_per > 0 ? R8_THREADS / dxo_per : 1;
  const int dxo_rsub = tid / dxo_per, dxo_q = (tid - dxo_rsub * dxo_per) >> 2;

  // ---- per-lane constants of the VALU phase: pair stream = wave, tap group = half wave ---------------------------------
  const int stream = wave;                                  // pixel pairs px0 = 8*stream + 2*j, j < 4 (pass B); colour walk (pass C)
  const int row = g.IW * RC_CB;
  const int csw = rc_sw(c);
  const int t0 = R8_NT * half;                              // first tap of the lane
  // the lane's 45 generate weights and 10 BatchNorm terms (raw form: u = w.x, v = a*u + b); image element i of channel c at
  // ((chunk*25 + (i >> 2))*32 + c)*4 + (i & 3) (pack.rfcbam_gen_weights_c); the empty slot holds zeros
  float wv[R8_NT][9], wa[R8_NT], wb[R8_NT];
  int tt[R8_NT];                                            // tap of slot j (clamped to 8 for the empty slot)
  bool tv_[R8_NT];
  {
    const float* src = P.wq + ((long)(c0 >> 5) * (RC_WQ / 4) * RC_CB + c) * 4;
    auto wel = [&](int i) -> float { return src[(i >> 2) * (RC_CB * 4) + (i & 3)]; };
#pragma unroll
    for (int j = 0; j < R8_NT; ++j) {
      tv_[j] = t0 + j < 9;
      tt[j] = tv_[j] ? t0 + j : 8;
#pragma unroll
      for (int u = 0; u < 9; ++u) {
        const float v = wel(tt[j] * 9 + u);
        wv[j][u] = tv_[j] ? v : 0.f;
      }
      const float b = wel(81 + tt[j]), a = wel(90 + tt[j]);
      wb[j] = tv_[j] ? b : 0.f;
      wa[j] = tv_[j] ? a : 0.f;
    }
  }
  const float cav = P.ca[(long)n * P.C + c0 + c];

  // pass state
  float s1[R8_NT], s2[R8_NT];                               // B: BatchNorm sums sum dv, sum dv u (both pixels; s2 of the slots whose A this pass
                                                            // holds is derived when the block ends: sum_p dv u = w[t] . A[t])
  constexpr int SA0 = MODE == R8_B ? 0 : R8_NA, SA1 = MODE == R8_B ? R8_NA : R8_NT;      // A slots of this pass
  float dwa[(SA1 - SA0) * 9 > 0 ? (SA1 - SA0) * 9 : 1];    // A[t][u'] = sum_p dv[p,t] x[p,u'] of the lane's taps (the slots this pass owns)
  if constexpr (MODE == R8_B) {
#pragma unroll
    for (int t = 0; t < R8_NT; ++t) { s1[t] = 0.f; s2[t] = 0.f; }
  }
#pragma unroll
  for (int i = 0; i < (SA1 - SA0) * 9; ++i) dwa[i] = 0.f;
  if constexpr (MODE == R8_C) {
    for (int i = tid; i < 27 * RC_CB; i += R8_THREADS) cfs[i] = P.coef[(long)(i / RC_CB) * P.C + c0 + (i % RC_CB)];
    // dx tile and carries start at zero
    for (int i = tid; i < (IHW + g.IH + S * P.TW * nct + 2) * RC_CB; i += R8_THREADS) dxs[i] = 0.f;
  }

  // ---- staging: x tile, du tile, per-pixel tables of the NEXT tile are requested while the current one is processed --------------
  RcStage<T, R8_THREADS> St;
  RcStageFix<T, R8_THREADS> Sf;
  rc_stage_fix(Sf, St, g, tid, P.W, P.ldx);
  ly_u32x4 dpv[NDU];
  bool dok[NDU];
  int dyx[NDU];                                        // du item e: pixel (ly << 16 | lx), -1: pixel slot past the tile / no item
#pragma unroll
  for (int e = 0; e < NDU; ++e) {
    const int idx = tid + e * R8_THREADS;
    const int px = idx / (O / 8);
    const int ly = px / P.TW, lx = px - ly * P.TW;
    dyx[e] = (idx < 64 * (O / 8) && px < g.NPX) ? (ly << 16) | lx : -1;
  }
  constexpr int NTB = (RC_TP * 9 + R8_THREADS - 1) / R8_THREADS;     // table items per thread (2)
  float tv[NTB][4];
  bool tok[NTB];
  int tyx[NTB], trel[NTB];                             // table item e: pixel (ly << 16 | lx) (-1: none), map offset (3*ly + ty)*WK + 3*lx + tx
#pragma unroll
  for (int e = 0; e < NTB; ++e) {
    const int idx = tid + e * R8_THREADS;
    const int px = idx / 9, t = idx - px * 9;
    const int ly = px / P.TW, lx = px - ly * P.TW;
    tyx[e] = (idx < RC_TP * 9 && px < g.NPX) ? (ly << 16) | lx : -1;
    trel[e] = (3 * ly + t / 3) * WK + 3 * lx + t % 3;
  }
  auto issue = [&](int tt_) {
    const int ct = tt_ % nct, rt = tt_ / nct;
    const int oy0 = rt * P.TH, ox0 = ct * P.TW;
    rc_stage_retarget(St, Sf, ((n * P.H + S * oy0 - 1) * P.W + S * ox0 - 1) * P.ldx, S * oy0 - 1, S * ox0 - 1, P.H, P.W);
    rc_stage_load(St, x, c0);
    const int mbase = (n * P.Ho + oy0) * P.Wo + ox0;
#pragma unroll
    for (int e = 0; e < NDU; ++e) {
      const int ly = dyx[e] >> 16, lx = dyx[e] & 0xffff;
      dok[e] = dyx[e] >= 0 && oy0 + ly < P.Ho && ox0 + lx < P.Wo;
      const int m = dok[e] ? mbase + ly * P.Wo + lx : 0;
      dpv[e] = *reinterpret_cast<const ly_u32x4*>(du + (long)m * P.lddu + 8 * ((tid + e * R8_THREADS) % (O / 8)));
    }
    const int pbase = (n * HK + 3 * oy0) * WK + 3 * ox0;
#pragma unroll
    for (int e = 0; e < NTB; ++e) {
      tok[e] = tyx[e] >= 0 && oy0 + (tyx[e] >> 16) < P.Ho && ox0 + (tyx[e] & 0xffff) < P.Wo;
      const int pos = tok[e] ? pbase + trel[e] : 0;
      tv[e][0] = P.rfa[pos];
      tv[e][1] = P.mm[2 * pos];
      const f32x2 d = *reinterpret_cast<const f32x2*>(P.d_mm + 2 * pos);
      tv[e][2] = d[0];
      tv[e][3] = d[1];                                   // (scaled by 1/C at commit: nothing here may wait for a load)
    }
  };
  auto commit = [&]() {
    rc_stage_store(St, xs);
#pragma unroll
    for (int e = 0; e < NDU; ++e) {
      const int idx = tid + e * R8_THREADS;
      const int px = idx / (O / 8), v8 = idx - px * (O / 8);
      if (idx < 64 * (O / 8)) *reinterpret_cast<ly_u32x4*>(dus + px * RSD + 16 * v8) = dok[e] ? dpv[e] : (ly_u32x4){0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int e = 0; e < NTB; ++e) {
      const int idx = tid + e * R8_THREADS;
      const int px = idx / 9, t = idx - px * 9;
      if (idx < RC_TP * 9) {
        float* d = tab + ((px >> 1) * 9 + t) * TABW + (px & 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) d[2 * q] = tok[e] ? (q == 3 ? tv[e][q] * invC : tv[e][q]) : 0.f;
        if (MODE == R8_C && t == 0) act[px] = tok[e] ? 1.f : 0.f;
      }
    }
  };

  // dcd contraction: wave w owns the 16-column tiles e_i = w + 8i (i < 3; the third exists for waves 0 and 1 only) of the 18.  Its Wc^T
  // fragments are the same for every unit (the chunk is fixed): a register ring runs one k-step ahead, ACROSS units.
  constexpr int NE = 3;
  const uint4* wpk = reinterpret_cast<const uint4*>(P.wct);
  const int ntile = nct * nrt;
  issue(0);
  for (int tt_ = 0; tt_ < ntile; ++tt_) {
    const int ct = tt_ % nct, rt = tt_ / nct;
    const int oy0 = rt * P.TH, ox0 = ct * P.TW;
    __syncthreads();                                   // the previous tile is done with every LDS region
    commit();
    issue(tt_ + 1 < ntile ? tt_ + 1 : tt_);            // unconditional (the last tile re-requests itself): no load under a branch
    __syncthreads();

    // ---- dcd tile on the MFMAs: D[px][(t, c)] = du[px][:] . Wc^T[:, (t, c)] ----
    // one column tile at a time (4 accumulator quads: the lane's 100 weight / accumulator registers stay live through this phase)
#pragma unroll 1
    for (int i = 0; i < NE; ++i) {
      const int e = wave + R8_NW * i;
      if (e >= 18) break;                                  // (wave-uniform)
      f32x4 acc[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt] = ly_zero4();
      const uint4* wf = wpk + (((e >> 1) * (P.C / 16) + chunk * 2 + (e & 1)) * KS) * 64 + lane;
      bf16x8 bfr = __builtin_bit_cast(bf16x8, wf[0]);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const bf16x8 cur = bfr;
        if (s + 1 < KS) bfr = __builtin_bit_cast(bf16x8, wf[(s + 1) * 64]);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = ly_mfma_bf16(ly_lds_frag(dus, (16 * mt + li) * RSD, s, lq), cur, acc[mt]);
      }
      const int kk = 16 * (e & 1) + li;
      char* drow = dt + ((e >> 1) * RC_CB + kk) * 128;
      const int sw = rc_sw(kk);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<bf16x4*>(drow + (((4 * mt + lq) ^ sw) << 3)) = ly_cvtb4(acc[mt]);
    }
    __syncthreads();

    // ---- VALU phase, lane = (channel, tap group) ---------------------------------------------------------
    // Pass C walks its pairs in FOUR COLOURS (output-row parity x pair-column parity): the 3x5 input patches of two pairs of one colour never
    // overlap, so the dx tile is updated with plain read-add-write and a barrier between colours (see ly_rf3c_bwd.hip).
#pragma unroll 1
    for (int col = 0; col < (MODE == R8_C ? 4 : 1); ++col) {
    int nit = 4, npx = 1, cnt = 0;
    const int ry = col >> 1, rx = col & 1;
    if constexpr (MODE == R8_C) {
      npx = ((P.TW >> 1) + 1 - rx) >> 1;
      cnt = ((P.TH + 1 - ry) >> 1) * npx;
      nit = (cnt + 7) >> 3;
      if (col > 0) __syncthreads();
    }
#pragma unroll 1
    for (int j = 0; j < nit; ++j) {
      int px0 = 8 * stream + 2 * j;
      if constexpr (MODE == R8_C) {
        const int idx = j * 8 + stream;
        if (idx >= cnt) continue;                        // (wave-uniform: the stream is the wave)
        const int ia = idx / npx, ib = idx - ia * npx;
        px0 = (2 * ia + ry) * P.TW + 2 * (2 * ib + rx);
      }
      const int pos0j = rc_pos0(g, px0);
      const float* xpj = xs + pos0j * RC_CB + c;
      const int goffj = c * 128 + ((((px0 >> 2) ^ csw) << 3) | ((px0 & 2) << 1));
      const int pair = px0 >> 1;
      f32x2 xv[9];
      rc_patch<S>(xpj, row, xv);
      const f32x4* tb = reinterpret_cast<const f32x4*>(tab) + pair * 18;
      const f32x2 actp = MODE == R8_C ? *reinterpret_cast<const f32x2*>(act + 2 * pair) : (f32x2){1.f, 1.f};
      f32x2 dvs[R8_NT];
#pragma unroll
      for (int s = 0; s < R8_NT; ++s) {
        // u = w . x (nine terms, the training forward's order: rc_generate<true>), v = a u + b (rc_affine)
        f32x2 u = {xv[0][0] * wv[s][0], xv[0][1] * wv[s][0]};
#pragma unroll
        for (int k = 1; k < 9; ++k) u = (f32x2){__builtin_fmaf(xv[k][0], wv[s][k], u[0]), __builtin_fmaf(xv[k][1], wv[s][k], u[1])};
        const f32x2 v = {__builtin_fmaf(u[0], wa[s], wb[s]), __builtin_fmaf(u[1], wa[s], wb[s])};
        const f32x4 ta = tb[2 * tt[s]];                  // (rfa0, rfa1, gmax0, gmax1)
        const f32x4 tc = tb[2 * tt[s] + 1];              // (dmax0, dmax1, dmean0/C, dmean1/C)
        const f32x2 dc = rc_unpack2(*reinterpret_cast<const unsigned*>(dt + tt[s] * (RC_CB * 128) + goffj));
        const f32x2 G = {rc_relu(v[0]), rc_relu(v[1])};
        f32x2 dG = {dc[0] * (ta[0] * cav) + tc[2], dc[1] * (ta[1] * cav) + tc[3]};
        dG[0] += G[0] == ta[2] ? tc[0] : 0.f;
        dG[1] += G[1] == ta[3] ? tc[1] : 0.f;
        const f32x2 dvv = {G[0] > 0.f ? dG[0] : 0.f, G[1] > 0.f ? dG[1] : 0.f};
        if (s >= SA0 && s < SA1) {
#pragma unroll
          for (int uu = 0; uu < 9; ++uu) dwa[(s - SA0) * 9 + uu] += dvv[0] * xv[uu][0] + dvv[1] * xv[uu][1];
        }
        if constexpr (MODE == R8_B) {
          s1[s] += dvv[0] + dvv[1];
          if (s >= SA1) s2[s] += dvv[0] * u[0] + dvv[1] * u[1];
        } else {
          // du_g = alpha*dv + kappa + lambda*u, zero for pixel slots outside the map and for the empty tap slot
          const float al = cfs[tt[s] * RC_CB + c], ka = cfs[(9 + tt[s]) * RC_CB + c], la = cfs[(18 + tt[s]) * RC_CB + c];
          const float m0 = tv_[s] ? actp[0] : 0.f, m1 = tv_[s] ? actp[1] : 0.f;
          dvs[s] = (f32x2){(dvv[0] * al + ka + u[0] * la) * m0, (dvv[1] * al + ka + u[1] * la) * m1};
        }
      }
      if constexpr (MODE == R8_C) {
        // data gradient of the patch over the lane's taps, completed by the partner half: dxc[u'] = sum_t w[t][u'] * du_g[t]
        float dx0[9], dx1[9];                            // the two pixels of the pair
#pragma unroll
        for (int uu = 0; uu < 9; ++uu) {
          f32x2 a = {dvs[0][0] * wv[0][uu], dvs[0][1] * wv[0][uu]};
#pragma unroll
          for (int s = 1; s < R8_NT; ++s) a = (f32x2){__builtin_fmaf(dvs[s][0], wv[s][uu], a[0]), __builtin_fmaf(dvs[s][1], wv[s][uu], a[1])};
          dx0[uu] = a[0]; dx1[uu] = a[1];
        }
        r8_sum32x9(dx0);
        r8_sum32x9(dx1);
        // the pair's 3 x 5 input patch (the middle column belongs to both pixels): element 5*uy + k; the lower half updates elements 0 .. 7, the
        // upper half 8 .. 14 (its eighth slot repeats element 14 and is not written)
        float el[15];
#pragma unroll
        for (int uy = 0; uy < 3; ++uy) {
          el[5 * uy] = dx0[uy * 3]; el[5 * uy + 1] = dx0[uy * 3 + 1]; el[5 * uy + 2] = dx0[uy * 3 + 2] + dx1[uy * 3];
          el[5 * uy + 3] = dx1[uy * 3 + 1]; el[5 * uy + 4] = dx1[uy * 3 + 2];
        }
        float* dp = dxs + pos0j * RC_CB + c;
        float old[8], add[8];
        int offs[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const int ia_ = m, ib_ = m + 8 < 15 ? m + 8 : 14;                   // element of the lower / upper half
          const int oa = (ia_ / 5) * row + (ia_ % 5) * RC_CB, ob = (ib_ / 5) * row + (ib_ % 5) * RC_CB;
          offs[m] = half ? ob : oa;
          add[m] = half ? el[ib_] : el[ia_];
          old[m] = dp[offs[m]];
        }
#pragma unroll
        for (int m = 0; m < 8; ++m)
          if (m < 7 || !half) dp[offs[m]] = old[m] + add[m];
      }
    }
    }

    if constexpr (MODE == R8_C) {
      // ---- dx: rows / columns this tile completes leave as T (+ the SE term); its last row / column is carried to the neighbours ----
      __syncthreads();
      const int iy0 = S * oy0 - 1, ix0 = S * ox0 - 1;
      const int RL = g.IH - 1, CL = g.IW - 1;            // the seam row / column (shared with the tile below / to the right)
      const bool lastr = rt == nrt - 1, lastc = ct == nct - 1;
      T* const dxo = reinterpret_cast<T*>(P.dx);
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
#pragma unroll 1
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
      // next tile (raster order): column 0 (rows above the seam row) from the right carry, row 0 from the bottom carry of the tile row above
      {
        const int tn = tt_ + 1;
        const int ctn = tn % nct, rtn = tn / nct;
        const bool lastrn = rtn == nrt - 1;
        const int ix0n = S * (ctn * P.TW) - 1;
        const int v4 = tid & 7;
        if (ctn > 0)
          for (int r = tid >> 3; r < g.IH; r += R8_THREADS / 8)
            if (r < RL || lastrn) *reinterpret_cast<f32x4*>(dxs + (r * g.IW) * RC_CB + 4 * v4) = *reinterpret_cast<f32x4*>(rcar + r * RC_CB + 4 * v4);
        if (rtn > 0)
          for (int q = (tid >> 3) + (ctn > 0 ? 1 : 0); q < g.IW; q += R8_THREADS / 8)
            *reinterpret_cast<f32x4*>(dxs + q * RC_CB + 4 * v4) = *reinterpret_cast<f32x4*>(bcar + (ix0n + q + 1) * RC_CB + 4 * v4);
      }
    }
  }

  // ---- flush the per-image accumulators (the lane's own taps: nothing to add across the halves) -------------------------
  if constexpr (MODE == R8_B) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(xs);          // [8 waves][18 + 81][32] = 101 KB from xs on (the tiles are dead: the last tile is done)
    constexpr int RW = 18 + 81;
#pragma unroll
    for (int s = 0; s < R8_NT; ++s)
      if (tv_[s]) {
        float q2 = s2[s];
        if (s < SA1) {
          q2 = 0.f;
#pragma unroll
          for (int uu = 0; uu < 9; ++uu) q2 = __builtin_fmaf(wv[s][uu], dwa[s * 9 + uu], q2);
        }
        red[(wave * RW + tt[s]) * RC_CB + c] = s1[s];
        red[(wave * RW + 9 + tt[s]) * RC_CB + c] = q2;
        if (s < SA1) {
#pragma unroll
          for (int uu = 0; uu < 9; ++uu) red[(wave * RW + 18 + tt[s] * 9 + uu) * RC_CB + c] = dwa[s * 9 + uu];
        }
      }
    __syncthreads();
    // sums stripe of this image: [2][9*C] in [t*C + c] order
    for (int i = tid; i < 18 * RC_CB; i += R8_THREADS) {
      const int q = i / RC_CB, cc = i - q * RC_CB;
      float sv = 0.f;
#pragma unroll
      for (int w_ = 0; w_ < R8_NW; ++w_) sv += red[w_ * RW * RC_CB + i];
      P.sums[(long)n * (18 * P.C) + (long)(q / 9) * (9 * P.C) + (q % 9) * P.C + c0 + cc] = sv;
    }
    // A row of this image, generate.0.weight layout [c*9 + t][u'] (ly_rf3c_dwg_finish turns the image sum into the weight gradient)
    for (int i = tid; i < 81 * RC_CB; i += R8_THREADS) {
      const int cc = i / 81, e = i - cc * 81;
      const int t = e / 9, sl = t >= R8_NT ? t - R8_NT : t;                 // the tap's slot in its half wave
      if (sl >= SA1) continue;                                              // (pass C's share)
      const int k = (18 + e) * RC_CB + cc;
      float sv = 0.f;
#pragma unroll
      for (int w_ = 0; w_ < R8_NW; ++w_) sv += red[w_ * RW * RC_CB + k];
      P.dwg[(long)n * (P.C * 81) + (long)(c0 + cc) * 81 + e] = sv;
    }
  }
  if constexpr (MODE == R8_C && SA1 > SA0) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(xs);          // [8 waves][81][32] = 83 KB from xs on (dead: the last tile is done; dxs lies beyond)
#pragma unroll
    for (int s = SA0; s < SA1; ++s)
      if (tv_[s]) {
#pragma unroll
        for (int uu = 0; uu < 9; ++uu) red[(wave * 81 + tt[s] * 9 + uu) * RC_CB + c] = dwa[(s - SA0) * 9 + uu];
      }
    __syncthreads();
    for (int i = tid; i < 81 * RC_CB; i += R8_THREADS) {
      const int cc = i / 81, e = i - cc * 81;
      const int t = e / 9, sl = t >= R8_NT ? t - R8_NT : t;
      if (sl < SA0) continue;                                               // (pass B's share)
      const int k = e * RC_CB + cc;
      float sv = 0.f;
#pragma unroll
      for (int w_ = 0; w_ < R8_NW; ++w_) sv += red[w_ * 81 * RC_CB + k];
      P.dwg[(long)n * (P.C * 81) + (long)(c0 + cc) * 81 + e] = sv;
    }
  }
}

template <int MODE, int KS>
static int r8_launch(const LyRf3cBwdParams& P, hipStream_t st) {
  const int nct = (P.Wo + P.TW - 1) / P.TW, nrt = (P.Ho + P.TH - 1) / P.TH;
  const int IH = 2 * (P.TH - 1) + 3, IW = 2 * (P.TW - 1) + 3;
  constexpr int O = 32 * KS;
  size_t lds = (size_t)IH * IW * RC_CB * 4 + RC_KR * 128 + RC_TP * (2 * O + 16) + (size_t)32 * 9 * 8 * 4 + 64 * 4;
  if (MODE == R8_C) lds += ((size_t)IH * IW + IH + 2 * P.TW * nct + 2 + 27) * RC_CB * 4;
  LY_CHECK(lds <= 160 * 1024, "rf3c_bwd8: pass %d needs %zu B LDS for this shape (tile %dx%d, W = %d, O = %d)", MODE, lds, P.TH, P.TW, P.W, O);
  // pass B's final reduction reuses the tile area: [8 waves][18 + 81][32] floats from xs on
  if (MODE == R8_B && lds < (size_t)R8_NW * 99 * RC_CB * 4) lds = (size_t)R8_NW * 99 * RC_CB * 4;
  auto k = ly_rf3c_bwd8_kernel<MODE, KS>;
  static LyDevOnce once;
  if (once.need()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    LY_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(k, dim3((unsigned)(P.n_img * (P.C / RC_CB))), dim3(R8_THREADS), lds, st, P, nct, nrt);
  LY_LAUNCH_CHECK();
  return 0;
}

template <int MODE>
static int r8_dispatch(const LyRf3cBwdParams& P, hipStream_t st) {
  if (P.O == 64) return r8_launch<MODE, 2>(P, st);
  if (P.O == 128) return r8_launch<MODE, 4>(P, st);
  return r8_launch<MODE, 8>(P, st);
}

// passes B (pass = 1) and C (pass = 2) of ly_rf3c_bwd on eight waves, taps split over the half waves; same parameters.  Pass B also leaves
// dwg[n][c*81 + t*9 + u'] = A (see the header), pass C writes dx only
extern "C" int ly_rf3c_bwd8(const LyRf3cBwdParams* p, int pass, void* stream) {
  LY_CHECK(p, "rf3c_bwd8: null params");
  const LyRf3cBwdParams& P = *p;
  LY_CHECK(P.dtype == LY_BF16, "rf3c_bwd8: built for bf16 storage (dtype %d)", P.dtype);
  LY_CHECK(P.s == 2 && (P.O == 64 || P.O == 128 || P.O == 256), "rf3c_bwd8: built for stride 2 and 64 / 128 / 256 output channels (s = %d, O = %d)", P.s, P.O);
  if (rc_check_tile("rf3c_bwd8", P.C, P.s, P.TH, P.TW, P.ldx, P.x)) return -1;
  LY_CHECK(P.x && P.du && P.wq && P.wct && P.ca && P.rfa && (P.lddu & 7) == 0 && ((uintptr_t)P.du & 15) == 0, "rf3c_bwd8: null / misaligned pointer");
  LY_CHECK((long)P.n_img * P.H * P.W * P.ldx < (1L << 31), "rf3c_bwd8: input exceeds the 31-bit offsets of the staging plan");
  LY_CHECK(pass != R8_C || (long)P.n_img * P.H * P.W * P.lddx < (1L << 31), "rf3c_bwd8: dx exceeds 31-bit offsets");
  LY_CHECK(P.Ho == (P.H + 2 - 3) / P.s + 1 && P.Wo == (P.W + 2 - 3) / P.s + 1, "rf3c_bwd8: inconsistent output size");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (pass == R8_B) {
    LY_CHECK(P.mm && P.d_mm && P.sums && P.dwg, "rf3c_bwd8: pass B needs mm, d_mm, sums and dwg (the A rows)");
    return r8_dispatch<R8_B>(P, st);
  }
  LY_CHECK(pass == R8_C && P.mm && P.d_mm && P.coef && P.dwg && P.dx && (P.lddx & 7) == 0 && ((uintptr_t)P.dx & 15) == 0, "rf3c_bwd8: pass C needs mm, d_mm, coef, dwg and an aligned dx");
  return r8_dispatch<R8_C>(P, st);
}

// d(generate.0.weight)[c][t][u'] (+)= alpha[t,c] * A[c][t][u'] + kappa[t,c] * m[c][u'] + lambda[t,c] * sum_u'' w[c][t][u''] * M[c][u''][u']
//   A: [C*81] the image sum of pass B's rows; coef: alpha | kappa | lambda, 9*C floats each in [t*C + c] order (ly_bn_bwd_coeffs);
//   mom: the forward's tap moments, double [54][C]: rows 0..8 = m[u'], rows 9..53 = the upper triangle of M in row-major order (ly_rfcbam_tap_moments);
//   w: generate.0.weight [C*9, 1, 3, 3] = [c][t][u''].
__global__ __launch_bounds__(256) void ly_rf3c_dwg_finish_kernel(const float* __restrict__ A, const float* __restrict__ coef, const double* __restrict__ mom,
                                                                const float* __restrict__ w, const int C, float* __restrict__ out, const int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= C * 81) return;
  const int c = i / 81, r = i - c * 81, t = r / 9, up = r - t * 9;
  const double al = coef[t * C + c], ka = coef[(9 + t) * C + c], la = coef[(18 + t) * C + c];
  double wm = 0.0;
#pragma unroll
  for (int u2 = 0; u2 < 9; ++u2) {
    const int a = u2 < up ? u2 : up, b = u2 < up ? up : u2;          // M is symmetric: entry (a <= b) of the upper triangle
    const int tri = a * 9 - a * (a - 1) / 2 + (b - a);
    wm += (double)w[(c * 9 + t) * 9 + u2] * mom[(long)(9 + tri) * C + c];
  }
  const float v = (float)(al * (double)A[i] + ka * mom[(long)up * C + c] + la * wm);
  out[i] = accumulate ? out[i] + v : v;
}

extern "C" int ly_rf3c_dwg_finish(const float* A, const float* coef, const double* mom, const float* w, int C, float* out, int accumulate, void* stream) {
  LY_CHECK(A && coef && mom && w && out && C > 0, "rf3c_dwg_finish: bad arguments");
  hipLaunchKernelGGL(ly_rf3c_dwg_finish_kernel, dim3((C * 81 + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), A, coef, mom, w, C, out, accumulate);
  LY_LAUNCH_CHECK();
  return 0;
}
