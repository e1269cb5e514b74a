// RFCBAMConv kernel_size 3 forward on the lane = channel core (ly_rf3c.cuh); reference models/rfa.py:113-129.
//
//   ly_rf3c_stats : [max_c, mean_c] of G = relu(bn(generate(x)))  ->  mm[n, 3Ho, 3Wo, 2]   (models/rfa.py:125-126)
//                   + the SE global-average-pool partials part[n][tile][C]             (models/rfa.py:90)   -- x is read ONCE for both
//   ly_rf3c_fwd   : out = relu(bn(conv_{3x3, stride 3}(G * ca * rfa)))                     (models/rfa.py:124, 128-129)
// Both regenerate G per (64-pixel tile, 32-channel chunk) on the VALU with lane = channel; nothing 9x-sized exists in HBM.
#include "ly_rf3c.cuh"
#include "ly_params.h"

// ---------------------------------------------------------------------------------------------------
// statistics pass.  LDS: x tile fp32 [IH*IW][32] | G tile fp32 [288][66] | red [4][32]
// ---------------------------------------------------------------------------------------------------
#define RC_GS 66          // floats per row of the fp32 G tile: 2 (mod 32) => the qword stores of 16 consecutive rows and the row reads are conflict-free

template <typename T>
__global__ __launch_bounds__(LY_THREADS) void ly_rf3c_stats_kernel(const T* __restrict__ x, int ldx, int H, int W, int C, int Ho, int Wo, int s,
                                                                   int TH, int TW, int nct, int nrt, const float* __restrict__ wq,
                                                                   float* __restrict__ mm, float* __restrict__ part) {
  extern __shared__ f32x4 rc_smem4[];
  const RcGeom g = rc_geom(s, TH, TW);
  float* xs = reinterpret_cast<float*>(rc_smem4);
  float* gt = xs + g.IH * g.IW * RC_CB;
  float* red = gt + RC_KR * RC_GS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, c = lane & 31;
  int b = blockIdx.x;
  const int ct = b % nct; b /= nct;
  const int rt = b % nrt;
  const int n = b / nrt;
  const int oy0 = rt * TH, ox0 = ct * TW;

  RcStage<T> S;
  rc_stage_plan(S, g, tid, n, H, W, ldx, s * oy0 - 1, s * ox0 - 1);
  rc_stage_load(S, x, 0);

  // reduce phase: lane = tile pixel, wave w takes taps w, w+4, w+8
  const int rly = lane / TW, rlx = lane - rly * TW;
  const bool ractive = lane < g.NPX && oy0 + rly < Ho && ox0 + rlx < Wo;
  float mx[3], sm[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) { mx[i] = 0.f; sm[i] = 0.f; }      // G >= 0: zero is the identity of the channel max

  const int stream = wave * 2 + half;
  for (int c0 = 0; c0 < C; c0 += RC_CB) {
    RcW w;
    rc_load_w(w, wq + (long)(c0 + c) * RC_WQ);
    __syncthreads();                              // previous chunk: reduce done with gt / red, generate done with xs
    rc_stage_store(S, xs);
    rc_stage_load(S, x, c0 + RC_CB < C ? c0 + RC_CB : 0);
    __syncthreads();
    f32x2 gap = {0.f, 0.f};
#pragma unroll 1
    for (int j = 0; j < 4; ++j) {
      const int px0 = 8 * stream + 2 * j;
      const bool pin = px0 < g.NPX;
      const int pxc = pin ? px0 : 0;
      const int ly = pxc / TW, lx = pxc - ly * TW;
      f32x2 xv[9], a[9];
      rc_patch(xs, g, (s * ly) * g.IW + s * lx, c, xv);
      rc_generate<true>(w, xv, a);
      // SE pooling: the pixel OWNS inputs (s*oy + dy, s*ox + dx), dy, dx < s  = patch offsets (1 + dy, 1 + dx); every input belongs to one pixel
      f32x2 own = xv[4];
      if (s == 2) own += xv[5] + xv[7] + xv[8];
      gap += pin ? own : (f32x2){0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const f32x2 gg = {fmaxf(a[t][0], 0.f), fmaxf(a[t][1], 0.f)};
        *reinterpret_cast<f32x2*>(gt + (t * RC_CB + c) * RC_GS + px0) = gg;
      }
    }
    {
      float gsum = gap[0] + gap[1];
      gsum += __shfl_xor(gsum, 32);
      if (half == 0) red[wave * RC_CB + c] = gsum;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int t = wave + 4 * i;
      if (t < 9) {
        const float* col = gt + (t * RC_CB) * RC_GS + lane;
        float m0 = mx[i], s0 = sm[i];
#pragma unroll 8
        for (int cc = 0; cc < RC_CB; ++cc) {
          const float v = col[cc * RC_GS];
          m0 = fmaxf(m0, v);
          s0 += v;
        }
        mx[i] = m0; sm[i] = s0;
      }
    }
    if (part && tid < RC_CB)
      part[((long)n * (nrt * nct) + rt * nct + ct) * C + c0 + tid] = (red[tid] + red[RC_CB + tid]) + (red[2 * RC_CB + tid] + red[3 * RC_CB + tid]);
  }
  if (ractive) {
    const float inv = 1.f / (float)C;
    const int HK = 3 * Ho, WK = 3 * Wo;
    const int oy = oy0 + rly, ox = ox0 + rlx;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int t = wave + 4 * i;
      if (t < 9) {
        const long o = (((long)n * HK + 3 * oy + t / 3) * WK + 3 * ox + t % 3) * 2;
        *reinterpret_cast<f32x2*>(mm + o) = (f32x2){mx[i], sm[i] * inv};
      }
    }
  }
}

static size_t rc_stats_lds(int s, int TH, int TW) {
  const int IH = s * (TH - 1) + 3, IW = s * (TW - 1) + 3;
  return sizeof(float) * ((size_t)IH * IW * RC_CB + (size_t)RC_KR * RC_GS + 4 * RC_CB);
}

static int rc_check_tile(const char* who, int C, int s, int TH, int TW, int ldx, const void* x) {
  LY_CHECK(C > 0 && (C % RC_CB) == 0, "%s: C=%d must be a multiple of %d", who, C, RC_CB);
  LY_CHECK(s == 1 || s == 2, "%s: stride %d is not built (1 or 2)", who, s);
  LY_CHECK(TH >= 1 && TW >= 2 && (TW & 1) == 0 && TH * TW <= RC_TP, "%s: bad tile %dx%d (TW even, TH*TW <= %d)", who, TH, TW, RC_TP);
  const int IH = s * (TH - 1) + 3, IW = s * (TW - 1) + 3;
  LY_CHECK(IH * IW <= RC_MAXPOS, "%s: the %dx%d tile reads %d input positions (max %d)", who, TH, TW, IH * IW, RC_MAXPOS);
  LY_CHECK((ldx & 7) == 0 && ((uintptr_t)x & 15) == 0, "%s: x must be 16-byte aligned with a row stride that is a multiple of 8", who);
  return 0;
}

extern "C" int ly_rf3c_stats(const void* x, int ldx, int n_img, int H, int W, int C, int s, const float* wq, int TH, int TW, float* mm,
                             float* part, int slices, int dtype, void* stream) {
  LY_CHECK_DTYPE(dtype, "rf3c_stats");
  LY_CHECK(x && wq && mm && n_img > 0, "rf3c_stats: null pointer");
  if (rc_check_tile("rf3c_stats", C, s, TH, TW, ldx, x)) return -1;
  LY_CHECK((long)n_img * H * W * ldx < (1L << 31), "rf3c_stats: input exceeds the 31-bit offsets of the staging plan");
  const int Ho = (H + 2 - 3) / s + 1, Wo = (W + 2 - 3) / s + 1;
  const int nct = (Wo + TW - 1) / TW, nrt = (Ho + TH - 1) / TH;
  LY_CHECK(!part || slices == nct * nrt, "rf3c_stats: the pooling partials are one row per tile: slices must be %d", nct * nrt);
  const size_t lds = rc_stats_lds(s, TH, TW);
  LY_CHECK(lds <= 160 * 1024, "rf3c_stats: tile needs %zu B LDS", lds);
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ly_rf3c_stats_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    LY_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(ly_rf3c_stats_kernel<__bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    LY_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    configured = true;
  }
  LY_WITH_T(dtype, hipLaunchKernelGGL(ly_rf3c_stats_kernel<T>, dim3((unsigned)(n_img * nrt * nct)), dim3(LY_THREADS), lds, reinterpret_cast<hipStream_t>(stream),
                                      reinterpret_cast<const T*>(x), ldx, H, W, C, Ho, Wo, s, TH, TW, nct, nrt, wq, mm, part));
  LY_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// main contraction.  LDS: x tile fp32 [IH*IW][32] | G' planes bf16 [PL][288][64] | rfa [64][12]
//   per chunk: generate (VALU, lane = channel) -> G' = G * ca * rfa as the K-major operand tile -> 9 k-steps (one per tap) of MFMAs
//   against conv.0.weight packed as [N][C/32][9 taps][32 channels]; the next chunk's input tile, generate weights and the conv
//   weight fragments (register ring) are requested a phase ahead.
// ---------------------------------------------------------------------------------------------------
template <typename T, int MT, int NW>
__global__ __launch_bounds__(NW * 64) void ly_rf3c_fwd_kernel(const LyRfcbam3Params P, const float* __restrict__ wq, const int gy, const int nct,
                                                                 const int nrt) {
  using TR = LyT<T>;
  constexpr int PL = TR::PL;
  const T* const x = reinterpret_cast<const T*>(P.x);
  T* const out = reinterpret_cast<T*>(P.out);
  extern __shared__ f32x4 rc_smem4[];
  const RcGeom g = rc_geom(P.s, P.TH, P.TW);
  float* xs = reinterpret_cast<float*>(rc_smem4);
  char* gs_hi = reinterpret_cast<char*>(xs + g.IH * g.IW * RC_CB);
  char* gs_lo = gs_hi + (PL - 1) * RC_KR * 128;
  float* rfs = reinterpret_cast<float*>(gs_hi + PL * RC_KR * 128);       // [64][12]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, c = lane & 31;
  const int li = lane & 15, lq = lane >> 4;
  int b = blockIdx.x;
  const int by = b % gy; b /= gy;
  const int ct = b % nct; b /= nct;
  const int rt = b % nrt;
  const int n = b / nrt;
  const int oy0 = rt * P.TH, ox0 = ct * P.TW;
  const f32x4 zero = ly_zero4();
  const int NCH = P.C / RC_CB;
  const int S = NCH * 9;                                  // k-steps of the packed conv weight
  const int Tt = (P.N + 15) >> 4;

  constexpr int NTHR = NW * 64;
  RcStage<T, NTHR> St;
  rc_stage_plan(St, g, tid, n, P.H, P.W, P.ldx, P.s * oy0 - 1, P.s * ox0 - 1);
  rc_stage_load(St, x, 0);

  // rfa of the tile's pixels, zero for pixels outside the map: G' = 0 there
  for (int i = tid; i < RC_TP * 9; i += NTHR) {
    const int px = i / 9, t = i - px * 9;
    const int ly = px / P.TW, lx = px - ly * P.TW;
    const int oy = oy0 + ly, ox = ox0 + lx;
    const bool ok = px < g.NPX && oy < P.Ho && ox < P.Wo;
    rfs[px * 12 + t] = ok ? P.rfa[((long)n * 3 * P.Ho + 3 * oy + t / 3) * (3 * P.Wo) + 3 * ox + t % 3] : 0.f;
  }

  f32x4 acc[MT][4];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[t][j] = zero;
  int tile[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int tt = (by * NW + wave) * MT + t;
    tile[t] = tt < Tt ? tt : Tt - 1;
  }
  const uint4* wpk = reinterpret_cast<const uint4*>(P.wp);
  constexpr int NF = 9 * MT;                               // fragments per chunk, k-step major
  constexpr int D = MT == 1 ? 3 : 6;                       // ring depth; NF % D == 0: the slot of a fragment does not depend on the chunk
  static_assert(NF % D == 0, "ring");
  LyWF<PL> ring[D];
  auto wfrag_at = [&](int sb, int q) -> LyWF<PL> { return ly_wfragp<PL>(wpk, (long)tile[q % MT] * S + sb + q / MT, lane); };
#pragma unroll
  for (int q = 0; q < D; ++q) ring[q] = wfrag_at(0, q);

  const RcTr tr = rc_tr_plan(lane);
  const int stream = wave * 2 + half;
  const int csw = rc_sw(c);                                 // rows k = t*32 + c: the swizzle depends on c only

  RcW w;
  rc_load_w(w, wq + (long)c * RC_WQ);
  float cav = P.ca[(long)n * P.C + c];

  for (int ch = 0; ch < NCH; ++ch) {
    const bool more = ch + 1 < NCH;
    __syncthreads();                                // previous chunk: MFMAs done with G', generate done with xs
    rc_stage_store(St, xs);
    rc_stage_load(St, x, more ? (ch + 1) * RC_CB : 0);
    __syncthreads();
    // ---- regenerate: G' = relu(v) * ca * rfa for the stream's 4 pixel pairs ----------------------------------
#pragma unroll 1
    for (int j = 0; j < 16 / NW; ++j) {
      const int px0 = (32 / NW) * stream + 2 * j;
      const int pxc = px0 < g.NPX ? px0 : 0;
      const int ly = pxc / P.TW, lx = pxc - ly * P.TW;
      f32x2 xv[9], a[9];
      rc_patch(xs, g, (P.s * ly) * g.IW + P.s * lx, c, xv);
      rc_generate<true>(w, xv, a);
      f32x4 r0[3], r1[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        r0[i] = *reinterpret_cast<const f32x4*>(rfs + px0 * 12 + 4 * i);
        r1[i] = *reinterpret_cast<const f32x4*>(rfs + (px0 + 1) * 12 + 4 * i);
      }
      const int goff = c * 128 + ((((px0 >> 2) ^ csw) << 3) | ((px0 & 2) << 1));
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const f32x2 gg = (f32x2){fmaxf(a[t][0], 0.f) * (cav * r0[t >> 2][t & 3]), fmaxf(a[t][1], 0.f) * (cav * r1[t >> 2][t & 3])};
        if constexpr (PL == 2) {
          const bf16x2 hi = __builtin_convertvector(gg, bf16x2);
          const f32x2 back = __builtin_convertvector(hi, f32x2);
          *reinterpret_cast<unsigned*>(gs_hi + t * (RC_CB * 128) + goff) = __builtin_bit_cast(unsigned, hi);
          *reinterpret_cast<unsigned*>(gs_lo + t * (RC_CB * 128) + goff) = rc_pack2(gg - back);
        } else {
          *reinterpret_cast<unsigned*>(gs_hi + t * (RC_CB * 128) + goff) = rc_pack2(gg);
        }
      }
    }
    // the next chunk's generate weights: in flight during the contraction (the last chunk re-requests chunk 0: no load under a branch)
    {
      const int cn = more ? (ch + 1) * RC_CB : 0;
      rc_load_w(w, wq + (long)(cn + c) * RC_WQ);
      cav = P.ca[(long)n * P.C + cn + c];
    }
    __syncthreads();
    // ---- contract the chunk: 9 k-steps (k = tap*32 + channel) -------------------------------------------------
    const int sbase = ch * 9;
#pragma unroll
    for (int st = 0; st < 9; ++st) {
      bf16x8 xh[4], xl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xh[j] = rc_tr_frag(gs_hi, tr, st * 32, j);
        if constexpr (PL == 2) xl[j] = rc_tr_frag(gs_lo, tr, st * 32, j);
        else xl[j] = xh[j];
      }
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const int q = st * MT + t;
        const LyWF<PL> wf = ring[q % D];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = ly_mfmap<PL>(wf, xh[j], xl[j], acc[t][j]);
        ring[q % D] = q + D < NF ? wfrag_at(sbase, q + D) : wfrag_at(more ? sbase + 9 : 0, q + D - NF);
        __builtin_amdgcn_sched_barrier(0x786);     // neither loads nor MFMAs may move across: the refills stay D fragments ahead
      }
    }
  }

  // ---- epilogue: conv.0 bias + conv.1 BatchNorm + ReLU (or the statistics pass / pre-BN value of the training forward) ----
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int tt = (by * NW + wave) * MT + t;
    const int cc = 16 * tt + 4 * lq;
    if (tt >= Tt || cc >= P.N) continue;
    float sc[4], sh[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = cc + r < P.N;
      sc[r] = ok ? P.e_scale[cc + r] : 1.f;
      sh[r] = ok ? P.e_shift[cc + r] : 0.f;
    }
    f32x4 s1 = zero, s2 = zero;
    const float lin_floor = P.linear ? -INFINITY : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pl = 16 * j + li;
      const int py = pl / P.TW, px = pl - py * P.TW;
      const int yy = oy0 + py, xx = ox0 + px;
      if (pl >= g.NPX || yy >= P.Ho || xx >= P.Wo) continue;
      f32x4 v;
      if (P.stats) {
        f32x4 u;
#pragma unroll
        for (int r = 0; r < 4; ++r) u[r] = acc[t][j][r] * sc[r] + sh[r];
        s1 += u;
        s2 += u * u;
        if (!out) continue;
        v = u;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(acc[t][j][r] * sc[r] + sh[r], lin_floor);
      }
      T* o = out + (((long)n * P.Ho + yy) * P.Wo + xx) * P.ldo + cc;
      if ((P.ldo & 3) == 0 && cc + 3 < P.N) {
        ly_st4<T>(o, v);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (cc + r < P.N) ly_st1<T>(o + r, v[r]);
      }
    }
    if (P.stats) ly_stats_flush(P.stats, P.N, cc, s1, s2);
  }
}

template <typename T, int MT, int NW>
static int rc_launch_fwd(const LyRfcbam3Params& P, const float* wq, hipStream_t st) {
  const int nct = (P.Wo + P.TW - 1) / P.TW, nrt = (P.Ho + P.TH - 1) / P.TH;
  const int gy = (P.N + 16 * NW * MT - 1) / (16 * NW * MT);
  const int IH = P.s * (P.TH - 1) + 3, IW = P.s * (P.TW - 1) + 3;
  const size_t lds = sizeof(float) * ((size_t)IH * IW * RC_CB + RC_TP * 12) + (size_t)LyT<T>::PL * RC_KR * 128;
  LY_CHECK(lds <= 160 * 1024, "rf3c_fwd: tile needs %zu B LDS", lds);
  auto k = ly_rf3c_fwd_kernel<T, MT, NW>;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    LY_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    configured = true;
  }
  const long nb = (long)P.n_img * nrt * nct * gy;
  LY_CHECK(nb < (1L << 31), "rf3c_fwd: grid too large");
  hipLaunchKernelGGL(k, dim3((unsigned)nb), dim3(NW * 64), lds, st, P, wq, gy, nct, nrt);
  LY_LAUNCH_CHECK();
  return 0;
}

template <typename T>
static int rc_dispatch_fwd(const LyRfcbam3Params& P, const float* wq, hipStream_t st) {
  // N <= 128: 4 waves x (MT x 16) output channels, two blocks per CU (bf16); wider: 8 waves share ONE regenerated tile (the VALU phase is
  // split over 16 pixel streams, the contraction over 8 x 32 output channels) instead of regenerating it per 128-channel group
  if constexpr (LyT<T>::BF) {
    if (P.N > 128) return rc_launch_fwd<T, 2, 8>(P, wq, st);
  }
  if (P.N > 64) return rc_launch_fwd<T, 2, 4>(P, wq, st);
  return rc_launch_fwd<T, 1, 4>(P, wq, st);
}

// P as for ly_rfcbam3_fwd with two differences: P.wg is ignored (wq = the lane-order generate weights [C][92]: w'[t][u], b'[t], 2 pad) and
// P.wp = conv.0.weight frag-packed as [N][C/32 chunks][9 taps][32 channels] (K = 9*C, no padding).
extern "C" int ly_rf3c_fwd(const LyRfcbam3Params* p, const float* wq, void* stream) {
  LY_CHECK(p && wq, "rf3c_fwd: null params");
  const LyRfcbam3Params& P = *p;
  LY_CHECK_DTYPE(P.dtype, "rf3c_fwd");
  LY_CHECK(P.x && P.ca && P.rfa && P.wp && P.e_scale && P.e_shift && (P.out || P.stats), "rf3c_fwd: null pointer");
  if (rc_check_tile("rf3c_fwd", P.C, P.s, P.TH, P.TW, P.ldx, P.x)) return -1;
  LY_CHECK((long)P.n_img * P.H * P.W * P.ldx < (1L << 31), "rf3c_fwd: input exceeds the 31-bit offsets of the staging plan");
  LY_CHECK(P.Ho == (P.H + 2 - 3) / P.s + 1 && P.Wo == (P.W + 2 - 3) / P.s + 1, "rf3c_fwd: inconsistent output size");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  return P.dtype == LY_BF16 ? rc_dispatch_fwd<__bf16>(P, wq, st) : rc_dispatch_fwd<float>(P, wq, st);
}
