// RFCBAMConv kernel_size 3 forward on the lane = channel core (ly_rf3c.hpp); reference models/rfa.py:113-129.
//
//   ly_rf3c_stats : [max_c, mean_c] of G = relu(bn(generate(x)))  ->  mm[n, 3Ho, 3Wo, 2]   (models/rfa.py:125-126)
//                   + the SE global-average-pool partials part[n][tile][C]             (models/rfa.py:90)   -- x is read ONCE for both
//   ly_rf3c_fwd   : out = relu(bn(conv_{3x3, stride 3}(G * ca * rfa)))                     (models/rfa.py:124, 128-129)
// Both regenerate G per (64-pixel tile, 32-channel chunk) on the VALU with lane = channel; nothing 9x-sized exists in HBM.
#include "ly_rf3c.hpp"
#include "ly_params.h"

// ---------------------------------------------------------------------------------------------------
// statistics pass.  LDS: x tile fp32 [IH*IW][32] | G tile fp32 [288][34] (HALF a tile: 32 pixels) | red [4][32]   (~76 KB: two blocks per CU)
//   per chunk and half tile: generate (lane = channel, 8 streams x 2 pixel pairs) -> G tile -> reduce with lane = pixel: thread group
//   g = tid / 32 folds the chunk's 32 channels of tap g into its running (max, sum), and channels 4g .. 4g+3 of tap 8 (the ninth tap is
//   shared by the 8 groups, which meet once, at the end).
// ---------------------------------------------------------------------------------------------------
#define RC_GS 34          // floats per row of the fp32 G tile: 2 (mod 32) => the qword stores of 16 consecutive rows and the row reads are conflict-free

template <typename T, int S, bool RAW>
__global__ __launch_bounds__(LY_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void ly_rf3c_stats_kernel(const T* __restrict__ x, int ldx, int H, int W, int C, int Ho, int Wo,
                                                                   int TH, int TW, int nct, int nrt, const float* __restrict__ wq,
                                                                   float* __restrict__ mm, float* __restrict__ part) {
  extern __shared__ f32x4 rc_smem4[];
  const RcGeom g = rc_geom(S, TH, TW);
  float* xs = reinterpret_cast<float*>(rc_smem4);
  float* gt = xs + g.IH * g.IW * RC_CB;
  float* red = gt + RC_KR * RC_GS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, c = lane & 31;
  int b = ly_xcd_remap((int)blockIdx.x, (int)gridDim.x);              // neighbouring tiles on one XCD's L2
  const int ct = b % nct; b /= nct;
  const int rt = b % nrt;
  const int n = b / nrt;
  const int oy0 = rt * TH, ox0 = ct * TW;

  RcStage<T> St;
  rc_stage_plan(St, g, tid, n, H, W, ldx, S * oy0 - 1, S * ox0 - 1);
  rc_stage_load(St, x, 0);

  // generate: stream (wave, half) handles, in half tile hp, the pixel pairs px0 = 32*hp + 4*stream + 2*j, j = 0, 1
  const int stream = wave * 2 + half;
  const int row = g.IW * RC_CB;
  const float* xp[2][2];
  bool pin[2][2];
#pragma unroll
  for (int hp = 0; hp < 2; ++hp)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int px0 = 32 * hp + 4 * stream + 2 * j;
      xp[hp][j] = xs + rc_pos0(g, px0) * RC_CB + c;
      pin[hp][j] = px0 < g.NPX;
    }
  // reduce: lane & 31 = pixel of the half tile, group = tid / 32: tap `grp` over all channels (a), tap 8 over channels 4*grp .. +3 (b)
  const int rpx = tid & 31, grp = tid >> 5;
  float mxa[2] = {0.f, 0.f}, sma[2] = {0.f, 0.f}, mxb[2] = {0.f, 0.f}, smb[2] = {0.f, 0.f};      // G >= 0: zero is the identity of the channel max

  RcW w;
  rc_load_w<RAW>(w, wq, 0, c);
  for (int c0 = 0; c0 < C; c0 += RC_CB) {
    const bool more = c0 + RC_CB < C;
    __syncthreads();                              // previous chunk: reduce done with gt / red, generate done with xs
    rc_stage_store(St, xs);
    rc_stage_load(St, x, more ? c0 + RC_CB : 0);
    __syncthreads();
    f32x2 gap = {0.f, 0.f};
#pragma unroll
    for (int hp = 0; hp < 2; ++hp) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x2 xv[9], a[9];
        rc_patch<S>(xp[hp][j], row, xv);
        rc_gen_bn<RAW>(w, xv, a);
        // SE pooling: the pixel OWNS inputs (S*oy + dy, S*ox + dx), dy, dx < S = patch offsets (1 + dy, 1 + dx); every input belongs to one pixel
        f32x2 own = xv[4];
        if constexpr (S == 2) own += xv[5] + xv[7] + xv[8];
        gap += pin[hp][j] ? own : (f32x2){0.f, 0.f};
        float* gp = gt + c * RC_GS + 4 * stream + 2 * j;
#pragma unroll
        for (int t = 0; t < 9; ++t) *reinterpret_cast<f32x2*>(gp + t * (RC_CB * RC_GS)) = (f32x2){rc_relu(a[t][0]), rc_relu(a[t][1])};
      }
      if (hp == 1) {
        // the next chunk's weights: in flight during the reduce and the next staging (the last chunk re-requests chunk 0: no load under a branch)
        rc_load_w<RAW>(w, wq, more ? c0 + RC_CB : 0, c);
        float gsum = gap[0] + gap[1];
        gsum += __shfl_xor(gsum, 32);
        if (half == 0) red[wave * RC_CB + c] = gsum;
      }
      __syncthreads();
      {
        const float* col = gt + (grp * RC_CB) * RC_GS + rpx;
        float m0 = mxa[hp], s0 = sma[hp];
#pragma unroll 8
        for (int cc = 0; cc < RC_CB; ++cc) {
          const float v = col[cc * RC_GS];
          m0 = fmaxf(m0, v);
          s0 += v;
        }
        mxa[hp] = m0; sma[hp] = s0;
        const float* col8 = gt + (8 * RC_CB + 4 * grp) * RC_GS + rpx;
        m0 = mxb[hp]; s0 = smb[hp];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const float v = col8[cc * RC_GS];
          m0 = fmaxf(m0, v);
          s0 += v;
        }
        mxb[hp] = m0; smb[hp] = s0;
      }
      if (hp == 0) __syncthreads();               // the second half tile overwrites gt
    }
    if (part && tid < RC_CB)
      part[((long)n * (nrt * nct) + rt * nct + ct) * C + c0 + tid] = (red[tid] + red[RC_CB + tid]) + (red[2 * RC_CB + tid] + red[3 * RC_CB + tid]);
  }
  // taps 0..7 leave from their group's registers; the ninth tap's 8 partials meet in LDS (the G tile's memory)
  __syncthreads();
  float* fin = gt;                                 // [8 groups][2 half tiles][max, sum][32 px]
#pragma unroll
  for (int hp = 0; hp < 2; ++hp) {
    fin[((grp * 2 + hp) * 2) * 32 + rpx] = mxb[hp];
    fin[((grp * 2 + hp) * 2 + 1) * 32 + rpx] = smb[hp];
  }
  __syncthreads();
  const float inv = 1.f / (float)C;
  const int HK = 3 * Ho, WK = 3 * Wo;
#pragma unroll
  for (int hp = 0; hp < 2; ++hp) {
    const int px = 32 * hp + rpx;
    const int ly = px / TW, lx = px - ly * TW;
    const int oy = oy0 + ly, ox = ox0 + lx;
    if (px >= g.NPX || oy >= Ho || ox >= Wo) continue;
    const long o = (((long)n * HK + 3 * oy + grp / 3) * WK + 3 * ox + grp % 3) * 2;
    *reinterpret_cast<f32x2*>(mm + o) = (f32x2){mxa[hp], sma[hp] * inv};
    if (grp == 0) {
      float m0 = 0.f, s0 = 0.f;
#pragma unroll
      for (int gg = 0; gg < 8; ++gg) {
        m0 = fmaxf(m0, fin[((gg * 2 + hp) * 2) * 32 + rpx]);
        s0 += fin[((gg * 2 + hp) * 2 + 1) * 32 + rpx];
      }
      const long o8 = (((long)n * HK + 3 * oy + 2) * WK + 3 * ox + 2) * 2;
      *reinterpret_cast<f32x2*>(mm + o8) = (f32x2){m0, s0 * inv};
    }
  }
}

static size_t rc_stats_lds(int s, int TH, int TW) {
  const int IH = s * (TH - 1) + 3, IW = s * (TW - 1) + 3;
  return sizeof(float) * ((size_t)IH * IW * RC_CB + (size_t)RC_KR * RC_GS + 4 * RC_CB);
}

template <typename T, int S, bool RAW>
static int rc_launch_stats(const void* x, int ldx, int n_img, int H, int W, int C, int Ho, int Wo, int TH, int TW, int nct, int nrt, const float* wq,
                           float* mm, float* part, size_t lds, hipStream_t st) {
  auto k = ly_rf3c_stats_kernel<T, S, RAW>;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    LY_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    configured = true;
  }
  hipLaunchKernelGGL(k, dim3((unsigned)(n_img * nrt * nct)), dim3(LY_THREADS), lds, st, reinterpret_cast<const T*>(x), ldx, H, W, C, Ho, Wo, TH, TW, nct, nrt,
                     wq, mm, part);
  LY_LAUNCH_CHECK();
  return 0;
}

extern "C" int ly_rf3c_stats(const void* x, int ldx, int n_img, int H, int W, int C, int s, const float* wq, int raw, int TH, int TW, float* mm,
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
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define RC_ST(T_, S_, R_) rc_launch_stats<T_, S_, R_>(x, ldx, n_img, H, W, C, Ho, Wo, TH, TW, nct, nrt, wq, mm, part, lds, st)
  if (dtype == LY_BF16) return s == 1 ? (raw ? RC_ST(__bf16, 1, true) : RC_ST(__bf16, 1, false)) : (raw ? RC_ST(__bf16, 2, true) : RC_ST(__bf16, 2, false));
  return s == 1 ? (raw ? RC_ST(float, 1, true) : RC_ST(float, 1, false)) : (raw ? RC_ST(float, 2, true) : RC_ST(float, 2, false));
#undef RC_ST
}

// ---------------------------------------------------------------------------------------------------
// main contraction.  LDS: x tile fp32 [IH*IW][32] | G' planes bf16 [PL][288][64] | rfa [64][12]
//   per chunk: generate (VALU, lane = channel) -> G' = G * ca * rfa as the K-major operand tile -> 9 k-steps (one per tap) of MFMAs
//   against conv.0.weight packed as [N][C/32][9 taps][32 channels]; the next chunk's input tile, generate weights and the conv
//   weight fragments (register ring) are requested a phase ahead.
// ---------------------------------------------------------------------------------------------------
#ifdef RC_PHASE_PROF                    // development: cycles per phase of the forward contraction kernel, block 0 / thread 0 (tools/rf3c_phase_prof.py fwd)
__device__ unsigned long long rcf_prof[8];
#define RCF_T(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); rcf_acc[i] += t_ - rcf_t0; rcf_t0 = t_; } while (0)
extern "C" int ly_rf3c_fwd_prof(unsigned long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(rcf_prof), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rcf_prof), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#else
#define RCF_T(i) do { } while (0)
#endif

template <typename T, int MT, int NW, int S, bool RAW>
__global__ __launch_bounds__(NW * 64) void ly_rf3c_fwd_kernel(const LyRfcbam3Params P, const float* __restrict__ wq, const int gy, const int nct,
                                                                 const int nrt) {
  using TR = LyT<T>;
  constexpr int PL = TR::PL;
  const T* const x = reinterpret_cast<const T*>(P.x);
  T* const out = reinterpret_cast<T*>(P.out);
  extern __shared__ f32x4 rc_smem4[];
  const RcGeom g = rc_geom(S, P.TH, P.TW);
  float* xs = reinterpret_cast<float*>(rc_smem4);
  char* gs_hi = reinterpret_cast<char*>(xs + g.IH * g.IW * RC_CB);
  char* gs_lo = gs_hi + (PL - 1) * RC_KR * 128;
  float* rfs = reinterpret_cast<float*>(gs_hi + PL * RC_KR * 128);       // [32 pixel pairs][9 taps][2]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, c = lane & 31;
  const int li = lane & 15, lq = lane >> 4;
  int b = ly_xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int by = b % gy; b /= gy;
  const int ct = b % nct; b /= nct;
  const int rt = b % nrt;
  const int n = b / nrt;
  const int oy0 = rt * P.TH, ox0 = ct * P.TW;
  const f32x4 zero = ly_zero4();
  const int NCH = P.C / RC_CB;
  const int KS = NCH * 9;                                 // k-steps of the packed conv weight
  const int Tt = (P.N + 15) >> 4;

  constexpr int NTHR = NW * 64;
#ifdef RC_PHASE_PROF
  unsigned long long rcf_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long rcf_t0 = __builtin_readcyclecounter();
#endif
  RcStage<T, NTHR> St;
  rc_stage_plan(St, g, tid, n, P.H, P.W, P.ldx, S * oy0 - 1, S * ox0 - 1);
  rc_stage_load(St, x, 0);

  // rfa of the tile's pixels as pixel-pair rows [pair][tap][2], zero for pixels outside the map (G' = 0 there).  All loads of the thread are
  // issued before the first LDS store: one memory round trip in the block's prologue, not one per item
  {
    constexpr int NI = (RC_TP * 9 + NTHR - 1) / NTHR;
    float rv[NI];
#pragma unroll
    for (int e = 0; e < NI; ++e) {
      const int i = tid + e * NTHR;
      const int px = i / 9, t = i - px * 9;
      const int ly = px / P.TW, lx = px - ly * P.TW;
      const int oy = oy0 + ly, ox = ox0 + lx;
      const bool ok = i < RC_TP * 9 && px < g.NPX && oy < P.Ho && ox < P.Wo;
      const float v = P.rfa[ok ? ((long)n * 3 * P.Ho + 3 * oy + t / 3) * (3 * P.Wo) + 3 * ox + t % 3 : 0];
      rv[e] = ok ? v : 0.f;
    }
#pragma unroll
    for (int e = 0; e < NI; ++e) {
      const int i = tid + e * NTHR;
      const int px = i / 9, t = i - px * 9;
      if (i < RC_TP * 9) rfs[((px >> 1) * 9 + t) * 2 + (px & 1)] = rv[e];
    }
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
  constexpr int D = (MT == 1 || NW == 8) ? 3 : 6;          // ring depth; NF % D == 0: the slot of a fragment does not depend on the chunk (8-wave blocks: 256 registers)
  static_assert(NF % D == 0, "ring");
  LyWF<PL> ring[D];
  auto wfrag_at = [&](int sb, int q) -> LyWF<PL> { return ly_wfragp<PL>(wpk, (long)tile[q % MT] * KS + sb + q / MT, lane); };
#pragma unroll
  for (int q = 0; q < D; ++q) ring[q] = wfrag_at(0, q);

  const RcTr tr = rc_tr_plan(lane);
  const int stream = wave * 2 + half;
  const int csw = rc_sw(c);                                 // rows k = t*32 + c: the swizzle depends on c only

  constexpr int NJ = 16 / NW;                              // pixel pairs per stream: px0 = 2*NJ*stream + 2*j
  const int row = g.IW * RC_CB;
  const float* xp[NJ];
  int goff[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int px0 = 2 * NJ * stream + 2 * j;
    xp[j] = xs + rc_pos0(g, px0) * RC_CB + c;
    goff[j] = c * 128 + ((((px0 >> 2) ^ csw) << 3) | ((px0 & 2) << 1));
  }

  RcW w;
  rc_load_w<RAW>(w, wq, 0, c);
  float cav = P.ca[(long)n * P.C + c];

  RCF_T(0);                                         // prologue: plan, first loads, rfa table, weights
  for (int ch = 0; ch < NCH; ++ch) {
    const bool more = ch + 1 < NCH;
    __syncthreads();                                // previous chunk: MFMAs done with G', generate done with xs
    RCF_T(1);
    rc_stage_store(St, xs);
    rc_stage_load(St, x, more ? (ch + 1) * RC_CB : 0);
    __syncthreads();
    RCF_T(2);
    // ---- regenerate: G' = relu(v) * ca * rfa for the stream's pixel pairs ----------------------------------
    const f32x2 cav2 = {cav, cav};
    // (8-wave blocks: 256 registers per lane; the two pair iterations stay rolled and recompute their addresses)
#pragma unroll(NW == 8 ? 1 : NJ)
    for (int j = 0; j < NJ; ++j) {
      f32x2 xv[9], a[9];
      const float* xpj;
      int goffj;
      if constexpr (NW == 8) {
        const int px0 = 2 * NJ * stream + 2 * j;
        xpj = xs + rc_pos0(g, px0) * RC_CB + c;
        goffj = c * 128 + ((((px0 >> 2) ^ csw) << 3) | ((px0 & 2) << 1));
      } else {
        xpj = xp[j]; goffj = goff[j];
      }
      rc_patch<S>(xpj, row, xv);
      rc_gen_bn<RAW>(w, xv, a);
      const f32x2* rfp = reinterpret_cast<const f32x2*>(rfs) + (NJ * stream + j) * 9;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const f32x2 gg = (f32x2){rc_relu(a[t][0]), rc_relu(a[t][1])} * (rfp[t] * cav2);
        if constexpr (PL == 2) {
          const bf16x2 hi = __builtin_convertvector(gg, bf16x2);
          const f32x2 back = __builtin_convertvector(hi, f32x2);
          *reinterpret_cast<unsigned*>(gs_hi + t * (RC_CB * 128) + goffj) = __builtin_bit_cast(unsigned, hi);
          *reinterpret_cast<unsigned*>(gs_lo + t * (RC_CB * 128) + goffj) = rc_pack2(gg - back);
        } else {
          *reinterpret_cast<unsigned*>(gs_hi + t * (RC_CB * 128) + goffj) = rc_pack2(gg);
        }
      }
    }
    // the next chunk's generate weights: in flight during the contraction (the last chunk re-requests chunk 0: no load under a branch)
    {
      const int cn = more ? (ch + 1) * RC_CB : 0;
      rc_load_w<RAW>(w, wq, cn, c);
      cav = P.ca[(long)n * P.C + cn + c];
    }
    __syncthreads();
    RCF_T(3);
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
    RCF_T(4);
  }

#ifdef RC_PHASE_PROF
  if (blockIdx.x == 0 && threadIdx.x == 0 && MT == 2 && NW == 4)
    for (int i = 0; i < 8; ++i) rcf_prof[i] += rcf_acc[i];
#endif
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

template <typename T, int MT, int NW, int S, bool RAW>
static int rc_launch_fwd_s(const LyRfcbam3Params& P, const float* wq, hipStream_t st) {
  const int nct = (P.Wo + P.TW - 1) / P.TW, nrt = (P.Ho + P.TH - 1) / P.TH;
  const int gy = (P.N + 16 * NW * MT - 1) / (16 * NW * MT);
  const int IH = P.s * (P.TH - 1) + 3, IW = P.s * (P.TW - 1) + 3;
  const size_t lds = sizeof(float) * ((size_t)IH * IW * RC_CB + RC_TP * 9) + (size_t)LyT<T>::PL * RC_KR * 128;
  LY_CHECK(lds <= 160 * 1024, "rf3c_fwd: tile needs %zu B LDS", lds);
  auto k = ly_rf3c_fwd_kernel<T, MT, NW, S, RAW>;
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

template <typename T, int MT, int NW>
static int rc_launch_fwd(const LyRfcbam3Params& P, const float* wq, bool raw, hipStream_t st) {
  if (raw) return P.s == 1 ? rc_launch_fwd_s<T, MT, NW, 1, true>(P, wq, st) : rc_launch_fwd_s<T, MT, NW, 2, true>(P, wq, st);
  return P.s == 1 ? rc_launch_fwd_s<T, MT, NW, 1, false>(P, wq, st) : rc_launch_fwd_s<T, MT, NW, 2, false>(P, wq, st);
}

template <typename T>
static int rc_dispatch_fwd(const LyRfcbam3Params& P, const float* wq, bool raw, hipStream_t st) {
  // N <= 128: 4 waves x (MT x 16) output channels, two blocks per CU (bf16); wider: 8 waves share ONE regenerated tile (the VALU phase is
  // split over 16 pixel streams, the contraction over 8 x 32 output channels) instead of regenerating it per 128-channel group
  if constexpr (LyT<T>::BF) {
    if (P.N > 128) return rc_launch_fwd<T, 2, 8>(P, wq, raw, st);
  }
  if (P.N > 64) return rc_launch_fwd<T, 2, 4>(P, wq, raw, st);
  return rc_launch_fwd<T, 1, 4>(P, wq, raw, st);
}

// P as for ly_rfcbam3_fwd with two differences: P.wg is ignored (wq = the lane-order generate weights, raw != 0: the training form, see
// ly_rf3c.hpp) and P.wp = conv.0.weight frag-packed as [N][C/32 chunks][9 taps][32 channels] (K = 9*C, no padding).
extern "C" int ly_rf3c_fwd(const LyRfcbam3Params* p, const float* wq, int raw, void* stream) {
  LY_CHECK(p && wq, "rf3c_fwd: null params");
  const LyRfcbam3Params& P = *p;
  LY_CHECK_DTYPE(P.dtype, "rf3c_fwd");
  LY_CHECK(P.x && P.ca && P.rfa && P.wp && P.e_scale && P.e_shift && (P.out || P.stats), "rf3c_fwd: null pointer");
  if (rc_check_tile("rf3c_fwd", P.C, P.s, P.TH, P.TW, P.ldx, P.x)) return -1;
  LY_CHECK((long)P.n_img * P.H * P.W * P.ldx < (1L << 31), "rf3c_fwd: input exceeds the 31-bit offsets of the staging plan");
  LY_CHECK(P.Ho == (P.H + 2 - 3) / P.s + 1 && P.Wo == (P.W + 2 - 3) / P.s + 1, "rf3c_fwd: inconsistent output size");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  return P.dtype == LY_BF16 ? rc_dispatch_fwd<__bf16>(P, wq, raw != 0, st) : rc_dispatch_fwd<float>(P, wq, raw != 0, st);
}
