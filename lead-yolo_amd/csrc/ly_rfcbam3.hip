// RFCBAMConv main contraction for kernel_size 3 (reference models/rfa.py:113-129), gfx950; storage dtype T = float / __bf16
// (x and out; the regenerate phase is fp32 VALU work in both, the contraction bf16x3 or plain bf16: ly_tile.hpp).
//
//   out[n, o, oy, ox] = relu( bn( bias[o] + sum_{c, t} Wc[o, c, t] * G[n, c, 3oy+ty, 3ox+tx] * ca[n, c] * rfa[n, 3oy+ty, 3ox+tx] ) )
//   G[n, c, 3oy+ty, 3ox+tx] = relu( bn_{c*9+t}( sum_u Wd[c*9+t, u] * x[n, c, s*oy+uy-1, s*ox+ux-1] ) )
//
// The reference materialises G (9x the input), multiplies it twice, and runs a stride-3 conv over it
// (~13 passes over the 9x tensor).  Here G is REGENERATED on chip: per 16-channel chunk the block
// stages the raw input tile in LDS, the VALU recomputes the depthwise "generate" conv + BN + ReLU
// (81 MAC per channel and pixel, lane = output pixel, wave-uniform weights), scales by ca and rfa and
// writes the [64 px x 144] operand tile (bf16 hi/lo planes) to LDS, which the MFMAs contract with the
// bf16x3 frag-packed conv.0.weight viewed as [Cout, C/16, 144 -> 160] (k = t*16 + channel inside a chunk).  HBM sees x once and out once.
#include "ly_tile.hpp"
#include "ly_params.h"

#define LY_GCC 16                       // channels regenerated per chunk
#define LY_GK 160                        // 9*16 = 144 k-values per chunk, zero padded to 5 k-steps of 32
#define LY_RSG (2 * LY_GK + 8)           // bytes per operand row, per plane: 82 dwords, so that the 64 rows written by one 8-byte store per lane (regenerate) and the 16 rows x 4 k-groups of an MFMA operand read both spread over all banks
#define LY_RF3_NV 8                      // float4 staging items per thread per chunk: IH*IW*4 <= 8*256
#define LY_RF3_WF (9 * 2 * 20)           // floats of folded generate weights per wave per chunk: [9 t][2 pairs][9 x (w_a, w_b) + (b_a, b_b)]

// SW: the folded generate weights come through the scalar cache (SGPR operands) instead of LDS broadcasts; see the regenerate
// step.  Every chunk's weights are read once per block, so each tap waits for a scalar-cache miss that only a second resident
// wave can hide; the launcher's rule is in launch_rf3.
template <typename T, int MT, bool SW>
__device__ __forceinline__ void ly_rfcbam3_body(const LyRfcbam3Params& P, const int gy, const int nct, const int nrt) {
  using TR = LyT<T>;
  using R4 = typename TR::R4;
  constexpr int PL = TR::PL;
  const T* const x = reinterpret_cast<const T*>(P.x);
  T* const out = reinterpret_cast<T*>(P.out);
  extern __shared__ f32x4 ly_smem4[];
  const int s = P.s, TH = P.TH, TW = P.TW;
  const int IH = s * (TH - 1) + 3, IW = s * (TW - 1) + 3;
  char* gs_hi = reinterpret_cast<char*>(ly_smem4);          // [64][LY_RSG]
  char* gs_lo = gs_hi + (PL - 1) * 64 * LY_RSG;
  float* wsm = reinterpret_cast<float*>(gs_hi + PL * 64 * LY_RSG);  // !SW: [4 waves][LY_RF3_WF] this chunk's folded generate weights
  float* xs = wsm + (SW ? 0 : 4 * LY_RF3_WF);                  // [IH*IW][LY_GCC + 1]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: depthwise weights become scalar loads
  const int li = lane & 15, lq = lane >> 4;
  int b = blockIdx.x;
  const int by = b % gy; b /= gy;
  const int ct = b % nct; b /= nct;
  const int rt = b % nrt;
  const int n = b / nrt;
  const int oy0 = rt * TH, ox0 = ct * TW;
  const int ly = lane / TW, lx = lane - ly * TW;
  const int oy = oy0 + ly, ox = ox0 + lx;
  const bool active = ly < TH && oy < P.Ho && ox < P.Wo;
  const int iy0 = s * oy0 - 1, ix0 = s * ox0 - 1;
  const f32x4 zero = ly_zero4();
  const int S = (P.C / LY_GCC) * (LY_GK / 32);
  const int Tt = (P.N + 15) >> 4;

  float rf[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
    rf[t] = active ? P.rfa[((long)n * 3 * P.Ho + 3 * oy + t / 3) * (3 * P.Wo) + 3 * ox + t % 3] : 0.f;

  f32x4 acc[MT][4];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[t][j] = zero;
  int tile[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    int tt = (by * 4 + wave) * MT + t;
    tile[t] = tt < Tt ? tt : Tt - 1;
  }
  const uint4* wpk = reinterpret_cast<const uint4*>(P.wp);
  ly_l2_warm(P.wp, (long)Tt * S * PL * 1024, P.stats ? reinterpret_cast<float*>(P.stats) : reinterpret_cast<float*>(P.out));
  for (int i = tid; i < PL * 64 * LY_RSG / 16; i += LY_THREADS) reinterpret_cast<uint4*>(gs_hi)[i] = make_uint4(0u, 0u, 0u, 0u);

  // staging plan of this thread (independent of the channel chunk): global element offset (or -1), LDS slot
  const int items = IH * IW * (LY_GCC / 4);
  int soff[LY_RF3_NV];             // element offsets fit 31 bits (checked by the launcher)
  int doff[LY_RF3_NV];
#pragma unroll
  for (int e = 0; e < LY_RF3_NV; ++e) {
    const int idx = tid + e * LY_THREADS;
    int so = -1;
    int dd = -1;
    if (idx < items) {
      const int ip = idx >> 2, c4 = idx & 3;
      const int r = ip / IW, q = ip - r * IW;
      const int iy = iy0 + r, ix = ix0 + q;
      dd = ip * (LY_GCC + 1) + 4 * c4;
      if (iy >= 0 && iy < P.H && ix >= 0 && ix < P.W) so = (int)((((long)n * P.H + iy) * P.W + ix) * P.ldx + 4 * c4);
    }
    soff[e] = so; doff[e] = dd;
  }
  // Everything a chunk reads from global memory is requested one chunk ahead, so no wave ever waits on a load it has just
  // issued: the raw input tile (pv) and the conv weight fragments (ring: D fragments in
  // flight, slot q % D refilled with fragment q + D -- which may belong to the next chunk -- right after fragment q's MFMAs).
  R4 pv[LY_RF3_NV];
  auto prefetch = [&](int c0) {
#pragma unroll
    for (int e = 0; e < LY_RF3_NV; ++e) pv[e] = ly_ldr4<T>(soff[e] >= 0 ? x + soff[e] + c0 : x);
  };
  constexpr int WV = SW ? 1 : (4 * LY_RF3_WF / 4 + LY_THREADS - 1) / LY_THREADS;
  f32x4 wv[WV];
  auto wprefetch = [&](int chunk) {
    if constexpr (!SW) {
      const float* wsrc = P.wg + (long)chunk * (4 * LY_RF3_WF);
#pragma unroll
      for (int e = 0; e < WV; ++e) {
        const int i = tid + e * LY_THREADS;
        wv[e] = ly_ldg4(wsrc + 4 * (i < 4 * LY_RF3_WF / 4 ? i : 0));
      }
    }
  };
  constexpr int D = 5, NF = (LY_GK / 32) * MT;           // ring depth, fragments per chunk (k-step major); NF % D == 0
  static_assert(NF % D == 0, "the ring slot of a fragment must not depend on the chunk");
  LyWF<PL> ring[D];
  auto wfrag_at = [&](int sb, int q) -> LyWF<PL> { return ly_wfragp<PL>(wpk, (long)tile[q % MT] * S + sb + q / MT, lane); };
#pragma unroll
  for (int q = 0; q < D; ++q) ring[q] = wfrag_at(0, q);
  wprefetch(0);
  prefetch(0);

  for (int c0 = 0; c0 < P.C; c0 += LY_GCC) {
    __syncthreads();                      // previous chunk: generate done with xs, MFMAs done with gs
    const bool more = c0 + LY_GCC < P.C;
    if constexpr (!SW) {
#pragma unroll
      for (int e = 0; e < WV; ++e) {
        const int i = tid + e * LY_THREADS;
        if (i < 4 * LY_RF3_WF / 4) reinterpret_cast<f32x4*>(wsm)[i] = wv[e];
      }
      wprefetch(more ? c0 / LY_GCC + 1 : 0);
    }
#pragma unroll
    for (int e = 0; e < LY_RF3_NV; ++e)
      if (doff[e] >= 0) {
        const bool ok = soff[e] >= 0;
        const f32x4 pf = ly_r4_f32(pv[e]);
        float* d = xs + doff[e];
        d[0] = ok ? pf[0] : 0.f; d[1] = ok ? pf[1] : 0.f; d[2] = ok ? pf[2] : 0.f; d[3] = ok ? pf[3] : 0.f;
      }
    prefetch(more ? c0 + LY_GCC : 0);    // next chunk's input in flight during generate + MFMA; unconditional (the last chunk
                                         // re-requests chunk 0) so that the compiler's s_waitcnt counts stay exact
    __syncthreads();
    // ---- regenerate G' for channels c0 + 4*wave .. +3 -----------------------------------------------
    // inputs of the wave's 4 channels as 2 packed pairs (v_pk_fma_f32 does two channels per instruction).  The folded weights
    // [t][pair][20] (pack.rfcbam_gen_weights(..., 16, True)) are wave-uniform: LDS broadcasts (10 ds_read_b128 per tap and
    // wave, staged a chunk ahead) or, SW, SGPR operands through the scalar cache.  The LDS pipe -- shared by the resident
    // waves and also carrying the G' writes and the MFMA operand reads -- is what this kernel is bound by, so: the two pairs
    // run as two independent accumulation chains (no dependent-issue bubbles), and a lane's 4 channels of one tap are adjacent
    // in k (k = t*16 + channel), so G' is written with one 8-byte store per plane and tap instead of four 2-byte ones.
    {
      f32x2 xv[2][9];
      f32x2 cav[2];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        cav[p] = (f32x2){ly_const(P.ca)[(long)n * P.C + c0 + 4 * wave + 2 * p], ly_const(P.ca)[(long)n * P.C + c0 + 4 * wave + 2 * p + 1]};
#pragma unroll
        for (int u = 0; u < 9; ++u) {
          const int xo = ((s * ly + u / 3) * IW + (s * lx + u % 3)) * (LY_GCC + 1) + 4 * wave + 2 * p;
          xv[p][u] = active ? (f32x2){xs[xo], xs[xo + 1]} : (f32x2){0.f, 0.f};
        }
      }
      const ly_cfloat* wq = ly_const(P.wg) + (long)(c0 / LY_GCC) * (4 * LY_RF3_WF) + wave * LY_RF3_WF;
      const float* wl = wsm + wave * LY_RF3_WF;               // !SW: wave-uniform address => LDS broadcast reads
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        f32x2 a0, a1;
        if constexpr (SW) {
          const ly_cfloat* q0 = wq + (t * 2) * 20;
          const ly_cfloat* q1 = q0 + 20;
          a0 = (f32x2){q0[18], q0[19]};
          a1 = (f32x2){q1[18], q1[19]};
#pragma unroll
          for (int u = 0; u < 9; ++u) {
            a0 += xv[0][u] * (f32x2){q0[2 * u], q0[2 * u + 1]};
            a1 += xv[1][u] * (f32x2){q1[2 * u], q1[2 * u + 1]};
          }
        } else {
          f32x4 q0[5], q1[5];
#pragma unroll
          for (int i = 0; i < 5; ++i) {
            q0[i] = reinterpret_cast<const f32x4*>(wl + (t * 2) * 20)[i];
            q1[i] = reinterpret_cast<const f32x4*>(wl + (t * 2 + 1) * 20)[i];
          }
          a0 = (f32x2){q0[4][2], q0[4][3]};
          a1 = (f32x2){q1[4][2], q1[4][3]};
#pragma unroll
          for (int u = 0; u < 9; ++u) {
            a0 += xv[0][u] * (f32x2){q0[u >> 1][2 * (u & 1)], q0[u >> 1][2 * (u & 1) + 1]};
            a1 += xv[1][u] * (f32x2){q1[u >> 1][2 * (u & 1)], q1[u >> 1][2 * (u & 1) + 1]};
          }
        }
        const float rft = active ? rf[t] : 0.f;
        const f32x2 g0 = (f32x2){fmaxf(a0[0], 0.f), fmaxf(a0[1], 0.f)} * cav[0] * rft;
        const f32x2 g1 = (f32x2){fmaxf(a1[0], 0.f), fmaxf(a1[1], 0.f)} * cav[1] * rft;
        ly_lds_put_f32<PL>(gs_hi, gs_lo, lane * LY_RSG, t * LY_GCC + 4 * wave, (f32x4){g0[0], g0[1], g1[0], g1[1]});   // k = t*16 + (4*wave + j)
      }
    }
    __syncthreads();
    // ---- contract the chunk's 160 (144 real) k-values --------------------------------------------
    const int sbase = (c0 / LY_GCC) * (LY_GK / 32);
#pragma unroll
    for (int st = 0; st < LY_GK / 32; ++st) {
      bf16x8 xh[4], xl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xh[j] = ly_lds_frag(gs_hi, (16 * j + li) * LY_RSG, st, lq);
        if constexpr (PL == 2) xl[j] = ly_lds_frag(gs_lo, (16 * j + li) * LY_RSG, st, lq);
        else xl[j] = xh[j];
      }
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const int q = st * MT + t;
        const LyWF<PL> wf = ring[q % D];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = ly_mfmap<PL>(wf, xh[j], xl[j], acc[t][j]);
        // (the last chunk re-requests chunk 0's fragments instead of branching: a conditional load would make every later
        //  s_waitcnt assume it was not issued and wait for the loads behind it as well)
        ring[q % D] = q + D < NF ? wfrag_at(sbase, q + D) : wfrag_at(more ? sbase + LY_GK / 32 : 0, q + D - NF);
        __builtin_amdgcn_sched_barrier(0x786);     // neither loads nor MFMAs may move across: the refills stay D fragments ahead
      }
    }
  }

#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int tt = (by * 4 + wave) * MT + t;
    const int c = 16 * tt + 4 * lq;
    if (tt >= Tt || c >= P.N) continue;
    float sc[4], sh[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = c + r < P.N;
      sc[r] = ok ? P.e_scale[c + r] : 1.f;
      sh[r] = ok ? P.e_shift[c + r] : 0.f;
    }
    f32x4 s1 = zero, s2 = zero;
    const float lin_floor = P.linear ? -INFINITY : 0.f;      // ReLU as max(v, floor): floor = -inf keeps the affine value
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pl = 16 * j + li;
      const int py = pl / TW, px = pl - py * TW;
      const int yy = oy0 + py, xx = ox0 + px;
      if (py >= TH || yy >= P.Ho || xx >= P.Wo) continue;
      f32x4 v;
      if (P.stats) {                 // conv.1 BatchNorm statistics pass: sums of the pre-BN value; with `out` it is stored as well (one
        f32x4 u;                     // contraction in training: y = relu(bn(u)) is then an elementwise pass and u is what the backward needs)
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
      T* o = out + (((long)n * P.Ho + yy) * P.Wo + xx) * P.ldo + c;
      if ((P.ldo & 3) == 0 && c + 3 < P.N) {
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

template <typename T, int MT>
__global__ __launch_bounds__(LY_THREADS) void ly_rfcbam3_kernel(const LyRfcbam3Params P, const int gy, const int nct, const int nrt) {
  ly_rfcbam3_body<T, MT, false>(P, gy, nct, nrt);
}
template <typename T, int MT>
__global__ __launch_bounds__(LY_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void ly_rfcbam3_sw_kernel(const LyRfcbam3Params P, const int gy, const int nct, const int nrt) {
  ly_rfcbam3_body<T, MT, true>(P, gy, nct, nrt);
}

template <typename T, int MT, bool SW>
static int launch_rf3_k(const LyRfcbam3Params& P, hipStream_t st) {
  const int nct = (P.Wo + P.TW - 1) / P.TW, nrt = (P.Ho + P.TH - 1) / P.TH;
  const int gy = (P.N + 64 * MT - 1) / (64 * MT);
  const int IH = P.s * (P.TH - 1) + 3, IW = P.s * (P.TW - 1) + 3;
  size_t lds = LyT<T>::PL * (size_t)64 * LY_RSG + sizeof(float) * ((size_t)(SW ? 0 : 4 * LY_RF3_WF) + (size_t)IH * IW * (LY_GCC + 1));
  LY_CHECK(lds <= 160 * 1024, "rfcbam3: tile needs %zu B LDS", lds);
  LY_CHECK(IH * IW * (LY_GCC / 4) <= LY_RF3_NV * LY_THREADS, "rfcbam3: input tile %dx%d exceeds the staging capacity", IH, IW);
  void (*k)(const LyRfcbam3Params, int, int, int);
  if constexpr (SW) k = ly_rfcbam3_sw_kernel<T, MT>;
  else k = ly_rfcbam3_kernel<T, MT>;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    LY_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    configured = true;
  }
  long nb = (long)P.n_img * nrt * nct * gy;
  LY_CHECK(nb < (1L << 31), "rfcbam3: grid too large");
  hipLaunchKernelGGL(k, dim3((unsigned)nb), dim3(LY_THREADS), lds, st, P, gy, nct, nrt);
  LY_LAUNCH_CHECK();
  return 0;
}

template <typename T, int MT>
static int launch_rf3(const LyRfcbam3Params& P, hipStream_t st) {
  const long nb = (long)P.n_img * ((P.Wo + P.TW - 1) / P.TW) * ((P.Ho + P.TH - 1) / P.TH) * ((P.N + 64 * MT - 1) / (64 * MT));
  // Measured: at equal occupancy the LDS path for the generate weights is never slower (128->128 @ 80x80x64: 347 vs 387 us).  The
  // scalar path's one advantage is registers: the MT=4 tile fits two waves per SIMD only with it (245 vs 316), which pays
  // once the grid has more than one block per CU (256->256 @ 40x40x64: 244 vs 292 us; at x32, 224 blocks: 180 vs 156 us).
  if constexpr (MT == 4) {
    if (nb > 256) return launch_rf3_k<T, MT, true>(P, st);
  }
  return launch_rf3_k<T, MT, false>(P, st);
}

// N > 128 runs as ONE 256-channel tile (MT = 4): with the loads prefetched the regenerate phase is what the kernel waits for, and
// the wide tile runs it once per pixel tile instead of twice (256->256 @ 40x40x32: module 191 -> 159 us)
template <typename T>
static int rf3_dispatch(const LyRfcbam3Params& P, hipStream_t st) {
  if (P.N > 128) return launch_rf3<T, 4>(P, st);
  if (P.N > 64) return launch_rf3<T, 2>(P, st);
  return launch_rf3<T, 1>(P, st);
}

extern "C" int ly_rfcbam3_fwd(const LyRfcbam3Params* p, void* stream) {
  LY_CHECK(p, "rfcbam3: null params");
  const LyRfcbam3Params& P = *p;
  LY_CHECK(P.dtype == LY_F32 || P.dtype == LY_BF16, "rfcbam3: unknown dtype %d", P.dtype);
  LY_CHECK(P.x && P.wg && P.ca && P.rfa && P.wp && P.e_scale && P.e_shift && (P.out || P.stats), "rfcbam3: null pointer");
  LY_CHECK((P.C & 15) == 0 && (P.ldx & 3) == 0 && ((uintptr_t)P.x & 15) == 0, "rfcbam3: C=%d must be a multiple of 16", P.C);
  LY_CHECK(P.s >= 1 && P.TH >= 1 && P.TW >= 1 && P.TH * P.TW <= 64, "rfcbam3: bad tile %dx%d", P.TH, P.TW);
  LY_CHECK((long)P.n_img * P.H * P.W * P.ldx < (1L << 31), "rfcbam3: input of %ld elements exceeds the 31-bit offsets of the staging plan", (long)P.n_img * P.H * P.W * P.ldx);
  LY_CHECK(P.Ho == (P.H + 2 - 3) / P.s + 1 && P.Wo == (P.W + 2 - 3) / P.s + 1, "rfcbam3: inconsistent output size");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  return P.dtype == LY_BF16 ? rf3_dispatch<__bf16>(P, st) : rf3_dispatch<float>(P, st);
}
