// RFCBAMConv backward (reference models/rfa.py:113-129 under autograd), gfx950; the expanded tensors, x and dx are T = float /
// __bf16 (bf16 halves the 9x streams), the maps, per-channel vectors and weight gradients fp32.
//
// Forward recap (k = kernel_size, KK = k*k, stride s, pad k/2; m = output pixel (n, ho, wo), t = tap):
//   ug[m][t][c] = sum_u wg[c*KK + t][u] * x_u(m)[c]                 depthwise `generate` conv (x_u = the KK input taps)
//   G = relu(ag*ug + bg)                                            generate.1 BatchNorm (batch stats) + ReLU
//   mm[pos(m,t)] = (max_c G, mean_c G),  rfa = sigmoid(conv3x3(mm)) receptive-field attention on the k*Ho x k*Wo map
//   cd = G * ca[n][c] * rfa[pos(m,t)]                               ca = SE channel attention
//   out = relu(BN(Wc . cd + bias))                                  conv (k x k stride k == GEMM over (t, c)) + BN + ReLU
//
// The backward materialises the KK-times expanded tensors once, in [m][t][c] order (channel contiguous, so that
// "thread = channel" kernels read and write fully coalesced rows):
//   ly_rf_generate     ug
//   ly_rf_bwd_attn     cd (for the conv weight gradient), d_rfa[pos] = sum_c dcd*G*ca, gmax[pos] = max_c G,
//                      d_ca[n][c] = sum_{m,t} dcd*rfa*G
//   ly_rfa_bwd         d_mm and d(get_weight) from d_rfa through sigmoid + 3x3 conv (maps are 2/C of the tensor)
//   ly_rf_bwd_relu     dv = [G>0] * (dcd*rfa*ca + d_mean/C + [G == gmax]*d_max)  (over dcd), BN sums s1 = sum dv, s2 = sum dv*ug
//   ly_rf_bwd_gen      dug = alpha*dv + kappa + lambda*ug (over dv), d(generate weight)[c][t][u] = sum_m dug[t]*x_u
//   ly_rf_bwd_dx       dx[n,hi,wi,c] = sum over the (m, u) that read this input pixel of sum_t dug[m][t][c]*wg[c][t][u]
// The two GEMM-shaped steps in between (dcd = du_out . Wc, dWc = du_out^T . cd) are ly_gemm_fwd / ly_wgrad.
//
// Thread = channel: a block covers CB = roundup64(min(C, 256)) channels x (256 / CB) pixels at a time, so the lanes
// of a wave are 64 consecutive channels of ONE pixel: per-pixel reductions over channels are wave shuffles, and
// per-channel sums over pixels stay in registers until one atomic flush at the end.
#include "ly_tile.hpp"
#include "ly_params.h"

struct RfGeom {
  int n_img, H, W, C, Ho, Wo, s, k, pad;
  int cb, subs;          // channels per block (multiple of 64), pixel sub-groups per block
  long Mo, chunk;        // output pixels, pixels per block
};

static RfGeom rf_geom(int n_img, int H, int W, int C, int k, int s, long pixels, int& gx, int& gy, long max_blocks = 2048) {
  RfGeom g;
  g.n_img = n_img; g.H = H; g.W = W; g.C = C; g.k = k; g.s = s; g.pad = k / 2;
  g.Ho = (H + 2 * g.pad - k) / s + 1;
  g.Wo = (W + 2 * g.pad - k) / s + 1;
  g.Mo = (long)n_img * g.Ho * g.Wo;
  const int cmin = C < 256 ? C : 256;
  g.cb = (cmin + 63) / 64 * 64;
  g.subs = LY_THREADS / g.cb;
  gy = (C + g.cb - 1) / g.cb;
  long blocks = max_blocks / gy;      // kernels that end with a per-thread atomic flush of many accumulators use fewer, longer blocks
  long chunk = (pixels + blocks - 1) / blocks;
  const long min_chunk = 8L * g.subs;
  if (chunk < min_chunk) chunk = min_chunk;
  chunk = (chunk + g.subs - 1) / g.subs * g.subs;
  g.chunk = chunk;
  gx = (int)((pixels + chunk - 1) / chunk);
  return g;
}

__device__ __forceinline__ float rf_wave_sum(float v) { return ly_group_sum(v, 64); }
__device__ __forceinline__ float rf_wave_max(float v) { return ly_group_max(v, 64); }

// Thread = channel means one element per lane and access: 4 bytes in fp32, but only 2 in bf16 — half-width requests, and the in-place
// updates become sub-dword read-modify-writes (ly_rf_bwd_relu: 235 us in fp32, 525 us in bf16).  For bf16 the two lanes of an
// (even, odd) channel pair therefore SHARE accesses: for tap t the lane with (t & 1) == (c & 1) moves the 4-byte pair of both
// channels and the halves are exchanged with a DPP shuffle — same registers per thread, half the requests, all of them dwords.
// off(t) = element offset of tap t for channel 0; every lane of a pair must call these together (no divergence around them).
// Measured (bf16, L17 + L20 per launch): relu 525 -> 185 us, attn 292 -> 249 us with PAIR; the kernels that also hold 81 weights
// or accumulators per thread got SLOWER (dx 263 -> 510, gen 213 -> 258, generate 116 -> 134 us: the exchange code lands between
// their dependent FMA chains), so they keep element accesses (PAIR = false).
__device__ __forceinline__ float rf_swap1(float v) {        // value of lane ^ 1: one DPP move (quad_perm [1,0,3,2]), no LDS crossbar
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
template <typename T, int KK, bool PAIR, class OffFn>
__device__ __forceinline__ void rf_ld_taps(const T* __restrict__ p, int c, OffFn off, float (&v)[KK]) {
  if constexpr (sizeof(T) == 4 || !PAIR) {
#pragma unroll
    for (int t = 0; t < KK; ++t) v[t] = ly_ld1<T>(p + off(t) + c);
  } else {
    // step j: the even lane fetches the channel PAIR of tap 2j, the odd lane that of tap 2j+1 — one full-wave dword load per two
    // taps; each lane keeps its own channel of the pair it fetched and hands the other half to its partner
    const bool odd = c & 1;
    const int ce = c - (int)odd;
#pragma unroll
    for (int j = 0; j < (KK + 1) / 2; ++j) {
      const int t0 = 2 * j, t1 = 2 * j + 1 < KK ? 2 * j + 1 : 2 * j;         // (odd KK: the last odd lane re-reads tap KK-1, unused)
      const long o = odd ? off(t1) : off(t0);
      const bf16x2 w = *reinterpret_cast<const bf16x2*>(p + o + ce);
      const float lo = (float)w[0], hi = (float)w[1];
      const float recv = rf_swap1(odd ? lo : hi);
      v[t0] = odd ? recv : lo;
      if (2 * j + 1 < KK) v[2 * j + 1] = odd ? hi : recv;
    }
  }
}
template <typename T, int KK, bool PAIR, class OffFn>
__device__ __forceinline__ void rf_st_taps(T* __restrict__ p, int c, OffFn off, const float (&v)[KK], bool ok) {
  if constexpr (sizeof(T) == 4 || !PAIR) {
    if (ok) {
#pragma unroll
      for (int t = 0; t < KK; ++t) ly_st1<T>(p + off(t) + c, v[t]);
    }
  } else {
    const bool odd = c & 1;
    const int ce = c - (int)odd;
#pragma unroll
    for (int j = 0; j < (KK + 1) / 2; ++j) {
      const int t0 = 2 * j;
      const bool has1 = 2 * j + 1 < KK;
      const float v1 = has1 ? v[2 * j + 1] : 0.f;
      // even lane stores the pair of tap t0 = (mine[t0], partner[t0]); odd lane the pair of tap t0+1 = (partner[t0+1], mine[t0+1])
      const float recv = rf_swap1(odd ? v[t0] : v1);          // partner receives: from the odd lane its v[t0], from the even lane its v[t0+1]
      const bf16x2 w = {(__bf16)(odd ? recv : v[t0]), (__bf16)(odd ? v1 : recv)};
      const long o = odd ? off(has1 ? 2 * j + 1 : t0) : off(t0);
      if (ok && (!odd || has1)) *reinterpret_cast<bf16x2*>(p + o + ce) = w;
    }
  }
}

// loads the KK input taps of output pixel (n, ho, wo) for channel c (zero padded)
template <typename T, int K>
__device__ __forceinline__ void rf_taps(const T* __restrict__ x, int ldx, const RfGeom& g, int n, int ho, int wo, int c, float (&xt)[K * K]) {
  bool okt[K * K];
  long offt[K * K];
#pragma unroll
  for (int uy = 0; uy < K; ++uy)
#pragma unroll
    for (int ux = 0; ux < K; ++ux) {
      const int hi = ho * g.s + uy - g.pad, wi = wo * g.s + ux - g.pad;
      okt[uy * K + ux] = hi >= 0 && hi < g.H && wi >= 0 && wi < g.W;
      offt[uy * K + ux] = okt[uy * K + ux] ? (((long)n * g.H + hi) * g.W + wi) * ldx : 0;
    }
  rf_ld_taps<T, K * K, false>(x, c, [&](int t) { return offt[t]; }, xt);
#pragma unroll
  for (int t = 0; t < K * K; ++t) xt[t] = okt[t] ? xt[t] : 0.f;
}

__device__ __forceinline__ void rf_pix(const RfGeom& g, long m, int& n, int& ho, int& wo) {
  const long row = m / g.Wo;
  wo = (int)(m - row * g.Wo);
  n = (int)(row / g.Ho);
  ho = (int)(row - (long)n * g.Ho);
}

// index into the k*Ho x k*Wo maps (rfa, mm, d_rfa, gmax) of tap t of output pixel (n, ho, wo)
template <int K>
__device__ __forceinline__ long rf_pos(const RfGeom& g, int n, int ho, int wo, int t) {
  const int ty = t / K, tx = t - ty * K;
  return (((long)n * g.Ho + ho) * K + ty) * ((long)K * g.Wo) + (long)wo * K + tx;
}

#define RF_THREAD_SETUP                                             \
  const int tid = threadIdx.x;                                      \
  const int cl = tid % g.cb, sub = tid / g.cb;                      \
  if (sub >= g.subs) return; /* whole waves beyond subs*cb idle */  \
  const int c_raw = blockIdx.y * g.cb + cl;                         \
  const bool cok = c_raw < g.C;                                     \
  const int c = cok ? c_raw : g.C - 1;                              \
  const long m_begin = (long)blockIdx.x * g.chunk;                  \
  const long m_end_ = m_begin + g.chunk;

// ---- ug ------------------------------------------------------------------------------------------
template <typename T, int K>
__global__ __launch_bounds__(LY_THREADS) void ly_rf_generate_kernel(const RfGeom g, const T* __restrict__ x, int ldx,
                                                                    const float* __restrict__ wg, T* __restrict__ ug) {
  constexpr int KK = K * K;
  RF_THREAD_SETUP
  const long m_end = m_end_ < g.Mo ? m_end_ : g.Mo;
  float w[KK * KK];
#pragma unroll
  for (int i = 0; i < KK * KK; ++i) w[i] = wg[(long)c * KK * KK + i];
  for (long m = m_begin + sub; m < m_end; m += g.subs) {
    int n, ho, wo;
    rf_pix(g, m, n, ho, wo);
    float xt[KK];
    rf_taps<T, K>(x, ldx, g, n, ho, wo, c, xt);
    float av[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t) {
      float a = 0.f;
#pragma unroll
      for (int u = 0; u < KK; ++u) a += w[t * KK + u] * xt[u];
      av[t] = a;
    }
    rf_st_taps<T, KK, false>(ug, c, [&](int t) { return (m * KK + t) * (long)g.C; }, av, cok);
  }
}

// ---- cd, d_rfa, gmax, d_ca -----------------------------------------------------------------------
template <typename T, int K>
__global__ __launch_bounds__(LY_THREADS) void ly_rf_bwd_attn_kernel(const RfGeom g, const T* __restrict__ ug, const T* __restrict__ dcd,
                                                                    const float* __restrict__ ag, const float* __restrict__ bg,
                                                                    const float* __restrict__ ca, const float* __restrict__ rfa,
                                                                    T* __restrict__ cd, float* __restrict__ d_rfa, float* __restrict__ gmax,
                                                                    float* __restrict__ d_ca) {
  constexpr int KK = K * K;
  RF_THREAD_SETUP
  const long m_end = m_end_ < g.Mo ? m_end_ : g.Mo;
  const int lane = tid & 63;
  float a[KK], b[KK];
#pragma unroll
  for (int t = 0; t < KK; ++t) { a[t] = ag[t * g.C + c]; b[t] = bg[t * g.C + c]; }
  int cur_n = -1;
  float dca = 0.f, cav = 0.f;
  for (long m = m_begin + sub; m < m_end; m += g.subs) {
    int n, ho, wo;
    rf_pix(g, m, n, ho, wo);
    if (n != cur_n) {
      if (cur_n >= 0 && cok) atomicAdd(d_ca + (long)cur_n * g.C + c, dca);
      dca = 0.f; cur_n = n;
      cav = ca[(long)n * g.C + c];
    }
    // all KK taps of the pixel are requested before any of them is used (18 independent loads in flight per lane; loaded and
    // consumed tap by tap, every tap was a full memory round trip: the atomics' branch makes the waits conservative)
    long pos[KK];
    float uv[KK], dv_[KK], rv[KK];
    const auto eoff = [&](int t) { return (m * KK + t) * (long)g.C; };
    rf_ld_taps<T, KK, true>(ug, c, eoff, uv);
    rf_ld_taps<T, KK, true>(dcd, c, eoff, dv_);
#pragma unroll
    for (int t = 0; t < KK; ++t) {
      pos[t] = rf_pos<K>(g, n, ho, wo, t);
      rv[t] = rfa[pos[t]];
    }
    float pr[KK], mx[KK], cdv[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t) {
      float G = fmaxf(__fmaf_rn(a[t], uv[t], b[t]), 0.f);
      float d = dv_[t];
      if (!cok) { G = 0.f; d = 0.f; }
      cdv[t] = G * cav * rv[t];
      dca += d * rv[t] * G;
      pr[t] = rf_wave_sum(d * G * cav);
      mx[t] = rf_wave_max(G);
    }
    rf_st_taps<T, KK, true>(cd, c, eoff, cdv, cok);
    if (lane == 0) {
#pragma unroll
      for (int t = 0; t < KK; ++t) {
        atomicAdd(d_rfa + pos[t], pr[t]);
        atomicMax(reinterpret_cast<unsigned int*>(gmax) + pos[t], __float_as_uint(mx[t]));
      }
    }
  }
  if (cur_n >= 0 && cok) atomicAdd(d_ca + (long)cur_n * g.C + c, dca);
}

// ---- get_weight (3x3 conv 2 -> 1 on the map) + sigmoid backward ------------------------------------
// rfa = sigmoid(pre), pre[p] = sum_{ch,dy,dx} w[ch][dy][dx] * mm[p + (dy-1, dx-1)][ch]
#define RFA_TH 16
#define RFA_TW 64
__global__ __launch_bounds__(LY_THREADS) void ly_rfa_bwd_kernel(const float* __restrict__ d_rfa, const float* __restrict__ rfa,
                                                                const float* __restrict__ mm, const float* __restrict__ w18, int n_img, int Hk,
                                                                int Wk, float* __restrict__ d_mm, float* __restrict__ dw18, int tiles_y, int tiles_x, const int f64) {
  // A block walks 16 x 64 tiles of the map (grid-stride over image x tile): d_pre = d_rfa * rfa * (1 - rfa) and mm are staged ONCE per tile
  // with a one-position halo (zeros outside the map), so a position's 9 + 9 neighbours are LDS reads.  (Until late round 3 every thread
  // fetched its 36 neighbour values from global memory, one dependent round trip after the other: 24 us per launch for a 3.7 MB map.)
  // The 18 weight-gradient partials stay in registers over all of a block's tiles and leave with ONE atomic per weight and block.
  __shared__ float dps[(RFA_TH + 2) * (RFA_TW + 2)];
  __shared__ float mms[(RFA_TH + 2) * (RFA_TW + 2) * 2];
  __shared__ float red[18][4];
  constexpr int HW_ = RFA_TW + 2;
  float w[18], dwl[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) { w[i] = w18[i]; dwl[i] = 0.f; }
  const int tid = threadIdx.x;
  const int tx = tid & (RFA_TW - 1), ty = tid >> 6;                  // 4 row groups x 64 columns
  const long ntile = (long)n_img * tiles_y * tiles_x;
  for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int txi = (int)(tile % tiles_x);
    const long r1 = tile / tiles_x;
    const int tyi = (int)(r1 % tiles_y);
    const long n = r1 / tiles_y;
    const int y0 = tyi * RFA_TH, x0 = txi * RFA_TW;
    __syncthreads();                                                // the previous tile's readers are done
    for (int i = tid; i < (RFA_TH + 2) * HW_; i += LY_THREADS) {
      const int r = i / HW_, cq = i - r * HW_;
      const int y = y0 + r - 1, x = x0 + cq - 1;
      const bool in = y >= 0 && y < Hk && x >= 0 && x < Wk;
      const long p = in ? (n * Hk + y) * Wk + x : 0;
      const float rp = rfa[p], dr = d_rfa[p];
      const float2 mv = *reinterpret_cast<const float2*>(mm + 2 * p);
      dps[i] = in ? dr * rp * (1.f - rp) : 0.f;
      mms[2 * i] = in ? mv.x : 0.f;
      mms[2 * i + 1] = in ? mv.y : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RFA_TH / 4; ++k) {
      const int ly = ty + 4 * k;
      const int yq = y0 + ly, xq = x0 + tx;
      if (yq < Hk && xq < Wk) {
        const int ctr = (ly + 1) * HW_ + tx + 1;
        const float dpre_q = dps[ctr];
        float g0 = 0.f, g1 = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            // d_mm[q] gathers d_pre at p = q - (dy-1, dx-1);  dw gathers mm at q + (dy-1, dx-1)
            const float dp = dps[ctr - (dy - 1) * HW_ - (dx - 1)];
            g0 += w[dy * 3 + dx] * dp;
            g1 += w[9 + dy * 3 + dx] * dp;
            const int pm = ctr + (dy - 1) * HW_ + (dx - 1);
            dwl[dy * 3 + dx] += dpre_q * mms[2 * pm];
            dwl[9 + dy * 3 + dx] += dpre_q * mms[2 * pm + 1];
          }
        *reinterpret_cast<float2*>(d_mm + 2 * ((n * Hk + yq) * Wk + xq)) = make_float2(g0, g1);
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 18; ++i) {
    const float sv = rf_wave_sum(dwl[i]);
    if (lane == 0) red[i][wave] = sv;
  }
  __syncthreads();
  if (threadIdx.x < 18) ly_gacc(dw18, threadIdx.x, red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3], f64);
}

// ---- dv (over dcd) + BN sums ---------------------------------------------------------------------
template <typename T, int K>
__global__ __launch_bounds__(LY_THREADS) void ly_rf_bwd_relu_kernel(const RfGeom g, const T* __restrict__ ug, T* dcd,
                                                                    const float* __restrict__ ag, const float* __restrict__ bg,
                                                                    const float* __restrict__ ca, const float* __restrict__ rfa,
                                                                    const float* __restrict__ gmax, const float* __restrict__ d_mm,
                                                                    float* __restrict__ sums) {
  constexpr int KK = K * K;
  RF_THREAD_SETUP
  const long m_end = m_end_ < g.Mo ? m_end_ : g.Mo;
  float a[KK], b[KK], s1[KK], s2[KK];
#pragma unroll
  for (int t = 0; t < KK; ++t) { a[t] = ag[t * g.C + c]; b[t] = bg[t * g.C + c]; s1[t] = 0.f; s2[t] = 0.f; }
  const float invC = 1.f / (float)g.C;
    for (long m = m_begin + sub; m < m_end; m += g.subs) {       // (lanes beyond C run along with clamped addresses: the pair shuffles need both lanes)
      int n, ho, wo;
      rf_pix(g, m, n, ho, wo);
      const float cav = ca[(long)n * g.C + c];
      // loads of all KK taps first, stores last: dcd is updated in place, so a load placed after a store cannot be hoisted
      // above it by the compiler and every tap would wait for its own round trip
      long pos[KK];
      float u[KK], dc[KK], rv[KK], gm[KK], dm0[KK], dm1[KK];
      const auto eoff = [&](int t) { return (m * KK + t) * (long)g.C; };
      rf_ld_taps<T, KK, true>(ug, c, eoff, u);
      rf_ld_taps<T, KK, true>(dcd, c, eoff, dc);
#pragma unroll
      for (int t = 0; t < KK; ++t) {
        pos[t] = rf_pos<K>(g, n, ho, wo, t);
        rv[t] = rfa[pos[t]];
        gm[t] = gmax[pos[t]];
        dm0[t] = d_mm[2 * pos[t]];
        dm1[t] = d_mm[2 * pos[t] + 1];
      }
      float dvv[KK];
#pragma unroll
      for (int t = 0; t < KK; ++t) {
        const float G = fmaxf(__fmaf_rn(a[t], u[t], b[t]), 0.f);
        float dG = dc[t] * rv[t] * cav + dm1[t] * invC;
        if (G == gm[t]) dG += dm0[t];
        dvv[t] = G > 0.f ? dG : 0.f;
        s1[t] += dvv[t];                      // BatchNorm sums from the fp32 value (before it is rounded to T)
        s2[t] += dvv[t] * u[t];
      }
      rf_st_taps<T, KK, true>(dcd, c, eoff, dvv, cok);
    }
  if (cok) {
    const int CK = g.C * KK;
#pragma unroll
    for (int t = 0; t < KK; ++t) {
      atomicAdd(sums + t * g.C + c, s1[t]);
      atomicAdd(sums + CK + t * g.C + c, s2[t]);
    }
  }
}

// ---- dug (over dv) + d(generate weight) ------------------------------------------------------------
template <typename T, int K>
__global__ __launch_bounds__(LY_THREADS) void ly_rf_bwd_gen_kernel(const RfGeom g, const T* __restrict__ x, int ldx, const T* __restrict__ ug,
                                                                   T* dv, const float* __restrict__ alpha, const float* __restrict__ kappa,
                                                                   const float* __restrict__ lambda, float* __restrict__ dwg) {
  constexpr int KK = K * K;
  RF_THREAD_SETUP
  const long m_end = m_end_ < g.Mo ? m_end_ : g.Mo;
  float al[KK], ka[KK], la[KK], acc[KK * KK];
#pragma unroll
  for (int t = 0; t < KK; ++t) { al[t] = alpha[t * g.C + c]; ka[t] = kappa[t * g.C + c]; la[t] = lambda[t * g.C + c]; }
#pragma unroll
  for (int i = 0; i < KK * KK; ++i) acc[i] = 0.f;
    for (long m = m_begin + sub; m < m_end; m += g.subs) {
      int n, ho, wo;
      rf_pix(g, m, n, ho, wo);
      float xt[KK];
      rf_taps<T, K>(x, ldx, g, n, ho, wo, c, xt);
      float dvl[KK], ugl[KK];              // loads first, in-place stores last (see ly_rf_bwd_relu_kernel)
      const auto eoff = [&](int t) { return (m * KK + t) * (long)g.C; };
      rf_ld_taps<T, KK, false>(dv, c, eoff, dvl);
      rf_ld_taps<T, KK, false>(ug, c, eoff, ugl);
#pragma unroll
      for (int t = 0; t < KK; ++t) {
        const float d = al[t] * dvl[t] + ka[t] + la[t] * ugl[t];
        dvl[t] = d;
#pragma unroll
        for (int u = 0; u < KK; ++u) acc[t * KK + u] += d * xt[u];
      }
      rf_st_taps<T, KK, false>(dv, c, eoff, dvl, cok);
    }
  if (cok) {
    // every (block, pixel sub-group) owns one row of the partial-sum matrix: plain stores, no atomics (81 accumulators per
    // thread x hundreds of blocks was 5-20 M float atomics, 0.65-1.2 ms); the caller sums the rows
    float* d = dwg + ((size_t)blockIdx.x * g.subs + sub) * g.C * KK * KK + (long)c * KK * KK;
#pragma unroll
    for (int i = 0; i < KK * KK; ++i) d[i] = acc[i];
  }
}

// ---- dx ----------------------------------------------------------------------------------------
// stride 2, k = 3, pad 1 (both k=3 layers of LEAD-YOLO): which taps u = (uy, ux) read input pixel (hi, wi) depends only on the
// parities PY = (hi+1)&1, PX = (wi+1)&1 -- uy in {0, 2} when hi+1 is even, uy = 1 otherwise -- so each parity class has a
// compile-time tap list (4 / 2 / 2 / 1 output pixels).  All their 9-tap rows are requested before the first is used; edge
// positions read row 0 and are masked.  (Tap by tap behind `continue`s, every row was its own memory round trip.)
template <typename T, int PY, int PX>
__device__ __forceinline__ float rf_dx_s2(const RfGeom& g, const T* __restrict__ dug, const float (&w)[81], int n, int hi, int wi, int c) {
  constexpr int NY = PY ? 1 : 2, NX = PX ? 1 : 2;
  float v[NY][NX][9];
  bool ok[NY][NX];
#pragma unroll
  for (int jy = 0; jy < NY; ++jy) {
    const int uy = PY ? 1 : 2 * jy;
    const int hh = hi + 1 - uy, ho = hh >> 1;
    const bool oky = hh >= 0 && ho < g.Ho;
#pragma unroll
    for (int jx = 0; jx < NX; ++jx) {
      const int ux = PX ? 1 : 2 * jx;
      const int ww = wi + 1 - ux, wo = ww >> 1;
      ok[jy][jx] = oky && ww >= 0 && wo < g.Wo;
      const long m = ok[jy][jx] ? ((long)n * g.Ho + ho) * g.Wo + wo : 0;
      rf_ld_taps<T, 9, false>(dug, c, [&](int t) { return (m * 9 + t) * (long)g.C; }, v[jy][jx]);
    }
  }
  float acc = 0.f;
#pragma unroll
  for (int jy = 0; jy < NY; ++jy)
#pragma unroll
    for (int jx = 0; jx < NX; ++jx) {
      const int u = (PY ? 1 : 2 * jy) * 3 + (PX ? 1 : 2 * jx);
      float a = 0.f;
#pragma unroll
      for (int t = 0; t < 9; ++t) a += v[jy][jx][t] * w[t * 9 + u];
      acc += ok[jy][jx] ? a : 0.f;
    }
  return acc;
}

// (three waves per SIMD: the k = 3 body sits at 168 registers without the addend and two more would cost a third of the occupancy)
template <typename T, int K, int ADD>
__global__ __launch_bounds__(LY_THREADS) __attribute__((amdgpu_waves_per_eu(3))) void ly_rf_bwd_dx_kernel(const RfGeom g, const T* __restrict__ dug, const float* __restrict__ wg,
                                                                  T* __restrict__ dx, int lddx, const float* __restrict__ addnc, float add_scale) {
  constexpr int KK = K * K;
  RF_THREAD_SETUP
  const long Mi = (long)g.n_img * g.H * g.W;
  const long m_end = m_end_ < Mi ? m_end_ : Mi;
  float w[KK * KK];
#pragma unroll
  for (int i = 0; i < KK * KK; ++i) w[i] = wg[(long)c * KK * KK + i];
  // ADD == 2: the block's pixel chunk touches at most two images (host-checked): their addends are loaded once, before the loop
  const int n_first = (int)(m_begin / ((long)g.H * g.W));
  float add0 = 0.f, add1 = 0.f;
  if constexpr (ADD == 2) {
    add0 = addnc[(long)n_first * g.C + c] * add_scale;
    add1 = addnc[(long)(n_first + 1 < g.n_img ? n_first + 1 : n_first) * g.C + c] * add_scale;
  }
  for (long p = m_begin + sub; p < m_end; p += g.subs) {        // (lanes beyond C run along: the pair shuffles need both lanes)
    const long row = p / g.W;
    const int wi = (int)(p - row * g.W);
    const int n = (int)(row / g.H);
    const int hi = (int)(row - (long)n * g.H);
    // the per-(image, channel) addend is requested first, unconditionally (a load under a run-time `if` costs every later wait its count)
    float addv = 0.f;
    if constexpr (ADD == 1) addv = addnc[(long)n * g.C + c] * add_scale;
    if constexpr (ADD == 2) addv = n == n_first ? add0 : add1;
    float acc = 0.f;
    if constexpr (K == 3) {
      if (g.s == 2) {
        const int py = (hi + 1) & 1, px = (wi + 1) & 1;      // uniform over the wave (all lanes share the pixel)
        if (py == 0 && px == 0) acc = rf_dx_s2<T, 0, 0>(g, dug, w, n, hi, wi, c);
        else if (py == 0) acc = rf_dx_s2<T, 0, 1>(g, dug, w, n, hi, wi, c);
        else if (px == 0) acc = rf_dx_s2<T, 1, 0>(g, dug, w, n, hi, wi, c);
        else acc = rf_dx_s2<T, 1, 1>(g, dug, w, n, hi, wi, c);
        if (cok) ly_st1<T>(dx + p * lddx + c, acc + addv);
        continue;
      }
    }
#pragma unroll
    for (int uy = 0; uy < K; ++uy) {
      const int hh = hi + g.pad - uy;
      if (hh < 0 || hh % g.s != 0) continue;
      const int ho = hh / g.s;
      if (ho >= g.Ho) continue;
#pragma unroll
      for (int ux = 0; ux < K; ++ux) {
        const int ww = wi + g.pad - ux;
        if (ww < 0 || ww % g.s != 0) continue;
        const int wo = ww / g.s;
        if (wo >= g.Wo) continue;
        const long m = ((long)n * g.Ho + ho) * g.Wo + wo;
#pragma unroll
        for (int t = 0; t < KK; ++t) acc += ly_ld1<T>(dug + (m * KK + t) * g.C + c) * w[t * KK + (uy * K + ux)];
      }
    }
    if (cok) ly_st1<T>(dx + p * lddx + c, acc + addv);
  }
}

template <int V>
struct RfIc { static constexpr int value = V; };
#ifndef LY_RF_DX_TILED
#define LY_RF_DX_TILED 1
#endif
#ifndef LY_RF_DX_TM
#define LY_RF_DX_TM 4
#endif

// ---- dx for k = 3, stride 2 through an LDS tile -----------------------------------------------------
// dx[p] = sum over the (<= 4) output pixels m with p = 2m + u - 1 of sum_t dug[m][t][c] * w[c][t][u]: every dug element is needed by the nine input
// pixels around its m, and the per-pixel gather above re-reads it nine times through L1 / L2 (279 us for a 236 MB tensor).  Here a block stages
// the dug rows of a TM x TM tile of output pixels (+ one halo row / column: odd input pixels also take from m + 1) x 64 channels ONCE
// (16-byte loads), then lane = channel walks the 2TM x 2TM input pixels by parity class from LDS.  Rows of m outside the map are staged as
// zeros, so the arithmetic needs no masks.
template <typename T, int TM, int ADD>
__global__ __launch_bounds__(LY_THREADS) void ly_rf_bwd_dx_s2t_kernel(const RfGeom g, const T* __restrict__ dug, const float* __restrict__ wg, T* __restrict__ dx,
                                                                      int lddx, const float* __restrict__ addnc, float add_scale, int tiles_y, int tiles_x) {
  constexpr int TE = TM + 1, ROWS = TE * TE * 9;
  constexpr int VPR = 64 * (int)sizeof(T) / 16;           // 16-byte vectors per 64-channel row
  constexpr int EPV = 16 / (int)sizeof(T);                // elements per vector
  extern __shared__ f32x4 ly_smem4[];
  T* const tile = reinterpret_cast<T*>(ly_smem4);          // [TE*TE][9][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // neighbouring tiles (overlapping input patches) of one channel group on one XCD's L2
  const int lid = ly_xcd_remap((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
  int b = lid % (int)gridDim.x;
  const int tx = b % tiles_x; b /= tiles_x;
  const int ty = b % tiles_y;
  const int n = b / tiles_y;
  const int c0 = (lid / (int)gridDim.x) * 64;
  const int my0 = ty * TM, mx0 = tx * TM;
  // ---- stage ------------------------------------------------------------------------------------
  constexpr int NV = (ROWS * VPR + LY_THREADS - 1) / LY_THREADS;
  ly_u32x4 sv[NV];
#pragma unroll
  for (int e = 0; e < NV; ++e) {
    const int idx = tid + e * LY_THREADS;
    const int row = idx / VPR, v = idx - row * VPR;
    const int ml = row / 9, t = row - ml * 9;
    const int my = my0 + ml / TE, mx = mx0 + ml % TE;
    const bool ok = idx < ROWS * VPR && my < g.Ho && mx < g.Wo && c0 + v * EPV < g.C;
    const long m = ok ? ((long)n * g.Ho + my) * g.Wo + mx : 0;
    sv[e] = *reinterpret_cast<const ly_u32x4*>(dug + (m * 9 + (ok ? t : 0)) * (long)g.C + (ok ? c0 + v * EPV : 0));
    if (!ok) sv[e] = (ly_u32x4){0u, 0u, 0u, 0u};
  }
#pragma unroll
  for (int e = 0; e < NV; ++e) {
    const int idx = tid + e * LY_THREADS;
    if (idx < ROWS * VPR) reinterpret_cast<ly_u32x4*>(tile)[idx] = sv[e];
  }
  const int c = c0 + lane;
  const bool cok = c < g.C;
  float w[81];
#pragma unroll
  for (int i = 0; i < 81; ++i) w[i] = wg[(long)(cok ? c : 0) * 81 + i];
  float addv = 0.f;
  if constexpr (ADD != 0) addv = addnc[(long)n * g.C + (cok ? c : 0)] * add_scale;
  __syncthreads();
  // ---- input pixels by parity class: PY = 1 -> even row (one m row, uy = 1), PY = 0 -> odd row (m rows a+1 with uy = 0 and a with uy = 2) ----
  auto klass = [&](auto pyC, auto pxC) {
    constexpr int PY = decltype(pyC)::value, PX = decltype(pxC)::value;
    constexpr int NY = PY ? 1 : 2, NX = PX ? 1 : 2;
    for (int i = wave; i < TM * TM; i += 4) {
      const int a = i / TM, bb = i - a * TM;
      const int hi = 2 * (my0 + a) + (PY ? 0 : 1), wi = 2 * (mx0 + bb) + (PX ? 0 : 1);
      float acc = 0.f;
#pragma unroll
      for (int jy = 0; jy < NY; ++jy) {
        const int uy = PY ? 1 : 2 * jy;
        const int ly_ = PY ? a : (jy == 0 ? a + 1 : a);
#pragma unroll
        for (int jx = 0; jx < NX; ++jx) {
          const int ux = PX ? 1 : 2 * jx;
          const int lx_ = PX ? bb : (jx == 0 ? bb + 1 : bb);
          const T* r = tile + ((ly_ * TE + lx_) * 9) * 64 + lane;
#pragma unroll
          for (int t = 0; t < 9; ++t) acc += (float)r[t * 64] * w[t * 9 + uy * 3 + ux];
        }
      }
      if (cok && hi < g.H && wi < g.W) ly_st1<T>(dx + (((long)n * g.H + hi) * g.W + wi) * lddx + c, acc + addv);
    }
  };
  klass(RfIc<1>(), RfIc<1>());
  klass(RfIc<1>(), RfIc<0>());
  klass(RfIc<0>(), RfIc<1>());
  klass(RfIc<0>(), RfIc<0>());
}

// ---- C entry points ------------------------------------------------------------------------------
#define RF_ARGS_OK(k, C) LY_CHECK(((k) == 1 || (k) == 3) && (C) > 0 && ((C) & 1) == 0, "rfcbam backward: kernel_size must be 1 or 3 and C even")
#define RF_LAUNCH(kern, grid, ...)                                                                          \
  LY_WITH_T(dtype, {                                                                                        \
    if (k == 3) hipLaunchKernelGGL((kern<T, 3>), grid, dim3(LY_THREADS), 0, st, __VA_ARGS__);                 \
    else hipLaunchKernelGGL((kern<T, 1>), grid, dim3(LY_THREADS), 0, st, __VA_ARGS__);                        \
  })
#define RF_T(p) reinterpret_cast<T*>(p)
#define RF_CT(p) reinterpret_cast<const T*>(p)

extern "C" int ly_rf_generate(const void* x, int ldx, int n_img, int H, int W, int C, int k, int s, const float* wg, void* ug, int dtype, void* stream) {
  RF_ARGS_OK(k, C);
  LY_CHECK_DTYPE(dtype, "rf_generate");
  LY_CHECK(x && wg && ug, "rf_generate: null pointer");
  int gx, gy;
  const long Mo = (long)n_img * ((H + 2 * (k / 2) - k) / s + 1) * ((W + 2 * (k / 2) - k) / s + 1);
  const RfGeom g = rf_geom(n_img, H, W, C, k, s, Mo, gx, gy);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  RF_LAUNCH(ly_rf_generate_kernel, dim3(gx, gy), g, RF_CT(x), ldx, wg, RF_T(ug));
  LY_LAUNCH_CHECK();
  return 0;
}

extern "C" int ly_rf_bwd_attn(int n_img, int H, int W, int C, int k, int s, const void* ug, const void* dcd, const float* ag, const float* bg,
                              const float* ca, const float* rfa, void* cd, float* d_rfa, float* gmax, float* d_ca, int dtype, void* stream) {
  RF_ARGS_OK(k, C);
  LY_CHECK_DTYPE(dtype, "rf_bwd_attn");
  LY_CHECK(ug && dcd && ag && bg && ca && rfa && cd && d_rfa && gmax && d_ca, "rf_bwd_attn: null pointer");
  int gx, gy;
  const long Mo = (long)n_img * ((H + 2 * (k / 2) - k) / s + 1) * ((W + 2 * (k / 2) - k) / s + 1);
  const RfGeom g = rf_geom(n_img, H, W, C, k, s, Mo, gx, gy);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  RF_LAUNCH(ly_rf_bwd_attn_kernel, dim3(gx, gy), g, RF_CT(ug), RF_CT(dcd), ag, bg, ca, rfa, RF_T(cd), d_rfa, gmax, d_ca);
  LY_LAUNCH_CHECK();
  return 0;
}

extern "C" int ly_rfa_bwd(const float* d_rfa, const float* rfa, const float* mm, const float* w18, int n_img, int Hk, int Wk, float* d_mm,
                          float* dw18, int dw18_f64, void* stream) {
  LY_CHECK(d_rfa && rfa && mm && w18 && d_mm && dw18 && n_img > 0 && Hk > 0 && Wk > 0, "rfa_bwd: bad arguments");
  const int tiles_y = (Hk + RFA_TH - 1) / RFA_TH, tiles_x = (Wk + RFA_TW - 1) / RFA_TW;
  long blocks = (long)n_img * tiles_y * tiles_x;
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(ly_rfa_bwd_kernel, dim3((unsigned)blocks), dim3(LY_THREADS), 0,
                     reinterpret_cast<hipStream_t>(stream), d_rfa, rfa, mm, w18, n_img, Hk, Wk, d_mm, dw18, tiles_y, tiles_x, dw18_f64);
  LY_LAUNCH_CHECK();
  return 0;
}

extern "C" int ly_rf_bwd_relu(int n_img, int H, int W, int C, int k, int s, const void* ug, void* dcd, const float* ag, const float* bg,
                              const float* ca, const float* rfa, const float* gmax, const float* d_mm, float* sums, int dtype, void* stream) {
  RF_ARGS_OK(k, C);
  LY_CHECK_DTYPE(dtype, "rf_bwd_relu");
  LY_CHECK(ug && dcd && ag && bg && ca && rfa && gmax && d_mm && sums, "rf_bwd_relu: null pointer");
  int gx, gy;
  const long Mo = (long)n_img * ((H + 2 * (k / 2) - k) / s + 1) * ((W + 2 * (k / 2) - k) / s + 1);
  const RfGeom g = rf_geom(n_img, H, W, C, k, s, Mo, gx, gy, 512);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  RF_LAUNCH(ly_rf_bwd_relu_kernel, dim3(gx, gy), g, RF_CT(ug), RF_T(dcd), ag, bg, ca, rfa, gmax, d_mm, sums);
  LY_LAUNCH_CHECK();
  return 0;
}

extern "C" int ly_rf_bwd_gen(const void* x, int ldx, int n_img, int H, int W, int C, int k, int s, const void* ug, void* dv, const float* alpha,
                             const float* kappa, const float* lambda, float* dwg, int part_rows, int dtype, void* stream) {
  RF_ARGS_OK(k, C);
  LY_CHECK_DTYPE(dtype, "rf_bwd_gen");
  LY_CHECK(x && ug && dv && alpha && kappa && lambda && dwg && part_rows >= 4, "rf_bwd_gen: bad arguments");
  int gx, gy;
  const long Mo = (long)n_img * ((H + 2 * (k / 2) - k) / s + 1) * ((W + 2 * (k / 2) - k) / s + 1);
  const int cmin = C < 256 ? C : 256;
  const int subs = LY_THREADS / ((cmin + 63) / 64 * 64);
  const int groups = (C + 255) / 256;
  const RfGeom g = rf_geom(n_img, H, W, C, k, s, Mo, gx, gy, (long)(part_rows / subs) * groups);
  LY_CHECK((long)gx * g.subs <= part_rows, "rf_bwd_gen: %d partial rows are not enough for %d blocks x %d", part_rows, gx, g.subs);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  RF_LAUNCH(ly_rf_bwd_gen_kernel, dim3(gx, gy), g, RF_CT(x), ldx, RF_CT(ug), RF_T(dv), alpha, kappa, lambda, dwg);
  LY_LAUNCH_CHECK();
  return 0;
}

extern "C" int ly_rf_bwd_dx(int n_img, int H, int W, int C, int k, int s, const void* dug, const float* wg, void* dx, int lddx, const float* addnc,
                            float add_scale, int dtype, void* stream) {
  RF_ARGS_OK(k, C);
  LY_CHECK_DTYPE(dtype, "rf_bwd_dx");
  LY_CHECK(dug && wg && dx, "rf_bwd_dx: null pointer");
  int gx, gy;
  const RfGeom g = rf_geom(n_img, H, W, C, k, s, (long)n_img * H * W, gx, gy);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (k == 3 && s == 2 && (H & 1) == 0 && (W & 1) == 0 && (C & 7) == 0 && LY_RF_DX_TILED) {
    // LDS-tiled gather (dug read ~1.6x from HBM instead of nine times through L2): 4 x 4 output pixels + halo per block (29 KB bf16 / 58 KB
    // fp32: occupancy decides — 8 x 8 = 93 KB, one block per CU: 248 us; 5 x 5: 120; 4 x 4: 110; 3 x 3: 167; the per-pixel gather: 279)
    const int Ho = H / 2, Wo = W / 2;
    LY_WITH_T(dtype, {
      constexpr int TM = sizeof(T) == 2 ? LY_RF_DX_TM : 4;
      const int tiles_y = (Ho + TM - 1) / TM, tiles_x = (Wo + TM - 1) / TM;
      const size_t lds = (size_t)(TM + 1) * (TM + 1) * 9 * 64 * sizeof(T);
      const dim3 grid((unsigned)(n_img * tiles_y * tiles_x), (unsigned)((C + 63) / 64));
      if (addnc) {
        static bool a1 = false;
        if (!a1) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ly_rf_bwd_dx_s2t_kernel<T, TM, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); a1 = true; }
        hipLaunchKernelGGL((ly_rf_bwd_dx_s2t_kernel<T, TM, 1>), grid, dim3(LY_THREADS), lds, st, g, RF_CT(dug), wg, RF_T(dx), lddx, addnc, add_scale, tiles_y, tiles_x);
      } else {
        static bool a0 = false;
        if (!a0) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ly_rf_bwd_dx_s2t_kernel<T, TM, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); a0 = true; }
        hipLaunchKernelGGL((ly_rf_bwd_dx_s2t_kernel<T, TM, 0>), grid, dim3(LY_THREADS), lds, st, g, RF_CT(dug), wg, RF_T(dx), lddx, addnc, add_scale, tiles_y, tiles_x);
      }
    });
    LY_LAUNCH_CHECK();
    return 0;
  }
  const int add = !addnc ? 0 : (g.chunk <= (long)H * W ? 2 : 1);
#define RF_DX(KV, AV) hipLaunchKernelGGL((ly_rf_bwd_dx_kernel<T, KV, AV>), dim3(gx, gy), dim3(LY_THREADS), 0, st, g, RF_CT(dug), wg, RF_T(dx), lddx, addnc, add_scale)
  LY_WITH_T(dtype, {
    if (k == 3) { if (add == 2) RF_DX(3, 2); else if (add == 1) RF_DX(3, 1); else RF_DX(3, 0); }
    else { if (add == 2) RF_DX(1, 2); else if (add == 1) RF_DX(1, 1); else RF_DX(1, 0); }
  });
#undef RF_DX
  LY_LAUNCH_CHECK();
  return 0;
}
