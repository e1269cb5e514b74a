// RFCBAMConv kernel_size 1 backward, fused recompute passes (reference models/rfa.py:113-129 with k = 1; autograd of it), gfx950.
//
// For k = 1 `generate` is a per-channel scale followed by BatchNorm and ReLU:  u = gw[c]*x,  G = relu(ag[c]*u + bg[c])  — one multiply-add
// per element.  The first-generation backward (ly_rfcbam_bwd.hip, thread = channel, 2-byte accesses) still materialised it and walked the
// expanded tensors five times: generate (write ug), attn (ug, dcd -> cd, d_rfa, d_ca), relu (ug, dcd -> dv over dcd, BatchNorm sums), gen
// (x, ug, dv -> dug, dwg), dx — 16 passes over [pixels][C] tensors at ~1.4 TB/s.  Here G is recomputed from x wherever it is needed, a lane
// owns 16 bytes of one pixel (8 bf16 / 4 fp32 channels), the lanes of a pixel form a power-of-two group (channel reductions = shuffles), and
// three passes read (x, dcd) and write only what somebody needs:
//   pass A: cd = G*ca*rfa (the conv weight gradient's operand), d_rfa[p] = sum_c dcd*G*ca, gmax[p] = max_c G, d_ca[n][c] += sum_p dcd*G*rfa
//   pass B: dv = (dcd*ca*rfa + d_mean/C + [G == gmax] d_max) [G > 0]  ->  BatchNorm sums  s1[c] += dv, s2[c] += dv*u      (nothing stored)
//   pass C: dv again, du = alpha*dv + kappa + lambda*u,  dgw[c] += sum_p du*x,  dx = du*gw + dgap[n][c]*scale
// A block walks pixels of ONE image; per-channel sums live in registers over the walk, meet in LDS and leave with one atomic per channel.
#include "ly_tile.hpp"
#include "ly_params.h"

#define RF1_A 0
#define RF1_B 1
#define RF1_C 2

template <typename T, int MODE>
__global__ __launch_bounds__(LY_THREADS) void ly_rf1_bwd_kernel(const LyRf1BwdParams P, const int lpp2, const int chunk) {
  constexpr int VW = LyT<T>::VW, NQ = VW / 4;
  using RV = typename LyT<T>::RV;
  __shared__ float red[4096];                              // [slot][quantity][padded C]: 256 lanes x 8 channels x (1 or 2 quantities)
  const int tid = threadIdx.x;
  const int l = tid & (lpp2 - 1), slot = tid / lpp2, PB = LY_THREADS / lpp2;
  const int C = P.C, cv = l * VW;
  const bool lok = cv < C;
  const int cvc = lok ? cv : 0;
  const long n = blockIdx.y;
  const long m_lo = n * P.HW + (long)blockIdx.x * chunk;
  const long m_hi_ = m_lo + chunk, m_img = (n + 1) * P.HW;
  const long m_hi = m_hi_ < m_img ? m_hi_ : m_img;
  const T* const x = reinterpret_cast<const T*>(P.x);
  const T* const dcd = reinterpret_cast<const T*>(P.dcd);
  f32x4 gw[NQ], ag[NQ], bg[NQ], cav[NQ], al[NQ], ka[NQ], la[NQ], dgp[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    gw[q] = ly_ldg4(P.gw + cvc + 4 * q);
    ag[q] = ly_ldg4(P.ag + cvc + 4 * q);
    bg[q] = ly_ldg4(P.bg + cvc + 4 * q);
    cav[q] = ly_ldg4(P.ca + n * C + cvc + 4 * q);
    if constexpr (MODE == RF1_C) {
      al[q] = ly_ldg4(P.alpha + cvc + 4 * q);
      ka[q] = ly_ldg4(P.kappa + cvc + 4 * q);
      la[q] = ly_ldg4(P.lambda + cvc + 4 * q);
      dgp[q] = P.dgap ? ly_ldg4(P.dgap + n * C + cvc + 4 * q) * P.dgap_scale : ly_zero4();
    }
  }
  const float invC = 1.f / (float)C;
  f32x4 acc1[NQ], acc2[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) { acc1[q] = ly_zero4(); acc2[q] = ly_zero4(); }

  // two pixels per trip: all loads of the trip are issued before the first use
  constexpr int UP = 2;
  for (long m0 = m_lo + slot; m0 < m_hi; m0 += UP * PB) {
    RV xr[UP], dr[UP];
    float rf[UP], gm[UP], dm0[UP], dm1[UP];
    bool live[UP];
#pragma unroll
    for (int u = 0; u < UP; ++u) {
      const long mm_ = m0 + (long)u * PB;
      live[u] = mm_ < m_hi;
      const long m = live[u] ? mm_ : m_hi - 1;
      xr[u] = ly_ldrv<T>(x + m * P.ldx + cvc);
      dr[u] = ly_ldrv<T>(dcd + m * C + cvc);
      rf[u] = P.rfa[m];
      if constexpr (MODE != RF1_A) {
        gm[u] = P.gmax[m];
        dm0[u] = P.d_mm[2 * m];
        dm1[u] = P.d_mm[2 * m + 1];
      }
    }
#pragma unroll
    for (int u = 0; u < UP; ++u) {
      const long m = m0 + (long)u * PB;
      const bool on = live[u] && lok;
      f32x4 xv[NQ], dv[NQ];
      ly_rv_unpack(xr[u], xv);
      ly_rv_unpack(dr[u], dv);
      if constexpr (MODE == RF1_A) {
        float srfa = 0.f, mx = 0.f;
        f32x4 cdv[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float uu = gw[q][r] * xv[q][r];
            const float G = fmaxf(__builtin_fmaf(ag[q][r], uu, bg[q][r]), 0.f);
            const float t = on ? dv[q][r] * G : 0.f;
            srfa += t * cav[q][r];
            mx = fmaxf(mx, on ? G : 0.f);
            acc1[q][r] += t * rf[u];
            cdv[q][r] = G * cav[q][r] * rf[u];
          }
        }
        srfa = ly_group_sum(srfa, lpp2);                      // (in-row steps on the vector ALU: ly_common.hpp)
        mx = ly_group_max(mx, lpp2);
        if (on) *reinterpret_cast<RV*>(reinterpret_cast<T*>(P.cd) + m * C + cv) = ly_rv_pack(cdv, (RV*)nullptr);
        if (live[u] && l == 0) { P.d_rfa[m] = srfa; P.gmax_out[m] = mx; }
      } else {
        f32x4 ov[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float uu = gw[q][r] * xv[q][r];
            const float G = fmaxf(__builtin_fmaf(ag[q][r], uu, bg[q][r]), 0.f);
            float dG = dv[q][r] * rf[u] * cav[q][r] + dm1[u] * invC;
            if (G == gm[u]) dG += dm0[u];
            const float d = (on && G > 0.f) ? dG : 0.f;
            if constexpr (MODE == RF1_B) {
              acc1[q][r] += d;
              acc2[q][r] += d * uu;
            } else {
              const float du = on ? al[q][r] * d + ka[q][r] + la[q][r] * uu : 0.f;
              acc1[q][r] += du * xv[q][r];
              ov[q][r] = du * gw[q][r] + dgp[q][r];
            }
          }
        }
        if constexpr (MODE == RF1_C) {
          if (on) *reinterpret_cast<RV*>(reinterpret_cast<T*>(P.dx) + m * P.lddx + cv) = ly_rv_pack(ov, (RV*)nullptr);
        }
      }
    }
  }

  // per-channel sums: slots meet in LDS, one atomic per channel (and quantity) and block
  constexpr int NQT = MODE == RF1_B ? 2 : 1;
  const int CS = lpp2 * VW;                                 // padded channel count of a slot row
  {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        red[(slot * NQT) * CS + cv + 4 * q + r] = acc1[q][r];
        if constexpr (MODE == RF1_B) red[(slot * NQT + 1) * CS + cv + 4 * q + r] = acc2[q][r];
      }
    __syncthreads();
    for (int e = tid; e < NQT * C; e += LY_THREADS) {
      const int qn = e / C, c = e - qn * C;
      float s = 0.f;
      for (int sl = 0; sl < PB; ++sl) s += red[(sl * NQT + qn) * CS + c];
      if constexpr (MODE == RF1_A) atomicAdd(P.d_ca + n * C + c, (double)s);                                  // double accumulators (ly_common.hpp ly_stats_flush)
      else if constexpr (MODE == RF1_B) atomicAdd(P.sums + (size_t)((blockIdx.x + blockIdx.y) & (LY_STATS_STRIPES - 1)) * 2 * C + qn * C + c, (double)s);
      else ly_gacc(P.dgw, c, s, P.dgw_f64);
    }
  }
}

template <typename T>
static int rf1_launch(const LyRf1BwdParams& P, int pass, hipStream_t st) {
  constexpr int VW = LyT<T>::VW;
  const int lpp = P.C / VW;
  int lpp2 = 1;
  while (lpp2 < lpp) lpp2 <<= 1;
  const int PB = LY_THREADS / lpp2;
  // ~1536 blocks; a block stays inside one image (d_ca and the dgap term are per image)
  long per_img = (1536 + P.n_img - 1) / P.n_img;
  const long max_b = (P.HW + 4L * PB - 1) / (4L * PB);
  if (per_img > max_b) per_img = max_b;
  if (per_img < 1) per_img = 1;
  long chunk = (P.HW + per_img - 1) / per_img;
  chunk = (chunk + PB - 1) / PB * PB;
  per_img = (P.HW + chunk - 1) / chunk;
  const dim3 grid((unsigned)per_img, (unsigned)P.n_img);
  if (pass == RF1_A) hipLaunchKernelGGL((ly_rf1_bwd_kernel<T, RF1_A>), grid, dim3(LY_THREADS), 0, st, P, lpp2, (int)chunk);
  else if (pass == RF1_B) hipLaunchKernelGGL((ly_rf1_bwd_kernel<T, RF1_B>), grid, dim3(LY_THREADS), 0, st, P, lpp2, (int)chunk);
  else hipLaunchKernelGGL((ly_rf1_bwd_kernel<T, RF1_C>), grid, dim3(LY_THREADS), 0, st, P, lpp2, (int)chunk);
  LY_LAUNCH_CHECK();
  return 0;
}

extern "C" int ly_rf1_bwd(const LyRf1BwdParams* p, int pass, void* stream) {
  LY_CHECK(p, "rf1_bwd: null params");
  const LyRf1BwdParams& P = *p;
  LY_CHECK_DTYPE(P.dtype, "rf1_bwd");
  const int vw = P.dtype == LY_BF16 ? 8 : 4;
  LY_CHECK(P.n_img > 0 && P.HW > 0 && P.C > 0 && (P.C % vw) == 0 && P.C / vw <= 64, "rf1_bwd: C=%d must be a multiple of %d, at most %d", P.C, vw, 64 * vw);
  LY_CHECK((P.ldx % vw) == 0 && ((uintptr_t)P.x & 15) == 0 && ((uintptr_t)P.dcd & 15) == 0, "rf1_bwd: x / dcd must be 16-byte aligned rows");
  LY_CHECK(P.x && P.dcd && P.gw && P.ag && P.bg && P.ca && P.rfa, "rf1_bwd: null pointer");
  LY_CHECK(pass >= RF1_A && pass <= RF1_C, "rf1_bwd: pass %d", pass);
  if (pass == RF1_A) LY_CHECK(P.cd && P.d_rfa && P.gmax_out && P.d_ca && ((uintptr_t)P.cd & 15) == 0, "rf1_bwd: pass A needs cd, d_rfa, gmax_out, d_ca");
  if (pass != RF1_A) LY_CHECK(P.gmax && P.d_mm, "rf1_bwd: passes B / C need gmax and d_mm");
  if (pass == RF1_B) LY_CHECK(P.sums, "rf1_bwd: pass B needs sums");
  if (pass == RF1_C) LY_CHECK(P.alpha && P.kappa && P.lambda && P.dx && P.dgw && (P.lddx % vw) == 0 && ((uintptr_t)P.dx & 15) == 0, "rf1_bwd: pass C needs alpha / kappa / lambda, dx, dgw");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  return P.dtype == LY_BF16 ? rf1_launch<__bf16>(P, pass, st) : rf1_launch<float>(P, pass, st);
}

// -------------------------------------------------------------------------------------------------
// RFCBAMConv kernel_size 3, first-generation (streamed) backward: the attention pass and the ReLU / routing pass over the expanded tensors
// ug, dcd [pixels][9 taps][C] with the same lane = 16 bytes layout (layer 20: O = 256 keeps the streamed path).  A block owns
// one image; slot = (pixel slot, tap): a lane's tap is fixed, so its BatchNorm constants and per-tap sums stay in registers.
//   pass 0 (replaces ly_rf_bwd_attn): cd = G*ca*rfa, d_rfa[pos], gmax[pos] = max_c G, d_ca[n][c] +=      G = relu(ag[t][c]*ug + bg[t][c])
//   pass 1 (replaces ly_rf_bwd_relu): dv = (dcd*ca*rfa + d_mean/C + [G == gmax] d_max) [G > 0] written over dcd; sums[t][c] += dv, dv*ug
// pos = position of (pixel, tap) in the [n][3Ho][3Wo] attention maps.
// -------------------------------------------------------------------------------------------------
template <typename T, int MODE>
__global__ __launch_bounds__(576) void ly_rf3s_bwd_kernel(const LyRf1BwdParams P, const int lpp2, const int pp, const int chunk, const int Ho,
                                                          const int Wo) {
  constexpr int VW = LyT<T>::VW, NQ = VW / 4;
  using RV = typename LyT<T>::RV;
  __shared__ float red[2 * 576 * 8];
  const int tid = threadIdx.x;
  const int l = tid & (lpp2 - 1), slot = tid / lpp2;
  const int t = slot % 9, ps = slot / 9;
  const int C = P.C, cv = l * VW;
  const bool lok = cv < C;
  const int cvc = lok ? cv : 0;
  const long n = blockIdx.y;
  const long HW = P.HW;                                        // output pixels per image (Ho * Wo)
  const long m_lo = (long)blockIdx.x * chunk;                  // within the image
  const long m_hi = m_lo + chunk < HW ? m_lo + chunk : HW;
  const T* const ug = reinterpret_cast<const T*>(P.x);         // (x slot of the parameter block: ug)
  T* const dcd = reinterpret_cast<T*>(const_cast<void*>(P.dcd));
  f32x4 ag[NQ], bg[NQ], cav[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    ag[q] = ly_ldg4(P.ag + (long)t * C + cvc + 4 * q);
    bg[q] = ly_ldg4(P.bg + (long)t * C + cvc + 4 * q);
    cav[q] = ly_ldg4(P.ca + n * C + cvc + 4 * q);
  }
  const float invC = 1.f / (float)C;
  const int ty = t / 3, tx = t - 3 * ty;
  f32x4 acc1[NQ], acc2[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) { acc1[q] = ly_zero4(); acc2[q] = ly_zero4(); }
  constexpr int UP = 2;
  for (long m0 = m_lo + ps; m0 < m_hi; m0 += (long)UP * pp) {
    RV ur[UP], dr[UP];
    float rf[UP], gm[UP], dm0[UP], dm1[UP];
    long pos[UP], row[UP];
    bool live[UP];
#pragma unroll
    for (int u = 0; u < UP; ++u) {
      const long mm_ = m0 + (long)u * pp;
      live[u] = mm_ < m_hi;
      const long ml = live[u] ? mm_ : m_hi - 1;
      const int ho = (int)(ml / Wo), wo = (int)(ml - (long)ho * Wo);
      row[u] = ((n * HW + ml) * 9 + t) * C + cvc;
      pos[u] = ((n * 3 * Ho + 3 * ho + ty) * (3L * Wo)) + 3 * wo + tx;
      ur[u] = ly_ldrv<T>(ug + row[u]);
      dr[u] = ly_ldrv<T>(dcd + row[u]);
      rf[u] = P.rfa[pos[u]];
      if constexpr (MODE == RF1_B) {
        gm[u] = P.gmax[pos[u]];
        dm0[u] = P.d_mm[2 * pos[u]];
        dm1[u] = P.d_mm[2 * pos[u] + 1];
      }
    }
#pragma unroll
    for (int u = 0; u < UP; ++u) {
      const bool on = live[u] && lok;
      f32x4 uv[NQ], dv[NQ], ov[NQ];
      ly_rv_unpack(ur[u], uv);
      ly_rv_unpack(dr[u], dv);
      if constexpr (MODE == RF1_A) {
        float srfa = 0.f, mx = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float G = fmaxf(__builtin_fmaf(ag[q][r], uv[q][r], bg[q][r]), 0.f);
            const float tt = on ? dv[q][r] * G : 0.f;
            srfa += tt * cav[q][r];
            mx = fmaxf(mx, on ? G : 0.f);
            acc1[q][r] += tt * rf[u];
            ov[q][r] = G * cav[q][r] * rf[u];
          }
        srfa = ly_group_sum(srfa, lpp2);                      // (in-row steps on the vector ALU: ly_common.hpp)
        mx = ly_group_max(mx, lpp2);
        if (on) *reinterpret_cast<RV*>(reinterpret_cast<T*>(P.cd) + row[u]) = ly_rv_pack(ov, (RV*)nullptr);
        if (live[u] && l == 0) { P.d_rfa[pos[u]] = srfa; P.gmax_out[pos[u]] = mx; }
      } else {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float G = fmaxf(__builtin_fmaf(ag[q][r], uv[q][r], bg[q][r]), 0.f);
            float dG = dv[q][r] * rf[u] * cav[q][r] + dm1[u] * invC;
            if (G == gm[u]) dG += dm0[u];
            const float d = (on && G > 0.f) ? dG : 0.f;
            acc1[q][r] += d;
            acc2[q][r] += d * uv[q][r];
            ov[q][r] = d;
          }
        if (on) *reinterpret_cast<RV*>(dcd + row[u]) = ly_rv_pack(ov, (RV*)nullptr);
      }
    }
  }
  // sums: A: d_ca[n][c] over every slot; B: s1 / s2 per (tap, channel) over the pixel slots of the tap
  const int CS = lpp2 * VW;
  constexpr int NQT = MODE == RF1_B ? 2 : 1;
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      red[(slot * NQT) * CS + cv + 4 * q + r] = acc1[q][r];
      if constexpr (MODE == RF1_B) red[(slot * NQT + 1) * CS + cv + 4 * q + r] = acc2[q][r];
    }
  __syncthreads();
  const int nslots = 9 * pp;
  if constexpr (MODE == RF1_A) {
    for (int c = tid; c < C; c += blockDim.x) {
      float s = 0.f;
      for (int sl = 0; sl < nslots; ++sl) s += red[sl * CS + c];
      atomicAdd(P.d_ca + n * C + c, (double)s);
    }
  } else {
    double* const sm = P.sums + (size_t)((blockIdx.x + blockIdx.y) & (LY_STATS_STRIPES - 1)) * 2 * 9 * C;
    for (int e = tid; e < 2 * 9 * C; e += blockDim.x) {
      const int qn = e / (9 * C), rem = e - qn * 9 * C;
      const int tt = rem / C, c = rem - tt * C;
      float s = 0.f;
      for (int k = 0; k < pp; ++k) s += red[((k * 9 + tt) * NQT + qn) * CS + c];
      atomicAdd(sm + qn * 9 * C + tt * C + c, (double)s);
    }
  }
}

// P as for ly_rf1_bwd with: x = ug [pixels][9][C] (dense), dcd [pixels][9][C] (pass 1 overwrites it with dv), ag / bg [9][C], HW = Ho*Wo
// output pixels per image, rfa / gmax / d_rfa [n][3Ho][3Wo], d_mm [n][3Ho][3Wo][2], sums [LY_STATS_STRIPES][2][9][C] (pass 1, zeroed)
extern "C" int ly_rf3s_bwd(const LyRf1BwdParams* p, int Ho, int Wo, int pass, void* stream) {
  LY_CHECK(p, "rf3s_bwd: null params");
  const LyRf1BwdParams& P = *p;
  LY_CHECK_DTYPE(P.dtype, "rf3s_bwd");
  const int vw = P.dtype == LY_BF16 ? 8 : 4;
  LY_CHECK(P.n_img > 0 && Ho > 0 && Wo > 0 && P.HW == (long)Ho * Wo && P.C > 0 && (P.C % vw) == 0 && P.C / vw <= 64, "rf3s_bwd: bad sizes (C=%d)", P.C);
  LY_CHECK(P.x && P.dcd && P.ag && P.bg && P.ca && P.rfa && ((uintptr_t)P.x & 15) == 0 && ((uintptr_t)P.dcd & 15) == 0, "rf3s_bwd: null / misaligned pointer");
  LY_CHECK(pass == RF1_A || pass == RF1_B, "rf3s_bwd: pass %d", pass);
  if (pass == RF1_A) LY_CHECK(P.cd && P.d_rfa && P.gmax_out && P.d_ca && ((uintptr_t)P.cd & 15) == 0, "rf3s_bwd: pass 0 needs cd, d_rfa, gmax_out, d_ca");
  else LY_CHECK(P.gmax && P.d_mm && P.sums, "rf3s_bwd: pass 1 needs gmax, d_mm, sums");
  int lpp2 = 1;
  while (lpp2 < P.C / vw) lpp2 <<= 1;
  int pp = 576 / (9 * lpp2);
  if (pp < 1) pp = 1;
  if (pp > 4) pp = 4;
  const int threads = 9 * lpp2 * pp;
  long per_img = (1024 + P.n_img - 1) / P.n_img;
  const long max_b = (P.HW + 2L * pp - 1) / (2L * pp);
  if (per_img > max_b) per_img = max_b;
  if (per_img < 1) per_img = 1;
  long chunk = (P.HW + per_img - 1) / per_img;
  chunk = (chunk + pp - 1) / pp * pp;
  per_img = (P.HW + chunk - 1) / chunk;
  const dim3 grid((unsigned)per_img, (unsigned)P.n_img);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define RF3S(T_, M_) hipLaunchKernelGGL((ly_rf3s_bwd_kernel<T_, M_>), grid, dim3(threads), 0, st, P, lpp2, pp, (int)chunk, Ho, Wo)
  if (P.dtype == LY_BF16) { if (pass == RF1_A) RF3S(__bf16, RF1_A); else RF3S(__bf16, RF1_B); }
  else { if (pass == RF1_A) RF3S(float, RF1_A); else RF3S(float, RF1_B); }
#undef RF3S
  LY_LAUNCH_CHECK();
  return 0;
}
