// ly_gemm_fwd: C ABI + the fp32-storage instantiations of the pointwise-convolution GEMM (kernel: ly_gemm.hpp; the
// bf16-storage instantiations are in ly_gemm_bf16.hip — two translation units only to build them in parallel).
#include "ly_gemm.hpp"

int ly_gemm_dispatch_f32(const LyGemmParams& P, hipStream_t st) { return ly_gemm_dispatch<float>(P, st); }

extern "C" int ly_gemm_fwd(const LyGemmParams* p, void* stream) {
  LY_CHECK(p, "gemm: null params");
  const LyGemmParams& P = *p;
  LY_CHECK(P.dtype == LY_F32 || P.dtype == LY_BF16, "gemm: unknown dtype %d", P.dtype);
  const bool image16 = P.gather == LY_GATHER_PATCH_NCHW_BF16 || P.gather == LY_GATHER_PATCH_NCHW_F16;
  const bool image = P.gather == LY_GATHER_PATCH_NCHW || P.gather == LY_GATHER_PATCH_NCHW_U8 || image16;
  const int vw = (P.dtype == LY_BF16 && !image) ? 8 : 4;   // elements of one source vector
  LY_CHECK(P.a0 && P.wp && (P.out || P.stats), "gemm: null a0/wp/out");
  LY_CHECK(P.M > 0 && P.K > 0 && P.N > 0 && P.H > 0 && P.W > 0, "gemm: bad sizes M=%ld K=%d N=%d", P.M, P.K, P.N);
  LY_CHECK(P.K % vw == 0, "gemm: K=%d must be a multiple of %d", P.K, vw);
  LY_CHECK(((uintptr_t)P.a0 & (P.gather == LY_GATHER_PATCH_NCHW_U8 ? 3 : image16 ? 7 : 15)) == 0 && ((uintptr_t)P.a1 & 15) == 0 && ((uintptr_t)P.res & 15) == 0,
           "gemm: sources must be 16-byte aligned");
  LY_CHECK(((uintptr_t)P.out & (P.dtype == LY_BF16 ? 7 : 15)) == 0 || (P.ldo & 3) != 0, "gemm: out is misaligned for vector stores");
  if (P.gather == LY_GATHER_ROWS || P.gather == LY_GATHER_UP2) {
    LY_CHECK(P.lda0 % vw == 0 && P.k0 % vw == 0 && P.k0 <= P.K && P.k0 > 0, "gemm: lda0=%d k0=%d must be multiples of %d", P.lda0, P.k0, vw);
    LY_CHECK(P.k0 == P.K || (P.a1 && P.lda1 % vw == 0), "gemm: second source missing or misaligned");
    if (P.gather == LY_GATHER_UP2) LY_CHECK((P.H & 1) == 0 && (P.W & 1) == 0, "gemm: upsample source needs even H, W");
  } else if (P.gather == LY_GATHER_PATCH) {
    LY_CHECK(P.ks > 0 && P.pk == P.ks * P.lda0 && P.lda0 % vw == 0 && P.K == P.ks * P.pk, "gemm: patch gather misconfigured");
    LY_CHECK(P.Hin >= P.H * P.ks && P.Win >= P.W * P.ks, "gemm: patch gather input too small");
  } else if (image) {
    LY_CHECK(P.ks == 4 && (P.Win & 3) == 0 && P.K == P.Cin * 16, "gemm: NCHW patch gather needs ks=4, Win%%4==0");
    LY_CHECK(P.Hin >= P.H * P.ks && P.Win >= P.W * P.ks, "gemm: patch gather input too small");
  } else {
    LY_CHECK(false, "gemm: unknown gather mode %d", P.gather);
  }
  if (P.pro == LY_PRO_GATE) LY_CHECK(P.g_h && P.g_w && (!P.res || P.ldres % vw == 0), "gemm: gate prologue needs g_h/g_w (and an aligned residual)");
  if (P.pro == LY_PRO_AFFINE_RELU_CA) LY_CHECK(P.p_scale && P.p_shift && P.p_ca && P.rowscale, "gemm: affine prologue needs scale/shift/ca and rowscale");
  else LY_CHECK(!P.rowscale, "gemm: rowscale is only built together with the affine (RFCBAM k=1) prologue");
  LY_CHECK(P.M < (1L << 24), "gemm: M=%ld pixels exceeds the 2^24 limit of the fast index path", P.M);
  if (P.eadd && !P.scat_ks) {                              // plain rows + eadd: the scatter epilogue with a 1 x 1 "patch" (identity addresses)
    LyGemmParams Q = P;
    Q.scat_ks = 1;
    Q.scat_c = (P.N + 3) & ~3;
    LY_CHECK((P.N & 3) == 0, "gemm: eadd needs N %% 4 == 0 (N=%d)", P.N);
    return ly_gemm_fwd(&Q, stream);
  }
  if (P.eadd) LY_CHECK(P.ldeadd >= P.scat_c && (P.ldeadd & 3) == 0 && ((uintptr_t)P.eadd & (P.dtype == LY_BF16 ? 7 : 15)) == 0, "gemm: eadd must be a 4-element aligned row matrix (ldeadd=%d)", P.ldeadd);
  if (P.scat_ks) {
    // the scatter store lives in the branch-free epilogue only: plain rows in, no prologue, no statistics, vector-friendly widths
    LY_CHECK(P.scat_ks > 0 && P.scat_c > 0 && (P.scat_c & 3) == 0 && P.N == P.scat_ks * P.scat_ks * P.scat_c, "gemm: scatter store needs N = ks*ks*c, c %% 4 == 0 (ks=%d c=%d N=%d)",
             P.scat_ks, P.scat_c, P.N);
    LY_CHECK(P.gather == LY_GATHER_ROWS && P.pro == LY_PRO_NONE && !P.stats && P.out && (P.ldo & 3) == 0 && P.M == (long)(P.M / ((long)P.H * P.W)) * P.H * P.W,
             "gemm: scatter store is built for plain-row sources without prologue / statistics, M a whole number of H x W maps");
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (ly_patch4_try(P, st)) {                              // PatchEmbed on an RGB image: its own LDS-free kernel (ly_patch4.hip)
    LY_LAUNCH_CHECK();
    return 0;
  }
  return P.dtype == LY_BF16 ? ly_gemm_dispatch_bf16(P, st) : ly_gemm_dispatch_f32(P, st);
}
