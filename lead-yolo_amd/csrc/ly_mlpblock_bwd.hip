// Fused MLPBlock backward: C ABI + the C = 16 / 24 / 40 instantiations (kernels: ly_mlpblock_bwd.hpp).
#include "ly_mlpblock_bwd.hpp"

// 1 when ly_mlpblock_bwd is built for (C, dtype): bf16 storage, C in {16, 24, 40, 80}
extern "C" int ly_mlpblock_bwd_ok(int C, int dtype) { return dtype == LY_BF16 && (C == 16 || C == 24 || C == 40 || C == 80); }

// floats of slab workspace pass 2 may need at most for channel count C (256 CUs x 8 resident blocks at most)
template <int C> static long slab_need() {               // pass 2 (<= 8 blocks per CU; C = 80: 147 KB of LDS, one block per CU) | the dx tail (<= 4 per CU)
  const long p2 = (C >= 80 ? 512L : 2048L) * MlpBwdGeom<C>::SLAB, dx = 1024L * 9 * MlpGeom<C>::PT * MlpGeom<C>::PT * 256;
  return p2 > dx ? p2 : dx;
}
extern "C" long ly_mlpblock_bwd_slab_floats(int C) {
  switch (C) {
    case 16: return slab_need<16>();
    case 24: return slab_need<24>();
    case 40: return slab_need<40>();
    case 80: return slab_need<80>();
    default: return 0;
  }
}

extern "C" int ly_mlpblock_bwd(const void* x, const void* dy, void* g, int n_img, int H, int W, int C, const void* wp, const void* w1,
                               const void* w2t, const void* w1t, const float* a, const float* b, const float* alpha, const float* kappa,
                               const float* lambda, double* stats, float* slab, long slab_floats, float* dw1, float* dw2, int pass, int dtype,
                               void* stream) {
  LY_CHECK(dtype == LY_BF16, "mlpblock_bwd: bf16 storage only (dtype %d)", dtype);
  LY_CHECK(pass == 1 || pass == 2, "mlpblock_bwd: pass must be 1 (BatchNorm sums) or 2 (g, weight gradients)");
  LY_CHECK(x && dy && wp && w1 && w2t && a && b && n_img > 0 && H > 0 && W > 0, "mlpblock_bwd: bad arguments");
  LY_CHECK(pass == 1 ? stats != nullptr : (g && w1t && alpha && kappa && lambda && slab && dw1 && dw2), "mlpblock_bwd: pass %d misses a pointer", pass);
  LY_CHECK(((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)g & 15) == 0 && ((uintptr_t)slab & 15) == 0, "mlpblock_bwd: x / dy / g / slab must be 16-byte aligned");
  const long M = (long)n_img * H * W;
  LY_CHECK(M < (1L << 24), "mlpblock_bwd: M=%ld pixels exceeds the 2^24 limit of the fast index path", M);
  LyMlpBwdArgs P;
  P.x = reinterpret_cast<const __bf16*>(x); P.dy = reinterpret_cast<const __bf16*>(dy); P.g = reinterpret_cast<__bf16*>(g);
  P.M = M; P.H = H; P.W = W; P.n_img = n_img; P.ntiles = 0;
  P.wp = reinterpret_cast<const uint4*>(wp); P.w1 = reinterpret_cast<const uint4*>(w1);
  P.w2t = reinterpret_cast<const uint4*>(w2t); P.w1t = reinterpret_cast<const uint4*>(w1t);
  P.a = a; P.b = b; P.alpha = alpha; P.kappa = kappa; P.lambda = lambda; P.stats = stats; P.slab = slab;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  switch (C) {
    case 16: return mlp_bwd_pass<16, 2>(P, pass, slab_floats, dw1, dw2, st);
    case 24: return mlp_bwd_pass<24, 2>(P, pass, slab_floats, dw1, dw2, st);
    case 40: return mlp_bwd_pass<40, 2>(P, pass, slab_floats, dw1, dw2, st);
    case 80: return ly_mlp_bwd_pass_80(P, pass, slab_floats, dw1, dw2, st);
    default:
      ly_set_error("mlpblock_bwd: unsupported channel count C=%d (built for 16/24/40/80)", C);
      return -1;
  }
}

// 1 when ly_mlpblock_bwd_dx is built for (C, map width W, dtype) and its tile fits the 160 KB of LDS (C = 320 never does: 118 KB of tap fragments
// + an 84 KB tile; the caller then takes the 3x3 data-gradient kernel and an elementwise add)
extern "C" int ly_mlpblock_bwd_dx_ok(int C, int W, int dtype) {
  if (dtype != LY_BF16 || W < 1) return 0;
  switch (C) {
    case 16: return mlp_bwd_dx_fits<16>(W);
    case 24: return mlp_bwd_dx_fits<24>(W);
    case 40: return mlp_bwd_dx_fits<40>(W);
    case 80: case 160: return ly_mlp_bwd_dx_fits_wide(C, W);
    default: return 0;
  }
}

// dx = dy + [pconv^T(g[:, :C/4]) | g[:, C/4:]] and (where built: 2-D patches, C/4 <= 32) dwp += the partial conv's weight gradient, one launch
// (+ a combine).  Returns 0: both done; 1: dx done, dwp left to the caller (ly_wgrad on g, x); < 0: error.
extern "C" int ly_mlpblock_bwd_dx(const void* g, const void* dy, const void* x, void* dx, int n_img, int H, int W, int C, const void* wpt, float* slab,
                                  long slab_floats, float* dwp, int lddw, int dw_ts, int dw_cs, int dtype, void* stream) {
  LY_CHECK(dtype == LY_BF16, "mlpblock_bwd_dx: bf16 storage only (dtype %d)", dtype);
  LY_CHECK(g && dy && x && dx && wpt && n_img > 0 && H > 0 && W > 0 && dx != g && dx != dy, "mlpblock_bwd_dx: bad arguments");
  LY_CHECK(((uintptr_t)g & 15) == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)dx & 15) == 0 && ((uintptr_t)slab & 15) == 0,
           "mlpblock_bwd_dx: g / dy / x / dx / slab must be 16-byte aligned");
  const long M = (long)n_img * H * W;
  LY_CHECK(M < (1L << 24), "mlpblock_bwd_dx: M=%ld pixels exceeds the 2^24 limit of the fast index path", M);
  LyMlpDxArgs P;
  P.g = reinterpret_cast<const __bf16*>(g); P.dy = reinterpret_cast<const __bf16*>(dy); P.x = reinterpret_cast<const __bf16*>(x);
  P.dx = reinterpret_cast<__bf16*>(dx); P.M = M; P.H = H; P.W = W; P.n_img = n_img; P.ntiles = 0;
  P.wpt = reinterpret_cast<const uint4*>(wpt); P.slab = slab;
  if (!slab) dwp = nullptr;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  switch (C) {
    case 16: return dispatch_mlp_bwd_dx<16>(P, slab_floats, dwp, lddw, dw_ts, dw_cs, st);
    case 24: return dispatch_mlp_bwd_dx<24>(P, slab_floats, dwp, lddw, dw_ts, dw_cs, st);
    case 40: return dispatch_mlp_bwd_dx<40>(P, slab_floats, dwp, lddw, dw_ts, dw_cs, st);
    case 80: return ly_mlp_bwd_dx_80(P, slab_floats, dwp, lddw, dw_ts, dw_cs, st);
    case 160: return ly_mlp_bwd_dx_160(P, slab_floats, dwp, lddw, dw_ts, dw_cs, st);
    default:
      ly_set_error("mlpblock_bwd_dx: unsupported channel count C=%d (built for 16/24/40/80/160)", C);
      return -1;
  }
}
