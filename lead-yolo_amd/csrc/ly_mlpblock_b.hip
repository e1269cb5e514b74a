// MLPBlock instantiations for C = 80 / 160 (see ly_mlpblock.hpp)
#include "ly_mlpblock_res.hpp"
#ifndef LY_RES_D
#define LY_RES_D 4
#endif
int ly_mlp_dispatch_80(LY_MLP_ARGS) {
  // bf16, >= 128 runs of 256 pixels: the resident-weights kernel (ly_mlpblock_res.hpp); anything else, and maps too wide for its halo plan,
  // the one-shot kernels
  if (dtype == LY_BF16 && M >= 128L * 256) {
    const int r = launch_mlp_res<__bf16, 80, 2, 2, LY_RES_D>(reinterpret_cast<const __bf16*>(x), reinterpret_cast<__bf16*>(y), M, H, W, wp, w1, w2, s, b, stats, st);
    if (r != 1) return r;
  }
  return dispatch_nt<80, 2, 4>(x, y, M, n_img, H, W, wp, w1, w2, s, b, stats, dtype, st);
}
int ly_mlp_dispatch_160(LY_MLP_ARGS) { return dispatch_nt<160, 4, 2>(x, y, M, n_img, H, W, wp, w1, w2, s, b, stats, dtype, st); }
int ly_mlp_pconv_80(const void* x, void* y, long M, int n_img, int H, int W, const void* wp, int dtype, hipStream_t st) { return dispatch_pconv<80, 2>(x, y, M, n_img, H, W, wp, dtype, st); }
int ly_mlp_pconv_160(const void* x, void* y, long M, int n_img, int H, int W, const void* wp, int dtype, hipStream_t st) { return dispatch_pconv<160, 4>(x, y, M, n_img, H, W, wp, dtype, st); }
