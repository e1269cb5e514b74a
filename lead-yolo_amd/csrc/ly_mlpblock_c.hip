// MLPBlock instantiations for C = 320 (see ly_mlpblock.hpp)
#include "ly_mlpblock.hpp"
int ly_mlp_dispatch_320(LY_MLP_ARGS) { return dispatch_nt<320, 4, 1>(x, y, M, n_img, H, W, wp, w1, w2, s, b, stats, dtype, st); }
int ly_mlp_pconv_320(const void* x, void* y, long M, int n_img, int H, int W, const void* wp, int dtype, hipStream_t st) { return dispatch_pconv<320, 4>(x, y, M, n_img, H, W, wp, dtype, st); }
