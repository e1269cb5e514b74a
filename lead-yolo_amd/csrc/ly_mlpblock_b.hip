// MLPBlock instantiations for C = 80 / 160 (see ly_mlpblock.hpp)
#include "ly_mlpblock.hpp"
int ly_mlp_dispatch_80(LY_MLP_ARGS) { return dispatch_nt<80, 2, 4>(x, y, M, n_img, H, W, wp, w1, w2, s, b, stats, dtype, st); }
int ly_mlp_dispatch_160(LY_MLP_ARGS) { return dispatch_nt<160, 4, 2>(x, y, M, n_img, H, W, wp, w1, w2, s, b, stats, dtype, st); }
int ly_mlp_pconv_80(const void* x, void* y, long M, int n_img, int H, int W, const void* wp, int dtype, hipStream_t st) { return dispatch_pconv<80, 2>(x, y, M, n_img, H, W, wp, dtype, st); }
int ly_mlp_pconv_160(const void* x, void* y, long M, int n_img, int H, int W, const void* wp, int dtype, hipStream_t st) { return dispatch_pconv<160, 4>(x, y, M, n_img, H, W, wp, dtype, st); }
