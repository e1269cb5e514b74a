// MLPBlock instantiations for C = 80 / 160 (see ly_mlpblock.cuh)
#include "ly_mlpblock.cuh"
int ly_mlp_dispatch_80(LY_MLP_ARGS) { return dispatch_nt<80, 2, 2>(x, y, M, n_img, H, W, wp, w1, w2, s, b, stats, dtype, st); }
int ly_mlp_dispatch_160(LY_MLP_ARGS) { return dispatch_nt<160, 4, 2>(x, y, M, n_img, H, W, wp, w1, w2, s, b, stats, dtype, st); }
