// Fused MLPBlock backward, C = 80 instantiation (see ly_mlpblock_bwd.hpp)
#include "ly_mlpblock_bwd.hpp"
int ly_mlp_bwd_pass_80(LyMlpBwdArgs P, int pass, long slab_floats, float* dw1, float* dw2, hipStream_t st) { return mlp_bwd_pass<80, 2>(P, pass, slab_floats, dw1, dw2, st); }
int ly_mlp_bwd_dx_80(LyMlpDxArgs P, long slab_floats, float* dwp, int lddw, int ts, int cs, hipStream_t st) { return dispatch_mlp_bwd_dx<80>(P, slab_floats, dwp, lddw, ts, cs, st); }
int ly_mlp_bwd_dx_160(LyMlpDxArgs P, long slab_floats, float* dwp, int lddw, int ts, int cs, hipStream_t st) { return dispatch_mlp_bwd_dx<160>(P, slab_floats, dwp, lddw, ts, cs, st); }
bool ly_mlp_bwd_dx_fits_wide(int C, int W) { return C == 80 ? mlp_bwd_dx_fits<80>(W) : C == 160 ? mlp_bwd_dx_fits<160>(W) : false; }
