// MLPBlock C ABI + the C = 16 / 24 / 40 instantiations; the kernel itself is in ly_mlpblock.hpp.
#include "ly_mlpblock.hpp"

extern "C" int ly_mlpblock_fwd(const void* x, void* y, int n_img, int H, int W, int C, const void* wp, const void* w1,
                               const void* w2, const float* bn_scale, const float* bn_shift, double* stats, int dtype, void* stream) {
  LY_CHECK(dtype == LY_F32 || dtype == LY_BF16, "mlpblock: unknown dtype %d", dtype);
  LY_CHECK(x && wp && w1 && w2 && (stats || (y && bn_scale && bn_shift)), "mlpblock: null pointer");
  LY_CHECK(x != y, "mlpblock: in-place call is not supported (neighbouring tiles read halo rows)");
  LY_CHECK(n_img > 0 && H > 0 && W > 0, "mlpblock: bad shape %d x %d x %d", n_img, H, W);
  LY_CHECK(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "mlpblock: x / y must be 16-byte aligned");
  long M = (long)n_img * H * W;
  LY_CHECK(M < (1L << 24), "mlpblock: M=%ld pixels exceeds the 2^24 limit of the fast index path", M);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  switch (C) {
    case 16:  return dispatch_nt<16, 2, 4>(x, y, M, n_img, H, W, wp, w1, w2, bn_scale, bn_shift, stats, dtype, st);
    case 24:  return dispatch_nt<24, 4, 4>(x, y, M, n_img, H, W, wp, w1, w2, bn_scale, bn_shift, stats, dtype, st);
    case 40:  return dispatch_nt<40, 2, 4>(x, y, M, n_img, H, W, wp, w1, w2, bn_scale, bn_shift, stats, dtype, st);
    case 80:  return ly_mlp_dispatch_80(x, y, M, n_img, H, W, wp, w1, w2, bn_scale, bn_shift, stats, dtype, st);
    case 160: return ly_mlp_dispatch_160(x, y, M, n_img, H, W, wp, w1, w2, bn_scale, bn_shift, stats, dtype, st);
    case 320: return ly_mlp_dispatch_320(x, y, M, n_img, H, W, wp, w1, w2, bn_scale, bn_shift, stats, dtype, st);
    default:
      ly_set_error("mlpblock: unsupported channel count C=%d (built for 16/24/40/80/160/320)", C);
      return -1;
  }
}

// z = [pconv3x3(x[:, :C/4]) | x[:, C/4:]] alone (Partial_conv3.forward_split_cat, models/common.py:1432-1437): returns 0 when launched, 1 when the
// persistent kernel is not built for this (C, map, dtype) — the caller then uses a copy + ly_conv3x3_fwd —, -1 on error.
extern "C" int ly_mlpblock_pconv(const void* x, void* z, int n_img, int H, int W, int C, const void* wp, int dtype, void* stream) {
  LY_CHECK(dtype == LY_F32 || dtype == LY_BF16, "mlpblock_pconv: unknown dtype %d", dtype);
  LY_CHECK(x && z && wp && x != z && n_img > 0 && H > 0 && W > 0, "mlpblock_pconv: bad arguments");
  LY_CHECK(((uintptr_t)x & 15) == 0 && ((uintptr_t)z & 15) == 0, "mlpblock_pconv: x / z must be 16-byte aligned");
  const long M = (long)n_img * H * W;
  LY_CHECK(M < (1L << 24), "mlpblock_pconv: M=%ld pixels exceeds the 2^24 limit of the fast index path", M);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  switch (C) {
    case 16: return dispatch_pconv<16, 2>(x, z, M, n_img, H, W, wp, dtype, st);
    case 24: return dispatch_pconv<24, 4>(x, z, M, n_img, H, W, wp, dtype, st);
    case 40: return dispatch_pconv<40, 2>(x, z, M, n_img, H, W, wp, dtype, st);
    case 80: return ly_mlp_pconv_80(x, z, M, n_img, H, W, wp, dtype, st);
    case 160: return ly_mlp_pconv_160(x, z, M, n_img, H, W, wp, dtype, st);
    case 320: return ly_mlp_pconv_320(x, z, M, n_img, H, W, wp, dtype, st);
    default: return 1;
  }
}

// geometry the host packer needs: hidden tiles (padded to even) for a given C
extern "C" int ly_mlpblock_hidden_tiles(int C) { return (2 * C / 16 + 1) / 2 * 2; }
