// Weight gradient of a 3x3 / stride 1 / pad 1 convolution, bf16 storage, gfx950 (reference: autograd of nn.Conv2d in `Conv`,
// models/common.py:1890-1910, and of Partial_conv3.partial_conv3, models/common.py:1412-1437):
//      dW[o][tap][c] += sum over pixels p of  du[p][o] * x[p + tap][c]
//
// The generic tiled kernel (ly_backward.hip) treats the nine taps as nine separate K tiles: every tap re-gathers x through L2 and re-reads
// the whole du slab (PMC: 3.7-4.9x the algorithmic HBM bytes, MFMA pipe 7-10 % busy), and it transposes both operands in registers on
// their way into [channel][pixel] LDS planes, a layout in which a tap shift is an unaligned access.  Here
//   * both operands stay in their global layout, [pixel][channel], in LDS: staging is a plain 16-byte copy, and the MFMA fragments (the
//     contraction index is the PIXEL) come out through transposed reads (ds_read_b64_tr_b16: 16 lanes fetch a 4-pixel x 16-channel block
//     and receive its transpose).  Every lane supplies its own address, so
//   * x is staged ONCE per tile as a 10 x 10 halo of an 8 x 8 pixel tile and all nine taps read it in place: a tap is a constant offset
//     of (ky*10 + kx) halo positions.  Zero padding and ragged map edges are zeros in the halo / zero du rows: no masks in the loop.
//   * block = (64 output channels) x (32 input channels x 9 taps) accumulators over a run of tiles; waves 2 (o) x 2 (tap halves).
// LDS: du tile [64 px][160 B] (128 B data + 32: the 8 pixel rows x 32 B of a half-wave read hit all 64 banks) | halo [100][96 B] (64 + 32).
#include "ly_tile.hpp"
#include "ly_params.h"
#include <stdlib.h>

#define W3_BN 64
#define W3_CK 32
#define W3_RSA 160
#define W3_RSX 96
#define W3_DU_BYTES (64 * W3_RSA)
#define W3_X_BYTES (100 * W3_RSX)

typedef short w3_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x4 w3_tr(const char* p) {
  typedef __attribute__((address_space(3))) w3_s16x4 lds_s16x4;
  return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p));
}

__global__ __launch_bounds__(LY_THREADS) __attribute__((amdgpu_waves_per_eu(4))) void ly_wgrad3_kernel(const LyWgradParams P, const int tiles_x,
                                                                                                        const int tiles_y, const int n_n,
                                                                                                        const int n_c, const int tchunk,
                                                                                                        const int total_tiles, float* __restrict__ const slab) {
  __shared__ __attribute__((aligned(16))) char w3_lds[W3_DU_BYTES + W3_X_BYTES];
  char* const ds = w3_lds;
  char* const xs = w3_lds + W3_DU_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  int b = ly_xcd_remap((int)blockIdx.x, (int)gridDim.x);              // the (output, input) tile combinations of one pixel chunk (same du / x tiles) on one XCD's L2
  const int cn = b % n_n; b /= n_n;
  const int cc = b % n_c;
  const int chunk = b / n_c;
  const int n0 = cn * W3_BN, c0 = cc * W3_CK;
  const int t_begin = chunk * tchunk;
  const int t_end = t_begin + tchunk < total_tiles ? t_begin + tchunk : total_tiles;
  if (t_begin >= t_end) return;
  const __bf16* const du = reinterpret_cast<const __bf16*>(P.du);
  const __bf16* const x = reinterpret_cast<const __bf16*>(P.x);
  const int H = P.H, W = P.W;
  const int tiles_img = tiles_x * tiles_y;

  // ---- staging plan (the same for every tile) ----
  // du: pieces (pixel slot, 16-byte piece of the 64-channel row): slots tid>>3 and 32 + tid>>3, piece tid & 7
  const int dpc = tid & 7, dpx = tid >> 3;
  const bool d_ok = n0 + 8 * dpc < P.N;
  // halo: pieces (position, 16-byte piece of the 32-channel row): positions tid>>2 and 64 + tid>>2 (< 100), piece tid & 3
  const int xpc = tid & 3;
  const int xp0 = tid >> 2, xp1 = 64 + (tid >> 2);
  const int hy0 = xp0 / 10, hx0 = xp0 - hy0 * 10;
  const int hy1 = xp1 / 10, hx1 = xp1 - hy1 * 10;
  const bool x_ok = c0 + 8 * xpc < P.Cin;
  const bool x1_live = xp1 < 100;
  // ONE register set (the next tile's pieces are in flight during the contraction) and as many resident blocks as the register file holds:
  // measured, a second register set (two tiles ahead) costs the third / fourth block per CU and loses (L16 shape: 84 -> 100 us)
  struct Regs {
    ly_u32x4 d[2], x[2];
    bool okd[2], okx[2];
  };
  Regs R0;
  auto prefetch = [&](Regs& R, int tile) {
    const int n = tile / tiles_img;
    const int r = tile - n * tiles_img;
    const int tyi = r / tiles_x, txi = r - tyi * tiles_x;
    const int oy0 = tyi * 8, ox0 = txi * 8;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int slot = dpx + 32 * e;
      const int oy = oy0 + (slot >> 3), ox = ox0 + (slot & 7);
      R.okd[e] = d_ok && oy < H && ox < W;
      const long off = R.okd[e] ? (((long)n * H + oy) * W + ox) * P.lddu + n0 + 8 * dpc : 0;
      R.d[e] = *reinterpret_cast<const ly_u32x4*>(du + off);
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int iy = oy0 - 1 + (e ? hy1 : hy0), ix = ox0 - 1 + (e ? hx1 : hx0);
      R.okx[e] = x_ok && (e == 0 || x1_live) && iy >= 0 && iy < H && ix >= 0 && ix < W;
      const long off = R.okx[e] ? (((long)n * H + iy) * W + ix) * P.ldx + c0 + 8 * xpc : 0;
      R.x[e] = *reinterpret_cast<const ly_u32x4*>(x + off);
    }
  };
  auto commit = [&](const Regs& R) {
    const ly_u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int e = 0; e < 2; ++e) *reinterpret_cast<ly_u32x4*>(ds + (dpx + 32 * e) * W3_RSA + 16 * dpc) = R.okd[e] ? R.d[e] : z;
    *reinterpret_cast<ly_u32x4*>(xs + xp0 * W3_RSX + 16 * xpc) = R.okx[0] ? R.x[0] : z;
    if (x1_live) *reinterpret_cast<ly_u32x4*>(xs + xp1 * W3_RSX + 16 * xpc) = R.okx[1] ? R.x[1] : z;
  };

  // ---- fragment addressing (tile-invariant) ----
  // transposed read: lane (li, lq) fetches the 8-byte chunk li & 3 of row 4*lq + (li >> 2) (and of row + 16) of a 32-row k-step and
  // receives column li of the 4 x 16 block its 16-lane group fetched: k = 4*lq .. 4*lq + 3 (and + 16), the k-set of ly_tile.hpp
  const int wn = wave & 1, wk = wave >> 1;
  const int r0 = 4 * lq + (li >> 2);
  const int chunk8 = (li & 3) * 8;
  const char* a_base = ds + r0 * W3_RSA + (wn * 32) * 2 + chunk8;            // + ks*32*RSA (+16*RSA) + i*32
  int hb[2][2];                                                              // halo byte offset of the lane's pixels [k-step][row / row + 16]
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int t = ks * 32 + 16 * e + r0;
      hb[ks][e] = ((t >> 3) * 10 + (t & 7)) * W3_RSX + chunk8;
    }
  // the wave's nine column tiles: ct = 9*wk + j -> tap ct >> 1, channel half ct & 1
  int coff[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    const int ct = 9 * wk + j;
    const int tap = ct >> 1;
    const int ky = tap / 3, kx = tap - 3 * ky;
    coff[j] = (ky * 10 + kx) * W3_RSX + (ct & 1) * 32;
  }

  f32x4 acc[2][9];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[i][j] = ly_zero4();

  auto contract = [&]() {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const char* p = a_base + (ks * 32) * W3_RSA + i * 32;
        af[i] = ly_cat8(w3_tr(p), w3_tr(p + 16 * W3_RSA));
      }
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        const bf16x8 bf = ly_cat8(w3_tr(xs + hb[ks][0] + coff[j]), w3_tr(xs + hb[ks][1] + coff[j]));
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][j] = ly_mfma_bf16(af[i], bf, acc[i][j]);
      }
    }
  };
  prefetch(R0, t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();                       // the previous tile's fragments have been read
    commit(R0);
    prefetch(R0, tile + 1 < t_end ? tile + 1 : tile);      // in flight during the contraction (the last tile re-requests itself: no load under a branch)
    __syncthreads();
    contract();
  }

  // ---- flush: D lane (li, lq) register r = dW[row 4*lq + r of the 16-row tile][column li] ----
  if (slab) {
    // the block's tile [64][288] (column = 16*ct + li) into its own slab: plain stores, 16 lanes = 64 contiguous bytes; ly_wgrad3_combine folds
    // the chunks in a fixed order and adds the result to dw
    float* const sl = slab + ((long)(chunk * n_c + cc) * n_n + cn) * (W3_BN * 288);
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const int ct = 9 * wk + j;
      if (c0 + (ct & 1) * 16 + li >= P.c_valid) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wn * 32 + 16 * i + 4 * lq + r;
          if (n0 + row < P.n_valid) sl[row * 288 + 16 * ct + li] = acc[i][j][r];
        }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    const int ct = 9 * wk + j;
    const int tap = ct >> 1;
    const int cch = c0 + (ct & 1) * 16 + li;
    if (cch >= P.c_valid) continue;
    const long cidx = (long)tap * P.dw_ts + (long)cch * P.dw_cs;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = n0 + wn * 32 + 16 * i + 4 * lq + r;
        if (row < P.n_valid) atomicAdd(P.dw + (long)row * P.lddw + cidx, acc[i][j][r]);
      }
  }
}

// dw[row][tap][c] += sum over chunks of slab[chunk][combo][row][16*ct + li]   (fixed order; one writer per element)
// block = 64 consecutive slab columns x 16 chunk lanes; grid.x covers the n_c * n_n * 64 * 288 / 64 column groups of one chunk
__global__ __launch_bounds__(1024) void ly_wgrad3_combine_kernel(const LyWgradParams P, const float* __restrict__ slab, const int chunks, const int n_n,
                                                                const int n_c, const int rls) {
  __shared__ float red[16][64];                            // rls row lanes walk the chunks (see ly_wgrad_combine_body, ly_backward.hip)
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const long E = (long)n_c * n_n * (W3_BN * 288);
  const long e = (long)blockIdx.x * 64 + cl;
  // decode (all lanes of the block share combo and row: 288 = 4.5 * 64, so a 64-column group may straddle two rows — decode per lane)
  const int combo = (int)(e / (W3_BN * 288));
  const int rem = (int)(e - (long)combo * (W3_BN * 288));
  const int row = rem / 288, col = rem - row * 288;
  const int cn = combo % n_n, cc = combo / n_n;
  const int ct = col >> 4, li = col & 15;
  const int orow = cn * W3_BN + row;
  const int cch = cc * W3_CK + (ct & 1) * 16 + li;
  const bool live = e < E && orow < P.n_valid && cch < P.c_valid;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (live) {
    const float* p = slab + e;                               // four loads in flight per lane (see ly_wgrad_combine_kernel)
    int c = rl;
    for (; c + 7 * rls < chunks; c += 8 * rls) {        // eight loads fenced ahead of the adds (the scheduler sinks them back otherwise)
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = p[(long)(c + rls * k) * E];
      __builtin_amdgcn_sched_barrier(0);
      a0 += v[0]; a1 += v[1]; a2 += v[2]; a3 += v[3];
      a0 += v[4]; a1 += v[5]; a2 += v[6]; a3 += v[7];
    }
    for (; c + 3 * rls < chunks; c += 4 * rls) {
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = p[(long)(c + rls * k) * E];
      __builtin_amdgcn_sched_barrier(0);
      a0 += v[0]; a1 += v[1]; a2 += v[2]; a3 += v[3];
    }
    if (c < chunks) a0 += p[(long)c * E];
    if (c + rls < chunks) a1 += p[(long)(c + rls) * E];
    if (c + 2 * rls < chunks) a2 += p[(long)(c + 2 * rls) * E];
  }
  red[rl][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (rl == 0 && live) {
    float s = 0.f;
    for (int i = 0; i < rls; ++i) s += red[i][cl];
    float* d = P.dw + (long)orow * P.lddw + (long)(ct >> 1) * P.dw_ts + (long)cch * P.dw_cs;
    *d += s;
  }
}

// true when the problem is one this kernel takes (bf16, 3x3, stride 1, pad 1, NHWC, 16-byte friendly widths and pointers)
bool ly_wgrad3_ok(const LyWgradParams& P) {
  return P.dtype == LY_BF16 && P.ks == 3 && P.stride == 1 && P.pad == 1 && !P.nchw && !P.up2 && P.Hin == P.H && P.Win == P.W &&
         (P.N & 7) == 0 && (P.Cin & 7) == 0 && (P.lddu & 7) == 0 && (P.ldx & 7) == 0 && (reinterpret_cast<uintptr_t>(P.du) & 15) == 0 &&
         (reinterpret_cast<uintptr_t>(P.x) & 15) == 0;
}

int ly_wgrad3_launch(const LyWgradParams& P, hipStream_t st) {
  const int tiles_x = (P.W + 7) / 8, tiles_y = (P.H + 7) / 8;
  const long n_img = P.M / ((long)P.H * P.W);
  const long total = n_img * tiles_x * tiles_y;
  LY_CHECK(total < (1L << 30), "wgrad3: too many tiles");
  const int n_n = (P.N + W3_BN - 1) / W3_BN, n_c = (P.Cin + W3_CK - 1) / W3_CK;
  // ~1024 blocks (four per CU); every block adds the valid part of its 64 x 288 tile to dw once
#ifdef LY_DEVEL
  static int target = 0;                                  // development builds (make DEVEL=1) read LY_W3_BLOCKS / LY_W3_ATOMIC
  if (!target) { const char* e = getenv("LY_W3_BLOCKS"); target = e ? atoi(e) : 768; }
  static const bool force_atomic = getenv("LY_W3_ATOMIC") != nullptr;
#else
  constexpr int target = 768;
  constexpr bool force_atomic = false;
#endif
  long chunks = (target + (long)n_n * n_c - 1) / ((long)n_n * n_c);
  if (chunks > total) chunks = total;
  if (chunks < 1) chunks = 1;
  const long tchunk = (total + chunks - 1) / chunks;
  chunks = (total + tchunk - 1) / tchunk;
  // partial tiles by plain stores + one combine launch when the scratch holds them; float atomics otherwise
  const long need = chunks * n_n * n_c * (long)(W3_BN * 288);
  float* slab = (P.ws && need <= P.ws_floats && !force_atomic) ? P.ws : nullptr;
  hipLaunchKernelGGL(ly_wgrad3_kernel, dim3((unsigned)(chunks * n_n * n_c)), dim3(LY_THREADS), 0, st, P, tiles_x, tiles_y, n_n, n_c, (int)tchunk,
                     (int)total, slab);
  if (slab) {
    const long E = (long)n_n * n_c * (W3_BN * 288);
    const int rls = chunks <= 192 ? 4 : 16;                  // row lanes of the combine (96 chunks: 16.7 us with 16, 13.7 with 4 or 8)
    hipLaunchKernelGGL(ly_wgrad3_combine_kernel, dim3((unsigned)((E + 63) / 64)), dim3(64 * rls), 0, st, P, slab, (int)chunks, n_n, n_c, rls);
  }
  LY_LAUNCH_CHECK();
  return 0;
}
