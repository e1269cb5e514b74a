/* lead_yolo_hip.h — C ABI of libleadyolo_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary of the LEAD-YOLO hot path: every entry point replaces the arithmetic of one
 * reference nn.Module.forward (file:line cited per function, relative to qingqing-zijin/LEAD-YOLO).
 * Plain pointers and sizes only; all pointers are DEVICE pointers unless stated; `stream` is a
 * hipStream_t (0 = default stream); nothing here synchronises, allocates or frees.
 * Activations are NHWC fp32 (N, H, W, C with C fastest).  Return value: 0 on success, <0 on error
 * (message via ly_last_error(), thread-local).
 */
#ifndef LEAD_YOLO_HIP_H
#define LEAD_YOLO_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

int ly_abi_version(void);
const char* ly_last_error(void);

/* FasterNet MLPBlock forward, eval form (BN folded to scale/shift):
 *   y = x + W2 . relu(scale * (W1 . [pconv3x3(x[..., :C/4]) | x[..., C/4:]]) + shift)
 * replaces Partial_conv3.forward_split_cat (models/common.py:1432-1437) + MLPBlock.forward
 * (models/common.py:1478-1482).  wp/w1/w2 are bf16x3 frag-packed (lead-yolo_amd/pack.py frag_pack3)
 * from spatial_mixing.partial_conv3.weight ([C/4, 9*ceil4(C/4)], k = tap*ceil4(C/4) + c),
 * mlp.0.weight (rows zero-padded to 16*ly_mlpblock_hidden_tiles(C)) and mlp.3.weight; bn_* have
 * 16*ly_mlpblock_hidden_tiles(C) entries (zero padded).  stats: NULL, or [2 * 16*hidden_tiles] zeroed
 * accumulators = STATISTICS PASS of the train-mode BatchNorm (sum / sum of squares of the pre-BN hidden
 * activations; y and bn_* are ignored, nothing is stored).  x and y must not alias.
 * Built for C in {16,24,40,80,160,320}. */
int ly_mlpblock_fwd(const float* x, float* y, int n_img, int H, int W, int C, const void* wp, const void* w1,
                    const void* w2, const float* bn_scale, const float* bn_shift, float* stats, void* stream);
/* ablation aid for profiling (4: skip halo staging, 8: skip stores) */
int ly_debug_set_mlp(int v);
/* tuning aid: 1 forces the flattened-run tiling (default: 8x16 patches where W % 16 == 0 and W >= 64);
 * 2 / 4 / 8 force flattened runs with 2 / 4 / 1 pixel tiles per wave where built */
int ly_debug_set_mlp_tile(int v);
/* hidden (2C) channel tiles of 16, padded to an even count */
int ly_mlpblock_hidden_tiles(int C);

/* ---- generic pointwise-convolution GEMM ------------------------------------------------------ */
enum { LY_ACT_NONE_ = 0, LY_ACT_RELU_ = 1, LY_ACT_SILU_ = 2 };          /* `act` values            */
enum { LY_GATHER_ROWS = 0,       /* A row m = a0[m, :k0] | a1[m, :K-k0]                              */
       LY_GATHER_UP2 = 1,        /* a0 is at half resolution: row (n, h/2, w/2)  (nearest 2x upsample) */
       LY_GATHER_PATCH = 2,      /* k x k stride-k patches of an NHWC tensor (PatchMerging)           */
       LY_GATHER_PATCH_NCHW = 3  /* 4 x 4 stride-4 patches of an NCHW tensor (PatchEmbed on images)   */ };
enum { LY_PRO_NONE = 0,
       LY_PRO_GATE = 1,              /* a0 part: x * g_w[n,w,:] * g_h[n,h,:] (+ res)  (CoordAtt gating) */
       LY_PRO_AFFINE_RELU_CA = 2     /* relu(x*p_scale + p_shift) * p_ca[n,:]         (RFCBAMConv k=1)  */ };

typedef struct LyGemmParams {
  long M;                 /* output pixels = n_img * H * W                                       */
  int H, W;               /* output spatial size                                                 */
  int K, N;               /* contraction length (multiple of 4), output channels                 */
  const float* a0; int lda0; int k0;   /* first source: row stride (floats), columns taken       */
  const float* a1; int lda1;           /* second source for columns [k0, K) or NULL               */
  int gather;             /* LY_GATHER_*                                                         */
  int Hin, Win, Cin, ks, pk;           /* patch gathers: input size, kernel=stride, pk = ks*lda0 */
  int pro;                /* LY_PRO_*                                                            */
  const float* g_h; const float* g_w;  /* [n_img, H, k0], [n_img, W, k0]                         */
  const float* res; int ldres;         /* optional residual added to the gated a0 part           */
  const float* p_scale; const float* p_shift; const float* p_ca;   /* [K], [K], [n_img, K]       */
  const void* wp;         /* bf16x3 frag-packed weights of W[N, K] (pack.frag_pack3)             */
  const float* e_scale; const float* e_shift;   /* [N] or NULL (=1 / =0)                         */
  const float* rowscale;  /* [M] or NULL                                                         */
  int act;                /* 0 none, 1 relu, 2 silu                                              */
  float* out; int ldo;    /* output row stride (floats); pointer may be pre-offset into a concat */
  float* stats;           /* NULL, or [2N] zero-initialised accumulators: STATISTICS PASS for train-mode
                             BatchNorm — adds sum / sum-of-squares over the M rows of the pre-activation value
                             (rowscale*e_scale*acc + e_shift) per output channel; stores nothing when out is NULL,
                             otherwise stores act(that value) as usual (one launch for statistics + pre-BN tensor) */
} LyGemmParams;

/* out[m, n] = act(rowscale[m] * e_scale[n] * sum_k A'[m, k] W[n, k] + e_shift[n]).
 * Replaces nn.Conv2d(k=1)+BatchNorm2d+SiLU/ReLU (effective Conv, models/common.py:1890-1910), the
 * torch.cat / nn.Upsample feeding it (models/common.py:531-538), CoordAtt's gating multiply
 * (models/common.py:1608), RFCBAMConv's k=1 path (models/rfa.py:113-129) and the FasterNet patch
 * convolutions (models/common.py:1528-1561), depending on gather/pro. */
int ly_gemm_fwd(const LyGemmParams* p, void* stream);
/* tuning aid: force the GEMM tile configuration (NT*100 + MT*10 + WC), 0 = heuristic */
int ly_debug_set_gemm_cfg(int cfg);
/* ablation aid for profiling (bit 0: skip split+LDS write, 1: skip MFMA, 2: skip prefetch loads, 3: skip stores) */
int ly_debug_set_gemm(int v);
/* tuning aid: K stage of ly_gemm_fwd, 0 / 64 = default, 128 = 128-wide stage (measured slower) */
int ly_debug_set_gemm_bk(int v);
/* A/B aid: 1 (default) = two-deep prefetch pipeline of ly_gemm_fwd, 0 = one-deep */
int ly_debug_set_gemm_d2(int v);


/* ---- 3x3 / s1 / p1 convolution (implicit GEMM) ------------------------------------------------- */
typedef struct LyConv3Params {
  long M; int H, W;        /* output (= input) pixels, spatial size                              */
  int Cin, N;              /* input channels (multiple of 4), output channels                    */
  int TH, TW;              /* pixel patch per block: TH*TW <= 128, (TH+2)*(TW+2) <= 192          */
  const float* x; int ldx; /* NHWC input, row stride (floats)                                    */
  const void* wp;          /* frag_pack3(conv_taps_matrix(weight, 32)): k = tap*ceil32(Cin) + c  */
  const float* e_scale; const float* e_shift;   /* folded BN / bias, [N] or NULL                 */
  int act;
  float* out; int ldo;
  float* stats;            /* NULL or [2N] accumulators: statistics pass (see LyGemmParams.stats)     */
} LyConv3Params;

/* Conv(c1, c2, 3, 1) = conv3x3(no bias) + BN + SiLU (CA_Bottleneck.cv2, models/common.py:1617,
 * 1890-1910). */
int ly_conv3x3_fwd(const LyConv3Params* p, void* stream);
/* ablation aid for profiling (bit 1: skip LDS reads + MFMA, 3: skip commit); the load switches (bits 0, 2) were removed:
 * a load under a run-time branch makes the compiler drain the prefetch queue after every tap */
int ly_debug_set_conv3(int v);
/* tuning aid: force the wave layout (MT*10 + WC), 0 = heuristic */
int ly_debug_set_conv3_cfg(int v);


/* ---- CoordAtt (models/common.py:1583-1609) ------------------------------------------------------- */
/* pool[n, pos, c]: pos < H -> mean over w of row pos; pos >= H -> mean over h of column pos-H.       */
int ly_pool_hw(const float* x, int ldx, int n_img, int H, int W, int C, float* pool, void* stream);
/* y = h_swish(w1 . pool + b1) (bn1 folded into w1/b1: [mip, C], [mip]); a_h[n,h,:] = sigmoid(wh . y + bh),
 * a_w[n,w,:] = sigmoid(ww . y + bw); wh/ww are [C, mip].                                              */
int ly_coordatt_mlp(const float* pool, int n_img, int H, int W, int C, int mip, const float* w1, const float* b1,
                    const float* wh, const float* bh, const float* ww, const float* bw, float* a_h, float* a_w,
                    void* stream);

/* out = x * a_w[n,w,:] * a_h[n,h,:] (+ res): the gating multiply as a standalone pass (only used
 * when no consumer GEMM can absorb it, e.g. CA_Bottleneck with a residual shortcut).               */
int ly_coordatt_gate(const float* x, int ldx, int n_img, int H, int W, int C, const float* a_h, const float* a_w,
                     const float* res, int ldres, float* out, int ldo, void* stream);

/* ---- RFCBAMConv (models/rfa.py:77-129) ----------------------------------------------------------- */
/* rfa.SE: ca[n, :] = sigmoid(wb . relu(wa . mean_hw(x)));  wa [R, C], wb [C, R]; part = workspace
 * of n_img * slices * C floats.                                                                     */
int ly_se_fwd(const float* x, int ldx, int n_img, int HW, int C, const float* wa, const float* wb, int R, float* part,
              int slices, float* ca, void* stream);
/* mm[n, y, x, 0:2] = (max_c, mean_c) of relu(bn(generate(x))) on the k-times expanded grid.
 * k = 1: a1/b1 = folded per-channel scale/shift.  k = 3: wg = pack.rfcbam_gen_weights(..., 32, False):
 * [C32/32][4 waves][9 t][4 channel pairs][20]; TH x TW (<= 64) = output-pixel tile per block.       */
int ly_rfcbam_stats(const float* x, int ldx, int n_img, int H, int W, int C, int k, int s, const float* wg,
                    const float* a1, const float* b1, int TH, int TW, float* mm, void* stream);
/* ablation aid for profiling (bit 0: skip staging, 1: skip generate/reduce) */
int ly_debug_set_stats3(int v);
/* rfa[n, y, x] = sigmoid(conv3x3_pad1(mm; w[2][3][3]))   (get_weight, models/rfa.py:107)             */
int ly_rfa_map(const float* mm, int n_img, int HK, int WK, const float* w, float* rfa, void* stream);

typedef struct LyRfcbam3Params {
  int n_img, H, W, C;          /* input NHWC                                                      */
  int Ho, Wo, N, s;            /* output size, output channels, stride                            */
  int TH, TW;                  /* output-pixel tile per block (TH*TW <= 64)                       */
  const float* x; int ldx;
  const float* wg;             /* pack.rfcbam_gen_weights(..., 16, True): [C/16][4][9 t][2][20]   */
  const float* ca;             /* [n_img, C]                                                      */
  const float* rfa;            /* [n_img, 3Ho, 3Wo]                                               */
  const void* wp;              /* frag_pack3(conv.0.weight as [N, C/16, 144 -> 160 zero padded])  */
  const float* e_scale; const float* e_shift;   /* conv.1 BN folded with conv.0.bias              */
  float* out; int ldo;
  float* stats;                /* NULL or [2N] accumulators: conv.1 BatchNorm statistics pass    */
  int linear;                  /* != 0: store the affine value without the ReLU (backward recompute) */
} LyRfcbam3Params;
/* RFCBAMConv kernel_size 3 main contraction (+ReLU); the k=1 case is ly_gemm_fwd with
 * LY_PRO_AFFINE_RELU_CA and rowscale = rfa.                                                         */
int ly_rfcbam3_fwd(const LyRfcbam3Params* p, void* stream);
/* ablation aid for profiling (bit 0: skip regenerate, 1: generate weights through LDS whatever the grid, 2: skip staging,
 * 3: two 128-channel groups instead of the 256-channel tile for N > 128); 0 = normal */
int ly_debug_set_rf3(int v);


/* ---- graph remainder ------------------------------------------------------------------------- */
/* SPPF pooling (models/common.py:348-366): out[n, p, :] = [x | m(x) | m(m(x)) | m(m(m(x)))] with m =
 * k x k / stride 1 / pad k//2 max pool; out row stride ldo >= 4C.  The map must fit LDS
 * (2*H*W*4 floats <= 160 KiB); C, ldx, ldo multiples of 4.                                                                    */
int ly_sppf_pool(const float* x, int ldx, int n_img, int H, int W, int C, int k, float* out, int ldo, void* stream);
/* Detect tail (models/yolo.py:95-120): y[n,h,w,na*no] (row stride ldy) -> p[n,na,h,w,no] and, if z is
 * not NULL, decoded rows z[n, zoff + (a*H + h)*W + w, :] of a [n_img, zrows, no] tensor;
 * anchors = [na,2] in grid units (Detect.anchors[i]), stride = Detect.stride[i].                     */
int ly_detect_tail(const float* y, int ldy, int n_img, int H, int W, int na, int no, const float* anchors, float stride,
                   float* p, float* z, long zrows, long zoff, void* stream);


/* ---- train-mode BatchNorm statistics passes ----------------------------------------------------- */
/* mom[c] += sum_rows x[r, c], mom[C + c] += sum_rows x[r, c]^2 over an [rows, C] row matrix (mom zeroed
 * by the caller).  Used for the k=1 `generate` BatchNorm of RFCBAMConv (models/rfa.py:101-106).       */
int ly_chan_moments(const float* x, int ldx, long rows, int C, float* mom, void* stream);
/* CoordAtt bn1 (models/common.py:1589,1602): sum / sum of squares of conv1(pool) + bias over all
 * n*(H+W) positions, stats[0:mip] and stats[mip:2mip] (zeroed by the caller).                        */
int ly_coordatt_conv1_stats(const float* pool, long positions, int C, int mip, const float* w1, const float* b1,
                            float* stats, void* stream);

/* RFCBAMConv k=3 `generate` BatchNorm statistics (train mode): mom[54][C] (zeroed by the caller) receives,
 * per input channel, the 9 first moments sum x_u and the 45 second moments sum x_u x_v (u <= v, row-major
 * upper triangle) of the zero-padded stride-s 3x3 taps over all n_img*Ho*Wo output pixels; mean and
 * variance of every (channel, tap-output) follow as w.m and w^T M w on the host.                        */
int ly_rfcbam_tap_moments(const float* x, int ldx, int n_img, int H, int W, int C, int s, float* mom, void* stream);

/* ---- backward building blocks of the training step (train.py:324 `scaler.scale(loss).backward()`) -------------
 * Data gradients of 1x1 / 3x3 stride-1 convolutions reuse ly_gemm_fwd / ly_conv3x3_fwd with transposed weights.   */

/* BatchNorm(train)+activation backward over an [rows, C] matrix.  Forward was v = a[c]*u + b[c], y = act(v)
 * (models/common.py:1906-1907 Conv.forward; :1478-1482 MLPBlock).  With dv = dy * act'(v):
 *   reduce: sums[c] += sum_r dv, sums[C + c] += sum_r dv*u   (sums zeroed by the caller)
 *   apply : du = alpha[c]*dv + kappa[c] + lambda[c]*u        (du may alias u or dy)                               */
/* y = act(a[c]*u + b[c]) over [rows, C]: the normalisation + activation half of a train-mode unit whose contraction pass
 * (stats != NULL AND out != NULL: statistics accumulated and the pre-BN value stored in ONE launch) produced u.            */
int ly_bnact_fwd(const float* u, int ldu, long rows, int C, const float* a, const float* b, int act, float* y, int ldy, void* stream);
int ly_bnact_bwd_reduce(const float* dy, int lddy, const float* u, int ldu, long rows, int C, const float* a, const float* b,
                        int act, float* sums, void* stream);
int ly_bnact_bwd_apply(const float* dy, int lddy, const float* u, int ldu, long rows, int C, const float* a, const float* b,
                       int act, const float* alpha, const float* kappa, const float* lambda, float* du, int lddu, void* stream);

/* Weight gradient of a convolution whose forward read input pixel (ho*stride + ky - pad, wo*stride + kx - pad):
 *   dw[n][(ky*ks + kx)*Cin + c] += sum_{m = (img, ho, wo)} du[m][n] * x[img, hi, wi, c]      (zero outside the map)
 * x is an NHWC map (row stride ldx) of Hin x Win pixels, or an NCHW image when nchw != 0, or — when up2 != 0 — a
 * half-resolution map read through nearest-2x upsampling (Hin, Win are the SOURCE sizes).  dw rows have stride lddw
 * and must be zeroed by the caller (partial sums of pixel chunks are added atomically).                            */
typedef struct LyWgradParams {
  long M;            /* output pixels = n_img * H * W */
  int H, W;          /* output map */
  int N;             /* output channels */
  const float* du; int lddu;
  const float* x; int ldx;
  int Hin, Win, Cin;
  int ks, stride, pad;
  int nchw, up2;
  float* dw; int lddw;
} LyWgradParams;
int ly_wgrad(const LyWgradParams* p, void* stream);
/* tuning aid: 1 = the 64 x 256 output tile for every N <= 64 shape (default: 32 x 128 / 64 x 128) */
int ly_debug_set_wgrad_tile(int v);

/* Adjoint of the nearest-2x read: out[n,h,w,:] = sum of d[n, 2h+{0,1}, 2w+{0,1}, :]  (d is a 2Hs x 2Ws map).      */
int ly_up2_bwd(const float* d, int ldd, int n_img, int Hs, int Ws, int C, float* out, int ldo, void* stream);
/* Adjoint of the k = s patch gather: g[m][(ky,kx,c)] -> dx[n, ks*ho+ky, ks*wo+kx, c] (dense NHWC, C % 4 == 0).    */
int ly_unpatch(const float* g, int n_img, int Ho, int Wo, int C, int ks, float* dx, void* stream);

/* CoordAtt backward (models/common.py:1595-1609).  Gate out = x*a_h[n,h,:]*a_w[n,w,:]:
 *   dx = dout*a_h*a_w,  da_h[n,h,c] += sum_w dout*x*a_w,  da_w[n,w,c] += sum_h dout*x*a_h  (caller zeroes both).
 * Pools pool[n,0:H]=mean_w x, pool[n,H:H+W]=mean_h x:  dx[n,h,w,c] = gp[n,h,c]/W + gp[n,H+w,c]/H.                  */
int ly_coordatt_gate_bwd(const float* dout, int ldd, const float* x, int ldx, int n_img, int H, int W, int C, const float* a_h,
                         const float* a_w, float* dx, int lddx, float* da_h, float* da_w, void* stream);
int ly_pool_hw_bwd(const float* gp, int n_img, int H, int W, int C, float* dx, int lddx, void* stream);
/* k x k / stride 1 / pad k//2 max-pool backward (SPPF, models/common.py:348-366): dx[argmax of window] += dy
 * (first maximum in row-major order, as ATen); dx is ACCUMULATED into, which lets the three chained pools add
 * into the gradient slots of the SPPF concat buffer in place.                                                       */
int ly_maxpool_bwd(const float* x, int ldx, const float* dy, int lddy, int n_img, int H, int W, int C, int k, float* dx, int lddx,
                   void* stream);

/* ---- RFCBAMConv backward (models/rfa.py:113-129 under autograd) ----------------------------------------------------
 * Expanded tensors are [m = (n, ho, wo)][t = tap][c] (channel contiguous), maps are [n, k*Ho, k*Wo]; per-(t, c) vectors
 * (ag, bg, alpha, kappa, lambda, sums) are indexed t*C + c.  See csrc/ly_rfcbam_bwd.hip for the step list.            */
/* ug[m][t][c] = sum_u wg[c*KK + t][u] * x_u(m)[c]  (depthwise `generate` conv before its BatchNorm)                    */
int ly_rf_generate(const float* x, int ldx, int n_img, int H, int W, int C, int k, int s, const float* wg, float* ug, void* stream);
/* G = relu(ag*ug + bg):  cd = G*ca*rfa (written),  d_rfa[pos] += sum_c dcd*G*ca,  gmax[pos] = max_c G (as float bits,
 * zeroed by the caller),  d_ca[n][c] += sum_{m,t} dcd*rfa*G  (d_rfa, d_ca zeroed by the caller)                        */
int ly_rf_bwd_attn(int n_img, int H, int W, int C, int k, int s, const float* ug, const float* dcd, const float* ag, const float* bg,
                   const float* ca, const float* rfa, float* cd, float* d_rfa, float* gmax, float* d_ca, void* stream);
/* rfa = sigmoid(conv3x3(mm; w18)) backward: d_mm[n,y,x,2] (written) and dw18[18] (+=, zeroed by the caller)            */
int ly_rfa_bwd(const float* d_rfa, const float* rfa, const float* mm, const float* w18, int n_img, int Hk, int Wk, float* d_mm,
               float* dw18, void* stream);
/* dv = [G>0]*(dcd*rfa*ca + d_mm.mean/C + [G==gmax]*d_mm.max) written over dcd; sums[t*C+c] += dv, sums[C*KK + ..] += dv*ug */
int ly_rf_bwd_relu(int n_img, int H, int W, int C, int k, int s, const float* ug, float* dcd, const float* ag, const float* bg,
                   const float* ca, const float* rfa, const float* gmax, const float* d_mm, float* sums, void* stream);
/* dug = alpha*dv + kappa + lambda*ug written over dv; dwg[row][c*KK + t][u] = partial sums of sum_m dug*x_u: a
 * [part_rows][C*KK][KK] matrix zeroed by the caller; every (block, pixel sub-group) stores one row, the caller sums rows */
int ly_rf_bwd_gen(const float* x, int ldx, int n_img, int H, int W, int C, int k, int s, const float* ug, float* dv, const float* alpha,
                  const float* kappa, const float* lambda, float* dwg, int part_rows, void* stream);
/* dx[n,hi,wi,c] = sum over (m, u) reading that input pixel of sum_t dug[m][t][c]*wg[c*KK + t][u]  (written)           */
int ly_rf_bwd_dx(int n_img, int H, int W, int C, int k, int s, const float* dug, const float* wg, float* dx, int lddx, void* stream);

/* ---- per-channel BatchNorm vector work and weight packing, one launch each ------------------------------------------
 * STATISTICS ACCUMULATORS ARE STRIPED: every `stats` / `sums` / `mom` argument of the statistics passes above
 * (ly_gemm_fwd, ly_conv3x3_fwd, ly_rfcbam3_fwd, ly_mlpblock_fwd, ly_chan_moments, ly_bnact_bwd_reduce) is
 * LY_STATS_STRIPES consecutive copies of the documented [2*nch] array (block b adds into copy b % LY_STATS_STRIPES);
 * the caller zeroes all copies and the two entry points below sum them (in double).                                  */
#define LY_STATS_STRIPES 32
/* Train-mode BatchNorm (nn.BatchNorm2d forward, training=True) from the striped sums of channels c_off .. c_off+N-1
 * (second moments at +nch): scale = gamma*invstd, shift = beta - mean*scale (+ bias*scale), batch mean / invstd for
 * the backward, running_mean/var updated with `momentum` (unbiased variance), *nbt += 1.  NULL = not wanted.           */
int ly_bn_finalize(const float* stats, int stripes, int nch, int c_off, int N, double count, const float* gamma, const float* beta,
                   const float* bias, float eps, float momentum, float* running_mean, float* running_var, long* nbt, float* scale,
                   float* shift, float* mean, float* invstd, void* stream);
/* BatchNorm backward coefficients from striped sums [2N] (sum dv, sum dv*u): dgamma, dbeta and
 * du = alpha*dv + kappa + lambda*u  (train != 0: batch statistics; else alpha = a, kappa = lambda = 0).               */
int ly_bn_bwd_coeffs(const float* sums, int stripes, int N, double count, const float* a, const float* mean, const float* invstd, int train,
                     float* dgamma, float* dbeta, float* alpha, float* kappa, float* lambda, void* stream);
/* bf16x3 fragment packing of W[r][k] = w[r*ld_r + k*ld_k] (R x K, rows zero padded to max(R, rows_to)) into the
 * [T][S][2 planes][64 lanes][8] bf16 layout the contraction kernels read (csrc/ly_tile.cuh).                          */
int ly_frag_pack3(const float* w, int R, int K, long ld_r, long ld_k, int rows_to, void* out, void* stream);

/* ---- detection loss on device (utils/loss.py:121-268 ComputeLoss / build_targets, utils/metrics.py:293-354 EIoU) --------
 * One pyramid level, forward and gradient (nc == 1, no focal loss): anchor matching with the reference's candidate order
 * (offset k, anchor a, target t), EIoU box loss with analytic gradient, per-cell "last writer wins" objectness target (the
 * highest candidate index, i.e. the reference's sequential CPU semantics), BCE objectness.
 *   p / dp   [bs][na][ny][nx][no] predictions / their gradient (dp zeroed by the caller; d(total loss)/dp on return)
 *   anchors  [na][2] in grid units (Detect.anchors[i]);  targets [nt][6] = (image, class, x, y, w, h) normalised
 *   tobj [cells] zeroed, winner [cells] filled with -1, cand_cell [5*na*nt], cand [5*na*nt][5] workspaces
 *   acc [4] zeroed: sum(1 - eiou), matches, sum of objectness BCE, rejected target rows  -> ly_loss_finish
 *   tbox     NULL or [5*na*nt][4]: (gx - gi, gy - gj, gw, gh) of every valid candidate (build_targets' `tbox`, utils/loss.py:262)
 *   match_only != 0: target assignment only (utils/loss.py:194-268 build_targets): cand_cell[idx] = flattened cell
 *            ((b*na + a)*ny + gj)*nx + gi of candidate idx = (k*na + a)*nt + t, or -1; tbox as above; p is read, dp/tobj untouched.
 * A target row whose image index is outside [0, bs) or that holds a NaN is rejected and counted in acc[3] (the torch
 * formulation raises IndexError there); ly_loss_finish then returns a NaN total.                                          */
int ly_loss_level(const float* p, float* dp, const float* anchors, const float* targets, int bs, int na, int ny, int nx, int no, long nt,
                  float anchor_t, float box_gain, float obj_gain, float balance, float* tobj, int* winner, long* cand_cell, float* cand,
                  float* acc, float* tbox, int match_only, void* stream);
/* out[0] = (lbox + lobj) * bs (NaN if any level rejected a target row), out[1] = lbox, out[2] = lobj, out[3] = lcls = 0 from
 * acc [nl][4]; cells / balance: [nl] floats                                                                                */
int ly_loss_finish(const float* acc, int nl, const float* cells, const float* balance, float box_gain, float obj_gain, int bs, float* out,
                   void* stream);

#ifdef __cplusplus
}
#endif
#endif
