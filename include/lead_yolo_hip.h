/* lead_yolo_hip.h — C ABI of libleadyolo_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary of the LEAD-YOLO hot path: every entry point replaces the arithmetic of one
 * reference nn.Module.forward (file:line cited per function, relative to qingqing-zijin/LEAD-YOLO).
 * Plain pointers and sizes only; all pointers are DEVICE pointers unless stated; `stream` is a
 * hipStream_t (0 = default stream); nothing here synchronises, allocates or frees.
 * Activations are NHWC (N, H, W, C with C fastest) in one of two storage dtypes, selected per call by `dtype`:
 *   LY_F32  (0): fp32 storage, bf16x3 split products on the bf16 matrix cores (~2^-16 per product)
 *   LY_BF16 (1): bf16 storage (BASELINE configs[2]-[4]), plain bf16 products; weights packed with ONE plane
 * Accumulators, BatchNorm statistics, attention tables (ca, rfa, a_h, a_w, pools), parameters and weight gradients are
 * fp32 in both.  Pointers documented as "T*" are `void*` to elements of the call's dtype; leading dimensions are in
 * ELEMENTS.  bf16 calls need channel counts / leading dimensions that are multiples of 8 where fp32 needs 4 (16-byte
 * vectors).  Return value: 0 on success, <0 on error (message via ly_last_error(), thread-local).
 */
#ifndef LEAD_YOLO_HIP_H
#define LEAD_YOLO_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

enum { LY_F32 = 0, LY_BF16 = 1 };
int ly_abi_version(void);      /* 2: dtype-polymorphic entry points */
const char* ly_last_error(void);

/* FasterNet MLPBlock forward, eval form (BN folded to scale/shift):
 *   y = x + W2 . relu(scale * (W1 . [pconv3x3(x[..., :C/4]) | x[..., C/4:]]) + shift)
 * replaces Partial_conv3.forward_split_cat (models/common.py:1432-1437) + MLPBlock.forward
 * (models/common.py:1478-1482).  wp/w1/w2 are frag-packed (lead-yolo_amd/pack.py frag_pack3, planes = 2 for LY_F32, 1 for LY_BF16)
 * from spatial_mixing.partial_conv3.weight ([C/4, 9*ceil4(C/4)], k = tap*ceil4(C/4) + c),
 * mlp.0.weight (rows zero-padded to 16*ly_mlpblock_hidden_tiles(C)) and mlp.3.weight; bn_* have
 * 16*ly_mlpblock_hidden_tiles(C) entries (zero padded).  stats: NULL, or [2 * 16*hidden_tiles] zeroed
 * accumulators = STATISTICS PASS of the train-mode BatchNorm (sum / sum of squares of the pre-BN hidden
 * activations; y and bn_* are ignored, nothing is stored).  x and y must not alias.
 * Built for C in {16,24,40,80,160,320}. */
int ly_mlpblock_fwd(const void* x /*T*/, void* y /*T*/, int n_img, int H, int W, int C, const void* wp, const void* w1,
                    const void* w2, const float* bn_scale, const float* bn_shift, double* stats, int dtype, void* stream);
/* The partial convolution alone: z = [conv3x3(x[:, :C/4]; wp) | x[:, C/4:]] (Partial_conv3.forward_split_cat, models/common.py:1432-1437) in one
 * read + one write of the map — what the training backward needs twice (z for the recomputed hidden tensor; with the transposed-flipped taps
 * applied to the gradient g: [d/dx of the conv | g[C/4:]]).  Persistent patch walk where the MLPBlock's applies (C < 80, W % 16 == 0, >= 1024
 * patches), the one-shot kernel stopped after the partial conv otherwise.  Returns 0 when launched, 1 for a channel count the MLPBlock
 * kernels are not built for (the caller can use a copy + ly_conv3x3_fwd), -1 on error.  x and z dense [n*H*W, C], not aliased.            */
int ly_mlpblock_pconv(const void* x /*T*/, void* z /*T*/, int n_img, int H, int W, int C, const void* wp, int dtype, void* stream);
/* FUSED MLPBlock BACKWARD, bf16 storage (autograd of MLPBlock.forward + Partial_conv3.forward_split_cat, models/common.py:1432-1437, 1478-1482,
 * under train.py:327 `scaler.scale(loss).backward()`; the hidden BatchNorm in train mode).  With z = [pconv3x3(x[:, :C/4]) | x[:, C/4:]],
 * u = W1 z, v = a u + b, h = relu(v), y = x + W2 h  and  dy = d(loss)/dy:
 *   pass 1:  stats[stripe][0:2C] += sum dv, stats[stripe][2C:4C] += sum dv u  with dv = (W2^T dy) [v > 0]   (LY_STATS_STRIPES zeroed double copies:
 *            the `sums` argument of ly_bn_bwd_coeffs, which turns them into dgamma, dbeta and alpha / kappa / lambda)
 *   pass 2:  g = W1^T du  with du = alpha dv + kappa + lambda u  (dense [n*H*W, C]: the gradient with respect to z),
 *            dw1 [2C, C] += sum du (x) z,  dw2 [C, 2C] += sum dy (x) h   (mlp.0.weight / mlp.3.weight gradients, ADDED to; per-block partial tiles go
 *            to `slab` — at least ly_mlpblock_bwd_slab_floats(C) floats, 16-byte aligned — and are folded in block order: bit-reproducible)
 * Nothing 2C wide touches HBM.  wp / w1 as for ly_mlpblock_fwd (planes = 1); w2t = frag-packed mlp.3.weight^T [2C (rows padded to
 * 16*hidden_tiles), C]; w1t = frag-packed mlp.0.weight^T [C, 2C]; a, b (and alpha, kappa, lambda) have 2C entries.  The caller finishes with
 * dx = dy + [pconv^T(g[:, :C/4]) | g[:, C/4:]] (ly_mlpblock_pconv_add) and the partial conv's weight gradient (ly_wgrad on g, x).
 * Built for C in {16, 24, 40, 80} (ly_mlpblock_bwd_ok); x, dy, g dense [n*H*W, C], 16-byte aligned.                                                  */
int ly_mlpblock_bwd_ok(int C, int dtype);
long ly_mlpblock_bwd_slab_floats(int C);
int ly_mlpblock_bwd(const void* x /*T*/, const void* dy /*T*/, void* g /*T*/, int n_img, int H, int W, int C, const void* wp, const void* w1,
                    const void* w2t, const void* w1t, const float* a, const float* b, const float* alpha, const float* kappa, const float* lambda,
                    double* stats, float* slab, long slab_floats, float* dw1, float* dw2, int pass, int dtype, void* stream);
/* The tail of the MLPBlock backward in one launch, bf16: dx = dy + [pconv^T(g[:, :C/4]) | g[:, C/4:]] (the partial conv's data gradient — wpt = the
 * transposed-flipped taps, frag-packed like wp — plus the residual of MLPBlock.forward, models/common.py:1478-1482) and, where built (2-D patches,
 * C/4 <= 32: returns 0), dwp[co * lddw + tap * dw_ts + ci * dw_cs] += sum_p g[p][co] x[p + tap][ci], the gradient of partial_conv3.weight
 * (models/common.py:1412-1437), through `slab` (ly_mlpblock_bwd_slab_floats(C) floats) and a fixed-order combine.  Returns 1 when only dx was
 * produced (the caller then runs ly_wgrad on (g, x)).  g, dy, x, dx dense [n*H*W, C], 16-byte aligned, dx not aliasing g / dy.
 * ly_mlpblock_bwd_dx_ok(C, W, dtype): 1 when the kernel is built for the channel count (16, 24, 40, 80, 160) and its tile fits LDS at map width W. */
int ly_mlpblock_bwd_dx_ok(int C, int W, int dtype);
int ly_mlpblock_bwd_dx(const void* g /*T*/, const void* dy /*T*/, const void* x /*T*/, void* dx /*T*/, int n_img, int H, int W, int C, const void* wpt,
                       float* slab, long slab_floats, float* dwp, int lddw, int dw_ts, int dw_cs, int dtype, void* stream);
/* hidden (2C) channel tiles of 16, padded to an even count */
int ly_mlpblock_hidden_tiles(int C);

/* ---- generic pointwise-convolution GEMM ------------------------------------------------------ */
enum { LY_ACT_NONE_ = 0, LY_ACT_RELU_ = 1, LY_ACT_SILU_ = 2 };          /* `act` values            */
enum { LY_GATHER_ROWS = 0,       /* A row m = a0[m, :k0] | a1[m, :K-k0]                              */
       LY_GATHER_UP2 = 1,        /* a0 is at half resolution: row (n, h/2, w/2)  (nearest 2x upsample) */
       LY_GATHER_PATCH = 2,      /* k x k stride-k patches of an NHWC tensor (PatchMerging)           */
       LY_GATHER_PATCH_NCHW = 3, /* 4 x 4 stride-4 patches of an fp32 NCHW tensor (PatchEmbed on images) */
       LY_GATHER_PATCH_NCHW_U8 = 4, /* the same on a uint8 NCHW image, pixel/255 on load: the `imgs.float() / 255` of the
                                      training loop (train.py:309) folded into the patch gather                      */
       LY_GATHER_PATCH_NCHW_BF16 = 5, /* the same on a bf16 / fp16 NCHW image (a0 8-byte aligned; LY_BF16 calls only): the            */
       LY_GATHER_PATCH_NCHW_F16 = 6   /* `im.half()` batch of a reduced-precision forward (val.py:207) read as it is                  */ };
enum { LY_PRO_NONE = 0,
       LY_PRO_GATE = 1,              /* a0 part: x * g_w[n,w,:] * g_h[n,h,:] (+ res)  (CoordAtt gating) */
       LY_PRO_AFFINE_RELU_CA = 2     /* relu(x*p_scale + p_shift) * p_ca[n,:]         (RFCBAMConv k=1)  */ };

typedef struct LyGemmParams {
  long M;                 /* output pixels = n_img * H * W                                       */
  int H, W;               /* output spatial size                                                 */
  int K, N;               /* contraction length (multiple of 4), output channels                 */
  const void* a0; int lda0; int k0;    /* first source (T): row stride (elements), columns taken  */
  const void* a1; int lda1;            /* second source (T) for columns [k0, K) or NULL           */
  int gather;             /* LY_GATHER_*                                                         */
  int Hin, Win, Cin, ks, pk;           /* patch gathers: input size, kernel=stride, pk = ks*lda0 */
  int pro;                /* LY_PRO_*                                                            */
  const float* g_h; const float* g_w;  /* [n_img, H, k0], [n_img, W, k0]                         */
  const void* res; int ldres;          /* optional residual (T) added to the gated a0 part       */
  const float* p_scale; const float* p_shift; const float* p_ca;   /* [K], [K], [n_img, K]       */
  const void* wp;         /* bf16x3 frag-packed weights of W[N, K] (pack.frag_pack3)             */
  const float* e_scale; const float* e_shift;   /* [N] or NULL (=1 / =0)                         */
  const float* rowscale;  /* [M] or NULL                                                         */
  int act;                /* 0 none, 1 relu, 2 silu                                              */
  void* out; int ldo;     /* output (T), row stride (elements); pointer may be pre-offset into a concat */
  double* stats;          /* NULL, or [LY_STATS_STRIPES = 32][2N] zero-initialised DOUBLE accumulators: STATISTICS PASS for train-mode
                             BatchNorm — adds sum / sum-of-squares over the M rows of the pre-activation value
                             (rowscale*e_scale*acc + e_shift) per output channel; stores nothing when out is NULL,
                             otherwise stores act(that value) as usual (one launch for statistics + pre-BN tensor) */
  int dtype;              /* LY_F32 / LY_BF16: element type T of a0, a1, res, out.  LY_GATHER_PATCH_NCHW(_U8 / _BF16 / _F16) reads an fp32 (uint8 / 16-bit)
                             IMAGE whatever dtype is (dtype then only selects the output element type)             */
  int scat_ks, scat_c;    /* scat_ks > 0 (round 6): the adjoint of the k = s patch gather folded into the store — output column
                             (ky*scat_ks + kx)*scat_c + c of row m = (n, h, w) goes to out[((n*ks*H + ks*h + ky)*ks*W + ks*w + kx)*ldo + c]
                             (N = ks*ks*scat_c, scat_c % 4 == 0; the data gradient of PatchMerging_FasterNet, models/common.py:1553-1561,
                             without the [M][ks*ks*c] intermediate).  0: plain rows                                  */
  const void* eadd; int ldeadd;  /* optional (T, same pixel order as out, row stride ldeadd): added to the value before the store — the gradient
                             another consumer of the same tensor already produced, so autograd's sum of the two (one more read-read-write
                             pass per shared tensor: the backbone / neck skip connections of models/LEAD-YOLO.yaml) is the store itself.
                             Plain-row sources, no prologue, no statistics, N % 4 == 0 (as scat_ks)                    */
} LyGemmParams;

/* out[m, n] = act(rowscale[m] * e_scale[n] * sum_k A'[m, k] W[n, k] + e_shift[n]).
 * Replaces nn.Conv2d(k=1)+BatchNorm2d+SiLU/ReLU (effective Conv, models/common.py:1890-1910), the
 * torch.cat / nn.Upsample feeding it (models/common.py:531-538), CoordAtt's gating multiply
 * (models/common.py:1608), RFCBAMConv's k=1 path (models/rfa.py:113-129) and the FasterNet patch
 * convolutions (models/common.py:1528-1561), depending on gather/pro. */
int ly_gemm_fwd(const LyGemmParams* p, void* stream);


/* ---- 3x3 / s1 / p1 convolution (implicit GEMM) ------------------------------------------------- */
typedef struct LyConv3Params {
  long M; int H, W;        /* output (= input) pixels, spatial size                              */
  int Cin, N;              /* input channels (multiple of 4), output channels                    */
  int TH, TW;              /* pixel patch per block: TH*TW <= 128, (TH+2)*(TW+2) <= 192          */
  const void* x; int ldx;  /* NHWC input (T), row stride (elements)                              */
  const void* wp;          /* frag_pack3(conv_taps_matrix(weight, 32)): k = tap*ceil32(Cin) + c  */
  const float* e_scale; const float* e_shift;   /* folded BN / bias, [N] or NULL                 */
  int act;
  void* out; int ldo;      /* T */
  double* stats;           /* NULL or [32][2N] double accumulators: statistics pass (see LyGemmParams.stats) */
  int dtype;               /* LY_F32 / LY_BF16 */
} LyConv3Params;

/* Conv(c1, c2, 3, 1) = conv3x3(no bias) + BN + SiLU (CA_Bottleneck.cv2, models/common.py:1617,
 * 1890-1910). */
int ly_conv3x3_fwd(const LyConv3Params* p, void* stream);


/* ---- CoordAtt (models/common.py:1583-1609) ------------------------------------------------------- */
/* pool[n, pos, c]: pos < H -> mean over w of row pos; pos >= H -> mean over h of column pos-H.       */
int ly_pool_hw(const void* x /*T*/, int ldx, int n_img, int H, int W, int C, float* pool, int dtype, void* stream);
/* y = h_swish((w1 . pool + b1) * sc + sh); a_h[n,h,:] = sigmoid(wh . y + bh), a_w[n,w,:] = sigmoid(ww . y + bw); w1 [mip, C],
 * wh/ww [C, mip].  sc/sh = bn1 as a per-channel affine (train mode: batch statistics from ly_coordatt_conv1_stats + ly_bn_finalize),
 * or both NULL when bn1 is already folded into w1/b1 (eval).                                                              */
int ly_coordatt_mlp(const float* pool, int n_img, int H, int W, int C, int mip, const float* w1, const float* b1,
                    const float* sc, const float* sh, const float* wh, const float* bh, const float* ww, const float* bw, float* a_h,
                    float* a_w, void* stream);
/* Backward of that MLP in training (models/common.py:1600-1607 under autograd), two launches: given da_h [n,H,C], da_w [n,W,C]
 * (from ly_coordatt_gate_bwd), the forward's pool, batch mean / invstd and a_h / a_w:
 *   dpool [n, H+W, C] is WRITTEN; dw1 [mip,C], dgamma, dbeta [mip], dwh/dww [C,mip], dbh/dbw [C] are ADDED TO (they may be the
 *   parameters' persistent gradient storage; fresh buffers must be zeroed by the caller); d/d conv1.bias is identically zero
 *   (bn1 removes the batch mean) and is not produced;
 *   ws: scratch [n*(H+W)][3*mip] floats, sums: [32][2*mip] zeroed by the caller.  Built for mip 8 / 16, C <= 512.            */
int ly_coordatt_mlp_bwd(const float* pool, int n_img, int H, int W, int C, int mip, const float* w1, const float* b1,
                        const float* mean, const float* invstd, const float* gamma, const float* beta, const float* wh,
                        const float* ww, const float* a_h, const float* a_w, const float* da_h, const float* da_w, float* ws,
                        double* sums /* [32][2 mip] doubles, zeroed */, float* dpool, float* dw1, float* dgamma, float* dbeta, float* dwh, float* dbh, float* dww,
                        float* dbw, int grads_f64 /* != 0: dw1 .. dbw are zeroed DOUBLE scratches of the same shapes, see ly_f64_add */, void* stream);

/* out = x * a_w[n,w,:] * a_h[n,h,:] (+ res): the gating multiply as a standalone pass (only used
 * when no consumer GEMM can absorb it, e.g. CA_Bottleneck with a residual shortcut).               */
int ly_coordatt_gate(const void* x /*T*/, int ldx, int n_img, int H, int W, int C, const float* a_h, const float* a_w,
                     const void* res /*T*/, int ldres, void* out /*T*/, int ldo, int dtype, void* stream);

/* ---- RFCBAMConv (models/rfa.py:77-129) ----------------------------------------------------------- */
/* rfa.SE: ca[n, :] = sigmoid(wb . relu(wa . mean_hw(x)));  wa [R, C], wb [C, R]; part = workspace
 * of n_img * slices * C floats.                                                                     */
int ly_se_fwd(const void* x /*T*/, int ldx, int n_img, int HW, int C, const float* wa, const float* wb, int R, float* part,
              int slices, float* ca, int dtype, void* stream);
/* mm[n, y, x, 0:2] = (max_c, mean_c) of relu(bn(generate(x))) on the k-times expanded grid.
 * k = 1: a1/b1 = folded per-channel scale/shift.  k = 3: wg = pack.rfcbam_gen_weights(..., 32, False):
 * [C32/32][4 waves][9 t][4 channel pairs][20]; TH x TW (<= 64) = output-pixel tile per block.
 * part != NULL: the SAME pass also leaves the partial sums of SE's global average pool (models/rfa.py:90), part[n][slice][C]
 * (k = 1: slices = pixel slices per image, free; k = 3: one row per output tile, slices = ceil(Ho/TH)*ceil(Wo/TW)) for
 * ly_rfcbam_mid / ly_se_mlp — x is then read once instead of twice before the contraction.                              */
int ly_rfcbam_stats(const void* x /*T*/, int ldx, int n_img, int H, int W, int C, int k, int s, const float* wg,
                    const float* a1, const float* b1, int TH, int TW, float* mm, float* part, int slices, int dtype, void* stream);
/* SE pooling partials alone (the first half of ly_se_fwd): part[n][slice][C] = sum of x over the pixels of slice `slice`.          */
int ly_colsum(const void* x /*T*/, int ldx, int n_img, int HW, int C, float* part, int slices, int dtype, void* stream);
/* SE's two linears + sigmoid from the pooling partials (-> ca [n_img, C]) and get_weight's 3x3 conv + sigmoid on the [max, mean]
 * map (-> rfa [n_img, HK, WK]) as ONE launch (two independent groups of blocks): the steps between ly_rfcbam_stats and the main
 * contraction (models/rfa.py:88-92, 107, 127).  HW = pixels per image of the pooled input.                                    */
int ly_rfcbam_mid(const float* part, int slices, int C, int HW, const float* wa, const float* wb, int R, float* ca, int n_img,
                  const float* mm, int HK, int WK, const float* w18, float* rfa, void* stream);
/* rfa[n, y, x] = sigmoid(conv3x3_pad1(mm; w[2][3][3]))   (get_weight, models/rfa.py:107)             */
int ly_rfa_map(const float* mm, int n_img, int HK, int WK, const float* w, float* rfa, void* stream);

typedef struct LyRfcbam3Params {
  int n_img, H, W, C;          /* input NHWC                                                      */
  int Ho, Wo, N, s;            /* output size, output channels, stride                            */
  int TH, TW;                  /* output-pixel tile per block (TH*TW <= 64)                       */
  const void* x; int ldx;      /* T */
  const float* wg;             /* pack.rfcbam_gen_weights(..., 16, True): [C/16][4][9 t][2][20]   */
  const float* ca;             /* [n_img, C]                                                      */
  const float* rfa;            /* [n_img, 3Ho, 3Wo]                                               */
  const void* wp;              /* frag_pack3(conv.0.weight as [N, C/16, 144 -> 160 zero padded])  */
  const float* e_scale; const float* e_shift;   /* conv.1 BN folded with conv.0.bias              */
  void* out; int ldo;          /* T */
  double* stats;               /* NULL or [32][2N] double accumulators: conv.1 BatchNorm statistics pass (sums of e_scale*acc + e_shift);
                                  with out != NULL that pre-BN value is stored as well (one contraction in training)  */
  int linear;                  /* != 0: store the affine value without the ReLU (backward recompute) */
  int dtype;                   /* LY_F32 / LY_BF16 */
} LyRfcbam3Params;
/* RFCBAMConv kernel_size 3 main contraction (+ReLU); the k=1 case is ly_gemm_fwd with
 * LY_PRO_AFFINE_RELU_CA and rowscale = rfa.                                                         */
int ly_rfcbam3_fwd(const LyRfcbam3Params* p, void* stream);

/* ---- RFCBAMConv kernel_size 3 on the lane = channel core (csrc/ly_rf3c.hpp; models/rfa.py:113-129) ----------------------------
 * C % 32 == 0, stride 1 or 2, tiles TH x TW with TW even, TH*TW <= 64 and (s(TH-1)+3)(s(TW-1)+3) <= 320 input positions.
 * wq = generate weights in lane order: per channel 100 floats — w[t][u] at i = t*9+u, b[t] at i = 81+t, a[t] at i = 90+t, 1 of padding —
 *      stored as float [C/32][25][32][4]: element i of channel c at ((c/32*25 + i/4)*32 + c%32)*4 + i%4.  raw == 0 (inference, folded):
 *      w = generate.0.weight * bn_scale, b = bn_shift, v = b + sum w x.  raw != 0 (training): w = generate.0.weight, a = bn_scale,
 *      b = bn_shift, v = a*(sum w x) + b — the form the backward kernels re-evaluate bit for bit (pack.rfcbam_gen_weights_c /
 *      ly_rfcbam_gen_prepare).
 * ly_rf3c_stats: ONE pass over x leaves mm[n, 3Ho, 3Wo, 2] = [max_c, mean_c] of relu(bn(generate(x))) (models/rfa.py:125-126) and, if
 *      part != NULL, the SE global-average-pool partials part[n][tile][C] (models/rfa.py:90; slices must equal the tile count).
 * ly_rf3c_fwd: the main contraction (models/rfa.py:124, 128-129); p as for ly_rfcbam3_fwd except that p->wg is ignored and
 *      p->wp = conv.0.weight frag-packed as [N][C/32 chunks][9 taps][32 channels] (K = 9*C).                                       */
int ly_rf3c_stats(const void* x /*T*/, int ldx, int n_img, int H, int W, int C, int s, const float* wq, int raw, int TH, int TW, float* mm,
                  float* part, int slices, int dtype, void* stream);
int ly_rf3c_fwd(const LyRfcbam3Params* p, const float* wq, int raw, void* stream);

/* ---- RFCBAMConv kernel_size 3 with `generate` on the matrix cores (csrc/ly_rf3m.hip; models/rfa.py:113-129) --------------------------
 * bf16 storage, inference form (BatchNorm folded), C % 32 == 0, stride 1 or 2, tiles TH x TW of at most 32 output pixels that read at most
 * 160 input positions.  The depthwise 3x3 `generate` is a block-diagonal v_mfma_f32_32x32x16_bf16 product whose accumulator layout is the
 * B operand of the main contraction: relu, * ca * rfa and the bf16 conversion happen in registers.
 * wst / p->wp = pack.rf3m_stream(...): every A fragment in consumption order (the header of csrc/ly_rf3m.hip and pack.py give the index map).
 * ly_rf3m_stats: mm[n, 3Ho, 3Wo, 2] = [max_c, mean_c] of relu(bn(generate(x))) (models/rfa.py:125-126) and, if part != NULL, the SE
 *      global-average-pool partials part[n][tile][C] (models/rfa.py:90; slices = tile count) from ONE pass over x.
 * ly_rf3m_fwd:  out = relu(bn(conv_{3x3, stride 3}(G * ca * rfa) + bias)) (models/rfa.py:124, 128-129); p as for ly_rfcbam3_fwd with
 *      dtype LY_BF16, stats == NULL, linear == 0, N % 64 == 0, p->wg ignored.                                                          */
int ly_rf3m_stats(const void* x /*bf16*/, int ldx, int n_img, int H, int W, int C, int s, const void* wst, int TH, int TW, float* mm, float* part,
                  int slices, void* stream);
int ly_rf3m_fwd(const LyRfcbam3Params* p, void* stream);

/* RFCBAMConv kernel_size 3 backward without 9x-sized tensors (csrc/ly_rf3c_bwd.hip; autograd of models/rfa.py:113-129): every pass
 * re-derives dcd = du . Wc^T (MFMA) and generate / BatchNorm / ReLU (VALU, bit-identical to the training forward, raw wq) on chip from
 * x and du.  bf16 storage, stride 2, C % 32 == 0, tiles as ly_rf3c_fwd.
 *   ly_rf3c_bwd pass 0 (A): d_rfa_part [C/32][n, 3Ho, 3Wo] (one slab per channel chunk: sum them), d_ca [n, C]        (O in {64, 128, 256})
 *               pass 1 (B): sums [n_img stripes][2][9C] in [t*C + c] order: sum dv, sum dv*u  (-> ly_bn_bwd_coeffs with n_img stripes)
 *               pass 2 (C): dwg [n_img][C*81] rows of d(generate.0.weight) (sum them), dx [n, H, W] rows of T (+ dgap[n, C] * dgap_scale)
 *   ly_rf3c_wgrad: dwc_part [ng][O][9][C] slabs of d(conv.0.weight) in (o, tap, c) order (sum them); image i goes to slab i % ng.        */
typedef struct LyRf3cBwdParams {
  int n_img, H, W, C;
  int Ho, Wo, O, s;
  int TH, TW;
  const void* x; int ldx;        /* T: the saved input */
  const void* du; int lddu;      /* T [n*Ho*Wo][O]: gradient of the conv's pre-BatchNorm output */
  const float* wq;               /* raw lane-order generate weights (see ly_rf3c_stats) */
  const void* wct;               /* conv.0.weight^T frag-packed (one plane): rows t*C + c, K = O */
  const float* ca;               /* [n, C] */
  const float* rfa;              /* [n, 3Ho, 3Wo] */
  const float* mm;               /* [n, 3Ho, 3Wo, 2] the forward's [max_c, mean_c] map */
  const float* d_mm;             /* [n, 3Ho, 3Wo, 2] its gradient (passes B, C) */
  const float* coef;             /* [3][9C]: alpha, kappa, lambda of ly_bn_bwd_coeffs in [t*C + c] order (pass C) */
  float* d_rfa_part; float* d_ca;
  float* sums;
  float* dwg;
  void* dx; int lddx;            /* T */
  const float* dgap; float dgap_scale;
  float* dwc_part; int ng;
  int dtype;
} LyRf3cBwdParams;
int ly_rf3c_bwd(const LyRf3cBwdParams* p, int pass, void* stream);
int ly_rf3c_wgrad(const LyRf3cBwdParams* p, void* stream);


/* ---- graph remainder ------------------------------------------------------------------------- */
/* SPPF pooling (models/common.py:348-366): out[n, p, :] = [x | m(x) | m(m(x)) | m(m(m(x)))] with m =
 * k x k / stride 1 / pad k//2 max pool; out row stride ldo >= 4C.  The map must fit LDS
 * (2*H*W*4 floats <= 160 KiB); C, ldx, ldo multiples of 4.                                                                    */
int ly_sppf_pool(const void* x /*T*/, int ldx, int n_img, int H, int W, int C, int k, void* out /*T*/, int ldo, int dtype, void* stream);
/* Detect tail (models/yolo.py:95-120): y[n,h,w,na*no] (row stride ldy) -> p[n,na,h,w,no] and, if z is
 * not NULL, decoded rows z[n, zoff + (a*H + h)*W + w, :] of a [n_img, zrows, no] tensor;
 * anchors = [na,2] in grid units (Detect.anchors[i]), stride = Detect.stride[i].                     */
int ly_detect_tail(const void* y /*T*/, int ldy, int n_img, int H, int W, int na, int no, const float* anchors, float stride,
                   float* p, float* z, long zrows, long zoff, int dtype, void* stream);
/* Detect level in ONE launch (eval; csrc/ly_detect.hip): the head's 1x1 convolution x[n*H*W, K] (row stride ldx, T) -> na*no <= 32 channels
 * + bias and the ly_detect_tail arithmetic on the fp32 accumulators.  wp: the head weight [na*no, K] as two 16-row tiles in NATURAL k
 * order — uint4 wp[((t*S + s)*PL + plane)*64 + lane], lane = g*16 + i holds row 16t + i, k = 32s + 8g + 0..7 (nat != 0: pack.frag_pack_nat)
 * — or, nat == 0, in the layout of every other contraction (ly_frag_pack3 with 32 rows: the training step's packs are used as they are);
 * PL = 2 hi / lo planes for LY_F32, 1 for LY_BF16; S = K/32 in {2, 4, 8} (16 in bf16).  p / z / zrows / zoff as ly_detect_tail (z may be
 * NULL: training, where the loss reads the raw maps).                                                                                   */
int ly_detect_level(const void* x /*T*/, int ldx, int n_img, int H, int W, int K, const void* wp, int nat, const float* bias, int na, int no,
                    const float* anchors, float stride, float* p, float* z, long zrows, long zoff, int dtype, void* stream);
int ly_detect_level_ok(int K, int na, int no, int dtype);     /* 1 when ly_detect_level is built for this shape */
/* Adjoint of that permute for the training step (models/yolo.py:88): dp fp32 [n, na, H, W, no] -> du rows [n*H*W][ldu] of T (column a*no+o;
 * columns >= na*no written as zero: the operand of the head's dgrad / wgrad), dbias[a*no+o] += sum over pixels.  W <= 160, na*no <= ldu <= 32. */
int ly_detect_head_bwd(const float* dp, int n_img, int H, int W, int na, int no, void* du /*T*/, int ldu, float* dbias,
                       int dbias_f64 /* != 0: dbias is a zeroed DOUBLE scratch, see ly_f64_add */, int dtype, void* stream);


/* ---- train-mode BatchNorm statistics passes ----------------------------------------------------- */
/* mom[c] += sum_rows x[r, c], mom[C + c] += sum_rows x[r, c]^2 over an [rows, C] row matrix (mom zeroed
 * by the caller).  Used for the k=1 `generate` BatchNorm of RFCBAMConv (models/rfa.py:101-106).       */
int ly_chan_moments(const void* x /*T*/, int ldx, long rows, int C, double* mom /* [32][2C] doubles, zeroed */, int dtype, void* stream);
/* CoordAtt bn1 (models/common.py:1589,1602): sum / sum of squares of conv1(pool) + bias over all
 * n*(H+W) positions, stats[0:mip] and stats[mip:2mip] (zeroed by the caller).                        */
int ly_coordatt_conv1_stats(const float* pool, long positions, int C, int mip, const float* w1, const float* b1,
                            double* stats /* [2 mip] doubles */, void* stream);
/* dst[c] = sum over r < R of src[r][c]: the stripes of a DOUBLE statistics accumulator folded in index order (ly_chan_moments) */
int ly_sum_rows_f64(const double* src, int R, int C, double* dst, void* stream);

/* RFCBAMConv k=3 `generate` BatchNorm statistics (train mode): mom[54][C] (zeroed by the caller) receives,
 * per input channel, the 9 first moments sum x_u and the 45 second moments sum x_u x_v (u <= v, row-major
 * upper triangle) of the zero-padded stride-s 3x3 taps over all n_img*Ho*Wo output pixels; mean and
 * variance of every (channel, tap-output) follow as w.m and w^T M w on the host.                        */
int ly_rfcbam_tap_moments(const void* x /*T*/, int ldx, int n_img, int H, int W, int C, int s, double* mom /* [54][C] doubles, zeroed */, int dtype, void* stream);
/* Train-mode `generate` BatchNorm of RFCBAMConv (models/rfa.py:101-106) from those moments (k = 3: mom [54][C]; k = 1: ly_chan_moments'
 * [2][C]) and generate.0.weight, ONE launch: batch statistics of every generate channel g = c*k*k + t, running statistics and
 * num_batches_tracked updated as nn.BatchNorm2d does (NULL = not tracked);
 *   out8 [8][C*k*k]: scale, shift, mean, invstd in [c*k*k + t] order, then the same four in [t*C + c] order (the backward kernels' order);
 *   k = 3: wq_stats / wq_main = the folded weights (w*scale | shift) in the LDS orders of ly_rfcbam_stats / ly_rfcbam3_fwd
 *          ([ceil(C/32)*32 | ceil(C/16)*16][9][10] floats each);  k = 1: a1[c] = w[c]*scale[c].                                  */
int ly_rfcbam_gen_prepare(const double* mom, int C, int k, const float* gen_w, const float* gamma, const float* beta, float eps,
                          float momentum, double count, float* running_mean, float* running_var, long* nbt, float* out8, float* a1,
                          float* wq_stats, float* wq_main, float* wq_c /* NULL or [C*100]: the RAW lane-order image of ly_rf3c_* */, int mom_stripes /* 1; k = 1: the striped [mom_stripes][2C] array of ly_chan_moments, folded here */, void* stream);

/* ---- eval tail: non_max_suppression on the device (utils/general.py:884-994; detect.py:149, val.py:230-234) --------------------------
 * pred [bs, N, no = 5 + nc] fp32 (xywh, obj, class confidences: Detect's inference output).
 * ly_nms_candidates: score[b, i] = obj * best class confidence when obj > conf_thres, that product > conf_thres and the class is allowed
 *   (class_mask bit c, 0 = every class), else -1;  det[b, i] = (x1, y1, x2, y2, conf, cls).
 * The caller sorts score per image, descending and stable, and passes the sorted values + permutation to
 * ly_nms_greedy: greedy NMS over the sorted candidates (IoU > iou_thres with a kept box drops a box; boxes are offset by cls * max_wh,
 *   max_wh = 0 for class-agnostic NMS), at most max_nms candidates, at most max_det kept: keep[b, 0:count[b]] = indices into N.        */
int ly_nms_candidates(const float* pred, int bs, int N, int no, float conf_thres, unsigned long long class_mask, float* score, float* det,
                      void* stream);
/* multi_label form (utils/general.py:921, 951-955): score / det have bs * N * nc entries, one per (box, class) pair in the reference's row order
 * box * nc + class; the sort and ly_nms_greedy then run over N * nc candidates per image.                                                   */
int ly_nms_candidates_ml(const float* pred, int bs, int N, int no, float conf_thres, unsigned long long class_mask, float* score, float* det,
                         void* stream);
int ly_nms_greedy(const float* det, const long* order, const float* sorted_score, int bs, int N, float iou_thres, float max_wh, int max_det,
                  int max_nms, int* keep, int* count, void* stream);

/* ---- backward building blocks of the training step (train.py:324 `scaler.scale(loss).backward()`) -------------
 * Data gradients of 1x1 / 3x3 stride-1 convolutions reuse ly_gemm_fwd / ly_conv3x3_fwd with transposed weights.   */

/* BatchNorm(train)+activation backward over an [rows, C] matrix.  Forward was v = a[c]*u + b[c], y = act(v)
 * (models/common.py:1906-1907 Conv.forward; :1478-1482 MLPBlock).  With dv = dy * act'(v):
 *   reduce: sums[c] += sum_r dv, sums[C + c] += sum_r dv*u   (sums zeroed by the caller)
 *   apply : du = alpha[c]*dv + kappa[c] + lambda[c]*u        (du may alias u or dy)                               */
/* y = act(a[c]*u + b[c]) over [rows, C]: the normalisation + activation half of a train-mode unit whose contraction pass
 * (stats != NULL AND out != NULL: statistics accumulated and the pre-BN value stored in ONE launch) produced u.            */
int ly_bnact_fwd(const void* u /*T*/, int ldu, long rows, int C, const float* a, const float* b, int act, void* y /*T*/, int ldy, int dtype, void* stream);
int ly_bnact_bwd_reduce(const void* dy /*T*/, int lddy, const void* u /*T*/, int ldu, long rows, int C, const float* a, const float* b,
                        int act, double* sums /* [32][2C] doubles, zeroed */, int dtype, void* stream);
int ly_bnact_bwd_apply(const void* dy /*T*/, int lddy, const void* u /*T*/, int ldu, long rows, int C, const float* a, const float* b,
                       int act, const float* alpha, const float* kappa, const float* lambda, void* du /*T*/, int lddu, int dtype, void* stream);
/* Pair forms: TWO conv -> BN(train) -> act units over one stacked pre-activation tensor u [rows, C] — channels [0, csplit) and [csplit, C): C3_CA's
 * cv1 | cv2 over their shared input (models/common.py:1630-1636) — in ONE pass each: unit 1's gradient dy1 [rows, csplit] (row stride lddy1), unit 2's
 * dy2 [rows, C - csplit]; sums1 / sums2 are the striped accumulators of the two units (as `sums` of ly_bnact_bwd_reduce), a / b / alpha / kappa /
 * lambda hold both units' C entries side by side.                                                                                              */
int ly_bnact_bwd_reduce_pair(const void* dy1 /*T*/, int lddy1, const void* dy2 /*T*/, int lddy2, int csplit, const void* u /*T*/, int ldu, long rows, int C,
                             const float* a, const float* b, int act, double* sums1, double* sums2, int dtype, void* stream);
int ly_bnact_bwd_apply_pair(const void* dy1 /*T*/, int lddy1, const void* dy2 /*T*/, int lddy2, int csplit, const void* u /*T*/, int ldu, long rows, int C,
                            const float* a, const float* b, int act, const float* alpha, const float* kappa, const float* lambda, void* du /*T*/, int lddu,
                            int dtype, void* stream);

/* Weight gradient of a convolution whose forward read input pixel (ho*stride + ky - pad, wo*stride + kx - pad):
 *   dw[n][(ky*ks + kx)*Cin + c] += sum_{m = (img, ho, wo)} du[m][n] * x[img, hi, wi, c]      (zero outside the map)
 * x is an NHWC map (row stride ldx) of Hin x Win pixels, or an NCHW image when nchw != 0, or — when up2 != 0 — a
 * half-resolution map read through nearest-2x upsampling (Hin, Win are the SOURCE sizes).  dw rows have stride lddw
 * and must be zeroed by the caller (partial sums of pixel chunks are added atomically).                            */
typedef struct LyWgradParams {
  long M;            /* output pixels = n_img * H * W */
  int H, W;          /* output map */
  int N;             /* output channels */
  const void* du; int lddu;   /* T */
  const void* x; int ldx;     /* T */
  int Hin, Win, Cin;
  int ks, stride, pad;
  int nchw, up2;
  float* dw; int lddw;        /* fp32 */
  int dtype;                  /* LY_F32 / LY_BF16 */
  /* layout of a dw row: entry (tap, c) lives at dw[n*lddw + tap*dw_ts + c*dw_cs] (packed [tap][c]: dw_ts = Cin, dw_cs = 1; the
   * torch weight layout [cout][cin][kh][kw]: dw_ts = 1, dw_cs = ks*ks, lddw = cin*ks*ks), and only rows n < n_valid / channels
   * c < c_valid are written (padded contractions) — so the kernel can add straight into a parameter's .grad storage          */
  int dw_ts, dw_cs, n_valid, c_valid;
  /* optional prologue on x, plain-row 1x1 problems of the tiled path only (both NULL: x as stored):
   *   x'[p][c] = max(x[p][c]*x_scale[c] + x_shift[c], 0)
   * — a BatchNorm + ReLU whose output the producer never materialised (the MLPBlock's hidden tensor, models/common.py:1478-1482,
   * in the training backward: dW2 = dy^T . relu(BN(u1)) is contracted from u1 directly)                                          */
  const float* x_scale; const float* x_shift;
  /* optional scratch for the partial tiles of the pixel chunks (fp32, ws_floats elements, contents irrelevant): kernels that can use it
   * store every chunk's tile with plain stores and fold the chunks in a FIXED order in a second launch — no float atomics (~1.3 TB/s
   * chip-wide, half the time of a 3x3 weight gradient) and a bit-reproducible dw.  NULL or too small: atomic accumulation.          */
  float* ws; long ws_floats;
} LyWgradParams;
int ly_wgrad(const LyWgradParams* p, void* stream);
/* n independent weight gradients in ONE launch when all of them are plain-row 1x1 problems of the 128 x 128 tile class (N > 64, no
 * gather) and n <= 4: they share the launch's ~512 blocks, so every block walks a longer pixel chunk (fewer first-load waits and atomic tile
 * flushes per pixel).  Any other mix: the same as n calls of ly_wgrad.                                                                  */
int ly_wgrad_group(const LyWgradParams* arr, int n, void* stream);
/* development switch (returns the previous value): 0 sends 3x3 / stride-1 bf16 problems to the generic tiled kernel instead of the halo-tile
 * kernel of csrc/ly_wgrad3.hip (A/B timing and the equivalence test); default 1.                                                           */
int ly_tune_wgrad3(int on);

/* Adjoint of the nearest-2x read: out[n,h,w,:] = sum of d[n, 2h+{0,1}, 2w+{0,1}, :]  (d is a 2Hs x 2Ws map).      */
int ly_up2_bwd(const void* d /*T*/, int ldd, int n_img, int Hs, int Ws, int C, void* out /*T*/, int ldo, int dtype, void* stream);
/* Adjoint of the k = s patch gather: g[m][(ky,kx,c)] -> dx[n, ks*ho+ky, ks*wo+kx, c] (dense NHWC, C % 4 == 0).    */
int ly_unpatch(const void* g /*T*/, int n_img, int Ho, int Wo, int C, int ks, void* dx /*T*/, int dtype, void* stream);
/* dst[c] (+)= sum over r < R of src[r*ld + c], fp32, fixed summation order (no atomics): folds the partial-row buffers the backward kernels
 * leave (autograd's reduction of per-block parameter-gradient partials, reference models/rfa.py:113-129 backward).                        */
int ly_sum_rows(const float* src, long R, long C, long ld, float* dst, int accumulate, void* stream);
/* Space-to-depth of the uint8 NCHW image [n, C, H, W] (H, W multiples of 4) for PatchEmbed's weight gradient (models/common.py:1537-1550,
 * the `imgs` of train.py:309 before `.float() / 255`): rows[m][c*16 + ky*4 + kx] = (T)img[n, c, 4*ho+ky, 4*wo+kx], m = (n*Ho + ho)*Wo + wo.
 * Integer values (exact in bf16); the caller scales the weight gradient by 1/255.                                                          */
int ly_patch4_rows_u8(const unsigned char* img, int n_img, int C, int H, int W, void* rows /*T [n*H/4*W/4][16*C]*/, int dtype, void* stream);
/* PatchEmbed's weight gradient straight from the uint8 image (round 6; the same reference lines: PatchEmbed_FasterNet's Conv2d(3, N, 4, 4) under
 * autograd, models/common.py:1537-1550, train.py:327): dw[n][c*16 + ky*4 + kx] += scale * sum_m du[m][n] * img[n_img(m), c, 4*ho+ky, 4*wo+kx],
 * m = (n_img*Ho + ho)*Wo + wo — no space-to-depth copy.  bf16 du [M][lddu] (dtype = LY_BF16), C = 3, N a multiple of 8 up to 64; blocks leave
 * [N][48] partials in `slab` (at least 1536*N*48 floats, or units*N*48 for M/128 < 1536 units) which are folded in block order
 * (ly_sum_rows): no atomics.  scale = 1/255 for the `imgs / 255` of train.py:309.                                                           */
int ly_patch4_wgrad_u8(const unsigned char* img, int n_img, int C, int H, int W, const void* du /*bf16*/, int lddu, int N, float scale, float* slab,
                       long slab_floats, float* dw /*[N][16*C], += */, int dtype, void* stream);

/* CoordAtt backward (models/common.py:1595-1609).  Gate out = x*a_h[n,h,:]*a_w[n,w,:]:
 *   dx = dout*a_h*a_w,  da_h[n,h,c] = sum_w dout*x*a_w,  da_w[n,w,c] = sum_h dout*x*a_h — written as PARTIAL sums, one plain store per
 *   element and block: da_h [slabs][n][H][C] (slabs = ceil(W / (8 * (256 / (C/4))))), da_w [bands][n][W][C] (bands = ceil(H / 8)); the caller
 *   passes the two counts (checked) and folds the partials in index order (ly_sum_rows): no atomics, the same bits in every run.
 * Pools pool[n,0:H]=mean_w x, pool[n,H:H+W]=mean_h x:  dx[n,h,w,c] (+)= gp[n,h,c]/W + gp[n,H+w,c]/H  (accumulate != 0: added to dx). */
int ly_coordatt_gate_bwd(const void* dout /*T*/, int ldd, const void* x /*T*/, int ldx, int n_img, int H, int W, int C, const float* a_h,
                         const float* a_w, void* dx /*T*/, int lddx, float* da_h, float* da_w, int bands, int slabs, int dtype, void* stream);
int ly_pool_hw_bwd(const float* gp, int n_img, int H, int W, int C, void* dx /*T*/, int lddx, int accumulate, int dtype, void* stream);
/* SPPF backward as a gather (no atomics, deterministic): ly_maxpool_arg stores, for every window of a k x k / s1 / pad k//2 max-pool over
 * x [n, H, W, C] (row stride ldx), the tap index (0 .. k*k-1, row-major) of its first maximum — ATen's routing rule — as one byte per element
 * (arg: row stride lda);  ly_maxpool_gather then computes  out[p] = d_own[p] + sum over the windows q containing p of
 * [arg(q) == tap of p in q] * d_up[q]  (d_up: fp32 total gradient of the pooled map; d_own: T-typed direct gradient of the pool's input).
 * Built for k = 5 (SPPF).                                                                                                             */
int ly_maxpool_arg(const void* x /*T*/, int ldx, int n_img, int H, int W, int C, int k, unsigned char* arg, int lda, int dtype, void* stream);
int ly_maxpool_gather(const unsigned char* arg, int lda, const float* d_up, int ldu, const void* d_own, int ldd, int own_dtype, int n_img, int H,
                      int W, int C, int k, void* out, int ldo, int out_dtype, void* stream);
/* The same backward in ONE launch when the map fits LDS (H*W*104 bytes <= 150 KB, k = 5, c % 8 == 0): a block owns one image x 8 channels, the three
 * levels run from LDS with the routing rule and summation order of the two entries above (bit-identical).  buf = [y | m(y) | m(m(y)) | m(m(m(y)))]
 * rows (ldb), d = its gradient (ldd), out = d/dy (ldo), all T.  Returns 1 without launching when the shape does not fit.                        */
int ly_sppf_bwd(const void* buf /*T*/, int ldb, const void* d /*T*/, int ldd, int n_img, int H, int W, int c, int k, void* out /*T*/, int ldo, int dtype,
                void* stream);
/* MLPBlock backward, last step (models/common.py:1478-1482 under autograd): dx[r, c] = dy[r, c] + (c < c4 ? t[r, c] : g[r, c]) over dense
 * [rows, C] matrices (t: row stride ldt >= ceil4(c4)) — the residual + the 1x1's gradient, with the partial 3x3 conv's gradient in its channels. */
int ly_mlp_dx(const void* dy /*T*/, const void* g /*T*/, const void* t /*T*/, int ldt, long rows, int C, int c4, void* dx /*T*/, int dtype, void* stream);
/* k x k / stride 1 / pad k//2 max-pool backward (SPPF, models/common.py:348-366): dx[argmax of window] += dy
 * (first maximum in row-major order, as ATen); dx is ACCUMULATED into, which lets the three chained pools add
 * into the gradient slots of the SPPF concat buffer in place.  x is T; dy and dx are ALWAYS fp32 (atomic accumulation).   */
int ly_maxpool_bwd(const void* x /*T*/, int ldx, const float* dy, int lddy, int n_img, int H, int W, int C, int k, float* dx, int lddx,
                   int dtype, void* stream);

/* ---- RFCBAMConv backward (models/rfa.py:113-129 under autograd) ----------------------------------------------------
 * Expanded tensors are [m = (n, ho, wo)][t = tap][c] (channel contiguous), maps are [n, k*Ho, k*Wo]; per-(t, c) vectors
 * (ag, bg, alpha, kappa, lambda, sums) are indexed t*C + c.  See csrc/ly_rfcbam_bwd.hip for the step list.            */
/* ug[m][t][c] = sum_u wg[c*KK + t][u] * x_u(m)[c]  (depthwise `generate` conv before its BatchNorm)                    */
int ly_rf_generate(const void* x /*T*/, int ldx, int n_img, int H, int W, int C, int k, int s, const float* wg, void* ug /*T*/, int dtype, void* stream);
/* G = relu(ag*ug + bg):  cd = G*ca*rfa (written),  d_rfa[pos] += sum_c dcd*G*ca,  gmax[pos] = max_c G (as float bits,
 * zeroed by the caller),  d_ca[n][c] += sum_{m,t} dcd*rfa*G  (d_rfa, d_ca zeroed by the caller)                        */
int ly_rf_bwd_attn(int n_img, int H, int W, int C, int k, int s, const void* ug /*T*/, const void* dcd /*T*/, const float* ag, const float* bg,
                   const float* ca, const float* rfa, void* cd /*T*/, float* d_rfa, float* gmax, float* d_ca, int dtype, void* stream);
/* rfa = sigmoid(conv3x3(mm; w18)) backward: d_mm[n,y,x,2] (written) and dw18[18] (+=, zeroed by the caller)            */
int ly_rfa_bwd(const float* d_rfa, const float* rfa, const float* mm, const float* w18, int n_img, int Hk, int Wk, float* d_mm,
               float* dw18, int dw18_f64 /* != 0: dw18 is a zeroed DOUBLE scratch [18], see ly_f64_add */, void* stream);
/* dv = [G>0]*(dcd*rfa*ca + d_mm.mean/C + [G==gmax]*d_mm.max) written over dcd; sums[t*C+c] += dv, sums[C*KK + ..] += dv*ug */
int ly_rf_bwd_relu(int n_img, int H, int W, int C, int k, int s, const void* ug /*T*/, void* dcd /*T*/, const float* ag, const float* bg,
                   const float* ca, const float* rfa, const float* gmax, const float* d_mm, float* sums, int dtype, void* stream);
/* dug = alpha*dv + kappa + lambda*ug written over dv; dwg[row][c*KK + t][u] = partial sums of sum_m dug*x_u: a
 * [part_rows][C*KK][KK] matrix zeroed by the caller; every (block, pixel sub-group) stores one row, the caller sums rows */
int ly_rf_bwd_gen(const void* x /*T*/, int ldx, int n_img, int H, int W, int C, int k, int s, const void* ug /*T*/, void* dv /*T*/, const float* alpha,
                  const float* kappa, const float* lambda, float* dwg, int part_rows, int dtype, void* stream);
/* dx[n,hi,wi,c] = sum over (m, u) reading that input pixel of sum_t dug[m][t][c]*wg[c*KK + t][u]  (written)           */
int ly_rf_bwd_dx(int n_img, int H, int W, int C, int k, int s, const void* dug /*T*/, const float* wg, void* dx /*T*/, int lddx,
                 const float* addnc /* NULL, or [n_img, C]: dx += addnc[n, c] * add_scale (the SE pooling's gradient) */, float add_scale,
                 int dtype, void* stream);
/* SE backward in training (models/rfa.py:88-92), two launches: from the forward's pooling partials part[n][slices][C] (HW pixels per
 * image), ca = sigmoid(wb relu(wa mean)) and d_ca: dwa [R, C], dwb [C, R] are ADDED TO (one thread per weight sums the per-image outer
 * products kept in ws: no atomics, deterministic), dgap [n_img, C] = d/d(mean x) is written.  ws: scratch, n_img * (2C + 2R) floats. */
int ly_se_bwd(const float* part, int slices, int n_img, int HW, int C, const float* wa, const float* wb, int R, const float* ca,
              const void* d_ca /* floats, or the double accumulators of ly_rf1_bwd / ly_rf3c_bwd if d_ca_f64 */, int d_ca_f64, float* dwa, float* dwb,
              float* dgap, float* ws, void* stream);

/* ---- per-channel BatchNorm vector work and weight packing, one launch each ------------------------------------------
 * STATISTICS ACCUMULATORS ARE STRIPED: every `stats` / `sums` / `mom` argument of the statistics passes above
 * (ly_gemm_fwd, ly_conv3x3_fwd, ly_rfcbam3_fwd, ly_mlpblock_fwd, ly_chan_moments, ly_bnact_bwd_reduce) is
 * LY_STATS_STRIPES consecutive copies of the documented [2*nch] array (block b adds into copy b % LY_STATS_STRIPES);
 * the caller zeroes all copies and the two entry points below sum them (in double).                                  */
#define LY_STATS_STRIPES 32
/* Train-mode BatchNorm (nn.BatchNorm2d forward, training=True) from the striped sums of channels c_off .. c_off+N-1
 * (second moments at +nch): scale = gamma*invstd, shift = beta - mean*scale (+ bias*scale), batch mean / invstd for
 * the backward, running_mean/var updated with `momentum` (unbiased variance), *nbt += 1.  NULL = not wanted.           */
int ly_bn_finalize(const void* stats /* [stripes][2 nch] floats, or doubles if stats_f64 */, int stats_f64, int stripes, int nch, int c_off, int N, double count, const float* gamma, const float* beta,
                   const float* bias, float eps, float momentum, float* running_mean, float* running_var, long* nbt, float* scale,
                   float* shift, float* mean, float* invstd, void* stream);
/* BatchNorm backward coefficients from striped sums [2N] (sum dv, sum dv*u): dgamma, dbeta (ADDED to their targets, which the
 * caller zeroes: they may be the parameters' persistent gradient storage) and
 * du = alpha*dv + kappa + lambda*u  (train != 0: batch statistics; else alpha = a, kappa = lambda = 0).               */
int ly_bn_bwd_coeffs(const void* sums /* floats, or doubles if sums_f64 */, int sums_f64, int stripes, int N, double count, const float* a,
                     const float* mean, const float* invstd, int train, float* dgamma, float* dbeta, float* alpha, float* kappa, float* lambda,
                     int tr_a, int tr_b /* tr_a > 0: channel j = t*tr_b + c of the sums goes to dgamma / dbeta [c*tr_a + t] (RFCBAMConv's generate BatchNorm:
                                           sums in [tap][channel] order, parameters in [channel][tap]); 0, 0: same index */, void* stream);
/* The two entry points above for TWO BatchNorms over one stacked output (C3_CA's cv1 | cv2, models/common.py:1630-1636: channels
 * [0, c_half) and [c_half, 2 c_half) of one statistics array) in one launch each — these coefficient kernels are dependent ~4 us
 * launches on the step's critical path.  scale / shift / mean / invstd, a, alpha / kappa / lambda: stacked [2 c_half] vectors;
 * parameters, running statistics and gradient targets per unit; sums0 / sums1: the two [stripes][2 c_half] arrays of
 * ly_bnact_bwd_reduce_pair.  Always train-mode coefficients.                                                             */
int ly_bn_finalize_pair(const void* stats, int stats_f64, int stripes, int c_half, double count, const float* gamma0, const float* beta0, float eps0,
                        float momentum0, float* running_mean0, float* running_var0, long* nbt0, const float* gamma1, const float* beta1, float eps1,
                        float momentum1, float* running_mean1, float* running_var1, long* nbt1, float* scale, float* shift, float* mean,
                        float* invstd, void* stream);
int ly_bn_bwd_coeffs_pair(const void* sums0, const void* sums1, int sums_f64, int stripes, int c_half, double count, const float* a, const float* mean,
                          const float* invstd, float* dgamma0, float* dbeta0, float* dgamma1, float* dbeta1, float* alpha, float* kappa,
                          float* lambda, void* stream);
/* Fragment packing of the fp32 matrix W[r][k] = w[r*ld_r + k*ld_k] (R x K, rows zero padded to max(R, rows_to)) into the
 * [T][S][planes][64 lanes][8] bf16 layout the contraction kernels read (csrc/ly_tile.hpp): planes = 2 (hi = bf16(W),
 * lo = bf16(W - hi): the bf16x3 operand of LY_F32 calls) or 1 (hi only: LY_BF16 calls).                                */
int ly_frag_pack3(const float* w, int R, int K, long ld_r, long ld_k, int rows_to, int planes, void* out, void* stream);

/* Batched packing: one launch refreshes every packed weight image listed in a device-resident table (forward, transposed, tap-flipped,
 * concatenated ... layouts), reading the fp32 parameters in place through an index map:
 *   packed row r = ra*nrb + rb, column k = (a*nb + b)*nc + c;  element = src[ra*sra + rb*srb + a*sa + b*sb + c*sc] if r < r_valid,
 *   k < K, b < vb, c < vc, else 0;  written to row tiles [t0, t0 + T) of dst = [Ttot][S][planes][64][8] bf16.
 * blk_desc[b] = index of the descriptor block b works on; blk0 = first block of the descriptor; a descriptor takes ceil(T*S*64/256) blocks. */
typedef struct LyPackDesc {
  const float* src; void* dst;
  int r_valid, K, planes, S, t0, T;
  int nrb, nb, nc, vb, vc;
  long sra, srb, sa, sb, sc;
  long blk0;
} LyPackDesc;
int ly_pack_table(const LyPackDesc* table, const int* blk_desc, int n_blocks, void* stream);

/* ---- RFCBAMConv kernel_size 1 backward, fused recompute passes (models/rfa.py:113-129 with k = 1: `generate` is a per-channel scale +
 * BatchNorm + ReLU, so G is recomputed from x wherever it is needed instead of being materialised) --------------------------------------
 * x [n_img*HW][C] rows (stride ldx), dcd [n_img*HW][C] dense = d(loss)/d(conv input) (du . Wc^T), gw / ag / bg [C] = generate.0.weight and the
 * generate BatchNorm's batch scale / shift, ca [n_img][C], rfa [n_img*HW].
 *   pass 0: cd = G*ca*rfa (T, dense [.][C]: the conv weight gradient's operand), d_rfa[p], gmax_out[p] = max_c G, d_ca[n][c] += (zeroed by caller)
 *   pass 1: BatchNorm sums of dv into `sums` [LY_STATS_STRIPES][2C] (zeroed by caller); needs gmax (= pass 0's gmax_out) and d_mm [p][2]
 *   pass 2: du = alpha*dv + kappa + lambda*u;  dgw[c] += sum_p du*x (atomic);  dx = du*gw + dgap[n][c]*dgap_scale (dgap may be NULL)      */
typedef struct LyRf1BwdParams {
  int n_img; long HW; int C;
  const void* x; int ldx;          /* T */
  const void* dcd;                 /* T, dense */
  const float* gw; const float* ag; const float* bg; const float* ca; const float* rfa;
  void* cd; float* d_rfa; float* gmax_out; double* d_ca;                /* pass 0 outputs (d_ca: zeroed DOUBLE accumulators [n, C]) */
  const float* gmax; const float* d_mm; double* sums;                   /* pass 1 (gmax / d_mm also pass 2); sums: [32][2][taps][C] zeroed doubles */
  const float* alpha; const float* kappa; const float* lambda; const float* dgap; float dgap_scale;
  void* dx; int lddx; float* dgw;                                       /* pass 2 outputs */
  int dtype;
  int dgw_f64;                     /* != 0: dgw is a zeroed DOUBLE scratch [C] (see ly_f64_add) */
} LyRf1BwdParams;
int ly_rf1_bwd(const LyRf1BwdParams* p, int pass, void* stream);
/* Small parameter-gradient reductions, reproducibly: the kernels above that add a handful of values from many blocks (Detect bias, get_weight,
 * k = 1 generate weight, CoordAtt's MLP) can accumulate into zeroed DOUBLE scratches instead of the fp32 gradients; ly_f64_add then rounds each
 * sum into its fp32 target (dst[i] += (float)src[i]) — ONE launch for up to LY_F64_ADD_MAX vectors, after the backward pass.                   */
#define LY_F64_ADD_MAX 64
typedef struct LyF64AddTable {
  const double* src[LY_F64_ADD_MAX];
  float* dst[LY_F64_ADD_MAX];
  int n[LY_F64_ADD_MAX];
  int count;
} LyF64AddTable;
int ly_f64_add(const LyF64AddTable* t, void* stream);
/* kernel_size 3, streamed backward (the expanded tensors exist): the attention pass (pass 0) and the ReLU / routing pass (pass 1) with the
 * same 16-bytes-per-lane layout.  P.x = ug [pixels][9][C] dense, P.dcd [pixels][9][C] (pass 1 overwrites it with dv), ag / bg [9][C],
 * P.HW = Ho*Wo output pixels per image, rfa / gmax / d_rfa [n][3Ho][3Wo], d_mm [n][3Ho][3Wo][2], sums [LY_STATS_STRIPES][2][9][C] zeroed.   */
int ly_rf3s_bwd(const LyRf1BwdParams* p, int Ho, int Wo, int pass, void* stream);

/* ---- detection loss on device (utils/loss.py:121-268 ComputeLoss / build_targets, utils/metrics.py:293-354 EIoU) --------
 * One pyramid level, forward and gradient (no focal loss): anchor matching with the reference's candidate order
 * (offset k, anchor a, target t), EIoU box loss with analytic gradient, per-cell "last writer wins" objectness target (the
 * highest candidate index, i.e. the reference's sequential CPU semantics), BCE objectness (positive weight obj_pw) and — when
 * nc = no - 5 > 1 — the class BCE of every matched row (utils/loss.py:168-173: targets cn, cp at the row's class; positive weight cls_pw;
 * mean over rows x classes; gain cls_gain).
 *   p / dp   [bs][na][ny][nx][no] predictions / their gradient (dp zeroed by the caller; d(total loss)/dp on return)
 *   anchors  [na][2] in grid units (Detect.anchors[i]);  targets [nt][6] = (image, class, x, y, w, h) normalised
 *   tobj [cells] zeroed, winner [cells] filled with -1, cand_cell [5*na*nt], cand [5*na*nt][5] workspaces
 *   acc [8] zeroed: sum(1 - eiou), matches, sum of objectness BCE, rejected target rows, sum of class BCE, 3 unused  -> ly_loss_finish
 *   tbox     NULL or [5*na*nt][4]: (gx - gi, gy - gj, gw, gh) of every valid candidate (build_targets' `tbox`, utils/loss.py:262)
 *   match_only != 0: target assignment only (utils/loss.py:194-268 build_targets): cand_cell[idx] = flattened cell
 *            ((b*na + a)*ny + gj)*nx + gi of candidate idx = (k*na + a)*nt + t, or -1; tbox as above; p is read, dp/tobj untouched.
 * A target row whose image index is exactly -1 is padding and is ignored (fixed-shape target buffers of a captured step).
 * A target row whose image index is outside [0, bs) or that holds a NaN is rejected and counted in acc[3] (the torch
 * formulation raises IndexError there); ly_loss_finish then returns a NaN total.                                          */
int ly_loss_level(const float* p, float* dp, const float* anchors, const float* targets, int bs, int na, int ny, int nx, int no, long nt,
                  float anchor_t, float box_gain, float obj_gain, float balance, float* tobj, int* winner, long* cand_cell, float* cand,
                  float* acc, float* tbox, int match_only, float cls_gain, float cp, float cn, float cls_pw, float obj_pw, void* stream);
/* out[0] = (lbox + lobj + lcls) * bs (NaN if any level rejected a target row), out[1] = lbox, out[2] = lobj, out[3] = lcls (0 for nc == 1)
 * from acc [nl][8]; cells / balance: [nl] floats                                                                           */
int ly_loss_finish(const float* acc, int nl, const float* cells, const float* balance, float box_gain, float obj_gain, float cls_gain, int nc,
                   int bs, float* out, void* stream);

/* ---- fused multi-tensor optimiser step (train.py:330-341; utils/torch_utils.py:318-346 smart_optimizer, 404-432 ModelEMA) ----------
 * One table entry per state tensor (all pointers DEVICE, fp32).  Entries with a gradient take clip + SGD-nesterov + zero_grad
 * (+ EMA when `ema` is set); entries without (g == NULL: BatchNorm running statistics) take the EMA update only.              */
typedef struct LyOptTensor {
  float* p;            /* parameter (or buffer) */
  float* g;            /* gradient, zeroed after use; NULL = no optimiser update */
  float* buf;          /* momentum buffer (same size); unused when g == NULL */
  float* ema;          /* EMA copy of p, or NULL */
  long n;              /* elements */
  float wd;            /* weight decay of the tensor's group */
  int group;           /* index of its learning rate in hyper[0..2] */
  int taps, cin;       /* taps > 1: g is stored tap-major [cout][taps][cin] while p is [cout][cin][taps] (k x k conv weights) */
} LyOptTensor;
/* table [n_tensors] (device); blk_tensor / blk_off [n_blocks] (device): block b updates elements [blk_off[b], blk_off[b] + 4096) of
 * tensor blk_tensor[b].  ws: 1 double, zero before the first call (re-zeroed by every call).  hyper (device, 10 floats):
 * lr[3], momentum, max_norm (<= 0: none), ema decay (< 0: none), ema tau, updates so far, first-step flag (1 before the first call),
 * gradient scale (every gradient is read as g * scale: 1/world_size when the buckets were all-reduced as SUMs — DDP's averaging,
 * reference train.py:233-235, without a division pass per bucket)
 * — device-resident so that the call can sit inside a captured hipGraph while the host schedule rewrites it.  norm_out: NULL or
 * 1 float receiving the pre-clip global gradient norm (what clip_grad_norm_ returns).                                              */
int ly_optim_step(const LyOptTensor* table, const int* blk_tensor, const long* blk_off, int n_blocks, double* ws, float* hyper,
                  float* norm_out, void* stream);

/* ---- events across a hipGraph boundary (data-parallel step; replaces what DistributedDataParallel's reducer does with autograd hooks and
 * side streams, reference utils/torch_utils.py:55-63, train.py:233-235) ----------------------------------------------------------
 * ly_event_record on a stream that is being captured adds an event-record NODE behind the stream's capture dependencies;
 * every replay records the event when those dependencies have run, so a stream outside the graph that calls ly_stream_wait_event after the
 * replay was launched is released in the middle of the graph (gradient bucket complete -> its RCCL all-reduce starts while the rest of
 * backward executes).  On a stream that is not capturing: hipEventRecord.  Events are created without timing.            */
int ly_event_create(void** event);
int ly_event_destroy(void* event);
int ly_event_record(void* event, void* stream);
int ly_stream_wait_event(void* stream, void* event);

#ifdef __cplusplus
}
#endif
#endif
