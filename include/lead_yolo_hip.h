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
 * (models/common.py:1478-1482).  wp/w1/w2 are frag-packed (lead-yolo_amd/pack.py) from
 * spatial_mixing.partial_conv3.weight / mlp.0.weight / mlp.3.weight; bn_* have 2C entries.
 * x and y must not alias. Built for C in {16,24,40,80,160,320}. */
int ly_mlpblock_fwd(const float* x, float* y, int n_img, int H, int W, int C, const float* wp, const float* w1,
                    const float* w2, const float* bn_scale, const float* bn_shift, void* stream);
/* number of floats of the three packed weight buffers for a given C */
int ly_mlpblock_pack_sizes(int C, long* n_wp, long* n_w1, long* n_w2);

#ifdef __cplusplus
}
#endif
#endif
