// Fused multi-tensor optimiser step of the training loop, fp32 master weights, gfx950.
//
// Replaces, per optimisation step (reference train.py:330-341, utils/torch_utils.py:318-346, 404-432):
//   scaler.unscale_ / torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=10.0)     global L2 norm over ALL gradients
//   torch.optim.SGD(momentum, nesterov=True) with three parameter groups (lr, weight_decay per group)
//   optimizer.zero_grad()
//   ModelEMA.update: ema = d*ema + (1-d)*model  for every floating state entry, d = decay*(1 - exp(-updates/tau))
// — ~50 foreach launches plus a Python loop over 319 state tensors — with TWO launches over a device-resident table of tensors:
//   ly_optim_norm   sum of squares of every gradient  -> ws[0] (double; ws zeroed by the caller / by the previous step's update)
//   ly_optim_update coef = min(1, max_norm / (sqrt(ws[0]) + 1e-6));  g' = g*coef (+ wd*p);  buf = mom*buf + g' (first step: g');
//                   p -= lr * (g' + mom*buf);  g = 0;  ema = d*ema + (1-d)*p;  entries without a gradient (BatchNorm running
//                   statistics) only take the EMA update.
// Nothing here depends on host values that change per step: learning rates, the EMA decay and the step counter live in the
// device array `hyper` (the host schedule writes it), so the two launches can sit inside a captured hipGraph of the whole step.
#include "ly_common.hpp"
#include "ly_params.h"

#define LY_OPT_CHUNK 4096                 // elements per block

// hyper: [0..2] lr of groups 0..2, [3] momentum, [4] max_norm (<= 0: no clipping), [5] ema decay (< 0: no EMA), [6] ema tau,
//        [7] updates so far (incremented by the update kernel), [8] 1.0 until the first step has initialised the momentum buffers,
//        [9] gradient scale: gradients are read as g * scale (1/world_size after a SUM all-reduce of the buckets; 1 on one GPU)
__global__ __launch_bounds__(LY_THREADS) void ly_optim_norm_kernel(const LyOptTensor* __restrict__ tab, const int* __restrict__ blk_tensor,
                                                                   const long* __restrict__ blk_off, double* __restrict__ ws) {
  __shared__ float red[4];
  const LyOptTensor t = tab[blk_tensor[blockIdx.x]];
  float s = 0.f;
  if (t.g) {
    const long off = blk_off[blockIdx.x];
    const long end = off + LY_OPT_CHUNK < t.n ? off + LY_OPT_CHUNK : t.n;
    for (long i = off + threadIdx.x; i < end; i += LY_THREADS) { const float v = t.g[i]; s += v * v; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0 && t.g) atomicAdd(ws, (double)(red[0] + red[1] + red[2] + red[3]));
}

__global__ __launch_bounds__(LY_THREADS) void ly_optim_update_kernel(const LyOptTensor* __restrict__ tab, const int* __restrict__ blk_tensor,
                                                                     const long* __restrict__ blk_off, const double* __restrict__ ws,
                                                                     const float* __restrict__ hyper, float* __restrict__ norm_out) {
  const LyOptTensor t = tab[blk_tensor[blockIdx.x]];
  const long off = blk_off[blockIdx.x];
  const long end = off + LY_OPT_CHUNK < t.n ? off + LY_OPT_CHUNK : t.n;
  const float mom = hyper[3], max_norm = hyper[4], ema_decay = hyper[5], tau = hyper[6], updates = hyper[7] + 1.f;
  const bool first = hyper[8] != 0.f;
  const float gscale = hyper[9];
  const float total = (float)sqrt(ws[0]) * gscale;          // norm of the scaled gradients
  float coef = 1.f;
  if (max_norm > 0.f) { coef = max_norm / (total + 1e-6f); coef = coef > 1.f ? 1.f : coef; }
  if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) *norm_out = total;
  const float d = ema_decay >= 0.f ? ema_decay * (1.f - __expf(-updates / tau)) : 0.f;
  const float lr = t.group >= 0 ? hyper[t.group] : 0.f;
  const long tc = (long)t.taps * t.cin;
  for (long i = off + threadIdx.x; i < end; i += LY_THREADS) {
    float p = t.p[i];
    if (t.g) {
      // gradient storage of k x k convolution weights may be tap-major ([cout][kh*kw][cin], what ly_wgrad writes with contiguous
      // atomics) while the parameter is [cout][cin][kh*kw]
      long gi = i;
      if (t.taps > 1) {
        const long co = i / tc, r = i - co * tc;
        const int c = (int)(r / t.taps), tp = (int)(r - (long)c * t.taps);
        gi = (co * t.taps + tp) * t.cin + c;
      }
      float g = t.g[gi] * (coef * gscale);
      if (t.wd != 0.f) g += t.wd * p;
      const float b = first ? g : mom * t.buf[i] + g;
      t.buf[i] = b;
      p -= lr * (g + mom * b);
      t.p[i] = p;
      t.g[gi] = 0.f;
    }
    if (t.ema && ema_decay >= 0.f) t.ema[i] = d * t.ema[i] + (1.f - d) * p;
  }
}

// last launch of the step: step counter, first-step flag and the norm accumulator for the next step
__global__ void ly_optim_finish_kernel(double* __restrict__ ws, float* __restrict__ hyper) {
  if (threadIdx.x == 0) { ws[0] = 0.0; hyper[7] += 1.f; hyper[8] = 0.f; }
}

extern "C" int ly_optim_step(const LyOptTensor* table, const int* blk_tensor, const long* blk_off, int n_blocks, double* ws, float* hyper,
                             float* norm_out, void* stream) {
  LY_CHECK(table && blk_tensor && blk_off && ws && hyper && n_blocks > 0, "optim_step: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(ly_optim_norm_kernel, dim3((unsigned)n_blocks), dim3(LY_THREADS), 0, st, table, blk_tensor, blk_off, ws);
  hipLaunchKernelGGL(ly_optim_update_kernel, dim3((unsigned)n_blocks), dim3(LY_THREADS), 0, st, table, blk_tensor, blk_off, ws, hyper, norm_out);
  hipLaunchKernelGGL(ly_optim_finish_kernel, dim3(1), dim3(64), 0, st, ws, hyper);
  LY_LAUNCH_CHECK();
  return 0;
}
