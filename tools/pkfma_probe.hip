// Probe for the packed-FMA finding of csrc/ly_rf3c.hpp: do hand-written v_pk_fma_f32 with op_sel weight broadcasts return
// wrong values when another wave of the SIMD issues MFMAs?  Which ingredient matters: the asm, the packing, the op_sel broadcast?
//
//   hipcc -O3 --offload-arch=gfx950 -I lead-yolo_amd/csrc -I include tools/pkfma_probe.hip -o /tmp/pkfma_probe && /tmp/pkfma_probe
//
// Victim kernels evaluate the 9 x 9 generate chains (81 FMAs per pixel pair, inputs small integers: every result is exact in fp32, so
// ANY deviation is an error, not rounding) in five forms; the aggressor is a zero-LDS kernel of back-to-back bf16 MFMAs on a second stream
// with few enough registers to co-reside with the victims.  Output: mismatching lanes per variant, alone and beside the aggressor.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { V_PLAIN = 0, V_ASM_OPSEL = 1, V_ASM_SPLAT = 2, V_VEC = 3, V_ASM_OPSEL_NOP = 4, V_ASM_SCALAR = 5, V_ASM_DST_IS_BCAST_SRC = 6, NVAR = 7 };
static const char* vname[NVAR] = {"plain fmaf (compiler)", "asm v_pk_fma_f32 op_sel broadcast", "asm v_pk_fma_f32, weight pre-splatted (no op_sel)",
                                  "vector type fma (compiler may pack)", "asm v_pk_fma_f32 op_sel + s_nop 1", "asm v_fma_f32 x2 (not packed)",
                                  "asm v_pk_fma_f32, dst pair == lo-broadcast weight pair"};

template <int V>
__device__ __forceinline__ f32x2 pkfma(const f32x2 x, const f32x2 w, f32x2 acc, const int sel) {
  if constexpr (V == V_PLAIN) {
    const float ww = w[sel];
    return (f32x2){__builtin_fmaf(x[0], ww, acc[0]), __builtin_fmaf(x[1], ww, acc[1])};
  } else if constexpr (V == V_ASM_OPSEL) {
    if (sel) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(x), "v"(w));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(x), "v"(w));
    return acc;
  } else if constexpr (V == V_ASM_OPSEL_NOP) {
    if (sel) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\ts_nop 1" : "+v"(acc) : "v"(x), "v"(w));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\ts_nop 1" : "+v"(acc) : "v"(x), "v"(w));
    return acc;
  } else if constexpr (V == V_ASM_DST_IS_BCAST_SRC) {
    // what an "=v" output (no early clobber) allows the register allocator to do when the weight pair dies at the instruction:
    // the destination IS the pair whose low dword both halves read.  If the two halves execute as two passes, the second one may
    // read the first one's result instead of the weight.
    f32x2 d = w;
    if (sel) asm("v_pk_fma_f32 %0, %1, %0, %2 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(d) : "v"(x), "v"(acc));
    else asm("v_pk_fma_f32 %0, %1, %0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(d) : "v"(x), "v"(acc));
    return d;
  } else if constexpr (V == V_ASM_SPLAT) {
    const f32x2 ws = {w[sel], w[sel]};
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(ws));
    return acc;
  } else if constexpr (V == V_ASM_SCALAR) {
    const float ww = w[sel];
    float a0 = acc[0], a1 = acc[1];
    asm("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(x[0]), "v"(ww));
    asm("v_fma_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(x[1]), "v"(ww));
    return (f32x2){a0, a1};
  } else {
    const f32x2 ws = {w[sel], w[sel]};
    return __builtin_elementwise_fma(x, ws, acc);
  }
}

// out[thread][it][9 taps][2]: the same arithmetic as rc_generate<false> of ly_rf3c.hpp: a[t] = b[t] + sum_u w[t][u] * x[u]
template <int V>
__global__ __launch_bounds__(256) void victim(const float* __restrict__ wsrc, const float* __restrict__ xsrc, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  f32x2 w[46];
#pragma unroll
  for (int i = 0; i < 46; ++i) w[i] = (f32x2){wsrc[(tid & 63) * 92 + 2 * i], wsrc[(tid & 63) * 92 + 2 * i + 1]};
  f32x2 sum[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) sum[t] = (f32x2){0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    f32x2 x[9], a[9];
#pragma unroll
    for (int u = 0; u < 9; ++u) x[u] = (f32x2){xsrc[((tid + it * 7) & 1023) * 18 + 2 * u], xsrc[((tid + it * 7) & 1023) * 18 + 2 * u + 1]};
#pragma unroll
    for (int t = 0; t < 9; ++t) a[t] = (f32x2){w[(81 + t) >> 1][(81 + t) & 1], w[(81 + t) >> 1][(81 + t) & 1]};
#pragma unroll
    for (int u = 0; u < 9; ++u)
#pragma unroll
      for (int t = 0; t < 9; ++t) a[t] = pkfma<V>(x[u], w[(t * 9 + u) >> 1], a[t], (t * 9 + u) & 1);
#pragma unroll
    for (int t = 0; t < 9; ++t) sum[t] += a[t];            // integer valued, < 2^24: exact
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    out[(long)tid * 18 + 2 * t] = sum[t][0];
    out[(long)tid * 18 + 2 * t + 1] = sum[t][1];
  }
}

__global__ __launch_bounds__(256) void aggressor(float* __restrict__ sink, int iters) {
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)((threadIdx.x + i) & 3); b[i] = (__bf16)(float)((threadIdx.x * 3 + i) & 3); }
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
  }
  if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 1.2345f) sink[threadIdx.x] = acc[0][0];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

template <int V>
static long run(const float* dw, const float* dx, float* dout, const std::vector<float>& want, int blocks, int iters, hipStream_t sv, hipStream_t sa, bool beside, float* sink) {
  std::vector<float> got(want.size());
  long bad = 0;
  for (int rep = 0; rep < 10; ++rep) {
    if (beside) hipLaunchKernelGGL(aggressor, dim3(1024), dim3(256), 0, sa, sink, 60000);
    hipLaunchKernelGGL(victim<V>, dim3(blocks), dim3(256), 0, sv, dw, dx, dout, iters);
    CK(hipStreamSynchronize(sv));
    CK(hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < got.size(); ++i) bad += got[i] != want[i];
    CK(hipStreamSynchronize(sa));
  }
  return bad;
}

int main() {
  const int blocks = 1024, iters = 50;
  std::vector<float> w(64 * 92), x(1024 * 18);
  srand(7);
  const bool exact = getenv("PK_EXACT") != nullptr;      // PK_EXACT=1: small integers (no rounding anywhere); default: random floats, the
                                                          // expected value is the same fused chain evaluated with the host's fmaf (bitwise)
  for (auto& v : w) v = exact ? (float)(rand() % 7 - 3) : (float)rand() / RAND_MAX - 0.5f;
  for (auto& v : x) v = exact ? (float)(rand() % 5 - 2) : (float)rand() / RAND_MAX * 2.f - 1.f;
  const size_t nout = (size_t)blocks * 256 * 18;
  std::vector<float> want(nout);
  for (long tid = 0; tid < (long)blocks * 256; ++tid) {
    float s[18] = {0};
    for (int it = 0; it < iters; ++it)
      for (int t = 0; t < 9; ++t)
        for (int p = 0; p < 2; ++p) {
          float a = w[(tid & 63) * 92 + 81 + t];
          for (int u = 0; u < 9; ++u) a = fmaf(x[((tid + it * 7) & 1023) * 18 + 2 * u + p], w[(tid & 63) * 92 + t * 9 + u], a);
          s[2 * t + p] += a;
        }
    for (int i = 0; i < 18; ++i) want[tid * 18 + i] = s[i];
  }
  float *dw, *dx, *dout, *sink;
  CK(hipMalloc(&dw, w.size() * 4)); CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dout, nout * 4)); CK(hipMalloc(&sink, 4096));
  CK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
  hipStream_t sv, sa;
  CK(hipStreamCreate(&sv)); CK(hipStreamCreate(&sa));
  for (int beside = 0; beside < 2; ++beside) {
    long bad[NVAR];
    bad[0] = run<0>(dw, dx, dout, want, blocks, iters, sv, sa, beside, sink);
    bad[1] = run<1>(dw, dx, dout, want, blocks, iters, sv, sa, beside, sink);
    bad[2] = run<2>(dw, dx, dout, want, blocks, iters, sv, sa, beside, sink);
    bad[3] = run<3>(dw, dx, dout, want, blocks, iters, sv, sa, beside, sink);
    bad[4] = run<4>(dw, dx, dout, want, blocks, iters, sv, sa, beside, sink);
    bad[5] = run<5>(dw, dx, dout, want, blocks, iters, sv, sa, beside, sink);
    bad[6] = run<6>(dw, dx, dout, want, blocks, iters, sv, sa, beside, sink);
    for (int v = 0; v < NVAR; ++v)
      printf("%-14s %-52s wrong values: %ld of %zu x 10 runs\n", beside ? "beside MFMAs:" : "alone:", vname[v], bad[v], nout);
  }
  return 0;
}
