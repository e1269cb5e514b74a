// Issue rate of the vector instructions a lane = pixel `generate` could be made of, by waves per SIMD (round 6):
//   v_fmac_f32 (SGPR weight), v_dot2c_f32_bf16 (SGPR weight pair), v_pk_fma_f32, v_max_f32
// build: hipcc -O3 --offload-arch=gfx950 tools/valu_probe.hip -o tools/bin/valu_probe ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define REP 16
#define BODY(OP) \
  float a0 = x[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
  for (int i = 0; i < iters; ++i) { \
    _Pragma("unroll") for (int r = 0; r < REP; ++r) { OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7) } \
  } \
  o[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
#define OP_FMAC(a) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a) : "s"(w), "v"(xv));
#define OP_DOT2(a) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a) : "s"(wi), "v"(xi));
#define OP_MAX(a) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(xv));
__global__ void k_fmac(const float* x, float* o, int iters, float w) { float xv = x[threadIdx.x + 64]; BODY(OP_FMAC) }
__global__ void k_dot2(const float* x, float* o, int iters, unsigned wi) { unsigned xi = __builtin_bit_cast(unsigned, x[threadIdx.x + 64]); BODY(OP_DOT2) }
__global__ void k_max(const float* x, float* o, int iters, float w) { float xv = x[threadIdx.x + 64]; BODY(OP_MAX) }
__global__ void k_pkfma(const float* x, float* o, int iters, float w) {
  f32x2 xv = {x[threadIdx.x + 64], x[threadIdx.x + 65]}, wv = {w, w};
  f32x2 a[8];
  for (int j = 0; j < 8; ++j) a[j] = (f32x2){x[threadIdx.x] + j, x[threadIdx.x] - j};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < REP; ++r)
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[j]) : "v"(xv), "v"(wv));
  }
  float s = 0;
  for (int j = 0; j < 8; ++j) s += a[j][0] + a[j][1];
  o[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F> static double run(F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  float *x, *o; hipMalloc(&x, 4096); hipMalloc(&o, 1 << 26); hipMemset(x, 0, 4096);
  const int iters = 2000;
  for (int wps = 1; wps <= 8; wps *= 2) {
    // blocks of 64 * 4 threads: one wave per SIMD each; wps blocks per CU
    const int blocks = 256 * wps;
    const double n = (double)iters * REP * 8;      // wave-instructions per wave
    double t0 = run([&] { hipLaunchKernelGGL(k_fmac, dim3(blocks), dim3(256), 0, 0, x, o, iters, 1.0f); });
    double t1 = run([&] { hipLaunchKernelGGL(k_dot2, dim3(blocks), dim3(256), 0, 0, x, o, iters, 0x3f803f80u); });
    double t2 = run([&] { hipLaunchKernelGGL(k_pkfma, dim3(blocks), dim3(256), 0, 0, x, o, iters, 1.0f); });
    double t3 = run([&] { hipLaunchKernelGGL(k_max, dim3(blocks), dim3(256), 0, 0, x, o, iters, 1.0f); });
    // ns per wave-instruction per SIMD = t / (n * wps)
    printf("waves/SIMD %d: ns per wave-instruction per SIMD (x ~2.4 = cycles):  fmac %.3f  dot2c %.3f  pk_fma %.3f  max %.3f\n", wps,
           t0 * 1e6 / (n * wps), t1 * 1e6 / (n * wps), t2 * 1e6 / (n * wps), t3 * 1e6 / (n * wps));
  }
  return 0;
}
