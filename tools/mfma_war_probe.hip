// Probe for the cause of the "packed FMA beside MFMAs" corruption recorded in csrc/ly_rf3c.hpp (round 3).
// tools/pkfma_probe.hip shows v_pk_fma_f32 itself — every operand form, also with the destination aliasing the broadcast source —
// bit-exact beside an MFMA stream (0 wrong values of 47 M x 7 forms, round 6).  What an inline-asm vector instruction does NOT get is
// the compiler's MFMA hazard handling: GCNHazardRecognizer inserts the wait states between an MFMA that still READS an operand register
// and a later vector instruction that WRITES it (and between an MFMA's result and its reader) by looking at instruction classes, and an
// `asm` statement has none.  This probe puts exactly that pair in one wave:
//      acc = v_mfma_f32_16x16x16_f16(a, b, acc);   <then>   a = garbage        (a = the MFMA's A operand pair, dead after the MFMA)
// with the overwrite written (0) as a C++ statement (hipcc emits the vector instruction and whatever s_nop the hazard table asks for) and
// (1) as inline asm (same instruction, no hazard handling), (2) as inline asm behind an explicit s_nop 7, alone and beside an aggressor
// kernel that keeps the SIMDs' matrix pipes busy.  Expected: every value exact in (0) and (2); wrong values in (1) = the mechanism.
//
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_war_probe.hip -o tools/bin/mfma_war_probe ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int WHICH>
__global__ __launch_bounds__(256) void victim(const float* __restrict__ src, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  // small integers: every product and sum is exact in f16 / f32
  h4 a_keep, b_keep;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a_keep[i] = (_Float16)(float)((lane + i) % 3 - 1);
    b_keep[i] = (_Float16)(float)((lane * 2 + i) % 3 - 1);
  }
  const f32x2 garbage = {src[lane], src[lane + 64]};       // 1000.5, -777.25: never a legal operand value
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    h4 a = a_keep, b = b_keep;
    asm volatile("" : "+v"(a), "+v"(b));                    // operands in registers of their own for this trip
    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, acc, 0, 0, 0);
    // overwrite an operand pair right behind the MFMA (the pair is dead: the next trip rebuilds it)
    f32x2 victim_pair = __builtin_bit_cast(f32x2, WHICH == 0 ? a : b);
    if constexpr (MODE == 0) {
      victim_pair = victim_pair * garbage + garbage;
      asm volatile("" ::"v"(victim_pair));                  // keep the write alive
    } else if constexpr (MODE == 1) {
      asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(victim_pair) : "v"(garbage));
    } else {
      asm volatile("s_nop 7\n\tv_pk_fma_f32 %0, %0, %1, %1" : "+v"(victim_pair) : "v"(garbage));
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) out[(long)tid * 4 + r] = acc[r];
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

template <int MODE, int WHICH>
static long run(const float* dsrc, float* dout, std::vector<float>& ref, int blocks, int iters, hipStream_t sv, hipStream_t sa, bool beside, float* sink, bool make_ref) {
  std::vector<float> got((size_t)blocks * 256 * 4);
  long bad = 0;
  for (int rep = 0; rep < 10; ++rep) {
    if (beside) hipLaunchKernelGGL(aggressor, dim3(1024), dim3(256), 0, sa, sink, 20000);
    hipLaunchKernelGGL((victim<MODE, WHICH>), dim3(blocks), dim3(256), 0, sv, dsrc, dout, iters);
    CK(hipStreamSynchronize(sv));
    CK(hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost));
    if (make_ref && rep == 0) ref = got;
    for (size_t i = 0; i < got.size(); ++i) bad += got[i] != ref[i];
    CK(hipStreamSynchronize(sa));
  }
  return bad;
}

int main() {
  const int blocks = 1024, iters = 200;
  std::vector<float> src(128);
  for (int i = 0; i < 128; ++i) src[i] = i < 64 ? 1000.5f : -777.25f;
  float *dsrc, *dout, *sink;
  CK(hipMalloc(&dsrc, 512)); CK(hipMalloc(&dout, (size_t)blocks * 256 * 16)); CK(hipMalloc(&sink, 4096));
  CK(hipMemcpy(dsrc, src.data(), 512, hipMemcpyHostToDevice));
  hipStream_t sv, sa;
  CK(hipStreamCreate(&sv)); CK(hipStreamCreate(&sa));
  // reference: the compiler-scheduled form alone on the chip
  std::vector<float> ref;
  run<0, 0>(dsrc, dout, ref, blocks, iters, sv, sa, false, sink, true);
  const char* mname[3] = {"C++ overwrite (compiler's hazard handling)", "inline-asm overwrite, no s_nop", "inline-asm overwrite behind s_nop 7"};
  for (int beside = 0; beside < 2; ++beside) {
    long bad[6];
    bad[0] = run<0, 0>(dsrc, dout, ref, blocks, iters, sv, sa, beside, sink, false);
    bad[1] = run<1, 0>(dsrc, dout, ref, blocks, iters, sv, sa, beside, sink, false);
    bad[2] = run<2, 0>(dsrc, dout, ref, blocks, iters, sv, sa, beside, sink, false);
    bad[3] = run<0, 1>(dsrc, dout, ref, blocks, iters, sv, sa, beside, sink, false);
    bad[4] = run<1, 1>(dsrc, dout, ref, blocks, iters, sv, sa, beside, sink, false);
    bad[5] = run<2, 1>(dsrc, dout, ref, blocks, iters, sv, sa, beside, sink, false);
    for (int v = 0; v < 6; ++v)
      printf("%-14s operand %c  %-46s wrong values: %ld of %d x 10 runs\n", beside ? "beside MFMAs:" : "alone:", v < 3 ? 'A' : 'B', mname[v % 3], bad[v], blocks * 256 * 4);
  }
  return 0;
}
