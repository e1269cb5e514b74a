// bf16-storage instantiations of the pointwise-convolution GEMM (kernel: ly_gemm.hpp, C ABI: ly_gemm.hip)
#include "ly_gemm.hpp"

int ly_gemm_dispatch_bf16(const LyGemmParams& P, hipStream_t st) { return ly_gemm_dispatch<__bf16>(P, st); }
