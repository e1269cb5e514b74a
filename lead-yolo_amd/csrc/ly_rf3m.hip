// RFCBAMConv kernel_size 3, bf16 storage, inference form: `generate` ON THE MATRIX CORES (reference models/rfa.py:113-129), gfx950.
//
// The lane = channel kernels (ly_rf3c.hip) regenerate G = relu(bn(generate(x))) on the VALU: 81 MAC per (output pixel, channel), ~650 vector
// instructions per wave and 32-channel chunk, then hand G * ca * rfa to the MFMAs through an LDS operand tile behind two barriers.  Here the
// depthwise 3x3 is itself an MFMA product.  Per group of 4 input channels the 4 x 9 taps x 9 patch weights are a block-diagonal matrix
//      A_g [32 rows = (channel, tap 0..7)] x [k = (patch slot u, channel)]        (+ one row block per chunk for tap 8 of 32 channels)
// and the input patches of 32 output pixels are the B operand, read straight from a bf16 copy of x in LDS (8 bytes = 4 channels of one
// position).  v_mfma_f32_32x32x16_bf16 leaves D[(channel, tap)][pixel] with the pixel on the lane and 16 (channel, tap) rows in registers —
// which IS the B-operand layout of the main contraction over k = (channel, tap) (MI355X guide: "an accumulator tile as the next MFMA's
// operand"), so  relu, * ca[c] * rfa[pixel, tap], bf16 conversion  happen in registers and G' never touches LDS.  BatchNorm's shift rides in
// a spare patch slot whose B entries are 1.0 (split hi + lo + lolo: exact in fp32).  ~3/4 of the generate MFMA flops multiply zeros; at 16x
// the vector rate that is still 4x cheaper than the VALU form, and the VALU is left with ~2 instructions per generated value.
//
// Work split: one wave = one tile of <= 32 output pixels (TH x TW), all input channels (chunks of 16), MT x 32 output channels; the 8 waves
// of a block (two per SIMD: one wave's VALU phase runs under the other's MFMAs) share only the WEIGHT STREAM: all A fragments in consumption
// order (pack.rf3m_stream), copied by LDS-DMA (global_load_lds_dwordx4, one 1 KiB fragment per wave instruction) into a two-stage LDS ring,
// one barrier per stage.  The x tile is private to the wave.
//
//   ly_rf3m_stats : [max_c, mean_c] of G -> mm[n, 3Ho, 3Wo, 2] and the SE pooling partials part[n][tile][C]   (models/rfa.py:90, 125-126)
//   ly_rf3m_fwd   : out = relu(bn(conv_{3x3, stride 3}(G * ca * rfa)))                                         (models/rfa.py:124, 128-129)
#include "ly_tile.hpp"
#include "ly_params.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

#define RM_CB 16                        // channels per chunk
#define RM_UNITS 4                      // 4-channel groups per chunk
#define RM_PS 40                        // bytes per input position of the wave's x tile: 16 channels bf16 + 8
#define RM_MAXPOS 160                   // input positions of a wave tile: 8 x 4 output pixels at stride 2 read 17 x 9 = 153
#define RM_XTILE 6464                   // (RM_MAXPOS + 1) * RM_PS = 6440, + the position that holds 1.0 in every channel (BatchNorm shift slot), rounded to 64
#define RM_NV (RM_MAXPOS * 2 / 64)      // 16-byte staging items per lane and chunk (RM_MAXPOS * 2 = 64 RM_NV exactly)
#define RM_GF 6                         // generate fragments per unit: 3 k-steps of the (4 channels x 8 taps) tile + 3 of the tap-8 tile
#define RM_WAVES 8
#define RM_THREADS (RM_WAVES * 64)

__device__ __forceinline__ f32x16 rm_mfma(const bf16x8 a, const bf16x8 b, const f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

// LDS-DMA of one weight fragment (1 KiB): lane l copies 16 bytes from gsrc to lds_dst + 16 l.  Not visible to hipcc's waitcnt bookkeeping
// (an asm statement): the kernels below wait for it by hand (s_waitcnt vmcnt) before the barrier that publishes a stage.
__device__ __forceinline__ void rm_dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void rm_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

__device__ __forceinline__ unsigned rm_lds_addr(const void* p) { return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p; }

// geometry + staging plan of the wave's tile
struct RmTile {
  int n, oy0, ox0;            // image, first output pixel
  bool valid;
};
__device__ __forceinline__ RmTile rm_tile(long wt, long total, int nct, int nrt, int TH, int TW) {
  RmTile t;
  t.valid = wt < total;
  long w = t.valid ? wt : total - 1;
  const int ct = (int)(w % nct); w /= nct;
  const int rt = (int)(w % nrt);
  t.n = __builtin_amdgcn_readfirstlane((int)(w / nrt));
  t.oy0 = __builtin_amdgcn_readfirstlane(rt * TH);
  t.ox0 = __builtin_amdgcn_readfirstlane(ct * TW);
  return t;
}

struct RmStage {
  int soff[RM_NV];            // element offset of the item in x at channel 0 of the chunk; -1: outside the image or past the tile (stored as zero)
  ly_u32x4 pv[RM_NV];
};
__device__ __forceinline__ void rm_stage_plan(RmStage& S, int lane, const RmTile& T, int s, int TH, int TW, int H, int W, int ldx) {
  const int IH = s * (TH - 1) + 3, IW = s * (TW - 1) + 3;
  const int iy0 = s * T.oy0 - 1, ix0 = s * T.ox0 - 1;
#pragma unroll
  for (int e = 0; e < RM_NV; ++e) {
    const int idx = lane + 64 * e;
    const int ip = idx >> 1, v = idx & 1;
    int so = -1;
    if (ip < IH * IW) {
      const int r = ip / IW, q = ip - r * IW;
      const int iy = iy0 + r, ix = ix0 + q;
      so = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? (int)((((long)T.n * H + iy) * W + ix) * ldx + 8 * v) : -1;
    }
    S.soff[e] = so;
  }
}
__device__ __forceinline__ void rm_stage_load(RmStage& S, const __bf16* __restrict__ x, int c0) {
#pragma unroll
  for (int e = 0; e < RM_NV; ++e) S.pv[e] = *reinterpret_cast<const ly_u32x4*>(S.soff[e] >= 0 ? x + S.soff[e] + c0 : x);     // clamped address, never a branch around a load
}
__device__ __forceinline__ void rm_stage_store(const RmStage& S, char* __restrict__ xs, int lane) {
#pragma unroll
  for (int e = 0; e < RM_NV; ++e) {
    // every lane stores every item (64 RM_NV items = exactly the RM_MAXPOS positions of the tile's LDS image; positions past the tile take
    // zeros): no branch around a store, so the wait for the prefetch registers is unconditional and hipcc needs no later vmcnt(0) — which
    // would also drain the weight copies issued by hand in between
    const int idx = lane + 64 * e;
    const ly_u32x4 v = S.soff[e] >= 0 ? S.pv[e] : (ly_u32x4){0u, 0u, 0u, 0u};
    char* d = xs + (idx >> 1) * RM_PS + (idx & 1) * 16;
    *reinterpret_cast<ly_u32x2*>(d) = (ly_u32x2){v[0], v[1]};
    *reinterpret_cast<ly_u32x2*>(d + 8) = (ly_u32x2){v[2], v[3]};
  }
}
// LDS byte offsets (relative to the wave's x tile) of the lane's six patch slots: k-step st, half k: slot u = 4 st + 2 h + k; u < 9: the
// patch position (u / 3, u % 3) of the lane's pixel, u >= 9: the position of ones
__device__ __forceinline__ void rm_patch_offsets(int (&bo)[3][2], int lane, int s, int TH, int TW) {
  const int IW = s * (TW - 1) + 3;
  const int px = lane & 31, h = lane >> 5;
  const int pxc = px < TH * TW ? px : 0;
  const int ly = pxc / TW, lx = pxc - ly * TW;
  const int pos0 = s * ly * IW + s * lx;
#pragma unroll
  for (int st = 0; st < 3; ++st)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int u = 4 * st + 2 * h + k;
      bo[st][k] = (u < 9 ? pos0 + (u / 3) * IW + (u % 3) : RM_MAXPOS) * RM_PS;
    }
}
// B operand of generate k-step st for 4-channel group j of the chunk
__device__ __forceinline__ bf16x8 rm_patch_frag(const char* __restrict__ xs, const int (&bo)[3][2], int st, int j) {
  const ly_u32x2 a = *reinterpret_cast<const ly_u32x2*>(xs + bo[st][0] + 8 * j);
  const ly_u32x2 b = *reinterpret_cast<const ly_u32x2*>(xs + bo[st][1] + 8 * j);
  return __builtin_bit_cast(bf16x8, (ly_u32x4){a[0], a[1], b[0], b[1]});
}
__device__ __forceinline__ bf16x8 rm_wfrag(const char* __restrict__ ring, int f) { return *reinterpret_cast<const bf16x8*>(ring + f * 1024); }

// two generated values -> relu(v) * f as two bf16 (f >= 0: relu(v * f) = relu(v) * f; the relu is an integer max on the packed pair)
__device__ __forceinline__ unsigned rm_post2(float v0, float v1, const f32x2 f) {
  const f32x2 p = (f32x2){v0, v1} * f;
  const s16x2 b = __builtin_bit_cast(s16x2, __builtin_convertvector(p, bf16x2));
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(b, (s16x2){0, 0}));
}

// ---------------------------------------------------------------------------------------------------
// main contraction.  LDS: ring [2][RB] | epilogue scale / shift | 8 x (x tile | ca[C])
// ---------------------------------------------------------------------------------------------------
// development: cycles per phase of the contraction kernel (s_memtime stamps; wave 0 of every 16th block adds its sums): tools/rf3m_check.py --prof
__device__ unsigned long long rm_prof[10];        // [8], [9]: s_memtime / s_memrealtime (100 MHz) spans of the sampled waves: the in-kernel clock
extern "C" int ly_rf3m_prof(unsigned long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(rm_prof), sizeof(unsigned long long) * 10) != hipSuccess) return -1;
  if (reset) { unsigned long long z[10] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rm_prof), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#define RM_T(i) do { if constexpr (PROF) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); pacc[i] += t_ - pt0; pt0 = t_; __builtin_amdgcn_sched_barrier(0); } } while (0)

template <int MT, bool PROF>
__global__ __launch_bounds__(RM_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void ly_rf3m_fwd_kernel(const LyRfcbam3Params P, const int nct, const int nrt,
                                                                                                            const int gy, const int lds_x, const int dbg) {
  unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long pt0 = PROF ? __builtin_readcyclecounter() : 0;
  const unsigned long long pclk0 = pt0, preal0 = PROF ? __builtin_amdgcn_s_memrealtime() : 0;
  constexpr int UF = RM_GF + 2 * MT;                   // fragments per unit: generate (6) + main (2 k-steps x MT)
  constexpr int SF = 2 * UF;                           // ... per stage (2 units); the second stage of a chunk carries the tap-8 tile's MT main fragments
  constexpr int CF = 2 * SF + MT;                      // ... per chunk
  constexpr int RB = (SF + MT) * 1024;                 // bytes of one ring buffer
  extern __shared__ __attribute__((aligned(16))) char rm_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5;
  const __bf16* const x = reinterpret_cast<const __bf16*>(P.x);
  __bf16* const out = reinterpret_cast<__bf16*>(P.out);
  const int bid = ly_xcd_remap((int)blockIdx.x, (int)gridDim.x);            // neighbouring tiles on one XCD's L2
  const int by = bid % gy;
  const long pb = bid / gy;
  const long total = (long)P.n_img * nrt * nct;
  const RmTile T = rm_tile(pb * RM_WAVES + wave, total, nct, nrt, P.TH, P.TW);
  const int NCH = P.C / RM_CB;
  const int NST = 2 * NCH;

  char* const ring = rm_smem;
  float* const ess = reinterpret_cast<float*>(rm_smem + 2 * RB);        // [2][MT * 32]: the block's epilogue scale / shift
  char* const xs = rm_smem + 2 * RB + 2 * MT * 32 * 4 + wave * lds_x;
  float* const cas = reinterpret_cast<float*>(xs + RM_XTILE);
  const char* const rl0 = ring + lane * 16;             // the lane's 16 bytes of fragment 0, ring buffer 0 / 1
  const char* const rl1 = rl0 + RB;
  const unsigned ring_lds = rm_lds_addr(ring);
  const char* const wsrc = reinterpret_cast<const char*>(P.wp) + (long)by * NCH * CF * 1024 + lane * 16;

  // stage `sidx` (global stage counter, 2 per chunk) of this block's stream -> ring buffer sidx & 1; wave w copies fragments w, w + 8, ...
  auto issue = [&](int sidx) {
    const int ch = sidx >> 1, q = sidx & 1;
    const char* src = wsrc + ((long)ch * CF + q * SF) * 1024;
    const unsigned dst = ring_lds + (sidx & 1) * RB;
    const int cnt = (dbg & 1) ? 0 : (q == 1 ? SF + MT : SF);
    for (int f = wave; f < cnt; f += RM_WAVES) rm_dma16(src + f * 1024, dst + f * 1024);
  };
  issue(0);

  RmStage St;
  rm_stage_plan(St, lane, T, P.s, P.TH, P.TW, P.H, P.W, P.ldx);
  rm_stage_load(St, x, 0);
  int bo[3][2];
  rm_patch_offsets(bo, lane, P.s, P.TH, P.TW);

  // the lane's pixel and its nine rfa factors: taps 4h .. 4h+3 (rows of the (channel, tap) tiles) and tap 8; zero outside the map (G' = 0)
  const int px = lane & 31;
  const int ly = px / P.TW, lx = px - ly * P.TW;
  const int oy = T.oy0 + ly, ox = T.ox0 + lx;
  const bool pok = T.valid && px < P.TH * P.TW && oy < P.Ho && ox < P.Wo;
  f32x2 rf[2];
  float rf8;
  {
    const float* rp = P.rfa + ((long)T.n * 3 * P.Ho + 3 * (pok ? oy : 0)) * (3 * P.Wo) + 3 * (pok ? ox : 0);
    float r[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int t = i < 4 ? 4 * h + i : 8;
      r[i] = rp[(t / 3) * (3 * P.Wo) + (t % 3)];
    }
    rf[0] = pok ? (f32x2){r[0], r[1]} : (f32x2){0.f, 0.f};
    rf[1] = pok ? (f32x2){r[2], r[3]} : (f32x2){0.f, 0.f};
    rf8 = pok ? r[4] : 0.f;
  }
  if (tid < 2 * MT * 32) ess[tid] = (tid < MT * 32 ? P.e_scale : P.e_shift - MT * 32)[by * MT * 32 + tid];        // (read behind the stage barriers)
  // ca of the wave's image in LDS (the tap-8 tile needs a per-lane channel), and the position of ones
  for (int i = lane; i < P.C; i += 64) cas[i] = P.ca[(long)T.n * P.C + i];
  if (lane < RM_PS / 8) *reinterpret_cast<ly_u32x2*>(xs + RM_MAXPOS * RM_PS + 8 * lane) = (ly_u32x2){0x3F803F80u, 0x3F803F80u};

  f32x16 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  f32x16 zero16;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero16[r] = 0.f;

  // Software pipeline.  hipcc sinks LDS reads to their use, and two waves per SIMD do not hide a ~150-cycle LDS round trip per MFMA (measured:
  // 12 k cycles per chunk against 1.9 k of MFMA issue) — so every phase ISSUES the fragment reads of the next phase first and is fenced:
  //   unit j, phase 1:  [reads: main fragments of unit j]                      generate MFMAs of unit j (fragments + patches read in unit j-1)
  //           boundary (j odd): every read of this stage's ring buffer has been issued -> barrier (drains them), copies of the stage after next
  //           phase 2:  [reads: generate fragments + patches of unit j+1]      relu * ca * rfa -> bf16 (VALU), main MFMAs of unit j
  rm_stage_store(St, xs, lane);
  rm_wait_vm<0>();
  __syncthreads();                                        // stage 0, ess, the waves' own ca / ones rows: visible
  RM_T(0);                                                // prologue
  if (NST > 1) issue(1);
  rm_stage_load(St, x, NCH > 1 ? RM_CB : 0);
  bf16x8 G[RM_GF], Bp[3];
#pragma unroll
  for (int f = 0; f < RM_GF; ++f) G[f] = rm_wfrag(rl0, f);
#pragma unroll
  for (int st = 0; st < 3; ++st) Bp[st] = rm_patch_frag(xs, bo, st, 0);

  for (int ch = 0; ch < NCH; ++ch) {
    f32x16 d8 = zero16;
    const float* cac = cas + ch * RM_CB;
    bf16x8 Tf[MT];
    f32x4 c8[2];
#pragma unroll
    for (int j = 0; j < RM_UNITS; ++j) {
      const int q = j >> 1;
      const char* rl = (q & 1) ? rl1 : rl0;
      const int fb = (j & 1) * UF;
      // ---- phase 1 ----
      bf16x8 Mf[2 * MT];
#pragma unroll
      for (int f = 0; f < 2 * MT; ++f) Mf[f] = rm_wfrag(rl, fb + RM_GF + f);
      const f32x4 cj = *reinterpret_cast<const f32x4*>(cac + 4 * j);      // ca of the unit's 4 channels (a broadcast read: the address is wave-uniform)
      if (j == RM_UNITS - 1) {
#pragma unroll
        for (int f = 0; f < MT; ++f) Tf[f] = rm_wfrag(rl, 2 * UF + f);
#pragma unroll
        for (int g = 0; g < 2; ++g) c8[g] = *reinterpret_cast<const f32x4*>(cac + 8 * g + 4 * h);
      }
      f32x16 d = rm_mfma(G[0], Bp[0], zero16);
      d = rm_mfma(G[1], Bp[1], d);
      d = rm_mfma(G[2], Bp[2], d);
#pragma unroll
      for (int st = 0; st < 3; ++st) d8 = rm_mfma(G[3 + st], Bp[st], d8);
      __builtin_amdgcn_sched_barrier(0);
      RM_T(1);                                              // phase 1: main-fragment reads + generate MFMAs
      if (j & 1) {
        // stage boundary: behind the barrier this stage's buffer takes the copies of the stage after next; the next stage (copied a stage
        // ago) is published.  At j == 3 the patches of the chunk's last unit are in registers: the x tile takes the next chunk
        const int k = ch * 2 + q + 1;                       // the stage that must be visible from here on
        if (j == RM_UNITS - 1) rm_stage_store(St, xs, lane);
        RM_T(2);                                            // x tile store
        rm_wait_vm<0>();
        RM_T(3);                                            // wait for the weight copies
        if (!(dbg & 4)) __syncthreads();
        RM_T(4);                                            // barrier
        if (k + 1 < NST) issue(k + 1);
        if (j == RM_UNITS - 1) rm_stage_load(St, x, (dbg & 2) ? 0 : (ch + 2 < NCH ? (ch + 2) * RM_CB : 0));
        __builtin_amdgcn_sched_barrier(0);
        RM_T(5);                                            // issue copies + x prefetch
      }
      // ---- phase 2 ----
      {
        const int jn = (j + 1) & (RM_UNITS - 1);
        const char* rn = (((jn >> 1) & 1) ? rl1 : rl0) + (jn & 1) * UF * 1024;
#pragma unroll
        for (int f = 0; f < RM_GF; ++f) G[f] = rm_wfrag(rn, f);
#pragma unroll
        for (int st = 0; st < 3; ++st) Bp[st] = rm_patch_frag(xs, bo, st, jn);
      }
      // rows 8g + 4h + i of the tile = (channel 4j + g, tap 4h + i)
      ly_u32x4 gk[2];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        gk[g >> 1][2 * (g & 1)] = rm_post2(d[4 * g], d[4 * g + 1], rf[0] * cj[g]);
        gk[g >> 1][2 * (g & 1) + 1] = rm_post2(d[4 * g + 2], d[4 * g + 3], rf[1] * cj[g]);
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 bm = __builtin_bit_cast(bf16x8, gk[s2]);
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t] = rm_mfma(Mf[s2 * MT + t], bm, acc[t]);
      }
      __builtin_amdgcn_sched_barrier(0);
      RM_T(6);                                              // phase 2: next unit's reads, relu * ca * rfa, main MFMAs
    }
    // the tap-8 tile: rows 0 .. 15 = channel 8g + 4h + i of the chunk (g = 0, 1): ONE k-step of the main contraction
    {
      ly_u32x4 gk;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const f32x4 c4 = c8[g] * rf8;
        gk[2 * g] = rm_post2(d8[4 * g], d8[4 * g + 1], (f32x2){c4[0], c4[1]});
        gk[2 * g + 1] = rm_post2(d8[4 * g + 2], d8[4 * g + 3], (f32x2){c4[2], c4[3]});
      }
      const bf16x8 bm = __builtin_bit_cast(bf16x8, gk);
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t] = rm_mfma(Tf[t], bm, acc[t]);
      __builtin_amdgcn_sched_barrier(0);
      RM_T(7);                                              // tap-8 tile
    }
  }
  if constexpr (PROF) {
    if ((blockIdx.x & 15) == 0 && tid == 0)
    {
      for (int i = 0; i < 8; ++i) atomicAdd(&rm_prof[i], pacc[i]);
      atomicAdd(&rm_prof[8], __builtin_readcyclecounter() - pclk0);
      atomicAdd(&rm_prof[9], __builtin_amdgcn_s_memrealtime() - preal0);
    }
  }

  // ---- epilogue: conv.0 bias + conv.1 BatchNorm (folded) + ReLU.  The lane holds, per 32-channel tile, 4 x 4 consecutive channels of ITS pixel:
  // stored as they stand that is 4 MT stores of 8 bytes per lane into 32 different rows per instruction (the store tail of the MI355X guide,
  // issue-bound).  The tile goes through LDS instead — the ring is dead behind the last barrier — as [32 px][MT * 32 ch] bf16, 8-byte chunk q of
  // pixel row p at chunk q ^ (p & 15) (the 16 lanes of a store group hold 16 pixels and one q: 16 different bank pairs), and leaves as whole
  // pixel rows, 16 bytes per lane: MT * 2 instructions of 1 KiB
  __syncthreads();                                        // every wave is done with the ring
  {
    constexpr int ROWB = MT * 64;                         // bytes per pixel row
    char* const ot = rm_smem + wave * (32 * ROWB);
    const float* es = ess + 4 * h;
    const float* eb = ess + MT * 32 + 4 * h;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(es + 32 * t + 8 * g), sh = *reinterpret_cast<const f32x4*>(eb + 32 * t + 8 * g);
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(acc[t][4 * g + r] * sc[r] + sh[r], 0.f);
        const int q = 8 * t + 2 * g + h;                  // the lane's 8-byte chunk: channels 32 t + 8 g + 4 h ..
        *reinterpret_cast<bf16x4*>(ot + px * ROWB + ((q ^ (px & 15)) << 3)) = ly_cvtb4(v);
      }
    // rows back: lane -> (pixel 4 i + lane / 16 ... for MT = 4: 16 lanes per 256-byte row; MT = 2: 8 lanes per 128-byte row)
    constexpr int LPR = ROWB / 16;                        // lanes per pixel row
    constexpr int PPI = 64 / LPR;                         // pixel rows per instruction
    const int pr = lane / LPR, c16 = lane % LPR;
    __bf16* const ob = out + (long)by * MT * 32 + 8 * c16;
#pragma unroll
    for (int i = 0; i < 32 / PPI; ++i) {
      const int p = i * PPI + pr;
      const int py = p / P.TW, pxx = p - py * P.TW;
      const int oyy = T.oy0 + py, oxx = T.ox0 + pxx;
      // chunks 2 c16, 2 c16 + 1 of row p sit at (2 c16) ^ (p & 15) and its neighbour: one 16-byte slot, halves swapped for odd p
      const int slot = (2 * c16) ^ (p & 14);
      ly_u32x4 v = *reinterpret_cast<const ly_u32x4*>(ot + p * ROWB + (slot << 3));
      if (p & 1) v = (ly_u32x4){v[2], v[3], v[0], v[1]};
      if (T.valid && p < P.TH * P.TW && oyy < P.Ho && oxx < P.Wo)
        *reinterpret_cast<ly_u32x4*>(ob + (((long)T.n * P.Ho + oyy) * P.Wo + oxx) * P.ldo) = v;
    }
  }
}

static int rm_check(const char* who, int C, int s, int TH, int TW, int ldx, const void* x) {
  LY_CHECK(C > 0 && (C % 32) == 0, "%s: C=%d must be a multiple of 32", who, C);
  LY_CHECK(s == 1 || s == 2, "%s: stride %d is not built (1 or 2)", who, s);
  LY_CHECK(TH >= 1 && TW >= 1 && TH * TW <= 32, "%s: bad tile %dx%d (TH*TW <= 32)", who, TH, TW);
  LY_CHECK((s * (TH - 1) + 3) * (s * (TW - 1) + 3) <= RM_MAXPOS, "%s: the %dx%d tile reads more than %d input positions", who, TH, TW, RM_MAXPOS);
  LY_CHECK((ldx & 7) == 0 && ((uintptr_t)x & 15) == 0, "%s: x must be 16-byte aligned with a row stride that is a multiple of 8", who);
  return 0;
}

template <int MT>
static int rm_launch_fwd(const LyRfcbam3Params& P, hipStream_t st) {
  const int nct = (P.Wo + P.TW - 1) / P.TW, nrt = (P.Ho + P.TH - 1) / P.TH;
  const int gy = P.N / (32 * MT);
  const int lds_x = (int)((RM_XTILE + (size_t)P.C * 4 + 63) / 64 * 64);
  const size_t lds = (size_t)2 * (2 * (RM_GF + 2 * MT) + MT) * 1024 + 2 * MT * 32 * 4 + RM_WAVES * (size_t)lds_x;
  LY_CHECK(lds <= 160 * 1024, "rf3m_fwd: needs %zu B LDS", lds);
  // development builds (make CXXFLAGS+=-DLY_RM_DEVEL) read LY_RM_PROF / LY_RM_DBG / LY_RM_SDBG: the profiling kernel and the timing-only ablations
  // (wrong results by construction) are not reachable from the environment in the shipped library
#ifdef LY_RM_DEVEL
  static const bool prof = getenv("LY_RM_PROF") != nullptr;
  auto k = prof ? ly_rf3m_fwd_kernel<MT, true> : ly_rf3m_fwd_kernel<MT, false>;
#else
  auto k = ly_rf3m_fwd_kernel<MT, false>;
#endif
  static LyDevOnce once;
  if (once.need()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    LY_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  const long tiles = (long)P.n_img * nrt * nct;
  const long nb = (tiles + RM_WAVES - 1) / RM_WAVES * gy;
  LY_CHECK(nb < (1L << 31), "rf3m_fwd: grid too large");
#ifdef LY_RM_DEVEL
  static const int dbg = getenv("LY_RM_DBG") ? atoi(getenv("LY_RM_DBG")) : 0;          // development: 1 = no weight copies (wrong results, timing only)
#else
  const int dbg = 0;
#endif
  hipLaunchKernelGGL(k, dim3((unsigned)nb), dim3(RM_THREADS), lds, st, P, nct, nrt, gy, lds_x, dbg);
  LY_LAUNCH_CHECK();
  return 0;
}

// P as for ly_rfcbam3_fwd with: dtype LY_BF16, stats == NULL, linear == 0 (inference form), P.wg ignored, TH*TW <= 32, and
// P.wp = the weight stream of pack.rf3m_stream (N % 64 == 0: blocks of 128 output channels, or 64 when N % 128 != 0)
extern "C" int ly_rf3m_fwd(const LyRfcbam3Params* p, void* stream) {
  LY_CHECK(p, "rf3m_fwd: null params");
  const LyRfcbam3Params& P = *p;
  LY_CHECK(P.dtype == LY_BF16, "rf3m_fwd: bf16 storage only (dtype %d)", P.dtype);
  LY_CHECK(P.x && P.ca && P.rfa && P.wp && P.e_scale && P.e_shift && P.out && !P.stats && !P.linear, "rf3m_fwd: inference form only (out, no stats, relu)");
  if (rm_check("rf3m_fwd", P.C, P.s, P.TH, P.TW, P.ldx, P.x)) return -1;
  LY_CHECK(P.N > 0 && (P.N % 64) == 0 && (P.ldo & 7) == 0 && ((uintptr_t)P.out & 15) == 0, "rf3m_fwd: N=%d must be a multiple of 64 (out 16-byte aligned, ldo a multiple of 8)", P.N);
  LY_CHECK((long)P.n_img * P.H * P.W * P.ldx < (1L << 31), "rf3m_fwd: input exceeds the 31-bit offsets of the staging plan");
  LY_CHECK(P.Ho == (P.H + 2 - 3) / P.s + 1 && P.Wo == (P.W + 2 - 3) / P.s + 1, "rf3m_fwd: inconsistent output size");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  return (P.N % 128) == 0 ? rm_launch_fwd<4>(P, st) : rm_launch_fwd<2>(P, st);
}

// ---------------------------------------------------------------------------------------------------
// statistics pass: mm[n, 3oy + t/3, 3ox + t%3] = [max_c, mean_c] of G, and the SE pooling partials part[n][tile][C] (the sum of the tile's OWN
// input positions per channel: every input position belongs to exactly one tile).  Same generate products as the contraction kernel; the
// channel reductions run in the accumulator layout: a lane holds, for ITS pixel, 4 channels x 4 taps per tile — max / sum over the channel
// index are register operations, the two half waves meet once per tile walk for tap 8.  The pooling sums are one more product: spare rows of
// the tap-8 tile hold ones over the patch slots a pixel owns (the statistics stream is packed with pool_stride), one 32-lane tree per chunk.
// LDS: ring [2][24 KiB] | 8 x x tile
// ---------------------------------------------------------------------------------------------------
template <int DBG>
__global__ __launch_bounds__(RM_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void ly_rf3m_stats_kernel(const __bf16* __restrict__ x, int ldx, int n_img, int H,
                                                                                                              int W, int C, int Ho, int Wo, int s, int TH, int TW,
                                                                                                              int nct, int nrt, const void* __restrict__ wst,
                                                                                                              float* __restrict__ mm, float* __restrict__ part) {
  constexpr int SF = RM_UNITS * RM_GF;                 // fragments per stage = per chunk
  constexpr int RB = SF * 1024;
  extern __shared__ __attribute__((aligned(16))) char rm_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5;
  const long total = (long)n_img * nrt * nct;
  const long wt = (long)ly_xcd_remap((int)blockIdx.x, (int)gridDim.x) * RM_WAVES + wave;
  const RmTile T = rm_tile(wt, total, nct, nrt, TH, TW);
  const int NCH = C / RM_CB;
  char* const ring = rm_smem;
  char* const xs = rm_smem + 2 * RB + wave * RM_XTILE;
  const char* const rl0 = ring + lane * 16;
  const char* const rl1 = rl0 + RB;
  const unsigned ring_lds = rm_lds_addr(ring);
  const char* const wsrc = reinterpret_cast<const char*>(wst) + lane * 16;
  auto issue = [&](int sidx) {                          // stage = chunk sidx -> ring buffer sidx & 1
    const char* src = wsrc + (long)sidx * SF * 1024;
    const unsigned dst = ring_lds + (sidx & 1) * RB;
    for (int f = wave; f < SF; f += RM_WAVES) rm_dma16(src + f * 1024, dst + f * 1024);
  };
  issue(0);

  RmStage St;
  rm_stage_plan(St, lane, T, s, TH, TW, H, W, ldx);
  rm_stage_load(St, x, 0);
  int bo[3][2];
  rm_patch_offsets(bo, lane, s, TH, TW);
  if (lane < RM_PS / 8) *reinterpret_cast<ly_u32x2*>(xs + RM_MAXPOS * RM_PS + 8 * lane) = (ly_u32x2){0x3F803F80u, 0x3F803F80u};
  float mx[4] = {0.f, 0.f, 0.f, 0.f}, mx8 = 0.f;       // G >= 0: zero is the identity of the channel maximum
  f32x2 sm[2] = {{0.f, 0.f}, {0.f, 0.f}};
  float sm8 = 0.f;
  f32x16 zero16;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero16[r] = 0.f;

  // SE pooling (models/rfa.py:90): rows 16 .. 31 of the tap-8 tile carry ones over the patch slots a pixel OWNS (pack.rf3m_stream,
  // pool_stride), so d8[8 + k] is, per pixel lane, the sum of its own input positions for channel 4h + k (k < 4) / 8 + 4h + k - 4 of the
  // chunk: one masked 32-lane tree per chunk and the tile's row of `part` is complete.  (Until round 4 the sums were taken out of the
  // staging registers — bit extraction, selects and packed adds per 16-byte item: 14.5 of the kernel's 74 us at layer 17.)
  const bool lane_px = (lane & 31) < TH * TW;
  auto pool_store = [&](const f32x16& d8v, int cc) {
    if constexpr (DBG & 1) return;
    if (!part) return;
    // reduce-scatter butterfly over the 32 pixel lanes of a half wave: 4 + 2 + 1 + 1 + 1 cross-lane moves instead of 8 x 5, and only the first
    // four (across 16-lane rows) go through the LDS pipe (the plain tree of ds_bpermute cost as much as the staging-register arithmetic it
    // replaced); lane 4 j of a half ends up with value j
    float pv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) pv[k] = lane_px ? d8v[8 + k] : 0.f;
    float a4[4], b2[2];
    const bool u16 = lane & 16, u8 = lane & 8, u4 = lane & 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) a4[k] = (u16 ? pv[k + 4] : pv[k]) + ly_lane_xor<16>(u16 ? pv[k] : pv[k + 4]);
#pragma unroll
    for (int k = 0; k < 2; ++k) b2[k] = (u8 ? a4[k + 2] : a4[k]) + ly_lane_xor<8>(u8 ? a4[k] : a4[k + 2]);      // (inside a 16-lane row: DPP moves)
    float c1 = (u4 ? b2[1] : b2[0]) + ly_lane_xor<4>(u4 ? b2[0] : b2[1]);
    c1 += ly_lane_xor<2>(c1);
    c1 += ly_lane_xor<1>(c1);
    if ((lane & 3) == 0 && T.valid) {
      const int idx = ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);      // value index: channel 4h + idx (idx < 4) / 8 + 4h + idx - 4
      part[wt * C + cc * RM_CB + (idx < 4 ? 4 * h + idx : 4 + 4 * h + idx)] = c1;
    }
  };
  auto pool_and_store = [&](int) { rm_stage_store(St, xs, lane); };
  // software pipeline as in the contraction kernel: unit j issues the reads of unit j + 1 before its own MFMAs
  pool_and_store(0);
  rm_wait_vm<0>();
  __syncthreads();                                      // chunk 0's fragments: visible
  if (NCH > 1) issue(1);
  rm_stage_load(St, x, NCH > 1 ? RM_CB : 0);
  bf16x8 G[RM_GF], Bp[3];
#pragma unroll
  for (int f = 0; f < RM_GF; ++f) G[f] = rm_wfrag(rl0, f);
#pragma unroll
  for (int st = 0; st < 3; ++st) Bp[st] = rm_patch_frag(xs, bo, st, 0);

  for (int ch = 0; ch < NCH; ++ch) {
    f32x16 d8 = zero16;
#pragma unroll
    for (int j = 0; j < RM_UNITS; ++j) {
      bf16x8 Gc[RM_GF], Bc[3];
#pragma unroll
      for (int f = 0; f < RM_GF; ++f) Gc[f] = G[f];
#pragma unroll
      for (int st = 0; st < 3; ++st) Bc[st] = Bp[st];
      if (j == RM_UNITS - 1 && !(DBG & 8)) {
        // chunk boundary: the chunk's last patches and fragments are in registers.  The x tile takes the next chunk (whose pooling sums
        // leave from the staging registers); behind the barrier this chunk's ring buffer takes the chunk after next
        pool_and_store(ch + 1);
        rm_wait_vm<0>();
        __syncthreads();
        if (ch + 2 < NCH) issue(ch + 2);
        rm_stage_load(St, x, ch + 2 < NCH ? (ch + 2) * RM_CB : 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (!(DBG & 4)) {
        const int jn = (j + 1) & (RM_UNITS - 1);
        const char* rn = ((((j == RM_UNITS - 1) ? ch + 1 : ch) & 1) ? rl1 : rl0) + jn * RM_GF * 1024;
#pragma unroll
        for (int f = 0; f < RM_GF; ++f) G[f] = rm_wfrag(rn, f);
#pragma unroll
        for (int st = 0; st < 3; ++st) Bp[st] = rm_patch_frag(xs, bo, st, jn);
      }
      f32x16 d = rm_mfma(Gc[0], Bc[0], zero16);
      d = rm_mfma(Gc[1], Bc[1], d);
      d = rm_mfma(Gc[2], Bc[2], d);
#pragma unroll
      for (int st = 0; st < 3; ++st) d8 = rm_mfma(Gc[3 + st], Bc[st], d8);
      // rows 8g + 4h + i = (channel 4j + g, tap 4h + i): fold the 4 channels into the lane's running max / sum of taps 4h .. 4h+3
      if constexpr (DBG & 2) {
        mx[0] = fmaxf(mx[0], d[0]);
      } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float r0 = fmaxf(d[i], 0.f), r1 = fmaxf(d[4 + i], 0.f), r2 = fmaxf(d[8 + i], 0.f), r3 = fmaxf(d[12 + i], 0.f);
        mx[i] = fmaxf(fmaxf(mx[i], r0), fmaxf(fmaxf(r1, r2), r3));
        sm[i >> 1][i & 1] += (r0 + r1) + (r2 + r3);
      }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // tap 8: rows 0 .. 15 = the 16 channels of the chunk, 8 per lane
#pragma unroll
    for (int r = 0; r < 8; r += 4) {
      const float r0 = fmaxf(d8[r], 0.f), r1 = fmaxf(d8[r + 1], 0.f), r2 = fmaxf(d8[r + 2], 0.f), r3 = fmaxf(d8[r + 3], 0.f);
      mx8 = fmaxf(fmaxf(mx8, r0), fmaxf(fmaxf(r1, r2), r3));
      sm8 += (r0 + r1) + (r2 + r3);
    }
    pool_store(d8, ch);
  }
  mx8 = fmaxf(mx8, __shfl_xor(mx8, 32));
  sm8 += __shfl_xor(sm8, 32);
  const int px = lane & 31;
  const int ly = px / TW, lx = px - ly * TW;
  const int oy = T.oy0 + ly, ox = T.ox0 + lx;
  if (!(T.valid && px < TH * TW && oy < Ho && ox < Wo)) return;
  const float inv = 1.f / (float)C;
  float* mp = mm + (((long)T.n * 3 * Ho + 3 * oy) * (3 * Wo) + 3 * ox) * 2;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = 4 * h + i;
    *reinterpret_cast<f32x2*>(mp + ((t / 3) * (3 * Wo) + (t % 3)) * 2) = (f32x2){mx[i], sm[i >> 1][i & 1] * inv};
  }
  if (h == 0) *reinterpret_cast<f32x2*>(mp + (2 * (3 * Wo) + 2) * 2) = (f32x2){mx8, sm8 * inv};
}

// wst = pack.rf3m_stream(generate.0.weight, bn_scale, bn_shift) (the statistics stream); mm [n, 3Ho, 3Wo, 2]; part NULL or [n][tiles][C]
// (slices must equal the tile count ceil(Ho/TH) * ceil(Wo/TW)).  bf16 storage, folded (inference) BatchNorm.
extern "C" int ly_rf3m_stats(const void* x, int ldx, int n_img, int H, int W, int C, int s, const void* wst, int TH, int TW, float* mm, float* part,
                             int slices, void* stream) {
  LY_CHECK(x && wst && mm && n_img > 0, "rf3m_stats: null pointer");
  if (rm_check("rf3m_stats", C, s, TH, TW, ldx, x)) return -1;
  LY_CHECK((long)n_img * H * W * ldx < (1L << 31), "rf3m_stats: input exceeds the 31-bit offsets of the staging plan");
  const int Ho = (H + 2 - 3) / s + 1, Wo = (W + 2 - 3) / s + 1;
  const int nct = (Wo + TW - 1) / TW, nrt = (Ho + TH - 1) / TH;
  LY_CHECK(!part || slices == nct * nrt, "rf3m_stats: the pooling partials are one row per tile: slices must be %d", nct * nrt);
  const long tiles = (long)n_img * nrt * nct;
  const size_t lds = (size_t)2 * RM_UNITS * RM_GF * 1024 + RM_WAVES * (size_t)RM_XTILE;
  auto k = ly_rf3m_stats_kernel<0>;
  static LyDevOnce once;
  bool need = once.need();
#ifdef LY_RM_DEVEL
  static const int sdbg = getenv("LY_RM_SDBG") ? atoi(getenv("LY_RM_SDBG")) : 0;     // development: ablations of the loop (timing only)
  switch (sdbg) {
    case 1: k = ly_rf3m_stats_kernel<1>; break;
    case 2: k = ly_rf3m_stats_kernel<2>; break;
    case 4: k = ly_rf3m_stats_kernel<4>; break;
    case 8: k = ly_rf3m_stats_kernel<8>; break;
    case 15: k = ly_rf3m_stats_kernel<15>; break;
    default: break;
  }
  if (sdbg) need = true;
#endif
  if (need) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    LY_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(k, dim3((unsigned)((tiles + RM_WAVES - 1) / RM_WAVES)), dim3(RM_THREADS), lds, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const __bf16*>(x), ldx, n_img, H, W, C, Ho, Wo, s, TH, TW, nct, nrt, wst, mm, part);
  LY_LAUNCH_CHECK();
  return 0;
}
