// Eval tail on the device: the reference's `non_max_suppression` (utils/general.py:884-994) for the options the detector's
// own callers use — best class per box, optional class filter, optional class-agnostic mode, max_det — without a host round trip per image.
//
//   ly_nms_candidates   per (image, box): obj > conf_thres, conf = obj * best class confidence (first maximum), conf > conf_thres, class
//                       allowed -> det row (x1, y1, x2, y2, conf, cls) and its score; rejected boxes get score -1          (:914-960)
//   (sort)              the host mirror sorts the scores of every image descending, stable (torch.sort on the device)      (:970)
//   ly_nms_greedy       one block per image walks the sorted candidates: the first box still alive is kept, every later alive box whose
//                       IoU with it exceeds iou_thres is dropped (boxes offset by cls * max_wh unless agnostic) — torchvision.ops.nms'
//                       contract — until max_det boxes are kept or none is left                                              (:973-976)
// The alive set is an LDS bitset (max_nms = 30000 candidates = 938 words); per kept box the block's 256 threads share the later candidates.
#include "ly_common.hpp"
#include "ly_params.h"

static long ly_nms_blocks(long items) {
  long b = (items + LY_THREADS - 1) / LY_THREADS;
  return b < 1 ? 1 : (b > 8192 ? 8192 : b);
}

#define LY_NMS_MAXN 30016                 // >= max_nms (utils/general.py:919), multiple of 32
#define LY_NMS_WORDS (LY_NMS_MAXN / 32)

__global__ __launch_bounds__(LY_THREADS) void ly_nms_candidates_kernel(const float* __restrict__ pred, long total, int no, float conf_thres,
                                                                       unsigned long long class_mask, float* __restrict__ score,
                                                                       float* __restrict__ det) {
  const int nc = no - 5;
  for (long i = (long)blockIdx.x * LY_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * LY_THREADS) {
    const float* p = pred + i * no;
    const float obj = p[4];
    float sc = -1.f, best = 0.f;
    int cls = 0;
    if (obj > conf_thres) {
      best = p[5] * obj;
      for (int j = 1; j < nc; ++j) {
        const float v = p[5 + j] * obj;
        if (v > best) { best = v; cls = j; }
      }
      const bool allowed = class_mask == 0ull || (cls < 64 && ((class_mask >> cls) & 1ull));
      if (best > conf_thres && allowed) sc = best;
    }
    score[i] = sc;
    float* d = det + i * 6;
    const float cx = p[0], cy = p[1], w = p[2], h = p[3];
    d[0] = cx - w / 2; d[1] = cy - h / 2; d[2] = cx + w / 2; d[3] = cy + h / 2;      // xywh2xyxy (utils/general.py:760-767)
    d[4] = best; d[5] = (float)cls;
  }
}

// multi_label (utils/general.py:921, 951-955; val.py's setting for nc > 1): every (box, class) pair whose obj * class confidence exceeds the
// threshold is a candidate of its own — pair index box * nc + class, the row order of the reference's `(x[:, 5:] > conf_thres).nonzero()`
__global__ __launch_bounds__(LY_THREADS) void ly_nms_candidates_ml_kernel(const float* __restrict__ pred, long total_pairs, int no, float conf_thres,
                                                                          unsigned long long class_mask, float* __restrict__ score,
                                                                          float* __restrict__ det) {
  const int nc = no - 5;
  for (long i = (long)blockIdx.x * LY_THREADS + threadIdx.x; i < total_pairs; i += (long)gridDim.x * LY_THREADS) {
    const long box = i / nc;
    const int j = (int)(i - box * nc);
    const float* p = pred + box * no;
    const float obj = p[4];
    const float conf = p[5 + j] * obj;
    const bool allowed = class_mask == 0ull || (j < 64 && ((class_mask >> j) & 1ull));
    score[i] = (obj > conf_thres && conf > conf_thres && allowed) ? conf : -1.f;
    float* d = det + i * 6;
    const float cx = p[0], cy = p[1], w = p[2], h = p[3];
    d[0] = cx - w / 2; d[1] = cy - h / 2; d[2] = cx + w / 2; d[3] = cy + h / 2;
    d[4] = conf; d[5] = (float)j;
  }
}

__global__ __launch_bounds__(LY_THREADS) void ly_nms_greedy_kernel(const float* __restrict__ det, const long* __restrict__ order,
                                                                   const float* __restrict__ sorted_score, int N, float iou_thres, float max_wh,
                                                                   int max_det, int max_nms, int* __restrict__ keep, int* __restrict__ count) {
  __shared__ unsigned alive[LY_NMS_WORDS];
  __shared__ int s_n, s_next;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* dets = det + (long)b * N * 6;
  const long* ord = order + (long)b * N;
  const float* ss = sorted_score + (long)b * N;
  if (tid == 0) s_n = 0;
  __syncthreads();
  // candidates = the sorted prefix with score >= 0 (rejected boxes carry -1)
  int cnt = 0;
  for (int j = tid; j < N; j += LY_THREADS) cnt += ss[j] >= 0.f ? 1 : 0;
  atomicAdd(&s_n, cnt);
  __syncthreads();
  int n = s_n;
  n = n < max_nms ? n : max_nms;
  n = n < LY_NMS_MAXN ? n : LY_NMS_MAXN;
  for (int wi = tid; wi < LY_NMS_WORDS; wi += LY_THREADS) {
    const int lo = wi * 32;
    alive[wi] = lo + 32 <= n ? 0xffffffffu : (lo < n ? ((1u << (n - lo)) - 1u) : 0u);
  }
  __syncthreads();
  int kept = 0, cur = 0;
  while (kept < max_det && cur < n) {
    if (tid == 0) s_next = n;
    __syncthreads();
    // first alive candidate >= cur
    for (int wi = (cur >> 5) + tid; wi * 32 < n; wi += LY_THREADS) {
      unsigned m = alive[wi];
      if (wi == (cur >> 5)) m &= ~((1u << (cur & 31)) - 1u);
      if (m) { atomicMin(&s_next, wi * 32 + __ffs(m) - 1); break; }
    }
    __syncthreads();
    const int i = s_next;
    if (i >= n) break;
    const long oi = ord[i];
    if (tid == 0) keep[(long)b * max_det + kept] = (int)oi;
    const float* di = dets + oi * 6;
    const float off_i = di[5] * max_wh;
    const float ax1 = di[0] + off_i, ay1 = di[1] + off_i, ax2 = di[2] + off_i, ay2 = di[3] + off_i;
    const float area_i = (ax2 - ax1) * (ay2 - ay1);
    for (int j = i + 1 + tid; j < n; j += LY_THREADS) {
      if (!((alive[j >> 5] >> (j & 31)) & 1u)) continue;
      const float* dj = dets + ord[j] * 6;
      const float off_j = dj[5] * max_wh;
      const float bx1 = dj[0] + off_j, by1 = dj[1] + off_j, bx2 = dj[2] + off_j, by2 = dj[3] + off_j;
      const float iw = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1), 0.f), ih = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1), 0.f);
      const float inter = iw * ih;
      const float area_j = (bx2 - bx1) * (by2 - by1);
      const float iou = inter / (area_i + area_j - inter);
      if (iou > iou_thres) atomicAnd(&alive[j >> 5], ~(1u << (j & 31)));
    }
    ++kept;
    cur = i + 1;
    __syncthreads();
  }
  if (tid == 0) count[b] = kept;
}

extern "C" int ly_nms_candidates(const float* pred, int bs, int N, int no, float conf_thres, unsigned long long class_mask, float* score, float* det,
                                 void* stream) {
  LY_CHECK(pred && score && det && bs > 0 && N > 0 && no >= 6 && no - 5 <= 4096, "nms_candidates: bad arguments (no=%d)", no);
  const long total = (long)bs * N;
  hipLaunchKernelGGL(ly_nms_candidates_kernel, dim3((unsigned)ly_nms_blocks(total)), dim3(LY_THREADS), 0, reinterpret_cast<hipStream_t>(stream), pred, total, no,
                     conf_thres, class_mask, score, det);
  LY_LAUNCH_CHECK();
  return 0;
}

extern "C" int ly_nms_candidates_ml(const float* pred, int bs, int N, int no, float conf_thres, unsigned long long class_mask, float* score, float* det,
                                    void* stream) {
  LY_CHECK(pred && score && det && bs > 0 && N > 0 && no >= 7 && no - 5 <= 4096, "nms_candidates_ml: bad arguments (no=%d: multi_label needs nc > 1)", no);
  const long total = (long)bs * N * (no - 5);
  hipLaunchKernelGGL(ly_nms_candidates_ml_kernel, dim3((unsigned)ly_nms_blocks(total)), dim3(LY_THREADS), 0, reinterpret_cast<hipStream_t>(stream), pred, total,
                     no, conf_thres, class_mask, score, det);
  LY_LAUNCH_CHECK();
  return 0;
}

extern "C" int ly_nms_greedy(const float* det, const long* order, const float* sorted_score, int bs, int N, float iou_thres, float max_wh, int max_det,
                             int max_nms, int* keep, int* count, void* stream) {
  LY_CHECK(det && order && sorted_score && keep && count && bs > 0 && N > 0 && max_det > 0 && max_nms > 0, "nms_greedy: bad arguments");
  LY_CHECK(max_nms <= LY_NMS_MAXN, "nms_greedy: max_nms=%d exceeds the %d candidates the alive set holds", max_nms, LY_NMS_MAXN);
  hipLaunchKernelGGL(ly_nms_greedy_kernel, dim3((unsigned)bs), dim3(LY_THREADS), 0, reinterpret_cast<hipStream_t>(stream), det, order, sorted_score, N, iou_thres,
                     max_wh, max_det, max_nms, keep, count);
  LY_LAUNCH_CHECK();
  return 0;
}
