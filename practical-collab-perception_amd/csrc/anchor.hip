// SURVEY 8(f) row 3 -- AnchorHeadSingle box decoding and candidate selection on the device.
//
// Replaces pcdet/models/dense_heads/anchor_head_template.py:225-272 (generate_predicted_boxes: anchors.repeat(B), ResidualCoder.decode_torch
// of pcdet/utils/box_coder_utils.py:46-78, direction-classifier correction with common_utils.limit_period) and the class-agnostic branch
// of Detector3DTemplate.post_processing (detector3d_template.py:262-326: sigmoid, max over classes, score mask, torch.topk) up to the
// NMS call; the NMS itself is pcp_nms_rotated.  Candidates leave this file already sorted by descending score (ties: lower anchor
// index), gathered as (B, K, 7) boxes, so the NMS kernel skips its own sort.
// pcp_anchor_decode is a pure stream (reads the head map once, writes boxes / logits / score keys); pcp_topk_boxes is one workgroup
// per frame: 3-pass radix select over the score keys (L2 resident) + bitonic sort of the K survivors in LDS.
#include "pcp_common.h"

#pragma clang fp contract(off)

namespace {

typedef unsigned long long u64;

__global__ __launch_bounds__(256) void k_anchor_decode(pcp_anchor_t d, const float *__restrict__ head, const float *__restrict__ anchors,
                                                      float *__restrict__ boxes, float *__restrict__ cls, unsigned *__restrict__ keys,
                                                      int *__restrict__ labels) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long per_frame = (long long)d.h * d.w * d.anchors_per_loc;
  if (t >= per_frame * d.batch) return;
  const int b = (int)(t / per_frame);
  const long long i = t % per_frame;
  const int a = (int)(i % d.anchors_per_loc);
  const long long loc = i / d.anchors_per_loc;
  const float *px = head + ((long long)b * d.h * d.w + loc) * d.ld;
  const float *an = anchors + i * 7;
  const float *e = px + d.ch_box + a * 7;
  const float xa = an[0], ya = an[1], za = an[2], dxa = an[3], dya = an[4], dza = an[5], ra = an[6];
  const float diagonal = sqrtf(dxa * dxa + dya * dya);
  float bx[7];
  bx[0] = e[0] * diagonal + xa;
  bx[1] = e[1] * diagonal + ya;
  bx[2] = e[2] * dza + za;
  bx[3] = expf(e[3]) * dxa;
  bx[4] = expf(e[4]) * dya;
  bx[5] = expf(e[5]) * dza;
  float rg = e[6] + ra;
  if (d.num_dir_bins > 0) {
    const float *dp = px + d.ch_dir + a * d.num_dir_bins;
    int lab = 0;
    float best = dp[0];
    for (int k = 1; k < d.num_dir_bins; ++k)
      if (dp[k] > best) { best = dp[k]; lab = k; }
    const float period = d.dir_period;
    const float val = rg - d.dir_offset;
    const float dir_rot = val - floorf(val / period + d.dir_limit_offset) * period;      // common_utils.limit_period
    rg = dir_rot + d.dir_offset + period * (float)lab;
  }
  bx[6] = rg;
  float *ob = boxes + t * 7;
#pragma unroll
  for (int k = 0; k < 7; ++k) ob[k] = bx[k];
  const float *cp = px + d.ch_cls + a * d.num_class;
  float mx = cp[0];
  int lab = 0;
  for (int c = 0; c < d.num_class; ++c) {
    const float v = cp[c];
    cls[t * d.num_class + c] = v;
    if (v > mx) { mx = v; lab = c; }
  }
  const float s = 1.0f / (1.0f + expf(-mx));
  keys[t] = (!d.use_score_thresh || s >= d.score_thresh) ? __float_as_uint(s) + 1u : 0u;     // >= : model_nms_utils.py:9
  labels[t] = lab;
}

constexpr int TK_THREADS = 1024;
constexpr int TK_KMAX = 4096;
constexpr int TK_BINS = 2048;

__device__ __forceinline__ int wave_incl_scan_i(int v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) {
    const int u = __shfl_up(v, s, 64);
    if (lane >= s) v += u;
  }
  return v;
}

__device__ __forceinline__ int block_excl_scan_i(int v, int *scratch, int *total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int incl = wave_incl_scan_i(v);
  if (lane == 63) scratch[wave] = incl;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int w = threadIdx.x < TK_THREADS / 64 ? scratch[threadIdx.x] : 0;
    const int wi = wave_incl_scan_i(w);
    if (threadIdx.x < TK_THREADS / 64) scratch[threadIdx.x] = wi - w;
    if (threadIdx.x == TK_THREADS / 64 - 1) scratch[TK_THREADS / 64] = wi;
  }
  __syncthreads();
  const int res = scratch[wave] + incl - v;
  *total = scratch[TK_THREADS / 64];
  __syncthreads();
  return res;
}

__global__ __launch_bounds__(TK_THREADS) void k_topk_boxes(const unsigned *__restrict__ keys, const int *__restrict__ labels,
                                                          const float *__restrict__ boxes, long long n, int kmax, float *__restrict__ out_boxes,
                                                          float *__restrict__ out_scores, int *__restrict__ out_labels, int *__restrict__ out_idx,
                                                          int *__restrict__ count) {
  __shared__ int hist[TK_BINS];
  __shared__ u64 cand[TK_KMAX];
  __shared__ int scratch[32];
  __shared__ int sel_digit, sel_above, cand_count, n_valid;
  const int b = blockIdx.x, tid = threadIdx.x;
  const unsigned *kb = keys + (long long)b * n;
  // how many candidates pass the score mask: K = min(kmax, n_valid)  (torch.topk(k = min(NMS_PRE_MAXSIZE, n)))
  int local = 0;
  for (long long i = tid; i < n; i += TK_THREADS) local += kb[i] != 0u;
  int tot;
  block_excl_scan_i(local, scratch, &tot);
  if (tid == 0) n_valid = tot;
  __syncthreads();
  const int K = min(kmax, n_valid);
  if (K == 0) {
    if (tid == 0) count[b] = 0;
    return;
  }
  unsigned prefix = 0u, pmask = 0u;
  int need = K;
  const int shifts[3] = {21, 10, 0};
  const int nbits[3] = {11, 11, 10};
  for (int pass = 0; pass < 3; pass++) {
    const int sh = shifts[pass];
    const unsigned dm = (1u << nbits[pass]) - 1u;
    for (int i = tid; i < TK_BINS; i += TK_THREADS) hist[i] = 0;
    __syncthreads();
    for (long long i = tid; i < n; i += TK_THREADS) {
      const unsigned k = kb[i];
      if (k != 0u && (k & pmask) == prefix) atomicAdd(&hist[(k >> sh) & dm], 1);
    }
    __syncthreads();
    if (tid < 64) {
      const int top = TK_BINS - 1 - 32 * tid;
      int s = 0;
      for (int q = 0; q < 32; q++) s += hist[top - q];
      const int incl = wave_incl_scan_i(s);
      const int before = incl - s;
      if (before < need && incl >= need) {
        int run = before;
        for (int q = 0; q < 32; q++) {
          const int c = hist[top - q];
          if (run + c >= need) { sel_digit = top - q; sel_above = run; break; }
          run += c;
        }
      }
    }
    __syncthreads();
    prefix |= ((unsigned)sel_digit) << sh;
    pmask |= dm << sh;
    need -= sel_above;
    __syncthreads();
  }
  const unsigned kth = prefix;
  if (tid == 0) cand_count = 0;
  for (int i = tid; i < TK_KMAX; i += TK_THREADS) cand[i] = 0ULL;
  __syncthreads();
  // elements above the K-th key are all taken; of the elements EQUAL to it the first `need` in index order
  int eq_seen = 0;
  for (long long base = 0; base < n; base += TK_THREADS) {
    const long long i = base + tid;
    const unsigned k = i < n ? kb[i] : 0u;
    const bool is_eq = k == kth;
    int chunk_total;
    const int rank = block_excl_scan_i(is_eq ? 1 : 0, scratch, &chunk_total);
    const bool take = k > kth || (is_eq && eq_seen + rank < need);
    eq_seen += chunk_total;
    if (take) {
      const int pos = atomicAdd(&cand_count, 1);
      if (pos < TK_KMAX) cand[pos] = ((u64)k << 32) | (u64)(0xffffffffu - (unsigned)i);
    }
  }
  __syncthreads();
  int cap = 64;
  while (cap < K) cap <<= 1;
  for (int k2 = 2; k2 <= cap; k2 <<= 1)
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < cap / 2; t += TK_THREADS) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int q = i | j;
        const bool desc = (i & k2) == 0;
        const u64 a = cand[i], c = cand[q];
        if ((a < c) == desc) { cand[i] = c; cand[q] = a; }
      }
      __syncthreads();
    }
  for (int r = tid; r < K; r += TK_THREADS) {
    const u64 kk = cand[r];
    const unsigned idx = 0xffffffffu - (unsigned)(kk & 0xffffffffULL);
    const long long o = (long long)b * kmax + r;
    const float *src = boxes + ((long long)b * n + idx) * 7;
#pragma unroll
    for (int c = 0; c < 7; ++c) out_boxes[o * 7 + c] = src[c];
    out_scores[o] = __uint_as_float((unsigned)(kk >> 32) - 1u);
    out_labels[o] = labels[(long long)b * n + idx];
    out_idx[o] = (int)idx;
  }
  if (tid == 0) count[b] = K;
}

}  // namespace

extern "C" {

int pcp_anchor_decode(const pcp_anchor_t *d, const float *head, const float *anchors, float *boxes, float *cls_logits, uint32_t *score_keys,
                      int32_t *labels, void *stream) {
  if (!d || !head || !anchors || !boxes || !cls_logits || !score_keys || !labels) return PCP_ERR_ARG;
  if (d->batch <= 0 || d->h <= 0 || d->w <= 0 || d->anchors_per_loc <= 0 || d->num_class <= 0 || d->num_dir_bins < 0) return PCP_ERR_ARG;
  const long long total = (long long)d->batch * d->h * d->w * d->anchors_per_loc;
  hipLaunchKernelGGL(k_anchor_decode, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *d, head, anchors, boxes,
                     cls_logits, score_keys, labels);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

int pcp_topk_boxes(const uint32_t *score_keys, const int32_t *labels, const float *boxes, int32_t batch, int64_t n, int32_t k,
                   float *out_boxes, float *out_scores, int32_t *out_labels, int32_t *out_index, int32_t *count, void *stream) {
  if (!score_keys || !labels || !boxes || !out_boxes || !out_scores || !out_labels || !out_index || !count) return PCP_ERR_ARG;
  if (batch <= 0 || n <= 0 || k <= 0) return PCP_ERR_ARG;
  if (k > TK_KMAX) return PCP_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_topk_boxes, dim3(batch), dim3(TK_THREADS), 0, (hipStream_t)stream, score_keys, labels, boxes, (long long)n, k, out_boxes,
                     out_scores, out_labels, out_index, count);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

}  // extern "C"
