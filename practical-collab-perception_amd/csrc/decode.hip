// a8 -- CenterHead decode on the device: sigmoid, exact top-K (radix select + bitonic sort of the K survivors), gather,
// exp / atan2 box assembly, range + score mask, order-preserving compaction.
//
// Replaces pcdet/models/dense_heads/center_head.py:302-333 and pcdet/models/model_utils/centernet_utils.py:127-214
// (two torch.topk, five gathers, atan2/exp/sigmoid launches, boolean-mask indexing with host syncs).
// One workgroup (1024 lanes) per frame; everything stays in registers/LDS; no host round trip.
//
// Top-K semantics: K largest sigmoid(hm) over (class, cell), descending, ties broken by the lower flat index
// (torch.topk leaves tie order unspecified).  Per-class top-K followed by a cross-class top-K
// (centernet_utils.py:137-143) selects the same set as one top-K over all (class, cell) pairs.
#include "pcp_common.h"

#pragma clang fp contract(off)

namespace {

typedef unsigned long long u64;
constexpr int DEC_THREADS = 1024;
constexpr int DEC_IPT = 16;                     // items per thread -> up to 16384 (class, cell) pairs
constexpr int DEC_CAP = DEC_THREADS * DEC_IPT;
constexpr int DEC_KMAX = 1024;
constexpr int HIST_BINS = 2048;

struct DecParams {
  pcp_decode_t d;
  const float *head;
  float *boxes, *scores;
  int *labels, *cell, *count;
};

__device__ __forceinline__ int wave_incl_scan(int v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) {
    int u = __shfl_up(v, s, 64);
    if (lane >= s) v += u;
  }
  return v;
}

// exclusive scan over the 1024 threads of the block; *total receives the block sum. scratch: >= 17 ints of LDS.
__device__ __forceinline__ int block_excl_scan(int v, int *scratch, int *total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = wave_incl_scan(v);
  if (lane == 63) scratch[wave] = incl;
  __syncthreads();
  if (threadIdx.x < 64) {
    int w = threadIdx.x < DEC_THREADS / 64 ? scratch[threadIdx.x] : 0;
    int wi = wave_incl_scan(w);
    if (threadIdx.x < DEC_THREADS / 64) scratch[threadIdx.x] = wi - w;
    if (threadIdx.x == DEC_THREADS / 64 - 1) scratch[DEC_THREADS / 64] = wi;
  }
  __syncthreads();
  int res = scratch[wave] + incl - v;
  *total = scratch[DEC_THREADS / 64];
  __syncthreads();
  return res;
}

__global__ __launch_bounds__(DEC_THREADS) void k_decode(DecParams p) {
  __shared__ int hist[HIST_BINS];
  __shared__ u64 cand[DEC_KMAX];
  __shared__ int scratch[32];
  __shared__ int sel_digit, sel_above, cand_count;

  const pcp_decode_t &d = p.d;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int HW = d.h * d.w;
  const int total = HW * d.num_class;
  const int K = min(d.k, total);
  const float *hb = p.head + (long long)b * HW * d.ld;

  // ---- scores: item i of thread t is flat index i*1024 + t (coalesced across the wave) -----------------------------------
  unsigned key[DEC_IPT];
#pragma unroll
  for (int i = 0; i < DEC_IPT; i++) {
    int idx = i * DEC_THREADS + tid;
    key[i] = 0u;
    if (idx < total) {
      int cls = idx / HW, cellid = idx % HW;
      float hm = hb[(long long)cellid * d.ld + d.ch_hm + cls];
      // activated = 1: the caller already applied sigmoid (centernet_utils.decode_bbox_from_heatmap's convention): the value IS the score
      float s = d.activated ? fminf(fmaxf(hm, 0.0f), 1.0f) : 1.0f / (1.0f + expf(-hm));
      key[i] = __float_as_uint(s) + 1u;        // s in [0,1]: bits are monotone; +1 keeps 0 for "absent"
    }
  }

  // ---- radix select of the K-th largest key (3 digit passes: 11 + 11 + 10 bits) ------------------------------------------
  unsigned prefix = 0u, pmask = 0u;
  int need = K;
  const int shifts[3] = {21, 10, 0};
  const int nbits[3] = {11, 11, 10};
  for (int pass = 0; pass < 3; pass++) {
    const int sh = shifts[pass];
    const unsigned dm = (1u << nbits[pass]) - 1u;
    for (int i = tid; i < HIST_BINS; i += DEC_THREADS) hist[i] = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < DEC_IPT; i++)
      if (key[i] != 0u && (key[i] & pmask) == prefix) atomicAdd(&hist[(key[i] >> sh) & dm], 1);
    __syncthreads();
    if (tid < 64) {
      // lane l owns bins [2047-32l-31, 2047-32l] walking downwards
      int top = HIST_BINS - 1 - 32 * tid;
      int s = 0;
      for (int q = 0; q < 32; q++) s += hist[top - q];
      int incl = wave_incl_scan(s);
      int before = incl - s;                    // elements in strictly higher bins owned by lower lanes
      bool mine = (before < need) && (incl >= need);
      if (mine) {
        int run = before;
        for (int q = 0; q < 32; q++) {
          int c = hist[top - q];
          if (run + c >= need) {
            sel_digit = top - q;
            sel_above = run;
            break;
          }
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
  const unsigned kth = prefix;      // exact key of the K-th largest; `need` of the elements equal to it are taken

  // ---- selection; ties at the threshold go to the lowest flat indices ----------------------------------------------------
  int eq_local = 0;
#pragma unroll
  for (int i = 0; i < DEC_IPT; i++) eq_local += (key[i] == kth && kth != 0u) ? 1 : 0;
  int eq_total;
  block_excl_scan(eq_local, scratch, &eq_total);
  if (tid == 0) cand_count = 0;
  for (int i = tid; i < DEC_KMAX; i += DEC_THREADS) cand[i] = 0ULL;
  __syncthreads();
  int eq_before_item = 0;           // equals in items < i (all threads)
#pragma unroll
  for (int i = 0; i < DEC_IPT; i++) {
    bool is_eq = (key[i] == kth) && kth != 0u;
    bool take = key[i] > kth;
    if (eq_total == need) {
      take = take || is_eq;
    } else {                        // rare: more equals than needed -> rank them in index order (item-major, thread-minor)
      int item_total;
      int rank_in_item = block_excl_scan(is_eq ? 1 : 0, scratch, &item_total);
      if (is_eq && eq_before_item + rank_in_item < need) take = true;
      eq_before_item += item_total;
    }
    if (take) {
      int pos = atomicAdd(&cand_count, 1);
      int idx = i * DEC_THREADS + tid;
      if (pos < DEC_KMAX) cand[pos] = ((u64)key[i] << 32) | (u64)(0xffffffffu - (unsigned)idx);
    }
  }
  __syncthreads();

  // ---- bitonic sort of the K survivors, descending -----------------------------------------------------------------------
  int cap = 64;
  while (cap < K) cap <<= 1;
  for (int k2 = 2; k2 <= cap; k2 <<= 1)
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      if (tid < cap / 2) {
        int i = ((tid & ~(j - 1)) << 1) | (tid & (j - 1));
        int q = i | j;
        bool desc = (i & k2) == 0;
        u64 a = cand[i], c = cand[q];
        if ((a < c) == desc) { cand[i] = c; cand[q] = a; }
      }
      __syncthreads();
    }

  // ---- decode + mask + ordered compaction ----------------------------------------------------------------------------------
  float box[7];
  float score = 0.f;
  int flat = 0;
  bool ok = false;
  if (tid < K) {
    u64 kk = cand[tid];
    flat = (int)(0xffffffffu - (unsigned)(kk & 0xffffffffULL));
    score = __uint_as_float((unsigned)(kk >> 32) - 1u);
    int cellid = flat % HW;
    const float *px = hb + (long long)cellid * d.ld;
    float xs = (float)(cellid % d.w), ys = (float)(cellid / d.w);
    xs = xs + px[d.ch_center];
    ys = ys + px[d.ch_center + 1];
    // ((xs * stride) * voxel) + min, one rounding each (centernet_utils.py:173-174)
    box[0] = xs * d.stride * d.voxel_x + d.min_x;
    box[1] = ys * d.stride * d.voxel_y + d.min_y;
    box[2] = px[d.ch_z];
    box[3] = d.activated ? px[d.ch_dim] : expf(px[d.ch_dim]);
    box[4] = d.activated ? px[d.ch_dim + 1] : expf(px[d.ch_dim + 1]);
    box[5] = d.activated ? px[d.ch_dim + 2] : expf(px[d.ch_dim + 2]);
    box[6] = atan2f(px[d.ch_rot + 1], px[d.ch_rot]);
    ok = box[0] >= d.limit[0] && box[1] >= d.limit[1] && box[2] >= d.limit[2] && box[0] <= d.limit[3] &&
         box[1] <= d.limit[4] && box[2] <= d.limit[5];
    if (d.use_score_thresh) ok = ok && (score > d.score_thresh);
  }
  int n_ok;
  int pos = block_excl_scan(ok ? 1 : 0, scratch, &n_ok);
  if (ok) {
    long long o = (long long)b * d.k + pos;
#pragma unroll
    for (int c = 0; c < 7; c++) p.boxes[o * 7 + c] = box[c];
    p.scores[o] = score;
    if (p.labels) p.labels[o] = flat / HW;
    if (p.cell) p.cell[o] = flat % HW;
  }
  if (tid == 0) p.count[b] = n_ok;
}


// ---- a8 / a9 tail: the per-frame "boxes[keep], scores[keep], mapping[labels[keep]] + 1, cat over heads" of
//      center_head.py:335-357 as ONE launch for all frames and heads -------------------------------------------------------------
struct GatherParams {
  pcp_det_head_t heads[PCP_DET_MAX_HEADS];
  int n_heads, batch, out_max;
  float *out_boxes;
  float *out_scores;
  long long *out_labels;
  int *out_count;
};

__global__ __launch_bounds__(256) void k_gather_detections(GatherParams p) {
  const int b = blockIdx.x;
  int base = 0;
  for (int hi = 0; hi < p.n_heads; hi++) {
    const pcp_det_head_t &h = p.heads[hi];
    int cnt = h.keep_count[b];
    cnt = cnt < h.keep_max ? cnt : h.keep_max;
    if (base + cnt > p.out_max) cnt = p.out_max - base;
    for (int i = threadIdx.x; i < cnt; i += 256) {
      const int src = h.keep[(long long)b * h.keep_max + i];
      const long long so = (long long)b * h.k + src, dq = (long long)b * p.out_max + base + i;
#pragma unroll
      for (int c = 0; c < 7; c++) p.out_boxes[dq * 7 + c] = h.boxes[so * 7 + c];
      p.out_scores[dq] = h.scores[so];
      const int lab = h.labels ? h.labels[so] : 0;
      p.out_labels[dq] = (long long)(h.class_map ? h.class_map[lab] : lab) + 1;
    }
    base += cnt;
  }
  if (threadIdx.x == 0) p.out_count[b] = base;
}

}  // namespace

extern "C" size_t pcp_decode_workspace_bytes(const pcp_decode_t *desc) {
  (void)desc;
  return 256;   // everything lives in LDS; kept for ABI stability
}

extern "C" int pcp_centerhead_decode(const pcp_decode_t *desc, const float *head, void *workspace, size_t workspace_bytes,
                                     float *boxes, float *scores, int32_t *labels, int32_t *cell, int32_t *count,
                                     void *stream_) {
  (void)workspace;
  (void)workspace_bytes;
  if (!desc || !head || !boxes || !scores || !count) return PCP_ERR_ARG;
  if (desc->batch <= 0 || desc->h <= 0 || desc->w <= 0 || desc->k <= 0 || desc->num_class <= 0) return PCP_ERR_ARG;
  if (desc->k > DEC_KMAX) return PCP_ERR_UNSUPPORTED;
  if ((long long)desc->h * desc->w * desc->num_class > DEC_CAP) return PCP_ERR_UNSUPPORTED;
  DecParams p;
  p.d = *desc;
  p.head = head;
  p.boxes = boxes; p.scores = scores; p.labels = labels; p.cell = cell; p.count = count;
  hipLaunchKernelGGL(k_decode, dim3(desc->batch), dim3(DEC_THREADS), 0, (hipStream_t)stream_, p);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

extern "C" int pcp_gather_detections(const pcp_det_head_t *heads, int32_t n_heads, int32_t batch, int32_t out_max, float *out_boxes,
                                     float *out_scores, int64_t *out_labels, int32_t *out_count, void *stream_) {
  if (!heads || n_heads <= 0 || n_heads > PCP_DET_MAX_HEADS || batch <= 0 || out_max <= 0) return PCP_ERR_ARG;
  if (!out_boxes || !out_scores || !out_labels || !out_count) return PCP_ERR_ARG;
  GatherParams p;
  for (int i = 0; i < n_heads; i++) {
    if (!heads[i].boxes || !heads[i].scores || !heads[i].keep || !heads[i].keep_count || heads[i].k <= 0 || heads[i].keep_max <= 0)
      return PCP_ERR_ARG;
    p.heads[i] = heads[i];
  }
  p.n_heads = n_heads; p.batch = batch; p.out_max = out_max;
  p.out_boxes = out_boxes; p.out_scores = out_scores; p.out_labels = reinterpret_cast<long long *>(out_labels); p.out_count = out_count;
  hipLaunchKernelGGL(k_gather_detections, dim3(batch), dim3(256), 0, (hipStream_t)stream_, p);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}
