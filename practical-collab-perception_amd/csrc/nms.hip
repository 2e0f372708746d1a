// a9 -- rotated BEV IoU and NMS entirely on the device.
//
// Replaces pcdet/ops/iou3d_nms/src/iou3d_nms.cpp:52-136 + iou3d_nms_kernel.cu:236-311 (boxes_overlap_bev_gpu,
// boxes_iou_bev_gpu, nms_gpu) and the python-side score sort of pcdet/ops/iou3d_nms/iou3d_nms_utils.py:92-95.
// The reference computes the 64x64-tile bit mask on the GPU, then cudaMalloc/cudaMemcpy-D2H/cudaFree and a CPU greedy
// loop per call.  Here: (1) one-workgroup bitonic sort of (score, index) keys in LDS, (2) the
// upper-triangle mask, one wavefront (ballot) per 64-bit word with a conservative centre-distance reject, (3) a
// single-wavefront greedy sweep that resolves each 64-box diagonal block in registers (v_readlane broadcasts) and ORs
// the kept rows' mask words into the later column words, lane j owning word j.  No host round trip.
//
// The geometry follows the reference arithmetic operation for operation (fp32, one rounding per operation: FMA
// contraction is disabled for this file) so that the keep set matches away from the threshold.
#include "pcp_common.h"

#pragma clang fp contract(off)

namespace {

typedef unsigned long long u64;

constexpr float NMS_EPS = 1e-8f;
constexpr float NMS_MARGIN = 1e-2f;

struct P2 { float x, y; };

__device__ __forceinline__ float cross3(P2 a, P2 b, P2 o) { return (a.x - o.x) * (b.y - o.y) - (b.x - o.x) * (a.y - o.y); }

__device__ __forceinline__ bool rects_touch(P2 p1, P2 p2, P2 q1, P2 q2) {
  return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
         fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

__device__ __forceinline__ bool seg_hit(P2 p1, P2 p0, P2 q1, P2 q0, P2 &out) {
  if (!rects_touch(p0, p1, q0, q1)) return false;
  float s1 = cross3(q0, p1, p0);
  float s2 = cross3(p1, q1, p0);
  float s3 = cross3(p0, q1, q0);
  float s4 = cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
  float s5 = cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > NMS_EPS) {
    out.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    out.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    float D = a0 * b1 - a1 * b0;
    out.x = (b0 * c1 - b1 * c0) / D;
    out.y = (a1 * c0 - a0 * c1) / D;
  }
  return true;
}

struct Box {
  float x, y, dx, dy, ang;
};

__device__ __forceinline__ bool corner_in(const Box &b, P2 p) {
  float c = cosf(-b.ang), s = sinf(-b.ang);
  float rx = (p.x - b.x) * c + (p.y - b.y) * (-s);
  float ry = (p.x - b.x) * s + (p.y - b.y) * c;
  return fabsf(rx) < b.dx / 2 + NMS_MARGIN && fabsf(ry) < b.dy / 2 + NMS_MARGIN;
}

__device__ __forceinline__ void corners_of(const Box &b, P2 (&c)[5]) {
  float hx = b.dx / 2, hy = b.dy / 2;
  float x1 = b.x - hx, y1 = b.y - hy, x2 = b.x + hx, y2 = b.y + hy;
  float ca = cosf(b.ang), sa = sinf(b.ang);
  float rx[4] = {x1, x2, x2, x1}, ry[4] = {y1, y1, y2, y2};
#pragma unroll
  for (int k = 0; k < 4; k++) {
    float ddx = rx[k] - b.x, ddy = ry[k] - b.y;
    c[k].x = ddx * ca + ddy * (-sa) + b.x;
    c[k].y = ddx * sa + ddy * ca + b.y;
  }
  c[4] = c[0];
}

__device__ float overlap_area(const Box &a, const Box &b) {
  P2 ca[5], cb[5], poly[16], ctr;
  ctr.x = 0.f;
  ctr.y = 0.f;
  int cnt = 0;
  corners_of(a, ca);
  corners_of(b, cb);
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      P2 hit;
      if (seg_hit(ca[i + 1], ca[i], cb[j + 1], cb[j], hit)) {
        poly[cnt] = hit;
        ctr.x = ctr.x + hit.x;
        ctr.y = ctr.y + hit.y;
        cnt++;
      }
    }
  for (int k = 0; k < 4; k++) {
    if (corner_in(a, cb[k])) { ctr.x = ctr.x + cb[k].x; ctr.y = ctr.y + cb[k].y; poly[cnt++] = cb[k]; }
    if (corner_in(b, ca[k])) { ctr.x = ctr.x + ca[k].x; ctr.y = ctr.y + ca[k].y; poly[cnt++] = ca[k]; }
  }
  if (cnt < 3) return 0.f;          // reference: loops below do not run / a degenerate fan has zero area
  ctr.x /= cnt;
  ctr.y /= cnt;
  for (int j = 0; j < cnt - 1; j++)
    for (int i = 0; i < cnt - j - 1; i++) {
      float ai = atan2f(poly[i].y - ctr.y, poly[i].x - ctr.x);
      float an = atan2f(poly[i + 1].y - ctr.y, poly[i + 1].x - ctr.x);
      if (ai > an) { P2 t = poly[i]; poly[i] = poly[i + 1]; poly[i + 1] = t; }
    }
  float area = 0.f;
  for (int k = 0; k < cnt - 1; k++) {
    float ux = poly[k].x - poly[0].x, uy = poly[k].y - poly[0].y;
    float vx = poly[k + 1].x - poly[0].x, vy = poly[k + 1].y - poly[0].y;
    area += ux * vy - uy * vx;
  }
  return fabsf(area) / 2.0f;
}

__device__ __forceinline__ bool far_apart(const Box &a, const Box &b) {
  // conservative: circumscribed circles (+ margin) do not meet -> no crossing, no corner within the 1e-2 margin
  float ra = 0.5f * sqrtf(a.dx * a.dx + a.dy * a.dy), rb = 0.5f * sqrtf(b.dx * b.dx + b.dy * b.dy);
  float ddx = a.x - b.x, ddy = a.y - b.y, R = ra + rb + 0.1f;
  return ddx * ddx + ddy * ddy > R * R * 1.0001f;
}

__device__ __forceinline__ float iou_of(const Box &a, const Box &b) {
  float sa = a.dx * a.dy, sb = b.dx * b.dy;
  float ov = overlap_area(a, b);
  return ov / fmaxf(sa + sb - ov, NMS_EPS);
}

// axis-aligned IoU of nms_normal_gpu (iou3d_nms_kernel.cu:314-325: the heading is ignored), operation for operation
__device__ __forceinline__ float iou_normal_of(const Box &a, const Box &b) {
  float left = fmaxf(a.x - a.dx / 2, b.x - b.dx / 2), right = fminf(a.x + a.dx / 2, b.x + b.dx / 2);
  float top = fmaxf(a.y - a.dy / 2, b.y - b.dy / 2), bottom = fminf(a.y + a.dy / 2, b.y + b.dy / 2);
  float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
  float inter = width * height;
  float sa = a.dx * a.dy, sb = b.dx * b.dy;
  return inter / fmaxf(sa + sb - inter, NMS_EPS);
}

__device__ __forceinline__ Box load_box(const float *p) {
  Box b;
  b.x = p[0]; b.y = p[1]; b.dx = p[3]; b.dy = p[4]; b.ang = p[6];
  return b;
}

// ---- (1) sort -------------------------------------------------------------------------------------------------------
constexpr int SORT_CAP = 4096;
constexpr int SORT_THREADS = 1024;

__device__ __forceinline__ u64 score_key(float s, int idx) {
  // total order on floats (negative scores allowed), descending sort => larger key first; ties: lower index first
  unsigned u = __float_as_uint(s);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((u64)u << 32) | (u64)(0xffffffffu - (unsigned)idx);
}

__global__ __launch_bounds__(SORT_THREADS) void k_nms_sort(const float *__restrict__ boxes, const float *__restrict__ scores,
                                                           int n_max, const int *__restrict__ n_dev, int pre_max,
                                                           int *__restrict__ order, float *__restrict__ sorted_boxes,
                                                           int *__restrict__ n_eff_out) {
  __shared__ u64 keys[SORT_CAP];
  const int fb = blockIdx.x;                       // frame
  boxes += (size_t)fb * n_max * 7;
  scores += (size_t)fb * n_max;
  order += (size_t)fb * n_max;
  sorted_boxes += (size_t)fb * n_max * 8;
  n_eff_out += fb;
  int n = n_dev ? min(n_dev[fb], n_max) : n_max;
  n = max(n, 0);
  int cap = 64;
  while (cap < n) cap <<= 1;
  for (int i = threadIdx.x; i < cap; i += SORT_THREADS) keys[i] = i < n ? score_key(scores[i], i) : 0ULL;
  __syncthreads();
  for (int k = 2; k <= cap; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < cap / 2; t += SORT_THREADS) {
        int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        int q = i | j;
        bool desc = (i & k) == 0;
        u64 a = keys[i], b = keys[q];
        if ((a < b) == desc) { keys[i] = b; keys[q] = a; }
      }
      __syncthreads();
    }
  int n_eff = min(n, pre_max);
  for (int i = threadIdx.x; i < n_eff; i += SORT_THREADS) {
    int src = (int)(0xffffffffu - (unsigned)(keys[i] & 0xffffffffULL));
    order[i] = src;
#pragma unroll
    for (int c = 0; c < 7; c++) sorted_boxes[i * 8 + c] = boxes[src * 7 + c];
    sorted_boxes[i * 8 + 7] = 0.f;
  }
  if (threadIdx.x == 0) *n_eff_out = n_eff;
}

// boxes already in descending score order: identity order, just repack to 8 floats
__global__ void k_nms_identity(const float *__restrict__ boxes, int n_max, const int *__restrict__ n_dev, int pre_max,
                               int *__restrict__ order, float *__restrict__ sorted_boxes, int *__restrict__ n_eff_out) {
  const int fb = blockIdx.y;
  boxes += (size_t)fb * n_max * 7;
  order += (size_t)fb * n_max;
  sorted_boxes += (size_t)fb * n_max * 8;
  n_eff_out += fb;
  int n = n_dev ? min(n_dev[fb], n_max) : n_max;
  n = min(max(n, 0), pre_max);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    order[i] = i;
#pragma unroll
    for (int c = 0; c < 7; c++) sorted_boxes[i * 8 + c] = boxes[i * 7 + c];
    sorted_boxes[i * 8 + 7] = 0.f;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *n_eff_out = n;
}

// ---- (2) suppression mask: word (row i, col block c) bit t = iou(i, 64c + t) > thresh, for columns > i ---------------------
// One wavefront per 64-bit mask word: lane t evaluates the pair (i, 64c + t) and the word is the wave ballot.  The n^2/2
// pair tests are spread over n * ceil(n/64) / 2 independent wavefronts instead of a serial 64-iteration loop per lane.
template <bool NORMAL>
__global__ __launch_bounds__(64) void k_nms_mask(const float *__restrict__ sorted_boxes, const int *__restrict__ n_eff_p,
                                                 int col_blocks, float thresh, u64 *__restrict__ mask, int n_max) {
  const int fb = blockIdx.z;
  sorted_boxes += (size_t)fb * n_max * 8;
  mask += (size_t)fb * n_max * col_blocks;
  const int n = n_eff_p[fb];
  const int row = blockIdx.y, col_blk = blockIdx.x;
  if (row >= n || col_blk * 64 >= n || col_blk < (row >> 6)) return;
  const int lane = threadIdx.x;
  const int col = col_blk * 64 + lane;
  bool hit = false;
  if (col < n && col > row) {
    Box a = load_box(sorted_boxes + (size_t)row * 8);
    const float4 *src = reinterpret_cast<const float4 *>(sorted_boxes + (size_t)col * 8);
    float4 lo = src[0], hi = src[1];
    Box b;
    b.x = lo.x; b.y = lo.y; b.dx = lo.w; b.dy = hi.x; b.ang = hi.z;
    if (NORMAL) hit = iou_normal_of(a, b) > thresh;
    else if (!far_apart(a, b)) hit = iou_of(a, b) > thresh;
  }
  u64 word = __ballot(hit);
  if (lane == 0) mask[(size_t)row * col_blocks + col_blk] = word;
}

// ---- (3) greedy sweep, one wavefront, n <= 4096 ---------------------------------------------------------------------------
__device__ __forceinline__ u64 readlane64(u64 v, int lane) {
  unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)(v & 0xffffffffULL), lane);
  unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane);
  return ((u64)hi << 32) | lo;
}

__global__ __launch_bounds__(64) void k_nms_greedy(const u64 *__restrict__ mask, const int *__restrict__ n_eff_p, int col_blocks,
                                                   const int *__restrict__ order, int post_max, int *__restrict__ keep,
                                                   int *__restrict__ keep_count, int n_max) {
  const int fb = blockIdx.x;
  mask += (size_t)fb * n_max * col_blocks;
  order += (size_t)fb * n_max;
  keep += (size_t)fb * post_max;
  keep_count += fb;
  const int n = n_eff_p[fb];
  const int lane = threadIdx.x;
  const int nblk = (n + 63) >> 6;
  u64 remv = 0;          // lane j: removed bits of column block j
  int kept = 0;
  for (int blk = 0; blk < nblk; blk++) {
    const int row = blk * 64 + lane;
    const bool row_ok = row < n;
    u64 diag = row_ok ? mask[(size_t)row * col_blocks + blk] : 0ULL;
    u64 R = readlane64(remv, blk);
    const int rows_here = min(64, n - blk * 64);
    u64 kept_bits = 0;
    for (int i = 0; i < rows_here; i++) {
      if (!((R >> i) & 1ULL)) {
        kept_bits |= 1ULL << i;
        R |= readlane64(diag, i);
      }
    }
    // emit kept rows of this block in order
    if ((kept_bits >> lane) & 1ULL) {
      int pos = kept + __popcll(kept_bits & ((1ULL << lane) - 1ULL));
      if (pos < post_max) keep[pos] = order[row];
    }
    kept += __popcll(kept_bits);
    if (kept >= post_max) break;             // later boxes cannot enter keep[:post_max]
    // fold the kept rows' words into the later column blocks: lane j owns word j
    if (lane > blk && lane < nblk) {
      u64 acc = 0;
      u64 kb = kept_bits;
      while (kb) {
        int i = __ffsll((long long)kb) - 1;
        kb &= kb - 1;
        acc |= mask[(size_t)(blk * 64 + i) * col_blocks + lane];
      }
      remv |= acc;
    }
  }
  if (lane == 0) *keep_count = min(kept, post_max);
}

__global__ void k_pairwise(const float *__restrict__ a, int na, const float *__restrict__ b, int nb, int mode,
                           float *__restrict__ out) {
  int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (i >= na || j >= nb) return;
  Box A = load_box(a + (size_t)i * 7), B = load_box(b + (size_t)j * 7);
  out[(size_t)i * nb + j] = mode == 0 ? overlap_area(A, B) : iou_of(A, B);
}

struct NmsLayout {
  size_t order, sorted_boxes, mask, n_eff, total;
};
inline NmsLayout nms_layout(int n_max, int batch) {
  NmsLayout L;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = pcp_align_up(off + bytes, 256); return o; };
  int cb = (n_max + 63) / 64;
  L.order = take((size_t)batch * n_max * 4);
  L.sorted_boxes = take((size_t)batch * n_max * 8 * 4);
  L.mask = take((size_t)batch * n_max * cb * 8);
  L.n_eff = take((size_t)batch * 4 + 64);
  L.total = off;
  return L;
}

}  // namespace

extern "C" size_t pcp_nms_workspace_bytes(int32_t n_max, int32_t batch) {
  if (n_max <= 0 || batch <= 0) return 256;
  return nms_layout(n_max, batch).total;
}

namespace {
int nms_run(bool normal, const float *boxes, const float *scores, int32_t batch, int32_t n_max, const int32_t *n_dev,
            float thresh, int32_t pre_max, int32_t post_max, void *workspace, size_t workspace_bytes,
            int32_t *keep, int32_t *keep_count, void *stream_) {
  if (!keep || !keep_count || n_max < 0 || batch <= 0 || post_max <= 0 || pre_max <= 0) return PCP_ERR_ARG;
  hipStream_t st = (hipStream_t)stream_;
  if (n_max == 0) return pcp_zero_async(keep_count, 4 * (size_t)batch, st);
  if (!boxes || !workspace) return PCP_ERR_ARG;
  if (n_max > SORT_CAP || batch > 65535) return PCP_ERR_UNSUPPORTED;
  NmsLayout L = nms_layout(n_max, batch);
  if (workspace_bytes < L.total) return PCP_ERR_WORKSPACE;
  char *ws = (char *)workspace;
  int *order = (int *)(ws + L.order);
  float *sorted_boxes = (float *)(ws + L.sorted_boxes);
  u64 *mask = (u64 *)(ws + L.mask);
  int *n_eff = (int *)(ws + L.n_eff);
  const int cb = (n_max + 63) / 64;
  if (scores) {
    hipLaunchKernelGGL(k_nms_sort, dim3(batch), dim3(SORT_THREADS), 0, st, boxes, scores, n_max, n_dev, pre_max, order,
                       sorted_boxes, n_eff);
  } else {  // scores == NULL: every frame's boxes are already in descending score order
    hipLaunchKernelGGL(k_nms_identity, dim3((n_max + 255) / 256, batch), dim3(256), 0, st, boxes, n_max, n_dev, pre_max, order,
                       sorted_boxes, n_eff);
  }
  PCP_CHECK_LAUNCH();
  if (normal) hipLaunchKernelGGL(k_nms_mask<true>, dim3(cb, n_max, batch), dim3(64), 0, st, sorted_boxes, n_eff, cb, thresh, mask, n_max);
  else hipLaunchKernelGGL(k_nms_mask<false>, dim3(cb, n_max, batch), dim3(64), 0, st, sorted_boxes, n_eff, cb, thresh, mask, n_max);
  PCP_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_nms_greedy, dim3(batch), dim3(64), 0, st, mask, n_eff, cb, order, post_max, keep, keep_count, n_max);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}
}  // namespace

extern "C" int pcp_nms_rotated(const float *boxes, const float *scores, int32_t batch, int32_t n_max, const int32_t *n_dev,
                               float thresh, int32_t pre_max, int32_t post_max, void *workspace, size_t workspace_bytes,
                               int32_t *keep, int32_t *keep_count, void *stream_) {
  return nms_run(false, boxes, scores, batch, n_max, n_dev, thresh, pre_max, post_max, workspace, workspace_bytes, keep, keep_count, stream_);
}

extern "C" int pcp_nms_normal(const float *boxes, const float *scores, int32_t batch, int32_t n_max, const int32_t *n_dev,
                              float thresh, int32_t pre_max, int32_t post_max, void *workspace, size_t workspace_bytes,
                              int32_t *keep, int32_t *keep_count, void *stream_) {
  return nms_run(true, boxes, scores, batch, n_max, n_dev, thresh, pre_max, post_max, workspace, workspace_bytes, keep, keep_count, stream_);
}

extern "C" int pcp_boxes_bev_pairwise(const float *a, int32_t na, const float *b, int32_t nb, int32_t mode, float *out,
                                      void *stream_) {
  if (na < 0 || nb < 0 || (mode != 0 && mode != 1)) return PCP_ERR_ARG;
  if (na == 0 || nb == 0) return PCP_OK;
  if (!a || !b || !out) return PCP_ERR_ARG;
  hipLaunchKernelGGL(k_pairwise, dim3((nb + 63) / 64, na), dim3(64), 0, (hipStream_t)stream_, a, na, b, nb, mode, out);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}
