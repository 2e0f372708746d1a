// the producer and consumer ends of the lately-fusion exchange, on the device.
//
//  producer (remote agent, after its forward pass)
//    pcp_hunter_foreground_rows  pcdet/models/bev_layers/hunter_jr.py:377-397: rows whose background probability is < 0.3 are
//                                sent as [point features (without the frame index), sigmoid(cls logits)(3), flow(3)] -- a boolean-mask
//                                copy + cat in the reference; here an order-preserving stream compaction (block counts, one scan
//                                block, scatter).
//  consumer (ego, before its VFE)
//    pcp_points_in_boxes         pcdet/ops/roiaware_pool3d/src/roiaware_pool3d_kernel.cu:23-36,313-336 (points_in_boxes_gpu: index of
//                                the FIRST box containing the point, -1 if none; |z - cz| <= dz/2, |local x|, |local y| < d/2 + 1e-5)
//    pcp_modar_ingest            pcdet/datasets/v2x_sim/v2x_sim_dataset_ego.py:196-232: shift each MoDAR box by twice the mean flow of
//                                the foreground points inside it, map centre + heading to the ego frame (float64 pose, like
//                                apply_se3_, nuscenes_temporal_utils.py:66-70) and emit the 13-column point rows the ego model reads
//                                [x,y,z, 0, 0, dx,dy,dz, heading, score, label, max_sweep_idx, -1].
// All three are tiny (<= 83 boxes, a few thousand foreground points per agent): latency-bound, one launch each.
#include "pcp_common.h"

#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ int pt_in_box(const float *pt, const float *b) {
  const float x = pt[0], y = pt[1], z = pt[2];
  const float cx = b[0], cy = b[1], cz = b[2], dx = b[3], dy = b[4], dz = b[5], rz = b[6];
  if (fabsf(z - cz) > dz / 2.0f) return 0;
  const float cosa = cosf(-rz), sina = sinf(-rz);
  const float sx = x - cx, sy = y - cy;
  const float lx = sx * cosa + sy * (-sina);
  const float ly = sx * sina + sy * cosa;
  return (fabsf(lx) < dx / 2.0f + 1e-5f) && (fabsf(ly) < dy / 2.0f + 1e-5f);
}

__global__ void k_points_in_boxes(int batch, int nb, int np, const float *__restrict__ boxes, int box_stride, const float *__restrict__ pts,
                                  int pt_stride, int *__restrict__ out) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np) return;
  const float *p = pts + ((long long)b * np + i) * pt_stride;
  const float *bx = boxes + (long long)b * nb * box_stride;
  int idx = -1;
  for (int k = 0; k < nb; ++k)
    if (pt_in_box(p, bx + (long long)k * box_stride)) { idx = k; break; }
  out[(long long)b * np + i] = idx;
}

// ---- foreground rows: order-preserving compaction ---------------------------------------------------------------------------
constexpr int FG_THREADS = 256;

__device__ __forceinline__ int fg_flag(const float *head, int ld_head, long long i, float thresh) {
  const float pbg = 1.f / (1.f + expf(-head[i * ld_head]));
  return pbg < thresh ? 1 : 0;
}

__global__ __launch_bounds__(FG_THREADS) void k_fg_count(const float *__restrict__ head, int ld_head, long long n, float thresh,
                                                        int *__restrict__ block_counts) {
  __shared__ int wsum[FG_THREADS / 64];
  const long long i = (long long)blockIdx.x * FG_THREADS + threadIdx.x;
  const int f = i < n ? fg_flag(head, ld_head, i, thresh) : 0;
  const unsigned long long bal = __ballot(f);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(bal);
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(1024) void k_fg_scan(int *block_counts, int nblocks, int *total) {
  // exclusive scan in place by one workgroup (nblocks <= ~100k): chunked Hillis-Steele over 1024 lanes
  __shared__ int sh[1024];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nblocks; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < nblocks ? block_counts[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const int t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nblocks) block_counts[i] = carry + sh[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += sh[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(FG_THREADS) void k_fg_write(const float *__restrict__ points, int stride, const float *__restrict__ head,
                                                        int ld_head, long long n, float thresh, const int *__restrict__ block_offsets,
                                                        float *__restrict__ rows, int row_cols, int *__restrict__ row_batch) {
  __shared__ int wsum[FG_THREADS / 64];
  const long long i = (long long)blockIdx.x * FG_THREADS + threadIdx.x;
  const int f = i < n ? fg_flag(head, ld_head, i, thresh) : 0;
  const unsigned long long bal = __ballot(f);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) wsum[wv] = __popcll(bal);
  __syncthreads();
  if (!f) return;
  int off = block_offsets[blockIdx.x] + __popcll(bal & ((1ull << lane) - 1ull));
  for (int w = 0; w < wv; ++w) off += wsum[w];
  const float *src = points + i * stride;
  float *dst = rows + (long long)off * row_cols;
  const int nf = stride - 1;
  for (int k = 0; k < nf; ++k) dst[k] = src[1 + k];
  const float *h = head + i * ld_head;
#pragma unroll
  for (int k = 0; k < 3; ++k) dst[nf + k] = 1.f / (1.f + expf(-h[k]));
#pragma unroll
  for (int k = 0; k < 3; ++k) dst[nf + 3 + k] = h[3 + k];
  row_batch[off] = (int)src[0];
}

// ---- MoDAR ingestion: one workgroup per box ---------------------------------------------------------------------------------
struct Pose { double m[12]; };

__global__ __launch_bounds__(256) void k_modar_ingest(const float *__restrict__ modar, int n, const float *__restrict__ fg, int m, int fg_cols,
                                                     Pose T, float max_sweep_idx, float *__restrict__ rows) {
  __shared__ float red[4][4];
  const int k = blockIdx.x;
  const float *bx = modar + (long long)k * 9;
  float sx = 0.f, sy = 0.f, sz = 0.f, cnt = 0.f;
  for (int i = threadIdx.x; i < m; i += blockDim.x) {
    const float *p = fg + (long long)i * fg_cols;
    if (!pt_in_box(p, bx)) continue;
    bool earlier = false;                                    // points_in_boxes_gpu keeps the FIRST box that contains the point
    for (int j = 0; j < k && !earlier; ++j) earlier = pt_in_box(p, modar + (long long)j * 9);
    if (earlier) continue;
    sx += p[fg_cols - 3]; sy += p[fg_cols - 2]; sz += p[fg_cols - 1];
    cnt += 1.f;
  }
  float v[4] = {sx, sy, sz, cnt};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    for (int o = 32; o > 0; o >>= 1) v[q] += __shfl_down(v[q], o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][q] = v[q];
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  float tot[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) tot[q] = (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]);
  float c[3] = {bx[0], bx[1], bx[2]};
  if (tot[3] > 0.f) {
#pragma unroll
    for (int a = 0; a < 3; ++a) c[a] = c[a] + (tot[a] / tot[3]) * 2.f;          // scatter(mean) * 2, added in float32 (:216-218)
  }
  float *o = rows + (long long)k * 13;
#pragma unroll
  for (int a = 0; a < 3; ++a)
    o[a] = (float)((double)c[0] * T.m[4 * a] + (double)c[1] * T.m[4 * a + 1] + (double)c[2] * T.m[4 * a + 2] + T.m[4 * a + 3]);
  o[3] = 0.f;
  o[4] = 0.f;
  o[5] = bx[3]; o[6] = bx[4]; o[7] = bx[5];
  const float yaw = (float)((double)bx[6] + atan2(T.m[4], T.m[0]));
  o[8] = atan2f(sinf(yaw), cosf(yaw));
  o[9] = bx[7];
  o[10] = bx[8];
  o[11] = max_sweep_idx;
  o[12] = -1.f;
}

// ---- batched, device-driven ingestion (config 3 on one GPU: all remote agents of all frames in three launches, no host sync) ----------
//      group g = one (frame, remote agent) pair; detections as pcp_gather_detections leaves them (padded to det_max per group).

__global__ void k_fg_box_index(const float *__restrict__ det_boxes, const int *__restrict__ det_count, int det_max, const float *__restrict__ fg,
                               int fg_cols, const int *__restrict__ fg_group, const int *__restrict__ fg_count, int groups,
                               int *__restrict__ box_idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= *fg_count) return;
  const int g = fg_group[i];
  int idx = -1;
  if (g >= 0 && g < groups) {
    const float *bx = det_boxes + (long long)g * det_max * 7;
    const int nb = min(det_count[g], det_max);
    const float *p = fg + (long long)i * fg_cols;
    for (int k = 0; k < nb; ++k)
      if (pt_in_box(p, bx + k * 7)) { idx = k; break; }       // points_in_boxes_gpu: the FIRST box that contains the point
  }
  box_idx[i] = idx;
}

// start[g] = first foreground row of group g (rows are in the original point order, i.e. ascending group), start[groups] = count
__global__ void k_fg_group_starts(const int *__restrict__ fg_group, const int *__restrict__ fg_count, int groups, int *__restrict__ start) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g > groups) return;
  const int n = *fg_count;
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (fg_group[mid] < g) lo = mid + 1; else hi = mid;
  }
  start[g] = lo;
}

struct PoseTableD {
  const double *poses;          // device: [groups][12]
  const float *max_sweep;       // device: [groups]
  const int *frame;             // device: [groups] frame index written to column 0 of the emitted rows
};

__global__ __launch_bounds__(256) void k_modar_ingest_batched(const float *__restrict__ det_boxes, const float *__restrict__ det_scores,
                                                             const long long *__restrict__ det_labels, const int *__restrict__ det_count,
                                                             int det_max, const float *__restrict__ fg, int fg_cols,
                                                             const int *__restrict__ box_idx, const int *__restrict__ start, PoseTableD pt,
                                                             float *__restrict__ rows) {
  __shared__ float red[4][4];
  const int k = blockIdx.x, g = blockIdx.y;
  float *o = rows + ((long long)g * det_max + k) * 14;
  if (k >= det_count[g]) {                                   // padding slot: a row the pillariser drops (frame index -1)
    if (threadIdx.x < 14) o[threadIdx.x] = threadIdx.x == 0 ? -1.f : 0.f;
    return;
  }
  const float *bx = det_boxes + ((long long)g * det_max + k) * 7;
  float sx = 0.f, sy = 0.f, sz = 0.f, cnt = 0.f;
  for (int i = start[g] + threadIdx.x; i < start[g + 1]; i += blockDim.x) {
    if (box_idx[i] != k) continue;
    const float *p = fg + (long long)i * fg_cols;
    sx += p[fg_cols - 3]; sy += p[fg_cols - 2]; sz += p[fg_cols - 1];
    cnt += 1.f;
  }
  float v[4] = {sx, sy, sz, cnt};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    for (int of = 32; of > 0; of >>= 1) v[q] += __shfl_down(v[q], of);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][q] = v[q];
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  float tot[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) tot[q] = (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]);
  float c[3] = {bx[0], bx[1], bx[2]};
  if (tot[3] > 0.f) {
#pragma unroll
    for (int a = 0; a < 3; ++a) c[a] = c[a] + (tot[a] / tot[3]) * 2.f;
  }
  const double *T = pt.poses + (long long)g * 12;
  o[0] = (float)pt.frame[g];
#pragma unroll
  for (int a = 0; a < 3; ++a)
    o[1 + a] = (float)((double)c[0] * T[4 * a] + (double)c[1] * T[4 * a + 1] + (double)c[2] * T[4 * a + 2] + T[4 * a + 3]);
  o[4] = 0.f;
  o[5] = 0.f;
  o[6] = bx[3]; o[7] = bx[4]; o[8] = bx[5];
  const float yaw = (float)((double)bx[6] + atan2(T[4], T[0]));
  o[9] = atan2f(sinf(yaw), cosf(yaw));
  o[10] = det_scores[(long long)g * det_max + k];
  o[11] = (float)det_labels[(long long)g * det_max + k];
  o[12] = pt.max_sweep[g];
  o[13] = -1.f;
}

}  // namespace

extern "C" {

size_t pcp_modar_ingest_batched_workspace_bytes(int32_t groups, int64_t max_foreground) {
  return pcp_align_up((size_t)(max_foreground > 0 ? max_foreground : 1) * 4, 256) + pcp_align_up((size_t)(groups + 2) * 4, 256);
}

int pcp_modar_ingest_batched(const float *det_boxes, const float *det_scores, const int64_t *det_labels, const int32_t *det_count,
                             int32_t groups, int32_t det_max, const float *foreground, int32_t foreground_cols,
                             const int32_t *foreground_group, const int32_t *foreground_count, int64_t max_foreground,
                             const double *poses, const float *max_sweep_idx, const int32_t *frame_of_group, void *workspace,
                             size_t workspace_bytes, float *rows, void *stream) {
  if (groups <= 0 || det_max <= 0) return PCP_OK;
  if (!det_boxes || !det_scores || !det_labels || !det_count || !poses || !max_sweep_idx || !frame_of_group || !workspace || !rows ||
      !foreground_count || max_foreground < 0)
    return PCP_ERR_ARG;
  if (max_foreground > 0 && (!foreground || !foreground_group || foreground_cols < 6)) return PCP_ERR_ARG;
  if (workspace_bytes < pcp_modar_ingest_batched_workspace_bytes(groups, max_foreground)) return PCP_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  int *box_idx = (int *)workspace;
  int *start = (int *)((char *)workspace + pcp_align_up((size_t)(max_foreground > 0 ? max_foreground : 1) * 4, 256));
  if (max_foreground > 0) {
    hipLaunchKernelGGL(k_fg_box_index, dim3((unsigned)((max_foreground + 255) / 256)), dim3(256), 0, s, det_boxes, det_count, det_max, foreground,
                       foreground_cols, foreground_group, foreground_count, groups, box_idx);
    PCP_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(k_fg_group_starts, dim3((groups + 1 + 63) / 64), dim3(64), 0, s, foreground_group, foreground_count, groups, start);
  PCP_CHECK_LAUNCH();
  PoseTableD pt;
  pt.poses = poses; pt.max_sweep = max_sweep_idx; pt.frame = frame_of_group;
  hipLaunchKernelGGL(k_modar_ingest_batched, dim3(det_max, groups), dim3(256), 0, s, det_boxes, det_scores, (const long long *)det_labels,
                     det_count, det_max, foreground, foreground_cols, box_idx, start, pt, rows);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

int pcp_points_in_boxes(const float *boxes, int32_t batch, int32_t n_boxes, int32_t box_stride, const float *points, int32_t n_points,
                        int32_t point_stride, int32_t *box_idx, void *stream) {
  if (!boxes || !points || !box_idx || batch <= 0 || n_boxes < 0 || n_points < 0 || box_stride < 7 || point_stride < 3) return PCP_ERR_ARG;
  if (n_points == 0) return PCP_OK;
  hipLaunchKernelGGL(k_points_in_boxes, dim3((n_points + 255) / 256, batch), dim3(256), 0, (hipStream_t)stream, batch, n_boxes, n_points, boxes,
                     box_stride, points, point_stride, box_idx);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

size_t pcp_hunter_foreground_workspace_bytes(int64_t n) { return (size_t)((n + FG_THREADS - 1) / FG_THREADS + 2) * sizeof(int32_t); }

int pcp_hunter_foreground_rows(const float *points, int64_t n, int32_t row_stride, const float *head, int32_t ld_head, float thresh_bg,
                               void *workspace, size_t workspace_bytes, float *rows, int32_t *row_batch, int32_t *count, void *stream) {
  if (!points || !head || !workspace || !rows || !row_batch || !count || n < 0 || row_stride < 4 || ld_head < 6) return PCP_ERR_ARG;
  if (workspace_bytes < pcp_hunter_foreground_workspace_bytes(n)) return PCP_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) return pcp_zero_async(count, sizeof(int32_t), s);
  const int nblocks = (int)((n + FG_THREADS - 1) / FG_THREADS);
  int *bc = (int *)workspace;
  hipLaunchKernelGGL(k_fg_count, dim3(nblocks), dim3(FG_THREADS), 0, s, head, ld_head, (long long)n, thresh_bg, bc);
  hipLaunchKernelGGL(k_fg_scan, dim3(1), dim3(1024), 0, s, bc, nblocks, count);
  hipLaunchKernelGGL(k_fg_write, dim3(nblocks), dim3(FG_THREADS), 0, s, points, row_stride, head, ld_head, (long long)n, thresh_bg, bc, rows,
                     row_stride - 1 + 6, row_batch);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

int pcp_modar_ingest(const float *modar, int32_t n_modar, const float *foreground, int32_t n_foreground, int32_t foreground_cols,
                     const double *target_se3_lidar_host, float max_sweep_idx, float *rows, void *stream) {
  if (n_modar == 0) return PCP_OK;
  if (!modar || !target_se3_lidar_host || !rows || n_modar < 0 || n_foreground < 0) return PCP_ERR_ARG;
  if (n_foreground > 0 && (!foreground || foreground_cols < 6)) return PCP_ERR_ARG;
  Pose T;
  for (int i = 0; i < 12; ++i) T.m[i] = target_se3_lidar_host[i];
  hipLaunchKernelGGL(k_modar_ingest, dim3(n_modar), dim3(256), 0, (hipStream_t)stream, modar, n_modar, foreground, n_foreground,
                     foreground_cols, T, max_sweep_idx, rows);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

}  // extern "C"
