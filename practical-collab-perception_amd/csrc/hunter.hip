// a14 -- HunterJr point <-> BEV operators (configs 1 and 2 only).
//
// Replaces pcdet/models/bev_layers/hunter_toolbox.py:8-39,94-127 (bilinear sampling of the BEV map at every point:
// four fancy-index gathers, four weight vectors and three transposes per frame) and :65-91 (bev_scatter: float-coordinate
// mask, truncation, torch.unique + torch_scatter.scatter_mean, dense re-layout).
//
//  * pcp_bev_sample_bilinear: one thread per (point, 4 channels); the four neighbour rows are contiguous C-float rows of
//    the NHWC map (L2-resident: 25 MB at 384x128x128), products and sums in the reference's order.
//  * pcp_bev_scatter_mean: counting sort of the kept points by BEV cell (dense B*H*W table, scan, bucket fill), then one
//    workgroup per cell sums its rows IN POINT-INDEX ORDER (the bucket is sorted in LDS first), so the fp32 result is
//    independent of atomic arrival order and equals a sequential index_add_; empty cells are written as zeros by the
//    same kernel (no separate clear of the 25 MB map).
#include "pcp_common.h"

#pragma clang fp contract(off)

namespace {

__global__ void k_bilinear(const float *__restrict__ bev, int batch, int h, int w, int c4, int ld_bev,
                           const float *__restrict__ points, long long n, int stride, float min_x, float min_y, float pix_x,
                           float pix_y, const unsigned char *__restrict__ row_mask, float *__restrict__ out, int ld_out) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * c4) return;
  int q = (int)(t % c4);
  long long i = t / c4;
  if (row_mask && !row_mask[i]) return;
  const float *row = points + i * stride;
  int b = (int)row[0];
  if (b < 0 || b >= batch) return;
  float x = __fdiv_rn(row[1] - min_x, pix_x), y = __fdiv_rn(row[2] - min_y, pix_y);
  // floor / +1 / clamp on the integer side, weights from the CLAMPED corners (hunter_toolbox.py:19-37)
  float fx0 = floorf(x), fy0 = floorf(y);
  // .long() of an out-of-range float is undefined in the reference; clamp in float first (same result in range)
  fx0 = fminf(fmaxf(fx0, -2.0f), (float)w + 1.0f);
  fy0 = fminf(fmaxf(fy0, -2.0f), (float)h + 1.0f);
  int x0 = (int)fx0, y0 = (int)fy0;
  int x1 = x0 + 1, y1 = y0 + 1;
  x0 = min(max(x0, 0), w - 1);
  x1 = min(max(x1, 0), w - 1);
  y0 = min(max(y0, 0), h - 1);
  y1 = min(max(y1, 0), h - 1);
  float wa = ((float)x1 - x) * ((float)y1 - y);
  float wb = ((float)x1 - x) * (y - (float)y0);
  float wc = (x - (float)x0) * ((float)y1 - y);
  float wd = (x - (float)x0) * (y - (float)y0);
  const float *img = bev + (long long)b * h * w * ld_bev + q * 4;
  float4 Ia = *reinterpret_cast<const float4 *>(img + ((long long)y0 * w + x0) * ld_bev);
  float4 Ib = *reinterpret_cast<const float4 *>(img + ((long long)y1 * w + x0) * ld_bev);
  float4 Ic = *reinterpret_cast<const float4 *>(img + ((long long)y0 * w + x1) * ld_bev);
  float4 Id = *reinterpret_cast<const float4 *>(img + ((long long)y1 * w + x1) * ld_bev);
  float4 r;
  r.x = Ia.x * wa + Ib.x * wb + Ic.x * wc + Id.x * wd;
  r.y = Ia.y * wa + Ib.y * wb + Ic.y * wc + Id.y * wd;
  r.z = Ia.z * wa + Ib.z * wb + Ic.z * wc + Id.z * wd;
  r.w = Ia.w * wa + Ib.w * wb + Ic.w * wc + Id.w * wd;
  *reinterpret_cast<float4 *>(out + i * ld_out + q * 4) = r;
}

struct ScLayout {
  size_t cell_count, cell_fill, cell_start, point_cell, bucket, big_list, total;
};
constexpr int SC_BIG = 256;                 // cells of more rows than this are listed by the scan and summed by k_sc_mean_big
inline ScLayout sc_layout(long long cells, long long n) {
  ScLayout L;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = pcp_align_up(off + bytes, 256); return o; };
  L.cell_count = take((size_t)cells * 4);
  L.cell_fill = take((size_t)cells * 4);
  L.cell_start = take((size_t)(cells + 1) * 4);
  L.point_cell = take((size_t)(n > 0 ? n : 1) * 4);
  L.bucket = take((size_t)(n > 0 ? n : 1) * 4);
  L.big_list = take((size_t)((n > 0 ? n : 1) / SC_BIG + 2) * 4);          // [0] = count, then the cells
  L.total = off;
  return L;
}

__global__ void k_sc_cells(const float *__restrict__ points, long long n, int stride, int batch, int h, int w, float min_x,
                           float min_y, float pix_x, float pix_y, int *__restrict__ cell_count, int *__restrict__ point_cell) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float *row = points + i * stride;
  int b = (int)row[0];
  float x = __fdiv_rn(row[1] - min_x, pix_x), y = __fdiv_rn(row[2] - min_y, pix_y);
  // strict float comparisons BEFORE truncation (hunter_toolbox.py:80-84)
  bool ok = (x > 0.0f) && (x < (float)w) && (y > 0.0f) && (y < (float)h) && b >= 0 && b < batch;
  int cell = -1;
  if (ok) {
    cell = b * h * w + (int)y * w + (int)x;
    atomicAdd(&cell_count[cell], 1);
  }
  point_cell[i] = cell;
}

// exclusive scan of the per-cell counts (B*H*W = 65 536 cells at 4 frames) in one workgroup: 64 cells per thread (16 independent
// 16-byte loads in flight), 65 536 per round -- one round, three barriers at 4 frames
__global__ __launch_bounds__(1024) void k_sc_scan(const int *__restrict__ cell_count, long long cells, int *__restrict__ cell_start,
                                                  int *__restrict__ big_list) {
  constexpr int IT = 64;
  __shared__ int wave_sum[16];
  __shared__ int carry_s;
  __shared__ int n_big;
  if (threadIdx.x == 0) {
    carry_s = 0;
    n_big = 0;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long long base = 0; base < cells; base += 1024 * IT) {
    const long long idx = base + (long long)threadIdx.x * IT;
    int v[IT];
    if (idx + IT <= cells) {
#pragma unroll
      for (int q = 0; q < IT / 4; q++) {
        const int4 a = *reinterpret_cast<const int4 *>(cell_count + idx + 4 * q);
        v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < IT; i++) v[i] = idx + i < cells ? cell_count[idx + i] : 0;
    }
    int local = 0;
#pragma unroll
    for (int i = 0; i < IT; i++) {
      local += v[i];
      if (v[i] > SC_BIG) big_list[1 + atomicAdd(&n_big, 1)] = (int)(idx + i);      // crowded cells: k_sc_mean_big (any order)
    }
    int incl = local;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
      int u = __shfl_up(incl, s, 64);
      if (lane >= s) incl += u;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    int wbase = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      int s = wave_sum[k];
      if (k < wave) wbase += s;
      tot += s;
    }
    const int carry = carry_s;
    int run = carry + wbase + incl - local;
    if (idx + IT <= cells) {
#pragma unroll
      for (int q = 0; q < IT / 4; q++) {
        int4 o;
        o.x = run; run += v[4 * q];
        o.y = run; run += v[4 * q + 1];
        o.z = run; run += v[4 * q + 2];
        o.w = run; run += v[4 * q + 3];
        *reinterpret_cast<int4 *>(cell_start + idx + 4 * q) = o;
      }
    } else {
#pragma unroll
      for (int i = 0; i < IT; i++) {
        if (idx + i < cells) cell_start[idx + i] = run;
        run += v[i];
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    cell_start[cells] = carry_s;
    big_list[0] = n_big;
  }
}

__global__ void k_sc_fill(const int *__restrict__ point_cell, long long n, const int *__restrict__ cell_start,
                          int *__restrict__ cell_fill, int *__restrict__ bucket) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int c = point_cell[i];
  if (c < 0) return;
  bucket[cell_start[c] + atomicAdd(&cell_fill[c], 1)] = (int)i;
}

constexpr int SC_THREADS = 128;
constexpr int SC_SORT_CAP = 1024;

__global__ __launch_bounds__(SC_THREADS) void k_sc_mean(const int *__restrict__ cell_start, const int *__restrict__ bucket,
                                                        const float *__restrict__ feat, int ld_feat, int c4,
                                                        float *__restrict__ out, int ld_out) {
  __shared__ int ids[SC_SORT_CAP];
  const long long cell = blockIdx.x;
  const int s0 = cell_start[cell], cnt = cell_start[cell + 1] - s0;
  float4 *orow = reinterpret_cast<float4 *>(out + cell * ld_out);
  if (cnt == 0) {
    for (int q = threadIdx.x; q < c4; q += SC_THREADS) orow[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  if (cnt > SC_BIG) return;                      // listed by the scan: k_sc_mean_big
  const bool sorted = cnt <= SC_SORT_CAP;
  if (sorted) {
    int cap = 2;
    while (cap < cnt) cap <<= 1;
    for (int i = threadIdx.x; i < cap; i += SC_THREADS) ids[i] = i < cnt ? bucket[s0 + i] : 0x7fffffff;
    __syncthreads();
    for (int k = 2; k <= cap; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int t = threadIdx.x; t < cap / 2; t += SC_THREADS) {
          int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
          int q = i | j;
          bool asc = (i & k) == 0;
          int a = ids[i], b = ids[q];
          if ((a > b) == asc) { ids[i] = b; ids[q] = a; }
        }
        __syncthreads();
      }
  }
  const float inv_unused = 0.f;
  (void)inv_unused;
  for (int q = threadIdx.x; q < c4; q += SC_THREADS) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < cnt; k++) {
      int pid = sorted ? ids[k] : bucket[s0 + k];
      float4 v = *reinterpret_cast<const float4 *>(feat + (long long)pid * ld_feat + q * 4);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    float fc = (float)cnt;
    acc.x /= fc; acc.y /= fc; acc.z /= fc; acc.w /= fc;
    orow[q] = acc;
  }
}

// Crowded cells (more than SC_BIG rows: the cells under the sensor of a LiDAR-like cloud hold thousands): a 512-thread workgroup per listed
// cell, 512 / c4 ROW SLOTS (slot s takes the rows s, s + slots, ... in ascending order, eight loads in flight), the slot sums added in slot
// order -- the one-workgroup-of-128 walk above kept 12 KB in flight per cell and took 0.8 ms on the 60 k-point ring cloud.  Up to
// SC_BIG_SORT rows the cell's rows are first sorted by point index (as above), so the result is a function of the input alone; it is
// grouped differently from the sequential index_add_ (last bits).
constexpr int SC_BIG_THREADS = 512;
constexpr int SC_BIG_SORT = 4096;

__global__ __launch_bounds__(SC_BIG_THREADS) void k_sc_mean_big(const int *__restrict__ cell_start, const int *__restrict__ bucket,
                                                               const int *__restrict__ big_list, const float *__restrict__ feat, int ld_feat,
                                                               int c4, float *__restrict__ out, int ld_out) {
  __shared__ int ids[SC_BIG_SORT];
  extern __shared__ float4 part[];                                  // [slots][c4]
  const int n_big = big_list[0];
  const int slots = c4 <= SC_BIG_THREADS ? SC_BIG_THREADS / c4 : 0;
  for (int li = blockIdx.x; li < n_big; li += gridDim.x) {
    const long long cell = big_list[1 + li];
    const int s0 = cell_start[cell], cnt = cell_start[cell + 1] - s0;
    const bool sorted = cnt <= SC_BIG_SORT;
    if (sorted) {
      int cap = 2;
      while (cap < cnt) cap <<= 1;
      for (int i = threadIdx.x; i < cap; i += SC_BIG_THREADS) ids[i] = i < cnt ? bucket[s0 + i] : 0x7fffffff;
      __syncthreads();
      for (int k = 2; k <= cap; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
          for (int t = threadIdx.x; t < cap / 2; t += SC_BIG_THREADS) {
            const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
            const int q = i | j;
            const bool asc = (i & k) == 0;
            const int a = ids[i], b = ids[q];
            if ((a > b) == asc) { ids[i] = b; ids[q] = a; }
          }
          __syncthreads();
        }
    }
    float4 *orow = reinterpret_cast<float4 *>(out + cell * ld_out);
    if (slots == 0) {                                               // wider than the workgroup: a thread per channel quad, round robin
      for (int q = threadIdx.x; q < c4; q += SC_BIG_THREADS) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < cnt; k++) {
          const int pid = sorted ? ids[k] : bucket[s0 + k];
          const float4 v = *reinterpret_cast<const float4 *>(feat + (long long)pid * ld_feat + q * 4);
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        const float fc = (float)cnt;
        orow[q] = make_float4(acc.x / fc, acc.y / fc, acc.z / fc, acc.w / fc);
      }
      __syncthreads();
      continue;
    }
    const int q = threadIdx.x % c4, slot = threadIdx.x / c4;
    if (slot < slots) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      int k = slot;
      for (; k + 7 * slots < cnt; k += 8 * slots) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int pid = sorted ? ids[k + u * slots] : bucket[s0 + k + u * slots];
          v[u] = *reinterpret_cast<const float4 *>(feat + (long long)pid * ld_feat + q * 4);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
      }
      for (; k < cnt; k += slots) {
        const int pid = sorted ? ids[k] : bucket[s0 + k];
        const float4 v = *reinterpret_cast<const float4 *>(feat + (long long)pid * ld_feat + q * 4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      part[slot * c4 + q] = acc;
    }
    __syncthreads();
    if (slot == 0) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int sl = 0; sl < slots; sl++) {
        const float4 v = part[sl * c4 + q];
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
      }
      const float fc = (float)cnt;
      orow[q] = make_float4(t.x / fc, t.y / fc, t.z / fc, t.w / fc);
    }
    __syncthreads();
  }
}

__global__ void k_apply_flow(float *__restrict__ points, long long n, int stride, const float *__restrict__ head, int ld_head,
                             float thresh, unsigned char *__restrict__ row_mask) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float *hrow = head + i * ld_head;
  float p0 = 1.0f / (1.0f + expf(-hrow[0])), p1 = 1.0f / (1.0f + expf(-hrow[1])), p2 = 1.0f / (1.0f + expf(-hrow[2]));
  // torch.max over dim=1 returns the FIRST maximal index: class 2 wins only when strictly larger than both
  bool dyn = (p2 > p0) && (p2 > p1) && (p2 > thresh);
  if (dyn) {
    float *row = points + i * stride;
    row[1] = row[1] + hrow[3];
    row[2] = row[2] + hrow[4];
    row[3] = row[3] + hrow[5];
  }
  row_mask[i] = dyn ? 1 : 0;
}

// bev_scatter backward: every kept point receives d map[cell] / count[cell]; rows of `dyn` points go to dfeat_dyn (overwritten), the
// others are ADDED to dfeat_acc (they carry the first sampling's features)
__global__ void k_sc_mean_backward(const int *__restrict__ cell_count, const int *__restrict__ point_cell, long long n, const float *__restrict__ dmap,
                                   int ld_dmap, int c, const unsigned char *__restrict__ dyn, float *__restrict__ dfeat_acc, int ld_acc,
                                   float *__restrict__ dfeat_dyn, int ld_dyn) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * c) return;
  const long long i = t / c;
  const int ch = (int)(t % c);
  const int cell = point_cell[i];
  const float v = cell >= 0 ? dmap[(long long)cell * ld_dmap + ch] / (float)cell_count[cell] : 0.f;
  if (dyn[i]) dfeat_dyn[i * ld_dyn + ch] = v; else dfeat_acc[i * ld_acc + ch] += v;
}

constexpr int MAX_POSE_BATCH = 16;
struct PoseTable {
  float m[MAX_POSE_BATCH][12];
  unsigned char present[MAX_POSE_BATCH];
};

__global__ void k_select_transform(const float *__restrict__ points, long long n, int stride, int agent_col, float agent,
                                   int batch, PoseTable pt, float *__restrict__ out, float batch_offset) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float *row = points + i * stride;
  float *o = out + i * stride;
  int b = (int)row[0];
  bool mine = (row[agent_col] == agent) && b >= 0 && b < batch && pt.present[b];
  for (int c = 0; c < stride; c++) o[c] = row[c];
  if (!mine) {
    o[0] = -1.0f;
    return;
  }
  const float *T = pt.m[b];
  float x = row[1], y = row[2], z = row[3];
  o[0] = row[0] + batch_offset;
  // bev_maker.py:179 `p @ R^T + t` on the reference's CPU path (torch -> BLAS sgemm, K = 3) is, bit for bit, the FMA chain
  // fma(z, r2, fma(y, r1, x * r0)) followed by a separately rounded + t (pinned on 60 k-row clouds: tests/golden/g2_disco_full.npz
  // holds the SHA-256 of the transformed rows).  Spelled with explicit roundings so no contraction setting can reorder it.
  o[1] = __fadd_rn(__fmaf_rn(z, T[2], __fmaf_rn(y, T[1], __fmul_rn(x, T[0]))), T[3]);
  o[2] = __fadd_rn(__fmaf_rn(z, T[6], __fmaf_rn(y, T[5], __fmul_rn(x, T[4]))), T[7]);
  o[3] = __fadd_rn(__fmaf_rn(z, T[10], __fmaf_rn(y, T[9], __fmul_rn(x, T[8]))), T[11]);
}


// ---- a13: which agent ids occur in a column (bev_maker.py:156 torch.unique(points[:, -1])): 64-bit presence mask of the ids
//      0..63 (values truncated like .long()); out[1] counts rows outside that range (the caller raises: there is no fallback) ----------------
__global__ __launch_bounds__(256) void k_column_id_mask(const float *__restrict__ points, long long n, int stride, int col,
                                                        unsigned long long *__restrict__ out) {
  __shared__ unsigned long long s_mask, s_bad;
  if (threadIdx.x == 0) { s_mask = 0ULL; s_bad = 0ULL; }
  __syncthreads();
  unsigned long long m = 0ULL, bad = 0ULL;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float v = points[i * stride + col];
    const int iv = (int)v;                                  // truncation toward zero = the reference's points[:, -1].long()
    if (v > -1.f && v < 64.f) m |= 1ULL << iv; else bad++;
  }
  if (m) atomicOr(&s_mask, m);
  if (bad) atomicAdd(&s_bad, bad);
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_mask) atomicOr(&out[0], s_mask);
    if (s_bad) atomicAdd(&out[1], s_bad);
  }
}

}  // namespace

extern "C" int pcp_hunter_apply_flow(float *points, int64_t n, int32_t row_stride, const float *head, int32_t ld_head,
                                     float thresh, uint8_t *row_mask, void *stream_) {
  if (n < 0 || row_stride < 4 || ld_head < 6) return PCP_ERR_ARG;
  if (n == 0) return PCP_OK;
  if (!points || !head || !row_mask) return PCP_ERR_ARG;
  hipLaunchKernelGGL(k_apply_flow, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, points, (long long)n,
                     row_stride, head, ld_head, thresh, row_mask);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

extern "C" int pcp_select_transform_points(const float *points, int64_t n, int32_t row_stride, int32_t agent_col, float agent,
                                           int32_t batch, const float *poses_host, const uint8_t *present_host, float *out,
                                           int32_t out_batch_offset, void *stream_) {
  if (n < 0 || row_stride < 4 || agent_col < 0 || agent_col >= row_stride || batch <= 0 || batch > MAX_POSE_BATCH || out_batch_offset < 0)
    return PCP_ERR_ARG;
  if (n == 0) return PCP_OK;
  if (!points || !poses_host || !present_host || !out || points == out) return PCP_ERR_ARG;
  PoseTable pt;
  for (int b = 0; b < MAX_POSE_BATCH; b++) {
    pt.present[b] = b < batch ? present_host[b] : 0;
    for (int k = 0; k < 12; k++) pt.m[b][k] = b < batch ? poses_host[b * 12 + k] : 0.f;
  }
  hipLaunchKernelGGL(k_select_transform, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, points,
                     (long long)n, row_stride, agent_col, agent, batch, pt, out, (float)out_batch_offset);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

extern "C" int pcp_bev_sample_bilinear(const float *bev, int32_t batch, int32_t h, int32_t w, int32_t c, int32_t ld_bev,
                                       const float *points, int64_t n, int32_t row_stride, float min_x, float min_y, float pix_x,
                                       float pix_y, const uint8_t *row_mask, float *out, int32_t ld_out, void *stream_) {
  if (!bev || !out || n < 0 || batch <= 0 || h <= 0 || w <= 0 || c <= 0 || (c & 3) || (ld_bev & 3) || (ld_out & 3) ||
      row_stride < 3)
    return PCP_ERR_ARG;
  if (n == 0) return PCP_OK;
  if (!points) return PCP_ERR_ARG;
  long long total = (long long)n * (c / 4);
  hipLaunchKernelGGL(k_bilinear, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, bev, batch, h, w, c / 4,
                     ld_bev, points, (long long)n, row_stride, min_x, min_y, pix_x, pix_y, row_mask, out, ld_out);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

extern "C" size_t pcp_bev_scatter_mean_workspace_bytes(int32_t batch, int32_t h, int32_t w, int64_t n) {
  return sc_layout((long long)batch * h * w, n).total;
}

extern "C" int pcp_bev_scatter_mean(const float *points, int64_t n, int32_t row_stride, const float *feat, int32_t ld_feat,
                                    int32_t c, int32_t batch, int32_t h, int32_t w, float min_x, float min_y, float pix_x,
                                    float pix_y, void *workspace, size_t workspace_bytes, float *out, int32_t ld_out,
                                    void *stream_) {
  if (!workspace || !out || n < 0 || batch <= 0 || h <= 0 || w <= 0 || c <= 0 || (c & 3) || (ld_feat & 3) || (ld_out & 3) ||
      row_stride < 3)
    return PCP_ERR_ARG;
  if (n > 0 && (!points || !feat)) return PCP_ERR_ARG;
  const long long cells = (long long)batch * h * w;
  ScLayout L = sc_layout(cells, n);
  if (workspace_bytes < L.total) return PCP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream_;
  char *ws = (char *)workspace;
  int *cell_count = (int *)(ws + L.cell_count), *cell_fill = (int *)(ws + L.cell_fill);
  int *cell_start = (int *)(ws + L.cell_start), *point_cell = (int *)(ws + L.point_cell), *bucket = (int *)(ws + L.bucket);
  if (pcp_zero_async(cell_count, L.cell_start - L.cell_count, st) != PCP_OK) return PCP_ERR_LAUNCH;
  if (n > 0) {
    hipLaunchKernelGGL(k_sc_cells, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, points, (long long)n, row_stride, batch, h, w,
                       min_x, min_y, pix_x, pix_y, cell_count, point_cell);
    PCP_CHECK_LAUNCH();
  }
  int *big_list = (int *)(ws + L.big_list);
  hipLaunchKernelGGL(k_sc_scan, dim3(1), dim3(1024), 0, st, cell_count, cells, cell_start, big_list);
  PCP_CHECK_LAUNCH();
  if (n > 0) {
    hipLaunchKernelGGL(k_sc_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, point_cell, (long long)n, cell_start,
                       cell_fill, bucket);
    PCP_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(k_sc_mean, dim3((unsigned)cells), dim3(SC_THREADS), 0, st, cell_start, bucket, feat, ld_feat, c / 4, out,
                     ld_out);
  PCP_CHECK_LAUNCH();
  // the crowded cells the kernel above left out (none in most clouds: the workgroups read the list length and leave)
  const int c4 = c / 4;
  const size_t part_bytes = (size_t)(c4 <= SC_BIG_THREADS ? (SC_BIG_THREADS / c4) * c4 : 1) * sizeof(float4);
  hipLaunchKernelGGL(k_sc_mean_big, dim3(256), dim3(SC_BIG_THREADS), part_bytes, st, cell_start, bucket, (const int *)big_list, feat, ld_feat, c4,
                     out, ld_out);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}


// backward of pcp_bev_scatter_mean (training, include/pcp_hip_train.h): reads the cell counts and the per-point cell list the forward left
// in ITS workspace (same batch / h / w / n)
extern "C" int pcp_bev_scatter_mean_backward(const void *scatter_workspace, int32_t batch, int32_t h, int32_t w, int64_t n, const float *dmap,
                                             int32_t ld_dmap, int32_t c, const uint8_t *dyn_mask, float *dfeat_acc, int32_t ld_acc,
                                             float *dfeat_dyn, int32_t ld_dyn, void *stream_) {
  if (!scatter_workspace || !dmap || !dyn_mask || !dfeat_acc || !dfeat_dyn || n <= 0 || c <= 0 || batch <= 0 || h <= 0 || w <= 0) return PCP_ERR_ARG;
  ScLayout L = sc_layout((long long)batch * h * w, n);
  const char *ws = (const char *)scatter_workspace;
  const int *cell_count = (const int *)(ws + L.cell_count), *point_cell = (const int *)(ws + L.point_cell);
  hipLaunchKernelGGL(k_sc_mean_backward, dim3((unsigned)((n * c + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, cell_count, point_cell,
                     (long long)n, dmap, ld_dmap, c, dyn_mask, dfeat_acc, ld_acc, dfeat_dyn, ld_dyn);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

extern "C" int pcp_column_id_mask(const float *points, int64_t n, int32_t row_stride, int32_t col, uint64_t *out2, void *stream_) {
  if (!out2 || n < 0 || row_stride <= 0 || col < 0 || col >= row_stride) return PCP_ERR_ARG;
  hipStream_t st = (hipStream_t)stream_;
  int rc = pcp_zero_async(out2, 16, st);
  if (rc != PCP_OK || n == 0) return rc;
  if (!points) return PCP_ERR_ARG;
  long long blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(k_column_id_mask, dim3((unsigned)blocks), dim3(256), 0, st, points, (long long)n, row_stride, col,
                     reinterpret_cast<unsigned long long *>(out2));
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}
