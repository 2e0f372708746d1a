// a1 / a4 -- dynamic pillarisation without a sort.
//
// Reference behaviour reproduced bit for bit (pcdet/models/backbones_3d/vfe/dynamic_pillar_vfe.py:96-108, 137-147):
//   c = floor((xy - min_xy) / voxel_xy)  in IEEE fp32 (subtract, then a correctly rounded DIVIDE -- not a multiply by 5),
//   keep rows with 0 <= c < grid on x and y only, merged = b*nx*ny + cx*ny + cy, torch.unique(sorted) -> pillar rank.
// Because the merged ids live in a dense table of B*nx*ny cells, "sorted unique + inverse + counts" is an exclusive scan
// of the per-cell occupancy: rank(cell) = #occupied cells with a smaller id.  No radix sort, no host sync.
//
// HBM traffic per call: points read twice (n * row_stride * 4 B, second pass only column 0..2 -> same lines),
// 4 dense int32 tables of B*nx*ny (1 MiB each at 512x512), and O(n) int32 side arrays.
#include "pcp_common.h"

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 4;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;   // 1024 elements per block

typedef unsigned long long u64;

__device__ __forceinline__ int point_to_cell(const float *row, const pcp_grid_t g) {
  // column 0 = batch index, 1 = x, 2 = y
  float fb = row[0];
  float x = row[1], y = row[2];
  float cx = floorf(__fdiv_rn(__fsub_rn(x, g.min_x), g.voxel_x));
  float cy = floorf(__fdiv_rn(__fsub_rn(y, g.min_y), g.voxel_y));
  // float comparisons: NaN / inf / out-of-range all fail (torch's (coords >= 0) & (coords < grid) on the int32 cast
  // gives the same verdict for every finite in-int-range value)
  bool ok = (cx >= 0.0f) && (cx < (float)g.nx) && (cy >= 0.0f) && (cy < (float)g.ny) && (fb >= 0.0f) &&
            (fb < (float)g.batch_size);
  if (!ok) return -1;
  int b = (int)fb;
  return b * (g.nx * g.ny) + (int)cx * g.ny + (int)cy;
}

// exclusive block scan of one value per (thread, item); returns block total through *total (all threads)
template <typename T>
__device__ __forceinline__ void block_scan_excl(T (&v)[SCAN_ITEMS], T *lds_wave /*[SCAN_THREADS/64 + 1]*/, T *total) {
  T local = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) local += v[i];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T incl = local;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    T up = __shfl_up(incl, d, 64);
    if (lane >= d) incl += up;
  }
  if (lane == 63) lds_wave[wave] = incl;
  __syncthreads();
  T wave_base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < SCAN_THREADS / 64; w++) {
    T s = lds_wave[w];
    if (w < wave) wave_base += s;
    tot += s;
  }
  T run = wave_base + incl - local;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    T t = v[i];
    v[i] = run;
    run += t;
  }
  *total = tot;
  __syncthreads();
}

// pass 1 over points: cell id per row, per-cell histogram, valid rows per 1024-row tile
__global__ __launch_bounds__(SCAN_THREADS) void k_point_cells(const float *__restrict__ points, long long n, int stride,
                                                              pcp_grid_t g, int *__restrict__ cell_count,
                                                              int *__restrict__ point_cell, int *__restrict__ pt_block_sums) {
  __shared__ int wave_tot[SCAN_THREADS / 64];
  long long base = (long long)blockIdx.x * SCAN_TILE;
  int valid = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    long long r = base + i * SCAN_THREADS + threadIdx.x;     // coalesced over rows
    if (r < n) {
      int c = point_to_cell(points + r * stride, g);
      point_cell[r] = c;
      if (c >= 0) {
        atomicAdd(&cell_count[c], 1);
        valid++;
      }
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) valid += __shfl_xor(valid, d, 64);
  if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = valid;
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
    for (int w = 0; w < SCAN_THREADS / 64; w++) s += wave_tot[w];
    pt_block_sums[blockIdx.x] = s;
  }
}

__device__ __forceinline__ u64 pack_cell(int cnt) { return cnt > 0 ? ((1ULL << 32) | (u64)(unsigned)cnt) : 0ULL; }

// pass 1 over cells: (occupied cells, points) per 1024-cell tile, packed hi/lo in one u64
__global__ __launch_bounds__(SCAN_THREADS) void k_cell_tile_sums(const int *__restrict__ cell_count, long long cells,
                                                                 u64 *__restrict__ cell_block_sums) {
  __shared__ u64 wave_tot[SCAN_THREADS / 64];
  long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
  u64 s = 0;
  if (base + SCAN_ITEMS <= cells) {
    int4 c = *reinterpret_cast<const int4 *>(cell_count + base);
    s = pack_cell(c.x) + pack_cell(c.y) + pack_cell(c.z) + pack_cell(c.w);
  } else {
    for (int i = 0; i < SCAN_ITEMS; i++)
      if (base + i < cells) s += pack_cell(cell_count[base + i]);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
  if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 t = 0;
    for (int w = 0; w < SCAN_THREADS / 64; w++) t += wave_tot[w];
    cell_block_sums[blockIdx.x] = t;
  }
}

// pass 2: block 0 turns the cell tile sums into exclusive prefixes (and publishes P, N'); block 1 does the point tiles.
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_tile_sums(u64 *__restrict__ cell_block_sums, int n_cblk,
                                                                 int *__restrict__ pt_block_sums, int n_pblk,
                                                                 int *__restrict__ counters_ws, int *__restrict__ counters_out,
                                                                 int *__restrict__ pillar_start) {
  __shared__ u64 lds64[SCAN_THREADS / 64 + 1];
  __shared__ int lds32[SCAN_THREADS / 64 + 1];
  if (blockIdx.x == 0) {
    u64 carry = 0;
    for (int base = 0; base < n_cblk; base += SCAN_TILE) {
      u64 v[SCAN_ITEMS];
#pragma unroll
      for (int i = 0; i < SCAN_ITEMS; i++) {
        int idx = base + threadIdx.x * SCAN_ITEMS + i;
        v[i] = idx < n_cblk ? cell_block_sums[idx] : 0ULL;
      }
      u64 tot;
      block_scan_excl<u64>(v, lds64, &tot);
#pragma unroll
      for (int i = 0; i < SCAN_ITEMS; i++) {
        int idx = base + threadIdx.x * SCAN_ITEMS + i;
        if (idx < n_cblk) cell_block_sums[idx] = v[i] + carry;
      }
      carry += tot;
    }
    if (threadIdx.x == 0) {
      int P = (int)(carry >> 32), Nv = (int)(carry & 0xffffffffULL);
      counters_ws[0] = P;
      counters_ws[1] = Nv;
      counters_ws[2] = 0;
      counters_ws[3] = 0;
      if (counters_out) {
        counters_out[0] = P;
        counters_out[1] = Nv;
        counters_out[2] = 0;
        counters_out[3] = 0;
      }
      pillar_start[P] = Nv;
    }
  } else {
    int carry = 0;
    for (int base = 0; base < n_pblk; base += SCAN_TILE) {
      int v[SCAN_ITEMS];
#pragma unroll
      for (int i = 0; i < SCAN_ITEMS; i++) {
        int idx = base + threadIdx.x * SCAN_ITEMS + i;
        v[i] = idx < n_pblk ? pt_block_sums[idx] : 0;
      }
      int tot;
      block_scan_excl<int>(v, lds32, &tot);
#pragma unroll
      for (int i = 0; i < SCAN_ITEMS; i++) {
        int idx = base + threadIdx.x * SCAN_ITEMS + i;
        if (idx < n_pblk) pt_block_sums[idx] = v[i] + carry;
      }
      carry += tot;
    }
  }
}

// pass 3 over cells: rank + first slot of every cell; per-pillar outputs (coords, counts, pillar tables)
__global__ __launch_bounds__(SCAN_THREADS) void k_cell_finish(const int *__restrict__ cell_count, long long cells, pcp_grid_t g,
                                                              const u64 *__restrict__ cell_block_sums,
                                                              int *__restrict__ cell_rank, int *__restrict__ cell_start,
                                                              int *__restrict__ pillar_cell, int *__restrict__ pillar_start,
                                                              int *__restrict__ voxel_coords, int *__restrict__ unq_cnt) {
  __shared__ u64 lds64[SCAN_THREADS / 64 + 1];
  long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
  int cnt[SCAN_ITEMS];
  u64 v[SCAN_ITEMS];
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    cnt[i] = (base + i < cells) ? cell_count[base + i] : 0;
    v[i] = pack_cell(cnt[i]);
  }
  u64 tot;
  block_scan_excl<u64>(v, lds64, &tot);
  const u64 blk = cell_block_sums[blockIdx.x];
  const int plane = g.nx * g.ny;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    long long c = base + i;
    if (c >= cells) continue;
    u64 e = v[i] + blk;
    int rank = (int)(e >> 32), start = (int)(e & 0xffffffffULL);
    cell_start[c] = start;
    if (cnt[i] > 0) {
      cell_rank[c] = rank;
      pillar_cell[rank] = (int)c;
      pillar_start[rank] = start;
      int b = (int)(c / plane), rem = (int)(c % plane);
      int cx = rem / g.ny, cy = rem % g.ny;
      int4 vc = make_int4(b, 0, cy, cx);                        // [batch, z, y, x] (dynamic_pillar_vfe.py:138-143)
      *reinterpret_cast<int4 *>(voxel_coords + 4LL * rank) = vc;
      if (unq_cnt) unq_cnt[rank] = cnt[i];
    } else {
      cell_rank[c] = -1;
    }
  }
}

// pass 3 over points: stable compaction position -> unq_inv; bucket slot -> bucket_order
__global__ __launch_bounds__(SCAN_THREADS) void k_point_finish(const int *__restrict__ point_cell, long long n,
                                                               const int *__restrict__ pt_block_sums,
                                                               const int *__restrict__ cell_rank, const int *__restrict__ cell_start,
                                                               int *__restrict__ cell_fill, long long *__restrict__ unq_inv,
                                                               int *__restrict__ bucket_order) {
  __shared__ int lds32[SCAN_THREADS / 64 + 1];
  // items of one thread must be consecutive rows for a stable compaction
  long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
  int cell[SCAN_ITEMS], v[SCAN_ITEMS];
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    cell[i] = (base + i < n) ? point_cell[base + i] : -1;
    v[i] = cell[i] >= 0 ? 1 : 0;
  }
  int tot;
  block_scan_excl<int>(v, lds32, &tot);
  const int blk = pt_block_sums[blockIdx.x];
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    if (cell[i] < 0) continue;
    int pos = v[i] + blk;
    if (unq_inv) unq_inv[pos] = (long long)cell_rank[cell[i]];
    int slot = cell_start[cell[i]] + atomicAdd(&cell_fill[cell[i]], 1);
    bucket_order[slot] = (int)(base + i);
  }
}

}  // namespace

extern "C" size_t pcp_voxelize_workspace_bytes(const pcp_grid_t *grid, int64_t max_points) {
  if (!grid || max_points < 0) return 0;
  int64_t cells = (int64_t)grid->batch_size * grid->nx * grid->ny;
  return pcp_vox_layout(cells, max_points > 0 ? max_points : 1).total;
}

extern "C" int pcp_voxelize(const float *points, int64_t n, int32_t row_stride, const pcp_grid_t *grid, void *workspace,
                            size_t workspace_bytes, int32_t *voxel_coords, int64_t *unq_inv, int32_t *unq_cnt,
                            int32_t *counters, void *stream_) {
  if (!grid || !workspace || !voxel_coords || n < 0 || row_stride < 3) return PCP_ERR_ARG;
  if (n > 0 && !points) return PCP_ERR_ARG;
  if (grid->nx <= 0 || grid->ny <= 0 || grid->batch_size <= 0) return PCP_ERR_ARG;
  const int64_t cells = (int64_t)grid->batch_size * grid->nx * grid->ny;
  if (cells >= (1LL << 31) || n >= (1LL << 31)) return PCP_ERR_UNSUPPORTED;
  const int64_t n_alloc = n > 0 ? n : 1;
  VoxLayout L = pcp_vox_layout(cells, n_alloc);
  if (workspace_bytes < L.total) return PCP_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  char *ws = (char *)workspace;
  int *cell_count = (int *)(ws + L.cell_count);
  int *cell_fill = (int *)(ws + L.cell_fill);
  int *cell_rank = (int *)(ws + L.cell_rank);
  int *cell_start = (int *)(ws + L.cell_start);
  int *point_cell = (int *)(ws + L.point_cell);
  int *bucket_order = (int *)(ws + L.bucket_order);
  int *pillar_cell = (int *)(ws + L.pillar_cell);
  int *pillar_start = (int *)(ws + L.pillar_start);
  const int n_cblk = (int)((cells + SCAN_TILE - 1) / SCAN_TILE);
  const int n_pblk = (int)((n + SCAN_TILE - 1) / SCAN_TILE);
  u64 *cell_bs = (u64 *)(ws + L.block_sums);
  int *pt_bs = (int *)(ws + L.block_sums + (size_t)n_cblk * 8 + 8);
  int *counters_ws = (int *)(ws + L.counters);

  // cell_count and cell_fill are adjacent (layout keeps 256-B alignment between them): one memset
  if (pcp_zero_async(cell_count, L.cell_rank - L.cell_count, stream) != PCP_OK) return PCP_ERR_LAUNCH;
  if (n_pblk > 0) {
    hipLaunchKernelGGL(k_point_cells, dim3(n_pblk), dim3(SCAN_THREADS), 0, stream, points, (long long)n, (int)row_stride,
                       *grid, cell_count, point_cell, pt_bs);
    PCP_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(k_cell_tile_sums, dim3(n_cblk), dim3(SCAN_THREADS), 0, stream, cell_count, (long long)cells, cell_bs);
  PCP_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_scan_tile_sums, dim3(2), dim3(SCAN_THREADS), 0, stream, cell_bs, n_cblk, pt_bs, n_pblk, counters_ws,
                     counters, pillar_start);
  PCP_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_cell_finish, dim3(n_cblk), dim3(SCAN_THREADS), 0, stream, cell_count, (long long)cells, *grid, cell_bs,
                     cell_rank, cell_start, pillar_cell, pillar_start, voxel_coords, unq_cnt);
  PCP_CHECK_LAUNCH();
  if (n_pblk > 0) {
    hipLaunchKernelGGL(k_point_finish, dim3(n_pblk), dim3(SCAN_THREADS), 0, stream, point_cell, (long long)n, pt_bs, cell_rank,
                       cell_start, cell_fill, (long long *)unq_inv, bucket_order);
    PCP_CHECK_LAUNCH();
  }
  return PCP_OK;
}


// ---- a processing order for per-point kernels that gather from BEV maps (HunterJr point head) -------------------------------------
// order[0 .. N') = the bucket order (points grouped by pillar, pillars ascending in (frame, x, y): neighbours in this order
// sample neighbouring BEV pixels, so their 4 x C-float gathers hit L2 instead of HBM); order[N' .. n) = the rows the pillariser
// masked (out of range / foreign frame), in arbitrary order.  Every row appears exactly once.
namespace {
__global__ __launch_bounds__(256) void k_row_order(const int *__restrict__ point_cell, const int *__restrict__ bucket_order,
                                                  const int *__restrict__ counters, long long n, int *__restrict__ order,
                                                  int *__restrict__ cursor) {
  __shared__ int wave_cnt[4];
  __shared__ int block_base;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int kept = counters[1];
  if (t < n && t < kept) order[t] = bucket_order[t];
  const int rej = (t < n && point_cell[t] < 0) ? 1 : 0;
  // one atomic per workgroup (a single hot cursor serialises: 7 k per-thread atomics cost 40 us)
  const unsigned long long bal = __ballot(rej);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) wave_cnt[wv] = __popcll(bal);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int tot = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    block_base = tot ? atomicAdd(cursor, tot) : 0;
  }
  __syncthreads();
  if (rej) {
    int off = block_base + __popcll(bal & ((1ull << lane) - 1ull));
    for (int w = 0; w < wv; ++w) off += wave_cnt[w];
    order[kept + off] = (int)t;
  }
}
}  // namespace

extern "C" int pcp_voxelize_row_order(const pcp_grid_t *grid, const void *workspace, int64_t n, int32_t *order, int32_t *cursor_scratch,
                                      void *stream_) {
  if (!grid || !workspace || !order || !cursor_scratch || n < 0) return PCP_ERR_ARG;
  if (n == 0) return PCP_OK;
  const int64_t cells = (int64_t)grid->batch_size * grid->nx * grid->ny;
  const VoxLayout L = pcp_vox_layout(cells, n);
  const char *ws = (const char *)workspace;
  hipStream_t stream = (hipStream_t)stream_;
  if (pcp_zero_async(cursor_scratch, sizeof(int32_t), stream) != PCP_OK) return PCP_ERR_LAUNCH;
  hipLaunchKernelGGL(k_row_order, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const int *)(ws + L.point_cell),
                     (const int *)(ws + L.bucket_order), (const int *)(ws + L.counters), (long long)n, order, cursor_scratch);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}
