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
#include <stdlib.h>
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

__device__ __forceinline__ u64 pack_cell(int cnt) { return cnt > 0 ? ((1ULL << 32) | (u64)(unsigned)cnt) : 0ULL; }

// ---- round 5: the cell passes -------------------------------------------------------------------------------------------------------------
// Round 4 ran three launches (tile sums, a one-workgroup scan of them, per-cell outputs).  A single-pass decoupled look-back scan was built
// and measured first this round: 25 us per 1 M cells, 69 us per 5.2 M -- every workgroup polls the same few status lines of one L2 channel
// (512 x 256 polls per round).  What runs now has NO inter-workgroup dependency: k_cell_tile_sums leaves one (pillars, points) pair per
// 2048-cell tile, and every workgroup of k_cell_finish adds up the pairs in front of its own tile itself (at most 2 560 x 8 bytes, coalesced,
// from L2) before it scans its tile.
constexpr int CS_ITEMS = 8;
constexpr int CS_TILE = SCAN_THREADS * CS_ITEMS;     // 2048 cells per workgroup

struct CellScanOut {
  int *cell_rank, *pillar_cell, *pillar_start, *voxel_coords, *unq_cnt, *counters_ws, *counters_out;
  int2 *cell_rs;
  int2 *tile_desc;               // rows mode only
  int4 *crowd_list;              // rows mode only: pillars of >= crowd records {first slot, records, rank, canvas row}; counters_ws[4] of them
  int *long_list;                // ranks of the pillars of more than PCP_LONG_PILLAR points; counters_ws[6] of them
  int crowd;                     // threshold (0: none)
};
constexpr int CROWD_TAG = 0x40000000;      // on a cell's first slot in cell_rs: its records carry the rank with the sign bit set

__device__ __forceinline__ void load_cell_counts(const int *__restrict__ cell_count, long long base, long long cells, int (&cnt)[CS_ITEMS]) {
  if (base + CS_ITEMS <= cells) {
    const int4 a = *reinterpret_cast<const int4 *>(cell_count + base), b = *reinterpret_cast<const int4 *>(cell_count + base + 4);
    cnt[0] = a.x; cnt[1] = a.y; cnt[2] = a.z; cnt[3] = a.w; cnt[4] = b.x; cnt[5] = b.y; cnt[6] = b.z; cnt[7] = b.w;
  } else {
#pragma unroll
    for (int i = 0; i < CS_ITEMS; i++) cnt[i] = (base + i < cells) ? cell_count[base + i] : 0;
  }
}

// SEG (rows mode): pillars of exactly one point are counted apart -- their records go behind those of all multi-point pillars, where
// pcp_pfn_rows runs them through a path without per-pillar reductions.  The scanned pair is then (pillars << 32 | slots of the multi-point
// pillars) plus a 32-bit count of singles; without SEG every pillar counts as "multi" and the singles count stays zero.
template <bool SEG>
__device__ __forceinline__ u64 pack_cell_seg(int cnt) {
  if (cnt <= 0) return 0ULL;
  return (1ULL << 32) | (u64)(unsigned)((SEG && cnt == 1) ? 0 : cnt);
}

template <bool SEG>
__global__ __launch_bounds__(SCAN_THREADS) void k_cell_tile_sums(const int *__restrict__ cell_count, long long cells, u64 *__restrict__ tile_sums,
                                                                 unsigned *__restrict__ tile_singles, int *__restrict__ crowd_count) {
  __shared__ u64 wave_tot[SCAN_THREADS / 64];
  __shared__ unsigned wave_one[SCAN_THREADS / 64];
  if (crowd_count && blockIdx.x == 0 && threadIdx.x == 0) {                      // k_cell_finish (the next launch) appends to the lists
    crowd_count[0] = 0;                                                           // counters_ws[4]: crowded pillars (rows mode)
    crowd_count[2] = 0;                                                           // counters_ws[6]: long pillars
  }
  int cnt[CS_ITEMS];
  load_cell_counts(cell_count, (long long)blockIdx.x * CS_TILE + (long long)threadIdx.x * CS_ITEMS, cells, cnt);
  u64 s = 0;
  unsigned ones = 0;
#pragma unroll
  for (int i = 0; i < CS_ITEMS; i++) {
    s += pack_cell_seg<SEG>(cnt[i]);
    ones += (SEG && cnt[i] == 1) ? 1u : 0u;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    s += __shfl_xor(s, d, 64);
    ones += __shfl_xor(ones, d, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    wave_tot[threadIdx.x >> 6] = s;
    wave_one[threadIdx.x >> 6] = ones;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 t = 0;
    unsigned t1 = 0;
    for (int w = 0; w < SCAN_THREADS / 64; w++) {
      t += wave_tot[w];
      t1 += wave_one[w];
    }
    tile_sums[blockIdx.x] = t;
    tile_singles[blockIdx.x] = t1;
  }
}

template <bool SEG>
__global__ __launch_bounds__(SCAN_THREADS) void k_cell_finish(const int *__restrict__ cell_count, long long cells, pcp_grid_t g,
                                                              const u64 *__restrict__ tile_sums, const unsigned *__restrict__ tile_singles,
                                                              int n_tiles, CellScanOut o) {
  __shared__ u64 lds64[SCAN_THREADS / 64 + 1];
  __shared__ unsigned lds32[SCAN_THREADS / 64 + 1];
  __shared__ u64 red[SCAN_THREADS / 64];
  __shared__ unsigned red1[SCAN_THREADS / 64];
  const int tile = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long base = (long long)tile * CS_TILE + (long long)threadIdx.x * CS_ITEMS;
  int cnt[CS_ITEMS];
  load_cell_counts(cell_count, base, cells, cnt);
  // the sums of the tiles in front of this one
  u64 pre = 0;
  unsigned pre1 = 0;
  for (int i = threadIdx.x; i < tile; i += SCAN_THREADS) {
    pre += tile_sums[i];
    if (SEG) pre1 += tile_singles[i];
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    pre += __shfl_xor(pre, d, 64);
    if (SEG) pre1 += __shfl_xor(pre1, d, 64);
  }
  if (lane == 0) {
    red[wave] = pre;
    red1[wave] = pre1;
  }
  // exclusive scan of the (pillars, multi slots) pairs and of the singles inside the workgroup
  u64 v[CS_ITEMS], local = 0;
  unsigned v1[CS_ITEMS], local1 = 0;
#pragma unroll
  for (int i = 0; i < CS_ITEMS; i++) {
    v[i] = local;
    v1[i] = local1;
    local += pack_cell_seg<SEG>(cnt[i]);
    local1 += (SEG && cnt[i] == 1) ? 1u : 0u;
  }
  u64 incl = local;
  unsigned incl1 = local1;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const u64 up = __shfl_up(incl, d, 64);
    const unsigned up1 = SEG ? __shfl_up(incl1, d, 64) : 0u;
    if (lane >= d) {
      incl += up;
      incl1 += up1;
    }
  }
  if (lane == 63) {
    lds64[wave] = incl;
    lds32[wave] = incl1;
  }
  __syncthreads();
  u64 wave_base = 0, agg = 0, excl = 0;
  unsigned wave_base1 = 0, agg1 = 0, excl1 = 0;
#pragma unroll
  for (int w = 0; w < SCAN_THREADS / 64; w++) {
    const u64 sw = lds64[w];
    const unsigned sw1 = lds32[w];
    if (w < wave) {
      wave_base += sw;
      wave_base1 += sw1;
    }
    agg += sw;
    agg1 += sw1;
    excl += red[w];
    excl1 += red1[w];
  }
  if (threadIdx.x == 0) {
    if (tile == n_tiles - 1) {
      const u64 tot = excl + agg;
      const int P = (int)(tot >> 32), Nm = (int)(tot & 0xffffffffULL), S = (int)(excl1 + agg1);
      // counters: pillars, kept points, records of multi-point pillars (= kept points unless singles are set apart), single-point pillars
      o.counters_ws[0] = P; o.counters_ws[1] = Nm + S; o.counters_ws[2] = Nm; o.counters_ws[3] = S;
      if (o.counters_out) { o.counters_out[0] = P; o.counters_out[1] = Nm + S; o.counters_out[2] = SEG ? Nm : 0; o.counters_out[3] = SEG ? S : 0; }
      o.pillar_start[P] = Nm + S;
    }
    if (tile == 0 && o.tile_desc) o.tile_desc[0] = make_int2(0, 0);
  }
  const u64 blk = excl + wave_base + incl - local;
  const unsigned blk1 = excl1 + wave_base1 + incl1 - local1;
  const int plane = g.nx * g.ny;
  int rank_out[CS_ITEMS];
#pragma unroll
  for (int i = 0; i < CS_ITEMS; i++) {
    const long long c = base + i;
    rank_out[i] = -1;
    if (c < cells && cnt[i] > 0) {
      const u64 e = v[i] + blk;
      const int rank = (int)(e >> 32);
      const bool single = SEG && cnt[i] == 1;
      // first slot: a multi-point pillar's run among the multi-point records; a single's index among the singles, tagged (its slot is that
      // index behind ALL multi-point records, whose number only the last workgroup knows: the point pass adds counters[2])
      const int start = single ? (int)(0x80000000u | (v1[i] + blk1)) : (int)(e & 0xffffffffULL);
      rank_out[i] = rank;
      const bool crowded = o.crowd_list && o.crowd > 0 && cnt[i] >= o.crowd;
      o.cell_rs[c] = make_int2(rank, crowded ? (start | CROWD_TAG) : start);
      if (cnt[i] > PCP_LONG_PILLAR) o.long_list[atomicAdd(&o.counters_ws[6], 1)] = rank;
      if (crowded) {
        const int b = (int)(c / plane), rem = (int)(c % plane);
        const int cx = rem / g.ny, cy = rem % g.ny;
        const int at = atomicAdd(&o.counters_ws[4], 1);
        o.crowd_list[at] = make_int4(start, cnt[i], rank, (b * g.ny + cy) * g.nx + cx);
      }
      o.pillar_cell[rank] = (int)c;
      o.pillar_start[rank] = start;
      if (o.voxel_coords) {
        const int b = (int)(c / plane), rem = (int)(c % plane);
        const int cx = rem / g.ny, cy = rem % g.ny;
        *reinterpret_cast<int4 *>(o.voxel_coords + 4LL * rank) = make_int4(b, 0, cy, cx);       // [batch, z, y, x] (dynamic_pillar_vfe.py:138-143)
      }
      if (o.unq_cnt) o.unq_cnt[rank] = cnt[i];
      if (o.tile_desc && !single) {
        // the NEXT multi-point pillar is the first one of every wave tile whose first slot lies in (start, start + cnt]: {its first
        // record, that record's slot}; the record carries the pillar's rank
        const int end = start + cnt[i];
        for (int t = start / PCP_PFN_TILE + 1; t <= end / PCP_PFN_TILE; ++t) o.tile_desc[t] = make_int2(end, end);
      }
    }
  }
  // the cell -> rank table (every cell: the sparse first layer and the PFN's gap fill read empty cells too) as two 16-byte stores
  if (base + CS_ITEMS <= cells) {
    *reinterpret_cast<int4 *>(o.cell_rank + base) = make_int4(rank_out[0], rank_out[1], rank_out[2], rank_out[3]);
    *reinterpret_cast<int4 *>(o.cell_rank + base + 4) = make_int4(rank_out[4], rank_out[5], rank_out[6], rank_out[7]);
  } else {
#pragma unroll
    for (int i = 0; i < CS_ITEMS; i++)
      if (base + i < cells) o.cell_rank[base + i] = rank_out[i];
  }
}

// exclusive prefix over the per-tile kept-row counts (only the unq_inv path needs it)
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_point_tiles(int *__restrict__ pt_block_sums, int n_pblk) {
  __shared__ int lds32[SCAN_THREADS / 64 + 1];
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

// the row of the bucket order a kept point lands in; rows mode: the point's row is written there, ready for pcp_pfn_rows
template <int RS>
__device__ __forceinline__ void emit_point(const float *__restrict__ points, int stride, int num_raw, const pcp_grid_t &g, long long r, int cell,
                                           int rank, int slot, int *__restrict__ bucket_order, float *__restrict__ srows) {
  if (bucket_order) bucket_order[slot] = (int)r;
  if (RS > 0) {
    const float *row = points + r * stride;
    const int plane = g.nx * g.ny;
    const int b = cell / plane, rem = cell - b * plane;
    const int cx = rem / g.ny, cy = rem - cx * g.ny;
    float v[RS > 0 ? RS : 1];
#pragma unroll
    for (int k = 0; k < RS - 3; k++) v[k] = k < num_raw ? row[1 + k] : 0.f;
    v[RS - 3] = __int_as_float(rank);
    v[RS - 2] = __int_as_float((cx << 16) | cy);
    v[RS - 1] = __int_as_float((b * g.ny + cy) * g.nx + cx);      // row of the (B, ny, nx, 64) canvas
    f32x4 *dst = reinterpret_cast<f32x4 *>(srows + (long long)slot * RS);
#pragma unroll
    for (int q = 0; q < RS / 4; q++) dst[q] = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
  }
}

// last pass over points, one thread per row (no unq_inv)
template <int RS>
__global__ __launch_bounds__(SCAN_THREADS) void k_point_place(const float *__restrict__ points, long long n, int stride, int num_raw, pcp_grid_t g,
                                                              const int *__restrict__ point_cell, const int *__restrict__ point_rank,
                                                              const int2 *__restrict__ cell_rs, const int *__restrict__ counters,
                                                              int *__restrict__ bucket_order, float *__restrict__ srows) {
  const long long r = (long long)blockIdx.x * SCAN_THREADS + threadIdx.x;
  if (r >= n) return;
  const int cell = point_cell[r];
  if (cell < 0) return;
  const int2 rs = cell_rs[cell];
  const bool crowded = rs.y >= 0 && (rs.y & CROWD_TAG);
  const int slot = rs.y < 0 ? counters[2] + (rs.y & 0x7fffffff) : (rs.y & ~CROWD_TAG) + point_rank[r];      // tagged: a single-point pillar (rows mode)
  emit_point<RS>(points, stride, num_raw, g, r, cell, crowded ? (int)(rs.x | 0x80000000u) : rs.x, slot, bucket_order, srows);
}

// last pass over points with the stable compaction position -> unq_inv
template <int RS>
__global__ __launch_bounds__(SCAN_THREADS) void k_point_finish(const float *__restrict__ points, long long n, int stride, int num_raw, pcp_grid_t g,
                                                               const int *__restrict__ point_cell, const int *__restrict__ pt_block_sums,
                                                               const int2 *__restrict__ cell_rs, const int *__restrict__ counters,
                                                               const int *__restrict__ point_rank, long long *__restrict__ unq_inv,
                                                               int *__restrict__ bucket_order, float *__restrict__ srows) {
  __shared__ int lds32[SCAN_THREADS / 64 + 1];
  // items of one thread must be consecutive rows for a stable compaction
  long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
  int cell[SCAN_ITEMS], v[SCAN_ITEMS], prank[SCAN_ITEMS];
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    cell[i] = (base + i < n) ? point_cell[base + i] : -1;
    prank[i] = cell[i] >= 0 ? point_rank[base + i] : 0;
    v[i] = cell[i] >= 0 ? 1 : 0;
  }
  int tot;
  block_scan_excl<int>(v, lds32, &tot);
  const int blk = pt_block_sums[blockIdx.x];
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    if (cell[i] < 0) continue;
    const int2 rs = cell_rs[cell[i]];
    unq_inv[v[i] + blk] = (long long)rs.x;
    const bool crowded = rs.y >= 0 && (rs.y & CROWD_TAG);
    const int slot = rs.y < 0 ? counters[2] + (rs.y & 0x7fffffff) : (rs.y & ~CROWD_TAG) + prank[i];
    emit_point<RS>(points, stride, num_raw, g, base + i, cell[i], crowded ? (int)(rs.x | 0x80000000u) : rs.x, slot, bucket_order, srows);
  }
}

}  // namespace

extern "C" size_t pcp_voxelize_workspace_bytes(const pcp_grid_t *grid, int64_t max_points) {
  if (!grid || max_points < 0) return 0;
  int64_t cells = (int64_t)grid->batch_size * grid->nx * grid->ny;
  return pcp_vox_layout(cells, max_points > 0 ? max_points : 1).total;
}

extern "C" size_t pcp_pillarise_rows_workspace_bytes(const pcp_grid_t *grid, int64_t max_points, int32_t num_raw) {
  if (!grid || max_points < 0 || num_raw < 3 || num_raw > 13) return 0;
  int64_t cells = (int64_t)grid->batch_size * grid->nx * grid->ny;
  return pcp_rows_layout(cells, max_points > 0 ? max_points : 1, num_raw).total;
}

namespace {

struct ScanScratch { u64 *tile_sums; unsigned *tile_singles; int *pt_bs; int n_ctiles, n_pblk; };
inline ScanScratch scan_scratch(char *ws, const VoxLayout &L, int64_t cells, int64_t n) {
  ScanScratch s;
  s.n_ctiles = (int)((cells + CS_TILE - 1) / CS_TILE);
  s.n_pblk = (int)((n + SCAN_TILE - 1) / SCAN_TILE);
  s.tile_sums = (u64 *)(ws + L.block_sums);
  s.tile_singles = (unsigned *)(ws + L.block_sums + pcp_align_up((size_t)s.n_ctiles * 8 + 8, 16));
  s.pt_bs = (int *)(ws + L.block_sums + pcp_align_up((size_t)s.n_ctiles * 8 + 8, 16) + pcp_align_up((size_t)s.n_ctiles * 4 + 4, 16));
  return s;
}

// pass 1 over points: cell id per row, per-cell histogram (its return value = the row's slot inside its cell), kept rows per 1024-row tile.
// AGG (round 6): the 1024 rows of a workgroup are first counted in an LDS hash table keyed by cell (open addressing, 2048 slots), then ONE global
// atomic per distinct (workgroup, cell) pair reserves that many slots of the cell and every row takes base + its rank in the table entry.
// A uniform cloud gains nothing (its 1024 rows hit 1024 different cells: +4 us on 1.44 M rows), but the cell under a LiDAR holds hundreds of
// points per frame and agent, and same-address atomics serialise at the memory side: pcp_pillarise_rows on the 6-agent ring cloud 204 -> 136 us
// (profiles/r06_vox_aggregate_ab.txt).  The slot order inside a cell stays "arrival order", which nothing downstream depends on (exact
// fixed-point sums, order-free maxima; the training path sorts the rows of a pillar by row index).  Not applied to the makers' compaction
// (k_stc_scatter): in the agents' own frames every (agent, frame) has its own hot cell, the contention is a sixth of the merged cloud's, and
// the table cost that register-heavy kernel 17 us per launch on the uniform cloud for nothing on the ring (measured, reverted).
constexpr int VOX_HT = 2048;

template <bool AGG>
__global__ __launch_bounds__(SCAN_THREADS) void k_point_cells(const float *__restrict__ points, long long n, int stride,
                                                              pcp_grid_t g, int *__restrict__ cell_count,
                                                              int *__restrict__ point_cell, int *__restrict__ point_rank,
                                                              int *__restrict__ pt_block_sums) {
  __shared__ int wave_tot[SCAN_THREADS / 64];
  __shared__ int h_key[AGG ? VOX_HT : 1], h_cnt[AGG ? VOX_HT : 1];
  long long base = (long long)blockIdx.x * SCAN_TILE;
  int valid = 0;
  if (AGG) {
    for (int e = threadIdx.x; e < VOX_HT; e += SCAN_THREADS) {
      h_key[e] = -1;
      h_cnt[e] = 0;
    }
    __syncthreads();
  }
  int hslot[SCAN_ITEMS], hloc[SCAN_ITEMS];
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    long long r = base + i * SCAN_THREADS + threadIdx.x;     // coalesced over rows
    hslot[i] = -1;
    hloc[i] = 0;
    if (r < n) {
      int c = point_to_cell(points + r * stride, g);
      point_cell[r] = c;
      if (c >= 0) {
        if (AGG) {
          unsigned h = ((unsigned)c * 2654435761u) >> 21;                    // 11 bits
          while (true) {
            const int prev = atomicCAS(&h_key[h], -1, c);
            if (prev == -1 || prev == c) break;
            h = (h + 1) & (VOX_HT - 1);
          }
          hslot[i] = (int)h;
          hloc[i] = atomicAdd(&h_cnt[h], 1);
        } else {
          // the histogram atomic's return value IS the row's slot inside its cell: the last pass needs no second atomic pass (and no
          // second zeroed table); the order inside a cell is arrival order either way
          point_rank[r] = atomicAdd(&cell_count[c], 1);
        }
        valid++;
      }
    }
  }
  if (AGG) {
    __syncthreads();
    for (int e = threadIdx.x; e < VOX_HT; e += SCAN_THREADS) {
      const int c = h_key[e];
      if (c >= 0) h_cnt[e] = atomicAdd(&cell_count[c], h_cnt[e]);            // the entry now holds the first slot its rows take in the cell
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++)
      if (hslot[i] >= 0) point_rank[base + i * SCAN_THREADS + threadIdx.x] = h_cnt[hslot[i]] + hloc[i];
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

// pillars of at least this many records are listed for k_pfn_crowd (pcp_common.h); PCP_OPT_PFN_CROWD overrides (0: never)
inline int crowd_threshold() {
  const long long v = pcp_option(PCP_OPT_PFN_CROWD, PCP_PFN_CROWD);
  if (v <= 0) return 0;
  return v < PCP_PFN_CROWD_MIN ? PCP_PFN_CROWD_MIN : (int)(v > 0x3fffffff ? 0x3fffffff : v);
}

// The pillariser: zero fill of the histogram, k_point_cells, k_cell_tile_sums, k_cell_finish, k_point_place (five launches; round 4: six).
// cells_ready: cell_count / point_cell / point_rank were filled by the caller's own pass over the rows (pcp_select_transform_compact emits them
// while the transformed row is in registers): three launches.  The stable compaction position (unq_inv)
// needs k_point_cells' per-tile sums and is not available in that mode.  rows_raw > 0: rows mode (RowsLayout workspace).
int vox_passes(const float *points, int64_t n, int32_t row_stride, const pcp_grid_t *grid, void *workspace, size_t workspace_bytes,
               int32_t *voxel_coords, int64_t *unq_inv, int32_t *unq_cnt, int32_t *counters, hipStream_t stream, bool cells_ready,
               int rows_raw, bool want_bucket_order) {
  if (!grid || !workspace || n < 0 || row_stride < 3) return PCP_ERR_ARG;
  if (rows_raw == 0 && !voxel_coords) return PCP_ERR_ARG;
  if (n > 0 && !points) return PCP_ERR_ARG;
  if (grid->nx <= 0 || grid->ny <= 0 || grid->batch_size <= 0) return PCP_ERR_ARG;
  if (cells_ready && unq_inv) return PCP_ERR_ARG;
  if (rows_raw && (rows_raw < 3 || rows_raw > 13 || row_stride < 1 + rows_raw || grid->nx > 65535 || grid->ny > 65535)) return PCP_ERR_ARG;
  const int64_t cells = (int64_t)grid->batch_size * grid->nx * grid->ny;
  if (cells >= (1LL << 31) || n >= (1LL << 31)) return PCP_ERR_UNSUPPORTED;
  if (rows_raw && n >= (1LL << 30)) return PCP_ERR_UNSUPPORTED;                   // bit 30 of a first slot tags the crowded pillars
  const int64_t n_alloc = n > 0 ? n : 1;
  VoxLayout L = pcp_vox_layout(cells, n_alloc);
  RowsLayout R;
  if (rows_raw) {
    R = pcp_rows_layout(cells, n_alloc, rows_raw);
    if (workspace_bytes < R.total) return PCP_ERR_WORKSPACE;
  } else if (workspace_bytes < L.total) {
    return PCP_ERR_WORKSPACE;
  }
  char *ws = (char *)workspace;
  int *cell_count = (int *)(ws + L.cell_count);
  int *point_rank = (int *)(ws + L.point_rank);
  int *point_cell = (int *)(ws + L.point_cell);
  const ScanScratch sc = scan_scratch(ws, L, cells, n);
  CellScanOut o;
  o.cell_rank = (int *)(ws + L.cell_rank);
  o.cell_rs = (int2 *)(ws + L.cell_rs);
  o.pillar_cell = (int *)(ws + L.pillar_cell);
  o.pillar_start = (int *)(ws + L.pillar_start);
  o.voxel_coords = voxel_coords;
  o.unq_cnt = unq_cnt;
  o.counters_ws = (int *)(ws + L.counters);
  o.counters_out = counters;
  o.tile_desc = rows_raw ? (int2 *)(ws + R.tile_desc) : nullptr;
  o.crowd_list = rows_raw ? (int4 *)(ws + R.crowd_list) : nullptr;
  o.crowd = rows_raw ? crowd_threshold() : 0;
  o.long_list = (int *)(ws + L.long_list);
  int *crowd_count = o.counters_ws + 4;              // the two list lengths (counters_ws[4], [6]) are zeroed by k_cell_tile_sums in every mode
  int *bucket_order = (want_bucket_order || !rows_raw) ? (int *)(ws + L.bucket_order) : nullptr;
  float *srows = rows_raw ? (float *)(ws + R.srows) : nullptr;
  const int rs = rows_raw ? pcp_rows_stride(rows_raw) : 0;

  if (!cells_ready) {
    if (pcp_zero_async(cell_count, (size_t)cells * 4, stream) != PCP_OK) return PCP_ERR_LAUNCH;
    if (sc.n_pblk > 0) {
      if (pcp_option(PCP_OPT_VOX_AGGREGATE, 1) != 0)
        hipLaunchKernelGGL(k_point_cells<true>, dim3(sc.n_pblk), dim3(SCAN_THREADS), 0, stream, points, (long long)n, (int)row_stride,
                           *grid, cell_count, point_cell, point_rank, sc.pt_bs);
      else
        hipLaunchKernelGGL(k_point_cells<false>, dim3(sc.n_pblk), dim3(SCAN_THREADS), 0, stream, points, (long long)n, (int)row_stride,
                           *grid, cell_count, point_cell, point_rank, sc.pt_bs);
      PCP_CHECK_LAUNCH();
    }
  }
  // Rows mode sets the single-point pillars apart (pcp_pfn_rows runs them without per-pillar reductions) -- unless the caller also wants the
  // bucket order: its consumer (HunterJr's point head) walks it for the spatial locality of its BEV gathers, and two interleaved sweeps over
  // the map (multi-point pillars, then singles) cost that kernel 15 % (measured on configs 1 - 3) where the PFN gains a third of that.
  if (rows_raw && !want_bucket_order) {
    hipLaunchKernelGGL(k_cell_tile_sums<true>, dim3(sc.n_ctiles), dim3(SCAN_THREADS), 0, stream, cell_count, (long long)cells, sc.tile_sums,
                       sc.tile_singles, crowd_count);
    PCP_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_cell_finish<true>, dim3(sc.n_ctiles), dim3(SCAN_THREADS), 0, stream, cell_count, (long long)cells, *grid,
                       sc.tile_sums, sc.tile_singles, sc.n_ctiles, o);
  } else {
    hipLaunchKernelGGL(k_cell_tile_sums<false>, dim3(sc.n_ctiles), dim3(SCAN_THREADS), 0, stream, cell_count, (long long)cells, sc.tile_sums,
                       sc.tile_singles, crowd_count);
    PCP_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_cell_finish<false>, dim3(sc.n_ctiles), dim3(SCAN_THREADS), 0, stream, cell_count, (long long)cells, *grid,
                       sc.tile_sums, sc.tile_singles, sc.n_ctiles, o);
  }
  PCP_CHECK_LAUNCH();
  if (sc.n_pblk > 0) {
    const int nb1 = (int)((n + SCAN_THREADS - 1) / SCAN_THREADS);
    if (unq_inv) {
      hipLaunchKernelGGL(k_scan_point_tiles, dim3(1), dim3(SCAN_THREADS), 0, stream, sc.pt_bs, sc.n_pblk);
      PCP_CHECK_LAUNCH();
#define PCP_FINISH(RS_)                                                                                                                      \
  hipLaunchKernelGGL(k_point_finish<RS_>, dim3(sc.n_pblk), dim3(SCAN_THREADS), 0, stream, points, (long long)n, (int)row_stride, rows_raw,   \
                     *grid, point_cell, sc.pt_bs, o.cell_rs, o.counters_ws, point_rank, (long long *)unq_inv, bucket_order, srows)
      if (rs == 0) PCP_FINISH(0); else if (rs == 8) PCP_FINISH(8); else PCP_FINISH(16);
#undef PCP_FINISH
    } else {
#define PCP_PLACE(RS_)                                                                                                                       \
  hipLaunchKernelGGL(k_point_place<RS_>, dim3(nb1), dim3(SCAN_THREADS), 0, stream, points, (long long)n, (int)row_stride, rows_raw, *grid,    \
                     point_cell, point_rank, o.cell_rs, o.counters_ws, bucket_order, srows)
      if (rs == 0) PCP_PLACE(0); else if (rs == 8) PCP_PLACE(8); else PCP_PLACE(16);
#undef PCP_PLACE
    }
    PCP_CHECK_LAUNCH();
  }
  return PCP_OK;
}

}  // namespace

extern "C" int pcp_voxelize(const float *points, int64_t n, int32_t row_stride, const pcp_grid_t *grid, void *workspace,
                            size_t workspace_bytes, int32_t *voxel_coords, int64_t *unq_inv, int32_t *unq_cnt,
                            int32_t *counters, void *stream_) {
  return vox_passes(points, n, row_stride, grid, workspace, workspace_bytes, voxel_coords, unq_inv, unq_cnt, counters,
                    (hipStream_t)stream_, false, 0, true);
}

extern "C" int pcp_voxelize_cells_ready(const float *points, int64_t n, int32_t row_stride, const pcp_grid_t *grid, void *workspace,
                                        size_t workspace_bytes, int32_t *voxel_coords, int32_t *unq_cnt, int32_t *counters,
                                        void *stream_) {
  return vox_passes(points, n, row_stride, grid, workspace, workspace_bytes, voxel_coords, nullptr, unq_cnt, counters,
                    (hipStream_t)stream_, true, 0, true);
}

extern "C" int pcp_pillarise_rows(const float *points, int64_t n, int32_t row_stride, int32_t num_raw, const pcp_grid_t *grid, void *workspace,
                                  size_t workspace_bytes, int32_t *voxel_coords, int64_t *unq_inv, int32_t *unq_cnt, int32_t *counters,
                                  int32_t flags, void *stream_) {
  if (num_raw < 3 || (flags & ~3)) return PCP_ERR_ARG;
  return vox_passes(points, n, row_stride, grid, workspace, workspace_bytes, voxel_coords, unq_inv, unq_cnt, counters, (hipStream_t)stream_,
                    (flags & PCP_ROWS_CELLS_READY) != 0, num_raw, (flags & PCP_ROWS_BUCKET_ORDER) != 0);
}


// ---- a13 (round 3): BEVMaker input preparation as a STABLE COMPACTION --------------------------------------------------------------
// Reference: pcdet/models/bev_layers/bev_maker.py:168-190 -- per agent `points[mask]` (boolean-mask copy, input order kept), per frame
// `p @ R^T + t`, one pass of the frozen chain per agent.  Round 2 stacked the agents into one pass by writing, per agent, a FULL copy of
// the cloud with the foreign rows masked (frame index -1): the pillariser then scanned slots x n rows to keep n.  Here the rows of all
// agent slots of a pass are written once, compactly: slot s's rows (input order) directly behind slot s-1's, frame index + s * batch,
// xyz mapped by the (slot, frame) pose -- exactly cat_s(transform(points[mask_s])).  Rows [total, out_rows) are filled with frame
// index -1 (the host sizes `out` from the per-agent row counts of pcp_column_id_counts; rows of absent frames make total smaller).
// With a pillariser workspace the transformed row's cell id is emitted while the row is in registers (k_point_cells fused).
//   k_stc_count   per 1024-row tile and slot: rows kept                          (reads the frame and agent columns)
//   (round 5: no scan launch -- every workgroup of k_stc_scatter adds the per-tile counts in front of it up itself)
//   k_stc_scatter rank inside the tile (ballot prefix, item-major row order), transform, write, [cell id + histogram], tail fill
namespace {

constexpr int STC_MAX_SLOTS = 8;
constexpr int STC_MAX_POSES = 64;                    // n_slots * batch: the table travels as a kernel argument (< 4 KB)
struct StcTable {
  float m[STC_MAX_POSES][12];
  unsigned char present[STC_MAX_POSES];
  int agent[STC_MAX_SLOTS];                          // ids compared after truncation, as the reference's points[:, -1].long() (bev_maker.py:154,169)
  int n_slots, batch;
};

// dev_present / dev_m (nullable, pcp_select_transform_compact_dev): the presence flags and poses in DEVICE memory instead of the kernel argument, so
// that a captured hipGraph serves every pose set (the host refreshes the two small tables before it replays the graph)
__device__ __forceinline__ int stc_slot_of(const float *row, int agent_col, const StcTable &t, const unsigned char *__restrict__ dev_present = nullptr) {
  const float a = row[agent_col];
  const int ai = (a > -1.f && a < 64.f) ? (int)a : -1;      // ids outside 0..63 never reach here (pcp_column_id_counts reports them)
  const int b = (int)row[0];
  int s = -1;
#pragma unroll
  for (int k = 0; k < STC_MAX_SLOTS; k++)
    if (k < t.n_slots && ai == t.agent[k]) s = k;
  if (s >= 0 && !(b >= 0 && b < t.batch && (dev_present ? dev_present[s * t.batch + b] : t.present[s * t.batch + b]))) s = -1;
  return s;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_stc_count(const float *__restrict__ points, long long n, int stride, int agent_col,
                                                            StcTable t, int n_tiles, int *__restrict__ tile_cnt,
                                                            const unsigned char *__restrict__ dev_present) {
  __shared__ int cnt[SCAN_ITEMS][SCAN_THREADS / 64][STC_MAX_SLOTS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long base = (long long)blockIdx.x * SCAN_TILE;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    const long long r = base + i * SCAN_THREADS + threadIdx.x;
    const int s = r < n ? stc_slot_of(points + r * stride, agent_col, t, dev_present) : -1;
#pragma unroll
    for (int k = 0; k < STC_MAX_SLOTS; k++) {
      const unsigned long long bal = __ballot(s == k);
      if (lane == 0) cnt[i][wave][k] = __popcll(bal);
    }
  }
  __syncthreads();
  if (threadIdx.x < t.n_slots) {
    int tot = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++)
#pragma unroll
      for (int w = 0; w < SCAN_THREADS / 64; w++) tot += cnt[i][w][threadIdx.x];
    tile_cnt[(long long)threadIdx.x * n_tiles + blockIdx.x] = tot;
  }
}

__global__ __launch_bounds__(SCAN_THREADS) void k_stc_scatter(const float *__restrict__ points, long long n, int stride, int agent_col,
                                                              StcTable t, int n_tiles, const int *__restrict__ tile_off,
                                                              int *__restrict__ slot_start_out, float *__restrict__ out,
                                                              long long out_rows, int emit_cells, pcp_grid_t g,
                                                              int *__restrict__ cell_count, int *__restrict__ point_cell,
                                                              int *__restrict__ point_rank, const float *__restrict__ dev_m,
                                                              const unsigned char *__restrict__ dev_present) {
  __shared__ int cnt[SCAN_ITEMS][SCAN_THREADS / 64][STC_MAX_SLOTS];       // rows kept per (item, wave, slot) -> exclusive prefix
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long base = (long long)blockIdx.x * SCAN_TILE;
  const unsigned long long lt = (1ULL << lane) - 1ULL;
  int slot[SCAN_ITEMS], rank[SCAN_ITEMS];
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    const long long r = base + i * SCAN_THREADS + threadIdx.x;
    slot[i] = r < n ? stc_slot_of(points + r * stride, agent_col, t, dev_present) : -1;
    rank[i] = 0;
#pragma unroll
    for (int k = 0; k < STC_MAX_SLOTS; k++) {
      const unsigned long long bal = __ballot(slot[i] == k);
      if (lane == 0) cnt[i][wave][k] = __popcll(bal);
      if (slot[i] == k) rank[i] = __popcll(bal & lt);
    }
  }
  // first row of (slot, this tile) = rows of all earlier slots + rows of this slot in earlier tiles: every workgroup adds the per-tile
  // counts up itself (n_slots x n_tiles ints, a few KB from L2) -- no scan launch, no dependency between workgroups
  __shared__ int slot_total[STC_MAX_SLOTS], slot_before[STC_MAX_SLOTS], red_tot[SCAN_THREADS / 64][STC_MAX_SLOTS], red_bef[SCAN_THREADS / 64][STC_MAX_SLOTS];
  {
    int tot[STC_MAX_SLOTS], bef[STC_MAX_SLOTS];
#pragma unroll
    for (int k = 0; k < STC_MAX_SLOTS; k++) tot[k] = bef[k] = 0;
    for (int i = threadIdx.x; i < n_tiles; i += SCAN_THREADS) {
#pragma unroll
      for (int k = 0; k < STC_MAX_SLOTS; k++)
        if (k < t.n_slots) {
          const int c = tile_off[(long long)k * n_tiles + i];
          tot[k] += c;
          if (i < (int)blockIdx.x) bef[k] += c;
        }
    }
#pragma unroll
    for (int k = 0; k < STC_MAX_SLOTS; k++) {
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        tot[k] += __shfl_xor(tot[k], d, 64);
        bef[k] += __shfl_xor(bef[k], d, 64);
      }
      if (lane == 0) {
        red_tot[wave][k] = tot[k];
        red_bef[wave][k] = bef[k];
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < STC_MAX_SLOTS) {
    int a = 0, b = 0;
    for (int w = 0; w < SCAN_THREADS / 64; w++) {
      a += red_tot[w][threadIdx.x];
      b += red_bef[w][threadIdx.x];
    }
    slot_total[threadIdx.x] = a;
    slot_before[threadIdx.x] = b;
  }
  __syncthreads();
  int total = 0;
  {
    int run = 0;
#pragma unroll
    for (int k = 0; k < STC_MAX_SLOTS; k++) {
      if (k < t.n_slots) {
        if (blockIdx.x == 0 && threadIdx.x == 0 && slot_start_out) slot_start_out[k] = run;
        if ((int)threadIdx.x == k) slot_before[k] += run;     // the owner thread of slot k below
        run += slot_total[k];
      }
    }
    total = run;
    if (blockIdx.x == 0 && threadIdx.x == 0 && slot_start_out) slot_start_out[t.n_slots] = total;
  }
  __syncthreads();
  if (threadIdx.x < t.n_slots) {                       // rows of one slot keep the input order: items, then waves, then lanes
    int run = slot_before[threadIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++)
#pragma unroll
      for (int w = 0; w < SCAN_THREADS / 64; w++) {
        const int c = cnt[i][w][threadIdx.x];
        cnt[i][w][threadIdx.x] = run;
        run += c;
      }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    const long long r = base + i * SCAN_THREADS + threadIdx.x;
    if (slot[i] >= 0) {
      const float *row = points + r * stride;
      const int s = slot[i];
      const long long pos = (long long)cnt[i][wave][s] + rank[i];
      if (pos < out_rows) {                             // the host sized `out` from the same column: always true
        float *o = out + pos * stride;
        const int b = (int)row[0];
        const float *T = dev_m ? dev_m + 12 * (s * t.batch + b) : t.m[s * t.batch + b];
        const float x = row[1], y = row[2], z = row[3];
        const float fb = row[0] + (float)(s * t.batch);
        // bev_maker.py:179 `p @ R^T + t` on the reference's CPU path (torch -> BLAS sgemm, K = 3) is, bit for bit, the FMA chain
        // fma(z, r2, fma(y, r1, x * r0)) followed by a separately rounded + t (pinned: tests/golden/g2_disco_full.npz holds the SHA-256
        // of the transformed rows).  Spelled with explicit roundings so no contraction setting can reorder it.
        const float tx = __fadd_rn(__fmaf_rn(z, T[2], __fmaf_rn(y, T[1], __fmul_rn(x, T[0]))), T[3]);
        const float ty = __fadd_rn(__fmaf_rn(z, T[6], __fmaf_rn(y, T[5], __fmul_rn(x, T[4]))), T[7]);
        const float tz = __fadd_rn(__fmaf_rn(z, T[10], __fmaf_rn(y, T[9], __fmul_rn(x, T[8]))), T[11]);
        o[0] = fb;
        o[1] = tx;
        o[2] = ty;
        o[3] = tz;
        for (int c = 4; c < stride; c++) o[c] = row[c];
        if (emit_cells) {
          const float cell_row[3] = {fb, tx, ty};
          const int c = point_to_cell(cell_row, g);
          point_cell[pos] = c;
          if (c >= 0) point_rank[pos] = atomicAdd(&cell_count[c], 1);
        }
      }
    }
    if (r >= total && r < out_rows) {                   // tail of the destination: rows the pillariser must drop
      float *o = out + r * stride;
      o[0] = -1.0f;
      for (int c = 1; c < stride; c++) o[c] = 0.0f;
      if (emit_cells) point_cell[r] = -1;
    }
  }
}

// per-id row counts of an id column (values truncated like .long()): LDS histogram per workgroup, one global add per id and workgroup
__global__ __launch_bounds__(256) void k_column_id_counts(const float *__restrict__ points, long long n, int stride, int col,
                                                          unsigned long long *__restrict__ out) {
  __shared__ unsigned int s_cnt[64];
  __shared__ unsigned int s_bad;
  if (threadIdx.x < 64) s_cnt[threadIdx.x] = 0u;
  if (threadIdx.x == 0) s_bad = 0u;
  __syncthreads();
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float v = points[i * stride + col];
    if (v > -1.f && v < 64.f) atomicAdd(&s_cnt[(int)v], 1u); else atomicAdd(&s_bad, 1u);
  }
  __syncthreads();
  if (threadIdx.x < 64 && s_cnt[threadIdx.x]) {
    atomicAdd(&out[2 + threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
    atomicOr(&out[0], 1ULL << threadIdx.x);
  }
  if (threadIdx.x == 0 && s_bad) atomicAdd(&out[1], (unsigned long long)s_bad);
}

struct StcLayout { size_t tile_cnt, slot_start, total; };
inline StcLayout stc_layout(int64_t n, int n_slots) {
  StcLayout L;
  const int64_t n_tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  L.tile_cnt = 0;
  L.slot_start = pcp_align_up((size_t)(n_tiles > 0 ? n_tiles : 1) * n_slots * 4, 256);
  L.total = L.slot_start + 256;
  return L;
}

}  // namespace

extern "C" int pcp_column_id_counts(const float *points, int64_t n, int32_t row_stride, int32_t col, uint64_t *out66, void *stream_) {
  if (!out66 || n < 0 || row_stride <= 0 || col < 0 || col >= row_stride || (((uintptr_t)out66) & 15)) return PCP_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  if (pcp_zero_async(out66, 66 * sizeof(uint64_t), stream) != PCP_OK) return PCP_ERR_LAUNCH;
  if (n == 0) return PCP_OK;
  if (!points) return PCP_ERR_ARG;
  long long blocks = (n + 256 * 16 - 1) / (256 * 16);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_column_id_counts, dim3((unsigned)blocks), dim3(256), 0, stream, points, (long long)n, (int)row_stride, (int)col,
                     (unsigned long long *)out66);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

// ---- which (agent, frame) maps of a BEV maker exist, decided on the DEVICE (round 5: DiscoNet under hipGraph) -------------------------------------
// Reference behaviour (bev_maker.py:153-190): an agent without a single row in the batch is skipped -- it is not in batch_dict['bev_img'] and
// takes no part in the fusion's softmax; which FRAMES of an agent exist is metadata (se3_from_ego), known on the host.  The eager path reads
// the per-agent row counts back to decide this (the reference syncs at the same place); a captured graph cannot, so the makers run for every
// agent the metadata lists and the maps of agents without rows are zeroed (pcp_zero_maps_unless) and leave the softmax
// (pcp_disco_weight_fuse_live) from these flags.
//   live[a * batch + b] = 1 iff agent a holds a row anywhere in the batch (the same value for every frame b of the agent).
namespace {
__global__ __launch_bounds__(256) void k_agent_frame_hist(const float *__restrict__ points, long long n, int stride, int col, int batch,
                                                          int *__restrict__ hist) {
  extern __shared__ int s_hist[];                       // [64][batch]
  for (int i = threadIdx.x; i < 64 * batch; i += 256) s_hist[i] = 0;
  __syncthreads();
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float a = points[i * stride + col], fb = points[i * stride];
    if (a > -1.f && a < 64.f && fb >= 0.f && fb < (float)batch) atomicAdd(&s_hist[(int)a * batch + (int)fb], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * batch; i += 256)
    if (s_hist[i]) atomicAdd(&hist[i], s_hist[i]);
}
__global__ void k_agent_frame_live(int *__restrict__ hist_live, int batch) {
  const int a = threadIdx.x;                            // 64 threads: one agent each, flags written over its own histogram row
  int total = 0;
  for (int b = 0; b < batch; b++) total += hist_live[a * batch + b];
  for (int b = 0; b < batch; b++) hist_live[a * batch + b] = total > 0 ? 1 : 0;
}
constexpr int ZM_MAX = 256;
struct ZmTable { int idx[ZM_MAX]; };
__global__ __launch_bounds__(256) void k_zero_maps_unless(float *__restrict__ maps, long long map_elems, ZmTable t, const int *__restrict__ live) {
  const int m = blockIdx.y;
  const int k = t.idx[m];
  if (k < 0 || live[k] != 0) return;
  f32x4 *dst = reinterpret_cast<f32x4 *>(maps + (long long)m * map_elems);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < map_elems / 4; i += (long long)gridDim.x * 256) dst[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}
}  // namespace

extern "C" int pcp_agent_frame_live(const float *points, int64_t n, int32_t row_stride, int32_t agent_col, int32_t batch, int32_t *live,
                                    void *stream_) {
  if (!live || n < 0 || row_stride <= 0 || agent_col < 0 || agent_col >= row_stride || batch <= 0 || batch > 64 || (((uintptr_t)live) & 15))
    return PCP_ERR_ARG;
  if (n > 0 && !points) return PCP_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  if (pcp_zero_async(live, (size_t)64 * batch * 4, stream) != PCP_OK) return PCP_ERR_LAUNCH;
  if (n > 0) {
    long long blocks = (n + 256 * 16 - 1) / (256 * 16);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_agent_frame_hist, dim3((unsigned)blocks), dim3(256), (size_t)64 * batch * 4, stream, points, (long long)n, (int)row_stride,
                       (int)agent_col, (int)batch, live);
    PCP_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(k_agent_frame_live, dim3(1), dim3(64), 0, stream, live, (int)batch);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

extern "C" int pcp_zero_maps_unless(float *maps, int64_t map_elems, int32_t n_maps, const int32_t *flag_index_host, const int32_t *live,
                                    void *stream_) {
  if (!maps || !flag_index_host || !live || map_elems <= 0 || map_elems % 4 != 0 || n_maps <= 0 || n_maps > ZM_MAX || (((uintptr_t)maps) & 15))
    return PCP_ERR_ARG;
  ZmTable t;
  for (int i = 0; i < ZM_MAX; i++) t.idx[i] = i < n_maps ? flag_index_host[i] : -1;
  long long bx = (map_elems / 4 + 256 * 8 - 1) / (256 * 8);
  if (bx > 512) bx = 512;
  hipLaunchKernelGGL(k_zero_maps_unless, dim3((unsigned)bx, (unsigned)n_maps), dim3(256), 0, (hipStream_t)stream_, maps, (long long)map_elems, t,
                     live);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

extern "C" size_t pcp_select_transform_compact_workspace_bytes(int64_t n, int32_t n_slots) {
  if (n < 0 || n_slots <= 0 || n_slots > STC_MAX_SLOTS) return 0;
  return stc_layout(n, n_slots).total;
}

static int stc_impl(const float *points, int64_t n, int32_t row_stride, int32_t agent_col, int32_t n_slots, const float *agents_host, int32_t batch,
                    const float *poses_host, const uint8_t *present_host, const float *poses_dev, const uint8_t *present_dev, float *out,
                    int64_t out_rows, void *workspace, size_t workspace_bytes, int32_t *slot_start, const pcp_grid_t *vox_grid, void *vox_workspace,
                    size_t vox_workspace_bytes, void *stream_) {
  if (n < 0 || row_stride < 4 || agent_col < 0 || agent_col >= row_stride || n_slots <= 0 || n_slots > STC_MAX_SLOTS || batch <= 0 ||
      n_slots * batch > STC_MAX_POSES || out_rows < 0 || n >= (1LL << 31) || out_rows >= (1LL << 31))
    return PCP_ERR_ARG;
  const bool dev_tables = poses_dev != nullptr;
  if (dev_tables ? !present_dev : (!poses_host || !present_host)) return PCP_ERR_ARG;
  if (!agents_host || !workspace || (out_rows > 0 && !out) || (n > 0 && !points) || points == out)
    return PCP_ERR_ARG;
  const StcLayout L = stc_layout(n, n_slots);
  if (workspace_bytes < L.total) return PCP_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  StcTable t;
  t.n_slots = n_slots;
  t.batch = batch;
  for (int k = 0; k < STC_MAX_SLOTS; k++) t.agent[k] = k < n_slots ? (int)agents_host[k] : -2;
  for (int i = 0; i < STC_MAX_POSES; i++) {
    t.present[i] = (!dev_tables && i < n_slots * batch) ? present_host[i] : 0;
    for (int k = 0; k < 12; k++) t.m[i][k] = (!dev_tables && i < n_slots * batch) ? poses_host[i * 12 + k] : 0.f;
  }
  pcp_grid_t g;
  int *cell_count = nullptr, *point_cell = nullptr, *point_rank = nullptr;
  if (vox_grid) {
    if (!vox_workspace || vox_grid->nx <= 0 || vox_grid->ny <= 0 || vox_grid->batch_size <= 0) return PCP_ERR_ARG;
    const int64_t cells = (int64_t)vox_grid->batch_size * vox_grid->nx * vox_grid->ny;
    if (cells >= (1LL << 31)) return PCP_ERR_UNSUPPORTED;
    const VoxLayout V = pcp_vox_layout(cells, out_rows > 0 ? out_rows : 1);
    if (vox_workspace_bytes < V.total) return PCP_ERR_WORKSPACE;
    g = *vox_grid;
    cell_count = (int *)((char *)vox_workspace + V.cell_count);
    point_cell = (int *)((char *)vox_workspace + V.point_cell);
    point_rank = (int *)((char *)vox_workspace + V.point_rank);
    if (pcp_zero_async(cell_count, (size_t)cells * 4, stream) != PCP_OK) return PCP_ERR_LAUNCH;
  } else {
    g = pcp_grid_t{};
  }
  char *ws = (char *)workspace;
  int *tile_cnt = (int *)(ws + L.tile_cnt);
  // the grid covers the source rows and the destination (tail fill)
  const long long span = n > out_rows ? n : out_rows;
  const int n_tiles = (int)((n + SCAN_TILE - 1) / SCAN_TILE);
  const int n_blocks = (int)((span + SCAN_TILE - 1) / SCAN_TILE);
  if (n_tiles > 0) {
    hipLaunchKernelGGL(k_stc_count, dim3(n_tiles), dim3(SCAN_THREADS), 0, stream, points, (long long)n, (int)row_stride, (int)agent_col, t,
                       n_tiles, tile_cnt, (const unsigned char *)present_dev);
    PCP_CHECK_LAUNCH();
  }
  if (n_blocks > 0) {
    hipLaunchKernelGGL(k_stc_scatter, dim3(n_blocks), dim3(SCAN_THREADS), 0, stream, points, (long long)n, (int)row_stride, (int)agent_col, t,
                       n_tiles, tile_cnt, slot_start, out, (long long)out_rows, vox_grid ? 1 : 0, g, cell_count,
                       point_cell, point_rank, poses_dev, (const unsigned char *)present_dev);
    PCP_CHECK_LAUNCH();
  } else if (slot_start && pcp_zero_async(slot_start, (size_t)(n_slots + 1) * 4, stream) != PCP_OK) {
    return PCP_ERR_LAUNCH;
  }
  return PCP_OK;
}

extern "C" int pcp_select_transform_compact(const float *points, int64_t n, int32_t row_stride, int32_t agent_col, int32_t n_slots,
                                            const float *agents_host, int32_t batch, const float *poses_host,
                                            const uint8_t *present_host, float *out, int64_t out_rows, void *workspace,
                                            size_t workspace_bytes, int32_t *slot_start, const pcp_grid_t *vox_grid,
                                            void *vox_workspace, size_t vox_workspace_bytes, void *stream_) {
  if (!poses_host || !present_host) return PCP_ERR_ARG;
  return stc_impl(points, n, row_stride, agent_col, n_slots, agents_host, batch, poses_host, present_host, nullptr, nullptr, out, out_rows, workspace,
                  workspace_bytes, slot_start, vox_grid, vox_workspace, vox_workspace_bytes, stream_);
}

extern "C" int pcp_select_transform_compact_dev(const float *points, int64_t n, int32_t row_stride, int32_t agent_col, int32_t n_slots,
                                                const float *agents_host, int32_t batch, const float *poses_dev,
                                                const uint8_t *present_dev, float *out, int64_t out_rows, void *workspace,
                                                size_t workspace_bytes, int32_t *slot_start, const pcp_grid_t *vox_grid,
                                                void *vox_workspace, size_t vox_workspace_bytes, void *stream_) {
  if (!poses_dev || !present_dev) return PCP_ERR_ARG;
  return stc_impl(points, n, row_stride, agent_col, n_slots, agents_host, batch, nullptr, nullptr, poses_dev, present_dev, out, out_rows, workspace,
                  workspace_bytes, slot_start, vox_grid, vox_workspace, vox_workspace_bytes, stream_);
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

// ---- reproducible bucket order (training) ---------------------------------------------------------------------------------------------------
// A point's slot inside its pillar comes from the histogram atomic of pass 1, so the ORDER of a pillar's points in the bucket order changes
// from run to run.  Inference does not see it (per-pillar max / fixed-point mean are order independent); the training path's per-point
// GEMMs sum over the rows in bucket order, so their last bits would.  One thread per pillar sorts its run by point index (runs are a few
// points long; insertion sort): the bucket order becomes a function of the input alone.  Runs of more than PCP_LONG_PILLAR points (a LiDAR-like
// cloud has cells with hundreds to thousands: one thread's insertion sort of 5 000 points in global memory took a SECOND) are ranked by the
// whole workgroup instead: an element's place is the number of smaller ones (the keys are distinct), counted against LDS tiles of the run,
// four elements per thread and pass; the sorted run is staged in the (by then idle) point_rank array and copied back.
namespace {
constexpr int SORT_TILE = 2048, SORT_EPT = 4;

// the workgroup ranks one long run (see above)
__device__ __forceinline__ void sort_long_run(int s0, int cnt, int *__restrict__ bucket_order, int *__restrict__ scratch, int *tile) {
  const int tid = threadIdx.x;
  for (int base = 0; base < cnt; base += 256 * SORT_EPT) {
    int v[SORT_EPT], rank[SORT_EPT];
#pragma unroll
    for (int u = 0; u < SORT_EPT; ++u) {
      const int e = base + u * 256 + tid;
      v[u] = e < cnt ? bucket_order[s0 + e] : 0x7fffffff;
      rank[u] = 0;
    }
    for (int tb = 0; tb < cnt; tb += SORT_TILE) {
      const int tn = min(SORT_TILE, cnt - tb);
      __syncthreads();
      for (int j = tid; j < tn; j += 256) tile[j] = bucket_order[s0 + tb + j];
      __syncthreads();
      for (int j = 0; j < tn; ++j) {
        const int w = tile[j];                                    // broadcast read
#pragma unroll
        for (int u = 0; u < SORT_EPT; ++u) rank[u] += w < v[u] ? 1 : 0;
      }
    }
#pragma unroll
    for (int u = 0; u < SORT_EPT; ++u)
      if (base + u * 256 + tid < cnt) scratch[s0 + rank[u]] = v[u];
  }
  __syncthreads();                                                 // the whole sorted run is in scratch; nobody reads the unsorted one any more
  for (int j = tid; j < cnt; j += 256) bucket_order[s0 + j] = scratch[s0 + j];
  __syncthreads();
}

// short runs: one thread each (the long ones are on the pillariser's list: k_sort_long_runs)
__global__ __launch_bounds__(256) void k_sort_runs(const int *__restrict__ pillar_start, const int *__restrict__ counters,
                                                  int *__restrict__ bucket_order) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= counters[0]) return;
  const int s0 = pillar_start[p], s1 = pillar_start[p + 1];
  if (s1 - s0 > PCP_LONG_PILLAR) return;
  for (int i = s0 + 1; i < s1; ++i) {
    const int v = bucket_order[i];
    int j = i - 1;
    while (j >= s0 && bucket_order[j] > v) {
      bucket_order[j + 1] = bucket_order[j];
      --j;
    }
    bucket_order[j + 1] = v;
  }
}

// one workgroup per listed run, grid-stride: the long runs of a LiDAR-like cloud sit next to each other in pillar order -- left to the
// workgroup that found them, a few workgroups sorted hundreds of runs each (4.9 ms), spread over the chip they take tens of microseconds
__global__ __launch_bounds__(256) void k_sort_long_runs(const int *__restrict__ pillar_start, int *__restrict__ bucket_order,
                                                       int *__restrict__ scratch, const int *__restrict__ long_list,
                                                       const int *__restrict__ counters) {
  __shared__ int tile[SORT_TILE];
  const int nl = counters[6];
  for (int li = blockIdx.x; li < nl; li += gridDim.x) {
    const int q = long_list[li];
    sort_long_run(pillar_start[q], pillar_start[q + 1] - pillar_start[q], bucket_order, scratch, tile);
  }
}
}  // namespace

extern "C" int pcp_voxelize_sort_pillar_rows(const pcp_grid_t *grid, void *workspace, int64_t n, void *stream_) {
  if (!grid || !workspace || n < 0) return PCP_ERR_ARG;
  if (n == 0) return PCP_OK;
  const int64_t cells = (int64_t)grid->batch_size * grid->nx * grid->ny;
  const VoxLayout L = pcp_vox_layout(cells, n);
  char *ws = (char *)workspace;
  const int64_t max_pillars = n < cells ? n : cells;
  hipStream_t st = (hipStream_t)stream_;
  hipLaunchKernelGGL(k_sort_runs, dim3((unsigned)((max_pillars + 255) / 256)), dim3(256), 0, st, (const int *)(ws + L.pillar_start),
                     (const int *)(ws + L.counters), (int *)(ws + L.bucket_order));
  PCP_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_sort_long_runs, dim3(1024), dim3(256), 0, st, (const int *)(ws + L.pillar_start), (int *)(ws + L.bucket_order),
                     (int *)(ws + L.point_rank), (const int *)(ws + L.long_list), (const int *)(ws + L.counters));
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

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

// ---- the reference's index tensors, on demand, from a pillar list that was built WITHOUT them ---------------------------------------------------
// The pipeline mode (bench.py, tools/test.py --fast) calls pcp_pillarise_rows with voxel_coords = unq_inv = NULL: nothing downstream reads them.
// pcp_pillar_index_export rebuilds them from what that call left in the workspace -- the very tables pcp_pfn_rows / pcp_sparse_conv3x3_s2 consume
// (point_cell, cell_rank, pillar_cell, the records) -- so a caller (a late `batch_dict['voxel_coords']` reader, the parity tests) can have
// dynamic_pillar_vfe.py:104-108,137-147's tensors for exactly the pillar list the maps were computed from.
namespace {

__global__ __launch_bounds__(256) void k_index_export(long long n, pcp_grid_t g, const int *__restrict__ point_cell, const int *__restrict__ cell_rank,
                                                      const int *__restrict__ pillar_cell, const int *__restrict__ counters_ws,
                                                      const float *__restrict__ srows, int rs, int *__restrict__ voxel_coords,
                                                      int *__restrict__ row_rank, int *__restrict__ slot_rank, int *__restrict__ slot_canvas_row,
                                                      int *__restrict__ counters_out) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r == 0 && counters_out) {
#pragma unroll
    for (int i = 0; i < PCP_VOX_COUNTERS; i++) counters_out[i] = counters_ws[i];
  }
  if (r >= n) return;
  if (row_rank) {
    const int c = point_cell[r];
    row_rank[r] = c >= 0 ? cell_rank[c] : -1;
  }
  if (voxel_coords && r < counters_ws[0]) {
    const int c = pillar_cell[r];
    const int plane = g.nx * g.ny;
    const int b = c / plane, rem = c - b * plane;
    const int cx = rem / g.ny, cy = rem - cx * g.ny;
    *reinterpret_cast<int4 *>(voxel_coords + 4 * r) = make_int4(b, 0, cy, cx);
  }
  if (srows && r < counters_ws[1]) {
    if (slot_rank) slot_rank[r] = __float_as_int(srows[r * rs + rs - 3]) & 0x7fffffff;        // the sign bit tags the records of crowded pillars
    if (slot_canvas_row) slot_canvas_row[r] = __float_as_int(srows[r * rs + rs - 1]);
  }
}

}  // namespace

extern "C" int pcp_pillar_index_export(const pcp_grid_t *grid, const void *workspace, int64_t n, int32_t num_raw, int32_t *voxel_coords,
                                       int32_t *row_rank, int32_t *slot_rank, int32_t *slot_canvas_row, int32_t *counters, void *stream_) {
  if (!grid || !workspace || n < 0 || grid->nx <= 0 || grid->ny <= 0 || grid->batch_size <= 0) return PCP_ERR_ARG;
  if (num_raw != 0 && (num_raw < 3 || num_raw > 13)) return PCP_ERR_ARG;
  if ((slot_rank || slot_canvas_row) && num_raw == 0) return PCP_ERR_ARG;                        // a pcp_voxelize workspace holds no records
  if (((uintptr_t)voxel_coords) & 15) return PCP_ERR_ARG;
  const int64_t cells = (int64_t)grid->batch_size * grid->nx * grid->ny;
  if (cells >= (1LL << 31) || n >= (1LL << 31)) return PCP_ERR_UNSUPPORTED;
  const int64_t n_alloc = n > 0 ? n : 1;
  const VoxLayout L = pcp_vox_layout(cells, n_alloc);
  const char *ws = (const char *)workspace;
  const float *srows = nullptr;
  int rs = 0;
  if (num_raw) {
    const RowsLayout R = pcp_rows_layout(cells, n_alloc, num_raw);
    srows = (const float *)(ws + R.srows);
    rs = pcp_rows_stride(num_raw);
  }
  hipLaunchKernelGGL(k_index_export, dim3((unsigned)((n_alloc + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, (long long)n, *grid,
                     (const int *)(ws + L.point_cell), (const int *)(ws + L.cell_rank), (const int *)(ws + L.pillar_cell),
                     (const int *)(ws + L.counters), srows, rs, voxel_coords, row_rank, slot_rank, slot_canvas_row, counters);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}
