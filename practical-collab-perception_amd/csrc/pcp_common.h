// Shared device/host helpers for libpcp_hip.so (gfx950 only: wavefront = 64, no portability layers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/pcp_hip.h"
#include "../../include/pcp_hip_train.h"

#define PCP_WAVE 64

#define PCP_CHECK_LAUNCH()                                   \
  do {                                                       \
    if (hipGetLastError() != hipSuccess) return PCP_ERR_LAUNCH; \
  } while (0)

static inline size_t pcp_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// abi.hip: the override of option `option` (PCP_OPT_*, include/pcp_hip.h) or `builtin` while none is set; the CU count of the current device
__attribute__((visibility("hidden"))) long long pcp_option(int option, long long builtin);
__attribute__((visibility("hidden"))) int pcp_current_device_cus();

// Zero fill as a KERNEL node (hipMemsetAsync nodes replayed incorrectly inside captured hipGraphs on this stack: the second
// replay of a graph containing them faulted, profiles/scripts/debug/dbg_graph.py): 16-byte stores, grid-stride.
__global__ static void pcp_k_zero(uint4 *__restrict__ p, size_t n16, unsigned char *__restrict__ tail, size_t ntail) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n16; i += stride) p[i] = make_uint4(0u, 0u, 0u, 0u);
  if (blockIdx.x == 0 && threadIdx.x < ntail) tail[threadIdx.x] = 0;
}

static inline int pcp_zero_async(void *ptr, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return PCP_OK;
  if (((uintptr_t)ptr) & 15) return hipMemsetAsync(ptr, 0, bytes, stream) == hipSuccess ? PCP_OK : PCP_ERR_LAUNCH;
  const size_t n16 = bytes / 16, ntail = bytes % 16;
  size_t blocks = (n16 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(pcp_k_zero, dim3((unsigned)blocks), dim3(256), 0, stream, (uint4 *)ptr, n16, (unsigned char *)ptr + n16 * 16, ntail);
  return hipGetLastError() == hipSuccess ? PCP_OK : PCP_ERR_LAUNCH;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- workspace layout of pcp_voxelize / pcp_pfn_scatter (offsets in bytes, all 256-B aligned) -----------------------
struct VoxLayout {
  size_t cell_count;    // int32 [cells]   points per cell (zeroed every call)
  size_t cell_rank;     // int32 [cells]   pillar rank, -1 if empty
  size_t cell_rs;       // int2  [cells]   occupied cells: {pillar rank, first slot of the cell in bucket order} (one 8-byte gather per point)
  size_t point_cell;    // int32 [n]       merged cell id or -1
  size_t point_rank;    // int32 [n]       arrival rank of the row inside its cell (the value the histogram atomic returned)
  size_t bucket_order;  // int32 [n]       point rows grouped by pillar (ascending merged id)
  size_t pillar_cell;   // int32 [n]       merged id of pillar r
  size_t pillar_start;  // int32 [n + 1]   first slot of pillar r (pillar_start[P] = N')
  size_t block_sums;    // int32 [3 * nblk_max + 8]
  size_t counters;      // int32 [16]      [0..3] the public counters (device side), [4] crowded pillars listed (rows mode), [6] long pillars listed
  size_t long_list;     // int32 [n / PCP_LONG_PILLAR + 2]  ranks of the pillars of more than PCP_LONG_PILLAR points, in no particular order: the
                        //                 training kernels give each of them a workgroup (one lane group walking 5 000 points took milliseconds)
  size_t total;
};
constexpr int PCP_LONG_PILLAR = 16;

static inline VoxLayout pcp_vox_layout(int64_t cells, int64_t n) {
  VoxLayout L;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = pcp_align_up(off + bytes, 256); return o; };
  L.cell_count = take((size_t)cells * 4);
  L.cell_rank = take((size_t)cells * 4);
  L.cell_rs = take((size_t)cells * 8);
  L.point_cell = take((size_t)n * 4);
  L.point_rank = take((size_t)n * 4);
  L.bucket_order = take((size_t)n * 4);
  L.pillar_cell = take((size_t)n * 4);
  L.pillar_start = take((size_t)(n + 1) * 4);
  int64_t nblk = (cells + 1023) / 1024 + (n + 1023) / 1024 + 2;
  L.block_sums = take((size_t)(3 * nblk + 8) * 4);
  L.counters = take(64);
  L.long_list = take((size_t)(n / PCP_LONG_PILLAR + 2) * 4);
  L.total = off;
  return L;
}

// ---- round 5: rows in pillar order + wave tiles for pcp_pfn_rows (csrc/pfn_rows.hip) ---------------------------------------------------
// The first VoxLayout fields keep their offsets, so a rows workspace is also a pcp_voxelize workspace (cell -> rank table for the sparse
// first layer, bucket order for the training kernels).
//   srows      float [n][rs]   the kept rows in SLOT order (slot = position in the bucket order: pillars ascending, a pillar's points
//                              consecutive): [raw 0 .. num_raw) | zero pad | pillar rank | cx << 16 | cy | (b * ny + cy) * nx + cx], rs = 8 (num_raw <= 5) or 16
//   tile_desc  int2  [n/T + 2] wave tile t owns the pillars whose first slot lies in [T t, T (t + 1)): {first such pillar, its first slot};
//                              written for every t with 0 < T t <= N' (entry 0 is {0, 0})
//   crowd_list int4  [n/64 + 2] pillars of at least `crowd` records (default PCP_PFN_CROWD; counters[4] of them, in no particular order):
//                              {first slot, records, pillar rank, canvas row}.  Their records carry the rank with the sign bit set: the wave
//                              tiles of pcp_pfn_rows pass over them and a workgroup per pillar (pfn_crowd_run, the front workgroups of k_pfn_rows) runs them instead -- one wave
//                              would otherwise walk thousands of records alone (LiDAR-like clouds: the cells next to the sensor)
constexpr int PCP_PFN_TILE = 30;
constexpr int PCP_PFN_CROWD = 192;              // default threshold; PCP_PFN_CROWD in the environment overrides (>= 64; 0 = never)
constexpr int PCP_PFN_CROWD_MIN = 64;
static inline int pcp_rows_stride(int num_raw) { return num_raw <= 5 ? 8 : 16; }
struct RowsLayout {
  VoxLayout v;
  size_t srows, tile_desc, crowd_list, total;
};
static inline RowsLayout pcp_rows_layout(int64_t cells, int64_t n, int num_raw) {
  RowsLayout R;
  R.v = pcp_vox_layout(cells, n);
  size_t off = R.v.total;
  auto take = [&](size_t bytes) { size_t o = off; off = pcp_align_up(off + bytes, 256); return o; };
  R.srows = take((size_t)n * pcp_rows_stride(num_raw) * 4);
  R.tile_desc = take((size_t)(n / PCP_PFN_TILE + 2) * 8);
  R.crowd_list = take((size_t)(n / PCP_PFN_CROWD_MIN + 2) * 16);
  R.total = off;
  return R;
}
