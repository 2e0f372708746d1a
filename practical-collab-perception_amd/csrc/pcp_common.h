// Shared device/host helpers for libpcp_hip.so (gfx950 only: wavefront = 64, no portability layers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/pcp_hip.h"

#define PCP_WAVE 64

#define PCP_CHECK_LAUNCH()                                   \
  do {                                                       \
    if (hipGetLastError() != hipSuccess) return PCP_ERR_LAUNCH; \
  } while (0)

static inline size_t pcp_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- workspace layout of pcp_voxelize / pcp_pfn_scatter (offsets in bytes, all 256-B aligned) -----------------------
struct VoxLayout {
  size_t cell_count;    // int32 [cells]   points per cell (zeroed every call)
  size_t cell_fill;     // int32 [cells]   bucket cursor   (zeroed every call; contiguous with cell_count)
  size_t cell_rank;     // int32 [cells]   pillar rank, -1 if empty
  size_t cell_start;    // int32 [cells]   first slot of the cell in bucket order
  size_t point_cell;    // int32 [n]       merged cell id or -1
  size_t bucket_order;  // int32 [n]       point rows grouped by pillar (ascending merged id)
  size_t pillar_cell;   // int32 [n]       merged id of pillar r
  size_t pillar_start;  // int32 [n + 1]   first slot of pillar r (pillar_start[P] = N')
  size_t block_sums;    // int32 [3 * nblk_max + 8]
  size_t counters;      // int32 [PCP_VOX_COUNTERS] copy of the public counters (device side)
  size_t total;
};

static inline VoxLayout pcp_vox_layout(int64_t cells, int64_t n) {
  VoxLayout L;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = pcp_align_up(off + bytes, 256); return o; };
  L.cell_count = take((size_t)cells * 4);
  L.cell_fill = take((size_t)cells * 4);
  L.cell_rank = take((size_t)cells * 4);
  L.cell_start = take((size_t)cells * 4);
  L.point_cell = take((size_t)n * 4);
  L.bucket_order = take((size_t)n * 4);
  L.pillar_cell = take((size_t)n * 4);
  L.pillar_start = take((size_t)(n + 1) * 4);
  int64_t nblk = (cells + 1023) / 1024 + (n + 1023) / 1024 + 2;
  L.block_sums = take((size_t)(3 * nblk + 8) * 4);
  L.counters = take(64);
  L.total = off;
  return L;
}
