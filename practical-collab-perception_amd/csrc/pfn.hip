// a2 / a3 / a5 -- point features + PillarFeatureNet (2 layers) + scatter to the dense NHWC canvas, fused in one kernel.
//
// Reference path replaced (pcdet/models/backbones_3d/vfe/dynamic_pillar_vfe.py:110-126, :35-46 and
// pcdet/models/backbones_2d/map_to_bev/pointpillar_scatter.py:14-37):
//   mean = scatter_mean(xyz); f = [raw, xyz - mean, xyz - cell_centre]; x = relu(bn(W0 f)); xm = scatter_max(x);
//   y = relu(bn(W1 [x, xm[inv]])); out = scatter_max(y); canvas[:, y, x] = out.
// Five (N', .) temporaries and three torch_scatter launches in the reference; here a workgroup owns PILLARS_PER_BLOCK
// consecutive pillars (their points are one contiguous run of the bucket order built by pcp_voxelize), keeps the
// per-pillar maxima in LDS and writes each finished pillar as one 256-byte row of the NHWC canvas.
//
// Determinism: the per-pillar mean is accumulated in 2^-24 fixed point (integer adds commute), the maxima are
// order-independent, so the result does not depend on the bucket order.
//
// HBM bytes per frame (algorithmic): n * row_stride * 4 (points, read once through the bucket gather; the second and
// third sweeps hit L2) + P * 256 (canvas rows) [+ P * 256 pillar_features when requested].
#include "pcp_common.h"

namespace {

constexpr int PFN_THREADS = 256;
constexpr int PILLARS_PER_BLOCK = 64;
constexpr int C0 = 32;   // first PFN layer width  (NUM_FILTERS[0] / 2)
constexpr int C1 = 64;   // second PFN layer width (NUM_FILTERS[1])

struct PfnParams {
  const float *points;
  long long n;
  int stride;
  int num_raw;
  pcp_grid_t g;
  const int *bucket_order;
  const int *pillar_cell;
  const int *pillar_start;
  const int *counters;
  const float *w0, *b0, *w1, *b1;
  float *pillar_features;
  float *canvas;
};

__device__ __forceinline__ int find_pillar(const int *pl_start, int np, int slot) {
  // largest p with pl_start[p] <= slot  (pl_start is ascending, pl_start[np] = end)
  int lo = 0, hi = np - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (pl_start[mid] <= slot) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__device__ __forceinline__ unsigned fkey(float v) {        // order-preserving float -> uint (handles negatives)
  unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// One workgroup = PILLARS_PER_BLOCK consecutive pillars = one contiguous run of the bucket order, processed in chunks of 64 points.
//   sweep 1: fixed-point xyz sums -> per-pillar mean
//   sweep 2, per chunk: lanes = points; wave w computes layer-0 channels [8w, 8w+8) (its 88 weights stay in SGPRs), publishes them to
//            xmax0 (running per-pillar max) and to the chunk's x tile in LDS; after one barrier the POINT half of layer 1,
//            d = x . W1[:, :32]^T  ([64 points x 32] x [32 x 64]), runs on the matrix pipe: wave = (32-point tile, 32-channel tile),
//            16 v_mfma_f32_32x32x2_f32 with the W1 fragments resident in VGPRs; the accumulators go to the running max dmax[pillar][o]
//   epilogue: out = relu(b1 + xmax0 . W1[:, 32:]^T + dmax): the same MFMA tiling over [64 pillars x 32] x [32 x 64], one 128-byte
//            segment of a canvas row per half-wave store.
//   Exact w.r.t. the reference's order of operations up to the summation order inside the two 32-deep dot products:
//   max_p fl(c + d_p) = fl(c + max_p d_p) (rounding is monotone).
constexpr int XLD = 36;           // padded row of the x / xmax0 tiles (floats): conflict-free ds_read_b128 groups

__device__ __forceinline__ f32x16 mfma32p(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// The kernel is latency bound (a 64-pillar workgroup holds ~120 points = two chunks behind a chain of dependent global loads): ONE x tile
// and an extra barrier per chunk keep the LDS at 39 KB, i.e. four workgroups per CU instead of three -- 267 -> 233 us on 1.44 M points.
#ifdef PFN_DOUBLE_BUF
constexpr int PFN_NBUF = 2, PFN_WGS = 3;
#else
constexpr int PFN_NBUF = 1, PFN_WGS = 4;
#endif

#ifdef PFN_STAMP
__device__ unsigned long long pfn_dbg[32];
#define PFN_STAMP_AT(k) do { if (blockIdx.x == PFN_STAMP && threadIdx.x == 0) pfn_dbg[k] = __builtin_readcyclecounter(); } while (0)
#else
#define PFN_STAMP_AT(k) do {} while (0)
#endif

template <int NUM_RAW>
__global__ __launch_bounds__(PFN_THREADS, PFN_WGS) void k_pfn(PfnParams p) {
  constexpr int F = NUM_RAW + 6;
  constexpr int PL_CAP = 512;                            // points of the block with a direct point -> pillar table
  __shared__ unsigned char pl_of[PL_CAP];
  __shared__ int pl_start[PILLARS_PER_BLOCK + 1];
  __shared__ long long sum_fx[PILLARS_PER_BLOCK][3];
  __shared__ float mean[PILLARS_PER_BLOCK][3];
  __shared__ long long row_off[PILLARS_PER_BLOCK];      // canvas row (in floats) of each pillar
  __shared__ float cell_xy[PILLARS_PER_BLOCK][2];       // cell indices of each pillar as floats
  __shared__ __attribute__((aligned(16))) float xmax0[PILLARS_PER_BLOCK * XLD];   // >= 0 after ReLU: int order == float order
  __shared__ __attribute__((aligned(16))) unsigned dmax[PILLARS_PER_BLOCK][C1];      // fkey-encoded running max of the point half of layer 1; the epilogue's output staging
  __shared__ __attribute__((aligned(16))) float xs[PFN_NBUF][64 * XLD];           // layer-0 output of the current chunk
  __shared__ int pl_s[PFN_NBUF][64];                           // pillar of each point of the chunk (-1: past the end)

  PFN_STAMP_AT(0);
  const int r0 = blockIdx.x * PILLARS_PER_BLOCK;
  const int tid = threadIdx.x, lane = tid & 63;
  // The workgroup's time is a chain of dependent global loads (counters -> pillar_start -> bucket_order -> point rows): everything that does
  // not depend on the pillar count is requested BEFORE the count is waited for (the tables have n + 1 / n entries: guarded, values past the
  // last pillar are never used), and the slot range comes from two scalar loads instead of an LDS round trip behind a barrier.
  int ps_reg = 0, pc_reg = 0;
  if (tid <= PILLARS_PER_BLOCK && (long long)r0 + tid <= p.n) ps_reg = p.pillar_start[r0 + tid];
  if (tid < PILLARS_PER_BLOCK && (long long)r0 + tid < p.n) pc_reg = p.pillar_cell[r0 + tid];
  const int s0 = (long long)r0 <= p.n ? p.pillar_start[r0] : 0;
  const int P = p.counters[0];
  if (r0 >= P) return;
  const int np = min(PILLARS_PER_BLOCK, P - r0);
  const int s1 = p.pillar_start[r0 + np];
  PFN_STAMP_AT(1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // SGPR: keeps the weight addresses scalar
  const int r = lane & 31, h = lane >> 5;
  const int rt = wave >> 1, ct = wave & 1;
  const int plane = p.g.nx * p.g.ny;

  // rows of the first chunk and the bucket order of the second: in flight during the LDS set-up and sweep 1
  float rawc[NUM_RAW], rawn[NUM_RAW];
  int bo_n = 0;
  {
    const int sa = s0 + lane, sb = s0 + 64 + lane;
    const int bo = sa < s1 ? p.bucket_order[sa] : 0;
    if (sb < s1) bo_n = p.bucket_order[sb];
    const float *row = p.points + (long long)bo * p.stride;
#pragma unroll
    for (int k = 0; k < NUM_RAW; k++) rawc[k] = row[1 + k];             // slot past the end: row 0, never used
  }
#ifndef PFN_OLD_SWEEP1
  // xyz of slot s0 + tid for the mean sweep: requested now, with everything else that does not depend on the LDS set-up (the sweep used to
  // start its own bucket_order -> row chain behind two barriers: one more dependent global round trip on the workgroup's critical path)
  float xyz1[3] = {0.f, 0.f, 0.f};
  {
    const int s = s0 + tid;
    if (s < s1) {
      const float *row = p.points + (long long)p.bucket_order[s] * p.stride;
      xyz1[0] = row[1];
      xyz1[1] = row[2];
      xyz1[2] = row[3];
    }
  }
#endif
  if (tid <= np) pl_start[tid] = ps_reg;
  for (int i = tid; i < PILLARS_PER_BLOCK * 3; i += PFN_THREADS) (&sum_fx[0][0])[i] = 0;
  for (int i = tid; i < PILLARS_PER_BLOCK * XLD; i += PFN_THREADS) xmax0[i] = 0.f;
  for (int i = tid; i < PILLARS_PER_BLOCK * C1; i += PFN_THREADS) (&dmax[0][0])[i] = 0u;
  for (int i = tid; i < PILLARS_PER_BLOCK; i += PFN_THREADS) {
    long long off = 0;
    float cxf = 0.f, cyf = 0.f;
    if (i < np) {
      const int cell = pc_reg;                             // i == tid: PILLARS_PER_BLOCK <= PFN_THREADS
      const int b = cell / plane, rem = cell % plane;
      const int cx = rem / p.g.ny, cy = rem % p.g.ny;
      off = (((long long)b * p.g.ny + cy) * p.g.nx + cx) * C1;
      cxf = (float)cx;
      cyf = (float)cy;
    }
    row_off[i] = off;
    cell_xy[i][0] = cxf;
    cell_xy[i][1] = cyf;
  }
  // W1 fragments of this wave's 32 output channels, k permuted as the ds_read_b128 of the A tiles delivers it:
  // MFMA (j, i) multiplies k = 8j + 4h + i.  wp: point half (k < 32), wm: max half (k >= 32)
  f32x4 wp[4], wm[4];
  {
    const float *wr = p.w1 + (ct * 32 + r) * (2 * C0) + 4 * h;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      wp[j] = *reinterpret_cast<const f32x4 *>(wr + 8 * j);
      wm[j] = *reinterpret_cast<const f32x4 *>(wr + C0 + 8 * j);
    }
  }
  PFN_STAMP_AT(12);
  __syncthreads();
  PFN_STAMP_AT(13);
  // point -> pillar of the block's first PL_CAP points (a pillar is a run of consecutive slots); later points (crowded pillars) search
  for (int i = tid; i < np; i += PFN_THREADS) {
    const int a = pl_start[i] - s0, b = min(pl_start[i + 1] - s0, PL_CAP);
    for (int q = a; q < b; q++) pl_of[q] = (unsigned char)i;
  }
  auto pillar_of = [&](int s) { return (s - s0 < PL_CAP) ? (int)pl_of[s - s0] : find_pillar(pl_start, np, s); };
  __syncthreads();

  PFN_STAMP_AT(2);
  // ---- sweep 1: per-pillar xyz sums in 2^-24 fixed point (integer adds commute -> deterministic) ---------------------------
#ifdef PFN_OLD_SWEEP1
  for (int s = s0 + tid; s < s1; s += PFN_THREADS) {
    const float *row = p.points + (long long)p.bucket_order[s] * p.stride;
    int pl = pillar_of(s);
#pragma unroll
    for (int a = 0; a < 3; a++) {
      long long q = __double2ll_rn((double)row[1 + a] * 16777216.0);
      atomicAdd(reinterpret_cast<unsigned long long *>(&sum_fx[pl][a]), (unsigned long long)q);
    }
  }
#else
  if (s0 + tid < s1) {
    const int pl = pillar_of(s0 + tid);
#pragma unroll
    for (int a = 0; a < 3; a++) {
      long long q = __double2ll_rn((double)xyz1[a] * 16777216.0);
      atomicAdd(reinterpret_cast<unsigned long long *>(&sum_fx[pl][a]), (unsigned long long)q);
    }
  }
  for (int s = s0 + PFN_THREADS + tid; s < s1; s += PFN_THREADS) {          // crowded blocks: more than 256 points in 64 pillars
    const float *row = p.points + (long long)p.bucket_order[s] * p.stride;
    int pl = pillar_of(s);
#pragma unroll
    for (int a = 0; a < 3; a++) {
      long long q = __double2ll_rn((double)row[1 + a] * 16777216.0);
      atomicAdd(reinterpret_cast<unsigned long long *>(&sum_fx[pl][a]), (unsigned long long)q);
    }
  }
#endif
  __syncthreads();
  for (int i = tid; i < np * 3; i += PFN_THREADS) {
    int pl = i / 3, a = i % 3;
    int cnt = pl_start[pl + 1] - pl_start[pl];
    mean[pl][a] = (float)(((double)sum_fx[pl][a] * (1.0 / 16777216.0)) / (double)cnt);
  }
  __syncthreads();

  PFN_STAMP_AT(3);
  // cell-centre offsets exactly as the reference constructor rounds them (dynamic_pillar_vfe.py:80-82)
  const float x_off = __fadd_rn(p.g.voxel_x * 0.5f, p.g.min_x);
  const float y_off = __fadd_rn(p.g.voxel_y * 0.5f, p.g.min_y);
  const float z_off = __fadd_rn(p.g.voxel_z * 0.5f, p.g.min_z);

  // ---- sweep 2 -----------------------------------------------------------------------------------------------------------
  int buf = 0;
  for (int base = s0; base < s1; base += 64, buf ^= (PFN_NBUF - 1)) {
    const int s = base + lane;
    // software pipeline: the next chunk's rows (their bucket order arrived during the previous iteration) and the bucket order of the
    // chunk after it are requested now and consumed one iteration later
    int bo_nn = 0;
    {
      const int sn = s + 64, sn2 = s + 128;
      if (sn < s1) {
        const float *rown = p.points + (long long)bo_n * p.stride;
#pragma unroll
        for (int k = 0; k < NUM_RAW; k++) rawn[k] = rown[1 + k];
      }
      if (sn2 < s1) bo_nn = p.bucket_order[sn2];
    }
    if (base == s0) PFN_STAMP_AT(4);
    int pl = -1;
    float x8[8];
#pragma unroll
    for (int c = 0; c < 8; c++) x8[c] = 0.f;
    if (s < s1) {
      pl = pillar_of(s);
      float f[F];
#pragma unroll
      for (int k = 0; k < NUM_RAW; k++) f[k] = rawc[k];
      f[NUM_RAW + 0] = __fsub_rn(f[0], mean[pl][0]);
      f[NUM_RAW + 1] = __fsub_rn(f[1], mean[pl][1]);
      f[NUM_RAW + 2] = __fsub_rn(f[2], mean[pl][2]);
      f[NUM_RAW + 3] = __fsub_rn(f[0], __fadd_rn(__fmul_rn(cell_xy[pl][0], p.g.voxel_x), x_off));
      f[NUM_RAW + 4] = __fsub_rn(f[1], __fadd_rn(__fmul_rn(cell_xy[pl][1], p.g.voxel_y), y_off));
      f[NUM_RAW + 5] = __fsub_rn(f[2], z_off);
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const int ch = wave * 8 + c;
        float acc = p.b0[ch];                          // wave-uniform addresses -> scalar loads
#pragma unroll
        for (int k = 0; k < F; k++) acc = fmaf(p.w0[ch * F + k], f[k], acc);
        x8[c] = fmaxf(acc, 0.0f);
        atomicMax(reinterpret_cast<int *>(&xmax0[pl * XLD + ch]), __float_as_int(x8[c]));
      }
    }
    *reinterpret_cast<f32x4 *>(&xs[buf][lane * XLD + wave * 8]) = f32x4{x8[0], x8[1], x8[2], x8[3]};
    *reinterpret_cast<f32x4 *>(&xs[buf][lane * XLD + wave * 8 + 4]) = f32x4{x8[4], x8[5], x8[6], x8[7]};
    if (wave == 0) pl_s[buf][lane] = pl;
    if (base == s0) PFN_STAMP_AT(5);
    __syncthreads();
    if (base == s0) PFN_STAMP_AT(6);
    // point half of layer 1 on the matrix pipe: rows = points rt*32 .. +31, columns = channels ct*32 .. +31
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.f;
#ifndef PFN_OLD_EPILOGUE
    // the pillar of each of this lane's 16 accumulator rows: requested before the MFMAs, so the running-max updates behind them issue back
    // to back instead of one LDS round trip per row (stamps: 3 k cycles per chunk for 16 atomics)
    int ppl_e[16];
#pragma unroll
    for (int e = 0; e < 16; e++) ppl_e[e] = pl_s[buf][rt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h];
#endif
    const float *xa = &xs[buf][(rt * 32 + r) * XLD + 4 * h];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const f32x4 a = *reinterpret_cast<const f32x4 *>(xa + 8 * j);
      acc = mfma32p(a.x, wp[j].x, acc);
      acc = mfma32p(a.y, wp[j].y, acc);
      acc = mfma32p(a.z, wp[j].z, acc);
      acc = mfma32p(a.w, wp[j].w, acc);
    }
    if (base == s0) PFN_STAMP_AT(7);
    const int o = ct * 32 + r;
#ifdef PFN_OLD_EPILOGUE
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const int prow = rt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      const int ppl = pl_s[buf][prow];
      if (ppl >= 0) atomicMax(&dmax[ppl][o], fkey(acc[e]));
    }
#else
    {
      // rows of one pillar are consecutive in this lane's row order: fold a run into ONE update (1.9 points per pillar: ~half the atomics)
      unsigned run = 0u;
      int run_pl = -1;
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int ppl = ppl_e[e];
        const unsigned k = fkey(acc[e]);
        if (ppl != run_pl) {
          if (run_pl >= 0) atomicMax(&dmax[run_pl][o], run);
          run_pl = ppl;
          run = k;
        } else {
          run = max(run, k);
        }
      }
      if (run_pl >= 0) atomicMax(&dmax[run_pl][o], run);
    }
#endif
#pragma unroll
    for (int k = 0; k < NUM_RAW; k++) rawc[k] = rawn[k];
    bo_n = bo_nn;
    if (base == s0) PFN_STAMP_AT(8);
    if (PFN_NBUF == 1) __syncthreads();
    if (base == s0) PFN_STAMP_AT(9);      // single buffer (four workgroups per CU): every wave is done with xs before the next chunk
    // xs / pl_s are double buffered: the next chunk writes the other buffer, and the barrier of the chunk after that orders the reuse
  }
  __syncthreads();

  PFN_STAMP_AT(10);
  // ---- epilogue: out = relu(b1 + xmax0 . W1[:, 32:]^T + dmax) for rows = pillars rt2*32 .. +31 --------------------------------
  const int o = ct * 32 + r;
  const float bias = p.b1[o];
#ifdef PFN_OLD_EPILOGUE
  for (int rt2 = rt; rt2 * 32 < np; rt2 += 2) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.f;
    const float *xa = &xmax0[(rt2 * 32 + r) * XLD + 4 * h];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const f32x4 a = *reinterpret_cast<const f32x4 *>(xa + 8 * j);
      acc = mfma32p(a.x, wm[j].x, acc);
      acc = mfma32p(a.y, wm[j].y, acc);
      acc = mfma32p(a.z, wm[j].z, acc);
      acc = mfma32p(a.w, wm[j].w, acc);
    }
    PFN_STAMP_AT(14);
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const int pl = rt2 * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (pl < np) {
        const float v = fmaxf((bias + acc[e]) + fkey_inv(dmax[pl][o]), 0.0f);
        if (p.pillar_features) p.pillar_features[(long long)(r0 + pl) * C1 + o] = v;
        if (p.canvas) p.canvas[row_off[pl] + o] = v;
      }
    }
  }
#else
  // the finished value replaces its own dmax word (one owner per (pillar, channel)); after a barrier every pillar row leaves as 16-byte
  // stores: a wave instruction writes four whole 256-byte canvas rows (the first version stored 128-byte half rows a dword per lane, with
  // an LDS round trip in front of each: 11 - 15 k cycles of a 50 k workgroup)
  for (int rt2 = rt; rt2 * 32 < np; rt2 += 2) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.f;
    const float *xa = &xmax0[(rt2 * 32 + r) * XLD + 4 * h];
    unsigned dm[16];
#pragma unroll
    for (int e = 0; e < 16; e++) dm[e] = dmax[rt2 * 32 + (e & 3) + 8 * (e >> 2) + 4 * h][o];      // all 16 reads in flight under the MFMAs
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const f32x4 a = *reinterpret_cast<const f32x4 *>(xa + 8 * j);
      acc = mfma32p(a.x, wm[j].x, acc);
      acc = mfma32p(a.y, wm[j].y, acc);
      acc = mfma32p(a.z, wm[j].z, acc);
      acc = mfma32p(a.w, wm[j].w, acc);
    }
    PFN_STAMP_AT(14);
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const int pl = rt2 * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      dmax[pl][o] = __float_as_uint(fmaxf((bias + acc[e]) + fkey_inv(dm[e]), 0.0f));
    }
  }
  __syncthreads();
  for (int i = tid; i < np * (C1 / 4); i += PFN_THREADS) {
    const int pl = i >> 4, q = i & 15;
    const f32x4 v = *reinterpret_cast<const f32x4 *>(&dmax[pl][4 * q]);
    if (p.pillar_features) *reinterpret_cast<f32x4 *>(p.pillar_features + (long long)(r0 + pl) * C1 + 4 * q) = v;
    if (p.canvas) *reinterpret_cast<f32x4 *>(p.canvas + row_off[pl] + 4 * q) = v;
  }
#endif
  PFN_STAMP_AT(11);
}

__global__ void k_canvas_clear(const int *__restrict__ pillar_cell, const int *__restrict__ counters, pcp_grid_t g,
                               float *__restrict__ canvas) {
  const int P = counters[0];
  const int plane = g.nx * g.ny;
  // 16 lanes x float4 per pillar row (64 floats)
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = t; i < (long long)P * 16; i += stride) {
    int r = (int)(i >> 4), q = (int)(i & 15);
    int cell = pillar_cell[r];
    int b = cell / plane, rem = cell % plane;
    int cx = rem / g.ny, cy = rem % g.ny;
    float4 *dst = reinterpret_cast<float4 *>(canvas + (((long long)b * g.ny + cy) * g.nx + cx) * C1) + q;
    *dst = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

}  // namespace

extern "C" int pcp_pfn_scatter(const float *points, int64_t n, int32_t row_stride, int32_t num_raw, const pcp_grid_t *grid,
                               const void *workspace, const float *w0, const float *b0, const float *w1, const float *b1,
                               float *pillar_features, float *canvas, void *stream_) {
  if (!grid || !workspace || !w0 || !b0 || !w1 || !b1 || n < 0) return PCP_ERR_ARG;
  if (row_stride < 1 + num_raw || num_raw < 3) return PCP_ERR_ARG;
  if (n == 0) return PCP_OK;
  const int64_t cells = (int64_t)grid->batch_size * grid->nx * grid->ny;
  VoxLayout L = pcp_vox_layout(cells, n);
  const char *ws = (const char *)workspace;
  PfnParams p;
  p.points = points;
  p.n = n;
  p.stride = row_stride;
  p.num_raw = num_raw;
  p.g = *grid;
  p.bucket_order = (const int *)(ws + L.bucket_order);
  p.pillar_cell = (const int *)(ws + L.pillar_cell);
  p.pillar_start = (const int *)(ws + L.pillar_start);
  p.counters = (const int *)(ws + L.counters);
  p.w0 = w0; p.b0 = b0; p.w1 = w1; p.b1 = b1;
  p.pillar_features = pillar_features;
  p.canvas = canvas;
  int64_t max_pillars = n < cells ? n : cells;
  int blocks = (int)((max_pillars + PILLARS_PER_BLOCK - 1) / PILLARS_PER_BLOCK);
  hipStream_t stream = (hipStream_t)stream_;
  switch (num_raw) {
    case 5: hipLaunchKernelGGL(k_pfn<5>, dim3(blocks), dim3(PFN_THREADS), 0, stream, p); break;
    case 11: hipLaunchKernelGGL(k_pfn<11>, dim3(blocks), dim3(PFN_THREADS), 0, stream, p); break;
    case 3: hipLaunchKernelGGL(k_pfn<3>, dim3(blocks), dim3(PFN_THREADS), 0, stream, p); break;
    case 4: hipLaunchKernelGGL(k_pfn<4>, dim3(blocks), dim3(PFN_THREADS), 0, stream, p); break;
    default: return PCP_ERR_UNSUPPORTED;
  }
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

extern "C" int pcp_canvas_clear(const pcp_grid_t *grid, const void *workspace, int64_t n, float *canvas, void *stream_) {
  if (!grid || !workspace || !canvas || n < 0) return PCP_ERR_ARG;
  if (n == 0) return PCP_OK;
  const int64_t cells = (int64_t)grid->batch_size * grid->nx * grid->ny;
  VoxLayout L = pcp_vox_layout(cells, n);
  const char *ws = (const char *)workspace;
  hipLaunchKernelGGL(k_canvas_clear, dim3(1024), dim3(256), 0, (hipStream_t)stream_, (const int *)(ws + L.pillar_cell),
                     (const int *)(ws + L.counters), *grid, canvas);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

extern "C" int pcp_fill_zero(void *ptr, size_t bytes, void *stream_) {
  if (!ptr && bytes) return PCP_ERR_ARG;
  if (bytes == 0) return PCP_OK;
  return pcp_zero_async(ptr, bytes, (hipStream_t)stream_);
}

#ifdef PFN_STAMP
extern "C" int pcp_debug_read_pfn(void *dst, size_t bytes) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(pfn_dbg), bytes) == hipSuccess ? 0 : 3;
}
#endif
