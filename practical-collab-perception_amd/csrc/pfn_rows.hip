// a2 / a3 / a5, round 5 -- PillarFeatureNet + scatter rebuilt for the memory system: wave-autonomous, register-chained, streaming.
//
// Reference path replaced (pcdet/models/backbones_3d/vfe/dynamic_pillar_vfe.py:110-126, :35-46 and
// pcdet/models/backbones_2d/map_to_bev/pointpillar_scatter.py:14-37):
//   mean = scatter_mean(xyz); f = [raw, xyz - mean, xyz - cell_centre]; x = relu(bn(W0 f)); xm = scatter_max(x);
//   y = relu(bn(W1 [x, xm[inv]])); out = scatter_max(y); canvas[:, y, x] = out.
//
// What k_pfn (csrc/pfn.hip, rounds 1-4) measured: 0.33 matrix-pipe busy, 0.39 of the wave cycles parked -- a 64-pillar workgroup sat
// behind a chain of dependent gathers (counters -> pillar_start -> bucket_order -> rows), three barriers per 64-point chunk and an LDS
// round trip of the layer-0 tile between the two layers.  Here:
//   * pcp_pillarise_rows leaves the kept rows IN PILLAR ORDER (32- or 64-byte records that carry the pillar rank, cell and frame), so a
//     wave's input is one contiguous run: coalesced 16-byte loads, no index chain;
//   * ONE WAVE owns a tile: the pillars whose first slot lies in [30 t, 30 (t + 1)) -- about 30 points, at most 30 pillars, any number of
//     points per pillar -- found from an 8-byte descriptor the pillariser's scan wrote.  No workgroup barrier anywhere; every per-pillar
//     reduction (fixed-point xyz sums, running maxima of both layers) is an LDS atomic in the wave's private 13 KB;
//   * both layers run on v_mfma_f32_16x16x4_f32 with the WEIGHTS as the A operand and 16 points as the B operand: the accumulator of layer 0
//     (lane (n, g): channels 16 b + 4 g + i of point n) IS the B operand of layer 1 -- no LDS round trip, no shuffle; the layer-0 bias rides
//     on a spare feature slot; a lane ends with four consecutive channels of one pillar: 16-byte stores;
//   * dense canvas: every wave also zero-fills the EMPTY cells of an equal slice of the cell -> rank table (last in its work), so the canvas
//     is written exactly once -- no clear-by-list pass, no zero fill;
//   * pillars of >= 192 records (LiDAR-like clouds: the cells next to the sensor) are passed over by the wave tiles and run by pfn_crowd_run
//     below, a workgroup per pillar (bit-identical results).
// Order independence (bitwise reproducible results whatever the arrival order inside a pillar): means accumulate in 2^-24 fixed point
// (integer adds commute), maxima are order independent, every per-point product has a fixed summation order.
//
// Algorithmic HBM bytes per launch: n' * 32 (rows) + P * 256 (pillar rows) or B * ny * nx * 256 (canvas).  Matrix work: 2 * (12 * 32 + 32 * 64)
// flop per point + 2 * 32 * 64 per pillar -- at the fp32 MFMA rate (157 TFLOP/s) about as long as the bytes take at the copy rate: the kernel
// is balanced between the two, which is why neither may wait for the other.
#include <stdlib.h>
#include "pcp_common.h"

namespace {

constexpr int PR_THREADS = 256;                 // four independent waves; the workgroup exists only to share the launch
constexpr int PR_T = PCP_PFN_TILE;
constexpr int PR_MAXP = 32;                     // pillars per wave tile (<= PR_T: every owned pillar starts at a different slot of the window)

struct PrParams {
  const float *srows;                           // [N'][RS]
  const int2 *tile_desc;
  const int *counters;                          // P, N'
  const int *cell_rank;                         // canvas mode: occupancy of the cells between this wave's pillars
  const int4 *crowd_list;                       // pillars of many records {first slot, records, rank, canvas row}; counters[4] of them (pfn_crowd_run)
  const float *w0, *b0, *w1, *b1;
  float *pillar_features;                       // (P, 64) or null
  float *canvas;                                // (B, ny, nx, 64) or null
  pcp_grid_t g;
  long long cells;
  int n_tiles_max;                              // tiles the host sized the grid for (ceil(n / T) + 1)
  int crowd_blocks;                             // workgroups at the front of the grid that run the crowded pillars
  unsigned long long plane_m, ny_m;             // exact division of a cell id (< 2^31) by nx * ny and by ny: (c * m) >> sh
  int plane_sh, ny_sh;
};

// floor(c / d) for 0 <= c < 2^31 as one 64-bit multiply and a shift: m = floor(2^(31 + s) / d) + 1 with 2^s >= d (exact for 31-bit c)
inline void magic_div(unsigned d, unsigned long long *m, int *sh) {
  int s = 0;
  while ((1ULL << s) < d) s++;
  *m = ((1ULL << (31 + s)) / d) + 1ULL;
  *sh = 31 + s;
}
__device__ __forceinline__ int div_magic(int c, unsigned long long m, int sh) { return (int)(((unsigned long long)(unsigned)c * m) >> sh); }

typedef float f32x4a __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4a mfma16(float a, float b, f32x4a c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// feature index of layer-0 k-step s for lane group g (-1: no feature, -2: the constant 1 that carries the bias)
template <int NUM_RAW>
__device__ __forceinline__ int feat_index(int s, int g) {
  if (g < 3) {
    if (s == 0) return g;                       // x, y, z
    if (s == 1) return NUM_RAW + g;             // f_cluster
    if (s == 2) return NUM_RAW + 3 + g;         // f_center
  } else {
    if (s == 0) return NUM_RAW > 3 ? 3 : -2;
    if (s == 1) return NUM_RAW > 4 ? 4 : (NUM_RAW > 3 ? -2 : -1);
    if (s == 2) return NUM_RAW > 5 ? 5 : (NUM_RAW > 4 ? -2 : -1);
  }
  const int k = 6 + 4 * (s - 3) + g;
  if (k < NUM_RAW) return k;
  return (k == NUM_RAW && NUM_RAW > 5) ? -2 : -1;
}
// number of layer-0 k-steps: every feature and the bias slot must have a (step, group) home
// relu on the bits: a negative float is a negative integer (one v_max_i32; fmaxf costs a canonicalising v_max_f32 more)
__device__ __forceinline__ float relu_f(float v) { return __int_as_float(max(__float_as_int(v), 0)); }

template <int NUM_RAW>
constexpr int l0_steps() { return NUM_RAW <= 5 ? 3 : 3 + (NUM_RAW - 6 + 1 + 3) / 4; }


// round(a * 2^24) as a 64-bit integer without a trip through f64: a - floor(a) is exact in fp32 and adding the (even) integer part does not
// change a round-half-even decision, so hi * 2^24 + rint(frac * 2^24) == rint(a * 2^24)
__device__ __forceinline__ long long fixed24(float a) {
  const float fl = floorf(a);
  const float fr = __fsub_rn(a, fl);
  const long long hi = (long long)(int)fl;
  const unsigned lo = (unsigned)(int)rintf(fr * 16777216.0f);
  return hi * 16777216LL + (long long)lo;
}

#ifdef PR_STAMP
// diagnostic build (csrc/build_variant.sh prstamp "-DPR_STAMP=<workgroup>"): shader cycles wave 0 of that workgroup spends per phase, summed
// over its tiles; read by tools/stamp_pfn_rows.py.  No stamp executes in the product build.
__device__ unsigned long long pr_dbg[16];
__device__ unsigned long long pr_wave_cycles[4096 * 3];        // per wave of the grid: cycles in the tiles | in the singles | in the canvas fill
__device__ unsigned int pr_tile_cycles[65536 * 4];             // per wave tile (the first 65536): cycles, records, pillars, column tiles walked
#define PR_MARK(k)                                                \
  do {                                                            \
    if (stamp) {                                                  \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
      st_acc[k] += now_ - st_last;                                \
      st_last = now_;                                             \
    }                                                             \
  } while (0)
#else
#define PR_MARK(k) do {} while (0)
#endif

// ---- crowded pillars: one workgroup per pillar ------------------------------------------------------------------------------------------------
// A LiDAR-like cloud puts hundreds to thousands of points into the cells next to the sensor.  The wave that owns such a pillar in k_pfn_rows
// would walk all of its records alone (5 000 records = 312 column tiles ~ 0.2 ms on one wave while the chip is done).  The pillariser lists
// every pillar of at least `crowd` records (default PCP_PFN_CROWD = 192) and tags its records; the wave tiles pass over them; the FIRST
// workgroups of k_pfn_rows' own grid (they start first and run beside the tile workgroups: as a launch of its own behind the tiles the same
// work cost 94 us on the 6-agent ring cloud, next to 135 us for all the tiles) take them instead, the four waves of a workgroup sharing a
// pillar's column tiles: fixed-point sums -> mean (one barrier), then features -> layer 0 -> point half of
// layer 1 with the running maxima kept in REGISTERS across the tiles (every record is the same pillar: no LDS atomics per point), one
// cross-lane + cross-wave reduction at the end, the pillar half of layer 1 on wave 0.  Same operations in the same order per point, exact
// integer sums, order-free maxima: BIT-identical to what the owner wave of k_pfn_rows computes for the same pillar.
// worker = index of this workgroup among the n_workers crowd workgroups; PC_WAVES waves each; the shared words are the caller's
template <int NUM_RAW, int PC_WAVES>
__device__ __forceinline__ void pfn_crowd_run(const PrParams &p, int worker, int n_workers, unsigned long long *s_sum, float *s_mean, float *s_x,
                                              float *s_d) {
  constexpr int RS = NUM_RAW <= 5 ? 8 : 16;
  constexpr int S0 = l0_steps<NUM_RAW>();
  constexpr int F = NUM_RAW + 6;
  const int n_crowd = p.counters[4];
  if (worker >= n_crowd) return;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  float w0f[2][S0];
#pragma unroll
  for (int b = 0; b < 2; b++)
#pragma unroll
    for (int s = 0; s < S0; s++) {
      const int fi = feat_index<NUM_RAW>(s, g);
      w0f[b][s] = fi >= 0 ? p.w0[(16 * b + n) * F + fi] : (fi == -2 ? p.b0[16 * b + n] : 0.f);
    }
  f32x4 w1a[4][2], w1b[4][2];
#pragma unroll
  for (int r = 0; r < 4; r++)
#pragma unroll
    for (int b = 0; b < 2; b++) {
      const float *wr = p.w1 + (16 * r + n) * 64 + 16 * b + 4 * g;
      w1a[r][b] = *reinterpret_cast<const f32x4 *>(wr);
      w1b[r][b] = *reinterpret_cast<const f32x4 *>(wr + 32);
    }
  const float x_off = __fadd_rn(p.g.voxel_x * 0.5f, p.g.min_x);
  const float y_off = __fadd_rn(p.g.voxel_y * 0.5f, p.g.min_y);
  const float z_off = __fadd_rn(p.g.voxel_z * 0.5f, p.g.min_z);
  const float vsel = g == 0 ? p.g.voxel_x : (g == 1 ? p.g.voxel_y : 0.f);
  const float osel = g == 0 ? x_off : (g == 1 ? y_off : z_off);
  const int acol = g < 3 ? g : 3;

  for (int e = worker; e < n_crowd; e += n_workers) {
    const int4 ent = p.crowd_list[e];
    const int start = ent.x, cnt = ent.y;
    const int ncol = (cnt + 15) >> 4;
    if (tid < 4) s_sum[tid] = 0ULL;
    if (tid < 32) s_x[tid] = 0.f;
    if (tid < 64) s_d[tid] = -__builtin_inff();
    __syncthreads();
    // ---- fixed-point sums of x, y, z (the count is known) -> mean, exactly the arithmetic of k_pfn_rows' phases A and B ---------------------
    {
      long long part = 0;
      if (g < 3)
        for (int j0 = wave; j0 < ncol; j0 += 4 * PC_WAVES) {                 // four loads in flight per lane
          float v[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int slot = start + 16 * (j0 + u * PC_WAVES) + n;
            v[u] = slot < start + cnt ? p.srows[(long long)slot * RS + g] : 0.f;
          }
#pragma unroll
          for (int u = 0; u < 4; u++)
            if (start + 16 * (j0 + u * PC_WAVES) + n < start + cnt) part += fixed24(v[u]);
        }
      if (g < 3) __hip_atomic_fetch_add(&s_sum[g], (unsigned long long)part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    if (tid < 3) {
      const double c = (double)(unsigned)cnt, sd = (double)(long long)s_sum[tid] * (1.0 / 16777216.0);
      double rc = __builtin_amdgcn_rcp(c);
      rc = __builtin_fma(__builtin_fma(-c, rc, 1.0), rc, rc);
      double qd = sd * rc;
      qd = __builtin_fma(__builtin_fma(-c, qd, sd), rc, qd);
      s_mean[tid] = (float)qd;
    }
    __syncthreads();
    const float mean = s_mean[acol < 3 ? acol : 0];
    // ---- per 16 points: features -> layer 0 -> point half of layer 1; maxima in registers --------------------------------------------------
    float xm0[4] = {0.f, 0.f, 0.f, 0.f}, xm1[4] = {0.f, 0.f, 0.f, 0.f};
    float dm[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int i = 0; i < 4; i++) dm[r][i] = -__builtin_inff();
    // the record of the wave's next column tile is requested before the current one is worked on
    struct CRec { float av; f32x4 q; float e4, e5, e6g, e10; };
    auto crec_load = [&](CRec &r, int j) {
      const int slot = start + 16 * j + n;
      const float *src = p.srows + (long long)(slot < start + cnt ? slot : start + cnt - 1) * RS;
      r.av = src[acol];
      r.q = *reinterpret_cast<const f32x4 *>(src + RS - 4);
      r.e4 = r.e5 = r.e6g = r.e10 = 0.f;
      if (RS == 16) {
        r.e4 = src[4];
        r.e5 = src[5];
        r.e6g = src[6 + g];
        r.e10 = src[10];
      }
    };
    CRec rcur, rnext;
    if (wave < ncol) crec_load(rcur, wave);
    rnext = rcur;
    for (int j = wave; j < ncol; j += PC_WAVES) {
      if (j + PC_WAVES < ncol) crec_load(rnext, j + PC_WAVES);
      const bool valid = start + 16 * j + n < start + cnt;
      const float av = rcur.av;
      const f32x4 q = rcur.q;
      const float e4 = rcur.e4, e5 = rcur.e5, e6g = rcur.e6g, e10 = rcur.e10;
      const int cxcy = __float_as_int(q.z);
      const float cf = (float)(g == 0 ? (cxcy >> 16) : (cxcy & 0xffff));
      const float centre = __fadd_rn(__fmul_rn(cf, vsel), osel);
      const float cluster = __fsub_rn(av, mean);
      float f[S0];
      const float raw4 = RS == 8 ? q.x : e4;
      f[0] = (g < 3 || NUM_RAW > 3) ? av : 1.f;
      f[1] = g < 3 ? cluster : (NUM_RAW > 4 ? raw4 : (NUM_RAW > 3 ? 1.f : 0.f));
      f[2] = g < 3 ? __fsub_rn(av, centre) : (NUM_RAW > 5 ? e5 : (NUM_RAW > 4 ? 1.f : 0.f));
      if (S0 > 3) f[3] = e6g;
      if (S0 > 4) f[4] = g == 0 ? e10 : (g == 1 ? 1.f : 0.f);
      f32x4a x0 = f32x4a{0.f, 0.f, 0.f, 0.f}, x1 = f32x4a{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < S0; s++) {
        x0 = mfma16(w0f[0][s], f[s], x0);
        x1 = mfma16(w0f[1][s], f[s], x1);
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        x0[i] = relu_f(x0[i]);
        x1[i] = relu_f(x1[i]);
      }
      f32x4a dacc[4];
#pragma unroll
      for (int r = 0; r < 4; r++) dacc[r] = f32x4a{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) dacc[r] = mfma16(w1a[r][0][i], x0[i], dacc[r]);
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) dacc[r] = mfma16(w1a[r][1][i], x1[i], dacc[r]);
      if (valid) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
          xm0[i] = fmaxf(xm0[i], x0[i]);
          xm1[i] = fmaxf(xm1[i], x1[i]);
#pragma unroll
          for (int r = 0; r < 4; r++) dm[r][i] = fmaxf(dm[r][i], dacc[r][i]);
        }
      }
      rcur = rnext;
    }
    // lane (n, g) holds channels 4 g + i (x0), 16 + 4 g + i (x1), 16 r + 4 g + i (d) of ITS points: maxima over the 16 lanes n and the waves
#pragma unroll
    for (int i = 0; i < 4; i++) {
      __hip_atomic_fetch_max(&s_x[4 * g + i], xm0[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_max(&s_x[16 + 4 * g + i], xm1[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
      for (int r = 0; r < 4; r++) __hip_atomic_fetch_max(&s_d[16 * r + 4 * g + i], dm[r][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    // ---- the pillar half of layer 1 (one pillar: column 0 of the 16-column tile), bias, ReLU, stores ---------------------------------------
    if (wave == 0) {
      f32x4a o[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(p.b1 + 16 * r + 4 * g);
        o[r] = f32x4a{bv.x, bv.y, bv.z, bv.w};
      }
      float b0v[4], b1v[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        b0v[i] = n == 0 ? s_x[4 * g + i] : 0.f;
        b1v[i] = n == 0 ? s_x[16 + 4 * g + i] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) o[r] = mfma16(w1b[r][0][i], b0v[i], o[r]);
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) o[r] = mfma16(w1b[r][1][i], b1v[i], o[r]);
      if (n == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          f32x4 v;
          v.x = relu_f(o[r][0] + s_d[16 * r + 4 * g + 0]);
          v.y = relu_f(o[r][1] + s_d[16 * r + 4 * g + 1]);
          v.z = relu_f(o[r][2] + s_d[16 * r + 4 * g + 2]);
          v.w = relu_f(o[r][3] + s_d[16 * r + 4 * g + 3]);
          if (p.pillar_features) *reinterpret_cast<f32x4 *>(p.pillar_features + (long long)ent.z * 64 + 16 * r + 4 * g) = v;
          if (p.canvas) *reinterpret_cast<f32x4 *>(p.canvas + (long long)ent.w * 64 + 16 * r + 4 * g) = v;
        }
      }
    }
    __syncthreads();
  }
}


// WPS: waves per SIMD the register allocation is held to (3: 168 registers, 2: 256)
template <int NUM_RAW, int WPS>
__global__ __launch_bounds__(PR_THREADS, WPS) void k_pfn_rows(PrParams p) {
  constexpr int RS = NUM_RAW <= 5 ? 8 : 16;
  constexpr int RQ = RS / 4;                                  // 16-byte pieces of a record
  constexpr int S0 = l0_steps<NUM_RAW>();
  constexpr int F = NUM_RAW + 6;
  constexpr int NPRE = 2;                                     // 16-point column tiles of a wave tile that travel in registers (32 slots: nearly all)
  // wave-private LDS (no barrier ever orders it: the LDS executes one wave's instructions in order)
  __shared__ __attribute__((aligned(16))) unsigned long long s_sum[4][PR_MAXP * 4];       // [pillar][x, y, z, count] fixed point; then means
  __shared__ __attribute__((aligned(16))) float s_xmax[4][32 * 32];                       // [channel][pillar column] running max of layer 0 (>= 0)
  __shared__ __attribute__((aligned(16))) float s_dmax[4][64 * 32];                       // [channel][pillar column] running max of W1a . x
  __shared__ __attribute__((aligned(16))) float s_b1[64];                                 // every wave writes the same values: no barrier needed

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  // the first workgroups of the grid take the crowded pillars (see pfn_crowd_run), all others the wave tiles
  if ((int)blockIdx.x < p.crowd_blocks) {
    pfn_crowd_run<NUM_RAW, 4>(p, blockIdx.x, p.crowd_blocks, s_sum[0], s_xmax[0], s_xmax[1], s_dmax[0]);
    return;
  }
  const int bid = blockIdx.x - p.crowd_blocks, n_blocks = gridDim.x - p.crowd_blocks;     // this workgroup among those of the wave tiles
  unsigned long long *sum = s_sum[wave];
  float *meanf = reinterpret_cast<float *>(sum);              // [pillar][4]: mean x, y, z | canvas row (int bits); aliases the sums
  float *xmax = s_xmax[wave];
  float *dmax = s_dmax[wave];
  const int skew = 16 * (g & 1);                              // pillar column of lane group g: (lp + skew) & 31 -> conflict-free banks

  // ---- once per wave: weight fragments, LDS init ----------------------------------------------------------------------------------
  float w0f[2][S0];
#pragma unroll
  for (int b = 0; b < 2; b++)
#pragma unroll
    for (int s = 0; s < S0; s++) {
      const int fi = feat_index<NUM_RAW>(s, g);
      w0f[b][s] = fi >= 0 ? p.w0[(16 * b + n) * F + fi] : (fi == -2 ? p.b0[16 * b + n] : 0.f);
    }
  f32x4 w1a[4][2], w1b[4][2];                                 // [row block r][k block b]: W1[16 r + n][16 b + 4 g + i], + 32 for the max half
#pragma unroll
  for (int r = 0; r < 4; r++)
#pragma unroll
    for (int b = 0; b < 2; b++) {
      const float *wr = p.w1 + (16 * r + n) * 64 + 16 * b + 4 * g;
      w1a[r][b] = *reinterpret_cast<const f32x4 *>(wr);
      w1b[r][b] = *reinterpret_cast<const f32x4 *>(wr + 32);
    }
  for (int i = lane; i < PR_MAXP * 4; i += 64) sum[i] = 0ULL;
  for (int i = lane; i < 32 * 32; i += 64) xmax[i] = 0.f;
  for (int i = lane; i < 64 * 32; i += 64) dmax[i] = -__builtin_inff();
  s_b1[lane] = p.b1[lane];

  // counters of pcp_pillarise_rows: pillars, kept points, records of multi-point pillars (slots [0, Nm)), single-point pillars (slots [Nm, N'))
  const int Nv = p.counters[1], Nm = p.counters[2], S = p.counters[3];
  const int n_tiles = min(Nm / PR_T + 1, p.n_tiles_max);      // wave tiles over the multi-point records
  const int n_sing = (S + 63) >> 6;                           // units of 64 single-point pillars
  const int plane = p.g.nx * p.g.ny;
  // cell-centre offsets exactly as the reference constructor rounds them (dynamic_pillar_vfe.py:80-82)
  const float x_off = __fadd_rn(p.g.voxel_x * 0.5f, p.g.min_x);
  const float y_off = __fadd_rn(p.g.voxel_y * 0.5f, p.g.min_y);
  const float z_off = __fadd_rn(p.g.voxel_z * 0.5f, p.g.min_z);
  const float vsel = g == 0 ? p.g.voxel_x : (g == 1 ? p.g.voxel_y : 0.f);
  const float osel = g == 0 ? x_off : (g == 1 ? y_off : z_off);
  // wave-uniform reads (tile descriptors, the record that ends a tile's cell range) go through the scalar cache: their waits count on
  // lgkmcnt, so they never order behind this wave's vector stores (the tables were written by the kernel in front of this one)
  typedef __attribute__((address_space(4))) const int *cint_p;
  const cint_p desc_c = (cint_p)(unsigned long long)p.tile_desc;
  const cint_p rows_c = (cint_p)(unsigned long long)p.srows;

  // Work distribution: the wave tiles are dealt out in CHUNKS of PR_CHUNK consecutive tiles, chunk k of wave w = chunk number k * waves + w,
  // so every wave's tiles sample the whole cloud (the tiles of a LiDAR-like frame's middle hold the long pillars and cost 2 - 4 x the
  // others; as ONE contiguous run per wave -- the first form -- they all sat in a few waves' runs).  Inside a chunk the tiles are
  // consecutive and share their boundary slot.  No counters, no atomics (profiles/experiments/r05_pfn_tickets: what drawing chunks at run
  // time costs).  Measured (tools/bench_frontend.py, three alternating runs per build on one box): chunks of 1, 2, 4 tiles and the
  // contiguous runs are within 2 % of each other on both clouds -- per-tile stamps (tools/stamp_pfn_rows.py) show why: the slowest wave is
  // set by single tiles of 130 - 220 records (one wave walks 9 - 14 column tiles, 58 k cycles against 15 k for a two-column tile), which
  // no static deal evens out.  Chunks of 2: with 4 the stride of a wave's chunks (waves * 4 tiles) came within 7 % of 1.5 frames of the
  // 6-agent cloud, and two of a wave's three chunks fell on the same part of a frame.
  // The singles keep one contiguous run per wave (they all cost the same); the canvas fill is interleaved like the tiles.
#ifndef PR_CHUNK_N
#define PR_CHUNK_N 2
#endif
  constexpr int PR_CHUNK = PR_CHUNK_N, BIG = 0x7fffffff;
  const int n_waves = n_blocks * 4;
  const int gw = bid * 4 + wave;
  auto next_tile = [&](int q) -> int {                          // the tile this wave takes after tile q (BIG: none)
    const int nq = ((q + 1) % PR_CHUNK != 0) ? q + 1 : q + 1 + (n_waves - 1) * PR_CHUNK;
    return nq < n_tiles ? nq : BIG;
  };

  auto first_slot = [&](int t) -> int {                        // first record of wave tile t; tiles past the multi-point records: Nm
    if (t == 0) return 0;
    return ((long long)t * PR_T <= Nm) ? desc_c[2 * (long long)t + 1] : Nm;
  };
  // What a lane keeps of a record: lane (n, g) multiplies point n's features {x | y | z | raw 3}[g] and their derived ones, so it loads
  // exactly its own column (a per-lane address: no register-indexed select) and the record's last 16 bytes.
  struct Rec {
    float a;                 // raw[min(g, 3)]
    f32x4 q;                 // RS 8: {raw 4, rank, cx << 16 | cy, canvas row}; RS 16: {pad, rank, cell, canvas row}
    float e4, e5, e6g, e10;  // RS 16 only: raw 4, raw 5, raw 6 + g, raw 10
  };
  const int acol = g < 3 ? g : 3;
  auto load_rec = [&](Rec &r, int slot) {
    const float *src = p.srows + (long long)slot * RS;
    r.a = src[acol];
    r.q = *reinterpret_cast<const f32x4 *>(src + RS - 4);
    if (RS == 16) {
      r.e4 = src[4];
      r.e5 = src[5];
      r.e6g = src[6 + g];
      r.e10 = src[10];
    }
  };
  auto load_recs = [&](Rec (&dst)[NPRE], int a0, int b0s) {
#pragma unroll
    for (int j = 0; j < NPRE; j++) {
      const int slot = a0 + 16 * j + n;
      if (slot < b0s) load_rec(dst[j], slot);
    }
  };
  // the features of a point (f_cluster = 0 for the only point of a pillar) and layer 0 on them: x^T (32 channels x 16 points) = W0 . f^T,
  // bias on the spare feature slot, ReLU
  auto layer0 = [&](const Rec &rr, float cluster, f32x4a &x0, f32x4a &x1) {
    const int cxcy = __float_as_int(rr.q.z);
    const float av = rr.a;
    const float cf = (float)(g == 0 ? (cxcy >> 16) : (cxcy & 0xffff));
    const float centre = __fadd_rn(__fmul_rn(cf, vsel), osel);
    float f[S0];
    const float raw4 = RS == 8 ? rr.q.x : rr.e4;
    f[0] = (g < 3 || NUM_RAW > 3) ? av : 1.f;
    f[1] = g < 3 ? cluster : (NUM_RAW > 4 ? raw4 : (NUM_RAW > 3 ? 1.f : 0.f));
    f[2] = g < 3 ? __fsub_rn(av, centre) : (NUM_RAW > 5 ? rr.e5 : (NUM_RAW > 4 ? 1.f : 0.f));
    if (S0 > 3) f[3] = rr.e6g;                                               // raws 6 .. 9
    if (S0 > 4) f[4] = g == 0 ? rr.e10 : (g == 1 ? 1.f : 0.f);               // raw 10 | the constant 1 of the bias
    x0 = f32x4a{0.f, 0.f, 0.f, 0.f};
    x1 = f32x4a{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < S0; s++) {
      x0 = mfma16(w0f[0][s], f[s], x0);
      x1 = mfma16(w0f[1][s], f[s], x1);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      x0[i] = relu_f(x0[i]);
      x1[i] = relu_f(x1[i]);
    }
  };

#ifdef PR_STAMP
  const unsigned long long wave_t0 = __builtin_amdgcn_s_memtime();
  unsigned long long wave_t1 = wave_t0, wave_t2 = wave_t0;
  const bool stamp = bid == PR_STAMP && wave == 0;
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
  int st_cols = 0, st_pillars = 0;
#endif

  // =================================== wave tiles over the records of multi-point pillars ===================================================
  int q0 = gw * PR_CHUNK < n_tiles ? gw * PR_CHUNK : BIG;
  if (q0 != BIG) {
    // the records [a, b) of the tile in hand and of the two behind it in this wave's sequence
    auto span = [&](int q, int prev_q, int prev_b, int &a, int &b) {
      a = b = Nm;
      if (q != BIG) {
        a = (q == prev_q + 1) ? prev_b : first_slot(q);          // consecutive tiles share the boundary
        b = first_slot(q + 1);
      }
    };
    int q1 = next_tile(q0), q2 = q1 == BIG ? BIG : next_tile(q1);
    int a0, b0, a1, b1, a2, b2;
    span(q0, -2, 0, a0, b0);
    span(q1, q0, b0, a1, b1);
    span(q2, q1, b1, a2, b2);
    Rec bufa[NPRE], bufb[NPRE];
#pragma unroll
    for (int j = 0; j < NPRE; j++) {
      bufa[j].a = bufb[j].a = 0.f;
      bufa[j].q = bufb[j].q = f32x4{0.f, 0.f, 0.f, 0.f};
      bufa[j].e4 = bufa[j].e5 = bufa[j].e6g = bufa[j].e10 = bufb[j].e4 = bufb[j].e5 = bufb[j].e6g = bufb[j].e10 = 0.f;
    }
    load_recs(bufa, a0, b0);
#ifdef PR_STAMP
    int tiles_done = 0;
#endif

    // one tile; `cur` holds its first 32 records, `nxt` receives the next tile's
    auto tile_body = [&](Rec (&cur)[NPRE], Rec (&nxt)[NPRE]) {
      const int a = a0, bslot = b0;
#ifdef PR_STAMP
      const unsigned long long tile_t0 = __builtin_amdgcn_s_memtime();
      int tile_pillars = 0, tile_walked = 0;
#endif
      // ---- prefetch: the next tile's records (its span arrived a tile ago), the span of the tile three on
      load_recs(nxt, a1, b1);                                  // behind the last tile: a1 == b1 == Nm, nothing is loaded
      const int q3 = q2 == BIG ? BIG : next_tile(q2);
      int a3, b3;
      span(q3, q2, b2, a3, b3);
      PR_MARK(0);

      if (a < bslot) {
        const int ncol = (bslot - a + 15) >> 4;
        // local pillar index of every record: the number of pillar heads (rank differs from the record in front) up to it.  The ranks of
        // a tile are not consecutive (single-point pillars lie between them in rank order), the records are.
        int heads = 0, last_rank = -1;
        unsigned head_mask = 0;                                  // heads of the column tile local_pillar saw last
        auto local_pillar = [&](const Rec &r, bool valid) -> int {
          const int rk = __float_as_int(r.q.y);
          const int prev = __builtin_amdgcn_update_dpp(last_rank, rk, 0x111, 0xf, 0xf, false);      // row_shr:1, lane n = 0 keeps last_rank
          const unsigned long long hm = __ballot(valid && rk != prev) & 0xffffULL;                    // row g = 0 (all four rows agree)
          head_mask = (unsigned)hm;
          const int lp = heads + __builtin_popcountll(hm & ((2ULL << n) - 1ULL)) - 1;
          heads += __builtin_popcountll(hm);
          last_rank = __builtin_amdgcn_readlane(rk, 15);
          return valid ? lp : 0;
        };
        // a record whose rank carries the sign bit belongs to a crowded pillar (pcp_common.h: crowd_list): pfn_crowd_run runs those, a
        // workgroup per pillar; here they are passed over
        auto rec_ok = [&](const Rec &r, int slot) -> bool { return slot < bslot && __float_as_int(r.q.y) >= 0; };
        // column tile j (>= NPRE) lies inside ONE crowded pillar: the tile index in front of the one that holds the first record behind
        // that pillar (every wave tile the pillar covers names its end), else j
        auto crowd_jump = [&](int j, int rk, bool in) -> int {
          const int rk0 = __builtin_amdgcn_readfirstlane(rk);
          if (rk0 >= 0) return j;
          if (__ballot(in && rk != rk0)) return j;
          const int s0 = a + 16 * j;
          const int tgt = first_slot(s0 / PR_T + 1);
          if (tgt <= s0 + 16 || tgt > bslot) return j;
          if (rows_c[(long long)(tgt - 1) * RS + RS - 3] != rk0) return j;
          last_rank = rk0;
          return (tgt - a) / 16 - 1;
        };
        // The column tiles past a tile's first 32 slots (pillars longer than the window), one per trip of ONE loop (the code of a column tile
        // exists once: unrolled groups of four had grown the kernel to 95 KB, past the 64 KB instruction cache); the next column tile's
        // records are requested before the current one is worked on.  col(rec, j) handles one column tile and returns crowd_jump's answer.
        auto load_col = [&](Rec &r, int j) {
          const int slot = a + 16 * j + n;
          load_rec(r, slot < bslot ? slot : bslot - 1);
        };
        auto walk_cols = [&](auto &&col) {
          if (NPRE >= ncol) return;
          Rec rc, rn;
          int j = NPRE;
          load_col(rc, j);
          rn = rc;
          while (true) {
            if (j + 1 < ncol) load_col(rn, j + 1);
            const int jn = col(rc, j);
            const int next = jn + 1;
            if (next >= ncol) break;
            if (jn != j) load_col(rn, next);                     // a jump over a crowded pillar: the prefetched column tile is not the next one
            rc = rn;
            j = next;
          }
        };
        int lpj[NPRE];
        bool onej[NPRE];                                         // the column tile holds records of ONE pillar (no head behind lane 0)
#pragma unroll
        for (int j = 0; j < NPRE; j++) {
          lpj[j] = (j < ncol) ? local_pillar(cur[j], rec_ok(cur[j], a + 16 * j + n)) : 0;
          onej[j] = (head_mask & 0xfffeu) == 0;
        }

        // ---- phase A: fixed-point xyz sums and the point count of every pillar ---------------------------------------------------------
        auto phase_a = [&](const Rec &r, int lp) {
          const unsigned long long q = g < 3 ? (unsigned long long)fixed24(r.a) : 1ULL;
          __hip_atomic_fetch_add(&sum[lp * 4 + g], q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
#pragma unroll
        for (int j = 0; j < NPRE; j++)
          if (rec_ok(cur[j], a + 16 * j + n)) phase_a(cur[j], lpj[j]);
        {
          // crowded tiles: the records past the first 32 slots are read where they are used (twice: here and in phase C)
          const int heads0 = heads, last0 = last_rank;
          walk_cols([&](const Rec &r, int j) -> int {
            const bool in = a + 16 * j + n < bslot;
            const int rk = __float_as_int(r.q.y);
            const bool valid = in && rk >= 0;
            const int lp = local_pillar(r, valid);
            if (valid) phase_a(r, lp);
            return crowd_jump(j, rk, in);
          });
          if (ncol > NPRE) {                                   // phase C walks the same records again from the same state
            heads = heads0;
            last_rank = last0;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        PR_MARK(1);
        // ---- phase B: means (the sums' memory is reused: [pillar][4] floats); 16 * k < np is decided on the counts themselves --------------
        {
          long long sv[2];
          unsigned cntv[2];
#pragma unroll
          for (int k = 0; k < 2; k++) {
            const int lp = n + 16 * k;
            sv[k] = (long long)sum[lp * 4 + g];
            cntv[k] = (unsigned)sum[lp * 4 + 3];
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
          for (int k = 0; k < 2; k++) {
            const int lp = n + 16 * k;
            if (g < 3 && cntv[k] > 0) {
              // sum / count in f64 by reciprocal + one Newton step + one residual correction (the f64 divide is ~10x the instructions)
              const double c = (double)cntv[k], sd = (double)sv[k] * (1.0 / 16777216.0);
              double rc = __builtin_amdgcn_rcp(c);
              rc = __builtin_fma(__builtin_fma(-c, rc, 1.0), rc, rc);
              double qd = sd * rc;
              qd = __builtin_fma(__builtin_fma(-c, qd, sd), rc, qd);
              meanf[lp * 4 + g] = (float)qd;
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }

        PR_MARK(2);
        // ---- phase C: per 16 points: features -> layer 0 -> running max -> point half of layer 1 -> running max --------------------------
        int *pinfo = reinterpret_cast<int *>(sum) + PR_MAXP * 4;     // [pillar][4] ints in the upper half of the sums' memory: rank, canvas row
        // maxima over the 16 lanes of a row (the 16 points of a column tile) by four rotations, eight values at a time: v = max(ror(v), v) as
        // ONE v_max_f32_dpp per value and step (the compiler's form of the same is five instructions and a hazard nop), the eight values
        // interleaved so that a value's next step sits eight instructions behind its last write (DPP reads need two wait states)
#define PR_ROR8(N)                                                                                                     \
  asm volatile("v_max_f32_dpp %0, %0, %0 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\t"                               \
               "v_max_f32_dpp %1, %1, %1 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\t"                               \
               "v_max_f32_dpp %2, %2, %2 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\t"                               \
               "v_max_f32_dpp %3, %3, %3 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\t"                               \
               "v_max_f32_dpp %4, %4, %4 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\t"                               \
               "v_max_f32_dpp %5, %5, %5 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\t"                               \
               "v_max_f32_dpp %6, %6, %6 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\t"                               \
               "v_max_f32_dpp %7, %7, %7 row_ror:" #N " row_mask:0xf bank_mask:0xf"                                    \
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]))
        auto row_max8 = [&](float (&v)[8]) {
          asm volatile("s_nop 1" ::: "memory");
          PR_ROR8(8);
          PR_ROR8(4);
          PR_ROR8(2);
          PR_ROR8(1);
        };
        // `one` (wave uniform): every record of the column tile belongs to one pillar -- its 16 lanes would hit each LDS word of the running
        // maxima 16 at a time (serialised: the tiles of a LiDAR-like cloud's medium-sized pillars ran at a third of the rate); the maxima are
        // taken across the lanes first and lane 0 of each row alone goes to the LDS
        auto phase_c = [&](const Rec &rr, int lp, bool valid, bool one) {
          const int col = (lp + skew) & 31;
          float *xb = xmax + (4 * g) * 32 + col, *db = dmax + (4 * g) * 32 + col;     // + compile-time offsets per (block, i)
          if (g == 3 && valid) *reinterpret_cast<float2 *>(&pinfo[lp * 4]) = make_float2(rr.q.y, rr.q.w);   // every point of the pillar writes the same pair
          const float mean = meanf[lp * 4 + acol];
          f32x4a x0, x1;
          layer0(rr, __fsub_rn(rr.a, mean), x0, x1);
#ifdef PR_DIAG_NO_CMAX
          if (x0[0] == 1.2345e30f)                               // timing-only build: never true, keeps the arithmetic
#endif
          if (one) {
            float m[8];
#pragma unroll
            for (int i = 0; i < 4; i++) {
              m[i] = valid ? x0[i] : 0.f;
              m[4 + i] = valid ? x1[i] : 0.f;
            }
            row_max8(m);
            if (valid && n == 0) {
#pragma unroll
              for (int i = 0; i < 4; i++) {
                __hip_atomic_fetch_max(&xb[i * 32], m[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_max(&xb[(16 + i) * 32], m[4 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              }
            }
          } else if (valid) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
              __hip_atomic_fetch_max(&xb[i * 32], x0[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              __hip_atomic_fetch_max(&xb[(16 + i) * 32], x1[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          }
          // point half of layer 1: d^T (64 x 16) = W1[:, :32] . x^T -- the layer-0 accumulators are the B operand as they stand
          f32x4a dacc[4];
#pragma unroll
          for (int r = 0; r < 4; r++) dacc[r] = f32x4a{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int i = 0; i < 4; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) dacc[r] = mfma16(w1a[r][0][i], x0[i], dacc[r]);
#pragma unroll
          for (int i = 0; i < 4; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) dacc[r] = mfma16(w1a[r][1][i], x1[i], dacc[r]);
#ifdef PR_DIAG_NO_CMAX
          if (dacc[0][0] == 1.2345e30f)
#endif
          if (one) {
#pragma unroll
            for (int rh = 0; rh < 2; rh++) {
              float m[8];
#pragma unroll
              for (int i = 0; i < 4; i++) {
                m[i] = valid ? dacc[2 * rh][i] : -__builtin_inff();
                m[4 + i] = valid ? dacc[2 * rh + 1][i] : -__builtin_inff();
              }
              row_max8(m);
              if (valid && n == 0) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                  __hip_atomic_fetch_max(&db[(16 * (2 * rh) + i) * 32], m[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                  __hip_atomic_fetch_max(&db[(16 * (2 * rh + 1) + i) * 32], m[4 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
              }
            }
          } else if (valid) {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
              for (int i = 0; i < 4; i++)
                __hip_atomic_fetch_max(&db[(16 * r + i) * 32], dacc[r][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        };
#pragma unroll
        for (int j = 0; j < NPRE; j++)
          if (j < ncol) phase_c(cur[j], lpj[j], rec_ok(cur[j], a + 16 * j + n), onej[j]);
        walk_cols([&](const Rec &r, int j) -> int {
          const bool in = a + 16 * j + n < bslot;
          const int rk = __float_as_int(r.q.y);
          const bool valid = in && rk >= 0;
          const int lp = local_pillar(r, valid);
          const int jn = crowd_jump(j, rk, in);
          if (jn == j) phase_c(r, lp, valid, (head_mask & 0xfffeu) == 0);      // (a column inside a crowded pillar does no arithmetic at all)
          return jn;
        });
#ifdef PR_STAMP
        if (stamp) {
          st_cols += ncol;
          st_pillars += heads;
        }
        tile_pillars = heads;
        tile_walked = ncol;
#endif
        const int np = heads;                                  // pillars of the tile (<= PR_T: each starts at another slot of the window)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

        PR_MARK(3);
        // Everything prefetched at the top of the tile is waited for HERE, in front of the first store of the tile: hipcc cannot count the
        // stores below (data-dependent loops), so any later wait on a load would be vmcnt(0) and would sit behind them.  From here to the
        // top of the next tile no vector load is waited for.
        __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0)
        // ---- phase D: per 16 pillars: out = relu(b1 + W1[:, 32:] . xmax + dmax), 16-byte stores; the LDS words are reset as they are read ----
        for (int k = 0; k * 16 < np; k++) {
          const int lp = n + 16 * k;
          const bool live = lp < np;
          const int col = (lp + skew) & 31;
          float *xb = xmax + (4 * g) * 32 + col, *db = dmax + (4 * g) * 32 + col;
          float xm0[4], xm1[4];
#pragma unroll
          for (int i = 0; i < 4; i++) {
            xm0[i] = __hip_atomic_exchange(&xb[i * 32], 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            xm1[i] = __hip_atomic_exchange(&xb[(16 + i) * 32], 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          f32x4a o[4];
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(&s_b1[16 * r + 4 * g]);
            o[r] = f32x4a{bv.x, bv.y, bv.z, bv.w};
          }
          float dm[4][4];
#pragma unroll
          for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 4; i++)
              dm[r][i] = __hip_atomic_exchange(&db[(16 * r + i) * 32], -__builtin_inff(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          const int2 pi = *reinterpret_cast<const int2 *>(&pinfo[min(lp, PR_MAXP - 1) * 4]);
#pragma unroll
          for (int i = 0; i < 4; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) o[r] = mfma16(w1b[r][0][i], xm0[i], o[r]);
#pragma unroll
          for (int i = 0; i < 4; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) o[r] = mfma16(w1b[r][1][i], xm1[i], o[r]);
          if (live) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
              f32x4 v;
              v.x = relu_f(o[r][0] + dm[r][0]);
              v.y = relu_f(o[r][1] + dm[r][1]);
              v.z = relu_f(o[r][2] + dm[r][2]);
              v.w = relu_f(o[r][3] + dm[r][3]);
              if (p.pillar_features) *reinterpret_cast<f32x4 *>(p.pillar_features + (long long)pi.x * 64 + 16 * r + 4 * g) = v;
              if (p.canvas) *reinterpret_cast<f32x4 *>(p.canvas + (long long)pi.y * 64 + 16 * r + 4 * g) = v;
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // the sums of the next tile start from zero (the means, ranks and canvas rows sat in their memory)
        *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(sum) + lane * 16) = make_uint4(0u, 0u, 0u, 0u);

        PR_MARK(4);
      }
      PR_MARK(5);
#ifdef PR_STAMP
      if (lane == 0 && q0 < 65536) {
        pr_tile_cycles[4 * q0] = (unsigned)(__builtin_amdgcn_s_memtime() - tile_t0);
        pr_tile_cycles[4 * q0 + 1] = (unsigned)(bslot - a);
        pr_tile_cycles[4 * q0 + 2] = (unsigned)tile_pillars;
        pr_tile_cycles[4 * q0 + 3] = (unsigned)tile_walked;
      }
#endif
      q0 = q1; q1 = q2; q2 = q3;
      a0 = a1; b0 = b1;
      a1 = a2; b1 = b2;
      a2 = a3; b2 = b3;
#ifdef PR_STAMP
      tiles_done++;
#endif
    };
    // ONE copy of the tile's code (alternating the two record buffers between two copies doubled the kernel; the copy below is 10 - 18 moves)
    while (q0 != BIG) {
      tile_body(bufa, bufb);
#pragma unroll
      for (int j = 0; j < NPRE; j++) bufa[j] = bufb[j];
    }
#ifdef PR_STAMP
    if (stamp && lane == 0) pr_dbg[8] = (unsigned long long)tiles_done;
#endif
  }

#ifdef PR_STAMP
  wave_t1 = __builtin_amdgcn_s_memtime();
#endif
  // =================================== single-point pillars: 64 per unit, no per-pillar reduction ============================================
  // The only point of a pillar is its own mean (f_cluster = 0, exactly what scatter_mean of one value gives) and its own maximum:
  // out = relu(b1 + (W1[:, :32] + W1[:, 32:]) . relu(W0 f + b0)) -- one 32-deep product instead of two, no LDS traffic at all.
  const int s_per = (n_sing + n_waves - 1) / n_waves;
  const int s_begin = gw * s_per, s_end = min(s_begin + s_per, n_sing);
  if (s_begin < s_end) {
    f32x4 w1c[4][2];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int b = 0; b < 2; b++) w1c[r][b] = w1a[r][b] + w1b[r][b];
    Rec rec[4], recn[4];
    auto load4 = [&](Rec (&dst)[4], int unit) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int slot = Nm + 64 * unit + 16 * j + n;
        dst[j].a = 0.f;
        dst[j].q = f32x4{0.f, 0.f, 0.f, 0.f};
        dst[j].e4 = dst[j].e5 = dst[j].e6g = dst[j].e10 = 0.f;
        if (slot < Nv) load_rec(dst[j], slot);
      }
    };
    auto unit_body = [&](int unit, Rec (&cur)[4], Rec (&nxt)[4]) {
      if (unit + 1 < s_end) load4(nxt, unit + 1);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const bool valid = Nm + 64 * unit + 16 * j + n < Nv;
        f32x4a x0, x1;
        layer0(cur[j], 0.f, x0, x1);
        f32x4a o[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const f32x4 bv = *reinterpret_cast<const f32x4 *>(&s_b1[16 * r + 4 * g]);
          o[r] = f32x4a{bv.x, bv.y, bv.z, bv.w};
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int r = 0; r < 4; r++) o[r] = mfma16(w1c[r][0][i], x0[i], o[r]);
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int r = 0; r < 4; r++) o[r] = mfma16(w1c[r][1][i], x1[i], o[r]);
        if (valid) {
          const long long prow = __float_as_int(cur[j].q.y), crow = __float_as_int(cur[j].q.w);
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const f32x4 v = f32x4{relu_f(o[r][0]), relu_f(o[r][1]), relu_f(o[r][2]), relu_f(o[r][3])};
            if (p.pillar_features) *reinterpret_cast<f32x4 *>(p.pillar_features + prow * 64 + 16 * r + 4 * g) = v;
            if (p.canvas) *reinterpret_cast<f32x4 *>(p.canvas + crow * 64 + 16 * r + 4 * g) = v;
          }
        }
      }
    };
    load4(rec, s_begin);
    for (int u = s_begin; u < s_end; u += 2) {
      unit_body(u, rec, recn);
      if (u + 1 < s_end) unit_body(u + 1, recn, rec);
    }
  }
#ifdef PR_STAMP
  wave_t2 = __builtin_amdgcn_s_memtime();
#endif
  // =================================== canvas: zero rows for the empty cells ==================================================================
  // Every wave takes an equal share of the cell -> rank table, whatever the cloud looks like.  (First form of this round: a tile filled the
  // cells between ITS pillars -- even work on a uniform cloud, but in a LiDAR-like cloud the few multi-point pillars of the outskirts
  // owned tens of thousands of empty cells each: +240 us on the 6-agent cloud.)  Last in the wave's work, so nothing waits behind the
  // stores; eight 64-cell pieces of the table per trip (one wait per trip: hipcc cannot count the data-dependent stores in between).
  if (p.canvas) {
    // 64-cell pieces of the table, piece k of wave w = piece number k * waves + w (a wave's pieces sample the whole canvas: the empty
    // outskirts and the full middle of a LiDAR-like frame cost every wave the same); eight pieces per trip
    const long long c_end = p.cells;
    for (long long cb = (long long)gw * 64; cb < c_end; cb += (long long)n_waves * 512) {
      int occ[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const long long c = cb + (long long)n_waves * 64 * k + lane;
        occ[k] = c < c_end ? p.cell_rank[c] : 0;
      }
#pragma unroll
      for (int k = 0; k < 8; k++) {
        unsigned long long m = __ballot(occ[k] < 0);
        const int c0 = (int)(cb + (long long)n_waves * 64 * k);
        while (m) {
          // four empty cells per store instruction: lane group g takes the g-th lowest set bit
          int pos[4];
#pragma unroll
          for (int q = 0; q < 4; q++) {
            pos[q] = m ? __builtin_ctzll(m) : -1;
            m &= m - 1;
          }
          const int mine = g == 0 ? pos[0] : (g == 1 ? pos[1] : (g == 2 ? pos[2] : pos[3]));
          if (mine >= 0) {
            const int cell = c0 + mine;
            const int fb = div_magic(cell, p.plane_m, p.plane_sh), rem = cell - fb * plane;
            const int cx = div_magic(rem, p.ny_m, p.ny_sh), cy = rem - cx * p.g.ny;
            const long long row = ((long long)fb * p.g.ny + cy) * p.g.nx + cx;
            *reinterpret_cast<f32x4 *>(p.canvas + row * 64 + 4 * n) = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
      }
    }
  }
#ifdef PR_STAMP
  if (lane == 0 && bid * 4 + wave < 4096) {
    const unsigned long long wave_t3 = __builtin_amdgcn_s_memtime();
    pr_wave_cycles[3 * gw] = wave_t1 - wave_t0;
    pr_wave_cycles[3 * gw + 1] = wave_t2 - wave_t1;
    pr_wave_cycles[3 * gw + 2] = wave_t3 - wave_t2;
  }
  if (stamp && lane == 0) {
    for (int k = 0; k < 8; k++) pr_dbg[k] = st_acc[k];
    pr_dbg[9] = (unsigned long long)st_cols;
    pr_dbg[10] = (unsigned long long)st_pillars;
  }
#endif
}

}  // namespace

#ifdef PR_STAMP
extern "C" int pcp_debug_read_pfn_rows(void *dst, size_t bytes) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(pr_dbg), bytes) == hipSuccess ? 0 : 3;
}
extern "C" int pcp_debug_read_pfn_tile_cycles(void *dst, size_t bytes) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(pr_tile_cycles), bytes) == hipSuccess ? 0 : 3;
}
extern "C" int pcp_debug_read_pfn_wave_cycles(void *dst, size_t bytes) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(pr_wave_cycles), bytes) == hipSuccess ? 0 : 3;
}
#endif

extern "C" int pcp_pfn_rows(const pcp_grid_t *grid, const void *workspace, int64_t n, int32_t num_raw, const float *w0, const float *b0,
                            const float *w1, const float *b1, float *pillar_features, float *canvas, void *stream_) {
  if (!grid || !workspace || !w0 || !b0 || !w1 || !b1 || n < 0) return PCP_ERR_ARG;
  if ((((uintptr_t)w1) & 15) || (((uintptr_t)b1) & 15) || (((uintptr_t)pillar_features) & 15) || (((uintptr_t)canvas) & 15)) return PCP_ERR_ARG;
  if (grid->batch_size <= 0 || grid->nx <= 0 || grid->ny <= 0 || grid->nx > 65535 || grid->ny > 65535) return PCP_ERR_ARG;
  const int64_t cells = (int64_t)grid->batch_size * grid->nx * grid->ny;
  if (cells >= (1LL << 31) || n >= (1LL << 31)) return PCP_ERR_UNSUPPORTED;
  if (num_raw != 3 && num_raw != 4 && num_raw != 5 && num_raw != 11) return PCP_ERR_UNSUPPORTED;
  const RowsLayout R = pcp_rows_layout(cells, n > 0 ? n : 1, num_raw);
  const char *ws = (const char *)workspace;
  PrParams p;
  p.srows = (const float *)(ws + R.srows);
  p.tile_desc = (const int2 *)(ws + R.tile_desc);
  p.counters = (const int *)(ws + R.v.counters);
  p.cell_rank = (const int *)(ws + R.v.cell_rank);
  p.crowd_list = (const int4 *)(ws + R.crowd_list);
  p.w0 = w0; p.b0 = b0; p.w1 = w1; p.b1 = b1;
  p.pillar_features = pillar_features;
  p.canvas = canvas;
  p.g = *grid;
  p.cells = cells;
  p.n_tiles_max = (int)(n / PR_T + 1);
  magic_div((unsigned)(grid->nx * grid->ny), &p.plane_m, &p.plane_sh);
  magic_div((unsigned)grid->ny, &p.ny_m, &p.ny_sh);
  // waves per SIMD the build is held to: three fit since the kernel shrank to 168 registers (5 raw columns; 11 would spill) and pay on the
  // large clouds (-3 % at 1.2 - 1.4 M points, -6 % on the LiDAR-like one), two stay better on small ones (+7 % at 240 k); PCP_OPT_PFN_WPS overrides
  const int wps_opt = (int)pcp_option(PCP_OPT_PFN_WPS, 0);
  const int wps = (wps_opt == 2 || wps_opt == 3) ? wps_opt : ((num_raw <= 5 && n >= 600000) ? 3 : 2);
  int blocks = (p.n_tiles_max + 3) / 4;
  if (canvas) {                                 // the waves also share the canvas's empty cells: at most ~512 cells each, however small the cloud
    const int64_t by_cells = (cells + 4 * 512 - 1) / (4 * 512);
    if (by_cells > blocks) blocks = (int)(by_cells < 256 * wps ? by_cells : 256 * wps);
  }
  if (blocks > 256 * wps) blocks = 256 * wps;
  // the crowded pillars the wave tiles pass over run on workgroups at the FRONT of the same grid (none in most clouds: those workgroups
  // read the list length and leave)
  const int cb_opt = (int)pcp_option(PCP_OPT_PFN_CROWD_BLOCKS, 0);                                                  // diagnostic override
  p.crowd_blocks = (cb_opt > 0 && cb_opt <= 4096) ? cb_opt : 128;         // 64 -> 128: 6-agent ring cloud 230 -> 197 us (more than 128: nothing); no crowded pillar, no cost
  blocks += p.crowd_blocks;
  hipStream_t stream = (hipStream_t)stream_;
#define PCP_PFN_ROWS(NR)                                                                                            \
  do {                                                                                                              \
    if (wps == 3) hipLaunchKernelGGL((k_pfn_rows<NR, 3>), dim3(blocks), dim3(PR_THREADS), 0, stream, p);             \
    else hipLaunchKernelGGL((k_pfn_rows<NR, 2>), dim3(blocks), dim3(PR_THREADS), 0, stream, p);                      \
  } while (0)
  switch (num_raw) {
    case 5: PCP_PFN_ROWS(5); break;
    case 11: PCP_PFN_ROWS(11); break;
    case 3: PCP_PFN_ROWS(3); break;
    case 4: PCP_PFN_ROWS(4); break;
    default: return PCP_ERR_UNSUPPORTED;
  }
#undef PCP_PFN_ROWS
  PCP_CHECK_LAUNCH();
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}
