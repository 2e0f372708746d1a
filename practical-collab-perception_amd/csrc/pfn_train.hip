// Training-mode PillarFeatureNet: the gather / per-pillar max / argmax-routing pieces around the two Linear layers.
//
// In train() mode the two BatchNorm1d layers (dynamic_pillar_vfe.py:29,40) normalise with statistics over ALL points of the batch,
// so the fused single-kernel PFN of the inference path (pfn.hip) cannot be used: the statistics sit between the Linear and the
// ReLU / scatter_max.  The step is split at those two global reductions; per-point tensors live in BUCKET ORDER (slot s of the
// counting sort pcp_voxelize leaves in its workspace: the points of a pillar are one contiguous run), the Linear layers and
// their weight / data gradients run on the MFMA pointwise kernels, BatchNorm on bn_train.hip, and this file supplies
//   forward   features  (scatter_mean, f_cluster, f_center, concat: dynamic_pillar_vfe.py:110-126)  -> F (N', 16)
//             mid       a0 = relu(bn0(x0)); m0 = scatter_max(a0); in1 = [a0, m0[inv]]   (:35-46, first PFNLayerV2)
//             out       scatter_max(relu(bn1(x1))) -> pillar_features, canvas            (:35-46 last layer; pointpillar_scatter.py:14-37)
//   backward  route_out dL/d relu(bn1(x1)) = dL/d pillar at the arg-max row, 0 elsewhere (scatter_max backward)
//             route_mid dL/d a0 = d in1[:, :32] + [row is arg-max] * sum_pillar d in1[:, 32:]
// Arg-max ties go to the first row in bucket order (torch_scatter keeps one index; ties need bit-identical rows).
#include "pcp_common.h"
#include "../../include/pcp_hip_mp.h"

namespace {

constexpr int PT_THREADS = 256;
constexpr int PT_PILLARS = 64;
// padded feature width: 16 floats for num_raw + 6 <= 16 (the 5-feature configs), 32 for the 11-feature lately-fusion cloud
constexpr int C0 = 32, C1 = 64;

struct FeatParams {
  const float *points;
  int stride;
  pcp_grid_t g;
  const int *bucket_order, *pillar_cell, *pillar_start, *counters;
  float *fbuf;
  int *slot_pillar;
};

__device__ __forceinline__ int find_pillar(const int *pl_start, int np, int slot) {
  int lo = 0, hi = np - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (pl_start[mid] <= slot) lo = mid; else hi = mid - 1;
  }
  return lo;
}

template <int NUM_RAW>
__global__ __launch_bounds__(PT_THREADS) void k_pfnt_feat(FeatParams p) {
  constexpr int F = NUM_RAW + 6;
  constexpr int FW = F <= 16 ? 16 : 32;
  static_assert(F <= FW, "feature row is padded to 16 or 32 floats");
  __shared__ int pl_start[PT_PILLARS + 1];
  __shared__ long long sum_fx[PT_PILLARS][3];
  __shared__ float mean[PT_PILLARS][3];
  const int P = p.counters[0];
  const int r0 = blockIdx.x * PT_PILLARS;
  if (r0 >= P) return;
  const int np = min(PT_PILLARS, P - r0);
  const int tid = threadIdx.x;
  for (int i = tid; i <= np; i += PT_THREADS) pl_start[i] = p.pillar_start[r0 + i];
  for (int i = tid; i < PT_PILLARS * 3; i += PT_THREADS) (&sum_fx[0][0])[i] = 0;
  __syncthreads();
  const int s0 = pl_start[0], s1 = pl_start[np];
  // per-pillar mean exactly as the inference kernel: 2^-24 fixed point sums (deterministic), one division
  for (int s = s0 + tid; s < s1; s += PT_THREADS) {
    const float *row = p.points + (long long)p.bucket_order[s] * p.stride;
    const int pl = find_pillar(pl_start, np, s);
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const long long q = __double2ll_rn((double)row[1 + a] * 16777216.0);
      atomicAdd(reinterpret_cast<unsigned long long *>(&sum_fx[pl][a]), (unsigned long long)q);
    }
  }
  __syncthreads();
  for (int i = tid; i < np * 3; i += PT_THREADS) {
    const int pl = i / 3, a = i % 3;
    const int cnt = pl_start[pl + 1] - pl_start[pl];
    mean[pl][a] = (float)(((double)sum_fx[pl][a] * (1.0 / 16777216.0)) / (double)cnt);
  }
  __syncthreads();
  const int plane = p.g.nx * p.g.ny;
  const float x_off = __fadd_rn(p.g.voxel_x * 0.5f, p.g.min_x);
  const float y_off = __fadd_rn(p.g.voxel_y * 0.5f, p.g.min_y);
  const float z_off = __fadd_rn(p.g.voxel_z * 0.5f, p.g.min_z);
  for (int s = s0 + tid; s < s1; s += PT_THREADS) {
    const int pl = find_pillar(pl_start, np, s);
    const float *row = p.points + (long long)p.bucket_order[s] * p.stride;
    float f[FW];
#pragma unroll
    for (int k = 0; k < FW; k++) f[k] = 0.f;
#pragma unroll
    for (int k = 0; k < NUM_RAW; k++) f[k] = row[1 + k];
    const int rem = p.pillar_cell[r0 + pl] % plane;
    const float cx = (float)(rem / p.g.ny), cy = (float)(rem % p.g.ny);
    f[NUM_RAW + 0] = __fsub_rn(f[0], mean[pl][0]);
    f[NUM_RAW + 1] = __fsub_rn(f[1], mean[pl][1]);
    f[NUM_RAW + 2] = __fsub_rn(f[2], mean[pl][2]);
    f[NUM_RAW + 3] = __fsub_rn(f[0], __fadd_rn(__fmul_rn(cx, p.g.voxel_x), x_off));
    f[NUM_RAW + 4] = __fsub_rn(f[1], __fadd_rn(__fmul_rn(cy, p.g.voxel_y), y_off));
    f[NUM_RAW + 5] = __fsub_rn(f[2], z_off);
    float4 *o = reinterpret_cast<float4 *>(p.fbuf + (long long)s * FW);
#pragma unroll
    for (int q = 0; q < FW / 4; q++) o[q] = make_float4(f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]);
    p.slot_pillar[s] = r0 + pl;
  }
}

// The PillarFeatureNet variants no config of the reference uses (WITH_DISTANCE, USE_ABSLOTE_XYZ False, any raw width): the same feature
// rows with the composition decided at run time (dynamic_pillar_vfe.py:117-126).  `fw` floats per row, zero padded.
struct FeatAnyParams {
  FeatParams b;
  int num_raw, use_abs, with_dist, fw;
};

__global__ __launch_bounds__(PT_THREADS) void k_pfn_feat_any(FeatAnyParams q) {
  const FeatParams &p = q.b;
  __shared__ int pl_start[PT_PILLARS + 1];
  __shared__ long long sum_fx[PT_PILLARS][3];
  __shared__ float mean[PT_PILLARS][3];
  const int P = p.counters[0];
  const int r0 = blockIdx.x * PT_PILLARS;
  if (r0 >= P) return;
  const int np = min(PT_PILLARS, P - r0);
  const int tid = threadIdx.x;
  for (int i = tid; i <= np; i += PT_THREADS) pl_start[i] = p.pillar_start[r0 + i];
  for (int i = tid; i < PT_PILLARS * 3; i += PT_THREADS) (&sum_fx[0][0])[i] = 0;
  __syncthreads();
  const int s0 = pl_start[0], s1 = pl_start[np];
  for (int s = s0 + tid; s < s1; s += PT_THREADS) {
    const float *row = p.points + (long long)p.bucket_order[s] * p.stride;
    const int pl = find_pillar(pl_start, np, s);
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const long long v = __double2ll_rn((double)row[1 + a] * 16777216.0);
      atomicAdd(reinterpret_cast<unsigned long long *>(&sum_fx[pl][a]), (unsigned long long)v);
    }
  }
  __syncthreads();
  for (int i = tid; i < np * 3; i += PT_THREADS) {
    const int pl = i / 3, a = i % 3;
    const int cnt = pl_start[pl + 1] - pl_start[pl];
    mean[pl][a] = (float)(((double)sum_fx[pl][a] * (1.0 / 16777216.0)) / (double)cnt);
  }
  __syncthreads();
  const int plane = p.g.nx * p.g.ny;
  const float x_off = __fadd_rn(p.g.voxel_x * 0.5f, p.g.min_x);
  const float y_off = __fadd_rn(p.g.voxel_y * 0.5f, p.g.min_y);
  const float z_off = __fadd_rn(p.g.voxel_z * 0.5f, p.g.min_z);
  const int first = q.use_abs ? 0 : 3;            // points[:, 1:] or points[:, 4:]
  const int nlead = q.num_raw - first;
  for (int s = s0 + tid; s < s1; s += PT_THREADS) {
    const int pl = find_pillar(pl_start, np, s);
    const float *row = p.points + (long long)p.bucket_order[s] * p.stride;
    float *o = p.fbuf + (long long)s * q.fw;
    const float x = row[1], y = row[2], z = row[3];
    for (int k = 0; k < nlead; k++) o[k] = row[1 + first + k];
    const int rem = p.pillar_cell[r0 + pl] % plane;
    const float cx = (float)(rem / p.g.ny), cy = (float)(rem % p.g.ny);
    o[nlead + 0] = __fsub_rn(x, mean[pl][0]);
    o[nlead + 1] = __fsub_rn(y, mean[pl][1]);
    o[nlead + 2] = __fsub_rn(z, mean[pl][2]);
    o[nlead + 3] = __fsub_rn(x, __fadd_rn(__fmul_rn(cx, p.g.voxel_x), x_off));
    o[nlead + 4] = __fsub_rn(y, __fadd_rn(__fmul_rn(cy, p.g.voxel_y), y_off));
    o[nlead + 5] = __fsub_rn(z, z_off);
    int k = nlead + 6;
    if (q.with_dist) o[k++] = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z)));
    for (; k < q.fw; k++) o[k] = 0.f;
    p.slot_pillar[s] = r0 + pl;
  }
}

// in_next[r] = [y[r, :c], pillar_max[slot_pillar[r], :c], 0...]: the concat at the end of a non-last PFNLayerV2 (dynamic_pillar_vfe.py:44-46)
__global__ __launch_bounds__(PT_THREADS) void k_pfn_cat_pillar_max(const float *__restrict__ y, int ld_y, const float *__restrict__ pmax, int ld_max,
                                                                   const int *__restrict__ slot_pillar, long long rows, int c,
                                                                   float *__restrict__ out, int ld_out) {
  const long long total = rows * ld_out;
  for (long long i = (long long)blockIdx.x * PT_THREADS + threadIdx.x; i < total; i += (long long)gridDim.x * PT_THREADS) {
    const long long r = i / ld_out;
    const int k = (int)(i - r * ld_out);
    float v = 0.f;
    if (k < c) v = y[r * ld_y + k];
    else if (k < 2 * c) v = pmax[(long long)slot_pillar[r] * ld_max + (k - c)];
    out[i] = v;
  }
}

// The four routing kernels: a SUB-GROUP of lanes per pillar (8 lanes x 4 channels for the 32-channel rows, 16 lanes x 4 channels for the
// 64-channel rows), so a wave works on 8 / 4 pillars at once with 16-byte accesses; a pillar's rows (two on average) are walked in bucket
// order by its own lanes, no cross-lane traffic.  (Rounds 1-3 gave a whole wave to a pillar: 64 lanes x 4 bytes per row and one pillar's
// chain of dependent loads per wave -- 214 .. 306 us per kernel on 1.4 M points.)
// RT: storage type of the per-point 64-channel rows (in1, x1, dz1, din1): float, or __bf16 in the bf16 training loop
// (include/pcp_hip_mp.h) -- the second PFN Linear then runs on pcp_mp_pointwise.
template <typename T> struct Row4;
template <> struct Row4<float> {
  static __device__ __forceinline__ f32x4 ld(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
  static __device__ __forceinline__ void st(float *p, const f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }
};
template <> struct Row4<__bf16> {
  typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ f32x4 ld(const __bf16 *p) {
    const bf16x4_t v = *reinterpret_cast<const bf16x4_t *>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  }
  static __device__ __forceinline__ void st(__bf16 *p, const f32x4 v) {
    bf16x4_t o;
    o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
    *reinterpret_cast<bf16x4_t *>(p) = o;
  }
};
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <typename RT>
__global__ __launch_bounds__(PT_THREADS) void k_pfnt_mid(const int *__restrict__ pillar_start, const int *__restrict__ counters,
                                                        const float *__restrict__ x0, const float *__restrict__ scale,
                                                        const float *__restrict__ shift, RT *__restrict__ in1, int *__restrict__ arg0) {
  const int P = counters[0];
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long nt = (long long)gridDim.x * blockDim.x;
  const int c = (int)(t & 7) * 4;                                  // this lane's four channels of the 32
  const f32x4 sc = *reinterpret_cast<const f32x4 *>(scale + c), sh = *reinterpret_cast<const f32x4 *>(shift + c);
  for (long long p = t >> 3; p < P; p += nt >> 3) {
    const int sb = pillar_start[p], se = pillar_start[p + 1];
    if (se - sb > PCP_LONG_PILLAR) continue;                        // k_pfnt_mid_long: a workgroup per long pillar
    f32x4 best = f32x4{-1.f, -1.f, -1.f, -1.f};
    i32x4 arg = i32x4{se, se, se, se};
    for (int s = sb; s < se; ++s) {
      const f32x4 x = *reinterpret_cast<const f32x4 *>(x0 + (long long)s * C0 + c);
      f32x4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i] = fmaxf(fmaf(x[i], sc[i], sh[i]), 0.f);
        if (v[i] > best[i]) { best[i] = v[i]; arg[i] = s; }         // first row in bucket order wins a tie
      }
      Row4<RT>::st(in1 + (long long)s * C1 + c, v);
    }
    *reinterpret_cast<i32x4 *>(arg0 + p * C0 + c) = arg;
    for (int s = sb; s < se; ++s) Row4<RT>::st(in1 + (long long)s * C1 + C0 + c, best);
  }
}

// CT: storage type of the canvas (float; __bf16 in the bf16 training loop: the first backbone layer and its weight gradient read the
// canvas as bf16 anyway, so the fp32 canvas + its cast are skipped)
template <typename CT, typename RT>
__global__ __launch_bounds__(PT_THREADS) void k_pfnt_out(const int *__restrict__ pillar_start, const int *__restrict__ pillar_cell,
                                                        const int *__restrict__ counters, pcp_grid_t g, const RT *__restrict__ x1,
                                                        const float *__restrict__ scale, const float *__restrict__ shift,
                                                        float *__restrict__ pf, int *__restrict__ arg1, CT *__restrict__ canvas) {
  const int P = counters[0];
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long nt = (long long)gridDim.x * blockDim.x;
  const int c = (int)(t & 15) * 4;                                 // this lane's four channels of the 64
  const f32x4 sc = *reinterpret_cast<const f32x4 *>(scale + c), sh = *reinterpret_cast<const f32x4 *>(shift + c);
  const int plane = g.nx * g.ny;
  for (long long p = t >> 4; p < P; p += nt >> 4) {
    const int sb = pillar_start[p], se = pillar_start[p + 1];
    if (se - sb > PCP_LONG_PILLAR) continue;                        // k_pfnt_out_long
    f32x4 best = f32x4{-1.f, -1.f, -1.f, -1.f};
    i32x4 arg = i32x4{sb, sb, sb, sb};
    for (int s = sb; s < se; ++s) {
      const f32x4 x = Row4<RT>::ld(x1 + (long long)s * C1 + c);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float v = fmaxf(fmaf(x[i], sc[i], sh[i]), 0.f);
        if (v > best[i]) { best[i] = v; arg[i] = s; }
      }
    }
    if (pf) *reinterpret_cast<f32x4 *>(pf + p * C1 + c) = best;
    *reinterpret_cast<i32x4 *>(arg1 + p * C1 + c) = arg;
    if (canvas) {
      const int cell = pillar_cell[p];
      const int b = cell / plane, rem = cell % plane;
      const int cx = rem / g.ny, cy = rem % g.ny;
      Row4<CT>::st(canvas + (((long long)b * g.ny + cy) * g.nx + cx) * C1 + c, best);
    }
  }
}

// writes EVERY row of dz1 (the gradient at the arg-max row, zero elsewhere): no zero fill of the 357-MB tensor in front of it
template <typename CT, typename RT>
__global__ __launch_bounds__(PT_THREADS) void k_pfnt_route_out(const int *__restrict__ pillar_start, const int *__restrict__ pillar_cell,
                                                              const int *__restrict__ counters, pcp_grid_t g, const CT *__restrict__ dcanvas,
                                                              const float *__restrict__ dpf, const int *__restrict__ arg1,
                                                              RT *__restrict__ dz1) {
  const int P = counters[0];
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long nt = (long long)gridDim.x * blockDim.x;
  const int c = (int)(t & 15) * 4;
  const int plane = g.nx * g.ny;
  for (long long p = t >> 4; p < P; p += nt >> 4) {
    const int sb = pillar_start[p], se = pillar_start[p + 1];
    if (se - sb > PCP_LONG_PILLAR) continue;                        // k_pfnt_route_out_long
    f32x4 gv;
    if (dcanvas) {
      const int cell = pillar_cell[p];
      const int b = cell / plane, rem = cell % plane;
      const int cx = rem / g.ny, cy = rem % g.ny;
      gv = Row4<CT>::ld(dcanvas + (((long long)b * g.ny + cy) * g.nx + cx) * C1 + c);
    } else {
      gv = *reinterpret_cast<const f32x4 *>(dpf + p * C1 + c);
    }
    const i32x4 arg = *reinterpret_cast<const i32x4 *>(arg1 + p * C1 + c);
    for (int s = sb; s < se; ++s) {
      f32x4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = arg[i] == s ? gv[i] : 0.f;
      Row4<RT>::st(dz1 + (long long)s * C1 + c, v);
    }
  }
}

template <typename RT>
__global__ __launch_bounds__(PT_THREADS) void k_pfnt_route_mid(const int *__restrict__ pillar_start, const int *__restrict__ counters,
                                                              const RT *__restrict__ din1, const int *__restrict__ arg0,
                                                              float *__restrict__ da0) {
  const int P = counters[0];
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long nt = (long long)gridDim.x * blockDim.x;
  const int c = (int)(t & 7) * 4;
  for (long long p = t >> 3; p < P; p += nt >> 3) {
    const int sb = pillar_start[p], se = pillar_start[p + 1];
    if (se - sb > PCP_LONG_PILLAR) continue;                        // k_pfnt_route_mid_long
    // fixed summation order (ascending slot)
    f32x4 dm = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = sb; s < se; ++s) dm += Row4<RT>::ld(din1 + (long long)s * C1 + C0 + c);
    const i32x4 a = *reinterpret_cast<const i32x4 *>(arg0 + p * C0 + c);
    for (int s = sb; s < se; ++s) {
      f32x4 v = Row4<RT>::ld(din1 + (long long)s * C1 + c);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += (s == a[i] ? dm[i] : 0.f);
      *reinterpret_cast<f32x4 *>(da0 + (long long)s * C0 + c) = v;
    }
  }
}

// ---- pillars of more than PCP_LONG_PILLAR points: a workgroup each --------------------------------------------------------------------------
// The four kernels above give a pillar one group of 8 / 16 lanes that walks its points one after the other: fine for the handful of points a
// pillar usually has, milliseconds for the cells next to the sensor of a LiDAR-like cloud (hundreds to thousands of points; the wave waits
// for its longest pillar).  The pillariser lists those pillars (VoxLayout::long_list); here 256 threads = 32 (mid) / 16 (out) row slots x the
// channel lanes share one listed pillar: every row slot walks rows s = sb + slot, + slots, ..., the per-slot results are combined in LDS in a
// FIXED order (max: larger value, then smaller slot index -- the first row in bucket order still wins a tie; sums: slot 0, 1, 2, ...), so the
// results are reproducible; the sums of route_mid are grouped differently from the sequential loop (last-bit differences).
constexpr int PL_THREADS = 256;
constexpr int PL_BLOCKS = 1024;                                  // grid-stride over the list

template <typename RT>
__global__ __launch_bounds__(PL_THREADS) void k_pfnt_mid_long(const int *__restrict__ pillar_start, const int *__restrict__ long_list,
                                                             const int *__restrict__ counters, const float *__restrict__ x0,
                                                             const float *__restrict__ scale, const float *__restrict__ shift,
                                                             RT *__restrict__ in1, int *__restrict__ arg0) {
  constexpr int SLOTS = PL_THREADS / 8;                            // 32 row slots x 8 lanes of four channels
  __shared__ float s_best[SLOTS][C0];
  __shared__ int s_arg[SLOTS][C0];
  const int tid = threadIdx.x, slot = tid >> 3, c = (tid & 7) * 4;
  const f32x4 sc = *reinterpret_cast<const f32x4 *>(scale + c), sh = *reinterpret_cast<const f32x4 *>(shift + c);
  const int nl = counters[6];
  for (int li = blockIdx.x; li < nl; li += gridDim.x) {
    const int p = long_list[li];
    const int sb = pillar_start[p], se = pillar_start[p + 1];
    f32x4 best = f32x4{-1.f, -1.f, -1.f, -1.f};
    i32x4 arg = i32x4{se, se, se, se};
    for (int s = sb + slot; s < se; s += SLOTS) {
      const f32x4 x = *reinterpret_cast<const f32x4 *>(x0 + (long long)s * C0 + c);
      f32x4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i] = fmaxf(fmaf(x[i], sc[i], sh[i]), 0.f);
        if (v[i] > best[i]) { best[i] = v[i]; arg[i] = s; }
      }
      Row4<RT>::st(in1 + (long long)s * C1 + c, v);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { s_best[slot][c + i] = best[i]; s_arg[slot][c + i] = arg[i]; }
    __syncthreads();
    // every thread combines the 32 slots of its four channels itself (same result in all of them: no second broadcast)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float b = -1.f;
      int a = se;
      for (int k = 0; k < SLOTS; ++k) {
        const float v = s_best[k][c + i];
        const int av = s_arg[k][c + i];
        if (v > b || (v == b && av < a)) { b = v; a = av; }
      }
      best[i] = b;
      arg[i] = a;
    }
    if (slot == 0) *reinterpret_cast<i32x4 *>(arg0 + (long long)p * C0 + c) = arg;
    for (int s = sb + slot; s < se; s += SLOTS) Row4<RT>::st(in1 + (long long)s * C1 + C0 + c, best);
    __syncthreads();
  }
}

template <typename CT, typename RT>
__global__ __launch_bounds__(PL_THREADS) void k_pfnt_out_long(const int *__restrict__ pillar_start, const int *__restrict__ pillar_cell,
                                                             const int *__restrict__ long_list, const int *__restrict__ counters,
                                                             pcp_grid_t g, const RT *__restrict__ x1, const float *__restrict__ scale,
                                                             const float *__restrict__ shift, float *__restrict__ pf, int *__restrict__ arg1,
                                                             CT *__restrict__ canvas) {
  constexpr int SLOTS = PL_THREADS / 16;                           // 16 row slots x 16 lanes of four channels
  __shared__ float s_best[SLOTS][C1];
  __shared__ int s_arg[SLOTS][C1];
  const int tid = threadIdx.x, slot = tid >> 4, c = (tid & 15) * 4;
  const f32x4 sc = *reinterpret_cast<const f32x4 *>(scale + c), sh = *reinterpret_cast<const f32x4 *>(shift + c);
  const int plane = g.nx * g.ny;
  const int nl = counters[6];
  for (int li = blockIdx.x; li < nl; li += gridDim.x) {
    const int p = long_list[li];
    const int sb = pillar_start[p], se = pillar_start[p + 1];
    f32x4 best = f32x4{-1.f, -1.f, -1.f, -1.f};
    i32x4 arg = i32x4{sb, sb, sb, sb};
    for (int s = sb + slot; s < se; s += SLOTS) {
      const f32x4 x = Row4<RT>::ld(x1 + (long long)s * C1 + c);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float v = fmaxf(fmaf(x[i], sc[i], sh[i]), 0.f);
        if (v > best[i]) { best[i] = v; arg[i] = s; }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { s_best[slot][c + i] = best[i]; s_arg[slot][c + i] = arg[i]; }
    __syncthreads();
    if (slot == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float b = -1.f;
        int a = sb;
        for (int k = 0; k < SLOTS; ++k) {
          const float v = s_best[k][c + i];
          const int av = s_arg[k][c + i];
          if (v > b || (v == b && v >= 0.f && av < a)) { b = v; a = av; }
        }
        best[i] = b;
        arg[i] = a;
      }
      if (pf) *reinterpret_cast<f32x4 *>(pf + (long long)p * C1 + c) = best;
      *reinterpret_cast<i32x4 *>(arg1 + (long long)p * C1 + c) = arg;
      if (canvas) {
        const int cell = pillar_cell[p];
        const int b = cell / plane, rem = cell % plane;
        const int cx = rem / g.ny, cy = rem % g.ny;
        Row4<CT>::st(canvas + (((long long)b * g.ny + cy) * g.nx + cx) * C1 + c, best);
      }
    }
    __syncthreads();
  }
}

template <typename CT, typename RT>
__global__ __launch_bounds__(PL_THREADS) void k_pfnt_route_out_long(const int *__restrict__ pillar_start, const int *__restrict__ pillar_cell,
                                                                   const int *__restrict__ long_list, const int *__restrict__ counters,
                                                                   pcp_grid_t g, const CT *__restrict__ dcanvas, const float *__restrict__ dpf,
                                                                   const int *__restrict__ arg1, RT *__restrict__ dz1) {
  constexpr int SLOTS = PL_THREADS / 16;
  const int tid = threadIdx.x, slot = tid >> 4, c = (tid & 15) * 4;
  const int plane = g.nx * g.ny;
  const int nl = counters[6];
  for (int li = blockIdx.x; li < nl; li += gridDim.x) {
    const int p = long_list[li];
    const int sb = pillar_start[p], se = pillar_start[p + 1];
    f32x4 gv;
    if (dcanvas) {
      const int cell = pillar_cell[p];
      const int b = cell / plane, rem = cell % plane;
      const int cx = rem / g.ny, cy = rem % g.ny;
      gv = Row4<CT>::ld(dcanvas + (((long long)b * g.ny + cy) * g.nx + cx) * C1 + c);
    } else {
      gv = *reinterpret_cast<const f32x4 *>(dpf + (long long)p * C1 + c);
    }
    const i32x4 arg = *reinterpret_cast<const i32x4 *>(arg1 + (long long)p * C1 + c);
    for (int s = sb + slot; s < se; s += SLOTS) {
      f32x4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = arg[i] == s ? gv[i] : 0.f;
      Row4<RT>::st(dz1 + (long long)s * C1 + c, v);
    }
  }
}

template <typename RT>
__global__ __launch_bounds__(PL_THREADS) void k_pfnt_route_mid_long(const int *__restrict__ pillar_start, const int *__restrict__ long_list,
                                                                   const int *__restrict__ counters, const RT *__restrict__ din1,
                                                                   const int *__restrict__ arg0, float *__restrict__ da0) {
  constexpr int SLOTS = PL_THREADS / 8;
  __shared__ float s_sum[SLOTS][C0];
  const int tid = threadIdx.x, slot = tid >> 3, c = (tid & 7) * 4;
  const int nl = counters[6];
  for (int li = blockIdx.x; li < nl; li += gridDim.x) {
    const int p = long_list[li];
    const int sb = pillar_start[p], se = pillar_start[p + 1];
    f32x4 dm = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = sb + slot; s < se; s += SLOTS) dm += Row4<RT>::ld(din1 + (long long)s * C1 + C0 + c);      // ascending rows of this slot
#pragma unroll
    for (int i = 0; i < 4; ++i) s_sum[slot][c + i] = dm[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float t = 0.f;
      for (int k = 0; k < SLOTS; ++k) t += s_sum[k][c + i];                                                  // slots in ascending order
      dm[i] = t;
    }
    const i32x4 a = *reinterpret_cast<const i32x4 *>(arg0 + (long long)p * C0 + c);
    for (int s = sb + slot; s < se; s += SLOTS) {
      f32x4 v = Row4<RT>::ld(din1 + (long long)s * C1 + c);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += (s == a[i] ? dm[i] : 0.f);
      *reinterpret_cast<f32x4 *>(da0 + (long long)s * C0 + c) = v;
    }
    __syncthreads();
  }
}

struct WsView { const int *bucket_order, *pillar_cell, *pillar_start, *counters, *long_list; };

inline WsView view_ws(const void *workspace, const pcp_grid_t *grid, int64_t n) {
  const int64_t cells = (int64_t)grid->batch_size * grid->nx * grid->ny;
  const VoxLayout L = pcp_vox_layout(cells, n);
  const char *ws = (const char *)workspace;
  WsView v;
  v.bucket_order = (const int *)(ws + L.bucket_order);
  v.pillar_cell = (const int *)(ws + L.pillar_cell);
  v.pillar_start = (const int *)(ws + L.pillar_start);
  v.counters = (const int *)(ws + L.counters);
  v.long_list = (const int *)(ws + L.long_list);
  return v;
}

inline int pillar_blocks(int64_t max_pillars, int lanes_per_pillar) {
  int64_t b = (max_pillars * lanes_per_pillar + PT_THREADS - 1) / PT_THREADS;
  if (b > 16384) b = 16384;                   // grid-stride beyond
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" {

int pcp_pfn_train_features(const float *points, int64_t n, int32_t row_stride, int32_t num_raw, const pcp_grid_t *grid,
                           const void *vox_workspace, float *fbuf, int32_t *slot_pillar, void *stream) {
  if (!points || !grid || !vox_workspace || !fbuf || !slot_pillar || n < 0 || row_stride < 1 + num_raw || num_raw < 3) return PCP_ERR_ARG;
  if (n == 0) return PCP_OK;
  const WsView v = view_ws(vox_workspace, grid, n);
  FeatParams p;
  p.points = points; p.stride = row_stride; p.g = *grid;
  p.bucket_order = v.bucket_order; p.pillar_cell = v.pillar_cell; p.pillar_start = v.pillar_start; p.counters = v.counters;
  p.fbuf = fbuf; p.slot_pillar = slot_pillar;
  const int64_t cells = (int64_t)grid->batch_size * grid->nx * grid->ny;
  const int64_t max_pillars = n < cells ? n : cells;
  const int blocks = (int)((max_pillars + PT_PILLARS - 1) / PT_PILLARS);
  hipStream_t s = (hipStream_t)stream;
  switch (num_raw) {
    case 3: hipLaunchKernelGGL(k_pfnt_feat<3>, dim3(blocks), dim3(PT_THREADS), 0, s, p); break;
    case 4: hipLaunchKernelGGL(k_pfnt_feat<4>, dim3(blocks), dim3(PT_THREADS), 0, s, p); break;
    case 5: hipLaunchKernelGGL(k_pfnt_feat<5>, dim3(blocks), dim3(PT_THREADS), 0, s, p); break;
    case 11: hipLaunchKernelGGL(k_pfnt_feat<11>, dim3(blocks), dim3(PT_THREADS), 0, s, p); break;
    default: return PCP_ERR_UNSUPPORTED;
  }
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

static int pfn_train_mid_impl(const pcp_grid_t *grid, const void *vox_workspace, int64_t n, const float *x0, const float *scale0,
                              const float *shift0, void *in1, int in1_bf16, int32_t *arg0, void *stream) {
  if (!grid || !vox_workspace || !x0 || !scale0 || !shift0 || !in1 || !arg0 || n < 0) return PCP_ERR_ARG;
  if (n == 0) return PCP_OK;
  const WsView v = view_ws(vox_workspace, grid, n);
  const int64_t cells = (int64_t)grid->batch_size * grid->nx * grid->ny;
  const dim3 gridd(pillar_blocks(n < cells ? n : cells, 8));
  if (in1_bf16)
    hipLaunchKernelGGL(k_pfnt_mid<__bf16>, gridd, dim3(PT_THREADS), 0, (hipStream_t)stream, v.pillar_start, v.counters, x0, scale0, shift0,
                       (__bf16 *)in1, arg0);
  else
    hipLaunchKernelGGL(k_pfnt_mid<float>, gridd, dim3(PT_THREADS), 0, (hipStream_t)stream, v.pillar_start, v.counters, x0, scale0, shift0,
                       (float *)in1, arg0);
  PCP_CHECK_LAUNCH();
  // the listed long pillars (none in most clouds: the workgroups read the list length and leave)
  if (in1_bf16)
    hipLaunchKernelGGL(k_pfnt_mid_long<__bf16>, dim3(PL_BLOCKS), dim3(PL_THREADS), 0, (hipStream_t)stream, v.pillar_start, v.long_list, v.counters, x0,
                       scale0, shift0, (__bf16 *)in1, arg0);
  else
    hipLaunchKernelGGL(k_pfnt_mid_long<float>, dim3(PL_BLOCKS), dim3(PL_THREADS), 0, (hipStream_t)stream, v.pillar_start, v.long_list, v.counters, x0,
                       scale0, shift0, (float *)in1, arg0);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

int pcp_pfn_train_mid(const pcp_grid_t *grid, const void *vox_workspace, int64_t n, const float *x0, const float *scale0,
                      const float *shift0, float *in1, int32_t *arg0, void *stream) {
  return pfn_train_mid_impl(grid, vox_workspace, n, x0, scale0, shift0, in1, 0, arg0, stream);
}

int pcp_mp_pfn_train_mid(const pcp_grid_t *grid, const void *vox_workspace, int64_t n, const float *x0, const float *scale0,
                         const float *shift0, void *in1, int32_t in1_dtype, int32_t *arg0, void *stream) {
  if (in1_dtype != 0 && in1_dtype != 1) return PCP_ERR_ARG;
  return pfn_train_mid_impl(grid, vox_workspace, n, x0, scale0, shift0, in1, in1_dtype, arg0, stream);
}

static int pfn_train_out_impl(const pcp_grid_t *grid, const void *vox_workspace, int64_t n, const void *x1, int x1_bf16, const float *scale1,
                              const float *shift1, float *pillar_features, int32_t *arg1, void *canvas, int canvas_bf16, void *stream) {
  if (!grid || !vox_workspace || !x1 || !scale1 || !shift1 || !arg1 || n < 0) return PCP_ERR_ARG;
  if (n == 0) return PCP_OK;
  const WsView v = view_ws(vox_workspace, grid, n);
  const int64_t cells = (int64_t)grid->batch_size * grid->nx * grid->ny;
  const dim3 gridd(pillar_blocks(n < cells ? n : cells, 16));
  hipStream_t st = (hipStream_t)stream;
#define PCP_PFNT_OUT(CT, RT)                                                                                                              \
  hipLaunchKernelGGL((k_pfnt_out<CT, RT>), gridd, dim3(PT_THREADS), 0, st, v.pillar_start, v.pillar_cell, v.counters, *grid, (const RT *)x1, \
                     scale1, shift1, pillar_features, arg1, (CT *)canvas)
  if (canvas_bf16) {
    if (x1_bf16) PCP_PFNT_OUT(__bf16, __bf16); else PCP_PFNT_OUT(__bf16, float);
  } else {
    if (x1_bf16) PCP_PFNT_OUT(float, __bf16); else PCP_PFNT_OUT(float, float);
  }
#undef PCP_PFNT_OUT
  PCP_CHECK_LAUNCH();
#define PCP_PFNT_OUT_LONG(CT, RT)                                                                                                          \
  hipLaunchKernelGGL((k_pfnt_out_long<CT, RT>), dim3(PL_BLOCKS), dim3(PL_THREADS), 0, st, v.pillar_start, v.pillar_cell, v.long_list, v.counters, \
                     *grid, (const RT *)x1, scale1, shift1, pillar_features, arg1, (CT *)canvas)
  if (canvas_bf16) {
    if (x1_bf16) PCP_PFNT_OUT_LONG(__bf16, __bf16); else PCP_PFNT_OUT_LONG(__bf16, float);
  } else {
    if (x1_bf16) PCP_PFNT_OUT_LONG(float, __bf16); else PCP_PFNT_OUT_LONG(float, float);
  }
#undef PCP_PFNT_OUT_LONG
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

int pcp_pfn_train_out(const pcp_grid_t *grid, const void *vox_workspace, int64_t n, const float *x1, const float *scale1,
                      const float *shift1, float *pillar_features, int32_t *arg1, float *canvas, void *stream) {
  return pfn_train_out_impl(grid, vox_workspace, n, x1, 0, scale1, shift1, pillar_features, arg1, canvas, 0, stream);
}

int pcp_mp_pfn_train_out(const pcp_grid_t *grid, const void *vox_workspace, int64_t n, const void *x1, int32_t x1_dtype, const float *scale1,
                         const float *shift1, float *pillar_features, int32_t *arg1, void *canvas, int32_t canvas_dtype, void *stream) {
  if ((canvas_dtype != 0 && canvas_dtype != 1) || (x1_dtype != 0 && x1_dtype != 1)) return PCP_ERR_ARG;
  return pfn_train_out_impl(grid, vox_workspace, n, x1, x1_dtype, scale1, shift1, pillar_features, arg1, canvas, canvas_dtype, stream);
}

static int pfn_route_out_impl(const pcp_grid_t *grid, const void *vox_workspace, int64_t n, int64_t kept_rows, const void *dcanvas,
                              int dcanvas_bf16, const float *dpillar, const int32_t *arg1, void *dz1, int dz1_bf16, void *stream) {
  if (!grid || !vox_workspace || !arg1 || !dz1 || n < 0 || kept_rows < 0 || kept_rows > n) return PCP_ERR_ARG;
  if ((dcanvas == nullptr) == (dpillar == nullptr)) return PCP_ERR_ARG;
  if (n == 0 || kept_rows == 0) return PCP_OK;
  const WsView v = view_ws(vox_workspace, grid, n);
  hipStream_t s = (hipStream_t)stream;
  // every kept row belongs to exactly one pillar and the kernel writes all of a pillar's rows: dz1 needs no zero fill
  const int64_t cells = (int64_t)grid->batch_size * grid->nx * grid->ny;
  const int64_t max_pillars = kept_rows < cells ? kept_rows : cells;
  const dim3 gridd(pillar_blocks(max_pillars, 16));
#define PCP_PFNT_ROUTE(CT, RT)                                                                                                           \
  hipLaunchKernelGGL((k_pfnt_route_out<CT, RT>), gridd, dim3(PT_THREADS), 0, s, v.pillar_start, v.pillar_cell, v.counters, *grid,            \
                     (const CT *)dcanvas, dpillar, arg1, (RT *)dz1)
  if (dcanvas_bf16) {
    if (dz1_bf16) PCP_PFNT_ROUTE(__bf16, __bf16); else PCP_PFNT_ROUTE(__bf16, float);
  } else {
    if (dz1_bf16) PCP_PFNT_ROUTE(float, __bf16); else PCP_PFNT_ROUTE(float, float);
  }
#undef PCP_PFNT_ROUTE
  PCP_CHECK_LAUNCH();
#define PCP_PFNT_ROUTE_LONG(CT, RT)                                                                                                        \
  hipLaunchKernelGGL((k_pfnt_route_out_long<CT, RT>), dim3(PL_BLOCKS), dim3(PL_THREADS), 0, s, v.pillar_start, v.pillar_cell, v.long_list,      \
                     v.counters, *grid, (const CT *)dcanvas, dpillar, arg1, (RT *)dz1)
  if (dcanvas_bf16) {
    if (dz1_bf16) PCP_PFNT_ROUTE_LONG(__bf16, __bf16); else PCP_PFNT_ROUTE_LONG(__bf16, float);
  } else {
    if (dz1_bf16) PCP_PFNT_ROUTE_LONG(float, __bf16); else PCP_PFNT_ROUTE_LONG(float, float);
  }
#undef PCP_PFNT_ROUTE_LONG
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

int pcp_pfn_train_route_out_grad(const pcp_grid_t *grid, const void *vox_workspace, int64_t n, int64_t kept_rows, const float *dcanvas,
                                 const float *dpillar, const int32_t *arg1, float *dz1, void *stream) {
  return pfn_route_out_impl(grid, vox_workspace, n, kept_rows, dcanvas, 0, dpillar, arg1, dz1, 0, stream);
}

int pcp_mp_pfn_train_route_out_grad(const pcp_grid_t *grid, const void *vox_workspace, int64_t n, int64_t kept_rows, const void *dcanvas,
                                    int32_t dcanvas_dtype, const float *dpillar, const int32_t *arg1, void *dz1, int32_t dz1_dtype, void *stream) {
  if ((dcanvas_dtype != 0 && dcanvas_dtype != 1) || (dz1_dtype != 0 && dz1_dtype != 1)) return PCP_ERR_ARG;
  return pfn_route_out_impl(grid, vox_workspace, n, kept_rows, dcanvas, dcanvas_dtype, dpillar, arg1, dz1, dz1_dtype, stream);
}

static int pfn_route_mid_impl(const pcp_grid_t *grid, const void *vox_workspace, int64_t n, const void *din1, int din1_bf16, const int32_t *arg0,
                              float *da0, void *stream) {
  if (!grid || !vox_workspace || !din1 || !arg0 || !da0 || n < 0) return PCP_ERR_ARG;
  if (n == 0) return PCP_OK;
  const WsView v = view_ws(vox_workspace, grid, n);
  const int64_t cells = (int64_t)grid->batch_size * grid->nx * grid->ny;
  const dim3 gridd(pillar_blocks(n < cells ? n : cells, 8));
  if (din1_bf16)
    hipLaunchKernelGGL(k_pfnt_route_mid<__bf16>, gridd, dim3(PT_THREADS), 0, (hipStream_t)stream, v.pillar_start, v.counters, (const __bf16 *)din1,
                       arg0, da0);
  else
    hipLaunchKernelGGL(k_pfnt_route_mid<float>, gridd, dim3(PT_THREADS), 0, (hipStream_t)stream, v.pillar_start, v.counters, (const float *)din1,
                       arg0, da0);
  PCP_CHECK_LAUNCH();
  if (din1_bf16)
    hipLaunchKernelGGL(k_pfnt_route_mid_long<__bf16>, dim3(PL_BLOCKS), dim3(PL_THREADS), 0, (hipStream_t)stream, v.pillar_start, v.long_list,
                       v.counters, (const __bf16 *)din1, arg0, da0);
  else
    hipLaunchKernelGGL(k_pfnt_route_mid_long<float>, dim3(PL_BLOCKS), dim3(PL_THREADS), 0, (hipStream_t)stream, v.pillar_start, v.long_list,
                       v.counters, (const float *)din1, arg0, da0);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

int pcp_pfn_train_route_mid_grad(const pcp_grid_t *grid, const void *vox_workspace, int64_t n, const float *din1, const int32_t *arg0,
                                 float *da0, void *stream) {
  return pfn_route_mid_impl(grid, vox_workspace, n, din1, 0, arg0, da0, stream);
}

int pcp_mp_pfn_train_route_mid_grad(const pcp_grid_t *grid, const void *vox_workspace, int64_t n, const void *din1, int32_t din1_dtype,
                                    const int32_t *arg0, float *da0, void *stream) {
  if (din1_dtype != 0 && din1_dtype != 1) return PCP_ERR_ARG;
  return pfn_route_mid_impl(grid, vox_workspace, n, din1, din1_dtype, arg0, da0, stream);
}

int pcp_pfn_features(const float *points, int64_t n, int32_t row_stride, int32_t num_raw, uint32_t flags, const pcp_grid_t *grid,
                     const void *vox_workspace, int32_t fw, float *fbuf, int32_t *slot_pillar, void *stream) {
  if (!points || !grid || !vox_workspace || !fbuf || !slot_pillar || n < 0 || row_stride < 1 + num_raw || num_raw < 3) return PCP_ERR_ARG;
  if (flags & ~(PCP_PFN_ABSOLUTE_XYZ | PCP_PFN_WITH_DISTANCE)) return PCP_ERR_ARG;
  const int use_abs = (flags & PCP_PFN_ABSOLUTE_XYZ) ? 1 : 0, with_dist = (flags & PCP_PFN_WITH_DISTANCE) ? 1 : 0;
  if (fw < num_raw - (use_abs ? 0 : 3) + 6 + with_dist) return PCP_ERR_ARG;
  if (n == 0) return PCP_OK;
  const WsView v = view_ws(vox_workspace, grid, n);
  FeatAnyParams q;
  q.b.points = points; q.b.stride = row_stride; q.b.g = *grid;
  q.b.bucket_order = v.bucket_order; q.b.pillar_cell = v.pillar_cell; q.b.pillar_start = v.pillar_start; q.b.counters = v.counters;
  q.b.fbuf = fbuf; q.b.slot_pillar = slot_pillar;
  q.num_raw = num_raw; q.use_abs = use_abs; q.with_dist = with_dist; q.fw = fw;
  const int64_t cells = (int64_t)grid->batch_size * grid->nx * grid->ny;
  const int64_t max_pillars = n < cells ? n : cells;
  const int blocks = (int)((max_pillars + PT_PILLARS - 1) / PT_PILLARS);
  hipLaunchKernelGGL(k_pfn_feat_any, dim3(blocks), dim3(PT_THREADS), 0, (hipStream_t)stream, q);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

int pcp_pfn_cat_pillar_max(const float *y, int32_t ld_y, const float *pillar_max, int32_t ld_max, const int32_t *slot_pillar, int64_t rows,
                           int32_t c, float *out, int32_t ld_out, void *stream) {
  if (!y || !pillar_max || !slot_pillar || !out || rows < 0 || c < 1 || ld_y < c || ld_max < c || ld_out < 2 * c) return PCP_ERR_ARG;
  if (rows == 0) return PCP_OK;
  int64_t blocks = (rows * ld_out + PT_THREADS - 1) / PT_THREADS;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_pfn_cat_pillar_max, dim3((unsigned)blocks), dim3(PT_THREADS), 0, (hipStream_t)stream, y, ld_y, pillar_max, ld_max,
                     slot_pillar, (long long)rows, c, out, ld_out);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

}  // extern "C"
