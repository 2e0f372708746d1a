// a11 / a12 -- DiscoNet mid fusion helpers: nearest affine warp of an agent's BEV map into the ego frame, and the
// pixel-wise softmax-over-agents weighted sum.
//
// Replaces pcdet/models/bev_layers/v2x_fusion_disco.py:29-45 (F.affine_grid + F.grid_sample(mode='nearest'),
// align_corners=False, zero padding: two launches per (agent, frame) plus the grid tensor) and :104-115
// (cat + softmax + stack + broadcast multiply + sum: five full-map temporaries).
// Both kernels are pure HBM streams: warp = C*H*W*4 read (gathered rows of C floats, contiguous in NHWC) + write;
// fuse = (n_agents + 1) * pixels * C * 4 bytes.
#include "pcp_common.h"

#pragma clang fp contract(off)

namespace {

struct Theta { float t[6]; };

// torch.linspace(-1, 1, n)[i] * (n - 1) / n   (ATen affine_grid base grid, align_corners=False)
__device__ __forceinline__ float base_coord(int i, int n) {
  if (n <= 1) return 0.f;
  float step = 2.0f / (float)(n - 1);
  float v = (i < n / 2) ? (-1.0f + step * (float)i) : (1.0f - step * (float)(n - 1 - i));
  return v * (float)(n - 1) / (float)n;
}

// storage-typed four-channel access (float | __bf16): the bf16 training loop keeps the fusion module's stacked maps as bf16
template <typename T> struct Map4;
template <> struct Map4<float> {
  static __device__ __forceinline__ float4 ld(const float *p) { return *reinterpret_cast<const float4 *>(p); }
  static __device__ __forceinline__ void st(float *p, const float4 &v) { *reinterpret_cast<float4 *>(p) = v; }
};
template <> struct Map4<__bf16> {
  typedef __bf16 b4 __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ float4 ld(const __bf16 *p) {
    const b4 v = *reinterpret_cast<const b4 *>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
  }
  static __device__ __forceinline__ void st(__bf16 *p, const float4 &v) {
    b4 o;
    o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
    *reinterpret_cast<b4 *>(p) = o;
  }
};

template <typename ST, typename DT>
__global__ void k_warp_nearest(const ST *__restrict__ src, DT *__restrict__ dst, int h, int w, int c4, int ld_src,
                               int ld_dst, Theta th, int accumulate) {
  // one thread per (pixel, four channels)
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)h * w * c4;
  if (t >= total) return;
  int q = (int)(t % c4);
  int pix = (int)(t / c4);
  int oy = pix / w, ox = pix % w;
  float xn = base_coord(ox, w), yn = base_coord(oy, h);
  float gx = xn * th.t[0] + yn * th.t[1] + th.t[2];
  float gy = xn * th.t[3] + yn * th.t[4] + th.t[5];
  // grid_sample unnormalise (ATen CPU kernel form): (g + 1) * (size / 2) - 0.5, then round half to even
  float fx = (gx + 1.0f) * ((float)w / 2.0f) - 0.5f;
  float fy = (gy + 1.0f) * ((float)h / 2.0f) - 0.5f;
  float rx = nearbyintf(fx), ry = nearbyintf(fy);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (rx >= 0.f && rx <= (float)(w - 1) && ry >= 0.f && ry <= (float)(h - 1)) {
    int sx = (int)rx, sy = (int)ry;
    v = Map4<ST>::ld(src + ((long long)sy * w + sx) * ld_src + q * 4);
  }
  DT *o = dst + (long long)pix * ld_dst + q * 4;
  if (accumulate) {
    float4 cur = Map4<DT>::ld(o);
    v.x += cur.x; v.y += cur.y; v.z += cur.z; v.w += cur.w;
  }
  Map4<DT>::st(o, v);
}

// every (agent, frame) warp of a DiscoNet forward in ONE launch: blockIdx.y = job; the per-pixel arithmetic is k_warp_nearest's
constexpr int WARP_MAX_JOBS = 32;
struct WarpJobs {
  const float *src[WARP_MAX_JOBS];
  float *dst[WARP_MAX_JOBS];
  Theta th[WARP_MAX_JOBS];
};

// theta_dev (nullable): the 2 x 3 affines of the jobs in DEVICE memory (6 floats per job, job-major, entry `theta_first + blockIdx.y`) instead of the
// kernel argument -- a captured hipGraph then serves every pose set: the host refreshes the table before it replays (pcp_warp_nearest_batch_dev)
__global__ void k_warp_nearest_batch(WarpJobs jobs, int h, int w, int c4, int ld_src, int ld_dst, int accumulate, const float *__restrict__ theta_dev,
                                     int theta_first) {
  const int j = blockIdx.y;
  const float *__restrict__ src = jobs.src[j];
  float *__restrict__ dst = jobs.dst[j];
  Theta th = jobs.th[j];
  if (theta_dev) {
#pragma unroll
    for (int i = 0; i < 6; i++) th.t[i] = theta_dev[(theta_first + j) * 6 + i];
  }
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)h * w * c4;
  if (t >= total) return;
  int q = (int)(t % c4);
  int pix = (int)(t / c4);
  int oy = pix / w, ox = pix % w;
  float xn = base_coord(ox, w), yn = base_coord(oy, h);
  float gx = xn * th.t[0] + yn * th.t[1] + th.t[2];
  float gy = xn * th.t[3] + yn * th.t[4] + th.t[5];
  float fx = (gx + 1.0f) * ((float)w / 2.0f) - 0.5f;
  float fy = (gy + 1.0f) * ((float)h / 2.0f) - 0.5f;
  float rx = nearbyintf(fx), ry = nearbyintf(fy);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (rx >= 0.f && rx <= (float)(w - 1) && ry >= 0.f && ry <= (float)(h - 1)) {
    int sx = (int)rx, sy = (int)ry;
    v = *reinterpret_cast<const float4 *>(src + ((long long)sy * w + sx) * ld_src + q * 4);
  }
  float4 *o = reinterpret_cast<float4 *>(dst + (long long)pix * ld_dst + q * 4);
  if (accumulate) {
    float4 cur = *o;
    v.x += cur.x; v.y += cur.y; v.z += cur.z; v.w += cur.w;
  }
  *o = v;
}

constexpr int MAX_AGENTS = 16;
struct MapPtrs { const float *p[MAX_AGENTS]; };
struct VMapPtrs { const void *p[MAX_AGENTS]; };

template <typename MT>
__global__ void k_softmax_fuse(VMapPtrs maps, int n_agents, const float *__restrict__ weights, int ld_w, long long pixels,
                               int c4, int ld_map, int ld_out, float *__restrict__ out) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= pixels * c4) return;
  int q = (int)(t % c4);
  long long pix = t / c4;
  float wv[MAX_AGENTS];
  float mx = -INFINITY;
#pragma unroll
  for (int a = 0; a < MAX_AGENTS; a++)
    if (a < n_agents) {
      wv[a] = weights[pix * ld_w + a];
      mx = fmaxf(mx, wv[a]);
    }
  float den = 0.f;
#pragma unroll
  for (int a = 0; a < MAX_AGENTS; a++)
    if (a < n_agents) {
      wv[a] = expf(wv[a] - mx);
      den += wv[a];
    }
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int a = 0; a < MAX_AGENTS; a++)
    if (a < n_agents) {
      float s = wv[a] / den;
      float4 m = Map4<MT>::ld(reinterpret_cast<const MT *>(maps.p[a]) + pix * ld_map + q * 4);
      acc.x += m.x * s; acc.y += m.y * s; acc.z += m.z * s; acc.w += m.w * s;
    }
  *reinterpret_cast<float4 *>(out + pix * ld_out + q * 4) = acc;
}

}  // namespace

extern "C" int pcp_warp_nearest(const float *src, float *dst, int32_t h, int32_t w, int32_t c, int32_t ld_src, int32_t ld_dst,
                                const float *theta_host, int32_t accumulate, void *stream_) {
  if (!src || !dst || !theta_host || h <= 0 || w <= 0 || c <= 0 || (c & 3) || (ld_src & 3) || (ld_dst & 3)) return PCP_ERR_ARG;
  if (src == dst) return PCP_ERR_ARG;
  Theta th;
  for (int i = 0; i < 6; i++) th.t[i] = theta_host[i];
  long long total = (long long)h * w * (c / 4);
  hipLaunchKernelGGL((k_warp_nearest<float, float>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, src, dst, h, w,
                     c / 4, ld_src, ld_dst, th, accumulate);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

static int warp_batch_impl(const float *const *src_host, float *const *dst_host, const float *theta_host, const float *theta_dev, int32_t n_jobs,
                           int32_t h, int32_t w, int32_t c, int32_t ld_src, int32_t ld_dst, int32_t accumulate, void *stream_) {
  if (!src_host || !dst_host || (!theta_host && !theta_dev) || n_jobs <= 0 || h <= 0 || w <= 0 || c <= 0 || (c & 3) || (ld_src & 3) || (ld_dst & 3))
    return PCP_ERR_ARG;
  const long long total = (long long)h * w * (c / 4);
  for (int j0 = 0; j0 < n_jobs; j0 += WARP_MAX_JOBS) {
    const int n = n_jobs - j0 < WARP_MAX_JOBS ? n_jobs - j0 : WARP_MAX_JOBS;
    WarpJobs jobs{};
    for (int j = 0; j < n; j++) {
      if (!src_host[j0 + j] || !dst_host[j0 + j] || src_host[j0 + j] == dst_host[j0 + j]) return PCP_ERR_ARG;
      jobs.src[j] = src_host[j0 + j];
      jobs.dst[j] = dst_host[j0 + j];
      for (int i = 0; i < 6; i++) jobs.th[j].t[i] = theta_host ? theta_host[(j0 + j) * 6 + i] : 0.f;
    }
    hipLaunchKernelGGL(k_warp_nearest_batch, dim3((unsigned)((total + 255) / 256), (unsigned)n), dim3(256), 0, (hipStream_t)stream_, jobs, h, w,
                       c / 4, ld_src, ld_dst, accumulate, theta_dev, j0);
    PCP_CHECK_LAUNCH();
  }
  return PCP_OK;
}

extern "C" int pcp_warp_nearest_batch(const float *const *src_host, float *const *dst_host, const float *theta_host, int32_t n_jobs,
                                      int32_t h, int32_t w, int32_t c, int32_t ld_src, int32_t ld_dst, int32_t accumulate, void *stream_) {
  if (!theta_host) return PCP_ERR_ARG;
  return warp_batch_impl(src_host, dst_host, theta_host, nullptr, n_jobs, h, w, c, ld_src, ld_dst, accumulate, stream_);
}

extern "C" int pcp_warp_nearest_batch_dev(const float *const *src_host, float *const *dst_host, const float *theta_dev, int32_t n_jobs,
                                          int32_t h, int32_t w, int32_t c, int32_t ld_src, int32_t ld_dst, int32_t accumulate, void *stream_) {
  if (!theta_dev) return PCP_ERR_ARG;
  return warp_batch_impl(src_host, dst_host, nullptr, theta_dev, n_jobs, h, w, c, ld_src, ld_dst, accumulate, stream_);
}

extern "C" int pcp_softmax_fuse(const float *const *maps_host, int32_t n_agents, const float *weights, int32_t ld_w,
                                int64_t pixels, int32_t c, int32_t ld_map, int32_t ld_out, float *out, void *stream_) {
  if (!maps_host || !weights || !out || n_agents <= 0 || n_agents > MAX_AGENTS || pixels <= 0 || c <= 0 || (c & 3) ||
      (ld_map & 3) || (ld_out & 3) || ld_w < n_agents)
    return PCP_ERR_ARG;
  VMapPtrs mp;
  for (int a = 0; a < MAX_AGENTS; a++) mp.p[a] = a < n_agents ? maps_host[a] : nullptr;
  long long total = pixels * (c / 4);
  hipLaunchKernelGGL((k_softmax_fuse<float>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, mp, n_agents,
                     weights, ld_w, (long long)pixels, c / 4, ld_map, ld_out, out);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

// storage-typed forms for the bf16 training loop (include/pcp_hip_mp.h)
extern "C" int pcp_mp_warp_nearest(const void *src, int32_t src_dtype, void *dst, int32_t dst_dtype, int32_t h, int32_t w, int32_t c,
                                   int32_t ld_src, int32_t ld_dst, const float *theta_host, int32_t accumulate, void *stream_) {
  if (!src || !dst || !theta_host || h <= 0 || w <= 0 || c <= 0 || (c & 3) || (ld_src & 3) || (ld_dst & 3)) return PCP_ERR_ARG;
  if (src == dst || (src_dtype | dst_dtype) & ~1) return PCP_ERR_ARG;
  Theta th;
  for (int i = 0; i < 6; i++) th.t[i] = theta_host[i];
  const long long total = (long long)h * w * (c / 4);
  const dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t st = (hipStream_t)stream_;
#define PCP_WARP(ST, DT) hipLaunchKernelGGL((k_warp_nearest<ST, DT>), grid, dim3(256), 0, st, (const ST *)src, (DT *)dst, h, w, c / 4, ld_src, ld_dst, th, accumulate)
  if (src_dtype == 1) { if (dst_dtype == 1) PCP_WARP(__bf16, __bf16); else PCP_WARP(__bf16, float); }
  else { if (dst_dtype == 1) PCP_WARP(float, __bf16); else PCP_WARP(float, float); }
#undef PCP_WARP
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

extern "C" int pcp_mp_softmax_fuse(const void *const *maps_host, int32_t map_dtype, int32_t n_agents, const float *weights, int32_t ld_w,
                                   int64_t pixels, int32_t c, int32_t ld_map, int32_t ld_out, float *out, void *stream_) {
  if (!maps_host || !weights || !out || n_agents <= 0 || n_agents > MAX_AGENTS || pixels <= 0 || c <= 0 || (c & 3) ||
      (ld_map & 3) || (ld_out & 3) || ld_w < n_agents || (map_dtype & ~1))
    return PCP_ERR_ARG;
  VMapPtrs mp;
  for (int a = 0; a < MAX_AGENTS; a++) mp.p[a] = a < n_agents ? maps_host[a] : nullptr;
  const long long total = pixels * (c / 4);
  const dim3 grid((unsigned)((total + 255) / 256));
  if (map_dtype == 1)
    hipLaunchKernelGGL((k_softmax_fuse<__bf16>), grid, dim3(256), 0, (hipStream_t)stream_, mp, n_agents, weights, ld_w, (long long)pixels, c / 4,
                       ld_map, ld_out, out);
  else
    hipLaunchKernelGGL((k_softmax_fuse<float>), grid, dim3(256), 0, (hipStream_t)stream_, mp, n_agents, weights, ld_w, (long long)pixels, c / 4,
                       ld_map, ld_out, out);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}


// ---- a12 (round 3): the whole pixel weightor + softmax over agents + weighted sum in ONE launch -----------------------------------------
// Reference: pcdet/models/bev_layers/v2x_fusion_disco.py:8-26 (PixelWeightedFusionSoftmax: Conv1x1 2C -> 64 + BN + ReLU, Conv1x1 64 -> 16 +
// BN + ReLU, Conv1x1 16 -> 1 + ReLU on cat[ego, map_a]) applied once per map (:85,:104) and :107-115 (softmax over the maps, weighted sum).
// Round 2 ran it as 3 pointwise launches per map + k_softmax_fuse: 19 launches and two (pixels, 64) / (pixels, 16) round trips per map.
// Here a workgroup owns 64 pixels: the ego half of the first layer (W1[:, :C] . ego) is computed ONCE per tile and reused for every map
// (7 instead of 12 K = 128 products for six maps), the map half runs on the matrix pipe with the W1 fragments resident in registers, the
// two small layers and the softmax stay on the VALU through LDS, and the weighted sum re-reads the map rows (L2-hot) exactly as
// k_softmax_fuse does (same operation order: acc += m * s, a = 0 .. n-1).
namespace {

constexpr int WF_PX = 64;
constexpr int WF_C = 128;            // compressed channels (COMPRESSED_CHANNELS of the shipped YAML)
constexpr int WF_H1 = 64, WF_H2 = 16;
constexpr int WF_H1_LD = 68;         // padded h1 row: conflict-free ds_read_b128 groups

struct WfParams {
  MapPtrs maps;
  int n_maps, ld_map, ld_out, ld_w;
  long long pixels;
  const float *w1, *b1, *w2, *b2, *w3, *b3;     // BN folded: w1 (64, 2C) row-major, w2 (16, 64), w3 (16)
  float *out, *logits;                          // logits: optional (pixels, ld_w) copy of the pre-softmax weights
  const int *live;                              // optional device flags: map a takes part in the softmax iff live_idx[a] < 0 or live[live_idx[a]] != 0
  int live_idx[MAX_AGENTS];
};

}  // namespace

#pragma clang fp contract(on)
namespace {
__device__ __forceinline__ f32x16 wf_mfma(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
}

namespace {

__global__ __launch_bounds__(256, 2) void k_weight_fuse(WfParams p) {
  __shared__ __attribute__((aligned(16))) float h1[WF_PX * WF_H1_LD];
  __shared__ float h2[WF_PX][WF_H2 + 1];
  __shared__ float lg[MAX_AGENTS][WF_PX];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rt = wave >> 1, ct = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  // W1 fragments of this wave's 32 hidden channels, k permuted as the 16-byte A loads deliver it: MFMA (j, i) multiplies k = 8j + 4h + i
  // (the map half stays in registers for every map of every tile; the ego half is used once per tile and re-read from L1 / L2)
  f32x4 wb[16];
  const float *w1_row = p.w1 + (long long)(ct * 32 + r) * (2 * WF_C) + 4 * h;
#pragma unroll
  for (int j = 0; j < 16; j++) wb[j] = *reinterpret_cast<const f32x4 *>(w1_row + WF_C + 8 * j);
  const float bias1 = p.b1[ct * 32 + r];
  const long long n_tiles = (p.pixels + WF_PX - 1) / WF_PX;
  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long long px0 = tile * WF_PX;
    long long prow = px0 + rt * 32 + r;                       // the pixel whose row this lane feeds into the matrix pipe
    if (prow >= p.pixels) prow = p.pixels - 1;                // clamped: rows past the end are computed and never stored
    f32x16 acc_e;
#pragma unroll
    for (int e = 0; e < 16; e++) acc_e[e] = 0.f;
    {
      const float *src = p.maps.p[0] + prow * p.ld_map + 4 * h;
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(src + 8 * j);
        const f32x4 w = *reinterpret_cast<const f32x4 *>(w1_row + 8 * j);
        acc_e = wf_mfma(a.x, w.x, acc_e);
        acc_e = wf_mfma(a.y, w.y, acc_e);
        acc_e = wf_mfma(a.z, w.z, acc_e);
        acc_e = wf_mfma(a.w, w.w, acc_e);
      }
    }
    for (int a_i = 0; a_i < p.n_maps; a_i++) {
      f32x16 acc = acc_e;
      const float *src = p.maps.p[a_i] + prow * p.ld_map + 4 * h;
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(src + 8 * j);
        acc = wf_mfma(a.x, wb[j].x, acc);
        acc = wf_mfma(a.y, wb[j].y, acc);
        acc = wf_mfma(a.z, wb[j].z, acc);
        acc = wf_mfma(a.w, wb[j].w, acc);
      }
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int row = rt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        h1[row * WF_H1_LD + ct * 32 + r] = fmaxf(acc[e] + bias1, 0.f);
      }
      __syncthreads();
      {   // layer 2 (64 -> 16): thread = (pixel, 4 outputs); the wave's four W2 rows are wave-uniform (scalar loads)
        const int px = lane, og = wave;
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = p.b2[og * 4 + i];
        const float *xr = &h1[px * WF_H1_LD];
#pragma unroll
        for (int k4 = 0; k4 < WF_H1 / 4; k4++) {
          const f32x4 x = *reinterpret_cast<const f32x4 *>(xr + 4 * k4);
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const float *wr = p.w2 + (og * 4 + i) * WF_H1 + 4 * k4;
            o[i] = fmaf(wr[0], x.x, o[i]);
            o[i] = fmaf(wr[1], x.y, o[i]);
            o[i] = fmaf(wr[2], x.z, o[i]);
            o[i] = fmaf(wr[3], x.w, o[i]);
          }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) h2[px][og * 4 + i] = fmaxf(o[i], 0.f);
      }
      __syncthreads();
      if (tid < WF_PX) {   // layer 3 (16 -> 1) + ReLU
        float l = p.b3[0];
#pragma unroll
        for (int k = 0; k < WF_H2; k++) l = fmaf(p.w3[k], h2[tid][k], l);
        l = fmaxf(l, 0.f);
        // a map that does not exist (hipGraph mode: an agent without rows, decided on the device) leaves the softmax: weight exactly 0,
        // the other weights exactly those of the softmax over the maps that do exist
        if (p.live && p.live_idx[a_i] >= 0 && p.live[p.live_idx[a_i]] == 0) l = -INFINITY;
        lg[a_i][tid] = l;
        if (p.logits && px0 + tid < p.pixels) p.logits[(px0 + tid) * p.ld_w + a_i] = l;
      }
      // the next map's h1 stores come after its 64 MFMAs and are ordered behind this map's layer-2 reads by the barrier above; its h2
      // stores are ordered behind this layer-3 read by the barrier that follows its h1 stores
    }
    __syncthreads();
    if (tid < WF_PX) {     // softmax over the maps, as k_softmax_fuse: max, expf, sum, divide
      float mx = -INFINITY;
      for (int a_i = 0; a_i < p.n_maps; a_i++) mx = fmaxf(mx, lg[a_i][tid]);
      float den = 0.f;
      for (int a_i = 0; a_i < p.n_maps; a_i++) {
        const float e = expf(lg[a_i][tid] - mx);
        lg[a_i][tid] = e;
        den += e;
      }
      for (int a_i = 0; a_i < p.n_maps; a_i++) lg[a_i][tid] = lg[a_i][tid] / den;
    }
    __syncthreads();
    {
#pragma clang fp contract(off)
      // maps outermost: the eight 16-byte loads of one map are independent and in flight together (with the maps innermost every load sat
      // behind the previous one's use: 48 dependent L2 round trips per tile, ~200 k cycles); per output the sum still runs a = 0 .. n-1
      constexpr int WF_Q = WF_PX * (WF_C / 4) / 256;
      float4 facc[WF_Q];
#pragma unroll
      for (int i = 0; i < WF_Q; i++) facc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int a_i = 0; a_i < p.n_maps; a_i++) {
        float4 m[WF_Q];
        float sv[WF_Q];
#pragma unroll
        for (int i = 0; i < WF_Q; i++) {
          const int idx = i * 256 + tid;
          const int px = idx >> 5, q = idx & 31;
          long long pp = px0 + px;
          if (pp >= p.pixels) pp = p.pixels - 1;
          m[i] = *reinterpret_cast<const float4 *>(p.maps.p[a_i] + pp * p.ld_map + q * 4);
          sv[i] = lg[a_i][px];
        }
#pragma unroll
        for (int i = 0; i < WF_Q; i++) {
          facc[i].x += m[i].x * sv[i]; facc[i].y += m[i].y * sv[i]; facc[i].z += m[i].z * sv[i]; facc[i].w += m[i].w * sv[i];
        }
      }
#pragma unroll
      for (int i = 0; i < WF_Q; i++) {
        const int idx = i * 256 + tid;
        const int px = idx >> 5, q = idx & 31;
        if (px0 + px < p.pixels) *reinterpret_cast<float4 *>(p.out + (px0 + px) * p.ld_out + q * 4) = facc[i];
      }
    }
    __syncthreads();          // lg / h1 / h2 are reused by the next tile
  }
}

}  // namespace

static int disco_weight_fuse_impl(const float *const *maps_host, int32_t n_maps, int32_t c, int32_t ld_map, int64_t pixels,
                                  const float *w1, const float *b1, const float *w2, const float *b2, const float *w3, const float *b3,
                                  float *out, int32_t ld_out, float *logits, int32_t ld_w, const int32_t *live_index_host, const int32_t *live,
                                  void *stream_) {
  if (!maps_host || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !out || n_maps <= 0 || n_maps > MAX_AGENTS || pixels <= 0) return PCP_ERR_ARG;
  if (c != WF_C) return PCP_ERR_UNSUPPORTED;
  if ((ld_map & 3) || (ld_out & 3) || ld_map < c || ld_out < c || (logits && ld_w < n_maps)) return PCP_ERR_ARG;
  if ((((uintptr_t)w1) & 15) || (((uintptr_t)out) & 15)) return PCP_ERR_ARG;
  WfParams p;
  for (int a = 0; a < MAX_AGENTS; a++) {
    p.maps.p[a] = a < n_maps ? maps_host[a] : nullptr;
    if (a < n_maps && (!maps_host[a] || (((uintptr_t)maps_host[a]) & 15))) return PCP_ERR_ARG;
  }
  p.n_maps = n_maps; p.ld_map = ld_map; p.ld_out = ld_out; p.ld_w = ld_w; p.pixels = pixels;
  p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.w3 = w3; p.b3 = b3; p.out = out; p.logits = logits;
  p.live = live_index_host ? live : nullptr;
  for (int a = 0; a < MAX_AGENTS; a++) p.live_idx[a] = (live_index_host && a < n_maps) ? live_index_host[a] : -1;
  const long long n_tiles = (pixels + WF_PX - 1) / WF_PX;
  const unsigned blocks = (unsigned)(n_tiles < 512 ? n_tiles : 512);
  hipLaunchKernelGGL(k_weight_fuse, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, p);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

extern "C" int pcp_disco_weight_fuse(const float *const *maps_host, int32_t n_maps, int32_t c, int32_t ld_map, int64_t pixels,
                                     const float *w1, const float *b1, const float *w2, const float *b2, const float *w3, const float *b3,
                                     float *out, int32_t ld_out, float *logits, int32_t ld_w, void *stream_) {
  return disco_weight_fuse_impl(maps_host, n_maps, c, ld_map, pixels, w1, b1, w2, b2, w3, b3, out, ld_out, logits, ld_w, nullptr, nullptr, stream_);
}

extern "C" int pcp_disco_weight_fuse_live(const float *const *maps_host, int32_t n_maps, int32_t c, int32_t ld_map, int64_t pixels,
                                          const float *w1, const float *b1, const float *w2, const float *b2, const float *w3, const float *b3,
                                          float *out, int32_t ld_out, const int32_t *live_index_host, const int32_t *live, void *stream_) {
  if (!live_index_host || !live) return PCP_ERR_ARG;
  return disco_weight_fuse_impl(maps_host, n_maps, c, ld_map, pixels, w1, b1, w2, b2, w3, b3, out, ld_out, nullptr, 0, live_index_host, live, stream_);
}
