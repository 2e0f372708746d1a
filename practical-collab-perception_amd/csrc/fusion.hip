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

__global__ void k_warp_nearest(const float *__restrict__ src, float *__restrict__ dst, int h, int w, int c4, int ld_src,
                               int ld_dst, Theta th, int accumulate) {
  // one thread per (pixel, float4 of channels)
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

__global__ void k_softmax_fuse(MapPtrs maps, int n_agents, const float *__restrict__ weights, int ld_w, long long pixels,
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
      float4 m = *reinterpret_cast<const float4 *>(maps.p[a] + pix * ld_map + q * 4);
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
  hipLaunchKernelGGL(k_warp_nearest, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, src, dst, h, w,
                     c / 4, ld_src, ld_dst, th, accumulate);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

extern "C" int pcp_softmax_fuse(const float *const *maps_host, int32_t n_agents, const float *weights, int32_t ld_w,
                                int64_t pixels, int32_t c, int32_t ld_map, int32_t ld_out, float *out, void *stream_) {
  if (!maps_host || !weights || !out || n_agents <= 0 || n_agents > MAX_AGENTS || pixels <= 0 || c <= 0 || (c & 3) ||
      (ld_map & 3) || (ld_out & 3) || ld_w < n_agents)
    return PCP_ERR_ARG;
  MapPtrs mp;
  for (int a = 0; a < MAX_AGENTS; a++) mp.p[a] = a < n_agents ? maps_host[a] : nullptr;
  long long total = pixels * (c / 4);
  hipLaunchKernelGGL(k_softmax_fuse, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, mp, n_agents,
                     weights, ld_w, (long long)pixels, c / 4, ld_map, ld_out, out);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}
