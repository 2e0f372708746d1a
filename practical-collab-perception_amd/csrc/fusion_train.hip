// DiscoNet mid fusion, training half: the last stage of the pixel weightor (Conv1x1 16 -> 1 + ReLU), and the backward of
// softmax-over-agents + weighted sum + that stage.
//
// Replaces the autograd graph of pcdet/models/bev_layers/v2x_fusion_disco.py:22-24 (relu(conv1_4(x))) and :109-115
// (cat + softmax + stack + broadcast multiply + sum).  Only the ego map carries a gradient: the agent maps come out of
// transform_bev_img, which is @torch.no_grad (:29), so dL/d(agent map) is never formed.
// One wavefront per pixel: the C = 128 channel dot products <dfused, map_a> are wave reductions, the per-agent scalars live in
// registers, conv1_4's 17 parameter gradients are accumulated per lane and reduced once per block in float64.
#include "pcp_common.h"

namespace {

constexpr int FT_MAX_AGENTS = 8;
constexpr int FT_MAXV = 4;          // channels per lane: c <= 256
constexpr int H2 = 16;              // conv1_4 input channels

struct Ptrs { const float *p[FT_MAX_AGENTS]; };
struct MutPtrs { float *p[FT_MAX_AGENTS]; };

__global__ __launch_bounds__(256) void k_weight_logits(Ptrs h2, int n_agents, int ld_h, const float *__restrict__ w4,
                                                      const float *__restrict__ b4, long long pixels, float *__restrict__ logits, int ld_w) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= pixels * n_agents) return;
  const int a = (int)(t % n_agents);
  const long long pix = t / n_agents;
  const float *row = h2.p[a] + pix * ld_h;
  float acc = b4[0];
#pragma unroll
  for (int j = 0; j < H2; ++j) acc = fmaf(row[j], w4[j], acc);
  logits[pix * ld_w + a] = fmaxf(acc, 0.f);
}

__device__ __forceinline__ float wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

struct VPtrs { const void *p[FT_MAX_AGENTS]; };

// MT: storage type of the stacked maps (float; __bf16 in the bf16 training loop)
template <typename MT>
__global__ __launch_bounds__(256) void k_fuse_backward(VPtrs maps, int n_agents, int ld_map, int c, const float *__restrict__ logits,
                                                      int ld_w, const float *__restrict__ dfused, int ld_df, Ptrs h2, int ld_h,
                                                      const float *__restrict__ w4, long long pixels, float *__restrict__ dmap0, int ld_dm,
                                                      MutPtrs dh2, double *acc) {
  __shared__ double red[4][H2 + 1];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nw = ((long long)gridDim.x * blockDim.x) >> 6;
  const float w4l = lane < H2 ? w4[lane] : 0.f;
  double accw = 0.0;                 // lanes 0..15: dw4[lane]; lane 16: db4
  for (long long pix = wave0; pix < pixels; pix += nw) {
    float df[FT_MAXV];
#pragma unroll
    for (int i = 0; i < FT_MAXV; ++i) {
      const int ch = lane + 64 * i;
      df[i] = ch < c ? dfused[pix * ld_df + ch] : 0.f;
    }
    float dot[FT_MAX_AGENTS], lg[FT_MAX_AGENTS];
    float mx = -INFINITY;
#pragma unroll
    for (int a = 0; a < FT_MAX_AGENTS; ++a) {
      dot[a] = 0.f;
      lg[a] = -INFINITY;
      if (a < n_agents) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < FT_MAXV; ++i) {
          const int ch = lane + 64 * i;
          if (ch < c) s = fmaf(df[i], (float)reinterpret_cast<const MT *>(maps.p[a])[pix * ld_map + ch], s);
        }
        dot[a] = wave_sum(s);
        lg[a] = logits[pix * ld_w + a];
        mx = fmaxf(mx, lg[a]);
      }
    }
    float wsum = 0.f, wgt[FT_MAX_AGENTS];
#pragma unroll
    for (int a = 0; a < FT_MAX_AGENTS; ++a) {
      wgt[a] = a < n_agents ? expf(lg[a] - mx) : 0.f;
      wsum += wgt[a];
    }
    float sdot = 0.f;
#pragma unroll
    for (int a = 0; a < FT_MAX_AGENTS; ++a) {
      wgt[a] /= wsum;
      sdot = fmaf(wgt[a], dot[a], sdot);
    }
#pragma unroll
    for (int i = 0; i < FT_MAXV; ++i) {
      const int ch = lane + 64 * i;
      if (ch < c) dmap0[pix * ld_dm + ch] = wgt[0] * df[i];
    }
    float dbsum = 0.f;
#pragma unroll
    for (int a = 0; a < FT_MAX_AGENTS; ++a) {
      if (a < n_agents) {
        const float dl = lg[a] > 0.f ? wgt[a] * (dot[a] - sdot) : 0.f;      // softmax backward, then ReLU of conv1_4
        dbsum += dl;
        if (lane < H2) {
          dh2.p[a][pix * ld_h + lane] = dl * w4l;
          accw += (double)(dl * h2.p[a][pix * ld_h + lane]);
        }
      }
    }
    if (lane == H2) accw += (double)dbsum;
  }
  if (lane <= H2) red[wv][lane] = accw;
  __syncthreads();
  if (threadIdx.x <= H2) {
    double s = 0;
    for (int w = 0; w < 4; ++w) s += red[w][threadIdx.x];
    atomicAdd(acc + threadIdx.x, s);
  }
}

__global__ void k_w4_finalize(const double *acc, float *dw4, float *db4, int accumulate) {
  const int i = threadIdx.x;
  if (i < H2) dw4[i] = accumulate ? dw4[i] + (float)acc[i] : (float)acc[i];
  if (i == H2) db4[0] = accumulate ? db4[0] + (float)acc[H2] : (float)acc[H2];
}

}  // namespace

extern "C" {

int pcp_disco_weight_logits(const float *const *h2_host, int32_t n_agents, int32_t ld_h, const float *w4, const float *b4,
                            int64_t pixels, float *logits, int32_t ld_w, void *stream) {
  if (!h2_host || !w4 || !b4 || !logits || n_agents <= 0 || n_agents > FT_MAX_AGENTS || pixels <= 0 || ld_w < n_agents || ld_h < H2)
    return PCP_ERR_ARG;
  Ptrs h;
  for (int a = 0; a < FT_MAX_AGENTS; ++a) h.p[a] = a < n_agents ? h2_host[a] : nullptr;
  const long long total = (long long)pixels * n_agents;
  hipLaunchKernelGGL(k_weight_logits, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h, n_agents, ld_h, w4, b4,
                     (long long)pixels, logits, ld_w);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

size_t pcp_disco_fuse_backward_workspace_bytes(void) { return 32 * sizeof(double); }

static int fuse_backward_impl(const void *const *maps_host, int map_bf16, int32_t n_agents, int32_t ld_map, int32_t c, const float *logits,
                              int32_t ld_w, const float *dfused, int32_t ld_df, const float *const *h2_host, int32_t ld_h, const float *w4,
                              int64_t pixels, float *dmap0, int32_t ld_dm, float *const *dh2_host, void *workspace, float *dw4, float *db4,
                              int32_t accumulate, void *stream) {
  if (!maps_host || !logits || !dfused || !h2_host || !w4 || !dmap0 || !dh2_host || !workspace || !dw4 || !db4) return PCP_ERR_ARG;
  if (n_agents <= 0 || n_agents > FT_MAX_AGENTS || c <= 0 || c > 64 * FT_MAXV || pixels <= 0) return PCP_ERR_ARG;
  VPtrs m;
  Ptrs h;
  MutPtrs dh;
  for (int a = 0; a < FT_MAX_AGENTS; ++a) {
    m.p[a] = a < n_agents ? maps_host[a] : nullptr;
    h.p[a] = a < n_agents ? h2_host[a] : nullptr;
    dh.p[a] = a < n_agents ? dh2_host[a] : nullptr;
  }
  hipStream_t s = (hipStream_t)stream;
  double *acc = (double *)workspace;
  if (pcp_zero_async(acc, 32 * sizeof(double), s) != PCP_OK) return PCP_ERR_LAUNCH;
  long long blocks = (pixels + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  if (map_bf16)
    hipLaunchKernelGGL((k_fuse_backward<__bf16>), dim3((unsigned)blocks), dim3(256), 0, s, m, n_agents, ld_map, c, logits, ld_w, dfused, ld_df, h,
                       ld_h, w4, (long long)pixels, dmap0, ld_dm, dh, acc);
  else
    hipLaunchKernelGGL((k_fuse_backward<float>), dim3((unsigned)blocks), dim3(256), 0, s, m, n_agents, ld_map, c, logits, ld_w, dfused, ld_df, h,
                       ld_h, w4, (long long)pixels, dmap0, ld_dm, dh, acc);
  hipLaunchKernelGGL(k_w4_finalize, dim3(1), dim3(64), 0, s, acc, dw4, db4, accumulate);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

int pcp_disco_fuse_backward(const float *const *maps_host, int32_t n_agents, int32_t ld_map, int32_t c, const float *logits, int32_t ld_w,
                            const float *dfused, int32_t ld_df, const float *const *h2_host, int32_t ld_h, const float *w4,
                            int64_t pixels, float *dmap0, int32_t ld_dm, float *const *dh2_host, void *workspace, float *dw4, float *db4,
                            int32_t accumulate, void *stream) {
  return fuse_backward_impl((const void *const *)maps_host, 0, n_agents, ld_map, c, logits, ld_w, dfused, ld_df, h2_host, ld_h, w4, pixels, dmap0,
                            ld_dm, dh2_host, workspace, dw4, db4, accumulate, stream);
}

int pcp_mp_disco_fuse_backward(const void *const *maps_host, int32_t map_dtype, int32_t n_agents, int32_t ld_map, int32_t c, const float *logits,
                               int32_t ld_w, const float *dfused, int32_t ld_df, const float *const *h2_host, int32_t ld_h, const float *w4,
                               int64_t pixels, float *dmap0, int32_t ld_dm, float *const *dh2_host, void *workspace, float *dw4, float *db4,
                               int32_t accumulate, void *stream) {
  if (map_dtype & ~1) return PCP_ERR_ARG;
  return fuse_backward_impl(maps_host, map_dtype, n_agents, ld_map, c, logits, ld_w, dfused, ld_df, h2_host, ld_h, w4, pixels, dmap0, ld_dm,
                            dh2_host, workspace, dw4, db4, accumulate, stream);
}

}  // extern "C"
