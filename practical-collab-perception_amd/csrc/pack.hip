// Weight repacking for the training step: the optimizer changes every weight every iteration, so the MFMA layouts of the forward
// conv AND of its data-gradient conv (flipped taps, swapped channel roles) are rebuilt per step.  One launch per 3x3 layer writes
// up to two layouts (direct implicit-GEMM [I/16][9][O_pad][16] and Winograd U = G g G^T [I/8][16][O_pad][8], float64 transform,
// one rounding -- same arithmetic as pcp_amd/pack.py) instead of ~20 small ATen launches + two einsum GEMMs.
#include "pcp_common.h"

namespace {

struct PackParams {
  const float *w;        // (cout, cin, 3, 3) PyTorch layout
  int cout, cin;
  int transpose;         // 0: E[o][i][t] = w[o][i][t];  1 (data gradient): E[o][i][ky][kx] = w[i][o][2-ky][2-kx]
  float *direct;         // may be NULL
  int direct_opad;
  float *wino;           // may be NULL
  int wino_opad;
  __bf16 *b3;            // may be NULL: split-bf16 layout of conv_bf16x3.hip [I/16][O_pad/64][hi|lo][9][2][64][8]
  int b3_opad;
};

__device__ __forceinline__ float eff(const PackParams &p, int o, int i, int tap) {
  if (!p.transpose) return p.w[((long long)o * p.cin + i) * 9 + tap];
  return p.w[((long long)i * p.cin + o) * 9 + (8 - tap)];
}

__device__ __forceinline__ void pack3x3_body(const PackParams &p, const long long t) {
  const int O = p.transpose ? p.cin : p.cout, I = p.transpose ? p.cout : p.cin;
  if (p.direct) {
    const long long total = (long long)(I / 16) * 9 * p.direct_opad * 16;
    if (t < total) {
      const int k = (int)(t & 15);
      long long r = t >> 4;
      const int o = (int)(r % p.direct_opad);
      r /= p.direct_opad;
      const int tap = (int)(r % 9);
      const int s = (int)(r / 9);
      p.direct[t] = o < O ? eff(p, o, s * 16 + k, tap) : 0.f;
    }
  }
  if (p.b3) {
    const long long total = (long long)(I / 16) * p.b3_opad * 9 * 16;              // one thread per (slice, cout, tap, channel)
    if (t < total) {
      const int j = (int)(t & 7), hh = (int)((t >> 3) & 1);
      long long r = t >> 4;
      const int n = (int)(r & 63);
      r >>= 6;
      const int tap = (int)(r % 9);
      r /= 9;
      const int n_ct = p.b3_opad / 64;
      const int ct = (int)(r % n_ct);
      const int sl = (int)(r / n_ct);
      const int o = ct * 64 + n;
      const float v = o < O ? eff(p, o, sl * 16 + hh * 8 + j, tap) : 0.f;
      const __bf16 hi = (__bf16)v;
      const __bf16 lo = (__bf16)(v - (float)hi);
      const long long base = ((long long)sl * n_ct + ct) * (2 * 9 * 2 * 64 * 8);
      const long long off = ((long long)(tap * 2 + hh) * 64 + n) * 8 + j;
      p.b3[base + off] = hi;
      p.b3[base + 9 * 2 * 64 * 8 + off] = lo;
    }
  }
  if (p.wino) {
    const long long total = (long long)(I / 8) * p.wino_opad * 8;
    if (t < total) {
      const int k = (int)(t & 7);
      long long r = t >> 3;
      const int o = (int)(r % p.wino_opad);
      const int s = (int)(r / p.wino_opad);
      double g[3][3];
#pragma unroll
      for (int a = 0; a < 9; ++a) g[a / 3][a % 3] = o < O ? (double)eff(p, o, s * 8 + k, a) : 0.0;
      // G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
      double gg[4][3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        gg[0][j] = g[0][j];
        gg[1][j] = 0.5 * (g[0][j] + g[1][j] + g[2][j]);
        gg[2][j] = 0.5 * (g[0][j] - g[1][j] + g[2][j]);
        gg[3][j] = g[2][j];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const double u0 = gg[i][0], u1 = 0.5 * (gg[i][0] + gg[i][1] + gg[i][2]), u2 = 0.5 * (gg[i][0] - gg[i][1] + gg[i][2]), u3 = gg[i][2];
        const double u[4] = {u0, u1, u2, u3};
#pragma unroll
        for (int j = 0; j < 4; ++j)
          p.wino[(((long long)s * 16 + i * 4 + j) * p.wino_opad + o) * 8 + k] = (float)u[j];
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_pack3x3(PackParams p) {
  pack3x3_body(p, (long long)blockIdx.x * blockDim.x + threadIdx.x);
}

// U = G g G^T of Winograd F(4x4,3x3) (float64, one rounding; G as in pcp_amd/pack.py) in the fragment orders of the two fused kernels:
//   u4f [I/8][36][O_pad][8]                      (csrc/wino4f.hip)
//   u4h [I/8][36][O_pad/64][64 lanes][8]         (csrc/wino4h.hip: lane = 16 kq + c, index 2 nb + ks <-> channel 4 ks + kq, output 16 nb + c)
// one thread per (slice, output channel, channel of the slice): 36 stores to each layout.
struct Pack4Params {
  const float *w;
  int cout, cin, transpose;
  float *u4f, *u4h;
  int opad;
};

__device__ __forceinline__ void pack3x3_wino4_body(const Pack4Params &q, const long long t) {
  const int O = q.transpose ? q.cin : q.cout, I = q.transpose ? q.cout : q.cin;
  if (t >= (long long)(I / 8) * q.opad * 8) return;
  const int k = (int)(t & 7);
  long long r = t >> 3;
  const int o = (int)(r % q.opad);
  const int s = (int)(r / q.opad);
  PackParams p{};
  p.w = q.w; p.cout = q.cout; p.cin = q.cin; p.transpose = q.transpose;
  double g[3][3];
#pragma unroll
  for (int a = 0; a < 9; ++a) g[a / 3][a % 3] = o < O ? (double)eff(p, o, s * 8 + k, a) : 0.0;
  const double G[6][3] = {{1.0 / 4, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                          {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
  double gg[6][3];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) gg[i][j] = G[i][0] * g[0][j] + G[i][1] * g[1][j] + G[i][2] * g[2][j];
  const int nblk = o >> 6, ol = o & 63;
  const int lane = 16 * (k & 3) + (ol & 15), idx = 2 * (ol >> 4) + (k >> 2);
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const float u = (float)(gg[i][0] * G[j][0] + gg[i][1] * G[j][1] + gg[i][2] * G[j][2]);
      const long long pos = (long long)s * 36 + i * 6 + j;
      if (q.u4f) q.u4f[(pos * q.opad + o) * 8 + k] = u;
      if (q.u4h) q.u4h[((pos * (q.opad >> 6) + nblk) * 64 + lane) * 8 + idx] = u;
    }
}

__global__ __launch_bounds__(256) void k_pack3x3_wino4(Pack4Params q) {
  pack3x3_wino4_body(q, (long long)blockIdx.x * blockDim.x + threadIdx.x);
}

// every 3x3 layer of the training step in ONE launch: block -> job by binary search over the jobs' first blocks
__global__ __launch_bounds__(256) void k_pack3x3_group(const pcp_pack_job_t *__restrict__ jobs, int n_jobs) {
  int lo = 0, hi = n_jobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block_start <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const pcp_pack_job_t j = jobs[lo];
  const long long t = (long long)((int)blockIdx.x - j.block_start) * blockDim.x + threadIdx.x;
  if (j.direct || j.winograd) {
    PackParams p{};
    p.w = j.w; p.cout = j.cout; p.cin = j.cin; p.transpose = j.transpose;
    p.direct = j.direct; p.direct_opad = j.direct_cout_pad; p.wino = j.winograd; p.wino_opad = j.winograd_cout_pad;
    pack3x3_body(p, t);
  }
  if (j.u4f || j.u4h) {
    Pack4Params q{j.w, j.cout, j.cin, j.transpose, j.u4f, j.u4h, j.f4_cout_pad};
    pack3x3_wino4_body(q, t);
  }
}

}  // namespace

extern "C" int pcp_pack_conv3x3_group_blocks(const pcp_pack_job_t *job) {
  if (!job || !job->w || job->cout <= 0 || job->cin <= 0) return -1;
  const int O = job->transpose ? job->cin : job->cout, I = job->transpose ? job->cout : job->cin;
  if ((job->direct && ((I & 15) || job->direct_cout_pad < O)) || (job->winograd && ((I & 7) || job->winograd_cout_pad < O)) ||
      ((job->u4f || job->u4h) && ((I & 7) || job->f4_cout_pad < O || (job->f4_cout_pad & 63))))
    return -1;
  long long n = 0;
  if (job->direct) n = (long long)(I / 16) * 9 * job->direct_cout_pad * 16;
  if (job->winograd) { const long long m = (long long)(I / 8) * job->winograd_cout_pad * 8; if (m > n) n = m; }
  if (job->u4f || job->u4h) { const long long m = (long long)(I / 8) * job->f4_cout_pad * 8; if (m > n) n = m; }
  if (n <= 0 || n > 0x7fffffffLL) return -1;
  return (int)((n + 255) / 256);
}

extern "C" int pcp_pack_conv3x3_group(const pcp_pack_job_t *jobs_device, int32_t n_jobs, int32_t total_blocks, void *stream) {
  if (!jobs_device || n_jobs <= 0 || total_blocks <= 0) return PCP_ERR_ARG;
  hipLaunchKernelGGL(k_pack3x3_group, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_device, n_jobs);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

extern "C" int pcp_pack_conv3x3_winograd4(const float *w, int32_t cout, int32_t cin, int32_t transpose, float *u4f, float *u4h,
                                          int32_t cout_pad, void *stream) {
  if (!w || cout <= 0 || cin <= 0 || (!u4f && !u4h)) return PCP_ERR_ARG;
  const int O = transpose ? cin : cout, I = transpose ? cout : cin;
  if ((I & 7) || cout_pad < O || (cout_pad & 63)) return PCP_ERR_ARG;
  Pack4Params q{w, cout, cin, transpose, u4f, u4h, cout_pad};
  const long long n = (long long)(I / 8) * cout_pad * 8;
  hipLaunchKernelGGL(k_pack3x3_wino4, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, q);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

extern "C" int pcp_pack_conv3x3(const float *w, int32_t cout, int32_t cin, int32_t transpose, float *direct, int32_t direct_cout_pad,
                                float *winograd, int32_t winograd_cout_pad, void *split_bf16, int32_t split_cout_pad, void *stream) {
  if (!w || cout <= 0 || cin <= 0 || (!direct && !winograd && !split_bf16)) return PCP_ERR_ARG;
  const int O = transpose ? cin : cout, I = transpose ? cout : cin;
  if (direct && ((I & 15) || direct_cout_pad < O)) return PCP_ERR_ARG;
  if (winograd && ((I & 7) || winograd_cout_pad < O)) return PCP_ERR_ARG;
  if (split_bf16 && ((I & 15) || split_cout_pad < O || (split_cout_pad & 63))) return PCP_ERR_ARG;
  PackParams p{w, cout, cin, transpose, direct, direct_cout_pad, winograd, winograd_cout_pad, (__bf16 *)split_bf16, split_cout_pad};
  long long n = 0;
  if (split_bf16) n = (long long)(I / 16) * split_cout_pad * 9 * 16;
  if (direct) { const long long m = (long long)(I / 16) * 9 * direct_cout_pad * 16; if (m > n) n = m; }
  if (winograd) { const long long m = (long long)(I / 8) * winograd_cout_pad * 8; if (m > n) n = m; }
  hipLaunchKernelGGL(k_pack3x3, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}
