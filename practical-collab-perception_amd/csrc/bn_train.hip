// Training-mode BatchNorm (batch statistics) + ReLU on NHWC / row-major (rows, c) fp32 tensors, forward and backward, plus the
// small row-wise helpers the backward pass needs (column sums for bias gradients, accumulate, 2x zero-insertion).
//
// Replaces, for the trainable branch of config 5, what cuDNN/ATen do behind
//   nn.BatchNorm2d / nn.BatchNorm1d in train() mode + nn.ReLU and their autograd nodes
//   (pcdet/models/backbones_2d/base_bev_backbone.py:37-44,56,67; backbones_3d/vfe/dynamic_pillar_vfe.py:29,40-43;
//    dense_heads/center_head.py:26,80; bev_layers/v2x_fusion_disco.py:13-16,53,60).
//
// All of it is HBM-bound streaming: one read for the statistics, one read + one write for the normalisation; backward is one
// read of (dout, x) for the two per-channel sums and one read + write for dx.  Per-channel sums are accumulated in float64
// (per-thread partials, LDS tree, one partial per block and channel in the workspace, reduced in a fixed order by the finalize
// kernel): bitwise reproducible, no atomics.
//
// Mixed-precision training (round 4, include/pcp_hip_mp.h): every kernel here is a template over the STORAGE type of each tensor it
// touches (float | bf16, four channels per thread: 16-byte / 8-byte accesses); the arithmetic is fp32 (sums float64) in every
// instantiation, and the <float, float> instantiations are the round-1..3 kernels unchanged.
#include "pcp_common.h"
#include "../../include/pcp_hip_mp.h"

namespace {

typedef __bf16 bf16_t;

template <typename T> struct IO4;
template <> struct IO4<float> {
  static __device__ __forceinline__ float4 ld(const float *p) { return *reinterpret_cast<const float4 *>(p); }
  static __device__ __forceinline__ void st(float *p, const float4 &v) { *reinterpret_cast<float4 *>(p) = v; }
};
template <> struct IO4<bf16_t> {
  static __device__ __forceinline__ float4 ld(const bf16_t *p) {
    const uint2 r = *reinterpret_cast<const uint2 *>(p);
    return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                       __uint_as_float(r.y & 0xffff0000u));
  }
  static __device__ __forceinline__ void st(bf16_t *p, const float4 &v) {
    typedef __bf16 b4 __attribute__((ext_vector_type(4)));
    b4 o;
    o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;      // round to nearest even (v_cvt_pk_bf16_f32)
    *reinterpret_cast<b4 *>(p) = o;
  }
};

constexpr int RED_THREADS = 256;
constexpr int RED_MAX_BLOCKS = 512;      // partials per reduction: the finalize kernels read nb x 2c doubles (1024 -> 512: their ~9 us halve, the streaming pass keeps its rate)

enum { RED_STATS = 0, RED_BNBWD = 1, RED_COLSUM = 2 };

struct RedParams {
  const void *x;         // STATS / COLSUM: the tensor; BNBWD: pre-BN conv output          (storage type XT)
  const void *dout;      // BNBWD: upstream gradient                                       (storage type DT)
  long long rows;
  int c, ld_x, ld_d;
  const float *scale, *shift, *mean, *invstd;   // BNBWD
  int relu;
  double *acc;           // [2][c]
};

// thread layout: cg = c/4 float4 column groups, rpb = RED_THREADS / cg row slots per block pass
template <int MODE, typename XT, typename DT>
__global__ __launch_bounds__(RED_THREADS) void k_col_reduce(RedParams p) {
  const XT *px = reinterpret_cast<const XT *>(p.x);
  const DT *pd = reinterpret_cast<const DT *>(p.dout);
  __shared__ double red[2][RED_THREADS][4];
  const int cg = p.c >> 2;
  const int rpb = RED_THREADS / cg;
  const int tid = threadIdx.x;
  const int slot = tid / cg, g = tid - slot * cg;
  double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
  if (slot < rpb) {
    float4 sc = make_float4(1, 1, 1, 1), sh = make_float4(0, 0, 0, 0), mu = sh, is = sc;
    if (MODE == RED_BNBWD) {
      sc = *reinterpret_cast<const float4 *>(p.scale + g * 4);
      sh = *reinterpret_cast<const float4 *>(p.shift + g * 4);
      mu = *reinterpret_cast<const float4 *>(p.mean + g * 4);
      is = *reinterpret_cast<const float4 *>(p.invstd + g * 4);
    }
    // four rows per trip: the loads are issued back to back (independent), so a thread keeps 64-128 B in flight -- with one row per
    // trip the kernel was latency bound at ~1 TB/s
    const long long step = (long long)gridDim.x * rpb;
    auto accum = [&](const float4 &v, const float4 &d) {
      const float xv[4] = {v.x, v.y, v.z, v.w};
      if (MODE == RED_STATS) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { s0[i] += (double)xv[i]; s1[i] += (double)xv[i] * (double)xv[i]; }
      } else if (MODE == RED_COLSUM) {
#pragma unroll
        for (int i = 0; i < 4; ++i) s0[i] += (double)xv[i];
      } else {
        const float dv[4] = {d.x, d.y, d.z, d.w};
        const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
        const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, isv[4] = {is.x, is.y, is.z, is.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float act = fmaf(xv[i], scv[i], shv[i]);
          const float dz = (p.relu && !(act > 0.f)) ? 0.f : dv[i];
          const float xh = (xv[i] - muv[i]) * isv[i];
          s0[i] += (double)dz;
          s1[i] += (double)dz * (double)xh;
        }
      }
    };
    long long r = (long long)blockIdx.x * rpb + slot;
    for (; r + 3 * step < p.rows; r += 4 * step) {
      float4 v[4], d[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        v[u] = IO4<XT>::ld(px + (r + u * step) * p.ld_x + g * 4);
        d[u] = MODE == RED_BNBWD ? IO4<DT>::ld(pd + (r + u * step) * p.ld_d + g * 4) : v[u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) accum(v[u], d[u]);
    }
    for (; r < p.rows; r += step) {
      const float4 v = IO4<XT>::ld(px + r * p.ld_x + g * 4);
      const float4 d = MODE == RED_BNBWD ? IO4<DT>::ld(pd + r * p.ld_d + g * 4) : v;
      accum(v, d);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) { red[0][tid][i] = s0[i]; red[1][tid][i] = s1[i]; }
  __syncthreads();
  if (slot == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      double a = 0, b = 0;
      for (int s = 0; s < rpb; ++s) { a += red[0][s * cg + g][i]; b += red[1][s * cg + g][i]; }
      // block partial (no atomics: 1024 f64 atomics per address serialise at L2 for ~40 us; the finalize kernels reduce the
      // partials in a fixed order, which also makes the sums bitwise reproducible)
      double *part = p.acc + (long long)blockIdx.x * 2 * p.c;
      part[g * 4 + i] = a;
      part[p.c + g * 4 + i] = b;
    }
  }
}

constexpr int FIN_PARTS = 64;     // threads cooperating on one channel
constexpr int FIN_CH = 4;         // channels per block

// sum over the nb block partials of channel ch: returns (sum0, sum1) on the thread with part == 0
__device__ __forceinline__ void reduce_partials(const double *acc, int nb, int c, int ch, int part, int lc, double &o0, double &o1) {
  __shared__ double sh[2][FIN_CH][FIN_PARTS];
  double a = 0, b = 0;
  if (ch < c)
    for (int blk = part; blk < nb; blk += FIN_PARTS) {
      a += acc[(long long)blk * 2 * c + ch];
      b += acc[(long long)blk * 2 * c + c + ch];
    }
  sh[0][lc][part] = a;
  sh[1][lc][part] = b;
  __syncthreads();
  o0 = o1 = 0;
  if (part == 0)
    for (int i = 0; i < FIN_PARTS; ++i) { o0 += sh[0][lc][i]; o1 += sh[1][lc][i]; }
}

__global__ __launch_bounds__(FIN_CH * FIN_PARTS) void k_bn_finalize(const double *acc, int nb, long long rows, int c, const float *gamma,
                                                                  const float *beta, float eps, float momentum, float *running_mean,
                                                                  float *running_var, float *scale, float *shift, float *mean,
                                                                  float *invstd) {
  const int lc = threadIdx.x / FIN_PARTS, part = threadIdx.x % FIN_PARTS;
  const int i = blockIdx.x * FIN_CH + lc;
  double s0, s1;
  reduce_partials(acc, nb, c, i, part, lc, s0, s1);
  if (part != 0 || i >= c) return;
  const double n = (double)rows;
  const double m = s0 / n;
  double var = s1 / n - m * m;
  if (var < 0) var = 0;
  const double is = 1.0 / sqrt(var + (double)eps);
  const float sc = (float)((double)gamma[i] * is);
  scale[i] = sc;
  shift[i] = (float)((double)beta[i] - m * (double)gamma[i] * is);
  mean[i] = (float)m;
  invstd[i] = (float)is;
  if (running_mean) {
    const double unb = rows > 1 ? var * n / (n - 1.0) : var;
    running_mean[i] = (float)((1.0 - (double)momentum) * (double)running_mean[i] + (double)momentum * m);
    running_var[i] = (float)((1.0 - (double)momentum) * (double)running_var[i] + (double)momentum * unb);
  }
}

// dgamma = sum dz xhat, dbeta = sum dz; coef[0][c] = dbeta / n, coef[1][c] = dgamma / n for the apply pass
__global__ __launch_bounds__(FIN_CH * FIN_PARTS) void k_bnbwd_finalize(const double *acc, int nb, long long rows, int c, float *dgamma,
                                                                     float *dbeta, int accumulate, float *coef) {
  const int lc = threadIdx.x / FIN_PARTS, part = threadIdx.x % FIN_PARTS;
  const int i = blockIdx.x * FIN_CH + lc;
  double s0, s1;
  reduce_partials(acc, nb, c, i, part, lc, s0, s1);
  if (part != 0 || i >= c) return;
  const double n = (double)rows;
  const float db = (float)s0, dg = (float)s1;
  if (accumulate) { dbeta[i] += db; dgamma[i] += dg; } else { dbeta[i] = db; dgamma[i] = dg; }
  coef[i] = (float)(s0 / n);
  coef[c + i] = (float)(s1 / n);
}

// block partials -> the two per-channel float64 sums themselves (sums[0][c], sums[1][c]): the hand-off point of cross-rank BatchNorm
__global__ __launch_bounds__(FIN_CH * FIN_PARTS) void k_sums_finalize(const double *acc, int nb, int c, double *sums) {
  const int lc = threadIdx.x / FIN_PARTS, part = threadIdx.x % FIN_PARTS;
  const int i = blockIdx.x * FIN_CH + lc;
  double s0, s1;
  reduce_partials(acc, nb, c, i, part, lc, s0, s1);
  if (part != 0 || i >= c) return;
  sums[i] = s0;
  sums[c + i] = s1;
}

// dgamma / dbeta from THIS rank's sums (the optimizer's gradient all-reduce combines them, as DDP does for nn.SyncBatchNorm); the mean
// terms of dx from the sums over ALL ranks
__global__ void k_bnbwd_finalize_sync(const double *local, const double *global, double n_total, int c, float *dgamma, float *dbeta,
                                      int accumulate, float *coef) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c) return;
  const float db = (float)local[i], dg = (float)local[c + i];
  if (accumulate) { dbeta[i] += db; dgamma[i] += dg; } else { dbeta[i] = db; dgamma[i] = dg; }
  coef[i] = (float)(global[i] / n_total);
  coef[c + i] = (float)(global[c + i] / n_total);
}

__global__ __launch_bounds__(FIN_CH * FIN_PARTS) void k_colsum_finalize(const double *acc, int nb, int c, float *out, int accumulate) {
  const int lc = threadIdx.x / FIN_PARTS, part = threadIdx.x % FIN_PARTS;
  const int i = blockIdx.x * FIN_CH + lc;
  double s0, s1;
  reduce_partials(acc, nb, c, i, part, lc, s0, s1);
  if (part != 0 || i >= c) return;
  if (accumulate) out[i] += (float)s0; else out[i] = (float)s0;
}

template <typename XT, typename OT>
__global__ void k_scale_shift_act(const XT *__restrict__ x, long long rows, int cg, int ld_x, const float *__restrict__ scale,
                                  const float *__restrict__ shift, int relu, OT *__restrict__ out, int ld_out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rows * cg) return;
  const int g = (int)(t % cg);
  const long long r = t / cg;
  const float4 v = IO4<XT>::ld(x + r * ld_x + g * 4);
  const float4 sc = *reinterpret_cast<const float4 *>(scale + g * 4);
  const float4 sh = *reinterpret_cast<const float4 *>(shift + g * 4);
  float4 o = make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w));
  if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
  IO4<OT>::st(out + r * ld_out + g * 4, o);
}

template <typename DT, typename XT, typename OT>
__global__ void k_bnbwd_apply(const DT *dout, int ld_d, const XT *__restrict__ x, int ld_x, long long rows, int cg,
                              int c, const float *__restrict__ scale, const float *__restrict__ shift, const float *__restrict__ mean,
                              const float *__restrict__ invstd, int relu, const float *__restrict__ coef, OT *dx,
                              int ld_dx) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rows * cg) return;
  const int g = (int)(t % cg);
  const long long r = t / cg;
  const float4 v = IO4<XT>::ld(x + r * ld_x + g * 4);
  const float4 d = IO4<DT>::ld(dout + r * ld_d + g * 4);
  const float xv[4] = {v.x, v.y, v.z, v.w}, dv[4] = {d.x, d.y, d.z, d.w};
  float o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ch = g * 4 + i;
    const float act = fmaf(xv[i], scale[ch], shift[ch]);
    const float dz = (relu && !(act > 0.f)) ? 0.f : dv[i];
    const float xh = (xv[i] - mean[ch]) * invstd[ch];
    o[i] = scale[ch] * (dz - coef[ch] - xh * coef[c + ch]);
  }
  IO4<OT>::st(dx + r * ld_dx + g * 4, make_float4(o[0], o[1], o[2], o[3]));
}

template <typename AT, typename BT>
__global__ void k_accumulate(AT *__restrict__ dst, int ld_dst, const BT *__restrict__ src, int ld_src, long long rows, int cg,
                             float alpha) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rows * cg) return;
  const int g = (int)(t % cg);
  const long long r = t / cg;
  float4 a = IO4<AT>::ld(dst + r * ld_dst + g * 4);
  const float4 b = IO4<BT>::ld(src + r * ld_src + g * 4);
  a.x = fmaf(alpha, b.x, a.x); a.y = fmaf(alpha, b.y, a.y); a.z = fmaf(alpha, b.z, a.z); a.w = fmaf(alpha, b.w, a.w);
  IO4<AT>::st(dst + r * ld_dst + g * 4, a);
}

// out (B, 2h, 2w, c): out[b, 2y, 2x] = in[b, y, x], zero elsewhere (gradient of a stride-2 3x3 conv = stride-1 conv of this)
template <typename T>
__global__ void k_dilate2x(const T *__restrict__ in, int batch, int h, int w, int cg, int ld_in, T *__restrict__ out,
                           int ld_out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)batch * 2 * h * 2 * w * cg;
  if (t >= total) return;
  const int g = (int)(t % cg);
  long long pix = t / cg;
  const int ox = (int)(pix % (2 * w));
  pix /= 2 * w;
  const int oy = (int)(pix % (2 * h));
  const int b = (int)(pix / (2 * h));
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!(ox & 1) && !(oy & 1)) v = IO4<T>::ld(in + (((long long)b * h + (oy >> 1)) * w + (ox >> 1)) * ld_in + g * 4);
  IO4<T>::st(out + (((long long)b * 2 * h + oy) * 2 * w + ox) * ld_out + g * 4, v);
}

inline bool red_shape_ok(long long rows, int c, int ld) { return rows > 0 && c >= 4 && (c & 3) == 0 && (c >> 2) <= RED_THREADS && (ld & 3) == 0 && ld >= c; }
inline bool al16(const void *p) { return (((uintptr_t)p) & 15) == 0; }
// four channels of a row: 16 bytes of float, 8 bytes of bf16
inline bool al4ch(const void *p, int dtype) { return (((uintptr_t)p) & (dtype == PCP_DT_BF16 ? 7 : 15)) == 0; }
inline bool dt_ok(int dtype) { return dtype == PCP_DT_F32 || dtype == PCP_DT_BF16; }

template <int MODE>
inline void launch_col_reduce(int nb, hipStream_t s, const RedParams &p, int xdt, int ddt) {
  if (xdt == PCP_DT_BF16) {
    if (ddt == PCP_DT_BF16) hipLaunchKernelGGL((k_col_reduce<MODE, bf16_t, bf16_t>), dim3(nb), dim3(RED_THREADS), 0, s, p);
    else hipLaunchKernelGGL((k_col_reduce<MODE, bf16_t, float>), dim3(nb), dim3(RED_THREADS), 0, s, p);
  } else {
    if (ddt == PCP_DT_BF16) hipLaunchKernelGGL((k_col_reduce<MODE, float, bf16_t>), dim3(nb), dim3(RED_THREADS), 0, s, p);
    else hipLaunchKernelGGL((k_col_reduce<MODE, float, float>), dim3(nb), dim3(RED_THREADS), 0, s, p);
  }
}

template <typename DT, typename XT>
inline void launch_bnbwd_apply2(int odt, unsigned blocks, hipStream_t s, const void *dout, int ld_d, const void *x, int ld_x, long long rows, int cg,
                                int c, const float *scale, const float *shift, const float *mean, const float *invstd, int relu,
                                const float *coef, void *dx, int ld_dx) {
  if (odt == PCP_DT_BF16)
    hipLaunchKernelGGL((k_bnbwd_apply<DT, XT, bf16_t>), dim3(blocks), dim3(256), 0, s, (const DT *)dout, ld_d, (const XT *)x, ld_x, rows, cg, c, scale,
                       shift, mean, invstd, relu, coef, (bf16_t *)dx, ld_dx);
  else
    hipLaunchKernelGGL((k_bnbwd_apply<DT, XT, float>), dim3(blocks), dim3(256), 0, s, (const DT *)dout, ld_d, (const XT *)x, ld_x, rows, cg, c, scale,
                       shift, mean, invstd, relu, coef, (float *)dx, ld_dx);
}

inline void launch_bnbwd_apply(int ddt, int xdt, int odt, unsigned blocks, hipStream_t s, const void *dout, int ld_d, const void *x, int ld_x,
                               long long rows, int cg, int c, const float *scale, const float *shift, const float *mean, const float *invstd,
                               int relu, const float *coef, void *dx, int ld_dx) {
#define PCP_BNB(DT, XT) launch_bnbwd_apply2<DT, XT>(odt, blocks, s, dout, ld_d, x, ld_x, rows, cg, c, scale, shift, mean, invstd, relu, coef, dx, ld_dx)
  if (ddt == PCP_DT_BF16) { if (xdt == PCP_DT_BF16) PCP_BNB(bf16_t, bf16_t); else PCP_BNB(bf16_t, float); }
  else { if (xdt == PCP_DT_BF16) PCP_BNB(float, bf16_t); else PCP_BNB(float, float); }
#undef PCP_BNB
}

// one f64 atomic per (block, channel) lands on 2c addresses: keep the block count near the CU count (256) so the atomics of a
// launch do not serialise at L2 (2048 blocks cost 40 us of pure atomic contention on a 64-channel map)
inline int red_grid(long long rows, int c) {
  const int rpb = RED_THREADS / (c >> 2);
  long long blocks = (rows + (long long)rpb * 8 - 1) / ((long long)rpb * 8);
  if (blocks > RED_MAX_BLOCKS) blocks = RED_MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

}  // namespace

// ---- implementations over (pointer, storage type) pairs; the fp32 entry points of pcp_hip_train.h pass PCP_DT_F32 everywhere ----------
namespace {

int bn_train_stats_impl(const void *x, int xdt, int64_t rows, int32_t c, int32_t ld, const float *gamma, const float *beta, float eps,
                        float momentum, float *running_mean, float *running_var, void *workspace, float *scale, float *shift, float *mean,
                        float *invstd, void *stream) {
  if (!x || !gamma || !beta || !workspace || !scale || !shift || !mean || !invstd || !red_shape_ok(rows, c, ld) || !dt_ok(xdt) || !al4ch(x, xdt))
    return PCP_ERR_ARG;
  if ((running_mean == nullptr) != (running_var == nullptr)) return PCP_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  double *acc = (double *)workspace;
  RedParams p{};
  p.x = x; p.rows = rows; p.c = c; p.ld_x = ld; p.acc = acc;
  const int nb = red_grid(rows, c);
  launch_col_reduce<RED_STATS>(nb, s, p, xdt, xdt);
  hipLaunchKernelGGL(k_bn_finalize, dim3((c + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_PARTS), 0, s, acc, nb, (long long)rows, c, gamma, beta,
                     eps, momentum, running_mean, running_var, scale, shift, mean, invstd);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

int scale_shift_act_impl(const void *x, int xdt, int64_t rows, int32_t c, int32_t ld_x, const float *scale, const float *shift, int32_t relu,
                         void *out, int odt, int32_t ld_out, void *stream) {
  if (!x || !scale || !shift || !out || rows <= 0 || c < 4 || (c & 3) || (ld_x & 3) || (ld_out & 3) || !dt_ok(xdt) || !dt_ok(odt) ||
      !al4ch(x, xdt) || !al4ch(out, odt))
    return PCP_ERR_ARG;
  const long long total = (long long)rows * (c >> 2);
  const dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t s = (hipStream_t)stream;
#define PCP_SSA(XT, OT) hipLaunchKernelGGL((k_scale_shift_act<XT, OT>), grid, dim3(256), 0, s, (const XT *)x, (long long)rows, c >> 2, ld_x, scale, shift, relu, (OT *)out, ld_out)
  if (xdt == PCP_DT_BF16) { if (odt == PCP_DT_BF16) PCP_SSA(bf16_t, bf16_t); else PCP_SSA(bf16_t, float); }
  else { if (odt == PCP_DT_BF16) PCP_SSA(float, bf16_t); else PCP_SSA(float, float); }
#undef PCP_SSA
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

int bn_act_backward_impl(const void *dout, int ddt, int32_t ld_dout, const void *x, int xdt, int32_t ld_x, int64_t rows, int32_t c,
                         const float *scale, const float *shift, const float *mean, const float *invstd, int32_t relu, void *workspace,
                         float *dgamma, float *dbeta, int32_t accumulate, void *dx, int odt, int32_t ld_dx, void *stream) {
  if (!dout || !x || !scale || !shift || !mean || !invstd || !workspace || !dgamma || !dbeta || !dx || !red_shape_ok(rows, c, ld_x) ||
      (ld_dout & 3) || (ld_dx & 3) || !dt_ok(ddt) || !dt_ok(xdt) || !dt_ok(odt) || !al4ch(dout, ddt) || !al4ch(x, xdt) || !al4ch(dx, odt))
    return PCP_ERR_ARG;
  if (dx == dout && odt != ddt) return PCP_ERR_ARG;          // in place only at one storage type
  hipStream_t s = (hipStream_t)stream;
  double *acc = (double *)workspace;
  float *coef = (float *)(acc + 2 * (size_t)c * RED_MAX_BLOCKS);
  RedParams p{};
  p.x = x; p.dout = dout; p.rows = rows; p.c = c; p.ld_x = ld_x; p.ld_d = ld_dout;
  p.scale = scale; p.shift = shift; p.mean = mean; p.invstd = invstd; p.relu = relu; p.acc = acc;
  const int nb = red_grid(rows, c);
  launch_col_reduce<RED_BNBWD>(nb, s, p, xdt, ddt);
  hipLaunchKernelGGL(k_bnbwd_finalize, dim3((c + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_PARTS), 0, s, acc, nb, (long long)rows, c, dgamma,
                     dbeta, accumulate, coef);
  const long long total = (long long)rows * (c >> 2);
  launch_bnbwd_apply(ddt, xdt, odt, (unsigned)((total + 255) / 256), s, dout, ld_dout, x, ld_x, (long long)rows, c >> 2, c, scale, shift, mean,
                     invstd, relu, coef, dx, ld_dx);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

int bn_train_sums_impl(const void *x, int xdt, int64_t rows, int32_t c, int32_t ld, void *workspace, double *sums, void *stream) {
  if (!x || !workspace || !sums || !red_shape_ok(rows, c, ld) || !dt_ok(xdt) || !al4ch(x, xdt)) return PCP_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  double *acc = (double *)workspace;
  RedParams p{};
  p.x = x; p.rows = rows; p.c = c; p.ld_x = ld; p.acc = acc;
  const int nb = red_grid(rows, c);
  launch_col_reduce<RED_STATS>(nb, s, p, xdt, xdt);
  hipLaunchKernelGGL(k_sums_finalize, dim3((c + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_PARTS), 0, s, acc, nb, c, sums);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

int bn_bwd_sums_impl(const void *dout, int ddt, int32_t ld_dout, const void *x, int xdt, int32_t ld_x, int64_t rows, int32_t c,
                     const float *scale, const float *shift, const float *mean, const float *invstd, int32_t relu, void *workspace,
                     double *sums, void *stream) {
  if (!dout || !x || !scale || !shift || !mean || !invstd || !workspace || !sums || !red_shape_ok(rows, c, ld_x) || (ld_dout & 3) ||
      !dt_ok(ddt) || !dt_ok(xdt) || !al4ch(dout, ddt) || !al4ch(x, xdt))
    return PCP_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  double *acc = (double *)workspace;
  RedParams p{};
  p.x = x; p.dout = dout; p.rows = rows; p.c = c; p.ld_x = ld_x; p.ld_d = ld_dout;
  p.scale = scale; p.shift = shift; p.mean = mean; p.invstd = invstd; p.relu = relu; p.acc = acc;
  const int nb = red_grid(rows, c);
  launch_col_reduce<RED_BNBWD>(nb, s, p, xdt, ddt);
  hipLaunchKernelGGL(k_sums_finalize, dim3((c + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_PARTS), 0, s, acc, nb, c, sums);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

int bn_bwd_apply_from_sums_impl(const void *dout, int ddt, int32_t ld_dout, const void *x, int xdt, int32_t ld_x, int64_t rows, int32_t c,
                                const float *scale, const float *shift, const float *mean, const float *invstd, int32_t relu,
                                const double *local_sums, const double *global_sums, int64_t total_rows, void *workspace, float *dgamma,
                                float *dbeta, int32_t accumulate, void *dx, int odt, int32_t ld_dx, void *stream) {
  if (!dout || !x || !scale || !shift || !mean || !invstd || !local_sums || !global_sums || !workspace || !dgamma || !dbeta || !dx ||
      !red_shape_ok(rows, c, ld_x) || (ld_dout & 3) || (ld_dx & 3) || !dt_ok(ddt) || !dt_ok(xdt) || !dt_ok(odt) || !al4ch(dout, ddt) ||
      !al4ch(x, xdt) || !al4ch(dx, odt) || total_rows < rows)
    return PCP_ERR_ARG;
  if (dx == dout && odt != ddt) return PCP_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  float *coef = (float *)((double *)workspace + 2 * (size_t)c * RED_MAX_BLOCKS);
  hipLaunchKernelGGL(k_bnbwd_finalize_sync, dim3((c + 255) / 256), dim3(256), 0, s, local_sums, global_sums, (double)total_rows, c, dgamma, dbeta,
                     accumulate, coef);
  const long long total = (long long)rows * (c >> 2);
  launch_bnbwd_apply(ddt, xdt, odt, (unsigned)((total + 255) / 256), s, dout, ld_dout, x, ld_x, (long long)rows, c >> 2, c, scale, shift, mean,
                     invstd, relu, coef, dx, ld_dx);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

int colsum_impl(const void *x, int xdt, int64_t rows, int32_t c, int32_t ld, void *workspace, float *out, int32_t accumulate, void *stream) {
  if (!x || !workspace || !out || !red_shape_ok(rows, c, ld) || !dt_ok(xdt) || !al4ch(x, xdt)) return PCP_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  double *acc = (double *)workspace;
  RedParams p{};
  p.x = x; p.rows = rows; p.c = c; p.ld_x = ld; p.acc = acc;
  const int nb = red_grid(rows, c);
  launch_col_reduce<RED_COLSUM>(nb, s, p, xdt, xdt);
  hipLaunchKernelGGL(k_colsum_finalize, dim3((c + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_PARTS), 0, s, acc, nb, c, out, accumulate);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

int accumulate_impl(void *dst, int adt, int32_t ld_dst, const void *src, int bdt, int32_t ld_src, int64_t rows, int32_t c, float alpha,
                    void *stream) {
  if (!dst || !src || rows <= 0 || c < 4 || (c & 3) || (ld_dst & 3) || (ld_src & 3) || !dt_ok(adt) || !dt_ok(bdt) || !al4ch(dst, adt) ||
      !al4ch(src, bdt))
    return PCP_ERR_ARG;
  const long long total = (long long)rows * (c >> 2);
  const dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t s = (hipStream_t)stream;
#define PCP_ACC(AT, BT) hipLaunchKernelGGL((k_accumulate<AT, BT>), grid, dim3(256), 0, s, (AT *)dst, ld_dst, (const BT *)src, ld_src, (long long)rows, c >> 2, alpha)
  if (adt == PCP_DT_BF16) { if (bdt == PCP_DT_BF16) PCP_ACC(bf16_t, bf16_t); else PCP_ACC(bf16_t, float); }
  else { if (bdt == PCP_DT_BF16) PCP_ACC(float, bf16_t); else PCP_ACC(float, float); }
#undef PCP_ACC
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

int dilate2x_impl(const void *in, int dt, int32_t batch, int32_t h, int32_t w, int32_t c, int32_t ld_in, void *out, int32_t ld_out,
                  void *stream) {
  if (!in || !out || batch <= 0 || h <= 0 || w <= 0 || c < 4 || (c & 3) || (ld_in & 3) || (ld_out & 3) || !dt_ok(dt) || !al4ch(in, dt) ||
      !al4ch(out, dt))
    return PCP_ERR_ARG;
  const long long total = (long long)batch * 4 * h * w * (c >> 2);
  const dim3 grid((unsigned)((total + 255) / 256));
  if (dt == PCP_DT_BF16)
    hipLaunchKernelGGL((k_dilate2x<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t *)in, batch, h, w, c >> 2, ld_in, (bf16_t *)out, ld_out);
  else
    hipLaunchKernelGGL((k_dilate2x<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float *)in, batch, h, w, c >> 2, ld_in, (float *)out, ld_out);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

}  // namespace

extern "C" {

size_t pcp_bn_workspace_bytes(int32_t c) { return (size_t)RED_MAX_BLOCKS * c * 2 * sizeof(double) + (size_t)c * 2 * sizeof(float); }

// ---- fp32 entry points (include/pcp_hip_train.h) --------------------------------------------------------------------------------------
int pcp_bn_train_stats(const float *x, int64_t rows, int32_t c, int32_t ld, const float *gamma, const float *beta, float eps,
                       float momentum, float *running_mean, float *running_var, void *workspace, float *scale, float *shift,
                       float *mean, float *invstd, void *stream) {
  return bn_train_stats_impl(x, PCP_DT_F32, rows, c, ld, gamma, beta, eps, momentum, running_mean, running_var, workspace, scale, shift, mean,
                             invstd, stream);
}

int pcp_scale_shift_act(const float *x, int64_t rows, int32_t c, int32_t ld_x, const float *scale, const float *shift, int32_t relu,
                        float *out, int32_t ld_out, void *stream) {
  return scale_shift_act_impl(x, PCP_DT_F32, rows, c, ld_x, scale, shift, relu, out, PCP_DT_F32, ld_out, stream);
}

int pcp_bn_act_backward(const float *dout, int32_t ld_dout, const float *x, int32_t ld_x, int64_t rows, int32_t c, const float *scale,
                        const float *shift, const float *mean, const float *invstd, int32_t relu, void *workspace, float *dgamma,
                        float *dbeta, int32_t accumulate, float *dx, int32_t ld_dx, void *stream) {
  return bn_act_backward_impl(dout, PCP_DT_F32, ld_dout, x, PCP_DT_F32, ld_x, rows, c, scale, shift, mean, invstd, relu, workspace, dgamma,
                              dbeta, accumulate, dx, PCP_DT_F32, ld_dx, stream);
}

// ---- cross-rank BatchNorm (nn.SyncBatchNorm, tools/train.py --sync_bn): the two calls above split at their reduction ------------------
int pcp_bn_train_sums(const float *x, int64_t rows, int32_t c, int32_t ld, void *workspace, double *sums, void *stream) {
  return bn_train_sums_impl(x, PCP_DT_F32, rows, c, ld, workspace, sums, stream);
}

int pcp_bn_train_stats_from_sums(const double *sums, int64_t total_rows, int32_t c, const float *gamma, const float *beta, float eps,
                                 float momentum, float *running_mean, float *running_var, float *scale, float *shift, float *mean,
                                 float *invstd, void *stream) {
  if (!sums || !gamma || !beta || !scale || !shift || !mean || !invstd || total_rows <= 0 || c <= 0) return PCP_ERR_ARG;
  if ((running_mean == nullptr) != (running_var == nullptr)) return PCP_ERR_ARG;
  // sums is laid out like ONE block partial ([2][c]): the finalize kernel of the single-rank path reads it with nb = 1
  hipLaunchKernelGGL(k_bn_finalize, dim3((c + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_PARTS), 0, (hipStream_t)stream, sums, 1,
                     (long long)total_rows, c, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, mean, invstd);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

int pcp_bn_bwd_sums(const float *dout, int32_t ld_dout, const float *x, int32_t ld_x, int64_t rows, int32_t c, const float *scale,
                    const float *shift, const float *mean, const float *invstd, int32_t relu, void *workspace, double *sums, void *stream) {
  return bn_bwd_sums_impl(dout, PCP_DT_F32, ld_dout, x, PCP_DT_F32, ld_x, rows, c, scale, shift, mean, invstd, relu, workspace, sums, stream);
}

int pcp_bn_bwd_apply_from_sums(const float *dout, int32_t ld_dout, const float *x, int32_t ld_x, int64_t rows, int32_t c, const float *scale,
                               const float *shift, const float *mean, const float *invstd, int32_t relu, const double *local_sums,
                               const double *global_sums, int64_t total_rows, void *workspace, float *dgamma, float *dbeta,
                               int32_t accumulate, float *dx, int32_t ld_dx, void *stream) {
  return bn_bwd_apply_from_sums_impl(dout, PCP_DT_F32, ld_dout, x, PCP_DT_F32, ld_x, rows, c, scale, shift, mean, invstd, relu, local_sums,
                                     global_sums, total_rows, workspace, dgamma, dbeta, accumulate, dx, PCP_DT_F32, ld_dx, stream);
}

int pcp_colsum(const float *x, int64_t rows, int32_t c, int32_t ld, void *workspace, float *out, int32_t accumulate, void *stream) {
  return colsum_impl(x, PCP_DT_F32, rows, c, ld, workspace, out, accumulate, stream);
}

int pcp_accumulate(float *dst, int32_t ld_dst, const float *src, int32_t ld_src, int64_t rows, int32_t c, float alpha, void *stream) {
  return accumulate_impl(dst, PCP_DT_F32, ld_dst, src, PCP_DT_F32, ld_src, rows, c, alpha, stream);
}

int pcp_dilate2x(const float *in, int32_t batch, int32_t h, int32_t w, int32_t c, int32_t ld_in, float *out, int32_t ld_out,
                 void *stream) {
  return dilate2x_impl(in, PCP_DT_F32, batch, h, w, c, ld_in, out, ld_out, stream);
}

// ---- mixed-precision entry points (include/pcp_hip_mp.h): the same kernels with a storage type per tensor ---------------------------
int pcp_mp_bn_train_stats(const void *x, int32_t x_dtype, int64_t rows, int32_t c, int32_t ld, const float *gamma, const float *beta, float eps,
                          float momentum, float *running_mean, float *running_var, void *workspace, float *scale, float *shift,
                          float *mean, float *invstd, void *stream) {
  return bn_train_stats_impl(x, x_dtype, rows, c, ld, gamma, beta, eps, momentum, running_mean, running_var, workspace, scale, shift, mean,
                             invstd, stream);
}

int pcp_mp_scale_shift_act(const void *x, int32_t x_dtype, int64_t rows, int32_t c, int32_t ld_x, const float *scale, const float *shift,
                           int32_t relu, void *out, int32_t out_dtype, int32_t ld_out, void *stream) {
  return scale_shift_act_impl(x, x_dtype, rows, c, ld_x, scale, shift, relu, out, out_dtype, ld_out, stream);
}

int pcp_mp_bn_act_backward(const void *dout, int32_t dout_dtype, int32_t ld_dout, const void *x, int32_t x_dtype, int32_t ld_x, int64_t rows,
                           int32_t c, const float *scale, const float *shift, const float *mean, const float *invstd, int32_t relu,
                           void *workspace, float *dgamma, float *dbeta, int32_t accumulate, void *dx, int32_t dx_dtype, int32_t ld_dx,
                           void *stream) {
  return bn_act_backward_impl(dout, dout_dtype, ld_dout, x, x_dtype, ld_x, rows, c, scale, shift, mean, invstd, relu, workspace, dgamma, dbeta,
                              accumulate, dx, dx_dtype, ld_dx, stream);
}

int pcp_mp_bn_train_sums(const void *x, int32_t x_dtype, int64_t rows, int32_t c, int32_t ld, void *workspace, double *sums, void *stream) {
  return bn_train_sums_impl(x, x_dtype, rows, c, ld, workspace, sums, stream);
}

int pcp_mp_bn_bwd_sums(const void *dout, int32_t dout_dtype, int32_t ld_dout, const void *x, int32_t x_dtype, int32_t ld_x, int64_t rows,
                       int32_t c, const float *scale, const float *shift, const float *mean, const float *invstd, int32_t relu,
                       void *workspace, double *sums, void *stream) {
  return bn_bwd_sums_impl(dout, dout_dtype, ld_dout, x, x_dtype, ld_x, rows, c, scale, shift, mean, invstd, relu, workspace, sums, stream);
}

int pcp_mp_bn_bwd_apply_from_sums(const void *dout, int32_t dout_dtype, int32_t ld_dout, const void *x, int32_t x_dtype, int32_t ld_x,
                                  int64_t rows, int32_t c, const float *scale, const float *shift, const float *mean, const float *invstd,
                                  int32_t relu, const double *local_sums, const double *global_sums, int64_t total_rows, void *workspace,
                                  float *dgamma, float *dbeta, int32_t accumulate, void *dx, int32_t dx_dtype, int32_t ld_dx, void *stream) {
  return bn_bwd_apply_from_sums_impl(dout, dout_dtype, ld_dout, x, x_dtype, ld_x, rows, c, scale, shift, mean, invstd, relu, local_sums,
                                     global_sums, total_rows, workspace, dgamma, dbeta, accumulate, dx, dx_dtype, ld_dx, stream);
}

int pcp_mp_colsum(const void *x, int32_t x_dtype, int64_t rows, int32_t c, int32_t ld, void *workspace, float *out, int32_t accumulate,
                  void *stream) {
  return colsum_impl(x, x_dtype, rows, c, ld, workspace, out, accumulate, stream);
}

int pcp_mp_accumulate(void *dst, int32_t dst_dtype, int32_t ld_dst, const void *src, int32_t src_dtype, int32_t ld_src, int64_t rows,
                      int32_t c, float alpha, void *stream) {
  return accumulate_impl(dst, dst_dtype, ld_dst, src, src_dtype, ld_src, rows, c, alpha, stream);
}

int pcp_mp_dilate2x(const void *in, int32_t dtype, int32_t batch, int32_t h, int32_t w, int32_t c, int32_t ld_in, void *out, int32_t ld_out,
                    void *stream) {
  return dilate2x_impl(in, dtype, batch, h, w, c, ld_in, out, ld_out, stream);
}

}  // extern "C"
