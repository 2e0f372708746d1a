// a14 -- HunterJr point head, fused: bilinear sampling of the BEV map at every point -> feature MLP (C -> 32 -> C, Linear+BN+ReLU
// twice) + residual -> the three point heads (C -> 3 | 3 | 2) in ONE kernel.
//
// Reference: pcdet/models/bev_layers/hunter_toolbox.py:8-39,94-127 (interpolate_points_feat_from_bev_img) and
// pcdet/models/bev_layers/hunter_jr.py:78-101 (HunterPointHead.forward).  Unfused this is five launches that stream the
// (N, 384) fp32 point-feature matrix (369 MB at 4 x 60k points) through HBM six times (0.96 ms); fused, a workgroup keeps its
// 32 points' features in LDS: sampled rows are written to HBM once (the scatter-mean needs them later), the hidden and final
// activations never leave the CU, and only (N, 8) head values come out.
//   phase 1  per-point corner indices / weights, then the 4-row gather + blend -> LDS tile F[32][C] (+ global pf)
//   phase 2  H1 = relu(F W1^T + b1)      fp32 MFMA, K = C split over the 4 waves, partials reduced through LDS
//   phase 3  F <- relu(H1 W2^T + b2) + F  fp32 MFMA, 3 column tiles per wave, in place (final features)
//   phase 4  head8 = F Wh^T + bh         one dot product per thread (VALU)
// Sampling arithmetic is bitwise that of k_bilinear / the reference (products and sums in its order, no FMA contraction).
#include "pcp_common.h"

namespace {

constexpr int PH_BM = 32;           // points per workgroup
constexpr int PH_C = 384;           // BEV channels (num_bev_features of the five configs)
constexpr int PH_H = 32;            // hidden width (POINT_HEAD_HIDDEN_CHANNELS: [32])
constexpr int PH_LDF = PH_C + 4;    // padded row: a ds_read_b128 group's 16 lanes hit distinct slots
constexpr int PH_LDH = PH_H + 4;
constexpr int PH_NOUT = 8;

struct PointHeadParams {
  const float *bev;
  int batch, h, w, ld_bev;
  const float *points;
  long long n;
  int stride;
  float min_x, min_y, pix_x, pix_y;
  const float *w1, *b1;   // [32][C], [32]
  const float *w2, *b2;   // [C][32], [C]
  const float *wh, *bh;   // [8][C], [8]
  float *pf;              // (n, ld_pf) sampled features
  int ld_pf;
  float *head;            // (n, 8)
  const int *order;       // optional: process rows order[0 .. *order_count) (results still land at the ORIGINAL row index)
  const int *order_count; // device scalar, NULL: n
  // optional fusion of hunter_jr.py:257-265 + the re-sampling of the corrected points (correct_bev_image): rows predicted dynamic
  // foreground get xyz += flow IN PLACE and their pf row re-sampled at the corrected location
  float *points_mut;      // NULL: disabled
  float flow_thresh;
  unsigned char *dyn_mask;
};

// blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a CONTIGUOUS run of point tiles, so that in the spatially
// sorted visiting order each XCD's 4 MiB L2 holds only its own stripe of the BEV map (bijective remap)
__device__ __forceinline__ long long xcd_remap_ph(long long bid, long long nwg) {
  const long long q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

__device__ __forceinline__ f32x16 mfma_ph(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__global__ __launch_bounds__(256, 2) void k_point_head(PointHeadParams p) {
  __shared__ __attribute__((aligned(16))) float Fs[PH_BM * PH_LDF];
  __shared__ __attribute__((aligned(16))) float H1s[PH_BM * PH_LDH];
  __shared__ float red[4][PH_BM][PH_H + 1];
  __shared__ long long c_off[PH_BM][4];     // float offsets of the four corner rows (Ia, Ib, Ic, Id)
  __shared__ float c_w[PH_BM][4];
  __shared__ int c_ok[PH_BM];
  __shared__ long long c_row[PH_BM];        // original row index of the tile's points (-1: none)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const long long i0 = (p.order ? xcd_remap_ph(blockIdx.x, gridDim.x) : (long long)blockIdx.x) * PH_BM;
  const long long n_rows = p.order ? (p.order_count ? (long long)*p.order_count : p.n) : p.n;
  if (i0 >= n_rows) return;

  // ---- phase 1a: corners and weights (hunter_toolbox.py:19-37), no FMA contraction ---------------------------------------
  if (tid < PH_BM) {
#pragma clang fp contract(off)
    const long long t = i0 + tid;
    const long long i = t < n_rows ? (p.order ? (long long)p.order[t] : t) : -1;
    c_row[tid] = i;
    int ok = 0;
    // masked rows gather the map's first row (valid memory) with zero weights: the gather loop below has no data-dependent
    // control flow, so all 16 loads of an unroll group are in flight together
#pragma unroll
    for (int k = 0; k < 4; ++k) { c_off[tid][k] = 0; c_w[tid][k] = 0.f; }
    if (i >= 0) {
      const float *row = p.points + i * p.stride;
      int b = (int)row[0];
      if (b >= 0 && b < p.batch) {
        ok = 1;
        float x = __fdiv_rn(row[1] - p.min_x, p.pix_x), y = __fdiv_rn(row[2] - p.min_y, p.pix_y);
        float fx0 = fminf(fmaxf(floorf(x), -2.0f), (float)p.w + 1.0f), fy0 = fminf(fmaxf(floorf(y), -2.0f), (float)p.h + 1.0f);
        int x0 = (int)fx0, y0 = (int)fy0;
        int x1 = x0 + 1, y1 = y0 + 1;
        x0 = min(max(x0, 0), p.w - 1); x1 = min(max(x1, 0), p.w - 1);
        y0 = min(max(y0, 0), p.h - 1); y1 = min(max(y1, 0), p.h - 1);
        c_w[tid][0] = ((float)x1 - x) * ((float)y1 - y);
        c_w[tid][1] = ((float)x1 - x) * (y - (float)y0);
        c_w[tid][2] = (x - (float)x0) * ((float)y1 - y);
        c_w[tid][3] = (x - (float)x0) * (y - (float)y0);
        const long long img = (long long)b * p.h * p.w;
        c_off[tid][0] = (img + (long long)y0 * p.w + x0) * p.ld_bev;
        c_off[tid][1] = (img + (long long)y1 * p.w + x0) * p.ld_bev;
        c_off[tid][2] = (img + (long long)y0 * p.w + x1) * p.ld_bev;
        c_off[tid][3] = (img + (long long)y1 * p.w + x1) * p.ld_bev;
      }
    }
    c_ok[tid] = ok;
  }
  __syncthreads();
  // ---- phase 1b: gather + blend, 32 points x 96 float4 --------------------------------------------------------------------
  {
#pragma clang fp contract(off)
    // 12 (point, float4) items per thread; the 16 corner loads of four items are issued before any is consumed (the kernel was
    // latency bound with 4 loads in flight per thread: 3.2 TB/s of L2-resident gathers)
    constexpr int ITEMS = PH_BM * (PH_C / 4) / 256;      // 12
    static_assert(ITEMS % 4 == 0, "unroll group");
    for (int it = 0; it < ITEMS; it += 4) {
      f32x4 Ia[4], Ib[4], Ic[4], Id[4];
      int pts[4], qs[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = tid + (it + u) * 256;
        pts[u] = idx / (PH_C / 4);
        qs[u] = idx % (PH_C / 4);
        const int pt = pts[u];
        const float *base = p.bev + qs[u] * 4;
        Ia[u] = *reinterpret_cast<const f32x4 *>(base + c_off[pt][0]);
        Ib[u] = *reinterpret_cast<const f32x4 *>(base + c_off[pt][1]);
        Ic[u] = *reinterpret_cast<const f32x4 *>(base + c_off[pt][2]);
        Id[u] = *reinterpret_cast<const f32x4 *>(base + c_off[pt][3]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pt = pts[u], q = qs[u];
        const float wa = c_w[pt][0], wb = c_w[pt][1], wc = c_w[pt][2], wd = c_w[pt][3];
        f32x4 v;
        v.x = Ia[u].x * wa + Ib[u].x * wb + Ic[u].x * wc + Id[u].x * wd;
        v.y = Ia[u].y * wa + Ib[u].y * wb + Ic[u].y * wc + Id[u].y * wd;
        v.z = Ia[u].z * wa + Ib[u].z * wb + Ic[u].z * wc + Id[u].z * wd;
        v.w = Ia[u].w * wa + Ib[u].w * wb + Ic[u].w * wc + Id[u].w * wd;
        const bool okp = c_ok[pt] != 0;                        // select, not a branch (0 * inf must not leak a NaN)
        v.x = okp ? v.x : 0.f; v.y = okp ? v.y : 0.f; v.z = okp ? v.z : 0.f; v.w = okp ? v.w : 0.f;
        if (c_row[pt] >= 0) *reinterpret_cast<f32x4 *>(p.pf + c_row[pt] * p.ld_pf + q * 4) = v;   // rows of foreign frames stay 0
        *reinterpret_cast<f32x4 *>(Fs + pt * PH_LDF + q * 4) = v;
      }
    }
  }
  __syncthreads();

#ifndef PH_DIAG_GATHER_ONLY
  // ---- phase 2: H1 partial over this wave's K quarter (96 channels = 12 groups of 8) -----------------------------------------
  {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.f;
    const float *asrc = Fs + r * PH_LDF + wave * 96 + 4 * h;
    const float *bsrc = p.w1 + r * PH_C + wave * 96 + 4 * h;          // B[k][n = r] = W1[r][k]
#pragma unroll 4
    for (int g = 0; g < 12; g++) {
      f32x4 a = *reinterpret_cast<const f32x4 *>(asrc + g * 8);
      f32x4 bq = *reinterpret_cast<const f32x4 *>(bsrc + g * 8);
      acc = mfma_ph(a.x, bq.x, acc);
      acc = mfma_ph(a.y, bq.y, acc);
      acc = mfma_ph(a.z, bq.z, acc);
      acc = mfma_ph(a.w, bq.w, acc);
    }
#pragma unroll
    for (int e = 0; e < 16; e++) red[wave][(e & 3) + 8 * (e >> 2) + 4 * h][r] = acc[e];
  }
  __syncthreads();
  for (int idx = tid; idx < PH_BM * PH_H; idx += 256) {
    const int m = idx / PH_H, nn = idx % PH_H;
    float v = red[0][m][nn] + red[1][m][nn] + red[2][m][nn] + red[3][m][nn] + p.b1[nn];
    H1s[m * PH_LDH + nn] = fmaxf(v, 0.f);
  }
  __syncthreads();

#ifndef PH_DIAG_SKIP34
  // ---- phase 3: final = relu(H1 W2^T + b2) + F, three 32-channel column tiles per wave, in place -----------------------------
  {
    f32x4 a[4];
#pragma unroll
    for (int g = 0; g < 4; g++) a[g] = *reinterpret_cast<const f32x4 *>(H1s + r * PH_LDH + g * 8 + 4 * h);
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int ct = wave * 3 + j;
      const float *bsrc = p.w2 + (ct * 32 + r) * PH_H + 4 * h;         // B[k][n] = W2[n][k]
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; e++) acc[e] = 0.f;
#pragma unroll
      for (int g = 0; g < 4; g++) {
        f32x4 bq = *reinterpret_cast<const f32x4 *>(bsrc + g * 8);
        acc = mfma_ph(a[g].x, bq.x, acc);
        acc = mfma_ph(a[g].y, bq.y, acc);
        acc = mfma_ph(a[g].z, bq.z, acc);
        acc = mfma_ph(a[g].w, bq.w, acc);
      }
      const int n = ct * 32 + r;
      const float bias = p.b2[n];
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int m = (e & 3) + 8 * (e >> 2) + 4 * h;
        float *f = Fs + m * PH_LDF + n;
        *f = fmaxf(acc[e] + bias, 0.f) + *f;
      }
    }
  }
  __syncthreads();

#endif
#if !defined(PH_DIAG_SKIP34) && !defined(PH_DIAG_SKIP4)
  // ---- phase 4: the three heads as one (8 x C) matrix, as an MFMA GEMM with N padded 8 -> 32 (lanes r >= 8 feed zeros): K = C split
  //      over the four waves like phase 2, partials reduced through LDS.  (A per-thread 384-long dot product cost 250 us per launch:
  //      8 threads walking each LDS row serialise on the banks.)
  {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.f;
    const float *asrc = Fs + r * PH_LDF + wave * 96 + 4 * h;
    const float *bsrc = p.wh + (r < PH_NOUT ? r : 0) * PH_C + wave * 96 + 4 * h;      // B[k][n = r] = Wh[r][k]
    const float bmask = r < PH_NOUT ? 1.f : 0.f;
#pragma unroll 4
    for (int g = 0; g < 12; g++) {
      f32x4 a = *reinterpret_cast<const f32x4 *>(asrc + g * 8);
      f32x4 bq = *reinterpret_cast<const f32x4 *>(bsrc + g * 8);
      acc = mfma_ph(a.x, bq.x * bmask, acc);
      acc = mfma_ph(a.y, bq.y * bmask, acc);
      acc = mfma_ph(a.z, bq.z * bmask, acc);
      acc = mfma_ph(a.w, bq.w * bmask, acc);
    }
    if (r < PH_NOUT) {
#pragma unroll
      for (int e = 0; e < 16; e++) red[wave][(e & 3) + 8 * (e >> 2) + 4 * h][r] = acc[e];
    }
  }
  __syncthreads();
  {
    const int pt = tid >> 3, o = tid & 7;
    const float v = red[0][pt][o] + red[1][pt][o] + red[2][pt][o] + red[3][pt][o] + p.bh[o];
    if (c_row[pt] >= 0) p.head[c_row[pt] * PH_NOUT + o] = v;
    if (p.points_mut) H1s[pt * PH_LDH + o] = v;               // H1s is free now: keep the 8 head values for the flow stage
  }
  if (p.points_mut) {
    // ---- phase 5 (optional): dynamic-foreground correction + re-sampling, same arithmetic as k_apply_flow / k_bilinear ----------
    __shared__ int dyn_list[PH_BM];
    __shared__ int n_dyn;
    if (tid == 0) n_dyn = 0;
    __syncthreads();
    if (tid < PH_BM) {
#pragma clang fp contract(off)
      const long long i = c_row[tid];
      if (i >= 0) {
        const float *hv = H1s + tid * PH_LDH;
        const float p0 = 1.0f / (1.0f + expf(-hv[0])), p1 = 1.0f / (1.0f + expf(-hv[1])), p2 = 1.0f / (1.0f + expf(-hv[2]));
        const bool dyn = (p2 > p0) && (p2 > p1) && (p2 > p.flow_thresh);
        if (p.dyn_mask) p.dyn_mask[i] = dyn ? 1 : 0;
        if (dyn) {
          float *row = p.points_mut + i * p.stride;
          const float nx = row[1] + hv[3], ny = row[2] + hv[4];
          row[1] = nx;
          row[2] = ny;
          row[3] = row[3] + hv[5];
          const int b = (int)row[0];
          if (b >= 0 && b < p.batch) {
            const float x = __fdiv_rn(nx - p.min_x, p.pix_x), y = __fdiv_rn(ny - p.min_y, p.pix_y);
            const float fx0 = fminf(fmaxf(floorf(x), -2.0f), (float)p.w + 1.0f), fy0 = fminf(fmaxf(floorf(y), -2.0f), (float)p.h + 1.0f);
            int x0 = (int)fx0, y0 = (int)fy0;
            int x1 = x0 + 1, y1 = y0 + 1;
            x0 = min(max(x0, 0), p.w - 1); x1 = min(max(x1, 0), p.w - 1);
            y0 = min(max(y0, 0), p.h - 1); y1 = min(max(y1, 0), p.h - 1);
            c_w[tid][0] = ((float)x1 - x) * ((float)y1 - y);
            c_w[tid][1] = ((float)x1 - x) * (y - (float)y0);
            c_w[tid][2] = (x - (float)x0) * ((float)y1 - y);
            c_w[tid][3] = (x - (float)x0) * (y - (float)y0);
            const long long img = (long long)b * p.h * p.w;
            c_off[tid][0] = (img + (long long)y0 * p.w + x0) * p.ld_bev;
            c_off[tid][1] = (img + (long long)y1 * p.w + x0) * p.ld_bev;
            c_off[tid][2] = (img + (long long)y0 * p.w + x1) * p.ld_bev;
            c_off[tid][3] = (img + (long long)y1 * p.w + x1) * p.ld_bev;
            dyn_list[atomicAdd(&n_dyn, 1)] = tid;
          }
        }
      }
    }
    __syncthreads();
    {
#pragma clang fp contract(off)
      const int items = n_dyn * (PH_C / 4);
      for (int idx = tid; idx < items; idx += 256) {
        const int pt = dyn_list[idx / (PH_C / 4)], q = idx % (PH_C / 4);
        const float *base = p.bev + q * 4;
        const f32x4 Ia = *reinterpret_cast<const f32x4 *>(base + c_off[pt][0]);
        const f32x4 Ib = *reinterpret_cast<const f32x4 *>(base + c_off[pt][1]);
        const f32x4 Ic = *reinterpret_cast<const f32x4 *>(base + c_off[pt][2]);
        const f32x4 Id = *reinterpret_cast<const f32x4 *>(base + c_off[pt][3]);
        const float wa = c_w[pt][0], wb = c_w[pt][1], wc = c_w[pt][2], wd = c_w[pt][3];
        f32x4 v;
        v.x = Ia.x * wa + Ib.x * wb + Ic.x * wc + Id.x * wd;
        v.y = Ia.y * wa + Ib.y * wb + Ic.y * wc + Id.y * wd;
        v.z = Ia.z * wa + Ib.z * wb + Ic.z * wc + Id.z * wd;
        v.w = Ia.w * wa + Ib.w * wb + Ic.w * wc + Id.w * wd;
        *reinterpret_cast<f32x4 *>(p.pf + c_row[pt] * p.ld_pf + q * 4) = v;
      }
    }
  }
#endif
#endif  // PH_DIAG_GATHER_ONLY
}

}  // namespace

static int point_head_launch(const float *bev, int32_t batch, int32_t h, int32_t w, int32_t c, int32_t ld_bev,
                             const float *points, int64_t n, int32_t row_stride, float min_x, float min_y, float pix_x,
                             float pix_y, const float *w1, const float *b1, const float *w2, const float *b2,
                             const float *wh, const float *bh, int32_t hidden, int32_t n_out, float *pf, int32_t ld_pf,
                             float *head, const int32_t *order, const int32_t *order_count, float *points_mut, float flow_thresh,
                             unsigned char *dyn_mask, void *stream_) {
  if (!bev || !w1 || !b1 || !w2 || !b2 || !wh || !bh || !pf || !head || n < 0 || batch <= 0 || h <= 0 || w <= 0) return PCP_ERR_ARG;
  if (c != PH_C || hidden != PH_H || n_out != PH_NOUT) return PCP_ERR_UNSUPPORTED;
  if ((ld_bev & 3) || (ld_pf & 3) || row_stride < 3 || (((uintptr_t)bev) & 15) || (((uintptr_t)pf) & 15)) return PCP_ERR_ARG;
  if (n == 0) return PCP_OK;
  if (!points) return PCP_ERR_ARG;
  PointHeadParams p;
  p.bev = bev; p.batch = batch; p.h = h; p.w = w; p.ld_bev = ld_bev;
  p.points = points; p.n = n; p.stride = row_stride;
  p.min_x = min_x; p.min_y = min_y; p.pix_x = pix_x; p.pix_y = pix_y;
  p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.wh = wh; p.bh = bh;
  p.pf = pf; p.ld_pf = ld_pf; p.head = head;
  p.order = order; p.order_count = order_count;
  p.points_mut = points_mut; p.flow_thresh = flow_thresh; p.dyn_mask = dyn_mask;
  long long blocks = (n + PH_BM - 1) / PH_BM;
  hipLaunchKernelGGL(k_point_head, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, p);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

extern "C" int pcp_hunter_point_head(const float *bev, int32_t batch, int32_t h, int32_t w, int32_t c, int32_t ld_bev,
                                     const float *points, int64_t n, int32_t row_stride, float min_x, float min_y, float pix_x,
                                     float pix_y, const float *w1, const float *b1, const float *w2, const float *b2,
                                     const float *wh, const float *bh, int32_t hidden, int32_t n_out, float *pf, int32_t ld_pf,
                                     float *head, void *stream_) {
  return point_head_launch(bev, batch, h, w, c, ld_bev, points, n, row_stride, min_x, min_y, pix_x, pix_y, w1, b1, w2, b2, wh, bh, hidden,
                           n_out, pf, ld_pf, head, nullptr, nullptr, nullptr, 0.f, nullptr, stream_);
}

extern "C" int pcp_hunter_point_head_ex(const float *bev, int32_t batch, int32_t h, int32_t w, int32_t c, int32_t ld_bev, float *points,
                                        int64_t n, int32_t row_stride, float min_x, float min_y, float pix_x, float pix_y, const float *w1,
                                        const float *b1, const float *w2, const float *b2, const float *wh, const float *bh, int32_t hidden,
                                        int32_t n_out, float *pf, int32_t ld_pf, float *head, const int32_t *order,
                                        const int32_t *order_count, int32_t apply_flow, float flow_thresh, uint8_t *dyn_mask, void *stream_) {
  return point_head_launch(bev, batch, h, w, c, ld_bev, points, n, row_stride, min_x, min_y, pix_x, pix_y, w1, b1, w2, b2, wh, bh, hidden,
                           n_out, pf, ld_pf, head, order, order_count, apply_flow ? points : nullptr, flow_thresh, dyn_mask, stream_);
}
