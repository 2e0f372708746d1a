// a12 / a14 -- 3x3 stride-1 convolution of the wide layers (cin >= 256: HunterJr conv_input / conv_weightor, DiscoNet compressor /
// decompressor) as Winograd F(4x4, 3x3) in three launches, fp32 throughout (fp32 MFMA, fp32 accumulation).
//
// Why not fused like wino.hip: F(4x4,3x3) keeps 36 product matrices alive per tile, so a workgroup's accumulators (the whole
// 512-KiB register file of a CU) cover only 64 tiles x 32 channels; at that tile the kernel re-reads inputs and weights from L2
// 3.6x faster than wino.hip does per unit of matrix-pipe time, i.e. beyond what the L2s deliver.  Going through memory instead costs
// ~1.3 GB of extra HBM / Infinity-Cache traffic for the 768 -> 768 layer at 4 frames (~0.3 ms) and removes 1.78x of the
// multiplies (4x fewer than the direct convolution):
//
//   k_w4_input   V[p][t][c]  = (B^T d B)[p]     one thread = one 6x6 input patch x 4 channels; coalesced along channels
//   k_w4_gemm    M[p][t][n]  = sum_c V[p][t][c] U[p][n][c]   36 independent GEMMs [tiles x cin] x [cin x cout] on
//                v_mfma_f32_32x32x2_f32: 128 x 128 macro tile per 4-wave workgroup (64 x 64 per wave), 32-deep K slices through
//                double-buffered LDS (rows padded to 36 floats: every ds_read_b128 / ds_write_b128 group is conflict free),
//                global -> register prefetch of slice s+1 under the multiply of slice s, one barrier per slice, 2 workgroups
//                per CU; workgroup order is XCD-contiguous with the N tile fastest, so the 6 workgroups sharing a V tile run
//                together and a position's U (2.4 MB) stays in that XCD's L2
//   k_w4_output  out = A^T M A + bias (ReLU)    one thread = one 4x4 output patch x 4 channels
//
// Transform points 0, +-1, +-2, inf (Lavin & Gray); measured error vs float64 on the 768-channel layer: 2e-5 of the output scale
// (F(2x2): 1e-6; parity bar 1e-3).
#include "pcp_common.h"

namespace {

constexpr int G_BM = 128, G_BN = 128, G_KS = 32, G_LD = 36, G_THREADS = 256;
constexpr int G_TILE_FLOATS = G_BM * G_LD;
constexpr int G_PERSISTENT_WGS = 512;       // 256 CUs x 2 resident workgroups

struct W4Geom {
  int batch, h, w;
  int tiles_x, tiles_y;
  long long tiles, m_pad;
  int cin, cout, n_pad;
  int ld_in, ld_out, relu;
};

__device__ __forceinline__ int xcd_remap_g(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// B^T x for the 6-point transform (Lavin & Gray, F(4,3))
__device__ __forceinline__ void bt6(const f32x4 d0, const f32x4 d1, const f32x4 d2, const f32x4 d3, const f32x4 d4, const f32x4 d5,
                                    f32x4 (&t)[6]) {
  const f32x4 p = d4 - 4.f * d2, q = d3 - 4.f * d1;
  const f32x4 r = d4 - d2, s = 2.f * (d3 - d1);
  t[0] = 4.f * d0 - 5.f * d2 + d4;
  t[1] = p + q;
  t[2] = p - q;
  t[3] = r + s;
  t[4] = r - s;
  t[5] = 4.f * d1 - 5.f * d3 + d5;
}

__global__ __launch_bounds__(256) void k_w4_input(const float *__restrict__ in, float *__restrict__ v, W4Geom g) {
  const int cq_n = g.cin >> 2;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long t = idx / cq_n;
  const int cq = (int)(idx - t * cq_n);
  if (t >= g.tiles) return;
  const int tx = (int)(t % g.tiles_x);
  const long long r0 = t / g.tiles_x;
  const int ty = (int)(r0 % g.tiles_y);
  const int b = (int)(r0 / g.tiles_y);
  const int y0 = ty * 4 - 1, x0 = tx * 4 - 1;
  const float *src = in + cq * 4;
  f32x4 c[6][6];                      // c[i][col]: column-transformed patch
#pragma unroll
  for (int col = 0; col < 6; col++) {
    f32x4 d[6];
    const int x = x0 + col;
#pragma unroll
    for (int a = 0; a < 6; a++) {
      const int y = y0 + a;
      d[a] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (y >= 0 && y < g.h && x >= 0 && x < g.w)
        d[a] = *reinterpret_cast<const f32x4 *>(src + ((long long)(b * g.h + y) * g.w + x) * g.ld_in);
    }
    f32x4 tt[6];
    bt6(d[0], d[1], d[2], d[3], d[4], d[5], tt);
#pragma unroll
    for (int i = 0; i < 6; i++) c[i][col] = tt[i];
  }
  float *dst = v + t * g.cin + cq * 4;
  const long long pstride = g.m_pad * g.cin;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    f32x4 o[6];
    bt6(c[i][0], c[i][1], c[i][2], c[i][3], c[i][4], c[i][5], o);
#pragma unroll
    for (int j = 0; j < 6; j++) *reinterpret_cast<f32x4 *>(dst + (i * 6 + j) * pstride) = o[j];
  }
}

// A^T m: 6 -> 4
__device__ __forceinline__ void at6(const f32x4 m0, const f32x4 m1, const f32x4 m2, const f32x4 m3, const f32x4 m4, const f32x4 m5,
                                    f32x4 (&y)[4]) {
  const f32x4 s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
  y[0] = m0 + s12 + s34;
  y[1] = d12 + 2.f * d34;
  y[2] = s12 + 4.f * s34;
  y[3] = d12 + 8.f * d34 + m5;
}

__global__ __launch_bounds__(256) void k_w4_output(const float *__restrict__ m, const float *__restrict__ bias, float *__restrict__ out,
                                                   W4Geom g) {
  const int nq_n = g.cout >> 2;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long t = idx / nq_n;
  const int nq = (int)(idx - t * nq_n);
  if (t >= g.tiles) return;
  const int tx = (int)(t % g.tiles_x);
  const long long r0 = t / g.tiles_x;
  const int ty = (int)(r0 % g.tiles_y);
  const int b = (int)(r0 / g.tiles_y);
  const float *src = m + t * g.n_pad + nq * 4;
  const long long pstride = g.m_pad * g.n_pad;
  f32x4 u[4][6];                      // u[k][j] = (A^T M)[k][j]
#pragma unroll
  for (int j = 0; j < 6; j++) {
    f32x4 col[6];
#pragma unroll
    for (int i = 0; i < 6; i++) col[i] = *reinterpret_cast<const f32x4 *>(src + (i * 6 + j) * pstride);
    f32x4 y[4];
    at6(col[0], col[1], col[2], col[3], col[4], col[5], y);
#pragma unroll
    for (int k = 0; k < 4; k++) u[k][j] = y[k];
  }
  const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + nq * 4);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    f32x4 y[4];
    at6(u[k][0], u[k][1], u[k][2], u[k][3], u[k][4], u[k][5], y);
    const int oy = ty * 4 + k;
    if (oy >= g.h) continue;
#pragma unroll
    for (int l = 0; l < 4; l++) {
      const int ox = tx * 4 + l;
      if (ox >= g.w) continue;
      f32x4 val = y[l] + bv;
      if (g.relu) {
        val.x = fmaxf(val.x, 0.f);
        val.y = fmaxf(val.y, 0.f);
        val.z = fmaxf(val.z, 0.f);
        val.w = fmaxf(val.w, 0.f);
      }
      *reinterpret_cast<f32x4 *>(out + ((long long)(b * g.h + oy) * g.w + ox) * g.ld_out + nq * 4) = val;
    }
  }
}

struct GemmParams {
  const float *a;     // [P][m_pad][k]
  const float *b;     // [P][n_pad][k]
  float *c;           // [P][m_pad][n_pad]
  long long m_pad;
  int n_pad, k;
  int m_tiles, n_tiles;
};

__device__ __forceinline__ f32x16 mfma32g(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

__global__ __launch_bounds__(G_THREADS, 2) void k_w4_gemm(GemmParams p) {
  __shared__ __attribute__((aligned(16))) float lds[4 * G_TILE_FLOATS];          // A[2] | B[2]
  float *as = lds, *bs = lds + 2 * G_TILE_FLOATS;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;

  // Persistent workgroups: XCD x (= blockIdx & 7; blocks are dealt round-robin over the XCDs) owns the contiguous tile range
  // [x * per_xcd, (x + 1) * per_xcd) and its workgroups take tiles j, j + wgs_per_xcd, ... of it: at any time one XCD works on
  // consecutive tiles (N tile fastest: the workgroups sharing a V tile run together, a position's U stays in that L2), and the
  // (tile, K slice) sequence of a workgroup is ONE software pipeline -- the first slice of the next tile is fetched under the last
  // multiply of the current one, the accumulators are stored while the next tile's slices are already in flight.
  const int total = 36 * p.m_tiles * p.n_tiles;
  const int xcd = blockIdx.x & 7, wgs_per_xcd = (gridDim.x + 7 - xcd) >> 3, j0 = blockIdx.x >> 3;
  const int per_xcd = (total + 7) >> 3;
  const int t_begin = xcd * per_xcd, t_end = min(total, t_begin + per_xcd);
  int tile = t_begin + j0;
  if (tile >= t_end) return;

  const float *ga, *gb;
  auto tile_ptrs = [&](int t) {
    const int nt = t % p.n_tiles;
    const int q = t / p.n_tiles;
    const int mt = q % p.m_tiles;
    const int pos = q / p.m_tiles;
    ga = p.a + ((long long)pos * p.m_pad + (long long)mt * G_BM) * p.k;        // workgroup-uniform
    gb = p.b + ((long long)pos * p.n_pad + (long long)nt * G_BN) * p.k;
  };
  // staging: thread -> (row = tid >> 3 (+32 i), 16-byte column tid & 7): 8 lanes cover one 128-byte row segment
  const int s_row = tid >> 3, s_c4 = tid & 7;
  unsigned goff[4];
  int soff[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    goff[i] = (unsigned)(((s_row + 32 * i) * p.k + s_c4 * 4) * 4);
    soff[i] = (s_row + 32 * i) * G_LD + s_c4 * 4;
  }
  f32x4 ra[4], rb[4];
  auto g_load = [&](int s) {
    const char *ba = reinterpret_cast<const char *>(ga + s * G_KS);
    const char *bb = reinterpret_cast<const char *>(gb + s * G_KS);
#pragma unroll
    for (int i = 0; i < 4; i++) ra[i] = *reinterpret_cast<const f32x4 *>(ba + goff[i]);
#pragma unroll
    for (int i = 0; i < 4; i++) rb[i] = *reinterpret_cast<const f32x4 *>(bb + goff[i]);
  };
  auto s_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; i++) *reinterpret_cast<f32x4 *>(as + buf * G_TILE_FLOATS + soff[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < 4; i++) *reinterpret_cast<f32x4 *>(bs + buf * G_TILE_FLOATS + soff[i]) = rb[i];
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][c][e] = 0.f;

  const int a_off = (wm * 64 + r) * G_LD + 4 * h;
  const int b_off = (wn * 64 + r) * G_LD + 4 * h;
  auto multiply = [&](int buf) {
    const float *pa = as + buf * G_TILE_FLOATS + a_off;
    const float *pb = bs + buf * G_TILE_FLOATS + b_off;
#pragma unroll
    for (int j = 0; j < G_KS / 8; j++) {
      f32x4 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; i++) af[i] = *reinterpret_cast<const f32x4 *>(pa + i * 32 * G_LD + j * 8);
#pragma unroll
      for (int c = 0; c < 2; c++) bf[c] = *reinterpret_cast<const f32x4 *>(pb + c * 32 * G_LD + j * 8);
      // k component outermost: four independent accumulators between two MFMAs on the same one
#pragma unroll
      for (int kk = 0; kk < 4; kk++)
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int c = 0; c < 2; c++) acc[i][c] = mfma32g(af[i][kk], bf[c][kk], acc[i][c]);
    }
  };
  auto store_tile = [&](int t) {
    const int nt = t % p.n_tiles;
    const int q = t / p.n_tiles;
    const int mt = q % p.m_tiles;
    const int pos = q / p.m_tiles;
    float *gc = p.c + ((long long)pos * p.m_pad + (long long)mt * G_BM + wm * 64) * p.n_pad + nt * G_BN + wn * 64 + r;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int c = 0; c < 2; c++)
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const int row = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          gc[(long long)row * p.n_pad + c * 32] = acc[i][c][e];
          acc[i][c][e] = 0.f;
        }
  };

  const int n_slices = p.k / G_KS;
  tile_ptrs(tile);
  g_load(0);
  s_store(0);
  __syncthreads();
  int cur = 0;
  while (true) {
    for (int s = 0; s < n_slices; s++, cur ^= 1) {
      const bool last_slice = s + 1 == n_slices;
      const int next_tile = tile + wgs_per_xcd;
      const bool more = !last_slice || next_tile < t_end;
      if (more) {
        if (last_slice) tile_ptrs(next_tile);
        g_load(last_slice ? 0 : s + 1);
      }
      multiply(cur);
      if (last_slice) store_tile(tile);
      if (more) s_store(cur ^ 1);
      __syncthreads();
    }
    tile += wgs_per_xcd;
    if (tile >= t_end) break;
  }
}

int w4_geom(const pcp_conv3x3_t *d, W4Geom *g) {
  if (!d) return PCP_ERR_ARG;
  if (d->stride != 1) return PCP_ERR_UNSUPPORTED;
  if (d->batch <= 0 || d->in_h <= 0 || d->in_w <= 0) return PCP_ERR_ARG;
  if (d->cin <= 0 || d->cin % G_KS != 0 || d->cout <= 0 || d->cout % 4 != 0) return PCP_ERR_ARG;
  if (d->cout_pad < d->cout || d->cout_pad % G_BN != 0) return PCP_ERR_ARG;
  if (d->ld_in % 4 != 0 || d->ld_out % 4 != 0) return PCP_ERR_ARG;
  g->batch = d->batch; g->h = d->in_h; g->w = d->in_w;
  g->tiles_x = (d->in_w + 3) / 4;
  g->tiles_y = (d->in_h + 3) / 4;
  g->tiles = (long long)d->batch * g->tiles_x * g->tiles_y;
  g->m_pad = (g->tiles + G_BM - 1) / G_BM * G_BM;
  g->cin = d->cin; g->cout = d->cout; g->n_pad = d->cout_pad;
  g->ld_in = d->ld_in; g->ld_out = d->ld_out; g->relu = d->relu;
  return PCP_OK;
}

}  // namespace

extern "C" int pcp_conv3x3_winograd4_workspace_bytes(const pcp_conv3x3_t *d, size_t *bytes) {
  W4Geom g;
  if (!bytes) return PCP_ERR_ARG;
  int rc = w4_geom(d, &g);
  if (rc != PCP_OK) return rc;
  *bytes = pcp_align_up((size_t)36 * g.m_pad * g.cin * 4, 256) + pcp_align_up((size_t)36 * g.m_pad * g.n_pad * 4, 256);
  return PCP_OK;
}

static int w4_run(const pcp_conv3x3_t *d, const float *in, const float *u_packed, const float *bias, float *out, void *workspace,
                  hipStream_t st, hipEvent_t *ev) {
  if (!in || !u_packed || !bias || !out || !workspace) return PCP_ERR_ARG;
  W4Geom g;
  int rc = w4_geom(d, &g);
  if (rc != PCP_OK) return rc;
  if ((((uintptr_t)in) & 15) || (((uintptr_t)u_packed) & 15) || (((uintptr_t)out) & 15) || (((uintptr_t)workspace) & 15) ||
      (((uintptr_t)bias) & 15))
    return PCP_ERR_ARG;
  float *v = reinterpret_cast<float *>(workspace);
  float *m = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + pcp_align_up((size_t)36 * g.m_pad * g.cin * 4, 256));

  const long long in_items = g.tiles * (g.cin / 4), out_items = g.tiles * (g.cout / 4);
  const long long in_blocks = (in_items + 255) / 256, out_blocks = (out_items + 255) / 256;
  GemmParams p;
  p.a = v; p.b = u_packed; p.c = m;
  p.m_pad = g.m_pad; p.n_pad = g.n_pad; p.k = g.cin;
  p.m_tiles = (int)(g.m_pad / G_BM);
  p.n_tiles = g.n_pad / G_BN;
  const long long gemm_tiles = 36LL * p.m_tiles * p.n_tiles;
  if (in_blocks > 0x7fffffffLL || out_blocks > 0x7fffffffLL || gemm_tiles > 0x7fffffffLL) return PCP_ERR_ARG;
  // persistent: two workgroups per CU (the kernel's LDS footprint allows no more), fewer when there are fewer tiles
  const long long gemm_blocks = gemm_tiles < G_PERSISTENT_WGS ? gemm_tiles : G_PERSISTENT_WGS;
  if ((long long)G_BM * g.cin * 4 > 0x7fffffffLL) return PCP_ERR_ARG;

  if (ev && hipEventRecord(ev[0], st) != hipSuccess) return PCP_ERR_LAUNCH;
  hipLaunchKernelGGL(k_w4_input, dim3((unsigned)in_blocks), dim3(256), 0, st, in, v, g);
  PCP_CHECK_LAUNCH();
  if (ev && hipEventRecord(ev[1], st) != hipSuccess) return PCP_ERR_LAUNCH;
  hipLaunchKernelGGL(k_w4_gemm, dim3((unsigned)gemm_blocks), dim3(G_THREADS), 0, st, p);
  PCP_CHECK_LAUNCH();
  if (ev && hipEventRecord(ev[2], st) != hipSuccess) return PCP_ERR_LAUNCH;
  hipLaunchKernelGGL(k_w4_output, dim3((unsigned)out_blocks), dim3(256), 0, st, m, bias, out, g);
  PCP_CHECK_LAUNCH();
  if (ev && hipEventRecord(ev[3], st) != hipSuccess) return PCP_ERR_LAUNCH;
  return PCP_OK;
}

extern "C" int pcp_conv3x3_winograd4(const pcp_conv3x3_t *d, const float *in, const float *u_packed, const float *bias, float *out,
                                     void *workspace, void *stream_) {
  return w4_run(d, in, u_packed, bias, out, workspace, (hipStream_t)stream_, nullptr);
}

extern "C" int pcp_conv3x3_winograd4_timed(const pcp_conv3x3_t *d, const float *in, const float *u_packed, const float *bias, float *out,
                                           void *workspace, void *stream_, float *stage_ms_host, double *gemm_flops_host) {
  if (!stage_ms_host) return PCP_ERR_ARG;
  hipStream_t st = (hipStream_t)stream_;
  hipEvent_t ev[4];
  int made = 0;
  for (; made < 4; made++)
    if (hipEventCreate(&ev[made]) != hipSuccess) break;
  int rc = made == 4 ? w4_run(d, in, u_packed, bias, out, workspace, st, ev) : PCP_ERR_LAUNCH;
  if (rc == PCP_OK && hipEventSynchronize(ev[3]) != hipSuccess) rc = PCP_ERR_LAUNCH;
  if (rc == PCP_OK)
    for (int i = 0; i < 3; i++)
      if (hipEventElapsedTime(&stage_ms_host[i], ev[i], ev[i + 1]) != hipSuccess) rc = PCP_ERR_LAUNCH;
  if (rc == PCP_OK && gemm_flops_host) {
    W4Geom g;
    w4_geom(d, &g);
    *gemm_flops_host = 2.0 * 36.0 * (double)g.m_pad * g.cin * g.n_pad;
  }
  for (int i = 0; i < made; i++) (void)hipEventDestroy(ev[i]);
  return rc;
}
