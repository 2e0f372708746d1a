// Weight gradients of the dense convolutions as fp32 MFMA GEMMs whose contraction index is the PIXEL:
//   3x3 (stride 1 / 2, padding 1):  dW[co][ci][ky][kx] = sum_{b,oy,ox} dy[b,oy,ox,co] * x[b, oy*s+ky-1, ox*s+kx-1, ci]
//   pointwise family:               dW[n][k]           = sum_r A[map_a(r)][n] * B[map_b(r)][k]
// Replaces the cuDNN backward-filter kernels autograd launches for the trainable branch of config 5
// (nn.Conv2d / nn.ConvTranspose2d of base_bev_backbone.py:30-69, center_head.py:24-29,75-82, v2x_fusion_disco.py:11-17,51-63).
//
// NHWC makes both operands K-major rows: for one pixel, the 32 channels a v_mfma_f32_32x32x2_f32 lane group needs are 128
// contiguous bytes.  A workgroup owns a 64(co) x 64(ci) tile of ALL nine taps (wave = one 32x32 quadrant, 9 x 16 accumulator
// VGPRs), stages one dy tile and ONE input patch with halo per pixel tile in LDS and re-reads the patch at the nine shifts
// (9 MFMAs per 10 ds_read_b32).  The pixel range is split over blockIdx.y; partial sums go to a workspace in a fixed order and
// are reduced by a second kernel (deterministic, no float atomics), which also emits PyTorch's (cout, cin, 3, 3) layout.
#include "pcp_common.h"

namespace {

constexpr int WG_THREADS = 256;
constexpr int WG8_THREADS = 512;        // the 3x3 kernel: eight waves, one workgroup per CU
constexpr int TILE_C = 64;

struct Wg3Params {
  const float *x, *dy;
  float *part;                  // [nsplit][9][cout_r][cin_r]
  int batch, in_h, in_w, out_h, out_w, cin, cout, ld_x, ld_dy;
  int tiles_y, tiles_x, n_tiles, nsplit, ci_tiles, cout_r, cin_r;
};

// Round 3 form: ONE eight-wave workgroup per CU.  Wave = (quadrant of the 64 x 64 tile, row parity of the pixel tile): two waves per SIMD
// accumulate the same quadrant over alternate tile rows and are summed through LDS once, at the end of the workgroup.  The pixel tiles are
// double buffered in LDS: the global loads of tile t+1 are requested (into registers) before the MFMAs of tile t and stored after them, one
// barrier per tile (the round-2 form staged a tile, waited, multiplied, and relied on the second workgroup of the CU to cover the wait; it
// also wrote twice as many partial tiles: 2 workgroups x 147 KB per CU).
template <int S, int TH, int TW>
__global__ __launch_bounds__(WG8_THREADS, 2) void k_wgrad3x3(Wg3Params p) {
  constexpr int PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3;
  constexpr int NPIX = TH * TW, NPATCH = PH * PW;
  constexpr int STAGE = (NPIX + NPATCH) * TILE_C;                       // floats per pipeline stage
  constexpr int RED = 4 * 9 * 16 * 64;                                  // the end-of-workgroup exchange: [quadrant][tap][register][lane]
  constexpr int LDS_FLOATS = 2 * STAGE > RED ? 2 * STAGE : RED;
  constexpr int DY_PER = (NPIX * 16 + WG8_THREADS - 1) / WG8_THREADS, X_PER = (NPATCH * 16 + WG8_THREADS - 1) / WG8_THREADS;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int hl = lane >> 5, l32 = lane & 31;
  const int quad = wave & 3, par = wave >> 2;
  const int co_tile = blockIdx.x / p.ci_tiles, ci_tile = blockIdx.x % p.ci_tiles;
  const int co0 = co_tile * TILE_C, ci0 = ci_tile * TILE_C;
  const int wco = (quad >> 1) * 32, wci = (quad & 1) * 32;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  float4 rdy[DY_PER], rx[X_PER];
  auto gload = [&](int tile) {
    const int tx = tile % p.tiles_x;
    const int ty = (tile / p.tiles_x) % p.tiles_y;
    const int b = tile / (p.tiles_x * p.tiles_y);
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
#pragma unroll
    for (int k = 0; k < DY_PER; ++k) {
      const int i = tid + k * WG8_THREADS;
      const int pix = i >> 4, q = i & 15;
      const int oy = oy0 + pix / TW, ox = ox0 + pix % TW;
      const int co = co0 + q * 4;
      rdy[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < NPIX * 16 && oy < p.out_h && ox < p.out_w && co < p.cout)
        rdy[k] = *reinterpret_cast<const float4 *>(p.dy + (((long long)b * p.out_h + oy) * p.out_w + ox) * p.ld_dy + co);
    }
#pragma unroll
    for (int k = 0; k < X_PER; ++k) {
      const int i = tid + k * WG8_THREADS;
      const int pp = i >> 4, q = i & 15;
      const int iy = iy0 + pp / PW, ix = ix0 + pp % PW;
      const int ci = ci0 + q * 4;
      rx[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < NPATCH * 16 && iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w && ci < p.cin)
        rx[k] = *reinterpret_cast<const float4 *>(p.x + (((long long)b * p.in_h + iy) * p.in_w + ix) * p.ld_x + ci);
    }
  };
  auto lstore = [&](int buf) {
    float *dyT = lds + buf * STAGE;                 // [NPIX][64]
    float *xP = dyT + NPIX * TILE_C;                // [PH * PW][64]
#pragma unroll
    for (int k = 0; k < DY_PER; ++k) {
      const int i = tid + k * WG8_THREADS;
      if (i < NPIX * 16) *reinterpret_cast<float4 *>(dyT + i * 4) = rdy[k];
    }
#pragma unroll
    for (int k = 0; k < X_PER; ++k) {
      const int i = tid + k * WG8_THREADS;
      if (i < NPATCH * 16) *reinterpret_cast<float4 *>(xP + i * 4) = rx[k];
    }
  };

  int tile = blockIdx.y;
  if (tile < p.n_tiles) {
    gload(tile);
    lstore(0);
  }
  __syncthreads();
  int buf = 0;
  for (; tile < p.n_tiles; tile += p.nsplit) {
    const int next = tile + p.nsplit;
    const bool has_next = next < p.n_tiles;
    if (has_next) gload(next);                       // in flight under this tile's MFMAs
    const float *dyT = lds + buf * STAGE;
    const float *xP = dyT + NPIX * TILE_C;
    // tile rows of this wave's parity, TW/2 pixel pairs per row fully unrolled: every LDS address is a loop-invariant base + an immediate
#pragma unroll 1
    for (int py = par; py < TH; py += 2) {
      const float *ab = dyT + (py * TW + hl) * TILE_C + wco + l32;
      const float *xb = xP + ((py * S) * PW + hl * S) * TILE_C + wci + l32;
#pragma unroll
      for (int pp = 0; pp < TW / 2; ++pp) {
        const float a = ab[2 * pp * TILE_C];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const float bv = xb[(ky * PW + kx + 2 * pp * S) * TILE_C];
            acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[ky * 3 + kx], 0, 0, 0);
          }
      }
    }
    if (has_next) lstore(buf ^ 1);                   // the other stage was last read before the previous barrier
    __syncthreads();
    buf ^= 1;
  }
  // the two row-parity waves of a quadrant exchange the taps they do not write: parity 0 keeps taps 0..4, parity 1 taps 5..8
  {
    float *red = lds + quad * (9 * 16 * 64) + lane;
#pragma unroll
    for (int t = 0; t < 9; ++t)
      if ((t < 5) == (par == 1))
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(t * 16 + r) * 64] = acc[t][r];
    __syncthreads();
    float *out = p.part + (long long)blockIdx.y * 9 * p.cout_r * p.cin_r;
#pragma unroll
    for (int t = 0; t < 9; ++t)
      if ((t < 5) == (par == 0))
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + wco + (r >> 2) * 8 + hl * 4 + (r & 3);
          out[((long long)t * p.cout_r + co) * p.cin_r + ci0 + wci + l32] = acc[t][r] + red[(t * 16 + r) * 64];
        }
  }
}

// dw[co][ci][tap] (+)= sum_s part[s][tap][co][ci].  Block = 64 consecutive outputs x 4 split lanes (coalesced 256-B reads per
// split, four independent load streams), combined in a fixed order.
__global__ __launch_bounds__(256) void k_wgrad3x3_reduce(const float *__restrict__ part, int nsplit, int cout, int cin, int cout_r, int cin_r,
                                                        float *__restrict__ dw, int accumulate) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const long long t = (long long)blockIdx.x * 64 + lane;
  const long long total = (long long)9 * cout * cin;
  float s = 0.f;
  int ci = 0, co = 0, tap = 0;
  if (t < total) {
    ci = (int)(t % cin);
    co = (int)((t / cin) % cout);
    tap = (int)(t / ((long long)cin * cout));
    const long long stride = (long long)9 * cout_r * cin_r;
    const float *src = part + ((long long)tap * cout_r + co) * cin_r + ci;
    for (int k = sl; k < nsplit; k += 4) s += src[k * stride];
  }
  red[sl][lane] = s;
  __syncthreads();
  if (sl == 0 && t < total) {
    const float v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    float *d = dw + ((long long)co * cin + ci) * 9 + tap;
    *d = accumulate ? *d + v : v;
  }
}

// ---- pointwise family -------------------------------------------------------------------------------------------------------
struct RowMap {
  const float *ptr;
  int ld, ch;                  // pixel stride, valid channels
  int mode;                    // 0: pixel = r;  1: r = (b, y, x) on an (h, w) grid -> pixel (b, 2y + ky, 2x + kx) of a (2h, 2w) grid
  int h, w, ky, kx;
};

__device__ __forceinline__ long long map_row(const RowMap &m, long long r) {
  if (m.mode == 0) return r;
  const int x = (int)(r % m.w);
  const long long t = r / m.w;
  const int y = (int)(t % m.h);
  const long long b = t / m.h;
  return (b * 2 * m.h + 2 * y + m.ky) * 2 * m.w + 2 * x + m.kx;
}

struct WgPwParams {
  RowMap a, b;                 // out[n][k] = sum_r a[map(r)][n] * b[map(r)][k]
  long long rows;
  float *part;                 // [nsplit][n_r][k_r]
  int k_tiles, n_r, k_r, nsplit, chunks;
};

constexpr int PW_ROWS = 128;

__global__ __launch_bounds__(WG_THREADS, 2) void k_wgrad_pw(WgPwParams p) {
  __shared__ float aT[PW_ROWS * TILE_C];
  __shared__ float bT[PW_ROWS * TILE_C];
  constexpr int PER = PW_ROWS * 16 / WG_THREADS;            // float4 items per thread, chunk and operand (8)
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int hl = lane >> 5, l32 = lane & 31;
  const int n0 = (blockIdx.x / p.k_tiles) * TILE_C, k0 = (blockIdx.x % p.k_tiles) * TILE_C;
  // quadrants of the 64 x 64 tile that hold channels: 1, 2 or 4 (a 32 x 16 layer has one).  With fewer than four, the idle waves take a
  // share of the chunk's ROWS for the same quadrant (the MFMA count per chunk is what bounds narrow layers: 64 per wave whatever the
  // channels) and the partial accumulators are summed through LDS in wave order at the end.
  const int qn = (p.a.ch - n0 > 32) ? 2 : 1, qk = (p.b.ch - k0 > 32) ? 2 : 1;
  const int nq = qn * qk, parts = 4 / nq;
  const int quad = wave % nq, part = wave / nq;
  const int wn = (qk == 2 ? (quad >> 1) : quad) * 32, wk = (qk == 2 ? (quad & 1) : 0) * 32;
  const int kk0 = part * (PW_ROWS / 2 / parts), kk1 = kk0 + PW_ROWS / 2 / parts;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // the rows of chunk c + 1 are requested before the 64 MFMAs of chunk c and stored to LDS after them (one set of registers in flight):
  // the synchronous form exposed a full memory latency per 128 rows (rows = 1.4 M in the PFN: 346 -> see profiles/r04_wgrad_pw_prefetch.txt)
  float4 ra[PER], rb[PER];
  auto fetch = [&](int chunk) {
    const long long r0 = (long long)chunk * PW_ROWS;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int i = tid + u * WG_THREADS;
      const int rr = i >> 4, q = i & 15;
      const long long r = r0 + rr;
      ra[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      rb[u] = ra[u];
      if (r < p.rows) {
        if (n0 + q * 4 < p.a.ch) ra[u] = *reinterpret_cast<const float4 *>(p.a.ptr + map_row(p.a, r) * p.a.ld + n0 + q * 4);
        if (k0 + q * 4 < p.b.ch) rb[u] = *reinterpret_cast<const float4 *>(p.b.ptr + map_row(p.b, r) * p.b.ld + k0 + q * 4);
      }
    }
  };
  int chunk = blockIdx.y;
  if (chunk < p.chunks) fetch(chunk);
  for (; chunk < p.chunks; chunk += p.nsplit) {
    __syncthreads();                              // every wave is done with the previous chunk's tiles
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int i = tid + u * WG_THREADS;
      const int rr = i >> 4, q = i & 15;
      *reinterpret_cast<float4 *>(aT + rr * TILE_C + q * 4) = ra[u];
      *reinterpret_cast<float4 *>(bT + rr * TILE_C + q * 4) = rb[u];
    }
    __syncthreads();
    if (chunk + p.nsplit < p.chunks) fetch(chunk + p.nsplit);
#pragma unroll 8
    for (int kk = kk0; kk < kk1; ++kk) {
      const int rr = 2 * kk + hl;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aT[rr * TILE_C + wn + l32], bT[rr * TILE_C + wk + l32], acc, 0, 0, 0);
    }
  }
  float *out = p.part + (long long)blockIdx.y * p.n_r * p.k_r;
  if (parts > 1) {
    __syncthreads();                              // the tiles are dead: aT becomes [wave][16][64 lanes]
#pragma unroll
    for (int r = 0; r < 16; ++r) aT[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    if (part != 0) return;
    for (int o = 1; o < parts; ++o)               // fixed order: parts 1, 2, 3 of this quadrant
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += aT[((quad + o * nq) * 16 + r) * 64 + lane];
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = n0 + wn + (r >> 2) * 8 + hl * 4 + (r & 3);
    out[(long long)n * p.k_r + k0 + wk + l32] = acc[r];
  }
  // (quadrants without channels are not written: k_wgrad_pw_reduce reads n < channels, k < channels only)
}

__global__ __launch_bounds__(256) void k_wgrad_pw_reduce(const float *__restrict__ part, int nsplit, int n, int k, int n_r, int k_r,
                                                        float *__restrict__ out, int ld_out, int accumulate) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const long long t = (long long)blockIdx.x * 64 + lane;
  float s = 0.f;
  int kk = 0, nn = 0;
  if (t < (long long)n * k) {
    kk = (int)(t % k);
    nn = (int)(t / k);
    const float *src = part + (long long)nn * k_r + kk;
    const long long stride = (long long)n_r * k_r;
    for (int i = sl; i < nsplit; i += 4) s += src[i * stride];
  }
  red[sl][lane] = s;
  __syncthreads();
  if (sl == 0 && t < (long long)n * k) {
    const float v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    float *d = out + (long long)nn * ld_out + kk;
    *d = accumulate ? *d + v : v;
  }
}

inline int round64(int v) { return (v + 63) / 64 * 64; }

struct Wg3Plan { int th, tw, tiles_y, tiles_x, n_tiles, nsplit, cout_r, cin_r; };

inline Wg3Plan plan3(const pcp_conv3x3_t *d) {
  Wg3Plan pl;
  const int oh = d->stride == 2 ? d->in_h / 2 : d->in_h, ow = d->stride == 2 ? d->in_w / 2 : d->in_w;
  pl.th = d->stride == 2 ? 4 : 8;
  pl.tw = d->stride == 2 ? 8 : 16;
  pl.tiles_y = (oh + pl.th - 1) / pl.th;
  pl.tiles_x = (ow + pl.tw - 1) / pl.tw;
  pl.n_tiles = d->batch * pl.tiles_y * pl.tiles_x;
  pl.cout_r = round64(d->cout);
  pl.cin_r = round64(d->cin);
  const int pairs = (pl.cout_r / 64) * (pl.cin_r / 64);
  int ns = 256 / pairs;          // <= 256 blocks: one eight-wave workgroup per CU, no second partial wave
  if (ns > 256) ns = 256;
  if (ns > pl.n_tiles) ns = pl.n_tiles;
  if (ns < 1) ns = 1;
  pl.nsplit = ns;
  return pl;
}

inline int pw_split(long long rows, int n_r, int k_r, int *chunks) {
  const int ch = (int)((rows + PW_ROWS - 1) / PW_ROWS);
  const int pairs = (n_r / 64) * (k_r / 64);
  int ns = 512 / pairs;
  if (ns > 256) ns = 256;
  if (ns > ch) ns = ch;
  if (ns < 1) ns = 1;
  *chunks = ch;
  return ns;
}

}  // namespace

extern "C" {

size_t pcp_conv3x3_wgrad_workspace_bytes(const pcp_conv3x3_t *d) {
  if (!d) return 0;
  const Wg3Plan pl = plan3(d);
  return (size_t)pl.nsplit * 9 * pl.cout_r * pl.cin_r * sizeof(float);
}

int pcp_conv3x3_wgrad(const pcp_conv3x3_t *d, const float *x, const float *dy, void *workspace, size_t workspace_bytes, float *dw,
                      int32_t accumulate, void *stream) {
  if (!d || !x || !dy || !workspace || !dw) return PCP_ERR_ARG;
  if (d->batch <= 0 || d->in_h <= 0 || d->in_w <= 0 || (d->cin & 3) || (d->cout & 3) || (d->ld_in & 3) || (d->ld_out & 3) ||
      (d->stride != 1 && d->stride != 2) || (d->stride == 2 && ((d->in_h | d->in_w) & 1)))
    return PCP_ERR_ARG;
  if ((((uintptr_t)x) | ((uintptr_t)dy)) & 15) return PCP_ERR_ARG;
  const Wg3Plan pl = plan3(d);
  if (workspace_bytes < pcp_conv3x3_wgrad_workspace_bytes(d)) return PCP_ERR_WORKSPACE;
  Wg3Params p{};
  p.x = x; p.dy = dy; p.part = (float *)workspace;
  p.batch = d->batch; p.in_h = d->in_h; p.in_w = d->in_w;
  p.out_h = d->stride == 2 ? d->in_h / 2 : d->in_h;
  p.out_w = d->stride == 2 ? d->in_w / 2 : d->in_w;
  p.cin = d->cin; p.cout = d->cout; p.ld_x = d->ld_in; p.ld_dy = d->ld_out;
  p.tiles_y = pl.tiles_y; p.tiles_x = pl.tiles_x; p.n_tiles = pl.n_tiles; p.nsplit = pl.nsplit;
  p.ci_tiles = pl.cin_r / 64; p.cout_r = pl.cout_r; p.cin_r = pl.cin_r;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((pl.cout_r / 64) * (pl.cin_r / 64), pl.nsplit);
  if (d->stride == 1) {
    hipLaunchKernelGGL((k_wgrad3x3<1, 8, 16>), grid, dim3(WG8_THREADS), 0, s, p);
  } else {
    hipLaunchKernelGGL((k_wgrad3x3<2, 4, 8>), grid, dim3(WG8_THREADS), 0, s, p);
  }
  const long long total = (long long)9 * d->cout * d->cin;
  hipLaunchKernelGGL(k_wgrad3x3_reduce, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, (const float *)workspace, pl.nsplit,
                     d->cout, d->cin, pl.cout_r, pl.cin_r, dw, accumulate);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

size_t pcp_pointwise_wgrad_workspace_bytes(int64_t rows, int32_t n, int32_t k) {
  int chunks;
  const int n_r = round64(n), k_r = round64(k);
  const int ns = pw_split(rows, n_r, k_r, &chunks);
  return (size_t)ns * n_r * k_r * sizeof(float);
}

int pcp_pointwise_wgrad(const pcp_rowmap_t *a, const pcp_rowmap_t *b, int64_t rows, void *workspace, size_t workspace_bytes,
                        float *out, int32_t ld_out, int32_t accumulate, void *stream) {
  if (!a || !b || !a->ptr || !b->ptr || !workspace || !out || rows <= 0) return PCP_ERR_ARG;
  if ((a->channels & 3) || (b->channels & 3) || (a->ld & 3) || (b->ld & 3) || a->channels <= 0 || b->channels <= 0) return PCP_ERR_ARG;
  if ((((uintptr_t)a->ptr) | ((uintptr_t)b->ptr)) & 15) return PCP_ERR_ARG;
  WgPwParams p{};
  auto cvt = [](const pcp_rowmap_t *m) {
    RowMap r;
    r.ptr = m->ptr; r.ld = m->ld; r.ch = m->channels; r.mode = m->lattice ? 1 : 0;
    r.h = m->grid_h; r.w = m->grid_w; r.ky = m->ky; r.kx = m->kx;
    return r;
  };
  p.a = cvt(a); p.b = cvt(b);
  if ((p.a.mode && (p.a.h <= 0 || p.a.w <= 0)) || (p.b.mode && (p.b.h <= 0 || p.b.w <= 0))) return PCP_ERR_ARG;
  p.rows = rows;
  p.n_r = round64(a->channels); p.k_r = round64(b->channels);
  p.k_tiles = p.k_r / 64;
  p.nsplit = pw_split(rows, p.n_r, p.k_r, &p.chunks);
  if (workspace_bytes < (size_t)p.nsplit * p.n_r * p.k_r * sizeof(float)) return PCP_ERR_WORKSPACE;
  p.part = (float *)workspace;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_wgrad_pw, dim3((p.n_r / 64) * p.k_tiles, p.nsplit), dim3(WG_THREADS), 0, s, p);
  const long long total = (long long)a->channels * b->channels;
  hipLaunchKernelGGL(k_wgrad_pw_reduce, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, (const float *)workspace, p.nsplit,
                     a->channels, b->channels, p.n_r, p.k_r, out, ld_out, accumulate);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

}  // extern "C"
