// a6 / a7 / a12 / a14 -- 3x3 stride-1 convolution as fused Winograd F(2x2, 3x3) on the fp32 MFMA pipe (gfx950).
//
// Same contract as pcp_conv3x3 (NHWC fp32 in/out, folded BN, bias + optional ReLU) but 2.25x fewer multiplies: the 3x3
// filter over a 2x2 output tile costs 16 instead of 36 multiply-adds per (cin, cout) pair.  Everything is fused in one launch:
//
//   workgroup (8 waves) = 16x16 output pixels (8x8 Winograd tiles) x 64 output channels, all 16 Winograd positions
//   per 8-channel slice:   raw 18x18x8 input patch (global -> registers -> LDS, two slices ahead)
//                          input transform  V = B^T d B   (VALU, LDS -> LDS, one slice ahead, double buffered)
//                          16 batched GEMMs [64 tiles x 8] x [8 x 64]  on v_mfma_f32_32x32x2_f32:
//                              A fragments  = V rows from LDS (one ds_read_b128 feeds 4 MFMAs, k permuted as in conv.hip)
//                              B fragments  = transformed weights U = G g G^T straight from global/L2 into registers:
//                                             wave w owns positions {2w, 2w+1}, so no two waves share a B fragment and the
//                                             weights never touch LDS; the packing [slice][pos][cout][8] makes each
//                                             (position, 32-channel) fragment one coalesced 1-KiB load
//   epilogue:              accumulators -> LDS in four (32 tiles x 32 channels x 16 positions) chunks -> output transform
//                          Y = A^T M A, bias, ReLU, 128-byte row stores
//   one barrier per slice; SIMD partners are de-phased (waves 0-3 transform first, waves 4-7 multiply first) so the matrix
//   pipe of every SIMD stays busy while its other wave does the VALU/LDS work.
//
// Accumulation is exact fp32 (MFMA = fmaf chain); the transforms add a few fp32 roundings (|error| ~1e-6 relative), well
// inside the 1e-3 parity bar.  Algorithmic FLOPs are those of the direct convolution (2*B*H*W*Cout*9*Cin): the achieved
// algorithmic rate can therefore exceed the MFMA peak (the kernel executes 4/9 of them).
#include "pcp_common.h"

namespace {

constexpr int WCK = 8;                    // input channels per slice
constexpr int WLD = 12;                   // padded V row (floats): 16 lanes of a ds_read_b128 group hit 16 distinct slots
constexpr int WTILES = 64;                // Winograd tiles per workgroup (8 x 8)
constexpr int WBN = 64;                   // output channels per workgroup
constexpr int WTHREADS = 512;
constexpr int RAW_W = 18;
constexpr int RAW_PIX = RAW_W * RAW_W;    // 324 pixels incl. halo
constexpr int RAW_FLOATS = RAW_PIX * WCK; // 2592
constexpr int V_FLOATS = 16 * WTILES * WLD;  // 12288
constexpr int MS_LD = 33;
constexpr int MS_FLOATS = 16 * 32 * MS_LD;   // 16896 (epilogue chunk), aliases the V buffers

struct WinoParams {
  const float *in;
  const float *u;       // [cin/8][16][cout_pad][8]
  const float *bias;
  float *out;
  int batch, h, w;
  int cin, cout, cout_pad;
  int ld_in, ld_out;
  int relu;
  int tiles_x, tiles_y, n_spatial;
};

__device__ __forceinline__ int xcd_remap_w(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

__device__ __forceinline__ f32x16 mfma32w(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__global__ __launch_bounds__(WTHREADS, 2) void k_conv3x3_wino(WinoParams p) {
  __shared__ __attribute__((aligned(16))) float lds[2 * RAW_FLOATS + 2 * V_FLOATS];
  float *rawb = lds;                       // [2][RAW_FLOATS]
  float *vb = lds + 2 * RAW_FLOATS;        // [2][V_FLOATS]   (epilogue: Ms chunk)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;

  const int lid = xcd_remap_w(blockIdx.x, gridDim.x);
  const int nt = lid / p.n_spatial;                    // N-tile is the slow index: one XCD works on few N-tiles at a time,
  int sp = lid % p.n_spatial;                          // so their transformed weights stay in that XCD's L2
  const int tile_x = sp % p.tiles_x;
  sp /= p.tiles_x;
  const int tile_y = sp % p.tiles_y;
  const int b = sp / p.tiles_y;
  const int oy0 = tile_y * 16, ox0 = tile_x * 16;
  const int n0 = nt * WBN;

  // ---- raw patch staging: 324 pixels x 2 float4; items tid and tid + 512 -----------------------------------------------
  const float *rsrc[2];
  int rdst[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    int idx = tid + i * WTHREADS;
    rsrc[i] = nullptr;
    rdst[i] = -1;
    if (idx < RAW_PIX * 2) {
      int pix = idx >> 1, q = idx & 1;
      int py = pix / RAW_W, px = pix % RAW_W;
      int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
      rdst[i] = pix * WCK + q * 4;
      if (iy >= 0 && iy < p.h && ix >= 0 && ix < p.w)
        rsrc[i] = p.in + ((long long)(b * p.h + iy) * p.w + ix) * p.ld_in + q * 4;
    }
  }
  f32x4 rreg[2];
  auto raw_load = [&](int slice) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      rreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (rsrc[i]) rreg[i] = *reinterpret_cast<const f32x4 *>(rsrc[i] + slice * WCK);
    }
  };
  auto raw_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; i++)
      if (rdst[i] >= 0) *reinterpret_cast<f32x4 *>(rawb + buf * RAW_FLOATS + rdst[i]) = rreg[i];
  };

  // ---- input transform item: (tile, channel) ----------------------------------------------------------------------------
  const int t_c = tid & 7, t_t = tid >> 3;                       // 64 tiles x 8 channels = 512 items
  const int t_src = ((t_t >> 3) * 2 * RAW_W + (t_t & 7) * 2) * WCK + t_c;
  const int t_dst = t_t * WLD + t_c;
  auto transform = [&](int rbuf, int vbuf) {
    const float *src = rawb + rbuf * RAW_FLOATS + t_src;
    float *dst = vb + vbuf * V_FLOATS + t_dst;
    // w = d B (row transform, one input row at a time), then V = B^T w
    float w[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++) {
      const float d0 = src[(a * RAW_W + 0) * WCK], d1 = src[(a * RAW_W + 1) * WCK];
      const float d2 = src[(a * RAW_W + 2) * WCK], d3 = src[(a * RAW_W + 3) * WCK];
      w[a][0] = d0 - d2;
      w[a][1] = d1 + d2;
      w[a][2] = d2 - d1;
      w[a][3] = d1 - d3;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      dst[(0 * 4 + j) * WTILES * WLD] = w[0][j] - w[2][j];
      dst[(1 * 4 + j) * WTILES * WLD] = w[1][j] + w[2][j];
      dst[(2 * 4 + j) * WTILES * WLD] = w[2][j] - w[1][j];
      dst[(3 * 4 + j) * WTILES * WLD] = w[1][j] - w[3][j];
    }
  };

  // ---- B fragments (transformed weights) from global: positions 2w, 2w+1; two 32-channel column tiles each ----------------
  const float *ubase = p.u + ((long long)(2 * wave) * p.cout_pad + n0 + r) * WCK + 4 * h;
  const long long u_pos = (long long)p.cout_pad * WCK;            // floats between positions
  const long long u_slice = 16 * u_pos;
  f32x4 bcur[2][2];
  auto b_load_pos = [&](int slice, int pi) {
    const float *s = ubase + slice * u_slice + pi * u_pos;
#pragma unroll
    for (int ct = 0; ct < 2; ct++) bcur[pi][ct] = *reinterpret_cast<const f32x4 *>(s + ct * 32 * WCK);
  };

  f32x16 acc[2][2][2];
#pragma unroll
  for (int pi = 0; pi < 2; pi++)
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
#pragma unroll
      for (int ct = 0; ct < 2; ct++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[pi][rt][ct][e] = 0.f;

  const int a_off = ((2 * wave) * WTILES + r) * WLD + 4 * h;
  // multiply slice `vbuf`'s V by the resident B fragments; as soon as a position's MFMAs are issued its B registers are
  // refilled with the NEXT slice's weights (one register set, prefetch distance = one whole slice)
  auto multiply = [&](int vbuf, int next_slice) {
    const float *vsrc = vb + vbuf * V_FLOATS + a_off;
#pragma unroll
    for (int pi = 0; pi < 2; pi++) {
#pragma unroll
      for (int rt = 0; rt < 2; rt++) {
        f32x4 a = *reinterpret_cast<const f32x4 *>(vsrc + (pi * WTILES + rt * 32) * WLD);
#pragma unroll
        for (int ct = 0; ct < 2; ct++) {
          acc[pi][rt][ct] = mfma32w(a.x, bcur[pi][ct].x, acc[pi][rt][ct]);
          acc[pi][rt][ct] = mfma32w(a.y, bcur[pi][ct].y, acc[pi][rt][ct]);
          acc[pi][rt][ct] = mfma32w(a.z, bcur[pi][ct].z, acc[pi][rt][ct]);
          acc[pi][rt][ct] = mfma32w(a.w, bcur[pi][ct].w, acc[pi][rt][ct]);
        }
      }
      if (next_slice >= 0) b_load_pos(next_slice, pi);
    }
  };

  const int n_slices = p.cin / WCK;
  // ---- prologue: raw(0) -> LDS, V(0); raw(1) -> LDS; B(0) ------------------------------------------------------------------
  raw_load(0);
  b_load_pos(0, 0);
  b_load_pos(0, 1);
  raw_store(0);
  if (n_slices > 1) raw_load(1);
  __syncthreads();
  transform(0, 0);
  if (n_slices > 1) raw_store(1);
  __syncthreads();

  for (int s = 0; s < n_slices; s++) {
    const int cur = s & 1, nxt = cur ^ 1;
    const bool has1 = s + 1 < n_slices, has2 = s + 2 < n_slices;
    if (has2) raw_load(s + 2);
    const int nxs = has1 ? s + 1 : -1;
    // SIMD partners run opposite orders so one multiplies while the other transforms
    // (one copy of the MFMA code: duplicating it in two branches makes the register allocator copy the accumulators)
    if (has1 && wave < 4) transform(nxt, nxt);
    multiply(cur, nxs);
    if (has1 && wave >= 4) transform(nxt, nxt);
    if (has2) raw_store(cur);           // raw[cur] was consumed by transform(s) one iteration ago
    __syncthreads();
  }

  // ---- epilogue: four chunks (row tile rt, column tile ct) through LDS ------------------------------------------------------
  float *ms = vb;
#pragma unroll
  for (int rt = 0; rt < 2; rt++) {
#pragma unroll
    for (int ct = 0; ct < 2; ct++) {
      if (rt + ct > 0) __syncthreads();
#pragma unroll
      for (int pi = 0; pi < 2; pi++) {
        const int pos = 2 * wave + pi;
#pragma unroll
        for (int e = 0; e < 16; e++) {
          int row = (e & 3) + 8 * (e >> 2) + 4 * h;
          ms[(pos * 32 + row) * MS_LD + r] = acc[pi][rt][ct][e];
        }
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int item = tid + k * WTHREADS;         // 32 tiles x 32 channels
        const int cc = item & 31, tt = item >> 5;
        float m[16];
#pragma unroll
        for (int q = 0; q < 16; q++) m[q] = ms[(q * 32 + tt) * MS_LD + cc];
        // Y = A^T M A,  A^T = [[1,1,1,0],[0,1,-1,-1]]
        float u0[4], u1[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          u0[j] = m[0 * 4 + j] + m[1 * 4 + j] + m[2 * 4 + j];
          u1[j] = m[1 * 4 + j] - m[2 * 4 + j] - m[3 * 4 + j];
        }
        float y[2][2];
        y[0][0] = u0[0] + u0[1] + u0[2];
        y[0][1] = u0[1] - u0[2] - u0[3];
        y[1][0] = u1[0] + u1[1] + u1[2];
        y[1][1] = u1[1] - u1[2] - u1[3];
        const int n = n0 + ct * 32 + cc;
        if (n < p.cout) {
          const float bias = p.bias[n];
          const int t = rt * 32 + tt;
          const int py = oy0 + (t >> 3) * 2, px = ox0 + (t & 7) * 2;
#pragma unroll
          for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
              int oy = py + i, ox = px + j;
              if (oy < p.h && ox < p.w) {
                float v = y[i][j] + bias;
                if (p.relu) v = fmaxf(v, 0.f);
                p.out[((long long)(b * p.h + oy) * p.w + ox) * p.ld_out + n] = v;
              }
            }
        }
      }
    }
  }
}

}  // namespace

extern "C" int pcp_conv3x3_winograd(const pcp_conv3x3_t *d, const float *in, const float *u_packed, const float *bias, float *out,
                                    void *stream_) {
  if (!d || !in || !u_packed || !bias || !out) return PCP_ERR_ARG;
  if (d->stride != 1) return PCP_ERR_UNSUPPORTED;
  if (d->cin <= 0 || d->cin % WCK != 0 || d->cout <= 0 || d->cout_pad < d->cout || d->cout_pad % WBN != 0) return PCP_ERR_ARG;
  if (d->ld_in % 4 != 0 || (((uintptr_t)in) & 15) || (((uintptr_t)u_packed) & 15)) return PCP_ERR_ARG;
  if (d->batch <= 0 || d->in_h <= 0 || d->in_w <= 0) return PCP_ERR_ARG;
  WinoParams p;
  p.in = in; p.u = u_packed; p.bias = bias; p.out = out;
  p.batch = d->batch; p.h = d->in_h; p.w = d->in_w;
  p.cin = d->cin; p.cout = d->cout; p.cout_pad = d->cout_pad;
  p.ld_in = d->ld_in; p.ld_out = d->ld_out; p.relu = d->relu;
  p.tiles_x = (d->in_w + 15) / 16;
  p.tiles_y = (d->in_h + 15) / 16;
  p.n_spatial = d->batch * p.tiles_x * p.tiles_y;
  long long blocks = (long long)p.n_spatial * (d->cout_pad / WBN);
  if (blocks <= 0 || blocks > 0x7fffffffLL) return PCP_ERR_ARG;
  hipLaunchKernelGGL(k_conv3x3_wino, dim3((unsigned)blocks), dim3(WTHREADS), 0, (hipStream_t)stream_, p);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}
