// a6 / a7 / a12 / a14 -- dense BEV convolutions as fp32 MFMA implicit GEMM on gfx950.
//
// Replaces the cuDNN Conv2d / ConvTranspose2d + BatchNorm2d + ReLU stacks of
//   pcdet/models/backbones_2d/base_bev_backbone.py:30-69, pcdet/models/dense_heads/center_head.py:24-29,75-82,
//   pcdet/models/bev_layers/v2x_fusion_disco.py:8-26,51-63, pcdet/models/bev_layers/hunter_jr.py:132,149-152.
//
// Design (MI355X first, not a cuDNN re-tiling):
//   * activations are NHWC fp32, so the contraction index (input channel) is contiguous for both operands;
//   * one workgroup = TH x TW output pixels x BN output channels.  Per 16-channel slice it stages ONE input patch with
//     halo ((TH-1)*S+3) x ((TW-1)*S+3) x 16 in LDS and re-reads it for all nine taps (nine-fold reuse of every staged
//     byte instead of an im2col gather), next to the slice's 9 x BN x 16 folded weights;
//   * the math is v_mfma_f32_32x32x2_f32: exact fp32 products and accumulation (bitwise an fmaf chain), 256 FLOP/clk/CU,
//     peak 157 TFLOP/s -- the parity mode of the 1e-3 bar;
//   * the MFMA k index is permuted so that each lane fetches FOUR k's with one ds_read_b128 and feeds four back-to-back
//     MFMAs: lane (r, h) reads channels 8g+4h .. 8g+4h+3 of row r; MFMA j of the group contracts the channel pair
//     {8g+j, 8g+4+j}.  A and B use the same permutation, so the sum is unchanged;
//   * rows are padded to 20 floats (80 B): the sixteen lanes of a ds_read_b128 group land on sixteen distinct 16-B slots;
//   * global -> register prefetch of slice c+1 is issued before the MFMAs of slice c (latency hidden without a second
//     LDS buffer); two workgroups per CU (LDS 61 KB each) overlap each other's barriers;
//   * blockIdx is remapped so the N-tiles of one spatial tile (which share the input patch) sit on one XCD's L2.
//
// Algorithmic FLOPs per launch: 2 * B*Ho*Wo * cout * 9*cin.
#include "pcp_common.h"

namespace {

constexpr int CK = 16;    // input channels per staged slice
constexpr int LDK = 20;   // padded LDS row length in floats
constexpr int CONV_THREADS = 256;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  // blocks b and b+8 share an XCD (round-robin dispatch); give each XCD a contiguous run of logical tiles. Bijective.
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

struct Conv3Params {
  const float *in;
  const float *w;       // [cin/16][9][cout_pad][16]
  const float *bias;    // [cout_pad]
  float *out;
  int batch, in_h, in_w, out_h, out_w;
  int cin, cout, cout_pad;
  int ld_in, ld_out;
  int relu;
  int tiles_x, tiles_y, n_tiles;
  int vec_out;          // output rows 16-byte aligned: 16-byte stores
};

template <int S, int TH, int TW, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(CONV_THREADS, 2) void k_conv3x3(Conv3Params p) {
  constexpr int BM = TH * TW;
  constexpr int PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MI = WTM / 32, NI = WTN / 32;
  static_assert(WAVES_M * WAVES_N * 64 == CONV_THREADS, "4 waves");
  static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile is a multiple of the 32x32 MFMA tile");
  constexpr int A_F4 = PH * PW * (CK / 4);                 // float4 items of the patch slice
  constexpr int B_F4 = 9 * BN * (CK / 4);
  constexpr int A_PER = (A_F4 + CONV_THREADS - 1) / CONV_THREADS;
  constexpr int B_PER = (B_F4 + CONV_THREADS - 1) / CONV_THREADS;

  __shared__ __attribute__((aligned(16))) float lds[(PH * PW + 9 * BN) * LDK];
  float *As = lds;
  float *Bs = lds + PH * PW * LDK;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  const int nwg = gridDim.x;
  int lid = xcd_remap(blockIdx.x, nwg);
  const int nt = lid % p.n_tiles;
  int sp = lid / p.n_tiles;
  const int tile_x = sp % p.tiles_x;
  sp /= p.tiles_x;
  const int tile_y = sp % p.tiles_y;
  const int b = sp / p.tiles_y;
  const int oy0 = tile_y * TH, ox0 = tile_x * TW;
  const int n0 = nt * BN;
  const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;

  // ---- per-thread staging coordinates (fixed across slices) -------------------------------------------------------
  const float *a_src[A_PER];
  int a_dst[A_PER];
#pragma unroll
  for (int i = 0; i < A_PER; i++) {
    int idx = tid + i * CONV_THREADS;
    a_src[i] = nullptr;
    a_dst[i] = -1;
    if (idx < A_F4) {
      int pix = idx >> 2, q = idx & 3;
      int py = pix / PW, px = pix % PW;
      int iy = iy0 + py, ix = ix0 + px;
      a_dst[i] = pix * LDK + q * 4;
      if (iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w)
        a_src[i] = p.in + ((long long)(b * p.in_h + iy) * p.in_w + ix) * p.ld_in + q * 4;
    }
  }
  const float *b_src0 = p.w + (long long)n0 * CK;           // + slice*9*cout_pad*16 + tap*cout_pad*16 + n*16 + q*4
  int b_goff[B_PER], b_dst[B_PER];
#pragma unroll
  for (int i = 0; i < B_PER; i++) {
    int idx = tid + i * CONV_THREADS;
    b_dst[i] = -1;
    b_goff[i] = 0;
    if (idx < B_F4) {
      int q = idx & 3, n = (idx >> 2) % BN, tap = idx / (4 * BN);
      b_goff[i] = (tap * p.cout_pad + n) * CK + q * 4;
      b_dst[i] = (tap * BN + n) * LDK + q * 4;
    }
  }

  f32x4 a_reg[A_PER], b_reg[B_PER];
  auto load_slice = [&](int slice) {
    const int c0 = slice * CK;
#pragma unroll
    for (int i = 0; i < A_PER; i++) {
      a_reg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (a_src[i]) a_reg[i] = *reinterpret_cast<const f32x4 *>(a_src[i] + c0);
    }
    const float *bs = b_src0 + (long long)slice * 9 * p.cout_pad * CK;
#pragma unroll
    for (int i = 0; i < B_PER; i++)
      if (b_dst[i] >= 0) b_reg[i] = *reinterpret_cast<const f32x4 *>(bs + b_goff[i]);
  };
  auto store_slice = [&]() {
#pragma unroll
    for (int i = 0; i < A_PER; i++)
      if (a_dst[i] >= 0) *reinterpret_cast<f32x4 *>(As + a_dst[i]) = a_reg[i];
#pragma unroll
    for (int i = 0; i < B_PER; i++)
      if (b_dst[i] >= 0) *reinterpret_cast<f32x4 *>(Bs + b_dst[i]) = b_reg[i];
  };

  // ---- fragment addresses ------------------------------------------------------------------------------------------
  int a_off[MI], b_off[NI];
#pragma unroll
  for (int i = 0; i < MI; i++) {
    int m = wm * WTM + i * 32 + r;
    int ty = m / TW, tx = m % TW;
    a_off[i] = ((ty * S) * PW + tx * S) * LDK + 4 * h;
  }
#pragma unroll
  for (int j = 0; j < NI; j++) b_off[j] = (wn * WTN + j * 32 + r) * LDK + 4 * h;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; i++)
#pragma unroll
    for (int j = 0; j < NI; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

  const int n_slices = p.cin / CK;
  load_slice(0);
  for (int slice = 0; slice < n_slices; slice++) {
    store_slice();
    __syncthreads();
    if (slice + 1 < n_slices) load_slice(slice + 1);
#pragma unroll
    for (int tap = 0; tap < 9; tap++) {
      const int ky = tap / 3, kx = tap % 3;
#pragma unroll
      for (int g = 0; g < CK / 8; g++) {
        f32x4 af[MI], bf[NI];
#pragma unroll
        for (int i = 0; i < MI; i++)
          af[i] = *reinterpret_cast<const f32x4 *>(As + a_off[i] + (ky * PW + kx) * LDK + g * 8);
#pragma unroll
        for (int j = 0; j < NI; j++)
          bf[j] = *reinterpret_cast<const f32x4 *>(Bs + b_off[j] + tap * BN * LDK + g * 8);
#pragma unroll
        for (int i = 0; i < MI; i++)
#pragma unroll
          for (int j = 0; j < NI; j++) {
            // weights = A operand (rows = output channels), pixels = B operand (columns): four consecutive channels of one pixel per
            // accumulator quad -> 16-byte stores
            acc[i][j] = mfma32(bf[j].x, af[i].x, acc[i][j]);
            acc[i][j] = mfma32(bf[j].y, af[i].y, acc[i][j]);
            acc[i][j] = mfma32(bf[j].z, af[i].z, acc[i][j]);
            acc[i][j] = mfma32(bf[j].w, af[i].w, acc[i][j]);
          }
      }
    }
    __syncthreads();
  }

  // ---- epilogue: bias + ReLU; lane (r, h) holds channels nb + 8 q + 4 h + (0..3) of pixel wm * WTM + i * 32 + r in accumulator quad q -----
#pragma unroll
  for (int j = 0; j < NI; j++) {
    const int nb = n0 + wn * WTN + j * 32;
#pragma unroll
    for (int i = 0; i < MI; i++) {
      const int m = wm * WTM + i * 32 + r;
      const int oy = oy0 + m / TW, ox = ox0 + m % TW;
      if (oy >= p.out_h || ox >= p.out_w) continue;
      float *orow = p.out + ((long long)(b * p.out_h + oy) * p.out_w + ox) * p.ld_out;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int co = nb + 8 * q + 4 * h;
        if (co >= p.cout) continue;
        const f32x4 bias = *reinterpret_cast<const f32x4 *>(p.bias + co);          // padded to cout_pad
        f32x4 v = f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]} + bias;
        if (p.relu) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        if (co + 3 < p.cout && p.vec_out) {
          *reinterpret_cast<f32x4 *>(orow + co) = v;
        } else {
          orow[co] = v.x;
          if (co + 1 < p.cout) orow[co + 1] = v.y;
          if (co + 2 < p.cout) orow[co + 2] = v.z;
          if (co + 3 < p.cout) orow[co + 3] = v.w;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// pointwise family: 1x1 conv / Linear (PLAIN), Conv2d k2 s2 (SPACE2DEPTH), ConvTranspose2d k2 s2 (DEPTH2SPACE)
// ---------------------------------------------------------------------------------------------------------------------
struct PwParams {
  const float *in;
  const float *in2;     // PLAIN only: second K source for k >= k_split (nullptr = single source)
  const float *residual;// PLAIN only: added after bias/activation (nullptr = none)
  int ld_in2, k_split, ld_res;
  const float *w;       // [K/16][n_total_pad][16], n_total_pad = taps_out * cout_pad
  const float *bias;    // [cout_pad]
  float *out;
  long long rows;
  int in_h, in_w;
  int cin, cout, cout_pad, n_total;
  int k_total;          // cin * taps_in
  int ld_in, ld_out;
  int relu;
  int n_tiles;
  int vec_out;          // output rows 16-byte aligned (ld_out % 4 == 0, aligned base): 16-byte stores
};

template <int MODE, int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(CONV_THREADS, 2) void k_pointwise(PwParams p) {
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MI = WTM / 32, NI = WTN / 32;
  constexpr int A_F4 = BM * (CK / 4), B_F4 = BN * (CK / 4);
  constexpr int A_PER = (A_F4 + CONV_THREADS - 1) / CONV_THREADS;
  constexpr int B_PER = (B_F4 + CONV_THREADS - 1) / CONV_THREADS;
  __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * LDK];
  float *As = lds;
  float *Bs = lds + BM * LDK;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int nt = lid % p.n_tiles;
  const long long m0 = (long long)(lid / p.n_tiles) * BM;
  const int n0 = nt * BN;

  // row base pointers; SPACE2DEPTH rows address the top-left pixel of their 2x2 input block
  const float *a_row[A_PER];
  const float *a_row2[A_PER];
  int a_dst[A_PER];
#pragma unroll
  for (int i = 0; i < A_PER; i++) {
    int idx = tid + i * CONV_THREADS;
    a_row[i] = nullptr;
    a_row2[i] = nullptr;
    a_dst[i] = -1;
    if (idx < A_F4) {
      int row = idx >> 2, q = idx & 3;
      long long m = m0 + row;
      a_dst[i] = row * LDK + q * 4;
      if (m < p.rows) {
        if (MODE == PCP_PW_PLAIN && p.in2) a_row2[i] = p.in2 + m * p.ld_in2 + q * 4;
        if (MODE == PCP_PW_SPACE2DEPTH) {
          int ow = p.in_w >> 1, oh = p.in_h >> 1;
          int ox = (int)(m % ow);
          long long t = m / ow;
          int oy = (int)(t % oh), bb = (int)(t / oh);
          a_row[i] = p.in + ((long long)(bb * p.in_h + 2 * oy) * p.in_w + 2 * ox) * p.ld_in + q * 4;
        } else {
          a_row[i] = p.in + m * p.ld_in + q * 4;
        }
      }
    }
  }
  int b_goff[B_PER], b_dst[B_PER];
#pragma unroll
  for (int i = 0; i < B_PER; i++) {
    int idx = tid + i * CONV_THREADS;
    b_dst[i] = -1;
    b_goff[i] = 0;
    if (idx < B_F4) {
      int n = idx >> 2, q = idx & 3;
      b_goff[i] = (n0 + n) * CK + q * 4;
      b_dst[i] = n * LDK + q * 4;
    }
  }
  const int slices_per_tap = p.cin / CK;
  f32x4 a_reg[A_PER], b_reg[B_PER];
  auto load_slice = [&](int slice) {
    int tap = 0, c0 = slice * CK;
    long long tap_off = 0;
    if (MODE == PCP_PW_SPACE2DEPTH) {
      tap = slice / slices_per_tap;
      c0 = (slice % slices_per_tap) * CK;
      tap_off = ((long long)(tap >> 1) * p.in_w + (tap & 1)) * p.ld_in;
    }
    const bool second = (MODE == PCP_PW_PLAIN) && p.in2 && c0 >= p.k_split;
#pragma unroll
    for (int i = 0; i < A_PER; i++) {
      a_reg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (second) {
        if (a_row2[i]) a_reg[i] = *reinterpret_cast<const f32x4 *>(a_row2[i] + (c0 - p.k_split));
      } else if (a_row[i]) {
        a_reg[i] = *reinterpret_cast<const f32x4 *>(a_row[i] + tap_off + c0);
      }
    }
    const float *bs = p.w + (long long)slice * p.n_total * CK;
#pragma unroll
    for (int i = 0; i < B_PER; i++)
      if (b_dst[i] >= 0) b_reg[i] = *reinterpret_cast<const f32x4 *>(bs + b_goff[i]);
  };
  int a_off[MI], b_off[NI];
#pragma unroll
  for (int i = 0; i < MI; i++) a_off[i] = (wm * WTM + i * 32 + r) * LDK + 4 * h;
#pragma unroll
  for (int j = 0; j < NI; j++) b_off[j] = (wn * WTN + j * 32 + r) * LDK + 4 * h;
  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; i++)
#pragma unroll
    for (int j = 0; j < NI; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

  const int n_slices = p.k_total / CK;
  load_slice(0);
  for (int slice = 0; slice < n_slices; slice++) {
#pragma unroll
    for (int i = 0; i < A_PER; i++)
      if (a_dst[i] >= 0) *reinterpret_cast<f32x4 *>(As + a_dst[i]) = a_reg[i];
#pragma unroll
    for (int i = 0; i < B_PER; i++)
      if (b_dst[i] >= 0) *reinterpret_cast<f32x4 *>(Bs + b_dst[i]) = b_reg[i];
    __syncthreads();
    if (slice + 1 < n_slices) load_slice(slice + 1);
#pragma unroll
    for (int g = 0; g < CK / 8; g++) {
      f32x4 af[MI], bf[NI];
#pragma unroll
      for (int i = 0; i < MI; i++) af[i] = *reinterpret_cast<const f32x4 *>(As + a_off[i] + g * 8);
#pragma unroll
      for (int j = 0; j < NI; j++) bf[j] = *reinterpret_cast<const f32x4 *>(Bs + b_off[j] + g * 8);
#pragma unroll
      for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NI; j++) {
          // the WEIGHTS are the A operand (rows = output channels), the pixels the B operand (columns): a lane ends up with four consecutive
          // channels of one pixel per accumulator quad -> 16-byte stores
          acc[i][j] = mfma32(bf[j].x, af[i].x, acc[i][j]);
          acc[i][j] = mfma32(bf[j].y, af[i].y, acc[i][j]);
          acc[i][j] = mfma32(bf[j].z, af[i].z, acc[i][j]);
          acc[i][j] = mfma32(bf[j].w, af[i].w, acc[i][j]);
        }
    }
    __syncthreads();
  }

  // epilogue: lane (r, h) holds channels nb + 8 q + 4 h + (0..3) of pixel m0 + wm * WTM + i * 32 + r in accumulator quad q
#pragma unroll
  for (int j = 0; j < NI; j++) {
    const int nb = n0 + wn * WTN + j * 32;
    int tap = 0, cb = nb;
    if (MODE == PCP_PW_DEPTH2SPACE) {
      tap = nb / p.cout_pad;
      cb = nb % p.cout_pad;
    }
#pragma unroll
    for (int i = 0; i < MI; i++) {
      const long long m = m0 + wm * WTM + i * 32 + r;
      if (m >= p.rows || nb >= p.n_total) continue;
      long long opix = m;
      if (MODE == PCP_PW_DEPTH2SPACE) {
        int ix = (int)(m % p.in_w);
        long long t = m / p.in_w;
        int iy = (int)(t % p.in_h), bb = (int)(t / p.in_h);
        opix = ((long long)bb * (2 * p.in_h) + 2 * iy + (tap >> 1)) * (2 * p.in_w) + 2 * ix + (tap & 1);
      }
      float *orow = p.out + opix * p.ld_out;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int co = cb + 8 * q + 4 * h;
        if (co >= p.cout) continue;
        const f32x4 bias = *reinterpret_cast<const f32x4 *>(p.bias + co);          // bias is padded to cout_pad
        f32x4 v = f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]} + bias;
        if (p.relu) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        if (MODE == PCP_PW_PLAIN && p.residual) {
          const float *rr = p.residual + m * p.ld_res + co;
          v.x += rr[0];
          if (co + 1 < p.cout) v.y += rr[1];
          if (co + 2 < p.cout) v.z += rr[2];
          if (co + 3 < p.cout) v.w += rr[3];
        }
        if (co + 3 < p.cout && p.vec_out) {
          *reinterpret_cast<f32x4 *>(orow + co) = v;
        } else {
          orow[co] = v.x;
          if (co + 1 < p.cout) orow[co + 1] = v.y;
          if (co + 2 < p.cout) orow[co + 2] = v.z;
          if (co + 3 < p.cout) orow[co + 3] = v.w;
        }
      }
    }
  }
}

template <int S, int TH, int TW, int BN, int WM, int WN>
int launch_conv3(const pcp_conv3x3_t *d, const float *in, const float *w, const float *bias, float *out, hipStream_t st) {
  Conv3Params p;
  p.in = in; p.w = w; p.bias = bias; p.out = out;
  p.batch = d->batch; p.in_h = d->in_h; p.in_w = d->in_w;
  p.out_h = (d->in_h + 2 - 3) / S + 1;
  p.out_w = (d->in_w + 2 - 3) / S + 1;
  p.cin = d->cin; p.cout = d->cout; p.cout_pad = d->cout_pad;
  p.ld_in = d->ld_in; p.ld_out = d->ld_out; p.relu = d->relu;
  p.tiles_x = (p.out_w + TW - 1) / TW;
  p.tiles_y = (p.out_h + TH - 1) / TH;
  p.n_tiles = d->cout_pad / BN;
  p.vec_out = (d->ld_out % 4 == 0 && (((uintptr_t)out) & 15) == 0) ? 1 : 0;
  long long blocks = (long long)d->batch * p.tiles_x * p.tiles_y * p.n_tiles;
  if (blocks <= 0 || blocks > 0x7fffffffLL) return PCP_ERR_ARG;
  hipLaunchKernelGGL((k_conv3x3<S, TH, TW, BN, WM, WN>), dim3((unsigned)blocks), dim3(CONV_THREADS), 0, st, p);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

template <int MODE, int BM, int BN, int WM, int WN>
int launch_pw(const PwParams &p0, hipStream_t st) {
  PwParams p = p0;
  p.n_tiles = (p.n_total + BN - 1) / BN;
  long long blocks = ((p.rows + BM - 1) / BM) * p.n_tiles;
  if (blocks <= 0 || blocks > 0x7fffffffLL) return PCP_ERR_ARG;
  hipLaunchKernelGGL((k_pointwise<MODE, BM, BN, WM, WN>), dim3((unsigned)blocks), dim3(CONV_THREADS), 0, st, p);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

}  // namespace

extern "C" int pcp_conv3x3(const pcp_conv3x3_t *d, const float *in, const float *w_packed, const float *bias, float *out,
                           void *stream_) {
  if (!d || !in || !w_packed || !bias || !out) return PCP_ERR_ARG;
  if (d->cin <= 0 || d->cin % CK != 0 || d->cout <= 0 || d->cout_pad < d->cout || d->cout_pad % 32 != 0) return PCP_ERR_ARG;
  // bias: cout_pad floats, read 16 bytes at a time by the epilogue (pack_conv3x3 pads it; an exact-length or unaligned slice is an error)
  if (d->ld_in % 4 != 0 || (((uintptr_t)in) & 15) || (((uintptr_t)w_packed) & 15) || (((uintptr_t)bias) & 15)) return PCP_ERR_ARG;
  if (d->batch <= 0 || d->in_h <= 0 || d->in_w <= 0) return PCP_ERR_ARG;
  hipStream_t st = (hipStream_t)stream_;
  if (d->stride == 1) {
    if (d->cout_pad % 64 == 0) return launch_conv3<1, 8, 16, 64, 2, 2>(d, in, w_packed, bias, out, st);
    return launch_conv3<1, 8, 16, 32, 4, 1>(d, in, w_packed, bias, out, st);
  } else if (d->stride == 2) {
    if (d->cout_pad % 64 == 0) return launch_conv3<2, 8, 8, 64, 2, 2>(d, in, w_packed, bias, out, st);
    return launch_conv3<2, 8, 16, 32, 4, 1>(d, in, w_packed, bias, out, st);
  }
  return PCP_ERR_UNSUPPORTED;
}

extern "C" int pcp_pointwise(const pcp_pointwise_t *d, const float *in, const float *w_packed, const float *bias, float *out,
                             void *stream_) {
  if (!d || !in || !w_packed || !bias || !out) return PCP_ERR_ARG;
  if (d->cin <= 0 || d->cin % CK != 0 || d->cout <= 0 || d->cout_pad < d->cout || d->cout_pad % 32 != 0) return PCP_ERR_ARG;
  // bias: cout_pad floats, read 16 bytes at a time by the epilogue
  if (d->ld_in % 4 != 0 || (((uintptr_t)in) & 15) || (((uintptr_t)w_packed) & 15) || (((uintptr_t)bias) & 15)) return PCP_ERR_ARG;
  hipStream_t st = (hipStream_t)stream_;
  PwParams p;
  p.in = in; p.w = w_packed; p.bias = bias; p.out = out;
  p.in2 = nullptr; p.residual = nullptr; p.ld_in2 = 0; p.k_split = 0; p.ld_res = 0;
  if (d->mode == PCP_PW_PLAIN) {
    p.in2 = d->in2; p.ld_in2 = d->ld_in2; p.k_split = d->k_split;
    p.residual = d->residual; p.ld_res = d->ld_res;
    if (p.in2 && (p.k_split <= 0 || p.k_split % CK != 0 || p.k_split >= d->cin || (p.ld_in2 & 3) || (((uintptr_t)p.in2) & 15)))
      return PCP_ERR_ARG;
  }
  p.in_h = d->in_h; p.in_w = d->in_w;
  p.cin = d->cin; p.cout = d->cout; p.cout_pad = d->cout_pad;
  p.ld_in = d->ld_in; p.ld_out = d->ld_out; p.relu = d->relu;
  p.n_tiles = 0;
  p.vec_out = (d->ld_out % 4 == 0 && (((uintptr_t)out) & 15) == 0) ? 1 : 0;
  switch (d->mode) {
    case PCP_PW_PLAIN:
      if (d->rows <= 0) return d->rows == 0 ? PCP_OK : PCP_ERR_ARG;
      p.rows = d->rows; p.k_total = d->cin; p.n_total = d->cout_pad;
      if (d->cout_pad % 64 == 0) return launch_pw<PCP_PW_PLAIN, 128, 64, 2, 2>(p, st);
      return launch_pw<PCP_PW_PLAIN, 128, 32, 4, 1>(p, st);
    case PCP_PW_SPACE2DEPTH:
      if (d->batch <= 0 || d->in_h <= 0 || d->in_w <= 0 || (d->in_h & 1) || (d->in_w & 1)) return PCP_ERR_ARG;
      p.rows = (long long)d->batch * (d->in_h / 2) * (d->in_w / 2);
      p.k_total = 4 * d->cin; p.n_total = d->cout_pad;
      if (d->cout_pad % 64 == 0) return launch_pw<PCP_PW_SPACE2DEPTH, 128, 64, 2, 2>(p, st);
      return launch_pw<PCP_PW_SPACE2DEPTH, 128, 32, 4, 1>(p, st);
    case PCP_PW_DEPTH2SPACE:
      if (d->batch <= 0 || d->in_h <= 0 || d->in_w <= 0) return PCP_ERR_ARG;
      p.rows = (long long)d->batch * d->in_h * d->in_w;
      p.k_total = d->cin; p.n_total = 4 * d->cout_pad;
      if (d->cout_pad % 64 == 0) return launch_pw<PCP_PW_DEPTH2SPACE, 128, 64, 2, 2>(p, st);
      return launch_pw<PCP_PW_DEPTH2SPACE, 128, 32, 4, 1>(p, st);
    default:
      return PCP_ERR_UNSUPPORTED;
  }
}
