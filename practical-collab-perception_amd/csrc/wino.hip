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

#ifdef WINO_STAMP
#ifndef WINO_STAMP_S0
#define WINO_STAMP_S0 16
#endif
#ifndef WINO_STAMP_BLOCK
#define WINO_STAMP_BLOCK 0
#endif
__device__ unsigned long long wino_dbg[8 * 8 * 8];          // [slice 0..7][wave][stamp]
#define STAMP(slot)                                                                             \
  do {                                                                                          \
    if (blockIdx.x == WINO_STAMP_BLOCK && s >= WINO_STAMP_S0 && s < WINO_STAMP_S0 + 8 && lane == 0) {                                    \
      unsigned long long t_;                                                                    \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                 \
      wino_dbg[((s - WINO_STAMP_S0) * 8 + wave) * 8 + (slot)] = t_;                                        \
    }                                                                                           \
  } while (0)
#else
#define STAMP(slot)
#endif

namespace {

constexpr int WCK = 8;                    // input channels per slice
constexpr int WLD = 12;                   // padded V row (floats): 16 lanes of a ds_read_b128 group hit 16 distinct slots
constexpr int RLD = 12;                   // padded raw-pixel row (floats): the 4 tiles of a 32-lane ds_read_b32 group hit 32 banks
constexpr int WBN = 64;                   // output channels per workgroup
constexpr int WTHREADS = 512;
constexpr int RAW_W = 18;
constexpr int MS_LD = 33;
constexpr int MS_FLOATS = 16 * 32 * MS_LD;   // 16896 floats: one epilogue chunk (32 tiles x 32 channels x 16 positions)

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

// RT = 32-tile row blocks per workgroup: RT = 2 -> 16x16 output pixels (64 tiles, 1 workgroup per CU, least weight traffic:
// the large-Cin layers); RT = 1 -> 8 rows x 16 columns of pixels (32 tiles, 2 workgroups per CU so one's prologue/epilogue
// hides under the other's main loop: the short-K layers).
template <int RT>
struct WinoCfg {
  static constexpr int TILES = 32 * RT;
  static constexpr int OUT_H = 8 * RT;                       // output rows per workgroup (16 columns always)
  static constexpr int RAW_H = OUT_H + 2;
  static constexpr int RAW_PIX = RAW_H * RAW_W;
  static constexpr int RAW_FLOATS = RAW_PIX * RLD;
  static constexpr int V_FLOATS = 16 * TILES * WLD;
  static constexpr int MAIN_FLOATS = 2 * RAW_FLOATS + 2 * V_FLOATS;
  static constexpr int LDS_FLOATS = MAIN_FLOATS > MS_FLOATS ? MAIN_FLOATS : MS_FLOATS;
  static constexpr int RAW_ITEMS = RAW_PIX * 2;               // float4 items
  static constexpr int RAW_PER = (RAW_ITEMS + WTHREADS - 1) / WTHREADS;
};

template <int RT>
__global__ __launch_bounds__(WTHREADS, (RT == 2 ? 2 : 4)) void k_conv3x3_wino(WinoParams p) {
  using C = WinoCfg<RT>;
  __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];
  float *rawb = lds;                            // [2][RAW_FLOATS]
  float *vb = lds + 2 * C::RAW_FLOATS;          // [2][V_FLOATS]
  float *ms = lds;                              // epilogue chunk (aliases everything; used after the last barrier)

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
  const int oy0 = tile_y * C::OUT_H, ox0 = tile_x * 16;
  const int n0 = nt * WBN;

  // ---- raw patch staging: RAW_PIX pixels x 2 float4.  Loads are UNCONDITIONAL (halo pixels outside the image read a valid
  //      address and are zeroed by a select) so that the number of outstanding VMEM operations is static and hipcc's
  //      counted s_waitcnt never has to wait for the youngest loads. ------------------------------------------------------
  unsigned roff[C::RAW_PER];        // byte offset from p.in (uniform base + 32-bit lane offset -> no 64-bit VALU address math)
  int rdst[C::RAW_PER];
  bool rin[C::RAW_PER];
#pragma unroll
  for (int i = 0; i < C::RAW_PER; i++) {
    int idx = tid + i * WTHREADS;
    roff[i] = 0u;
    rdst[i] = -1;
    rin[i] = false;
    if (idx < C::RAW_ITEMS) {
      int pix = idx >> 1, q = idx & 1;
      int py = pix / RAW_W, px = pix % RAW_W;
      int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
      rdst[i] = pix * RLD + q * 4;
      if (iy >= 0 && iy < p.h && ix >= 0 && ix < p.w) {
        rin[i] = true;
        roff[i] = (unsigned)((((long long)(b * p.h + iy) * p.w + ix) * p.ld_in + q * 4) * 4);
      }
    }
  }
  f32x4 rreg[C::RAW_PER];
  auto raw_load = [&](int slice) {
    const char *base = reinterpret_cast<const char *>(p.in + slice * WCK);       // wave-uniform
#pragma unroll
    for (int i = 0; i < C::RAW_PER; i++) rreg[i] = *reinterpret_cast<const f32x4 *>(base + roff[i]);
  };
  auto raw_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < C::RAW_PER; i++)
      if (rdst[i] >= 0) {
        f32x4 v = rreg[i];
        if (!rin[i]) v = f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4 *>(rawb + buf * C::RAW_FLOATS + rdst[i]) = v;
      }
  };

  // ---- input transform V = B^T d B, LDS -> LDS: item = (tile, channel), tiles are 8 per row.  RT == 2: all eight waves,
  //      de-phased (waves 0-3 before their multiply, waves 4-7 after); RT == 1: waves 0-3 (256 items), before the multiply.
  //      (A 16-byte-per-lane variant on two waves was measured 3-6 % slower on the long-K layers: on gfx950 the fp32 MFMA and
  //      the VALU contend for the SIMD and the two transform waves become the critical path.) -------------------------------
  const int t_c = tid & 7, t_t = tid >> 3;
  const bool t_on = t_t < C::TILES;
  const int t_src = ((t_t >> 3) * 2 * RAW_W + (t_t & 7) * 2) * RLD + t_c;
  const int t_dst = t_t * WLD + t_c;
  const bool xf_first = wave < 4;
  const bool xf_last = (RT == 2) && wave >= 4;
  auto transform = [&](int rbuf, int vbuf) {
    if (!t_on) return;
    const float *src = rawb + rbuf * C::RAW_FLOATS + t_src;
    float *dst = vb + vbuf * C::V_FLOATS + t_dst;
    // w = d B (row transform, one input row at a time), then V = B^T w
    float w[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++) {
      const float d0 = src[(a * RAW_W + 0) * RLD], d1 = src[(a * RAW_W + 1) * RLD];
      const float d2 = src[(a * RAW_W + 2) * RLD], d3 = src[(a * RAW_W + 3) * RLD];
      w[a][0] = d0 - d2;
      w[a][1] = d1 + d2;
      w[a][2] = d2 - d1;
      w[a][3] = d1 - d3;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      dst[(0 * 4 + j) * C::TILES * WLD] = w[0][j] - w[2][j];
      dst[(1 * 4 + j) * C::TILES * WLD] = w[1][j] + w[2][j];
      dst[(2 * 4 + j) * C::TILES * WLD] = w[2][j] - w[1][j];
      dst[(3 * 4 + j) * C::TILES * WLD] = w[1][j] - w[3][j];
    }
  };

  // ---- B fragments (transformed weights) from global: positions 2w, 2w+1; two 32-channel column tiles each.  Two register
  //      sets: the next slice's fragments are requested at the TOP of an iteration so the barrier at its end never waits
  //      on memory latency. -------------------------------------------------------------------------------------------------
  const float *ubase = p.u + ((long long)(2 * wave) * p.cout_pad + n0) * WCK;        // wave-uniform
  const unsigned u_lane = (unsigned)((r * WCK + 4 * h) * 4);                             // bytes
  const long long u_pos = (long long)p.cout_pad * WCK;            // floats between positions
  const long long u_slice = 16 * u_pos;
  f32x4 b0[2][2], b1[2][2];
  auto b_load = [&](int slice, f32x4 (&dstb)[2][2]) {
    const float *s = ubase + slice * u_slice;
#pragma unroll
    for (int pi = 0; pi < 2; pi++)
#pragma unroll
      for (int ct = 0; ct < 2; ct++)
        dstb[pi][ct] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(s + pi * u_pos + ct * 32 * WCK) + u_lane);
  };

  f32x16 acc[2][RT][2];
#pragma unroll
  for (int pi = 0; pi < 2; pi++)
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
      for (int ct = 0; ct < 2; ct++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[pi][rt][ct][e] = 0.f;

  const int a_off = ((2 * wave) * C::TILES + r) * WLD + 4 * h;
  auto multiply = [&](int vbuf, const f32x4 (&bf)[2][2]) {
    const float *vsrc = vb + vbuf * C::V_FLOATS + a_off;
#pragma unroll
    for (int pi = 0; pi < 2; pi++)
#pragma unroll
      for (int rt = 0; rt < RT; rt++) {
        f32x4 a = *reinterpret_cast<const f32x4 *>(vsrc + (pi * C::TILES + rt * 32) * WLD);
#pragma unroll
        for (int ct = 0; ct < 2; ct++) {
          acc[pi][rt][ct] = mfma32w(a.x, bf[pi][ct].x, acc[pi][rt][ct]);
          acc[pi][rt][ct] = mfma32w(a.y, bf[pi][ct].y, acc[pi][rt][ct]);
          acc[pi][rt][ct] = mfma32w(a.z, bf[pi][ct].z, acc[pi][rt][ct]);
          acc[pi][rt][ct] = mfma32w(a.w, bf[pi][ct].w, acc[pi][rt][ct]);
        }
      }
  };

  const int n_slices = p.cin / WCK;
  const int last = n_slices - 1;
  // One pipeline step: V(s) x B(s) on the matrix pipe while V(s+1) is produced; at the END of the multiply the step stores
  // raw(s+2) (fetched during the previous step) to LDS and requests B(s+1) and raw(s+3) -- every global load has a whole step
  // to land and no wait ever targets a just-issued load.  Indices are clamped instead of predicated (static VMEM counts).
  // SIMD partners are de-phased: waves 0-3 transform first, waves 4-7 multiply first (one copy of the MFMA code:
  // duplicating it in two branches makes the register allocator copy the accumulators).
  auto step = [&](int s, const f32x4 (&bcur)[2][2], f32x4 (&bnxt)[2][2]) {
    const int cur = s & 1, nxt = cur ^ 1;
    const bool has1 = s + 1 < n_slices, has2 = s + 2 < n_slices;
    STAMP(0);
    if (has1 && xf_first) transform(nxt, nxt);
    STAMP(1);
    multiply(cur, bcur);
    STAMP(2);
    if (has2) raw_store(cur);           // raw[cur] was consumed by transform(s) one step ago; rreg holds raw(s+2)
    b_load(min(s + 1, last), bnxt);
    raw_load(min(s + 3, last));
    STAMP(3);
    if (has1 && xf_last) transform(nxt, nxt);
    STAMP(4);
    __syncthreads();
    STAMP(5);
  };

  // ---- prologue: raw(0), raw(1) -> LDS; V(0); rreg <- raw(2); B(0) -------------------------------------------------------------
  raw_load(0);
  b_load(0, b0);
  raw_store(0);
  raw_load(min(1, last));
  __syncthreads();
  transform(0, 0);
  if (n_slices > 1) raw_store(1);
  raw_load(min(2, last));
  __syncthreads();

  // (static s_setprio for either half of the block was measured: no gain -- fp32 MFMA and VALU of SIMD partners serialise)
  for (int s = 0; s < n_slices; s += 2) {
    step(s, b0, b1);
    if (s + 1 < n_slices) step(s + 1, b1, b0);
  }

  // ---- epilogue: (row tile, column tile) chunks through LDS ------------------------------------------------------------------
#pragma unroll
  for (int rt = 0; rt < RT; rt++) {
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

template <int RT>
int launch_wino(const pcp_conv3x3_t *d, const float *in, const float *u, const float *bias, float *out, hipStream_t st) {
  WinoParams p;
  p.in = in; p.u = u; p.bias = bias; p.out = out;
  p.batch = d->batch; p.h = d->in_h; p.w = d->in_w;
  p.cin = d->cin; p.cout = d->cout; p.cout_pad = d->cout_pad;
  p.ld_in = d->ld_in; p.ld_out = d->ld_out; p.relu = d->relu;
  p.tiles_x = (d->in_w + 15) / 16;
  p.tiles_y = (d->in_h + WinoCfg<RT>::OUT_H - 1) / WinoCfg<RT>::OUT_H;
  p.n_spatial = d->batch * p.tiles_x * p.tiles_y;
  long long blocks = (long long)p.n_spatial * (d->cout_pad / WBN);
  if (blocks <= 0 || blocks > 0x7fffffffLL) return PCP_ERR_ARG;
  hipLaunchKernelGGL(k_conv3x3_wino<RT>, dim3((unsigned)blocks), dim3(WTHREADS), 0, st, p);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

}  // namespace

#ifdef WINO_STAMP
extern "C" int pcp_debug_read(void *dst, size_t bytes) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(wino_dbg), bytes) == hipSuccess ? 0 : 3;
}
#endif

// long-K layers: 64-tile workgroups (half the weight traffic per output); short-K layers: 32-tile workgroups, two per CU
static int wino_variant(const pcp_conv3x3_t *d) {
  const long long wg64 = (long long)d->batch * ((d->in_w + 15) / 16) * ((d->in_h + 15) / 16) * (d->cout_pad / WBN);
  return (d->cin >= 256 && wg64 >= 256) ? 2 : 1;
}

extern "C" int pcp_conv3x3_winograd(const pcp_conv3x3_t *d, const float *in, const float *u_packed, const float *bias, float *out,
                                    void *stream_) {
  if (!d || !in || !u_packed || !bias || !out) return PCP_ERR_ARG;
  if (d->stride != 1) return PCP_ERR_UNSUPPORTED;
  if (d->cin <= 0 || d->cin % WCK != 0 || d->cout <= 0 || d->cout_pad < d->cout || d->cout_pad % WBN != 0) return PCP_ERR_ARG;
  if (d->ld_in % 4 != 0 || (((uintptr_t)in) & 15) || (((uintptr_t)u_packed) & 15)) return PCP_ERR_ARG;
  if (d->batch <= 0 || d->in_h <= 0 || d->in_w <= 0) return PCP_ERR_ARG;
  hipStream_t st = (hipStream_t)stream_;
  if (wino_variant(d) == 2) return launch_wino<2>(d, in, u_packed, bias, out, st);
  return launch_wino<1>(d, in, u_packed, bias, out, st);
}

extern "C" int pcp_conv3x3_winograd_plan(const pcp_conv3x3_t *d, int32_t *variant, double *executed_flops) {
  if (!d || d->stride != 1 || d->cin <= 0 || d->cin % WCK != 0 || d->cout_pad % WBN != 0) return PCP_ERR_ARG;
  const int rt = wino_variant(d);
  if (variant) *variant = rt;
  if (executed_flops) {
    // every workgroup multiplies [32 * rt tiles x cin] x [cin x 64] at each of the 16 Winograd positions (padding tiles included)
    const long long tiles_y = (d->in_h + 8 * rt - 1) / (8 * rt), tiles_x = (d->in_w + 15) / 16;
    const double wgs = (double)d->batch * tiles_x * tiles_y * (d->cout_pad / WBN);
    *executed_flops = wgs * 2.0 * 16.0 * (32.0 * rt) * d->cin * WBN;
  }
  return PCP_OK;
}
