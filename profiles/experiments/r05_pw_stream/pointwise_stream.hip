// a6 -- the backbone's deblocks (1x1 conv, Conv2d k2 s2, ConvTranspose2d k2 s2: base_bev_backbone.py:48-69) as a STREAMING product (round 5).
//
// These layers do 32 - 86 flops per byte they move: next to the machine balance (157 TFLOP/s fp32 MFMA over ~4.5 TB/s), so every pixel row
// may be read once, every output row written once, and the matrix pipe has to stay busy in between.  k_pointwise (conv.hip) is a tiled GEMM
// with 16-deep K slices staged through LDS between two barriers and re-reads the pixel rows once per 64 output channels: it measures
// 18 - 36 TFLOP/s and 0.5 - 1.7 TB/s on them.  Here the roles are turned around:
//
//   * the WEIGHTS of one 128-channel output block (K x 128 floats, 64 - 128 KB) are copied into LDS once per workgroup, which then stays on
//     its CU and streams pixel tiles past them (one workgroup of eight waves per CU);
//   * a WAVE owns a tile of 32 pixels x 128 output channels (four 32x32 accumulators) and is autonomous: its pixel rows go from HBM straight
//     into the MFMA operand registers (lane (r, h) reads 16 bytes of row r at k = 8 j + 4 h -- no LDS store, no VALU, no barrier in the
//     loop; on this chip VALU / LDS-write work of any wave of a SIMD displaces fp32 MFMAs, DESIGN_HISTORY "what bounds the fused fp32
//     Winograd kernels"), the weight fragments come from LDS with ds_read_b128 (row pitch = 4 mod 64 floats: conflict free over the
//     instruction's lane groups), 16 MFMAs per 16-byte row load and four LDS reads;
//   * the (tile, 64-deep K chunk) sequence of a wave is ONE software pipeline: the next chunk -- of this tile or of the wave's next tile --
//     is requested before the current chunk's 128 MFMAs, the epilogue (bias, ReLU, sixteen 16-byte stores per lane) runs with those loads
//     in flight.
// The weights are the A operand (rows = output channels), so a lane ends up with four consecutive channels of one pixel per accumulator
// quad, as in k_pointwise.  Same products; the K order differs from k_pointwise's (sums agree to fp32 rounding, tests hold both to torch).
#include "pcp_common.h"
#include <cstdlib>
#include <cstring>

namespace {

constexpr int PS_WAVES = 8;
constexpr int PS_THREADS = PS_WAVES * 64;
constexpr int PS_BN = 128;                 // output channels per workgroup (one LDS weight image)
constexpr int PS_TM = 32;                  // pixels per wave tile

struct PsParams {
  const float *in;
  const float *w;       // [K/16][n_total][16]
  const float *bias;    // [cout_pad]
  float *out;
  long long rows, m_tiles;
  int in_h, in_w;
  int cin, cout, cout_pad, n_total, k_total;
  int ld_in, ld_out;
  int relu, vec_out;
  int n_tiles, wgs_per_nt;
};

__device__ __forceinline__ f32x16 mfma32s(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// CH = K per pipeline chunk (CH / 8 16-byte loads per lane), NCH = chunks per tile (even: a tile starts in register buffer 0); K = CH * NCH
template <int MODE, int CH, int NCH>
__global__ __launch_bounds__(PS_THREADS, 1) void k_pw_stream(PsParams p) {
  static_assert(NCH % 2 == 0, "chunk parity");
  constexpr int K = CH * NCH;
  constexpr int S = (K + 63) / 64 * 64 + 4;                       // weight row pitch in LDS (floats): = 4 (mod 64)
  __shared__ __attribute__((aligned(16))) float wl[PS_BN * S + PS_BN];
  float *bl = wl + PS_BN * S;                                     // this block's 128 bias values (the epilogue reads them with ds_read_b128:
                                                                  // a global load there would wait behind the prefetched pixel rows)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int nt = blockIdx.x % p.n_tiles;
  const int slot = blockIdx.x / p.n_tiles;
  const int n0 = nt * PS_BN;

  // ---- this block's weights -> LDS, [n][S] -----------------------------------------------------------------------------------------------
  {
    const int items = (K >> 4) * PS_BN * 4;                       // 16-byte pieces
    for (int idx = tid; idx < items; idx += PS_THREADS) {
      const int q = idx & 3, n = (idx >> 2) & (PS_BN - 1), s16 = idx >> 9;
      const f32x4 v = *reinterpret_cast<const f32x4 *>(p.w + ((long long)s16 * p.n_total + n0 + n) * 16 + q * 4);
      *reinterpret_cast<f32x4 *>(wl + n * S + s16 * 16 + q * 4) = v;
    }
  }
  if (tid < PS_BN) {
    const int cb0 = (MODE == PCP_PW_DEPTH2SPACE) ? n0 % p.cout_pad : n0;
    bl[tid] = p.bias[cb0 + tid];                                  // bias is padded to cout_pad
  }
  __syncthreads();

  const long long t_stride = (long long)PS_WAVES * p.wgs_per_nt;
  long long t = slot + (long long)wave * p.wgs_per_nt;
  if (t >= p.m_tiles) return;

  // pixel row of this lane in a tile (clamped: lanes past the end read the last row and store nothing)
  auto row_ptr = [&](long long tile) -> const float * {
    long long m = tile * PS_TM + r;
    if (m >= p.rows) m = p.rows - 1;
    if (MODE == PCP_PW_SPACE2DEPTH) {
      const int ow = p.in_w >> 1, oh = p.in_h >> 1;
      const int ox = (int)(m % ow);
      const long long q = m / ow;
      const int oy = (int)(q % oh), bb = (int)(q / oh);
      return p.in + ((long long)(bb * p.in_h + 2 * oy) * p.in_w + 2 * ox) * p.ld_in + 4 * h;
    }
    return p.in + m * p.ld_in + 4 * h;
  };
  // wave-uniform offset of chunk c inside a row
  long long coff[NCH];
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    coff[c] = (long long)c * CH;
    if (MODE == PCP_PW_SPACE2DEPTH) {
      const int k0 = c * CH, tap = k0 / p.cin, c0 = k0 % p.cin;
      coff[c] = ((long long)(tap >> 1) * p.in_w + (tap & 1)) * p.ld_in + c0;
    }
  }

  f32x4 a[2][CH / 8];
  f32x16 acc[4];
#pragma unroll
  for (int cb = 0; cb < 4; cb++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc[cb][e] = 0.f;
  const float *wbase = wl + r * S + 4 * h;

  const float *row = row_ptr(t);
#pragma unroll
  for (int jj = 0; jj < CH / 8; jj++) a[0][jj] = *reinterpret_cast<const f32x4 *>(row + coff[0] + jj * 8);

  // ---- the wave's (tile, chunk) pipeline: the loads of the next chunk -- of this tile or of the wave's next tile -- are issued before the
  // current chunk's MFMAs; always issued (the last tile re-reads its own first chunk), so the body is one straight line and the compiler's
  // vmcnt waits leave exactly the prefetch in flight ---------------------------------------------------------------------------------------
  while (true) {
    const long long tn = t + t_stride;
    const bool more = tn < p.m_tiles;
    const float *rown = row_ptr(more ? tn : t);
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      const float *src = (c + 1 < NCH) ? row + coff[(c + 1) % NCH] : rown + coff[0];
#pragma unroll
      for (int jj = 0; jj < CH / 8; jj++) a[(c + 1) & 1][jj] = *reinterpret_cast<const f32x4 *>(src + jj * 8);
      const float *wk = wbase + c * CH;
      // fenced blocks of 16 MFMAs: the weight fragments of block jj + 1 are requested in front of block jj's MFMAs (without the fences the
      // scheduler hoists every LDS read of the tile to its top and spills)
      f32x4 wf[2][4];
#pragma unroll
      for (int cb = 0; cb < 4; cb++) wf[0][cb] = *reinterpret_cast<const f32x4 *>(wk + cb * 32 * S);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int jj = 0; jj < CH / 8; jj++) {
        if (jj + 1 < CH / 8) {
#pragma unroll
          for (int cb = 0; cb < 4; cb++) wf[(jj + 1) & 1][cb] = *reinterpret_cast<const f32x4 *>(wk + cb * 32 * S + (jj + 1) * 8);
        }
        const f32x4 av = a[c & 1][jj];
        // k component outermost: four independent accumulators between two MFMAs on the same one
#pragma unroll
        for (int kk = 0; kk < 4; kk++)
#pragma unroll
          for (int cb = 0; cb < 4; cb++) acc[cb] = mfma32s(wf[jj & 1][cb][kk], av[kk], acc[cb]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // epilogue: bias, ReLU, stores; the next tile's first chunk is in flight
    {
      const long long m = t * PS_TM + r;
      int tap = 0, cbase = n0;
      long long opix = m;
      if (MODE == PCP_PW_DEPTH2SPACE) {
        tap = n0 / p.cout_pad;
        cbase = n0 % p.cout_pad;
        const int ix = (int)(m % p.in_w);
        const long long q = m / p.in_w;
        const int iy = (int)(q % p.in_h), bb = (int)(q / p.in_h);
        opix = ((long long)bb * (2 * p.in_h) + 2 * iy + (tap >> 1)) * (2 * p.in_w) + 2 * ix + (tap & 1);
      }
      float *orow = p.out + opix * p.ld_out + cbase + 4 * h;
      const bool live = m < p.rows;
      const bool whole = p.vec_out && cbase + PS_BN <= p.cout;      // wave-uniform: every channel of the block exists, rows are 16-byte aligned
      const float floor_v = p.relu ? 0.f : -__builtin_inff();        // ReLU without a branch per quad
      f32x4 v[4][4];
#pragma unroll
      for (int cb = 0; cb < 4; cb++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          v[cb][q] = f32x4{acc[cb][4 * q], acc[cb][4 * q + 1], acc[cb][4 * q + 2], acc[cb][4 * q + 3]} +
                     *reinterpret_cast<const f32x4 *>(bl + cb * 32 + 8 * q + 4 * h);
          acc[cb][4 * q] = 0.f; acc[cb][4 * q + 1] = 0.f; acc[cb][4 * q + 2] = 0.f; acc[cb][4 * q + 3] = 0.f;
          v[cb][q].x = fmaxf(v[cb][q].x, floor_v); v[cb][q].y = fmaxf(v[cb][q].y, floor_v);
          v[cb][q].z = fmaxf(v[cb][q].z, floor_v); v[cb][q].w = fmaxf(v[cb][q].w, floor_v);
        }
      if (whole) {
        if (live) {
#pragma unroll
          for (int cb = 0; cb < 4; cb++)
#pragma unroll
            for (int q = 0; q < 4; q++) *reinterpret_cast<f32x4 *>(orow + cb * 32 + 8 * q) = v[cb][q];
        }
      } else if (live) {
#pragma unroll
        for (int cb = 0; cb < 4; cb++)
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int co = cbase + cb * 32 + 8 * q + 4 * h;
            float *o = orow + cb * 32 + 8 * q;
            if (co < p.cout) o[0] = v[cb][q].x;
            if (co + 1 < p.cout) o[1] = v[cb][q].y;
            if (co + 2 < p.cout) o[2] = v[cb][q].z;
            if (co + 3 < p.cout) o[3] = v[cb][q].w;
          }
      }
    }
    if (!more) break;
    t = tn;
    row = rown;
  }
}

int cu_count() {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, v = 0;
    n_cu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  return n_cu;
}

template <int MODE>
int launch_ps(const PsParams &p, hipStream_t st) {
  const dim3 grid((unsigned)(p.wgs_per_nt * p.n_tiles));
  switch (p.k_total) {
    case 64: hipLaunchKernelGGL((k_pw_stream<MODE, 32, 2>), grid, dim3(PS_THREADS), 0, st, p); break;
    case 128: hipLaunchKernelGGL((k_pw_stream<MODE, 64, 2>), grid, dim3(PS_THREADS), 0, st, p); break;
    case 192: hipLaunchKernelGGL((k_pw_stream<MODE, 32, 6>), grid, dim3(PS_THREADS), 0, st, p); break;
    case 256: hipLaunchKernelGGL((k_pw_stream<MODE, 64, 4>), grid, dim3(PS_THREADS), 0, st, p); break;
    default: return PCP_ERR_UNSUPPORTED;
  }
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

}  // namespace

// 0: the streaming kernel took the launch; 1: shape not covered (the caller runs k_pointwise); < 0: error.
// PCP_PW_ALGO=tile sends everything to k_pointwise (A/B runs, tests).
int pcp_pointwise_stream_try(const pcp_pointwise_t *d, const float *in, const float *w_packed, const float *bias, float *out, void *stream_) {
  const char *e = getenv("PCP_PW_ALGO");              // read per call: the tests switch it inside one process
  if (e && !strcmp(e, "tile")) return 1;
  if (d->mode == PCP_PW_PLAIN && (d->in2 || d->residual)) return 1;
  if (d->cout_pad % PS_BN != 0 || d->cin % 64 != 0) return 1;
  PsParams p;
  p.in = in; p.w = w_packed; p.bias = bias; p.out = out;
  p.in_h = d->in_h; p.in_w = d->in_w;
  p.cin = d->cin; p.cout = d->cout; p.cout_pad = d->cout_pad;
  p.ld_in = d->ld_in; p.ld_out = d->ld_out; p.relu = d->relu;
  p.vec_out = (d->ld_out % 4 == 0 && (((uintptr_t)out) & 15) == 0) ? 1 : 0;
  switch (d->mode) {
    case PCP_PW_PLAIN:
      p.rows = d->rows; p.k_total = d->cin; p.n_total = d->cout_pad;
      break;
    case PCP_PW_SPACE2DEPTH:
      if (d->batch <= 0 || d->in_h <= 0 || d->in_w <= 0 || (d->in_h & 1) || (d->in_w & 1)) return PCP_ERR_ARG;
      p.rows = (long long)d->batch * (d->in_h / 2) * (d->in_w / 2);
      p.k_total = 4 * d->cin; p.n_total = d->cout_pad;
      break;
    case PCP_PW_DEPTH2SPACE:
      if (d->batch <= 0 || d->in_h <= 0 || d->in_w <= 0) return PCP_ERR_ARG;
      p.rows = (long long)d->batch * d->in_h * d->in_w;
      p.k_total = d->cin; p.n_total = 4 * d->cout_pad;
      break;
    default:
      return 1;
  }
  if (p.k_total > 256) return 1;                    // the weight image of one block must fit the CU's LDS (K in {64, 128, 192, 256})
  if (p.rows <= 0) return p.rows == 0 ? PCP_OK : PCP_ERR_ARG;
  p.n_tiles = p.n_total / PS_BN;
  p.m_tiles = (p.rows + PS_TM - 1) / PS_TM;
  long long per_nt = cu_count() / p.n_tiles;
  if (per_nt < 1) per_nt = 1;
  const long long need = (p.m_tiles + PS_WAVES - 1) / PS_WAVES;
  p.wgs_per_nt = (int)(need < per_nt ? need : per_nt);
  hipStream_t st = (hipStream_t)stream_;
  switch (d->mode) {
    case PCP_PW_PLAIN: return launch_ps<PCP_PW_PLAIN>(p, st);
    case PCP_PW_SPACE2DEPTH: return launch_ps<PCP_PW_SPACE2DEPTH>(p, st);
    default: return launch_ps<PCP_PW_DEPTH2SPACE>(p, st);
  }
}
