// 3x3 convolution (stride 1 / 2, padding 1) with fp32 inputs / outputs on the BF16 matrix cores by the split-operand scheme
// "bf16x3":  x = hi + lo (two bf16, 16 mantissa bits together),  x*w ~= hi_x*hi_w + hi_x*lo_w + lo_x*hi_w  (the dropped lo*lo term is
// 2^-16 relative), every partial product exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16.  Three bf16 MFMAs (32 cycles
// each, K = 16) replace eight fp32 MFMAs (64 cycles each, K = 2): 5.3x less matrix time than the fp32 implicit GEMM and 2.4x less
// than the fp32 Winograd kernel, at ~1e-5 relative error -- tighter than the TF32 (10-bit mantissa) arithmetic cuDNN uses by default
// for the reference's convolutions on NVIDIA GPUs (torch.backends.cudnn.allow_tf32 = True), and 100x inside the 1e-3 parity bar.
// OPT-IN (PCP_CONV_ALGO=bf16x3): the default build computes every convolution in fp32 (conv.hip / wino.hip).
//
// Replaces the same reference layers as pcp_conv3x3 (Conv2d 3x3 + folded BatchNorm + ReLU).
// Workgroup (4 waves) = 16x16 (stride 1) or 8x16 (stride 2) output pixels x 64 output channels; per 16-channel slice the input patch
// with halo is converted ONCE to (hi, lo) bf16 while it is staged in LDS and re-read for all nine taps; the slice's weights arrive
// pre-split from the host pack.  LDS images are [k-half][pixel | cout][8 bf16]: the 16 lanes of a ds_read_b128 group read 256
// contiguous bytes (conflict free, no padding).  Next slice: global -> registers under the current slice's 108 MFMAs per wave.
#include "pcp_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int B3_THREADS = 256;
constexpr int B3_TW = 16;
constexpr int B3_CK = 16;           // input channels per slice
constexpr int B3_BN = 64;           // output channels per workgroup
constexpr int B3_WSLICE = 2 * 9 * 2 * B3_BN * 8;       // bf16 elements of one (slice, cout tile): [hi|lo][tap][k-half][64][8]

struct B3Params {
  const float *in;
  const __bf16 *w;        // [cin/16][cout_pad/64][hi|lo][9][2][64][8]
  const float *bias;
  float *out;
  int batch, in_h, in_w, out_h, out_w, cin, cout, cout_pad, ld_in, ld_out, relu;
  int tiles_y, tiles_x, n_tiles;
};

__device__ __forceinline__ int xcd_remap_b3(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// S: stride; TH: output rows per workgroup (8 * M-tiles per wave); DB: double-buffered LDS (one barrier per slice, one workgroup per
// CU with a 128-pixel x 64-channel register tile per wave) or single-buffered (two workgroups per CU overlap each other)
// TERMS: 3 = split products hi*hi + hi*lo + lo*hi ("bf16x3", ~1e-5); 1 = plain bf16 products hi*hi (8 mantissa bits, ~3e-3: the
// mixed-precision TRAINING mode named by BASELINE.json's config 5; never used for inference parity)
template <int S, int TH, bool DB, int TERMS>
__global__ __launch_bounds__(B3_THREADS, DB ? 1 : 2) void k_conv3x3_bf16x3(B3Params p) {
  constexpr int PH = (TH - 1) * S + 3, PW = (B3_TW - 1) * S + 3;
  constexpr int NPIX = PH * PW;
  constexpr int MT = TH / 8;                       // 32-pixel M-tiles (2 rows x 16 columns) per wave
  constexpr int NPL = (NPIX * 4 + B3_THREADS - 1) / B3_THREADS;       // float4 patch loads per thread
  constexpr int NWL = B3_WSLICE / 8 / B3_THREADS;                     // 16-byte weight loads per thread (9)
  constexpr int NBUF = DB ? 2 : 1;
  static_assert(B3_WSLICE / 8 % B3_THREADS == 0, "weight slice is a whole number of 16-byte loads per thread");
  __shared__ __attribute__((aligned(16))) __bf16 patch[NBUF][2][2][NPIX][8];    // [buf][hi|lo][k-half][pixel][8]
  __shared__ __attribute__((aligned(16))) __bf16 wts[NBUF][B3_WSLICE];          // [buf][hi|lo][tap][k-half][64][8]

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int n_ct = p.cout_pad / B3_BN;
  int lid = xcd_remap_b3(blockIdx.x, gridDim.x);
  const int ct = lid % n_ct;                      // cout tiles of one spatial tile are neighbours: they share the patch in L2
  lid /= n_ct;
  const int tx = lid % p.tiles_x;
  lid /= p.tiles_x;
  const int ty = lid % p.tiles_y;
  const int b = lid / p.tiles_y;
  const int oy0 = ty * TH, ox0 = tx * B3_TW;
  const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;

  // ---- per-thread staging addresses (slice independent) ------------------------------------------------------------------------
  unsigned poff[NPL];          // float offsets (< 2^32 for every map of the path)
  int pdst[NPL];               // element offset inside one (hi | lo) image; bit 30 set: halo pixel outside the map (store zeros)
#pragma unroll
  for (int u = 0; u < NPL; ++u) {
    const int idx = tid + u * B3_THREADS;
    const int pix = idx >> 2, q = idx & 3;
    const int iy = iy0 + pix / PW, ix = ix0 + pix % PW;
    const bool in = idx < NPIX * 4 && iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w;
    poff[u] = in ? (unsigned)((((long long)b * p.in_h + iy) * p.in_w + ix) * p.ld_in + q * 4) : 0u;
    pdst[u] = idx < NPIX * 4 ? ((((q >> 1) * NPIX + pix) * 8 + (q & 1) * 4) | (in ? 0 : (1 << 30))) : -1;
  }
  const int n_slices = p.cin / B3_CK;
  const __bf16 *wsrc = p.w + (long long)ct * B3_WSLICE;
  const long long wstep = (long long)n_ct * B3_WSLICE;

  f32x4 preg[NPL];
  f32x4 wreg[NWL];            // raw 16-byte chunks of the packed bf16 weights
  auto prefetch = [&](int s) {
    const float *base = p.in + s * B3_CK;
#pragma unroll
    for (int u = 0; u < NPL; ++u) preg[u] = *reinterpret_cast<const f32x4 *>(base + poff[u]);
    const f32x4 *ws = reinterpret_cast<const f32x4 *>(wsrc + s * wstep);
#pragma unroll
    for (int u = 0; u < NWL; ++u) wreg[u] = ws[tid + u * B3_THREADS];
  };
  auto commit = [&](int buf) {
    __bf16 *ph = &patch[buf][0][0][0][0], *pl = &patch[buf][1][0][0][0];
#pragma unroll
    for (int u = 0; u < NPL; ++u) {
      f32x4 v = preg[u];
      if (pdst[u] & (1 << 30)) v = f32x4{0.f, 0.f, 0.f, 0.f};
      bf16x4 hi, lo;
      hi[0] = (__bf16)v.x; hi[1] = (__bf16)v.y; hi[2] = (__bf16)v.z; hi[3] = (__bf16)v.w;
      lo[0] = (__bf16)(v.x - (float)hi[0]); lo[1] = (__bf16)(v.y - (float)hi[1]);
      lo[2] = (__bf16)(v.z - (float)hi[2]); lo[3] = (__bf16)(v.w - (float)hi[3]);
      if (pdst[u] >= 0) {
        const int o = pdst[u] & ~(1 << 30);
        *reinterpret_cast<bf16x4 *>(ph + o) = hi;
        if (TERMS == 3) *reinterpret_cast<bf16x4 *>(pl + o) = lo;
      }
    }
    f32x4 *wd = reinterpret_cast<f32x4 *>(&wts[buf][0]);
#pragma unroll
    for (int u = 0; u < NWL; ++u) wd[tid + u * B3_THREADS] = wreg[u];
  };

  f32x16 acc[MT][2];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;

  // lane's pixel inside an M-tile: row r >> 4, column r & 15
  int abase[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) abase[m] = ((2 * (wave * MT + m) + (r >> 4)) * S) * PW + (r & 15) * S;

  // One multiply step = (tap, M-tile): 2 A fragments (hi, lo) x the tap's 4 B fragments -> 6 MFMAs.  The fragments of step i + 1 are
  // requested BEFORE the MFMAs of step i are issued (register double buffering), so no MFMA waits on an LDS read it has just issued
  // (the straightforward loop waited lgkmcnt(0) in front of every group of six: the matrix pipe idled ~40 % of the time).
  auto compute = [&](int buf) {
    const __bf16 *wb = &wts[buf][0];
    const __bf16 *pa_h = &patch[buf][0][h][0][0], *pa_l = &patch[buf][1][h][0][0];
    auto load_b = [&](int tap, bf16x8 (&bh)[2], bf16x8 (&bl)[2]) {
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        bh[n] = *reinterpret_cast<const bf16x8 *>(wb + (((0 * 9 + tap) * 2 + h) * B3_BN + n * 32 + r) * 8);
        bl[n] = *reinterpret_cast<const bf16x8 *>(wb + (((1 * 9 + tap) * 2 + h) * B3_BN + n * 32 + r) * 8);
      }
    };
    auto load_a = [&](int tap, int m, bf16x8 &ah, bf16x8 &al) {
      const int o = (abase[m] + (tap / 3) * PW + tap % 3) * 8;
      ah = *reinterpret_cast<const bf16x8 *>(pa_h + o);
      al = *reinterpret_cast<const bf16x8 *>(pa_l + o);
    };
    bf16x8 bh[2][2], bl[2][2], ah[2], al[2];
    load_b(0, bh[0], bl[0]);
    load_a(0, 0, ah[0], al[0]);
#pragma unroll
    for (int i = 0; i < 9 * MT; ++i) {
      const int tap = i / MT, m = i % MT;
      const int cb = tap & 1, ca = i & 1;
      if (i + 1 < 9 * MT) {
        const int ntap = (i + 1) / MT, nm = (i + 1) % MT;
        if (nm == 0) load_b(ntap, bh[ntap & 1], bl[ntap & 1]);
        load_a(ntap, nm, ah[ca ^ 1], al[ca ^ 1]);
      }
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        if (TERMS == 3) {
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ca], bh[cb][n], acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ca], bl[cb][n], acc[m][n], 0, 0, 0);
        }
        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ca], bh[cb][n], acc[m][n], 0, 0, 0);
      }
    }
  };

  prefetch(0);
  if (DB) {
    commit(0);
    __syncthreads();
    for (int s = 0; s < n_slices; ++s) {
      const bool more = s + 1 < n_slices;
#if defined(B3_DIAG_ONE_SLICE)
      compute(0);
#elif defined(B3_DIAG_NO_COMMIT)
      if (more) prefetch(s + 1);
      compute(0);
      asm volatile("" :: "v"(preg[0]), "v"(wreg[0]));
#else
      if (more) prefetch(s + 1);           // in flight during the whole multiply of slice s
      compute(s & 1);
      if (more) commit((s + 1) & 1);       // the other buffer: nobody reads it until the barrier below
      __syncthreads();
#endif
    }
  } else {
    for (int s = 0; s < n_slices; ++s) {
      __syncthreads();
      commit(0);
      __syncthreads();
      if (s + 1 < n_slices) prefetch(s + 1);
      compute(0);
    }
  }

  // ---- epilogue: bias + ReLU, 128-byte row stores ---------------------------------------------------------------------------------
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int co = ct * B3_BN + n * 32 + r;
    const float bias = p.bias[co];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int pr = (e & 3) + 8 * (e >> 2) + 4 * h;
        const int oy = oy0 + 2 * (wave * MT + m) + (pr >> 4), ox = ox0 + (pr & 15);
        if (co < p.cout && oy < p.out_h && ox < p.out_w) {
          float v = acc[m][n][e] + bias;
          if (p.relu) v = fmaxf(v, 0.f);
          p.out[(((long long)b * p.out_h + oy) * p.out_w + ox) * p.ld_out + co] = v;
        }
      }
  }
}

}  // namespace

template <int TERMS>
static int launch_b3(const pcp_conv3x3_t *d, const float *in, const void *w_packed, const float *bias, float *out, void *stream) {
  if (!d || !in || !w_packed || !bias || !out) return PCP_ERR_ARG;
  if (d->batch <= 0 || d->in_h <= 0 || d->in_w <= 0 || d->cin <= 0 || (d->cin % B3_CK) || d->cout <= 0 || d->cout_pad < d->cout ||
      (d->cout_pad % B3_BN) || (d->ld_in & 3) || (d->stride != 1 && d->stride != 2))
    return PCP_ERR_ARG;
  if ((((uintptr_t)in) & 15) || (((uintptr_t)w_packed) & 15)) return PCP_ERR_ARG;
  B3Params p;
  p.in = in; p.w = (const __bf16 *)w_packed; p.bias = bias; p.out = out;
  p.batch = d->batch; p.in_h = d->in_h; p.in_w = d->in_w;
  p.out_h = (d->in_h - 1) / d->stride + 1;
  p.out_w = (d->in_w - 1) / d->stride + 1;
  p.cin = d->cin; p.cout = d->cout; p.cout_pad = d->cout_pad; p.ld_in = d->ld_in; p.ld_out = d->ld_out; p.relu = d->relu;
  // stride 1: 32 x 16 pixels per workgroup (double-buffered, weights re-read half as often) when that still gives every CU >= 2
  // workgroups; otherwise 16 x 16 (two workgroups per CU)
  int th = d->stride == 1 ? 16 : 8;
  if (d->stride == 1) {
    const long long big = (long long)p.batch * ((p.out_h + 31) / 32) * ((p.out_w + B3_TW - 1) / B3_TW) * (p.cout_pad / B3_BN);
    if (big >= 512) th = 32;
  }
  p.tiles_y = (p.out_h + th - 1) / th;
  p.tiles_x = (p.out_w + B3_TW - 1) / B3_TW;
  p.n_tiles = p.batch * p.tiles_y * p.tiles_x;
  const long long blocks = (long long)p.n_tiles * (p.cout_pad / B3_BN);
  if (blocks >= (1LL << 31)) return PCP_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (d->stride == 1 && th == 32)
    hipLaunchKernelGGL((k_conv3x3_bf16x3<1, 32, true, TERMS>), dim3((unsigned)blocks), dim3(B3_THREADS), 0, s, p);
  else if (d->stride == 1)
    hipLaunchKernelGGL((k_conv3x3_bf16x3<1, 16, false, TERMS>), dim3((unsigned)blocks), dim3(B3_THREADS), 0, s, p);
  else
    hipLaunchKernelGGL((k_conv3x3_bf16x3<2, 8, false, TERMS>), dim3((unsigned)blocks), dim3(B3_THREADS), 0, s, p);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

extern "C" int pcp_conv3x3_bf16x3(const pcp_conv3x3_t *d, const float *in, const void *w_packed, const float *bias, float *out,
                                  void *stream) {
  return launch_b3<3>(d, in, w_packed, bias, out, stream);
}

extern "C" int pcp_conv3x3_bf16(const pcp_conv3x3_t *d, const float *in, const void *w_packed, const float *bias, float *out,
                                void *stream) {
  return launch_b3<1>(d, in, w_packed, bias, out, stream);
}
