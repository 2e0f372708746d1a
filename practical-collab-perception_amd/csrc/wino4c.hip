// a6 / a7 / a12 -- 3x3 stride-1 convolution as fused Winograd F(4x4, 3x3), two four-wave workgroups per CU, waves split over OUTPUT CHANNELS
// (round 4).
//
// k_wino4h (wino4h.hip) splits the 36 Winograd positions over its four waves, so the output transform Y = A^T M A needs values from all four
// waves: accumulators -> LDS (a 73.7 KB image per 32-channel half) -> two barriers per half -> transform -> stores, 11 % of an item's time
// (19 % on the 64 -> 64 layers) and the reason the kernel needs the LDS twice over.  Here wave w owns the 16 output channels 16w .. 16w + 15
// of the item at ALL 36 positions (36 accumulator blocks of v_mfma_f32_16x16x4_f32, rows = channels, columns = the 16 tiles): a lane ends
// up with the 36 position values of FOUR CONSECUTIVE CHANNELS of ONE tile in its own registers, so the output transform, bias and ReLU run
// in registers and leave as sixteen 16-byte stores per lane -- no LDS image, no barrier, and a wave that is done with its multiplies does
// not wait for the others.
//
//   item / workgroup / slices / input transform / raw staging: exactly k_wino4h (16 x 16 output pixels x 64 output channels, 8-channel
//              slices, V[pos][k][tile] in LDS, one barrier per slice in front of the slice's last block)
//   A operand = U^T fragments from L2: [cin/8][cout_pad/16][18 position pairs][64 lanes][4 = (position parity, k step)] -- a wave's slice is
//              one 18-KB run, one 16-byte load per lane and position PAIR (pack.repack_winograd4f_to_4c)
//   B operand = V[pos][k][tile]: one ds_read2st64_b32 per position (all 36 positions per wave: four times k_wino4h's LDS reads, still
//              under a quarter of the LDS bandwidth)
//   a slice = nine fenced blocks of four positions (eight MFMAs: k step 0 of the four, then k step 1 -- dependent MFMAs are four apart)
//   LDS: 69 KB per workgroup (raw x2, V x2).
//
// Same arithmetic as k_wino4c (same products, same k order); only the output transform's summation runs per lane instead of per thread pair
// -- identical operations in identical order, so the outputs are bit-identical to k_wino4c's (tests/test_gpu_ops.py).
#include "pcp_common.h"
#include <cstdlib>
#include <type_traits>

#ifdef H4_STAMP
__device__ unsigned long long c4_dbg[8192 * 8];               // [workgroup][stamp] (diagnostic build only)
#define H4_STAMP_AT(slot)                                                                        \
  do {                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    if (tid == 0 && blockIdx.x < 8192) {                                                         \
      unsigned long long t_;                                                                     \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                \
      c4_dbg[blockIdx.x * 8 + (slot)] = t_;                                                      \
    }                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                           \
  } while (0)
// per-wave slice stamps of ONE workgroup (blockIdx.x == H4_STAMP_WG): [wave][slice][0: step start, 1..9: after block, 10: LDS drained, 11: after the barrier]
__device__ unsigned long long c4_dbg2[4 * 64 * 12];
#ifndef H4_STAMP_WG
#define H4_STAMP_WG 1500
#endif
#define H4_STAMP2(slice, k)                                                                      \
  do {                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    if (blockIdx.x == H4_STAMP_WG && lane == 0 && (slice) < 64) {                                \
      unsigned long long t_;                                                                     \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                \
      c4_dbg2[(wave * 64 + (slice)) * 12 + (k)] = t_;                                            \
    }                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                           \
  } while (0)
#else
#define H4_STAMP_AT(slot)
#define H4_STAMP2(slice, k)
#endif

namespace {

constexpr int H4_THREADS = 256;
constexpr int H4_CK = 8;                                  // input channels per slice
constexpr int H4_RP = 20;                                 // raw plane row pitch (floats)
constexpr int H4_RAW_H = 18, H4_RAW_W = 18;               // 16 x 16 output pixels + halo
constexpr int H4_RAW_PIX = H4_RAW_H * H4_RAW_W;
constexpr int H4_PLANE = H4_RAW_H * H4_RP;                // 360 = 40 (mod 64)
constexpr int H4_RAW_FLOATS = H4_CK * H4_PLANE;           // 2880
constexpr int H4_VP = 160;                                // V position pitch: [8 k][16 tiles] + 32 (3 * VP = 32 mod 64: the two lane halves of
                                                          // the column pass store to disjoint banks)
constexpr int H4_V_FLOATS = 36 * H4_VP;                   // 5760
constexpr int H4_MAIN_FLOATS = 2 * H4_RAW_FLOATS + 2 * H4_V_FLOATS;      // 17280
constexpr int H4_LDS_FLOATS = H4_MAIN_FLOATS;              // 69 KB: no accumulator image (the output transform runs in registers)
constexpr int H4_RAW_ITEMS = H4_RAW_PIX * 2;              // float4 items per slice (648)
constexpr int H4_RAW_PER = (H4_RAW_ITEMS + H4_THREADS - 1) / H4_THREADS;   // 3
constexpr int H4_WBN = 64;
#ifndef H4_URING
#define H4_URING 4                    // positions the U fragments are requested ahead (3: +0.5 % time, 2: +2 %)
#endif
#ifndef H4_VRING
#define H4_VRING 2                    // BLOCKS (of four positions) the V fragments are read ahead: 3 spills seven registers into the loop (-7 % .. +37 % time)
#endif
#if !defined(H4_LATE_BARRIER) && !defined(H4_EARLY_BARRIER)
#define H4_EARLY_BARRIER 1             // the slice barrier in front of the last position (-0.5 .. -1 % against the barrier at the end of the step;
#endif                                 // -DH4_LATE_BARRIER keeps that form, which also carries the per-block stamps of the diagnostic build)

struct H4Params {
  const float *in;
  const float *u;       // [cin/8][cout_pad/16][18 position pairs][64 lanes][4]
  const float *bias;
  float *out;
  int batch, h, w;
  int cin, cout, cout_pad;
  int ld_in, ld_out;
  int relu;
  int tiles_x, tiles_y, n_spatial;
  unsigned in_bytes, u_bytes;
};

__device__ __forceinline__ int xcd_remap_h4(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// B^T x for the 6-point transform (points 0, +-1, +-2, inf).  Contraction is spelt out (no compiler-chosen fma grouping): k_wino4h and both
// forms of k_wino4c run exactly these operations, so their outputs agree bit for bit.
__device__ __forceinline__ void h4_bt6(const float d0, const float d1, const float d2, const float d3, const float d4, const float d5,
                                       float (&t)[6]) {
#pragma clang fp contract(off)
  const float p = __builtin_fmaf(-4.f, d2, d4), q = __builtin_fmaf(-4.f, d1, d3);
  const float r = d4 - d2, s = 2.f * (d3 - d1);
  t[0] = __builtin_fmaf(4.f, d0, __builtin_fmaf(-5.f, d2, d4));
  t[1] = p + q;
  t[2] = p - q;
  t[3] = r + s;
  t[4] = r - s;
  t[5] = __builtin_fmaf(4.f, d1, __builtin_fmaf(-5.f, d3, d5));
}

// A^T m for float4 lanes: 6 -> 4 (same rule: explicit fma)
__device__ __forceinline__ void h4_at6v(const f32x4 m0, const f32x4 m1, const f32x4 m2, const f32x4 m3, const f32x4 m4, const f32x4 m5,
                                        f32x4 (&y)[4]) {
#pragma clang fp contract(off)
  const f32x4 s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
  const f32x4 c2 = f32x4{2.f, 2.f, 2.f, 2.f}, c4 = f32x4{4.f, 4.f, 4.f, 4.f}, c8 = f32x4{8.f, 8.f, 8.f, 8.f};
  y[0] = (m0 + s12) + s34;
  y[1] = __builtin_elementwise_fma(c2, d34, d12);
  y[2] = __builtin_elementwise_fma(c4, s34, s12);
  y[3] = __builtin_elementwise_fma(c8, d34, d12) + m5;
}

// NW = waves per workgroup = 16-channel blocks per item.  NW = 4 (the default): 64-channel items, two workgroups per CU.  NW = 8 (opt-in,
// see the launcher): 128-channel items, ONE eight-wave workgroup per CU -- the eight waves share one V image, so the input transform and the
// raw staging are done once per 128 output channels instead of once per 64 (waves 0-3 transform, all eight stage the raw patch and multiply).
// On this chip every non-MFMA instruction of a SIMD's waves costs matrix-pipe time (profiles/experiments/r04_wino4s); halving them per
// product buys 3 - 4 % per layer, which the exposed prologue of a single workgroup per CU gives back inside the pipelined step.
template <int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void k_wino4c(H4Params p) {
  constexpr int THREADS = NW * 64;
  constexpr int RAW_PER = (H4_RAW_ITEMS + THREADS - 1) / THREADS;            // 3 | 2
  constexpr int WBN = NW * 16;                                              // output channels per item
  constexpr int URING = NW == 4 ? H4_URING : 2;     // blocks the U fragments are requested ahead (the eight-wave form spills at 3)
  __shared__ __attribute__((aligned(16))) float lds[H4_LDS_FLOATS];
  float *rawb = lds;                            // [2][H4_RAW_FLOATS]
  float *vb = lds + 2 * H4_RAW_FLOATS;          // [2][H4_V_FLOATS]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5;

  H4_STAMP_AT(0);
#ifdef H4_STAMP
  if (tid == 0 && blockIdx.x < 8192)
    c4_dbg[blockIdx.x * 8 + 7] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                                 ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32);       // HW_ID, XCC_ID
#endif
#if defined(H4_PRIO_SLOT)
  // the two workgroups of a CU sit in wave slots 0 and 1 of every SIMD (HW_ID[3:0]): the odd slot gets the higher issue priority
  if (__builtin_amdgcn_s_getreg((3 << 11) | 4) & 1) __builtin_amdgcn_s_setprio(2);
#endif
  const int lid = xcd_remap_h4(blockIdx.x, gridDim.x);
#ifdef H4_N_FAST
  const int n_blocks = p.cout_pad / WBN;     // N tile as the fast index: the workgroups sharing a raw patch run together on one XCD
  const int nt = lid % n_blocks;
  int sp = lid / n_blocks;
#else
  const int nt = lid / p.n_spatial;             // N tile is the slow index (weights stay in the XCD's L2)
  int sp = lid % p.n_spatial;
#endif
  const int tile_x = sp % p.tiles_x;
  sp /= p.tiles_x;
  const int tile_y = sp % p.tiles_y;
  const int b = sp / p.tiles_y;
  const int oy0 = tile_y * 16, ox0 = tile_x * 16;
  const int n0 = nt * WBN;

  // ---- raw patch staging (as in wino4f: clamped / out-of-range buffer offsets, static load count) ---------------------------------------
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in), 0, p.in_bytes, 0x00020000);
  unsigned roff[RAW_PER];
  int rdst[RAW_PER];
#pragma unroll
  for (int i = 0; i < RAW_PER; i++) {
    int idx = tid + i * THREADS;
    if (idx >= H4_RAW_ITEMS) idx -= H4_RAW_ITEMS;          // surplus threads repeat an item
    const int q = idx & 1, pix = idx >> 1;
    const int py = pix / H4_RAW_W, px = pix % H4_RAW_W;
    const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
    rdst[i] = (4 * q) * H4_PLANE + py * H4_RP + px;
    roff[i] = 0x80000000u;                                 // out of range -> 0
    if (iy >= 0 && iy < p.h && ix >= 0 && ix < p.w) roff[i] = (unsigned)((((long long)(b * p.h + iy) * p.w + ix) * p.ld_in + q * 4) * 4);
  }
  f32x4 rreg[RAW_PER];
  auto raw_load = [&](int slice) {
#ifdef H4_DIAG_NO_RLOAD
    if (slice > 2) return;                                   // timing-only build: no raw loads in the main loop (the LDS stores of stale registers stay)
#endif
    const int soff = slice * (H4_CK * 4);
#pragma unroll
    for (int i = 0; i < RAW_PER; i++)
      rreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)roff[i], soff, 0));
  };
  auto raw_store = [&](int buf) {
    float *dst = rawb + buf * H4_RAW_FLOATS;
#pragma unroll
    for (int i = 0; i < RAW_PER; i++) {
      const f32x4 v = rreg[i];
      dst[rdst[i]] = v.x;
      dst[rdst[i] + H4_PLANE] = v.y;
      dst[rdst[i] + 2 * H4_PLANE] = v.z;
      dst[rdst[i] + 3 * H4_PLANE] = v.w;
    }
  };

  // ---- input transform: item = (tile, channel) on the lane pair (l, l + 32); wave w owns channels 2w, 2w + 1 of all 16 tiles ------------
  const int t_li = lane & 31;
  const bool tf = NW == 4 || wave < 4;                    // this wave runs the input transform (wave-uniform)
  const int t_tile = t_li & 15, t_ch = 2 * (wave & 3) + (t_li >> 4);
  const int t_src = t_ch * H4_PLANE + (4 * (t_tile >> 2) + 3 * h) * H4_RP + 4 * (t_tile & 3);
  const int t_dst = t_ch * 16 + t_tile + (3 * h) * H4_VP;
  auto transform = [&](int rbuf, int vbuf) {
    const float *src = rawb + rbuf * H4_RAW_FLOATS + t_src;
    float *dst = vb + vbuf * H4_V_FLOATS + t_dst;
    float wr[3][6];
#pragma unroll
    for (int rr = 0; rr < 3; rr++) {
      const f32x4 lo = *reinterpret_cast<const f32x4 *>(src + rr * H4_RP);
      const float2 hi = *reinterpret_cast<const float2 *>(src + rr * H4_RP + 4);
      h4_bt6(lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, wr[rr]);
    }
    float top[3][3], bot[3][3];
#pragma unroll
    for (int rr = 0; rr < 3; rr++)
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(wr[rr][c]), __float_as_uint(wr[rr][3 + c]), false, false);
        top[rr][c] = __uint_as_float(sw[0]);
        bot[rr][c] = __uint_as_float(sw[1]);
      }
#pragma unroll
    for (int c = 0; c < 3; c++) {
      float o[6];
      h4_bt6(top[0][c], top[1][c], top[2][c], bot[0][c], bot[1][c], bot[2][c], o);
#pragma unroll
      for (int i = 0; i < 6; i++) dst[(i * 6 + c) * H4_VP] = o[i];
    }
  };

  // ---- U fragments from global / L2: per position PAIR one f32x4 per lane = {pos 2q k0, pos 2q k1, pos 2q+1 k0, pos 2q+1 k1} of this
  // wave's 16 output channels --------------------------------------------------------------------------------------------------------------
  const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.u), 0, p.u_bytes, 0x00020000);
  const int u_lane = lane * 16;                                                       // bytes, per lane
  const int u_slice = (p.cout_pad / 16) * (18 * 64 * 4 * 4);                          // bytes between slices
  const int u_base = (nt * NW + wave) * (18 * 64 * 4 * 4);                             // wave-uniform: this wave's 16-channel block
  const int n_slices = p.cin / H4_CK;
  const int last = n_slices - 1;
  f32x4 uq[18];
  auto u_load = [&](int slice, int q) {
#ifdef H4_DIAG_NO_ULOAD
    if (slice > 0) return;                                   // timing-only build: the first slice's fragments stay in registers
#endif
    const int so = u_base + min(slice, last) * u_slice + q * (64 * 4 * 4);
    uq[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, u_lane, so, 0));
  };

  f32x4 acc[36];
#pragma unroll
  for (int i = 0; i < 36; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int v_off = lane;
#ifdef H4_EARLY_BARRIER
  // loop-carried operands: the first H4_VRING V fragments of the NEXT slice and the raw rows of the transform after it are requested right
  // behind the barrier, which sits in front of the slice's last position -- its eight MFMAs cover the LDS latency the next slice used to
  // start with
  float vq[H4_VRING][4][2];
  f32x4 lo[3];
  float2 hi[3];
#endif

  // One pipeline step = nine fenced blocks (wino4f's round-3 schedule): block pi = position pi's eight MFMAs + the V read H4_VRING positions
  // ahead + the U fragment URING positions ahead (wrapping into the next slice) + one ninth of the slice's other work.
  auto step_blocks = [&](int s, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    const int cur = s & 1, nxt = cur ^ 1;
    const float *vsrc = vb + cur * H4_V_FLOATS + v_off;
    const float *tsrc = rawb + nxt * H4_RAW_FLOATS + t_src;
    float *tdst = vb + nxt * H4_V_FLOATS + t_dst;
    float *rdstb = rawb + cur * H4_RAW_FLOATS;
#ifndef H4_EARLY_BARRIER
    float vq[H4_VRING][4][2];
#pragma unroll
    for (int i = 0; i < H4_VRING; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        vq[i][j][0] = vsrc[(4 * i + j) * H4_VP];
        vq[i][j][1] = vsrc[(4 * i + j) * H4_VP + 64];
      }
#endif
    // block pi = positions 4 pi .. 4 pi + 3: k step 0 of the four, then k step 1 (an accumulator's two MFMAs are four instructions apart)
    auto mm = [&](int pi) {
      const f32x4 ua = uq[2 * pi], ub = uq[2 * pi + 1];
      float (&v)[4][2] = vq[pi % H4_VRING];
      acc[4 * pi + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua.x, v[0][0], acc[4 * pi + 0], 0, 0, 0);
      acc[4 * pi + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua.z, v[1][0], acc[4 * pi + 1], 0, 0, 0);
      acc[4 * pi + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ub.x, v[2][0], acc[4 * pi + 2], 0, 0, 0);
      acc[4 * pi + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(ub.z, v[3][0], acc[4 * pi + 3], 0, 0, 0);
      acc[4 * pi + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua.y, v[0][1], acc[4 * pi + 0], 0, 0, 0);
      acc[4 * pi + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua.w, v[1][1], acc[4 * pi + 1], 0, 0, 0);
      acc[4 * pi + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ub.y, v[2][1], acc[4 * pi + 2], 0, 0, 0);
      acc[4 * pi + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(ub.w, v[3][1], acc[4 * pi + 3], 0, 0, 0);
#ifndef H4_DIAG_NO_VREAD
      if (pi + H4_VRING < 9)
#else
      if (false)
#endif
      {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          v[j][0] = vsrc[(4 * (pi + H4_VRING) + j) * H4_VP];
          v[j][1] = vsrc[(4 * (pi + H4_VRING) + j) * H4_VP + 64];
        }
      }
#pragma unroll
      for (int e = 0; e < 2; e++) {
        const int q = 2 * (pi + URING) + e;
        if (q < 18) u_load(s, q);
        else if (!LAST) u_load(s + 1, q - 18);
      }
    };
    auto rstore = [&](int i) {
      const f32x4 v = rreg[i];
#ifdef H4_DIAG_NO_RSTORE
      if (v.x + v.y + v.z + v.w != 1.2345e30f) return;        // timing-only build: the loads stay (their values are consumed), the LDS stores go
#endif
      rdstb[rdst[i]] = v.x;
      rdstb[rdst[i] + H4_PLANE] = v.y;
      rdstb[rdst[i] + 2 * H4_PLANE] = v.z;
      rdstb[rdst[i] + 3 * H4_PLANE] = v.w;
    };
    const auto fence = [] { __builtin_amdgcn_sched_barrier(0); };
    fence();
#ifndef H4_EARLY_BARRIER
    f32x4 lo[3];
    float2 hi[3];
    if (!LAST) {
#pragma unroll
      for (int rr = 0; rr < 3; rr++) {
        lo[rr] = *reinterpret_cast<const f32x4 *>(tsrc + rr * H4_RP);
        hi[rr] = *reinterpret_cast<const float2 *>(tsrc + rr * H4_RP + 4);
      }
    }
#endif
    float wr[3][6];
    float top[3][3], bot[3][3];
    auto other = [&](int blk) {
      if (blk == 0) rstore(0);
      if (blk == 1) rstore(1);
      if (blk == 2 && RAW_PER > 2) rstore(RAW_PER > 2 ? 2 : 0);
      if (blk == 3) raw_load(min(s + 3, last));
      if (!tf) return;
      if (blk == 1) h4_bt6(lo[0].x, lo[0].y, lo[0].z, lo[0].w, hi[0].x, hi[0].y, wr[0]);
      if (blk == 2) h4_bt6(lo[1].x, lo[1].y, lo[1].z, lo[1].w, hi[1].x, hi[1].y, wr[1]);
      if (blk == 3) h4_bt6(lo[2].x, lo[2].y, lo[2].z, lo[2].w, hi[2].x, hi[2].y, wr[2]);
      if (blk == 4) {
#pragma unroll
        for (int rr = 0; rr < 3; rr++)
#pragma unroll
          for (int c = 0; c < 3; c++) {
#ifdef H4_DIAG_NO_SWAP
            top[rr][c] = wr[rr][c];                 // timing-only build: wrong values, no cross-lane exchange
            bot[rr][c] = wr[rr][3 + c];
#else
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(wr[rr][c]), __float_as_uint(wr[rr][3 + c]), false, false);
            top[rr][c] = __uint_as_float(sw[0]);
            bot[rr][c] = __uint_as_float(sw[1]);
#endif
          }
      }
      if (blk >= 5 && blk <= 7) {
        const int c = blk - 5;
        float o[6];
        h4_bt6(top[0][c], top[1][c], top[2][c], bot[0][c], bot[1][c], bot[2][c], o);
#ifdef H4_DIAG_NO_VSTORE
        if (o[0] + o[1] + o[2] + o[3] + o[4] + o[5] == 1.2345e30f)      // timing-only build: never true, keeps the arithmetic
#endif
#pragma unroll
        for (int i = 0; i < 6; i++) tdst[(i * 6 + c) * H4_VP] = o[i];
      }
    };
#ifdef H4_EARLY_BARRIER
#pragma unroll
    for (int blk = 0; blk < 8; blk++) {
      mm(blk);
      fence();
      if (!LAST) {
        other(blk);
        fence();
      }
    }
    if (!LAST) {
      __syncthreads();                       // V[nxt] and raw[cur] are complete; every read of V[cur] has returned (the ring is 3 deep)
      const float *vn = vb + nxt * H4_V_FLOATS + v_off;
      const float *tn = rawb + cur * H4_RAW_FLOATS + t_src;
#pragma unroll
      for (int i = 0; i < H4_VRING; i++)
        if (i != 8 % H4_VRING) {
#pragma unroll
          for (int j = 0; j < 4; j++) {
            vq[i][j][0] = vn[(4 * i + j) * H4_VP];
            vq[i][j][1] = vn[(4 * i + j) * H4_VP + 64];
          }
        }
      if (tf) {
#pragma unroll
        for (int rr = 0; rr < 3; rr++) {
          lo[rr] = *reinterpret_cast<const f32x4 *>(tn + rr * H4_RP);
          hi[rr] = *reinterpret_cast<const float2 *>(tn + rr * H4_RP + 4);
        }
      }
      fence();
      mm(8);
      fence();
#pragma unroll
      for (int j = 0; j < 4; j++) {
        vq[8 % H4_VRING][j][0] = vn[(4 * (8 % H4_VRING) + j) * H4_VP];
        vq[8 % H4_VRING][j][1] = vn[(4 * (8 % H4_VRING) + j) * H4_VP + 64];
      }
      fence();
    } else {
      mm(8);                                 // no barrier behind the last block: the epilogue touches no LDS
    }
#else
    H4_STAMP2(s, 0);
#pragma unroll
    for (int blk = 0; blk < 9; blk++) {
      mm(blk);
      fence();
#ifndef H4_DIAG_NO_OTHER
      if (!LAST) {
        other(blk);
        fence();
      }
#endif
      H4_STAMP2(s, 1 + blk);
    }
#ifdef H4_STAMP
    __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0): the stamped build separates the LDS drain from the barrier wait
    H4_STAMP2(s, 10);
#endif
    __syncthreads();
    H4_STAMP2(s, 11);
#endif
  };

  // ---- prologue: raw(0), raw(1) -> LDS; V(0); rreg <- raw(2); first U fragments -----------------------------------------------------------
  {
    f32x4 r0[RAW_PER];
    raw_load(0);
#pragma unroll
    for (int i = 0; i < RAW_PER; i++) r0[i] = rreg[i];
    raw_load(min(1, last));
#pragma unroll
    for (int q = 0; q < 2 * URING; q++) u_load(0, q);
    float *dst = rawb;
#pragma unroll
    for (int i = 0; i < RAW_PER; i++) {
      const f32x4 v = r0[i];
      dst[rdst[i]] = v.x;
      dst[rdst[i] + H4_PLANE] = v.y;
      dst[rdst[i] + 2 * H4_PLANE] = v.z;
      dst[rdst[i] + 3 * H4_PLANE] = v.w;
    }
    if (n_slices > 1) raw_store(1);
    raw_load(min(2, last));
  }
  __syncthreads();
  if (tf) transform(0, 0);
  __syncthreads();
#ifdef H4_EARLY_BARRIER
#pragma unroll
  for (int i = 0; i < H4_VRING; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      vq[i][j][0] = vb[v_off + (4 * i + j) * H4_VP];
      vq[i][j][1] = vb[v_off + (4 * i + j) * H4_VP + 64];
    }
  if (tf) {
#pragma unroll
    for (int rr = 0; rr < 3; rr++) {                     // raw(1) for the transform that runs beside slice 0 (a dead read when cin = 8)
      lo[rr] = *reinterpret_cast<const f32x4 *>(rawb + H4_RAW_FLOATS + t_src + rr * H4_RP);
      hi[rr] = *reinterpret_cast<const float2 *>(rawb + H4_RAW_FLOATS + t_src + rr * H4_RP + 4);
    }
  }
#endif

  H4_STAMP_AT(1);
#if defined(H4_PRIO_MAIN)
  __builtin_amdgcn_s_setprio(2);
#endif
  for (int s = 0; s < last; s++) step_blocks(s, std::false_type{});
  step_blocks(last, std::true_type{});        // ends with the barrier after which the LDS belongs to the epilogue
#if defined(H4_PRIO_MAIN)
  __builtin_amdgcn_s_setprio(0);
#elif defined(H4_PRIO_EPI)
  __builtin_amdgcn_s_setprio(2);
#endif

  // ---- epilogue: Y = A^T M A per lane (tile = lane & 15, channel quad = lane >> 4), bias, ReLU, sixteen 16-byte streaming stores ------------
  H4_STAMP_AT(2);
  {
    const int e_tile = lane & 15, e_kq = lane >> 4;
    const int n = n0 + 16 * wave + 4 * e_kq;
    if (n < p.cout) {
      const f32x4 bias = *reinterpret_cast<const f32x4 *>(p.bias + n);
      const int py = oy0 + (e_tile >> 2) * 4, px = ox0 + (e_tile & 3) * 4;
      f32x4 t[4][6];                                   // t[r][j] = row r of A^T M, column j
#pragma unroll
      for (int j = 0; j < 6; j++) {
        f32x4 y[4];
        h4_at6v(acc[j], acc[6 + j], acc[12 + j], acc[18 + j], acc[24 + j], acc[30 + j], y);
#pragma unroll
        for (int r = 0; r < 4; r++) t[r][j] = y[r];
      }
#pragma unroll
      for (int r = 0; r < 4; r++) {
        f32x4 y[4];
        h4_at6v(t[r][0], t[r][1], t[r][2], t[r][3], t[r][4], t[r][5], y);
        if (py + r < p.h) {
          float *o = p.out + ((long long)(b * p.h + py + r) * p.w + px) * p.ld_out + n;
#pragma unroll
          for (int c2 = 0; c2 < 4; c2++)
            if (px + c2 < p.w) {
              f32x4 v = y[c2] + bias;
              if (p.relu) {
                v.x = fmaxf(v.x, 0.f);
                v.y = fmaxf(v.y, 0.f);
                v.z = fmaxf(v.z, 0.f);
                v.w = fmaxf(v.w, 0.f);
              }
              __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(o + (long long)c2 * p.ld_out));
            }
        }
      }
    }
  }
  H4_STAMP_AT(6);
}

int h4_geom(const pcp_conv3x3_t *d, H4Params *p) {
  if (!d || d->stride != 1) return PCP_ERR_UNSUPPORTED;
  if (d->cin <= 0 || d->cin % H4_CK != 0 || d->cout <= 0 || d->cout_pad < d->cout || d->cout_pad % H4_WBN != 0) return PCP_ERR_ARG;
  if (d->ld_in % 4 != 0 || d->ld_out % 4 != 0 || d->cout % 4 != 0 || d->batch <= 0 || d->in_h <= 0 || d->in_w <= 0) return PCP_ERR_ARG;
  p->batch = d->batch; p->h = d->in_h; p->w = d->in_w;
  p->cin = d->cin; p->cout = d->cout; p->cout_pad = d->cout_pad;
  p->ld_in = d->ld_in; p->ld_out = d->ld_out; p->relu = d->relu;
  p->tiles_x = (d->in_w + 15) / 16;
  p->tiles_y = (d->in_h + 15) / 16;
  p->n_spatial = d->batch * p->tiles_x * p->tiles_y;
  const long long in_bytes = (long long)d->batch * d->in_h * d->in_w * d->ld_in * 4;
  const long long u_bytes = (long long)(d->cin / H4_CK) * 36 * d->cout_pad * H4_CK * 4;       // [cin/8][cout_pad/16][18][64][4] floats
  if (in_bytes > 0x7fffffffLL || u_bytes > 0x7fffffffLL) return PCP_ERR_UNSUPPORTED;
  p->in_bytes = (unsigned)in_bytes;
  p->u_bytes = (unsigned)u_bytes;
  return PCP_OK;
}

}  // namespace

extern "C" int pcp_conv3x3_winograd4c(const pcp_conv3x3_t *d, const float *in, const float *u_packed, const float *bias, float *out,
                                      void *stream_) {
  if (!d || !in || !u_packed || !bias || !out) return PCP_ERR_ARG;
  H4Params p;
  int rc = h4_geom(d, &p);
  if (rc != PCP_OK) return rc;
  if ((((uintptr_t)in) & 15) || (((uintptr_t)u_packed) & 15) || (((uintptr_t)out) & 15) || (((uintptr_t)bias) & 15)) return PCP_ERR_ARG;
  p.in = in; p.u = u_packed; p.bias = bias; p.out = out;
  // The 128-channel form (one eight-wave workgroup per CU, the input transform shared by all eight waves) is OPT-IN (PCP_WINO4C_NW=8, on layers
  // with cout_pad % 128 == 0): interleaved per-layer timing has it 3 - 4 % ahead of the 64-channel form on the 128^2 maps (B20 128->128: 308.7 ->
  // 297.6 us) and 15 - 28 % behind on 64^2 maps whose 128-channel items cover the chip 0.5 or 1.25 times; inside bench.py -- two replicas'
  // kernels sharing the chip -- a dispatch rule that picks it only where it wins measured 374.6 against 376.6 frames/s without it, so the
  // default stays the 64-channel form (profiles/r04_wino4c_ab.txt).  Same weights, same bits.
  const bool wide = pcp_option(PCP_OPT_WINO4C_NW, 4) == 8 && d->cout_pad % 128 == 0;      // layers without whole 128-channel blocks keep the 64-channel form
  const long long blocks = (long long)p.n_spatial * (d->cout_pad / (wide ? 128 : 64));
  if (blocks <= 0 || blocks > 0x7fffffffLL) return PCP_ERR_ARG;
  if (wide) hipLaunchKernelGGL(k_wino4c<8>, dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream_, p);
  else hipLaunchKernelGGL(k_wino4c<4>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, p);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

#ifdef H4_STAMP
extern "C" int pcp_debug_read_c4(void *dst, size_t bytes) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(c4_dbg), bytes) == hipSuccess ? 0 : 3;
}
extern "C" int pcp_debug_read_c4_slices(void *dst, size_t bytes) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(c4_dbg2), bytes) == hipSuccess ? 0 : 3;
}
#endif

extern "C" int pcp_conv3x3_winograd4c_plan(const pcp_conv3x3_t *d, double *executed_flops) {
  H4Params p;
  int rc = h4_geom(d, &p);
  if (rc != PCP_OK) return rc;
  // every workgroup multiplies [16 tiles x cin] x [cin x 64] at each of the 36 Winograd positions (padding tiles / channels included)
  if (executed_flops) *executed_flops = (double)p.n_spatial * (d->cout_pad / H4_WBN) * 2.0 * 36.0 * 16.0 * d->cin * H4_WBN;
  return PCP_OK;
}
