// a6 / a7 / a12 -- 3x3 stride-1 convolution as fused Winograd F(4x4, 3x3), TWO four-wave workgroups per CU (round 3).
//
// Why a second fused kernel: k_wino4f (wino4f.hip) gives one eight-wave workgroup the whole CU, so its serial prologue (3.3 us) and its
// epilogue (8.3 us: accumulator dump, output transform, 131 KB of stores) leave the matrix pipe idle -- 11.6 of the 52.6 us an item of a
// 128 -> 128 layer takes.  Here an item is half as large (16 Winograd tiles = 16 x 16 output pixels x 64 output channels), a
// workgroup is four waves (one per SIMD) and TWO workgroups share a CU: while one is in its prologue or epilogue the other one's MFMAs
// have the pipe, and the hardware -- not a software pipeline -- does the interleaving.
//
//   workgroup (4 waves, 256 VGPRs each) = 16 tiles x 64 output channels x 36 Winograd positions;
//              wave w holds positions 9w .. 9w+8 as 9 x 4 accumulator blocks of v_mfma_f32_16x16x4_f32
//              (rows = 16 output channels, columns = 16 tiles: a lane ends up with FOUR CONSECUTIVE CHANNELS of one tile)
//   per 8-channel slice:
//     raw 18 x 18 x 8 input patch     global -> registers -> channel-planar LDS image (two slices ahead), buffer-descriptor loads
//     input transform V = B^T d B     as in wino4f: item = (tile, channel) on a lane pair, row pass / 9 v_permlane32_swap / column pass
//     A operand = U^T fragments       straight from L2: [cin/8][36][cout_pad/64][64 lanes][8] -- two 16-byte loads per lane and position
//                                     (all four channel blocks, both k steps), requested four positions ahead
//     B operand = V[pos][k][tile]     two ds_read_b32 per position (the V row of a position is exactly the lane order: conflict free)
//     one barrier per slice
//   epilogue, one 32-channel half at a time (the M image of a half is 73.7 KB: two workgroups fit the CU's 160 KB):
//     accumulators -> LDS (16-byte writes, XOR-swizzled channel quads) -> Y = A^T M A + bias (ReLU); unit = (tile, channel quad) on a
//     thread PAIR that splits the output rows (0, 2) / (1, 3) -> 8 16-byte streaming stores per thread and half.
//
// Same arithmetic as k_wino4f up to the summation order inside the MFMAs (k is consumed 4 + 4 instead of 2 + 2 + 2 + 2 per slice).
#include "pcp_common.h"
#include <type_traits>

#ifdef H4_STAMP
__device__ unsigned long long h4_dbg[8192 * 8];               // [workgroup][stamp] (diagnostic build only)
#define H4_STAMP_AT(slot)                                                                        \
  do {                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    if (tid == 0 && blockIdx.x < 8192) {                                                         \
      unsigned long long t_;                                                                     \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                \
      h4_dbg[blockIdx.x * 8 + (slot)] = t_;                                                      \
    }                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                           \
  } while (0)
// per-wave slice stamps of ONE workgroup (blockIdx.x == H4_STAMP_WG): [wave][slice][0: step start, 1..9: after block, 10: LDS drained, 11: after the barrier]
__device__ unsigned long long h4_dbg2[4 * 64 * 12];
#ifndef H4_STAMP_WG
#define H4_STAMP_WG 1500
#endif
#define H4_STAMP2(slice, k)                                                                      \
  do {                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    if (blockIdx.x == H4_STAMP_WG && lane == 0 && (slice) < 64) {                                \
      unsigned long long t_;                                                                     \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                \
      h4_dbg2[(wave * 64 + (slice)) * 12 + (k)] = t_;                                            \
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
constexpr int H4_MS_FLOATS = 36 * 16 * 32;                // 18432 floats = 73.7 KB: M[pos][tile][32 channels, quads swizzled]
constexpr int H4_LDS_FLOATS = H4_MS_FLOATS > H4_MAIN_FLOATS ? H4_MS_FLOATS : H4_MAIN_FLOATS;
constexpr int H4_RAW_ITEMS = H4_RAW_PIX * 2;              // float4 items per slice (648)
constexpr int H4_RAW_PER = (H4_RAW_ITEMS + H4_THREADS - 1) / H4_THREADS;   // 3
constexpr int H4_WBN = 64;
#ifndef H4_URING
#define H4_URING 4                    // positions the U fragments are requested ahead (3: +0.5 % time, 2: +2 %)
#endif
#ifndef H4_VRING
#define H4_VRING 3
#endif
#if !defined(H4_LATE_BARRIER) && !defined(H4_EARLY_BARRIER)
#define H4_EARLY_BARRIER 1             // the slice barrier in front of the last position (-0.5 .. -1 % against the barrier at the end of the step;
#endif                                 // -DH4_LATE_BARRIER keeps that form, which also carries the per-block stamps of the diagnostic build)

struct H4Params {
  const float *in;
  const float *u;       // [cin/8][36 (i*6+j)][cout_pad/64][64 lanes][8]
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

__global__ __launch_bounds__(H4_THREADS, 2) void k_wino4h(H4Params p) {
  __shared__ __attribute__((aligned(16))) float lds[H4_LDS_FLOATS];
  float *rawb = lds;                            // [2][H4_RAW_FLOATS]
  float *vb = lds + 2 * H4_RAW_FLOATS;          // [2][H4_V_FLOATS]
  float *ms = lds;                              // epilogue (aliases everything; used after the last barrier)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5;

  H4_STAMP_AT(0);
#ifdef H4_STAMP
  if (tid == 0 && blockIdx.x < 8192)
    h4_dbg[blockIdx.x * 8 + 7] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                                 ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32);       // HW_ID, XCC_ID
#endif
#if defined(H4_PRIO_SLOT)
  // the two workgroups of a CU sit in wave slots 0 and 1 of every SIMD (HW_ID[3:0]): the odd slot gets the higher issue priority
  if (__builtin_amdgcn_s_getreg((3 << 11) | 4) & 1) __builtin_amdgcn_s_setprio(2);
#endif
  const int lid = xcd_remap_h4(blockIdx.x, gridDim.x);
#ifdef H4_N_FAST
  const int n_blocks = p.cout_pad / H4_WBN;     // N tile as the fast index: the workgroups sharing a raw patch run together on one XCD
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
  const int n0 = nt * H4_WBN;

  // ---- raw patch staging (as in wino4f: clamped / out-of-range buffer offsets, static load count) ---------------------------------------
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in), 0, p.in_bytes, 0x00020000);
  unsigned roff[H4_RAW_PER];
  int rdst[H4_RAW_PER];
#pragma unroll
  for (int i = 0; i < H4_RAW_PER; i++) {
    int idx = tid + i * H4_THREADS;
    if (idx >= H4_RAW_ITEMS) idx -= H4_RAW_ITEMS;          // surplus threads repeat an item
    const int q = idx & 1, pix = idx >> 1;
    const int py = pix / H4_RAW_W, px = pix % H4_RAW_W;
    const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
    rdst[i] = (4 * q) * H4_PLANE + py * H4_RP + px;
    roff[i] = 0x80000000u;                                 // out of range -> 0
    if (iy >= 0 && iy < p.h && ix >= 0 && ix < p.w) roff[i] = (unsigned)((((long long)(b * p.h + iy) * p.w + ix) * p.ld_in + q * 4) * 4);
  }
  f32x4 rreg[H4_RAW_PER];
  auto raw_load = [&](int slice) {
#ifdef H4_DIAG_NO_RLOAD
    if (slice > 2) return;                                   // timing-only build: no raw loads in the main loop (the LDS stores of stale registers stay)
#endif
    const int soff = slice * (H4_CK * 4);
#pragma unroll
    for (int i = 0; i < H4_RAW_PER; i++)
      rreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)roff[i], soff, 0));
  };
  auto raw_store = [&](int buf) {
    float *dst = rawb + buf * H4_RAW_FLOATS;
#pragma unroll
    for (int i = 0; i < H4_RAW_PER; i++) {
      const f32x4 v = rreg[i];
      dst[rdst[i]] = v.x;
      dst[rdst[i] + H4_PLANE] = v.y;
      dst[rdst[i] + 2 * H4_PLANE] = v.z;
      dst[rdst[i] + 3 * H4_PLANE] = v.w;
    }
  };

  // ---- input transform: item = (tile, channel) on the lane pair (l, l + 32); wave w owns channels 2w, 2w + 1 of all 16 tiles ------------
  const int t_li = lane & 31;
  const int t_tile = t_li & 15, t_ch = 2 * wave + (t_li >> 4);
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

  // ---- U fragments from global / L2: per position two f32x4 per lane = {nb0k0, nb0k1, nb1k0, nb1k1}, {nb2k0, nb2k1, nb3k0, nb3k1} ------------
  const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.u), 0, p.u_bytes, 0x00020000);
  const int u_lane = lane * 32;                                                       // bytes, per lane
  const int u_pos = (p.cout_pad / H4_WBN) * (64 * 8 * 4);                             // bytes between positions
  const int u_slice = 36 * u_pos;
  const int u_base = (9 * wave) * u_pos + nt * (64 * 8 * 4);                          // wave-uniform
  const int n_slices = p.cin / H4_CK;
  const int last = n_slices - 1;
  f32x4 uq[9][2];
  auto u_load = [&](int slice, int pi) {
#ifdef H4_DIAG_NO_ULOAD
    if (slice > 0) return;                                   // timing-only build: the first slice's fragments stay in registers
#endif
    const int so = u_base + min(slice, last) * u_slice + pi * u_pos;
    uq[pi][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, u_lane, so, 0));
    uq[pi][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, u_lane + 16, so, 0));
  };

  f32x4 acc[9][4];
#pragma unroll
  for (int i = 0; i < 9; i++)
#pragma unroll
    for (int nb = 0; nb < 4; nb++) acc[i][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int v_off = (9 * wave) * H4_VP + lane;
#ifdef H4_EARLY_BARRIER
  // loop-carried operands: the first H4_VRING V fragments of the NEXT slice and the raw rows of the transform after it are requested right
  // behind the barrier, which sits in front of the slice's last position -- its eight MFMAs cover the LDS latency the next slice used to
  // start with
  float vq[H4_VRING][2];
  f32x4 lo[3];
  float2 hi[3];
#endif

  // One pipeline step = nine fenced blocks (wino4f's round-3 schedule): block pi = position pi's eight MFMAs + the V read H4_VRING positions
  // ahead + the U fragment H4_URING positions ahead (wrapping into the next slice) + one ninth of the slice's other work.
  auto step_blocks = [&](int s, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    const int cur = s & 1, nxt = cur ^ 1;
    const float *vsrc = vb + cur * H4_V_FLOATS + v_off;
    const float *tsrc = rawb + nxt * H4_RAW_FLOATS + t_src;
    float *tdst = vb + nxt * H4_V_FLOATS + t_dst;
    float *rdstb = rawb + cur * H4_RAW_FLOATS;
#ifndef H4_EARLY_BARRIER
    float vq[H4_VRING][2];
#pragma unroll
    for (int i = 0; i < H4_VRING; i++) {
      vq[i][0] = vsrc[i * H4_VP];
      vq[i][1] = vsrc[i * H4_VP + 64];
    }
#endif
    auto mm = [&](int pi) {
      const float v0 = vq[pi % H4_VRING][0], v1 = vq[pi % H4_VRING][1];
      const f32x4 ua = uq[pi][0], ub = uq[pi][1];
      acc[pi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua.x, v0, acc[pi][0], 0, 0, 0);
      acc[pi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua.z, v0, acc[pi][1], 0, 0, 0);
      acc[pi][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ub.x, v0, acc[pi][2], 0, 0, 0);
      acc[pi][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(ub.z, v0, acc[pi][3], 0, 0, 0);
      acc[pi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua.y, v1, acc[pi][0], 0, 0, 0);
      acc[pi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua.w, v1, acc[pi][1], 0, 0, 0);
      acc[pi][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ub.y, v1, acc[pi][2], 0, 0, 0);
      acc[pi][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(ub.w, v1, acc[pi][3], 0, 0, 0);
#ifndef H4_DIAG_NO_VREAD
      if (pi + H4_VRING < 9)
#else
      if (false)
#endif
      {
        vq[pi % H4_VRING][0] = vsrc[(pi + H4_VRING) * H4_VP];
        vq[pi % H4_VRING][1] = vsrc[(pi + H4_VRING) * H4_VP + 64];
      }
      if (pi + H4_URING < 9) u_load(s, pi + H4_URING);
      else if (!LAST) u_load(s + 1, pi + H4_URING - 9);
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
#ifdef H4_DIAG_NO_RAW
      if (blk == 1) h4_bt6(lo[0].x, lo[0].y, lo[0].z, lo[0].w, hi[0].x, hi[0].y, wr[0]);      // timing-only build: no raw staging at all
      if (blk == 2) h4_bt6(lo[1].x, lo[1].y, lo[1].z, lo[1].w, hi[1].x, hi[1].y, wr[1]);
      if (blk == 3) h4_bt6(lo[2].x, lo[2].y, lo[2].z, lo[2].w, hi[2].x, hi[2].y, wr[2]);
#elif defined(H4_RAW_EARLY)
      if (blk == 0) { rstore(0); rstore(1); rstore(2); raw_load(min(s + 3, last)); }       // three more blocks of flight time for the raw patch
      if (blk == 1) h4_bt6(lo[0].x, lo[0].y, lo[0].z, lo[0].w, hi[0].x, hi[0].y, wr[0]);
      if (blk == 2) h4_bt6(lo[1].x, lo[1].y, lo[1].z, lo[1].w, hi[1].x, hi[1].y, wr[1]);
      if (blk == 3) h4_bt6(lo[2].x, lo[2].y, lo[2].z, lo[2].w, hi[2].x, hi[2].y, wr[2]);
#else
      if (blk == 0) rstore(0);
      if (blk == 1) { h4_bt6(lo[0].x, lo[0].y, lo[0].z, lo[0].w, hi[0].x, hi[0].y, wr[0]); rstore(1); }
      if (blk == 2) { h4_bt6(lo[1].x, lo[1].y, lo[1].z, lo[1].w, hi[1].x, hi[1].y, wr[1]); rstore(2); }
      if (blk == 3) { h4_bt6(lo[2].x, lo[2].y, lo[2].z, lo[2].w, hi[2].x, hi[2].y, wr[2]); raw_load(min(s + 3, last)); }
#endif
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
          vq[i][0] = vn[i * H4_VP];
          vq[i][1] = vn[i * H4_VP + 64];
        }
#pragma unroll
      for (int rr = 0; rr < 3; rr++) {
        lo[rr] = *reinterpret_cast<const f32x4 *>(tn + rr * H4_RP);
        hi[rr] = *reinterpret_cast<const float2 *>(tn + rr * H4_RP + 4);
      }
      fence();
      mm(8);
      fence();
      vq[8 % H4_VRING][0] = vn[(8 % H4_VRING) * H4_VP];
      vq[8 % H4_VRING][1] = vn[(8 % H4_VRING) * H4_VP + 64];
      fence();
    } else {
      mm(8);
      __syncthreads();
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
    f32x4 r0[H4_RAW_PER];
    raw_load(0);
#pragma unroll
    for (int i = 0; i < H4_RAW_PER; i++) r0[i] = rreg[i];
    raw_load(min(1, last));
#pragma unroll
    for (int pi = 0; pi < H4_URING; pi++) u_load(0, pi);
    float *dst = rawb;
#pragma unroll
    for (int i = 0; i < H4_RAW_PER; i++) {
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
  transform(0, 0);
  __syncthreads();
#ifdef H4_EARLY_BARRIER
#pragma unroll
  for (int i = 0; i < H4_VRING; i++) {
    vq[i][0] = vb[v_off + i * H4_VP];
    vq[i][1] = vb[v_off + i * H4_VP + 64];
  }
#pragma unroll
  for (int rr = 0; rr < 3; rr++) {                       // raw(1) for the transform that runs beside slice 0 (a dead read when cin = 8)
    lo[rr] = *reinterpret_cast<const f32x4 *>(rawb + H4_RAW_FLOATS + t_src + rr * H4_RP);
    hi[rr] = *reinterpret_cast<const float2 *>(rawb + H4_RAW_FLOATS + t_src + rr * H4_RP + 4);
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

  // ---- epilogue ------------------------------------------------------------------------------------------------------------------------------
  // dump of a 32-channel half: lane (tile = l & 15, kq = l >> 4) holds channels 16 nb + 4 kq .. + 3 of its tile; quad index inside the half =
  // 4 (nb & 1) + kq, stored at quad ^ (tile >> 1) (16 lanes of a ds_write_b128 phase = 16 tiles: 2 x 8 distinct bank quads)
  const int d_tile = lane & 15, d_kq = lane >> 4;
  auto dump = [&](int half) {
#pragma unroll
    for (int pi = 0; pi < 9; pi++) {
      const int pos = 9 * wave + pi;
#pragma unroll
      for (int nbl = 0; nbl < 2; nbl++) {
        const int quad = (4 * nbl + d_kq) ^ (d_tile >> 1);
        *reinterpret_cast<f32x4 *>(ms + (pos * 16 + d_tile) * 32 + 4 * quad) = acc[pi][2 * half + nbl];
      }
    }
  };
  // unit = (tile e_tt, channel quad e_q) on a thread pair: e_par = 0 -> output rows 0 and 2, e_par = 1 -> rows 1 and 3
  const int e_q = tid & 7, e_par = (tid >> 3) & 1, e_tt = tid >> 4;
  const float c_sg = e_par ? -1.f : 1.f, c_m0 = e_par ? 0.f : 1.f, c_a = e_par ? 2.f : 1.f, c_b = e_par ? 8.f : 4.f, c_m5 = e_par ? 1.f : 0.f;
  f32x4 ua[6], ub[6];                                  // ua[j] = row (0 | 1) of A^T M, ub[j] = row (2 | 3)
  auto finish_read = [&]() {
    const float *src = ms + e_tt * 32 + 4 * (e_q ^ (e_tt >> 1));
#pragma unroll
    for (int j = 0; j < 6; j++) {
      f32x4 m[6];
#pragma unroll
      for (int i = 0; i < 6; i++) m[i] = *reinterpret_cast<const f32x4 *>(src + (i * 6 + j) * (16 * 32));
      const f32x4 t12 = m[1] + c_sg * m[2], t34 = m[3] + c_sg * m[4];
      ua[j] = c_m0 * m[0] + t12 + c_a * t34;          // row 0: m0 + s12 + s34        row 1: d12 + 2 d34
      ub[j] = t12 + c_b * t34 + c_m5 * m[5];          // row 2: s12 + 4 s34           row 3: d12 + 8 d34 + m5
      asm volatile("" : "+v"(ua[j]), "+v"(ub[j]));    // one column at a time (keeps the 36 reads from being hoisted into 144 registers)
    }
  };
  auto finish_store = [&](int half) {
    const int n = n0 + half * 32 + 4 * e_q;
    if (n < p.cout) {
      const f32x4 bias = *reinterpret_cast<const f32x4 *>(p.bias + n);
      const int py = oy0 + (e_tt >> 2) * 4 + e_par, px = ox0 + (e_tt & 3) * 4;
#pragma unroll
      for (int a = 0; a < 2; a++) {
        f32x4 y[4];
        if (a == 0) h4_at6v(ua[0], ua[1], ua[2], ua[3], ua[4], ua[5], y);
        else h4_at6v(ub[0], ub[1], ub[2], ub[3], ub[4], ub[5], y);
        if (py + 2 * a < p.h) {
          float *o = p.out + ((long long)(b * p.h + py + 2 * a) * p.w + px) * p.ld_out + n;
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
  };
  H4_STAMP_AT(2);
  dump(0);
  __syncthreads();
  finish_read();
  __syncthreads();
  H4_STAMP_AT(3);
  dump(1);                                             // the LDS writes of half 1 drain under half 0's second pass and stores
  finish_store(0);
  H4_STAMP_AT(4);
  __syncthreads();
  finish_read();
  H4_STAMP_AT(5);
  finish_store(1);
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
  const long long u_bytes = (long long)(d->cin / H4_CK) * 36 * d->cout_pad * H4_CK * 4;
  if (in_bytes > 0x7fffffffLL || u_bytes > 0x7fffffffLL) return PCP_ERR_UNSUPPORTED;
  p->in_bytes = (unsigned)in_bytes;
  p->u_bytes = (unsigned)u_bytes;
  return PCP_OK;
}

}  // namespace

extern "C" int pcp_conv3x3_winograd4h(const pcp_conv3x3_t *d, const float *in, const float *u_packed, const float *bias, float *out,
                                      void *stream_) {
  if (!d || !in || !u_packed || !bias || !out) return PCP_ERR_ARG;
  H4Params p;
  int rc = h4_geom(d, &p);
  if (rc != PCP_OK) return rc;
  if ((((uintptr_t)in) & 15) || (((uintptr_t)u_packed) & 15) || (((uintptr_t)out) & 15) || (((uintptr_t)bias) & 15)) return PCP_ERR_ARG;
  p.in = in; p.u = u_packed; p.bias = bias; p.out = out;
  const long long blocks = (long long)p.n_spatial * (d->cout_pad / H4_WBN);
  if (blocks <= 0 || blocks > 0x7fffffffLL) return PCP_ERR_ARG;
  hipLaunchKernelGGL(k_wino4h, dim3((unsigned)blocks), dim3(H4_THREADS), 0, (hipStream_t)stream_, p);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

#ifdef H4_STAMP
extern "C" int pcp_debug_read_h4(void *dst, size_t bytes) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(h4_dbg), bytes) == hipSuccess ? 0 : 3;
}
extern "C" int pcp_debug_read_h4_slices(void *dst, size_t bytes) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(h4_dbg2), bytes) == hipSuccess ? 0 : 3;
}
#endif

extern "C" int pcp_conv3x3_winograd4h_plan(const pcp_conv3x3_t *d, double *executed_flops) {
  H4Params p;
  int rc = h4_geom(d, &p);
  if (rc != PCP_OK) return rc;
  // every workgroup multiplies [16 tiles x cin] x [cin x 64] at each of the 36 Winograd positions (padding tiles / channels included)
  if (executed_flops) *executed_flops = (double)p.n_spatial * (d->cout_pad / H4_WBN) * 2.0 * 36.0 * 16.0 * d->cin * H4_WBN;
  return PCP_OK;
}
