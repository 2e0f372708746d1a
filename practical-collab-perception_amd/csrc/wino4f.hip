// a6 / a7 / a12 -- 3x3 stride-1 convolution as FUSED Winograd F(4x4, 3x3) on the fp32 MFMA pipe (gfx950), one launch.
//
// Why: the fused F(2x2,3x3) kernels (wino.hip, wino_ws.hip) sit at 90-96 TFLOP/s of EXECUTED flops on the 64 / 128-channel layers and a
// build of the same launch with everything but the MFMAs removed reaches only 117-123 (the chip holds ~2.0 GHz under fp32 MFMA load):
// there is at most ~25 % left in making that loop tighter.  F(4x4,3x3) executes 36 products per 4x4 output tile instead of 4 x 16:
// 1.78x fewer matrix flops for the same convolution, fp32 arithmetic and accumulation throughout (transform rounding ~1e-5 of the output
// scale, the same as the through-memory F(4x4) path of wino4.hip that all parity tests already run with).
//
// wino4.hip goes through HBM (V and M round trips) and only pays for wide layers.  This kernel keeps everything on chip by giving ONE
// workgroup per CU the CU's whole register file:
//
//   workgroup (8 waves, 2 per SIMD, 256 VGPRs each) = 32 Winograd tiles (4 x 8 tiles = 16 x 32 output pixels) x 64 output channels,
//              all 36 Winograd positions: wave w holds 9 positions x 32 tiles x 32 channels = 9 accumulator tiles (144 VGPRs)
//   per 8-channel slice:
//     raw 18 x 34 x 8 input patch      global -> registers -> channel-planar LDS image (two slices ahead)
//     input transform V = B^T d B      VALU, LDS -> LDS, one slice ahead, double buffered; item = (tile, channel), split over two
//                                      threads by output column triple (78 operations each, all 512 threads busy)
//     36 GEMMs [32 tiles x 8] x [8 x 64] on v_mfma_f32_32x32x2_f32: A fragments = V rows from LDS (one ds_read_b128 feeds four MFMAs,
//                                      k permuted as in wino.hip), B fragments = U = G g G^T straight from L2 into registers
//                                      ([cin/8][36][cout_pad][8]: one coalesced 1-KiB load per position), reloaded for the next slice
//                                      right after their last use
//     one barrier per slice; SIMD partners de-phased (waves 0-3 transform first, waves 4-7 multiply first)
//   epilogue: accumulators -> LDS ([36][32 tiles][32 channels], one cout half at a time: 147 KB) -> Y = A^T M A + bias (ReLU) by all 512
//             threads (unit = tile x channel quad x output row pair) -> 16-byte stores.
#include "pcp_common.h"
#include <type_traits>

#ifdef F4_STAMP
__device__ unsigned long long f4_dbg[8 * 32];                 // [wave][stamp] of workgroup F4_STAMP (diagnostic build only)
#define F4_STAMP_AT(slot)                                                                        \
  do {                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    if (blockIdx.x == F4_STAMP && lane == 0) {                                                   \
      unsigned long long t_;                                                                     \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                \
      f4_dbg[wave * 32 + (slot)] = t_;                                                           \
    }                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                           \
  } while (0)
#else
#define F4_STAMP_AT(slot)
#endif

namespace {

constexpr int F4_THREADS = 512;
constexpr int F4_CK = 8;                                  // input channels per slice
constexpr int F4_RP = 40;                                 // raw plane row pitch (floats): 4 * RP mod 64 = 32
constexpr int F4_RAW_H = 18, F4_RAW_W = 34;               // 16 x 32 output pixels + halo
constexpr int F4_RAW_PIX = F4_RAW_H * F4_RAW_W;
constexpr int F4_PLANE = F4_RAW_H * F4_RP;                // 720 = 16 (mod 64): the 16-lane groups of the transform's ds_read_b128 hit 64 distinct banks
constexpr int F4_RAW_FLOATS = F4_CK * F4_PLANE;           // 5824
constexpr int F4_VLD = 12;                                // padded V row (floats): the 16 lanes of a ds_read_b128 group hit 64 distinct banks
constexpr int F4_V_FLOATS = 36 * 32 * F4_VLD;             // 13824: V[pos][tile][8 channels + 4 pad]
constexpr int F4_MAIN_FLOATS = 2 * F4_RAW_FLOATS + 2 * F4_V_FLOATS;
constexpr int F4_MS_LD = 32;
constexpr int F4_MS_FLOATS = 36 * 32 * F4_MS_LD;          // 36864 floats = 147 KB: M[pos][tile][32 channels]
constexpr int F4_LDS_FLOATS = (F4_MS_FLOATS > F4_MAIN_FLOATS ? F4_MS_FLOATS : F4_MAIN_FLOATS) + 4;   // + the half-1 barrier counter of the epilogue
constexpr int F4_RAW_ITEMS = F4_RAW_PIX * 2;              // float4 items per slice
constexpr int F4_RAW_PER = (F4_RAW_ITEMS + F4_THREADS - 1) / F4_THREADS;
constexpr int F4_WBN = 64;

struct F4Params {
  const float *in;
  const float *u;       // [cin/8][36 (i*6+j)][cout_pad][8]
  const float *bias;
  float *out;
  int batch, h, w;
  int cin, cout, cout_pad;
  int ld_in, ld_out;
  int relu;
  int tiles_x, tiles_y, n_spatial;
  unsigned in_bytes, u_bytes;     // extents for the buffer descriptors (range-checked loads)
};

__device__ __forceinline__ int xcd_remap_f4(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// B^T x for the 6-point transform (Lavin & Gray F(4,3): points 0, +-1, +-2, inf), all six outputs
__device__ __forceinline__ void f4_bt6(const float d0, const float d1, const float d2, const float d3, const float d4, const float d5,
                                       float (&t)[6]) {
  const float p = d4 - 4.f * d2, q = d3 - 4.f * d1;
  const float r = d4 - d2, s = 2.f * (d3 - d1);
  t[0] = 4.f * d0 - 5.f * d2 + d4;
  t[1] = p + q;
  t[2] = p - q;
  t[3] = r + s;
  t[4] = r - s;
  t[5] = 4.f * d1 - 5.f * d3 + d5;
}

// A^T m: 6 -> 4
__device__ __forceinline__ void f4_at6(const float m0, const float m1, const float m2, const float m3, const float m4, const float m5,
                                       float (&y)[4]) {
  const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
  y[0] = m0 + s12 + s34;
  y[1] = d12 + 2.f * d34;
  y[2] = s12 + 4.f * s34;
  y[3] = d12 + 8.f * d34 + m5;
}

// A^T m for float4 lanes: 6 -> 4
__device__ __forceinline__ void f4_at6v(const f32x4 m0, const f32x4 m1, const f32x4 m2, const f32x4 m3, const f32x4 m4, const f32x4 m5,
                                        f32x4 (&y)[4]) {
  const f32x4 s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
  y[0] = m0 + s12 + s34;
  y[1] = d12 + 2.f * d34;
  y[2] = s12 + 4.f * s34;
  y[3] = d12 + 8.f * d34 + m5;
}

__global__ __launch_bounds__(F4_THREADS, 2) void k_wino4f(F4Params p) {
  __shared__ __attribute__((aligned(16))) float lds[F4_LDS_FLOATS];
  float *rawb = lds;                            // [2][F4_RAW_FLOATS]
  float *vb = lds + 2 * F4_RAW_FLOATS;          // [2][F4_V_FLOATS]
  float *ms = lds;                              // epilogue (aliases everything; used after the last barrier)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int pg = wave & 3;                      // position group: positions 9 * pg .. 9 * pg + 8
  const int cb = wave >> 2;                     // 32-channel half of the workgroup's 64 output channels (waves 0-3 / 4-7: one per SIMD)

  const int lid = xcd_remap_f4(blockIdx.x, gridDim.x);
  const int nt = lid / p.n_spatial;             // N tile is the slow index: an XCD works on few N tiles at a time (weights stay in its L2)
  int sp = lid % p.n_spatial;
  const int tile_x = sp % p.tiles_x;
  sp /= p.tiles_x;
  const int tile_y = sp % p.tiles_y;
  const int b = sp / p.tiles_y;
  const int oy0 = tile_y * 16, ox0 = tile_x * 32;
  const int n0 = nt * F4_WBN;

  // ---- raw patch staging: item = (pixel, channel quad).  Unconditional loads with clamped addresses (static VMEM counts); pixels
  //      outside the image are zeroed by a select.  LDS image is channel-planar: [8 channels][18 rows][pitch 40]. ------------------------
  // Loads go through a buffer descriptor: the per-lane byte offset sits in one VGPR, the slice offset in an SGPR (no vector address
  // arithmetic in the loop -- VALU work displaces fp32 MFMAs on gfx950), and pixels outside the image carry an offset past the end of
  // the buffer, for which the range check returns zeros (no select either).  The load count per step is static (counted vmcnt waits).
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in), 0, p.in_bytes, 0x00020000);
  unsigned roff[F4_RAW_PER];
  int rdst[F4_RAW_PER];
#pragma unroll
  for (int i = 0; i < F4_RAW_PER; i++) {
    int idx = tid + i * F4_THREADS;
    if (idx >= F4_RAW_ITEMS) idx -= F4_RAW_ITEMS;          // surplus threads repeat an item (same value to the same address): the staging
    const int q = idx & 1, pix = idx >> 1;                 // code stays free of divergent branches
    const int py = pix / F4_RAW_W, px = pix % F4_RAW_W;
    const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
    rdst[i] = (4 * q) * F4_PLANE + py * F4_RP + px;
    roff[i] = 0x80000000u;                                 // out of range -> 0
    if (iy >= 0 && iy < p.h && ix >= 0 && ix < p.w) roff[i] = (unsigned)((((long long)(b * p.h + iy) * p.w + ix) * p.ld_in + q * 4) * 4);
  }
  f32x4 rreg[F4_RAW_PER];
  auto raw_load = [&](int slice) {
    const int soff = slice * (F4_CK * 4);                  // wave-uniform byte offset of the slice's first channel
#pragma unroll
    for (int i = 0; i < F4_RAW_PER; i++)
      rreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)roff[i], soff, 0));
  };
  auto raw_store = [&](int buf) {
    float *dst = rawb + buf * F4_RAW_FLOATS;
#pragma unroll
    for (int i = 0; i < F4_RAW_PER; i++) {
      const f32x4 v = rreg[i];
      dst[rdst[i]] = v.x;
      dst[rdst[i] + F4_PLANE] = v.y;
      dst[rdst[i] + 2 * F4_PLANE] = v.z;
      dst[rdst[i] + 3 * F4_PLANE] = v.w;
    }
  };

  // ---- input transform V = B^T d B, LDS -> LDS.  Item = (tile, channel), one per lane PAIR (l, l + 32) of a wave, ONE instruction stream
  //      for all lanes (no wave-uniform branches: the step below is a single scheduling region):
  //        row pass:    lane half h transforms raw rows 3h .. 3h+2 of the 6 x 6 patch (full 6-point B^T: 13 operations per row)
  //        exchange:    9 v_permlane32_swap: half 0 hands columns 3..5 of its rows to half 1 and receives columns 0..2 of rows 3..5
  //        column pass: half h transforms columns 3h .. 3h+2 (6-point B^T down the column) -> V[i][3h + c], i = 0..5
  //      87 VALU + 9 swaps, 6 LDS reads (16 + 8 bytes per raw row), 18 LDS stores per lane and slice. ----------------------------------------
  const int t_li = lane & 31;
  const int t_ty = wave >> 1, t_tx = (wave & 1) * 4 + (t_li >> 3), t_ch = t_li & 7;
  const int t_src = t_ch * F4_PLANE + (4 * t_ty + 3 * h) * F4_RP + 4 * t_tx;
  const int t_dst = ((t_ty * 8 + t_tx) * F4_VLD + t_ch) + (3 * h) * (32 * F4_VLD);
  auto transform = [&](int rbuf, int vbuf) {
#ifdef F4_DIAG_NO_XFORM
    return;                                                  // timing-only build
#endif
    const float *src = rawb + rbuf * F4_RAW_FLOATS + t_src;
    float *dst = vb + vbuf * F4_V_FLOATS + t_dst;
    float wr[3][6];
#pragma unroll
    for (int rr = 0; rr < 3; rr++) {
      const f32x4 lo = *reinterpret_cast<const f32x4 *>(src + rr * F4_RP);
      const float2 hi = *reinterpret_cast<const float2 *>(src + rr * F4_RP + 4);
      f4_bt6(lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, wr[rr]);
    }
    float top[3][3], bot[3][3];                               // after the swap: column 3h + c, rows rr (top) and 3 + rr (bot)
#pragma unroll
    for (int rr = 0; rr < 3; rr++)
#pragma unroll
      for (int c = 0; c < 3; c++) {
        // X = columns 0..2 (kept by half 0, given by half 1), Y = columns 3..5 (given by half 0, kept by half 1):
        // v_permlane32_swap: X[32..63] <-> Y[0..31]
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(wr[rr][c]), __float_as_uint(wr[rr][3 + c]), false, false);
        top[rr][c] = __uint_as_float(sw[0]);
        bot[rr][c] = __uint_as_float(sw[1]);
      }
#pragma unroll
    for (int c = 0; c < 3; c++) {
      float o[6];
      f4_bt6(top[0][c], top[1][c], top[2][c], bot[0][c], bot[1][c], bot[2][c], o);
#pragma unroll
      for (int i = 0; i < 6; i++) dst[(i * 6 + c) * (32 * F4_VLD)] = o[i];
    }
  };

  // ---- B fragments (transformed weights) from global / L2: one f32x4 per position = channels 4h .. 4h+3 of output channel r ------------
  const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.u), 0, p.u_bytes, 0x00020000);
  const int u_lane = (r * F4_CK + 4 * h) * 4;                                         // bytes, per lane
  const int u_pos = p.cout_pad * (F4_CK * 4);                                         // bytes between positions
  const int u_slice = 36 * u_pos;
  const int u_base = (9 * pg) * u_pos + (n0 + cb * 32) * (F4_CK * 4);                 // wave-uniform
  const int n_slices = p.cin / F4_CK;
  const int last = n_slices - 1;
  f32x4 bq[9];
  auto b_load_one = [&](int slice, int pi) {
#ifdef F4_DIAG_NO_BLOAD
    if (slice > 0) return;                                   // timing-only build: B stays in registers
#endif
    bq[pi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, u_lane, u_base + min(slice, last) * u_slice + pi * u_pos, 0));
  };

  f32x16 acc[9];
#pragma unroll
  for (int i = 0; i < 9; i++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc[i][e] = 0.f;

  const int a_off = ((9 * pg) * 32 + r) * F4_VLD + 4 * h;
  // multiply slice s (V[vbuf] x bq) and, position by position, request the NEXT slice's B fragment right after its last use
  auto multiply = [&](int vbuf, int next_slice) {
    const float *vsrc = vb + vbuf * F4_V_FLOATS + a_off;
    // A fragments run three positions ahead of the MFMAs that consume them (in-order issue: a read requested right before its first use
    // would park the wave -- and, whenever the SIMD partner is not multiplying, the matrix pipe -- for the LDS latency, nine times a slice)
    f32x4 aq[3];
#ifndef F4_DIAG_NO_AREAD
#pragma unroll
    for (int i = 0; i < 3; i++) aq[i] = *reinterpret_cast<const f32x4 *>(vsrc + i * (32 * F4_VLD));
#endif
#pragma unroll
    for (int pi = 0; pi < 9; pi++) {
#ifdef F4_DIAG_NO_AREAD
      const f32x4 a = f32x4{(float)pi, (float)lane, 1.f, 2.f};   // timing-only build: no LDS reads in the multiply
#else
      const f32x4 a = aq[pi % 3];
#endif
      acc[pi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq[pi].x, acc[pi], 0, 0, 0);
      acc[pi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq[pi].y, acc[pi], 0, 0, 0);
      acc[pi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq[pi].z, acc[pi], 0, 0, 0);
      acc[pi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq[pi].w, acc[pi], 0, 0, 0);
#ifndef F4_DIAG_NO_AREAD
      if (pi + 3 < 9) aq[pi % 3] = *reinterpret_cast<const f32x4 *>(vsrc + (pi + 3) * (32 * F4_VLD));
#endif
      b_load_one(next_slice, pi);
    }
  };

  // One pipeline step as ONE scheduling region (straight-line code): the next slice's input transform, the raw staging and this slice's
  // 36 MFMAs are interleaved instruction by instruction inside every wave (an MFMA holds the matrix pipe for 64 cycles but the issue port
  // for a fraction of that), so a wave keeps the pipe fed by itself instead of relying on its SIMD partner being in the opposite phase
  // (stamps of the phase-separated version: 7.5 k cycles per slice against 4.6 k of MFMA time; a lone multiplying wave reached ~60 % of
  // the pipe).  Nothing in it is conditional: the staging of the last steps re-fetches the last slice (clamped) into a dead buffer.
  auto step = [&](int s) {
    const int cur = s & 1, nxt = cur ^ 1;
#ifdef F4_STAMP
    if (s == 4) F4_STAMP_AT(10);
#endif
    __builtin_amdgcn_sched_barrier(0);
    // source order = dependence order the compiler must assume: the A reads of V[cur] come BEFORE the transform's stores to V[nxt] (it
    // cannot prove the two LDS windows distinct), so the transform's reads and arithmetic are free to move up between the MFMAs
    multiply(cur, s + 1);
    transform(nxt, nxt);
#ifndef F4_DIAG_NO_STAGE
    raw_store(cur);                      // raw[cur] was consumed by transform(s) one step ago; rreg holds raw(s + 2)
    raw_load(min(s + 3, last));
#endif
#ifndef F4_NO_PIPELINE_SPEC
#ifdef F4_PIPE_V2
    // full pipeline: LDS reads (the A ring's first three fragments + the transform's six raw reads) up front; per MFMA three VALU and one
    // LDS store (raw stores first -- their data is two slices old --, V stores as the column pass produces them) so that the 61 KB a slice
    // writes to LDS drain UNDER the matrix work instead of in one burst in front of the barrier; per position the A read three positions
    // ahead and the next slice's B fragment
    __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
#pragma unroll
    for (int g = 0; g < 36; g++) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);     // 3 VALU
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);     // 1 LDS store
      if ((g & 3) == 3) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // A fragment of position + 3
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // B fragment of the next slice
      }
    }
#else
#pragma unroll
    for (int g = 0; g < 36; g++) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);     // 3 VALU
    }
#endif
#endif
    __builtin_amdgcn_sched_barrier(0);
#ifdef F4_STAMP
    if (s == 4) F4_STAMP_AT(14);
#endif
#ifndef F4_DIAG_NO_BARRIER
    __syncthreads();
#endif
#ifdef F4_STAMP
    if (s == 4) F4_STAMP_AT(15);
#endif
  };

#ifndef F4_STEP_COMPILER
  // Round 3: the same step as NINE fenced blocks, one per Winograd position of the wave: block pi = the position's four MFMAs + the A read
  // three positions ahead + the next slice's B fragment + one ninth of the step's other work (raw reads / raw stores / row pass / swaps /
  // column pass with its V stores).  hipcc's own order put every LDS store of the step (61 KB per workgroup) into one burst in front of the
  // barrier and sank the A reads to just before their MFMAs (s_waitcnt lgkmcnt(0) nine times a slice); sched_barrier(0) between the blocks
  // pins the hand order, inside a block the compiler still interleaves freely.
  auto step_blocks = [&](int s, auto mfma_first_tag) {
    constexpr bool MF = decltype(mfma_first_tag)::value;
    const int cur = s & 1, nxt = cur ^ 1;
    const float *vsrc = vb + cur * F4_V_FLOATS + a_off;
    const float *tsrc = rawb + nxt * F4_RAW_FLOATS + t_src;
    float *tdst = vb + nxt * F4_V_FLOATS + t_dst;
    float *rdstb = rawb + cur * F4_RAW_FLOATS;
#ifndef F4_RING
#define F4_RING 3
#endif
    f32x4 aq[F4_RING];
#pragma unroll
    for (int i = 0; i < F4_RING; i++) aq[i] = *reinterpret_cast<const f32x4 *>(vsrc + i * (32 * F4_VLD));
    auto mm = [&](int pi) {
      const f32x4 a = aq[pi % F4_RING];
      acc[pi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq[pi].x, acc[pi], 0, 0, 0);
      acc[pi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq[pi].y, acc[pi], 0, 0, 0);
      acc[pi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq[pi].z, acc[pi], 0, 0, 0);
      acc[pi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq[pi].w, acc[pi], 0, 0, 0);
      if (pi + F4_RING < 9) aq[pi % F4_RING] = *reinterpret_cast<const f32x4 *>(vsrc + (pi + F4_RING) * (32 * F4_VLD));
      b_load_one(s + 1, pi);
    };
    auto rstore = [&](int i) {
      const f32x4 v = rreg[i];
      rdstb[rdst[i]] = v.x;
      rdstb[rdst[i] + F4_PLANE] = v.y;
      rdstb[rdst[i] + 2 * F4_PLANE] = v.z;
      rdstb[rdst[i] + 3 * F4_PLANE] = v.w;
    };
    __builtin_amdgcn_sched_barrier(0);
    f32x4 lo[3];
    float2 hi[3];
#pragma unroll
    for (int rr = 0; rr < 3; rr++) {
      lo[rr] = *reinterpret_cast<const f32x4 *>(tsrc + rr * F4_RP);
      hi[rr] = *reinterpret_cast<const float2 *>(tsrc + rr * F4_RP + 4);
    }
    // SIMD partners (waves w and w + 4) run the two halves of every block in OPPOSITE order: one issues its four MFMAs while the other does
    // its share of the transform / staging, then they swap -- in lockstep both would queue on the matrix pipe and then both leave it idle
    // (MI355X_MICROARCH.md "two waves per SIMD", item 9)
    const auto fence = [] { __builtin_amdgcn_sched_barrier(0); };
    float wr[3][6];
    float top[3][3], bot[3][3];
    auto other = [&](int blk) {
      if (blk == 0) rstore(0);
      if (blk == 1) { f4_bt6(lo[0].x, lo[0].y, lo[0].z, lo[0].w, hi[0].x, hi[0].y, wr[0]); if (F4_RAW_PER > 1) rstore(1); }
      if (blk == 2) { f4_bt6(lo[1].x, lo[1].y, lo[1].z, lo[1].w, hi[1].x, hi[1].y, wr[1]); if (F4_RAW_PER > 2) rstore(2); }
      if (blk == 3) { f4_bt6(lo[2].x, lo[2].y, lo[2].z, lo[2].w, hi[2].x, hi[2].y, wr[2]); raw_load(min(s + 3, last)); }
      if (blk == 4) {
#pragma unroll
        for (int rr = 0; rr < 3; rr++)
#pragma unroll
          for (int c = 0; c < 3; c++) {
#ifdef F4_DIAG_NO_SWAP
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
        f4_bt6(top[0][c], top[1][c], top[2][c], bot[0][c], bot[1][c], bot[2][c], o);
#pragma unroll
        for (int i = 0; i < 6; i++) tdst[(i * 6 + c) * (32 * F4_VLD)] = o[i];
      }
    };
#ifdef F4_STAMP
    if (s == 4) F4_STAMP_AT(16);
#endif
#pragma unroll
    for (int blk = 0; blk < 9; blk++) {
      if (MF) {
        mm(blk);
        fence();
        other(blk);
      } else {
        other(blk);
        fence();
        mm(blk);
      }
      fence();
#ifdef F4_STAMP
      if (s == 4) F4_STAMP_AT(17 + blk);
#endif
    }
    __syncthreads();
#ifdef F4_STAMP
    if (s == 4) F4_STAMP_AT(26);
#endif
  };
#endif

  // ---- prologue: raw(0), raw(1) -> LDS; V(0); rreg <- raw(2); B(0) ------------------------------------------------------------------------
  {
    f32x4 r0[F4_RAW_PER];
    raw_load(0);
#pragma unroll
    for (int i = 0; i < F4_RAW_PER; i++) r0[i] = rreg[i];
    raw_load(min(1, last));                                  // both slices in flight before the first wait
#pragma unroll
    for (int pi = 0; pi < 9; pi++) b_load_one(0, pi);
    float *dst = rawb;
#pragma unroll
    for (int i = 0; i < F4_RAW_PER; i++) {
      const f32x4 v = r0[i];
      dst[rdst[i]] = v.x;
      dst[rdst[i] + F4_PLANE] = v.y;
      dst[rdst[i] + 2 * F4_PLANE] = v.z;
      dst[rdst[i] + 3 * F4_PLANE] = v.w;
    }
    if (n_slices > 1) raw_store(1);
    raw_load(min(2, last));
  }
  __syncthreads();
  transform(0, 0);
  __syncthreads();

  F4_STAMP_AT(0);
#ifndef F4_STEP_COMPILER
#ifdef F4_STEP_DEPHASE
  if (cb == 0) {
    for (int s = 0; s < last; s++) step_blocks(s, std::true_type{});
  } else {
    for (int s = 0; s < last; s++) step_blocks(s, std::false_type{});
  }
#else
#ifdef F4_OTHER_FIRST
  for (int s = 0; s < last; s++) step_blocks(s, std::false_type{});
#else
  for (int s = 0; s < last; s++) step_blocks(s, std::true_type{});
#endif
#endif
#else
  for (int s = 0; s < last; s++) step(s);
#endif
  multiply(last & 1, last);              // the last slice: nothing left to transform or stage (the B reload is a harmless re-read)
  __syncthreads();                       // every wave is done reading V before the epilogue reuses the LDS
  F4_STAMP_AT(1);

  // ---- epilogue: one 32-channel half at a time through LDS: M[pos][tile][channel] -> Y = A^T M A + bias (ReLU).
  //      The four waves that own the half (one per SIMD) dump their accumulators and, their registers now free, transform it: work unit =
  //      (tile, channel quad): 36 16-byte LDS reads, 400 VALU operations, 16 16-byte global stores (1 KiB per store instruction; the first
  //      version's dword-per-lane stores and per-row recomputation cost 18 k cycles per workgroup, half of a K = 64 layer). --------------
  const int e_q = lane & 7, e_tt = (wave & 3) * 8 + (lane >> 3);
#ifdef F4_DIAG_NO_EPI
  {                                                      // timing-only build: no epilogue at all (one never-taken store keeps the MFMAs)
    float t = 0.f;
#pragma unroll
    for (int pi = 0; pi < 9; pi++) t += acc[pi][0] + acc[pi][7] + acc[pi][15];
    if (t == 1.2345e30f) p.out[tid] = t;
    return;
  }
#endif
  auto dump = [&]() {
#ifdef F4_DIAG_NO_DUMP
    if (acc[0][0] != 1.2345e30f) return;               // timing-only build
#endif
#pragma unroll
    for (int pi = 0; pi < 9; pi++) {
      const int pos = 9 * pg + pi;
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
        ms[(pos * 32 + row) * F4_MS_LD + r] = acc[pi][e];
      }
    }
  };
  // The output transform in two steps with the barrier BETWEEN them: stage 1 pulls the unit's 36 M values out of the LDS (first A^T pass on
  // the fly: 24 f32x4 stay in registers); after it the LDS is free for the other half's dump while this half computes and stores.
  f32x4 u[4][6];                                     // u[a][j] = sum_i AT[a][i] M[i][j]
  auto finish_read = [&]() {
#ifdef F4_DIAG_NO_FINISH
    return;                                            // timing-only build
#endif
    const float *src = ms + e_tt * F4_MS_LD + 4 * e_q;
#pragma unroll
    for (int j = 0; j < 6; j++) {
      f32x4 m[6];
#pragma unroll
      for (int i = 0; i < 6; i++) m[i] = *reinterpret_cast<const f32x4 *>(src + (i * 6 + j) * (32 * F4_MS_LD));
      f32x4 y[4];
      f4_at6v(m[0], m[1], m[2], m[3], m[4], m[5], y);
#pragma unroll
      for (int a = 0; a < 4; a++) {
        u[a][j] = y[a];
        asm volatile("" : "+v"(u[a][j]));            // materialise here: one column at a time (the compiler otherwise hoists all 36
      }                                              // 16-byte reads -- 144 registers -- above the arithmetic and spills)
    }
  };
  auto finish_store = [&](int half) {
#ifdef F4_DIAG_NO_FINISH
    return;                                            // timing-only build
#endif
    F4_STAMP_AT(6 + half * 2);
    const int n = n0 + half * 32 + 4 * e_q;
    if (n < p.cout) {
      const f32x4 bias = *reinterpret_cast<const f32x4 *>(p.bias + n);
      const int py = oy0 + (e_tt >> 3) * 4, px = ox0 + (e_tt & 7) * 4;
#pragma unroll
      for (int a = 0; a < 4; a++) {
        f32x4 y[4];
        f4_at6v(u[a][0], u[a][1], u[a][2], u[a][3], u[a][4], u[a][5], y);
        if (py + a < p.h) {
          float *o = p.out + ((long long)(b * p.h + py + a) * p.w + px) * p.ld_out + n;
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
#ifdef F4_DIAG_NO_STORE
              if (v.x == 1.2345e30f)                  // timing-only build: never true, keeps the values alive
#endif
#ifdef F4_DIAG_CONTIG_STORE
              // timing-only build: the same bytes to a permuted, fully contiguous place (1 KiB per store instruction, 64 KiB per half)
              *reinterpret_cast<f32x4 *>(p.out + ((((long long)lid * 2 + half) * 16 + a * 4 + c2) * 4 + (wave & 3)) * 256 + lane * 4) = v;
#elif defined(F4_DIAG_PLAIN_STORE)
              *reinterpret_cast<f32x4 *>(o + (long long)c2 * p.ld_out) = v;
#else
              // streaming (non-temporal) stores: the 131 KB a workgroup writes do not displace the weights / halo rows the running
              // workgroups keep hitting in L2 (interleaved A/B on MI355X, 20 frames: 128->128 @128^2 315 -> 298 us, 64->64 @256^2 391 -> 376)
              __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(o + (long long)c2 * p.ld_out));
#endif
            }
        }
      }
    }
  };
  // A barrier of the four waves of channel half 1 only (the hardware barrier counts all eight waves): a counter in the LDS words the M
  // image leaves free, polled with s_sleep.  Lets half 1 go from its dump to its reads without waiting for half 0 to finish issuing stores.
  auto half1_barrier = [&]() {
    volatile int *cnt = reinterpret_cast<volatile int *>(lds + F4_MS_FLOATS);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) atomicAdd(const_cast<int *>(cnt), 1);
    while (*cnt < 4) __builtin_amdgcn_s_sleep(2);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };
  // two explicit paths with matching barrier counts: on each path the accumulators are dead once dumped (a loop over the halves would
  // keep them live through the other half's transform and spill).
  //   half 0: dump | B1 | read M            | B2 | compute + stores
  //   half 1:      | B1 | (zero the counter) | B2 | dump | half-1 barrier | read M, compute + stores
  // so half 1's dump and reads run under half 0's store tail (the stores of a workgroup leave at ~10 B/clk: 2 x 6 600 cycles)
#ifdef F4_EPI_SERIAL
  if (cb == 0) {
    dump();
    __syncthreads();
    finish_read();
    finish_store(0);
    __syncthreads();
    __syncthreads();
  } else {
    __syncthreads();
    __syncthreads();
    dump();
    __syncthreads();
    finish_read();
    finish_store(1);
  }
#else
  if (cb == 0) {
    dump();
    F4_STAMP_AT(2);
    __syncthreads();
    F4_STAMP_AT(3);
    finish_read();
    F4_STAMP_AT(4);
    __syncthreads();
    finish_store(0);
    F4_STAMP_AT(5);
  } else {
    __syncthreads();
    if (tid == 256) lds[F4_MS_FLOATS] = 0.f;           // the counter word (bit pattern 0), ordered before its use by B2
    __syncthreads();
    F4_STAMP_AT(2);
    dump();
    F4_STAMP_AT(3);
    half1_barrier();
    F4_STAMP_AT(4);
    finish_read();
    finish_store(1);
    F4_STAMP_AT(5);
  }
#endif
}

int f4_geom(const pcp_conv3x3_t *d, F4Params *p) {
  if (!d || d->stride != 1) return PCP_ERR_UNSUPPORTED;
  if (d->cin <= 0 || d->cin % F4_CK != 0 || d->cout <= 0 || d->cout_pad < d->cout || d->cout_pad % F4_WBN != 0) return PCP_ERR_ARG;
  if (d->ld_in % 4 != 0 || d->ld_out % 4 != 0 || d->cout % 4 != 0 || d->batch <= 0 || d->in_h <= 0 || d->in_w <= 0) return PCP_ERR_ARG;
  p->batch = d->batch; p->h = d->in_h; p->w = d->in_w;
  p->cin = d->cin; p->cout = d->cout; p->cout_pad = d->cout_pad;
  p->ld_in = d->ld_in; p->ld_out = d->ld_out; p->relu = d->relu;
  p->tiles_x = (d->in_w + 31) / 32;
  p->tiles_y = (d->in_h + 15) / 16;
  p->n_spatial = d->batch * p->tiles_x * p->tiles_y;
  // buffer-descriptor addressing: 32-bit byte offsets (2 GiB of activations per launch, far above any BEV map here)
  const long long in_bytes = (long long)d->batch * d->in_h * d->in_w * d->ld_in * 4;
  const long long u_bytes = (long long)(d->cin / F4_CK) * 36 * d->cout_pad * F4_CK * 4;
  if (in_bytes > 0x7fffffffLL || u_bytes > 0x7fffffffLL) return PCP_ERR_UNSUPPORTED;
  p->in_bytes = (unsigned)in_bytes;
  p->u_bytes = (unsigned)u_bytes;
  return PCP_OK;
}

}  // namespace

extern "C" int pcp_conv3x3_winograd4f(const pcp_conv3x3_t *d, const float *in, const float *u_packed, const float *bias, float *out,
                                      void *stream_) {
  if (!d || !in || !u_packed || !bias || !out) return PCP_ERR_ARG;
  F4Params p;
  int rc = f4_geom(d, &p);
  if (rc != PCP_OK) return rc;
  if ((((uintptr_t)in) & 15) || (((uintptr_t)u_packed) & 15) || (((uintptr_t)out) & 15) || (((uintptr_t)bias) & 15)) return PCP_ERR_ARG;
  p.in = in; p.u = u_packed; p.bias = bias; p.out = out;
  const long long blocks = (long long)p.n_spatial * (d->cout_pad / F4_WBN);
  if (blocks <= 0 || blocks > 0x7fffffffLL) return PCP_ERR_ARG;
  hipLaunchKernelGGL(k_wino4f, dim3((unsigned)blocks), dim3(F4_THREADS), 0, (hipStream_t)stream_, p);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

#ifdef F4_STAMP
extern "C" int pcp_debug_read_f4(void *dst, size_t bytes) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(f4_dbg), bytes) == hipSuccess ? 0 : 3;
}
#endif

extern "C" int pcp_conv3x3_winograd4f_plan(const pcp_conv3x3_t *d, double *executed_flops) {
  F4Params p;
  int rc = f4_geom(d, &p);
  if (rc != PCP_OK) return rc;
  // every workgroup multiplies [32 tiles x cin] x [cin x 64] at each of the 36 Winograd positions (padding tiles / channels included)
  if (executed_flops) *executed_flops = (double)p.n_spatial * (d->cout_pad / F4_WBN) * 2.0 * 36.0 * 32.0 * d->cin * F4_WBN;
  return PCP_OK;
}
