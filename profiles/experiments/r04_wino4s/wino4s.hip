// a6 / a7 / a12 -- 3x3 stride-1 convolution as fused Winograd F(4x4, 3x3): a PERSISTENT eight-wave workgroup per CU with SPECIALISED waves
// (round 4).
//
// k_wino4h / k_wino4c give every wave the whole job -- raw-patch loads, input transform, operand fetches, MFMAs -- and rely on a second wave
// per SIMD to fill the matrix pipe while the first one does anything else: in-kernel stamps put the pipe at 55 % busy.  Here the
// two waves of a SIMD do DIFFERENT things:
//   waves 0-3, CONSUMERS (one per SIMD): wave w owns output channels 16w .. 16w + 15 of the item at all 36 Winograd positions (36 accumulator
//              blocks of v_mfma_f32_16x16x4_f32, as k_wino4c).  Their instruction stream is MFMAs plus operand fetches only: U^T fragments
//              from L2 a WHOLE SLICE ahead (one 16-byte load per position pair, issued into the registers a block has just consumed), V from
//              LDS two blocks ahead.  At the end of an item: output transform in registers, bias, ReLU, sixteen 16-byte stores per lane.
//   waves 4-7, PRODUCERS: raw 18 x 18 x 8 patch global -> registers -> planar LDS image (three slices ahead), input transform V = B^T d B
//              (lane pairs, nine v_permlane32_swap) into the V image of the NEXT slice.  They run one slice ahead of the consumers and do not
//              know about item boundaries: the flattened (item, slice) sequence of the workgroup is one pipeline, so an item has NO prologue
//              -- the first slices of item i + 1 are loaded and transformed while the consumers multiply the last slices of item i and run
//              its epilogue.
//   One s_barrier per slice (in front of the consumers' last block, as in k_wino4h); LDS 69 KB (raw x2, V x2).
// Item = 16 x 16 output pixels x 64 output channels; a workgroup takes a contiguous run of items (XCD-contiguous: neighbouring patches and
// one weight block per run stay in that XCD's L2).  Weights: k_wino4c's order ([cin/8][cout_pad/16][18 position pairs][64 lanes][4]).
// Arithmetic: the products, the k order and the transforms of k_wino4h / k_wino4c -- bitwise the same outputs.
#include "pcp_common.h"
#include <type_traits>

#ifdef S4_STAMP
__device__ unsigned long long s4_dbg[256 * 16];               // [workgroup][slot] (diagnostic build only)
#define S4_T(slot)                                                                                       \
  do {                                                                                                   \
    if (lane == 0 && (wave == 0 || wave == 4) && blockIdx.x < 256 && (slot) < 8) {                        \
      unsigned long long t_;                                                                             \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                        \
      s4_dbg[blockIdx.x * 16 + (wave >> 2) * 8 + (slot)] = t_;                                           \
    }                                                                                                    \
  } while (0)
#define S4_BARRIER()                                                                                     \
  do {                                                                                                   \
    unsigned long long t0_, t1_;                                                                         \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0_)::"memory"); \
    __syncthreads();                                                                                     \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1_)::"memory");                        \
    s4_wait += t1_ - t0_;                                                                                \
  } while (0)
#else
#define S4_T(slot) do { } while (0)
#define S4_BARRIER() __syncthreads()
#endif

namespace {

constexpr int S4_THREADS = 512;
constexpr int S4_PROD = 256;                              // producer threads
constexpr int S4_CK = 8;                                  // input channels per slice
constexpr int S4_RP = 20;                                 // raw plane row pitch (floats)
constexpr int S4_RAW_H = 18, S4_RAW_W = 18;
constexpr int S4_RAW_PIX = S4_RAW_H * S4_RAW_W;
constexpr int S4_PLANE = S4_RAW_H * S4_RP;                // 360
constexpr int S4_RAW_FLOATS = S4_CK * S4_PLANE;           // 2880
constexpr int S4_VP = 160;                                // V position pitch: [8 k][16 tiles] + 32
constexpr int S4_V_FLOATS = 36 * S4_VP;                   // 5760
constexpr int S4_LDS_FLOATS = 2 * S4_RAW_FLOATS + 2 * S4_V_FLOATS;      // 17280 floats = 69 KB
constexpr int S4_RAW_ITEMS = S4_RAW_PIX * 2;              // float4 items per slice (648)
constexpr int S4_RAW_PER = (S4_RAW_ITEMS + S4_PROD - 1) / S4_PROD;      // 3
constexpr int S4_WBN = 64;
constexpr int S4_UDIST = 12;                              // position pairs the consumers request their U fragments ahead (18 per slice)
constexpr int S4_RAW_DEPTH = 4;                           // steps a raw-patch load stays in flight (the step loop is unrolled by it)
#ifndef S4_VRING
#define S4_VRING 3                                         // blocks (of four positions) the V fragments are read ahead
#endif

struct S4Params {
  const float *in;
  const float *u;       // [cin/8][cout_pad/16][18 position pairs][64 lanes][4]
  const float *bias;
  float *out;
  int batch, h, w;
  int cin, cout, cout_pad;
  int ld_in, ld_out;
  int relu;
  int tiles_x, tiles_y, n_spatial;
  int n_items, items_per_wg;
  unsigned in_bytes, u_bytes;
};

__device__ __forceinline__ int xcd_remap_s4(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// B^T x for the 6-point transform (points 0, +-1, +-2, inf)
__device__ __forceinline__ void s4_bt6(const float d0, const float d1, const float d2, const float d3, const float d4, const float d5,
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

// A^T m for float4 lanes: 6 -> 4
__device__ __forceinline__ void s4_at6v(const f32x4 m0, const f32x4 m1, const f32x4 m2, const f32x4 m3, const f32x4 m4, const f32x4 m5,
                                        f32x4 (&y)[4]) {
  const f32x4 s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
  y[0] = m0 + s12 + s34;
  y[1] = d12 + 2.f * d34;
  y[2] = s12 + 4.f * s34;
  y[3] = d12 + 8.f * d34 + m5;
}

struct S4Item { int nt, b, oy0, ox0; };
__device__ __forceinline__ S4Item s4_decode(const S4Params &p, int it) {
  S4Item c;
  c.nt = it / p.n_spatial;                      // N tile is the slow index (weights stay in the XCD's L2)
  int sp = it - c.nt * p.n_spatial;
  const int tx = sp % p.tiles_x;
  sp /= p.tiles_x;
  const int ty = sp % p.tiles_y;
  c.b = sp / p.tiles_y;
  c.oy0 = ty * 16;
  c.ox0 = tx * 16;
  return c;
}

__global__ __launch_bounds__(S4_THREADS, 1) void k_wino4s(S4Params p) {
  __shared__ __attribute__((aligned(16))) float lds[S4_LDS_FLOATS];
  float *rawb = lds;                            // [2][S4_RAW_FLOATS]
  float *vb = lds + 2 * S4_RAW_FLOATS;          // [2][S4_V_FLOATS]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5;

  const int lid = xcd_remap_s4(blockIdx.x, gridDim.x);
  const int it0 = lid * p.items_per_wg;
  const int n_my = min(p.items_per_wg, p.n_items - it0);
  if (n_my <= 0) return;                        // uniform: before any barrier
  const int n_slices = p.cin / S4_CK;
  const int g_total = n_my * n_slices;
#ifdef S4_STAMP
  unsigned long long s4_wait = 0;
#endif
  S4_T(0);

  if (wave >= 4) {
    // =========================================================== producers ================================================================
    const int ptid = tid - S4_PROD, pw = wave - 4;
#ifdef S4_PROD_PRIO
    __builtin_amdgcn_s_setprio(S4_PROD_PRIO);               // the producers' short VALU / LDS instructions go first; the consumer's MFMAs fill the rest
#endif
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in), 0, p.in_bytes, 0x00020000);
    int rdst[S4_RAW_PER], rq[S4_RAW_PER], rpy[S4_RAW_PER], rpx[S4_RAW_PER];
#pragma unroll
    for (int i = 0; i < S4_RAW_PER; i++) {
      int idx = ptid + i * S4_PROD;
      if (idx >= S4_RAW_ITEMS) idx -= S4_RAW_ITEMS;        // surplus threads repeat an item
      rq[i] = idx & 1;
      const int pix = idx >> 1;
      rpy[i] = pix / S4_RAW_W;
      rpx[i] = pix % S4_RAW_W;
      rdst[i] = (4 * rq[i]) * S4_PLANE + rpy[i] * S4_RP + rpx[i];
    }
    unsigned roff[S4_RAW_PER];
    auto set_item = [&](int it) {
      const S4Item c = s4_decode(p, it);
#pragma unroll
      for (int i = 0; i < S4_RAW_PER; i++) {
        const int iy = c.oy0 - 1 + rpy[i], ix = c.ox0 - 1 + rpx[i];
        roff[i] = 0x80000000u;                             // out of range -> 0
        if (iy >= 0 && iy < p.h && ix >= 0 && ix < p.w) roff[i] = (unsigned)((((long long)(c.b * p.h + iy) * p.w + ix) * p.ld_in + rq[i] * 4) * 4);
      }
    };
    // the raw stream: slice rl_s of item it0 + rl_it next; past the last slice of the last item it keeps re-reading that slice (never used)
    int rl_it = 0, rl_s = 0;
    set_item(it0);
    // the loads of a slice stay in flight S4_RAW_DEPTH steps (an HBM miss takes longer than a step takes the consumers): a ring of register sets
    f32x4 rring[S4_RAW_DEPTH][S4_RAW_PER];
    auto raw_load = [&](f32x4 (&dstr)[S4_RAW_PER]) {
      const int soff = rl_s * (S4_CK * 4);
#pragma unroll
      for (int i = 0; i < S4_RAW_PER; i++)
        dstr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)roff[i], soff, 0));
      if (rl_s + 1 < n_slices) {
        ++rl_s;
      } else if (rl_it + 1 < n_my) {
        rl_s = 0;
        ++rl_it;
        set_item(it0 + rl_it);
      }
    };
    auto raw_store = [&](int buf, const f32x4 (&srcr)[S4_RAW_PER]) {
      float *dst = rawb + buf * S4_RAW_FLOATS;
#pragma unroll
      for (int i = 0; i < S4_RAW_PER; i++) {
        const f32x4 v = srcr[i];
        dst[rdst[i]] = v.x;
        dst[rdst[i] + S4_PLANE] = v.y;
        dst[rdst[i] + 2 * S4_PLANE] = v.z;
        dst[rdst[i] + 3 * S4_PLANE] = v.w;
      }
    };
    // input transform: item = (tile, channel) on the lane pair (l, l + 32); producer wave pw owns channels 2 pw, 2 pw + 1 of all 16 tiles
    const int t_li = lane & 31;
    const int t_tile = t_li & 15, t_ch = 2 * pw + (t_li >> 4);
    const int t_src = t_ch * S4_PLANE + (4 * (t_tile >> 2) + 3 * h) * S4_RP + 4 * (t_tile & 3);
    const int t_dst = t_ch * 16 + t_tile + (3 * h) * S4_VP;
    f32x4 lo[3];
    float2 hi[3];
    auto rows_load = [&](int rbuf) {
      const float *src = rawb + rbuf * S4_RAW_FLOATS + t_src;
#pragma unroll
      for (int rr = 0; rr < 3; rr++) {
        lo[rr] = *reinterpret_cast<const f32x4 *>(src + rr * S4_RP);
        hi[rr] = *reinterpret_cast<const float2 *>(src + rr * S4_RP + 4);
      }
    };
    auto transform = [&](int vbuf) {                      // lo / hi -> V[vbuf]
      float *dst = vb + vbuf * S4_V_FLOATS + t_dst;
      float wr[3][6];
#pragma unroll
      for (int rr = 0; rr < 3; rr++) s4_bt6(lo[rr].x, lo[rr].y, lo[rr].z, lo[rr].w, hi[rr].x, hi[rr].y, wr[rr]);
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
        s4_bt6(top[0][c], top[1][c], top[2][c], bot[0][c], bot[1][c], bot[2][c], o);
#pragma unroll
        for (int i = 0; i < 6; i++) dst[(i * 6 + c) * S4_VP] = o[i];
      }
    };

    // ---- prologue: raw(0), raw(1) -> LDS; V(0); the ring <- raw(2 .. 1 + S4_RAW_DEPTH) --------------------------------------------------------
    {
      f32x4 r0[S4_RAW_PER], r1[S4_RAW_PER];
      raw_load(r0);
      raw_load(r1);
#pragma unroll
      for (int d = 0; d < S4_RAW_DEPTH; d++) raw_load(rring[d]);
      raw_store(0, r0);
      raw_store(1, r1);
    }
    __syncthreads();                                       // A: raw[0], raw[1] complete
    rows_load(0);
    transform(0);
    __syncthreads();                                       // B: V[0] complete
    rows_load(1);
    S4_T(1);
    // step g: slice g + 2 (ring slot g % DEPTH) -> raw[g & 1]; slice g + 1 -> V[(g + 1) & 1]; request slice g + 2 + DEPTH into the slot
    auto step = [&](auto slot_tag, int g) {
      constexpr int SLOT = decltype(slot_tag)::value;
      const int cur = g & 1, nxt = cur ^ 1;
#ifndef S4_DIAG_NO_PRODUCE                // (timing-only build: the producers only keep the barrier count)
      raw_store(cur, rring[SLOT]);
      transform(nxt);
      raw_load(rring[SLOT]);
#endif
      S4_BARRIER();                                        // the step barrier: V[nxt] and raw[cur] complete
      rows_load(cur);                                      // slice g + 2, for the next step's transform
    };
    int g = 0;
    for (; g + S4_RAW_DEPTH <= g_total; g += S4_RAW_DEPTH) {
      step(std::integral_constant<int, 0>{}, g);
      step(std::integral_constant<int, 1>{}, g + 1);
      step(std::integral_constant<int, 2>{}, g + 2);
      step(std::integral_constant<int, 3>{}, g + 3);
    }
    if (g < g_total) { step(std::integral_constant<int, 0>{}, g); ++g; }
    if (g < g_total) { step(std::integral_constant<int, 1>{}, g); ++g; }
    if (g < g_total) { step(std::integral_constant<int, 2>{}, g); ++g; }
    S4_T(2);
#ifdef S4_STAMP
    if (lane == 0 && wave == 4 && blockIdx.x < 256) s4_dbg[blockIdx.x * 16 + 8 + 3] = s4_wait;
#endif
    return;
  }

  // ============================================================= consumers ==================================================================
  const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.u), 0, p.u_bytes, 0x00020000);
  const int u_lane = lane * 16;                                                       // bytes, per lane
  const int u_slice = (p.cout_pad / 16) * (18 * 64 * 4 * 4);                          // bytes between slices
  const int u_blk = 18 * 64 * 4 * 4;                                                  // bytes of one 16-channel block of a slice
  // U ring: position pair q of step g lives in slot (q + 6 (g & 1)) % 12 and is requested S4_UDIST = 12 pairs (24 positions, 48 MFMAs) ahead,
  // into the slot the pair just consumed leaves -- the step loop is unrolled by two so that every slot index is a compile-time constant
  f32x4 uq[S4_UDIST];
  auto u_load = [&](int off, int q, int slot) __attribute__((always_inline)) {
    uq[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, u_lane, off + q * (64 * 4 * 4), 0));
  };
  S4Item item = s4_decode(p, it0);
  int u_off = (item.nt * 4 + wave) * u_blk;                                           // (item, slice 0): wave-uniform
#pragma unroll
  for (int q = 0; q < S4_UDIST; q++) u_load(u_off, q, q);

  f32x4 acc[36];
#pragma unroll
  for (int i = 0; i < 36; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float vq[S4_VRING][4][2];
  const int v_off = lane;
  __syncthreads();                                         // A
  __syncthreads();                                         // B: V[0] complete
#pragma unroll
  for (int i = 0; i < S4_VRING; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      vq[i][j][0] = vb[v_off + (4 * i + j) * S4_VP];
      vq[i][j][1] = vb[v_off + (4 * i + j) * S4_VP + 64];
    }
  S4_T(1);

  const auto fence = [] { __builtin_amdgcn_sched_barrier(0); };
  // block pi of a step with ring phase PAR = positions 4 pi .. 4 pi + 3: k step 0 of the four, then k step 1 (an accumulator's two MFMAs are
  // four instructions apart); then the V fragments S4_VRING blocks ahead and the U pairs S4_UDIST pairs ahead (into the slots just read)
  auto mm = [&](auto par_tag, auto pi_tag, const float *vsrc, int u_nxt) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_tag)::value, pi = decltype(pi_tag)::value;
    constexpr int qa = 2 * pi, qb = 2 * pi + 1;
    constexpr int sa = (qa + 6 * PAR) % S4_UDIST, sb = (qb + 6 * PAR) % S4_UDIST;
    const f32x4 ua = uq[sa], ub = uq[sb];
    float (&v)[4][2] = vq[pi % S4_VRING];
#ifdef S4_DIAG_NO_MFMA                    // (timing-only build: the producers alone)
    if (ua.x == 1.2345e30f)
#endif
    {
    acc[4 * pi + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua.x, v[0][0], acc[4 * pi + 0], 0, 0, 0);
    acc[4 * pi + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua.z, v[1][0], acc[4 * pi + 1], 0, 0, 0);
    acc[4 * pi + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ub.x, v[2][0], acc[4 * pi + 2], 0, 0, 0);
    acc[4 * pi + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(ub.z, v[3][0], acc[4 * pi + 3], 0, 0, 0);
    acc[4 * pi + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua.y, v[0][1], acc[4 * pi + 0], 0, 0, 0);
    acc[4 * pi + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua.w, v[1][1], acc[4 * pi + 1], 0, 0, 0);
    acc[4 * pi + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ub.y, v[2][1], acc[4 * pi + 2], 0, 0, 0);
    acc[4 * pi + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(ub.w, v[3][1], acc[4 * pi + 3], 0, 0, 0);
    }
#ifndef S4_DIAG_NO_VREAD
    if (pi + S4_VRING < 9) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        v[j][0] = vsrc[(4 * (pi + S4_VRING) + j) * S4_VP];
        v[j][1] = vsrc[(4 * (pi + S4_VRING) + j) * S4_VP + 64];
      }
    }
#endif
#ifndef S4_DIAG_NO_ULOAD                  // (timing-only builds: wrong results)
    if (qa + S4_UDIST < 18) u_load(u_off, qa + S4_UDIST, sa); else u_load(u_nxt, qa + S4_UDIST - 18, sa);
    if (qb + S4_UDIST < 18) u_load(u_off, qb + S4_UDIST, sb); else u_load(u_nxt, qb + S4_UDIST - 18, sb);
#endif
  };
  auto blocks07 = [&](auto par_tag, int u_nxt) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_tag)::value;
    const float *vsrc = vb + PAR * S4_V_FLOATS + v_off;
    fence();
    mm(par_tag, std::integral_constant<int, 0>{}, vsrc, u_nxt); fence();
    mm(par_tag, std::integral_constant<int, 1>{}, vsrc, u_nxt); fence();
    mm(par_tag, std::integral_constant<int, 2>{}, vsrc, u_nxt); fence();
    mm(par_tag, std::integral_constant<int, 3>{}, vsrc, u_nxt); fence();
    mm(par_tag, std::integral_constant<int, 4>{}, vsrc, u_nxt); fence();
    mm(par_tag, std::integral_constant<int, 5>{}, vsrc, u_nxt); fence();
    mm(par_tag, std::integral_constant<int, 6>{}, vsrc, u_nxt); fence();
    mm(par_tag, std::integral_constant<int, 7>{}, vsrc, u_nxt); fence();
  };
  // a step that is not an item's last: the next slice's first V blocks are requested in front of the last block (its eight MFMAs cover their
  // LDS latency)
  auto step_mid = [&](auto par_tag) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_tag)::value;
    const int u_nxt = u_off + u_slice;
    blocks07(par_tag, u_nxt);
    S4_BARRIER();                            // V[PAR ^ 1] complete; every read of V[PAR] has returned
    const float *vn = vb + (PAR ^ 1) * S4_V_FLOATS + v_off;
#pragma unroll
    for (int i = 0; i < S4_VRING; i++)
      if (i != 8 % S4_VRING) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          vq[i][j][0] = vn[(4 * i + j) * S4_VP];
          vq[i][j][1] = vn[(4 * i + j) * S4_VP + 64];
        }
      }
    fence();
    mm(par_tag, std::integral_constant<int, 8>{}, vn, u_nxt);
    fence();
#pragma unroll
    for (int j = 0; j < 4; j++) {
      vq[8 % S4_VRING][j][0] = vn[(4 * (8 % S4_VRING) + j) * S4_VP];
      vq[8 % S4_VRING][j][1] = vn[(4 * (8 % S4_VRING) + j) * S4_VP + 64];
    }
    fence();
    u_off = u_nxt;
  };
  // an item's last step: the U ring runs on into the next item; output transform in registers; the V ring is refilled behind it (nothing of it
  // is live across the transform)
  int it_rel = 0;
  auto step_last = [&](auto par_tag) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_tag)::value;
    S4Item next_item = item;
    int u_nxt = u_off;
    if (it_rel + 1 < n_my) {
      next_item = s4_decode(p, it0 + it_rel + 1);
      u_nxt = (next_item.nt * 4 + wave) * u_blk;
    }
    blocks07(par_tag, u_nxt);
    S4_BARRIER();
    const float *vn = vb + (PAR ^ 1) * S4_V_FLOATS + v_off;
    fence();
    mm(par_tag, std::integral_constant<int, 8>{}, vn, u_nxt);
    fence();
    u_off = u_nxt;
    // ---- Y = A^T M A per lane (tile = lane & 15, channel quad = lane >> 4), bias, ReLU, sixteen 16-byte streaming stores --------------------
#ifdef S4_DIAG_NO_EPI
    if (acc[0].x == 1.2345e30f)
#endif
    {
      const int e_tile = lane & 15, e_kq = lane >> 4;
      const int n = item.nt * S4_WBN + 16 * wave + 4 * e_kq;
      const bool n_ok = n < p.cout;
      const f32x4 bias = n_ok ? *reinterpret_cast<const f32x4 *>(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
      const int py = item.oy0 + (e_tile >> 2) * 4, px = item.ox0 + (e_tile & 3) * 4;
      // column pass IN PLACE (rows 0..3 of A^T M replace the accumulators of positions (0..3, j)), one fenced column / row at a time
#pragma unroll
      for (int j = 0; j < 6; j++) {
        f32x4 y[4];
        s4_at6v(acc[j], acc[6 + j], acc[12 + j], acc[18 + j], acc[24 + j], acc[30 + j], y);
        acc[j] = y[0];
        acc[6 + j] = y[1];
        acc[12 + j] = y[2];
        acc[18 + j] = y[3];
        fence();
      }
#pragma unroll
      for (int r = 0; r < 4; r++) {
        f32x4 y[4];
        s4_at6v(acc[6 * r], acc[6 * r + 1], acc[6 * r + 2], acc[6 * r + 3], acc[6 * r + 4], acc[6 * r + 5], y);
        if (n_ok && py + r < p.h) {
          float *o = p.out + ((long long)(item.b * p.h + py + r) * p.w + px) * p.ld_out + n;
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
        fence();
      }
#pragma unroll
      for (int i = 0; i < 36; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < S4_VRING; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        vq[i][j][0] = vn[(4 * i + j) * S4_VP];
        vq[i][j][1] = vn[(4 * i + j) * S4_VP + 64];
      }
    ++it_rel;
    item = next_item;
  };
  // an item = an EVEN number of steps (the launcher refuses cin % 16 != 0): the ring phase of every step is a compile-time constant
  for (int i = 0; i < n_my; ++i) {
    for (int k = 0; k + 2 < n_slices; k += 2) {
      step_mid(std::integral_constant<int, 0>{});
      step_mid(std::integral_constant<int, 1>{});
    }
    step_mid(std::integral_constant<int, 0>{});
    step_last(std::integral_constant<int, 1>{});
  }
  S4_T(2);
#ifdef S4_STAMP
  if (lane == 0 && wave == 0 && blockIdx.x < 256) s4_dbg[blockIdx.x * 16 + 3] = s4_wait;
#endif
}

int s4_geom(const pcp_conv3x3_t *d, S4Params *p) {
  if (!d || d->stride != 1) return PCP_ERR_UNSUPPORTED;
  if (d->cin <= 0 || d->cin % S4_CK != 0 || d->cout <= 0 || d->cout_pad < d->cout || d->cout_pad % S4_WBN != 0) return PCP_ERR_ARG;
  if (d->ld_in % 4 != 0 || d->ld_out % 4 != 0 || d->cout % 4 != 0 || d->batch <= 0 || d->in_h <= 0 || d->in_w <= 0) return PCP_ERR_ARG;
  if (d->cin % (2 * S4_CK) != 0) return PCP_ERR_UNSUPPORTED;          // an even number of slices per item (the U ring's phase is compiled in)
  p->batch = d->batch; p->h = d->in_h; p->w = d->in_w;
  p->cin = d->cin; p->cout = d->cout; p->cout_pad = d->cout_pad;
  p->ld_in = d->ld_in; p->ld_out = d->ld_out; p->relu = d->relu;
  p->tiles_x = (d->in_w + 15) / 16;
  p->tiles_y = (d->in_h + 15) / 16;
  p->n_spatial = d->batch * p->tiles_x * p->tiles_y;
  const long long items = (long long)p->n_spatial * (d->cout_pad / S4_WBN);
  const long long in_bytes = (long long)d->batch * d->in_h * d->in_w * d->ld_in * 4;
  const long long u_bytes = (long long)(d->cin / S4_CK) * 36 * d->cout_pad * S4_CK * 4;
  if (in_bytes > 0x7fffffffLL || u_bytes > 0x7fffffffLL || items > 0x7fffffffLL) return PCP_ERR_UNSUPPORTED;
  p->n_items = (int)items;
  p->in_bytes = (unsigned)in_bytes;
  p->u_bytes = (unsigned)u_bytes;
  return PCP_OK;
}

}  // namespace

extern "C" int pcp_conv3x3_winograd4s(const pcp_conv3x3_t *d, const float *in, const float *u_packed, const float *bias, float *out,
                                      void *stream_) {
  if (!d || !in || !u_packed || !bias || !out) return PCP_ERR_ARG;
  S4Params p;
  int rc = s4_geom(d, &p);
  if (rc != PCP_OK) return rc;
  if ((((uintptr_t)in) & 15) || (((uintptr_t)u_packed) & 15) || (((uintptr_t)out) & 15) || (((uintptr_t)bias) & 15)) return PCP_ERR_ARG;
  p.in = in; p.u = u_packed; p.bias = bias; p.out = out;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return PCP_ERR_LAUNCH;
    n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const int wgs = p.n_items < n_cu ? p.n_items : n_cu;                // one persistent workgroup per CU
  p.items_per_wg = (p.n_items + wgs - 1) / wgs;
  const int grid = (p.n_items + p.items_per_wg - 1) / p.items_per_wg;
  hipLaunchKernelGGL(k_wino4s, dim3((unsigned)grid), dim3(S4_THREADS), 0, (hipStream_t)stream_, p);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

#ifdef S4_STAMP
extern "C" int pcp_debug_read_s4(void *dst, size_t bytes) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(s4_dbg), bytes) == hipSuccess ? 0 : 3;
}
#endif

extern "C" int pcp_conv3x3_winograd4s_plan(const pcp_conv3x3_t *d, double *executed_flops) {
  S4Params p;
  int rc = s4_geom(d, &p);
  if (rc != PCP_OK) return rc;
  // every item multiplies [16 tiles x cin] x [cin x 64] at each of the 36 Winograd positions (padding tiles / channels included)
  if (executed_flops) *executed_flops = (double)p.n_items * 2.0 * 36.0 * 16.0 * d->cin * S4_WBN;
  return PCP_OK;
}
