// Weight gradient of the 3x3 convolution for the bf16 training loop (include/pcp_hip_mp.h: pcp_mp_conv3x3_wgrad):
//   dw[co][ci][ky][kx] (+)= sum over output pixels p of dy[p][co] * x[S p + (ky - 1, kx - 1)][ci]
// a GEMM whose contraction runs over PIXELS, on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Replaces the weight-gradient half of the
// autograd node of nn.Conv2d(k=3) (pcdet/models/backbones_2d/base_bev_backbone.py:30-69, dense_heads/center_head.py:24-29,75-82,
// bev_layers/v2x_fusion_disco.py:51-63) as it runs under torch.cuda.amp.autocast; round 3's fp32 kernel is csrc/wgrad.hip.
//
// Both operands are NHWC (channel-contiguous) but the MFMA wants 8 consecutive k = 8 consecutive PIXELS of one channel per lane: the rows
// are copied global -> LDS as they lie (128-byte pixel records = 64 channels, `buffer_load_dwordx4 ... lds`) and read back TRANSPOSED by
// ds_read_b64_tr_b16 (a 4-pixel x 16-channel block per 16 lanes).  The 32-byte channel chunks of a record are XOR-swizzled by bit 1 of the
// pixel position (on the source address of the copy), which makes the four pixel rows of a transposed read hit four different bank groups.
//
// workgroup (8 waves = the quadrants of a 64 (co) x 64 (ci) tile x two tap halves, five / four 32 x 32 accumulators each) owns one (co block, ci
// block) pair, one frame, one 32-column strip and a run of output rows; it walks the run TR rows at a time.  LDS: dy rows double buffered,
// x rows in a ring of three 4-row groups (every x row is fetched ONCE per strip: the halo rows of a stage are the previous / next group).
// The copy of the next stage runs under the MFMAs of the current one; one barrier per stage.  Stride 2: the x rows are stored with even
// and odd columns apart (the loader permutes), so the 16 pixels of a k step are consecutive records again.
// Partial tiles (fp32, accumulator order) go to the workspace; k_mp_wgrad_reduce sums them in split order (bitwise reproducible).
// The kernel is HBM-bound by design: a 64 x 64 x 9 tile gives 224 flop per input byte, under the bf16 ridge of ~400.
#include "pcp_common.h"
#include "../../include/pcp_hip_mp.h"
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr unsigned WG_OOB = 0x80000000u;
constexpr int WG_THREADS = 512;            // wave = (quadrant of the tile, half of the taps)
constexpr int WG_DW = 32;                 // dy columns of a strip

#ifndef WG_DEPTH_S1
#define WG_DEPTH_S1 2
#endif
template <int S> struct WgCfg;
template <> struct WgCfg<1> { static constexpr int TR = 4, XWP = 34, XHALF = 0, DEPTH = WG_DEPTH_S1; };      // XWP: x records per row in LDS; DEPTH: stages of copy in flight (2 measured no faster)
template <> struct WgCfg<2> { static constexpr int TR = 2, XWP = 72, XHALF = 36, DEPTH = 1; };     // [parity][36]; a third group ahead would not fit the LDS

struct WgParams {
  const void *x, *dy;
  float *partial;
  int batch, h, w, oh, ow, cin, cout, ld_x, ld_dy;
  int n_cob, n_cib, n_strips, n_rsplit, rows_per, n_split;
  unsigned x_bytes, dy_bytes;
};

__device__ __forceinline__ int wg_xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

#ifdef WG_STAMP
__device__ unsigned long long wg_stamps[1024 * 24];
#define WG_T(i)                                                                                         \
  do {                                                                                                  \
    if (tid == 0 && blockIdx.x < 1024 && (i) < 24) wg_stamps[blockIdx.x * 24 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define WG_T(i) do { } while (0)
#endif

template <int S>
__global__ __launch_bounds__(WG_THREADS, 2) void k_mp_wgrad3x3(WgParams p) {
  typedef WgCfg<S> C;
  constexpr int TR = C::TR, XWP = C::XWP;
  constexpr int XG_BYTES = 4 * XWP * 128;                 // one x group (4 rows)
  constexpr int XGI = XG_BYTES / 1024;                    // wave-instructions per x group (17 | 36)
  constexpr int DY_BYTES = TR * WG_DW * 128;              // one dy stage
  constexpr int DYI = DY_BYTES / 1024;                    // 16 | 8
  constexpr int NXL = (XGI + 7) / 8, NDL = (DYI + 7) / 8; // copy instructions per wave (all eight waves copy)
  constexpr int D = C::DEPTH;                             // stages the copy runs ahead: 33 KB in flight per CU (D = 1) held the strip walk at ~3 TB/s
  constexpr int NXG = D + 2, NDY = D + 1;                 // x groups / dy stages resident
  static_assert(XG_BYTES % 1024 == 0 && DY_BYTES % 1024 == 0, "whole wave instructions");
  static_assert(NXG * XG_BYTES + NDY * DY_BYTES <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NXG * XG_BYTES + NDY * DY_BYTES];
  unsigned char *xring = lds, *dybuf = lds + NXG * XG_BYTES;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  // EIGHT multiplying waves, two per SIMD: wave = (quadrant of the 64 x 64 tile, half of the taps: 0-4 | 5-8).  With ONE wave per SIMD a
  // `buffer_load ... lds` blocked the only instruction stream of the SIMD for hundreds of cycles (in-kernel stamps, tools/stamp_wg.py: a
  // stage took 7 500 cycles with its nine copies, 3 300 without); four extra loader waves did not fit the register file beside 144
  // accumulators.  Two waves with 80 / 64 accumulators do, and each one's copy stalls run under the other one's MFMAs.
  const int lw = wave & 3;                                // quadrant
  const int th = wave >> 2;                               // tap half
  const int wm = lw >> 1, wn = lw & 1;                    // co half, ci half of the 64 x 64 tile
  WG_T(0);

  // ---- which (pair, split) --------------------------------------------------------------------------------------------------------------
  const int n_pairs = p.n_cob * p.n_cib;
  int lid = wg_xcd_remap(blockIdx.x, gridDim.x);
  const int pair = lid % n_pairs;
  const int split = lid / n_pairs;
  const int cob = pair / p.n_cib, cib = pair % p.n_cib;
  int sp = split;
  const int rs = sp % p.n_rsplit;
  sp /= p.n_rsplit;
  const int strip = sp % p.n_strips;
  const int b = sp / p.n_strips;
  const int r0 = rs * p.rows_per, r1 = min(r0 + p.rows_per, p.oh);
  const int c0 = strip * WG_DW;
  const int n_st = (r1 - r0 + TR - 1) / TR;

  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t dy_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.dy), 0, p.dy_bytes, 0x00020000);

  // ---- copy plan ----------------------------------------------------------------------------------------------------------------------------
  // x group instruction k of this wave = wave-instruction i = wave + 4k: LDS piece q = i*64 + lane = record (row, pos), 16-byte piece;
  // the LDS chunk position c2' = piece >> 1 holds channel chunk c2 = c2' ^ sw(pos)
  int xrow[NXL], xcol[NXL], xsrc[NXL];
#pragma unroll
  for (int k = 0; k < NXL; ++k) {
    const int q = (wave + 8 * k) * 64 + lane;
    const int rec = q >> 3, piece = q & 7;
    xrow[k] = rec / XWP;
    const int pos = rec - xrow[k] * XWP;
    int xc = pos;
    if (S == 2) { const int par = pos >= C::XHALF ? 1 : 0, idx = pos - par * C::XHALF; xc = 2 * idx + par; }
    xcol[k] = (xc <= S * (WG_DW - 1) + 2) ? xc : -100000;                    // columns past the last tap of the strip: zeros, no traffic
    const int c2 = (piece >> 1) ^ (((pos >> 1) & 1) << 1);
    xsrc[k] = cib * 128 + ((c2 << 1) | (piece & 1)) * 16;
  }
  int drow[NDL], dcol[NDL], dsrc[NDL];
#pragma unroll
  for (int k = 0; k < NDL; ++k) {
    const int q = (wave + 8 * k) * 64 + lane;
    const int rec = q >> 3, piece = q & 7;
    drow[k] = rec / WG_DW;
    dcol[k] = rec % WG_DW;
    const int c2 = (piece >> 1) ^ (((dcol[k] >> 1) & 1) << 1);
    dsrc[k] = cob * 128 + ((c2 << 1) | (piece & 1)) * 16;
  }
  const int gx0 = S * c0 - 1;
  // one copy instruction at a time: inside the main loop they are spread over the multiply steps (issued back to back they block the wave
  // for ~200 cycles each: the CU's vector-memory path takes 64 B/clk -- measured on k_mp_conv3x3_s1, tools/stamp_mc.py)
  // per copy instruction: the byte offset for group / stage 0, its first global row and whether its column is inside the map -- the offset
  // of group g is then one multiply-add away (the copy's own address arithmetic runs on the SIMD the MFMAs want)
  unsigned xoff0[NXL], doff0[NDL];
  int xgy0[NXL], dgy0[NDL];
  bool xcol_ok[NXL], dcol_ok[NDL];
#pragma unroll
  for (int k = 0; k < NXL; ++k) {
    xgy0[k] = S * r0 - 1 + xrow[k];
    const int gx = gx0 + xcol[k];
    xcol_ok[k] = gx >= 0 && gx < p.w && (wave + 8 * k) < XGI;
    xoff0[k] = (unsigned)(((b * p.h + xgy0[k]) * p.w + gx) * p.ld_x * 2 + xsrc[k]);
  }
#pragma unroll
  for (int k = 0; k < NDL; ++k) {
    dgy0[k] = r0 + drow[k];
    const int gx = c0 + dcol[k];
    dcol_ok[k] = gx < p.ow && (wave + 8 * k) < DYI;
    doff0[k] = (unsigned)(((b * p.oh + dgy0[k]) * p.ow + gx) * p.ld_dy * 2 + dsrc[k]);
  }
  const unsigned xstep = (unsigned)(4 * p.w * p.ld_x * 2), dstep = (unsigned)(TR * p.ow * p.ld_dy * 2);
  auto issue_x_one = [&](int k, int g) {                  // x rows [S r0 - 1 + 4g, +4) -> ring slot g % NXG
    unsigned char *base = xring + (g % NXG) * XG_BYTES;
    const int i = wave + 8 * k;
    if (i < XGI) {
      const int gy = xgy0[k] + 4 * g;
      const unsigned off = (xcol_ok[k] && gy >= 0 && gy < p.h) ? xoff0[k] + (unsigned)g * xstep : WG_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_void *)(base + i * 1024), 16, (int)off, 0, 0, 0);
    }
  };
  auto issue_dy_one = [&](int k, int s) {                 // dy rows [r0 + TR s, +TR) -> buffer s % NDY
    unsigned char *base = dybuf + (s % NDY) * DY_BYTES;
    const int i = wave + 8 * k;
    if (i < DYI) {
      const unsigned off = (dcol_ok[k] && dgy0[k] + TR * s < r1) ? doff0[k] + (unsigned)s * dstep : WG_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(dy_rsrc, (lds_void *)(base + i * 1024), 16, (int)off, 0, 0, 0);
    }
  };
  auto issue_x = [&](int g) {
#pragma unroll
    for (int k = 0; k < NXL; ++k) issue_x_one(k, g);
  };
  auto issue_dy = [&](int s) {
#pragma unroll
    for (int k = 0; k < NDL; ++k) issue_dy_one(k, s);
  };

  // ---- transposed-read addresses (bytes inside a row image; see the header comment) -------------------------------------------------------
  // lane = (hh = lane >> 5, blk = (lane >> 4) & 1, q = (lane >> 2) & 3, pp = lane & 3): read t of k step kk takes pixel 16kk + 8hh + 4t + q,
  // channels 32 half + 16 blk + 4 pp .. + 3
  const int hh = lane >> 5, blk = (lane >> 4) & 1, q4 = (lane >> 2) & 3, pp = lane & 3;
  const int a_lane = (8 * hh + q4) * 128 + (((2 * wm + blk) ^ ((q4 >> 1) << 1)) * 32) + pp * 8;       // dy: pos = 16kk + 8hh + 4t + q
  int b_lane[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    int pos = 8 * hh + q4, posoff;
    if (S == 1) { pos += kx; posoff = 0; }
    else { pos += (kx >> 1); posoff = (kx & 1) * C::XHALF; }
    // the swizzle bit is bit 1 of the record position; 16kk + 4t and XHALF (36) leave it unchanged
    b_lane[kx] = (pos + posoff) * 128 + (((2 * wn + blk) ^ (((pos >> 1) & 1) << 1)) * 32) + pp * 8;
  }

  f32x16 acc[5];                                          // taps 5 th .. 5 th + 4 (the second half has four)
#pragma unroll
  for (int t = 0; t < 5; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  auto tr_read = [&](const unsigned char *addr) -> s16x4 {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(addr));
  };

  // copy instructions this wave issues per batch (x group + dy stage): what may stay in flight across the barrier when D = 2
  int n_batch = 0;
#pragma unroll
  for (int k = 0; k < NXL; ++k) n_batch += (wave + 8 * k < XGI) ? 1 : 0;
#pragma unroll
  for (int k = 0; k < NDL; ++k) n_batch += (wave + 8 * k < DYI) ? 1 : 0;
  issue_x(0);
  issue_x(1);
  issue_dy(0);
  if (D == 2 && n_st > 1) {
    issue_x(2);
    issue_dy(1);
  }
  // one stage of products for a tap half known at compile time (TH: 0 -> taps 0..4, 1 -> taps 5..8)
  auto stage = [&](auto th_tag, int s, bool copy_next) {
    constexpr int TH = decltype(th_tag)::value;
    constexpr int T0 = 5 * TH, NT = TH ? 4 : 5;
    constexpr int NIT = TR * 2, NC = NXL + NDL;            // 16-pixel k steps of a stage, copy instructions of a stage (this wave)
    constexpr int NSTEP = NIT * NT;                        // one MFMA per (k step, tap)
    const unsigned char *dyb = dybuf + (s % NDY) * DY_BYTES + a_lane;
    const int g0 = s % NXG, g1 = (s + 1) % NXG;
    // operand reads, software pipelined by hand: the two transposed reads of step st + 1 are issued BEFORE the MFMA of step st (fenced:
    // hipcc otherwise sinks every read to just in front of its use)
    auto rd_a = [&](int itn) -> bf16x8 {
      const int rr = itn >> 1, kk = itn & 1;
      const unsigned char *ap = dyb + (rr * WG_DW + 16 * kk) * 128;
      const s16x4 a0 = tr_read(ap), a1 = tr_read(ap + 4 * 128);
      return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto rd_b = [&](int st) -> bf16x8 {
      const int itn = st / NT, tap = T0 + st % NT, rr = itn >> 1, kk = itn & 1, ky = tap / 3, kx = tap % 3;
      const int rel = S * rr + ky;                                       // x row relative to group s: 0 .. 5
      const unsigned char *bp = xring + ((rel >> 2) ? g1 : g0) * XG_BYTES + (rel & 3) * (XWP * 128) + b_lane[kx] + (16 * kk) * 128;
      const s16x4 b0 = tr_read(bp), b1 = tr_read(bp + 4 * 128);
      return __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    bf16x8 aq[2], bq[2];
    aq[0] = rd_a(0);
    bq[0] = rd_b(0);
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
      const int itn = st / NT, tl = st % NT;
      if (st + 1 < NSTEP) bq[(st + 1) & 1] = rd_b(st + 1);
      if (tl == NT - 2 && itn + 1 < NIT) aq[(itn + 1) & 1] = rd_a(itn + 1);
      __builtin_amdgcn_sched_barrier(0);
      acc[tl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[itn & 1], bq[st & 1], acc[tl], 0, 0, 0);
      // this wave's copy instructions of stage s + D, spread over the k steps (x group s + D + 1 first, then dy stage s + D)
      if (tl == NT - 1 && copy_next) {
#pragma unroll
        for (int c = 0; c < NC; ++c)
          if (c * (NIT / 2) / NC + (NIT / 2) * TH == itn) {          // the two waves of a SIMD copy in different halves of the stage
            if (c < NXL) issue_x_one(c, s + D + 1);
            else issue_dy_one(c - NXL, s + D);
          }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  for (int s = 0; s < n_st; ++s) {
    // stage s needs x groups s, s + 1 and dy stage s; with D = 2 the batch issued during stage s - 1 (x group s + 2, dy stage s + 1) may
    // still be in flight: the wait leaves exactly this wave's share of it outstanding (a wave's copies complete in issue order)
    if (D == 2 && s + 1 < n_st) {
      if (n_batch == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else if (n_batch == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (s < 10) WG_T(1 + 2 * s);
    const bool copy_next = s + D < n_st;
    if (th == 0) stage(std::integral_constant<int, 0>{}, s, copy_next);
    else stage(std::integral_constant<int, 1>{}, s, copy_next);
    if (s < 10) WG_T(2 + 2 * s);
  }
  WG_T(21);

  // ---- partial tile in ACCUMULATOR order: [pair][split][tap][quadrant][quad g][lane][4] -- every store instruction of a wave writes 1 KB of
  // contiguous memory (16 bytes per lane); k_mp_wgrad_reduce undoes the permutation.  Element (t, w, g, l, i) is dw[co][ci][t] with
  // co = 32 (w >> 1) + 8 g + 4 (l >> 5) + i, ci = 32 (w & 1) + (l & 31).  (144 dword stores per wave in [co][ci] order made the tail as long
  // as the whole multiply loop -- the epilogue is store-ISSUE bound; 16-byte stores in a [ci][co] order scatter 64 pieces per instruction
  // and were slower still: profiles/r04_mp_wgrad_variants.txt) -----------------------------------------------------------------------------------
  float *dst = p.partial + ((long long)pair * p.n_split + split) * (9 * 64 * 64);
#pragma unroll
  for (int tl = 0; tl < 5; ++tl) {
    const int t = 5 * th + tl;
    if (t < 9) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4 *>(dst + ((((t * 4 + lw) * 4 + g) * 64 + lane) << 2)) =
            f32x4{acc[tl][4 * g], acc[tl][4 * g + 1], acc[tl][4 * g + 2], acc[tl][4 * g + 3]};
    }
  }
  WG_T(22);
#ifdef WG_STAMP
  if (tid == 0 && blockIdx.x < 1024) wg_stamps[blockIdx.x * 24 + 23] = (unsigned long long)n_st;
#endif
}

// dw[co][ci][tap] (+)= sum over splits, in split order
__global__ void k_mp_wgrad_reduce(const float *__restrict__ partial, int n_split, int n_cob, int n_cib, int cout, int cin, float *__restrict__ dw,
                                  int accumulate) {
  const int pair = blockIdx.y;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;             // element of the 9 x 64 x 64 tile in the partials' order
  if (e >= 9 * 64 * 64) return;
  const int i4 = e & 3, l = (e >> 2) & 63, g = (e >> 8) & 3, w = (e >> 10) & 3, tap = e >> 12;      // accumulator order of k_mp_wgrad3x3
  const int co_l = 32 * (w >> 1) + 8 * g + 4 * (l >> 5) + i4, ci_l = 32 * (w & 1) + (l & 31);
  const int cob = pair / n_cib, cib = pair % n_cib;
  const int co = cob * 64 + co_l, ci = cib * 64 + ci_l;
  if (co >= cout || ci >= cin) return;
  const float *src = partial + (long long)pair * n_split * (9 * 64 * 64) + e;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int i = 0;
  for (; i + 3 < n_split; i += 4) {                                // four loads in flight; summed in split order
    const float v0 = src[(long long)i * (9 * 64 * 64)], v1 = src[(long long)(i + 1) * (9 * 64 * 64)];
    const float v2 = src[(long long)(i + 2) * (9 * 64 * 64)], v3 = src[(long long)(i + 3) * (9 * 64 * 64)];
    s0 += v0; s1 += v1; s2 += v2; s3 += v3;
  }
  for (; i < n_split; ++i) s0 += src[(long long)i * (9 * 64 * 64)];
  const float sum = (s0 + s1) + (s2 + s3);
  float *o = dw + ((long long)co * cin + ci) * 9 + tap;
  *o = accumulate ? *o + sum : sum;
}

struct WgPlan { int n_cob, n_cib, n_strips, n_rsplit, rows_per, n_split, oh, ow; };

bool wg_plan(const pcp_mp_wgrad3x3_t *d, WgPlan &pl) {
  if (!d || d->batch <= 0 || d->in_h <= 0 || d->in_w <= 0 || d->cin <= 0 || d->cout <= 0 || (d->stride != 1 && d->stride != 2)) return false;
  if ((d->cin % 8) || (d->cout % 8) || (d->ld_x % 8) || (d->ld_dy % 8)) return false;
  if (d->x_dtype != PCP_DT_BF16 || d->dy_dtype != PCP_DT_BF16) return false;
  if (d->stride == 2 && ((d->in_h | d->in_w) & 1)) return false;
  pl.oh = d->in_h / d->stride;
  pl.ow = d->in_w / d->stride;
  pl.n_cob = (d->cout + 63) / 64;
  pl.n_cib = (d->cin + 63) / 64;
  pl.n_strips = (pl.ow + WG_DW - 1) / WG_DW;
  const int tr = d->stride == 1 ? 4 : 2;
  const long long base = (long long)d->batch * pl.n_strips * pl.n_cob * pl.n_cib;
  long long rsplit = (256 + base / 2) / base;                          // about one workgroup per CU
  const int max_rs = (pl.oh + 2 * tr - 1) / (2 * tr);                  // at least two stages per workgroup
  if (rsplit > max_rs) rsplit = max_rs;
  if (rsplit < 1) rsplit = 1;
  pl.rows_per = (int)(((pl.oh + rsplit - 1) / rsplit + tr - 1) / tr * tr);
  pl.n_rsplit = (pl.oh + pl.rows_per - 1) / pl.rows_per;
  pl.n_split = d->batch * pl.n_strips * pl.n_rsplit;
  return true;
}

}  // namespace

extern "C" {

#ifdef WG_STAMP
int pcp_debug_read_wg(void *dst, size_t bytes) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(wg_stamps), bytes < sizeof(wg_stamps) ? bytes : sizeof(wg_stamps)) == hipSuccess ? 0 : 1;
}
#endif

size_t pcp_mp_conv3x3_wgrad_workspace_bytes(const pcp_mp_wgrad3x3_t *d) {
  WgPlan pl;
  if (!wg_plan(d, pl)) return 0;
  return (size_t)pl.n_cob * pl.n_cib * pl.n_split * (9 * 64 * 64) * sizeof(float);
}

int pcp_mp_conv3x3_wgrad(const pcp_mp_wgrad3x3_t *d, const void *x, const void *dy, float *dw, void *workspace, size_t workspace_bytes,
                         void *stream) {
  WgPlan pl;
  if (!x || !dy || !dw || !workspace || !wg_plan(d, pl)) return PCP_ERR_ARG;
  if ((((uintptr_t)x) & 15) || (((uintptr_t)dy) & 15) || (((uintptr_t)workspace) & 15)) return PCP_ERR_ARG;
  if (workspace_bytes < pcp_mp_conv3x3_wgrad_workspace_bytes(d)) return PCP_ERR_WORKSPACE;
  const long long xb = (long long)d->batch * d->in_h * d->in_w * d->ld_x * 2, db = (long long)d->batch * pl.oh * pl.ow * d->ld_dy * 2;
  if (xb >= 0x7fffffffLL || db >= 0x7fffffffLL) return PCP_ERR_UNSUPPORTED;
  WgParams p;
  p.x = x; p.dy = dy; p.partial = (float *)workspace;
  p.batch = d->batch; p.h = d->in_h; p.w = d->in_w; p.oh = pl.oh; p.ow = pl.ow; p.cin = d->cin; p.cout = d->cout;
  p.ld_x = d->ld_x; p.ld_dy = d->ld_dy;
  p.n_cob = pl.n_cob; p.n_cib = pl.n_cib; p.n_strips = pl.n_strips; p.n_rsplit = pl.n_rsplit; p.rows_per = pl.rows_per; p.n_split = pl.n_split;
  p.x_bytes = (unsigned)xb; p.dy_bytes = (unsigned)db;
  hipStream_t s = (hipStream_t)stream;
  const unsigned nwg = (unsigned)(pl.n_cob * pl.n_cib * pl.n_split);
  if (d->stride == 1) hipLaunchKernelGGL(k_mp_wgrad3x3<1>, dim3(nwg), dim3(WG_THREADS), 0, s, p);
  else hipLaunchKernelGGL(k_mp_wgrad3x3<2>, dim3(nwg), dim3(WG_THREADS), 0, s, p);
  hipLaunchKernelGGL(k_mp_wgrad_reduce, dim3((9 * 64 * 64 + 255) / 256, pl.n_cob * pl.n_cib), dim3(256), 0, s, (const float *)workspace, pl.n_split,
                     pl.n_cob, pl.n_cib, d->cout, d->cin, dw, d->accumulate ? 1 : 0);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

}  // extern "C"
