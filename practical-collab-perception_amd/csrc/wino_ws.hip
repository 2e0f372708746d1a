// a6 / a7 / a12 -- 3x3 stride-1 convolution as fused Winograd F(2x2,3x3), "wave-stationary" form (gfx950, fp32 MFMA).
//
// Same arithmetic as wino.hip (16 products per 2x2 output tile instead of 36; exact-fp32 MFMA accumulation) with a different
// decomposition, built for the SHORT-K layers (64 / 128 input channels: the BEV backbone blocks, the CenterHead branches, DiscoNet's
// compressor) where wino.hip's per-8-channel  transform -> LDS -> barrier -> multiply  chain, not the matrix pipe, set the pace
// (PMC: matrix pipe 56 % busy, waves parked 36 %):
//
//   * a wave owns 32 Winograd tiles (8 x 16 output pixels) x 32 output channels x 8 of the 16 Winograd positions (the position rows
//     {0,1} or {2,3}: 8 accumulator tiles = 128 VGPRs) for the WHOLE contraction;
//   * the MFMA A operand (transformed input V, lane = (tile, channel)) is produced in the wave's own registers: the lane reads the three
//     raw input rows its position half needs (conflict-free ds_read_b64 from a channel-planar LDS image), 16 VALU operations give its
//     8 position values -- V never goes through LDS and there is NO barrier between transform and multiply;
//   * the B operand (transformed weights U = G g G^T) streams from L2 straight into registers, two coalesced 1-KiB loads per channel
//     pair, prefetched two channel pairs ahead (the host packs it in exactly that fragment order);
//   * the raw input patch is staged global -> registers -> LDS in chunks of CHUNK channels, double buffered: ONE workgroup barrier per
//     CHUNK / 2 channel pairs (= 8 or 16 K steps of 8 MFMAs per wave) instead of one per 8 channels;
//   * 4 waves per workgroup (one per SIMD), TWO workgroups per CU (256-VGPR budget, 62 KB of LDS each): the two waves of a SIMD belong
//     to different workgroups, so one workgroup's prologue (patch fetch) and epilogue (transform, exchange, stores) run under the
//     other's MFMAs -- measured: with one 8-wave workgroup per CU those serial phases were 10 % (K = 128) to 30 % (K = 64) of the kernel;
//   * epilogue in registers: each wave applies A^T (.) A to its position half (linear, so the halves just add), exchanges ONE 2x2-tile
//     row with its partner through LDS, adds bias, ReLU, and stores 128-byte rows.
//
// Workgroup = 32 tiles (8 x 16 output pixels) x 64 output channels = {2 cout blocks of 32} x {2 position halves}.
#include "pcp_common.h"

namespace {

constexpr int RP = 24;            // row pitch of a channel plane in floats: 2 * RP * ty mod 64 = {0, 48, 32, 16} -> the 32 tiles of a
                                  // ds_read_b64 lane group hit 64 distinct banks
constexpr int RAW_W = 18;
constexpr int WS_THREADS = 256;

struct WsParams {
  const float *in;
  const float *u;       // [cin/2][cout_pad/32][2 (position half)][2 (row of the half)][64 lanes][4 (position column)]
  const float *bias;
  float *out;
  int batch, h, w;
  int cin, cout, cout_pad;
  int ld_in, ld_out;
  int relu;
  int tiles_x, tiles_y, n_spatial;
};

__device__ __forceinline__ int xcd_remap_ws(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

template <int CHUNK>
struct WsCfg {
  static constexpr int CB = 2;                      // 32-channel output blocks per workgroup
  static constexpr int WBN = 32 * CB;
  static constexpr int OUT_H = 8;
  static constexpr int RAW_H = OUT_H + 2;
  static constexpr int RAW_PIX = RAW_H * RAW_W;
  static constexpr int PLANE = RAW_H * RP + 2;      // +2: consecutive channel quads of one pixel land 8 banks apart when staged
  static constexpr int BUF = CHUNK * PLANE;
  static constexpr int Q = CHUNK / 4;
  static constexpr int ITEMS = RAW_PIX * Q;         // float4 items per stage
  static constexpr int PER = (ITEMS + WS_THREADS - 1) / WS_THREADS;
  static constexpr int KP = CHUNK / 2;              // channel pairs (MFMA K steps) per stage
  static constexpr int XCH = 2 * 2 * 32 * 64;       // epilogue exchange: 2 wave pairs x 2 writers x 32 values x 64 lanes
  static constexpr int LDS_FLOATS = 2 * BUF > XCH ? 2 * BUF : XCH;
};

template <int CHUNK>
__global__ __launch_bounds__(WS_THREADS, 2) void k_wino_ws(WsParams p) {
  using C = WsCfg<CHUNK>;
  __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int ph = wave >> 1;                     // position half: Winograd position rows {0,1} or {2,3}
  const int cb = wave & 1;                      // 32-channel output block inside the workgroup's 64-channel N tile

  const int lid = xcd_remap_ws(blockIdx.x, gridDim.x);
  const int nt = lid / p.n_spatial;             // N tile is the slow index (its weights stay in the XCD's L2)
  int sp = lid % p.n_spatial;
  const int tile_x = sp % p.tiles_x;
  sp /= p.tiles_x;
  const int tile_y = sp % p.tiles_y;
  const int b = sp / p.tiles_y;
  const int oy0 = tile_y * C::OUT_H, ox0 = tile_x * 16;
  const int nblk = p.cout_pad >> 5;
  const int blk = nt * C::CB + cb;

  // ---- raw patch staging: item = (pixel, channel quad), quad fastest (the CHUNK * 4 bytes of a pixel are one contiguous run).
  //      Loads are unconditional with clamped addresses (static VMEM counts); pixels outside the image are zeroed by a select. -------
  unsigned roff[C::PER];
  int rdst[C::PER];
  bool rin[C::PER];
#pragma unroll
  for (int i = 0; i < C::PER; i++) {
    const int idx = tid + i * WS_THREADS;
    roff[i] = 0u;
    rdst[i] = -1;
    rin[i] = false;
    if (idx < C::ITEMS) {
      const int q = idx % C::Q, pix = idx / C::Q;
      const int py = pix / RAW_W, px = pix % RAW_W;
      const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
      rdst[i] = (4 * q) * C::PLANE + py * RP + px;
      if (iy >= 0 && iy < p.h && ix >= 0 && ix < p.w) {
        rin[i] = true;
        roff[i] = (unsigned)((((long long)(b * p.h + iy) * p.w + ix) * p.ld_in + q * 4) * 4);
      }
    }
  }
  f32x4 rreg[C::PER];
  auto raw_load = [&](int stage) {
    const char *base = reinterpret_cast<const char *>(p.in + stage * CHUNK);       // wave-uniform
#pragma unroll
    for (int i = 0; i < C::PER; i++) rreg[i] = *reinterpret_cast<const f32x4 *>(base + roff[i]);
  };
  auto raw_store = [&](int buf) {
    float *dst = lds + buf * C::BUF;
#pragma unroll
    for (int i = 0; i < C::PER; i++)
      if (rdst[i] >= 0) {
        f32x4 v = rreg[i];
        if (!rin[i]) v = f32x4{0.f, 0.f, 0.f, 0.f};
        dst[rdst[i]] = v.x;
        dst[rdst[i] + C::PLANE] = v.y;
        dst[rdst[i] + 2 * C::PLANE] = v.z;
        dst[rdst[i] + 3 * C::PLANE] = v.w;
      }
  };

  // ---- A operand: lane = (tile r = ty * 8 + tx, channel 2 * kp + h).  Position half 0 needs V rows 0, 1 = (d0 - d2, d1 + d2) B,
  //      half 1 needs V rows 2, 3 = (d2 - d1, d1 - d3) B: with (P, Q, S) = raw rows (0, 2, 1) resp. (2, 1, 3) both are
  //      (P - Q, Q + sgn * S), sgn = +1 / -1 -- one instruction stream for both halves. ------------------------------------------------
  const int ty = r >> 3, tx = r & 7;
  const int row0 = 2 * ty;
  const int lane_base = h * C::PLANE + 2 * tx;
  const int offP = lane_base + (row0 + (ph ? 2 : 0)) * RP;
  const int offQ = lane_base + (row0 + (ph ? 1 : 2)) * RP;
  const int offS = lane_base + (row0 + (ph ? 3 : 1)) * RP;
  const float sgn = ph ? -1.f : 1.f;

  struct Raw { float2 p0, p1, q0, q1, s0, s1; };
  auto raw_read = [&](const float *buf, int kp) {
    const float *src = buf + 2 * kp * C::PLANE;
    Raw d;
#ifdef WS_DIAG_NO_LDSREAD
    d.p0 = d.p1 = d.q0 = d.q1 = d.s0 = d.s1 = float2{sgn * (float)kp, (float)lane};   // timing-only build: no LDS reads in the loop
    return d;
#endif
    d.p0 = *reinterpret_cast<const float2 *>(src + offP);
    d.p1 = *reinterpret_cast<const float2 *>(src + offP + 2);
    d.q0 = *reinterpret_cast<const float2 *>(src + offQ);
    d.q1 = *reinterpret_cast<const float2 *>(src + offQ + 2);
    d.s0 = *reinterpret_cast<const float2 *>(src + offS);
    d.s1 = *reinterpret_cast<const float2 *>(src + offS + 2);
    return d;
  };
  auto transform = [&](const Raw &d, float (&a)[8]) {
#ifdef WS_DIAG_NO_XFORM
    a[0] = d.p0.x; a[1] = d.p0.y; a[2] = d.p1.x; a[3] = d.p1.y; a[4] = d.q0.x; a[5] = d.q0.y; a[6] = d.s0.x; a[7] = d.s1.y;   // timing only
    return;
#endif
    const float ta0 = d.p0.x - d.q0.x, ta1 = d.p0.y - d.q0.y, ta2 = d.p1.x - d.q1.x, ta3 = d.p1.y - d.q1.y;
    const float tb0 = fmaf(sgn, d.s0.x, d.q0.x), tb1 = fmaf(sgn, d.s0.y, d.q0.y), tb2 = fmaf(sgn, d.s1.x, d.q1.x),
                tb3 = fmaf(sgn, d.s1.y, d.q1.y);
    a[0] = ta0 - ta2;
    a[1] = ta1 + ta2;
    a[2] = ta2 - ta1;
    a[3] = ta1 - ta3;
    a[4] = tb0 - tb2;
    a[5] = tb1 + tb2;
    a[6] = tb2 - tb1;
    a[7] = tb1 - tb3;
  };

  // ---- B operand ring: 4 slots of (2 x float4), prefetch distance 2 channel pairs.  The host pads the packed weights with two zero
  //      channel pairs, so the prefetch runs past the end without a clamp (uniform base + 32-bit lane offset: no vector address math) ---
  const long long u_kp_bytes = (long long)nblk * 4096;                               // bytes per channel pair
  const char *ubase = reinterpret_cast<const char *>(p.u + ((long long)blk * 4 + ph * 2) * 256);
  const unsigned u_lane = (unsigned)lane * 16u;
  f32x4 bq[4][2];
  auto b_load = [&](int kpg, f32x4 (&dst)[2]) {
#ifdef WS_DIAG_NO_BLOAD
    if (kpg > 1) return;                                                             // timing-only build: B stays in registers
#endif
    const char *s = ubase + kpg * u_kp_bytes;                                        // wave-uniform
    dst[0] = *reinterpret_cast<const f32x4 *>(s + u_lane);
    dst[1] = *reinterpret_cast<const f32x4 *>(s + 1024 + u_lane);
  };

  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc[i][e] = 0.f;

  const int n_stages = p.cin / CHUNK;

  // ---- prologue ---------------------------------------------------------------------------------------------------------------------
  raw_load(0);
  b_load(0, bq[0]);
  b_load(1, bq[1]);
  raw_store(0);
  __syncthreads();

  for (int s = 0; s < n_stages; s++) {
    const float *buf = lds + (s & 1) * C::BUF;
    const int kp0 = s * C::KP;
    float a_cur[8];
    {
      Raw d = raw_read(buf, 0);
      transform(d, a_cur);
    }
#pragma unroll
    for (int kp = 0; kp < C::KP; kp++) {
      Raw dn;
      if (kp + 1 < C::KP) dn = raw_read(buf, kp + 1);
      b_load(kp0 + kp + 2, bq[(kp + 2) & 3]);
      // the next stage's raw patch is requested late in this stage so that the in-order vmcnt of the B fragments never waits on it
#ifndef WS_DIAG_NO_STAGE
      if (kp == C::KP - 3) raw_load(min(s + 1, n_stages - 1));
#endif
      __builtin_amdgcn_sched_barrier(0);          // every request above stays ahead of this step's MFMAs in program order
      const f32x4 b0 = bq[kp & 3][0], b1 = bq[kp & 3][1];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[0], b0.x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[1], b0.y, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[2], b0.z, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[3], b0.w, acc[3], 0, 0, 0);
      acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[4], b1.x, acc[4], 0, 0, 0);
      acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[5], b1.y, acc[5], 0, 0, 0);
      acc[6] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[6], b1.z, acc[6], 0, 0, 0);
      acc[7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[7], b1.w, acc[7], 0, 0, 0);
      float a_nxt[8];
      if (kp + 1 < C::KP) {
        transform(dn, a_nxt);
        // the 16 VALU operations of the next step's transform ride in the issue gaps of this step's MFMAs
#pragma unroll
        for (int g = 0; g < 8; g++) {
          __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x2, 2, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (kp + 1 < C::KP) {
#pragma unroll
        for (int i = 0; i < 8; i++) a_cur[i] = a_nxt[i];
      }
    }
#ifndef WS_DIAG_NO_STAGE
    if (s + 1 < n_stages) raw_store((s + 1) & 1);
    __syncthreads();
#endif
  }
#ifdef WS_DIAG_NO_STAGE
  __syncthreads();
#endif

  // ---- epilogue: Y = A^T M A with A^T = [[1,1,1,0],[0,1,-1,-1]], split over the two position halves.
  //      half 0 holds M rows 0, 1:  u0 = M0 + M1, u1 = M1;  half 1 holds M rows 2, 3:  u0 = M2, u1 = -(M2 + M3).
  //      y[a][0] = u_a[0] + u_a[1] + u_a[2], y[a][1] = u_a[1] - u_a[2] - u_a[3].  Wave `ph` finishes output row a = ph of every tile:
  //      it keeps its own y[ph] and receives the partner's y[ph] through LDS (the raw buffers are free after the last barrier). -------
  float mine[2][16], theirs[2][16];
#pragma unroll
  for (int e = 0; e < 16; e++) {
    float u0[4], u1[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float m0 = acc[j][e], m1 = acc[4 + j][e];
      const float sum = m0 + m1;
      u0[j] = ph ? m0 : sum;
      u1[j] = ph ? -sum : m1;
    }
    const float y00 = u0[0] + u0[1] + u0[2], y01 = u0[1] - u0[2] - u0[3];
    const float y10 = u1[0] + u1[1] + u1[2], y11 = u1[1] - u1[2] - u1[3];
    mine[0][e] = ph ? y10 : y00;
    mine[1][e] = ph ? y11 : y01;
    theirs[0][e] = ph ? y00 : y10;
    theirs[1][e] = ph ? y01 : y11;
  }
  float *xw = lds + ((cb * 2 + ph) * 32) * 64 + lane;
  const float *xr = lds + ((cb * 2 + (ph ^ 1)) * 32) * 64 + lane;
#pragma unroll
  for (int e = 0; e < 16; e++) {
    xw[e * 64] = theirs[0][e];
    xw[(16 + e) * 64] = theirs[1][e];
  }
  __syncthreads();
  const int n = nt * C::WBN + cb * 32 + r;
  if (n < p.cout) {
    const float bias = p.bias[n];
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const int oy = oy0 + 2 * (e >> 2) + ph;
      const int ox = ox0 + 2 * ((e & 3) + 4 * h);
      float v0 = mine[0][e] + xr[e * 64] + bias;
      float v1 = mine[1][e] + xr[(16 + e) * 64] + bias;
      if (p.relu) {
        v0 = fmaxf(v0, 0.f);
        v1 = fmaxf(v1, 0.f);
      }
      if (oy < p.h) {
        float *o = p.out + ((long long)(b * p.h + oy) * p.w + ox) * p.ld_out + n;
        if (ox < p.w) o[0] = v0;
        if (ox + 1 < p.w) o[p.ld_out] = v1;
      }
    }
  }
}

template <int CHUNK>
int launch_ws(const pcp_conv3x3_t *d, const float *in, const float *u, const float *bias, float *out, hipStream_t st) {
  using C = WsCfg<CHUNK>;
  WsParams p;
  p.in = in; p.u = u; p.bias = bias; p.out = out;
  p.batch = d->batch; p.h = d->in_h; p.w = d->in_w;
  p.cin = d->cin; p.cout = d->cout; p.cout_pad = d->cout_pad;
  p.ld_in = d->ld_in; p.ld_out = d->ld_out; p.relu = d->relu;
  p.tiles_x = (d->in_w + 15) / 16;
  p.tiles_y = (d->in_h + C::OUT_H - 1) / C::OUT_H;
  p.n_spatial = d->batch * p.tiles_x * p.tiles_y;
  long long blocks = (long long)p.n_spatial * (d->cout_pad / C::WBN);
  if (blocks <= 0 || blocks > 0x7fffffffLL) return PCP_ERR_ARG;
  hipLaunchKernelGGL((k_wino_ws<CHUNK>), dim3((unsigned)blocks), dim3(WS_THREADS), 0, st, p);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

}  // namespace

extern "C" int pcp_conv3x3_winograd_ws_supported(const pcp_conv3x3_t *d) {
  if (!d || d->stride != 1) return 0;
  if (d->cin <= 0 || d->cin % 32 != 0 || d->cout <= 0 || d->cout_pad < d->cout || d->cout_pad % 64 != 0) return 0;
  if (d->ld_in % 4 != 0 || d->batch <= 0 || d->in_h <= 0 || d->in_w <= 0) return 0;
  return 1;
}

extern "C" int pcp_conv3x3_winograd_ws(const pcp_conv3x3_t *d, const float *in, const float *u_packed, const float *bias, float *out,
                                       void *stream_) {
  if (!d || !in || !u_packed || !bias || !out) return PCP_ERR_ARG;
  if (d->stride != 1) return PCP_ERR_UNSUPPORTED;
  if (!pcp_conv3x3_winograd_ws_supported(d)) return PCP_ERR_ARG;
  if ((((uintptr_t)in) & 15) || (((uintptr_t)u_packed) & 15)) return PCP_ERR_ARG;
  return launch_ws<32>(d, in, u_packed, bias, out, (hipStream_t)stream_);
}

extern "C" int pcp_conv3x3_winograd_ws_plan(const pcp_conv3x3_t *d, int32_t *variant, double *executed_flops) {
  if (!pcp_conv3x3_winograd_ws_supported(d)) return PCP_ERR_ARG;
  if (variant) *variant = 1;
  if (executed_flops) {
    // every workgroup multiplies [32 tiles x cin] x [cin x 64] at each of the 16 Winograd positions (padding tiles included)
    const double wgs = (double)d->batch * ((d->in_w + 15) / 16) * ((d->in_h + 7) / 8) * (d->cout_pad / 64);
    *executed_flops = wgs * 2.0 * 16.0 * 32.0 * d->cin * 64.0;
  }
  return PCP_OK;
}
