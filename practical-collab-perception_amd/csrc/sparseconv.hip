// a5 + a6 (first layer) -- the stride-2 3x3 convolution that opens the BEV backbone (base_bev_backbone.py:36-44: ZeroPad2d(1),
// Conv2d(64, C, 3, stride 2), BN, ReLU) computed from the PILLAR LIST instead of the dense canvas.
//
// The canvas PointPillarScatter builds (pointpillar_scatter.py:14-37) is zero except at the P pillars (20 % of the 512 x 512 cells for a
// 60 k-point cloud, less for real sweeps).  The dense kernel reads 67 MB per frame and multiplies 80 % zeros; here a workgroup owns
// 8 x 16 output pixels and, tap by tap in a fixed order,
//   looks the 128 input cells of the tap up in the pillariser's cell -> pillar-rank table (pcp_voxelize workspace),
//   compacts the occupied ones (ballot prefix, pixel order) into rows of at most 32,
//   gathers their 64-float pillar rows into an LDS A tile (four chunks in flight in a register ring) and multiplies
//   [32 x 64] x [64 x C] on v_mfma_f32_16x16x4_f32 (wave = 16-row half x 32-channel half over the whole K, the tap's weight fragments
//   from L2 straight into registers, two items ahead),
//   adds its final product block straight into the pixels' accumulators in LDS (one barrier per chunk).
// No atomics and a fixed summation order (taps ascending, K halves fixed): deterministic.  Work: 2.25 products per pillar instead of
// 9 per output pixel, i.e. ~4x fewer MFMAs at 20 % occupancy after padding the row tiles to 32; the dense canvas is neither read nor
// (when no caller asks for `spatial_features`) written.
#include "pcp_common.h"

namespace {

constexpr int SP_TH = 8, SP_TW = 16, SP_PIX = SP_TH * SP_TW;     // output pixels per workgroup
constexpr int SP_CIN = 64;
constexpr int SP_ALD = 68;                                        // padded A row: conflict-free ds_read_b128 groups
constexpr int SP_THREADS = 256;

struct SpParams {
  const float *pf;          // (P, 64) pillar features in pillar-rank order
  const int *cell_rank;     // [B * nx * ny]: merged id b*nx*ny + cx*ny + cy -> pillar rank, -1 = empty
  const float *w;           // [9][64 (cout)][64 (cin)], k contiguous
  const float *bias;        // [64]
  float *out;               // (B, ho, wo, ld_out)
  int batch, nx, ny, ho, wo, ld_out, cout, relu;
  int tiles_x, tiles_y;
};


// OUT_BF16: the output map stored as bf16 (the frozen teachers of the bf16 training loop, include/pcp_hip_mp.h); same arithmetic
template <bool OUT_BF16>
__global__ __launch_bounds__(SP_THREADS, 2) void k_sparse_conv_s2(SpParams p) {
  __shared__ __attribute__((aligned(16))) float acc[SP_PIX * 64];          // per-pixel accumulators
#ifdef SP_ROW_SPLIT
  __shared__ __attribute__((aligned(16))) float atile[2][32 * SP_ALD];      // gathered pillar rows of the chunk (double buffered)
#else
  __shared__ __attribute__((aligned(16))) float atile[4][32 * SP_ALD];      // gathered pillar rows of two stages x two items
#endif
  __shared__ int row_rank[9][SP_PIX];                                       // compacted (tap-wise) pillar ranks ...
  __shared__ unsigned char row_pix[9][SP_PIX];                              // ... and the output pixel each row belongs to
  __shared__ int cnt[9][2];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int sp = blockIdx.x;
  const int tile_x = sp % p.tiles_x;
  sp /= p.tiles_x;
  const int tile_y = sp % p.tiles_y;
  const int b = sp / p.tiles_y;
  const int oy0 = tile_y * SP_TH, ox0 = tile_x * SP_TW;
  const int plane = p.nx * p.ny;

  // ---- occupancy of the 9 x 128 (tap, pixel) input cells, compacted per tap in pixel order ------------------------------------------
  int rank[9];
  {
    const int py = tid >> 4, px = tid & 15;                                 // threads 0..127 <-> pixels
    const int oy = oy0 + py, ox = ox0 + px;
    const bool pix_ok = tid < SP_PIX && oy < p.ho && ox < p.wo;
#pragma unroll
    for (int t = 0; t < 9; t++) {
      const int iy = 2 * oy + t / 3 - 1, ix = 2 * ox + t % 3 - 1;
      rank[t] = -1;
      if (pix_ok && iy >= 0 && iy < p.ny && ix >= 0 && ix < p.nx) rank[t] = p.cell_rank[(long long)b * plane + ix * p.ny + iy];
    }
  }
  for (int i = tid; i < SP_PIX * 64; i += SP_THREADS) acc[i] = 0.f;
  int pre[9];
#pragma unroll
  for (int t = 0; t < 9; t++) {
    const unsigned long long bal = __ballot(rank[t] >= 0);
    pre[t] = __popcll(bal & ((1ULL << lane) - 1ULL));
    if (lane == 0 && wave < 2) cnt[t][wave] = __popcll(bal);
  }
  __syncthreads();
  int total = 0;
#pragma unroll
  for (int t = 0; t < 9; t++) {
    const int c0 = cnt[t][0], c1 = cnt[t][1];
    total += c0 + c1;
    if (tid < SP_PIX && rank[t] >= 0) {
      const int row = (wave == 1 ? c0 : 0) + pre[t];
      row_rank[t][row] = rank[t];
      row_pix[t][row] = (unsigned char)tid;
    }
  }
  __syncthreads();

#ifdef SP_ROW_SPLIT
  if (total > 0) {
    // wave = (16-row half of the chunk, 32-channel half of the outputs): its product block [16 x 32] over the whole K = 64 is final, so
    // it is added straight into the accumulators of its rows' pixels -- no partial sums, no result tile in LDS, ONE barrier per item
    // (in-kernel stamps of the first version, which split K over the waves: result tile write 430 + second barrier 150 +
    // scatter-add 1 450 of 4 300 cycles per item)
    const int rh = wave >> 1, nh = wave & 1;
    const int m16 = lane & 15, q4 = lane >> 4;
    const int g_row = tid >> 3, g_q = (tid & 7) * 2;                       // gather: row, first of two 16-byte columns
    // work items = (tap, 32-row chunk) in ascending order (thread 0 lists them; at most 9 taps x 4 chunks)
    __shared__ unsigned char item_tap[36], item_chunk[36];
    __shared__ int n_items_s;
    if (tid == 0) {
      int k = 0;
      for (int t = 0; t < 9; t++)
        for (int c0 = 0; c0 < cnt[t][0] + cnt[t][1]; c0 += 32) { item_tap[k] = (unsigned char)t; item_chunk[k] = (unsigned char)(c0 >> 5); k++; }
      n_items_s = k;
    }
    __syncthreads();
    const int n_items = n_items_s;
    // Register rings, statically indexed through the 4x unrolled loop: the pillar rows of four items and the weight fragments of two
    // items ahead are in flight (a dependent L2 / HBM access costs ~2 us, an item ~1 us).  All loads are UNCONDITIONAL with clamped
    // indices (rows past the chunk re-read a valid row and are zeroed when they are stored to LDS), so the number of outstanding
    // loads is static and hipcc's counted s_waitcnt waits for the oldest slot only.
    constexpr int DEPTH = 4;
    f32x4 g[DEPTH][2];
    f32x4 wq[2][2][4];                                   // [ring slot][16-channel subtile][k group j]: W[t][32 nh + 16 s + m16][16 j + 4 q4 ..]
    auto gather = [&](int k, f32x4 (&dst)[2]) {
      const int kc = min(k, n_items - 1);
      const int t = item_tap[kc], c0 = item_chunk[kc] * 32;
      const int row = min(c0 + g_row, cnt[t][0] + cnt[t][1] - 1);
      const float *src = p.pf + (long long)row_rank[t][row] * SP_CIN + g_q * 4;
      dst[0] = *reinterpret_cast<const f32x4 *>(src);
      dst[1] = *reinterpret_cast<const f32x4 *>(src + 4);
    };
    auto wload = [&](int k, f32x4 (&dst)[2][4]) {
      const int t = item_tap[min(k, n_items - 1)];
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++) {
        const float *wr = p.w + ((long long)(t * 64 + nh * 32 + s2 * 16 + m16)) * SP_CIN + 4 * q4;
#pragma unroll
        for (int j = 0; j < 4; j++) dst[s2][j] = *reinterpret_cast<const f32x4 *>(wr + 16 * j);
      }
    };
    wload(0, wq[0]);
    __builtin_amdgcn_sched_barrier(0);
    wload(1, wq[1]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < DEPTH; u++) {
      gather(u, g[u]);
      __builtin_amdgcn_sched_barrier(0);
    }
    // one pipeline stage = one item; ring slots are indexed by the compile-time stage number u
    auto stage = [&](int k, f32x4 (&gu)[2], f32x4 (&wu)[2][4], float *at) {
          const int t = item_tap[k], c0 = item_chunk[k] * 32;
      const int rows = min(32, cnt[t][0] + cnt[t][1] - c0);
      const bool live = g_row < rows;
      const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4 *>(at + g_row * SP_ALD + g_q * 4) = live ? gu[0] : zero4;
      *reinterpret_cast<f32x4 *>(at + g_row * SP_ALD + g_q * 4 + 4) = live ? gu[1] : zero4;
      __builtin_amdgcn_sched_barrier(0);
      gather(k + DEPTH, gu);
      __builtin_amdgcn_sched_barrier(0);
      // pixels of this lane's four rows (16 rh + 4 q4 + i): four consecutive bytes of the compaction table, one LDS read
      const unsigned pix4 = *reinterpret_cast<const unsigned *>(&row_pix[t][min(c0 + rh * 16 + q4 * 4, SP_PIX - 4)]);
      __syncthreads();
      // [16 rows x 64] x [64 x 32] on v_mfma_f32_16x16x4_f32; MFMA (j, kk) multiplies k = 16 j + 4 q4 + kk
      typedef float f32x4c __attribute__((ext_vector_type(4)));
      f32x4c c0v = {0.f, 0.f, 0.f, 0.f}, c1v = {0.f, 0.f, 0.f, 0.f};
      const float *xa = at + (rh * 16 + m16) * SP_ALD + 4 * q4;
      f32x4 a[4];
#pragma unroll
      for (int j = 0; j < 4; j++) a[j] = *reinterpret_cast<const f32x4 *>(xa + 16 * j);
      // the accumulators' old values do not depend on the products: requested BEFORE the MFMAs so their LDS round trip (and that of
      // the pixel look-up above) runs under the matrix work instead of after it (in-kernel stamps: 1 600 + 800 of 4 700 cycles per item)
      int pixs[4];
      float old0[4], old1[4];
      const int col = nh * 32 + m16;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int row = rh * 16 + q4 * 4 + i;
        pixs[i] = row < rows ? (int)((pix4 >> (8 * i)) & 255u) : -1;
        const int pa = pixs[i] >= 0 ? pixs[i] : 0;
        old0[i] = acc[pa * 64 + col];
        old1[i] = acc[pa * 64 + col + 16];
      }
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          c0v = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][kk], wu[0][j][kk], c0v, 0, 0, 0);
          c1v = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][kk], wu[1][j][kk], c1v, 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
      wload(k + 2, wu);                       // this slot's fragments are consumed: refill it for the item after next
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; i++)
        if (pixs[i] >= 0) {
          acc[pixs[i] * 64 + col] = old0[i] + c0v[i];
          acc[pixs[i] * 64 + col + 16] = old1[i] + c1v[i];
        }
    };
    // Main loop: groups of DEPTH items with NO conditional stage, so the number of loads in flight at every wait is static (a guard
    // around a stage makes it path dependent and hipcc falls back to s_waitcnt vmcnt(0): every item then waits for the gather it has
    // just issued -- the reason prefetch depth, resident weights and occupancy all measured the same 127-133 us).
    int k0 = 0;
    for (; k0 + DEPTH <= n_items; k0 += DEPTH) {
      stage(k0 + 0, g[0], wq[0], atile[0]);
      stage(k0 + 1, g[1], wq[1], atile[1]);
      stage(k0 + 2, g[2], wq[0], atile[0]);
      stage(k0 + 3, g[3], wq[1], atile[1]);
    }
    // tail: the remaining (< DEPTH) items already sit in ring slots 0 ..
    if (k0 + 0 < n_items) stage(k0 + 0, g[0], wq[0], atile[0]);
    if (k0 + 1 < n_items) stage(k0 + 1, g[1], wq[1], atile[1]);
    if (k0 + 2 < n_items) stage(k0 + 2, g[2], wq[0], atile[0]);
    __syncthreads();
  }

#else
  if (total > 0) {
    // Round 3: wave w owns output channels [16 w, 16 w + 16) for ALL rows of an item, so an accumulator word (pixel, channel) is only ever
    // touched by one wave: the read-modify-write of the per-pixel accumulators needs no cross-wave ordering and TWO items (taps) share one
    // barrier -- 64 instead of 32 MFMAs per wave between barriers, half the barriers (the row-split form sat at 22 - 25 % matrix-pipe busy:
    // ~4 300 cycles per item around 1 024 cycles of MFMA).  Same products in the same order per (row, channel): results are bit-identical.
    const int m16 = lane & 15, q4 = lane >> 4;
    const int g_row = tid >> 3, g_q = (tid & 7) * 2;                       // gather: row, first of two 16-byte columns
    __shared__ unsigned char item_tap[36], item_chunk[36];
    __shared__ int n_items_s;
    if (tid == 0) {
      int k = 0;
      for (int t = 0; t < 9; t++)
        for (int c0 = 0; c0 < cnt[t][0] + cnt[t][1]; c0 += 32) { item_tap[k] = (unsigned char)t; item_chunk[k] = (unsigned char)(c0 >> 5); k++; }
      n_items_s = k;
    }
    __syncthreads();
    const int n_items = n_items_s;
    // register rings over FOUR items (two stages of two): gathered pillar rows and this wave's 16-channel weight fragments.  All loads are
    // unconditional with clamped indices (static count of outstanding loads: counted s_waitcnt).
    f32x4 g[4][2];
    f32x4 wq[4][4];                                      // [ring slot][k group j]: W[t][16 wave + m16][16 j + 4 q4 ..]
    auto gather = [&](int k, f32x4 (&dst)[2]) {
      const int kc = min(k, n_items - 1);
      const int t = item_tap[kc], c0 = item_chunk[kc] * 32;
      const int row = min(c0 + g_row, cnt[t][0] + cnt[t][1] - 1);
      const float *src = p.pf + (long long)row_rank[t][row] * SP_CIN + g_q * 4;
      dst[0] = *reinterpret_cast<const f32x4 *>(src);
      dst[1] = *reinterpret_cast<const f32x4 *>(src + 4);
    };
    auto wload = [&](int k, f32x4 (&dst)[4]) {
      const int t = item_tap[min(k, n_items - 1)];
      const float *wr = p.w + ((long long)(t * 64 + wave * 16 + m16)) * SP_CIN + 4 * q4;
#pragma unroll
      for (int j = 0; j < 4; j++) dst[j] = *reinterpret_cast<const f32x4 *>(wr + 16 * j);
    };
#pragma unroll
    for (int u = 0; u < 4; u++) {
      wload(u, wq[u]);
      gather(u, g[u]);
      __builtin_amdgcn_sched_barrier(0);
    }
    typedef float f32x4c __attribute__((ext_vector_type(4)));
    const int col = wave * 16 + m16;
    // one stage = items k and k + 1 (k + 1 may lie past the end: its rows are stored as zeros and its updates masked off)
    auto stage = [&](int k, f32x4 (&g0)[2], f32x4 (&g1)[2], f32x4 (&w0)[4], f32x4 (&w1)[4], float *at0, float *at1) {
      const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
      int rows_i[2], tap_i[2], c0_i[2];
#pragma unroll
      for (int it = 0; it < 2; it++) {
        const int kk = min(k + it, n_items - 1);
        tap_i[it] = item_tap[kk];
        c0_i[it] = item_chunk[kk] * 32;
        rows_i[it] = (k + it < n_items) ? min(32, cnt[tap_i[it]][0] + cnt[tap_i[it]][1] - c0_i[it]) : 0;
      }
      *reinterpret_cast<f32x4 *>(at0 + g_row * SP_ALD + g_q * 4) = g_row < rows_i[0] ? g0[0] : zero4;
      *reinterpret_cast<f32x4 *>(at0 + g_row * SP_ALD + g_q * 4 + 4) = g_row < rows_i[0] ? g0[1] : zero4;
      *reinterpret_cast<f32x4 *>(at1 + g_row * SP_ALD + g_q * 4) = g_row < rows_i[1] ? g1[0] : zero4;
      *reinterpret_cast<f32x4 *>(at1 + g_row * SP_ALD + g_q * 4 + 4) = g_row < rows_i[1] ? g1[1] : zero4;
      __builtin_amdgcn_sched_barrier(0);
      gather(k + 4, g0);
      gather(k + 5, g1);
      __builtin_amdgcn_sched_barrier(0);
      // pixels of this lane's rows (16 rtile + 4 q4 + i): four consecutive bytes of the compaction table per row tile
      unsigned pix4[2][2];
#pragma unroll
      for (int it = 0; it < 2; it++)
#pragma unroll
        for (int rtile = 0; rtile < 2; rtile++)
          pix4[it][rtile] = *reinterpret_cast<const unsigned *>(&row_pix[tap_i[it]][min(c0_i[it] + rtile * 16 + q4 * 4, SP_PIX - 4)]);
      __syncthreads();
      // this lane's accumulator words of both items: pixel (or 0 for a row past the chunk: read, never written) and the old values of item
      // k, requested BEFORE the MFMAs so that their LDS round trip runs under the matrix work (a `+=` per word would serialise 16 dependent
      // round trips: the compiler cannot prove the words distinct)
      int pixs[2][2][4];
      float old0[2][4];
#pragma unroll
      for (int it = 0; it < 2; it++)
#pragma unroll
        for (int rtile = 0; rtile < 2; rtile++)
#pragma unroll
          for (int i = 0; i < 4; i++)
            pixs[it][rtile][i] = (rtile * 16 + q4 * 4 + i < rows_i[it]) ? (int)((pix4[it][rtile] >> (8 * i)) & 255u) : -1;
#pragma unroll
      for (int rtile = 0; rtile < 2; rtile++)
#pragma unroll
        for (int i = 0; i < 4; i++) old0[rtile][i] = acc[max(pixs[0][rtile][i], 0) * 64 + col];
      f32x4c cv[2][2];
#pragma unroll
      for (int it = 0; it < 2; it++) {
        const float *at = it == 0 ? at0 : at1;
#pragma unroll
        for (int rtile = 0; rtile < 2; rtile++) {
          f32x4c c = {0.f, 0.f, 0.f, 0.f};
          // a row tile without rows (the second tile of an item of at most 16 rows -- the thin outskirts of a LiDAR-like frame put one to
          // five rows into a tap -- and both tiles of the item past an odd count) is passed over: its products would multiply zero rows
          // and be added nowhere.  Wave-uniform branch.
          if (rows_i[it] > 16 * rtile) {
            f32x4 a[4];
#pragma unroll
            for (int j = 0; j < 4; j++) a[j] = *reinterpret_cast<const f32x4 *>(at + (rtile * 16 + m16) * SP_ALD + 16 * j + 4 * q4);
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
              for (int kk = 0; kk < 4; kk++) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][kk], (it == 0 ? w0 : w1)[j][kk], c, 0, 0, 0);
          }
          cv[it][rtile] = c;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      wload(k + 4, w0);                          // both slots' fragments are consumed: refill them for the stage after next
      wload(k + 5, w1);
      __builtin_amdgcn_sched_barrier(0);
      // accumulate, item k first (a pixel may receive both taps: the in-order LDS operations of one wave keep the order)
#pragma unroll
      for (int rtile = 0; rtile < 2; rtile++)
#pragma unroll
        for (int i = 0; i < 4; i++)
          if (pixs[0][rtile][i] >= 0) acc[pixs[0][rtile][i] * 64 + col] = old0[rtile][i] + cv[0][rtile][i];
      float old1[2][4];
#pragma unroll
      for (int rtile = 0; rtile < 2; rtile++)
#pragma unroll
        for (int i = 0; i < 4; i++) old1[rtile][i] = acc[max(pixs[1][rtile][i], 0) * 64 + col];
#pragma unroll
      for (int rtile = 0; rtile < 2; rtile++)
#pragma unroll
        for (int i = 0; i < 4; i++)
          if (pixs[1][rtile][i] >= 0) acc[pixs[1][rtile][i] * 64 + col] = old1[rtile][i] + cv[1][rtile][i];
    };
    int k0 = 0;
    for (; k0 + 4 <= n_items; k0 += 4) {
      stage(k0 + 0, g[0], g[1], wq[0], wq[1], atile[0], atile[1]);
      stage(k0 + 2, g[2], g[3], wq[2], wq[3], atile[2], atile[3]);
    }
    if (k0 + 0 < n_items) stage(k0 + 0, g[0], g[1], wq[0], wq[1], atile[0], atile[1]);
    if (k0 + 2 < n_items) stage(k0 + 2, g[2], g[3], wq[2], wq[3], atile[2], atile[3]);
    __syncthreads();
  }
#endif

  // ---- epilogue: bias + ReLU, one 256-byte row per pixel --------------------------------------------------------------------------
  {
    const int n4 = (tid & 15) * 4;
    const f32x4 bv = *reinterpret_cast<const f32x4 *>(p.bias + n4);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int pix = (tid >> 4) + 16 * i;
      const int oy = oy0 + (pix >> 4), ox = ox0 + (pix & 15);
      if (oy < p.ho && ox < p.wo && n4 < p.cout) {
        f32x4 v = *reinterpret_cast<const f32x4 *>(acc + pix * 64 + n4) + bv;
        if (p.relu) {
          v.x = fmaxf(v.x, 0.f);
          v.y = fmaxf(v.y, 0.f);
          v.z = fmaxf(v.z, 0.f);
          v.w = fmaxf(v.w, 0.f);
        }
        const long long o = ((long long)(b * p.ho + oy) * p.wo + ox) * p.ld_out + n4;
        if (OUT_BF16) {
          typedef __bf16 b4 __attribute__((ext_vector_type(4)));
          b4 q;
          q[0] = (__bf16)v.x; q[1] = (__bf16)v.y; q[2] = (__bf16)v.z; q[3] = (__bf16)v.w;
          *reinterpret_cast<b4 *>(reinterpret_cast<__bf16 *>(p.out) + o) = q;
        } else {
          *reinterpret_cast<f32x4 *>(p.out + o) = v;
        }
      }
    }
  }
}

}  // namespace

static int sparse_conv_impl(const float *pillar_features, const pcp_grid_t *grid, const void *vox_workspace, int64_t n, const float *w_packed,
                            const float *bias, int32_t cout, int32_t relu, void *out, int out_bf16, int32_t ld_out, void *stream_) {
  if (!pillar_features || !grid || !vox_workspace || !w_packed || !bias || !out || n < 0) return PCP_ERR_ARG;
  if (cout <= 0 || cout > 64 || cout % 4 != 0 || ld_out % 4 != 0 || ld_out < cout) return PCP_ERR_UNSUPPORTED;
  if ((((uintptr_t)pillar_features) & 15) || (((uintptr_t)w_packed) & 15) || (((uintptr_t)bias) & 15) || (((uintptr_t)out) & (out_bf16 ? 7 : 15)))
    return PCP_ERR_ARG;
  if (grid->batch_size <= 0 || grid->nx <= 0 || grid->ny <= 0) return PCP_ERR_ARG;
  const int64_t cells = (int64_t)grid->batch_size * grid->nx * grid->ny;
  VoxLayout L = pcp_vox_layout(cells, n);
  SpParams p;
  p.pf = pillar_features;
  p.cell_rank = reinterpret_cast<const int *>(reinterpret_cast<const char *>(vox_workspace) + L.cell_rank);
  p.w = w_packed; p.bias = bias; p.out = (float *)out;
  p.batch = grid->batch_size; p.nx = grid->nx; p.ny = grid->ny;
  p.ho = (grid->ny - 1) / 2 + 1;
  p.wo = (grid->nx - 1) / 2 + 1;
  p.ld_out = ld_out; p.cout = cout; p.relu = relu;
  p.tiles_x = (p.wo + SP_TW - 1) / SP_TW;
  p.tiles_y = (p.ho + SP_TH - 1) / SP_TH;
  const long long blocks = (long long)p.batch * p.tiles_x * p.tiles_y;
  if (blocks > 0x7fffffffLL) return PCP_ERR_ARG;
  if (out_bf16) hipLaunchKernelGGL(k_sparse_conv_s2<true>, dim3((unsigned)blocks), dim3(SP_THREADS), 0, (hipStream_t)stream_, p);
  else hipLaunchKernelGGL(k_sparse_conv_s2<false>, dim3((unsigned)blocks), dim3(SP_THREADS), 0, (hipStream_t)stream_, p);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

extern "C" int pcp_sparse_conv3x3_s2(const float *pillar_features, const pcp_grid_t *grid, const void *vox_workspace, int64_t n,
                                     const float *w_packed, const float *bias, int32_t cout, int32_t relu, float *out, int32_t ld_out,
                                     void *stream_) {
  return sparse_conv_impl(pillar_features, grid, vox_workspace, n, w_packed, bias, cout, relu, out, 0, ld_out, stream_);
}

// include/pcp_hip_mp.h: the same launch with a storage type for the output map
extern "C" int pcp_mp_sparse_conv3x3_s2(const float *pillar_features, const pcp_grid_t *grid, const void *vox_workspace, int64_t n,
                                        const float *w_packed, const float *bias, int32_t cout, int32_t relu, void *out, int32_t out_dtype,
                                        int32_t ld_out, void *stream_) {
  if (out_dtype != 0 && out_dtype != 1) return PCP_ERR_ARG;
  return sparse_conv_impl(pillar_features, grid, vox_workspace, n, w_packed, bias, cout, relu, out, out_dtype, ld_out, stream_);
}
