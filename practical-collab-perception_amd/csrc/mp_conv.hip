// Mixed-precision 3x3 convolution for the bf16 training loop of config 5 (include/pcp_hip_mp.h): NHWC activations stored as bf16,
// products on v_mfma_f32_32x32x16_bf16, fp32 accumulation, bias (+ ReLU) in the epilogue.  Forward of the trainable branch, its data
// gradient (flipped / transposed weight form) and the three frozen BEV makers (BatchNorm folded into the bf16 weights + fp32 bias).
// Reference layers: pcdet/models/backbones_2d/base_bev_backbone.py:30-69, dense_heads/center_head.py:24-29,75-82,
// bev_layers/v2x_fusion_disco.py:51-63 (nn.Conv2d 3x3 [+ BatchNorm2d + ReLU]) as they run under torch.cuda.amp.autocast.
//
// k_mp_conv3x3_s1 -- stride 1, bf16 input, cin % 32 == 0: PERSISTENT workgroups (one per CU), nothing but MFMA operands in registers.
//   item     = TH x 32 output pixels x 64 output channels (TH = 16: 8 waves, TH = 8: 4 waves); a workgroup walks a contiguous run of items
//   stage    = one 32-channel slice of one item: the (TH+2) x 34 pixel patch (64 B per pixel) and the slice's 9 x 32 x 64 weights
//              (36 KB), both copied global -> LDS by `buffer_load_dwordx4 ... lds` (no registers, no VALU; halo pixels outside the map are
//              out-of-range buffer offsets = zeros), two stages in LDS: the copy of stage q+1 (which may belong to the NEXT item) runs
//              under the MFMAs of stage q; one barrier per stage
//   products = weights are the A operand (rows = 32 output channels), pixels the B operand (columns = 32 consecutive pixels of one
//              output row): a lane ends up with 4 consecutive channels of one pixel per accumulator quad -> v_permlane32_swap pairs ->
//              16-byte stores (cdna_hip_programming.md T21).  Wave = 2 output rows x 64 channels: 2 + 2 ds_read_b128 per 4 MFMAs.
//   LDS      = pixel records XOR-swizzled by column ((px >> 2) & 3 on the 16-byte chunk index): the 16 lanes of a ds_read_b128 group
//              (32 consecutive pixels of a row, any tap) hit 16 different bank groups; the swizzle is applied on the SOURCE address of
//              the copy (the LDS side of an LDS-DMA is lane-linear); weight fragments are lane-linear 16-byte rows.
// k_mp_conv3x3_gen -- everything else (stride 2, fp32 input, cin % 32 == 16): register-staged, converts while staging.
#include "pcp_common.h"
#include "../../include/pcp_hip_mp.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef unsigned u32x4 __attribute__((__vector_size__(4 * sizeof(unsigned))));

constexpr int MC_TW = 32;                       // output pixels per tile row (= one MFMA column block)
constexpr int MC_CK = 32;                       // input channels per stage
constexpr int MC_BN = 64;                       // output channels per item
constexpr int MC_WCHUNK = 9 * 2 * 64 * 8;       // bf16 elements of one (16-channel k step, 64-channel block) of the packed weights
constexpr int MC_WBYTES = 2 * MC_WCHUNK * 2;    // weight bytes of one stage (two k steps)
constexpr unsigned MC_OOB = 0x80000000u;
constexpr int MC_MAX_COUT = 512;                // bias row kept in LDS by the fast kernel

struct McParams {
  const void *in;
  const __bf16 *w;        // [cin/16][cout_pad/64][9][2][64][8]
  const float *bias;
  void *out;
  int batch, h, w_, cin, cout, cout_pad, ld_in, ld_out, relu;
  int tiles_y, tiles_x, n_nb, n_items, n_slices;
  unsigned in_bytes, w_bytes, out_bytes;
  int diag;               // timing-only builds (PCP_MP_DIAG): 4 = no copies after the first stage (wrong results)
};

__device__ __forceinline__ int mc_xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

#ifdef MC_STAMP
// diagnostic build (csrc/build_variant.sh stamp "-DMC_STAMP", tools/stamp_mc.py): s_memtime stamps of wave 0 of every workgroup, kept in
// memory nothing else reads
__device__ unsigned long long mc_stamps[1024 * 16];
#define MC_T(i)                                                                                         \
  do {                                                                                                  \
    if (tid == 0 && blockIdx.x < 1024) mc_stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define MC_T(i) do { } while (0)
#endif

template <int TH, bool OUT_BF16>
__global__ __launch_bounds__(TH * 32, TH == 16 ? 2 : 1) void k_mp_conv3x3_s1(McParams p) {
  constexpr int NTHR = TH * 32, NWAVE = TH / 2;
  constexpr int PH = TH + 2, PW = MC_TW + 2, NPIX = PH * PW;
  constexpr int PB = (NPIX * 64 + 1023) / 1024 * 1024;            // patch bytes of a stage, whole 1-KB wave instructions
  constexpr int PBI = PB / 1024;                                  // patch wave-instructions per stage
  constexpr int WI = PBI + MC_WBYTES / 1024;                      // + 36 weight wave-instructions
  constexpr int NLD = (WI + NWAVE - 1) / NWAVE;                   // copy instructions per wave and stage (last may be idle)
  constexpr int STAGE = PB + MC_WBYTES;
  constexpr int NST = OUT_BF16 ? 8 : 16;                         // store instructions per wave and item (static: masked lanes store out of range)
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE + MC_MAX_COUT * 4];

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, c32 = lane & 31, h = lane >> 5;
  MC_T(0);

  // ---- this workgroup's run of items (XCD-contiguous: neighbours in the run share patch halos and weights in one L2) --------------------
  const int nwg = gridDim.x;
  const int lid = mc_xcd_remap(blockIdx.x, nwg);
  const int per = p.n_items / nwg, rem = p.n_items % nwg;
  const int it_begin = lid * per + min(lid, rem);
  const int it_end = it_begin + per + (lid < rem ? 1 : 0);
  if (it_begin >= it_end) return;
  const int n_stages = (it_end - it_begin) * p.n_slices;

  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(p.w), 0, p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);

  // ---- copy plan of this wave: instruction k moves wave-instruction i = wave + k * NWAVE of the stage ------------------------------------
  // patch instruction: chunk q = i * 64 + lane of the LDS image = pixel pp = q >> 2, LDS chunk c' = q & 3, which holds channel chunk
  // c = c' ^ ((px >> 2) & 3) of that pixel
  int ppy[NLD], ppx[NLD], pch[NLD];
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int i = wave + k * NWAVE;
    const int q = i * 64 + lane;
    const int pp = q >> 2;
    ppy[k] = pp / PW;
    ppx[k] = pp - ppy[k] * PW;
    pch[k] = ((q & 3) ^ ((ppx[k] >> 2) & 3)) * 16;
    if (pp >= NPIX) ppy[k] = -10000;                              // padding chunks of the last patch instruction: zeros
  }
  unsigned poff[NLD];
  int ld_nb = 0;                                                  // 64-channel block of the item whose stage is being COPIED
  // item -> (64-channel block fastest, tile column, tile row, frame): decoded ONCE (integer divisions by run-time values cost ~50
  // instructions each on this ISA), then advanced as counters
  struct Coords { int nb, tx, ty, b; };
  auto decode = [&](int it) {
    Coords c;
    c.nb = it % p.n_nb;
    int sp = it / p.n_nb;
    c.tx = sp % p.tiles_x;
    sp /= p.tiles_x;
    c.ty = sp % p.tiles_y;
    c.b = sp / p.tiles_y;
    return c;
  };
  auto advance = [&](Coords &c) {
    if (++c.nb == p.n_nb) {
      c.nb = 0;
      if (++c.tx == p.tiles_x) {
        c.tx = 0;
        if (++c.ty == p.tiles_y) { c.ty = 0; ++c.b; }
      }
    }
  };
  Coords ld_c = decode(it_begin), cur_c = ld_c;                   // the item being COPIED / being COMPUTED
  auto plan_item = [&]() {
    const int oy0 = ld_c.ty * TH, ox0 = ld_c.tx * MC_TW;
    ld_nb = ld_c.nb;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int iy = oy0 - 1 + ppy[k], ix = ox0 - 1 + ppx[k];
      poff[k] = MC_OOB;
      if (iy >= 0 && iy < p.h && ix >= 0 && ix < p.w_) poff[k] = (unsigned)(((ld_c.b * p.h + iy) * p.w_ + ix) * p.ld_in * 2 + pch[k]);
    }
  };
  // copy instruction k of this wave for (slice, buffer); the instructions of a stage are issued ONE PER MULTIPLY STEP inside the product
  // loop of the previous stage: issued back to back they block the wave for ~200 cycles each (in-kernel stamps, tools/stamp_mc.py: the
  // CU's vector-memory path takes 64 B/clk, a stage is 76 KB), i.e. ~2 000 cycles per stage in which the wave multiplies nothing
  auto issue_one = [&](int k, int slice, int buf) {
    unsigned char *base = lds + buf * STAGE;
    const int i = wave + k * NWAVE;                                // wave-uniform
    if (i < PBI) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_void *)(base + i * 1024), 16, (int)poff[k], slice * (MC_CK * 2), 0, 0);
    } else if (i < WI) {
      const int j = i - PBI;                                       // 0 .. 35: 18 wave-instructions per 16-channel k step
      const int ks = j >= 18 ? 1 : 0;
      const int soff = ((slice * 2 + ks) * p.n_nb + ld_nb) * (MC_WCHUNK * 2) + (j - ks * 18) * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void *)(base + i * 1024), 16, lane * 16, soff, 0, 0);
    }
  };
  static_assert(NLD <= 18, "one copy instruction per multiply step");

  // ---- fragment addresses --------------------------------------------------------------------------------------------------------------------
  // B (pixels): lane (c32, h) of output row r, tap (ky, kx), k step ks reads the 16 bytes of channels ks*16 + h*8 .. +7 of patch pixel
  // (r + ky, c32 + kx): chunk (ks*2 + h) ^ ((px >> 2) & 3)
  int bcol[3][2];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int px = c32 + kx;
      bcol[kx][ks] = px * 64 + (((ks * 2 + h) ^ ((px >> 2) & 3)) * 16);
    }
  const int brow0 = (wave * 2) * (PW * 64);
  // A (weights): [ks][tap][h][64][8] -> lane reads 16 bytes at ((ks*9 + tap)*2 + h)*1024 + (nt*32 + c32)*16
  const int aoff = h * 1024 + c32 * 16;

  f32x16 acc[2][2];                                               // [nt (32 output channels)][mt (output row)]
  float *bias_lds = reinterpret_cast<float *>(lds + 2 * STAGE);

  for (int i = tid; i < p.cout_pad; i += NTHR) bias_lds[i] = p.bias[i];      // before the first copy is issued: no plain load inside the loop
  plan_item();
#pragma unroll
  for (int k = 0; k < NLD; ++k) issue_one(k, 0, 0);
  int it = it_begin, slice = 0;
  bool stores_behind = false;                                     // the previous stage ended an item: its NST stores are younger than the copy
  for (int q = 0; q < n_stages; ++q) {
    const int buf = q & 1;
    if (stores_behind) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (q < 4) MC_T(1 + 3 * q);
    // next stage: same item next slice, or first slice of the next item
    int nslice = slice + 1, nit = it;
    if (nslice == p.n_slices) { nslice = 0; nit = it + 1; }
    const bool copy_next = q + 1 < n_stages && !(p.diag & 4);
    if (copy_next && nslice == 0) {
      advance(ld_c);
      plan_item();
    }
    if (slice == 0) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[nt][mt][e] = 0.f;
    }
    if (q < 4) MC_T(2 + 3 * q);
    // ---- the stage's products: 2 k steps x 9 taps x (2 x 2) MFMAs per wave ---------------------------------------------------------------
    {
      const unsigned char *pb = lds + buf * STAGE + brow0;
      const unsigned char *wb = lds + buf * STAGE + PB + aoff;
      bf16x8 a[2][2], b[2][2];
      auto load = [&](int step, bf16x8 (&af)[2], bf16x8 (&bf)[2]) {
        const int ks = step / 9, tap = step % 9, ky = tap / 3, kx = tap % 3;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) af[nt] = *reinterpret_cast<const bf16x8 *>(wb + ((ks * 9 + tap) * 2) * 1024 + nt * 512);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) bf[mt] = *reinterpret_cast<const bf16x8 *>(pb + (mt + ky) * (PW * 64) + bcol[kx][ks]);
      };
      load(0, a[0], b[0]);
#pragma unroll
      for (int step = 0; step < 18; ++step) {
        const int cur = step & 1;
        // the four fragment reads of step + 1 are ISSUED before the four MFMAs of this step (fenced: hipcc otherwise sinks every read to
        // just in front of its first use, and the single wave of a SIMD then waits out the LDS latency once per step)
        if (step + 1 < 18) load(step + 1, a[cur ^ 1], b[cur ^ 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][nt], b[cur][mt], acc[nt][mt], 0, 0, 0);
        // copies early in the stage: two per step over the first half of the copy list's steps, so that the last one is issued ~13 steps
        // (about a memory latency) before the wait at the top of the next stage
        if (copy_next) {
          if (2 * step < NLD) issue_one(2 * step, nslice, buf ^ 1);
          if (2 * step + 1 < NLD) issue_one(2 * step + 1, nslice, buf ^ 1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (q < 4) MC_T(3 + 3 * q);
    stores_behind = false;
    if (slice == p.n_slices - 1) {
      // ---- epilogue: bias (+ ReLU), 16-byte stores; lane (c32, h) holds channels nt*32 + 8g + 4h + (0..3) of pixel (row mt, column c32) ------
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int oy = cur_c.ty * TH + wave * 2 + mt, ox = cur_c.tx * MC_TW + c32;
        const bool pix_ok = oy < p.h && ox < p.w_;
        const unsigned pix_off = (unsigned)(((cur_c.b * p.h + oy) * p.w_ + ox) * p.ld_out);   // elements
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const int ch0 = cur_c.nb * MC_BN + nt * 32;
          float v[16];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 bq = *reinterpret_cast<const f32x4 *>(bias_lds + ch0 + 8 * g + 4 * h);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              float t = acc[nt][mt][4 * g + i] + bq[i];
              if (p.relu) t = fmaxf(t, 0.f);
              v[4 * g + i] = t;
            }
          }
          if (OUT_BF16) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {                                  // channel groups g = 2j, 2j + 1 -> one 16-byte store
              bf16x4 lo, hi;
#pragma unroll
              for (int i = 0; i < 4; ++i) { lo[i] = (__bf16)v[8 * j + i]; hi[i] = (__bf16)v[8 * j + 4 + i]; }
              uint2 ga = __builtin_bit_cast(uint2, lo), gb = __builtin_bit_cast(uint2, hi);
              const auto s0 = __builtin_amdgcn_permlane32_swap(ga.x, gb.x, false, false);
              const auto s1 = __builtin_amdgcn_permlane32_swap(ga.y, gb.y, false, false);
              // lanes 0-31: [own g=2j | upper's g=2j] = channels 16j .. 16j+7; lanes 32-63: [lower's g=2j+1 | own g=2j+1] = 16j+8 .. 16j+15
              const uint4 o = make_uint4(s0[0], s1[0], s0[1], s1[1]);
              const int ch = ch0 + 16 * j + 8 * h;
              unsigned off = MC_OOB;
              if (pix_ok && ch < p.cout) off = (pix_off + ch) * 2;
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), out_rsrc, (int)off, 0, 0);
            }
          } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int ch = ch0 + 8 * g + 4 * h;
              unsigned off = MC_OOB;
              if (pix_ok && ch < p.cout) off = (pix_off + ch) * 4;
              const uint4 o = make_uint4(__float_as_uint(v[4 * g]), __float_as_uint(v[4 * g + 1]), __float_as_uint(v[4 * g + 2]), __float_as_uint(v[4 * g + 3]));
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), out_rsrc, (int)off, 0, 0);
            }
          }
        }
      }
      stores_behind = true;
    }
    slice = nslice;
    if (nit != it) {
      it = nit;
      advance(cur_c);
    }
    if (q == 3) MC_T(13);
  }
  MC_T(14);
#ifdef MC_STAMP
  if (tid == 0 && blockIdx.x < 1024) mc_stamps[blockIdx.x * 16 + 15] = (unsigned long long)n_stages;
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// general kernel: register-staged (round 3's bf16 kernel with a storage type per tensor): pixels are the A operand, a lane ends up with ONE
// output channel of 16 pixels.  Workgroup (4 waves) = 16x16 (stride 1) or 8x16 (stride 2) output pixels x 64 output channels; per
// 16-channel slice the patch with halo is rounded to bf16 while it is staged in LDS and re-read for all nine taps.
// ---------------------------------------------------------------------------------------------------------------------------------------
constexpr int MG_THREADS = 256;
constexpr int MG_TW = 16;
constexpr int MG_CK = 16;

template <typename T> struct In4;
template <> struct In4<float> {
  typedef f32x4 raw;
  static __device__ __forceinline__ raw ld(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
  static __device__ __forceinline__ bf16x4 cvt(const raw &v) {
    bf16x4 o;
    o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
    return o;
  }
};
template <> struct In4<__bf16> {
  typedef uint2 raw;
  static __device__ __forceinline__ raw ld(const __bf16 *p) { return *reinterpret_cast<const uint2 *>(p); }
  static __device__ __forceinline__ bf16x4 cvt(const raw &v) { return __builtin_bit_cast(bf16x4, v); }
};
template <typename T> __device__ __forceinline__ void mg_store(T *p, float v);
template <> __device__ __forceinline__ void mg_store<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void mg_store<__bf16>(__bf16 *p, float v) { *p = (__bf16)v; }

struct MgParams {
  const void *in;
  const __bf16 *w;
  const float *bias;
  void *out;
  int batch, in_h, in_w, out_h, out_w, cin, cout, cout_pad, ld_in, ld_out, relu;
  int tiles_y, tiles_x;
};

template <int S, int TH, typename IT, typename OT>
__global__ __launch_bounds__(MG_THREADS, 2) void k_mp_conv3x3_gen(MgParams p) {
  constexpr int PH = (TH - 1) * S + 3, PW = (MG_TW - 1) * S + 3;
  constexpr int NPIX = PH * PW;
  constexpr int MT = TH / 8;
  constexpr int NPL = (NPIX * 4 + MG_THREADS - 1) / MG_THREADS;
  constexpr int NWL = MC_WCHUNK / 8 / MG_THREADS;                 // 16-byte weight loads per thread (4.5 -> 5 with a guard)
  constexpr int NWLC = (MC_WCHUNK / 8 + MG_THREADS - 1) / MG_THREADS;
  __shared__ __attribute__((aligned(16))) __bf16 patch[2][NPIX][8];          // [k half][pixel][8]
  __shared__ __attribute__((aligned(16))) __bf16 wts[MC_WCHUNK];             // [tap][k half][64][8]
  (void)NWL;
  const IT *in = reinterpret_cast<const IT *>(p.in);
  OT *out = reinterpret_cast<OT *>(p.out);
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int n_ct = p.cout_pad / MC_BN;
  int lid = mc_xcd_remap(blockIdx.x, gridDim.x);
  const int ct = lid % n_ct;
  lid /= n_ct;
  const int tx = lid % p.tiles_x;
  lid /= p.tiles_x;
  const int ty = lid % p.tiles_y;
  const int b = lid / p.tiles_y;
  const int oy0 = ty * TH, ox0 = tx * MG_TW;
  const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;

  long long poff[NPL];
  int pdst[NPL];               // element offset inside the image; bit 30: halo pixel outside the map (zeros); -1: no item
#pragma unroll
  for (int u = 0; u < NPL; ++u) {
    const int idx = tid + u * MG_THREADS;
    const int pix = idx >> 2, q = idx & 3;
    const int iy = iy0 + pix / PW, ix = ix0 + pix % PW;
    const bool inside = idx < NPIX * 4 && iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w;
    poff[u] = inside ? (((long long)b * p.in_h + iy) * p.in_w + ix) * p.ld_in + q * 4 : 0;
    pdst[u] = idx < NPIX * 4 ? ((((q >> 1) * NPIX + pix) * 8 + (q & 1) * 4) | (inside ? 0 : (1 << 30))) : -1;
  }
  const int n_slices = p.cin / MG_CK;
  const __bf16 *wsrc = p.w + (long long)ct * MC_WCHUNK;
  const long long wstep = (long long)n_ct * MC_WCHUNK;

  typename In4<IT>::raw preg[NPL];
  f32x4 wreg[NWLC];
  auto prefetch = [&](int s) {
    const IT *base = in + s * MG_CK;
#pragma unroll
    for (int u = 0; u < NPL; ++u) preg[u] = In4<IT>::ld(base + poff[u]);
    const f32x4 *ws = reinterpret_cast<const f32x4 *>(wsrc + s * wstep);
#pragma unroll
    for (int u = 0; u < NWLC; ++u) {
      const int j = tid + u * MG_THREADS;
      wreg[u] = ws[j < MC_WCHUNK / 8 ? j : 0];
    }
  };
  auto commit = [&]() {
    __bf16 *ph = &patch[0][0][0];
#pragma unroll
    for (int u = 0; u < NPL; ++u) {
      bf16x4 v = In4<IT>::cvt(preg[u]);
      if (pdst[u] & (1 << 30)) v = bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
      if (pdst[u] >= 0) *reinterpret_cast<bf16x4 *>(ph + (pdst[u] & ~(1 << 30))) = v;
    }
    f32x4 *wd = reinterpret_cast<f32x4 *>(&wts[0]);
#pragma unroll
    for (int u = 0; u < NWLC; ++u) {
      const int j = tid + u * MG_THREADS;
      if (j < MC_WCHUNK / 8) wd[j] = wreg[u];
    }
  };

  f32x16 acc[MT][2];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;
  int abase[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) abase[m] = ((2 * (wave * MT + m) + (r >> 4)) * S) * PW + (r & 15) * S;

  auto compute = [&]() {
    const __bf16 *wb = &wts[0];
    const __bf16 *pa = &patch[h][0][0];
    bf16x8 bw[2][2], af[2];
    auto load_b = [&](int tap, bf16x8 (&bb)[2]) {
#pragma unroll
      for (int n = 0; n < 2; ++n) bb[n] = *reinterpret_cast<const bf16x8 *>(wb + ((tap * 2 + h) * MC_BN + n * 32 + r) * 8);
    };
    auto load_a = [&](int tap, int m, bf16x8 &a) { a = *reinterpret_cast<const bf16x8 *>(pa + (abase[m] + (tap / 3) * PW + tap % 3) * 8); };
    load_b(0, bw[0]);
    load_a(0, 0, af[0]);
#pragma unroll
    for (int i = 0; i < 9 * MT; ++i) {
      const int tap = i / MT, m = i % MT;
      const int cb = tap & 1, ca = i & 1;
      if (i + 1 < 9 * MT) {
        const int ntap = (i + 1) / MT, nm = (i + 1) % MT;
        if (nm == 0) load_b(ntap, bw[ntap & 1]);
        load_a(ntap, nm, af[ca ^ 1]);
      }
#pragma unroll
      for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ca], bw[cb][n], acc[m][n], 0, 0, 0);
    }
  };

  prefetch(0);
  for (int s = 0; s < n_slices; ++s) {
    __syncthreads();
    commit();
    __syncthreads();
    if (s + 1 < n_slices) prefetch(s + 1);
    compute();
  }
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int co = ct * MC_BN + n * 32 + r;
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
          mg_store<OT>(out + (((long long)b * p.out_h + oy) * p.out_w + ox) * p.ld_out + co, v);
        }
      }
  }
}

// ---- weight pack: (cout, cin, 3, 3) float32 -> [cin'/16][out_pad/64][9][2][64][8] bf16 -------------------------------------------------------
// forward: contraction = cin, outputs = cout, tap t.  transpose (data gradient): contraction = cout, outputs = cin, tap 8 - t.
__global__ void k_mp_pack3x3(const float *__restrict__ w, int cout, int cin, int transpose, const float *__restrict__ fold, __bf16 *__restrict__ dst,
                             int out_pad, long long total) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int K = transpose ? cout : cin, O = transpose ? cin : cout;
  (void)K;
  int j = (int)(t & 7);
  long long u = t >> 3;
  const int o64 = (int)(u & 63);
  u >>= 6;
  const int hh = (int)(u & 1);
  u >>= 1;
  const int tap = (int)(u % 9);
  u /= 9;
  const int n_nb = out_pad / 64;
  const int nb = (int)(u % n_nb);
  const int ks = (int)(u / n_nb);
  const int o = nb * 64 + o64, k = ks * 16 + hh * 8 + j;
  float v = 0.f;
  if (o < O) {
    if (!transpose) v = w[(((long long)o * cin + k) * 9) + tap] * (fold ? fold[o] : 1.f);
    else v = w[(((long long)k * cin + o) * 9) + (8 - tap)] * (fold ? fold[k] : 1.f);
  }
  dst[t] = (__bf16)v;
}

// every 3x3 layer's two bf16 forms in ONE launch per optimizer step: block -> job by binary search over the jobs' first blocks
__global__ __launch_bounds__(256) void k_mp_pack3x3_group(const pcp_mp_pack_job_t *__restrict__ jobs, int n_jobs) {
  int lo = 0, hi = n_jobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block_start <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const pcp_mp_pack_job_t j = jobs[lo];
  const long long t = (long long)((int)blockIdx.x - j.block_start) * blockDim.x + threadIdx.x;
  const int K = j.transpose ? j.cout : j.cin;
  const long long total = (long long)(K / 16) * (j.out_pad / 64) * MC_WCHUNK;
  if (t >= total) return;
  const int jj = (int)(t & 7);
  long long u = t >> 3;
  const int o64 = (int)(u & 63);
  u >>= 6;
  const int hh = (int)(u & 1);
  u >>= 1;
  const int tap = (int)(u % 9);
  u /= 9;
  const int n_nb = j.out_pad / 64;
  const int nb = (int)(u % n_nb);
  const int ks = (int)(u / n_nb);
  const int o = nb * 64 + o64, k = ks * 16 + hh * 8 + jj;
  const int O = j.transpose ? j.cin : j.cout;
  float v = 0.f;
  if (o < O) v = j.transpose ? j.w[(((long long)k * j.cin + o) * 9) + (8 - tap)] : j.w[(((long long)o * j.cin + k) * 9) + tap];
  reinterpret_cast<__bf16 *>(j.packed)[t] = (__bf16)v;
}

bool mc_fast_ok(const pcp_mp_conv3x3_t *d) {
  return d->cout_pad <= MC_MAX_COUT && d->stride == 1 && d->in_dtype == PCP_DT_BF16 && d->cin % MC_CK == 0 && (d->ld_in & 7) == 0 &&
         (d->out_dtype == PCP_DT_BF16 ? ((d->cout & 7) == 0 && (d->ld_out & 7) == 0) : ((d->cout & 3) == 0 && (d->ld_out & 3) == 0));
}

int mc_tile_rows(const pcp_mp_conv3x3_t *d) {
  // 16-row items (8 waves, two per SIMD) once they give every CU an item, else 8-row items (4 waves): 128->128 @128^2 x 4 frames 21.7 vs
  // 25.9 us with 256 sixteen-row items, 256->256 @64^2 30.3 vs 24.0 us with only 128 of them
  const long long big = (long long)d->batch * ((d->in_h + 15) / 16) * ((d->in_w + MC_TW - 1) / MC_TW) * (d->cout_pad / MC_BN);
  long long min16 = pcp_option(PCP_OPT_MP_TH16_MIN, 256);        // A/B knob for the item-size rule
  if (min16 <= 0) min16 = 256;
  return big >= min16 ? 16 : 8;
}

}  // namespace

extern "C" {

#ifdef MC_STAMP
int pcp_debug_read_mc(void *dst, size_t bytes) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(mc_stamps), bytes < sizeof(mc_stamps) ? bytes : sizeof(mc_stamps)) == hipSuccess ? 0 : 1;
}
#endif

size_t pcp_mp_conv3x3_packed_bytes(int32_t k_channels, int32_t out_pad) {
  if (k_channels <= 0 || (k_channels % 16) || out_pad <= 0 || (out_pad % 64)) return 0;
  return (size_t)(k_channels / 16) * (out_pad / 64) * MC_WCHUNK * 2;
}

int pcp_mp_pack_conv3x3(const float *w, int32_t cout, int32_t cin, int32_t transpose, const float *fold_scale, void *packed, int32_t out_pad,
                        void *stream) {
  if (!w || !packed || cout <= 0 || cin <= 0 || out_pad <= 0 || (out_pad % 64)) return PCP_ERR_ARG;
  const int K = transpose ? cout : cin, O = transpose ? cin : cout;
  if ((K % 16) || out_pad < O || (((uintptr_t)packed) & 15)) return PCP_ERR_ARG;
  const long long total = (long long)(K / 16) * (out_pad / 64) * MC_WCHUNK;
  hipLaunchKernelGGL(k_mp_pack3x3, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, cout, cin, transpose ? 1 : 0,
                     fold_scale, (__bf16 *)packed, out_pad, total);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

int pcp_mp_pack_conv3x3_group_blocks(const pcp_mp_pack_job_t *job) {
  if (!job || !job->w || !job->packed || job->cout <= 0 || job->cin <= 0 || job->out_pad <= 0 || (job->out_pad % 64)) return -1;
  const int K = job->transpose ? job->cout : job->cin, O = job->transpose ? job->cin : job->cout;
  if ((K % 16) || job->out_pad < O) return -1;
  const long long total = (long long)(K / 16) * (job->out_pad / 64) * MC_WCHUNK;
  if (total > 0x7fffffffLL) return -1;
  return (int)((total + 255) / 256);
}

int pcp_mp_pack_conv3x3_group(const pcp_mp_pack_job_t *jobs_device, int32_t n_jobs, int32_t total_blocks, void *stream) {
  if (!jobs_device || n_jobs <= 0 || total_blocks <= 0) return PCP_ERR_ARG;
  hipLaunchKernelGGL(k_mp_pack3x3_group, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_device, n_jobs);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

int pcp_mp_conv3x3_plan(const pcp_mp_conv3x3_t *d, int32_t *fast_kernel, double *executed_flops) {
  if (!d) return PCP_ERR_ARG;
  const bool fast = mc_fast_ok(d);
  if (fast_kernel) *fast_kernel = fast ? 1 : 0;
  if (executed_flops) {
    const int s = d->stride;
    const long long ho = (d->in_h - 1) / s + 1, wo = (d->in_w - 1) / s + 1;
    long long px;
    if (fast) {
      const int th = mc_tile_rows(d);
      px = (long long)d->batch * ((ho + th - 1) / th * th) * ((wo + MC_TW - 1) / MC_TW * MC_TW);
    } else {
      const int th = s == 1 ? 16 : 8;
      px = (long long)d->batch * ((ho + th - 1) / th * th) * ((wo + MG_TW - 1) / MG_TW * MG_TW);
    }
    *executed_flops = 2.0 * (double)px * d->cout_pad * 9.0 * d->cin;
  }
  return PCP_OK;
}

int pcp_mp_conv3x3(const pcp_mp_conv3x3_t *d, const void *in, const void *w_packed, const float *bias, void *out, void *stream) {
  if (!d || !in || !w_packed || !bias || !out) return PCP_ERR_ARG;
  if (d->batch <= 0 || d->in_h <= 0 || d->in_w <= 0 || d->cin <= 0 || (d->cin % 16) || d->cout <= 0 || d->cout_pad < d->cout ||
      (d->cout_pad % MC_BN) || (d->ld_in & 3) || (d->ld_out & 3) || (d->stride != 1 && d->stride != 2) ||
      (d->in_dtype != PCP_DT_F32 && d->in_dtype != PCP_DT_BF16) || (d->out_dtype != PCP_DT_F32 && d->out_dtype != PCP_DT_BF16))
    return PCP_ERR_ARG;
  if ((((uintptr_t)in) & (d->in_dtype == PCP_DT_BF16 ? 7 : 15)) || (((uintptr_t)w_packed) & 15)) return PCP_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int out_h = (d->in_h - 1) / d->stride + 1, out_w = (d->in_w - 1) / d->stride + 1;
  const size_t w_bytes = pcp_mp_conv3x3_packed_bytes(d->cin, d->cout_pad);
  if (mc_fast_ok(d) && !(((uintptr_t)in) & 15) && !(((uintptr_t)out) & 15)) {
    const long long in_bytes = (long long)d->batch * d->in_h * d->in_w * d->ld_in * 2;
    const long long out_bytes = (long long)d->batch * out_h * out_w * d->ld_out * (d->out_dtype == PCP_DT_BF16 ? 2 : 4);
    if (in_bytes < 0x7fffffffLL && out_bytes < 0x7fffffffLL && w_bytes < 0x7fffffffULL) {
      McParams p;
      p.in = in; p.w = (const __bf16 *)w_packed; p.bias = bias; p.out = out;
      p.batch = d->batch; p.h = d->in_h; p.w_ = d->in_w; p.cin = d->cin; p.cout = d->cout; p.cout_pad = d->cout_pad;
      p.ld_in = d->ld_in; p.ld_out = d->ld_out; p.relu = d->relu;
      const int th = mc_tile_rows(d);
      p.tiles_y = (out_h + th - 1) / th;
      p.tiles_x = (out_w + MC_TW - 1) / MC_TW;
      p.n_nb = d->cout_pad / MC_BN;
      p.n_items = d->batch * p.tiles_y * p.tiles_x * p.n_nb;
      p.n_slices = d->cin / MC_CK;
      p.in_bytes = (unsigned)in_bytes; p.w_bytes = (unsigned)w_bytes; p.out_bytes = (unsigned)out_bytes;
      p.diag = (int)pcp_option(PCP_OPT_MP_DIAG, 0);
      const int n_cu = pcp_current_device_cus();                  // cached per device ordinal (abi.hip)
      const int nwg = p.n_items < n_cu ? p.n_items : n_cu;
      const bool ob = d->out_dtype == PCP_DT_BF16;
      if (th == 16) {
        if (ob) hipLaunchKernelGGL((k_mp_conv3x3_s1<16, true>), dim3(nwg), dim3(512), 0, s, p);
        else hipLaunchKernelGGL((k_mp_conv3x3_s1<16, false>), dim3(nwg), dim3(512), 0, s, p);
      } else {
        if (ob) hipLaunchKernelGGL((k_mp_conv3x3_s1<8, true>), dim3(nwg), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((k_mp_conv3x3_s1<8, false>), dim3(nwg), dim3(256), 0, s, p);
      }
      PCP_CHECK_LAUNCH();
      return PCP_OK;
    }
  }
  MgParams g;
  g.in = in; g.w = (const __bf16 *)w_packed; g.bias = bias; g.out = out;
  g.batch = d->batch; g.in_h = d->in_h; g.in_w = d->in_w; g.out_h = out_h; g.out_w = out_w;
  g.cin = d->cin; g.cout = d->cout; g.cout_pad = d->cout_pad; g.ld_in = d->ld_in; g.ld_out = d->ld_out; g.relu = d->relu;
  const int th = d->stride == 1 ? 16 : 8;
  g.tiles_y = (out_h + th - 1) / th;
  g.tiles_x = (out_w + MG_TW - 1) / MG_TW;
  const long long blocks = (long long)g.batch * g.tiles_y * g.tiles_x * (g.cout_pad / MC_BN);
  if (blocks >= (1LL << 31)) return PCP_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)blocks), blk(MG_THREADS);
#define PCP_MG(S, TH, IT, OT) hipLaunchKernelGGL((k_mp_conv3x3_gen<S, TH, IT, OT>), grid, blk, 0, s, g)
  const bool ib = d->in_dtype == PCP_DT_BF16, ob = d->out_dtype == PCP_DT_BF16;
  if (d->stride == 1) {
    if (ib) { if (ob) PCP_MG(1, 16, __bf16, __bf16); else PCP_MG(1, 16, __bf16, float); }
    else { if (ob) PCP_MG(1, 16, float, __bf16); else PCP_MG(1, 16, float, float); }
  } else {
    if (ib) { if (ob) PCP_MG(2, 8, __bf16, __bf16); else PCP_MG(2, 8, __bf16, float); }
    else { if (ob) PCP_MG(2, 8, float, __bf16); else PCP_MG(2, 8, float, float); }
  }
#undef PCP_MG
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

}  // extern "C"
