// The pointwise family of the bf16 training loop (include/pcp_hip_mp.h: pcp_mp_pointwise, pcp_mp_pointwise_wgrad): Conv2d 1x1 (PLAIN), Conv2d k2 s2
// (SPACE2DEPTH) and ConvTranspose2d k2 s2 (DEPTH2SPACE) on bf16 or fp32 activations, products on v_mfma_f32_32x32x16_bf16, fp32 accumulation.
// Layers: the up / down-sampling `deblocks` of BaseBEVBackbone (pcdet/models/backbones_2d/base_bev_backbone.py:48-69) as they run under
// torch.cuda.amp.autocast; the fp32 family is csrc/conv.hip (k_pointwise) and csrc/wgrad.hip (k_wgrad_pw).
//
// These layers are HBM-bound (K = 64 .. 512 per output, 2 bytes per element): the kernels are built around wide memory instructions, not
// around the matrix pipe.
//   k_mp_pw      forward and data gradient.  The WEIGHTS are the A operand (rows = output channels) and the pixels the B operand (columns), so
//                a lane ends up with 4 consecutive channels of ONE pixel in every accumulator quad: fp32 outputs leave as 16-byte stores,
//                bf16 outputs as 16-byte stores after one v_permlane32_swap per pair of quads (the epilogue of k_mp_conv3x3_s1).
//                Workgroup (4 waves) = 128 pixels x 64 output channels; per 32-deep K slice the pixel tile [2 k steps][128][16] and the weight
//                tile [2][64][16] are staged in LDS (32-byte rows: a wave's 16-byte operand reads cover 1 KB contiguous -> conflict free),
//                double buffered, register prefetch of slice s + 1 under the MFMAs of slice s, one barrier per slice.
//   k_mp_pw_wgrad  weight gradient: out[n][k] = sum_r a[map(r)][n] * b[map(r)][k], the contraction runs over PIXELS.  Rows are copied global ->
//                LDS as they lie (128-byte records = 64 channels, `buffer_load_dwordx4 ... lds`, 32-byte chunks XOR-swizzled on the source
//                address) and read back transposed by ds_read_b64_tr_b16, as in k_mp_wgrad3x3 (mp_wgrad.hip); a ring of four 32-row stages
//                keeps 24 KB per workgroup in flight; split-K partials are reduced in split order by k_mp_pw_reduce (bitwise reproducible).
#include "pcp_common.h"
#include "../../include/pcp_hip_mp.h"

namespace {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr unsigned MPW_OOB = 0x80000000u;
constexpr int MPW_THREADS = 256;
constexpr int MPW_BM = 128;      // pixels per workgroup
constexpr int MPW_BN = 64;       // output channels per workgroup
constexpr int MPW_CK = 32;       // K per slice (two MFMA k steps)

__device__ __forceinline__ int mpw_xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// eight consecutive channels of a row as bf16 (16 bytes)
template <typename T> struct Ld8;
template <> struct Ld8<__bf16> {
  typedef uint4 raw;
  static __device__ __forceinline__ raw ld(const __bf16 *p) { return *reinterpret_cast<const uint4 *>(p); }
  static __device__ __forceinline__ raw zero() { return make_uint4(0u, 0u, 0u, 0u); }
  static __device__ __forceinline__ uint4 cvt(const raw &v) { return v; }
};
template <> struct Ld8<float> {
  struct raw { f32x4 a, b; };
  static __device__ __forceinline__ raw ld(const float *p) {
    raw r;
    r.a = *reinterpret_cast<const f32x4 *>(p);
    r.b = *reinterpret_cast<const f32x4 *>(p + 4);
    return r;
  }
  static __device__ __forceinline__ raw zero() {
    raw r;
    r.a = f32x4{0.f, 0.f, 0.f, 0.f};
    r.b = r.a;
    return r;
  }
  static __device__ __forceinline__ uint4 cvt(const raw &v) {
    bf16x8 o;
    o[0] = (__bf16)v.a.x; o[1] = (__bf16)v.a.y; o[2] = (__bf16)v.a.z; o[3] = (__bf16)v.a.w;
    o[4] = (__bf16)v.b.x; o[5] = (__bf16)v.b.y; o[6] = (__bf16)v.b.z; o[7] = (__bf16)v.b.w;
    return __builtin_bit_cast(uint4, o);
  }
};

struct MpwParams {
  const void *in;
  const __bf16 *w;      // [k_total/16][n_total][16], n_total = taps_out * cout_pad
  const float *bias;    // [cout_pad]
  void *out;
  long long rows;
  int in_h, in_w;
  int cin, cout, cout_pad, n_total, k_total;
  int ld_in, ld_out, relu, n_tiles;
  unsigned out_bytes;
};

template <int MODE, typename IT, bool OUT_BF16>
__global__ __launch_bounds__(MPW_THREADS, 2) void k_mp_pw(MpwParams p) {
  constexpr int PT = MPW_BM * 32, WT = MPW_BN * 32;                         // bytes of one k step of the pixel / weight tile
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][2 * PT + 2 * WT];       // 2 x 12 KB
  const IT *in = reinterpret_cast<const IT *>(p.in);
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int pw = wave >> 1, cw = wave & 1;                                  // pixel half (64), channel half (32) of the tile
  const int lid = mpw_xcd_remap(blockIdx.x, gridDim.x);
  const int nt = lid % p.n_tiles;
  const long long m0 = (long long)(lid / p.n_tiles) * MPW_BM;
  const int n0 = nt * MPW_BN;

  // ---- staging plan: pixel units (pixel, 8-channel chunk c of the slice) two per thread, weight unit one per thread ----------------------
  const IT *prow[2];
  int pdst[2], pk[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int u = tid + i * MPW_THREADS;
    const int px = u >> 2, c = u & 3;
    const long long m = m0 + px;
    pdst[i] = ((c >> 1) * MPW_BM + px) * 32 + (c & 1) * 16;
    pk[i] = c * 8;
    prow[i] = nullptr;
    if (m < p.rows) {
      if (MODE == PCP_PW_SPACE2DEPTH) {
        const int ow = p.in_w >> 1, oh = p.in_h >> 1;
        const int ox = (int)(m % ow);
        const long long t = m / ow;
        const int oy = (int)(t % oh), bb = (int)(t / oh);
        prow[i] = in + ((long long)(bb * p.in_h + 2 * oy) * p.in_w + 2 * ox) * p.ld_in;
      } else {
        prow[i] = in + m * p.ld_in;
      }
    }
  }
  const int wch = tid >> 2, wc = tid & 3;
  const int wdst = 2 * PT + ((wc >> 1) * MPW_BN + wch) * 32 + (wc & 1) * 16;
  const long long wsrc = ((long long)(wc >> 1) * p.n_total + n0 + wch) * 16 + (wc & 1) * 8;      // elements, inside a 32-deep slice
  const int slices_per_tap = p.cin / MPW_CK;

  typename Ld8<IT>::raw preg[2];
  uint4 wreg;
  auto prefetch = [&](int slice) {
    int c0 = slice * MPW_CK;
    long long tap_off = 0;
    if (MODE == PCP_PW_SPACE2DEPTH) {
      const int tap = slice / slices_per_tap;
      c0 = (slice % slices_per_tap) * MPW_CK;
      tap_off = ((long long)(tap >> 1) * p.in_w + (tap & 1)) * p.ld_in;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) preg[i] = prow[i] ? Ld8<IT>::ld(prow[i] + tap_off + c0 + pk[i]) : Ld8<IT>::zero();
    wreg = *reinterpret_cast<const uint4 *>(p.w + (long long)slice * 2 * p.n_total * 16 + wsrc);
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<uint4 *>(&lds[buf][pdst[i]]) = Ld8<IT>::cvt(preg[i]);
    *reinterpret_cast<uint4 *>(&lds[buf][wdst]) = wreg;
  };

  f32x16 acc[2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
  const int a_off = 2 * PT + (cw * 32 + r) * 32 + h * 16;                   // weights: row = channel
  const int b_off = (pw * 64 + r) * 32 + h * 16;                            // pixels: column

  const int n_slices = p.k_total / MPW_CK;
  prefetch(0);
  commit(0);
  for (int s = 0; s < n_slices; ++s) {
    const int buf = s & 1;
    __syncthreads();
    if (s + 1 < n_slices) prefetch(s + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 a = *reinterpret_cast<const bf16x8 *>(&lds[buf][a_off + ks * WT]);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const bf16x8 bv = *reinterpret_cast<const bf16x8 *>(&lds[buf][b_off + ks * PT + b * 32 * 32]);
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bv, acc[b], 0, 0, 0);
      }
    }
    if (s + 1 < n_slices) commit(buf ^ 1);      // buffer buf ^ 1 was last read in iteration s - 1: every wave is past this iteration's barrier
  }

  // ---- epilogue: bias (+ ReLU); lane (r, h) holds channels cw*32 + 8g + 4h + (0..3) of pixel pw*64 + b*32 + r ---------------------------
  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
  const int nb = n0 + cw * 32;                                              // first packed column of this wave
  int tap = 0, cb = nb;
  if (MODE == PCP_PW_DEPTH2SPACE) {
    tap = nb / p.cout_pad;
    cb = nb % p.cout_pad;
  }
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const long long m = m0 + pw * 64 + b * 32 + r;
    const bool pix_ok = m < p.rows;
    long long opix = m;
    if (MODE == PCP_PW_DEPTH2SPACE) {
      const int ix = (int)(m % p.in_w);
      const long long t = m / p.in_w;
      const int iy = (int)(t % p.in_h), bb = (int)(t / p.in_h);
      opix = ((long long)bb * (2 * p.in_h) + 2 * iy + (tap >> 1)) * (2 * p.in_w) + 2 * ix + (tap & 1);
    }
    const unsigned pix_off = (unsigned)(opix * p.ld_out);                   // elements
    float v[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ch = cb + 8 * g + 4 * h;
      f32x4 bq = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ch < p.cout_pad) bq = *reinterpret_cast<const f32x4 *>(p.bias + ch);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float t = acc[b][4 * g + i] + bq[i];
        if (p.relu) t = fmaxf(t, 0.f);
        v[4 * g + i] = t;
      }
    }
    if (OUT_BF16) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bf16x4 lo, hi;
#pragma unroll
        for (int i = 0; i < 4; ++i) { lo[i] = (__bf16)v[8 * j + i]; hi[i] = (__bf16)v[8 * j + 4 + i]; }
        uint2 ga = __builtin_bit_cast(uint2, lo), gb = __builtin_bit_cast(uint2, hi);
        const auto s0 = __builtin_amdgcn_permlane32_swap(ga.x, gb.x, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(ga.y, gb.y, false, false);
        // lanes 0-31: channels 16j .. 16j+7 of their pixel; lanes 32-63: 16j+8 .. 16j+15
        const uint4 o = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        const int ch = cb + 16 * j + 8 * h;
        unsigned off = MPW_OOB;
        if (pix_ok && ch < p.cout) off = (pix_off + ch) * 2;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), out_rsrc, (int)off, 0, 0);
      }
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = cb + 8 * g + 4 * h;
        unsigned off = MPW_OOB;
        if (pix_ok && ch < p.cout) off = (pix_off + ch) * 4;
        const uint4 o = make_uint4(__float_as_uint(v[4 * g]), __float_as_uint(v[4 * g + 1]), __float_as_uint(v[4 * g + 2]), __float_as_uint(v[4 * g + 3]));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), out_rsrc, (int)off, 0, 0);
      }
    }
  }
}

template <int MODE, typename IT, bool OB>
int launch_mpw(MpwParams p, hipStream_t st) {
  p.n_tiles = p.n_total / MPW_BN;
  const long long blocks = ((p.rows + MPW_BM - 1) / MPW_BM) * p.n_tiles;
  if (blocks <= 0 || blocks > 0x7fffffffLL) return PCP_ERR_ARG;
  hipLaunchKernelGGL((k_mp_pw<MODE, IT, OB>), dim3((unsigned)blocks), dim3(MPW_THREADS), 0, st, p);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

template <int MODE>
int dispatch_mpw(const MpwParams &p, int in_dt, int out_dt, hipStream_t st) {
  if (in_dt == PCP_DT_BF16) return out_dt == PCP_DT_BF16 ? launch_mpw<MODE, __bf16, true>(p, st) : launch_mpw<MODE, __bf16, false>(p, st);
  return out_dt == PCP_DT_BF16 ? launch_mpw<MODE, float, true>(p, st) : launch_mpw<MODE, float, false>(p, st);
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------------------------------------------------------------
constexpr int MWG_THREADS = 256;         // wave = quadrant of the 64 (n) x 64 (k) tile
constexpr int MWG_ROWS = 32;             // rows of a stage (two MFMA k steps)
constexpr int MWG_NST = 4;               // stages resident (three in flight)
constexpr int MWG_OP = MWG_ROWS * 128;   // bytes of one operand of a stage

struct MwgMap {
  const void *ptr;
  int ld, ch, mode, h, w, ky, kx;
  unsigned bytes;
};
struct MwgParams {
  MwgMap a, b;
  long long rows;
  float *part;                           // [nsplit][n_r][k_r]
  int k_tiles, n_r, k_r, nsplit, chunks;
};

__device__ __forceinline__ long long mwg_row(const MwgMap &m, long long r) {
  if (m.mode == 0) return r;
  const int x = (int)(r % m.w);
  const long long t = r / m.w;
  const int y = (int)(t % m.h);
  const long long b = t / m.h;
  return (b * 2 * m.h + 2 * y + m.ky) * 2 * m.w + 2 * x + m.kx;
}

__global__ __launch_bounds__(MWG_THREADS, 2) void k_mp_pw_wgrad(MwgParams p) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[MWG_NST * 2 * MWG_OP];     // 32 KB
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int wn = wave >> 1, wk = wave & 1;
  const int n0 = (blockIdx.x / p.k_tiles) * 64, k0 = (blockIdx.x % p.k_tiles) * 64;
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.a.ptr), 0, p.a.bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.b.ptr), 0, p.b.bytes, 0x00020000);

  // copy plan: an operand of a stage = 32 records x 8 pieces of 16 bytes = 4 wave instructions; wave w issues instruction w of BOTH operands.
  // LDS piece q = w * 64 + lane = record (q >> 3), piece (q & 7); LDS chunk position c2' = piece >> 1 holds channel chunk c2 = c2' ^ sw(rec)
  const int rec = (wave * 64 + lane) >> 3, piece = lane & 7;
  const int c2 = (piece >> 1) ^ (((rec >> 1) & 1) << 1);
  const int ch_in = ((c2 << 1) | (piece & 1)) * 8;                          // first channel of the 16-byte piece inside the 64-channel block
  const bool a_ch_ok = n0 + ch_in < p.a.ch, b_ch_ok = k0 + ch_in < p.b.ch;
  auto issue = [&](int st, int chunk) {
    const long long rr = (long long)chunk * MWG_ROWS + rec;
    unsigned oa = MPW_OOB, ob = MPW_OOB;
    if (rr < p.rows) {
      if (a_ch_ok) oa = (unsigned)((mwg_row(p.a, rr) * p.a.ld + n0 + ch_in) * 2);
      if (b_ch_ok) ob = (unsigned)((mwg_row(p.b, rr) * p.b.ld + k0 + ch_in) * 2);
    }
    unsigned char *base = lds + (st % MWG_NST) * 2 * MWG_OP + wave * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_void *)base, 16, (int)oa, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (lds_void *)(base + MWG_OP), 16, (int)ob, 0, 0, 0);
  };
  // transposed reads (mp_wgrad.hip): lane = (hh, blk, q4, pp); read t of k step kk takes record 16kk + 8hh + 4t + q4, channels 32 half + 16 blk + 4 pp ..
  const int hh = lane >> 5, blk = (lane >> 4) & 1, q4 = (lane >> 2) & 3, pp = lane & 3;
  const int a_lane = (8 * hh + q4) * 128 + (((2 * wn + blk) ^ ((q4 >> 1) << 1)) * 32) + pp * 8;
  const int b_lane = MWG_OP + (8 * hh + q4) * 128 + (((2 * wk + blk) ^ ((q4 >> 1) << 1)) * 32) + pp * 8;
  auto tr_read = [&](const unsigned char *addr) -> s16x4 { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(addr)); };

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  // this workgroup's chunks: blockIdx.y, + nsplit, ...
  const int first = blockIdx.y;
  const int n_my = first < p.chunks ? (p.chunks - first + p.nsplit - 1) / p.nsplit : 0;
#pragma unroll
  for (int d = 0; d < MWG_NST - 1; ++d)
    if (d < n_my) issue(d, first + d * p.nsplit);
  for (int s = 0; s < n_my; ++s) {
    // copies of stages s + 1, s + 2 (two instructions each) may stay in flight
    const int ahead = min(n_my - 1 - s, MWG_NST - 2);
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // stage s + 3 goes into the slot stage s - 1 was read from: every wave is past that stage's reads (this barrier)
    if (s + MWG_NST - 1 < n_my) issue(s + MWG_NST - 1, first + (s + MWG_NST - 1) * p.nsplit);
    const unsigned char *sb = lds + (s % MWG_NST) * 2 * MWG_OP;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const unsigned char *ap = sb + a_lane + 16 * kk * 128, *bp = sb + b_lane + 16 * kk * 128;
      const s16x4 a0 = tr_read(ap), a1 = tr_read(ap + 4 * 128);
      const s16x4 b0 = tr_read(bp), b1 = tr_read(bp + 4 * 128);
      const bf16x8 av = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
      const bf16x8 bv = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
    }
  }
  float *out = p.part + (long long)blockIdx.y * p.n_r * p.k_r;
  const int hl = lane >> 5, l32 = lane & 31;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int n = n0 + wn * 32 + (e >> 2) * 8 + hl * 4 + (e & 3);
    out[(long long)n * p.k_r + k0 + wk * 32 + l32] = acc[e];
  }
}

__global__ __launch_bounds__(256) void k_mp_pw_reduce(const float *__restrict__ part, int nsplit, int n, int k, int n_r, int k_r,
                                                     float *__restrict__ out, int ld_out, int accumulate) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const long long t = (long long)blockIdx.x * 64 + lane;
  float s = 0.f;
  int kk = 0, nn = 0;
  if (t < (long long)n * k) {
    kk = (int)(t % k);
    nn = (int)(t / k);
    const float *src = part + (long long)nn * k_r + kk;
    const long long stride = (long long)n_r * k_r;
    for (int i = sl; i < nsplit; i += 4) s += src[i * stride];
  }
  red[sl][lane] = s;
  __syncthreads();
  if (sl == 0 && t < (long long)n * k) {
    const float v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    float *d = out + (long long)nn * ld_out + kk;
    *d = accumulate ? *d + v : v;
  }
}

inline int mwg_round64(int v) { return (v + 63) / 64 * 64; }
inline int mwg_split(long long rows, int n_r, int k_r, int *chunks) {
  const long long c = (rows + MWG_ROWS - 1) / MWG_ROWS;
  *chunks = (int)c;
  const int tiles = (n_r / 64) * (k_r / 64);
  long long want = (1024 + tiles - 1) / tiles;                 // ~four workgroups per CU: the kernel lives on bytes in flight
  if (want > c / 8) want = c / 8;                              // at least eight stages per workgroup
  if (want < 1) want = 1;
  if (want > 512) want = 512;
  return (int)want;
}

}  // namespace

extern "C" {

int pcp_mp_pointwise(const pcp_mp_pointwise_t *d, const void *in, const void *w_packed, const float *bias, void *out, void *stream_) {
  if (!d || !in || !w_packed || !bias || !out) return PCP_ERR_ARG;
  if ((d->in_dtype != PCP_DT_F32 && d->in_dtype != PCP_DT_BF16) || (d->out_dtype != PCP_DT_F32 && d->out_dtype != PCP_DT_BF16)) return PCP_ERR_ARG;
  if (d->cin <= 0 || d->cin % MPW_CK != 0 || d->cout <= 0 || d->cout_pad < d->cout || d->cout_pad % MPW_BN != 0) return PCP_ERR_UNSUPPORTED;
  if (d->cout % 8 != 0 || d->ld_in % 8 != 0 || d->ld_out % 8 != 0) return PCP_ERR_UNSUPPORTED;
  if ((((uintptr_t)in) | ((uintptr_t)w_packed) | ((uintptr_t)out) | ((uintptr_t)bias)) & 15) return PCP_ERR_ARG;
  hipStream_t st = (hipStream_t)stream_;
  MpwParams p{};
  p.in = in; p.w = (const __bf16 *)w_packed; p.bias = bias; p.out = out;
  p.in_h = d->in_h; p.in_w = d->in_w;
  p.cin = d->cin; p.cout = d->cout; p.cout_pad = d->cout_pad;
  p.ld_in = d->ld_in; p.ld_out = d->ld_out; p.relu = d->relu;
  const size_t osz = d->out_dtype == PCP_DT_BF16 ? 2 : 4;
  long long out_pixels;
  switch (d->mode) {
    case PCP_PW_PLAIN:
      if (d->rows <= 0) return d->rows == 0 ? PCP_OK : PCP_ERR_ARG;
      p.rows = d->rows; p.k_total = d->cin; p.n_total = d->cout_pad;
      out_pixels = p.rows;
      break;
    case PCP_PW_SPACE2DEPTH:
      if (d->batch <= 0 || d->in_h <= 0 || d->in_w <= 0 || (d->in_h & 1) || (d->in_w & 1)) return PCP_ERR_ARG;
      p.rows = (long long)d->batch * (d->in_h / 2) * (d->in_w / 2);
      p.k_total = 4 * d->cin; p.n_total = d->cout_pad;
      out_pixels = p.rows;
      break;
    case PCP_PW_DEPTH2SPACE:
      if (d->batch <= 0 || d->in_h <= 0 || d->in_w <= 0) return PCP_ERR_ARG;
      p.rows = (long long)d->batch * d->in_h * d->in_w;
      p.k_total = d->cin; p.n_total = 4 * d->cout_pad;
      out_pixels = 4 * p.rows;
      break;
    default:
      return PCP_ERR_UNSUPPORTED;
  }
  const unsigned long long ob = (unsigned long long)out_pixels * d->ld_out * osz;
  if (ob > 0x7fffffffULL) return PCP_ERR_UNSUPPORTED;               // 32-bit buffer offsets
  p.out_bytes = (unsigned)ob;
  switch (d->mode) {
    case PCP_PW_PLAIN: return dispatch_mpw<PCP_PW_PLAIN>(p, d->in_dtype, d->out_dtype, st);
    case PCP_PW_SPACE2DEPTH: return dispatch_mpw<PCP_PW_SPACE2DEPTH>(p, d->in_dtype, d->out_dtype, st);
    default: return dispatch_mpw<PCP_PW_DEPTH2SPACE>(p, d->in_dtype, d->out_dtype, st);
  }
}

size_t pcp_mp_pointwise_wgrad_workspace_bytes(int64_t rows, int32_t n, int32_t k) {
  int chunks;
  const int n_r = mwg_round64(n), k_r = mwg_round64(k);
  const int ns = mwg_split(rows, n_r, k_r, &chunks);
  return (size_t)ns * n_r * k_r * sizeof(float);
}

int pcp_mp_pointwise_wgrad(const pcp_mp_rowmap_t *a, const pcp_mp_rowmap_t *b, int64_t rows, void *workspace, size_t workspace_bytes,
                           float *out, int32_t ld_out, int32_t accumulate, void *stream) {
  if (!a || !b || !a->ptr || !b->ptr || !workspace || !out || rows <= 0) return PCP_ERR_ARG;
  if (a->dtype != PCP_DT_BF16 || b->dtype != PCP_DT_BF16) return PCP_ERR_UNSUPPORTED;
  if ((a->channels & 7) || (b->channels & 7) || (a->ld & 7) || (b->ld & 7) || a->channels <= 0 || b->channels <= 0) return PCP_ERR_UNSUPPORTED;
  if ((((uintptr_t)a->ptr) | ((uintptr_t)b->ptr)) & 15) return PCP_ERR_ARG;
  if (a->extent_bytes == 0 || b->extent_bytes == 0 || a->extent_bytes > 0x7fffffffULL || b->extent_bytes > 0x7fffffffULL) return PCP_ERR_UNSUPPORTED;
  MwgParams p{};
  auto cvt = [](const pcp_mp_rowmap_t *m) {
    MwgMap r;
    r.ptr = m->ptr; r.ld = m->ld; r.ch = m->channels; r.mode = m->lattice ? 1 : 0;
    r.h = m->grid_h; r.w = m->grid_w; r.ky = m->ky; r.kx = m->kx;
    r.bytes = (unsigned)m->extent_bytes;
    return r;
  };
  p.a = cvt(a); p.b = cvt(b);
  if ((p.a.mode && (p.a.h <= 0 || p.a.w <= 0)) || (p.b.mode && (p.b.h <= 0 || p.b.w <= 0))) return PCP_ERR_ARG;
  p.rows = rows;
  p.n_r = mwg_round64(a->channels); p.k_r = mwg_round64(b->channels);
  p.k_tiles = p.k_r / 64;
  p.nsplit = mwg_split(rows, p.n_r, p.k_r, &p.chunks);
  if (workspace_bytes < (size_t)p.nsplit * p.n_r * p.k_r * sizeof(float)) return PCP_ERR_WORKSPACE;
  p.part = (float *)workspace;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_mp_pw_wgrad, dim3((p.n_r / 64) * p.k_tiles, p.nsplit), dim3(MWG_THREADS), 0, s, p);
  const long long total = (long long)a->channels * b->channels;
  hipLaunchKernelGGL(k_mp_pw_reduce, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, (const float *)workspace, p.nsplit,
                     a->channels, b->channels, p.n_r, p.k_r, out, ld_out, accumulate);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

}  // extern "C"
