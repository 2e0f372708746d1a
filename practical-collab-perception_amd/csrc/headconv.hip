// a7 -- the final 3x3 convolutions of the CenterHead branches as ONE grouped small-N kernel.
//
// Reference: SeparateHead (pcdet/models/dense_heads/center_head.py:13-47): per branch Conv3x3(64->64)+BN+ReLU, Conv3x3(64->k),
// k in {2,1,3,2,1}.  The five 64->64 convs share their input and run as one 64->320 MFMA launch; what is left is a grouped conv
// (group g: its own 64 input channels -> k_g <= 4 outputs, 9 real output channels in total).  An MFMA tile is 32 channels
// wide, so the tensor path would execute 32/9 x 5 (block-diagonal zeros) = 17x the useful FLOPs (measured 156 us); here the
// 0.7 GFLOP run on the VALU with LDS-staged patches:
//   workgroup = 8x8 output pixels; per group: 10x10 pixels x 64 channels staged in LDS (rows padded to 68 floats: the 16
//   lanes of a ds_read_b128 group hit distinct slots); wave = channel quarter (weight addresses wave-uniform -> scalar
//   loads), lane = pixel; partial sums of the four quarters are combined through LDS.
#include "pcp_common.h"

namespace {

constexpr int HG_MAX_GROUPS = 8;
constexpr int HG_MAX_K = 4;
constexpr int HG_CG = 64;            // channels per group
constexpr int HG_LD = 68;
constexpr int HG_T = 8;              // 8x8 output tile
constexpr int HG_P = HG_T + 2;

struct HeadConvParams {
  const float *in;      // (B, H, W, ld_in): group g reads channels [g*64, g*64+64)
  const float *w;       // [n_out][9][64 * chunks]
  const float *bias;    // [n_out]
  float *out;           // (B, H, W, ld_out): output channel o at offset o
  int batch, h, w_, ld_in, ld_out;
  int groups;
  int chunks;                    // 64-channel chunks per group (input channels per group = 64 * chunks)
  int goff[HG_MAX_GROUPS + 1];   // output-channel range of each group
  int tiles_x, tiles_y;
};

__global__ __launch_bounds__(256) void k_head_grouped(HeadConvParams p) {
  __shared__ __attribute__((aligned(16))) float patch[HG_P * HG_P * HG_LD];
  __shared__ __attribute__((aligned(16))) float wsl[HG_MAX_K * 9 * HG_CG];     // the slice's weights [k][tap][64 channels]
  __shared__ float part[4][64][HG_MAX_K];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int sp = blockIdx.x;
  const int tile_x = sp % p.tiles_x;
  sp /= p.tiles_x;
  const int tile_y = sp % p.tiles_y;
  const int b = sp / p.tiles_y;
  const int oy0 = tile_y * HG_T, ox0 = tile_x * HG_T;
  const int py = lane >> 3, px = lane & 7;

  const int cg = HG_CG * p.chunks;
  // software pipeline over the (group, chunk) slices: the next slice's 10x10x64 patch is loaded global -> registers while the
  // current one is multiplied (the kernel was bound by the exposed global-load latency between its two barriers per slice)
  constexpr int NLD = (HG_P * HG_P * 16 + 255) / 256;       // 7 float4 per thread
  const int n_slices = p.groups * p.chunks;
  f32x4 pre[NLD];
  constexpr int WLD4 = (HG_MAX_K * 9 * HG_CG / 4 + 255) / 256;     // 3 float4 per thread (576 items for k = 4)
  f32x4 prew[WLD4];
  auto load_slice = [&](int s) {
    const int g = s / p.chunks, ch = s % p.chunks;
    {
      const int o0 = p.goff[g], kg = p.goff[g + 1] - o0;
#pragma unroll
      for (int u = 0; u < WLD4; ++u) {
        const int idx = tid + u * 256;                       // (k * 9 + tap) * 16 + q
        const int kt = idx >> 4, q = idx & 15;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (kt < kg * 9) v = *reinterpret_cast<const f32x4 *>(p.w + ((long long)o0 * 9 + kt) * cg + ch * HG_CG + q * 4);
        prew[u] = v;
      }
    }
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int idx = tid + u * 256;
      const int pix = idx >> 4, q = idx & 15;
      const int iy = oy0 - 1 + pix / HG_P, ix = ox0 - 1 + pix % HG_P;
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (idx < HG_P * HG_P * 16 && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w_)
        v = *reinterpret_cast<const f32x4 *>(p.in + ((long long)(b * p.h + iy) * p.w_ + ix) * p.ld_in + g * cg + ch * HG_CG + q * 4);
      pre[u] = v;
    }
  };
  auto store_slice = [&]() {
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int idx = tid + u * 256;
      if (idx < HG_P * HG_P * 16) *reinterpret_cast<f32x4 *>(patch + (idx >> 4) * HG_LD + (idx & 15) * 4) = pre[u];
    }
#pragma unroll
    for (int u = 0; u < WLD4; ++u) {
      const int idx = tid + u * 256;
      if (idx < HG_MAX_K * 9 * 16) *reinterpret_cast<f32x4 *>(wsl + idx * 4) = prew[u];
    }
  };
  load_slice(0);
  float acc[HG_MAX_K] = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < n_slices; s++) {
    const int g = s / p.chunks, ch = s % p.chunks;
    const int o0 = p.goff[g], kg = p.goff[g + 1] - o0;
    __syncthreads();                       // previous slice's reads of `patch` (and of `part`) are done
    store_slice();
    __syncthreads();
    if (s + 1 < n_slices) load_slice(s + 1);
#pragma unroll
    for (int tap = 0; tap < 9; tap++) {
      const float *src = patch + ((py + tap / 3) * HG_P + px + tap % 3) * HG_LD + wave * 16;
      f32x4 x0 = *reinterpret_cast<const f32x4 *>(src), x1 = *reinterpret_cast<const f32x4 *>(src + 4);
      f32x4 x2 = *reinterpret_cast<const f32x4 *>(src + 8), x3 = *reinterpret_cast<const f32x4 *>(src + 12);
      const float xs[16] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y, x2.z, x2.w, x3.x, x3.y, x3.z, x3.w};
#pragma unroll
      for (int k = 0; k < HG_MAX_K; k++) {
        if (k < kg) {                                           // wave-uniform
          // all lanes read the same 64 bytes: LDS broadcast (the first version fetched them with 18 dependent s_load_dwordx16 per slice)
          const float *wr = wsl + (k * 9 + tap) * HG_CG + wave * 16;
          const f32x4 w0 = *reinterpret_cast<const f32x4 *>(wr), w1 = *reinterpret_cast<const f32x4 *>(wr + 4);
          const f32x4 w2 = *reinterpret_cast<const f32x4 *>(wr + 8), w3 = *reinterpret_cast<const f32x4 *>(wr + 12);
          const float ws[16] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w, w3.x, w3.y, w3.z, w3.w};
#pragma unroll
          for (int c = 0; c < 16; c++) acc[k] = fmaf(ws[c], xs[c], acc[k]);
        }
      }
    }
    if (ch == p.chunks - 1) {              // group finished: combine the four channel quarters
#pragma unroll
      for (int k = 0; k < HG_MAX_K; k++) { part[wave][lane][k] = acc[k]; acc[k] = 0.f; }
      __syncthreads();
      if (wave == 0) {
        const int oy = oy0 + py, ox = ox0 + px;
        if (oy < p.h && ox < p.w_) {
          float *dst = p.out + ((long long)(b * p.h + oy) * p.w_ + ox) * p.ld_out + o0;
          for (int k = 0; k < kg; k++)
            dst[k] = part[0][lane][k] + part[1][lane][k] + part[2][lane][k] + part[3][lane][k] + p.bias[o0 + k];
        }
      }
    }
  }
}

}  // namespace

extern "C" int pcp_conv3x3_grouped_small(const float *in, int32_t batch, int32_t h, int32_t w, int32_t ld_in, int32_t groups,
                                         int32_t cin_per_group, const int32_t *group_out_offsets_host, const float *weights, const float *bias,
                                         float *out, int32_t ld_out, void *stream_) {
  if (!in || !weights || !bias || !out || !group_out_offsets_host) return PCP_ERR_ARG;
  if (batch <= 0 || h <= 0 || w <= 0 || groups <= 0 || groups > HG_MAX_GROUPS || (ld_in & 3) || cin_per_group <= 0 ||
      cin_per_group % HG_CG != 0 || ld_in < groups * cin_per_group)
    return PCP_ERR_ARG;
  if ((((uintptr_t)in) & 15)) return PCP_ERR_ARG;
  HeadConvParams p;
  p.in = in; p.w = weights; p.bias = bias; p.out = out;
  p.batch = batch; p.h = h; p.w_ = w; p.ld_in = ld_in; p.ld_out = ld_out; p.groups = groups; p.chunks = cin_per_group / HG_CG;
  for (int g = 0; g <= HG_MAX_GROUPS; g++) p.goff[g] = g <= groups ? group_out_offsets_host[g] : 0;
  for (int g = 0; g < groups; g++) {
    int kg = p.goff[g + 1] - p.goff[g];
    if (kg <= 0 || kg > HG_MAX_K) return PCP_ERR_UNSUPPORTED;
  }
  if (p.goff[groups] > ld_out) return PCP_ERR_ARG;
  p.tiles_x = (w + HG_T - 1) / HG_T;
  p.tiles_y = (h + HG_T - 1) / HG_T;
  long long blocks = (long long)batch * p.tiles_x * p.tiles_y;
  hipLaunchKernelGGL(k_head_grouped, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, p);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}
