// a17 -- HunterJr TRAINING branch on the device (configs 1 / 2: v2x_pointpillar_basic_car.yaml / _rsu.yaml).
//
// Replaces, with their autograd graphs:
//   HunterJr._build_meta                    pcdet/models/bev_layers/hunter_jr.py:165-196   (2 x torch.unique + scatter_max / scatter_min)
//   HunterObjectHead.forward                hunter_jr.py:42-76                             (scatter_mean, 3 x scatter_max, cat)
//   HunterJr.assign_target                  hunter_jr.py:198-260
//   HunterPointHead.get_loss_distill        hunter_jr.py:106-113
//   HunterJr.get_training_loss              hunter_jr.py:401-495
//   CELovaszLoss, Lovasz_softmax            pcdet/models/loss_fnc/pcaccum_ce_lovasz_loss.py:20-71, lovasz_softmax.py:56-95
//   quat2mat, remove_gt_boxes_outside_range, hard_mining_regression_loss   hunter_toolbox.py:42-62,161-184,187-219
//   the backward of bilinear_interpolate_torch (incl. the gradient w.r.t. the sampling position that reaches the flow head through the
//   in-place xyz update of hunter_jr.py:265), of bev_scatter and of the 2-way softmax blend      hunter_toolbox.py:8-39,65-91, hunter_jr.py:281-285
//
// (batch, instance, sweep) keys are bounded by B * N_inst_max * NUM_SWEEPS, so both torch.unique calls are a presence table + one scan;
// the sorted order of torch.unique is the key order.  Scalar losses are float64 sums.  The only library call is hipCUB's radix sort for the
// three descending error orders of the Lovasz extension.
#include "pcp_common.h"

#include <hipcub/hipcub.hpp>

#pragma clang fp contract(off)

namespace {

constexpr int HT = 256;

__device__ __forceinline__ double block_sum_d(double v, double *sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wv] = v;
  __syncthreads();
  double t = 0;
  if (threadIdx.x == 0) for (unsigned w = 0; w < (blockDim.x + 63) / 64; ++w) t += sh[w];
  return t;   // valid on thread 0
}

__device__ __forceinline__ float sl1(float d) { const float a = fabsf(d); return a < 1.f ? 0.5f * d * d : a - 0.5f; }
__device__ __forceinline__ float sl1_grad(float d) { return fabsf(d) < 1.f ? d : (d > 0.f ? 1.f : -1.f); }

// ---------------------------------------------------------------------------------------------------------------------
// locals / instances  (hunter_jr.py:165-196)
// ---------------------------------------------------------------------------------------------------------------------
struct MetaWs {
  int *table_local;   // [T]   presence -> rank of the (batch, instance, sweep) key
  int *table_inst;    // [B*M] presence -> rank of the (batch, instance) key
  int *block_count;   // [blocks + 1] foreground rows per 256-row block -> exclusive offsets
};

__host__ __device__ inline size_t a16(size_t v) { return (v + 15) & ~(size_t)15; }

inline size_t meta_ws_bytes(int T, int BM, long long blocks) { return a16(4 * (size_t)T) + a16(4 * (size_t)BM) + a16(4 * (size_t)(blocks + 1)); }

__host__ __device__ inline MetaWs meta_carve(void *ws, int T, int BM) {
  char *p = static_cast<char *>(ws);
  MetaWs w;
  w.table_local = reinterpret_cast<int *>(p); p += a16(4 * (size_t)T);
  w.table_inst = reinterpret_cast<int *>(p); p += a16(4 * (size_t)BM);
  w.block_count = reinterpret_cast<int *>(p);
  return w;
}

__device__ __forceinline__ int point_key(const pcp_hunter_meta_t &d, const float *row) {
  const int inst = (int)row[d.inst_col];                   // .long() truncation; foreground test is inst > -1 on the float
  if (!(row[d.inst_col] > -1.f)) return -1;
  const int b = (int)row[0], sw = (int)row[d.sweep_col];
  if (b < 0 || b >= d.batch || inst >= d.max_inst || sw < 0 || sw >= d.num_sweeps) return -2;   // outside the key table: reported
  return (b * d.max_inst + inst) * d.num_sweeps + sw;
}

__global__ __launch_bounds__(HT) void k_hm_mark(pcp_hunter_meta_t d, const float *__restrict__ points, long long n, int stride, MetaWs w,
                                               int *__restrict__ counts) {
  __shared__ int wave_cnt[HT / 64];
  const long long i = (long long)blockIdx.x * HT + threadIdx.x;
  int fg = 0;
  if (i < n) {
    const int key = point_key(d, points + i * stride);
    if (key >= 0) { fg = 1; w.table_local[key] = 1; }
    if (key == -2) atomicAdd(&counts[3], 1);
  }
  const unsigned long long bal = __ballot(fg);
  if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = __popcll(bal);
  __syncthreads();
  if (threadIdx.x == 0) { int t = 0; for (int k = 0; k < HT / 64; ++k) t += wave_cnt[k]; w.block_count[blockIdx.x] = t; }
}

// exclusive scan of `len` ints in place by ONE workgroup of 1024 threads; returns the total (all threads)
__device__ int wg_exclusive_scan(int *data, long long len, int *sh) {
  __shared__ int running;
  if (threadIdx.x == 0) running = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (long long base = 0; base < len; base += blockDim.x) {
    const long long i = base + threadIdx.x;
    const int v = i < len ? data[i] : 0;
    int inc = v;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) sh[wv] = inc;
    __syncthreads();
    int woff = 0;
    for (int k = 0; k < wv; ++k) woff += sh[k];
    const int start = running;
    if (i < len) data[i] = start + woff + inc - v;
    __syncthreads();
    if (threadIdx.x == blockDim.x - 1) running = start + woff + inc;
    __syncthreads();
  }
  return running;
}

__global__ __launch_bounds__(1024) void k_hm_scan(pcp_hunter_meta_t d, MetaWs w, long long blocks, int *__restrict__ local_key,
                                                 int *__restrict__ local_inst, int *__restrict__ inst_key, int *__restrict__ inst_first,
                                                 int *__restrict__ inst_last, int *__restrict__ counts) {
  __shared__ int sh[16];
  const int T = d.batch * d.max_inst * d.num_sweeps, BM = d.batch * d.max_inst;
  const int tid = threadIdx.x;
  // instance presence from local presence
  for (int k = tid; k < BM; k += 1024) w.table_inst[k] = 0;
  __syncthreads();
  for (int k = tid; k < T; k += 1024) if (w.table_local[k]) w.table_inst[k / d.num_sweeps] = 1;
  __syncthreads();
  // ranks: tables become exclusive prefix sums; an entry is present iff next rank > its rank -> remember presence in the sign of a copy
  // (presence is re-derived below from the key lists)
  // 1. locals
  for (int k = tid; k < T; k += 1024) local_inst[k] = w.table_local[k];            // presence copy (local_inst is T long)
  __syncthreads();
  const int n_local = wg_exclusive_scan(w.table_local, T, sh);
  for (int k = tid; k < T; k += 1024) {
    if (local_inst[k]) local_key[w.table_local[k]] = k; else w.table_local[k] = -1;
  }
  __syncthreads();
  // 2. instances
  for (int k = tid; k < BM; k += 1024) inst_first[k] = w.table_inst[k];            // presence copy (inst_first is BM long)
  __syncthreads();
  const int n_inst = wg_exclusive_scan(w.table_inst, BM, sh);
  for (int k = tid; k < BM; k += 1024) {
    if (inst_first[k]) inst_key[w.table_inst[k]] = k; else w.table_inst[k] = -1;
  }
  __syncthreads();
  // 3. locals -> instances; the locals of an instance are contiguous and ascending in sweep: first = min sweep, last = max sweep
  for (int l = tid; l < n_local; l += 1024) local_inst[l] = w.table_inst[local_key[l] / d.num_sweeps];
  __syncthreads();
  for (int l = tid; l < n_local; l += 1024) {
    const int i = local_inst[l];
    if (l == 0 || local_inst[l - 1] != i) inst_first[i] = l;
    if (l == n_local - 1 || local_inst[l + 1] != i) inst_last[i] = l;
  }
  __syncthreads();
  // 4. block offsets of the ordered foreground list
  const int n_fg = wg_exclusive_scan(w.block_count, blocks, sh);
  if (tid == 0) { counts[0] = n_fg; counts[1] = n_local; counts[2] = n_inst; }
}

__global__ __launch_bounds__(HT) void k_hm_fill(pcp_hunter_meta_t d, const float *__restrict__ points, long long n, int stride, MetaWs w,
                                               int *__restrict__ fg_idx, int *__restrict__ fg_local) {
  __shared__ int wave_cnt[HT / 64];
  const long long i = (long long)blockIdx.x * HT + threadIdx.x;
  int key = -1;
  if (i < n) key = point_key(d, points + i * stride);
  const int fg = key >= 0 ? 1 : 0;
  const unsigned long long bal = __ballot(fg);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) wave_cnt[wv] = __popcll(bal);
  __syncthreads();
  if (!fg) return;
  int pos = w.block_count[blockIdx.x] + __popcll(bal & ((1ull << lane) - 1ull));
  for (int k = 0; k < wv; ++k) pos += wave_cnt[k];
  fg_idx[pos] = (int)i;
  fg_local[pos] = w.table_local[key];
}

// ---------------------------------------------------------------------------------------------------------------------
// segment max with arg-max routing (torch_scatter.scatter_max), row gather / scatter helpers
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int f2key(float v) { const int b = __float_as_int(v); return b >= 0 ? b : b ^ 0x7fffffff; }
__device__ __forceinline__ float key2f(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }

__global__ void k_sm_init(int *__restrict__ out, int ld_out, int *__restrict__ arg, long long n_seg, int c) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_seg * c) return;
  out[(t / c) * ld_out + t % c] = (int)0x80000000;
  arg[t] = 0x7fffffff;
}

__global__ void k_sm_max(const float *__restrict__ src, int ld_src, const int *__restrict__ row_index, long long rows,
                         const int *__restrict__ seg, int c, int *__restrict__ out, int ld_out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rows * c) return;
  const long long r = t / c;
  const int ch = (int)(t % c);
  const long long sr = row_index ? row_index[r] : r;
  atomicMax(&out[(long long)seg[r] * ld_out + ch], f2key(src[sr * ld_src + ch]));
}

__global__ void k_sm_arg(const float *__restrict__ src, int ld_src, const int *__restrict__ row_index, long long rows,
                         const int *__restrict__ seg, int c, const int *__restrict__ out, int ld_out, int *__restrict__ arg) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rows * c) return;
  const long long r = t / c;
  const int ch = (int)(t % c);
  const long long sr = row_index ? row_index[r] : r;
  if (f2key(src[sr * ld_src + ch]) == out[(long long)seg[r] * ld_out + ch]) atomicMin(&arg[(long long)seg[r] * c + ch], (int)r);
}

__global__ void k_sm_finish(int *__restrict__ out, int ld_out, long long n_seg, int c) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_seg * c) return;
  int *p = out + (t / c) * ld_out + t % c;
  *reinterpret_cast<float *>(p) = key2f(*p);
}

// dsrc[row(arg[s, ch]), ch] += dout[s, ch]   (segments own disjoint rows: no two writers per address)
__global__ void k_sm_backward(const float *__restrict__ dout, int ld_dout, const int *__restrict__ arg, long long n_seg, int c,
                              const int *__restrict__ row_index, float *__restrict__ dsrc, int ld_dsrc) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_seg * c) return;
  const int ch = (int)(t % c);
  const int r = arg[t];
  if (r == 0x7fffffff) return;
  const long long sr = row_index ? row_index[r] : r;
  dsrc[sr * ld_dsrc + ch] += dout[(t / c) * ld_dout + ch];
}

__global__ void k_rows_scatter_add(const float *__restrict__ src, int ld_src, const int *__restrict__ row_index, long long rows, int c,
                                   float *__restrict__ dst, int ld_dst) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rows * c) return;
  const long long r = t / c;
  const int ch = (int)(t % c);
  dst[(long long)row_index[r] * ld_dst + ch] += src[r * ld_src + ch];          // row_index is injective (ordered foreground list)
}

// ---------------------------------------------------------------------------------------------------------------------
// object head glue (hunter_jr.py:50-70)
// ---------------------------------------------------------------------------------------------------------------------
__global__ void k_centroid_sum(const float *__restrict__ points, int stride, const int *__restrict__ fg_idx, const int *__restrict__ fg_local,
                               int n_fg, double *__restrict__ acc) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_fg) return;
  const float *row = points + (long long)fg_idx[j] * stride;
  double *a = acc + (long long)fg_local[j] * 4;
  atomicAdd(a + 0, (double)row[1]);
  atomicAdd(a + 1, (double)row[2]);
  atomicAdd(a + 2, (double)row[3]);
  atomicAdd(a + 3, 1.0);
}

__global__ void k_centroid_finish(const double *__restrict__ acc, int n_local, float *__restrict__ centroid) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_local * 3) return;
  const int l = t / 3, k = t % 3;
  const double cnt = acc[l * 4 + 3];
  centroid[t] = (float)acc[l * 4 + k] / (float)(cnt < 1.0 ? 1.0 : cnt);
}

__global__ void k_centered(const float *__restrict__ points, int stride, const int *__restrict__ fg_idx, const int *__restrict__ fg_local,
                           int n_fg, const float *__restrict__ centroid, float *__restrict__ out, int ld_out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n_fg * ld_out) return;
  const int j = (int)(t / ld_out), k = (int)(t % ld_out);
  float v = 0.f;
  if (k < 3) v = points[(long long)fg_idx[j] * stride + 1 + k] - centroid[fg_local[j] * 3 + k];
  out[t] = v;
}

__global__ void k_obj_cat(const float *__restrict__ lf0, const float *__restrict__ gf, const float *__restrict__ centroid,
                          const int *__restrict__ local_inst, const int *__restrict__ inst_last, int n_local, int c, float *__restrict__ out,
                          int ld_out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n_local * ld_out) return;
  const int l = (int)(t / ld_out), k = (int)(t % ld_out);
  const int inst = local_inst[l];
  float v = 0.f;
  if (k < c) v = lf0[(long long)l * c + k];
  else if (k < 2 * c) v = gf[(long long)inst * c + k - c];
  else if (k < 2 * c + 3) v = centroid[l * 3 + k - 2 * c];
  else if (k < 2 * c + 6) v = centroid[inst_last[inst] * 3 + k - 2 * c - 3];
  out[t] = v;
}

__global__ void k_obj_cat_backward(const float *__restrict__ dcat, int ld, const int *__restrict__ inst_first,
                                   const int *__restrict__ inst_last, int n_local, int n_inst, int c, float *__restrict__ dlf0,
                                   float *__restrict__ dgf) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < (long long)n_local * c) dlf0[t] = dcat[(t / c) * ld + t % c];
  if (t < (long long)n_inst * c) {
    const int i = (int)(t / c), k = (int)(t % c);
    float s = 0.f;
    for (int l = inst_first[i]; l <= inst_last[i]; ++l) s += dcat[(long long)l * ld + c + k];
    dgf[t] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// losses
// ---------------------------------------------------------------------------------------------------------------------
// acc (double): [0] ce weighted nll sum [1] ce weight sum [2..4] lovasz per class [5] embed sum [6] distill sum
//               [8] fg_offset loss [9] transl [10] rot [11] recon      ints (as double): [12..14] class counts
constexpr int LACC = 16;

// per foreground point: class target, embedding / offset targets, embedding loss, per-point offset loss value
__global__ __launch_bounds__(HT) void k_hl_points(pcp_hunter_loss_t d, int *__restrict__ labels, unsigned char *__restrict__ fg_dyn,
                                                 float *__restrict__ off_val, float *__restrict__ tgt_emb, float *__restrict__ tgt_off,
                                                 unsigned char *__restrict__ local_mos, double *__restrict__ acc) {
  __shared__ double sh[HT / 64];
  const int j = blockIdx.x * HT + threadIdx.x;
  double emb = 0;
  if (j < d.n_fg) {
    const long long i = d.fg_idx[j];
    const int l = d.fg_local[j];
    const int ikey = d.inst_key[d.local_inst[l]];
    const float *t0 = d.instances_tf + ((long long)ikey * d.num_sweeps + 0) * 12;
    const float nrm = sqrtf(t0[3] * t0[3] + t0[7] * t0[7] + t0[11] * t0[11]);
    const bool mos = nrm > 0.5f;
    labels[i] = mos ? 2 : 1;
    fg_dyn[j] = mos ? 1 : 0;
    local_mos[l] = mos ? 1 : 0;                              // same value from every point of the local
    const float *row = d.points + i * d.stride;
    const float *hp = d.head + i * d.ld_head;
    const float ex = d.gt_boxes[(long long)ikey * 8 + 0] - row[1], ey = d.gt_boxes[(long long)ikey * 8 + 1] - row[2];
    emb = (double)(sl1(hp[6] - ex) + sl1(hp[7] - ey));
    const float *tf = d.instances_tf + (long long)d.local_key[l] * 12;
    float off[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float cr = tf[r * 4 + 0] * row[1] + tf[r * 4 + 1] * row[2] + tf[r * 4 + 2] * row[3] + tf[r * 4 + 3];
      off[r] = cr - row[1 + r];
    }
    off_val[j] = sl1(hp[3] - off[0]) + sl1(hp[4] - off[1]) + sl1(hp[5] - off[2]);
    tgt_emb[j * 2 + 0] = ex; tgt_emb[j * 2 + 1] = ey;
    tgt_off[j * 3 + 0] = off[0]; tgt_off[j * 3 + 1] = off[1]; tgt_off[j * 3 + 2] = off[2];
  }
  const double s = block_sum_d(emb, sh);
  if (threadIdx.x == 0 && s != 0.0) atomicAdd(acc + 5, s);
}

__device__ __forceinline__ void quat_to_mat(const float *q, float *R) {
  const float x = q[0], y = q[1], z = q[2], w = q[3];
  const float w2 = w * w, x2 = x * x, y2 = y * y, z2 = z * z;
  const float wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
  R[0] = w2 + x2 - y2 - z2; R[1] = 2 * xy - 2 * wz; R[2] = 2 * wy + 2 * xz;
  R[3] = 2 * wz + 2 * xy; R[4] = w2 - x2 + y2 - z2; R[5] = 2 * yz - 2 * wx;
  R[6] = 2 * xz - 2 * wy; R[7] = 2 * wx + 2 * yz; R[8] = w2 - x2 - y2 + z2;
}

// per local: translation / rotation loss values
__global__ void k_hl_locals(pcp_hunter_loss_t d, float *__restrict__ tr_val, float *__restrict__ rot_val) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= d.n_local) return;
  const float *p = d.locals_tf + (long long)l * d.ld_locals_tf;
  const float *tf = d.instances_tf + (long long)d.local_key[l] * 12;
  tr_val[l] = sl1(p[0] - tf[3]) + sl1(p[1] - tf[7]) + sl1(p[2] - tf[11]);
  float R[9];
  quat_to_mat(p + 3, R);
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int k = 0; k < 3; ++k) { const float e = R[r * 3 + k] - tf[r * 4 + k]; s += e * e; }
  rot_val[l] = sqrtf(s);
}

// per foreground point: reconstruction loss value (needs the predicted rotation of its local)
__global__ void k_hl_recon(pcp_hunter_loss_t d, float *__restrict__ rec_val) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= d.n_fg) return;
  const int l = d.fg_local[j];
  const float *row = d.points + (long long)d.fg_idx[j] * d.stride;
  const float *p = d.locals_tf + (long long)l * d.ld_locals_tf;
  const float *tf = d.instances_tf + (long long)d.local_key[l] * 12;
  float R[9];
  quat_to_mat(p + 3, R);
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float gtc = tf[r * 4 + 0] * row[1] + tf[r * 4 + 1] * row[2] + tf[r * 4 + 2] * row[3] + tf[r * 4 + 3];
    const float prc = R[r * 3 + 0] * row[1] + R[r * 3 + 1] * row[2] + R[r * 3 + 2] * row[3] + p[r];
    s += sl1(prc - gtc);
  }
  rec_val[j] = s;
}

// hard_mining_regression_loss (hunter_toolbox.py:187-219) by ONE workgroup: loss = mean(positives) + mean(top-k negatives) and
// w[i] = d loss / d val[i].  The k-th largest negative is found by a 4 x 8-bit radix select on the float bits (values >= 0); elements tied
// with it share the remaining slots equally (any split is a valid subgradient; torch.topk's choice among ties is unspecified).
__global__ __launch_bounds__(1024) void k_hard_mining(const float *__restrict__ val, const unsigned char *__restrict__ pos_mask, int n, int ratio,
                                                     int n_neg_when_no_pos, double scale, double *__restrict__ loss_out,
                                                     float *__restrict__ w) {
  __shared__ int hist[256];
  __shared__ double shd[16];
  __shared__ int s_bin, s_rem;
  __shared__ double s_tmp;
  const int tid = threadIdx.x;
  if (n <= 0) { if (tid == 0) *loss_out = 0.0; return; }
  double c = 0, sp = 0;
  for (int i = tid; i < n; i += 1024) if (pos_mask[i]) { c += 1.0; sp += (double)val[i]; }
  double t = block_sum_d(c, shd);
  if (tid == 0) s_tmp = t;
  __syncthreads();
  const int n_pos = (int)s_tmp;
  t = block_sum_d(sp, shd);
  __syncthreads();
  if (tid == 0) s_tmp = t;
  __syncthreads();
  const double sum_pos = s_tmp;
  // candidate set for the top-k: the negatives, or everything when there is no positive
  const bool all = n_pos == 0;
  const int n_cand = all ? n : n - n_pos;
  int k = all ? (n_neg_when_no_pos < n ? n_neg_when_no_pos : n) : (n_pos * ratio < n_cand ? n_pos * ratio : n_cand);
  const float wp = n_pos > 0 ? (float)(scale / n_pos) : 0.f;
  if (n_cand == 0 || k <= 0) {
    for (int i = tid; i < n; i += 1024) w[i] = pos_mask[i] ? wp : 0.f;
    if (tid == 0) *loss_out = n_pos > 0 ? scale * sum_pos / n_pos : 0.0;
    return;
  }
  unsigned prefix = 0;
  int rem = k;                                               // rank still to locate inside the current prefix bucket
  for (int shift = 24; shift >= 0; shift -= 8) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const unsigned himask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
    for (int i = tid; i < n; i += 1024) {
      if (!all && pos_mask[i]) continue;
      const unsigned b = (unsigned)__float_as_int(fmaxf(val[i], 0.f));
      if ((b & himask) == prefix) atomicAdd(&hist[(b >> shift) & 255], 1);
    }
    __syncthreads();
    if (tid == 0) {
      int r = rem, bin = 255;
      for (; bin > 0; --bin) { if (hist[bin] >= r) break; r -= hist[bin]; }
      s_bin = bin; s_rem = r;
    }
    __syncthreads();
    prefix |= (unsigned)s_bin << shift;
    rem = s_rem;
    __syncthreads();
  }
  const float thr = __int_as_float((int)prefix);
  double cgt = 0, sgt = 0, ceq = 0;
  for (int i = tid; i < n; i += 1024) {
    if (!all && pos_mask[i]) continue;
    const float v = fmaxf(val[i], 0.f);
    if (v > thr) { cgt += 1.0; sgt += (double)v; } else if (v == thr) ceq += 1.0;
  }
  t = block_sum_d(cgt, shd); __syncthreads(); if (tid == 0) s_tmp = t; __syncthreads(); const double n_gt = s_tmp; __syncthreads();
  t = block_sum_d(sgt, shd); __syncthreads(); if (tid == 0) s_tmp = t; __syncthreads(); const double sum_gt = s_tmp; __syncthreads();
  t = block_sum_d(ceq, shd); __syncthreads(); if (tid == 0) s_tmp = t; __syncthreads(); const double n_eq = s_tmp; __syncthreads();
  const double take = (double)k - n_gt;                      // slots left for the tied elements, 1 <= take <= n_eq
  const float w_gt = (float)(scale / k), w_eq = (float)(scale * take / (n_eq * k));
  for (int i = tid; i < n; i += 1024) {
    float wi;
    if (!all && pos_mask[i]) wi = wp;
    else { const float v = fmaxf(val[i], 0.f); wi = v > thr ? w_gt : (v == thr ? w_eq : 0.f); }
    w[i] = wi;
  }
  if (tid == 0) *loss_out = scale * ((n_pos > 0 ? sum_pos / n_pos : 0.0) + (sum_gt + take * (double)thr) / k);
}

// ---- cross entropy + Lovasz ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(HT) void k_ce_counts(const int *__restrict__ labels, long long n, double *__restrict__ acc) {
  __shared__ double sh[HT / 64];
  double c0 = 0, c1 = 0, c2 = 0;
  for (long long i = (long long)blockIdx.x * HT + threadIdx.x; i < n; i += (long long)gridDim.x * HT) {
    const int y = labels[i];
    c0 += y == 0; c1 += y == 1; c2 += y == 2;
  }
  double s = block_sum_d(c0, sh); if (threadIdx.x == 0 && s != 0.0) atomicAdd(acc + 12, s);
  s = block_sum_d(c1, sh); if (threadIdx.x == 0 && s != 0.0) atomicAdd(acc + 13, s);
  s = block_sum_d(c2, sh); if (threadIdx.x == 0 && s != 0.0) atomicAdd(acc + 14, s);
}

__device__ __forceinline__ void ce_weights(const double *acc, float *w) {
  const float tot = (float)(acc[12] + acc[13] + acc[14]);
#pragma unroll
  for (int c = 0; c < 3; ++c) w[c] = fminf(fmaxf(sqrtf(tot / (float)acc[12 + c]), 0.f), 50.f);      // count 0 -> inf -> 50
}

__device__ __forceinline__ void softmax3(const float *x, float *p) {
  const float m = fmaxf(x[0], fmaxf(x[1], x[2]));
  const float e0 = expf(x[0] - m), e1 = expf(x[1] - m), e2 = expf(x[2] - m);
  const float s = e0 + e1 + e2;
  p[0] = e0 / s; p[1] = e1 / s; p[2] = e2 / s;
}

__global__ __launch_bounds__(HT) void k_ce_forward(const float *__restrict__ head, int ld_head, const int *__restrict__ labels, long long n,
                                                  float *__restrict__ err, int *__restrict__ idx, double *__restrict__ acc) {
  __shared__ double sh[HT / 64];
  float w[3];
  ce_weights(acc, w);
  double nll = 0, ws = 0;
  const long long i = (long long)blockIdx.x * HT + threadIdx.x;
  if (i < n) {
    const float *x = head + i * ld_head;
    float p[3];
    softmax3(x, p);
    const int y = labels[i];
    const float m = fmaxf(x[0], fmaxf(x[1], x[2]));
    const float lse = m + logf(expf(x[0] - m) + expf(x[1] - m) + expf(x[2] - m));
    nll = (double)(w[y] * (lse - x[y]));
    ws = (double)w[y];
#pragma unroll
    for (int c = 0; c < 3; ++c) err[(long long)c * n + i] = fabsf((y == c ? 1.f : 0.f) - p[c]);
    idx[i] = (int)i;
  }
  double s = block_sum_d(nll, sh); if (threadIdx.x == 0) atomicAdd(acc + 0, s);
  s = block_sum_d(ws, sh); if (threadIdx.x == 0) atomicAdd(acc + 1, s);
}

// Lovasz gradient of one class by ONE workgroup: cumulative foreground count along the descending-error order, Jaccard differences,
// loss = <sorted errors, differences>; g_point[perm[k]] = difference k.
__global__ __launch_bounds__(1024) void k_lovasz_class(const float *__restrict__ err_sorted, const int *__restrict__ perm,
                                                      const int *__restrict__ labels, long long n, int cls, const double *__restrict__ acc_in,
                                                      float *__restrict__ g_point, double *__restrict__ acc) {
  __shared__ int shw[16];
  __shared__ double shd[16];
  __shared__ int running;
  const double gts = acc_in[12 + cls];
  if (gts == 0.0) return;                                    // class absent: skipped by the reference (lovasz_softmax.py:72-73)
  if (threadIdx.x == 0) running = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double loss = 0;
  for (long long base = 0; base < n; base += 1024) {
    const long long k = base + threadIdx.x;
    const int fg = (k < n && labels[perm[k]] == cls) ? 1 : 0;
    int inc = fg;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) shw[wv] = inc;
    __syncthreads();
    int woff = 0;
    for (int q = 0; q < wv; ++q) woff += shw[q];
    const int start = running;
    const int cum = start + woff + inc;                      // inclusive cumulative foreground count at k
    if (k < n) {
      // float32 arithmetic of lovasz_grad (cumsum of float 0/1 is exact below 2^24)
      const float fg_f = (float)fg, cum_f = (float)cum, gts_f = (float)gts;
      const float inter = gts_f - cum_f, uni = gts_f + ((float)(k + 1) - cum_f);
      float jac = 1.f - inter / uni;
      if (k > 0) {
        const float cum_p = cum_f - fg_f;
        const float inter_p = gts_f - cum_p, uni_p = gts_f + ((float)k - cum_p);
        jac = jac - (1.f - inter_p / uni_p);
      }
      g_point[perm[k]] = jac;
      loss += (double)(err_sorted[k] * jac);
    }
    __syncthreads();
    if (threadIdx.x == 1023) running = cum;
    __syncthreads();
  }
  const double s = block_sum_d(loss, shd);
  if (threadIdx.x == 0) acc[2 + cls] = s;
}

__global__ __launch_bounds__(HT) void k_ce_grad(const float *__restrict__ head, int ld_head, const int *__restrict__ labels, long long n,
                                               const float *__restrict__ g_point, const double *__restrict__ acc, float grad_scale,
                                               float *__restrict__ dhead, int ld_dhead) {
  const long long i = (long long)blockIdx.x * HT + threadIdx.x;
  if (i >= n) return;
  float w[3];
  ce_weights(acc, w);
  const int present = (acc[12] > 0.0) + (acc[13] > 0.0) + (acc[14] > 0.0);
  const float inv_present = present > 0 ? 1.f / (float)present : 0.f;
  const float inv_ws = 1.f / (float)acc[1];
  const float *x = head + i * ld_head;
  float p[3], dp[3];
  softmax3(x, p);
  const int y = labels[i];
  float dot = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float tt = y == c ? 1.f : 0.f;
    const float e = tt - p[c];                              // err = |e|; d err / d p = -sign(e)
    const float sg = e > 0.f ? -1.f : (e < 0.f ? 1.f : 0.f);
    dp[c] = acc[12 + c] > 0.0 ? g_point[(long long)c * n + i] * sg * inv_present : 0.f;
    dot += dp[c] * p[c];
  }
  float *g = dhead + i * ld_dhead;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float tt = y == c ? 1.f : 0.f;
    g[c] = (p[c] * (dp[c] - dot) + w[y] * (p[c] - tt) * inv_ws) * grad_scale;
  }
}

// ---- gradients of the regression terms ------------------------------------------------------------------------------------
// per foreground point: d flow (hard-mined offset loss), d embedding, reconstruction gradient accumulated per local (double)
__global__ void k_hl_point_grads(pcp_hunter_loss_t d, const float *__restrict__ tgt_emb, const float *__restrict__ tgt_off,
                                 const float *__restrict__ w_off, const float *__restrict__ w_rec, double *__restrict__ local_acc) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= d.n_fg) return;
  const long long i = d.fg_idx[j];
  const int l = d.fg_local[j];
  const float *hp = d.head + i * d.ld_head;
  float *g = d.dhead + i * d.ld_dhead;
  const float inv_nf = 1.f / (float)d.n_fg;
#pragma unroll
  for (int r = 0; r < 3; ++r) g[3 + r] = w_off[j] * sl1_grad(hp[3 + r] - tgt_off[j * 3 + r]) * d.grad_scale;
  g[6] = sl1_grad(hp[6] - tgt_emb[j * 2 + 0]) * inv_nf * d.grad_scale;
  g[7] = sl1_grad(hp[7] - tgt_emb[j * 2 + 1]) * inv_nf * d.grad_scale;
  // reconstruction: corr = R(q) p + t_pred  ->  d t_pred += w g_r, d R[r][k] += w g_r p_k
  const float *row = d.points + i * d.stride;
  const float *p = d.locals_tf + (long long)l * d.ld_locals_tf;
  const float *tf = d.instances_tf + (long long)d.local_key[l] * 12;
  float R[9];
  quat_to_mat(p + 3, R);
  double *a = local_acc + (long long)l * 12;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float gtc = tf[r * 4 + 0] * row[1] + tf[r * 4 + 1] * row[2] + tf[r * 4 + 2] * row[3] + tf[r * 4 + 3];
    const float prc = R[r * 3 + 0] * row[1] + R[r * 3 + 1] * row[2] + R[r * 3 + 2] * row[3] + p[r];
    const float gr = w_rec[j] * sl1_grad(prc - gtc);
    if (gr != 0.f) {
      atomicAdd(a + r * 4 + 0, (double)(gr * row[1]));
      atomicAdd(a + r * 4 + 1, (double)(gr * row[2]));
      atomicAdd(a + r * 4 + 2, (double)(gr * row[3]));
      atomicAdd(a + r * 4 + 3, (double)gr);
    }
  }
}

__global__ void k_hl_local_grads(pcp_hunter_loss_t d, const float *__restrict__ w_tr, const float *__restrict__ w_rot,
                                 const float *__restrict__ rot_val, const double *__restrict__ local_acc) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= d.n_local) return;
  const float *p = d.locals_tf + (long long)l * d.ld_locals_tf;
  const float *tf = d.instances_tf + (long long)d.local_key[l] * 12;
  const double *a = local_acc + (long long)l * 12;
  float *g = d.dlocals_tf + (long long)l * d.ld_dlocals_tf;
  float R[9], dR[9];
  quat_to_mat(p + 3, R);
  const float nrm = rot_val[l];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    g[r] = (w_tr[l] * sl1_grad(p[r] - tf[r * 4 + 3]) + (float)a[r * 4 + 3]) * d.grad_scale;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      dR[r * 3 + k] = (nrm > 0.f ? w_rot[l] * (R[r * 3 + k] - tf[r * 4 + k]) / nrm : 0.f) + (float)a[r * 4 + k];
  }
  const float x = p[3], y = p[4], z = p[5], w = p[6];
  // R = [[w2+x2-y2-z2, 2xy-2wz, 2wy+2xz], [2wz+2xy, w2-x2+y2-z2, 2yz-2wx], [2xz-2wy, 2wx+2yz, w2-x2-y2+z2]]
  const float dx = 2 * x * dR[0] + 2 * y * dR[1] + 2 * z * dR[2] + 2 * y * dR[3] - 2 * x * dR[4] - 2 * w * dR[5] + 2 * z * dR[6] + 2 * w * dR[7] - 2 * x * dR[8];
  const float dy = -2 * y * dR[0] + 2 * x * dR[1] + 2 * w * dR[2] + 2 * x * dR[3] + 2 * y * dR[4] + 2 * z * dR[5] - 2 * w * dR[6] + 2 * z * dR[7] - 2 * y * dR[8];
  const float dz = -2 * z * dR[0] - 2 * w * dR[1] + 2 * x * dR[2] + 2 * w * dR[3] - 2 * z * dR[4] + 2 * y * dR[5] + 2 * x * dR[6] + 2 * y * dR[7] + 2 * z * dR[8];
  const float dw = 2 * w * dR[0] - 2 * z * dR[1] + 2 * y * dR[2] + 2 * z * dR[3] + 2 * w * dR[4] - 2 * x * dR[5] - 2 * y * dR[6] + 2 * x * dR[7] + 2 * w * dR[8];
  g[3] = dx * d.grad_scale; g[4] = dy * d.grad_scale; g[5] = dz * d.grad_scale; g[6] = dw * d.grad_scale;
  for (int k = 7; k < d.ld_dlocals_tf; ++k) g[k] = 0.f;
}

// feature distillation: 0.1 * mean_j sum_c sl1(local_feat[fg j, c] - locals_feat[local(j), c])  (the "label" is NOT detached, hunter_jr.py:111-112)
__global__ __launch_bounds__(HT) void k_hl_distill(pcp_hunter_loss_t d, double *__restrict__ dlocals_acc, double *__restrict__ acc) {
  __shared__ double sh[HT / 64];
  const long long t = (long long)blockIdx.x * HT + threadIdx.x;
  double v = 0;
  if (t < (long long)d.n_fg * d.c) {
    const int j = (int)(t / d.c), ch = (int)(t % d.c);
    const int l = d.fg_local[j];
    const float diff = d.local_feat[(long long)d.fg_idx[j] * d.ld_local_feat + ch] - d.locals_feat[(long long)l * d.ld_locals_feat + ch];
    v = (double)sl1(diff);
    const float g = 0.1f * sl1_grad(diff) / (float)d.n_fg * d.grad_scale;
    d.dlocal_feat_fg[(long long)j * d.c + ch] = g;
    if (g != 0.f) atomicAdd(dlocals_acc + (long long)l * d.c + ch, -(double)g);
  }
  const double s = block_sum_d(v, sh);
  if (threadIdx.x == 0 && s != 0.0) atomicAdd(acc + 6, s);
}

__global__ void k_d2f(const double *__restrict__ src, float *__restrict__ dst, long long n) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) dst[t] = (float)src[t];
}

__global__ void k_hl_finalize(pcp_hunter_loss_t d, const double *__restrict__ acc, float *__restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int present = (acc[12] > 0.0) + (acc[13] > 0.0) + (acc[14] > 0.0);
  double lov = 0;
  for (int c = 0; c < 3; ++c) if (acc[12 + c] > 0.0) lov += acc[2 + c];
  const double cls = acc[0] / acc[1] + (present > 0 ? lov / present : 0.0);
  const double emb = d.n_fg > 0 ? acc[5] / d.n_fg : 0.0;
  const double dtl = d.n_fg > 0 ? 0.1 * acc[6] / d.n_fg : 0.0;
  out[0] = (float)cls; out[1] = (float)emb; out[2] = (float)acc[8]; out[3] = (float)acc[9]; out[4] = (float)acc[10];
  out[5] = (float)acc[11]; out[6] = (float)dtl;
  out[7] = (float)(cls + emb + acc[8] + acc[9] + acc[10] + acc[11] + dtl);
}

__global__ void k_zero_head_tail(float *__restrict__ dhead, int ld, long long n) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * (ld - 3)) return;
  dhead[(t / (ld - 3)) * ld + 3 + t % (ld - 3)] = 0.f;
}

// ---------------------------------------------------------------------------------------------------------------------
// backward of the BEV correction
// ---------------------------------------------------------------------------------------------------------------------
// fused = map0 * w0 + map1 * w1, (w0, w1) = softmax(logits): one wavefront per pixel
__global__ __launch_bounds__(HT) void k_fuse2_backward(const float *__restrict__ dfused, int ld_df, const float *__restrict__ cat, int ld_cat,
                                                      const float *__restrict__ logits, int ld_logits, long long pixels, int c,
                                                      float *__restrict__ dcat, int ld_dcat, float *__restrict__ dlogits, int ld_dl) {
  const int lane = threadIdx.x & 63;
  const long long pix = ((long long)blockIdx.x * HT + threadIdx.x) >> 6;
  if (pix >= pixels) return;
  const float l0 = logits[pix * ld_logits], l1 = logits[pix * ld_logits + 1];
  const float m = fmaxf(l0, l1);
  const float e0 = expf(l0 - m), e1 = expf(l1 - m);
  const float w0 = e0 / (e0 + e1), w1 = e1 / (e0 + e1);
  float s0 = 0.f, s1 = 0.f;
  for (int ch = lane; ch < c; ch += 64) {
    const float g = dfused[pix * ld_df + ch];
    const float a = cat[pix * ld_cat + ch], b = cat[pix * ld_cat + c + ch];
    s0 += g * a; s1 += g * b;
    dcat[pix * ld_dcat + ch] = g * w0;
    dcat[pix * ld_dcat + c + ch] = g * w1;
  }
  for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
  const float dot = w0 * s0 + w1 * s1;
  for (int k = lane; k < ld_dl; k += 64) dlogits[pix * ld_dl + k] = k == 0 ? w0 * (s0 - dot) : (k == 1 ? w1 * (s1 - dot) : 0.f);
}

// bilinear_interpolate_torch backward: one wavefront per point.  d bev (4 corners, atomic) and, when `bev` is given, the gradient w.r.t.
// the sampling position, added to dxyz[i, 0..1] (the flow-head gradient of the points the in-place correction moved).
__global__ __launch_bounds__(HT) void k_bilinear_backward(const float *__restrict__ dfeat, int ld_dfeat, const unsigned char *__restrict__ row_mask,
                                                         const float *__restrict__ points, long long n, int stride, const float *__restrict__ bev,
                                                         int ld_bev, int batch, int h, int w, int c, float min_x, float min_y, float pix_x,
                                                         float pix_y, float *__restrict__ dbev, int ld_dbev, float *__restrict__ dxyz, int ld_dxyz) {
  const int lane = threadIdx.x & 63;
  const long long i = ((long long)blockIdx.x * HT + threadIdx.x) >> 6;
  if (i >= n) return;
  if (row_mask && !row_mask[i]) return;
  const float *row = points + i * stride;
  const int b = (int)row[0];
  if (b < 0 || b >= batch) return;
  const float x = __fdiv_rn(row[1] - min_x, pix_x), y = __fdiv_rn(row[2] - min_y, pix_y);
  float fx0 = floorf(x), fy0 = floorf(y);
  fx0 = fminf(fmaxf(fx0, -2.0f), (float)w + 1.0f);
  fy0 = fminf(fmaxf(fy0, -2.0f), (float)h + 1.0f);
  int x0 = (int)fx0, y0 = (int)fy0;
  int x1 = x0 + 1, y1 = y0 + 1;
  x0 = min(max(x0, 0), w - 1); x1 = min(max(x1, 0), w - 1);
  y0 = min(max(y0, 0), h - 1); y1 = min(max(y1, 0), h - 1);
  const float ax = (float)x1 - x, bx = x - (float)x0, ay = (float)y1 - y, by = y - (float)y0;
  const float wa = ax * ay, wb = ax * by, wc = bx * ay, wd = bx * by;
  const long long base = (long long)b * h * w;
  const long long oa = (base + (long long)y0 * w + x0), ob = (base + (long long)y1 * w + x0), oc = (base + (long long)y0 * w + x1),
                  od = (base + (long long)y1 * w + x1);
  float gx = 0.f, gy = 0.f;
  for (int ch = lane; ch < c; ch += 64) {
    const float g = dfeat[i * ld_dfeat + ch];
    if (g != 0.f) {
      atomicAdd(dbev + oa * ld_dbev + ch, g * wa);
      atomicAdd(dbev + ob * ld_dbev + ch, g * wb);
      atomicAdd(dbev + oc * ld_dbev + ch, g * wc);
      atomicAdd(dbev + od * ld_dbev + ch, g * wd);
    }
    if (bev) {
      const float Ia = bev[oa * ld_bev + ch], Ib = bev[ob * ld_bev + ch], Ic = bev[oc * ld_bev + ch], Id = bev[od * ld_bev + ch];
      gx += g * ((Ic - Ia) * ay + (Id - Ib) * by);
      gy += g * ((Ib - Ia) * ax + (Id - Ic) * bx);
    }
  }
  if (bev) {
    for (int o = 32; o > 0; o >>= 1) { gx += __shfl_xor(gx, o); gy += __shfl_xor(gy, o); }
    if (lane == 0) {
      dxyz[i * ld_dxyz + 0] += gx / pix_x;
      dxyz[i * ld_dxyz + 1] += gy / pix_y;
    }
  }
}

// remove_gt_boxes_outside_range (hunter_toolbox.py:161-184): ordered compaction per frame, zero padding (the row count M is kept)
__global__ __launch_bounds__(HT) void k_filter_gt(const float *__restrict__ gt, int m, float lo_x, float lo_y, float lo_z, float hi_x, float hi_y,
                                                 float hi_z, float *__restrict__ out) {
  __shared__ int wave_cnt[HT / 64];
  __shared__ int base;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) base = 0;
  for (int i = tid; i < m * 8; i += HT) out[(long long)b * m * 8 + i] = 0.f;
  __syncthreads();
  for (int start = 0; start < m; start += HT) {
    const int i = start + tid;
    int keep = 0;
    const float *r = gt + ((long long)b * m + (i < m ? i : 0)) * 8;
    if (i < m) keep = (r[0] >= lo_x && r[0] < hi_x && r[1] >= lo_y && r[1] < hi_y && r[2] >= lo_z && r[2] < hi_z) ? 1 : 0;
    const unsigned long long bal = __ballot(keep);
    const int lane = tid & 63, wv = tid >> 6;
    if (lane == 0) wave_cnt[wv] = __popcll(bal);
    __syncthreads();
    int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
    for (int k = 0; k < wv; ++k) pos += wave_cnt[k];
    if (keep) for (int k = 0; k < 8; ++k) out[((long long)b * m + pos) * 8 + k] = r[k];
    __syncthreads();
    if (tid == 0) { int t = 0; for (int k = 0; k < HT / 64; ++k) t += wave_cnt[k]; base += t; }
    __syncthreads();
  }
}

inline unsigned nblk(long long n, int per = HT) { return (unsigned)((n + per - 1) / per); }

struct LossWs {
  double *acc;            // LACC
  double *local_acc;      // n_local * 12
  double *dlocals_acc;    // n_local * c
  float *err, *err_sorted, *g_point;      // 3 * n each
  int *idx, *perm;        // n, 3 * n
  float *off_val, *rec_val, *w_off, *w_rec, *tgt_emb, *tgt_off;   // n_fg (x2 / x3 for the targets)
  float *tr_val, *rot_val, *w_tr, *w_rot;                          // n_local
  unsigned char *fg_dyn, *local_mos;
  void *cub; size_t cub_bytes;
  size_t total;
};

inline LossWs loss_carve(void *ws, long long n, int n_fg, int n_local, int c) {
  LossWs L;
  size_t off = 0;
  char *base = static_cast<char *>(ws);
  auto take = [&](size_t bytes) { size_t o = off; off = pcp_align_up(off + (bytes ? bytes : 16), 256); return base ? base + o : (char *)nullptr; };
  const size_t nf = n_fg > 0 ? n_fg : 1, nl = n_local > 0 ? n_local : 1, nn = n > 0 ? n : 1;
  L.acc = (double *)take(sizeof(double) * LACC);
  L.local_acc = (double *)take(sizeof(double) * nl * 12);
  L.dlocals_acc = (double *)take(sizeof(double) * nl * c);
  L.err = (float *)take(4 * 3 * nn); L.err_sorted = (float *)take(4 * 3 * nn); L.g_point = (float *)take(4 * 3 * nn);
  L.idx = (int *)take(4 * nn); L.perm = (int *)take(4 * 3 * nn);
  L.off_val = (float *)take(4 * nf); L.rec_val = (float *)take(4 * nf); L.w_off = (float *)take(4 * nf); L.w_rec = (float *)take(4 * nf);
  L.tgt_emb = (float *)take(4 * 2 * nf); L.tgt_off = (float *)take(4 * 3 * nf);
  L.tr_val = (float *)take(4 * nl); L.rot_val = (float *)take(4 * nl); L.w_tr = (float *)take(4 * nl); L.w_rot = (float *)take(4 * nl);
  L.fg_dyn = (unsigned char *)take(nf); L.local_mos = (unsigned char *)take(nl);
  size_t cub = 0;
  (void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, cub, (const float *)nullptr, (float *)nullptr, (const int *)nullptr, (int *)nullptr,
                                               (int)nn);
  L.cub_bytes = cub;
  L.cub = take(cub);
  L.total = off;
  return L;
}

}  // namespace

extern "C" {

size_t pcp_hunter_meta_workspace_bytes(const pcp_hunter_meta_t *d, int64_t n) {
  if (!d || d->batch <= 0 || d->max_inst <= 0 || d->num_sweeps <= 0 || n < 0) return 0;
  return meta_ws_bytes(d->batch * d->max_inst * d->num_sweeps, d->batch * d->max_inst, (n + HT - 1) / HT);
}

int pcp_hunter_meta(const pcp_hunter_meta_t *d, const float *points, int64_t n, int32_t stride, void *workspace, size_t workspace_bytes,
                    int32_t *fg_idx, int32_t *fg_local, int32_t *local_key, int32_t *local_inst, int32_t *inst_key, int32_t *inst_first,
                    int32_t *inst_last, int32_t *counts, void *stream) {
  if (!d || !points || !workspace || !fg_idx || !fg_local || !local_key || !local_inst || !inst_key || !inst_first || !inst_last || !counts)
    return PCP_ERR_ARG;
  if (d->batch <= 0 || d->max_inst <= 0 || d->num_sweeps <= 0 || n <= 0 || n > 0x7fffffffLL) return PCP_ERR_ARG;
  if (d->sweep_col <= 0 || d->sweep_col >= stride || d->inst_col <= 0 || d->inst_col >= stride) return PCP_ERR_ARG;
  const long long T = (long long)d->batch * d->max_inst * d->num_sweeps;
  if (T > (1 << 24)) return PCP_ERR_UNSUPPORTED;
  const long long blocks = (n + HT - 1) / HT;
  if (workspace_bytes < meta_ws_bytes((int)T, d->batch * d->max_inst, blocks)) return PCP_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MetaWs w = meta_carve(workspace, (int)T, d->batch * d->max_inst);
  if (pcp_zero_async(workspace, a16(4 * (size_t)T), st) != PCP_OK) return PCP_ERR_LAUNCH;
  if (pcp_zero_async(counts, 16, st) != PCP_OK) return PCP_ERR_LAUNCH;
  k_hm_mark<<<(unsigned)blocks, HT, 0, st>>>(*d, points, n, stride, w, counts);
  k_hm_scan<<<1, 1024, 0, st>>>(*d, w, blocks, local_key, local_inst, inst_key, inst_first, inst_last, counts);
  k_hm_fill<<<(unsigned)blocks, HT, 0, st>>>(*d, points, n, stride, w, fg_idx, fg_local);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

int pcp_segment_max(const float *src, int32_t ld_src, const int32_t *row_index, int64_t rows, const int32_t *seg, int64_t n_seg, int32_t c,
                    float *out, int32_t ld_out, int32_t *arg, void *stream) {
  if (!src || !seg || !out || !arg || rows < 0 || n_seg <= 0 || c <= 0 || ld_src < c || ld_out < c) return PCP_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int *oi = reinterpret_cast<int *>(out);
  k_sm_init<<<nblk(n_seg * c), HT, 0, st>>>(oi, ld_out, arg, n_seg, c);
  if (rows > 0) {
    k_sm_max<<<nblk(rows * c), HT, 0, st>>>(src, ld_src, row_index, rows, seg, c, oi, ld_out);
    k_sm_arg<<<nblk(rows * c), HT, 0, st>>>(src, ld_src, row_index, rows, seg, c, oi, ld_out, arg);
  }
  k_sm_finish<<<nblk(n_seg * c), HT, 0, st>>>(oi, ld_out, n_seg, c);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

int pcp_segment_max_backward(const float *dout, int32_t ld_dout, const int32_t *arg, int64_t n_seg, int32_t c, const int32_t *row_index,
                             float *dsrc, int32_t ld_dsrc, void *stream) {
  if (!dout || !arg || !dsrc || n_seg <= 0 || c <= 0) return PCP_ERR_ARG;
  k_sm_backward<<<nblk(n_seg * c), HT, 0, static_cast<hipStream_t>(stream)>>>(dout, ld_dout, arg, n_seg, c, row_index, dsrc, ld_dsrc);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

int pcp_rows_scatter_add(const float *src, int32_t ld_src, const int32_t *row_index, int64_t rows, int32_t c, float *dst, int32_t ld_dst,
                         void *stream) {
  if (!src || !row_index || !dst || rows < 0 || c <= 0) return PCP_ERR_ARG;
  if (rows == 0) return PCP_OK;
  k_rows_scatter_add<<<nblk(rows * c), HT, 0, static_cast<hipStream_t>(stream)>>>(src, ld_src, row_index, rows, c, dst, ld_dst);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

int pcp_hunter_local_centroids(const float *points, int32_t stride, const int32_t *fg_idx, const int32_t *fg_local, int32_t n_fg,
                               int32_t n_local, void *workspace, size_t workspace_bytes, float *centroid, float *centered, int32_t ld_centered,
                               void *stream) {
  if (!points || !fg_idx || !fg_local || !workspace || !centroid || !centered || n_fg <= 0 || n_local <= 0 || ld_centered < 3) return PCP_ERR_ARG;
  if (workspace_bytes < sizeof(double) * 4 * (size_t)n_local) return PCP_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  double *acc = static_cast<double *>(workspace);
  if (pcp_zero_async(acc, sizeof(double) * 4 * (size_t)n_local, st) != PCP_OK) return PCP_ERR_LAUNCH;
  k_centroid_sum<<<nblk(n_fg), HT, 0, st>>>(points, stride, fg_idx, fg_local, n_fg, acc);
  k_centroid_finish<<<nblk(n_local * 3), HT, 0, st>>>(acc, n_local, centroid);
  k_centered<<<nblk((long long)n_fg * ld_centered), HT, 0, st>>>(points, stride, fg_idx, fg_local, n_fg, centroid, centered, ld_centered);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

int pcp_hunter_object_cat(const float *lf0, const float *gf, const float *centroid, const int32_t *local_inst, const int32_t *inst_last,
                          int32_t n_local, int32_t c, float *out, int32_t ld_out, void *stream) {
  if (!lf0 || !gf || !centroid || !local_inst || !inst_last || !out || n_local <= 0 || c <= 0 || ld_out < 2 * c + 6) return PCP_ERR_ARG;
  k_obj_cat<<<nblk((long long)n_local * ld_out), HT, 0, static_cast<hipStream_t>(stream)>>>(lf0, gf, centroid, local_inst, inst_last, n_local, c,
                                                                                             out, ld_out);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

int pcp_hunter_object_cat_backward(const float *dcat, int32_t ld, const int32_t *inst_first, const int32_t *inst_last, int32_t n_local,
                                   int32_t n_inst, int32_t c, float *dlf0, float *dgf, void *stream) {
  if (!dcat || !inst_first || !inst_last || !dlf0 || !dgf || n_local <= 0 || n_inst <= 0 || c <= 0 || ld < 2 * c) return PCP_ERR_ARG;
  k_obj_cat_backward<<<nblk((long long)n_local * c), HT, 0, static_cast<hipStream_t>(stream)>>>(dcat, ld, inst_first, inst_last, n_local, n_inst,
                                                                                                 c, dlf0, dgf);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

size_t pcp_hunter_loss_workspace_bytes(int64_t n, int32_t n_fg, int32_t n_local, int32_t c) {
  if (n <= 0 || n > 0x7fffffffLL || c <= 0) return 0;
  return loss_carve(nullptr, n, n_fg, n_local, c).total;
}

int pcp_hunter_losses(const pcp_hunter_loss_t *d, void *workspace, size_t workspace_bytes, void *stream) {
  if (!d || !workspace || !d->points || !d->gt_boxes || !d->instances_tf || !d->head || !d->dhead || !d->losses || !d->labels) return PCP_ERR_ARG;
  if (d->n <= 0 || d->n > 0x7fffffffLL || d->c <= 0 || d->n_fg < 0 || d->n_local < 0 || d->ld_head < 8 || d->ld_dhead < 8) return PCP_ERR_ARG;
  if (d->n_fg > 0 && (!d->fg_idx || !d->fg_local || !d->local_key || !d->local_inst || !d->inst_key || !d->local_feat || !d->locals_feat ||
                      !d->locals_tf || !d->dlocal_feat_fg || !d->dlocals_feat || !d->dlocals_tf || d->n_local <= 0 || d->ld_locals_tf < 7 ||
                      d->ld_dlocals_tf < 7))
    return PCP_ERR_ARG;
  LossWs L = loss_carve(workspace, d->n, d->n_fg, d->n_local, d->c);
  if (workspace_bytes < L.total) return PCP_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long n = d->n;
  if (pcp_zero_async(L.acc, sizeof(double) * LACC, st) != PCP_OK) return PCP_ERR_LAUNCH;
  if (pcp_zero_async(d->labels, 4 * (size_t)n, st) != PCP_OK) return PCP_ERR_LAUNCH;            // background = class 0
  k_zero_head_tail<<<nblk(n * (d->ld_dhead - 3)), HT, 0, st>>>(d->dhead, d->ld_dhead, n);
  if (d->n_fg > 0) {
    if (pcp_zero_async(L.local_acc, sizeof(double) * 12 * (size_t)d->n_local, st) != PCP_OK) return PCP_ERR_LAUNCH;
    if (pcp_zero_async(L.dlocals_acc, sizeof(double) * (size_t)d->n_local * d->c, st) != PCP_OK) return PCP_ERR_LAUNCH;
    k_hl_points<<<nblk(d->n_fg), HT, 0, st>>>(*d, d->labels, L.fg_dyn, L.off_val, L.tgt_emb, L.tgt_off, L.local_mos, L.acc);
    k_hl_locals<<<nblk(d->n_local), HT, 0, st>>>(*d, L.tr_val, L.rot_val);
    k_hl_recon<<<nblk(d->n_fg), HT, 0, st>>>(*d, L.rec_val);
    const int rf = (int)d->coef_fg, rl = (int)d->coef_locals;
    k_hard_mining<<<1, 1024, 0, st>>>(L.off_val, L.fg_dyn, d->n_fg, rf, 100, 1.0, L.acc + 8, L.w_off);
    k_hard_mining<<<1, 1024, 0, st>>>(L.tr_val, L.local_mos, d->n_local, rl, 100, 1.0, L.acc + 9, L.w_tr);
    k_hard_mining<<<1, 1024, 0, st>>>(L.rot_val, L.local_mos, d->n_local, rl, 100, 1.0, L.acc + 10, L.w_rot);
    // the per-point motion flag of the reconstruction term is the flag of the point's local = fg_dyn
    k_hard_mining<<<1, 1024, 0, st>>>(L.rec_val, L.fg_dyn, d->n_fg, rf, 100, 0.1, L.acc + 11, L.w_rec);
    k_hl_point_grads<<<nblk(d->n_fg), HT, 0, st>>>(*d, L.tgt_emb, L.tgt_off, L.w_off, L.w_rec, L.local_acc);
    k_hl_local_grads<<<nblk(d->n_local), HT, 0, st>>>(*d, L.w_tr, L.w_rot, L.rot_val, L.local_acc);
    k_hl_distill<<<nblk((long long)d->n_fg * d->c), HT, 0, st>>>(*d, L.dlocals_acc, L.acc);
    k_d2f<<<nblk((long long)d->n_local * d->c), HT, 0, st>>>(L.dlocals_acc, d->dlocals_feat, (long long)d->n_local * d->c);
    if (d->tgt_embedding && hipMemcpyAsync(d->tgt_embedding, L.tgt_emb, 4 * 2 * (size_t)d->n_fg, hipMemcpyDeviceToDevice, st) != hipSuccess)
      return PCP_ERR_LAUNCH;
    if (d->tgt_offset && hipMemcpyAsync(d->tgt_offset, L.tgt_off, 4 * 3 * (size_t)d->n_fg, hipMemcpyDeviceToDevice, st) != hipSuccess)
      return PCP_ERR_LAUNCH;
  }
  // segmentation: weighted cross entropy + Lovasz softmax over the three classes
  k_ce_counts<<<(unsigned)(nblk(n) < 512 ? nblk(n) : 512), HT, 0, st>>>(d->labels, n, L.acc);
  k_ce_forward<<<nblk(n), HT, 0, st>>>(d->head, d->ld_head, d->labels, n, L.err, L.idx, L.acc);
  for (int c = 0; c < 3; ++c) {
    size_t cub = L.cub_bytes;
    if (hipcub::DeviceRadixSort::SortPairsDescending(L.cub, cub, L.err + (size_t)c * n, L.err_sorted + (size_t)c * n, L.idx, L.perm + (size_t)c * n,
                                                     (int)n, 0, 32, st) != hipSuccess)
      return PCP_ERR_LAUNCH;
    k_lovasz_class<<<1, 1024, 0, st>>>(L.err_sorted + (size_t)c * n, L.perm + (size_t)c * n, d->labels, n, c, L.acc, L.g_point + (size_t)c * n,
                                       L.acc);
  }
  k_ce_grad<<<nblk(n), HT, 0, st>>>(d->head, d->ld_head, d->labels, n, L.g_point, L.acc, d->grad_scale, d->dhead, d->ld_dhead);
  k_hl_finalize<<<1, 64, 0, st>>>(*d, L.acc, d->losses);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

int pcp_softmax_fuse2_backward(const float *dfused, int32_t ld_df, const float *cat, int32_t ld_cat, const float *logits, int32_t ld_logits,
                               int64_t pixels, int32_t c, float *dcat, int32_t ld_dcat, float *dlogits, int32_t ld_dlogits, void *stream) {
  if (!dfused || !cat || !logits || !dcat || !dlogits || pixels <= 0 || c <= 0 || ld_cat < 2 * c || ld_dcat < 2 * c || ld_logits < 2 || ld_dlogits < 2)
    return PCP_ERR_ARG;
  k_fuse2_backward<<<nblk(pixels * 64), HT, 0, static_cast<hipStream_t>(stream)>>>(dfused, ld_df, cat, ld_cat, logits, ld_logits, pixels, c, dcat,
                                                                                    ld_dcat, dlogits, ld_dlogits);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

int pcp_bev_sample_bilinear_backward(const float *dfeat, int32_t ld_dfeat, const uint8_t *row_mask, const float *points, int64_t n, int32_t stride,
                                     const float *bev, int32_t ld_bev, int32_t batch, int32_t h, int32_t w, int32_t c, float min_x, float min_y,
                                     float pix_x, float pix_y, float *dbev, int32_t ld_dbev, float *dxyz, int32_t ld_dxyz, void *stream) {
  if (!dfeat || !points || !dbev || n <= 0 || c <= 0 || batch <= 0 || h <= 0 || w <= 0 || (bev && !dxyz)) return PCP_ERR_ARG;
  k_bilinear_backward<<<nblk(n * 64), HT, 0, static_cast<hipStream_t>(stream)>>>(dfeat, ld_dfeat, row_mask, points, n, stride, bev, ld_bev, batch, h,
                                                                                  w, c, min_x, min_y, pix_x, pix_y, dbev, ld_dbev, dxyz, ld_dxyz);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

int pcp_filter_gt_boxes(const float *gt_boxes, int32_t batch, int32_t max_boxes, const float *range6_host, float *out, void *stream) {
  if (!gt_boxes || !out || !range6_host || batch <= 0 || max_boxes <= 0) return PCP_ERR_ARG;
  k_filter_gt<<<batch, HT, 0, static_cast<hipStream_t>(stream)>>>(gt_boxes, max_boxes, range6_host[0], range6_host[1], range6_host[2],
                                                                    range6_host[3], range6_host[4], range6_host[5], out);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

}  // extern "C"
