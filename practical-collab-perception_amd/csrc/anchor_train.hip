// a16 -- AnchorHeadSingle training on the device: AxisAlignedTargetAssigner, the three anchor-head losses and dL/d(head maps).
//
// Replaces (python loops over frames x anchor classes, an (anchors x boxes) IoU matrix per pair, .nonzero() host syncs, and the
// autograd graph of ~40 small ATen ops in the reference):
//   AxisAlignedTargetAssigner.assign_targets / assign_targets_single   pcdet/models/dense_heads/target_assigner/axis_aligned_target_assigner.py:37-210
//     (POS_FRACTION < 0, MATCH_HEIGHT False, NORM_BY_NUM_EXAMPLES False: the settings of every anchor-head YAML of the reference)
//   boxes3d_nearest_bev_iou, boxes_iou_normal, boxes3d_lidar_to_aligned_bev_boxes   pcdet/utils/box_utils.py:291-340
//   ResidualCoder.encode_torch                                                      pcdet/utils/box_coder_utils.py:13-44
//   AnchorHeadTemplate.get_cls_layer_loss / add_sin_difference / get_direction_target / get_box_reg_layer_loss / get_loss
//                                                                                   pcdet/models/dense_heads/anchor_head_template.py:99-216
//   SigmoidFocalClassificationLoss, WeightedSmoothL1Loss, WeightedCrossEntropyLoss  pcdet/utils/loss_utils.py:9-148,180-208
//
// The IoU matrix is never stored: an anchor's row is recomputed in both passes with the same float32 expression order (contraction off),
// so "this anchor holds a box's best overlap" is an exact equality against an order-independent atomic max.  Scalar losses are float64
// block sums + atomics (order independent to ~1e-16); the labels are integers and bit exact.
#include "pcp_common.h"

#pragma clang fp contract(off)

namespace {

constexpr int AT_THREADS = 256;
constexpr int AT_MAX_BOXES = 1024;

struct GtBev { float x1, y1, x2, y2; };

// box_utils.py:314-325 / common_utils.py:25-28 in float32
__device__ __forceinline__ GtBev aligned_bev(const float *b) {
  const float pi = 3.14159265358979323846f;
  const float r = b[6];
  const float rot = fabsf(r - floorf(r / pi + 0.5f) * pi);
  const bool keep = rot < 0.78539816339744830962f;
  const float dx = keep ? b[3] : b[4];
  const float dy = keep ? b[4] : b[3];
  GtBev o;
  o.x1 = b[0] - dx / 2.f; o.y1 = b[1] - dy / 2.f;
  o.x2 = b[0] + dx / 2.f; o.y2 = b[1] + dy / 2.f;
  return o;
}

// box_utils.py:291-311
__device__ __forceinline__ float iou_normal(const GtBev &a, const GtBev &b) {
  const float x_min = fmaxf(a.x1, b.x1), x_max = fminf(a.x2, b.x2);
  const float y_min = fmaxf(a.y1, b.y1), y_max = fminf(a.y2, b.y2);
  const float x_len = fmaxf(x_max - x_min, 0.f), y_len = fmaxf(y_max - y_min, 0.f);
  const float area_a = (a.x2 - a.x1) * (a.y2 - a.y1);
  const float area_b = (b.x2 - b.x1) * (b.y2 - b.y1);
  const float inter = x_len * y_len;
  return inter / fmaxf(area_a + area_b - inter, 1e-6f);
}

// workspace: int count[B] | float gt_max[B][G][M] (as int bits) | GtBev bev[B][M] | int cls_index[B][M]
struct AssignWs {
  int *count;
  int *gt_max;
  GtBev *bev;
  int *cls;
};

__host__ __device__ inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

__host__ __device__ inline AssignWs carve(void *ws, int B, int G, int M) {
  char *p = static_cast<char *>(ws);
  AssignWs w;
  w.count = reinterpret_cast<int *>(p); p += align16(sizeof(int) * B);
  w.gt_max = reinterpret_cast<int *>(p); p += align16(sizeof(int) * (size_t)B * G * M);
  w.bev = reinterpret_cast<GtBev *>(p); p += align16(sizeof(GtBev) * (size_t)B * M);
  w.cls = reinterpret_cast<int *>(p);
  return w;
}

inline size_t assign_ws_bytes(int B, int G, int M) {
  return align16(sizeof(int) * B) + align16(sizeof(int) * (size_t)B * G * M) + align16(sizeof(GtBev) * (size_t)B * M) +
         align16(sizeof(int) * (size_t)B * M);
}

// one workgroup per frame: rows past the last non-zero-sum row are collate padding (axis_aligned_target_assigner.py:54-58: the scan
// stops at row 0, which is always kept); class c selects CLASS_NAMES[c - 1] with numpy's negative wrap (:62-66: class 0 -> the LAST name)
__global__ __launch_bounds__(AT_THREADS) void k_assign_prepare(pcp_anchor_assign_t d, const float *__restrict__ gt, int m, AssignWs w) {
  __shared__ int last;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float *g = gt + (long long)b * m * 8;
  if (tid == 0) last = 0;
  __syncthreads();
  for (int i = tid; i < m; i += AT_THREADS) {
    const float *r = g + i * 8;
    float s = r[0];
    for (int j = 1; j < 7; ++j) s += r[j];
    if (s != 0.f) atomicMax(&last, i);
    w.bev[(long long)b * m + i] = aligned_bev(r);
    int c = (int)r[7] - 1;                                  // .int() truncates
    if (c < 0) c += d.num_class;
    w.cls[(long long)b * m + i] = (c >= 0 && c < d.num_class) ? c : -1;
  }
  for (int i = tid; i < d.num_groups * m; i += AT_THREADS) w.gt_max[(long long)b * d.num_groups * m + i] = 0;
  __syncthreads();
  if (tid == 0) w.count[b] = m > 0 ? last + 1 : 0;
}

// pass 1: gt_max[b][g][j] = max over the anchors of anchor class g of IoU(anchor, box j)   (IoU >= 0: int bit order = float order)
__global__ __launch_bounds__(AT_THREADS) void k_assign_gt_max(pcp_anchor_assign_t d, const float *__restrict__ anchors, int m, AssignWs w) {
  __shared__ GtBev s_bev[AT_MAX_BOXES];
  __shared__ int s_cls[AT_MAX_BOXES];
  __shared__ int s_max[AT_MAX_BOXES];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int n = d.h * d.w * d.anchors_per_loc;
  const int cnt = w.count[b];
  for (int j = tid; j < cnt; j += AT_THREADS) {
    s_bev[j] = w.bev[(long long)b * m + j];
    s_cls[j] = w.cls[(long long)b * m + j];
  }
  const int i0 = blockIdx.x * AT_THREADS;
  // the anchors of one workgroup span at most ceil(256 / A) + 1 locations, all anchor classes: one LDS max row per anchor class would be
  // G x M ints; instead the workgroup walks the anchor classes one at a time (G <= 8)
  for (int g = 0; g < d.num_groups; ++g) {
    for (int j = tid; j < cnt; j += AT_THREADS) s_max[j] = 0;
    __syncthreads();
    const int i = i0 + tid;
    if (i < n && d.slot_group[i % d.anchors_per_loc] == g) {
      const GtBev a = aligned_bev(anchors + (long long)i * 7);
      const int want = d.group_class[g];
      for (int j = 0; j < cnt; ++j) {
        if (s_cls[j] != want) continue;
        const float v = iou_normal(a, s_bev[j]);
        if (v > 0.f) atomicMax(&s_max[j], __float_as_int(v));
      }
    }
    __syncthreads();
    for (int j = tid; j < cnt; j += AT_THREADS)
      if (s_max[j] > 0) atomicMax(&w.gt_max[((long long)b * d.num_groups + g) * m + j], s_max[j]);
    __syncthreads();
  }
}

// pass 2: labels, regression targets, weights
__global__ __launch_bounds__(AT_THREADS) void k_assign_labels(pcp_anchor_assign_t d, const float *__restrict__ anchors,
                                                             const float *__restrict__ gt, int m, AssignWs w, int *__restrict__ labels,
                                                             float *__restrict__ reg_targets, float *__restrict__ reg_weights) {
  __shared__ GtBev s_bev[AT_MAX_BOXES];
  __shared__ int s_cls[AT_MAX_BOXES];
  __shared__ float s_max[AT_MAX_BOXES];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int n = d.h * d.w * d.anchors_per_loc;
  const int cnt = w.count[b];
  for (int j = tid; j < cnt; j += AT_THREADS) {
    s_bev[j] = w.bev[(long long)b * m + j];
    s_cls[j] = w.cls[(long long)b * m + j];
  }
  const int i = blockIdx.x * AT_THREADS + tid;
  const int g = i < n ? d.slot_group[i % d.anchors_per_loc] : 0;
  // every thread of the workgroup needs the max row of ITS anchor class: stage class by class
  int label = -1, arg = -1;
  float amax = 0.f;
  bool forced = false;
  int selected = 0;
  for (int gg = 0; gg < d.num_groups; ++gg) {
    __syncthreads();
    for (int j = tid; j < cnt; j += AT_THREADS) {
      const float v = __int_as_float(w.gt_max[((long long)b * d.num_groups + gg) * m + j]);
      s_max[j] = v == 0.f ? -1.f : v;                       // empty_gt_mask: a box no anchor touches forces nothing
    }
    __syncthreads();
    if (i < n && g == gg) {
      const GtBev a = aligned_bev(anchors + (long long)i * 7);
      const int want = d.group_class[g];
      for (int j = 0; j < cnt; ++j) {
        if (s_cls[j] != want) continue;
        const float v = iou_normal(a, s_bev[j]);
        if (selected == 0 || v > amax) { amax = v; arg = j; }   // argmax: first maximum
        forced = forced || (v == s_max[j]);
        ++selected;
      }
    }
  }
  if (i >= n) return;
  const long long o = (long long)b * n + i;
  float t[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (selected == 0) {
    label = 0;
  } else {
    const int cls = (int)gt[((long long)b * m + arg) * 8 + 7];    // gt_classes[anchor_to_gt_argmax]
    if (forced || amax >= d.matched[g]) label = cls;
    else if (amax < d.unmatched[g]) label = 0;
    if (label > 0) {
      // ResidualCoder.encode_torch
      const float *bx = gt + ((long long)b * m + arg) * 8;
      const float *an = anchors + (long long)i * 7;
      const float dxa = fmaxf(an[3], 1e-5f), dya = fmaxf(an[4], 1e-5f), dza = fmaxf(an[5], 1e-5f);
      const float dxg = fmaxf(bx[3], 1e-5f), dyg = fmaxf(bx[4], 1e-5f), dzg = fmaxf(bx[5], 1e-5f);
      const float diag = sqrtf(dxa * dxa + dya * dya);
      t[0] = (bx[0] - an[0]) / diag;
      t[1] = (bx[1] - an[1]) / diag;
      t[2] = (bx[2] - an[2]) / dza;
      t[3] = logf(dxg / dxa);
      t[4] = logf(dyg / dya);
      t[5] = logf(dzg / dza);
      t[6] = bx[6] - an[6];
    }
  }
  labels[o] = label;
  reg_weights[o] = label > 0 ? 1.f : 0.f;
#pragma unroll
  for (int j = 0; j < 7; ++j) reg_targets[o * 7 + j] = t[j];
}

// ---- losses ---------------------------------------------------------------------------------------------------------------
// workspace (double): [0] cls sum  [1] loc sum  [2] dir sum  [3] unused  [4 + b] positives of frame b
__device__ __forceinline__ double block_sum(double v, double *sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wv] = v;
  __syncthreads();
  double t = 0;
  if (threadIdx.x == 0) for (unsigned w = 0; w < (blockDim.x + 63) / 64; ++w) t += sh[w];
  return t;   // valid on thread 0
}

__global__ __launch_bounds__(AT_THREADS) void k_count_positives(int n, const int *__restrict__ labels, double *acc) {
  __shared__ double sh[AT_THREADS / 64];
  const int b = blockIdx.y;
  double c = 0;
  for (int i = blockIdx.x * AT_THREADS + threadIdx.x; i < n; i += gridDim.x * AT_THREADS) c += labels[(long long)b * n + i] > 0 ? 1.0 : 0.0;
  const double s = block_sum(c, sh);
  if (threadIdx.x == 0 && s != 0.0) atomicAdd(acc + 4 + b, s);
}

constexpr int AL_MAX_CLASS = 16;
constexpr int AL_MAX_BINS = 8;

// one thread per (frame, anchor): the three loss terms and every gradient channel of the anchor; slot 0 also clears the padding channels
__global__ __launch_bounds__(AT_THREADS) void k_anchor_loss(pcp_anchor_loss_t d, const float *__restrict__ head,
                                                           const float *__restrict__ anchors, const int *__restrict__ labels,
                                                           const float *__restrict__ reg_targets, float grad_scale, double *acc,
                                                           float *__restrict__ dhead) {
  __shared__ double sh[AT_THREADS / 64];
  const int b = blockIdx.y;
  const int A = d.anchors_per_loc;
  const int n = d.h * d.w * A;
  const int i = blockIdx.x * AT_THREADS + threadIdx.x;
  double l_cls = 0, l_loc = 0, l_dir = 0;
  if (i < n) {
    const int slot = i % A;
    const long long pix = (long long)b * d.h * d.w + i / A;
    const float *hp = head + pix * d.ld;
    float *gp = dhead ? dhead + pix * d.ld_d : nullptr;
    const long long o = (long long)b * n + i;
    const int label = labels[o];
    const float norm = fmaxf((float)acc[4 + b], 1.f);
    const float inv_b = 1.f / (float)d.batch;
    // ---- classification: sigmoid focal loss (alpha 0.25, gamma 2) over num_class logits; one-hot target = label (class-agnostic: 1)
    {
      const float w = label >= 0 ? 1.f / norm : 0.f;
      const int target = label > 0 ? (d.num_class == 1 ? 1 : label) : 0;
      for (int c = 0; c < d.num_class; ++c) {
        const int ch = d.ch_cls + slot * d.num_class + c;
        float gval = 0.f;
        if (w != 0.f) {
          const float x = hp[ch];
          const float tt = (target == c + 1) ? 1.f : 0.f;
          const float p = 1.f / (1.f + expf(-x));
          const float aw = tt * 0.25f + (1.f - tt) * 0.75f;
          const float pt = tt * (1.f - p) + (1.f - tt) * p;
          const float bce = fmaxf(x, 0.f) - x * tt + log1pf(expf(-fabsf(x)));
          l_cls += (double)(aw * (pt * pt) * bce * w);
          const float dpt = (1.f - 2.f * tt) * p * (1.f - p);
          gval = aw * (2.f * pt * dpt * bce + pt * pt * (p - tt)) * w * d.cls_weight * inv_b * grad_scale;
        }
        if (gp) gp[ch] = gval;
      }
    }
    // ---- localisation: smooth L1 (beta 1/9) on the code residuals, heading through sin(a - b) = sin a cos b - cos a sin b
    const bool pos = label > 0;
    const float rw = pos ? 1.f / norm : 0.f;
    const float beta = 1.f / 9.f;
    float tg6 = 0.f;
    {
      const float *tp = reg_targets + o * 7;
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const int ch = d.ch_box + slot * 7 + j;
        float gval = 0.f;
        const float tv = tp[j];
        if (j == 6) tg6 = tv;
        if (pos && tv == tv) {
          const float pv = hp[ch];
          float diff, chain = 1.f;
          if (j == 6) {
            const float sp = sinf(pv), cp = cosf(pv), st = sinf(tv), ct = cosf(tv);
            diff = sp * ct - cp * st;
            chain = cp * ct + sp * st;
          } else {
            diff = pv - tv;
          }
          diff *= d.code_weights[j];
          const float a = fabsf(diff);
          l_loc += (double)((a < beta ? 0.5f * a * a / beta : a - 0.5f * beta) * rw);
          const float ds = a < beta ? diff / beta : (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f));
          gval = ds * d.code_weights[j] * chain * rw * d.loc_weight * inv_b * grad_scale;
        }
        if (gp) gp[ch] = gval;
      }
    }
    // ---- direction classifier: cross entropy against the bin of (target heading + anchor heading - DIR_OFFSET) in [0, 2 pi)
    if (d.num_dir_bins > 0) {
      const int nb = d.num_dir_bins;
      const int ch0 = d.ch_dir + slot * nb;
      if (pos) {
        const float two_pi = 6.283185307179586f;
        const float rot_gt = tg6 + anchors[(long long)i * 7 + 6];
        const float v = rot_gt - d.dir_offset;
        const float off = v - floorf(v / two_pi + 0.f) * two_pi;
        int bin = (int)floorf(off / d.dir_period);
        bin = bin < 0 ? 0 : (bin > nb - 1 ? nb - 1 : bin);
        float mx = -INFINITY;
        for (int k = 0; k < nb; ++k) mx = fmaxf(mx, hp[ch0 + k]);
        float se = 0.f;
        for (int k = 0; k < nb; ++k) se += expf(hp[ch0 + k] - mx);
        const float lse = mx + logf(se);
        l_dir += (double)((lse - hp[ch0 + bin]) * rw);
        if (gp)
          for (int k = 0; k < nb; ++k)
            gp[ch0 + k] = (expf(hp[ch0 + k] - lse) - (k == bin ? 1.f : 0.f)) * rw * d.dir_weight * inv_b * grad_scale;
      } else if (gp) {
        for (int k = 0; k < nb; ++k) gp[ch0 + k] = 0.f;
      }
    }
    if (gp && slot == 0) {
      const int used = d.ch_dir + A * d.num_dir_bins;
      for (int ch = used; ch < d.ld_d; ++ch) gp[ch] = 0.f;
    }
  }
  double s = block_sum(l_cls, sh);
  if (threadIdx.x == 0 && s != 0.0) atomicAdd(acc + 0, s);
  s = block_sum(l_loc, sh);
  if (threadIdx.x == 0 && s != 0.0) atomicAdd(acc + 1, s);
  s = block_sum(l_dir, sh);
  if (threadIdx.x == 0 && s != 0.0) atomicAdd(acc + 2, s);
}

// losses_out: [0] rpn_loss_cls  [1] rpn_loss_loc  [2] rpn_loss_dir  [3] rpn_loss  [4] positives over the batch
__global__ void k_anchor_loss_finalize(pcp_anchor_loss_t d, const double *acc, float *losses_out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double cls = acc[0] / (double)d.batch * (double)d.cls_weight;
  const double loc = acc[1] / (double)d.batch * (double)d.loc_weight;
  const double dir = d.num_dir_bins > 0 ? acc[2] / (double)d.batch * (double)d.dir_weight : 0.0;
  double npos = 0;
  for (int b = 0; b < d.batch; ++b) npos += acc[4 + b];
  losses_out[0] = (float)cls;
  losses_out[1] = (float)loc;
  losses_out[2] = (float)dir;
  losses_out[3] = (float)(cls + loc + dir);
  losses_out[4] = (float)npos;
}

}  // namespace

extern "C" {

size_t pcp_anchor_assign_workspace_bytes(const pcp_anchor_assign_t *d, int32_t max_boxes) {
  if (!d || d->batch <= 0 || d->num_groups <= 0 || max_boxes < 0) return 0;
  return assign_ws_bytes(d->batch, d->num_groups, max_boxes > 0 ? max_boxes : 1);
}

int pcp_anchor_assign_targets(const pcp_anchor_assign_t *d, const float *anchors, const float *gt_boxes, int32_t max_boxes, void *workspace,
                              size_t workspace_bytes, int32_t *labels, float *reg_targets, float *reg_weights, void *stream) {
  if (!d || !anchors || !labels || !reg_targets || !reg_weights || !workspace) return PCP_ERR_ARG;
  if (d->batch <= 0 || d->h <= 0 || d->w <= 0 || d->anchors_per_loc <= 0 || d->anchors_per_loc > PCP_ANCHOR_MAX_SLOTS) return PCP_ERR_ARG;
  if (d->num_groups <= 0 || d->num_groups > PCP_ANCHOR_MAX_GROUPS || d->num_class <= 0) return PCP_ERR_ARG;
  if (max_boxes < 0 || max_boxes > AT_MAX_BOXES || (max_boxes > 0 && !gt_boxes)) return PCP_ERR_ARG;
  for (int s = 0; s < d->anchors_per_loc; ++s)
    if (d->slot_group[s] < 0 || d->slot_group[s] >= d->num_groups) return PCP_ERR_ARG;
  const int m = max_boxes > 0 ? max_boxes : 1;
  if (workspace_bytes < assign_ws_bytes(d->batch, d->num_groups, m)) return PCP_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  AssignWs w = carve(workspace, d->batch, d->num_groups, m);
  const int n = d->h * d->w * d->anchors_per_loc;
  if (max_boxes == 0) {
    if (pcp_zero_async(w.count, sizeof(int) * d->batch, st) != PCP_OK) return PCP_ERR_LAUNCH;
  } else {
    k_assign_prepare<<<d->batch, AT_THREADS, 0, st>>>(*d, gt_boxes, max_boxes, w);
    k_assign_gt_max<<<dim3((n + AT_THREADS - 1) / AT_THREADS, d->batch), AT_THREADS, 0, st>>>(*d, anchors, max_boxes, w);
  }
  k_assign_labels<<<dim3((n + AT_THREADS - 1) / AT_THREADS, d->batch), AT_THREADS, 0, st>>>(*d, anchors, gt_boxes, max_boxes, w, labels,
                                                                                             reg_targets, reg_weights);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

size_t pcp_anchor_loss_workspace_bytes(int32_t batch) { return batch > 0 ? sizeof(double) * (size_t)(4 + batch) : 0; }

int pcp_anchor_loss(const pcp_anchor_loss_t *d, const float *head, const float *anchors, const int32_t *labels, const float *reg_targets,
                    float grad_scale, void *workspace, size_t workspace_bytes, float *losses, float *dhead, void *stream) {
  if (!d || !head || !anchors || !labels || !reg_targets || !workspace || !losses) return PCP_ERR_ARG;
  if (d->batch <= 0 || d->h <= 0 || d->w <= 0 || d->anchors_per_loc <= 0 || d->num_class <= 0 || d->num_class > AL_MAX_CLASS) return PCP_ERR_ARG;
  if (d->num_dir_bins < 0 || d->num_dir_bins > AL_MAX_BINS) return PCP_ERR_ARG;
  const int used = d->anchors_per_loc * (d->num_class + 7 + d->num_dir_bins);
  if (d->ld < used || (dhead && d->ld_d < used)) return PCP_ERR_ARG;
  if (workspace_bytes < pcp_anchor_loss_workspace_bytes(d->batch)) return PCP_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  double *acc = static_cast<double *>(workspace);
  if (pcp_zero_async(acc, pcp_anchor_loss_workspace_bytes(d->batch), st) != PCP_OK) return PCP_ERR_LAUNCH;
  const int n = d->h * d->w * d->anchors_per_loc;
  const int blocks = (n + AT_THREADS - 1) / AT_THREADS;
  k_count_positives<<<dim3(blocks < 64 ? blocks : 64, d->batch), AT_THREADS, 0, st>>>(n, labels, acc);
  k_anchor_loss<<<dim3(blocks, d->batch), AT_THREADS, 0, st>>>(*d, head, anchors, labels, reg_targets, grad_scale, acc, dhead);
  k_anchor_loss_finalize<<<1, 64, 0, st>>>(*d, acc, losses);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

}  // extern "C"
