/*
 * pcp_hip_train.h -- C ABI of libpcp_hip.so, training half (SURVEY.md 8 row a15 + appendix C: config 5, DiscoNet).
 *
 * Same conventions as pcp_hip.h (device pointers, caller-owned workspaces, explicit stream, int status, no allocation, no
 * synchronisation).  The reference trains through torch.autograd + cuDNN; there is no native training ABI to mirror, so each
 * entry point names the autograd node(s) of the reference module it stands for.  Forward convolutions of a training step are
 * the inference entry points of pcp_hip.h called with identity folding (raw weights, zero bias, relu = 0); data gradients of
 * convolutions are the same entry points called with transposed / flipped weights.
 */
#ifndef PCP_HIP_TRAIN_H
#define PCP_HIP_TRAIN_H

#include "pcp_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------------------------
 * BatchNorm in train() mode + ReLU on (rows, c) row-major / NHWC float32 (rows = B*H*W or N' points), c % 4 == 0, c <= 1024.
 * Replaces nn.BatchNorm2d / nn.BatchNorm1d (+ nn.ReLU) forward and backward as used by
 *   pcdet/models/backbones_2d/base_bev_backbone.py:37-44,56,67 (eps 1e-3, momentum 0.01),
 *   pcdet/models/backbones_3d/vfe/dynamic_pillar_vfe.py:29,40-43, pcdet/models/dense_heads/center_head.py:26,80,
 *   pcdet/models/bev_layers/v2x_fusion_disco.py:13-16,53,60 (eps 1e-5, momentum 0.1).
 * pcp_bn_train_stats: batch mean / biased variance (float64 accumulation) -> scale = gamma * invstd, shift = beta - mean * scale,
 *   saved mean / invstd, running stats updated in place with the unbiased variance (NULL, NULL: not tracked).
 * pcp_scale_shift_act: out = act(x * scale + shift) (one fma per element -- the same expression the backward mask recomputes).
 * pcp_bn_act_backward: given dout = dL/d relu(bn(x)): dgamma, dbeta (written or accumulated) and
 *   dx = scale * (dz - mean(dz) - xhat * mean(dz * xhat)), dz = dout * [bn(x) > 0]; dx may alias dout.
 * workspace: pcp_bn_workspace_bytes(c) bytes.
 * ------------------------------------------------------------------------------------------------------------------ */
size_t pcp_bn_workspace_bytes(int32_t c);
int pcp_bn_train_stats(const float *x, int64_t rows, int32_t c, int32_t ld, const float *gamma, const float *beta, float eps,
                       float momentum, float *running_mean, float *running_var, void *workspace, float *scale, float *shift,
                       float *mean, float *invstd, void *stream);
int pcp_scale_shift_act(const float *x, int64_t rows, int32_t c, int32_t ld_x, const float *scale, const float *shift, int32_t relu,
                        float *out, int32_t ld_out, void *stream);
int pcp_bn_act_backward(const float *dout, int32_t ld_dout, const float *x, int32_t ld_x, int64_t rows, int32_t c, const float *scale,
                        const float *shift, const float *mean, const float *invstd, int32_t relu, void *workspace, float *dgamma,
                        float *dbeta, int32_t accumulate, float *dx, int32_t ld_dx, void *stream);
/* Cross-rank BatchNorm (nn.SyncBatchNorm: reference tools/train.py:37,128-129 `--sync_bn`): the two calls above split where the ranks
 * exchange.  sums = 2 * c float64 on the device ([sum x | sum x^2] forward, [sum dz | sum dz * xhat] backward); the host all-reduces them
 * (and the row count) over the ranks between the two halves.  Forward statistics then come from the global sums; backward dgamma / dbeta
 * from the LOCAL sums (the gradient all-reduce combines them, as DDP does) and the mean terms of dx from the global ones. */
int pcp_bn_train_sums(const float *x, int64_t rows, int32_t c, int32_t ld, void *workspace, double *sums, void *stream);
int pcp_bn_train_stats_from_sums(const double *sums, int64_t total_rows, int32_t c, const float *gamma, const float *beta, float eps,
                                 float momentum, float *running_mean, float *running_var, float *scale, float *shift, float *mean,
                                 float *invstd, void *stream);
int pcp_bn_bwd_sums(const float *dout, int32_t ld_dout, const float *x, int32_t ld_x, int64_t rows, int32_t c, const float *scale,
                    const float *shift, const float *mean, const float *invstd, int32_t relu, void *workspace, double *sums, void *stream);
int pcp_bn_bwd_apply_from_sums(const float *dout, int32_t ld_dout, const float *x, int32_t ld_x, int64_t rows, int32_t c, const float *scale,
                               const float *shift, const float *mean, const float *invstd, int32_t relu, const double *local_sums,
                               const double *global_sums, int64_t total_rows, void *workspace, float *dgamma, float *dbeta,
                               int32_t accumulate, float *dx, int32_t ld_dx, void *stream);
/* out[c] (+)= sum over rows (bias gradients); workspace as above */
int pcp_colsum(const float *x, int64_t rows, int32_t c, int32_t ld, void *workspace, float *out, int32_t accumulate, void *stream);
/* dst[r, :c] += alpha * src[r, :c] (gradient fan-in) */
int pcp_accumulate(float *dst, int32_t ld_dst, const float *src, int32_t ld_src, int64_t rows, int32_t c, float alpha, void *stream);
/* out (B, 2h, 2w, c): in at the even pixels, zero elsewhere.  The data gradient of a stride-2 3x3 conv (ZeroPad2d(1) + Conv2d s2,
 * base_bev_backbone.py:33-36) is the stride-1 conv of this map with the flipped, transposed weights. */
int pcp_dilate2x(const float *in, int32_t batch, int32_t h, int32_t w, int32_t c, int32_t ld_in, float *out, int32_t ld_out, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Weight gradients (cuDNN backward-filter of nn.Conv2d / nn.ConvTranspose2d), fp32 MFMA with the pixel as contraction index;
 * split-K partials in the workspace are reduced in a fixed order (deterministic).
 * pcp_conv3x3_wgrad: desc as for pcp_conv3x3 (ld_in = pixel stride of x, ld_out = pixel stride of dy; cout_pad / relu ignored);
 *   dw is PyTorch's (cout, cin, 3, 3) float32, written or accumulated.  cin % 4 == 0, cout % 4 == 0.
 * pcp_pointwise_wgrad: out[n][k] (+)= sum_r a[map_a(r)][n] * b[map_b(r)][k] with n < a.channels, k < b.channels; a row map is the
 *   identity (lattice = 0) or sends r = (b, y, x) on a (grid_h, grid_w) grid to pixel (b, 2y + ky, 2x + kx) of the (2 grid_h,
 *   2 grid_w) map (the taps of Conv2d k2 s2 / ConvTranspose2d k2 s2, base_bev_backbone.py:48-69).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct {
  const float *ptr;
  int32_t ld, channels;
  int32_t lattice, grid_h, grid_w, ky, kx;
} pcp_rowmap_t;

size_t pcp_conv3x3_wgrad_workspace_bytes(const pcp_conv3x3_t *desc);
int pcp_conv3x3_wgrad(const pcp_conv3x3_t *desc, const float *x, const float *dy, void *workspace, size_t workspace_bytes, float *dw,
                      int32_t accumulate, void *stream);
size_t pcp_pointwise_wgrad_workspace_bytes(int64_t rows, int32_t n, int32_t k);
int pcp_pointwise_wgrad(const pcp_rowmap_t *a, const pcp_rowmap_t *b, int64_t rows, void *workspace, size_t workspace_bytes,
                        float *out, int32_t ld_out, int32_t accumulate, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * a15  CenterHead training: target assignment, focal + L1 losses and their gradients, DiscoNet distillation loss.
 * Replaces: pcdet/models/dense_heads/center_head.py:104-164,166-268 (assign_targets: per-frame .cpu() + python loop over
 *           boxes), pcdet/models/model_utils/centernet_utils.py:8-68 (gaussian_radius, gaussian2D, draw_gaussian_to_heatmap),
 *           center_head.py:270-300 (get_loss), pcdet/utils/loss_utils.py:264-375 (neg_loss_cornernet, _reg_loss,
 *           RegLossCenterNet) and pcdet/models/bev_layers/v2x_fusion_disco.py:119-123 (loss_distill), plus their autograd.
 * One head (all five configs); classes of gt_boxes[..., 7] in 1..num_class, 0 = padding row (dataset.py collate_batch).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t batch, h, w;            /* feature map */
  int32_t num_class;
  int32_t k;                      /* NUM_MAX_OBJS */
  float stride;                   /* FEATURE_MAP_STRIDE */
  float voxel_x, voxel_y, min_x, min_y;
  float gaussian_overlap;         /* GAUSSIAN_OVERLAP */
  int32_t min_radius;             /* MIN_RADIUS */
} pcp_target_t;

/* gt_boxes (B, max_boxes, 8) [x,y,z,dx,dy,dz,heading,class] (max_boxes <= 1024).  Outputs: heatmap (B, H, W, num_class) NHWC,
 * target_boxes (B, k, 8) = [dx, dy, z, log dims(3), cos, sin], inds (B, k) int32 flat y*W+x, mask (B, k) int32. */
int pcp_centerhead_targets(const pcp_target_t *desc, const float *gt_boxes, int32_t max_boxes, float *heatmap, float *target_boxes,
                           int32_t *inds, int32_t *mask, void *stream);

typedef struct {
  int32_t batch, h, w;
  int32_t ld;                     /* pixel stride of the head-map buffer */
  int32_t ld_d;                   /* pixel stride of the gradient buffer (channel numbering identical to the head buffer) */
  int32_t num_class, ch_hm;       /* heat-map logits at channels [ch_hm, ch_hm + num_class) */
  int32_t reg_ch[8];              /* channel of each regression code in HEAD_ORDER: center(2), center_z, dim(3), rot(2) */
  int32_t k;
  float cls_weight, loc_weight, code_weights[8];
} pcp_headloss_t;

size_t pcp_loss_workspace_bytes(void);
/* losses (4,) float32 device: [hm_loss * cls_weight, loc_loss * loc_weight, their sum, num_pos].  dhead (B, H, W, ld_d) or NULL:
 * every channel written (zero where no loss term reads the map), scaled by grad_scale. */
int pcp_centerhead_loss(const pcp_headloss_t *desc, const float *head, const float *heatmap, const float *target_boxes,
                        const int32_t *inds, const int32_t *mask, float grad_scale, void *workspace, float *losses, float *dhead,
                        void *stream);
/* loss (1,) = weight * mean smooth_l1(softmax_c(fused) - softmax_c(early)); dfused (pixels, ld_d) written or accumulated (NULL: none);
 * c <= 512.  Shares the workspace of pcp_centerhead_loss. */
int pcp_distill_loss(const float *fused, int32_t ld_f, const float *early, int32_t ld_e, int64_t pixels, int32_t c, float weight,
                     float grad_scale, void *workspace, float *loss, float *dfused, int32_t ld_d, int32_t accumulate, void *stream);
/* HunterJr teacher-BEV term (hunter_jr.py:352-365): loss (1,) = mean over the pixels with ||teacher[p, :]||_2 > thresh of
 * sum_c smooth_l1(fused[p, c] - teacher[p, c]) (nan when no pixel qualifies, as torch's mean of an empty selection).  Value only: the
 * reference keeps it in forward_return_dict['loss_dtl_bev_img'] and never adds it to the training loss (hunter_jr.py:490-494).
 * workspace: pcp_loss_workspace_bytes() bytes. */
int pcp_masked_smooth_l1_rows(const float *fused, int32_t ld_f, const float *teacher, int32_t ld_t, int64_t pixels, int32_t c, float thresh,
                              void *workspace, float *loss, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Training-mode PillarFeatureNet (BatchNorm1d with batch statistics splits the fused inference kernel at its two global
 * reductions).  Per-point tensors are in BUCKET ORDER (the counting sort pcp_voxelize leaves in its workspace; slot s, 0 <= s < N');
 * the Linear layers run on pcp_pointwise, their gradients on pcp_pointwise / pcp_pointwise_wgrad, BatchNorm on pcp_bn_*.
 * Replaces dynamic_pillar_vfe.py:110-126 (features), :35-46 (scatter_max + concat) and their autograd (scatter_max backward
 * routes to one arg-max row per (pillar, channel); ties -> first row in bucket order), pointpillar_scatter.py:14-37.
 *   features      fbuf (N', FW) = [raw(num_raw), f_cluster(3), f_center(3), 0...], FW = 16 if num_raw + 6 <= 16 else 32 (num_raw 3, 4, 5, 11);
 *                 slot_pillar (N',) int32 pillar rank of each slot
 *   mid           in1 (N', 64) = [relu(x0 * scale0 + shift0), per-pillar max of it]; arg0 (P, 32) int32 arg-max slot
 *   out           pillar_features (P, 64) (may be NULL), arg1 (P, 64), canvas (B, ny, nx, 64) rows (may be NULL)
 *   route_out     dz1 (kept_rows, 64) = 0 except dz1[arg1[p, c], c] = dcanvas[cell(p), c]  (or dpillar[p, c]; exactly one non-NULL)
 *   route_mid     da0 (N', 32) = din1[:, :32] + [slot == arg0] * sum over the pillar of din1[:, 32:]
 * ------------------------------------------------------------------------------------------------------------------ */
int pcp_pfn_train_features(const float *points, int64_t n, int32_t row_stride, int32_t num_raw, const pcp_grid_t *grid,
                           const void *vox_workspace, float *fbuf, int32_t *slot_pillar, void *stream);
int pcp_pfn_train_mid(const pcp_grid_t *grid, const void *vox_workspace, int64_t n, const float *x0, const float *scale0,
                      const float *shift0, float *in1, int32_t *arg0, void *stream);
int pcp_pfn_train_out(const pcp_grid_t *grid, const void *vox_workspace, int64_t n, const float *x1, const float *scale1,
                      const float *shift1, float *pillar_features, int32_t *arg1, float *canvas, void *stream);
int pcp_pfn_train_route_out_grad(const pcp_grid_t *grid, const void *vox_workspace, int64_t n, int64_t kept_rows, const float *dcanvas,
                                 const float *dpillar, const int32_t *arg1, float *dz1, void *stream);
int pcp_pfn_train_route_mid_grad(const pcp_grid_t *grid, const void *vox_workspace, int64_t n, const float *din1, const int32_t *arg0,
                                 float *da0, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * DiscoNet mid fusion, training: last weightor stage and the backward of softmax-over-agents + weighted sum.
 * Replaces the autograd graph of pcdet/models/bev_layers/v2x_fusion_disco.py:22-24 and :109-115.  Agent maps are constants
 * (transform_bev_img is @torch.no_grad, :29): only map 0 (ego) receives a gradient.  n_agents <= 8, c <= 256.
 * pcp_disco_weight_logits: logits[p, a] = relu(<h2_a[p, 0:16], w4> + b4).
 * pcp_disco_fuse_backward: dmap0 = softmax_a(logits)[0] * dfused;  dh2_a[p, :] = dlogit_a * [logit_a > 0] * w4;
 *   dw4 (16), db4 (1) written or accumulated.  *_host arrays are HOST arrays of device pointers.
 * ------------------------------------------------------------------------------------------------------------------ */
int pcp_disco_weight_logits(const float *const *h2_host, int32_t n_agents, int32_t ld_h, const float *w4, const float *b4,
                            int64_t pixels, float *logits, int32_t ld_w, void *stream);
size_t pcp_disco_fuse_backward_workspace_bytes(void);
int pcp_disco_fuse_backward(const float *const *maps_host, int32_t n_agents, int32_t ld_map, int32_t c, const float *logits, int32_t ld_w,
                            const float *dfused, int32_t ld_df, const float *const *h2_host, int32_t ld_h, const float *w4,
                            int64_t pixels, float *dmap0, int32_t ld_dm, float *const *dh2_host, void *workspace, float *dw4, float *db4,
                            int32_t accumulate, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * a16  AnchorHeadSingle training (MODEL.NAME PointPillar): target assignment, the three losses and dL/d(head maps).
 * Replaces: pcdet/models/dense_heads/target_assigner/axis_aligned_target_assigner.py:37-210 (assign_targets: python loop over frames x
 *           anchor classes, an anchors x boxes IoU matrix and .nonzero() host syncs per pair), pcdet/utils/box_utils.py:291-340
 *           (boxes3d_nearest_bev_iou), pcdet/utils/box_coder_utils.py:13-44 (ResidualCoder.encode_torch),
 *           pcdet/models/dense_heads/anchor_head_template.py:99-216 (get_cls_layer_loss, add_sin_difference, get_direction_target,
 *           get_box_reg_layer_loss, get_loss), pcdet/utils/loss_utils.py:9-148,180-208 (SigmoidFocalClassificationLoss,
 *           WeightedSmoothL1Loss, WeightedCrossEntropyLoss) and their autograd.
 * Covers POS_FRACTION < 0, MATCH_HEIGHT False, NORM_BY_NUM_EXAMPLES False, ResidualCoder without sin/cos (every anchor YAML of the
 * reference).  Anchors are flat in the order of torch.cat(anchors, dim=-3).view(-1, 7): index = (y * W + x) * A + slot.
 * ------------------------------------------------------------------------------------------------------------------ */
#define PCP_ANCHOR_MAX_SLOTS 32
#define PCP_ANCHOR_MAX_GROUPS 8
typedef struct {
  int32_t batch, h, w;                          /* feature map */
  int32_t anchors_per_loc;                      /* A */
  int32_t num_class;                            /* len(CLASS_NAMES); box class c names CLASS_NAMES[c - 1] (0 wraps to the last, like numpy) */
  int32_t num_groups;                           /* ANCHOR_GENERATOR_CONFIG entries (anchor classes) */
  int32_t slot_group[PCP_ANCHOR_MAX_SLOTS];     /* anchor class of each per-location slot */
  int32_t group_class[PCP_ANCHOR_MAX_GROUPS];   /* 0-based index of the anchor class's class_name in CLASS_NAMES (-1: not a detected class) */
  float matched[PCP_ANCHOR_MAX_GROUPS], unmatched[PCP_ANCHOR_MAX_GROUPS];
} pcp_anchor_assign_t;

size_t pcp_anchor_assign_workspace_bytes(const pcp_anchor_assign_t *desc, int32_t max_boxes);
/* anchors (H*W*A, 7); gt_boxes (B, max_boxes, 8) [x,y,z,dx,dy,dz,heading,class], max_boxes <= 1024 (0: no boxes at all).
 * Outputs: labels (B, N) int32 (-1 ignored, 0 background, > 0 class), reg_targets (B, N, 7), reg_weights (B, N). */
int pcp_anchor_assign_targets(const pcp_anchor_assign_t *desc, const float *anchors, const float *gt_boxes, int32_t max_boxes,
                              void *workspace, size_t workspace_bytes, int32_t *labels, float *reg_targets, float *reg_weights,
                              void *stream);

typedef struct {
  int32_t batch, h, w;
  int32_t ld, ld_d;                             /* pixel stride of the head buffer / of the gradient buffer (same channel numbering) */
  int32_t anchors_per_loc, num_class, num_dir_bins;     /* num_dir_bins 0: no direction classifier */
  int32_t ch_cls, ch_box, ch_dir;               /* first channel of conv_cls / conv_box / conv_dir_cls outputs */
  float dir_offset, dir_period;                 /* DIR_OFFSET, 2 pi / NUM_DIR_BINS */
  float cls_weight, loc_weight, dir_weight, code_weights[7];
} pcp_anchor_loss_t;

size_t pcp_anchor_loss_workspace_bytes(int32_t batch);
/* losses (5,) float32 device: [rpn_loss_cls, rpn_loss_loc, rpn_loss_dir, rpn_loss, positives].  dhead (B, H, W, ld_d) or NULL: every
 * channel written (padding channels zero), scaled by grad_scale. */
int pcp_anchor_loss(const pcp_anchor_loss_t *desc, const float *head, const float *anchors, const int32_t *labels,
                    const float *reg_targets, float grad_scale, void *workspace, size_t workspace_bytes, float *losses, float *dhead,
                    void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * a17  HunterJr training branch (configs 1 / 2).
 * Replaces: pcdet/models/bev_layers/hunter_jr.py:165-196 (_build_meta: two torch.unique + scatter_max / scatter_min), :42-76
 *           (HunterObjectHead: scatter_mean, three scatter_max, cat), :198-260 (assign_target), :106-113 (get_loss_distill), :401-495
 *           (get_training_loss), pcdet/models/loss_fnc/pcaccum_ce_lovasz_loss.py:20-71, lovasz_softmax.py:56-95,
 *           hunter_toolbox.py:42-62 (quat2mat), :161-184 (remove_gt_boxes_outside_range), :187-219 (hard_mining_regression_loss), and the
 *           autograd of hunter_toolbox.py:8-39 (bilinear sampling incl. its position gradient), :65-91 (bev_scatter), hunter_jr.py:281-285.
 * Rows of `points` are [frame, x, y, z, ..., sweep, instance]; foreground = instance > -1 (hunter_jr.py:323).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t batch, max_inst, num_sweeps;          /* key = (frame * max_inst + instance) * num_sweeps + sweep; max_inst = gt_boxes.shape[1] */
  int32_t sweep_col, inst_col;                  /* columns of the point rows */
} pcp_hunter_meta_t;

size_t pcp_hunter_meta_workspace_bytes(const pcp_hunter_meta_t *desc, int64_t n);
/* fg_idx / fg_local (n): rows of the foreground points in ascending order and their local index (locals2fg);
 * local_key / local_inst (batch*max_inst*num_sweeps): locals_bis / inst2locals; inst_key / inst_first / inst_last (batch*max_inst):
 * instance_bi / indices_locals_min_sweep / indices_locals_max_sweep; counts (4) int32 = [n_fg, n_local, n_inst, rows with a key outside
 * the table (treated as background)]. */
int pcp_hunter_meta(const pcp_hunter_meta_t *desc, const float *points, int64_t n, int32_t row_stride, void *workspace, size_t workspace_bytes,
                    int32_t *fg_idx, int32_t *fg_local, int32_t *local_key, int32_t *local_inst, int32_t *inst_key, int32_t *inst_first,
                    int32_t *inst_last, int32_t *counts, void *stream);

/* torch_scatter.scatter_max(src[row_index], seg): out (n_seg, c) and the arg-max row (first maximal row) per (segment, channel);
 * backward: dsrc[row_index[arg[s, ch]], ch] += dout[s, ch]. */
int pcp_segment_max(const float *src, int32_t ld_src, const int32_t *row_index, int64_t rows, const int32_t *seg, int64_t n_seg, int32_t c,
                    float *out, int32_t ld_out, int32_t *arg, void *stream);
int pcp_segment_max_backward(const float *dout, int32_t ld_dout, const int32_t *arg, int64_t n_seg, int32_t c, const int32_t *row_index,
                             float *dsrc, int32_t ld_dsrc, void *stream);
/* dst[row_index[r], :c] += src[r, :c]  (row_index injective) */
int pcp_rows_scatter_add(const float *src, int32_t ld_src, const int32_t *row_index, int64_t rows, int32_t c, float *dst, int32_t ld_dst,
                         void *stream);
/* centroid (n_local, 3) = scatter_mean(fg xyz); centered (n_fg, ld) = [xyz - centroid[local], 0 ...]; workspace >= 32 * n_local bytes */
int pcp_hunter_local_centroids(const float *points, int32_t row_stride, const int32_t *fg_idx, const int32_t *fg_local, int32_t n_fg,
                               int32_t n_local, void *workspace, size_t workspace_bytes, float *centroid, float *centered, int32_t ld_centered,
                               void *stream);
/* out (n_local, ld) = [lf0 | gf[inst] | centroid | centroid[last local of inst] | 0 ...]   (hunter_jr.py:64-69) and its backward */
int pcp_hunter_object_cat(const float *lf0, const float *gf, const float *centroid, const int32_t *local_inst, const int32_t *inst_last,
                          int32_t n_local, int32_t c, float *out, int32_t ld_out, void *stream);
int pcp_hunter_object_cat_backward(const float *dcat, int32_t ld, const int32_t *inst_first, const int32_t *inst_last, int32_t n_local,
                                   int32_t n_inst, int32_t c, float *dlf0, float *dgf, void *stream);

typedef struct {
  int64_t n;
  int32_t stride, n_fg, n_local, n_inst, c;
  int32_t batch, max_inst, num_sweeps;
  const float *points;                          /* rows BEFORE the in-place correction */
  const float *gt_boxes;                        /* (batch, max_inst, 8) */
  const float *instances_tf;                    /* (batch, max_inst, num_sweeps, 3, 4) */
  const int32_t *fg_idx, *fg_local, *local_key, *local_inst, *inst_key;
  const float *head;                            /* (n, ld_head): [cls(3) | flow(3) | embedding(2)] */
  int32_t ld_head;
  const float *local_feat;                      /* (n, ld): point head's local feature */
  int32_t ld_local_feat;
  const float *locals_feat;                     /* (n_local, ld): object head's local feature */
  int32_t ld_locals_feat;
  const float *locals_tf;                       /* (n_local, ld): [t(3) | quaternion xyzw(4)] */
  int32_t ld_locals_tf;
  float coef_fg, coef_locals;                   /* LOSS_HARD_MINING_STATIC_FG_COEF / _LOCALS_COEF */
  float grad_scale;
  float *dhead;                                 /* (n, ld_dhead), every channel written */
  int32_t ld_dhead;
  float *dlocal_feat_fg;                        /* (n_fg, c): gradient of local_feat at the foreground rows */
  float *dlocals_feat;                          /* (n_local, c) */
  float *dlocals_tf;                            /* (n_local, ld_dlocals_tf), every channel written */
  int32_t ld_dlocals_tf;
  float *losses;                                /* (8): l_points_cls, l_points_embed, l_fg_offset, l_locals_transl, l_locals_rot, l_recon,
                                                   l_dtl_locals_feat, their sum */
  int32_t *labels;                              /* (n): point class targets 0 background / 1 static / 2 moving foreground */
  float *tgt_embedding, *tgt_offset;            /* optional (n_fg, 2) / (n_fg, 3) */
} pcp_hunter_loss_t;

size_t pcp_hunter_loss_workspace_bytes(int64_t n, int32_t n_fg, int32_t n_local, int32_t c);
int pcp_hunter_losses(const pcp_hunter_loss_t *desc, void *workspace, size_t workspace_bytes, void *stream);

/* fused = map0 * w0 + map1 * w1, (w0, w1) = softmax(logits[:, 0:2]); cat rows hold [map0 (c) | map1 (c)].  dcat (pixels, ld_dcat) and
 * dlogits (pixels, ld_dlogits; channels >= 2 zeroed) are overwritten. */
int pcp_softmax_fuse2_backward(const float *dfused, int32_t ld_df, const float *cat, int32_t ld_cat, const float *logits, int32_t ld_logits,
                               int64_t pixels, int32_t c, float *dcat, int32_t ld_dcat, float *dlogits, int32_t ld_dlogits, void *stream);
/* backward of pcp_bev_scatter_mean through the workspace its forward left: rows with dyn_mask go to dfeat_dyn (overwritten), the others are
 * ADDED to dfeat_acc */
int pcp_bev_scatter_mean_backward(const void *scatter_workspace, int32_t batch, int32_t h, int32_t w, int64_t n, const float *dmap,
                                  int32_t ld_dmap, int32_t c, const uint8_t *dyn_mask, float *dfeat_acc, int32_t ld_acc, float *dfeat_dyn,
                                  int32_t ld_dyn, void *stream);
/* backward of pcp_bev_sample_bilinear: dbev += (atomic); with bev != NULL also dxyz[i, 0..1] += d loss / d (x, y) of point i */
int pcp_bev_sample_bilinear_backward(const float *dfeat, int32_t ld_dfeat, const uint8_t *row_mask, const float *points, int64_t n,
                                     int32_t row_stride, const float *bev, int32_t ld_bev, int32_t batch, int32_t h, int32_t w, int32_t c,
                                     float min_x, float min_y, float pix_x, float pix_y, float *dbev, int32_t ld_dbev, float *dxyz,
                                     int32_t ld_dxyz, void *stream);
/* remove_gt_boxes_outside_range: rows whose centre lies in [range[0:3], range[3:6]) keep their order, the rest of (batch, max_boxes, 8) is
 * zero (the reference additionally shrinks max_boxes to the largest kept count; zero rows are padding to every consumer) */
int pcp_filter_gt_boxes(const float *gt_boxes, int32_t batch, int32_t max_boxes, const float *range6_host, float *out, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Optimizer step on flat buffers: global-norm clipping + Adam with decoupled weight decay.
 * Replaces tools/train_utils/train_utils.py:57-58 (clip_grad_norm_ + optimizer.step()) and
 *          tools/train_utils/optimization/fastai_optim.py:104-122 (p.mul_(1 - wd * lr); torch.optim.Adam.step, betas (mom, 0.99)).
 * pcp_grad_sqnorm: *sqnorm (double, device) (+)= sum g^2.
 * pcp_adam_step (torch.optim.Adam arithmetic, amsgrad off, bias-corrected):
 *   g = grad * grad_scale * min(1, max_norm / (sqrt(*sqnorm) * |grad_scale| + 1e-6))   (sqnorm NULL: no clipping)
 *   p *= 1 - weight_decay * lr;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr / (1-b1^step) * m / (sqrt(v) / sqrt(1-b2^step) + eps)
 * ------------------------------------------------------------------------------------------------------------------ */
int pcp_grad_sqnorm(const float *grad, int64_t n, double *sqnorm, int32_t accumulate, void *stream);
int pcp_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int64_t step, float max_norm, const double *sqnorm, float grad_scale, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Per-step weight repacking of a 3x3 conv (the optimizer rewrites the weights every iteration): w is PyTorch's (cout, cin, 3, 3);
 * transpose = 0 packs the forward conv, transpose = 1 the conv that computes its DATA gradient (channel roles swapped, taps
 * flipped: the autograd "conv_transpose" of nn.Conv2d).  direct: [I/16][9][O_pad][16]; winograd: U = G g G^T as [I/8][16][O_pad][8]
 * (layouts of pcp_conv3x3 / pcp_conv3x3_winograd); split_bf16: the hi / lo bf16 layout of pcp_conv3x3_bf16x3 (opt-in arithmetic).
 * Any of the three may be NULL.  (I, O) = (cin, cout) or swapped.
 * ------------------------------------------------------------------------------------------------------------------ */
int pcp_pack_conv3x3(const float *w, int32_t cout, int32_t cin, int32_t transpose, float *direct, int32_t direct_cout_pad,
                     float *winograd, int32_t winograd_cout_pad, void *split_bf16, int32_t split_cout_pad, void *stream);

/* The same per-step repacking into the weight forms of the two fused Winograd F(4x4,3x3) kernels: U = G g G^T (float64, one rounding --
 * the arithmetic of pcp_amd/pack.py::pack_conv3x3_winograd4f) as u4f [I/8][36][cout_pad][8] (pcp_conv3x3_winograd4f) and / or u4h
 * [I/8][36][cout_pad/64][64][8] (pcp_conv3x3_winograd4h); either may be NULL.  I % 8 == 0, cout_pad % 64 == 0, cout_pad >= O. */
int pcp_pack_conv3x3_winograd4(const float *w, int32_t cout, int32_t cin, int32_t transpose, float *u4f, float *u4h, int32_t cout_pad,
                               void *stream);

/* All 3x3 layers of a training step repacked by ONE launch.  A job is one (layer, direction): the arguments of pcp_pack_conv3x3 (without the
 * split-bf16 form) and of pcp_pack_conv3x3_winograd4; any destination may be NULL.  block_start = first block of the job in the grouped
 * launch (jobs in ascending order: job i covers blocks [block_start_i, block_start_i + pcp_pack_conv3x3_group_blocks(job_i))), total_blocks
 * their sum.  `jobs_device` is the table in DEVICE memory (it holds device pointers; the caller keeps it and the buffers alive).
 * _group_blocks: blocks of 256 threads the job needs, -1 for an invalid job (host-side helper, no launch). */
typedef struct pcp_pack_job {
  const float *w;              /* (cout, cin, 3, 3) */
  int32_t cout, cin, transpose;
  int32_t direct_cout_pad;
  float *direct;
  float *winograd;
  int32_t winograd_cout_pad;
  int32_t f4_cout_pad;
  float *u4f;
  float *u4h;
  int32_t block_start;
  int32_t reserved;
} pcp_pack_job_t;
int pcp_pack_conv3x3_group_blocks(const pcp_pack_job_t *job);
int pcp_pack_conv3x3_group(const pcp_pack_job_t *jobs_device, int32_t n_jobs, int32_t total_blocks, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PCP_HIP_TRAIN_H */
