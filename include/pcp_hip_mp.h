/* libpcp_hip.so -- mixed-precision TRAINING entry points (BASELINE.json config 5: "v2x_pointpillar_disco.yaml ... training loop bf16").
 *
 * What "bf16 training loop" means here (the torch.cuda.amp-style recipe the reference's GPU loop would run under autocast;
 * tools/train_utils/train_utils.py:24-75 is the loop, SURVEY appendix C the contract):
 *   - activations and activation gradients of the conv stacks are STORED as bf16 between layers (NHWC, 2 bytes per channel);
 *   - every convolution -- forward, data gradient AND weight gradient -- multiplies bf16 operands on v_mfma_f32_32x32x16_bf16 and
 *     accumulates in fp32;
 *   - master weights, weight gradients, BatchNorm statistics / parameters, losses and the optimizer stay fp32 (sums float64);
 *   - the three frozen BEV makers (teachers) run the same bf16 kernels with BatchNorm folded into bf16 weights + fp32 bias.
 * Opt-in (PCP_CONV_ALGO=bf16 / bench.py --train --conv-algo bf16); never used where parity with the reference's fp32 results is claimed.
 *
 * Every tensor argument is (pointer, storage type): PCP_DT_F32 rows of float, PCP_DT_BF16 rows of bfloat16 (round to nearest even on
 * store).  Rows are NHWC pixels / row-major rows with a leading dimension in ELEMENTS; four-channel groups must be 16-byte (float) /
 * 8-byte (bf16) aligned.
 */
#ifndef PCP_HIP_MP_H
#define PCP_HIP_MP_H

#include <stddef.h>
#include <stdint.h>

#include "pcp_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PCP_DT_F32 0
#define PCP_DT_BF16 1

/* ------------------------------------------------------------------------------------------------------------------
 * Elementwise / reduction kernels of the training path with a storage type per tensor.  Same semantics, same fp32 / float64
 * arithmetic and the same workspace (pcp_bn_workspace_bytes) as the fp32 entry points of pcp_hip_train.h they generalise:
 * pcp_bn_train_stats, pcp_scale_shift_act, pcp_bn_act_backward (dx may alias dout when both have one storage type), pcp_bn_train_sums,
 * pcp_bn_bwd_sums, pcp_bn_bwd_apply_from_sums, pcp_colsum, pcp_accumulate, pcp_dilate2x.
 * Replace nn.BatchNorm2d (train mode) + nn.ReLU and their autograd nodes under autocast:
 * pcdet/models/backbones_2d/base_bev_backbone.py:37-44,56,67; dense_heads/center_head.py:26,80; bev_layers/v2x_fusion_disco.py:13-16,53,60.
 * ------------------------------------------------------------------------------------------------------------------ */
int pcp_mp_bn_train_stats(const void *x, int32_t x_dtype, int64_t rows, int32_t c, int32_t ld, const float *gamma, const float *beta, float eps,
                          float momentum, float *running_mean, float *running_var, void *workspace, float *scale, float *shift,
                          float *mean, float *invstd, void *stream);
int pcp_mp_scale_shift_act(const void *x, int32_t x_dtype, int64_t rows, int32_t c, int32_t ld_x, const float *scale, const float *shift,
                           int32_t relu, void *out, int32_t out_dtype, int32_t ld_out, void *stream);
int pcp_mp_bn_act_backward(const void *dout, int32_t dout_dtype, int32_t ld_dout, const void *x, int32_t x_dtype, int32_t ld_x, int64_t rows,
                           int32_t c, const float *scale, const float *shift, const float *mean, const float *invstd, int32_t relu,
                           void *workspace, float *dgamma, float *dbeta, int32_t accumulate, void *dx, int32_t dx_dtype, int32_t ld_dx,
                           void *stream);
int pcp_mp_bn_train_sums(const void *x, int32_t x_dtype, int64_t rows, int32_t c, int32_t ld, void *workspace, double *sums, void *stream);
int pcp_mp_bn_bwd_sums(const void *dout, int32_t dout_dtype, int32_t ld_dout, const void *x, int32_t x_dtype, int32_t ld_x, int64_t rows,
                       int32_t c, const float *scale, const float *shift, const float *mean, const float *invstd, int32_t relu,
                       void *workspace, double *sums, void *stream);
int pcp_mp_bn_bwd_apply_from_sums(const void *dout, int32_t dout_dtype, int32_t ld_dout, const void *x, int32_t x_dtype, int32_t ld_x,
                                  int64_t rows, int32_t c, const float *scale, const float *shift, const float *mean, const float *invstd,
                                  int32_t relu, const double *local_sums, const double *global_sums, int64_t total_rows, void *workspace,
                                  float *dgamma, float *dbeta, int32_t accumulate, void *dx, int32_t dx_dtype, int32_t ld_dx, void *stream);
int pcp_mp_colsum(const void *x, int32_t x_dtype, int64_t rows, int32_t c, int32_t ld, void *workspace, float *out, int32_t accumulate,
                  void *stream);
int pcp_mp_accumulate(void *dst, int32_t dst_dtype, int32_t ld_dst, const void *src, int32_t src_dtype, int32_t ld_src, int64_t rows,
                      int32_t c, float alpha, void *stream);
int pcp_mp_dilate2x(const void *in, int32_t dtype, int32_t batch, int32_t h, int32_t w, int32_t c, int32_t ld_in, void *out, int32_t ld_out,
                    void *stream);

/* The PFN training kernels of pcp_hip_train.h with storage types.  (a) The BEV canvas and its gradient: in the bf16 loop the canvas is
 * written as bf16 by the PFN's last kernel (it is the bf16 input of the first backbone layer and of that layer's weight gradient) and the
 * canvas gradient the backbone hands back is read as bf16 -- no fp32 canvas, no casts.  (b) The per-point 64-channel rows around the
 * second PFN Linear (dynamic_pillar_vfe.py:35-46 under autocast): in1 = [relu(bn(x0)) | pillar max] is written, x1 = Linear(in1) read,
 * dz1 = dL/dx1 written and din1 = dL/din1 read in the dtype given, so that Linear and its two gradient GEMMs run on pcp_mp_pointwise /
 * pcp_mp_pointwise_wgrad and every pass over those rows moves half the bytes.  The first Linear (point coordinates) stays float32. */
int pcp_mp_pfn_train_mid(const pcp_grid_t *grid, const void *vox_workspace, int64_t n, const float *x0, const float *scale0,
                         const float *shift0, void *in1, int32_t in1_dtype, int32_t *arg0, void *stream);
int pcp_mp_pfn_train_out(const pcp_grid_t *grid, const void *vox_workspace, int64_t n, const void *x1, int32_t x1_dtype, const float *scale1,
                         const float *shift1, float *pillar_features, int32_t *arg1, void *canvas, int32_t canvas_dtype, void *stream);
int pcp_mp_pfn_train_route_out_grad(const pcp_grid_t *grid, const void *vox_workspace, int64_t n, int64_t kept_rows, const void *dcanvas,
                                    int32_t dcanvas_dtype, const float *dpillar, const int32_t *arg1, void *dz1, int32_t dz1_dtype, void *stream);
int pcp_mp_pfn_train_route_mid_grad(const pcp_grid_t *grid, const void *vox_workspace, int64_t n, const void *din1, int32_t din1_dtype,
                                    const int32_t *arg0, float *da0, void *stream);

/* pcp_sparse_conv3x3_s2 (pcp_hip.h: the first backbone layer run from the pillar list) with a storage type for its OUTPUT map: the frozen
 * teachers of the bf16 loop hand bf16 to their second layer directly (fp32 arithmetic inside, one rounding on store). */
int pcp_mp_sparse_conv3x3_s2(const float *pillar_features, const pcp_grid_t *grid, const void *vox_workspace, int64_t n, const float *w_packed,
                             const float *bias, int32_t cout, int32_t relu, void *out, int32_t out_dtype, int32_t ld_out, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * 3x3 convolution (padding 1, stride 1 | 2) + bias (+ ReLU) on the bf16 matrix cores, fp32 accumulation.
 * Replaces nn.Conv2d(k=3) [+ folded eval-mode BatchNorm + ReLU for the frozen teachers] of base_bev_backbone.py:30-69,
 * center_head.py:24-29,75-82, v2x_fusion_disco.py:51-63 under autocast, and -- with the flipped / transposed weight form -- its data
 * gradient.  in: f32 | bf16 (f32 is rounded to bf16 while it is staged); out: f32 | bf16; w_packed from pcp_mp_pack_conv3x3; bias: cout_pad
 * floats.  cin % 16 == 0, cout_pad % 64 == 0, ld_in % 4 == 0, ld_out % 4 == 0 (bf16 out of the fast kernel: cout % 8 == 0 and ld_out % 8 == 0).
 * Two kernels behind one entry point (pcp_mp_conv3x3_plan tells which): stride 1, bf16 input, cin % 32 == 0 goes to the persistent
 * direct-to-LDS kernel (csrc/mp_conv.hip: k_mp_conv3x3_s1); everything else to the register-staged general kernel.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t batch, in_h, in_w;   /* NHWC input (batch, in_h, in_w, ld_in) */
  int32_t cin, cout, cout_pad;
  int32_t stride;              /* 1 | 2; output (batch, (in_h-1)/stride+1, (in_w-1)/stride+1, ld_out) */
  int32_t ld_in, ld_out;       /* elements per pixel of the buffers the channel windows live in */
  int32_t relu;
  int32_t in_dtype, out_dtype; /* PCP_DT_* */
} pcp_mp_conv3x3_t;

/* bf16 weight form of a (cout, cin, 3, 3) float32 tensor: [cin/16][cout_pad/64][9 (ky*3+kx)][2 (k half)][64 (cout)][8 (k)] bf16, zero for
 * cout..cout_pad.  transpose != 0: the data-gradient form -- input / output channels swapped and taps flipped, i.e. the weights of the
 * stride-1 convolution (cout channels in, cin channels out) that maps dL/dy to dL/dx; `cout_pad` then pads cin.  Device-side pack (one launch). */
size_t pcp_mp_conv3x3_packed_bytes(int32_t contraction_channels, int32_t out_channels_pad);
int pcp_mp_pack_conv3x3(const float *w, int32_t cout, int32_t cin, int32_t transpose, const float *fold_scale, void *packed, int32_t out_pad,
                        void *stream);
/* The same pack for MANY layers in one launch (every 3x3 layer of the trainable branch, both forms, once per optimizer step): a job table
 * in DEVICE memory; block_start = first block of the job in the grouped launch (jobs in ascending order),
 * pcp_mp_pack_conv3x3_group_blocks(job) = blocks the job needs (-1: invalid).  Bit-identical to per-layer pcp_mp_pack_conv3x3 calls. */
typedef struct {
  const float *w;              /* (cout, cin, 3, 3) float32 master weights */
  void *packed;                /* destination, pcp_mp_conv3x3_packed_bytes */
  int32_t cout, cin, transpose, out_pad;
  int32_t block_start, reserved;
} pcp_mp_pack_job_t;
int pcp_mp_pack_conv3x3_group_blocks(const pcp_mp_pack_job_t *job);
int pcp_mp_pack_conv3x3_group(const pcp_mp_pack_job_t *jobs_device, int32_t n_jobs, int32_t total_blocks, void *stream);
int pcp_mp_conv3x3(const pcp_mp_conv3x3_t *desc, const void *in, const void *w_packed, const float *bias, void *out, void *stream);
/* which kernel a descriptor goes to (1 = k_mp_conv3x3_s1, the direct-to-LDS persistent kernel; 0 = the general kernel) and the flops the
 * launch executes on the matrix pipe (padding channels / tiles included); either output may be NULL */
int pcp_mp_conv3x3_plan(const pcp_mp_conv3x3_t *desc, int32_t *fast_kernel, double *executed_flops);

/* ------------------------------------------------------------------------------------------------------------------
 * Weight gradient of the 3x3 convolution: dw[co][ci][ky][kx] (+)= sum over pixels dy[p][co] * x[p * stride + (ky-1, kx-1)][ci], a
 * pixel-contraction GEMM on v_mfma_f32_32x32x16_bf16 (operands transposed out of the NHWC LDS images by ds_read_b64_tr_b16), fp32
 * accumulation, split over the pixels with a fixed-order reduction (bitwise reproducible).  x, dy: bf16 (the activations / gradients the
 * bf16 loop stores; the descriptor's dtype fields must say PCP_DT_BF16), 16-byte aligned, ld % 8 == 0; dw float32 (cout, cin, 3, 3).
 * cin % 8 == 0, cout % 8 == 0 (padded to 64 x 64 tiles internally); stride 2 needs even in_h, in_w.
 * Replaces the weight-gradient half of nn.Conv2d's autograd node under autocast.  workspace: pcp_mp_conv3x3_wgrad_workspace_bytes.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t batch, in_h, in_w;   /* of x; dy is (batch, in_h / stride, in_w / stride) */
  int32_t cin, cout, stride;
  int32_t ld_x, ld_dy;
  int32_t x_dtype, dy_dtype;
  int32_t accumulate;          /* dw += instead of = */
} pcp_mp_wgrad3x3_t;
size_t pcp_mp_conv3x3_wgrad_workspace_bytes(const pcp_mp_wgrad3x3_t *desc);
int pcp_mp_conv3x3_wgrad(const pcp_mp_wgrad3x3_t *desc, const void *x, const void *dy, float *dw, void *workspace, size_t workspace_bytes,
                         void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * The pointwise family (Conv2d 1x1 / Conv2d k2 s2 / ConvTranspose2d k2 s2; modes and packed-weight layout as pcp_pointwise in pcp_hip.h,
 * the weights as bf16: [k_total/16][taps_out * cout_pad][16]) on bf16 or fp32 activations, bf16 products, fp32 accumulation, bias (+ ReLU)
 * in fp32, output bf16 or fp32 -- the up / down-sampling `deblocks` of BaseBEVBackbone (base_bev_backbone.py:48-69) as they run under
 * autocast.  cin % 32 == 0, cout % 8 == 0, cout_pad % 64 == 0, ld % 8 == 0, 16-byte aligned pointers, outputs below 2 GiB; anything else
 * returns PCP_ERR_UNSUPPORTED (the caller keeps the fp32 entry point).  No second K source / residual.
 * pcp_mp_pointwise_wgrad: out[n][k] (+)= sum_r a[map_a(r)][n] * b[map_b(r)][k] (row maps as pcp_rowmap_t), both operands bf16, float32
 * result; the contraction runs over the rows on v_mfma_f32_32x32x16_bf16 (operands transposed out of LDS by ds_read_b64_tr_b16), split
 * over the rows with a fixed-order reduction.  channels % 8 == 0, ld % 8 == 0; extent_bytes = bytes addressable from ptr (< 2 GiB).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t mode;                   /* PCP_PW_PLAIN | PCP_PW_SPACE2DEPTH | PCP_PW_DEPTH2SPACE */
  int64_t rows;                   /* PLAIN: number of rows; otherwise derived from batch / in_h / in_w */
  int32_t batch, in_h, in_w;
  int32_t cin, cout, cout_pad;
  int32_t ld_in, ld_out;
  int32_t relu;
  int32_t in_dtype, out_dtype;    /* PCP_DT_F32 | PCP_DT_BF16 */
} pcp_mp_pointwise_t;
int pcp_mp_pointwise(const pcp_mp_pointwise_t *desc, const void *in, const void *w_packed_bf16, const float *bias, void *out, void *stream);

typedef struct {
  const void *ptr;
  int32_t ld, channels;
  int32_t lattice, grid_h, grid_w, ky, kx;
  int32_t dtype;                  /* PCP_DT_BF16 */
  uint64_t extent_bytes;
} pcp_mp_rowmap_t;
size_t pcp_mp_pointwise_wgrad_workspace_bytes(int64_t rows, int32_t n, int32_t k);
int pcp_mp_pointwise_wgrad(const pcp_mp_rowmap_t *a, const pcp_mp_rowmap_t *b, int64_t rows, void *workspace, size_t workspace_bytes,
                           float *out, int32_t ld_out, int32_t accumulate, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * The DiscoNet fusion module's stacked maps ([ego | warped agent] per map, v2x_fusion_disco.py:85-115) as bf16 in the bf16 loop:
 * pcp_warp_nearest, pcp_softmax_fuse (pcp_hip.h) and pcp_disco_fuse_backward (pcp_hip_train.h) with a storage type for the maps.
 * Arithmetic (softmax, weighted sum, dot products) stays float32; only the map loads / the warp's stores change type.
 * ------------------------------------------------------------------------------------------------------------------ */
int pcp_mp_warp_nearest(const void *src, int32_t src_dtype, void *dst, int32_t dst_dtype, int32_t h, int32_t w, int32_t c, int32_t ld_src,
                        int32_t ld_dst, const float *theta_host, int32_t accumulate, void *stream);
int pcp_mp_softmax_fuse(const void *const *maps_host, int32_t map_dtype, int32_t n_agents, const float *weights, int32_t ld_w, int64_t pixels,
                        int32_t c, int32_t ld_map, int32_t ld_out, float *out, void *stream);
int pcp_mp_disco_fuse_backward(const void *const *maps_host, int32_t map_dtype, int32_t n_agents, int32_t ld_map, int32_t c, const float *logits,
                               int32_t ld_w, const float *dfused, int32_t ld_df, const float *const *h2_host, int32_t ld_h, const float *w4,
                               int64_t pixels, float *dmap0, int32_t ld_dm, float *const *dh2_host, void *workspace, float *dw4, float *db4,
                               int32_t accumulate, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PCP_HIP_MP_H */
