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

#ifdef __cplusplus
}
#endif
#endif /* PCP_HIP_TRAIN_H */
