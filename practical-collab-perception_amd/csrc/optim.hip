// Optimizer step of the reference's training loop on one flat parameter buffer: global gradient-norm clipping + Adam with
// decoupled weight decay, two launches per iteration instead of ~6 small ATen ops per parameter tensor (117 tensors in config 5).
//
// Replaces tools/train_utils/train_utils.py:57-58 (clip_grad_norm_(model.parameters(), GRAD_NORM_CLIP); optimizer.step()),
// tools/train_utils/optimization/fastai_optim.py:104-122 (OptimWrapper.step: p.mul_(1 - wd * lr) on every group, then
// torch.optim.Adam.step with weight_decay = 0) -- the one-cycle lr / beta1 of learning_schedules_fastai.py:44-77 are host scalars.
// HBM-bound: reads p, g, m, v and writes p, m, v once (28 B per parameter).
#include "pcp_common.h"

namespace {

__global__ __launch_bounds__(256) void k_sqnorm(const float *__restrict__ x, long long n, double *acc) {
  __shared__ double sh[4];
  double s = 0;
  const long long n4 = n >> 2;
  const float4 *x4 = reinterpret_cast<const float4 *>(x);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const float4 v = x4[i];
    s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = x[n4 * 4 + threadIdx.x]; s += (double)v * v; }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(acc, sh[0] + sh[1] + sh[2] + sh[3]);
}

struct AdamParams {
  float *p, *m, *v;
  const float *g;
  long long n;
  float lr, beta1, beta2, eps, decay;     // decay = 1 - wd * lr
  float step_size, inv_sqrt_bc2;          // lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t)
  float max_norm, grad_scale;
  const double *sqnorm;                   // device, of the UNSCALED gradient buffer; NULL: no clipping
};

__global__ __launch_bounds__(256) void k_adam(AdamParams a) {
  float coef = a.grad_scale;
  if (a.sqnorm) {
    const float norm = (float)sqrt(*a.sqnorm) * fabsf(a.grad_scale);
    const float c = a.max_norm / (norm + 1e-6f);       // torch.nn.utils.clip_grad_norm_
    coef *= fminf(c, 1.0f);
  }
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (long long)gridDim.x * blockDim.x) {
    const float g = a.g[i] * coef;
    float p = a.p[i] * a.decay;
    const float m = a.beta1 * a.m[i] + (1.f - a.beta1) * g;
    const float v = a.beta2 * a.v[i] + (1.f - a.beta2) * g * g;
    const float denom = sqrtf(v) * a.inv_sqrt_bc2 + a.eps;
    p -= a.step_size * (m / denom);
    a.p[i] = p;
    a.m[i] = m;
    a.v[i] = v;
  }
}

}  // namespace

extern "C" {

int pcp_grad_sqnorm(const float *grad, int64_t n, double *sqnorm, int32_t accumulate, void *stream) {
  if (!grad || !sqnorm || n < 0 || (((uintptr_t)grad) & 15)) return PCP_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate && pcp_zero_async(sqnorm, sizeof(double), s) != PCP_OK) return PCP_ERR_LAUNCH;
  if (n == 0) return PCP_OK;
  long long blocks = (n / 4 + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_sqnorm, dim3((unsigned)blocks), dim3(256), 0, s, grad, (long long)n, sqnorm);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

int pcp_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int64_t step, float max_norm, const double *sqnorm, float grad_scale, void *stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || step < 1) return PCP_ERR_ARG;
  if (n == 0) return PCP_OK;
  AdamParams a;
  a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.n = n;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  a.decay = (float)(1.0 - (double)weight_decay * (double)lr);
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  a.step_size = (float)((double)lr / bc1);
  a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  a.max_norm = max_norm; a.grad_scale = grad_scale; a.sqnorm = sqnorm;
  long long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  PCP_CHECK_LAUNCH();
  return PCP_OK;
}

}  // extern "C"
