// cfg_ddim.hip - fused classifier-free guidance + DDIM latent update (LatentDiffusionInpainting.py:198-212).
// One fp32 pass: reads eps_uncond, eps_cond, x, noise; writes x' (and optionally pred_x0 and the latent
// channels of the next UNet input).  Built with -ffp-contract=off and evaluated in the reference's
// operation order, so with identical inputs the result is bit-identical to the torch expression chain.
#include "common.h"

__global__ __launch_bounds__(256) void cfg_ddim_kernel(const float* __restrict__ eps, const float* x,
                                                       const float* __restrict__ noise, float guidance, float s1m,
                                                       float sqrt_at, float sqrt_aprev, float dir_coef, float sigma,
                                                       float temperature, float* x_out,
                                                       float* pred_x0, float* xc_out,
                                                       int xc_channels, int b, int chw) {
  const long long total = (long long)b * chw;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const float eu = eps[i], ec = eps[total + i];
    const float e = eu + guidance * (ec - eu);
    const float px0 = (x[i] - s1m * e) / sqrt_at;
    const float dir = dir_coef * e;
    const float nz = sigma * noise[i] * temperature;
    const float xn = sqrt_aprev * px0 + dir + nz;
    x_out[i] = xn;
    if (pred_x0) pred_x0[i] = px0;
    if (xc_out) {
      const long long img = i / chw, r = i - img * chw;
      const long long hw = chw / 4;
      xc_out[img * xc_channels * hw + r] = xn;
      xc_out[(img + b) * xc_channels * hw + r] = xn;
    }
  }
}

extern "C" int sdmi_cfg_ddim_step(const float* eps, const float* x, const float* noise, float guidance, float s1m, float a_t,
                                  float a_prev, float sigma, float temperature, float* x_out, float* pred_x0, float* xc_out,
                                  int xc_channels, int b, int chw, void* stream) {
  SDMI_REQUIRE(eps && x && noise && x_out && b > 0 && chw > 0, "cfg_ddim_step: bad argument");
  SDMI_REQUIRE(!xc_out || (xc_channels >= 4 && chw % 4 == 0), "cfg_ddim_step: xc_out needs xc_channels>=4 and 4 latent channels");
  // scalar coefficients exactly as torch evaluates them on fp32 tensors (:208-212)
  const float sqrt_at = sqrtf(a_t);
  const float sqrt_aprev = sqrtf(a_prev);
  const float dir_coef = sqrtf((1.0f - a_prev) - sigma * sigma);
  const long long total = (long long)b * chw;
  long long nb = (total + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(cfg_ddim_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, eps, x, noise, guidance, s1m,
                     sqrt_at, sqrt_aprev, dir_coef, sigma, temperature, x_out, pred_x0, xc_out, xc_channels, b, chw);
  return sdmi_check_launch("cfg_ddim_step");
}
