// Head of the HED annotator (controlnet_aux HEDdetector.__call__, the reference's CONTROLNET = "hed" control image,
// run_aug/run_aug.py:311-312, :438-439): the five side outputs of the VGG stack (run as saspa_gemm convs) are
// resized to the image size with cv2.resize(INTER_LINEAR) semantics for float32 -- separable, float weights (1 - f, f),
// separate multiply / add roundings, horizontal pass first --, averaged in float32 in stack order, passed through the
// logistic function in fp64, scaled by 255, clipped and TRUNCATED to u8 (numpy astype), and replicated to three channels
// (HWC3).  One lane = one output pixel; tables (they depend on the sizes only) are host set-up.  Bit-exact against
// oracle/hed.py given the same side outputs.
#include "common.h"

namespace {

// plain * and + are contracted into v_fmac_f32 by hipcc (-ffp-contract=fast); OpenCV's scalar loop rounds each separately
__device__ __forceinline__ float mul_rn(float a, float b) {
  float r;
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float add_rn(float a, float b) {
  float r;
  asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__global__ __launch_bounds__(256) void hed_fuse_kernel(const SaspaHedFuseParams p) {
  const long long total = (long long)p.n * p.H * p.W;
  for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long long)gridDim.x * 256) {
    const int dx = (int)(it % p.W);
    const long long r = it / p.W;
    const int dy = (int)(r % p.H);
    const int b = (int)(r / p.H);
    float acc = 0.f;
    for (int k = 0; k < p.nmaps; ++k) {
      const int mw = p.mw[k], ld = p.ld[k];
      const float* m = p.map[k] + (long long)b * p.mh[k] * mw * ld;
      const int x0 = p.xofs[k][dx];
      const int x1 = min(x0 + 1, mw - 1);
      const float a0 = p.xw[k][2 * dx], a1 = p.xw[k][2 * dx + 1];
      const int r0 = p.yofs[k][2 * dy], r1 = p.yofs[k][2 * dy + 1];
      const float b0 = p.yw[k][2 * dy], b1 = p.yw[k][2 * dy + 1];
      const float* s0 = m + (long long)r0 * mw * ld;
      const float* s1 = m + (long long)r1 * mw * ld;
      const float h0 = add_rn(mul_rn(s0[(long long)x0 * ld], a0), mul_rn(s0[(long long)x1 * ld], a1));
      const float h1 = add_rn(mul_rn(s1[(long long)x0 * ld], a0), mul_rn(s1[(long long)x1 * ld], a1));
      const float v = add_rn(mul_rn(h0, b0), mul_rn(h1, b1));
      acc = k == 0 ? v : add_rn(acc, v);
    }
    const float mean = __fdiv_rn(acc, (float)p.nmaps);
    double e = 1.0 / (1.0 + exp(-(double)mean));
    e *= 255.0;
    e = e < 0.0 ? 0.0 : (e > 255.0 ? 255.0 : e);
    const uint8_t u = (uint8_t)(int)e;
    uint8_t* d = p.dst + it * 3;
    d[0] = u; d[1] = u; d[2] = u;
  }
}

}  // namespace

extern "C" int saspa_hed_fuse(const SaspaHedFuseParams* pp, void* stream) {
  if (!pp) return SASPA_EINVAL;
  const SaspaHedFuseParams& p = *pp;
  if (p.nmaps < 1 || p.nmaps > 5 || p.n <= 0 || p.H <= 0 || p.W <= 0 || !p.dst) return SASPA_EINVAL;
  for (int k = 0; k < p.nmaps; ++k) {
    if (!p.map[k] || !p.xofs[k] || !p.xw[k] || !p.yofs[k] || !p.yw[k]) return SASPA_EINVAL;
    if (p.mh[k] <= 0 || p.mw[k] <= 0 || p.ld[k] <= 0) return SASPA_EINVAL;
  }
  const long long total = (long long)p.n * p.H * p.W;
  long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(hed_fuse_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
  SASPA_CHECK_LAUNCH();
  return 0;
}
