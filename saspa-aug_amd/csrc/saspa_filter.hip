// Kernels of the filter stage that follows generation (all_utils/utils.py:306-323, :357-375): pooling of the
// CLIP-RN50 / ResNet feature extractors and the sign-sqrt + L2-normalise tail of WSDAN_CAL's bilinear attention pooling.
// Both are HBM-bound streaming kernels (16-byte vectors of 8 channels per lane); the convolutions and linears of the two
// models are saspa_gemm launches (BatchNorm folded into weights + bias, ReLU in the epilogue).
#include "common.h"

namespace {

template <typename T>
__device__ __forceinline__ void ld8(const T* p, float* v) {
  if constexpr (sizeof(T) == 2) {
    Elem<T>::load_chunk(p, v);
  } else {
    Elem<T>::load_chunk(p, v);
    Elem<T>::load_chunk(p + 4, v + 4);
  }
}
template <typename T>
__device__ __forceinline__ void st8(T* p, const float* v) {
  if constexpr (sizeof(T) == 2) {
    Elem<T>::store_chunk(p, v);
  } else {
    Elem<T>::store_chunk(p, v);
    Elem<T>::store_chunk(p + 4, v + 4);
  }
}

// one lane = 8 channels of one output pixel; window taps outside the image are skipped (max) -- the average form has pad 0
template <typename T, int MODE>
__global__ __launch_bounds__(256) void pool2d_kernel(const T* x, int ldx, T* y, int ldy, int batch, int hin, int win, int C, int k,
                                                     int stride, int pad, int hout, int wout) {
  const int C8 = C >> 3;
  const long long total = (long long)batch * hout * wout * C8;
  const float inv = 1.0f / (float)(k * k);
  for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long long)gridDim.x * 256) {
    const int c = (int)(it % C8) * 8;
    long long pix = it / C8;
    const int ox = (int)(pix % wout);
    pix /= wout;
    const int oy = (int)(pix % hout);
    const int b = (int)(pix / hout);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = MODE == 0 ? -3.0e38f : 0.0f;
    for (int ty = 0; ty < k; ++ty) {
      const int iy = oy * stride - pad + ty;
      if ((unsigned)iy >= (unsigned)hin) continue;
      for (int tx = 0; tx < k; ++tx) {
        const int ix = ox * stride - pad + tx;
        if ((unsigned)ix >= (unsigned)win) continue;
        float v[8];
        ld8(x + ((long long)(b * hin + iy) * win + ix) * ldx + c, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = MODE == 0 ? fmaxf(acc[j], v[j]) : acc[j] + v[j];
      }
    }
    if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] *= inv;
    }
    st8(y + ((long long)(b * hout + oy) * wout + ox) * ldy + c, acc);
  }
}

// one workgroup per row: y = sign(x) sqrt(|x| + eps) kept in registers / re-derived, sum of squares in fp32 per lane then
// fp64 across the workgroup (65 536 elements per row for ResNet-101 x 32 attention maps), out = scale * y / max(|y|, 1e-12)
__global__ __launch_bounds__(256) void signsqrt_l2norm_kernel(const float* x, long long ldx, float* y, long long ldy, long long C,
                                                              float eps, float scale) {
  __shared__ double part[4];
  const float* xr = x + (long long)blockIdx.x * ldx;
  float* yr = y + (long long)blockIdx.x * ldy;
  float ss = 0.0f;
  for (long long i = threadIdx.x; i < C; i += 256) {
    const float v = xr[i];
    ss += fabsf(v) + eps;                                 // y^2 = |x| + eps exactly (before rounding of the sqrt)
  }
  double d = (double)wave_sum(ss);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = d;
  __syncthreads();
  const double tot = part[0] + part[1] + part[2] + part[3];
  const float inv = scale / fmaxf((float)sqrt(tot), 1e-12f);
  for (long long i = threadIdx.x; i < C; i += 256) {
    const float v = xr[i];
    const float r = sqrtf(fabsf(v) + eps);
    yr[i] = (v > 0.0f ? r : (v < 0.0f ? -r : 0.0f)) * inv;
  }
}

}  // namespace

extern "C" int saspa_pool2d(int dtype, int mode, const void* x, int ldx, void* y, int ldy, int batch, int hin, int win, int C, int k,
                            int stride, int pad, void* stream) {
  if (!x || !y || batch <= 0 || hin <= 0 || win <= 0 || C <= 0 || k <= 0 || stride <= 0 || pad < 0) return SASPA_EINVAL;
  if (mode != 0 && mode != 1) return SASPA_EINVAL;
  if (C % 8 || ldx % 8 || ldy % 8 || ldx < C || ldy < C || !aligned16(x) || !aligned16(y)) return SASPA_EALIGN;
  if (mode == 1 && pad != 0) return SASPA_ERANGE;         // AvgPool2d(k) only: every window lies inside the image
  const int hout = (hin + 2 * pad - k) / stride + 1, wout = (win + 2 * pad - k) / stride + 1;
  if (hout <= 0 || wout <= 0) return SASPA_ERANGE;
  if (pad >= k) return SASPA_ERANGE;                      // a window must hold at least one real pixel
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  long long blocks = ((long long)batch * hout * wout * (C / 8) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  const dim3 g((unsigned)blocks), t(256);
  if (dtype == SASPA_BF16) {
    if (mode == 0) hipLaunchKernelGGL((pool2d_kernel<bf16_t, 0>), g, t, 0, s, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, batch, hin, win, C, k, stride, pad, hout, wout);
    else hipLaunchKernelGGL((pool2d_kernel<bf16_t, 1>), g, t, 0, s, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, batch, hin, win, C, k, stride, pad, hout, wout);
  } else if (dtype == SASPA_F32) {
    if (mode == 0) hipLaunchKernelGGL((pool2d_kernel<float, 0>), g, t, 0, s, (const float*)x, ldx, (float*)y, ldy, batch, hin, win, C, k, stride, pad, hout, wout);
    else hipLaunchKernelGGL((pool2d_kernel<float, 1>), g, t, 0, s, (const float*)x, ldx, (float*)y, ldy, batch, hin, win, C, k, stride, pad, hout, wout);
  } else {
    return SASPA_EINVAL;
  }
  SASPA_CHECK_LAUNCH();
  return 0;
}

extern "C" int saspa_signsqrt_l2norm(const float* x, long long ldx, float* y, long long ldy, int rows, long long C, float eps,
                                     float scale, void* stream) {
  if (!x || !y || rows <= 0 || C <= 0 || ldx < C || ldy < C) return SASPA_EINVAL;
  hipLaunchKernelGGL(signsqrt_l2norm_kernel, dim3((unsigned)rows), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, ldx, y, ldy,
                     C, eps, scale);
  SASPA_CHECK_LAUNCH();
  return 0;
}
