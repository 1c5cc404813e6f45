// cv2.resize for 8-bit RGB in the modes the reference's resize_image uses (all_utils/utils.py:58-79):
//   saspa_resize_taps_u8   separable fixed-point filters of OpenCV's resizeGeneric_ (11-bit weights, 32-bit intermediate,
//                          no rounding between the passes): mode 0 = INTER_LANCZOS4 (8 taps, (v + 2^21) >> 22), mode 1 = the
//                          8-bit bilinear specialisation INTER_AREA falls back to when a side is up-scaled
//                          ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2 >> 2).  Taps are clamped to the image.
//   saspa_resize_area_u8   INTER_AREA down-scaling: resizeArea_<uchar, float> -- per destination cell the (source index,
//                          float weight) lists of computeResizeAreaTab, accumulated in float32 IN TABLE ORDER with separate
//                          multiply and add roundings (no FMA contraction), cvRound at the end -- and resizeAreaFast_ for
//                          integer scale factors.
// Tables are host set-up (they depend on the two sizes only); one lane = one output sample.  Byte / integer work on images
// of at most ~1.2 MP: launch-latency bound, not reshaped into anything else.  Parity: bit-exact against oracle/cv_resize.py,
// which restates OpenCV's published resize.cpp (no cv2 on either box: unpinned).
#include "common.h"

namespace {

// OpenCV's scalar area loop rounds every product and every sum separately.  HIP's __fmul_rn / __fadd_rn are plain * and +
// that hipcc contracts into v_fmac_f32 (-ffp-contract=fast is the HIP default; `#pragma clang fp contract(off)` did not
// stop it, checked in the ISA), which differs in the last bit: the two roundings are pinned with single instructions.
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

template <int NT, int MODE>
__global__ __launch_bounds__(256) void resize_taps_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int n, int h, int w,
                                                          int dh, int dw, const int* __restrict__ xofs, const short* __restrict__ xw,
                                                          const int* __restrict__ yofs, const short* __restrict__ yw) {
  const long long total = (long long)n * dh * dw * 3;
  for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long long)gridDim.x * 256) {
    const int c = (int)(it % 3);
    long long r = it / 3;
    const int dx = (int)(r % dw);
    r /= dw;
    const int dy = (int)(r % dh);
    const int b = (int)(r / dh);
    const uint8_t* img = src + (long long)b * h * w * 3 + c;
    const int x0 = xofs[dx], y0 = yofs[dy];
    int rows[NT];
#pragma unroll
    for (int ky = 0; ky < NT; ++ky) {
      const int sy = min(max(y0 + ky, 0), h - 1);
      const uint8_t* row = img + (long long)sy * w * 3;
      unsigned acc = 0;                                   // unsigned: 32-bit wrap-around exactly like the int arithmetic
#pragma unroll
      for (int kx = 0; kx < NT; ++kx) {
        const int sx = min(max(x0 + kx, 0), w - 1);
        acc += (unsigned)((int)row[sx * 3] * (int)xw[dx * NT + kx]);
      }
      rows[ky] = (int)acc;
    }
    int out;
    if (MODE == 0) {
      unsigned v = 0;
#pragma unroll
      for (int ky = 0; ky < NT; ++ky) v += (unsigned)(rows[ky] * (int)yw[dy * NT + ky]);
      out = ((int)(v + (1u << 21))) >> 22;
    } else {
      const int b0 = yw[dy * NT], b1 = yw[dy * NT + 1];
      out = (((b0 * (rows[0] >> 4)) >> 16) + ((b1 * (rows[1] >> 4)) >> 16) + 2) >> 2;
    }
    dst[it] = (uint8_t)min(max(out, 0), 255);
  }
}

__global__ __launch_bounds__(256) void resize_area_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int n, int h, int w,
                                                          int dh, int dw, const int* __restrict__ xstart, const int* __restrict__ xsi,
                                                          const float* __restrict__ xalpha, const int* __restrict__ ystart,
                                                          const int* __restrict__ ysi, const float* __restrict__ ybeta) {
  const long long total = (long long)n * dh * dw * 3;
  for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long long)gridDim.x * 256) {
    const int c = (int)(it % 3);
    long long r = it / 3;
    const int dx = (int)(r % dw);
    r /= dw;
    const int dy = (int)(r % dh);
    const int b = (int)(r / dh);
    const uint8_t* img = src + (long long)b * h * w * 3 + c;
    float sum = 0.0f;
    for (int j = ystart[dy]; j < ystart[dy + 1]; ++j) {
      const uint8_t* row = img + (long long)ysi[j] * w * 3;
      float buf = 0.0f;
      for (int k = xstart[dx]; k < xstart[dx + 1]; ++k)
        buf = add_rn(buf, mul_rn((float)row[xsi[k] * 3], xalpha[k]));      // buf += S * alpha (two roundings)
      sum = add_rn(sum, mul_rn(ybeta[j], buf));                               // sum += beta * buf
    }
    dst[it] = (uint8_t)min(max((int)rintf(sum), 0), 255);                          // saturate_cast<uchar>: round half to even
  }
}

__global__ __launch_bounds__(256) void resize_area_fast_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int n, int h, int w,
                                                               int dh, int dw, int isx, int isy) {
  const long long total = (long long)n * dh * dw * 3;
  const float scale = 1.0f / (float)(isx * isy);
  for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long long)gridDim.x * 256) {
    const int c = (int)(it % 3);
    long long r = it / 3;
    const int dx = (int)(r % dw);
    r /= dw;
    const int dy = (int)(r % dh);
    const int b = (int)(r / dh);
    const uint8_t* img = src + (long long)b * h * w * 3 + c;
    int sum = 0;
    for (int y = 0; y < isy; ++y)
      for (int x = 0; x < isx; ++x) sum += img[((long long)(dy * isy + y) * w + dx * isx + x) * 3];
    const int out = (isx == 2 && isy == 2) ? (sum + 2) >> 2 : (int)rintf(mul_rn((float)sum, scale));
    dst[it] = (uint8_t)min(max(out, 0), 255);
  }
}

inline unsigned blocks_for(long long items) {
  long long b = (items + 255) / 256;
  return (unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int saspa_resize_taps_u8(const uint8_t* src, uint8_t* dst, int n, int h, int w, int dh, int dw, const int* xofs,
                                    const short* xw, const int* yofs, const short* yw, int ntaps, int mode, void* stream) {
  if (!src || !dst || !xofs || !xw || !yofs || !yw || n <= 0 || h <= 0 || w <= 0 || dh <= 0 || dw <= 0) return SASPA_EINVAL;
  if (!((ntaps == 8 && mode == 0) || (ntaps == 2 && mode == 1))) return SASPA_ERANGE;
  if ((long long)h * w * 3 >= (1ll << 31)) return SASPA_ERANGE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 g(blocks_for((long long)n * dh * dw * 3)), t(256);
  if (mode == 0) hipLaunchKernelGGL((resize_taps_kernel<8, 0>), g, t, 0, s, src, dst, n, h, w, dh, dw, xofs, xw, yofs, yw);
  else hipLaunchKernelGGL((resize_taps_kernel<2, 1>), g, t, 0, s, src, dst, n, h, w, dh, dw, xofs, xw, yofs, yw);
  SASPA_CHECK_LAUNCH();
  return 0;
}

extern "C" int saspa_resize_area_u8(const uint8_t* src, uint8_t* dst, int n, int h, int w, int dh, int dw, const int* xstart,
                                    const int* xsi, const float* xalpha, const int* ystart, const int* ysi, const float* ybeta,
                                    int isx, int isy, void* stream) {
  if (!src || !dst || n <= 0 || h <= 0 || w <= 0 || dh <= 0 || dw <= 0) return SASPA_EINVAL;
  if ((long long)h * w * 3 >= (1ll << 31)) return SASPA_ERANGE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 g(blocks_for((long long)n * dh * dw * 3)), t(256);
  if (isx > 0 && isy > 0) {
    if (dw * isx > w || dh * isy > h) return SASPA_ERANGE;
    hipLaunchKernelGGL(resize_area_fast_kernel, g, t, 0, s, src, dst, n, h, w, dh, dw, isx, isy);
  } else {
    if (!xstart || !xsi || !xalpha || !ystart || !ysi || !ybeta) return SASPA_EINVAL;
    hipLaunchKernelGGL(resize_area_kernel, g, t, 0, s, src, dst, n, h, w, dh, dw, xstart, xsi, xalpha, ystart, ysi, ybeta);
  }
  SASPA_CHECK_LAUNCH();
  return 0;
}
