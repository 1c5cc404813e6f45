// Image pre-processing kernels of the safety-checker / subject front-ends (integer exact):
//   saspa_resample_u8    one separable pass of PIL's ImagingResample for 8-bit channels (the antialiased bicubic
//                        resize CLIPImageProcessor / BlipImageProcessor run through PIL.Image.resize): fixed-point
//                        coefficients with 22 fractional bits, accumulator seeded with 1 << 21, result clamped to
//                        0..255 -- bit-exact against Pillow.  The coefficient / bounds tables are host set-up.
//   saspa_u8_to_act_norm u8 RGB -> channels-last activations ((x / 255) - mean) / std, 8-channel pixels.
// HBM-bound byte work (a 512x512x3 image is 786 KB); nothing here is GEMM-shaped.
#include "common.h"

namespace {

constexpr int kPrecisionBits = 32 - 8 - 2;   // Pillow: PRECISION_BITS

// out[(o * out_len + t) * inner + i] = clip8( (1 << 21) + sum_k in[(o * in_len + min_t + k) * inner + i] * coef[t][k] )
__global__ __launch_bounds__(256) void resample_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                          long long total, int in_len, int out_len, int inner,
                                                          const int* __restrict__ bounds, const int* __restrict__ coeffs,
                                                          int ksize) {
  for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long long)gridDim.x * 256) {
    const int i = (int)(it % inner);
    const long long r = it / inner;
    const int t = (int)(r % out_len);
    const long long o = r / out_len;
    int lo = bounds[2 * t], cnt = bounds[2 * t + 1];
    cnt = min(cnt, ksize);
    lo = max(lo, 0);
    const int* k = coeffs + (long long)t * ksize;
    const uint8_t* p = src + (o * in_len) * inner + i;
    int ss = 1 << (kPrecisionBits - 1);
    for (int x = 0; x < cnt; ++x) {
      const int xi = min(lo + x, in_len - 1);           // tables are trusted input; never read outside the row anyway
      ss += (int)p[(long long)xi * inner] * k[x];
    }
    ss >>= kPrecisionBits;                              // arithmetic shift, then Pillow's clip8 lookup
    dst[it] = (uint8_t)min(max(ss, 0), 255);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void u8_to_act_norm_kernel(const uint8_t* src, T* dst, long long npix, float m0, float m1,
                                                             float m2, float s0, float s1, float s2) {
  for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < npix; it += (long long)gridDim.x * 256) {
    const uint8_t* s = src + it * 3;
    // rescale in double then round to fp32 (transformers' rescale), normalise in fp32
    const float r0 = (float)((double)s[0] * (1.0 / 255.0)), r1 = (float)((double)s[1] * (1.0 / 255.0)),
                r2 = (float)((double)s[2] * (1.0 / 255.0));
    float a[8] = {(r0 - m0) / s0, (r1 - m1) / s1, (r2 - m2) / s2, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (sizeof(T) == 2) {
      *reinterpret_cast<uint4*>(dst + it * 8) = pack8(a);
    } else {
      Elem<float>::store_chunk((float*)dst + it * 8, a);
      Elem<float>::store_chunk((float*)dst + it * 8 + 4, a + 4);
    }
  }
}

// StableDiffusionSafetyChecker's per-image decision + black-out without a host round trip: grid (slices, images); every
// block re-derives its image's flag from the 20 similarities (fp64, the same operation order as upstream's Python:
// cos = dot / |e|, score = cos - weight + adjustment, "round(score, 3) > 0" as the equivalent threshold compare).
__global__ __launch_bounds__(256) void safety_decide_kernel(const float* __restrict__ dots, int ldd, const float* __restrict__ gram,
                                                            int ldg, const double* __restrict__ special_w, int ns,
                                                            const double* __restrict__ concept_w, int nc, double thr,
                                                            uint8_t* images, long long bytes_per_image, int* flags) {
  __shared__ int flagged;
  const int img = blockIdx.y;
  if (threadIdx.x == 0) {
    const double norm = sqrt((double)gram[(long long)img * ldg + img]);
    double adj = 0.0;
    for (int j = 0; j < ns; ++j) {
      const double sc = (double)dots[(long long)img * ldd + j] / norm - special_w[j] + adj;
      if (sc >= thr) { adj = 0.01; break; }         // upstream: adjustment once ANY special-care score is positive
    }
    int f = 0;
    for (int j = 0; j < nc; ++j) {
      const double sc = (double)dots[(long long)img * ldd + ns + j] / norm - concept_w[j] + adj;
      if (sc >= thr) f = 1;
    }
    flagged = f;
    if (blockIdx.x == 0) flags[img] = f;
  }
  __syncthreads();
  if (!flagged) return;
  uint8_t* base = images + (long long)img * bytes_per_image;
  // 16-byte stores over the image's slice of this block (bytes_per_image % 16 == 0 is checked by the launcher)
  const long long n16 = bytes_per_image / 16;
  for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < n16; it += (long long)gridDim.x * 256)
    reinterpret_cast<uint4*>(base)[it] = make_uint4(0u, 0u, 0u, 0u);
}

unsigned grid_for(long long items) {
  long long g = (items + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 65536 ? 65536 : g));
}

}  // namespace

extern "C" int saspa_resample_u8(const uint8_t* src, uint8_t* dst, long long outer, int in_len, int out_len, int inner,
                                 const int* bounds, const int* coeffs, int ksize, void* stream) {
  if (!src || !dst || !bounds || !coeffs || outer <= 0 || in_len <= 0 || out_len <= 0 || inner <= 0 || ksize <= 0)
    return SASPA_EINVAL;
  if (ksize > 4096) return SASPA_ERANGE;
  const long long total = outer * out_len * inner;
  hipLaunchKernelGGL(resample_u8_kernel, dim3(grid_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, dst,
                     total, in_len, out_len, inner, bounds, coeffs, ksize);
  SASPA_CHECK_LAUNCH();
  return 0;
}

extern "C" int saspa_u8_to_act_norm(int dtype, const uint8_t* src, void* dst, long long npix, float mean0, float mean1,
                                    float mean2, float std0, float std1, float std2, void* stream) {
  if (!src || !dst || npix <= 0) return SASPA_EINVAL;
  if (!aligned16(dst)) return SASPA_EALIGN;
  if (!(std0 > 0.f) || !(std1 > 0.f) || !(std2 > 0.f)) return SASPA_EINVAL;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == SASPA_BF16)
    hipLaunchKernelGGL(u8_to_act_norm_kernel<bf16_t>, dim3(grid_for(npix)), dim3(256), 0, s, src, (bf16_t*)dst, npix, mean0, mean1, mean2, std0, std1, std2);
  else if (dtype == SASPA_F32)
    hipLaunchKernelGGL(u8_to_act_norm_kernel<float>, dim3(grid_for(npix)), dim3(256), 0, s, src, (float*)dst, npix, mean0, mean1, mean2, std0, std1, std2);
  else
    return SASPA_EINVAL;
  SASPA_CHECK_LAUNCH();
  return 0;
}

extern "C" int saspa_safety_decide(const float* dots, int ldd, const float* gram, int ldg, int nimg, const double* special_w,
                                   int n_special, const double* concept_w, int n_concepts, double threshold, uint8_t* images,
                                   long long bytes_per_image, int* flags, void* stream) {
  if (!dots || !gram || !special_w || !concept_w || !images || !flags || nimg <= 0 || n_special < 0 || n_concepts <= 0)
    return SASPA_EINVAL;
  if (ldd < n_special + n_concepts || ldg < nimg) return SASPA_ERANGE;
  if (bytes_per_image <= 0 || bytes_per_image % 16 || !aligned16(images)) return SASPA_EALIGN;
  const unsigned slices = (unsigned)((bytes_per_image / 16 + 256 * 16 - 1) / (256 * 16));
  hipLaunchKernelGGL(safety_decide_kernel, dim3(slices < 1 ? 1 : (slices > 256 ? 256 : slices), nimg), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), dots, ldd, gram, ldg, special_w, n_special, concept_w, n_concepts,
                     threshold, images, bytes_per_image, flags);
  SASPA_CHECK_LAUNCH();
  return 0;
}
