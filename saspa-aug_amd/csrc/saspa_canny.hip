// Canny edge extractor, integer-exact against cv2.Canny(img, low, high) (aperture 3,
// L1 gradient, 3-channel max-magnitude selection) -- replaces all_utils/utils.py:81-99.
//
// Three launches, all byte/integer work (HBM / LDS bound, ~2 MB per 512x512 image):
//   1. sobel:      u8 RGB -> per-pixel (dx, dy, |dx|+|dy|) of the channel with the largest
//                  magnitude (first one on ties), BORDER_REPLICATE.
//   2. nms:        fixed-point tangent test (TG22 = 13573, shift 15), asymmetric
//                  comparisons as OpenCV; map = 2 strong / 0 weak candidate / 1 none.
//   3. hysteresis: ONE workgroup per image keeps the strong / weak bitmaps (2 bits per
//                  pixel) in LDS and iterates word-parallel 8-neighbour dilation
//                  strong |= weak & dilate(strong) to the fixed point, then writes the
//                  3-channel 0/255 image (HWC3).  The fixed point equals OpenCV's stack
//                  flood fill (the set of weak pixels 8-connected to a strong one).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void canny_sobel_kernel(const uint8_t* src, short* dxy, short* mag, int n, int H, int W) {
  const long long total = (long long)n * H * W;
  for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long long)gridDim.x * 256) {
    const int x = (int)(it % W);
    const int y = (int)((it / W) % H);
    const long long img = it / ((long long)W * H);
    const uint8_t* base = src + img * H * W * 3;
    const int xm = max(x - 1, 0), xp = min(x + 1, W - 1);
    const int ym = max(y - 1, 0), yp = min(y + 1, H - 1);
    int bdx = 0, bdy = 0, bm = -1;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      auto px = [&](int yy, int xx) { return (int)base[((long long)yy * W + xx) * 3 + c]; };
      const int tl = px(ym, xm), tc = px(ym, x), tr = px(ym, xp);
      const int ml = px(y, xm), mr = px(y, xp);
      const int bl = px(yp, xm), bc = px(yp, x), br = px(yp, xp);
      const int dx = (tr + 2 * mr + br) - (tl + 2 * ml + bl);
      const int dy = (bl + 2 * bc + br) - (tl + 2 * tc + tr);
      const int m = abs(dx) + abs(dy);
      if (m > bm) { bm = m; bdx = dx; bdy = dy; }
    }
    dxy[it * 2 + 0] = (short)bdx;
    dxy[it * 2 + 1] = (short)bdy;
    mag[it] = (short)bm;
  }
}

__global__ __launch_bounds__(256) void canny_nms_kernel(const short* dxy, const short* mag, uint8_t* map, int n, int H, int W,
                                                        int low, int high) {
  const long long total = (long long)n * H * W;
  for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long long)gridDim.x * 256) {
    const int x = (int)(it % W);
    const int y = (int)((it / W) % H);
    const long long img = it / ((long long)W * H);
    const short* mg = mag + img * H * W;
    auto M = [&](int yy, int xx) -> int {
      return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? (int)mg[(long long)yy * W + xx] : 0;
    };
    const int m = (int)mag[it];
    uint8_t out = 1;
    if (m > low) {
      const int xs = dxy[it * 2], ys = dxy[it * 2 + 1];
      const long long ax = abs(xs);
      const long long ay = (long long)abs(ys) << 15;
      const long long tg22x = ax * 13573;
      bool keep;
      if (ay < tg22x) {
        keep = m > M(y, x - 1) && m >= M(y, x + 1);
      } else {
        const long long tg67x = tg22x + (ax << 16);
        if (ay > tg67x) {
          keep = m > M(y - 1, x) && m >= M(y + 1, x);
        } else {
          const int s = ((xs ^ ys) < 0) ? -1 : 1;
          keep = m > M(y - 1, x - s) && m > M(y + 1, x + s);
        }
      }
      if (keep) out = (m > high) ? 2 : 0;
    }
    map[it] = out;
  }
}

// one workgroup (1024 threads) per image; dynamic LDS = 2 * H * W32 words.  GLOBAL: images whose bitmaps exceed one CU's
// LDS (e.g. 1024 x 1024) keep them in the caller's scratch instead -- same algorithm, same fixed point (the growth is
// monotone, so the benign read/write overlap inside a sweep cannot change the result), L2-resident (256 KB per image).
template <bool GLOBAL>
__global__ __launch_bounds__(1024) void canny_hysteresis_kernel(const uint8_t* map, uint8_t* dst, int H, int W, uint32_t* gbits) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_bits[];
  __shared__ int changed;
  const int W32 = (W + 31) >> 5;
  const int nwords = H * W32;
  uint32_t* bits = GLOBAL ? gbits + (long long)blockIdx.x * 2 * nwords : lds_bits;
  uint32_t* strong = bits;
  uint32_t* weak = bits + nwords;
  const uint8_t* mp = map + (long long)blockIdx.x * H * W;
  uint8_t* out = dst + (long long)blockIdx.x * H * W * 3;
  const int tid = threadIdx.x;

  for (int wi = tid; wi < nwords; wi += 1024) {
    const int r = wi / W32, w = wi - r * W32;
    uint32_t s = 0, k = 0;
    for (int i = 0; i < 32; ++i) {
      const int x = w * 32 + i;
      if (x < W) {
        const uint8_t v = mp[(long long)r * W + x];
        s |= (uint32_t)(v == 2) << i;
        k |= (uint32_t)(v == 0) << i;
      }
    }
    strong[wi] = s;
    weak[wi] = k;
  }
  const int max_iter = H * W + 1;  // monotone growth: a fixed point is reached long before
  for (int iter = 0; iter < max_iter; ++iter) {
    __syncthreads();
    if (tid == 0) changed = 0;
    __syncthreads();
    bool mine = false;
    for (int wi = tid; wi < nwords; wi += 1024) {
      const uint32_t k = weak[wi];
      const uint32_t s0 = strong[wi];
      if ((k & ~s0) == 0) continue;
      const int r = wi / W32, w = wi - r * W32;
      uint32_t dil = 0;
#pragma unroll
      for (int dr = -1; dr <= 1; ++dr) {
        const int rr = r + dr;
        if (rr < 0 || rr >= H) continue;
        const uint32_t cur = strong[rr * W32 + w];
        const uint32_t left = (w > 0) ? strong[rr * W32 + w - 1] : 0u;
        const uint32_t right = (w + 1 < W32) ? strong[rr * W32 + w + 1] : 0u;
        dil |= cur | (cur << 1) | (cur >> 1) | (left >> 31) | (right << 31);
      }
      const uint32_t s1 = s0 | (k & dil);
      if (s1 != s0) {
        strong[wi] = s1;
        mine = true;
      }
    }
    if (mine) changed = 1;
    __syncthreads();
    if (!changed) break;
  }
  __syncthreads();
  const int npix = H * W;
  for (int pi = tid; pi < npix; pi += 1024) {
    const int r = pi / W, x = pi - r * W;
    const uint8_t v = ((strong[r * W32 + (x >> 5)] >> (x & 31)) & 1u) ? 255 : 0;
    out[(long long)pi * 3 + 0] = v;
    out[(long long)pi * 3 + 1] = v;
    out[(long long)pi * 3 + 2] = v;
  }
}

}  // namespace

extern "C" int saspa_canny(const uint8_t* src, uint8_t* dst, uint8_t* work, int n, int H, int W, int low, int high,
                           void* stream) {
  if (!src || !dst || !work || n <= 0 || H <= 0 || W <= 0) return SASPA_EINVAL;
  if (!aligned16(work)) return SASPA_EALIGN;
  if (low > high) { const int t = low; low = high; high = t; }
  const int W32 = (W + 31) / 32;
  const size_t lds_bytes = (size_t)2 * H * W32 * 4;
  const bool in_lds = lds_bytes <= 160 * 1024 - 64;      // else the bitmaps live in the scratch tail (needs W >= 64)
  if (!in_lds && W < 64) return SASPA_ERANGE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long long npix = (long long)n * H * W;
  // work layout: dxy int16[2*npix] | mag int16[npix] | map u8[npix]   (7 bytes / pixel)
  short* dxy = reinterpret_cast<short*>(work);
  short* mag = dxy + 2 * npix;
  uint8_t* map = reinterpret_cast<uint8_t*>(mag + npix);
  long long blocks = (npix + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(canny_sobel_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, dxy, mag, n, H, W);
  SASPA_CHECK_LAUNCH();
  hipLaunchKernelGGL(canny_nms_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dxy, mag, map, n, H, W, low, high);
  SASPA_CHECK_LAUNCH();
  if (in_lds) {
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(canny_hysteresis_kernel<false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
      if (e != hipSuccess) return (int)e;
      attr_set = true;
    }
    hipLaunchKernelGGL(canny_hysteresis_kernel<false>, dim3(n), dim3(1024), lds_bytes, s, map, dst, H, W, (uint32_t*)nullptr);
  } else {
    // scratch tail after the 7 bytes / pixel above: n * 2 * H * W32 words <= 0.4 bytes / pixel at W >= 64
    uint32_t* gbits = reinterpret_cast<uint32_t*>(work + ((7 * npix + 15) & ~15ll));
    hipLaunchKernelGGL(canny_hysteresis_kernel<true>, dim3(n), dim3(1024), 0, s, map, dst, H, W, gbits);
  }
  SASPA_CHECK_LAUNCH();
  return 0;
}
