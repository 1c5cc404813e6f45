// What does a partner wave's LDS fragment traffic cost the wave that owns the matrix pipe?  One workgroup of 8 waves per CU:
// waves 0-3 (one per SIMD) issue bf16 MFMAs back to back, waves 4-7 (their SIMD partners) issue conflict-free ds_read_b128
// at a set rate (R reads per "gap" of the partner's MFMA stream, paced with s_nop / s_sleep-free loops), or nothing.
// Reported: cycles per MFMA of the MFMA waves, for v_mfma_f32_16x16x32_bf16 and v_mfma_f32_32x32x16_bf16.
//   build + run (GPU box):  hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_lds_contention.hip -o /tmp/mlc && /tmp/mlc
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

// KIND 0: 16x16x32, 1: 32x32x16.  READS: ds_read_b128 the partner issues per loop iteration (its loop has no other work);
// SELF: the MFMA wave itself issues SELF ds_read_b128 per 8 MFMAs (interleaved), partner idle if READS == 0
template <int KIND, int READS, int SELF>
__global__ __launch_bounds__(512) void contend(long long* out, float* sink, int iters) {
  __shared__ f32x4 lds[4096];                      // 64 KB
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = f32x4{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  // conflict-free 16-byte reads: lane l reads row (l & 15) of 128 bytes, chunk (l >> 4) ^ (row & 7)
  const int rd = ((lane & 15) * 8 + ((lane >> 4) ^ (lane & 7))) * 16 + (wave & 3) * 2048;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)((threadIdx.x) & 3); b[i] = (__bf16)1.0f; }
  f32x4 s4[10];
  f32x16 s16[4];
  for (int n = 0; n < 10; ++n) s4[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int n = 0; n < 4; ++n)
    for (int i = 0; i < 16; ++i) s16[n][i] = 0.f;
  f32x4 r[8];
  for (int i = 0; i < 8; ++i) r[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  long long t0 = 0, t1 = 0;
  if (wave < 4) {
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 40; ++m) {
        if (KIND == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(s4[m % 10]) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s16[m & 3]) : "v"(a), "v"(b));
        if (SELF > 0 && (m % 8) < SELF)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[m & 7]) : "v"(rd), "n"(((m * 5) & 15) * 128 * 16 % 32768));
      }
      if (SELF > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    t1 = __builtin_amdgcn_s_memtime();
  } else if (READS > 0) {
    // the partner: READS reads, wait, repeat -- for as long as the MFMA waves run (flag in LDS set by wave 0's lanes at the end)
    volatile int* done = reinterpret_cast<volatile int*>(&lds[4095]);
    for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
      for (int k = 0; k < READS; ++k)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[k & 7]) : "v"(rd), "n"((k & 15) * 128 * 16 % 32768));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if ((it & 63) == 0 && done[0] == 0x600D) break;
    }
  }
  if (wave < 4) {
    __builtin_amdgcn_s_waitcnt(0);
    if (wave == 0 && lane == 0) reinterpret_cast<volatile int*>(&lds[4095])[0] = 0x600D;
  }
  float s = 0.f;
  for (int n = 0; n < 10; ++n) s += s4[n][0];
  for (int n = 0; n < 4; ++n) s += s16[n][0];
  for (int i = 0; i < 8; ++i) s += r[i][0];
  if (s == 12345.f) sink[threadIdx.x] = s;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int KIND, int READS, int SELF>
void run(const char* name, long long* d, float* sink) {
  const int iters = 5000;
  hipLaunchKernelGGL((contend<KIND, READS, SELF>), dim3(256), dim3(512), 0, 0, d, sink, iters);
  hipDeviceSynchronize();
  long long h = 0;
  hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-78s %6.2f cycles per MFMA\n", name, (double)h / iters / 40);
}

int main() {
  long long* d; float* sink;
  if (hipMalloc(&d, 4096) != hipSuccess || hipMalloc(&sink, 4096) != hipSuccess) return 1;
  run<0, 0, 0>("16x16x32, partner idle", d, sink);
  run<0, 4, 0>("16x16x32, partner: 4 x ds_read_b128 + wait, looping", d, sink);
  run<0, 8, 0>("16x16x32, partner: 8 x ds_read_b128 + wait, looping", d, sink);
  run<0, 0, 2>("16x16x32, own stream: 2 ds_read_b128 per 8 MFMAs (10 per 40)", d, sink);
  run<0, 0, 4>("16x16x32, own stream: 4 ds_read_b128 per 8 MFMAs (20 per 40)", d, sink);
  run<1, 0, 0>("32x32x16, partner idle", d, sink);
  run<1, 4, 0>("32x32x16, partner: 4 x ds_read_b128 + wait, looping", d, sink);
  run<1, 8, 0>("32x32x16, partner: 8 x ds_read_b128 + wait, looping", d, sink);
  run<1, 0, 2>("32x32x16, own stream: 2 ds_read_b128 per 8 MFMAs (10 per 40)", d, sink);
  run<1, 0, 4>("32x32x16, own stream: 4 ds_read_b128 per 8 MFMAs (20 per 40)", d, sink);
  run<1, 0, 8>("32x32x16, own stream: 8 ds_read_b128 per 8 MFMAs (40 per 40)", d, sink);
  return 0;
}
