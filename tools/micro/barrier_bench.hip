// What does one s_barrier cost a 512-thread (8-wave) workgroup on gfx950, with nothing else in the loop?  And how long is the
// matrix pipe idle when one wave group hands it to the other through a barrier (the 8-wave GEMM's ping-pong)?
//   build: hipcc -O3 --offload-arch=gfx950 tools/micro/barrier_bench.hip -o gpurun_out/barrier_bench
//   run:   gpurun_out/barrier_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

template <int NT>
__global__ __launch_bounds__(NT) void bar_only(long long* out, int iters) {
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_barrier();
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

// ping-pong: group g (waves 4g..4g+3) runs NM MFMAs between its first and second barrier; group 1 is one barrier behind
// NT = 256: one wave per SIMD, MFMAs only (what ONE wave's back-to-back issue reaches)
template <int NM, int KIND>
__global__ __launch_bounds__(256) void onewave(long long* out, float* sink, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
  f32x16 big[2];
  for (int i = 0; i < 16; ++i) big[0][i] = big[1][i] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)1.0f; }
  const long long t0 = __builtin_amdgcn_s_memtime();
  const long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      if (KIND == 0) acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m & 7], 0, 0, 0);
      else big[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, big[m & 1], 0, 0, 0);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  const long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += big[0][i] + big[1][i];
  if (s == 12345.f) sink[threadIdx.x] = s;
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}

template <int NM, bool PINGPONG>
__global__ __launch_bounds__(512) void pingpong(long long* out, float* sink, int iters) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2;
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)1.0f; }
  if (PINGPONG && wm == 1) __builtin_amdgcn_s_barrier();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (PINGPONG) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int m = 0; m < NM; ++m) acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m & 7], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    if (PINGPONG) __builtin_amdgcn_s_barrier();
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (PINGPONG && wm == 0) __builtin_amdgcn_s_barrier();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.f) sink[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
  long long* d; float* sink;
  hipMalloc(&d, 256 * 8 * sizeof(long long)); hipMalloc(&sink, 4096);
  long long h[8];
  const int iters = 20000;
  auto clk = [&](const char* name, double per) {
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-58s %8.1f shader cycles per iteration (wave 0), %8.1f (wave 4)\n", name, h[0] / per, h[4] / per);
  };
  // s_memtime counts at a fixed 100 MHz on this chip; calibrate with an MFMA-only loop whose cycle count is known (16 / MFMA)
  hipLaunchKernelGGL((pingpong<40, false>), dim3(256), dim3(512), 0, 0, d, sink, iters);
  hipDeviceSynchronize();
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const double ticks_per_iter = (double)h[0] / iters;            // 2 waves x 40 MFMAs x 16 cycles = 1280 pipe cycles per iteration
  const double cyc_per_tick = 1280.0 / ticks_per_iter;
  printf("calibration: two waves / SIMD, 40 MFMAs each, no barrier: %.3f ticks per iteration -> %.2f pipe cycles per s_memtime tick\n", ticks_per_iter, cyc_per_tick);
  const double per = iters / cyc_per_tick;
  for (int kind = 0; kind < 2; ++kind) {
    if (kind == 0) hipLaunchKernelGGL((onewave<40, 0>), dim3(256), dim3(256), 0, 0, d, sink, iters);
    else hipLaunchKernelGGL((onewave<40, 1>), dim3(256), dim3(256), 0, 0, d, sink, iters);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("ONE wave per SIMD, 40 x %s back to back: %.2f s_memtime ticks per MFMA = %.2f calibrated cycles; %.2f ns per MFMA (s_memrealtime, 100 MHz)\n",
           kind ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_16x16x32_bf16", (double)h[0] / iters / 40, (double)h[0] / iters / 40 * cyc_per_tick, (double)h[1] * 10.0 / iters / 40);
  }
  hipLaunchKernelGGL((bar_only<512>), dim3(256), dim3(512), 0, 0, d, iters); clk("s_barrier only, 512 threads", per);
  hipLaunchKernelGGL((bar_only<256>), dim3(256), dim3(256), 0, 0, d, iters); clk("s_barrier only, 256 threads", per);
  hipLaunchKernelGGL((bar_only<128>), dim3(256), dim3(128), 0, 0, d, iters); clk("s_barrier only, 128 threads", per);
  hipLaunchKernelGGL((pingpong<20, true>), dim3(256), dim3(512), 0, 0, d, sink, iters); clk("ping-pong, 20 MFMAs per block (2 x 320 pipe cycles)", per);
  hipLaunchKernelGGL((pingpong<40, true>), dim3(256), dim3(512), 0, 0, d, sink, iters); clk("ping-pong, 40 MFMAs per block (2 x 640 pipe cycles)", per);
  hipLaunchKernelGGL((pingpong<80, true>), dim3(256), dim3(512), 0, 0, d, sink, iters); clk("ping-pong, 80 MFMAs per block (2 x 1280 pipe cycles)", per);
  return 0;
}
