// Issue rate of back-to-back independent bf16 MFMAs on gfx950, per SIMD: one wave alone against two waves sharing the SIMD,
// v_mfma_f32_16x16x32_bf16 (4 passes) against v_mfma_f32_32x32x16_bf16 (8 passes), accumulators in VGPRs or AGPRs.
// Why: the 8-wave GEMM hands the matrix pipe from one wave to its partner (ping-pong), so ONE wave's issue rate is its ceiling.
//   build + run (GPU box):  hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_issue_bench.hip -o /tmp/mib && /tmp/mib
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

// KIND 0: 16x16x32, accumulators "+v";  1: 16x16x32, "+a";  2: 32x32x16 "+v";  3: 32x32x16 "+a"
// NACC independent accumulators used round-robin; DISTINCT: every MFMA of a round reads its own A / B operand registers
// RANDOM: operands are pseudo-random bf16 in (-1, 1) (every lane, register and element different) instead of small constants:
// the data the matrix pipe toggles on decides what clock the power limit leaves it
template <int KIND, int NACC, bool DISTINCT, bool RANDOM = false>
__global__ void issue(long long* out, float* sink, int iters) {
  bf16x8 a[4], b[4];
  for (int k = 0; k < 4; ++k)
    for (int i = 0; i < 8; ++i) {
      if (RANDOM) {
        unsigned h = (threadIdx.x * 2654435761u) ^ ((blockIdx.x * 40503u + k * 97u + i * 13u) * 2246822519u);
        h ^= h >> 15; h *= 2654435761u; h ^= h >> 13;
        a[k][i] = (__bf16)(((int)(h & 0xffff) - 32768) / 32768.0f);
        b[k][i] = (__bf16)(((int)(h >> 16) - 32768) / 32768.0f);
      } else {
        a[k][i] = (__bf16)(float)((threadIdx.x + k) & 3); b[k][i] = (__bf16)1.0f;
      }
    }
  f32x4 s4[NACC];
  f32x16 s16[KIND >= 2 ? NACC : 1];
  for (int n = 0; n < NACC; ++n) s4[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int n = 0; n < (KIND >= 2 ? NACC : 1); ++n)
    for (int i = 0; i < 16; ++i) s16[n][i] = 0.f;
  const long long r0 = __builtin_amdgcn_s_memrealtime();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 40; ++m) {
      const int n = m % NACC, k = DISTINCT ? (m & 3) : 0;
      if (KIND == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(s4[n]) : "v"(a[k]), "v"(b[k]));
      if (KIND == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(s4[n]) : "v"(a[k]), "v"(b[k]));
      if (KIND == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s16[n]) : "v"(a[k]), "v"(b[k]));
      if (KIND == 3) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(s16[n]) : "v"(a[k]), "v"(b[k]));
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  const long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int n = 0; n < NACC; ++n) s += s4[n][0] + s4[n][3];
  for (int n = 0; n < (KIND >= 2 ? NACC : 1); ++n) s += s16[n][0] + s16[n][15];
  if (s == 12345.f) sink[threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}

template <int KIND, int NACC, bool DISTINCT, bool RANDOM = false>
void run(const char* name, long long* d, float* sink) {
  const int iters = RANDOM ? 200000 : 20000;
  for (int nt = 256; nt <= 512; nt += 256) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((issue<KIND, NACC, DISTINCT, RANDOM>), dim3(256), dim3(nt), 0, 0, d, sink, RANDOM ? iters : 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((issue<KIND, NACC, DISTINCT, RANDOM>), dim3(256), dim3(nt), 0, 0, d, sink, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    long long h[2];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const int wps = nt / 256;                          // waves per SIMD
    const double n_simd = 40.0 * iters * wps;          // MFMAs one SIMD executes
    const double flop = (KIND < 2 ? 16384.0 : 32768.0);
    printf("%-46s %d wave%s/SIMD: %6.2f s_memtime ticks, %6.2f ns (realtime), %6.2f ns (event) per MFMA per SIMD -> %6.0f TFLOP/s chip\n", name, wps,
           wps > 1 ? "s" : " ", h[0] / n_simd, h[1] * 10.0 / n_simd, ms * 1e6 / n_simd, flop / (ms * 1e6 / n_simd) * 1024 / 1e3);
  }
}

int main() {
  long long* d; float* sink;
  if (hipMalloc(&d, 64) != hipSuccess || hipMalloc(&sink, 4096) != hipSuccess) return 1;
  run<0, 8, false>("16x16x32 acc VGPR, 8 accumulators", d, sink);
  run<1, 8, false>("16x16x32 acc AGPR, 8 accumulators", d, sink);
  run<0, 10, true>("16x16x32 acc VGPR, 10 acc, 4 operand sets", d, sink);
  run<1, 10, true>("16x16x32 acc AGPR, 10 acc, 4 operand sets", d, sink);
  run<0, 2, false>("16x16x32 acc VGPR, 2 accumulators", d, sink);
  run<2, 2, false>("32x32x16 acc VGPR, 2 accumulators", d, sink);
  run<3, 2, false>("32x32x16 acc AGPR, 2 accumulators", d, sink);
  run<2, 4, true>("32x32x16 acc VGPR, 4 acc, 4 operand sets", d, sink);
  // ~60 - 110 ms of back-to-back MFMAs on random data (after an equally long warm-up launch): the clock the power limit allows
  run<0, 10, true, true>("RANDOM data: 16x16x32, 10 acc, 4 operand sets", d, sink);
  run<2, 4, true, true>("RANDOM data: 32x32x16, 4 acc, 4 operand sets", d, sink);
  run<0, 10, true, true>("RANDOM data: 16x16x32 again", d, sink);
  run<2, 4, true, true>("RANDOM data: 32x32x16 again", d, sink);
  return 0;
}
