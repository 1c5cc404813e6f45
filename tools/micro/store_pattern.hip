// Store-pattern probe (diagnostic): 256 workgroups x 512 threads write a 42 MB bf16 conv output (65536 x 320) the way the
// 8-wave conv kernel's epilogue does (each workgroup its own contiguous 160 KB, 8 KB per step) against other assignments.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/store_pattern.hip -o tools/micro/store_pattern && tools/micro/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// mode 0: workgroup w owns bytes [w*S, (w+1)*S), step i writes 8 KB at i*8 KB            (the epilogue today)
// mode 1: step-major: step i of workgroup w writes 8 KB at (i*G + w) * 8 KB                (fill-like)
// mode 2: as 0, but the two 80 KB halves are separated by a long ALU delay (half 1 waits)  (epilogue with its barrier + LDS pass)
// mode 3: as 0 with nontemporal stores
__global__ __launch_bounds__(512) void wr(u32x4* out, int steps, int mode, int delay) {
  const int w = blockIdx.x, G = gridDim.x, tid = threadIdx.x;
  u32x4 v = {(unsigned)w, (unsigned)tid, 1u, 2u};
  for (int i = 0; i < steps; ++i) {
    size_t idx;
    if (mode == 1) idx = ((size_t)i * G + w) * 512 + tid;
    else idx = ((size_t)w * steps + i) * 512 + tid;
    if (mode == 3) __builtin_nontemporal_store(v, out + idx);
    else out[idx] = v;
    if (mode == 2 && i == steps / 2 - 1) {
      for (int d = 0; d < delay; ++d) asm volatile("s_nop 15");
    }
  }
}

// reads back-to-back with the writes: the residual epilogue (mode 0 pattern): out[idx] = res[idx] + 1
__global__ __launch_bounds__(512) void rw(u32x4* out, const u32x4* res, int steps, int mode) {
  const int w = blockIdx.x, G = gridDim.x, tid = threadIdx.x;
  for (int i = 0; i < steps; ++i) {
    size_t idx = mode == 1 ? ((size_t)i * G + w) * 512 + tid : ((size_t)w * steps + i) * 512 + tid;
    u32x4 v = res[idx];
    v.x += 1;
    out[idx] = v;
  }
}

int main() {
  const int G = 256, steps = 20;
  const size_t n = (size_t)G * steps * 512;       // u32x4 elements: 41.9 MB
  // NBUF distinct outputs used in rotation: a buffer written again 20 launches later has left L2 (32 MB) and the Infinity
  // Cache (256 MB) -- rewriting ONE 42 MB buffer mostly overwrites lines that are still dirty in L2 and flatters the rate
  const int NBUF = 16;
  u32x4 *out0, *res;
  hipMalloc(&out0, n * 16 * NBUF);
  hipMalloc(&res, n * 16 * NBUF);
  hipMemset(out0, 0, n * 16 * NBUF);
  hipMemset(res, 1, n * 16 * NBUF);
  int rot = 0;
  u32x4* out = out0;
  const u32x4* res0 = res;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 5; ++i) { out = out0 + (size_t)(rot++ % NBUF) * n; res = const_cast<u32x4*>(res0) + (size_t)(rot % NBUF) * n; launch(); }
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
      hipEventRecord(e0);
      for (int i = 0; i < 20; ++i) { out = out0 + (size_t)(rot++ % NBUF) * n; res = const_cast<u32x4*>(res0) + (size_t)(rot % NBUF) * n; launch(); }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%-64s %7.1f us   %5.2f TB/s written\n", name, best / 20 * 1e3, n * 16 / (best / 20 * 1e-3) / 1e12);
  };
  run("0 per-workgroup contiguous 160 KB (epilogue today)", [&] { hipLaunchKernelGGL(wr, dim3(G), dim3(512), 0, 0, out, steps, 0, 0); });
  run("1 step-major (fill-like)", [&] { hipLaunchKernelGGL(wr, dim3(G), dim3(512), 0, 0, out, steps, 1, 0); });
  run("3 per-workgroup contiguous, nontemporal", [&] { hipLaunchKernelGGL(wr, dim3(G), dim3(512), 0, 0, out, steps, 3, 0); });
  run("0 with 512 workgroups x 10 steps (80 KB each)", [&] { hipLaunchKernelGGL(wr, dim3(2 * G), dim3(512), 0, 0, out, steps / 2, 0, 0); });
  run("0 with 2048 workgroups x 512 thr x 2.5 steps ~ 20 KB", [&] { hipLaunchKernelGGL(wr, dim3(8 * G), dim3(512), 0, 0, out, 2, 0, 0); });
  run("rw 0: residual read + write, per-workgroup contiguous", [&] { hipLaunchKernelGGL(rw, dim3(G), dim3(512), 0, 0, out, res, steps, 0); });
  run("rw 1: residual read + write, step-major", [&] { hipLaunchKernelGGL(rw, dim3(G), dim3(512), 0, 0, out, res, steps, 1); });
  return 0;
}
