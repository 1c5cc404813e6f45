// The cross-attention half of a level-0 transformer block in ONE launch (C = 320 channels, 8 heads of 40, <= 96 text keys):
//     y = x + to_out( softmax( to_q(LayerNorm(x)) K^T ) V ) + bias
// i.e. BasicTransformerBlock.norm2 -> attn2.to_q -> scaled_dot_product_attention(K, V of the 77 text tokens) -> attn2.to_out ->
// residual add -- three launches (A-stationary LayerNorm + to_q, flash attention over 77 keys, A-stationary to_out + residual)
// that each stream the [M, 320] token matrix through HBM (5 x 42 MB at M = 65 536), here one read of x and one write of y.
//
// Built on the A-stationary GEMM (saspa_gemm_as.hip; same W ring, same vmcnt bookkeeping, same epilogue): a wave keeps its
// 32 token rows in registers as MFMA B-operand fragments for the whole launch and every product is taken transposed
// (features / keys on the accumulator rows, tokens on the lanes), so each stage's accumulators ARE the next stage's B operand
// after packing to bf16 (cdna_hip_programming.md, "An accumulator tile as the next MFMA's operand"): no LDS round trip and no
// cross-lane movement between LayerNorm, to_q, Q K^T, softmax, P V and to_out.
//   1. x rows -> registers, LayerNorm in place (a row lives in two lanes).
//   2. Q^T = Wq' X^T, five 64-row slices of Wq' through the LDS ring.  Wq' row order: blocks 0-7 = the first 32 channels of
//      heads 0-7, blocks 8-9 = the last 8 channels of each head (head t / 8 of the 64 tail rows) -- a head's 40 channels are
//      one whole 32-row accumulator block (two K-steps of the next product) plus half a K-step taken out of a tail block as a
//      register pair (rows 0-7 of a K-step fragment are its first two registers, rows 8-15 its last two).
//   3. per head: S^T = K_h Q_h^T (96 keys x 32 tokens: 3 key blocks x 3 K-steps, the third K-step = the 8 tail channels and
//      zeros), softmax down the accumulator rows (48 values per lane + one lane^32 exchange; exp2: the softmax scale * log2 e is
//      folded into Wq'), P^T packed to bf16 = B operand of O^T = V_h^T P^T (2 blocks x 6 K-steps); row 40 of V_h^T is all ones,
//      so the softmax denominator comes out of the same MFMAs.  K_h / V_h^T arrive as ready-made A-operand fragments
//      (SaspaXattnBlockParams.kf / vf: time-invariant, built once per generation) straight from L2 into registers.
//   4. O^T / denominator -> bf16 -> the B operand of Y^T = Wo' O^T (K order of Wo' = the order the fragments come out in), five
//      more slices through the same ring; epilogue = bias + residual through the wave-private staging tile, 16-byte stores.
#include <cstdlib>

#include "common.h"
#include "gemm_internal.h"

namespace {

constexpr int XA_K = 320;
constexpr int XA_KS = XA_K / 16;          // MFMA K-steps of the two projections
constexpr int XA_BM = 256;                // rows per workgroup: 8 waves x 32
constexpr int XA_BN = 64;                 // W rows per ring slot
constexpr int XA_PITCH = 40;              // 16-byte chunks per W row in LDS (chunk kc of row n at kc ^ ((n >> 1) & 7))
constexpr int XA_NDMA = 5;
constexpr int XA_STAGE = 8 * XA_NDMA * 64 + 64;
constexpr int XA_RING = 3;
constexpr int XA_NSL = 10;                // slices: 5 of Wq', 5 of Wo'
constexpr int XA_STG_PITCH = 136;
constexpr int XA_STG_WAVE = 32 * XA_STG_PITCH;
constexpr int XA_STG_CHUNKS = 8 * XA_STG_WAVE / 16;
constexpr int XA_BIAS_SLOT = 8 * XA_NDMA * 64;
constexpr int XA_KF_HEAD = 9 * 1024;      // bytes of one head's K fragments: 3 key blocks x 3 K-steps x 64 lanes x 16 B
constexpr int XA_VF_HEAD = 12 * 1024;     // 2 channel blocks x 6 K-steps

template <int N> __device__ __forceinline__ void xwait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void xwait_vm_dyn(int n_any) {
  const int n = __builtin_amdgcn_readfirstlane(n_any);
#define SASPA_W1(i) case i: xwait_vm<i>(); break;
#define SASPA_W8(i) SASPA_W1(i) SASPA_W1(i + 1) SASPA_W1(i + 2) SASPA_W1(i + 3) SASPA_W1(i + 4) SASPA_W1(i + 5) SASPA_W1(i + 6) SASPA_W1(i + 7)
  switch (n < 63 ? (n < 0 ? 0 : n) : 63) {
    SASPA_W8(0) SASPA_W8(8) SASPA_W8(16) SASPA_W8(24) SASPA_W8(32) SASPA_W8(40) SASPA_W8(48) SASPA_W8(56)
    default: xwait_vm<0>(); break;
  }
#undef SASPA_W8
#undef SASPA_W1
}

__device__ __forceinline__ void xunpack_opaque(const u32x4& a, float* v) {
  u32x4 t = a;
  asm volatile("" : "+v"(t));
  unpack8(__builtin_bit_cast(uint4, t), v);
}

__device__ __forceinline__ f32x16 xmfma(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__global__ __launch_bounds__(512, 1) void xattn_block_kernel(const SaspaXattnBlockParams p, const int abl) {
  __shared__ u32x4 lds[XA_RING * XA_STAGE + XA_STG_CHUNKS];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, h = lane >> 5;
  const long long row = (long long)blockIdx.x * XA_BM + wave * 32 + m;      // < M: the host checks M % 256 == 0

  // ---- this lane's half of its row: channels 16 s + 8 h .. + 8 ----
  const rsrc_t rsa = make_rsrc(p.x);
  const unsigned aoff = (unsigned)(row * p.ldx * 2 + h * 16);
  u32x4 af[XA_KS];
#pragma unroll
  for (int s = 0; s < XA_KS; ++s) af[s] = buf_load(rsa, aoff, s * 32);

  // ---- W slices through the ring (as gemm_as_kernel): slice t = rows [64 t, 64 t + 64) of the stacked [Wq' ; Wo'] ----
  const rsrc_t rsw = make_rsrc(p.w);
  const rsrc_t rsb = make_rsrc(p.bias);
  // lane id re-derived at the point of use behind an asm barrier (2 VALU): the per-piece (row, chunk) pairs below are lane
  // constants the compiler would otherwise keep in 10 registers across the whole kernel, which has none to spare
  auto lane_now = []() __attribute__((always_inline)) {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
  };
  const int ndma = XA_NDMA + (wave == 7 ? 1 : 0);
  auto dma_piece = [&](int step, int i) __attribute__((always_inline)) {
    const bool real = step < XA_NSL;
    if (i < XA_NDMA) {
      const int q = (wave * XA_NDMA + i) * 64 + lane_now();
      const int n = (int)(((unsigned)q * 52429u) >> 21);                // q / 40 for q < 2560 (exact: 52429 = ceil(2^21 / 40))
      const int kcp = q - n * XA_PITCH;
      const int dkc = kcp ^ ((n >> 1) & 7);
      const unsigned off = real ? (unsigned)((step * XA_BN + n) * p.ldw * 2 + dkc * 16) : kInvalid;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_void_t*)(lds + (step % XA_RING) * XA_STAGE + (wave * XA_NDMA + i) * 64), 16, (int)off, 0, 0, 0);
    } else if (wave == 7) {
      const unsigned off = (real && lane < 16) ? (unsigned)((step * XA_BN + 4 * lane) * 4) : kInvalid;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (lds_void_t*)(lds + (step % XA_RING) * XA_STAGE + XA_BIAS_SLOT), 16, (int)off, 0, 0, 0);
    }
  };
  auto dma_slice = [&](int step) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i <= XA_NDMA; ++i) dma_piece(step, i);
  };
  dma_slice(0);
  dma_slice(1);

  // ---- LayerNorm of the row, in registers (the arithmetic of layernorm_kernel / gemm_as_kernel) ----
  // (a run-time branch on purpose: as straight-line code the scheduler hoists all eighty gamma / beta loads -- 320 registers --
  // to the top of the kernel and spills them; inside its own basic block four K-steps of them are in flight)
  if (p.ln_gamma != nullptr) {
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < XA_KS; ++s) {
      float v[8];
      xunpack_opaque(af[s], v);
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += v[j];
    }
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum / (float)XA_K;
    float sq = 0.f;
#pragma unroll
    for (int s = 0; s < XA_KS; ++s) {
      float v[8];
      xunpack_opaque(af[s], v);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[j] - mean; sq += d * d; }
    }
    sq += __shfl_xor(sq, 32, 64);
    const float rstd = 1.0f / sqrtf(sq / (float)XA_K + p.ln_eps);
    // compiler-level fences: the gamma / beta loads are speculatable and were hoisted -- all eighty of them, 320 registers -- to
    // the top of the kernel (sched_barrier only binds the machine scheduler); a "memory" clobber keeps four K-steps in flight
    asm volatile("" ::: "memory");
#pragma unroll
    for (int s = 0; s < XA_KS; ++s) {
      const int k0 = 16 * s + 8 * h;
      float v[8];
      xunpack_opaque(af[s], v);
      const f32x4 g0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.ln_gamma + k0)), g1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.ln_gamma + k0 + 4));
      const f32x4 b0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.ln_beta + k0)), b1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.ln_beta + k0 + 4));
      const float g[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
      const float bb[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (v[j] - mean) * rstd * g[j] + bb[j];
      af[s] = __builtin_bit_cast(u32x4, pack8(v));
      if ((s & 3) == 3) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
      }
    }
  }

  const unsigned char* fbase = reinterpret_cast<const unsigned char*>(lds) + m * XA_PITCH * 16;
  const int fkey = (m >> 1) & 7;
  int foff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) foff[j] = (((2 * j) | h) ^ fkey) << 4;
  constexpr int LPR = 8, RPI = 64 / LPR, NST = 32 / RPI, SPITCH = XA_STG_PITCH;

  // mma(step, B fragments, acc, between): the slice's 40 MFMAs (bias = the accumulators' initial value) in groups of 4 with
  // the W fragments of the next two groups in flight; between(gi) is dealt in behind group gi (gemm_as_kernel)
  constexpr int NG = XA_KS / 2;
  auto mma = [&](int step, const u32x4 (&bf)[XA_KS], f32x16 (&acc)[2], auto&& between) __attribute__((always_inline)) {
    const unsigned char* fs = fbase + (step % XA_RING) * (XA_STAGE * 16);
    const unsigned char* bs = reinterpret_cast<const unsigned char*>(lds) + ((step % XA_RING) * XA_STAGE + XA_BIAS_SLOT) * 16;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bs + (nb * 8 + 2 * g + h) * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[nb][4 * g + j] = bv[j];
      }
    auto frag = [&](int s, int nb) __attribute__((always_inline)) -> u32x4 {
      return *reinterpret_cast<const u32x4*>(fs + foff[s & 3] + (nb * 32 * XA_PITCH * 16 + ((2 * s) & ~7) * 16));
    };
    u32x4 wf[3][4];
#pragma unroll
    for (int gi = 0; gi < 2; ++gi)
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[gi][j] = frag(2 * gi + (j >> 1), j & 1);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      if (gi + 2 < NG) {
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[(gi + 2) % 3][j] = frag(2 * (gi + 2) + (j >> 1), j & 1);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int sx = 2 * gi + (j >> 1), nb = j & 1;
        acc[nb] = xmfma(wf[gi % 3][j], bf[sx], acc[nb]);
      }
      between(gi);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- vmcnt bookkeeping (gemm_as_kernel): `issued` counts every vector-memory instruction of this wave in issue order ----
  xwait_vm_dyn(ndma);
  int issued = 0;
  int mark0 = -64, mark1 = 0, mark2 = 0;
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);

  // =====================================================================================================================
  // stage 2: Q^T = Wq' LN(x)^T, slices 0 .. 4; block 2 t + nb of slice t -> K-steps qf[2 (2 t + nb)], qf[2 (2 t + nb) + 1]
  // =====================================================================================================================
  u32x4 qf[XA_KS];
  f32x16 acc[2];
  auto pack_block = [&](const f32x16& a, u32x4& k0, u32x4& k1) __attribute__((always_inline)) {
    k0 = u32x4{pack2(a[0], a[1]), pack2(a[2], a[3]), pack2(a[4], a[5]), pack2(a[6], a[7])};
    k1 = u32x4{pack2(a[8], a[9]), pack2(a[10], a[11]), pack2(a[12], a[13]), pack2(a[14], a[15])};
  };
#pragma unroll
  for (int step = 0; step < 5; ++step) {
    if (step > 0) xwait_vm_dyn(issued - mark0);
    __builtin_amdgcn_s_barrier();           // publishes this step's slice; the slot the DMA below overwrites is free
    mma(step, af, acc, [&](int gi) __attribute__((always_inline)) {
      if (gi <= XA_NDMA) dma_piece(step + 2, gi);
    });
    issued += ndma;
    mark2 = issued;
    pack_block(acc[0], qf[4 * step], qf[4 * step + 1]);
    pack_block(acc[1], qf[4 * step + 2], qf[4 * step + 3]);
    // pin the PACKED fragments here: left alone, LLVM sinks the bf16 conversion to the heads that consume it and keeps (spills)
    // the fp32 accumulator values instead -- twice the registers, reloaded behind vmcnt(0) in the attention stage
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      // (through a FLOAT vector: integer 4-vectors as asm in/out operands get wrong sub-registers from hipcc 7.2)
      f32x4 t = __builtin_bit_cast(f32x4, qf[4 * step + q]);
      asm volatile("" : "+v"(t));
      qf[4 * step + q] = __builtin_bit_cast(u32x4, t);
    }
    mark0 = mark1;
    mark1 = mark2;
  }

  // =====================================================================================================================
  // stage 3: attention over the text keys, head by head; the normalised O^T fragments replace LN(x) in af[]
  // =====================================================================================================================
  {
    // K_h / V_h^T fragments go through LDS once per WORKGROUP (all eight waves multiply against the same sample's keys): head h
    // = 21 wave-wide DMA instructions of 1 KB (9 K fragments, 12 V^T fragments; 24 slots, three per wave, the last three load
    // nothing), double-buffered in ring slot 1 (free since step 4) and in the epilogue staging area (unused until step 6) --
    // loaded straight from L2 into registers instead, each wave waited ~21 L2 round trips per head with nothing to overlap them
    // (91 us per launch at M = 65 536 against 100 us for the three launches it replaces).
    const int sample = (int)(((long long)blockIdx.x * XA_BM) / p.rows_per_sample);      // wave-uniform: one sample per workgroup
    const rsrc_t rsk = make_rsrc(p.kf);
    const rsrc_t rsv = make_rsrc(p.vf);
    const unsigned lo = (unsigned)(lane * 16);
    const unsigned kbase = (unsigned)((long long)sample * p.kf_stride);
    const unsigned vbase = (unsigned)((long long)sample * p.vf_stride);
    const int nk = p.nk;
    auto kvbuf = [&](int hd) __attribute__((always_inline)) -> u32x4* { return lds + ((hd & 1) ? XA_RING * XA_STAGE : XA_STAGE); };
    auto kv_dma = [&](int hd) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int piece = wave * 3 + i;                 // wave-uniform
        lds_void_t* dst = (lds_void_t*)(kvbuf(hd) + piece * 64);
        if (piece < 9) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsk, dst, 16, (int)lo, (int)(kbase + hd * XA_KF_HEAD + piece * 1024), 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsv, dst, 16, (int)(piece < 21 ? lo : kInvalid), (int)(vbase + hd * XA_VF_HEAD + (piece - 9) * 1024), 0, 0);
      }
    };
    __builtin_amdgcn_s_barrier();           // every wave is done reading slice 4 out of ring slot 1
    if (!(abl & 1)) {                        // (diagnostics: SASPA_XATTN_ABLATE=1 skips the attention stage, timing only)
    kv_dma(0);
    kv_dma(1);
#pragma unroll
    for (int hd = 0; hd < 8; ++hd) {
      if (hd == 0) xwait_vm<3>(); else xwait_vm<0>();        // this wave's pieces of head hd (and everything older) have landed
      __builtin_amdgcn_s_barrier();                          // ... everyone's have; the other buffer is free again
      if (hd >= 1 && hd + 1 < 8) kv_dma(hd + 1);
      const u32x4* kv = kvbuf(hd) + lane;
      const u32x4 q0 = qf[2 * hd], q1 = qf[2 * hd + 1];
      const u32x4 tq = qf[16 + 2 * (hd >> 2) + ((hd & 3) >> 1)];
      const u32x4 q2 = (hd & 1) ? u32x4{tq.z, tq.w, 0u, 0u} : u32x4{tq.x, tq.y, 0u, 0u};
      f32x16 s[3];
#pragma unroll
      for (int kb = 0; kb < 3; ++kb) {
        const u32x4 k0 = kv[(kb * 3 + 0) * 64], k1 = kv[(kb * 3 + 1) * 64], k2 = kv[(kb * 3 + 2) * 64];
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
        s[kb] = xmfma(k0, q0, s[kb]);
        s[kb] = xmfma(k1, q1, s[kb]);
        s[kb] = xmfma(k2, q2, s[kb]);
      }
      // softmax down the accumulator rows: key of register r of block kb = 32 kb + (r & 3) + 8 (r >> 2) + 4 h
      float mx = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < 3; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (32 * kb + 32 > nk) s[kb][r] = key < nk ? s[kb][r] : -INFINITY;
          mx = fmaxf(mx, s[kb][r]);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      u32x4 pf[6];
#pragma unroll
      for (int kb = 0; kb < 3; ++kb) {
        float e[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) e[r] = __builtin_amdgcn_exp2f(s[kb][r] - mx);
        pf[2 * kb] = u32x4{pack2(e[0], e[1]), pack2(e[2], e[3]), pack2(e[4], e[5]), pack2(e[6], e[7])};
        pf[2 * kb + 1] = u32x4{pack2(e[8], e[9]), pack2(e[10], e[11]), pack2(e[12], e[13]), pack2(e[14], e[15])};
      }
      f32x16 o[2];
#pragma unroll
      for (int db = 0; db < 2; ++db) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) o[db] = xmfma(kv[(9 + db * 6 + ks) * 64], pf[ks], o[db]);
      }
      // denominator = row 40 of O^T (the ones row of V^T) = register 4 of block 1 on the h = 0 lanes
      const float den = __shfl(o[1][4], m, 64);
      const float inv = 1.0f / den;
      u32x4 f0, f1;
      f32x16 on = o[0];
#pragma unroll
      for (int r = 0; r < 16; ++r) on[r] *= inv;
      pack_block(on, f0, f1);
      af[2 * hd] = f0;
      af[2 * hd + 1] = f1;
      const unsigned t0 = pack2(o[1][0] * inv, o[1][1] * inv), t1 = pack2(o[1][2] * inv, o[1][3] * inv);
      if (hd & 1) {
        af[16 + (hd >> 1)].z = t0;
        af[16 + (hd >> 1)].w = t1;
      } else {
        af[16 + (hd >> 1)].x = t0;
        af[16 + (hd >> 1)].y = t1;
      }
    }
    }
  }
  // everything this wave issued has landed (its shares of slices 5 and 6 included): the bookkeeping restarts from zero
  xwait_vm<0>();
  issued = 0;
  mark0 = 0;
  mark1 = 0;
  mark2 = 0;

  // =====================================================================================================================
  // stage 4: Y^T = Wo' O^T + bias, slices 5 .. 9, epilogue of slice t - 1 dealt into the MFMA groups of slice t
  // =====================================================================================================================
  const rsrc_t rsr = make_rsrc(p.residual);
  const rsrc_t rso = make_rsrc(p.out);
  unsigned char* stg = reinterpret_cast<unsigned char*>(lds + XA_RING * XA_STAGE) + wave * XA_STG_WAVE;
  const long long row0 = (long long)blockIdx.x * XA_BM + wave * 32;
  unsigned so_off[NST], sr_off[NST];
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    const long long r = row0 + RPI * i + lane / LPR;
    so_off[i] = (unsigned)(r * p.ldo * 2 + (lane % LPR) * 16);
    sr_off[i] = (unsigned)(r * p.ldr * 2 + (lane % LPR) * 16);
  }
  const unsigned char* stg_rd = stg + (lane / LPR) * SPITCH + (lane % LPR) * 16;
  unsigned char* stg_wr = stg + m * SPITCH + h * 8;
  u32x2 pk[8];
  u32x4 rv[NST] = {};
  auto finish = [&](f32x16 (&a)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) pk[nb * 4 + g] = u32x2{pack2(a[nb][4 * g], a[nb][4 * g + 1]), pack2(a[nb][4 * g + 2], a[nb][4 * g + 3])};
  };
  auto load_res = [&](int t) __attribute__((always_inline)) {        // t: output slice 0 .. 4
#pragma unroll
    for (int i = 0; i < NST; ++i) rv[i] = buf_load(rsr, sr_off[i] + t * XA_BN * 2, 0);
    issued += NST;
  };
  auto epilogue_part = [&](int gi, int t) __attribute__((always_inline)) {
    if (gi < 4) {
#pragma unroll
      for (int q = 2 * gi; q < 2 * gi + 2; ++q) *reinterpret_cast<u32x2*>(stg_wr + (q >> 2) * 64 + (q & 3) * 16) = pk[q];
    }
    const int i = gi - 5;
    if (i >= 0 && i < NST) {
      const u32x2 lo2 = *reinterpret_cast<const u32x2*>(stg_rd + i * RPI * SPITCH), hi2 = *reinterpret_cast<const u32x2*>(stg_rd + i * RPI * SPITCH + 8);
      u32x4 o4 = {lo2.x, lo2.y, hi2.x, hi2.y};
      float a[8], r8[8];
      unpack8(__builtin_bit_cast(uint4, o4), a);
      unpack8(__builtin_bit_cast(uint4, rv[i]), r8);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += r8[j];
      o4 = __builtin_bit_cast(u32x4, pack8(a));
      __builtin_amdgcn_raw_buffer_store_b128(o4, rso, (int)(so_off[i] + t * XA_BN * 2), 0, 0);
    }
  };
  // step 5: nothing to finish yet
  __builtin_amdgcn_s_barrier();             // every wave has its O^T fragments and its DMA shares of slices 5 / 6 in place
  mma(5, af, acc, [&](int gi) __attribute__((always_inline)) {
    if (gi <= XA_NDMA) dma_piece(7, gi);
  });
  issued += ndma;
  mark2 = issued;
  finish(acc);
  mark0 = mark1;
  mark1 = mark2;
  for (int step = 6; step < XA_NSL; ++step) {
    const int tp = step - 6;                // output slice whose epilogue rides in this step
    xwait_vm_dyn(issued - mark0);
    __builtin_amdgcn_s_barrier();
    load_res(tp);
    mma(step, af, acc, [&](int gi) __attribute__((always_inline)) {
      if (gi <= XA_NDMA) dma_piece(step + 2, gi);
      epilogue_part(gi, tp);
    });
    issued += ndma;
    mark2 = issued;
    issued += NST;
    finish(acc);
    mark0 = mark1;
    mark1 = mark2;
  }
  load_res(4);
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) epilogue_part(gi, 4);
}

}  // namespace

extern "C" int saspa_xattn_block(const SaspaXattnBlockParams* pp, void* stream) {
  if (!pp) return SASPA_EINVAL;
  const SaspaXattnBlockParams& p = *pp;
  if (!p.x || !p.residual || !p.out || !p.w || !p.bias || !p.kf || !p.vf || !p.ln_gamma || !p.ln_beta) return SASPA_EINVAL;
  if (p.M <= 0 || p.rows_per_sample <= 0 || p.nk <= 0) return SASPA_EINVAL;
  if (p.M % XA_BM || p.rows_per_sample % XA_BM || p.M % p.rows_per_sample || p.nk > 96) return SASPA_ERANGE;
  if (p.ldx < XA_K || p.ldr < XA_K || p.ldo < XA_K || p.ldw < XA_K) return SASPA_ERANGE;
  if (p.ldx % 8 || p.ldr % 8 || p.ldo % 8 || p.ldw % 8) return SASPA_EALIGN;
  if (!aligned16(p.x) || !aligned16(p.residual) || !aligned16(p.out) || !aligned16(p.w) || !aligned16(p.bias) || !aligned16(p.kf) ||
      !aligned16(p.vf) || !aligned16(p.ln_gamma) || !aligned16(p.ln_beta))
    return SASPA_EALIGN;
  const long long ld = p.ldx > p.ldo ? (p.ldx > p.ldr ? p.ldx : p.ldr) : (p.ldo > p.ldr ? p.ldo : p.ldr);
  if ((long long)p.M * ld * 2 >= 0x7fffffffLL) return SASPA_ERANGE;                       // 32-bit buffer offsets
  const long long nsamp = p.M / p.rows_per_sample;
  if (p.kf_stride < 8 * XA_KF_HEAD || p.vf_stride < 8 * XA_VF_HEAD || p.kf_stride % 16 || p.vf_stride % 16) return SASPA_ERANGE;
  if (nsamp * p.kf_stride >= 0x7fffffffLL || nsamp * p.vf_stride >= 0x7fffffffLL) return SASPA_ERANGE;
  static const int abl = getenv("SASPA_XATTN_ABLATE") ? atoi(getenv("SASPA_XATTN_ABLATE")) : 0;      // diagnostics only
  hipLaunchKernelGGL(xattn_block_kernel, dim3((unsigned)(p.M / XA_BM)), dim3(512), 0, reinterpret_cast<hipStream_t>(stream), p, abl);
  SASPA_CHECK_LAUNCH();
  return 0;
}
