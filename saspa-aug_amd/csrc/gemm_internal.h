// Internal (not exported) entry points shared between the GEMM translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "saspa_hip.h"

// 8-wave 256 x (64*fn) ping-pong kernel (saspa_gemm_pp.hip).  bf16, "fast" operand layout only
// (see saspa_gemm.hip); fn in {4, 5}.  Returns SASPA_ERANGE if the problem is not eligible.
__attribute__((visibility("hidden"))) int saspa_gemm_pp_launch(const SaspaGemmParams& p, hipStream_t s, int ksplit, int fn);
__attribute__((visibility("hidden"))) bool saspa_gemm_pp_eligible(const SaspaGemmParams& p);
// split-K reduce + epilogue launch shared by both variants (saspa_gemm.hip)
__attribute__((visibility("hidden"))) int saspa_gemm_splitk_reduce(const SaspaGemmParams& p, hipStream_t s, int ksplit);
// N-partitioned tile order (weight-heavy problems): nbn / 8 if the launch should use it, else 0 (saspa_gemm.hip)
__attribute__((visibility("hidden"))) int saspa_gemm_npart8(const SaspaGemmParams& p, int BM, int BN, int G, int tiles);
// wave-specialised 8-wave kernel for short-K bf16 layers (saspa_gemm_ws.hip)
__attribute__((visibility("hidden"))) int saspa_gemm_ws_launch(const SaspaGemmParams& p, hipStream_t s);
__attribute__((visibility("hidden"))) bool saspa_gemm_ws_eligible(const SaspaGemmParams& p);
// A-stationary kernel for K = 320 pointwise layers (saspa_gemm_as.hip): fused LayerNorm, transposed second output
__attribute__((visibility("hidden"))) int saspa_gemm_as_launch(const SaspaGemmParams& p, hipStream_t s);
__attribute__((visibility("hidden"))) bool saspa_gemm_as_ok(const SaspaGemmParams& p);
// channel slabs of the GroupNorm statistics pass for C / 8 chunks (saspa_norm.hip: layout of SaspaGroupNormParams.partial)
__attribute__((visibility("hidden"))) int saspa_gn_slabs(int c8);
