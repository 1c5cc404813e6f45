// Internal (not exported) entry points shared between the GEMM translation units.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>

#include "saspa_hip.h"

// 8-wave 256 x (64*fn) ping-pong kernel (saspa_gemm_pp.hip).  bf16, "fast" operand layout only
// (see saspa_gemm.hip); fn in {4, 5}.  Returns SASPA_ERANGE if the problem is not eligible.
// Dry dispatch (ABI 20, saspa_gemm_which): with `on` set for the calling thread, the kernel launchers below record which kernel family
// (SASPA_GEMM_TILED / WIDE / WS / AS = SaspaGemmParams.variant codes) and how many K slices dispatch() chose and return WITHOUT launching.
struct SaspaDryRun { bool on; int family; int ksplit; };
__attribute__((visibility("hidden"))) SaspaDryRun* saspa_dry_state();     // the calling thread's (saspa_gemm.hip)
#define SASPA_DRY_RETURN(fam_, ks_) do { SaspaDryRun* d_ = saspa_dry_state(); if (d_->on) { d_->family = (fam_); d_->ksplit = (ks_); return 0; } } while (0)
__attribute__((visibility("hidden"))) int saspa_gemm_pp_launch(const SaspaGemmParams& p, hipStream_t s, int ksplit, int fn);
__attribute__((visibility("hidden"))) bool saspa_gemm_pp_eligible(const SaspaGemmParams& p);
// split-K reduce + epilogue launch shared by both variants (saspa_gemm.hip)
__attribute__((visibility("hidden"))) int saspa_gemm_splitk_reduce(const SaspaGemmParams& p, hipStream_t s, int ksplit);
// N-partitioned tile order (weight-heavy problems): nbn / 8 if the launch should use it, else 0 (saspa_gemm.hip)
__attribute__((visibility("hidden"))) int saspa_gemm_npart8(const SaspaGemmParams& p, int BM, int BN, int G, int tiles);
// Persistent grid of a one-workgroup-per-CU kernel (round 5): `tiles` work items over at most `cap` workgroups.  With tiles just above
// a whole number of rounds (352 row tiles of the 64x88 level at 512x704: 1.375 rounds) a grid of `cap` workgroups makes a few of
// them run one tile more than the rest while the others idle -- and every tile pays the contention of a full chip.  A grid of
// ceil(tiles / rounds) workgroups gives every workgroup the SAME number of tiles on fewer CUs: 176 workgroups x 2 tiles run the
// level-0 conv of that bucket in 149 us against 174 (tools/msplit_704.py: fewer CUs store at once and the power limit leaves them
// a higher clock).  SASPA_GEMM_BALANCE=0 = the old rule (A/B).
static inline int saspa_balanced_grid(int tiles, int cap) {
  if (cap < 1) cap = 1;
  if (tiles <= cap) return tiles;
  static const bool off = getenv("SASPA_GEMM_BALANCE") && atoi(getenv("SASPA_GEMM_BALANCE")) == 0;
  if (off) return cap;
  const int rounds = (tiles + cap - 1) / cap;
  return (tiles + rounds - 1) / rounds;
}

// wave-specialised 8-wave kernel for short-K bf16 layers (saspa_gemm_ws.hip)
__attribute__((visibility("hidden"))) int saspa_gemm_ws_launch(const SaspaGemmParams& p, hipStream_t s);
__attribute__((visibility("hidden"))) bool saspa_gemm_ws_eligible(const SaspaGemmParams& p);
// A-stationary kernel for K = 320 pointwise layers (saspa_gemm_as.hip): fused LayerNorm, transposed second output
__attribute__((visibility("hidden"))) int saspa_gemm_as_launch(const SaspaGemmParams& p, hipStream_t s);
__attribute__((visibility("hidden"))) bool saspa_gemm_as_ok(const SaspaGemmParams& p);
extern "C" int saspa_gemm_as_auto(const SaspaGemmParams* p);
// channel slabs of the GroupNorm statistics pass for C / 8 chunks (saspa_norm.hip: layout of SaspaGroupNormParams.partial)
__attribute__((visibility("hidden"))) int saspa_gn_slabs(int c8);
