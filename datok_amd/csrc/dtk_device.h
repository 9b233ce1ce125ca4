// dtk_device.h -- what the kernel units share: wave helpers, the windows macro of the walk kernels, the per-document
// repair step (used by the verification's fix kernel and by the one-block scan).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "dtk_internal.h"

#define WAVE 64
#ifndef DTK_WARM_TAG
#define DTK_WARM_TAG 64u
#endif  // how far behind a warm-up start an opening angle bracket is looked for
// knock-out builds for cost measurements (scripts/ko.sh): results are wrong, only timings mean something
#ifndef DTK_KO
#define DTK_KO 0
#endif

// ------------------------------------------------------------------ helpers

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & (WAVE - 1); }
__device__ __forceinline__ unsigned long long lanemask_lt() { return (1ull << lane_id()) - 1ull; }
__device__ __forceinline__ int highest(unsigned long long m) { return 63 - __clzll((long long)m); }
__device__ __forceinline__ uint32_t popc(unsigned long long m) { return (uint32_t)__popcll(m); }

// exclusive prefix sum over the 64 lanes; total = sum of all lanes.  DPP row shifts inside the
// rows of 16 lanes, then the two row broadcasts (lane 15 -> next row, lane 31 -> upper half):
// six adds, no LDS traffic (a __shfl_up ladder is six ds_bpermute round trips).
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t &total) {
  uint32_t x = v;
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);  // row_shr:1
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);  // row_shr:2
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);  // row_shr:4
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);  // row_shr:8
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false); // row_bcast:15 -> rows 1, 3
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false); // row_bcast:31 -> rows 2, 3
  total = (uint32_t)__builtin_amdgcn_readlane((int)x, WAVE - 1);
  return x - v;
}

__device__ __forceinline__ uint32_t doc_of(const uint64_t *__restrict__ doc_off, uint32_t lo, uint32_t hi,
                                           uint64_t g) {
  // largest d in [lo, hi) with doc_off[d] <= g   (invariant: doc_off[lo] <= g < doc_off[hi])
  while (hi - lo > 1) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (doc_off[mid] <= g) lo = mid; else hi = mid;
  }
  return lo;
}

// One thread per document: nothing to do unless a lane failed; then the lane
// `bad` started from a true state (every lane before it checked out), so where it
// really ended is the true record of its successor: redo from `bad` on.
__device__ __forceinline__ void mark_redo(const DtkSpecArgs &S, uint32_t d, uint32_t bad, uint32_t *redo_out,
                                          uint32_t *n_bad) {
  const uint32_t L0 = S.chunk_off[d], L1 = S.chunk_off[d + 1];
  DtkLaneState en = S.lane_end[bad];
  en.flags &= (LANE_F_SENT | LANE_F_TEXT | LANE_F_OK);
  if (en.p != 0xFFFFFFFFu && bad + 1 < L1) S.lane_start[bad + 1] = en;
  // (ran to EOF: no later lane has a sync point -- k_redo_spread withdraws their records)
  // The round walks again from the last lane before `bad` that owns anything: it started from a true record as
  // well, and everything a lane behind the broken link can have reported lies behind that record -- the round
  // clears from there on without having to tell true reports from false ones.
  uint32_t r0 = bad;
  if (bad > L0) {
    r0 = bad - 1u;
    while (r0 > L0 && S.lane_end[r0].p == S.lane_start[r0].p) r0--;
  }
  redo_out[d] = r0;
  atomicAdd(n_bad, 1u);
}

