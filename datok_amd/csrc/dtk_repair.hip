// dtk_repair.hip -- verification and repair of the speculative chunk lanes: link, verify,
// fix, and the per-lane / per-document / per-word kernels of a repair round.  No walking here.
#include "dtk_device.h"

// One thread per lane: is my successor's record present and not before mine?
// first_bad[d] becomes the first chunk index without such a successor (stored
// bit-inverted so that a zero fill means "none yet"); the last lane of a document
// never has one.
__global__ __launch_bounds__(256) void k_spec_link(DtkSpecArgs S) {
  if (S.go && *S.go == 0u) return;
  const uint32_t L = blockIdx.x * blockDim.x + threadIdx.x;
  if (L >= S.n_lanes) return;
  const uint32_t d = S.lane_doc[L];
  const uint32_t L0 = S.chunk_off[d], L1 = S.chunk_off[d + 1];
  bool linked = false;
  if (L + 1 < L1) {
    const uint32_t np = S.lane_start[L + 1].p, mp = S.lane_start[L].p;
    linked = np != 0xFFFFFFFFu && mp != 0xFFFFFFFFu && np >= mp;
  }
  if (!linked) atomicMax(&S.first_bad[d], ~(L - L0));  // stored inverted: zero fill = none
}

// One thread per lane: did I arrive exactly at my successor's record (position,
// state, flags)?  Lanes that did add their counts / status to the document; the
// first lane that did not is recorded (bit-inverted, zero = none) in fail_lane[d].
//
// local_link (the first pass, where every lane with a record has walked to the first sync point behind its own chunk):
// the lane decides from its own and its successor's record what k_spec_link + first_bad[d] decide otherwise -- one
// launch and a round of atomics less.  A lane with a record either has a successor record at or behind its own (then it
// must have arrived exactly there) or it is the chain's last lane (then it must have reached EOF, and no later lane
// may have a record: the lane in front of such a record, which has none itself, finds the chain's last lane by
// walking back).  The first lane that fails is the same one in both formulations.
// (256 threads.  With 1024 -- 16 waves adding up before their atomics -- one long document verified faster still, but
//  with three batches in flight the blocks have to wait for four free wave slots on every SIMD of one CU, which the
//  other batches' walks, six waves per SIMD, rarely leave: 154 -> 143 GB/s.)
#define VERIFY_TB 256u
__global__ __launch_bounds__(VERIFY_TB) void k_spec_verify(DtkWalkArgs A, DtkSpecArgs S, uint32_t cmp_mask, uint32_t local_link) {
  if (S.go && *S.go == 0u) return;
  const uint32_t L = blockIdx.x * blockDim.x + threadIdx.x;
  bool live = L < S.n_lanes;  // every lane stays for the wave reduction below
  uint32_t d = 0xFFFFFFFFu;
  DtkLaneCount c{0u, 0u, 0u, 0u, 0u, 0xFFFFFFFFu, 0u, 0u};
  if (live) {
    d = S.lane_doc[L];
    if (S.redo_from && S.redo_from[d] == 0xFFFFFFFFu) { live = false; d = 0xFFFFFFFFu; }  // a repair round: not this document
  }
  if (live && local_link) {
    const uint32_t L0 = S.chunk_off[d], L1 = S.chunk_off[d + 1];
    const bool has_next = L + 1u < L1;
    const DtkLaneState own = S.lane_start[L], en = S.lane_end[L];
    DtkLaneState nx{0xFFFFFFFFu, 0u, 0u, 0u};
    if (has_next) nx = S.lane_start[L + 1];
    uint32_t link_ok = 1u;
    if (own.p == 0xFFFFFFFFu) {
      // behind the chain: legitimate only if the chain ran to EOF and no later lane found a sync point
      if (nx.p != 0xFFFFFFFFu) {
        uint32_t x = L - 1u;  // (lane 0 of a document always has a record)
        while (x > L0 && S.lane_start[x].p == 0xFFFFFFFFu) x--;
        atomicMax(&S.fail_lane[d], ~x);
      }
    } else {
      bool good;
      if (nx.p != 0xFFFFFFFFu && nx.p >= own.p) {
        good = en.p == nx.p && en.t == nx.t && ((en.flags ^ nx.flags) & cmp_mask) == 0 && !(en.flags & LANE_F_DROPPED);
      } else {
        // the chain's last lane must reach EOF (a successor record before my own breaks the chain right here)
        good = nx.p == 0xFFFFFFFFu && en.p == 0xFFFFFFFFu && !(en.flags & LANE_F_IDLE);
      }
      if (good) c = S.lane_cnt[L]; else atomicMax(&S.fail_lane[d], ~L);
      link_ok = good ? 1u : 0u;
    }
    S.lane_plan[L].pad = link_ok;
  } else if (live) {
    const uint32_t L0 = S.chunk_off[d];
    const uint32_t k = L - L0, fb = ~S.first_bad[d];
    uint32_t link_ok = 1u;  // did I arrive exactly at my successor's record (k_redo_spread reads it)
    if (k > fb) {
      // a lane behind the chain: legitimate only if the chain ran to EOF and I found no sync point
      if (S.lane_start[L].p != 0xFFFFFFFFu) atomicMax(&S.fail_lane[d], ~(L0 + fb));
    } else {
      const DtkLaneState en = S.lane_end[L];
      bool good;
      if (k < fb) {
        const DtkLaneState nx = S.lane_start[L + 1];
        good = en.p == nx.p && en.t == nx.t && ((en.flags ^ nx.flags) & cmp_mask) == 0 &&
               !(en.flags & LANE_F_DROPPED);
      } else {
        good = en.p == 0xFFFFFFFFu && !(en.flags & LANE_F_IDLE);  // the chain's last lane must reach EOF
      }
      if (good) c = S.lane_cnt[L]; else atomicMax(&S.fail_lane[d], ~L);
      link_ok = good ? 1u : 0u;
    }
    S.lane_plan[L].pad = link_ok;
  }
  // The lanes of a document are consecutive: add up the counts of each run of equal
  // documents inside the wave, then one atomic per run instead of one per lane.
  uint32_t tok = c.tok, sent = c.sent, text = c.text, st = c.status;
#pragma unroll
  for (int o = 1; o < WAVE; o <<= 1) {
    const uint32_t dn = __shfl_down(d, o);
    const uint32_t t2 = __shfl_down(tok, o), s2 = __shfl_down(sent, o), x2 = __shfl_down(text, o),
                   st2 = __shfl_down(st, o);
    const bool same = (lane_id() + o < WAVE) && dn == d;
    tok += same ? t2 : 0u; sent += same ? s2 : 0u; text += same ? x2 : 0u; st |= same ? st2 : 0u;
  }
  const uint32_t dprev = __shfl_up(d, 1);
  const bool head = live && (lane_id() == 0 || dprev != d);
  // A block whose waves all lie inside one document adds up once more: the waves of a long document otherwise
  // queue their atomics at the same three addresses (one 64 MiB document: 4096 waves, 106 us of verification).
  __shared__ uint32_t s_d[VERIFY_TB / WAVE], s_v[VERIFY_TB / WAVE][4];
  const uint32_t wid = threadIdx.x >> 6;
  const bool whole = __ballot(head) == 1ull;  // one run of lanes, starting at lane 0
  if (lane_id() == 0) {
    s_d[wid] = whole ? d : 0xFFFFFFFFu;
    s_v[wid][0] = tok; s_v[wid][1] = sent; s_v[wid][2] = text; s_v[wid][3] = st;
  }
  __syncthreads();
  bool merged = s_d[0] != 0xFFFFFFFFu;
#pragma unroll
  for (uint32_t w = 1; w < VERIFY_TB / WAVE; w++) merged = merged && s_d[w] == s_d[0];
  if (merged) {
    if (threadIdx.x == 0) {
      uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
      for (uint32_t w = 0; w < VERIFY_TB / WAVE; w++) { a0 += s_v[w][0]; a1 += s_v[w][1]; a2 += s_v[w][2]; a3 |= s_v[w][3]; }
      if (a0) atomicAdd((unsigned long long *)&A.tok_cnt[d], (unsigned long long)a0);
      if (a1) atomicAdd((unsigned long long *)&A.sent_cnt[d], (unsigned long long)a1);
      if (a2) atomicAdd((unsigned long long *)&A.text_cnt[d], (unsigned long long)a2);
      if (a3) atomicOr(&A.status[d], a3);
    }
  } else if (head) {
    if (tok) atomicAdd((unsigned long long *)&A.tok_cnt[d], (unsigned long long)tok);
    if (sent) atomicAdd((unsigned long long *)&A.sent_cnt[d], (unsigned long long)sent);
    if (text) atomicAdd((unsigned long long *)&A.text_cnt[d], (unsigned long long)text);
    if (st) atomicOr(&A.status[d], st);
  }
}

__global__ __launch_bounds__(256) void k_spec_fix(DtkWalkArgs A, DtkSpecArgs S, uint32_t *redo_out,
                                                  uint32_t *n_bad) {
  if (S.go && *S.go == 0u) return;
  const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= A.n_docs) return;
  const uint32_t bad = ~S.fail_lane[d];
  if (bad == 0xFFFFFFFFu) { redo_out[d] = 0xFFFFFFFFu; return; }
  mark_redo(S, d, bad, redo_out, n_bad);
}

// ---- repair rounds (a document whose chain broke is redone from the last owning lane before its first bad lane on).
// All per lane / per document / per bitmap word, so that a long document repairs as fast
// as a batch of short ones:
//   k_redo_spread : the first bad lane started from a true state, so k_spec_fix made its end the
//                   record of its successor.  Further down the document, a lane whose predecessor
//                   arrived exactly but which itself missed its successor is in the same position
//                   with high probability: its end becomes its successor's record too (speculation
//                   again -- the next verification decides), so that one round repairs all isolated
//                   misses of a document, not just the first.
//   k_redo_reset  : per document, the counters and check words the round re-derives, and the tail word.
//   k_redo_clear  : event bits behind the record the round walks from.
// then k_spec_link, k_spec_walk (redone lanes only), k_spec_verify (repaired documents only), k_spec_fix.
__global__ __launch_bounds__(256) void k_redo_spread(DtkSpecArgs S) {
  if (S.go && *S.go == 0u) return;
  const uint32_t L = blockIdx.x * blockDim.x + threadIdx.x;
  if (L >= S.n_lanes) return;
  const uint32_t d = S.lane_doc[L];
  if (S.redo_from[d] == 0xFFFFFFFFu) return;
  const uint32_t bad = ~S.fail_lane[d];  // the first lane that missed (k_redo_reset clears the word afterwards)
  if (L <= bad) return;
  const uint32_t L1 = S.chunk_off[d + 1];
  if (S.lane_end[bad].p == 0xFFFFFFFFu) {  // the first bad lane ran to EOF: no later lane has a sync point
    S.lane_start[L].p = 0xFFFFFFFFu;
    return;
  }
  if (L + 1 >= L1) return;
  DtkLaneState en = S.lane_end[L];
  if (S.lane_plan[L].pad == 0u && S.lane_plan[L - 1].pad != 0u && L - 1 != bad &&
      en.p != 0xFFFFFFFFu && !(en.flags & LANE_F_IDLE)) {  // (a lane that overshot dropped events: still a true end)
    en.flags &= (LANE_F_SENT | LANE_F_TEXT | LANE_F_OK);
    S.lane_start[L + 1] = en;
  }
}

__global__ __launch_bounds__(256) void k_redo_reset(DtkWalkArgs A, DtkSpecArgs S) {
  if (S.go && *S.go == 0u) return;
  const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= A.n_docs || S.redo_from[d] == 0xFFFFFFFFu) return;
  A.tok_cnt[d] = 0; A.sent_cnt[d] = 0; A.text_cnt[d] = 0; A.status[d] = 0;
  S.first_bad[d] = 0; S.fail_lane[d] = 0;
  A.doc_tail[d] = 0;
}

// One thread per bitmap word.  In a repaired document every bit behind the record the round walks from is
// cleared; at that very position only the opening kinds are (the closing kinds there were reported by the lane
// that stopped at it: a true report that nobody makes again).
__global__ __launch_bounds__(256) void k_redo_clear(DtkWalkArgs A, DtkSpecArgs S) {
  if (S.go && *S.go == 0u) return;
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= A.bit_words) return;
  const uint64_t G0 = 32ull * j, G1 = G0 + 32ull;
  // the last document that starts at or before bit G0
  uint32_t lo = 0, hi = A.n_docs;
  while (hi - lo > 1) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (DTK_EV_BIT(A.doc_off[mid], mid) <= G0) lo = mid; else hi = mid;
  }
  uint32_t m_all = 0, m_open = 0;
  for (uint32_t d = lo; d < A.n_docs; d++) {
    const uint64_t key = DTK_EV_BIT(A.doc_off[d], d);
    if (key >= G1) break;
    const uint32_t r0 = S.redo_from[d];
    if (r0 == 0xFFFFFFFFu) continue;
    const uint32_t from = S.lane_start[r0].p;
    if (from == 0xFFFFFFFFu) continue;
    const uint64_t Gf = key + from, Ge = DTK_EV_BIT(A.doc_off[d + 1], d + 1);  // bits (Gf, Ge) and, opening kinds, Gf
    const uint64_t a = Gf + 1 > G0 ? Gf + 1 : G0, b = Ge < G1 ? Ge : G1;
    if (a < b) {
      const uint32_t n = (uint32_t)(b - a), sh = (uint32_t)(a - G0);
      const uint32_t m = (n >= 32u ? 0xFFFFFFFFu : ((1u << n) - 1u)) << sh;
      m_all |= m; m_open |= m;
    }
    if (Gf >= G0 && Gf < G1) m_open |= 1u << (uint32_t)(Gf - G0);
  }
  if (m_all) {
    A.bits[EVB_END * A.bit_words + j] &= ~m_all;
    A.bits[EVB_TEOT * A.bit_words + j] &= ~m_all;
    A.bits[EVB_SEOT * A.bit_words + j] &= ~m_all;
  }
  if (m_open) {
    A.bits[EVB_START * A.bit_words + j] &= ~m_open;
    A.bits[EVB_SEPS * A.bit_words + j] &= ~m_open;
  }
}

// ---------------------------------------------------------------- launchers

// the stages of dtk_launch_spec (dtk_walk.hip) that do not walk: 1 link, 3 verify, 7 link + verify, 4 fix, 5 spread + reset
extern "C" int dtk_launch_spec_check(const DtkWalkArgs *args, const DtkSpecArgs *spec, int stage, uint32_t cmp_mask,
                                     uint32_t *redo_out, uint32_t *n_bad, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  const uint32_t lane_blocks256 = (spec->n_lanes + 255) / 256;
  const uint32_t doc_blocks = (args->n_docs + 255) / 256;
  switch (stage) {
    case 1:
      hipLaunchKernelGGL(k_spec_link, dim3(lane_blocks256), dim3(256), 0, s, *spec);
      return (int)hipGetLastError();
    case 3:
      hipLaunchKernelGGL(k_spec_verify, dim3((spec->n_lanes + VERIFY_TB - 1u) / VERIFY_TB), dim3(VERIFY_TB), 0, s, *args, *spec, cmp_mask, 0u);
      return (int)hipGetLastError();
    case 7:  // first pass behind k_spec_both: link + verify in one
      hipLaunchKernelGGL(k_spec_verify, dim3((spec->n_lanes + VERIFY_TB - 1u) / VERIFY_TB), dim3(VERIFY_TB), 0, s, *args, *spec, cmp_mask, 1u);
      return (int)hipGetLastError();
    case 4:
      hipLaunchKernelGGL(k_spec_fix, dim3(doc_blocks), dim3(256), 0, s, *args, *spec, redo_out, n_bad);
      return (int)hipGetLastError();
    case 5:  // repair round, before link / walk / verify / fix
      hipLaunchKernelGGL(k_redo_spread, dim3(lane_blocks256), dim3(256), 0, s, *spec);
      hipLaunchKernelGGL(k_redo_reset, dim3(doc_blocks), dim3(256), 0, s, *args, *spec);
      return (int)hipGetLastError();
  }
  return -1;
}

extern "C" int dtk_launch_redo_clear(const DtkWalkArgs *args, const DtkSpecArgs *spec, void *stream) {
  if (args->bit_words == 0) return 0;
  hipLaunchKernelGGL(k_redo_clear, dim3((args->bit_words + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, *args, *spec);
  return (int)hipGetLastError();
}

