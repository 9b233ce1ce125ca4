// dtk_walk.hip -- gfx950 (MI355X, wave64) kernel 2 of the batch tokenizer: the walk.
//
// Pipeline per batch (one HIP stream, no host round trip in the common case), one unit per stage:
//   dtk_symbolize.hip  bytes -> one code per byte of the symbol stream (UTF-8 decode with Go's DecodeRune rules + sigma
//                      lookup) + the rune-start bitmap; clears the run's event bitmaps and accumulators on the way.
//   dtk_walk.hip       the FSA transition walk of matrix.go:348-698 / datok.go:781-1135 (dtk_walk_core.h), one chunk
//                      of a document per lane, speculative starts; one bit per event and cursor position.
//   dtk_repair.hip     proves that every lane arrived exactly at its successor's start, repairs what did not.
//   dtk_compact.hip    event bitmaps -> the offset arrays NewTokenWriter (token_writer.go:36-175) would have collected;
//                      the scan that sizes the CSR rows; results into page-locked host memory.
//   dtk_render.hip     NewTokenWriter's bytes for all 16 writer modes (optional).
//
// Integer table lookups only: no MFMA.
#include "dtk_walk_core.h"

// ---- the exact pass: one lane per irregular document ----
//
// NewTokenWriter (token_writer.go:36-175) with TOKEN_POS | SENTENCE_POS semantics, fed by the walk in the
// reference's own call order: what the compaction derives from position-indexed event bytes for every
// other document.  Writes the document's rows of the result arrays and, call by call, the list a closure
// replay needs (kind 0 Token: a = byte position of buffer[0], b = of buffer[offset], c = end;
// kind 1 SentenceEnd(a); kind 2 TextEnd(a) -- the int arguments as upstream: matrix.go:575,597,600,684,691
// pass buffc, datok.go:1015,1026,1119,1127 pass 0 and only :1023 buffc).
struct ExactSink {
  DtkSymAt s;          // the document's stretch of the symbol stream (rune starts)
  const uint8_t *txt;  // the document's bytes
  bool nl_rule, write;
  DtkCall *log;
  int32_t *rstart, *rend, *sent;
  uint32_t *bstart, *bend, *ttok, *tsent, *sbefore, *ts_end;
  uint32_t tok_n, sent_n, text_n;  // row lengths
  // token_writer.go:38-42
  int32_t posC, last_rend;
  bool init, sentB;
  uint32_t n_tok, n_sent, n_text, n_sev, text_tok0, n_calls;
  uint32_t st;
  __device__ __forceinline__ void start() {
    posC = 0; last_rend = 0; init = true; sentB = true;
    n_tok = n_sent = n_text = n_sev = text_tok0 = n_calls = 0; st = 0;
  }
  __device__ __forceinline__ int32_t runes(uint32_t from, uint32_t to) const {
    return to > from ? (int32_t)count_runes(s, from, to) : 0;
  }
  __device__ __forceinline__ void call(uint32_t kind, int32_t a, uint32_t b, uint32_t c) {
    if (write) log[n_calls] = DtkCall{kind, a, b, c};
    n_calls++;
  }
  __device__ __forceinline__ void push_sent(int32_t v) {
    if (write) { if (n_sent < sent_n) sent[n_sent] = v; else st |= ST_INTERNAL; }
    n_sent++;
  }
  __device__ __forceinline__ void sentence_end(int32_t arg) {  // token_writer.go:104-115
    call(1u, arg, 0u, 0u);
    n_sev++;
    if (n_tok == text_tok0) st |= ST_EMPTY_TEXT; else push_sent(last_rend);
    sentB = true;
  }
  __device__ __forceinline__ void text_end(int32_t arg) {  // token_writer.go:131-159
    call(2u, arg, 0u, 0u);
    if (n_tok == text_tok0) st |= ST_EMPTY_TEXT;
    if (write) {
      if (n_text < text_n) {
        ttok[n_text] = n_tok; tsent[n_text] = n_sent;
        if (ts_end) ts_end[n_text] = n_sev;
      } else st |= ST_INTERNAL;
    }
    n_text++;
    sentB = true; posC = 0; text_tok0 = n_tok;
  }
  template <bool IS_MATRIX>
  __device__ __forceinline__ void token(uint32_t bs, uint32_t tp, uint32_t p, bool) {  // token_writer.go:58-88
    call(0u, (int32_t)bs, tp, p);
    const int32_t off_r = runes(bs, tp), len_r = runes(bs, p);  // offset, len(buf)
    if (posC == 0 && nl_rule && p > bs && txt[bs] == '\n' && !init) posC--;  // :66-68
    init = false;
    posC += off_r;
    const int32_t rs = posC;
    if (sentB) { sentB = false; push_sent(rs); }
    posC += len_r - off_r;
    last_rend = posC;
    if (write) {
      if (n_tok < tok_n) {
        if (rstart) { rstart[n_tok] = rs; rend[n_tok] = posC; }
        if (bstart) { bstart[n_tok] = tp < p ? tp : p; bend[n_tok] = p; }  // an empty surface: the empty range at the end of the buffer
        if (sbefore) sbefore[n_tok] = n_sev;
      } else st |= ST_INTERNAL;
    }
    n_tok++;
  }
  template <bool IS_MATRIX>
  __device__ __forceinline__ void eot(uint32_t bs, uint32_t p, bool with_sentence, bool) {
    const int32_t buffc = runes(bs, p);
    if (with_sentence) sentence_end(buffc);      // matrix.go:597 / datok.go:1023
    text_end(IS_MATRIX ? buffc : 0);             // matrix.go:600 / datok.go:1026
  }
  template <bool IS_MATRIX>
  __device__ __forceinline__ void sentence(uint32_t bs, uint32_t p, bool) {
    sentence_end(IS_MATRIX ? runes(bs, p) : 0);  // matrix.go:575 / datok.go:1015
  }
  template <bool IS_MATRIX>
  __device__ __forceinline__ void tail(uint32_t bs, uint32_t p, bool sentence_end_, bool text_end_, bool) {
    const int32_t arg = IS_MATRIX ? runes(bs, p) : 0;
    if (!sentence_end_) sentence_end(arg);       // matrix.go:683-684 / datok.go:1118-1119
    if (!text_end_) text_end(arg);               // matrix.go:690-691 / datok.go:1126-1127
  }
  __device__ __forceinline__ void out_of_order() {}  // (call order is what this sink records)
};

template <typename TRANS, bool IS_MATRIX>
__global__ __launch_bounds__(WAVE) void k_exact_doc(TRANS tr, DtkExactArgs X, uint32_t epsilon, uint32_t unknown,
                                                    uint32_t identity) {
  __shared__ uint16_t s_win[WAVE * DTK_WIN_ROW];
  uint16_t *win_row = s_win + threadIdx.x * DTK_WIN_ROW;
  const uint32_t i = blockIdx.x * WAVE + threadIdx.x;
  if (i >= X.n) return;
  const uint32_t d = X.docs[i];
  const uint64_t off = X.doc_off[d];
  const uint32_t len = (uint32_t)(X.doc_off[d + 1] - off);
  ExactSink sink;
  sink.s = DtkSymAt{X.sym, off}; sink.txt = X.text + off;
  sink.nl_rule = (X.flags & 16u) != 0; sink.write = X.pass != 0;
  sink.log = X.pass ? X.calls + X.call_off[i] : nullptr;
  const uint64_t t0 = X.tok_off[d], s0 = X.sent_off[d], x0 = X.text_off[d];
  sink.rstart = X.tok_rstart ? X.tok_rstart + t0 : nullptr; sink.rend = X.tok_rend ? X.tok_rend + t0 : nullptr;
  sink.bstart = X.tok_bstart ? X.tok_bstart + t0 : nullptr; sink.bend = X.tok_bend ? X.tok_bend + t0 : nullptr;
  sink.sbefore = X.tok_sbefore ? X.tok_sbefore + t0 : nullptr;
  sink.sent = X.sent + s0;
  sink.ttok = X.text_tok_end + x0; sink.tsent = X.text_sent_end + x0;
  sink.ts_end = X.text_s_end ? X.text_s_end + x0 : nullptr;
  sink.tok_n = (uint32_t)(X.tok_off[d + 1] - t0); sink.sent_n = (uint32_t)(X.sent_off[d + 1] - s0);
  sink.text_n = (uint32_t)(X.text_off[d + 1] - x0);
  sink.start();
  DtkLaneState init{0u, tr.start_state(), tr.start_aux(), 0u}, fin;
  uint32_t st = 0, steps = 0;
  walk_lane<TRANS, IS_MATRIX, MODE_DOC, ExactSink>(tr, X.sym, off, len, init, 0u, sink, epsilon, unknown, identity,
                                                   step_cap(X.step_factor, len), fin, st, steps, win_row);
  if (X.pass == 0) { X.n_calls[i] = sink.n_calls; return; }
  // the rows were sized by the first walk's counts: both walks make the same calls
  if (sink.n_tok != sink.tok_n || sink.n_sent != sink.sent_n || sink.n_text != sink.text_n) st |= ST_INTERNAL;
  X.status[d] = st | sink.st;
  if (X.doc_ns) X.doc_ns[d] = sink.n_sev;
}

// ---- one document per lane (no speculation) ----
template <typename TRANS, bool IS_MATRIX>
__global__ __launch_bounds__(WAVE) void k_walk_doc(TRANS tr, DtkWalkArgs A, uint32_t epsilon,
                                                   uint32_t unknown, uint32_t identity) {
  DTK_WINDOWS(TRANS, A.sym)
  const uint32_t d = blockIdx.x * WAVE + threadIdx.x;
  uint32_t steps = 0;
  if (d < A.n_docs) {
    const uint64_t off = A.doc_off[d];
    const uint32_t len = (uint32_t)(A.doc_off[d + 1] - off);
    EventSink sink;
    sink.init(A, off, d, 0u, 0xFFFFFFFFu);  // (documents of any length: the bits go straight to memory)
    DtkLaneState init{0u, tr.start_state(), tr.start_aux(), 0u}, fin;
    uint32_t st;
    walk_any<TRANS, IS_MATRIX, MODE_DOC>(tr, A.sym, off, len, init, 0u, sink, epsilon, unknown,
                                          identity, step_cap(A.step_factor, len), fin, st, steps, win_row, s_lut);
    A.status[d] = st | sink.st;
    A.tok_cnt[d] = sink.c_tok; A.sent_cnt[d] = sink.c_sent; A.text_cnt[d] = sink.c_text;
  }
  add_steps(A.steps, steps);
}

// ---- speculative chunk lanes ----
//
// Lane (d, k) covers the rewinds ("sync points": the moments the reference
// rewinds its window, where the whole loop state is (position, state, three
// flags)) that fall into [k*C, (k+1)*C) of document d.
//   k_spec_start : lane k >= 1 walks from k*C - W in the start state and records
//                  the first sync point at or after k*C  (the automaton
//                  re-synchronises within a token or two).
//   k_spec_link  : per document, the first lane without a linked successor.
//   k_spec_walk  : each lane walks from its record to the next lane's record,
//                  storing events inside its window only.
//   k_spec_verify: per lane, verifies that it arrived exactly at its
//                  successor's record (position, state, flags).  A document that
//                  fails is repaired from the first bad lane on (host loop,
//                  normally never entered) -- the result is exact either way.

#ifdef DTK_PROBE
__shared__ unsigned long long s_probe_mark;
__device__ unsigned long long g_phase[8];  // cycles per wave: prologue+search, warm-up walk, chunk walk, epilogue; waves
extern "C" int dtk_phase_read(unsigned long long *out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(g_phase));
  if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)); }
  return 0;
}
#endif

// the start record of lane k >= 0 of document d (k_spec_start, k_spec_both)
template <typename TRANS, bool IS_MATRIX>
__device__ __forceinline__ DtkLaneState start_record(const TRANS &tr, const DtkWalkArgs &A, const DtkSpecArgs &S,
                                                     uint32_t k, uint64_t off, uint32_t len, uint32_t epsilon,
                                                     uint32_t unknown, uint32_t identity, uint16_t *win_row,
                                                     const uint16_t *s_lut, uint32_t &steps) {
  DtkLaneState rec{0u, tr.start_state(), tr.start_aux(), 0u};
  if (k > 0) {
    const uint32_t kc = k * S.chunk;
    uint32_t sp = kc > S.warm ? kc - S.warm : 0u;
    if (sp > 0 && S.warm_ws && S.text) {
      // The walk re-synchronises at token boundaries, and blanks are boundaries in every
      // tokenizer of this kind: start behind the warm_ws-th run of blanks before the chunk
      // instead of a fixed distance (never further back than `warm`).  A wrong guess only
      // costs a repair round.
      const uint8_t *tx = S.text + off;
      uint32_t runs = 0, q = kc > S.warm_min ? kc - S.warm_min : 0u;
      if (q < sp) q = sp;
      bool in_ws = false;
      while (q > sp) {
        const uint8_t c = tx[q - 1u];
        const bool ws = c == ' ' || c == '\n' || c == '\t' || c == '\r';
        if (in_ws && !ws) { if (++runs == S.warm_ws) break; }
        in_ws = ws;
        q--;
      }
      sp = q;  // first byte of the run of blanks (the walk skips them), or the fixed start
    }
    if (sp > 0 && S.warm_extend && S.text) {
      // A start inside a long blank-free token (a URL, say) makes the warm-up invent token ends the
      // real walk does not have -- a repair round.  Blanks are token boundaries in every tokenizer of
      // this kind: move the start back to the previous blank (typically half a dozen bytes; at
      // most warm_extend).
      // (8 bytes per step: the dwords around them are read whole, a blank is found with byte-wise
      // zero tests on  bytes ^ pattern)
      const uint8_t *tx = S.text + off;
      // bytes tx[q-8 .. q-1] as one 64-bit word, tx[q-1] on top
      auto load8 = [&](uint32_t q) -> uint64_t {
        if (off + q >= 16u) {
          const uint64_t a0 = reinterpret_cast<uint64_t>(tx + q) - 8u;  // address of the first of the 8 bytes
          const uint32_t sh = (uint32_t)(a0 & 3u) * 8u;
          const uint32_t *wp = reinterpret_cast<const uint32_t *>(a0 & ~3ull);
          const uint64_t lo64 = (uint64_t)wp[0] | ((uint64_t)wp[1] << 32);
          return sh ? (lo64 >> sh) | ((uint64_t)wp[2] << (64u - sh)) : lo64;
        }
        uint64_t w = 0;  // the first bytes of the batch: one by one
        for (uint32_t k = 0; k < 8u; k++)
          if (q + k >= 8u) w |= (uint64_t)tx[q + k - 8u] << (8u * k);
        return w;
      };
      auto zb = [](uint64_t x) {
        return ~(((x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | x) & 0x8080808080808080ull;
      };
      const uint32_t lim = sp > S.warm_extend ? sp - S.warm_extend : 0u;
      uint32_t q = sp;
      while (q > lim) {
        const uint64_t w = load8(q);
        uint64_t m = zb(w ^ 0x2020202020202020ull) | zb(w ^ 0x0A0A0A0A0A0A0A0Aull) |
                     zb(w ^ 0x0909090909090909ull) | zb(w ^ 0x0D0D0D0D0D0D0D0Dull);
        if (q < 8u) m &= ~0ull << ((8u - q) * 8u);  // bytes before the document do not count
        if (m) {  // start behind the last blank
          q = q - 8u + (7u - ((uint32_t)__clzll((long long)m) >> 3)) + 1u;
          break;
        }
        q = q >= 8u ? q - 8u : 0u;
      }
      if (q < lim) q = lim;
      sp = q;
      // Inside a markup tag (<a href="...">, <!-- a comment -->) blanks are no token boundaries: if the nearest
      // angle bracket behind the start is an opening one, the warm-up starts at it (text with a tag every few hundred
      // bytes otherwise costs a repair round in every batch).  At most DTK_WARM_TAG bytes back.
      // (all loads first: one after the other they cost a cache round trip each, 8 us per batch on plain text,
      //  where no bracket ends the search early)
      uint64_t wt[DTK_WARM_TAG / 8u];
#pragma unroll
      for (uint32_t i = 0; i < DTK_WARM_TAG / 8u; i++) wt[i] = sp > 8u * i ? load8(sp - 8u * i) : 0ull;
#pragma unroll
      for (uint32_t i = 0; i < DTK_WARM_TAG / 8u; i++) {
        const uint32_t qq = sp > 8u * i ? sp - 8u * i : 0u;  // wt[i] = bytes qq-8 .. qq-1
        uint64_t mo = zb(wt[i] ^ 0x3C3C3C3C3C3C3C3Cull), mc = zb(wt[i] ^ 0x3E3E3E3E3E3E3E3Eull);
        if (qq < 8u) { const uint64_t in = qq ? ~0ull << ((8u - qq) * 8u) : 0ull; mo &= in; mc &= in; }
        if (mo | mc) {
          if (mo > mc) sp = qq - 8u + (7u - ((uint32_t)__clzll((long long)mo) >> 3));  // the nearest one opens a tag
          break;
        }
      }
    }
#ifdef DTK_PROBE
    s_probe_mark = clock64();  // (the wave is in lockstep: every lane that comes here writes the same time)
#endif
    EventSink sink;  // (a warm-up reports nothing)
    sink.g = nullptr; sink.lds = (dtk_lds_u32 *)nullptr; sink.tailw = nullptr; sink.lw = 0; sink.lo = sink.hi = 0;
    uint32_t st;
    if (sp > 0) {
      while (sp < len && !dtk_sym_is_start(A.sym, off + sp)) sp++;
      DtkLaneState init{sp, tr.start_state(), tr.start_aux(), 0u};
      walk_any<TRANS, IS_MATRIX, MODE_START>(tr, A.sym, off, len, init, kc, sink, epsilon, unknown,
                                              identity, step_cap(A.step_factor, len), rec, st, steps, win_row, s_lut);
    } else {
      // sp == 0: the walk from the true initial state; its first sync point at/after kc
      walk_any<TRANS, IS_MATRIX, MODE_START>(tr, A.sym, off, len, rec, kc, sink, epsilon, unknown,
                                              identity, step_cap(A.step_factor, len), rec, st, steps, win_row, s_lut);
    }
  }
  return rec;
}

template <typename TRANS, bool IS_MATRIX>
__global__ __launch_bounds__(WAVE) void k_spec_start(TRANS tr, DtkWalkArgs A, DtkSpecArgs S,
                                                     uint32_t epsilon, uint32_t unknown,
                                                     uint32_t identity) {
  DTK_WINDOWS(TRANS, A.sym)
  const uint32_t L = blockIdx.x * WAVE + threadIdx.x;
  uint32_t steps = 0;
  if (L < S.n_lanes) {
    const uint32_t d = S.lane_doc[L];
    const uint32_t k = L - S.chunk_off[d];
    const uint64_t off = A.doc_off[d];
    const uint32_t len = (uint32_t)(A.doc_off[d + 1] - off);
    S.lane_start[L] = start_record<TRANS, IS_MATRIX>(tr, A, S, k, off, len, epsilon, unknown, identity, win_row, s_lut, steps);
  }
  add_steps(A.steps, steps);
}

// First pass in one launch: the lane finds its start record (as k_spec_start) and walks on from it
// to its first sync point at or behind the end of its chunk -- which is the record its successor
// finds for itself if the speculation holds, and k_spec_verify checks exactly that.  Every lane
// with a record walks; what lanes behind a broken chain stored is cleared by the repair round.
extern __shared__ uint32_t s_dyn_bits[];  // the wave's event bitmaps (3 kinds x S.lds_words), if any

#ifndef DTK_WALK_OCC
#define DTK_WALK_OCC 7  // the lean loop in 72 VGPRs (the allocator finds them without another spill): 7 waves per SIMD where LDS allows (three batches in flight: walk -5 %)
#endif
template <typename TRANS, bool IS_MATRIX>
__global__ __launch_bounds__(WAVE, TRANS::LEAN ? DTK_WALK_OCC : 1) void k_spec_both(TRANS tr, DtkWalkArgs A, DtkSpecArgs S,
                                                    uint32_t epsilon, uint32_t unknown,
                                                    uint32_t identity) {
  DTK_WINDOWS(TRANS, A.sym)
  uint32_t *lds_bits = S.lds_words ? s_dyn_bits : nullptr;
#ifdef DTK_PROBE
  const unsigned long long pt0 = clock64();
  unsigned long long pt1 = pt0, pt2 = pt0;
  s_probe_mark = pt0;
#endif
  const uint32_t L = blockIdx.x * WAVE + threadIdx.x;
  const uint32_t w0 = S.lds_words ? lds_bits_word0(A, S, blockIdx.x * WAVE) : 0u;
  if (lds_bits) lds_bits_clear(lds_bits, S.lds_words);
  uint32_t steps = 0;
  if (L < S.n_lanes) {
    const uint32_t d = S.lane_doc[L];
    const uint32_t k = L - S.chunk_off[d];
    const uint64_t off = A.doc_off[d];
    const uint32_t len = (uint32_t)(A.doc_off[d + 1] - off);
    const DtkLaneState rec =
        start_record<TRANS, IS_MATRIX>(tr, A, S, k, off, len, epsilon, unknown, identity, win_row, s_lut, steps);
#ifdef DTK_PROBE
    pt1 = clock64();
#endif
    S.lane_start[L] = rec;
    DtkLaneState fin{0xFFFFFFFFu, 0u, 0u, LANE_F_IDLE};
    DtkLaneCount cnt{0u, 0u, 0u, 0u, 0u, 0xFFFFFFFFu, 0u, 0u};
    const uint32_t stop = (L + 1u < S.chunk_off[d + 1]) ? (k + 1u) * S.chunk : 0xFFFFFFFFu;
    if (rec.p != 0xFFFFFFFFu && rec.p >= stop) {
      // my first sync point lies behind my whole chunk (a token longer than a chunk): it is my
      // successor's record too, and I own nothing
      fin = rec;
      fin.flags &= (LANE_F_SENT | LANE_F_TEXT | LANE_F_OK);
    } else if (rec.p != 0xFFFFFFFFu) {
      EventSink sink;
      sink.init(A, off, d, rec.p, 0xFFFFFFFFu, lds_bits, S.lds_words, w0);
      uint32_t st = 0, steps2 = 0;
      walk_any<TRANS, IS_MATRIX, MODE_CHUNK, true>(tr, A.sym, off, len, rec, stop, sink, epsilon, unknown,
                                                    identity, step_cap(A.step_factor, len), fin, st, steps2, win_row, s_lut);
      steps += steps2;
      if (sink.dropped) fin.flags |= LANE_F_DROPPED;
      cnt.tok = sink.c_tok; cnt.sent = sink.c_sent; cnt.text = sink.c_text; cnt.status = st | sink.st;
      cnt.sev = sink.c_sev; cnt.e_pos = sink.e_pos; cnt.e_tok = sink.e_tok;
    }
#ifdef DTK_PROBE
    pt2 = clock64();
#endif
    S.lane_end[L] = fin;
    S.lane_cnt[L] = cnt;
  }
  if (lds_bits) lds_bits_flush(lds_bits, S.lds_words, A.bits, A.bit_words, w0);
  add_steps(A.steps, steps);
#ifdef DTK_PROBE
  {
    __syncthreads();
    const unsigned long long pt3 = clock64(), mark = s_probe_mark;
    if (threadIdx.x == 0) {
      atomicAdd(&g_phase[0], mark - pt0); atomicAdd(&g_phase[1], pt1 - mark); atomicAdd(&g_phase[2], pt2 - pt1);
      atomicAdd(&g_phase[3], pt3 - pt2); atomicAdd(&g_phase[4], 1ull);
    }
  }
#endif
}

// Every lane derives its window from the start records and first_bad[d] (k_spec_link), in the
// first pass and in repair rounds alike.
__device__ __forceinline__ DtkLanePlan plan_of(const DtkSpecArgs &S, uint32_t L, uint32_t d) {
  const uint32_t L0 = S.chunk_off[d], L1 = S.chunk_off[d + 1];
  const uint32_t k = L - L0, fb = ~S.first_bad[d];
  DtkLanePlan pl;
  pl.pad = 0;
  if (k < fb) {
    pl.stop = pl.wend = S.lane_start[L + 1].p;
    pl.mode = PLAN_CHAINED;
  } else if (k == fb) {
    // last enabled lane: stops at the first sync point behind its own chunk (or EOF)
    pl.stop = (L + 1 < L1) ? (k + 1u) * S.chunk : 0xFFFFFFFFu;
    pl.wend = 0xFFFFFFFFu;
    pl.mode = PLAN_LAST;
  } else {
    pl.stop = pl.wend = 0;
    pl.mode = PLAN_OFF;
  }
  return pl;
}

template <typename TRANS, bool IS_MATRIX>
__global__ __launch_bounds__(WAVE) void k_spec_walk(TRANS tr, DtkWalkArgs A, DtkSpecArgs S,
                                                    uint32_t epsilon, uint32_t unknown,
                                                    uint32_t identity) {
  DTK_WINDOWS(TRANS, A.sym)
  if (S.go && *S.go == 0u) return;
  uint32_t *lds_bits = S.lds_words ? s_dyn_bits : nullptr;
  const uint32_t L = blockIdx.x * WAVE + threadIdx.x;
  const uint32_t w0 = S.lds_words ? lds_bits_word0(A, S, blockIdx.x * WAVE) : 0u;
  if (lds_bits) lds_bits_clear(lds_bits, S.lds_words);
  uint32_t steps = 0;
  if (L < S.n_lanes) {
    const uint32_t d = S.lane_doc[L];
    const bool redo = S.redo_from != nullptr;
    if (!redo || (S.redo_from[d] != 0xFFFFFFFFu && L >= S.redo_from[d])) {
      const DtkLanePlan pl = plan_of(S, L, d);
      DtkLaneState fin{0xFFFFFFFFu, 0u, 0u, LANE_F_IDLE};
      DtkLaneCount cnt{0u, 0u, 0u, 0u, 0u, 0xFFFFFFFFu, 0u, 0u};
      const DtkLaneState init0 = pl.mode == PLAN_CHAINED ? S.lane_start[L] : DtkLaneState{0xFFFFFFFFu, 0u, 0u, 0u};
      if (pl.mode == PLAN_CHAINED && init0.p != 0xFFFFFFFFu && init0.p >= pl.stop) {
        // my record is my successor's too (a token longer than a chunk): nothing of it is mine
        fin = init0;
        fin.flags &= (LANE_F_SENT | LANE_F_TEXT | LANE_F_OK);
      } else if (pl.mode != PLAN_OFF) {
        const uint64_t off = A.doc_off[d];
        const uint32_t len = (uint32_t)(A.doc_off[d + 1] - off);
        const DtkLaneState init = S.lane_start[L];
        EventSink sink;
        sink.init(A, off, d, init.p, pl.wend, lds_bits, S.lds_words, w0);
        uint32_t st = 0;
        walk_any<TRANS, IS_MATRIX, MODE_CHUNK>(tr, A.sym, off, len, init, pl.stop, sink, epsilon, unknown,
                                                identity, step_cap(A.step_factor, len), fin, st, steps, win_row, s_lut);
        if (sink.dropped) fin.flags |= LANE_F_DROPPED;
        cnt.tok = sink.c_tok; cnt.sent = sink.c_sent; cnt.text = sink.c_text; cnt.status = st | sink.st;
        cnt.sev = sink.c_sev; cnt.e_pos = sink.e_pos; cnt.e_tok = sink.e_tok;
      }
      S.lane_end[L] = fin;
      S.lane_cnt[L] = cnt;
    }
  }
  if (lds_bits) lds_bits_flush(lds_bits, S.lds_words, A.bits, A.bit_words, w0);
  add_steps(A.steps, steps);
}

// ---------------------------------------------------------------- launchers

extern "C" int dtk_launch_walk(const DtkTableDev *tab, const DtkWalkArgs *args, void *stream) {
  if (args->n_docs == 0) return 0;
  const uint32_t blocks = (args->n_docs + WAVE - 1) / WAVE;
  hipStream_t s = (hipStream_t)stream;
  return with_trans(tab, args->sym.lut != nullptr, [&](auto tr, auto is_matrix) {
    using TR = decltype(tr);
    hipLaunchKernelGGL((k_walk_doc<TR, decltype(is_matrix)::value>), dim3(blocks), dim3(WAVE), 0, s, tr, *args,
                       tab->epsilon, tab->unknown, tab->identity);
  });
}

// stage: 6 start records + walk, 7 link + verify, 4 fix (first pass; or 0 start records, 1 link, 2 walk, 3 verify, 4);
//        5 clear, 6 plan, 2 walk, 7 check (repair rounds, spec->redo_from set)
extern "C" int dtk_launch_spec(const DtkTableDev *tab, const DtkWalkArgs *args, const DtkSpecArgs *spec,
                               int stage, uint32_t cmp_mask, uint32_t *redo_out, uint32_t *n_bad,
                               void *stream) {
  if (args->n_docs == 0 || spec->n_lanes == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const uint32_t lane_blocks = (spec->n_lanes + WAVE - 1) / WAVE;
  switch (stage) {
    case 0:
      return with_trans(tab, args->sym.lut != nullptr, [&](auto tr, auto is_matrix) {
        using TR = decltype(tr);
        hipLaunchKernelGGL((k_spec_start<TR, decltype(is_matrix)::value>), dim3(lane_blocks), dim3(WAVE), 0, s,
                           tr, *args, *spec, tab->epsilon, tab->unknown, tab->identity);
      });
    case 6:  // first pass: start records and chunk walk in one launch
      return with_trans(tab, args->sym.lut != nullptr, [&](auto tr, auto is_matrix) {
        using TR = decltype(tr);
        hipLaunchKernelGGL((k_spec_both<TR, decltype(is_matrix)::value>), dim3(lane_blocks), dim3(WAVE),
                           3u * spec->lds_words * sizeof(uint32_t), s, tr, *args, *spec, tab->epsilon, tab->unknown,
                           tab->identity);
      });
    case 2:
      return with_trans(tab, args->sym.lut != nullptr, [&](auto tr, auto is_matrix) {
        using TR = decltype(tr);
        hipLaunchKernelGGL((k_spec_walk<TR, decltype(is_matrix)::value>), dim3(lane_blocks), dim3(WAVE),
                           3u * spec->lds_words * sizeof(uint32_t), s, tr, *args, *spec, tab->epsilon, tab->unknown,
                           tab->identity);
      });
    case 1: case 3: case 7: case 4: case 5:  // link, verify, link + verify, fix, spread + reset: dtk_repair.hip
      return dtk_launch_spec_check(args, spec, stage, cmp_mask, redo_out, n_bad, stream);
  }
  return -1;
}

// the exact pass over the listed documents (one lane each; the general loop for every table kind)
extern "C" int dtk_launch_exact(const DtkTableDev *tab, const DtkExactArgs *args, void *stream) {
  if (args->n == 0) return 0;
  const uint32_t blocks = (args->n + WAVE - 1) / WAVE;
  hipStream_t s = (hipStream_t)stream;
  return with_trans(tab, args->sym.lut != nullptr, [&](auto tr, auto is_matrix) {
    using TR = decltype(tr);
    if constexpr (TR::LEAN) {
      const MatrixFusedTrans base = tr;
      hipLaunchKernelGGL((k_exact_doc<MatrixFusedTrans, decltype(is_matrix)::value>), dim3(blocks), dim3(WAVE), 0, s, base, *args,
                         tab->epsilon, tab->unknown, tab->identity);
    } else {
      hipLaunchKernelGGL((k_exact_doc<TR, decltype(is_matrix)::value>), dim3(blocks), dim3(WAVE), 0, s, tr, *args,
                         tab->epsilon, tab->unknown, tab->identity);
    }
  });
}

