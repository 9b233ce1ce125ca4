// dtk_walk_core.h -- the FSA transition walk itself (matrix.go:348-698 / datok.go:781-1135) as device templates:
// transition policies per table encoding, the event sink, the general loop (walk_lane) and the lean loop
// (walk_fused).  Included by dtk_walk.hip, whose kernels decide what a lane is asked to do.
#pragma once
#include "dtk_device.h"

// --------------------------------------------------------------------- walk
//
// Transition policies.  A "state" is (t, aux): aux is unused for the matrix and
// holds the device base word of t for the double array.

template <typename CELL>
struct MatrixTrans {
  const CELL *tab;
  uint32_t stride, n_eps, start;
  static constexpr CELL FLAG = (CELL)((CELL)1 << (sizeof(CELL) * 8 - 1));
  static constexpr bool FUSED = false;
  static constexpr bool LEAN = false;
  __device__ __forceinline__ uint32_t start_state() const { return start; }
  __device__ __forceinline__ uint32_t start_aux() const { return 0; }
  // matrix.go:442 `array[(epsilon-1)*stateCount+t0] != 0` after renumbering
  __device__ __forceinline__ bool has_eps(uint32_t t, uint32_t) const { return t <= n_eps; }
  // matrix.go:459-464; column 0 is zero so a == 0 fails without a branch
  __device__ __forceinline__ bool step(uint32_t t0, uint32_t, uint32_t a, uint32_t &t,
                                       uint32_t &aux, bool &nontoken, uint32_t &st, uint32_t &via) const {
    via = 0;
    const CELL x = tab[(size_t)t0 * stride + a];
    t = (uint32_t)(x & (CELL)~FLAG);  // matrix.go:629 t &= ^FIRSTBIT
    nontoken = (x & FLAG) != 0;       // matrix.go:584
    aux = 0;
    (void)st;
    return t != 0;  // matrix.go:472
  }
};

// Matrix with fused cells (uint32): where state t has an epsilon arc to e and no arc on
// symbol a, but e has one, the cell (t, a) holds  1<<31 | e<<16 | cell(e, a).  The walk then
// does in one lookup what the reference does in three (matrix.go:472-497 fail + backtrack to
// the state remembered at this very rune, :563-576 epsilon step, :579-591 the rune from e).
// Only built when state ids fit 15 bits.
struct MatrixFusedTrans {
  const uint32_t *tab;
  uint32_t stride, n_eps, start;
  uint32_t ident_guard;  // the identity symbol if the model has arcs on `unknown`, else no symbol
  static constexpr bool FUSED = true;
  static constexpr bool LEAN = false;
  __device__ __forceinline__ uint32_t start_state() const { return start; }
  __device__ __forceinline__ uint32_t start_aux() const { return 0; }
  __device__ __forceinline__ bool has_eps(uint32_t t, uint32_t) const { return t <= n_eps; }
  // via: 0, or the epsilon target e the fused cell goes through
  __device__ __forceinline__ bool step(uint32_t t0, uint32_t, uint32_t a, uint32_t &t, uint32_t &aux,
                                       bool &nontoken, uint32_t &st, uint32_t &via) const {
    const uint32_t x = tab[(size_t)t0 * stride + a];
    t = x & 0x7FFFu;
    nontoken = (x & 0x8000u) != 0;
    via = x >> 31 ? (x >> 16) & 0x7FFFu : 0u;
    aux = 0;
    (void)st;
    return t != 0;
  }
};

// The same table walked by the lean loop (walk_fused): chosen by the launcher when no state has an
// arc on `unknown` (a separate type so that the kernels only carry one loop: fewer registers).
struct MatrixLeanTrans : MatrixFusedTrans {
  static constexpr bool LEAN = true;
};

struct DaTrans {
  const uint2 *arr;  // .x base (bit31 separate, bit30 has-epsilon cache), .y check
  uint32_t len, size, base1;
  static constexpr bool FUSED = false;
  static constexpr bool LEAN = false;
  __device__ __forceinline__ uint32_t start_state() const { return 1u; }  // datok.go:784
  __device__ __forceinline__ uint32_t start_aux() const { return base1; }
  // datok.go:876, precomputed per index at load
  __device__ __forceinline__ bool has_eps(uint32_t, uint32_t aux) const {
    return (aux & DTK_SECONDBIT) != 0;
  }
  // datok.go:889-901 and :1056-1058
  __device__ __forceinline__ bool step(uint32_t t0, uint32_t aux0, uint32_t a, uint32_t &t,
                                       uint32_t &aux, bool &nontoken, uint32_t &st, uint32_t &via) const {
    via = 0;
    const uint32_t idx = (aux0 & DTK_RESTBIT) + a;
    if (idx >= len) { st |= ST_BAD_MODEL; return false; }  // Go: index panic
    const uint2 ta = arr[idx];
    if (idx > size || (ta.y & DTK_RESTBIT) != t0) return false;
    nontoken = (ta.y & DTK_FIRSTBIT) != 0;  // datok.go:994 isNonToken
    if (ta.x & DTK_FIRSTBIT) {              // isSeparate: move to the representative
      t = ta.x & DTK_RESTBIT;
      if (t >= len) { st |= ST_BAD_MODEL; return false; }
      aux = arr[t].x;
    } else {
      t = idx;
      aux = ta.x;
    }
    return true;
  }
};

// What the walk reports: one bit per event in the bitmap of its kind (dtk_internal.h).  Token ends, token starts
// and epsilon SentenceEnds -- three bits per token and a bit -- are OR-ed into the wave's bitmaps in LDS, which the
// wave writes out as whole words when its lanes are done; the rare kinds (EOT calls) and positions outside the
// wave's range (a lane's last token may end far behind its chunk) go straight to memory.  A lane only reports inside
// its window
//   opening kinds (START, SEPS): lo <= pos < hi      closing kinds (END, TEOT, SEOT): lo < pos <= hi
// (whole document: lo = 0, hi = 0xFFFFFFFF); an event outside is dropped and remembered, the check pass then
// knows the lane left its window.
// Calls the bitmaps cannot order flag the document ST_IRREGULAR for the exact pass: a second epsilon SentenceEnd
// at one cursor, a Token call with an empty surface or a negative one (two starts or two ends on one bit), an
// EOT fired twice at one position (double array, datok.go:1019-1030 keeps its window).
// (an LDS pointer that stays one: through a plain pointer the compiler loses the address space and emits FLAT atomics)
typedef __attribute__((address_space(3))) uint32_t dtk_lds_u32;

struct EventSink {
  uint32_t *g;         // the batch's bitmaps
  uint32_t gw;         // words per kind
  uint32_t gb;         // bit of position 0 of the document
  uint32_t gbr;        // the same, counted from LDS word 0
  dtk_lds_u32 *lds;    // the wave's bitmaps in LDS (END, START, SEPS); lw == 0: none
  uint32_t lw;         // words per kind there
  uint32_t w0;         // global word of LDS word 0
  uint32_t *tailw;     // the document's tail word
  uint32_t lo, hi;     // window
  uint32_t last_s_p1;  // position of the last epsilon SentenceEnd, plus one (0: none)
  uint32_t last_eot_p; // position of the last EOT pair
  uint32_t st;
  uint32_t dropped;
  // what NewTokenWriter would have collected from this lane's calls
  // (token_writer.go:72-81, 104-109, 131-159): tokens, ints of the sentence list, texts
  uint32_t c_tok, c_sent, c_text;
  uint32_t c_sev;         // SentenceEnd calls (all of them, also where the reference would panic)
  uint32_t e_pos, e_tok;  // the last EOT TextEnd of this lane: position, Token calls before it
  __device__ __forceinline__ void init(const DtkWalkArgs &A, uint64_t off, uint32_t d, uint32_t wlo, uint32_t whi,
                                       uint32_t *lds_bits = nullptr, uint32_t lds_words = 0, uint32_t word0 = 0) {
    g = A.bits; gw = A.bit_words; gb = (uint32_t)DTK_EV_BIT(off, d); tailw = A.doc_tail ? A.doc_tail + d : nullptr;
    lds = (dtk_lds_u32 *)lds_bits; lw = lds_bits ? lds_words : 0u; w0 = word0; gbr = gb - (word0 << 5);
    lo = wlo; hi = whi;
    last_s_p1 = 0u; last_eot_p = 0xFFFFFFFFu; st = 0; dropped = 0;
    c_tok = c_sent = c_text = 0;
    c_sev = 0; e_pos = 0xFFFFFFFFu; e_tok = 0;
  }
  __device__ __forceinline__ void put(uint32_t kind, uint32_t pos) {
    if (DTK_KO & 8) return;
    const uint32_t G = gb + pos, m = 1u << (G & 31u), w = (G >> 5) - w0;
    // (one wave-uniform test keeps the common case free of exec-mask juggling: all lanes inside the LDS range)
    if (__builtin_amdgcn_ballot_w64(w >= lw) == 0ull) {
      __hip_atomic_fetch_or(&lds[kind * lw + w], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // ds_or_b32
    } else if (w < lw) {
      __hip_atomic_fetch_or(&lds[kind * lw + w], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      atomicOr(&g[kind * gw + (G >> 5)], m);
    }
  }
  __device__ __forceinline__ bool in_closing(uint32_t p) const { return p > lo && p <= hi; }
  __device__ __forceinline__ bool in_opening(uint32_t p) const { return p >= lo && p < hi; }

  // Token(bufft, buffer[:buffc]) -- matrix.go:528,569,675
  // sent_first: no token since the last SentenceEnd / TextEnd (the writer's sentB)
  template <bool IS_MATRIX>
  __device__ __forceinline__ void token(uint32_t /*bs*/, uint32_t tp, uint32_t p, bool sent_first) {
    if (!in_closing(p)) { dropped = 1; return; }
    c_tok++;
    c_sent += sent_first ? 1u : 0u;
    if (p <= tp) st |= ST_IRREGULAR;
    // The double array keeps its window over an EOT: a token may be flushed AFTER the EOT's SentenceEnd / TextEnd and
    // end BEFORE it (the walk backtracked behind the EOT; if the retry then reads the EOT as an ordinary rune,
    // matrix.go:555 / datok.go, nothing fires twice).  Calls out of position order: the exact pass.
    if (!IS_MATRIX && last_eot_p != 0xFFFFFFFFu && p < last_eot_p) st |= ST_IRREGULAR;
    put(EVB_END, p);
    put(EVB_START, tp);
  }
  // SentenceEnd? + TextEnd fired by an EOT rune -- matrix.go:593-600
  // has_tok: the current text has a token (else the reference panics in position modes)
  template <bool IS_MATRIX>
  __device__ __forceinline__ void eot(uint32_t /*bs*/, uint32_t p, bool with_sentence, bool has_tok) {
    // (The double array has no upper bound here: it keeps its window over an EOT, so the EOT is no sync point and a
    //  lane may fire one behind its stop position and then backtrack to a token end in front of it -- the construct
    //  of the exact pass.  The fire is the lane's: counted, its bit set; the successor's second fire finds the bit
    //  and flags the document.  Dropped, it made the lane fail its check in every repair round.)
    if (IS_MATRIX ? !in_closing(p) : p <= lo) { dropped = 1; return; }
    c_text++;
    c_sev += with_sentence ? 1u : 0u;
    e_pos = p; e_tok = c_tok;
    if (has_tok) c_sent += with_sentence ? 1u : 0u; else st |= ST_EMPTY_TEXT;
    last_eot_p = p;
    const uint32_t G = gb + p, m = 1u << (G & 31u);
    if (atomicOr(&g[EVB_TEOT * gw + (G >> 5)], m) & m) st |= ST_IRREGULAR;  // the same EOT fired before (by any lane)
    if (with_sentence) atomicOr(&g[EVB_SEOT * gw + (G >> 5)], m);
  }
  // SentenceEnd from an epsilon arc on an empty token -- matrix.go:574-575
  template <bool IS_MATRIX>
  __device__ __forceinline__ void sentence(uint32_t /*bs*/, uint32_t p, bool has_tok) {
    if (!in_opening(p)) { dropped = 1; return; }
    c_sev++;
    if (has_tok) c_sent++; else st |= ST_EMPTY_TEXT;
    if (p < last_s_p1) st |= ST_IRREGULAR;  // twice at one position, or behind a backtrack: not in position order
    last_s_p1 = p + 1u;
    put(EVB_SEPS, p);
  }
  // First pass (k_spec_both), events before the lane's stop position: the window is open-ended and the position lies
  // inside the wave's LDS bitmaps (their 64 chunks plus a bit per document boundary; DTK_LDS_BIT_WORDS) -- no window
  // test, no range test.  What a lane reports at or behind its stop position goes through the calls above.
  __device__ __forceinline__ void put_first(uint32_t kind, uint32_t pos) {
    if (DTK_KO & 8) return;
    // (k_spec_both only runs with LDS bitmaps: without them dtk_batch_run launches start records and walk apart;
    //  gbr = the document's bit base relative to the wave's first LDS word)
    const uint32_t G = gbr + pos;
    __hip_atomic_fetch_or(&lds[kind * lw + (G >> 5)], 1u << (G & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  template <bool IS_MATRIX>
  __device__ __forceinline__ void token_first(uint32_t tp, uint32_t p, bool sent_first) {
    c_tok++;
    c_sent += sent_first ? 1u : 0u;
    if (!IS_MATRIX && last_eot_p != 0xFFFFFFFFu && p < last_eot_p) st |= ST_IRREGULAR;  // (see token())
    put_first(EVB_END, p);
    put_first(EVB_START, tp);
  }
  __device__ __forceinline__ void sentence_first(uint32_t p, bool has_tok) {
    c_sev++;
    if (has_tok) c_sent++; else st |= ST_EMPTY_TEXT;
    if (p < last_s_p1) st |= ST_IRREGULAR;  // twice at one position, or behind a backtrack: not in position order
    last_s_p1 = p + 1u;
    put_first(EVB_SEPS, p);
  }
  // final SentenceEnd / TextEnd -- matrix.go:683-691
  template <bool IS_MATRIX>
  __device__ __forceinline__ void tail_(uint32_t p, bool sentence_end, bool text_end, bool has_tok) {
    const uint32_t bits = (sentence_end ? 0u : DTK_TAIL_S) | (text_end ? 0u : DTK_TAIL_E);
    if (!bits) return;
    if (!in_opening(p)) { dropped = 1; return; }
    if (!text_end) c_text++;
    c_sev += sentence_end ? 0u : 1u;
    if (has_tok) c_sent += sentence_end ? 0u : 1u; else st |= ST_EMPTY_TEXT;
    if (tailw && !(DTK_KO & 8)) *tailw = (p << 2) | bits;
  }
  template <bool IS_MATRIX>
  __device__ __forceinline__ void tail(uint32_t /*bs*/, uint32_t p, bool sentence_end, bool text_end, bool has_tok) {
    tail_<IS_MATRIX>(p, sentence_end, text_end, has_tok);
  }
  // calls that the position-indexed bitmaps cannot order
  __device__ __forceinline__ void out_of_order() { st |= ST_IRREGULAR; }
  __device__ __forceinline__ void flush() {}
};

// the wave's LDS bitmaps: cleared before the lanes walk, OR-ed into memory afterwards (all 64 lanes take part;
// one wave per block, LDS operations of a wave complete in order)
__device__ __forceinline__ void lds_bits_clear(uint32_t *lds_, uint32_t lw) {
  dtk_lds_u32 *lds = (dtk_lds_u32 *)lds_;
  for (uint32_t j = threadIdx.x; j < 3u * lw; j += WAVE) lds[j] = 0u;
  __syncthreads();
}
__device__ __forceinline__ void lds_bits_flush(const uint32_t *lds_, uint32_t lw, uint32_t *g, uint32_t gw, uint32_t w0) {
  const dtk_lds_u32 *lds = (const dtk_lds_u32 *)lds_;
  __syncthreads();
  for (uint32_t j = threadIdx.x; j < lw; j += WAVE) {
    if (w0 + j >= gw) break;
#pragma unroll
    for (uint32_t k = 0; k < 3u; k++) {
      const uint32_t v = lds[k * lw + j];
      if (v) atomicOr(&g[k * gw + w0 + j], v);
    }
  }
}
// global word of LDS word 0 for the wave whose first lane is L0
__device__ __forceinline__ uint32_t lds_bits_word0(const DtkWalkArgs &A, const DtkSpecArgs &S, uint32_t L0) {
  const uint32_t d0 = S.lane_doc[L0];
  return (uint32_t)((DTK_EV_BIT(A.doc_off[d0], d0) + (uint64_t)(L0 - S.chunk_off[d0]) * S.chunk) >> 5);
}

template <typename TRANS>
__device__ __forceinline__ uint32_t guard_of(const TRANS &tr) {
  if constexpr (TRANS::FUSED) return tr.ident_guard; else return 0xFFFFFFFFu;
}

// What a lane is asked to do.
enum { MODE_DOC = 0,    // whole document from the initial state, all events
       MODE_START = 1,  // speculative warm-up: no events, stop at the first rewind at/after stop_pos
       MODE_CHUNK = 2   // walk from a recorded start, events inside the window, stop at the first
                        // rewind at/after stop_pos (or run the EOF tail)
};

// Runes in [from, to): only needed when a window may have outgrown the
// reference's 1024-rune buffer (matrix.go:365), i.e. when it spans > 1024 bytes.
struct DtkSymAt { DtkSym S; uint64_t off; };  // a document's stretch of the stream
__device__ __noinline__ uint32_t count_runes(const DtkSymAt &s, uint32_t from, uint32_t to) {
  uint32_t n = 0;
  for (uint32_t i = from; i < to; i++) n += dtk_sym_is_start(s.S, s.off + i) ? 1u : 0u;
  return n;
}

// window of the symbol stream in LDS (see walk_fused)
typedef uint2 __attribute__((may_alias)) dtk_u2a;
typedef uint16_t __attribute__((may_alias)) dtk_u16a;
typedef uint8_t __attribute__((may_alias)) dtk_u8a;
// (the general loop keeps entries in its rows: from a stream of codes they are translated on the way in -- 32 table
//  reads per refill, cached; the loop itself does not know the difference.  `at` = stream index of the row's entry 0,
//  a multiple of 8)
__device__ __forceinline__ void win_fill(dtk_u16a *row, const DtkSym &S, uint64_t at) {
  if (S.lut) {
    const uint2 *__restrict__ g = reinterpret_cast<const uint2 *>(static_cast<const uint8_t *>(S.base) + at);
    uint2 c[DTK_WIN / 8u];
#pragma unroll
    for (uint32_t i = 0; i < DTK_WIN / 8u; i++) c[i] = g[i];
    dtk_u2a *r = reinterpret_cast<dtk_u2a *>(row);
#pragma unroll
    for (uint32_t i = 0; i < DTK_WIN / 8u; i++) {
      const uint32_t lo = c[i].x, hi = c[i].y;
      r[2 * i] = make_uint2((uint32_t)S.lut[lo & 255u] | ((uint32_t)S.lut[(lo >> 8) & 255u] << 16),
                            (uint32_t)S.lut[(lo >> 16) & 255u] | ((uint32_t)S.lut[lo >> 24] << 16));
      r[2 * i + 1] = make_uint2((uint32_t)S.lut[hi & 255u] | ((uint32_t)S.lut[(hi >> 8) & 255u] << 16),
                                (uint32_t)S.lut[(hi >> 16) & 255u] | ((uint32_t)S.lut[hi >> 24] << 16));
    }
    return;
  }
  const uint4 *__restrict__ g = reinterpret_cast<const uint4 *>(static_cast<const uint16_t *>(S.base) + at);
  uint4 v[DTK_WIN / 8u];
#pragma unroll
  for (uint32_t i = 0; i < DTK_WIN / 8u; i++) v[i] = g[i];  // all loads before the first LDS write
  dtk_u2a *r = reinterpret_cast<dtk_u2a *>(row);
#pragma unroll
  for (uint32_t i = 0; i < DTK_WIN / 8u; i++) {
    r[2 * i] = make_uint2(v[i].x, v[i].y);
    r[2 * i + 1] = make_uint2(v[i].z, v[i].w);
  }
}

// the lean loop's row: DTK_WIN8 codes
__device__ __forceinline__ void win_fill8(dtk_u8a *row, const uint8_t *__restrict__ from) {
  const uint4 *__restrict__ g = reinterpret_cast<const uint4 *>(from);
  uint4 v[DTK_WIN8 / 16u];
#pragma unroll
  for (uint32_t i = 0; i < DTK_WIN8 / 16u; i++) v[i] = g[i];  // all loads before the first LDS write
  dtk_u2a *r = reinterpret_cast<dtk_u2a *>(row);
#pragma unroll
  for (uint32_t i = 0; i < DTK_WIN8 / 16u; i++) {
    r[2 * i] = make_uint2(v[i].x, v[i].y);
    r[2 * i + 1] = make_uint2(v[i].z, v[i].w);
  }
}

// The walk of matrix.go:348-698 / datok.go:781-1135 for one lane.
// Returns through `fin`: p == 0xFFFFFFFF means "ran to EOF" (MODE_START: no
// rewind found; otherwise: tail done).
//
// The reference's rune window is not materialised: p / tp / bs / hi are byte
// positions of buffer[buffc] / buffer[bufft] / buffer[0] / buffer[buffi], and the
// symbol stream (read through the lane's window in LDS) replaces the rune -> symbol lookups
// of matrix.go:421-435.
template <typename TRANS, bool IS_MATRIX, int MODE, typename SINK = EventSink>
__device__ __forceinline__ void walk_lane(const TRANS &tr, const DtkSym &sym,
                                          uint64_t off, uint32_t len, DtkLaneState init, uint32_t stop_pos,
                                          SINK &sink, uint32_t epsilon, uint32_t unknown,
                                          uint32_t identity, uint32_t cap, DtkLaneState &fin,
                                          uint32_t &st_out, uint32_t &steps_out, uint16_t *win_row) {
  // the lane's window of the symbol stream in LDS: entries (pos + o7) in [wbase, wbase + DTK_WIN)
  dtk_u16a *row = reinterpret_cast<dtk_u16a *>(win_row);
  const uint32_t o7 = (uint32_t)(off & 7u);
  const uint64_t aligned = off - o7;
  const DtkSymAt s{sym, off};

  uint32_t a = 0, t0 = 0, aux0 = 0;
  uint32_t t = init.t, aux = init.aux;  // matrix.go:351 `t := uint32(1)`
  const uint32_t t_start = tr.start_state(), aux_start = tr.start_aux();
  bool ok = (init.flags & LANE_F_OK) != 0;  // sticky `ok` of matrix.go:352 / datok.go:785
  uint32_t eps_t = 0, eps_aux = 0, eps_p = 0;  // epsilonState / epsilonOffset
  bool sentence_end = (init.flags & LANE_F_SENT) != 0, text_end = (init.flags & LANE_F_TEXT) != 0;
  uint32_t p = init.p;   // buffer[buffc]
  uint32_t tp = init.p;  // buffer[bufft]
  uint32_t bs = init.p;  // buffer[0]: position of the last rewind
  uint32_t hi = init.p;  // behind buffer[buffi-1]: read high-water mark
  uint32_t w = 1;        // width of the rune at p
  bool eot = false, newchar = true;
  uint32_t st = 0, my_steps = 0;
  fin.p = 0xFFFFFFFFu; fin.t = 0; fin.aux = 0; fin.flags = 0;
  bool stopped = false;
  // NewTokenWriter state that the counts need.  A lane starts right after a rewind:
  // p > 0 means a token was flushed or an EOT fired there, so "a token exists in the
  // document" is p > 0 and "in the current text" additionally needs !textEnd.
  bool any_tok = init.p > 0;               // some Token call happened (else sentB is still true)
  bool has_tok = init.p > 0 && !text_end;  // pos[] of the current text is not empty

  uint32_t wbase = (init.p + o7) & ~7u;
  win_fill(row, sym, aligned + wbase);

  // One table lookup per iteration (the reference's loop body, matrix.go:384-635),
  // written as predicates + selects so that the 64 lanes of a wave, which are all
  // in different phases of their tokens, share one short instruction stream.  Real
  // branches are kept for the symbol fetch, the two event stores and three rare
  // paths (EOF drain; hard fail; EOT / window limit / end of chunk).
  bool done = false;
  do {
    if (newchar && p >= len) {
      // reader at EOF: the drain of matrix.go:650-668 / datok.go:1085-1103
      const bool he = tr.has_eps(t, aux);          // goto PARSECHARM with a = epsilon
      const bool bt = !he && eps_t != 0;           // or pop the remembered epsilon state
      t0 = bt ? eps_t : t; aux0 = bt ? eps_aux : aux;
      p = bt ? eps_p : p;
      eps_t = bt ? 0u : eps_t;
      a = epsilon;
      newchar = false;
      done = !he && !bt;
    }
    // a lane that needs a rune outside its window: all lanes of the wave re-base theirs
    if (__builtin_amdgcn_ballot_w64(newchar && (p + o7 - wbase) >= DTK_WIN) != 0ull) {
      wbase = (p + o7) & ~7u;
      win_fill(row, sym, aligned + wbase);
    }
    if (newchar) {
      const uint32_t e = row[p + o7 - wbase];
      a = e & DTK_SYM_MASK;
      w = DTK_SYM_WIDTH(e);
      const uint32_t cls = (e >> DTK_SYM_CLS_SHIFT) & 3u;
      hi = max(hi, p + w);             // matrix.go:388-408: runes enter the window once
      eot = cls == 1u;                 // matrix.go:422
      ok = cls >= 2u ? cls == 2u : ok; // matrix.go:427: only runes >= 256 write `ok`
      t0 = t; aux0 = aux;              // matrix.go:437
      const bool he = tr.has_eps(t0, aux0);  // matrix.go:442-454
      eps_t = he ? t0 : eps_t; eps_aux = he ? aux0 : eps_aux;
      eps_p = he ? p : eps_p;
    }

    bool nontoken = false;
    uint32_t via = 0;
    const bool fresh = newchar;  // this lookup is the first one for the rune at p
    bool good = tr.step(t0, aux0, a, t, aux, nontoken, st, via);  // a finished lane looks up harmlessly
    const bool act = !done;
    my_steps += act ? 1u : 0u;
    const bool is_eps = a == epsilon;
    // A fused cell stands for: this rune has no arc here, the epsilon state remembered at this
    // very rune is t0 itself (matrix.go:442-454), take its epsilon arc to `via`, then the rune
    // from there.  Not taken where the reference would first retry with the unknown symbol
    // (matrix.go:478-485; only observable if the model has such arcs), nor where the rewind
    // would end this lane's chunk (the plain path then stops at the rewind).
    bool comp = false;
    if (TRANS::FUSED) {
      comp = act && good && via != 0 && fresh && !(!ok && a == guard_of(tr)) &&
             !(MODE != MODE_DOC && p >= stop_pos);
      good = good && (via == 0 || comp);
    }
    const bool succ = act && good && !comp, fail = act && !good;
    const bool retry_unknown = fail && !ok && a == identity;               // matrix.go:478-485
    const bool backtrack = fail && !retry_unknown && !is_eps && eps_t != 0;   // matrix.go:487-497
    bool hardfail = fail && !retry_unknown && !backtrack;                  // matrix.go:499-552
    const bool flush_eps = succ && is_eps && p > tp;                       // matrix.go:565-572
    const bool sent_eps = succ && is_eps && p <= tp;                       // matrix.go:573-576
    const bool advance = succ && !is_eps;                                  // matrix.go:579-591
    // matrix.go:593-605: after ANY successful step while `eot` is set.  It is set by the rune just read, cleared by a
    // retry (:555) and by the next rune -- so an epsilon step sees it only in the EOF drain behind a hard fail on a
    // trailing EOT (the hard-fail branch leaves it set, :499-552, and no rune follows to clear it).
    const bool eot_now = (succ || comp) && eot;

    if (hardfail || my_steps > cap) {  // rare
      if (hardfail) {  // drop what is buffered as a token, restart at state 1
        if (is_eps) { st |= ST_BAD_MODEL; done = true; hardfail = false; }  // stale-buffer case
        else if (p <= tp) { p += w; }                                        // matrix.go:515-516
        if (hardfail && p < tp) st |= ST_BAD_OFFSET;  // Token(bufft, buffer[:buffc]) with bufft > buffc
        t = t_start; aux = aux_start;                                        // matrix.go:548
      }
      if (my_steps > cap) { st |= ST_STEP_LIMIT; done = true; hardfail = false; }
    }
    const bool flush_c = comp && p > tp, sent_c = comp && p <= tp;  // the epsilon half of a fused cell
    const bool flush = flush_eps || hardfail || flush_c;
    if (MODE != MODE_START) {
      if (flush)  // matrix.go:528 / 569
        sink.template token<IS_MATRIX>(bs, tp, p, sentence_end || text_end || !any_tok);
      if (sent_eps || sent_c) sink.template sentence<IS_MATRIX>(bs, p, has_tok);  // matrix.go:575
    }
    any_tok = any_tok || flush;
    has_tok = has_tok || flush;
    // consume the rune (for a fused cell: from the epsilon target, right after its rewind, so
    // the rune is the first of the window)
    // (a fused cell's rune is the first of its token if the epsilon half flushed, or if the token was empty; after a
    //  backtrack to a slot BEHIND the token start -- bufft > buffc, the reference's own odd case -- it is neither)
    const bool skip = (advance && p == tp && nontoken) || (comp && nontoken && p >= tp);  // matrix.go:584-588
    const uint32_t p_old = p;
    p = (advance || comp) ? p + w : p;
    tp = skip ? p : (flush_c ? p_old : tp);  // (the epsilon half of a fused cell rewinds only if it flushed)
    // the EOT fires a SentenceEnd unless one is pending (after the epsilon half of a fused cell)
    const bool eot_sent = !((flush_c || flush_eps) ? false : ((sent_c || sent_eps) ? true : sentence_end));
    sentence_end = eot_now ? true : (flush ? false : ((sent_eps || sent_c) ? true : sentence_end));
    text_end = eot_now ? true : (flush ? false : text_end);
    // retries keep the rune, everything else fetches a new one
    t0 = backtrack ? eps_t : t0; aux0 = backtrack ? eps_aux : aux0;
    p = backtrack ? eps_p : p;
    a = backtrack ? epsilon : (retry_unknown ? unknown : a);
    eot = (retry_unknown || backtrack) ? false : eot;  // matrix.go:555: a retry forgets that the rune was EOT
    newchar = succ || hardfail || comp;
    eot = eot_now ? false : eot;  // matrix.go:594
    const bool rewind = flush || (IS_MATRIX && eot_now);  // matrix.go:601 vs datok.go:1019-1030
    eps_t = (backtrack || rewind || comp) ? 0u : eps_t;
    if (TRANS::FUSED) {
      // the epsilon target is the state the rune was read in: remembered if it has an epsilon arc
      const bool he2 = comp && !(IS_MATRIX && eot_now) && tr.has_eps(via, 0u);
      eps_t = he2 ? via : eps_t; eps_p = he2 ? p_old : eps_p;
    }
    // rare: EOT calls, the reference's 1024-rune window limit (checked where the window was
    // longest), end of this lane's chunk
    // a fused cell rewinds before its rune: that rewind is at p_old, never the end of the chunk
    const bool rewind_end = (flush && !comp) || (IS_MATRIX && eot_now);
    const bool long_win = hi - bs > DTK_WINDOW_BYTES;  // overflowed for certain: the lane stops (see walk_fused)
    // (a hard fail on the document's last rune, an EOT: `eot` stays set for the EOF drain, and a start record has no
    //  place for it -- the lane that read the rune runs the drain itself)
    const bool at_stop = rewind_end && MODE != MODE_DOC && p >= stop_pos && !(hardfail && eot && p >= len);
    if (eot_now || (rewind && hi - bs > DTK_WINDOW) || at_stop || long_win) {
      if (eot_now) {
        // (fired by an epsilon step -- the stale `eot` -- the TextEnd follows a Token that ends at the same position:
        //  rows in call order, the exact pass)
        if (is_eps && MODE != MODE_START) sink.out_of_order();
        // (the epsilon half of a fused cell has rewound the window to p_old before its rune was read)
        if (MODE != MODE_START) sink.template eot<IS_MATRIX>(flush_c ? p_old : bs, p, eot_sent, has_tok);
        has_tok = false;  // TextEnd: pos = pos[:0] (token_writer.go:158)
      }
      if (rewind) {
        if (hi - bs > DTK_WINDOW && count_runes(s, bs, hi) > DTK_WINDOW) st |= ST_WINDOW_OVERFLOW;
        if (at_stop) {
          fin.p = p; fin.t = t; fin.aux = aux;
          fin.flags = (sentence_end ? LANE_F_SENT : 0u) | (text_end ? LANE_F_TEXT : 0u) |
                      (ok ? LANE_F_OK : 0u);
          stopped = true;
          done = true;
        }
      }
      // (the tail below then closes the document at this position: every document keeps its TextEnd)
      if (long_win && !done) { st |= ST_WINDOW_OVERFLOW; done = true; }
    }
    tp = rewind_end ? p : tp;  // matrix.go:537-543 / 608-627
    bs = rewind_end ? p : (flush_c ? p_old : bs);
  } while (!done);

  if (!stopped && !(st & (ST_STEP_LIMIT | ST_BAD_MODEL))) {
    if (hi - bs > DTK_WINDOW && count_runes(s, bs, hi) > DTK_WINDOW) st |= ST_WINDOW_OVERFLOW;
    if (MODE != MODE_START) {
      if (p > tp) {  // matrix.go:671-678
        sink.template token<IS_MATRIX>(bs, tp, p, sentence_end || text_end || !any_tok);
        sentence_end = false; text_end = false;
        has_tok = true;
      }
      sink.template tail<IS_MATRIX>(bs, p, sentence_end, text_end, has_tok);  // matrix.go:683-691
    }
  }
  st_out = st;
  steps_out = my_steps;
}

// The same walk for the common case -- matrix with fused cells, no arc on the `unknown` symbol
// anywhere (so the sticky `ok` and the retry of matrix.go:478-485 have no observable effect) --
// written for a short instruction stream: one symbol prefetch and one cell load per iteration,
// lane state in plain integers, and one guarded block for everything that happens less than
// once per token (hard fail, EOT, end of input, end of the chunk, the window limit).
// Behaviour is identical to walk_lane<MatrixFusedTrans, true, MODE> for such models.
//
// Symbol stream: every lane keeps a window of DTK_WIN entries of its own stretch of the stream in
// LDS (its private row; rows are 72 B apart so that the 64 lanes start in different banks).  The
// window is filled with four 16-byte loads per lane and read with one ds_read_u16 per iteration;
// when any lane of the wave leaves its window all lanes re-base theirs (a wave-uniform branch,
// once per ~28 iterations).  Read straight from memory in 8-byte groups, the lanes' 64 stream
// lines and the table lines evict each other from the 32 KiB L1 and every group load goes to L2.
#ifdef DTK_PROBE
// (scripts/probe.py) cycles the waves spend waiting for the cell and the entry at the end of an iteration / in the
// loop / iterations / waves -- chunk walks [0..3], warm-up walks [4..7]
__device__ unsigned long long g_probe[8];
extern "C" int dtk_probe_read(unsigned long long *out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_probe), sizeof(g_probe));
  if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_probe), z, sizeof(z)); }
  return 0;
}
#endif

template <int MODE, bool FIRST = false, bool IS_MATRIX = true>
__device__ __forceinline__ void walk_fused(const MatrixFusedTrans &tr, const DtkSym &sym,
                                           uint64_t off, uint32_t len, DtkLaneState init, uint32_t stop_pos,
                                           EventSink &sink, uint32_t epsilon, uint32_t cap, DtkLaneState &fin,
                                           uint32_t &st_out, uint32_t &steps_out, uint16_t *win_row,
                                           const uint16_t *lut) {
  const DtkSymAt s{sym, off};
  const uint32_t *__restrict__ tab = tr.tab;
  const uint32_t stride = tr.stride, n_eps = tr.n_eps;
  uint32_t t = init.t;
  uint32_t p = init.p, tp = init.p, bs = init.p, hi = init.p;
  uint32_t eps_t = 0, eps_p = 0;
  // F: 1 sentenceEnd, 2 textEnd (matrix.go:360-363), 4 some Token call happened in this document,
  //    8 the current text has a token (what NewTokenWriter's sentB / pos need)
  uint32_t F = (init.flags & (LANE_F_SENT | LANE_F_TEXT)) | (init.p > 0 ? 4u : 0u);
  F |= (init.p > 0 && !(init.flags & LANE_F_TEXT)) ? 8u : 0u;
  static_assert(LANE_F_SENT == 1u && LANE_F_TEXT == 2u, "flag layout");
  uint32_t st = 0;
  uint32_t budget = cap;  // lookups left; the one that finds none left sets ST_STEP_LIMIT
  fin.p = 0xFFFFFFFFu; fin.t = 0; fin.aux = 0; fin.flags = 0;  // p stays "ran to EOF" unless the lane stops
  bool done = false;
  // the lane's window of the symbol stream: the CODE of position q at row[q - wb7], q - wb7 in [0, DTK_WIN8); its
  // entry is lut[code] (the model's code table, in LDS)
  dtk_u8a *row = reinterpret_cast<dtk_u8a *>(win_row);
  const uint32_t o7 = (uint32_t)(off & 15u);  // (windows start at multiples of 16 codes: 16-byte loads)
  const uint8_t *__restrict__ aligned = static_cast<const uint8_t *>(sym.base) + (off - o7);
#define DTK_ENTRY(q_) ((uint32_t)lut[row[(q_) - wb7]])
#define DTK_REFILL(q_) { wb7 = (((q_) + o7) & ~15u) - o7; win_fill8(row, aligned + (wb7 + o7)); }
  uint32_t wb7;
  DTK_REFILL(p)
  // The entry the next lookup is made with: the stream entry of the rune at p -- or, right after a backtrack,
  // the bare epsilon symbol: width 0, so that iteration consumes nothing and reads no rune (matrix.go:487-497).
  uint32_t e = DTK_ENTRY(p);

  // Reader at EOF before a rune is read (matrix.go:650-668): epsilon arcs are taken as long as the state has one
  // (here, on the spot: one lookup each); then the remembered epsilon state is popped -- the walk goes on from
  // there with an epsilon iteration -- or the walk is over.
  auto eof_drain = [&](const bool eot_stale) __attribute__((always_inline)) {
    if (p >= len) {
      bool first_ = true;
      while (t <= n_eps && !done) {
        const uint32_t x_ = tab[__umul24(t, stride) + epsilon];
        const bool ov_ = __builtin_usub_overflow(budget, 1u, &budget);
        if ((int32_t)x_ <= 0) { st |= ST_BAD_MODEL; done = true; break; }
        if (p > tp) { /* matrix.go:565-572 */
          if (MODE != MODE_START) sink.template token<IS_MATRIX>(bs, tp, p, ((F ^ 4u) & 7u) != 0);
          F = 12u;
          if (hi - bs > DTK_WINDOW && count_runes(s, bs, hi) > DTK_WINDOW) st |= ST_WINDOW_OVERFLOW;
          tp = p; bs = p; eps_t = 0;
          if (MODE != MODE_DOC && p >= stop_pos) {
            fin.p = p; fin.t = x_ & 0x7FFFu; fin.aux = 0; fin.flags = init.flags & LANE_F_OK;
            done = true;
          }
        } else { /* matrix.go:573-576 */
          if (MODE != MODE_START) sink.template sentence<IS_MATRIX>(bs, p, (F & 8u) != 0);
          F |= 1u;
        }
        t = x_ & 0x7FFFu;
        if (eot_stale && first_ && !done) { /* matrix.go:593-605 behind the first successful step, see the hard-fail block */
          /* (a TextEnd behind a Token that ends at the same position: rows in call order, the exact pass) */
          if (MODE != MODE_START) sink.out_of_order();
          if (MODE != MODE_START) sink.template eot<IS_MATRIX>(bs, p, (F & 1u) == 0u, (F & 8u) != 0);
          F = (F & 4u) | 3u;
          if (IS_MATRIX) {
            eps_t = 0; tp = p; bs = p;
            if (MODE != MODE_DOC && p >= stop_pos) {
              fin.p = p; fin.t = t; fin.aux = 0; fin.flags = (F & 3u) | (init.flags & LANE_F_OK);
              done = true;
            }
          }
        }
        first_ = false;
        if (ov_) { st |= ST_STEP_LIMIT; done = true; }
      }
      if (!done) {
        if (eps_t != 0) { t = eps_t; p = eps_p; eps_t = 0; e = epsilon; } else done = true;
      }
    }
  };
  eof_drain(false);
  // The loop is rotated: the cell of the NEXT lookup is requested as soon as this one's cell says where the walk goes
  // (a dozen instructions behind its arrival), and everything else an iteration does -- events, token window, flags,
  // the epsilon slot -- runs while that request is under way.  In program order the whole iteration used to stand
  // between a cell's arrival and the next request, and a wave issues in order.  What the rare block decides (hard
  // fail, EOT, EOF drain) is not known yet when the request leaves: it asks again.
  //   x  : the cell (t, e)          en : the stream entry behind the rune at p (position p + width(e))
  uint32_t x = 0, en = 0;
#define DTK_TAB(t_, e_) (*reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(tab) +                  \
                                                            ((__umul24((t_), stride) + ((e_) & DTK_SYM_MASK)) << 2)))
  // the fused table is at most 2^15 states x 2^11 symbols x 4 B: a 32-bit byte offset from the
  // uniform base (one 24-bit multiply-add) instead of 64-bit address arithmetic
  if (!done) {
    x = DTK_TAB(t, e);
    const uint32_t pn0 = p + ((e >> DTK_SYM_W_SHIFT) & 7u);
    if (pn0 - wb7 >= DTK_WIN8) DTK_REFILL(pn0)
    en = DTK_ENTRY(pn0);
  }
#ifdef DTK_PROBE
  unsigned long long pr_wait = 0, pr_t0 = clock64(), pr_n = 0, pr_refill = 0, pr_nrefill = 0;
#endif
  while (!done) {
    // (the lookup cap as a budget counted down: the borrow of the subtraction is the test -- one instruction, not two)
    const bool over = __builtin_usub_overflow(budget, 1u, &budget);
    const uint32_t w = (e >> DTK_SYM_W_SHIFT) & 7u;     // bytes of the rune at p; 0: an epsilon iteration
    const uint32_t pn = p + w;
    // matrix.go:442-454.  (An epsilon iteration -- state and position of the slot it was popped from -- would put
    // the same slot back; it is dropped again below because every epsilon step drops it.)
    const bool he = t <= n_eps;
    eps_t = he ? t : eps_t; eps_p = he ? p : eps_p;
    const bool r = w == 0u;
    const uint32_t tgt = x & 0x7FFFu, via = (x >> 16) & 0x7FFFu;
    const bool comp = (int32_t)x < 0;                   // a fused cell: the epsilon arc of t, then the rune from there
    const bool plain = (int32_t)x > 0;
    const bool fail = x == 0u;
    const bool advance = comp || (plain && !r);         // the rune is consumed, matrix.go:579-591
    const bool backtrack = fail && !r && eps_t != 0;    // matrix.go:487-497
    // ---- where the walk goes, and the request for its cell
    const uint32_t t_n = backtrack ? eps_t : (fail ? t : tgt);
    const uint32_t p_n = backtrack ? eps_p : (advance ? pn : p);
    // right after a backtrack the bare epsilon symbol: width 0, that iteration consumes nothing (matrix.go:487-497)
    const uint32_t e_n = backtrack ? epsilon : en;
    const uint32_t x_n = DTK_TAB(t_n, e_n);
    // nontoken && (comp || (advance && p == tp)), matrix.go:584-588.  (As lane masks combined in scalar registers,
    //  and here, in the block of the comparisons: written with && / || or & / | further down the compiler builds
    //  the predicate from 0/1 integers in vector registers, seven instructions instead of two.)
    const unsigned long long m_comp = __builtin_amdgcn_ballot_w64((int32_t)x < 0),
                             m_adv = m_comp | (__builtin_amdgcn_ballot_w64((int32_t)x > 0) & ~__builtin_amdgcn_ballot_w64(w == 0u));
    // (a fused cell's rune is the first of its token unless the walk has backtracked to a slot BEHIND the token
    //  start -- bufft > buffc, the reference's own odd case: then its epsilon half neither flushes nor rewinds)
    const unsigned long long m_skip = __builtin_amdgcn_ballot_w64((x & 0x8000u) != 0u) &
                                      ((m_comp & __builtin_amdgcn_ballot_w64(p > tp)) | (m_adv & __builtin_amdgcn_ballot_w64(p == tp)));
    uint32_t code_n;  // (its entry is looked up at the end of the iteration: the code has arrived by then)
    {
      const uint32_t pn_n = p_n + ((e_n >> DTK_SYM_W_SHIFT) & 7u);
      uint32_t iw = pn_n - wb7;
      if (__builtin_amdgcn_ballot_w64(iw >= DTK_WIN8) != 0ull) {  // also a backtrack to before the window
#ifdef DTK_PROBE
        const unsigned long long r0_ = clock64();
#endif
        DTK_REFILL(pn_n)
        iw = pn_n - wb7;
#ifdef DTK_PROBE
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        pr_refill += clock64() - r0_;
        pr_nrefill++;
#endif
      }
      code_n = row[iw];
    }
    // ---- this iteration's bookkeeping, under the request
    hi = max(hi, pn);                                   // matrix.go:388-408
    const bool epsE = comp || (plain && r);             // an epsilon arc is taken at p
    const bool flush = epsE && p > tp;                  // matrix.go:565-572
    const bool sentE = epsE && p <= tp;                 // matrix.go:573-576
    const bool hardfail = fail && !backtrack;
    // (first pass: an epsilon step at or behind the stop position is the rare block's -- the lane's last token, or a
    //  SentenceEnd on its way there; the positions before it need no test, see EventSink::put_first)
    const bool beyond = MODE != MODE_DOC && p >= stop_pos;
    const uint32_t tp_old = tp, F_old = F;
    if (MODE != MODE_START && !(DTK_KO & 1)) {
      if (FIRST) {
        if (flush && !beyond) sink.template token_first<IS_MATRIX>(tp, p, ((F ^ 4u) & 7u) != 0);
        if (sentE && !beyond) sink.sentence_first(p, (F & 8u) != 0);
      } else {
        if (flush) sink.template token<IS_MATRIX>(bs, tp, p, ((F ^ 4u) & 7u) != 0);
        if (sentE) sink.template sentence<IS_MATRIX>(bs, p, (F & 8u) != 0);
      }
    }
    const uint32_t win = (DTK_KO & 2) ? 0u : hi - bs;   // bytes the window holds (before this iteration's rewind)
    const uint32_t bs_old = bs, p_old = p;
    F = flush ? 12u : (F | (sentE ? 1u : 0u));
    bs = flush ? p_old : bs;

    tp = flush ? p_old : tp;  // (a fused cell with p <= tp takes its epsilon arc without a rewind)
    tp = __builtin_amdgcn_inverse_ballot_w64(m_skip) ? pn : tp;
    // the epsilon slot: dropped by a backtrack and by every epsilon step; a fused cell remembers the state it
    // read its rune in (the epsilon target, at p_old) if that state has an epsilon arc
    const bool he2 = comp && via <= n_eps;
    eps_t = he2 ? via : ((backtrack || epsE) ? 0u : eps_t);
    eps_p = he2 ? p_old : eps_p;
    const uint32_t e_cur = e, e_next = en;
    p = p_n; t = t_n; e = e_n; x = x_n; en = lut[code_n];
    // everything that happens less than once per token: hard fail, EOT, the first rewind at or behind the end of
    // the chunk (a fused cell's too: the lane then ends BEFORE the cell's rune), the window limit, the lookup cap,
    // the reader at EOF
    const bool eot_now = advance && ((e_cur >> DTK_SYM_CLS_SHIFT) & 3u) == 1u;  // matrix.go:593-605
    const bool at_stop = flush && beyond;
    // (one chain of bit operations: `||` makes the compiler branch between the tests)
    if (hardfail | eot_now | (beyond & (FIRST ? epsE : flush)) | (flush & (win > DTK_WINDOW)) | (win > DTK_WINDOW_BYTES) |
        over | ((p >= len) & !backtrack)) {
      if (FIRST && MODE != MODE_START && beyond && !(DTK_KO & 1)) {  // what the common path left to this block
        if (flush) sink.template token<IS_MATRIX>(bs_old, tp_old, p_old, ((F_old ^ 4u) & 7u) != 0);
        if (sentE) sink.template sentence<IS_MATRIX>(bs_old, p_old, (F_old & 8u) != 0);
      }
      if (flush && win > DTK_WINDOW && count_runes(s, bs_old, hi) > DTK_WINDOW) st |= ST_WINDOW_OVERFLOW;
      // `eot` of the reference survives a hard fail (matrix.go:499-552 does not clear it; the next rune does): if that
      // rune was the document's last, the first successful epsilon step of the EOF drain -- right below: the hard fail
      // has dropped the epsilon slot, so the drain either takes that step or ends the walk -- fires the EOT's
      // SentenceEnd / TextEnd
      bool eot_stale = false;
      if (at_stop) {
        // the state right after the rewind at p_old: the target of the epsilon arc
        fin.p = p_old; fin.t = comp ? via : tgt; fin.aux = 0;
        fin.flags = init.flags & LANE_F_OK;
        done = true;
      } else {
        if (hardfail) {  // matrix.go:499-552: drop what is buffered as a token, restart at state 1
          if (r) { st |= ST_BAD_MODEL; done = true; }
          else {
            if (p <= tp) { p = pn; }  // matrix.go:515-516
            if (p < tp) st |= ST_BAD_OFFSET;  // Token(bufft, buffer[:buffc]) with bufft > buffc
            e = p == pn ? e_next : e_cur;     // the rune at p (read again if it was not consumed)
            eot_stale = p == pn && p >= len && ((e_cur >> DTK_SYM_CLS_SHIFT) & 3u) == 1u;
            if (MODE != MODE_START) sink.template token<IS_MATRIX>(bs, tp, p, ((F ^ 4u) & 7u) != 0);
            F = 12u;
            if (hi - bs > DTK_WINDOW && count_runes(s, bs, hi) > DTK_WINDOW) st |= ST_WINDOW_OVERFLOW;
            t = tr.start; eps_t = 0;
            tp = p; bs = p;
            // (with a stale `eot` the lane goes on into the EOF drain itself: a start record has no place for it)
            if (MODE != MODE_DOC && p >= stop_pos && !eot_stale) {
              fin.p = p; fin.t = t; fin.aux = 0;
              fin.flags = (init.flags & LANE_F_OK);
              done = true;
            }
          }
        }
        if (eot_now) {
          if (MODE != MODE_START) sink.template eot<IS_MATRIX>(bs, p, (F & 1u) == 0u, (F & 8u) != 0);
          F = (F & 4u) | 3u;  // sentenceEnd, textEnd; TextEnd: pos = pos[:0] (token_writer.go:158)
          if (IS_MATRIX) {    // matrix.go:601 rewinds; the double array keeps window and epsilon slot (datok.go:1019-1030)
            eps_t = 0;
            if (hi - bs > DTK_WINDOW && count_runes(s, bs, hi) > DTK_WINDOW) st |= ST_WINDOW_OVERFLOW;
            tp = p; bs = p;
            if (MODE != MODE_DOC && p >= stop_pos && !done) {
              fin.p = p; fin.t = t; fin.aux = 0;
              fin.flags = (F & 3u) | (init.flags & LANE_F_OK);
              done = true;
            }
          }
        }
        if (over && !done) { st |= ST_STEP_LIMIT; done = true; }
        // More bytes buffered than 1024 runes can have: the reference's window has overflowed for certain
        // (matrix.go:365,406).  The lane stops there -- a blank-free blob of megabytes would otherwise be walked to
        // its end by every lane whose chunk lies inside it.  (The tail below then closes the document at this
        // position: every document keeps its TextEnd.)
        if (hi - bs > DTK_WINDOW_BYTES && !done) { st |= ST_WINDOW_OVERFLOW; done = true; }
      }
      if (!done && !backtrack) eof_drain(eot_stale);
      if (!done) {  // ask again: state, position or entry may have changed
        x = DTK_TAB(t, e);
        const uint32_t pn2 = p + ((e >> DTK_SYM_W_SHIFT) & 7u);
        if (pn2 - wb7 >= DTK_WIN8) DTK_REFILL(pn2)
        en = DTK_ENTRY(pn2);
      }
    }
#ifdef DTK_PROBE
    {  // the end of the iteration: what is left of the wait for the next cell and entry
      const unsigned long long a_ = clock64();
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      pr_wait += clock64() - a_;
      pr_n++;
    }
#endif
  }
#undef DTK_TAB
#undef DTK_ENTRY
#undef DTK_REFILL
#ifdef DTK_PROBE
  {
    // (the wave's clock: every lane reads the same counter; the wave leaves the loop with its last lane)
    const unsigned long long tot_ = clock64() - pr_t0;
    unsigned long long wmax = pr_wait, nmax = pr_n;
    for (int o = 32; o; o >>= 1) {
      const unsigned long long a_ = __shfl_xor(wmax, o), b_ = __shfl_xor(nmax, o);
      wmax = a_ > wmax ? a_ : wmax; nmax = b_ > nmax ? b_ : nmax;
    }
    if (MODE == MODE_CHUNK && lane_id() == 0) {
      atomicAdd(&g_probe[0], wmax); atomicAdd(&g_probe[1], tot_); atomicAdd(&g_probe[2], nmax); atomicAdd(&g_probe[3], 1ull);
    }
    if (MODE == MODE_CHUNK && lane_id() == 0) {  // the wave's refills of the lanes' windows (uniform: the branch is)
      atomicAdd(&g_probe[4], pr_refill); atomicAdd(&g_probe[5], pr_nrefill);
    }
  }
#endif

  if (fin.p == 0xFFFFFFFFu && !(st & (ST_STEP_LIMIT | ST_BAD_MODEL))) {
    if (hi - bs > DTK_WINDOW && count_runes(s, bs, hi) > DTK_WINDOW) st |= ST_WINDOW_OVERFLOW;
    if (MODE != MODE_START) {
      if (p > tp) {  // matrix.go:671-678
        sink.template token<IS_MATRIX>(bs, tp, p, ((F ^ 4u) & 7u) != 0);
        F = (F & ~3u) | 8u;
      }
      sink.template tail<IS_MATRIX>(bs, p, (F & 1u) != 0, (F & 2u) != 0, (F & 8u) != 0);  // matrix.go:683-691
    }
  }
  st_out = st;
  steps_out = cap - budget;  // lookups (modulo 2^32: the budget wraps when it runs out)
}

// the lean walk: fused cells and no arc on `unknown` (MatrixLeanTrans, picked by the launcher)
template <typename TRANS, bool IS_MATRIX, int MODE, bool FIRST = false>
__device__ __forceinline__ void walk_any(const TRANS &tr, const DtkSym &sym, uint64_t off,
                                         uint32_t len, DtkLaneState init, uint32_t stop_pos, EventSink &sink,
                                         uint32_t epsilon, uint32_t unknown, uint32_t identity, uint32_t cap,
                                         DtkLaneState &fin, uint32_t &st_out, uint32_t &steps_out,
                                         uint16_t *win_row, const uint16_t *lut) {
  if constexpr (TRANS::LEAN)
    walk_fused<MODE, FIRST, IS_MATRIX>(tr, sym, off, len, init, stop_pos, sink, epsilon, cap, fin, st_out, steps_out, win_row, lut);
  else
    walk_lane<TRANS, IS_MATRIX, MODE>(tr, sym, off, len, init, stop_pos, sink, epsilon, unknown, identity, cap,
                                      fin, st_out, steps_out, win_row);
}

__device__ __forceinline__ uint32_t step_cap(uint32_t factor, uint32_t len) {
  unsigned long long c = (unsigned long long)factor * ((unsigned long long)len + 2ull);
  return c > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)c;
}

__device__ __forceinline__ void add_steps(unsigned long long *counter, uint32_t mine) {
  unsigned long long tot = mine;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) tot += __shfl_down(tot, o);
  if (lane_id() == 0 && tot) atomicAdd(counter + (blockIdx.x & (DTK_STEP_STRIPES - 1u)) * 16u, tot);  // see DTK_STEP_STRIPES
}

// the lanes' windows of the symbol stream and, for the lean loop, the model's code table (one wave per block)
#ifndef DTK_ROW_PAD
#define DTK_ROW_PAD 0u  // bytes between the lanes' rows of codes.  (8 until round 3, to stagger the banks: without them a wave's LDS is 5728 B at 128-byte chunks and seven waves per SIMD fit a CU; the conflicts of the one-byte reads cost nothing measurable: walk 53.8 -> 51.9 us saturated)
#endif
#define DTK_WINDOWS(TRANS, SYM)                                                                       \
  constexpr uint32_t WIN_ROW_ = TRANS::LEAN ? (DTK_WIN8 + DTK_ROW_PAD) / 2u : DTK_WIN_ROW;            \
  __shared__ uint16_t s_win[WAVE * WIN_ROW_];                                                         \
  __shared__ uint16_t s_lut[TRANS::LEAN ? 256 : 1];                                                   \
  uint16_t *win_row = s_win + threadIdx.x * WIN_ROW_;                                                 \
  if constexpr (TRANS::LEAN) {                                                                        \
    for (uint32_t i_ = threadIdx.x; i_ < 256u; i_ += WAVE) s_lut[i_] = (SYM).lut[i_];                 \
    __syncthreads();                                                                                  \
  }

// codes: the symbol stream holds codes (DtkSym::lut) -- what the lean loop reads
template <typename F>
static int with_trans(const DtkTableDev *tab, bool codes, F &&f) {
  if (tab->kind == DTK_KIND_MATRIX) {
    if (tab->fused) {
      MatrixFusedTrans tr{(const uint32_t *)tab->tab, tab->stride, tab->n_eps, tab->start, tab->ident_guard};
      // (da_dense: a double-array tokenizer laid out as a fused matrix -- the table's walk, datok.go's EOT rules)
      auto call = [&](auto t) { if (tab->da_dense) f(t, std::false_type{}); else f(t, std::true_type{}); };
      if (tab->ident_guard == 0xFFFFFFFFu && !tab->plain_walk && codes) {  // the lean loop applies
        MatrixLeanTrans lt;
        static_cast<MatrixFusedTrans &>(lt) = tr;
        call(lt);
      } else {
        call(tr);
      }
    } else if (tab->entry_bytes == 2) {
      MatrixTrans<uint16_t> tr{(const uint16_t *)tab->tab, tab->stride, tab->n_eps, tab->start};
      f(tr, std::true_type{});
    } else {
      MatrixTrans<uint32_t> tr{(const uint32_t *)tab->tab, tab->stride, tab->n_eps, tab->start};
      f(tr, std::true_type{});
    }
  } else {
    DaTrans tr{(const uint2 *)tab->tab, tab->da_len, tab->da_size, tab->da_base1};
    f(tr, std::false_type{});
  }
  return (int)hipGetLastError();
}

