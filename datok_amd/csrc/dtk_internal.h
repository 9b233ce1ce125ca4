// dtk_internal.h -- structures shared by the host side and the gfx950 kernels.
#pragma once
#include <stddef.h>
#include <stdint.h>

#define DTK_FIRSTBIT 0x80000000u   // datok.go:43
#define DTK_SECONDBIT 0x40000000u  // datok.go:44
#define DTK_RESTBIT 0x3fffffffu    // datok.go:45
#define DTK_EOT 4u                 // matrix.go:13
#define DTK_WINDOW 1024u           // matrix.go:365
#define DTK_WINDOW_BYTES 4104u     // more bytes than 1024 runes can have: the window has overflowed for certain

// ---- symbol stream entry (uint16 per input byte), written by the symbolise kernel
//   [10:0]  sigma index a           (sigmaASCII[c] / sigma[c] / identity, matrix.go:421-435)
//   [13:11] bytes of the rune that starts at this byte (Go DecodeRune width, 1..4); 0: no rune starts here
//   [15:14] class: 0 rune<256, 1 rune==EOT, 2 rune>=256 in sigma (ok=true), 3 not in sigma (ok=false)
// (an entry with width 0 and the epsilon symbol is what the lean walk feeds itself for an epsilon iteration)
#define DTK_SYM_MASK 0x7FFu
#define DTK_SYM_MAX 2047u
#define DTK_SYM_W_SHIFT 11
#define DTK_SYM_CLS_SHIFT 14
#define DTK_SYM_WIDTH(e) (((uint32_t)(e) >> DTK_SYM_W_SHIFT) & 7u)
#define DTK_SYM_IS_START(e) (DTK_SYM_WIDTH(e) != 0u)
// the general walk's per-lane window of the symbol stream in LDS: entries, and u16 per row (72 B)
#define DTK_WIN 32u
#define DTK_WIN_ROW 36u
// ---- the stream in memory: one BYTE per input byte where the model's entries fit a code table (DtkSigmaDev::n_codes:
// the shipped models have some 170 symbols and 200 distinct entries), else the 16-bit entries themselves.  A code
// is an index into `lut` (256 entries); DTK_SYM_CONT = no rune starts here.  Half the stream's traffic -- it was 44 %
// of what a batch moves -- and the lean walk's windows (codes) in 40 instead of 72 bytes of LDS per lane.  The code of a
// byte below 128 is the byte itself (upload() in dtk_host.cpp; k_symbolize copies those).
#define DTK_SYM_CONT 0xFFu
#ifndef DTK_WIN8
#define DTK_WIN8 32u  // positions in a lean-walk row of codes (a multiple of 16; the row has 8 bytes more: 40 B)
#endif
struct DtkSym {
  const void *base;     // uint8_t codes if lut, else uint16_t entries
  const uint16_t *lut;  // [256] entry of a code (lut[DTK_SYM_CONT] has width 0), or null
};
#ifdef __HIP__
__device__ __forceinline__ uint32_t dtk_sym_entry(const DtkSym &S, uint64_t i) {
  return S.lut ? (uint32_t)S.lut[static_cast<const uint8_t *>(S.base)[i]] : (uint32_t)static_cast<const uint16_t *>(S.base)[i];
}
__device__ __forceinline__ bool dtk_sym_is_start(const DtkSym &S, uint64_t i) {
  return S.lut ? static_cast<const uint8_t *>(S.base)[i] != DTK_SYM_CONT
               : DTK_SYM_IS_START(static_cast<const uint16_t *>(S.base)[i]);
}
#endif

// ---- walk output: event bitmaps over cursor positions.  Position p (0..len) of document d, which starts at
// input byte `off`, is bit  G = off + d + p  (one position more than the document has bytes, so the ranges of
// consecutive documents follow each other without sharing a bit).  One bitmap per kind of event, `bit_words`
// 32-bit words each, kind k at bits + k * bit_words; all cleared before a run.  Order of the calls at one cursor
// position: SEOT, TEOT (the cursor only reaches the byte behind an EOT by consuming it, which fires SentenceEnd /
// TextEnd at once, matrix.go:593-600), END (a token that ends there: double array only, which keeps its window),
// SEPS.  START is no call: it marks the first byte of the token whose END is the next END bit.
// The final SentenceEnd / TextEnd (matrix.go:683-691) are two bits of the document's tail word, with the
// cursor they fire at:  doc_tail[d] = cursor << 2 | 1 (SentenceEnd) | 2 (TextEnd).
// The walk collects END / START / SEPS bits in LDS (one set of bitmaps per wave, covering the positions of its
// 64 chunk lanes) and writes whole words out at the end: one scattered byte store per event cost a third of the
// pipeline's throughput (each reaches memory as its own partial-sector write).
enum { EVB_END = 0, EVB_START = 1, EVB_SEPS = 2, EVB_TEOT = 3, EVB_SEOT = 4, EVB_KINDS = 5 };
#define DTK_EV_BIT(off, d) ((uint64_t)(off) + (uint64_t)(d))
#define DTK_TAIL_S 1u
#define DTK_TAIL_E 2u
// LDS words per kind for a wave of 64 chunk lanes of `chunk` bytes: 64 * (chunk + 1) positions + alignment slack
#define DTK_LDS_BIT_WORDS(chunk) (2u * (chunk) + 8u)
#define DTK_LDS_BIT_CHUNK_MAX 256u  // larger chunks write their bits straight to memory

// flags of one queued cursor position inside the compaction (bit order = call order)
#define EV_S_EOT 0x01u
#define EV_E_EOT 0x02u
#define EV_TOK_END 0x04u
#define EV_S_EPS 0x08u
#define EV_S_EOF 0x20u
#define EV_E_EOF 0x40u

// per-document status (mirrors DTK_ST_* of datok_gpu.h)
#define ST_WINDOW_OVERFLOW 1u
#define ST_EMPTY_TEXT 2u
#define ST_BAD_MODEL 4u
#define ST_IRREGULAR 8u
#define ST_STEP_LIMIT 16u
#define ST_BAD_OFFSET 64u  // Token call with its offset behind its buffer (the reference panics when it prints the surface)
#define ST_INTERNAL 32u  // the walk's counts and the compaction disagree (a bug, never expected)

struct DtkSigmaDev {
  const uint16_t *ascii;  // [256] symbol per rune < 256 (identity pre-filled, matrix.go:289-293)
  const uint32_t *runes;  // sorted runes of the sigma map (matrix.go:301)
  const uint16_t *syms;   // their symbols
  uint32_t n_runes;
  uint32_t identity;
  // code table (n_codes != 0: the stream holds codes)
  uint32_t n_codes;
  const uint16_t *code_entry;  // [256] code -> entry
  const uint8_t *code_lt256;   // [256] rune < 256 -> code (one byte wide below 128, two from 128 on)
  const uint8_t *code_runes;   // [n_runes] sigma rune >= 256 in its UTF-8 width -> code
  uint8_t code_ident[5];       // [w] a rune of w bytes that is not in the sigma (identity, ok = false)
  uint8_t code_fffd1;          // an invalid byte (U+FFFD, one byte wide) if U+FFFD is in the sigma
};

enum { DTK_KIND_MATRIX = 0, DTK_KIND_DA = 1 };

// Device table handed to the walk kernels.
// The walk's lookup count (a statistic) is added up in 32 counters a cache line apart, picked by block id: one
// counter took 2048 adds to one address from the waves of a batch as they finished, 5 us at the end of the walk.
#define DTK_STEP_STRIPES 32u
#define DTK_TOTALS_BYTES (128u + DTK_STEP_STRIPES * 128u)  // a batch's totals block + the counters behind it

struct DtkTableDev {
  int kind;
  // matrix: state-major rows, cell (t, a) at tab[t*stride + a]; column 0 is all
  // zero (the a == 0 guard of matrix.go:459).  cell = target | nontoken flag in
  // the top bit.  States are renumbered so that exactly the states 1..n_eps
  // have an epsilon arc (the probe of matrix.go:442 becomes a compare).
  // double array: {base, check} pairs as in the file (datok.go:56-59); bit 30 of
  // base (unused upstream, masked by getBase, datok.go:271-273) caches "this
  // index has an epsilon arc" so the probe of datok.go:876 needs no extra load.
  const void *tab;
  uint32_t entry_bytes;  // 2 or 4 (matrix), 8 (double array)
  uint32_t stride;       // matrix: cells per row
  uint32_t n_states;     // matrix: highest state id
  uint32_t n_eps;        // matrix: states 1..n_eps have an epsilon arc
  uint32_t start;        // image of the reference's state 1
  uint32_t fused;        // matrix: uint32 cells with fused epsilon+rune entries (see MatrixFusedTrans)
  uint32_t ident_guard;  // identity symbol if arcs on `unknown` exist, else 0xFFFFFFFF
  uint32_t plain_walk;   // 1: always use the general walk loop (env DATOK_PLAIN_WALK, for A/B runs and tests)
  uint32_t da_dense;     // 1: a double-array tokenizer whose transitions were laid out as a (fused) matrix at load
  uint32_t da_len;       // double array: pairs
  uint32_t da_size;      // array[1].check & RESTBIT (datok.go:333-335)
  uint32_t da_base1;     // device base word of index 1
  uint32_t epsilon, unknown, identity;
};

// ---- speculative chunk lanes
#define LANE_F_SENT 1u     // sentenceEnd (matrix.go:360)
#define LANE_F_TEXT 2u     // textEnd (matrix.go:363)
#define LANE_F_OK 4u       // sticky ok (matrix.go:352)
#define LANE_F_DROPPED 8u  // the lane produced an event outside its window
#define LANE_F_IDLE 16u
enum { PLAN_OFF = 0, PLAN_CHAINED = 1, PLAN_LAST = 2 };

// Loop state at a "sync point" = right after the reference rewinds its window
// (matrix.go:537-543, 608-627): nothing else survives a rewind.
struct DtkLaneState {
  uint32_t p;      // byte position in the document; 0xFFFFFFFF = none / ran to EOF
  uint32_t t;      // automaton state
  uint32_t aux;    // double array: device base word of t
  uint32_t flags;  // LANE_F_*
};
struct DtkLanePlan {
  uint32_t stop;   // stop at the first sync point at or behind this position
  uint32_t wend;   // events are stored up to here (the next lane's start position)
  uint32_t mode;   // PLAN_*
  uint32_t pad;
};
struct DtkLaneCount {
  uint32_t tok, sent, text;  // Token calls, ints of the sentence list, TextEnd calls
  uint32_t status;
  // for compacting a long document in segments (k_seg_*): SentenceEnd calls, and the lane's last
  // TextEnd fired by an EOT: its position (0xFFFFFFFF = none) and the lane's Token calls before it
  uint32_t sev, e_pos, e_tok;
  uint32_t pad;
};
struct DtkSpecArgs {
  uint32_t n_lanes;
  uint32_t chunk, warm;             // chunk size C and warm-up overlap W in bytes
  const uint32_t *lane_doc;         // lane -> document
  const uint32_t *chunk_off;        // document -> first lane (n_docs + 1)
  struct DtkLaneState *lane_start;  // record each lane starts from
  struct DtkLaneState *lane_end;    // where it stopped
  struct DtkLanePlan *lane_plan;
  struct DtkLaneCount *lane_cnt;
  uint32_t *first_bad;              // per document: first chunk without a linked successor
  uint32_t *fail_lane;              // per document: first lane that missed its successor's record
  const uint32_t *redo_from;        // repair rounds: first lane to redo per document, or null
  const uint8_t *text;              // input bytes (k_spec_start: whitespace-guided warm-up), or null
  uint32_t warm_ws;                 // start the warm-up behind the warm_ws-th whitespace run before the chunk (0: fixed)
  uint32_t warm_min;                // ... looking backwards from chunk start - warm_min
  uint32_t first_repair;            // repair rounds: 1 in the round that follows the first pass
  uint32_t warm_extend;             // move the warm-up start back to the previous blank, at most this many bytes (0: off)
  uint32_t lds_words;               // LDS bitmap words per kind for one wave (0: event bits go straight to memory)
  // A repair round enqueued ahead of time (device-side repair): its kernels return at once unless *go != 0
  // (the number of documents the previous verification found broken); null: run.
  const uint32_t *go;
};

struct DtkWalkArgs {
  struct DtkSym sym;        // symbol stream, one entry per input byte
  const uint64_t *doc_off;  // n_docs + 1
  uint32_t n_docs;
  uint32_t *bits;           // event bitmaps (EVB_KINDS x bit_words words), zero-filled
  uint32_t bit_words;
  uint32_t *doc_tail;       // per document, zero-filled
  uint32_t *status;         // per document, OR-ed
  uint64_t *tok_cnt, *sent_cnt, *text_cnt;  // per document: what the writer would have collected
  unsigned long long *steps;  // global lookup counter
  uint32_t step_factor;     // cap = step_factor * (len + 2) lookups per document
};

struct DtkCompactArgs {
  const uint8_t *text;
  const uint32_t *rs_bits;  // bit g: input byte g starts a rune (k_symbolize)
  const uint64_t *doc_off;
  uint32_t n_docs;
  const uint32_t *bits;     // event bitmaps of the walk
  uint32_t bit_words;
  const uint32_t *doc_tail;
  uint32_t *status;
  uint32_t flags;           // DTK_NEWLINE_AFTER_EOT
  int kind;                 // matrix / double array (EOT rewind rule differs)
  // pass 1 output: per document counts (tokens, sentence ints, texts)
  uint64_t *tok_off, *sent_off, *text_off;  // n_docs+1; counts in [d], scanned in place
  // pass 2 output
  int32_t *tok_rstart, *tok_rend;
  uint32_t *tok_bstart, *tok_bend;
  int32_t *sent;
  uint32_t *text_tok_end, *text_sent_end;
  // for the renderer: SentenceEnd calls (of the document) before each Token / TextEnd call, and per document
  uint32_t *tok_sbefore, *text_s_end, *doc_ns;
  const uint64_t *totals;   // [0..2] tokens, sentence ints, texts (written by the scan)
  uint64_t tok_cap, sent_cap, text_cap;
  // Long documents are compacted in segments of DTK_SEG_LANES chunk lanes (matrix walk with chunk
  // lanes only; null = one wave per document): segment -> document / first lane / lanes
  const uint32_t *seg_doc, *seg_lane0, *seg_nl;
  uint32_t n_segs;
  const uint32_t *chunk_off;              // document -> first lane
  const struct DtkLaneState *lane_start;  // sync points = where segments begin
  const struct DtkLaneCount *lane_cnt;
  struct DtkSegSum *seg_sum;              // k_seg_sum: what a segment adds
  struct DtkSegIn *seg_in;                // k_seg_scan: the carries a segment starts with
  uint32_t *doc_seq;                      // k_seg_scan: 1 = this long document must be compacted sequentially
  uint32_t *any_irregular;                // set to 1 if a document is flagged ST_IRREGULAR (the host then runs the exact pass)
  uint32_t *any_eot;                      // set to 1 by k_compact_plain if a document is left to k_compact_eot
  const uint32_t *skip_if;                // documents still to repair: the pass does nothing unless this is 0 (null: run)
};

#define DTK_SEG_LANES 64u
// what the lanes of one segment add up to (k_seg_sum)
struct DtkSegSum {
  uint32_t tok, sent, text, sev;  // Token calls, sentence ints, TextEnd calls, SentenceEnd calls
  uint32_t runes;                 // rune starts in the segment's positions
  uint32_t e_pos;                 // last EOT TextEnd inside (0xFFFFFFFF: none) ...
  uint32_t e_tok, e_runes;        // ... Token calls / rune starts of the segment before it
};
// the state k_compact starts a segment with: everything that precedes it in the document
struct DtkSegIn {
  uint32_t tok, sent, text, sev, runes;
  uint32_t e_pos, e_tok, e_runes;  // last EOT TextEnd before the segment (absolute), 0xFFFFFFFF: none
};

// ---- the exact pass: documents whose call order the position-indexed event bytes cannot express
// (ST_IRREGULAR: the double array consuming one EOT twice after a backtrack, datok.go:916-926 +
// 1019-1030; a third epsilon SentenceEnd at one cursor, matrix.go:573-576) are walked again by one
// lane each, in the reference's own order, writing their rows of the result arrays directly and
// listing their calls (dtk_call of datok_gpu.h) for closure replays.
struct DtkCall { uint32_t kind; int32_t a; uint32_t b, c; };
struct DtkExactArgs {
  struct DtkSym sym;
  const uint8_t *text;
  const uint64_t *doc_off;
  uint32_t n;               // documents to walk
  const uint32_t *docs;     // their ids
  uint32_t pass;            // 0: count the calls (n_calls[i]); 1: write rows and calls
  uint32_t *n_calls;        // [n]
  const uint64_t *call_off; // [n + 1], pass 1: calls of docs[i] at calls[call_off[i] ..)
  struct DtkCall *calls;
  uint32_t *status;
  uint32_t flags;           // DTK_NEWLINE_AFTER_EOT
  uint32_t step_factor;
  const uint64_t *tok_off, *sent_off, *text_off;  // CSR rows (sized by the walk's counts)
  int32_t *tok_rstart, *tok_rend;
  uint32_t *tok_bstart, *tok_bend;
  int32_t *sent;
  uint32_t *text_tok_end, *text_sent_end;
  uint32_t *tok_sbefore, *text_s_end, *doc_ns;    // renderer bookkeeping (may be null)
};

// NewTokenWriter's byte output on the device (dtk_render.hip)
struct DtkRenderArgs {
  const uint8_t *text;
  const uint64_t *doc_off;
  uint32_t n_docs;
  uint32_t flags;  // TOKENS 1, SENTENCES 2, TOKEN_POS 4, SENTENCE_POS 8
  const uint64_t *tok_off, *sent_off, *text_off;  // CSR rows per document
  uint64_t n_tok, n_sent, n_text;
  const int32_t *rstart, *rend, *sent;
  const uint32_t *bstart, *bend, *sbefore;  // per token
  const uint32_t *ttok, *tsent, *ts_end;    // per text: tokens / sentence ints / SentenceEnd calls up to its TextEnd
  const uint32_t *doc_ns;                   // SentenceEnd calls per document
  struct DtkSym sym;                        // symbol stream; base non-null only if the batch has invalid UTF-8 bytes
  // workspace
  uint64_t *A, *P, *Q;           // exclusive scans: surface bytes, position digits (n_tok+1), sentence digits (n_sent+1)
  uint64_t *blkA, *blkP, *blkQ;  // per-tile sums
  uint64_t *ns_off;              // n_docs+1
  uint64_t *tx_base, *tx_stream, *tx_pos, *tx_sent;  // n_text+1
  // output
  uint64_t *out_off;  // n_docs+1
  uint8_t *out;
  uint64_t out_total;
};

// Results straight into page-locked host memory from the batch's own stream (k_to_host): the sizes of the offset
// arrays are only known on the device when the copies have to be enqueued, so a copy engine cannot be asked ahead of
// time -- a kernel can: it reads the totals and streams the rows over the link with 16-byte stores.
#define DTK_TOHOST_MAX 14
struct DtkToHostArgs {
  const void *src[DTK_TOHOST_MAX];
  void *dst[DTK_TOHOST_MAX];           // mapped page-locked host memory
  uint64_t bytes[DTK_TOHOST_MAX];      // size in bytes (count_from < 0), else bytes per element
  int32_t count_from[DTK_TOHOST_MAX];  // < 0: fixed size; 0..2: totals[count_from] elements
  uint64_t cap[DTK_TOHOST_MAX];        // elements the destination holds (count_from >= 0)
  uint32_t n;
  const uint64_t *totals;              // device totals block (scan3)
  const uint32_t *skip_if;             // documents still to repair: nothing is copied unless this is 0 (null: copy)
  uint64_t *done;                      // device word: set to the run's epoch when the copy was made
  uint64_t epoch;
};

#ifdef __cplusplus
extern "C" {
#endif
int dtk_launch_to_host(const struct DtkToHostArgs *args, void *stream);
int dtk_launch_pack_r16(const int32_t *rs, const int32_t *re, uint32_t *out, uint64_t n, void *stream);
// launchers (dtk_symbolize / dtk_walk / dtk_repair / dtk_compact .hip); stream is a hipStream_t
int dtk_launch_symbolize(const uint8_t *text, const uint64_t *doc_off, uint32_t n_docs,
                         uint64_t total, const struct DtkSigmaDev *sig, void *sym, int padded,
                         const uint32_t *blk_doc, unsigned long long *n_invalid, uint32_t *rs_bits,
                         uint32_t *ev_bits, uint32_t bit_words, void *acc, uint64_t acc_bytes, uint64_t epoch, void *stream);
#ifndef DTK_SYM_BLOCK_BYTES
#define DTK_SYM_BLOCK_BYTES 4096u  // input bytes per symbolise block (blk_doc granularity)
#endif
int dtk_launch_walk(const struct DtkTableDev *tab, const struct DtkWalkArgs *args, void *stream);
int dtk_launch_spec(const struct DtkTableDev *tab, const struct DtkWalkArgs *args,
                    const struct DtkSpecArgs *spec, int stage, uint32_t cmp_mask, uint32_t *redo_out,
                    uint32_t *n_bad, void *stream);
int dtk_launch_spec_check(const struct DtkWalkArgs *args, const struct DtkSpecArgs *spec, int stage, uint32_t cmp_mask,
                          uint32_t *redo_out, uint32_t *n_bad, void *stream);
int dtk_launch_redo_clear(const struct DtkWalkArgs *args, const struct DtkSpecArgs *spec, void *stream);
int dtk_launch_compact(const struct DtkCompactArgs *args, uint32_t small_max, const uint32_t *big_docs, uint32_t n_big,
                       int which, void *stream);
int dtk_launch_exact(const struct DtkTableDev *tab, const struct DtkExactArgs *args, void *stream);
int dtk_launch_seg_prepare(const struct DtkCompactArgs *args, const uint32_t *doc_seg0, void *stream);
int dtk_launch_render(const struct DtkRenderArgs *args, int stage, void *stream);
uint32_t dtk_render_tiles(uint64_t n);
int dtk_launch_clear2(void *a, uint64_t a_bytes, void *b, uint64_t b_bytes, void *stream);
int dtk_launch_scan3(const uint64_t *ca, const uint64_t *cb, const uint64_t *cc, uint64_t *a, uint64_t *b,
                     uint64_t *c, uint32_t n_docs, uint64_t *totals, const uint32_t *status, uint64_t *ws,
                     const struct DtkSpecArgs *fix_spec, uint32_t *redo_out, uint32_t *n_bad, const uint32_t *skip_if,
                     void *stream);
#ifdef __cplusplus
}
#endif
