/*
 * datok_oracle.h -- CPU restatement of the Datok tokenizer hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle for the HIP path in
 * datok_amd/csrc: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may call it.  The product (libdatok_gpu.so) never links,
 * loads or falls back to anything in this directory.
 *
 * Parity status: PINNED.  The restatement reproduces the reference's own
 * golden strings (tests/golden/ JSON files, transcribed from matrix_test.go,
 * datok_test.go, token_writer_test.go) -- see tests/test_oracle_golden.py.
 * Unpinned corner: invalid UTF-8 input (no reference test feeds any); it is
 * specified from Go's unicode/utf8.DecodeRune documentation.
 *
 * Every function cites the reference file:line (relative to the upstream
 * KorAP/Datok tree) whose behaviour it restates.
 */
#ifndef DATOK_ORACLE_H
#define DATOK_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* token_writer.go:17-25 */
enum {
  ORC_TOKENS = 1,
  ORC_SENTENCES = 2,
  ORC_TOKEN_POS = 4,
  ORC_SENTENCE_POS = 8,
  ORC_NEWLINE_AFTER_EOT = 16,
  ORC_SIMPLE = 3
};

/* per-document status bits: inputs on which the reference would panic */
enum {
  ORC_ST_WINDOW_OVERFLOW = 1, /* matrix.go:365,406 buffer[1024] index panic */
  ORC_ST_EMPTY_TEXT = 2,      /* token_writer.go:108,135,145 pos[-1]/pos[0] panic */
  ORC_ST_BAD_MODEL = 4,       /* walk left the table / would read stale buffer */
  ORC_ST_BAD_OFFSET = 64      /* Token(offset, buf) with offset > len(buf): string(buf[offset:]) panics
                                 (token_writer.go:85,93); same bit as DTK_ST_BAD_OFFSET */
};

enum { ORC_KIND_MATRIX = 0, ORC_KIND_DA = 1 };

typedef struct orc_model orc_model;

/* fomafile.go:452-484 LoadTokenizerFile: gunzip, sniff magic, dispatch. */
orc_model *orc_load_file(const char *path);
/* matrix.go:235-337 ParseMatrix / datok.go:621-729 ParseDatok on raw bytes. */
orc_model *orc_parse(const uint8_t *raw, size_t n);
/* fomafile.go:56-450 LoadFomaFile/ParseFoma + matrix.go:30-99 ToMatrix: Foma text net -> matrix. */
orc_model *orc_load_foma_file(const char *path);
orc_model *orc_parse_foma(const uint8_t *raw, size_t n);
void orc_free_model(orc_model *m);
const char *orc_type(const orc_model *m); /* matrix.go:102, datok.go:252 */

typedef struct {
  int kind, epsilon, unknown, identity, final_, sigma_count;
  uint32_t state_count; /* matrix: stateCount; DA: array[1].check (size) */
  uint64_t array_len;   /* matrix: u32 cells; DA: bc pairs */
  int n_sigma_runes;    /* entries of the sigma map */
} orc_model_info;
void orc_info(const orc_model *m, orc_model_info *out);
const uint32_t *orc_array(const orc_model *m);       /* raw table as parsed */
const int *orc_sigma_ascii(const orc_model *m);      /* [256] */
int orc_sigma_lookup(const orc_model *m, uint32_t rune, int *ok);

/* Go unicode/utf8.DecodeRune on p[0..n): returns width, *r = rune. */
int orc_decode_rune(const uint8_t *p, size_t n, uint32_t *r);

/*
 * Rendered output: what TransduceTokenWriter(reader, NewTokenWriter(w, flags))
 * writes to w (matrix.go:348-698 / datok.go:781-1135 + token_writer.go:36-175).
 * Returns a malloc'd buffer (caller frees with orc_free) and its length.
 * status receives ORC_ST_* bits.
 */
char *orc_transduce_string(const orc_model *m, const uint8_t *text, size_t n,
                           unsigned flags, size_t *out_len, unsigned *status);

/*
 * Structured result for one document, i.e. one TransduceTokenWriter call with
 * a fresh TOKEN_POS|SENTENCE_POS[|NEWLINE_AFTER_EOT] writer whose ints are
 * captured instead of printed.
 */
typedef struct {
  uint32_t n_tok;
  int32_t *tok_rstart, *tok_rend;  /* rune offsets, text relative: pos[] pairs */
  uint32_t *tok_bstart, *tok_bend; /* byte offsets, document relative */
  uint32_t n_sent;                 /* ints in sent[], flat (token_writer.go:78,108) */
  int32_t *sent;
  uint32_t n_text;                 /* TextEnd calls */
  uint32_t *text_tok_end;          /* tokens seen when the i-th TextEnd fired */
  uint32_t *text_sent_end;         /* sent ints seen when it fired */
  uint32_t n_sent_events;          /* SentenceEnd calls */
  unsigned status;
  uint64_t steps;                  /* table lookups performed */
} orc_doc_result;

void orc_transduce_doc(const orc_model *m, const uint8_t *text, size_t n,
                       unsigned flags, orc_doc_result *out);
void orc_free_doc_result(orc_doc_result *r);

/*
 * Raw event replay list, in reference call order (what a Go shim would feed
 * to the caller's TokenWriter closures).
 * kind 0: Token(offset=a, buf=runes[buf_start .. end)), b=byte start of buf[0],
 *         c=byte start of buf[offset], d=byte end.
 * kind 1: SentenceEnd(a).   kind 2: TextEnd(a).
 */
typedef struct { uint32_t kind; int32_t a; uint32_t b, c, d; } orc_event;
orc_event *orc_transduce_events(const orc_model *m, const uint8_t *text,
                                size_t n, size_t *n_events, unsigned *status);

/*
 * Batch throughput leg for bench.py's cpu_baseline: walks every document with
 * a counting sink (tokens, sentence ends, text ends are tallied per document,
 * nothing is rendered), nthreads POSIX threads over contiguous doc ranges.
 * counts: 3 * n_docs uint32 (tok, sent_events, text_ends).
 */
void orc_count_batch(const orc_model *m, const uint8_t *text,
                     const uint64_t *doc_off, uint32_t n_docs, int nthreads,
                     uint32_t *counts);

void orc_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
