/*
 * datok_oracle.c -- CPU restatement of the Datok tokenizer hot path (C99).
 *
 * TEST INFRASTRUCTURE ONLY (see datok_oracle.h).  Parity status: PINNED by
 * the reference's golden strings; invalid UTF-8 is unpinned (Go doc spec).
 *
 * All file:line citations are relative to the upstream KorAP/Datok tree.
 */
#include "datok_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#define FIRSTBIT 0x80000000u  /* datok.go:43 */
#define SECONDBIT 0x40000000u /* datok.go:44 */
#define RESTBIT 0x3fffffffu   /* datok.go:45 */
#define EOT 4                 /* matrix.go:13 */
#define GO_WINDOW 1024        /* matrix.go:365, datok.go:812 */

struct orc_model {
  int kind;
  int epsilon, unknown, identity, final_, sigma_count;
  uint32_t state_count; /* matrix only */
  int sigma_ascii[256];
  /* sigma map[rune]int as a sorted array (last writer wins, like a Go map) */
  int n_sigma;
  uint32_t *sigma_rune;
  int *sigma_sym;
  /* matrix: (state_count+1)*sigma_count cells; DA: 2*array_len (base,check) */
  uint32_t *array;
  uint64_t array_len;
};

/* ------------------------------------------------------------------ utf8 */

/* Go unicode/utf8.DecodeRune (the decoder behind bufio.ReadRune,
 * matrix.go:392, datok.go:839, and behind the sigma parser, matrix.go:296):
 * invalid or short sequences yield (U+FFFD, width 1); overlongs, surrogates
 * and values above U+10FFFF are invalid. */
int orc_decode_rune(const uint8_t *p, size_t n, uint32_t *r) {
  if (n == 0) { *r = 0xFFFD; return 0; }
  uint32_t b0 = p[0];
  if (b0 < 0x80) { *r = b0; return 1; }
  if (b0 < 0xC2 || b0 > 0xF4) { *r = 0xFFFD; return 1; }
  if (b0 < 0xE0) {
    if (n < 2 || (p[1] & 0xC0) != 0x80) { *r = 0xFFFD; return 1; }
    *r = ((b0 & 0x1F) << 6) | (p[1] & 0x3F);
    return 2;
  }
  if (b0 < 0xF0) {
    uint32_t lo = 0x80, hi = 0xBF;
    if (b0 == 0xE0) lo = 0xA0;
    if (b0 == 0xED) hi = 0x9F;
    if (n < 3 || p[1] < lo || p[1] > hi || (p[2] & 0xC0) != 0x80) { *r = 0xFFFD; return 1; }
    *r = ((b0 & 0x0F) << 12) | ((uint32_t)(p[1] & 0x3F) << 6) | (p[2] & 0x3F);
    return 3;
  }
  {
    uint32_t lo = 0x80, hi = 0xBF;
    if (b0 == 0xF0) lo = 0x90;
    if (b0 == 0xF4) hi = 0x8F;
    if (n < 4 || p[1] < lo || p[1] > hi || (p[2] & 0xC0) != 0x80 || (p[3] & 0xC0) != 0x80) {
      *r = 0xFFFD; return 1;
    }
    *r = ((b0 & 0x07) << 18) | ((uint32_t)(p[1] & 0x3F) << 12) |
         ((uint32_t)(p[2] & 0x3F) << 6) | (p[3] & 0x3F);
    return 4;
  }
}

/* Go string(rune): invalid runes encode as U+FFFD. */
static int encode_rune(uint32_t r, uint8_t *o) {
  if (r > 0x10FFFF || (r >= 0xD800 && r <= 0xDFFF)) r = 0xFFFD;
  if (r < 0x80) { o[0] = (uint8_t)r; return 1; }
  if (r < 0x800) { o[0] = 0xC0 | (r >> 6); o[1] = 0x80 | (r & 0x3F); return 2; }
  if (r < 0x10000) {
    o[0] = 0xE0 | (r >> 12); o[1] = 0x80 | ((r >> 6) & 0x3F); o[2] = 0x80 | (r & 0x3F);
    return 3;
  }
  o[0] = 0xF0 | (r >> 18); o[1] = 0x80 | ((r >> 12) & 0x3F);
  o[2] = 0x80 | ((r >> 6) & 0x3F); o[3] = 0x80 | (r & 0x3F);
  return 4;
}

/* ------------------------------------------------------------- containers */

typedef struct { char *p; size_t n, cap; } sbuf;
static void sb_put(sbuf *s, const void *d, size_t k) {
  if (s->n + k + 1 > s->cap) {
    size_t c = s->cap ? s->cap * 2 : 256;
    while (c < s->n + k + 1) c *= 2;
    s->p = (char *)realloc(s->p, c);
    s->cap = c;
  }
  memcpy(s->p + s->n, d, k);
  s->n += k;
  s->p[s->n] = 0;
}
static void sb_byte(sbuf *s, char c) { sb_put(s, &c, 1); }
static void sb_int(sbuf *s, int v) { /* strconv.Itoa, token_writer.go:135-148 */
  char tmp[16];
  int k = snprintf(tmp, sizeof tmp, "%d", v);
  sb_put(s, tmp, (size_t)k);
}

typedef struct { int32_t *p; size_t n, cap; } ivec;
static void iv_push(ivec *v, int32_t x) {
  if (v->n == v->cap) {
    v->cap = v->cap ? v->cap * 2 : 64;
    v->p = (int32_t *)realloc(v->p, v->cap * sizeof(int32_t));
  }
  v->p[v->n++] = x;
}
typedef struct { uint32_t *p; size_t n, cap; } uvec;
static void uv_push(uvec *v, uint32_t x) {
  if (v->n == v->cap) {
    v->cap = v->cap ? v->cap * 2 : 64;
    v->p = (uint32_t *)realloc(v->p, v->cap * sizeof(uint32_t));
  }
  v->p[v->n++] = x;
}

/* ------------------------------------------------------------ model load */

static int sigma_cmp(const void *a, const void *b) {
  const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
  return (x > y) - (x < y);
}

/* Builds the rune->symbol map the way repeated Go map stores do
 * (matrix.go:295-303, datok.go:686-694): a later index overwrites. */
static void build_sigma(orc_model *m, const uint32_t *runes, const int *syms, int k) {
  uint64_t *tmp = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(k ? k : 1));
  for (int i = 0; i < k; i++) tmp[i] = ((uint64_t)runes[i] << 32) | (uint32_t)syms[i];
  qsort(tmp, (size_t)k, sizeof(uint64_t), sigma_cmp);
  m->sigma_rune = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(k ? k : 1));
  m->sigma_sym = (int *)malloc(sizeof(int) * (size_t)(k ? k : 1));
  int o = 0;
  for (int i = 0; i < k; i++) {
    uint32_t r = (uint32_t)(tmp[i] >> 32);
    int s = (int)(uint32_t)tmp[i];
    if (o > 0 && m->sigma_rune[o - 1] == r) m->sigma_sym[o - 1] = s; /* larger index wins */
    else { m->sigma_rune[o] = r; m->sigma_sym[o] = s; o++; }
  }
  m->n_sigma = o;
  free(tmp);
}

static uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static uint32_t rd32(const uint8_t *p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

/* Sigma block shared by matrix.go:288-303 and datok.go:679-694. */
static size_t parse_sigma(orc_model *m, const uint8_t *raw, size_t n, size_t off, int sigma_count) {
  uint32_t *runes = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(sigma_count ? sigma_count : 1));
  int *syms = (int *)malloc(sizeof(int) * (size_t)(sigma_count ? sigma_count : 1));
  int k = 0;
  for (int i = 0; i < 256; i++) m->sigma_ascii[i] = m->identity; /* identity != -1 always */
  for (int x = 0; x < sigma_count; x++) {
    uint32_t r;
    if (off >= n) continue; /* ReadRune error: entry skipped (err != nil) */
    int w = orc_decode_rune(raw + off, n - off, &r);
    off += (size_t)w;
    if (r != 0) {
      if (r < 256) m->sigma_ascii[r] = x;
      runes[k] = r; syms[k] = x; k++;
    }
  }
  build_sigma(m, runes, syms, k);
  free(runes); free(syms);
  return off;
}

/* matrix.go:235-337 ParseMatrix */
static orc_model *parse_matrix(const uint8_t *raw, size_t n) {
  if (n < 5 + 14) return NULL;
  const uint8_t *h = raw + 5;
  if (rd16(h) != 1) return NULL; /* VERSION, matrix.go:274-279 */
  orc_model *m = (orc_model *)calloc(1, sizeof *m);
  m->kind = ORC_KIND_MATRIX;
  m->epsilon = rd16(h + 2);
  m->unknown = rd16(h + 4);
  m->identity = rd16(h + 6);
  m->state_count = rd32(h + 8);
  m->sigma_count = rd16(h + 12);
  m->array_len = ((uint64_t)m->state_count + 1) * (uint64_t)m->sigma_count; /* :286 */
  size_t off = parse_sigma(m, raw, n, 19, m->sigma_count);
  if (off >= n || raw[off] != 'M') { orc_free_model(m); return NULL; } /* :305-315 */
  off++;
  if (n - off < m->array_len * 4) { orc_free_model(m); return NULL; } /* :327-330 */
  m->array = (uint32_t *)malloc((size_t)(m->array_len ? m->array_len : 1) * 4);
  for (uint64_t x = 0; x < m->array_len; x++) m->array[x] = rd32(raw + off + x * 4);
  return m;
}

/* datok.go:621-729 ParseDatok */
static orc_model *parse_datok(const uint8_t *raw, size_t n) {
  if (n < 5 + 16) return NULL;
  const uint8_t *h = raw + 5;
  if (rd16(h) != 1) return NULL;
  orc_model *m = (orc_model *)calloc(1, sizeof *m);
  m->kind = ORC_KIND_DA;
  m->epsilon = rd16(h + 2);
  m->unknown = rd16(h + 4);
  m->identity = rd16(h + 6);
  m->final_ = rd16(h + 8);
  m->sigma_count = rd16(h + 10);
  m->array_len = rd32(h + 12) / 2; /* "Legacy support", datok.go:674 */
  size_t off = parse_sigma(m, raw, n, 21, m->sigma_count);
  if (off >= n || raw[off] != 'T') { orc_free_model(m); return NULL; }
  off++;
  if (n - off < m->array_len * 8) { orc_free_model(m); return NULL; }
  m->array = (uint32_t *)malloc((size_t)(m->array_len ? m->array_len : 1) * 8);
  for (uint64_t x = 0; x < m->array_len * 2; x++) m->array[x] = rd32(raw + off + x * 4);
  return m;
}

orc_model *orc_parse(const uint8_t *raw, size_t n) {
  if (n < 5) return NULL;
  if (memcmp(raw, "MATOK", 5) == 0) return parse_matrix(raw, n); /* fomafile.go:476 */
  if (memcmp(raw, "DATOK", 5) == 0) return parse_datok(raw, n);  /* fomafile.go:478 */
  return NULL;
}

orc_model *orc_load_file(const char *path) {
  gzFile f = gzopen(path, "rb");
  if (!f) return NULL;
  size_t cap = 1 << 20, n = 0;
  uint8_t *raw = (uint8_t *)malloc(cap);
  for (;;) {
    if (n == cap) { cap *= 2; raw = (uint8_t *)realloc(raw, cap); }
    int k = gzread(f, raw + n, (unsigned)(cap - n));
    if (k < 0) { gzclose(f); free(raw); return NULL; }
    if (k == 0) break;
    n += (size_t)k;
  }
  /* gzip.NewReader fails on non-gzip input (fomafile.go:460); gzopen would
   * pass plain bytes through, so reject files lacking the gzip magic. */
  int direct = gzdirect(f);
  gzclose(f);
  orc_model *m = direct ? NULL : orc_parse(raw, n);
  free(raw);
  return m;
}

void orc_free_model(orc_model *m) {
  if (!m) return;
  free(m->sigma_rune); free(m->sigma_sym); free(m->array); free(m);
}

const char *orc_type(const orc_model *m) { return m->kind == ORC_KIND_MATRIX ? "MATOK" : "DATOK"; }

void orc_info(const orc_model *m, orc_model_info *o) {
  o->kind = m->kind; o->epsilon = m->epsilon; o->unknown = m->unknown;
  o->identity = m->identity; o->final_ = m->final_; o->sigma_count = m->sigma_count;
  o->state_count = m->kind == ORC_KIND_MATRIX ? m->state_count
                   : (m->array_len > 1 ? (m->array[3] & RESTBIT) : 0); /* datok.go:333-335 */
  o->array_len = m->array_len; o->n_sigma_runes = m->n_sigma;
}
const uint32_t *orc_array(const orc_model *m) { return m->array; }
const int *orc_sigma_ascii(const orc_model *m) { return m->sigma_ascii; }

/* `a, ok = mat.sigma[char]` (matrix.go:427, datok.go:865) */
int orc_sigma_lookup(const orc_model *m, uint32_t rune, int *ok) {
  int lo = 0, hi = m->n_sigma - 1;
  while (lo <= hi) {
    int mid = (lo + hi) >> 1;
    if (m->sigma_rune[mid] == rune) { *ok = 1; return m->sigma_sym[mid]; }
    if (m->sigma_rune[mid] < rune) lo = mid + 1; else hi = mid - 1;
  }
  *ok = 0;
  return 0;
}

/* ------------------------------------------------------------------ sinks */

/* The four closures of token_writer.go:27-33.  Token additionally receives
 * the byte start/width of every buffered rune so byte offsets can be
 * reported; the reference algorithm does not depend on them. */
typedef struct sink {
  void *ctx;
  void (*token)(void *ctx, int offset, const uint32_t *buf, int len,
                const uint32_t *bstart, const uint8_t *bwidth);
  void (*sentence_end)(void *ctx, int arg);
  void (*text_end)(void *ctx, int arg);
} sink;

/* ---- NewTokenWriter, token_writer.go:36-175 (rendering) ---- */
typedef struct {
  unsigned flags;
  int posC, sentB, init;
  ivec pos, sent;
  sbuf out;
  unsigned status;
} twriter;

static void tw_init(twriter *w, unsigned flags) {
  memset(w, 0, sizeof *w);
  w->flags = flags;
  w->sentB = 1; /* :40 */
  w->init = 1;  /* :42 */
}

static void tw_surface(twriter *w, int offset, const uint32_t *buf, int len) {
  uint8_t tmp[4];
  for (int i = offset; i < len; i++) sb_put(&w->out, tmp, (size_t)encode_rune(buf[i], tmp));
  sb_byte(&w->out, '\n');
}

static void tw_token(void *ctx, int offset, const uint32_t *buf, int len,
                     const uint32_t *bstart, const uint8_t *bwidth) {
  twriter *w = (twriter *)ctx;
  (void)bstart; (void)bwidth;
  if (offset > len && (w->flags & ORC_TOKENS)) w->status |= ORC_ST_BAD_OFFSET; /* string(buf[offset:]): slice bounds panic */
  if (w->flags & (ORC_TOKEN_POS | ORC_SENTENCE_POS)) { /* :49-88 */
    if (w->posC == 0 && (w->flags & ORC_NEWLINE_AFTER_EOT) && len > 0 && buf[0] == '\n' && !w->init)
      w->posC--; /* :66-68 */
    w->init = 0;
    w->posC += offset;
    iv_push(&w->pos, w->posC);
    if (w->sentB) { w->sentB = 0; iv_push(&w->sent, w->posC); }
    w->posC += len - offset;
    iv_push(&w->pos, w->posC);
    if (w->flags & ORC_TOKENS) tw_surface(w, offset, buf, len);
  } else if (w->flags & ORC_TOKENS) { /* :91-95 */
    tw_surface(w, offset, buf, len);
  }
}

static void tw_sentence_end(void *ctx, int arg) {
  twriter *w = (twriter *)ctx;
  (void)arg;
  if (w->flags & ORC_SENTENCE_POS) { /* :103-115 */
    if (w->pos.n == 0) w->status |= ORC_ST_EMPTY_TEXT; /* Go: index -1 panic */
    else iv_push(&w->sent, w->pos.p[w->pos.n - 1]);
    w->sentB = 1;
    if (w->flags & ORC_SENTENCES) sb_byte(&w->out, '\n');
  } else if (w->flags & ORC_SENTENCES) { /* :118-122 */
    sb_byte(&w->out, '\n');
  }
}

static void tw_text_end(void *ctx, int arg) {
  twriter *w = (twriter *)ctx;
  (void)arg;
  if (w->flags & (ORC_TOKEN_POS | ORC_SENTENCE_POS)) { /* :130-159 */
    if (w->flags & ORC_TOKEN_POS) {
      if (w->pos.n == 0) w->status |= ORC_ST_EMPTY_TEXT; /* Go: pos[0] panic */
      else {
        sb_int(&w->out, w->pos.p[0]);
        for (size_t i = 1; i < w->pos.n; i++) { sb_byte(&w->out, ' '); sb_int(&w->out, w->pos.p[i]); }
        sb_byte(&w->out, '\n');
      }
    }
    if (w->flags & ORC_SENTENCE_POS) {
      if (w->sent.n == 0) w->status |= ORC_ST_EMPTY_TEXT; /* Go: sent[0] panic */
      else {
        sb_int(&w->out, w->sent.p[0]);
        for (size_t i = 1; i < w->sent.n; i++) { sb_byte(&w->out, ' '); sb_int(&w->out, w->sent.p[i]); }
        sb_byte(&w->out, '\n');
      }
      w->sent.n = 0;
      w->sentB = 1;
    }
    w->posC = 0;
    w->pos.n = 0;
  } else { /* :162-166 */
    sb_byte(&w->out, '\n');
  }
}

/* ---- capturing writer: TOKEN_POS|SENTENCE_POS[|NEWLINE_AFTER_EOT] ints ---- */
typedef struct {
  unsigned flags;
  int posC, sentB, init;
  uint32_t text_tok0; /* tokens before the current text (pos = pos[:0]) */
  ivec rstart, rend, sent;
  uvec bstart, bend, text_tok_end, text_sent_end;
  uint32_t n_sent_events;
  unsigned status;
} capture;

static void cap_token(void *ctx, int offset, const uint32_t *buf, int len,
                      const uint32_t *bstart, const uint8_t *bwidth) {
  capture *c = (capture *)ctx;
  if (offset > len) c->status |= ORC_ST_BAD_OFFSET; /* Go: slice bounds panic where the surface is printed */
  if (c->posC == 0 && (c->flags & ORC_NEWLINE_AFTER_EOT) && len > 0 && buf[0] == '\n' && !c->init)
    c->posC--;
  c->init = 0;
  c->posC += offset;
  iv_push(&c->rstart, c->posC);
  if (c->sentB) { c->sentB = 0; iv_push(&c->sent, c->posC); }
  c->posC += len - offset;
  iv_push(&c->rend, c->posC);
  /* byte range of the surface buf[offset:len]; an empty surface (offset == len: the second Token
     call for a document like "...\n") is the empty range at the end of the buffer */
  {
    const uint32_t bend = len > 0 ? bstart[len - 1] + bwidth[len - 1] : 0u;
    uv_push(&c->bstart, offset < len ? bstart[offset] : bend);
    uv_push(&c->bend, bend);
  }
}
static void cap_sentence_end(void *ctx, int arg) {
  capture *c = (capture *)ctx;
  (void)arg;
  c->n_sent_events++;
  if (c->rend.n == c->text_tok0) c->status |= ORC_ST_EMPTY_TEXT;
  else iv_push(&c->sent, c->rend.p[c->rend.n - 1]);
  c->sentB = 1;
}
static void cap_text_end(void *ctx, int arg) {
  capture *c = (capture *)ctx;
  (void)arg;
  if (c->rend.n == c->text_tok0) c->status |= ORC_ST_EMPTY_TEXT;
  uv_push(&c->text_tok_end, (uint32_t)c->rend.n);
  uv_push(&c->text_sent_end, (uint32_t)c->sent.n);
  c->sentB = 1;
  c->posC = 0;
  c->text_tok0 = (uint32_t)c->rend.n;
}

/* ---- raw event list ---- */
typedef struct { orc_event *p; size_t n, cap; } evec;
static void ev_push(evec *v, orc_event e) {
  if (v->n == v->cap) {
    v->cap = v->cap ? v->cap * 2 : 64;
    v->p = (orc_event *)realloc(v->p, v->cap * sizeof(orc_event));
  }
  v->p[v->n++] = e;
}
static void ev_token(void *ctx, int offset, const uint32_t *buf, int len,
                     const uint32_t *bstart, const uint8_t *bwidth) {
  (void)buf;
  const uint32_t bend_ = len > 0 ? bstart[len - 1] + bwidth[len - 1] : 0u;
  orc_event e = {0, offset, len > 0 ? bstart[0] : 0u, offset < len ? bstart[offset] : bend_, bend_};
  ev_push((evec *)ctx, e);
}
static void ev_sentence_end(void *ctx, int arg) { orc_event e = {1, arg, 0, 0, 0}; ev_push((evec *)ctx, e); }
static void ev_text_end(void *ctx, int arg) { orc_event e = {2, arg, 0, 0, 0}; ev_push((evec *)ctx, e); }

/* ---- counting sink (cpu_baseline) ---- */
typedef struct { uint32_t tok, sent, text; } counter;
static void cnt_token(void *ctx, int offset, const uint32_t *buf, int len,
                      const uint32_t *bstart, const uint8_t *bwidth) {
  (void)offset; (void)buf; (void)len; (void)bstart; (void)bwidth;
  ((counter *)ctx)->tok++;
}
static void cnt_sentence_end(void *ctx, int arg) { (void)arg; ((counter *)ctx)->sent++; }
static void cnt_text_end(void *ctx, int arg) { (void)arg; ((counter *)ctx)->text++; }

/* ------------------------------------------------------------ rune window */

/* `buffer := make([]rune, 1024)` (matrix.go:365, datok.go:812) plus, per rune,
 * its byte start and width.  The reference panics when a 1025th rune would be
 * stored; the oracle flags ORC_ST_WINDOW_OVERFLOW and keeps going with a
 * larger window so the GPU path can be compared on such inputs too. */
typedef struct {
  uint32_t *rune, *bstart;
  uint8_t *bwidth;
  int cap;
} window;

static void win_init(window *b) {
  b->cap = 2048;
  b->rune = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)b->cap);
  b->bstart = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)b->cap);
  b->bwidth = (uint8_t *)malloc((size_t)b->cap);
}
static void win_free(window *b) { free(b->rune); free(b->bstart); free(b->bwidth); }
static void win_grow(window *b) {
  b->cap *= 2;
  b->rune = (uint32_t *)realloc(b->rune, sizeof(uint32_t) * (size_t)b->cap);
  b->bstart = (uint32_t *)realloc(b->bstart, sizeof(uint32_t) * (size_t)b->cap);
  b->bwidth = (uint8_t *)realloc(b->bwidth, (size_t)b->cap);
}
/* copy(buffer[0:], buffer[buffc:buffi]) (matrix.go:537,614) */
static void win_shift(window *b, int from, int to) {
  int k = to - from;
  if (k > 0 && from > 0) {
    memmove(b->rune, b->rune + from, sizeof(uint32_t) * (size_t)k);
    memmove(b->bstart, b->bstart + from, sizeof(uint32_t) * (size_t)k);
    memmove(b->bwidth, b->bwidth + from, (size_t)k);
  }
}

/* ------------------------------------------------------------ matrix walk */

/* matrix.go:348-698 MatrixTokenizer.TransduceTokenWriter */
static int walk_matrix(const orc_model *m, const uint8_t *in, size_t n, const sink *w,
                       unsigned *status, uint64_t *steps_out) {
  const uint32_t *arr = m->array;
  const uint64_t N = m->state_count, alen = m->array_len;
  const int eps = m->epsilon, unknown = m->unknown, identity = m->identity;
  int a = 0;
  uint32_t t0 = 0, t = 1; /* :351 initial state */
  int ok = 0, rewind_buffer;
  uint32_t epsilon_state = 0; /* :356 */
  int epsilon_offset = 0;
  int sentence_end = 0, text_end = 0; /* :360,363 */
  window buf;
  int bufft = 0, buffc = 0, buffi = 0; /* :366-368 */
  size_t rd = 0; /* bufio.Reader cursor */
  uint32_t ch;
  int eof = 0, eot = 0, newchar = 1;
  uint64_t steps = 0;
  int ret = 1;
  win_init(&buf);

  for (;;) { /* PARSECHARM re-entry via goto (:658,:667) */
    for (;;) { /* :384 */
      if (newchar) {
        if (buffc >= buffi) { /* :388 */
          if (eof) break;
          if (rd >= n) { eof = 1; break; } /* io.EOF :396-399 */
          uint32_t r;
          int wd = orc_decode_rune(in + rd, n - rd, &r);
          if (buffi >= GO_WINDOW) *status |= ORC_ST_WINDOW_OVERFLOW; /* :406 would panic */
          if (buffi >= buf.cap) win_grow(&buf);
          buf.rune[buffi] = r;
          buf.bstart[buffi] = (uint32_t)rd;
          buf.bwidth[buffi] = (uint8_t)wd;
          rd += (size_t)wd;
          buffi++;
        }
        ch = buf.rune[buffc]; /* :410 */
        eot = 0;
        if (ch < 256) { /* :421-426 */
          eot = ch == EOT;
          a = m->sigma_ascii[ch];
        } else { /* :427-435; `ok` keeps its value between these visits */
          a = orc_sigma_lookup(m, ch, &ok);
          if (!ok && identity != -1) a = identity; /* a map miss leaves a == 0 */
        }
        t0 = t; /* :437 */
        {
          uint64_t ei = (uint64_t)(eps - 1) * N + t0; /* :442 */
          if (ei < alen && arr[ei] != 0) {
            epsilon_state = t0;
            epsilon_offset = buffc;
          }
        }
      }

      if (a == 0) { /* :459 */
        t = 0;
      } else {
        uint64_t ix = (uint64_t)(a - 1) * N + t0; /* :463 */
        if (ix >= alen) { *status |= ORC_ST_BAD_MODEL; t = 0; }
        else t = arr[ix];
      }
      steps++;

      if (t == 0) { /* :472 */
        if (!ok && a == identity) { /* :478 */
          a = unknown;
        } else if (a != eps && epsilon_state != 0) { /* :487 */
          t0 = epsilon_state;
          epsilon_state = 0;
          buffc = epsilon_offset;
          a = eps;
        } else { /* :499 hard fail */
          if (buffc - bufft <= 0) {
            buffc++; /* :515-516 */
            if (buffc > buffi) { /* would hand out stale buffer runes */
              *status |= ORC_ST_BAD_MODEL;
              ret = 0;
              goto done;
            }
          }
          w->token(w->ctx, bufft, buf.rune, buffc, buf.bstart, buf.bwidth); /* :528 */
          sentence_end = 0;
          text_end = 0;
          win_shift(&buf, buffc, buffi); /* :537 */
          buffi -= buffc;
          epsilon_state = 0;
          buffc = 0;
          bufft = 0;
          a = eps;
          t = 1; /* :548 */
          newchar = 1;
          continue;
        }
        newchar = 0; /* :554 */
        eot = 0;
        continue;
      }

      rewind_buffer = 0; /* :560 */
      if (a == eps) { /* :563 */
        if (buffc - bufft > 0) {
          w->token(w->ctx, bufft, buf.rune, buffc, buf.bstart, buf.bwidth); /* :569 */
          rewind_buffer = 1;
          sentence_end = 0;
          text_end = 0;
        } else {
          sentence_end = 1;
          w->sentence_end(w->ctx, buffc); /* :575 */
        }
      } else { /* :579 */
        buffc++;
        if (buffc - bufft == 1 && (t & FIRSTBIT) != 0) bufft++; /* :584-588 */
      }

      if (eot) { /* :593 */
        eot = 0;
        if (!sentence_end) {
          sentence_end = 1;
          w->sentence_end(w->ctx, buffc);
        }
        text_end = 1;
        w->text_end(w->ctx, buffc);
        rewind_buffer = 1;
      }

      if (rewind_buffer) { /* :608 */
        win_shift(&buf, buffc, buffi);
        buffi -= buffc;
        epsilon_offset = 0;
        epsilon_state = 0;
        buffc = 0;
        bufft = 0;
      }

      t &= ~FIRSTBIT; /* :629 */
      newchar = 1;
    }

    if (!eof) { ret = 0; goto done; } /* :638-644 */

    t0 = t; /* :651 */
    {
      uint64_t ei = (uint64_t)(eps - 1) * N + t0;
      t = ei < alen ? arr[ei] : 0;
    }
    a = eps;
    newchar = 0;
    if (t != 0) continue; /* :656-658 */
    if (epsilon_state != 0) { /* :660-667 */
      t0 = epsilon_state;
      epsilon_state = 0;
      buffc = epsilon_offset;
      continue;
    }
    break;
  }

  if (buffc - bufft > 0) { /* :671 */
    w->token(w->ctx, bufft, buf.rune, buffc, buf.bstart, buf.bwidth);
    sentence_end = 0;
    text_end = 0;
  }
  if (!sentence_end) w->sentence_end(w->ctx, buffc); /* :683 */
  if (!text_end) w->text_end(w->ctx, buffc);         /* :690 */

done:
  win_free(&buf);
  if (steps_out) *steps_out = steps;
  return ret;
}

/* ---------------------------------------------------------------- DA walk */

#define DA_BASE(i) (arr[2 * (uint64_t)(i)])
#define DA_CHECK(i) (arr[2 * (uint64_t)(i) + 1])

/* datok.go:781-1135 DaTokenizer.TransduceTokenWriter */
static int walk_da(const orc_model *m, const uint8_t *in, size_t n, const sink *w,
                   unsigned *status, uint64_t *steps_out) {
  const uint32_t *arr = m->array;
  const uint64_t alen = m->array_len;
  const int eps = m->epsilon, unknown = m->unknown, identity = m->identity;
  const uint32_t size = alen > 1 ? (DA_CHECK(1) & RESTBIT) : 0; /* datok.go:333-335 */
  int a = 0;
  uint32_t t0 = 0, t = 1;
  int ok = 0, rewind_buffer;
  uint32_t epsilon_state = 0;
  int epsilon_offset = 0;
  int sentence_end = 0, text_end = 0;
  window buf;
  int bufft = 0, buffc = 0, buffi = 0;
  size_t rd = 0;
  uint32_t ch;
  int eof = 0, eot = 0, newchar = 1;
  uint64_t steps = 0;
  int ret = 1;
  win_init(&buf);

  for (;;) { /* PARSECHAR re-entry (:1093,:1102) */
    for (;;) { /* :831 */
      uint32_t ta_base, ta_check;
      if (newchar) {
        if (buffc >= buffi) { /* :835 */
          if (eof) break;
          if (rd >= n) { eof = 1; break; } /* any read error ends the input :842-845 */
          uint32_t r;
          int wd = orc_decode_rune(in + rd, n - rd, &r);
          if (buffi >= GO_WINDOW) *status |= ORC_ST_WINDOW_OVERFLOW;
          if (buffi >= buf.cap) win_grow(&buf);
          buf.rune[buffi] = r;
          buf.bstart[buffi] = (uint32_t)rd;
          buf.bwidth[buffi] = (uint8_t)wd;
          rd += (size_t)wd;
          buffi++;
        }
        ch = buf.rune[buffc]; /* :850 */
        eot = 0;
        if (ch < 256) { /* :861-863 */
          eot = ch == EOT;
          a = m->sigma_ascii[ch];
        } else { /* :865-870 */
          a = orc_sigma_lookup(m, ch, &ok);
          if (!ok && identity != -1) a = identity; /* a map miss leaves a == 0 */
        }
        t0 = t; /* :873 */
        if (t0 >= alen) { *status |= ORC_ST_BAD_MODEL; ret = 0; goto done; }
        {
          uint64_t ei = (uint64_t)(DA_BASE(t0) & RESTBIT) + (uint32_t)eps; /* :876 */
          if (ei < alen && (DA_CHECK(ei) & RESTBIT) == t0) {
            epsilon_state = t0;
            epsilon_offset = buffc;
          }
        }
      }

      if (t0 >= alen) { *status |= ORC_ST_BAD_MODEL; ret = 0; goto done; }
      t = (DA_BASE(t0) & RESTBIT) + (uint32_t)a; /* :889 */
      steps++;
      if (t >= alen) { /* Go: index panic at :890 */
        *status |= ORC_ST_BAD_MODEL;
        ta_base = 0; ta_check = 0;
      } else {
        ta_base = DA_BASE(t);
        ta_check = DA_CHECK(t);
      }

      if (t > size || (ta_check & RESTBIT) != t0) { /* :901 */
        if (!ok && a == identity) { /* :907 */
          a = unknown;
        } else if (a != eps && epsilon_state != 0) { /* :916 */
          t0 = epsilon_state;
          epsilon_state = 0;
          buffc = epsilon_offset;
          a = eps;
        } else { /* :928 hard fail */
          if (buffc - bufft <= 0) {
            buffc++;
            if (buffc > buffi) { *status |= ORC_ST_BAD_MODEL; ret = 0; goto done; }
          }
          w->token(w->ctx, bufft, buf.rune, buffc, buf.bstart, buf.bwidth); /* :953 */
          sentence_end = 0;
          text_end = 0;
          win_shift(&buf, buffc, buffi);
          buffi -= buffc;
          epsilon_state = 0;
          buffc = 0;
          bufft = 0;
          a = eps;
          t = 1; /* :973 */
          newchar = 1;
          continue;
        }
        newchar = 0; /* :979 */
        eot = 0;
        continue;
      }

      rewind_buffer = 0; /* :985 */
      if (a != eps) { /* :988 */
        buffc++;
        if (buffc - bufft == 1 && (ta_check & FIRSTBIT) != 0) bufft++; /* :994-998 */
      } else {
        if (buffc - bufft > 0) { /* :1005 */
          w->token(w->ctx, bufft, buf.rune, buffc, buf.bstart, buf.bwidth);
          rewind_buffer = 1;
          sentence_end = 0;
          text_end = 0;
        } else {
          sentence_end = 1;
          w->sentence_end(w->ctx, 0); /* :1015 */
        }
      }

      if (eot) { /* :1019 -- no buffer rewind here, unlike matrix.go:601 */
        eot = 0;
        if (!sentence_end) {
          sentence_end = 1;
          w->sentence_end(w->ctx, buffc); /* :1023 */
        }
        text_end = 1;
        w->text_end(w->ctx, 0); /* :1026 */
      }

      if (rewind_buffer) { /* :1033 */
        win_shift(&buf, buffc, buffi);
        buffi -= buffc;
        epsilon_offset = 0;
        epsilon_state = 0;
        buffc = 0;
        bufft = 0;
      }

      if (ta_base & FIRSTBIT) t = ta_base & RESTBIT; /* :1056-1058 representative */
      newchar = 1;
    }

    if (!eof) { ret = 0; goto done; } /* :1073 */

    t0 = t; /* :1086 */
    if (t0 >= alen) { *status |= ORC_ST_BAD_MODEL; ret = 0; goto done; }
    t = (DA_BASE(t0) & RESTBIT) + (uint32_t)eps;
    a = eps;
    newchar = 0;
    if (t < alen && (DA_CHECK(t) & RESTBIT) == t0) continue; /* :1091-1093 */
    if (epsilon_state != 0) { /* :1095-1102 */
      t0 = epsilon_state;
      epsilon_state = 0;
      buffc = epsilon_offset;
      continue;
    }
    break;
  }

  if (buffc - bufft > 0) { /* :1106 */
    w->token(w->ctx, bufft, buf.rune, buffc, buf.bstart, buf.bwidth);
    sentence_end = 0;
    text_end = 0;
  }
  if (!sentence_end) w->sentence_end(w->ctx, 0); /* :1118 */
  if (!text_end) w->text_end(w->ctx, 0);         /* :1126 */

done:
  win_free(&buf);
  if (steps_out) *steps_out = steps;
  return ret;
}

static int walk(const orc_model *m, const uint8_t *in, size_t n, const sink *w,
                unsigned *status, uint64_t *steps) {
  return m->kind == ORC_KIND_MATRIX ? walk_matrix(m, in, n, w, status, steps)
                                    : walk_da(m, in, n, w, status, steps);
}

/* -------------------------------------------------------------- public API */

char *orc_transduce_string(const orc_model *m, const uint8_t *text, size_t n,
                           unsigned flags, size_t *out_len, unsigned *status) {
  twriter tw;
  tw_init(&tw, flags);
  sink s = {&tw, tw_token, tw_sentence_end, tw_text_end};
  unsigned st = 0;
  walk(m, text, n, &s, &st, NULL);
  st |= tw.status;
  if (status) *status = st;
  if (!tw.out.p) sb_put(&tw.out, "", 0);
  if (out_len) *out_len = tw.out.n;
  free(tw.pos.p); free(tw.sent.p);
  return tw.out.p;
}

void orc_transduce_doc(const orc_model *m, const uint8_t *text, size_t n,
                       unsigned flags, orc_doc_result *out) {
  capture c;
  memset(&c, 0, sizeof c);
  c.flags = flags;
  c.sentB = 1;
  c.init = 1;
  sink s = {&c, cap_token, cap_sentence_end, cap_text_end};
  unsigned st = 0;
  uint64_t steps = 0;
  walk(m, text, n, &s, &st, &steps);
  memset(out, 0, sizeof *out);
  out->n_tok = (uint32_t)c.rstart.n;
  out->tok_rstart = c.rstart.p; out->tok_rend = c.rend.p;
  out->tok_bstart = c.bstart.p; out->tok_bend = c.bend.p;
  out->n_sent = (uint32_t)c.sent.n; out->sent = c.sent.p;
  out->n_text = (uint32_t)c.text_tok_end.n;
  out->text_tok_end = c.text_tok_end.p; out->text_sent_end = c.text_sent_end.p;
  out->n_sent_events = c.n_sent_events;
  out->status = st | c.status;
  out->steps = steps;
}

void orc_free_doc_result(orc_doc_result *r) {
  free(r->tok_rstart); free(r->tok_rend); free(r->tok_bstart); free(r->tok_bend);
  free(r->sent); free(r->text_tok_end); free(r->text_sent_end);
  memset(r, 0, sizeof *r);
}

orc_event *orc_transduce_events(const orc_model *m, const uint8_t *text, size_t n,
                                size_t *n_events, unsigned *status) {
  evec v = {0, 0, 0};
  sink s = {&v, ev_token, ev_sentence_end, ev_text_end};
  unsigned st = 0;
  walk(m, text, n, &s, &st, NULL);
  if (status) *status = st;
  *n_events = v.n;
  if (!v.p) v.p = (orc_event *)malloc(sizeof(orc_event));
  return v.p;
}

typedef struct {
  const orc_model *m;
  const uint8_t *text;
  const uint64_t *doc_off;
  uint32_t lo, hi;
  uint32_t *counts;
} batch_job;

static void *batch_thread(void *arg) {
  batch_job *j = (batch_job *)arg;
  for (uint32_t d = j->lo; d < j->hi; d++) {
    counter c = {0, 0, 0};
    sink s = {&c, cnt_token, cnt_sentence_end, cnt_text_end};
    unsigned st = 0;
    walk(j->m, j->text + j->doc_off[d], (size_t)(j->doc_off[d + 1] - j->doc_off[d]), &s, &st, NULL);
    j->counts[3 * (size_t)d] = c.tok;
    j->counts[3 * (size_t)d + 1] = c.sent;
    j->counts[3 * (size_t)d + 2] = c.text;
  }
  return NULL;
}

void orc_count_batch(const orc_model *m, const uint8_t *text, const uint64_t *doc_off,
                     uint32_t n_docs, int nthreads, uint32_t *counts) {
  if (nthreads < 1) nthreads = 1;
  if ((uint32_t)nthreads > n_docs) nthreads = n_docs ? (int)n_docs : 1;
  batch_job *jobs = (batch_job *)calloc((size_t)nthreads, sizeof *jobs);
  pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof *th);
  /* contiguous doc ranges balanced by bytes */
  uint64_t total = doc_off[n_docs] - doc_off[0];
  uint32_t d = 0;
  for (int i = 0; i < nthreads; i++) {
    uint64_t target = doc_off[0] + total * (uint64_t)(i + 1) / (uint64_t)nthreads;
    uint32_t lo = d;
    while (d < n_docs && (doc_off[d + 1] <= target || i == nthreads - 1)) d++;
    jobs[i].m = m; jobs[i].text = text; jobs[i].doc_off = doc_off;
    jobs[i].lo = lo; jobs[i].hi = d; jobs[i].counts = counts;
  }
  jobs[nthreads - 1].hi = n_docs;
  if (nthreads == 1) {
    batch_thread(&jobs[0]);
  } else {
    for (int i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, batch_thread, &jobs[i]);
    for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
  }
  free(jobs); free(th);
}

void orc_free(void *p) { free(p); }

/* ------------------------------------------------------------ foma -> matrix */

/* fomafile.go:77-450 ParseFoma + matrix.go:30-99 Automaton.ToMatrix, restated: a Foma
 * text net becomes the matrix tokenizer `foma.ToMatrix()` would build.  Used for the
 * reference's tiny-FST tests (matrix_test.go:25-105,182-206, datok_test.go:57-230). */
typedef struct { int in_sym, end, nontoken, used; } fedge;

static int str_runecount(const char *s, size_t n) {
  int k = 0; size_t i = 0; uint32_t r;
  while (i < n) { i += (size_t)orc_decode_rune((const uint8_t *)s + i, n - i, &r); k++; }
  return k;
}

orc_model *orc_parse_foma(const uint8_t *raw, size_t n) {
  int eps = -1, unk = -1, idt = -1, fin = -1, tokend = -1, sigma_count = 0, state_count = -1;
  uint32_t *sig_rune = NULL; char *sig_is_mcs = NULL; int sig_cap = 0;
  /* transitions[state][sym] as a dense table grown on demand */
  fedge *tr = NULL; int tr_syms = 0;
  int mode = 0, state = 0, in_sym = 0, out_sym = 0, end = 0, final = 0;
  size_t pos = 0;
  orc_model *m = NULL;
#define SIG_ENSURE(k) do { if ((k) >= sig_cap) { int nc = (k) + 64; \
    sig_rune = (uint32_t *)realloc(sig_rune, sizeof(uint32_t) * (size_t)nc); \
    sig_is_mcs = (char *)realloc(sig_is_mcs, (size_t)nc); \
    for (int q = sig_cap; q < nc; q++) { sig_rune[q] = 0; sig_is_mcs[q] = 0; } sig_cap = nc; } } while (0)
  while (pos < n) {
    /* r.ReadString('\n'): a last line without newline is dropped (fomafile.go:101-108) */
    const uint8_t *nl = (const uint8_t *)memchr(raw + pos, '\n', n - pos);
    if (!nl) break;
    const char *line = (const char *)raw + pos;
    size_t len = (size_t)(nl - (raw + pos)) + 1; /* includes '\n' */
    pos += len;
    if (len >= 2 && line[0] == '#' && line[1] == '#') { /* :111-135 */
      if (!strncmp(line, "##props##", 9)) mode = 1;
      else if (!strncmp(line, "##states##", 10)) { mode = 3; sigma_count++; fin = sigma_count; }
      else if (!strncmp(line, "##sigma##", 9)) mode = 2;
      else if (!strncmp(line, "##end##", 7)) mode = 4;
      else if (strncmp(line, "##foma-net", 10)) break;
      continue;
    }
    if (mode == 1) { /* props :140-187 */
      int f[13]; int nf = 0; const char *p = line;
      char fields[13][32];
      while (nf < 13) {
        const char *sp = (const char *)memchr(p, ' ', (size_t)(line + len - p));
        size_t l = sp ? (size_t)(sp - p) : (size_t)(line + len - p);
        if (l > 31) l = 31;
        memcpy(fields[nf], p, l); fields[nf][l] = 0; nf++;
        if (!sp) break;
        p = sp + 1;
      }
      (void)f;
      if (nf < 10 || strcmp(fields[6], "1") || strcmp(fields[9], "1")) goto fail; /* deterministic, eps free */
      state_count = atoi(fields[2]);
      continue;
    }
    if (mode == 2) { /* sigma :372-444 */
      const char *body = line; size_t bl = len - 1; /* line[0:len-1] */
      const char *sp = (const char *)memchr(body, ' ', bl);
      if (!sp) goto fail;
      int number = atoi(body) + 1;
      sigma_count = number;
      const char *sym = sp + 1; size_t sl = bl - (size_t)(sym - body);
      int rc = str_runecount(sym, sl);
      uint32_t symbol;
      SIG_ENSURE(number);
      if (rc == 1) {
        orc_decode_rune((const uint8_t *)sym, sl, &symbol);
      } else if (rc > 1) {
        if (sl == 18 && !memcmp(sym, "@_EPSILON_SYMBOL_@", 18)) eps = number;
        else if (sl == 18 && !memcmp(sym, "@_UNKNOWN_SYMBOL_@", 18)) unk = number;
        else if (sl == 19 && !memcmp(sym, "@_IDENTITY_SYMBOL_@", 19)) idt = number;
        else if (sl == 16 && !memcmp(sym, "@_TOKEN_SYMBOL_@", 16)) tokend = number;
        else if (sl == 15 && !memcmp(sym, "@_TOKEN_BOUND_@", 15)) tokend = number;
        else sig_is_mcs[number] = 1;
        continue;
      } else { /* probably the newline symbol: its second half is the next line */
        const uint8_t *nl2 = pos < n ? (const uint8_t *)memchr(raw + pos, '\n', n - pos) : NULL;
        if (!nl2) goto fail;
        size_t l2 = (size_t)(nl2 - (raw + pos)) + 1;
        pos += l2;
        if (l2 != 1) { sig_is_mcs[number] = 1; continue; }
        symbol = '\n';
      }
      sig_rune[number] = symbol;
      continue;
    }
    if (mode == 3) { /* states :189-369 */
      int e[5], ne = 0; const char *p = line; const char *lim = line + len - 1;
      if (state_count < 0) goto fail;
      while (ne < 5 && p <= lim) {
        const char *sp = (const char *)memchr(p, ' ', (size_t)(lim - p));
        e[ne++] = atoi(p);
        if (!sp) break;
        p = sp + 1;
      }
      if (ne == 0 || e[0] == -1) continue;
      if (!tr) {
        tr_syms = sigma_count + 2;
        tr = (fedge *)calloc((size_t)(state_count + 2) * (size_t)tr_syms, sizeof(fedge));
      }
#define TR(s, a) tr[(size_t)(s) * (size_t)tr_syms + (size_t)(a)]
      if (ne == 5) { state = e[0]; in_sym = e[1]; out_sym = e[2]; end = e[3]; final = e[4]; }
      else if (ne == 4) {
        if (e[1] == -1) { /* final state without outgoing edges */
          state = e[0]; final = e[3];
          if (final == 1 && state + 1 <= state_count) { TR(state + 1, fin).used = 1; TR(state + 1, fin).end = 0; }
          continue;
        }
        state = e[0]; in_sym = e[1]; end = e[2]; final = e[3]; out_sym = in_sym;
      } else if (ne == 3) { in_sym = e[0]; out_sym = e[1]; end = e[2]; }
      else if (ne == 2) { in_sym = e[0]; end = e[1]; out_sym = in_sym; }
      else continue;
      {
        int is = in_sym + 1, os = out_sym + 1, nontoken = 0; /* :263-264 */
        if (state + 1 > state_count) goto fail;
        if (is != os) {
          if (os == tokend && is == eps) { /* tokenend arc, stored under epsilon */ }
          else if (os == eps) nontoken = 1;
          else goto fail; /* unsupported transition */
        } else if (is == tokend) continue;
        else if (is == eps) goto fail;
        else if (is >= 0 && is < sig_cap && sig_is_mcs[is]) continue;
        if (is >= 0 && is < tr_syms) {
          TR(state + 1, is).used = 1; TR(state + 1, is).end = end + 1; TR(state + 1, is).nontoken = nontoken;
        }
        if (final == 1) { TR(state + 1, fin).used = 1; TR(state + 1, fin).end = 0; TR(state + 1, fin).nontoken = 0; }
      }
      continue;
    }
  }
  if (state_count < 0 || eps < 1 || !tr) goto fail;
  {
    /* ToMatrix, matrix.go:30-99 */
    int max = idt != -1 ? idt : 0;
    for (int k = 1; k <= sigma_count && k < sig_cap; k++) if (sig_rune[k] && k > max) max = k;
    const uint64_t N = (uint64_t)state_count, S = (uint64_t)max + 1;
    m = (orc_model *)calloc(1, sizeof *m);
    m->kind = ORC_KIND_MATRIX;
    m->epsilon = eps; m->unknown = unk; m->identity = idt; m->state_count = (uint32_t)N; m->sigma_count = (int)S;
    m->array_len = (N + 1) * S;
    m->array = (uint32_t *)calloc((size_t)m->array_len, 4);
    for (int i = 0; i < 256; i++) m->sigma_ascii[i] = idt != -1 ? idt : 0; /* :43-48: zero-valued without identity */
    {
      uint32_t *runes = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(sigma_count + 1));
      int *syms = (int *)malloc(sizeof(int) * (size_t)(sigma_count + 1));
      int k2 = 0;
      for (int k = 1; k <= sigma_count && k < sig_cap; k++)
        if (sig_rune[k]) { if (sig_rune[k] < 256) m->sigma_ascii[sig_rune[k]] = k; runes[k2] = sig_rune[k]; syms[k2] = k; k2++; }
      build_sigma(m, runes, syms, k2);
      free(runes); free(syms);
    }
    /* transitions reachable from state 1 only (:76-96) */
    char *seen = (char *)calloc((size_t)state_count + 2, 1);
    int *stack = (int *)malloc(sizeof(int) * (size_t)(state_count + 2));
    int sp = 0;
    stack[sp++] = 1; seen[1] = 1;
    while (sp > 0) {
      int s0 = stack[--sp];
      for (int a = 1; a < tr_syms; a++) {
        fedge *e2 = &TR(s0, a);
        if (!e2->used) continue;
        if ((uint64_t)(a - 1) * N + (uint64_t)s0 < m->array_len)
          m->array[(uint64_t)(a - 1) * N + (uint64_t)s0] = (uint32_t)e2->end | (e2->nontoken ? FIRSTBIT : 0u);
        if (e2->end >= 1 && e2->end <= state_count && !seen[e2->end]) { seen[e2->end] = 1; stack[sp++] = e2->end; }
      }
    }
    free(seen); free(stack);
  }
  free(tr); free(sig_rune); free(sig_is_mcs);
  return m;
fail:
  free(tr); free(sig_rune); free(sig_is_mcs);
  return NULL;
#undef TR
#undef SIG_ENSURE
}

/* fomafile.go:56-75 LoadFomaFile + ToMatrix */
orc_model *orc_load_foma_file(const char *path) {
  gzFile f = gzopen(path, "rb");
  if (!f) return NULL;
  size_t cap = 1 << 20, n = 0;
  uint8_t *raw = (uint8_t *)malloc(cap);
  for (;;) {
    if (n == cap) { cap *= 2; raw = (uint8_t *)realloc(raw, cap); }
    int k = gzread(f, raw + n, (unsigned)(cap - n));
    if (k < 0) { gzclose(f); free(raw); return NULL; }
    if (k == 0) break;
    n += (size_t)k;
  }
  int direct = gzdirect(f);
  gzclose(f);
  orc_model *m = direct ? NULL : orc_parse_foma(raw, n);
  free(raw);
  return m;
}
