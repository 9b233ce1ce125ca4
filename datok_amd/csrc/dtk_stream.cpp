// dtk_stream.cpp -- Tokenizer.Transduce / TransduceTokenWriter with a stock NewTokenWriter for
// ONE stream.  dtk_transduce renders on the device (dtk_render.hip); dtk_transduce_replay goes
// through the C++ host mirror (include/datok.hpp): the event bytes are replayed into the
// writer's closures, which is the path a custom TokenWriter takes.
#include <cstdlib>
#include <cstring>
#include <sstream>

#include "../../include/datok.hpp"

namespace {
constexpr uint32_t kReplayFields = DTK_R_EVENTS | DTK_R_TOK_BYTE | DTK_R_CSR | DTK_R_STATUS;
// One batch per calling thread, kept between calls (creating one costs a dozen device allocations and
// a stream: milliseconds, far more than tokenizing a short string) and replaced when an input does
// not fit.  dtk_transduce_release() frees it; so does the end of the thread.
struct BatchCache {
  dtk_batch *b = nullptr;
  uint64_t cap = 0;
  ~BatchCache() { dtk_batch_free(b); }
  int get(uint64_t n, dtk_batch **out) {
    if (!b || n > cap) {
      dtk_batch_free(b);
      b = nullptr;
      cap = n < (64u << 10) ? (64u << 10) : n + n / 4;
      int rc = dtk_batch_create(cap, 1, &b);
      if (rc != DTK_OK) { b = nullptr; cap = 0; return rc; }
    }
    *out = b;
    return DTK_OK;
  }
  void drop() { dtk_batch_free(b); b = nullptr; cap = 0; }
};
thread_local BatchCache g_cache;

// set_input + run on the cached batch; a batch made for another device is replaced once
// fields != 0: these result arrays leave for the host inside the run (DTK_R_EAGER: one short stream needs no copy
// engine set up seven times behind it)
int run_cached(const dtk_model *m, const uint8_t *text, size_t n, uint32_t flags, dtk_batch **bp, uint32_t fields = 0) {
  const uint64_t off[2] = {0, (uint64_t)n};
  for (int attempt = 0; attempt < 2; attempt++) {
    int rc = g_cache.get(n ? n : 1, bp);
    if (rc != DTK_OK) return rc;
    if ((rc = dtk_batch_set_input(*bp, text, off, 1)) != DTK_OK) return rc;
    if ((rc = dtk_batch_set_result_fields(*bp, fields ? (fields | DTK_R_EAGER) : (uint32_t)DTK_R_ALL)) != DTK_OK) return rc;
    rc = dtk_batch_run(m, *bp, flags);
    if (rc == DTK_E_ARG && attempt == 0) { g_cache.drop(); continue; }  // the model lives on another device
    return rc;
  }
  return DTK_E_ARG;
}
}  // namespace

extern "C" void dtk_transduce_release(void) { g_cache.drop(); }

// The walk of ONE stream on the calling thread's cached batch; the host view stays valid until this
// thread's next dtk_transduce* call.  What a TransduceTokenWriter with a custom writer replays from.
extern "C" int dtk_transduce_result(const dtk_model *m, const uint8_t *text, size_t n, uint32_t flags,
                                    dtk_result_view *view) {
  if (!m || !view || (n && !text)) return DTK_E_ARG;
  dtk_batch *b = nullptr;
  // what a closure replay reads (include/datok.hpp replay_view, datok_amd/host.py): the events, the tokens' byte
  // ranges, the row offsets and the status -- the rune offset arrays stay on the device
  int rc = run_cached(m, text, n, flags & DTK_NEWLINE_AFTER_EOT, &b, kReplayFields);
  if (rc != DTK_OK) return rc;
  return dtk_batch_result_host(b, view);
}

extern "C" int dtk_transduce(const dtk_model *m, const uint8_t *text, size_t n, uint32_t bits,
                             char **out, size_t *out_len, uint32_t *status) {
  if (!m || !out || (n && !text)) return DTK_E_ARG;
  *out = nullptr;
  dtk_batch *b = nullptr;
  dtk_render_view v;
  int rc;
  if ((rc = run_cached(m, text, n, bits & DTK_NEWLINE_AFTER_EOT, &b)) != DTK_OK ||
      (rc = dtk_batch_render_host(b, bits, &v)) != DTK_OK)
    return rc;
  uint32_t st = 0;
  if ((rc = dtk_batch_status_host(b, &st, 1)) != DTK_OK) return rc;
  if (status) {
    // an empty text only breaks the position modes (token_writer.go:108,135,145)
    *status = st;
    if (!(bits & (DTK_TOKEN_POS | DTK_SENTENCE_POS))) *status &= ~(uint32_t)DTK_ST_EMPTY_TEXT;
  }
  char *p = (char *)malloc((size_t)v.total + 1);
  if (!p) return DTK_E_NOMEM;
  memcpy(p, v.bytes, (size_t)v.total);
  p[v.total] = 0;
  *out = p;
  if (out_len) *out_len = (size_t)v.total;
  return DTK_OK;
}

extern "C" int dtk_transduce_replay(const dtk_model *m, const uint8_t *text, size_t n, uint32_t bits,
                             char **out, size_t *out_len, uint32_t *status) {
  if (!m || !out || (n && !text)) return DTK_E_ARG;
  *out = nullptr;
  dtk_batch *b = nullptr;
  dtk_result_view v;
  int rc;
  if ((rc = run_cached(m, text, n, bits & DTK_NEWLINE_AFTER_EOT, &b, kReplayFields)) != DTK_OK ||
      (rc = dtk_batch_result_host(b, &v)) != DTK_OK)
    return rc;
  std::ostringstream os;
  auto tw = datok::NewTokenWriter(os, (datok::Bits)bits);
  datok::detail::replay_view(std::strcmp(dtk_model_type(m), "MATOK") == 0, text, n, v, *tw);
  tw->Flush();
  if (status) {
    // an empty text only breaks the position modes (token_writer.go:108,135,145)
    *status = v.status[0];
    if (!(bits & (DTK_TOKEN_POS | DTK_SENTENCE_POS))) *status &= ~(uint32_t)DTK_ST_EMPTY_TEXT;
  }
  const std::string s = os.str();
  char *p = (char *)malloc(s.size() + 1);
  if (!p) return DTK_E_NOMEM;
  memcpy(p, s.data(), s.size());
  p[s.size()] = 0;
  *out = p;
  if (out_len) *out_len = s.size();
  return DTK_OK;
}
