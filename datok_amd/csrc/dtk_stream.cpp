// dtk_stream.cpp -- Tokenizer.Transduce / TransduceTokenWriter with a stock NewTokenWriter for
// ONE stream.  dtk_transduce renders on the device (dtk_render.hip); dtk_transduce_replay goes
// through the C++ host mirror (include/datok.hpp): the event bytes are replayed into the
// writer's closures, which is the path a custom TokenWriter takes.
#include <cstdlib>
#include <cstring>
#include <sstream>

#include "../../include/datok.hpp"

namespace {
// Borrowing view: the model stays owned by the caller.
struct Borrowed {
  datok::GpuTokenizer tok;
  explicit Borrowed(const dtk_model *m) : tok(const_cast<dtk_model *>(m)) {}
};
}  // namespace

extern "C" int dtk_transduce(const dtk_model *m, const uint8_t *text, size_t n, uint32_t bits,
                             char **out, size_t *out_len, uint32_t *status) {
  if (!m || !out || (n && !text)) return DTK_E_ARG;
  *out = nullptr;
  dtk_batch *b = nullptr;
  int rc = dtk_batch_create(n ? n : 1, 1, &b);
  if (rc != DTK_OK) return rc;
  const uint64_t off[2] = {0, (uint64_t)n};
  dtk_render_view v;
  dtk_result_view r;
  if ((rc = dtk_batch_set_input(b, text, off, 1)) != DTK_OK ||
      (rc = dtk_batch_run(m, b, bits & DTK_NEWLINE_AFTER_EOT)) != DTK_OK ||
      (rc = dtk_batch_render_host(b, bits, &v)) != DTK_OK ||
      (rc = dtk_batch_result_device(b, &r)) != DTK_OK) {
    dtk_batch_free(b);
    return rc;
  }
  uint32_t st = 0;
  rc = dtk_batch_status_host(b, &st, 1);
  if (rc != DTK_OK) { dtk_batch_free(b); return rc; }
  if (status) {
    // an empty text only breaks the position modes (token_writer.go:108,135,145)
    *status = st;
    if (!(bits & (DTK_TOKEN_POS | DTK_SENTENCE_POS))) *status &= ~(uint32_t)DTK_ST_EMPTY_TEXT;
  }
  char *p = (char *)malloc((size_t)v.total + 1);
  if (!p) { dtk_batch_free(b); return DTK_E_NOMEM; }
  memcpy(p, v.bytes, (size_t)v.total);
  p[v.total] = 0;
  dtk_batch_free(b);
  *out = p;
  if (out_len) *out_len = (size_t)v.total;
  return DTK_OK;
}

extern "C" int dtk_transduce_replay(const dtk_model *m, const uint8_t *text, size_t n, uint32_t bits,
                             char **out, size_t *out_len, uint32_t *status) {
  if (!m || !out || (n && !text)) return DTK_E_ARG;
  *out = nullptr;
  dtk_batch *b = nullptr;
  int rc = dtk_batch_create(n ? n : 1, 1, &b);
  if (rc != DTK_OK) return rc;
  const uint64_t off[2] = {0, (uint64_t)n};
  dtk_result_view v;
  if ((rc = dtk_batch_set_input(b, text, off, 1)) != DTK_OK ||
      (rc = dtk_batch_run(m, b, bits & DTK_NEWLINE_AFTER_EOT)) != DTK_OK ||
      (rc = dtk_batch_result_host(b, &v)) != DTK_OK) {
    dtk_batch_free(b);
    return rc;
  }
  std::ostringstream os;
  auto tw = datok::NewTokenWriter(os, (datok::Bits)bits);
  datok::detail::replay(std::strcmp(dtk_model_type(m), "MATOK") == 0, text, n, v.events, v.events_open,
                        v.tok_bstart, *tw);
  tw->Flush();
  if (status) {
    // an empty text only breaks the position modes (token_writer.go:108,135,145)
    *status = v.status[0];
    if (!(bits & (DTK_TOKEN_POS | DTK_SENTENCE_POS))) *status &= ~(uint32_t)DTK_ST_EMPTY_TEXT;
  }
  dtk_batch_free(b);
  const std::string s = os.str();
  char *p = (char *)malloc(s.size() + 1);
  if (!p) return DTK_E_NOMEM;
  memcpy(p, s.data(), s.size());
  p[s.size()] = 0;
  *out = p;
  if (out_len) *out_len = s.size();
  return DTK_OK;
}
