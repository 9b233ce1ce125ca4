// datok.hpp -- C++ host mirror of Datok's public Go surface, on top of the
// C-ABI of libdatok_gpu.so (datok_gpu.h).  Header only.
//
// The reference is Go; no Go toolchain exists in the build image, so the host
// side above the C-ABI is C++ with the reference's names, argument meaning and
// error behaviour:
//
//   Go (reference)                                   here
//   ------------------------------------------------ ---------------------------------
//   Bits, TOKENS ... SIMPLE   token_writer.go:9-25   datok::Bits
//   TokenWriter{SentenceEnd,TextEnd,Flush,Token}     datok::TokenWriter (std::function)
//                             token_writer.go:27-33
//   NewTokenWriter(w, flags)  token_writer.go:36-175 datok::NewTokenWriter(std::ostream&, Bits)
//   Tokenizer interface       fomafile.go:29-33      datok::Tokenizer
//   LoadTokenizerFile(file)   fomafile.go:452-484    datok::LoadTokenizerFile -> nullptr + log on error
//   (*T).Transduce(r, w)      matrix.go:340-342      Tokenizer::Transduce(std::istream&, std::ostream&)
//   (*T).TransduceTokenWriter matrix.go:348-698      Tokenizer::TransduceTokenWriter(std::istream&, TokenWriter&)
//   (*T).Type()               matrix.go:102          Tokenizer::Type()
//
// The FSA walk itself runs on the GPU (dtk_batch_run); this header only drains
// the reader, calls the C-ABI and REPLAYS the returned events into the
// caller's four closures in the order the reference would have called them.
#pragma once

#include <cstdint>
#include <cstdio>
#include <functional>
#include <istream>
#include <iterator>
#include <memory>
#include <ostream>
#include <string>
#include <vector>

#include "datok_gpu.h"

namespace datok {

// token_writer.go:9-25
using Bits = uint8_t;
constexpr Bits TOKENS = 1, SENTENCES = 2, TOKEN_POS = 4, SENTENCE_POS = 8, NEWLINE_AFTER_EOT = 16,
               SIMPLE = TOKENS | SENTENCES;

using rune = char32_t;

// token_writer.go:27-33.  Token receives (offset, buf): buf is the rune window
// from its start to the end of the token, offset the number of leading runes
// that are not part of the token (skipped blanks), exactly as upstream.
struct TokenWriter {
  std::function<void(int)> SentenceEnd;
  std::function<void(int)> TextEnd;
  std::function<int()> Flush;
  std::function<void(int, const std::vector<rune> &)> Token;
};

namespace detail {

// Go unicode/utf8.DecodeRune
inline int decode_rune(const uint8_t *p, size_t n, rune *r) {
  *r = 0xFFFD;
  if (n == 0) return 0;
  uint32_t b0 = p[0];
  if (b0 < 0x80) { *r = b0; return 1; }
  if (b0 < 0xC2 || b0 > 0xF4) return 1;
  if (b0 < 0xE0) {
    if (n < 2 || (p[1] & 0xC0) != 0x80) return 1;
    *r = ((b0 & 0x1F) << 6) | (p[1] & 0x3F);
    return 2;
  }
  if (b0 < 0xF0) {
    uint32_t lo = b0 == 0xE0 ? 0xA0 : 0x80, hi = b0 == 0xED ? 0x9F : 0xBF;
    if (n < 3 || p[1] < lo || p[1] > hi || (p[2] & 0xC0) != 0x80) return 1;
    *r = ((b0 & 0x0F) << 12) | ((uint32_t)(p[1] & 0x3F) << 6) | (p[2] & 0x3F);
    return 3;
  }
  uint32_t lo = b0 == 0xF0 ? 0x90 : 0x80, hi = b0 == 0xF4 ? 0x8F : 0xBF;
  if (n < 4 || p[1] < lo || p[1] > hi || (p[2] & 0xC0) != 0x80 || (p[3] & 0xC0) != 0x80) return 1;
  *r = ((b0 & 0x07) << 18) | ((uint32_t)(p[1] & 0x3F) << 12) | ((uint32_t)(p[2] & 0x3F) << 6) |
       (p[3] & 0x3F);
  return 4;
}

// Go string(rune): invalid runes become U+FFFD
inline void append_rune(std::string &s, rune r) {
  if (r > 0x10FFFF || (r >= 0xD800 && r <= 0xDFFF)) r = 0xFFFD;
  if (r < 0x80) s.push_back((char)r);
  else if (r < 0x800) { s.push_back((char)(0xC0 | (r >> 6))); s.push_back((char)(0x80 | (r & 0x3F))); }
  else if (r < 0x10000) {
    s.push_back((char)(0xE0 | (r >> 12))); s.push_back((char)(0x80 | ((r >> 6) & 0x3F)));
    s.push_back((char)(0x80 | (r & 0x3F)));
  } else {
    s.push_back((char)(0xF0 | (r >> 18))); s.push_back((char)(0x80 | ((r >> 12) & 0x3F)));
    s.push_back((char)(0x80 | ((r >> 6) & 0x3F))); s.push_back((char)(0x80 | (r & 0x3F)));
  }
}

inline int count_runes(const uint8_t *p, size_t n) {
  int k = 0;
  size_t i = 0;
  rune r;
  while (i < n) { i += (size_t)decode_rune(p + i, n - i, &r); k++; }
  return k;
}

// Replays the events of document d of a host dtk_result_view (the bitmaps of ev_bits, the tail word, the
// token start offsets) into the closures.  Order at one cursor position: SEOT, TEOT, END, SEPS; the final
// SentenceEnd / TextEnd come last.  The int arguments follow the reference: the matrix passes buffc everywhere
// (matrix.go:575,597,600,684,691), the double array passes 0 except for the SentenceEnd fired by EOT
// (datok.go:1015,1023,1026,1119,1127).
inline void replay(bool is_matrix, const uint8_t *text, size_t n, const dtk_result_view &v, uint64_t doc_off_d,
                   uint32_t d, TokenWriter &w) {
  const uint64_t g0 = DTK_EVENT_BIT(doc_off_d, d);
  auto bit = [&](int kind, size_t p) {
    const uint64_t g = g0 + p;
    return (v.ev_bits[(uint64_t)kind * v.ev_words + (g >> 5)] >> (g & 31)) & 1u;
  };
  const uint32_t *tok_bstart = v.tok_bstart + v.tok_off[d];
  size_t B = 0;      // byte position of the window start (last rewind)
  size_t k = 0;      // tokens replayed so far (index into tok_bstart)
  std::vector<rune> buf;
  auto buffc = [&](size_t p) { return count_runes(text + B, p - B); };
  for (size_t p = 0; p <= n; p++) {
    if (bit(DTK_EVB_SEOT, p)) w.SentenceEnd(buffc(p));
    if (bit(DTK_EVB_TEOT, p)) {
      w.TextEnd(is_matrix ? buffc(p) : 0);
      if (is_matrix) B = p;  // matrix.go:601 rewinds, datok.go:1019-1030 does not
    }
    if (bit(DTK_EVB_END, p)) {
      const size_t start = tok_bstart[k++];
      buf.clear();
      int offset = 0;
      size_t i = B;
      while (i < p) {
        rune r;
        int wd = decode_rune(text + i, n - i, &r);
        if (i < start) offset++;
        buf.push_back(r);
        i += (size_t)wd;
      }
      w.Token(offset, buf);
      B = p;
    }
    if (bit(DTK_EVB_SEPS, p)) w.SentenceEnd(is_matrix ? buffc(p) : 0);
  }
  const uint32_t tail = v.doc_tail[d];
  const size_t pt = tail >> 2;
  if (tail & DTK_TAIL_S) w.SentenceEnd(is_matrix ? buffc(pt) : 0);
  if (tail & DTK_TAIL_E) w.TextEnd(is_matrix ? buffc(pt) : 0);
}

// The same for a document walked by the exact pass (dtk_result_view.calls): the calls are listed in
// the reference's order with their arguments.
inline void replay_calls(const uint8_t *text, size_t n, const dtk_call *calls, size_t n_calls, TokenWriter &w) {
  std::vector<rune> buf;
  for (size_t k = 0; k < n_calls; k++) {
    const dtk_call &c = calls[k];
    if (c.kind == 0) {
      buf.clear();
      int offset = 0;
      size_t i = (size_t)c.a;
      while (i < c.c && i < n) {
        rune r;
        int wd = decode_rune(text + i, n - i, &r);
        if (i < c.b) offset++;
        buf.push_back(r);
        i += (size_t)wd;
      }
      if (c.b > c.c) offset += count_runes(text + c.c, (c.b < n ? c.b : n) - c.c);  // DTK_ST_BAD_OFFSET: offset > len(buf)
      w.Token(offset, buf);
    } else if (c.kind == 1) {
      w.SentenceEnd(c.a);
    } else {
      w.TextEnd(c.a);
    }
  }
}

// document 0 of a one-stream batch (host view): its event bitmaps, or the call list if the exact pass walked it
inline void replay_view(bool is_matrix, const uint8_t *text, size_t n, const dtk_result_view &v, TokenWriter &w) {
  if (v.n_exact && v.exact_doc[0] == 0)
    replay_calls(text, n, v.calls + v.exact_off[0], (size_t)(v.exact_off[1] - v.exact_off[0]), w);
  else
    replay(is_matrix, text, n, v, 0, 0, w);
}

}  // namespace detail

// token_writer.go:36-175
inline std::unique_ptr<TokenWriter> NewTokenWriter(std::ostream &w, Bits flags) {
  struct State {
    int posC = 0;
    std::vector<int> pos, sent;
    bool sentB = true, init = true;
    std::string out;  // bufio.Writer
  };
  auto st = std::make_shared<State>();
  auto tw = std::make_unique<TokenWriter>();
  std::ostream *os = &w;
  auto flush = [st, os]() {
    os->write(st->out.data(), (std::streamsize)st->out.size());
    st->out.clear();
    os->flush();
    return os->good() ? 0 : -1;
  };
  auto surface = [st](int offset, const std::vector<rune> &buf) {
    for (size_t i = (size_t)offset; i < buf.size(); i++) detail::append_rune(st->out, buf[i]);
    st->out.push_back('\n');
  };

  if (flags & (TOKEN_POS | SENTENCE_POS)) {  // :49-88
    tw->Token = [st, flags, surface](int offset, const std::vector<rune> &buf) {
      if (st->posC == 0 && (flags & NEWLINE_AFTER_EOT) && !buf.empty() && buf[0] == U'\n' && !st->init)
        st->posC--;
      st->init = false;
      st->posC += offset;
      st->pos.push_back(st->posC);
      if (st->sentB) { st->sentB = false; st->sent.push_back(st->posC); }
      st->posC += (int)buf.size() - offset;
      st->pos.push_back(st->posC);
      if (flags & TOKENS) surface(offset, buf);
    };
  } else if (flags & TOKENS) {  // :91-95
    tw->Token = [surface](int offset, const std::vector<rune> &buf) { surface(offset, buf); };
  } else {
    tw->Token = [](int, const std::vector<rune> &) {};
  }

  if (flags & SENTENCE_POS) {  // :103-115 (Go panics on an empty pos; we skip the append)
    tw->SentenceEnd = [st, flags](int) {
      if (!st->pos.empty()) st->sent.push_back(st->pos.back());
      st->sentB = true;
      if (flags & SENTENCES) st->out.push_back('\n');
    };
  } else if (flags & SENTENCES) {  // :118-122
    tw->SentenceEnd = [st, flush](int) { st->out.push_back('\n'); flush(); };
  } else {
    tw->SentenceEnd = [](int) {};
  }

  if (flags & (TOKEN_POS | SENTENCE_POS)) {  // :130-159
    tw->TextEnd = [st, flags, flush](int) {
      auto ints = [st](const std::vector<int> &v) {
        for (size_t i = 0; i < v.size(); i++) {
          if (i) st->out.push_back(' ');
          st->out += std::to_string(v[i]);
        }
        st->out.push_back('\n');
      };
      if ((flags & TOKEN_POS) && !st->pos.empty()) ints(st->pos);
      if (flags & SENTENCE_POS) {
        if (!st->sent.empty()) ints(st->sent);
        st->sent.clear();
        st->sentB = true;
      }
      flush();
      st->posC = 0;
      st->pos.clear();
    };
  } else {  // :162-166
    tw->TextEnd = [st, flush](int) { st->out.push_back('\n'); flush(); };
  }
  tw->Flush = flush;  // :170-172
  return tw;
}

// fomafile.go:29-33
class Tokenizer {
 public:
  virtual ~Tokenizer() = default;
  virtual bool Transduce(std::istream &r, std::ostream &w) = 0;
  virtual bool TransduceTokenWriter(std::istream &r, TokenWriter &w) = 0;
  virtual std::string Type() const = 0;
};

// MatrixTokenizer / DaTokenizer behind one class: the encoding is a property of
// the loaded file (matrix.go:16-26, datok.go:63-76).
class GpuTokenizer final : public Tokenizer {
 public:
  explicit GpuTokenizer(dtk_model *m) : m_(m) {}
  ~GpuTokenizer() override { dtk_model_free(m_); }
  GpuTokenizer(const GpuTokenizer &) = delete;
  GpuTokenizer &operator=(const GpuTokenizer &) = delete;

  std::string Type() const override { return dtk_model_type(m_); }
  const dtk_model *model() const { return m_; }

  // matrix.go:340-342 / datok.go:769-771: TransduceTokenWriter(r, NewTokenWriter(w, SIMPLE)).
  // The stock writer's bytes are rendered on the device (dtk_transduce); a custom writer goes
  // through TransduceTokenWriter below.
  bool Transduce(std::istream &r, std::ostream &w) override { return TransduceBits(r, w, SIMPLE); }

  // TransduceTokenWriter(r, NewTokenWriter(w, bits)) for any Bits
  bool TransduceBits(std::istream &r, std::ostream &w, Bits bits) {
    std::string text((std::istreambuf_iterator<char>(r)), std::istreambuf_iterator<char>());
    char *out = nullptr;
    size_t n = 0;
    last_status_ = 0;
    if (dtk_transduce(m_, (const uint8_t *)text.data(), text.size(), (uint32_t)bits, &out, &n, &last_status_) != DTK_OK)
      return false;
    w.write(out, (std::streamsize)n);
    w.flush();
    dtk_free(out);
    return true;
  }

  // matrix.go:348-698 / datok.go:781-1135: one stream = one document.
  bool TransduceTokenWriter(std::istream &r, TokenWriter &w) override {
    std::string text((std::istreambuf_iterator<char>(r)), std::istreambuf_iterator<char>());
    last_status_ = 0;
    const bool ok = TransduceBytes((const uint8_t *)text.data(), text.size(), w);
    w.Flush();  // `defer w.Flush()`, matrix.go:374
    return ok;
  }

  bool TransduceBytes(const uint8_t *text, size_t n, TokenWriter &w) {
    dtk_result_view v;  // on the calling thread's cached batch
    const bool ok = dtk_transduce_result(m_, text, n, 0, &v) == DTK_OK;
    if (ok) {
      last_status_ = v.status[0];
      detail::replay_view(Type() == "MATOK", text, n, v, w);
      // the reference cannot finish such a document whatever the writer does (matrix.go:365,406 index panic,
      // a walk that left the table): false
      if (last_status_ & (DTK_ST_WINDOW_OVERFLOW | DTK_ST_BAD_MODEL | DTK_ST_STEP_LIMIT | DTK_ST_INTERNAL | DTK_ST_IRREGULAR))
        return false;
    }
    return ok;
  }
  uint32_t last_status() const { return last_status_; }

 private:
  dtk_model *m_;
  uint32_t last_status_ = 0;
};

// fomafile.go:452-484: nil + log line on any failure.
inline std::unique_ptr<Tokenizer> LoadTokenizerFile(const std::string &file) {
  dtk_model *m = nullptr;
  const int rc = dtk_model_load(file.c_str(), &m);
  if (rc != DTK_OK) {
    std::fprintf(stderr, "datok: %s: %s\n", file.c_str(), dtk_strerror(rc));
    return nullptr;
  }
  return std::make_unique<GpuTokenizer>(m);
}

}  // namespace datok
