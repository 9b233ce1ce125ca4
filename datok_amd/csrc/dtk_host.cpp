// dtk_host.cpp -- host side of libdatok_gpu.so: model files, device tables,
// batch plumbing.  Everything that computes runs in the .hip units (symbolize, walk, repair, compact, render); nothing
// here walks the automaton.
#include <hip/hip_runtime.h>
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <string_view>
#include <vector>

#include "../../include/datok_gpu.h"
#include "dtk_internal.h"

// --------------------------------------------------------------- error state

static thread_local std::string g_hip_err;

static int hip_fail(hipError_t e, const char *what) {
  g_hip_err = std::string(what) + ": " + hipGetErrorString(e);
  return e == hipErrorNoDevice || e == hipErrorInvalidDevice ? DTK_E_NO_DEVICE : DTK_E_HIP;
}
#define HIP_TRY(call)                                   \
  do {                                                  \
    hipError_t e_ = (call);                             \
    if (e_ != hipSuccess) return hip_fail(e_, #call);   \
  } while (0)

extern "C" const char *dtk_last_hip_error(void) { return g_hip_err.c_str(); }

extern "C" const char *dtk_strerror(int code) {
  switch (code) {
    case DTK_OK: return "ok";
    case DTK_E_IO: return "cannot open or read the tokenizer file";
    case DTK_E_FORMAT: return "not a gzip'd MATOK/DATOK file (magic, version or length)";
    case DTK_E_NO_DEVICE: return "no usable HIP device (this library has no CPU path)";
    case DTK_E_HIP: return "HIP runtime error (see dtk_last_hip_error)";
    case DTK_E_ARG: return "invalid argument";
    case DTK_E_MODEL: return "model outside device limits (symbols >= 2048, special ids out of range, epsilon cycle)";
    case DTK_E_CAPACITY: return "batch exceeds the capacity it was created with";
    case DTK_E_STATE: return "call out of order";
    case DTK_E_NOMEM: return "out of host memory";
    default: return "unknown error";
  }
}

extern "C" int dtk_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" int dtk_set_device(int device) {
  HIP_TRY(hipSetDevice(device));
  return DTK_OK;
}

// ---------------------------------------------------------------- test hooks
//
// The library's behaviour does not depend on the caller's environment: nothing here reads it.  The switches
// below select code paths that the library otherwise picks by model, shape or history, so that the tests can run the
// whole suite over each of them; only dtk_debug_configure sets them (the Python harness forwards DATOK_* variables
// to it, datok_amd/_lib.py -- the shipped entry points never look).  None of them changes a result.
struct DtkDebug {
  int sym16 = 0;         // 16-bit stream entries (and the general loop) although the model's entries fit a code table
  int force_wide = 0;    // 32-bit plain cells for any model (MatrixTrans<uint32_t>)
  int file_columns = 0;  // keep the file's column order
  int no_fused = 0;      // plain uint16 cells: no fused epsilon + rune cells
  int plain_walk = 0;    // the general loop for a model the lean loop would serve
  int no_dense = 0;      // walk a double array's {base, check} pairs instead of its dense layout
  int small_max = -1;    // documents of at most this many bytes are compacted one per lane (-1: by batch shape)
  int warm_ws = 0, warm_min = 0;  // warm-up start behind the n-th run of blanks (0: off)
  int warm_extend = -1;  // overrides dtk_batch_set_warm_extend (-1: do not)
  int lds_bits = 1;      // 0: event bits straight to memory
  int split_start = 0;   // start records and chunk walk as two launches
  int dev_rounds = -1;   // repair rounds enqueued with every run (-1: two after a run that had to repair)
  int compact_full = 0;  // both compaction kernels with every run
  int clear_kernel = 0;  // the accumulator block is cleared by k_clear2, not by k_symbolize's blocks
  int round_limit = -1;  // host repair rounds before the one-lane-per-document fallback (-1: the longest document's lanes)
  int debug_repair = 0;  // print the lane records of documents that stay broken
  int exp_skip = 0;      // (DTK_EXPERIMENTS builds) stages skipped from the second run on
};
static DtkDebug g_dbg;

extern "C" int dtk_debug_configure(const char *key, const char *value) {
  if (!key) return DTK_E_ARG;
  const int v = value ? atoi(value) : 1;
  const struct { const char *name; int *field; bool flag; } tab[] = {
      {"SYM16", &g_dbg.sym16, true}, {"FORCE_WIDE", &g_dbg.force_wide, true}, {"FILE_COLUMNS", &g_dbg.file_columns, true},
      {"NO_FUSED", &g_dbg.no_fused, true}, {"PLAIN_WALK", &g_dbg.plain_walk, true}, {"NO_DENSE", &g_dbg.no_dense, true},
      {"SMALL_MAX", &g_dbg.small_max, false}, {"WARM_WS", &g_dbg.warm_ws, false}, {"WARM_MIN", &g_dbg.warm_min, false},
      {"WARM_EXTEND", &g_dbg.warm_extend, false}, {"LDS_BITS", &g_dbg.lds_bits, false}, {"SPLIT_START", &g_dbg.split_start, false},
      {"DEV_ROUNDS", &g_dbg.dev_rounds, false}, {"COMPACT_FULL", &g_dbg.compact_full, true}, {"CLEAR_KERNEL", &g_dbg.clear_kernel, true},
      {"ROUND_LIMIT", &g_dbg.round_limit, false}, {"DEBUG_REPAIR", &g_dbg.debug_repair, true}, {"EXP_SKIP", &g_dbg.exp_skip, false}};
  if (strncmp(key, "DATOK_", 6) == 0) key += 6;
  for (const auto &e : tab)
    if (strcmp(key, e.name) == 0) {
      // (a flag is on by being named, as the environment variables were: `DATOK_NO_FUSED=` or `=1` both switch it on)
      *e.field = e.flag ? ((value && value[0] == '0' && value[1] == 0) ? 0 : 1) : v;
      return DTK_OK;
    }
  return DTK_E_ARG;
}

// ------------------------------------------------------------------- model

struct dtk_model {
  int kind = 0;
  int epsilon = 0, unknown = 0, identity = 0, final_state = 0, sigma_count = 0;
  uint32_t state_count = 0;
  std::vector<uint16_t> col;   // symbol -> column of the device table (layout_matrix); empty: the symbol itself
  uint32_t dense_states = 0;   // double array laid out as a matrix (densify in build_datok): its states; 0: the pairs are walked
  uint64_t array_len = 0;
  uint32_t n_eps_states = 0, max_eps_chain = 0, unknown_used = 0;
  uint64_t device_bytes = 0;
  int device = 0;
  // host copy of the sigma map, for rendering (Go string(rune) of a token surface)
  std::vector<uint32_t> sigma_runes;
  std::vector<uint16_t> sigma_syms;
  uint16_t ascii[256];
  // device
  void *d_tab = nullptr;
  uint16_t *d_ascii = nullptr;
  uint32_t *d_runes = nullptr;
  uint16_t *d_syms = nullptr;
  void *d_codes = nullptr;  // code_entry [256] u16, code_lt256 [256] u8, code_runes [n_runes] u8
  DtkTableDev tab{};
  DtkSigmaDev sig{};
};

// Go unicode/utf8.DecodeRune (host copy, used for the sigma block of the model
// files, matrix.go:296 / datok.go:687, and for rendering surfaces).
static int go_decode_host(const uint8_t *p, size_t n, uint32_t *r) {
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
  *r = ((b0 & 0x07) << 18) | ((uint32_t)(p[1] & 0x3F) << 12) | ((uint32_t)(p[2] & 0x3F) << 6) | (p[3] & 0x3F);
  return 4;
}

static uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static uint32_t rd32(const uint8_t *p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

// gzip.NewReader + ReadAll (matrix.go:222, fomafile.go:460)
static int gunzip(const uint8_t *gz, size_t n, std::vector<uint8_t> &out) {
  if (n < 18 || gz[0] != 0x1f || gz[1] != 0x8b) return DTK_E_FORMAT;
  z_stream zs;
  memset(&zs, 0, sizeof zs);
  if (inflateInit2(&zs, 16 + MAX_WBITS) != Z_OK) return DTK_E_FORMAT;
  zs.next_in = const_cast<Bytef *>(gz);
  zs.avail_in = (uInt)n;
  out.resize(std::max<size_t>(n * 8, 1 << 16));
  size_t have = 0;
  int rc;
  do {
    if (have == out.size()) out.resize(out.size() * 2);
    zs.next_out = out.data() + have;
    zs.avail_out = (uInt)std::min<size_t>(out.size() - have, 1u << 30);
    const size_t before = zs.avail_out;
    rc = inflate(&zs, Z_NO_FLUSH);
    have += before - zs.avail_out;
  } while (rc == Z_OK);
  inflateEnd(&zs);
  if (rc != Z_STREAM_END) return DTK_E_FORMAT;
  out.resize(have);
  return DTK_OK;
}

// Sigma block: sigmaCount UTF-8 runes, NUL = "no character" (matrix.go:288-303,
// datok.go:679-694).  A later index overwrites an earlier one, like the Go map.
static size_t parse_sigma(dtk_model *m, const std::vector<uint8_t> &raw, size_t off) {
  for (int i = 0; i < 256; i++) m->ascii[i] = (uint16_t)m->identity;
  std::vector<std::pair<uint32_t, uint16_t>> ent;
  for (int x = 0; x < m->sigma_count; x++) {
    if (off >= raw.size()) continue;
    uint32_t r;
    int w = go_decode_host(raw.data() + off, raw.size() - off, &r);
    off += (size_t)w;
    if (r != 0) {
      if (r < 256) m->ascii[r] = (uint16_t)x;
      ent.emplace_back(r, (uint16_t)x);
    }
  }
  std::stable_sort(ent.begin(), ent.end(),
                   [](const auto &a, const auto &b) { return a.first < b.first; });
  for (auto &e : ent) {
    if (!m->sigma_runes.empty() && m->sigma_runes.back() == e.first) m->sigma_syms.back() = e.second;
    else { m->sigma_runes.push_back(e.first); m->sigma_syms.push_back(e.second); }
  }
  return off;
}

static int upload(dtk_model *m, const void *tab, size_t tab_bytes) {
  HIP_TRY(hipGetDevice(&m->device));
  // (2048 cells of slack behind the last row: the lean walk asks for its next cell before it knows that the reader is
  //  at EOF, with whatever stream entry lies behind the document -- any 11-bit symbol index; the cell is not used)
  const size_t slack = (DTK_SYM_MASK + 1u) * 4u;
  HIP_TRY(hipMalloc(&m->d_tab, std::max<size_t>(tab_bytes, 16) + slack));
  HIP_TRY(hipMemset((char *)m->d_tab + tab_bytes, 0, slack));
  HIP_TRY(hipMemcpy(m->d_tab, tab, tab_bytes, hipMemcpyHostToDevice));
  // (symbols as the device sees them: the column of the table they index, see layout_matrix)
  auto colof = [&](int sym) -> uint32_t {
    return (sym >= 0 && (size_t)sym < m->col.size()) ? m->col[(size_t)sym] : (uint32_t)sym;
  };
  uint16_t ascii_dev[256];
  for (int i = 0; i < 256; i++) ascii_dev[i] = (uint16_t)colof(m->ascii[i]);
  HIP_TRY(hipMalloc((void **)&m->d_ascii, 256 * sizeof(uint16_t)));
  HIP_TRY(hipMemcpy(m->d_ascii, ascii_dev, 256 * sizeof(uint16_t), hipMemcpyHostToDevice));
  // the device's sorted rune list only holds runes >= 256 (the others go through the 256-entry table)
  size_t first = 0;
  while (first < m->sigma_runes.size() && m->sigma_runes[first] < 256u) first++;
  const size_t nr = m->sigma_runes.size() - first;
  HIP_TRY(hipMalloc((void **)&m->d_runes, std::max<size_t>(nr, 1) * sizeof(uint32_t)));
  HIP_TRY(hipMalloc((void **)&m->d_syms, std::max<size_t>(nr, 1) * sizeof(uint16_t)));
  if (nr) {
    HIP_TRY(hipMemcpy(m->d_runes, m->sigma_runes.data() + first, nr * sizeof(uint32_t), hipMemcpyHostToDevice));
    std::vector<uint16_t> syms_dev(nr);
    for (size_t i = 0; i < nr; i++) syms_dev[i] = (uint16_t)colof(m->sigma_syms[first + i]);
    HIP_TRY(hipMemcpy(m->d_syms, syms_dev.data(), nr * sizeof(uint16_t), hipMemcpyHostToDevice));
  }
  m->device_bytes = tab_bytes + slack + 512 + nr * 6;
  {
    // The stream's code table (dtk_internal.h): every entry the symboliser can write for this model, numbered.
    std::vector<uint16_t> entries;
    bool fits = true;
    auto code_of = [&](uint32_t e) -> uint8_t {
      for (size_t i = 0; i < entries.size(); i++)
        if (entries[i] == (uint16_t)e) return (uint8_t)i;
      if (entries.size() >= DTK_SYM_CONT) { fits = false; return 0; }
      entries.push_back((uint16_t)e);
      return (uint8_t)(entries.size() - 1);
    };
    const uint32_t ident = m->identity < 0 ? 0u : colof(m->identity);
    std::vector<uint8_t> bytes(256 + nr);
    for (uint32_t i = 0; i < 256; i++) {  // matrix.go:421-426; a rune of 128..255 takes two bytes
      const uint32_t e = (ascii_dev[i] & DTK_SYM_MASK) | (i == DTK_EOT ? 1u << DTK_SYM_CLS_SHIFT : 0u) |
                         ((i < 128 ? 1u : 2u) << DTK_SYM_W_SHIFT);
      // (the code of a byte < 128 is the byte: the symboliser copies those, k_symbolize's light path)
      if (i < 128) { entries.push_back((uint16_t)e); bytes[i] = (uint8_t)i; } else bytes[i] = code_of(e);
    }
    uint32_t fffd = (ident & DTK_SYM_MASK) | (3u << DTK_SYM_CLS_SHIFT);
    for (size_t i = 0; i < nr; i++) {
      const uint32_t r = m->sigma_runes[first + i], w = r < 0x800u ? 2u : (r < 0x10000u ? 3u : 4u);
      const uint32_t e = (colof(m->sigma_syms[first + i]) & DTK_SYM_MASK) | (2u << DTK_SYM_CLS_SHIFT);
      bytes[256 + i] = code_of(e | (w << DTK_SYM_W_SHIFT));
      if (r == 0xFFFDu) fffd = e;
    }
    for (uint32_t w = 1; w <= 4; w++)
      m->sig.code_ident[w] = code_of((ident & DTK_SYM_MASK) | (3u << DTK_SYM_CLS_SHIFT) | (w << DTK_SYM_W_SHIFT));
    m->sig.code_ident[0] = DTK_SYM_CONT;
    m->sig.code_fffd1 = code_of(fffd | (1u << DTK_SYM_W_SHIFT));  // an invalid byte decodes to U+FFFD, one byte wide
    m->sig.n_codes = (fits && !g_dbg.sym16) ? (uint32_t)entries.size() : 0u;
    entries.resize(256, 0);  // (DTK_SYM_CONT and the unused codes: width 0)
    HIP_TRY(hipMalloc(&m->d_codes, 512 + bytes.size()));
    HIP_TRY(hipMemcpy(m->d_codes, entries.data(), 512, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy((char *)m->d_codes + 512, bytes.data(), bytes.size(), hipMemcpyHostToDevice));
    m->sig.code_entry = (const uint16_t *)m->d_codes;
    m->sig.code_lt256 = (const uint8_t *)m->d_codes + 512;
    m->sig.code_runes = (const uint8_t *)m->d_codes + 768;
    m->device_bytes += 512 + bytes.size();
  }
  m->sig.ascii = m->d_ascii;
  m->sig.runes = m->d_runes;
  m->sig.syms = m->d_syms;
  m->sig.n_runes = (uint32_t)nr;
  // a net without identity symbol (-1, fomafile.go:88): unmapped runes get symbol 0, which has
  // no arcs (matrix.go:459), and no symbol ever equals the identity (matrix.go:478)
  m->sig.identity = m->identity < 0 ? 0u : colof(m->identity);
  m->tab.tab = m->d_tab;
  m->tab.epsilon = colof(m->epsilon);
  m->tab.unknown = m->unknown < 0 ? (uint32_t)m->unknown : colof(m->unknown);
  m->tab.identity = m->identity < 0 ? (uint32_t)m->identity : colof(m->identity);
  if (m->tab.ident_guard != 0xFFFFFFFFu) m->tab.ident_guard = colof((int)m->tab.ident_guard);
  return DTK_OK;
}

static bool special_ids_ok(const dtk_model *m) {
  // The walk compares symbols against these ids; the symbol stream has 11 bits.
  // identity == unknown makes the retry of matrix.go:478-485 spin forever upstream.
  return m->sigma_count >= 1 && m->sigma_count <= (int)DTK_SYM_MAX && m->epsilon >= 1 &&
         m->epsilon < m->sigma_count && m->unknown < m->sigma_count && m->identity < m->sigma_count &&
         (m->identity != m->unknown || m->identity < 0) && m->epsilon != m->identity;
}

static int layout_matrix(dtk_model *m, const std::vector<uint32_t> &arr, uint64_t n_states, bool da_dense);

// ParseMatrix (matrix.go:235-337)
static int build_matrix(dtk_model *m, const std::vector<uint8_t> &raw) {
  if (raw.size() < 19) return DTK_E_FORMAT;
  const uint8_t *h = raw.data() + 5;
  if (rd16(h) != 1) return DTK_E_FORMAT;  // VERSION
  m->kind = DTK_KIND_MATRIX;
  m->epsilon = rd16(h + 2);
  m->unknown = rd16(h + 4);
  m->identity = rd16(h + 6);
  m->state_count = rd32(h + 8);
  m->sigma_count = rd16(h + 12);
  m->array_len = ((uint64_t)m->state_count + 1) * (uint64_t)m->sigma_count;  // matrix.go:286
  size_t off = parse_sigma(m, raw, 19);
  if (off >= raw.size() || raw[off] != 'M') return DTK_E_FORMAT;  // matrix.go:305-315
  off++;
  if (raw.size() - off < m->array_len * 4) return DTK_E_FORMAT;   // matrix.go:327-330
  if (!special_ids_ok(m) || m->state_count == 0 || m->state_count >= 0x7FFFFFFFu) return DTK_E_MODEL;
  std::vector<uint32_t> arr(m->array_len);
  for (uint64_t x = 0; x < m->array_len; x++) arr[x] = rd32(raw.data() + off + x * 4);
  return layout_matrix(m, arr, m->state_count, false);
}

// Device layout of a matrix tokenizer whose header fields and sigma are set in `m` and whose
// symbol-major array (matrix.go:463) is `arr`.
// (n_states / da_dense: the same layout for the transitions of a double-array tokenizer, see densify_datok)
static int layout_matrix(dtk_model *m, const std::vector<uint32_t> &arr, uint64_t n_states, bool da_dense) {
  const uint64_t N = n_states, S = (uint64_t)m->sigma_count;
  auto cell = [&](uint64_t a, uint64_t t) -> uint32_t {  // array[(a-1)*stateCount + t], matrix.go:463
    return arr[(a - 1) * N + t];
  };

  // renumber: states with an epsilon arc first (probe of matrix.go:442 -> compare)
  std::vector<uint32_t> newid(N + 1, 0);
  uint32_t next = 1;
  for (uint64_t t = 1; t <= N; t++)
    if (cell((uint64_t)m->epsilon, t) != 0) newid[t] = next++;
  m->n_eps_states = next - 1;
  for (uint64_t t = 1; t <= N; t++)
    if (newid[t] == 0) newid[t] = next++;

  // epsilon chains: reject cycles (the reference would never return, matrix.go:633-634)
  {
    std::vector<uint32_t> depth(N + 1, 0);
    uint32_t best = 0;
    for (uint64_t t = 1; t <= N; t++) {
      uint64_t cur = t;
      uint32_t len = 0;
      while (true) {
        uint32_t nx = cell((uint64_t)m->epsilon, cur) & ~DTK_FIRSTBIT;
        if (nx == 0 || nx > N) break;
        len++;
        if (len > N) return DTK_E_MODEL;
        cur = nx;
      }
      best = std::max(best, len);
    }
    m->max_eps_chain = best;
  }

  // 32-bit cells for automata with 32767 states or more; DATOK_FORCE_WIDE=1 selects them for any
  // model (no shipped model is that large: this is how the tests reach MatrixTrans<uint32_t>)
  const bool wide = (N + 1) > 0x7FFFu || g_dbg.force_wide;
  const uint32_t stride = (uint32_t)((S + 7) & ~7ull);
  const size_t cells_total = (size_t)(N + 1) * stride;
  // Columns: the symbols of running text first -- blank, the lower-case letters by frequency, full stop, comma,
  // newline, umlauts, digits, capitals --, so that the cells a row is mostly asked for share one or two cache lines
  // (in file order the letters, the blank and the punctuation of a row lie in four or five).  Symbol 0 keeps
  // column 0 (no arcs, matrix.go:459).  The device's symbol tables and special symbols are mapped in upload().
  std::vector<uint16_t> &col = m->col;
  col.assign((size_t)S, 0xFFFFu);
  {
    static const uint32_t order[] = {' ', 'e', 'n', 'i', 's', 'r', 'a', 't', 'd', 'h', 'u', 'l', 'c', 'g', 'm', 'o', 'b', 'w', 'f',
                                     'k', 'z', 'p', 'v', '.', ',', '\n', 0xFC, 0xE4, 0xF6, 0xDF, 'j', 'y', 'x', 'q', '-', '\'', '"',
                                     '0', '1', '2', '3', '4', '5', '6', '7', '8', '9', 'S', 'D', 'A', 'E', 'B', 'M', 'K', 'W', 'G',
                                     'H', 'T', 'I', 'P', 'L', 'R', 'F', 'N', 'V', 'Z', 'U', 'O', 'J', 'C', ':', ';', '?', '!', '(', ')',
                                     '/', '\t', '\r'};
    uint32_t next_col = 1;
    col[0] = 0;
    if (!g_dbg.file_columns)
      for (uint32_t r : order)
        for (size_t i = 0; i < m->sigma_runes.size(); i++)
          if (m->sigma_runes[i] == r) {
            const uint32_t a = m->sigma_syms[i];
            if (a > 0 && a < S && col[a] == 0xFFFFu) col[a] = (uint16_t)next_col++;
          }
    for (uint64_t a = 1; a < S; a++)
      if (col[a] == 0xFFFFu) col[a] = (uint16_t)next_col++;
  }
  // 15-bit state ids: uint32 cells with fused epsilon+rune entries (MatrixFusedTrans);
  // DATOK_NO_FUSED=1 keeps the plain uint16 table (for A/B measurements)
  const bool fused = !wide && !g_dbg.no_fused;
  const size_t cell_bytes = (wide || fused) ? 4 : 2;
  std::vector<uint8_t> host(cells_total * cell_bytes, 0);
  auto put = [&](size_t at, uint32_t v) {
    if (cell_bytes == 4) memcpy(host.data() + at * 4, &v, 4);
    else { uint16_t h = (uint16_t)v; memcpy(host.data() + at * 2, &h, 2); }
  };
  for (uint64_t a = 1; a < S; a++) {
    for (uint64_t t = 1; t <= N; t++) {
      const uint32_t x = cell(a, t);
      const uint32_t tgt = x & ~DTK_FIRSTBIT;
      if (tgt == 0) continue;
      if (tgt > N) return DTK_E_MODEL;
      if ((int)a == m->unknown) m->unknown_used = 1;
      const size_t at = (size_t)newid[t] * stride + col[a];
      if (wide) put(at, newid[tgt] | (x & DTK_FIRSTBIT));
      else put(at, newid[tgt] | ((x & DTK_FIRSTBIT) ? 0x8000u : 0u));
    }
  }
  if (fused) {
    // (t, a) empty, t --epsilon--> e, (e, a) present:  1<<31 | e<<16 | cell(e, a)
    for (uint64_t t = 1; t <= N; t++) {
      const uint32_t e = cell((uint64_t)m->epsilon, t) & ~DTK_FIRSTBIT;
      if (e == 0 || e > N) continue;
      for (uint64_t a = 1; a < S; a++) {
        if ((int)a == m->epsilon || (int)a == m->unknown) continue;
        if ((cell(a, t) & ~DTK_FIRSTBIT) != 0) continue;
        const uint32_t x2 = cell(a, e);
        const uint32_t tgt2 = x2 & ~DTK_FIRSTBIT;
        if (tgt2 == 0 || tgt2 > N) continue;
        put((size_t)newid[t] * stride + col[a],
            0x80000000u | (newid[e] << 16) | newid[tgt2] | ((x2 & DTK_FIRSTBIT) ? 0x8000u : 0u));
      }
    }
  }
  m->tab.kind = DTK_KIND_MATRIX;
  m->tab.entry_bytes = (uint32_t)cell_bytes;
  m->tab.fused = fused ? 1u : 0u;
  m->tab.ident_guard = m->unknown_used ? (uint32_t)m->identity : 0xFFFFFFFFu;
  m->tab.plain_walk = g_dbg.plain_walk ? 1u : 0u;
  m->tab.da_dense = da_dense ? 1u : 0u;
  m->tab.stride = stride;
  m->tab.n_states = (uint32_t)N;
  m->tab.n_eps = m->n_eps_states;
  m->tab.start = newid[1];
  return upload(m, host.data(), host.size());
}

// ParseDatok (datok.go:621-729) + device layout.
static int build_datok(dtk_model *m, const std::vector<uint8_t> &raw) {
  if (raw.size() < 21) return DTK_E_FORMAT;
  const uint8_t *h = raw.data() + 5;
  if (rd16(h) != 1) return DTK_E_FORMAT;
  m->kind = DTK_KIND_DA;
  m->epsilon = rd16(h + 2);
  m->unknown = rd16(h + 4);
  m->identity = rd16(h + 6);
  m->final_state = rd16(h + 8);
  m->sigma_count = rd16(h + 10);
  m->array_len = rd32(h + 12) / 2;  // "Legacy support", datok.go:674
  size_t off = parse_sigma(m, raw, 21);
  if (off >= raw.size() || raw[off] != 'T') return DTK_E_FORMAT;
  off++;
  if (raw.size() - off < m->array_len * 8) return DTK_E_FORMAT;
  if (!special_ids_ok(m) || m->array_len < 2 || m->array_len >= DTK_RESTBIT) return DTK_E_MODEL;

  const uint64_t L = m->array_len;
  std::vector<uint32_t> bc(L * 2);
  for (uint64_t i = 0; i < L * 2; i++) bc[i] = rd32(raw.data() + off + i * 4);
  const uint32_t size = bc[3] & DTK_RESTBIT;  // array[1].check, datok.go:333-335
  m->state_count = size;
  auto has_eps = [&](uint64_t s) -> bool {  // datok.go:876
    const uint64_t ei = (uint64_t)(bc[2 * s] & DTK_RESTBIT) + (uint32_t)m->epsilon;
    return ei < L && (bc[2 * ei + 1] & DTK_RESTBIT) == s;
  };
  // epsilon arc target incl. representative hop (datok.go:889-901, 1056-1058)
  auto eps_target = [&](uint64_t s) -> uint64_t {
    const uint64_t ei = (uint64_t)(bc[2 * s] & DTK_RESTBIT) + (uint32_t)m->epsilon;
    if (ei >= L || ei > size || (bc[2 * ei + 1] & DTK_RESTBIT) != s) return 0;
    if (bc[2 * ei] & DTK_FIRSTBIT) return bc[2 * ei] & DTK_RESTBIT;
    return ei;
  };
  uint32_t neps = 0, best = 0;
  for (uint64_t s = 1; s < L; s++) {
    if (!has_eps(s)) continue;
    neps++;
    uint64_t cur = s;
    uint32_t len = 0;
    while (true) {
      uint64_t nx = eps_target(cur);
      if (nx == 0 || nx >= L) break;
      len++;
      if (len > 4096) return DTK_E_MODEL;  // epsilon cycle
      cur = nx;
    }
    best = std::max(best, len);
  }
  m->n_eps_states = neps;
  m->max_eps_chain = best;
  for (uint64_t i = 1; i < L && i <= size; i++) {
    const uint64_t par = bc[2 * i + 1] & DTK_RESTBIT;
    if (par && par < L && (uint64_t)(bc[2 * par] & DTK_RESTBIT) + (uint32_t)m->unknown == i) {
      m->unknown_used = 1;
      break;
    }
  }
  // The double array is a compressed encoding of the same kind of automaton the matrix holds dense: a shipped
  // model has some 20 000 states hidden in its 2.9 million pairs.  If they fit the fused cells (15-bit state ids),
  // the device walks them as a matrix -- one load per step instead of two dependent ones, the lean loop, fused
  // epsilon cells -- under the double array's own rules for EOT (datok.go:1019-1030; walk_fused<.., IS_MATRIX =
  // false>, compaction and replay go by m->kind).  DATOK_NO_DENSE=1 keeps the pairs (DaTrans; what the tests of that
  // path set).
  // (only with fused cells: the lean loop and the fused general loop carry the double array's EOT rules; the plain
  //  matrix encodings that DATOK_NO_FUSED / DATOK_FORCE_WIDE select for the tests are not run with them)
  if (!g_dbg.no_dense && !g_dbg.no_fused && !g_dbg.force_wide) {
    std::vector<uint32_t> arr;
    uint64_t n_dense = 0;
    // one step of datok.go:888-901 + 1055-1063 from state s0 on symbol a: 0 = no arc, else target | FIRSTBIT if
    // non-token; false = an access the reference would panic on (no dense form for such a file)
    auto step = [&](uint64_t s0, uint32_t a, uint32_t &out) -> bool {
      out = 0;
      const uint64_t idx = (uint64_t)(bc[2 * s0] & DTK_RESTBIT) + a;
      if (idx >= L) return false;          // datok.go:888-889 reads array[t] before any test: an index panic
      if (idx > size) {                    // datok.go:896: t > check(1) fails ...
        // ... but the epsilon probe of datok.go:876 has no such bound: a hand-made file whose state "has" an epsilon
        // arc behind check(1) remembers a slot the dense layout would not know -- the pairs are walked
        if (a == (uint32_t)m->epsilon && (bc[2 * idx + 1] & DTK_RESTBIT) == s0) return false;
        return true;
      }
      if ((bc[2 * idx + 1] & DTK_RESTBIT) != s0) return true;
      uint64_t t = idx;
      if (bc[2 * idx] & DTK_FIRSTBIT) {    // separate: the representative
        t = bc[2 * idx] & DTK_RESTBIT;
        if (t >= L || t == 0) return false;
      }
      out = (uint32_t)t | ((bc[2 * idx + 1] & DTK_FIRSTBIT) ? DTK_FIRSTBIT : 0u);
      return true;
    };
    bool ok = true;
    std::vector<uint32_t> id(L, 0);        // double-array index -> dense state number (1 = start)
    std::vector<uint64_t> order{0, 1};     // dense state number -> index
    id[1] = 1;
    const uint32_t S = (uint32_t)m->sigma_count;
    std::vector<uint32_t> trans;           // state-major while the states are being discovered
    for (uint64_t q = 1; ok && q < order.size(); q++) {
      const uint64_t s0 = order[q];
      uint32_t x;
      ok = step(s0, 0u, x) && x == 0u;     // symbol 0 must have no arc (the matrix's column 0, matrix.go:459)
      for (uint32_t a = 1; ok && a < S; a++) {
        ok = step(s0, a, x);
        const uint64_t t = x & ~DTK_FIRSTBIT;
        if (ok && t != 0 && id[t] == 0) {
          id[t] = (uint32_t)order.size();
          order.push_back(t);
          ok = order.size() <= 0x7FFFu;    // the fused cells' 15-bit state ids
        }
        trans.push_back(ok && t != 0 ? (id[t] | (x & DTK_FIRSTBIT)) : 0u);
      }
    }
    if (ok) {
      n_dense = order.size() - 1;
      arr.assign((size_t)(S - 1) * n_dense + n_dense + 1, 0);  // array[(a-1)*N + t], t in 1..N (matrix.go:463)
      for (uint64_t q = 1; q <= n_dense; q++)
        for (uint32_t a = 1; a < S; a++) arr[(size_t)(a - 1) * n_dense + q] = trans[(size_t)(q - 1) * (S - 1) + (a - 1)];
      const uint32_t keep_states = m->state_count;
      const int rc = layout_matrix(m, arr, n_dense, true);
      m->state_count = keep_states;
      m->dense_states = (uint32_t)n_dense;
      return rc;
    }
  }
  std::vector<uint32_t> dev(L * 2);
  for (uint64_t s = 0; s < L; s++) {
    dev[2 * s] = (bc[2 * s] & (DTK_FIRSTBIT | DTK_RESTBIT)) | (s >= 1 && has_eps(s) ? DTK_SECONDBIT : 0u);
    dev[2 * s + 1] = bc[2 * s + 1];
  }
  m->tab.kind = DTK_KIND_DA;
  m->tab.entry_bytes = 8;
  m->tab.da_len = (uint32_t)L;
  m->tab.da_size = size;
  m->tab.da_base1 = dev[2];
  m->tab.start = 1;
  return upload(m, dev.data(), dev.size() * 4);
}

// ------------------------------------------------------------- foma text nets
//
// LoadFomaFile/ParseFoma (fomafile.go:56-450) followed by Automaton.ToMatrix (matrix.go:30-99):
// a deterministic, epsilon-free Foma net in text form becomes the matrix tokenizer.  The
// double-array construction (ToDoubleArray, datok.go:82-238) follows further down (dtk_foma_to_datok).
namespace {
struct FomaArc { int32_t end; bool nontoken; bool tokenend = false; };
struct FomaNet {
  int epsilon = -1, unknown = -1, identity = -1, final_sym = -1, tokenend = -1;
  int sigma_count = 0, state_count = -1;
  std::vector<std::pair<int, uint32_t>> chars;  // (symbol number, rune) -- sigmaRev
  std::vector<char> mcs;                        // symbol number -> multi-character symbol
  std::vector<std::vector<std::pair<int, FomaArc>>> arcs;  // per state (1-based): last write wins
};

static std::vector<std::string_view> split_sp(std::string_view s, size_t max_parts) {
  std::vector<std::string_view> out;
  while (out.size() + 1 < max_parts) {
    const size_t k = s.find(' ');
    if (k == std::string_view::npos) break;
    out.push_back(s.substr(0, k));
    s.remove_prefix(k + 1);
  }
  out.push_back(s);
  return out;
}

static bool to_int(std::string_view s, int &v) {  // strconv.Atoi
  if (s.empty()) return false;
  size_t i = (s[0] == '-' || s[0] == '+') ? 1 : 0;
  if (i == s.size()) return false;
  long long x = 0;
  for (; i < s.size(); i++) {
    if (s[i] < '0' || s[i] > '9') return false;
    x = x * 10 + (s[i] - '0');
    if (x > 0x7FFFFFFFll) return false;
  }
  v = (int)(s[0] == '-' ? -x : x);
  return true;
}

static int parse_foma(const std::vector<uint8_t> &raw, FomaNet &net) {
  enum { NONE, PROPS, SIGMA, STATES } mode = NONE;
  std::string_view rest((const char *)raw.data(), raw.size());
  int state = 0, in_sym = 0, out_sym = 0, end = 0, fin = 0;
  auto next_line = [&](std::string_view &line) -> bool {  // ReadString('\n'), fomafile.go:101-108
    const size_t k = rest.find('\n');
    if (k == std::string_view::npos) return false;  // an unterminated last line is dropped
    line = rest.substr(0, k);                       // without the newline
    rest.remove_prefix(k + 1);
    return true;
  };
  auto set_arc = [&](int st, int sym, FomaArc a) {
    auto &v = net.arcs[(size_t)st];
    for (auto &e : v) if (e.first == sym) { e.second = a; return; }
    v.emplace_back(sym, a);
  };
  std::string_view line;
  while (next_line(line)) {
    if (line.substr(0, 2) == "##") {  // fomafile.go:111-135
      if (line.substr(0, 9) == "##props##") mode = PROPS;
      else if (line.substr(0, 10) == "##states##") { mode = STATES; net.final_sym = ++net.sigma_count; }
      else if (line.substr(0, 9) == "##sigma##") mode = SIGMA;
      else if (line.substr(0, 7) == "##end##") mode = NONE;
      else if (line.substr(0, 10) != "##foma-net") break;
      continue;
    }
    if (mode == PROPS) {  // fomafile.go:140-187
      auto f = split_sp(line, 64);
      if (f.size() < 10 || f[6] != "1" || f[9] != "1") return DTK_E_MODEL;  // deterministic, epsilon free
      int arcs_n;
      if (!to_int(f[1], arcs_n) || !to_int(f[2], net.state_count) || net.state_count < 1) return DTK_E_FORMAT;
      net.arcs.assign((size_t)net.state_count + 2, {});
    } else if (mode == SIGMA) {  // fomafile.go:372-444
      auto f = split_sp(line, 2);
      int number;
      if (f.size() < 2 || !to_int(f[0], number) || number < 0 || number > 0xFFFF) return DTK_E_FORMAT;
      number++;
      net.sigma_count = number;
      if ((size_t)number >= net.mcs.size()) net.mcs.resize((size_t)number + 1, 0);
      const std::string_view sym = f[1];
      size_t runes = 0;
      uint32_t r = 0;
      for (size_t i = 0; i < sym.size(); runes++)
        i += (size_t)go_decode_host((const uint8_t *)sym.data() + i, sym.size() - i, &r);
      if (runes == 1) {
        net.chars.emplace_back(number, r);
      } else if (runes > 1) {
        if (sym == "@_EPSILON_SYMBOL_@") net.epsilon = number;
        else if (sym == "@_UNKNOWN_SYMBOL_@") net.unknown = number;
        else if (sym == "@_IDENTITY_SYMBOL_@") net.identity = number;
        else if (sym == "@_TOKEN_SYMBOL_@" || sym == "@_TOKEN_BOUND_@") net.tokenend = number;
        else net.mcs[(size_t)number] = 1;
      } else {  // the newline symbol spans two lines (fomafile.go:423-436)
        std::string_view more;
        if (!next_line(more)) return DTK_E_FORMAT;
        if (more.empty()) net.chars.emplace_back(number, (uint32_t)'\n');
        else net.mcs[(size_t)number] = 1;
      }
    } else if (mode == STATES) {  // fomafile.go:189-369
      if (net.state_count < 0) return DTK_E_FORMAT;
      auto f = split_sp(line, 64);
      if (f[0] == "-1") continue;
      int e[5] = {0, 0, 0, 0, 0};
      const size_t nf = f.size();
      for (size_t i = 0; i < nf && i < 5; i++)
        if (!to_int(f[i], e[i])) return DTK_E_FORMAT;
      if (nf == 5) { state = e[0]; in_sym = e[1]; out_sym = e[2]; end = e[3]; fin = e[4]; }
      else if (nf == 4) {
        if (e[1] == -1) {  // final state without outgoing arcs
          state = e[0]; fin = e[3];
          if (state < 0 || state + 1 > net.state_count) return DTK_E_MODEL;
          if (fin == 1) set_arc(state + 1, net.final_sym, FomaArc{0, false});
          continue;
        }
        state = e[0]; in_sym = out_sym = e[1]; end = e[2]; fin = e[3];
      } else if (nf == 3) { in_sym = e[0]; out_sym = e[1]; end = e[2]; }
      else if (nf == 2) { in_sym = out_sym = e[0]; end = e[1]; }
      const int is = in_sym + 1, os = out_sym + 1;
      bool nontoken = false, tokenend = false;
      if (is != os) {
        if (os == net.tokenend && is == net.epsilon) tokenend = true;  // token boundary, kept under epsilon (fomafile.go:293)
        else if (os == net.epsilon) nontoken = true;
        else return DTK_E_MODEL;  // unsupported transition
      } else if (is == net.tokenend) continue;
      else if (is == net.epsilon) return DTK_E_MODEL;  // general epsilon transitions
      else if (is >= 0 && (size_t)is < net.mcs.size() && net.mcs[(size_t)is]) continue;
      if (state < 0 || state + 1 > net.state_count || end < -1 || end + 1 > net.state_count) return DTK_E_MODEL;
      if (is >= 0) set_arc(state + 1, is, FomaArc{end + 1, nontoken, tokenend});
      if (fin == 1) set_arc(state + 1, net.final_sym, FomaArc{0, false});
    }
  }
  if (net.state_count < 1 || net.epsilon < 1) return DTK_E_FORMAT;
  return DTK_OK;
}

// Automaton.ToMatrix, matrix.go:30-99: header fields and sigma into `m`, the array into `arr`
static int foma_to_matrix(dtk_model *m, const std::vector<uint8_t> &raw, std::vector<uint32_t> &arr) {
  FomaNet net;
  int rc = parse_foma(raw, net);
  if (rc != DTK_OK) return rc;
  int max = net.identity != -1 ? net.identity : 0;
  for (auto &c : net.chars) max = std::max(max, c.first);
  m->kind = DTK_KIND_MATRIX;
  m->epsilon = net.epsilon; m->unknown = net.unknown; m->identity = net.identity;
  m->state_count = (uint32_t)net.state_count;
  m->sigma_count = max + 1;
  m->array_len = ((uint64_t)m->state_count + 1) * (uint64_t)m->sigma_count;
  if (!special_ids_ok(m)) return DTK_E_MODEL;
  for (int i = 0; i < 256; i++) m->ascii[i] = net.identity != -1 ? (uint16_t)net.identity : 0;  // :43-48
  std::vector<std::pair<uint32_t, uint16_t>> ent;
  for (auto &c : net.chars) {
    if (c.second < 256) m->ascii[c.second] = (uint16_t)c.first;
    ent.emplace_back(c.second, (uint16_t)c.first);
  }
  std::stable_sort(ent.begin(), ent.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
  for (auto &e : ent) {
    if (!m->sigma_runes.empty() && m->sigma_runes.back() == e.first) m->sigma_syms.back() = e.second;
    else { m->sigma_runes.push_back(e.first); m->sigma_syms.push_back(e.second); }
  }
  // only what is reachable from state 1 enters the matrix (matrix.go:76-96)
  const uint64_t N = m->state_count;
  arr.assign(m->array_len, 0);
  std::vector<char> seen(N + 2, 0);
  std::vector<uint32_t> todo{1};
  seen[1] = 1;
  while (!todo.empty()) {
    const uint32_t st = todo.back();
    todo.pop_back();
    for (auto &e : net.arcs[st]) {
      const uint64_t at = (uint64_t)(e.first - 1) * N + st;
      if (e.first >= 1 && at < arr.size())
        arr[at] = (uint32_t)e.second.end | (e.second.nontoken ? DTK_FIRSTBIT : 0u);
      const int32_t to = e.second.end;
      if (to >= 1 && (uint64_t)to <= N && !seen[(size_t)to]) { seen[(size_t)to] = 1; todo.push_back((uint32_t)to); }
    }
  }
  return DTK_OK;
}

static int build_foma(dtk_model *m, const std::vector<uint8_t> &raw) {
  std::vector<uint32_t> arr;
  int rc = foma_to_matrix(m, raw, arr);
  return rc != DTK_OK ? rc : layout_matrix(m, arr, m->state_count, false);
}

static void put_rune(std::vector<uint8_t> &o, uint32_t r) {  // bufio.Writer.WriteRune
  if (r > 0x10FFFF || (r >= 0xD800 && r <= 0xDFFF)) r = 0xFFFD;
  if (r < 0x80) o.push_back((uint8_t)r);
  else if (r < 0x800) { o.push_back(0xC0 | (r >> 6)); o.push_back(0x80 | (r & 0x3F)); }
  else if (r < 0x10000) { o.push_back(0xE0 | (r >> 12)); o.push_back(0x80 | ((r >> 6) & 0x3F)); o.push_back(0x80 | (r & 0x3F)); }
  else { o.push_back(0xF0 | (r >> 18)); o.push_back(0x80 | ((r >> 12) & 0x3F)); o.push_back(0x80 | ((r >> 6) & 0x3F)); o.push_back(0x80 | (r & 0x3F)); }
}
}  // namespace

// gzip.NewWriter(f) over a finished image; the caller frees *out
static int gzip_image(const std::vector<uint8_t> &img, void **out, size_t *out_n) {
  z_stream zs;
  memset(&zs, 0, sizeof zs);
  if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, 16 + MAX_WBITS, 8, Z_DEFAULT_STRATEGY) != Z_OK)
    return DTK_E_NOMEM;
  const uLong bound = deflateBound(&zs, (uLong)img.size());
  uint8_t *buf = (uint8_t *)malloc(bound);
  if (!buf) { deflateEnd(&zs); return DTK_E_NOMEM; }
  zs.next_in = const_cast<uint8_t *>(img.data()); zs.avail_in = (uInt)img.size();
  zs.next_out = buf; zs.avail_out = (uInt)bound;
  const int rc = deflate(&zs, Z_FINISH);
  const size_t have = bound - zs.avail_out;
  deflateEnd(&zs);
  if (rc != Z_STREAM_END) { free(buf); return DTK_E_NOMEM; }
  *out = buf; *out_n = have;
  return DTK_OK;
}

// `datok convert -f foma -t file` without --double-array (cmd/datok.go:50-70): LoadFomaFile,
// ToMatrix, then MatrixTokenizer.Save/WriteTo (matrix.go:107-210) into a gzip image.  Host only.
extern "C" int dtk_foma_to_matok(const void *gz_bytes, size_t n, void **out, size_t *out_n) {
  if (!gz_bytes || !out || !out_n) return DTK_E_ARG;
  *out = nullptr; *out_n = 0;
  std::vector<uint8_t> raw;
  int rc = gunzip((const uint8_t *)gz_bytes, n, raw);
  if (rc != DTK_OK) return rc;
  if (raw.size() < 10 || memcmp(raw.data(), "##foma-net", 10) != 0) return DTK_E_FORMAT;
  dtk_model m;
  std::vector<uint32_t> arr;
  rc = foma_to_matrix(&m, raw, arr);
  if (rc != DTK_OK) return rc;
  // WriteTo: the sigma list ends at the largest character symbol (matrix.go:138-153)
  uint32_t max = 0;
  for (auto s : m.sigma_syms) max = std::max<uint32_t>(max, s);
  std::vector<uint32_t> list(max + 1, 0);
  for (size_t i = 0; i < m.sigma_runes.size(); i++) list[m.sigma_syms[i]] = m.sigma_runes[i];
  std::vector<uint8_t> img;
  auto p16 = [&](uint32_t v) { img.push_back((uint8_t)v); img.push_back((uint8_t)(v >> 8)); };
  auto p32 = [&](uint32_t v) { p16(v & 0xFFFF); p16(v >> 16); };
  img.insert(img.end(), {'M', 'A', 'T', 'O', 'K'});
  p16(1); p16((uint32_t)m.epsilon); p16((uint32_t)m.unknown); p16((uint32_t)m.identity);
  p32(m.state_count); p16(max + 1);
  for (uint32_t r : list) put_rune(img, r);
  img.push_back('M');
  for (uint32_t x : arr) p32(x);
  return gzip_image(img, out, out_n);
}

// `datok convert ... --double-array` (cmd/datok.go:50-70): LoadFomaFile, Automaton.ToDoubleArray (datok.go:82-238,
// after Mizobuchi et al. 2000 with the xCheckSkipNiu search, datok.go:385-406), DaTokenizer.WriteTo (datok.go:502-596)
// into a gzip image.  Host only.
// The reference walks each state's outgoing symbols in Go's map order (getSet, fomafile.go:488-495, "sort not
// required") -- random per run, and the order decides which state is laid out next and so where everything after it
// lands: two runs of the reference give different arrays for the same net.  Here the symbols are taken in ascending
// order; the image is one of those the reference can produce, not a particular one.
extern "C" int dtk_foma_to_datok(const void *gz_bytes, size_t n, void **out, size_t *out_n) {
  if (!gz_bytes || !out || !out_n) return DTK_E_ARG;
  *out = nullptr; *out_n = 0;
  std::vector<uint8_t> raw;
  int rc = gunzip((const uint8_t *)gz_bytes, n, raw);
  if (rc != DTK_OK) return rc;
  if (raw.size() < 10 || memcmp(raw.data(), "##foma-net", 10) != 0) return DTK_E_FORMAT;
  FomaNet net;
  rc = parse_foma(raw, net);
  if (rc != DTK_OK) return rc;
  const uint32_t final_ = (uint32_t)net.final_sym;
  struct BC { uint32_t base = 0, check = 0; };
  std::vector<BC> arr;
  auto resize = [&](size_t l) { if (arr.size() <= l) arr.resize(arr.size() + l); };  // datok.go:257-263
  resize(final_);
  int64_t max_size = 0;
  // table: state of the net -> its index in the array, in the order of discovery (datok.go:122-127)
  std::vector<uint32_t> target((size_t)net.state_count + 2, 0);
  std::vector<uint32_t> queue{1};
  target[1] = 1;
  std::vector<int> A;
  for (size_t mark = 0; mark < queue.size(); mark++) {
    const uint32_t s = queue[mark], t = target[s];
    A.clear();
    for (auto &e : net.arcs[s]) A.push_back(e.first);
    std::sort(A.begin(), A.end());
    if (!A.empty() && (A.front() < 1 || (uint32_t)A.back() > final_)) return DTK_E_MODEL;  // (a sigma block behind the states)
    // xCheckSkipNiu, datok.go:385-406
    uint32_t base = 1;
    if (A.size() >= 3) base = (uint32_t)std::fabs((double)(max_size - 1) * .9) + 1;
    for (;;) {
      resize((size_t)base + final_ + 1);
      bool clash = false;
      for (int a : A)
        if ((arr[(size_t)base + (size_t)a].check & DTK_RESTBIT) != 0) { clash = true; break; }
      if (!clash) break;
      base++;
    }
    if (base > DTK_RESTBIT) return DTK_E_MODEL;
    arr[t].base = base;
    for (int a : A) {
      const FomaArc *arc = nullptr;
      for (auto &e : net.arcs[s]) if (e.first == a) arc = &e.second;
      if ((uint32_t)a != final_) {
        const uint32_t s1 = (uint32_t)arc->end, t1 = base + (uint32_t)a;
        arr[t1].check = t;
        if (max_size < (int64_t)t1) max_size = t1;
        if (arc->nontoken) arr[t1].check |= DTK_FIRSTBIT;
        if (arc->tokenend) arr[t1].check |= DTK_SECONDBIT;
        if (s1 < 1 || s1 > (uint32_t)net.state_count) return DTK_E_MODEL;
        if (target[s1] == 0) { target[s1] = t1; queue.push_back(s1); }  // no representative yet: this index is the state
        else arr[t1].base = target[s1] | DTK_FIRSTBIT;                 // separate: points to the representative
      } else {
        arr[(size_t)base + final_].check = t;
        if (max_size < (int64_t)base + final_) max_size = (int64_t)base + final_;  // datok.go:215-218
      }
    }
  }
  // datok.go:224-231: the size in check(1), a little larger than needed so that no lookup has to test a bound
  const size_t len = (size_t)max_size + final_;
  if (arr.size() <= 1) arr.resize(2);
  arr[1].check = (uint32_t)len;
  if (arr.size() < len) arr.resize(arr.size() + final_);
  arr.resize(len);
  // WriteTo: the sigma list ends at the largest character symbol (datok.go:515-528)
  // (dat.sigma maps rune -> symbol: of two symbols for one rune the later one stays, datok.go:104-109)
  std::vector<std::pair<uint32_t, int>> by_rune;
  for (auto &c : net.chars) {
    bool found = false;
    for (auto &e : by_rune) if (e.first == c.second) { e.second = c.first; found = true; }
    if (!found) by_rune.emplace_back(c.second, c.first);
  }
  uint32_t max = 0;
  for (auto &e : by_rune) max = std::max<uint32_t>(max, (uint32_t)e.second);
  std::vector<uint32_t> list(max + 1, 0);
  for (auto &e : by_rune) list[(size_t)e.second] = e.first;
  std::vector<uint8_t> img;
  auto p16 = [&](uint32_t v) { img.push_back((uint8_t)v); img.push_back((uint8_t)(v >> 8)); };
  auto p32 = [&](uint32_t v) { p16(v & 0xFFFF); p16(v >> 16); };
  img.insert(img.end(), {'D', 'A', 'T', 'O', 'K'});
  p16(1); p16((uint32_t)net.epsilon); p16((uint32_t)net.unknown); p16((uint32_t)net.identity);
  p16(final_); p16(max + 1); p32((uint32_t)(arr.size() * 2));
  for (uint32_t r : list) put_rune(img, r);
  img.push_back('T');
  for (auto &bc : arr) { p32(bc.base); p32(bc.check); }
  return gzip_image(img, out, out_n);
}

extern "C" int dtk_model_load_mem(const void *gz_bytes, size_t n, dtk_model **out) {
  if (!gz_bytes || !out) return DTK_E_ARG;
  *out = nullptr;
  if (dtk_device_count() <= 0) return DTK_E_NO_DEVICE;
  std::vector<uint8_t> raw;
  int rc = gunzip((const uint8_t *)gz_bytes, n, raw);
  if (rc != DTK_OK) return rc;
  if (raw.size() < 5) return DTK_E_FORMAT;
  dtk_model *m = new dtk_model();
  if (memcmp(raw.data(), "MATOK", 5) == 0) rc = build_matrix(m, raw);      // fomafile.go:476
  else if (memcmp(raw.data(), "DATOK", 5) == 0) rc = build_datok(m, raw);  // fomafile.go:478
  else if (raw.size() >= 10 && memcmp(raw.data(), "##foma-net", 10) == 0) rc = build_foma(m, raw);
  else rc = DTK_E_FORMAT;                                                  // fomafile.go:482
  if (rc != DTK_OK) { dtk_model_free(m); return rc; }
  *out = m;
  return DTK_OK;
}

extern "C" int dtk_model_load(const char *path, dtk_model **out) {
  if (!path || !out) return DTK_E_ARG;
  *out = nullptr;
  FILE *f = fopen(path, "rb");
  if (!f) return DTK_E_IO;
  std::vector<uint8_t> gz;
  uint8_t buf[1 << 16];
  size_t k;
  while ((k = fread(buf, 1, sizeof buf, f)) > 0) gz.insert(gz.end(), buf, buf + k);
  const bool bad = ferror(f) != 0;
  fclose(f);
  if (bad) return DTK_E_IO;
  return dtk_model_load_mem(gz.data(), gz.size(), out);
}

extern "C" void dtk_model_free(dtk_model *m) {
  if (!m) return;
  if (m->d_tab) (void)hipFree(m->d_tab);
  if (m->d_ascii) (void)hipFree(m->d_ascii);
  if (m->d_runes) (void)hipFree(m->d_runes);
  if (m->d_syms) (void)hipFree(m->d_syms);
  if (m->d_codes) (void)hipFree(m->d_codes);
  delete m;
}

extern "C" const char *dtk_model_type(const dtk_model *m) {
  return m->kind == DTK_KIND_MATRIX ? "MATOK" : "DATOK";
}

extern "C" int dtk_model_get_info(const dtk_model *m, dtk_model_info *o) {
  if (!m || !o) return DTK_E_ARG;
  o->kind = m->kind; o->epsilon = m->epsilon; o->unknown = m->unknown; o->identity = m->identity;
  o->final_state = m->final_state; o->sigma_count = m->sigma_count; o->state_count = m->state_count;
  o->array_len = m->array_len; o->n_eps_states = m->n_eps_states; o->max_eps_chain = m->max_eps_chain;
  o->entry_bytes = m->tab.entry_bytes; o->device_bytes = m->device_bytes; o->unknown_used = m->unknown_used;
  o->dense_states = m->dense_states;
  o->stream_codes = m->sig.n_codes;
  return DTK_OK;
}

// -------------------------------------------------------------------- batch

struct dtk_batch {
  int device = 0;
  hipStream_t stream = nullptr;
  uint64_t max_bytes = 0;
  uint32_t max_docs = 0;
  // inputs
  uint8_t *d_text_own = nullptr;
  uint64_t *d_off_own = nullptr;
  const uint8_t *d_text = nullptr;
  const uint64_t *d_off = nullptr;
  uint32_t n_docs = 0;
  uint64_t total = 0;
  // intermediates
  uint16_t *d_sym = nullptr;
  uint32_t *d_rsbits = nullptr;                // rune-start bitmap of the input (1 bit per byte)
  uint32_t *d_bits = nullptr;                  // event bitmaps of the walk (EVB_KINDS kinds), cleared every run
  uint32_t bit_words = 0;                      // words per kind of the current input
  uint32_t *d_doc_tail = nullptr;              // per document: final SentenceEnd / TextEnd (carved from d_acc)
  uint8_t *d_acc = nullptr;                    // per-document accumulators + totals (one memset)
  uint64_t acc_bytes = 0;
  uint32_t *d_status = nullptr;
  // speculative chunk lanes
  std::vector<uint64_t> h_doc_off;   // host copy of the document offsets (lane planning)
  uint32_t cfg_chunk = 0xFFFFFFFFu;  // 0 = one lane per document, 0xFFFFFFFF = automatic
  uint32_t cfg_extend = 240;         // move the warm-up start back to the previous blank, at most this far
  uint32_t cfg_warm = 8;             // (16 until round 3: 8 costs no repair round on any corpus and 5 % fewer lookups) plus the way back to the previous blank (cfg_extend); a miss only costs a repair round
  uint32_t chunk = 0;                // chunk size of the current plan (0 = none)
  bool plan_valid = false;
  uint32_t n_lanes = 0, lane_cap = 0;
  uint32_t *d_lane_doc = nullptr, *d_chunk_off = nullptr, *d_redo = nullptr;
  // long documents are compacted in segments of DTK_SEG_LANES lanes (tables built with the lane plan)
  uint32_t *d_seg_tab = nullptr;     // seg_doc | seg_lane0 | seg_nl | doc_seg0
  DtkSegSum *d_seg_sum = nullptr;
  DtkSegIn *d_seg_in = nullptr;
  uint32_t n_segs = 0, seg_cap = 0;
  bool long_docs = false;            // some document has more than one segment
  uint32_t max_doc_lanes = 0;        // lanes of the longest document (bounds the repair rounds)
  uint32_t *d_blk_doc = nullptr;     // document of the first byte of every 4 KiB input block
  // compaction: documents of at most small_max bytes go one per lane (k_compact_small), the others one per wave
  uint32_t small_max = 0, n_big = 0;
  uint32_t *d_big_docs = nullptr;    // ids of the documents above small_max
  uint32_t *d_first_bad = nullptr, *d_fail_lane = nullptr;
  DtkLaneCount *d_lane_cnt = nullptr;
  DtkLaneState *d_lane_start = nullptr, *d_lane_end = nullptr;
  DtkLanePlan *d_lane_plan = nullptr;
  uint32_t repair_rounds = 0;        // of the last run
  const dtk_model *last_model = nullptr;
  uint32_t last_flags = 0;
  uint64_t *d_csr = nullptr;  // tok_off | sent_off | text_off
  uint64_t *d_tok_off = nullptr, *d_sent_off = nullptr, *d_text_off = nullptr;
  uint64_t *d_tok_cnt = nullptr, *d_sent_cnt = nullptr, *d_text_cnt = nullptr;  // per-document counts
  uint64_t *d_scan_ws = nullptr;  // tile sums of the multi-block scan (many documents)
  uint64_t *d_totals = nullptr;  // [0..3] scan totals, [4] walk steps, [6] invalid UTF-8 bytes, [7] irregular flag,
                                 // [8..9] as u32[4]: documents to repair after the first pass / after each device-side round
  uint32_t dev_rounds = 0;       // repair rounds enqueued ahead of time in the last run
  bool expect_repairs = false;   // the last run needed repairs: enqueue rounds ahead of time in the next one
  uint32_t round_limit = 0xFFFFFFFFu;  // repair rounds from the host before the one-lane-per-document fallback (DATOK_ROUND_LIMIT)
  bool acc_primed = false;       // the accumulator block has been cleared whole once (k_symbolize clears it from then on)
  uint64_t epoch = 0;            // number of the run (k_symbolize marks runs that saw invalid UTF-8 with it)
  bool expect_eot = false;       // the last run had documents with EOT calls: launch their compaction kernel with the run
  bool ran_full = false;         // that kernel has run since the last dtk_batch_run
  uint64_t *h_totals = nullptr;  // pinned
  uint64_t *h_off_pin = nullptr; // pinned staging of the document offsets (a copy from pageable memory would block until
                                 // the text copy in front of it has finished: 0.7 ms per 16 MiB batch)
  // outputs (grown on demand, never inside a run unless a re-launch is needed)
  uint64_t tok_cap = 0, sent_cap = 0, text_cap = 0;
  int32_t *d_rstart = nullptr, *d_rend = nullptr, *d_sent = nullptr;
  uint32_t *d_bstart = nullptr, *d_bend = nullptr, *d_ttok = nullptr, *d_tsent = nullptr;
  uint32_t *d_sbefore = nullptr, *d_ts_end = nullptr, *d_doc_ns = nullptr;  // renderer inputs (compact)
  // device rendering of the writer output (dtk_batch_render): workspace + output, grown on demand
  uint64_t *d_rws = nullptr;  uint64_t rws_cap = 0;   // scans, tile sums, per-text regions (u64 words)
  uint64_t *d_out_off = nullptr;
  uint8_t *d_out = nullptr;   uint64_t out_cap = 0;
  uint64_t out_total = 0;
  uint64_t n_invalid = 0;     // nonzero: the last run saw invalid UTF-8 (each such byte prints as U+FFFD, 3 bytes)
  uint32_t render_flags = 0xFFFFFFFFu;  // flags of the rendering held in d_out (none)
  std::vector<uint8_t> h_out;
  std::vector<uint64_t> h_out_off;
  // the exact pass over ST_IRREGULAR documents (normally none): ids, call counts / offsets, calls
  uint32_t *d_exact_ids = nullptr, *d_exact_cnt = nullptr;
  uint64_t *d_exact_off = nullptr;
  DtkCall *d_calls = nullptr;
  uint32_t exact_cap = 0;
  uint64_t calls_cap = 0;
  std::vector<uint32_t> h_exact_ids;
  std::vector<uint64_t> h_exact_off;
  std::vector<DtkCall> h_calls;
  // optional stage timing
  bool profiling = false;
  hipEvent_t ev[DTK_N_STAGES + 1] = {};
  // last run
  bool ran = false, totals_valid = false;
  uint32_t exp_runs = 0;  // DTK_EXP_SKIP experiments only
  DtkCompactArgs last_args{};
  dtk_totals totals{};
  // Results on the host (dtk_batch_result_host): page-locked buffers owned by the batch, filled by one chain of
  // asynchronous copies on a stream of their own (dl_stream) -- the batch's own stream is free for the next kernels,
  // the copy engine for the next slice's upload (PCIe is full duplex).  `fields` (DTK_R_*) selects what is copied.
  enum { PB_TOK_OFF, PB_SENT_OFF, PB_TEXT_OFF, PB_RSTART, PB_REND, PB_BSTART, PB_BEND, PB_SENT, PB_TTOK, PB_TSENT,
         PB_STATUS, PB_BITS, PB_TAIL, PB_R16, PB_N };
  struct PinBuf { void *p = nullptr; size_t cap = 0; } pin[PB_N];
  PinBuf h_plan;            // staging of the lane plan's tables (plan_lanes)
  uint32_t fields = DTK_R_ALL;
  uint32_t *d_r16 = nullptr;      // DTK_R_TOK_RUNE16: the packed rune offsets (filled on the download stream)
  uint64_t r16_cap = 0;
  uint64_t max_doc_bytes = 0;     // of the current input (what decides whether the narrow form exists)
  bool max_doc_valid = false;
  hipStream_t dl_stream = nullptr;  // created with the first download, unless the caller lends one (a pipeline's slices share one:
  bool dl_own = false;              //  the runtime maps streams onto four hardware queues, and streams that share a queue serialise)
  hipEvent_t ev_ran = nullptr;      // behind the last launch of dtk_batch_run (dtk_batch_done)
  bool ev_ran_valid = false;
  // Lent streams (dtk_batch_set_streams): the batches of a pipeline share one stream for their kernels and one for their
  // uploads -- the runtime has four hardware queues, and a pipeline of any depth then needs three (kernels, uploads,
  // downloads).  The upload's end is an event the kernels wait for.
  bool stream_own = true;
  hipStream_t up_stream = nullptr;  // null: uploads run on `stream`
  hipEvent_t ev_up = nullptr;
  bool up_pending = false;          // an upload on up_stream has not been waited for yet
  uint32_t eager_fields = 0;  // the last run's k_to_host was asked for these (0: none); finish() decides whether it counts
  bool results_changed = false;  // finish() had to touch the result arrays after the run (repair, growth, EOT kernel, exact pass)
  hipEvent_t ev_dl = nullptr;  // behind the batch's copies on the (possibly shared) download stream
  bool dl_waited = true;
  bool dl_begun = false;    // the copies of the last run's results have been enqueued
  uint32_t dl_fields = 0;   // ... these fields
};

// page-locked memory for n bytes in `pb` (grown with a quarter of slack: allocation costs milliseconds)
static int pin_fit(dtk_batch::PinBuf &pb, size_t n) {
  if (n <= pb.cap && pb.p) return DTK_OK;
  if (pb.p) HIP_TRY(hipHostFree(pb.p));
  pb.p = nullptr; pb.cap = 0;
  const size_t cap = std::max<size_t>(n + n / 4, 256);
  HIP_TRY(hipHostMalloc(&pb.p, cap, hipHostMallocDefault));
  pb.cap = cap;
  return DTK_OK;
}

static int alloc_outputs(dtk_batch *b, uint64_t tok, uint64_t sent, uint64_t text) {
  auto grow = [&](auto *&p, uint64_t n) -> int {
    if (p) HIP_TRY(hipFree(p));
    p = nullptr;
    HIP_TRY(hipMalloc((void **)&p, std::max<uint64_t>(n, 4) * 4));
    return DTK_OK;
  };
  int rc;
  if (tok > b->tok_cap) {
    if ((rc = grow(b->d_rstart, tok))) return rc;
    if ((rc = grow(b->d_rend, tok))) return rc;
    if ((rc = grow(b->d_bstart, tok))) return rc;
    if ((rc = grow(b->d_bend, tok))) return rc;
    if ((rc = grow(b->d_sbefore, tok))) return rc;
    b->tok_cap = tok;
  }
  if (sent > b->sent_cap) {
    if ((rc = grow(b->d_sent, sent))) return rc;
    b->sent_cap = sent;
  }
  if (text > b->text_cap) {
    if ((rc = grow(b->d_ttok, text))) return rc;
    if ((rc = grow(b->d_tsent, text))) return rc;
    if ((rc = grow(b->d_ts_end, text))) return rc;
    b->text_cap = text;
  }
  return DTK_OK;
}

extern "C" int dtk_batch_create(uint64_t max_bytes, uint32_t max_docs, dtk_batch **out) {
  if (!out || max_docs == 0) return DTK_E_ARG;
  *out = nullptr;
  if (dtk_device_count() <= 0) return DTK_E_NO_DEVICE;
  dtk_batch *b = new dtk_batch();
  b->max_bytes = max_bytes;
  b->max_docs = max_docs;
  auto fail = [&](int rc) { dtk_batch_free(b); return rc; };
#define B_TRY(call)                                                  \
  do {                                                               \
    hipError_t e_ = (call);                                          \
    if (e_ != hipSuccess) return fail(hip_fail(e_, #call));          \
  } while (0)
  B_TRY(hipGetDevice(&b->device));
  B_TRY(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
  if (g_dbg.round_limit >= 0) b->round_limit = (uint32_t)g_dbg.round_limit;
  const uint64_t pad = 256;
  B_TRY(hipMalloc((void **)&b->d_text_own, max_bytes + pad));
  B_TRY(hipMalloc((void **)&b->d_off_own, ((uint64_t)max_docs + 1) * 8));
  B_TRY(hipMalloc((void **)&b->d_sym, (max_bytes + pad) * 2));
  B_TRY(hipMalloc((void **)&b->d_rsbits, (max_bytes + pad) / 8 + 64));
  // per array: total + 4 * n_docs + 4 slots, rounded up to 256 by dtk_batch_run
  // one bit per cursor position and kind: total + n_docs positions, rounded up to 16 bytes per kind, two words of slack
  B_TRY(hipMalloc((void **)&b->d_bits, (EVB_KINDS * ((max_bytes + max_docs) / 32 + 8) + 8) * 4));
  b->acc_bytes = DTK_TOTALS_BYTES + 3 * ((uint64_t)max_docs + 1) * 8 + 4 * (uint64_t)max_docs * 4 + 64;
  B_TRY(hipMalloc((void **)&b->d_acc, b->acc_bytes));
  B_TRY(hipMalloc((void **)&b->d_redo, (uint64_t)max_docs * 4));
  B_TRY(hipMalloc((void **)&b->d_blk_doc, (max_bytes / DTK_SYM_BLOCK_BYTES + 3) * 4));

  B_TRY(hipMalloc((void **)&b->d_chunk_off, ((uint64_t)max_docs + 1) * 4));
  B_TRY(hipMalloc((void **)&b->d_big_docs, ((uint64_t)max_docs + 1) * 4));
  // the three row-offset arrays, carved from one block per run (n_docs + 1 words each, back to back: one copy brings
  // them to the host)
  B_TRY(hipMalloc((void **)&b->d_csr, 3 * ((uint64_t)max_docs + 1) * 8));
  b->d_tok_off = b->d_csr; b->d_sent_off = b->d_csr + ((uint64_t)max_docs + 1); b->d_text_off = b->d_csr + 2 * ((uint64_t)max_docs + 1);
  B_TRY(hipMalloc((void **)&b->d_doc_ns, ((uint64_t)max_docs + 1) * 4));
  B_TRY(hipMalloc((void **)&b->d_scan_ws, ((uint64_t)max_docs / 2048 + 2) * 4 * 8));
  B_TRY(hipMalloc((void **)&b->d_out_off, ((uint64_t)max_docs + 1) * 8));

  // [0..15] device totals ([10] doubles as the render size), [16..] the striped lookup counters
  B_TRY(hipHostMalloc((void **)&b->h_totals, DTK_TOTALS_BYTES, hipHostMallocDefault));
  B_TRY(hipHostMalloc((void **)&b->h_off_pin, ((uint64_t)max_docs + 1) * 8, hipHostMallocDefault));
#undef B_TRY
  // typical German: 0.18 tokens and 0.06 sentence ints per byte; grown on demand
  int rc = alloc_outputs(b, max_bytes / 3 + max_docs + 16, max_bytes / 8 + 2ull * max_docs + 16,
                         max_bytes / 64 + 2ull * max_docs + 16);
  if (rc != DTK_OK) return fail(rc);
  *out = b;
  return DTK_OK;
}

extern "C" void dtk_batch_free(dtk_batch *b) {
  if (!b) return;
  if (b->stream) (void)hipStreamSynchronize(b->stream);
  if (b->up_stream) (void)hipStreamSynchronize(b->up_stream);
  if (b->ev_up) (void)hipEventDestroy(b->ev_up);
  if (b->dl_begun && !b->dl_waited) (void)hipEventSynchronize(b->ev_dl);
  if (b->dl_stream && b->dl_own) (void)hipStreamDestroy(b->dl_stream);
  if (b->ev_ran) (void)hipEventDestroy(b->ev_ran);
  if (b->ev_dl) (void)hipEventDestroy(b->ev_dl);
  for (auto &pb : b->pin)
    if (pb.p) (void)hipHostFree(pb.p);
  if (b->h_plan.p) (void)hipHostFree(b->h_plan.p);
  void *ptrs[] = {b->d_text_own, b->d_off_own, b->d_sym, b->d_rsbits, b->d_bits, b->d_acc, b->d_redo, b->d_chunk_off, b->d_blk_doc, b->d_big_docs,
                  b->d_lane_doc, b->d_lane_cnt, b->d_lane_start, b->d_lane_end, b->d_lane_plan,
                  b->d_seg_tab, b->d_seg_sum, b->d_seg_in,
                  b->d_csr, b->d_rstart, b->d_rend, b->d_sent,
                  b->d_bstart, b->d_bend, b->d_ttok, b->d_tsent,
                  b->d_sbefore, b->d_ts_end, b->d_doc_ns, b->d_scan_ws, b->d_rws, b->d_out_off, b->d_out,
                  b->d_exact_ids, b->d_exact_cnt, b->d_exact_off, b->d_calls, b->d_r16};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  if (b->h_totals) (void)hipHostFree(b->h_totals);
  if (b->h_off_pin) (void)hipHostFree(b->h_off_pin);
  for (hipEvent_t e : b->ev)
    if (e) (void)hipEventDestroy(e);
  if (b->stream && b->stream_own) (void)hipStreamDestroy(b->stream);
  delete b;
}

extern "C" void *dtk_batch_stream(dtk_batch *b) { return b ? (void *)b->stream : nullptr; }

// Everything this batch has enqueued so far has finished.  With a stream of its own: the stream; on a lent stream
// (shared with other batches) only the batch's own last run, by its event.
static int wait_ran(dtk_batch *b) {  // the kernels of the batch's last run (not an upload on the lent upload stream)
  if (b->stream_own) HIP_TRY(hipStreamSynchronize(b->stream));
  else if (b->ev_ran_valid) HIP_TRY(hipEventSynchronize(b->ev_ran));
  return DTK_OK;
}
static int wait_own(dtk_batch *b) {
  if (b->up_pending) { HIP_TRY(hipEventSynchronize(b->ev_up)); }
  return wait_ran(b);
}

extern "C" int dtk_batch_set_streams(dtk_batch *b, void *compute, void *upload) {
  if (!b) return DTK_E_ARG;
  int rc = wait_own(b);
  if (rc != DTK_OK) return rc;
  if (compute) {
    if (b->stream_own && b->stream) HIP_TRY(hipStreamDestroy(b->stream));
    b->stream = (hipStream_t)compute;
    b->stream_own = false;
  }
  b->up_stream = (hipStream_t)upload;
  if (upload && !b->ev_up) HIP_TRY(hipEventCreateWithFlags(&b->ev_up, hipEventDisableTiming));
  return DTK_OK;
}

extern "C" int dtk_batch_set_input(dtk_batch *b, const uint8_t *text, const uint64_t *doc_off, uint32_t n_docs) {
  if (!b || !doc_off || n_docs == 0) return DTK_E_ARG;
  if (n_docs > b->max_docs) return DTK_E_CAPACITY;
  if (doc_off[0] != 0) return DTK_E_ARG;
  for (uint32_t d = 0; d < n_docs; d++) {
    if (doc_off[d + 1] < doc_off[d]) return DTK_E_ARG;
    if (doc_off[d + 1] - doc_off[d] >= 0x7FFFFFF0ull) return DTK_E_ARG;  // 31-bit cursor positions
  }
  const uint64_t total = doc_off[n_docs];
  if (total > b->max_bytes || total + n_docs + 64 >= (1ull << 32)) return DTK_E_CAPACITY;  // 32-bit position bits
  if (total && !text) return DTK_E_ARG;
  { int rc = wait_own(b); if (rc != DTK_OK) return rc; }  // the previous run may still read the buffers
  hipStream_t us = b->up_stream ? b->up_stream : b->stream;
  if (total) HIP_TRY(hipMemcpyAsync(b->d_text_own, text, total, hipMemcpyHostToDevice, us));
  memcpy(b->h_off_pin, doc_off, ((size_t)n_docs + 1) * 8);
  HIP_TRY(hipMemcpyAsync(b->d_off_own, b->h_off_pin, ((uint64_t)n_docs + 1) * 8, hipMemcpyHostToDevice, us));
  if (b->up_stream) { HIP_TRY(hipEventRecord(b->ev_up, us)); b->up_pending = true; }
  // the lane plan only depends on the offsets: a stream of equally shaped batches keeps it
  const bool same = b->plan_valid && b->d_off == b->d_off_own && b->n_docs == n_docs &&
                    b->h_doc_off.size() == (size_t)n_docs + 1 &&
                    memcmp(b->h_doc_off.data(), doc_off, ((size_t)n_docs + 1) * 8) == 0;
  b->d_text = b->d_text_own;
  b->d_off = b->d_off_own;
  b->n_docs = n_docs;
  b->total = total;
  b->ran = false;
  if (!same) {
    b->h_doc_off.assign(doc_off, doc_off + n_docs + 1);
    b->plan_valid = false;
    b->max_doc_valid = false;
  }
  return DTK_OK;
}

extern "C" int dtk_batch_set_input_device(dtk_batch *b, const void *d_text, const void *d_doc_off,
                                          uint32_t n_docs, uint64_t total_bytes) {
  if (!b || !d_doc_off || n_docs == 0 || (total_bytes && !d_text)) return DTK_E_ARG;
  if (n_docs > b->max_docs || total_bytes > b->max_bytes || total_bytes + n_docs + 64 >= (1ull << 32)) return DTK_E_CAPACITY;
  b->d_text = (const uint8_t *)d_text;
  b->d_off = (const uint64_t *)d_doc_off;
  b->n_docs = n_docs;
  b->total = total_bytes;
  b->ran = false;
  // lane planning needs the offsets on the host: one copy per input, not per run
  b->h_doc_off.resize((size_t)n_docs + 1);
  HIP_TRY(hipMemcpy(b->h_doc_off.data(), d_doc_off, ((size_t)n_docs + 1) * 8, hipMemcpyDeviceToHost));
  if (b->h_doc_off[0] != 0 || b->h_doc_off[n_docs] != total_bytes) return DTK_E_ARG;
  for (uint32_t d = 0; d < n_docs; d++)
    if (b->h_doc_off[d + 1] < b->h_doc_off[d] || b->h_doc_off[d + 1] - b->h_doc_off[d] >= 0x7FFFFFF0ull)
      return DTK_E_ARG;
  b->plan_valid = false;
  b->max_doc_valid = false;
  return DTK_OK;
}

extern "C" int dtk_batch_set_chunking(dtk_batch *b, uint32_t chunk_bytes, uint32_t warm_bytes) {
  if (!b) return DTK_E_ARG;
  if (chunk_bytes != 0 && chunk_bytes != 0xFFFFFFFFu && chunk_bytes < 16) return DTK_E_ARG;
  b->cfg_chunk = chunk_bytes;
  b->cfg_warm = warm_bytes;
  b->plan_valid = false;
  return DTK_OK;
}

extern "C" int dtk_batch_set_warm_extend(dtk_batch *b, uint32_t max_bytes) {
  if (!b) return DTK_E_ARG;
  b->cfg_extend = max_bytes > 4096u ? 4096u : max_bytes;
  return DTK_OK;
}

// Splits the documents into chunk lanes (host side of the speculative walk).
static int plan_lanes(dtk_batch *b) {
  if (b->plan_valid) return DTK_OK;
  // The tables go to the device as asynchronous copies from one page-locked staging buffer, on the stream the input
  // was uploaded on: a pipeline of ragged slices (every slice another plan) used to wait here for the slice's upload
  // and then for eight synchronous copies, one after the other.
  { const int rc_ = wait_ran(b); if (rc_ != DTK_OK) return rc_; }  // (the previous run read the old tables)
  std::vector<uint8_t> stage;
  struct Put { void *dst; size_t at, n; };
  std::vector<Put> puts;
  auto put = [&](void *dst, const void *src, size_t n) {
    if (!n) return;
    const size_t at = (stage.size() + 15) & ~(size_t)15;
    stage.resize(at + n);
    memcpy(stage.data() + at, src, n);
    puts.push_back(Put{dst, at, n});
  };
  {
    // document of the first byte of every symbolise block (+ one entry behind the end)
    const uint64_t nblk = (b->total + DTK_SYM_BLOCK_BYTES - 1) / DTK_SYM_BLOCK_BYTES;
    std::vector<uint32_t> blk((size_t)nblk + 2);
    uint32_t d = 0;
    for (uint64_t i = 0; i <= nblk; i++) {
      const uint64_t g = i * DTK_SYM_BLOCK_BYTES;
      while (d + 1 < b->n_docs && b->h_doc_off[d + 1] <= g) d++;
      blk[i] = d;
    }
    blk[nblk + 1] = b->n_docs - 1;
    put(b->d_blk_doc, blk.data(), blk.size() * 4);
  }
  {
    // One lane per document pays for many tiny documents (tweets, single sentences): 64-byte documents compact five
    // times faster that way.  From 256 bytes on the lanes' scattered row stores cost more than a wave per document
    // (measured: 256 B 245 -> 274 us, 1 KiB 81 -> 282 us per 32 MiB), hence the low limit.
    const bool e_sm = g_dbg.small_max >= 0;
    uint32_t sm = b->n_docs >= 2048u ? 160u : 0u;
    if (e_sm) sm = (uint32_t)g_dbg.small_max;
    std::vector<uint32_t> big;
    if (sm)
      for (uint32_t d = 0; d < b->n_docs; d++)
        if (b->h_doc_off[d + 1] - b->h_doc_off[d] > sm) big.push_back(d);
    // (the lane kernel's time is a fixed 30 us of latency -- a lane's sequential loop over its document -- which only
    //  pays once it replaces 16 384 waves or more; 8192 Zipf-length documents: compaction 68 -> 98 us with it)
    if (!e_sm && b->n_docs - big.size() < 16384u) { sm = 0; big.clear(); }
    b->small_max = sm;
    b->n_big = (uint32_t)big.size();
    put(b->d_big_docs, big.data(), big.size() * 4);
  }
  uint32_t C = b->cfg_chunk;
  if (C == 0xFFFFFFFFu) {
    // enough lanes to give every SIMD of the chip a few waves, but chunks no shorter than
    // a few warm-ups: 256 CUs x 4 SIMDs x 64 lanes = 65536 lanes per "wave per SIMD"
    // measured: 128 is best for 16 MiB (131072 lanes = 2 waves per SIMD); large batches are
    // flat from 128 to 1024 and lose beyond (fewer lanes than the chip holds)
    const uint64_t want_lanes = 4ull * 65536ull;
    uint64_t c = b->total / want_lanes;
    uint32_t p2 = 128;
    while (p2 * 2 <= c && p2 < DTK_LDS_BIT_CHUNK_MAX) p2 <<= 1;  // (up to what a wave's LDS bitmaps cover)
    C = p2;
  }
  b->chunk = C;
  auto flush = [&]() -> int {
    int rc_ = pin_fit(b->h_plan, stage.size());
    if (rc_ != DTK_OK) return rc_;
    memcpy(b->h_plan.p, stage.data(), stage.size());
    hipStream_t us = b->up_stream ? b->up_stream : b->stream;
    for (const Put &q : puts)
      HIP_TRY(hipMemcpyAsync(q.dst, (const uint8_t *)b->h_plan.p + q.at, q.n, hipMemcpyHostToDevice, us));
    if (b->up_stream) { HIP_TRY(hipEventRecord(b->ev_up, us)); b->up_pending = true; }  // (the kernels wait for this too)
    return DTK_OK;
  };
  if (C == 0) {
    b->n_lanes = 0;
    int rc_ = flush();
    if (rc_ != DTK_OK) return rc_;
    b->plan_valid = true;
    return DTK_OK;
  }
  const uint32_t nd = b->n_docs;
  std::vector<uint32_t> chunk_off((size_t)nd + 1);
  uint64_t lanes = 0;
  for (uint32_t d = 0; d < nd; d++) {
    chunk_off[d] = (uint32_t)lanes;
    const uint64_t len = b->h_doc_off[d + 1] - b->h_doc_off[d];
    lanes += len ? (len + C - 1) / C : 1;
    if (lanes >= 0x7FFFFFFFull) return DTK_E_CAPACITY;
  }
  chunk_off[nd] = (uint32_t)lanes;
  std::vector<uint32_t> lane_doc((size_t)lanes);
  for (uint32_t d = 0; d < nd; d++)
    for (uint32_t L = chunk_off[d]; L < chunk_off[d + 1]; L++) lane_doc[L] = d;
  if (lanes > b->lane_cap) {
    void *old[] = {b->d_lane_doc, b->d_lane_cnt, b->d_lane_start, b->d_lane_end, b->d_lane_plan};
    for (void *p : old)
      if (p) HIP_TRY(hipFree(p));
    b->d_lane_doc = nullptr;
    b->d_lane_cnt = nullptr;
    b->d_lane_start = b->d_lane_end = nullptr;
    b->d_lane_plan = nullptr;
    const uint64_t cap = lanes + lanes / 8 + 64;
    HIP_TRY(hipMalloc((void **)&b->d_lane_doc, cap * 4));
    HIP_TRY(hipMalloc((void **)&b->d_lane_cnt, cap * sizeof(DtkLaneCount)));
    HIP_TRY(hipMalloc((void **)&b->d_lane_start, cap * sizeof(DtkLaneState)));
    HIP_TRY(hipMalloc((void **)&b->d_lane_end, cap * sizeof(DtkLaneState)));
    HIP_TRY(hipMalloc((void **)&b->d_lane_plan, cap * sizeof(DtkLanePlan)));
    b->lane_cap = (uint32_t)cap;
  }
  // segments of DTK_SEG_LANES lanes: the unit of k_compact for documents with many lanes
  std::vector<uint32_t> seg_doc, seg_lane0, seg_nl, doc_seg0((size_t)nd + 1);
  b->long_docs = false;
  b->max_doc_lanes = 0;
  for (uint32_t d = 0; d < nd; d++) {
    doc_seg0[d] = (uint32_t)seg_doc.size();
    const uint32_t L0 = chunk_off[d], L1 = chunk_off[d + 1];
    if (L1 - L0 > DTK_SEG_LANES) b->long_docs = true;
    b->max_doc_lanes = std::max(b->max_doc_lanes, L1 - L0);
    for (uint32_t L = L0; L < L1; L += DTK_SEG_LANES) {
      seg_doc.push_back(d);
      seg_lane0.push_back(L);
      seg_nl.push_back(std::min<uint32_t>(DTK_SEG_LANES, L1 - L));
    }
  }
  doc_seg0[nd] = (uint32_t)seg_doc.size();
  const uint32_t ns = (uint32_t)seg_doc.size();
  if (ns > b->seg_cap) {
    void *old[] = {b->d_seg_tab, b->d_seg_sum, b->d_seg_in};
    for (void *p : old)
      if (p) HIP_TRY(hipFree(p));
    b->d_seg_tab = nullptr; b->d_seg_sum = nullptr; b->d_seg_in = nullptr;
    const uint64_t cap = (uint64_t)ns + ns / 8 + 64;
    HIP_TRY(hipMalloc((void **)&b->d_seg_tab, (3 * cap + 2 * ((uint64_t)b->max_docs + 1)) * 4));
    HIP_TRY(hipMalloc((void **)&b->d_seg_sum, cap * sizeof(DtkSegSum)));
    HIP_TRY(hipMalloc((void **)&b->d_seg_in, cap * sizeof(DtkSegIn)));
    b->seg_cap = (uint32_t)cap;
  }
  b->n_segs = ns;
  put(b->d_seg_tab, seg_doc.data(), (size_t)ns * 4);
  put(b->d_seg_tab + b->seg_cap, seg_lane0.data(), (size_t)ns * 4);
  put(b->d_seg_tab + 2 * (size_t)b->seg_cap, seg_nl.data(), (size_t)ns * 4);
  put(b->d_seg_tab + 3 * (size_t)b->seg_cap, doc_seg0.data(), ((size_t)nd + 1) * 4);
  put(b->d_lane_doc, lane_doc.data(), lanes * 4);
  put(b->d_chunk_off, chunk_off.data(), ((size_t)nd + 1) * 4);
  { const int rc_ = flush(); if (rc_ != DTK_OK) return rc_; }
  b->n_lanes = (uint32_t)lanes;
  b->plan_valid = true;
  return DTK_OK;
}

// the batch's symbol stream as its last run wrote it: codes + the model's table, or 16-bit entries
static DtkSym sym_of(const dtk_batch *b) {
  const dtk_model *m = b->last_model;
  return DtkSym{b->d_sym, (m && m->sig.n_codes) ? m->sig.code_entry : nullptr};
}

static DtkWalkArgs walk_args(dtk_batch *b) {
  DtkWalkArgs w{};
  w.sym = sym_of(b); w.doc_off = b->d_off; w.n_docs = b->n_docs;
  w.bits = b->d_bits; w.bit_words = b->bit_words; w.doc_tail = b->d_doc_tail; w.status = b->d_status;
  w.tok_cnt = b->d_tok_cnt; w.sent_cnt = b->d_sent_cnt; w.text_cnt = b->d_text_cnt;
  w.steps = (unsigned long long *)(b->d_totals + 16);
  w.step_factor = 2048;  // look-ahead is bounded by the 1024-rune window (matrix.go:365)
  return w;
}

static DtkSpecArgs spec_args(dtk_batch *b, bool redo) {
  DtkSpecArgs s{};
  s.n_lanes = b->n_lanes; s.chunk = b->chunk; s.warm = b->cfg_warm;
  s.lane_doc = b->d_lane_doc; s.chunk_off = b->d_chunk_off;
  s.lane_start = b->d_lane_start; s.lane_end = b->d_lane_end; s.lane_plan = b->d_lane_plan;
  s.lane_cnt = b->d_lane_cnt; s.first_bad = b->d_first_bad; s.fail_lane = b->d_fail_lane;
  s.redo_from = redo ? b->d_redo : nullptr;
  s.text = b->d_text;
  {
    s.warm_ws = (uint32_t)g_dbg.warm_ws;
    s.warm_min = (uint32_t)g_dbg.warm_min;
    s.warm_extend = g_dbg.warm_extend >= 0 ? (uint32_t)g_dbg.warm_extend : b->cfg_extend;
    // the walk collects its event bits in LDS, one set of bitmaps per wave (test hook LDS_BITS=0: straight to memory)
    const bool lds = g_dbg.lds_bits != 0;
    s.lds_words = (lds && b->chunk <= DTK_LDS_BIT_CHUNK_MAX) ? DTK_LDS_BIT_WORDS(b->chunk) : 0u;
  }
  return s;
}

static uint32_t cmp_mask_of(const dtk_model *m) {
  // the sticky `ok` only matters where an arc on the unknown symbol exists (matrix.go:478-485)
  return LANE_F_SENT | LANE_F_TEXT | (m->unknown_used ? LANE_F_OK : 0u);
}

// which: 1 the documents without an EOT call, 2 those with one, 3 both (dtk_launch_compact)
static int launch_compact2(dtk_batch *b, int which) {
  DtkCompactArgs a = b->last_args;
  const bool no_rune = (b->last_flags & DTK_NO_RUNE_OFFSETS) != 0, no_byte = (b->last_flags & DTK_NO_BYTE_OFFSETS) != 0;
  a.tok_rstart = no_rune ? nullptr : b->d_rstart; a.tok_rend = no_rune ? nullptr : b->d_rend;
  a.tok_bstart = no_byte ? nullptr : b->d_bstart; a.tok_bend = no_byte ? nullptr : b->d_bend;
  a.sent = b->d_sent; a.text_tok_end = b->d_ttok; a.text_sent_end = b->d_tsent;
  const bool ro = (b->last_flags & (DTK_OFFSETS_ONLY | DTK_NO_RUNE_OFFSETS | DTK_NO_BYTE_OFFSETS)) != 0;  // no renderer bookkeeping
  a.tok_sbefore = ro ? nullptr : b->d_sbefore; a.text_s_end = ro ? nullptr : b->d_ts_end;
  a.doc_ns = ro ? nullptr : b->d_doc_ns;
  a.tok_cap = b->tok_cap; a.sent_cap = b->sent_cap; a.text_cap = b->text_cap;
  // a document of many lanes is compacted by one wave per DTK_SEG_LANES lanes (a double-array document
  // with an EOT inside stays sequential: k_seg_scan decides)
  const bool seg = b->chunk != 0 && b->long_docs;
  a.seg_doc = seg ? b->d_seg_tab : nullptr;
  a.seg_lane0 = b->d_seg_tab + b->seg_cap; a.seg_nl = b->d_seg_tab + 2 * (size_t)b->seg_cap;
  a.n_segs = b->n_segs; a.chunk_off = b->d_chunk_off; a.lane_start = b->d_lane_start; a.lane_cnt = b->d_lane_cnt;
  a.seg_sum = b->d_seg_sum; a.seg_in = b->d_seg_in;
  a.doc_seq = b->d_seg_tab + 3 * (size_t)b->seg_cap + b->max_docs + 1;
  a.any_irregular = (uint32_t *)(b->d_totals + 7);
  a.any_eot = a.any_irregular + 1;
  b->last_args = a;
  if (seg && (which & 1) && dtk_launch_seg_prepare(&a, b->d_seg_tab + 3 * (size_t)b->seg_cap, b->stream))
    return hip_fail(hipGetLastError(), "segment carries");
  if (dtk_launch_compact(&a, b->small_max, b->d_big_docs, b->n_big, which, b->stream))
    return hip_fail(hipGetLastError(), "compact pass 2");
  if (which & 2) b->ran_full = true;
  return DTK_OK;
}

// One repair round on the batch's stream: spread + reset, clear, then the stages of the first pass restricted to
// what is repaired; the documents still broken afterwards are counted in *n_bad_out (zero before the round).
static int repair_round(dtk_batch *b, const dtk_model *m, const DtkWalkArgs *w, const DtkSpecArgs *sp, uint32_t *n_bad_out) {
  hipStream_t s = b->stream;
  if (dtk_launch_spec(&m->tab, w, sp, 5, cmp_mask_of(m), b->d_redo, n_bad_out, s) ||
      dtk_launch_redo_clear(w, sp, s))
    return hip_fail(hipGetLastError(), "speculative repair");
  for (int stage = 1; stage <= 4; stage++)
    if (dtk_launch_spec(&m->tab, w, sp, stage, cmp_mask_of(m), b->d_redo, n_bad_out, s))
      return hip_fail(hipGetLastError(), "speculative repair");
  return DTK_OK;
}

static int launch_to_host(dtk_batch *b);

extern "C" int dtk_batch_run(const dtk_model *m, dtk_batch *b, uint32_t flags) {
  if (!m || !b) return DTK_E_ARG;
  if (b->n_docs == 0 || !b->d_off) return DTK_E_STATE;
  if (m->device != b->device) return DTK_E_ARG;
  int prc = plan_lanes(b);
  if (prc != DTK_OK) return prc;
  if (b->dl_begun && !b->dl_waited) { HIP_TRY(hipEventSynchronize(b->ev_dl)); b->dl_waited = true; }  // (the last run's results on their way out)
  b->dl_begun = false;
  if (b->up_pending) {  // the kernels wait for the input's upload on the other stream
    HIP_TRY(hipStreamWaitEvent(b->stream, b->ev_up, 0));
    b->up_pending = false;
  }
  b->last_model = m;
  b->last_flags = flags;
  b->repair_rounds = 0;
  hipStream_t s = b->stream;
  const bool prof = b->profiling;
#define STAGE(i) do { if (prof) HIP_TRY(hipEventRecord(b->ev[i], s)); } while (0)
#ifdef DTK_EXPERIMENTS
  // knock-out timing (scripts/knockout.sh): from the second run on, skip the stages named by the
  // bit mask; with an unchanged input their outputs of the first run are still valid
  const int exp_skip = g_dbg.exp_skip;
  const int skip = b->exp_runs++ > 0 ? exp_skip : 0;
#else
  const int skip = 0;
#endif
  STAGE(0);
  bool fold_acc = false;
  size_t acc_bytes = 0;
  {
    // carve the accumulator block for this run's document count: totals, counts, status,
    // check words -- cleared together with the two event arrays by one launch
    const size_t nd = b->n_docs;
    uint8_t *q = b->d_acc;
    b->d_tok_off = b->d_csr; b->d_sent_off = b->d_csr + (nd + 1); b->d_text_off = b->d_csr + 2 * (nd + 1);
    b->d_totals = (uint64_t *)q; q += DTK_TOTALS_BYTES;  // (the striped lookup counters right behind the totals)
    b->d_tok_cnt = (uint64_t *)q; q += (nd + 1) * 8;
    b->d_sent_cnt = (uint64_t *)q; q += (nd + 1) * 8;
    b->d_text_cnt = (uint64_t *)q; q += (nd + 1) * 8;
    b->d_status = (uint32_t *)q; q += nd * 4;
    b->d_first_bad = (uint32_t *)q; q += nd * 4;
    b->d_fail_lane = (uint32_t *)q; q += nd * 4;
    b->d_doc_tail = (uint32_t *)q; q += nd * 4;
    const size_t acc_used = ((size_t)(q - b->d_acc) + 15) & ~(size_t)15;  // the block has 64 bytes of slack
    b->bit_words = (uint32_t)(((b->total + nd) / 32 + 8) & ~(uint64_t)3);  // 16-byte multiples per kind
    // (the event bitmaps are cleared by k_symbolize's blocks, unless it does not run)
    const bool fold = b->total > 0 && !(skip & 1);
    // ... and the accumulator block too, once it has been cleared whole (totals[6], which k_symbolize's own blocks
    // write, is left out there: it holds the number of the last run that saw invalid UTF-8)
    fold_acc = fold && b->acc_primed && acc_used / 16 < 0xFFFFFFFFull &&
               !g_dbg.clear_kernel;
    if (!fold_acc) {
      if (dtk_launch_clear2(b->d_acc, acc_used, b->d_bits, (fold || (skip & 4)) ? 0 : (size_t)EVB_KINDS * b->bit_words * 4, s))
        return hip_fail(hipGetLastError(), "clear");
      b->acc_primed = true;
    }
    acc_bytes = acc_used;
  }
  STAGE(1);
  if (!(skip & 1) && dtk_launch_symbolize(b->d_text, b->d_off, b->n_docs, b->total, &m->sig, b->d_sym,
                           b->d_text == b->d_text_own, b->d_blk_doc, (unsigned long long *)(b->d_totals + 6), b->d_rsbits,
                           b->d_bits, b->bit_words, fold_acc ? b->d_acc : nullptr, acc_bytes, ++b->epoch, s))
    return hip_fail(hipGetLastError(), "symbolize");
  STAGE(2);
  DtkWalkArgs w = walk_args(b);
  bool fix_in_scan = false;
  DtkSpecArgs sp_first{};
  if (b->chunk == 0) {
    STAGE(3); STAGE(4);
    if (dtk_launch_walk(&m->tab, &w, s)) return hip_fail(hipGetLastError(), "walk");
    STAGE(5); STAGE(6); STAGE(7);
  } else {
    DtkSpecArgs sp = spec_args(b, false);
    sp_first = sp;
    uint32_t *nb = (uint32_t *)(b->d_totals + 8);  // nb[0]: broken documents after the first pass, nb[r + 1]: after round r
    // DATOK_SPLIT_START=1: start records and chunk walk as two launches (the repair rounds' kernels)
    const bool split_env = g_dbg.split_start != 0;
    const bool split = split_env || sp.lds_words == 0;  // (k_spec_both reports through the wave's LDS bitmaps)
    // Device-side repair: if the batch's last run had to repair (text with tags, say), two repair rounds are
    // enqueued right behind the first pass; their kernels return at once when the verification before them found
    // nothing broken, and the scan / compaction behind them only run once nothing is.  A miss then costs no host
    // round trip.  (Not done blindly: ten empty launches cost a clean corpus some 15 us per batch.)
    b->dev_rounds = g_dbg.dev_rounds >= 0 ? (uint32_t)g_dbg.dev_rounds : (b->expect_repairs ? 2u : 0u);
    if (b->dev_rounds > 3u) b->dev_rounds = 3u;
    fix_in_scan = b->n_docs <= 8192u && b->dev_rounds == 0;  // k_spec_fix's step rides in the one-block scan kernel
    for (int stage = 0; stage < 5; stage++) {
      if (((skip & 2) && stage <= 1) || ((skip & 4) && stage == 2) || (stage == 4 && fix_in_scan)) { STAGE(3 + stage); continue; }
      // one launch for start records + walk (timed as "walk"), one for link + verify
      const int what = split ? stage : (stage <= 1 ? -1 : stage == 2 ? 6 : stage == 3 ? 7 : stage);
      if (what >= 0 && dtk_launch_spec(&m->tab, &w, &sp, what, cmp_mask_of(m), b->d_redo, nb, s))
        return hip_fail(hipGetLastError(), "speculative walk");
      STAGE(3 + stage);  // ends: start records, link, chunk walk, verify, fix (split) / -, -, start + walk, link + verify, fix
    }
    for (uint32_t r = 0; r < b->dev_rounds; r++) {
      DtkSpecArgs rp = spec_args(b, true);
      rp.go = nb + r;
      int rc = repair_round(b, m, &w, &rp, nb + r + 1);
      if (rc != DTK_OK) return rc;
    }
  }
  DtkCompactArgs c{};
  c.text = b->d_text; c.rs_bits = b->d_rsbits; c.doc_off = b->d_off; c.n_docs = b->n_docs;
  c.bits = b->d_bits; c.bit_words = b->bit_words; c.doc_tail = b->d_doc_tail; c.status = b->d_status;
  c.flags = flags & DTK_NEWLINE_AFTER_EOT; c.kind = m->kind;
  c.tok_off = b->d_tok_off; c.sent_off = b->d_sent_off; c.text_off = b->d_text_off;
  c.totals = b->d_totals;
  // rows are sized by the walk's own counts (no counting pass)
  if (dtk_launch_scan3(b->d_tok_cnt, b->d_sent_cnt, b->d_text_cnt, b->d_tok_off, b->d_sent_off, b->d_text_off,
                       b->n_docs, b->d_totals, b->d_status, b->d_scan_ws, fix_in_scan ? &sp_first : nullptr, b->d_redo,
                       (uint32_t *)(b->d_totals + 8), b->chunk ? (const uint32_t *)(b->d_totals + 8) + b->dev_rounds : nullptr, s))
    return hip_fail(hipGetLastError(), "scan");
  STAGE(8);
  c.skip_if = b->chunk ? (const uint32_t *)(b->d_totals + 8) + b->dev_rounds : nullptr;
  b->last_args = c;
  // (the kernel for documents with EOT calls only if this batch object's last run had such documents; finish()
  //  launches it when the other kernel reports one after all)
  b->ran_full = false;
  const bool eager_full = g_dbg.compact_full != 0;  // (tests: both kernels with every run)
  int rc = (skip & 8) ? DTK_OK : launch_compact2(b, (b->expect_eot || eager_full) ? 3 : 1);
  if (rc != DTK_OK) return rc;
  STAGE(9);
#undef STAGE
  b->eager_fields = 0;
  b->results_changed = false;
  if ((b->fields & DTK_R_EAGER) && !(skip & 8)) {
    rc = launch_to_host(b);
    if (rc != DTK_OK) return rc;
  }
  if (!(skip & 16))  // (knock-out: what the totals copy costs; the first run's values stand in)
    HIP_TRY(hipMemcpyAsync(b->h_totals, b->d_totals, DTK_TOTALS_BYTES, hipMemcpyDeviceToHost, s));
  if (!b->ev_ran) HIP_TRY(hipEventCreateWithFlags(&b->ev_ran, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(b->ev_ran, s));
  b->ev_ran_valid = true;
  b->ran = true;
  b->totals_valid = false;
  b->render_flags = 0xFFFFFFFFu;
  return DTK_OK;
}

// 1: everything dtk_batch_run enqueued has finished (dtk_batch_totals & co. will not wait for the GPU unless the
// run needs a repair); 0: not yet; never blocks.  (An event query: hipStreamQuery on a busy stream was seen to
// return only when the stream had drained.)
extern "C" int dtk_batch_done(dtk_batch *b) {
  if (!b || !b->ran || !b->ev_ran) return 0;
  const hipError_t e = hipEventQuery(b->ev_ran);
  if (e != hipSuccess) (void)hipGetLastError();
  return e == hipSuccess ? 1 : 0;
}

extern "C" int dtk_batch_set_profiling(dtk_batch *b, int enable) {
  if (!b) return DTK_E_ARG;
  if (enable && !b->ev[0])
    for (auto &e : b->ev) HIP_TRY(hipEventCreate(&e));
  b->profiling = enable != 0;
  return DTK_OK;
}

extern "C" int dtk_batch_stage_ms(dtk_batch *b, float ms[DTK_N_STAGES]) {
  if (!b || !ms) return DTK_E_ARG;
  if (!b->ran || !b->profiling) return DTK_E_STATE;
  HIP_TRY(hipStreamSynchronize(b->stream));
  for (int i = 0; i < DTK_N_STAGES; i++) HIP_TRY(hipEventElapsedTime(&ms[i], b->ev[i], b->ev[i + 1]));
  return DTK_OK;
}

extern "C" int dtk_batch_sync(dtk_batch *b) {
  if (!b) return DTK_E_ARG;
  return wait_own(b);
}

// The exact pass (k_exact_doc): documents flagged ST_IRREGULAR by the walk are walked again by one lane each,
// which writes their rows of the result arrays in the reference's call order and lists their calls.  Rare
// by construction (no shipped model produces one on any test corpus), hence the host round trips.
static int run_exact(dtk_batch *b) {
  const dtk_model *m = b->last_model;
  hipStream_t s = b->stream;
  const uint32_t nd = b->n_docs;
  std::vector<uint32_t> st(nd);
  HIP_TRY(hipMemcpy(st.data(), b->d_status, (size_t)nd * 4, hipMemcpyDeviceToHost));
  for (uint32_t d = 0; d < nd; d++)
    if (st[d] & ST_IRREGULAR) b->h_exact_ids.push_back(d);
  const uint32_t n = (uint32_t)b->h_exact_ids.size();
  if (n == 0) return DTK_OK;
  if (n > b->exact_cap) {
    void *old[] = {b->d_exact_ids, b->d_exact_cnt, b->d_exact_off};
    for (void *p : old)
      if (p) HIP_TRY(hipFree(p));
    b->d_exact_ids = b->d_exact_cnt = nullptr; b->d_exact_off = nullptr; b->exact_cap = 0;
    const uint64_t cap = (uint64_t)n + n / 4 + 16;
    HIP_TRY(hipMalloc((void **)&b->d_exact_ids, cap * 4));
    HIP_TRY(hipMalloc((void **)&b->d_exact_cnt, cap * 4));
    HIP_TRY(hipMalloc((void **)&b->d_exact_off, (cap + 1) * 8));
    b->exact_cap = (uint32_t)cap;
  }
  HIP_TRY(hipMemcpy(b->d_exact_ids, b->h_exact_ids.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  DtkExactArgs X{};
  X.sym = sym_of(b); X.text = b->d_text; X.doc_off = b->d_off; X.n = n; X.docs = b->d_exact_ids;
  X.n_calls = b->d_exact_cnt; X.call_off = b->d_exact_off; X.status = b->d_status;
  X.flags = b->last_flags & DTK_NEWLINE_AFTER_EOT; X.step_factor = walk_args(b).step_factor;
  X.tok_off = b->d_tok_off; X.sent_off = b->d_sent_off; X.text_off = b->d_text_off;
  const bool no_rune = (b->last_flags & DTK_NO_RUNE_OFFSETS) != 0, no_byte = (b->last_flags & DTK_NO_BYTE_OFFSETS) != 0;
  X.tok_rstart = no_rune ? nullptr : b->d_rstart; X.tok_rend = no_rune ? nullptr : b->d_rend;
  X.tok_bstart = no_byte ? nullptr : b->d_bstart; X.tok_bend = no_byte ? nullptr : b->d_bend;
  X.sent = b->d_sent; X.text_tok_end = b->d_ttok; X.text_sent_end = b->d_tsent;
  const bool ro = (b->last_flags & (DTK_OFFSETS_ONLY | DTK_NO_RUNE_OFFSETS | DTK_NO_BYTE_OFFSETS)) != 0;
  X.tok_sbefore = ro ? nullptr : b->d_sbefore; X.text_s_end = ro ? nullptr : b->d_ts_end;
  X.doc_ns = ro ? nullptr : b->d_doc_ns;
  X.pass = 0;
  if (dtk_launch_exact(&m->tab, &X, s)) return hip_fail(hipGetLastError(), "exact pass (count)");
  std::vector<uint32_t> cnt(n);
  HIP_TRY(hipMemcpyAsync(cnt.data(), b->d_exact_cnt, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  b->h_exact_off.assign((size_t)n + 1, 0);
  for (uint32_t i = 0; i < n; i++) b->h_exact_off[i + 1] = b->h_exact_off[i] + cnt[i];
  const uint64_t total = b->h_exact_off[n];
  if (total > b->calls_cap) {
    if (b->d_calls) HIP_TRY(hipFree(b->d_calls));
    b->d_calls = nullptr; b->calls_cap = 0;
    HIP_TRY(hipMalloc((void **)&b->d_calls, (total + total / 4 + 16) * sizeof(DtkCall)));
    b->calls_cap = total + total / 4 + 16;
  }
  HIP_TRY(hipMemcpy(b->d_exact_off, b->h_exact_off.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice));
  X.calls = b->d_calls;
  X.pass = 1;
  if (dtk_launch_exact(&m->tab, &X, s)) return hip_fail(hipGetLastError(), "exact pass");
  b->h_calls.resize((size_t)total);
  if (total) HIP_TRY(hipMemcpyAsync(b->h_calls.data(), b->d_calls, (size_t)total * sizeof(DtkCall), hipMemcpyDeviceToHost, s));
  // the documents' status words were rewritten: count the flagged ones again
  HIP_TRY(hipMemcpyAsync(st.data(), b->d_status, (size_t)nd * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  uint64_t flagged = 0;
  for (uint32_t d = 0; d < nd; d++) flagged += st[d] != 0;
  b->h_totals[3] = flagged;
  return DTK_OK;
}

// Reads the totals; if pass 2 found its arrays too small, grows them and runs
// pass 2 again (the only allocation that can follow a run).
static int finish(dtk_batch *b) {
  if (!b->ran) return DTK_E_STATE;
  if (b->totals_valid) return DTK_OK;
  { const int rc_ = wait_own(b); if (rc_ != DTK_OK) return rc_; }  // (on a lent stream: this batch's run, not its neighbours')
  // Speculation check failed somewhere: those documents are repaired from the last owning lane before their first
  // bad lane on (fix records, clear, re-link, re-walk, re-verify) until every lane chains.  Rounds enqueued ahead of
  // time (dtk_batch_run) have run on the device already; what is still broken behind them is repaired from here,
  // and the scan / compaction -- which did nothing in that case -- run afterwards.
  if (b->chunk != 0) {
    const uint32_t *hnb = (const uint32_t *)(b->h_totals + 8);
    for (uint32_t r = 0; r < b->dev_rounds; r++)
      if (hnb[r] != 0) b->repair_rounds++;
    uint32_t left = hnb[b->dev_rounds];
    if (left != 0) {
      b->results_changed = true;
      const dtk_model *m = b->last_model;
      hipStream_t s = b->stream;
      DtkWalkArgs w = walk_args(b);
      uint32_t *n_bad = (uint32_t *)(b->d_totals + 8) + b->dev_rounds;  // (the counter the scan / compaction looked at)
      // Every round verifies at least one more lane of every broken document (the first bad lane started from a true
      // state), so the longest document's lane count bounds the rounds.  Should they run out all the same, the batch
      // is walked again with one lane per document -- no speculation, nothing to repair.
      const uint32_t max_rounds = b->max_doc_lanes + 16u;
      while (left != 0) {
        b->repair_rounds++;
        DtkSpecArgs sp = spec_args(b, true);
        HIP_TRY(hipMemsetAsync(n_bad, 0, 4, s));
        int rc = repair_round(b, m, &w, &sp, n_bad);
        if (rc != DTK_OK) return rc;
        HIP_TRY(hipMemcpyAsync(&left, n_bad, 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (left != 0 && g_dbg.debug_repair && b->repair_rounds < 12) {
          std::vector<uint32_t> redo(b->n_docs), chunk_off(b->n_docs + 1);
          HIP_TRY(hipMemcpy(redo.data(), b->d_redo, (size_t)b->n_docs * 4, hipMemcpyDeviceToHost));
          HIP_TRY(hipMemcpy(chunk_off.data(), b->d_chunk_off, ((size_t)b->n_docs + 1) * 4, hipMemcpyDeviceToHost));
          for (uint32_t d = 0, shown = 0; d < b->n_docs && shown < 3; d++)
            if (redo[d] != 0xFFFFFFFFu) {
              const uint32_t L0 = chunk_off[d], L1 = chunk_off[d + 1], n = std::min<uint32_t>(L1 - L0, 24u);
              std::vector<DtkLaneState> st(n), en(n);
              HIP_TRY(hipMemcpy(st.data(), b->d_lane_start + L0, n * sizeof(DtkLaneState), hipMemcpyDeviceToHost));
              HIP_TRY(hipMemcpy(en.data(), b->d_lane_end + L0, n * sizeof(DtkLaneState), hipMemcpyDeviceToHost));
              fprintf(stderr, "round %u: %u broken; doc %u lanes %u redo from lane %u:", b->repair_rounds, left, d, L1 - L0, redo[d] - L0);
              for (uint32_t k = 0; k < n; k++)
                fprintf(stderr, " [%u: s %d/%u/%x e %d/%u/%x]", k, (int)st[k].p, st[k].t, st[k].flags, (int)en[k].p, en[k].t, en[k].flags);
              fprintf(stderr, "\n");
              shown++;
            }
        }
        if (left != 0 && (b->repair_rounds > max_rounds || b->repair_rounds >= b->round_limit)) {
          const uint32_t rounds = b->repair_rounds, keep = b->cfg_chunk;
          b->cfg_chunk = 0; b->plan_valid = false;
          rc = dtk_batch_run(m, b, b->last_flags);
          if (rc == DTK_OK) rc = finish(b);
          b->cfg_chunk = keep; b->plan_valid = false;
          b->repair_rounds = rounds; b->totals.repair_rounds = rounds;
          b->expect_repairs = true;
          return rc;
        }
      }
      if (dtk_launch_scan3(b->d_tok_cnt, b->d_sent_cnt, b->d_text_cnt, b->d_tok_off, b->d_sent_off, b->d_text_off,
                         b->n_docs, b->d_totals, b->d_status, b->d_scan_ws, nullptr, nullptr, nullptr, nullptr, s))
        return hip_fail(hipGetLastError(), "scan");
      b->last_args.skip_if = nullptr;
      int rc = launch_compact2(b, 3);
      if (rc != DTK_OK) return rc;
      HIP_TRY(hipMemcpyAsync(b->h_totals, b->d_totals, DTK_TOTALS_BYTES, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
    }
    b->expect_repairs = b->repair_rounds != 0;
  }
  const uint64_t nt = b->h_totals[0], ns = b->h_totals[1], nx = b->h_totals[2];
  b->n_invalid = b->h_totals[6] == b->epoch ? 1u : 0u;  // (the number of the last run that saw one: k_symbolize)
  if (nt > b->tok_cap || ns > b->sent_cap || nx > b->text_cap) {
    b->results_changed = true;
    int rc = alloc_outputs(b, nt + nt / 8 + 16, ns + ns / 8 + 16, nx + nx / 8 + 16);
    if (rc != DTK_OK) return rc;
    rc = launch_compact2(b, 3);
    if (rc != DTK_OK) return rc;
    HIP_TRY(hipMemcpyAsync(b->h_totals + 7, b->d_totals + 7, 8, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
  }
  // documents with EOT calls that the first compaction kernel left alone
  {
    const bool eot = (b->h_totals[7] >> 32) != 0;
    if (eot && !b->ran_full) {
      b->results_changed = true;
      int rc = launch_compact2(b, 2);
      if (rc != DTK_OK) return rc;
      HIP_TRY(hipMemcpyAsync(b->h_totals + 7, b->d_totals + 7, 8, hipMemcpyDeviceToHost, b->stream));
      HIP_TRY(hipStreamSynchronize(b->stream));
    }
    b->expect_eot = eot;
  }
  // documents whose calls are not in position order (ST_IRREGULAR): their rows come from the exact pass
  b->h_exact_ids.clear(); b->h_exact_off.assign(1, 0); b->h_calls.clear();
  if ((uint32_t)b->h_totals[7] != 0) {
    b->results_changed = true;
    int rc = run_exact(b);
    if (rc != DTK_OK) return rc;
  }
  // the arrays k_to_host brought over inside the run count if the kernel made its copy and nothing was touched since
  if (b->eager_fields && !b->results_changed && b->h_totals[11] == b->epoch) {
    b->dl_begun = true;
    b->dl_waited = true;
    b->dl_fields = b->eager_fields;
  }
  b->totals.n_docs = b->n_docs;
  b->totals.n_bytes = b->total;
  b->totals.n_tokens = nt;
  b->totals.n_sent = ns;
  b->totals.n_texts = nx;
  b->totals.n_flagged = b->h_totals[3];
  b->totals.walk_steps = 0;
  for (uint32_t i = 0; i < DTK_STEP_STRIPES; i++) b->totals.walk_steps += b->h_totals[16 + 16 * i];
  b->totals.n_lanes = b->chunk ? b->n_lanes : b->n_docs;
  b->totals.chunk_bytes = b->chunk;
  b->totals.repair_rounds = b->repair_rounds;
  b->totals_valid = true;
  return DTK_OK;
}

extern "C" int dtk_batch_totals(dtk_batch *b, dtk_totals *out) {
  if (!b || !out) return DTK_E_ARG;
  int rc = finish(b);
  if (rc != DTK_OK) return rc;
  *out = b->totals;
  return DTK_OK;
}

extern "C" int dtk_batch_status_host(dtk_batch *b, uint32_t *status, uint32_t n) {
  if (!b || !status || n > b->n_docs) return DTK_E_ARG;
  int rc = finish(b);
  if (rc != DTK_OK) return rc;
  if (n) HIP_TRY(hipMemcpy(status, b->d_status, (size_t)n * 4, hipMemcpyDeviceToHost));
  return DTK_OK;
}

extern "C" int dtk_batch_result_device(dtk_batch *b, dtk_result_view *o) {
  if (!b || !o) return DTK_E_ARG;
  int rc = finish(b);
  if (rc != DTK_OK) return rc;
  o->tok_r16 = nullptr;  // (host results only)
  o->tok_off = b->d_tok_off; o->sent_off = b->d_sent_off; o->text_off = b->d_text_off;
  const bool no_rune = (b->last_flags & DTK_NO_RUNE_OFFSETS) != 0, no_byte = (b->last_flags & DTK_NO_BYTE_OFFSETS) != 0;
  o->tok_rstart = no_rune ? nullptr : b->d_rstart; o->tok_rend = no_rune ? nullptr : b->d_rend;
  o->tok_bstart = no_byte ? nullptr : b->d_bstart; o->tok_bend = no_byte ? nullptr : b->d_bend;
  o->sent = b->d_sent; o->text_tok_end = b->d_ttok; o->text_sent_end = b->d_tsent;
  o->status = b->d_status; o->ev_bits = b->d_bits; o->ev_words = b->bit_words; o->doc_tail = b->d_doc_tail;
  o->n_exact = (uint32_t)b->h_exact_ids.size();
  o->exact_doc = b->d_exact_ids; o->exact_off = b->d_exact_off; o->calls = (const dtk_call *)b->d_calls;
  return DTK_OK;
}

extern "C" int dtk_batch_set_result_fields(dtk_batch *b, uint32_t fields) {
  if (!b || (fields & ~(uint32_t)(DTK_R_ALL | DTK_R_TOK_RUNE16 | DTK_R_EAGER))) return DTK_E_ARG;
  b->fields = fields;
  return DTK_OK;
}

// DTK_R_EAGER: the selected arrays leave for the host inside the run, by a kernel behind the compaction that reads
// the sizes where they are -- on the device (k_to_host).  Page-locked buffers sized like the device arrays.
static int launch_to_host(dtk_batch *b) {
  uint32_t f = b->fields;
  if (b->last_flags & DTK_NO_RUNE_OFFSETS) f &= ~(uint32_t)DTK_R_TOK_RUNE;
  if (b->last_flags & DTK_NO_BYTE_OFFSETS) f &= ~(uint32_t)DTK_R_TOK_BYTE;
  DtkToHostArgs a{};
  const uint64_t nd = b->n_docs;
  auto add = [&](int which, const void *src, uint64_t bytes_or_elem, int count_from, uint64_t cap) -> int {
    const uint64_t need = count_from >= 0 ? cap * bytes_or_elem : bytes_or_elem;
    int rc = pin_fit(b->pin[which], (size_t)need);
    if (rc != DTK_OK) return rc;
    void *dp = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&dp, b->pin[which].p, 0));
    a.src[a.n] = src; a.dst[a.n] = dp; a.bytes[a.n] = bytes_or_elem; a.count_from[a.n] = count_from;
    a.cap[a.n] = count_from >= 0 ? b->pin[which].cap / bytes_or_elem : 0;
    a.n++;
    return DTK_OK;
  };
  int rc;
  if (f & DTK_R_TOK_RUNE) {
    if ((rc = add(dtk_batch::PB_RSTART, b->d_rstart, 4, 0, b->tok_cap))) return rc;
    if ((rc = add(dtk_batch::PB_REND, b->d_rend, 4, 0, b->tok_cap))) return rc;
  }
  if (f & DTK_R_TOK_BYTE) {
    if ((rc = add(dtk_batch::PB_BSTART, b->d_bstart, 4, 0, b->tok_cap))) return rc;
    if ((rc = add(dtk_batch::PB_BEND, b->d_bend, 4, 0, b->tok_cap))) return rc;
  }
  if (f & DTK_R_EVENTS) {
    if ((rc = add(dtk_batch::PB_BITS, b->d_bits, (uint64_t)EVB_KINDS * b->bit_words * 4, -1, 0))) return rc;
    if ((rc = add(dtk_batch::PB_TAIL, b->d_doc_tail, nd * 4, -1, 0))) return rc;
  }
  if (f & DTK_R_SENT)
    if ((rc = add(dtk_batch::PB_SENT, b->d_sent, 4, 1, b->sent_cap))) return rc;
  if (f & DTK_R_TEXTS) {
    if ((rc = add(dtk_batch::PB_TTOK, b->d_ttok, 4, 2, b->text_cap))) return rc;
    if ((rc = add(dtk_batch::PB_TSENT, b->d_tsent, 4, 2, b->text_cap))) return rc;
  }
  if (f & DTK_R_CSR)
    if ((rc = add(dtk_batch::PB_TOK_OFF, b->d_csr, 3 * (nd + 1) * 8, -1, 0))) return rc;
  if (f & DTK_R_STATUS)
    if ((rc = add(dtk_batch::PB_STATUS, b->d_status, nd * 4, -1, 0))) return rc;
  if (a.n == 0) return DTK_OK;
  a.totals = b->d_totals;
  a.skip_if = b->last_args.skip_if;
  a.done = b->d_totals + 11;
  a.epoch = b->epoch;
  if (dtk_launch_to_host(&a, b->stream)) return hip_fail(hipGetLastError(), "results to the host");
  b->eager_fields = f & DTK_R_ALL;
  return DTK_OK;
}

// Completes the run (as dtk_batch_totals: speculation check, repairs, capacity check -- the batch's stream is idle
// afterwards) and enqueues the copies of the selected result arrays on the download stream.  Returns at once.
extern "C" int dtk_batch_set_download_stream(dtk_batch *b, void *stream) {
  if (!b) return DTK_E_ARG;
  if (b->dl_begun && !b->dl_waited) { HIP_TRY(hipEventSynchronize(b->ev_dl)); b->dl_waited = true; }
  if (b->dl_stream && b->dl_own) HIP_TRY(hipStreamDestroy(b->dl_stream));
  b->dl_stream = (hipStream_t)stream;
  b->dl_own = false;
  return DTK_OK;
}

extern "C" void *dtk_batch_download_stream(dtk_batch *b) {
  if (!b) return nullptr;
  if (!b->dl_stream) {
    if (hipStreamCreateWithFlags(&b->dl_stream, hipStreamNonBlocking) != hipSuccess) { b->dl_stream = nullptr; return nullptr; }
    b->dl_own = true;
  }
  return (void *)b->dl_stream;
}

extern "C" int dtk_batch_download_begin(dtk_batch *b) {
  if (!b) return DTK_E_ARG;
  int rc = finish(b);
  if (rc != DTK_OK) return rc;

  uint32_t sel = b->fields & (DTK_R_ALL | DTK_R_TOK_RUNE16);
  if (sel & DTK_R_TOK_RUNE16) {
    // the narrow form holds every offset of a document of at most 32 767 bytes; a batch with a longer one gets the
    // 32-bit arrays in its place
    if (!b->max_doc_valid) {
      uint64_t m = 0;
      for (uint32_t d = 0; d < b->n_docs; d++) m = std::max(m, b->h_doc_off[d + 1] - b->h_doc_off[d]);
      b->max_doc_bytes = m;
      b->max_doc_valid = true;
    }
    if (b->max_doc_bytes > 32767u) sel = (sel & ~(uint32_t)DTK_R_TOK_RUNE16) | DTK_R_TOK_RUNE;
  }
  if (b->last_flags & DTK_NO_RUNE_OFFSETS) sel &= ~(uint32_t)(DTK_R_TOK_RUNE | DTK_R_TOK_RUNE16);  // (not written by this run)
  if (b->last_flags & DTK_NO_BYTE_OFFSETS) sel &= ~(uint32_t)DTK_R_TOK_BYTE;
  if (b->dl_begun && (b->dl_fields & sel) == sel) return DTK_OK;
  const uint32_t want = sel & ~(b->dl_begun ? b->dl_fields : 0u);
  if (!dtk_batch_download_stream(b)) return hip_fail(hipGetLastError(), "download stream");
  const uint64_t nd = b->n_docs, nt = b->totals.n_tokens, ns = b->totals.n_sent, nx = b->totals.n_texts;
  auto get = [&](int which, const void *src, uint64_t bytes) -> int {
    int rc2 = pin_fit(b->pin[which], (size_t)bytes);
    if (rc2 != DTK_OK) return rc2;
    if (bytes) HIP_TRY(hipMemcpyAsync(b->pin[which].p, src, (size_t)bytes, hipMemcpyDeviceToHost, b->dl_stream));
    return DTK_OK;
  };
  // (the large arrays first: the small ones ride behind them)
  if ((want & DTK_R_TOK_RUNE16) && nt) {
    // packed on the download stream itself, in front of its copy: the batch's stream is idle (finish()) and stays free
    if (b->r16_cap < b->tok_cap) {
      if (b->d_r16) HIP_TRY(hipFree(b->d_r16));
      b->d_r16 = nullptr; b->r16_cap = 0;
      HIP_TRY(hipMalloc((void **)&b->d_r16, std::max<uint64_t>(b->tok_cap, 4) * 4));
      b->r16_cap = b->tok_cap;
    }
    if (dtk_launch_pack_r16(b->d_rstart, b->d_rend, b->d_r16, nt, b->dl_stream)) return hip_fail(hipGetLastError(), "pack r16");
    if ((rc = get(dtk_batch::PB_R16, b->d_r16, nt * 4))) return rc;
  }
  if (want & DTK_R_TOK_RUNE) {
    if ((rc = get(dtk_batch::PB_RSTART, b->d_rstart, nt * 4))) return rc;
    if ((rc = get(dtk_batch::PB_REND, b->d_rend, nt * 4))) return rc;
  }
  if (want & DTK_R_TOK_BYTE) {
    if ((rc = get(dtk_batch::PB_BSTART, b->d_bstart, nt * 4))) return rc;
    if ((rc = get(dtk_batch::PB_BEND, b->d_bend, nt * 4))) return rc;
  }
  if (want & DTK_R_EVENTS) {
    if ((rc = get(dtk_batch::PB_BITS, b->d_bits, (uint64_t)EVB_KINDS * b->bit_words * 4))) return rc;
    if ((rc = get(dtk_batch::PB_TAIL, b->d_doc_tail, nd * 4))) return rc;
  }
  if (want & DTK_R_SENT)
    if ((rc = get(dtk_batch::PB_SENT, b->d_sent, ns * 4))) return rc;
  if (want & DTK_R_TEXTS) {
    if ((rc = get(dtk_batch::PB_TTOK, b->d_ttok, nx * 4))) return rc;
    if ((rc = get(dtk_batch::PB_TSENT, b->d_tsent, nx * 4))) return rc;
  }
  if (want & DTK_R_CSR)  // (tok_off | sent_off | text_off lie back to back: dtk_batch_run)
    if ((rc = get(dtk_batch::PB_TOK_OFF, b->d_csr, 3 * (nd + 1) * 8))) return rc;
  if (want & DTK_R_STATUS)
    if ((rc = get(dtk_batch::PB_STATUS, b->d_status, nd * 4))) return rc;
  if (!b->ev_dl) HIP_TRY(hipEventCreateWithFlags(&b->ev_dl, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(b->ev_dl, b->dl_stream));
  b->dl_waited = false;
  b->dl_fields = (b->dl_begun ? b->dl_fields : 0u) | want;
  b->dl_begun = true;
  return DTK_OK;
}

extern "C" int dtk_batch_result_host(dtk_batch *b, dtk_result_view *o) {
  if (!b || !o) return DTK_E_ARG;
  int rc = dtk_batch_download_begin(b);
  if (rc != DTK_OK) return rc;
  if (!b->dl_waited) { HIP_TRY(hipEventSynchronize(b->ev_dl)); b->dl_waited = true; }
  const uint32_t f = b->dl_fields;
  auto at = [&](int which, uint32_t field) -> const void * { return (f & field) ? b->pin[which].p : nullptr; };
  memset(o, 0, sizeof(*o));
  o->tok_off = (const uint64_t *)at(dtk_batch::PB_TOK_OFF, DTK_R_CSR);
  o->sent_off = o->tok_off ? o->tok_off + (b->n_docs + 1) : nullptr;
  o->text_off = o->tok_off ? o->tok_off + 2 * ((uint64_t)b->n_docs + 1) : nullptr;
  o->tok_rstart = (const int32_t *)at(dtk_batch::PB_RSTART, DTK_R_TOK_RUNE);
  o->tok_rend = (const int32_t *)at(dtk_batch::PB_REND, DTK_R_TOK_RUNE);
  o->tok_bstart = (const uint32_t *)at(dtk_batch::PB_BSTART, DTK_R_TOK_BYTE);
  o->tok_bend = (const uint32_t *)at(dtk_batch::PB_BEND, DTK_R_TOK_BYTE);
  o->sent = (const int32_t *)at(dtk_batch::PB_SENT, DTK_R_SENT);
  o->text_tok_end = (const uint32_t *)at(dtk_batch::PB_TTOK, DTK_R_TEXTS);
  o->text_sent_end = (const uint32_t *)at(dtk_batch::PB_TSENT, DTK_R_TEXTS);
  o->status = (const uint32_t *)at(dtk_batch::PB_STATUS, DTK_R_STATUS);
  o->ev_bits = (const uint32_t *)at(dtk_batch::PB_BITS, DTK_R_EVENTS);
  o->ev_words = b->bit_words;
  o->doc_tail = (const uint32_t *)at(dtk_batch::PB_TAIL, DTK_R_EVENTS);
  o->tok_r16 = (const uint32_t *)at(dtk_batch::PB_R16, DTK_R_TOK_RUNE16);
  o->n_exact = (uint32_t)b->h_exact_ids.size();
  o->exact_doc = b->h_exact_ids.data(); o->exact_off = b->h_exact_off.data(); o->calls = (const dtk_call *)b->h_calls.data();
  return DTK_OK;
}

// ---------------------------------------------------------------- rendering
//
// NewTokenWriter(w, bits) for every document of the batch, on the device (dtk_render.hip).
static int render(dtk_batch *b, uint32_t bits) {
  int rc = finish(b);
  if (rc != DTK_OK) return rc;
  if (bits & ~31u) return DTK_E_ARG;
  if (b->last_flags & (DTK_OFFSETS_ONLY | DTK_NO_RUNE_OFFSETS | DTK_NO_BYTE_OFFSETS)) return DTK_E_STATE;  // the run skipped what the renderer reads
  // the positions were computed under the run's NEWLINE_AFTER_EOT rule (token_writer.go:66-68)
  if ((bits ^ b->last_flags) & DTK_NEWLINE_AFTER_EOT) return DTK_E_ARG;
  bits &= 15u;
  if (b->render_flags == bits) return DTK_OK;
  hipStream_t s = b->stream;
  const uint64_t nt = b->totals.n_tokens, ns = b->totals.n_sent, nx = b->totals.n_texts, nd = b->n_docs;
  const uint64_t tt = dtk_render_tiles(nt), st = dtk_render_tiles(ns);
  const uint64_t words = 2 * (nt + 1) + (ns + 1) + 2 * tt + st + (nd + 1) + 4 * (nx + 1) + 8;
  if (words > b->rws_cap) {
    if (b->d_rws) HIP_TRY(hipFree(b->d_rws));
    b->d_rws = nullptr; b->rws_cap = 0;
    HIP_TRY(hipMalloc((void **)&b->d_rws, (words + words / 8) * 8));
    b->rws_cap = words + words / 8;
  }
  DtkRenderArgs R{};
  R.text = b->d_text; R.doc_off = b->d_off; R.n_docs = b->n_docs; R.flags = bits;
  R.tok_off = b->d_tok_off; R.sent_off = b->d_sent_off; R.text_off = b->d_text_off;
  R.n_tok = nt; R.n_sent = ns; R.n_text = nx;
  R.rstart = b->d_rstart; R.rend = b->d_rend; R.sent = b->d_sent;
  R.bstart = b->d_bstart; R.bend = b->d_bend; R.sbefore = b->d_sbefore;
  R.ttok = b->d_ttok; R.tsent = b->d_tsent; R.ts_end = b->d_ts_end; R.doc_ns = b->d_doc_ns;
  R.sym = sym_of(b); if (!b->n_invalid) R.sym.base = nullptr;
  uint64_t *q = b->d_rws;
  R.A = q; q += nt + 1; R.P = q; q += nt + 1; R.Q = q; q += ns + 1;
  R.blkA = q; q += tt; R.blkP = q; q += tt; R.blkQ = q; q += st;
  R.ns_off = q; q += nd + 1;
  R.tx_base = q; q += nx + 1; R.tx_stream = q; q += nx + 1; R.tx_pos = q; q += nx + 1; R.tx_sent = q; q += nx + 1;
  R.out_off = b->d_out_off;
  if (dtk_launch_render(&R, 0, s)) return hip_fail(hipGetLastError(), "render sizes");
  HIP_TRY(hipMemcpyAsync(b->h_totals + 10, R.tx_base + nx, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  const uint64_t total = b->h_totals[10];
  if (total > b->out_cap) {
    if (b->d_out) HIP_TRY(hipFree(b->d_out));
    b->d_out = nullptr; b->out_cap = 0;
    HIP_TRY(hipMalloc((void **)&b->d_out, total + total / 8 + 256));
    b->out_cap = total + total / 8 + 256;
  }
  R.out = b->d_out; R.out_total = total;
  if (total) {
    HIP_TRY(hipMemsetAsync(b->d_out, '\n', total, s));  // every separator that is not a space
    if (dtk_launch_render(&R, 1, s)) return hip_fail(hipGetLastError(), "render bytes");
  }
  b->out_total = total;
  b->render_flags = bits;
  return DTK_OK;
}

extern "C" int dtk_batch_render_device(dtk_batch *b, uint32_t bits, dtk_render_view *o) {
  if (!b || !o) return DTK_E_ARG;
  int rc = render(b, bits);
  if (rc != DTK_OK) return rc;
  o->bytes = b->d_out; o->doc_off = b->d_out_off; o->total = b->out_total;
  return DTK_OK;
}

extern "C" int dtk_batch_render_host(dtk_batch *b, uint32_t bits, dtk_render_view *o) {
  if (!b || !o) return DTK_E_ARG;
  int rc = render(b, bits);
  if (rc != DTK_OK) return rc;
  b->h_out.resize(std::max<uint64_t>(b->out_total, 1));
  b->h_out_off.resize((size_t)b->n_docs + 1);
  if (b->out_total)
    HIP_TRY(hipMemcpyAsync(b->h_out.data(), b->d_out, b->out_total, hipMemcpyDeviceToHost, b->stream));
  HIP_TRY(hipMemcpyAsync(b->h_out_off.data(), b->d_out_off, ((size_t)b->n_docs + 1) * 8, hipMemcpyDeviceToHost,
                         b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  o->bytes = b->h_out.data(); o->doc_off = b->h_out_off.data(); o->total = b->out_total;
  return DTK_OK;
}

extern "C" void dtk_free(void *p) { free(p); }
