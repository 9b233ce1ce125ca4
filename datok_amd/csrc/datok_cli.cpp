// datok_cli.cpp -- the reference's command line (cmd/datok.go:18-134) over the C-ABI.
//
//   datok tokenize -t <tokenizer> [--[no-]tokens] [--[no-]sentences] [-p|--token-positions]
//                  [--sentence-positions] [--newline-after-eot] <file | ->
//   datok convert  -i <foma.fst> -o <tokenizer> [-d]
//
// `tokenize` walks and renders on the GPU (dtk_transduce); the input stream is one document,
// U+0004 ends a text inside it.  `convert` is host only; -d: the double array (ToDoubleArray, datok.go:82-238).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/datok_gpu.h"

static int usage(const char *msg) {
  if (msg) std::fprintf(stderr, "datok: error: %s\n", msg);
  std::fprintf(stderr,
               "Usage: datok <command>\n\nFSA based tokenizer\n\nCommands:\n"
               "  convert --foma=STRING --tokenizer=STRING\n"
               "    Convert a compiled foma FST file to a Matrix or Double Array tokenizer\n\n"
               "  tokenize --tokenizer=STRING <input>\n    Tokenize a text\n");
  return 1;
}

static bool read_all(FILE *f, std::vector<uint8_t> &out) {
  uint8_t buf[1 << 16];
  size_t k;
  while ((k = std::fread(buf, 1, sizeof buf, f)) > 0) out.insert(out.end(), buf, buf + k);
  return !std::ferror(f);
}

// --name=value / --name value / -n value
static bool opt_value(int argc, char **argv, int &i, const char *lng, char sht, std::string &out) {
  const std::string a = argv[i];
  const std::string l = std::string("--") + lng;
  if (a.rfind(l + "=", 0) == 0) { out = a.substr(l.size() + 1); return true; }
  if (a == l || (a.size() == 2 && a[0] == '-' && a[1] == sht)) {
    if (i + 1 >= argc) return false;
    out = argv[++i];
    return true;
  }
  if (a.size() > 2 && a[0] == '-' && a[1] == sht && a[1] != '-') { out = a.substr(a[2] == '=' ? 3 : 2); return true; }
  return false;
}

static int cmd_convert(int argc, char **argv) {
  std::string foma, tok;
  bool da = false;
  for (int i = 2; i < argc; i++) {
    if (opt_value(argc, argv, i, "foma", 'i', foma) || opt_value(argc, argv, i, "tokenizer", 'o', tok)) continue;
    if (!std::strcmp(argv[i], "-d") || !std::strcmp(argv[i], "--double-array")) { da = true; continue; }
    return usage((std::string("unknown flag ") + argv[i]).c_str());
  }
  if (foma.empty()) return usage("missing flags: --foma=STRING");
  if (tok.empty()) return usage("missing flags: --tokenizer=STRING");
  FILE *f = std::fopen(foma.c_str(), "rb");
  std::vector<uint8_t> gz;
  if (!f || !read_all(f, gz)) {
    std::fprintf(stderr, "Unable to load foma file\n");  // cmd/datok.go:53
    if (f) std::fclose(f);
    return 1;
  }
  std::fclose(f);
  void *img = nullptr;
  size_t n = 0;
  const int rc = da ? dtk_foma_to_datok(gz.data(), gz.size(), &img, &n)  // cmd/datok.go:58-66
                    : dtk_foma_to_matok(gz.data(), gz.size(), &img, &n);
  if (rc != DTK_OK) {
    std::fprintf(stderr, "Unable to load foma file: %s\n", dtk_strerror(rc));
    return 1;
  }
  FILE *o = std::fopen(tok.c_str(), "wb");
  const bool ok = o && std::fwrite(img, 1, n, o) == n && std::fclose(o) == 0;
  dtk_free(img);
  if (!ok) { std::perror(tok.c_str()); return 1; }
  std::printf("File successfully converted.\n");  // cmd/datok.go:69
  return 0;
}

static int cmd_tokenize(int argc, char **argv) {
  std::string tok, input;
  bool tokens = true, sentences = true, tpos = false, spos = false, nl = false, have_input = false;
  for (int i = 2; i < argc; i++) {
    const std::string a = argv[i];
    if (opt_value(argc, argv, i, "tokenizer", 't', tok)) continue;
    if (a == "--tokens") tokens = true;
    else if (a == "--no-tokens") tokens = false;
    else if (a == "--sentences") sentences = true;
    else if (a == "--no-sentences") sentences = false;
    else if (a == "-p" || a == "--token-positions") tpos = true;
    else if (a == "--sentence-positions") spos = true;
    else if (a == "--newline-after-eot") nl = true;
    else if (a == "-" || a[0] != '-') {
      if (have_input) return usage("unexpected argument");
      input = a; have_input = true;
    } else return usage((std::string("unknown flag ") + a).c_str());
  }
  if (tok.empty()) return usage("missing flags: --tokenizer=STRING");
  if (!have_input) return usage("expected \"<input>\"");
  dtk_model *m = nullptr;
  int rc = dtk_model_load(tok.c_str(), &m);
  if (rc != DTK_OK) {
    std::fprintf(stderr, "%s\nUnable to load file\n", dtk_strerror(rc));  // cmd/datok.go:77-80
    return 1;
  }
  // cmd/datok.go:83-102
  uint32_t flags = 0;
  if (tokens) flags |= DTK_TOKENS;
  if (tpos) flags |= DTK_TOKEN_POS;
  if (sentences) flags |= DTK_SENTENCES;
  if (spos) flags |= DTK_SENTENCE_POS;
  if (nl) flags |= DTK_NEWLINE_AFTER_EOT;
  std::vector<uint8_t> text;
  FILE *f = input == "-" ? stdin : std::fopen(input.c_str(), "rb");
  if (!f || !read_all(f, text)) {
    std::perror(input.c_str());
    dtk_model_free(m);
    return 1;
  }
  if (f != stdin) std::fclose(f);
  char *out = nullptr;
  size_t n = 0;
  uint32_t status = 0;
  rc = dtk_transduce(m, text.data(), text.size(), flags, &out, &n, &status);
  dtk_model_free(m);
  if (rc != DTK_OK) {
    std::fprintf(stderr, "datok: %s %s\n", dtk_strerror(rc), dtk_last_hip_error());
    return 1;
  }
  std::fwrite(out, 1, n, stdout);
  dtk_free(out);
  if (status) {  // the reference panics here (1024-rune window, positions of a text without tokens)
    std::fprintf(stderr, "datok: input outside the reference's contract (status %u)\n", status);
    return 2;
  }
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 2) return usage("expected one of \"convert\", \"tokenize\"");
  if (!std::strcmp(argv[1], "convert")) return cmd_convert(argc, argv);
  if (!std::strcmp(argv[1], "tokenize")) return cmd_tokenize(argc, argv);
  if (!std::strcmp(argv[1], "-h") || !std::strcmp(argv[1], "--help")) { usage(nullptr); return 0; }
  return usage("expected one of \"convert\", \"tokenize\"");
}
