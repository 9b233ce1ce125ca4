#!/usr/bin/env python3
"""Transcribe the reference's test expectations into JSON golden vectors.

Reads the Go test files of the upstream KorAP/Datok tree (default
/root/reference) as TEXT, interprets the small statement vocabulary they use
(model loads, ttokenize / Transduce / TransduceTokenWriter calls, assert.Equal
on tokens / sentences / w.String()) and writes, per test file, a list of
cases:

    {"src": "matrix_test.go:256", "test": "TestMatrix...", "calls":
        [{"model": "tokenizer_de.matok", "input": "...", "flags": 3}],
     "checks": [{"line": 259, "kind": "raw_eq", "value": "..."}, ...]}

Only inputs and expected outputs are stored (data, not source).  The output
of this script is committed (tests/golden/*.json); the script is re-run only
when the reference snapshot changes.  Nothing under tests/ reads
/root/reference at test time.

check kinds (applied to the concatenated output of `calls`):
  raw_eq v            w.String() == v
  contains v          v in w.String()
  ttok_at i v         ttokenize view (split on \\n+, last dropped)[i] == v
  ttok_len n          len(ttokenize view) == n
  ttok_join v         "\\n".join(ttokenize view) == v          (ttokenizeStr)
  ttok_len_gt n       len(ttokenize view) > n
  nl_at i v / nl_len n        strings.Split(out, "\\n")
  nlnl_at i v / nlnl_len n    strings.Split(out, "\\n\\n")
"""
import json
import os
import re
import sys

BITS = {"TOKENS": 1, "SENTENCES": 2, "TOKEN_POS": 4, "SENTENCE_POS": 8,
        "NEWLINE_AFTER_EOT": 16, "SIMPLE": 3}


# ------------------------------------------------------------------ lexer
def lex(src):
    """Yields (kind, value, line). kinds: id, str, int, op, nl."""
    i, n, line = 0, len(src), 1
    out = []
    while i < n:
        c = src[i]
        if c == "\n":
            out.append(("nl", "\n", line))
            line += 1
            i += 1
        elif c in " \t\r":
            i += 1
        elif src.startswith("//", i):
            while i < n and src[i] != "\n":
                i += 1
        elif src.startswith("/*", i):
            j = src.index("*/", i + 2)
            line += src.count("\n", i, j)
            i = j + 2
        elif c == "`":
            j = src.index("`", i + 1)
            val = src[i + 1:j].replace("\r", "")
            out.append(("str", val, line))
            line += src.count("\n", i, j)
            i = j + 1
        elif c == '"':
            j = i + 1
            buf = bytearray()
            while src[j] != '"':
                if src[j] == "\\":
                    e = src[j + 1]
                    simple = {"n": 10, "t": 9, "r": 13, "\\": 92, '"': 34, "'": 39,
                              "a": 7, "b": 8, "f": 12, "v": 11}
                    if e in simple:
                        buf.append(simple[e]); j += 2
                    elif e == "x":
                        buf.append(int(src[j + 2:j + 4], 16)); j += 4
                    elif e == "u":
                        buf += chr(int(src[j + 2:j + 6], 16)).encode("utf-8"); j += 6
                    elif e == "U":
                        buf += chr(int(src[j + 2:j + 10], 16)).encode("utf-8"); j += 10
                    elif e in "01234567":
                        buf.append(int(src[j + 1:j + 4], 8)); j += 4
                    else:
                        raise ValueError("escape \\%s line %d" % (e, line))
                else:
                    buf += src[j].encode("utf-8"); j += 1
            # Go strings are bytes; keep them as latin-1-safe JSON via a list if
            # they are not valid UTF-8 (none are in this snapshot).
            out.append(("str", buf.decode("utf-8"), line))
            i = j + 1
        elif c == "'":
            j = src.index("'", i + 1)
            if src[i + 1] == "\\":
                j = src.index("'", i + 3)
            out.append(("op", src[i:j + 1], line))
            i = j + 1
        elif c.isalpha() or c == "_":
            j = i
            while j < n and (src[j].isalnum() or src[j] == "_"):
                j += 1
            out.append(("id", src[i:j], line))
            i = j
        elif c.isdigit():
            j = i
            while j < n and src[j].isdigit():
                j += 1
            out.append(("int", int(src[i:j]), line))
            i = j
        else:
            two = src[i:i + 2]
            if two in (":=", "==", "!=", "<=", ">=", "&&", "||", "++", "--", "+="):
                out.append(("op", two, line)); i += 2
            else:
                out.append(("op", c, line)); i += 1
    return out


def statements(toks):
    """Split into statements at newlines / braces / semicolons at paren depth 0."""
    cur, depth = [], 0
    for t in toks:
        k, v, _ = t
        if k == "op" and v in "([":
            depth += 1
        elif k == "op" and v in ")]":
            depth -= 1
        if depth == 0 and ((k == "nl") or (k == "op" and v in "{};")):
            if cur:
                yield cur
            cur = []
            continue
        if k != "nl":
            cur.append(t)
    if cur:
        yield cur


def sig(st):
    """Compact signature string: ids/ops literal, S for str, N for int."""
    parts = []
    for k, v, _ in st:
        parts.append("S" if k == "str" else "N" if k == "int" else str(v))
    return " ".join(parts)


# ------------------------------------------------------------- interpreter
class Interp:
    def __init__(self, fname):
        self.fname = fname
        self.cases = []
        self.globals_str = {}
        self.reset_func(None)

    def reset_func(self, name):
        self.test = name
        self.models = {}     # var -> model file (persists for package globals too)
        self.foma = {}
        self.strs = dict(self.globals_str)
        self.readers = {}    # var -> input string
        self.writers = {}    # var -> flags
        self.calls = []      # since last w.Reset()
        self.case = None
        self.rawvars = {}

    # package-level model variables (mat_de, mat_en, dat) keep their binding
    GLOBAL_MODELS = {}

    def model_of(self, var):
        return self.models.get(var) or Interp.GLOBAL_MODELS.get(var)

    def strval(self, tok):
        k, v, _ = tok
        if k == "str":
            return v
        if k == "id":
            return self.strs.get(v)
        return None

    def new_call(self, model, inp, flags, line, ttok=False):
        if model is None or inp is None:
            self.case = None
            return
        if ttok:
            self.calls = []
        self.calls.append({"model": model, "input": inp, "flags": flags})
        self.case = {"src": "%s:%d" % (self.fname, line), "test": self.test,
                     "calls": list(self.calls), "checks": []}
        self.cases.append(self.case)

    def check(self, line, kind, **kw):
        if self.case is None:
            return
        d = {"line": line, "kind": kind}
        d.update(kw)
        self.case["checks"].append(d)

    def run(self, src):
        for st in statements(lex(src)):
            self.stmt(st)
        # xTest.../XTest... functions are disabled upstream: not goldens
        return [c for c in self.cases
                if c["checks"] and c["test"] and c["test"].startswith("Test")]

    def stmt(self, st):
        s = sig(st)
        line = st[0][2]
        ids = [v for k, v, _ in st]

        if s.startswith("func "):
            self.reset_func(st[1][1])
            return
        # var s string = `...`
        if s.startswith("var ") and s.endswith("= S") and self.test is None:
            self.globals_str[st[1][1]] = st[-1][1]
            self.strs[st[1][1]] = st[-1][1]
            return
        # model loads
        m = re.match(r"^(\w+) (:=|=) (LoadMatrixFile|LoadDatokFile|LoadTokenizerFile) \( S \)$", s)
        if m:
            f = os.path.basename(st[4][1])
            self.models[m.group(1)] = f
            if m.group(2) == "=":
                Interp.GLOBAL_MODELS[m.group(1)] = f
            return
        m = re.match(r"^(\w+) (:=|=) LoadFomaFile \( S \)$", s)
        if m:
            self.foma[m.group(1)] = os.path.basename(st[4][1])
            return
        m = re.match(r"^(\w+) (:=|=) (\w+) \. (ToMatrix|ToDoubleArray) \( \)$", s)
        if m and m.group(3) in self.foma:
            self.models[m.group(1)] = "fst:%s:%s" % (
                self.foma[m.group(3)], "matrix" if m.group(4) == "ToMatrix" else "datok")
            return
        # derived models (WriteTo/Parse round trips) behave like their source
        m = re.match(r"^(\w+) (:=|=) (ParseMatrix|ParseDatok) \( \w+ \)$", s)
        if m:
            src_var = "mat" if m.group(3) == "ParseMatrix" else "dat"
            if self.model_of(src_var):
                self.models[m.group(1)] = self.model_of(src_var)
            return
        # string variables
        m = re.match(r"^(\w+) (:=|=) S$", s)
        if m:
            self.strs[m.group(1)] = st[2][1]
            return
        # readers
        m = re.match(r"^(\w+) (:=|=) strings \. NewReader \( (S|\w+) \)$", s)
        if m:
            self.readers[m.group(1)] = self.strval(st[6])
            return
        m = re.match(r"^(\w+) \. Reset \( (S|\w+) \)$", s)
        if m and m.group(1) in self.readers:
            self.readers[m.group(1)] = self.strval(st[4])
            return
        if re.match(r"^w \. Reset \( \)$", s):
            self.calls = []
            self.case = None
            return
        # token writers
        m = re.match(r"^(\w+) (:=|=) NewTokenWriter \( w , (.+) \)$", s)
        if m:
            fl = 0
            for name in m.group(3).split(" | "):
                fl |= BITS[name.strip()]
            self.writers[m.group(1)] = fl
            return
        # ttokenize
        m = re.match(r"^tokens (:=|=) ttokenize \( (\w+) , w , (S|\w+) \)$", s)
        if m:
            self.new_call(self.model_of(m.group(2)), self.strval(st[8]), 3, line, ttok=True)
            self.view = "ttok"
            return
        # Transduce / TransduceTokenWriter, optionally wrapped in assert.True(...)
        m = re.search(r"(\w+) \. (Transduce|TransduceTokenWriter) \( (strings \. NewReader \( (S|\w+) \)|\w+) , (\w+) \)", s)
        if m:
            # locate the input token
            inp = None
            for idx, (k, v, _) in enumerate(st):
                if k == "id" and v in ("Transduce", "TransduceTokenWriter"):
                    rest = st[idx + 2:]
                    if rest[0][1] == "strings":
                        inp = self.strval(rest[4])
                    else:
                        inp = self.readers.get(rest[0][1])
                    break
            flags = 3 if m.group(2) == "Transduce" else self.writers.get(m.group(5))
            if flags is None:
                self.case = None
                return
            self.new_call(self.model_of(m.group(1)), inp, flags, line)
            return
        # views
        m = re.match(r"^(tokens|sentences) (:=|=) strings \. Split \( w \. String \( \) , S \)$", s)
        if m:
            sep = st[-2][1]
            self.view = {"\n": "nl", "\n\n": "nlnl"}.get(sep)
            return
        m = re.match(r"^(\w+) (:=|=) w \. String \( \)$", s)
        if m:
            self.rawvars[m.group(1)] = True
            return
        # assertions
        if s.startswith("assert . True ( strings . Contains ( w . String ( ) , S )"):
            self.check(line, "contains", value=st[-3][1])
            return
        m = re.match(r"^assert \. Equal \( ttokenizeStr \( (\w+) , (S|\w+) \) , S \)$", s)
        if m:
            self.new_call(self.model_of(m.group(1)), self.strval(st[8]), 3, line, ttok=True)
            self.check(line, "ttok_join", value=st[-2][1])
            self.case = None
            return
        if s.startswith("assert . Equal ("):
            inner = st[4:-1]
            # split the two arguments at the top-level comma
            depth, cut = 0, None
            for idx, (k, v, _) in enumerate(inner):
                if k == "op" and v in "([":
                    depth += 1
                elif k == "op" and v in ")]":
                    depth -= 1
                elif k == "op" and v == "," and depth == 0:
                    cut = idx
                    break
            if cut is None:
                return
            a, b = inner[:cut], inner[cut + 1:]
            for lit, expr in ((a, b), (b, a)):
                if len(lit) != 1 or lit[0][0] not in ("str", "int"):
                    continue
                e = sig(expr)
                val = lit[0][1]
                view = getattr(self, "view", None)
                if lit[0][0] == "str":
                    if e == "w . String ( )" or (len(expr) == 1 and expr[0][1] in self.rawvars):
                        self.check(line, "raw_eq", value=val)
                    elif re.match(r"^(tokens|sentences) \[ N \]$", e) and view:
                        self.check(line, view + "_at", index=expr[2][1], value=val)
                else:
                    if re.match(r"^len \( (tokens|sentences) \)$", e) and view:
                        self.check(line, view + "_len", value=val)
                return
            return


def file_driven(ref):
    """datok_test.go:1201-1236: one ttokenize per line of split/dontsplit.txt."""
    cases = []
    for fname, kind, line in (("dontsplit.txt", "dont", 1212), ("split.txt", "split", 1232)):
        path = os.path.join(ref, "testdata", "de", fname)
        for raw in open(path, encoding="utf-8"):
            tok = raw.strip()
            if not tok or tok.startswith("#"):
                continue
            checks = ([{"line": line + 1, "kind": "ttok_len", "value": 1},
                       {"line": line + 3, "kind": "ttok_at", "index": 0, "value": tok}]
                      if kind == "dont" else
                      [{"line": line + 1, "kind": "ttok_len_gt", "value": 1}])
            cases.append({"src": "datok_test.go:%d" % line,
                          "test": "GenderDontSplitFromFile" if kind == "dont" else "GenderSplitFromFile",
                          "file": "de/" + fname,
                          "calls": [{"model": "tokenizer_de.datok", "input": tok, "flags": 3}],
                          "checks": checks})
    return cases


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    here = os.path.dirname(os.path.abspath(__file__))
    total = 0
    for fname in ("matrix_test.go", "datok_test.go", "token_writer_test.go"):
        src = open(os.path.join(ref, fname), encoding="utf-8").read()
        it = Interp(fname)
        cases = it.run(src)
        if fname == "datok_test.go":
            cases += file_driven(ref)
        nchecks = sum(len(c["checks"]) for c in cases)
        total += nchecks
        out = os.path.join(here, fname.replace("_test.go", "_goldens.json"))
        with open(out, "w", encoding="utf-8") as f:
            json.dump({"source": "KorAP/Datok " + fname, "license": "Apache-2.0 (see LICENSE.datok)",
                       "strings": it.globals_str, "cases": cases}, f, ensure_ascii=False, indent=1)
        print("%s: %d cases, %d checks -> %s" % (fname, len(cases), nchecks, os.path.basename(out)))
    print("total checks:", total)


if __name__ == "__main__":
    main()
