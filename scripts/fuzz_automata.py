#!/usr/bin/env python3
"""Random small automata (tests/craft.py writes them in both file formats) x random documents x chunkings x flags:
every document -- offsets, status and the rendered writer output -- against the oracle.  The shipped models only
exercise what real tokenizers do; this looks for constructs nobody wrote a test for.
usage: fuzz_automata.py [automata] [first seed]      (on an MI355X)"""
import gzip
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import craft  # noqa: E402
import datok_amd  # noqa: E402
from datok_amd import corpus  # noqa: E402
from oracle import oracle as O  # noqa: E402
from parity import assert_batch_equals_oracle  # noqa: E402

n_auto = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
SYMS = [craft.A, craft.B, craft.E, craft.SP, craft.DOT, craft.NL]


def automaton(rng):
    n = int(rng.integers(3, 9))
    arcs = {}
    for t in range(1, n + 1):
        row = {}
        for a in SYMS:
            if rng.random() < 0.55:
                row[a] = (int(rng.integers(1, n + 1)), bool(rng.random() < (0.6 if a in (craft.SP, craft.NL, craft.E) else 0.1)))
        if t < n and rng.random() < 0.45:          # epsilon arcs only upwards: no cycles (the loaders reject those)
            row[craft.EPS] = (int(rng.integers(t + 1, n + 1)), False)
        if rng.random() < 0.08:
            row[craft.UNKNOWN] = (int(rng.integers(1, n + 1)), False)
        if rng.random() < 0.08:
            row[craft.IDENTITY] = (int(rng.integers(1, n + 1)), bool(rng.random() < 0.3))
        arcs[t] = row
    if not arcs[1]:
        arcs[1][craft.A] = (1, False)
    return arcs


def documents(rng, n=160):
    alpha = "aab b \x04.\n" + ("xä" if rng.random() < 0.5 else "")   # x, a-umlaut: not in the sigma (identity / unknown)
    docs = [b"", b"a", b"\x04", b" ", b"a\x04a", b"a. b.\x04\n\na"]
    for _ in range(n):
        k = int(rng.integers(0, 90))
        docs.append("".join(alpha[int(i)] for i in rng.integers(0, len(alpha), size=k)).encode())
    long_ = [b"".join(docs[int(i)] for i in rng.integers(0, len(docs), size=30)) for _ in range(6)]
    return docs + long_


runs = docs_checked = 0
for seed in range(first, first + n_auto):
    rng = np.random.default_rng(seed)
    arcs = automaton(rng)
    docs = documents(rng)
    text, off = corpus.concat_docs(docs)
    for kind in ("matok", "datok"):
        blob = getattr(craft, kind + "_from")(arcs)
        path = "/tmp/fuzz_%d.%s" % (os.getpid(), kind)
        open(path, "wb").write(blob)
        tok = datok_amd.load_tokenizer_file(path)
        if tok is None:
            print("seed %d %s: the loader rejects the model" % (seed, kind))
            continue
        om = O.Model(raw=gzip.decompress(blob))
        for chunk, warm in ((0, 0), (16, 0), (16, 8), (32, 4), (None, 16)):
            for flags in (0, 16):
                with datok_amd.Batch(max(len(text), 1), len(docs)) as b:
                    if chunk is not None:
                        b.set_chunking(chunk, warm, extend=0 if warm < 8 else 16)
                    b.set_input(text, off)
                    b.run(tok, flags)
                    res = b.result()
                    try:
                        docs_checked += assert_batch_equals_oracle(om, res, text, off, flags)
                        if chunk in (0, 16) and warm == 0:
                            for bits in (3, 15):
                                data, o = b.render(bits | flags)
                                for d, doc in enumerate(docs):
                                    exp, est = om.transduce(doc, bits | flags)
                                    if est == 0 and not (int(res.status[d]) & ~datok_amd.ST_EMPTY_TEXT):
                                        assert data[int(o[d]):int(o[d + 1])] == exp, ("render", bits, d, doc)
                    except AssertionError as e:
                        print("FAIL seed %d %s chunk %s warm %d flags %d: %s" % (seed, kind, chunk, warm, flags, str(e)[:400]))
                        print("arcs =", arcs)
                        sys.exit(1)
                    runs += 1
    print("seed %d ok (%d states)" % (seed, max(arcs)), flush=True)
print("FUZZ OK: %d automata, %d runs, %d documents compared" % (n_auto, runs, docs_checked))
