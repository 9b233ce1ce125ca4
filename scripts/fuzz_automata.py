#!/usr/bin/env python3
"""Random small automata (tests/craft.py writes them in both file formats) x random documents x chunkings x flags:
every document -- offsets, status and the rendered writer output -- against the oracle.  The shipped models only
exercise what real tokenizers do; this looks for constructs nobody wrote a test for.
usage: fuzz_automata.py [automata] [first seed] [wide|long]     (on an MI355X)
wide: up to 24 states, documents up to 400 bytes with invalid UTF-8 and 3- and 4-byte runes
long: four more documents of 40-300 KB glued from the others (segment compaction, thousands of lanes per document)"""
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
wide = len(sys.argv) > 3 and sys.argv[3] == "wide"
long_ = len(sys.argv) > 3 and sys.argv[3] == "long"
RAW = (b"\xff", b"\xc3", b"\xe2\x82\xac", b"\xf0\x9f\x98\x80", b"\xe2\x82", b"\x80")
runs = docs_checked = 0
for seed in range(first, first + n_auto):
    rng = np.random.default_rng(seed)
    arcs = craft.random_automaton(rng, 24) if wide else craft.random_automaton(rng)
    docs = craft.random_documents(rng, 160, 400, RAW) if wide else craft.random_documents(rng)
    if long_:
        docs = docs + [b"".join(docs[int(i)] for i in rng.integers(0, len(docs), size=int(k))) for k in (1200, 2500, 5000, 9000)]
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
