#!/usr/bin/env python3
"""Speculation on tag-heavy text (XML tags with blanks inside, 30-90 bytes, every few hundred bytes): repair rounds
per warm-up distance, with the blank-guided start."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datok_amd
from datok_amd import corpus
from oracle import oracle as O
from parity import assert_batch_equals_oracle
M = os.path.join(ROOT, "tests", "golden", "models")
rng = np.random.default_rng(11)
text, off = corpus.german_docs(1024, 4096, seed=4)
raw = bytearray(text.tobytes())
tags = [b'<span class="foo bar" id="x12" lang="de">', b'</span>', b'<a href="http://www.example.org/a/b?c=d" target="_blank" rel="nofollow noopener">', b'</a>',
        b'<w lemma="gehen" pos="VVFIN" msd="3 Sg Pres Ind">', b'</w>', b'<br />', b'<img src="bild.png" alt="ein Bild mit Text" width="100" height="50" />']
n = 0
for p in range(300, len(raw) - 300, int(sys.argv[1]) if len(sys.argv) > 1 else 400):
    q = raw.find(b" ", p)
    t = tags[int(rng.integers(0, len(tags)))]
    if q < 0 or (q % 4096) + len(t) + 2 > 4000:
        continue
    raw[q + 1:q + 1 + len(t)] = t
    raw[q + 1 + len(t)] = 0x20
    n += 1
text = np.frombuffer(bytes(raw), dtype=np.uint8).copy()
for name in ("tokenizer_de.matok", "tokenizer_en.matok"):
    tok = datok_amd.load_tokenizer_file(os.path.join(M, name)); om = O.Model(os.path.join(M, name))
    for warm in (48, 32, 24, 16, 8):
        for extend in (240, 0):
            with datok_amd.Batch(len(text), len(off) - 1) as b:
                b.set_chunking(128, warm, extend=extend)
                b.set_input(text, off); b.run(tok, 0)
                tot = b.totals(); res = b.result()
                assert_batch_equals_oracle(om, res, text, off, docs=range(0, 1024, 37))
                print("%s: %d tags in %d KB, warm %2d extend %3d: repair rounds %d" % (name, n, len(text) >> 10, warm, extend, tot["repair_rounds"]))
