#!/usr/bin/env python3
"""A blank-free blob of megabytes inside a document (minified code, base64): out of contract for the reference
(its 1024-rune window overflows) -- the document must come back flagged, quickly, and its neighbours untouched."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datok_amd
from datok_amd import corpus
from oracle import oracle as O
from parity import assert_batch_equals_oracle
M = os.path.join(ROOT, "tests", "golden", "models")
for name in ("tokenizer_de.matok", "tokenizer_de.datok"):
    tok = datok_amd.load_tokenizer_file(os.path.join(M, name)); om = O.Model(os.path.join(M, name))
    for mb in (0.02, 1, 16):
        n = int(mb * (1 << 20))
        rng = np.random.default_rng(3)
        blob = bytes(rng.choice(np.frombuffer(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789+/", dtype=np.uint8), size=n))
        docs = [b"Davor ein Satz. Und noch einer.", b"Anfang " + blob + b" Ende. Danach.", b"Danach ein Dokument."]
        text, off = corpus.concat_docs(docs)
        for chunk in (None, 0):
            if chunk == 0 and mb > 1:
                continue
            with datok_amd.Batch(len(text), len(off) - 1) as b:
                if chunk is not None:
                    b.set_chunking(chunk, 48)
                b.set_input(text, off)
                t0 = time.perf_counter()
                b.run(tok, 0); tot = b.totals(); res = b.result()
                dt = time.perf_counter() - t0
                st = [int(x) for x in res.status]
                assert st[1] & datok_amd.ST_WINDOW_OVERFLOW and st[0] == 0 and st[2] == 0, st
                assert_batch_equals_oracle(om, res, text, off, docs=[0, 2])
                print("%s blob %5.2f MB chunk %s: %.1f ms, repair rounds %d, status %s, lanes %d" % (name, mb, chunk, dt * 1e3, tot["repair_rounds"], st, tot["n_lanes"]))
