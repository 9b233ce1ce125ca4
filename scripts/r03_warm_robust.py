#!/usr/bin/env python3
"""Warm-up distance 16 vs 8 on the three corpora: GB/s (one batch / three in flight), repair rounds, lookups per byte."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import datok_amd
from datok_amd import corpus
M = os.path.join(ROOT, "tests", "golden", "models")
cases = [("bench de", "tokenizer_de.matok", lambda s: corpus.german_docs(4096, 4096, seed=s)),
         ("rich de", "tokenizer_de.matok", lambda s: corpus.german_rich_docs(4096, 4096, seed=s)),
         ("rich de x4 tags", "tokenizer_de.matok", lambda s: corpus.german_rich_docs(4096, 4096, seed=s, p_special=0.048)),
         ("zipf en", "tokenizer_en.matok", lambda s: corpus.english_zipf_docs(8192, seed=s))]
for name, model, gen in cases:
    tok = datok_amd.load_tokenizer_file(os.path.join(M, model))
    inputs = [gen(2 + k) for k in range(3)]
    for warm in (8, 4, 2, 0):
        bs = []
        for t, o in inputs:
            b = datok_amd.Batch(len(t), len(o) - 1)
            b.set_chunking(datok_amd.Batch.AUTO_CHUNK, warm)
            b.set_input(t, o); b.run(tok, 256 | 512); b.totals()
            bs.append(b)
        out = []
        for s in (1, 3):
            best = 0.0
            for rep in range(3):
                ran = [False] * s
                t0 = time.perf_counter()
                for i in range(45):
                    k = i % s
                    if ran[k]: bs[k].totals()
                    bs[k].run(tok, 256 | 512); ran[k] = True
                for k in range(s): bs[k].totals()
                best = max(best, 45 * len(inputs[0][0]) / (time.perf_counter() - t0) / 1e9)
            out.append(best)
        tot = bs[0].totals()
        print("%-16s warm %2d: %.1f / %.1f GB/s, repair rounds %d, lookups/byte %.3f" % (name, warm, out[0], out[1], tot["repair_rounds"], tot["walk_steps"] / tot["n_bytes"]))
        for b in bs: b.close()
