#!/usr/bin/env python3
"""Throughput of the bench batch over (chunk bytes, batches in flight), completion (totals) inside the clock.
usage: sweep2.py [chunks] [streams]   e.g.  sweep2.py 64,96,128,192,256 1,2,3,4"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import datok_amd  # noqa: E402
from datok_amd import corpus  # noqa: E402

chunks = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "64,96,128,192,256").split(",")]
streams = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,3,4").split(",")]
model = os.environ.get("MODEL", "tokenizer_de.matok")
tok = datok_amd.load_tokenizer_file(os.path.join(ROOT, "tests", "golden", "models", model))
inputs = [corpus.german_docs(4096, 4096, seed=2 + k) for k in range(max(streams))]
total = len(inputs[0][0])
steps = int(os.environ.get("STEPS", "60"))
for c in chunks:
    bs = []
    for t, o in inputs:
        b = datok_amd.Batch(total, 4096)
        b.set_chunking(c, int(os.environ.get("WARM", "16")))
        b.set_input(t, o)
        bs.append(b)
    for b in bs:
        b.run(tok, 256); b.totals()
    row = []
    for s in streams:
        best = 0.0
        for rep in range(3):
            ran = [False] * s
            t0 = time.perf_counter()
            for i in range(steps):
                k = i % s
                if ran[k]:
                    bs[k].totals()
                bs[k].run(tok, 256)
                ran[k] = True
            for k in range(s):
                bs[k].totals()
            best = max(best, total * steps / (time.perf_counter() - t0) / 1e9)
        row.append(best)
    tot = bs[0].totals()
    bs[0].set_profiling(True); bs[0].run(tok, 256); st = bs[0].stage_ms(); bs[0].set_profiling(False)
    print("chunk %4d lanes %7d lookups %9d repairs %d | GB/s by batches in flight %s | one batch stages us: %s" % (
        c, tot["n_lanes"], tot["walk_steps"], tot["repair_rounds"], " ".join("%d:%6.1f" % (s, v) for s, v in zip(streams, row)),
        " ".join("%s=%.0f" % (k, v * 1e3) for k, v in st.items() if v > 0.002)), flush=True)
    for b in bs:
        b.close()
