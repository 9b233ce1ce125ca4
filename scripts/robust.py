#!/usr/bin/env python3
"""Bench corpus vs robustness corpus vs English: MB/s (1 and 3 batches in flight, completion inside the clock),
lookups per byte, repair rounds.  usage: robust.py [model]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import datok_amd  # noqa: E402
from datok_amd import corpus  # noqa: E402

M = os.path.join(ROOT, "tests", "golden", "models")
steps = int(os.environ.get("STEPS", "45"))


def measure(name, model, gen):
    tok = datok_amd.load_tokenizer_file(os.path.join(M, model))
    inputs = [gen(2 + k) for k in range(3)]
    total = len(inputs[0][0])
    bs = []
    for t, o in inputs:
        b = datok_amd.Batch(total, len(o) - 1)
        b.set_input(t, o)
        b.run(tok, 256); b.totals()
        bs.append(b)
    out = []
    for s in (1, 3):
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
        out.append(best)
    tot = bs[0].totals()
    bs[0].set_profiling(True); bs[0].run(tok, 256); st = bs[0].stage_ms(); bs[0].totals(); bs[0].set_profiling(False)
    print("%-28s %-20s %6.1f MB | %6.1f GB/s alone %6.1f GB/s 3 in flight | lookups/byte %.3f tokens/byte %.3f repairs %d chunk %d | %s"
          % (name, model, total / 1e6, out[0], out[1], tot["walk_steps"] / total, tot["n_tokens"] / total, tot["repair_rounds"],
             tot["chunk_bytes"], " ".join("%s=%.0f" % (k, v * 1e3) for k, v in st.items() if v > 0.006)), flush=True)
    for b in bs:
        b.close()


for model in (sys.argv[1:] or ["tokenizer_de.matok"]):
    measure("bench (650 word types)", model, lambda s: corpus.german_docs(4096, 4096, seed=s))
    measure("rich (30k types, tags, URLs)", model, lambda s: corpus.german_rich_docs(4096, 4096, seed=s))
    measure("rich, tags x4", model, lambda s: corpus.german_rich_docs(4096, 4096, seed=s, p_special=0.05))
measure("english zipf 8k docs", "tokenizer_en.matok", lambda s: corpus.english_zipf_docs(8192, seed=s, max_bytes=16384))
