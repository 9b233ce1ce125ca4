"""Throughput for very many tiny documents (one chunk lane and one compaction wave per document)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import datok_amd
from datok_amd import corpus
M = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "models")
tok = datok_amd.load_tokenizer_file(os.path.join(M, "tokenizer_de.matok"))
text, _ = corpus.german_docs(8192, 4096, seed=3)   # 32 MiB
for size in (64, 256, 1024):
    n = len(text) // size
    off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(size))
    t = text[: n * size]
    with datok_amd.Batch(len(t), n) as b:
        b.set_input(t, off)
        b.set_profiling(True)
        b.run(tok, 0); b.sync(); tot = b.totals()
        t0 = time.perf_counter()
        b.run(tok, 0); b.sync()
        dt = time.perf_counter() - t0
        print("%7d documents of %4d B: %.3f ms = %.1f GB/s, lanes %d, stages %s" % (
            n, size, dt * 1e3, len(t) / dt / 1e9, tot["n_lanes"],
            {k: round(v, 3) for k, v in b.stage_ms().items()}), flush=True)
