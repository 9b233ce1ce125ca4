"""Throughput for ONE long document (the CLI / stdin use case), stage by stage."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import datok_amd
from datok_amd import corpus
M = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "models")
tok = datok_amd.load_tokenizer_file(os.path.join(M, "tokenizer_de.matok"))
for mib in (1, 16, 64):
    text, _ = corpus.german_docs(mib * 256, 4096, seed=3)
    off = np.array([0, len(text)], dtype=np.uint64)
    with datok_amd.Batch(len(text), 1) as b:
        b.set_input(text, off)
        b.set_profiling(True)
        b.run(tok, 0); b.sync(); tot = b.totals()
        t0 = time.perf_counter()
        b.run(tok, 0); b.sync()
        dt = time.perf_counter() - t0
        print("%3d MiB in one document: %.3f ms = %.1f GB/s, lanes %d chunk %d, stages %s" % (
            mib, dt * 1e3, len(text) / dt / 1e9, tot["n_lanes"], tot["chunk_bytes"],
            {k: round(v, 3) for k, v in b.stage_ms().items()}), flush=True)
