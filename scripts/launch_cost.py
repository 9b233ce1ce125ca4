"""Host-side cost of dtk_batch_run (11 asynchronous launches) vs the device time per batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import datok_amd
from datok_amd import corpus
MODELS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "models")
text, off = corpus.german_docs(4096, 4096, seed=2)
tok = datok_amd.load_tokenizer_file(os.path.join(MODELS, "tokenizer_de.matok"))
for ns in (1, 2, 3, 4, 6, 8):
    bs = [datok_amd.Batch(len(text), 4096) for _ in range(ns)]
    for b in bs:
        b.set_input(text, off); b.run(tok, 0); b.sync(); b.totals()
    K = 120
    t0 = time.perf_counter(); host = 0.0
    for i in range(K):
        h0 = time.perf_counter()
        bs[i % ns].run(tok, 0)
        host += time.perf_counter() - h0
    for b in bs: b.sync()
    dt = time.perf_counter() - t0
    print("streams %d: %.1f us/batch wall, host launch %.1f us/batch, %.1f GB/s" % (ns, dt / K * 1e6, host / K * 1e6, len(text) / (dt / K) / 1e9))
    for b in bs: b.close()
