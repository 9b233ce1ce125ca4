"""Times dtk_batch_render_device on the bench batch (config 2): sizes pass + bytes pass, per writer mode."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import datok_amd
from datok_amd import corpus

MODELS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "models")
docs = int(os.environ.get("DOCS", 4096))
text, off = corpus.german_docs(docs, 4096, seed=2)
tok = datok_amd.load_tokenizer_file(os.path.join(MODELS, "tokenizer_de.matok"))
with datok_amd.Batch(len(text), docs) as b:
    b.set_input(text, off)
    b.run(tok, 0)
    b.sync()
    t = b.totals()
    print("tokens", t["n_tokens"], "sent ints", t["n_sent"])
    for bits in (3, 1, 7, 15, 12, 3):
        best = 1e9
        for rep in range(6):
            b.run(tok, 0)          # a new run invalidates the rendering
            b.totals()
            t0 = time.perf_counter()
            v = b.render_device(bits)
            b.sync()
            best = min(best, time.perf_counter() - t0)
        print("bits %2d: %8.1f us  out %d bytes  (%.1f GB/s of input)" % (bits, best * 1e6, v.total, len(text) / best / 1e9))
