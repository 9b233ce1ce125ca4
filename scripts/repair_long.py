import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import datok_amd
from datok_amd import corpus
M = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'models')
tok = datok_amd.load_tokenizer_file(os.path.join(M, "tokenizer_de.matok"))
text, _ = corpus.german_docs(4096, 4096, seed=3)
off = np.array([0, len(text)], dtype=np.uint64)
for warm in (48, 32, 16, 4, 0):
    with datok_amd.Batch(len(text), 1) as b:
        b.set_chunking(128, warm, extend=0)
        b.set_input(text, off)
        b.run(tok, 0); b.sync(); b.totals()
        t0 = time.perf_counter()
        b.run(tok, 0); tot = b.totals()
        dt = time.perf_counter() - t0
        print("16 MiB single document, warm %d: %.2f ms, repair rounds %d" % (warm, dt * 1e3, tot["repair_rounds"]), flush=True)
