#!/usr/bin/env python3
"""One batch of ~1.25 GiB (a config-5 shard) and one of ~3.4 GiB (close to the batch limit): totals must be
the multiple of the 16 MiB batch's, sampled documents must equal the oracle's.  Run on an MI355X."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import datok_amd  # noqa: E402
from datok_amd import corpus  # noqa: E402
from oracle import oracle as O  # noqa: E402
from parity import assert_batch_equals_oracle  # noqa: E402

M = os.path.join(ROOT, "tests", "golden", "models")
tok = datok_amd.load_tokenizer_file(os.path.join(M, "tokenizer_de.matok"))
om = O.Model(os.path.join(M, "tokenizer_de.matok"))
text1, off1 = corpus.german_docs(4096, 4096, seed=5)
with datok_amd.Batch(len(text1), 4096) as b:
    b.set_input(text1, off1); b.run(tok, 0); base = b.totals()
for reps in (80, 215):
    text = np.tile(text1, reps)
    n_docs = 4096 * reps
    off = (np.arange(n_docs + 1, dtype=np.uint64) * np.uint64(4096))
    t0 = time.time()
    with datok_amd.Batch(len(text), n_docs) as b:
        b.set_input(text, off)
        t1 = time.time()
        b.run(tok, 0)
        tot = b.totals()
        t2 = time.time()
        b.run(tok, 0); b.sync()
        t3 = time.time()
        assert tot["n_tokens"] == base["n_tokens"] * reps and tot["n_sent"] == base["n_sent"] * reps, (tot, base)
        assert tot["n_flagged"] == 0 and tot["repair_rounds"] == 0
        res = b.result()
        sample = list(range(0, 32)) + list(range(n_docs // 2, n_docs // 2 + 32)) + list(range(n_docs - 32, n_docs))
        assert_batch_equals_oracle(om, res, text, off, docs=sample)
        print("%.2f GiB, %d docs, %d lanes, chunk %d: upload %.2f s, first run %.3f s, second run %.3f s = %.1f GB/s; "
              "%d tokens; sampled documents equal the oracle" % (
                  len(text) / 2**30, n_docs, tot["n_lanes"], tot["chunk_bytes"], t1 - t0, t2 - t1, t3 - t2,
                  len(text) / (t3 - t2) / 1e9, tot["n_tokens"]), flush=True)
    del text, res
