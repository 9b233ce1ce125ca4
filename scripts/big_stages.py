#!/usr/bin/env python3
"""Stage times of one large batch (every kernel alone on a full chip): where the time of a saturated GPU goes.
usage: big_stages.py [copies of the 16 MiB bench batch, default 32]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import datok_amd  # noqa: E402
from datok_amd import corpus  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 32
model = os.environ.get("MODEL", "tokenizer_de.matok")
tok = datok_amd.load_tokenizer_file(os.path.join(ROOT, "tests", "golden", "models", model))
parts = [corpus.german_docs(4096, 4096, seed=2 + k)[0] for k in range(min(reps, 8))]
text = np.concatenate([parts[k % len(parts)] for k in range(reps)])
n_docs = 4096 * reps
off = np.arange(n_docs + 1, dtype=np.uint64) * np.uint64(4096)
with datok_amd.Batch(len(text), n_docs) as b:
    if os.environ.get("CHUNK"):
        b.set_chunking(int(os.environ["CHUNK"]), 16)
    b.set_input(text, off)
    for _ in range(int(os.environ.get("WARM_RUNS", "6"))):  # (the model learns its hot cells from the first runs)
        b.run(tok, 256); tot = b.totals()
    b.set_profiling(True); b.run(tok, 256); st = b.stage_ms(); b.set_profiling(False)
    total_us = sum(st.values()) * 1e3
    print("%d x 16 MiB, %d lanes, chunk %d, repairs %d: per 16 MiB, us: %s | sum %.1f us = %.1f GB/s" % (
        reps, tot["n_lanes"], tot["chunk_bytes"], tot["repair_rounds"],
        " ".join("%s=%.1f" % (k, v * 1e3 / reps) for k, v in st.items() if v > 0.002), total_us / reps,
        len(text) / total_us / 1e3))
