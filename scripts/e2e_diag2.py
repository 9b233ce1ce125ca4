#!/usr/bin/env python3
"""Where does a pipeline step's time go when results come to the host?  (round 3)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("WITH_TORCH"):
    import torch
    _t = torch.zeros(1 << 20, device="cuda"); torch.cuda.synchronize()
import datok_amd
from datok_amd import corpus

tok = datok_amd.load_tokenizer_file(os.path.join(ROOT, "tests", "golden", "models", "tokenizer_de.matok"))
inputs = [corpus.german_docs(4096, 4096, seed=2 + k) for k in range(3)]
total = int(inputs[0][1][-1]); n_docs = 4096; n_slices = 24
pin = datok_amd.PinnedBuffer(total * n_slices)
for i in range(n_slices):
    pin.array[i * total:(i + 1) * total] = inputs[i % 3][0]
big_off = np.concatenate([inputs[i % 3][1][(1 if i else 0):] + np.uint64(i * total) for i in range(n_slices)])
B = datok_amd.Batch
CASES = ((0, 3), (B.R_TOK_RUNE | B.R_SENT | B.R_CSR | B.R_STATUS, 3), (B.R_TOK_RUNE | B.R_SENT | B.R_CSR | B.R_STATUS, 4),
                      (B.R_TOK_RUNE | B.R_SENT | B.R_CSR | B.R_STATUS, 6), (B.R_EVENTS | B.R_CSR | B.R_STATUS, 4))
if os.environ.get("E2E_ONLY"):
    CASES = ((B.R_TOK_RUNE | B.R_SENT | B.R_CSR | B.R_STATUS, 3),)
import ctypes
if os.environ.get("E2E_CASES"):
    CASES = tuple((B.R_TOK_RUNE | B.R_SENT | B.R_CSR | B.R_STATUS, int(d)) for d in os.environ["E2E_CASES"].split(","))
for fields, depth in CASES:
    pipe = datok_amd.Pipeline(total, n_docs, depth=depth)
    if fields:
        pipe.set_result_fields(fields)
    stamps = []

    def on_slice(first, n, bb):
        a = time.perf_counter()
        bb.totals()
        b_ = time.perf_counter()
        if fields:
            bb.result(copy=False)
        stamps.append((a, b_, time.perf_counter()))
    pipe.run(tok, pin.array, big_off, 256, on_slice)
    for rep in range(2):
        stamps.clear()
        t0 = time.perf_counter()
        pipe.run(tok, pin.array, big_off, 256, on_slice)
        e = time.perf_counter() - t0
        gaps = [stamps[i + 1][0] - stamps[i][2] for i in range(len(stamps) - 1)]
        print("fields %3d depth %d: %.1f GB/s, %.3f ms/slice; in callback: totals %.3f ms, result wait %.3f ms; between callbacks %.3f ms" % (
            fields, depth, total * n_slices / e / 1e9, e / n_slices * 1e3,
            np.mean([s[1] - s[0] for s in stamps]) * 1e3, np.mean([s[2] - s[1] for s in stamps]) * 1e3, np.mean(gaps) * 1e3))
    pipe.close()
