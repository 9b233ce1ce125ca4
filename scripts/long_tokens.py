#!/usr/bin/env python3
"""Speculation on text with long blank-free tokens (URLs longer than the warm-up): repair rounds and
throughput when every run is finished (results fetched), 3 batches in flight."""
import os, sys, time
import numpy as np
import torch
torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datok_amd
from datok_amd import corpus
from oracle import oracle as O
from parity import assert_batch_equals_oracle
M = os.path.join(ROOT, "tests", "golden", "models")
tok = datok_amd.load_tokenizer_file(os.path.join(M, "tokenizer_de.matok"))
om = O.Model(os.path.join(M, "tokenizer_de.matok"))
rng = np.random.default_rng(7)
text, off = corpus.german_docs(4096, 4096, seed=2)
raw = bytearray(text.tobytes())
every = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n_urls = 0
for p in range(500, len(raw) - 400, every):
    q = raw.find(b" ", p)
    if q < 0 or (q % 4096) > 3700:
        continue
    L = int(rng.integers(60, 160))
    url = (b"https://www.example.org/" + bytes(rng.choice(list(b"abcdefghijklmnopqrstuvwxyz0123456789/_-%"), size=L)))[:L]
    raw[q + 1:q + 1 + len(url)] = url + b" " * 0
    raw[q + 1 + len(url)] = 0x20
    n_urls += 1
text = np.frombuffer(bytes(raw), dtype=np.uint8).copy()
bs = [datok_amd.Batch(len(text), 4096) for _ in range(3)]
t_text = torch.from_numpy(text).cuda(); t_off = torch.from_numpy(off.view(np.int64)).cuda(); torch.cuda.synchronize()
for b in bs:
    b.set_input_device(t_text.data_ptr(), t_off.data_ptr(), 4096, len(text), keep=(t_text, t_off), doc_off_host=off); b.run(tok, 0)
tot = bs[0].totals()
res = bs[0].result()
assert_batch_equals_oracle(om, res, text, off, docs=range(0, 4096, 64))
print("%d long tokens in %d MB: repair rounds of one run %d, flagged %d" % (n_urls, len(text) >> 20, tot["repair_rounds"], tot["n_flagged"]))
bs[0].set_profiling(True); bs[0].run(tok, 0); print("  stages of one run (ms):", {k: round(v, 3) for k, v in bs[0].stage_ms().items()}); bs[0].set_profiling(False)
for finish in (False, True):
    K = 60
    t0 = time.perf_counter()
    for i in range(K):
        b = bs[i % 3]
        if finish and i >= 3:
            b.totals()          # fetch the previous run's totals: runs the repair rounds if any
        b.run(tok, 0)
    for b in bs:
        b.totals() if finish else b.sync()
    dt = time.perf_counter() - t0
    print("  %s: %.3f ms per batch = %.1f GB/s" % ("results fetched every run" if finish else "runs only (repairs deferred)", dt / K * 1e3, len(text) / (dt / K) / 1e9))
