#!/usr/bin/env python3
"""Where the time goes in the host-fed loop: set_input / run / totals per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import datok_amd
from datok_amd import corpus
tok = datok_amd.load_tokenizer_file(os.path.join(ROOT, "tests", "golden", "models", "tokenizer_de.matok"))
inputs = [corpus.german_docs(4096, 4096, seed=2 + k) for k in range(3)]
total = len(inputs[0][0])
for pin_off in (False, True):
    pinned = [(torch.from_numpy(t).pin_memory().numpy(), torch.from_numpy(o.view(np.int64)).pin_memory().numpy().view(np.uint64) if pin_off else o) for t, o in inputs]
    hb = [datok_amd.Batch(total, 4096) for _ in range(3)]
    for bb, (t, o) in zip(hb, pinned):
        bb.set_input(t, o); bb.run(tok, 256); bb.totals()
    acc = {"totals": 0.0, "set_input": 0.0, "run": 0.0}
    n = 30
    t00 = time.perf_counter()
    for i in range(n):
        k = i % 3
        t0 = time.perf_counter(); hb[k].totals(); t1 = time.perf_counter()
        hb[k].set_input(*pinned[k]); t2 = time.perf_counter()
        hb[k].run(tok, 256); t3 = time.perf_counter()
        acc["totals"] += t1 - t0; acc["set_input"] += t2 - t1; acc["run"] += t3 - t2
    for bb in hb: bb.totals()
    el = time.perf_counter() - t00
    print("offsets pinned" if pin_off else "offsets pageable", "%.1f GB/s" % (total * n / el / 1e9), {k: round(v / n * 1e6, 1) for k, v in acc.items()}, "us per step")
    for bb in hb: bb.close()
