#!/usr/bin/env python3
"""dtk_multi against dtk_pipeline on one box: the bench corpus as 24 slices from page-locked memory, rune offsets brought
to the host.  (One GPU: the device is listed once, twice, three times -- the workers share the link and the chip; the
figure shows what the threading and the hand-over cost, not scaling.)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import datok_amd
from datok_amd import corpus
model = os.path.join(ROOT, "tests", "golden", "models", "tokenizer_de.matok")
inputs = [corpus.german_docs(4096, 4096, seed=2 + k) for k in range(3)]
total = int(inputs[0][1][-1]); n_slices = 24
pin = datok_amd.PinnedBuffer(total * n_slices)
for i in range(n_slices):
    pin.array[i * total:(i + 1) * total] = inputs[i % 3][0]
big_off = np.concatenate([inputs[i % 3][1][(1 if i else 0):] + np.uint64(i * total) for i in range(n_slices)])
B = datok_amd.Batch
fields = B.R_TOK_RUNE | B.R_SENT | B.R_CSR | B.R_STATUS
seen = [0]
def on_slice(first, n, bb):
    r = bb.result(copy=False); seen[0] += int(r.tok_off[-1])
tok = datok_amd.load_tokenizer_file(model)
def timed(run):
    run(); best = 0
    for _ in range(3):
        t0 = time.perf_counter(); run(); best = max(best, total * n_slices / (time.perf_counter() - t0) / 1e9)
    return best
with datok_amd.Pipeline(total, 4096, depth=4) as p:
    p.set_result_fields(fields)
    print("dtk_pipeline, depth 4:            %.1f GB/s" % timed(lambda: p.run(tok, pin.array, big_off, 256 | 512, on_slice)))
for devs in ([0], [0, 0], [0, 0, 0]):
    with datok_amd.MultiPipeline(model, devs, total, 4096, depth=3) as mp:
        mp.set_result_fields(fields)
        print("dtk_multi, devices %-12s %.1f GB/s" % (str(devs) + ":", timed(lambda: mp.run(pin.array, big_off, 256 | 512, on_slice))))
