#!/usr/bin/env python3
"""Throughput of the BASELINE.json configs 1-4 (config 5 is the multi-GPU bench run), each
checked against the oracle on a sample of documents before timing.  Prints one line per config."""
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


def run(name, model, text, off, in_flight=3, steps=30):
    tok = datok_amd.load_tokenizer_file(os.path.join(M, model))
    om = O.Model(os.path.join(M, model))
    n_docs = len(off) - 1
    bs = []
    for _ in range(in_flight):
        b = datok_amd.Batch(max(len(text), 1), n_docs)
        if os.environ.get("CHUNK"):
            b.set_chunking(int(os.environ["CHUNK"]))
        b.set_input(text, off)
        bs.append(b)
    bs[0].run(tok, 0)
    res, tot = bs[0].result(), bs[0].totals()
    step = max(1, n_docs // 128)
    assert_batch_equals_oracle(om, res, text, off, docs=range(0, n_docs, step))
    for b in bs:
        b.run(tok, 0)
    for b in bs:
        b.sync()
    out = {}
    for k in (1, in_flight):
        t0 = time.perf_counter()
        for i in range(steps):
            bs[i % k].run(tok, 0)
        for b in bs[:k]:
            b.sync()
        dt = time.perf_counter() - t0
        out[k] = len(text) * steps / dt / 1e6
    t0 = time.perf_counter()
    c = om.count_batch(text, off, os.cpu_count() or 1)
    cpu = len(text) / (time.perf_counter() - t0) / 1e6
    print("%-46s %9.1f MB  docs %6d  lanes %7d chunk %4d | GPU %8.1f MB/s (1 in flight) %8.1f MB/s (%d in flight) | "
          "CPU port %7.1f MB/s (%d threads) | tokens %d flagged %d repairs %d" % (
              name, len(text) / 1e6, n_docs, tot["n_lanes"], tot["chunk_bytes"], out[1], out[in_flight], in_flight,
              cpu, os.cpu_count() or 1, tot["n_tokens"], tot["n_flagged"], tot["repair_rounds"]), flush=True)
    for b in bs:
        b.close()


if __name__ == "__main__":
    if os.environ.get("ONLY") == "3":
        t, o = corpus.english_zipf_docs(65536, seed=3)
        run("config3 tokenizer_en.matok 64k Zipf 64B-64KiB", "tokenizer_en.matok", t, o, steps=10)
        sys.exit(0)
    t, o = corpus.simple_ascii(1, 1024)
    run("config1 simpletok.matok 1 KiB", "simpletok.matok", t, o, steps=200)
    t, o = corpus.german_docs(4096, 4096, seed=2)
    run("config2 tokenizer_de.matok 4096x4KiB", "tokenizer_de.matok", t, o)
    run("config4 tokenizer_de.datok 4096x4KiB", "tokenizer_de.datok", t, o)
    t, o = corpus.english_zipf_docs(65536, seed=3)
    run("config3 tokenizer_en.matok 64k Zipf 64B-64KiB", "tokenizer_en.matok", t, o, steps=10)
