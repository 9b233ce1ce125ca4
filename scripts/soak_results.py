#!/usr/bin/env python3
"""Soak of the round-3 surfaces: random corpora through dtk_pipeline / dtk_multi with random slice sizes, depths, result
field masks and offset flags -- every slice's host arrays (page-locked buffers) against the oracle, bit exact.
    python scripts/soak_results.py [seeds] [first_seed]          (on an MI355X)"""
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
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
first = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
B = datok_amd.Batch
RUNE = ("tok_rstart", "tok_rend", "sent", "text_tok_end", "text_sent_end")
BYTE = ("tok_bstart", "tok_bend", "sent", "text_tok_end", "text_sent_end")
docs_checked = 0
t00 = time.time()
for seed in range(first, first + n_seeds):
    rng = np.random.default_rng(seed)
    model = ["tokenizer_de.matok", "tokenizer_en.matok", "tokenizer_de.datok"][seed % 3]
    gen = [lambda s: corpus.german_rich_docs(int(rng.integers(300, 1500)), int(rng.integers(200, 3000)), seed=s),
           lambda s: corpus.english_zipf_docs(int(rng.integers(300, 2500)), seed=s, max_bytes=int(2 ** rng.integers(8, 15))),
           lambda s: corpus.german_docs(int(rng.integers(300, 1500)), int(rng.integers(100, 2000)), seed=s)][int(rng.integers(0, 3))]
    text, off = gen(seed)
    # a few EOT texts and empty documents in between
    docs = [text[int(off[d]):int(off[d + 1])].tobytes() for d in range(len(off) - 1)]
    for k in rng.integers(0, len(docs), 6):
        docs[int(k)] = docs[int(k)][:40] + b"\n\x04\n" + docs[int(k)][40:80] + b"\x04"
    for k in rng.integers(0, len(docs), 4):
        docs[int(k)] = b""
    text, off = corpus.concat_docs(docs)
    om = O.Model(os.path.join(M, model))
    slice_bytes = max(int(rng.integers(1 << 16, 1 << 21)), max(len(d) for d in docs) + 1)
    slice_docs = int(rng.integers(50, 600))
    depth = int(rng.integers(1, 6))
    run_flags = [0, 16, 256, 256 | 512, 256 | 1024][int(rng.integers(0, 5))]
    fields_all = [0, B.R_ALL, B.R_TOK_RUNE | B.R_SENT | B.R_CSR | B.R_STATUS | B.R_TEXTS,
                  B.R_TOK_BYTE | B.R_SENT | B.R_CSR | B.R_STATUS | B.R_TEXTS | B.R_EVENTS,
                  B.R_TOK_RUNE16 | B.R_SENT | B.R_CSR | B.R_STATUS | B.R_TEXTS,   # (int16 pairs, or the 32-bit arrays if a document is long)
                  B.R_TOK_RUNE16 | B.R_ALL][int(rng.integers(0, 6))]
    cmp_fields = tuple(f for f in ("tok_rstart", "tok_rend", "tok_bstart", "tok_bend", "sent", "text_tok_end", "text_sent_end")
                       if not ((run_flags & 512) and f.startswith("tok_b")) and not ((run_flags & 1024) and f.startswith("tok_r"))
                       and (fields_all in (0, B.R_ALL) or not (f.startswith("tok_r") and not fields_all & (B.R_TOK_RUNE | B.R_TOK_RUNE16))
                            and not (f.startswith("tok_b") and not fields_all & B.R_TOK_BYTE)))
    seen = [0]

    def on_slice(first_d, n, b):
        global docs_checked
        assert first_d == seen[0]
        seen[0] += n
        res = b.result(copy=bool(rng.integers(0, 2)))
        sub_off = (off[first_d:first_d + n + 1] - off[first_d]).astype(np.uint64)
        sub = text[int(off[first_d]):int(off[first_d + n])]
        docs_checked += assert_batch_equals_oracle(om, res, sub, sub_off, run_flags & 16, docs=range(0, n, 3), fields=cmp_fields,
                                                   allow_status=2)
    kind = int(rng.integers(0, 3))
    if kind < 2:
        tok = datok_amd.load_tokenizer_file(os.path.join(M, model))
        with datok_amd.Pipeline(slice_bytes, slice_docs, depth=depth) as p:
            if fields_all:
                p.set_result_fields(fields_all)
            for rep in range(2):
                seen[0] = 0
                p.run(tok, text, off, run_flags, on_slice)
                assert seen[0] == len(docs)
    else:
        with datok_amd.MultiPipeline(os.path.join(M, model), [0] * int(rng.integers(1, 4)), slice_bytes, slice_docs,
                                     depth=min(depth, 3)) as mp:
            if fields_all:
                mp.set_result_fields(fields_all)
            for rep in range(2):
                seen[0] = 0
                mp.run(text, off, run_flags, on_slice)
                assert seen[0] == len(docs)
    print("seed %d ok: %s, %d docs, %s, slices of %d B / %d docs, depth %d, flags %d, fields %d" % (
        seed, model, len(docs), ("pipeline", "pipeline", "multi")[kind], slice_bytes, slice_docs, depth, run_flags, fields_all), flush=True)
print("SOAK OK: %d seeds, %d documents compared, %.0f s" % (n_seeds, docs_checked, time.time() - t00))
