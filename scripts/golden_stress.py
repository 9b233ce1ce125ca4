#!/usr/bin/env python3
"""A stress corpus made of the reference's own test inputs (XML tags, URLs, e-mail addresses, abbreviations,
emoticons, clitics ...), shuffled into 4 KiB documents: how often does the chunk speculation need a repair
round there, for the default warm-up and for the experimental whitespace-guided one?  Every document is
checked against the oracle."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import datok_amd  # noqa: E402
from datok_amd import corpus  # noqa: E402
from oracle import oracle as O  # noqa: E402
from parity import assert_batch_equals_oracle  # noqa: E402

M = os.path.join(ROOT, "tests", "golden", "models")
G = os.path.join(ROOT, "tests", "golden")
inputs = []
for stem in ("matrix", "datok", "token_writer"):
    d = json.load(open(os.path.join(G, stem + "_goldens.json"), encoding="utf-8"))
    for case in d["cases"]:
        for c in case["calls"]:
            s = c["input"].replace("\x04", " ")
            if len(s) > 3:
                inputs.append(s)
inputs = sorted(set(inputs))
print(len(inputs), "distinct inputs,", sum(len(s.encode()) for s in inputs), "bytes")
n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(11)
docs = []
for _ in range(n_docs):
    parts, n = [], 0
    while n < 4000:
        s = inputs[int(rng.integers(0, len(inputs)))]
        parts.append(s)
        n += len(s.encode()) + 1
    docs.append((str(rng.choice([" ", "\n", "  "]))).join(parts).encode()[:4096])
text, off = corpus.concat_docs(docs)
for model in ("tokenizer_de.matok", "tokenizer_en.matok"):
    tok = datok_amd.load_tokenizer_file(os.path.join(M, model))
    om = O.Model(os.path.join(M, model))
    for chunk, warm in ((None, 48), (128, 32), (128, 24), (128, 16), (128, 64)):
        with datok_amd.Batch(len(text), n_docs) as b:
            if chunk is not None:
                b.set_chunking(chunk, warm, extend=int(os.environ.get('EXTEND', '240')))
            b.set_input(text, off)
            b.run(tok, 0)
            tot = b.totals()
            res = b.result()
            assert_batch_equals_oracle(om, res, text, off, docs=range(0, n_docs, max(1, n_docs // 512)))
            print("%s chunk %s warm %d (DATOK_WARM_WS=%s MIN=%s): lanes %d, repair rounds %d, flagged %d" % (
                model, chunk, warm, os.environ.get("DATOK_WARM_WS", "0"), os.environ.get("DATOK_WARM_MIN", "0"),
                tot["n_lanes"], tot["repair_rounds"], tot["n_flagged"]), flush=True)
