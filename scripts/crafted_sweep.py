#!/usr/bin/env python3
"""The crafted tokenizers of tests/craft.py (a double array that consumes an EOT twice, three epsilon SentenceEnds at
one cursor) over chunk sizes, warm-up distances and flags: every document against the oracle, repair rounds and
the fallback reported.  Run on an MI355X, also with DATOK_SPLIT_START=1 / DATOK_NO_DENSE=1 / DATOK_DEV_ROUNDS=2."""
import gzip
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import craft  # noqa: E402
import datok_amd  # noqa: E402
from datok_amd import corpus  # noqa: E402
from oracle import oracle as O  # noqa: E402
from parity import assert_batch_equals_oracle  # noqa: E402

n_runs = fallbacks = max_rounds = 0
for kind in ("datok", "matok"):
    for triple in (False, True):
        blob = getattr(craft, kind)(triple)
        path = "/tmp/crafted_%s_%d.%s" % (kind, triple, kind)
        open(path, "wb").write(blob)
        tok, om = datok_amd.load_tokenizer_file(path), O.Model(raw=gzip.decompress(blob))
        for seed in (5, 6, 7, 105, 106):
            docs = craft.documents(np.random.default_rng(seed % 100))
            if seed >= 100:  # long documents (segments of 64 lanes at 16-byte chunks): 40 short ones glued together
                rng = np.random.default_rng(seed)
                docs = [b"".join(docs[int(i)] for i in rng.integers(0, len(docs), size=40)) for _ in range(24)]
            text, off = corpus.concat_docs(docs)
            for chunk in (16, 17, 24, 33, 64, 128):
                for warm in (0, 2, 8, 32):
                    for flags in (0, 16):
                        with datok_amd.Batch(max(len(text), 1), len(docs)) as b:
                            b.set_chunking(chunk, warm, extend=0 if warm < 8 else 16)
                            b.set_input(text, off)
                            for _ in range(2):  # the second run with what the first one learnt (device rounds, EOT kernel)
                                b.run(tok, flags)
                                res, tot = b.result(), b.totals()
                                assert_batch_equals_oracle(om, res, text, off, flags)
                                if chunk in (16, 33) and warm == 2:  # the writer's bytes, rendered on the device
                                    for bits in (3, 15):
                                        data, o = b.render(bits | flags)
                                        for d, doc in enumerate(docs):
                                            exp, est = om.transduce(doc, bits | flags)
                                            if est == 0 and not (int(res.status[d]) & ~datok_amd.ST_EMPTY_TEXT):
                                                assert data[int(o[d]):int(o[d + 1])] == exp, (kind, triple, seed, chunk, bits, d, doc)
                                n_runs += 1
                                fallbacks += tot["chunk_bytes"] != chunk
                                max_rounds = max(max_rounds, tot["repair_rounds"])
print("CRAFTED OK: %d runs, %d fell back to one lane per document, at most %d repair rounds" % (n_runs, fallbacks, max_rounds))
