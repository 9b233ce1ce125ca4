#!/usr/bin/env python3
"""Stage times of config 3 (tokenizer_en.matok, 65 536 Zipf-length documents, one batch)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import datok_amd  # noqa: E402
from datok_amd import corpus  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
tok = datok_amd.load_tokenizer_file(os.path.join(ROOT, "tests", "golden", "models", "tokenizer_en.matok"))
text, off = corpus.english_zipf_docs(n, seed=3)
with datok_amd.Batch(len(text), n) as b:
    b.set_input(text, off)
    b.run(tok, 256); tot = b.totals()
    b.set_profiling(True); b.run(tok, 256); st = b.stage_ms(); b.set_profiling(False)
    us = sum(st.values()) * 1e3
    print("%d docs, %.1f MB, %d lanes, chunk %d, %d tokens: us: %s | sum %.0f us = %.1f GB/s" % (
        n, len(text) / 1e6, tot["n_lanes"], tot["chunk_bytes"], tot["n_tokens"],
        " ".join("%s=%.0f" % (k, v * 1e3) for k, v in st.items() if v > 0.002), us, len(text) / us / 1e3))
