#!/usr/bin/env python3
"""A few runs of the rich corpus (repair rounds) for a kernel trace: rocprofv3 --kernel-trace --stats -- python scripts/prof_rich.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import datok_amd
from datok_amd import corpus
tok = datok_amd.load_tokenizer_file(os.path.join(ROOT, "tests", "golden", "models", "tokenizer_de.matok"))
t, o = corpus.german_rich_docs(4096, 4096, seed=2)
b = datok_amd.Batch(len(t), 4096)
b.set_input(t, o)
for i in range(12):
    b.run(tok, 256); tot = b.totals()
print(tot)
