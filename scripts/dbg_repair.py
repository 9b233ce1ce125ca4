import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import craft, datok_amd
from datok_amd import corpus
blob = craft.datok(False)
open("/tmp/crafted.datok", "wb").write(blob)
tok = datok_amd.load_tokenizer_file("/tmp/crafted.datok")
docs = craft.documents(np.random.default_rng(5))
text, off = corpus.concat_docs(docs)
with datok_amd.Batch(max(len(text), 1), len(docs)) as b:
    b.set_chunking(16, 8, extend=0)
    b.set_input(text, off)
    b.run(tok, 16)
    tot = b.totals()
    print(tot)
    res = b.result()
    bad = [d for d in range(len(docs)) if False]
for d in (int(x) for x in os.environ.get("SHOW", "").split(",") if x):
    print(d, docs[d])
