import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import datok_amd
from datok_amd import corpus
from test_gpu_parity import _edge_docs
tok = datok_amd.load_tokenizer_file("tests/golden/models/tokenizer_de.matok")
docs=_edge_docs()
text,off=corpus.concat_docs(docs)
for chunk in (0,16):
    with datok_amd.Batch(max(len(text),1), len(docs)) as b:
        b.set_chunking(chunk,64); b.set_input(text,off); b.run(tok,0); r=b.result(); t=b.totals()
    bad=[d for d in range(len(docs)) if r.status[d]&32]
    print(chunk, t, "internal docs", bad[:20])
    for d in bad[:5]:
        print(d, docs[d][:60], r.status[d], r.tok_off[d:d+2], r.sent_off[d:d+2], r.text_off[d:d+2])
