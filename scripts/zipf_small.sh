cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys, time
sys.path.insert(0, '.')
import datok_amd
from datok_amd import corpus
tok = datok_amd.load_tokenizer_file("tests/golden/models/tokenizer_en.matok")
t, o = corpus.english_zipf_docs(8192, seed=2, max_bytes=16384)
for sm in ("0", "160", "320"):
    os.environ["DATOK_SMALL_MAX"] = sm
PY
for sm in 0 160; do DATOK_SMALL_MAX=$sm python - <<'PY'
import os, sys, time
sys.path.insert(0, '.')
import datok_amd
from datok_amd import corpus
tok = datok_amd.load_tokenizer_file("tests/golden/models/tokenizer_en.matok")
t, o = corpus.english_zipf_docs(8192, seed=2, max_bytes=16384)
b = datok_amd.Batch(len(t), 8192); b.set_input(t, o)
b.run(tok, 256); b.totals()
b.set_profiling(True)
for i in range(3):
    b.run(tok, 256); st = b.stage_ms(); b.totals()
print("SMALL_MAX", os.environ["DATOK_SMALL_MAX"], {k: round(v*1e3) for k, v in st.items()})
PY
done
