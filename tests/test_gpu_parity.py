"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle and
against the reference's golden vectors.  Bit exact: all outputs are integers."""
import io
import os
import sys

import ctypes

import numpy as np
import pytest

from conftest import MODELS, ROOT
from goldens import failed_checks, golden_strings, load_cases, model_file
from parity import assert_batch_equals_oracle, oracle_doc

pytestmark = pytest.mark.gpu
C_byref = ctypes.byref

TOKENS, SENTENCES, TOKEN_POS, SENTENCE_POS, NEWLINE_AFTER_EOT = 1, 2, 4, 8, 16
SIMPLE = TOKENS | SENTENCES


@pytest.fixture(scope="module")
def gpu():
    import datok_amd
    assert datok_amd.lib().dtk_device_count() > 0, "no HIP device: the product path has no CPU fallback"
    cache = {}

    def get(name):
        name = model_file(name)   # `fst:` goldens: the Foma net itself, converted by the loader
        if name not in cache:
            cache[name] = datok_amd.load_tokenizer_file(os.path.join(MODELS, name))
            assert cache[name] is not None
        return cache[name]
    return get


def run_batch(tok, text, doc_off, flags=0, chunk=None, warm=64):
    """chunk=None: library default (automatic chunking); 0: one lane per document."""
    import datok_amd
    with datok_amd.Batch(max(len(text), 1), len(doc_off) - 1) as b:
        if chunk is not None:
            # short warm-ups are there to force mispredictions: no help from the previous blank then
            b.set_chunking(chunk, warm, extend=0 if warm < 16 else None)
        b.set_input(text, doc_off)
        b.run(tok, flags)
        return b.result(), b.totals()


# ---------------------------------------------------------------- golden vectors
CASES = [c for c in load_cases() if not c["stale"]]


def test_goldens_through_cabi_transduce(gpu):
    """Every live expectation of matrix_test.go / datok_test.go / token_writer_test.go,
    rendered by dtk_transduce (GPU walk + C++ host mirror of NewTokenWriter)."""
    n = 0
    for case in CASES:
        out = b""
        for c in case["calls"]:
            o, status = gpu(c["model"]).transduce_bytes(c["input"].encode("utf-8"), c["flags"])
            assert status == 0, (case["src"], status)
            out += o
        bad = failed_checks(case, out.decode("utf-8"))
        assert not bad, (case["src"], bad[:3], out[:200])
        n += len(case["checks"])
    assert n >= 900


def test_goldens_through_python_token_writer(gpu):
    """token_writer_test.go:34-109 through the Python mirror (closures replayed)."""
    import datok_amd
    cases = [c for c in CASES if c["src"].startswith("token_writer_test.go")]
    assert len(cases) == 8
    for case in cases:
        w = io.BytesIO()
        for c in case["calls"]:
            tw = datok_amd.new_token_writer(w, c["flags"])
            assert gpu(c["model"]).transduce_token_writer(io.BytesIO(c["input"].encode()), tw)
        assert not failed_checks(case, w.getvalue().decode()), (case["src"], w.getvalue())


def test_type_and_loader(gpu, tmp_path):
    """datok_test.go:252-261; loader failure returns None like the reference's nil."""
    import datok_amd
    assert gpu("simpletok.datok").type() == "DATOK"
    assert gpu("simpletok.matok").type() == "MATOK"
    assert datok_amd.load_tokenizer_file(str(tmp_path / "missing.matok")) is None
    (tmp_path / "junk.matok").write_bytes(b"not gzip at all")
    assert datok_amd.load_tokenizer_file(str(tmp_path / "junk.matok")) is None
    info = gpu("tokenizer_de.matok").info
    assert (info["epsilon"], info["unknown"], info["identity"]) == (1, 2, 3)
    assert info["state_count"] == 18400 and info["sigma_count"] == 171 and info["entry_bytes"] in (2, 4)


def test_matok_datok_equivalence(gpu):
    """matrix_test.go:1248-1275 on the 750-byte benchmark string."""
    s = golden_strings()["s"].encode()
    a, _ = gpu("tokenizer_de.datok").transduce_bytes(s)
    b, _ = gpu("tokenizer_de.matok").transduce_bytes(s)
    assert a == b and a.count(b"\n") > 130


def test_foma_nets_load_like_their_matok(gpu, oracle_models):
    """LoadFomaFile(...).ToMatrix() on the device (fomafile.go:56-450, matrix.go:30-99): the net and the
    .matok the reference built from it give the same tokenizer."""
    from datok_amd import corpus
    for stem in ("tokenizer_de", "clitic_test", "simpletok"):
        a, b = gpu(stem + ".fst"), gpu(stem + ".matok")
        assert a.type() == "MATOK"
        ia, ib = a.info, b.info
        assert {k: ia[k] for k in ia if k != "device_bytes"} == {k: ib[k] for k in ib if k != "device_bytes"}
    text, off = corpus.german_docs(256, 4096, seed=21)
    res, tot = run_batch(gpu("tokenizer_de.fst"), text, off)
    assert tot["n_flagged"] == 0
    assert_batch_equals_oracle(oracle_models("tokenizer_de.matok"), res, text, off)


# ------------------------------------------------------------ batch vs oracle
def test_config1_simpletok_1k(gpu, oracle_models):
    from datok_amd import corpus
    text, off = corpus.simple_ascii(seed=1, n=1024)
    res, tot = run_batch(gpu("simpletok.matok"), text, off)
    assert assert_batch_equals_oracle(oracle_models("simpletok.matok"), res, text, off) == 1
    res, _ = run_batch(gpu("simpletok.datok"), text, off)
    assert_batch_equals_oracle(oracle_models("simpletok.datok"), res, text, off)
    out, st = gpu("simpletok.matok").transduce_bytes(text.tobytes())
    assert (out, st) == oracle_models("simpletok.matok").transduce(text.tobytes())


@pytest.mark.parametrize("chunk", [None, 0])
def test_config2_german_4096x4096_full(gpu, oracle_models, chunk):
    """The bench workload itself, every document, every offset; default (speculative chunk
    lanes) and one-lane-per-document walk."""
    from datok_amd import corpus
    text, off = corpus.german_docs(4096, 4096, seed=2)
    res, tot = run_batch(gpu("tokenizer_de.matok"), text, off, chunk=chunk)
    assert tot["n_flagged"] == 0 and tot["n_texts"] == 4096
    assert (tot["n_lanes"] > 4096) == (chunk is None)
    n = assert_batch_equals_oracle(oracle_models("tokenizer_de.matok"), res, text, off)
    assert n == 4096
    counts = oracle_models("tokenizer_de.matok").count_batch(text, off, 4)
    assert tot["n_tokens"] == int(counts[:, 0].sum())


def test_config4_double_array_full(gpu, oracle_models):
    """BASELINE.json configs[3] at its full size: tokenizer_de.datok on the 4096 x 4 KiB batch of config 2 --
    every field equals the matrix run's (no U+0004 in the corpus, matrix_test.go:1248-1275) and every
    document equals the oracle's double-array walk."""
    from datok_amd import corpus
    text, off = corpus.german_docs(4096, 4096, seed=2)
    rd, td = run_batch(gpu("tokenizer_de.datok"), text, off)
    rm, _ = run_batch(gpu("tokenizer_de.matok"), text, off)
    assert td["n_flagged"] == 0 and td["n_docs"] == 4096
    for f in ("tok_off", "tok_rstart", "tok_rend", "tok_bstart", "tok_bend", "sent_off", "sent",
              "text_tok_end", "text_sent_end", "status"):
        assert np.array_equal(getattr(rd, f), getattr(rm, f)), f
    assert assert_batch_equals_oracle(oracle_models("tokenizer_de.datok"), rd, text, off) == 4096


def _check_size_independent(res, off, counts):
    """Properties that hold at any size: per-document token / text counts equal the oracle's counting pass
    (orc_count_batch), tokens are non-empty, sorted and non-overlapping inside their document, rune offsets
    never exceed byte offsets."""
    ntok = np.diff(res.tok_off.astype(np.int64))
    assert np.array_equal(ntok, counts[:, 0].astype(np.int64))
    assert np.array_equal(np.diff(res.text_off.astype(np.int64)), counts[:, 2].astype(np.int64))
    assert np.all(res.tok_bend > res.tok_bstart) and np.all(res.tok_rend > res.tok_rstart)
    same_doc = np.ones(len(res.tok_bstart) - 1, dtype=bool)
    first = res.tok_off[1:-1].astype(np.int64)
    same_doc[first[(first > 0) & (first < len(res.tok_bstart))] - 1] = False
    assert np.all(res.tok_bstart[1:][same_doc] >= res.tok_bend[:-1][same_doc])
    doc_len = np.diff(off.astype(np.int64))
    has = ntok > 0
    last = (res.tok_off[1:].astype(np.int64) - 1)[has]
    assert np.all(res.tok_bend[last] <= doc_len[has])
    assert np.all(res.tok_rend <= res.tok_bend.astype(np.int64))


def test_config3_english_zipf(gpu, oracle_models):
    from datok_amd import corpus
    text, off = corpus.english_zipf_docs(4096, seed=3)
    res, tot = run_batch(gpu("tokenizer_en.matok"), text, off)
    assert tot["n_flagged"] == 0
    assert_batch_equals_oracle(oracle_models("tokenizer_en.matok"), res, text, off)


def test_config3_english_zipf_full(gpu, oracle_models):
    """BASELINE.json configs[2] at its full size: 65 536 documents, Zipf lengths 64 B .. 64 KiB (about 294 MB): the
    multi-block scan, the separate k_spec_fix launch (more than 8192 documents) and the load-balance stress the
    config exists for.  Every document's counts against the oracle's counting pass, and every offset of 2 112
    documents -- 64 of the longest (64 KiB) and a seeded random sample -- against the oracle."""
    from datok_amd import corpus
    text, off = corpus.english_zipf_docs(65536, seed=3)
    assert len(off) - 1 == 65536 and len(text) > 250_000_000
    om = oracle_models("tokenizer_en.matok")
    res, tot = run_batch(gpu("tokenizer_en.matok"), text, off)
    assert tot["n_flagged"] == 0 and tot["n_docs"] == 65536 and tot["n_texts"] == 65536
    counts = om.count_batch(text, off, os.cpu_count() or 1)
    assert tot["n_tokens"] == int(counts[:, 0].sum())
    _check_size_independent(res, off, counts)
    lens = np.diff(off.astype(np.int64))
    longest = np.flatnonzero(lens == lens.max())
    rng = np.random.default_rng(3)
    sample = sorted(set(longest[:64].tolist()) | set(rng.choice(65536, size=2048, replace=False).tolist()))
    assert len(sample) >= 2048 and lens.max() == 65536
    assert assert_batch_equals_oracle(om, res, text, off, docs=sample) == len(sample)


def test_config5_one_shard_of_the_10gib_corpus(gpu, oracle_models):
    """BASELINE.json configs[4] as one GPU sees it: 10 GiB = 2 621 440 documents x 4 KiB sharded 8 ways =
    327 680 documents (1.25 GiB) per GPU; rank 0's shard (seed 5000 + rank, what bench.py --gpus 8 walks).
    Per-document counts of all 327 680 documents against the oracle's counting pass, the size-independent
    properties, and every offset of a sample of documents against the oracle."""
    import datok_amd
    from datok_amd import corpus
    n_docs = 327680
    text, off = corpus.german_docs_sharded(n_docs, 4096, seed=5000)
    assert len(text) == n_docs * 4096
    om, tok = oracle_models("tokenizer_de.matok"), gpu("tokenizer_de.matok")
    with datok_amd.Batch(len(text), n_docs) as b:
        b.set_input(text, off)
        b.run(tok, datok_amd.host.OFFSETS_ONLY)
        tot = b.totals()
        assert tot["n_flagged"] == 0 and tot["n_docs"] == n_docs and tot["n_texts"] == n_docs
        res = b.result()
    counts = om.count_batch(text, off, os.cpu_count() or 1)
    assert tot["n_tokens"] == int(counts[:, 0].sum()) and tot["n_tokens"] > 150_000_000
    _check_size_independent(res, off, counts)
    rng = np.random.default_rng(5)
    sample = sorted(set(range(64)) | set(range(n_docs - 64, n_docs)) | set(rng.choice(n_docs, size=1024, replace=False).tolist()))
    assert assert_batch_equals_oracle(om, res, text, off, docs=sample) == len(sample)


def _edge_docs():
    rng = np.random.default_rng(7)
    docs = [b"", b" ", b"\n", b".", b"a", b"\x04", b"\x04\x04", b"A.\x04", b"A.\x04\x04", b"\x04\nA",
            "This.\n\x04And.\n\x04\n".encode(), "\nThis.\n\x04\nAnd.\n\x04\n".encode(),
            "Tree\n\x04\n".encode(), "Erste.\n\n\n\n\x04\nNächst.\x04".encode(),
            "word\x04 more words. And\x04more".encode(), "a\x04b\x04c".encode(),
            "x\x04y \x04 z\x04".encode(), "ibauamt\x04dead. \x04".encode(),
            "„Zitat“ – so … ‚x‘ »y« ∞ ≠ ≤ 日本語 テスト".encode(), "😀 emoji 👍🏽 ok".encode(),
            b"\xff\xfe invalid \x80\x80 bytes \xc3", b"\xe2\x82", b"\xf0\x9f\x98", b"ab\xc0\xafcd",
            b"x" * 1100, b" " * 1100 + b"x", b"a " * 700, ("ä" * 1030).encode(), b"." * 300,
            "Der Vorsitzende der Abk. hat gewählt. Gefunden auf wikipedia.org.".encode(),
            # token length field of the closing event byte: 30 / 31 / 32 bytes and longer (start marks),
            # also right behind a sentence end without a blank and as the first / last token
            b"z" * 30, b"z" * 31, b"z" * 32, b"A." + b"x" * 40, b"Satz. " + b"q" * 31 + b" und " + b"q" * 30,
            ("ä" * 15).encode(), ("ä" * 16).encode(), b"Hallo! " + b"w" * 64 + b" Ende.",
            b"http://www.example.org/a/very/long/path/with/many/segments/index.html?x=1&y=2 ok",
            b"k" * 31 + b"\x04" + b"m" * 33 + b"\x04\n" + b"n" * 31]
    alphabet = list(" \n\t.,;:!?'\"()-@/&%abcdefgABCDE0123äöüß„“»«…€") + ["\x04"]
    for _ in range(300):
        k = int(rng.integers(0, 200))
        docs.append("".join(alphabet[int(i)] for i in rng.integers(0, len(alphabet), size=k)).encode())
    for _ in range(100):  # raw bytes, mostly invalid UTF-8
        docs.append(bytes(rng.integers(0, 256, size=int(rng.integers(1, 120)), dtype=np.uint8)))
    return docs


@pytest.mark.parametrize("model", ["tokenizer_de.matok", "tokenizer_en.matok", "clitic_test.matok",
                                   "simpletok.matok", "simpletok.datok", "tokenizer_de.datok",
                                   # Foma nets converted by the loader; the first two have no identity symbol
                                   "bauamt.fst", "wahlamt.fst", "ignorable_mcs.fst"])
@pytest.mark.parametrize("flags,chunk,warm", [(0, 0, 64), (NEWLINE_AFTER_EOT, 0, 64), (0, 16, 64),
                                              (NEWLINE_AFTER_EOT, 32, 8), (0, 64, 0)])
def test_edge_documents(gpu, oracle_models, model, flags, chunk, warm):
    """Empty / ragged / EOT / invalid UTF-8 / window-overflow documents in one batch, walked
    one lane per document and as speculative chunks (small warm-ups force repair rounds)."""
    from datok_amd import corpus, ST_IRREGULAR
    docs = _edge_docs()
    text, off = corpus.concat_docs(docs)
    res, tot = run_batch(gpu(model), text, off, flags, chunk=chunk, warm=warm)
    # every document, both encodings: calls that are not in position order (the double array revisiting an
    # EOT, datok.go:1019-1030) are handled by the exact pass and never reported as a status
    assert not any(int(s) & ST_IRREGULAR for s in res.status)
    n = assert_batch_equals_oracle(oracle_models(model), res, text, off, flags)
    assert n > 200


@pytest.mark.parametrize("model", ["tokenizer_de.matok", "tokenizer_de.datok", "tokenizer_en.matok"])
@pytest.mark.parametrize("chunk,warm", [(64, 64), (256, 64), (1024, 32), (128, 0), (48, 4), (4096, 64)])
def test_speculative_chunks_are_exact(gpu, oracle_models, model, chunk, warm):
    """Chunk lanes + check/repair must reproduce the sequential walk bit for bit, also when the
    warm-up is too short to re-synchronise (warm 0 / 4: about a third of the lanes mispredict)."""
    from datok_amd import corpus
    if model.endswith("en.matok"):
        text, off = corpus.english_zipf_docs(512, seed=4, max_bytes=16384)
    else:
        text, off = corpus.german_docs(384, 4096, seed=9)
    res, tot = run_batch(gpu(model), text, off, chunk=chunk, warm=warm)
    assert tot["n_flagged"] == 0 and tot["chunk_bytes"] == chunk
    if warm == 0:
        assert tot["repair_rounds"] > 0
    assert_batch_equals_oracle(oracle_models(model), res, text, off)


def test_long_single_stream_is_chunked(gpu, oracle_models):
    """One 2 MiB document (the reference's one-reader use): thousands of lanes, same offsets."""
    from datok_amd import corpus
    text, _ = corpus.german_docs(512, 4096, seed=21)
    off = np.array([0, len(text)], dtype=np.uint64)
    res, tot = run_batch(gpu("tokenizer_de.matok"), text, off)
    assert tot["n_lanes"] > 1000 and tot["n_flagged"] == 0
    assert_batch_equals_oracle(oracle_models("tokenizer_de.matok"), res, text, off)


def test_rendered_output_all_flag_combinations(gpu, oracle_models):
    texts = ["This.\n\x04And.\n\x04\n", "\nThis.\n\x04\nAnd.\n\x04\n", "Der alte Mann. Er ging!", "", " ",
             "„Hallo“, sagte er. »Nein!«"]
    for model in ("tokenizer_de.matok", "tokenizer_de.datok"):
        for flags in range(32):
            for t in texts:
                exp, est = oracle_models(model).transduce(t.encode(), flags)
                if est:
                    continue
                got, st = gpu(model).transduce_bytes(t.encode(), flags)
                assert (got, st) == (exp, 0), (model, flags, t)
                # the closure-replay path (custom TokenWriter) prints the same bytes
                assert gpu(model).transduce_bytes(t.encode(), flags, replay=True) == (exp, 0)


# ------------------------------------------------- NewTokenWriter on the device
ALL_MODELS = ["tokenizer_de.matok", "tokenizer_en.matok", "clitic_test.matok", "simpletok.matok",
              "simpletok.datok", "tokenizer_de.datok", "bauamt.fst", "ignorable_mcs.fst"]


@pytest.mark.parametrize("model", ALL_MODELS)
def test_device_rendering_edge_documents(gpu, oracle_models, model):
    """dtk_batch_render_*: bytes[doc_off[d]:doc_off[d+1]] == what NewTokenWriter(w, bits) prints for
    document d (token_writer.go:36-175), for all 16 writer modes x NEWLINE_AFTER_EOT, on the edge
    documents (EOT texts, invalid UTF-8, empty documents, window overflows excluded by status)."""
    import datok_amd
    from datok_amd import corpus
    docs = _edge_docs()
    text, off = corpus.concat_docs(docs)
    om = oracle_models(model)
    checked = 0
    with datok_amd.Batch(len(text), len(docs)) as b:
        b.set_input(text, off)
        for nl in (0, NEWLINE_AFTER_EOT):
            b.run(gpu(model), nl)
            status = b.result().status
            for bits in range(16):
                data, o = b.render(bits | nl)
                assert o[0] == 0 and o[-1] == len(data) and np.all(np.diff(o.astype(np.int64)) >= 0)
                for d, doc in enumerate(docs):
                    exp, est = om.transduce(doc, bits | nl)
                    if est or (int(status[d]) & ~datok_amd.ST_EMPTY_TEXT):
                        continue   # the reference panics / out of contract (flagged)
                    got = data[int(o[d]):int(o[d + 1])]
                    assert got == exp, (model, bits | nl, d, doc[:80], got[:120], exp[:120])
                    checked += 1
    assert checked > 16 * 2 * 200


@pytest.mark.parametrize("model", ["tokenizer_de.matok", "tokenizer_de.datok"])
def test_device_rendering_config2(gpu, oracle_models, model):
    """The bench batch rendered on the device: SIMPLE (= Transduce, matrix.go:340-342) and everything
    at once, every document against the oracle; rendering twice with other bits needs no new run."""
    import datok_amd
    from datok_amd import corpus
    text, off = corpus.german_docs(4096, 4096, seed=2)
    raw = text.tobytes()
    om = oracle_models(model)
    with datok_amd.Batch(len(text), len(off) - 1) as b:
        b.set_input(text, off)
        b.run(gpu(model), 0)
        for bits in (SIMPLE, 15, TOKEN_POS):
            data, o = b.render(bits)
            for d in range(len(off) - 1):
                exp, est = om.transduce(raw[int(off[d]):int(off[d + 1])], bits)
                assert est == 0 and data[int(o[d]):int(o[d + 1])] == exp, (bits, d)
        with pytest.raises(datok_amd.DatokGpuError):
            b.render(SIMPLE | NEWLINE_AFTER_EOT)   # positions were computed without that rule
        # offsets only (what bench.py runs): same arrays, no renderer bookkeeping
        full = b.result()
        b.run(gpu(model), datok_amd.host.OFFSETS_ONLY)
        lean = b.result()
        for f in ("tok_off", "tok_rstart", "tok_rend", "tok_bstart", "tok_bend", "sent", "text_tok_end", "status"):
            assert np.array_equal(getattr(full, f), getattr(lean, f)), f
        with pytest.raises(datok_amd.DatokGpuError):
            b.render(SIMPLE)


def test_device_rendering_single_long_stream(gpu, oracle_models):
    """One 2 MiB stream with in-document EOT texts (a DeReKo-style stream, Readme.md:54)."""
    import datok_amd
    from datok_amd import corpus
    text, off = corpus.german_docs(512, 4096, seed=31)
    raw = bytearray(text.tobytes())
    for d in range(1, 512):        # every 4 KiB: "\x04\n" ends a text
        raw[d * 4096 - 2:d * 4096] = b"\x04\n"
    raw = bytes(raw)
    one = np.frombuffer(raw, dtype=np.uint8)
    tok, om = gpu("tokenizer_de.matok"), oracle_models("tokenizer_de.matok")
    with datok_amd.Batch(len(one), 1) as b:
        b.set_input(one, np.array([0, len(one)], dtype=np.uint64))
        b.run(tok, NEWLINE_AFTER_EOT)
        for bits in (SIMPLE, 15, SENTENCE_POS):
            data, o = b.render(bits | NEWLINE_AFTER_EOT)
            exp, est = om.transduce(raw, bits | NEWLINE_AFTER_EOT)
            assert est == 0 and data == exp


def test_size_independent_properties_large(gpu):
    """64 MiB: offsets sorted, non-overlapping, inside the document; tokens never empty;
    re-running gives identical results (idempotence); a permuted batch permutes rows."""
    from datok_amd import corpus
    text, off = corpus.german_docs(16384, 4096, seed=5)
    tok = gpu("tokenizer_de.matok")
    res, tot = run_batch(tok, text, off)
    assert tot["n_flagged"] == 0
    assert np.all(res.tok_bend > res.tok_bstart)
    assert np.all(res.tok_rend > res.tok_rstart)
    same_doc = np.ones(len(res.tok_bstart) - 1, dtype=bool)
    same_doc[(res.tok_off[1:-1] - 1).astype(np.int64)] = False
    assert np.all(res.tok_bstart[1:][same_doc] >= res.tok_bend[:-1][same_doc])
    last = (res.tok_off[1:] - 1).astype(np.int64)
    assert np.all(res.tok_bend[last] <= 4096)
    res2, _ = run_batch(tok, text, off)
    assert np.array_equal(res.tok_rstart, res2.tok_rstart) and np.array_equal(res.sent, res2.sent)
    # reverse the document order
    t2 = text.reshape(16384, 4096)[::-1].copy().ravel()
    res3, _ = run_batch(tok, t2, off)
    ntok = np.diff(res.tok_off.astype(np.int64))
    assert np.array_equal(np.diff(res3.tok_off.astype(np.int64)), ntok[::-1])
    d = 123
    a3 = res3.doc(16384 - 1 - d)
    a1 = res.doc(d)
    assert np.array_equal(a1["tok_rstart"], a3["tok_rstart"]) and np.array_equal(a1["sent"], a3["sent"])


# ------------------------------------------------------------ CLI and C++ mirror
def test_cli_tokenize(oracle_models, tmp_path):
    """`datok tokenize -t tok [flags] file|-` (cmd/datok.go:74-133): flag -> Bits mapping and output."""
    import subprocess
    import datok_amd
    exe = os.path.join(os.path.dirname(datok_amd.__file__), "datok")
    text = "Der alte Mann. Er ging! „Wirklich?“\n\x04\nZweiter Text, z.B. hier.\n\x04\n".encode()
    inp = tmp_path / "in.txt"
    inp.write_bytes(text)
    model = os.path.join(MODELS, "tokenizer_de.matok")
    om = oracle_models("tokenizer_de.matok")
    for args, bits in [([], SIMPLE), (["-p"], SIMPLE | TOKEN_POS), (["--no-tokens", "--sentence-positions"], SENTENCES | SENTENCE_POS),
                       (["--no-sentences", "--token-positions", "--newline-after-eot"], TOKENS | TOKEN_POS | NEWLINE_AFTER_EOT),
                       (["--no-tokens", "--no-sentences"], 0)]:
        exp, est = om.transduce(text, bits)
        assert est == 0
        r = subprocess.run([exe, "tokenize", "-t", model] + args + [str(inp)], capture_output=True)
        assert r.returncode == 0 and r.stdout == exp, (args, r.stdout, r.stderr)
    r = subprocess.run([exe, "tokenize", "--tokenizer=" + model, "-"], input=text, capture_output=True)
    assert r.returncode == 0 and r.stdout == om.transduce(text, SIMPLE)[0]
    r = subprocess.run([exe, "tokenize", "-t", str(tmp_path / "none.matok"), str(inp)], capture_output=True)
    assert r.returncode == 1 and b"Unable to load file" in r.stderr


def test_cpp_mirror_end_to_end(oracle_models, tmp_path):
    """include/datok.hpp: LoadTokenizerFile / Transduce (device rendering) / TransduceTokenWriter with a
    custom writer (closure replay) -- the Go surface of fomafile.go:29-33 in C++."""
    import subprocess
    import datok_amd
    lib = datok_amd.build()
    src = tmp_path / "e2e.cpp"
    src.write_text(r'''
#include <fstream>
#include <iostream>
#include <sstream>
#include "datok.hpp"
int main(int argc, char **argv) {
  auto tok = datok::LoadTokenizerFile(argv[1]);
  if (!tok) return 3;
  std::cout << tok->Type() << "\n";
  { std::ifstream in(argv[2], std::ios::binary); if (!tok->Transduce(in, std::cout)) return 4; }
  std::cout << "--\n";
  { std::ifstream in(argv[2], std::ios::binary);
    auto tw = datok::NewTokenWriter(std::cout, datok::TOKENS | datok::TOKEN_POS | datok::SENTENCE_POS);
    if (!tok->TransduceTokenWriter(in, *tw)) return 5; }
  std::cout << "--\n";
  { std::ifstream in(argv[2], std::ios::binary);
    datok::TokenWriter tw;   // a custom writer: counts calls
    int nt = 0, ns = 0, ne = 0;
    tw.Token = [&](int, const std::vector<datok::rune> &) { nt++; };
    tw.SentenceEnd = [&](int) { ns++; };
    tw.TextEnd = [&](int) { ne++; };
    tw.Flush = [] { return 0; };
    if (!tok->TransduceTokenWriter(in, tw)) return 6;
    std::cout << nt << " " << ns << " " << ne << "\n"; }
  return datok::LoadTokenizerFile("/nonexistent.matok") ? 7 : 0;
}
''')
    exe = tmp_path / "e2e"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           lib, "-Wl,-rpath," + os.path.dirname(lib)])
    text = "Der alte Mann. Er ging!\n\x04\nNoch ein Text.".encode()
    inp = tmp_path / "in.txt"
    inp.write_bytes(text)
    om = oracle_models("tokenizer_de.matok")
    r = subprocess.run([str(exe), os.path.join(MODELS, "tokenizer_de.matok"), str(inp)], capture_output=True)
    assert r.returncode == 0, r.stderr
    simple = om.transduce(text, SIMPLE)[0]
    full = om.transduce(text, TOKENS | TOKEN_POS | SENTENCE_POS)[0]
    head, rest = r.stdout.split(b"\n", 1)
    assert head == b"MATOK"
    a, b, c = rest.split(b"--\n")
    assert a == simple and b == full
    nt, ns, ne = (int(x) for x in c.split())
    assert nt == simple.count(b"\n") - ns - ne and ne == 2 and ns >= 3


def test_many_small_documents(gpu, oracle_models):
    """20 000 documents of 10..200 bytes (the row offsets then come from the multi-block scan; lanes
    are single chunks): every document against the oracle."""
    from datok_amd import corpus
    text, off = corpus.german_docs(512, 4096, seed=77)
    raw = text.tobytes()
    rng = np.random.default_rng(5)
    cuts = [0]
    while cuts[-1] < len(raw) and len(cuts) <= 20000:
        nxt = min(len(raw), cuts[-1] + int(rng.integers(10, 200)))
        while nxt < len(raw) and (raw[nxt] & 0xC0) == 0x80:   # keep the pieces valid UTF-8
            nxt += 1
        cuts.append(nxt)
    cuts = np.array(cuts, dtype=np.uint64)
    piece = np.frombuffer(raw[:int(cuts[-1])], dtype=np.uint8)
    assert len(cuts) - 1 > 8192
    res, tot = run_batch(gpu("tokenizer_de.matok"), piece, cuts)
    assert tot["n_docs"] == len(cuts) - 1
    assert assert_batch_equals_oracle(oracle_models("tokenizer_de.matok"), res, piece, cuts) > 8192


def test_event_buffer_rounding(gpu, oracle_models):
    """A batch created for exactly its input: the event arrays are rounded up to 256 B per run, which must
    fit the allocation for every size (found by scripts/soak.py: total + 4 n_docs + 4 = 1..3 mod 256)."""
    from datok_amd import corpus
    for n in (249, 250, 251, 252, 505, 1017):
        text, off = corpus.concat_docs([b"ab " * (n // 3) + b"c" * (n % 3)])
        assert len(text) == n
        res, _ = run_batch(gpu("tokenizer_de.matok"), text, off)
        assert_batch_equals_oracle(oracle_models("tokenizer_de.matok"), res, text, off)


@pytest.mark.parametrize("model", ["tokenizer_de.matok", "clitic_test.matok"])
@pytest.mark.parametrize("flags,chunk", [(0, 48), (NEWLINE_AFTER_EOT, 128), (NEWLINE_AFTER_EOT, None)])
def test_long_documents_with_eot_texts_in_segments(gpu, oracle_models, model, flags, chunk):
    """Documents of many chunk lanes are compacted in segments of 64 lanes whose carries come from the
    lanes' totals (k_seg_sum / k_seg_scan): long documents stuffed with EOT texts (also several in a row,
    at the start and at the end), sentence ends and invalid bytes, offsets and rendered text."""
    import datok_amd
    from datok_amd import corpus
    rng = np.random.default_rng(99)
    docs = []
    edge = _edge_docs()
    for k in range(6):
        parts = [edge[int(i)] for i in rng.integers(0, len(edge), size=int(rng.integers(200, 500)))]
        parts = [p for p in parts if len(p) < 400]
        sep = [b" ", b"\n", b"\x04", b"\x04\n", b". ", b"\x04\x04"]
        raw = b"".join(p + sep[int(rng.integers(0, len(sep)))] for p in parts)
        if k == 0:
            raw = b"\x04\x04" + raw
        if k == 1:
            raw = raw + b"\x04"
        docs.append(raw)
    text, off = corpus.concat_docs(docs)
    om = oracle_models(model)
    with datok_amd.Batch(len(text), len(docs)) as b:
        if chunk is not None:
            b.set_chunking(chunk, 48)
        b.set_input(text, off)
        b.run(gpu(model), flags)
        res, tot = b.result(), b.totals()
        assert tot["n_lanes"] > 64 * len(docs)
        ok = [d for d in range(len(docs)) if not (int(res.status[d]) & ~datok_amd.ST_EMPTY_TEXT)]
        assert len(ok) >= 1
        assert_batch_equals_oracle(om, res, text, off, flags, docs=range(len(docs)))
        for bits in (SIMPLE, 15):
            data, o = b.render(bits | flags)
            for d in ok:
                exp, est = om.transduce(docs[d], bits | flags)
                if est == 0:
                    assert data[int(o[d]):int(o[d + 1])] == exp, (model, flags, chunk, bits, d)


def test_double_array_long_documents(gpu, oracle_models):
    """The double array's long documents: in segments when they hold no EOT, sequentially otherwise
    (datok.go:1019-1030 keeps the window over an EOT, so the segment carries are not closed-form)."""
    from datok_amd import corpus
    text, off = corpus.german_docs(6, 60000, seed=41)
    raw = bytearray(text.tobytes())
    for p in (70000, 70001, 130000, 200000):     # documents 1, 2 and 3 get EOTs; 0, 4, 5 stay clean
        raw[p] = 4
    text = np.frombuffer(bytes(raw), dtype=np.uint8)
    res, tot = run_batch(gpu("tokenizer_de.datok"), text, off, NEWLINE_AFTER_EOT, chunk=128)
    # (document 1 holds two EOTs in a row: a text without a token, DTK_ST_EMPTY_TEXT as for the oracle)
    assert tot["n_lanes"] > 64 * 6 and tot["n_flagged"] == 1 and int(res.status[1]) == 2
    assert assert_batch_equals_oracle(oracle_models("tokenizer_de.datok"), res, text, off, NEWLINE_AFTER_EOT) == 5
    a, _ = run_batch(gpu("tokenizer_de.matok"), text, off, NEWLINE_AFTER_EOT, chunk=128)
    for d in (0, 4, 5):   # without EOT both encodings give the same offsets
        assert np.array_equal(a.doc(d)["tok_rstart"], res.doc(d)["tok_rstart"])


_VARIANT_SCRIPT = r"""
import os, sys
import numpy as np
ROOT = sys.argv[1]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datok_amd
from datok_amd import corpus
from oracle import oracle as O
from parity import assert_batch_equals_oracle, oracle_doc
M = os.path.join(ROOT, "tests", "golden", "models")
for name, (text, off) in (("tokenizer_de.matok", corpus.german_docs(96, 4096, seed=11)),
                          ("tokenizer_en.matok", corpus.english_zipf_docs(64, seed=12, max_bytes=8192)),
                          ("tokenizer_de.datok", corpus.german_docs(32, 2048, seed=13))):
    tok = datok_amd.load_tokenizer_file(os.path.join(M, name)); om = O.Model(os.path.join(M, name))
    for chunk, warm in ((None, 48), (64, 8), (256, 48)):
        with datok_amd.Batch(len(text), len(off) - 1) as b:
            if chunk is not None:
                b.set_chunking(chunk, warm)
            b.set_input(text, off); b.run(tok, 0)
            res = b.result()
            assert not res.status.any()
            assert_batch_equals_oracle(om, res, text, off)
print("VARIANT OK")
"""


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"DATOK_LDS_BITS": "0"}, {"DATOK_SPLIT_START": "1"},
                                 {"DATOK_SPLIT_START": "1", "DATOK_LDS_BITS": "0"}, {"DATOK_CLEAR_KERNEL": "1"},
                                 {"DATOK_COMPACT_FULL": "1", "DATOK_DEV_ROUNDS": "2"}, {"DATOK_FILE_COLUMNS": "1"},
                                 {"DATOK_ROUND_LIMIT": "1"}],
                         ids=["no-lds-bitmaps", "split-start", "split-start+no-lds-bitmaps", "clear-kernel",
                              "both-compactions+device-rounds", "file-column-order", "one-lane-fallback"])
def test_kernel_variants_forced_by_environment(env, tmp_path):
    """The library runs the first pass as one launch that reports through the waves' LDS bitmaps, clears its
    accumulators in k_symbolize and launches what a batch's last run needed; the other paths (event bits straight to
    memory, start records and walk as two launches -- which is also what repair rounds use --, a clear kernel, both
    compaction kernels and device-side repair rounds with every run; the walk with one lane per document that takes
    over when the repair rounds run out, here after the first) must give the same offsets.  The switches are read
    once per process."""
    import subprocess
    script = tmp_path / "variant.py"
    script.write_text(_VARIANT_SCRIPT)
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, env=e, timeout=600)
    assert r.returncode == 0 and b"VARIANT OK" in r.stdout, r.stderr.decode()[-2000:]


@pytest.mark.gpu
def test_double_array_dense_layout_and_pairs_path(gpu):
    """A double-array tokenizer is walked through a dense (fused matrix) layout of its own transitions built at load
    (dtk_host.cpp build_datok) -- every double-array test of this suite runs that way.  DATOK_NO_DENSE=1 keeps the
    {base, check} pairs of the file on the device (DaTrans, two dependent loads per step): the same tests in a
    process of their own, so that path stays exact too."""
    import subprocess
    if os.environ.get("DATOK_NO_FUSED") or os.environ.get("DATOK_FORCE_WIDE") or os.environ.get("DATOK_NO_DENSE"):
        pytest.skip("the dense layout needs the fused cells (those switches select other table encodings)")
    for name in ("tokenizer_de.datok", "simpletok.datok"):
        info = gpu(name).info
        assert info["dense_states"] > 0 and info["entry_bytes"] == 4, info
    assert gpu("tokenizer_de.matok").info["dense_states"] == 0
    e = dict(os.environ); e["DATOK_NO_DENSE"] = "1"
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", "-k",
                        "(datok or double_array or out_of_position or closure_int or edge_documents or config4 or "
                        "speculative_chunks) and not pairs_path", os.path.join(ROOT, "tests")],
                       capture_output=True, env=e, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, (r.stdout.decode()[-1500:], r.stderr.decode()[-500:])
    assert b" passed" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["tokenizer_de.matok", "tokenizer_de.datok"])
@pytest.mark.parametrize("chunk", [64, 128, 256])
def test_tokens_longer_than_warmup_and_chunk(gpu, oracle_models, model, chunk):
    """Blank-free tokens of 60..700 bytes (URLs, runs of letters): longer than the warm-up, and the longest longer
    than several chunks, so that lanes exist whose whole chunk lies inside one token (they own nothing).  The start
    of a warm-up inside such a token moves back to the previous blank; with that switched off the same inputs go
    through repair rounds.  Offsets equal the oracle's either way."""
    import datok_amd
    from datok_amd import corpus
    rng = np.random.default_rng(21)
    text, off = corpus.german_docs(96, 4096, seed=21)
    raw = bytearray(text.tobytes())
    alpha = list(b"abcdefghijklmnopqrstuvwxyz0123456789/_-%")
    for d in range(96):
        p = d * 4096 + int(rng.integers(200, 1500))
        for _ in range(3):
            q = raw.find(b" ", p)
            L = int(rng.choice([60, 90, 130, 200, 330, 700]))
            if q < 0 or q + L + 2 >= (d + 1) * 4096 - 8:
                break
            body = bytes(rng.choice(alpha, size=L)) if rng.integers(0, 2) else b"x" * L
            tok_bytes = (b"https://www.example.org/" + body)[:L]
            raw[q + 1:q + 1 + L] = tok_bytes
            raw[q + 1 + L] = 0x20
            p = q + L + int(rng.integers(100, 600))
    text = np.frombuffer(bytes(raw), dtype=np.uint8).copy()
    tok, om = gpu(model), oracle_models(model)
    rounds = {}
    for extend in (240, 0):
        with datok_amd.Batch(len(text), len(off) - 1) as b:
            b.set_chunking(chunk, 48, extend=extend)
            b.set_input(text, off)
            b.run(tok, 0)
            res, tot = b.result(), b.totals()
            assert tot["n_flagged"] == 0
            assert_batch_equals_oracle(om, res, text, off)
            rounds[extend] = tot["repair_rounds"]
    assert rounds[0] > 0  # the fixed distance does mispredict here
    assert rounds[240] <= rounds[0]


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["tokenizer_de.matok", "tokenizer_en.matok", "tokenizer_de.datok"])
def test_empty_second_token_of_a_dotted_line(gpu, oracle_models, model):
    """A document that is nothing but full stops and a newline makes the reference emit a second Token call with
    an empty surface (offset == len(buffer); rune offsets (n + 1, n + 1)).  Its byte range is the empty range at the
    end of the document, for short and for long (31+ bytes: start mark + saturated length field) first tokens."""
    from datok_amd import corpus
    docs = [b"." * n + b"\n" for n in (1, 2, 3, 29, 30, 31, 40, 400)] + [b"Hi " + b"." * 40 + b"\nmore.", b"." * 40 + b" a"]
    text, off = corpus.concat_docs(docs)
    om = oracle_models(model)
    for chunk in (0, 64, None):
        res, tot = run_batch(gpu(model), text, off, chunk=chunk)
        assert_batch_equals_oracle(om, res, text, off)
    exp = om.transduce_doc(docs[6], 0)  # 40 full stops
    assert list(zip(exp.tok_bstart.tolist(), exp.tok_bend.tolist())) == [(0, 40), (41, 41)]
    assert list(zip(exp.tok_rstart.tolist(), exp.tok_rend.tolist())) == [(0, 40), (41, 41)]


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["tokenizer_de.matok", "tokenizer_en.matok", "tokenizer_de.datok"])
def test_token_offset_behind_its_buffer_is_reported(gpu, oracle_models, model):
    """"x ....\\n\\n" at the end of a document: after the hard fail the epsilon slot lies behind the token start, and
    the reference calls Token(offset, buf) with offset > len(buf) -- string(buf[offset:]) panics where surfaces are
    printed (token_writer.go:85,93).  Reported as DTK_ST_BAD_OFFSET by the GPU path and by the oracle alike;
    neighbouring documents are untouched."""
    import datok_amd
    from datok_amd import corpus
    docs = [b"Vorher. ", b"x ....\n\n", b"a ....." + b"\n" * 400 + b" ", b"x ... \n", b"Nachher ...\n"]
    text, off = corpus.concat_docs(docs)
    om = oracle_models(model)
    for chunk in (0, 64, None):
        res, tot = run_batch(gpu(model), text, off, chunk=chunk)
        assert_batch_equals_oracle(om, res, text, off)
        st = [int(res.status[d]) for d in range(len(docs))]
        assert st[1] & datok_amd.ST_BAD_OFFSET and st[2] & datok_amd.ST_BAD_OFFSET
        assert st[0] == 0 and st[3] == 0 and st[4] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["tokenizer_de.matok", "tokenizer_de.datok"])
def test_blank_free_blob_is_flagged_quickly(gpu, oracle_models, model):
    """A blank-free blob of 2 MB inside a document (minified code, base64): the reference's 1024-rune window
    overflows, the document is out of contract.  Every lane whose window holds more bytes than 1024 runes can have
    stops, so the blob is not walked to its end by each of the 16 000 lanes whose chunk lies inside it (quadratic);
    the document comes back flagged and closed, its neighbours exact, in well under a second."""
    import time
    import datok_amd
    from datok_amd import corpus
    rng = np.random.default_rng(3)
    blob = bytes(rng.choice(np.frombuffer(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789+/", dtype=np.uint8),
                            size=2 << 20))
    docs = [b"Davor ein Satz. Und noch einer.", b"Anfang " + blob + b" Ende. Danach.", b"Danach ein Dokument."]
    text, off = corpus.concat_docs(docs)
    with datok_amd.Batch(len(text), len(off) - 1) as b:
        b.set_input(text, off)
        b.run(gpu(model), 0); b.totals()          # first run: allocations
        t0 = time.perf_counter()
        b.run(gpu(model), 0); tot = b.totals(); res = b.result()
        dt = time.perf_counter() - t0
    st = [int(x) for x in res.status]
    assert st[1] & datok_amd.ST_WINDOW_OVERFLOW and st[0] == 0 and st[2] == 0, st
    assert_batch_equals_oracle(oracle_models(model), res, text, off, docs=[0, 2])  # (the oracle is quadratic on the blob)
    assert dt < 0.5, dt


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["tokenizer_de.matok", "tokenizer_de.datok"])
def test_rich_corpus_is_exact(gpu, oracle_models, model):
    """The robustness corpus (30 000 word types with a Zipf tail, XML tags with attributes, URLs, e-mail addresses,
    abbreviations with blanks inside): speculation misses here (tags and tokens with blanks cost repair rounds),
    the result is exact all the same -- every document, default chunking and short chunks."""
    from datok_amd import corpus
    text, off = corpus.german_rich_docs(768, 4096, seed=11)
    om = oracle_models(model)
    for chunk in (None, 64):
        res, tot = run_batch(gpu(model), text, off, chunk=chunk, warm=16)
        assert tot["n_flagged"] == 0
        assert assert_batch_equals_oracle(om, res, text, off) == 768


@pytest.mark.gpu
def test_round1_soak_failure_stays_fixed(gpu, oracle_models):
    """The batch scripts/soak.py kept when it failed in round 1 (tests/golden/soak_fail_r01.npz: one 68 KB
    document with runs beyond the 1024-rune window, chunk 1024 / warm-up 48 / extend 240): over-long windows
    stop the lane and flag the document; everything the oracle defines must match (the parity helper compares
    only the WINDOW_OVERFLOW bit behind an overflow, where the reference has panicked)."""
    import datok_amd
    z = np.load(os.path.join(os.path.dirname(MODELS), "soak_fail_r01.npz"))
    name, text, off = str(z["model"]), z["text"], z["off"]
    with datok_amd.Batch(len(text), len(off) - 1) as b:
        b.set_chunking(int(z["chunk"]), int(z["warm"]), extend=int(z["extend"]))
        b.set_input(text, off)
        b.run(gpu(name), int(z["flags"]))
        res = b.result()
    assert_batch_equals_oracle(oracle_models(name), res, text, off, int(z["flags"]))


@pytest.mark.gpu
@pytest.mark.parametrize("pinned,prefetch", [(True, 0), (False, 0), (True, 127)])
def test_pipeline_slices_equal_the_oracle(gpu, oracle_models, pinned, prefetch):
    """dtk_pipeline: a ragged corpus cut into slices at document boundaries (by bytes and by document count), three
    batches in turn, uploads from page-locked memory (dtk_pinned_alloc) or from ordinary memory (page-locked for the
    call): every slice arrives in order, covers its documents exactly, and every document equals the oracle."""
    import datok_amd
    from datok_amd import corpus
    text, off = corpus.english_zipf_docs(3000, seed=9, max_bytes=16384)
    tok, om = gpu("tokenizer_en.matok"), oracle_models("tokenizer_en.matok")
    buf = None
    if pinned:
        buf = datok_amd.PinnedBuffer(len(text))
        buf.array[:] = text
        src = buf.array
    else:
        src = text
    seen, n_tok = [], [0]

    def on_slice(first, n, b):
        assert (not seen and first == 0) or first == seen[-1][0] + seen[-1][1]
        seen.append((first, n))
        res, tot = b.result(), b.totals()
        assert tot["n_docs"] == n and tot["n_flagged"] == 0
        n_tok[0] += tot["n_tokens"]
        sub_off = (off[first:first + n + 1] - off[first]).astype(np.uint64)
        sub = text[int(off[first]):int(off[first + n])]
        assert_batch_equals_oracle(om, res, sub, sub_off, docs=range(0, n, 5))
    with datok_amd.Pipeline(1 << 20, 400, depth=3) as p:
        if prefetch:  # every slice's arrays come to the host under the next slices' work (dtk_pipeline_set_result_fields)
            p.set_result_fields(prefetch)
        p.run(tok, src, off, 0, on_slice)
        assert seen[-1][0] + seen[-1][1] == 3000 and len(seen) >= 8
        assert all(n <= 400 for _, n in seen)
        counts = om.count_batch(text, off, 4)
        assert n_tok[0] == int(counts[:, 0].sum())
        # a second corpus through the same pipeline, no callback
        p.run(tok, src[:int(off[100])], off[:101], 0, None)
    if buf is not None:
        buf.close()


@pytest.mark.gpu
def test_result_fields_select_what_comes_to_the_host(gpu, oracle_models):
    """dtk_batch_set_result_fields: only the selected arrays are copied (page-locked buffers, asynchronous chain on the
    batch's download stream); what was not selected comes back NULL / empty; a later, wider selection for the same run
    fetches the rest; a new run starts afresh."""
    import datok_amd
    from datok_amd import corpus
    B = datok_amd.Batch
    text, off = corpus.german_docs(300, 1500, seed=21)
    tok, om = gpu("tokenizer_de.matok"), oracle_models("tokenizer_de.matok")
    with B(len(text), 300) as b:
        b.set_input(text, off)
        b.run(tok, 0)
        b.set_result_fields(B.R_TOK_RUNE | B.R_SENT | B.R_CSR | B.R_STATUS)
        b.download_begin()
        r = b.result()
        assert len(r.tok_rstart) == b.totals()["n_tokens"] and len(r.tok_bstart) == 0 and r.ev_bits.shape[1] == 0
        assert len(r.text_tok_end) == 0 and len(r.status) == 300
        for d in range(0, 300, 7):
            exp = oracle_doc(om, text[int(off[d]):int(off[d + 1])].tobytes())
            a, e = int(r.tok_off[d]), int(r.tok_off[d + 1])
            assert np.array_equal(r.tok_rstart[a:e], exp["tok_rstart"]) and np.array_equal(r.tok_rend[a:e], exp["tok_rend"])
            s0, s1 = int(r.sent_off[d]), int(r.sent_off[d + 1])
            assert np.array_equal(r.sent[s0:s1], exp["sent"])
        b.set_result_fields(B.R_ALL)            # the same run, everything: the missing arrays are fetched now
        assert_batch_equals_oracle(om, b.result(), text, off, docs=range(0, 300, 11))
        views = b.result(copy=False)            # views of the page-locked buffers themselves
        assert views.tok_rstart.flags["OWNDATA"] is False and np.array_equal(views.tok_rstart, r.tok_rstart)
        b.set_result_fields(B.R_EVENTS | B.R_TOK_BYTE | B.R_CSR | B.R_STATUS)   # what a closure replay needs
        text2, off2 = corpus.german_docs(200, 900, seed=22)
        b.set_input(text2, off2)
        b.run(tok, 0)
        r2 = b.result()
        assert len(r2.tok_rstart) == 0 and len(r2.tok_bstart) == b.totals()["n_tokens"] and r2.ev_bits.shape[1] > 0
        for d in (0, 57, 199):
            exp = oracle_doc(om, text2[int(off2[d]):int(off2[d + 1])].tobytes())
            a, e = int(r2.tok_off[d]), int(r2.tok_off[d + 1])
            assert np.array_equal(r2.tok_bstart[a:e], exp["tok_bstart"]) and np.array_equal(r2.tok_bend[a:e], exp["tok_bend"])


@pytest.mark.gpu
def test_rune_offsets_as_int16_pairs(gpu, oracle_models):
    """DTK_R_TOK_RUNE16: a token's rune offsets as the halves of one word -- half the bytes on the link.  Equal to the
    oracle's offsets (the -1 of token_writer.go:66-68 included); a batch with a document longer than 32 767 bytes gets
    the 32-bit arrays in its place; alone, in a pipeline, and beside the 32-bit arrays."""
    import datok_amd
    from datok_amd import corpus
    B = datok_amd.Batch
    tok, om = gpu("tokenizer_de.matok"), oracle_models("tokenizer_de.matok")
    docs = [b"\nThis.\n\x04\nAnd.\n\x04\n", b"", b"Tree\n\x04\n"]
    t0, o0 = corpus.german_rich_docs(400, 1200, seed=31)
    docs += [t0[int(o0[d]):int(o0[d + 1])].tobytes() for d in range(400)]
    text, off = corpus.concat_docs(docs)
    narrow = B.R_TOK_RUNE16 | B.R_SENT | B.R_CSR | B.R_STATUS | B.R_TEXTS
    for flags in (0, 16, 256 | 512):
        with B(len(text), len(docs)) as b:
            b.set_input(text, off)
            b.set_result_fields(narrow)
            for _ in range(2):
                b.run(tok, flags)
                r = b.result()
                assert r.tok_r16.shape == (b.totals()["n_tokens"], 2) and len(r.tok_rstart) == 0 and len(r.tok_bstart) == 0
                assert_batch_equals_oracle(om, r, text, off, flags & 16, fields=("tok_rstart", "tok_rend", "sent", "text_tok_end",
                                                                                "text_sent_end"))
            b.set_result_fields(narrow | B.R_TOK_RUNE)  # both forms of the same run
            r = b.result()
            assert np.array_equal(r.tok_r16[:, 0].astype(np.int32), r.tok_rstart) and np.array_equal(r.tok_r16[:, 1].astype(np.int32), r.tok_rend)
    # a document of 40 000 bytes: its offsets do not fit
    long_text, long_off = corpus.concat_docs([docs[5] * 40, docs[6], (docs[7] + b" ") * 60][:3])
    assert int(np.diff(long_off.astype(np.int64)).max()) > 32767
    with B(len(long_text), 3) as b:
        b.set_input(long_text, long_off)
        b.set_result_fields(narrow)
        b.run(tok, 0)
        r = b.result()
        assert r.tok_r16.shape[0] == 0 and len(r.tok_rstart) == b.totals()["n_tokens"]
        assert_batch_equals_oracle(om, r, long_text, long_off, fields=("tok_rstart", "tok_rend", "sent"))
        b.run(tok, 256 | 1024)                     # DTK_NO_RUNE_OFFSETS: neither form
        r = b.result()
        assert r.tok_r16.shape[0] == 0 and len(r.tok_rstart) == 0
    # slices of a pipeline
    seen = [0]

    def on_slice(first, n, b):
        r = b.result(copy=False)
        assert r.tok_r16.shape[0] == b.totals()["n_tokens"] and len(r.tok_rstart) == 0
        sub_off = (off[first:first + n + 1] - off[first]).astype(np.uint64)
        assert_batch_equals_oracle(om, r, text[int(off[first]):int(off[first + n])], sub_off, docs=range(0, n, 5),
                                   fields=("tok_rstart", "tok_rend", "sent"))
        seen[0] += n
    with datok_amd.Pipeline(1 << 17, 128, depth=3) as p:
        p.set_result_fields(narrow)
        p.run(tok, text, off, 256 | 512, on_slice)
    assert seen[0] == len(docs)


@pytest.mark.gpu
def test_pipeline_survives_a_failing_callback(gpu):
    """a callback that raises must not leave a Batch view that owns the pipeline's batch (the view's
    __del__ would free it a second time).  The pipeline is used again and closed afterwards."""
    import datok_amd
    from datok_amd import corpus
    text, off = corpus.german_docs(64, 2048, seed=5)
    tok = gpu("tokenizer_de.matok")
    kept = []

    def bad(first, n, b):
        kept.append(b)  # (keeps the view alive beyond the callback, as a traceback would)
        raise RuntimeError("boom")
    with datok_amd.Pipeline(32 << 10, 16, depth=3) as p:
        with pytest.raises(RuntimeError):
            p.run(tok, text, off, 0, bad)
        assert kept[0]._h is None
        kept.clear()
        seen = []
        p.run(tok, text, off, 0, lambda first, n, b: seen.append(b.totals()["n_tokens"]))
        assert len(seen) == 4 and all(seen)


@pytest.mark.gpu
@pytest.mark.parametrize("devices,prefetch", [([0, 0], 127), ([0], 0), ([0, 0, 0], 0)])
def test_multi_device_pipeline_equals_the_oracle(gpu, oracle_models, devices, prefetch):
    """dtk_multi: one worker thread, model replica and pipeline per listed device, slices dealt round-robin, handed to
    the callback in corpus order (the sharding behind the C-ABI).  The one-GPU box lists its
    device several times: the machinery (threads, hand-over, ordering, per-worker page-locked buffers) is the same."""
    import datok_amd
    from datok_amd import corpus
    text, off = corpus.english_zipf_docs(2500, seed=13, max_bytes=8192)
    om = oracle_models("tokenizer_en.matok")
    seen, n_tok = [], [0]

    def on_slice(first, n, b):
        assert (not seen and first == 0) or first == seen[-1][0] + seen[-1][1]
        seen.append((first, n))
        res, tot = b.result(), b.totals()
        assert tot["n_docs"] == n and tot["n_flagged"] == 0
        n_tok[0] += tot["n_tokens"]
        sub_off = (off[first:first + n + 1] - off[first]).astype(np.uint64)
        assert_batch_equals_oracle(om, res, text[int(off[first]):int(off[first + n])], sub_off, docs=range(0, n, 6))
    with datok_amd.MultiPipeline(os.path.join(MODELS, "tokenizer_en.matok"), devices, 1 << 19, 300, depth=3) as mp:
        assert mp.type() == "MATOK"
        if prefetch:
            mp.set_result_fields(prefetch)
        mp.run(text, off, 0, on_slice)
        assert seen[-1][0] + seen[-1][1] == 2500 and len(seen) >= 9
        assert n_tok[0] == int(om.count_batch(text, off, 4)[:, 0].sum())
        # again (the workers wait for the next run), without a callback; then a callback that fails
        mp.run(text[:int(off[700])], off[:701], 0, None)

        def bad(first, n, b):
            if first > 0:
                raise RuntimeError("boom")
        with pytest.raises(RuntimeError):
            mp.run(text, off, 0, bad)
        seen.clear(); n_tok[0] = 0
        mp.run(text, off, 0, on_slice)      # and it still works afterwards
        assert seen[-1][0] + seen[-1][1] == 2500


@pytest.mark.gpu
def test_leading_empty_documents_on_a_reused_batch(gpu, oracle_models):
    """k_symbolize's blocks clear the event bitmaps; block 0 must start at word 0 -- with 32 or more leading
    empty documents its first document is not document 0, and the words of the empty documents' positions kept the
    previous run's bits (the result stayed right only because those documents then went through the exact pass)."""
    import datok_amd
    from datok_amd import corpus
    tok, om = gpu("tokenizer_de.matok"), oracle_models("tokenizer_de.matok")
    t1, o1 = corpus.german_docs(300, 600, seed=31)
    body, ob = corpus.german_docs(200, 600, seed=32)
    t2 = body
    o2 = np.concatenate([np.zeros(100, np.uint64), ob])      # 100 empty documents in front
    with datok_amd.Batch(len(t1) + 1024, 400) as b:
        b.set_input(t1, o1); b.run(tok, 0); b.totals()           # leaves bits all over the first words
        b.set_input(t2, o2); b.run(tok, 0)
        res = b.result()
        v = datok_amd._lib.ResultView()
        datok_amd.lib().dtk_batch_result_device(b._h, C_byref(v))
        assert v.n_exact == 0                                     # nobody needed the exact pass
        assert_batch_equals_oracle(om, res, t2, o2, docs=list(range(0, 100, 9)) + list(range(100, 300, 7)), allow_status=2)
        assert all(int(res.tok_off[d + 1]) == int(res.tok_off[d]) for d in range(100))


@pytest.mark.gpu
def test_one_kind_of_token_offsets(gpu, oracle_models):
    """DTK_NO_BYTE_OFFSETS / DTK_NO_RUNE_OFFSETS: the compaction writes two of its four offset arrays; the others come
    back empty, everything else is unchanged.  EOT texts, tiny documents (the lane-per-document kernel) and the exact
    pass's rows included."""
    import datok_amd
    from datok_amd import corpus
    tok, om = gpu("tokenizer_de.matok"), oracle_models("tokenizer_de.matok")
    t1, o1 = corpus.german_docs(500, 700, seed=41)
    docs = [t1[int(o1[d]):int(o1[d + 1])].tobytes() for d in range(500)]
    docs[7] = "Erste.\n\n\x04\nNächst.\x04".encode()            # matrix_test.go:1308-1310
    docs[8] = b"This.\n\x04And.\n\x04\n"                          # token_writer_test.go:50
    docs += [b"Hallo Welt. " * 3] * 2600                            # enough tiny documents for k_compact_small
    text, off = corpus.concat_docs(docs)
    rune = ("tok_rstart", "tok_rend", "sent", "text_tok_end", "text_sent_end")
    byte = ("tok_bstart", "tok_bend", "sent", "text_tok_end", "text_sent_end")
    for flags, fields, empty in ((datok_amd.NO_BYTE_OFFSETS, rune, "tok_bstart"), (datok_amd.NO_RUNE_OFFSETS, byte, "tok_rstart"),
                                 (datok_amd.NO_BYTE_OFFSETS | NEWLINE_AFTER_EOT, rune, "tok_bend")):
        for chunk in (None, 0, 64):
            with datok_amd.Batch(len(text), len(docs)) as b:
                if chunk is not None:
                    b.set_chunking(chunk, 16)
                b.set_input(text, off)
                b.run(tok, flags)
                res = b.result()
                assert len(getattr(res, empty)) == 0
                n = assert_batch_equals_oracle(om, res, text, off, flags & NEWLINE_AFTER_EOT, docs=range(0, len(docs), 7),
                                               fields=fields)
                assert n > 400
                with pytest.raises(datok_amd.DatokGpuError):
                    b.render(3)      # (the renderer needs both kinds)
