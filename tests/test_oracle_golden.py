"""Pins the CPU oracle (oracle/datok_oracle.c) to the reference's own golden vectors.

Every expectation transcribed from matrix_test.go / datok_test.go /
token_writer_test.go that the shipped fixtures satisfy must hold for the
oracle; the stale ones (tests/golden/stale_sites.json) must fail identically
for both table encodings.
"""
import os

import pytest

from goldens import failed_checks, golden_strings, load_cases
from oracle import oracle as O

CASES = load_cases()


def _render(case, models):
    out = b""
    for c in case["calls"]:
        o, status = models(c["model"]).transduce(c["input"].encode("utf-8"), c["flags"])
        assert status == 0
        out += o
    return out.decode("utf-8")


def test_golden_volume():
    live = [c for c in CASES if not c["stale"]]
    assert sum(len(c["checks"]) for c in live) >= 900
    assert len([c for c in CASES if c["stale"]]) >= 17


@pytest.mark.parametrize("case", [c for c in CASES if not c["stale"]],
                         ids=lambda c: c["src"] + ("" if "file" not in c else ":" + c["calls"][0]["input"]))
def test_oracle_matches_reference_expectation(case, oracle_models):
    bad = failed_checks(case, _render(case, oracle_models))
    assert not bad, (case["src"], bad[:3])


def test_stale_expectations_fail_the_same_way_in_both_encodings(oracle_models):
    """the shipped de binaries predate some datok_test.go expectations."""
    n = 0
    for case in CASES:
        if not case["stale"]:
            continue
        assert failed_checks(case, _render(case, oracle_models)), case["src"]
        for c in case["calls"]:
            a, _ = oracle_models("tokenizer_de.datok").transduce(c["input"].encode(), c["flags"])
            b, _ = oracle_models("tokenizer_de.matok").transduce(c["input"].encode(), c["flags"])
            assert a == b
        n += 1
    assert n >= 17 + 40


def test_matok_datok_equivalence(oracle_models):
    """matrix_test.go:1248-1275 TestMatokDatokEquivalence on the benchmark string s."""
    s = golden_strings()["s"].encode("utf-8")
    assert len(s) == 750
    a, _ = oracle_models("tokenizer_de.datok").transduce(s)
    b, _ = oracle_models("tokenizer_de.matok").transduce(s)
    assert a == b and a.count(b"\n") > 130


def test_loader_dispatch_and_header_ids(oracle_models):
    """datok_test.go:252-261 (Type), :240-242 (epsilon/unknown/identity)."""
    assert oracle_models("simpletok.datok").type() == "DATOK"
    assert oracle_models("simpletok.matok").type() == "MATOK"
    info = oracle_models("tokenizer_de.datok").info
    assert (info["epsilon"], info["unknown"], info["identity"]) == (1, 2, 3)


def test_format_kat_sizes():
    """matrix_test.go:167 (230 bytes) and datok_test.go:186 (296 bytes) for simpletok."""
    import gzip
    from conftest import MODELS
    assert len(gzip.open(os.path.join(MODELS, "simpletok.matok")).read()) == 230
    assert len(gzip.open(os.path.join(MODELS, "simpletok.datok")).read()) == 296


def test_token_writer_alone():
    """token_writer_test.go:11-32: Token(0,"abc") Token(1,"def") SentenceEnd TextEnd -> abc\\nef\\n\\n\\n.

    The oracle's writer is only reachable through a walk; "abc def" through
    simpletok exercises the same offset-1 surface cut (leading blank skipped)."""
    from conftest import MODELS
    m = O.Model(os.path.join(MODELS, "simpletok.matok"))
    out, _ = m.transduce(b"abc def")
    assert out == b"abc\ndef\n\n\n"
    ev, _ = m.events(b"abc def")
    assert ev[0][:2] == (0, 0) and ev[1][:2] == (0, 1)   # Token(0,..), Token(1,..)


def test_loader_rejects_garbage(tmp_path):
    import gzip
    p = tmp_path / "bad.matok"
    p.write_bytes(gzip.compress(b"NOTOK" + b"\0" * 64))
    with pytest.raises(ValueError):
        O.Model(str(p))
    q = tmp_path / "plain.matok"
    q.write_bytes(b"MATOK" + b"\0" * 64)      # not gzip: gzip.NewReader fails upstream
    with pytest.raises(ValueError):
        O.Model(str(q))


def test_go_utf8_decoder_spec():
    """Go unicode/utf8.DecodeRune: invalid -> (U+FFFD, 1). Unpinned by the reference's tests."""
    assert O.decode_rune("ä".encode()) == (0xE4, 2)
    assert O.decode_rune("€".encode()) == (0x20AC, 3)
    assert O.decode_rune("😀".encode()) == (0x1F600, 4)
    assert O.decode_rune(b"\x80") == (0xFFFD, 1)
    assert O.decode_rune(b"\xc0\xaf") == (0xFFFD, 1)          # overlong
    assert O.decode_rune(b"\xed\xa0\x80") == (0xFFFD, 1)      # surrogate
    assert O.decode_rune(b"\xf4\x90\x80\x80") == (0xFFFD, 1)  # > U+10FFFF
    assert O.decode_rune(b"\xe2\x82") == (0xFFFD, 1)          # truncated
