"""CPU tests of the host side: the Python and C++ mirrors of NewTokenWriter and
the replay of event bytes.  The GPU walk is replaced here by the oracle's call
list (converted to the event-byte form the kernels store); the code under test
-- new_token_writer, replay, datok.hpp -- is the product's host code."""
import io
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from goldens import failed_checks, load_cases

EV_S_EOT, EV_E_EOT, EV_TOK_END, EV_S_EPS, EV_S_EPS2, EV_S_EOF, EV_E_EOF = 1, 2, 4, 8, 16, 32, 64


def _advance(raw, pos, k):
    """Byte offset after k runes from pos (Go decoding: an invalid byte is one rune)."""
    from datok_amd.host import _decode_runes
    for _ in range(k):
        r = _decode_runes(raw[pos:pos + 4])[0]
        valid_w = len(chr(r).encode("utf-8"))
        pos += valid_w if raw[pos:pos + valid_w] == chr(r).encode("utf-8") else 1
    return pos


def matrix_events(om, raw: bytes):
    """Event bytes + token start offsets a matrix walk stores, rebuilt from the oracle's calls.
    SentenceEnd/TextEnd carry buffc (runes since the last rewind, matrix.go:575,597,600)."""
    calls, _ = om.events(raw)
    n = len(raw)
    ev = np.zeros(n + 1, dtype=np.uint8)
    starts = []
    B = 0
    for i, (kind, a, b, c, d) in enumerate(calls):
        if kind == 0:
            ev[d] |= EV_TOK_END
            starts.append(c)
            B = d
            continue
        p = _advance(raw, B, a)
        eot_here = p > 0 and raw[p - 1] == 4 and not (ev[p] & EV_E_EOT)
        if kind == 2:
            if eot_here:
                ev[p] |= EV_E_EOT
                B = p                      # matrix.go:601 rewinds
            else:
                ev[p] |= EV_E_EOF
        else:
            nxt = calls[i + 1] if i + 1 < len(calls) else None
            if eot_here and nxt is not None and nxt[0] == 2 and _advance(raw, B, nxt[1]) == p:
                ev[p] |= EV_S_EOT
            elif not (ev[p] & EV_S_EPS):
                ev[p] |= EV_S_EPS
            elif not (ev[p] & EV_S_EPS2):
                ev[p] |= EV_S_EPS2
            else:
                ev[p] |= EV_S_EOF
    return ev, np.array(starts, dtype=np.uint32)


TEXTS = ["Der alte Mann. Er ging!", "This.\n\x04And.\n\x04\n", "\nThis.\n\x04\nAnd.\n\x04\n",
         "„Hallo“, sagte er. »Nein!«", "Tree\n\x04\n", "a", "tra. u Du?", "  Erste."]


@pytest.mark.parametrize("flags", [3, 1, 2, 4, 8, 12, 7, 15, 31, 28, 20])
def test_python_token_writer_matches_oracle(oracle_models, flags):
    """new_token_writer (token_writer.go:36-175 mirror) driven by the oracle's call list prints
    what the oracle's own writer prints."""
    import datok_amd
    from datok_amd.host import _decode_runes
    om = oracle_models("tokenizer_de.matok")
    for text in TEXTS:
        raw = text.encode()
        exp, st = om.transduce(raw, flags)
        if st:
            continue
        calls, _ = om.events(raw)
        w = io.BytesIO()
        tw = datok_amd.new_token_writer(w, flags)
        for kind, a, b, c, d in calls:
            if kind == 0:
                tw.Token(a, _decode_runes(raw[b:d]))
            elif kind == 1:
                tw.SentenceEnd(a)
            else:
                tw.TextEnd(a)
        tw.Flush()
        assert w.getvalue() == exp, (text, flags)


@pytest.mark.parametrize("flags", [3, 7, 12, 28])
def test_event_replay_matches_oracle(oracle_models, flags):
    """host.replay turns event bytes back into the reference's call sequence."""
    import datok_amd
    from datok_amd.host import replay
    om = oracle_models("tokenizer_de.matok")
    for text in TEXTS:
        raw = text.encode()
        exp, st = om.transduce(raw, flags)
        if st:
            continue
        ev, starts = matrix_events(om, raw)
        w = io.BytesIO()
        tw = datok_amd.new_token_writer(w, flags)
        replay(True, raw, ev, starts, tw)
        tw.Flush()
        assert w.getvalue() == exp, (text, flags)


def test_golden_token_writer_cases_through_replay(oracle_models):
    """token_writer_test.go:45-108 (the only offset goldens upstream) through event replay."""
    import datok_amd
    from datok_amd.host import replay
    cases = [c for c in load_cases() if c["src"].startswith("token_writer_test.go")]
    assert len(cases) == 8
    om = oracle_models("tokenizer_de.matok")
    for case in cases:
        w = io.BytesIO()
        for c in case["calls"]:
            raw = c["input"].encode()
            ev, starts = matrix_events(om, raw)
            tw = datok_amd.new_token_writer(w, c["flags"])
            replay(True, raw, ev, starts, tw)
            tw.Flush()
        assert not failed_checks(case, w.getvalue().decode()), (case["src"], w.getvalue())


def test_cpp_mirror_token_writer_alone(tmp_path):
    """token_writer_test.go:11-32 through include/datok.hpp: Token(0,"abc") Token(1,"def")
    SentenceEnd TextEnd -> "abc\\nef\\n\\n\\n"; plus the position modes."""
    import datok_amd
    lib = datok_amd.build()
    src = tmp_path / "tw.cpp"
    src.write_text(r'''
#include <iostream>
#include <sstream>
#include "datok.hpp"
int main() {
  std::ostringstream os;
  auto tws = datok::NewTokenWriter(os, datok::SIMPLE);
  tws->Token(0, {U'a', U'b', U'c'});
  tws->Token(1, {U'd', U'e', U'f'});
  tws->SentenceEnd(0);
  tws->TextEnd(0);
  tws->Flush();
  std::cout << os.str();
  std::ostringstream os2;
  auto tw2 = datok::NewTokenWriter(os2, datok::TOKEN_POS | datok::SENTENCE_POS);
  tw2->Token(0, {U'a', U'b'});
  tw2->Token(1, {U' ', U'c'});
  tw2->SentenceEnd(0);
  tw2->TextEnd(0);
  tw2->Flush();
  std::cout << os2.str();
  return 0;
}
''')
    exe = tmp_path / "tw"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           lib, "-Wl,-rpath," + os.path.dirname(lib)])
    out = subprocess.check_output([str(exe)])
    assert out == b"abc\nef\n\n\n" + b"0 2 3 4\n0 4\n"


def test_corpus_generators_respect_their_contract():
    """valid UTF-8, no U+0004, fixed sizes, deterministic."""
    from datok_amd import corpus
    t, off = corpus.german_docs(64, 4096, seed=2)
    assert len(t) == 64 * 4096 and off[-1] == len(t) and bytes(t).decode("utf-8") and 4 not in t
    t2, _ = corpus.german_docs(64, 4096, seed=2)
    assert np.array_equal(t, t2)
    te, oe = corpus.english_zipf_docs(256, seed=3)
    lens = np.diff(oe.astype(np.int64))
    assert lens.min() >= 64 and lens.max() <= 65536 and bytes(te).decode("utf-8")
    ts, os_ = corpus.simple_ascii(1, 1024)
    assert len(ts) == 1024 and ts[-1] == ord(".")
