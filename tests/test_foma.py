"""Foma text net -> matrix tokenizer: LoadFomaFile/ParseFoma (fomafile.go:56-450)
+ Automaton.ToMatrix (matrix.go:30-99) + MatrixTokenizer.Save (matrix.go:107-210).

Pinned against the reference's own artefacts: the .matok files it ships were produced from the
.fst files it ships, so converting the latter must reproduce the former bit for bit -- for the
oracle's restatement and for the product's host code (dtk_foma_to_matok needs no device)."""
import gzip
import os

import numpy as np
import pytest

from conftest import MODELS
from oracle import oracle as O

PAIRS = ["simpletok", "clitic_test", "tokenizer_de", "tokenizer_en"]


def _read(name):
    with open(os.path.join(MODELS, name), "rb") as f:
        return f.read()


@pytest.mark.parametrize("stem", PAIRS)
def test_oracle_foma_to_matrix_equals_shipped_matok(stem):
    a, b = O.Model(os.path.join(MODELS, stem + ".fst")), O.Model(os.path.join(MODELS, stem + ".matok"))
    assert a.type() == "MATOK"
    assert a.info == b.info
    assert np.array_equal(a.array(), b.array())
    assert np.array_equal(a.sigma_ascii(), b.sigma_ascii())


@pytest.mark.parametrize("stem", PAIRS)
def test_product_convert_equals_shipped_matok(stem):
    """`datok convert -f x.fst -t x.matok`: the decompressed image is the shipped file's."""
    import datok_amd
    img = datok_amd.foma_to_matok(_read(stem + ".fst"))
    assert img[:2] == b"\x1f\x8b"
    assert gzip.decompress(img) == gzip.decompress(_read(stem + ".matok"))


def test_product_convert_is_loadable_by_the_oracle(tmp_path):
    import datok_amd
    for stem in ("bauamt", "wahlamt", "ignorable_mcs"):
        if stem == "ignorable_mcs":   # has an identity symbol: round-trips through the file format
            p = tmp_path / (stem + ".matok")
            p.write_bytes(datok_amd.foma_to_matok(_read(stem + ".fst")))
            a, b = O.Model(str(p)), O.Model(os.path.join(MODELS, stem + ".fst"))
            assert np.array_equal(a.array()[:len(b.array())], b.array()) or np.array_equal(a.array(), b.array())
            for s in (b"ab<ab>a", b"a<ab>b", "a ü <ab> \x04 b".encode()):
                assert a.transduce(s) == b.transduce(s)
        else:                         # no identity: the header would carry uint16(-1) (matrix.go:158)
            raw = gzip.decompress(datok_amd.foma_to_matok(_read(stem + ".fst")))
            assert raw[:5] == b"MATOK" and raw[5 + 6:5 + 8] == b"\xff\xff"


def test_identityless_net_walk():
    """bauamt.fst has no @_IDENTITY_SYMBOL_@: sigmaASCII stays zero (matrix.go:43-48) and a rune
    outside sigma has symbol 0 (matrix.go:427-435), which never has an arc (matrix.go:459)."""
    m = O.Model(os.path.join(MODELS, "bauamt.fst"))
    assert (m.info["identity"], m.info["unknown"]) == (-1, -1)
    assert m.transduce(b"ibauamt")[0] == b"i\nbauamt\n\n\n"
    assert m.transduce("bauäbau日amt".encode())[0] == "bau\nä\nbau\n日\na\nm\nt\n\n\n".encode()


def _gz(text):
    return gzip.compress(text.encode())


NET = ("##foma-net 1.0##\n##props##\n1 %d %d 3 1 1 %s 1 1 %s 1 2 x\n##sigma##\n0 @_EPSILON_SYMBOL_@\n"
       "3 a\n4 @_TOKEN_BOUND_@\n##states##\n%s-1 -1 -1 -1 -1\n##end##\n")


def test_product_convert_rejections():
    import datok_amd
    from datok_amd import _lib
    ok = NET % (2, 2, "1", "1", "0 3 3 1 0\n1 0 4 0 1\n")
    img = datok_amd.foma_to_matok(_gz(ok))
    assert gzip.decompress(img)[:5] == b"MATOK"
    for bad, code in [
        (NET % (2, 2, "0", "1", "0 3 3 1 0\n1 0 4 0 1\n"), _lib.E_MODEL),    # not deterministic (:159)
        (NET % (2, 2, "1", "0", "0 3 3 1 0\n1 0 4 0 1\n"), _lib.E_MODEL),    # not epsilon free (:164)
        (NET % (2, 2, "1", "1", "0 3 4 1 0\n"), _lib.E_MODEL),               # a:TOKEN_BOUND unsupported (:290)
        (NET % (2, 2, "1", "1", "0 0 0 1 0\n"), _lib.E_MODEL),               # general epsilon arc (:306)
        ("no net at all\n", _lib.E_FORMAT),
    ]:
        with pytest.raises(_lib.DatokGpuError) as e:
            datok_amd.foma_to_matok(_gz(bad))
        assert e.value.code == code, bad
    with pytest.raises(_lib.DatokGpuError):
        datok_amd.foma_to_matok(b"plain text, not gzip")     # fomafile.go:63-67


def test_cli_convert(tmp_path):
    """`datok convert -i x.fst -o x.matok` (cmd/datok.go:50-70) over the C-ABI; host only."""
    import subprocess
    import datok_amd
    datok_amd.build()
    exe = os.path.join(os.path.dirname(datok_amd.__file__), "datok")
    out = tmp_path / "clitic.matok"
    r = subprocess.run([exe, "convert", "-i", os.path.join(MODELS, "clitic_test.fst"), "-o", str(out)],
                       capture_output=True)
    assert r.returncode == 0 and r.stdout == b"File successfully converted.\n"
    assert gzip.decompress(out.read_bytes()) == gzip.decompress(_read("clitic_test.matok"))
    r = subprocess.run([exe, "convert", "--foma=" + os.path.join(MODELS, "simpletok.fst"),
                        "--tokenizer=" + str(out), "-d"], capture_output=True)
    assert r.returncode == 1 and b"double array" in r.stderr           # ToDoubleArray is not provided
    r = subprocess.run([exe, "convert", "-i", str(tmp_path / "missing.fst"), "-o", str(out)], capture_output=True)
    assert r.returncode == 1 and b"Unable to load foma file" in r.stderr
    assert subprocess.run([exe, "tokenize", "-t", "x"], capture_output=True).returncode == 1   # missing <input>
