"""The C-ABI library loads and exports every symbol include/datok_gpu.h declares.
No compute call is made here (no GPU in the CPU test tier)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "datok_gpu.h"), encoding="utf-8").read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dtk_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_the_header():
    import datok_amd
    path = datok_amd.build()
    lib = ctypes.CDLL(path)
    names = declared_symbols()
    assert len(names) >= 24
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # the Python binding knows every export too
    from datok_amd import _lib
    assert sorted(_lib.EXPORTS) == names


def test_no_cpu_fallback_without_a_device():
    """On a box without a HIP device the product path must fail loudly, never fall back."""
    import datok_amd
    from datok_amd import _lib
    L = datok_amd.lib()
    if L.dtk_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.DatokGpuError) as e:
        datok_amd.load_tokenizer_file(os.path.join(ROOT, "tests", "golden", "models", "simpletok.matok"))
    assert e.value.code == _lib.E_NO_DEVICE
    with pytest.raises(_lib.DatokGpuError):
        datok_amd.Batch(1024, 4)
    with pytest.raises(_lib.DatokGpuError) as e:
        datok_amd.MultiPipeline(os.path.join(ROOT, "tests", "golden", "models", "simpletok.matok"), [0, 0], 1 << 16, 16)
    assert e.value.code == _lib.E_NO_DEVICE
    assert b"no CPU path" in L.dtk_strerror(_lib.E_NO_DEVICE)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under datok_amd/ or include/ may reference it."""
    for base in ("datok_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", "Makefile")):
                    txt = open(os.path.join(dirpath, f), encoding="utf-8", errors="replace").read()
                    assert "liboracle" not in txt and "datok_oracle" not in txt, os.path.join(dirpath, f)
                    assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), os.path.join(dirpath, f)


def test_error_strings_and_constants():
    import datok_amd
    L = datok_amd.lib()
    for code in range(0, -9, -1):
        assert L.dtk_strerror(code)
    assert (datok_amd.TOKENS, datok_amd.SENTENCES, datok_amd.TOKEN_POS, datok_amd.SENTENCE_POS,
            datok_amd.NEWLINE_AFTER_EOT, datok_amd.SIMPLE) == (1, 2, 4, 8, 16, 3)   # token_writer.go:17-25


def test_library_never_reads_the_environment():
    """kernel selection of a shipped library must not depend on the caller's environment.  No getenv in the
    native sources; the test hooks go through dtk_debug_configure (which this harness feeds from DATOK_* variables)."""
    for base in (os.path.join("datok_amd", "csrc"), "include"):
        for f in os.listdir(os.path.join(ROOT, base)):
            if f.endswith((".cpp", ".hip", ".h", ".hpp")):
                txt = open(os.path.join(ROOT, base, f), encoding="utf-8", errors="replace").read()
                assert "getenv" not in txt, f
    import datok_amd
    from datok_amd import _lib
    L = datok_amd.lib()
    assert L.dtk_debug_configure(b"DATOK_NO_SUCH_SWITCH", b"1") == _lib.E_ARG
    assert L.dtk_debug_configure(b"WARM_WS", b"0") == _lib.OK          # (its default: nothing changes)
    assert L.dtk_debug_configure(b"DATOK_WARM_MIN", b"0") == _lib.OK


def test_python_structs_mirror_the_header(tmp_path):
    """The ctypes structures of datok_amd/_lib.py against the C compiler's layout of include/datok_gpu.h: a member
    added to the header and not to the mirror makes the library write behind the Python object."""
    import subprocess
    from datok_amd import _lib
    pairs = {"dtk_result_view": _lib.ResultView, "dtk_totals": _lib.Totals, "dtk_render_view": _lib.RenderView}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "datok_gpu.h"', 'int main(void) {']
    for cname, cls in pairs.items():
        src.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            src.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    src.append("return 0; }")
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(c)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    for cname, cls in pairs.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got["%s.%s" % (cname, fname)]) == getattr(cls, fname).offset, (cname, fname)
