"""Shared evaluation of the transcribed reference expectations (tests/golden/*.json)."""
import json
import os
import re

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def model_file(name):
    """Fixture file for a golden call's model.  `fst:<net>:<kind>` is a tokenizer the reference's test
    builds from a Foma net (LoadFomaFile(...).ToMatrix() / .ToDoubleArray()): the matrix is built from
    the net here too; ToDoubleArray is not rebuilt (offline construction), so its cases run on the
    shipped double array of the same net where the reference ships one (simpletok.datok) and on the
    matrix of the net otherwise -- the expected tokens do not depend on the table encoding."""
    if name.startswith("fst:"):
        _, net, kind = name.split(":")
        if kind == "datok" and net == "simpletok.fst":
            return "simpletok.datok"
        return net
    return name


def load_cases(include_fst=True):
    stale = json.load(open(os.path.join(GOLDEN, "stale_sites.json"), encoding="utf-8"))
    out = []
    for stem in ("matrix", "datok", "token_writer"):
        d = json.load(open(os.path.join(GOLDEN, stem + "_goldens.json"), encoding="utf-8"))
        for case in d["cases"]:
            fname, line = case["src"].split(":")
            case["stale"] = (int(line) in stale.get(fname, [])
                             or (case.get("file") == "de/dontsplit.txt"
                                 and case["calls"][0]["input"] in stale["dontsplit_stale_inputs"]))
            case["needs_fst"] = any(c["model"].startswith("fst:") for c in case["calls"])
            if case["needs_fst"] and not include_fst:
                continue
            out.append(case)
    return out


def golden_strings():
    d = json.load(open(os.path.join(GOLDEN, "matrix_goldens.json"), encoding="utf-8"))
    return d["strings"]


def failed_checks(case, rendered: str):
    """Returns the checks of `case` that `rendered` (the concatenated writer output) fails."""
    s = rendered
    ttok = re.split("\n+", s)[:-1]   # datok_test.go:23-33 ttokenize
    nl = s.split("\n")
    nlnl = s.split("\n\n")
    bad = []
    for ch in case["checks"]:
        k, v, i = ch["kind"], ch.get("value"), ch.get("index")
        try:
            ok = {
                "raw_eq": lambda: s == v,
                "contains": lambda: v in s,
                "ttok_at": lambda: ttok[i] == v,
                "ttok_len": lambda: len(ttok) == v,
                "ttok_join": lambda: "\n".join(ttok) == v,
                "ttok_len_gt": lambda: len(ttok) > v,
                "nl_at": lambda: nl[i] == v,
                "nl_len": lambda: len(nl) == v,
                "nlnl_at": lambda: nlnl[i] == v,
                "nlnl_len": lambda: len(nlnl) == v,
            }[k]()
        except IndexError:
            ok = False
        if not ok:
            bad.append(ch)
    return bad
