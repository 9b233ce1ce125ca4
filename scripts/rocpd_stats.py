#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 rocpd database (ROCm 7.2 writes SQLite by default):
dispatch count, average / min / max duration, share of the GPU time; PMC counters averaged per dispatch.
usage: rocpd_stats.py results.db [skip_first_n_dispatches_per_kernel]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cur = db.cursor()
rows = cur.execute("select k.kernel_name, d.start, d.end, d.id, d.event_id from rocpd_kernel_dispatch d "
                   "join rocpd_info_kernel_symbol k on k.id = d.kernel_id order by d.start").fetchall()
by = collections.defaultdict(list)
for name, s, e, did, ev in rows:
    by[name.split("(")[0]].append((e - s, ev))
tot = sum(sum(x[0] for x in v[skip:]) for v in by.values()) or 1
print("%-72s %6s %10s %10s %10s %6s" % ("kernel", "calls", "avg us", "min us", "max us", "%"))
for name, v in sorted(by.items(), key=lambda kv: -sum(x[0] for x in kv[1][skip:])):
    d = [x[0] for x in v[skip:]]
    if not d:
        continue
    print("%-72s %6d %10.2f %10.2f %10.2f %6.1f" % (name[:72], len(d), sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3,
                                                     100.0 * sum(d) / tot))
try:
    pm = cur.execute("select e.event_id, p.name, e.value from rocpd_pmc_event e join rocpd_info_pmc p on p.id = e.pmc_id").fetchall()
except sqlite3.Error:
    pm = []
if pm:
    ev2k = {}
    for name, v in by.items():
        for i, (dur, ev) in enumerate(v):
            if i >= skip:
                ev2k[ev] = name
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    seen = set()
    for ev, cname, val in pm:
        k = ev2k.get(ev)
        if k is None:
            continue
        agg[k][cname] += val
        if (ev, cname) not in seen:
            seen.add((ev, cname)); cnt[(k, cname)] += 1
    print()
    for k, cs in agg.items():
        print(k[:90])
        for c, val in sorted(cs.items()):
            print("    %-32s %18.1f per dispatch" % (c, val / max(1, cnt[(k, c)])))
