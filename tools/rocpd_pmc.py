#!/usr/bin/env python
"""Per-kernel aggregation of rocprofv3 --pmc passes (rocpd .db):  python tools/rocpd_pmc.py a.db [b.db ...]
Prints, per kernel: calls, total duration, and the SUM of every collected counter (FETCH_SIZE / WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE under-counts wide coalesced reads by 2x -- MI355X_MICROARCH.md 'HBM')."""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:<>, ]+?)\(", name)
    return (m.group(1) if m else name)[:60]


def main(paths):
    agg = defaultdict(lambda: defaultdict(float))
    for path in paths:
        db = sqlite3.connect(path)
        seen = set()
        for name, disp, cname, val, dur in db.execute(
                "select name, dispatch_id, counter_name, counter_value, duration from pmc_events"):
            k = short(name)
            agg[k][cname] += val
            if (path, disp) not in seen:
                seen.add((path, disp))
                agg[k]["calls@" + path] += 1
                agg[k]["ns@" + path] += dur
    counters = sorted({c for a in agg.values() for c in a if "@" not in c})
    print("# " + " ".join(paths))
    print(f"{'kernel':62s} {'calls':>6s} {'ms':>9s} " + " ".join(f"{c[:22]:>22s}" for c in counters))
    def tot(a):
        return max([v for c, v in a.items() if c.startswith("ns@")] or [0])
    for k, a in sorted(agg.items(), key=lambda kv: -tot(kv[1])):
        calls = max([v for c, v in a.items() if c.startswith("calls@")] or [0])
        print(f"{k:62s} {int(calls):6d} {tot(a) / 1e6:9.3f} " + " ".join(f"{a.get(c, 0):22.4g}" for c in counters))


if __name__ == "__main__":
    main(sys.argv[1:])
