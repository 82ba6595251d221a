#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd .db (kernel-trace) into the per-kernel stats table rocprofv3 --stats prints:
   python tools/rocpd_stats.py gpurun_out/prof1/r01_results.db > profiles/r01_kernel_stats.txt"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:<>, ]+?)\(", name)
    return (m.group(1) if m else name)[:90]


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {namecol}, start, end from kernels").fetchall()
    agg = {}
    for n, s, e in rows:
        k = short(n)
        a = agg.setdefault(k, [0, 0, 10**18, 0])
        d = e - s
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    span = max(r[2] for r in rows) - min(r[1] for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    print(f"# {len(rows)} dispatches, {len(agg)} kernels, total kernel time {tot / 1e6:.2f} ms, first->last span {span / 1e6:.2f} ms")
    print(f"{'kernel':92s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:92s} {a[0]:7d} {a[1] / 1e6:10.3f} {a[1] / a[0] / 1e3:10.2f} {a[2] / 1e3:9.2f} {a[3] / 1e3:9.2f} {100.0 * a[1] / tot:6.2f}")


HEAVY = re.compile(r"gemm_f16x3|attn_fwd16|attn_bwd16|gemm_f32")


def timeline(path, tail_frac=0.5):
    """How the two student streams share the chip: over the last `tail_frac` of the trace (the timed steps), the time with
    >= 1 MFMA-heavy kernel (split GEMMs, attention) in flight, with only light kernels, and with nothing at all; and how
    long 2 heavy kernels overlap.  python tools/rocpd_stats.py --timeline x.db"""
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {namecol}, start, end from kernels order by start").fetchall()
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    lo = t1 - (t1 - t0) * tail_frac
    ev = []
    for n, s, e in rows:
        if e <= lo:
            continue
        h = 1 if HEAVY.search(n) else 0
        ev.append((max(s, lo), 1, h))
        ev.append((e, -1, h))
    ev.sort()
    heavy = light = 0
    acc = {"idle": 0, "light only": 0, "1 heavy": 0, "1 heavy + light": 0, ">= 2 heavy": 0}
    prev = lo
    for t, d, h in ev:
        dt = t - prev
        if dt > 0:
            if heavy == 0 and light == 0: acc["idle"] += dt
            elif heavy == 0: acc["light only"] += dt
            elif heavy == 1 and light == 0: acc["1 heavy"] += dt
            elif heavy == 1: acc["1 heavy + light"] += dt
            else: acc[">= 2 heavy"] += dt
        prev = t
        if h: heavy += d
        else: light += d
    tot = t1 - lo
    print(f"# chip occupancy over the last {tail_frac:.0%} of {path} ({tot / 1e6:.1f} ms)")
    for k, v in acc.items():
        print(f"{k:18s} {v / 1e6:9.2f} ms {100.0 * v / tot:6.1f} %")


def light_time(path, tail_frac=0.5, top=25):
    """Where the time WITHOUT an MFMA-heavy kernel in flight goes: per light kernel name the time it ran while no heavy kernel
    did (split evenly among the light kernels in flight), and the idle time charged to the kernel that ended the gap.
    python tools/rocpd_stats.py --light x.db"""
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {namecol}, start, end from kernels order by start").fetchall()
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    lo = t1 - (t1 - t0) * tail_frac
    ev = []
    for i, (n, s, e) in enumerate(rows):
        if e <= lo:
            continue
        ev.append((max(s, lo), 1, i))
        ev.append((e, -1, i))
    ev.sort()
    live = set()
    heavy = 0
    light_by, idle_by = {}, {}
    prev = lo
    for t, d, i in ev:
        dt = t - prev
        if dt > 0 and heavy == 0:
            if live:
                for j in live:
                    k = short(rows[j][0])
                    light_by[k] = light_by.get(k, 0) + dt / len(live)
            elif d > 0:
                k = short(rows[i][0])
                idle_by[k] = idle_by.get(k, 0) + dt
        prev = t
        h = 1 if HEAVY.search(rows[i][0]) else 0
        if d > 0:
            live.add(i)
        else:
            live.discard(i)
        heavy += d * h
    tot = t1 - lo
    print(f"# time without an MFMA-heavy kernel in flight, last {tail_frac:.0%} of {path} ({tot / 1e6:.1f} ms)")
    print(f"# light kernels running alone: {sum(light_by.values()) / 1e6:.2f} ms; idle: {sum(idle_by.values()) / 1e6:.2f} ms")
    print("## light kernels (time while no heavy kernel was in flight)")
    for k, v in sorted(light_by.items(), key=lambda kv: -kv[1])[:top]:
        print(f"{k:70s} {v / 1e6:8.3f} ms")
    print("## idle gaps, by the kernel that ended them")
    for k, v in sorted(idle_by.items(), key=lambda kv: -kv[1])[:top]:
        print(f"{k:70s} {v / 1e6:8.3f} ms")


def gaps(path, tail_frac=0.5, min_us=150.0, top=60):
    """Every moment of the last `tail_frac` of the trace with NO kernel in flight for >= min_us: its length, the kernel that ran
    before it and the one that ended it (host-side stalls show up here).  python tools/rocpd_stats.py --gaps x.db [frac] [min_us]"""
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {namecol}, start, end from kernels order by start").fetchall()
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    lo = t1 - (t1 - t0) * tail_frac
    out = []
    busy_until, last = None, None
    for n, s, e in rows:
        if busy_until is not None and s > busy_until and s >= lo and (s - busy_until) / 1e3 >= min_us:
            out.append(((s - busy_until) / 1e3, (busy_until - lo) / 1e6, short(last), short(n)))
        if busy_until is None or e > busy_until:
            busy_until, last = e, n
    print(f"# idle gaps >= {min_us:.0f} us over the last {tail_frac:.0%} of {path}: {len(out)} gaps, {sum(o[0] for o in out) / 1e3:.2f} ms")
    print(f"{'gap_us':>9s} {'at_ms':>9s}  after -> before")
    for g, at, a, b in sorted(out, key=lambda o: o[1])[:top]:
        print(f"{g:9.1f} {at:9.2f}  {a[:60]} -> {b[:60]}")


def sequence(path, tail_frac=0.2):
    """The kernels of the last `tail_frac` of the trace in start order: start (us from the window's first kernel), duration, idle gap
    since the previous kernel ended, name -- to read a step's serial sections (what runs between the forward and the backward pass)
    launch by launch.  python tools/rocpd_stats.py --sequence x.db [frac]"""
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {namecol}, start, end from kernels order by start").fetchall()
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    lo = t1 - (t1 - t0) * tail_frac
    rows = [r for r in rows if r[1] >= lo]
    base, prev_end = rows[0][1], rows[0][1]
    print(f"# {len(rows)} kernels of the last {tail_frac:.2f} of {path}")
    print(f"{'start_us':>10s} {'dur_us':>8s} {'gap_us':>8s}  kernel")
    for n, s_, e in rows:
        print(f"{(s_ - base) / 1e3:10.1f} {(e - s_) / 1e3:8.1f} {max(0, s_ - prev_end) / 1e3:8.1f}  {short(n)[:70]}")
        prev_end = max(prev_end, e)


if __name__ == "__main__":
    if sys.argv[1] == "--sequence":
        sequence(sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 0.2)
        sys.exit(0)
    if sys.argv[1] == "--gaps":
        gaps(sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 0.5, float(sys.argv[4]) if len(sys.argv) > 4 else 150.0)
        sys.exit(0)
    if sys.argv[1] == "--light":
        light_time(sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 0.5)
    elif sys.argv[1] == "--timeline":
        timeline(sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 0.5)
    else:
        main(sys.argv[1])
