#!/usr/bin/env python3
"""Workgroup timelines of the count kernels, from the `TL` lines a -DSGC_STAMPS=1 library prints with dbg 1048576.

    SGC_HIPCC_FLAGS=-DSGC_STAMPS=1 python3 -c "from sgcount_amd import build as b; b.build_one(b.SO, force=True)"
    python3 tools/tune.py --variants "4:dbg=1048576" --rounds 1 --steps 4 --notiming --nocheck > tl.txt
    python3 tools/wg_timeline.py tl.txt

Per kernel (the last pass of the steps: the buffers are overwritten pass by pass; the LAST dump in the log): the span from the first workgroup's start to the last one's end, how busy the
workgroup slots were over that span (sum of lifetimes / (slots x span) with slots = the largest number alive at once), the spread
of starts and ends, and the mean lifetime by XCD — the per-XCD means tell a placement effect (some XCDs see slower memory) from a
work imbalance (lifetimes follow the `extra` column: blocks / chunks of the workgroup).
Times are ticks of the constant 100 MHz clock (s_memrealtime): 10 ns.
"""
import collections
import re
import statistics
import sys

PAT = re.compile(r"^TL (\S+) wg (\d+) xcc (\d+) hw ([0-9a-f]+) begin (\d+) end (\d+) extra (\d+)(?: cycles (\d+))?(?: ph (\d+) (\d+) (\d+) (\d+))?")


def main():
    rows = collections.defaultdict(list)
    for line in open(sys.argv[1]):
        m = PAT.match(line)
        if m:
            rows[m.group(1)].append((int(m.group(5)), int(m.group(6)), int(m.group(2)), int(m.group(3)), int(m.group(4), 16), int(m.group(7)), int(m.group(8) or 0), tuple(int(m.group(k) or 0) for k in (9, 10, 11, 12))))
    for name, rs in rows.items():
        rs.sort()
        # launches of one kernel are >= 0.25 ms apart (the other kernels of the pass run between them)
        cut = 0
        for i in range(1, len(rs)):
            if rs[i][0] - rs[i - 1][0] > 25000:
                cut = i
        rs = rs[cut:]
        t0 = min(r[0] for r in rs)
        t1 = max(r[1] for r in rs)
        span = t1 - t0
        ev = sorted([(r[0], 1) for r in rs] + [(r[1], -1) for r in rs])
        alive = peak = 0
        for _, d in ev:
            alive += d
            peak = max(peak, alive)
        life = [r[1] - r[0] for r in rs]
        print("%s: %d workgroups, span %.1f us, lifetimes min %.1f / median %.1f / max %.1f us, peak alive %d, slot use %.3f" % (
            name, len(rs), span / 100, min(life) / 100, statistics.median(life) / 100, max(life) / 100, peak, sum(life) / (peak * span)))
        starts = sorted(r[0] - t0 for r in rs)
        ends = sorted(t1 - r[1] for r in rs)
        q = lambda v, f: v[min(len(v) - 1, int(f * len(v)))] / 100
        print("   starts after the first: p50 %.1f p90 %.1f max %.1f us; ends before the last: p50 %.1f p90 %.1f max %.1f us" % (
            q(starts, .5), q(starts, .9), starts[-1] / 100, q(ends, .5), q(ends, .9), ends[-1] / 100))
        by = collections.defaultdict(list)
        for r in rs:
            by[r[3]].append(r)
        print("   by XCD: " + "  ".join("%d: n %d life %.1f end %.1f" % (x, len(v), statistics.mean(a[1] - a[0] for a in v) / 100,
                                                                         statistics.mean(a[1] - t0 for a in v) / 100) for x, v in sorted(by.items())))
        if any(r[6] for r in rs):
            mhz = [r[6] / ((r[1] - r[0]) / 100) for r in rs if r[1] - r[0] > 1000]
            print("   s_memtime ticks per us of lifetime: min %.0f median %.0f max %.0f" % (min(mhz), statistics.median(mhz), max(mhz)))
        if any(any(r[7]) for r in rs):
            tick = statistics.median(mhz) if any(r[6] for r in rs) else 2300.0
            print("   phases (s_memtime ticks -> us), mean over workgroups: " + "  ".join("%.1f" % (statistics.mean(r[7][k] for r in rs) / tick) for k in range(4)))
        ex = [r[5] for r in rs]
        if len(set(ex)) > 1:
            mx = statistics.mean(ex)
            ml = statistics.mean(life)
            cov = sum((a - mx) * (b - ml) for a, b in zip(ex, life))
            va, vb = sum((a - mx) ** 2 for a in ex), sum((b - ml) ** 2 for b in life)
            print("   extra (work units): min %d max %d, correlation with lifetime %.2f" % (min(ex), max(ex), cov / (va * vb) ** .5 if va and vb else 0))
        cus = collections.defaultdict(list)
        for r in rs:
            cus[(r[3], (r[4] >> 8) & 0xF, (r[4] >> 13) & 0x7, (r[4] >> 12) & 1)].append(r)
        print("   distinct (xcd, cu, se, sh): %d; workgroups per CU min %d max %d" % (len(cus), min(map(len, cus.values())), max(map(len, cus.values()))))


if __name__ == "__main__":
    main()
