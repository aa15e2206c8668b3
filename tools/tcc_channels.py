#!/usr/bin/env python3
"""Per-channel view of the L2 <-> memory traffic of the count kernels, from rocprofv3 JSON output (round 4: what differs
between a process in the fast and one in the slow placement regime).

    rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ TCC_EA0_RDREQ TCC_EA0_WRREQ_STALL TCC_EA0_RDREQ_LEVEL --output-format json -d DIR -- python3 tools/tune.py --variants 4 --rounds 1 --steps 2
    python3 tools/tcc_channels.py DIR [DIR ...]

The TCC counters have one instance per L2 channel: 16 per XCD x 8 XCDs.  For every traced kernel of the pass (last dispatch of each) prints
its duration and, per counter, the sum and the spread over the 128 channels (min / mean / max, max/mean, coefficient of variation), the
per-channel read latency RDREQ_LEVEL / RDREQ where both were collected, and the same folded by channel index over the XCDs (channel i of
every XCD serves the same slice of the address space)."""
import glob
import json
import os
import statistics
import sys

KERNELS = ("k_partition", "k_count_slices", "k_core<false", "k_core<true")


def load(d):
    f = sorted(glob.glob(os.path.join(d, "**", "*_results.json"), recursive=True))
    if not f:
        raise SystemExit("no *_results.json under " + d)
    r = json.load(open(f[-1]))["rocprofiler-sdk-tool"][0]
    names = {c["id"]["handle"]: c["name"] for c in r["counters"]}
    inst = {}
    for c in r["counters"]:
        inst[c["id"]["handle"]] = len(c.get("instances", [])) or 1
    ksym = {k["kernel_id"]: k.get("formatted_kernel_name") or k.get("demangled_kernel_name") or k.get("kernel_name") for k in r["kernel_symbols"]}
    out = {}
    for rec in r["callback_records"]["counter_collection"]:
        di = rec["dispatch_data"]
        name = ksym.get(di["dispatch_info"]["kernel_id"], "?")
        name = name[5:] if name.startswith("void ") else name
        key = next((k for k in KERNELS if name.startswith(k)), None)
        if key is None:
            continue
        vals = {}
        for x in rec["records"]:
            vals.setdefault(names[x["counter_id"]["handle"]], []).append(x["value"])
        out[key] = {"name": name, "ns": di["end_timestamp"] - di["start_timestamp"], "counters": vals}      # the last dispatch of the kernel wins
    return out


def spread(v):
    m = statistics.fmean(v)
    return "sum %.4g  min %.4g  mean %.4g  max %.4g  max/mean %.3f  cv %.3f" % (sum(v), min(v), m, max(v), max(v) / m if m else 0, statistics.pstdev(v) / m if m else 0)


def main():
    for d in sys.argv[1:]:
        ks = load(d)
        print("==", d)
        for key in KERNELS:
            if key not in ks:
                continue
            k = ks[key]
            print("  %-16s %.1f us" % (key, k["ns"] / 1e3))
            for cname, v in sorted(k["counters"].items()):
                print("     %-28s n=%d  %s" % (cname, len(v), spread(v)))
                if len(v) == 128:
                    fold = [sum(v[x * 16 + i] for x in range(8)) for i in range(16)]        # instance index fastest (as listed in "instances")
                    print("     %-28s by channel over the XCDs: %s" % ("", " ".join("%.3g" % f for f in fold)))
                    perx = [sum(v[x * 16 + i] for i in range(16)) for x in range(8)]
                    print("     %-28s by XCD: %s" % ("", " ".join("%.3g" % f for f in perx)))
            c = k["counters"]
            if "TCC_EA0_RDREQ_LEVEL" in c and "TCC_EA0_RDREQ" in c and len(c["TCC_EA0_RDREQ"]) == len(c["TCC_EA0_RDREQ_LEVEL"]):
                lat = [a / b for a, b in zip(c["TCC_EA0_RDREQ_LEVEL"], c["TCC_EA0_RDREQ"]) if b]
                print("     %-28s %s" % ("read latency (LEVEL/RDREQ)", spread(lat)))
            if "TCC_EA0_WRREQ_LEVEL" in c and "TCC_EA0_WRREQ" in c and len(c["TCC_EA0_WRREQ"]) == len(c["TCC_EA0_WRREQ_LEVEL"]):
                lat = [a / b for a, b in zip(c["TCC_EA0_WRREQ_LEVEL"], c["TCC_EA0_WRREQ"]) if b]
                print("     %-28s %s" % ("write latency (LEVEL/WRREQ)", spread(lat)))


if __name__ == "__main__":
    main()
