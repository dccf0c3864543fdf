#!/usr/bin/env python3
"""Condenses a rocprofv3 ``*_kernel_stats.csv`` into a short table: kernel names
cut to a readable length (template arguments of library kernels run to several
kB), instances merged per shortened name, sorted by total time."""
import collections
import csv
import re
import sys


def short(name, width=96):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"_ZN2ck.*?(kernel_[a-z_0-9]+).*?Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", name)
    if m:
        return f"ck::{m.group(1)}<{m.group(2)},{m.group(3)},{m.group(4)},{m.group(5)},...>"
    return name if len(name) <= width else name[: width - 3] + "..."


def main(src, dst, top=40):
    rows = list(csv.DictReader(open(src)))
    agg = collections.OrderedDict()
    for r in rows:
        key = short(r["Name"])
        a = agg.setdefault(key, [0, 0.0])
        a[0] += int(r["Calls"])
        a[1] += float(r["TotalDurationNs"])
    total = sum(a[1] for a in agg.values()) or 1.0
    ordered = sorted(agg.items(), key=lambda kv: -kv[1][1])
    with open(dst, "w") as out:
        out.write("kernel,calls,total_ms,avg_us,percent\n")
        for name, (calls, ns) in ordered[:top]:
            out.write(f"\"{name}\",{calls},{ns / 1e6:.3f},{ns / calls / 1e3:.2f},{100 * ns / total:.2f}\n")
        rest = ordered[top:]
        if rest:
            ns = sum(a[1] for _, a in rest)
            calls = sum(a[0] for _, a in rest)
            out.write(f"\"({len(rest)} more kernels)\",{calls},{ns / 1e6:.3f},,{100 * ns / total:.2f}\n")
        out.write(f"\"TOTAL GPU kernel time\",,{total / 1e6:.3f},,100.00\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 40)
