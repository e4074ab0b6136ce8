"""Condense a rocprofv3 `--kernel-trace --stats` kernel_stats.csv into a short table (names cut to 70 chars).
usage: python tools/prof_summary.py <kernel_stats.csv> [out.csv]"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:<>, ]+?)\(", name)
    if m:
        name = m.group(1)
    return name[:70]


rows = list(csv.DictReader(open(sys.argv[1])))
out = [("kernel", "calls", "total_ms", "avg_us", "pct")]
for r in rows:
    out.append((short(r["Name"]), r["Calls"], f'{int(r["TotalDurationNs"]) / 1e6:.3f}', f'{float(r["AverageNs"]) / 1e3:.1f}',
                f'{float(r["Percentage"]):.2f}'))
dst = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
for o in out[:40]:
    dst.write(",".join(str(x) for x in o) + "\n")
