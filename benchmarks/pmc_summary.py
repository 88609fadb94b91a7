#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output: per kernel (name filter) mean counter values per dispatch.

    python benchmarks/pmc_summary.py <dir> [substring-of-kernel-name] [--skip N]
Walks <dir> for *counter_collection.csv files (one rocprofv3 pass each).
"""
import csv
import json
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ""
    skip = int(sys.argv[sys.argv.index("--skip") + 1]) if "--skip" in sys.argv else 0
    acc = defaultdict(lambda: defaultdict(list))
    for dp, _dn, fn in os.walk(root):
        for f in fn:
            if not f.endswith("counter_collection.csv"):
                continue
            per_disp = defaultdict(dict)
            names = {}
            with open(os.path.join(dp, f)) as fh:
                for row in csv.DictReader(fh):
                    kn = row["Kernel_Name"]
                    if filt and filt not in kn:
                        continue
                    did = int(row["Dispatch_Id"])
                    names[did] = kn
                    c = row["Counter_Name"]
                    per_disp[did][c] = per_disp[did].get(c, 0.0) + float(row["Counter_Value"])
            seen = defaultdict(int)
            for did in sorted(per_disp):
                kn = names[did]
                seen[kn] += 1
                if seen[kn] <= skip:
                    continue
                for c, v in per_disp[did].items():
                    acc[kn][c].append(v)
    out = {}
    for kn, cs in acc.items():
        short = kn[:110]
        out[short] = {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in sorted(cs.items())}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
