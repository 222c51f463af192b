#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files into one small JSON.

usage: pmc_summary.py OUT.json DIR [DIR ...]

For every kernel and counter: number of dispatches and the mean counter value
per dispatch.  FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; the
derived `hbm_bytes_per_launch` applies the gfx950 correction of
MI355X_MICROARCH.md (FETCH_SIZE counts 128-B requests as 64 B => x2).
"""
import csv, glob, json, os, re, sys
from collections import defaultdict

def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)          # drop the argument list
    name = re.sub(r"^void\s+", "", name)
    return name.replace("ggc::", "").strip()

def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    a = acc[short(row["Kernel_Name"])][row["Counter_Name"]]
                    a[0] += 1
                    a[1] += float(row["Counter_Value"])
    res = {}
    for k, cs in sorted(acc.items()):
        e = {"dispatches": max(v[0] for v in cs.values())}
        for c, (n, s) in sorted(cs.items()):
            e[c] = s / n
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_launch"] = int(2 * e["FETCH_SIZE"] * 1024 + e["WRITE_SIZE"] * 1024)
        res[k] = e
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    print("wrote", out, len(res), "kernels")

if __name__ == "__main__":
    main()
