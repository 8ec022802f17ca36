#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc SQ passes (CSV) to per-kernel sums and dispatch counts.

usage: sq_reduce.py <out.json> <command text> <pass1.csv> [<pass2.csv> ...]
Every pass is a separate rocprofv3 run of the same command (kernel-trace only), so dispatch counts agree between passes.
"""
import collections
import csv
import json
import sys


def main():
    out, cmd, files = sys.argv[1], sys.argv[2], sys.argv[3:]
    kernels = collections.defaultdict(lambda: {"dispatches": 0, "counters": collections.defaultdict(float)})
    for f in files:
        try:
            rows = list(csv.DictReader(open(f)))
        except OSError as e:
            print(f, e)
            continue
        seen = collections.defaultdict(set)
        for r in rows:
            k = r["Kernel_Name"].split("(")[0]
            if "rp::" not in k:
                continue
            kernels[k]["counters"][r["Counter_Name"]] += float(r["Counter_Value"])
            seen[k].add(r["Dispatch_Id"])
        for k, d in seen.items():
            kernels[k]["dispatches"] = max(kernels[k]["dispatches"], len(d))
    doc = {"note": "rocprofv3 --pmc <one SQ group per run> --kernel-trace of: " + cmd + "; sums over all dispatches of each kernel",
           "kernels": {k: {"dispatches": v["dispatches"], "counters": dict(v["counters"])} for k, v in kernels.items()}}
    json.dump(doc, open(out, "w"), indent=1)
    for k, v in doc["kernels"].items():
        print(k, v["dispatches"], {c: f"{x:.3e}" for c, x in v["counters"].items()})


if __name__ == "__main__":
    main()
