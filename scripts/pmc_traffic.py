#!/usr/bin/env python3
"""Reduce two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; CSV output) to HBM bytes per launch per kernel.

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> <note> <batch> <update>

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): the counters are in KiB, and
gfx950's FETCH_SIZE reports half of the bytes of wide coalesced reads, so reads are doubled.
"""
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path, counter):
    tot, n = defaultdict(float), defaultdict(set)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            tot[name] += float(row["Counter_Value"])
            n[name].add(row["Dispatch_Id"])
    return {k: tot[k] / max(len(n[k]), 1) for k in tot}


def main():
    fetch_csv, write_csv, out, note = sys.argv[1:5]
    batch, update = int(sys.argv[5]), sys.argv[6]
    fetch = per_kernel(fetch_csv, "FETCH_SIZE")
    write = per_kernel(write_csv, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        if "rp::" not in k:
            continue
        fk, wk = fetch.get(k, 0.0), write.get(k, 0.0)
        kernels[k] = {"fetch_kb_raw": fk, "write_kb_raw": wk, "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024.0}
    json.dump({"note": note, "batch": batch, "update": update, "kernels": kernels}, open(out, "w"), indent=1)
    for k, v in kernels.items():
        print(f"{v['hbm_bytes_per_launch'] / 1e6:12.3f} MB/launch  {k}")


if __name__ == "__main__":
    main()
