#!/usr/bin/env python
"""Summarise the HBM-side traffic of each kernel from two rocprofv3 PMC passes over bench.py
(FETCH_SIZE in one pass, WRITE_SIZE in another — they do not fit the TCC counter slots together;
MI355X_MICROARCH.md "rocprofv3 PMC slots").  Units and corrections as that guide prescribes:
FETCH_SIZE / WRITE_SIZE are KiB-scaled request counts (x1024 -> bytes); on gfx950 FETCH_SIZE reports HALF the
bytes of wide (16 B/lane) coalesced reads — every hot kernel here reads with 16-B lanes or LDS-DMA dwordx4 — so
fetch bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE is used as is.  Infinity-Cache hits are included in both (the
counters sit on the L2's fabric side).

usage: pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv out.json"""
import collections
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:80]


def load(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write), key=lambda k: -sum(fetch.get(k, [0]))):
        f, w = fetch.get(k, []), write.get(k, [])
        n = max(len(f), len(w))
        if not n:
            continue
        out[k] = {
            "launches": n,
            "fetch_bytes_per_launch": 2.0 * 1024.0 * sum(f) / max(1, len(f)),
            "write_bytes_per_launch": 1024.0 * sum(w) / max(1, len(w)),
            "fetch_size_raw_kib_per_launch": sum(f) / max(1, len(f)),
        }
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    head = open(os.path.join(root, ".td_head")).read().strip() if os.path.exists(os.path.join(root, ".td_head")) else None
    sys.path.insert(0, root)
    from bench import kernel_source_digest
    json.dump({"commit": head, "kernel_source_digest": kernel_source_digest(), "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py --steps 1 --warmup 0 "
                       "--no-graph; fetch = 2 x FETCH_SIZE x 1024 B (gfx950 16-B-lane correction), write = WRITE_SIZE x 1024 B; "
                       "Infinity-Cache hits are counted", "kernels": out}, open(sys.argv[3], "w"), indent=1)
    for k, v in list(out.items())[:14]:
        print(f"{k[:60]:60s} n={v['launches']:5d} fetch {v['fetch_bytes_per_launch']/1e6:9.1f} MB  write {v['write_bytes_per_launch']/1e6:8.1f} MB")


if __name__ == "__main__":
    main()
