#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per kernel:  python tools/pmc_summary.py <output dir> [...]

Reads every *counter_collection.csv below the given directories (one row per dispatch and counter) and prints, per kernel
(template arguments kept), the number of dispatches and the sum of every counter, plus the derived active-lane fraction
SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU) when both were collected."""
import collections
import csv
import glob
import re
import sys


def short(name):
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def source_hash():
    """digest of the kernel sources the profiled library was compiled from (rt_source_hash of librt_engine.so)"""
    import ctypes
    import os
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracer_amd", "librt_engine.so")
    try:
        L = ctypes.CDLL(lib)
        L.rt_source_hash.restype = ctypes.c_char_p
        return L.rt_source_hash().decode()
    except (OSError, AttributeError):
        return "unknown"


def main():
    print(f"source_hash: {source_hash()}")
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for d in sys.argv[1:]:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
                calls[k].add((f, r["Dispatch_Id"]))
    for k in sorted(tot):
        print(f"{k}  dispatches/pass={len(calls[k])}")
        c = tot[k]
        for name, v in sorted(c.items()):
            print(f"   {name:28s} {v:.6g}")
        if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_THREAD_CYCLES_VALU"):
            print(f"   {'active-lane fraction':28s} {c['SQ_THREAD_CYCLES_VALU'] / (64 * c['SQ_ACTIVE_INST_VALU']):.3f}")


if __name__ == "__main__":
    main()
