#!/usr/bin/env python3
"""
HBM traffic per launch of the stacking kernels from stored PMC passes (FETCH_SIZE and WRITE_SIZE are
collected in separate rocprofv3 runs, tools/prof_counters.sh), as the JSON file bench.py reads for
`roofline.traffic`.

usage: python tools/pmc_traffic.py <label>=<fetch.csv>,<write.csv> ... > profiles/rNN_pmc_traffic.json
       label = "<config>:<what>", e.g. C3:detect  C3L:volume

Units / calibration (MI355X_MICROARCH.md, "HBM"): both counters are in KB; they derive from the L2's
memory-side request counters.  On gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced
reads at 64 bytes, so the bytes actually moved lie between 1x and 2x the counter; WRITE_SIZE matched
a known byte count to 2-3 % in this code (the locate volume: 13.45 GB counted for 13.09 GB stored).
Both raw values are kept; `fetch_bytes_upper` is the doubled one.
"""
import csv
import hashlib
import json
import os
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "quakemigrate_amd", "csrc")
KERNEL_SOURCES = ("qm_kernels.hpp", "qm_pair.hpp", "qm_shift.hpp", "qm_shift_asm.inc")


def kernel_code_digest():
    """what the stored byte counts were measured on: the stacking kernels' sources (bench.py reports a
    stored figure only while this still matches the tree it runs from)"""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        h.update(open(os.path.join(CSRC, name), "rb").read())
    return h.hexdigest()[:16]


def per_launch(path, counter):
    total, launches, name = 0.0, 0, None
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if r["Counter_Name"] != counter or ("stack_" not in k and "screen_lds" not in k):
            continue
        if float(r["Counter_Value"]) <= 0 and "stack_" in k:
            continue
        total += float(r["Counter_Value"])
        launches += 1
        name = k
    return (total * 1024.0 / max(launches, 1), launches, name)


def main():
    out = {"_about": __doc__.strip().split("\n\n")[2], "_kernel_code": kernel_code_digest()}
    for spec in sys.argv[1:]:
        label, files = spec.split("=")
        fetch, write = files.split(",")
        fb, fl, name = per_launch(fetch, "FETCH_SIZE")
        wb, wl, name2 = per_launch(write, "WRITE_SIZE")
        out[label] = {"kernel": (name or name2 or "").split("(")[0], "fetch_bytes": fb,
                      "fetch_bytes_upper": 2.0 * fb, "write_bytes": wb,
                      "launches_profiled": [fl, wl], "source": [fetch.split("/")[-1], write.split("/")[-1]]}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
