#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (separate runs, as MI355X_MICROARCH.md prescribes)
into the per-launch HBM-side traffic of the dominant kernel.  Corrections of the guide for gfx950: counters are in KB;
FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (doubled here); WRITE_SIZE is exact for
16-B-per-lane streaming stores (taken as is).  usage: tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> [kernel-substring]"""
import csv
import glob
import json
import sys


def mean_counter(d, name, kernel):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"] and r["Counter_Name"] == name]
    return sum(vals) / len(vals), len(vals)


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    kernel = sys.argv[4] if len(sys.argv) > 4 else "qp_solve_kernel"
    fs, nf = mean_counter(fetch_dir, "FETCH_SIZE", kernel)
    wsz, nw = mean_counter(write_dir, "WRITE_SIZE", kernel)
    res = {"kernel": kernel, "launches_averaged": [nf, nw], "FETCH_SIZE_KB_raw": fs, "WRITE_SIZE_KB_raw": wsz,
           "fetch_bytes_corrected": 2.0 * fs * 1024.0, "write_bytes": wsz * 1024.0,
           "traffic_bytes_per_launch": 2.0 * fs * 1024.0 + wsz * 1024.0,
           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); counters are fabric-side "
                   "(Infinity-Cache hits appear to be counted), so this is L2-miss traffic, an upper bound on HBM bytes"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
