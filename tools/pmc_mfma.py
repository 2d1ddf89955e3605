#!/usr/bin/env python3
"""Matrix-core busy share of the dominant kernel from a rocprofv3 pass with
  --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE  (+ optionally SQ_WAIT_ANY ... in a second pass)
mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 256 CUs x 4 SIMDs): MFMA-busy cycles summed over all SIMDs over the
SIMD-cycles the dispatch lasted (GRBM_GUI_ACTIVE is the sum over the 8 XCDs, MI355X_MICROARCH.md; this is the gfx94x MfmaUtil formula).
usage: tools/pmc_mfma.py <pmc_dir> <out.json> [kernel-substring] [wait_pmc_dir]"""
import csv
import glob
import json
import sys


def counters(d, kernel):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    acc = {}
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    d, out = sys.argv[1:3]
    kernel = sys.argv[3] if len(sys.argv) > 3 else "qp_solve_kernel"
    c, n = counters(d, kernel)
    simd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4
    res = {"kernel": kernel, "launches_averaged": n.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), "counters_mean_per_launch": c,
           "mfma_busy_frac": c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles,
           "mfma_f64_ops_per_launch": c.get("SQ_INSTS_VALU_MFMA_MOPS_F64"),
           "formula": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)"}
    if len(sys.argv) > 4:
        w, _ = counters(sys.argv[4], kernel)
        res["wait_counters_mean_per_launch"] = w
        if "SQ_WAVE_CYCLES" in w and w["SQ_WAVE_CYCLES"] > 0:
            res["wave_cycle_shares"] = {k: w[k] / w["SQ_WAVE_CYCLES"] for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if k in w}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: res[k] for k in ("kernel", "mfma_busy_frac", "launches_averaged")}))


if __name__ == "__main__":
    main()
