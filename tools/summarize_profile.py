#!/usr/bin/env python3
"""Condenses rocprofv3 output directories (gpurun_out/prof_*) into the small files kept under
profiles/: the kernel-stats CSV of the --kernel-trace --stats run and a JSON with the PMC
counters of the dominant kernel.

  python tools/summarize_profile.py <round-tag> <workload> <trace_dir> [<fetch_dir> <write_dir>]

A frame (one bench step) with >= 64 spp is TWO launches of render_kernel (capi.cpp render_device), so the
figures are per FRAME: total time / counter sum of the kernel's launches divided by the number of frames
(FRAMES_TRACE = 3 for --steps 2 --warmup 1, FRAMES_PMC = 1 for --steps 1 --warmup 0).
"""
import csv
import glob
import json
import os
import sys

KERNEL = "render_kernel<false, false"


def one(pattern):
    f = glob.glob(pattern, recursive=True)
    if not f:
        raise SystemExit(f"nothing matches {pattern}")
    return f[0]


def main():
    tag, workload, trace = sys.argv[1:4]
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(out_dir, exist_ok=True)
    rows = list(csv.DictReader(open(one(os.path.join(trace, "**", "*kernel_stats.csv")))))
    with open(os.path.join(out_dir, f"{tag}_{workload}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in rows:
            r["Name"] = r["Name"][:160]
            w.writerow(r)
    k = [r for r in rows if KERNEL in r["Name"]][0]
    frames_trace = int(os.environ.get("FRAMES_TRACE", "3"))
    frames_pmc = int(os.environ.get("FRAMES_PMC", "1"))
    calls = int(k["Calls"])
    summary = {"workload": workload, "kernel": k["Name"], "calls": calls, "frames": frames_trace,
               "launches_per_frame": calls / frames_trace, "avg_ms": float(k["TotalDurationNs"]) / 1e6 / frames_trace,
               "avg_ms_per_launch": float(k["AverageNs"]) / 1e6, "min_ms": float(k["MinNs"]) / 1e6, "max_ms": float(k["MaxNs"]) / 1e6,
               "share_of_gpu_time_pct": float(k["Percentage"]),
               "source": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 2 --warmup 1; avg_ms = per frame"}
    if len(sys.argv) >= 6:
        for name, d in (("FETCH_SIZE", sys.argv[4]), ("WRITE_SIZE", sys.argv[5])):
            cr = list(csv.DictReader(open(one(os.path.join(d, "**", "*counter_collection.csv")))))
            vals = [float(r["Counter_Value"]) for r in cr if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == name]
            summary[name + "_KB_per_launch"] = sum(vals) / frames_pmc  # per frame (all launches of the frame)
            kr = [r for r in cr if KERNEL in r["Kernel_Name"]][0]
            summary.update({"vgpr": int(kr["VGPR_Count"]), "sgpr": int(kr["SGPR_Count"]), "lds_bytes": int(kr["LDS_Block_Size"]),
                            "grid": int(kr["Grid_Size"]), "workgroup": int(kr["Workgroup_Size"])})
        # MI355X_MICROARCH.md "HBM": FETCH_SIZE = TCC_EA0_RDREQ x 64 B in KB; it reads exactly half of the
        # bytes of a WIDE COALESCED 16 B/lane stream; other access shapes (ours: divergent 16-B loads of 32-B
        # nodes) are uncalibrated, so both the raw and the doubled figure are kept.
        summary["traffic_bytes_raw"] = (summary["FETCH_SIZE_KB_per_launch"] + summary["WRITE_SIZE_KB_per_launch"]) * 1024
        summary["traffic_bytes_fetch_doubled"] = (2 * summary["FETCH_SIZE_KB_per_launch"] + summary["WRITE_SIZE_KB_per_launch"]) * 1024
        summary["pmc_source"] = ("separate passes: rocprofv3 --pmc FETCH_SIZE --kernel-trace / --pmc WRITE_SIZE --kernel-trace "
                                 "-- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-counters")
    with open(os.path.join(out_dir, f"{tag}_{workload}_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
