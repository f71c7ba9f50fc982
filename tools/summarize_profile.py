#!/usr/bin/env python3
"""Condenses what tools/measure_round.sh left under gpurun_out/ into the small files kept under profiles/.

  python tools/summarize_profile.py <round-tag> <workload> [<workload> ...]

Per workload: <tag>_<wl>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of bench.py), <tag>_<wl>_pmc_counters.txt
(the PMC passes of the low-spp probe), <tag>_<wl>_summary.json = pmc_<wl>.json (kernel time per frame, FETCH_SIZE /
WRITE_SIZE per frame from the two dedicated passes, and the VALU-issue model of tools/valu_model.py: what bench.py's
`roofline` reads), and <tag>_<wl>_bench.json for every bench line found.
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "render_kernel<false, false"
FRAMES_TRACE, FRAMES_PMC = 3, 1  # --steps 2 --warmup 1 / --steps 1 --warmup 0


def one(pattern):
    f = glob.glob(pattern, recursive=True)
    return max(f, key=os.path.getmtime) if f else None  # (a re-run of one workload leaves the older files beside the new ones)


def main():
    tag = sys.argv[1]
    G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
    os.makedirs(P, exist_ok=True)
    for wl in sys.argv[2:]:
        summary = {"workload": wl, "round": tag}
        stats = one(os.path.join(G, f"prof_{tag}_{wl}_trace", "**", "*kernel_stats.csv"))
        if stats:
            rows = list(csv.DictReader(open(stats)))
            with open(os.path.join(P, f"{tag}_{wl}_kernel_stats.csv"), "w", newline="") as f:
                w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
                w.writeheader()
                for r in rows:
                    r["Name"] = r["Name"][:160]
                    w.writerow(r)
            k = [r for r in rows if KERNEL in r["Name"]][0]
            calls = int(k["Calls"])
            summary.update({"kernel": k["Name"], "calls": calls, "frames": FRAMES_TRACE, "launches_per_frame": calls / FRAMES_TRACE,
                            "avg_ms": float(k["TotalDurationNs"]) / 1e6 / FRAMES_TRACE, "min_ms": float(k["MinNs"]) / 1e6, "max_ms": float(k["MaxNs"]) / 1e6,
                            "share_of_gpu_time_pct": float(k["Percentage"]),
                            "source": f"rocprofv3 --kernel-trace --stats -- python3 bench.py --workload {wl} --steps 2 --warmup 1 --no-cpu-baseline; avg_ms = per frame"})
        for name, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
            f = one(os.path.join(G, f"prof_{tag}_{wl}_{sub}", "**", "*counter_collection.csv"))
            if not f:
                continue
            cr = list(csv.DictReader(open(f)))
            vals = [float(r["Counter_Value"]) for r in cr if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == name]
            summary[name + "_KB_per_launch"] = sum(vals) / FRAMES_PMC
            kr = [r for r in cr if KERNEL in r["Kernel_Name"]][0]
            summary.update({"vgpr": int(kr["VGPR_Count"]), "sgpr": int(kr["SGPR_Count"]), "lds_bytes": int(kr["LDS_Block_Size"]),
                            "grid": int(kr["Grid_Size"]), "workgroup": int(kr["Workgroup_Size"])})
        if "FETCH_SIZE_KB_per_launch" in summary and "WRITE_SIZE_KB_per_launch" in summary:
            # MI355X_MICROARCH.md "HBM": FETCH_SIZE = TCC_EA0_RDREQ x 64 B in KB, L2-miss requests including Infinity-Cache hits; it reads
            # exactly half of the bytes of a WIDE COALESCED 16 B/lane stream; other access shapes (ours: divergent 16-B loads) are
            # uncalibrated, so both the raw and the doubled figure are kept.
            summary["traffic_bytes_raw"] = (summary["FETCH_SIZE_KB_per_launch"] + summary["WRITE_SIZE_KB_per_launch"]) * 1024
            summary["traffic_bytes_fetch_doubled"] = (2 * summary["FETCH_SIZE_KB_per_launch"] + summary["WRITE_SIZE_KB_per_launch"]) * 1024
            summary["pmc_source"] = ("separate passes: rocprofv3 --pmc FETCH_SIZE --kernel-trace / --pmc WRITE_SIZE --kernel-trace -- python3 bench.py "
                                     f"--workload {wl} --steps 1 --warmup 0 --no-cpu-baseline --no-counters")
        pmc = os.path.join(G, f"pmc_{tag}_{wl}", "summary.txt")
        if os.path.exists(pmc):
            shutil.copy(pmc, os.path.join(P, f"{tag}_{wl}_pmc_counters.txt"))
            model = json.loads(subprocess.run([sys.executable, os.path.join(ROOT, "tools", "valu_model.py"), pmc], capture_output=True, text=True, check=True).stdout)
            summary.update(model)
            summary["valu_source"] = f"tools/pmc_passes.sh on tools/pmc_probe.py {wl} (low-spp frame) -> profiles/{tag}_{wl}_pmc_counters.txt -> tools/valu_model.py"
        # the library the counters were taken on (bench.py withholds figures priced with another build's profile)
        for line_file in (os.path.join(G, f"prof_{tag}_{wl}_trace.json"), os.path.join(G, f"bench_{wl}_{tag}.json")):
            try:
                cfg = json.load(open(line_file))["config"]
                summary["build_id"] = cfg["library_build_id"]
                # the tree the per-ray counters belong to (tools/measure_round.sh: PROBE_BUILDER = bench.py's default builder)
                summary["builder"] = cfg.get("accelerator", {}).get("builder", "gpu")
                break
            except (OSError, ValueError, KeyError):
                continue
        # ... and, what the validity of the profile is decided by: the machine code of the kernel the counters belong to
        # (pbrt_amd/isa_id.py; computed from the library beside this script -- on the GPU box the one that was measured)
        try:
            sys.path.insert(0, ROOT)
            from pbrt_amd import build, isa_id
            if summary.get("kernel"):
                summary["kernel_isa_id"] = isa_id.kernel_id(build.LIB_PATH, summary["kernel"])
                summary["compiler"] = build.compiler_version().splitlines()[0]
        except Exception as e:  # noqa: BLE001
            summary["kernel_isa_id_error"] = repr(e)[:200]
        with open(os.path.join(P, f"{tag}_{wl}_summary.json"), "w") as f:
            json.dump(summary, f, indent=1)
        shutil.copy(os.path.join(P, f"{tag}_{wl}_summary.json"), os.path.join(P, f"pmc_{wl}.json"))
        print(wl, {k: summary.get(k) for k in ("avg_ms", "traffic_bytes_raw", "valu_busy_frac_at_profile_clock", "lane_utilisation", "valu_issue_cycles_per_ray")})
    for f in glob.glob(os.path.join(G, f"bench_*_{tag}.json")):
        wl = os.path.basename(f)[len("bench_"):-len(f"_{tag}.json")]  # "c3", "c3_gpubuilder", ...
        if os.path.getsize(f):
            shutil.copy(f, os.path.join(P, f"{tag}_{wl}_bench.json"))


if __name__ == "__main__":
    main()
