#!/usr/bin/env python3
"""The VALU-issue roof of render_kernel from a PMC summary (tools/pmc_passes.sh -> summary.txt).

gfx950 issues vector instructions in two classes (tools/ubench/valu_issue.hip, profiles/r02_valu_issue_ubench.txt; cycles per
wave64 instruction per SIMD with >= 2 waves resident): FAST 2.33 (fma / mul / add / sub f32, add / sub / and / or / xor /
right shifts, mov), SLOW 4.2 (min / max / min3 / max3, every cvt, every cmp, cndmask, packed f32, left shift, 3-operand
integer forms, mul_lo), transcendental 8.1, 64-bit integer mad 5.1.  The SQ's instruction-class counters tell the classes
apart only partly (calibrated on the ubench, profiles/r02_pmc_class_calibration.txt): ADD_F32 and MUL_F32 are FAST; CVT is
SLOW; FMA_F32 counts v_fma_f32 (FAST) and v_pk_fma_f32 (SLOW) alike; INT32 mixes both; min / max / cmp / cndmask / mov /
logic are not counted at all.  The unresolved groups are split by a STATIC census of the kernel in two parts
(tools/isa_stats.py): one node step, whose dynamic count follows from the CVT counter (24 v_cvt_f32_ubyte per step), and
the rest of the kernel (leaf pass, service stage: far more v_mov and logic, which are FAST); per part, the share of SLOW
opcodes among the instructions of each group.  Result: issue cycles per instruction, per ray, VALU-busy fraction, and the bracket
[every unresolved instruction FAST, every one SLOW].

usage: tools/valu_model.py <summary.txt> [n_simds=1024]  ->  JSON on stdout
"""
import json
import os
import re
import subprocess
import sys

FAST, SLOW, TRANS, INT64 = 2.33, 4.2, 8.1, 5.1
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# which opcodes increment which class counter (profiles/r02_pmc_class_calibration.txt)
COUNTED = {
    "FMA_F32": {"v_fma_f32", "v_fmac_f32", "v_pk_fma_f32"},
    "INT32": {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_cmp_eq_u32", "v_cmp_ne_u32", "v_cmp_lt_u32", "v_cmp_gt_u32", "v_cmp_le_u32", "v_cmp_ge_u32",
              "v_cmp_lt_i32", "v_cmp_gt_i32", "v_cmp_le_i32", "v_cmp_ge_i32", "v_cmp_eq_i32", "v_cmp_ne_i32",
              "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_mul_u32_u24", "v_lshl_add_u32", "v_add3_u32", "v_bfe_u32", "v_add_co_u32",
              "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_min_u32", "v_max_u32", "v_add_lshl_u32", "v_mad_i32_i24"},
    "ADD_F32": {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_pk_add_f32"},
    "MUL_F32": {"v_mul_f32", "v_pk_mul_f32"},
    "CVT": None, "TRANS_F32": None, "INT64": None,
}


def _group_of(op):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    for g in ("FMA_F32", "INT32", "ADD_F32", "MUL_F32"):
        if base in COUNTED[g]:
            return g
    if base.startswith("v_cvt"):
        return "CVT"
    if base.startswith(("v_rcp", "v_sqrt", "v_rsq", "v_exp", "v_log", "v_sin", "v_cos")):
        return "TRANS_F32"
    if base.startswith("v_mad_u64") or base.startswith("v_lshl_add_u64") or base.startswith("v_lshlrev_b64"):
        return "INT64"
    return "OTHER"


def static_census():
    """Static census of the production kernel in two parts: one unrolled node step, and the rest of the kernel (leaf pass,
    service stage, scheduling).  Per part and per counter group: instructions and the share of SLOW opcodes among them."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_stats

    def hist(*flags):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_stats.py"), "--raw", *flags], capture_output=True, text=True, check=True).stdout
        return {o: c for o, c in json.loads(out.strip().split("\n")[-1]).items() if o.startswith("v_")}
    step, kernel = hist("--step"), hist()
    n_cvt_step = sum(c for o, c in step.items() if o.startswith("v_cvt_f32_ubyte"))
    n_cvt_kernel = sum(c for o, c in kernel.items() if o.startswith("v_cvt_f32_ubyte"))
    unroll = max(1, round(n_cvt_kernel / max(1, n_cvt_step)))
    rest = {o: max(0, c - unroll * step.get(o, 0)) for o, c in kernel.items()}

    def part(h):
        res = {}
        for g in ("FMA_F32", "INT32", "OTHER", "CVT"):
            n = {"fast": 0, "slow": 0, "trans": 0}
            for op, c in h.items():
                if _group_of(op) == g:
                    n[isa_stats.klass(op)] += c
            tot = n["fast"] + n["slow"] + n["trans"]
            res[g] = {"n": tot, "slow_share": (n["slow"] / tot) if tot else 0.5}
        res["valu"] = sum(h.values())
        return res
    return {"step": part(step), "rest": part(rest), "unroll": unroll}


def main():
    c = {}
    for line in open(sys.argv[1]):
        k, v = line.split()
        c[k] = float(v)
    n_simd = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    total = c["SQ_INSTS_VALU"]
    grp = {k: c.get("SQ_INSTS_VALU_" + k, 0.0) for k in ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "CVT", "INT32", "INT64")}
    other = total - sum(grp.values())
    census = static_census()
    known = grp["ADD_F32"] * FAST + grp["MUL_F32"] * FAST + grp["CVT"] * SLOW + grp["TRANS_F32"] * TRANS + grp["INT64"] * INT64
    # how many node steps ran: every step converts its 24 plane bytes (v_cvt_f32_ubyte), nearly all the CVT count of the kernel
    step_passes = grp["CVT"] / max(1, census["step"]["CVT"]["n"])
    dyn = {"FMA_F32": grp["FMA_F32"], "INT32": grp["INT32"], "OTHER": other}

    def cycles(pick):
        tot = known
        for g, n_dyn in dyn.items():
            n_step = min(n_dyn, step_passes * census["step"][g]["n"])  # the part of the group executed inside node steps
            for n, where in ((n_step, "step"), (n_dyn - n_step, "rest")):
                sh = pick(census[where][g]["slow_share"])
                tot += n * (sh * SLOW + (1 - sh) * FAST)
        return tot

    best = cycles(lambda sh: sh)
    lo, hi = cycles(lambda sh: 0.0), cycles(lambda sh: 1.0)
    share = {"step": {g: census["step"][g]["slow_share"] for g in dyn}, "rest": {g: census["rest"][g]["slow_share"] for g in dyn},
             "node_step_passes": step_passes, "valu_per_step": census["step"]["valu"],
             "share_of_instructions_in_node_steps": step_passes * census["step"]["valu"] / total}
    kernel_cycles = c["GRBM_GUI_ACTIVE"] / 8.0  # the counter sums the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back)
    simd_cycles = kernel_cycles * n_simd
    rays = c.get("PROBE_RAYS", 0.0)
    out = {
        "valu_instructions": total, "class_counters": grp, "uncounted_instructions": other, "static_census": share,
        "valu_issue_cycles": best, "mean_issue_cycles_per_instruction": best / total,
        "kernel_cycles": kernel_cycles, "clock_ghz_in_profile": kernel_cycles / c["PROBE_KERNEL_NS"] if c.get("PROBE_KERNEL_NS") else None,
        "valu_busy_frac_at_profile_clock": best / simd_cycles, "valu_busy_bracket": [lo / simd_cycles, hi / simd_cycles],
        "lane_utilisation": c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]) if c.get("SQ_ACTIVE_INST_VALU") else None,
        "l2_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]) if c.get("TCC_HIT_sum") else None,
        "wave_wait_frac": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None,
        # where a resident wave's time goes: waiting for memory (s_waitcnt), waiting for its turn to issue, issuing
        "wave_issue_wait_frac": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") and c.get("SQ_WAIT_INST_ANY") else None,
        "wave_issuing_frac": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") and c.get("SQ_ACTIVE_INST_ANY") else None,
        # the CU's vector L1 (profiles/r03end_c3_l1_path_counters.txt): cache accesses per CU and clock (a lane's 16-byte gather is one
        # access; the L1 handles one per clock), and how often the address unit waits for it
        "l1_accesses_per_cu_clock": c["TCP_TOTAL_CACHE_ACCESSES_sum"] / (kernel_cycles * n_simd / 4.0) if c.get("TCP_TOTAL_CACHE_ACCESSES_sum") else None,
        "l1_accesses_per_ray": c["TCP_TOTAL_CACHE_ACCESSES_sum"] / rays if c.get("TCP_TOTAL_CACHE_ACCESSES_sum") and rays else None,
        "ta_stalled_by_l1_frac": c["TA_ADDR_STALLED_BY_TC_CYCLES_sum"] / (kernel_cycles * n_simd / 4.0) if c.get("TA_ADDR_STALLED_BY_TC_CYCLES_sum") else None,
        "l1_tag_conflict_stall_frac": c["TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"] / (kernel_cycles * n_simd / 4.0) if c.get("TCP_READ_TAGCONFLICT_STALL_CYCLES_sum") else None,
        "salu_per_valu": c.get("SQ_INSTS_SALU", 0.0) / total,
        "probe_rays": rays, "probe_samples": c.get("PROBE_SAMPLES"),
        "valu_instructions_per_ray": total / rays if rays else None, "valu_issue_cycles_per_ray": best / rays if rays else None,
        "fetch_bytes_per_ray": c.get("FETCH_SIZE", 0.0) * 1024 / rays if rays else None,
        "write_bytes_per_ray": c.get("WRITE_SIZE", 0.0) * 1024 / rays if rays else None,
        "probe_counter_gbps": (c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024 / c["PROBE_KERNEL_NS"] if c.get("PROBE_KERNEL_NS") else None,
        "costs": {"fast": FAST, "slow": SLOW, "trans": TRANS, "int64": INT64, "source": "profiles/r02_valu_issue_ubench.txt"},
    }
    # ---- MEASURED occupancy of the vector issue port (r03): quad-cycles in which a vector instruction was issued =
    # SQ_ACTIVE_INST_VALU - SQ_ACTIVE_INST_VALU2 (a quad-cycle takes one instruction, or two of the full-rate class:
    # profiles/r03c_issue_counter_calibration.txt), over the SIMDs' quad-cycles of the same pass.  No census, no prices.
    if c.get("ISSUE_SQ_ACTIVE_INST_VALU") and c.get("ISSUE_GRBM_GUI_ACTIVE"):
        busy_qc = c["ISSUE_SQ_ACTIVE_INST_VALU"] - c.get("ISSUE_SQ_ACTIVE_INST_VALU2", 0.0)
        total_qc = c["ISSUE_GRBM_GUI_ACTIVE"] / 8.0 / 4.0 * n_simd
        out["valu_issue_busy_measured"] = busy_qc / total_qc
        out["valu_issue_quadcycles_per_ray"] = busy_qc / rays if rays else None
        out["valu_dual_issue_share_of_instructions"] = 2.0 * c.get("ISSUE_SQ_ACTIVE_INST_VALU2", 0.0) / c["ISSUE_SQ_INSTS_VALU"] if c.get("ISSUE_SQ_INSTS_VALU") else None
        out["valu_issue_source"] = ("SQ_ACTIVE_INST_VALU - SQ_ACTIVE_INST_VALU2 over GRBM_GUI_ACTIVE / 8 / 4 x SIMDs, one rocprofv3 --pmc pass "
                                    "(tools/pmc_passes.sh group ISSUE); what the two counters count: profiles/r03c_issue_counter_calibration.txt")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
