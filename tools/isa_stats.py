#!/usr/bin/env python3
"""ISA census of a kernel in the built library: VGPR / spill / scratch / LDS from the code-object notes, and a histogram of
the VALU instructions of the kernel (or of the address range of its node step: the region between the first and last
v_cvt_f32_ubyte of an unrolled step), priced with the issue classes measured by tools/ubench/valu_issue.hip
(profiles/r02_valu_issue_ubench.txt): FAST 2.33, SLOW 4.2, TRANS 8.1 cycles per wave64 instruction per SIMD.

usage: tools/isa_stats.py [kernel-name-regex] [--step]     default regex: render_kernel<false, false, false, 0, 3, false, false>
"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
FAST = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_fmac_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32",
        "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32", "v_not_b32", "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32",
        "v_subb_co_u32", "v_accvgpr_write_b32", "v_accvgpr_read_b32", "v_nop"}
TRANS = {"v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_rcp_f64", "v_sqrt_f64", "v_rsq_f64", "v_exp_f32", "v_log_f32", "v_div_scale_f64",
         "v_fma_f64", "v_mul_f64", "v_add_f64", "v_div_fmas_f64", "v_div_fixup_f64", "v_rcp_iflag_f32"}
COST = {"fast": 2.33, "slow": 4.2, "trans": 8.1}


def klass(op):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if base in FAST:
        return "fast"
    if base in TRANS:
        return "trans"
    return "slow"


def code_objects(lib):
    td = tempfile.mkdtemp()
    so = shutil.copy(lib, td)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", os.path.basename(so)], cwd=td, check=True, capture_output=True)
    return sorted(glob.glob(os.path.join(td, "*gfx950")))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    pat = re.compile(args[0] if args else r"render_kernel<false, false, false, 0, 3, false, false>")
    lib = os.environ.get("PBRT_HIP_LIB_DIR", os.path.join(ROOT, "pbrt_amd", "lib")) + "/libpbrt_hip.so"
    for co in code_objects(lib):
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--demangle", co], capture_output=True, text=True, check=True).stdout
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
        blocks = re.split(r"\n(?=[0-9a-f]{16} <)", dis)
        for blk in blocks:
            head = blk.split("\n", 1)[0]
            m = re.match(r"[0-9a-f]{16} <(.*)>:", head)
            if not m or not pat.search(m.group(1)):
                continue
            name = m.group(1)
            lines = [l.split("//")[0].strip() for l in blk.split("\n")[1:] if l.strip() and not l.strip().startswith("<")]
            ops = [l.split()[0] for l in lines if l and not l.endswith(":")]
            if "--step" in sys.argv:
                idx = [i for i, o in enumerate(ops) if o.startswith("v_cvt_f32_ubyte")]
                # one unrolled step: from the global_load_dwordx4 group before the first cvt to the ds_read that pops
                first = idx[0]
                while first > 0 and not ops[first].startswith("global_load"):
                    first -= 1
                while first > 0 and ops[first - 1].startswith(("global_load", "s_waitcnt", "v_lshlrev")):
                    first -= 1
                last = idx[0]
                while last < len(ops) - 1 and not ops[last].startswith("ds_read"):
                    last += 1
                ops = ops[first:last + 1]
            hist = {}
            for o in ops:
                hist[o] = hist.get(o, 0) + 1
            if "--raw" in sys.argv:
                import json
                print(json.dumps(hist))
                return
            valu = {o: c for o, c in hist.items() if o.startswith("v_")}
            tot = {"fast": 0, "slow": 0, "trans": 0}
            for o, c in valu.items():
                tot[klass(o)] += c
            cyc = sum(tot[k] * COST[k] for k in tot)
            print(f"== {name}" + ("  [one node step]" if "--step" in sys.argv else ""))
            mangled = None
            print(f"   instructions {len(ops)}: VALU {sum(valu.values())} (fast {tot['fast']}, slow {tot['slow']}, trans {tot['trans']}) = {cyc:.0f} issue cycles; "
                  f"SALU {sum(c for o, c in hist.items() if o.startswith('s_'))}, VMEM {sum(c for o, c in hist.items() if o.startswith(('global_', 'buffer_', 'flat_', 'scratch_')))}, "
                  f"LDS {sum(c for o, c in hist.items() if o.startswith('ds_'))}")
            for o, c in sorted(valu.items(), key=lambda kv: -kv[1] * COST[klass(kv[0])])[:40]:
                print(f"      {c:5d} x {o:28s} {klass(o):5s} {c * COST[klass(o)]:7.0f}")
        # resource notes of the matching kernels
        for blk in notes.split(".agpr_count:")[1:]:
            nm = re.search(r"\.name:\s+(\S+)", blk)
            if not nm:
                continue
            dem = subprocess.run(["c++filt", nm.group(1)], capture_output=True, text=True).stdout.strip()
            if pat.search(dem):
                f = {k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1) for k in
                     ("vgpr_count", "vgpr_spill_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size")}
                print(f"   notes {dem}: {f}")


if __name__ == "__main__":
    main()
