#!/bin/bash
# usage (on the GPU box): tools/fetch_size_calibration.sh [out-subdir]     (round 6)
# rocprofv3's FETCH_SIZE against a KNOWN byte count in the render kernel's access shape: tools/ubench/gather_wide pmc gathers 64-byte
# records (four own-lane dwordx4 loads) at random from a 112 MB table (past the 4 MiB L2s, inside the 256 MiB Infinity Cache) and from a
# 1.5 GB one (past it), one dispatch per line with its exact bytes.  The program itself follows `--` (no shell, no env between).
# MI355X_MICROARCH.md (HBM): FETCH_SIZE tallies 128-byte requests of a wide coalesced stream at 64 B -- "calibrate on a known byte count
# in your own access pattern".  Output: <out>/fetch_size_calibration.txt -> profiles/r06_fetch_size_calibration.txt.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${1:-fetch_cal}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE TCC_EA0_RDREQ_sum; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -- $R/tools/ubench/gather_wide pmc > $OUT/$c.log 2>&1
  echo "$c rc=$?"
done
timeout 300 $R/tools/ubench/gather_wide pmc > $OUT/plain.log 2>&1
python3 - "$OUT" <<'PY' | tee $OUT/fetch_size_calibration.txt
import csv, glob, re, sys
out = sys.argv[1]
known = [tuple(map(float, re.match(r"DISPATCH (\d+) table_MB (\d+) record_B 64 records (\d+) bytes (\d+)", l).groups())) for l in open(out + "/FETCH_SIZE.log") if l.startswith("DISPATCH")]
print("# FETCH_SIZE / TCC_EA0_RDREQ of rocprofv3 against known bytes: 64-byte records gathered at random (4 x dwordx4 per lane), tools/ubench/gather_wide pmc")
res = {}
for c in ("FETCH_SIZE", "TCC_EA0_RDREQ_sum"):
    rows = []
    for f in sorted(glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True)):
        rows += [r for r in csv.DictReader(open(f)) if "gather" in r["Kernel_Name"] and r["Counter_Name"] == c]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    # (a dispatch may be reported once per XCD / dimension: sum per dispatch id)
    per = {}
    for r in rows:
        per[int(r["Dispatch_Id"])] = per.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    res[c] = [per[k] for k in sorted(per)]
for i, (n, mb, recs, nbytes) in enumerate(known):
    fs = res["FETCH_SIZE"][i] * 1024 if i < len(res["FETCH_SIZE"]) else float("nan")
    rq = res["TCC_EA0_RDREQ_sum"][i] if i < len(res["TCC_EA0_RDREQ_sum"]) else float("nan")
    print(f"dispatch {int(n)} table {int(mb):5d} MB  records {recs:.4g}  known bytes {nbytes:.5g}  FETCH_SIZE x 1024 = {fs:.5g}  ratio {fs / nbytes:.4f}   "
          f"TCC_EA0_RDREQ {rq:.5g} = {rq / recs:.4f} per record")
r112 = [res["FETCH_SIZE"][i] * 1024 / k[3] for i, k in enumerate(known) if k[1] == 112 and i < len(res["FETCH_SIZE"])]
rbig = [res["FETCH_SIZE"][i] * 1024 / k[3] for i, k in enumerate(known) if k[1] != 112 and i < len(res["FETCH_SIZE"])]
if r112 and rbig:
    print(f"FETCH_SIZE_BYTES_PER_KNOWN_BYTE 112MB {sum(r112) / len(r112):.4f} 1536MB {sum(rbig) / len(rbig):.4f}")
PY
