#!/bin/bash
# HBM traffic of the render kernel per BASELINE config: FETCH_SIZE and WRITE_SIZE in separate
# rocprofv3 --pmc runs (one launch of the scene's frame each), summed over the kernel's dispatches.
#   tools/traffic_all.sh <tag>   ->  gpurun_out/<tag>_traffic.json   (run on the GPU box from the repo root)
set -u
TAG=${1:-r3}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for S in cornell_box orthographic obj materials materials_hdr mesh1m; do
  SPP=""
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/${TAG}_pmc_${S}_$C
    timeout -s KILL 150 rocprofv3 --pmc $C --output-format csv -d $OUT/${TAG}_pmc_${S}_$C -- python3 $GRAFT_REPO_ROOT/tools/quick_perf.py $S.yaml 1 $SPP > $OUT/${TAG}_pmc_${S}_$C.log 2>&1 || { echo "pmc $S $C failed"; tail -3 $OUT/${TAG}_pmc_${S}_$C.log; exit 1; }
  done
done
python3 - <<PY
import csv, glob, json, os
out = {}
for s in ("cornell_box", "orthographic", "obj", "materials", "materials_hdr", "mesh1m"):
    rec = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        tot = 0.0
        for f in glob.glob("$OUT/${TAG}_pmc_%s_%s/**/*counter_collection.csv" % (s, c), recursive=True):
            for row in csv.DictReader(open(f)):
                if "render_fused" in row["Kernel_Name"] and row["Counter_Name"] == c:
                    tot += float(row["Counter_Value"])
        rec[c + "_KiB"] = tot        # the counter counts kilobytes
    # gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes (calibrated on 16-byte-per-lane streaming
    # reads; other access widths are uncalibrated), WRITE_SIZE reads exactly: traffic = WRITE + 2 x FETCH
    rec["bytes_per_launch"] = int((2.0 * rec["FETCH_SIZE_KiB"] + rec["WRITE_SIZE_KiB"]) * 1024)
    rec["source"] = "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs (KiB, summed over the kernel's dispatches), bytes = (WRITE_SIZE + 2 x FETCH_SIZE) x 1024"
    rec["launch"] = "one frame of scenes/%s.yaml" % s
    out[s + ".yaml"] = rec
json.dump(out, open("$OUT/${TAG}_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
