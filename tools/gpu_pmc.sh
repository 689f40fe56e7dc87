#!/bin/bash
# rocprofv3 PMC passes (separate runs per counter group; --pmc only with --kernel-trace)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
run() { name=$1; shift; rm -rf gpurun_out/pmc/$name; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc/$name -- python3 tools/exp/pmc_run.py > gpurun_out/pmc/$name.log 2>&1; }
run lds SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
python3 - <<'PY'
import csv, glob, collections
summary = {}
for name in ("lds", "sq", "fetch", "write", "grbm"):
    files = glob.glob("gpurun_out/pmc/%s/**/*counter_collection.csv" % name, recursive=True)
    for f in files:
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "bu_" not in k: continue
            acc[k.split("(")[0][-40:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, d in acc.items():
            print(name, k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=%d" % len(next(iter(d.values()))))
            for c, v in d.items():
                summary.setdefault(k, {})[c] = sum(v) / len(v)
import json
for k, d in summary.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 reports half of a wide coalesced read stream (MI355X_MICROARCH.md, HBM)
        d["hbm_bytes_per_launch"] = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
        d["kernel"] = k
        json.dump(d, open("gpurun_out/pmc/pmc_bc7.json", "w"), indent=1)
        print("hbm bytes per launch", d["hbm_bytes_per_launch"])
PY
