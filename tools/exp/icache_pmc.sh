#!/bin/bash
# instruction-cache counters of the UASTC kernels (EXPERIMENT): one --pmc pass, --kernel-trace only beside it
export TMPDIR=/tmp
mkdir -p gpurun_out/icache
rocprofv3 -L 2>/dev/null | grep -i -E "ICACHE|IFETCH|INST_LEVEL" | head -40 > gpurun_out/icache/avail.txt
rm -rf gpurun_out/icache/p1
timeout 600 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --output-format csv -d gpurun_out/icache/p1 -- python3 tools/exp/pmc_run_all.py bc7 astc etc1 etc2 rgba > gpurun_out/icache/p1.log 2>&1
python3 - <<'PY' | tee gpurun_out/icache/summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/icache/p1/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "bu_uastc" not in k: continue
        acc[k.split("(")[0][-60:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
tail -5 gpurun_out/icache/p1.log
