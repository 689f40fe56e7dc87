#!/bin/bash
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
R=$GRAFT_REPO_ROOT
rm -rf /tmp/mp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d /tmp/mp -- python3 $R/tools/exp/r6_host.py multi2 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/mp/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "sorted_kernel<1" in k or "multi_kernel<1" in k:
        key = ("multi" if "multi" in k else "plain") + " grid " + r["Grid_Size"]
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, d in acc.items():
    print(key, {c: round(sum(v) / len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
