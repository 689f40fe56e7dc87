#!/bin/bash
# EXPERIMENT: VALU instruction counts of two library variants: tools/exp/valu_ab.sh libA.so libB.so   (TARGETS="etc1 etc2")
export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  rm -rf gpurun_out/valu_ab/$name; mkdir -p gpurun_out/valu_ab
  BASISU_HIP_LIB=$PWD/$lib timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES --output-format csv -d gpurun_out/valu_ab/$name -- python3 tools/exp/pmc_run_all.py ${TARGETS:-etc1 etc2} > gpurun_out/valu_ab/$name.log 2>&1
done
python3 - "$@" <<'PY'
import csv, glob, collections, re, sys, os
for lib in sys.argv[1:]:
    name = os.path.basename(lib)[:-3]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("gpurun_out/valu_ab/%s/**/*counter_collection.csv" % name, recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.search(r"bu_uastc_\w+<[^>]*>", row["Kernel_Name"])
            if m: acc[m.group(0)][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in sorted(acc.items()):
        print(name, k, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())})
PY
