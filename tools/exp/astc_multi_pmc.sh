#!/bin/bash
# tools/exp/astc_multi_pmc.sh [target] : the plain large launch (64 adjacent atlases = one run of 2^26 blocks) against the multi-run launch (the same atlases in 64 separate
# allocations), counters per launch of the big kernels only
T=${1:-astc}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for sep in 0 1; do
  export SEP=$sep
  for grp in "lds SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS" "sq2 SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
    set -- $grp; name=$1; shift
    rm -rf /tmp/amp_$sep_$name
    timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/amp_${sep}_$name -- python3 $R/tools/exp/etc_long_walk.py $R/basisu_rs_amd/libbasisu_hip.so $T 0 > /dev/null 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections
for sep in (0, 1):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for name in ("lds", "sq", "sq2"):
        for f in glob.glob("/tmp/amp_%d_%s/**/*counter_collection.csv" % (sep, name), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if ("sorted_kernel" in k or "multi_kernel" in k) and int(r["Grid_Size"]) >= 512 * 1024:
                    key = ("multi" if "multi" in k else "plain") + " grid " + r["Grid_Size"] + " wg " + r["Workgroup_Size"]
                    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key, d in acc.items():
        n = len(next(iter(d.values())))
        if n < 4: continue
        print("SEP=%d" % sep, key, "launches", n)
        for c, v in sorted(d.items()):
            v = sorted(v); print("    %-24s %14.0f (median)" % (c, v[len(v) // 2]))
PY
