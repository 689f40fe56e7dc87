#!/bin/bash
# rocprofv3 passes over EVERY shipped kernel (tools/exp/pmc_run_all.py): one --kernel-trace --stats run, then separate
# --pmc runs per counter group (--pmc only ever combined with --kernel-trace; the program sits directly after `--`).
# Output: gpurun_out/pmc/{kernel_stats.csv, pmc_all.json, pmc_bc7.json, pmc_summary.txt}; copy what is judged into profiles/.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
rm -rf gpurun_out/pmc/trace
export PMC_REPS=16   # 384 launches per kernel in the kernel-trace pass: steady-state durations (the counter passes below use one rotation)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc/trace -- python3 tools/exp/pmc_run_all.py "$@" > gpurun_out/pmc/trace.log 2>&1
export PMC_REPS=1
find gpurun_out/pmc/trace -name "*kernel_stats*.csv" | head -1 | while read f; do cp "$f" gpurun_out/pmc/kernel_stats.csv; done
run() { name=$1; shift; rm -rf gpurun_out/pmc/$name; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc/$name -- python3 tools/exp/pmc_run_all.py ${TARGETS:-} > gpurun_out/pmc/$name.log 2>&1; }
run lds SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS
run sq2 SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
python3 - <<'PY' | tee gpurun_out/pmc/pmc_summary.txt
import csv, glob, collections, json, re
def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    return k.split("(")[0].strip()
summary = collections.defaultdict(dict)
for name in ("lds", "sq", "sq2", "fetch", "write", "grbm"):
    for f in glob.glob("gpurun_out/pmc/%s/**/*counter_collection.csv" % name, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "bu_" not in k: continue
            acc[short(k)][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, d in acc.items():
            for c, v in d.items():
                summary[k][c] = sum(v) / len(v)
                summary[k]["n_launches"] = len(v)
for f in glob.glob("gpurun_out/pmc/kernel_stats.csv"):
    for row in csv.DictReader(open(f)):
        k = short(row["Name"])
        if k in summary:
            summary[k]["trace_avg_ns"] = float(row["AverageNs"]); summary[k]["trace_calls"] = int(row["Calls"])
            summary[k]["trace_min_ns"] = float(row["MinNs"]); summary[k]["trace_max_ns"] = float(row["MaxNs"])
for k, d in sorted(summary.items()):
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 reports half of a wide coalesced read stream (MI355X_MICROARCH.md, HBM)
        d["hbm_bytes_per_launch"] = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
    if "SQ_WAVE_CYCLES" in d and d["SQ_WAVE_CYCLES"]:
        d["wait_share"] = d.get("SQ_WAIT_ANY", 0) / d["SQ_WAVE_CYCLES"]
    if d.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_conflict_share"] = d.get("SQ_LDS_BANK_CONFLICT", 0) / d["SQ_LDS_IDX_ACTIVE"]
    d["kernel"] = k
    print(k)
    for c in sorted(d):
        if c != "kernel": print("    %-28s %s" % (c, round(d[c], 4) if isinstance(d[c], float) else d[c]))
json.dump(summary, open("gpurun_out/pmc/pmc_all.json", "w"), indent=1, sort_keys=True)
for k, d in summary.items():
    if re.match(r"bu_uastc_sorted_kernel<1,", k) and "hbm_bytes_per_launch" in d:  # the BC7 headline kernel
        json.dump(d, open("gpurun_out/pmc/pmc_bc7.json", "w"), indent=1, sort_keys=True)
PY
