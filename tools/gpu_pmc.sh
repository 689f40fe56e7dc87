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
    k = k.replace("(anonymous namespace)::", "").replace("void ", "").replace("(int)", "").replace("(bool)", "")
    return k.split("(")[0].strip()
# one entry per (kernel instantiation, grid size in threads): the exclusive BC7 shape runs at 2^20 blocks and as `array512` at 2^25
summary = collections.defaultdict(dict)
for name in ("lds", "sq", "sq2", "fetch", "write", "grbm"):
    for f in glob.glob("gpurun_out/pmc/%s/**/*counter_collection.csv" % name, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "bu_" not in k: continue
            acc["%s grid=%s" % (short(k), row.get("Grid_Size", "?"))][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, d in acc.items():
            for c, v in d.items():
                summary[k][c] = sum(v) / len(v)
                summary[k]["n_launches"] = len(v)
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc/trace/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "bu_" not in row["Kernel_Name"]: continue
        g = int(row["Grid_Size_X"]) * int(row.get("Grid_Size_Y", 1) or 1) * int(row.get("Grid_Size_Z", 1) or 1)
        dur["%s grid=%d" % (short(row["Kernel_Name"]), g)].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
for k, v in dur.items():
    v = v[len(v) // 4:] if len(v) >= 8 else v  # the first launches of a kernel start from whatever the previous one left
    summary[k]["trace_avg_ns"] = sum(v) / len(v); summary[k]["trace_calls"] = len(v); summary[k]["trace_min_ns"] = min(v); summary[k]["trace_max_ns"] = max(v)
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
    # the BC7 headline kernel: the shared-policy shape (256 threads x 4 blocks per thread, rectangular tiles: layout 1), 512 workgroups
    if re.match(r"bu_uastc_sorted_kernel<1, *256, *4,.*, *1> grid=131072$", k) and "hbm_bytes_per_launch" in d:
        json.dump(d, open("gpurun_out/pmc/pmc_bc7.json", "w"), indent=1, sort_keys=True)
PY
# ---- BASELINE config 5's launch on its own: the 512-slice array (2^25 blocks) through the exclusive BC7 shape, kernel trace + HBM counters ----
export PMC_REPS=160   # 640 launches = 125 ms: the 1 GiB launches need ~100 ms of the same load before the clocks settle (bench.py waits that long too); the last half counts
rm -rf gpurun_out/pmc/a512_trace gpurun_out/pmc/a512_fetch gpurun_out/pmc/a512_write
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc/a512_trace -- python3 tools/exp/pmc_run_all.py array512 > gpurun_out/pmc/a512_trace.log 2>&1
export PMC_REPS=1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc/a512_fetch -- python3 tools/exp/pmc_run_all.py array512 > gpurun_out/pmc/a512_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc/a512_write -- python3 tools/exp/pmc_run_all.py array512 > gpurun_out/pmc/a512_write.log 2>&1
find gpurun_out/pmc/a512_trace -name "*kernel_stats*.csv" | head -1 | while read f; do cp "$f" gpurun_out/pmc/kernel_stats_array512.csv; done
python3 - <<'PY' | tee gpurun_out/pmc/pmc_summary_array512.txt
import csv, glob, json
d = {"kernel": "bu_uastc_sorted_kernel<BC7, 512, 2> (exclusive shape), 2^25 blocks in one launch: BASELINE config 5 on one GPU", "blocks": 1 << 25, "algorithmic_bytes": 32 << 25}
for name, c in (("a512_fetch", "FETCH_SIZE"), ("a512_write", "WRITE_SIZE")):
    v = []
    for f in glob.glob("gpurun_out/pmc/%s/**/*counter_collection.csv" % name, recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == c and "bu_uastc_sorted_kernel<1," in row["Kernel_Name"].replace("(int)", ""): v.append(float(row["Counter_Value"]))
    if v: d[c] = sum(v) / len(v); d[c + "_launches"] = len(v)
if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
    d["hbm_bytes_per_launch"] = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024  # KiB; gfx950 x2 read correction (MI355X_MICROARCH.md)
    d["traffic_over_algorithmic"] = d["hbm_bytes_per_launch"] / d["algorithmic_bytes"]
dur = []
for f in glob.glob("gpurun_out/pmc/a512_trace/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "bu_uastc_sorted_kernel<1," in row["Kernel_Name"].replace("(int)", ""): dur.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
if dur:
    d["trace_calls_all"] = len(dur); d["trace_avg_ns_all_launches"] = sum(dur) / len(dur)
    dur = dur[len(dur) // 2:]   # (steady clocks)
    d["trace_avg_ns"] = sum(dur) / len(dur); d["trace_calls"] = len(dur); d["trace_min_ns"] = min(dur); d["trace_max_ns"] = max(dur)
    d["frac_of_8TBs_by_rocprofv3_kernel_avg"] = d["algorithmic_bytes"] / d["trace_avg_ns"] / 8000.0
for k in sorted(d): print("%-40s %s" % (k, d[k]))
json.dump(d, open("gpurun_out/pmc/pmc_bc7_array512.json", "w"), indent=1, sort_keys=True)
PY
rm -rf gpurun_out/pmc/a512_trace gpurun_out/pmc/a512_fetch gpurun_out/pmc/a512_write
