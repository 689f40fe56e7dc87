"""Reads a rocprofv3 kernel-trace CSV of a multi-stream run (tools/exp/ab_streams.py under `rocprofv3 --kernel-trace`) and prints, for the
mode-sorted kernel's last N dispatches: hardware queues used per stream, how many dispatches overlap their predecessor, the average
number of kernels running, and how far the streams drift apart (end time of each stream's last dispatch relative to the earliest)."""
import csv, sys, collections
f, last = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 304
rows = [r for r in csv.DictReader(open(f)) if "sorted_kernel" in r["Kernel_Name"]]
rows = rows[-last:]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
q = collections.defaultdict(set)
for r in rows: q[r["Stream_Id"]].add(r["Queue_Id"])
print("stream -> queues:", {k: sorted(v) for k, v in sorted(q.items())})
se = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
dur = [e - s for s, e in se]
wall = max(e for _, e in se) - se[0][0]
print("dispatches %d  span avg %.0f ns  start-to-start %.0f ns  overlapping predecessor %d  avg running %.2f  wall/dispatch %.0f ns" % (
    len(se), sum(dur) / len(dur), (se[-1][0] - se[0][0]) / (len(se) - 1), sum(1 for i in range(len(se) - 1) if se[i + 1][0] < se[i][1]), sum(dur) / wall, wall / len(se)))
lastend = {}
for r in rows: lastend[r["Stream_Id"]] = max(lastend.get(r["Stream_Id"], 0), int(r["End_Timestamp"]))
m = min(lastend.values())
print("drift: last end per stream relative to the earliest (us):", {k: round((v - m) / 1e3, 1) for k, v in sorted(lastend.items())})
