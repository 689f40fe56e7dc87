"""python tools/exp/trace_periods.py kernel_trace.csv [name-substring] [group]: what rocprofv3's kernel trace says about a pipeline of overlapping launches of one kernel --
span of a dispatch, completion period (end-to-end of consecutive completions), dispatches running on average, hardware queues; `group` dispatches form one unit of work
(the four launches of a texture array) whose time is group x period.  The last three quarters of the dispatches are counted (clocks)."""
import csv, sys
f = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else "sorted_kernel"; group = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rows = [r for r in csv.DictReader(open(f)) if sub in r["Kernel_Name"]]
# the pipeline's launches all have the same grid: take the most frequent grid size
import collections
g = collections.Counter(r["Grid_Size_X"] for r in rows).most_common(1)[0][0]
rows = [r for r in rows if r["Grid_Size_X"] == g]
rows.sort(key=lambda r: int(r["End_Timestamp"]))
rows = rows[len(rows) // 4:]
se = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
dur = [e - s for s, e in se]
ends = [e for _, e in se]
gaps = [b - a for a, b in zip(ends, ends[1:])]
med = sorted(gaps)[len(gaps) // 2]
gaps_ok = [x for x in gaps if x < 20 * med]  # (pauses between the bench's phases are not periods)
wall = sum(gaps_ok)
print("kernel: %s   grid %s threads   dispatches counted: %d" % (rows[0]["Kernel_Name"][:90], g, len(rows)))
print("span of one dispatch        avg %.1f us  min %.1f  max %.1f" % (sum(dur) / len(dur) / 1e3, min(dur) / 1e3, max(dur) / 1e3))
print("completion period           median %.2f us -> %d dispatches = %.1f us   (average over everything incl. the pauses between the bench's phases: %.2f us -> %.1f us)" % (
    med / 1e3, group, group * med / 1e3, wall / len(gaps_ok) / 1e3, group * wall / len(gaps_ok) / 1e3))
sg = sorted(gaps_ok)
print("completion gaps, percentiles  p10 %.2f  p25 %.2f  p50 %.2f  p75 %.2f  p90 %.2f us;  share of gaps within 10 %% of the median: %.0f %%" % (
    sg[len(sg) // 10] / 1e3, sg[len(sg) // 4] / 1e3, sg[len(sg) // 2] / 1e3, sg[3 * len(sg) // 4] / 1e3, sg[9 * len(sg) // 10] / 1e3,
    100.0 * sum(1 for x in sg if abs(x - med) <= 0.1 * med) / len(sg)))
# steady stretches: runs of >= 64 consecutive completions without a pause (gap < 3 x median): period inside them
runs, cur = [], []
for x in gaps:
    if x < 3 * med: cur.append(x)
    else:
        if len(cur) >= 64: runs.append(cur)
        cur = []
if len(cur) >= 64: runs.append(cur)
if runs:
    tot = sum(sum(r) for r in runs); n = sum(len(r) for r in runs)
    print("steady stretches (>= 64 completions without a pause): %d stretches, %d completions, %.2f us per completion" % (len(runs), n, tot / n / 1e3))
print("dispatches running (avg)    %.2f   (span / period)" % ((sum(dur) / len(dur)) / (wall / len(gaps_ok))))
print("starting before predecessor's end: %d of %d   hardware queues: %s" % (sum(1 for i in range(len(se) - 1) if sorted(se)[i + 1][0] < sorted(se)[i][1]), len(se) - 1, sorted(set(r["Queue_Id"] for r in rows))))
