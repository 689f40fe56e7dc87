"""python tools/exp/bench_summary.py bench.json ... : the fields of a bench line one looks at first"""
import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f)); r = d["roofline"]; e = d.get("extra", {})
    print("==", f)
    print("value %.0f Mblocks/s  ms_per_step %.6f  frac %.4f  drain-incl %.4f  first-start-to-last-end/K %s ns" % (
        d["value"], d["ms_per_step"], r["frac"], r.get("frac_by_strict_bracket", 0), r.get("strict_bracket_ns_per_step")))
    print("  span %s  period(rocprof, steady) %s  frac(rocprof period) %s  traffic %s  host/event ms %s/%s late %s" % (
        r.get("kernel_span_ns"), r.get("period_ns_by_rocprofv3"), r.get("frac_by_rocprofv3_period"), r.get("traffic"),
        d["config"]["timed_region"]["host_ms"], d["config"]["timed_region"]["event_ms"], d["config"]["timed_region"]["host_started_late"]))
    t = r.get("rocprofv3_this_run") or {}
    print("  trace:", {k: v for k, v in t.items() if k != "source"})
    for k in ("one_launch_at_a_time", "launches_in_flight_matrix", "atlases_2_one_launch", "uastc_to_astc", "uastc_to_etc1", "uastc_to_etc2", "uastc_to_rgba32",
              "array512_one_launch", "array512_four_launches_in_flight", "array512_one_call_in_flight", "batch_8_atlases_separate_allocations", "batch_in_flight_512_atlases_one_call", "copy_ceiling", "coherent_atlas"):
        v = e.get(k)
        if isinstance(v, dict):
            v = {a: b for a, b in v.items() if a not in ("note", "per_launch")}
        print("  %s: %s" % (k, json.dumps(v)[:520]))
    print("  errors:", {k: e[k] for k in e if k.endswith("error")})
    if "cpu_baseline" in d: print("  cpu:", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
