"""Static per-(target, mode) instruction counts of the per-block code: compile tools/exp/mode_isa.hip to gfx950
assembly and count VALU / SALU / LDS / VMEM instructions of each kernel (minus a fixed prologue measured on the
cheapest kernel is NOT subtracted: compare rows, not absolutes)."""
import os, re, subprocess, sys, collections
HERE = os.path.dirname(os.path.abspath(__file__))
out = "/tmp/mode_isa.s"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out,
                       os.path.join(HERE, "mode_isa.hip")])
cur = None; counts = {}
for line in open(out):
    m = re.match(r"^(_Z\w*mode_kernel\w*):", line)
    if m:
        cur = m.group(1); counts[cur] = collections.Counter(); continue
    if cur is None: continue
    if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"): cur = None; continue
    t = line.strip().split()
    if not t or t[0].startswith((".", ";")) or t[0].endswith(":"): continue
    op = t[0]
    k = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"
    counts[cur][k] += 1
names = ["ASTC", "BC7", "ETC1", "ETC2", "RGBA"]
rows = {}
for k, c in counts.items():
    m = re.search(r"mode_kernelILi(\d+)ELi(\d+)E", k)
    rows[(int(m.group(1)), int(m.group(2)))] = c
print("mode " + " ".join("%-16s" % n for n in names) + "   (valu/lds)")
for md in range(19):
    print("%4d " % md + " ".join("%6d/%-9d" % (rows[(t, md)]["valu"], rows[(t, md)]["lds"]) for t in range(5)))
print(" avg " + " ".join("%6.0f/%-9.0f" % (sum(rows[(t, m)]["valu"] for m in range(19)) / 19, sum(rows[(t, m)]["lds"] for m in range(19)) / 19) for t in range(5)))
