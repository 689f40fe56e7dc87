"""Static per-(target, mode) instruction counts of the per-block code: compile tools/exp/mode_isa.hip to gfx950
assembly and count VALU / SALU / LDS / VMEM instructions of each kernel (minus a fixed prologue measured on the
cheapest kernel is NOT subtracted: compare rows, not absolutes)."""
import os, re, subprocess, sys, collections
HERE = os.path.dirname(os.path.abspath(__file__))
out = "/tmp/mode_isa.s"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out,
                       os.path.join(HERE, "mode_isa.hip")])
# issue cost per wave-instruction per SIMD at full occupancy, measured on gfx950 (tools/exp/opbench.hip, profiles/r02_opbench.txt)
FAST = {"v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32",
        "v_add_co_u32", "v_sub_co_u32", "v_xnor_b32"}
def clk(op, operands):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if base in FAST and not op.endswith(("_e64", "_sdwa", "_dpp")) and not re.search(r"\bs\d+\b|\bs\[|\bvcc\b|\bexec\b", operands):
        return 2.25
    if base == "v_mad_u16": return 8.3
    return 4.2
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
    if k == "valu": counts[cur]["clk"] += clk(op, line.split(None, 1)[1] if len(t) > 1 else "")
    if k == "lds": counts[cur]["clk_lds"] += (13.6 if "write_b128" in op or "write2_b64" in op else 8 if "add" in op else 4)
names = ["ASTC", "BC7", "ETC1", "ETC2", "RGBA"]
rows = {}
for k, c in counts.items():
    m = re.search(r"mode_kernelILi(\d+)ELi(\d+)E", k)
    rows[(int(m.group(1)), int(m.group(2)))] = c
only = [int(x) for x in sys.argv[1:]] or range(5)
print("mode " + " ".join("%-22s" % names[t] for t in only) + "   (valu instrs / est. SIMD clk / lds instrs)")
for md in range(19):
    print("%4d " % md + " ".join("%5d/%6.0f/%-9d" % (rows[(t, md)]["valu"], rows[(t, md)]["clk"], rows[(t, md)]["lds"]) for t in only))
print(" avg " + " ".join("%5.0f/%6.0f/%-9.0f" % (sum(rows[(t, m)]["valu"] for m in range(19)) / 19, sum(rows[(t, m)]["clk"] for m in range(19)) / 19, sum(rows[(t, m)]["lds"] for m in range(19)) / 19) for t in only))
