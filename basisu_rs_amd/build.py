"""Builds the gfx950 shared library (the C ABI of include/basisu_hip.h) in-tree with hipcc.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
basisu_rs_amd/libbasisu_hip.so travels to the GPU box with the repository snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbasisu_hip.so")


def _newer_than(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def sources():
    out = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp", ".h"))]
    out.append(os.path.join(HERE, "..", "include", "basisu_hip.h"))
    return out


LIB_ST0 = os.path.join(HERE, "libbasisu_hip_st0.so")  # canary build: intrinsic nontemporal stores instead of the `sc1 nt` asm stores


def build_hip(force=False, verbose=False, canary=False):
    """Compile csrc/bu_hip.hip for gfx950 -> libbasisu_hip.so.  Returns the library path.
    canary=True builds libbasisu_hip_st0.so instead: the same sources with -DBU_ST_MODE=0 (every result store is the compiler's
    own nontemporal store, no inline asm, no hand-placed s_nop); tests/test_gpu_round3.py compares its output with the shipped
    build's, so a compiler that schedules differently around the asm store shows up as a difference."""
    LIB = LIB_ST0 if canary else globals()["LIB"]
    if not force and _newer_than(LIB, sources()):
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-pthread",
           # machine-LICM hoists every mode path's constants out of the chunk loop of the sorted kernel: ~30 extra VGPRs
           # (BC7 93 -> 62 without it), which decides between 16 and 32 resident waves per CU
           "-mllvm", "-disable-machine-licm",
           # every atomic here is either per-lane on different LDS counters or issued by one elected lane; the optimizer's
           # wave-reduction scaffolding (mbcnt / readlane loops) around them is pure overhead (BC7 11.45 -> 11.15 us)
           "-mllvm", "-amdgpu-atomic-optimizer-strategy=None",
           # kernel arguments arrive in SGPRs at wave launch (gfx950 kernarg preload; older firmware runs the compiler's
           # fallback prologue that loads them all up front).  Without it hipcc loads each argument where it is first used --
           # several s_load round trips behind the first barrier, issued while the chip's 16 MiB of block loads are in
           # flight (BC7 10.73 -> 10.41 us, ASTC 9.97 -> 9.72 in an A/B run)
           "-mllvm", "-amdgpu-kernarg-preload-count=16",
           "-o", LIB, os.path.join(CSRC, "bu_hip.hip")]
    if canary:
        cmd.insert(1, "-DBU_ST_MODE=0")
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed building " + os.path.basename(LIB))
    if verbose:
        sys.stderr.write(r.stderr)
    return LIB


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose="-v" in sys.argv))
