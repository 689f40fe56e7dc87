"""The exact per-block code the HIP kernels run (basisu_rs_amd/csrc/bu_uastc_*.hpp), compiled for the
host by tests/host_emul and compared with the oracle.  This is how the kernel logic is iterated on in
a container without a GPU; the `-m gpu` tests repeat the comparison through the C ABI on the device."""
import numpy as np
import pytest

from basisu_rs_amd import synth

ALL = ["astc", "bc7", "etc1", "etc2", "rgba"]


def _compare(emul, oracle, target, blocks):
    eo, es = emul.batch(target, blocks)
    oo, os_ = oracle.batch(target, blocks)
    assert (es == os_).all(), "status differs at block %d" % np.nonzero(es != os_)[0][0]
    ok = os_ == 0
    bad = np.nonzero((eo[ok] != oo[ok]).any(axis=1))[0]
    assert bad.size == 0, "output differs: block %s mode %d" % (blocks[ok][bad[0]].tobytes().hex(), synth.block_modes(blocks[ok][bad[:1]])[0])
    return int(ok.sum())


@pytest.mark.parametrize("target", ALL)
def test_emul_reproduces_reference_vectors(golden, emul, target):
    out, st = emul.batch(target, golden["uastc"])
    assert (st == 0).all() and (out == golden[target]).all()


@pytest.mark.parametrize("target", ALL)
def test_emul_matches_oracle_on_raw_random_blocks(emul, oracle, target):
    rng = np.random.default_rng(11)
    blocks = rng.integers(0, 256, size=(300_000, 16), dtype=np.uint8)
    _compare(emul, oracle, target, blocks)


@pytest.mark.parametrize("target", ALL)
def test_emul_matches_oracle_on_random_valid_blocks(emul, oracle, target):
    blocks = synth.atlas_rand(300_000, seed=3)
    assert _compare(emul, oracle, target, blocks) == blocks.shape[0]


@pytest.mark.parametrize("target", ALL)
def test_emul_matches_oracle_on_high_contrast_blocks(emul, oracle, target):
    """endpoints at the extremes, most texels at one of them (synth.atlas_contrast): the ETC modifier clamps, lumas more than
    2^15 away from the thresholds (saturating i16 lanes of the selector stage), EAC tables run into 0 / 255"""
    blocks = synth.atlas_contrast(300_000, seed=17)
    assert _compare(emul, oracle, target, blocks) == blocks.shape[0]


@pytest.mark.parametrize("target", ALL)
def test_emul_matches_oracle_per_mode_dense(golden, emul, oracle, target):
    """every mode equally: keep a golden block's mode code, randomise everything after it"""
    rng = np.random.default_rng(5)
    base = np.repeat(golden["uastc"], 400, axis=0)  # 243 200 blocks, 12 800 per mode
    noise = rng.integers(0, 256, size=base.shape, dtype=np.uint8)
    blocks = noise.copy()
    blocks[:, 0] = (base[:, 0] & 0x7F) | (noise[:, 0] & 0x80)
    _compare(emul, oracle, target, blocks)


def test_emul_solid_colour_blocks_cover_bc7_mode5_fallback(emul, oracle):
    """UASTC mode 8 -> BC7 mode 5 happens only when a channel is 0 and another 255 (bc7.rs:335-352);
    the reference vectors never reach it."""
    vals = np.array([0, 1, 2, 127, 128, 254, 255], dtype=np.uint64)
    r, g, b, a = np.meshgrid(vals, vals, vals, vals, indexing="ij")
    rgba = (r | (g << 8) | (b << 16) | (a << 24)).reshape(-1)
    rng = np.random.default_rng(9)
    blocks = rng.integers(0, 256, size=(rgba.size, 16), dtype=np.uint8)
    lo = blocks[:, :8].copy().view("<u8").reshape(-1)
    lo = (lo & ~np.uint64((1 << 37) - 1)) | np.uint64(0x17) | (rgba << np.uint64(5))  # mode 8 code = 0b10111
    blocks[:, :8] = lo.view(np.uint8).reshape(-1, 8)
    assert (synth.block_modes(blocks) == 8).all()
    for t in ALL:
        _compare(emul, oracle, t, blocks)
    out, _ = emul.batch("bc7", blocks)
    assert ((out[:, 0] & 0x3F) == 0x20).any(), "BC7 mode 5 fallback not exercised"


def test_emul_is_clean_under_ubsan(golden):
    """the per-block code once more under -fsanitize=undefined (aborts on the first report): reference vectors, raw random
    blocks and every mode densely, all five targets -- in a child process, so that an abort is a test failure"""
    import os
    import subprocess
    import sys

    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_emul")
    subprocess.check_call(["make", "-s", "-C", here, "libbu_emul_ubsan.so"])
    code = r"""
import ctypes, sys, numpy as np
sys.path.insert(0, %r)
from basisu_rs_amd import synth
lib = ctypes.CDLL(%r)
lib.bu_emul_batch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
lib.bu_emul_batch.restype = None
g = synth.load_golden(%r)
rng = np.random.default_rng(21)
raw = rng.integers(0, 256, size=(40000, 16), dtype=np.uint8)
base = np.repeat(g["uastc"], 40, axis=0)
noise = rng.integers(0, 256, size=base.shape, dtype=np.uint8)
dense = noise.copy()
dense[:, 0] = (base[:, 0] & 0x7F) | (noise[:, 0] & 0x80)
for blocks in (g["uastc"], raw, dense):
    b = np.ascontiguousarray(blocks)
    for t in range(5):
        out = np.zeros((b.shape[0], 64), dtype=np.uint8)
        st = np.zeros(b.shape[0], dtype=np.uint8)
        lib.bu_emul_batch(t, b.ctypes.data, b.shape[0], out.ctypes.data, st.ctypes.data)
print("clean")
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(here, "libbu_emul_ubsan.so"),
       os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "uastc_kat.bin"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "clean" in r.stdout, r.stderr[-2000:]
