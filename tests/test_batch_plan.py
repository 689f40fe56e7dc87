"""Host logic of the pipelined batch call (csrc/bu_batch_plan.hpp, compiled as it is into the test-only host build): slices -> runs ->
launches.  Whatever the slice table, every block of every slice lands in exactly one launch, in order, with its block number; launches hold
about 2^20 blocks or more; a batch that would make fewer launches than streams has its large runs cut on tile / block-row boundaries."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMUL = os.path.join(ROOT, "tests", "host_emul", "libbu_emul.so")


@pytest.fixture(scope="module")
def plan():
    subprocess.run(["make", "-C", os.path.dirname(EMUL), "libbu_emul.so"], check=True, capture_output=True)
    lib = ctypes.CDLL(EMUL)
    U64P, SZP = ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_size_t)
    lib.bu_emul_plan_in_flight.argtypes = [ctypes.c_size_t, U64P, SZP, U64P, ctypes.c_size_t, U64P, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, U64P,
                                           ctypes.c_size_t, SZP]
    lib.bu_emul_plan_in_flight.restype = ctypes.c_size_t

    def run(in_addr, n_blocks, out_addr, bb, base, n_streams, bpr, max_runs=96, group_blocks=1 << 20):
        n = len(n_blocks)
        A = (ctypes.c_uint64 * n)(*in_addr)
        O = (ctypes.c_uint64 * n)(*out_addr)
        N = (ctypes.c_size_t * n)(*n_blocks)
        B = (ctypes.c_uint64 * n)(*base) if base is not None else None
        cap = n + 64
        rows = (ctypes.c_uint64 * (5 * cap))()
        nl = ctypes.c_size_t(0)
        got = lib.bu_emul_plan_in_flight(n, A, N, O, bb, B, n_streams, bpr, max_runs, group_blocks, rows, cap, ctypes.byref(nl))
        assert got <= cap
        return np.array(list(rows[:5 * got]), dtype=np.uint64).reshape(got, 5), nl.value

    return run


def _check_cover(rows, in_addr, n_blocks, out_addr, bb, base):
    """the rows, in order, walk every non-empty slice's blocks exactly once: same input bytes, output bytes and block numbers"""
    want = []
    nb = 0
    for i, n in enumerate(n_blocks):
        b = base[i] if base is not None else nb
        nb = b + n
        if n:
            want.append([in_addr[i], out_addr[i], n, b])
    # flatten both into (in, out, base) triples per maximal contiguous extent
    def extents(seq):
        out = []
        for a, o, n, b in seq:
            if out and out[-1][0] + out[-1][2] * 16 == a and out[-1][1] + out[-1][2] * bb == o and out[-1][3] + out[-1][2] == b:
                out[-1][2] += n
            else:
                out.append([a, o, n, b])
        return out
    got = extents([[int(r[1]), int(r[2]), int(r[3]), int(r[4])] for r in rows])
    assert got == extents(want)


def test_large_slices_one_launch_each_small_ones_grouped(plan):
    big, small = 1 << 20, 65536
    sizes = [big, big + 5, small, small, small, 0, small, big]
    in_addr = [0x1000_0000 + i * 0x1000_0000 for i in range(len(sizes))]   # separate allocations
    out_addr = [0x9000_0000_00 + i * 0x1000_0000 for i in range(len(sizes))]
    rows, launches = plan(in_addr, sizes, out_addr, 16, None, 4, 0)
    _check_cover(rows, in_addr, sizes, out_addr, 16, None)
    # launches: big | big+5 | four small ones together (262 144 blocks, then the next run would pass 2^20) | big
    assert launches == 4
    assert [int(x) for x in rows[:, 0]] == [0, 1, 2, 2, 2, 2, 3]


def test_bc7_astc_group_up_to_2_23_blocks_per_launch(plan):
    """the BC7 / ASTC plan (group_blocks = 2^23): 64 atlases of 2^20 blocks in separate allocations are eight launches of eight runs, 512 slices of
    65 536 blocks six launches of at most 96 runs; a run of 2^23 blocks or more stays alone and one beyond that is cut"""
    def table(n, per):
        return [0x1000_0000 + i * 0x4000_0000 for i in range(n)], [0x9000_0000_00 + i * 0x4000_0000 for i in range(n)]
    a, o = table(64, 1 << 20)
    rows, launches = plan(a, [1 << 20] * 64, o, 16, None, 4, 1024, group_blocks=1 << 23)
    _check_cover(rows, a, [1 << 20] * 64, o, 16, None)
    assert launches == 8 and [int(x) for x in rows[:, 0]] == [k // 8 for k in range(64)]
    a, o = table(512, 65536)
    rows, launches = plan(a, [65536] * 512, o, 16, None, 4, 256, group_blocks=1 << 23)
    _check_cover(rows, a, [65536] * 512, o, 16, None)
    assert launches == 6 and [int((rows[:, 0] == j).sum()) for j in range(6)] == [96] * 5 + [32]
    sizes = [1 << 20, 1 << 23, 3 << 20, (1 << 23) + 1024, 5]
    a, o = table(len(sizes), 0)
    rows, launches = plan(a, sizes, o, 16, None, 4, 0, group_blocks=1 << 23)
    _check_cover(rows, a, sizes, o, 16, None)
    # 2^20 alone (the next run would pass 2^23) | 2^23 | 3 Mi alone (as before) | two pieces of the long run | the last 5 blocks
    assert launches == 6 and [int(r[3]) for r in rows] == [1 << 20, 1 << 23, 3 << 20, (1 << 22) + 1024, 1 << 22, 5]
    # the same table at the other targets' 2^20: the small runs group, everything else goes alone
    rows20, launches20 = plan(a, sizes, o, 16, None, 4, 0)
    assert launches20 == 6


def test_contiguous_array_is_cut_into_pieces_of_at_most_2_23_blocks_and_one_per_stream_at_least(plan):
    n_slices, per = 512, 65536
    in_addr = [0x4000_0000 + k * per * 16 for k in range(n_slices)]
    out_addr = [0x9000_0000_00 + k * per * 16 for k in range(n_slices)]
    for streams, want in ((1, 1), (2, 4), (3, 4), (4, 4), (8, 8)):  # 2^25 blocks: one launch on one stream, else pieces of <= 2^23, at least one per stream
        rows, launches = plan(in_addr, [per] * n_slices, out_addr, 16, None, streams, 256)
        _check_cover(rows, in_addr, [per] * n_slices, out_addr, 16, None)
        assert launches == want and len(rows) == want, (streams, launches)
        assert streams == 1 or max(int(r[3]) for r in rows) <= (1 << 23)
        align = 16 * 256  # lcm(16 rows of 256 blocks, 1024)
        assert all(int(r[3]) % align == 0 for r in rows[:-1])
        assert max(int(r[3]) for r in rows) - min(int(r[3]) for r in rows) <= align
    # RGBA32 with a pitch that is no multiple of 64: pieces are whole block rows (and whole 1024-block tiles)
    per, bpr = 192 * 4096, 192
    in_addr = [0x4000_0000 + k * per * 16 for k in range(8)]
    out_addr = [0x9000_0000_00 + k * per * 64 for k in range(8)]
    rows, launches = plan(in_addr, [per] * 8, out_addr, 64, None, 4, bpr)
    _check_cover(rows, in_addr, [per] * 8, out_addr, 64, None)
    assert launches == 4 and all(int(r[3]) % bpr == 0 and int(r[3]) % 1024 == 0 for r in rows[:-1]) and int(rows[-1][3]) % bpr == 0
    # several large arrays in one call (more launches than streams already): every one is still cut to <= 2^23-block launches
    big = 1 << 25
    rows, launches = plan([0x4000_0000 + k * (big * 16 + 4096) for k in range(5)], [big] * 5, [0x9000_0000_00 + k * (big * 16 + 4096) for k in range(5)], 16, None, 4, 256)
    _check_cover(rows, [0x4000_0000 + k * (big * 16 + 4096) for k in range(5)], [big] * 5, [0x9000_0000_00 + k * (big * 16 + 4096) for k in range(5)], 16, None)
    assert launches == 20 and all(int(r[3]) == (1 << 23) for r in rows)
    # a piece never falls below 2^20 blocks: 3 Mi blocks on 8 streams make three launches, 1.5 Mi blocks one
    for total, want in ((3 << 20, 3), (3 << 19, 1)):
        rows, launches = plan([0x4000_0000], [total], [0x9000_0000_00], 16, None, 8, 0)
        assert launches == want and sum(int(r[3]) for r in rows) == total


def test_random_slice_tables_cover_every_block_once(plan):
    rng = np.random.default_rng(5)
    for case in range(400):
        bb = int(rng.choice([8, 16, 64]))
        bpr = int(rng.choice([0, 64, 100, 192, 1024]))
        n = int(rng.integers(1, 260))
        sizes = [int(rng.choice([0, 1, 63, 1024, 65536, 1 << 20, int(rng.integers(1, 1 << 23))])) for _ in range(n)]
        if bpr:
            sizes = [s // bpr * bpr for s in sizes]
        in_addr, out_addr, base = [], [], []
        a, o, b = 0x1000_0000, 0x9000_0000_00, int(rng.integers(0, 1 << 40))
        for s in sizes:
            if rng.random() < 0.4:   # a gap in the input, the output or the numbering: a new run
                which = rng.integers(0, 3)
                a += 4096 if which == 0 else 0
                o += 4096 if which == 1 else 0
                b += 7 if which == 2 else 0
            in_addr.append(a); out_addr.append(o); base.append(b)
            a += s * 16; o += s * bb; b += s
        use_base = bool(rng.integers(0, 2))
        if not use_base:   # numbering back to back: recompute what the call will assume
            base = None
            # (gaps in the numbering cannot be expressed without index_base: remove them from the expectation by construction)
        streams = int(rng.integers(1, 9))
        grp = int(rng.choice([1 << 20, 1 << 22, 1 << 23]))
        rows, launches = plan(in_addr, sizes, out_addr, bb, base, streams, bpr, max_runs=int(rng.choice([4, 96])), group_blocks=grp)
        _check_cover(rows, in_addr, sizes, out_addr, bb, base)
        per_launch = {}
        for r in rows:
            per_launch.setdefault(int(r[0]), []).append(int(r[3]))
        assert sorted(per_launch) == list(range(launches))
        for j, ns in per_launch.items():
            # a launch of several runs holds at most `group_blocks` blocks
            assert len(ns) == 1 or sum(ns) <= grp, (case, j, ns)
            assert streams == 1 or len(ns) > 1 or ns[0] <= (1 << 23), (case, j, ns)
