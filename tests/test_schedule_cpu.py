"""Host-side proof of the persistent contraction kernel's schedule (fastspeech2_amd/csrc/fs2_sched.h: p_plan / p_unit, the same
source the device compiles, reached through the host-only test-aid library tests/aids/libfs2_testaid.so - include/fs2hip_testaid.h;
the product library exports none of it): for ANY number of real M-tiles - it is only known on the device - the workgroups'
unit lists cover every output tile's reduction range exactly once, tail parts sit where p_tail_finalize_kernel looks for them,
and no workgroup the launcher admits holds more than the 64 units its two-VGPR table can carry.  No GPU needed."""
import ctypes
import os
import random

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_aid = None


def aid():
    global _aid
    if _aid is None:
        path = os.path.join(ROOT, "tests", "aids", "libfs2_testaid.so")
        assert os.path.exists(path), "tests/aids/libfs2_testaid.so missing: run `make` (or __graft_entry__.build())"
        _aid = ctypes.CDLL(path)
    return _aid


def units_of(n_real, ntn, G, order, ks, nkc, tks_max, b, cap=64):
    buf = (ctypes.c_int * (cap * 5))()
    n = aid().fs2t_conv_gemm_p_units(n_real, ntn, G, order, ks, nkc, tks_max, b, ctypes.cast(buf, ctypes.c_void_p), cap)
    assert n >= 0
    assert n <= cap, (n, cap)
    return [tuple(buf[5 * i:5 * i + 5]) for i in range(n)]


def check(n_real, ntn, G, order, ks, nkc, tks_max):
    cover = {}
    parts_at = {}
    for b in range(G):
        us = units_of(n_real, ntn, G, order, ks, nkc, tks_max, b)
        assert len(us) <= 64
        for i, (mi, nt, kc0, nk, np_) in enumerate(us):
            assert 0 <= mi < n_real and 0 <= nt < ntn and nk > 0 and 0 <= kc0 and kc0 + nk <= nkc, (b, us[i])
            cover.setdefault((mi, nt), []).append((kc0, nk))
            if np_ > 1:
                assert i == len(us) - 1 and ks == 1 and np_ <= tks_max and nk * np_ == nkc, (b, us[i])      # a tail part is a workgroup's LAST unit
                parts_at.setdefault((mi, nt), []).append((b, kc0 // nk, np_))
            else:
                assert nk == nkc // ks
    assert len(cover) == n_real * ntn                                             # every tile ...
    for tile, rs in cover.items():
        rs.sort()
        pos = 0
        for kc0, nk in rs:                                                        # ... exactly once over its whole reduction
            assert kc0 == pos, (tile, rs)
            pos += nk
        assert pos == nkc, (tile, rs)
    for tile, ps in parts_at.items():                                             # where the finalize kernel expects the slabs
        ps.sort(key=lambda t: t[1])
        np_ = ps[0][2]
        assert [q for _, q, _ in ps] == list(range(np_)) and all(p[2] == np_ for p in ps)
        b0 = ps[0][0]
        if order == 0:
            assert b0 % 2 == 0 and b0 % np_ == 0 and [b for b, _, _ in ps] == [b0 + q for q in range(np_)]
        else:
            x, j0 = b0 & 7, b0 >> 3
            assert j0 % np_ == 0 and [b for b, _, _ in ps] == [x + 8 * (j0 + q) for q in range(np_)]
    return len(parts_at)


@pytest.mark.parametrize("order", [0, 1])
def test_every_tile_is_covered_exactly_once(order):
    rng = random.Random(11 + order)
    split_seen = 0
    for G in (256, 64):
        for ntn in (1, 2, 4, 8):
            for nkc, tks_max in ((16, 8), (16, 4), (8, 2), (4, 4), (4, 1), (9, 1)):
                for n_real in sorted({0, 1, 7, G // ntn, G // ntn + 1} | {rng.randrange(1, 200) for _ in range(6)}):
                    if n_real * ntn > 64 * G:
                        continue
                    split_seen += check(n_real, ntn, G, order, 1, nkc, tks_max)
    assert split_seen > 100                                                       # the sweep did exercise split tails


def test_uniform_ksplit_units():
    for order in (0, 1):
        for ks, nkc in ((2, 16), (4, 16), (2, 8)):
            for n_real, ntn in ((24, 2), (150, 2), (7, 1)):
                check(n_real, ntn, 256, order, ks, nkc, 1)


def test_production_shapes_tail_plan():
    """the launches the bench's step actually splits: k=9 data gradient (~165 real tiles x 2, 16 chunks) -> 4- or 2-way tails;
    the encoder's (24 x 2 tiles on 256 workgroups) -> all tail, 4 parts each"""
    assert check(165, 2, 256, 0, 1, 16, 8) == 165 * 2 - 256
    us = [units_of(24, 2, 256, 0, 1, 16, 8, b) for b in range(256)]
    assert sum(len(u) for u in us) == 192 and all(u[0][4] == 4 for u in us if u)


def test_launcher_bound_matches_the_unit_lists_near_64_rounds():
    """with order = 1 (per-XCD dealing) one group can need one more round than ceil(tiles / G): n_real = 2729,
    ntn = 6, G = 256 gives group 0 a 65th unit although ceil(16374 / 256) = 64.  The launcher now bounds BOTH orders with
    p_max_units; here that bound is checked to be exactly the longest unit list, and launches it admits (<= 64) are covered
    exactly once."""
    lib = aid()
    for n_real, ntn in ((2729, 6), (2730, 6), (2731, 6), (2720, 6), (16384, 1), (16385, 1), (8190, 2), (8193, 2)):
        for order in (0, 1):
            G = 256
            longest = max(len(units_of(n_real, ntn, G, order, 1, 4, 1, b, cap=80)) for b in list(range(16)) + [G - 1])
            bound = lib.fs2t_conv_gemm_p_max_units(n_real, ntn, 1, G, order)
            assert longest == bound, (n_real, ntn, order, longest, bound)
    assert lib.fs2t_conv_gemm_p_max_units(2729, 6, 1, 256, 0) == 64 and lib.fs2t_conv_gemm_p_max_units(2729, 6, 1, 256, 1) == 65
    # the largest launch both orders admit: covered exactly once
    check(2720, 6, 256, 1, 1, 4, 1)
    check(2720, 6, 256, 0, 1, 4, 1)


def test_epilogue_staging_layout_is_conflict_free_for_the_lds_service_groups():
    """fs2_tile_col128 (fs2_gemm.hip): the bf16 staging layout against the LDS service groups of MI355X_MICROARCH.md - a
    ds_read_b128 is served in four 16-lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32), bank quad = (byte / 16) mod 16;
    a ds_write_b32 in two 32-lane groups, bank = (byte / 4) mod 32.  Reader lane l of a wave: row l >> 4, columns 8 (l & 15) ..
    + 7 as two 16-byte reads; writer half-wave: 32 consecutive columns of one row.  fp32 keeps the natural layout."""
    lib = aid()
    FS2_F32, FS2_BF16 = 4, 2                                                      # element bytes
    col = [lib.fs2t_stage_tile_col(c, FS2_BF16) for c in range(128)]
    assert sorted(col) == list(range(128)) and all(col[c] % 4 == c % 4 and col[c] // 4 == col[c - c % 4] // 4 for c in range(128))
    assert [lib.fs2t_stage_tile_col(c, FS2_F32) for c in range(128)] == list(range(128))
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]
    for e in (0, 4):                                                              # the lane's two reads
        for g in groups:
            quads = {(((l >> 4) * 128 + col[8 * (l & 15) + e]) * 4 // 16) % 16 for l in g}
            assert len(quads) == 16, (e, g)
    for base in (0, 32, 64, 96):                                                  # a half-wave's ds_write_b32
        assert len({col[base + fl] % 32 for fl in range(32)}) == 32
