"""Index invariants the banded LDL^T KKT kernel relies on (contactimplicitmpc/jl_amd/csrc/kkt_dense.hip: kkt_banded_kernel, round 4),
restated in Python and checked exhaustively over the sizes the kernel accepts (w <= 190).  These are properties of the SCHEME -
the formulas below are copied from the kernel's comments and code, the device results themselves are checked in the GPU tests
(test_gpu_parity.py::test_kkt_dense_lu_backend, test_gpu_round4.py::test_banded_kernel_forms_agree):

* back substitution: the pending sum of column j lives in lane (j + 192) % 64, register ((j + 192) / 64) % 3 of one wavefront; the rows
  of L are staged in exactly that layout.  A band of w < 192 columns never puts two columns of one lane into one register, every band
  entry of a row is staged exactly once, and the row's own column is never among them;
* update tiles: the table lists every tile (I, J), J <= I, of the lower triangle once, tile (0, 0) first and tile column 0 before the rest;
* entering rows: the RB (w + 2) entries of a block are spread over the threads from 128 on, at most two each, each entry once;
* tiles of the bulk: the rank of a wavefront in the round-robin is a permutation of 0 .. 14 with the wavefronts of SIMD 0 last."""
import numpy as np
import pytest


def staged(i, w, sl, ln):
    """(valid, offset c in the row of L) of what lane ln keeps in register sl while row i is processed (kkt_banded_kernel: stage())"""
    bb = i - w + 192
    blk = bb >> 6
    dB = (sl - blk % 3) % 3
    jj = 64 * (blk + dB) + ln
    if jj < bb:
        jj += 192
    return (jj >= 192 and jj < i + 192), jj - bb, jj


@pytest.mark.parametrize("w", [1, 5, 27, 63, 64, 65, 81, 107, 127, 128, 131, 189, 190])
def test_back_substitution_register_layout(w):
    N = max(3 * w + 7, 200)
    for i in list(range(0, min(N, w + 70))) + list(range(N - 70, N)):
        cols = {}
        for sl in range(3):
            for ln in range(64):
                ok, c, jj = staged(i, w, sl, ln)
                if not ok:
                    continue
                j = jj - 192
                assert 0 <= c < w and c == j - (i - w)                       # offset in the stored row of L
                assert jj % 64 == ln and (jj >> 6) % 3 == sl                  # the column's owner, whatever the row
                assert (sl, ln) not in cols
                cols[(sl, ln)] = j
        assert sorted(cols.values()) == list(range(max(0, i - w), i))        # the band of row i, each column once
        own = ((i + 192) & 63, ((i + 192) >> 6) % 3)                          # where x_i's sum is read off: not a band entry of row i
        assert (own[1], own[0]) not in cols


def tile_table(mt):
    nT = (mt + 15) >> 4
    ntiles = nT * (nT + 1) // 2
    tab = []
    for tid in range(ntiles):
        I, J = tid, 0
        if tid >= nT:
            t2 = tid - nT
            I = 0
            while (I + 1) * (I + 2) // 2 <= t2:
                I += 1
            J = t2 - I * (I + 1) // 2 + 1
            I += 1
        tab.append((I, J))
    return nT, tab


@pytest.mark.parametrize("mt", [1, 15, 16, 17, 64, 107, 128, 131, 190])
def test_tile_table_lists_the_lower_triangle_once(mt):
    nT, tab = tile_table(mt)
    assert len(tab) <= 96                                                     # the table's room in LDS
    assert tab[0] == (0, 0) and [t for t in tab[:nT]] == [(I, 0) for I in range(nT)]
    assert sorted(tab) == sorted((I, J) for I in range(nT) for J in range(I + 1))


@pytest.mark.parametrize("RB,w", [(8, 1), (8, 27), (8, 107), (8, 126), (4, 131), (4, 190)])
def test_entering_entries_are_spread_once(RB, w):
    nt, NE, EW = 1024, 2, w + 2
    seen = {}
    for tid in range(nt):
        for n in range(NE):
            idx = tid - 128 + n * (nt - 128)
            if tid >= 128 and idx < RB * EW:
                key = (idx // EW, idx % EW)
                assert key not in seen
                seen[key] = tid
    assert sorted(seen) == [(t, e) for t in range(RB) for e in range(EW)]
    need2 = RB * EW > nt - 128                                                # the second entry per thread exists only then
    assert need2 == any(tid - 128 + (nt - 128) < RB * EW for tid in range(128, nt))


def test_bulk_rank_is_a_permutation_with_simd0_last():
    nty = 16
    rank = {wv: ((wv >> 2) * 3 + (wv & 3) - 1) if (wv & 3) != 0 else (nty - nty // 4) + (wv >> 2) - 1 for wv in range(1, nty)}
    assert sorted(rank.values()) == list(range(nty - 1))
    assert sorted(wv for wv, r in rank.items() if r >= 12) == [4, 8, 12]      # wave v runs on SIMD v % 4: wavefront 0's SIMD comes last


@pytest.mark.parametrize("RB,w", [(8, 107), (8, 126), (4, 131), (4, 190)])
def test_p2_covers_the_rows_below_a_block(RB, w):
    """P2: 16 / RB rows per 16 lanes of 1024 threads, row q = RB + (tid >> 4) (16 / RB) + (tid & 15) / RB"""
    NG = 16 // RB
    rows = {RB + (tid >> 4) * NG + (tid & 15) // RB for tid in range(1024)}
    assert set(range(RB, RB + w)) <= rows
