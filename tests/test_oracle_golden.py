"""CPU suite: the C oracle (oracle/jps_oracle.c) against the golden vectors captured from the real
jps1.py (tests/golden/make_golden.py), and the synthetic-input generators against each other."""
import hashlib

import numpy as np
import pytest

from conftest import grid_from_bits, load_golden, pairs


def check(oracle, grid, rec, literal=True):
    path, cost, st = oracle.plan(grid, rec["start"], rec["goal"], rec["hchoice"], literal=literal)
    if rec["path"] is None:
        assert path == 0
    else:
        assert path == pairs(rec["path"])
        if rec["start"] != rec["goal"]:
            assert cost.hex() == rec["cost_hex"]            # bit-exact float64
            assert repr(cost) == rec["printed"]             # what jps1.py:207 prints
        else:
            assert rec["printed"] == "0" and cost == 0.0
    if literal and "cells" in rec:
        assert (st["cells"], st["pushes"], st["pops"]) == (rec["cells"], rec["pushes"], rec["pops"])


def test_known_answers(oracle):
    recs = load_golden("known_answers.json")
    assert len(recs) >= 15
    for rec in recs:
        grid = np.array(rec["grid"], dtype=np.float64).reshape(rec["shape"])
        check(oracle, grid, rec)
        check(oracle, grid, rec, literal=False)


def test_start_out_of_bounds_is_an_error(oracle):
    with pytest.raises(ValueError):
        oracle.plan(np.zeros((5, 5)), (5, 5), (1, 1))
    with pytest.raises(ValueError):
        oracle.plan(np.zeros((5, 5)), (-1, 0), (1, 1))
    with pytest.raises(ValueError):
        oracle.plan(np.zeros((5, 5)), (0, 0), (1, 1), hchoice=3)


def test_random_small(oracle):
    recs = load_golden("random_small.json")
    assert len(recs) == 400
    for rec in recs:
        grid = grid_from_bits(rec["grid_bits"], rec["shape"])
        check(oracle, grid, rec)
        check(oracle, grid, rec, literal=False)


def test_reference_maps(oracle, map_grids):
    recs = load_golden("maps_png.json")
    assert len({r["map"] for r in recs}) == 35
    for rec in recs:
        bits = np.unpackbits(map_grids[rec["map"]])
        if "canvas" in rec:
            # BASELINE config 1: the 147x112 snapshot pasted into a zero 256x256 canvas
            occ = np.zeros((256, 256), dtype=np.uint8)
            occ[:147, :112] = bits[:147 * 112].reshape(147, 112)
        else:
            W, H = rec["shape"]
            occ = bits[:W * H].reshape(W, H)
        assert hashlib.sha256(np.ascontiguousarray(occ, dtype=np.uint8).tobytes()).hexdigest() == rec["sha256"]
        check(oracle, occ, rec)


def test_config1_vector_from_survey(oracle, map_grids):
    """(0,0)->(146,111) on the canvas: 37 points, cost 195.49242404917496."""
    rec = [r for r in load_golden("maps_png.json") if "canvas" in r and r["start"] == [0, 0] and r["goal"] == [146, 111]][0]
    assert len(rec["path"]) // 2 == 37 and rec["printed"] == "195.49242404917496"


def test_synth1024(oracle):
    g = load_golden("synth1024.json")
    occ = oracle.synth_grid(g["W"], g["H"], g["grid_seed"], g["p"])
    assert hashlib.sha256(occ.tobytes()).hexdigest() == g["grid_sha256"]
    s, t = oracle.synth_queries(occ, g["qseed"], len(g["queries"]))
    assert hashlib.sha256(s.tobytes()).hexdigest() == g["starts_sha256"]
    assert hashlib.sha256(t.tobytes()).hexdigest() == g["goals_sha256"]
    for i, rec in enumerate(g["queries"]):
        assert rec["start"] == list(map(int, s[i])) and rec["goal"] == list(map(int, t[i]))
        check(oracle, occ, rec)


def test_batch_api_matches_single(oracle):
    occ = oracle.synth_grid(96, 80, 7, 0.25)
    s, t = oracle.synth_queries(occ, 3, 200)
    for lit in (False, True):
        cells, ln, cost, st = oracle.plan_batch(occ, s, t, 2, literal=lit, max_len=512, nthreads=4, want_stats=True)
        for q in range(0, 200, 17):
            path, c, _ = oracle.plan(occ, s[q], t[q], 2, literal=lit)
            if path == 0:
                assert ln[q] == 0
            else:
                assert [tuple(x) for x in cells[q, :ln[q]]] == path and cost[q] == c


def test_generators_agree(oracle):
    from fuxi_planner_amd import synth
    for W, H, seed, p in [(1024, 1024, 1, 0.2), (100, 37, 5, 0.33), (3, 3, 9, 0.5)]:
        a = synth.synth_grid(W, H, seed, p)
        b = oracle.synth_grid(W, H, seed, p)
        assert np.array_equal(a, b)
        if (a == 0).sum() >= 2:
            sa, ga = synth.synth_queries(a, seed + 1, 300, first=12345)
            sb, gb = oracle.synth_queries(b, seed + 1, 300, first=12345)
            assert np.array_equal(sa, sb) and np.array_equal(ga, gb)
            assert (a[sa[:, 0], sa[:, 1]] == 0).all() and (a[ga[:, 0], ga[:, 1]] == 0).all()
            assert ((sa != ga).any(axis=1)).all()


def test_splitmix_known_values(oracle):
    from fuxi_planner_amd import synth
    L = oracle.lib()
    for x in (0, 1, 2, 0xDEADBEEF, 2**64 - 1):
        assert int(synth.splitmix64(np.array([x], dtype=np.uint64))[0]) == L.fxo_splitmix64(x)
    assert L.fxo_splitmix64(0) == 0xE220A8397B1DCDAF  # published splitmix64 test vector
