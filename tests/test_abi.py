"""The C-ABI library loads (no GPU needed) and exports every symbol include/w2l_hip.h declares."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "w2l_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(w2l_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ("w2l_conv_create", "w2l_conv_forward", "w2l_melspectrogram", "w2l_datagen_pack", "w2l_plan_run",
                 "w2l_frames_to_u8", "w2l_mel_gather", "w2l_l2norm_rows", "w2l_cosine_bce", "w2l_bn_fold"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from wav2lip_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build with `make -C wav2lip_amd/csrc` or __graft_entry__.build()"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_ctypes_signatures_cover_the_header():
    from wav2lip_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    lib = _lib.load()
    assert lib.w2l_abi_version() == 1
    assert lib.w2l_conv_cin_padded(6) == 8 and lib.w2l_conv_cin_padded(1) == 4
    assert lib.w2l_mel_num_frames(48000) == 241


def test_argument_errors_are_reported_not_thrown():
    from wav2lip_amd import _lib
    lib = _lib.load()
    g = _lib.ConvGeom(0, 6, 16, 9, 9, 1, 1, 3, 3, 0, 0, 1)   # 9x9 kernel: unsupported
    ho, wo = ctypes.c_int(), ctypes.c_int()
    rc = lib.w2l_conv_out_hw(ctypes.byref(g), 96, 96, ctypes.byref(ho), ctypes.byref(wo))
    assert rc == -1 and b"unsupported" in lib.w2l_last_error()
    g = _lib.ConvGeom(0, 6, 16, 7, 7, 1, 1, 3, 3, 0, 0, 1)
    assert lib.w2l_conv_out_hw(ctypes.byref(g), 96, 96, ctypes.byref(ho), ctypes.byref(wo)) == 0
    assert (ho.value, wo.value) == (96, 96)
    assert lib.w2l_conv_macs(ctypes.byref(g), 1, 96, 96) == 96 * 96 * 6 * 16 * 49
    gt = _lib.ConvGeom(1, 1024, 512, 3, 3, 2, 2, 1, 1, 1, 1, 1)
    assert lib.w2l_conv_out_hw(ctypes.byref(gt), 3, 3, ctypes.byref(ho), ctypes.byref(wo)) == 0
    assert (ho.value, wo.value) == (6, 6)


def test_committed_tune_table_loads_into_this_library_build():
    """wav2lip_amd/tune_table.json (the committed launch configurations) belongs to THIS library build: key width
    and configuration-id range match, every entry is accepted by w2l_tune_set, the export returns the same entries, every id
    maps to a kernel family, and keys are unique (a shape has exactly one configuration - bit-reproducibility rests on it)"""
    import json
    from wav2lip_amd import _lib
    lib = _lib.load()
    doc = json.load(open(_lib.TUNE_TABLE_PATH))
    nk = lib.w2l_tune_key_ints()
    # configuration ids are append-only: a table written before a family was added stays valid in a later build
    assert doc["key_ints"] == nk == 17 and 13 <= doc["num_configs"] <= lib.w2l_conv_num_tiles()
    entries = [list(map(int, e)) for e in doc["entries"]]
    assert len(entries) > 500 and all(len(e) == nk + 2 for e in entries)
    assert len({tuple(e[:nk]) for e in entries}) == len(entries), "duplicate shape keys"
    for e in entries:
        assert 0 <= e[nk] < lib.w2l_conv_num_tiles() and 1 <= e[nk + 1] <= 64, e
        assert lib.w2l_conv_config_family(e[nk]) in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9), e
        # the recorded id is one the shape can actually run (round 2's table held ids that fell through to the heuristic at
        # launch time; tools/resolve_tune_table.py rewrote them as what they resolve to)
        assert lib.w2l_tune_entry_applicable((ctypes.c_int * nk)(*e[:nk]), e[nk]) == 1, e
    assert lib.w2l_conv_config_family(lib.w2l_conv_num_tiles()) == -1
    lib.w2l_tune_clear()
    assert _lib.load_tune_table(lib) == len(entries) == lib.w2l_tune_count()
    assert _lib.export_tune_table(lib) == sorted(entries)


def test_split_operand_entries_of_the_committed_table_and_their_fp32_predecessors():
    """the committed table names the split-operand families (5: implicit GEMM, 6: F(2x2) Winograd, 7: fused-phase transposed) for
    generator-inference launches of the tuned batch sizes; tune_table_nosplit.json (W2L_SPLIT=0) holds an fp32-pipe entry for exactly those keys, and loading it
    on top changes exactly those entries"""
    import ctypes as C
    import json
    from wav2lip_amd import _lib
    lib = _lib.load()
    nk = lib.w2l_tune_key_ints()
    base = {tuple(e[:nk]): tuple(e[nk:]) for e in json.load(open(_lib.TUNE_TABLE_PATH))["entries"]}
    split_keys = {k for k, v in base.items() if lib.w2l_conv_config_family(v[0]) in (_lib.FAMILY_SPLIT, _lib.FAMILY_WINO2S, _lib.FAMILY_TP2S, _lib.FAMILY_STEM7S, _lib.FAMILY_K3S)}
    assert 8 <= len(split_keys) <= 128 and all(k[11] == 0 and k[14] in (1, 8, 16, 32, 64, 128, 256) for k in split_keys)    # fp32 layers, tuned batches
    doc = json.load(open(_lib.NOSPLIT_TABLE_PATH))
    assert doc["key_ints"] == nk and {tuple(e[:nk]) for e in doc["entries"]} == split_keys
    for e in doc["entries"]:
        assert lib.w2l_conv_config_family(e[nk]) in (0, 1, 2, 3, 4) and 1 <= e[nk + 1] <= 64, e
        assert lib.w2l_tune_entry_applicable((C.c_int * nk)(*e[:nk]), e[nk]) == 1, e
    try:
        lib.w2l_tune_clear()
        n = _lib.load_tune_table(lib, _lib.TUNE_TABLE_PATH) + _lib.load_tune_table(lib, _lib.NOSPLIT_TABLE_PATH)
        assert n == len(base) + len(split_keys) and lib.w2l_tune_count() == len(base)
        now = {tuple(e[:nk]): tuple(e[nk:]) for e in _lib.export_tune_table(lib)}
        assert {k for k in base if now[k] != base[k]} == split_keys
    finally:
        lib.w2l_tune_clear()
        _lib.load_tune_table(lib)


def test_committed_plan_lists_are_well_formed_and_selected_by_batch_size(monkeypatch):
    """wav2lip_amd/plan_configs.json (per-plan launch lists of generator inference for the batch sizes the tune table does not
    hold, engine.apply_plan_configs): one list per listed batch size over the SAME launch names, ids and split-K inside this
    library's ranges, batches 2..7 tuned on their own, the table's batches listed but never applied; any other batch size
    borrows the next listed one's; a list naming other launches than the plan is an error, W2L_PLAN_CONFIGS=0 switches it off"""
    from wav2lip_amd import _lib, engine
    lib = _lib.load()
    doc = engine.load_plan_configs()
    assert set(doc) == {"generator_96"}
    d = doc["generator_96"]
    assert d["table"] == [1, 8, 16, 32, 64, 128, 256] and set(d["table"]) <= set(d["plans"]) and set(range(2, 8)) <= set(d["plans"])
    names = [e[0] for e in d["plans"][1]]
    assert len(names) == len(set(names)) >= 50
    for b, lst in d["plans"].items():
        assert [e[0] for e in lst] == names, b
        for _, c, k in lst:
            assert 0 <= c < lib.w2l_conv_num_tiles() and 1 <= k <= 64, (b, c, k)
            assert lib.w2l_conv_config_family(c) in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9)
    want = {1: None, 2: 2, 7: 7, 8: None, 9: 16, 37: 64, 100: 128, 128: None, 200: 256, 256: None, 700: 256}
    assert {n: engine.plan_config_source("generator_96", n) for n in want} == want
    assert engine.plan_config_source("no_such_plan", 3) is None

    class FakePlan:
        def __init__(self, names):
            self.records, self.calls = [(n,) for n in names], []

        def set_config(self, i, tile, ksplit):
            self.calls.append((i, tile, ksplit))

    for var in ("W2L_PLAN_CONFIGS", "W2L_TUNE_TABLE"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setattr(engine, "AUTOTUNE", False)
    monkeypatch.setattr(_lib, "EXACT", False)
    p = FakePlan(names)
    assert engine.apply_plan_configs(p, "generator_96", 5) == 5
    assert p.calls == [(i, c, k) for i, (_, c, k) in enumerate(d["plans"][5])]
    p = FakePlan(names)
    assert engine.apply_plan_configs(p, "generator_96", 128) is None and p.calls == []     # the BASELINE batch: the table, as before
    with pytest.raises(RuntimeError, match="plan_configs.json"):       # tests / tools: a stale list is an error
        engine.apply_plan_configs(FakePlan(names[:-1]), "generator_96", 5, strict=True)
    stale = FakePlan(names[:-1])                                        # inference: one warning, then the table / heuristic
    monkeypatch.setattr(engine, "_PLAN_MISMATCH_WARNED", [False])
    with pytest.warns(UserWarning, match="plan_configs.json"):
        assert engine.apply_plan_configs(stale, "generator_96", 5) is None and stale.calls == []
    monkeypatch.setenv("W2L_PLAN_CONFIGS", "0")
    p = FakePlan(names)
    assert engine.apply_plan_configs(p, "generator_96", 5) is None and p.calls == []
    monkeypatch.delenv("W2L_PLAN_CONFIGS")
    monkeypatch.setattr(engine, "AUTOTUNE", True)       # stopwatch tuning and the exact table own their plans
    assert engine.apply_plan_configs(FakePlan(names), "generator_96", 5) is None


def test_one_in_flight_table_is_loadable_and_differs_from_the_default_only_by_split_operand_entries():
    """wav2lip_amd/tune_table_one_in_flight.json (W2L_TUNE_TABLE=<it>: the launch choices that are faster with ONE batch in flight,
    profiles/r05/k_*): same keys as the committed table, every entry runnable, and every entry that differs names a split-operand
    configuration (family 5 or 6) on a generator-inference shape"""
    import json
    from wav2lip_amd import _lib
    lib = _lib.load()
    nk = lib.w2l_tune_key_ints()
    base = {tuple(e[:nk]): tuple(e[nk:]) for e in json.load(open(_lib.TUNE_TABLE_PATH))["entries"]}
    doc = json.load(open(os.path.join(os.path.dirname(_lib.TUNE_TABLE_PATH), "tune_table_one_in_flight.json")))
    alt = {tuple(e[:nk]): tuple(e[nk:]) for e in doc["entries"]}
    assert doc["key_ints"] == nk and doc["num_configs"] <= lib.w2l_conv_num_tiles() and set(alt) == set(base)
    diff = [k for k in alt if alt[k] != base[k]]
    assert 20 <= len(diff) <= 120
    for k in diff:
        assert lib.w2l_conv_config_family(alt[k][0]) in (_lib.FAMILY_SPLIT, _lib.FAMILY_WINO2S, _lib.FAMILY_TP2S, _lib.FAMILY_STEM7S, _lib.FAMILY_K3S) and k[11] == 0 and k[14] in (1, 8, 16, 32, 64, 128, 256), k
        assert lib.w2l_tune_entry_applicable((ctypes.c_int * nk)(*k), alt[k][0]) == 1, k
    assert any(lib.w2l_conv_config_family(alt[k][0]) == _lib.FAMILY_WINO2S for k in diff)


def test_every_workgroup_order_of_the_implicit_gemm_is_a_bijection_and_keeps_its_promises():
    """w2l_igemm_block_order = the kernels' own workgroup -> (phase, M-tile, cout-tile) decode, run on the host: every order visits
    every tile exactly once for ragged grids too (a tile computed twice or never is silent corruption); the phase-blocked order
    puts the phases of one group of M-tiles on ONE XCD back to back (that is where its L2 reuse comes from) and keeps workgroups
    that are dispatched together in the same phase (phases running side by side measured 30-40 % slower); the cout-slowest order
    gives every XCD at most two cout-tiles' weights per phase and stays phase-major"""
    import numpy as np
    from wav2lip_amd import _lib
    lib = _lib.load()

    def order(o, r, tm, tn, ny):
        out = np.empty((tm * tn * ny, 3), dtype=np.int32)
        assert lib.w2l_igemm_block_order(o, r, tm, tn, ny, out.ctypes.data_as(ctypes.c_void_p)) == 0
        return out

    for tm in (1, 3, 18, 37, 288):
        for tn in (1, 2, 3, 4):
            for ny in (1, 4, 9):
                for o, r in ((0, 1), (1, 1), (2, 1), (2, 5), (2, 8), (2, 32)):
                    got = order(o, r, tm, tn, ny)
                    assert got.min() >= 0 and (got.max(axis=0) == (ny - 1, tm - 1, tn - 1)).all()
                    assert len({tuple(t) for t in got.tolist()}) == tm * tn * ny, (o, r, tm, tn, ny)
    # 512 -> 128 transposed at 24x24, 128 frames: 1152 M-tiles of 64 rows, one cout-tile, four phases, groups of 32
    got = order(2, 32, 1152, 1, 4)
    xcd = np.arange(len(got)) % 8
    for x in range(8):
        mine = got[xcd == x]                      # in the order this XCD receives them
        first = {}
        for i, (ph, m, _) in enumerate(mine.tolist()):
            first.setdefault((m // 32, ph), i)
        groups = sorted({g for g, _ in first})
        whole = [g for g in groups if all((g, ph) in first for ph in range(4))]
        assert len(whole) >= len(groups) - 2      # only the groups cut by the XCD's range ends are shared with a neighbour
        for g in whole:
            starts = [first[(g, ph)] for ph in range(4)]
            assert starts == sorted(starts) and starts[3] - starts[0] <= 3 * 32
        for i in range(0, len(mine) - 32, 32):    # a round of 32 workgroups (one per CU of the XCD) holds at most two phases
            assert len(set(mine[i:i + 32, 0].tolist())) <= 2
    # weight-heavy layers (1024 -> 512 transposed at 3x3; 512 -> 512 at 3x3)
    for tm, tn, ny in ((18, 4, 4), (36, 3, 4), (18, 8, 1)):
        got = order(1, 1, tm, tn, ny)
        assert (got[:, 0] == np.repeat(np.arange(ny), tm * tn)).all()                    # phase-major in dispatch order
        xcd = np.arange(len(got)) % 8
        for x in range(8):
            for ph in range(ny):
                assert len(set(got[(xcd == x) & (got[:, 0] == ph)][:, 2].tolist())) <= 2
    assert lib.w2l_igemm_block_order(3, 1, 4, 4, 1, None) != 0
