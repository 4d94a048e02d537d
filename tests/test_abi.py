"""The C-ABI library loads (no GPU needed) and exports every symbol include/w2l_hip.h declares."""
import ctypes
import os
import re

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
    assert doc["key_ints"] == nk == 17 and doc["num_configs"] == lib.w2l_conv_num_tiles()
    entries = [list(map(int, e)) for e in doc["entries"]]
    assert len(entries) > 500 and all(len(e) == nk + 2 for e in entries)
    assert len({tuple(e[:nk]) for e in entries}) == len(entries), "duplicate shape keys"
    for e in entries:
        assert 0 <= e[nk] < lib.w2l_conv_num_tiles() and 1 <= e[nk + 1] <= 64, e
        assert lib.w2l_conv_config_family(e[nk]) in (0, 1, 2, 3, 4), e
        # the recorded id is one the shape can actually run (round 2's table held ids that fell through to the heuristic at
        # launch time; tools/resolve_tune_table.py rewrote them as what they resolve to)
        assert lib.w2l_tune_entry_applicable((ctypes.c_int * nk)(*e[:nk]), e[nk]) == 1, e
    assert lib.w2l_conv_config_family(lib.w2l_conv_num_tiles()) == -1
    lib.w2l_tune_clear()
    assert _lib.load_tune_table(lib) == len(entries) == lib.w2l_tune_count()
    assert _lib.export_tune_table(lib) == sorted(entries)
