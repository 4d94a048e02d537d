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
