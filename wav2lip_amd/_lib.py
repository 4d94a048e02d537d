"""ctypes binding of libw2l_hip.so (the C ABI declared in include/w2l_hip.h).

There is no fallback: if the shared library is missing or a call fails, a RuntimeError is raised
(the exception type the reference's callers already expect from a failing GPU op, inference.py:79).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# W2L_HIP_LIB selects an alternative build of the same ABI (kernel A/B experiments); default = the in-tree build
LIB_PATH = os.environ.get("W2L_HIP_LIB") or os.path.join(_HERE, "lib", "libw2l_hip.so")

ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_LEAKY = 0, 1, 2, 3
PREC_F32, PREC_BF16 = 0, 1


class ConvGeom(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("transposed", "cin", "cout", "kh", "kw", "sh", "sw", "ph", "pw", "oph", "opw", "act")]


class AdamTensor(C.Structure):
    """w2l_adam_tensor: device pointers + element count of one parameter"""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("n", C.c_longlong)]


_vp, _i, _ll, _f = C.c_void_p, C.c_int, C.c_longlong, C.c_float

# name -> (restype, argtypes); must list every symbol of include/w2l_hip.h (tests/test_abi.py checks it)
SIGNATURES = {
    "w2l_last_error": (C.c_char_p, []),
    "w2l_abi_version": (_i, []),
    "w2l_device_count": (_i, []),
    "w2l_device_arch": (_i, [_i, C.c_char_p, C.c_size_t]),
    "w2l_conv_create": (_i, [C.POINTER(ConvGeom), _vp, _vp, _vp, _vp, C.POINTER(_vp)]),
    "w2l_conv_destroy": (_i, [_vp]),
    "w2l_conv_cin_padded": (_i, [_i]),
    "w2l_conv_out_hw": (_i, [C.POINTER(ConvGeom), _i, _i, C.POINTER(_i), C.POINTER(_i)]),
    "w2l_conv_forward": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i]),
    "w2l_conv_attach_head": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "w2l_conv_macs": (_ll, [C.POINTER(ConvGeom), _i, _i, _i]),
    "w2l_conv_set_precision": (_i, [_vp, _i]),
    "w2l_conv_set_tile": (_i, [_vp, _i]),
    "w2l_conv_num_tiles": (_i, []),
    "w2l_bn_fold": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp]),
    "w2l_nchw_to_nhwc": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _i]),
    "w2l_nhwc_to_nchw": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "w2l_datagen_pack": (_i, [_vp, _i, _i, _vp, _vp, _i, _i]),
    "w2l_frames_to_u8": (_i, [_vp, _i, _i, _i, _vp, _i, _vp]),
    "w2l_crop_resize_u8": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp, _i, _vp]),
    "w2l_resize_u8": (_i, [_vp, _i, _vp, _i, _i, _vp, _i, _i]),
    "w2l_resize_paste_u8": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i]),
    "w2l_s3fd_pack": (_i, [_vp, _ll, _vp, _vp, _i]),
    "w2l_maxpool2x2": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _i]),
    "w2l_l2norm_scale": (_i, [_vp, _ll, _i, _vp, _i, _vp, _vp, _i]),
    "w2l_s3fd_decode": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _vp, _i, _vp]),
    "w2l_s3fd_nms": (_i, [_vp, _i, _i, _vp, _f, _f, _vp, _vp, _vp, _ll]),
    "w2l_mel_create": (_i, [_vp, _vp, C.POINTER(_vp)]),
    "w2l_mel_destroy": (_i, [_vp]),
    "w2l_mel_num_frames": (_i, [_ll]),
    "w2l_melspectrogram": (_i, [_vp, _vp, _vp, _ll, _vp]),
    "w2l_mel_gather": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _i]),
    "w2l_resample_sinc": (_i, [_vp, _vp, _i, _vp, _i, C.c_double, _vp, _vp, _i, _i, _vp]),
    "w2l_l2norm_rows": (_i, [_vp, _i, _i, _vp, _i, _vp]),
    "w2l_cosine_bce": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "w2l_bce_mean": (_i, [_vp, _i, _vp, _vp, _vp]),
    "w2l_conv_update": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "w2l_conv_wgrad": (_i, [C.POINTER(ConvGeom), _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp]),
    "w2l_conv_wgrad_prec": (_i, [C.POINTER(ConvGeom), _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i]),
    "w2l_convb_create": (_i, [C.POINTER(ConvGeom), _vp, _vp, C.POINTER(_vp)]),
    "w2l_convb_update": (_i, [_vp, _vp, _vp]),
    "w2l_convb_update_many": (_i, [_i, _vp, _vp, _vp]),
    "w2l_convb_destroy": (_i, [_vp]),
    "w2l_convb_forward": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i]),
    "w2l_convb_forward_bn": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "w2l_convb_forward_bnbwd": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                                     C.POINTER(C.c_int)]),
    "w2l_convb_forward_actbwd": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _vp, C.POINTER(C.c_int)]),
    "w2l_convb_set_tile": (_i, [_vp, _i]),
    "w2l_convb_num_tiles": (_i, []),
    "w2l_conv_wgrad_bf16": (_i, [C.POINTER(ConvGeom), _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp]),
    "w2l_bn_train_stats_bf16": (_i, [_vp, _ll, _i, _i, _vp, _i, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "w2l_affine_act_bf16": (_i, [_vp, _ll, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _i]),
    "w2l_bn_train_bwd_bf16": (_i, [_vp, _ll, _i, _i, _vp, _i, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i]),
    "w2l_bn_train_bwd_apply_bf16": (_i, [_vp, _ll, _i, _vp, _i, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i]),
    "w2l_act_bwd_bf16": (_i, [_vp, _ll, _i, _vp, _i, _vp, _i, _i, _vp, _vp, _i, _vp, _i]),
    "w2l_add_rows_bf16": (_i, [_vp, _ll, _i, _vp, _i, _vp, _i, _vp, _i]),
    "w2l_col_sum_bf16": (_i, [_vp, _ll, _i, _vp, _i, _vp]),
    "w2l_thin1x1_forward_bf16": (_i, [_vp, _ll, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _i]),
    "w2l_thin1x1_dgrad_bf16": (_i, [_vp, _ll, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _i]),
    "w2l_thin1x1_wgrad_bf16": (_i, [_vp, _ll, _i, _i, _vp, _i, _vp, _i, _vp, _vp]),
    "w2l_nchw_to_nhwc_bf16": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _i]),
    "w2l_nhwc_bf16_to_nchw": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "w2l_bn_train_stats": (_i, [_vp, _ll, _i, _vp, _i, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "w2l_affine_act": (_i, [_vp, _ll, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _i]),
    "w2l_bn_train_bwd": (_i, [_vp, _ll, _i, _vp, _i, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i]),
    "w2l_act_bwd": (_i, [_vp, _ll, _i, _vp, _i, _vp, _i, _i, _vp, _vp, _i, _vp, _i]),
    "w2l_add_rows": (_i, [_vp, _ll, _i, _vp, _i, _vp, _i, _vp, _i]),
    "w2l_col_sum": (_i, [_vp, _ll, _i, _vp, _i, _vp]),
    "w2l_l1_mean": (_i, [_vp, _ll, _vp, _vp, _vp]),
    "w2l_l1_bwd": (_i, [_vp, _ll, _vp, _vp, _vp, _vp]),
    "w2l_cosine_bce_bwd": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "w2l_l2norm_bwd": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _i]),
    "w2l_bce_bwd": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "w2l_shifted_pdist": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "w2l_adam_create": (_i, [_i, C.POINTER(_ll), C.POINTER(_vp)]),
    "w2l_adam_destroy": (_i, [_vp]),
    "w2l_adam_step": (_i, [_vp, _vp, C.POINTER(AdamTensor), _f, _f, _f, _f, _f, _i]),
    "w2l_plan_create": (_i, [C.POINTER(_vp)]),
    "w2l_plan_destroy": (_i, [_vp]),
    "w2l_plan_add_conv": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i]),
    "w2l_plan_copy_item": (_i, [_vp, _vp, _i]),
    "w2l_plan_run": (_i, [_vp, _vp]),
    "w2l_plan_size": (_i, [_vp]),
    "w2l_plan_autotune": (_i, [_vp, _vp, _i]),
    "w2l_plan_get_config": (_i, [_vp, _i, C.POINTER(_i), C.POINTER(_i)]),
    "w2l_plan_set_config": (_i, [_vp, _i, _i, _i]),
    "w2l_plan_profile": (_i, [_vp, _vp, _i, _vp]),
    "w2l_plan_executed_flops": (_i, [_vp, C.POINTER(_ll), C.POINTER(_i)]),
    "w2l_conv_num_igemm_tiles": (_i, []),
    "w2l_flops_begin": (_i, []),
    "w2l_flops_end": (_ll, [C.POINTER(_ll)]),
    "w2l_clock_probe": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "w2l_igemm_block_order": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "w2l_conv_config_family": (_i, [_i]),
    "w2l_conv_exclude_families": (_i, [_i]),
    "w2l_tune_key_ints": (_i, []),
    "w2l_tune_set": (_i, [C.POINTER(_i), _i, _i]),
    "w2l_tune_entry_applicable": (_i, [C.POINTER(_i), _i]),
    "w2l_tune_clear": (_i, []),
    "w2l_tune_count": (_i, []),
    "w2l_tune_export": (_i, [C.POINTER(_i), _i]),
}

_lib = None

# Shape-keyed launch configurations (include/w2l_hip.h, "tune table"): the committed file is loaded into the library once per
# process, so that a layer's (tile, split-K) - and with it its summation order and every bit of its output - is a function of
# its shape alone.  W2L_TUNE_TABLE=<path> selects another file, W2L_TUNE_TABLE=0 none (heuristic configurations only).
TUNE_TABLE_PATH = os.path.join(_HERE, "tune_table.json")
# W2L_EXACT=1 (bench.py --exact): the launch table tuned WITHOUT the F(4x4,3x3) Winograd kernel, and that family switched off
# in the library: F(2x2) / implicit-GEMM launches only, about half the rounding error of the default table at a measured cost
# in frames/s.  Both tables are functions of the shape: either mode is bit-reproducible on its own.
EXACT = os.environ.get("W2L_EXACT", "0") == "1"
EXACT_TABLE_PATH = os.path.join(_HERE, "tune_table_exact.json")
FAMILY_WINO4 = 4
FAMILY_SPLIT = 5     # conv_igemm_bf16_kernel<.., 3>: fp32 operands as three bf16 pieces on the bf16 matrix cores
FAMILY_WINO2S = 6    # conv_wino2s_kernel: F(2x2,3x3) Winograd with the transformed operands as three bf16 pieces
FAMILY_TP2S = 7      # conv_tp2s_kernel: the fused-phase stride-2 transposed kernel with the operands as three bf16 pieces
FAMILY_STEM7S = 8    # conv_stem7s_kernel: the 7x7 first layer, input region staged and split once, contraction out of LDS
FAMILY_K3S = 9       # conv_k3s_kernel: direct 3x3 for 32-cout layers (+ fused 1x1 head), input block staged and split once per K-step


# W2L_SPLIT=0 switches the split-operand implicit GEMM (family 5: fp32-accurate results on the bf16 matrix cores) off AND
# puts the tuned fp32-pipe entries back for the launches the committed table gives to that family (tune_table_nosplit.json), so
# that the switch is an A/B against the tuned fp32 kernels, not against the heuristic.
NO_SPLIT = os.environ.get("W2L_SPLIT", "1") == "0"
NOSPLIT_TABLE_PATH = os.path.join(_HERE, "tune_table_nosplit.json")


def load_tune_table(lib, path=None):
    """push the entries of a tune-table JSON file into the library; returns the number of entries loaded"""
    import json
    if path is None and NO_SPLIT and not EXACT and not os.environ.get("W2L_TUNE_TABLE"):
        return load_tune_table(lib, TUNE_TABLE_PATH) + load_tune_table(lib, NOSPLIT_TABLE_PATH)
    path = path or os.environ.get("W2L_TUNE_TABLE") or (EXACT_TABLE_PATH if EXACT else TUNE_TABLE_PATH)
    if path == "0" or not os.path.exists(path):
        return 0
    with open(path) as fh:
        doc = json.load(fh)
    nk = lib.w2l_tune_key_ints()
    if doc.get("key_ints") != nk or doc.get("num_configs", 0) > lib.w2l_conv_num_tiles():
        raise RuntimeError("tune table %s was written for another library build (key_ints %s / configs %s)"
                           % (path, doc.get("key_ints"), doc.get("num_configs")))
    n = 0
    for e in doc["entries"]:
        key = (C.c_int * nk)(*e[:nk])
        check_rc = lib.w2l_tune_set(key, int(e[nk]), int(e[nk + 1]))
        if check_rc != 0:
            raise RuntimeError("tune table %s: bad entry %s" % (path, e))
        n += 1
    return n


def export_tune_table(lib):
    """entries currently in the library's tune table as lists of ints (key..., tile, ksplit), sorted"""
    nk = lib.w2l_tune_key_ints()
    n = lib.w2l_tune_count()
    buf = (C.c_int * (max(n, 1) * (nk + 2)))()
    n = lib.w2l_tune_export(buf, n)
    return sorted([int(buf[i * (nk + 2) + j]) for j in range(nk + 2)] for i in range(n))


def save_tune_table(lib, path, note=""):
    import json
    nk = lib.w2l_tune_key_ints()
    for e in export_tune_table(lib):
        if not lib.w2l_tune_entry_applicable((C.c_int * nk)(*e[:nk]), e[nk]):
            raise RuntimeError("tune table entry %s names a configuration its shape cannot run" % e)
    doc = {"key_ints": lib.w2l_tune_key_ints(), "num_configs": lib.w2l_conv_num_tiles(),
           "key": "transposed cin cout kh kw sh sw ph pw oph opw precision has_residual head_c N H W -> config ksplit",
           "note": note, "entries": export_tune_table(lib)}
    with open(path, "w") as fh:
        fh.write(json.dumps(doc, indent=None, separators=(",", ":")).replace("],[", "],\n[") + "\n")
    return len(doc["entries"])


def load():
    """Load libw2l_hip.so (built in-tree by __graft_entry__.build() / `make -C wav2lip_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  -- torch's bundled HIP runtime (libamdhip64.so.7) must be the one this process binds:
    # the library shares streams and device pointers with torch, so both must sit on the same runtime instance
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "wav2lip_amd: HIP library %s is missing; build it with `make -C wav2lip_amd/csrc` "
            "(there is no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    # W2L_EXACT=1: no F(4x4) Winograd.  W2L_SPLIT=0: no split-operand implicit GEMM (see NO_SPLIT above).
    mask = ((1 << FAMILY_WINO4) if EXACT else 0) | (((1 << FAMILY_SPLIT) | (1 << FAMILY_WINO2S) | (1 << FAMILY_TP2S) | (1 << FAMILY_STEM7S) | (1 << FAMILY_K3S)) if NO_SPLIT else 0)
    if mask and lib.w2l_conv_exclude_families(mask) != 0:
        raise RuntimeError("wav2lip_amd: could not switch kernel families off (mask %d)" % mask)
    load_tune_table(lib)
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().w2l_last_error()
        raise RuntimeError("libw2l_hip %s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


def ptr(t):
    """device pointer of a torch tensor (or None)"""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
