"""ctypes loader for libvmasr_hip.so (C ABI: include/vmasr_hip.h).

The library is built in-tree by `make -C vm_asr_amd/csrc` (see __graft_entry__.build).
Missing library == hard error; there is deliberately no fallback path.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VMASR_LIB") or os.path.join(_HERE, "libvmasr_hip.so")   # VMASR_LIB: A/B builds (dev)

c_i32, c_i64, c_vp, c_sz = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_size_t

F32, F16, BF16 = 0, 1, 2


class SScanParams(ctypes.Structure):
    """POD mirror of vmasr_sscan_params (include/vmasr_hip.h)."""
    _fields_ = (
        [(n, c_i32) for n in ("batch", "dim", "seqlen", "dstate", "n_groups", "n_chunks", "dtype",
                              "delta_softplus")]
        + [(n, c_i64) for n in ("A_d_stride", "A_dstate_stride",
                                "B_batch_stride", "B_group_stride", "B_dstate_stride",
                                "C_batch_stride", "C_group_stride", "C_dstate_stride",
                                "u_batch_stride", "u_d_stride", "delta_batch_stride", "delta_d_stride",
                                "out_batch_stride", "out_d_stride")]
        + [(n, c_vp) for n in ("A_ptr", "B_ptr", "C_ptr", "D_ptr", "u_ptr", "delta_ptr",
                               "delta_bias_ptr", "out_ptr", "x_ptr")]
    )


class SScanBwdParams(ctypes.Structure):
    """POD mirror of vmasr_sscan_bwd_params."""
    _fields_ = (
        [("f", SScanParams)]
        + [(n, c_i64) for n in ("dout_batch_stride", "dout_d_stride", "du_batch_stride", "du_d_stride",
                                "ddelta_batch_stride", "ddelta_d_stride", "dA_d_stride", "dA_dstate_stride")]
        + [(n, c_vp) for n in ("dout_ptr", "du_ptr", "ddelta_ptr", "dA_ptr", "dB_ptr", "dC_ptr", "dD_ptr",
                               "ddelta_bias_ptr", "ws_ptr")]
        + [("ws_bytes", c_sz)]
    )


class SS2DParams(ctypes.Structure):
    """POD mirror of vmasr_ss2d_params."""
    _fields_ = ([(n, c_i32) for n in ("B", "D", "H", "W", "dtype", "flags")]
                + [(n, c_vp) for n in ("x", "xT", "Wx", "Wdt", "dtb", "Alog", "Ds", "state", "out02", "out13", "y",
                                       "dy", "dyT", "adj", "part", "dx", "dWx", "dWdt", "ddtb", "dAlog", "dDs")])


class CgSlot(ctypes.Structure):
    """POD mirror of vmasr_cg_slot (one stacked discriminator of a convolution launch, csrc/convgemm.hip)."""
    _fields_ = ([(n, c_vp) for n in ("ah", "al", "bh", "bl", "c0", "c1", "ch", "cl", "bias")]
                + [("nseq", c_i64), ("H", c_i32), ("reserved", c_i32)])


class CgGeluBwd(ctypes.Structure):
    """POD mirror of vmasr_cg_gelu_bwd (the activation backward fused into an input-gradient launch, csrc/convgemm.hip)."""
    _fields_ = [("pre", c_vp), ("sgn", c_vp), ("valid", c_i64), ("scale", ctypes.c_float), ("reserved", c_i32), ("db", c_vp)]


class SS2DDeepParams(ctypes.Structure):
    """POD mirror of vmasr_ss2d_deep_params."""
    _fields_ = ([(n, c_i32) for n in ("B", "D", "H", "W", "R", "dtype")]
                + [(n, c_vp) for n in ("x", "WxT", "Wdt", "dtb", "Alog", "Ds", "xdbl", "y", "dy", "du", "tp", "tb", "tc", "pg",
                                       "g32", "dx", "gpos")])


# name -> (restype, argtypes); every symbol declared in include/vmasr_hip.h
SYMBOLS = {
    "vmasr_abi_version": (ctypes.c_int, []),
    "vmasr_last_error": (ctypes.c_char_p, []),
    "vmasr_sscan_chunk": (ctypes.c_int, []),
    "vmasr_sscan_fwd": (ctypes.c_int, [ctypes.POINTER(SScanParams), c_vp]),
    "vmasr_sscan_bwd_workspace": (c_sz, [ctypes.POINTER(SScanBwdParams)]),
    "vmasr_sscan_bwd": (ctypes.c_int, [ctypes.POINTER(SScanBwdParams), c_vp]),
    "vmasr_sscan_tune": (None, [ctypes.c_int, ctypes.c_int]),
    "vmasr_layer_norm_fwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, ctypes.c_float, c_i32, c_i32, c_vp]),
    "vmasr_layer_norm_bwd_workspace": (c_sz, [c_i32, c_i32]),
    "vmasr_layer_norm_bwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "vmasr_layer_norm_bwd_res": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "vmasr_layer_norm_bwd_blocks": (c_i32, [c_i32, c_i32]),
    "vmasr_layer_norm_bwd_reduce_multi": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "vmasr_mlp_supported": (ctypes.c_int, [c_i32, c_i32]),
    "vmasr_mlp_fwd": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_float, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i64, c_i32, c_i32, c_vp]),
    "vmasr_mlp_bwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, ctypes.c_float, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32,
                                     c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp]),
    "vmasr_stft_loss_blocks": (c_i32, []),
    "vmasr_stft_loss_fwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "vmasr_stft_loss_bwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "vmasr_outproj_supported": (ctypes.c_int, [c_i32, c_i32]),
    "vmasr_outproj_fwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i64, c_i32, c_i32, c_vp]),
    "vmasr_outproj_bwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp]),
    "vmasr_inproj_supported": (ctypes.c_int, [c_i32, c_i32, c_i64]),
    "vmasr_inproj_fwd": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_float, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp]),
    "vmasr_inproj_bwd": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_float, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                        c_i64, c_i32, c_i32, c_i32, c_vp]),
    "vmasr_small_linear_supported": (ctypes.c_int, [c_i32, c_i32]),
    "vmasr_small_linear_fwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "vmasr_small_linear_bwd_workspace": (c_sz, [c_i64, c_i32, c_i32]),
    "vmasr_small_linear_bwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "vmasr_cross_scan_cvt": (ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "vmasr_cross_merge_cvt": (ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "vmasr_xproj_supported": (ctypes.c_int, [c_i32, c_i32, c_i32]),
    "vmasr_xproj_fwd": (ctypes.c_int, [c_vp] * 7 + [c_i32] * 7 + [c_vp]),
    "vmasr_xproj_bwd": (ctypes.c_int, [c_vp] * 12 + [c_i32] * 7 + [c_vp]),
    "vmasr_xproj_n_supported": (ctypes.c_int, [c_i32, c_i32, c_i32]),
    "vmasr_xproj_n_ws_floats": (c_sz, [c_i32, c_i32, c_i32, c_i32]),
    "vmasr_xproj_n_fwd": (ctypes.c_int, [c_vp] * 7 + [c_i32] * 7 + [c_vp]),
    "vmasr_xproj_n_bwd": (ctypes.c_int, [c_vp] * 12 + [c_i32] * 7 + [c_vp]),
    "vmasr_spectral_power_iter": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, ctypes.c_float, c_vp]),
    "vmasr_spectral_power_iter_batched": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i64, c_i32, ctypes.c_float, c_vp, c_vp]),
    "vmasr_im2col_kx1": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_i32, c_vp]),
    "vmasr_im2col_kx1_split": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_vp]),
    "vmasr_im2col_kx1_split3_multi": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_vp]),
    "vmasr_col2im_kx1": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "vmasr_im2col_kx1_split_multi": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_vp]),
    "vmasr_col2im_kx1_multi": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_i32, c_vp]),
    "vmasr_col2im_kx1_stacked": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_i64, c_i32, c_vp]),
    "vmasr_stack_rows": (ctypes.c_int, [c_vp, c_vp, c_i32, c_vp, c_i64, c_i64, c_vp]),
    "vmasr_split_bf16": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_vp]),
    "vmasr_bias_gelu_fwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_vp]),
    "vmasr_gelu_bwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_vp]),
    "vmasr_sn_dot_blocks": (c_i32, []),
    "vmasr_sn_stack_fwd": (ctypes.c_int, [c_vp, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "vmasr_sn_stack_bwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "vmasr_conv_post_supported": (ctypes.c_int, [c_i32, c_i32]),
    "vmasr_conv_post_fwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_vp]),
    "vmasr_conv_post_bwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_vp]),
    "vmasr_conv_first_fwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "vmasr_conv_first_bwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "vmasr_adamw_chunk": (c_i32, []),
    "vmasr_adamw_step": (ctypes.c_int, [c_vp, c_vp, c_i32, c_i64, c_vp, c_vp, ctypes.c_float, ctypes.c_float, ctypes.c_float, c_vp]),
    "vmasr_masked_l1_blocks": (c_i32, []),
    "vmasr_masked_l1_fwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i64, c_i32, c_vp]),
    "vmasr_masked_l1_bwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_vp]),
    "vmasr_lsgan_fwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "vmasr_lsgan_bwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "vmasr_masked_l1_bwd_add": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_vp]),
    "vmasr_sum_parts": (ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i64, c_vp]),
    "vmasr_weight_prep_split": (ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "vmasr_gelu_bwd_split": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_vp]),
    "vmasr_conv_mfma_supported": (ctypes.c_int, [c_i32, c_i32, c_i32, c_i32]),
    "vmasr_conv_mfma_supported_launch": (ctypes.c_int, [c_i32, c_i32, c_i32, c_i32, c_i32, ctypes.c_int64]),
    "vmasr_mark_time": (ctypes.c_int, [c_vp, c_vp]),
    "vmasr_im2col2d_rows": (ctypes.c_int, [c_vp, c_vp] + [c_i32] * 10 + [c_vp, c_i32, c_i32, c_vp]),
    "vmasr_col2im2d_rows": (ctypes.c_int, [c_vp, c_vp] + [c_i32] * 10 + [c_vp, c_i32, c_i32, c_vp]),
    "vmasr_skinny_linear_supported": (ctypes.c_int, [c_i64, c_i32, c_i32]),
    "vmasr_skinny_linear": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i64, c_i64, c_i32, c_i32, c_vp]),
    "vmasr_conv_set_cu_limit": (None, [c_i32]),
    "vmasr_conv_get_cu_limit": (c_i32, []),
    "vmasr_conv_mfma_fwd": (ctypes.c_int, [ctypes.POINTER(CgSlot), c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_i32, c_vp]),
    "vmasr_conv_mfma_dgrad": (ctypes.c_int, [ctypes.POINTER(CgSlot), c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_vp]),
    "vmasr_conv_f32_fwd": (ctypes.c_int, [ctypes.POINTER(CgSlot), c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_i32, c_vp]),
    "vmasr_conv_f32_dgrad": (ctypes.c_int, [ctypes.POINTER(CgSlot), c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_vp]),
    "vmasr_conv_mfma_wgrad": (ctypes.c_int, [ctypes.POINTER(CgSlot), c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "vmasr_conv_mfma_dgrad_gelu": (ctypes.c_int, [ctypes.POINTER(CgSlot), ctypes.POINTER(CgGeluBwd), c_vp, c_i32, c_i32, c_i32, c_i32, c_i32,
                                                  c_i32, c_i64, c_vp]),
    "vmasr_wgrad_finish_multi": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "vmasr_linear_f64acc": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp]),
    "vmasr_ss2d_glue_supported": (ctypes.c_int, [c_i32, c_i32, c_i32]),
    "vmasr_ss2d_pre_fwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "vmasr_ss2d_pre_bwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "vmasr_ln_gate_fwd": (ctypes.c_int, [c_vp] * 7 + [c_i32, c_i32, c_i32, ctypes.c_float, c_i32, c_vp]),
    "vmasr_ln_gate_bwd": (ctypes.c_int, [c_vp] * 11 + [c_i32, c_i32, c_i32, c_i32, c_vp]),
    "vmasr_ln_gate_bwd_workspace": (c_i64, [c_i32, c_i32, c_i32, c_i32]),
    "vmasr_ln_gate_bwd_ws": (ctypes.c_int, [c_vp] * 12 + [c_i32, c_i32, c_i32, c_i32, c_vp]),
    "vmasr_ln_gate_pair_supported": (ctypes.c_int, [c_i32, c_i32, c_i32]),
    "vmasr_ln_gate_pair_fwd": (ctypes.c_int, [c_vp] * 8 + [c_i32, c_i32, c_i32, c_i32, ctypes.c_float, c_i32, c_vp]),
    "vmasr_ln_gate_pair_bwd": (ctypes.c_int, [c_vp] * 13 + [c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "vmasr_ss2d_supported": (ctypes.c_int, [c_i32] * 5),
    "vmasr_ss2d_part_floats": (c_sz, [c_i32] * 4),
    "vmasr_ss2d_fwd": (ctypes.c_int, [ctypes.POINTER(SS2DParams), c_vp]),
    "vmasr_ss2d_bwd": (ctypes.c_int, [ctypes.POINTER(SS2DParams), c_vp]),
    "vmasr_ss2d_deep_supported": (ctypes.c_int, [c_i32] * 5),
    "vmasr_ss2d_deep_waves_per_row": (c_i32, [c_i32, c_i32]),
    "vmasr_ss2d_deep_fwd": (ctypes.c_int, [ctypes.POINTER(SS2DDeepParams), c_vp]),
    "vmasr_ss2d_deep_bwd": (ctypes.c_int, [ctypes.POINTER(SS2DDeepParams), c_vp]),
    "vmasr_set_deterministic": (None, [ctypes.c_int]),
    "vmasr_get_deterministic": (ctypes.c_int, []),
    "vmasr_det_timeouts": (ctypes.c_int64, []),
    "vmasr_prof_enable": (None, [ctypes.c_int]),
    "vmasr_prof_reset": (None, []),
    "vmasr_prof_name": (ctypes.c_char_p, [ctypes.c_int]),
    "vmasr_prof_collect": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_int64),
                                          ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "vmasr_prof_collect_shapes": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64),
                                                 ctypes.POINTER(ctypes.c_double)]),
    "vmasr_cross_scan": (ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "vmasr_cross_merge": (ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "vmasr_dwconv_silu_fwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "vmasr_dwconv_silu_bwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                             c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "vmasr_stft": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "vmasr_stft_bwd_workspace": (c_sz, [c_i32, c_i32, c_i32, c_i32]),
    "vmasr_stft_bwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_sz, c_vp]),
    "vmasr_istft_workspace": (c_sz, [c_i32, c_i32, c_i32, c_i32]),
    "vmasr_istft": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_sz, c_vp]),
    "vmasr_istft_bwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
}

_lib = None


def lib():
    """Load libvmasr_hip.so (once) and type its entry points."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `make -C vm_asr_amd/csrc` "
                "(or __graft_entry__.build()). vm_asr_amd has no CPU fallback.")
        # torch bundles its own libamdhip64 (soname libamdhip64.so.7); importing it first makes the
        # loader bind this library to that same runtime instead of a second copy from /opt/rocm
        import torch  # noqa: F401
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)  # AttributeError if the .so is stale
            fn.restype, fn.argtypes = res, args
        if l.vmasr_abi_version() != 1:
            raise RuntimeError("libvmasr_hip.so ABI version mismatch")
        if os.environ.get("VMASR_DETERMINISTIC", "0") == "1":
            l.vmasr_set_deterministic(1)
        _lib = l
    return _lib


def check(code, what):
    if code != 0:
        msg = lib().vmasr_last_error().decode() or "unknown error"
        raise RuntimeError(f"{what} failed ({code}): {msg}")


def torch_dtype_code(dt):
    import torch
    try:
        return {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}[dt]
    except KeyError:
        raise RuntimeError(f"unsupported dtype {dt}: expected float32 / float16 / bfloat16")


def det_mode():
    """Deterministic-reduction mode is in force: VMASR_DETERMINISTIC=1 in the environment OR switched on through the library
    (vmasr_set_deterministic: what the tests and embedding programs use).  Every stream-layout decision (trainer._two_streams,
    model._lanes, enable_graphs' variants, bench.py's labels) asks HERE — the ordered-accumulation tickets are per kernel id, not
    per stream (csrc/common.h), so the mode keeps the one-stream layout however it was switched on."""
    if os.environ.get("VMASR_DETERMINISTIC", "0") == "1":
        return True
    try:
        return bool(lib().vmasr_get_deterministic())
    except (OSError, RuntimeError, AttributeError):
        return False


def current_stream(device):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


K_COUNT = 60


def zeros_f32(device, *shapes):
    """Zero-initialised fp32 tensors of the given shapes (None passes through) carved from ONE
    allocation / one fill kernel, each 16-byte aligned (the kernels' accumulators: dA, dB, dC, ...)."""
    import torch
    sizes = [0 if sh is None else -(-int(torch.Size(sh).numel()) // 4) * 4 for sh in shapes]
    flat = torch.zeros(max(1, sum(sizes)), dtype=torch.float32, device=device)
    out, off = [], 0
    for sh, n in zip(shapes, sizes):
        out.append(None if sh is None else flat[off:off + torch.Size(sh).numel()].view(sh))
        off += n
    return out


def prof_enable(on=True):
    lib().vmasr_prof_enable(int(bool(on)))


def prof_reset():
    lib().vmasr_prof_reset()


def prof_collect():
    """-> {kernel name: dict(launches, ms, alg_bytes)} for kernels launched since the last reset
    (waits for their events)."""
    l = lib()
    out = {}
    for k in range(K_COUNT):
        n, ms, by = ctypes.c_int64(0), ctypes.c_double(0.0), ctypes.c_double(0.0)
        check(l.vmasr_prof_collect(k, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(by)), "prof_collect")
        if n.value:
            out[l.vmasr_prof_name(k).decode()] = dict(launches=n.value, ms=ms.value, alg_bytes=by.value)
    return out


def prof_collect_shapes(name, max_groups=16):
    """-> [dict(alg_bytes_per_launch, launches, ms)] of kernel `name`, one entry per distinct call shape (= algorithmic byte count)."""
    l = lib()
    for k in range(K_COUNT):
        if l.vmasr_prof_name(k).decode() == name:
            by, n, ms = (ctypes.c_double * max_groups)(), (ctypes.c_int64 * max_groups)(), (ctypes.c_double * max_groups)()
            ng = l.vmasr_prof_collect_shapes(k, max_groups, by, n, ms)
            if ng < 0:
                raise RuntimeError(f"prof_collect_shapes failed ({ng})")
            return [dict(alg_bytes_per_launch=by[i], launches=n[i], ms=ms[i]) for i in range(ng)]
    raise KeyError(name)
