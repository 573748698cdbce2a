"""ctypes binding of libmfpa.so (the C ABI declared in include/mfpa.h).

There is no CPU fallback: if the HIP library is missing or a call fails, this
module raises.  torch is imported first so that the HIP runtime already mapped by
PyTorch-ROCm (SONAME libamdhip64.so.7) is the one libmfpa.so binds to -- kernels
are then enqueued on torch's current stream and operate on torch allocations.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_double, c_float, c_int, c_longlong, c_uint, c_void_p

import torch  # noqa: F401  (must precede the dlopen below)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmfpa.so")    # the environment cannot redirect it; tools/ A/B runs call set_library_path()

EINVAL = -22
EHIP = -1000
F32, F64 = 0, 1
STFT_TABLE_LEN = 1288
ABI_VERSION = 38


class MfpaError(RuntimeError):
    pass


_SIGNATURES = {
    "mfpa_version": ([], c_int),
    "mfpa_stft_tables": ([c_void_p, c_void_p], c_int),
    "mfpa_stft_frames": ([c_int], c_int),
    "mfpa_specgram_frames": ([c_int], c_int),
    "mfpa_stft_mag": ([c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p], c_int),
    "mfpa_specgram_psd": ([c_void_p, c_int, c_int, c_double, c_void_p, c_void_p, c_void_p, c_void_p], c_int),
    "mfpa_div_by_reciprocal": ([c_void_p, c_void_p, c_longlong, c_void_p, c_void_p], c_int),
    "mfpa_normalize_f32": ([c_void_p, c_int, c_longlong, c_void_p, c_void_p, c_void_p], c_int),
    "mfpa_normalize": ([c_void_p, c_int, c_int, c_longlong, c_void_p, c_int, c_void_p], c_int),
    "mfpa_f64_to_f32": ([c_void_p, c_void_p, c_longlong, c_void_p], c_int),
    "mfpa_audfprint_prepare": ([c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_double,
                                c_void_p, c_void_p, c_void_p], c_int),
    "mfpa_audfprint_prune": ([c_void_p, c_int, c_int, c_int, c_void_p, c_double, c_int, c_void_p, c_void_p,
                              c_void_p], c_int),
    "mfpa_audfprint_pick": ([c_void_p, c_void_p, c_int, c_int, c_int, c_double, c_void_p, c_double, c_int, c_void_p, c_void_p,
                             c_void_p, c_void_p], c_int),
    "mfpa_dejavu_prepare": ([c_void_p, c_int, c_int, c_int, c_void_p, c_double, c_int, c_void_p, c_void_p], c_int),
    "mfpa_dejavu_prepare_f32": ([c_void_p, c_int, c_int, c_int, c_int, c_double, c_int, c_void_p, c_void_p], c_int),
    "mfpa_localmax2d": ([c_void_p, c_int, c_int, c_int, c_int, c_double, c_void_p, c_void_p, c_void_p], c_int),
    "mfpa_dejavu_pick_work_doubles": ([c_int, c_int, c_void_p], c_int),
    "mfpa_dejavu_pick": ([c_void_p, c_void_p, c_int, c_int, c_int, c_double, c_int, c_int, c_double, c_void_p, c_void_p, c_void_p,
                          c_void_p], c_int),
    "mfpa_peak_metrics": ([c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p], c_int),
    "mfpa_psnr_stats": ([c_void_p, c_int, c_void_p, c_int, c_longlong, c_void_p, c_void_p], c_int),
    "mfpa_audfprint_landmarks": ([c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p], c_int),
    "mfpa_dejavu_hashes": ([c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                            c_void_p, c_void_p], c_int),
    "mfpa_conv3x3_bn_relu": ([c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int,
                              c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p], c_int),
    "mfpa_conv3x3_c1_bn_relu": ([c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int,
                                 c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p], c_int),
    "mfpa_maxpool2": ([c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p], c_int),
    "mfpa_convT2x2": ([c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p,
                       c_void_p], c_int),
    "mfpa_conv1x1_out": ([c_void_p, c_longlong, c_int, c_void_p, c_float, c_void_p, c_void_p], c_int),
    "mfpa_conv_mfma": ([c_void_p, c_void_p], c_int),
    "mfpa_conv_weight_layout": ([c_int, c_int, c_int, c_int, c_int, c_int], c_int),
    "mfpa_conv_scale_folds": ([c_int, c_int, c_int, c_int], c_int),
    "mfpa_conv_c1_layout": ([c_int, c_int], c_int),
    "mfpa_upconv_fused": ([c_void_p, c_void_p], c_int),
    "mfpa_upconv_pack": ([c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p], c_int),
    "mfpa_upconv_serves": ([c_int, c_int, c_int, c_int, c_int, c_int, c_int], c_int),
    "mfpa_gemm_mfma": ([c_void_p, c_void_p], c_int),
    "mfpa_lowpass_taps": ([c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p], c_int),
    "mfpa_fir": ([c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                  c_void_p, c_void_p], c_int),
    "mfpa_scale_rows": ([c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p], c_int),
    "mfpa_gather_background": ([c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p], c_int),
    "mfpa_mix_background": ([c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p], c_int),
    "mfpa_clip_quantile": ([c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p], c_int),
    "mfpa_clip_quantile_flat": ([c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p], c_int),
    "mfpa_demucs_prep": ([c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p], c_int),
    "mfpa_upsample2": ([c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p], c_int),
    "mfpa_downsample2": ([c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p], c_int),
    "mfpa_conv1d_c1_relu": ([c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p], c_int),
    "mfpa_convT1d_c1": ([c_void_p, c_int, c_int, c_int, c_void_p, c_float, c_void_p, c_void_p], c_int),
    "mfpa_convT1d_c1_dev": ([c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p], c_int),
    "mfpa_loss_blocks": ([], c_int),
    "mfpa_reflect_pad": ([c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p], c_int),
    "mfpa_dft_mag": ([c_void_p, c_longlong, c_int, c_longlong, c_int, c_void_p, c_void_p], c_int),
    "mfpa_stft_loss_sums": ([c_void_p, c_void_p, c_longlong, c_int, c_longlong, c_int, c_void_p, c_void_p, c_void_p], c_int),
    "mfpa_stft_loss_grad": ([c_void_p, c_void_p, c_longlong, c_int, c_longlong, c_int, c_void_p, c_double, c_double, c_void_p], c_int),
    "mfpa_frames_adjoint": ([c_void_p, c_int, c_int, c_longlong, c_int, c_int, c_int, c_int, c_void_p, c_void_p], c_int),
    "mfpa_reflect_pad_adjoint": ([c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p], c_int),
    "mfpa_lstm_step": ([c_void_p, c_longlong, c_void_p, c_void_p, c_longlong, c_void_p, c_int, c_int, c_void_p, c_longlong,
                        c_void_p, c_void_p, c_longlong, c_void_p], c_int),
    "mfpa_lstm_cell": ([c_void_p, c_longlong, c_void_p, c_int, c_int, c_void_p, c_longlong, c_void_p, c_void_p,
                        c_longlong, c_void_p], c_int),
    "mfpa_conv1d_c1": ([c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p], c_int),
    "mfpa_lstm_step_train": ([c_void_p, c_longlong, c_void_p, c_void_p, c_longlong, c_void_p, c_longlong, c_void_p, c_longlong,
                              c_int, c_int, c_void_p, c_longlong, c_void_p, c_void_p, c_longlong, c_void_p, c_longlong,
                              c_void_p], c_int),
    "mfpa_lstm_step_bwd": ([c_void_p, c_longlong, c_void_p, c_void_p, c_longlong, c_void_p, c_longlong, c_void_p, c_longlong,
                            c_void_p, c_longlong, c_void_p, c_int, c_int, c_void_p], c_int),
    "mfpa_lstm_layer": ([c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p],
                        c_int),
    "mfpa_lstm_layer_bwd": ([c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p], c_int),
    "mfpa_lstm_layer_range": ([c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                               c_int, c_void_p], c_int),
    "mfpa_conv1d_c1_glu": ([c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p], c_int),
    "mfpa_glu_convT1d_c1": ([c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p], c_int),
    "mfpa_lstm_layer_seq": ([c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                             c_int, c_int, c_void_p, c_void_p], c_int),
    "mfpa_lstm_seq_workgroups": ([c_int, c_int, c_int, c_void_p], c_int),
    "mfpa_lstm_bwd_seq_workgroups": ([c_int, c_int, c_int, c_void_p], c_int),
    "mfpa_lstm_layer_bwd_seq": ([c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
                                 c_int),
    "mfpa_lstm_bwd_seq_work_bytes": ([c_int, c_int, c_void_p], c_int),
    "mfpa_lstm_seq_work_bytes": ([c_int, c_int, c_void_p], c_int),
    "mfpa_lstm_seq_error_offset": ([], c_int),
    "mfpa_lstm_layer_bwd_range": ([c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
                                  c_int),
    "mfpa_gemm_tn": ([c_void_p, c_void_p], c_int),
    "mfpa_glu_bwd": ([c_void_p, c_longlong, c_int, c_int, c_void_p, c_longlong, c_void_p], c_int),
    "mfpa_colsum_any": ([c_void_p, c_longlong, c_int, c_longlong, c_void_p, c_void_p], c_int),
    "mfpa_c1_wgrad": ([c_void_p, c_longlong, c_void_p, c_longlong, c_longlong, c_int, c_int, c_int, c_void_p, c_void_p], c_int),
    "mfpa_downsample2_adjoint": ([c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p], c_int),
    "mfpa_red_blocks": ([], c_int),
    "mfpa_bn_stats": ([c_void_p, c_longlong, c_int, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p, c_void_p,
                       c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p], c_int),
    "mfpa_bn_relu_bwd": ([c_void_p, c_void_p, c_longlong, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                          c_void_p, c_void_p, c_void_p, c_void_p, c_uint, c_uint, c_float, c_void_p, c_int, c_int, c_void_p], c_int),
    "mfpa_conv_stats_rows": ([c_int, c_int, c_int, c_int, c_int], c_int),
    "mfpa_conv_stats_reduce": ([c_void_p, c_longlong, c_int, c_void_p, c_void_p, c_void_p], c_int),
    "mfpa_bn_stats_sums": ([c_void_p, c_longlong, c_int, c_void_p, c_void_p, c_int, c_void_p], c_int),
    "mfpa_bn_stats_finish": ([c_void_p, c_double, c_int, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_void_p, c_void_p, c_void_p], c_int),
    "mfpa_bn_relu_bwd_sums": ([c_void_p, c_void_p, c_longlong, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_uint, c_uint, c_float, c_int, c_void_p], c_int),
    "mfpa_bn_relu_bwd_finish": ([c_void_p, c_void_p, c_longlong, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_double, c_void_p, c_void_p, c_void_p, c_uint, c_uint, c_float,
                                 c_void_p, c_int, c_int, c_int, c_void_p], c_int),
    "mfpa_colsum": ([c_void_p, c_longlong, c_int, c_void_p, c_void_p, c_void_p], c_int),
    "mfpa_conv_stats_bn_finish": ([c_void_p, c_longlong, c_int, c_double, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p], c_int),
    "mfpa_bn_relu_bwd_from_part": ([c_void_p, c_void_p, c_longlong, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_longlong,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_uint, c_uint, c_float, c_void_p, c_int, c_int, c_int, c_void_p], c_int),
    "mfpa_maxpool2_bwd_bn_relu_bwd": ([c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                       c_uint, c_uint, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p], c_int),
    "mfpa_bn_relu_pool": ([c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_uint, c_uint, c_float,
                           c_int, c_int, c_void_p], c_int),
    "mfpa_maxpool2_bwd_add": ([c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_uint, c_uint, c_float, c_int, c_void_p], c_int),
    "mfpa_maxpool2_bwd_add_sums": ([c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_uint, c_uint, c_float, c_void_p, c_int, c_void_p], c_int),
    "mfpa_wgrad_mfma": ([c_void_p, c_void_p], c_int),
    "mfpa_wgrad_c1": ([c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p], c_int),
    "mfpa_outconv_fwd": ([c_void_p, c_longlong, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
                         c_int),
    "mfpa_outconv_bwd": ([c_void_p, c_void_p, c_longlong, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                          c_void_p, c_int, c_void_p], c_int),
    "mfpa_outconv_bwd_rows": ([c_longlong, c_int, c_void_p], c_int),
    "mfpa_outconv_bwd_sums": ([c_void_p, c_void_p, c_longlong, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_int, c_void_p], c_int),
    "mfpa_bn_relu_bwd_finish_rank1": ([c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p], c_int),
    "mfpa_l1_loss": ([c_void_p, c_void_p, c_longlong, c_void_p, c_void_p, c_void_p, c_void_p], c_int),
    "mfpa_act_to_bf16": ([c_void_p, c_longlong, c_int, c_void_p, c_void_p, c_uint, c_uint, c_float, c_void_p, c_void_p], c_int),
    "mfpa_pack_conv_weights": ([c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p], c_int),
    "mfpa_pack_conv_weights_batch": ([c_void_p, c_int, c_longlong, c_void_p], c_int),
    "mfpa_adam_step": ([c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_float, c_float, c_float, c_float, c_int,
                        c_float, c_void_p], c_int),
}


class PackJob(ctypes.Structure):
    """mfpa_pack_job (include/mfpa.h)."""
    _fields_ = [("w", c_void_p), ("out", c_void_p), ("taps", c_int), ("Co", c_int), ("Ci", c_int), ("flip_transpose", c_int), ("row0", c_int),
                ("nrows", c_int), ("precision", c_int), ("pad_", c_int), ("tile0", c_longlong)]


class UpconvDesc(ctypes.Structure):
    """mfpa_upconv_desc of include/mfpa.h."""
    _fields_ = [("skip", c_void_p), ("low", c_void_p), ("w_skip", c_void_p), ("w_up", c_void_p), ("shift", c_void_p), ("bias_tab", c_void_p),
                ("y", c_void_p), ("B", c_int), ("H", c_int), ("W", c_int), ("Cs", c_int), ("Hl", c_int), ("Wl", c_int), ("Cl", c_int),
                ("Cout", c_int), ("relu", c_int), ("precision", c_int)]


class ConvDesc(ctypes.Structure):
    """mfpa_conv_desc of include/mfpa.h."""
    _fields_ = [("x0", c_void_p), ("in_scale0", c_void_p), ("in_shift0", c_void_p), ("x1", c_void_p),
                ("w", c_void_p), ("out_scale", c_void_p), ("out_shift", c_void_p), ("y", c_void_p),
                ("C0", c_int), ("C1", c_int), ("H1", c_int), ("W1", c_int),
                ("B", c_int), ("H", c_int), ("W", c_int), ("Cout", c_int), ("relu", c_int),
                ("yH", c_int), ("yW", c_int), ("mode", c_int),
                ("drop_seed", c_uint), ("drop_thresh", c_uint), ("drop_scale", c_float), ("precision", c_int),
                ("y_pool", c_void_p), ("w1x1", c_void_p), ("b1x1", c_float), ("y1x1", c_void_p),
                ("c1_x32", c_void_p), ("c1_spec64", c_void_p), ("c1_denom", c_void_p),
                ("c1_w", c_void_p), ("c1_scale", c_void_p), ("c1_shift", c_void_p), ("w_layout", c_int),
                ("x0_bf16", c_void_p), ("x1_bf16", c_void_p), ("y_bf16", c_void_p), ("stats_part", c_void_p),
                ("bwd_z", c_void_p), ("bwd_scale", c_void_p), ("bwd_shift", c_void_p), ("bwd_mean", c_void_p), ("bwd_invstd", c_void_p),
                ("x0_is_bf16", c_int), ("x0_split", c_int), ("x1_split", c_int), ("y_split", c_int), ("y_pool_split", c_int), ("bwd_z_is_bf16", c_int)]


class GemmDesc(ctypes.Structure):
    """mfpa_gemm_desc of include/mfpa.h."""
    _fields_ = [("A", c_void_p), ("lda", c_longlong), ("strideA", c_longlong), ("W", c_void_p), ("bias", c_void_p),
                ("addend", c_void_p), ("ldadd", c_longlong), ("strideAdd", c_longlong),
                ("C", c_void_p), ("ldc", c_longlong), ("strideC", c_longlong),
                ("batch", c_int), ("M", c_int), ("N", c_int), ("K", c_int), ("npad", c_int), ("mode", c_int),
                ("relu", c_int), ("precision", c_int),
                ("c1_x", c_void_p), ("c1_lin", c_longlong), ("c1_w", c_void_p), ("c1_b", c_void_p),
                ("C2", c_void_p), ("ldc2", c_longlong), ("strideC2", c_longlong)]


class GemmTnDesc(ctypes.Structure):
    """mfpa_gemm_tn_desc of include/mfpa.h."""
    _fields_ = [("A", c_void_p), ("lda", c_longlong), ("strideA", c_longlong),
                ("Bm", c_void_p), ("ldb", c_longlong), ("strideB", c_longlong),
                ("C", c_void_p), ("ldc", c_longlong),
                ("batch", c_int), ("R", c_int), ("M", c_int), ("N", c_int), ("colsum", c_void_p), ("precision", c_int)]


class WgradDesc(ctypes.Structure):
    """mfpa_wgrad_desc of include/mfpa.h."""
    _fields_ = [("dz", c_void_p), ("x0", c_void_p), ("in_scale0", c_void_p), ("in_shift0", c_void_p),
                ("x1", c_void_p), ("dw", c_void_p),
                ("C0", c_int), ("C1", c_int), ("H1", c_int), ("W1", c_int),
                ("B", c_int), ("H", c_int), ("W", c_int), ("Cout", c_int), ("mode", c_int),
                ("drop_seed", c_uint), ("drop_thresh", c_uint), ("drop_scale", c_float), ("precision", c_int)]

_lib = None


def exported_symbols():
    """Names include/mfpa.h declares (used by the CPU-side ABI test)."""
    return sorted(_SIGNATURES)


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MfpaError(
                f"{LIB_PATH} is missing: build it with `python -m musicfpaugment_amd.csrc.build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (argtypes, restype) in _SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise MfpaError(f"libmfpa.so does not export {name}; rebuild it") from e
            fn.argtypes = argtypes
            fn.restype = restype
        if handle.mfpa_version() != ABI_VERSION:
            raise MfpaError(f"libmfpa.so ABI {handle.mfpa_version()} != expected {ABI_VERSION}; rebuild it")
        _lib = handle
    return _lib


def set_library_path(path: str) -> None:
    """tools/ and tests only: bind another build of the library (e.g. libmfpa_exp.so, the -DMFPA_EXPERIMENTS build) before the
    first call.  The product path never calls this, and no environment variable redirects the library."""
    global LIB_PATH, _lib
    if _lib is not None:
        raise MfpaError("set_library_path() must be called before the first libmfpa call")
    LIB_PATH = os.path.abspath(path)


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    if rc == EINVAL:
        raise ValueError(f"{what}: invalid argument (MFPA_EINVAL)")
    if rc <= EHIP:
        raise MfpaError(f"{what}: HIP error {EHIP - rc}")
    raise MfpaError(f"{what}: error {rc}")


def ptr(t) -> int:
    """Device pointer of a contiguous CUDA(HIP) tensor, or 0 for None."""
    if t is None:
        return 0
    if not t.is_cuda:
        raise MfpaError("libmfpa operates on GPU tensors only (no CPU fallback)")
    if not t.is_contiguous():
        raise MfpaError("libmfpa needs contiguous tensors")
    if t.device.index != torch.cuda.current_device():
        # kernels are enqueued on the CURRENT device's stream: a tensor of another GPU would fault or go through peer access
        raise MfpaError(f"tensor on {t.device} but the current device is cuda:{torch.cuda.current_device()}: "
                        "call torch.cuda.set_device(...) (one process per GPU) or wrap the call in torch.cuda.device(t.device)")
    return t.data_ptr()


def stream() -> int:
    """HIP stream handle of torch's current stream on the current device (ptr() checks that every operand lives there)."""
    return torch.cuda.current_stream().cuda_stream


def require_gpu(t, name="input"):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise MfpaError(f"{name} must be a tensor on the MI355X (got {type(t).__name__}"
                        f"{'' if not isinstance(t, torch.Tensor) else ' on ' + str(t.device)}); no CPU fallback")
