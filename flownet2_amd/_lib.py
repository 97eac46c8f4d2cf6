"""ctypes binding of libflownet2_hip.so (the C ABI in include/flownet2_hip.h).

There is NO fallback: if the shared library is missing or does not load, every operator raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libflownet2_hip.so")

# every symbol include/flownet2_hip.h declares
EXPORTS = [
    "fn2_version", "fn2_last_error_string",
    "fn2_correlation_out_shape", "fn2_correlation_workspace_bytes", "fn2_correlation_forward", "fn2_correlation_forward_fused", "fn2_correlation_backward",
    "fn2_correlation1d_out_shape", "fn2_correlation1d_forward", "fn2_correlation1d_backward",
    "fn2_flow_warp_forward", "fn2_flow_warp_forward_slices", "fn2_flow_warp_backward_workspace_bytes", "fn2_flow_warp_backward",
    "fn2_resample_forward", "fn2_resample_forward_slices",
    "fn2_l1loss_workspace_bytes", "fn2_l1loss_forward", "fn2_l1loss_backward",
    "fn2_l1loss_multi_workspace_bytes", "fn2_l1loss_multi_sync_bytes", "fn2_l1loss_forward_multi", "fn2_l1loss_backward_multi",
    "fn2_channel_norm_forward", "fn2_channel_norm_forward_slices", "fn2_channel_norm_backward",
    "fn2_downsample_forward",
    "fn2_downsample_forward_multi",
    "fn2_predict_flow_conv_workspace_bytes", "fn2_predict_flow_conv_forward", "fn2_upsample_flow_deconv_forward", "fn2_upsample_flow_deconv_forward_into",
    "fn2_predict_flow_conv_backward_supported", "fn2_predict_flow_conv_backward_workspace_bytes", "fn2_predict_flow_conv_backward", "fn2_upsample_flow_deconv_backward_workspace_bytes", "fn2_upsample_flow_deconv_backward",
    "fn2_bias_leaky_relu_forward", "fn2_scale_shift_forward", "fn2_bias_leaky_relu_backward_workspace_bytes", "fn2_bias_leaky_relu_backward", "fn2_bias_leaky_relu_backward_slices", "fn2_bias_leaky_relu_backward_slices2",
    "fn2_conv_k7s2_relu_supported", "fn2_conv_k7s2_relu_forward",
    "fn2_conv_k7s2_wgrad_supported", "fn2_conv_k7s2_wgrad_ksplit", "fn2_conv_k7s2_wgrad_workspace_bytes", "fn2_conv_k7s2_wgrad",
    "fn2_conv_mfma_supported", "fn2_conv_mfma_packed_floats", "fn2_conv_mfma_pack_weights", "fn2_conv_mfma_pack_weights_view", "fn2_conv_mfma_forward",
    "fn2_conv_mfma_num_variants", "fn2_debug_set_conv_variant",
    "fn2_caffemodel_index", "fn2_caffemodel_read_blob", "fn2_hdf5_index", "fn2_hdf5_read_float",
    "fn2_conv_wino_supported", "fn2_conv_wino_packed_floats", "fn2_conv_wino_pack_weights", "fn2_conv_wino_forward",
    "fn2_conv_wino_num_variants", "fn2_debug_set_wino_variant",
    "fn2_conv_plane_supported", "fn2_conv_plane_ksplit", "fn2_conv_plane_workspace_bytes", "fn2_conv_plane_forward",
    "fn2_conv_plane_k_supported", "fn2_conv_plane_k_ksplit", "fn2_conv_plane_k_workspace_bytes", "fn2_conv_plane_k_forward",
    "fn2_conv_plane_num_variants", "fn2_debug_set_plane_variant", "fn2_debug_set_plane_ksplit", "fn2_set_batch_invariant", "fn2_get_batch_invariant",
    "fn2_deconv_plane_supported", "fn2_deconv_plane_ksplit", "fn2_deconv_plane_workspace_bytes", "fn2_deconv_plane_packed_floats",
    "fn2_deconv_plane_pack_weights", "fn2_deconv_plane_pack_weights_k", "fn2_deconv_plane_forward",
    "fn2_tconv_supported", "fn2_tconv_forward", "fn2_tconv_num_variants", "fn2_debug_set_tconv_variant",
    "fn2_debug_set_wgrad_buffers", "fn2_debug_set_wgrad_chunk", "fn2_conv_wgrad_supported", "fn2_conv_wgrad_ksplit", "fn2_conv_wgrad_workspace_bytes", "fn2_conv_wgrad",
    "fn2_conv_route", "fn2_conv_packed_weight_floats", "fn2_conv_pack_weights", "fn2_conv_workspace_bytes", "fn2_conv_forward",
    "fn2_deconv_route", "fn2_deconv_packed_weight_floats", "fn2_deconv_pack_weights", "fn2_deconv_workspace_bytes", "fn2_deconv_forward",
    "fn2_conv_backward_data_route", "fn2_conv_backward_data_packed_weight_floats", "fn2_conv_backward_data_pack_workspace_bytes",
    "fn2_conv_backward_data_pack_weights", "fn2_conv_backward_data_workspace_bytes", "fn2_conv_backward_data_workspace_bytes_with_room", "fn2_conv_backward_data_computed_channels", "fn2_conv_backward_data",
    "fn2_conv_backward_data_masked_supported", "fn2_conv_backward_data_masked", "fn2_conv_backward_weights_bias_fused", "fn2_conv_backward_weights_bias",
    "fn2_conv_backward_weights_supported", "fn2_conv_backward_weights_workspace_bytes", "fn2_conv_backward_weights", "fn2_conv_backward_bias",
    "fn2_im2col_forward", "fn2_col2im_bias_relu_forward", "fn2_col2im_bias_relu_forward_into",
    "fn2_datum_parse", "fn2_datum_float_data", "fn2_datum_serialize",
    "fn2_custom_data_sample_bytes", "fn2_custom_data_encode_sample", "fn2_custom_data_stage_records", "fn2_custom_data_decode_forward",
    "fn2_augmentation_matrix", "fn2_flow_augmentation_forward",
    "fn2_data_augmentation_workspace_bytes", "fn2_data_augmentation_forward",
]


class Fn2Error(RuntimeError):
    """Non-zero status from the C ABI (the reference would have hit CHECK / LOG(FATAL))."""

    def __init__(self, status: int, message: str):
        super().__init__(f"[fn2 status {status}] {message}")
        self.status = status


class CaffemodelEntry(C.Structure):
    """fn2_caffemodel_entry (include/flownet2_hip.h)."""
    _fields_ = [("name_off", C.c_size_t), ("name_len", C.c_size_t), ("type_off", C.c_size_t), ("type_len", C.c_size_t),
                ("v1_type", C.c_longlong), ("v1", C.c_int), ("blob_index", C.c_int), ("num_axes", C.c_int), ("dim", C.c_longlong * 8),
                ("count", C.c_size_t), ("is_double", C.c_int), ("blob_off", C.c_size_t), ("blob_len", C.c_size_t)]


class Hdf5Entry(C.Structure):
    """fn2_hdf5_entry (include/flownet2_hip.h)."""
    _fields_ = [("path", C.c_char * 256), ("num_axes", C.c_int), ("dim", C.c_longlong * 8), ("count", C.c_size_t), ("type_class", C.c_int),
                ("type_size", C.c_int), ("type_signed", C.c_int), ("big_endian", C.c_int), ("layout", C.c_int), ("num_filters", C.c_int),
                ("header_off", C.c_size_t)]


class ConvDesc(C.Structure):
    """fn2_conv_desc (include/flownet2_hip.h): the geometry the library routes a Convolution / Deconvolution by."""
    _fields_ = [("N", C.c_int), ("Cin", C.c_int), ("Hin", C.c_int), ("Win", C.c_int), ("Cout", C.c_int), ("kernel", C.c_int), ("stride", C.c_int), ("pad", C.c_int)]


class CorrParams(C.Structure):
    _fields_ = [("pad", C.c_int), ("kernel_size", C.c_int), ("max_displacement", C.c_int),
                ("stride1", C.c_int), ("stride2", C.c_int), ("corr_type", C.c_int), ("do_abs", C.c_int),
                ("single_direction", C.c_int)]


class DatumView(C.Structure):
    _fields_ = [("channels", C.c_int), ("height", C.c_int), ("width", C.c_int), ("label", C.c_int), ("encoded", C.c_int),
                ("data", C.c_void_p), ("data_bytes", C.c_size_t), ("float_data_count", C.c_size_t)]


class DataAugParams(C.Structure):
    _fields_ = [("crop_width", C.c_int), ("crop_height", C.c_int), ("max_multiplier", C.c_float), ("has_chromatic_eigvec", C.c_int),
                ("chromatic_eigvec", C.c_float * 9), ("mean_mode", C.c_int), ("noise_seed", C.c_ulonglong), ("noise_stream", C.c_ulonglong)]


class L1LossParams(C.Structure):
    _fields_ = [("l2_per_location", C.c_int), ("l2_prescale_by_channels", C.c_int),
                ("normalize_by_num_entries", C.c_int), ("epsilon", C.c_float), ("plateau", C.c_float)]


class L1LossScale(C.Structure):
    _fields_ = [("bottom0", C.c_void_p), ("bottom1", C.c_void_p), ("bottom0_diff", C.c_void_p), ("bottom1_diff", C.c_void_p),
                ("N", C.c_int), ("C", C.c_int), ("H", C.c_int), ("W", C.c_int), ("loss_weight", C.c_float)]


_lib = None


def lib():
    """Load (once) and return the CDLL.  Raises if the native library is unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(
            f"{SO_PATH} not found: the HIP extension is not built. Run `python -m flownet2_amd.build` "
            "(or __graft_entry__.build()). flownet2_amd has no CPU / PyTorch fallback.")
    L = C.CDLL(SO_PATH)
    vp, fp, i, sz = C.c_void_p, C.c_void_p, C.c_int, C.c_size_t
    L.fn2_version.restype = C.c_char_p
    L.fn2_last_error_string.restype = C.c_char_p
    L.fn2_correlation_out_shape.argtypes = [C.POINTER(CorrParams), i, i, i, C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    L.fn2_correlation_workspace_bytes.argtypes = [C.POINTER(CorrParams), i, i, i, i]
    L.fn2_correlation_workspace_bytes.restype = sz
    L.fn2_correlation_forward.argtypes = [C.POINTER(CorrParams), fp, fp, fp, i, i, i, i, vp, sz, vp]
    L.fn2_correlation_forward_fused.argtypes = [C.POINTER(CorrParams), fp, fp, fp, i, i, i, i, i, i, i, C.c_float, vp, sz, vp]
    L.fn2_correlation_backward.argtypes = [C.POINTER(CorrParams), fp, fp, fp, fp, fp, i, i, i, i, vp, sz, vp]
    L.fn2_correlation1d_out_shape.argtypes = [C.POINTER(CorrParams), i, i, i, C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    L.fn2_correlation1d_forward.argtypes = [C.POINTER(CorrParams), fp, fp, fp, i, i, i, i, vp]
    L.fn2_correlation1d_backward.argtypes = [C.POINTER(CorrParams), fp, fp, fp, fp, fp, i, i, i, i, vp]
    L.fn2_flow_warp_forward.argtypes = [fp, fp, fp, i, i, i, i, i, vp]
    L.fn2_flow_warp_forward_slices.argtypes = [fp, i, i, fp, i, i, fp, i, i, i, i, i, i, i, vp]
    L.fn2_flow_warp_backward_workspace_bytes.argtypes = [i, i, i, i]
    L.fn2_flow_warp_backward_workspace_bytes.restype = sz
    L.fn2_flow_warp_backward.argtypes = [fp, fp, fp, fp, fp, i, i, i, i, i, i, vp, sz, vp]
    L.fn2_resample_forward.argtypes = [fp, fp, i, i, i, i, i, i, i, i, vp]
    L.fn2_resample_forward_slices.argtypes = [fp, C.c_float, fp, i, i, fp, i, i, C.c_float, i, i, i, i, i, i, i, i, vp]
    L.fn2_l1loss_workspace_bytes.argtypes = [i, i, i, i]
    L.fn2_l1loss_workspace_bytes.restype = sz
    L.fn2_l1loss_forward.argtypes = [C.POINTER(L1LossParams), fp, fp, fp, i, i, i, i, vp, sz, vp]
    L.fn2_l1loss_backward.argtypes = [C.POINTER(L1LossParams), fp, fp, C.c_float, fp, fp, i, i, i, i, vp, sz, vp]
    L.fn2_l1loss_multi_workspace_bytes.argtypes = [i]
    L.fn2_l1loss_multi_workspace_bytes.restype = sz
    L.fn2_l1loss_multi_sync_bytes.argtypes = []
    L.fn2_l1loss_multi_sync_bytes.restype = sz
    L.fn2_l1loss_forward_multi.argtypes = [C.POINTER(L1LossParams), i, C.POINTER(L1LossScale), fp, fp, vp, sz, vp, vp]
    L.fn2_l1loss_backward_multi.argtypes = [C.POINTER(L1LossParams), i, C.POINTER(L1LossScale), fp, vp, sz, vp]
    L.fn2_channel_norm_forward.argtypes = [fp, fp, i, i, i, i, vp]
    L.fn2_channel_norm_forward_slices.argtypes = [fp, i, i, fp, i, i, fp, i, i, i, i, i, i, vp]
    L.fn2_channel_norm_backward.argtypes = [fp, fp, fp, fp, i, i, i, i, vp]
    L.fn2_downsample_forward.argtypes = [fp, fp, i, i, i, i, i, i, vp]
    L.fn2_downsample_forward_multi.argtypes = [fp, C.POINTER(fp), C.POINTER(i), C.POINTER(i), i, i, i, i, i, vp]
    L.fn2_predict_flow_conv_workspace_bytes.argtypes = [i, i, i, i]
    L.fn2_predict_flow_conv_workspace_bytes.restype = sz
    L.fn2_predict_flow_conv_forward.argtypes = [fp, fp, fp, fp, i, i, i, i, vp, sz, vp]
    L.fn2_predict_flow_conv_backward_supported.argtypes = [i, i, i, i]
    L.fn2_predict_flow_conv_backward_workspace_bytes.argtypes = [i, i, i, i]
    L.fn2_predict_flow_conv_backward_workspace_bytes.restype = sz
    L.fn2_predict_flow_conv_backward.argtypes = [fp, i, i, fp, fp, fp, fp, fp, i, i, i, i, i, vp, sz, vp]
    L.fn2_upsample_flow_deconv_backward_workspace_bytes.argtypes = [i, i, i]
    L.fn2_upsample_flow_deconv_backward_workspace_bytes.restype = sz
    L.fn2_upsample_flow_deconv_backward.argtypes = [fp, fp, fp, fp, fp, fp, i, i, i, i, vp, sz, vp]
    L.fn2_upsample_flow_deconv_forward.argtypes = [fp, fp, fp, fp, i, i, i, vp]
    L.fn2_upsample_flow_deconv_forward_into.argtypes = [fp, fp, fp, fp, i, i, i, i, i, vp]
    L.fn2_bias_leaky_relu_forward.argtypes = [fp, fp, i, i, i, i, C.c_float, vp]
    L.fn2_scale_shift_forward.argtypes = [fp, fp, fp, i, i, i, i, i, i, C.c_float, vp]
    L.fn2_bias_leaky_relu_backward_workspace_bytes.argtypes = [i, i, i, i]
    L.fn2_bias_leaky_relu_backward_workspace_bytes.restype = sz
    L.fn2_bias_leaky_relu_backward.argtypes = [fp, fp, fp, fp, i, i, i, i, C.c_float, vp, sz, vp]
    L.fn2_bias_leaky_relu_backward_slices.argtypes = [fp, fp, i, i, fp, fp, i, i, i, i, C.c_float, vp, sz, vp]
    L.fn2_bias_leaky_relu_backward_slices2.argtypes = [fp, i, i, fp, i, i, fp, fp, i, i, i, i, C.c_float, vp, sz, vp]
    L.fn2_conv_k7s2_relu_supported.argtypes = [i, i, i, i]
    L.fn2_im2col_forward.argtypes = [fp, fp, i, i, i, i, i, i, i, vp]
    L.fn2_col2im_bias_relu_forward.argtypes = [fp, fp, fp, i, i, i, i, i, i, i, i, C.c_float, vp]
    L.fn2_col2im_bias_relu_forward_into.argtypes = [fp, fp, fp, i, i, i, i, i, i, i, i, C.c_float, i, i, vp]
    L.fn2_conv_k7s2_relu_forward.argtypes = [fp, fp, fp, fp, i, i, i, i, i, C.c_float, vp]
    L.fn2_conv_mfma_supported.argtypes = [i] * 7
    L.fn2_conv_mfma_packed_floats.argtypes = [i, i, i]
    L.fn2_conv_mfma_packed_floats.restype = sz
    L.fn2_conv_mfma_pack_weights.argtypes = [fp, fp, i, i, i, vp]
    L.fn2_conv_mfma_pack_weights_view.argtypes = [fp, fp, i, i, i, i, i, C.c_longlong, C.c_longlong, i, vp]
    L.fn2_conv_mfma_forward.argtypes = [fp, fp, fp, fp] + [i] * 13 + [C.c_float, vp]
    L.fn2_debug_set_conv_variant.argtypes = [i]
    L.fn2_conv_wino_supported.argtypes = [i] * 5
    L.fn2_conv_wino_packed_floats.argtypes = [i, i]
    L.fn2_conv_wino_packed_floats.restype = sz
    L.fn2_conv_wino_pack_weights.argtypes = [fp, fp, i, i, vp]
    L.fn2_conv_wino_forward.argtypes = [fp, fp, fp, fp] + [i] * 11 + [C.c_float, vp]
    L.fn2_debug_set_wino_variant.argtypes = [i]
    L.fn2_conv_plane_supported.argtypes = [i] * 7
    L.fn2_conv_plane_ksplit.argtypes = [i] * 7
    L.fn2_conv_plane_workspace_bytes.argtypes = [i] * 7
    L.fn2_conv_plane_workspace_bytes.restype = sz
    L.fn2_conv_plane_forward.argtypes = [fp, fp, fp, fp] + [i] * 12 + [C.c_float, vp, sz, vp]
    L.fn2_conv_k7s2_wgrad_supported.argtypes = [i] * 5
    L.fn2_conv_k7s2_wgrad_ksplit.argtypes = [i] * 5
    L.fn2_conv_k7s2_wgrad_workspace_bytes.argtypes = [i] * 5
    L.fn2_conv_k7s2_wgrad_workspace_bytes.restype = sz
    L.fn2_conv_k7s2_wgrad.argtypes = [fp, fp, fp] + [i] * 6 + [vp, sz, vp]
    L.fn2_conv_plane_k_supported.argtypes = [i] * 8
    L.fn2_conv_plane_k_ksplit.argtypes = [i] * 8
    L.fn2_conv_plane_k_workspace_bytes.argtypes = [i] * 8
    L.fn2_conv_plane_k_workspace_bytes.restype = sz
    L.fn2_conv_plane_k_forward.argtypes = [fp, fp, fp, fp] + [i] * 13 + [C.c_float, vp, sz, vp]
    L.fn2_debug_set_plane_variant.argtypes = [i]
    L.fn2_debug_set_plane_ksplit.argtypes = [i]
    L.fn2_set_batch_invariant.argtypes = [i]
    L.fn2_get_batch_invariant.argtypes = []
    L.fn2_tconv_supported.argtypes = [i] * 8
    L.fn2_tconv_forward.argtypes = [fp, fp, fp, fp] + [i] * 14 + [C.c_float, vp]
    L.fn2_debug_set_tconv_variant.argtypes = [i]
    L.fn2_debug_set_wgrad_buffers.argtypes = [i]
    L.fn2_debug_set_wgrad_chunk.argtypes = [i]
    L.fn2_conv_wgrad_supported.argtypes = [i] * 10
    L.fn2_conv_wgrad_ksplit.argtypes = [i] * 10
    L.fn2_conv_wgrad_workspace_bytes.argtypes = [i] * 10
    L.fn2_conv_wgrad_workspace_bytes.restype = sz
    L.fn2_conv_wgrad.argtypes = [fp, fp, fp] + [i] * 15 + [vp, sz, vp]
    L.fn2_deconv_plane_supported.argtypes = [i] * 5
    L.fn2_deconv_plane_ksplit.argtypes = [i] * 5
    L.fn2_deconv_plane_workspace_bytes.argtypes = [i] * 5
    L.fn2_deconv_plane_workspace_bytes.restype = sz
    L.fn2_deconv_plane_packed_floats.argtypes = [i, i]
    L.fn2_deconv_plane_packed_floats.restype = sz
    L.fn2_deconv_plane_pack_weights.argtypes = [fp, fp, i, i, vp]
    L.fn2_deconv_plane_pack_weights_k.argtypes = [fp, fp, i, i, i, vp]
    L.fn2_deconv_plane_forward.argtypes = [fp, fp, fp, fp] + [i] * 10 + [C.c_float, vp, sz, vp]
    L.fn2_caffemodel_index.argtypes = [vp, sz, C.POINTER(CaffemodelEntry), i, C.POINTER(C.c_int)]
    L.fn2_caffemodel_read_blob.argtypes = [vp, sz, C.POINTER(CaffemodelEntry), fp, sz]
    L.fn2_hdf5_index.argtypes = [vp, sz, C.POINTER(Hdf5Entry), i, C.POINTER(C.c_int)]
    L.fn2_hdf5_read_float.argtypes = [vp, sz, C.POINTER(Hdf5Entry), fp, sz]
    ip = C.POINTER(C.c_int)
    L.fn2_datum_parse.argtypes = [vp, sz, C.POINTER(DatumView)]
    L.fn2_datum_float_data.argtypes = [vp, sz, fp, sz]
    L.fn2_datum_serialize.argtypes = [i, i, i, vp, sz, i, vp, sz]
    L.fn2_datum_serialize.restype = C.c_longlong
    L.fn2_custom_data_sample_bytes.argtypes = [i, i, i, ip, i, ip, i]
    L.fn2_custom_data_sample_bytes.restype = sz
    L.fn2_custom_data_encode_sample.argtypes = [vp, vp, vp, vp, i, i, vp, sz]
    L.fn2_custom_data_stage_records.argtypes = [C.POINTER(C.c_void_p), C.POINTER(sz), i, vp, sz, ip, ip, ip, C.POINTER(sz), ip]
    L.fn2_custom_data_decode_forward.argtypes = [vp, sz, i, i, i, i, ip, i, ip, i, i, fp, C.c_float, C.POINTER(C.c_void_p), vp]
    L.fn2_augmentation_matrix.argtypes = [fp, i, i, i, i, i, fp]
    L.fn2_flow_augmentation_forward.argtypes = [fp, fp, fp, fp, i, i, i, i, i, vp]
    L.fn2_data_augmentation_workspace_bytes.argtypes = [i]
    L.fn2_data_augmentation_workspace_bytes.restype = sz
    L.fn2_data_augmentation_forward.argtypes = [C.POINTER(DataAugParams), fp, fp, fp, fp, i, i, i, i, vp, sz, vp]
    dp = C.POINTER(ConvDesc)
    for pre in ("fn2_conv", "fn2_deconv"):
        getattr(L, pre + "_route").argtypes = [dp, i]
        getattr(L, pre + "_packed_weight_floats").argtypes = [dp, i]
        getattr(L, pre + "_packed_weight_floats").restype = sz
        getattr(L, pre + "_pack_weights").argtypes = [dp, i, fp, fp, vp]
        getattr(L, pre + "_workspace_bytes").argtypes = [dp, i]
        getattr(L, pre + "_workspace_bytes").restype = sz
        getattr(L, pre + "_forward").argtypes = [dp, i, fp, i, i, fp, fp, fp, i, i, i, C.c_float, vp, sz, vp]
    L.fn2_conv_backward_data_route.argtypes = [dp, i]
    L.fn2_conv_backward_data_packed_weight_floats.argtypes = [dp, i, i]
    L.fn2_conv_backward_data_packed_weight_floats.restype = sz
    L.fn2_conv_backward_data_pack_workspace_bytes.argtypes = [dp, i, i]
    L.fn2_conv_backward_data_pack_workspace_bytes.restype = sz
    L.fn2_conv_backward_data_pack_weights.argtypes = [dp, i, i, fp, fp, vp, sz, vp]
    L.fn2_conv_backward_data_workspace_bytes.argtypes = [dp, i, i]
    L.fn2_conv_backward_data_workspace_bytes.restype = sz
    L.fn2_conv_backward_data_computed_channels.argtypes = [dp, i, i]
    L.fn2_conv_backward_data.argtypes = [dp, i, i, fp, i, i, fp, fp, i, i, i, vp, sz, vp]
    L.fn2_conv_backward_weights_supported.argtypes = [dp, i]
    L.fn2_conv_backward_weights_workspace_bytes.argtypes = [dp, i]
    L.fn2_conv_backward_weights_workspace_bytes.restype = sz
    L.fn2_conv_backward_weights.argtypes = [dp, i, fp, i, i, fp, i, i, fp, i, vp, sz, vp]
    L.fn2_conv_backward_data_workspace_bytes_with_room.argtypes = [dp, i, i, i]
    L.fn2_conv_backward_data_workspace_bytes_with_room.restype = sz
    L.fn2_conv_backward_data_masked_supported.argtypes = [dp, i, i]
    L.fn2_conv_backward_data_masked.argtypes = [dp, i, i, fp, i, i, fp, fp, i, i, fp, i, i, C.c_float, vp]
    L.fn2_conv_backward_weights_bias_fused.argtypes = [dp, i]
    L.fn2_conv_backward_weights_bias.argtypes = [dp, i, fp, fp, fp, fp, i, vp, sz, vp]
    L.fn2_conv_backward_bias.argtypes = [fp, i, i, fp, i, i, i, i, i, vp, sz, vp]
    if hasattr(L, "fn2_debug_set_correlation_impl"):
        L.fn2_debug_set_correlation_impl.argtypes = [i]
    if hasattr(L, "fn2_debug_correlation_units_plan"):
        L.fn2_debug_correlation_units_plan.argtypes = [i, i, i, i, C.POINTER(C.c_uint), i]
    if hasattr(L, "fn2_debug_set_resample_generic"):
        L.fn2_debug_set_resample_generic.argtypes = [i]
    for name in EXPORTS:
        if not hasattr(L, name):
            raise RuntimeError(f"{SO_PATH} does not export {name}")
        fn = getattr(L, name)
        if fn.restype is C.c_int:   # default restype: status code
            pass
    _lib = L
    return L


def check(status: int):
    if status != 0:
        raise Fn2Error(status, lib().fn2_last_error_string().decode())


def version() -> str:
    return lib().fn2_version().decode()
