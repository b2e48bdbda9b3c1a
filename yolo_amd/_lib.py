"""ctypes binding of libymx.so (the HIP kernels behind the C ABI in include/ymx.h).

There is no CPU fallback: a missing library or a non-GPU tensor raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# YMX_LIBRARY names another build of the same library (diagnostic builds); the default is the
# one next to this file
LIB_PATH = os.environ.get("YMX_LIBRARY") or os.path.join(_HERE, "libymx.so")

F32, BF16 = 0, 1
ACT_NONE, ACT_LEAKY, ACT_MISH = 0, 1, 2
ACT_CODES = {None: ACT_NONE, "leaky": ACT_LEAKY, "mish": ACT_MISH}

_vp, _i, _f, _sz, _ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_longlong


class ymx_op(ctypes.Structure):
    """Mirror of ``struct ymx_op`` in include/ymx.h."""
    _fields_ = [(n, ctypes.c_int32) for n in
                ("kind", "B", "H", "W", "Cin", "Cout", "k", "stride", "act", "dtype", "x_dtype",
                 "out_dtype", "x_ld", "res_ld", "y_ld", "C1", "C2", "min_tiles")] + \
               [(n, ctypes.c_void_p) for n in ("x", "w", "scale", "shift", "residual", "y")]


# name -> (restype, argtypes); every symbol include/ymx.h declares
SIGNATURES = {
    "ymx_version": (_i, []),
    "ymx_last_error": (ctypes.c_char_p, []),
    "ymx_set_option": (_i, [ctypes.c_char_p, _ll]),
    "ymx_packed_weight_bytes": (_sz, [_i, _i, _i, _i]),
    "ymx_pack_weights": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "ymx_pack_weights_dgrad": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ymx_conv_wgrad_first_workspace_bytes": (_sz, [_i] * 5),
    "ymx_conv_wgrad_first": (_i, [_vp, _i, _vp, _i, _vp] + [_i] * 5 + [_vp, _sz, _vp]),
    "ymx_spp_concat_train": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ymx_spp_concat_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ymx_bn_workspace_bytes": (_sz, [_ll, _i]),
    "ymx_bn_stats": (_i, [_vp, _ll, _i, _i, _vp, _vp, _sz, _vp]),
    "ymx_bn_act_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _i, _vp]),
    "ymx_bn_act_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "ymx_bn_backward_apply": (_i, [_vp] * 6 + [_ll, _i, _i, _vp]),
    "ymx_pack_weights_pair": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ymx_pack_weights_dgrad_s2": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "ymx_conv2d_dgrad_s2": (_i, [_vp, _vp, _vp] + [_i] * 7 + [_vp]),
    "ymx_conv2d_fused": (_i, [_vp] * 6 + [_i] * 9 + [_vp]),
    "ymx_conv2d_fused_z": (_i, [_vp] * 7 + [_i] * 9 + [_vp]),
    "ymx_conv2d_fused_ex": (_i, [_vp] * 6 + [_i] * 14 + [_vp]),
    "ymx_upsample2x_concat": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ymx_conv_stem_fused": (_i, [_vp] * 8 + [_i] * 5 + [_vp]),
    "ymx_conv_resblock_fused": (_i, [_vp] * 8 + [_i] * 7 + [_vp]),
    "ymx_run_program": (_i, [ctypes.POINTER(ymx_op), _i, _vp]),
    "ymx_conv_kernel_label": (ctypes.c_char_p, [ctypes.POINTER(ymx_op)]),
    "ymx_decode_workspace_bytes": (_sz, [_i, _i, ctypes.POINTER(_i)]),
    "ymx_decode_filter": (_i, [ctypes.POINTER(_vp), _i, ctypes.POINTER(_i), _i, _i, _i,
                               ctypes.POINTER(_f), ctypes.POINTER(_f), ctypes.POINTER(_f),
                               _i, _f, _f, _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "ymx_decode_filter_strided": (_i, [ctypes.POINTER(_vp), ctypes.POINTER(_i), _i, ctypes.POINTER(_i), _i, _i, _i,
                               ctypes.POINTER(_f), ctypes.POINTER(_f), ctypes.POINTER(_f),
                               _i, _f, _f, _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "ymx_decode_split": (_i, [ctypes.POINTER(_vp), _i, ctypes.POINTER(_i), _i, _i, _i,
                              ctypes.POINTER(_f), ctypes.POINTER(_f), ctypes.POINTER(_f),
                              _vp, _vp, _vp, _vp, _i, _vp]),
    "ymx_filter_workspace_bytes": (_sz, [_i, _i]),
    "ymx_filter_boxes": (_i, [_vp, _vp, _ll, _ll, _ll, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _i,
                              _vp, _sz, _vp]),
    "ymx_nms_workspace_bytes": (_sz, [_i, _i, _i]),
    "ymx_nms_combined": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp,
                              _vp, _sz, _vp]),
    "ymx_nms_combined_static": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ymx_spp_concat": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ymx_yolo_loss_grad": (_i, [_vp, _i, _vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _i, _i, _i, _i, _f, _f, ctypes.POINTER(_f), _f,
                                _vp, _vp, _vp]),
    "ymx_act_apply": (_i, [_vp, _vp, _vp, ctypes.c_longlong, _i, _i, _vp]),
    "ymx_act_backward": (_i, [_vp, _vp, _vp, ctypes.c_longlong, _i, _i, _vp]),
    "ymx_act_backward_bias_workspace_bytes": (_sz, [ctypes.c_longlong, _i, _i, _i]),
    "ymx_act_backward_bias": (_i, [_vp, _vp, _vp, ctypes.c_longlong, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "ymx_bias_grad_workspace_bytes": (_sz, [ctypes.c_longlong, _i]),
    "ymx_bias_grad": (_i, [_vp, ctypes.c_longlong, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "ymx_conv_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "ymx_conv_wgrad": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "ymx_crc32c": (ctypes.c_uint, [ctypes.c_char_p, _sz, ctypes.c_uint]),
    "ymx_conv_wgrad_kernel_label": (ctypes.c_char_p, [_i] * 7),
    "ymx_upsample2x_backward": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ymx_letterbox": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _vp]),
    "ymx_draw_boxes": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _vp]),
    "ymx_draw_boxes_labeled": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _vp, _i, _i, _vp]),
    "ymx_yolo_loss_workspace_bytes": (_sz, [_i, _i]),
    "ymx_yolo_loss": (_i, [_vp, _i, _vp, _vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _i, _i, _i, _i, _f, _f, ctypes.POINTER(_f), _f, _vp, _vp, _vp, _sz, _vp]),
    "ymx_assign_labels": (_i, [_vp, _vp, _i, _i, _i, _i, ctypes.POINTER(_f), ctypes.POINTER(_f), ctypes.POINTER(_vp),
                               ctypes.POINTER(_vp), _vp]),
    "ymx_affine_act": (_i, [_vp, _vp, _vp, _vp, _ll, _i, _i, _i, _vp]),
    "ymx_bbox_iou": (_i, [_vp, _vp, _vp, _ll, _i, _vp]),
}

_lib = None


def lib():
    """The loaded library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "yolo_amd: %s is missing - build it with `make -C yolo_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc):
    """Turn a negative ymx return code into the matching Python exception."""
    if rc == 0:
        return
    msg = lib().ymx_last_error().decode("utf-8", "replace")
    if rc in (-1, -2):
        raise ValueError("ymx: " + msg)
    raise RuntimeError("ymx (%d): %s" % (rc, msg))


def dtype_code(dt):
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    raise ValueError("yolo_amd supports float32 and bfloat16 tensors, got %s" % dt)


def require_gpu(*tensors):
    """Product ops run on the GPU only; anything else is an error, not a fallback."""
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("yolo_amd: tensor on %s; the HIP path needs a GPU tensor "
                               "(there is no CPU fallback)" % t.device)


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream(ref=None):
    """The current HIP stream of ref's device (of the current device without ref)."""
    dev = None if ref is None else ref.device
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def launch(name, ref, *args):
    """Call the launching entry point ``name`` with ``args`` plus, as its last argument, the
    current stream of the device that tensor ``ref`` lives on, with that device current
    (kernel attributes and launches are per device), and raise on a negative return code."""
    fn = getattr(lib(), name)
    dev = ref.device
    if dev.index is None or dev.index == torch.cuda.current_device():
        check(fn(*args, stream(ref)))
    else:
        with torch.cuda.device(dev):
            check(fn(*args, stream(ref)))
