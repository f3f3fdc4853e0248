"""ctypes binding of libd3f_hip.so (include/d3f_hip.h).

The library is the product path: there is no CPU / eager fallback.  If it cannot be loaded
the import of any op raises, loudly.
"""
import ctypes as C
import os
import re
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ["D3F_LIB"]) if os.environ.get("D3F_LIB") else _HERE / "csrc" / "libd3f_hip.so"  # D3F_LIB: profiling builds
HEADER_PATH = _HERE.parent / "include" / "d3f_hip.h"

F32, BF16, F32X3 = 0, 1, 2  # F32X3: fp32 tensors, contractions on the bf16 matrix pipe (exact 3-way split)


class D3FError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("B", "H", "W", "C0", "C1", "upsample0", "Cout", "KH", "KW", "stride", "pad", "CinReal")]


_p, _i, _i64, _f, _sz, _dbl = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t, C.c_double
_desc = C.POINTER(ConvDesc)

# name -> (restype, argtypes); must list every symbol include/d3f_hip.h declares
PROTOTYPES = {
    "d3f_version": (_i, []),
    "d3f_last_error": (C.c_char_p, []),
    "d3f_source_digest": (C.c_char_p, []),
    "d3f_profile_enable": (_i, [_i]),
    "d3f_profile_classes": (_i, [_i]),
    "d3f_profile_collect": (_i, [C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "d3f_unet_create": (_i, [C.c_char_p, _i, _i, _i, _i, _i, _i, C.POINTER(_p)]),
    "d3f_unet_destroy": (_i, [_p]),
    "d3f_unet_create_nets": (_i, [C.c_char_p, _i, _i, _i, _i, _i, _i, _i, _i, C.POINTER(_p)]),
    "d3f_unet_nets": (_i, [_p]),
    "d3f_unet_net_workspace_stride": (_sz, [_p]),
    "d3f_unet_pair_pack_weights": (_i, [_p, C.POINTER(_p), _p, _p]),
    "d3f_unet_pair_forward": (_i, [_p, C.POINTER(_p), C.POINTER(_p), C.POINTER(_p), C.POINTER(_p), _p, _p]),
    "d3f_unet_pair_backward": (_i, [_p, C.POINTER(_p), C.POINTER(_p), C.POINTER(_p), _p, _i, _i, _i, _p]),
    "d3f_unet_num_params": (_i, [_p]),
    "d3f_unet_param_info": (_i, [_p, _i, C.c_char_p, _i, C.POINTER(C.c_int32), C.POINTER(_i), C.POINTER(_i64)]),
    "d3f_unet_param_floats": (_i64, [_p]),
    "d3f_unet_num_bn": (_i, [_p]),
    "d3f_unet_bn_info": (_i, [_p, _i, C.c_char_p, _i, C.POINTER(_i), C.POINTER(_i64), C.POINTER(_i64)]),
    "d3f_unet_bnstat_floats": (_i64, [_p]),
    "d3f_unet_workspace_bytes": (_sz, [_p]),
    "d3f_unet_forward_flops": (_dbl, [_p]),
    "d3f_unet_backward_flops": (_dbl, [_p]),
    "d3f_unet_pack_weights": (_i, [_p, _p, _p, _p]),
    "d3f_unet_forward": (_i, [_p, _p, _p, _p, _p, _p, _i, _p]),
    "d3f_unet_forward_graph": (_i, [_p, _p, _p, _p, _p, _p, _p]),
    "d3f_noise_blend_fixed": (_i, [_p, _p, _p, _p, _i, _i64, _p]),
    "d3f_l1_per_image_workspace_bytes": (_sz, [_i]),
    "d3f_l1_per_image": (_i, [_p, _p, _p, _p, _i, _i64, _p]),
    "d3f_affine_warp": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "d3f_u8rgb_normalise": (_i, [_p, _p, _i, _i, _i, C.POINTER(_f), C.POINTER(_f), _p]),
    "d3f_unet_predict_u8": (_i, [_p, _p, _p, _p, _p, C.POINTER(_f), C.POINTER(_f), _p, _i, _p]),
    "d3f_unet_num_segments": (_i, [_p]),
    "d3f_unet_plan_counts": (_i, [_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "d3f_unet_segment_range": (_i, [_p, _i, C.POINTER(_i64), C.POINTER(_i64)]),
    "d3f_unet_backward": (_i, [_p, _p, _p, _p, _p, _i, _i, _p]),
    "d3f_unet_backward_nojoin": (_i, [_p, _p, _p, _p, _p, _i, _i, _p]),
    "d3f_unet_side_stream": (_i, [_p, C.POINTER(_p)]),
    "d3f_unet_backward_join": (_i, [_p, _p]),
    "d3f_adam_coefficients": (_i, [_f, _f, _f, _f, _i, _f, C.POINTER(_f)]),
    "d3f_unet_train_step": (_i, [_p, _p, _f, _f, _f, _p, _i, _p]),
    "d3f_unet_set_bn_sync": (_i, [_p, _p, _p, _i]),
    "d3f_unet_export": (_i, [_p, C.c_char_p, _p, _p, _p]),
    "d3f_unet_export_shape": (_i, [_p, C.c_char_p, C.POINTER(C.c_int32)]),
    "d3f_conv_upsample_folded": (_i, [_i, _p]),
    "d3f_conv_upsample_summed": (_i, [_i, _p]),
    "d3f_conv_packed_bytes": (_sz, [_i, _desc, _i]),
    "d3f_conv_pack_weights": (_i, [_i, _desc, _p, _p, _p, _p]),
    "d3f_conv_workspace_bytes": (_sz, [_i, _desc, _i]),
    "d3f_conv_stats_floats": (_sz, [_i, _desc, _i, C.POINTER(_i)]),
    "d3f_conv_forward": (_i, [_i, _desc, _p, _p, _p, _p, _p, _p, _p]),
    "d3f_conv_winograd_applies": (_i, [_i, _desc]),
    "d3f_conv_winograd_filter_bytes": (_sz, [_desc]),
    "d3f_conv_winograd_stats_floats": (_sz, [_desc, C.POINTER(_i)]),
    "d3f_conv_winograd_pack": (_i, [_desc, _p, _p, _p]),
    "d3f_conv_winograd_forward": (_i, [_desc, _p, _p, _p, _p, _p, _p, _p, _i, _p]),
    "d3f_conv_backward_data": (_i, [_i, _desc, _p, _p, _p, _p, _i, _i, _p, _p]),
    "d3f_conv_backward_weight_workspace_bytes": (_sz, [_i, _desc]),
    "d3f_conv_backward_weight": (_i, [_i, _desc, _p, _p, _p, _p, _p, _p]),
    "d3f_bn_finalize": (_i, [_p, _i, _i, _i64, _p, _p, _p, _p, _p, _p]),
    "d3f_bn_apply": (_i, [_i, _p, _p, _i, _i64, _p, _i, _p, _p]),
    "d3f_bn_backward_workspace_bytes": (_sz, [_i, _i, _i64]),
    "d3f_bn_backward": (_i, [_i, _p, _p, _p, _p, _p, _i, _i64, _p, _p, _p, _p, _p, _p]),
    "d3f_maxpool3x3s2_forward": (_i, [_i, _p, _p, _p, _i, _i, _i, _i, _p]),
    "d3f_maxpool3x3s2_backward": (_i, [_i, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "d3f_upsample2x_backward": (_i, [_i, _p, _p, _i, _i, _i, _i, _p]),
    "d3f_nchw_to_nhwc": (_i, [_i, _p, _p, _i, _i, _i, _i, _i, _p]),
    "d3f_nhwc_to_nchw": (_i, [_i, _p, _p, _i, _i, _i, _i, _i, _p]),
    "d3f_noise_blend": (_i, [_p, _p, _p, _f, _p, _p, _i, _i64, _p]),
    "d3f_mse_ssim_loss_workspace_bytes": (_sz, [_i, _i, _i]),
    "d3f_mse_ssim_loss": (_i, [_p, _p, _f, _f, _p, _p, _p, _i, _i, _i, _p]),
    "d3f_adam_step": (_i, [_p, _p, _p, _p, _i64, _f, _f, _f, _f, _i, _f, _p]),
    "d3f_ema_lerp": (_i, [_p, _p, _i64, _f, _p]),
}

# d3f_allreduce_fn: int (*)(void* ctx, float* data, int64_t count, void* stream)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)

_lib = None


def header_symbols():
    """Function names declared in include/d3f_hip.h."""
    text = HEADER_PATH.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(d3f_[a-z0-9_]+)\s*\(", text)))


def sources_present():
    """are the kernel sources and the public header the digest covers next to the library?"""
    return HEADER_PATH.exists() and any((_HERE / "csrc").glob("*.hip"))


def source_digest():
    """sha256 (first 16 hex digits) over the kernel sources csrc/*.hip, csrc/*.h and the public header -- ties a
    counter file under profiles/ to the code it was collected on (the GPU box has no .git to ask)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted((_HERE / "csrc").glob("*.hip")) + sorted((_HERE / "csrc").glob("*.h")) + [HEADER_PATH]
    for f in files:
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def lib():
    """Load (once) and return the ctypes handle; raises D3FError if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise D3FError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C {LIB_PATH.parent}` (hipcc, gfx950). There is no CPU fallback.")
    handle = C.CDLL(str(LIB_PATH))
    for name, (res, args) in PROTOTYPES.items():
        if os.environ.get("D3F_LIB") and not hasattr(handle, name):
            continue  # an older profiling / A-B build given explicitly: calling the missing entry point fails loudly
        fn = getattr(handle, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if not os.environ.get("D3F_LIB") and sources_present():
        # (a layout without the sources next to the library -- a non-editable install that ships only the .so -- has
        # nothing to compare against: the library is taken as built)
        built, here = handle.d3f_source_digest().decode(), source_digest()
        if built != here:
            raise D3FError(
                f"{LIB_PATH} was built from other sources (digest {built}) than the ones next to it ({here}): "
                f"rebuild it with `make -C {LIB_PATH.parent}`, or name a variant build explicitly through D3F_LIB")
    _lib = handle
    return _lib


def built_digest():
    """Digest baked into the loaded library (equals source_digest() unless D3F_LIB names a variant build)."""
    handle = lib()
    return handle.d3f_source_digest().decode() if hasattr(handle, "d3f_source_digest") else None


def check(rc):
    if rc != 0:
        raise D3FError(lib().d3f_last_error().decode("utf-8", "replace"))


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def ptr2(a, b):
    """the `T* const x[2]` argument of the pair entry points: the two networks' buffers"""
    return (C.c_void_p * 2)(a.data_ptr(), b.data_ptr())
