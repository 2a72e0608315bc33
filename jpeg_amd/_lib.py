"""ctypes loader for libjpeg_amd.so (the C ABI of include/jpeg_amd.h).

There is no CPU fallback: if the library is missing or no gfx950 device is
usable, loading or init raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libjpeg_amd.so")

_c = ctypes
_u8p, _i16p, _i32p, _f32p = (_c.POINTER(t) for t in (_c.c_uint8, _c.c_int16, _c.c_int32, _c.c_float))

# name -> argtypes; every function returns int except last_error
SIGNATURES = {
    "jpegamd_init": [_c.c_int],
    "jpegamd_abi_version": [],
    "jpegamd_fdct_quant_blocks": [_c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_void_p],
    "jpegamd_dequant_idct_blocks": [_c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_void_p],
    "jpegamd_fdct_quant_blocks_dev": [_c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_void_p, _c.c_void_p],
    "jpegamd_dequant_idct_blocks_dev": [_c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_void_p, _c.c_void_p],
    "jpegamd_encode_plane_dev": [_c.c_void_p, _c.c_int, _c.c_int, _c.c_size_t, _c.c_void_p, _c.c_void_p, _c.c_void_p],
    "jpegamd_decode_plane_dev": [_c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p],
    "jpegamd_encode_image_dev": [_c.c_void_p, _c.c_int, _c.c_int, _c.c_size_t, _c.c_void_p, _c.c_void_p,
                                 _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p],
    "jpegamd_colour_planes_dev": [_c.c_void_p, _c.c_int, _c.c_int, _c.c_size_t, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                  _c.c_void_p],
    "jpegamd_decode_image_dev": [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p,
                                 _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p],
    "jpegamd_encode_image": [_c.c_void_p, _c.c_int, _c.c_int, _c.c_size_t, _c.c_void_p, _c.c_void_p,
                             _c.c_void_p, _c.c_void_p, _c.c_void_p],
    "jpegamd_decode_image": [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p,
                             _c.c_void_p, _c.c_void_p, _c.c_size_t],
    "jpegamd_rle_encode_plane_dev": [_c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_void_p,
                                     _c.c_size_t, _c.c_void_p],
    "jpegamd_rle_encode_image_dev": [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                     _c.c_size_t, _c.c_void_p],
    "jpegamd_encode_image_rle_dev": [_c.c_void_p, _c.c_int, _c.c_int, _c.c_size_t, _c.c_void_p, _c.c_void_p,
                                     _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_size_t,
                                     _c.c_void_p],
    "jpegamd_rle_encode_plane_noalloc_dev": [_c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_size_t, _c.c_void_p,
                                             _c.c_void_p],
    "jpegamd_rle_decode_plane_dev": [_c.c_void_p, _c.c_size_t, _c.c_size_t, _c.c_void_p, _c.c_void_p,
                                     _c.c_size_t, _c.c_void_p],
    "jpegamd_bin_encode": [_c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_size_t,
                           _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_size_t,
                           _c.POINTER(_c.c_size_t)],
    "jpegamd_bin_encode_dev": [_c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_size_t,
                               _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_size_t,
                               _c.POINTER(_c.c_size_t)],
    "jpegamd_bin_decode": [_c.c_void_p, _c.c_size_t, _c.c_void_p,
                           _c.c_void_p, _c.c_size_t, _c.POINTER(_c.c_size_t),
                           _c.c_void_p, _c.c_size_t, _c.POINTER(_c.c_size_t),
                           _c.c_void_p, _c.c_size_t, _c.POINTER(_c.c_size_t)],
    "jpegamd_bin_decode_dev": [_c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p],
    "jpegamd_quality_tables": [_c.c_int, _c.c_void_p, _c.c_void_p],
    "jpegamd_analysis_tables": [_c.c_int, _c.c_void_p, _c.c_void_p],
    "jpegamd_compress_image": [_c.c_void_p, _c.c_int, _c.c_int, _c.c_size_t, _c.c_void_p, _c.c_void_p,
                               _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_size_t)],
    "jpegamd_decompress_image": [_c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_size_t,
                                 _c.c_size_t, _c.POINTER(_c.c_int), _c.POINTER(_c.c_int)],
    "jpegamd_context_create": [_c.c_int, _c.POINTER(_c.c_void_p)],
    "jpegamd_context_compress_image": [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_size_t, _c.c_void_p,
                                       _c.c_void_p, _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_size_t)],
    "jpegamd_context_decompress_image": [_c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_void_p,
                                         _c.c_void_p, _c.c_size_t, _c.c_size_t, _c.POINTER(_c.c_int),
                                         _c.POINTER(_c.c_int)],
    "jpegamd_time_image_dev": [_c.c_void_p, _c.c_int, _c.c_int, _c.c_size_t, _c.c_void_p, _c.c_void_p,
                               _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p,
                               _c.POINTER(_c.c_float), _c.POINTER(_c.c_float)],
}

# the exported functions that do not return an int status (set up by hand in load())
OTHER_RETURN_TYPES = ("jpegamd_last_error", "jpegamd_rle_workspace_bytes", "jpegamd_rle_noalloc_max_blocks",
                      "jpegamd_free", "jpegamd_context_destroy")

_lib = None


class JpegAmdError(RuntimeError):
    pass


def load():
    """Load the shared library (no GPU needed for this step)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise JpegAmdError(
                f"{LIB_PATH} is missing: build it with `make -C jpeg_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback")
        # torch ships a HIP runtime of its own.  The library must bind to the SAME runtime as
        # the tensors whose device pointers it is handed, and a process that loads the system's
        # runtime first and torch's second ends up with two: the second one then finds "no
        # ROCm-capable device".  Importing torch first makes the loader resolve
        # libamdhip64.so.7 to the copy that is already there.
        import torch  # noqa: F401
        lib = _c.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
            fn.argtypes = argtypes
            fn.restype = _c.c_int
        lib.jpegamd_rle_workspace_bytes.argtypes = [_c.c_size_t]
        lib.jpegamd_rle_workspace_bytes.restype = _c.c_size_t
        lib.jpegamd_rle_noalloc_max_blocks.argtypes = []
        lib.jpegamd_rle_noalloc_max_blocks.restype = _c.c_size_t
        lib.jpegamd_free.argtypes = [_c.c_void_p]
        lib.jpegamd_free.restype = None
        lib.jpegamd_context_destroy.argtypes = [_c.c_void_p]
        lib.jpegamd_context_destroy.restype = None
        lib.jpegamd_last_error.argtypes = []
        lib.jpegamd_last_error.restype = _c.c_char_p
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        raise JpegAmdError(load().jpegamd_last_error().decode("utf-8", "replace"))
