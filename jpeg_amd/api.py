"""Thin Python face of the C ABI (include/jpeg_amd.h).

Names follow the reference: quality tables as main.cpp builds them, planes,
blocks, zigzag coefficient planes.  Every compute call goes through
libjpeg_amd.so; nothing here computes pixels or coefficients on the CPU.
"""
import ctypes

import numpy as np

from . import _lib

# scan position i holds natural index ZIGZAG[i] (encode_decode.cpp:312-348)
ZIGZAG = np.array([
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5,
    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63], dtype=np.int64)

_BASE_LUMA = np.array([  # main.cpp:2-10
    16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55,
    14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
    18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99], dtype=np.int64)
_BASE_CHROMA = np.array([  # main.cpp:12-20
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
    24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99] + [99] * 32, dtype=np.int64)
_ANALYSIS_HI_LUMA = np.array([  # analysis.cpp:45-54
    3, 2, 2, 3, 5, 8, 10, 12, 2, 2, 3, 4, 5, 12, 12, 11,
    3, 3, 3, 5, 8, 11, 14, 11, 3, 3, 4, 6, 10, 17, 16, 12,
    4, 4, 7, 11, 14, 22, 21, 15, 5, 7, 11, 13, 16, 12, 23, 18,
    10, 13, 16, 17, 21, 24, 24, 21, 14, 18, 19, 20, 22, 20, 20, 20], dtype=np.int64)
_ANALYSIS_HI_CHROMA = np.array([  # analysis.cpp:56-65
    11, 12, 14, 19, 26, 58, 60, 55, 12, 18, 21, 28, 32, 57, 59, 56,
    14, 21, 25, 30, 59, 59, 59, 59, 19, 28, 30, 59, 59, 59, 59, 59,
    26, 32, 59, 59, 59, 59, 59, 59, 58, 57, 59, 59, 59, 59, 59, 59,
    60, 59, 59, 59, 59, 59, 59, 59, 55, 56, 59, 59, 59, 59, 59, 59], dtype=np.int64)


def quality_tables(quality):
    """main.cpp:51-80: (luma, chroma) int32[64] for quality 1..100."""
    quality = int(quality)
    if not 1 <= quality <= 100:
        raise ValueError("Quality must be between 1 and 100")  # main.cpp:36-39
    sf = 5000 // quality if quality < 50 else 200 - quality * 2
    scale = lambda t: np.clip((t * sf + 50) // 100, 1, 255).astype(np.int32)
    return scale(_BASE_LUMA), scale(_BASE_CHROMA)


def analysis_tables(quality):
    """analysis.cpp:43-90: the fixed (unscaled) table pairs."""
    hi = int(quality) >= 90
    return ((_ANALYSIS_HI_LUMA if hi else _BASE_LUMA).astype(np.int32),
            (_ANALYSIS_HI_CHROMA if hi else _BASE_CHROMA).astype(np.int32))


def init(device=0):
    _lib.check(_lib.load().jpegamd_init(int(device)))


def _q(t):
    t = np.ascontiguousarray(np.asarray(t, dtype=np.int32).reshape(64))
    return t, t.ctypes.data


def _stream(stream):
    if stream is None:
        import torch
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    return ctypes.c_void_p(int(stream))


def _dev(t, dtype, what, numel=None):
    """device pointer of a contiguous cuda tensor of `dtype` (and exactly `numel` elements: the
    C ABI takes bare pointers, a wrong size would be an out-of-bounds access, not an error)"""
    import torch
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise TypeError(f"{what}: expected a contiguous {dtype} tensor on the GPU")
    if numel is not None and t.numel() != numel:
        raise ValueError(f"{what}: {t.numel()} elements, expected {numel}")
    return t.data_ptr()


def _plane_sizes(rows, cols):
    if rows <= 0 or cols <= 0 or rows % 16 or cols % 16:
        raise _lib.JpegAmdError(f"rows={rows} cols={cols} must be positive multiples of 16 "
                                "(the reference resizes before this path, encode_decode.cpp:29-37)")
    return (rows // 8) * (cols // 8) * 64, (rows // 16) * (cols // 16) * 64


def encode_image(bgr, q_luma, q_chroma, out=None, stream=None):
    """bgr: cuda uint8 [rows, cols, 3] -> (coef_y, coef_cb, coef_cr) cuda int16
    [blocks, 64] in zigzag order (jpegamd_encode_image_dev)."""
    import torch
    if bgr.dim() != 3 or int(bgr.shape[2]) != 3:
        raise ValueError("encode_image: expected a [rows, cols, 3] image")
    rows, cols = int(bgr.shape[0]), int(bgr.shape[1])
    ny, nc = _plane_sizes(rows, cols)
    src = _dev(bgr, torch.uint8, "encode_image", rows * cols * 3)
    if out is None:
        out = tuple(torch.empty((n // 64, 64), dtype=torch.int16, device=bgr.device) for n in (ny, nc, nc))
    ql, pl = _q(q_luma)
    qc, pc = _q(q_chroma)
    _lib.check(_lib.load().jpegamd_encode_image_dev(
        src, rows, cols, cols * 3, pl, pc,
        _dev(out[0], torch.int16, "coef_y", ny), _dev(out[1], torch.int16, "coef_cb", nc),
        _dev(out[2], torch.int16, "coef_cr", nc), _stream(stream)))
    return out


def colour_planes(bgr, stream=None):
    """RGB2YCbCr + chromaSubsampling (encode_decode.cpp:39-75, 117-152) through the fused
    encoder's colour stage: cuda uint8 [rows, cols, 3] -> (Y [rows, cols], Cb, Cr [rows/2, cols/2])."""
    import torch
    if bgr.dim() != 3 or int(bgr.shape[2]) != 3:
        raise ValueError("colour_planes: expected a [rows, cols, 3] image")
    rows, cols = int(bgr.shape[0]), int(bgr.shape[1])
    _plane_sizes(rows, cols)
    src = _dev(bgr, torch.uint8, "colour_planes", rows * cols * 3)
    y = torch.empty((rows, cols), dtype=torch.uint8, device=bgr.device)
    cb = torch.empty((rows // 2, cols // 2), dtype=torch.uint8, device=bgr.device)
    cr = torch.empty_like(cb)
    _lib.check(_lib.load().jpegamd_colour_planes_dev(src, rows, cols, cols * 3, y.data_ptr(), cb.data_ptr(),
                                                     cr.data_ptr(), _stream(stream)))
    return y, cb, cr


def decode_image(coef_y, coef_cb, coef_cr, rows, cols, q_luma, q_chroma, out=None, stream=None):
    """inverse of encode_image -> cuda uint8 [rows, cols, 3] (jpegamd_decode_image_dev)."""
    import torch
    rows, cols = int(rows), int(cols)
    ny, nc = _plane_sizes(rows, cols)
    if out is None:
        out = torch.empty((rows, cols, 3), dtype=torch.uint8, device=coef_y.device)
    ql, pl = _q(q_luma)
    qc, pc = _q(q_chroma)
    _lib.check(_lib.load().jpegamd_decode_image_dev(
        _dev(coef_y, torch.int16, "coef_y", ny), _dev(coef_cb, torch.int16, "coef_cb", nc),
        _dev(coef_cr, torch.int16, "coef_cr", nc), rows, cols, pl, pc,
        _dev(out, torch.uint8, "bgr", rows * cols * 3), cols * 3, _stream(stream)))
    return out


def encode_images(batch, q_luma, q_chroma, stream=None):
    """batch: cuda uint8 [n, rows, cols, 3], contiguous -> (coef_y, coef_cb, coef_cr) cuda int16
    [n, blocks, 64].  MCU rows are independent and an image is a whole number of them, so a
    batch of equally sized images IS one image of n * rows rows: ONE launch for all of them (a
    1024^2 image alone fills 128 of the chip's 3072 wave slots)."""
    n, rows, cols = int(batch.shape[0]), int(batch.shape[1]), int(batch.shape[2])
    if not batch.is_contiguous():
        raise ValueError("encode_images: the batch must be contiguous")
    cy, cb, cr = encode_image(batch.view(n * rows, cols, 3), q_luma, q_chroma, stream=stream)
    return cy.view(n, -1, 64), cb.view(n, -1, 64), cr.view(n, -1, 64)


def decode_images(coef_y, coef_cb, coef_cr, rows, cols, q_luma, q_chroma, stream=None):
    """inverse of encode_images: coefficient planes [n, blocks, 64] -> cuda uint8 [n, rows, cols, 3]"""
    n = int(coef_y.shape[0])
    if not (coef_y.is_contiguous() and coef_cb.is_contiguous() and coef_cr.is_contiguous()):
        raise ValueError("decode_images: the coefficient planes must be contiguous")
    out = decode_image(coef_y.view(-1, 64), coef_cb.view(-1, 64), coef_cr.view(-1, 64), n * int(rows), int(cols),
                       q_luma, q_chroma, stream=stream)
    return out.view(n, int(rows), int(cols), 3)


def encode_plane(plane, q, stream=None):
    """plane: cuda uint8 [rows, cols] (any size) -> cuda int16 [blocks, 64] zigzag."""
    import torch
    rows, cols = int(plane.shape[0]), int(plane.shape[1])
    nb = ((rows + 7) // 8) * ((cols + 7) // 8)
    out = torch.empty((nb, 64), dtype=torch.int16, device=plane.device)
    qq, pq = _q(q)
    _lib.check(_lib.load().jpegamd_encode_plane_dev(
        _dev(plane, torch.uint8, "plane", rows * cols), rows, cols, cols, pq, out.data_ptr(), _stream(stream)))
    return out


def decode_plane(coef, rows, cols, q, stream=None):
    import torch
    rows, cols = int(rows), int(cols)
    out = torch.zeros((rows, cols), dtype=torch.uint8, device=coef.device)  # Mat::zeros, :793
    qq, pq = _q(q)
    _lib.check(_lib.load().jpegamd_decode_plane_dev(
        _dev(coef, torch.int16, "coef", ((rows + 7) // 8) * ((cols + 7) // 8) * 64), rows, cols, pq, out.data_ptr(),
        cols, _stream(stream)))
    return out


def fdct_quant_blocks(blocks, q):
    """kernel.cuh dctQuantizationParallel on flat host arrays: int32 [nb, 64] -> float32 [nb, 64]."""
    blocks = np.ascontiguousarray(blocks, dtype=np.int32).reshape(-1, 64)
    out = np.empty(blocks.shape, dtype=np.float32)
    qq, pq = _q(q)
    _lib.check(_lib.load().jpegamd_fdct_quant_blocks(blocks.ctypes.data, blocks.shape[0], pq, out.ctypes.data))
    return out


def dequant_idct_blocks(coef, q):
    """kernel.cuh dequantizeInverseDCTParallel on flat host arrays: float32 [nb, 64] -> int32 [nb, 64]."""
    coef = np.ascontiguousarray(coef, dtype=np.float32).reshape(-1, 64)
    out = np.empty(coef.shape, dtype=np.int32)
    qq, pq = _q(q)
    _lib.check(_lib.load().jpegamd_dequant_idct_blocks(coef.ctypes.data, coef.shape[0], pq, out.ctypes.data))
    return out


def time_image(bgr, coefs, bgr_out, q_luma, q_chroma, iters, stream=None):
    """HIP-event timing of `iters` steps (encode, decode, encode, ...) on the stream they run
    on, one event between every two launches; returns the mean (enc_ms, dec_ms) per launch."""
    import torch
    rows, cols = int(bgr.shape[0]), int(bgr.shape[1])
    ql, pl = _q(q_luma)
    qc, pc = _q(q_chroma)
    e, d = ctypes.c_float(0), ctypes.c_float(0)
    ny, nc = _plane_sizes(rows, cols)
    _lib.check(_lib.load().jpegamd_time_image_dev(
        _dev(bgr, torch.uint8, "bgr", rows * cols * 3), rows, cols, cols * 3, pl, pc,
        _dev(coefs[0], torch.int16, "coef_y", ny), _dev(coefs[1], torch.int16, "coef_cb", nc),
        _dev(coefs[2], torch.int16, "coef_cr", nc),
        _dev(bgr_out, torch.uint8, "bgr_out", rows * cols * 3), int(iters), _stream(stream),
        ctypes.byref(e), ctypes.byref(d)))
    return e.value, d.value


def _sync(stream):
    import torch
    if stream is None:
        torch.cuda.current_stream().synchronize()
    else:
        torch.cuda.ExternalStream(int(stream)).synchronize()


def rle_encode_plane(coef, stream=None):
    """run_length_encode over a zigzag coefficient plane (encode_decode.cpp:350-373,
    :643-645): cuda int16 [nb, 64] -> (pairs cuda int32 [n, 2] = (value, run),
    block_offsets cuda int32 [nb + 1]) via jpegamd_rle_encode_plane_dev."""
    (pairs, offs), = rle_encode_planes([coef], stream=stream)
    return pairs, offs


def rle_encode_planes(coefs, stream=None):
    """The same for one plane (jpegamd_rle_encode_plane_dev) or the three planes of an image in
    ONE launch (jpegamd_rle_encode_image_dev): list of cuda int16 [nb, 64] -> list of
    (pairs, block_offsets)."""
    import torch
    lib = _lib.load()
    if len(coefs) not in (1, 3):
        raise ValueError("rle_encode_planes: one plane or three")
    nbs = [int(c.shape[0]) for c in coefs]
    srcs = [_dev(c, torch.int16, "rle_encode_planes", nb * 64) for c, nb in zip(coefs, nbs)]
    dev = coefs[0].device
    work = torch.empty(lib.jpegamd_rle_workspace_bytes(sum(nbs)), dtype=torch.uint8, device=dev)
    offs = [torch.empty(nb + 1, dtype=torch.int32, device=dev) for nb in nbs]
    caps = [min(nb * 64, max(1024, nb * 12)) for nb in nbs]  # typical planes need far less than the worst case
    while True:
        pairs = [torch.empty((cap, 2), dtype=torch.int32, device=dev) for cap in caps]
        if len(coefs) == 1:
            _lib.check(lib.jpegamd_rle_encode_plane_dev(srcs[0], nbs[0], pairs[0].data_ptr(), caps[0],
                                                        offs[0].data_ptr(), work.data_ptr(), work.numel(),
                                                        _stream(stream)))
        else:
            arr = lambda vals, t: (t * 3)(*vals)
            _lib.check(lib.jpegamd_rle_encode_image_dev(
                arr(srcs, ctypes.c_void_p), arr(nbs, ctypes.c_size_t),
                arr([p.data_ptr() for p in pairs], ctypes.c_void_p), arr(caps, ctypes.c_size_t),
                arr([o.data_ptr() for o in offs], ctypes.c_void_p), work.data_ptr(), work.numel(), _stream(stream)))
        _sync(stream)  # the totals are read on the host: wait for the stream the kernels ran on
        totals = [int(o[nb].item()) & 0xFFFFFFFF for o, nb in zip(offs, nbs)]
        if all(t <= c for t, c in zip(totals, caps)):
            return [(p[:t], o) for p, t, o in zip(pairs, totals, offs)]
        caps = [max(t, c) for t, c in zip(totals, caps)]


def encode_image_rle(bgr, q_luma, q_chroma, stream=None):
    """jpegamd_encode_image_rle_dev: fused encode + run-length stage of the three planes, the
    pair counts handed from the encoder to the run-length kernels.  Returns
    ((coef_y, coef_cb, coef_cr), [(pairs, block_offsets)] * 3)."""
    import torch
    lib = _lib.load()
    if bgr.dim() != 3 or int(bgr.shape[2]) != 3:
        raise ValueError("encode_image_rle: expected a [rows, cols, 3] image")
    rows, cols = int(bgr.shape[0]), int(bgr.shape[1])
    _plane_sizes(rows, cols)
    src = _dev(bgr, torch.uint8, "encode_image_rle", rows * cols * 3)
    dev = bgr.device
    nbs = [(rows // 8) * (cols // 8)] + [(rows // 16) * (cols // 16)] * 2
    coefs = tuple(torch.empty((n, 64), dtype=torch.int16, device=dev) for n in nbs)
    work = torch.empty(lib.jpegamd_rle_workspace_bytes(sum(nbs)) + 256, dtype=torch.uint8, device=dev)
    wptr = (work.data_ptr() + 255) & ~255
    offs = [torch.empty(nb + 1, dtype=torch.int32, device=dev) for nb in nbs]
    caps = [min(nb * 64, max(1024, nb * 12)) for nb in nbs]
    ql, pl = _q(q_luma)
    qc, pc = _q(q_chroma)
    arr = lambda vals, t: (t * 3)(*vals)
    while True:
        pairs = [torch.empty((cap, 2), dtype=torch.int32, device=dev) for cap in caps]
        _lib.check(lib.jpegamd_encode_image_rle_dev(
            src, rows, cols, cols * 3, pl, pc, arr([c.data_ptr() for c in coefs], ctypes.c_void_p),
            arr([p.data_ptr() for p in pairs], ctypes.c_void_p), arr(caps, ctypes.c_size_t),
            arr([o.data_ptr() for o in offs], ctypes.c_void_p), wptr, work.numel() - 256, _stream(stream)))
        _sync(stream)
        totals = [int(o[nb].item()) & 0xFFFFFFFF for o, nb in zip(offs, nbs)]
        if all(t <= c for t, c in zip(totals, caps)):
            return coefs, [(p[:t], o) for p, t, o in zip(pairs, totals, offs)]
        caps = [max(t, c) for t, c in zip(totals, caps)]


def rle_decode_plane(pairs, nb, stream=None):
    """decodeRLE (encode_decode.cpp:484-518): cuda int32 [n, 2] pairs -> cuda int16 [nb, 64]."""
    import torch
    lib = _lib.load()
    n = int(pairs.shape[0])
    src = _dev(pairs, torch.int32, "rle_decode_plane") if n else 0
    work = torch.empty(lib.jpegamd_rle_workspace_bytes(nb), dtype=torch.uint8, device=pairs.device)
    out = torch.empty((nb, 64), dtype=torch.int16, device=pairs.device)
    _lib.check(lib.jpegamd_rle_decode_plane_dev(src, n, int(nb), out.data_ptr(), work.data_ptr(),
                                                work.numel(), _stream(stream)))
    return out


def bin_encode(sym_y, sym_cb, sym_cr, y_rows, y_cols):
    """Three (value, run) symbol streams (numpy int32, flattened) -> the bytes of the
    reference's .bin container (jpegamd_bin_encode; host only, no GPU)."""
    lib = _lib.load()
    s = [np.ascontiguousarray(np.asarray(x).reshape(-1), dtype=np.int32) for x in (sym_y, sym_cb, sym_cr)]
    need = ctypes.c_size_t(0)
    args = [v for x in s for v in (x.ctypes.data, x.size)] + [int(y_rows), int(y_cols), int(y_rows) // 2, int(y_cols) // 2]
    lib.jpegamd_bin_encode(*args, None, 0, ctypes.byref(need))  # sizing call
    out = np.empty(need.value, np.uint8)
    _lib.check(lib.jpegamd_bin_encode(*args, out.ctypes.data, out.size, ctypes.byref(need)))
    return out[:need.value].tobytes()


def bin_encode_dev(sym_y, sym_cb, sym_cr, y_rows, y_cols):
    """The same container from cuda int32 symbol streams (e.g. the flattened pairs of
    rle_encode_plane): both passes over the symbols run on the GPU (jpegamd_bin_encode_dev)."""
    import torch
    lib = _lib.load()
    s = [x.reshape(-1) for x in (sym_y, sym_cb, sym_cr)]
    ptrs = [_dev(x, torch.int32, "bin_encode_dev") if x.numel() else 0 for x in s]
    need = ctypes.c_size_t(0)
    args = [v for p, x in zip(ptrs, s) for v in (p, x.numel())] + [int(y_rows), int(y_cols), int(y_rows) // 2, int(y_cols) // 2]
    # ONE call (a sizing call would run the whole stage twice): a symbol's code is at most 32 bits here, a
    # dictionary at most 64 Ki entries of 8 bytes per plane; the pages of the unused rest are never touched
    out = np.empty(48 + 4 * sum(x.numel() for x in s) + 3 * 8 * 65536, np.uint8)
    _lib.check(lib.jpegamd_bin_encode_dev(*args, out.ctypes.data, out.size, ctypes.byref(need)))
    return out[:need.value].tobytes()


def bin_decode(data):
    """.bin bytes -> (dims[6], [sym_y, sym_cb, sym_cr]) (jpegamd_bin_decode; host only)."""
    lib = _lib.load()
    buf = np.frombuffer(data, np.uint8)
    dims = np.zeros(6, np.int32)
    n = [ctypes.c_size_t(0) for _ in range(3)]
    syms = [np.empty(0, np.int32) for _ in range(3)]
    for _ in range(4):  # sizes are reported plane by plane
        args = [v for x, c in zip(syms, n) for v in (x.ctypes.data, x.size, ctypes.byref(c))]
        rc = lib.jpegamd_bin_decode(buf.ctypes.data, buf.size, dims.ctypes.data, *args)
        if rc == 0:
            return dims, [x[:c.value].copy() for x, c in zip(syms, n)]
        if all(c.value <= x.size for x, c in zip(syms, n)):
            _lib.check(rc)
        syms = [np.empty(max(c.value, x.size), np.int32) for x, c in zip(syms, n)]
    _lib.check(rc)


def bin_decode_dev(data, device="cuda:0"):
    """.bin bytes -> (dims[6], [sym_y, sym_cb, sym_cr] as cuda int32 tensors): huffman_decode on the
    device (jpegamd_bin_decode_dev).  ONE call: the buffers are sized from the header -- a plane of nb
    blocks holds at most 64 pairs per block (plus what up to 7 pad bits decode to) and no more symbols
    than payload bits (what jpegamd_decompress_image does too); a file that decodes to more than that is
    not an image of its header's size and gets the exact-size second call."""
    import torch
    lib = _lib.load()
    buf = np.frombuffer(data, np.uint8)
    dims = np.zeros(6, np.int32)
    n = (ctypes.c_size_t * 3)(0, 0, 0)
    caps = [1, 1, 1]
    if buf.size >= 36:
        head = np.frombuffer(buf[:36].tobytes(), np.int32)
        for k in range(3):
            r, c, ln = int(head[2 * k]), int(head[2 * k + 1]), int(head[6 + k])
            if 0 < r <= 1 << 20 and 0 < c <= 1 << 20 and ln >= 0:
                caps[k] = max(1, min(((r + 7) // 8) * ((c + 7) // 8) * 128 + 16, ln * 8))
    for _ in range(2):
        syms = [torch.empty(c, dtype=torch.int32, device=device) for c in caps]
        ptrs = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in syms])
        rc = lib.jpegamd_bin_decode_dev(buf.ctypes.data, buf.size, dims.ctypes.data, ptrs, (ctypes.c_size_t * 3)(*caps), n)
        if rc == 0:
            return dims, [t[:c] for t, c in zip(syms, n)]
        if all(c <= cap for c, cap in zip(n, caps)):
            _lib.check(rc)
        caps = [max(int(c), 1) for c in n]
    _lib.check(rc)


def compress_image(bgr, q_luma, q_chroma):
    """numpy uint8 [rows, cols, 3] on the HOST -> .bin bytes (jpegamd_compress_image)."""
    lib = _lib.load()
    bgr = np.ascontiguousarray(bgr, np.uint8)
    ql, pl = _q(q_luma)
    qc, pc = _q(q_chroma)
    out, n = ctypes.c_void_p(), ctypes.c_size_t(0)
    _lib.check(lib.jpegamd_compress_image(bgr.ctypes.data, bgr.shape[0], bgr.shape[1], bgr.shape[1] * 3, pl, pc,
                                          ctypes.byref(out), ctypes.byref(n)))
    try:
        return ctypes.string_at(out, n.value)
    finally:
        lib.jpegamd_free(out)


def decompress_image(data, q_luma, q_chroma):
    """.bin bytes -> numpy uint8 [rows, cols, 3] on the HOST (jpegamd_decompress_image)."""
    lib = _lib.load()
    buf = np.frombuffer(data, np.uint8)
    ql, pl = _q(q_luma)
    qc, pc = _q(q_chroma)
    r, c = ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(lib.jpegamd_decompress_image(buf.ctypes.data, buf.size, pl, pc, None, 0, 0,
                                            ctypes.byref(r), ctypes.byref(c)))
    out = np.empty((r.value, c.value, 3), np.uint8)
    _lib.check(lib.jpegamd_decompress_image(buf.ctypes.data, buf.size, pl, pc, out.ctypes.data, out.size,
                                            c.value * 3, ctypes.byref(r), ctypes.byref(c)))
    return out


class Context:
    """jpegamd_context: compress_image / decompress_image with device buffers kept from call
    to call (use as a context manager, or call close())."""

    def __init__(self, device=0):
        self._ctx = ctypes.c_void_p()  # (set before anything can raise: __del__ looks at it)
        self._lib = _lib.load()
        _lib.check(self._lib.jpegamd_context_create(int(device), ctypes.byref(self._ctx)))

    def close(self):
        if self._ctx:
            self._lib.jpegamd_context_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        self.close()

    def compress_image(self, bgr, q_luma, q_chroma):
        bgr = np.ascontiguousarray(bgr, np.uint8)
        ql, pl = _q(q_luma)
        qc, pc = _q(q_chroma)
        out, n = ctypes.c_void_p(), ctypes.c_size_t(0)
        _lib.check(self._lib.jpegamd_context_compress_image(self._ctx, bgr.ctypes.data, bgr.shape[0], bgr.shape[1],
                                                            bgr.shape[1] * 3, pl, pc, ctypes.byref(out),
                                                            ctypes.byref(n)))
        try:
            return ctypes.string_at(out, n.value)
        finally:
            self._lib.jpegamd_free(out)

    def decompress_image(self, data, q_luma, q_chroma, out=None):
        buf = np.frombuffer(data, np.uint8)
        ql, pl = _q(q_luma)
        qc, pc = _q(q_chroma)
        r, c = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(self._lib.jpegamd_context_decompress_image(self._ctx, buf.ctypes.data, buf.size, pl, pc, None, 0, 0,
                                                              ctypes.byref(r), ctypes.byref(c)))
        if out is None:
            out = np.empty((r.value, c.value, 3), np.uint8)
        if not (isinstance(out, np.ndarray) and out.dtype == np.uint8 and out.flags.c_contiguous
                and out.size == r.value * c.value * 3):
            raise ValueError("decompress_image: `out` must be a C-contiguous uint8 array of rows x cols x 3")
        _lib.check(self._lib.jpegamd_context_decompress_image(self._ctx, buf.ctypes.data, buf.size, pl, pc,
                                                              out.ctypes.data, out.size, c.value * 3,
                                                              ctypes.byref(r), ctypes.byref(c)))
        return out
