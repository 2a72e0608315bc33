"""ctypes faces of the TEST-ONLY libraries: oracle/liboracle.so (CPU restatement
of the reference), oracle/_ref/libjpegref.so (the reference's own
jpeg_concept/jpeg_function.cpp, built where it lies) and
tests/model/libjmmodel.so (host build of the kernels' arithmetic)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libjpegref.so")
MODEL_SO = os.path.join(ROOT, "tests", "model", "libjmmodel.so")

ZIGZAG = np.array([
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5,
    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63], dtype=np.int64)


def _p(a):
    return C.c_void_p(a.ctypes.data)


def _build(so, directory, refresh=False):
    """refresh: run make even when the library exists (its Makefile knows what it is built from)"""
    import subprocess
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", directory], stdout=subprocess.DEVNULL)
    elif refresh:
        subprocess.call(["make", "-C", directory], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return C.CDLL(so)


class Oracle:
    def __init__(self):
        self.lib = L = _build(ORACLE_SO, os.path.join(ROOT, "oracle"))
        L.orc_encode_plane.restype = C.c_size_t
        L.orc_decode_rle.restype = C.c_size_t
        L.orc_write_bin.restype = C.c_size_t
        L.orc_num_blocks.restype = C.c_size_t
        L.orc_mse.restype = C.c_double
        L.orc_psnr.restype = C.c_double
        L.orc_psnr.argtypes = [C.c_double]
        L.orc_init()

    def quality_tables(self, qf):
        a, b = np.zeros(64, np.int32), np.zeros(64, np.int32)
        self.lib.orc_quality_tables(int(qf), _p(a), _p(b))
        return a, b

    def analysis_tables(self, qf):
        a, b = np.zeros(64, np.int32), np.zeros(64, np.int32)
        self.lib.orc_analysis_tables(int(qf), _p(a), _p(b))
        return a, b

    def rgb2ycbcr(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        out = np.empty_like(img)
        self.lib.orc_rgb2ycbcr(_p(img), img.shape[0], img.shape[1], _p(out))
        return out

    def ycbcr2rgb(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        out = np.empty_like(img)
        self.lib.orc_ycbcr2rgb(_p(img), img.shape[0], img.shape[1], _p(out))
        return out

    def chroma_subsample(self, ycc):
        r, c = ycc.shape[:2]
        y = np.empty((r, c), np.uint8)
        cb = np.empty((r // 2, c // 2), np.uint8)
        cr = np.empty_like(cb)
        self.lib.orc_chroma_subsample(_p(np.ascontiguousarray(ycc)), r, c, _p(y), _p(cb), _p(cr))
        return y, cb, cr

    def encode_plane(self, plane, q, flavour=0, want_symbols=True):
        plane = np.ascontiguousarray(plane, np.uint8)
        r, c = plane.shape
        nb = ((r + 7) // 8) * ((c + 7) // 8)
        coef = np.empty((nb, 64), np.int16)
        sym = np.empty(nb * 128, np.int32) if want_symbols else None
        q = np.ascontiguousarray(q, np.int32)
        n = self.lib.orc_encode_plane(_p(plane), r, c, _p(q), int(flavour), _p(coef),
                                      _p(sym) if want_symbols else None)
        return coef, (sym[:n].copy() if want_symbols else None)

    def decode_plane(self, coef, rows, cols, q, flavour=0):
        coef = np.ascontiguousarray(coef, np.int16)
        out = np.empty((rows, cols), np.uint8)
        q = np.ascontiguousarray(q, np.int32)
        self.lib.orc_decode_plane(_p(coef), int(rows), int(cols), _p(q), int(flavour), _p(out))
        return out

    def encode_image(self, bgr, qy, qc, flavour=0, threads=8):
        bgr = np.ascontiguousarray(bgr, np.uint8)
        r, c = bgr.shape[:2]
        cy = np.empty(((r // 8) * (c // 8), 64), np.int16)
        cb = np.empty(((r // 16) * (c // 16), 64), np.int16)
        cr = np.empty_like(cb)
        qy, qc = np.ascontiguousarray(qy, np.int32), np.ascontiguousarray(qc, np.int32)
        rc = self.lib.orc_encode_image(_p(bgr), r, c, _p(qy), _p(qc), int(flavour), int(threads),
                                       _p(cy), _p(cb), _p(cr), None)
        assert rc == 0
        return cy, cb, cr

    def decode_image(self, cy, cb, cr, rows, cols, qy, qc, flavour=0, threads=8):
        out = np.empty((rows, cols, 3), np.uint8)
        qy, qc = np.ascontiguousarray(qy, np.int32), np.ascontiguousarray(qc, np.int32)
        cy, cb, cr = (np.ascontiguousarray(x, np.int16) for x in (cy, cb, cr))
        rc = self.lib.orc_decode_image(_p(cy), _p(cb), _p(cr), int(rows), int(cols), _p(qy), _p(qc),
                                       int(flavour), int(threads), _p(out))
        assert rc == 0
        return out

    def rle_stream(self, coef):
        """zigzag coefficient plane [nb, 64] -> the reference's (value, run) int32 stream."""
        coef = np.ascontiguousarray(coef, np.int16)
        nb = coef.shape[0]
        out = np.empty(nb * 128, np.int32)
        n = 0
        blk = np.empty(64, np.int32)
        for b in range(nb):
            blk[:] = coef[b]
            n += self.lib.orc_rle_block(_p(blk), C.c_void_p(out.ctypes.data + 4 * n))
        return out[:n].copy()

    def decode_rle(self, symbols, nb):
        symbols = np.ascontiguousarray(symbols, np.int32)
        coef = np.zeros((nb, 64), np.int16)
        done = self.lib.orc_decode_rle(_p(symbols), C.c_size_t(symbols.size), C.c_size_t(nb), _p(coef))
        return coef, int(done)

    def write_bin(self, sy, scb, scr, y_rows, y_cols):
        cap = 64 + 4 * (sy.size + scb.size + scr.size) + (1 << 16)
        out = np.empty(cap, np.uint8)
        sy, scb, scr = (np.ascontiguousarray(s, np.int32) for s in (sy, scb, scr))
        n = self.lib.orc_write_bin(_p(sy), C.c_size_t(sy.size), _p(scb), C.c_size_t(scb.size),
                                   _p(scr), C.c_size_t(scr.size), int(y_rows), int(y_cols),
                                   int(y_rows) // 2, int(y_cols) // 2, _p(out), C.c_size_t(cap))
        assert n > 0
        return out[:n].tobytes()

    def read_bin(self, data):
        buf = np.frombuffer(data, np.uint8).copy()
        dims = np.zeros(6, np.int32)
        cap = 8 * buf.size + 64
        syms = [np.empty(cap, np.int32) for _ in range(3)]
        ns = [C.c_size_t(0) for _ in range(3)]
        rc = self.lib.orc_read_bin(_p(buf), C.c_size_t(buf.size), _p(dims),
                                   _p(syms[0]), C.c_size_t(cap), C.byref(ns[0]),
                                   _p(syms[1]), C.c_size_t(cap), C.byref(ns[1]),
                                   _p(syms[2]), C.c_size_t(cap), C.byref(ns[2]))
        assert rc == 0, rc
        return dims, [s[:n.value].copy() for s, n in zip(syms, ns)]

    def mse(self, a, b):
        a, b = np.ascontiguousarray(a, np.uint8), np.ascontiguousarray(b, np.uint8)
        return self.lib.orc_mse(_p(a), _p(b), C.c_size_t(a.size))

    def psnr(self, mse):
        return self.lib.orc_psnr(float(mse))

    def dct_block(self, recentred):
        a = np.ascontiguousarray(recentred, np.int32).reshape(64)
        out = np.empty(64, np.float32)
        self.lib.orc_dct_block(_p(a), _p(out))
        return out

    def quantize_block(self, blk, q):
        out = np.ascontiguousarray(blk, np.float32).reshape(64).copy()
        q = np.ascontiguousarray(q, np.int32)
        self.lib.orc_quantize_block(_p(out), _p(q))
        return out

    def idct_block(self, coef):
        a = np.ascontiguousarray(coef, np.float32).reshape(64)
        out = np.empty(64, np.int32)
        self.lib.orc_idct_block(_p(a), _p(out))
        return out

    def zigzag_block(self, blk):
        a = np.ascontiguousarray(blk, np.float32).reshape(64)
        out = np.empty(64, np.int32)
        self.lib.orc_zigzag_block(_p(a), _p(out))
        return out

    def rle_block(self, zz):
        a = np.ascontiguousarray(zz, np.int32).reshape(64)
        out = np.empty(128, np.int32)
        n = self.lib.orc_rle_block(_p(a), _p(out))
        return out[:n].copy()


class Reference:
    """oracle/_ref: the reference's jpeg_concept/jpeg_function.cpp, compiled as is."""

    def __init__(self):
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO)
        self.lib = C.CDLL(REF_SO)
        self.lib.ref_huffman_encode.restype = C.c_long
        self.lib.ref_freq_dict_order.restype = C.c_long
        self.lib.ref_huffman_roundtrip.restype = C.c_long
        self.lib.ref_decode_rle.restype = C.c_long

    def luma_table(self, qf):
        q = np.zeros(64, np.int32)
        self.lib.ref_luma_table(int(qf), _p(q))
        return q

    def recenter_dct(self, pixels):
        a = np.ascontiguousarray(pixels, np.int32).reshape(64)
        out = np.empty(64, np.float32)
        self.lib.ref_recenter_dct(_p(a), _p(out))
        return out

    def quantize(self, blk, q):
        out = np.ascontiguousarray(blk, np.float32).reshape(64).copy()
        q = np.ascontiguousarray(q, np.int32)
        self.lib.ref_quantize(_p(out), _p(q))
        return out

    def zigzag(self, blk):
        a = np.ascontiguousarray(blk, np.float32).reshape(64)
        out = np.empty(64, np.int32)
        self.lib.ref_zigzag(_p(a), _p(out))
        return out

    def rle(self, zz):
        a = np.ascontiguousarray(zz, np.int32).reshape(64)
        out = np.empty(128, np.int32)
        n = self.lib.ref_rle(_p(a), _p(out))
        return out[:n].copy()

    def huffman_encode(self, sym):
        sym = np.ascontiguousarray(sym, np.int32)
        cap = 64 * sym.size + 64
        buf = C.create_string_buffer(cap)
        n = self.lib.ref_huffman_encode(_p(sym), C.c_long(sym.size), buf, C.c_long(cap))
        assert n >= 0
        return buf.raw[:n].decode()

    def freq_dict_order(self, sym):
        sym = np.ascontiguousarray(sym, np.int32)
        out = np.empty(2 * (sym.size + 1), np.int32)
        n = self.lib.ref_freq_dict_order(_p(sym), C.c_long(sym.size), _p(out), C.c_long(sym.size + 1))
        return out[:2 * n].reshape(n, 2).copy()

    def decode_block(self, zz, q):
        a = np.ascontiguousarray(zz, np.int32).reshape(64)
        q = np.ascontiguousarray(q, np.int32)
        out = np.empty(64, np.int32)
        self.lib.ref_decode_block(_p(a), _p(q), _p(out))
        return out


class KernelMathModel:
    def __init__(self):
        self.lib = L = _build(MODEL_SO, os.path.join(ROOT, "tests", "model"), refresh=True)
        for f in (L.jmm_check_const_div, L.jmm_check_round, L.jmm_check_quant):
            f.restype = C.c_long

    def fwd_colour(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        out = np.empty_like(img)
        self.lib.jmm_fwd_colour(_p(img), C.c_size_t(img.size // 3), _p(out))
        return out

    def fwd_colour_fast(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        out = np.empty_like(img)
        cand = C.c_long(0)
        self.lib.jmm_fwd_colour_fast(_p(img), C.c_size_t(img.size // 3), _p(out), C.byref(cand))
        return out, cand.value

    def inv_colour(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        out = np.empty_like(img)
        self.lib.jmm_inv_colour(_p(img), C.c_size_t(img.size // 3), _p(out))
        return out

    def inv_rows(self, y4, cbcr):
        """y4: [n, 4] Y bytes; cbcr: [n, 4] = cb_a, cr_a, cb_b, cr_b -> [n, 4, 3] output bytes"""
        y4 = np.ascontiguousarray(y4, np.uint8)
        cbcr = np.ascontiguousarray(cbcr, np.uint8)
        out = np.empty((y4.shape[0], 4, 3), np.uint8)
        self.lib.jmm_inv_rows(_p(y4), _p(cbcr), C.c_size_t(y4.shape[0]), _p(out))
        return out

    def fdct_quant_blocks(self, px, q):
        px = np.ascontiguousarray(px, np.int32).reshape(-1, 64)
        out = np.empty_like(px)
        q = np.ascontiguousarray(q, np.int32)
        self.lib.jmm_fdct_quant_blocks(_p(px), C.c_size_t(px.shape[0]), _p(q), _p(out))
        return out

    def dequant_idct_blocks(self, coef, q):
        coef = np.ascontiguousarray(coef, np.int32).reshape(-1, 64)
        out = np.empty_like(coef)
        q = np.ascontiguousarray(q, np.int32)
        self.lib.jmm_dequant_idct_blocks(_p(coef), C.c_size_t(coef.shape[0]), _p(q), _p(out))
        return out

    def mfma_fdct_quant_blocks(self, px, q):
        px = np.ascontiguousarray(px, np.int32).reshape(-1, 64)
        out = np.empty_like(px)
        q = np.ascontiguousarray(q, np.int32)
        self.lib.jmm_mfma_fdct_quant_blocks(_p(px), C.c_size_t(px.shape[0]), _p(q), _p(out))
        return out

    def mfma_dequant_idct_blocks(self, coef, q):
        coef = np.ascontiguousarray(coef, np.int32).reshape(-1, 64)
        out = np.empty_like(coef)
        q = np.ascontiguousarray(q, np.int32)
        self.lib.jmm_mfma_dequant_idct_blocks(_p(coef), C.c_size_t(coef.shape[0]), _p(q), _p(out))
        return out


def plane_to_blocks(plane):
    """extractBlocks layout (zero fill) as int32 [nb, 64]."""
    r, c = plane.shape
    R, Cc = (r + 7) // 8 * 8, (c + 7) // 8 * 8
    pad = np.zeros((R, Cc), np.int32)
    pad[:r, :c] = plane
    return np.ascontiguousarray(pad.reshape(R // 8, 8, Cc // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64))


def blocks_to_plane(blocks, rows, cols):
    R, Cc = (rows + 7) // 8 * 8, (cols + 7) // 8 * 8
    full = np.asarray(blocks).reshape(R // 8, Cc // 8, 8, 8).transpose(0, 2, 1, 3).reshape(R, Cc)
    return full[:rows, :cols]
