"""Synthetic input images: integer-only generators, so the same
seed gives the same bytes on the CPU and on the GPU.

  photo : smooth two-octave value field + small triangular noise (realistic
          sparsity after quantisation, ~5-8 non-zero coefficients per block)
  noise : i.i.d. uniform bytes (dense coefficients, longest run-length output,
          exercises the Cb/Cr = 256 wrap of encode_decode.cpp:67-70)
  gray  : photo luminance replicated to B=G=R (what cv::imread makes of a
          grayscale file: Y = g, Cb = Cr = 128)
"""
import torch

_M32 = 0xFFFFFFFF


def _hash(ix, iy, c, seed):
    x = (ix * 0x1E3779B1 + iy * 0x05EBCA77 + c * 0x42B2AE3D + seed * 0x27D4EB2F) & _M32
    x = x ^ (x >> 15)
    x = (x * 0x2C1B3C6D) & _M32
    x = x ^ (x >> 12)
    x = (x * 0x297A2D39) & _M32
    return x ^ (x >> 15)


def _octave(yy, xx, c, seed, cell):
    gy, fy = yy // cell, yy % cell
    gx, fx = xx // cell, xx % cell
    g = lambda a, b: _hash(a, b, c, seed) & 255
    top = (cell - fx) * g(gx, gy) + fx * g(gx + 1, gy)
    bot = (cell - fx) * g(gx, gy + 1) + fx * g(gx + 1, gy + 1)
    return ((cell - fy) * top + fy * bot) // (cell * cell)


def synth_image(rows, cols, seed=0, kind="photo", device="cpu", row_offset=0):
    """uint8 tensor [rows, cols, 3] in cv::imread channel order.  row_offset: generate rows
    [row_offset, row_offset + rows) of a taller image of the same seed (a rank's MCU-row strip)."""
    yy = torch.arange(row_offset, row_offset + rows, dtype=torch.int64, device=device).view(rows, 1)
    xx = torch.arange(cols, dtype=torch.int64, device=device).view(1, cols)
    out = torch.empty((rows, cols, 3), dtype=torch.uint8, device=device)
    for c in range(3):
        ch = 0 if kind == "gray" else c
        if kind == "noise":
            v = _hash(xx, yy, c, seed + 77) & 255
        else:
            v = (3 * _octave(yy, xx, ch, seed, 64) + _octave(yy, xx, ch, seed + 1, 16)) // 4
            h = _hash(xx, yy, ch, seed + 2)
            v = v + (h & 7) + ((h >> 3) & 7) - 7
        out[:, :, c] = v.clamp_(0, 255).to(torch.uint8)
    return out


def upscale_integer_bilinear(src, factor):
    """uint8 [n, m, 3] -> uint8 [n * factor, m * factor, 3] with an integer bilinear filter
    (weights k / factor, rounded once at the end): no library's resize filter is involved, so the
    "Lena-derived" inputs of SURVEY.md 8d (config 5 and the size ladder of image_resize.py:19-29)
    are the same bytes on every machine.  Works on the tensor's device."""
    f = int(factor)
    s = src.to(torch.int64)
    dev = src.device

    def taps(n):
        idx = torch.arange(n * f, device=dev)
        i0 = (idx // f).clamp(max=n - 1)
        return i0, (i0 + 1).clamp(max=n - 1), idx % f

    r0, r1, wr = taps(s.shape[0])
    c0, c1, wc = taps(s.shape[1])
    rows = s[r0] * (f - wr).view(-1, 1, 1) + s[r1] * wr.view(-1, 1, 1)  # scaled by f
    img = (rows[:, c0] * (f - wc).view(1, -1, 1) + rows[:, c1] * wc.view(1, -1, 1) + f * f // 2) // (f * f)
    return img.to(torch.uint8).contiguous()


def lena_upscaled(factor, device="cpu", path=None):
    """The committed 1024^2 Lena photograph (tests/golden/lena_color_1024.png, the reference's
    img/lena_color_1024.png) in cv::imread channel order, x `factor` by upscale_integer_bilinear:
    4 -> 4096^2 (config 5), 8 -> 8192^2."""
    import os
    import numpy as np
    from PIL import Image
    if path is None:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                            "lena_color_1024.png")
    rgb = np.array(Image.open(path).convert("RGB"))
    bgr = torch.from_numpy(np.ascontiguousarray(rgb[:, :, ::-1])).to(device)
    return upscale_integer_bilinear(bgr, factor) if int(factor) > 1 else bgr.contiguous()
