"""the reference's size sweep in the reference's own output shape.

metric_image_file.py runs ./analysis on img/resized/lena_color_{0512..8192}.tif, parses its 13
numbers by position (:41-65), writes result/metrics_image_file.csv (:89-102) and three plots
(:124-227).  This does the same with jpegamd_analysis (the GPU leg) for the same ladder of sizes made from the one
Lena file the repository holds (tests/golden/lena_color_1024.png, PIL resize like image_resize.py:19-29).
The library has no CPU back-end, so the CPU series is the ORACLE (oracle/liboracle.so: the single-thread
restatement of encodeChannel / decodeChannel + the reference's entropy coder and container, kind "port"
like bench.py's cpu_baseline) timed HERE, in this tool, over the reference's timed region (mainEncode /
mainDecode: three planes from subsampled planes to symbols and code, and back; colour conversion and file
I/O outside, encode_decode.cpp:1310-1375); its times go to jpegamd_analysis as --cpu-encode-ms /
--cpu-decode-ms, its ratio, MSE and PSNR fill the CPU columns.  The OMP columns stay nan.
Same 15 CSV columns in the same order, plus both back-ends' times at the end.

    python tools/metrics_sweep.py [out_dir] [max_size]
"""
import csv
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out")
max_size = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
from PIL import Image
import matplotlib
matplotlib.use("Agg")
import matplotlib.pyplot as plt

import time
import numpy as np
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import oracle_lib  # the checker, used here as the timed CPU series (never by the library)
from jpeg_amd.api import analysis_tables
orc = oracle_lib.Oracle()
QY, QC = analysis_tables(50)  # analysis.cpp:43-90: the fixed tables, what jpegamd_analysis uses too


def cpu_leg(rgb):
    """the oracle over the reference's timed region; returns (encode ms, decode ms, ratio, MSE, PSNR)"""
    bgr = np.ascontiguousarray(rgb[:, :, ::-1])
    rows, cols = bgr.shape[:2]
    planes = orc.chroma_subsample(orc.rgb2ycbcr(bgr))  # outside the reference's timers
    qs = (QY, QC, QC)
    t0 = time.perf_counter()
    syms = [orc.encode_plane(p, q, 0, True)[1] for p, q in zip(planes, qs)]
    data = orc.write_bin(*syms, rows, cols)
    enc_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    dims, back = orc.read_bin(data)
    rec = []
    for s, p, q in zip(back, planes, qs):
        r, c = p.shape
        rec.append(orc.decode_plane(orc.decode_rle(s, (r // 8) * (c // 8))[0], r, c, q))
    dec_ms = (time.perf_counter() - t0) * 1e3
    up = lambda c: np.repeat(np.repeat(c, 2, axis=0), 2, axis=1)
    out = orc.ycbcr2rgb(np.ascontiguousarray(np.stack([rec[0], up(rec[1]), up(rec[2])], -1)))
    mse = orc.mse(bgr, out)
    return enc_ms, dec_ms, rows * cols * 3 / len(data), mse, orc.psnr(mse)


src = Image.open(os.path.join(ROOT, "tests", "golden", "lena_color_1024.png")).convert("RGB")
exe = os.path.join(ROOT, "jpeg_amd", "jpegamd_analysis")
rows, widths = [], []
with tempfile.TemporaryDirectory() as tmp:
    os.mkdir(os.path.join(tmp, "output"))
    for size in range(512, max_size + 1, 512):
        name = f"lena_color_{size:04d}.png"
        path = os.path.join(tmp, name)
        img = src.resize((size, size))
        img.save(path, compress_level=1)
        c_enc, c_dec, c_ratio, c_mse, c_psnr = cpu_leg(np.array(img))
        r = subprocess.run([exe, path, "--cpu-encode-ms", repr(c_enc), "--cpu-decode-ms", repr(c_dec)], cwd=tmp,
                           capture_output=True, text=True)
        if r.returncode:
            print(f"{name}: {r.stderr.strip()}", file=sys.stderr)
            continue
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        v = [float(ln.split(":")[-1].strip()) for ln in lines[:13]]
        m = re.search(r"encode ([\d.e+-]+) ms, decode ([\d.e+-]+) ms", r.stderr)
        v[4], v[7], v[8] = c_ratio, c_mse, c_psnr  # the CPU columns: the oracle's file and image
        rows.append([name, os.path.getsize(path) / 1024] + v + [float(m.group(1)), float(m.group(2)), c_enc, c_dec])
        widths.append(size)
        print(rows[-1], flush=True)
os.makedirs(out_dir, exist_ok=True)
with open(os.path.join(out_dir, "metrics_image_file.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Image File", "File Size (KB)", "Encoding Performance Gain", "Decoding Performance Gain",
                "Encoding Performance Gain (OMP)", "Decoding Performance Gain (OMP)",
                "Compression Ratio (CPU)", "Compression Ratio (GPU)", "Compression Ratio (CPU OMP)",
                "CPU MSE", "CPU PSNR", "GPU MSE", "GPU PSNR", "CPU MSE (OMP)", "CPU PSNR (OMP)",
                "GPU encode ms (host to host)", "GPU decode ms (host to host)",
                "CPU encode ms (oracle port, 1 thread)", "CPU decode ms (oracle port, 1 thread)"])
    w.writerows(rows)
col = lambda i: [r[i] for r in rows]
# the three figures of metric_image_file.py:124-227 (CPU = the oracle port, GPU = this library; no OMP series)
fig, axs = plt.subplots(1, 3, figsize=(20, 5))
axs[0].plot(widths, col(17), color="red", marker="o", label="Encoding time CPU (oracle port, 1 thread)")
axs[0].plot(widths, col(15), color="blue", marker="o", label="Encoding time GPU (host to host)")
axs[1].plot(widths, col(18), color="red", marker="o", label="Decoding time CPU (oracle port, 1 thread)")
axs[1].plot(widths, col(16), color="blue", marker="o", label="Decoding time GPU (host to host)")
axs[2].plot(widths, col(2), color="blue", marker="o", label="Encoding Performance Gain")
axs[2].plot(widths, col(3), color="green", marker="o", label="Decoding Performance Gain")
for a, t in zip(axs, ("Image Width vs Encoding Time (ms)", "Image Width vs Decoding Time (ms)", "Image Width vs Performance Gain (CPU time / GPU time)")):
    a.set_xlabel("Image Width (pixels)"), a.set_title(t), a.legend(), a.grid(True)
axs[0].set_yscale("log"), axs[1].set_yscale("log")
fig.savefig(os.path.join(out_dir, "metrics_time_vs_size.png"), dpi=60)
fig, ax = plt.subplots(figsize=(8, 5))
ax.plot(widths, col(6), color="red", marker="o", label="Compression Ratio CPU")
ax.plot(widths, col(7), color="blue", marker="x", label="Compression Ratio GPU")
ax.set_xlabel("Image Width (pixels)"), ax.set_ylabel("Compression Ratio"), ax.legend(), ax.grid(True)
fig.savefig(os.path.join(out_dir, "metrics_compression_ratio.png"), dpi=60)
fig, axs = plt.subplots(1, 2, figsize=(15, 5))
axs[0].plot(widths, col(9), color="red", marker="o", label="MSE CPU")
axs[0].plot(widths, col(11), color="blue", marker="x", label="MSE GPU")
axs[1].plot(widths, col(10), color="red", marker="o", label="PSNR CPU")
axs[1].plot(widths, col(12), color="blue", marker="x", label="PSNR GPU")
for a, t in zip(axs, ("MSE", "PSNR (dB)")):
    a.set_xlabel("Image Width (pixels)"), a.set_ylabel(t), a.legend(), a.grid(True)
fig.savefig(os.path.join(out_dir, "metrics_mse_psnr.png"), dpi=60)
