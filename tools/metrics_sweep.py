"""the reference's size sweep in the reference's own output shape.

metric_image_file.py runs ./analysis on img/resized/lena_color_{0512..8192}.tif, parses its 13
numbers by position (:41-65), writes result/metrics_image_file.csv (:89-102) and three plots
(:124-227).  This does the same with jpegamd_analysis (the GPU leg: the CPU / OMP columns stay
nan, there is no such back-end here), for the same ladder of sizes made from the one Lena file
the repository holds (tests/golden/lena_color_1024.png, PIL resize like image_resize.py:19-29).
Same 15 CSV columns in the same order, plus this back-end's host-to-host times at the end.

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

src = Image.open(os.path.join(ROOT, "tests", "golden", "lena_color_1024.png")).convert("RGB")
exe = os.path.join(ROOT, "jpeg_amd", "jpegamd_analysis")
rows, widths = [], []
with tempfile.TemporaryDirectory() as tmp:
    os.mkdir(os.path.join(tmp, "output"))
    for size in range(512, max_size + 1, 512):
        name = f"lena_color_{size:04d}.png"
        path = os.path.join(tmp, name)
        src.resize((size, size)).save(path, compress_level=1)
        r = subprocess.run([exe, path], cwd=tmp, capture_output=True, text=True)
        if r.returncode:
            print(f"{name}: {r.stderr.strip()}", file=sys.stderr)
            continue
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        v = [float(ln.split(":")[-1].strip()) for ln in lines[:13]]
        m = re.search(r"encode ([\d.e+-]+) ms, decode ([\d.e+-]+) ms", r.stderr)
        rows.append([name, os.path.getsize(path) / 1024] + v + [float(m.group(1)), float(m.group(2))])
        widths.append(size)
        print(rows[-1], flush=True)
os.makedirs(out_dir, exist_ok=True)
with open(os.path.join(out_dir, "metrics_image_file.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Image File", "File Size (KB)", "Encoding Performance Gain", "Decoding Performance Gain",
                "Encoding Performance Gain (OMP)", "Decoding Performance Gain (OMP)",
                "Compression Ratio (CPU)", "Compression Ratio (GPU)", "Compression Ratio (CPU OMP)",
                "CPU MSE", "CPU PSNR", "GPU MSE", "GPU PSNR", "CPU MSE (OMP)", "CPU PSNR (OMP)",
                "GPU encode ms (host to host)", "GPU decode ms (host to host)"])
    w.writerows(rows)
col = lambda i: [r[i] for r in rows]
# the three figures of metric_image_file.py:124-227 (GPU series; the CPU / OMP ones have no data here)
fig, axs = plt.subplots(1, 2, figsize=(15, 5))
axs[0].plot(widths, col(15), color="blue", marker="o", label="Encoding time GPU (ms, host to host)")
axs[1].plot(widths, col(16), color="blue", marker="o", label="Decoding time GPU (ms, host to host)")
for a, t in zip(axs, ("Image Width vs Encoding Time", "Image Width vs Decoding Time")):
    a.set_xlabel("Image Width (pixels)"), a.set_ylabel("ms"), a.set_title(t), a.legend(), a.grid(True)
fig.savefig(os.path.join(out_dir, "metrics_time_vs_size.png"), dpi=60)
fig, ax = plt.subplots(figsize=(8, 5))
ax.plot(widths, col(7), color="blue", marker="o", label="Compression Ratio GPU")
ax.set_xlabel("Image Width (pixels)"), ax.set_ylabel("Compression Ratio"), ax.legend(), ax.grid(True)
fig.savefig(os.path.join(out_dir, "metrics_compression_ratio.png"), dpi=60)
fig, axs = plt.subplots(1, 2, figsize=(15, 5))
axs[0].plot(widths, col(11), color="blue", marker="o", label="MSE GPU")
axs[1].plot(widths, col(12), color="blue", marker="o", label="PSNR GPU")
for a, t in zip(axs, ("MSE", "PSNR (dB)")):
    a.set_xlabel("Image Width (pixels)"), a.set_ylabel(t), a.legend(), a.grid(True)
fig.savefig(os.path.join(out_dir, "metrics_mse_psnr.png"), dpi=60)
