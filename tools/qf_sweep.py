"""quality sweep on a 4096^2 image derived from the Lena photograph (the
committed 1024^2 golden input, x4 with an integer bilinear filter so that no library's resize
filter is involved).  Per QF: kernel times in sequence, non-zero coefficients and run-length
pairs per block, .bin size.  python tools/qf_sweep.py > profiles/<name>.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
import jpeg_amd

jpeg_amd.init(0)
dev = torch.device("cuda:0")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from jpeg_amd.synth import lena_upscaled
F = 4
img = lena_upscaled(F, device=dev)
n = img.shape[0] // F
R = C = n * F
print(f"# QF sweep, {R}x{C} image (Lena 1024^2 x{F}, integer bilinear), MI355X; kernel times in sequence")
print("# QF  enc_us  dec_us  Mpixel/s(kernels)  nnz/block Y Cb Cr      pairs/block Y Cb Cr    bin_bytes  ratio  PSNR_dB")
for qf in (1, 5, 10, 25, 50, 75, 90, 95, 100):
    qy, qc = jpeg_amd.quality_tables(qf)
    coefs = jpeg_amd.encode_image(img, qy, qc)
    rec = torch.empty_like(img)
    jpeg_amd.time_image(img, coefs, rec, qy, qc, 10)
    e, d = jpeg_amd.time_image(img, coefs, rec, qy, qc, 30)
    nnz = [float((c != 0).sum().item()) / c.shape[0] for c in coefs]
    pairs = [jpeg_amd.rle_encode_plane(c)[0] for c in coefs]
    ppb = [p.shape[0] / c.shape[0] for p, c in zip(pairs, coefs)]
    data = jpeg_amd.bin_encode_dev(*[p.reshape(-1) for p in pairs], R, C)
    mse = float(((img.float() - rec.float()) ** 2).mean().item())
    psnr = 10 * np.log10(255.0 ** 2 / mse)
    print(f"{qf:4d}  {e * 1e3:6.1f}  {d * 1e3:6.1f}  {R * C / ((e + d) * 1e-3) / 1e6:10.0f}        "
          f"{nnz[0]:5.2f} {nnz[1]:5.2f} {nnz[2]:5.2f}      {ppb[0]:5.2f} {ppb[1]:5.2f} {ppb[2]:5.2f}   "
          f"{len(data):10d}  {R * C * 3 / len(data):5.1f}  {psnr:6.2f}")
