"""the reference's size sweep (image_resize.py: 512^2 ... 8192^2 in steps of 512)
with this back-end.  Per size: kernel times in sequence and the host-pointer calls image ->
.bin -> image (second call of each size; wall clock).  Synthetic photo-like images, QF=50.
python tools/size_sweep.py > profiles/<name>.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, jpeg_amd
from jpeg_amd.synth import synth_image

jpeg_amd.init(0)
dev = torch.device("cuda:0")
qy, qc = jpeg_amd.quality_tables(50)
print("# size sweep, synthetic photo-like images, QF=50, MI355X; kernels timed in sequence (HIP time stamps),")
print("# image -> .bin and .bin -> image from pageable host memory (wall clock, second call)")
print("# size   enc_us  dec_us  kernels_Mpixel/s   to_bin_ms  from_bin_ms  end_to_end_Mpixel/s  bin_bytes")
for size in range(512, 8192 + 1, 512):
    img = synth_image(size, size, 0x8192, "photo", device=dev)
    coefs = jpeg_amd.encode_image(img, qy, qc)
    rec = torch.empty_like(img)
    jpeg_amd.time_image(img, coefs, rec, qy, qc, 10)
    e, d = jpeg_amd.time_image(img, coefs, rec, qy, qc, 30)
    host = img.cpu().numpy()
    for _ in range(2):
        t0 = time.perf_counter(); data = jpeg_amd.compress_image(host, qy, qc); t1 = time.perf_counter()
        back = jpeg_amd.decompress_image(data, qy, qc); t2 = time.perf_counter()
    npx = size * size
    print(f"{size:6d}  {e * 1e3:6.1f}  {d * 1e3:6.1f}  {npx / ((e + d) * 1e-3) / 1e6:12.0f}      "
          f"{1e3 * (t1 - t0):8.2f}  {1e3 * (t2 - t1):8.2f}  {npx / (t2 - t0) / 1e6:14.0f}      {len(data):10d}")
