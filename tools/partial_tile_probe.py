"""Developer helper: what the half-empty last tile column of a 3840-wide image costs -- the same number of pixels as
(8192 x 3840) and as (7680 x 4096), kernel times from jpegamd_time_image_dev (HIP events, encode/decode in sequence)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, jpeg_amd
from jpeg_amd.synth import synth_image
jpeg_amd.init(0)
dev = torch.device("cuda:0")
qy, qc = jpeg_amd.quality_tables(50)
shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]] or [(8192, 3840), (7680, 4096), (17280, 3840), (16192, 4096), (8640, 1920), (8096, 2048), (11520, 1280), (9600, 1536)]
for rows, cols in shapes:
    bgr = synth_image(rows, cols, 0x8192, "photo", device=dev)
    coefs = jpeg_amd.encode_image(bgr, qy, qc)
    rec = torch.empty_like(bgr)
    jpeg_amd.time_image(bgr, coefs, rec, qy, qc, 5)
    res = [jpeg_amd.time_image(bgr, coefs, rec, qy, qc, 30) for _ in range(3)]
    enc, dec = min(r[0] for r in res) * 1e3, min(r[1] for r in res) * 1e3
    mp = rows * cols / 1e6
    print(f"{rows:6d} x {cols:4d}  {mp:7.2f} Mpixel  encode {enc:7.1f} us ({enc / mp:.3f} us/Mpx)  decode {dec:7.1f} us ({dec / mp:.3f} us/Mpx)")
