"""Developer helper: in-sequence kernel times at 8192^2 for the benchmark's synthetic image and
for the Lena photograph tiled 8 x 8 (the encoder's exact-tie visits depend on the data)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
import jpeg_amd
from jpeg_amd.synth import synth_image

jpeg_amd.init(0)
dev = torch.device("cuda:0")
qy, qc = jpeg_amd.quality_tables(50)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rgb = np.array(Image.open(os.path.join(root, "tests", "golden", "lena_color_1024.png")).convert("RGB"))
lena = torch.from_numpy(np.ascontiguousarray(np.tile(rgb[:, :, ::-1], (8, 8, 1)))).to(dev)
for name, img in (("synthetic photo", synth_image(8192, 8192, 0x8192, "photo", device=dev)), ("lena x 64", lena)):
    coefs = jpeg_amd.encode_image(img, qy, qc)
    rec = torch.empty_like(img)
    e, d = jpeg_amd.time_image(img, coefs, rec, qy, qc, 50)
    e, d = jpeg_amd.time_image(img, coefs, rec, qy, qc, 50)
    print(json.dumps({"image": name, "enc_us": e * 1e3, "dec_us": d * 1e3,
                      "mpixel_s_kernels": 8192 * 8192 / ((e + d) * 1e-3) / 1e6}))
