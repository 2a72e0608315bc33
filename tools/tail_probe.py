"""Developer helper: kernel time per tile for image heights that fill the chip's 3072 wave
slots (256 CUs x 12 waves) a whole or a fractional number of times: is there a tail?"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, jpeg_amd
from jpeg_amd.synth import synth_image
jpeg_amd.init(0)
dev = torch.device("cuda:0")
qy, qc = jpeg_amd.quality_tables(50)
for rows in (3072, 4096, 6144, 8192, 9216, 10240, 12288):
    img = synth_image(rows, 8192, 0x8192, "photo", device=dev)
    coefs = jpeg_amd.encode_image(img, qy, qc)
    rec = torch.empty_like(img)
    jpeg_amd.time_image(img, coefs, rec, qy, qc, 20)
    e, d = jpeg_amd.time_image(img, coefs, rec, qy, qc, 40)
    tiles = rows  # 16 tiles per 16 rows
    print(json.dumps({"rows": rows, "tiles": tiles, "rounds": tiles / 3072, "enc_us": e * 1e3, "dec_us": d * 1e3,
                      "enc_ns_per_tile": e * 1e6 / tiles, "dec_ns_per_tile": d * 1e6 / tiles}))
