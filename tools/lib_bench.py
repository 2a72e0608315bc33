"""Developer helper: encoder/decoder kernel times for an alternative build of the library
(ablation builds under build/): python tools/lib_bench.py <path-to-.so> [size]."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jpeg_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import jpeg_amd
from jpeg_amd.synth import synth_image
size = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
jpeg_amd.init(0)
dev = torch.device("cuda:0")
qy, qc = jpeg_amd.quality_tables(50)
bgr = synth_image(size, size, 0x8192, "photo", device=dev)
coefs = jpeg_amd.encode_image(bgr, qy, qc)
rec = torch.empty_like(bgr)
jpeg_amd.time_image(bgr, coefs, rec, qy, qc, 5)
res = [jpeg_amd.time_image(bgr, coefs, rec, qy, qc, 30) for _ in range(3)]
print(json.dumps({"lib": sys.argv[1], "transform": os.environ.get("JPEGAMD_TRANSFORM", "valu"),
                  "enc_us": min(r[0] for r in res) * 1e3, "dec_us": min(r[1] for r in res) * 1e3}))
