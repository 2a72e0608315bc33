"""Developer helper for rocprofv3 --pmc passes: a handful of launches, nothing else."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, jpeg_amd
from jpeg_amd.synth import synth_image
size = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
jpeg_amd.init(0)
dev = torch.device("cuda:0")
qy, qc = jpeg_amd.quality_tables(50)
bgr = synth_image(size, size, 0x8192, "photo", device=dev)
coefs = jpeg_amd.encode_image(bgr, qy, qc)
rec = jpeg_amd.decode_image(*coefs, size, size, qy, qc)
for _ in range(3):
    jpeg_amd.encode_image(bgr, qy, qc, out=coefs)
    jpeg_amd.decode_image(*coefs, size, size, qy, qc, out=rec)
torch.cuda.synchronize()
