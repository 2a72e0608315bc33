"""Developer helper for rocprofv3 passes over the NON-headline kernels: run-length stage (rle_*, rld_*),
entropy encode (sym_*) and entropy decode (hd_*) of an 8192^2 image, a few launches each, nothing else."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, jpeg_amd
from jpeg_amd.synth import synth_image
size = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
jpeg_amd.init(0)
dev = torch.device("cuda:0")
qy, qc = jpeg_amd.quality_tables(50)
bgr = synth_image(size, size, 0x8192, "photo", device=dev)
for _ in range(3):
    coefs, streams = jpeg_amd.encode_image_rle(bgr, qy, qc)
torch.cuda.synchronize()
pairs = [p for p, _ in streams]
for _ in range(2):
    data = jpeg_amd.bin_encode_dev(*[p.reshape(-1) for p in pairs], size, size)
for _ in range(2):
    dims, syms = jpeg_amd.bin_decode_dev(data)
for s, c in zip(syms, coefs):
    back = os.environ.get("JPEGAMD_TIMING_ONLY") or jpeg_amd.rle_decode_plane(s[:(s.numel() // 2) * 2].reshape(-1, 2), c.shape[0])
torch.cuda.synchronize()
assert os.environ.get("JPEGAMD_TIMING_ONLY") or all(torch.equal(s[:p.numel()], p.reshape(-1)) for s, p in zip(syms, pairs))
print("bin bytes", len(data), "pairs", [int(p.shape[0]) for p in pairs])
