"""Developer helper: compare an encoder variant (JPEGAMD_TRANSFORM) with the default VALU encoder."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jpeg_amd
from jpeg_amd.synth import synth_image

jpeg_amd.init(0)
dev = torch.device("cuda:0")
variant, rows, cols = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
for qf, kind in ((95, "noise"), (50, "photo")):
    qy, qc = jpeg_amd.quality_tables(qf)
    bgr = synth_image(rows, cols, 5, kind, device=dev)
    os.environ["JPEGAMD_TRANSFORM"] = "valu"
    ref = [c.cpu().numpy().reshape(-1, 64) for c in jpeg_amd.encode_image(bgr, qy, qc)]
    os.environ["JPEGAMD_TRANSFORM"] = variant
    got = [c.cpu().numpy().reshape(-1, 64) for c in jpeg_amd.encode_image(bgr, qy, qc)]
    for name, r, g in zip("Y Cb Cr".split(), ref, got):
        key = {tuple(b[:12]): i for i, b in enumerate(r)}
        perm = [key.get(tuple(b[:12]), -1) for b in g]
        wrong = [(i, p) for i, p in enumerate(perm) if p != i]
        d = np.abs(r.astype(int) - g.astype(int))
        print(kind, qf, name, "blocks", len(perm), "misplaced", len(wrong), "first", wrong[:12], "max diff", d.max(), "n diff", (d != 0).sum(), "of", d.size)
