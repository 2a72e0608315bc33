"""Developer probe: where the fused decoder differs from the oracle at one quality factor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
if len(sys.argv) > 2:
    from jpeg_amd import _lib
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
import jpeg_amd
from jpeg_amd.synth import synth_image
from oracle_lib import Oracle
qf = int(sys.argv[1]) if len(sys.argv) > 1 else 17
orc = Oracle()
jpeg_amd.init(0)
qy, qc = jpeg_amd.quality_tables(qf)
bgr = synth_image(128, 256, 1000 + qf, "photo").numpy()
dev = torch.device("cuda:0")
coefs = jpeg_amd.encode_image(torch.from_numpy(bgr).to(dev), qy, qc)
host = [c.cpu().numpy() for c in coefs]
rec = jpeg_amd.decode_image(*coefs, 128, 256, qy, qc).cpu().numpy()
want = orc.decode_image(*host, 128, 256, qy, qc)
d = rec.astype(int) - want.astype(int)
print("fused decode vs oracle: differing samples per channel", [(int((d[..., c] != 0).sum()), int(np.abs(d[..., c]).max())) for c in range(3)])
for name, c, r, cc, q in (("Y", coefs[0], 128, 256, qy), ("Cb", coefs[1], 64, 128, qc), ("Cr", coefs[2], 64, 128, qc)):
    got = jpeg_amd.decode_plane(c, r, cc, q).cpu().numpy()
    w = orc.decode_plane(c.cpu().numpy(), r, cc, q)
    dd = got.astype(int) - w.astype(int)
    print(name, "plane kernel vs oracle:", int((dd != 0).sum()), "of", dd.size, "max", int(np.abs(dd).max()))
