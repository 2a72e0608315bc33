"""Register/LDS use and instruction mix of the image kernels from `make -C jpeg_amd/csrc asm`."""
import re
import sys
from collections import Counter

s = "".join(open(f).read() for f in (sys.argv[1:] or ["/tmp/jpeg_kernels.s", "/tmp/encode_variants.s"]))
for name in ("encode_image_xdl_kernel", "encode_image_mfma_kernel", "encode_image_kernel", "decode_image_kernel"):
    sym = re.search(r"^(_ZN\S*\d+" + name + r"E\S*):", s, re.M).group(1)
    blk = re.search(r"\.amdhsa_kernel " + re.escape(sym) + r".*?\.end_amdhsa_kernel", s, re.S).group(0)
    meta = {k: re.search(k + r"\s+(\S+)", blk).group(1) for k in
            ("next_free_vgpr", "accum_offset", "group_segment_fixed_size", "private_segment_fixed_size")}
    body = s[s.index(sym + ":"):]
    body = body[:body.index(".amdhsa_kernel")]
    c = Counter(l.split()[0] for l in body.splitlines()
                if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";")))
    print(name, meta)
    print("  ", {k: v for k, v in sorted(c.items()) if k.startswith(("v_mfma", "v_accvgpr", "ds_", "global_", "s_nop", "scratch", "s_waitcnt"))})
    print("   static VALU:", sum(v for k, v in c.items() if k.startswith("v_")), " total:", sum(c.values()))
