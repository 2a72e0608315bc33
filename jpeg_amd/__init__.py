"""jpeg_amd -- MI355X (gfx950) implementation of the Fulim13/jpeg block-transform
hot path, behind the C ABI of include/jpeg_amd.h.  Python here is plumbing for
tests and bench.py: device memory (torch), streams, torch.distributed."""
from ._lib import JpegAmdError, load, LIB_PATH  # noqa: F401
from .api import (  # noqa: F401
    quality_tables, analysis_tables, init,
    encode_image, decode_image, colour_planes, encode_images, decode_images, encode_plane, decode_plane,
    fdct_quant_blocks, dequant_idct_blocks, time_image, ZIGZAG,
    rle_encode_plane, rle_encode_planes, rle_decode_plane, encode_image_rle,
    bin_encode, bin_encode_dev, bin_decode, bin_decode_dev, compress_image, decompress_image, Context,
)
