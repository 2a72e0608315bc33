"""MCU-row strip sharding of one image across ranks.

Every 16x16 MCU (4 Y + Cb + Cr blocks) is independent up to and including the
per-block run-length pairs (encode_decode.cpp:350-373: runs never cross
blocks, no DC prediction), so a rank encodes rows [r0, r1) of the image on its
own and the concatenation of the ranks' coefficient planes in rank order IS the
whole image's planes in block raster order (encode_decode.cpp:643-645).  The
only exchange is the gather of the encoded strips; there is no data-path
collective.

The functions here are host logic over torch tensors and torch.distributed
(backend nccl = RCCL on the GPUs, gloo in the CPU tests); the compute of a
strip is jpeg_amd.encode_image / decode_image on that rank's GPU.
"""
import torch

MCU = 16


def strip_rows(rows, world, rank):
    """Pixel rows [r0, r1) owned by `rank`: contiguous MCU rows, sizes differ by at
    most one MCU row; a rank may own nothing when world > rows/16."""
    if rows % MCU:
        raise ValueError("rows must be a multiple of 16 (ensureMultipleOf16, encode_decode.cpp:29-37)")
    if not 0 <= rank < world:
        raise ValueError("rank outside the world")
    n = rows // MCU
    return MCU * (rank * n // world), MCU * ((rank + 1) * n // world)


def strip_block_counts(rows, cols, world, rank):
    """(Y blocks, chroma blocks per plane) of the rank's strip."""
    r0, r1 = strip_rows(rows, world, rank)
    h = r1 - r0
    return (h // 8) * (cols // 8), (h // 16) * (cols // 16)


def split_planes(coef_y, coef_cb, coef_cr, rows, cols, world, rank):
    """The rank's strip of whole-image coefficient planes (views, no copy):
    what the root scatters before the inverse transform."""
    offs_y = offs_c = 0
    for g in range(rank):
        ny, nc = strip_block_counts(rows, cols, world, g)
        offs_y += ny
        offs_c += nc
    ny, nc = strip_block_counts(rows, cols, world, rank)
    return coef_y[offs_y:offs_y + ny], coef_cb[offs_c:offs_c + nc], coef_cr[offs_c:offs_c + nc]


def gather_planes(strip_y, strip_cb, strip_cr, rows, cols, group=None):
    """All ranks' strips -> whole-image planes on every rank, in rank order.

    Strips differ in size by at most one MCU row, so one all_gather on buffers
    padded to the largest strip moves everything (over xGMI each rank's strip is
    one direct peer write per destination); the counts are known from the
    partition and need no exchange."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    counts = [strip_block_counts(rows, cols, world, g) for g in range(world)]
    max_y = max(c[0] for c in counts)
    max_c = max(c[1] for c in counts)
    rank = dist.get_rank(group)
    if (strip_y.shape[0], strip_cb.shape[0], strip_cr.shape[0]) != (counts[rank][0], counts[rank][1], counts[rank][1]):
        raise ValueError("strip planes do not match this rank's share of the partition")
    # one buffer per rank: [Y | Cb | Cr], padded
    mine = torch.zeros((max_y + 2 * max_c, 64), dtype=strip_y.dtype, device=strip_y.device)
    mine[:strip_y.shape[0]] = strip_y
    mine[max_y:max_y + strip_cb.shape[0]] = strip_cb
    mine[max_y + max_c:max_y + max_c + strip_cr.shape[0]] = strip_cr
    # moved as raw bytes: gloo has no int16 collectives, and the payload is opaque anyway
    raw = [torch.empty_like(mine).view(torch.uint8) for _ in range(world)]
    dist.all_gather(raw, mine.view(torch.uint8), group=group)
    bufs = [r.view(mine.dtype) for r in raw]
    ys = [b[:c[0]] for b, c in zip(bufs, counts)]
    cbs = [b[max_y:max_y + c[1]] for b, c in zip(bufs, counts)]
    crs = [b[max_y + max_c:max_y + max_c + c[1]] for b, c in zip(bufs, counts)]
    return torch.cat(ys), torch.cat(cbs), torch.cat(crs)


def encode_image_sharded(bgr_strip, rows, cols, q_luma, q_chroma, group=None):
    """Each rank passes ITS strip of the image (rows strip_rows(...) of it, on its
    GPU); returns the whole image's coefficient planes on every rank."""
    import torch.distributed as dist
    from . import api
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    r0, r1 = strip_rows(rows, world, rank)
    if tuple(bgr_strip.shape) != (r1 - r0, cols, 3):
        raise ValueError("bgr_strip is not this rank's strip")
    if r1 > r0:
        sy, scb, scr = api.encode_image(bgr_strip, q_luma, q_chroma)
    else:
        sy, scb, scr = (torch.empty((0, 64), dtype=torch.int16, device=bgr_strip.device) for _ in range(3))
    return gather_planes(sy, scb, scr, rows, cols, group)


def decode_image_sharded(coef_y, coef_cb, coef_cr, rows, cols, q_luma, q_chroma, group=None):
    """Every rank holds the whole planes (after the root's entropy decode and a
    broadcast); each decodes its own strip and returns it (rows strip_rows(...))."""
    import torch.distributed as dist
    from . import api
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    r0, r1 = strip_rows(rows, world, rank)
    sy, scb, scr = split_planes(coef_y, coef_cb, coef_cr, rows, cols, world, rank)
    if r1 == r0:
        return torch.empty((0, cols, 3), dtype=torch.uint8, device=coef_y.device)
    return api.decode_image(sy.contiguous(), scb.contiguous(), scr.contiguous(), r1 - r0, cols, q_luma, q_chroma)
