"""Instance-mask bookkeeping of the validation path: what the reference gets from ``pycocotools.mask``
(src/utils/coco_eval_util.py:6,101-111 encodes the thresholded predictions; src/utils/coco_util.py:36-37,187 turns
polygons / tensors into masks and RLEs; COCOeval decodes both for the segm IoU).

pycocotools is third-party and absent from this image (PARITY UNPINNED for its arithmetic; oracle/pycoco_r.py restates
maskApi.c line by line, tests/test_coco_eval_cpu.py compares this module with it).  Host-side numpy: run-length
bookkeeping of at most 100 masks per image, not a device workload.

  * RLE = run lengths of the COLUMN-MAJOR mask, first run counts zeros (COCO convention); kept as uint32 arrays;
  * compressed strings (``counts`` of result files / crowd annotations): 6-bit groups, 5 data bits + continuation,
    chars offset by 48, counts from the third on stored as differences to the count two places back;
  * polygons: maskApi.c rleFrPoly -- vertices scaled by 5 and rounded, every edge walked one step per unit of its
    longer axis, a boundary point wherever x changes, mapped back to pixel columns (only exact column centres count)
    with y clipped to [0, h] and rounded up; each point toggles the fill from its (x, y) down the column-major order.
"""
import numpy as np


# ------------------------------------------------------------------------------------------------- RLE
def encode(mask):
    """mask [h, w] (bool / 0-1) -> uint32 run lengths over the column-major order, starting with zeros"""
    m = np.asarray(mask)
    flat = (m != 0).reshape(m.shape[0], m.shape[1]).ravel(order='F')
    if flat.size == 0:
        return np.zeros(1, dtype=np.uint32)
    edges = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    runs = np.diff(np.concatenate(([0], edges, [flat.size])))
    if flat[0]:
        runs = np.concatenate(([0], runs))
    return runs.astype(np.uint32)


def encode_probs(probs, threshold=0.5):
    """float mask probabilities [n, h, w] on the GPU -> list of uint32 run lengths of ``probs > threshold``, equal to
    ``encode`` of each thresholded mask: one HIP launch (hnd_mask_run_boundaries) finds the run boundaries where the
    masks live and only those few thousand keys cross PCIe; the host sorts them and takes differences."""
    import torch
    from .. import _lib, ops
    n, h, w = probs.shape
    if n == 0:
        return []
    p = probs.contiguous().float()
    lib = _lib.load()
    capacity = max(1 << 16, 64 * n * (h + w))
    while True:
        out = torch.empty(capacity, dtype=torch.int64, device=p.device)
        count = torch.empty(1, dtype=torch.int64, device=p.device)
        first = torch.empty(n, dtype=torch.uint8, device=p.device)
        _lib.check(lib.hnd_mask_run_boundaries(p.data_ptr(), n, h, w, float(threshold), out.data_ptr(), capacity,
                                               count.data_ptr(), first.data_ptr(), ops.stream_ptr()),
                   'hnd_mask_run_boundaries')
        found = int(count.item())
        if found <= capacity:
            break
        capacity = found                          # (pathological masks: every pixel a boundary)
    keys = np.sort(out[:found].cpu().numpy())
    first = first.cpu().numpy()
    splits = np.searchsorted(keys, np.arange(n + 1, dtype=np.int64) * (h * w))
    res = []
    for i in range(n):
        pos = keys[splits[i]:splits[i + 1]] - i * (h * w)
        runs = np.diff(np.concatenate(([0], pos, [h * w])))
        if first[i]:
            runs = np.concatenate(([0], runs))
        res.append(runs.astype(np.uint32))
    return res


def encode_batch(bits):
    """torch bool masks [n, h, w] on any device -> list of uint32 run lengths, equal to ``encode`` of each mask.  The
    run boundaries are found where the masks live (a 100 x 800 x 1333 stack is 107 MB; its boundaries a few thousand
    indices), so only those cross PCIe on the validation path."""
    import torch
    n, h, w = bits.shape
    if n == 0:
        return []
    flat = bits.transpose(1, 2).reshape(n, h * w)               # column-major order of every mask
    idx = (flat[:, 1:] != flat[:, :-1]).nonzero()               # sorted by (mask, position)
    rows, pos = idx[:, 0].cpu().numpy(), idx[:, 1].cpu().numpy() + 1
    first = flat[:, 0].cpu().numpy()
    splits = np.searchsorted(rows, np.arange(n + 1))
    out = []
    for i in range(n):
        runs = np.diff(np.concatenate(([0], pos[splits[i]:splits[i + 1]], [h * w])))
        if first[i]:
            runs = np.concatenate(([0], runs))
        out.append(runs.astype(np.uint32))
    return out


def decode(counts, h, w):
    """run lengths -> bool mask [h, w]"""
    counts = np.asarray(counts, dtype=np.int64)
    vals = (np.arange(len(counts)) & 1).astype(bool)
    flat = np.repeat(vals, counts)
    if flat.size < h * w:
        flat = np.concatenate((flat, np.zeros(h * w - flat.size, dtype=bool)))
    return flat[:h * w].reshape((h, w), order='F')


def area(counts):
    return int(np.asarray(counts, dtype=np.int64)[1::2].sum())


def counts_to_string(counts):
    out = bytearray()
    counts = [int(c) for c in counts]
    for i, x in enumerate(counts):
        if i > 2:
            x -= counts[i - 2]
        while True:
            c = x & 0x1f
            x >>= 5
            more = (x != -1) if (c & 0x10) else (x != 0)
            out.append((c | 0x20 if more else c) + 48)
            if not more:
                break
    return bytes(out)


def string_to_counts(s):
    if isinstance(s, str):
        s = s.encode('ascii')
    counts, p = [], 0
    while p < len(s):
        x, k = 0, 0
        while True:
            c = s[p] - 48
            x |= (c & 0x1f) << (5 * k)
            p += 1
            k += 1
            if not (c & 0x20):
                if c & 0x10:
                    x |= -1 << (5 * k)
                break
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return np.asarray(counts, dtype=np.uint32)


# ------------------------------------------------------------------------------------------------- polygons
def _edge_points(xs, ys, xe, ye):
    """dense integer points of one upsampled edge, from its start to its end (rleFrPoly's inner loops)"""
    dx, dy = abs(xe - xs), abs(ys - ye)
    flip = (dx >= dy and xs > xe) or (dx < dy and ys > ye)
    if flip:
        xs, xe, ys, ye = xe, xs, ye, ys
    n = max(dx, dy)
    t = np.arange(n + 1, dtype=np.int64)
    if flip:
        t = n - t
    if dx >= dy:
        slope = (ye - ys) / dx if dx else 0.0
        return t + xs, np.trunc(ys + slope * t + .5).astype(np.int64)
    slope = (xe - xs) / dy
    return np.trunc(xs + slope * t + .5).astype(np.int64), t + ys


def polygon_toggles(xy, h, w):
    """column-major positions at which the fill of polygon ``xy`` = [x0, y0, x1, y1, ...] toggles"""
    k = len(xy) // 2
    scale = 5.0
    px = [int(scale * float(xy[2 * j]) + .5) for j in range(k)]
    py = [int(scale * float(xy[2 * j + 1]) + .5) for j in range(k)]
    px.append(px[0])
    py.append(py[0])
    us, vs = zip(*[_edge_points(px[j], py[j], px[j + 1], py[j + 1]) for j in range(k)])
    u, v = np.concatenate(us), np.concatenate(vs)
    j = np.flatnonzero(u[1:] != u[:-1]) + 1
    xd = np.where(u[j] < u[j - 1], u[j], u[j] - 1).astype(np.float64)
    xd = (xd + .5) / scale - .5
    ok = (np.floor(xd) == xd) & (xd >= 0) & (xd <= w - 1)
    yd = np.minimum(v[j], v[j - 1]).astype(np.float64)
    yd = np.ceil(np.clip((yd + .5) / scale - .5, 0, h))
    return (xd[ok].astype(np.int64) * int(h) + yd[ok].astype(np.int64))


def polygons_to_mask(polygons, h, w):
    """COCO polygon list of one object (its parts are OR-ed: annToRLE's merge) -> bool mask [h, w]"""
    out = np.zeros((h, w), dtype=bool)
    for poly in polygons:
        if len(poly) < 6:           # pycocotools rasterises anything; a segment / point encloses no pixel centre
            continue
        flips = np.zeros(h * w + 1, dtype=np.int64)
        np.add.at(flips, polygon_toggles(poly, h, w), 1)
        out |= (np.cumsum(flips)[:h * w] & 1).astype(bool).reshape((h, w), order='F')
    return out


def bbox_to_mask(bbox, h, w):
    """frBbox: the rectangle polygon of an xywh box"""
    x, y, bw, bh = (float(v) for v in bbox)
    return polygons_to_mask([[x, y, x, y + bh, x + bw, y + bh, x + bw, y]], h, w)


def segmentation_to_mask(segm, h, w):
    """COCO.annToMask: polygons, uncompressed RLE ({'counts': [...]}) or compressed RLE ({'counts': str | bytes})"""
    if isinstance(segm, list):
        return polygons_to_mask(segm, h, w)
    counts = segm['counts']
    hh, ww = segm.get('size', (h, w))
    if isinstance(counts, (str, bytes)):
        counts = string_to_counts(counts)
    return decode(counts, int(hh), int(ww))
