"""Minimal PNG codec for the TUM RGB-D harness (tools/run_tum.py): the sequences the reference is run on
(Examples/RGB-D/rgbd_tum.cc:87-88, cv::imread(..., CV_LOAD_IMAGE_UNCHANGED)) are 8-bit RGB colour images and 16-bit
grayscale depth maps.  No image library ships with this image, so the decoder is written on zlib + numpy:
non-interlaced PNG, colour types 0 (gray), 2 (RGB), 4 (gray + alpha), 6 (RGBA), bit depths 8 and 16, all five
scan-line filters.  Palette and interlaced files raise ValueError.  write_png() exists for the tests (it can emit any
fixed or per-row filter so that the decoder's un-filtering is exercised)."""
import struct
import zlib

import numpy as np

_SIG = b"\x89PNG\r\n\x1a\n"
_CHANNELS = {0: 1, 2: 3, 4: 2, 6: 4}


def _unfilter(raw, h, stride, bpp):
    """raw: h * (1 + stride) bytes.  None / Up / Sub are vectorised per row (Sub = a prefix sum per byte lane, modulo 256); Average and Paeth walk the row"""
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.uint8)
    data = np.frombuffer(raw, np.uint8).reshape(h, 1 + stride)
    for y in range(h):
        ft = int(data[y, 0])
        line = data[y, 1:]
        if ft == 0:
            cur = line.copy()
        elif ft == 2:
            cur = line + prev                                   # uint8 arithmetic wraps: modulo 256, as specified
        elif ft == 1:
            # cur[i] = line[i] + cur[i - bpp]: a prefix sum per channel, modulo 256
            cur = np.cumsum(line.reshape(-1, bpp).astype(np.uint32), axis=0).astype(np.uint8).reshape(-1)
        elif ft in (3, 4):
            # Average / Paeth: each byte depends on the reconstructed byte bpp to its left -- a plain byte loop (faster than numpy on bpp-wide vectors)
            ln, up, cb = line.tobytes(), prev.tobytes(), bytearray(stride)
            if ft == 3:
                for i in range(stride):
                    left = cb[i - bpp] if i >= bpp else 0
                    cb[i] = (ln[i] + ((left + up[i]) >> 1)) & 255
            else:
                for i in range(stride):
                    if i >= bpp:
                        a, c = cb[i - bpp], up[i - bpp]
                    else:
                        a = c = 0
                    b = up[i]
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                    cb[i] = (ln[i] + pred) & 255
            cur = np.frombuffer(bytes(cb), np.uint8)
        else:
            raise ValueError("PNG: unknown filter type %d" % ft)
        out[y] = cur
        prev = cur
    return out


def read_png(path):
    """-> numpy array (H, W) or (H, W, C), uint8 or uint16 (native byte order), exactly what cv::imread(path, IMREAD_UNCHANGED) holds except that
    colour channels stay in file order (R, G, B[, A]); OpenCV stores B, G, R"""
    with open(path, "rb") as fh:
        buf = fh.read()
    if buf[:8] != _SIG:
        raise ValueError("%s: not a PNG file" % path)
    pos, idat, hdr = 8, [], None
    while pos + 8 <= len(buf):
        n, typ = struct.unpack(">I4s", buf[pos:pos + 8])
        body = buf[pos + 8:pos + 8 + n]
        pos += 12 + n
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"IEND":
            break
    if hdr is None:
        raise ValueError("%s: no IHDR chunk" % path)
    w, h, depth, ctype, comp, flt, interlace = hdr
    if ctype not in _CHANNELS or depth not in (8, 16) or comp != 0 or flt != 0:
        raise ValueError("%s: unsupported PNG (colour type %d, bit depth %d)" % (path, ctype, depth))
    if interlace:
        raise ValueError("%s: interlaced PNG is not supported" % path)
    ch = _CHANNELS[ctype]
    bpp = ch * depth // 8
    stride = w * bpp
    raw = zlib.decompress(b"".join(idat))
    if len(raw) != h * (1 + stride):
        raise ValueError("%s: %d bytes of image data, expected %d" % (path, len(raw), h * (1 + stride)))
    px = _unfilter(raw, h, stride, bpp)
    if depth == 16:
        px = px.reshape(h, w * ch, 2)
        img = (px[:, :, 0].astype(np.uint16) << 8) | px[:, :, 1].astype(np.uint16)    # big-endian samples
    else:
        img = px
    return img.reshape(h, w) if ch == 1 else img.reshape(h, w, ch)


def _filter_row(ft, line, prev, bpp):
    ln = line.astype(np.int32)
    left = np.concatenate([np.zeros(bpp, np.int32), ln[:-bpp]])
    up = prev.astype(np.int32)
    upleft = np.concatenate([np.zeros(bpp, np.int32), up[:-bpp]])
    if ft == 0:
        pred = 0
    elif ft == 1:
        pred = left
    elif ft == 2:
        pred = up
    elif ft == 3:
        pred = (left + up) >> 1
    else:
        p = left + up - upleft
        pa, pb, pc = np.abs(p - left), np.abs(p - up), np.abs(p - upleft)
        pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, upleft))
    return ((ln - pred) & 255).astype(np.uint8)


def write_png(path, img, filter_type=0, level=6):
    """img: (H, W) or (H, W, C) uint8 / uint16, C in 2..4.  filter_type: 0..4, or -1 = a different filter per row (y % 5)"""
    img = np.asarray(img)
    if img.dtype not in (np.uint8, np.uint16):
        raise ValueError("uint8 or uint16 expected")
    h, w = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[ch]
    depth = 8 * img.dtype.itemsize
    rows = img.reshape(h, w * ch)
    if depth == 16:
        rows = np.stack([(rows >> 8).astype(np.uint8), (rows & 255).astype(np.uint8)], axis=-1).reshape(h, w * ch * 2)
    bpp = ch * depth // 8
    prev = np.zeros(rows.shape[1], np.uint8)
    parts = []
    for y in range(h):
        ft = filter_type if filter_type >= 0 else y % 5
        parts.append(bytes([ft]) + _filter_row(ft, rows[y], prev, bpp).tobytes())
        prev = rows[y]

    def chunk(typ, body):
        return struct.pack(">I", len(body)) + typ + body + struct.pack(">I", zlib.crc32(typ + body) & 0xFFFFFFFF)
    with open(path, "wb") as fh:
        fh.write(_SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) +
                 chunk(b"IDAT", zlib.compress(b"".join(parts), level)) + chunk(b"IEND", b""))


def read_associations(path):
    """TUM association file (Examples/RGB-D/associations/fr1_desk.txt; LoadImages in Examples/RGB-D/rgbd_tum.cc:150-177): one line per frame,
    `t_rgb rgb/<file>.png t_depth depth/<file>.png`.  -> list of (timestamp, rgb path, depth path); blank lines are skipped as the reference does"""
    out = []
    with open(path) as fh:
        for line in fh:
            parts = line.split()
            if not parts:
                continue
            if len(parts) < 4:
                raise ValueError("%s: expected `t rgb_file t depth_file`, got %r" % (path, line))
            out.append((float(parts[0]), parts[1], parts[3]))
    return out
