"""Generates the image fixtures under tests/golden/images/ and tests/golden/images.npz: what the reference's
Image(path) (stb_image 2.27, compiled into oracle/_ref/ref_dump, mode --image) decodes them to.  The files are made
here (PNG by the small writer below, JPEG by Pillow); none is one of the reference's data files.  Run where
/root/reference exists; files and result are committed.

    python oracle/gen_image_golden.py
"""
import io
import os
import struct
import subprocess
import tempfile
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
OUT = os.path.join(ROOT, "tests", "golden", "images")


# ---------------------------------------------------------------------------------------------------------
# PNG writer: any colour type / bit depth / interlace, chosen row filters, optional PLTE / tRNS, split IDAT
# ---------------------------------------------------------------------------------------------------------
def _chunk(kind, body):
    return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    if pa <= pb and pa <= pc:
        return a
    return b if pb <= pc else c


def _pack_rows(samples, depth):
    """samples: (h, w * channels) integers below 2**depth -> list of packed byte rows"""
    rows = []
    for r in samples:
        if depth == 8:
            rows.append(bytes(int(v) for v in r))
        elif depth == 16:
            rows.append(b"".join(struct.pack(">H", int(v)) for v in r))
        else:
            bits = "".join(format(int(v), "0%db" % depth) for v in r)
            bits += "0" * (-len(bits) % 8)
            rows.append(bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)))
    return rows


def _filter_rows(rows, bpp, filters):
    out = bytearray()
    prev = bytes(len(rows[0])) if rows else b""
    for j, row in enumerate(rows):
        f = filters[j % len(filters)]
        out.append(f)
        for k, v in enumerate(row):
            a = row[k - bpp] if k >= bpp else 0
            b = prev[k]
            c = prev[k - bpp] if k >= bpp else 0
            pred = (0, a, b, (a + b) >> 1, _paeth(a, b, c))[f]
            out.append((v - pred) & 0xFF)
        prev = row
    return bytes(out)


def write_png(samples, colour, depth, interlace=False, filters=(0, 1, 2, 3, 4), palette=None, trns=None, level=6, split=0, extra=()):
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[colour]
    h, wc = samples.shape
    w = wc // channels
    bpp = max(1, channels * depth // 8)
    px = samples.reshape(h, w, channels)
    raw = b""
    if not interlace:
        raw = _filter_rows(_pack_rows(px.reshape(h, -1), depth), bpp, filters)
    else:
        for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = px[y0::dy, x0::dx]
            if sub.shape[0] == 0 or sub.shape[1] == 0:
                continue
            raw += _filter_rows(_pack_rows(sub.reshape(sub.shape[0], -1), depth), bpp, filters)
    data = zlib.compress(raw, level)
    out = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, colour, 0, 0, 1 if interlace else 0))
    for kind, body in extra:
        out += _chunk(kind, body)
    if palette is not None:
        out += _chunk(b"PLTE", bytes(int(v) for v in np.asarray(palette).reshape(-1)))
    if trns is not None:
        out += _chunk(b"tRNS", trns)
    if split:
        for i in range(0, len(data), split):
            out += _chunk(b"IDAT", data[i:i + split])
    else:
        out += _chunk(b"IDAT", data)
    return out + _chunk(b"IEND", b"")


def png_fixtures(rng):
    files = {}

    def img(h, w, channels, depth):
        return rng.integers(0, 2 ** depth, size=(h, w * channels))

    for depth in (1, 2, 4, 8, 16):
        files[f"grey{depth}.png"] = write_png(img(9, 13, 1, depth), 0, depth)
        files[f"grey{depth}_interlaced.png"] = write_png(img(11, 10, 1, depth), 0, depth, interlace=True)
    for depth in (8, 16):
        files[f"rgb{depth}.png"] = write_png(img(7, 9, 3, depth), 2, depth)
        files[f"rgba{depth}.png"] = write_png(img(6, 5, 4, depth), 6, depth)
        files[f"greyalpha{depth}.png"] = write_png(img(5, 8, 2, depth), 4, depth)
        files[f"rgb{depth}_interlaced.png"] = write_png(img(9, 9, 3, depth), 2, depth, interlace=True)
        files[f"rgba{depth}_interlaced.png"] = write_png(img(3, 17, 4, depth), 6, depth, interlace=True)
    for depth in (1, 2, 4, 8):
        pal = rng.integers(0, 256, size=(2 ** depth, 3))
        files[f"palette{depth}.png"] = write_png(img(10, 11, 1, depth), 3, depth, palette=pal)
        files[f"palette{depth}_trns_interlaced.png"] = write_png(img(9, 6, 1, depth), 3, depth, interlace=True, palette=pal, trns=bytes(rng.integers(0, 256, size=2 ** depth - 1).tolist()))
    # a short palette: indices beyond it are not used by the picture
    files["palette8_short.png"] = write_png(rng.integers(0, 5, size=(4, 4)), 3, 8, palette=rng.integers(0, 256, size=(5, 3)))
    # colour-key transparency (ignored for an RGB result), ancillary chunks, one pixel, one column, one row
    files["rgb8_trns.png"] = write_png(img(4, 4, 3, 8), 2, 8, trns=struct.pack(">HHH", 1, 2, 3))
    files["grey4_trns.png"] = write_png(img(4, 5, 1, 4), 0, 4, trns=struct.pack(">H", 3))
    files["rgb8_ancillary.png"] = write_png(img(3, 3, 3, 8), 2, 8, extra=((b"gAMA", struct.pack(">I", 45455)), (b"tEXt", b"Comment\0fixture")))
    files["one_pixel.png"] = write_png(img(1, 1, 3, 8), 2, 8)
    files["one_pixel_interlaced.png"] = write_png(img(1, 1, 1, 1), 0, 1, interlace=True)
    files["column_interlaced.png"] = write_png(img(19, 1, 1, 2), 0, 2, interlace=True)
    files["row.png"] = write_png(img(1, 23, 1, 1), 0, 1)
    # each filter type alone, stored (uncompressed) deflate blocks, fixed-Huffman blocks, IDAT cut into pieces
    for f in range(5):
        files[f"rgb8_filter{f}.png"] = write_png(img(6, 6, 3, 8), 2, 8, filters=(f,))
    files["rgb8_stored.png"] = write_png(img(8, 8, 3, 8), 2, 8, level=0)
    smooth = (np.add.outer(np.arange(64), np.arange(64 * 3)) // 3 % 256)
    files["rgb8_smooth.png"] = write_png(smooth, 2, 8, level=9, split=100)
    files["grey8_tiny_fixed.png"] = write_png(np.zeros((2, 3), dtype=int), 0, 8, level=1)
    # broken files
    good = write_png(img(5, 5, 3, 8), 2, 8)
    files["broken_truncated.png"] = good[: len(good) - 30]
    files["broken_filter.png"] = _bad_filter()
    files["broken_no_idat.png"] = good[:33] + _chunk(b"IEND", b"")
    files["broken_depth.png"] = good[:24] + b"\x03" + good[25:]
    files["broken_critical_chunk.png"] = good[:33] + _chunk(b"ABCD", b"12") + good[33:]
    files["broken_not_an_image.png"] = b"this is not an image file at all\n"
    return files


def _bad_filter():
    raw = bytes([9, 1, 2, 3])  # filter type 9 on a one-pixel row
    return b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", 1, 1, 8, 2, 0, 0, 0)) + _chunk(b"IDAT", zlib.compress(raw)) + _chunk(b"IEND", b"")


# ---------------------------------------------------------------------------------------------------------
# JPEG files by Pillow
# ---------------------------------------------------------------------------------------------------------
def jpeg_fixtures(rng):
    from PIL import Image

    files = {}

    def picture(h, w, mode="RGB"):
        # smooth gradients plus some noise and a few hard edges, so that all coefficient sizes occur
        y, x = np.mgrid[0:h, 0:w]
        ch = {"L": 1, "RGB": 3, "CMYK": 4}[mode]
        a = np.stack([(np.sin(x / (3.0 + c) + c) * 60 + np.cos(y / (2.0 + c)) * 60 + 128) for c in range(ch)], axis=-1)
        a += rng.normal(0, 12, size=a.shape)
        a[h // 3: h // 2, w // 4: w // 2] = 255 - a[h // 3: h // 2, w // 4: w // 2]
        a = np.clip(a, 0, 255).astype(np.uint8)
        return Image.fromarray(a[..., 0] if ch == 1 else a, mode)

    def save(im, **kw):
        b = io.BytesIO()
        im.save(b, "JPEG", **kw)
        return b.getvalue()

    for name, sub in (("444", "4:4:4"), ("422", "4:2:2"), ("420", "4:2:0")):
        files[f"ycc_{name}.jpg"] = save(picture(37, 53), quality=85, subsampling=sub)
        files[f"ycc_{name}_progressive.jpg"] = save(picture(41, 35), quality=70, subsampling=sub, progressive=True)
    for name, sub in (("411", "4:1:1"),):
        try:
            files[f"ycc_{name}.jpg"] = save(picture(33, 49), quality=80, subsampling=sub)
        except Exception as e:  # older Pillow: sampling not offered
            print("skipped", name, e)
    files["grey.jpg"] = save(picture(29, 31, "L"), quality=90)
    files["grey_progressive.jpg"] = save(picture(40, 24, "L"), quality=60, progressive=True)
    files["quality_low.jpg"] = save(picture(64, 64), quality=5, subsampling="4:2:0")
    files["quality_100.jpg"] = save(picture(48, 40), quality=100, subsampling="4:4:4")
    files["optimized.jpg"] = save(picture(50, 50), quality=75, optimize=True)
    files["one_pixel.jpg"] = save(picture(1, 1), quality=90, subsampling="4:2:0")
    files["narrow.jpg"] = save(picture(17, 1), quality=90, subsampling="4:2:0")
    files["flat_17x9.jpg"] = save(picture(9, 17), quality=90, subsampling="4:2:2")
    files["width_2.jpg"] = save(picture(3, 2), quality=90, subsampling="4:2:0")
    files["cmyk.jpg"] = save(picture(20, 28, "CMYK"), quality=85)
    try:
        files["rgb_direct.jpg"] = save(picture(24, 24), quality=90, keep_rgb=True)
    except Exception as e:
        print("skipped keep_rgb", e)
    try:
        files["restart_rows.jpg"] = save(picture(64, 48), quality=80, subsampling="4:2:0", restart_marker_rows=1)
        files["restart_blocks.jpg"] = save(picture(40, 56), quality=80, subsampling="4:4:4", restart_marker_blocks=3)
        files["restart_progressive.jpg"] = save(picture(40, 40), quality=80, progressive=True, restart_marker_blocks=2)
    except Exception as e:
        print("skipped restart markers", e)
    good = files["ycc_420.jpg"]
    files["broken_truncated.jpg"] = good[: len(good) // 2]
    sof = good.index(b"\xff\xc0")
    files["broken_12_bit.jpg"] = good[: sof + 4] + b"\x0c" + good[sof + 5:]  # sample precision 12: refused
    files["broken_no_soi.jpg"] = b"\xff\xd9" + good[2:]
    return files


# ---------------------------------------------------------------------------------------------------------
# Baseline JPEG writer for what Pillow does not offer: any sampling factors, one scan per component, 16-bit
# quantisation tables, restart intervals in such files.  Huffman tables: the ones of a Pillow file (Annex K).
# ---------------------------------------------------------------------------------------------------------
_ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
           30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


def _huffman_tables_of(jpeg):
    """{(class, slot): (counts[16], values)} from the DHT segments of a JPEG file"""
    tables, pos = {}, 2
    while jpeg[pos + 1] != 0xDA:
        length = struct.unpack(">H", jpeg[pos + 2:pos + 4])[0]
        if jpeg[pos + 1] == 0xC4:
            p, end = pos + 4, pos + 2 + length
            while p < end:
                counts = list(jpeg[p + 1:p + 17])
                n = sum(counts)
                tables[(jpeg[p] >> 4, jpeg[p] & 15)] = (counts, list(jpeg[p + 17:p + 17 + n]))
                p += 17 + n
        pos += 2 + length
    return tables


def _codes(counts, values):
    out, code, k = {}, 0, 0
    for length in range(1, 17):
        for _ in range(counts[length - 1]):
            out[values[k]] = (code, length)
            code += 1
            k += 1
        code <<= 1
    return out


class _Bits:
    def __init__(self):
        self.out, self.acc, self.n = bytearray(), 0, 0

    def put(self, value, length):
        self.acc = (self.acc << length) | (value & ((1 << length) - 1))
        self.n += length
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 0xFF
            self.out.append(b)
            if b == 0xFF:
                self.out.append(0)
            self.n -= 8

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)


def write_jpeg(planes, factors, width, height, quant, tables, one_scan_per_component=False, wide_quant=False, restart=0):
    """planes[i]: the samples of component i at its own resolution ((height * v_i / v_max) x (width * h_i / h_max),
    rounded up); factors[i] = (h_i, v_i)."""
    n = len(planes)
    hmax, vmax = max(f[0] for f in factors), max(f[1] for f in factors)
    mx, my = -(-width // (8 * hmax)), -(-height // (8 * vmax))
    k = np.arange(8)
    C = np.cos((2 * k[None, :] + 1) * k[:, None] * np.pi / 16) * np.where(k[:, None] == 0, np.sqrt(1 / 8), np.sqrt(2 / 8))
    q = np.asarray(quant).reshape(8, 8)
    coeffs = []
    for (h, v), plane in zip(factors, planes):
        bw, bh = mx * h, my * v
        padded = np.pad(plane, ((0, bh * 8 - plane.shape[0]), (0, bw * 8 - plane.shape[1])), mode="edge").astype(float) - 128
        blocks = np.empty((bh, bw, 64), dtype=int)
        for by in range(bh):
            for bx in range(bw):
                d = C @ padded[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8] @ C.T
                blocks[by, bx] = np.rint(d / q).astype(int).reshape(64)[_ZIGZAG]
        coeffs.append(blocks)
    dc, ac = _codes(*tables[(0, 0)]), _codes(*tables[(1, 0)])
    out = bytearray(b"\xff\xd8")
    zz = [int(q.reshape(64)[i]) for i in _ZIGZAG]
    if wide_quant:
        out += b"\xff\xdb" + struct.pack(">HB", 2 + 1 + 128, 0x10) + b"".join(struct.pack(">H", v) for v in zz)
    else:
        out += b"\xff\xdb" + struct.pack(">HB", 2 + 1 + 64, 0x00) + bytes(zz)
    out += b"\xff\xc0" + struct.pack(">HBHHB", 8 + 3 * n, 8, height, width, n)
    for i, (h, v) in enumerate(factors):
        out += bytes([i + 1, (h << 4) | v, 0])
    for (cls, slot), (counts, values) in tables.items():
        if slot == 0:
            out += b"\xff\xc4" + struct.pack(">HB", 2 + 1 + 16 + len(values), (cls << 4)) + bytes(counts) + bytes(values)
    if restart:
        out += b"\xff\xdd" + struct.pack(">HH", 4, restart)

    def size_and_bits(v):
        a = abs(v)
        s = a.bit_length()
        return s, (v if v >= 0 else v + (1 << s) - 1)

    def scan(components):
        out.extend(b"\xff\xda" + struct.pack(">HB", 6 + 2 * len(components), len(components)))
        for c in components:
            out.extend(bytes([c + 1, 0x00]))
        out.extend(bytes([0, 63, 0]))
        bits, pred, done, rst = _Bits(), [0] * n, 0, 0

        def block(c, data):
            s, b = size_and_bits(int(data[0]) - pred[c])
            pred[c] = int(data[0])
            bits.put(*dc[s])
            if s:
                bits.put(b, s)
            run = 0
            last = max([i for i in range(1, 64) if data[i]], default=0)
            for i in range(1, last + 1):
                if data[i] == 0:
                    run += 1
                    continue
                while run > 15:
                    bits.put(*ac[0xF0])
                    run -= 16
                s, b = size_and_bits(int(data[i]))
                bits.put(*ac[(run << 4) | s])
                bits.put(b, s)
                run = 0
            if last < 63:
                bits.put(*ac[0x00])

        def mcu_end(is_last):
            nonlocal done, rst, pred
            done += 1
            if restart and done % restart == 0 and not is_last:
                bits.flush()
                bits.out.extend(bytes([0xFF, 0xD0 + rst % 8]))
                rst += 1
                pred = [0] * n

        if len(components) == 1:
            c = components[0]
            h, v = factors[c]
            bw, bh = -(-(-(-width * h // hmax)) // 8), -(-(-(-height * v // vmax)) // 8)
            for by in range(bh):
                for bx in range(bw):
                    block(c, coeffs[c][by, bx])
                    mcu_end(by == bh - 1 and bx == bw - 1)
        else:
            for j in range(my):
                for i in range(mx):
                    for c in components:
                        h, v = factors[c]
                        for y in range(v):
                            for x in range(h):
                                block(c, coeffs[c][j * v + y, i * h + x])
                    mcu_end(j == my - 1 and i == mx - 1)
        bits.flush()
        out.extend(bits.out)

    if one_scan_per_component:
        for c in range(n):
            scan([c])
    else:
        scan(list(range(n)))
    return bytes(out + b"\xff\xd9")


def own_jpeg_fixtures(rng, pillow_file):
    tables = _huffman_tables_of(pillow_file)
    files = {}
    quant = np.clip(np.add.outer(np.arange(8), np.arange(8)) * 3 + 4, 1, 255)

    def planes(width, height, factors):
        hmax, vmax = max(f[0] for f in factors), max(f[1] for f in factors)
        out = []
        for c, (h, v) in enumerate(factors):
            w, ht = -(-width * h // hmax), -(-height * v // vmax)
            y, x = np.mgrid[0:ht, 0:w]
            p = np.sin(x / (2.0 + c)) * 70 + np.cos(y / (3.0 + c)) * 50 + 128 + rng.normal(0, 10, size=(ht, w))
            out.append(np.clip(p, 0, 255).astype(np.uint8))
        return out

    cases = {
        "own_440": ((1, 2), (1, 1), (1, 1)),     # chroma halved vertically only
        "own_410": ((4, 1), (1, 1), (1, 1)),     # factor 4 horizontally
        "own_4x4": ((4, 4), (1, 1), (2, 2)),     # factors 4x4 and 2x2 in one picture
        "own_2x4": ((2, 4), (2, 1), (1, 2)),
        "own_chroma_full": ((1, 1), (2, 2), (2, 2)),  # luma is the subsampled one
        "own_422": ((2, 1), (1, 1), (1, 1)),
        "own_420": ((2, 2), (1, 1), (1, 1)),
    }
    for name, factors in cases.items():
        w, h = int(rng.integers(20, 45)), int(rng.integers(18, 40))
        files[name + ".jpg"] = write_jpeg(planes(w, h, factors), factors, w, h, quant, tables)
        files[name + "_separate_scans.jpg"] = write_jpeg(planes(w, h, factors), factors, w, h, quant, tables, one_scan_per_component=True)
    f = cases["own_420"]
    files["own_wide_quant.jpg"] = write_jpeg(planes(30, 30, f), f, 30, 30, quant * 40, tables, wide_quant=True)
    files["own_restart.jpg"] = write_jpeg(planes(50, 34, f), f, 50, 34, quant, tables, restart=2)
    files["own_restart_separate.jpg"] = write_jpeg(planes(33, 33, f), f, 33, 33, quant, tables, one_scan_per_component=True, restart=3)
    files["own_grey_2x2.jpg"] = write_jpeg(planes(21, 19, ((2, 2),)), ((2, 2),), 21, 19, quant, tables)
    files["own_saturating.jpg"] = write_jpeg([np.where(np.indices((16, 16)).sum(0) % 2, 255, 0).astype(np.uint8)] * 3, ((1, 1),) * 3, 16, 16, np.ones((8, 8), dtype=int), tables)
    return files


def main():
    rng = np.random.default_rng(20240611)
    files = {}
    files.update(png_fixtures(rng))
    files.update(jpeg_fixtures(rng))
    files.update(own_jpeg_fixtures(rng, files["ycc_444.jpg"]))
    os.makedirs(OUT, exist_ok=True)
    for old in os.listdir(OUT):
        os.remove(os.path.join(OUT, old))
    golden = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, data in sorted(files.items()):
            path = os.path.join(OUT, name)
            with open(path, "wb") as f:
                f.write(data)
            npy = os.path.join(tmp, "out.npy")
            rc = subprocess.run([REF, "--image", path, npy], stderr=subprocess.DEVNULL).returncode
            if rc == 0:
                golden[name] = np.load(npy)
            elif rc == 3:
                golden[name] = np.zeros((0, 0, 3), dtype=np.uint8)  # the reference refuses the file
            else:
                raise RuntimeError(f"ref_dump --image {name}: exit status {rc}")
            print(f"{name:36s} {len(data):7d} bytes -> {'refused' if rc else golden[name].shape}")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "images.npz"), **golden)
    print("total bytes", sum(len(d) for d in files.values()))


if __name__ == "__main__":
    main()
