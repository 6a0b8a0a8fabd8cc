"""Generates the small image fixtures under tests/golden/images (run in the build container; PIL writes the JPEGs, the PNGs
come from the minimal encoder below so that every colour type, bit depth, scanline filter and Adam7 are covered).
The expected texels are NOT produced here: oracle/gen_golden.py images hashes what the reference's own image.cpp returns
for these files (oracle/ref_img.cpp) into tests/golden/ref_images.json."""
import os, struct, zlib
import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "images")
os.makedirs(OUT, exist_ok=True)
rng = np.random.default_rng(5)
W, H = 45, 37        # ragged against 8x8 blocks, 16x16 MCUs and the Adam7 lattice
y, x = np.mgrid[0:H, 0:W]
base = np.stack([(x * 5 + y * 2) % 256, (255 - x * 3 + (y * y) // 7) % 256, ((x * y) // 3 + 40 * np.sin(x / 5.0)) % 256], -1).astype(np.uint8)
noise = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
pic = ((base.astype(int) * 3 + noise) // 4).astype(np.uint8)
grey = ((pic[..., 0].astype(int) * 77 + pic[..., 1] * 150 + pic[..., 2] * 29) >> 8).astype(np.uint8)


def chunk(t, body):
    return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body))


def filt(rows, bpp):
    """Filters the byte rows with types 0..4 in turn (a conforming decoder must undo whichever is chosen)."""
    out, prev = bytearray(), np.zeros_like(rows[0], dtype=np.int64)
    for j, row in enumerate(rows):
        row = row.astype(np.int64)
        ft = j % 5
        left = np.concatenate([np.zeros(bpp, np.int64), row[:-bpp]]) if len(row) > bpp else np.zeros_like(row)
        ul = np.concatenate([np.zeros(bpp, np.int64), prev[:-bpp]]) if len(row) > bpp else np.zeros_like(row)
        if ft == 0: f = row
        elif ft == 1: f = row - left
        elif ft == 2: f = row - prev
        elif ft == 3: f = row - ((left + prev) >> 1)
        else:
            p = left + prev - ul
            pa, pb, pc = abs(p - left), abs(p - prev), abs(p - ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
            f = row - pred
        out.append(ft); out += bytes((f & 255).astype(np.uint8))
        prev = row
    return bytes(out)


def pack(samples, depth):
    """samples: (h, w, n) integer array -> list of byte rows, most significant bits first."""
    h, w, n = samples.shape
    rows = []
    for j in range(h):
        s = samples[j].reshape(-1).astype(np.uint32)
        if depth == 16: rows.append(np.stack([s >> 8, s & 255], -1).reshape(-1).astype(np.uint8))
        elif depth == 8: rows.append(s.astype(np.uint8))
        else:
            per = 8 // depth
            pad = (-len(s)) % per
            s = np.concatenate([s, np.zeros(pad, np.uint32)]).reshape(-1, per)
            rows.append(sum(s[:, k] << (depth * (per - 1 - k)) for k in range(per)).astype(np.uint8))
    return rows


def write_png(name, samples, color, depth, interlace=False, plte=None, trns=None):
    h, w, n = samples.shape
    bpp = max(1, n * depth // 8)
    if not interlace:
        raw = filt(pack(samples, depth), bpp)
    else:
        raw = b""
        for xo, yo, xs, ys in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = samples[yo::ys, xo::xs]
            if sub.shape[0] and sub.shape[1]:
                raw += filt(pack(sub, depth), bpp)
    body = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color, 0, 0, 1 if interlace else 0))
    body += chunk(b"gAMA", struct.pack(">I", 45455))          # ancillary: ignored by the reference's decoder
    if plte is not None: body += chunk(b"PLTE", bytes(np.asarray(plte, np.uint8).reshape(-1)))
    if trns is not None: body += chunk(b"tRNS", trns)
    z = zlib.compress(raw, 9)
    body += chunk(b"IDAT", z[:len(z) // 2]) + chunk(b"IDAT", z[len(z) // 2:]) + chunk(b"IEND", b"")
    open(os.path.join(OUT, name), "wb").write(body)


pal = rng.integers(0, 256, (200, 3), dtype=np.uint8)
idx8 = (grey.astype(int) * 199 // 255)[..., None]
deep = (pic.astype(np.uint32) * 257 + noise[..., ::-1]) & 0xFFFF
write_png("rgb8.png", pic, 2, 8)
write_png("rgb8_adam7.png", pic, 2, 8, interlace=True)
write_png("rgb8_key.png", pic, 2, 8, trns=struct.pack(">HHH", *[int(v) for v in pic[3, 4]]))
write_png("rgba8.png", np.dstack([pic, noise[..., 0]]), 6, 8)
write_png("rgba8_adam7.png", np.dstack([pic, noise[..., 0]]), 6, 8, interlace=True)
write_png("rgb16.png", deep, 2, 16)
write_png("rgba16_adam7.png", np.dstack([deep, deep[..., 0] ^ 0x5a5a]), 6, 16, interlace=True)
write_png("grey8.png", grey[..., None], 0, 8)
write_png("grey16.png", deep[..., 1:2], 0, 16)
write_png("grey16_key.png", deep[..., 1:2], 0, 16, trns=struct.pack(">H", int(deep[5, 6, 1])))
write_png("greyalpha8.png", np.dstack([grey, noise[..., 1]]), 4, 8)
write_png("greyalpha16.png", np.dstack([deep[..., 2], deep[..., 0]]), 4, 16)
for d in (1, 2, 4):
    g = (grey.astype(int) >> (8 - d))[..., None]
    write_png(f"grey{d}.png", g, 0, d)
    write_png(f"grey{d}_adam7.png", g, 0, d, interlace=True)
    write_png(f"palette{d}.png", g, 3, d, plte=pal[:1 << d])
write_png("palette8.png", idx8, 3, 8, plte=pal)
write_png("palette8_adam7_trns.png", idx8, 3, 8, interlace=True, plte=pal, trns=bytes(range(0, 200)))
write_png("palette4_adam7.png", (grey.astype(int) >> 4)[..., None], 3, 4, interlace=True, plte=pal[:16])
write_png("tiny_1x1.png", pic[:1, :1], 2, 8, interlace=True)
write_png("narrow_3x9_grey2_adam7.png", (grey[:9, :3].astype(int) >> 6)[..., None], 0, 2, interlace=True)

rgb = Image.fromarray(pic, "RGB")
rgb.save(os.path.join(OUT, "prog_420.jpg"), progressive=True, quality=70, subsampling=2)
rgb.save(os.path.join(OUT, "prog_444.jpg"), progressive=True, quality=92, subsampling=0)
rgb.save(os.path.join(OUT, "prog_422.jpg"), progressive=True, quality=50, subsampling=1)
Image.fromarray(grey, "L").save(os.path.join(OUT, "prog_grey.jpg"), progressive=True, quality=80)
Image.fromarray(np.tile(pic, (3, 4, 1))[:101, :150], "RGB").save(os.path.join(OUT, "prog_big_q35.jpg"), progressive=True, quality=35, subsampling=2)
rgb.save(os.path.join(OUT, "base_420_optimized.jpg"), quality=75, subsampling=2, optimize=True)
for f in sorted(os.listdir(OUT)):
    print(f, os.path.getsize(os.path.join(OUT, f)))
