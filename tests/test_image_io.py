"""imwrite drop-in: PFM (fp32, '-1' scale line, rows as stored) and EXR (RGB half, scanline) by suffix
(src/image.cpp:135-173). Mirrors the reference's image round-trip test (src/tests/image.cpp:4-36)."""
import struct

import numpy as np


def ramp(w=32, h=24):
    y, x = np.mgrid[0:h, 0:w]
    return np.stack([x / w, y / h, (x + y) / (w + h)], axis=-1).astype(np.float64)


def test_pfm_layout(G, tmp_path):
    img = ramp()
    p = tmp_path / "o.pfm"
    G.imwrite(str(p), img)
    raw = p.read_bytes()
    assert raw.startswith(b"PF\n32 24\n-1\n")
    data = np.frombuffer(raw[len(b"PF\n32 24\n-1\n"):], dtype="<f4").reshape(24, 32, 3)
    assert np.array_equal(data, img.astype(np.float32))          # top row first, as stored


def test_exr_half_scanline(G, tmp_path):
    img = ramp() * 3.0 - 0.5
    img[0, 0] = [65504.0, 1e-8, -2.0]
    p = tmp_path / "o.exr"
    G.imwrite(str(p), img)
    raw = p.read_bytes()
    assert struct.unpack("<I", raw[:4])[0] == 20000630 and raw[4] == 2
    assert b"channels\x00chlist\x00" in raw and b"compression\x00compression\x00" in raw
    # parse: header ends with a lone NUL after the last attribute; then the offset table and the scanlines
    pos = 8
    while raw[pos] != 0:
        pos = raw.index(b"\x00", pos) + 1          # name
        pos = raw.index(b"\x00", pos) + 1          # type
        size = struct.unpack("<i", raw[pos:pos + 4])[0]
        pos += 4 + size
    pos += 1
    w, h = 32, 24
    offsets = struct.unpack("<%dQ" % h, raw[pos:pos + 8 * h])
    got = np.empty((h, w, 3), dtype=np.float16)
    for y in range(h):
        o = offsets[y]
        yy, nbytes = struct.unpack("<ii", raw[o:o + 8])
        assert yy == y and nbytes == w * 3 * 2
        line = np.frombuffer(raw[o + 8:o + 8 + nbytes], dtype="<f2").reshape(3, w)     # B, G, R planes
        got[y, :, 2], got[y, :, 1], got[y, :, 0] = line[0], line[1], line[2]
    want = img.astype(np.float32).astype(np.float16)     # round-to-nearest-even, like numpy
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))
    # fp16 storage keeps the reference test's 1e-2 tolerance
    assert np.max(np.abs(got[1:].astype(np.float64) - img[1:])) < 1e-2


def test_unknown_suffix_writes_nothing(G, tmp_path):
    p = tmp_path / "o.png"
    G.imwrite(str(p), ramp())
    assert not p.exists()
