"""imwrite drop-in: PFM (fp32, '-1' scale line, rows as stored) and EXR (RGB half, scanline) by suffix
(src/image.cpp:135-173). Mirrors the reference's image round-trip test (src/tests/image.cpp:4-36)."""
import struct

import pytest
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


def half_like_tinyexr(a):
    """fp32 -> fp16 as the reference's writer rounds (tinyexr float_to_half_full, src/3rdparty/tinyexr.h:889-924): the
    first dropped bit decides, ties go away from zero; float subnormals flush to zero."""
    x = np.asarray(a, dtype=np.float32).view(np.uint32).astype(np.int64)
    sign = (x >> 16) & 0x8000
    e = (x >> 23) & 0xFF
    m = x & 0x7FFFFF
    ne = e - 127 + 15
    h = np.zeros_like(x)
    normal = (e > 0) & (e < 255) & (ne > 0) & (ne < 31)
    h = np.where(normal, (ne.clip(0, 31) << 10) | (m >> 13), h)
    h = np.where(normal & ((m & 0x1000) != 0), h + 1, h)
    sub = (e > 0) & (e < 255) & (ne <= 0) & (14 - ne <= 24)
    sh = (14 - ne).clip(1, 62)
    mant = m | 0x800000
    h = np.where(sub, (mant >> sh) + ((mant >> (sh - 1)) & 1), h)
    h = np.where((e > 0) & (e < 255) & (ne >= 31), 0x7C00, h)
    h = np.where(e == 255, 0x7C00 | np.where(m != 0, 0x200, 0), h)
    return (sign | h).astype(np.uint16).view(np.float16)


def test_exr_write_read_round_trip(G, tmp_path):
    """ZIP-compressed half scanline file (what tinyexr's SaveEXR writes for images >= 16 pixels in an extent,
    src/image.cpp:155-171) read back by the build's own reader; rounding as the reference's writer does it."""
    img = ramp() * 3.0 - 0.5
    img[0, 0] = [65504.0, 1e-8, -2.0]
    img[0, 1] = [3.1259765625, -3.1259765625, 70000.0]      # exact ties (go away from zero) and overflow to inf
    img[0, 2] = [6.1e-5, 5.96e-8, 2.98e-8]                   # half subnormals
    p = tmp_path / "o.exr"
    G.imwrite(str(p), img)
    raw = p.read_bytes()
    assert struct.unpack("<I", raw[:4])[0] == 20000630 and raw[4] == 2
    assert b"compression\x00compression\x00\x01\x00\x00\x00\x03" in raw              # ZIP
    got = G.imread(str(p), 3)
    want = half_like_tinyexr(img.astype(np.float32)).astype(np.float64)
    assert np.array_equal(got, want)
    assert got[0, 1, 0] == 3.126953125 and got[0, 1, 1] == -3.126953125 and np.isinf(got[0, 1, 2])
    assert np.max(np.abs(got[1:] - img[1:])) < 1e-2          # the reference test's tolerance (src/tests/image.cpp)
    g1 = G.imread(str(p), 1)[..., 0]                          # imread1: (R+G+B)/3 in fp32 (src/image.cpp:63-65)
    w32 = want.astype(np.float32)
    assert np.array_equal(g1[1:], ((w32[1:, :, 0] + w32[1:, :, 1] + w32[1:, :, 2]) / np.float32(3)).astype(np.float64))


def test_exr_small_images_are_uncompressed(G, tmp_path):
    img = ramp(8, 5)
    p = tmp_path / "s.exr"
    G.imwrite(str(p), img)
    raw = p.read_bytes()
    assert b"compression\x00compression\x00\x01\x00\x00\x00\x00" in raw              # NONE below 16x16
    assert np.array_equal(G.imread(str(p), 3), half_like_tinyexr(img.astype(np.float32)).astype(np.float64))


def _golden():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_images.json")))


def _test_picture(w, h):
    x = np.arange(w, dtype=np.float64)[None, :]
    y = np.arange(h, dtype=np.float64)[:, None]
    u, v = (x + 0.5) / w, (y + 0.5) / h
    img = np.zeros((h, w, 3))
    img[..., 0] = u * u * 3.0 - 0.25
    img[..., 1] = (v - 0.5) * 1e-5 + 0 * u
    img[..., 2] = 100.0 * u * v + 1.0 / 1024.0
    img[1, 1, 2] += 70000.0
    return img


def test_exr_against_reference_written_files(G, tmp_path):
    """tests/golden/ref_images.json holds EXR files written by the reference's own imwrite (tinyexr) and the pixel
    checksum the reference's reader computes from them and from THIS build's files of the same picture."""
    import base64
    import zlib
    for key, rec in _golden()["exr"].items():
        w, h = rec["width"], rec["height"]
        ref_file = tmp_path / f"ref_{key}.exr"
        ref_file.write_bytes(base64.b64decode(rec["reference_file_b64"]))
        seen = G.imread(str(ref_file), 3)                    # own reader on the reference's file
        assert seen.shape == (h, w, 3)
        assert zlib.crc32(seen.astype(np.float32).tobytes()) == rec["pixels_crc32_as_read_by_reference"]
        own = tmp_path / f"own_{key}.exr"
        G.imwrite(str(own), _test_picture(w, h))            # own writer: the bytes the reference reader validated
        assert zlib.crc32(own.read_bytes()) == rec["own_file_crc32"]
        assert np.array_equal(G.imread(str(own), 3), seen)


def test_piz_compressed_exr_matches_the_reference_reader(G):
    """scenes/matpreview/envmap.exr (PIZ: wavelet + Huffman, the environment map of the reference's Disney / matpreview
    scenes) through the build's own reader == the pixels tinyexr's LoadEXR hands the reference (CRC-32 of fp32 RGB)."""
    import os
    import zlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rec = _golden()["exr_inputs"]["matpreview/envmap.exr"]
    a = G.imread(os.path.join(root, "scenes", "matpreview", "envmap.exr"), 3)
    assert a.shape == (rec["height"], rec["width"], 3)
    assert zlib.crc32(a.astype(np.float32).tobytes()) == rec["pixels_crc32"]
    assert np.allclose(a.mean(axis=(0, 1)), rec["mean"], rtol=1e-12)


def test_jpeg_decoder_matches_the_reference_decoder(G):
    """Every sponza texture through the build's baseline JPEG decoder == what the reference's imread3 / imread1
    (stb_image v2.27 as vendored there) return: CRC-32 of the fp32 texels, generated by oracle/ref_img.cpp."""
    import os
    import zlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = _golden()
    assert len(gold["imread3"]) >= 11
    for name, rec in gold["imread3"].items():
        path = os.path.join(root, "scenes", "sponza", "textures", name)
        a = G.imread(path, 3)
        assert a.shape == (rec["height"], rec["width"], 3)
        assert zlib.crc32(a.astype(np.float32).tobytes()) == rec["crc32"], name
        assert np.allclose(a[0, :2].reshape(-1), rec["first"], rtol=1e-7)
        a1 = G.imread(path, 1)
        assert zlib.crc32(a1.astype(np.float32).tobytes()) == gold["imread1"][name]["crc32"], name


def test_jpeg_decoder_rejects_what_it_does_not_decode(G, tmp_path):
    import pytest
    p = tmp_path / "bad.jpg"
    p.write_bytes(b"\xff\xd8\xff\xc9\x00\x0b\x08\x00\x10\x00\x10\x01\x01\x11\x00\xff\xd9")     # SOF9: arithmetic coding (stb_image refuses it too)
    with pytest.raises(G.GdptError):
        G.imread(str(p), 3)
    p.write_bytes(b"\xff\xd8\xff\xc2\x00\x0b\x08\x00\x10\x00\x10\x01\x01\x11\x00\xff\xd9")     # SOF2 without a scan
    with pytest.raises(G.GdptError, match="without image data"):
        G.imread(str(p), 3)
    p.write_bytes(b"not a jpeg")
    with pytest.raises(G.GdptError):
        G.imread(str(p), 3)


def test_unknown_suffix_writes_nothing(G, tmp_path):
    p = tmp_path / "o.png"
    G.imwrite(str(p), ramp())
    assert not p.exists()


def test_decoders_reject_corrupt_input_without_crashing(G, tmp_path):
    """Mutated JPEG / EXR files either decode or raise GdptError (the same harness ran 6800 mutations under
    AddressSanitizer + UBSan on the CPU build of the readers; this is the in-process smoke of it)."""
    import os
    rng = np.random.default_rng(11)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    jpg = open(os.path.join(root, "scenes", "sponza", "textures", "01_S_ba.JPG"), "rb").read()
    exr_path = tmp_path / "src.exr"
    G.imwrite(str(exr_path), ramp(33, 40) * 4 - 1)
    exr = exr_path.read_bytes()
    outcomes = {"ok": 0, "rejected": 0}
    piz = open(os.path.join(root, "scenes", "matpreview", "envmap.exr"), "rb").read()
    for kind, blob in (("jpg", jpg), ("exr", exr), ("piz.exr", piz)):
        for it in range(120):
            b = bytearray(blob)
            mode = it % 3
            if mode == 0:
                for _ in range(1 + it % 6):
                    b[int(rng.integers(len(b)))] = int(rng.integers(256))
            elif mode == 1:
                b = b[: int(rng.integers(1, len(b)))]
            else:
                for _ in range(1 + it % 4):
                    b[int(rng.integers(min(len(b), 600)))] = int(rng.integers(256))
            p = tmp_path / f"m.{kind}"
            p.write_bytes(bytes(b))
            try:
                a = G.imread(str(p), 3 if it % 2 else 1)
                assert a.ndim == 3 and a.size > 0
                outcomes["ok"] += 1
            except G.GdptError:
                outcomes["rejected"] += 1
    assert outcomes["ok"] > 0 and outcomes["rejected"] > 0


def test_gdtex_companion_rejects_lying_headers(G, tmp_path):
    """A .gdtex companion whose header promises more texels (or a longer zlib body) than the file holds must be
    refused before anything is allocated from the header; a well-formed one decodes (v1 fp32 and v2 zlib/8-bit)."""
    import struct
    import zlib
    tex = (np.arange(4 * 3 * 3) % 251).astype(np.uint8).reshape(3, 4, 3)
    z = zlib.compress(tex.tobytes(), 9)
    good2 = b"GDTEX2\n" + struct.pack("<iiiI", 4, 3, 3, len(z)) + z
    good1 = b"GDTEX1\n" + struct.pack("<iii", 4, 3, 3) + tex.astype(np.float32).tobytes()
    for blob in (good1, good2):
        p = tmp_path / "t.tga"                      # a format without a native decoder -> the loader looks for the companion
        (tmp_path / "t.tga.gdtex").write_bytes(blob)
        a = G.imread(str(p), 3)
        assert a.shape == (3, 4, 3)
    expect = np.float32(tex / np.float32(255.0)) ** np.float32(2.2)
    assert np.abs(a - expect).max() < 1e-6         # v2 widened like stbi_loadf
    bad = [
        b"GDTEX1\n" + struct.pack("<iii", 60000, 60000, 3) + b"\0" * 64,            # 43 GB promised, 64 bytes there
        b"GDTEX2\n" + struct.pack("<iiiI", 60000, 60000, 3, 16) + b"\0" * 16,       # ditto for the inflated size
        b"GDTEX2\n" + struct.pack("<iiiI", 4, 3, 3, 0xFFFFFFF0) + z,                # 4 GB zlib body promised
        b"GDTEX2\n" + struct.pack("<iiiI", 70000, 3, 3, len(z)) + z,                # extent cap
        good2[:-3],                                                                 # truncated body
        good1[:-5],
    ]
    for blob in bad:
        (tmp_path / "t.tga.gdtex").write_bytes(blob)
        with pytest.raises(G.GdptError):
            G.imread(str(tmp_path / "t.tga"), 3)


def test_png_and_progressive_jpeg_decoders_match_the_reference_decoder(G):
    """tests/golden/images (made by tests/golden/make_images.py: PNGs of every colour type, bit depth 1..16, all five scanline
    filters, Adam7, palettes, tRNS; progressive JPEGs at 4:2:0 / 4:2:2 / 4:4:4 / grey) through the build's own decoders ==
    what the reference's imread3 / imread1 (stb_image v2.27 as vendored there) return: CRC-32 of the fp32 texels,
    generated by oracle/ref_img.cpp from the reference's own src/image.cpp."""
    import os
    import zlib
    here = os.path.dirname(os.path.abspath(__file__))
    gold = _golden()
    assert len(gold["fixtures_imread3"]) >= 30
    kinds = set()
    for name, rec in gold["fixtures_imread3"].items():
        path = os.path.join(here, "golden", "images", name)
        a = G.imread(path, 3)
        assert a.shape == (rec["height"], rec["width"], 3), name
        assert zlib.crc32(a.astype(np.float32).tobytes()) == rec["crc32"], name
        a1 = G.imread(path, 1)
        assert zlib.crc32(a1.astype(np.float32).tobytes()) == gold["fixtures_imread1"][name]["crc32"], name
        kinds.add(name.split(".")[-1])
    assert kinds == {"png", "jpg"}


def test_corrupt_png_and_jpeg_are_errors_not_crashes(G, tmp_path):
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    rng = np.random.default_rng(3)
    for name in ("rgba16_adam7.png", "palette4_adam7.png", "grey2.png", "prog_420.jpg", "prog_big_q35.jpg"):
        raw = bytearray(open(os.path.join(here, "golden", "images", name), "rb").read())
        for trial in range(60):
            b = bytearray(raw)
            for _ in range(1 + trial % 4):
                b[int(rng.integers(8, len(b)))] = int(rng.integers(0, 256))
            if trial % 7 == 0:
                b = b[: int(rng.integers(10, len(b)))]
            p = tmp_path / ("m_" + name)
            p.write_bytes(bytes(b))
            try:
                a = G.imread(str(p), 3)
                assert a.ndim == 3 and a.shape[2] == 3
            except G.GdptError:
                pass
