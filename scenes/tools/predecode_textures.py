#!/usr/bin/env python3
"""Pre-decodes bitmap textures a scene XML references into `<file>.gdtex` companions, for formats the library does
not decode itself (PNG, TGA, BMP, progressive JPEG ...; baseline JPEG, PFM and scanline EXR are read natively — the
reference decodes everything through stb_image / tinyexr at load time, src/image.cpp:26-133).

Format GDTEX2: b"GDTEX2\\n", int32 width, height, channels(=3), uint32 zlen, zlib(uint8 RGB texels, row-major).
The loader widens 8-bit texels exactly as stbi_loadf does: (float)pow(v/255.0f, 2.2f) (src/3rdparty/stb_image.h:1849).
Decoding here uses PIL/libjpeg; stb_image's IDCT can differ from it by one 8-bit step on some texels.

    python scenes/tools/predecode_textures.py scenes/sponza/sponza.xml
"""
import os
import re
import struct
import sys
import zlib

import numpy as np
from PIL import Image


def main(xml_path):
    base = os.path.dirname(os.path.abspath(xml_path))
    text = open(xml_path).read()
    files = sorted(set(re.findall(r'<texture type="bitmap"[^>]*>\s*<string name="filename" value="([^"]+)"', text)))
    for rel in files:
        src = os.path.join(base, rel)
        if rel.lower().endswith((".pfm", ".exr")):
            print("skip (read natively):", rel)
            continue
        img = np.asarray(Image.open(src).convert("RGB"), dtype=np.uint8)
        h, w, _ = img.shape
        z = zlib.compress(img.tobytes(), 9)
        with open(src + ".gdtex", "wb") as f:
            f.write(b"GDTEX2\n" + struct.pack("<iiiI", w, h, 3, len(z)) + z)
        print(f"{rel}: {w}x{h} -> {len(z)} bytes")


if __name__ == "__main__":
    main(sys.argv[1])
