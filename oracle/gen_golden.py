#!/usr/bin/env python3
"""Generates tests/golden/ref_kat.json from the reference's own code (development container only).

    make -C oracle ref && python oracle/gen_golden.py

Runs oracle/_ref/ref_kat (built from /root/reference sources by oracle/Makefile) and stores its JSON.
The spectra it integrates are read here from the scene XML and rounded through fp32 exactly as
parse_spectrum's std::stof does (src/parsers/parse_scene.cpp:159-176).
"""
import json
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("REF", "/root/reference")


def main():
    exe = os.path.join(HERE, "_ref", "ref_kat")
    if not os.path.exists(exe):
        sys.exit("oracle/_ref/ref_kat is missing: run `make -C oracle ref` in the container that has /root/reference")
    xml = open(os.path.join(ROOT, "scenes", "cbox", "cbox_gdpt.xml")).read()
    spectra = re.findall(r'<spectrum name="\w+" value="([^"]+)"', xml)
    lines = []
    for s in spectra:
        pairs = [p for p in re.split(r"[, ]+", s.strip()) if p]
        if len(pairs) < 2:
            continue
        vals = []
        for p in pairs:
            w, v = p.split(":")
            vals += [repr(float(np.float32(w))), repr(float(np.float32(v)))]
        lines.append(" ".join(vals))
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        f.write("\n".join(lines) + "\n")
        spec_path = f.name
    out = subprocess.check_output([exe, os.path.join(REF, "scenes", "cbox"), spec_path])
    os.unlink(spec_path)
    # printf spells non-finite doubles inf/nan; JSON (python dialect) wants Infinity/NaN
    text = re.sub(r"(?<=[:\[,])-?nan\b", "NaN", out.decode())
    text = re.sub(r"(?<=[:\[,])(-?)inf\b", r"\1Infinity", text)
    doc = json.loads(text)
    doc["spectra_source"] = "scenes/cbox/cbox_gdpt.xml <spectrum> values in document order (those with >1 entry)"
    dst = os.path.join(ROOT, "tests", "golden", "ref_kat.json")
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    with open(dst, "w") as f:
        json.dump(doc, f, separators=(",", ":"))
    print("wrote", dst, os.path.getsize(dst), "bytes;", {k: (len(v) if isinstance(v, list) else "obj") for k, v in doc.items() if k != "generator"})


def test_picture(w, h):
    """The deterministic picture of oracle/ref_img.cpp::test_pixel, same operations in the same order (fp64)."""
    x = np.arange(w, dtype=np.float64)[None, :]
    y = np.arange(h, dtype=np.float64)[:, None]
    u, v = (x + 0.5) / w, (y + 0.5) / h
    img = np.zeros((h, w, 3))
    img[..., 0] = u * u * 3.0 - 0.25
    img[..., 1] = (v - 0.5) * 1e-5 + 0 * u
    img[..., 2] = 100.0 * u * v + 1.0 / 1024.0
    img[1, 1, 2] += 70000.0
    return img


def gen_images():
    """tests/golden/ref_images.json: what the reference's own image.cpp (stb_image / tinyexr as vendored there) reads
    from the scene textures and writes for a test picture, plus the reference reader's verdict on files written by
    this build's writer."""
    import base64
    import glob
    import zlib
    exe = os.path.join(HERE, "_ref", "ref_img")
    if not os.path.exists(exe):
        sys.exit("oracle/_ref/ref_img is missing: run `make -C oracle ref`")
    sys.path.insert(0, ROOT)
    import gdpt_amd as G
    doc = {"generator": "oracle/ref_img.cpp linked against the reference's src/image.cpp (see oracle/Makefile)"}
    jpgs = sorted(glob.glob(os.path.join(ROOT, "scenes", "sponza", "textures", "*.JPG")))
    doc["imread3"] = json.loads(subprocess.check_output([exe, "hash3"] + jpgs))
    doc["imread1"] = json.loads(subprocess.check_output([exe, "hash1"] + jpgs))
    # PNG (every colour type / depth / filter / Adam7) and progressive JPEG fixtures written by tests/golden/make_images.py
    fx = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "images", "*.png")) + glob.glob(os.path.join(ROOT, "tests", "golden", "images", "*.jpg")))
    doc["fixtures_imread3"] = json.loads(subprocess.check_output([exe, "hash3"] + fx))
    doc["fixtures_imread1"] = json.loads(subprocess.check_output([exe, "hash1"] + fx))
    tmp = tempfile.mkdtemp()
    doc["exr"] = {}
    for (w, h) in ((19, 7), (8, 5), (40, 33)):
        pic = test_picture(w, h)
        ref_path, own_path = os.path.join(tmp, f"ref_{w}x{h}.exr"), os.path.join(tmp, f"own_{w}x{h}.exr")
        subprocess.check_call([exe, "writeexr", ref_path, str(w), str(h)])
        G.imwrite(own_path, pic)
        ref_seen = json.loads(subprocess.check_output([exe, "readexr", ref_path]))
        own_seen = json.loads(subprocess.check_output([exe, "readexr", own_path]))
        assert ref_seen == own_seen, (ref_seen, own_seen)      # the reference reader sees the same pixels in both files
        doc["exr"][f"{w}x{h}"] = {
            "width": w, "height": h,
            "reference_file_b64": base64.b64encode(open(ref_path, "rb").read()).decode(),   # written by the reference's imwrite
            "pixels_crc32_as_read_by_reference": ref_seen["crc32"],
            "own_file_crc32": zlib.crc32(open(own_path, "rb").read()),   # this build's file, validated above by the reference reader
        }
    # an EXR the reference's scenes use as input: PIZ-compressed environment map (scenes/matpreview/envmap.exr)
    env = os.path.join(ROOT, "scenes", "matpreview", "envmap.exr")
    seen = json.loads(subprocess.check_output([exe, "readexr", env]))
    own = G.imread(env, 3)
    assert zlib.crc32(own.astype(np.float32).tobytes()) == seen["crc32"]
    doc["exr_inputs"] = {"matpreview/envmap.exr": {"width": seen["width"], "height": seen["height"], "pixels_crc32": seen["crc32"],
                                                  "mean": own.mean(axis=(0, 1)).tolist()}}
    # the reference's own renders (fp16 + ZIP EXR written by lajolla): coarse statistics for end-to-end sanity checks
    doc["reference_renders"] = {}
    for rel in ("cbox_gdpt/cb_16.exr", "cbox_gdpt/cb_4.exr", "cbox_gdpt/cb_1.exr", "gdpt_renders/tmp_gdpt_0.04.exr",
                "cbox_path/cb_1000.exr", "cbox_path/cb_256.exr", "cbox_path/cb_16.exr",
                "extra_images/disney_glass_eta_1.5.exr", "extra_images/disney_sheen_test_1.0.exr",
                "gdpt_renders/sponza_grad_path_trace/s_gp_256.exr", "gdpt_renders/sponza_grad_path_trace/s_gp_16.exr",
                "gdpt_renders/sponza_regular_path_trace/sp_256.exr", "gdpt_renders/sponza_reg_path_non_nee/sp_256.exr",
                "gdpt_renders/sponza.exr"):
        path = os.path.join(REF, rel)
        if not os.path.exists(path):
            continue
        seen = json.loads(subprocess.check_output([exe, "readexr", path]))
        img = G.imread(path, 3)
        assert zlib.crc32(img.astype(np.float32).tobytes()) == seen["crc32"], rel    # own reader == reference reader
        h, w, _ = img.shape
        bs = 32
        thumb = img[: h // bs * bs, : w // bs * bs].reshape(h // bs, bs, w // bs, bs, 3).mean(axis=(1, 3))
        doc["reference_renders"][rel] = {
            "width": w, "height": h, "pixels_crc32": seen["crc32"],
            "mean": img.mean(axis=(0, 1)).tolist(), "negative_fraction": float((img < 0).any(axis=2).mean()),
            "percentiles_1_50_99": np.percentile(img.mean(axis=2), [1, 50, 99]).tolist(),
            "block_mean_32": np.round(thumb, 6).tolist(),
        }
    dst = os.path.join(ROOT, "tests", "golden", "ref_images.json")
    with open(dst, "w") as f:
        json.dump(doc, f, separators=(",", ":"))
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "images":
        gen_images()
    else:
        main()
