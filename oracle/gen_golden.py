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


if __name__ == "__main__":
    main()
