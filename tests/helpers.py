"""Shared helpers for the test-suite (scene variants, error metrics)."""
import os
import re
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
ORACLE_DIR = os.path.join(ROOT, "oracle")
if ORACLE_DIR not in sys.path:
    sys.path.insert(0, ORACLE_DIR)

SCENES = os.path.join(ROOT, "scenes")


def scene_variant(tmp_dir, scene_rel, width=None, height=None, integrator=None, max_depth=None, spp=None):
    """Copies the scene's folder tree (XML + meshes; sibling folders it references) to tmp_dir and edits
    film size / integrator type / maxDepth / sampleCount. Returns the path of the edited XML."""
    src_dir = os.path.dirname(os.path.join(SCENES, scene_rel))
    dst_root = os.path.join(str(tmp_dir), "scenes")
    if not os.path.exists(dst_root):
        shutil.copytree(SCENES, dst_root)
    dst = os.path.join(dst_root, scene_rel)
    text = open(dst).read()
    if width is not None:
        text = re.sub(r'(<integer name="width" value=")\d+(")', r"\g<1>%d\2" % width, text)
    if height is not None:
        text = re.sub(r'(<integer name="height" value=")\d+(")', r"\g<1>%d\2" % height, text)
    if integrator is not None:
        text = re.sub(r'(<integrator type=")\w+(")', r"\g<1>%s\2" % integrator, text)
    if max_depth is not None:
        text = re.sub(r'(<integer name="maxDepth" value=")-?\d+(")', r"\g<1>%d\2" % max_depth, text)
    if spp is not None:
        text = re.sub(r'(<integer name="sampleCount" value=")\d+(")', r"\g<1>%d\2" % spp, text)
    out = dst.replace(".xml", "_variant.xml")
    with open(out, "w") as f:
        f.write(text)
    return out


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = np.linalg.norm((a - b).ravel())
    n = np.linalg.norm(b.ravel())
    return d / n if n > 0 else d
