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


# ---- building GdptSceneDesc objects from Python (tests only) -------------------------------------
import ctypes as _C


class DescBuilder:
    """Owns the ctypes arrays behind a GdptSceneDesc assembled in Python."""

    def __init__(self, G):
        self.G = G
        self.keep = []
        self.materials, self.shapes, self.lights, self.images = [], [], [], []
        self.desc = G.GdptSceneDesc()
        cam = self.desc.camera
        for i in range(16):
            cam.sample_to_cam[i] = 1.0 if i % 5 == 0 else 0.0
            cam.cam_to_world[i] = 1.0 if i % 5 == 0 else 0.0
        cam.width, cam.height, cam.filter_type, cam.filter_param = 16, 16, G.FILTER_BOX, 1.0
        self.desc.integrator, self.desc.samples_per_pixel = G.INTEGRATOR_GRADPATH, 4
        self.desc.max_depth, self.desc.rr_depth = -1, 5

    @staticmethod
    def const_tex(G, v):
        t = G.GdptTexture()
        t.type, t.image_id = G.TEX_CONSTANT, -1
        vals = [v, v, v] if not hasattr(v, "__len__") else list(v)
        for i in range(3):
            t.v0[i] = vals[i]
        t.uscale = t.vscale = 1.0
        return t

    def checker_tex(self, c0, c1, us, vs, uo, vo):
        t = self.G.GdptTexture()
        t.type, t.image_id = self.G.TEX_CHECKERBOARD, -1
        for i in range(3):
            t.v0[i], t.v1[i] = c0[i], c1[i]
        t.uscale, t.vscale, t.uoffset, t.voffset = us, vs, uo, vo
        return t

    def image_tex(self, texels, w, h, channels, us, vs, uo, vo):
        arr = (_C.c_double * len(texels))(*texels)
        self.keep.append(arr)
        im = self.G.GdptImage()
        im.width, im.height, im.channels = w, h, channels
        im.texels = _C.cast(arr, _C.POINTER(_C.c_double))
        self.images.append(im)
        t = self.G.GdptTexture()
        t.type, t.image_id = self.G.TEX_IMAGE, len(self.images) - 1
        t.uscale, t.vscale, t.uoffset, t.voffset = us, vs, uo, vo
        return t

    def material(self, mtype, texs, eta=1.5):
        m = self.G.GdptMaterial()
        m.type, m.eta = mtype, eta
        for i in range(12):
            m.tex[i] = texs[i] if i < len(texs) else self.const_tex(self.G, 0.0)
        self.materials.append(m)
        return len(self.materials) - 1

    def mesh(self, positions, indices, material_id, normals=None, uvs=None, light=None):
        G = self.G
        s = G.GdptShape()
        s.type, s.material_id, s.area_light_id = G.SHAPE_TRIMESH, material_id, -1
        s.num_vertices, s.num_triangles = len(positions) // 3, len(indices) // 3
        p = (_C.c_double * len(positions))(*positions)
        ix = (_C.c_int32 * len(indices))(*indices)
        self.keep += [p, ix]
        s.positions = _C.cast(p, _C.POINTER(_C.c_double))
        s.indices = _C.cast(ix, _C.POINTER(_C.c_int32))
        if normals is not None:
            n = (_C.c_double * len(normals))(*normals)
            self.keep.append(n)
            s.normals = _C.cast(n, _C.POINTER(_C.c_double))
        if uvs is not None:
            u = (_C.c_double * len(uvs))(*uvs)
            self.keep.append(u)
            s.uvs = _C.cast(u, _C.POINTER(_C.c_double))
        if light is not None:
            l = G.GdptLight()
            l.shape_id = len(self.shapes)
            for i in range(3):
                l.intensity[i] = light[i]
            s.area_light_id = len(self.lights)
            self.lights.append(l)
        self.shapes.append(s)
        return len(self.shapes) - 1

    def sphere(self, center, radius, material_id, light=None):
        G = self.G
        s = G.GdptShape()
        s.type, s.material_id, s.area_light_id = G.SHAPE_SPHERE, material_id, -1
        for i in range(3):
            s.center[i] = center[i]
        s.radius = radius
        if light is not None:
            l = G.GdptLight()
            l.shape_id = len(self.shapes)
            for i in range(3):
                l.intensity[i] = light[i]
            s.area_light_id = len(self.lights)
            self.lights.append(l)
        self.shapes.append(s)
        return len(self.shapes) - 1

    def finish(self):
        G, d = self.G, self.desc

        def arr(lst, typ):
            a = (typ * max(1, len(lst)))(*lst)
            self.keep.append(a)
            return a
        self._m = arr(self.materials, G.GdptMaterial)
        self._s = arr(self.shapes, G.GdptShape)
        self._l = arr(self.lights, G.GdptLight)
        self._i = arr(self.images, G.GdptImage)
        d.num_materials, d.num_shapes, d.num_lights, d.num_images = len(self.materials), len(self.shapes), len(self.lights), len(self.images)
        d.materials = _C.cast(self._m, _C.POINTER(G.GdptMaterial))
        d.shapes = _C.cast(self._s, _C.POINTER(G.GdptShape))
        d.lights = _C.cast(self._l, _C.POINTER(G.GdptLight))
        d.images = _C.cast(self._i, _C.POINTER(G.GdptImage))
        return _C.pointer(d)
