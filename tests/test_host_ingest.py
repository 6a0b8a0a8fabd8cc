"""Host-side scene ingest (the parse_scene drop-in) checked against the reference's own parse_obj /
compute_normal / integrate_XYZ / Camera outputs (tests/golden/ref_kat.json) and the reference's error behaviour."""
import os

import numpy as np
import pytest

from helpers import SCENES, scene_variant


def _mesh(shape):
    nv, nt = shape.num_vertices, shape.num_triangles
    pos = np.ctypeslib.as_array(shape.positions, shape=(nv * 3,)).copy()
    idx = np.ctypeslib.as_array(shape.indices, shape=(nt * 3,)).copy()
    nrm = np.ctypeslib.as_array(shape.normals, shape=(nv * 3,)).copy() if shape.normals else None
    return pos, idx, nrm


def test_cbox_scene_matches_reference_pieces(G, golden):
    sd = G.parse_scene(os.path.join(SCENES, "cbox", "cbox_gdpt.xml"))
    d = sd.desc
    assert (d.camera.width, d.camera.height) == (512, 512)
    assert d.integrator == G.INTEGRATOR_GRADPATH and d.max_depth == -1 and d.rr_depth == 5
    assert d.samples_per_pixel == 4            # <sampler sampleCount="4">, ignored by the reference's render loop
    assert d.camera.filter_type == G.FILTER_GAUSSIAN and d.camera.filter_param == 0.5
    assert d.output_filename == b"image.exr"
    assert d.num_shapes == 8 and d.num_materials == 5 and d.num_lights == 1
    # camera matrices: golden camera 0 is this scene's sensor built by the reference's Camera ctor
    cam = golden["cameras"][0]
    np.testing.assert_allclose(list(d.camera.cam_to_world), cam["cam_to_world"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(list(d.camera.sample_to_cam), cam["sample_to_cam"], rtol=1e-13, atol=1e-15)
    # meshes, in document order = golden "obj" order (luminaire first, translated by -0.5)
    total_tris = 0
    for i, ref in enumerate(golden["obj"]):
        if ref["variant"] == 2:
            continue      # the transformed large box is checked in test_obj_with_general_transform
        pos, idx, nrm = _mesh(d.shapes[i])
        assert np.array_equal(idx, np.array(ref["indices"], dtype=np.int32)), ref["file"]
        np.testing.assert_array_equal(pos, np.array(ref["positions"]))          # bit-exact
        np.testing.assert_allclose(nrm, np.array(ref["normals"]), rtol=0, atol=1e-15)
        total_tris += d.shapes[i].num_triangles
    assert total_tris + d.shapes[7].num_triangles == 38
    # reflectance spectra -> RGB, document order: box, white, red, green, light, then the emitter
    spectra = golden["spectra"]
    for m, s in zip(range(5), spectra[:5]):
        np.testing.assert_allclose(list(d.materials[m].tex[0].v0), s["rgb"], rtol=1e-14)
    np.testing.assert_allclose(list(d.lights[0].intensity), spectra[5]["rgb"], rtol=1e-14)
    assert d.lights[0].shape_id == 0 and d.shapes[0].area_light_id == 0
    assert [d.shapes[i].material_id for i in range(8)] == [4, 1, 1, 1, 3, 2, 0, 0]


def test_obj_with_general_transform(G, golden, tmp_path):
    ref = [o for o in golden["obj"] if o["variant"] == 2][0]
    m = np.array(ref["to_world"]).reshape(4, 4)
    xml = f"""<scene version="0.5.0"><integrator type="gradpath"/>
      <bsdf type="diffuse" id="a"/><shape type="obj"><string name="filename" value="{SCENES}/cbox/meshes/{ref['file']}"/>
      <transform name="toWorld"><matrix value="{' '.join(repr(float(v)) for v in m.ravel())}"/></transform><ref id="a"/></shape></scene>"""
    p = tmp_path / "t.xml"
    p.write_text(xml)
    sd = G.parse_scene(str(p))
    pos, idx, nrm = _mesh(sd.desc.shapes[0])
    # the matrix goes through std::stof in both parsers; the golden used the fp64 matrix, so compare at fp32 accuracy
    np.testing.assert_allclose(pos, np.array(ref["positions"]), rtol=2e-6, atol=1e-4)
    np.testing.assert_allclose(nrm, np.array(ref["normals"]), rtol=0, atol=2e-6)
    assert np.array_equal(idx, np.array(ref["indices"], dtype=np.int32))


def test_scene_defaults_and_quirks(G, tmp_path):
    # defaults of parse_scene(): path integrator, 4 spp, 256x256, box filter, image.exr
    p = tmp_path / "empty.xml"
    p.write_text('<?xml version="1.0"?><scene version="0.5.0"></scene>')
    sd = G.parse_scene(str(p))
    d = sd.desc
    assert (d.camera.width, d.camera.height, d.integrator, d.samples_per_pixel) == (256, 256, G.INTEGRATOR_PATH, 4)
    assert d.camera.filter_type == G.FILTER_BOX and d.output_filename == b"image.exr"
    # $defaults, stof truncation ("1e5+1" -> 1e5), single-valued spectrum reflectance -> (1,1,1), sphere ignores toWorld
    import numpy as np
    G.imwrite(str(tmp_path / "env.pfm"), np.full((2, 4, 3), 0.5))
    p = tmp_path / "q.xml"
    p.write_text("""<scene version="0.5.0"><default name="res" value="48"/><default name="s" value="7"/>
      <integrator type="gradpath"><integer name="maxDepth" value="3"/><integer name="rrDepth" value="2"/></integrator>
      <sensor type="perspective"><float name="fov" value="30"/>
        <sampler type="independent"><integer name="sampleCount" value="$s"/></sampler>
        <film type="hdrfilm"><integer name="width" value="$res"/><integer name="height" value="32"/>
          <string name="filename" value="o.pfm"/><rfilter type="tent"/></film></sensor>
      <shape type="sphere"><point name="center" x="1e5+1" y="2" z="3"/><float name="radius" value="0.5"/>
        <transform name="toWorld"><translate x="100"/></transform>
        <bsdf type="diffuse"><spectrum name="reflectance" value="0.3"/></bsdf>
        <emitter type="area"><rgb name="radiance" value="1 2 3"/></emitter></shape>
      <emitter type="envmap"><string name="filename" value="env.pfm"/><float name="scale" value="2.5"/></emitter>
    </scene>""")
    sd = G.parse_scene(str(p))
    d = sd.desc
    # the environment map is a light in parse order (src/parsers/parse_scene.cpp:1484-1508): placeholder + GdptEnvmap
    assert d.has_envmap == 1 and d.num_lights == 2 and d.lights[1].shape_id == -1
    assert d.envmap.light_id == 1 and d.envmap.scale == 2.5 and d.images[d.envmap.image_id].width == 4
    assert list(d.envmap.to_world) == list(d.envmap.to_local) == [1.0 if i % 5 == 0 else 0.0 for i in range(16)]
    missing = tmp_path / "m.xml"
    missing.write_text(p.read_text().replace("env.pfm", "nowhere.exr"))
    with pytest.raises(G.GdptError, match="Failure when loading image"):
        G.parse_scene(str(missing))
    assert (d.camera.width, d.camera.height, d.samples_per_pixel, d.max_depth, d.rr_depth) == (48, 32, 7, 3, 2)
    assert d.camera.filter_type == G.FILTER_TENT and d.camera.filter_param == 2.0 and d.output_filename == b"o.pfm"
    s = d.shapes[0]
    assert s.type == G.SHAPE_SPHERE and list(s.center) == [100000.0, 2.0, 3.0] and s.radius == 0.5
    assert list(d.materials[s.material_id].tex[0].v0) == [1.0, 1.0, 1.0]
    assert list(d.lights[0].intensity) == [1.0, 2.0, 3.0] and d.num_lights == 2   # area light + the envmap placeholder


@pytest.mark.parametrize("body,msg", [
    ('<integrator type="bogus"/>', "Unsupported integrator"),
    ('<bsdf type="phong" id="x"/>', "Unknown BSDF"),
    ('<shape type="cube"/>', "Unknown shape"),
    ('<shape type="obj"><string name="filename" value="missing.obj"/><bsdf type="diffuse"/></shape>', "Unable to open the obj file"),
    ('<shape type="sphere"><ref id="nope"/></shape>', "Material reference nope not found"),
    ('<sensor type="orthographic"/>', "Unsupported sensor"),
    ('<bsdf type="diffuse" id="a"><ref name="reflectance" id="t"/></bsdf>', "Texture not found"),
    ('<texture type="wood" id="t"/>', "Unknown texture type"),
])
def test_error_behaviour_mirrors_reference(G, tmp_path, body, msg):
    # the reference throws fl_exception with these messages (src/parsers/parse_scene.cpp); here: status + message
    p = tmp_path / "bad.xml"
    p.write_text(f'<scene version="0.5.0">{body}</scene>')
    with pytest.raises(G.GdptError) as e:
        G.parse_scene(str(p))
    assert msg in str(e.value)


def test_missing_file_and_malformed_xml(G, tmp_path):
    with pytest.raises(G.GdptError):
        G.parse_scene(str(tmp_path / "does_not_exist.xml"))
    p = tmp_path / "broken.xml"
    p.write_text("<scene><shape></scene>")
    with pytest.raises(G.GdptError):
        G.parse_scene(str(p))


def test_serialized_and_disney_scene_loads(G):
    sd = G.parse_scene(os.path.join(SCENES, "disney_bsdf_test", "disney_bsdf.xml"))
    d = sd.desc
    assert (d.camera.width, d.camera.height) == (683, 512) and d.integrator == G.INTEGRATOR_PATH
    assert d.num_shapes == 3
    tris = sum(d.shapes[i].num_triangles for i in range(3))
    assert tris == 61600          # matpreview geometry
    mats = [d.materials[d.shapes[i].material_id] for i in range(3)]
    assert mats[0].type == G.MAT_DISNEY_BSDF and mats[0].eta == 1.5
    assert abs(mats[0].tex[5].v0[0] - np.float32(0.1)) < 1e-12     # roughness through stof
    assert mats[2].type == G.MAT_LAMBERTIAN and mats[2].tex[0].type == G.TEX_CHECKERBOARD and mats[2].tex[0].uscale == 8.0
    for i in range(3):
        assert d.shapes[i].normals      # serialized normals or Nelson-Max ones
