"""ctypes mirrors of the structs in include/gdpt.h (kept field-for-field in sync with the header)."""
import ctypes as C

GDPT_MAT_MAX_TEX = 12

TEX_CONSTANT, TEX_IMAGE, TEX_CHECKERBOARD = 0, 1, 2
(MAT_LAMBERTIAN, MAT_ROUGHPLASTIC, MAT_ROUGHDIELECTRIC, MAT_DISNEY_DIFFUSE, MAT_DISNEY_METAL,
 MAT_DISNEY_GLASS, MAT_DISNEY_CLEARCOAT, MAT_DISNEY_SHEEN, MAT_DISNEY_BSDF) = range(9)
SHAPE_SPHERE, SHAPE_TRIMESH = 0, 1
FILTER_BOX, FILTER_TENT, FILTER_GAUSSIAN = 0, 1, 2
INTEGRATOR_PATH, INTEGRATOR_GRADPATH, INTEGRATOR_OTHER = 5, 7, -1
RNG_TILE, RNG_SAMPLE = 0, 2
SHIFT_REFERENCE, SHIFT_RECONNECT = 0, 1
SOLVER_CG, SOLVER_DCT, SOLVER_DCT_MFMA = 0, 1, 2
SOLVER_DEFAULT = SOLVER_DCT_MFMA      # GDPT_SOLVER_DEFAULT (include/gdpt.h)


class GdptTexture(C.Structure):
    _fields_ = [("type", C.c_int32), ("image_id", C.c_int32),
                ("v0", C.c_double * 3), ("v1", C.c_double * 3),
                ("uscale", C.c_double), ("vscale", C.c_double), ("uoffset", C.c_double), ("voffset", C.c_double)]


class GdptMaterial(C.Structure):
    _fields_ = [("type", C.c_int32), ("_pad", C.c_int32), ("eta", C.c_double),
                ("tex", GdptTexture * GDPT_MAT_MAX_TEX)]


class GdptImage(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("channels", C.c_int32), ("_pad", C.c_int32),
                ("texels", C.POINTER(C.c_double))]


class GdptShape(C.Structure):
    _fields_ = [("type", C.c_int32), ("material_id", C.c_int32), ("area_light_id", C.c_int32),
                ("num_vertices", C.c_int32), ("num_triangles", C.c_int32), ("_pad", C.c_int32),
                ("center", C.c_double * 3), ("radius", C.c_double),
                ("positions", C.POINTER(C.c_double)), ("indices", C.POINTER(C.c_int32)),
                ("normals", C.POINTER(C.c_double)), ("uvs", C.POINTER(C.c_double))]


class GdptLight(C.Structure):
    _fields_ = [("shape_id", C.c_int32), ("_pad", C.c_int32), ("intensity", C.c_double * 3)]


class GdptCamera(C.Structure):
    _fields_ = [("sample_to_cam", C.c_double * 16), ("cam_to_world", C.c_double * 16),
                ("width", C.c_int32), ("height", C.c_int32), ("filter_type", C.c_int32), ("_pad", C.c_int32),
                ("filter_param", C.c_double)]


class GdptEnvmap(C.Structure):
    _fields_ = [("light_id", C.c_int32), ("image_id", C.c_int32), ("scale", C.c_double),
                ("to_world", C.c_double * 16), ("to_local", C.c_double * 16)]


class GdptSceneDesc(C.Structure):
    _fields_ = [("camera", GdptCamera),
                ("integrator", C.c_int32), ("samples_per_pixel", C.c_int32),
                ("max_depth", C.c_int32), ("rr_depth", C.c_int32),
                ("num_materials", C.c_int32), ("num_shapes", C.c_int32),
                ("num_lights", C.c_int32), ("num_images", C.c_int32),
                ("materials", C.POINTER(GdptMaterial)), ("shapes", C.POINTER(GdptShape)),
                ("lights", C.POINTER(GdptLight)), ("images", C.POINTER(GdptImage)),
                ("output_filename", C.c_char * 256), ("has_envmap", C.c_int32), ("_pad", C.c_int32),
                ("envmap", GdptEnvmap)]


class GdptRenderParams(C.Structure):
    _fields_ = [("spp", C.c_int32), ("rng_scheme", C.c_int32), ("row_begin", C.c_int32), ("row_end", C.c_int32),
                ("max_depth_override", C.c_int32), ("shift_mode", C.c_int32), ("plan_rows", C.c_int32), ("reserved", C.c_int32)]


class GdptRenderStats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays", C.c_uint64), ("bounces", C.c_uint64),
                ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64), ("nonfinite_samples", C.c_uint64),
                ("render_ms", C.c_double), ("node_bytes", C.c_uint64),
                ("wave_node_trips", C.c_uint64), ("wave_leaf_trips", C.c_uint64), ("wave_steps", C.c_uint64), ("lane_steps", C.c_uint64)]


class GdptPoissonStats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("solver", C.c_int32), ("rel_residual", C.c_double), ("solve_ms", C.c_double)]


GDPT_MULTI_MAX_DEVICES = 16
EXCHANGE_RCCL, EXCHANGE_PEER_COPY = 0, 1


class GdptMultiConfig(C.Structure):
    _fields_ = [("num_devices", C.c_int32), ("exchange", C.c_int32), ("devices", C.c_int32 * GDPT_MULTI_MAX_DEVICES),
                ("balance", C.c_int32), ("reserved", C.c_int32)]


class GdptMultiStats(C.Structure):
    _fields_ = [("num_devices", C.c_int32), ("exchange", C.c_int32),
                ("row_begin", C.c_int32 * GDPT_MULTI_MAX_DEVICES), ("row_end", C.c_int32 * GDPT_MULTI_MAX_DEVICES),
                ("render_ms", C.c_double * GDPT_MULTI_MAX_DEVICES), ("render_ms_max", C.c_double),
                ("exchange_ms", C.c_double), ("solve_ms", C.c_double), ("wall_ms", C.c_double)]
