"""ctypes binding of libpathtrace_hip.so (include/pt_hip.h), named after the reference's interface.

``Scene``            Scene::Scene / Scene::getIntersection        (reference include/PathTrace/scene/scene.h:32,41)
``process_item``     processItem(WorkItem, RandomEngine&)          (include/PathTrace/worker.h:69)
``process_job``      processJob(FrameRenderJob)                    (include/PathTrace/worker.h:83-84)

The library is the only implementation behind these calls: if it is missing or no HIP device is usable they raise.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build
from .scenes import MATERIAL_DTYPE

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PT_LIB_OVERRIDE") or os.path.join(HERE, "libpathtrace_hip.so")  # override: A/B builds of the same ABI

PT_OK = 0
ERRORS = {1: "PT_ERR_INVALID", 2: "PT_ERR_NO_DEVICE", 3: "PT_ERR_HIP", 4: "PT_ERR_UNSUPPORTED", 5: "PT_ERR_NOMEM"}

EXPORTS = ["pt_device_count", "pt_last_error", "pt_scene_create", "pt_scene_destroy", "pt_scene_info", "pt_scene_emissive", "pt_scene_bvh_dump", "pt_intersect_batch",
           "pt_render_streams", "pt_render_item", "pt_render_tiles", "pt_render_tiles_progress", "pt_render_tiles_multi", "pt_render_tiles_device", "pt_job_tiles", "pt_pixel_seed", "pt_rng_seed_to_state", "pt_post_process", "pt_post_process_device"]


class PtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("%s: %s" % (ERRORS.get(code, code), message))
        self.code = code


POST_TONE_MAP, POST_GAMMA = 1, 2


def post_process(image, steps=POST_TONE_MAP | POST_GAMMA, gamma=1.8, device=0):
    """toneMap / gammaCorrect / postProcess (post_processing.h:14,22,30) of an (h, w, 4) float32 frame on the GPU; returns a new array."""
    img = np.array(image, dtype=np.float32, order="C", copy=True)
    h, w = img.shape[:2]
    _check(load().pt_post_process(C.c_int(device), _ptr(img), C.c_int32(w), C.c_int32(h), C.c_uint32(steps), C.c_float(gamma)))
    return img


class SceneDesc(C.Structure):
    _fields_ = [("n_objects", C.c_uint32), ("obj_kind", C.c_void_p),
                ("n_triangles", C.c_uint32), ("tri_pos", C.c_void_p), ("tri_nrm", C.c_void_p), ("tri_cull", C.c_void_p),
                ("tri_material", C.c_void_p),
                ("n_spheres", C.c_uint32), ("sph", C.c_void_p), ("sph_material", C.c_void_p),
                ("n_materials", C.c_uint32), ("materials", C.c_void_p),
                ("n_point_lights", C.c_uint32), ("light_pos", C.c_void_p), ("light_spectrum", C.c_void_p)]


class CameraParams(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("look_at", C.c_float * 3), ("up", C.c_float * 3), ("focal_length", C.c_float),
                ("height", C.c_float), ("aspect_ratio", C.c_float), ("aperture_width", C.c_float), ("aperture_height", C.c_float),
                ("aperture_kind", C.c_int32), ("hex_ratio", C.c_float), ("focal_plane_dist", C.c_float)]


class Options(C.Structure):
    _fields_ = [("image_width", C.c_int32), ("image_height", C.c_int32), ("min_sample_count", C.c_int32),
                ("max_sample_count", C.c_int32), ("epsilon", C.c_float)]


class Stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays_traced", C.c_uint64), ("shadow_rays_traced", C.c_uint64), ("node_visits", C.c_uint64),
                ("leaf_tests", C.c_uint64), ("vertices", C.c_uint64), ("launches", C.c_uint64), ("kernel_ms", C.c_double),
                ("wave_steps", C.c_uint64), ("shading_passes", C.c_uint64), ("wavefronts", C.c_uint64), ("slot_rows", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


TILE_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("w", "<i4"), ("h", "<i4")])
STREAM_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("w", "<i4"), ("h", "<i4"), ("rng_state", "<u8")])

_lib = None


def load(build_if_missing=True):
    """Load libpathtrace_hip.so; raises if it cannot be built/loaded (there is no fallback implementation)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        if not build_if_missing:
            raise FileNotFoundError(LIB_PATH + " is missing: run python -c 'import __graft_entry__ as g; g.build()'")
        _build.build()
    lib = C.CDLL(LIB_PATH)
    lib.pt_last_error.restype = C.c_char_p
    lib.pt_job_tiles.restype = C.c_size_t
    lib.pt_pixel_seed.restype = C.c_uint64
    lib.pt_rng_seed_to_state.restype = C.c_uint64
    lib.pt_scene_destroy.restype = None
    _lib = lib
    return lib


def _check(rc):
    if rc != PT_OK:
        raise PtError(rc, load().pt_last_error().decode())


def _ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def device_count():
    return load().pt_device_count()


def seed_to_state(seed):
    return load().pt_rng_seed_to_state(C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF))


def pixel_seed(base_seed, x, y):
    return load().pt_pixel_seed(C.c_uint64(base_seed), C.c_int32(x), C.c_int32(y))


def job_tiles(width, height):
    """The WorkItem list processJob builds (reference src/worker.cpp:398-414)."""
    lib = load()
    n = lib.pt_job_tiles(C.c_int32(width), C.c_int32(height), None, C.c_size_t(0))
    tiles = np.zeros(n, dtype=TILE_DTYPE)
    lib.pt_job_tiles(C.c_int32(width), C.c_int32(height), _ptr(tiles), C.c_size_t(n))
    return tiles


def _camera(cam):
    p = CameraParams()
    p.origin[:] = [float(v) for v in cam["origin"]]
    p.look_at[:] = [float(v) for v in cam["look_at"]]
    p.up[:] = [float(v) for v in cam["up"]]
    p.focal_length, p.height, p.aspect_ratio = cam["focal_length"], cam["height"], cam["aspect_ratio"]
    p.aperture_width, p.aperture_height = cam.get("aperture_width", 0.0), cam.get("aperture_height", 0.0)
    p.aperture_kind, p.hex_ratio = cam.get("aperture_kind", 0), cam.get("hex_ratio", 0.0)
    p.focal_plane_dist = cam.get("focal_plane_dist", 0.0)
    return p


def _options(opt):
    return Options(int(opt["image_width"]), int(opt["image_height"]), int(opt["min_sample_count"]), int(opt["max_sample_count"]),
                   float(opt["epsilon"]))


class Scene:
    """A scene resident on one MI355X (the reference's Scene: objects + lights + BVH)."""

    def __init__(self, scene, device=0):
        lib = load()
        keep = {
            "obj_kind": np.ascontiguousarray(scene["obj_kind"], dtype=np.uint8),
            "tri_pos": np.ascontiguousarray(scene["tri_pos"], dtype=np.float32).reshape(-1, 9),
            "tri_nrm": None if scene.get("tri_nrm") is None else np.ascontiguousarray(scene["tri_nrm"], dtype=np.float32).reshape(-1, 9),
            "tri_cull": np.ascontiguousarray(scene["tri_cull"], dtype=np.uint8),
            "tri_material": np.ascontiguousarray(scene["tri_material"], dtype=np.uint32),
            "sph": np.ascontiguousarray(scene["sph"], dtype=np.float32).reshape(-1, 4),
            "sph_material": np.ascontiguousarray(scene["sph_material"], dtype=np.uint32),
            "materials": np.ascontiguousarray(scene["materials"], dtype=MATERIAL_DTYPE),
            "light_pos": np.ascontiguousarray(scene["light_pos"], dtype=np.float32).reshape(-1, 3),
            "light_spectrum": np.ascontiguousarray(scene["light_spectrum"], dtype=np.float32).reshape(-1, 4),
        }
        d = SceneDesc()
        d.n_objects, d.obj_kind = len(keep["obj_kind"]), _ptr(keep["obj_kind"])
        d.n_triangles, d.tri_pos, d.tri_nrm = len(keep["tri_pos"]), _ptr(keep["tri_pos"]), _ptr(keep["tri_nrm"])
        d.tri_cull, d.tri_material = _ptr(keep["tri_cull"]), _ptr(keep["tri_material"])
        d.n_spheres, d.sph, d.sph_material = len(keep["sph"]), _ptr(keep["sph"]), _ptr(keep["sph_material"])
        d.n_materials, d.materials = len(keep["materials"]), _ptr(keep["materials"])
        d.n_point_lights, d.light_pos, d.light_spectrum = len(keep["light_pos"]), _ptr(keep["light_pos"]), _ptr(keep["light_spectrum"])
        self.n_objects = d.n_objects
        self._h = C.c_void_p()
        _check(lib.pt_scene_create(C.c_int(device), C.byref(d), C.byref(self._h)))
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            load().pt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        n, depth, ne = C.c_uint64(), C.c_uint32(), C.c_uint32()
        _check(load().pt_scene_info(self._h, C.byref(n), C.byref(depth), C.byref(ne)))
        return {"n_nodes": n.value, "depth": depth.value, "n_emissive": ne.value}

    def emissive(self):
        """Emissive objects in Scene::registerEmissiveObjects order (scene.cpp:183-208) and their normalised CDF."""
        n = self.info()["n_emissive"]
        obj, cdf = np.empty(max(n, 1), np.int32), np.empty(max(n, 1), np.float32)
        written = C.c_uint64()
        _check(load().pt_scene_emissive(self._h, _ptr(obj), _ptr(cdf), C.c_uint64(n), C.byref(written)))
        return obj[:n], cdf[:n]

    def bvh_dump(self):
        n = max(2 * self.n_objects - 1, 1)
        obj, box = np.empty(n, np.int32), np.empty((n, 6), np.float32)
        written = C.c_uint64()
        _check(load().pt_scene_bvh_dump(self._h, _ptr(obj), _ptr(box), C.c_uint64(n), C.byref(written)))
        return obj[:written.value], box[:written.value]

    def get_intersection(self, rays):
        """Scene::getIntersection for a batch: rays (n, 6) -> (t, object index or -1)."""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        t, obj = np.empty(len(rays), np.float32), np.empty(len(rays), np.int32)
        _check(load().pt_intersect_batch(self._h, _ptr(rays), C.c_size_t(len(rays)), _ptr(t), _ptr(obj)))
        return t, obj

    def process_item(self, camera, options, streams, image=None, want_stats=False):
        """processItem for many WorkItems at once; streams: array of STREAM_DTYPE (rect + raw engine state)."""
        streams = np.ascontiguousarray(streams, dtype=STREAM_DTYPE)
        if image is None:
            image = np.zeros((options["image_height"], options["image_width"], 4), np.float32)
        states = np.empty(len(streams), np.uint64)
        cp, op, st = _camera(camera), _options(options), Stats()
        _check(load().pt_render_streams(self._h, C.byref(cp), C.byref(op), _ptr(streams), C.c_size_t(len(streams)), _ptr(image), _ptr(states),
                                        C.byref(st) if want_stats else None))
        return (image, states, st.as_dict()) if want_stats else (image, states)

    def process_job(self, camera, options, base_seed=1234, tiles=None, image=None, want_stats=False):
        """processJob: every pixel of the given tiles (default: all tiles of the image) with per-pixel engines."""
        if tiles is None:
            tiles = job_tiles(options["image_width"], options["image_height"])
        tiles = np.ascontiguousarray(tiles, dtype=TILE_DTYPE)
        if image is None:
            image = np.zeros((options["image_height"], options["image_width"], 4), np.float32)
        cp, op, st = _camera(camera), _options(options), Stats()
        _check(load().pt_render_tiles(self._h, C.byref(cp), C.byref(op), _ptr(tiles), C.c_size_t(len(tiles)), C.c_uint64(base_seed), _ptr(image),
                                      C.byref(st) if want_stats else None))
        return (image, st.as_dict()) if want_stats else image

    def process_work_item(self, camera, options, x, y, w, h, rng_state, want_stats=False):
        """processItem(WorkItem(job, x, y, w, h), engine) returning the item's own (h, w, 4) tile and the engine state afterwards."""
        item = np.zeros(1, dtype=STREAM_DTYPE)
        item["x"], item["y"], item["w"], item["h"], item["rng_state"] = x, y, w, h, rng_state
        tile = np.zeros((max(h, 0), max(w, 0), 4), np.float32)
        state = C.c_uint64()
        cp, op, st = _camera(camera), _options(options), Stats()
        _check(load().pt_render_item(self._h, C.byref(cp), C.byref(op), _ptr(item), _ptr(tile) if tile.size else None, C.byref(state),
                                     C.byref(st) if want_stats else None))
        return (tile, state.value, st.as_dict()) if want_stats else (tile, state.value)

    def process_job_progress(self, camera, options, progress, base_seed=1234, tiles=None):
        """processJob with its progress callback (worker.h:75-84): progress(completed, total) from the calling thread while the device renders."""
        if tiles is None:
            tiles = job_tiles(options["image_width"], options["image_height"])
        tiles = np.ascontiguousarray(tiles, dtype=TILE_DTYPE)
        image = np.zeros((options["image_height"], options["image_width"], 4), np.float32)
        cb = PROGRESS_FN(lambda done, total, user: progress(done, total))
        cp, op = _camera(camera), _options(options)
        _check(load().pt_render_tiles_progress(self._h, C.byref(cp), C.byref(op), _ptr(tiles), C.c_size_t(len(tiles)), C.c_uint64(base_seed), _ptr(image), None,
                                               cb, None))
        return image

    def process_job_device(self, camera, options, d_image_ptr, stream_ptr, base_seed=1234, tiles=None, want_stats=False):
        """processJob writing into device memory (d_image_ptr: device address of width*height*4 floats)."""
        if tiles is None:
            tiles = job_tiles(options["image_width"], options["image_height"])
        tiles = np.ascontiguousarray(tiles, dtype=TILE_DTYPE)
        cp, op, st = _camera(camera), _options(options), Stats()
        _check(load().pt_render_tiles_device(self._h, C.byref(cp), C.byref(op), _ptr(tiles), C.c_size_t(len(tiles)), C.c_uint64(base_seed),
                                             C.c_void_p(d_image_ptr), C.c_void_p(stream_ptr), C.byref(st) if want_stats else None))
        return st.as_dict() if want_stats else None


PROGRESS_FN = C.CFUNCTYPE(None, C.c_int, C.c_int, C.c_void_p)


def process_job_multi(scenes, camera, options, base_seed=1234, tiles=None, progress=None, want_stats=False):
    """processJob over several Scene replicas (one per device): tile k is rendered by scenes[k % len(scenes)]."""
    if tiles is None:
        tiles = job_tiles(options["image_width"], options["image_height"])
    tiles = np.ascontiguousarray(tiles, dtype=TILE_DTYPE)
    image = np.zeros((options["image_height"], options["image_width"], 4), np.float32)
    handles = (C.c_void_p * len(scenes))(*[s._h for s in scenes])
    stats = (Stats * len(scenes))()
    cb = PROGRESS_FN(lambda done, total, user: progress(done, total)) if progress is not None else None
    cp, op = _camera(camera), _options(options)
    _check(load().pt_render_tiles_multi(handles, C.c_int(len(scenes)), C.byref(cp), C.byref(op), _ptr(tiles), C.c_size_t(len(tiles)), C.c_uint64(base_seed),
                                        _ptr(image), stats if want_stats else None, cb, None))
    return (image, [s.as_dict() for s in stats]) if want_stats else image


def pixel_streams(xs, ys, states):
    s = np.zeros(len(xs), dtype=STREAM_DTYPE)
    s["x"], s["y"], s["w"], s["h"], s["rng_state"] = xs, ys, 1, 1, states
    return s
