"""Parity of the HIP path (through the C ABI, include/pt_hip.h) with the golden vectors recorded from the compiled reference
and with the CPU oracle on the same seeded inputs.  Everything here needs a real MI355X: run with -m gpu."""
import numpy as np
import pytest

import oracle
from cpupathtrace_amd import binding, scenes
from tests.cases import golden, golden_mesh, opt_from, post_cases, scene_set
from tests.util import assert_bits_equal, miss_equal

pytestmark = pytest.mark.gpu

SCENES = ["box", "cornell", "advanced", "simple", "meshbox", "cornellmesh"]


@pytest.fixture(scope="module")
def sset():
    return scene_set(golden_mesh())


@pytest.fixture(scope="module")
def gpu_scenes(sset):
    assert binding.device_count() > 0, "no HIP device: the HIP path cannot run (there is no fallback)"
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = binding.Scene(sset[name][0])
        return cache[name]

    yield get
    for s in cache.values():
        s.close()


@pytest.mark.parametrize("name", SCENES)
def test_bvh_topology(gpu_scenes, name):
    g = golden("scene_" + name)
    obj, box = gpu_scenes(name).bvh_dump()
    assert_bits_equal(obj, g["bvh_obj"], "bvh topology")
    assert_bits_equal(box, g["bvh_box"], "bvh boxes")


@pytest.mark.parametrize("name", SCENES)
def test_closest_hit(gpu_scenes, name):
    """Scene::getIntersection: hit distance bit-exact, same object (reference test/scene/scene_test.cpp pins object identity)."""
    g = golden("scene_" + name)
    t, obj = gpu_scenes(name).get_intersection(g["rays"])
    miss_equal(t, g["t"], "closest hit t")
    hit = g["t"] >= 0
    assert_bits_equal(obj[hit], g["obj"][hit], "hit object")
    assert (obj[~hit] == -1).all()


def test_scene_test_cases():
    """test/scene/scene_test.cpp:8-47."""
    sc = binding.Scene(scenes.two_spheres_scene())
    rays = np.array([[-0.5, -0.5, -5, 0, 0, 1], [0.5, 0.5, -5, 0, 0, 1], [0, 0, 0, 0, 0, 1]], np.float32)
    t, obj = sc.get_intersection(rays)
    assert t[0] >= 0 and obj[0] == 0
    assert t[1] >= 0 and obj[1] == 1
    assert t[2] < 0 and obj[2] == -1
    sc.close()


@pytest.mark.parametrize("name", SCENES)
@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "e"])
def test_pixels(gpu_scenes, sset, name, tag):
    """processItem on 1x1 WorkItems with given engines: pixel value and engine state afterwards, bit for bit."""
    g = golden("pixels_%s_%s" % (name, tag))
    cam = sset[name][1]
    img, st = gpu_scenes(name).process_item(cam, opt_from(g["options"]), binding.pixel_streams(g["xs"], g["ys"], g["states"]))
    assert_bits_equal(st, g["states_out"], "engine state after the pixel (draw count)")
    key = g["ys"].astype(np.int64) * 65536 + g["xs"]
    uniq, first, counts = np.unique(key, return_index=True, return_counts=True)
    single = first[counts == 1]
    assert_bits_equal(img[g["ys"][single], g["xs"][single]], g["rgba"][single], "pixel value")


BRANCHES = ["hex_cornell", "hex_meshbox", "nolens_box", "nolens_advanced", "oneway", "oneway_hex"]


@pytest.mark.parametrize("name", BRANCHES)
def test_branch_pixels(sset, name):
    """The device branches no other scene reaches -- HexagonalApertureSampler (src/camera.cpp:21-50), an aperture without a focal plane
    (camera.cpp:93-99,109) and MirrorBRDF(one_way = true) (src/scene/propagation.cpp:178-217) -- against values and engine states
    recorded from the compiled reference (tests/golden/branch_*.npz)."""
    from tests.cases import branch_cases
    g = golden("branch_" + name)
    desc, cam = branch_cases(sset)[name]
    sc = binding.Scene(desc)
    try:
        for tag in "abc":
            xs, ys = g[tag + "_xs"], g[tag + "_ys"]
            img, st = sc.process_item(cam, opt_from(g[tag + "_options"]), binding.pixel_streams(xs, ys, g[tag + "_states"]))
            assert_bits_equal(st, g[tag + "_states_out"], "engine state after the pixel (draw count)")
            key = ys.astype(np.int64) * 65536 + xs
            _, first, counts = np.unique(key, return_index=True, return_counts=True)
            single = first[counts == 1]
            assert_bits_equal(img[ys[single], xs[single]], g[tag + "_rgba"][single], "pixel value")
    finally:
        sc.close()


def _tile_stream(x, y, w, h, seed):
    s = np.zeros(1, dtype=binding.STREAM_DTYPE)
    s["x"], s["y"], s["w"], s["h"], s["rng_state"] = x, y, w, h, binding.seed_to_state(seed)
    return s


def test_tiles_one_engine(gpu_scenes, sset):
    """processItem on whole tiles through ONE engine, as the reference's workers run it (pixels in row-major order)."""
    g = golden("tiles")
    cam = sset["cornell"][1]
    sc = gpu_scenes("cornell")
    for key, skey, opt, rect, seed in [("cornell_16_16", "cornell_16_16_state", scenes.options(256, 256, 16, 16), (0, 0, 32, 32), 1234),
                                       ("cornell_mid_16_64", "cornell_mid_state", scenes.options(256, 256, 16, 64), (96, 128, 32, 32), 99)]:
        img, st = sc.process_item(cam, opt, _tile_stream(*rect, seed))
        x, y, w, h = rect
        assert_bits_equal(img[y:y + h, x:x + w], g[key], key)
        assert_bits_equal(st, g[skey], skey)
    adv_cam = sset["advanced"][1]
    img, st = gpu_scenes("advanced").process_item(adv_cam, scenes.options(132, 68, 5, 10), _tile_stream(128, 64, 4, 4, 7))
    assert_bits_equal(img[64:68, 128:132], g["advanced_edge"], "clipped edge tile")
    assert_bits_equal(st, g["advanced_edge_state"], "edge tile state")


@pytest.mark.parametrize("name,w,h,mn,mx", [("box", 64, 64, 16, 64), ("cornell", 48, 40, 16, 16), ("advanced", 66, 34, 5, 10), ("meshbox", 32, 32, 8, 8), ("dragons16", 48, 48, 8, 8)])
def test_process_job_vs_oracle(gpu_scenes, sset, oracle_lib, name, w, h, mn, mx):
    """processJob with per-pixel engines against the CPU oracle run on the same engines.  Stated tolerance: per-pixel
    L2 error < 1e-4 (BASELINE.json north_star); the implementation is expected to be bit-identical."""
    sc_desc, cam = sset[name]
    if name == "cornell":
        cam = dict(cam, aspect_ratio=-float(np.float32(w) / np.float32(h)))
    opt = scenes.options(w, h, mn, mx)
    img = gpu_scenes(name).process_job(cam, opt, base_seed=1234)
    ys, xs = np.mgrid[0:h, 0:w]
    xs, ys = xs.ravel().astype(np.int32), ys.ravel().astype(np.int32)
    states = np.array([binding.seed_to_state(binding.pixel_seed(1234, int(x), int(y))) for x, y in zip(xs, ys)], np.uint64)
    ref_img, _ = oracle_lib.scene_create(sc_desc).render_streams(cam, opt, oracle.pixel_streams(xs, ys, states), n_threads=8)
    l2 = np.sqrt(((img.astype(np.float64) - ref_img.astype(np.float64)) ** 2).sum(axis=2))
    assert l2.max() < 1e-4, "per-pixel L2 %g" % l2.max()
    same = (img.view(np.uint32) == ref_img.view(np.uint32)).all(axis=2).mean()
    assert same == 1.0, "only %.4f of the pixels are bit-identical" % same


def _scene_with(mode, desc):
    import os
    old = os.environ.get("PT_BUILD")
    os.environ["PT_BUILD"] = mode
    try:
        return binding.Scene(desc)
    finally:
        if old is None:
            del os.environ["PT_BUILD"]
        else:
            os.environ["PT_BUILD"] = old


def _tie_heavy_scene():
    """Many equal low coordinates: the stable partition keeps all ties on the left and scene.cpp:89-94 has to move objects."""
    rng = np.random.default_rng(77)
    sb = scenes.SceneBuilder()
    lit = sb.material((1, 1, 1, 1), 1.0, (2, 1, 0.5, 1))
    one = np.array([[[0.1, 0.2, 0.3], [0.4, 0.2, 0.3], [0.1, 0.5, 0.6]]], np.float32)
    sb.triangles(np.repeat(one, 700, axis=0))                                    # 700 identical triangles
    grid = np.floor(rng.uniform(-4, 4, (1500, 1, 3))).astype(np.float32) * np.float32(0.25)
    sb.triangles(grid + rng.uniform(0, 0.2, (1500, 3, 3)).astype(np.float32) * np.array([[[0, 0, 0]], [[1, 1, 1]], [[1, 1, 1]]], np.float32).reshape(1, 3, 3))
    sb.triangles(np.repeat(one + np.float32(1.0), 130, axis=0), lit, cull=True)   # identical AND emissive
    sb.sphere((0.3, 0.1, -0.2), 0.25, sb.material((0, 0, 1, 1), bsdf=scenes.BSDF_MIRROR))
    sb.sphere((0.3, 0.1, -0.2), 0.25, lit)
    sb.triangles(rng.uniform(-1, 1, (900, 3, 3)).astype(np.float32), lit)
    return sb.build()


def _many_materials_scene():
    """24 materials (more than the path kernel keeps in LDS: the material table is read from global memory) of all three BSDF kinds on
    a wall of small quads, lit by an emissive quad and an emissive sphere (3 emitters: the emitter tables ARE in LDS)."""
    sb = scenes.SceneBuilder()
    sb.triangles(scenes.make_box((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)), sb.material((0.8, 0.8, 0.8, 1.0)))
    lit = sb.material((1, 1, 1, 1), 1.0, (3, 2.5, 2, 1))
    sb.triangles(scenes.make_plane((-0.3, 0.95, -0.3), (0.3, 0.95, 0.3)), lit)
    sb.sphere((-0.6, 0.5, -0.5), 0.12, sb.material((1, 1, 1, 1), 1.0, (1, 2, 4, 1)))
    rng = np.random.default_rng(5)
    for k in range(21):
        kind = (scenes.BSDF_LAMBERTIAN, scenes.BSDF_GLASS, scenes.BSDF_MIRROR)[k % 3]
        colour = tuple(float(v) for v in rng.uniform(0.2, 1.0, 3)) + (1.0,)
        m = sb.material(colour, 1.0 + 0.05 * k, bsdf=kind, specular=tuple(float(v) for v in rng.uniform(0.5, 1.0, 3)) + (1.0,))
        x, y = -0.9 + 0.25 * (k % 7), -0.8 + 0.45 * (k // 7)
        z = 0.3 + 0.02 * k
        sb.triangles([[(x, y, z), (x + 0.22, y, z + 0.05), (x, y + 0.4, z)], [(x + 0.22, y, z + 0.05), (x + 0.22, y + 0.4, z + 0.05), (x, y + 0.4, z)]], m)
    return sb.build()


@pytest.mark.parametrize("name", SCENES + ["dragons16", "ties", "ties_face_normals"])
def test_device_build_matches_host_build(sset, oracle_lib, name):
    """the tree built level by level on the device (pt_build.hip) against the host recursion (pt_bvh.cpp) and the
    CPU oracle: same pre-order topology, same boxes bit for bit, same emissive registration order and CDF, same closest hits."""
    desc = _tie_heavy_scene() if name.startswith("ties") else sset[name][0]
    oracle_desc = desc
    if name == "ties_face_normals":
        desc = dict(desc, tri_nrm=None)  # Triangle::Triangle without normals: face normals made by the record kernel
    dev, host = _scene_with("device", desc), _scene_with("host", desc)
    try:
        (od, bd), (oh, bh) = dev.bvh_dump(), host.bvh_dump()
        assert_bits_equal(od, oh, "topology, device build vs host build")
        assert_bits_equal(bd, bh, "boxes, device build vs host build")
        oo, bo = oracle_lib.bvh_dump(oracle_desc)  # normals play no part in the tree
        assert_bits_equal(od, oo, "topology, device build vs oracle")
        assert_bits_equal(bd, bo, "boxes, device build vs oracle")
        assert dev.info() == host.info()
        for a, b in zip(dev.emissive(), host.emissive()):
            assert_bits_equal(a, b, "emissive objects / CDF")
        rng = np.random.default_rng(11)
        d = rng.normal(size=(20000, 3))
        rays = np.concatenate([rng.uniform(-1, 1, (20000, 3)), d / np.linalg.norm(d, axis=1, keepdims=True)], axis=1).astype(np.float32)
        (td, hd), (th, hh) = dev.get_intersection(rays), host.get_intersection(rays)
        assert_bits_equal(td, th, "closest hit t")
        assert_bits_equal(hd, hh, "closest hit object")
        cam = scenes.camera((0, 0, -3), (0, 0, 0), (0, 1, 0), 1.0, 1.0, -1.0)
        opt = scenes.options(24, 24, 4, 4)
        assert_bits_equal(dev.process_job(cam, opt, base_seed=5), host.process_job(cam, opt, base_seed=5), "rendered frame")
    finally:
        dev.close()
        host.close()


def test_reference_render_tests(gpu_scenes, sset):
    """test/render_test.cpp: empty scene -> exact zero; simple and advanced scenes -> zero corner, centre alpha > 0."""
    sc, cam = scenes.empty_scene()
    s = binding.Scene(sc)
    img = s.process_job(cam, scenes.options(1, 1, 1, 1))
    assert img[0, 0].tolist() == [0.0, 0.0, 0.0, 0.0]
    s.close()
    img = gpu_scenes("simple").process_job(sset["simple"][1], scenes.options(16, 16, 2, 2))
    assert img[0, 0].tolist() == [0.0, 0.0, 0.0, 0.0] and img[8, 8, 3] > 0
    img = gpu_scenes("advanced").process_job(sset["advanced"][1], scenes.options(132, 68, 5, 10))
    assert img[0, 0].tolist() == [0.0, 0.0, 0.0, 0.0] and img[32, 64, 3] > 0


def test_image_independent_of_tiling(gpu_scenes, sset):
    """Per-pixel engines make the image a function of (scene, camera, options, base seed) only -- the property the multi-GPU
    sharding relies on."""
    cam = sset["box"][1]
    opt = scenes.options(96, 64, 8, 32)
    sc = gpu_scenes("box")
    full = sc.process_job(cam, opt, base_seed=7)
    tiles = binding.job_tiles(96, 64)
    assert len(tiles) == 6 * 4 and tiles["w"].max() == 16
    halves = np.zeros_like(full)
    sc.process_job(cam, opt, base_seed=7, tiles=tiles[0::2], image=halves)
    sc.process_job(cam, opt, base_seed=7, tiles=tiles[1::2], image=halves)
    assert_bits_equal(halves, full, "interleaved tiles")
    odd = np.array([(0, 0, 96, 1), (0, 1, 5, 63), (5, 1, 91, 63)], dtype=binding.TILE_DTYPE)
    assert_bits_equal(sc.process_job(cam, opt, base_seed=7, tiles=odd), full, "ragged tiles")


def test_work_item_returns_its_own_tile(gpu_scenes, sset):
    """pt_render_item = processItem returning the item's rectangle only (the C++ processItem's path): the golden tiles again, and the
    frame-sized call on the same item."""
    g = golden("tiles")
    cam = sset["cornell"][1]
    sc = gpu_scenes("cornell")
    opt = scenes.options(256, 256, 16, 64)
    tile, state = sc.process_work_item(cam, opt, 96, 128, 32, 32, binding.seed_to_state(99))
    assert_bits_equal(tile, g["cornell_mid_16_64"], "tile")
    assert state == int(g["cornell_mid_state"][0])
    img, st = sc.process_item(cam, opt, _tile_stream(96, 128, 32, 32, 99))
    assert_bits_equal(tile, img[128:160, 96:128], "same item through the frame-sized call")
    tile, state = sc.process_work_item(cam, opt, 3, 3, 0, 0, 1234)   # zero-area item: nothing rendered, engine untouched
    assert tile.size == 0 and state == 1234
    with pytest.raises(binding.PtError):
        sc.process_work_item(cam, opt, 250, 250, 32, 32, 1)


def test_progress_and_replicas(sset):
    """pt_render_tiles_progress / pt_render_tiles_multi: the callback sees 1 .. n_tiles in order from the calling thread; two replicas of a
    scene (both on this box's one GPU) deal the tiles out between them and produce the single-scene frame bit for bit."""
    import threading
    desc, cam = sset["box"]
    opt = scenes.options(160, 96, 16, 16)
    a, b = binding.Scene(desc), binding.Scene(desc)
    try:
        want = a.process_job(cam, opt, base_seed=9)
        calls, threads = [], set()
        img = a.process_job_progress(cam, opt, lambda done, total: (calls.append((done, total)), threads.add(threading.get_ident())), base_seed=9)
        n_tiles = len(binding.job_tiles(160, 96))
        assert calls == [(k + 1, n_tiles) for k in range(n_tiles)]
        assert threads == {threading.get_ident()}
        assert_bits_equal(img, want, "frame rendered with a progress callback")
        calls.clear()
        img2, stats = binding.process_job_multi([a, b], cam, opt, base_seed=9, progress=lambda done, total: calls.append((done, total)), want_stats=True)
        assert_bits_equal(img2, want, "frame rendered by two replicas")
        assert [c[0] for c in calls] == list(range(1, n_tiles + 1))
        assert sum(s["samples"] for s in stats) == 160 * 96 * 16 and all(s["samples"] > 0 for s in stats)
    finally:
        a.close()
        b.close()


def test_threads_share_a_scene(gpu_scenes, sset):
    """Render calls on one scene from several threads are serialised inside the library (one workspace per scene): same results as alone."""
    import threading
    cam = sset["box"][1]
    sc = gpu_scenes("box")
    opt = scenes.options(64, 64, 8, 8)
    alone = [sc.process_work_item(cam, opt, 8 * k, 4 * k, 16, 16, binding.seed_to_state(100 + k)) for k in range(4)]
    got = [None] * 4

    def work(k):
        for _ in range(3):
            got[k] = sc.process_work_item(cam, opt, 8 * k, 4 * k, 16, 16, binding.seed_to_state(100 + k))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for k in range(4):
        assert_bits_equal(got[k][0], alone[k][0], "tile %d" % k)
        assert got[k][1] == alone[k][1]


def test_edge_cases(gpu_scenes, sset):
    sc = gpu_scenes("box")
    cam = sset["box"][1]
    # empty inputs
    t, obj = sc.get_intersection(np.zeros((0, 6), np.float32))
    assert len(t) == 0
    img, st = sc.process_item(cam, scenes.options(8, 8, 1, 1), np.zeros(0, dtype=binding.STREAM_DTYPE))
    assert (img == 0).all()
    # a zero-area WorkItem renders nothing and leaves its engine untouched
    s = _tile_stream(3, 3, 0, 0, 5)
    img, st = sc.process_item(cam, scenes.options(8, 8, 1, 1), s)
    assert (img == 0).all() and st[0] == s["rng_state"][0]
    # max_sample_count 0: no sample is drawn
    img, st = sc.process_item(cam, scenes.options(8, 8, 0, 0), _tile_stream(0, 0, 8, 8, 5))
    assert (img == 0).all() and st[0] == binding.seed_to_state(5)
    # rectangle outside the image is refused
    with pytest.raises(binding.PtError):
        sc.process_item(cam, scenes.options(8, 8, 1, 1), _tile_stream(4, 4, 8, 8, 5))


def test_post_processing_golden():
    """Device toneMap / gammaCorrect / postProcess (pt_post.hip) against the frames the compiled reference produced."""
    for img, steps, gamma, want, label in post_cases():
        assert_bits_equal(binding.post_process(img, steps, gamma), want, label)


def test_post_processing_full_frame(gpu_scenes, sset, oracle_lib):
    """A rendered 1024 x 1024 frame (BASELINE.json's size) through postProcess on the device and through the CPU oracle."""
    cam = dict(sset["cornell"][1], aspect_ratio=-1.0)
    frame = gpu_scenes("cornell").process_job(cam, scenes.options(1024, 1024, 2, 2), base_seed=1234)
    for steps, gamma in ((1, 1.8), (2, 2.2), (3, 1.8)):
        assert_bits_equal(binding.post_process(frame, steps, gamma), oracle_lib.post_process(frame, steps, gamma), "1024x1024 steps %d" % steps)
    # idempotence-like property at full size: gamma 1 leaves the frame as it is (reference test/post_processing_test.cpp:36-46)
    lit = frame[..., :3].max(axis=2) > 0
    assert_bits_equal(binding.post_process(frame, 2, 1.0)[lit], frame[lit], "gamma 1")


@pytest.fixture(scope="module")
def mesh7m(oracle_lib):
    """bench.py's default workload itself: DragonBox with the 7.2 M-triangle stand-in mesh, a 30-level tree built on the device, walks that
    outgrow the LDS window of the traversal stack -- on the GPU and in the CPU oracle (same arrays)."""
    desc, cam = scenes.dragon_box_scene(*scenes.bumpy_sphere_mesh(1900, 1900, scenes.DRAGON_BOX_TRANSFORM))
    scene = binding.Scene(desc)
    handle = oracle_lib.scene_create(desc)
    yield desc, cam, scene, handle
    scene.close()
    handle.close()


def _check_frame_against_oracle(scene, handle, cam, opt, seed, n_pixels=3000, quadrants=True):
    """3000 random pixels rendered by the CPU oracle with the same per-pixel engines must equal the frame's pixels bit for bit; the frame is
    reproducible; the sum of its four quadrants rendered as separate jobs (disjoint tile sets) reproduces it exactly."""
    w, h = opt["image_width"], opt["image_height"]
    frame = scene.process_job(cam, opt, base_seed=seed)
    rng = np.random.default_rng(5)
    xs, ys = rng.integers(0, w, n_pixels).astype(np.int32), rng.integers(0, h, n_pixels).astype(np.int32)
    states = np.array([binding.seed_to_state(binding.pixel_seed(seed, int(x), int(y))) for x, y in zip(xs, ys)], np.uint64)
    want, _ = handle.render_streams(cam, opt, oracle.pixel_streams(xs, ys, states), n_threads=16)
    assert_bits_equal(frame[ys, xs], want[ys, xs], "sampled pixels of the full-size frame")
    assert_bits_equal(scene.process_job(cam, opt, base_seed=seed), frame, "same job twice")
    if quadrants:
        tiles = binding.job_tiles(w, h)
        parts = np.zeros_like(frame)
        for qx in (0, 1):
            for qy in (0, 1):
                mine = tiles[(tiles["x"] // (w // 2) == qx) & (tiles["y"] // (h // 2) == qy)]
                parts += scene.process_job(cam, opt, base_seed=seed, tiles=mine)
        assert_bits_equal(parts, frame, "four quadrant jobs add up to the frame")
    return frame


@pytest.mark.parametrize("w,h", [(700, 700), (1000, 1000), (1448, 1448), (1031, 517)])
def test_frame_sizes_between_the_grid_steps(sset, oracle_lib, w, h):
    """Frames that do not fill the wavefronts' slots evenly (the first round of streams rounds a wavefront's slots up to whole pieces, or
    hands out more streams than slots: DESIGN.md 4), with tiles cut at the frame's edges: sampled against the oracle."""
    desc, cam = sset["cornell"]
    scene = binding.Scene(desc)
    handle = oracle_lib.scene_create(desc)
    try:
        _check_frame_against_oracle(scene, handle, cam, scenes.options(w, h, 3, 3), seed=77, n_pixels=1500, quadrants=False)
    finally:
        scene.close()
        handle.close()


@pytest.mark.parametrize("which,spp_min,spp_max", [("cornell", 16, 16), ("mesh80k", 8, 8), ("cornell_adaptive", 4, 24), ("dragons16_180k", 8, 8)])
def test_full_size_frame_sampled_against_oracle(sset, oracle_lib, which, spp_min, spp_max):
    """BASELINE.json's frame size (1024 x 1024, one stream per pixel = 1 M streams in flight) checked where the oracle can follow.
    dragons16_180k is configs[4] at 16 x 179,400 = 2.87 M triangles (the bench runs it at 16 x 7.2 M, where the CPU oracle cannot follow)."""
    if which.startswith("cornell"):
        desc, cam = sset["cornell"]
        cam = dict(cam, aspect_ratio=-1.0)
    elif which == "dragons16_180k":
        desc, cam = scenes.dragon_grid_scene(*scenes.bumpy_sphere_mesh(300, 300, scenes.DRAGON_BOX_TRANSFORM), grid=4)
    else:
        desc, cam = scenes.dragon_box_scene(*scenes.bumpy_sphere_mesh(200, 200, scenes.DRAGON_BOX_TRANSFORM))
    scene = binding.Scene(desc)
    handle = oracle_lib.scene_create(desc)
    try:
        _check_frame_against_oracle(scene, handle, cam, scenes.options(1024, 1024, spp_min, spp_max), 77)
    finally:
        scene.close()
        handle.close()


def test_configs4_at_full_size(oracle_lib):
    """BASELINE.json configs[4] at its full size -- 16 transformed copies of the 7.2 M-triangle stand-in = 115,459,214 triangles, a 34-level
    tree built on the device -- 1024 x 1024 at 2 spp: 512 random pixels rendered by the CPU oracle (whose tree of this size takes about 75 s
    and 25 GB to build) with the same per-pixel engines, bit for bit."""
    desc, cam = scenes.dragon_grid_scene(*scenes.bumpy_sphere_mesh(1900, 1900, scenes.DRAGON_BOX_TRANSFORM), grid=4)
    assert len(desc["tri_pos"]) == 115459214
    opt = scenes.options(1024, 1024, 2, 2)
    scene = binding.Scene(desc)
    try:
        frame = scene.process_job(cam, opt, base_seed=77)
    finally:
        scene.close()
    handle = oracle_lib.scene_create(desc)
    try:
        rng = np.random.default_rng(5)
        xs, ys = rng.integers(0, 1024, 512).astype(np.int32), rng.integers(0, 1024, 512).astype(np.int32)
        states = np.array([binding.seed_to_state(binding.pixel_seed(77, int(x), int(y))) for x, y in zip(xs, ys)], np.uint64)
        want, _ = handle.render_streams(cam, opt, oracle.pixel_streams(xs, ys, states), n_threads=16)
    finally:
        handle.close()
    assert_bits_equal(frame[ys, xs], want[ys, xs], "sampled pixels of the 115 M-triangle frame")
    assert (want[ys, xs][:, 3] == 1).all()


def test_mesh7m_frame(mesh7m):
    """configs[2] at frame size: the 7.2 M-triangle DragonBox, 1024 x 1024, 4 spp, sampled against the oracle."""
    desc, cam, scene, handle = mesh7m
    _check_frame_against_oracle(scene, handle, cam, scenes.options(1024, 1024, 4, 4), 77)


@pytest.mark.parametrize("w,spp,n_pixels", [(1024, 256, 1024), (2048, 4096, 24)])
def test_mesh7m_long_chains(mesh7m, w, spp, n_pixels):
    """configs[2] / configs[3] per pixel: 256-sample (4096-sample) chains on the big mesh as processItem streams with given engines --
    pixel value AND engine state afterwards (= the exact number of draws of every path of the chain) against the oracle."""
    desc, cam, scene, handle = mesh7m
    opt = scenes.options(w, w, spp, spp)
    rng = np.random.default_rng(w + spp)
    # half of the pixels anywhere, half inside the mesh's silhouette (long glass paths, suspended walks)
    xs = np.concatenate([rng.integers(0, w, n_pixels // 2), rng.integers(int(0.36 * w), int(0.64 * w), n_pixels - n_pixels // 2)]).astype(np.int32)
    ys = np.concatenate([rng.integers(0, w, n_pixels // 2), rng.integers(int(0.36 * w), int(0.64 * w), n_pixels - n_pixels // 2)]).astype(np.int32)
    keep = np.unique(ys.astype(np.int64) * 65536 + xs, return_index=True)[1]
    xs, ys = xs[keep], ys[keep]
    states = rng.integers(1, 2**63, len(xs)).astype(np.uint64)
    img, after = scene.process_item(cam, opt, binding.pixel_streams(xs, ys, states))
    want, want_after = handle.render_streams(cam, opt, oracle.pixel_streams(xs, ys, states), n_threads=16)
    assert_bits_equal(img[ys, xs], want[ys, xs], "pixel values after %d-sample chains" % spp)
    assert_bits_equal(after, want_after, "engine states after %d-sample chains" % spp)
    assert (img[ys, xs][:, 3] == 1).mean() > 0.5


def test_mesh7m_c4_frame_and_rank_replay(mesh7m):
    """BASELINE.json configs[3] -- the dragon at 2048 x 2048 = 4096 tiles, sharded over 8 GPUs -- at 2 spp: the frame against oracle-sampled
    pixels, and the frame assembled from the 8 ranks' tile sets (played one after the other on this GPU with sharding.py's own split and
    gather indices) against the single-rank frame, bit for bit."""
    from cpupathtrace_amd import sharding
    desc, cam, scene, handle = mesh7m
    side, world = 2048, 8
    opt = scenes.options(side, side, 2, 2)
    full = _check_frame_against_oracle(scene, handle, cam, opt, 21, n_pixels=2000, quadrants=False)
    tiles = binding.job_tiles(side, side)
    assert len(tiles) == 4096
    assembled = np.zeros_like(full).reshape(-1, 4)
    covered = np.zeros(side * side, np.int32)
    for rank in range(world):
        mine = sharding.local_tiles(tiles, rank, world)
        assert len(mine) == 512
        img = scene.process_job(cam, opt, base_seed=21, tiles=mine).reshape(-1, 4)
        idx = sharding.pixel_indices(mine, side)
        assembled[idx] = img[idx]
        covered[idx] += 1
    assert (covered == 1).all()
    assert_bits_equal(assembled.reshape(full.shape), full, "2048x2048 frame assembled from 8 ranks")


@pytest.mark.parametrize("world,side", [(2, 362), (4, 512), (8, 724)])
def test_frame_identical_for_any_gpu_count(gpu_scenes, sset, world, side):
    """the assembled frame is bit-identical for 1, 2, 4, 8 ranks.  The ranks of a `world` are played one after the other
    on this GPU with sharding.py's own tile split and gather indices (frame sides = bench.py's weak-scaling sides / 4)."""
    from cpupathtrace_amd import sharding
    cam = dict(sset["cornell"][1], aspect_ratio=-1.0)
    opt = scenes.options(side, side, 4, 4)
    sc = gpu_scenes("cornell")
    full = sc.process_job(cam, opt, base_seed=21)
    tiles = binding.job_tiles(side, side)
    assembled = np.zeros_like(full).reshape(-1, 4)
    covered = np.zeros(side * side, np.int32)
    for rank in range(world):
        mine = sharding.local_tiles(tiles, rank, world)
        img = sc.process_job(cam, opt, base_seed=21, tiles=mine).reshape(-1, 4)
        idx = sharding.pixel_indices(mine, side)
        assembled[idx] = img[idx]      # what rank 0 does with the gathered chunk of this rank
        covered[idx] += 1
    assert (covered == 1).all(), "every pixel belongs to exactly one rank"
    assert_bits_equal(assembled.reshape(full.shape), full, "frame assembled from %d ranks" % world)


@pytest.mark.parametrize("name,w,h,mn,mx,eps", [
    ("advanced", 1, 1, 1, 1, 1e-3),         # a single pixel, a single sample
    ("advanced", 3, 5, 0, 7, 1e-3),         # min 0: the estimator may stop at its first check
    ("advanced", 5, 3, 9, 4, 1e-3),         # min > max: max decides (worker.cpp:193)
    ("cornell", 2, 2, 4096, 4096, 1e-3),    # long sequential chains: 4096 samples through every statistics batch
    ("cornell", 7, 6, 64, 2048, 1e-3),      # adaptive, early acceptance
    ("advanced", 16, 9, 8, 8, 1e-2),        # large epsilon: shadow thresholds below zero, self-intersection offsets
    ("box", 11, 13, 8, 8, 1e-6),            # tiny epsilon
    ("ties", 12, 12, 6, 6, 1e-3),           # 1031 emissive objects: 5 object samples per vertex, duplicated triangles
    ("many_materials", 24, 20, 12, 12, 1e-3),  # 24 materials (table in global memory), 3 emitters incl. a sphere (tables in LDS)
])
def test_unusual_options_vs_oracle(sset, oracle_lib, name, w, h, mn, mx, eps):
    default_cam = scenes.camera((0, 0, -3), (0, 0, 0), (0, 1, 0), 1.0, 1.0, -1.0)
    desc, cam = (_tie_heavy_scene(), default_cam) if name == "ties" else (_many_materials_scene(), default_cam) if name == "many_materials" else sset[name]
    opt = scenes.options(w, h, mn, mx, eps)
    scene = binding.Scene(desc)
    try:
        img = scene.process_job(cam, opt, base_seed=4321)
        ys, xs = np.mgrid[0:h, 0:w]
        xs, ys = xs.ravel().astype(np.int32), ys.ravel().astype(np.int32)
        states = np.array([binding.seed_to_state(binding.pixel_seed(4321, int(x), int(y))) for x, y in zip(xs, ys)], np.uint64)
        want, states_after = oracle_lib.scene_create(desc).render_streams(cam, opt, oracle.pixel_streams(xs, ys, states), n_threads=8)
        assert_bits_equal(img, want, "%s %dx%d spp %d..%d eps %g" % (name, w, h, mn, mx, eps))
        # the same pixels as explicit streams: the engines must come back in the oracle's state
        img2, after = scene.process_item(cam, opt, binding.pixel_streams(xs, ys, states))
        assert_bits_equal(img2, want, "streams")
        assert_bits_equal(after, states_after, "engine states after the pixels")
    finally:
        scene.close()


def _pixel_job(gpu, chk_scene, cam, opt, seed=2468):
    """One processJob on the device (with its work counters) and the same pixels through the oracle (with its counters)."""
    w, h = opt["image_width"], opt["image_height"]
    img, st = gpu.process_job(cam, opt, base_seed=seed, want_stats=True)
    ys, xs = np.mgrid[0:h, 0:w]
    xs, ys = xs.ravel().astype(np.int32), ys.ravel().astype(np.int32)
    states = np.array([binding.seed_to_state(binding.pixel_seed(seed, int(x), int(y))) for x, y in zip(xs, ys)], np.uint64)
    chk_scene.counters_reset()
    want, _ = chk_scene.render_streams(cam, opt, oracle.pixel_streams(xs, ys, states), n_threads=8)
    assert_bits_equal(img, want, "frame")
    return st, chk_scene.counters()


@pytest.mark.parametrize("name,w,h,mn,mx", [("box", 40, 40, 8, 8), ("meshbox", 32, 32, 8, 8), ("cornell", 32, 32, 4, 24)])
def test_kernel_work_counters_vs_oracle(gpu_scenes, sset, oracle_lib, name, w, h, mn, mx):
    """The counters that feed bench.py's roofline (pt_stats: samples, vertices, rays, node visits, leaf tests -- counted by the kernel in
    per-wave registers) against the oracle's own counters for the same pixels.  Samples, path vertices and extension rays (camera +
    bounce) must be EQUAL.  Shadow rays may only be fewer: the kernel traces none where the reference's result does not depend on it
    (specular BSDFs: p = 0 for synthetic pairs, worker.cpp:92; contributions of +-0; thresholds <= 0), and a shadow walk stops at its
    first occluder, so slab and leaf tests are bounded by the oracle's, never above."""
    desc, cam = sset[name]
    if name == "cornell":
        cam = dict(cam, aspect_ratio=-float(np.float32(w) / np.float32(h)))
    st, c = _pixel_job(gpu_scenes(name), oracle_lib.scene_create(desc), cam, scenes.options(w, h, mn, mx))
    assert st["samples"] == c["samples"] > 0
    assert st["vertices"] == c["vertices"] > 0
    assert st["rays_traced"] - st["shadow_rays_traced"] == c["scene_queries"] - c["shadow_rays"], "extension rays (camera + bounce)"
    assert 0 < st["shadow_rays_traced"] <= c["shadow_rays"]
    assert st["rays_traced"] <= c["scene_queries"]
    assert 2 * st["node_visits"] + st["rays_traced"] <= c["aabb_tests"], "two slab tests per inner node + the root test of every ray"
    assert st["leaf_tests"] <= c["tri_tests"] + c["sphere_tests"]
    # ... and not far below: the shortcuts only ever drop shadow rays
    assert 2 * st["node_visits"] + st["rays_traced"] >= 0.3 * c["aabb_tests"]


def test_kernel_work_counters_exact_without_lights(oracle_lib):
    """No emitter, no point light -> no shadow rays at all: every counter of the kernel equals the oracle's (each walk visits exactly the
    nodes and leaves the reference's recursion visits)."""
    sb = scenes.SceneBuilder()
    sb.triangles(scenes.make_box((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)), sb.material((0.8, 0.7, 0.6, 1.0)))
    sb.sphere((0.3, -0.6, 0.2), 0.4, sb.material((1, 1, 1, 1), 1.5, bsdf=scenes.BSDF_GLASS))
    sb.sphere((-0.4, -0.7, -0.2), 0.3, sb.material((1, 1, 1, 1), bsdf=scenes.BSDF_MIRROR))
    rng = np.random.default_rng(9)
    sb.triangles(rng.uniform(-0.9, 0.9, (1500, 1, 3)).astype(np.float32) + rng.uniform(-0.08, 0.08, (1500, 3, 3)).astype(np.float32), sb.material((0.5, 0.9, 0.5, 1.0)))
    desc = sb.build()
    cam = scenes.camera((0, 0, -3), (0, 0, 0), (0, 1, 0), 1.0, 1.0, -1.0)
    for mode in ("host", "device"):
        scene = _scene_with(mode, desc)
        try:
            st, c = _pixel_job(scene, oracle_lib.scene_create(desc), cam, scenes.options(48, 48, 6, 6))
        finally:
            scene.close()
        assert c["shadow_rays"] == 0 and st["shadow_rays_traced"] == 0
        assert st["samples"] == c["samples"] == 48 * 48 * 6
        assert st["vertices"] == c["vertices"] > 0
        assert st["rays_traced"] == c["scene_queries"]
        assert 2 * st["node_visits"] + st["rays_traced"] == c["aabb_tests"]
        assert st["leaf_tests"] == c["tri_tests"] + c["sphere_tests"]


def _lit_room(n_point_lights, emitters):
    """A closed room with Lambertian, glass and mirror objects, `n_point_lights` PointLightSources and the given emissive geometry."""
    sb = scenes.SceneBuilder()
    sb.triangles(scenes.make_box((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)), sb.material((0.75, 0.7, 0.65, 1.0)))
    sb.sphere((0.35, -0.6, 0.1), 0.35, sb.material((1, 1, 1, 1), 1.5, bsdf=scenes.BSDF_GLASS))
    sb.sphere((-0.45, -0.7, -0.3), 0.28, sb.material((0.9, 0.9, 1.0, 1), bsdf=scenes.BSDF_MIRROR))
    rng = np.random.default_rng(31)
    sb.triangles(rng.uniform(-0.8, 0.8, (300, 1, 3)).astype(np.float32) + rng.uniform(-0.1, 0.1, (300, 3, 3)).astype(np.float32), sb.material((0.4, 0.8, 0.5, 1.0)))
    emitters(sb)
    for k in range(n_point_lights):
        a = 2.0 * np.pi * k / max(n_point_lights, 1)
        sb.point_light((0.7 * np.cos(a), 0.3 + 0.05 * k, 0.7 * np.sin(a)), (0.2 + 0.05 * k, 0.3, 0.5 - 0.02 * k, 1.0))
    return sb.build()


def _three_emitters(sb):
    sb.triangles(scenes.make_plane((-0.25, 0.97, -0.25), (0.25, 0.97, 0.25)), sb.material((1, 1, 1, 1), 1.0, (4, 3.5, 3, 1)), cull=True)
    sb.sphere((-0.6, 0.4, 0.5), 0.1, sb.material((1, 1, 1, 1), 1.0, (1, 2, 4, 1)))


def _emissive_mesh(sb):
    """100,352 small emissive triangles: min(2 + int(log10(E + 1)), E) = 7 object samples per vertex (scene.cpp:226)"""
    n = 224
    xs, zs = np.meshgrid(np.linspace(-0.6, 0.6, n + 1, dtype=np.float32), np.linspace(-0.6, 0.6, n + 1, dtype=np.float32))
    y = (np.float32(0.9) + np.float32(0.03) * np.sin(7 * xs) * np.cos(5 * zs)).astype(np.float32)
    p = np.stack([xs, y, zs], axis=-1)
    a, b, c, d = p[:-1, :-1], p[:-1, 1:], p[1:, :-1], p[1:, 1:]
    tris = np.concatenate([np.stack([a, c, b], axis=2).reshape(-1, 3, 3), np.stack([b, c, d], axis=2).reshape(-1, 3, 3)])
    sb.triangles(tris, sb.material((1, 1, 1, 1), 1.0, (2, 1.8, 1.5, 1)))


@pytest.mark.parametrize("name,n_lights,emitters,expect_samples,w,h,mn,mx", [
    ("12 point lights + 3 emitters", 12, _three_emitters, 14, 40, 40, 6, 6),
    ("emissive mesh (k = 7) + 2 point lights", 2, _emissive_mesh, 9, 24, 24, 4, 4),
    ("30 point lights + 3 emitters, adaptive", 30, _three_emitters, 32, 24, 24, 4, 12)])
def test_many_light_samples_vs_oracle(oracle_lib, name, n_lights, emitters, expect_samples, w, h, mn, mx):
    """Scene::sampleLights returns every LightSource and min(2 + int(log10(E + 1)), E) emitter samples per path vertex (scene.cpp:226,231-289)
    with no upper bound; the device keeps a vertex's visibility bits in a 32-bit mask (round 2: 8).  Frames and engine states against the oracle."""
    desc = _lit_room(n_lights, emitters)
    cam = scenes.camera((0, 0, -3), (0, 0, 0), (0, 1, 0), 1.0, 1.0, -1.0)
    for mode in ("host", "device"):
        scene = _scene_with(mode, desc)
        try:
            n_emis = scene.info()["n_emissive"]
            assert n_lights + min(2 + int(np.log10(n_emis + 1)), n_emis) == expect_samples
            st, c = _pixel_job(scene, oracle_lib.scene_create(desc), cam, scenes.options(w, h, mn, mx))
            assert st["shadow_rays_traced"] > 0 and st["samples"] == c["samples"]
            ys, xs = np.mgrid[0:h:3, 0:w:3]
            xs, ys = xs.ravel().astype(np.int32), ys.ravel().astype(np.int32)
            states = np.arange(1, len(xs) + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
            opt = scenes.options(w, h, mn, mx)
            img, after = scene.process_item(cam, opt, binding.pixel_streams(xs, ys, states))
            want, want_after = oracle_lib.scene_create(desc).render_streams(cam, opt, oracle.pixel_streams(xs, ys, states), n_threads=8)
            assert_bits_equal(img, want, name + ": streams")
            assert_bits_equal(after, want_after, name + ": engine states")
        finally:
            scene.close()


def test_more_than_32_light_samples_are_refused():
    """The remaining limit (include/pt_hip.h): 33 samples per vertex do not fit the visibility mask; the scene is refused
    when it is created, with a message that says why -- never rendered wrongly."""
    desc = _lit_room(31, _three_emitters)  # 31 + 2
    with pytest.raises(binding.PtError) as e:
        binding.Scene(desc)
    assert e.value.code == 4 and "more than 32 light samples" in str(e.value) and "31 point lights + 2 emitter samples" in str(e.value)


@pytest.mark.gpu
def test_frame_independent_of_first_round_placement(gpu_scenes, sset):
    """The diagnostics behind DESIGN.md 5.1's placement experiments: a first round of streams made by the caller (pt_debug_set_place: which
    stream starts in which slot of which wavefront, with empty slots) must render the same frame bit for bit as the library's own, and the
    cost counters (pt_debug_collect_costs: wave steps while a ray of the stream walks) must cover every stream."""
    import ctypes as C
    lib = binding.load()
    desc, cam = sset["meshbox"]
    gpu = gpu_scenes("meshbox")
    w = h = 96
    opt = scenes.options(w, h, 6, 6)
    want, _ = gpu.process_job(cam, opt, base_seed=77, want_stats=True)
    want = want.copy()
    n = w * h
    rng = np.random.default_rng(5)
    waves, slots = 512, 40
    table = np.full(waves * slots, 0xFFFFFFFF, np.uint32)
    table[rng.permutation(waves * slots)[:n]] = rng.permutation(n).astype(np.uint32)  # any stream anywhere, more than half of the slots empty
    binding._check(lib.pt_debug_collect_costs(gpu._h, 1))
    try:
        binding._check(lib.pt_debug_set_place(gpu._h, C.c_uint32(waves), C.c_uint32(slots), table.ctypes.data_as(C.c_void_p)))
        got, st = gpu.process_job(cam, opt, base_seed=77, want_stats=True)
        assert_bits_equal(got, want, "frame under a caller-made first round")
        assert st["wavefronts"] == waves and st["samples"] == 6 * n
        costs = np.zeros(n, np.uint32)
        binding._check(lib.pt_debug_stream_costs(gpu._h, costs.ctypes.data_as(C.c_void_p), C.c_size_t(n)))
        assert (costs > 0).mean() > 0.99, "(nearly) every stream walked a ray for at least one step"
        # (the table is used once: the next call is the library's own first round again)
        again, st2 = gpu.process_job(cam, opt, base_seed=77, want_stats=True)
        assert_bits_equal(again, want, "frame after the table was used")
    finally:
        binding._check(lib.pt_debug_collect_costs(gpu._h, 0))


@pytest.mark.gpu
def test_every_device_of_the_host_renders_its_tiles(sset):
    """pt_render_tiles_multi with one scene PER DEVICE (what $PATHTRACE_DEVICES / doWorkParallel over GPUs does, src/host/scene.cpp): the
    frame equals the single-device frame and every device renders samples.  Needs two GPUs in one process: skipped on a one-GPU box
    (there the same code path runs with two replicas on device 0, test_progress_and_replicas); multi-GPU hardware runs are the driver's."""
    n_dev = binding.device_count()
    if n_dev < 2:
        pytest.skip("one HIP device on this host")
    desc, cam = sset["meshbox"]
    opt = scenes.options(256, 192, 8, 8)
    replicas = [binding.Scene(desc, device=d) for d in range(min(n_dev, 8))]
    try:
        want = replicas[0].process_job(cam, opt, base_seed=31)
        img, stats = binding.process_job_multi(replicas, cam, opt, base_seed=31, want_stats=True)
        assert_bits_equal(img, want, "frame rendered by %d devices" % len(replicas))
        assert sum(s["samples"] for s in stats) == 256 * 192 * 8 and all(s["samples"] > 0 for s in stats)
    finally:
        for r in replicas:
            r.close()


def _scene_with_env(desc, **env):
    """binding.Scene(desc) under the given environment (the library reads its knobs when a scene is created)."""
    import os
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return binding.Scene(desc)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


@pytest.mark.gpu
def test_compacted_passes_render_the_same_frame(oracle_lib):
    """Adaptive sampling thins out the slots of every wavefront (pixels stop at different samples and a 1024 x 1024 job has no stream left to
    take their place); a shading pass then runs over a LIST of the ready slots instead of the rows they sit in (pt_path.hip, the pass loop).
    Which lane shades which slot must not matter: the frame equals the one rendered row by row (PT_COMPACT=0) bit for bit, both with the
    tree in HBM and with the scene in LDS, and equals the oracle on sampled pixels."""
    cases = [("mesh80k", scenes.dragon_box_scene(*scenes.bumpy_sphere_mesh(200, 200, scenes.DRAGON_BOX_TRANSFORM)), scenes.options(1024, 1024, 4, 24)),
             ("box", scenes.box_scene(), scenes.options(1024, 1024, 8, 40)),
             # 14 light samples per vertex: the kernel with the 64-bit slot word (its waiting mask lives with the slot's state in HBM)
             ("lit room", (_lit_room(12, _three_emitters), scenes.camera((0, 0, -3), (0, 0, 0), (0, 1, 0), 1.0, 1.0, -1.0)), scenes.options(1024, 1024, 3, 12))]
    for name, (desc, cam), opt in cases:
        by_rows = _scene_with_env(desc, PT_COMPACT=0)
        by_list = _scene_with_env(desc, PT_COMPACT=1)
        handle = oracle_lib.scene_create(desc)
        try:
            want, st_rows = by_rows.process_job(cam, opt, base_seed=91, want_stats=True)
            want = want.copy()
            assert st_rows["samples"] < 1024 * 1024 * opt["max_sample_count"], "%s: no pixel stopped early, the case tests nothing" % name
            got, st_list = by_list.process_job(cam, opt, base_seed=91, want_stats=True)
            assert_bits_equal(got, want, "%s: frame of compacted passes against row-by-row passes" % name)
            assert st_list["samples"] == st_rows["samples"] and st_list["vertices"] == st_rows["vertices"]
            rng = np.random.default_rng(17)
            xs, ys = rng.integers(0, 1024, 1000).astype(np.int32), rng.integers(0, 1024, 1000).astype(np.int32)
            states = np.array([binding.seed_to_state(binding.pixel_seed(91, int(x), int(y))) for x, y in zip(xs, ys)], np.uint64)
            ref, _ = handle.render_streams(cam, opt, oracle.pixel_streams(xs, ys, states), n_threads=16)
            assert_bits_equal(got[ys, xs], ref[ys, xs], "%s: sampled pixels against the oracle" % name)
        finally:
            by_rows.close()
            by_list.close()
            handle.close()


@pytest.mark.gpu
@pytest.mark.parametrize("env", [
    {"PT_BURST": 1, "PT_LEAF_MIN": 64},
    {"PT_REFILL_IDLE": 64, "PT_MIN_READY": 512},
    {"PT_REFILL_IDLE": 1, "PT_MIN_READY": 1, "PT_READY_SHIFT": 3},
    {"PT_PASS_Q_LOW": 64, "PT_EARLY_READY": 48},        # passes that start while the ring still holds rays
    {"PT_PASS_Q_LOW": 1000, "PT_EARLY_READY": 1, "PT_BURST": 3},
    {"PT_DEBUG_LANES": 5, "PT_REFILL_IDLE": 2},         # five lanes of a wavefront take rays
    {"PT_ROWS": 2, "PT_BLOCKS_PER_CU": 1},
    {"PT_ROWS": 3, "PT_FIRST_LANES": 4, "PT_COMPACT": 0},
], ids=lambda e: ",".join("%s=%s" % (k[3:].lower(), v) for k, v in e.items()))
def test_scheduler_knobs_do_not_change_the_frame(env):
    """How a wavefront schedules its work -- burst length, refill and pass thresholds, early passes, rows of slots, lanes that take rays,
    compacted passes -- decides WHEN a stream's next draw is made, never which: a 1024 x 1024 adaptive frame (several rows of slots per
    wavefront, slots that die at different samples) is the same bit for bit under extreme settings, with the tree in HBM and in LDS."""
    cases = [("meshbox", scenes.dragon_box_scene(*scenes.bumpy_sphere_mesh(24, 24, scenes.DRAGON_BOX_TRANSFORM)), scenes.options(1024, 1024, 3, 14)),
             ("box", scenes.box_scene(), scenes.options(1024, 1024, 4, 12))]
    for name, (desc, cam), opt in cases:
        for lds_small in (1, 0):
            plain = _scene_with_env(desc, PT_LDS_SMALL=lds_small)
            tuned = _scene_with_env(desc, PT_LDS_SMALL=lds_small, **env)
            try:
                want, st0 = plain.process_job(cam, opt, base_seed=55, want_stats=True)
                want = want.copy()
                got, st1 = tuned.process_job(cam, opt, base_seed=55, want_stats=True)
                assert_bits_equal(got, want, "%s (small trees %s): frame under %s" % (name, "in LDS" if lds_small else "in HBM", env))
                assert st1["samples"] == st0["samples"] and st1["vertices"] == st0["vertices"] and st1["rays_traced"] == st0["rays_traced"]
            finally:
                plain.close()
                tuned.close()
