// pt_api.cpp -- host side of libpathtrace_hip.so: the C ABI of include/pt_hip.h.
//
// Scene creation flattens the caller's object list into the HBM layout of pt_types.h (building the reference's BVH
// topology on the way, pt_bvh.cpp / pt_build.hip); a render call is ONE launch of the persistent path kernel (pt_path.hip), whose
// wavefronts render the call's streams until none is left.  There is no CPU rendering path in this library.
#include "../../include/pt_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "pt_build.h"
#include "pt_bvh.h"
#include "pt_kernels.h"
#include "pt_post.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

#define PT_HIP(call)                                                                                               \
    do {                                                                                                           \
        hipError_t err_ = (call);                                                                                  \
        if(err_ != hipSuccess) {                                                                                   \
            return fail(PT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(err_));                          \
        }                                                                                                          \
    } while(0)

int env_int(const char *name, int fallback) {
    const char *v = std::getenv(name);
    return (v != nullptr && *v != '\0') ? std::atoi(v) : fallback;
}

inline float fmin_std(float a, float b) {
    return (b < a) ? b : a;
}
inline float fmax_std(float a, float b) {
    return (a < b) ? b : a;
}

struct Vec3 {
    float x, y, z;
};
inline Vec3 sub(Vec3 a, Vec3 b) {
    return {a.x - b.x, a.y - b.y, a.z - b.z};
}
inline Vec3 scale(Vec3 a, float f) {
    return {a.x * f, a.y * f, a.z * f};
}
inline float dot(Vec3 a, Vec3 b) {
    float d = 0.0F;
    d += a.x * b.x;
    d += a.y * b.y;
    d += a.z * b.z;
    return d;
}
inline Vec3 cross(Vec3 a, Vec3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline Vec3 normalize(Vec3 a) {
    const float inv = 1.0F / std::sqrt(dot(a, a));
    return scale(a, inv);
}
inline Vec3 ld(const float *p) {
    return {p[0], p[1], p[2]};
}

uint32_t bits(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}
float from_bits(uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

template<typename T>
struct DevBuf {
    T *ptr = nullptr;
    size_t count = 0;
    ~DevBuf() { release(); }
    void release() {
        if(ptr != nullptr) {
            (void)hipFree(ptr);
            ptr = nullptr;
            count = 0;
        }
    }
    hipError_t ensure(size_t n) {
        if(n <= count && ptr != nullptr) {
            return hipSuccess;
        }
        release();
        const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&ptr), bytes);
        if(e == hipSuccess) {
            count = std::max<size_t>(n, 1);
        }
        return e;
    }
    hipError_t upload(const std::vector<T> &host) {
        hipError_t e = ensure(host.size());
        if(e != hipSuccess || host.empty()) {
            return e;
        }
        return hipMemcpy(ptr, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice);
    }
};

struct F4 {
    float x, y, z, w;
};

} // namespace

struct pt_scene {
    int device = 0;
    hipStream_t stream = nullptr;
    int cu_count = 256;

    // host copies kept for introspection and for mapping references back to object indices
    ptb::Tree tree;          // host-built scenes only (PT_BUILD=host or few objects); empty when the device built the tree
    uint64_t n_nodes = 0;    // 2 * n_objects - 1
    uint32_t depth = 0;      // levels of the tree (a single leaf has depth 1)
    bool device_built = false;
    float build_ms[4] = {0, 0, 0, 0}; // host preparation, upload, device tree construction, emissive registration + rest
    std::vector<uint32_t> tri_obj;
    std::vector<uint32_t> sph_obj;
    uint32_t n_objects = 0;
    uint32_t n_emissive = 0;
    std::vector<int32_t> emissive_obj;
    std::vector<float> emissive_cdf;

    // device scene
    DevBuf<F4> recs, pairs, tris, tri_shade, spheres, materials, lights, emis; // (pairs and tris only while the scene is being built: linked into recs)
    DevBuf<uint2> sph_meta;
    DevBuf<float> emis_cdf;
    PtDevScene dev{};

    DevBuf<PtDevCounters> counters;
    DevBuf<F4> image;
    DevBuf<int4> tiles;
    DevBuf<uint32_t> tile_offset;
    DevBuf<float> batch_rays;

    // One render call at a time per scene: the workspace below is shared by every entry point (processItem may be called from several
    // threads on one const Scene, reference worker.h:66-69 / src/worker.cpp:328-362: such callers are serialised here).
    std::mutex render_mutex;

    // workspace of the persistent path kernel (pt_path.hip), grown on demand and reused between calls
    PtPathConfig path_cfg{};
    int path_blocks_per_cu = 0;
    uint32_t path_slots = 0, path_waves = 0, path_cap = 0;
    DevBuf<uint32_t> sl_stream, sl_nee_mask, pull_counter, tile_left;
    DevBuf<int4> sl_rect, st_rect;
    DevBuf<int32_t> sl_cursor, sl_path_length;
    DevBuf<uint64_t> sl_rng, st_rng;
    DevBuf<F4> sl_ray_o, sl_ray_d, sl_spectrum, sl_out, sl_nee, lq_ray_o, lq_ray_d;
    DevBuf<double> sl_divisor, sl_bounce_pd;
    DevBuf<PtEstimator> sl_est;
    DevBuf<PtCandidate> sl_cand;
    DevBuf<uint2> path_spill, closest_out;
    DevBuf<uint32_t> walk_save;
    DevBuf<PtPathArgs> path_args;  // the kernel's arguments in device memory
    PtPathArgs host_path_args{};   // ... and the host copy they are uploaded from
    DevBuf<unsigned long long> path_wave_counters;
    uint32_t *host_tiles_done = nullptr; // pinned: tiles finished so far, written by the kernel (progress callback)
    unsigned long long *host_streams_done = nullptr; // pinned: the launch's count of finished streams, copied behind every launch
    uint64_t streams_expected = 0;                   // ... and what it must read once the stream has drained (finish_path)
    // cost-aware placement (render_tiles_impl): what every stream of the pilot launch cost, and the stream every slot of the main launch starts with
    DevBuf<uint32_t> sl_cost, stream_cost, place;
    bool debug_collect_costs = false;                // pt_debug_collect_costs: every launch records them
    std::vector<uint32_t> debug_place;               // pt_debug_set_place: the next launch starts from this table ...
    uint32_t debug_place_waves = 0, debug_place_slots = 0; // ... with this many wavefronts and slots in each

    ~pt_scene() {
        if(host_tiles_done != nullptr) {
            (void)hipHostFree(host_tiles_done);
        }
        if(host_streams_done != nullptr) {
            (void)hipHostFree(host_streams_done);
        }
        if(stream != nullptr) {
            (void)hipStreamDestroy(stream);
        }
    }
};

namespace {

int device_count_quiet() {
    int n = 0;
    if(hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

PtDevCamera derive_camera(const pt_camera_params *c) {
    // Camera::Camera, src/camera.cpp:53-76
    PtDevCamera cam{};
    const Vec3 origin = ld(c->origin);
    const Vec3 forward_dir = normalize(sub(ld(c->look_at), origin));
    const Vec3 forward = scale(forward_dir, c->focal_length);
    const Vec3 up_dir = normalize(ld(c->up));
    const float height_half = c->height / 2.0F;
    const Vec3 up = scale(up_dir, height_half);
    const Vec3 right_dir = normalize(cross(forward, up));
    const float width_half = height_half * c->aspect_ratio;
    const Vec3 right = scale(right_dir, width_half);
    const Vec3 v[4] = {origin, forward, up, right};
    float *dst[4] = {cam.origin, cam.forward, cam.up, cam.right};
    for(int i = 0; i < 4; i++) {
        dst[i][0] = v[i].x;
        dst[i][1] = v[i].y;
        dst[i][2] = v[i].z;
    }
    cam.aperture_width_half = c->aperture_width / 2.0F;
    cam.aperture_height_half = c->aperture_height / 2.0F;
    cam.aperture_kind = c->aperture_kind;
    cam.hex_ratio = fmin_std(fmax_std(c->hex_ratio, 0.0F), 1.0F); // camera.cpp:22-24
    cam.focal_plane_dist = c->focal_plane_dist;
    return cam;
}

int derive_options(const pt_options *o, PtDevOptions *out) {
    PtDevOptions d{};
    d.image_width = o->image_width;
    d.image_height = o->image_height;
    d.min_sample_count = o->min_sample_count;
    d.max_sample_count = o->max_sample_count;
    d.epsilon = o->epsilon;
    d.pixel_width = 1.0F / static_cast<float>(o->image_width);
    d.pixel_height = 1.0F / static_cast<float>(o->image_height);
    // worker.cpp:158-164
    d.stats_sample_count = std::min(std::max(o->min_sample_count / 4, 1), 64);
    d.candidate_batch_count = std::max(std::max(o->min_sample_count, o->max_sample_count / 4) / d.stats_sample_count, 2);
    d.check_sample_count =
      std::min(std::max({o->min_sample_count / 2, (o->max_sample_count - o->min_sample_count) / 8, 8, d.stats_sample_count}), 1024) / d.stats_sample_count;
    // closed candidates a pixel can accumulate (worker.cpp:214-222).  A candidate closes after candidate_batch_count >= max / (4 S) batches
    // and a pixel has at most max / S of them, so at most 4 candidates ever close: the check below is a guard against a change of the
    // formulas above, not a limit a caller can reach (PT_MAX_CANDIDATES = 8).
    const int batches = std::max(o->max_sample_count, 0) / d.stats_sample_count;
    const int closed = batches > 0 ? (batches - 1) / d.candidate_batch_count : 0;
    if(closed > PT_MAX_CANDIDATES) {
        return fail(PT_ERR_UNSUPPORTED, "internal: the estimator's constants allow more than 8 closed candidates per pixel (worker.cpp:158-164 give at most 4)");
    }
    *out = d;
    return PT_OK;
}

int check_render_args(pt_scene *scene, const pt_camera_params *camera, const pt_options *options) {
    if(scene == nullptr || camera == nullptr || options == nullptr) {
        return fail(PT_ERR_INVALID, "null argument");
    }
    if(options->image_width <= 0 || options->image_height <= 0) {
        return fail(PT_ERR_INVALID, "image size must be positive");
    }
    return PT_OK;
}

// ---- the persistent path kernel (pt_path.hip): one launch per render call ---------------------------------------------------------------

struct Event {
    hipEvent_t e = nullptr;
    hipError_t create(unsigned flags = hipEventDefault) { return hipEventCreateWithFlags(&e, flags); }
    ~Event() {
        if(e != nullptr) {
            (void)hipEventDestroy(e);
        }
    }
};


int setup_path(pt_scene *s) {
    PtPathConfig &cfg = s->path_cfg;
    if(cfg.rows != 0) {
        return PT_OK;
    }
    cfg.in_lds = (s->dev.n_lds_pairs == s->dev.n_pairs && s->dev.n_lds_tris == s->dev.n_tris && s->dev.n_lds_pairs + s->dev.n_lds_tris > 0) ? 1 : 0;
    // 8 stack entries per lane in LDS (16 KB per workgroup) let four workgroups share a CU; deeper walks use the HBM spill area.  A scene
    // staged in LDS that leaves no room for four workgroups that way gets a window of 4 entries (pt_path.hip, PT_PATH_STACK_LDS_SMALL)
    cfg.wide = s->dev.n_lights + s->dev.n_object_samples > 8U ? 1 : 0;
    cfg.rows = std::min(std::max(env_int("PT_ROWS", 4), 1), PT_MAX_ROWS);
    cfg.stack_lds = pt_path_stack_lds(cfg.in_lds, pt_path_lds_bytes(cfg.wide, cfg.rows, 8, cfg.in_lds ? s->dev.n_lds_pairs : 0U, cfg.in_lds ? s->dev.pair_base : 0U));
    if(cfg.in_lds && env_int("PT_STACK_WINDOW", 0) > 0) {
        cfg.stack_lds = env_int("PT_STACK_WINDOW", 0) <= 4 ? 4 : 8; // (A/B: force the window of a scene in LDS)
    }
    cfg.lds_bytes = pt_path_lds_bytes(cfg.wide, cfg.rows, cfg.stack_lds, cfg.in_lds ? s->dev.n_lds_pairs : 0U, cfg.in_lds ? s->dev.pair_base : 0U);
    const int per_cu = pt_path_blocks_per_cu(cfg);
    const int limit = env_int("PT_BLOCKS_PER_CU", 0);
    s->path_blocks_per_cu = (limit > 0 && limit < per_cu) ? limit : per_cu;
    // a walk's stack holds at most one parked node per level of the tree and the sentinel at its bottom (pt_path.hip); what does not fit the LDS window spills
    cfg.spill_depth = s->depth + 2U > static_cast<uint32_t>(cfg.stack_lds) ? s->depth + 2U - static_cast<uint32_t>(cfg.stack_lds) : 1U;
    cfg.refill_idle = std::min(std::max(env_int("PT_REFILL_IDLE", 12), 1), 64);
    cfg.min_ready = std::min(std::max(env_int("PT_MIN_READY", 32), 1), 64 * PT_MAX_ROWS);
    cfg.ready_shift = std::min(std::max(env_int("PT_READY_SHIFT", 1), 0), 31);
    cfg.pass_q_low = std::max(env_int("PT_PASS_Q_LOW", 0), 0);
    cfg.early_ready = std::min(std::max(env_int("PT_EARLY_READY", 0), 0), 64 * PT_MAX_ROWS);
    cfg.compact_passes = env_int("PT_COMPACT", 1) != 0 ? 1 : 0;
    cfg.debug_lanes = std::min(std::max(env_int("PT_DEBUG_LANES", 64), 1), 64);
    // (burst_steps and leaf_min depend on the job's size as well: ensure_path_workspace sets them per job and keeps the last job's here)
    cfg.burst_steps = 24;
    cfg.leaf_min = 8;
    if(env_int("PT_DEBUG", 0) != 0) {
        std::fprintf(stderr, "[pt] path kernel: %d CUs x %d workgroups, %d rows of slots per wavefront, stack_lds %d, scene %s, lds %zu B, spill depth %u\n", s->cu_count,
                     s->path_blocks_per_cu, cfg.rows, cfg.stack_lds, cfg.in_lds ? "in LDS" : "in HBM", cfg.lds_bytes, cfg.spill_depth);
    }
    return PT_OK;
}

// A first round chosen by the host instead of the kernel's arithmetic: `waves` wavefronts (a multiple of 4) with `slots_per_wave` slots each,
// slot q of wavefront w starting with stream place[w * slots_per_wave + q] (device memory; 0xffffffff = the slot stays empty).
struct PathPlan {
    uint32_t waves = 0, slots_per_wave = 0;
    const uint32_t *d_place = nullptr;
};

// Grid and slot rows for n streams, and the buffers they need.
int ensure_path_workspace(pt_scene *s, uint32_t n, PtPathConfig *out_cfg, const PathPlan *plan = nullptr) {
    int rc = setup_path(s);
    if(rc != PT_OK) {
        return rc;
    }
    PtPathConfig cfg = s->path_cfg;
    const uint32_t max_grid = static_cast<uint32_t>(s->cu_count) * static_cast<uint32_t>(s->path_blocks_per_cu);
    // A stream's samples are sequential, so only more streams in flight shorten a job: a small job is spread over `spread` wavefronts
    // (a few per CU: enough to hide latency, few enough that a traversal step still serves many walks) before any wavefront gets a
    // full row of 64 slots; a large one fills the rows of every wavefront the chip holds.
    const uint32_t spread = std::min<uint32_t>(max_grid * 4U, static_cast<uint32_t>(std::max(env_int("PT_SPREAD_WAVES", 1024), 4)));
    uint32_t waves_wanted = (n + 63U) / 64U;                       // one row each
    if(waves_wanted < spread) {
        waves_wanted = std::min<uint32_t>(spread, n);              // thin rows
    }
    uint32_t grid = std::max<uint32_t>(1U, std::min<uint32_t>(max_grid, (waves_wanted + 3U) / 4U));
    if(plan != nullptr) {
        grid = std::max<uint32_t>(1U, std::min<uint32_t>(max_grid, plan->waves / 4U));
    }
    const uint32_t waves = grid * 4U;
    uint32_t slots_per_wave = std::min<uint32_t>(static_cast<uint32_t>(cfg.rows) * 64U, std::max<uint32_t>(1U, (n + waves - 1U) / waves));
    if(plan != nullptr) {
        if(waves != plan->waves || plan->slots_per_wave == 0 || plan->slots_per_wave > static_cast<uint32_t>(cfg.rows) * 64U) {
            return fail(PT_ERR_INVALID, "placement: " + std::to_string(plan->waves) + " wavefronts x " + std::to_string(plan->slots_per_wave) + " slots do not fit this device");
        }
        slots_per_wave = plan->slots_per_wave;
    }
    // The first round of streams goes to the slots in pieces of `first_lanes` neighbouring slots (pt_path.hip, stream hand-out): a
    // wavefront's slots are a whole number of pieces (a large job gets up to 7 more slots per wavefront, a small one pieces of 1).
    // A job that fits the slots in ONE round (nothing left to pull: every strong-scaling share of a frame, every small frame) has no
    // dynamic balance at all, and its duration is that of the wavefront with the most expensive streams -- whose samples are sequential, so
    // the streams that happen to share a wavefront with them wait for the same passes.  Such a job is dealt stream by stream (pieces of 1:
    // slot q of wavefront w renders stream q * waves + w), which gives every wavefront a sample of the whole job: the 1/8 share of the
    // benchmark frame 273 -> 221 ms at 256 spp, the 1/4 share 317 -> 259 (profiles/r03_share_rehearsal.txt).
    const bool single_round = plan != nullptr || (static_cast<uint64_t>(waves) * slots_per_wave >= n && slots_per_wave <= 128U); // (a full grid of 4 rows balances well in pieces of 8: 423 against 409 Msamples/s)
    uint32_t first_lanes = plan != nullptr ? 1U : static_cast<uint32_t>(env_int("PT_FIRST_LANES", single_round ? 1 : 8)); // full frame: 64 -> 402, 32 -> 403, 16 -> 434, 8 -> 440, 4 -> 431 Msamples/s
    if(first_lanes == 0 || first_lanes > 64 || (first_lanes & (first_lanes - 1U)) != 0) {
        first_lanes = single_round ? 1 : 8;
    }
    if(slots_per_wave % first_lanes != 0) {
        if(slots_per_wave >= 64U) {
            slots_per_wave = (slots_per_wave + first_lanes - 1U) / first_lanes * first_lanes; // (rows * 64 is a multiple of every piece size)
        }
        else {
            first_lanes = 1;
        }
    }
    cfg.first_lanes = static_cast<int>(first_lanes);
    // Steps between two looks at the ring.  Trees in HBM: 8 -> 397, 12 -> 407, 16 -> 412, 24 -> 422, 32 -> 421 Msamples/s on the benchmark frame
    // (round 3 made the step cheaper, looking at the ring costs what it did); scenes in LDS keep 12 on a full grid (Cornell: 700 against 659
    // with 24) and take 24 when a wavefront has less than a row of slots (the reference's benchmark program, 128 x 128: 136 -> 176 Msamples/s).
    // Shallower trees in HBM have shorter walks, and looking at the ring more often pays again (profiles/r03_tree_size_knobs.txt): 160-330
    // triangles (10, 11 levels) 12 -> 795 / 764 against 788 / 756 with 24; 3 K (14 levels) 16 -> 662 against 610 (one and two rows of slots:
    // 522 against 502, 621 against 575); 20 K (17 levels) 16 -> 589 against 564; from 180 K (22 levels) on 24 wins.  Scenes in LDS: 12 with
    // several rows of slots (Cornell 1024 x 1024: 700 against 659, 724 x 724: 656 against 635), 24 with one (256 x 256: 221 against 218, Box 406 against 373).
    int burst_default = 24;
    if(cfg.in_lds) {
        burst_default = slots_per_wave > 64U ? 12 : 24;
    }
    else if(slots_per_wave >= 64U) {
        burst_default = s->depth <= 12U ? 12 : (s->depth <= 18U ? 16 : 24);
    }
    cfg.burst_steps = std::min(std::max(env_int("PT_BURST", burst_default), 1), 64);
    // Lanes that wait for the rare step (leaves) before it runs: 2 -> 374, 4 -> 396, 8 -> 414, 12 -> 415 Msamples/s on the benchmark frame; a
    // wavefront with 16 slots cannot wait for 8 of them (128 x 128, 180 k triangles: 8 -> 54, 4 -> 59, 2 -> 62 Msamples/s)
    // Scenes in LDS (a leaf test is a larger share of a walk of 7-10 nodes): 8 -> 685 / 1160, 16 -> 724 / 1201, 24 -> 729 / 1193, 32 -> 707 / 1189 Msamples/s on
    // Cornell / Box with full rows (profiles/r03_lds_scene_knobs.txt); wavefronts with less than a row of slots keep 8
    // trees in HBM of up to 24 levels (720 K triangles) with full rows: 12 instead of 8 brings 1-4 % (3 K triangles 599 -> 610, 20 K 552 -> 564, 180 K 523 -> 530,
    // 720 K 483 -> 488); the benchmark's 30 levels keep 8 (430 against 425)
    int leaf_default = cfg.in_lds ? (slots_per_wave >= 64U ? 16 : 8) : static_cast<int>(std::min<uint32_t>(std::max<uint32_t>(slots_per_wave / 8U, 2U), 8U));
    if(!cfg.in_lds && slots_per_wave >= 64U && s->depth <= 24U) {
        leaf_default = 12;
    }
    cfg.leaf_min = std::min(std::max(env_int("PT_LEAF_MIN", leaf_default), 1), 64);
    const uint32_t rows = (slots_per_wave + 63U) / 64U;
    const uint32_t total = waves * rows * 64U;
    const uint32_t rays_per_slot = 1U + s->dev.n_lights + s->dev.n_object_samples;
    uint32_t cap = rows * 64U * rays_per_slot;
    const int ring_log = env_int("PT_RING_LOG_RAYS", 0); // diagnostic: rings that never wrap keep every ray of the frame (pt_debug_replay_rays)
    if(ring_log > 0) {
        cap = std::max<uint32_t>(cap, static_cast<uint32_t>(ring_log));
    }
    cfg.grid = static_cast<int>(grid);
    cfg.rows = static_cast<int>(rows);
    cfg.slots_per_wave = static_cast<int>(slots_per_wave);
    PT_HIP(s->sl_stream.ensure(total));
    PT_HIP(s->sl_rect.ensure(total));
    PT_HIP(s->sl_cursor.ensure(total));
    PT_HIP(s->sl_rng.ensure(total));
    PT_HIP(s->sl_ray_o.ensure(total));
    PT_HIP(s->sl_ray_d.ensure(total));
    PT_HIP(s->sl_spectrum.ensure(total));
    PT_HIP(s->sl_out.ensure(total));
    PT_HIP(s->sl_divisor.ensure(total));
    PT_HIP(s->sl_bounce_pd.ensure(total));
    PT_HIP(s->sl_path_length.ensure(total));
    PT_HIP(s->sl_nee.ensure(static_cast<size_t>(total) * std::max<uint32_t>(rays_per_slot - 1U, 1U)));
    PT_HIP(s->sl_nee_mask.ensure(total));
    PT_HIP(s->sl_cost.ensure(total));
    PT_HIP(s->sl_est.ensure(total));
    PT_HIP(s->sl_cand.ensure(static_cast<size_t>(total) * PT_MAX_CANDIDATES));
    PT_HIP(s->lq_ray_o.ensure(static_cast<size_t>(waves) * cap));
    PT_HIP(s->lq_ray_d.ensure(static_cast<size_t>(waves) * cap));
    if(ring_log > 0) {
        PT_HIP(hipMemsetAsync(s->lq_ray_d.ptr, 0xff, static_cast<size_t>(waves) * cap * 4 * sizeof(float), s->stream));
    }
    PT_HIP(s->path_spill.ensure(static_cast<size_t>(waves) * 64U * cfg.spill_depth));
    PT_HIP(s->path_wave_counters.ensure(static_cast<size_t>(waves) * 8U));
    PT_HIP(s->walk_save.ensure(static_cast<size_t>(waves) * 64U * PT_WALK_SAVE_WORDS));
    PT_HIP(s->pull_counter.ensure(64));
    PT_HIP(s->counters.ensure(1));
    cfg.spill = s->path_spill.ptr;
    cfg.walk_save = s->walk_save.ptr;
    cfg.wave_counters = s->path_wave_counters.ptr;
    s->path_slots = total;
    s->path_waves = waves;
    s->path_cap = cap;
    s->path_cfg.burst_steps = cfg.burst_steps; // (the diagnostics that follow a render -- pt_debug_replay_rays -- run with its settings)
    s->path_cfg.leaf_min = cfg.leaf_min;
    *out_cfg = cfg;
    return PT_OK;
}

// Wait for the scene's stream and make sure the last launch rendered every stream it was given: a wavefront that left early or a stream
// lost in the hand-out would otherwise return stale pixels with PT_OK.  Every entry point that synchronises anyway ends with this.
int finish_path(pt_scene *s) {
    PT_HIP(hipStreamSynchronize(s->stream));
    if(s->host_streams_done != nullptr && s->streams_expected != 0) {
        const unsigned long long done = *static_cast<volatile unsigned long long *>(s->host_streams_done);
        const uint64_t expected = s->streams_expected;
        s->streams_expected = 0;
        if(done != expected) {
            return fail(PT_ERR_HIP, "path kernel ended with " + std::to_string(done) + " of " + std::to_string(expected) + " streams finished");
        }
    }
    return PT_OK;
}

// Render the streams described by T (device pointers) with one launch on the scene's stream.  With a progress function the host polls
// the count of finished tiles (pinned memory, written by the kernel) while the launch runs and reports every step from the calling thread.
int run_path(pt_scene *s, const PtDevCamera &cam, const PtDevOptions &opt, PtStreams T, float4 *d_image, pt_stats *stats, pt_progress_fn progress, void *progress_user,
             const PathPlan *plan = nullptr, bool want_costs = false) {
    PtPathConfig cfg;
    hipStream_t st = s->stream;
    PathPlan debug_plan;
    if(plan == nullptr && !s->debug_place.empty()) {
        // pt_debug_set_place: this launch only
        PT_HIP(s->place.ensure(s->debug_place.size()));
        PT_HIP(hipMemcpyAsync(s->place.ptr, s->debug_place.data(), s->debug_place.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        PT_HIP(hipStreamSynchronize(st));
        debug_plan.waves = s->debug_place_waves;
        debug_plan.slots_per_wave = s->debug_place_slots;
        debug_plan.d_place = s->place.ptr;
        s->debug_place.clear();
        plan = &debug_plan;
    }
    int rc = ensure_path_workspace(s, T.n, &cfg, plan);
    if(rc != PT_OK) {
        return rc;
    }
    want_costs = want_costs || s->debug_collect_costs;
    T.place = plan != nullptr ? plan->d_place : nullptr;
    T.cost = nullptr;
    if(want_costs) {
        PT_HIP(s->stream_cost.ensure(std::max<uint32_t>(T.n, 1U)));
        T.cost = s->stream_cost.ptr;
    }
    PtSlots S{};
    S.total = s->path_slots;
    S.stream = s->sl_stream.ptr;
    S.rect = s->sl_rect.ptr;
    S.cursor = s->sl_cursor.ptr;
    S.rng = s->sl_rng.ptr;
    S.ray_o = reinterpret_cast<float4 *>(s->sl_ray_o.ptr);
    S.ray_d = reinterpret_cast<float4 *>(s->sl_ray_d.ptr);
    S.spectrum = reinterpret_cast<float4 *>(s->sl_spectrum.ptr);
    S.out = reinterpret_cast<float4 *>(s->sl_out.ptr);
    S.divisor = s->sl_divisor.ptr;
    S.bounce_pd = s->sl_bounce_pd.ptr;
    S.path_length = s->sl_path_length.ptr;
    S.nee = reinterpret_cast<float4 *>(s->sl_nee.ptr);
    S.nee_mask = s->sl_nee_mask.ptr;
    S.cost = s->sl_cost.ptr;
    S.est = s->sl_est.ptr;
    S.cand = s->sl_cand.ptr;
    PtLocalQueue Q{};
    Q.ray_o = reinterpret_cast<float4 *>(s->lq_ray_o.ptr);
    Q.ray_d = reinterpret_cast<float4 *>(s->lq_ray_d.ptr);
    Q.cap = s->path_cap;
    T.next = s->pull_counter.ptr;
    T.first_total = plan != nullptr ? T.n : s->path_waves * static_cast<uint32_t>(cfg.slots_per_wave); // (a placement names every stream: nothing is left to pull)
    T.n_waves = s->path_waves;
    // The first round (pt_path.hip, stream hand-out): piece q of wavefront w -- `first_lanes` neighbouring slots -- starts on the chunk
    // q * waves + w of as many streams, moved q steps sideways in a regular tile grid.
    T.first_spread = env_int("PT_FIRST_SPREAD", 1) != 0 ? 1U : 0U;
    T.first_lanes = static_cast<uint32_t>(cfg.first_lanes);
    T.first_shift = static_cast<uint32_t>(std::max(env_int("PT_FIRST_SHIFT", 1), 0));
    {
        // the sideways move needs: a regular grid, a first round that does not reach beyond the job and covers whole grid rows per piece,
        // and as many tiles per grid row as a multiple of the pieces of a wavefront
        const uint32_t pieces = static_cast<uint32_t>(cfg.slots_per_wave) / T.first_lanes;
        const unsigned long long per_grid_row = static_cast<unsigned long long>(T.chunks_per_tile) * (64U / T.first_lanes) * T.tiles_per_row;
        if(env_int("PT_FIRST_SPREAD", 1) == 2 || per_grid_row == 0 || s->path_waves % per_grid_row != 0 || T.first_total > T.n || (T.tiles_per_row % pieces != 0 && pieces % T.tiles_per_row != 0)) {
            T.tiles_per_row = 0;
        }
    }
    T.tile_left = nullptr;
    T.tiles_done = nullptr;
    if(progress != nullptr && T.rect == nullptr && T.n_tiles > 0) {
        if(s->host_tiles_done == nullptr) {
            PT_HIP(hipHostMalloc(reinterpret_cast<void **>(&s->host_tiles_done), 64, hipHostMallocDefault));
        }
        *s->host_tiles_done = 0;
        T.tiles_done = s->host_tiles_done;
        T.tile_left = s->tile_left.ptr; // filled by the caller (pixels per tile)
    }
    PT_HIP(hipMemsetAsync(s->counters.ptr, 0, sizeof(PtDevCounters), st));
    PT_HIP(hipMemsetAsync(s->pull_counter.ptr, 0, 64 * sizeof(uint32_t), st));
    PT_HIP(hipMemsetAsync(s->path_wave_counters.ptr, 0, static_cast<size_t>(s->path_waves) * 8U * sizeof(unsigned long long), st));
    Event ev_begin, ev_end;
    PT_HIP(ev_begin.create());
    PT_HIP(ev_end.create());
    PT_HIP(hipEventRecord(ev_begin.e, st));
    PT_HIP(s->path_args.ensure(1));
    pt_launch_path(st, s->dev, cam, opt, S, T, Q, cfg, d_image, s->counters.ptr, &s->host_path_args, s->path_args.ptr);
    PT_HIP(hipGetLastError());
    PT_HIP(hipEventRecord(ev_end.e, st));
    // every launch leaves its count of finished streams in pinned memory; whoever waits for the stream next compares it (finish_path)
    if(s->host_streams_done == nullptr) {
        PT_HIP(hipHostMalloc(reinterpret_cast<void **>(&s->host_streams_done), 64, hipHostMallocDefault));
    }
    *s->host_streams_done = ~0ULL;
    s->streams_expected = T.n;
    PT_HIP(hipMemcpyAsync(s->host_streams_done, &s->counters.ptr->streams_done, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    if(T.tiles_done != nullptr) {
        const int total = static_cast<int>(T.n_tiles);
        int reported = 0;
        for(;;) {
            const hipError_t q = hipEventQuery(ev_end.e);
            const int done = std::min(static_cast<int>(*static_cast<volatile uint32_t *>(s->host_tiles_done)), total);
            while(reported < done) {
                progress(++reported, total, progress_user);
            }
            if(q == hipSuccess) {
                break;
            }
            if(q != hipErrorNotReady) {
                return fail(PT_ERR_HIP, std::string("path kernel: ") + hipGetErrorString(q));
            }
            std::this_thread::sleep_for(std::chrono::microseconds(500));
        }
        const int done = std::min(static_cast<int>(*static_cast<volatile uint32_t *>(s->host_tiles_done)), total);
        while(reported < done) {
            progress(++reported, total, progress_user);
        }
    }
    if(stats != nullptr) {
        PT_HIP(hipEventSynchronize(ev_end.e));
        float ms = 0.0F;
        PT_HIP(hipEventElapsedTime(&ms, ev_begin.e, ev_end.e));
        std::vector<unsigned long long> slots(static_cast<size_t>(s->path_waves) * 8U);
        PT_HIP(hipMemcpy(slots.data(), s->path_wave_counters.ptr, slots.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, kcycles[3] = {0, 0, 0};
        for(size_t i = 0; i < slots.size(); i++) {
            // (a PT_PATH_TIMING build packs kilo-cycle totals into the high words of three slots; they are zero otherwise)
            const size_t k = i & 7U;
            sum[k] += (k >= 3 && k <= 5) ? (slots[i] & 0xffffffffULL) : slots[i];
            if(k >= 3 && k <= 5) {
                kcycles[k - 3] += slots[i] >> 32;
            }
        }
        if(kcycles[0] != 0 && env_int("PT_DEBUG", 0) != 0) {
            std::fprintf(stderr, "[pt] wave time: %.1f %% in shading passes, %.1f %% in traversal bursts (of the waves' lifetimes; %llu kilo-cycles in all)\n",
                         100.0 * static_cast<double>(kcycles[2]) / static_cast<double>(kcycles[0]), 100.0 * static_cast<double>(kcycles[1]) / static_cast<double>(kcycles[0]), kcycles[0]);
        }
        PtDevCounters done{};
        PT_HIP(hipMemcpy(&done, s->counters.ptr, sizeof(done), hipMemcpyDeviceToHost));
        if(done.streams_done != T.n) {
            return fail(PT_ERR_HIP, "path kernel ended with " + std::to_string(done.streams_done) + " of " + std::to_string(T.n) + " streams finished");
        }
        stats->node_visits = sum[0];
        stats->leaf_tests = sum[1];
        stats->rays_traced = sum[2];
        stats->shadow_rays_traced = sum[3];
        stats->samples = sum[6];
        stats->vertices = sum[7];
        stats->launches = 1;
        stats->kernel_ms = ms;
        stats->wave_steps = sum[4];
        stats->shading_passes = sum[5];
        stats->wavefronts = s->path_waves;
        stats->slot_rows = static_cast<uint64_t>(cfg.rows);
        if(env_int("PT_DEBUG", 0) != 0) {
            std::fprintf(stderr, "[pt] path kernel: %.2f ms, grid %d x 256, %d rows; wave steps %llu (%.1f lanes of 64 busy per step), shading passes %llu, rays %llu\n", ms, cfg.grid,
                         cfg.rows, sum[4], sum[4] ? static_cast<double>(sum[0] + sum[1]) / static_cast<double>(sum[4]) : 0.0, sum[5], sum[2]);
            // balance: a wavefront's busy time follows its steps; the launch lasts as long as the busiest one
            std::vector<unsigned long long> steps(s->path_waves);
            for(size_t w = 0; w < steps.size(); w++) {
                steps[w] = slots[8 * w + 4] & 0xffffffffULL;
            }
            std::sort(steps.begin(), steps.end());
            const double mean = static_cast<double>(sum[4]) / static_cast<double>(steps.size());
            std::fprintf(stderr, "[pt] wave steps per wavefront: mean %.0f, min %llu, median %llu, 90 %% %llu, 99 %% %llu, max %llu (max / mean %.3f)\n", mean, steps.front(),
                         steps[steps.size() / 2], steps[steps.size() * 9 / 10], steps[steps.size() * 99 / 100], steps.back(), static_cast<double>(steps.back()) / mean);
        }
    }
    return PT_OK;
}

} // namespace

extern "C" {

int pt_device_count(void) {
    return device_count_quiet();
}

const char *pt_last_error(void) {
    return g_last_error.c_str();
}

uint64_t pt_pixel_seed(uint64_t base_seed, int32_t x, int32_t y) {
    // splitmix64 finaliser over (base_seed, x, y); the device computes the same (pt_shading.h pixel_seed)
    uint64_t z = base_seed + 0x9E3779B97F4A7C15ULL * (1ULL + (static_cast<uint64_t>(static_cast<uint32_t>(y)) << 32) + static_cast<uint64_t>(static_cast<uint32_t>(x)));
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

uint64_t pt_rng_seed_to_state(uint64_t seed) {
    return seed ^ (~seed << 32);
}

size_t pt_job_tiles(int32_t image_width, int32_t image_height, pt_tile *out, size_t capacity) {
    // processJob, src/worker.cpp:389-414
    const int width = std::max(image_width, 0);
    const int height = std::max(image_height, 0);
    if(width == 0 || height == 0) {
        return 0;
    }
    const int tile_size = std::max(std::min(std::min(width, height) / 4, 32), 1);
    const int horizontal_tiles = (width + (tile_size - 1)) / tile_size;
    const int vertical_tiles = (height + (tile_size - 1)) / tile_size;
    size_t n = 0;
    for(int ty = 0; ty < vertical_tiles; ty++) {
        for(int tx = 0; tx < horizontal_tiles; tx++) {
            if(out != nullptr && n < capacity) {
                const int ox = tx * tile_size, oy = ty * tile_size;
                out[n] = pt_tile{ox, oy, std::min(width - ox, tile_size), std::min(height - oy, tile_size)};
            }
            n++;
        }
    }
    return n;
}

int pt_scene_create(int device, const pt_scene_desc *d, pt_scene **out) {
    if(d == nullptr || out == nullptr) {
        return fail(PT_ERR_INVALID, "null argument");
    }
    *out = nullptr;
    const int n_dev = device_count_quiet();
    if(n_dev <= 0) {
        return fail(PT_ERR_NO_DEVICE, "no HIP device available; libpathtrace_hip has no CPU path");
    }
    if(device < 0 || device >= n_dev) {
        return fail(PT_ERR_NO_DEVICE, "device index out of range");
    }
    if(d->n_objects != d->n_triangles + d->n_spheres) {
        return fail(PT_ERR_INVALID, "n_objects must equal n_triangles + n_spheres");
    }
    if((d->n_objects > 0 && d->obj_kind == nullptr) || (d->n_triangles > 0 && (d->tri_pos == nullptr || d->tri_cull == nullptr || d->tri_material == nullptr)) ||
       (d->n_spheres > 0 && (d->sph == nullptr || d->sph_material == nullptr)) || (d->n_materials > 0 && d->materials == nullptr) ||
       (d->n_point_lights > 0 && (d->light_pos == nullptr || d->light_spectrum == nullptr))) {
        return fail(PT_ERR_INVALID, "missing array in scene description");
    }
    if(d->n_triangles > PT_REF_INDEX || d->n_spheres > PT_REF_INDEX) {
        return fail(PT_ERR_UNSUPPORTED, "too many objects");
    }

    std::unique_ptr<pt_scene> s(new pt_scene());
    s->device = device;
    PT_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    PT_HIP(hipGetDeviceProperties(&prop, device));
    s->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    PT_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    s->n_objects = d->n_objects;

    using clock = std::chrono::steady_clock;
    const auto t_begin = clock::now();
    auto ms_since = [](clock::time_point t0) { return std::chrono::duration<float, std::milli>(clock::now() - t0).count(); };

    // ---- objects in construction order: typed indices and the reference word of every leaf ---------------------------------
    std::vector<uint32_t> leaf_ref(d->n_objects);
    s->tri_obj.resize(d->n_triangles);
    s->sph_obj.resize(d->n_spheres);
    {
        uint32_t ti = 0, si = 0;
        for(uint32_t i = 0; i < d->n_objects; i++) {
            if(d->obj_kind[i] == PT_OBJ_TRIANGLE) {
                if(ti >= d->n_triangles) {
                    return fail(PT_ERR_INVALID, "obj_kind lists more triangles than n_triangles");
                }
                leaf_ref[i] = PT_REF_LEAF | ti;
                s->tri_obj[ti] = i;
                if(d->tri_material[ti] != PT_NO_MATERIAL && d->tri_material[ti] >= d->n_materials) {
                    return fail(PT_ERR_INVALID, "triangle material index out of range");
                }
                ti++;
            }
            else if(d->obj_kind[i] == PT_OBJ_SPHERE) {
                if(si >= d->n_spheres) {
                    return fail(PT_ERR_INVALID, "obj_kind lists more spheres than n_spheres");
                }
                leaf_ref[i] = PT_REF_LEAF | PT_REF_SPHERE | si;
                s->sph_obj[si] = i;
                if(d->sph_material[si] != PT_NO_MATERIAL && d->sph_material[si] >= d->n_materials) {
                    return fail(PT_ERR_INVALID, "sphere material index out of range");
                }
                si++;
            }
            else {
                return fail(PT_ERR_INVALID, "unknown object kind");
            }
        }
    }
    s->n_nodes = d->n_objects > 0 ? 2ULL * d->n_objects - 1ULL : 0ULL;

    // Where the tree is built.  PT_BUILD=device | host forces one; by default scenes of 1024 objects or more are built on the
    // device (pt_build.hip) and smaller ones by the host recursion (pt_bvh.cpp).  Both produce the same arrays, bit for bit.
    bool use_device = d->n_objects >= static_cast<uint32_t>(std::max(env_int("PT_BUILD_DEVICE_MIN", 1024), 2));
    if(const char *mode = std::getenv("PT_BUILD")) {
        if(std::strcmp(mode, "host") == 0) {
            use_device = false;
        }
        else if(std::strcmp(mode, "device") == 0) {
            use_device = d->n_objects >= 2;
        }
    }
    s->device_built = use_device;
    const bool align_siblings = env_int("PT_ALIGN_SIBLINGS", 1) != 0;

    std::vector<int32_t> dfs; // leaves depth-first, left to right (Scene::registerEmissiveObjects order); device path: only those with an emissive material
    uint32_t n_pairs = 0, root_ref = PT_REF_NONE;
    float root_lo[3] = {0, 0, 0}, root_hi[3] = {0, 0, 0};
    if(use_device) {
        // ---- device: upload the caller's arrays as they are; records, leaf boxes and the tree are made in HBM ----------------------
        s->build_ms[0] = ms_since(t_begin);
        const auto t_upload = clock::now();
        DevBuf<float> raw_pos, raw_nrm, raw_sph;
        DevBuf<uint8_t> raw_cull;
        DevBuf<uint32_t> raw_tri_mat, raw_tri_obj, raw_sph_mat, raw_sph_obj;
        auto up = [](auto &buf, const auto *src, size_t count) -> hipError_t {
            hipError_t e = buf.ensure(count);
            if(e != hipSuccess || count == 0) {
                return e;
            }
            return hipMemcpy(buf.ptr, src, count * sizeof(*src), hipMemcpyHostToDevice);
        };
        PT_HIP(up(raw_pos, d->tri_pos, 9 * static_cast<size_t>(d->n_triangles)));
        if(d->tri_nrm != nullptr) {
            PT_HIP(up(raw_nrm, d->tri_nrm, 9 * static_cast<size_t>(d->n_triangles)));
        }
        PT_HIP(up(raw_cull, d->tri_cull, d->n_triangles));
        PT_HIP(up(raw_tri_mat, d->tri_material, d->n_triangles));
        PT_HIP(up(raw_tri_obj, s->tri_obj.data(), d->n_triangles));
        PT_HIP(up(raw_sph, d->sph, 4 * static_cast<size_t>(d->n_spheres)));
        PT_HIP(up(raw_sph_mat, d->sph_material, d->n_spheres));
        PT_HIP(up(raw_sph_obj, s->sph_obj.data(), d->n_spheres));
        PT_HIP(s->tris.ensure(PT_TRI_QUADS * (static_cast<size_t>(d->n_triangles) + 1 + d->n_spheres))); // triangles, a spare record, spheres (pt_types.h)
        PT_HIP(hipMemsetAsync(s->tris.ptr + PT_TRI_QUADS * static_cast<size_t>(d->n_triangles), 0, PT_TRI_QUADS * sizeof(F4), s->stream));
        PT_HIP(s->tri_shade.ensure(8 * static_cast<size_t>(d->n_triangles)));
        PT_HIP(s->spheres.ensure(d->n_spheres));
        PT_HIP(s->sph_meta.ensure(d->n_spheres));
        s->build_ms[1] = ms_since(t_upload);

        PtBuildInput in;
        in.n_objects = d->n_objects;
        in.n_triangles = d->n_triangles;
        in.n_spheres = d->n_spheres;
        in.tri_pos = raw_pos.ptr;
        in.tri_nrm = d->tri_nrm != nullptr ? raw_nrm.ptr : nullptr;
        in.tri_cull = raw_cull.ptr;
        in.tri_material = raw_tri_mat.ptr;
        in.tri_obj = raw_tri_obj.ptr;
        in.sph = raw_sph.ptr;
        in.sph_material = raw_sph_mat.ptr;
        in.sph_obj = raw_sph_obj.ptr;
        in.align_siblings = align_siblings;
        PtBuildOutput built;
        built.tris = reinterpret_cast<float4 *>(s->tris.ptr);
        built.tri_shade = reinterpret_cast<float4 *>(s->tri_shade.ptr);
        built.spheres = reinterpret_cast<float4 *>(s->spheres.ptr);
        built.sph_meta = s->sph_meta.ptr;
        const char *what = "";
        const hipError_t e = pt_build_scene_device(s->stream, in, built, &what);
        if(e != hipSuccess) {
            return fail(PT_ERR_HIP, std::string("device scene build (") + what + "): " + hipGetErrorString(e));
        }
        s->pairs.ptr = reinterpret_cast<F4 *>(built.pairs);
        s->pairs.count = 4 * static_cast<size_t>(built.n_pairs);
        DevBuf<uint32_t> dfs_dev;
        dfs_dev.ptr = built.dfs;
        dfs_dev.count = d->n_objects;
        s->build_ms[2] = built.build_ms;
        s->depth = built.depth;
        if(s->depth > PT_MAX_DEPTH) {
            return fail(PT_ERR_UNSUPPORTED, "internal: BVH deeper than 128 levels (impl::constructBVH keeps a child at two thirds of its parent at most: 53 levels for 2^30 objects; tests/test_oracle_golden.py)");
        }
        n_pairs = built.n_pairs;
        root_ref = built.root_ref;
        for(int k = 0; k < 3; k++) {
            root_lo[k] = built.root_lo[k];
            root_hi[k] = built.root_hi[k];
        }
        // Only objects with an emissive material matter to registerEmissiveObjects: pick them out in construction order on the
        // host (sequential reads) and let the device put them into depth-first order.
        std::vector<uint8_t> lit(d->n_materials, 0);
        bool any_lit = false;
        for(uint32_t m = 0; m < d->n_materials; m++) {
            const float *e = d->materials[m].emission;
            lit[m] = (e[0] + e[1] + e[2]) * e[3] > 0.0F ? 1 : 0;
            any_lit = any_lit || lit[m] != 0;
        }
        if(any_lit) {
            std::vector<uint32_t> mask((static_cast<size_t>(d->n_objects) + 31) / 32, 0U);
            uint32_t n_selected = 0;
            for(uint32_t t = 0; t < d->n_triangles; t++) {
                const uint32_t m = d->tri_material[t];
                if(m != PT_NO_MATERIAL && lit[m] != 0) {
                    const uint32_t o = s->tri_obj[t];
                    mask[o >> 5] |= 1U << (o & 31U);
                    n_selected++;
                }
            }
            for(uint32_t i = 0; i < d->n_spheres; i++) {
                const uint32_t m = d->sph_material[i];
                if(m != PT_NO_MATERIAL && lit[m] != 0) {
                    const uint32_t o = s->sph_obj[i];
                    mask[o >> 5] |= 1U << (o & 31U);
                    n_selected++;
                }
            }
            std::vector<uint32_t> ordered;
            PT_HIP(pt_build_order_subset(s->stream, dfs_dev.ptr, d->n_objects, mask, n_selected, ordered));
            dfs.assign(ordered.begin(), ordered.end());
        }
    }
    else {
        // ---- host: leaf boxes, the recursion of pt_bvh.cpp, breadth-first flattening, records -----------------------------------
        std::vector<ptb::Box> boxes(d->n_objects);
        for(uint32_t i = 0; i < d->n_objects; i++) {
            ptb::Box &b = boxes[i];
            const uint32_t idx = leaf_ref[i] & PT_REF_INDEX;
            if((leaf_ref[i] & PT_REF_SPHERE) == 0) {
                const float *p = d->tri_pos + 9 * static_cast<size_t>(idx);
                for(int k = 0; k < 3; k++) { // Triangle::getBoundingVolume, object.cpp:184-186
                    b.lo[k] = fmin_std(fmin_std(p[k], p[3 + k]), p[6 + k]);
                    b.hi[k] = fmax_std(fmax_std(p[k], p[3 + k]), p[6 + k]);
                }
            }
            else {
                const float *sp = d->sph + 4 * static_cast<size_t>(idx);
                for(int k = 0; k < 3; k++) { // Sphere::getBoundingVolume, object.cpp:90-93
                    b.lo[k] = sp[k] - sp[3];
                    b.hi[k] = sp[k] + sp[3];
                }
            }
        }
        int threads = env_int("PT_BUILD_THREADS", static_cast<int>(std::thread::hardware_concurrency()));
        threads = std::max(1, std::min(threads, 64));
        s->tree = ptb::build_reference_bvh(boxes, threads);
        s->depth = s->tree.depth;
        if(s->depth > PT_MAX_DEPTH) {
            return fail(PT_ERR_UNSUPPORTED, "internal: BVH deeper than 128 levels (impl::constructBVH keeps a child at two thirds of its parent at most: 53 levels for 2^30 objects; tests/test_oracle_golden.py)");
        }
        ptb::FlatBvh flat = ptb::flatten_breadth_first(s->tree, leaf_ref, align_siblings);
        n_pairs = flat.n_pairs;
        root_ref = flat.root_ref;
        for(int k = 0; k < 3; k++) {
            root_lo[k] = flat.root_box.lo[k];
            root_hi[k] = flat.root_box.hi[k];
        }
        ptb::leaves_depth_first(s->tree, dfs);
        s->build_ms[0] = ms_since(t_begin);
        const auto t_upload = clock::now();

        std::vector<F4> tris(PT_TRI_QUADS * (static_cast<size_t>(d->n_triangles) + 1 + d->n_spheres), F4{0.0F, 0.0F, 0.0F, 0.0F}), shade(8 * static_cast<size_t>(d->n_triangles), F4{0.0F, 0.0F, 0.0F, 0.0F});
        for(uint32_t t = 0; t < d->n_triangles; t++) {
            const float *p = d->tri_pos + 9 * static_cast<size_t>(t);
            const Vec3 a = ld(p), b = ld(p + 3), c = ld(p + 6);
            const Vec3 ab = sub(b, a), ac = sub(c, a);
            const uint32_t obj_cull = s->tri_obj[t] | (d->tri_cull[t] != 0 ? 0x80000000U : 0U);
            tris[PT_TRI_QUADS * static_cast<size_t>(t) + 0] = {a.x, a.y, a.z, ab.x};
            tris[PT_TRI_QUADS * static_cast<size_t>(t) + 1] = {ab.y, ab.z, ac.x, ac.y};
            tris[PT_TRI_QUADS * static_cast<size_t>(t) + 2] = {ac.z, from_bits(d->tri_material[t]), from_bits(obj_cull), 0.0F};
            Vec3 na, nb, nc;
            if(d->tri_nrm != nullptr) {
                const float *q = d->tri_nrm + 9 * static_cast<size_t>(t);
                na = ld(q);
                nb = ld(q + 3);
                nc = ld(q + 6);
            }
            else {
                na = nb = nc = normalize(cross(ab, ac)); // Triangle::Triangle, object.cpp:118-124
            }
            for(int k = 0; k < 3; k++) {
                shade[8 * static_cast<size_t>(t) + k] = tris[PT_TRI_QUADS * static_cast<size_t>(t) + k];
            }
            shade[8 * static_cast<size_t>(t) + 3] = {na.x, na.y, na.z, nb.x};
            shade[8 * static_cast<size_t>(t) + 4] = {nb.y, nb.z, nc.x, nc.y};
            shade[8 * static_cast<size_t>(t) + 5] = {nc.z, 0.0F, 0.0F, 0.0F};
        }
        std::vector<F4> spheres(d->n_spheres);
        std::vector<uint2> sph_meta(d->n_spheres);
        for(uint32_t i = 0; i < d->n_spheres; i++) {
            const float *sp = d->sph + 4 * static_cast<size_t>(i);
            spheres[i] = {sp[0], sp[1], sp[2], sp[3]};
            tris[PT_TRI_QUADS * (static_cast<size_t>(d->n_triangles) + 1 + i)] = spheres[i]; // the record the traversal fetches (pt_types.h)
            sph_meta[i] = make_uint2(d->sph_material[i], s->sph_obj[i]);
        }
        std::vector<F4> pairs(4 * static_cast<size_t>(flat.n_pairs));
        std::memcpy(pairs.data(), flat.pairs.data(), flat.pairs.size() * sizeof(float));
        PT_HIP(s->pairs.upload(pairs));
        PT_HIP(s->tris.upload(tris));
        PT_HIP(s->tri_shade.upload(shade));
        PT_HIP(s->spheres.upload(spheres));
        PT_HIP(s->sph_meta.upload(sph_meta));
        s->build_ms[1] = ms_since(t_upload);
    }
    const auto t_rest = clock::now();

    std::vector<F4> materials(4 * static_cast<size_t>(d->n_materials));
    for(uint32_t i = 0; i < d->n_materials; i++) {
        const pt_material &m = d->materials[i];
        materials[4 * static_cast<size_t>(i) + 0] = {m.diffuse[0], m.diffuse[1], m.diffuse[2], m.diffuse[3]};
        materials[4 * static_cast<size_t>(i) + 1] = {m.specular[0], m.specular[1], m.specular[2], m.specular[3]};
        materials[4 * static_cast<size_t>(i) + 2] = {m.emission[0], m.emission[1], m.emission[2], m.emission[3]};
        materials[4 * static_cast<size_t>(i) + 3] = {m.ior, from_bits(static_cast<uint32_t>(m.bsdf)), from_bits(static_cast<uint32_t>(m.one_way != 0)), 0.0F};
        if(m.bsdf < PT_BSDF_LAMBERTIAN || m.bsdf > PT_BSDF_MIRROR) {
            return fail(PT_ERR_INVALID, "unknown BSDF kind");
        }
    }
    std::vector<F4> lights(2 * static_cast<size_t>(d->n_point_lights));
    for(uint32_t i = 0; i < d->n_point_lights; i++) {
        const float *p = d->light_pos + 3 * static_cast<size_t>(i);
        const float *c = d->light_spectrum + 4 * static_cast<size_t>(i);
        lights[2 * static_cast<size_t>(i) + 0] = {p[0], p[1], p[2], 0.0F};
        lights[2 * static_cast<size_t>(i) + 1] = {c[0], c[1], c[2], c[3]};
    }

    // ---- emissive objects: Scene::registerEmissiveObjects + CDF (scene.cpp:183-208, 167-180) -----------------------------------
    std::vector<F4> emis;
    std::vector<float> cdf;
    const float pi = static_cast<float>(M_PI);
    for(int32_t obj : dfs) {
        const uint32_t ref = leaf_ref[obj];
        const uint32_t idx = ref & PT_REF_INDEX;
        const bool is_sphere = (ref & PT_REF_SPHERE) != 0;
        const uint32_t mat = is_sphere ? d->sph_material[idx] : d->tri_material[idx];
        if(mat == PT_NO_MATERIAL) {
            continue; // default material has no emission
        }
        const float *e = d->materials[mat].emission;
        const float emissive_power = (e[0] + e[1] + e[2]) * e[3];
        if(emissive_power <= 0.0F) {
            continue;
        }
        float area;
        if(is_sphere) {
            const float r = d->sph[4 * static_cast<size_t>(idx) + 3];
            area = 4.0F * pi * (r * r); // object.cpp:95-99
        }
        else {
            const float *p = d->tri_pos + 9 * static_cast<size_t>(idx);
            const Vec3 c = cross(sub(ld(p + 3), ld(p)), sub(ld(p + 6), ld(p)));
            area = std::sqrt(dot(c, c)) / 2.0F; // object.cpp:188-190
        }
        const float object_probability = emissive_power * area;
        if(object_probability <= 0.0F) {
            continue;
        }
        if(is_sphere) {
            const float *sp = d->sph + 4 * static_cast<size_t>(idx);
            emis.push_back({sp[0], sp[1], sp[2], sp[3]});
            emis.push_back({0.0F, 0.0F, 0.0F, 0.0F});
            emis.push_back({0.0F, from_bits(ref), 0.0F, from_bits(mat)});
        }
        else {
            const float *p = d->tri_pos + 9 * static_cast<size_t>(idx);
            emis.push_back({p[0], p[1], p[2], p[3]});
            emis.push_back({p[4], p[5], p[6], p[7]});
            emis.push_back({p[8], from_bits(ref), from_bits(d->tri_cull[idx] != 0 ? 1U : 0U), from_bits(mat)});
        }
        emis.push_back({e[0], e[1], e[2], e[3]});
        cdf.push_back(object_probability);
        s->emissive_obj.push_back(obj);
    }
    {
        float cumulative_probability = 0.0F;
        for(float &v : cdf) {
            const float probability = v;
            v += cumulative_probability;
            cumulative_probability += probability;
        }
        for(float &v : cdf) {
            v /= cumulative_probability;
        }
    }
    s->n_emissive = static_cast<uint32_t>(cdf.size());
    s->emissive_cdf = cdf;
    const int emissive_object_count = static_cast<int>(cdf.size());
    const int object_sample_count = std::min(2 + static_cast<int>(std::log10(emissive_object_count + 1)), emissive_object_count); // scene.cpp:226
    if(d->n_point_lights + static_cast<uint32_t>(object_sample_count) > PT_MAX_NEE) {
        return fail(PT_ERR_UNSUPPORTED, "more than 32 light samples per path vertex (" + std::to_string(d->n_point_lights) + " point lights + " + std::to_string(object_sample_count) +
                                        " emitter samples): the visibility mask of a path vertex has 32 bits");
    }

    // ---- upload --------------------------------------------------------------------------------------------------------------
    PT_HIP(s->materials.upload(materials));
    PT_HIP(s->lights.upload(lights));
    PT_HIP(s->emis.upload(emis));
    PT_HIP(s->emis_cdf.upload(cdf));

    // ---- link: leaf records and pair records become ONE array, and the tree's references indices into it (pt_types.h) --------------
    // (the pair records start on an even record: two sibling nodes on slots 2k, 2k + 1 share one aligned 128-byte line only then)
    const uint32_t sphere_base = d->n_triangles + 1U, leaf_count = sphere_base + d->n_spheres, pair_base = (leaf_count + 1U) & ~1U;
    if(static_cast<uint64_t>(pair_base) + n_pairs > PT_REF_INDEX) {
        return fail(PT_ERR_UNSUPPORTED, "too many records for 30-bit references");
    }
    PT_HIP(s->recs.ensure(4 * (static_cast<size_t>(pair_base) + n_pairs)));
    PT_HIP(hipMemcpyAsync(s->recs.ptr, s->tris.ptr, 4 * static_cast<size_t>(leaf_count) * sizeof(F4), hipMemcpyDeviceToDevice, s->stream));
    if(pair_base > leaf_count) {
        PT_HIP(hipMemsetAsync(s->recs.ptr + 4 * static_cast<size_t>(leaf_count), 0, 4 * sizeof(F4), s->stream));
    }
    if(n_pairs > 0) {
        PT_HIP(hipMemcpyAsync(s->recs.ptr + 4 * static_cast<size_t>(pair_base), s->pairs.ptr, 4 * static_cast<size_t>(n_pairs) * sizeof(F4), hipMemcpyDeviceToDevice, s->stream));
        PT_HIP(pt_link_records(s->stream, reinterpret_cast<float4 *>(s->recs.ptr), pair_base, n_pairs, sphere_base));
    }
    PT_HIP(hipStreamSynchronize(s->stream));
    s->tris.release();
    s->pairs.release();
    if(root_ref != PT_REF_NONE) {
        root_ref += (root_ref & PT_REF_LEAF) == 0 ? pair_base : ((root_ref & PT_REF_SPHERE) != 0 ? sphere_base : 0U);
    }

    PtDevScene &dev = s->dev;
    dev.recs = reinterpret_cast<const float4 *>(s->recs.ptr);
    dev.pairs = dev.recs + 4 * static_cast<size_t>(pair_base);
    dev.tris = dev.recs;
    dev.pair_base = pair_base;
    dev.tri_shade = reinterpret_cast<const float4 *>(s->tri_shade.ptr);
    dev.spheres = reinterpret_cast<const float4 *>(s->spheres.ptr);
    dev.sph_meta = s->sph_meta.ptr;
    dev.materials = reinterpret_cast<const float4 *>(s->materials.ptr);
    dev.lights = reinterpret_cast<const float4 *>(s->lights.ptr);
    dev.emis = reinterpret_cast<const float4 *>(s->emis.ptr);
    dev.emis_cdf = s->emis_cdf.ptr;
    for(int k = 0; k < 3; k++) {
        dev.root_lo[k] = root_lo[k];
        dev.root_hi[k] = root_hi[k];
    }
    dev.root_ref = root_ref;
    dev.n_pairs = n_pairs;
    dev.n_tris = d->n_triangles;
    dev.n_spheres = d->n_spheres;
    dev.n_lights = d->n_point_lights;
    dev.n_emis = s->n_emissive;
    dev.n_materials = d->n_materials;
    dev.n_object_samples = static_cast<uint32_t>(object_sample_count);
    // LDS staging: a scene whose whole tree and leaf records fit in LDS NEXT TO everything else a workgroup keeps there, four workgroups
    // to the CU, lives in LDS entirely (the path kernel's IN_LDS variant): up to 15.8 KB of records with the small stack window, i.e. about
    // 120 triangles.  Larger ones are read through the caches like any tree: staged at three workgroups per CU they are slower than that
    // (176 triangles: 754 against 785 Msamples/s; at two, 256 triangles: 581 against 715 -- profiles/r03_lds_threshold.txt; round 2 staged up to
    // 24 KB).  An LDS copy of only the top of a larger tree was measured in round 1 and does not pay.  PT_LDS_SMALL_BYTES overrides the limit.
    const size_t small_bytes = (static_cast<size_t>(n_pairs) + pair_base) * 64;
    const int wide_word = d->n_point_lights + static_cast<uint32_t>(object_sample_count) > 8U ? 1 : 0;
    const size_t other_lds = pt_path_lds_bytes(wide_word, std::min(std::max(env_int("PT_ROWS", 4), 1), PT_MAX_ROWS), 4, 0U, 0U);
    const size_t room = other_lds < 40960 ? 40960 - other_lds : 0;
    if(small_bytes <= static_cast<size_t>(std::max(env_int("PT_LDS_SMALL_BYTES", static_cast<int>(room)), 0)) && env_int("PT_LDS_SMALL", 1) != 0) {
        dev.n_lds_pairs = n_pairs;
        dev.n_lds_tris = d->n_triangles;
    }
    else {
        dev.n_lds_pairs = 0;
        dev.n_lds_tris = 0;
    }

    int rc = setup_path(s.get());
    if(rc != PT_OK) {
        return rc;
    }
    s->build_ms[3] = ms_since(t_rest);
    if(env_int("PT_DEBUG", 0) != 0) {
        std::fprintf(stderr, "[pt] scene build (%s): %u objects, %u pair records, depth %u; host preparation %.1f ms, upload %.1f ms, device tree %.1f ms, rest %.1f ms\n",
                     s->device_built ? "device" : "host", d->n_objects, n_pairs, s->depth, s->build_ms[0], s->build_ms[1], s->build_ms[2], s->build_ms[3]);
    }
    *out = s.release();
    return PT_OK;
}

void pt_scene_destroy(pt_scene *scene) {
    if(scene == nullptr) {
        return;
    }
    (void)hipSetDevice(scene->device);
    if(scene->stream != nullptr) {
        (void)hipStreamSynchronize(scene->stream);
    }
    delete scene;
}

int pt_scene_info(const pt_scene *scene, uint64_t *n_nodes, uint32_t *depth, uint32_t *n_emissive) {
    if(scene == nullptr) {
        return fail(PT_ERR_INVALID, "null scene");
    }
    if(n_nodes != nullptr) {
        *n_nodes = scene->n_nodes;
    }
    if(depth != nullptr) {
        *depth = scene->depth;
    }
    if(n_emissive != nullptr) {
        *n_emissive = scene->n_emissive;
    }
    return PT_OK;
}

int pt_scene_emissive(const pt_scene *scene, int32_t *out_obj, float *out_cdf, uint64_t capacity, uint64_t *n_written) {
    if(scene == nullptr) {
        return fail(PT_ERR_INVALID, "null scene");
    }
    const uint64_t n = std::min<uint64_t>(scene->emissive_obj.size(), capacity);
    for(uint64_t i = 0; i < n; i++) {
        if(out_obj != nullptr) {
            out_obj[i] = scene->emissive_obj[i];
        }
        if(out_cdf != nullptr) {
            out_cdf[i] = scene->emissive_cdf[i];
        }
    }
    if(n_written != nullptr) {
        *n_written = scene->emissive_obj.size();
    }
    return PT_OK;
}

int pt_scene_bvh_dump(const pt_scene *scene, int32_t *out_obj, float *out_box, uint64_t capacity, uint64_t *n_written) {
    if(scene == nullptr || out_obj == nullptr || out_box == nullptr) {
        return fail(PT_ERR_INVALID, "null argument");
    }
    std::vector<int32_t> obj;
    std::vector<ptb::Box> box;
    if(!scene->device_built) {
        ptb::dump_preorder(scene->tree, obj, box);
    }
    else {
        // rebuild the pre-order listing from the pair records in HBM
        const uint32_t pair_base = scene->dev.pair_base, sphere_base = scene->dev.n_tris + 1U;
        std::vector<F4> pairs(4 * static_cast<size_t>(scene->dev.n_pairs));
        if(hipSetDevice(scene->device) != hipSuccess ||
           hipMemcpy(pairs.data(), scene->dev.pairs, pairs.size() * sizeof(F4), hipMemcpyDeviceToHost) != hipSuccess) {
            return fail(PT_ERR_HIP, "downloading the pair records failed");
        }
        struct Item {
            uint32_t ref;
            ptb::Box box;
        };
        ptb::Box root;
        for(int k = 0; k < 3; k++) {
            root.lo[k] = scene->dev.root_lo[k];
            root.hi[k] = scene->dev.root_hi[k];
        }
        std::vector<Item> stack{{scene->dev.root_ref, root}};
        obj.reserve(scene->n_nodes);
        box.reserve(scene->n_nodes);
        while(!stack.empty()) {
            const Item it = stack.back();
            stack.pop_back();
            box.push_back(it.box);
            if((it.ref & PT_REF_LEAF) != 0) {
                const uint32_t idx = it.ref & PT_REF_INDEX; // (record indices: pt_types.h)
                obj.push_back(static_cast<int32_t>((it.ref & PT_REF_SPHERE) != 0 ? scene->sph_obj[idx - sphere_base] : scene->tri_obj[idx]));
            }
            else {
                obj.push_back(-1);
                const float *q = &pairs[4 * static_cast<size_t>(it.ref - pair_base)].x;
                Item l, r;
                std::memcpy(&l.box, q, 24);
                std::memcpy(&r.box, q + 6, 24);
                std::memcpy(&l.ref, q + 12, 4);
                std::memcpy(&r.ref, q + 13, 4);
                stack.push_back(r);
                stack.push_back(l);
            }
        }
    }
    const uint64_t n = std::min<uint64_t>(obj.size(), capacity);
    for(uint64_t i = 0; i < n; i++) {
        out_obj[i] = obj[i];
        std::memcpy(out_box + 6 * i, &box[i], sizeof(float) * 6);
    }
    if(n_written != nullptr) {
        *n_written = obj.size();
    }
    return PT_OK;
}

int pt_intersect_batch(pt_scene *s, const float *rays, size_t n, float *out_t, int32_t *out_obj) {
    if(s == nullptr || (n > 0 && (rays == nullptr || out_t == nullptr || out_obj == nullptr))) {
        return fail(PT_ERR_INVALID, "null argument");
    }
    if(n == 0) {
        return PT_OK;
    }
    if(n > 0x7fffffffULL) {
        return fail(PT_ERR_INVALID, "too many rays in one batch");
    }
    std::lock_guard<std::mutex> lock(s->render_mutex);
    PT_HIP(hipSetDevice(s->device));
    const uint32_t n32 = static_cast<uint32_t>(n);
    {
        int rc = setup_path(s);
        if(rc != PT_OK) {
            return rc;
        }
        PtPathConfig cfg = s->path_cfg;
        PT_HIP(s->batch_rays.ensure(6 * n));
        PT_HIP(s->closest_out.ensure(n));
        PT_HIP(s->path_spill.ensure(((n + 255) / 256) * 256 * cfg.spill_depth));
        cfg.spill = s->path_spill.ptr;
        hipStream_t st = s->stream;
        PT_HIP(hipMemcpyAsync(s->batch_rays.ptr, rays, 6 * n * sizeof(float), hipMemcpyHostToDevice, st));
        pt_launch_closest(st, s->dev, s->batch_rays.ptr, n32, s->closest_out.ptr, cfg);
        PT_HIP(hipGetLastError());
        std::vector<uint2> hits(n);
        PT_HIP(hipMemcpyAsync(hits.data(), s->closest_out.ptr, n * sizeof(uint2), hipMemcpyDeviceToHost, st));
        PT_HIP(hipStreamSynchronize(st));
        for(size_t i = 0; i < n; i++) {
            const float t = from_bits(hits[i].x);
            const uint32_t ref = hits[i].y;
            out_t[i] = t;
            out_obj[i] = (t < 0.0F || ref == PT_REF_NONE) ? -1 : static_cast<int32_t>((ref & PT_REF_SPHERE) ? s->sph_obj[(ref & PT_REF_INDEX) - (s->dev.n_tris + 1U)] : s->tri_obj[ref & PT_REF_INDEX]);
        }
        return PT_OK;
    }
}

// Diagnostic, not part of include/pt_hip.h: walks n rays, `lanes_per_wave` of them per wavefront, with every traversal step stamped.
// out[4 * i ..] = steps, cycles spent waiting for records (flags bit 1: stamped run), cycles of the whole walk, -; flags bit 0: unused;
// behind the n results, 8 segment totals of 8 bytes per ray from a -DPT_STEP_STAMPS build (zeros otherwise): out holds 20 * n words (tools/step_timing.py).
// Diagnostics of the cost-aware placement (tools/place_probe.py): record what every stream of the following launches costs / read the
// last launch's costs / give the NEXT launch its first round as a table (waves x slots_per_wave entries, 0xffffffff = empty slot).
extern "C" int pt_debug_collect_costs(pt_scene *s, int on) {
    if(s == nullptr) {
        return fail(PT_ERR_INVALID, "null scene");
    }
    s->debug_collect_costs = on != 0;
    return PT_OK;
}

extern "C" int pt_debug_stream_costs(pt_scene *s, uint32_t *out, size_t n) {
    if(s == nullptr || out == nullptr || n > s->stream_cost.count) {
        return fail(PT_ERR_INVALID, "no costs of that many streams");
    }
    PT_HIP(hipSetDevice(s->device));
    PT_HIP(hipStreamSynchronize(s->stream));
    PT_HIP(hipMemcpy(out, s->stream_cost.ptr, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return PT_OK;
}

extern "C" int pt_debug_set_place(pt_scene *s, uint32_t waves, uint32_t slots_per_wave, const uint32_t *table) {
    if(s == nullptr || table == nullptr || waves == 0 || waves % 4U != 0 || slots_per_wave == 0) {
        return fail(PT_ERR_INVALID, "placement table");
    }
    s->debug_place.assign(table, table + static_cast<size_t>(waves) * slots_per_wave);
    s->debug_place_waves = waves;
    s->debug_place_slots = slots_per_wave;
    return PT_OK;
}

extern "C" int pt_debug_step_timing(pt_scene *s, const float *rays, size_t n, int lanes_per_wave, int flags, uint32_t *out) {
    if(s == nullptr || rays == nullptr || out == nullptr || n == 0 || n > 0x3fffffULL || lanes_per_wave < 1 || lanes_per_wave > 64) {
        return fail(PT_ERR_INVALID, "bad argument");
    }
    std::lock_guard<std::mutex> lock(s->render_mutex);
    PT_HIP(hipSetDevice(s->device));
    int rc = setup_path(s);
    if(rc != PT_OK) {
        return rc;
    }
    if(s->path_cfg.in_lds) {
        return fail(PT_ERR_UNSUPPORTED, "step timing is for scenes in HBM");
    }
    const size_t waves = (n + static_cast<size_t>(lanes_per_wave) - 1) / static_cast<size_t>(lanes_per_wave);
    const size_t threads = (waves + 3) / 4 * 256;
    DevBuf<float> d_rays;
    DevBuf<uint4> d_out;
    DevBuf<uint2> d_spill;
    PT_HIP(d_rays.ensure(6 * n));
    PT_HIP(d_out.ensure(5 * n));
    PT_HIP(hipMemsetAsync(d_out.ptr, 0, 5 * n * sizeof(uint4), s->stream));
    PT_HIP(d_spill.ensure(threads * s->path_cfg.spill_depth));
    hipStream_t st = s->stream;
    PT_HIP(hipMemcpyAsync(d_rays.ptr, rays, 6 * n * sizeof(float), hipMemcpyHostToDevice, st));
    pt_launch_steptime(st, s->dev, d_rays.ptr, static_cast<uint32_t>(n), static_cast<uint32_t>(lanes_per_wave), d_out.ptr, d_spill.ptr, s->path_cfg.spill_depth, flags);
    PT_HIP(hipGetLastError());
    PT_HIP(hipMemcpyAsync(out, d_out.ptr, 5 * n * sizeof(uint4), hipMemcpyDeviceToHost, st));
    PT_HIP(hipStreamSynchronize(st));
    return PT_OK;
}

// Diagnostic, not part of include/pt_hip.h: replays the rays the last render left in its rings (PT_RING_LOG_RAYS) through the traversal
// alone, at `waves_per_simd` wavefronts per SIMD with every ring cut into `parts`.  out[0..4] = rays, node visits, leaf tests, wave
// steps, checksum; *out_ms = kernel time; *out_blocks = resident workgroups per CU.
extern "C" int pt_debug_replay_rays(pt_scene *s, int waves_per_simd, int parts, unsigned long long *out, float *out_ms, int *out_blocks) {
    if(s == nullptr || out == nullptr || out_ms == nullptr || out_blocks == nullptr || parts < 1 || s->path_waves == 0) {
        return fail(PT_ERR_INVALID, "nothing to replay");
    }
    std::lock_guard<std::mutex> lock(s->render_mutex);
    PT_HIP(hipSetDevice(s->device));
    PtPathConfig cfg = s->path_cfg;
    const uint32_t waves = s->path_waves * static_cast<uint32_t>(parts);
    PT_HIP(s->path_spill.ensure(static_cast<size_t>(waves) * 64U * cfg.spill_depth));
    PT_HIP(s->path_wave_counters.ensure(8));
    PT_HIP(hipMemsetAsync(s->path_wave_counters.ptr, 0, 8 * sizeof(unsigned long long), s->stream));
    PtLocalQueue Q{};
    Q.ray_o = reinterpret_cast<float4 *>(s->lq_ray_o.ptr);
    Q.ray_d = reinterpret_cast<float4 *>(s->lq_ray_d.ptr);
    Q.cap = s->path_cap;
    Event e0, e1;
    PT_HIP(e0.create());
    PT_HIP(e1.create());
    PT_HIP(hipEventRecord(e0.e, s->stream));
    *out_blocks = pt_launch_replay(s->stream, s->dev, Q, s->path_waves, static_cast<uint32_t>(parts), waves_per_simd, cfg, s->path_spill.ptr, s->path_wave_counters.ptr);
    PT_HIP(hipGetLastError());
    PT_HIP(hipEventRecord(e1.e, s->stream));
    PT_HIP(hipMemcpyAsync(out, s->path_wave_counters.ptr, 5 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s->stream));
    PT_HIP(hipStreamSynchronize(s->stream));
    PT_HIP(hipEventElapsedTime(out_ms, e0.e, e1.e));
    return PT_OK;
}


int pt_render_streams(pt_scene *s, const pt_camera_params *camera, const pt_options *options, const pt_stream *streams, size_t n, float *out_image,
                      uint64_t *out_states, pt_stats *stats) {
    int rc = check_render_args(s, camera, options);
    if(rc != PT_OK) {
        return rc;
    }
    if(n > 0 && (streams == nullptr || out_image == nullptr)) {
        return fail(PT_ERR_INVALID, "null argument");
    }
    if(stats != nullptr) {
        std::memset(stats, 0, sizeof(*stats));
    }
    if(n == 0) {
        return PT_OK;
    }
    if(n > 0x0fffffffULL) {
        return fail(PT_ERR_INVALID, "too many streams");
    }
    PtDevOptions opt;
    rc = derive_options(options, &opt);
    if(rc != PT_OK) {
        return rc;
    }
    const PtDevCamera cam = derive_camera(camera);
    std::vector<int4> rects(n);
    std::vector<uint64_t> states(n);
    for(size_t i = 0; i < n; i++) {
        const pt_stream &t = streams[i];
        if(t.w < 0 || t.h < 0 || t.x < 0 || t.y < 0 || t.x + t.w > options->image_width || t.y + t.h > options->image_height) {
            return fail(PT_ERR_INVALID, "stream rectangle outside the image");
        }
        rects[i] = make_int4(t.x, t.y, t.w, t.h);
        states[i] = t.rng_state;
    }
    std::lock_guard<std::mutex> lock(s->render_mutex);
    PT_HIP(hipSetDevice(s->device));
    const uint32_t n32 = static_cast<uint32_t>(n);
    const size_t pixels = static_cast<size_t>(options->image_width) * static_cast<size_t>(options->image_height);
    PT_HIP(s->image.ensure(pixels));
    hipStream_t st = s->stream;
    {
        PT_HIP(s->st_rect.ensure(n));
        PT_HIP(s->st_rng.ensure(n));
        // pixels not covered by a stream keep the caller's values
        PT_HIP(hipMemcpyAsync(s->image.ptr, out_image, pixels * sizeof(F4), hipMemcpyHostToDevice, st));
        PT_HIP(hipMemcpyAsync(s->st_rect.ptr, rects.data(), n * sizeof(int4), hipMemcpyHostToDevice, st));
        PT_HIP(hipMemcpyAsync(s->st_rng.ptr, states.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        PtStreams T{};
        T.n = n32;
        T.rect = s->st_rect.ptr;
        T.rng = s->st_rng.ptr;
        rc = run_path(s, cam, opt, T, reinterpret_cast<float4 *>(s->image.ptr), stats, nullptr, nullptr);
        if(rc != PT_OK) {
            (void)hipStreamSynchronize(st); // rects / states are this function's vectors
            return rc;
        }
        PT_HIP(hipMemcpyAsync(out_image, s->image.ptr, pixels * sizeof(F4), hipMemcpyDeviceToHost, st));
        if(out_states != nullptr) {
            PT_HIP(hipMemcpyAsync(out_states, s->st_rng.ptr, n * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        }
        return finish_path(s);
    }
}

static int render_tiles_impl(pt_scene *s, const pt_camera_params *camera, const pt_options *options, const pt_tile *tiles, size_t n_tiles, uint64_t base_seed,
                             float4 *d_image, pt_stats *stats, pt_progress_fn progress = nullptr, void *progress_user = nullptr) {
    PtDevOptions opt;
    int rc = derive_options(options, &opt);
    if(rc != PT_OK) {
        return rc;
    }
    const PtDevCamera cam = derive_camera(camera);
    uint64_t total = 0;
    for(size_t i = 0; i < n_tiles; i++) {
        const pt_tile &t = tiles[i];
        if(t.w <= 0 || t.h <= 0 || t.x < 0 || t.y < 0 || t.x + t.w > options->image_width || t.y + t.h > options->image_height) {
            return fail(PT_ERR_INVALID, "tile outside the image or empty");
        }
        total += static_cast<uint64_t>(t.w) * static_cast<uint64_t>(t.h);
    }
    if(total > 0x0fffffffULL) {
        return fail(PT_ERR_INVALID, "too many pixels in one call");
    }
    const uint32_t n32 = static_cast<uint32_t>(total);
    {
        // stream i = pixel i of the tiles laid end to end; the kernel derives rectangle and engine from the tile table
        std::vector<int4> rects(n_tiles);
        std::vector<uint32_t> offsets(n_tiles), left(n_tiles);
        uint64_t at = 0;
        for(size_t k = 0; k < n_tiles; k++) {
            const pt_tile &t = tiles[k];
            rects[k] = make_int4(t.x, t.y, t.w, t.h);
            offsets[k] = static_cast<uint32_t>(at);
            left[k] = static_cast<uint32_t>(t.w) * static_cast<uint32_t>(t.h);
            at += left[k];
        }
        PT_HIP(s->tiles.ensure(n_tiles));
        PT_HIP(s->tile_offset.ensure(n_tiles));
        hipStream_t st = s->stream;
        PT_HIP(hipMemcpyAsync(s->tiles.ptr, rects.data(), n_tiles * sizeof(int4), hipMemcpyHostToDevice, st));
        PT_HIP(hipMemcpyAsync(s->tile_offset.ptr, offsets.data(), n_tiles * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        if(progress != nullptr) {
            PT_HIP(s->tile_left.ensure(n_tiles));
            PT_HIP(hipMemcpyAsync(s->tile_left.ptr, left.data(), n_tiles * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        }
        PT_HIP(hipStreamSynchronize(st)); // the tables are this function's vectors
        PtStreams T{};
        T.n = n32;
        T.tiles = s->tiles.ptr;
        T.tile_offset = s->tile_offset.ptr;
        T.n_tiles = static_cast<uint32_t>(n_tiles);
        T.base_seed = base_seed;
        // A regular grid of equal tiles (what pt_job_tiles makes of a frame whose sides are multiples of the tile size) lets the kernel
        // spread a wavefront's first rows over the frame's columns as well as over its bands: tiles per grid row, 64-stream chunks per tile.
        T.tiles_per_row = 0;
        T.chunks_per_tile = 0;
        if(n_tiles > 0 && (static_cast<uint32_t>(tiles[0].w) * static_cast<uint32_t>(tiles[0].h)) % 64U == 0) {
            bool regular = true;
            uint32_t per_row = 0;
            for(size_t k = 0; k < n_tiles && regular; k++) {
                regular = tiles[k].w == tiles[0].w && tiles[k].h == tiles[0].h;
                if(per_row == 0 && k > 0 && tiles[k].y != tiles[0].y) {
                    per_row = static_cast<uint32_t>(k);
                }
            }
            if(per_row == 0) {
                per_row = static_cast<uint32_t>(n_tiles);
            }
            for(size_t k = 0; k < n_tiles && regular; k++) {
                regular = tiles[k].x == tiles[0].x + static_cast<int32_t>(k % per_row) * tiles[0].w && tiles[k].y == tiles[0].y + static_cast<int32_t>(k / per_row) * tiles[0].h;
            }
            if(regular && n_tiles % per_row == 0 && per_row % 4 == 0) {
                T.tiles_per_row = per_row;
                T.chunks_per_tile = static_cast<uint32_t>(tiles[0].w) * static_cast<uint32_t>(tiles[0].h) / 64U;
            }
        }
        return run_path(s, cam, opt, T, d_image, stats, progress, progress_user);
    }
}

int pt_render_tiles(pt_scene *s, const pt_camera_params *camera, const pt_options *options, const pt_tile *tiles, size_t n_tiles, uint64_t base_seed,
                    float *out_image, pt_stats *stats) {
    return pt_render_tiles_progress(s, camera, options, tiles, n_tiles, base_seed, out_image, stats, nullptr, nullptr);
}

int pt_render_tiles_progress(pt_scene *s, const pt_camera_params *camera, const pt_options *options, const pt_tile *tiles, size_t n_tiles, uint64_t base_seed,
                             float *out_image, pt_stats *stats, pt_progress_fn progress, void *progress_user) {
    int rc = check_render_args(s, camera, options);
    if(rc != PT_OK) {
        return rc;
    }
    if(stats != nullptr) {
        std::memset(stats, 0, sizeof(*stats));
    }
    if(n_tiles == 0) {
        return PT_OK;
    }
    if(tiles == nullptr || out_image == nullptr) {
        return fail(PT_ERR_INVALID, "null argument");
    }
    std::lock_guard<std::mutex> lock(s->render_mutex);
    PT_HIP(hipSetDevice(s->device));
    const size_t pixels = static_cast<size_t>(options->image_width) * static_cast<size_t>(options->image_height);
    PT_HIP(s->image.ensure(pixels));
    PT_HIP(hipMemcpyAsync(s->image.ptr, out_image, pixels * sizeof(F4), hipMemcpyHostToDevice, s->stream));
    rc = render_tiles_impl(s, camera, options, tiles, n_tiles, base_seed, reinterpret_cast<float4 *>(s->image.ptr), stats, progress, progress_user);
    if(rc != PT_OK) {
        return rc;
    }
    PT_HIP(hipMemcpyAsync(out_image, s->image.ptr, pixels * sizeof(F4), hipMemcpyDeviceToHost, s->stream));
    return finish_path(s);
}

int pt_render_item(pt_scene *s, const pt_camera_params *camera, const pt_options *options, const pt_stream *item, float *out_tile, uint64_t *out_state,
                   pt_stats *stats) {
    int rc = check_render_args(s, camera, options);
    if(rc != PT_OK) {
        return rc;
    }
    if(item == nullptr) {
        return fail(PT_ERR_INVALID, "null argument");
    }
    if(stats != nullptr) {
        std::memset(stats, 0, sizeof(*stats));
    }
    if(item->w < 0 || item->h < 0 || item->x < 0 || item->y < 0 || item->x + item->w > options->image_width || item->y + item->h > options->image_height) {
        return fail(PT_ERR_INVALID, "work item outside the image");
    }
    if(out_state != nullptr) {
        *out_state = item->rng_state;
    }
    if(item->w == 0 || item->h == 0) {
        return PT_OK; // a zero-area WorkItem renders nothing and leaves its engine untouched
    }
    if(out_tile == nullptr) {
        return fail(PT_ERR_INVALID, "null argument");
    }
    PtDevOptions opt;
    rc = derive_options(options, &opt);
    if(rc != PT_OK) {
        return rc;
    }
    const PtDevCamera cam = derive_camera(camera);
    std::lock_guard<std::mutex> lock(s->render_mutex);
    PT_HIP(hipSetDevice(s->device));
    // the frame exists in device memory only; the host sees the item's rectangle
    const size_t pixels = static_cast<size_t>(options->image_width) * static_cast<size_t>(options->image_height);
    PT_HIP(s->image.ensure(pixels));
    PT_HIP(s->st_rect.ensure(1));
    PT_HIP(s->st_rng.ensure(1));
    hipStream_t st = s->stream;
    const int4 rect = make_int4(item->x, item->y, item->w, item->h);
    PT_HIP(hipMemcpyAsync(s->st_rect.ptr, &rect, sizeof(rect), hipMemcpyHostToDevice, st));
    PT_HIP(hipMemcpyAsync(s->st_rng.ptr, &item->rng_state, sizeof(uint64_t), hipMemcpyHostToDevice, st));
    PT_HIP(hipStreamSynchronize(st));
    PtStreams T{};
    T.n = 1;
    T.rect = s->st_rect.ptr;
    T.rng = s->st_rng.ptr;
    rc = run_path(s, cam, opt, T, reinterpret_cast<float4 *>(s->image.ptr), stats, nullptr, nullptr);
    if(rc != PT_OK) {
        return rc;
    }
    const F4 *first = s->image.ptr + static_cast<size_t>(item->y) * static_cast<size_t>(options->image_width) + static_cast<size_t>(item->x);
    PT_HIP(hipMemcpy2DAsync(out_tile, static_cast<size_t>(item->w) * sizeof(F4), first, static_cast<size_t>(options->image_width) * sizeof(F4),
                            static_cast<size_t>(item->w) * sizeof(F4), static_cast<size_t>(item->h), hipMemcpyDeviceToHost, st));
    if(out_state != nullptr) {
        PT_HIP(hipMemcpyAsync(out_state, s->st_rng.ptr, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    }
    return finish_path(s);
}

int pt_render_tiles_multi(pt_scene *const *scenes, int n_scenes, const pt_camera_params *camera, const pt_options *options, const pt_tile *tiles, size_t n_tiles,
                          uint64_t base_seed, float *out_image, pt_stats *stats, pt_progress_fn progress, void *progress_user) {
    if(scenes == nullptr || n_scenes < 1) {
        return fail(PT_ERR_INVALID, "no scenes");
    }
    for(int i = 0; i < n_scenes; i++) {
        const int rc = check_render_args(scenes[i], camera, options);
        if(rc != PT_OK) {
            return rc;
        }
    }
    if(n_tiles == 0) {
        return PT_OK;
    }
    if(tiles == nullptr || out_image == nullptr) {
        return fail(PT_ERR_INVALID, "null argument");
    }
    for(size_t k = 0; k < n_tiles; k++) {
        const pt_tile &t = tiles[k];
        if(t.w <= 0 || t.h <= 0 || t.x < 0 || t.y < 0 || t.x + t.w > options->image_width || t.y + t.h > options->image_height) {
            return fail(PT_ERR_INVALID, "tile outside the image or empty");
        }
    }
    // The multi-device form of doWorkParallel (src/worker.cpp:364-387): the tiles are dealt round-robin to the scenes (each a replica on its
    // own device; along the diagonals of a grid whose rows hold a multiple of n_scenes tiles, so that no device gets whole columns of the
    // frame -- cpupathtrace_amd/sharding.py uses the same rule), one host thread per scene drives its device, every device renders into its
    // own frame in HBM and only the rectangles of ITS tiles travel to the caller's image.  Engines are per pixel, so the image does not
    // depend on n_scenes.  progress calls are serialised and counted over all devices.
    size_t per_row = 0;
    while(per_row < n_tiles && tiles[per_row].y == tiles[0].y) {
        per_row++;
    }
    const bool diagonal = n_scenes > 1 && per_row > 0 && n_tiles % per_row == 0 && per_row % static_cast<size_t>(n_scenes) == 0;
    auto owner = [&](size_t k) -> int {
        return static_cast<int>((diagonal ? k % per_row + k / per_row : k) % static_cast<size_t>(n_scenes));
    };
    struct Shared {
        std::mutex progress_mutex;
        int completed = 0, total = 0;
        pt_progress_fn fn = nullptr;
        void *user = nullptr;
    } shared;
    shared.total = static_cast<int>(n_tiles);
    shared.fn = progress;
    shared.user = progress_user;
    auto trampoline = [](int, int, void *p) {
        Shared *sh = static_cast<Shared *>(p);
        std::lock_guard<std::mutex> lock(sh->progress_mutex);
        sh->completed++;
        sh->fn(sh->completed, sh->total, sh->user);
    };
    std::vector<int> rcs(static_cast<size_t>(n_scenes), PT_OK);
    std::vector<std::string> errors(static_cast<size_t>(n_scenes));
    const size_t width = static_cast<size_t>(options->image_width), pixels = width * static_cast<size_t>(options->image_height);
    auto work = [&](int i) {
        std::vector<pt_tile> mine;
        for(size_t k = 0; k < n_tiles; k++) {
            if(owner(k) == i) {
                mine.push_back(tiles[k]);
            }
        }
        if(mine.empty()) {
            return;
        }
        pt_scene *s = scenes[i];
        auto run = [&]() -> int {
            std::lock_guard<std::mutex> lock(s->render_mutex);
            PT_HIP(hipSetDevice(s->device));
            PT_HIP(s->image.ensure(pixels));
            int rc = render_tiles_impl(s, camera, options, mine.data(), mine.size(), base_seed, reinterpret_cast<float4 *>(s->image.ptr),
                                       stats != nullptr ? stats + i : nullptr, progress != nullptr ? static_cast<pt_progress_fn>(trampoline) : nullptr, &shared);
            if(rc != PT_OK) {
                return rc;
            }
            for(const pt_tile &t : mine) {
                const size_t at = static_cast<size_t>(t.y) * width + static_cast<size_t>(t.x);
                PT_HIP(hipMemcpy2DAsync(out_image + at * 4, width * sizeof(F4), s->image.ptr + at, width * sizeof(F4), static_cast<size_t>(t.w) * sizeof(F4),
                                        static_cast<size_t>(t.h), hipMemcpyDeviceToHost, s->stream));
            }
            return finish_path(s);
        };
        rcs[static_cast<size_t>(i)] = run();
        if(rcs[static_cast<size_t>(i)] != PT_OK) {
            errors[static_cast<size_t>(i)] = g_last_error; // this thread's message
        }
    };
    if(stats != nullptr) {
        std::memset(stats, 0, sizeof(*stats) * static_cast<size_t>(n_scenes));
    }
    std::vector<std::thread> threads;
    for(int i = 1; i < n_scenes; i++) {
        threads.emplace_back(work, i);
    }
    work(0);
    for(std::thread &t : threads) {
        t.join();
    }
    for(int i = 0; i < n_scenes; i++) {
        if(rcs[static_cast<size_t>(i)] != PT_OK) {
            return fail(rcs[static_cast<size_t>(i)], "scene " + std::to_string(i) + ": " + errors[static_cast<size_t>(i)]);
        }
    }
    return PT_OK;
}

int pt_render_tiles_device(pt_scene *s, const pt_camera_params *camera, const pt_options *options, const pt_tile *tiles, size_t n_tiles, uint64_t base_seed,
                           float *d_out_image, void *stream, pt_stats *stats) {
    int rc = check_render_args(s, camera, options);
    if(rc != PT_OK) {
        return rc;
    }
    if(stats != nullptr) {
        std::memset(stats, 0, sizeof(*stats));
    }
    if(n_tiles == 0) {
        return PT_OK;
    }
    if(tiles == nullptr || d_out_image == nullptr) {
        return fail(PT_ERR_INVALID, "null argument");
    }
    std::lock_guard<std::mutex> lock(s->render_mutex);
    PT_HIP(hipSetDevice(s->device));
    // order after the caller's stream, render on the library's stream, then make the caller's stream wait for it
    hipStream_t caller = static_cast<hipStream_t>(stream);
    Event ev;
    PT_HIP(ev.create(hipEventDisableTiming));
    PT_HIP(hipEventRecord(ev.e, caller));
    PT_HIP(hipStreamWaitEvent(s->stream, ev.e, 0));
    rc = render_tiles_impl(s, camera, options, tiles, n_tiles, base_seed, reinterpret_cast<float4 *>(d_out_image), stats);
    if(rc == PT_OK) {
        PT_HIP(hipEventRecord(ev.e, s->stream));
        PT_HIP(hipStreamWaitEvent(caller, ev.e, 0));
        // This entry point does not wait for the device (the frame stays in HBM for the caller's stream).  With statistics it has waited
        // and checked already (run_path); PT_VERIFY=1 makes every call wait and check.
        if(stats == nullptr && env_int("PT_VERIFY", 0) != 0) {
            rc = finish_path(s);
        }
    }
    return rc;
}

} // extern "C"

// ---- post-processing (pt_post.hip) ----------------------------------------------------------------------------------------------

static int post_check(int device, const float *rgba, int32_t width, int32_t height, uint32_t steps, float gamma) {
    if(width < 0 || height < 0 || (rgba == nullptr && static_cast<long long>(width) * height > 0)) {
        return fail(PT_ERR_INVALID, "bad image");
    }
    if((steps & ~(PT_POST_TONE_MAP | PT_POST_GAMMA)) != 0 || steps == 0) {
        return fail(PT_ERR_INVALID, "steps must be PT_POST_TONE_MAP and/or PT_POST_GAMMA");
    }
    if((steps & PT_POST_GAMMA) != 0 && !(gamma == gamma)) {
        return fail(PT_ERR_INVALID, "gamma is NaN");
    }
    const int n_dev = device_count_quiet();
    if(n_dev <= 0) {
        return fail(PT_ERR_NO_DEVICE, "no HIP device available; libpathtrace_hip has no CPU path");
    }
    if(device < 0 || device >= n_dev) {
        return fail(PT_ERR_NO_DEVICE, "device index out of range");
    }
    return PT_OK;
}

int pt_post_process_device(int device, float *d_rgba, int32_t width, int32_t height, uint32_t steps, float gamma, void *stream) {
    const int rc = post_check(device, d_rgba, width, height, steps, gamma);
    if(rc != PT_OK) {
        return rc;
    }
    PT_HIP(hipSetDevice(device));
    static_assert(PT_POST_TONE_MAP == PT_POST_STEP_TONE_MAP && PT_POST_GAMMA == PT_POST_STEP_GAMMA, "step bits");
    PT_HIP(pt_post_run(static_cast<hipStream_t>(stream), reinterpret_cast<float4 *>(d_rgba), width, height, steps, gamma));
    return PT_OK;
}

int pt_post_process(int device, float *rgba, int32_t width, int32_t height, uint32_t steps, float gamma) {
    const int rc = post_check(device, rgba, width, height, steps, gamma);
    if(rc != PT_OK) {
        return rc;
    }
    const size_t count = static_cast<size_t>(width) * static_cast<size_t>(height);
    if(count == 0) {
        return PT_OK;
    }
    PT_HIP(hipSetDevice(device));
    DevBuf<F4> frame;
    PT_HIP(frame.ensure(count));
    PT_HIP(hipMemcpy(frame.ptr, rgba, count * sizeof(F4), hipMemcpyHostToDevice));
    PT_HIP(pt_post_run(nullptr, reinterpret_cast<float4 *>(frame.ptr), width, height, steps, gamma));
    PT_HIP(hipMemcpy(rgba, frame.ptr, count * sizeof(F4), hipMemcpyDeviceToHost));
    return PT_OK;
}
