// pt_api.cpp -- host side of libpathtrace_hip.so: the C ABI of include/pt_hip.h.
//
// Scene creation flattens the caller's object list into the HBM layout of pt_types.h (building the reference's BVH
// topology on the way, pt_bvh.cpp); rendering drives the wavefront loop: shade -> trace -> shade -> ... until every stream
// has rendered all its pixels.  There is no CPU rendering path in this library.
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

#define PT_MAX_GROUPS 8 /* independent slices of the streams pipelined on separate HIP streams */

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
    DevBuf<F4> pairs, quads, tris, tri_shade, spheres, materials, lights, emis;
    DevBuf<uint2> sph_meta;
    DevBuf<float> emis_cdf;
    PtDevScene dev{};

    // render workspace (grown on demand, reused between calls)
    uint32_t ws_slots = 0;
    DevBuf<int4> rect;
    DevBuf<uint64_t> rng;
    DevBuf<int32_t> cursor, path_length;
    DevBuf<uint32_t> flags, nee_mask, vis;
    DevBuf<F4> ray_o, ray_d, spectrum, out, nee;
    DevBuf<double> divisor, bounce_pd;
    DevBuf<PtEstimator> est;
    DevBuf<PtCandidate> cand;
    DevBuf<uint2> hit;
    DevBuf<F4> q_ray_o, q_ray_d;
    DevBuf<uint32_t> q_header; // count[8], head[8], each word on its own 256-byte line
    DevBuf<uint2> spill;
    DevBuf<PtDevCounters> counters;
    DevBuf<F4> image;
    DevBuf<int4> tiles;
    DevBuf<uint32_t> tile_offset;
    DevBuf<float> batch_rays;
    DevBuf<uint32_t> walk_hist;
    DevBuf<unsigned long long> shade_wave_counters, trace_wave_counters;
    hipEvent_t check_event[2 * PT_MAX_GROUPS] = {};
    // suspended walks: two pools per group
    DevBuf<F4> carry_o, carry_d;
    DevBuf<uint4> carry_state;
    DevBuf<uint32_t> carry_sp, carry_header;
    DevBuf<uint2> carry_stack;
    uint32_t carry_cap = 0; // per pool
    uint32_t shard_capacity = 0;   // per group
    uint32_t ws_groups = 1;        // groups the queue / spill / counter buffers are sized for
    PtTraceConfig trace_cfg{};
    PtDevCounters *host_counters = nullptr; // pinned, 2 x PT_MAX_GROUPS entries (the done-check reads the previous batch's copy)
    hipStream_t group_stream[PT_MAX_GROUPS] = {};
    uint32_t concurrent_streams = 0; // group streams found to run side by side (pick_concurrent_streams)

    // One render call at a time per scene: the workspace below is shared by every entry point (processItem may be called from several
    // threads on one const Scene, reference worker.h:66-69 / src/worker.cpp:328-362: such callers are serialised here).
    std::mutex render_mutex;

    // workspace of the persistent path kernel (pt_path.hip)
    bool use_path = true;
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
    DevBuf<unsigned long long> path_wave_counters;
    uint32_t *host_tiles_done = nullptr; // pinned: tiles finished so far, written by the kernel (progress callback)

    ~pt_scene() {
        if(host_counters != nullptr) {
            (void)hipHostFree(host_counters);
        }
        if(host_tiles_done != nullptr) {
            (void)hipHostFree(host_tiles_done);
        }
        for(hipEvent_t e : check_event) {
            if(e != nullptr) {
                (void)hipEventDestroy(e);
            }
        }
        for(size_t g = 0; g < PT_MAX_GROUPS; g++) {
            // (later entries may repeat an earlier stream when the runtime offered fewer concurrent ones than groups)
            if(group_stream[g] != nullptr && std::find(group_stream, group_stream + g, group_stream[g]) == group_stream + g) {
                (void)hipStreamDestroy(group_stream[g]);
            }
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

// view of the slots [start, start + n) of the workspace
PtPaths make_paths(pt_scene *s, uint32_t n, uint32_t start = 0) {
    PtPaths P{};
    P.n = n;
    P.rect = s->rect.ptr + start;
    P.rng = s->rng.ptr + start;
    P.cursor = s->cursor.ptr + start;
    P.flags = s->flags.ptr + start;
    P.ray_o = reinterpret_cast<float4 *>(s->ray_o.ptr) + start;
    P.ray_d = reinterpret_cast<float4 *>(s->ray_d.ptr) + start;
    P.spectrum = reinterpret_cast<float4 *>(s->spectrum.ptr) + start;
    P.out = reinterpret_cast<float4 *>(s->out.ptr) + start;
    P.divisor = s->divisor.ptr + start;
    P.bounce_pd = s->bounce_pd.ptr + start;
    P.path_length = s->path_length.ptr + start;
    P.nee = reinterpret_cast<float4 *>(s->nee.ptr) + start;
    P.nee_stride = s->ws_slots;
    P.nee_mask = s->nee_mask.ptr + start;
    P.est = s->est.ptr + start;
    P.cand = s->cand.ptr + static_cast<size_t>(start) * PT_MAX_CANDIDATES;
    P.hit = s->hit.ptr + start;
    P.vis = s->vis.ptr + start;
    P.wave_counters = s->shade_wave_counters.ptr + 2 * static_cast<size_t>(start / 64);
    return P;
}

PtQueue make_queue(pt_scene *s, uint32_t group = 0) {
    PtQueue q{};
    const size_t rays = static_cast<size_t>(s->shard_capacity) * PT_SHARDS;
    q.ray_o = reinterpret_cast<float4 *>(s->q_ray_o.ptr) + group * rays;
    q.ray_d = reinterpret_cast<float4 *>(s->q_ray_d.ptr) + group * rays;
    // two headers per group (count[8] + head[8] words each), used by alternate launches
    q.count = s->q_header.ptr + static_cast<size_t>(group) * 4 * PT_SHARDS * PT_QSTRIDE;
    q.head = q.count + PT_SHARDS * PT_QSTRIDE;
    q.next_header = nullptr;
    q.shard_capacity = s->shard_capacity;
    return q;
}

PtCarry make_carry(pt_scene *s, uint32_t group = 0) {
    PtCarry c{};
    const size_t base = static_cast<size_t>(group) * 2 * s->carry_cap;
    c.ray_o = reinterpret_cast<float4 *>(s->carry_o.ptr) + base;
    c.ray_d = reinterpret_cast<float4 *>(s->carry_d.ptr) + base;
    c.state = s->carry_state.ptr + base;
    c.sp = s->carry_sp.ptr + base;
    c.depth = std::max<uint32_t>(s->depth, 1U);
    c.stack = s->carry_stack.ptr + base * c.depth;
    c.count = s->carry_header.ptr + static_cast<size_t>(group) * 4 * PT_QSTRIDE;
    c.head = c.count + 2 * PT_QSTRIDE;
    c.cap = s->carry_cap;
    return c;
}

// Group streams that really run side by side.  The runtime multiplexes HIP streams onto a few hardware queues (four by default),
// and kernels of two streams that share a queue run one after the other -- which stream lands on which queue depends on what else
// the process has created.  Candidates are therefore PROBED: a 400 us do-nothing kernel on each of two streams takes 400 us when
// they are concurrent and 800 us when they are not.  `wanted` streams that are pairwise concurrent are kept; if the runtime offers
// fewer, the later groups share the last stream found (they then simply queue up).
int pick_concurrent_streams(pt_scene *s, uint32_t wanted) {
    using clock = std::chrono::steady_clock;
    std::vector<hipStream_t> candidates, chosen;
    for(hipStream_t gs : s->group_stream) {
        if(gs != nullptr && std::find(chosen.begin(), chosen.end(), gs) == chosen.end()) {
            chosen.push_back(gs); // streams picked earlier stay
        }
    }
    auto concurrent = [&](hipStream_t a, hipStream_t b, bool &result) -> int {
        double best = 1e30;
        for(int attempt = 0; attempt < 2; attempt++) { // the faster of two tries: a slow first launch must not look like queueing
            PT_HIP(hipStreamSynchronize(a));
            PT_HIP(hipStreamSynchronize(b));
            const auto t0 = clock::now();
            pt_launch_spin(a, 400);
            pt_launch_spin(b, 400);
            PT_HIP(hipStreamSynchronize(a));
            PT_HIP(hipStreamSynchronize(b));
            best = std::min(best, std::chrono::duration<double, std::micro>(clock::now() - t0).count());
        }
        result = best < 650.0;
        return PT_OK;
    };
    const int max_candidates = 12;
    for(int i = 0; i < max_candidates && chosen.size() < wanted; i++) {
        hipStream_t c = nullptr;
        PT_HIP(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
        candidates.push_back(c);
        pt_launch_spin(c, 1); // first use of a stream may create its queue: keep that out of the probe
        bool ok = true;
        for(hipStream_t other : chosen) {
            bool side_by_side = false;
            int rc = concurrent(c, other, side_by_side);
            if(rc != PT_OK) {
                return rc;
            }
            if(!side_by_side) {
                ok = false;
                break;
            }
        }
        if(ok) {
            chosen.push_back(c);
        }
    }
    for(hipStream_t c : candidates) {
        if(std::find(chosen.begin(), chosen.end(), c) == chosen.end()) {
            (void)hipStreamDestroy(c);
        }
    }
    s->concurrent_streams = static_cast<uint32_t>(chosen.size());
    for(uint32_t g = 0; g < PT_MAX_GROUPS; g++) {
        s->group_stream[g] = chosen.empty() ? nullptr : (g < chosen.size() ? chosen[g] : (g < wanted ? chosen.back() : nullptr));
    }
    if(env_int("PT_DEBUG", 0) != 0) {
        std::fprintf(stderr, "[pt] %zu of %u wanted group streams run concurrently (probed %zu candidates)\n", chosen.size(), wanted, candidates.size());
    }
    return chosen.empty() ? fail(PT_ERR_HIP, "no usable HIP stream") : PT_OK;
}

// workspace for n stream slots and a queue of `queue_rays_per_slot` rays per slot
int ensure_workspace(pt_scene *s, uint32_t n, uint32_t rays_per_slot, uint32_t groups = 1) {
    // each group holds ceil(n / groups) slots (rounded up to whole 2048-slot units, see group_ranges)
    const uint32_t per_group = ((n + groups - 1) / groups + 2047U) / 2048U * 2048U;
    const uint32_t blocks = (per_group + 255) / 256;
    const uint32_t blocks_per_shard = (blocks + PT_SHARDS - 1) / PT_SHARDS;
    const uint32_t cap = std::max<uint32_t>(blocks_per_shard * 256U * rays_per_slot, 256U);
    PT_HIP(s->rect.ensure(n));
    PT_HIP(s->rng.ensure(n));
    PT_HIP(s->cursor.ensure(n));
    PT_HIP(s->path_length.ensure(n));
    PT_HIP(s->flags.ensure(n));
    PT_HIP(s->nee_mask.ensure(n));
    PT_HIP(s->vis.ensure(static_cast<size_t>(n) * PT_MAX_NEE));
    PT_HIP(s->ray_o.ensure(n));
    PT_HIP(s->ray_d.ensure(n));
    PT_HIP(s->spectrum.ensure(n));
    PT_HIP(s->out.ensure(n));
    PT_HIP(s->nee.ensure(static_cast<size_t>(n) * PT_MAX_NEE));
    PT_HIP(s->divisor.ensure(n));
    PT_HIP(s->bounce_pd.ensure(n));
    PT_HIP(s->est.ensure(n));
    PT_HIP(s->cand.ensure(static_cast<size_t>(n) * PT_MAX_CANDIDATES));
    PT_HIP(s->hit.ensure(n));
    // `nee` is addressed as [plane][slot] with the allocation's slot count as stride: keep the stride in step with it
    s->ws_slots = static_cast<uint32_t>(std::min<size_t>(s->nee.count / PT_MAX_NEE, 0xffffffffu));
    const size_t want_rays = static_cast<size_t>(cap) * PT_SHARDS * groups;
    if(s->q_ray_o.count < want_rays || s->ws_groups != groups || s->shard_capacity < cap) {
        PT_HIP(s->q_ray_o.ensure(want_rays));
        PT_HIP(s->q_ray_d.ensure(want_rays));
        s->shard_capacity = cap;
        s->ws_groups = groups;
    }
    PT_HIP(s->q_header.ensure(static_cast<size_t>(PT_MAX_GROUPS) * 4 * PT_SHARDS * PT_QSTRIDE));
    PT_HIP(s->counters.ensure(PT_MAX_GROUPS));
    PT_HIP(s->shade_wave_counters.ensure(2 * (static_cast<size_t>(n) / 64 + PT_MAX_GROUPS)));
    PT_HIP(s->trace_wave_counters.ensure(static_cast<size_t>(s->trace_cfg.grid) * 4 * 8 * groups));
    PT_HIP(s->spill.ensure(static_cast<size_t>(s->trace_cfg.grid) * 256 * s->trace_cfg.spill_depth * groups));
    {
        // room for a quarter of a group's rays to be suspended at once (a full pool only means walks are not suspended)
        const uint32_t want_cap = std::max<uint32_t>(4096U, per_group / 4U * rays_per_slot);
        if(s->carry_cap < want_cap || s->carry_o.count < static_cast<size_t>(want_cap) * 2 * groups) {
            s->carry_cap = std::max(s->carry_cap, want_cap);
            const size_t total = static_cast<size_t>(s->carry_cap) * 2 * PT_MAX_GROUPS;
            PT_HIP(s->carry_o.ensure(total));
            PT_HIP(s->carry_d.ensure(total));
            PT_HIP(s->carry_state.ensure(total));
            PT_HIP(s->carry_sp.ensure(total));
            PT_HIP(s->carry_stack.ensure(total * std::max<uint32_t>(s->depth, 1U)));
        }
        PT_HIP(s->carry_header.ensure(static_cast<size_t>(PT_MAX_GROUPS) * 4 * PT_QSTRIDE));
    }
    s->trace_cfg.spill = s->spill.ptr;
    if(env_int("PT_WALK_HIST", 0) != 0 && s->walk_hist.ptr == nullptr) {
        PT_HIP(s->walk_hist.ensure(64));
        PT_HIP(hipMemset(s->walk_hist.ptr, 0, 64 * sizeof(uint32_t)));
        s->trace_cfg.walk_hist = s->walk_hist.ptr;
    }
    if(s->host_counters == nullptr) {
        PT_HIP(hipHostMalloc(reinterpret_cast<void **>(&s->host_counters), sizeof(PtDevCounters) * 2 * PT_MAX_GROUPS, hipHostMallocDefault));
        for(auto &e : s->check_event) {
            PT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
    }
    if(groups > 1 && s->group_stream[groups - 1] == nullptr) {
        int rc = pick_concurrent_streams(s, groups);
        if(rc != PT_OK) {
            return rc;
        }
    }
    else if(s->group_stream[0] == nullptr) {
        PT_HIP(hipStreamCreateWithFlags(&s->group_stream[0], hipStreamNonBlocking));
    }
    return PT_OK;
}

int setup_trace(pt_scene *s) {
    PtTraceConfig &cfg = s->trace_cfg;
    if(cfg.grid != 0) {
        return PT_OK;
    }
    int stack_lds = env_int("PT_STACK_LDS", 16);
    if(stack_lds != 8 && stack_lds != 16 && stack_lds != 24) {
        stack_lds = 16;
    }
    cfg.stack_lds = stack_lds;
    cfg.lds_mode = (s->dev.n_lds_pairs == 0 && s->dev.n_lds_tris == 0) ? (s->dev.quads != nullptr ? 3 : 0) : ((s->dev.n_lds_pairs == s->dev.n_pairs && s->dev.n_lds_tris == s->dev.n_tris) ? 2 : 1);
    cfg.lds_bytes = static_cast<size_t>(stack_lds) * 256 * sizeof(uint2) + static_cast<size_t>(s->dev.n_lds_pairs) * 64 + static_cast<size_t>(s->dev.n_lds_tris) * 48;
    const int per_cu = pt_trace_blocks_per_cu(stack_lds, cfg.lds_mode, cfg.lds_bytes);
    const int limit = env_int("PT_TRACE_BLOCKS_PER_CU", 0);
    cfg.grid = s->cu_count * ((limit > 0 && limit < per_cu) ? limit : per_cu);
    cfg.spill_depth = s->depth > static_cast<uint32_t>(stack_lds) ? s->depth - static_cast<uint32_t>(stack_lds) : 1U;
    cfg.spill = nullptr; // allocated with the workspace (one area per group)
    // measured (profiles/): short LDS-resident walks want few dequeue atomics (256 rays each, refill at 20 idle lanes); long walks
    // through HBM-resident trees want fine-grained balancing between wavefronts (96 rays, refill at 12)
    const bool small_scene = cfg.lds_mode == 2;
    cfg.refill_idle = std::min(std::max(env_int("PT_REFILL_IDLE", small_scene ? 20 : 12), 1), 64);
    cfg.leaf_min = std::min(std::max(env_int("PT_LEAF_MIN", 1), 1), 64);
    cfg.chunk = std::min(std::max(env_int("PT_QCHUNK", small_scene ? 256 : 96), 16), 4096);
    cfg.burst_steps = std::min(std::max(env_int("PT_BURST", 4), 1), 64);
    if(env_int("PT_DEBUG", 0) != 0) {
        std::fprintf(stderr, "[pt] trace config: grid %d (%d CUs x %d blocks), stack_lds %d, lds mode %d, lds %zu B, spill depth %u, lds pairs %u, lds tris %u\n", cfg.grid,
                     s->cu_count, per_cu, stack_lds, cfg.lds_mode, cfg.lds_bytes, cfg.spill_depth, s->dev.n_lds_pairs, s->dev.n_lds_tris);
    }
    return PT_OK;
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
    // closed candidates a pixel can accumulate (worker.cpp:214-222)
    const int batches = std::max(o->max_sample_count, 0) / d.stats_sample_count;
    const int closed = batches > 0 ? (batches - 1) / d.candidate_batch_count : 0;
    if(closed > PT_MAX_CANDIDATES) {
        return fail(PT_ERR_UNSUPPORTED, "sample counts give more than 8 estimator candidates per pixel");
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

// The wavefront loop over an initialised set of `n` stream slots.
//
// The slots are cut into up to PT_MAX_GROUPS contiguous groups, each with its own ray queue, counters and HIP stream.  A
// group's iteration is shade -> trace on its stream; the groups are independent (a ray's destination is a slot of its own
// group), so the drain phase of one group's persistent traversal kernel -- a few long walks through the glass mesh --
// overlaps with the other groups' work instead of idling the chip.
int run_wavefront(pt_scene *s, const PtDevCamera &cam, const PtDevOptions &opt, uint32_t n, uint32_t groups, float4 *d_image, pt_stats *stats) {
    const uint32_t rays_per_slot = 1U + s->dev.n_lights + s->dev.n_object_samples;
    struct Group {
        uint32_t start, count;
        PtPaths P;
        PtQueue q;
        PtCarry carry;
        PtTraceConfig cfg;
        hipStream_t st;
        bool done;
    };
    std::vector<Group> G(groups);
    const int shard_mode = env_int("PT_SHARD_MODE", 1);
    const int drain_lanes = std::min(std::max(env_int("PT_DRAIN_LANES", 16), 0), 64);
    const int endgame_percent = std::min(std::max(env_int("PT_ENDGAME_PERCENT", 90), 1), 101);
    const int max_steps_env = env_int("PT_MAX_STEPS", 256);
    const int max_steps = max_steps_env > 0 ? max_steps_env : 0x7fffffff;
    const int group_blocks_per_cu = std::max(env_int("PT_GROUP_BLOCKS_PER_CU", 2), 1);
    const uint32_t per_group = ((n + groups - 1) / groups + 2047U) / 2048U * 2048U;
    uint32_t n_groups = 0;
    for(uint32_t g = 0; g < groups; g++) {
        const uint32_t begin = std::min<uint64_t>(static_cast<uint64_t>(g) * per_group, n);
        const uint32_t end = std::min<uint64_t>(static_cast<uint64_t>(g + 1) * per_group, n);
        if(end <= begin) {
            break;
        }
        Group &gr = G[n_groups];
        gr.start = begin;
        gr.count = end - begin;
        gr.P = make_paths(s, gr.count, begin);
        gr.q = make_queue(s, n_groups);
        gr.carry = make_carry(s, n_groups);
        gr.cfg = s->trace_cfg;
        gr.cfg.max_steps = max_steps;
        gr.cfg.drain_lanes = drain_lanes;
        // no more workgroups than the rays one iteration can produce
        const uint64_t blocks = (static_cast<uint64_t>(gr.count) * rays_per_slot + 255) / 256;
        gr.cfg.grid = static_cast<int>(std::max<uint64_t>(1, std::min<uint64_t>(static_cast<uint64_t>(gr.cfg.grid), blocks)));
        if(groups > 1 && gr.cfg.lds_mode != 2) {
            // concurrent groups share the CUs: two persistent workgroups per CU each (three groups fill a CU's LDS and wave slots).
            // (Giving the groups that are still rendering the CUs of the finished ones was measured: no gain -- by then they are in
            // the latency-bound end of the frame.)
            gr.cfg.grid = std::min(gr.cfg.grid, s->cu_count * group_blocks_per_cu);
        }
        gr.cfg.spill = s->spill.ptr + static_cast<size_t>(n_groups) * s->trace_cfg.grid * 256 * s->trace_cfg.spill_depth;
        gr.cfg.wave_counters = s->trace_wave_counters.ptr + static_cast<size_t>(n_groups) * s->trace_cfg.grid * 32;
        gr.st = s->group_stream[n_groups];
        gr.done = false;
        n_groups++;
    }
    G.resize(n_groups);

    // order the groups' streams after the initialisation that ran on the main stream
    hipEvent_t ev_init;
    PT_HIP(hipEventCreateWithFlags(&ev_init, hipEventDisableTiming));
    PT_HIP(hipMemsetAsync(s->counters.ptr, 0, sizeof(PtDevCounters) * PT_MAX_GROUPS, s->stream));
    PT_HIP(hipMemsetAsync(s->carry_header.ptr, 0, static_cast<size_t>(PT_MAX_GROUPS) * 4 * PT_QSTRIDE * sizeof(uint32_t), s->stream));
    PT_HIP(hipMemsetAsync(s->q_header.ptr, 0, s->q_header.count * sizeof(uint32_t), s->stream));
    PT_HIP(hipMemsetAsync(s->shade_wave_counters.ptr, 0, s->shade_wave_counters.count * sizeof(unsigned long long), s->stream));
    PT_HIP(hipMemsetAsync(s->trace_wave_counters.ptr, 0, s->trace_wave_counters.count * sizeof(unsigned long long), s->stream));
    PT_HIP(hipEventRecord(ev_init, s->stream));
    for(Group &gr : G) {
        PT_HIP(hipStreamWaitEvent(gr.st, ev_init, 0));
    }

    const bool timing = stats != nullptr;
    const int kEventPairs = 64; // launches per half of the event ring: a full half is read while the other one fills
    std::vector<hipEvent_t> ev;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    if(timing) {
        ev.resize(4 * 2 * kEventPairs);
        for(auto &e : ev) {
            PT_HIP(hipEventCreate(&e));
        }
        PT_HIP(hipEventCreate(&ev_begin));
        PT_HIP(hipEventCreate(&ev_end));
        PT_HIP(hipEventRecord(ev_begin, s->stream));
    }
    double trace_ms = 0.0, shade_ms = 0.0;
    uint64_t launches = 0;
    std::vector<std::pair<float, float>> trace_intervals; // [start, end) of every traversal launch, ms since ev_begin
    auto drain_events = [&](int first, int used) -> int {
        for(int i = first; i < first + used; i++) {
            float a = 0.0F, b = 0.0F, t0 = 0.0F;
            PT_HIP(hipEventSynchronize(ev[4 * i + 3]));
            PT_HIP(hipEventElapsedTime(&a, ev[4 * i + 0], ev[4 * i + 1]));
            PT_HIP(hipEventElapsedTime(&b, ev[4 * i + 2], ev[4 * i + 3]));
            PT_HIP(hipEventElapsedTime(&t0, ev_begin, ev[4 * i + 2]));
            shade_ms += a;
            trace_ms += b;
            trace_intervals.emplace_back(t0, t0 + b);
        }
        return PT_OK;
    };

    const int check_every = n <= 65536 ? 4 : env_int("PT_CHECK_EVERY", 8);
    uint64_t iterations = 0;
    int pending = 0;          // next free pair of the event ring
    bool other_half_used = false; // the half of the ring that `pending` is not in holds unread events
    uint64_t batch = 0;
    for(;;) {
        for(int k = 0; k < check_every; k++) {
            for(size_t g = 0; g < G.size(); g++) {
                Group &gr = G[g];
                if(gr.done) {
                    continue;
                }
                PtDevCounters *cnt = s->counters.ptr + g;
                if(timing) {
                    PT_HIP(hipEventRecord(ev[4 * pending + 0], gr.st));
                }
                const int parity = static_cast<int>(iterations & 1U);
                gr.cfg.parity = parity;
                // this launch's queue header, and the other one for the shading kernel to clear (both were zeroed before the loop)
                PtQueue q_now = gr.q;
                q_now.count = gr.q.count + static_cast<size_t>(parity) * 2 * PT_SHARDS * PT_QSTRIDE;
                q_now.head = q_now.count + PT_SHARDS * PT_QSTRIDE;
                q_now.next_header = gr.q.count + static_cast<size_t>(parity ^ 1) * 2 * PT_SHARDS * PT_QSTRIDE;
                pt_launch_shade(gr.st, s->dev, cam, opt, gr.P, q_now, gr.carry, parity, shard_mode, d_image, cnt);
                if(timing) {
                    PT_HIP(hipEventRecord(ev[4 * pending + 1], gr.st));
                    PT_HIP(hipEventRecord(ev[4 * pending + 2], gr.st));
                }
                pt_launch_trace(gr.st, s->dev, q_now, gr.carry, gr.P, gr.cfg, cnt);
                launches++;
                if(timing) {
                    PT_HIP(hipEventRecord(ev[4 * pending + 3], gr.st));
                    pending++;
                    if(pending == kEventPairs || pending == 2 * kEventPairs) {
                        // this half is full: read the OTHER half (recorded at least 64 launches ago) and continue into it
                        const int other = pending == kEventPairs ? kEventPairs : 0;
                        if(other_half_used) {
                            int rc = drain_events(other, kEventPairs);
                            if(rc != PT_OK) {
                                return rc;
                            }
                        }
                        other_half_used = true;
                        pending = other;
                    }
                }
            }
            iterations++;
        }
        // The counters of this batch are copied out behind it; the host looks at the copy of the PREVIOUS batch, so the next batch
        // is queued while the device still works on this one (a finished frame costs one batch of empty launches).
        bool all_done = true;
        const size_t slot = static_cast<size_t>(batch & 1U) * PT_MAX_GROUPS, prev_slot = static_cast<size_t>((batch + 1U) & 1U) * PT_MAX_GROUPS;
        for(size_t g = 0; g < G.size(); g++) {
            if(!G[g].done) {
                PT_HIP(hipMemcpyAsync(s->host_counters + slot + g, s->counters.ptr + g, sizeof(PtDevCounters), hipMemcpyDeviceToHost, G[g].st));
                PT_HIP(hipEventRecord(s->check_event[(batch & 1U) * PT_MAX_GROUPS + g], G[g].st));
            }
        }
        for(size_t g = 0; g < G.size(); g++) {
            if(!G[g].done) {
                if(batch == 0) {
                    all_done = false;
                    continue;
                }
                PT_HIP(hipEventSynchronize(s->check_event[((batch + 1U) & 1U) * PT_MAX_GROUPS + g]));
                const unsigned long long finished = s->host_counters[prev_slot + g].streams_done;
                G[g].done = finished >= G[g].count;
                // Endgame: once most streams have rendered all their pixels the launches are small, and suspending the long walks
                // of the remaining streams only multiplies the number of (fixed-cost) iterations they need.
                if(finished * 100ULL >= static_cast<unsigned long long>(G[g].count) * static_cast<unsigned long long>(endgame_percent)) {
                    G[g].cfg.max_steps = 0x7fffffff;
                    G[g].cfg.drain_lanes = 0;
                }
            }
            all_done = all_done && G[g].done;
        }
        batch++;
        PT_HIP(hipGetLastError());
        if(all_done) {
            break;
        }
        if(iterations > (1ULL << 40)) {
            return fail(PT_ERR_HIP, "wavefront loop did not terminate");
        }
    }
    // the main stream continues after every group
    for(Group &gr : G) {
        PT_HIP(hipEventRecord(ev_init, gr.st));
        PT_HIP(hipStreamWaitEvent(s->stream, ev_init, 0));
    }
    (void)hipEventDestroy(ev_init);
    if(timing) {
        int rc = PT_OK;
        if(other_half_used) {
            rc = drain_events(pending < kEventPairs ? kEventPairs : 0, kEventPairs);
        }
        if(rc == PT_OK) {
            const int first = pending < kEventPairs ? 0 : kEventPairs;
            rc = drain_events(first, pending - first);
        }
        if(rc != PT_OK) {
            return rc;
        }
        PT_HIP(hipEventRecord(ev_end, s->stream));
        PT_HIP(hipEventSynchronize(ev_end));
        float total = 0.0F;
        PT_HIP(hipEventElapsedTime(&total, ev_begin, ev_end));
        PtDevCounters c{};
        {
            std::vector<unsigned long long> shade_slots(s->shade_wave_counters.count), trace_slots(s->trace_wave_counters.count);
            PT_HIP(hipMemcpy(shade_slots.data(), s->shade_wave_counters.ptr, shade_slots.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            PT_HIP(hipMemcpy(trace_slots.data(), s->trace_wave_counters.ptr, trace_slots.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            for(size_t i = 0; i + 1 < shade_slots.size(); i += 2) {
                c.samples += shade_slots[i];
                c.vertices += shade_slots[i + 1];
            }
            unsigned long long diag[4] = {0, 0, 0, 0};
            for(size_t i = 0; i + 7 < trace_slots.size(); i += 8) {
                c.node_visits += trace_slots[i];
                c.leaf_tests += trace_slots[i + 1];
                c.rays += trace_slots[i + 2];
                c.shadow_rays += trace_slots[i + 3];
                for(int k = 0; k < 4; k++) {
                    diag[k] += trace_slots[i + 4 + k];
                }
            }
            if(env_int("PT_DEBUG", 0) != 0) {
                std::fprintf(stderr, "[pt] trace diagnostics: wave steps %llu (%.1f lanes of 64 on an inner node per step), leaf phases %llu (%.1f leaves each), refills %llu, suspended walks %llu\n",
                             diag[0], diag[0] ? static_cast<double>(c.node_visits) / static_cast<double>(diag[0]) : 0.0, diag[1],
                             diag[1] ? static_cast<double>(c.leaf_tests) / static_cast<double>(diag[1]) : 0.0, diag[2], diag[3]);
            }
        }
        stats->samples = c.samples;
        stats->rays_traced = c.rays;
        stats->shadow_rays_traced = c.shadow_rays;
        stats->node_visits = c.node_visits;
        stats->leaf_tests = c.leaf_tests;
        stats->vertices = c.vertices;
        stats->iterations = launches;
        stats->trace_ms = trace_ms;
        {
            // union of the traversal launches' intervals (the groups' launches overlap)
            std::sort(trace_intervals.begin(), trace_intervals.end());
            double busy = 0.0;
            float open_from = 0.0F, open_to = -1.0F;
            for(const auto &iv : trace_intervals) {
                if(iv.first > open_to) {
                    busy += open_to > open_from ? static_cast<double>(open_to - open_from) : 0.0;
                    open_from = iv.first;
                    open_to = iv.second;
                }
                else {
                    open_to = std::max(open_to, iv.second);
                }
            }
            busy += open_to > open_from ? static_cast<double>(open_to - open_from) : 0.0;
            stats->trace_busy_ms = busy;
            stats->groups = G.size();
        }
        stats->shade_ms = shade_ms;
        stats->total_ms = total;
        for(auto &e : ev) {
            (void)hipEventDestroy(e);
        }
        (void)hipEventDestroy(ev_begin);
        (void)hipEventDestroy(ev_end);
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
    // 8 stack entries per lane in LDS (16 KB per workgroup) let four workgroups share a CU; deeper walks use the HBM spill area
    int stack_lds = env_int("PT_STACK_LDS", 8);
    cfg.stack_lds = stack_lds == 8 ? 8 : 16;
    cfg.rows = std::min(std::max(env_int("PT_ROWS", 4), 1), PT_MAX_ROWS);
    cfg.lds_bytes = pt_path_lds_bytes(cfg.stack_lds, cfg.rows, cfg.in_lds ? s->dev.n_lds_pairs : 0U, cfg.in_lds ? s->dev.n_lds_tris : 0U);
    const int per_cu = pt_path_blocks_per_cu(cfg.stack_lds, cfg.in_lds, cfg.lds_bytes);
    const int limit = env_int("PT_BLOCKS_PER_CU", 0);
    s->path_blocks_per_cu = (limit > 0 && limit < per_cu) ? limit : per_cu;
    cfg.spill_depth = s->depth > static_cast<uint32_t>(cfg.stack_lds) ? s->depth - static_cast<uint32_t>(cfg.stack_lds) : 1U;
    cfg.refill_idle = std::min(std::max(env_int("PT_REFILL_IDLE", 12), 1), 64);
    cfg.min_ready = std::min(std::max(env_int("PT_MIN_READY", 32), 1), 64 * PT_MAX_ROWS);
    cfg.burst_steps = std::min(std::max(env_int("PT_BURST", 12), 1), 64);
    cfg.leaf_min = std::min(std::max(env_int("PT_LEAF_MIN", 8), 1), 64);
    if(env_int("PT_DEBUG", 0) != 0) {
        std::fprintf(stderr, "[pt] path kernel: %d CUs x %d workgroups, %d rows of slots per wavefront, stack_lds %d, scene %s, lds %zu B, spill depth %u\n", s->cu_count,
                     s->path_blocks_per_cu, cfg.rows, cfg.stack_lds, cfg.in_lds ? "in LDS" : "in HBM", cfg.lds_bytes, cfg.spill_depth);
    }
    return PT_OK;
}

// Grid and slot rows for n streams, and the buffers they need.
int ensure_path_workspace(pt_scene *s, uint32_t n, PtPathConfig *out_cfg) {
    int rc = setup_path(s);
    if(rc != PT_OK) {
        return rc;
    }
    PtPathConfig cfg = s->path_cfg;
    const uint32_t max_grid = static_cast<uint32_t>(s->cu_count) * static_cast<uint32_t>(s->path_blocks_per_cu);
    const uint32_t grid = std::max<uint32_t>(1U, std::min<uint32_t>(max_grid, (n + 255U) / 256U)); // every wavefront gets at least one row of 64 streams
    const uint32_t waves = grid * 4U;
    const uint32_t rows = std::min<uint32_t>(static_cast<uint32_t>(cfg.rows), std::max<uint32_t>(1U, (n + waves * 64U - 1U) / (waves * 64U)));
    const uint32_t total = waves * rows * 64U;
    const uint32_t rays_per_slot = 1U + s->dev.n_lights + s->dev.n_object_samples;
    const uint32_t cap = rows * 64U * rays_per_slot;
    cfg.grid = static_cast<int>(grid);
    cfg.rows = static_cast<int>(rows);
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
    PT_HIP(s->sl_nee_mask.ensure(total));
    PT_HIP(s->sl_nee.ensure(static_cast<size_t>(total) * std::max<uint32_t>(rays_per_slot - 1U, 1U)));
    PT_HIP(s->sl_est.ensure(total));
    PT_HIP(s->sl_cand.ensure(static_cast<size_t>(total) * PT_MAX_CANDIDATES));
    PT_HIP(s->lq_ray_o.ensure(static_cast<size_t>(waves) * cap));
    PT_HIP(s->lq_ray_d.ensure(static_cast<size_t>(waves) * cap));
    PT_HIP(s->path_spill.ensure(static_cast<size_t>(waves) * 64U * cfg.spill_depth));
    PT_HIP(s->path_wave_counters.ensure(static_cast<size_t>(waves) * 8U));
    PT_HIP(s->walk_save.ensure(static_cast<size_t>(waves) * 64U * PT_WALK_SAVE_WORDS));
    PT_HIP(s->pull_counter.ensure(64));
    PT_HIP(s->counters.ensure(PT_MAX_GROUPS));
    cfg.spill = s->path_spill.ptr;
    cfg.walk_save = s->walk_save.ptr;
    cfg.wave_counters = s->path_wave_counters.ptr;
    s->path_slots = total;
    s->path_waves = waves;
    s->path_cap = cap;
    *out_cfg = cfg;
    return PT_OK;
}

// Render the streams described by T (device pointers) with one launch on the scene's stream.  With a progress function the host polls
// the count of finished tiles (pinned memory, written by the kernel) while the launch runs and reports every step from the calling thread.
int run_path(pt_scene *s, const PtDevCamera &cam, const PtDevOptions &opt, PtStreams T, float4 *d_image, pt_stats *stats, pt_progress_fn progress, void *progress_user) {
    PtPathConfig cfg;
    int rc = ensure_path_workspace(s, T.n, &cfg);
    if(rc != PT_OK) {
        return rc;
    }
    hipStream_t st = s->stream;
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
    S.nee_mask = s->sl_nee_mask.ptr;
    S.nee = reinterpret_cast<float4 *>(s->sl_nee.ptr);
    S.est = s->sl_est.ptr;
    S.cand = s->sl_cand.ptr;
    PtLocalQueue Q{};
    Q.ray_o = reinterpret_cast<float4 *>(s->lq_ray_o.ptr);
    Q.ray_d = reinterpret_cast<float4 *>(s->lq_ray_d.ptr);
    Q.cap = s->path_cap;
    T.next = s->pull_counter.ptr;
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
    pt_launch_path(st, s->dev, cam, opt, S, T, Q, cfg, d_image, s->counters.ptr);
    PT_HIP(hipGetLastError());
    PT_HIP(hipEventRecord(ev_end.e, st));
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
        stats->node_visits = sum[0];
        stats->leaf_tests = sum[1];
        stats->rays_traced = sum[2];
        stats->shadow_rays_traced = sum[3];
        stats->samples = sum[6];
        stats->vertices = sum[7];
        stats->iterations = 1;
        stats->trace_ms = ms;
        stats->trace_busy_ms = ms;
        stats->shade_ms = 0.0;
        stats->total_ms = ms;
        stats->groups = 1;
        if(env_int("PT_DEBUG", 0) != 0) {
            std::fprintf(stderr, "[pt] path kernel: %.2f ms, grid %d x 256, %d rows; wave steps %llu (%.1f lanes of 64 busy per step), shading passes %llu, rays %llu\n", ms, cfg.grid,
                         cfg.rows, sum[4], sum[4] ? static_cast<double>(sum[0] + sum[1]) / static_cast<double>(sum[4]) : 0.0, sum[5], sum[2]);
        }
    }
    return PT_OK;
}

uint32_t choose_groups(pt_scene *s, uint32_t n) {
    // Measured: three groups of streams on three HIP streams, so that one group's shading (and the thin end of its
    // traversal launches) overlaps the others' traversal.  Scenes that live in LDS: +23 % (two groups +13 %, four -3 %).  HBM-resident
    // trees: +12 %, provided every group's persistent traversal grid is limited to two workgroups per CU (run_wavefront) -- with
    // full-size grids the kernels of the groups only queue up behind each other (-4 %).
    const int by_scene = 3;
    const int want = std::min(std::max(env_int("PT_GROUPS", by_scene), 1), PT_MAX_GROUPS);
    // a group should still fill the chip's lanes on its own now and then: at least 128 K streams per group
    const uint32_t by_size = std::max<uint32_t>(1U, n / 131072U);
    uint32_t groups = std::min<uint32_t>(static_cast<uint32_t>(want), by_size);
    if(groups > 1) {
        // no more groups than streams that really run side by side (groups sharing a stream would only queue up)
        if(s->group_stream[groups - 1] == nullptr && pick_concurrent_streams(s, groups) != PT_OK) {
            return 1;
        }
        groups = std::max<uint32_t>(1U, std::min<uint32_t>(groups, s->concurrent_streams));
    }
    return groups;
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
    return pt_host_pixel_seed(base_seed, x, y);
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
    std::vector<uint32_t> level_begin; // pair slots of the tree's levels (for the two-level records)
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
        PT_HIP(s->tris.ensure(3 * static_cast<size_t>(d->n_triangles) + 1)); // + 1: a traversal step reads 64 bytes of a 48-byte record
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
            return fail(PT_ERR_UNSUPPORTED, "BVH deeper than 128 levels");
        }
        n_pairs = built.n_pairs;
        root_ref = built.root_ref;
        level_begin = built.level_begin;
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
            return fail(PT_ERR_UNSUPPORTED, "BVH deeper than 128 levels");
        }
        ptb::FlatBvh flat = ptb::flatten_breadth_first(s->tree, leaf_ref, align_siblings);
        n_pairs = flat.n_pairs;
        root_ref = flat.root_ref;
        level_begin = flat.level_begin;
        for(int k = 0; k < 3; k++) {
            root_lo[k] = flat.root_box.lo[k];
            root_hi[k] = flat.root_box.hi[k];
        }
        ptb::leaves_depth_first(s->tree, dfs);
        s->build_ms[0] = ms_since(t_begin);
        const auto t_upload = clock::now();

        std::vector<F4> tris(3 * static_cast<size_t>(d->n_triangles) + 1, F4{0.0F, 0.0F, 0.0F, 0.0F}), shade(8 * static_cast<size_t>(d->n_triangles), F4{0.0F, 0.0F, 0.0F, 0.0F});
        for(uint32_t t = 0; t < d->n_triangles; t++) {
            const float *p = d->tri_pos + 9 * static_cast<size_t>(t);
            const Vec3 a = ld(p), b = ld(p + 3), c = ld(p + 6);
            const Vec3 ab = sub(b, a), ac = sub(c, a);
            const uint32_t obj_cull = s->tri_obj[t] | (d->tri_cull[t] != 0 ? 0x80000000U : 0U);
            tris[3 * static_cast<size_t>(t) + 0] = {a.x, a.y, a.z, ab.x};
            tris[3 * static_cast<size_t>(t) + 1] = {ab.y, ab.z, ac.x, ac.y};
            tris[3 * static_cast<size_t>(t) + 2] = {ac.z, from_bits(d->tri_material[t]), from_bits(obj_cull), 0.0F};
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
                shade[8 * static_cast<size_t>(t) + k] = tris[3 * static_cast<size_t>(t) + k];
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
        return fail(PT_ERR_UNSUPPORTED, "more than 8 light samples per path vertex");
    }

    // ---- upload --------------------------------------------------------------------------------------------------------------
    PT_HIP(s->materials.upload(materials));
    PT_HIP(s->lights.upload(lights));
    PT_HIP(s->emis.upload(emis));
    PT_HIP(s->emis_cdf.upload(cdf));

    PtDevScene &dev = s->dev;
    dev.pairs = reinterpret_cast<const float4 *>(s->pairs.ptr);
    dev.tris = reinterpret_cast<const float4 *>(s->tris.ptr);
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
    dev.n_object_samples = static_cast<uint32_t>(object_sample_count);
    // LDS staging: a scene whose whole tree and triangle records fit in 24 KiB lives in LDS entirely; for larger scenes the
    // traversal kernel is bound by instruction issue, not by node latency, and an LDS copy of the top of the tree only costs
    // occupancy (measured: profiles/), so it is off unless PT_LDS_PAIRS asks for it.
    const size_t small_bytes = static_cast<size_t>(n_pairs) * 64 + static_cast<size_t>(d->n_triangles) * 48;
    if(small_bytes <= 24 * 1024 && env_int("PT_LDS_SMALL", 1) != 0) {
        dev.n_lds_pairs = n_pairs;
        dev.n_lds_tris = d->n_triangles;
    }
    else {
        dev.n_lds_pairs = std::min(n_pairs, static_cast<uint32_t>(std::max(env_int("PT_LDS_PAIRS", 0), 0)));
        dev.n_lds_tris = 0;
    }

    dev.quads = nullptr;
    dev.n_quads = 0;
    if(dev.n_lds_pairs == 0 && n_pairs > 0 && (root_ref & PT_REF_LEAF) == 0 && env_int("PT_WIDE", 0) != 0) {
        // HBM-resident tree: two-level records, one dependent fetch per two levels of a walk (pt_trace.hip, LDS_MODE 3)
        float4 *quads = nullptr;
        uint32_t n_quads = 0;
        PT_HIP(pt_build_quads(s->stream, reinterpret_cast<const float4 *>(s->pairs.ptr), level_begin, &quads, &n_quads));
        s->quads.ptr = reinterpret_cast<F4 *>(quads);
        s->quads.count = 12 * static_cast<size_t>(n_quads);
        dev.quads = quads;
        dev.n_quads = n_quads;
    }

    int rc = setup_trace(s.get());
    if(rc != PT_OK) {
        return rc;
    }
    if(const char *kernel = std::getenv("PT_KERNEL")) {
        s->use_path = std::strcmp(kernel, "wavefront") != 0; // A/B against round 1's two-kernel wavefront loop
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
    if(scene->walk_hist.ptr != nullptr) {
        uint32_t h[64];
        if(hipMemcpy(h, scene->walk_hist.ptr, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
            std::fprintf(stderr, "[pt] inner-node steps per walk (bucket b: 2^(b-1) <= steps < 2^b):");
            for(int b = 0; b < 16; b++) {
                std::fprintf(stderr, " %u", h[b]);
            }
            std::fprintf(stderr, "\n");
        }
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
        std::vector<F4> pairs(scene->pairs.count);
        if(hipSetDevice(scene->device) != hipSuccess ||
           hipMemcpy(pairs.data(), scene->pairs.ptr, pairs.size() * sizeof(F4), hipMemcpyDeviceToHost) != hipSuccess) {
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
                const uint32_t idx = it.ref & PT_REF_INDEX;
                obj.push_back(static_cast<int32_t>((it.ref & PT_REF_SPHERE) != 0 ? scene->sph_obj[idx] : scene->tri_obj[idx]));
            }
            else {
                obj.push_back(-1);
                const float *q = &pairs[4 * static_cast<size_t>(it.ref)].x;
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
    if(s->use_path) {
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
            out_obj[i] = (t < 0.0F || ref == PT_REF_NONE) ? -1 : static_cast<int32_t>((ref & PT_REF_SPHERE) ? s->sph_obj[ref & PT_REF_INDEX] : s->tri_obj[ref & PT_REF_INDEX]);
        }
        return PT_OK;
    }
    int rc = ensure_workspace(s, n32, 1U + s->dev.n_lights + s->dev.n_object_samples);
    if(rc != PT_OK) {
        return rc;
    }
    PT_HIP(s->batch_rays.ensure(6 * n));
    PtPaths P = make_paths(s, n32);
    PtQueue q = make_queue(s);
    if(q.shard_capacity < (n32 + PT_SHARDS - 1) / PT_SHARDS) {
        return fail(PT_ERR_NOMEM, "queue too small for batch");
    }
    hipStream_t st = s->stream;
    PT_HIP(hipMemcpyAsync(s->batch_rays.ptr, rays, 6 * n * sizeof(float), hipMemcpyHostToDevice, st));
    PT_HIP(hipMemsetAsync(s->q_header.ptr, 0, 2 * PT_SHARDS * PT_QSTRIDE * sizeof(uint32_t), st));
    PT_HIP(hipMemsetAsync(s->counters.ptr, 0, sizeof(PtDevCounters), st));
    pt_launch_batch_rays(st, s->batch_rays.ptr, n32, q);
    PtTraceConfig trace_cfg = s->trace_cfg;
    trace_cfg.spill = s->spill.ptr;
    trace_cfg.grid = static_cast<int>(std::max<uint32_t>(1U, std::min<uint32_t>(static_cast<uint32_t>(trace_cfg.grid), (n32 + 255U) / 256U)));
    trace_cfg.max_steps = 0x7fffffff;
    trace_cfg.drain_lanes = 0;
    trace_cfg.parity = 0;
    trace_cfg.wave_counters = s->trace_wave_counters.ptr;
    PT_HIP(hipMemsetAsync(s->carry_header.ptr, 0, 4 * PT_QSTRIDE * sizeof(uint32_t), st));
    PtCarry carry = make_carry(s, 0);
    carry.cap = 0;
    pt_launch_trace(st, s->dev, q, carry, P, trace_cfg, s->counters.ptr);
    std::vector<uint2> hits(n);
    PT_HIP(hipMemcpyAsync(hits.data(), s->hit.ptr, n * sizeof(uint2), hipMemcpyDeviceToHost, st));
    PT_HIP(hipStreamSynchronize(st));
    PT_HIP(hipGetLastError());
    for(size_t i = 0; i < n; i++) {
        const float t = from_bits(hits[i].x);
        const uint32_t ref = hits[i].y;
        out_t[i] = t;
        if(t < 0.0F || ref == PT_REF_NONE) {
            out_obj[i] = -1;
        }
        else if(ref & PT_REF_SPHERE) {
            out_obj[i] = static_cast<int32_t>(s->sph_obj[ref & PT_REF_INDEX]);
        }
        else {
            out_obj[i] = static_cast<int32_t>(s->tri_obj[ref & PT_REF_INDEX]);
        }
    }
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
    if(s->use_path) {
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
        PT_HIP(hipStreamSynchronize(st));
        return PT_OK;
    }
    const uint32_t groups = choose_groups(s, n32);
    rc = ensure_workspace(s, n32, 1U + s->dev.n_lights + s->dev.n_object_samples, groups);
    if(rc != PT_OK) {
        return rc;
    }
    // pixels not covered by a stream keep the caller's values
    PT_HIP(hipMemcpyAsync(s->image.ptr, out_image, pixels * sizeof(F4), hipMemcpyHostToDevice, st));
    PT_HIP(hipMemcpyAsync(s->rect.ptr, rects.data(), n * sizeof(int4), hipMemcpyHostToDevice, st));
    PT_HIP(hipMemcpyAsync(s->rng.ptr, states.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    pt_launch_init_streams(st, make_paths(s, n32));
    rc = run_wavefront(s, cam, opt, n32, groups, reinterpret_cast<float4 *>(s->image.ptr), stats);
    if(rc != PT_OK) {
        return rc;
    }
    PT_HIP(hipMemcpyAsync(out_image, s->image.ptr, pixels * sizeof(F4), hipMemcpyDeviceToHost, st));
    if(out_states != nullptr) {
        PT_HIP(hipMemcpyAsync(out_states, s->rng.ptr, n * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    }
    PT_HIP(hipStreamSynchronize(st));
    return PT_OK;
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
    if(s->use_path) {
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
        return run_path(s, cam, opt, T, d_image, stats, progress, progress_user);
    }
    const uint32_t groups = choose_groups(s, n32);
    // Stream slots are laid out tile after tile and the stream groups are contiguous ranges of slots: with the tiles in the caller's
    // (row-major) order a group is a horizontal band of the frame.  That is deliberate: the bands differ in cost, so the groups reach
    // the thin end of their work at different times and one group's tail overlaps the others' full launches (dealing the tiles out
    // to the groups in turn was measured: -2 %).
    std::vector<int4> rects(n_tiles);
    std::vector<uint32_t> offsets(n_tiles);
    {
        uint64_t at = 0;
        for(size_t k = 0; k < n_tiles; k++) {
            const pt_tile &t = tiles[k];
            rects[k] = make_int4(t.x, t.y, t.w, t.h);
            offsets[k] = static_cast<uint32_t>(at);
            at += static_cast<uint64_t>(t.w) * static_cast<uint64_t>(t.h);
        }
    }
    rc = ensure_workspace(s, n32, 1U + s->dev.n_lights + s->dev.n_object_samples, groups);
    if(rc != PT_OK) {
        return rc;
    }
    PT_HIP(s->tiles.ensure(n_tiles));
    PT_HIP(s->tile_offset.ensure(n_tiles));
    hipStream_t st = s->stream;
    PT_HIP(hipMemcpyAsync(s->tiles.ptr, rects.data(), n_tiles * sizeof(int4), hipMemcpyHostToDevice, st));
    PT_HIP(hipMemcpyAsync(s->tile_offset.ptr, offsets.data(), n_tiles * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    pt_launch_init_tiles(st, make_paths(s, n32), s->tiles.ptr, s->tile_offset.ptr, static_cast<uint32_t>(n_tiles), base_seed);
    PT_HIP(hipStreamSynchronize(st)); // rects/offsets are stack-owned host vectors
    return run_wavefront(s, cam, opt, n32, groups, d_image, stats);
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
    PT_HIP(hipStreamSynchronize(s->stream));
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
