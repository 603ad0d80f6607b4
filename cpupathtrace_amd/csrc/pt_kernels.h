// pt_kernels.h -- launch interface between the host side of libpathtrace_hip.so (pt_api.cpp) and its kernels.
#ifndef PT_KERNELS_H
#define PT_KERNELS_H

#include <hip/hip_runtime.h>

#include "pt_types.h"

#define PT_QSTRIDE 64 /* every queue counter sits alone in a 256-byte line: device-scope atomics on words that share a line serialise */
#define PT_QCHUNK 256 /* rays a wavefront reserves per dequeue atomic */
#define PT_SHARDS 8 /* ray-queue shards = XCDs; a workgroup appends to and first drains shard blockIdx.x % 8 */

// Wavefront state of all streams in flight (structure of arrays, one element per stream slot).
struct PtPaths {
    uint32_t n;            // stream slots
    int4 *rect;            // WorkItem rectangle x, y, w, h
    uint64_t *rng;         // xorshift state of the stream's engine
    int32_t *cursor;       // index of the current pixel inside the rectangle (row-major)
    uint32_t *flags;       // PT_F_* bits
    float4 *ray_o;         // current ray origin; w = contribution_unweighted (worker.cpp:38)
    float4 *ray_d;         // current ray direction
    float4 *spectrum;      // sample_spectrum (worker.cpp:41)
    float4 *out;           // out_spectrum (worker.cpp:42)
    double *divisor;       // sample_divisor (worker.cpp:39)
    double *bounce_pd;     // sample_bounce_pd (worker.cpp:40)
    int32_t *path_length;  // worker.cpp:43
    float4 *nee;           // [PT_MAX_NEE][nee_stride] weighed_spectrum of the pending shadow rays (worker.cpp:97)
    uint32_t nee_stride;   // slots between two light-sample planes of `nee` (the workspace size; >= n for a group view)
    uint32_t *nee_mask;    // bit j: light sample j of the last vertex contributes if its shadow ray is unoccluded
    PtEstimator *est;      // per-pixel estimator (worker.cpp:172-192)
    PtCandidate *cand;     // [n][PT_MAX_CANDIDATES]
    // results of the traversal kernel
    uint2 *hit;            // [n] (bits t, ref) of the extension ray
    uint32_t *vis;         // [PT_MAX_NEE][nee_stride] 1 = shadow ray unoccluded, 0 = occluded, PT_VIS_PENDING = still walking
    unsigned long long *wave_counters; // [ceil(n / 64)][2] samples finished, vertices shaded (plain adds, one slot per wave)
};

#define PT_F_DONE 1u      /* the stream has rendered all its pixels */
#define PT_F_IN_FLIGHT 2u /* a path is in flight (otherwise the next invocation starts a sample) */
#define PT_F_HAS_EXT 4u   /* an extension (camera/bounce) ray was traced for it */
#define PT_F_COLLECTED 8u /* sample_collected (worker.cpp:37) */
#define PT_F_PIXEL 16u    /* the estimator of the current pixel is initialised */
#define PT_F_SAFE 64u     /* the estimator cannot stop at the sample in flight and another sample of the pixel follows (see pt_shade.hip) */
#define PT_F_OVERLAP 32u  /* the next sample's camera ray is already in flight while the previous sample waits for its last shadow rays */

// Ray queue: PT_SHARDS append-only segments of `shard_capacity` rays each.
struct PtQueue {
    float4 *ray_o;   // origin xyz, w = shadow threshold |to_light| - epsilon (worker.cpp:86) or unused
    float4 *ray_d;   // direction xyz, w = bits destination: bit 31 = shadow ray, low bits = slot (ext) or j * nee_stride + slot
    uint32_t *count; // [PT_SHARDS * PT_QSTRIDE] rays appended to shard s at count[s * PT_QSTRIDE]
    uint32_t *head;  // [PT_SHARDS * PT_QSTRIDE] rays dequeued from shard s at head[s * PT_QSTRIDE]
    uint32_t shard_capacity;
    uint32_t *next_header; // count and head words of the NEXT launch (the headers alternate): cleared by the shading kernel, or null
};

// Walks suspended by one traversal launch and resumed by the next one (two pools used alternately).  A launch gives every walk
// a budget of inner-node steps; the few walks that need thousands of steps (rays grazing the mesh) would otherwise keep the
// whole launch -- and with it every stream of the wavefront -- waiting for them.
struct PtCarry {
    float4 *ray_o;   // [2][cap] origin, shadow threshold
    float4 *ray_d;   // [2][cap] direction, destination
    uint4 *state;    // [2][cap] bits best_t, best_ref, bits t_max, current node
    uint32_t *sp;    // [2][cap] saved stack entries
    uint2 *stack;    // [2][cap][depth]
    uint32_t *count; // [2 * PT_QSTRIDE] walks stored in pool i at count[i * PT_QSTRIDE] (may exceed cap: the excess was not stored)
    uint32_t *head;  // [2 * PT_QSTRIDE] walks taken out of pool i
    uint32_t cap;
    uint32_t depth;
};

#define PT_DEST_SHADOW 0x80000000u
#define PT_DEST_NULL 0xffffffffu /* hole in the queue: reserved but not used */

struct PtTraceConfig {
    int grid;
    int stack_lds;        // stack entries per lane kept in LDS
    uint32_t spill_depth; // further entries per lane in HBM
    uint2 *spill;
    size_t lds_bytes;
    int refill_idle;      // a wavefront refills from the queue once this many of its 64 lanes are idle
    int leaf_min;         // leaf tests run once this many lanes stand on a leaf (or no lane has an inner node left)
    int lds_mode;         // 0: no LDS staging, 1: top of the tree in LDS, 2: whole tree + triangles in LDS
    uint32_t *walk_hist;  // optional [33] histogram of inner-node steps per finished walk (PT_WALK_HIST=1), else null
    int max_steps;        // inner-node steps a walk may take in one launch before it is suspended
    int parity;           // this launch resumes pool `parity` and suspends into pool `parity ^ 1`
    int chunk;            // rays a wavefront reserves per dequeue atomic
    int burst_steps;      // inner-node steps between two looks at the leaves / the queue
    int drain_lanes;      // with the queue empty, a wavefront with at most this many walks left suspends them (0: never)
    unsigned long long *wave_counters; // [grid * 4 waves][8] node visits, leaf tests, rays, shadow rays, wave steps, leaf phases, refills, suspended walks
};

void pt_launch_init_tiles(hipStream_t stream, PtPaths paths, const int4 *tiles, const uint32_t *tile_offset, uint32_t n_tiles, uint64_t base_seed);
void pt_launch_init_streams(hipStream_t stream, PtPaths paths);
// one wavefront that does nothing for `microseconds` (probe: do two HIP streams run their kernels at the same time?)
void pt_launch_spin(hipStream_t stream, uint32_t microseconds);
void pt_launch_shade(hipStream_t stream, const PtDevScene &scene, const PtDevCamera &camera, const PtDevOptions &options, PtPaths paths, PtQueue queue,
                     PtCarry carry, int parity, int shard_mode, float4 *image, PtDevCounters *counters);
void pt_launch_trace(hipStream_t stream, const PtDevScene &scene, PtQueue queue, PtCarry carry, PtPaths paths, const PtTraceConfig &cfg,
                     PtDevCounters *counters);
void pt_launch_batch_rays(hipStream_t stream, const float *rays6, uint32_t n, PtQueue queue);
int pt_trace_blocks_per_cu(int stack_lds, int lds_mode, size_t lds_bytes);


// ---- the persistent path kernel (pt_path.hip) ---------------------------------------------------------------------------------------
// One launch renders a whole set of streams.  Every wavefront of the grid is an independent renderer: it owns `rows` x 64 stream SLOTS,
// a ray queue of its own and its share of the per-slot path state, and alternates between shading the slots whose rays have come back
// and tracing the rays that produced -- without ever synchronising with another wavefront.  A slot that has finished its stream pulls
// the next one from a global counter.

#define PT_WALK_SAVE_WORDS 17
#define PT_MAX_ROWS 8      /* rows of 64 slots per wavefront */
#define PT_F_STREAM 128u   /* the slot holds a stream (flag bit, next to PT_F_*) */

// Path state of the slots (structure of arrays).  Slot index = (global wave index * rows + row) * 64 + lane, so that every access of a
// shading pass over one row is one coalesced run of 64 elements.
struct PtSlots {
    uint32_t total;        // slots of the whole grid (stride of the `nee` planes)
    uint32_t *stream;      // stream index held by the slot
    int4 *rect;            // its WorkItem rectangle
    int32_t *cursor;       // index of the current pixel inside the rectangle (row-major)
    uint64_t *rng;         // xorshift state of the stream's engine
    float4 *ray_o;         // current ray origin; w = contribution_unweighted (worker.cpp:38)
    float4 *ray_d;         // current ray direction
    float4 *spectrum;      // sample_spectrum (worker.cpp:41)
    float4 *out;           // out_spectrum (worker.cpp:42)
    double *divisor;       // sample_divisor (worker.cpp:39)
    double *bounce_pd;     // sample_bounce_pd (worker.cpp:40)
    int32_t *path_length;  // worker.cpp:43
    uint32_t *nee_mask;    // bit j: light sample j of the last vertex contributes if its shadow ray is unoccluded
    float4 *nee;           // [PT_MAX_NEE][total] weighed_spectrum of the pending shadow rays (worker.cpp:97)
    PtEstimator *est;      // per-pixel estimator (worker.cpp:172-192)
    PtCandidate *cand;     // [total][PT_MAX_CANDIDATES]
};

// The streams of one render call.  Either explicit (rect + engine state per stream: processItem calls) or the pixels of a tile list, each
// its own 1x1 stream seeded from (base_seed, x, y) (processJob): then rect and rng are null and stream i is pixel i of the tiles laid end
// to end.
struct PtStreams {
    uint32_t n;
    const int4 *rect;            // [n] or null
    uint64_t *rng;               // [n] engine state in, engine state out; or null
    const int4 *tiles;           // [n_tiles] x, y, w, h
    const uint32_t *tile_offset; // [n_tiles] first stream of each tile
    uint32_t n_tiles;
    uint64_t base_seed;
    uint32_t *next;              // global pull counter (alone in its cache line)
    uint32_t *tile_left;         // [n_tiles] pixels of the tile not yet finished, or null: no progress reporting
    uint32_t *tiles_done;        // HOST-visible count of finished tiles (pinned memory), or null
};

// Ray queues, one private ring per wavefront: entries [wave * cap, (wave + 1) * cap)
struct PtLocalQueue {
    float4 *ray_o;   // origin xyz, w = shadow threshold |to_light| - epsilon (worker.cpp:86) or unused
    float4 *ray_d;   // direction xyz, w = bits destination: bit 31 = shadow ray, bits 16..19 = light sample, bits 0..15 = slot of the wave
    uint32_t cap;    // rows * 64 * rays per slot
};

struct PtPathConfig {
    int grid;             // workgroups of 256 threads
    int rows;             // rows of 64 slots per wavefront (<= PT_MAX_ROWS)
    int stack_lds;        // traversal stack entries per lane kept in LDS
    uint32_t spill_depth; // further entries per lane in HBM
    uint2 *spill;
    uint32_t *walk_save;  // [PT_WALK_SAVE_WORDS][grid * 256]: where a lane parks its walk during a shading pass
    size_t lds_bytes;
    int in_lds;           // whole tree + triangle records staged in LDS (small scenes)
    int refill_idle;      // idle lanes that make a wavefront refill from its queue (or shade when the queue is empty)
    int min_ready;        // slots that must be ready before a wavefront with walks in progress stops tracing to shade
    int burst_steps;      // traversal steps between two looks at the queue
    int leaf_min;         // lanes that must stand on a leaf before the leaf code runs (while other lanes still have nodes to visit)
    unsigned long long *wave_counters; // [grid * 4 waves][8] node visits, leaf tests, rays, shadow rays, wave steps, shading passes, samples, vertices
};

void pt_launch_path(hipStream_t stream, const PtDevScene &scene, const PtDevCamera &camera, const PtDevOptions &options, PtSlots slots, PtStreams streams,
                    PtLocalQueue queue, const PtPathConfig &cfg, float4 *image, PtDevCounters *counters);
int pt_path_blocks_per_cu(int stack_lds, int in_lds, size_t lds_bytes);
size_t pt_path_lds_bytes(int stack_lds, int rows, uint32_t n_lds_pairs, uint32_t n_lds_tris);
// Scene::getIntersection for n rays (6 floats each): out[i] = (bits t, ref)
void pt_launch_closest(hipStream_t stream, const PtDevScene &scene, const float *rays6, uint32_t n, uint2 *out, const PtPathConfig &cfg);

uint64_t pt_host_pixel_seed(uint64_t base_seed, int32_t x, int32_t y);

#endif
