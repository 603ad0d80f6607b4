// pt_kernels.h -- launch interface between the host side of libpathtrace_hip.so (pt_api.cpp) and the kernels of pt_path.hip.
#ifndef PT_KERNELS_H
#define PT_KERNELS_H

#include <hip/hip_runtime.h>

#include "pt_types.h"

// flag bits of a stream slot
#define PT_F_DONE 1u      /* the slot will never hold a stream again */
#define PT_F_IN_FLIGHT 2u /* a path is in flight (otherwise the next invocation starts a sample) */
#define PT_F_HAS_EXT 4u   /* an extension (camera/bounce) ray was traced for it */
#define PT_F_COLLECTED 8u /* sample_collected (worker.cpp:37) */
#define PT_F_PIXEL 16u    /* the estimator of the current pixel is initialised */
#define PT_F_SAFE 64u     /* the estimator cannot stop at the sample in flight and another sample of the pixel follows (see pt_path.hip) */
#define PT_F_OVERLAP 32u  /* the next sample's camera ray is already in flight while the previous sample waits for its last shadow rays */

#define PT_DEST_SHADOW 0x80000000u
#define PT_DEST_NULL 0xffffffffu /* hole in the queue: reserved but not used */

// ---- the persistent path kernel (pt_path.hip) ---------------------------------------------------------------------------------------
// One launch renders a whole set of streams.  Every wavefront of the grid is an independent renderer: it owns `rows` x 64 stream SLOTS,
// a ray queue of its own and its share of the per-slot path state, and alternates between shading the slots whose rays have come back
// and tracing the rays that produced -- without ever synchronising with another wavefront.  A slot that has finished its stream pulls
// the next one from a global counter.

// Small tables the shading pass reads per lane live in LDS when they have at most PT_LDS_TABLE_MAX entries: the emitters' CDF (4 B),
// sampling records (64 B) and shading records (96 B: the vertex normals of an emissive triangle), and the materials (64 B).
#define PT_LDS_TABLE_MAX 16
#define PT_LDS_TABLE_BYTES (PT_LDS_TABLE_MAX * (4 + 64 + 96 + 64))
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
    float4 *nee;           // [light samples per vertex][total] weighed_spectrum of the pending shadow rays (worker.cpp:97)
    uint32_t *nee_mask;    // which of them wait for their shadow ray (or needed none): bit per light sample
    uint32_t *cost;        // wave steps the stream's rays spent in traversal so far (only when PtStreams::cost is wanted)
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
    uint32_t first_total;        // streams 0 .. first_total-1 belong to the wavefronts' slots from the start (wavefront * slots_per_wave + slot)
    uint32_t first_lanes;        // streams per piece of the first round (64 = a row of slots; 32, 16: parts of a row)
    uint32_t first_shift;        // sideways steps per piece
    uint32_t n_waves, first_spread; // first_spread: row r of wavefront w starts on the 64-stream chunk r * n_waves + w ...
    uint32_t tiles_per_row, chunks_per_tile; // ... moved sideways by r * tiles_per_row / 4 tiles in a regular tile grid (0 = no grid)
    uint32_t *next;              // global pull counter (alone in its cache line)
    uint32_t *tile_left;         // [n_tiles] pixels of the tile not yet finished, or null: no progress reporting
    uint32_t *tiles_done;        // HOST-visible count of finished tiles (pinned memory), or null
    uint32_t *cost;              // [n] out, or null: traversal steps of the wavefront that passed while a ray of the stream was walking (summed over its rays)
    const uint32_t *place;       // [n_waves * slots per wave] or null: the stream that starts in every slot (0xffffffff: none) instead of the arithmetic first round
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
    int slots_per_wave;   // slots a wavefront really uses (<= rows * 64; fewer for small jobs)
    int wide;             // more than 8 light samples per path vertex: the slots use the 64-bit word (pt_path.hip, SlotWord)
    int stack_lds;        // traversal stack entries per lane kept in LDS
    uint32_t spill_depth; // further entries per lane in HBM
    uint2 *spill;
    uint32_t *walk_save;  // [PT_WALK_SAVE_WORDS][grid * 256]: where a lane parks its walk during a shading pass
    size_t lds_bytes;
    int in_lds;           // whole tree + triangle records staged in LDS (small scenes)
    int refill_idle;      // idle lanes that make a wavefront refill from its queue (or shade when the queue is empty)
    int min_ready;        // slots that must be ready before a wavefront with walks in progress stops tracing to shade ...
    int ready_shift;      // ... or (slots that still hold or may get a stream) >> ready_shift, if that is less: a wavefront whose last streams are running shades them as they come
    int pass_q_low;       // a pass may also start while the ring still holds rays, if it holds at most this many ...
    int early_ready;      // ... and this many slots are ready (0 = never: a pass waits for the ring to run empty)
    int compact_passes;   // a pass whose ready slots fit fewer chunks of 64 than they occupy rows runs over a list of them (pt_path.hip)
    int debug_lanes;      // diagnostic: lanes of a wavefront that take rays (64)
    int burst_steps;      // traversal steps between two looks at the queue
    int first_lanes;      // slots per piece of the first round of streams (slots_per_wave is a multiple of it)
    int leaf_min;         // lanes that must stand on a leaf before the leaf code runs (while other lanes still have nodes to visit)
    unsigned long long *wave_counters; // [grid * 4 waves][8] node visits, leaf tests, rays, shadow rays, wave steps, shading passes, samples, vertices
};

// Everything the path kernel is told, in DEVICE memory: the kernel takes one pointer and reads what it needs where it needs it -- the
// traversal loop a handful of values once, a shading pass the rest each time it runs -- instead of holding ~100 scalar registers of
// kernel arguments live through the traversal loop (they do not fit: the compiler parked them in vector-register lanes there).
struct PtPathArgs {
    PtDevScene sc;
    PtDevCamera cam;
    PtDevOptions opt;
    PtSlots S;
    PtStreams T;
    PtLocalQueue Q;
    int rows, slots_per_wave, refill_idle, min_ready, burst_steps, leaf_min, ready_shift, pass_q_low, early_ready, compact_passes, debug_lanes;
    uint2 *spill;
    uint32_t spill_depth;
    uint32_t save_stride;
    uint32_t *walk_save;
    float4 *image;
    PtDevCounters *counters;
    unsigned long long *wave_counters;
};

// fills *host_args (which must stay valid until the launch has been issued), copies it to d_args on `stream` and launches
void pt_launch_path(hipStream_t stream, const PtDevScene &scene, const PtDevCamera &camera, const PtDevOptions &options, PtSlots slots, PtStreams streams,
                    PtLocalQueue queue, const PtPathConfig &cfg, float4 *image, PtDevCounters *counters, PtPathArgs *host_args, PtPathArgs *d_args);
int pt_path_blocks_per_cu(const PtPathConfig &cfg); // resident workgroups per CU of the instantiation cfg selects (wide, in_lds, stack_lds) with cfg.lds_bytes
size_t pt_path_lds_bytes(int wide, int rows, int stack_lds, uint32_t n_lds_pairs, uint32_t n_lds_leaf_records); // leaf records: triangles + 1 spare + spheres, 0 = scene not in LDS
int pt_path_stack_lds(int in_lds, size_t lds_bytes_with_default_window); // entries of the stack window: 8, or 4 for a scene in LDS that would not leave room for four workgroups per CU
// Scene::getIntersection for n rays (6 floats each): out[i] = (bits t, ref)
// diagnostic (tools/replay_probe.py): the traversal alone over the rays a render left in its rings; returns the resident workgroups per CU
int pt_launch_replay(hipStream_t stream, const PtDevScene &scene, const PtLocalQueue &Q, uint32_t n_logs, uint32_t parts, int waves_per_simd, const PtPathConfig &cfg,
                     uint2 *spill, unsigned long long *out);
void pt_launch_closest(hipStream_t stream, const PtDevScene &scene, const float *rays6, uint32_t n, uint2 *out, const PtPathConfig &cfg);
// diagnostic (tools/step_timing.py): stamped walks, `lanes_per_wave` rays per wavefront; out[ray] = (steps, cycles waiting for records, cycles in all, price of a stamp pair)
void pt_launch_steptime(hipStream_t stream, const PtDevScene &scene, const float *rays6, uint32_t n, uint32_t lanes_per_wave, uint4 *out, uint2 *spill, uint32_t spill_depth, int flags);


#endif
