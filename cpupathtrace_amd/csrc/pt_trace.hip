// pt_trace.hip -- closest-hit / shadow traversal of the reference's BVH: persistent wavefronts over a sharded ray queue.
//
// What it computes: Scene::getIntersection (src/scene/scene.cpp:210-220) = root slab test + the ordered recursive
// impl::getChildIntersection (scene.cpp:104-150), restated as an iterative depth-first walk that visits exactly the leaves
// the recursion visits, in the same order:
//   * at an inner node both child boxes are tested (bounding_box.cpp:38-73); the nearer child is entered first, on equal entry
//     distances the RIGHT child is the nearer one (scene.cpp:120-121);
//   * a child is entered only if 0 <= entry < t_max (scene.cpp:124,137), where t_max is the smallest hit distance found so
//     far -- in the recursion t_max is threaded by value, but at every decision point it equals that global minimum, and
//     the early return of scene.cpp:129-132 is the same test (close hit < far entry  <=>  far entry >= t_max);
//   * the far child is parked on a per-lane stack TOGETHER with its entry distance and re-tested against the then-current
//     t_max when it is popped;
//   * a leaf reports Object::getIntersection unconditionally (scene.cpp:105-109); among non-negative hits the smallest wins
//     and a later-visited leaf wins ties (scene.cpp:141-146).
// Shadow rays (worker.cpp:84-86) only need "is there a visited leaf with 0 <= t < |to_light| - epsilon"; the walk stops at
// the first such leaf, which cannot change the answer.
//
// How it maps to gfx950:
//   * one ray per lane, 256-thread workgroups, persistent: a wavefront refills its idle lanes from the queue whenever at least
//     `refill_idle` lanes are idle (ballot + popcount; lane i takes the i-th ray of the wave's reservation by
//     prefix-popcount of the idle mask; one atomic reserves PT_QCHUNK rays), so lanes that drew short traversals do not
//     wait for the longest one;
//   * the queue has one shard per XCD: a workgroup drains shard blockIdx.x % 8 first (rays appended by shading workgroups
//     of the same residue, i.e. neighbouring pixels share an L2), then steals from the others;
//   * inner nodes are 64-byte records holding BOTH child boxes, so a traversal step is one 64-byte fetch (4 x dwordx4) per
//     lane; scenes whose whole tree and triangle records fit in 24 KiB are staged in LDS once per workgroup (for larger ones an
//     LDS copy of the top of the tree was measured and does not pay); everything else is a random L1/L2/HBM fetch -- the kernel
//     is bound by dependent fetches executed in lockstep, there is no matrix work in it (no MFMA);
//   * the traversal stack lives in LDS ([slot][lane] layout: conflict-free ds_read_b64/ds_write_b64) as a window over the
//     top STACK_LDS entries, with older entries of unusually deep walks in a per-lane HBM spill area.
#include "pt_device.h"
#include "pt_kernels.h"

using namespace ptd;

namespace {

// Explicit address spaces: a pointer that is "LDS or HBM depending on the index" makes hipcc fall back to flat_load (generic
// addressing, split into odd dwordx3 pieces and routed through the texture path even for LDS); with typed pointers the two
// sides stay ds_read_b128 and global_load_dwordx4.
typedef float f4v __attribute__((ext_vector_type(4)));
typedef unsigned int u2v __attribute__((ext_vector_type(2)));
typedef const f4v __attribute__((address_space(3))) *lds_f4_cptr;
typedef u2v __attribute__((address_space(3))) *lds_u2_ptr;
typedef const f4v __attribute__((address_space(1))) *glb_f4_cptr;
typedef u2v __attribute__((address_space(1))) *glb_u2_ptr;

PT_D float4 to_f4(f4v v) {
    return make_float4(v.x, v.y, v.z, v.w);
}

// slab test of the device walk: bounding_box.cpp:38-73 with std::min/std::max replaced by v_min_f32/v_max_f32.
// The two differ only for NaN operands (the reference asserts there are none, bounding_box.cpp:60-61) and in the sign of a
// zero result, and every use of the returned distance is an ordered comparison, for which -0 and +0 are the same value.
PT_D float slab_walk(V3 lo, V3 hi, V3 o, V3 inv) {
    const float t1 = (lo.x - o.x) * inv.x;
    const float t2 = (hi.x - o.x) * inv.x;
    const float t3 = (lo.y - o.y) * inv.y;
    const float t4 = (hi.y - o.y) * inv.y;
    const float t5 = (lo.z - o.z) * inv.z;
    const float t6 = (hi.z - o.z) * inv.z;
    const float t_min = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t1, t2), __builtin_fminf(t3, t4)), __builtin_fminf(t5, t6));
    const float t_max = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t1, t2), __builtin_fmaxf(t3, t4)), __builtin_fmaxf(t5, t6));
    if(t_max < 0.0f || t_min > t_max) {
        return -1.0f;
    }
    return t_min < 0.0f ? 0.0f : t_min; // origin inside: t_min < 0 <= t_max (bounding_box.cpp:68-70)
}

// LDS_MODE: 0 = every node/triangle record comes from HBM/L2 (large scenes: the LDS copy only costs occupancy),
//           1 = records with index < n_lds_* come from LDS, the rest from HBM (breadth-first top of the tree),
//           2 = the whole tree and all triangles are in LDS (small scenes),
//           3 = as 0, but the walk reads the two-level records (sc.quads): two levels of the tree per dependent fetch.
template<int STACK_LDS, int LDS_MODE, bool COUNT>
__global__ __launch_bounds__(256) void pt_trace_kernel(PtDevScene sc, PtQueue q, PtCarry carry, int parity, int max_steps, int drain_lanes, int chunk, int burst_steps,
                                                       uint2 *__restrict__ hit,
                                                       uint32_t *__restrict__ vis,
                                                       uint2 *__restrict__ spill, uint32_t spill_depth, int refill_idle, int leaf_min,
                                                       unsigned long long *__restrict__ wave_counters, uint32_t *walk_hist) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    uint2 *lds_stack = reinterpret_cast<uint2 *>(lds_raw);
    float4 *lds_pairs = reinterpret_cast<float4 *>(lds_raw + (size_t)STACK_LDS * 256 * sizeof(uint2));
    float4 *lds_tris = lds_pairs + 4 * (size_t)sc.n_lds_pairs;

    const int tid = threadIdx.x;
    const int lane = tid & 63;

    if(LDS_MODE == 1 || LDS_MODE == 2) {
        // stage the top of the tree (and the triangles of small scenes) in LDS
        for(uint32_t i = tid; i < 4 * sc.n_lds_pairs; i += 256) {
            lds_pairs[i] = sc.pairs[i];
        }
        for(uint32_t i = tid; i < 3 * sc.n_lds_tris; i += 256) {
            lds_tris[i] = sc.tris[i];
        }
        __syncthreads();
    }

    glb_u2_ptr my_spill = (glb_u2_ptr)(spill + ((size_t)blockIdx.x * 256 + tid) * spill_depth);
    lds_u2_ptr stack_l = (lds_u2_ptr)lds_stack;
    lds_f4_cptr pairs_l = (lds_f4_cptr)lds_pairs;
    lds_f4_cptr tris_l = (lds_f4_cptr)lds_tris;
    glb_f4_cptr pairs_g = (glb_f4_cptr)sc.pairs;
    glb_f4_cptr tris_g = (glb_f4_cptr)sc.tris;
    glb_f4_cptr quads_g = (glb_f4_cptr)sc.quads;

    // The traversal stack: the top STACK_LDS entries of a lane live in LDS (slot = index mod STACK_LDS), older ones in the
    // lane's HBM spill area.  Pushing onto a full window first moves the entry that is about to be overwritten to HBM;
    // popping from a deep stack brings it back.  Walks rarely go deeper than the window, so the common path is one LDS
    // access without any branching between address spaces.
    auto stack_push = [&](int &sp_, uint32_t ref, float entry_t) {
        const int slot = (sp_ % STACK_LDS) * 256 + tid;
        if(sp_ >= STACK_LDS) {
            my_spill[sp_ - STACK_LDS] = stack_l[slot];
        }
        const u2v ev = {ref, __float_as_uint(entry_t)};
        stack_l[slot] = ev;
        sp_++;
    };
    auto stack_pop = [&](int &sp_) -> u2v {
        sp_--;
        const int slot = (sp_ % STACK_LDS) * 256 + tid;
        const u2v e = stack_l[slot];
        if(sp_ >= STACK_LDS) {
            stack_l[slot] = my_spill[sp_ - STACK_LDS];
        }
        return e;
    };
    // entry i of a stack of depth sp_ (for suspending a walk)
    auto stack_peek = [&](int sp_, int i) -> u2v {
        return (i >= sp_ - STACK_LDS) ? stack_l[(i % STACK_LDS) * 256 + tid] : my_spill[i];
    };

    // wave-uniform queue cursor
    uint32_t shard = blockIdx.x % PT_SHARDS;
    uint32_t shards_tried = 0;
    bool exhausted = false;
    // rays reserved by this wavefront and not yet handed to a lane (wave-uniform); suspended walks of the previous launch
    // (pool `parity`) are reserved first, so the long walks start early
    uint32_t res_next = 0, res_end = 0;
    bool res_is_carry = false;
    const uint32_t carry_in = parity, carry_out = parity ^ 1;
    const uint32_t carry_count_raw = carry.count[carry_in * PT_QSTRIDE];
    const uint32_t carry_count = carry_count_raw < carry.cap ? carry_count_raw : carry.cap;
    bool carry_done = carry_count == 0;

    // per-lane traversal state; cur == PT_REF_NONE on an active lane means "walk finished, result not yet written"
    bool active = false;
    V3 o = v3(0, 0, 0), d = v3(0, 0, 0), inv = v3(0, 0, 0);
    float thr = 0.0f;  // shadow threshold
    uint32_t dest = 0; // destination word of the ray
    float best_t = 0.0f;
    uint32_t best_ref = PT_REF_NONE;
    float t_max = FLT_MAX;
    uint32_t cur = PT_REF_NONE;
    int sp = 0;
    uint32_t n_nodes = 0, n_leaves = 0, n_rays = 0, n_shadow = 0;
    uint32_t walk_start = 0; // n_nodes when the current walk began
    uint32_t w_steps = 0, w_leaf_phases = 0, w_refills = 0, w_suspended = 0; // wave-level diagnostics (same value in every lane)

    for(;;) {
        // ---- 1. retire finished walks ----------------------------------------------------------------------------------
        if(active && cur == PT_REF_NONE) {
            if(dest & PT_DEST_SHADOW) {
                vis[dest & ~PT_DEST_SHADOW] = 1u; // no visited leaf was closer than the light
            }
            else {
                hit[dest] = make_uint2(__float_as_uint(best_ref == PT_REF_NONE ? -1.0f : best_t), best_ref);
            }
            active = false;
            if(COUNT && walk_hist != nullptr) {
                const uint32_t steps = n_nodes - walk_start;
                atomicAdd(&walk_hist[32 - __clz((int)steps)], 1u); // bucket b: 2^(b-1) <= steps < 2^b, bucket 0: no step
            }
        }

        // ---- 2. refill idle lanes ----------------------------------------------------------------------------------------
        const unsigned long long idle_mask = __ballot(!active);
        const int n_idle = __popcll(idle_mask);
        if(n_idle >= refill_idle && (res_next < res_end || !exhausted || !carry_done)) {
            if(res_next >= res_end) {
                res_is_carry = false;
                if(!carry_done) {
                    uint32_t base = 0;
                    if(lane == 0) {
                        base = atomicAdd(&carry.head[carry_in * PT_QSTRIDE], (uint32_t)chunk);
                    }
                    base = __builtin_amdgcn_readfirstlane(base);
                    if(base < carry_count) {
                        res_next = carry_in * carry.cap + base;
                        res_end = res_next + (carry_count - base < (uint32_t)chunk ? carry_count - base : (uint32_t)chunk);
                        res_is_carry = true;
                    }
                    else {
                        carry_done = true;
                    }
                }
                // reserve the next PT_QCHUNK rays: own shard first, then the others
                while(!res_is_carry && !exhausted) {
                    const uint32_t count = q.count[shard * PT_QSTRIDE];
                    // heads only grow: a (possibly stale) head at or past the end means the shard is drained, no atomic needed
                    if(__hip_atomic_load(&q.head[shard * PT_QSTRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < count) {
                        uint32_t base = 0;
                        if(lane == 0) {
                            base = atomicAdd(&q.head[shard * PT_QSTRIDE], (uint32_t)chunk);
                        }
                        base = __builtin_amdgcn_readfirstlane(base);
                        if(base < count) {
                            res_next = shard * q.shard_capacity + base;
                            res_end = res_next + (count - base < (uint32_t)chunk ? count - base : (uint32_t)chunk);
                            break;
                        }
                    }
                    shard = (shard + 1) % PT_SHARDS;
                    if(++shards_tried >= PT_SHARDS) {
                        exhausted = true;
                    }
                }
            }
            w_refills++;
            const uint32_t avail = res_end - res_next;
            const uint32_t first = res_next;
            res_next += avail < (uint32_t)n_idle ? avail : (uint32_t)n_idle;
            if(!active) {
                const uint32_t rank = (uint32_t)__popcll(idle_mask & ((1ULL << lane) - 1ULL));
                if(rank < avail && res_is_carry) {
                    // resume a suspended walk
                    const uint32_t c = first + rank;
                    const float4 ro = carry.ray_o[c];
                    const float4 rd = carry.ray_d[c];
                    const uint4 st = carry.state[c];
                    o = v3(ro.x, ro.y, ro.z);
                    d = v3(rd.x, rd.y, rd.z);
                    thr = ro.w;
                    dest = __float_as_uint(rd.w);
                    inv = slab_inverse(d);
                    best_t = __uint_as_float(st.x);
                    best_ref = st.y;
                    t_max = __uint_as_float(st.z);
                    cur = st.w;
                    sp = (int)carry.sp[c];
                    const uint2 *saved = carry.stack + (size_t)c * carry.depth;
                    for(int i = 0; i < sp; i++) {
                        const uint2 e = saved[i];
                        const u2v ev = {e.x, e.y};
                        if(i >= sp - STACK_LDS) {
                            stack_l[(i % STACK_LDS) * 256 + tid] = ev;
                        }
                        else {
                            my_spill[i] = ev;
                        }
                    }
                    active = true;
                    walk_start = n_nodes;
                }
                else if(rank < avail) {
                    const float4 ro = q.ray_o[first + rank];
                    const float4 rd = q.ray_d[first + rank];
                    dest = __float_as_uint(rd.w);
                    if(dest != PT_DEST_NULL) {
                        o = v3(ro.x, ro.y, ro.z);
                        d = v3(rd.x, rd.y, rd.z);
                        thr = ro.w;
                        inv = slab_inverse(d);
                        best_ref = PT_REF_NONE;
                        best_t = -1.0f;
                        // a shadow ray only asks for a hit nearer than the light: nothing that is entered at or beyond that
                        // distance can hold one, so the walk starts with the threshold as its pruning distance
                        t_max = (dest & PT_DEST_SHADOW) ? thr : FLT_MAX;
                        sp = 0;
                        active = true;
                        if(COUNT) {
                            n_rays++;
                            n_shadow += (dest & PT_DEST_SHADOW) ? 1 : 0;
                        }
                        walk_start = n_nodes;
                        // Scene::getIntersection: root box first (scene.cpp:211-219)
                        cur = PT_REF_NONE;
                        if(sc.root_ref != PT_REF_NONE) {
                            const float t_root = slab_walk(ld3(sc.root_lo), ld3(sc.root_hi), o, inv);
                            if(t_root >= 0.0f) {
                                cur = sc.root_ref;
                            }
                        }
                    }
                }
            }
        }
        if(__ballot(active) == 0ULL) {
            if(exhausted && carry_done && res_next >= res_end) {
                break;
            }
            continue;
        }

        // ---- 3. inner nodes: a burst of steps for the lanes standing on an inner node -----------------------------------------
        // Lanes that reach a leaf wait (their order of visits is unchanged) until enough of them can share the leaf code.
#pragma unroll 1
        for(int burst = 0; burst < burst_steps; burst++) {
            const bool on_inner = active && !(cur & PT_REF_LEAF);
            if(__ballot(on_inner) == 0ULL) {
                break;
            }
            w_steps++;
            if(LDS_MODE == 3) {
                if(on_inner) {
                    // Two levels per fetch: the record of an even-level node carries its children's pair records, so the walk
                    // learns the entry distances of the node's two children AND of its (up to) four grandchildren from one
                    // dependent access.  The visit order is still the recursion's: close child before far child, within each the
                    // close grandchild first; a leaf child stands for itself.  Everything that is not entered right away is
                    // parked with its entry distance and re-tested against the then-current t_max when it is popped -- for a
                    // grandchild that test implies its parent's (a box inside a box is entered no earlier), which is the test
                    // the recursion makes when it comes back to the far child (scene.cpp:129-137).
                    glb_f4_cptr p = quads_g + 12 * (size_t)cur;
                    const float4 a0 = to_f4(p[0]), a1 = to_f4(p[1]), a2 = to_f4(p[2]), a3 = to_f4(p[3]);
                    const float4 l0 = to_f4(p[4]), l1 = to_f4(p[5]), l2 = to_f4(p[6]), l3 = to_f4(p[7]);
                    const float4 r0 = to_f4(p[8]), r1 = to_f4(p[9]), r2 = to_f4(p[10]), r3 = to_f4(p[11]);
                    const float t_l = slab_walk(v3(a0.x, a0.y, a0.z), v3(a0.w, a1.x, a1.y), o, inv);
                    const float t_r = slab_walk(v3(a1.z, a1.w, a2.x), v3(a2.y, a2.z, a2.w), o, inv);
                    const uint32_t ref_l = __float_as_uint(a3.x), ref_r = __float_as_uint(a3.y);
                    const float t_ll = slab_walk(v3(l0.x, l0.y, l0.z), v3(l0.w, l1.x, l1.y), o, inv);
                    const float t_lr = slab_walk(v3(l1.z, l1.w, l2.x), v3(l2.y, l2.z, l2.w), o, inv);
                    const float t_rl = slab_walk(v3(r0.x, r0.y, r0.z), v3(r0.w, r1.x, r1.y), o, inv);
                    const float t_rr = slab_walk(v3(r1.z, r1.w, r2.x), v3(r2.y, r2.z, r2.w), o, inv);
                    // per side: first and second candidate in visit order (a leaf child is its own single candidate)
                    const bool l_leaf = (ref_l & PT_REF_LEAF) != 0, r_leaf = (ref_r & PT_REF_LEAF) != 0;
                    const bool ll_close = t_ll < t_lr, rl_close = t_rl < t_rr;
                    const uint32_t l_first_ref = l_leaf ? ref_l : __float_as_uint(ll_close ? l3.x : l3.y);
                    const float l_first_t = l_leaf ? t_l : (ll_close ? t_ll : t_lr);
                    const uint32_t l_second_ref = __float_as_uint(ll_close ? l3.y : l3.x);
                    const float l_second_t = l_leaf ? -1.0f : (ll_close ? t_lr : t_ll);
                    const uint32_t r_first_ref = r_leaf ? ref_r : __float_as_uint(rl_close ? r3.x : r3.y);
                    const float r_first_t = r_leaf ? t_r : (rl_close ? t_rl : t_rr);
                    const uint32_t r_second_ref = __float_as_uint(rl_close ? r3.y : r3.x);
                    const float r_second_t = r_leaf ? -1.0f : (rl_close ? t_rr : t_rl);
                    const bool ok_l = t_l >= 0.0f && t_l < t_max, ok_r = t_r >= 0.0f && t_r < t_max;
                    const bool left_close = t_l < t_r; // equal entry distances: right is "close" (scene.cpp:120-121)
                    // candidates c0..c3 in visit order
                    const bool ok_a = left_close ? ok_l : ok_r, ok_b = left_close ? ok_r : ok_l;
                    const uint32_t c0_ref = left_close ? l_first_ref : r_first_ref, c1_ref = left_close ? l_second_ref : r_second_ref;
                    const uint32_t c2_ref = left_close ? r_first_ref : l_first_ref, c3_ref = left_close ? r_second_ref : l_second_ref;
                    const float c0_t = left_close ? l_first_t : r_first_t, c1_t = left_close ? l_second_t : r_second_t;
                    const float c2_t = left_close ? r_first_t : l_first_t, c3_t = left_close ? r_second_t : l_second_t;
                    const bool v0 = ok_a && c0_t >= 0.0f && c0_t < t_max, v1 = ok_a && c1_t >= 0.0f && c1_t < t_max;
                    const bool v2 = ok_b && c2_t >= 0.0f && c2_t < t_max, v3_ = ok_b && c3_t >= 0.0f && c3_t < t_max;
                    if(COUNT) {
                        // binary-equivalent node visits: this node and each inner child that was entered
                        n_nodes += 1u + ((ok_l && !l_leaf) ? 1u : 0u) + ((ok_r && !r_leaf) ? 1u : 0u);
                    }
                    if(v3_ && (v0 || v1 || v2)) {
                        stack_push(sp, c3_ref, c3_t);
                    }
                    if(v2 && (v0 || v1)) {
                        stack_push(sp, c2_ref, c2_t);
                    }
                    if(v1 && v0) {
                        stack_push(sp, c1_ref, c1_t);
                    }
                    cur = v0 ? c0_ref : (v1 ? c1_ref : (v2 ? c2_ref : (v3_ ? c3_ref : PT_REF_NONE)));
                    if(!(v0 || v1 || v2 || v3_)) {
                        while(sp > 0) {
                            const u2v e = stack_pop(sp);
                            if(__uint_as_float(e.y) < t_max) {
                                cur = e.x;
                                break;
                            }
                        }
                    }
                }
            }
            else if(on_inner) {
                float4 q0, q1, q2, q3;
                if(LDS_MODE == 2 || (LDS_MODE == 1 && cur < sc.n_lds_pairs)) {
                    lds_f4_cptr p = pairs_l + 4 * cur;
                    q0 = to_f4(p[0]);
                    q1 = to_f4(p[1]);
                    q2 = to_f4(p[2]);
                    q3 = to_f4(p[3]);
                }
                else {
                    glb_f4_cptr p = pairs_g + 4 * (size_t)cur;
                    q0 = to_f4(p[0]);
                    q1 = to_f4(p[1]);
                    q2 = to_f4(p[2]);
                    q3 = to_f4(p[3]);
                }
                if(COUNT) {
                    n_nodes++;
                }
                const float left_t = slab_walk(v3(q0.x, q0.y, q0.z), v3(q0.w, q1.x, q1.y), o, inv);
                const float right_t = slab_walk(v3(q1.z, q1.w, q2.x), v3(q2.y, q2.z, q2.w), o, inv);
                const uint32_t left_ref = __float_as_uint(q3.x);
                const uint32_t right_ref = __float_as_uint(q3.y);
                const bool left_close = left_t < right_t; // equal entry distances: right is "close" (scene.cpp:120-121)
                const float close_t = left_close ? left_t : right_t;
                const float far_t = left_close ? right_t : left_t;
                const uint32_t close_ref = left_close ? left_ref : right_ref;
                const uint32_t far_ref = left_close ? right_ref : left_ref;
                const bool go_close = close_t >= 0.0f && close_t < t_max;
                const bool go_far = far_t >= 0.0f && far_t < t_max;
                if(go_close && go_far) {
                    stack_push(sp, far_ref, far_t);
                }
                cur = go_close ? close_ref : (go_far ? far_ref : PT_REF_NONE);
                if(!go_close && !go_far) {
                    // pop: the first parked node whose entry distance is still below t_max
                    while(sp > 0) {
                        const u2v e = stack_pop(sp);
                        if(__uint_as_float(e.y) < t_max) {
                            cur = e.x;
                            break;
                        }
                    }
                }
            }
        }

        // ---- 3b. suspend walks that used up this launch's step budget -------------------------------------------------------------
        {
            // ... and, once the queue has nothing left for this wavefront, the last few walks of the wave (after at least 32 steps
            // each, so every walk makes progress in every launch): the launch then ends with the bulk of the rays instead of
            // idling the chip behind a handful of long walks; those resume first in the next launch.
            const bool draining = exhausted && carry_done && res_next >= res_end;
            const uint32_t steps_here = n_nodes - walk_start;
            const bool few_left = draining && __popcll(__ballot(active)) <= drain_lanes;
            const bool over = active && cur != PT_REF_NONE && (steps_here >= (uint32_t)max_steps || (few_left && steps_here >= 32u));
            const unsigned long long over_mask = __ballot(over);
            if(over_mask != 0ULL) {
                w_suspended += (uint32_t)__popcll(over_mask);
                uint32_t base = 0;
                if(lane == 0) {
                    base = atomicAdd(&carry.count[carry_out * PT_QSTRIDE], (uint32_t)__popcll(over_mask));
                }
                base = __builtin_amdgcn_readfirstlane(base);
                const uint32_t slot = base + (uint32_t)__popcll(over_mask & ((1ULL << lane) - 1ULL));
                if(over && slot < carry.cap) { // a full pool just means the walk keeps running in this launch
                    const uint32_t c = carry_out * carry.cap + slot;
                    carry.ray_o[c] = make_float4(o.x, o.y, o.z, thr);
                    carry.ray_d[c] = make_float4(d.x, d.y, d.z, __uint_as_float(dest));
                    carry.state[c] = make_uint4(__float_as_uint(best_t), best_ref, __float_as_uint(t_max), cur);
                    carry.sp[c] = (uint32_t)sp;
                    uint2 *saved = carry.stack + (size_t)c * carry.depth;
                    for(int i = 0; i < sp; i++) {
                        const u2v e = stack_peek(sp, i);
                        saved[i] = make_uint2(e.x, e.y);
                    }
                    active = false;
                }
            }
        }

        // ---- 4. leaves: Object::getIntersection for the lanes standing on a leaf ------------------------------------------------
        const bool on_leaf = active && (cur & PT_REF_LEAF) && cur != PT_REF_NONE;
        const unsigned long long leaf_mask = __ballot(on_leaf);
        if(leaf_mask != 0ULL && (__popcll(leaf_mask) >= leaf_min || __ballot(active && !(cur & PT_REF_LEAF)) == 0ULL)) {
            w_leaf_phases++;
            if(on_leaf) {
                const uint32_t idx = cur & PT_REF_INDEX;
                float t;
                if(cur & PT_REF_SPHERE) {
                    const float4 s = sc.spheres[idx];
                    t = sphere_intersect(v3(s.x, s.y, s.z), s.w, o, d);
                }
                else {
                    float4 t0, t1, t2;
                    if(LDS_MODE == 2 || (LDS_MODE == 1 && idx < sc.n_lds_tris)) {
                        lds_f4_cptr p = tris_l + 3 * idx;
                        t0 = to_f4(p[0]);
                        t1 = to_f4(p[1]);
                        t2 = to_f4(p[2]);
                    }
                    else {
                        glb_f4_cptr p = tris_g + 3 * (size_t)idx;
                        t0 = to_f4(p[0]);
                        t1 = to_f4(p[1]);
                        t2 = to_f4(p[2]);
                    }
                    const TriRec tr = tri_unpack(t0, t1, t2);
                    t = tri_intersect(tr.a, tr.ab, tr.ac, (tr.obj_cull >> 31) != 0, o, d);
                }
                if(COUNT) {
                    n_leaves++;
                }
                bool occluded = false;
                if(t >= 0.0f) {
                    if((dest & PT_DEST_SHADOW) && t < thr) {
                        occluded = true; // worker.cpp:86: a hit closer than the light
                    }
                    else {
                        if(best_ref == PT_REF_NONE || !(best_t < t)) {
                            best_t = t;
                            best_ref = cur;
                        }
                        t_max = fmin_std(t_max, t);
                    }
                }
                if(occluded) {
                    vis[dest & ~PT_DEST_SHADOW] = 0u;
                    active = false;
                }
                else {
                    cur = PT_REF_NONE;
                    while(sp > 0) {
                        const u2v e = stack_pop(sp);
                        if(__uint_as_float(e.y) < t_max) {
                            cur = e.x;
                            break;
                        }
                    }
                }
            }
        }
    }

    if(COUNT) {
        // Work counters: every wave owns one 64-byte slot and adds its totals with plain stores.  (Atomics on one shared line
        // from every wave of the grid serialise at the memory side -- about 40 ns each -- and were a visible part of the launch.)
        for(int off = 32; off > 0; off >>= 1) {
            n_nodes += __shfl_down(n_nodes, off);
            n_leaves += __shfl_down(n_leaves, off);
            n_rays += __shfl_down(n_rays, off);
            n_shadow += __shfl_down(n_shadow, off);
        }
        if(lane == 0) {
            unsigned long long *slot = wave_counters + 8 * ((size_t)blockIdx.x * 4 + (tid >> 6));
            slot[0] += n_nodes;
            slot[1] += n_leaves;
            slot[2] += n_rays;
            slot[3] += n_shadow;
            slot[4] += w_steps;
            slot[5] += w_leaf_phases;
            slot[6] += w_refills;
            slot[7] += w_suspended;
        }
    }
}

// pt_intersect_batch: one closest-hit ray per input ray, destination = its index
__global__ void pt_batch_rays_kernel(const float *__restrict__ rays6, uint32_t n, PtQueue q) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n) {
        return;
    }
    // shard s holds the contiguous chunk [s * per, (s + 1) * per)
    const uint32_t per = (n + PT_SHARDS - 1) / PT_SHARDS;
    const uint32_t s = i / per;
    const uint32_t k = i - s * per;
    const float *r = rays6 + 6 * (size_t)i;
    q.ray_o[s * q.shard_capacity + k] = make_float4(r[0], r[1], r[2], 0.0f);
    q.ray_d[s * q.shard_capacity + k] = make_float4(r[3], r[4], r[5], __uint_as_float(i));
    if(k == 0) {
        const uint32_t end = (s + 1) * per < n ? (s + 1) * per : n;
        q.count[s * PT_QSTRIDE] = end - s * per;
        q.head[s * PT_QSTRIDE] = 0;
    }
}

template<int STACK_LDS, int LDS_MODE>
void launch_trace(hipStream_t stream, const PtDevScene &scene, PtQueue queue, PtCarry carry, PtPaths paths, const PtTraceConfig &cfg,
                  PtDevCounters *counters) {
    hipLaunchKernelGGL((pt_trace_kernel<STACK_LDS, LDS_MODE, true>), dim3(cfg.grid), dim3(256), cfg.lds_bytes, stream, scene, queue, carry, cfg.parity, cfg.max_steps, cfg.drain_lanes, cfg.chunk, cfg.burst_steps, paths.hit,
                       paths.vis,
                       cfg.spill, cfg.spill_depth, cfg.refill_idle, cfg.leaf_min, cfg.wave_counters, cfg.walk_hist);
}

template<int STACK_LDS>
void launch_trace_mode(hipStream_t stream, const PtDevScene &scene, PtQueue queue, PtCarry carry, PtPaths paths, const PtTraceConfig &cfg,
                       PtDevCounters *counters) {
    switch(cfg.lds_mode) {
        case 0:
            launch_trace<STACK_LDS, 0>(stream, scene, queue, carry, paths, cfg, counters);
            break;
        case 2:
            launch_trace<STACK_LDS, 2>(stream, scene, queue, carry, paths, cfg, counters);
            break;
        case 3:
            launch_trace<STACK_LDS, 3>(stream, scene, queue, carry, paths, cfg, counters);
            break;
        default:
            launch_trace<STACK_LDS, 1>(stream, scene, queue, carry, paths, cfg, counters);
            break;
    }
}

template<int STACK_LDS>
int occupancy_mode(int lds_mode, size_t lds_bytes) {
    int blocks = 0;
    hipError_t err;
    switch(lds_mode) {
        case 0:
            err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, pt_trace_kernel<STACK_LDS, 0, true>, 256, lds_bytes);
            break;
        case 2:
            err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, pt_trace_kernel<STACK_LDS, 2, true>, 256, lds_bytes);
            break;
        case 3:
            err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, pt_trace_kernel<STACK_LDS, 3, true>, 256, lds_bytes);
            break;
        default:
            err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, pt_trace_kernel<STACK_LDS, 1, true>, 256, lds_bytes);
            break;
    }
    return (err != hipSuccess || blocks < 1) ? 1 : blocks;
}

} // namespace

void pt_launch_trace(hipStream_t stream, const PtDevScene &scene, PtQueue queue, PtCarry carry, PtPaths paths, const PtTraceConfig &cfg,
                     PtDevCounters *counters) {
    switch(cfg.stack_lds) {
        case 8:
            launch_trace_mode<8>(stream, scene, queue, carry, paths, cfg, counters);
            break;
        case 24:
            launch_trace_mode<24>(stream, scene, queue, carry, paths, cfg, counters);
            break;
        default:
            launch_trace_mode<16>(stream, scene, queue, carry, paths, cfg, counters);
            break;
    }
}

int pt_trace_blocks_per_cu(int stack_lds, int lds_mode, size_t lds_bytes) {
    switch(stack_lds) {
        case 8:
            return occupancy_mode<8>(lds_mode, lds_bytes);
        case 24:
            return occupancy_mode<24>(lds_mode, lds_bytes);
        default:
            return occupancy_mode<16>(lds_mode, lds_bytes);
    }
}

void pt_launch_batch_rays(hipStream_t stream, const float *rays6, uint32_t n, PtQueue queue) {
    if(n == 0) {
        return;
    }
    hipLaunchKernelGGL(pt_batch_rays_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, rays6, n, queue);
}
