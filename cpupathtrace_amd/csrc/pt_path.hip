// pt_path.hip -- the hot path as ONE persistent kernel: per-pixel propagation loop (impl::getSample, src/worker.cpp:26-146), per-pixel
// estimator (processItem, src/worker.cpp:149-326) and closest-hit / shadow traversal of the reference's BVH
// (Scene::getIntersection, src/scene/scene.cpp:104-150,210-220).
//
// Every wavefront of the grid is an independent renderer.  It owns `rows` x 64 stream SLOTS (a slot = one processItem stream in
// progress: one engine, one path in flight), a private ring of rays in HBM and a few words per slot in LDS, and runs
//
//     loop:  retire finished walks (results -> LDS)
//            idle lanes?  queue has rays -> hand them out
//                         queue empty    -> SHADE every slot whose rays have all come back (one pass per row of 64 slots: coalesced
//                                           path state, exactly the reference's order of draws per stream), which appends new rays
//            TRACE: a burst of traversal steps for the 64 walks in the lanes
//
// until its slots have no stream left and the global stream counter is exhausted.  Nothing is ever synchronised between wavefronts
// (no grid barrier, no shared queue, no launch boundary): a ray that needs thousands of traversal steps keeps ONE lane busy and makes
// ONE stream miss a few shading passes, while the other 63 lanes and 255 slots go on -- in the two-kernel wavefront design of round 1
// every launch lasted as long as its slowest walk and the average wavefront was alive for 48 % of it (profiles/r01_pmc_*).
// Lanes are not tied to slots: a lane takes the next ray of the wave's queue whatever slot it belongs to, so the traversal runs on
// compacted, full wavefronts although the streams progress at different speeds.
//
// Traversal: the ordered recursion of impl::getChildIntersection restated as an iterative walk that visits exactly the leaves the
// recursion visits, in the same order:
//   * at an inner node both child boxes are tested (bounding_box.cpp:38-73); the nearer child is entered first, on equal entry
//     distances the RIGHT child is the nearer one (scene.cpp:120-121);
//   * a child is entered only if 0 <= entry < t_max (scene.cpp:124,137), where t_max is the smallest hit distance found so far -- in
//     the recursion t_max is threaded by value, but at every decision point it equals that global minimum, and the early return of
//     scene.cpp:129-132 is the same test (close hit < far entry  <=>  far entry >= t_max);
//   * the far child is parked on a per-lane stack TOGETHER with its entry distance and re-tested against the then-current t_max when
//     it is popped;
//   * a leaf reports Object::getIntersection unconditionally (scene.cpp:105-109); among non-negative hits the smallest wins and a
//     later-visited leaf wins ties (scene.cpp:141-146).
// Shadow rays (worker.cpp:84-86) only need "is there a visited leaf with 0 <= t < |to_light| - epsilon"; the walk stops at the first
// such leaf, which cannot change the answer.
// One traversal step serves inner nodes and leaves alike: every lane fetches the 64-byte record it stands on (a node's two child
// boxes, or a triangle) with the same four loads, so leaf tests cost no memory round trip of their own.
//
// gfx950 specifics: 64-wide ballots/popcounts for compaction, typed LDS/global address spaces (a pointer that may be either makes
// hipcc emit flat_load), per-lane traversal stack in LDS as [entry][lane] (conflict-free ds_read/write_b64) with an HBM spill area
// for unusually deep walks, LDS atomics for the per-slot completion words, no MFMA (there is no dense contraction on this path).
#include <type_traits>

#include "pt_device.h"
#include "pt_kernels.h"
#include "pt_shading.h"

using namespace ptd;

namespace {

typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef unsigned int u2v __attribute__((ext_vector_type(2)));
typedef const f4v __attribute__((address_space(3))) *lds_f4_cptr;
typedef const f4v __attribute__((address_space(1))) *glb_f4_cptr;
typedef u2v __attribute__((address_space(3))) *lds_u2_ptr;
typedef u2v __attribute__((address_space(1))) *glb_u2_ptr;
typedef unsigned int __attribute__((address_space(3))) *lds_u32_ptr;
typedef const PtPathArgs __attribute__((address_space(4))) *args_c4;

template<bool IN_LDS>
struct RecPtr {
    typedef glb_f4_cptr type;
};
template<>
struct RecPtr<true> {
    typedef lds_f4_cptr type;
};

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

// The root of the tree: its box is tested before anything else (Scene::getIntersection, scene.cpp:211-219)
struct RootBox {
    float lo[3], hi[3];
    uint32_t ref;
};

// One walk (one ray) in a lane.
//
// `cur` is where the walk stands: the reference of an inner node (bits 31, 30 = 00) or of a leaf (bit 31 set, pt_types.h), PT_REF_NONE when
// the walk is over, or PT_REF_POPPING when it has to return to a parked node but the entry on top of its stack has already been
// discarded (see Tracer::node_step).  The stack of parked nodes has a SENTINEL as entry 0 -- (PT_REF_NONE, -1) -- so "the stack is
// empty" needs no test anywhere: popping the sentinel ends the walk, and its distance passes every pruning test.
#define PT_REF_POPPING 0xfffffffeu
struct Walk {
    V3 o, d, inv;
    f2v o_xy, o_zx, o_yz, i_xy, i_zx, i_yz; // origin and inverse direction again, as the register pairs of the packed slab arithmetic
    float thr;       // shadow threshold |to_light| - epsilon (worker.cpp:86)
    uint32_t dest;   // destination word of the ray
    float best_t;
    uint32_t best_ref;
    float t_max;     // pruning distance: the smallest hit distance so far (scene.cpp:124,137)
    float t_lim;     // the largest float below t_max: x < t_max  <=>  x <= t_lim, which lets min() fold the pruning test into the box test
    uint32_t cur;
    uint32_t sp;     // entries on the stack, the sentinel included
    bool occluded;   // shadow ray: a leaf closer than the light was found

    PT_D void pack() {
        o_xy = (f2v){o.x, o.y};
        o_zx = (f2v){o.z, o.x};
        o_yz = (f2v){o.y, o.z};
        i_xy = (f2v){inv.x, inv.y};
        i_zx = (f2v){inv.z, inv.x};
        i_yz = (f2v){inv.y, inv.z};
    }
    // t_max is never negative (hit distances are >= 0; it may be -0): below zero there is nothing, and every entry distance is >= 0
    PT_D void set_t_max(float t) {
        t_max = t;
        t_lim = t > 0.0f ? __uint_as_float(__float_as_uint(t) - 1u) : -1.0f;
    }
};

// The traversal machinery of one lane: record arrays (LDS or HBM), the stack window in LDS and its HBM spill area.
//
// What a step costs on this chip (tools/issue_probe.hip, profiles/r03_issue_probe.txt): a wavefront that is alone on its SIMD issues one
// instruction per 4.5 cycles whatever the instruction; a taken branch costs 22 cycles, a not-taken one 13, a wave-uniform branch on a
// ballot (v_cmp into an SGPR pair, s_cmp, s_cbranch) 35-52, a lane mask that goes through the scalar unit on its way to a v_cndmask
// 16 more than one that stays in vcc, an LDS round trip 55, an L1 hit 112.  A stream's samples are sequential, so at the end of every
// launch -- and for the whole of a strong-scaling share -- the frame time is the length of a few such lonely chains.  Hence:
//   * node_step is STRAIGHT-LINE code for all 64 lanes, no branch and no exec mask but the one around the record loads.  Conditions
//     never meet in scalar registers: a child that is not entered gets the entry distance +inf, and min / max / compare-with-inf on
//     the two distances yield near child, far child, "both" and "none" (each v_cmp feeds the v_cndmask or the add-with-carry behind
//     it through vcc).  The far child is written ABOVE the top of the stack by every lane (it only becomes an entry where the stack
//     pointer moves) and the top entry is read by every lane ahead of the arithmetic, so the first pop of the recursion's return
//     (scene.cpp:137) costs no trip to LDS;
//   * everything rare -- leaves, a popped entry that fails the distance test, a stack deeper than its LDS window -- is left to
//     slow_step, which the traversal loop enters through ONE wave-uniform branch per step.
// Diagnostic build (-DPT_STEP_STAMPS, tools/step_timing.py): s_memtime stamps inside the step; segment k collects the cycles from the
// previous stamp to stamp k, each inflated by the round trip of the previous stamp itself (segment 7 = two stamps back to back: that price)
#ifdef PT_STEP_STAMPS
#define PT_STAMP(k)                                                      \
    do {                                                                 \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               \
        stamp_acc[k] += now_ - stamp_last;                               \
        stamp_last = now_;                                               \
    } while(0)
#else
#define PT_STAMP(k) do { } while(0)
#endif

template<int STACK_LDS, bool IN_LDS>
struct Tracer {
    static_assert((STACK_LDS & (STACK_LDS - 1)) == 0, "the stack window is indexed with a mask");
#ifdef PT_STEP_STAMPS
    mutable unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last = 0;
#endif
    typedef typename RecPtr<IN_LDS>::type rec_ptr;
    rec_ptr recs;         // the 64-byte records the tree's references index: leaves, then pairs (pt_types.h)
    lds_u2_ptr stack_l;   // this thread's column: entry e at stack_l[(e mod STACK_LDS) * 256]
    glb_u2_ptr my_spill;  // entries that have left the window: entry e at my_spill[e]

    // The 64-byte record a walk stands on, requested as soon as the walk knows where it goes next: a node's pair of child boxes, a
    // triangle, or a sphere's (origin, radius).  A reference's low 30 bits ARE the record's index: one mask, one shift-add.
    struct Rec {
        f4v r0, r1, r2, r3;
    };
    PT_D void fetch(uint32_t cur, Rec &R) const {
        rec_ptr p = recs + 4 * (size_t)(cur & PT_REF_INDEX);
        R.r0 = p[0];
        R.r1 = p[1];
        R.r2 = p[2];
        R.r3 = p[3];
    }
    // Registers that a load fills and nobody reads (the last two words of a record) become the compiler's scratch registers, and their
    // first use then has to wait for the load: a full memory latency at the end of every step.  Every consumer of a record calls this.
    static PT_D void whole(const Rec &R) {
        asm volatile("" ::"v"(R.r0), "v"(R.r1), "v"(R.r2), "v"(R.r3));
    }

    // Where a walk stands, classified with one comparison each: a reference below 2^30 = an inner node (node_step moves it); at or above,
    // except PT_REF_NONE = it waits for slow_step (a leaf, or a walk that has to go on popping).
    static PT_D unsigned long long node_lanes(uint32_t cur) { return __builtin_amdgcn_uicmp(cur, 0x40000000u, 36); }       // cur < 2^30
    static PT_D unsigned long long slow_lanes(uint32_t cur) { return __builtin_amdgcn_uicmp(cur + 1u, 0x40000000u, 34); }  // 2^30 <= cur < NONE

    // Start a walk: Scene::getIntersection tests the root box first (scene.cpp:211-219).
    PT_D void start(Walk &w, Rec &R, const RootBox &root, float4 ro, float4 rd) const {
        w.o = v3(ro.x, ro.y, ro.z);
        w.d = v3(rd.x, rd.y, rd.z);
        w.thr = ro.w;
        w.dest = __float_as_uint(rd.w);
        w.inv = slab_inverse(w.d);
        w.pack();
        w.best_ref = PT_REF_NONE;
        w.best_t = -1.0f;
        // A shadow ray is a closest-hit query like any other in the reference (worker.cpp:83-86) and must be pruned like one: starting it
        // with the light's distance as pruning distance is NOT the same thing in floating point.  The sampled point lies on an emitter,
        // the ray starts epsilon in front of the vertex and the threshold is |to_light| - epsilon: the emitter's own hit distance and
        // the threshold are the same number up to rounding, and the reference finds "occluded" whenever the hit comes out an ulp
        // below.  A box around a flat, axis-aligned emitter is entered at that very distance (again up to rounding, of the slab test
        // this time), so pruning at the threshold skipped the emitter in cases where the reference tested it and found t < threshold.
        // The walk still ends at the first hit below the threshold (the closest hit can only be nearer).
        w.set_t_max(FLT_MAX);
        const u2v sentinel = {PT_REF_NONE, __float_as_uint(-1.0f)};
        stack_l[0] = sentinel;
        w.sp = 1;
        w.occluded = false;
        w.cur = PT_REF_NONE;
        if(root.ref != PT_REF_NONE) {
            const float t_root = slab_walk(ld3(root.lo), ld3(root.hi), w.o, w.inv);
            if(t_root >= 0.0f) {
                w.cur = root.ref;
                fetch(w.cur, R);
            }
        }
    }

    // One step of every walk that stands on an inner node (`node_mask`); the record of where a walk stands next is requested.
    // AABB::getIntersection of both children (bounding_box.cpp:38-73): a box is hit iff t_max >= 0 and t_min <= t_max -- the
    // same as max(t_min, 0) <= t_max -- and its entry distance is max(t_min, 0) (0 = origin inside, :68-70).
    // impl::getChildIntersection (scene.cpp:113-146): a child is entered iff it is hit and its entry distance is below the pruning
    // distance (entry < t_max <=> entry <= t_lim, so both tests are ONE comparison with min(box exit, t_lim)); with both entered the
    // nearer one comes first -- on equal distances the RIGHT one (scene.cpp:120-121): "left first" is a strict less-than -- and the other
    // is parked with its entry distance; with none entered the walk returns to the node on top of its stack if that one's entry
    // distance is still below the pruning distance (scene.cpp:137), and goes on popping in slow_step otherwise.
    // The stack: the top STACK_LDS entries of a lane live in LDS (slot = index mod STACK_LDS), older ones in the lane's HBM spill area.
    // `deep_mask`: the lanes whose stack has left the window -- for them a push first moves the entry it overwrites to the spill area and
    // a pop brings the entry that left the window last back into the slot that has become free; if there is no such lane (one scalar
    // branch) the far child is simply written ABOVE the top of the stack by every lane: it only becomes an entry where the pointer moves.
    PT_D void node_step(Walk &w, Rec &R, unsigned long long node_mask, unsigned long long deep_mask) const {
        if(!__builtin_amdgcn_inverse_ballot_w64(node_mask)) {
            return; // (the one exec mask of the step; the caller knows the mask is not empty)
        }
#ifdef PT_STEP_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        PT_STAMP(2); // waiting for the record
        whole(R);
        const f4v q0 = R.r0, q1 = R.r1, q2 = R.r2;
        const uint32_t sp = w.sp;
        const u2v top = stack_l[((sp - 1u) & (uint32_t)(STACK_LDS - 1)) * 256u]; // read ahead of the arithmetic that decides whether it is needed
        // twelve differences and products, two per instruction (v_pk_add_f32 with a negated operand is the IEEE subtraction,
        // v_pk_mul_f32 the IEEE product: nothing is fused, every result is the reference's)
        const f2v a0 = ((f2v){q0.x, q0.y} - w.o_xy) * w.i_xy; // L.lo.x, L.lo.y
        const f2v a1 = ((f2v){q0.z, q0.w} - w.o_zx) * w.i_zx; // L.lo.z, L.hi.x
        const f2v a2 = ((f2v){q1.x, q1.y} - w.o_yz) * w.i_yz; // L.hi.y, L.hi.z
        const f2v a3 = ((f2v){q1.z, q1.w} - w.o_xy) * w.i_xy; // R.lo.x, R.lo.y
        const f2v a4 = ((f2v){q2.x, q2.y} - w.o_zx) * w.i_zx; // R.lo.z, R.hi.x
        const f2v a5 = ((f2v){q2.z, q2.w} - w.o_yz) * w.i_yz; // R.hi.y, R.hi.z
        const float l1 = a0.x, l2 = a1.y, l3 = a0.y, l4 = a2.x, l5 = a1.x, l6 = a2.y;
        const float r1 = a3.x, r2 = a4.y, r3 = a3.y, r4 = a5.x, r5 = a4.x, r6 = a5.y;
        const float l_min = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(l1, l2), __builtin_fminf(l3, l4)), __builtin_fminf(l5, l6));
        const float l_max = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(l1, l2), __builtin_fmaxf(l3, l4)), __builtin_fmaxf(l5, l6));
        const float r_min = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(r1, r2), __builtin_fminf(r3, r4)), __builtin_fminf(r5, r6));
        const float r_max = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(r1, r2), __builtin_fmaxf(r3, r4)), __builtin_fmaxf(r5, r6));
        const float left_t = __builtin_fmaxf(l_min, 0.0f), right_t = __builtin_fmaxf(r_min, 0.0f);
        const float inf = __builtin_inff();
        const float t_lim = w.t_lim;
        const float tl = left_t <= __builtin_fminf(l_max, t_lim) ? left_t : inf;   // entry distance of a child that is entered, else +inf
        const float tr = right_t <= __builtin_fminf(r_max, t_lim) ? right_t : inf;
        const bool left_first = tl < tr;
        const uint32_t left_ref = __float_as_uint(R.r3.x), right_ref = __float_as_uint(R.r3.y);
        const uint32_t near_ref = left_first ? left_ref : right_ref;
        const float near_t = __builtin_fminf(tl, tr), far_t = __builtin_fmaxf(tl, tr);
        const u2v far = {left_first ? right_ref : left_ref, __float_as_uint(far_t)};
        const bool both = far_t < inf, entered = near_t < inf;
        const uint32_t slot = (sp & (uint32_t)(STACK_LDS - 1)) * 256u;
        if(deep_mask == 0ULL) {
            stack_l[slot] = far;
        }
        else if(both) {
            if(sp >= (uint32_t)STACK_LDS) {
                my_spill[sp - STACK_LDS] = stack_l[slot];
            }
            stack_l[slot] = far;
        }
        const uint32_t popped = __uint_as_float(top.y) <= t_lim ? top.x : PT_REF_POPPING;
        const uint32_t next = entered ? near_ref : popped;
        w.sp = entered ? sp + (both ? 1u : 0u) : sp - 1u;
        w.cur = next;
        asm volatile("" ::"v"(w.sp), "v"(w.cur)); // (the walk's new state is complete before the record is requested: nothing is left to do behind the loads)
        PT_STAMP(3); // slab tests, decision, stack
        // the lanes that moved onto a record (not the ones whose walk ended or that go on popping: the two codes at the top)
        if(next < PT_REF_POPPING) {
            fetch(next, R);
        }
        if(deep_mask != 0ULL) {
            if(!entered & (sp - 1u >= (uint32_t)STACK_LDS)) {
                stack_l[((sp - 1u) & (uint32_t)(STACK_LDS - 1)) * 256u] = my_spill[sp - 1u - STACK_LDS]; // the window moves down
            }
        }
        PT_STAMP(4); // address and request of the next record
    }

    // Everything that is not the common step: the leaves in `leaf_mask` (Object::getIntersection) and the walks that must (go on) pop(ping).
    // n_leaves counts visits for the whole wavefront (the same value in every lane).
    PT_D void slow_step(Walk &w, Rec &R, unsigned long long leaf_mask, uint32_t &n_leaves) const {
        uint32_t cur = w.cur;
        bool need_pop = cur == PT_REF_POPPING;
        bool moved = false;
        if(leaf_mask != 0ULL) {
            n_leaves += (uint32_t)__popcll(leaf_mask);
            if(__builtin_amdgcn_inverse_ballot_w64(leaf_mask)) {
                // a leaf reports Object::getIntersection unconditionally (scene.cpp:105-109); among the non-negative hits the smallest wins and a
                // later-visited leaf wins ties (scene.cpp:141-146); a shadow walk ends at its first hit below the threshold (worker.cpp:86)
                const float4 q0 = to_f4(R.r0), q1 = to_f4(R.r1), q2 = to_f4(R.r2);
                float t_leaf;
                if(cur & PT_REF_SPHERE) {
                    t_leaf = sphere_intersect(v3(q0.x, q0.y, q0.z), q0.w, w.o, w.d);
                }
                else {
                    const TriRec tr = tri_unpack(q0, q1, q2);
                    t_leaf = tri_intersect(tr.a, tr.ab, tr.ac, (tr.obj_cull >> 31) != 0, w.o, w.d);
                }
                need_pop = true;
                if(t_leaf >= 0.0f) {
                    if((w.dest & PT_DEST_SHADOW) && t_leaf < w.thr) {
                        w.occluded = true;
                        need_pop = false;
                    }
                    else {
                        if(w.best_ref == PT_REF_NONE || !(w.best_t < t_leaf)) {
                            w.best_t = t_leaf;
                            w.best_ref = cur;
                        }
                        w.set_t_max(fmin_std(w.t_max, t_leaf));
                    }
                }
                cur = PT_REF_NONE;
            }
        }
        // pop: the first parked node whose entry distance is still below t_max (scene.cpp:137: re-tested against the then-current distance);
        // the sentinel at the bottom passes the test and ends the walk
        if(__ballot(need_pop) != 0ULL) {
            uint32_t sp = w.sp;
            const float t_max = w.t_max;
            while(__ballot(need_pop) != 0ULL) {
                if(need_pop) {
                    sp--;
                    const uint32_t slot = (sp & (uint32_t)(STACK_LDS - 1)) * 256u;
                    const u2v e = stack_l[slot];
                    if(sp >= (uint32_t)STACK_LDS) {
                        stack_l[slot] = my_spill[sp - STACK_LDS]; // the window moves down: the entry that left it last comes back
                    }
                    if(__uint_as_float(e.y) < t_max) {
                        cur = e.x;
                        need_pop = false;
                        moved = true;
                    }
                }
            }
            w.sp = sp;
        }
        w.cur = cur;
        if(moved & (cur != PT_REF_NONE)) {
            fetch(cur, R);
        }
    }

    // One step of the wavefront: the common step for the lanes on inner nodes; then, if no lane is left on one or `leaf_min` lanes wait for
    // it, the rare one.  Returns false when no lane of the wavefront stands anywhere any more.
    PT_D bool step(Walk &w, Rec &R, int leaf_min, uint32_t &n_nodes, uint32_t &n_leaves) const {
        PT_STAMP(0); // loop back, the caller's code between two steps
        PT_STAMP(7); // (nothing: what a stamp costs)
        const unsigned long long nodes = node_lanes(w.cur), slow = slow_lanes(w.cur);
        if((nodes | slow) == 0ULL) {
            return false;
        }
        PT_STAMP(1); // classification
        if(nodes != 0ULL) {
            n_nodes += (uint32_t)__popcll(nodes);
            node_step(w, R, nodes, nodes & __builtin_amdgcn_uicmp(w.sp, (uint32_t)STACK_LDS, 35));
        }
        PT_STAMP(5); // leaving the common step
        // The rare step, for the leaves and the walks that have to go on popping: its code is long (a triangle test is 100 instructions,
        // a division among them), so the waiting lanes share it -- not before `leaf_min` of them wait, unless no lane stands on a node any
        // more.  (Serving the popping walks at once instead of letting them wait with the leaves: 422 against 430 Msamples/s.)
        if(slow != 0ULL && (nodes == 0ULL || __popcll(slow) >= leaf_min)) {
            // (slow was taken before the common step: a lane that has just reached a leaf is not in it, its record is on its way)
            slow_step(w, R, slow & __builtin_amdgcn_uicmp(w.cur, PT_REF_POPPING, 36), n_leaves);
        }
        PT_STAMP(6); // the rare step (or the test for it)
        return true;
    }
};

// ---- the per-slot word in LDS -----------------------------------------------------------------------------------------------------------
// What the lanes that finish a slot's rays and the shading pass tell each other:
//   visibility of the last vertex's light samples (1 = unoccluded), set by the lanes that finish the shadow rays;
//   rays of the slot still in the queue or being walked;  PT_F_* flags of the slot.
// A lane that finishes a shadow ray adds (visibility bit) - (one ray) with ONE LDS atomic.
// Scene::sampleLights has no upper bound on the samples per vertex (every LightSource + min(2 + log10(E + 1), E) emitters, scene.cpp:226,231).
// Scenes with at most 8 of them -- every scene of the reference's programs -- use a 32-bit word that also holds which samples have a
// contribution waiting in S.nee (COMPACT: bits 0-7 visibility, 8-15 rays, 16-23 flags, 24-31 waiting); scenes with up to PT_MAX_NEE = 32
// use a 64-bit word (WIDE: bits 0-31 visibility, 32-39 rays, 40-47 flags) and keep the waiting mask with the slot's state in HBM
// (S.nee_mask: only the shading pass needs it).  The wide form costs the benchmark scene 5 % (one more load and store per slot and pass,
// 64-bit LDS traffic), which is why the compact one stays.
template<bool WIDE>
struct SlotWord {
    typedef uint32_t T;
    typedef uint32_t __attribute__((address_space(3))) *lds_ptr;
    static constexpr uint32_t j_mask = 7u;
    static PT_D uint32_t vis(T w) { return w & 0xffu; }
    static PT_D uint32_t pending(T w) { return (w >> 8) & 0xffu; }
    static PT_D uint32_t flags(T w) { return (w >> 16) & 0xffu; }
    static PT_D T one_ray() { return 0x100u; }
    static PT_D T make(uint32_t flags, uint32_t rays, uint32_t vis, uint32_t waiting) { return (waiting << 24) | (flags << 16) | (rays << 8) | vis; }
    static PT_D uint32_t waiting(T w, const PtSlots &, size_t) { return w >> 24; }
    static PT_D void keep_waiting(const PtSlots &, size_t, uint32_t) {}
};
template<>
struct SlotWord<true> {
    typedef unsigned long long T;
    typedef unsigned long long __attribute__((address_space(3))) *lds_ptr;
    static constexpr uint32_t j_mask = 31u;
    static PT_D uint32_t vis(T w) { return (uint32_t)w; }
    static PT_D uint32_t pending(T w) { return (uint32_t)(w >> 32) & 0xffu; }
    static PT_D uint32_t flags(T w) { return (uint32_t)(w >> 40) & 0xffu; }
    static PT_D T one_ray() { return 1ULL << 32; }
    static PT_D T make(uint32_t flags, uint32_t rays, uint32_t vis, uint32_t) { return ((T)flags << 40) | ((T)rays << 32) | (T)vis; }
    static PT_D uint32_t waiting(T, const PtSlots &S, size_t p) { return S.nee_mask[p]; }
    static PT_D void keep_waiting(const PtSlots &S, size_t p, uint32_t mask) { S.nee_mask[p] = mask; }
};

#define PT_DEST_SLOT_MASK 0xffffu
#define PT_NO_SLOT 0xffffffffu
#define PT_DEST_J_SHIFT 16

// What a wavefront keeps about itself (everything wave-uniform).
struct WaveCtx {
    uint32_t q_head;   // index of the oldest queued ray inside the wave's ring
    uint32_t q_count;  // queued rays
    uint32_t n_dead;   // slots that will never get a stream again
    bool pool_empty;   // the global stream counter has run out
    uint32_t first_rows; // rows that have not taken their first streams yet (bit per row)
    uint32_t first_base; // the wave's own first streams: first_base + slot number
    uint32_t wave;       // number of the wavefront in the grid
};

// Stream i of a tile job: pixel i of the tiles laid end to end (pt_render_tiles)
PT_D void tile_stream(const PtStreams &T, uint32_t i, int4 &rect, uint64_t &rng, uint32_t &tile) {
    uint32_t lo = 0, hi = T.n_tiles;
    while(hi - lo > 1) {
        const uint32_t mid = (lo + hi) / 2;
        if(T.tile_offset[mid] <= i) {
            lo = mid;
        }
        else {
            hi = mid;
        }
    }
    const int4 t = T.tiles[lo];
    const uint32_t k = i - T.tile_offset[lo];
    const int32_t x = t.x + (int32_t)(k % (uint32_t)t.z), y = t.y + (int32_t)(k / (uint32_t)t.z);
    rect = make_int4(x, y, 1, 1);
    const uint64_t seed = pixel_seed(T.base_seed, x, y);
    rng = seed ^ (~seed << 32); // RandomEngine(seed), base.h:26
    tile = lo;
}

// The small tables of the shading pass in LDS (PT_LDS_TABLE_MAX entries at most each; larger ones are read from global memory)
struct EmisLds {
    const float __attribute__((address_space(3))) *cdf_l;
    lds_f4_cptr rec_l;   // [4 * n_emis]
    lds_f4_cptr light_l; // [6 * n_emis]: the shading record of an emissive triangle (unused for spheres)
    PT_D float cdf(int i) const { return cdf_l[i]; }
    PT_D float4 rec(int i, int k) const { return to_f4(rec_l[4 * i + k]); }
    PT_D V3 tri_normal_at(int i, uint32_t, V3 pos) const {
        uint32_t mat_unused;
        return tri_shade_normal(light_l + 6 * i, pos, mat_unused);
    }
};
struct ShadeTables {
    EmisLds emis;
    lds_f4_cptr materials_l;
    bool emis_in_lds, materials_in_lds;
};

// One shading pass over row `row` of the wave's slots: the state machine of one stream per lane.
//   * a slot without a stream takes the next one from the global counter (or dies when there is none left);
//   * a slot whose rays have all come back first adds the unoccluded light samples of its previous vertex to out_spectrum in the
//     reference's order (worker.cpp:76-103), then looks at the extension ray's hit: miss -> the sample is finished (estimator, next
//     sample or pixel); hit -> shade that vertex: emission (worker.cpp:62-64), Russian-roulette draw (:67-70), light sampling
//     (Scene::sampleLights, scene.cpp:222-289) with one shadow ray per light sample, BSDF sample (:117-131).
// A path that ends at a vertex (roulette) still has that vertex's shadow rays to wait for; where the estimator provably cannot stop
// at this sample, the NEXT sample's camera ray is drawn and traced together with them (PT_F_OVERLAP) -- the draws keep their order.
// The random draws of one stream are consumed strictly in the reference's order because a stream has at most one path in flight
// and every draw of a vertex (roulette, lights, BSDF) is made by the single invocation that shades the vertex.
// New rays go to the wave's ring: extension rays first, then the shadow rays light sample by light sample (ballot + prefix popcount
// give every lane its position; rays of one kind from neighbouring pixels end up in neighbouring lanes of the traversal).
template<bool WIDE>
PT_D void shade_row(const PtDevScene &sc, const PtDevCamera &cam, const PtDevOptions &opt, const PtSlots &S, const PtStreams &T, const PtLocalQueue &Q,
                    WaveCtx &ctx, uint32_t row, uint32_t ls_in, uint32_t lane, size_t slot_base, size_t queue_base, typename SlotWord<WIDE>::lds_ptr word_l, lds_u2_ptr hit_l,
                    float4 *__restrict__ image, PtDevCounters *counters, const ShadeTables &tb, uint32_t &n_samples, uint32_t &n_vertices) {
    // the lane's slot of the wave: lane `lane` of row `row`, or -- in a compacted pass (see the kernel) -- the slot the list names; PT_NO_SLOT = none
    const bool have_slot = ls_in != PT_NO_SLOT;
    const uint32_t ls = have_slot ? ls_in : 0u;
    const size_t p = slot_base + ls;     // slot of the grid
    const unsigned long long lt = (1ULL << lane) - 1ULL;
    const uint32_t n_light_samples = sc.n_lights + sc.n_object_samples;
    typedef SlotWord<WIDE> SW;
    const typename SW::T word = word_l[ls];
    uint32_t flags = SW::flags(word);
    const uint32_t vis_bits = SW::vis(word);
    bool ready = have_slot && !(flags & PT_F_DONE) && SW::pending(word) == 0;
    if(__ballot(ready) == 0ULL) {
        return;
    }

    // ---- slots without a stream: take the next one ------------------------------------------------------------------------------------
    {
        const bool want = ready && !(flags & PT_F_STREAM);
        const unsigned long long want_mask = __ballot(want);
        if(want_mask != 0ULL) {
            // A wavefront's FIRST streams are fixed (no race for the counter at kernel start: the same frame takes the same time twice) and
            // spread over the job.  The wavefront's slots are cut into pieces of `first_lanes` neighbouring slots (8: a quarter of a
            // tile's scanline); piece q of wavefront w starts on the chunk q * waves + w of as many streams, so the pieces of a wavefront
            // lie 1/32 of the first round apart and neighbouring wavefronts work on the same tiles; in a regular tile grid piece q is also
            // moved q steps sideways, so that a wavefront samples the frame's columns as well as its bands.  A wavefront costs what its
            // pixels cost, and the launch lasts as long as the most expensive wavefront: with 256 neighbouring pixels per wavefront the
            // benchmark frame runs at 294 Msamples/s, with four rows from four places (what the race for the counter used to produce)
            // at 393, with 32 pieces of 8 at 440.  Streams beyond the first round come from the global counter as the slots free up.
            uint32_t base = T.n;
            uint32_t mine;
            if((ctx.first_rows >> row) & 1u) {
                ctx.first_rows &= ~(1u << row);
                base = ctx.first_base + row * 64u;
                mine = base + lane;
                if(T.place != nullptr) {
                    // an explicit first round (cost-aware placement, pt_api.cpp): the table names the stream of every slot, or none
                    base = 0;
                    mine = T.place[ctx.first_base + ls];
                }
                else if(T.first_spread) {
                    // piece q of the wavefront (a row, or a part of one) starts on chunk q * waves + w of `first_lanes` streams
                    const uint32_t g = T.first_lanes, q = row * (64u / g) + lane / g;
                    uint32_t chunk = q * T.n_waves + ctx.wave;
                    if(T.tiles_per_row != 0) {
                        const uint32_t per_tile = T.chunks_per_tile * (64u / g), pieces = (T.first_total / T.n_waves) / g;
                        const uint32_t tile = chunk / per_tile, sub = chunk % per_tile;
                        const uint32_t step = T.tiles_per_row >= pieces ? T.tiles_per_row / pieces : 1u;
                        const uint32_t tx = (tile % T.tiles_per_row + q * T.first_shift * step) % T.tiles_per_row, ty = tile / T.tiles_per_row;
                        chunk = (ty * T.tiles_per_row + tx) * per_tile + sub;
                    }
                    base = 0; // (only compared with T.n below)
                    mine = chunk * g + lane % g;
                }
            }
            else {
                if(!ctx.pool_empty) {
                    if(lane == 0) {
                        base = T.first_total + atomicAdd(T.next, (uint32_t)__popcll(want_mask));
                    }
                    base = __builtin_amdgcn_readfirstlane(base);
                }
                mine = base + (uint32_t)__popcll(want_mask & lt);
                if(base >= T.n || base + (uint32_t)__popcll(want_mask) > T.n) {
                    ctx.pool_empty = true;
                }
            }
            const bool got = want && base < T.n && mine < T.n;
            if(got) {
                int4 rc;
                uint64_t r;
                uint32_t tile = 0;
                if(T.rect != nullptr) {
                    rc = T.rect[mine];
                    r = T.rng[mine];
                }
                else {
                    tile_stream(T, mine, rc, r, tile);
                }
                S.stream[p] = mine;
                S.rect[p] = rc;
                S.rng[p] = r;
                S.cursor[p] = 0;
                if(T.cost != nullptr) {
                    __hip_atomic_store(&S.cost[p], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                flags = PT_F_STREAM;
            }
            else if(want) {
                flags = PT_F_DONE;
                ready = false;
                word_l[ls] = SW::make(PT_F_DONE, 0u, 0u, 0u);
            }
            ctx.n_dead += (uint32_t)__popcll(__ballot(want && !got));
        }
    }
    const bool alive = ready;

    // ---- phase A: consume the results of the rays that came back -----------------------------------------------------------------------
    uint64_t rng = 0;
    C4 out = c4(0, 0, 0, 0), spectrum = c4(1, 1, 1, 1);
    V3 ro = v3(0, 0, 0), rd = v3(0, 0, 1);
    float contribution_unweighted = 1.0f;
    double divisor = 1.0, bounce_pd = 1.0;
    int path_length = 0;
    bool shade_vertex = false; // the extension ray hit something
    bool start_sample = false; // generate a camera ray
    float hit_t = -1.0f;
    uint32_t hit_ref = PT_REF_NONE;
    int4 rect = make_int4(0, 0, 0, 0);
    int32_t cursor = 0;
    bool stream_finished = false;

    if(alive) {
        rng = S.rng[p];
        rect = S.rect[p];
        cursor = S.cursor[p];
        if(flags & PT_F_IN_FLIGHT) {
            // Everything the slot may need is requested here, in one batch (one memory round trip instead of one per use): the path
            // state, and -- a single word, to bring the line into the caches -- the shading record of the triangle that was hit, which
            // is the one access of this pass that usually comes from HBM.
            const float4 out4 = S.out[p];
            uint32_t mask = SW::waiting(word, S, p);
            // the first two light samples (most scenes have no more) are fetched with the batch, the others one by one below
            const uint32_t lit = mask & vis_bits;
            float4 nee0 = make_float4(0, 0, 0, 0), nee1 = make_float4(0, 0, 0, 0);
            if(lit & 1u) {
                nee0 = S.nee[p];
            }
            if(lit & 2u) {
                nee1 = S.nee[S.total + p];
            }
            const float4 o4 = S.ray_o[p], d4 = S.ray_d[p], spectrum4 = S.spectrum[p];
            const double divisor_in = S.divisor[p], bounce_pd_in = S.bounce_pd[p];
            const int path_length_in = S.path_length[p];
            uint32_t warm = 0;
            if(flags & PT_F_HAS_EXT) {
                const u2v h = hit_l[ls];
                hit_t = __uint_as_float(h.x);
                hit_ref = h.y;
                if(!(hit_t < 0.0f) && !(hit_ref & PT_REF_SPHERE)) {
                    warm = *reinterpret_cast<const uint32_t *>(sc.tri_shade + 8 * (size_t)(hit_ref & PT_REF_INDEX));
                }
            }
            out = c4(out4);
            // shadow rays of the previous vertex, in light order (worker.cpp:76-103)
            if(lit & 1u) {
                out = out + c4(nee0);
            }
            if(lit & 2u) {
                out = out + c4(nee1);
            }
            mask >>= 2;
            for(uint32_t j = 2; mask != 0; j++, mask >>= 1) {
                if((mask & 1u) && ((vis_bits >> j) & 1u)) {
                    out = out + c4(S.nee[(size_t)j * S.total + p]);
                }
            }
            asm volatile("" ::"v"(warm)); // (the word itself is not used)
            if(flags & PT_F_OVERLAP) {
                // the previous sample is complete now (worker.cpp:141-145, 196-237); it was collected (it had a vertex), it is
                // not the pixel's last sample and the estimator cannot accept at it (see estimator_safe_to_overlap)
                PtEstimator e = S.est[p];
                out.a = 1.0f;
                (void)estimator_add(e, S.cand + p * PT_MAX_CANDIDATES, opt, out);
                e.pixel_sample++;
                n_samples++;
                S.est[p] = e;
                flags &= ~(PT_F_OVERLAP | PT_F_COLLECTED | PT_F_SAFE);
                if(estimator_safe_to_overlap(e, opt)) {
                    flags |= PT_F_SAFE;
                }
                out = c4(0, 0, 0, 0);
            }
            bool finished = true;
            if((flags & PT_F_HAS_EXT) && !(hit_t < 0.0f)) {
                finished = false;
                shade_vertex = true;
            }
            if(finished) {
                // getSample returns (worker.cpp:141-145); run the estimator
                PtEstimator e = S.est[p];
                PtCandidate *cand = S.cand + p * PT_MAX_CANDIDATES;
                bool accepted = false;
                if(flags & PT_F_COLLECTED) {
                    out.a = 1.0f;
                    accepted = estimator_add(e, cand, opt, out);
                }
                e.pixel_sample++;
                n_samples++;
                if(accepted || e.pixel_sample >= opt.max_sample_count) {
                    // pixel finished (worker.cpp:263-319)
                    const C4 value = estimator_finish(e, cand, opt, accepted);
                    const int32_t px = rect.x + cursor % rect.z, py = rect.y + cursor / rect.z;
                    image[(size_t)py * opt.image_width + px] = f4(value);
                    cursor++;
                    flags &= ~PT_F_PIXEL;
                }
                S.est[p] = e;
                flags &= ~(PT_F_IN_FLIGHT | PT_F_HAS_EXT | PT_F_COLLECTED | PT_F_SAFE);
                if((flags & PT_F_PIXEL) && estimator_safe_to_overlap(e, opt)) {
                    flags |= PT_F_SAFE; // for the pixel's next sample, which starts below
                }
                start_sample = true;
            }
            else {
                ro = v3(o4.x, o4.y, o4.z);
                rd = v3(d4.x, d4.y, d4.z);
                contribution_unweighted = o4.w;
                spectrum = c4(spectrum4);
                divisor = divisor_in;
                bounce_pd = bounce_pd_in;
                path_length = path_length_in;
            }
        }
        else {
            start_sample = true;
        }
    }

    // ---- start the next sample / pixel ------------------------------------------------------------------------------------
    bool emit_ext = false;
    Ray ext;
    ext.o = v3(0, 0, 0);
    ext.d = v3(0, 0, 1);
    // camera ray through pixel `cur` of the stream's rectangle (worker.cpp:168-170, camera.cpp:78-113)
    auto shoot_camera = [&](const int4 rc, int32_t cur) {
        const int32_t px = rc.x + cur % rc.z, py = rc.y + cur / rc.z;
        const float one_half = 1.0f / 2.0f;
        const float x_camera = 2 * (((float)px + one_half) / (float)opt.image_width - one_half);
        float y_camera = 2 * (((float)py + one_half) / (float)opt.image_height - one_half);
        y_camera = -y_camera;
        return camera_shoot(cam, x_camera, y_camera, opt.pixel_width, opt.pixel_height, rng);
    };
    if(start_sample) {
        bool have_pixel = false;
        while(cursor < rect.z * rect.w) {
            if(!(flags & PT_F_PIXEL)) {
                PtEstimator e;
                estimator_reset(e, opt);
                S.est[p] = e;
                flags = (flags | PT_F_PIXEL) & ~PT_F_SAFE;
                if(estimator_safe_to_overlap(e, opt)) {
                    flags |= PT_F_SAFE;
                }
                if(opt.max_sample_count <= 0) {
                    // no sample at all: the pixel stays (0, 0, 0, 0) (worker.cpp:193,263-265)
                    const int32_t px = rect.x + cursor % rect.z, py = rect.y + cursor / rect.z;
                    image[(size_t)py * opt.image_width + px] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    cursor++;
                    flags &= ~PT_F_PIXEL;
                    continue;
                }
            }
            have_pixel = true;
            break;
        }
        if(!have_pixel) {
            // the stream has rendered its whole rectangle: hand the engine back and free the slot (it takes a new stream in the next pass)
            stream_finished = true;
            flags = 0;
        }
        else {
            ext = shoot_camera(rect, cursor);
            emit_ext = true;
            out = c4(0, 0, 0, 0);
            spectrum = c4(1, 1, 1, 1);
            contribution_unweighted = 1.0f;
            divisor = 1.0;
            bounce_pd = 1.0;
            path_length = 0;
            flags |= PT_F_IN_FLIGHT | PT_F_HAS_EXT;
        }
    }
    {
        const unsigned long long fin_mask = __ballot(stream_finished);
        if(fin_mask != 0ULL) {
            if(stream_finished) {
                const uint32_t si = S.stream[p];
                if(T.rng != nullptr) {
                    T.rng[si] = rng;
                }
                if(T.cost != nullptr) {
                    T.cost[si] = __hip_atomic_load(&S.cost[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (written by atomics of this wavefront's lanes: read it where they landed)
                }
                if(T.tile_left != nullptr) {
                    // progress: the last pixel of a tile reports the tile to the host (processJob's callback, worker.cpp:354-360)
                    int4 rc_unused;
                    uint64_t r_unused;
                    uint32_t tile = 0;
                    tile_stream(T, si, rc_unused, r_unused, tile);
                    if(atomicSub(&T.tile_left[tile], 1u) == 1u) {
                        __hip_atomic_fetch_add(T.tiles_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
            }
            if(lane == 0) {
                atomicAdd(&counters->streams_done, (unsigned long long)__popcll(fin_mask));
            }
        }
    }

    // ---- vertex, part 1: everything up to the Russian-roulette draw (worker.cpp:50-70) ---------------------------------------
    V3 pos = v3(0, 0, 0), n = v3(0, 1, 0);
    Material mat;
    mat.bsdf = 0;
    bool do_bounce = false;
    float bounce_probability = 1.0f;
    bool want_nee = false;
    bool safe_overlap = false;
    if(shade_vertex) {
        path_length++;
        flags |= PT_F_COLLECTED;
        n_vertices++;
        pos = ro + rd * hit_t;
        uint32_t material_index;
        n = object_normal(sc, hit_ref, pos, material_index);
        mat = tb.materials_in_lds ? material_load(tb.materials_l, material_index) : material_load(sc.materials, material_index);

        out = out + (spectrum * mat.emission) / (float)(divisor * bounce_pd);

        bounce_probability = path_length <= 4 ? 1.0f : 0.1f + 0.1f * fmin_std(contribution_unweighted * get_contribution(spectrum), 1.0f);
        do_bounce = rng_uniform01(rng) < bounce_probability;
        // BSDF::getSpectrum(..., synthetic = true) returns p = 0 for glass and mirror: their light samples never
        // contribute (worker.cpp:92), so no shadow ray is needed -- the light-sampling draws are still consumed below.
        want_nee = mat.bsdf == 0 && n_light_samples > 0;
        safe_overlap = (flags & PT_F_SAFE) != 0; // decided when the sample started (estimator_safe_to_overlap)
        // an extension ray: the bounce (may still be cancelled by the 1E-20 guards, worker.cpp:112,134) or the next sample's camera ray
        emit_ext = do_bounce || safe_overlap;
    }

    // ---- positions in the wave's ring: extension rays first (the long walks start first) -------------------------------------------------
    uint32_t tail = ctx.q_head + ctx.q_count; // may be >= cap: wrapped per entry below
    const unsigned long long ext_mask = __ballot(emit_ext);
    const uint32_t ext_pos = tail + (uint32_t)__popcll(ext_mask & lt);
    tail += (uint32_t)__popcll(ext_mask);
    ctx.q_count += (uint32_t)__popcll(ext_mask);
    auto ring = [&](uint32_t i) -> size_t { return queue_base + (i >= Q.cap ? i - Q.cap : i); };

    // ---- vertex, part 2: light sampling, shadow rays (worker.cpp:73-103) -------------------------------------------------------------------
    uint32_t nee_out_mask = 0;
    uint32_t vis_init = 0;
    uint32_t n_rays = 0;
    const float epsilon = opt.epsilon;
    for(uint32_t j = 0; j < n_light_samples; j++) { // wave-uniform trip count
        bool need_ray = false;
        float4 so = make_float4(0, 0, 0, 0), sd = make_float4(0, 0, 0, 0);
        if(shade_vertex) {
            V3 light_pos;
            C4 light_spectrum;
            float lpd;
            bool valid;
            if(j < sc.n_lights) {
                // PointLightSource: its position, its spectrum, pd = 1 (light.cpp:35-41)
                const float4 lp = sc.lights[2 * j];
                light_pos = v3(lp.x, lp.y, lp.z);
                light_spectrum = c4(sc.lights[2 * j + 1]);
                lpd = 1.0f;
                valid = true;
            }
            else {
                valid = tb.emis_in_lds ? sample_emissive(sc, tb.emis, pos, rng, light_pos, light_spectrum, lpd)
                                       : sample_emissive(sc, EmisGlobal{sc}, pos, rng, light_pos, light_spectrum, lpd);
            }
            if(valid && want_nee) {
                const V3 to_light = light_pos - pos;
                const V3 light_dir = normalize(to_light);
                float shading_factor, shadow_ray_pd;
                const C4 base_spectrum = bsdf_spectrum(mat, rd, light_dir, n, light_spectrum, true, shading_factor, shadow_ray_pd);
                if(shadow_ray_pd > 0.0f) {
                    const C4 combined = (base_spectrum * shading_factor) * spectrum;
                    const C4 weighed = combined / (float)(divisor * bounce_pd * lpd * shadow_ray_pd);
                    // adding +-0 never changes out_spectrum (which is never -0), so such a sample needs no ray
                    if(!(weighed.r == 0.0f && weighed.g == 0.0f && weighed.b == 0.0f)) {
                        S.nee[(size_t)j * S.total + p] = f4(weighed);
                        nee_out_mask |= 1u << j;
                        const float threshold = len(to_light) - epsilon;
                        if(threshold <= 0.0f) {
                            // light_t < 0 || light_t >= threshold holds for every light_t
                            vis_init |= 1u << j;
                        }
                        else {
                            need_ray = true;
                            const V3 o2 = pos + light_dir * epsilon;
                            so = make_float4(o2.x, o2.y, o2.z, threshold);
                            sd = make_float4(light_dir.x, light_dir.y, light_dir.z, __uint_as_float(PT_DEST_SHADOW | (j << PT_DEST_J_SHIFT) | ls));
                        }
                    }
                }
            }
        }
        const unsigned long long ray_mask = __ballot(need_ray);
        if(ray_mask != 0ULL) {
            if(need_ray) {
                const size_t at = ring(tail + (uint32_t)__popcll(ray_mask & lt));
                Q.ray_o[at] = so;
                Q.ray_d[at] = sd;
                n_rays++;
            }
            tail += (uint32_t)__popcll(ray_mask);
            ctx.q_count += (uint32_t)__popcll(ray_mask);
        }
    }

    // ---- vertex, part 3: the bounce (worker.cpp:105-138) ---------------------------------------------------------------------------------
    if(shade_vertex) {
        bool cancel_ext = false;
        if(!do_bounce) {
            // worker.cpp:106-109: the path ends here
        }
        else {
            bounce_pd *= bounce_probability;
            if(bounce_pd <= 1E-20) {
                cancel_ext = true;
            }
            else {
                float ray_factor, ray_pd;
                const Ray next_ray = bsdf_propagate(mat, rd, pos, n, epsilon, rng, ray_factor, ray_pd);
                divisor *= ray_pd;
                divisor /= ray_factor;
                contribution_unweighted *= ray_factor;
                float shading_factor, shading_pd;
                const C4 shaded = bsdf_spectrum(mat, rd, next_ray.d, n, spectrum, false, shading_factor, shading_pd);
                divisor *= shading_pd;
                divisor /= shading_factor;
                contribution_unweighted *= shading_factor;
                spectrum = shaded;
                if(divisor <= 1E-20) {
                    cancel_ext = true;
                }
                else {
                    ext = next_ray;
                }
            }
        }
        const bool path_ends = !do_bounce || cancel_ext;
        if(path_ends && safe_overlap) {
            // start the next sample of this pixel now; this sample is finished by the next pass over the slot (PT_F_OVERLAP)
            ext = shoot_camera(rect, cursor);
            spectrum = c4(1, 1, 1, 1);
            contribution_unweighted = 1.0f;
            divisor = 1.0;
            bounce_pd = 1.0;
            path_length = 0;
            flags |= PT_F_OVERLAP;
        }
        else if(path_ends && emit_ext) {
            // the reserved position stays a hole (a bounce cancelled by the 1E-20 guards: rare)
            Q.ray_d[ring(ext_pos)] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(PT_DEST_NULL));
            emit_ext = false;
        }
        if(emit_ext) {
            flags |= PT_F_HAS_EXT;
        }
        else {
            flags &= ~PT_F_HAS_EXT;
        }
    }

    // ---- write the extension ray and the path state ------------------------------------------------------------------------------
    if(emit_ext) {
        const size_t at = ring(ext_pos);
        Q.ray_o[at] = make_float4(ext.o.x, ext.o.y, ext.o.z, 0.0f);
        Q.ray_d[at] = make_float4(ext.d.x, ext.d.y, ext.d.z, __uint_as_float(ls));
        n_rays++;
    }
    if(alive) {
        word_l[ls] = SW::make(flags, n_rays, vis_init, nee_out_mask);
        SW::keep_waiting(S, p, nee_out_mask);
        S.rng[p] = rng;
        S.cursor[p] = cursor;
        if(flags & PT_F_IN_FLIGHT) {
            S.ray_o[p] = make_float4(ext.o.x, ext.o.y, ext.o.z, contribution_unweighted);
            S.ray_d[p] = make_float4(ext.d.x, ext.d.y, ext.d.z, 0.0f);
            S.spectrum[p] = f4(spectrum);
            S.out[p] = f4(out);
            S.divisor[p] = divisor;
            S.bounce_pd[p] = bounce_pd;
            S.path_length[p] = path_length;
        }
    }
}

// ---- the kernel -------------------------------------------------------------------------------------------------------------------------

#define PT_PATH_STACK_LDS 8 /* entries of a lane's traversal stack kept in LDS (16 KB per workgroup: four workgroups share a CU); deeper ones spill to HBM */
// A scene staged in LDS whose records leave no room for four workgroups per CU beside an 8-entry window gets a 4-entry one (its tree has at most
// 384 records: few walks go deeper, and those spill as on any tree).  176 / 98 / 72 triangles in the benchmark's box: 600 -> 757, 855 -> 949, 802 -> 862
// Msamples/s; where four workgroups fit anyway the small window costs 2-4 % (Cornell 718 -> 705), and on trees in HBM 9 % (profiles/r03_stack_window_ab.txt).
#define PT_PATH_STACK_LDS_SMALL 4
#ifndef PT_COST_LDS_BYTES
#define PT_COST_LDS_BYTES 1024 /* one word per lane: the wave step at which its walk began (stream cost diagnostics); 0 in builds that need the LDS */
#endif
#ifndef PT_PATH_WAVES
#define PT_PATH_WAVES 4 /* 128 VGPRs: the traversal loop has no spills there; three waves per SIMD hide less of the node-fetch latency (profiles/) */
#endif

template<bool WIDE, bool IN_LDS, int STACK_LDS>
__global__ __launch_bounds__(256, PT_PATH_WAVES) void pt_path_kernel(const PtPathArgs *__restrict__ args) {
    typedef SlotWord<WIDE> SW;
    // The argument block is read-only for the whole launch: it is addressed as CONSTANT memory (scalar loads; and pointers loaded from
    // constant memory are known to be global ones, so everything reached through them stays global_load / global_store).
    const args_c4 A4 = (args_c4)args;
    const PtPathArgs *A = (const PtPathArgs *)A4;
    // what the traversal loop needs, read once
    const int rows = A->rows, slots_per_wave = A->slots_per_wave, refill_idle = A->refill_idle, min_ready = A->min_ready, burst_steps = A->burst_steps,
              leaf_min = A->leaf_min, ready_shift = A->ready_shift, pass_q_low = A->early_ready > 0 ? A->pass_q_low : 0, early_ready = A->early_ready;
    // diagnostic (PT_DEBUG_LANES): only the first `debug_lanes` lanes of a wavefront take rays -- throughput against walks per step with everything else equal
    const unsigned long long lane_cap = A->debug_lanes >= 64 ? ~0ULL : ((1ULL << A->debug_lanes) - 1ULL);
    PtLocalQueue Q = A->Q;
    RootBox root;
    root.ref = A->sc.root_ref;
    for(int k = 0; k < 3; k++) {
        root.lo[k] = A->sc.root_lo[k];
        root.hi[k] = A->sc.root_hi[k];
    }
    extern __shared__ __align__(16) unsigned char lds_raw[];
    // LDS of the workgroup: traversal stacks [STACK_LDS][256] | hit records [4 waves][rows * 64] | slot words [4 waves][rows * 64] |
    // emitter and material tables (PT_LDS_TABLE_BYTES) | start step of every lane's walk | (small scenes) the whole tree and all triangle records
    const int tid = threadIdx.x;
    const uint32_t lane = (uint32_t)tid & 63u;
    const uint32_t wave_in_block = (uint32_t)tid >> 6;
    const uint32_t wave = blockIdx.x * 4u + wave_in_block;
    const uint32_t n_slots = (uint32_t)rows * 64u;
    unsigned char *at = lds_raw;
    lds_u2_ptr stack_l = (lds_u2_ptr)reinterpret_cast<uint2 *>(at) + tid;
    at += (size_t)STACK_LDS * 256 * sizeof(uint2);
    lds_u2_ptr hit_l = (lds_u2_ptr)reinterpret_cast<uint2 *>(at) + wave_in_block * n_slots;
    at += (size_t)4 * n_slots * sizeof(uint2);
    typename SW::lds_ptr word_l = (typename SW::lds_ptr)reinterpret_cast<typename SW::T *>(at) + wave_in_block * n_slots;
    at += (size_t)4 * n_slots * sizeof(typename SW::T);
    float *cdf_l = reinterpret_cast<float *>(at);
    at += (size_t)PT_LDS_TABLE_MAX * sizeof(float);
    float4 *emis_l = reinterpret_cast<float4 *>(at);
    at += (size_t)PT_LDS_TABLE_MAX * 4 * sizeof(float4);
    float4 *light_l = reinterpret_cast<float4 *>(at);
    at += (size_t)PT_LDS_TABLE_MAX * 6 * sizeof(float4);
    float4 *materials_l = reinterpret_cast<float4 *>(at);
    at += (size_t)PT_LDS_TABLE_MAX * 4 * sizeof(float4);
    uint32_t __attribute__((address_space(3))) *born_l = (uint32_t __attribute__((address_space(3))) *)reinterpret_cast<uint32_t *>(at) + tid; // wave step at which the lane's walk began
    // (the same kilobyte holds the list of ready slots of a compacted shading pass, one byte per slot of up to four rows, when the cost diagnostics are off)
    typedef unsigned char __attribute__((address_space(3))) *lds_u8_ptr;
    lds_u8_ptr list_l = (lds_u8_ptr)reinterpret_cast<unsigned char *>(at) + wave_in_block * 256u;
    at += (size_t)PT_COST_LDS_BYTES;
    float4 *lds_recs = reinterpret_cast<float4 *>(at); // (small scenes) every record, in the order of `recs`
    const bool cost_on = PT_COST_LDS_BYTES != 0 && A->T.cost != nullptr;
    uint32_t *const slot_cost = A->S.cost;

    ShadeTables tb;
    tb.emis.cdf_l = (const float __attribute__((address_space(3))) *)cdf_l;
    tb.emis.rec_l = (lds_f4_cptr)emis_l;
    tb.emis.light_l = (lds_f4_cptr)light_l;
    tb.materials_l = (lds_f4_cptr)materials_l;
    {
        const uint32_t n_emis = A->sc.n_emis, n_materials = A->sc.n_materials;
        tb.emis_in_lds = n_emis > 0 && n_emis <= PT_LDS_TABLE_MAX;
        tb.materials_in_lds = n_materials > 0 && n_materials <= PT_LDS_TABLE_MAX;
        if(tb.emis_in_lds) {
            const float4 *src_emis = A->sc.emis, *src_shade = A->sc.tri_shade;
            const float *src_cdf = A->sc.emis_cdf;
            for(uint32_t i = tid; i < n_emis; i += 256) {
                cdf_l[i] = src_cdf[i];
            }
            for(uint32_t i = tid; i < 4 * n_emis; i += 256) {
                emis_l[i] = src_emis[i];
            }
            for(uint32_t i = tid; i < 6 * n_emis; i += 256) {
                const uint32_t ref = __float_as_uint(src_emis[4 * (i / 6) + 2].y);
                light_l[i] = (ref & PT_REF_SPHERE) ? make_float4(0, 0, 0, 0) : src_shade[8 * (size_t)(ref & PT_REF_INDEX) + i % 6];
            }
        }
        if(tb.materials_in_lds) {
            const float4 *src_materials = A->sc.materials;
            for(uint32_t i = tid; i < 4 * n_materials; i += 256) {
                materials_l[i] = src_materials[i];
            }
        }
    }

    if(IN_LDS) {
        const uint32_t n_lds_quads = 4u * (A->sc.pair_base + A->sc.n_pairs);
        const float4 *src_recs = A->sc.recs;
        for(uint32_t i = tid; i < n_lds_quads; i += 256) {
            lds_recs[i] = src_recs[i];
        }
    }
    for(uint32_t i = lane; i < n_slots; i += 64) {
        // no stream, nothing pending: ready to take a stream.  (A small job uses only the first slots of every wavefront: its streams are
        // spread over all the wavefronts the chip holds, because a stream's samples are sequential and only more wavefronts shorten the chain.)
        word_l[i] = i < (uint32_t)slots_per_wave ? (typename SW::T)0 : SW::make(PT_F_DONE, 0u, 0u, 0u);
    }
    __syncthreads(); // the only barrier: from here on the four wavefronts of the workgroup never wait for each other

    Tracer<STACK_LDS, IN_LDS> tr;
    if(IN_LDS) {
        tr.recs = (typename RecPtr<IN_LDS>::type)(lds_f4_cptr)lds_recs;
    }
    else {
        tr.recs = (typename RecPtr<IN_LDS>::type)(glb_f4_cptr)A->sc.recs;
    }
    tr.stack_l = stack_l;
    tr.my_spill = (glb_u2_ptr)(A->spill + ((size_t)wave * 64 + lane) * A->spill_depth);

    const size_t slot_base = (size_t)wave * n_slots;
    const size_t queue_base = (size_t)wave * Q.cap;
    WaveCtx ctx;
    ctx.q_head = 0;
    ctx.q_count = 0;
    ctx.n_dead = n_slots - (uint32_t)slots_per_wave;
    ctx.pool_empty = false;
    ctx.first_rows = (1u << rows) - 1u;
    ctx.first_base = wave * (uint32_t)slots_per_wave;
    ctx.wave = wave;

    bool active = false;
    Walk w;
    w.o = v3(0, 0, 0);
    w.d = v3(0, 0, 1);
    w.inv = v3(0, 0, 0);
    w.pack();
    w.thr = 0.0f;
    w.dest = 0;
    w.best_t = 0.0f;
    w.best_ref = PT_REF_NONE;
    w.set_t_max(FLT_MAX);
    w.cur = PT_REF_NONE;
    w.sp = 0;
    w.occluded = false;
    typename Tracer<STACK_LDS, IN_LDS>::Rec rec;
    rec.r0 = rec.r1 = rec.r2 = rec.r3 = (f4v){0.0f, 0.0f, 0.0f, 0.0f};
    float4 win_o = make_float4(0, 0, 0, 0), win_d = make_float4(0, 0, 0, __uint_as_float(PT_DEST_NULL)); // the lane's ray of the ring's window
    uint32_t n_nodes = 0, n_leaves = 0, n_rays = 0, n_shadow = 0, n_samples = 0, n_vertices = 0;
    uint32_t w_steps = 0, w_passes = 0; // wave-level diagnostics (same value in every lane)
#ifdef PT_PATH_TIMING
    unsigned long long t_shade = 0, t_burst = 0;
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif

    // Two nested loops.  The inner one is the hot path -- retire, hand out queued rays, a burst of traversal steps -- and contains no
    // shading code, so its registers are allocated for it alone; it is left when a shading pass is called for (queue empty, enough
    // slots ready) or when the wavefront has nothing left to do.  The outer one runs the shading pass.
    bool want_pass = false;
    for(;;) {
        if(want_pass) {
            want_pass = false;
            w_passes++;
#ifdef PT_PATH_TIMING
            const unsigned long long t_pass = __builtin_amdgcn_s_memtime();
#endif
            // The pass reads its arguments from *A now (an opaque zero offset keeps the compiler from reading them once, above the
            // traversal loop, and carrying them through it).
            size_t pass_offset = 0;
            asm volatile("" : "+s"(pass_offset));
            const PtPathArgs *P = (const PtPathArgs *)(args_c4)((const char __attribute__((address_space(4))) *)A4 + pass_offset);
            uint32_t *const walk_save = P->walk_save;
            const size_t save_stride = P->save_stride;
            // Nothing of the traversal lives in registers across a shading pass (which needs them all): the walks in progress and the
            // lane's counters are parked in this lane's column of the save area and read back afterwards.
            {
                uint32_t *sv = walk_save + (size_t)wave * 64 + lane;
                const size_t st = save_stride;
                sv[0 * st] = __float_as_uint(w.o.x);
                sv[1 * st] = __float_as_uint(w.o.y);
                sv[2 * st] = __float_as_uint(w.o.z);
                sv[3 * st] = __float_as_uint(w.d.x);
                sv[4 * st] = __float_as_uint(w.d.y);
                sv[5 * st] = __float_as_uint(w.d.z);
                sv[6 * st] = __float_as_uint(w.thr);
                sv[7 * st] = w.dest;
                sv[8 * st] = __float_as_uint(w.best_t);
                sv[9 * st] = w.best_ref;
                sv[10 * st] = __float_as_uint(w.t_max);
                sv[11 * st] = w.cur;
                sv[12 * st] = w.sp | (w.occluded ? 0x80000000u : 0u);
                sv[13 * st] = n_nodes;
                sv[14 * st] = n_leaves;
                sv[15 * st] = n_rays;
                sv[16 * st] = n_shadow;
            }
            // A pass costs a chain of memory round trips per ROW it visits, however few of the row's slots take part.  Once streams end
            // for good (adaptive sampling stops pixels early; the last streams of any job) the ready slots thin out in every row alike, so
            // when they fit fewer chunks of 64 than they occupy rows, the pass runs over a LIST of them instead (slot numbers in LDS, in
            // slot order): lane i of chunk k shades the (64 k + i)-th ready slot.  Its state accesses are gathers then, which is why a
            // well-filled pass keeps the rows.  (The first round of streams is dealt by row: no list before every row has had its turn.)
            uint32_t n_listed = 0;
            bool compact = false;
            if(PT_COST_LDS_BYTES >= 1024 && !cost_on && P->compact_passes != 0 && rows > 1 && rows <= 4 && ctx.first_rows == 0u) {
                uint32_t rows_used = 0;
                for(uint32_t r = 0; r < (uint32_t)rows; r++) {
                    const typename SW::T word = word_l[r * 64 + lane];
                    const bool is_ready = !(SW::flags(word) & PT_F_DONE) && SW::pending(word) == 0;
                    const unsigned long long m = __ballot(is_ready);
                    if(is_ready) {
                        list_l[n_listed + (uint32_t)__popcll(m & ((1ULL << lane) - 1ULL))] = (unsigned char)(r * 64 + lane);
                    }
                    n_listed += (uint32_t)__popcll(m);
                    rows_used += m != 0ULL ? 1u : 0u;
                }
                compact = (n_listed + 63u) / 64u < rows_used;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); // (the list is read by other lanes of this wavefront)
            }
            const uint32_t n_chunks = compact ? (n_listed + 63u) / 64u : (uint32_t)rows;
#pragma unroll 1
            for(uint32_t k = 0; k < n_chunks; k++) {
                // (in a compacted pass `row` is not used: the first round is over)
                uint32_t ls = k * 64u + lane;
                if(compact) {
                    ls = ls < n_listed ? (uint32_t)list_l[ls] : PT_NO_SLOT;
                }
                shade_row<WIDE>(P->sc, P->cam, P->opt, P->S, P->T, P->Q, ctx, k, ls, lane, slot_base, queue_base, word_l, hit_l, P->image, P->counters, tb, n_samples, n_vertices);
            }
            // the rays just written are read back by other lanes of this wavefront
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
            {
                const uint32_t *sv = walk_save + (size_t)wave * 64 + lane;
                const size_t st = save_stride;
                w.o = v3(__uint_as_float(sv[0 * st]), __uint_as_float(sv[1 * st]), __uint_as_float(sv[2 * st]));
                w.d = v3(__uint_as_float(sv[3 * st]), __uint_as_float(sv[4 * st]), __uint_as_float(sv[5 * st]));
                w.inv = slab_inverse(w.d);
                w.pack();
                w.thr = __uint_as_float(sv[6 * st]);
                w.dest = sv[7 * st];
                w.best_t = __uint_as_float(sv[8 * st]);
                w.best_ref = sv[9 * st];
                w.set_t_max(__uint_as_float(sv[10 * st]));
                w.cur = sv[11 * st];
                const uint32_t packed = sv[12 * st];
                w.sp = packed & 0x7fffffffu;
                w.occluded = (packed >> 31) != 0;
                n_nodes = sv[13 * st];
                n_leaves = sv[14 * st];
                n_rays = sv[15 * st];
                n_shadow = sv[16 * st];
            }
#ifdef PT_PATH_TIMING
            t_shade += __builtin_amdgcn_s_memtime() - t_pass;
#endif
            // the ring has new rays: its first 64 go to the lanes' window registers (which did not live across the pass)
            win_o = make_float4(0, 0, 0, 0);
            win_d = make_float4(0, 0, 0, __uint_as_float(PT_DEST_NULL));
            if(lane < ctx.q_count) {
                uint32_t i = ctx.q_head + lane;
                i = i >= Q.cap ? i - Q.cap : i;
                win_o = Q.ray_o[queue_base + i];
                win_d = Q.ray_d[queue_base + i];
            }
            // the record registers do not live across a shading pass: walks in progress fetch theirs again
            rec.r0 = rec.r1 = rec.r2 = rec.r3 = (f4v){0.0f, 0.0f, 0.0f, 0.0f};
                    if(active && w.cur < PT_REF_POPPING) { // (a walk that is over or about to pop stands on no record)
                tr.fetch(w.cur, rec);
            }
        }

        bool finished = false;
        for(;;) {
            // ---- 1. retire finished walks: the result goes to the slot's words in LDS -----------------------------------------------------
            if(active && w.cur == PT_REF_NONE) {
                const uint32_t ls = w.dest & PT_DEST_SLOT_MASK;
                if(w.dest & PT_DEST_SHADOW) {
                    const uint32_t j = (w.dest >> PT_DEST_J_SHIFT) & SW::j_mask;
                    // one ray less pending; an unoccluded light sample sets its visibility bit
                    __hip_atomic_fetch_add(&word_l[ls], (typename SW::T)(w.occluded ? 0u : (1u << j)) - SW::one_ray(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                else {
                    const u2v h = {__float_as_uint(w.best_ref == PT_REF_NONE ? -1.0f : w.best_t), w.best_ref};
                    hit_l[ls] = h;
                    __hip_atomic_fetch_add(&word_l[ls], (typename SW::T)0 - SW::one_ray(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if(cost_on) {
                    // what the stream cost: the steps this wavefront made while the ray was walking (a stream's rays follow one another,
                    // so their sum is the length of its chain in steps)
                    __hip_atomic_fetch_add(&slot_cost[slot_base + ls], w_steps - *born_l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                active = false;
            }

            // ---- 2. idle lanes: hand out queued rays; with the queue empty, see whether enough slots are ready for a shading pass --------
            const unsigned long long idle_mask = __ballot(!active) & lane_cap;
            const uint32_t n_idle = (uint32_t)__popcll(idle_mask);
            if(n_idle >= (uint32_t)refill_idle) {
                if(ctx.q_count <= (uint32_t)pass_q_low && ctx.n_dead < n_slots) {
                    // slots whose rays have all come back (or that wait for a stream)
                    uint32_t n_ready = 0;
                    for(uint32_t r = 0; r < (uint32_t)rows; r++) {
                        const typename SW::T word = word_l[r * 64 + lane];
                        n_ready += (uint32_t)__popcll(__ballot(!(SW::flags(word) & PT_F_DONE) && SW::pending(word) == 0));
                    }
                    // enough of them for a pass: `min_ready`, or a share of the slots that are still alive -- a wavefront that is down to its
                    // last few streams must not make each of them wait for all the others (their samples are sequential: the launch lasts as
                    // long as its slowest stream)
                    const uint32_t live_share = (n_slots - ctx.n_dead) >> ready_shift;
                    const uint32_t need = live_share < (uint32_t)min_ready ? (live_share > 1u ? live_share : 1u) : (uint32_t)min_ready;
                    // (with rays left in the ring -- pass_q_low > 0 -- the pass is an early one: it tops the ring up before the lanes run dry, and
                    // is only worth its fixed price when `early_ready` slots take part)
                    if(ctx.q_count == 0 ? (n_ready >= need || n_idle == (uint32_t)__popcll(lane_cap)) : n_ready >= (uint32_t)early_ready) {
                        want_pass = true;
                        break;
                    }
                }
                if(ctx.q_count > 0) {
                    // The ring's next 64 rays are already in registers, one per lane (requested when the ring's head last moved:
                    // reading them here would stall the whole wavefront, walks in progress included, for a memory round trip); an
                    // idle lane takes the ray of the lane whose number is its rank among the idle ones.
                    const uint32_t take = ctx.q_count < n_idle ? ctx.q_count : n_idle;
                    const uint32_t rank = (uint32_t)__popcll(idle_mask & ((1ULL << lane) - 1ULL));
                    const int src = (int)(rank & 63u);
                    const float4 ro = make_float4(__shfl(win_o.x, src), __shfl(win_o.y, src), __shfl(win_o.z, src), __shfl(win_o.w, src));
                    const float4 rd = make_float4(__shfl(win_d.x, src), __shfl(win_d.y, src), __shfl(win_d.z, src), __shfl(win_d.w, src));
                    if(!active && rank < take && __float_as_uint(rd.w) != PT_DEST_NULL) {
                        tr.start(w, rec, root, ro, rd);
                        if(cost_on) {
                            *born_l = w_steps;
                        }
                        active = true;
                        n_rays++;
                        n_shadow += (w.dest & PT_DEST_SHADOW) ? 1u : 0u;
                    }
                    ctx.q_head += take;
                    ctx.q_head = ctx.q_head >= Q.cap ? ctx.q_head - Q.cap : ctx.q_head;
                    ctx.q_count -= take;
                    if(lane < ctx.q_count) {
                        uint32_t i = ctx.q_head + lane;
                        i = i >= Q.cap ? i - Q.cap : i;
                        win_o = Q.ray_o[queue_base + i];
                        win_d = Q.ray_d[queue_base + i];
                    }
                }
            }
            if(__ballot(active) == 0ULL) {
                if(ctx.q_count == 0 && ctx.n_dead >= n_slots) {
                    finished = true; // every slot is dead, nothing queued, nothing walking
                    break;
                }
                continue;
            }

            // ---- 3. a burst of traversal steps ---------------------------------------------------------------------------------------------
            // Lanes that stand on a leaf wait (their order of visits is unchanged) until `leaf_min` of them can share the leaf code, or no
            // lane has an inner node left; the leaf test then rides along with the other lanes' node step (one memory round trip for both).
#ifdef PT_PATH_TIMING
            const unsigned long long t_b0 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll 1
            for(int burst = 0; burst < burst_steps; burst++) {
                // (a lane without a walk in progress has cur == PT_REF_NONE: retire leaves it there, and nothing else changes it)
                w_steps++;
                if(!tr.step(w, rec, leaf_min, n_nodes, n_leaves)) {
                    w_steps--;
                    break;
                }
            }
#ifdef PT_PATH_TIMING
            t_burst += __builtin_amdgcn_s_memtime() - t_b0;
#endif
        }
        if(finished) {
            break;
        }
    }

    // Work counters: every wave owns one 64-byte slot (plain stores; atomics on a shared line from every wave serialise at the memory side)
    for(int off = 32; off > 0; off >>= 1) {
        n_rays += __shfl_down(n_rays, off); // (n_nodes and n_leaves are counted for the whole wavefront: Tracer::step)
        n_shadow += __shfl_down(n_shadow, off);
        n_samples += __shfl_down(n_samples, off);
        n_vertices += __shfl_down(n_vertices, off);
    }
    if(lane == 0) {
        unsigned long long *slot = A->wave_counters + 8 * (size_t)wave;
        slot[0] += n_nodes;
        slot[1] += n_leaves;
        slot[2] += n_rays;
        slot[3] += n_shadow;
        slot[4] += w_steps;
        slot[5] += w_passes;
        slot[6] += n_samples;
        slot[7] += n_vertices;
#ifdef PT_PATH_TIMING
        // diagnostic build: shader-clock cycles of this wavefront in shading passes, in traversal bursts, and in all
        slot[5] = (unsigned long long)w_passes | ((t_shade >> 10) << 32);
        slot[4] = (unsigned long long)w_steps | ((t_burst >> 10) << 32);
        slot[3] = (unsigned long long)n_shadow | (((__builtin_amdgcn_s_memtime() - t_begin) >> 10) << 32);
#endif
    }
}

// Scene::getIntersection for a batch of rays: one walk per lane, the same traversal machinery
template<int STACK_LDS, bool IN_LDS>
__global__ __launch_bounds__(256) void pt_closest_kernel(PtDevScene sc, const float *__restrict__ rays6, uint32_t n, uint2 *__restrict__ out, uint2 *__restrict__ spill,
                                                         uint32_t spill_depth) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int tid = threadIdx.x;
    lds_u2_ptr stack_l = (lds_u2_ptr)reinterpret_cast<uint2 *>(lds_raw) + tid;
    float4 *lds_recs = reinterpret_cast<float4 *>(lds_raw + (size_t)STACK_LDS * 256 * sizeof(uint2));
    if(IN_LDS) {
        for(uint32_t i = tid; i < 4u * (sc.pair_base + sc.n_pairs); i += 256) {
            lds_recs[i] = sc.recs[i];
        }
        __syncthreads();
    }
    Tracer<STACK_LDS, IN_LDS> tr;
    if(IN_LDS) {
        tr.recs = (typename RecPtr<IN_LDS>::type)(lds_f4_cptr)lds_recs;
    }
    else {
        tr.recs = (typename RecPtr<IN_LDS>::type)(glb_f4_cptr)sc.recs;
    }
    tr.stack_l = stack_l;
    const size_t gid = (size_t)blockIdx.x * 256 + tid;
    tr.my_spill = (glb_u2_ptr)(spill + gid * spill_depth);
    if(gid >= n) {
        return;
    }
    const float *r = rays6 + 6 * gid;
    Walk w;
    typename Tracer<STACK_LDS, IN_LDS>::Rec rec;
    rec.r0 = rec.r1 = rec.r2 = rec.r3 = (f4v){0.0f, 0.0f, 0.0f, 0.0f};
    RootBox root;
    root.ref = sc.root_ref;
    for(int k = 0; k < 3; k++) {
        root.lo[k] = sc.root_lo[k];
        root.hi[k] = sc.root_hi[k];
    }
    tr.start(w, rec, root, make_float4(r[0], r[1], r[2], 0.0f), make_float4(r[3], r[4], r[5], __uint_as_float(0u)));
    uint32_t n_nodes = 0, n_leaves = 0;
    while(tr.step(w, rec, 1, n_nodes, n_leaves)) {
    }
    out[gid] = make_uint2(__float_as_uint(w.best_ref == PT_REF_NONE ? -1.0f : w.best_t), w.best_ref);
}

// ---- diagnostic: where the cycles of a traversal step go -------------------------------------------------------------------------------
// One walk per lane as in pt_closest_kernel, but only the first `lanes_per_wave` lanes of every wavefront get a ray, and every step is
// stamped (s_memtime): cycles spent waiting for the record that was requested at the end of the previous step, and everything else.
// out[ray] = (steps, cycles waiting for records, cycles of the whole walk, cycles of two back-to-back stamps = the stamps' own price).
template<int STACK_LDS, bool STAMP>
__global__ __launch_bounds__(256) void pt_steptime_kernel(PtDevScene sc, const float *__restrict__ rays6, uint32_t n, uint32_t lanes_per_wave, uint4 *__restrict__ out,
                                                          uint2 *__restrict__ spill, uint32_t spill_depth) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int tid = threadIdx.x;
    const uint32_t lane = (uint32_t)tid & 63u;
    const uint32_t wave = blockIdx.x * 4u + ((uint32_t)tid >> 6);
    Tracer<STACK_LDS, false> tr;
    tr.recs = (glb_f4_cptr)sc.recs;
    tr.stack_l = (lds_u2_ptr)reinterpret_cast<uint2 *>(lds_raw) + tid;
    const size_t gid = (size_t)blockIdx.x * 256 + tid;
    tr.my_spill = (glb_u2_ptr)(spill + gid * spill_depth);
    const uint32_t ray = wave * lanes_per_wave + lane;
    if(lane >= lanes_per_wave || ray >= n) {
        return;
    }
    const float *r = rays6 + 6 * (size_t)ray;
    Walk w;
    typename Tracer<STACK_LDS, false>::Rec rec;
    rec.r0 = rec.r1 = rec.r2 = rec.r3 = (f4v){0.0f, 0.0f, 0.0f, 0.0f};
    RootBox root;
    root.ref = sc.root_ref;
    for(int k = 0; k < 3; k++) {
        root.lo[k] = sc.root_lo[k];
        root.hi[k] = sc.root_hi[k];
    }
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
    const unsigned long long t_again = __builtin_amdgcn_s_memtime();
#ifdef PT_STEP_STAMPS
    tr.stamp_last = t_again;
#endif
    tr.start(w, rec, root, make_float4(r[0], r[1], r[2], 0.0f), make_float4(r[3], r[4], r[5], __uint_as_float(0u)));
    uint32_t n_nodes = 0, n_leaves = 0, steps = 0;
    unsigned long long waiting = 0;
    for(;;) {
        if(STAMP) {
            const unsigned long long t1 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            const unsigned long long t2 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            waiting += t2 - t1;
        }
        if(!tr.step(w, rec, 1, n_nodes, n_leaves)) {
            break;
        }
        steps += 1u;
    }
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    out[ray] = make_uint4(steps, (uint32_t)waiting, (uint32_t)(t_end - t_begin), (uint32_t)(t_again - t_begin));
#ifdef PT_STEP_STAMPS
    // (the stamped build reports its segments behind the n results: 8 x 8 bytes per ray)
    unsigned long long *seg = reinterpret_cast<unsigned long long *>(out + n) + 8 * (size_t)ray;
    for(int k = 0; k < 8; k++) {
        seg[k] = tr.stamp_acc[k];
    }
#endif
}

// ---- diagnostic: the traversal alone on the rays of a finished render ----------------------------------------------------------------
// With PT_RING_LOG_RAYS set, the wavefronts' rings are long enough never to wrap, so after a render they hold every ray of the frame
// in the order the wavefront traced them.  This kernel replays them: the same hand-out / burst / leaf-batching loop as the path
// kernel, no shading, results folded into a checksum -- at WAVES wavefronts per SIMD, which the path kernel cannot choose freely
// (the shading code's registers cap it at four).  It answers what a tracer that is not tied to the shading code would deliver.
// Wavefront v replays part (v / n_logs) of `parts` equal parts of ring (v % n_logs).
template<int STACK_LDS, int WAVES>
__global__ __launch_bounds__(256, WAVES) void pt_replay_kernel(PtDevScene sc, PtLocalQueue Q, uint32_t n_logs, uint32_t parts, int refill_idle, int burst_steps,
                                                                int leaf_min, uint2 *__restrict__ spill, uint32_t spill_depth,
                                                                unsigned long long *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int tid = threadIdx.x;
    const uint32_t lane = (uint32_t)tid & 63u;
    const uint32_t wave = blockIdx.x * 4u + ((uint32_t)tid >> 6);
    if(wave >= n_logs * parts) {
        return;
    }
    Tracer<STACK_LDS, false> tr;
    tr.recs = (glb_f4_cptr)sc.recs;
    tr.stack_l = (lds_u2_ptr)reinterpret_cast<uint2 *>(lds_raw) + tid;
    tr.my_spill = (glb_u2_ptr)(spill + ((size_t)wave * 64 + lane) * spill_depth);
    RootBox root;
    root.ref = sc.root_ref;
    for(int k = 0; k < 3; k++) {
        root.lo[k] = sc.root_lo[k];
        root.hi[k] = sc.root_hi[k];
    }
    const size_t base = (size_t)(wave % n_logs) * Q.cap;
    // rays written: the prefix of the ring whose direction words are not the 0xff fill
    uint32_t lo = 0, hi = Q.cap;
    while(lo < hi) {
        const uint32_t mid = (lo + hi) / 2;
        if(__float_as_uint(Q.ray_d[base + mid].x) == 0xffffffffu) {
            hi = mid;
        }
        else {
            lo = mid + 1;
        }
    }
    const uint32_t part = wave / n_logs;
    uint32_t pos = (uint32_t)((unsigned long long)lo * part / parts);
    const uint32_t end = (uint32_t)((unsigned long long)lo * (part + 1) / parts);

    bool active = false;
    Walk w;
    w.o = v3(0, 0, 0);
    w.d = v3(0, 0, 1);
    w.inv = v3(0, 0, 0);
    w.pack();
    w.thr = 0.0f;
    w.dest = 0;
    w.best_t = 0.0f;
    w.best_ref = PT_REF_NONE;
    w.set_t_max(FLT_MAX);
    w.cur = PT_REF_NONE;
    w.sp = 0;
    w.occluded = false;
    typename Tracer<STACK_LDS, false>::Rec rec;
    rec.r0 = rec.r1 = rec.r2 = rec.r3 = (f4v){0.0f, 0.0f, 0.0f, 0.0f};
    uint32_t n_nodes = 0, n_leaves = 0, n_rays = 0, checksum = 0, w_steps = 0;
    for(;;) {
        if(active && w.cur == PT_REF_NONE) {
            checksum += (w.dest & PT_DEST_SHADOW) ? (w.occluded ? 1u : 2u) : (w.best_ref ^ __float_as_uint(w.best_t));
            active = false;
        }
        const unsigned long long idle_mask = __ballot(!active);
        const uint32_t n_idle = (uint32_t)__popcll(idle_mask);
        if(n_idle >= (uint32_t)refill_idle && pos < end) {
            const uint32_t left = end - pos;
            const uint32_t take = left < n_idle ? left : n_idle;
            if(!active) {
                const uint32_t rank = (uint32_t)__popcll(idle_mask & ((1ULL << lane) - 1ULL));
                if(rank < take) {
                    const float4 ro = Q.ray_o[base + pos + rank];
                    const float4 rd = Q.ray_d[base + pos + rank];
                    if(__float_as_uint(rd.w) != PT_DEST_NULL) {
                        tr.start(w, rec, root, ro, rd);
                        active = true;
                        n_rays++;
                    }
                }
            }
            pos += take;
        }
        if(__ballot(active) == 0ULL) {
            if(pos >= end) {
                break;
            }
            continue;
        }
#pragma unroll 1
        for(int burst = 0; burst < burst_steps; burst++) {
            w_steps++;
            if(!tr.step(w, rec, leaf_min, n_nodes, n_leaves)) {
                w_steps--;
                break;
            }
        }
    }
    for(int off = 32; off > 0; off >>= 1) {
        n_rays += __shfl_down(n_rays, off); // (n_nodes and n_leaves are counted for the whole wavefront: Tracer::step)
        checksum += __shfl_down(checksum, off);
    }
    if(lane == 0) {
        atomicAdd(&out[0], (unsigned long long)n_rays);
        atomicAdd(&out[1], (unsigned long long)n_nodes);
        atomicAdd(&out[2], (unsigned long long)n_leaves);
        atomicAdd(&out[3], (unsigned long long)w_steps);
        atomicAdd(&out[4], (unsigned long long)checksum);
    }
}

template<int STACK_LDS, int WAVES>
int launch_replay(hipStream_t stream, const PtDevScene &scene, const PtLocalQueue &Q, uint32_t n_logs, uint32_t parts, const PtPathConfig &cfg, uint2 *spill,
                  unsigned long long *out) {
    const size_t lds = (size_t)STACK_LDS * 256 * sizeof(uint2);
    int blocks = 0;
    if(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, pt_replay_kernel<STACK_LDS, WAVES>, 256, lds) != hipSuccess) {
        blocks = -1;
    }
    const uint32_t waves = n_logs * parts;
    hipLaunchKernelGGL((pt_replay_kernel<STACK_LDS, WAVES>), dim3((waves + 3) / 4), dim3(256), lds, stream, scene, Q, n_logs, parts, cfg.refill_idle,
                       cfg.burst_steps, cfg.leaf_min, spill, cfg.spill_depth, out);
    return blocks;
}

template<bool WIDE, bool IN_LDS, int STACK_LDS>
void launch_path(hipStream_t stream, const PtPathConfig &cfg, const PtPathArgs *d_args) {
    hipLaunchKernelGGL((pt_path_kernel<WIDE, IN_LDS, STACK_LDS>), dim3(cfg.grid), dim3(256), cfg.lds_bytes, stream, d_args);
}

template<bool WIDE, bool IN_LDS, int STACK_LDS>
void occupancy(size_t lds_bytes, int *out) {
    int blocks = 0;
    const hipError_t err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, pt_path_kernel<WIDE, IN_LDS, STACK_LDS>, 256, lds_bytes);
    *out = (err != hipSuccess || blocks < 1) ? 1 : blocks;
}

template<int STACK_LDS, bool IN_LDS>
void launch_closest(hipStream_t stream, const PtDevScene &scene, const float *rays6, uint32_t n, uint2 *out, const PtPathConfig &cfg) {
    const size_t lds = (size_t)STACK_LDS * 256 * sizeof(uint2) + (IN_LDS ? ((size_t)scene.n_lds_pairs + scene.pair_base) * 64 : 0);
    hipLaunchKernelGGL((pt_closest_kernel<STACK_LDS, IN_LDS>), dim3((n + 255) / 256), dim3(256), lds, stream, scene, rays6, n, out, cfg.spill, cfg.spill_depth);
}

} // namespace

// the instantiations of the path kernel: slot word (compact | wide) x records (HBM | LDS) x stack window (8 entries; 4 for scenes in LDS that need the room)
#define PT_DISPATCH_PATH(fn, cfg, ...)                                              \
    do {                                                                            \
        if((cfg).in_lds && (cfg).stack_lds == PT_PATH_STACK_LDS_SMALL) {            \
            if((cfg).wide) {                                                        \
                fn<true, true, PT_PATH_STACK_LDS_SMALL>(__VA_ARGS__);               \
            }                                                                       \
            else {                                                                  \
                fn<false, true, PT_PATH_STACK_LDS_SMALL>(__VA_ARGS__);              \
            }                                                                       \
        }                                                                           \
        else if((cfg).in_lds) {                                                     \
            if((cfg).wide) {                                                        \
                fn<true, true, PT_PATH_STACK_LDS>(__VA_ARGS__);                     \
            }                                                                       \
            else {                                                                  \
                fn<false, true, PT_PATH_STACK_LDS>(__VA_ARGS__);                    \
            }                                                                       \
        }                                                                           \
        else {                                                                      \
            if((cfg).wide) {                                                        \
                fn<true, false, PT_PATH_STACK_LDS>(__VA_ARGS__);                    \
            }                                                                       \
            else {                                                                  \
                fn<false, false, PT_PATH_STACK_LDS>(__VA_ARGS__);                   \
            }                                                                       \
        }                                                                           \
    } while(0)

void pt_launch_path(hipStream_t stream, const PtDevScene &scene, const PtDevCamera &camera, const PtDevOptions &options, PtSlots slots, PtStreams streams,
                    PtLocalQueue queue, const PtPathConfig &cfg, float4 *image, PtDevCounters *counters, PtPathArgs *host_args, PtPathArgs *d_args) {
    PtPathArgs &a = *host_args;
    a.sc = scene;
    a.cam = camera;
    a.opt = options;
    a.S = slots;
    a.T = streams;
    a.Q = queue;
    a.rows = cfg.rows;
    a.slots_per_wave = cfg.slots_per_wave;
    a.refill_idle = cfg.refill_idle;
    a.min_ready = cfg.min_ready;
    a.burst_steps = cfg.burst_steps;
    a.leaf_min = cfg.leaf_min;
    a.ready_shift = cfg.ready_shift;
    a.pass_q_low = cfg.pass_q_low;
    a.early_ready = cfg.early_ready;
    a.compact_passes = cfg.compact_passes;
    a.debug_lanes = cfg.debug_lanes;
    a.spill = cfg.spill;
    a.spill_depth = cfg.spill_depth;
    a.save_stride = (uint32_t)cfg.grid * 256u;
    a.walk_save = cfg.walk_save;
    a.image = image;
    a.counters = counters;
    a.wave_counters = cfg.wave_counters;
    (void)hipMemcpyAsync(d_args, host_args, sizeof(PtPathArgs), hipMemcpyHostToDevice, stream);
    PT_DISPATCH_PATH(launch_path, cfg, stream, cfg, d_args);
}

void pt_launch_steptime(hipStream_t stream, const PtDevScene &scene, const float *rays6, uint32_t n, uint32_t lanes_per_wave, uint4 *out, uint2 *spill, uint32_t spill_depth, int flags) {
    const uint32_t waves = (n + lanes_per_wave - 1) / lanes_per_wave;
    if(flags & 2) { // bit 1: stamp the waits (each stamp is a scalar memory round trip of its own: the totals of such a run are inflated)
        hipLaunchKernelGGL((pt_steptime_kernel<8, true>), dim3((waves + 3) / 4), dim3(256), (size_t)8 * 256 * sizeof(uint2), stream, scene, rays6, n, lanes_per_wave, out, spill, spill_depth);
    }
    else {
        hipLaunchKernelGGL((pt_steptime_kernel<8, false>), dim3((waves + 3) / 4), dim3(256), (size_t)8 * 256 * sizeof(uint2), stream, scene, rays6, n, lanes_per_wave, out, spill, spill_depth);
    }
}

void pt_launch_closest(hipStream_t stream, const PtDevScene &scene, const float *rays6, uint32_t n, uint2 *out, const PtPathConfig &cfg) {
    if(n == 0) {
        return;
    }
    if(cfg.in_lds && cfg.stack_lds == PT_PATH_STACK_LDS_SMALL) {
        launch_closest<PT_PATH_STACK_LDS_SMALL, true>(stream, scene, rays6, n, out, cfg);
    }
    else if(cfg.in_lds) {
        launch_closest<PT_PATH_STACK_LDS, true>(stream, scene, rays6, n, out, cfg);
    }
    else {
        launch_closest<PT_PATH_STACK_LDS, false>(stream, scene, rays6, n, out, cfg);
    }
}

int pt_launch_replay(hipStream_t stream, const PtDevScene &scene, const PtLocalQueue &Q, uint32_t n_logs, uint32_t parts, int waves_per_simd, const PtPathConfig &cfg,
                     uint2 *spill, unsigned long long *out) {
    switch(waves_per_simd) {
    case 4: return launch_replay<8, 4>(stream, scene, Q, n_logs, parts, cfg, spill, out);
    case 5: return launch_replay<8, 5>(stream, scene, Q, n_logs, parts, cfg, spill, out);
    case 6: return launch_replay<8, 6>(stream, scene, Q, n_logs, parts, cfg, spill, out);
    case 7: return launch_replay<8, 7>(stream, scene, Q, n_logs, parts, cfg, spill, out);
    default: return launch_replay<8, 8>(stream, scene, Q, n_logs, parts, cfg, spill, out);
    }
}

size_t pt_path_lds_bytes(int wide, int rows, int stack_lds, uint32_t n_lds_pairs, uint32_t n_lds_leaf_records) {
    const size_t scene = ((size_t)n_lds_pairs + n_lds_leaf_records) * 64;
    return (size_t)stack_lds * 256 * sizeof(uint2) + (size_t)4 * rows * 64 * (sizeof(uint2) + (wide ? sizeof(unsigned long long) : sizeof(uint32_t))) + PT_LDS_TABLE_BYTES + PT_COST_LDS_BYTES + scene;
}

int pt_path_blocks_per_cu(const PtPathConfig &cfg) {
    int blocks = 1;
    PT_DISPATCH_PATH(occupancy, cfg, cfg.lds_bytes, &blocks);
    return blocks;
}

int pt_path_stack_lds(int in_lds, size_t lds_bytes_with_default_window) {
    // (four workgroups of the path kernel share the 160 KB of a CU when each stays within 40 KB)
    return (in_lds && lds_bytes_with_default_window > 40960) ? PT_PATH_STACK_LDS_SMALL : PT_PATH_STACK_LDS;
}
