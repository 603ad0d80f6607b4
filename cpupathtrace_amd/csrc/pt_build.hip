// pt_build.hip -- scene construction on the device: leaf records and the REFERENCE-SHAPED BVH, built in HBM.
//
// What is built is exactly the tree of impl::constructBVH (src/scene/scene.cpp:12-102) -- closest-hit ties and pruning order depend
// on its shape, so a faster heuristic (LBVH, binned SAH) is not an option -- but level by level and for all nodes of a level at
// once instead of by a serial recursion.  Per node the reference does (pt_bvh.cpp has the host restatement):
//   1. per axis, the median of the boxes' LOW coordinates = element n/2 - 1 of the sorted coordinates          (scene.cpp:24-36)
//   2. per axis, the summed surface area of the boxes of the groups {low <= median} and {low > median}           (:38-62)
//   3. the axis with the smallest sum, the first one on ties                                                    (:64-72)
//   4. a STABLE partition of the node's objects in input order                                                  (:74-87)
//   5. while left has more than one and more than twice right's objects: move left's last object to right's end (:89-94)
// All five are functions of VALUES and of the input order only, so they map onto data-parallel primitives:
//   * the objects of every node are a contiguous range of four arrays of object indices: `ord` (input order) and `srt[k]` (sorted
//     by the low coordinate on axis k; three radix sorts once, up front).  A stable partition of a sorted list stays sorted, so the
//     median of step 1 is ONE LOOKUP per node and axis;
//   * step 2: ranges of up to 128 objects are reduced by one thread; longer ones by all their objects in parallel through
//     order-preserving integer keys and atomic min/max (wave-aggregated when a wavefront lies inside one range: the top levels);
//   * steps 4 and 5 are one exclusive scan of the "goes left" flags over `ord` (new position = range start + rank, the moved tail
//     of step 5 is appended to the right range in reverse order) and one 3-wide scan for the three sorted lists;
//   * the children that are inner nodes get their pair-record slots by a scan over the level's nodes whose operator carries the
//     parity of the running slot index (two inner siblings share one aligned 128-byte line, pt_bvh.cpp flatten_breadth_first),
//     which makes the result bit-identical to the host path's array (tests/test_gpu_parity.py::test_device_build_*).
// The levels come out in breadth-first order, which IS the layout the traversal kernel wants; boxes are filled bottom-up afterwards
// (child boxes are unions in left-then-right operand order, bounding_box.cpp:8-10,18-24).  The final `ord` array lists the leaves
// depth-first, left to right: the order Scene::registerEmissiveObjects visits them in (scene.cpp:183-208).
#include "pt_build.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <utility>
#include <vector>

#include "pt_types.h"

namespace {

constexpr uint32_t NONE = 0xffffffffu;
constexpr uint32_t SMALL = 128; // ranges up to this many objects are reduced by a single thread

__device__ __forceinline__ float fmin_std(float a, float b) {
    return (b < a) ? b : a;
}
__device__ __forceinline__ float fmax_std(float a, float b) {
    return (a < b) ? b : a;
}

// order-preserving float <-> uint32 (for atomicMin / atomicMax)
__device__ __forceinline__ uint32_t fkey(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
constexpr uint32_t KEY_PINF = 0xff800000u; // fkey(+inf)
constexpr uint32_t KEY_NINF = 0x007fffffu; // fkey(-inf)

struct LeafArrays {
    float *lo[3];
    float *hi[3];
    uint32_t *ref;
};

struct Lists {
    uint32_t *ord;
    uint32_t *srt[3];
};

struct SegTable {
    uint32_t *start;
    uint32_t *count;
    uint32_t *slot;
    uint32_t *acc; // index of the range's accumulator block (ranges longer than SMALL) or NONE
};

// scan element of the per-level child numbering: cnt = inner children so far; a0 / a1 = pair slots appended so far when the slot
// index at the start of the level is even / odd
struct Cn {
    uint32_t cnt, a0, a1;
};
struct CnOp {
    __host__ __device__ Cn operator()(const Cn &f, const Cn &g) const {
        Cn h;
        h.cnt = f.cnt + g.cnt;
        h.a0 = f.a0 + ((f.a0 & 1u) ? g.a1 : g.a0);
        h.a1 = f.a1 + (((1u + f.a1) & 1u) ? g.a1 : g.a0);
        return h;
    }
};
struct U3 {
    uint32_t x, y, z;
};
struct U3Op {
    __host__ __device__ U3 operator()(const U3 &a, const U3 &b) const {
        return U3{a.x + b.x, a.y + b.y, a.z + b.z};
    }
};

// ---- leaf records ---------------------------------------------------------------------------------------------------------------

// Triangle::Triangle / getBoundingVolume (object.cpp:118-124,184-186) and the records of pt_types.h
__global__ void k_triangle_records(uint32_t n, const float *__restrict__ pos, const float *__restrict__ nrm, const uint8_t *__restrict__ cull,
                                   const uint32_t *__restrict__ material, const uint32_t *__restrict__ obj, float4 *__restrict__ tris,
                                   float4 *__restrict__ shade, LeafArrays leaf) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= n) {
        return;
    }
    const float *p = pos + 9 * static_cast<size_t>(t);
    float v[9];
    for(int k = 0; k < 9; k++) {
        v[k] = p[k];
    }
    const float abx = v[3] - v[0], aby = v[4] - v[1], abz = v[5] - v[2];
    const float acx = v[6] - v[0], acy = v[7] - v[1], acz = v[8] - v[2];
    const uint32_t o = obj[t];
    const uint32_t obj_cull = o | (cull[t] != 0 ? 0x80000000u : 0u);
    const float4 r0 = make_float4(v[0], v[1], v[2], abx);
    const float4 r1 = make_float4(aby, abz, acx, acy);
    const float4 r2 = make_float4(acz, __uint_as_float(material[t]), __uint_as_float(obj_cull), 0.0f);
    float na[3], nb[3], nc[3];
    if(nrm != nullptr) {
        const float *q = nrm + 9 * static_cast<size_t>(t);
        for(int k = 0; k < 3; k++) {
            na[k] = q[k];
            nb[k] = q[3 + k];
            nc[k] = q[6 + k];
        }
    }
    else {
        // normalize(cross(b - a, c - a)): dot accumulates from 0, normalize multiplies by the reciprocal length (vector.h:152)
        const float cx = aby * acz - abz * acy, cy = abz * acx - abx * acz, cz = abx * acy - aby * acx;
        float d = 0.0f;
        d += cx * cx;
        d += cy * cy;
        d += cz * cz;
        const float inv = 1.0f / __builtin_sqrtf(d);
        na[0] = nb[0] = nc[0] = cx * inv;
        na[1] = nb[1] = nc[1] = cy * inv;
        na[2] = nb[2] = nc[2] = cz * inv;
    }
    float4 *tr = tris + PT_TRI_QUADS * static_cast<size_t>(t);
    tr[0] = r0;
    tr[1] = r1;
    tr[2] = r2;
    tr[3] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float4 *sh = shade + 8 * static_cast<size_t>(t);
    sh[0] = r0;
    sh[1] = r1;
    sh[2] = r2;
    sh[3] = make_float4(na[0], na[1], na[2], nb[0]);
    sh[4] = make_float4(nb[1], nb[2], nc[0], nc[1]);
    sh[5] = make_float4(nc[2], 0.0f, 0.0f, 0.0f);
    sh[6] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    sh[7] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for(int k = 0; k < 3; k++) {
        leaf.lo[k][o] = fmin_std(fmin_std(v[k], v[3 + k]), v[6 + k]);
        leaf.hi[k][o] = fmax_std(fmax_std(v[k], v[3 + k]), v[6 + k]);
    }
    leaf.ref[o] = PT_REF_LEAF | t;
}

// Sphere::getBoundingVolume (object.cpp:90-93)
__global__ void k_sphere_records(uint32_t n, const float *__restrict__ sph, const uint32_t *__restrict__ material, const uint32_t *__restrict__ obj,
                                 float4 *__restrict__ spheres, uint2 *__restrict__ meta, float4 *__restrict__ walk_records, LeafArrays leaf) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n) {
        return;
    }
    const float *sp = sph + 4 * static_cast<size_t>(i);
    const float c[3] = {sp[0], sp[1], sp[2]};
    const float r = sp[3];
    const uint32_t o = obj[i];
    spheres[i] = make_float4(c[0], c[1], c[2], r);
    // the record the traversal fetches: behind the triangle records and their spare one (pt_types.h)
    float4 *wr = walk_records + PT_TRI_QUADS * static_cast<size_t>(i);
    wr[0] = make_float4(c[0], c[1], c[2], r);
    wr[1] = wr[2] = wr[3] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    meta[i] = make_uint2(material[i], o);
    for(int k = 0; k < 3; k++) {
        leaf.lo[k][o] = c[k] - r;
        leaf.hi[k][o] = c[k] + r;
    }
    leaf.ref[o] = PT_REF_LEAF | PT_REF_SPHERE | i;
}

__global__ void k_iota(uint32_t n, uint32_t *a, uint32_t *segid) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n) {
        a[i] = i;
        segid[i] = 0;
    }
}

__global__ void k_first_segment(uint32_t n, SegTable t) {
    t.start[0] = 0;
    t.count[0] = n;
    t.slot[0] = 0;
    t.acc[0] = n > SMALL ? 0u : NONE;
}

// ---- one level ----------------------------------------------------------------------------------------------------------------

// step 1: the three medians of every range; ranges reduced in parallel get their accumulators reset
__global__ void k_median(uint32_t n_seg, SegTable t, Lists cur, LeafArrays leaf, float *__restrict__ med3, uint32_t *__restrict__ accbuf) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if(s >= n_seg) {
        return;
    }
    const uint32_t m = t.start[s] + t.count[s] / 2 - 1;
    for(int k = 0; k < 3; k++) {
        med3[3 * static_cast<size_t>(s) + k] = leaf.lo[k][cur.srt[k][m]];
    }
    const uint32_t a = t.acc[s];
    if(a != NONE) {
        uint32_t *A = accbuf + 36 * static_cast<size_t>(a);
        for(int kg = 0; kg < 6; kg++) {
            for(int c = 0; c < 3; c++) {
                A[kg * 6 + c] = KEY_PINF;
                A[kg * 6 + 3 + c] = KEY_NINF;
            }
        }
    }
}

// step 2 for long ranges: every object adds its box to the two group boxes of each axis.  A wavefront walks ACC_STEPS x 64
// consecutive positions; while they belong to one range every lane keeps private minima/maxima in registers, and only when the
// range changes (or the walk ends) are the 36 values reduced across the wave and merged with one atomic each.  On the top levels
// that is 36 atomics per 2048 objects instead of 36 per 64 -- the same few addresses are the bottleneck there.
constexpr int ACC_STEPS = 32;

__device__ __forceinline__ void acc_flush(uint32_t *__restrict__ A, uint32_t (&kmin)[18], uint32_t (&kmax)[18], int lane) {
    for(int v = 0; v < 18; v++) {
        uint32_t a = kmin[v], b = kmax[v];
        for(int off = 32; off > 0; off >>= 1) {
            const uint32_t o1 = __shfl_xor(a, off), o2 = __shfl_xor(b, off);
            a = o1 < a ? o1 : a;
            b = o2 > b ? o2 : b;
        }
        // v = (axis * 2 + group) * 3 + coordinate; accumulator layout: [axis][group][lo xyz, hi xyz]
        const int kg = v / 3, c = v % 3;
        if(lane == 0) {
            if(a != KEY_PINF) {
                atomicMin(&A[kg * 6 + c], a);
            }
            if(b != KEY_NINF) {
                atomicMax(&A[kg * 6 + 3 + c], b);
            }
        }
        kmin[v] = KEY_PINF;
        kmax[v] = KEY_NINF;
    }
}

__global__ __launch_bounds__(256) void k_accumulate(uint32_t n, const uint32_t *__restrict__ segid, SegTable t, Lists cur, LeafArrays leaf,
                                                     const float *__restrict__ med3, uint32_t *__restrict__ accbuf) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t first = wave * (64u * ACC_STEPS);
    uint32_t kmin[18], kmax[18];
    for(int v = 0; v < 18; v++) {
        kmin[v] = KEY_PINF;
        kmax[v] = KEY_NINF;
    }
    uint32_t held = NONE; // accumulator block the private values belong to (wave-uniform)
    for(int step = 0; step < ACC_STEPS; step++) {
        const uint32_t i = first + 64u * static_cast<uint32_t>(step) + static_cast<uint32_t>(lane);
        const uint32_t seg = i < n ? segid[i] : NONE;
        const uint32_t a = seg != NONE ? t.acc[seg] : NONE;
        const bool active = a != NONE;
        const unsigned long long act = __ballot(active);
        if(act == 0ULL) {
            continue;
        }
        float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}, m[3] = {0, 0, 0};
        if(active) {
            const uint32_t id = cur.ord[i];
            for(int k = 0; k < 3; k++) {
                lo[k] = leaf.lo[k][id];
                hi[k] = leaf.hi[k][id];
                m[k] = med3[3 * static_cast<size_t>(seg) + k];
            }
        }
        const int head = __ffsll(static_cast<long long>(act)) - 1;
        const uint32_t a_head = __shfl(a, head);
        const bool uniform = __ballot(active && a != a_head) == 0ULL;
        if(uniform) {
            if(held != a_head) {
                if(held != NONE) {
                    acc_flush(accbuf + 36 * static_cast<size_t>(held), kmin, kmax, lane);
                }
                held = a_head;
            }
            if(active) {
                for(int k = 0; k < 3; k++) {
                    const int g = lo[k] <= m[k] ? 0 : 1;
                    for(int c = 0; c < 3; c++) {
                        const uint32_t l = fkey(lo[c]), h = fkey(hi[c]);
                        for(int gg = 0; gg < 2; gg++) { // both groups spelled out: no dynamic register indexing
                            const int v = (k * 2 + gg) * 3 + c;
                            const bool in = g == gg;
                            kmin[v] = (in && l < kmin[v]) ? l : kmin[v];
                            kmax[v] = (in && h > kmax[v]) ? h : kmax[v];
                        }
                    }
                }
            }
        }
        else if(active) {
            // a range boundary inside the wavefront: every lane merges its own box
            uint32_t *A = accbuf + 36 * static_cast<size_t>(a);
            for(int k = 0; k < 3; k++) {
                const int g = lo[k] <= m[k] ? 0 : 1;
                for(int c = 0; c < 3; c++) {
                    atomicMin(&A[(k * 2 + g) * 6 + c], fkey(lo[c]));
                    atomicMax(&A[(k * 2 + g) * 6 + 3 + c], fkey(hi[c]));
                }
            }
        }
    }
    if(held != NONE) {
        acc_flush(accbuf + 36 * static_cast<size_t>(held), kmin, kmax, lane);
    }
}

// steps 2 (short ranges) and 3: surface areas and the split axis
__global__ __launch_bounds__(256) void k_choose(uint32_t n_seg, SegTable t, Lists cur, LeafArrays leaf, const float *__restrict__ med3,
                                                 const uint32_t *__restrict__ accbuf, uint32_t *__restrict__ axis_out, float *__restrict__ med_out) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if(s >= n_seg) {
        return;
    }
    const float inf = __builtin_huge_valf();
    float clo[3][2][3], chi[3][2][3];
    float m[3];
    for(int k = 0; k < 3; k++) {
        m[k] = med3[3 * static_cast<size_t>(s) + k];
    }
    const uint32_t a = t.acc[s];
    if(a == NONE) {
        for(int k = 0; k < 3; k++) {
            for(int g = 0; g < 2; g++) {
                for(int c = 0; c < 3; c++) {
                    clo[k][g][c] = inf;
                    chi[k][g][c] = -inf;
                }
            }
        }
        const uint32_t start = t.start[s], n = t.count[s];
        for(uint32_t j = 0; j < n; j++) {
            const uint32_t id = cur.ord[start + j];
            float lo[3], hi[3];
            for(int c = 0; c < 3; c++) {
                lo[c] = leaf.lo[c][id];
                hi[c] = leaf.hi[c][id];
            }
            for(int k = 0; k < 3; k++) {
                const bool left = lo[k] <= m[k];
                for(int c = 0; c < 3; c++) {
                    const float l0 = fmin_std(clo[k][0][c], lo[c]), l1 = fmin_std(clo[k][1][c], lo[c]);
                    const float h0 = fmax_std(chi[k][0][c], hi[c]), h1 = fmax_std(chi[k][1][c], hi[c]);
                    clo[k][0][c] = left ? l0 : clo[k][0][c];
                    clo[k][1][c] = left ? clo[k][1][c] : l1;
                    chi[k][0][c] = left ? h0 : chi[k][0][c];
                    chi[k][1][c] = left ? chi[k][1][c] : h1;
                }
            }
        }
    }
    else {
        const uint32_t *A = accbuf + 36 * static_cast<size_t>(a);
        for(int k = 0; k < 3; k++) {
            for(int g = 0; g < 2; g++) {
                for(int c = 0; c < 3; c++) {
                    clo[k][g][c] = fkey_inv(A[(k * 2 + g) * 6 + c]);
                    chi[k][g][c] = fkey_inv(A[(k * 2 + g) * 6 + 3 + c]);
                }
            }
        }
    }
    int axis = 0;
    float best = 0.0f;
    for(int k = 0; k < 3; k++) {
        float surface_area = 0.0f;
        for(int g = 0; g < 2; g++) {
            const float d0 = chi[k][g][0] - clo[k][g][0];
            const float d1 = chi[k][g][1] - clo[k][g][1];
            const float d2 = chi[k][g][2] - clo[k][g][2];
            surface_area += 2.0f * (d0 * d1 + d1 * d2 + d0 * d2);
        }
        if(k == 0 || surface_area < best) {
            best = surface_area;
            axis = k;
        }
    }
    axis_out[s] = static_cast<uint32_t>(axis);
    med_out[s] = m[axis];
}

// step 4, part 1: "goes left" flags in input order (entry n is the scan's sentinel)
__global__ void k_flag(uint32_t n, const uint32_t *__restrict__ segid, Lists cur, LeafArrays leaf, const uint32_t *__restrict__ axis, const float *__restrict__ med,
                       uint32_t *__restrict__ flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i > n) {
        return;
    }
    uint32_t f = 0;
    if(i < n) {
        const uint32_t seg = segid[i];
        if(seg != NONE) {
            const uint32_t id = cur.ord[i];
            const uint32_t ax = axis[seg];
            const float lo = ax == 0 ? leaf.lo[0][id] : (ax == 1 ? leaf.lo[1][id] : leaf.lo[2][id]);
            f = lo <= med[seg] ? 1u : 0u;
        }
    }
    flag[i] = f;
}

// steps 4 and 5 per node: sizes of the two children, and how many inner children it has (input of the numbering scan)
__global__ void k_split(uint32_t n_seg, SegTable t, const uint32_t *__restrict__ flag_ex, int align, uint32_t *__restrict__ nl_final, uint32_t *__restrict__ nl_orig,
                        Cn *__restrict__ child) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if(s >= n_seg) {
        return;
    }
    const uint32_t start = t.start[s], n = t.count[s];
    const uint32_t nl = flag_ex[start + n] - flag_ex[start];
    const uint32_t nr = n - nl;
    uint32_t moved = 0;
    if(nl > 1 && nl > 2 * nr) {
        // while(left > 1 && left > 2 * right) { left--; right++; }: left - 2 * right drops by 3 per move
        moved = (nl - 2 * nr + 2) / 3;
        if(moved > nl - 1) {
            moved = nl - 1;
        }
    }
    const uint32_t nlf = nl - moved;
    nl_final[s] = nlf;
    nl_orig[s] = nl;
    const uint32_t inner = (nlf >= 2 ? 1u : 0u) + (n - nlf >= 2 ? 1u : 0u);
    Cn c;
    c.cnt = inner;
    c.a0 = inner;
    c.a1 = (inner == 2 && align != 0) ? 3u : inner;
    child[s] = c;
}

// step 4, part 2: the stable partition of `ord`, with step 5's moved tail appended to the right child in reverse
__global__ void k_scatter_ord(uint32_t n, const uint32_t *__restrict__ segid, SegTable t, const uint32_t *__restrict__ flag, const uint32_t *__restrict__ flag_ex,
                              const uint32_t *__restrict__ nl_final, const uint32_t *__restrict__ nl_orig, const uint32_t *__restrict__ ord_in,
                              uint32_t *__restrict__ ord_out, uint8_t *__restrict__ side) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n) {
        return;
    }
    const uint32_t id = ord_in[i];
    const uint32_t seg = segid[i];
    if(seg == NONE) {
        ord_out[i] = id;
        return;
    }
    const uint32_t start = t.start[seg], count = t.count[seg];
    const uint32_t nlf = nl_final[seg], nlo = nl_orig[seg];
    const uint32_t r = flag_ex[i] - flag_ex[start];
    uint32_t pos;
    uint8_t left;
    if(flag[i] != 0) {
        if(r < nlf) {
            pos = start + r;
            left = 1;
        }
        else {
            pos = start + nlf + (count - nlo) + (nlo - 1 - r);
            left = 0;
        }
    }
    else {
        pos = start + nlf + ((i - start) - r);
        left = 0;
    }
    ord_out[pos] = id;
    side[id] = left;
}

__global__ void k_flag3(uint32_t n, const uint32_t *__restrict__ segid, Lists cur, const uint8_t *__restrict__ side, U3 *__restrict__ flag3) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i > n) {
        return;
    }
    U3 f{0, 0, 0};
    if(i < n && segid[i] != NONE) {
        f.x = side[cur.srt[0][i]];
        f.y = side[cur.srt[1][i]];
        f.z = side[cur.srt[2][i]];
    }
    flag3[i] = f;
}

// the same partition applied to the three sorted lists (they stay sorted within the children)
__global__ void k_scatter_srt(uint32_t n, const uint32_t *__restrict__ segid, SegTable t, const U3 *__restrict__ flag3, const U3 *__restrict__ flag3_ex,
                              const uint32_t *__restrict__ nl_final, Lists cur, Lists next) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n) {
        return;
    }
    const uint32_t seg = segid[i];
    const uint32_t id0 = cur.srt[0][i], id1 = cur.srt[1][i], id2 = cur.srt[2][i];
    if(seg == NONE) {
        next.srt[0][i] = id0;
        next.srt[1][i] = id1;
        next.srt[2][i] = id2;
        return;
    }
    const uint32_t start = t.start[seg];
    const uint32_t nlf = nl_final[seg];
    const U3 f = flag3[i], e = flag3_ex[i], e0 = flag3_ex[start];
    const uint32_t off = i - start;
    const uint32_t r0 = e.x - e0.x, r1 = e.y - e0.y, r2 = e.z - e0.z;
    next.srt[0][f.x ? start + r0 : start + nlf + (off - r0)] = id0;
    next.srt[1][f.y ? start + r1 : start + nlf + (off - r1)] = id1;
    next.srt[2][f.z ? start + r2 : start + nlf + (off - r2)] = id2;
}

// Numbers the inner children (next level's range table, breadth-first pair slots with sibling alignment), writes the node's two
// child references and the boxes of leaf children into its pair record.
__global__ void k_children(uint32_t n_seg, SegTable t, const Cn *__restrict__ child, const Cn *__restrict__ child_ex, uint32_t total_slots, int align,
                           const uint32_t *__restrict__ nl_final, const uint32_t *__restrict__ ord_new, LeafArrays leaf, SegTable next, float *__restrict__ pairs,
                           uint32_t *__restrict__ parent, uint32_t *__restrict__ n_large, uint32_t *__restrict__ totals) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if(s >= n_seg) {
        return;
    }
    const uint32_t start = t.start[s], n = t.count[s], slot = t.slot[s];
    const uint32_t nlf = nl_final[s];
    const Cn e = child_ex[s];
    const uint32_t odd = total_slots & 1u;
    uint32_t base = total_slots + (odd ? e.a1 : e.a0);
    const bool left_inner = nlf >= 2, right_inner = n - nlf >= 2;
    if(left_inner && right_inner && align != 0 && (base & 1u) != 0) {
        base++; // an unused slot, so that the two siblings share one aligned 128-byte line
    }
    uint32_t idx = e.cnt;
    float *q = pairs + 16 * static_cast<size_t>(slot);
    uint32_t refs[2];
    for(int c = 0; c < 2; c++) {
        const uint32_t c_start = c == 0 ? start : start + nlf;
        const uint32_t c_count = c == 0 ? nlf : n - nlf;
        if(c_count >= 2) {
            next.start[idx] = c_start;
            next.count[idx] = c_count;
            next.slot[idx] = base;
            next.acc[idx] = c_count > SMALL ? atomicAdd(n_large, 1u) : NONE;
            parent[base] = slot * 2 + static_cast<uint32_t>(c);
            refs[c] = base;
            idx++;
            base++;
        }
        else {
            const uint32_t id = ord_new[c_start];
            for(int k = 0; k < 3; k++) {
                q[6 * c + k] = leaf.lo[k][id];
                q[6 * c + 3 + k] = leaf.hi[k][id];
            }
            refs[c] = leaf.ref[id];
        }
    }
    q[12] = __uint_as_float(refs[0]);
    q[13] = __uint_as_float(refs[1]);
    q[14] = 0.0f;
    q[15] = 0.0f;
    if(s == n_seg - 1) {
        const Cn incl = CnOp()(e, child[s]);
        totals[0] = incl.cnt;
        totals[1] = odd ? incl.a1 : incl.a0;
    }
}

__global__ void k_segid(uint32_t n, uint32_t *__restrict__ segid, SegTable t, const uint32_t *__restrict__ nl_final, const Cn *__restrict__ child_ex) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n) {
        return;
    }
    const uint32_t seg = segid[i];
    if(seg == NONE) {
        return;
    }
    const uint32_t start = t.start[seg], count = t.count[seg], nlf = nl_final[seg];
    const bool left = i < start + nlf;
    const uint32_t c_count = left ? nlf : count - nlf;
    uint32_t out = NONE;
    if(c_count >= 2) {
        out = child_ex[seg].cnt + ((!left && nlf >= 2) ? 1u : 0u);
    }
    segid[i] = out;
}

// bottom-up: the box of an inner node is the union of its two child boxes (left operand first); it goes into the parent's record
__global__ void k_union(uint32_t begin, uint32_t end, float *__restrict__ pairs, const uint32_t *__restrict__ parent, float *__restrict__ root_box) {
    const uint32_t slot = begin + blockIdx.x * blockDim.x + threadIdx.x;
    if(slot >= end) {
        return;
    }
    const float *q = pairs + 16 * static_cast<size_t>(slot);
    if(__float_as_uint(q[12]) == 0u && __float_as_uint(q[13]) == 0u) {
        return; // alignment slot
    }
    float box[6];
    for(int k = 0; k < 3; k++) {
        box[k] = fmin_std(q[k], q[6 + k]);
        box[3 + k] = fmax_std(q[3 + k], q[9 + k]);
    }
    float *dst = slot == 0 ? root_box : pairs + 16 * static_cast<size_t>(parent[slot] >> 1) + 6 * (parent[slot] & 1u);
    for(int k = 0; k < 6; k++) {
        dst[k] = box[k];
    }
}

__global__ void k_select(uint32_t n, const uint32_t *__restrict__ dfs, const uint32_t *__restrict__ mask, uint2 *__restrict__ out, uint32_t *__restrict__ count,
                         uint32_t capacity) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n) {
        return;
    }
    const uint32_t obj = dfs[i];
    if((mask[obj >> 5] >> (obj & 31u)) & 1u) {
        const uint32_t k = atomicAdd(count, 1u);
        if(k < capacity) {
            out[k] = make_uint2(i, obj);
        }
    }
}


struct Scratch {
    std::vector<void *> ptrs;
    ~Scratch() {
        for(void *p : ptrs) {
            (void)hipFree(p);
        }
    }
    template<typename T>
    hipError_t get(T **out, size_t count) {
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
        if(e == hipSuccess) {
            ptrs.push_back(p);
            *out = static_cast<T *>(p);
        }
        return e;
    }
    void keep(void *p) { // ownership leaves the scratch set
        ptrs.erase(std::remove(ptrs.begin(), ptrs.end(), p), ptrs.end());
    }
};

inline dim3 grid_for(size_t n) {
    return dim3(static_cast<unsigned>((n + 255) / 256));
}

} // namespace

#define PTB_TRY(call, what)              \
    do {                                 \
        hipError_t e_ = (call);          \
        if(e_ != hipSuccess) {           \
            if(error_text != nullptr) {  \
                *error_text = what;      \
            }                            \
            return e_;                   \
        }                                \
    } while(0)

hipError_t pt_build_scene_device(hipStream_t stream, const PtBuildInput &in, PtBuildOutput &out, const char **error_text) {
    const uint32_t n = in.n_objects;
    Scratch scratch;

    LeafArrays leaf;
    for(int k = 0; k < 3; k++) {
        PTB_TRY(scratch.get(&leaf.lo[k], n), "leaf boxes");
        PTB_TRY(scratch.get(&leaf.hi[k], n), "leaf boxes");
    }
    PTB_TRY(scratch.get(&leaf.ref, n), "leaf references");

    if(in.n_triangles > 0) {
        hipLaunchKernelGGL(k_triangle_records, grid_for(in.n_triangles), dim3(256), 0, stream, in.n_triangles, in.tri_pos, in.tri_nrm, in.tri_cull, in.tri_material,
                           in.tri_obj, out.tris, out.tri_shade, leaf);
    }
    if(in.n_spheres > 0) {
        hipLaunchKernelGGL(k_sphere_records, grid_for(in.n_spheres), dim3(256), 0, stream, in.n_spheres, in.sph, in.sph_material, in.sph_obj, out.spheres, out.sph_meta,
                           out.tris + PT_TRI_QUADS * (static_cast<size_t>(in.n_triangles) + 1), leaf);
    }
    PTB_TRY(hipGetLastError(), "leaf record kernels");

    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    PTB_TRY(hipEventCreate(&ev0), "event");
    PTB_TRY(hipEventCreate(&ev1), "event");
    struct EventGuard {
        hipEvent_t a, b;
        ~EventGuard() {
            (void)hipEventDestroy(a);
            (void)hipEventDestroy(b);
        }
    } guard{ev0, ev1};
    PTB_TRY(hipEventRecord(ev0, stream), "event");

    // ---- lists ---------------------------------------------------------------------------------------------------------------
    Lists lists[2];
    for(int b = 0; b < 2; b++) {
        PTB_TRY(scratch.get(&lists[b].ord, n), "object lists");
        for(int k = 0; k < 3; k++) {
            PTB_TRY(scratch.get(&lists[b].srt[k], n), "object lists");
        }
    }
    uint32_t *segid = nullptr, *flag = nullptr, *flag_ex = nullptr;
    uint8_t *side = nullptr;
    U3 *flag3 = nullptr, *flag3_ex = nullptr;
    PTB_TRY(scratch.get(&segid, n), "range ids");
    PTB_TRY(scratch.get(&side, n), "sides");
    PTB_TRY(scratch.get(&flag, static_cast<size_t>(n) + 1), "flags");
    PTB_TRY(scratch.get(&flag_ex, static_cast<size_t>(n) + 1), "flags");
    PTB_TRY(scratch.get(&flag3, static_cast<size_t>(n) + 1), "flags");
    PTB_TRY(scratch.get(&flag3_ex, static_cast<size_t>(n) + 1), "flags");

    const size_t seg_cap = static_cast<size_t>(n) / 2 + 2;
    SegTable tables[2];
    for(int b = 0; b < 2; b++) {
        PTB_TRY(scratch.get(&tables[b].start, seg_cap), "range table");
        PTB_TRY(scratch.get(&tables[b].count, seg_cap), "range table");
        PTB_TRY(scratch.get(&tables[b].slot, seg_cap), "range table");
        PTB_TRY(scratch.get(&tables[b].acc, seg_cap), "range table");
    }
    float *med3 = nullptr, *med = nullptr;
    uint32_t *axis = nullptr, *nl_final = nullptr, *nl_orig = nullptr, *accbuf = nullptr, *n_large = nullptr;
    Cn *child = nullptr, *child_ex = nullptr;
    PTB_TRY(scratch.get(&med3, 3 * seg_cap), "medians");
    PTB_TRY(scratch.get(&med, seg_cap), "medians");
    PTB_TRY(scratch.get(&axis, seg_cap), "axes");
    PTB_TRY(scratch.get(&nl_final, seg_cap), "child sizes");
    PTB_TRY(scratch.get(&nl_orig, seg_cap), "child sizes");
    PTB_TRY(scratch.get(&child, seg_cap), "child counts");
    PTB_TRY(scratch.get(&child_ex, seg_cap), "child counts");
    PTB_TRY(scratch.get(&accbuf, 36 * (static_cast<size_t>(n) / SMALL + 2)), "group boxes");
    PTB_TRY(scratch.get(&n_large, 1), "counter");

    const size_t pair_cap = static_cast<size_t>(n) + static_cast<size_t>(n) / 2 + 2; // n - 1 inner nodes + at most one unused slot per two of them
    float *pairs = nullptr;
    uint32_t *parent = nullptr;
    float *root_box = nullptr;
    PTB_TRY(scratch.get(&pairs, 16 * pair_cap), "pair records");
    PTB_TRY(scratch.get(&parent, pair_cap), "parent links");
    PTB_TRY(scratch.get(&root_box, 6), "root box");
    PTB_TRY(hipMemsetAsync(pairs, 0, 16 * pair_cap * sizeof(float), stream), "clearing the pair records");

    uint32_t *totals = nullptr; // host-visible: {inner children of the level, slots appended by the level}
    PTB_TRY(hipHostMalloc(reinterpret_cast<void **>(&totals), 2 * sizeof(uint32_t), hipHostMallocDefault), "pinned totals");
    struct HostGuard {
        void *p;
        ~HostGuard() { (void)hipHostFree(p); }
    } host_guard{totals};

    // temporary storage for the scans and the sorts
    size_t temp_bytes = 0;
    {
        size_t b = 0;
        PTB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, b, flag, flag_ex, static_cast<int>(n + 1), stream), "scan sizing");
        temp_bytes = std::max(temp_bytes, b);
        PTB_TRY(hipcub::DeviceScan::ExclusiveScan(nullptr, b, flag3, flag3_ex, U3Op(), U3{0, 0, 0}, static_cast<int>(n + 1), stream), "scan sizing");
        temp_bytes = std::max(temp_bytes, b);
        PTB_TRY(hipcub::DeviceScan::ExclusiveScan(nullptr, b, child, child_ex, CnOp(), Cn{0, 0, 0}, static_cast<int>(seg_cap), stream), "scan sizing");
        temp_bytes = std::max(temp_bytes, b);
        PTB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, b, leaf.lo[0], med3, lists[0].ord, lists[0].srt[0], static_cast<int>(n), 0, 32, stream), "sort sizing");
        temp_bytes = std::max(temp_bytes, b);
    }
    void *temp = nullptr;
    {
        uint8_t *t8 = nullptr;
        PTB_TRY(scratch.get(&t8, temp_bytes), "scan temporary storage");
        temp = t8;
    }

    // ---- the three sorted lists -------------------------------------------------------------------------------------------------
    hipLaunchKernelGGL(k_iota, grid_for(n), dim3(256), 0, stream, n, lists[0].ord, segid);
    {
        float *keys_out = nullptr;
        PTB_TRY(scratch.get(&keys_out, n), "sort keys");
        for(int k = 0; k < 3; k++) {
            size_t b = temp_bytes;
            PTB_TRY(hipcub::DeviceRadixSort::SortPairs(temp, b, leaf.lo[k], keys_out, lists[0].ord, lists[0].srt[k], static_cast<int>(n), 0, 32, stream), "radix sort");
        }
    }
    hipLaunchKernelGGL(k_first_segment, dim3(1), dim3(1), 0, stream, n, tables[0]);

    // ---- levels ---------------------------------------------------------------------------------------------------------------
    const int align = in.align_siblings ? 1 : 0;
    std::vector<std::pair<uint32_t, uint32_t>> level_slots; // [begin, end) of every level's pair slots
    level_slots.push_back({0u, 1u});
    uint32_t n_seg = 1, total_slots = 1;
    int cur = 0;
    while(n_seg > 0) {
        const Lists &L = lists[cur], &Lnext = lists[cur ^ 1];
        const SegTable &T = tables[cur], &Tnext = tables[cur ^ 1];
        hipLaunchKernelGGL(k_median, grid_for(n_seg), dim3(256), 0, stream, n_seg, T, L, leaf, med3, accbuf);
        hipLaunchKernelGGL(k_accumulate, grid_for((static_cast<size_t>(n) + ACC_STEPS - 1) / ACC_STEPS), dim3(256), 0, stream, n, segid, T, L, leaf, med3, accbuf);
        hipLaunchKernelGGL(k_choose, grid_for(n_seg), dim3(256), 0, stream, n_seg, T, L, leaf, med3, accbuf, axis, med);
        hipLaunchKernelGGL(k_flag, grid_for(static_cast<size_t>(n) + 1), dim3(256), 0, stream, n, segid, L, leaf, axis, med, flag);
        size_t b = temp_bytes;
        PTB_TRY(hipcub::DeviceScan::ExclusiveSum(temp, b, flag, flag_ex, static_cast<int>(n + 1), stream), "flag scan");
        hipLaunchKernelGGL(k_split, grid_for(n_seg), dim3(256), 0, stream, n_seg, T, flag_ex, align, nl_final, nl_orig, child);
        hipLaunchKernelGGL(k_scatter_ord, grid_for(n), dim3(256), 0, stream, n, segid, T, flag, flag_ex, nl_final, nl_orig, L.ord, Lnext.ord, side);
        hipLaunchKernelGGL(k_flag3, grid_for(static_cast<size_t>(n) + 1), dim3(256), 0, stream, n, segid, L, side, flag3);
        b = temp_bytes;
        PTB_TRY(hipcub::DeviceScan::ExclusiveScan(temp, b, flag3, flag3_ex, U3Op(), U3{0, 0, 0}, static_cast<int>(n + 1), stream), "list scan");
        hipLaunchKernelGGL(k_scatter_srt, grid_for(n), dim3(256), 0, stream, n, segid, T, flag3, flag3_ex, nl_final, L, Lnext);
        b = temp_bytes;
        PTB_TRY(hipcub::DeviceScan::ExclusiveScan(temp, b, child, child_ex, CnOp(), Cn{0, 0, 0}, static_cast<int>(n_seg), stream), "child scan");
        PTB_TRY(hipMemsetAsync(n_large, 0, sizeof(uint32_t), stream), "counter reset");
        hipLaunchKernelGGL(k_children, grid_for(n_seg), dim3(256), 0, stream, n_seg, T, child, child_ex, total_slots, align, nl_final, Lnext.ord, leaf, Tnext, pairs, parent,
                           n_large, totals);
        hipLaunchKernelGGL(k_segid, grid_for(n), dim3(256), 0, stream, n, segid, T, nl_final, child_ex);
        PTB_TRY(hipGetLastError(), "level kernels");
        PTB_TRY(hipStreamSynchronize(stream), "level");
        const uint32_t next_seg = totals[0], added = totals[1];
        if(static_cast<size_t>(total_slots) + added > pair_cap || next_seg > seg_cap) {
            if(error_text != nullptr) {
                *error_text = "pair record capacity exceeded";
            }
            return hipErrorOutOfMemory;
        }
        if(added > 0) {
            level_slots.push_back({total_slots, total_slots + added});
        }
        total_slots += added;
        n_seg = next_seg;
        cur ^= 1;
        if(level_slots.size() > PT_MAX_DEPTH + 1) {
            break; // deeper than the traversal supports; the caller rejects the scene by its depth
        }
    }

    // ---- boxes, bottom-up -------------------------------------------------------------------------------------------------------
    for(size_t l = level_slots.size(); l-- > 0;) {
        const uint32_t begin = level_slots[l].first, end = level_slots[l].second;
        hipLaunchKernelGGL(k_union, grid_for(end - begin), dim3(256), 0, stream, begin, end, pairs, parent, root_box);
    }
    PTB_TRY(hipGetLastError(), "box kernels");
    PTB_TRY(hipEventRecord(ev1, stream), "event");
    float box_host[6];
    PTB_TRY(hipMemcpyAsync(box_host, root_box, sizeof(box_host), hipMemcpyDeviceToHost, stream), "root box");
    PTB_TRY(hipStreamSynchronize(stream), "build");
    PTB_TRY(hipEventElapsedTime(&out.build_ms, ev0, ev1), "event");

    out.n_pairs = total_slots;
    out.level_begin.clear();
    for(const auto &range : level_slots) {
        out.level_begin.push_back(range.first);
    }
    out.level_begin.push_back(total_slots);
    out.depth = static_cast<uint32_t>(level_slots.size()) + 1;
    out.root_ref = 0;
    for(int k = 0; k < 3; k++) {
        out.root_lo[k] = box_host[k];
        out.root_hi[k] = box_host[3 + k];
    }
    out.pairs = reinterpret_cast<float4 *>(pairs);
    scratch.keep(pairs);
    out.dfs = lists[cur].ord;
    scratch.keep(lists[cur].ord);
    return hipSuccess;
}

__global__ void k_link_records(float4 *__restrict__ recs, uint32_t pair_base, uint32_t n_pairs, uint32_t sphere_base) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if(p >= n_pairs) {
        return;
    }
    float4 *q3 = recs + 4 * (static_cast<size_t>(pair_base) + p) + 3;
    float4 v = *q3;
    uint32_t ref[2] = {__float_as_uint(v.x), __float_as_uint(v.y)};
    for(int c = 0; c < 2; c++) {
        if(ref[c] == PT_REF_NONE) {
            continue; // (an unused slot of the sibling alignment)
        }
        if((ref[c] & PT_REF_LEAF) == 0) {
            ref[c] += pair_base;
        }
        else if((ref[c] & PT_REF_SPHERE) != 0) {
            ref[c] += sphere_base;
        }
    }
    v.x = __uint_as_float(ref[0]);
    v.y = __uint_as_float(ref[1]);
    *q3 = v;
}

hipError_t pt_link_records(hipStream_t stream, float4 *recs, uint32_t pair_base, uint32_t n_pairs, uint32_t sphere_base) {
    if(n_pairs == 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_link_records, grid_for(n_pairs), dim3(256), 0, stream, recs, pair_base, n_pairs, sphere_base);
    return hipGetLastError();
}

hipError_t pt_build_order_subset(hipStream_t stream, const uint32_t *dfs, uint32_t n_objects, const std::vector<uint32_t> &mask_bits, uint32_t n_selected,
                                 std::vector<uint32_t> &ordered) {
    ordered.clear();
    if(n_selected == 0 || n_objects == 0) {
        return hipSuccess;
    }
    const char **error_text = nullptr;
    Scratch scratch;
    uint32_t *mask = nullptr, *count = nullptr;
    uint2 *found = nullptr;
    PTB_TRY(scratch.get(&mask, mask_bits.size()), "mask");
    PTB_TRY(scratch.get(&count, 1), "counter");
    PTB_TRY(scratch.get(&found, n_selected), "selection");
    PTB_TRY(hipMemcpyAsync(mask, mask_bits.data(), mask_bits.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream), "mask upload");
    PTB_TRY(hipMemsetAsync(count, 0, sizeof(uint32_t), stream), "counter reset");
    hipLaunchKernelGGL(k_select, grid_for(n_objects), dim3(256), 0, stream, n_objects, dfs, mask, found, count, n_selected);
    PTB_TRY(hipGetLastError(), "selection kernel");
    std::vector<uint2> host(n_selected);
    uint32_t got = 0;
    PTB_TRY(hipMemcpyAsync(&got, count, sizeof(uint32_t), hipMemcpyDeviceToHost, stream), "count");
    PTB_TRY(hipMemcpyAsync(host.data(), found, sizeof(uint2) * n_selected, hipMemcpyDeviceToHost, stream), "selection");
    PTB_TRY(hipStreamSynchronize(stream), "selection");
    if(got != n_selected) {
        return hipErrorUnknown; // every object appears exactly once in the leaf order
    }
    std::sort(host.begin(), host.end(), [](const uint2 &a, const uint2 &b) { return a.x < b.x; });
    ordered.resize(n_selected);
    for(uint32_t i = 0; i < n_selected; i++) {
        ordered[i] = host[i].y;
    }
    return hipSuccess;
}

