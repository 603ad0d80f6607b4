// pt_shade.hip -- the per-path state machine: camera ray generation, vertex shading with next-event estimation, Russian
// roulette, BSDF sampling (impl::getSample, src/worker.cpp:26-146) and the per-pixel adaptive estimator
// (processItem, src/worker.cpp:149-326), one stream slot per lane.
//
// One wavefront iteration = one launch of this kernel + one launch of the traversal kernel (pt_trace.hip):
//   shade(i):  every stream with a path in flight first adds the shadow-ray results of its previous vertex to out_spectrum in
//              the reference's order (worker.cpp:76-103), then looks at the extension ray's hit: miss -> the sample is
//              finished; hit -> shade that vertex: emission (worker.cpp:62-64), Russian-roulette draw (:67-70), light sampling
//              (Scene::sampleLights, scene.cpp:222-289) with one shadow ray per light sample, BSDF sample (:117-131).
//              A finished sample goes through the estimator and the stream immediately starts its next sample (or pixel).
//   trace(i):  all rays appended by shade(i).
// A path that ends at a vertex (roulette) still has that vertex's shadow rays to wait for; where the estimator provably cannot
// stop at this sample, the NEXT sample's camera ray is drawn and traced in the same iteration (PT_F_OVERLAP) -- the draws keep
// their order (all draws of the ending sample are made), and one iteration per sample is saved.
// The random draws of one stream are consumed strictly in the reference's order because a stream has at most one path in
// flight and every draw of a vertex (roulette, lights, BSDF) is made by the single invocation that shades the vertex.
//
// Rays are appended to the queue shard of this workgroup (blockIdx.x % 8): each lane reserves its worst case
// (1 extension ray + L + k shadow rays) and the workgroup makes ONE atomic for all its lanes, lane offsets coming from
// ballot/popcount prefix sums and a per-wave table in LDS; light samples that turn out not to need a ray leave a PT_DEST_NULL hole.
#include "pt_device.h"
#include "pt_kernels.h"
#include "pt_shading.h"

using namespace ptd;

namespace {

// ---- the kernel -------------------------------------------------------------------------------------------------------

#ifndef PT_SHADE_WAVES
#define PT_SHADE_WAVES 3
#endif
__global__ __launch_bounds__(256, PT_SHADE_WAVES) void pt_shade_kernel(PtDevScene sc, PtDevCamera cam, PtDevOptions opt, PtPaths P, PtQueue q, PtCarry carry, int parity,
                                                       int shard_mode, float4 *__restrict__ image, PtDevCounters *counters) {
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    // Contiguous eighths of the streams (= contiguous image regions) feed one queue shard each; the traversal workgroups of one
    // XCD drain one shard first, so the subtrees a region's rays walk stay in that XCD's L2.  (shard_mode 0: round-robin.)
    const uint32_t shard = shard_mode != 0 ? (uint32_t)(((unsigned long long)blockIdx.x * PT_SHARDS) / gridDim.x) : blockIdx.x % PT_SHARDS;
    const uint32_t n_light_samples = sc.n_lights + sc.n_object_samples;

    if(p == 0) {
        // the traversal launch that follows resumes pool `parity` and suspends into pool `parity ^ 1`
        carry.count[(parity ^ 1) * PT_QSTRIDE] = 0;
        carry.head[parity * PT_QSTRIDE] = 0;
    }
    if(q.next_header != nullptr && p < 2 * PT_SHARDS) {
        // the queue headers alternate between launches: the one the next launch appends to was last read two kernels ago
        q.next_header[p * PT_QSTRIDE] = 0;
    }

    uint32_t flags = PT_F_DONE;
    if(p < P.n) {
        flags = P.flags[p];
    }
    bool alive = !(flags & PT_F_DONE);
    if(alive && (flags & PT_F_IN_FLIGHT)) {
        // a stream whose rays are still walking (suspended by the last traversal launch) sits this iteration out
        bool ready = !(flags & PT_F_HAS_EXT) || P.hit[p].y != PT_REF_PENDING;
        uint32_t mask = P.nee_mask[p];
        for(uint32_t j = 0; mask != 0; j++, mask >>= 1) {
            if((mask & 1u) && P.vis[(size_t)j * P.nee_stride + p] == PT_VIS_PENDING) {
                ready = false;
            }
        }
        alive = ready;
    }

    // ---- phase A: consume the results of the previous iteration ------------------------------------------------------
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
    unsigned long long n_samples = 0, n_vertices = 0;

    if(alive) {
        rng = P.rng[p];
        if(flags & PT_F_IN_FLIGHT) {
            out = c4(P.out[p]);
            // shadow rays of the previous vertex, in light order (worker.cpp:76-103)
            uint32_t mask = P.nee_mask[p];
            for(uint32_t j = 0; mask != 0; j++, mask >>= 1) {
                if((mask & 1u) && P.vis[(size_t)j * P.nee_stride + p] != 0u) {
                    out = out + c4(P.nee[(size_t)j * P.nee_stride + p]);
                }
            }
            if(flags & PT_F_OVERLAP) {
                // the previous sample is complete now (worker.cpp:141-145, 196-237); it was collected (it had a vertex), it is
                // not the pixel's last sample and the estimator cannot accept at it (see safe_overlap below)
                PtEstimator e = P.est[p];
                out.a = 1.0f;
                (void)estimator_add(e, P.cand + (size_t)p * PT_MAX_CANDIDATES, opt, out);
                e.pixel_sample++;
                n_samples++;
                P.est[p] = e;
                flags &= ~(PT_F_OVERLAP | PT_F_COLLECTED | PT_F_SAFE);
                if(estimator_safe_to_overlap(e, opt)) {
                    flags |= PT_F_SAFE;
                }
                out = c4(0, 0, 0, 0);
            }
            bool finished = true;
            if(flags & PT_F_HAS_EXT) {
                const uint2 h = P.hit[p];
                hit_t = __uint_as_float(h.x);
                hit_ref = h.y;
                if(!(hit_t < 0.0f)) {
                    finished = false;
                    shade_vertex = true;
                }
            }
            if(finished) {
                // getSample returns (worker.cpp:141-145); run the estimator
                PtEstimator e = P.est[p];
                PtCandidate *cand = P.cand + (size_t)p * PT_MAX_CANDIDATES;
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
                    const int4 rc = P.rect[p];
                    const int32_t cur = P.cursor[p];
                    const int32_t px = rc.x + cur % rc.z, py = rc.y + cur / rc.z;
                    image[(size_t)py * opt.image_width + px] = f4(value);
                    P.cursor[p] = cur + 1;
                    flags &= ~PT_F_PIXEL;
                }
                P.est[p] = e;
                flags &= ~(PT_F_IN_FLIGHT | PT_F_HAS_EXT | PT_F_COLLECTED | PT_F_SAFE);
                if((flags & PT_F_PIXEL) && estimator_safe_to_overlap(e, opt)) {
                    flags |= PT_F_SAFE; // for the pixel's next sample, which starts below
                }
                start_sample = true;
            }
            else {
                const float4 o4 = P.ray_o[p], d4 = P.ray_d[p];
                ro = v3(o4.x, o4.y, o4.z);
                rd = v3(d4.x, d4.y, d4.z);
                contribution_unweighted = o4.w;
                spectrum = c4(P.spectrum[p]);
                divisor = P.divisor[p];
                bounce_pd = P.bounce_pd[p];
                path_length = P.path_length[p];
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
        const int4 rc = P.rect[p];
        int32_t cur = P.cursor[p];
        bool have_pixel = false;
        while(cur < rc.z * rc.w) {
            if(!(flags & PT_F_PIXEL)) {
                PtEstimator e;
                estimator_reset(e, opt);
                P.est[p] = e;
                flags = (flags | PT_F_PIXEL) & ~PT_F_SAFE;
                if(estimator_safe_to_overlap(e, opt)) {
                    flags |= PT_F_SAFE;
                }
                if(opt.max_sample_count <= 0) {
                    // no sample at all: the pixel stays (0, 0, 0, 0) (worker.cpp:193,263-265)
                    const int32_t px = rc.x + cur % rc.z, py = rc.y + cur / rc.z;
                    image[(size_t)py * opt.image_width + px] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    cur++;
                    flags &= ~PT_F_PIXEL;
                    continue;
                }
            }
            have_pixel = true;
            break;
        }
        P.cursor[p] = cur;
        if(!have_pixel) {
            flags = PT_F_DONE;
            atomicAdd(&counters->streams_done, 1ULL);
        }
        else {
            ext = shoot_camera(rc, cur);
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
        mat = material_load(sc.materials, material_index);

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

    // ---- reserve queue space: one atomic per workgroup ----------------------------------------------------------------------------
    // lane offsets come from ballot/popcount prefix sums, wave offsets from a 4-entry LDS table; thread 0 makes the one atomic
    __shared__ uint32_t wave_totals[4];
    __shared__ uint32_t block_base;
    const unsigned long long ext_mask = __ballot(emit_ext);
    const unsigned long long nee_mask_w = __ballot(want_nee);
    const unsigned long long lt = (1ULL << lane) - 1ULL;
    const uint32_t wave_total = (uint32_t)__popcll(ext_mask) + n_light_samples * (uint32_t)__popcll(nee_mask_w);
    const int wave = threadIdx.x >> 6;
    if(lane == 0) {
        wave_totals[wave] = wave_total;
    }
    __syncthreads();
    if(threadIdx.x == 0) {
        const uint32_t total = wave_totals[0] + wave_totals[1] + wave_totals[2] + wave_totals[3];
        block_base = total > 0 ? atomicAdd(&q.count[shard * PT_QSTRIDE], total) : 0u;
    }
    __syncthreads();
    uint32_t wave_base = block_base;
    for(int w = 0; w < wave; w++) {
        wave_base += wave_totals[w];
    }
    const size_t slot0 = (size_t)shard * q.shard_capacity + wave_base + (uint32_t)__popcll(ext_mask & lt) + n_light_samples * (uint32_t)__popcll(nee_mask_w & lt);
    // this lane's extension ray goes to slot0, its shadow rays to slot0 + (emit_ext ? 1 : 0) + j

    // ---- vertex, part 2: light sampling, shadow rays, bounce (worker.cpp:73-138) -------------------------------------------------
    uint32_t nee_out_mask = 0;
    if(shade_vertex) {
        const float epsilon = opt.epsilon;
        const size_t nee_slot = slot0 + (emit_ext ? 1 : 0);
        for(uint32_t j = 0; j < n_light_samples; j++) {
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
                valid = sample_emissive(sc, pos, rng, light_pos, light_spectrum, lpd);
            }
            bool need_ray = false;
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
                        P.nee[(size_t)j * P.nee_stride + p] = f4(weighed);
                        nee_out_mask |= 1u << j;
                        const float threshold = len(to_light) - epsilon;
                        if(threshold <= 0.0f) {
                            // light_t < 0 || light_t >= threshold holds for every light_t
                            P.vis[(size_t)j * P.nee_stride + p] = 1u;
                        }
                        else {
                            need_ray = true;
                            P.vis[(size_t)j * P.nee_stride + p] = PT_VIS_PENDING;
                            const V3 so = pos + light_dir * epsilon;
                            q.ray_o[nee_slot + j] = make_float4(so.x, so.y, so.z, threshold);
                            q.ray_d[nee_slot + j] =
                              make_float4(light_dir.x, light_dir.y, light_dir.z, __uint_as_float(PT_DEST_SHADOW | (j * P.nee_stride + p)));
                        }
                    }
                }
            }
            if(want_nee && !need_ray) {
                q.ray_d[nee_slot + j] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(PT_DEST_NULL));
            }
        }

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
            // start the next sample of this pixel now; this sample is finished by the next invocation (PT_F_OVERLAP)
            ext = shoot_camera(P.rect[p], P.cursor[p]);
            spectrum = c4(1, 1, 1, 1);
            contribution_unweighted = 1.0f;
            divisor = 1.0;
            bounce_pd = 1.0;
            path_length = 0;
            flags |= PT_F_OVERLAP;
        }
        else if(path_ends && emit_ext) {
            q.ray_d[slot0] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(PT_DEST_NULL));
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
        q.ray_o[slot0] = make_float4(ext.o.x, ext.o.y, ext.o.z, 0.0f);
        q.ray_d[slot0] = make_float4(ext.d.x, ext.d.y, ext.d.z, __uint_as_float(p));
        P.hit[p] = make_uint2(0u, PT_REF_PENDING);
    }
    if(alive) {
        P.flags[p] = flags;
        P.rng[p] = rng;
        if(!(flags & PT_F_DONE)) {
            P.ray_o[p] = make_float4(ext.o.x, ext.o.y, ext.o.z, contribution_unweighted);
            P.ray_d[p] = make_float4(ext.d.x, ext.d.y, ext.d.z, 0.0f);
            P.spectrum[p] = f4(spectrum);
            P.out[p] = f4(out);
            P.divisor[p] = divisor;
            P.bounce_pd[p] = bounce_pd;
            P.path_length[p] = path_length;
            P.nee_mask[p] = nee_out_mask;
        }
    }

    // counters: one slot per wave, plain adds (see pt_trace.hip)
    for(int off = 32; off > 0; off >>= 1) {
        n_samples += __shfl_down(n_samples, off);
        n_vertices += __shfl_down(n_vertices, off);
    }
    if(lane == 0 && (n_samples | n_vertices)) {
        unsigned long long *slot = P.wave_counters + 2 * (size_t)(p >> 6);
        slot[0] += n_samples;
        slot[1] += n_vertices;
    }
}

__global__ void pt_init_tiles_kernel(PtPaths P, const int4 *__restrict__ tiles, const uint32_t *__restrict__ tile_offset, uint32_t n_tiles, uint64_t base_seed) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if(p >= P.n) {
        return;
    }
    // binary search for the tile holding slot p
    uint32_t lo = 0, hi = n_tiles;
    while(hi - lo > 1) {
        const uint32_t mid = (lo + hi) / 2;
        if(tile_offset[mid] <= p) {
            lo = mid;
        }
        else {
            hi = mid;
        }
    }
    const int4 t = tiles[lo];
    const uint32_t k = p - tile_offset[lo];
    const int32_t x = t.x + (int32_t)(k % (uint32_t)t.z), y = t.y + (int32_t)(k / (uint32_t)t.z);
    P.rect[p] = make_int4(x, y, 1, 1);
    const uint64_t seed = pixel_seed(base_seed, x, y);
    P.rng[p] = seed ^ (~seed << 32); // RandomEngine(seed), base.h:26
    P.cursor[p] = 0;
    P.flags[p] = 0;
    P.nee_mask[p] = 0;
}

__global__ void pt_init_streams_kernel(PtPaths P) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if(p >= P.n) {
        return;
    }
    P.cursor[p] = 0;
    P.flags[p] = 0;
    P.nee_mask[p] = 0;
}

__global__ void pt_spin_kernel(unsigned long long ticks) {
    const unsigned long long start = wall_clock64(); // constant 100 MHz
    while(wall_clock64() - start < ticks) {
        __builtin_amdgcn_s_sleep(8);
    }
}

} // namespace

void pt_launch_spin(hipStream_t stream, uint32_t microseconds) {
    hipLaunchKernelGGL(pt_spin_kernel, dim3(1), dim3(64), 0, stream, 100ULL * microseconds);
}

uint64_t pt_host_pixel_seed(uint64_t base, int32_t x, int32_t y) {
    uint64_t z = base + 0x9E3779B97F4A7C15ULL * (1ULL + (((uint64_t)(uint32_t)y) << 32) + (uint64_t)(uint32_t)x);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

void pt_launch_init_tiles(hipStream_t stream, PtPaths paths, const int4 *tiles, const uint32_t *tile_offset, uint32_t n_tiles, uint64_t base_seed) {
    if(paths.n == 0) {
        return;
    }
    hipLaunchKernelGGL(pt_init_tiles_kernel, dim3((paths.n + 255) / 256), dim3(256), 0, stream, paths, tiles, tile_offset, n_tiles, base_seed);
}

void pt_launch_init_streams(hipStream_t stream, PtPaths paths) {
    if(paths.n == 0) {
        return;
    }
    hipLaunchKernelGGL(pt_init_streams_kernel, dim3((paths.n + 255) / 256), dim3(256), 0, stream, paths);
}

void pt_launch_shade(hipStream_t stream, const PtDevScene &scene, const PtDevCamera &camera, const PtDevOptions &options, PtPaths paths, PtQueue queue,
                     PtCarry carry, int parity, int shard_mode, float4 *image, PtDevCounters *counters) {
    if(paths.n == 0) {
        return;
    }
    hipLaunchKernelGGL(pt_shade_kernel, dim3((paths.n + 255) / 256), dim3(256), 0, stream, scene, camera, options, paths, queue, carry, parity, shard_mode, image, counters);
}
