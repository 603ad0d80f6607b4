// pt_shading.h -- device functions of the shading side of the hot path: the per-pixel adaptive estimator of processItem
// (src/worker.cpp:149-326), one light sample of Scene::sampleLights (src/scene/scene.cpp:238-286), the per-pixel engine seed.
// Shared by the kernels of pt_path.hip; everything keeps the reference's evaluation order (see pt_device.h).
#ifndef PT_SHADING_H
#define PT_SHADING_H

#include "pt_device.h"
#include "pt_kernels.h"

namespace ptd {

PT_D C4 ld4(const float *p) {
    return c4(p[0], p[1], p[2], p[3]);
}
PT_D void st4(float *p, C4 c) {
    p[0] = c.r;
    p[1] = c.g;
    p[2] = c.b;
    p[3] = c.a;
}
PT_D float4 f4(C4 c) {
    return make_float4(c.r, c.g, c.b, c.a);
}

// worker.cpp:12-14
PT_D float get_contribution(C4 c) {
    return (c.r + c.g + c.b) / 3.0f;
}

PT_D uint64_t pixel_seed(uint64_t base, int32_t x, int32_t y) {
    uint64_t z = base + 0x9E3779B97F4A7C15ULL * (1ULL + (((uint64_t)(uint32_t)y) << 32) + (uint64_t)(uint32_t)x);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// ---- per-pixel estimator: the body of processItem's sample loop after getSample returned (worker.cpp:196-260) -------------
// returns true when the loop breaks with accepted_candidate
PT_D bool estimator_add(PtEstimator &e, PtCandidate *cand, const PtDevOptions &opt, C4 color_contribution) {
    e.contribution_count++;
    e.stats_sample_index++;
    C4 agg = ld4(e.sample_aggregate) + color_contribution;

    if(e.stats_sample_index == opt.stats_sample_count) {
        agg = agg / (float)opt.stats_sample_count;

        C4 mean = ld4(e.contribution_mean);
        C4 delta = agg - mean;
        mean = mean + delta / (float)(e.contribution_count / opt.stats_sample_count);
        C4 delta2 = agg - mean;
        st4(e.contribution_mean, mean);
        st4(e.contribution_m2, ld4(e.contribution_m2) + delta * delta2);

        if(e.candidate_count == opt.candidate_batch_count) {
            if(e.n_candidates < PT_MAX_CANDIDATES) {
                PtCandidate &c = cand[e.n_candidates];
                for(int k = 0; k < 4; k++) {
                    c.mean[k] = e.candidate_mean[k];
                    c.m2[k] = e.candidate_m2[k];
                }
                c.count = e.candidate_count;
            }
            e.n_candidates++;
            st4(e.candidate_mean, c4(0, 0, 0, 0));
            st4(e.candidate_m2, c4(0, 0, 0, 0));
            e.candidate_count = 0;
        }

        e.candidate_count++;
        C4 cmean = ld4(e.candidate_mean);
        C4 cdelta = agg - cmean;
        cmean = cmean + cdelta / (float)e.candidate_count;
        C4 cdelta2 = agg - cmean;
        st4(e.candidate_mean, cmean);
        st4(e.candidate_m2, ld4(e.candidate_m2) + cdelta * cdelta2);

        e.stats_sample_index = 0;
        agg = c4(0, 0, 0, 0);
    }
    st4(e.sample_aggregate, agg);

    st4(e.pixel_value, ld4(e.pixel_value) + color_contribution);
    e.collected_sample_count++;

    const int min_needed = opt.min_sample_count > 2 ? opt.min_sample_count : 2;
    if(e.stats_sample_index == 0 && e.collected_sample_count >= min_needed) {
        bool passed_check = false;
        const int batches = e.contribution_count / opt.stats_sample_count;
        if(batches >= 2) {
            C4 m2w = ld4(e.contribution_m2) / (float)(batches - 1);
            float stddev = __builtin_sqrtf(m2w.r + m2w.g + m2w.b);
            // worker.cpp:245: the 1E-5 literal promotes the ratio to double
            if(stddev < 1E-4f || (double)stddev / ((double)(9.0f * get_contribution(ld4(e.contribution_mean))) + 1E-5) < (double)0.2f) {
                passed_check = true;
                e.remaining_checks--;
                if(e.remaining_checks <= 0) {
                    return true;
                }
            }
        }
        if(!passed_check) {
            e.remaining_checks = opt.check_sample_count;
        }
    }
    return false;
}

// after the sample loop: worker.cpp:263-319
PT_D C4 estimator_finish(PtEstimator &e, PtCandidate *cand, const PtDevOptions &opt, bool accepted) {
    C4 pixel_value = ld4(e.pixel_value);
    if(e.collected_sample_count > 0) {
        pixel_value = pixel_value * (1.0f / (float)e.collected_sample_count);
    }
    if(accepted) {
        return pixel_value;
    }
    int n = e.n_candidates < PT_MAX_CANDIDATES ? e.n_candidates : PT_MAX_CANDIDATES;
    // the open candidate is appended (worker.cpp:267-271)
    const bool has_open = e.candidate_count > 0;
    const int min_count = (opt.candidate_batch_count * 3) / 4 > 2 ? (opt.candidate_batch_count * 3) / 4 : 2;

    // qualifying candidates, insertion-sorted by stddev the way std::sort orders <= 16 elements (libstdc++ __insertion_sort)
    float sd[PT_MAX_CANDIDATES + 1];
    int id[PT_MAX_CANDIDATES + 1];
    int m = 0;
    for(int i = 0; i < n + (has_open ? 1 : 0); i++) {
        int count;
        C4 m2;
        if(i < n) {
            count = cand[i].count;
            m2 = ld4(cand[i].m2);
        }
        else {
            count = e.candidate_count;
            m2 = ld4(e.candidate_m2);
        }
        if(count < min_count) {
            continue;
        }
        C4 m2w = m2 / (float)count;
        float stddev = __builtin_sqrtf(m2w.r + m2w.g + m2w.b);
        int j = m;
        if(m > 0 && stddev < sd[0]) {
            for(; j > 0; j--) {
                sd[j] = sd[j - 1];
                id[j] = id[j - 1];
            }
        }
        else {
            while(j > 0 && stddev < sd[j - 1]) {
                sd[j] = sd[j - 1];
                id[j] = id[j - 1];
                j--;
            }
        }
        sd[j] = stddev;
        id[j] = i;
        m++;
    }
    if(m == 0) {
        return pixel_value;
    }
    auto mean_of = [&](int i) { return i < n ? ld4(cand[i].mean) : ld4(e.candidate_mean); };
    pixel_value = mean_of(id[0]);
    float stddev = sd[0];
    for(int i = 1; i < m; i++) {
        float stddev_other = sd[i];
        if(stddev_other < fmax_std(stddev + 0.005f, stddev * 1.01f)) {
            pixel_value = pixel_value + (mean_of(id[i]) - pixel_value) / (float)(i + 1);
            stddev = stddev_other;
        }
        else {
            break;
        }
    }
    return pixel_value;
}

// Can the sample AFTER the one that is about to start begin before that one is handed to the estimator?  Only if another sample
// of the same pixel certainly follows it: it is not the last one (worker.cpp:193) and the convergence test cannot run at it --
// the test runs only when a COLLECTED sample closes a statistics batch with at least max(min_sample_count, 2) samples collected
// (worker.cpp:239).  `e` is the estimator state before the sample; the question is only asked for samples that reach a vertex.
PT_D bool estimator_safe_to_overlap(const PtEstimator &e, const PtDevOptions &opt) {
    const bool closes_batch = e.stats_sample_index + 1 == opt.stats_sample_count;
    const int min_needed = opt.min_sample_count > 2 ? opt.min_sample_count : 2;
    return e.pixel_sample + 1 < opt.max_sample_count && !(closes_batch && e.collected_sample_count + 1 >= min_needed);
}

PT_D void estimator_reset(PtEstimator &e, const PtDevOptions &opt) {
    for(int k = 0; k < 4; k++) {
        e.pixel_value[k] = 0.0f;
        e.contribution_mean[k] = 0.0f;
        e.contribution_m2[k] = 0.0f;
        e.sample_aggregate[k] = 0.0f;
        e.candidate_mean[k] = 0.0f;
        e.candidate_m2[k] = 0.0f;
    }
    e.collected_sample_count = 0;
    e.contribution_count = 0;
    e.stats_sample_index = 0;
    e.candidate_count = 0;
    e.remaining_checks = opt.check_sample_count;
    e.n_candidates = 0;
    e.pixel_sample = 0;
    e.pad = 0;
}

// ---- Scene::sampleLights, one light sample (scene.cpp:238-286) ----------------------------------------------------------
// Draws 3 numbers; returns false when the reference `continue`s.
// The emitter tables as they lie in global memory ...
struct EmisGlobal {
    const PtDevScene &sc;
    PT_D float cdf(int i) const { return sc.emis_cdf[i]; }
    PT_D float4 rec(int i, int k) const { return sc.emis[4 * (size_t)i + k]; }
    PT_D V3 tri_normal_at(int, uint32_t ref, V3 pos) const {
        uint32_t mat_unused;
        return object_normal(sc, ref, pos, mat_unused);
    }
};

template<typename Tables>
PT_D bool sample_emissive(const PtDevScene &sc, const Tables &tb, V3 pos, uint64_t &rng, V3 &light_pos, C4 &spectrum, float &pd) {
    const float r = rng_uniform01(rng);
    // std::lower_bound over the CDF
    int lo = 0, n = (int)sc.n_emis;
    while(n > 0) {
        const int half = n >> 1;
        if(tb.cdf(lo + half) < r) {
            lo = lo + half + 1;
            n = n - half - 1;
        }
        else {
            n = half;
        }
    }
    const int object_index = lo;
    float selection_p = tb.cdf(object_index);
    if(object_index > 0) {
        selection_p -= tb.cdf(object_index - 1);
    }
    selection_p *= (float)sc.n_object_samples;

    const float4 e0 = tb.rec(object_index, 0), e1 = tb.rec(object_index, 1), e2 = tb.rec(object_index, 2), e3 = tb.rec(object_index, 3);
    const uint32_t ref = __float_as_uint(e2.y);
    V3 surface_pos, surface_n;
    float surface_p;
    bool surface_cull;
    if(ref & PT_REF_SPHERE) {
        // Sphere::sampleSurface, object.cpp:101-116
        const V3 origin = v3(e0.x, e0.y, e0.z);
        const float radius = e0.w;
        const float radius2 = radius * radius;
        const float theta = 2.0f * PT_PI_F * rng_uniform01(rng);
        const float phi = ptm::acosf_glibc(1.0f - 2.0f * rng_uniform01(rng));
        const float x = ptm::sinf_glibc(phi) * ptm::cosf_glibc(theta);
        const float y = ptm::sinf_glibc(phi) * ptm::sinf_glibc(theta);
        const float z = ptm::cosf_glibc(phi);
        surface_pos = origin + v3(x, y, z) * radius;
        surface_p = 1.0f / (4.0f * PT_PI_F * radius2);
        surface_cull = false;
        surface_n = normalize(surface_pos - origin);
    }
    else {
        // Triangle::sampleSurface, object.cpp:192-207
        const V3 a = v3(e0.x, e0.y, e0.z), b = v3(e0.w, e1.x, e1.y), c = v3(e1.z, e1.w, e2.x);
        const float r1 = rng_uniform01(rng);
        const float r2 = rng_uniform01(rng);
        const float rr1 = __builtin_sqrtf(r1);
        surface_pos = (a * (1.0f - rr1) + b * (rr1 * (1.0f - r2))) + c * (rr1 * r2);
        const float area = len(cross(b - a, c - a)) / 2.0f;
        surface_p = 1.0f / area;
        surface_cull = __float_as_uint(e2.z) != 0;
        surface_n = tb.tri_normal_at(object_index, ref, surface_pos);
    }

    const V3 to_light = surface_pos - pos;
    const V3 dir = normalize(to_light);
    const float abs_dot = __builtin_fabsf(dot(neg(dir), surface_n));
    if(!(abs_dot > 0.0f)) {
        return false;
    }
    if(!(len2(to_light) > 0.0f)) {
        return false;
    }
    if(surface_cull) {
        if(!(dot(dir, surface_n) < 0.0f)) {
            return false;
        }
    }
    const float conversion_factor = len2(to_light) / abs_dot;
    light_pos = surface_pos;
    spectrum = c4(e3);
    pd = selection_p * surface_p * conversion_factor;
    return true;
}

} // namespace ptd

#endif
