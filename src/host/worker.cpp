// src/host/worker.cpp -- processItem / processJob of PathTrace/worker.h on top of the C ABI (include/pt_hip.h).
#include <PathTrace/worker.h>

#include "../../include/pt_hip.h"

#include <algorithm>
#include <cstdlib>
#include <exception>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

namespace {

    pt_camera_params cameraParams(const Camera &camera) {
        const Camera::Parameters &p = camera.parameters();
        if(p.aperture_kind < 0) {
            throw std::invalid_argument("PathTrace: user-defined ApertureSampler classes cannot be rendered on the device");
        }
        pt_camera_params c{};
        for(int k = 0; k < 3; k++) {
            c.origin[k] = p.origin[k];
            c.look_at[k] = p.look_at[k];
            c.up[k] = p.up[k];
        }
        c.focal_length = p.focal_length;
        c.height = p.height;
        c.aspect_ratio = p.aspect_ratio;
        c.aperture_width = p.aperture_width;
        c.aperture_height = p.aperture_height;
        c.aperture_kind = p.aperture_kind;
        c.hex_ratio = p.hex_ratio;
        c.focal_plane_dist = p.focal_plane_dist;
        return c;
    }

    pt_options renderOptions(const RenderOptions &o) {
        return pt_options{o.image_width, o.image_height, o.min_sample_count, o.max_sample_count, o.epsilon};
    }

    void check(int status, const char *what) {
        if(status != PT_OK) {
            throw std::runtime_error(std::string("PathTrace: ") + what + " failed: " + pt_last_error());
        }
    }

} // namespace

WorkItem::WorkItem() noexcept : job(nullptr), offset_x(0), offset_y(0), width(0), height(0) {}

WorkItem::WorkItem(const FrameRenderJob *job, int offset_x, int offset_y, int width, int height) noexcept :
  job(job), offset_x(offset_x), offset_y(offset_y), width(width), height(height) {}

Image<> processItem(const WorkItem &item, RandomEngine &re) {
    Image<> tile(item.width, item.height);
    if(item.width <= 0 || item.height <= 0) {
        return tile;
    }
    const FrameRenderJob &job = *item.job;
    const pt_camera_params camera = cameraParams(job.camera);
    const pt_options options = renderOptions(job.options);
    const pt_stream stream{item.offset_x, item.offset_y, item.width, item.height, re.state()};

    // the device renders into a frame that lives in HBM only; the item's rectangle comes back straight into the tile
    static_assert(sizeof(Color<float>) == 4 * sizeof(float), "Image<Color<float>> is a packed RGBA float array");
    uint64_t state_after = stream.rng_state;
    check(pt_render_item(job.scene.deviceScene(), &camera, &options, &stream, reinterpret_cast<float *>(tile.data()), &state_after, nullptr), "processItem");
    re.setState(state_after);
    return tile;
}

Image<> processJob(const FrameRenderJob &job, const std::function<void(int, int)> &progress_callback, int worker_count) {
    const int width = std::max(job.options.image_width, 0);
    const int height = std::max(job.options.image_height, 0);
    Image<> frame(width, height);
    if(width == 0 || height == 0) {
        return frame;
    }
    const pt_camera_params camera = cameraParams(job.camera);
    const pt_options options = renderOptions(job.options);

    std::vector<pt_tile> tiles(pt_job_tiles(width, height, nullptr, 0));
    pt_job_tiles(width, height, tiles.data(), tiles.size());

    // one random base seed per call, like the reference's std::random_device-seeded workers; $PATHTRACE_SEED pins it
    uint64_t base_seed;
    if(const char *fixed = std::getenv("PATHTRACE_SEED")) {
        base_seed = std::strtoull(fixed, nullptr, 0);
    }
    else {
        std::random_device device;
        base_seed = (static_cast<uint64_t>(device()) << 32) | device();
    }

    // The tiles are dealt to the scene's device replicas ($PATHTRACE_DEVICES; one by default) and rendered by one persistent launch
    // per device.  progress_callback is called as the reference calls it (worker.h:75-78, src/worker.cpp:354-360): once per finished
    // tile, (completed, total), never concurrently -- while the devices are still rendering.
    // An exception thrown by the callback must not cross the C ABI (with several devices it would be thrown on a library thread and end
    // the program): it is kept, the remaining calls are skipped, and it is thrown again here once the devices have finished.
    struct Forward {
        const std::function<void(int, int)> *fn;
        std::exception_ptr failure;
    } forward{&progress_callback, nullptr};
    auto trampoline = [](int completed, int total, void *user) {
        Forward *f = static_cast<Forward *>(user);
        if(f->failure) {
            return;
        }
        try {
            (*f->fn)(completed, total);
        }
        catch(...) {
            f->failure = std::current_exception();
        }
    };
    static_assert(sizeof(Color<float>) == 4 * sizeof(float), "Image<Color<float>> is a packed RGBA float array");
    // worker_count (worker.h:83-84; threads in the reference, 0 = as many as the machine has): at most that many of the scene's device
    // replicas take part.  Every device runs one persistent launch, so there is nothing else for the count to choose.
    const std::vector<pt_scene *> &replicas = job.scene.deviceScenes();
    const int n_replicas = worker_count > 0 ? std::min(worker_count, static_cast<int>(replicas.size())) : static_cast<int>(replicas.size());
    const int status = pt_render_tiles_multi(replicas.data(), n_replicas, &camera, &options, tiles.data(), tiles.size(), base_seed,
                                             reinterpret_cast<float *>(frame.data()), nullptr, trampoline, &forward);
    if(forward.failure) {
        std::rethrow_exception(forward.failure);
    }
    check(status, "processJob");
    return frame;
}
