// PathTrace/worker.h -- render entry points of the PathTrace API, executed on an MI355X through include/pt_hip.h.
#ifndef PATHTRACE_WORKER_H
#define PATHTRACE_WORKER_H

#include <PathTrace/base.h>
#include <PathTrace/camera.h>
#include <PathTrace/image/image.h>
#include <PathTrace/scene/scene.h>

#include <functional>

struct RenderOptions {
    int image_width;
    int image_height;
    int min_sample_count; // samples every pixel gets at least
    int max_sample_count; // samples after which a pixel stops in any case
    float epsilon;        // ray offset / distance tolerance
    bool allow_bias = false; // accepted for source compatibility; has no effect (it has none in the reference either)
};

// what a render needs; the referenced objects must outlive the call
struct FrameRenderJob {
    const Camera &camera;
    const Scene &scene;
    const RenderOptions &options;
};

// a rectangle of the frame
struct WorkItem {
    const FrameRenderJob *job;
    int offset_x;
    int offset_y;
    int width;
    int height;

    WorkItem() noexcept;
    WorkItem(const FrameRenderJob *job, int offset_x, int offset_y, int width, int height) noexcept;
};

// Renders one WorkItem: its pixels in row-major order through the ONE engine `re`, exactly the sequence of draws a CPU
// worker makes; `re` is advanced.  Deterministic for a given engine state.  Throws std::runtime_error if the device fails.
Image<> processItem(const WorkItem &item, RandomEngine &re);

// Renders the whole frame.  All tiles are in flight on the GPU at once; every pixel has its own engine, seeded from one
// random base seed per call ($PATHTRACE_SEED fixes it).  progress_callback(completed, total) is called once per tile, in
// order, from the calling thread.  worker_count (threads in the reference) caps the number of device replicas of the scene that take part (0 = all of them).
Image<> processJob(
  const FrameRenderJob &job, const std::function<void(int, int)> &progress_callback = [](int, int) {}, int worker_count = 0);

#endif
