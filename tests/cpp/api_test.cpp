// tests/cpp/api_test.cpp -- exercises the reference-compatible C++ API (include/PathTrace) end to end on the GPU.
// The checks restate, in this project's own words, what the reference's test programs check for the hot path
// (test/render_test.cpp, test/scene/scene_test.cpp, test/scene/boundig_box_test.cpp), and add what only this
// implementation can promise: processItem is deterministic per engine state and advances the engine, processJob is
// reproducible under PATHTRACE_SEED, progress is reported once per tile in order, unsupported subclasses are refused.
#include <PathTrace/camera.h>
#include <PathTrace/post_processing.h>
#include <PathTrace/scene/light.h>
#include <PathTrace/scene/mesh.h>
#include <PathTrace/scene/object.h>
#include <PathTrace/scene/scene.h>
#include <PathTrace/worker.h>

#include <gmock/gmock.h>
#include <gtest/gtest.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <thread>

namespace {

    using Objects = std::vector<std::unique_ptr<Object>>;
    using Lights = std::vector<std::unique_ptr<LightSource>>;

    Scene boxScene() {
        Objects objects;
        Lights lights;
        auto walls = makeBox(vec3<float>{-1.0F, -1.0F, -1.0F}, vec3<float>{1.0F, 1.0F, 1.0F});
        moveObjects(objects, walls);
        auto lamp = makePlane(vec3<float>{-0.25F, 0.99F, -0.25F}, vec3<float>{0.25F, 0.99F, 0.25F});
        auto glow = std::make_shared<ConstantMaterial>(Color<float>(1.0F, 1.0F, 1.0F, 1.0F), 1.0F, Spectrum(Color<float>{1.0F, 1.0F, 1.0F, 1.0F}));
        auto handler = std::make_shared<ConstantMaterialHandler>(glow, std::make_shared<LambertianBRDF>());
        for(auto &t : lamp) {
            t.setMaterialHandler(handler);
        }
        moveObjects(objects, lamp);
        return Scene(std::move(objects), std::move(lights));
    }

    bool sameBits(const Image<> &a, const Image<> &b) {
        return a.getWidth() == b.getWidth() && a.getHeight() == b.getHeight() && std::memcmp(a.data(), b.data(), a.size() * sizeof(Color<float>)) == 0;
    }

} // namespace

TEST(Render, EmptySceneIsExactlyTransparentBlack) {
    Camera camera({0.0F, 0.0F, 0.0F}, {0.0F, 0.0F, 1.0F}, {0.0F, 1.0F, 0.0F}, 1.0F, 1.0F, 1.0F);
    Scene scene(Objects{}, Lights{});
    RenderOptions options{1, 1, 1, 1, 1E-3F};
    FrameRenderJob job{camera, scene, options};
    auto frame = processJob(job);
    EXPECT_THAT(frame(0, 0), testing::Eq(Color<float>{0.0F, 0.0F, 0.0F, 0.0F}));
}

TEST(Render, LitSphereCoversTheCentreOnly) {
    Camera camera({0.0F, 0.0F, 0.0F}, {0.0F, 0.0F, 1.0F}, {0.0F, 1.0F, 0.0F}, 0.1F, 1.0F, 1.0F);
    Objects objects;
    Lights lights;
    lights.emplace_back(std::make_unique<PointLightSource>(vec3<float>{0.0F, 1.0F, 0.0F}, Color<float>{1.0F, 1.0F, 1.0F, 1.0F}));
    objects.emplace_back(std::make_unique<Sphere>(vec3<float>{0.0F, 0.0F, 0.6F}, 0.5F));
    Scene scene(std::move(objects), std::move(lights));
    RenderOptions options{16, 16, 2, 2, 1E-3F};
    FrameRenderJob job{camera, scene, options};
    auto frame = processJob(job);
    EXPECT_THAT(frame(0, 0), testing::Eq(Color<float>{0.0F, 0.0F, 0.0F, 0.0F}));
    EXPECT_THAT(frame(8, 8)[3], testing::Gt(0.0F));
}

TEST(Render, NonSquareFrameWithClippedTilesAndAdaptiveSampling) {
    Camera camera({0.0F, 0.0F, 0.0F}, {0.0F, 0.0F, 1.0F}, {0.0F, 1.0F, 0.0F}, 0.2F, 0.5F, 1.94F);
    Objects objects;
    Lights lights;
    lights.emplace_back(std::make_unique<PointLightSource>(vec3<float>{0.0F, 1.0F, 0.0F}, Color<float>{1.0F, 1.0F, 1.0F, 1.0F}));
    auto diffuse = std::make_shared<LambertianBRDF>();
    auto glass = std::make_shared<GlassBDF>();
    auto s1 = std::make_unique<Sphere>(vec3<float>{0.1F, 0.1F, 1.0F}, 0.5F);
    s1->setMaterialHandler(std::make_shared<ConstantMaterialHandler>(std::make_shared<ConstantMaterial>(Color<float>(1.0F, 1.0F, 1.0F, 1.5F)), glass));
    objects.emplace_back(std::move(s1));
    auto s2 = std::make_unique<Sphere>(vec3<float>{-0.1F, 0.2F, 2.0F}, 0.6F);
    s2->setMaterialHandler(std::make_shared<ConstantMaterialHandler>(
      std::make_shared<ConstantMaterial>(Color<float>(0.8F, 0.4F, 0.6F, 1.0F), 1.0F, Spectrum(Color<float>{0.2F, 0.1F, 0.3F, 1.0F})), diffuse));
    objects.emplace_back(std::move(s2));
    auto floor = std::make_unique<Triangle>(vec3<float>{5.0F, -1.0F, 5.0F}, vec3<float>{0.0F, -1.0F, -5.0F}, vec3<float>{-5.0F, -1.0F, 5.0F});
    floor->setMaterialHandler(std::make_shared<ConstantMaterialHandler>(std::make_shared<ConstantMaterial>(Color<float>(0.4F, 0.6F, 0.4F, 1.0F)), diffuse));
    objects.emplace_back(std::move(floor));
    Scene scene(std::move(objects), std::move(lights));
    RenderOptions options{132, 68, 5, 10, 1E-3F};
    FrameRenderJob job{camera, scene, options};

    int calls = 0, last = 0, total_seen = 0;
    auto frame = processJob(job, [&](int done, int total) {
        calls++;
        EXPECT_THAT(done, testing::Eq(last + 1));
        last = done;
        total_seen = total;
    });
    EXPECT_THAT(calls, testing::Eq(8 * 4)); // 17-pixel tiles: 8 x 4
    EXPECT_THAT(total_seen, testing::Eq(32));
    EXPECT_THAT(frame(0, 0), testing::Eq(Color<float>{0.0F, 0.0F, 0.0F, 0.0F}));
    EXPECT_THAT(frame(64, 32)[3], testing::Gt(0.0F));
}

TEST(Scene, ClosestObjectIsReported) {
    Objects objects;
    const Sphere first(vec3<float>(-1.0F, -1.0F, -1.0F), 1.0F), second(vec3<float>(1.0F, 1.0F, 1.0F), 1.0F);
    objects.push_back(std::make_unique<Sphere>(first));
    objects.push_back(std::make_unique<Sphere>(second));
    Scene scene(std::move(objects), Lights{});
    {
        auto [t, object] = scene.getIntersection(Ray{vec3<float>(-0.5F, -0.5F, -5.0F), vec3<float>(0.0F, 0.0F, 1.0F)});
        EXPECT_THAT(t, testing::Ge(0.0F));
        EXPECT_THAT(object, testing::NotNull());
        if(object != nullptr) {
            EXPECT_THAT(object->getBoundingVolume().low, testing::Eq(first.getBoundingVolume().low));
        }
    }
    {
        auto [t, object] = scene.getIntersection(Ray{vec3<float>(0.5F, 0.5F, -5.0F), vec3<float>(0.0F, 0.0F, 1.0F)});
        EXPECT_THAT(t, testing::Ge(0.0F));
        EXPECT_THAT(object, testing::NotNull());
        if(object != nullptr) {
            EXPECT_THAT(object->getBoundingVolume().high, testing::Eq(second.getBoundingVolume().high));
        }
    }
    {
        auto [t, object] = scene.getIntersection(Ray{vec3<float>(0.0F, 0.0F, 0.0F), vec3<float>(0.0F, 0.0F, 1.0F)});
        EXPECT_THAT(t, testing::Lt(0.0F));
        EXPECT_THAT(object == nullptr, testing::Eq(true));
    }
}

TEST(Box, SlabTestKnownAnswers) {
    const Sphere unit(vec3<float>(0.0F, 0.0F, 0.0F), 1.0F);
    const AABB box(unit.getBoundingVolume(), std::make_unique<Sphere>(unit));
    EXPECT_THAT(box.getIntersection(Ray{vec3<float>(-5.0F, 0.0F, 0.0F), vec3<float>(1.0F, 0.0F, 0.0F)}), testing::FloatEq(4.0F));
    EXPECT_THAT(box.getIntersection(Ray{vec3<float>(0.0F, 0.5F, 0.0F), vec3<float>(0.0F, -1.0F, 0.0F)}), testing::FloatEq(0.0F));
    EXPECT_THAT(box.getIntersection(Ray{vec3<float>(0.0F, 0.0F, -5.0F), vec3<float>(0.0F, 0.0F, -1.0F)}), testing::Lt(0.0F));
    const auto diagonal = vec3<float>(vec3<float>(1.0F, 1.0F, 0.0F).normalize());
    EXPECT_THAT(box.getIntersection(Ray{vec3<float>(-1.5F, 0.0F, 0.0F), diagonal}), testing::FloatEq(std::sqrt(2.0F) / 2.0F));
}

TEST(WorkItem, ProcessItemIsDeterministicAndAdvancesItsEngine) {
    Scene scene = boxScene();
    Camera camera({0.0F, 0.0F, -3.0F}, {0.0F, 0.0F, 0.0F}, {0.0F, 1.0F, 0.0F}, 1.0F, 1.0F, -1.0F);
    RenderOptions options{64, 64, 8, 8, 1E-3F};
    FrameRenderJob job{camera, scene, options};
    WorkItem item(&job, 16, 24, 6, 5);

    RandomEngine first(1234), second(1234), other(99);
    const auto a = processItem(item, first);
    const auto b = processItem(item, second);
    const auto c = processItem(item, other);
    EXPECT_THAT(a.getWidth(), testing::Eq(6));
    EXPECT_THAT(a.getHeight(), testing::Eq(5));
    EXPECT_THAT(sameBits(a, b), testing::Eq(true));
    EXPECT_THAT(sameBits(a, c), testing::Eq(false));
    EXPECT_THAT(first.state(), testing::Eq(second.state()));
    EXPECT_THAT(first.state() != RandomEngine(1234).state(), testing::Eq(true));
    // the engine continues where the item left it: a second item through the same engine differs from a fresh one
    const auto d = processItem(item, first);
    EXPECT_THAT(sameBits(a, d), testing::Eq(false));
    EXPECT_THAT(a(3, 2)[3], testing::Eq(1.0F)); // inside the closed box every path hits something
}

TEST(WorkItem, ProcessJobIsReproducibleUnderAFixedSeed) {
    Scene scene = boxScene();
    Camera camera({0.0F, 0.0F, -3.0F}, {0.0F, 0.0F, 0.0F}, {0.0F, 1.0F, 0.0F}, 1.0F, 1.0F, -1.0F);
    RenderOptions options{48, 40, 4, 16, 1E-3F};
    FrameRenderJob job{camera, scene, options};
    setenv("PATHTRACE_SEED", "77", 1);
    const auto a = processJob(job);
    const auto b = processJob(job, [](int, int) {}, 3);
    setenv("PATHTRACE_SEED", "78", 1);
    const auto c = processJob(job);
    unsetenv("PATHTRACE_SEED");
    EXPECT_THAT(sameBits(a, b), testing::Eq(true));
    EXPECT_THAT(sameBits(a, c), testing::Eq(false));
}

TEST(WorkItem, ProgressIsReportedTileByTileWhileTheDeviceRenders) {
    // worker.h:75-78 / src/worker.cpp:354-360: (completed, total) once per finished tile, completed = 1 .. total in order, never
    // concurrently.  The first report must arrive well before the last one: tiles are reported as they finish, not after the frame.
    Scene scene = boxScene();
    Camera camera({0.0F, 0.0F, -3.0F}, {0.0F, 0.0F, 0.0F}, {0.0F, 1.0F, 0.0F}, 1.0F, 1.0F, -1.0F);
    RenderOptions options{256, 192, 256, 256, 1E-3F};
    FrameRenderJob job{camera, scene, options};
    std::vector<int> seen;
    std::vector<double> when;
    int total_seen = -1, inside = 0, overlaps = 0;
    const auto t0 = std::chrono::steady_clock::now();
    const auto frame = processJob(job, [&](int completed, int total) {
        overlaps += inside++ != 0 ? 1 : 0;
        seen.push_back(completed);
        total_seen = total;
        when.push_back(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        inside--;
    });
    const double all = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    EXPECT_THAT(total_seen, testing::Eq(8 * 6)); // 256 x 192 in 32 x 32 tiles (worker.cpp:398-414)
    EXPECT_THAT(static_cast<int>(seen.size()), testing::Eq(total_seen));
    bool in_order = true;
    for(size_t i = 0; i < seen.size(); i++) {
        in_order = in_order && seen[i] == static_cast<int>(i) + 1;
    }
    EXPECT_THAT(in_order, testing::Eq(true));
    EXPECT_THAT(overlaps, testing::Eq(0));
    EXPECT_THAT(when.front() < 0.9 * all, testing::Eq(true));
    EXPECT_THAT(frame(128, 96)[3], testing::Eq(1.0F));
}

TEST(WorkItem, ThreadsMayShareOneScene) {
    // the reference's workers call processItem concurrently on one const Scene, each with its own engine (src/worker.cpp:328-362)
    Scene scene = boxScene();
    Camera camera({0.0F, 0.0F, -3.0F}, {0.0F, 0.0F, 0.0F}, {0.0F, 1.0F, 0.0F}, 1.0F, 1.0F, -1.0F);
    RenderOptions options{64, 64, 8, 8, 1E-3F};
    FrameRenderJob job{camera, scene, options};
    std::vector<Image<>> alone, together(4);
    for(int k = 0; k < 4; k++) {
        RandomEngine re(100 + static_cast<uint64_t>(k));
        alone.push_back(processItem(WorkItem(&job, 8 * k, 4 * k, 16, 16), re));
    }
    std::vector<std::thread> threads;
    for(int k = 0; k < 4; k++) {
        threads.emplace_back([&, k]() {
            for(int round = 0; round < 3; round++) {
                RandomEngine re(100 + static_cast<uint64_t>(k));
                together[static_cast<size_t>(k)] = processItem(WorkItem(&job, 8 * k, 4 * k, 16, 16), re);
            }
        });
    }
    for(auto &t : threads) {
        t.join();
    }
    for(size_t k = 0; k < 4; k++) {
        EXPECT_THAT(sameBits(alone[k], together[k]), testing::Eq(true));
    }
}

TEST(WorkItem, DeviceReplicasRenderTheSameFrame) {
    // $PATHTRACE_DEVICES: the scene is replicated and processJob deals the tiles out (here: two replicas on the one device of a test box)
    Camera camera({0.0F, 0.0F, -3.0F}, {0.0F, 0.0F, 0.0F}, {0.0F, 1.0F, 0.0F}, 1.0F, 1.0F, -1.0F);
    RenderOptions options{96, 80, 8, 8, 1E-3F};
    setenv("PATHTRACE_SEED", "5", 1);
    Image<> one(0, 0), two(0, 0);
    {
        Scene scene = boxScene();
        FrameRenderJob job{camera, scene, options};
        one = processJob(job);
    }
    setenv("PATHTRACE_DEVICES", "1", 1);
    setenv("PATHTRACE_REPLICAS_SHARE_DEVICE", "2", 1);
    {
        Scene scene = boxScene();
        EXPECT_THAT(static_cast<int>(scene.deviceScenes().size()), testing::Eq(2));
        FrameRenderJob job{camera, scene, options};
        int calls = 0;
        two = processJob(job, [&](int, int) { calls++; });
        EXPECT_THAT(calls, testing::Eq(5 * 4)); // 96 x 80 in 20 x 20 tiles (worker.cpp:398-414)
    }
    unsetenv("PATHTRACE_DEVICES");
    unsetenv("PATHTRACE_REPLICAS_SHARE_DEVICE");
    unsetenv("PATHTRACE_SEED");
    EXPECT_THAT(sameBits(one, two), testing::Eq(true));
}

namespace {
    class Blob final : public Object {
      public:
        float getIntersection(const Ray &) const noexcept override { return -1.0F; }
        vec3<float> getSurfaceNormal(vec3<float>) const noexcept override { return {0.0F, 1.0F, 0.0F}; }
        AABBArea getBoundingVolume() const noexcept override { return {}; }
    };
}

TEST(Scene, UserDefinedObjectsAreRefusedNotMisrendered) {
    Objects objects;
    objects.emplace_back(std::make_unique<Blob>());
    bool refused = false;
    try {
        Scene scene(std::move(objects), Lights{});
    }
    catch(const std::invalid_argument &) {
        refused = true;
    }
    EXPECT_THAT(refused, testing::Eq(true));
}

// post-processing runs on the device; what must hold for any frame (post_processing.h:6-12,16-22): sizes are kept, alpha is
// untouched, tone mapping lands every peak channel in [0, 1] and keeps the order of the brightness heuristic, gamma 1 is the identity
TEST(Render, PostProcessingOnTheDevice) {
    Image<> frame(40, 30);
    uint64_t state = 0x9E3779B97F4A7C15ULL;
    for(int y = 0; y < frame.getHeight(); y++) {
        for(int x = 0; x < frame.getWidth(); x++) {
            float channel[3];
            for(float &c : channel) {
                state ^= state << 13;
                state ^= state >> 7;
                state ^= state << 17;
                c = 0.001F + static_cast<float>(state >> 40) / 16777216.0F * static_cast<float>(1 + (x * 7 + y) % 50);
            }
            frame(x, y) = Color<float>(channel[0], channel[1], channel[2], 1.0F);
        }
    }
    Image<> same = frame;
    gammaCorrect(same, 1.0F);
    EXPECT_THAT(sameBits(same, frame), testing::Eq(true));

    Image<> mapped = frame;
    toneMap(mapped);
    EXPECT_THAT(mapped.getWidth(), testing::Eq(frame.getWidth()));
    EXPECT_THAT(mapped.getHeight(), testing::Eq(frame.getHeight()));
    auto peak = [](const Color<float> &c) { return std::max({c[0], c[1], c[2]}); };
    auto heuristic = [&](const Color<float> &c) { return c[3] * ((c[0] + c[1] + c[2]) / 3.0F + peak(c)) / 2.0F; };
    int out_of_range = 0, order_violations = 0;
    for(int y = 0; y < frame.getHeight(); y++) {
        for(int x = 0; x < frame.getWidth(); x++) {
            const float p = peak(mapped(x, y));
            out_of_range += !(p >= 0.0F && p <= 1.0F);
            out_of_range += mapped(x, y)[3] != 1.0F;
            if(x > 0) {
                const bool brighter_before = heuristic(frame(x, y)) > heuristic(frame(x - 1, y));
                const bool brighter_after = peak(mapped(x, y)) >= peak(mapped(x - 1, y));
                order_violations += brighter_before && !brighter_after;
            }
        }
    }
    EXPECT_THAT(out_of_range, testing::Eq(0));
    EXPECT_THAT(order_violations, testing::Eq(0));

    Image<> both = frame, stepwise = frame;
    postProcess(both);
    toneMap(stepwise);
    gammaCorrect(stepwise); // default gamma, 1.8
    EXPECT_THAT(sameBits(both, stepwise), testing::Eq(true));
}

int main(int argc, char *argv[]) {
    testing::InitGoogleTest(&argc, argv);
    return RUN_ALL_TESTS();
}
