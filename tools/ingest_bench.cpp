// tools/ingest_bench.cpp -- times the scene-ingest path of the C++ API end to end: OBJ text ->
// io::loadMesh -> moveObjects -> Scene::Scene (flatten + device build) -> processJob.  Writes its own OBJ (a bumpy UV sphere of
// n x n quads, the stand-in of bench.py) to /tmp first.
#include <PathTrace/camera.h>
#include <PathTrace/scene/mesh.h>
#include <PathTrace/scene/scene.h>
#include <PathTrace/worker.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>

static double seconds_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 1900;
    const int spp = argc > 2 ? std::atoi(argv[2]) : 16;
    const std::string path = "/tmp/ingest_bench_" + std::to_string(n) + ".obj";
    auto t0 = std::chrono::steady_clock::now();
    {
        std::FILE *f = std::fopen(path.c_str(), "w");
        if(f == nullptr) {
            return 1;
        }
        for(int j = 0; j < n; j++) {
            const double v = 0.001 + (M_PI - 0.002) * j / (n - 1);
            for(int i = 0; i < n; i++) {
                const double u = 2.0 * M_PI * i / n;
                const double r = 40.0 * (1.0 + 0.2 * std::sin(5 * u) * std::sin(7 * v));
                std::fprintf(f, "v %.6f %.6f %.6f\n", r * std::sin(v) * std::cos(u), 50.0 + r * std::cos(v), r * std::sin(v) * std::sin(u));
            }
        }
        for(int j = 0; j + 1 < n; j++) {
            for(int i = 0; i < n; i++) {
                const int a = j * n + i + 1, b = j * n + (i + 1) % n + 1, c = a + n, d = b + n;
                std::fprintf(f, "f %d %d %d\nf %d %d %d\n", a, b, c, b, d, c);
            }
        }
        std::fclose(f);
    }
    std::printf("wrote %s in %.2f s\n", path.c_str(), seconds_since(t0));

    t0 = std::chrono::steady_clock::now();
    mat4<float> place{};
    place.rows[0][0] = place.rows[1][1] = place.rows[2][2] = 0.01F;
    place.rows[1][3] = -0.5F;
    place.rows[3][3] = 1.0F;
    auto mesh = io::loadMesh(std::filesystem::path(path), place, false, true);
    const double t_load = seconds_since(t0);

    t0 = std::chrono::steady_clock::now();
    std::vector<std::unique_ptr<Object>> objects;
    std::vector<std::unique_ptr<LightSource>> lights;
    auto glass = std::make_shared<ConstantMaterialHandler>(std::make_shared<ConstantMaterial>(Color<float>(1.0F, 1.0F, 1.0F, 1.0F), 1.5F), std::make_shared<GlassBDF>());
    for(auto &t : mesh) {
        t.setMaterialHandler(glass);
    }
    const size_t n_mesh = mesh.size();
    moveObjects(objects, mesh);
    auto walls = makeBox(vec3<float>{-1.0F, -1.0F, -1.0F}, vec3<float>{1.0F, 1.0F, 1.0F});
    moveObjects(objects, walls);
    auto lamp = makePlane(vec3<float>{-0.25F, 0.99F, -0.25F}, vec3<float>{0.25F, 0.99F, 0.25F}, true);
    auto glow = std::make_shared<ConstantMaterialHandler>(
      std::make_shared<ConstantMaterial>(Color<float>(1.0F, 1.0F, 1.0F, 1.0F), 1.0F, Spectrum(Color<float>{1.0F, 1.0F, 1.0F, 1.0F})), std::make_shared<LambertianBRDF>());
    for(auto &t : lamp) {
        t.setMaterialHandler(glow);
    }
    moveObjects(objects, lamp);
    const double t_move = seconds_since(t0);

    t0 = std::chrono::steady_clock::now();
    Scene scene(std::move(objects), std::move(lights));
    const double t_scene = seconds_since(t0);

    Camera camera(vec3<float>{0.0F, 0.0F, -3.0F}, vec3<float>{0.0F, 0.0F, 0.0F}, vec3<float>{0.0F, 1.0F, 0.0F}, 1.0F, 1.0F, -1.0F);
    RenderOptions options{1024, 1024, spp, spp, 1E-3F};
    FrameRenderJob job{camera, scene, options};
    t0 = std::chrono::steady_clock::now();
    Image<> image = processJob(job);
    const double t_render = seconds_since(t0);
    std::printf("%zu mesh triangles: loadMesh %.2f s, moveObjects %.2f s, Scene::Scene %.2f s, processJob 1024x1024x%d %.2f s (%.1f Msamples/s incl. workspace set-up)\n", n_mesh,
                t_load, t_move, t_scene, spp, t_render, 1024.0 * 1024.0 * spp / t_render / 1e6);
    std::remove(path.c_str());
    return image.getWidth() == 1024 ? 0 : 1;
}
