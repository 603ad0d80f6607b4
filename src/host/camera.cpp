// src/host/camera.cpp -- Camera and the aperture samplers of PathTrace/camera.h (host side; the renderer generates its camera
// rays on the device from Camera::parameters()).
#include <PathTrace/camera.h>

#include <algorithm>
#include <cmath>
#include <random>

namespace {
    constexpr float kPi = static_cast<float>(M_PI);
}

std::tuple<float, float> CircularApertureSampler::sampleAperture(RandomEngine &re) const noexcept {
    std::uniform_real_distribution<float> dist(0, 1);
    const float radius = std::sqrt(dist(re));
    const float angle = 2 * kPi * dist(re);
    return {radius * std::cos(angle), radius * std::sin(angle)};
}

HexagonalApertureSampler::HexagonalApertureSampler(float horizontal_ratio) noexcept : horizontal_ratio(std::min(std::max(horizontal_ratio, 0.0F), 1.0F)) {}

std::tuple<float, float> HexagonalApertureSampler::sampleAperture(RandomEngine &re) const noexcept {
    std::uniform_real_distribution<float> dist(0, 1);
    std::bernoulli_distribution mirror;
    float x, y;
    for(;;) { // rejection sampling in the first quadrant
        x = dist(re);
        y = dist(re);
        const float beyond_flat = x - horizontal_ratio;
        if(beyond_flat <= 0.0F || (beyond_flat / (1.0F - horizontal_ratio)) >= y) {
            break;
        }
    }
    if(mirror(re)) {
        x = -x;
    }
    if(mirror(re)) {
        y = -y;
    }
    return {x, y};
}

Camera::Camera(vec3<float> origin, vec3<float> look_at, vec3<float> up, float focal_length, float height, float aspect_ratio) noexcept :
  Camera(origin, look_at, up, focal_length, height, aspect_ratio, 0.0F, 0.0F, nullptr, 0.0F) {}

Camera::Camera(vec3<float> origin_, vec3<float> look_at, vec3<float> up_, float focal_length, float height, float aspect_ratio, float aperture_width,
               float aperture_height, std::unique_ptr<ApertureSampler> &&sampler, float focal_plane_dist_) noexcept {
    params = Parameters{origin_, look_at, up_, focal_length, height, aspect_ratio, aperture_width, aperture_height, 0, 0.0F, focal_plane_dist_};
    if(sampler != nullptr) {
        if(dynamic_cast<const CircularApertureSampler *>(sampler.get()) != nullptr) {
            params.aperture_kind = 1;
        }
        else if(const auto *hex = dynamic_cast<const HexagonalApertureSampler *>(sampler.get())) {
            params.aperture_kind = 2;
            params.hex_ratio = hex->getHorizontalRatio();
        }
        else {
            params.aperture_kind = -1;
        }
    }
    origin = origin_;
    forward = (look_at - origin_).normalize() * focal_length;
    const float half_height = height / 2.0F;
    up = up_.normalize() * half_height;
    right = cross(forward, up).normalize() * (half_height * aspect_ratio);
    aperture_width_half = aperture_width / 2.0F;
    aperture_height_half = aperture_height / 2.0F;
    aperture_sampler = std::move(sampler);
    focal_plane_dist = focal_plane_dist_;
}

Ray Camera::shootRay(float x, float y, float pixel_width, float pixel_height, RandomEngine &re) const noexcept {
    std::uniform_real_distribution<float> jitter_x(-pixel_width / 2.0F, pixel_width / 2.0F);
    std::uniform_real_distribution<float> jitter_y(-pixel_height / 2.0F, pixel_height / 2.0F);
    const float dx = jitter_x(re);
    const float dy = jitter_y(re);
    const vec3<float> sensor = origin - forward - up * (y + dy) - right * (x + dx);

    float lens_x = 0.0F, lens_y = 0.0F;
    if(aperture_sampler) {
        const auto [sx, sy] = aperture_sampler->sampleAperture(re);
        lens_x = sx * aperture_width_half;
        lens_y = sy * aperture_height_half;
    }
    const vec3<float> start = origin + up * lens_x + right * lens_y;
    if(focal_plane_dist > 0.0F) {
        // thin lens: all rays of a sensor point meet on the focal plane
        const vec3<float> chief = (origin - sensor).normalize();
        const vec3<float> focus = origin + chief * (focal_plane_dist / dot(forward, chief));
        return {start, vec3<float>((focus - start).normalize())};
    }
    return {start, vec3<float>((start - sensor).normalize())};
}
