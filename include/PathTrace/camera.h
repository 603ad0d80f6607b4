// PathTrace/camera.h -- pinhole / thin-lens camera of the PathTrace API.
//
// The camera keeps its constructor arguments next to the derived frame: the renderer hands the arguments to the device
// library (pt_camera_params in include/pt_hip.h), which derives the same frame.  shootRay is the host-side ray generator for
// callers that want single rays.
#ifndef PATHTRACE_CAMERA_H
#define PATHTRACE_CAMERA_H

#include <PathTrace/base.h>

#include <memory>
#include <tuple>

class ApertureSampler {
  public:
    virtual ~ApertureSampler() = default;
    // point in [-1, 1]^2 of the aperture
    virtual std::tuple<float, float> sampleAperture(RandomEngine &re) const noexcept = 0;
};

// uniform on the unit disc
class CircularApertureSampler final : public ApertureSampler {
  public:
    std::tuple<float, float> sampleAperture(RandomEngine &re) const noexcept override;
};

// hexagon whose flat part spans horizontal_ratio of the half width (rejection sampling, then random mirroring)
class HexagonalApertureSampler final : public ApertureSampler {
  public:
    HexagonalApertureSampler(float horizontal_ratio) noexcept;
    std::tuple<float, float> sampleAperture(RandomEngine &re) const noexcept override;
    float getHorizontalRatio() const noexcept { return horizontal_ratio; }

  private:
    float horizontal_ratio;
};

class Camera {
  public:
    Camera(vec3<float> origin, vec3<float> look_at, vec3<float> up, float focal_length, float height, float aspect_ratio) noexcept;
    Camera(vec3<float> origin, vec3<float> look_at, vec3<float> up, float focal_length, float height, float aspect_ratio, float aperture_width,
           float aperture_height, std::unique_ptr<ApertureSampler> &&aperture_sampler, float focal_plane_dist = 0.0F) noexcept;

    // ray through sensor position (x, y) in [-1, 1]^2, jittered inside a pixel of the given size
    Ray shootRay(float x, float y, float pixel_width, float pixel_height, RandomEngine &re) const noexcept;

    // constructor arguments, as the device library wants them
    struct Parameters {
        vec3<float> origin, look_at, up;
        float focal_length, height, aspect_ratio, aperture_width, aperture_height;
        int aperture_kind; // 0 none, 1 circular, 2 hexagonal, -1 a user-defined sampler (not renderable on the device)
        float hex_ratio;
        float focal_plane_dist;
    };
    const Parameters &parameters() const noexcept { return params; }

  private:
    Parameters params;
    vec3<float> origin, forward, up, right;
    float aperture_width_half, aperture_height_half;
    std::unique_ptr<ApertureSampler> aperture_sampler;
    float focal_plane_dist;
};

#endif
