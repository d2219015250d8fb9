// renderRayTracing (reference src/render.h:17, body at src/render.cpp:82-128) on top of librt_amd.so: the whole frame in
// one rt_render call, bloom in one rt_bloom call.  In the reference's build this definition takes the place of the one in
// src/render.cpp; getFinalColor (the GUI's debug-ray path, src/render.cpp:18-80) stays where it is and reaches the
// library one ray at a time through BvhInterface::intersect.
#include "render.h" // reference src/render.h, unmodified
#include "adapter_common.h"
#include "bloom.h"
#include "bvh_interface.h"
#include "dof.h"
#include "environment_mapping.h"
#include "gloss.h"
#include "light.h"
#include "multipleRays.h"
#include "screen.h"
#include <framework/trackball.h>

#include <iostream>

namespace {

// Trackball keeps its field of view private and offers no getter (framework/include/framework/trackball.h:49); a
// maintainer would add one.  Against the unmodified header the value is reached through an explicit template
// instantiation, where the language does not check access.
template <typename Tag, typename Tag::type Member> struct Expose {
    friend typename Tag::type member_of(Tag) { return Member; }
};
struct TrackballFovy {
    typedef float Trackball::*type;
    friend type member_of(TrackballFovy);
};
template struct Expose<TrackballFovy, &Trackball::m_fovy>;

} // namespace

void renderRayTracing(const Scene& scene, const Trackball& camera, const BvhInterface& bvh, Screen& screen, const Features& features)
{
    rt_bvh* handle = rt_adapter::handle_of(bvh);
    if (!handle) return;
    const glm::ivec2 res = screen.resolution();

    rt_camera cam;
    cam.fovy = camera.*member_of(TrackballFovy());
    cam.aspect = float(res.x) / float(res.y); // Window::getAspectRatio of a window of the screen's size (framework/src/window.cpp:379-384)
    rt_adapter::put3(cam.look_at, camera.lookAt());
    cam.distance = camera.distanceFromLookAt();
    rt_adapter::put3(cam.rotation, camera.rotationEulerAngles());

    // the reference reads its tunable globals while it renders; they are snapshotted here, once per frame
    rt_params p;
    rt_default_params(&p);
    p.samples_per_unit = samplesPerUnit;                  // src/light.cpp:11
    p.samples_per_unit_parallel = samplesPerUnitParallel; // src/light.cpp:16
    p.ray_multiplier = rayMultiplier;                     // src/multipleRays.cpp:11
    p.glossy_filter_size = extr_glossy_filterSize;        // src/gloss.cpp:9-10
    p.glossy_sigma = extr_glossy_sigma;
    p.dof_samples = extr_dof_samples;                     // src/dof.cpp:9-12
    p.dof_aperture = extr_dof_aperture;
    p.dof_f = extr_dof_f;
    p.dof = extr_dof;
    p.bloom_threshold = threshold;                        // src/bloom.cpp:5-11
    p.bloom_size = bloomsize;
    p.bloom_sigma = sigma;
    p.bloom_debug = debugBloom;
    p.bloom_gaussian = gaussian;
    p.bloom_scale = scale;
    p.enabled_skybox = extr_enabledSkyBox;                // src/environment_mapping.cpp:5-6
    p.enabled_refl_map = extr_enabledReflMap;

    const rt_features f = rt_adapter::to_rt_features(features);
    const std::vector<rt_light> lights = rt_adapter::to_rt_lights(scene);
    static_assert(sizeof(glm::vec3) == 12, "Screen::pixels() must be packed float RGB");
    float* out = &screen.pixels()[0].x; // Screen order: top row first (src/screen.cpp:37-43), which is what rt_render writes
    if (rt_render(handle, &cam, lights.data(), int32_t(lights.size()), &f, &p, res.x, res.y, 0, res.y, out, 0, nullptr, nullptr) != RT_OK) {
        std::cerr << "renderRayTracing: " << rt_last_error() << std::endl;
        return;
    }
    if (features.extra.enableBloomEffect && rt_bloom(out, res.x, res.y, &p, 0, nullptr) != RT_OK)
        std::cerr << "renderRayTracing (bloom): " << rt_last_error() << std::endl;
}
