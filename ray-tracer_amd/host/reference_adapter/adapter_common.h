// Shared by the two replacement translation units of the reference-side adapter (bvh_interface.cpp, render_gpu.cpp in
// this directory): conversions from the reference's own types to the plain structs of the C ABI.  Compiles only against
// the reference's headers (src/common.h, src/scene.h, framework/mesh.h ...), which the reference's build provides.
#pragma once
#include "common.h" // reference src/common.h: Features, lights, HitInfo
#include "scene.h"  // reference src/scene.h: Scene

#include <rt_amd.h>

#include <cstring>
#include <variant>
#include <vector>

struct rt_bvh;
class BvhInterface;

namespace rt_adapter {

// the library handle behind a BvhInterface (defined in bvh_interface.cpp, which owns the pimpl type)
rt_bvh* handle_of(const BvhInterface& bvh);

inline void put3(float* dst, const glm::vec3& v)
{
    dst[0] = v.x;
    dst[1] = v.y;
    dst[2] = v.z;
}

// reference src/common.h:54-77 -> rt_features, member by member
inline rt_features to_rt_features(const Features& f)
{
    rt_features o;
    std::memset(&o, 0, sizeof o);
    o.shading = f.enableShading;
    o.recursive = f.enableRecursive;
    o.hard_shadow = f.enableHardShadow;
    o.soft_shadow = f.enableSoftShadow;
    o.normal_interp = f.enableNormalInterp;
    o.texture_mapping = f.enableTextureMapping;
    o.accel_structure = f.enableAccelStructure;
    o.environment_mapping = f.extra.enableEnvironmentMapping;
    o.bvh_sah_binning = f.extra.enableBvhSahBinning;
    o.motion_blur = f.extra.enableMotionBlur;
    o.bloom = f.extra.enableBloomEffect;
    o.bilinear_filtering = f.extra.enableBilinearTextureFiltering;
    o.mipmap_filtering = f.extra.enableMipmapTextureFiltering;
    o.multiple_rays = f.extra.enableMultipleRaysPerPixel;
    o.glossy = f.extra.enableGlossyReflection;
    o.transparency = f.extra.enableTransparency;
    o.depth_of_field = f.extra.enableDepthOfField;
    return o;
}

// reference src/scene.h:35 (std::variant of the three light structs, src/common.h:37-52) -> rt_light[]
inline std::vector<rt_light> to_rt_lights(const Scene& scene)
{
    std::vector<rt_light> out;
    for (const auto& l : scene.lights) {
        rt_light r;
        std::memset(&r, 0, sizeof r);
        if (std::holds_alternative<PointLight>(l)) {
            const PointLight& p = std::get<PointLight>(l);
            r.type = RT_LIGHT_POINT;
            put3(r.v, p.position);
            put3(r.v + 3, p.color);
        } else if (std::holds_alternative<SegmentLight>(l)) {
            const SegmentLight& s = std::get<SegmentLight>(l);
            r.type = RT_LIGHT_SEGMENT;
            put3(r.v, s.endpoint0);
            put3(r.v + 3, s.endpoint1);
            put3(r.v + 6, s.color0);
            put3(r.v + 9, s.color1);
        } else {
            const ParallelogramLight& p = std::get<ParallelogramLight>(l);
            r.type = RT_LIGHT_PARALLELOGRAM;
            put3(r.v, p.v0);
            put3(r.v + 3, p.edge01);
            put3(r.v + 6, p.edge02);
            put3(r.v + 9, p.color0);
            put3(r.v + 12, p.color1);
            put3(r.v + 15, p.color2);
            put3(r.v + 18, p.color3);
        }
        out.push_back(r);
    }
    return out;
}

} // namespace rt_adapter
