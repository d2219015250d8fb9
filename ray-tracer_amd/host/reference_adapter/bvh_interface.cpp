// Replacement for the reference's src/bvh_interface.cpp (the pimpl forwarders of src/bvh_interface.h:12-41) on top of
// librt_amd.so.  With this file in the reference's build, src/bounding_volume_hierarchy.cpp is no longer compiled: the
// hierarchy lives in the library (built in the reference's node order, kept in GPU memory).  The header stays as it is.
//
// Build: compile instead of src/bvh_interface.cpp and src/bounding_volume_hierarchy.cpp, add -I<librt_amd>/include and
// link -lrt_amd.
#include "bvh_interface.h" // reference src/bvh_interface.h, unmodified
#include "adapter_common.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <memory>

// Tunables the reference defines in src/bounding_volume_hierarchy.cpp:16-18 and its GUI edits (src/main.cpp); they move
// here with the file that owned them.
int extr_max_level = 32;
int extr_sah_bins = 64;
bool extr_debugSAH = false;

// The type behind BvhInterface::m_impl (the header only forward-declares it).
class BoundingVolumeHierarchy {
public:
    rt_scene* scene = nullptr;
    rt_bvh* bvh = nullptr;
    Scene* source = nullptr; // the reference keeps this pointer too and reads materials through it at hit time
    ~BoundingVolumeHierarchy()
    {
        rt_bvh_destroy(bvh);
        rt_scene_destroy(scene);
    }
};

namespace {

// An Image holds floats that are byte / 255.0f (reference framework/src/image.cpp:29-32); the library takes the bytes.
rt_image_desc to_bytes(const Image& img, std::vector<uint8_t>& store)
{
    store.resize(img.pixels.size() * 3);
    for (size_t i = 0; i < img.pixels.size(); i++)
        for (int c = 0; c < 3; c++) store[3 * i + size_t(c)] = uint8_t(std::lround(img.pixels[i][c] * 255.0f));
    return rt_image_desc { store.data(), img.width, img.height };
}

} // namespace

rt_bvh* rt_adapter::handle_of(const BvhInterface& bvh)
{
    // m_impl is private and the header must stay untouched: BvhInterface is a single pointer (src/bvh_interface.h:39-40)
    static_assert(sizeof(BvhInterface) == sizeof(BoundingVolumeHierarchy*), "BvhInterface is expected to hold only m_impl");
    BoundingVolumeHierarchy* impl;
    std::memcpy(&impl, &bvh, sizeof impl);
    return impl ? impl->bvh : nullptr;
}

BvhInterface::BvhInterface(Scene* pScene, const Features& features)
{
    m_impl = new BoundingVolumeHierarchy; // (never freed by the reference either: src/bvh_interface.cpp:6-9 has no destructor)
    m_impl->source = pScene;
    static_assert(sizeof(Vertex) == 32 && sizeof(glm::uvec3) == 12, "Vertex = 8 packed floats, triangle = 3 packed indices");
    std::vector<rt_mesh_desc> meshes(pScene->meshes.size());
    std::vector<const Image*> images; // distinct textures, in first-use order
    for (size_t i = 0; i < pScene->meshes.size(); i++) {
        const Mesh& m = pScene->meshes[i];
        rt_mesh_desc& d = meshes[i];
        d.vertices = m.vertices.empty() ? nullptr : &m.vertices[0].position.x;
        d.triangles = m.triangles.empty() ? nullptr : &m.triangles[0].x;
        d.n_vertices = uint32_t(m.vertices.size());
        d.n_triangles = uint32_t(m.triangles.size());
        rt_adapter::put3(d.kd, m.material.kd);
        rt_adapter::put3(d.ks, m.material.ks);
        d.shininess = m.material.shininess;
        d.transparency = m.material.transparency;
        d.texture = -1;
        if (m.material.kdTexture) {
            const Image* img = m.material.kdTexture.get();
            auto it = std::find(images.begin(), images.end(), img);
            if (it == images.end()) {
                images.push_back(img);
                d.texture = int32_t(images.size() - 1);
            } else {
                d.texture = int32_t(it - images.begin());
            }
        }
    }
    std::vector<std::vector<uint8_t>> bytes(images.size() + 6);
    std::vector<rt_image_desc> textures;
    for (size_t i = 0; i < images.size(); i++) textures.push_back(to_bytes(*images[i], bytes[i]));
    std::vector<rt_image_desc> sky;
    if (pScene->skybox.size() == 6)
        for (size_t f = 0; f < 6; f++) sky.push_back(to_bytes(pScene->skybox[f], bytes[images.size() + f]));
    std::vector<rt_sphere_desc> spheres;
    for (const Sphere& s : pScene->spheres) {
        rt_sphere_desc d;
        rt_adapter::put3(d.center, s.center);
        d.radius = s.radius;
        rt_adapter::put3(d.kd, s.material.kd);
        rt_adapter::put3(d.ks, s.material.ks);
        d.shininess = s.material.shininess;
        d.transparency = s.material.transparency;
        spheres.push_back(d);
    }
    rt_scene_desc desc;
    desc.meshes = meshes.data();
    desc.n_meshes = uint32_t(meshes.size());
    desc.textures = textures.data();
    desc.n_textures = uint32_t(textures.size());
    desc.spheres = spheres.data();
    desc.n_spheres = uint32_t(spheres.size());
    desc.skybox = sky.empty() ? nullptr : sky.data();
    if (rt_scene_create(&desc, &m_impl->scene) != RT_OK) {
        std::fprintf(stderr, "BvhInterface: %s\n", rt_last_error());
        return;
    }
    const rt_bvh_options options { features.extra.enableBvhSahBinning ? 1 : 0, extr_max_level, extr_sah_bins };
    if (rt_bvh_build(m_impl->scene, &options, &m_impl->bvh) != RT_OK) std::fprintf(stderr, "BvhInterface: %s\n", rt_last_error());
}

int BvhInterface::numLevels() const { return m_impl->bvh ? rt_bvh_num_levels(m_impl->bvh) : 0; }
int BvhInterface::numLeaves() const { return m_impl->bvh ? rt_bvh_num_leaves(m_impl->bvh) : 0; }

// The reference draws node boxes with OpenGL here (src/bounding_volume_hierarchy.cpp:232-290).  A GUI build walks
// rt_bvh_export() and calls its own drawAABB; the command-line renderer never reaches these.
void BvhInterface::debugDrawLevel(int) {}
void BvhInterface::debugDrawLeaf(int) {}

// One closest-hit query (reference src/bounding_volume_hierarchy.cpp:459-548) through the batch entry point.  This is
// the path of the GUI's single debug ray (src/main.cpp:158); frames go through renderRayTracing, which never calls it.
bool BvhInterface::intersect(Ray& ray, HitInfo& hitInfo, const Features& features) const
{
    if (!m_impl->bvh) return false;
    rt_ray r;
    rt_adapter::put3(r.origin, ray.origin);
    rt_adapter::put3(r.direction, ray.direction);
    r.t = ray.t;
    const rt_features f = rt_adapter::to_rt_features(features);
    rt_hit h;
    if (rt_intersect_batch(m_impl->bvh, &r, 1, &f, &h, 0, nullptr) != RT_OK) {
        std::fprintf(stderr, "BvhInterface::intersect: %s\n", rt_last_error());
        return false;
    }
    if (!h.hit) return false;
    ray.t = h.t;
    hitInfo.normal = glm::vec3(h.normal[0], h.normal[1], h.normal[2]);
    hitInfo.barycentricCoord = glm::vec3(h.bary[0], h.bary[1], h.bary[2]);
    hitInfo.material = h.mesh >= 0 ? m_impl->source->meshes[size_t(h.mesh)].material : m_impl->source->spheres[size_t(h.triangle)].material;
    hitInfo.material.kd = glm::vec3(h.kd[0], h.kd[1], h.kd[2]); // the texel when texture mapping applies, else the material's kd
    return true;
}
