// Host-side scene copy, hierarchy construction in the reference's node order, and flattening into the
// device tables of rt_layout.h.
#pragma once
#include "rt_amd.h"
#include "rt_layout.h"

#include <string>
#include <vector>

namespace rt {

struct HostMesh {
    std::vector<float> vertices;     // 8 floats per vertex: position, normal, texcoord
    std::vector<uint32_t> triangles; // 3 indices per triangle
    MaterialRec material;
};

struct HostImage {
    int32_t width = 0, height = 0;
    std::vector<uint8_t> rgb;
};

struct HostScene {
    std::vector<HostMesh> meshes;
    std::vector<rt_sphere_desc> spheres;
    std::vector<HostImage> textures;
    std::vector<HostImage> skybox; // 0 or 6 faces
};

struct HostBvh {
    std::vector<rt_node_record> nodes; // reference order (DFS pre-order)
    std::vector<rt_prim_ref> prims;    // leaf primitive slots, leaf after leaf
    int32_t numLevels = 0, numLeaves = 0;
};

// Copies and validates a scene description.  Returns false and fills `error` on malformed input.
bool copy_scene(const rt_scene_desc& desc, HostScene& out, std::string& error);

// Builds the hierarchy exactly as the reference's constructor does (median split per level axis, or the
// SAH / binned-SAH variant), reference src/bounding_volume_hierarchy.cpp:41-208.
void build_bvh(const HostScene& scene, const rt_bvh_options& options, HostBvh& out);

struct FlatBvh {
    std::vector<NodeRec> nodes;
    std::vector<PrimRec> prims;
    std::vector<AttrRec> attrs;
    std::vector<uint32_t> primOrder;
    float rootLo[3], rootHi[3];
    int32_t rootRef = -1;
};
void flatten_bvh(const HostScene& scene, const HostBvh& bvh, FlatBvh& out);

} // namespace rt
