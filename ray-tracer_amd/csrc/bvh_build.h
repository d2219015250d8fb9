// Host-side scene copy, hierarchy construction in the reference's node order, and flattening into the
// device tables of rt_layout.h.
#pragma once
#include "rt_amd.h"
#include "rt_layout.h"

#include <hip/hip_runtime_api.h>

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

// (bvh_device.hip) The same hierarchy built on the current GPU, tables left in device memory.  Pointers are hipMalloc'd
// and owned by the struct's holder (release() frees what is still set).
struct DeviceBvh {
    NodeRec* nodes = nullptr;
    PrimRec* prims = nullptr;
    AttrRec* attrs = nullptr;
    uint32_t* primOrder = nullptr;
    rt_node_record* refNodes = nullptr; // the reference-order node list (what HostBvh::nodes holds after a host build)
    rt_prim_ref* refPrims = nullptr;    // ... and its leaf primitive slots
    size_t numInner = 0, numPrims = 0, numRefNodes = 0;
    float rootLo[3] = { 0, 0, 0 }, rootHi[3] = { 0, 0, 0 };
    int32_t rootRef = -1, numLevels = 0, numLeaves = 0;
    void release();
};
// Median-split build of the reference (src/bounding_volume_hierarchy.cpp:41-208) on the current GPU, node for node the
// tree build_bvh() makes: primitive boxes, the per-level std::sort calls reproduced move for move (including where
// equal keys end up), node numbering, boxes and the device tables.  `unsupported` is set (and nothing built) for the
// SAH variant, fewer than two primitives, a NaN centroid, leaves of more than 64 primitives (level limit), or a sort
// that reaches introsort's depth limit: the caller uses the host builder then.
hipError_t device_build(const HostScene& scene, const rt_bvh_options& options, DeviceBvh& out, bool& unsupported);

} // namespace rt
