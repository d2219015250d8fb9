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
// The same with the per-level sorts already done: `leafOrder` (primitive numbers in (mesh, triangle) then sphere order)
// is the order the leaves hold them in; the builder only cuts ranges, makes the boxes and numbers the nodes.
void build_bvh(const HostScene& scene, const rt_bvh_options& options, HostBvh& out, const std::vector<int>* leafOrder);
// Whether the per-level sorts can be done apart from the builder (median variant, tree not cut short by the level limit).
bool device_sort_applies(const HostScene& scene, const rt_bvh_options& options);
// Centroids of the primitives (3 floats each) in the builder's primitive order, computed with its operations.
void primitive_centroids(const HostScene& scene, std::vector<float>& centroids);

struct FlatBvh {
    std::vector<NodeRec> nodes;
    std::vector<PrimRec> prims;
    std::vector<AttrRec> attrs;
    std::vector<uint32_t> primOrder;
    float rootLo[3], rootHi[3];
    int32_t rootRef = -1;
};
void flatten_bvh(const HostScene& scene, const HostBvh& bvh, FlatBvh& out);

// std::sort of `count` primitive numbers by centroid coordinate exactly as the builder does it for a node (equal keys
// end up where the builder's std::sort puts them, given the same input order).
void reference_sort(int* indices, size_t count, const float* centroids, int axis);

// (bvh_device.hip) The leaf order the reference's median-split build produces, by one segmented radix sort per level on
// the current GPU.  A node where equal keys meet at its cut is sorted again on the host by reference_sort from the order
// the node's primitives arrive in — which is known whenever their keys on the parent's axis are all different.
// `ambiguous` is set instead when that is not the case (or a key is NaN, or there are very many such nodes): the
// outcome then depends on the whole history of std::sort's moves and the host build must be used.  `hostFixed` counts
// the nodes that were sorted again on the host.
hipError_t device_split_order(const float* centroids, size_t count, std::vector<int>& order, bool& ambiguous, int* hostFixed = nullptr);

} // namespace rt
