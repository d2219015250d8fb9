// HBM data layout of a scene and its hierarchy (what the kernels read), and the host-side mirror the
// builder fills before upload.
//
// Tables (each one contiguous array in device memory, 64-byte aligned):
//   nodes  : one 64-byte NodeRec per INNER node of the reference's binary tree, holding the boxes of BOTH
//            children plus the two child references, so one record feeds the two ray/box tests an inner
//            visit needs (the reference keeps a box per node and copies both children, reference
//            src/bounding_volume_hierarchy.cpp:405-424).
//   prims  : one 64-byte PrimRec per leaf primitive slot, in leaf order (the order the leaves appear in
//            the reference's DFS pre-order and, inside a leaf, the order of its ids vector).  A leaf is a
//            run of consecutive records; `more` counts the records that follow in the same leaf.
//   attrs  : one 64-byte AttrRec per prims[] entry (vertex normals and texture coordinates); only read
//            for the final hit when normals are interpolated or a texture is sampled.
//   mats   : one 48-byte MaterialRec per mesh, then one per sphere.
//   texels : all textures then the six cube-map faces as 8-bit RGB, with (offset,width,height) tables.
#pragma once
#include "rt_math.h"

namespace rt {

// child reference: >= 0 -> index into nodes[]; < 0 -> leaf starting at prims[~ref]
// The six planes of a child box are stored slab by slab: lo.x hi.x lo.y hi.y lo.z hi.z.  The two planes of a slab
// meet the same ray origin component and reciprocal, so a slab is one packed subtract and one packed multiply with
// the ray's scalars broadcast to both halves, and its near / far values are the min / max of the two halves.
struct alignas(64) NodeRec {
    float boxL[6];
    float boxR[6];
    int32_t refL, refR;
    int32_t level;   // depth of this inner node in the reference's tree (root = 0)
    int32_t refNode; // index of this node in the reference's node vector
};
static_assert(sizeof(NodeRec) == 64, "NodeRec must be 64 bytes");

// Triangle: vertices + the ray-independent part of the reference's test, computed once with the same fp32
// operations the reference repeats per test (trianglePlane, reference src/intersect.cpp:52-60, and the
// 1/|cross(v0-v2, v1-v2)| of pointInTriangle, :26).  Sphere (mesh == -1): v0 = centre, v1[0] = radius.
struct alignas(64) PrimRec {
    float v0[3], v1[3], v2[3];
    float n[3];     // plane normal ((1,0,0) for a degenerate triangle)
    float invArea;  // 1 / |cross(v0 - v2, v1 - v2)|
    int32_t more;   // further records of the same leaf after this one
    int32_t mesh;   // mesh id, -1 = sphere
    int32_t prim;   // triangle id inside the mesh / sphere id
};
static_assert(sizeof(PrimRec) == 64, "PrimRec must be 64 bytes");

struct alignas(64) AttrRec {
    float n0[3], n1[3], n2[3];
    float uv0[2], uv1[2], uv2[2];
    float pad;
};
static_assert(sizeof(AttrRec) == 64, "AttrRec must be 64 bytes");

struct alignas(16) MaterialRec {
    float kd[3], ks[3];
    float shininess, transparency;
    int32_t texture; // index into the texture table or -1
    int32_t pad[3];
};
static_assert(sizeof(MaterialRec) == 48, "MaterialRec must be 48 bytes");

struct ImageRec {
    uint32_t offset; // byte offset into texels[]
    int32_t width, height;
};

// What a kernel needs to find everything; passed by value as a kernel argument.
struct SceneView {
    const NodeRec* nodes;
    const PrimRec* prims;
    const AttrRec* attrs;
    const MaterialRec* mats;
    const uint8_t* texels;
    const ImageRec* textures;
    const uint32_t* primOrder; // prims[] indices in the reference's (mesh, triangle) then sphere order: brute-force path
    ImageRec skybox[6];
    float rootLo[3], rootHi[3];
    int32_t rootRef;    // child-style reference of the root node
    int32_t numPrims;
    int32_t numNodes;   // inner-node records
    int32_t numMeshes;
    int32_t hasSkybox;
    int32_t hasSpheres;
    int32_t stackDepth; // entries a traversal stack can need (number of tree levels)
    int32_t bounded;    // the root box (hence every box) lies within +-1e30: the packed box tests apply
};

} // namespace rt
