/*
 * rt_amd.h — C ABI of the MI355X render-loop library (librt_amd.so).
 *
 * The reference ray tracer has no FFI layer; its swap point is a pair of C++ entry points,
 *   BvhInterface(Scene*, const Features&) / BvhInterface::intersect   (reference src/bvh_interface.h:12-41)
 *   renderRayTracing(scene, camera, bvh, screen, features)             (reference src/render.h:17)
 * plus a set of global tunables the GUI edits between frames.  Each entry point below names the
 * reference interface it replaces.  Everything is plain-old-data: pointers, sizes and fixed-layout
 * structs; no C++ or torch types cross this boundary, nothing throws across it, every function returns
 * an rt_status (0 = success) and rt_last_error() describes the last failure of the calling thread.
 *
 * Conventions
 *   - all arithmetic is IEEE fp32 unless stated; vectors are 3 packed floats;
 *   - images (textures, cube-map faces) are 8-bit RGB, row-major, top row first; a texel value is
 *     byte / 255.0f exactly as the reference's loader computes it (reference framework/src/image.cpp:29-32);
 *   - framebuffers are float RGB, row-major, TOP row first: pixel (x, y) with y counted from the bottom is
 *     stored at ((H-1-y)*W + x) (reference src/screen.cpp:37-43).
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t rt_status;
enum {
    RT_OK = 0,
    RT_ERR_INVALID = 1,   /* bad argument (null pointer, negative size, index out of range ...) */
    RT_ERR_NO_DEVICE = 2, /* HIP device/runtime unavailable */
    RT_ERR_HIP = 3,       /* a HIP call failed; see rt_last_error() */
    RT_ERR_LIMIT = 4      /* input exceeds a documented limit (BVH deeper than RT_MAX_LEVELS ...) */
};

enum { RT_MAX_LEVELS = 33 /* extr_max_level = 32 gives at most 33 levels; reference src/bounding_volume_hierarchy.cpp:16,117 */ };

/* ---- scene description (replaces the reference's Scene / Mesh / Material / Image / Sphere structs:
 *      reference src/scene.h:31-37, framework/include/framework/mesh.h:14-43, image.h:11-18, src/common.h:31-35) */
typedef struct rt_image_desc {
    const uint8_t* rgb; /* width*height*3 bytes */
    int32_t width, height;
} rt_image_desc;

typedef struct rt_mesh_desc {
    const float* vertices;     /* n_vertices x 8 floats: position xyz, normal xyz, texcoord uv (Vertex) */
    const uint32_t* triangles; /* n_triangles x 3 vertex indices local to this mesh */
    uint32_t n_vertices, n_triangles;
    float kd[3], ks[3];
    float shininess, transparency;
    int32_t texture; /* index into rt_scene_desc.textures, or -1 (Material::kdTexture) */
} rt_mesh_desc;

typedef struct rt_sphere_desc {
    float center[3], radius;
    float kd[3], ks[3];
    float shininess, transparency;
} rt_sphere_desc;

typedef struct rt_scene_desc {
    const rt_mesh_desc* meshes;
    uint32_t n_meshes;
    const rt_image_desc* textures;
    uint32_t n_textures;
    const rt_sphere_desc* spheres;
    uint32_t n_spheres;
    const rt_image_desc* skybox; /* 6 faces in the order right, left, top, bottom, front, back
                                    (reference src/scene.cpp:108-116), or NULL */
} rt_scene_desc;

/* ---- lights (reference src/common.h:37-52); read at render time, so passed per call */
enum { RT_LIGHT_POINT = 0, RT_LIGHT_SEGMENT = 1, RT_LIGHT_PARALLELOGRAM = 2 };
typedef struct rt_light {
    int32_t type;
    /* point:         position, color
     * segment:       endpoint0, endpoint1, color0, color1
     * parallelogram: v0, edge01, edge02, color0, color1, color2, color3 */
    float v[21];
} rt_light;

/* ---- feature toggles (reference src/common.h:54-77, same meaning, one int each) */
typedef struct rt_features {
    int32_t shading, recursive, hard_shadow, soft_shadow, normal_interp, texture_mapping, accel_structure;
    int32_t environment_mapping, bvh_sah_binning, motion_blur, bloom, bilinear_filtering, mipmap_filtering,
        multiple_rays, glossy, transparency, depth_of_field;
} rt_features;

/* ---- snapshot of the reference's global tunables, taken at each call */
typedef struct rt_params {
    int32_t samples_per_unit;          /* light.cpp:11   (10) segment light */
    int32_t samples_per_unit_parallel; /* light.cpp:16   (10) parallelogram light */
    int32_t ray_multiplier;            /* multipleRays.cpp:11 (1): n x n samples per pixel */
    int32_t ray_depth;                 /* render.h:20    (4): specular bounce budget */
    int32_t glossy_filter_size;        /* gloss.cpp:9    (64) */
    float glossy_sigma;                /* gloss.cpp:10   (1) */
    int32_t dof_samples;               /* dof.cpp:9      (3) */
    float dof_aperture, dof_f, dof;    /* dof.cpp:10-12  (1.4, 2.5, 0) */
    float bloom_threshold;             /* bloom.cpp:5    (0.7) */
    int32_t bloom_size;                /* bloom.cpp:6    (2) */
    float bloom_sigma;                 /* bloom.cpp:7    (1) */
    int32_t bloom_debug;               /* bloom.cpp:9    (0) */
    int32_t bloom_gaussian;            /* bloom.cpp:10   (1) */
    float bloom_scale;                 /* bloom.cpp:11   (1) */
    int32_t enabled_skybox;            /* environment_mapping.cpp:5 (0) */
    int32_t enabled_refl_map;          /* environment_mapping.cpp:6 (0) */
    int32_t max_transparency_depth;    /* no counterpart: the reference's see-through continuation is unbounded
                                          (render.cpp:36-46); this caps it (default 16, at most 24; rt_stats'
                                          overflow counter reports if it was ever reached) */
    uint32_t seed;                     /* frame seed of the counter-based random stream */
    int32_t count_tests;               /* no counterpart: also count ray/box and ray/triangle tests into rt_stats
                                          (slower kernel variant; ray counts are always collected) */
} rt_params;

/* ---- camera (replaces Trackball: reference framework/src/trackball.cpp:23-31,71-74,101-126) */
typedef struct rt_camera {
    float fovy;        /* radians */
    float aspect;      /* width / height of the window (reference framework/src/window.cpp:379-384) */
    float look_at[3];
    float distance;
    float rotation[3]; /* Euler angles, radians */
} rt_camera;

typedef struct rt_ray {
    float origin[3], direction[3], t;
} rt_ray;

/* result of one closest-hit query; mesh/triangle are the ids the reference keeps in locals
 * (bounding_volume_hierarchy.cpp:536-537) */
typedef struct rt_hit {
    int32_t hit;      /* 0/1: BvhInterface::intersect's return value */
    int32_t mesh;     /* mesh index, -1 for a sphere, undefined on a miss */
    int32_t triangle; /* triangle index inside the mesh (sphere index for spheres) */
    float t;          /* ray.t after the call (unchanged on a miss) */
    float normal[3];  /* HitInfo::normal (geometric, or interpolated with normal_interp) */
    float bary[3];    /* HitInfo::barycentricCoord */
    float kd[3];      /* HitInfo::material.kd (texel when texture_mapping and the mesh has a texture) */
} rt_hit;

/* one node of the hierarchy in the reference's own order (DFS pre-order, std::vector<BVHNode>), for pinning
 * the build: reference src/bounding_volume_hierarchy.h:15-23 */
typedef struct rt_node_record {
    int32_t is_leaf, level;
    float lower[3], upper[3];
    int32_t a, b; /* inner: child node indices; leaf: first primitive slot, primitive count */
} rt_node_record;
/* one primitive slot of a leaf: (triangle id, mesh id) with mesh -1 for spheres (…hierarchy.cpp:119-122) */
typedef struct rt_prim_ref {
    int32_t triangle, mesh;
} rt_prim_ref;

typedef struct rt_bvh_options {
    int32_t sah_binning; /* Features::extra.enableBvhSahBinning */
    int32_t max_level;   /* extr_max_level (32) */
    int32_t sah_bins;    /* extr_sah_bins (64) */
} rt_bvh_options;

/* per-call counters; a "ray" is one BvhInterface::intersect query */
typedef struct rt_stats {
    uint64_t primary_rays, shadow_rays, bounce_rays, probe_rays;
    uint64_t box_tests, triangle_tests;
    uint64_t transparency_overflows;
    float kernel_ms;  /* device time of this call's traversal/shading kernel launches (HIP events on the call's stream) */
    float resolve_ms; /* device time of the sample-resolve pass (0 when one sample per pixel) */
    /* breakdown when the call ran as primary / shadow / shade passes in a single chunk (else zero): */
    uint64_t shadow_box_tests, shadow_triangle_tests; /* part of box_tests / triangle_tests done by the shadow pass */
    float primary_ms, shadow_ms, shade_ms;
    float cull_ms;
    /* shadow queries the reference would trace whose outcome cannot reach the pixel (the Phong term of that light
     * sample is exactly zero because the surface faces away from it) and that the library therefore does not trace;
     * shadow_rays counts traced queries only */
    uint64_t culled_shadow_rays;
    /* calls of the reference's getFinalColor (src/render.cpp:18-80) that were evaluated as nodes of wavefront passes
     * (one generation of the recursion per pass); 0 when the frame ran in the per-lane kernels */
    uint64_t wavefront_nodes;
    /* shadow pass (pair stream), only with rt_params.count_tests: packets of 64 queries, box steps (up to 64 stack
     * entries each), primitive steps, stack entries popped by the box steps */
    uint64_t stream_packets, stream_box_steps, stream_prim_steps, stream_entries;
    /* duration of the second, small shadow kernel of a pass (the lighting of the camera rays that were finished on the
     * library's own stream beside the main lighting round); shadow_ms is the main round's kernel alone */
    float shadow_b_ms;
} rt_stats;

typedef struct rt_scene rt_scene; /* device copy of geometry, materials, textures, skybox */
typedef struct rt_bvh rt_bvh;     /* hierarchy built on the host in the reference's order + its device layout */

/* Library / device ------------------------------------------------------------------------------------ */
rt_status rt_init(int32_t device); /* selects the HIP device for the calling thread; optional */
const char* rt_last_error(void);
void rt_default_params(rt_params* out);

/* Scene upload: replaces handing a Scene* to BvhInterface (src/bvh_interface.cpp:6-9).  Host pointers are
 * borrowed for the duration of the call only. */
rt_status rt_scene_create(const rt_scene_desc* desc, rt_scene** out);
void rt_scene_destroy(rt_scene* scene);

/* BVH build + flatten + upload: replaces BoundingVolumeHierarchy's constructor
 * (src/bounding_volume_hierarchy.cpp:41-208).  The scene must outlive the bvh. */
rt_status rt_bvh_build(rt_scene* scene, const rt_bvh_options* options, rt_bvh** out);
/* The same hierarchy, node for node, built on the current GPU: primitive boxes, the std::sort calls of the reference's
 * build (src/bounding_volume_hierarchy.cpp:133-197: one per node on axis level % 3 in the median variant; on x, y, z and
 * the winning axis in the SAH variant) reproduced move for move — equal centroid coordinates end up where libstdc++'s
 * introsort puts them —, the SAH cut search (first of equal costs, as the sequential search keeps it), node numbering,
 * boxes and the device tables, without a round trip through host memory (1.31 M triangles: median variant 0.03 s
 * against 0.47 s on 16 host threads, SAH variant with 64 bins 0.13 s against 1.9 s).  The hierarchy lives on the
 * device that is current at the call.  Falls back to rt_bvh_build for scenes of fewer than two primitives, NaN
 * coordinates, and when the device has no room for the builder's working tables.
 * rt_bvh_device_sorted tells which way a hierarchy was built (1 = on the GPU). */
rt_status rt_bvh_build_device(rt_scene* scene, const rt_bvh_options* options, rt_bvh** out);
int32_t rt_bvh_device_sorted(const rt_bvh* bvh);
void rt_bvh_destroy(rt_bvh* bvh);
int32_t rt_bvh_num_levels(const rt_bvh* bvh); /* BvhInterface::numLevels */
int32_t rt_bvh_num_leaves(const rt_bvh* bvh); /* BvhInterface::numLeaves */
int32_t rt_bvh_num_nodes(const rt_bvh* bvh);
int32_t rt_bvh_num_prim_refs(const rt_bvh* bvh);
/* copies the hierarchy out in the reference's node order (what debugDrawLevel / debugDrawLeaf walk in the reference);
 * a hierarchy built on the GPU is read back from device memory by this call */
rt_status rt_bvh_export(const rt_bvh* bvh, rt_node_record* nodes, int32_t n_nodes, rt_prim_ref* prims, int32_t n_prims);

/* Batched BvhInterface::intersect (src/bounding_volume_hierarchy.cpp:459-548): n independent closest-hit
 * queries, ray i's t on entry is its upper bound.  `rays`/`hits` are HOST pointers when device_ptrs == 0 and
 * DEVICE pointers otherwise.  `stream` is a hipStream_t or NULL. */
rt_status rt_intersect_batch(const rt_bvh* bvh, const rt_ray* rays, uint64_t n, const rt_features* features,
    rt_hit* hits, int32_t device_ptrs, void* stream);

/* Batched testVisibilityLightSample (src/light.cpp:68-89): for hit point i (ray + its hit t) and each of the
 * n_samples positions, out[i*n_samples + s] = 1 if the sample is visible, 0 if shadowed.  Host pointers. */
rt_status rt_visibility_batch(const rt_bvh* bvh, const rt_ray* rays_with_hit_t, uint64_t n, const float* sample_positions,
    int32_t n_samples, const rt_features* features, uint8_t* out);

/* renderRayTracing (src/render.cpp:82-128) for rows [row_begin, row_end) counted from the BOTTOM of a
 * width x height frame (the reference's y).  `out_rgb` receives (row_end-row_begin)*width float RGB pixels,
 * top-most of those rows first — i.e. the slice of the Screen buffer those rows occupy; host pointer when
 * device_ptr == 0, device pointer otherwise.  Bloom, which needs the whole frame, is NOT applied here even
 * if features->bloom is set: call rt_bloom on the assembled frame.
 * Several threads may render from one rt_bvh at the same time, as the reference's per-camera threads do
 * (src/main.cpp:533-552): every call works on a scratch set of its own (queues, sample planes, counters; up to eight
 * sets per rt_bvh, kept until rt_bvh_destroy; a ninth concurrent call waits for a set).  With a host frame or a stats
 * pointer the call returns when the frame is complete; a host-frame call that names no stream runs on a stream of its
 * scratch set, so such calls overlap on the GPU.  With a device frame and stats == NULL the call only queues its work on
 * `stream` and returns; the scratch set stays reserved until that work has finished (calls queued on the same stream
 * reuse it, stream order keeps them apart).  The library may run part of a call on a second stream of its own; that
 * stream is joined into the call's stream before the call's work there ends. */
rt_status rt_render(const rt_bvh* bvh, const rt_camera* camera, const rt_light* lights, int32_t n_lights,
    const rt_features* features, const rt_params* params, int32_t width, int32_t height, int32_t row_begin,
    int32_t row_end, float* out_rgb, int32_t device_ptr, void* stream, rt_stats* stats);

/* The same for an interleaved set of row blocks in ONE call: the frame's Screen rows (counted from the TOP)
 * are cut into blocks of `row_block` rows; this call renders blocks first_block, first_block + block_stride,
 * first_block + 2 * block_stride, ... and writes them one after the other into `out_rgb` (block i of the call
 * at rows [i * row_block, ...), top-most row first; only the frame's last block can be shorter).  This is the
 * reference's row parallelism (`#pragma omp parallel for` over y, src/render.cpp:86-89) with the rows dealt to
 * GPUs instead of threads: rank r of n renders (row_block, r, n) and the blocks are gathered afterwards. */
rt_status rt_render_blocks(const rt_bvh* bvh, const rt_camera* camera, const rt_light* lights, int32_t n_lights,
    const rt_features* features, const rt_params* params, int32_t width, int32_t height, int32_t row_block,
    int32_t first_block, int32_t block_stride, float* out_rgb, int32_t device_ptr, void* stream, rt_stats* stats);

/* addBloom (src/bloom.cpp:88-114) on a full width x height float RGB frame, in place.  With a device pointer the
 * work is queued on `stream` and the call returns without waiting; with a host pointer it returns when the pixels
 * are back.  The frame copy the filter reads from is a per-device buffer the library keeps between calls (allocated
 * on the first call on a device and when a larger frame arrives). */
rt_status rt_bloom(float* pixels, int32_t width, int32_t height, const rt_params* params, int32_t device_ptr, void* stream);

/* rt_bloom for callers that cannot allocate device memory (code under HIP graph capture, per-frame present loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  Same result as rt_bloom on a DEVICE frame,
 * queued on `stream`, no wait; it calls no hipMalloc / hipFree and nothing else that allocates: the frame copy goes
 * to a static array inside the library's code object (rt_bloom_static_max_pixels() * 12 bytes = 3840 x 2160 pixels;
 * every GPU the library runs on holds its copy until the process ends).  Larger frames are refused with RT_ERR_LIMIT.
 * rt_init(device) resolves the array's address for that device; otherwise the first call on a device does.  All
 * calls on one GPU share that GPU's array: calls that overlap in time on one GPU (different streams) race. */
rt_status rt_bloom_static(float* device_pixels, int32_t width, int32_t height, const rt_params* params, void* stream);
uint64_t rt_bloom_static_max_pixels(void);

/* Multi-GPU frame assembly, one rank (process or host thread) per GPU.  Replaces the shared Screen that the reference's
 * row threads write into (`#pragma omp parallel for` over y, src/render.cpp:86-89; the per-camera threads of
 * src/main.cpp:533-552 when the work is dealt to GPUs): rank r of `world` renders rt_render_blocks(row_block, r, world)
 * into device memory and rt_gather_blocks moves every block straight into its place in the root's device frame, as one
 * group of RCCL sends / receives over xGMI (no pass through host memory; the root's own blocks are device copies).
 * Bloom, which needs the whole frame, runs on the root afterwards (rt_bloom).
 * RCCL is loaded at run time (RT_RCCL_LIBRARY, default librccl.so.1); without it these calls return RT_ERR_NO_DEVICE.
 * A communicator belongs to the device that was current when it was made.  rt_comm_unique_id + rt_comm_init_rank
 * bootstrap ranks in different processes (the id travels by any channel the caller has); rt_comm_init_all makes the
 * communicators of n devices of one process. */
typedef struct rt_comm rt_comm;
#define RT_COMM_ID_BYTES 128
rt_status rt_comm_unique_id(uint8_t* id /* RT_COMM_ID_BYTES */);
rt_status rt_comm_init_rank(const uint8_t* id, int32_t rank, int32_t world, rt_comm** out);
rt_status rt_comm_init_all(int32_t n, const int32_t* devices /* NULL: 0..n-1 */, rt_comm** out /* n handles */);
void rt_comm_destroy(rt_comm* comm);
int32_t rt_comm_rank(const rt_comm* comm);
int32_t rt_comm_world(const rt_comm* comm);
/* `share`: this rank's blocks as rt_render_blocks wrote them (device memory); `frame`: width x height float RGB in
 * device memory on the root (ignored elsewhere).  Queued on `stream`; every rank of the communicator must call it. */
rt_status rt_gather_blocks(rt_comm* comm, int32_t root, const float* share, float* frame, int32_t width, int32_t height,
    int32_t row_block, void* stream);
/* Both steps in one call, for callers without device memory of their own (the C++ facade): renders the blocks of this
 * communicator's rank into a buffer the hierarchy keeps, gathers them, and on the root leaves the assembled frame in
 * `frame` (host pointer when device_ptr == 0: one device-to-host copy of the whole frame instead of one per share).
 * Returns when this rank's part is done.  `bvh` must live on the communicator's device. */
rt_status rt_render_blocks_gather(const rt_bvh* bvh, const rt_camera* camera, const rt_light* lights, int32_t n_lights,
    const rt_features* features, const rt_params* params, int32_t width, int32_t height, int32_t row_block, rt_comm* comm,
    int32_t root, float* frame, int32_t device_ptr, void* stream, rt_stats* stats);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
