// C ABI of librt_amd (declared in include/rt_amd.h): handle management, per-frame parameter snapshot,
// uploads and kernel launches.  Host code only; the kernels live in rt_kernels.hip.
#include "bvh_build.h"
#include "rt_amd.h"
#include "rt_kernels.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <exception>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

using namespace rt;

namespace {

thread_local std::string g_error;

rt_status fail(rt_status code, const std::string& msg)
{
    g_error = msg;
    return code;
}
rt_status hip_fail(hipError_t e, const char* what)
{
    return fail(e == hipErrorNoDevice || e == hipErrorInvalidDevice ? RT_ERR_NO_DEVICE : RT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define RT_HIP(call)                                            \
    do {                                                        \
        hipError_t e_ = (call);                                 \
        if (e_ != hipSuccess) return hip_fail(e_, #call);       \
    } while (0)

// Growable device buffer.
struct DevBuf {
    void* ptr = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(&ptr, bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release()
    {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
    }
    // takes over a hipMalloc'd block
    template <class T>
    void adopt(T*& p, size_t bytes)
    {
        release();
        ptr = p;
        cap = bytes;
        p = nullptr;
    }
};

template <class T>
hipError_t upload(DevBuf& buf, const std::vector<T>& v)
{
    const size_t bytes = v.size() * sizeof(T);
    hipError_t e = buf.reserve(bytes ? bytes : 64);
    if (e != hipSuccess) return e;
    if (bytes) e = hipMemcpy(buf.ptr, v.data(), bytes, hipMemcpyHostToDevice);
    return e;
}

} // namespace

// (for the library's other translation units: sets the calling thread's error text)
rt_status rt_set_error(rt_status code, const std::string& msg) { return fail(code, msg); }

// Share of the scene's triangle area that belongs to specular materials (ks != 0): how much of what a camera ray can
// hit spawns reflection rays.
float specular_area_share(const HostScene& scene)
{
    double total = 0.0, specular = 0.0;
    for (const HostMesh& m : scene.meshes) {
        double area = 0.0;
        for (size_t t = 0; t + 2 < m.triangles.size(); t += 3) {
            const float* a = &m.vertices[size_t(m.triangles[t]) * 8];
            const float* b = &m.vertices[size_t(m.triangles[t + 1]) * 8];
            const float* c = &m.vertices[size_t(m.triangles[t + 2]) * 8];
            const double u[3] = { double(b[0]) - a[0], double(b[1]) - a[1], double(b[2]) - a[2] };
            const double v[3] = { double(c[0]) - a[0], double(c[1]) - a[1], double(c[2]) - a[2] };
            const double n[3] = { u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0] };
            area += 0.5 * std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        }
        total += area;
        if (m.material.ks[0] != 0.0f || m.material.ks[1] != 0.0f || m.material.ks[2] != 0.0f) specular += area;
    }
    return total > 0.0 ? float(specular / total) : 0.0f;
}

struct rt_scene {
    HostScene host;
    // device copies are made on first use (so that building and inspecting a hierarchy needs no GPU)
    std::mutex lock;
    bool uploaded = false;
    int device = 0;
    DevBuf mats, texels, textures;
    ImageRec skybox[6] = {};
    float specularShare = 0.0f; // specular_area_share() of the scene
};

// Device memory, events and the second stream one rt_render call works with.  A call takes a set from its hierarchy's
// pool and gives it back when it returns, so calls on one rt_bvh from several threads (the reference renders one camera
// per thread from one BvhInterface, src/main.cpp:533-552) run side by side, each on its own set.
struct RenderScratch {
    DevBuf stats, frame, queue, queueCounters, visibility, shadowQueue, shadeList, pairSpill, samples, dofFlags, wfRays, wfNodes, wfMeta, wfValue, wfHits, wfHitNode, wfGlossy, gaveUpQueue, pairSpillAux, laneStacks, lights;
    std::vector<LightDev> hostLights; // (more than INLINE_LIGHTS lights: what `lights` was filled from)
    hipEvent_t evStart = nullptr, evMid = nullptr, evStop = nullptr, evSplit[4] = { nullptr, nullptr, nullptr, nullptr };
    // second stream (and its fork / join events) on which a direct-lighting pass finishes its long camera rays, and the
    // two marks around that stream's shadow kernel
    hipStream_t auxStream = nullptr;
    hipEvent_t evFork = nullptr, evJoin = nullptr, evShadowB[2] = { nullptr, nullptr };
    // stream of the set's own: a call that hands over host pixels and names no stream runs on it instead of the null
    // stream (such a call returns only when its frame is complete, so nothing else can be ordered against it)
    hipStream_t ownStream = nullptr;
    // a call that returned without waiting (device frame, no statistics) leaves the set in use by `lastStream` until `done`
    hipEvent_t done = nullptr;
    hipStream_t lastStream = nullptr;
    bool pending = false, busy = false;
    // pinned host memory the wavefront passes' device-side bookkeeping reports to (node count, overflow flag, table)
    unsigned long long* wfReport = nullptr;

    hipError_t init()
    {
        hipError_t e;
        if ((e = hipHostMalloc(reinterpret_cast<void**>(&wfReport), (2 + Q_COUNT - Q_WF_GEN) * sizeof(unsigned long long), hipHostMallocDefault)) != hipSuccess) return e;
        std::memset(wfReport, 0, (2 + Q_COUNT - Q_WF_GEN) * sizeof(unsigned long long));
        for (hipEvent_t* ev : { &evStart, &evMid, &evStop, &evSplit[0], &evSplit[1], &evSplit[2], &evSplit[3], &evShadowB[0], &evShadowB[1] })
            if ((e = hipEventCreate(ev)) != hipSuccess) return e;
        for (hipEvent_t* ev : { &evFork, &evJoin, &done })
            if ((e = hipEventCreateWithFlags(ev, hipEventDisableTiming)) != hipSuccess) return e;
        if ((e = hipStreamCreateWithFlags(&auxStream, hipStreamNonBlocking)) != hipSuccess) return e;
        return hipStreamCreateWithFlags(&ownStream, hipStreamNonBlocking);
    }
    ~RenderScratch()
    {
        if (pending && done) (void)hipEventSynchronize(done);
        if (auxStream) {
            (void)hipStreamSynchronize(auxStream); // (joined into the caller's stream at the end of every pass; nothing should be left)
            (void)hipStreamDestroy(auxStream);
        }
        if (ownStream) {
            (void)hipStreamSynchronize(ownStream);
            (void)hipStreamDestroy(ownStream);
        }
        if (wfReport) (void)hipHostFree(wfReport);
        for (DevBuf* d : { &stats, &frame, &queue, &queueCounters, &visibility, &shadowQueue, &shadeList, &pairSpill, &samples, &dofFlags, &wfRays, &wfNodes, &wfMeta, &wfValue, &wfHits, &wfHitNode, &wfGlossy, &gaveUpQueue, &pairSpillAux, &laneStacks })
            d->release();
        for (hipEvent_t ev : { evStart, evMid, evStop, evSplit[0], evSplit[1], evSplit[2], evSplit[3], evShadowB[0], evShadowB[1], evFork, evJoin, done })
            if (ev) (void)hipEventDestroy(ev);
    }
};

struct rt_bvh {
    rt_scene* scene = nullptr;
    HostBvh host;
    FlatBvh flat; // host image of the device tables; dropped after the upload
    bool uploaded = false;
    bool deviceBuilt = false; // built by rt_bvh_build_device on the GPU: `host` holds the counts only, the node and
                              // primitive lists stay in device memory (refNodes / refPrims) until rt_bvh_export asks
    DevBuf nodes, prims, attrs, primOrder, refNodes, refPrims;
    size_t numRefNodes = 0, numRefPrims = 0;
    SceneView view {};
    // `lock` guards the upload and the buffers of the batch entry points (those calls run one at a time)
    std::mutex lock, gatherLock;
    DevBuf rays, hits, misc;
    DevBuf gatherShare, gatherFrame; // rt_render_blocks_gather: this rank's blocks, and on the root the assembled frame
    // scratch sets of rt_render, guarded by `poolLock`
    enum { MAX_SCRATCH_SETS = 8 };
    std::mutex poolLock;
    std::condition_variable poolFree;
    std::vector<std::unique_ptr<RenderScratch>> pool;
};

namespace {

void fill_lights(const rt_light* lights, int n, const rt_features& f, const rt_params& p, LightDev* out, int& drawsPerHit)
{
    drawsPerHit = 0;
    for (int i = 0; i < n; i++) {
        LightDev& d = out[i];
        std::memset(&d, 0, sizeof d);
        d.type = lights[i].type;
        std::memcpy(d.v, lights[i].v, sizeof d.v);
        d.nA = d.nB = 1;
        if (d.type == RT_LIGHT_SEGMENT) {
            // reference src/light.cpp:151-160
            const vec3 e0 = load3(d.v), e1 = load3(d.v + 3);
            const float segmentLengthRec = 1.0f / length(e1 - e0);
            const float sampleCountRec = 1.0f / std::floor(std::fmax(float(p.samples_per_unit) / segmentLengthRec, 1.0f));
            const vec3 step = (e1 - e0) * sampleCountRec;
            const float sampleCount = 1.0f / sampleCountRec;
            d.recA = sampleCountRec;
            d.recB = segmentLengthRec;
            d.step[0] = step.x; d.step[1] = step.y; d.step[2] = step.z;
            int trips = 0;
            for (int k = 0; float(k) < sampleCount; k++) trips++;
            d.nA = trips;
            d.draws = (f.soft_shadow || f.shading) ? trips : 0;
        } else if (d.type == RT_LIGHT_PARALLELOGRAM) {
            // reference src/light.cpp:188-196
            const float recI = 1.0f / std::floor(std::fmax(float(p.samples_per_unit_parallel) * length(load3(d.v + 3)), 1.0f));
            const float recJ = 1.0f / std::floor(std::fmax(float(p.samples_per_unit_parallel) * length(load3(d.v + 6)), 1.0f));
            const float countI = 1.0f / recI, countJ = 1.0f / recJ;
            d.recA = recI;
            d.recB = recJ;
            int ti = 0, tj = 0;
            for (int k = 0; float(k) < countI; k++) ti++;
            for (int k = 0; float(k) < countJ; k++) tj++;
            d.nA = ti;
            d.nB = tj;
            d.draws = (f.soft_shadow || f.shading) ? 2 * ti * tj : 0;
        }
        d.drawBase = drawsPerHit;
        drawsPerHit += d.draws;
    }
}

// Numbers the light samples that cast a shadow ray under the given features; returns their count per hit.
int number_shadow_samples(LightDev* lights, int n, const rt_features& f)
{
    int at = 0;
    for (int i = 0; i < n; i++) {
        LightDev& d = lights[i];
        const bool enabled = d.type == RT_LIGHT_POINT || f.soft_shadow || f.shading;
        const bool shadowed = d.type == RT_LIGHT_POINT ? f.hard_shadow != 0 : f.soft_shadow != 0;
        d.shadowBase = -1;
        if (enabled && shadowed) {
            d.shadowBase = at;
            at += d.nA * d.nB;
        }
    }
    return at;
}

// Trackball state, reference framework/src/trackball.cpp:23-31,71-74,117-126 and glm's Euler-angle quaternion
// constructor (glm/detail/type_quat.inl:208-217).
void fill_camera(const rt_camera& c, FrameParams& P)
{
    const float ex = c.rotation[0] * 0.5f, ey = c.rotation[1] * 0.5f, ez = c.rotation[2] * 0.5f;
    const float cx = std::cos(ex), cy = std::cos(ey), cz = std::cos(ez);
    const float sx = std::sin(ex), sy = std::sin(ey), sz = std::sin(ez);
    quat q;
    q.w = cx * cy * cz + sx * sy * sz;
    q.x = sx * cy * cz - cx * sy * sz;
    q.y = cx * sy * cz + sx * cy * sz;
    q.z = cx * cy * sz - sx * sy * cz;
    P.camQ = q;
    const vec3 pos = load3(c.look_at) + rotate(q, make3(0.0f, 0.0f, -c.distance));
    const vec3 up = rotate(q, make3(0.0f, 1.0f, 0.0f));
    const vec3 left = rotate(q, make3(1.0f, 0.0f, 0.0f));
    P.camPos[0] = pos.x; P.camPos[1] = pos.y; P.camPos[2] = pos.z;
    P.camUp[0] = up.x; P.camUp[1] = up.y; P.camUp[2] = up.z;
    P.camLeft[0] = left.x; P.camLeft[1] = left.y; P.camLeft[2] = left.z;
    P.halfH = std::tan(c.fovy / 2.0f);
    P.halfW = c.aspect * P.halfH;
}

rt_status check_limits(const rt_bvh* bvh, const rt_features* f, const rt_params* p)
{
    (void)bvh;
    if (p) {
        if (p->ray_depth < 0 || p->ray_depth > MAX_SPECULAR_DEPTH)
            return fail(RT_ERR_LIMIT, "ray_depth must be in [0, " + std::to_string(int(MAX_SPECULAR_DEPTH)) + "]");
        if (p->max_transparency_depth < 0 || p->max_transparency_depth > MAX_BLEND_DEPTH)
            return fail(RT_ERR_LIMIT, "max_transparency_depth must be in [0, " + std::to_string(int(MAX_BLEND_DEPTH)) + "]");
        if (f->multiple_rays && (p->ray_multiplier < 1 || p->ray_multiplier > 64))
            return fail(RT_ERR_LIMIT, "ray_multiplier must be in [1, 64]");
        if (f->depth_of_field && (p->dof_samples < 0 || p->dof_samples > 4095))
            return fail(RT_ERR_LIMIT, "dof_samples must be in [0, 4095]");
        if (f->glossy && (p->glossy_filter_size < 0 || p->glossy_filter_size > 4096))
            return fail(RT_ERR_LIMIT, "glossy_filter_size must be in [0, 4096]");
        if (p->samples_per_unit < 0 || p->samples_per_unit_parallel < 0)
            return fail(RT_ERR_INVALID, "negative light sampling density");
    }
    return RT_OK;
}

rt_status upload_scene(rt_scene* s)
{
    std::lock_guard<std::mutex> guard(s->lock);
    if (s->uploaded) return RT_OK;
    RT_HIP(hipGetDevice(&s->device));
    // materials: meshes then spheres
    std::vector<MaterialRec> mats;
    for (const HostMesh& m : s->host.meshes) mats.push_back(m.material);
    for (const rt_sphere_desc& sp : s->host.spheres) {
        MaterialRec m;
        std::memset(&m, 0, sizeof m);
        std::memcpy(m.kd, sp.kd, 12);
        std::memcpy(m.ks, sp.ks, 12);
        m.shininess = sp.shininess;
        m.transparency = sp.transparency;
        m.texture = -1;
        mats.push_back(m);
    }
    // texels: textures then the six cube-map faces, each 4-byte aligned
    std::vector<uint8_t> texels;
    std::vector<ImageRec> recs;
    auto append = [&](const HostImage& im) {
        ImageRec r { uint32_t(texels.size()), im.width, im.height };
        texels.insert(texels.end(), im.rgb.begin(), im.rgb.end());
        while (texels.size() % 4) texels.push_back(0);
        return r;
    };
    for (const HostImage& im : s->host.textures) recs.push_back(append(im));
    for (size_t i = 0; i < s->host.skybox.size(); i++) s->skybox[i] = append(s->host.skybox[i]);
    RT_HIP(upload(s->mats, mats));
    RT_HIP(upload(s->texels, texels));
    RT_HIP(upload(s->textures, recs));
    s->uploaded = true;
    return RT_OK;
}

// The kernels' view of a hierarchy whose tables are in device memory.
void fill_view(rt_bvh* b, const float* rootLo, const float* rootHi, int32_t rootRef, size_t numPrims, size_t numNodes)
{
    rt_scene* scene = b->scene;
    SceneView& v = b->view;
    v.nodes = static_cast<const NodeRec*>(b->nodes.ptr);
    v.prims = static_cast<const PrimRec*>(b->prims.ptr);
    v.attrs = static_cast<const AttrRec*>(b->attrs.ptr);
    v.mats = static_cast<const MaterialRec*>(scene->mats.ptr);
    v.texels = static_cast<const uint8_t*>(scene->texels.ptr);
    v.textures = static_cast<const ImageRec*>(scene->textures.ptr);
    v.primOrder = static_cast<const uint32_t*>(b->primOrder.ptr);
    std::memcpy(v.skybox, scene->skybox, sizeof v.skybox);
    std::memcpy(v.rootLo, rootLo, 12);
    std::memcpy(v.rootHi, rootHi, 12);
    v.rootRef = rootRef;
    v.numPrims = int32_t(numPrims);
    v.numNodes = int32_t(numNodes);
    v.numMeshes = int32_t(scene->host.meshes.size());
    v.hasSkybox = scene->host.skybox.size() == 6;
    v.hasSpheres = !scene->host.spheres.empty();
    v.stackDepth = b->host.numLevels;
    v.bounded = 1;
    for (int i = 0; i < 3; i++)
        if (!(std::fabs(v.rootLo[i]) <= 1e30f && std::fabs(v.rootHi[i]) <= 1e30f)) v.bounded = 0;
}

// Makes the device tables of `b` (and of its scene) exist.  Caller holds b->lock.
rt_status upload_bvh(rt_bvh* b)
{
    if (b->uploaded) return RT_OK;
    rt_scene* scene = b->scene;
    rt_status st = upload_scene(scene);
    if (st != RT_OK) return st;
    RT_HIP(hipSetDevice(scene->device));
    const FlatBvh& flat = b->flat;
    RT_HIP(upload(b->nodes, flat.nodes));
    RT_HIP(upload(b->prims, flat.prims));
    RT_HIP(upload(b->attrs, flat.attrs));
    RT_HIP(upload(b->primOrder, flat.primOrder));
    fill_view(b, flat.rootLo, flat.rootHi, flat.rootRef, flat.prims.size(), flat.nodes.size());
    b->flat = FlatBvh {};
    b->uploaded = true;
    return RT_OK;
}

} // namespace


// ---- per-device state of the library (not tied to a scene or hierarchy handle) -----------------------------------
// Scratch of rt_bloom_static: a copy of the frame being filtered.  The name is a host-side stand-in for the symbol;
// host code uses the per-device address obtained with hipGetSymbolAddress (every GPU holds its own copy of the array
// at its own address until the process ends).
#define RT_STATIC_SCRATCH_BYTES (size_t(3840) * size_t(2160) * size_t(12))
// (no kernel names it: they receive the address as an argument.  The host-side use in resolve_scratch() is what makes
// the compiler keep the array in the code object under a name the loader can find; marking it `used` would keep it
// as a local symbol, which hipGetSymbolAddress cannot resolve.)
static __device__ __align__(256) unsigned char scratch[RT_STATIC_SCRATCH_BYTES];

namespace {

enum { MAX_DEVICES = 64 };
struct DeviceState {
    std::mutex lock;
    DevBuf bloomSrc, bloomDst;      // rt_bloom: frame copy (and result staging for host frames), grown on demand
    hipEvent_t bloomDone = nullptr; // recorded after rt_bloom's last use of bloomSrc
    void* scratch = nullptr;        // this device's address of `scratch`
};
DeviceState g_devices[MAX_DEVICES];

rt_status device_state(DeviceState** out)
{
    int dev = 0;
    RT_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEVICES) return fail(RT_ERR_LIMIT, "device index beyond the library's device table");
    *out = &g_devices[dev];
    return RT_OK;
}

// Address of the static scratch array on the current device (caller holds d.lock).
rt_status resolve_scratch(DeviceState& d)
{
    if (d.scratch) return RT_OK;
    void* p = nullptr;
    RT_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) return fail(RT_ERR_HIP, "hipGetSymbolAddress returned a null address for the static scratch");
    d.scratch = p;
    return RT_OK;
}

// gaussianKernel, reference src/bloom.cpp:14-34: the per-axis factors and the sum of all their products
rt_status fill_bloom(const rt_params& p, BloomParams& bp)
{
    if (p.bloom_size < 0 || p.bloom_size > BLOOM_MAX_SIZE) return fail(RT_ERR_LIMIT, "bloom_size must be in [0, " + std::to_string(int(BLOOM_MAX_SIZE)) + "]");
    std::memset(&bp, 0, sizeof bp);
    bp.threshold = p.bloom_threshold;
    bp.scale = p.bloom_scale;
    bp.size = p.bloom_size;
    bp.gaussian = p.bloom_gaussian;
    bp.debug = p.bloom_debug;
    bp.denom = 1.0f;
    if (p.bloom_gaussian) {
        const int k = p.bloom_size, size = 2 * k + 1;
        const float sd = float(std::pow(double(p.bloom_sigma), 2.0));
        const double spread = 1.0 / double(2 * sd);
        for (int i = 0; i < size; i++) {
            const int x = i - k;
            bp.f[i] = float(std::exp(double(-x * x) * spread));
        }
        float denom = 0;
        for (int i = 0; i < size; i++)
            for (int j = 0; j < size; j++) denom += bp.f[j] * bp.f[i];
        bp.denom = denom;
    }
    return RT_OK;
}

} // namespace

extern "C" {

const char* rt_last_error(void) { return g_error.c_str(); }

void rt_default_params(rt_params* p)
{
    if (!p) return;
    std::memset(p, 0, sizeof *p);
    p->samples_per_unit = 10;
    p->samples_per_unit_parallel = 10;
    p->ray_multiplier = 1;
    p->ray_depth = 4;
    p->glossy_filter_size = 64;
    p->glossy_sigma = 1.0f;
    p->dof_samples = 3;
    p->dof_aperture = 1.4f;
    p->dof_f = 2.5f;
    p->dof = 0.0f;
    p->bloom_threshold = 0.7f;
    p->bloom_size = 2;
    p->bloom_sigma = 1.0f;
    p->bloom_debug = 0;
    p->bloom_gaussian = 1;
    p->bloom_scale = 1.0f;
    p->enabled_skybox = 0;
    p->enabled_refl_map = 0;
    p->max_transparency_depth = 16;
    p->seed = 1;
    p->count_tests = 0;
}

rt_status rt_init(int32_t device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) return fail(RT_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= n) return fail(RT_ERR_INVALID, "device index out of range");
    RT_HIP(hipSetDevice(device));
    // set-up of the entry points that must not allocate later: this device's address of the static scratch
    DeviceState* d = nullptr;
    rt_status st = device_state(&d);
    if (st != RT_OK) return st;
    std::lock_guard<std::mutex> guard(d->lock);
    return resolve_scratch(*d);
}

rt_status rt_scene_create(const rt_scene_desc* desc, rt_scene** out)
{
    if (!desc || !out) return fail(RT_ERR_INVALID, "rt_scene_create: null argument");
    *out = nullptr;
    rt_scene* s = new (std::nothrow) rt_scene;
    if (!s) return fail(RT_ERR_INVALID, "out of host memory");
    std::string err;
    try {
        if (!copy_scene(*desc, s->host, err)) {
            delete s;
            return fail(RT_ERR_INVALID, err);
        }
    } catch (const std::exception& e) { // nothing may unwind through the C ABI
        delete s;
        return fail(RT_ERR_LIMIT, std::string("rt_scene_create: ") + e.what());
    } catch (...) {
        delete s;
        return fail(RT_ERR_LIMIT, "rt_scene_create: unknown failure");
    }
    s->specularShare = specular_area_share(s->host);
    *out = s;
    return RT_OK;
}

void rt_scene_destroy(rt_scene* s)
{
    if (!s) return;
    s->mats.release();
    s->texels.release();
    s->textures.release();
    delete s;
}

rt_status rt_bvh_build(rt_scene* scene, const rt_bvh_options* options, rt_bvh** out)
{
    if (!scene || !out) return fail(RT_ERR_INVALID, "rt_bvh_build: null argument");
    *out = nullptr;
    rt_bvh_options opt { 0, 32, 64 };
    if (options) opt = *options;
    if (opt.max_level < 0 || opt.max_level > RT_MAX_LEVELS - 1) return fail(RT_ERR_LIMIT, "max_level must be in [0, 32]");
    if (opt.sah_binning && opt.sah_bins < 1) return fail(RT_ERR_INVALID, "sah_bins must be positive");
    rt_bvh* b = new (std::nothrow) rt_bvh;
    if (!b) return fail(RT_ERR_INVALID, "out of host memory");
    b->scene = scene;
    try {
        build_bvh(scene->host, opt, b->host);
        flatten_bvh(scene->host, b->host, b->flat);
    } catch (const std::exception& e) { // out of host memory, thread limits: nothing may unwind through the C ABI
        delete b;
        return fail(RT_ERR_LIMIT, std::string("rt_bvh_build: ") + e.what());
    } catch (...) {
        delete b;
        return fail(RT_ERR_LIMIT, "rt_bvh_build: unknown failure");
    }
    *out = b;
    return RT_OK;
}

rt_status rt_bvh_build_device(rt_scene* scene, const rt_bvh_options* options, rt_bvh** out)
{
    if (!scene || !out) return fail(RT_ERR_INVALID, "rt_bvh_build_device: null argument");
    rt_bvh_options opt { 0, 32, 64 };
    if (options) opt = *options;
    if (opt.max_level < 0 || opt.max_level > RT_MAX_LEVELS - 1) return fail(RT_ERR_LIMIT, "max_level must be in [0, 32]");
    if (opt.sah_binning && opt.sah_bins < 1) return fail(RT_ERR_INVALID, "sah_bins must be positive");
    // the scene's own tables first: that also fixes the device the scene (and this hierarchy) lives on
    rt_status st = upload_scene(scene);
    if (st != RT_OK) return st;
    RT_HIP(hipSetDevice(scene->device));
    DeviceBvh dev;
    bool unsupported = true;
    try {
        const hipError_t e = device_build(scene->host, opt, dev, unsupported);
        if (e == hipErrorOutOfMemory) {
            // no room for the builder's working tables beside what the device already holds: the host builds
            dev.release();
            (void)hipGetLastError();
            unsupported = true;
        } else if (e != hipSuccess) {
            dev.release();
            return hip_fail(e, "rt_bvh_build_device");
        }
    } catch (const std::exception& e) {
        dev.release();
        return fail(RT_ERR_LIMIT, std::string("rt_bvh_build_device: ") + e.what());
    }
    if (unsupported) return rt_bvh_build(scene, options, out);
    *out = nullptr;
    rt_bvh* b = new (std::nothrow) rt_bvh;
    if (!b) {
        dev.release();
        return fail(RT_ERR_INVALID, "out of host memory");
    }
    b->scene = scene;
    b->deviceBuilt = true;
    b->host.numLevels = dev.numLevels;
    b->host.numLeaves = dev.numLeaves;
    b->numRefNodes = dev.numRefNodes;
    b->numRefPrims = dev.numPrims;
    b->nodes.adopt(dev.nodes, dev.numInner * sizeof(NodeRec));
    b->prims.adopt(dev.prims, dev.numPrims * sizeof(PrimRec));
    b->attrs.adopt(dev.attrs, dev.numPrims * sizeof(AttrRec));
    b->primOrder.adopt(dev.primOrder, dev.numPrims * sizeof(uint32_t));
    b->refNodes.adopt(dev.refNodes, dev.numRefNodes * sizeof(rt_node_record));
    b->refPrims.adopt(dev.refPrims, dev.numPrims * sizeof(rt_prim_ref));
    fill_view(b, dev.rootLo, dev.rootHi, dev.rootRef, dev.numPrims, dev.numInner);
    b->uploaded = true;
    *out = b;
    return RT_OK;
}

int32_t rt_bvh_device_sorted(const rt_bvh* b) { return b && b->deviceBuilt ? 1 : 0; }

void rt_bvh_destroy(rt_bvh* b)
{
    if (!b) return;
    b->pool.clear(); // (waits for whatever a call left queued on a set)
    for (DevBuf* d : { &b->nodes, &b->prims, &b->attrs, &b->primOrder, &b->refNodes, &b->refPrims, &b->rays, &b->hits, &b->misc, &b->gatherShare, &b->gatherFrame }) d->release();
    delete b;
}

int32_t rt_bvh_num_levels(const rt_bvh* b) { return b ? b->host.numLevels : 0; }
int32_t rt_bvh_num_leaves(const rt_bvh* b) { return b ? b->host.numLeaves : 0; }
int32_t rt_bvh_num_nodes(const rt_bvh* b) { return b ? int32_t(b->deviceBuilt ? b->numRefNodes : b->host.nodes.size()) : 0; }
int32_t rt_bvh_num_prim_refs(const rt_bvh* b) { return b ? int32_t(b->deviceBuilt ? b->numRefPrims : b->host.prims.size()) : 0; }

rt_status rt_bvh_export(const rt_bvh* b, rt_node_record* nodes, int32_t nNodes, rt_prim_ref* prims, int32_t nPrims)
{
    if (!b) return fail(RT_ERR_INVALID, "rt_bvh_export: null bvh");
    if (nNodes < rt_bvh_num_nodes(b) || nPrims < rt_bvh_num_prim_refs(b)) return fail(RT_ERR_INVALID, "rt_bvh_export: output arrays too small");
    if (b->deviceBuilt) {
        // a hierarchy built on the GPU keeps these lists there
        RT_HIP(hipSetDevice(b->scene->device));
        if (nodes) RT_HIP(hipMemcpy(nodes, b->refNodes.ptr, b->numRefNodes * sizeof(rt_node_record), hipMemcpyDeviceToHost));
        if (prims) RT_HIP(hipMemcpy(prims, b->refPrims.ptr, b->numRefPrims * sizeof(rt_prim_ref), hipMemcpyDeviceToHost));
        return RT_OK;
    }
    if (nodes) std::memcpy(nodes, b->host.nodes.data(), b->host.nodes.size() * sizeof(rt_node_record));
    if (prims) std::memcpy(prims, b->host.prims.data(), b->host.prims.size() * sizeof(rt_prim_ref));
    return RT_OK;
}

rt_status rt_intersect_batch(const rt_bvh* cb, const rt_ray* rays, uint64_t n, const rt_features* features, rt_hit* hits, int32_t devicePtrs, void* stream)
{
    rt_bvh* b = const_cast<rt_bvh*>(cb);
    if (!b || !features || (n && (!rays || !hits))) return fail(RT_ERR_INVALID, "rt_intersect_batch: null argument");
    if (n == 0) return RT_OK;
    std::lock_guard<std::mutex> guard(b->lock);
    {
        rt_status up = upload_bvh(b);
        if (up != RT_OK) return up;
    }
    RT_HIP(hipSetDevice(b->scene->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    FrameParams P;
    std::memset(&P, 0, sizeof P);
    P.scene = b->view;
    P.f = *features;
    P.stats = nullptr;
    const rt_ray* dRays = rays;
    rt_hit* dHits = hits;
    if (!devicePtrs) {
        RT_HIP(b->rays.reserve(n * sizeof(rt_ray)));
        RT_HIP(b->hits.reserve(n * sizeof(rt_hit)));
        RT_HIP(hipMemcpyAsync(b->rays.ptr, rays, n * sizeof(rt_ray), hipMemcpyHostToDevice, s));
        dRays = static_cast<const rt_ray*>(b->rays.ptr);
        dHits = static_cast<rt_hit*>(b->hits.ptr);
    }
    RT_HIP(launch_intersect(P, dRays, n, dHits, s));
    if (!devicePtrs) {
        RT_HIP(hipMemcpyAsync(hits, dHits, n * sizeof(rt_hit), hipMemcpyDeviceToHost, s));
        RT_HIP(hipStreamSynchronize(s));
    }
    return RT_OK;
}

rt_status rt_visibility_batch(const rt_bvh* cb, const rt_ray* rays, uint64_t n, const float* samplePositions, int32_t nSamples, const rt_features* features, uint8_t* out)
{
    rt_bvh* b = const_cast<rt_bvh*>(cb);
    if (!b || !features || (n && nSamples && (!rays || !samplePositions || !out)) || nSamples < 0) return fail(RT_ERR_INVALID, "rt_visibility_batch: bad argument");
    if (n == 0 || nSamples == 0) return RT_OK;
    std::lock_guard<std::mutex> guard(b->lock);
    {
        rt_status up = upload_bvh(b);
        if (up != RT_OK) return up;
    }
    RT_HIP(hipSetDevice(b->scene->device));
    FrameParams P;
    std::memset(&P, 0, sizeof P);
    P.scene = b->view;
    P.f = *features;
    const size_t outBytes = size_t(n) * size_t(nSamples);
    const size_t sampleBytes = size_t(nSamples) * 12;
    RT_HIP(b->rays.reserve(n * sizeof(rt_ray)));
    RT_HIP(b->misc.reserve(((outBytes + 63) / 64) * 64 + sampleBytes));
    uint8_t* dOut = static_cast<uint8_t*>(b->misc.ptr);
    float* dSamples = reinterpret_cast<float*>(dOut + ((outBytes + 63) / 64) * 64);
    RT_HIP(hipMemcpy(b->rays.ptr, rays, n * sizeof(rt_ray), hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(dSamples, samplePositions, sampleBytes, hipMemcpyHostToDevice));
    RT_HIP(launch_visibility(P, static_cast<const rt_ray*>(b->rays.ptr), n, dSamples, nSamples, dOut, nullptr));
    RT_HIP(hipMemcpy(out, dOut, outBytes, hipMemcpyDeviceToHost));
    return RT_OK;
}

namespace {
rt_status render_rows(const rt_bvh* cb, const rt_camera* camera, const rt_light* lights, int32_t nLights, const rt_features* features,
    const rt_params* params, int32_t width, int32_t height, int32_t rowsLocal, int32_t rowTop, int32_t rowBlock, int32_t blockStride,
    float* outRgb, int32_t devicePtr, void* stream, rt_stats* stats);
}

rt_status rt_render(const rt_bvh* cb, const rt_camera* camera, const rt_light* lights, int32_t nLights, const rt_features* features,
    const rt_params* params, int32_t width, int32_t height, int32_t rowBegin, int32_t rowEnd, float* outRgb, int32_t devicePtr, void* stream, rt_stats* stats)
{
    if (width <= 0 || height <= 0 || rowBegin < 0 || rowEnd > height || rowBegin > rowEnd) return fail(RT_ERR_INVALID, "rt_render: bad frame or row range");
    const int32_t rows = rowEnd - rowBegin;
    return render_rows(cb, camera, lights, nLights, features, params, width, height, rows, height - rowEnd, rows > 0 ? rows : 1, 1, outRgb, devicePtr, stream, stats);
}

rt_status rt_render_blocks(const rt_bvh* cb, const rt_camera* camera, const rt_light* lights, int32_t nLights, const rt_features* features,
    const rt_params* params, int32_t width, int32_t height, int32_t rowBlock, int32_t firstBlock, int32_t blockStride, float* outRgb, int32_t devicePtr,
    void* stream, rt_stats* stats)
{
    if (width <= 0 || height <= 0 || rowBlock <= 0 || firstBlock < 0 || blockStride <= 0) return fail(RT_ERR_INVALID, "rt_render_blocks: bad frame or block layout");
    // blocks firstBlock, firstBlock + blockStride, ... of rowBlock Screen rows each; only the frame's last block can be short
    const int64_t nBlocks = (int64_t(height) + rowBlock - 1) / rowBlock;
    int64_t rows = 0;
    for (int64_t blk = firstBlock; blk < nBlocks; blk += blockStride) {
        const int64_t top = blk * rowBlock, bottom = top + rowBlock < height ? top + rowBlock : height;
        rows += bottom - top;
    }
    return render_rows(cb, camera, lights, nLights, features, params, width, height, int32_t(rows), firstBlock * rowBlock, rowBlock, blockStride, outRgb, devicePtr, stream, stats);
}

rt_status rt_render_blocks_gather(const rt_bvh* cb, const rt_camera* camera, const rt_light* lights, int32_t nLights, const rt_features* features,
    const rt_params* params, int32_t width, int32_t height, int32_t rowBlock, rt_comm* comm, int32_t root, float* frame, int32_t devicePtr, void* stream,
    rt_stats* stats)
{
    rt_bvh* b = const_cast<rt_bvh*>(cb);
    if (!b || !comm) return fail(RT_ERR_INVALID, "rt_render_blocks_gather: null argument");
    const int32_t rank = rt_comm_rank(comm), world = rt_comm_world(comm);
    if (width <= 0 || height <= 0 || rowBlock <= 0 || root < 0 || root >= world) return fail(RT_ERR_INVALID, "rt_render_blocks_gather: bad frame, block layout or root");
    if (rank == root && !frame) return fail(RT_ERR_INVALID, "rt_render_blocks_gather: the root needs a frame");
    const int64_t nBlocks = (int64_t(height) + rowBlock - 1) / rowBlock;
    int64_t rows = 0;
    for (int64_t blk = rank; blk < nBlocks; blk += world) rows += (blk * rowBlock + rowBlock < height ? rowBlock : height - blk * rowBlock);
    // (calls on one hierarchy take turns here: they share its two buffers)
    std::lock_guard<std::mutex> turn(b->gatherLock);
    {
        std::lock_guard<std::mutex> guard(b->lock);
        rt_status up = upload_bvh(b);
        if (up != RT_OK) return up;
    }
    RT_HIP(hipSetDevice(b->scene->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t frameBytes = size_t(width) * size_t(height) * 12;
    RT_HIP(b->gatherShare.reserve(size_t(rows > 0 ? rows : 1) * size_t(width) * 12));
    float* share = static_cast<float*>(b->gatherShare.ptr);
    float* frameDev = nullptr;
    if (rank == root) {
        frameDev = frame;
        if (!devicePtr) {
            RT_HIP(b->gatherFrame.reserve(frameBytes));
            frameDev = static_cast<float*>(b->gatherFrame.ptr);
        }
    }
    rt_stats local;
    if (rows > 0) {
        const rt_status st = rt_render_blocks(cb, camera, lights, nLights, features, params, width, height, rowBlock, rank, world, share, 1, stream, stats ? stats : &local);
        if (st != RT_OK) return st; // (the other ranks then wait in their gather: the caller has to tear the job down)
    } else if (stats) {
        std::memset(stats, 0, sizeof *stats);
    }
    const rt_status g = rt_gather_blocks(comm, root, share, frameDev, width, height, rowBlock, stream);
    if (g != RT_OK) return g;
    if (rank == root && !devicePtr) RT_HIP(hipMemcpyAsync(frame, frameDev, frameBytes, hipMemcpyDeviceToHost, s));
    RT_HIP(hipStreamSynchronize(s)); // the share buffer (and a host frame) must be done with when the call returns
    return RT_OK;
}

namespace {

// Takes a scratch set of `b` for a call that will queue its work on `stream`: one nobody holds and whose last call has
// finished or was queued on the same stream (stream order then keeps the two apart); otherwise a new one, up to
// MAX_SCRATCH_SETS; otherwise it waits for one.  Released by the lease's destructor.
struct ScratchLease {
    rt_bvh* b = nullptr;
    RenderScratch* set = nullptr;
    hipStream_t stream = nullptr;
    bool waited = false; // the call has waited for its work: the set is idle when it is given back
    ~ScratchLease()
    {
        if (!set) return;
        std::lock_guard<std::mutex> guard(b->poolLock);
        set->pending = !waited && hipEventRecord(set->done, stream) == hipSuccess;
        set->lastStream = stream;
        set->busy = false;
        b->poolFree.notify_one();
    }
};
rt_status acquire_scratch(rt_bvh* b, hipStream_t stream, bool ownStream, ScratchLease& lease)
{
    std::unique_lock<std::mutex> guard(b->poolLock);
    lease.b = b;
    while (true) {
        RenderScratch* idle = nullptr;
        RenderScratch* queued = nullptr;
        for (auto& s : b->pool) {
            if (s->busy) continue;
            if (s->pending && !ownStream && s->lastStream == stream) s->pending = false; // the new work queues up behind the old
            if (s->pending && hipEventQuery(s->done) == hipSuccess) s->pending = false;
            if (!s->pending) {
                idle = s.get();
                break;
            }
            queued = s.get();
        }
        if (!idle && int(b->pool.size()) < int(rt_bvh::MAX_SCRATCH_SETS)) {
            auto fresh = std::make_unique<RenderScratch>();
            const hipError_t e = fresh->init();
            if (e != hipSuccess) return hip_fail(e, "rt_render: creating the events and streams of a scratch set");
            idle = fresh.get();
            b->pool.push_back(std::move(fresh));
        }
        if (!idle && queued) {
            RT_HIP(hipEventSynchronize(queued->done));
            queued->pending = false;
            idle = queued;
        }
        if (idle) {
            idle->busy = true;
            lease.set = idle;
            lease.stream = ownStream ? idle->ownStream : stream;
            return RT_OK;
        }
        b->poolFree.wait(guard); // every set is held by a running call
    }
}

rt_status render_rows(const rt_bvh* cb, const rt_camera* camera, const rt_light* lights, int32_t nLights, const rt_features* features,
    const rt_params* params, int32_t width, int32_t height, int32_t rowsLocal, int32_t rowTop, int32_t rowBlock, int32_t blockStride,
    float* outRgb, int32_t devicePtr, void* stream, rt_stats* stats)
{
    rt_bvh* b = const_cast<rt_bvh*>(cb);
    if (!b || !camera || !features || !params || (nLights && !lights) || nLights < 0) return fail(RT_ERR_INVALID, "rt_render: null argument");
    if (rowsLocal > 0 && !outRgb) return fail(RT_ERR_INVALID, "rt_render: null output");
    if (int64_t(width) * int64_t(height) > (int64_t(1) << 30)) return fail(RT_ERR_LIMIT, "frame larger than 2^30 pixels");
    if (features->environment_mapping && (params->enabled_skybox || params->enabled_refl_map) && b->scene->host.skybox.size() != 6)
        return fail(RT_ERR_INVALID, "environment mapping requested but the scene has no skybox");
    rt_status st = check_limits(b, features, params);
    if (st != RT_OK) return st;
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (rowsLocal == 0) return RT_OK;

    {
        std::lock_guard<std::mutex> guard(b->lock);
        rt_status up = upload_bvh(b);
        if (up != RT_OK) return up;
    }
    RT_HIP(hipSetDevice(b->scene->device));
    // a call that returns host pixels and names no stream gets the stream of its scratch set, so that such calls from
    // several threads overlap on the GPU instead of queueing up on the null stream
    ScratchLease lease;
    if ((st = acquire_scratch(b, static_cast<hipStream_t>(stream), !devicePtr && stream == nullptr, lease)) != RT_OK) return st;
    RenderScratch* w = lease.set;
    hipStream_t s = lease.stream;
    const size_t nPix = size_t(rowsLocal) * size_t(width);

    FrameParams P;
    std::memset(&P, 0, sizeof P);
    SplitAux aux {};
    bool useAux = false;
    P.scene = b->view;
    fill_camera(*camera, P);
    P.width = width;
    P.height = height;
    P.rowsLocal = rowsLocal;
    P.rowTop = rowTop;
    P.rowBlock = rowBlock;
    P.blockStride = blockStride;
    P.f = *features;
    P.p = *params;
    if (nLights > MAX_LIGHTS) return fail(RT_ERR_LIMIT, "at most " + std::to_string(int(MAX_LIGHTS)) + " lights per frame");
    if (nLights <= INLINE_LIGHTS) {
        fill_lights(lights, nLights, *features, *params, P.lights, P.drawsPerHit);
        P.shadowSamplesPerHit = number_shadow_samples(P.lights, nLights, *features);
    } else {
        // more lights than the kernel argument block has room for: the table goes to device memory, ahead of the
        // kernels on the call's stream (the host copy belongs to the scratch set and stays until its next call)
        w->hostLights.resize(size_t(nLights));
        fill_lights(lights, nLights, *features, *params, w->hostLights.data(), P.drawsPerHit);
        P.shadowSamplesPerHit = number_shadow_samples(w->hostLights.data(), nLights, *features);
        RT_HIP(w->lights.reserve(size_t(nLights) * sizeof(LightDev)));
        RT_HIP(hipMemcpyAsync(w->lights.ptr, w->hostLights.data(), size_t(nLights) * sizeof(LightDev), hipMemcpyHostToDevice, s));
        P.lightTable = static_cast<const LightDev*>(w->lights.ptr);
    }
    P.nLights = nLights;
    {
        static const int pairStackLds = [] {
            const char* e = std::getenv("RT_PAIR_STACK_LDS"); // tuning / test knob; default chosen by measurement
            const int v = e ? std::atoi(e) : 384;
            return v < 64 ? 64 : v;
        }();
        P.pairStackLds = pairStackLds;
        static const int primaryBudget = [] {
            const char* e = std::getenv("RT_PRIMARY_BUDGET"); // tuning / test knob; default chosen by measurement
            const int v = e ? std::atoi(e) : 64;
            return v < 1 ? 1 : v;
        }();
        P.primaryBudget = primaryBudget;
        static const int primaryStack = [] {
            const char* e = std::getenv("RT_PRIMARY_STACK"); // tuning / test knob; default chosen by measurement
            const int v = e ? std::atoi(e) : 10;
            return v < 1 ? 1 : v;
        }();
        P.primaryStack = primaryStack < b->host.numLevels ? primaryStack : (b->host.numLevels > 0 ? b->host.numLevels : 1);
        static const int streamGroup = [] {
            const char* e = std::getenv("RT_STREAM_GROUP"); // tuning / test knob; default chosen by measurement
            const int v = e ? std::atoi(e) : 2;
            return (v == 1 || v == 2 || v == 4) ? v : 2;
        }();
        P.streamGroup = streamGroup;
    }

    int spp = 1;
    if (features->multiple_rays) spp = params->ray_multiplier * params->ray_multiplier;
    else if (features->depth_of_field) spp = params->dof_samples + 1;
    P.samplesPerPixel = spp;
    if (spp > 1) {
        RT_HIP(w->samples.reserve(nPix * size_t(spp) * 12));
        P.samples = static_cast<float*>(w->samples.ptr);
    }
    if (features->depth_of_field && !features->multiple_rays) {
        RT_HIP(w->dofFlags.reserve(nPix));
        P.dofFlags = static_cast<uint8_t*>(w->dofFlags.ptr);
    } else {
        P.f.depth_of_field = 0; // the reference ignores it when supersampling is on (render.cpp:96-98)
    }
    {
        // Hit queue of the queue passes: one 16-byte entry per (pixel, sample) of a pass.  Every entry has a visibility
        // byte and a 4-byte work item per shadow-casting light sample, so the number of entries is bounded by the light
        // sampling too: items = entries x samples per hit stays below 2^32 (item numbers are 32 bits wide) and below
        // 2^30 (5 GB of visibility bytes and work items).  A pass takes whole sample planes, so the queue holds at least
        // one entry per pixel; frames beyond that are refused below.
        const size_t perHit = size_t(P.shadowSamplesPerHit > 0 ? P.shadowSamplesPerHit : 1);
        const size_t itemBudget = size_t(1) << 30;
        const size_t want = nPix * size_t(spp);
        size_t cap = want < (size_t(64) << 20) ? want : (size_t(64) << 20);
        if (cap > itemBudget / perHit) cap = itemBudget / perHit;
        if (cap < nPix) cap = nPix;
        // Wavefront evaluation of bounce features: node buffers for the call trees of a chunk of pixel samples, and a
        // hit queue with an entry per node.  By default only frames with glossy reflection take it (their fan-out is what
        // a lane of the general kernel cannot keep up with); RT_WAVEFRONT=1 sends every frame with a bounce feature
        // through it, RT_WAVEFRONT=0 none.
        static const int wavefrontMode = [] {
            const char* e = std::getenv("RT_WAVEFRONT");
            return e ? std::atoi(e) : -1;
        }();
        static const long forcedNodes = [] {
            const char* e = std::getenv("RT_WAVEFRONT_NODES"); // test knob: a small node buffer makes chunks overflow and split
            return e ? std::atol(e) : 0L;
        }();
        const bool bounce = features->recursive || features->glossy || features->transparency || (features->environment_mapping && params->enabled_refl_map);
        // Glossy frames run as wavefront passes (measured, 1920x1080, 36 spp, 64 samples per glossy call, node buffer
        // large enough for the frame to be one chunk: TransparentCornell 13.9 ms against 14.9 ms per lane, TransparentCubes
        // 8.7 / 8.9; where lighting casts shadow rays or most surfaces are specular the per-lane kernel is out of the
        // question: a lane traces a call's samples and their shadow rays one after the other, the dragon-class mesh under
        // the area light took 170 s and more per quarter-size frame, 21 s in passes).
        if (bounce && (wavefrontMode == 1 || (wavefrontMode < 0 && features->glossy))) {
            // At most 96 Mi nodes (a 1080p frame at 36 samples per pixel is then a single chunk; about 130 bytes per node
            // plus 10 per shadow-casting light sample, in node records, hit queue and shadow work items), no more than a
            // sixth of the device memory that is free, and no more than the hit queue may hold entries for: a chunk has at
            // most `nodes` roots and `nodes` nodes, each with at most one hit entry.
            static const long nodeLimitMi = [] {
                const char* e = std::getenv("RT_WAVEFRONT_NODE_LIMIT"); // Mi nodes
                return e ? std::atol(e) : 96L;
            }();
            size_t nodeLimit = size_t(nodeLimitMi < 1 ? 1 : nodeLimitMi) << 20;
            {
                size_t freeBytes = 0, totalBytes = 0;
                const size_t perNode = 132 + 10 * perHit;
                if (w->wfRays.cap == 0 && hipMemGetInfo(&freeBytes, &totalBytes) == hipSuccess && nodeLimit > freeBytes / 6 / perNode) nodeLimit = freeBytes / 6 / perNode;
                if (nodeLimit < (size_t(1) << 20)) nodeLimit = size_t(1) << 20;
            }
            if (nodeLimit > itemBudget / perHit / 2) nodeLimit = itemBudget / perHit / 2;
            size_t nodes = 4 * want;
            if (nodes < (size_t(1) << 20)) nodes = size_t(1) << 20;
            // room for the largest call tree one pixel sample can have (filter^depth glossy samples at the deepest level;
            // see-through continuations between the glossy levels add a few each), twice over, if the limit allows: a
            // chunk of a single sample then always fits.  Where it does not and a sample's tree really grows beyond the
            // buffer, the pass is handed to the per-lane kernel (launch_render).
            size_t worst = 1, level = 1;
            if (features->glossy)
                for (int d = 0; d < params->ray_depth && worst <= nodeLimit; d++) {
                    level *= size_t(params->glossy_filter_size > 0 ? params->glossy_filter_size : 1);
                    worst += level;
                }
            if (nodes < 2 * worst) nodes = 2 * worst;
            if (nodes > nodeLimit) nodes = nodeLimit;
            if (forcedNodes > 0) nodes = size_t(forcedNodes < 256 ? 256 : forcedNodes);
            const size_t calls = nodes / size_t(params->glossy_filter_size > 0 ? params->glossy_filter_size : 1) + 64;
            RT_HIP(w->wfRays.reserve(nodes * sizeof(WfRay)));
            RT_HIP(w->wfNodes.reserve(nodes * sizeof(WfNode)));
            RT_HIP(w->wfMeta.reserve(nodes * sizeof(uint32_t)));
            RT_HIP(w->wfValue.reserve(nodes * 12));
            RT_HIP(w->wfHits.reserve(nodes * sizeof(uint32_t)));
            RT_HIP(w->wfHitNode.reserve(2 * nodes * sizeof(uint32_t))); // one per hit-queue entry
            RT_HIP(w->wfGlossy.reserve(calls * sizeof(WfGlossy)));
            P.wfRays = static_cast<WfRay*>(w->wfRays.ptr);
            P.wfNodes = static_cast<WfNode*>(w->wfNodes.ptr);
            P.wfMeta = static_cast<uint32_t*>(w->wfMeta.ptr);
            P.wfValue = static_cast<float*>(w->wfValue.ptr);
            P.wfHits = static_cast<uint32_t*>(w->wfHits.ptr);
            P.wfHitNode = static_cast<uint32_t*>(w->wfHitNode.ptr);
            P.wfGlossy = static_cast<WfGlossy*>(w->wfGlossy.ptr);
            P.wfCapacity = uint32_t(nodes);
            P.wfGlossyCapacity = uint32_t(calls);
            P.wfReport = w->wfReport;
            if (cap < 2 * nodes) cap = 2 * nodes;
        }
        const size_t items = cap * perHit;
        if (items >= (size_t(1) << 32)) return fail(RT_ERR_LIMIT, "too many shadow samples per pass (pixels x light samples per hit must stay below 2^32)");
        RT_HIP(w->queue.reserve(cap * sizeof(HitEntry)));
        RT_HIP(w->queueCounters.reserve(3 * Q_COUNT * sizeof(unsigned long long)));
        RT_HIP(w->gaveUpQueue.reserve(cap * sizeof(uint32_t)));
        RT_HIP(w->visibility.reserve(items));
        RT_HIP(w->shadowQueue.reserve(items * sizeof(uint32_t)));
        P.shadowQueue = static_cast<uint32_t*>(w->shadowQueue.ptr);
        if (P.shadowSamplesPerHit > 0) {
            RT_HIP(w->shadeList.reserve(cap * sizeof(uint32_t)));
            P.shadeList = static_cast<uint32_t*>(w->shadeList.ptr);
        }
        rt::size_pair_stack(P);
        if (P.pairSpillEntries > 0) {
            RT_HIP(w->pairSpill.reserve(size_t(P.pairSpillWaves) * P.pairSpillEntries * sizeof(uint32_t)));
            P.pairSpill = static_cast<uint32_t*>(w->pairSpill.ptr);
        }
        P.queue = static_cast<HitEntry*>(w->queue.ptr);
        // counter blocks: primary pass, lighting round A, lighting round B (one block serves all when the rounds are not split)
        static const int overlap = [] {
            const char* e = std::getenv("RT_OVERLAP"); // 0: the camera rays that gave up are finished before any lighting starts
            return e ? std::atoi(e) : 1;
        }();
        P.primaryCounters = static_cast<unsigned long long*>(w->queueCounters.ptr);
        P.queueCounters = P.primaryCounters + (overlap ? Q_COUNT : 0);
        P.gaveUpQueue = static_cast<uint32_t*>(w->gaveUpQueue.ptr);
        P.shadowQueueItems = uint32_t(items);
        if (overlap) {
            aux.stream = w->auxStream;
            aux.fork = w->evFork;
            aux.join = w->evJoin;
            aux.shadowMarks = stats ? w->evShadowB : nullptr;
            aux.counters = P.primaryCounters + 2 * Q_COUNT;
            aux.pairSpillWaves = 2048; // the second round is small
            if (P.pairSpillEntries > 0) {
                RT_HIP(w->pairSpillAux.reserve(size_t(aux.pairSpillWaves) * P.pairSpillEntries * sizeof(uint32_t)));
                aux.pairSpill = static_cast<uint32_t*>(w->pairSpillAux.ptr);
            }
            // traversal stacks in device memory for that stream's per-lane kernels (no LDS: they run beside the shadow pass)
            static const int deviceStacks = [] {
                const char* e = std::getenv("RT_AUX_DEVICE_STACKS"); // 0: those kernels keep their stacks in LDS
                return e ? std::atoi(e) : 1;
            }();
            if (deviceStacks) {
                aux.laneStackWaves = 2048;
                const size_t depth = size_t(b->host.numLevels > 0 ? b->host.numLevels : 1);
                RT_HIP(w->laneStacks.reserve(size_t(aux.laneStackWaves) * depth * 64 * sizeof(uint2)));
                aux.laneStacks = static_cast<uint2*>(w->laneStacks.ptr);
            }
            useAux = true;
        }
        P.visibility = static_cast<uint8_t*>(w->visibility.ptr);
        P.queueCapacity = cap;
    }
    float* dOut = outRgb;
    if (!devicePtr) {
        RT_HIP(w->frame.reserve(nPix * 12));
        dOut = static_cast<float*>(w->frame.ptr);
    }
    P.out = dOut;
    // three blocks of counters: the call's, those of a wavefront chunk not yet known to stand, a copy kept across a pass
    RT_HIP(w->stats.reserve(3 * STAT_COUNT * sizeof(unsigned long long)));
    P.stats = static_cast<unsigned long long*>(w->stats.ptr);
    P.countTests = stats != nullptr && params->count_tests != 0;
    RT_HIP(hipMemsetAsync(w->stats.ptr, 0, 3 * STAT_COUNT * sizeof(unsigned long long), s));
    // a synchronous wait is only needed when the caller wants numbers or host pixels back
    const bool wait = stats != nullptr || !devicePtr;
    if (stats) RT_HIP(hipEventRecord(w->evStart, s));
    bool splitRecorded = false, shadowMarksRecorded = false;
    unsigned long long wavefrontNodes = 0;
    RT_HIP(launch_render(P, s, stats ? w->evMid : nullptr, stats ? w->evSplit : nullptr, &splitRecorded, &wavefrontNodes, useAux ? &aux : nullptr, &shadowMarksRecorded));
    if (stats) RT_HIP(hipEventRecord(w->evStop, s));
    if (!devicePtr) RT_HIP(hipMemcpyAsync(outRgb, dOut, nPix * 12, hipMemcpyDeviceToHost, s));
    if (wait) {
        RT_HIP(hipStreamSynchronize(s));
        lease.waited = true;
    }
    if (stats) {
        unsigned long long h[STAT_COUNT];
        RT_HIP(hipMemcpy(h, w->stats.ptr, sizeof h, hipMemcpyDeviceToHost));
        stats->primary_rays = h[STAT_PRIMARY];
        stats->shadow_rays = h[STAT_SHADOW];
        stats->bounce_rays = h[STAT_BOUNCE];
        stats->probe_rays = h[STAT_PROBE];
        stats->box_tests = h[STAT_BOX];
        stats->triangle_tests = h[STAT_TRI];
        stats->transparency_overflows = h[STAT_OVERFLOW];
        RT_HIP(hipEventElapsedTime(&stats->kernel_ms, w->evStart, w->evMid));
        RT_HIP(hipEventElapsedTime(&stats->resolve_ms, w->evMid, w->evStop));
        stats->shadow_box_tests = h[STAT_SHADOW_BOX];
        stats->shadow_triangle_tests = h[STAT_SHADOW_TRI];
        stats->culled_shadow_rays = h[STAT_CULLED];
        stats->wavefront_nodes = wavefrontNodes;
        stats->stream_packets = h[STAT_STREAM_PACKETS];
        stats->stream_box_steps = h[STAT_STREAM_BOX_STEPS];
        stats->stream_prim_steps = h[STAT_STREAM_PRIM_STEPS];
        stats->stream_entries = h[STAT_STREAM_ENTRIES];
        if (std::getenv("RT_DEBUG_STATS")) {
            std::fprintf(stderr, "rt_render debug slots:");
            for (int i = STAT_DEBUG0; i < STAT_DEBUG_END; i++) std::fprintf(stderr, " %llu", h[i]);
            std::fprintf(stderr, "\n");
        }
        if (splitRecorded) {
            RT_HIP(hipEventElapsedTime(&stats->primary_ms, w->evStart, w->evSplit[0]));
            RT_HIP(hipEventElapsedTime(&stats->cull_ms, w->evSplit[0], w->evSplit[1]));
            RT_HIP(hipEventElapsedTime(&stats->shadow_ms, w->evSplit[1], w->evSplit[2]));
            RT_HIP(hipEventElapsedTime(&stats->shade_ms, w->evSplit[2], w->evSplit[3]));
            if (shadowMarksRecorded) RT_HIP(hipEventElapsedTime(&stats->shadow_b_ms, w->evShadowB[0], w->evShadowB[1]));
        }
    }
    return RT_OK;
}
} // namespace

rt_status rt_bloom(float* pixels, int32_t width, int32_t height, const rt_params* params, int32_t devicePtr, void* stream)
{
    if (!pixels || !params || width <= 0 || height <= 0) return fail(RT_ERR_INVALID, "rt_bloom: bad argument");
    BloomParams bp;
    rt_status st = fill_bloom(*params, bp);
    if (st != RT_OK) return st;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t bytes = size_t(width) * size_t(height) * 12;
    DeviceState* d = nullptr;
    if ((st = device_state(&d)) != RT_OK) return st;
    // The filter reads neighbours of every pixel, so it cannot run in place: the frame is copied to a buffer this
    // device keeps between calls (grown when a larger frame arrives) and filtered from there back into `pixels`.
    // The buffer is shared by all calls on the device; an event orders a call behind the previous one's kernel
    // when the two are on different streams.
    std::lock_guard<std::mutex> guard(d->lock);
    RT_HIP(d->bloomSrc.reserve(bytes));
    if (!devicePtr) RT_HIP(d->bloomDst.reserve(bytes));
    if (!d->bloomDone) RT_HIP(hipEventCreateWithFlags(&d->bloomDone, hipEventDisableTiming));
    else RT_HIP(hipStreamWaitEvent(s, d->bloomDone, 0));
    float* src = static_cast<float*>(d->bloomSrc.ptr);
    float* target = devicePtr ? pixels : static_cast<float*>(d->bloomDst.ptr);
    RT_HIP(hipMemcpyAsync(src, pixels, bytes, devicePtr ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    RT_HIP(launch_bloom(src, target, width, height, bp, s));
    if (!devicePtr) RT_HIP(hipMemcpyAsync(pixels, target, bytes, hipMemcpyDeviceToHost, s));
    RT_HIP(hipEventRecord(d->bloomDone, s));
    if (!devicePtr) RT_HIP(hipStreamSynchronize(s)); // host pixels must be complete on return
    return RT_OK;
}

uint64_t rt_bloom_static_max_pixels(void) { return RT_STATIC_SCRATCH_BYTES / 12; }

rt_status rt_bloom_static(float* devicePixels, int32_t width, int32_t height, const rt_params* params, void* stream)
{
    if (!devicePixels || !params || width <= 0 || height <= 0) return fail(RT_ERR_INVALID, "rt_bloom_static: bad argument");
    const size_t bytes = size_t(width) * size_t(height) * 12;
    if (bytes > RT_STATIC_SCRATCH_BYTES)
        return fail(RT_ERR_LIMIT, "rt_bloom_static: frame of " + std::to_string(size_t(width) * size_t(height)) + " pixels exceeds the static scratch of " +
                                      std::to_string(size_t(RT_STATIC_SCRATCH_BYTES / 12)) + " pixels; use rt_bloom");
    BloomParams bp;
    rt_status st = fill_bloom(*params, bp);
    if (st != RT_OK) return st;
    DeviceState* d = nullptr;
    if ((st = device_state(&d)) != RT_OK) return st;
    void* scratchPtr = nullptr;
    {
        std::lock_guard<std::mutex> guard(d->lock);
        if ((st = resolve_scratch(*d)) != RT_OK) return st;
        scratchPtr = d->scratch;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    // frame -> this device's copy of the static array -> filtered back into the frame; nothing needs to start at zero
    // (the copy overwrites every byte the kernel reads)
    RT_HIP(hipMemcpyAsync(scratchPtr, devicePixels, bytes, hipMemcpyDeviceToDevice, s));
    RT_HIP(launch_bloom(static_cast<const float*>(scratchPtr), devicePixels, width, height, bp, s));
    return RT_OK;
}

} // extern "C"
