// fp32 vector arithmetic for host and device code of librt_amd.
//
// Every function spells out the operation order of the reference's vendored glm 0.9.9.9
// (reference framework/third_party/glm/glm/detail/func_geometric.inl:48-55 dot, :68-79 cross, :84-89
// normalize, :104-110 reflect; func_exponential.inl:134-139 inversesqrt = 1/sqrt), because the ray/AABB
// and ray/triangle decisions have to come out bit-identical to the reference's.  Translation units that
// include this header are compiled with -ffp-contract=off and correctly rounded fp32 divide/sqrt.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rt {

struct vec3 {
    float x, y, z;
};

RT_HD vec3 make3(float x, float y, float z) { return vec3 { x, y, z }; }
RT_HD vec3 splat3(float a) { return vec3 { a, a, a }; }
RT_HD vec3 load3(const float* p) { return vec3 { p[0], p[1], p[2] }; }
RT_HD vec3 operator+(vec3 a, vec3 b) { return vec3 { a.x + b.x, a.y + b.y, a.z + b.z }; }
RT_HD vec3 operator-(vec3 a, vec3 b) { return vec3 { a.x - b.x, a.y - b.y, a.z - b.z }; }
RT_HD vec3 operator-(vec3 a) { return vec3 { -a.x, -a.y, -a.z }; }
RT_HD vec3 operator*(vec3 a, vec3 b) { return vec3 { a.x * b.x, a.y * b.y, a.z * b.z }; }
RT_HD vec3 operator*(vec3 a, float s) { return vec3 { a.x * s, a.y * s, a.z * s }; }
RT_HD vec3 operator*(float s, vec3 a) { return vec3 { s * a.x, s * a.y, s * a.z }; }
RT_HD vec3 operator/(vec3 a, float s) { return vec3 { a.x / s, a.y / s, a.z / s }; }
RT_HD float dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RT_HD vec3 cross(vec3 a, vec3 b) { return vec3 { a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y }; }
RT_HD float length(vec3 v) { return sqrtf(dot(v, v)); }
RT_HD vec3 normalize(vec3 v) { return v * (1.0f / sqrtf(dot(v, v))); }
RT_HD vec3 reflect(vec3 I, vec3 N) { return I - N * dot(N, I) * 2.0f; }
// glm::max / glm::min are plain comparisons (their NaN behaviour differs from fmaxf/fminf)
RT_HD float sel_max(float a, float b) { return (a < b) ? b : a; }
RT_HD float sel_min(float a, float b) { return (b < a) ? b : a; }
// std::fmin / std::fmax of the hierarchy build (reference src/bounding_volume_hierarchy.cpp:21-31), spelled out so that
// host and device builders agree on signed zeros: the first argument unless the second is strictly smaller / larger (the
// sign of a zero bound never changes a ray/box decision); a NaN first argument yields the second.
RT_HD float min_first(float x, float y) { return x != x ? y : (y < x ? y : x); }
RT_HD float max_first(float x, float y) { return x != x ? y : (y > x ? y : x); }
// reference src/intersect.cpp:17-20
RT_HD bool is_zero(float a) { return fabsf(a) <= 0.000001f; }
RT_HD float comp(vec3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

// quaternion rotation, glm/detail/type_quat.inl:347-354
struct quat {
    float w, x, y, z;
};
RT_HD vec3 rotate(quat q, vec3 v)
{
    vec3 qv { q.x, q.y, q.z };
    vec3 uv = cross(qv, v);
    vec3 uuv = cross(qv, uv);
    return v + ((uv * q.w) + uuv) * 2.0f;
}

// Counter-based random stream: value = H(H(H(H(seed) ^ pixel) ^ (sample*0x9e3779b9 + stream)) ^ index)
// with H a 32-bit integer finaliser.  Stream 0 replaces the reference's rand() calls in call order (31-bit
// values), stream 1 drives the thin-lens samples, stream 2 the glossy perturbations (indexed from rng_ray_key).
RT_HD uint32_t hash32(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
enum { RNG_STREAM_RAND = 0, RNG_STREAM_DOF = 1, RNG_STREAM_GLOSS = 2 };
RT_HD uint32_t rng_pixel_key(uint32_t seed, uint32_t pixel) { return hash32(hash32(seed) ^ pixel); }
RT_HD uint32_t rng_u32(uint32_t pixelKey, uint32_t sample, uint32_t stream, uint32_t index)
{
    uint32_t h = hash32(pixelKey ^ (sample * 0x9e3779b9U + stream));
    return hash32(h ^ index);
}
// Key of one glossy-reflection call: a hash of the bits of the reflection ray it perturbs (origin, direction).  The
// two Gaussian deviates of the call's sample i are deviates key + 2i and key + 2i + 1 of the glossy stream, so a call's
// samples depend on nothing but the pixel sample and the call's own input, and calls can be evaluated in any order
// (the reference draws them from one global generator in whatever order its threads happen to run).
RT_HD uint32_t rng_bits(float f)
{
    union { float f; uint32_t u; } c;
    c.f = f;
    return c.u;
}
RT_HD uint32_t rng_ray_key(vec3 o, vec3 d)
{
    uint32_t h = hash32(rng_bits(o.x) ^ 0x9e3779b9U);
    h = hash32(h ^ rng_bits(o.y));
    h = hash32(h ^ rng_bits(o.z));
    h = hash32(h ^ rng_bits(d.x));
    h = hash32(h ^ rng_bits(d.y));
    return hash32(h ^ rng_bits(d.z));
}
RT_HD double rng_unit(uint32_t u) { return (double(u) + 0.5) * (1.0 / 4294967296.0); }

} // namespace rt
