// Launch wrappers around the HIP kernels (defined in rt_kernels.hip); called by the C ABI in rt_api.cpp.
#pragma once
#include "rt_device.h"

#include <hip/hip_runtime.h>

namespace rt {

enum { STAT_PRIMARY = 0, STAT_SHADOW, STAT_BOUNCE, STAT_PROBE, STAT_BOX, STAT_TRI, STAT_OVERFLOW, STAT_SHADOW_BOX, STAT_SHADOW_TRI, STAT_CULLED,
    STAT_STREAM_PACKETS, STAT_STREAM_BOX_STEPS, STAT_STREAM_PRIM_STEPS, STAT_STREAM_ENTRIES,
    STAT_DEBUG0, STAT_DEBUG_END = STAT_DEBUG0 + 8, // slots for timing builds of a kernel (printed when RT_DEBUG_STATS is set)
    STAT_COUNT = STAT_DEBUG_END };

// Renders the P.rowsLocal local rows of the call: the traversal/shading passes, then (more than one sample per pixel)
// the resolve pass.  `afterPasses`, when not null, is recorded between the two.  `split`, when not null, points
// to four events recorded after the primary, cull, shadow and shade kernels if the call runs as those three passes
// in a single chunk; *splitRecorded says whether that happened.
// *wavefrontNodes (when not null) receives the number of call-tree nodes the wavefront passes evaluated (0 when the
// call did not run as wavefront passes).
// `aux` (when not null): what a direct-lighting pass needs to finish the camera rays that gave up, and light their
// hits, on a second stream beside the lighting of the other hits: that stream, two events, a counter block (the third
// of the three blocks that start at P.primaryCounters; the second is P.queueCounters) and spill space for its shadow pass.
struct SplitAux {
    hipStream_t stream;
    hipEvent_t fork, join;
    hipEvent_t* shadowMarks; // null, or two events recorded on `stream` around the second round's shadow kernel
    unsigned long long* counters;
    uint32_t* pairSpill;
    uint32_t pairSpillWaves;
    uint2* laneStacks;       // traversal stacks in device memory for that stream's per-lane kernels (FrameParams::laneStacks)
    uint32_t laneStackWaves;
};
hipError_t launch_render(const FrameParams& P, hipStream_t stream, hipEvent_t afterPasses, hipEvent_t* split, bool* splitRecorded,
    unsigned long long* wavefrontNodes = nullptr, const SplitAux* aux = nullptr, bool* shadowMarksRecorded = nullptr);

// Shadow pass sizing: clamps P.pairStackLds to what the hierarchy can need and sets P.pairSpillEntries (per wave)
// and P.pairSpillWaves (the most waves the pass is launched with); the caller provides P.pairSpill accordingly.
void size_pair_stack(FrameParams& P);

// n closest-hit queries; rays/hits are device pointers.
hipError_t launch_intersect(const FrameParams& P, const rt_ray* rays, uint64_t n, rt_hit* hits, hipStream_t stream);

// out[i*nSamples+s] = visibility of samples[s] from hit point i (rays carry the hit distance in t).
hipError_t launch_visibility(const FrameParams& P, const rt_ray* rays, uint64_t n, const float* samples, int nSamples, uint8_t* out, hipStream_t stream);

enum { BLOOM_MAX_SIZE = 16, BLOOM_MAX_TAPS = 2 * BLOOM_MAX_SIZE + 1 };
// The Gaussian filter of reference src/bloom.cpp:14-48 is an outer product normalised by the sum of all products:
// gaus[a][b] = (f[b] * f[a]) / denom with f[i] = exp(-(i - size)^2 / (2 sigma^2)).  The factors travel in the kernel
// argument block and every block forms the (2*size+1)^2 weights in LDS with the same two fp32 operations, so no
// weight table lives in device memory.
struct BloomParams {
    float threshold, scale;
    int32_t size, gaussian, debug;
    float f[BLOOM_MAX_TAPS], denom;
};
// dst = src + scale * filtered(mask(src)) (or just the filtered term in debug mode); src and dst differ.
hipError_t launch_bloom(const float* src, float* dst, int width, int height, const BloomParams& bp, hipStream_t stream);

} // namespace rt
