// The per-level sorts of the hierarchy build on the GPU.
//
// The reference's builder (src/bounding_volume_hierarchy.cpp:106-208, median variant) sorts the primitives of a node by
// centroid coordinate on axis `level % 3` and hands the lower half (count / 2 primitives) to the left child.  Which
// primitives end up in which half depends only on the keys, not on the sorting algorithm, unless equal keys straddle
// the cut; and with one primitive per leaf the ranges of all nodes follow from the primitive count alone.  So the
// whole build is a sequence of SEGMENTED sorts, one per level, the segments being the node ranges of that level: done
// here as one radix sort per level on (segment number, order-preserving key bits), stable and therefore in general
// NOT in std::sort's order among equal keys, which is why every cut is checked: if the two keys next to a cut compare
// equal (as floats: -0 == +0), which primitives go left is std::sort's choice.  That choice is a function of the order
// the node's primitives arrive in, and that order is known — it is the sorted order on the parent's axis — whenever
// their keys on the parent's axis are pairwise different (the root's arrive in primitive order).  Such a node is sorted
// again on the host with the builder's own std::sort and written back; anything else (arrival order not determined, a
// NaN key, more ambiguous nodes than the list holds) makes the caller fall back to the host build.
#include "bvh_build.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

namespace rt {

namespace {

// node range of `level` that position i belongs to: ranges are halved like the builder does (left = count / 2), a
// range of fewer than two primitives is a leaf and stays as it is.  Returns the segment number (path bits).
__device__ __forceinline__ uint32_t segment_of(uint32_t i, uint32_t n, int level, uint32_t& first, uint32_t& last)
{
    first = 0;
    last = n;
    uint32_t seg = 0;
    for (int l = 0; l < level; l++) {
        seg <<= 1;
        if (last - first < 2u) continue;
        const uint32_t mid = first + (last - first) / 2u;
        if (i < mid) {
            last = mid;
        } else {
            first = mid;
            seg |= 1u;
        }
    }
    return seg;
}

// order-preserving bits of a float for an unsigned radix sort
__device__ __forceinline__ uint32_t ordered_bits(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ void k_level_keys(const float* __restrict__ centroids, const int* __restrict__ order, uint32_t n, int level, unsigned long long* __restrict__ keys, int* flags)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t first, last;
    const uint32_t seg = segment_of(i, n, level, first, last);
    const float c = centroids[size_t(order[i]) * 3 + size_t(level % 3)];
    if (c != c) flags[1] = 1; // a NaN coordinate: the host decides
    keys[i] = ((unsigned long long)seg << 32) | ordered_bits(c);
}

// after the sort of `level`: every cut (the middle of a range of two or more) must separate different keys
constexpr int MAX_TIED_NODES = 4096;

// after the sort of `level`: the nodes whose cut (the middle of a range of two or more) separates equal keys
__global__ void k_level_cuts(const float* __restrict__ centroids, const int* __restrict__ order, uint32_t n, int level, int* flags, uint2* tied)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || i == 0) return;
    uint32_t first, last;
    (void)segment_of(i, n, level, first, last);
    if (last - first < 2u || i != first + (last - first) / 2u) return;
    const int axis = level % 3;
    if (centroids[size_t(order[i - 1]) * 3 + size_t(axis)] == centroids[size_t(order[i]) * 3 + size_t(axis)]) {
        const int at = atomicAdd(&flags[0], 1);
        if (at < MAX_TIED_NODES) tied[at] = make_uint2(first, last);
    }
}

__global__ void k_iota(int* order, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) order[i] = int(i);
}

struct Buffers {
    float* centroids = nullptr;
    int *orderA = nullptr, *orderB = nullptr, *flags = nullptr;
    unsigned long long *keysA = nullptr, *keysB = nullptr;
    uint2* tied = nullptr;
    void* temp = nullptr;
    ~Buffers()
    {
        for (void* p : { (void*)centroids, (void*)orderA, (void*)orderB, (void*)flags, (void*)keysA, (void*)keysB, (void*)tied, temp })
            if (p) (void)hipFree(p);
    }
};

} // namespace

hipError_t device_split_order(const float* centroids, size_t count, std::vector<int>& order, bool& ambiguous, int* hostFixed)
{
    ambiguous = false;
    if (hostFixed) *hostFixed = 0;
    order.resize(count);
    if (count == 0) return hipSuccess;
    if (count >= (size_t(1) << 31)) {
        ambiguous = true;
        return hipSuccess;
    }
    const uint32_t n = uint32_t(count);
    int levels = 0; // levels that have a range of two or more primitives
    while ((size_t(1) << levels) < count) levels++;
    Buffers b;
    hipError_t e;
#define RT_TRY(call) do { if ((e = (call)) != hipSuccess) return e; } while (0)
    RT_TRY(hipMalloc(&b.centroids, count * 12));
    RT_TRY(hipMalloc(&b.orderA, count * sizeof(int)));
    RT_TRY(hipMalloc(&b.orderB, count * sizeof(int)));
    RT_TRY(hipMalloc(&b.keysA, count * sizeof(unsigned long long)));
    RT_TRY(hipMalloc(&b.keysB, count * sizeof(unsigned long long)));
    RT_TRY(hipMalloc(&b.flags, 2 * sizeof(int)));
    RT_TRY(hipMalloc(&b.tied, MAX_TIED_NODES * sizeof(uint2)));
    RT_TRY(hipMemcpy(b.centroids, centroids, count * 12, hipMemcpyHostToDevice));
    RT_TRY(hipMemset(b.flags, 0, 2 * sizeof(int)));
    size_t tempBytes = 0;
    RT_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tempBytes, b.keysA, b.keysB, b.orderA, b.orderB, int(n), 0, 64));
    RT_TRY(hipMalloc(&b.temp, tempBytes ? tempBytes : 16));
    const dim3 block(256), grid((n + 255u) / 256u);
    hipLaunchKernelGGL(k_iota, grid, block, 0, nullptr, b.orderA, n);
    int *cur = b.orderA, *next = b.orderB;
    for (int level = 0; level < levels; level++) {
        hipLaunchKernelGGL(k_level_keys, grid, block, 0, nullptr, b.centroids, cur, n, level, b.keysA, b.flags);
        RT_TRY(hipcub::DeviceRadixSort::SortPairs(b.temp, tempBytes, b.keysA, b.keysB, cur, next, int(n), 0, 32 + (level > 0 ? level : 1)));
        int* t = cur;
        cur = next;
        next = t; // (still holds the order the level's nodes received their primitives in)
        hipLaunchKernelGGL(k_level_cuts, grid, block, 0, nullptr, b.centroids, cur, n, level, b.flags, b.tied);
        RT_TRY(hipGetLastError());
        int flags[2] = { 0, 0 };
        RT_TRY(hipMemcpy(flags, b.flags, sizeof flags, hipMemcpyDeviceToHost));
        if (flags[1] || flags[0] > MAX_TIED_NODES) {
            ambiguous = true;
            return hipSuccess;
        }
        if (flags[0] > 0) {
            std::vector<uint2> tied;
            tied.resize(size_t(flags[0]));
            RT_TRY(hipMemcpy(tied.data(), b.tied, tied.size() * sizeof(uint2), hipMemcpyDeviceToHost));
            std::vector<int> arrived;
            for (const uint2& node : tied) {
                const size_t first = node.x, len = size_t(node.y - node.x);
                arrived.resize(len);
                RT_TRY(hipMemcpy(arrived.data(), next + first, len * sizeof(int), hipMemcpyDeviceToHost));
                if (level > 0) {
                    // the arrival order is the sorted order on the parent's axis only if no two keys there are equal
                    const int parentAxis = (level - 1) % 3;
                    for (size_t i = 0; i + 1 < len; i++)
                        if (!(centroids[size_t(arrived[i]) * 3 + size_t(parentAxis)] < centroids[size_t(arrived[i + 1]) * 3 + size_t(parentAxis)])) {
                            ambiguous = true;
                            return hipSuccess;
                        }
                }
                reference_sort(arrived.data(), len, centroids, level % 3);
                RT_TRY(hipMemcpy(cur + first, arrived.data(), len * sizeof(int), hipMemcpyHostToDevice));
                if (hostFixed) (*hostFixed)++;
            }
            RT_TRY(hipMemset(b.flags, 0, sizeof(int)));
        }
    }
    RT_TRY(hipMemcpy(order.data(), cur, count * sizeof(int), hipMemcpyDeviceToHost));
#undef RT_TRY
    return hipSuccess;
}

} // namespace rt
