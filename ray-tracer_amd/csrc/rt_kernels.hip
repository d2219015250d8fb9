// HIP kernels of librt_amd for gfx950 (MI355X).  Compiled with -ffp-contract=off: the ray/box and
// ray/triangle decisions must round exactly like the reference's scalar fp32 code.
#include "rt_kernels.h"

#include <cstdlib>
#include <vector>
#include <type_traits>

namespace rt {

namespace {

constexpr int BLOCK = 256;  // threads per block of the batch / bloom / resolve kernels
constexpr int WAVE = 64;    // the render kernels run one wave per block: an 8x8 pixel tile at one sample index
constexpr int TILE = 8;
constexpr uint32_t PRIMARY_CURSORS = 16;  // cursors that hand out k_primary's chunks
constexpr uint32_t PRIMARY_CHUNK_MAX = 16; // logical blocks per cursor grab of k_primary when the pass is large

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// set bits of m in the lanes below this one
__device__ __forceinline__ uint32_t lanes_below(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
}

__device__ __forceinline__ void flush_counts(const Counts& c, unsigned long long* stats, bool shadowPass = false)
{
    if (!stats) return;
    const uint32_t vals[7] = { c.primary, c.shadow, c.bounce, c.probe, c.box, c.tri, c.overflow };
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const uint32_t s = wave_sum(vals[i]);
        if ((threadIdx.x & 63) == 0 && s) {
            atomicAdd(&stats[i], (unsigned long long)s);
            if (shadowPass && i == STAT_BOX) atomicAdd(&stats[STAT_SHADOW_BOX], (unsigned long long)s);
            if (shadowPass && i == STAT_TRI) atomicAdd(&stats[STAT_SHADOW_TRI], (unsigned long long)s);
        }
    }
}

// One lane = one (pixel, sample).  A logical block is one wave's worth: an 8x8 pixel tile at one sample index;
// consecutive logical blocks are the samples of one tile, then the next tile.  The render kernels are launched
// with as many one-wave blocks as fit on the chip and each walks over logical blocks (launching one block per
// logical block leaves the chip mostly empty: blocks this short are created more slowly than they finish).
struct LaneId {
    int x, y, sidx;
    size_t pix;
    bool valid;
};
__device__ __forceinline__ LaneId lane_id(const FrameParams& P, uint32_t logical)
{
    const int lane = int(threadIdx.x);
    const int tilesX = (P.width + TILE - 1) / TILE;
    const int nS = P.sampleEnd - P.sampleBegin;
    const int tile = int(logical / uint32_t(nS));
    LaneId id;
    id.sidx = P.sampleBegin + int(logical % uint32_t(nS));
    id.x = (tile % tilesX) * TILE + (lane & 7);
    const int localRow = (tile / tilesX) * TILE + (lane >> 3);
    id.valid = id.x < P.width && localRow < P.rowsLocal;
    id.y = local_row_y(P, id.valid ? localRow : 0);
    id.pix = size_t(localRow) * size_t(P.width) + size_t(id.x);
    return id;
}

// General path: the whole evaluation of a sample (any feature combination) in one lane.
template <bool PRUNE, bool GLOSSY, bool COUNT, bool SPH>
__global__ void __launch_bounds__(WAVE) k_render(const FrameParams P)
{
    extern __shared__ uint2 lds_stack[];
    Stack st { lds_stack + threadIdx.x, uint32_t(WAVE) };
    Counts cnt;
    for (uint32_t logical = blockIdx.x; logical < P.logicalBlocks; logical += gridDim.x) {
        const LaneId id = lane_id(P, logical);
        if (!id.valid) continue;
        bool live = true;
        if (P.f.depth_of_field && id.sidx > 0) live = P.dofFlags[id.pix] != 0;
        if (live) {
            bool dofFlag = false;
            const vec3 c = render_sample<PRUNE, GLOSSY, COUNT, SPH>(P, id.x, id.y, id.sidx, st, cnt, dofFlag);
            float* o = sample_slot(P, id.pix, id.sidx);
            o[0] = c.x;
            o[1] = c.y;
            o[2] = c.z;
            if (P.f.depth_of_field && id.sidx == 0) P.dofFlags[id.pix] = dofFlag ? 1 : 0;
        }
    }
    flush_counts(cnt, P.stats);
}

// The ray whose hit is entry q of the hit queue, and the state of its random streams.  Direct-lighting frames: the
// camera ray, recomputed from the entry's work id.  Wavefront lighting: the ray of the node the entry belongs to; the
// node also knows where on the rand() stream its call's lighting starts.
__device__ __forceinline__ void hit_ray(const FrameParams& P, uint32_t q, const HitEntry& e, uint32_t nPix, Rng& rng, RayD& r, size_t& pix, int& sidx)
{
    sidx = P.sampleBegin + int(e.work / nPix);
    pix = size_t(e.work % nPix);
    const int x = int(pix % size_t(P.width));
    const int y = local_row_y(P, int(pix / size_t(P.width)));
    if (P.wfLighting && P.wfHitNode[q] != 0xffffffffu) {
        const uint32_t node = P.wfHitNode[q];
        const WfRay& w = P.wfRays[node];
        rng.pixelKey = rng_pixel_key(P.p.seed, uint32_t(y * P.width + x));
        rng.glossCall = 0;
        rng.sample = uint32_t(sidx);
        rng.randCounter = P.wfNodes[node].randBase;
        r = make_ray(load3(w.o), load3(w.d));
        return;
    }
    vec3 o, d;
    sample_camera_ray(P, x, y, sidx, rng, o, d);
    r = make_ray(o, d);
}
// where the colour of queued hit q goes: the pixel sample's slot, or the Lo of its node
__device__ __forceinline__ float* hit_colour_slot(const FrameParams& P, uint32_t q, size_t pix, int sidx)
{
    if (P.wfLighting && P.wfHitNode[q] != 0xffffffffu) {
        // a see-through call keeps its Lo for the blend; a call that ends here has Lo as its value
        const uint32_t node = P.wfHitNode[q];
        return wf_kind(P.wfMeta[node]) == WF_BLEND ? P.wfNodes[node].coef : P.wfValue + 3 * size_t(node);
    }
    return sample_slot(P, pix, sidx);
}

// Primary pass.  Three kernels:
//   k_primary   camera ray and closest hit by the reference's ordered, pruned traversal, one lane per ray — with
//               a budget: a ray that has visited P.primaryBudget inner nodes and still found nothing gives up and
//               is queued.  Such rays skim along the surface through thousands of boxes, nothing prunes them,
//               and one of them would hold its whole wave for milliseconds.
//   k_stream    (primary mode, below) answers "does the ray meet any primitive at all" for the queued rays, with
//               the boxes of each ray spread over all lanes.  Most of them meet nothing and are final.  The
//               reference finds a hit exactly when some primitive below boxes the ray enters passes its test
//               (nothing is pruned before the first acceptance), so this changes no result.
//   k_closest   the unbudgeted traversal for the few queued rays that do meet something.
// A miss is final (sky or black); a hit goes to the hit queue, appended with one atomic per wave so that the hits
// of a tile stay together.
// WAVES per SIMD: the pass is bound by the latency of a ray's chain of node visits, which more waves hide — eight of
// them (64 registers, a handful spilled) take the camera pass of the 1080p soft-shadow frame from 1.64 to 1.54 ms;
// with an environment map the rays that miss need the registers more (Monkey frame 0.65 -> 0.80 ms), and so do lens
// rays (4K depth-of-field frame +0.8 %): seven there.
template <bool PRUNE, bool COUNT, int WAVES>
__global__ void __launch_bounds__(WAVE, WAVES) k_primary(const FrameParams P)
{
    extern __shared__ uint2 lds_stack[];
    Stack st { lds_stack + threadIdx.x, uint32_t(WAVE) };
    Counts cnt;
    const int lane = int(threadIdx.x);
    const uint32_t nPix = uint32_t(size_t(P.rowsLocal) * size_t(P.width));
    // Work order.  A wave takes a CHUNK of P.primaryChunk logical blocks per cursor grab (the grab has to wait for the
    // wave's outstanding stores, so it must be rare when there is much to do), and the blocks of chunk c are c, c + n,
    // c + 2n, ... (n = number of chunks): one block out of each of that many horizontal bands of the frame.  Every chunk therefore costs
    // about the same — busy middle and empty margins alike — and the waves finish together.
    // The chunks are handed out by PRIMARY_CURSORS cursors, each over its own slice of the chunk numbers and in a
    // cache line of its own (atomics on one address are served one after the other, about 10 ns each: a single
    // cursor would cap the kernel at its number of grabs times that).  A wave starts at the cursor of its block
    // number and moves on to the next one when a slice is used up.
    const uint32_t chunkBlocks = P.primaryChunk;
    const uint32_t nChunks = (P.logicalBlocks + chunkBlocks - 1) / chunkBlocks;
    const uint32_t perCursor = (nChunks + PRIMARY_CURSORS - 1) / PRIMARY_CURSORS;
    uint32_t cursor = blockIdx.x % PRIMARY_CURSORS, cursorsTried = 0;
    uint32_t chunk = 0, member = chunkBlocks;
    while (true) {
        if (member == chunkBlocks) {
            bool gotChunk = false;
            while (cursorsTried < PRIMARY_CURSORS) {
                uint32_t i = 0;
                if (lane == 0) i = uint32_t(atomicAdd(&P.primaryCounters[Q_PRIMARY_CURSORS + cursor * 16u], 1ull));
                i = __builtin_amdgcn_readfirstlane(i);
                chunk = cursor * perCursor + i;
                if (i < perCursor && chunk < nChunks) {
                    gotChunk = true;
                    break;
                }
                cursor = (cursor + 1u) % PRIMARY_CURSORS;
                cursorsTried++;
            }
            if (!gotChunk) break;
            member = 0;
        }
        const uint32_t logical = chunk + (member++) * nChunks;
        if (logical >= P.logicalBlocks) continue;
        const LaneId id = lane_id(P, logical);
        int slot = -1;
        float t = FLT_MAX;
        // depth of field: the lens rays of a pixel are traced only when its centre ray (sample 0, an earlier pass) said so
        const bool dof = P.f.depth_of_field != 0;
        const bool live = id.valid && !(dof && id.sidx > 0 && P.dofFlags[id.pix] == 0);
        if (live) {
            Rng rng;
            vec3 o, d;
            sample_camera_ray(P, id.x, id.y, id.sidx, rng, o, d);
            const RayD r = make_ray(o, d);
            cnt.primary++;
            Attr attr;
#ifdef RT_SP_HIST
            cnt.overflow = 0;
#endif
            slot = traverse<PRUNE, false, COUNT, false, true>(P.scene, r, t, st, cnt, attr, P.primaryBudget, P.primaryStack);
#ifdef RT_SP_HIST
            if (P.stats) {
                const uint32_t m = cnt.overflow;
                atomicAdd(&P.stats[STAT_DEBUG0 + (m <= 4 ? 0 : m <= 6 ? 1 : m <= 8 ? 2 : m <= 10 ? 3 : m <= 12 ? 4 : 5)], 1ull);
                atomicAdd(&P.stats[STAT_DEBUG0 + 6], (unsigned long long)m);
            }
            cnt.overflow = 0;
#endif
            // (a centre ray that gave up gets its flag from the kernel that finishes it)
            if (dof && id.sidx == 0 && slot != TRAVERSE_GAVE_UP) P.dofFlags[id.pix] = dof_flag(P, d, slot >= 0 ? t : FLT_MAX) ? 1 : 0;
            if (slot == -1) {
                const vec3 c = (P.f.environment_mapping && P.p.enabled_skybox) ? skybox_color(P.scene, d) : splat3(0.0f);
                float* out = sample_slot(P, id.pix, id.sidx);
                out[0] = c.x;
                out[1] = c.y;
                out[2] = c.z;
            }
        }
        const uint32_t work = uint32_t(id.sidx - P.sampleBegin) * nPix + uint32_t(id.pix);
        const unsigned long long hits = __ballot(slot >= 0);
        if (hits) {
            const int leader = __ffsll((long long)hits) - 1;
            unsigned int base = 0;
            if (lane == leader) base = (unsigned int)atomicAdd(&P.primaryCounters[Q_HITS], (unsigned long long)__popcll(hits));
            base = __shfl(base, leader, 64);
            if (slot >= 0) {
                HitEntry e;
                e.work = work;
                e.t = t;
                e.slot = slot;
                e.pad = 0;
                P.queue[base + lanes_below(hits)] = e;
            }
        }
        const unsigned long long gaveUp = __ballot(slot == TRAVERSE_GAVE_UP);
        if (gaveUp) {
            const int leader = __ffsll((long long)gaveUp) - 1;
            unsigned int base = 0;
            if (lane == leader) base = (unsigned int)atomicAdd(&P.primaryCounters[Q_GAVE_UP], (unsigned long long)__popcll(gaveUp));
            base = __shfl(base, leader, 64);
            if (slot == TRAVERSE_GAVE_UP) P.gaveUpQueue[base + lanes_below(gaveUp)] = work;
        }
    }
    flush_counts(cnt, P.stats);
}

// Where a lane of a one-wave block keeps its traversal stack: in LDS, or — for k_closest on the second stream, which has
// to fit on CUs whose LDS the shadow pass of the first stream has taken — in the block's slice of P.laneStacks.  (With
// its stacks in LDS the kernel waited for the shadow pass to end: 8 ms instead of 0.5, and the second lighting round
// behind it ran after everything else.)
__device__ __forceinline__ uint2* lane_stack(const FrameParams& P, uint2* lds)
{
    if (!P.laneStacks) return lds + threadIdx.x;
    const size_t depth = size_t(P.scene.stackDepth < 1 ? 1 : P.scene.stackDepth);
    return P.laneStacks + size_t(blockIdx.x) * depth * WAVE + threadIdx.x;
}

// Primary pass, third kernel: one lane per hit-queue entry the stream kernel appended without a primitive
// (slot < 0); the entry's distance and primitive slot are filled in place.
template <bool PRUNE, bool COUNT>
__global__ void __launch_bounds__(WAVE) k_closest(const FrameParams P)
{
    extern __shared__ uint2 lds_stack[];
    // These few waves hold the longest rays of the frame and run beside the shadow pass, whose waves outnumber them twenty
    // to one on every SIMD: at the highest issue priority a ray's chain of node visits is not stretched by the wait for an
    // issue slot (slowest share of 8 ranks 1.82 -> 1.71 ms, of 4 ranks 3.14 -> 3.00; the whole frame is unchanged).
    __builtin_amdgcn_s_setprio(3);
    Stack st { lane_stack(P, lds_stack), uint32_t(WAVE) };
    Counts cnt;
    const unsigned int nHits = (unsigned int)P.primaryCounters[Q_HITS];
    const uint32_t nPix = uint32_t(size_t(P.rowsLocal) * size_t(P.width));
    for (unsigned int q = blockIdx.x * WAVE + threadIdx.x; q < nHits; q += gridDim.x * WAVE) {
        HitEntry e = P.queue[q];
        if (e.slot >= 0) continue;
        Rng rng;
        RayD r;
        size_t pix;
        int sidx;
        hit_ray(P, q, e, nPix, rng, r, pix, sidx);
        float t = FLT_MAX;
        e.slot = intersect<PRUNE, false, COUNT>(P, r, t, st, cnt);
        e.t = t;
        P.queue[q] = e;
        if (P.f.depth_of_field && sidx == 0) P.dofFlags[pix] = dof_flag(P, r.d, e.slot >= 0 ? t : FLT_MAX) ? 1 : 0;
        if (e.slot < 0) {
            // cannot happen (the traversal meets the primitive the stream kernel found, or an earlier one, before
            // anything is pruned); the later passes skip such an entry
            const vec3 c = (P.f.environment_mapping && P.p.enabled_skybox) ? skybox_color(P.scene, r.d) : splat3(0.0f);
            float* out = sample_slot(P, pix, sidx);
            out[0] = c.x;
            out[1] = c.y;
            out[2] = c.z;
        }
    }
    flush_counts(cnt, P.stats);
}

// Hands a lighting round (cull / shadow / shade) its part of the hit queue: from where `beginFrom` says (null: the
// start) to the queue's current length.
__global__ void k_range(unsigned long long* round, const unsigned long long* beginFrom, const unsigned long long* length)
{
    round[Q_RANGE_BEGIN] = beginFrom ? *beginFrom : 0ull;
    round[Q_RANGE_END] = *length;
}

// Cull pass: one lane per queued hit.  Decides, for every shadow-casting light sample of the hit, whether the
// visibility bit can reach the pixel at all.  With shading on, the reference multiplies the bit with a Phong
// term (src/light.cpp:141-143,175-178,209-212) that is exactly zero when dot(N, L) <= 0 (src/shading.cpp:14-38):
// diffuse and specular both vanish, so `shade * lighted` adds zero whatever the shadow ray would have found.
// Those samples get no ray (and their visibility byte stays whatever it was: any finite value times the zero term is
// the same zero); the others are appended to the shadow work queue, the items
// of a hit next to each other and hits in queue order (one atomic per wave).
template <bool TABLE>
__global__ void __launch_bounds__(BLOCK) k_cull(const FrameParams P)
{
    const unsigned int qBegin = (unsigned int)P.queueCounters[Q_RANGE_BEGIN], nHits = (unsigned int)P.queueCounters[Q_RANGE_END];
    const uint32_t nPix = uint32_t(size_t(P.rowsLocal) * size_t(P.width));
    const uint32_t perHit = uint32_t(P.shadowSamplesPerHit);
    const int lane = int(threadIdx.x) & 63;
    const unsigned int rounded = qBegin + ((nHits - qBegin + BLOCK - 1u) & ~(BLOCK - 1u)); // whole blocks stay in the loop together
    // The waves of a block reserve their room in the work queue with ONE atomic per block and iteration: atomics on one
    // address are served one after the other (a few nanoseconds each), and one per wave (390,000 for the headline
    // frame) was most of this kernel's time.
    __shared__ uint32_t waveTotals[2][BLOCK / 64], waveShade[2][BLOCK / 64];
    __shared__ unsigned long long blockBase[2], blockShadeBase[2];
    uint32_t iteration = 0;
    uint32_t culled = 0;
    bool everyLightShadowed = true;
    for (int li = 0; li < P.nLights; li++)
        if (light_enabled(P, light_in<TABLE>(P, li).type) && light_in<TABLE>(P, li).shadowBase < 0) everyLightShadowed = false;
    for (unsigned int q = qBegin + blockIdx.x * BLOCK + threadIdx.x; q < rounded; q += gridDim.x * BLOCK) {
        bool live = q < nHits;
        Rng rng;
        RayD r = make_ray(splat3(0.0f), splat3(1.0f));
        Surface s;
        s.normal = splat3(0.0f);
        float t = 0.0f;
        if (live) {
            const HitEntry e = P.queue[q];
            live = e.slot >= 0;
            if (live) {
                size_t pix;
                int sidx;
                hit_ray(P, q, e, nPix, rng, r, pix, sidx);
                s = surface_at(P, r, e.t, e.slot);
                t = e.t;
            }
        }
        const vec3 point = r.o + (r.d * t);
        const vec3 N = normalize(s.normal);
        // Most hits face a light with all of its samples or with none: decided from the light's corners.  Every sample
        // position lies within their hull (the sample parameters stay in [0, 1)), and N . (position - point) is linear
        // in the position, so its extremes over the samples are attained at corners.  Where the smallest corner value
        // exceeds a margin far above the rounding of the per-sample expression (relative 1e-4 against about 1e-6),
        // every sample's own test below would come out positive, where the largest is below minus the margin, none
        // would: those hits skip the per-sample evaluation (random draws, square root, divide).  Two bits per light: 1 =
        // all samples wanted, 2 = none, 0 = decide sample by sample.
        uint32_t lightClass = 0;
        if (live && P.f.shading)
            for (int li = 0; li < P.nLights; li++) {
                if (li >= 16) break; // (two bits per light in one word; further lights are decided sample by sample)
                const LightDev& L = light_in<TABLE>(P, li);
                if (L.shadowBase < 0 || L.type == RT_LIGHT_POINT) continue;
                vec3 corner[4];
                int corners = 2;
                corner[0] = load3(L.v);
                if (L.type == RT_LIGHT_SEGMENT) {
                    corner[1] = load3(L.v + 3);
                } else {
                    corner[1] = corner[0] + load3(L.v + 3);
                    corner[2] = corner[0] + load3(L.v + 6);
                    corner[3] = corner[1] + load3(L.v + 6);
                    corners = 4;
                }
                float lowest = FLT_MAX, highest = -FLT_MAX, farthest = 0.0f;
                for (int c = 0; c < corners; c++) {
                    const vec3 toCorner = corner[c] - point;
                    const float d = dot(N, toCorner);
                    lowest = d < lowest ? d : lowest;    // (a NaN leaves both as they are: decided sample by sample)
                    highest = d > highest ? d : highest;
                    farthest = fmaxf(farthest, dot(toCorner, toCorner));
                }
                const float margin = 0.0001f * sqrtf(farthest);
                if (lowest > margin && lowest <= FLT_MAX) lightClass |= 1u << (2 * li);
                else if (highest < -margin && highest >= -FLT_MAX) lightClass |= 2u << (2 * li);
            }
        // pass 0 decides and counts this hit's shadow rays (the first 64 decisions are kept as a bit mask), pass 1
        // writes them at the offset the wave reserved
        uint32_t need = 0;
        unsigned long long wantedBits = 0;
        for (int pass = 0; pass < 2; pass++) {
            uint32_t at = 0;
            if (pass == 1) {
                uint32_t prefix = need; // inclusive scan over the wave
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t v = __shfl_up(prefix, off, 64);
                    if (lane >= off) prefix += v;
                }
                const uint32_t total = __shfl(prefix, 63, 64);
                const uint32_t wave = threadIdx.x >> 6, slot = iteration & 1u;
                // (the same step lists the hits the shade pass has to evaluate: all but those written off below)
                const bool toShade = live && !(need == 0 && P.f.shading && everyLightShadowed);
                const unsigned long long shadeVotes = __ballot(toShade);
                if (lane == 63) {
                    waveTotals[slot][wave] = total;
                    waveShade[slot][wave] = uint32_t(__popcll(shadeVotes));
                }
                __syncthreads();
                if (threadIdx.x == 0) {
                    uint32_t sum = 0, shade = 0;
                    for (uint32_t w = 0; w < BLOCK / 64; w++) {
                        sum += waveTotals[slot][w];
                        shade += waveShade[slot][w];
                    }
                    blockBase[slot] = sum ? atomicAdd(&P.queueCounters[Q_SHADOW_ITEMS], (unsigned long long)sum) : 0ull;
                    blockShadeBase[slot] = shade ? atomicAdd(&P.queueCounters[Q_SHADE_ITEMS], (unsigned long long)shade) : 0ull;
                }
                __syncthreads();
                uint32_t before = 0, shadeBefore = 0;
                for (uint32_t w = 0; w < wave; w++) {
                    before += waveTotals[slot][w];
                    shadeBefore += waveShade[slot][w];
                }
                at = uint32_t(blockBase[slot]) + before + prefix - need;
                if (toShade && P.shadeList) {
                    const uint32_t pos = uint32_t(blockShadeBase[slot]) + shadeBefore + lanes_below(shadeVotes);
                    P.shadeList[P.shadowQueueReversed ? uint32_t(P.queueCapacity) - 1u - pos : pos] = q;
                }
                iteration++;
            }
            if (live) {
                for (int li = 0; li < P.nLights; li++) {
                    const LightDev& L = light_in<TABLE>(P, li);
                    if (L.shadowBase < 0) continue;
                    const int total = L.nA * L.nB;
                    const uint32_t per = light_draws_per_sample(L.type);
                    for (int k = 0; k < total; k++) {
                        const int j = L.shadowBase + k;
                        bool wanted = true;
                        const uint32_t decided = li < 16 ? (lightClass >> (2 * li)) & 3u : 0u;
                        if (pass == 1 && j < 64) {
                            wanted = (wantedBits >> j) & 1ull;
                        } else if (decided != 0u) {
                            wanted = decided == 1u;
                        } else if (P.f.shading) {
                            vec3 position, color;
                            light_sample(L, k, rng.pixelKey, rng.sample, rng.randCounter + uint32_t(L.drawBase) + uint32_t(k) * per, position, color);
                            wanted = dot(N, normalize(position - point)) > 0; // same expression as phong()
                        }
                        const uint32_t item = q * perHit + uint32_t(j);
                        if (pass == 0) {
                            need += wanted ? 1u : 0u;
                            if (wanted && j < 64) wantedBits |= 1ull << j;
                            if (!wanted) culled++; // its visibility byte is never written: k_shade multiplies it with exact zeros
                        } else if (wanted) {
                            P.shadowQueue[P.shadowQueueReversed ? P.shadowQueueItems - 1u - at : at] = item;
                            at++;
                        }
                    }
                }
            }
        }
        // A hit none of whose light samples is wanted, in a frame where every enabled light casts shadows, sums exact
        // zeros (+0 + -0 = +0): its colour is written here and k_shade skips it.
        if (live && need == 0 && P.f.shading && everyLightShadowed) {
            const HitEntry e = P.queue[q];
            float* out = hit_colour_slot(P, q, size_t(e.work % nPix), P.sampleBegin + int(e.work / nPix));
            out[0] = 0.0f;
            out[1] = 0.0f;
            out[2] = 0.0f;
            P.queue[q].pad = 1;
        }
    }
    if (P.stats) {
        const uint32_t c = wave_sum(culled);
        if (lane == 0 && c) atomicAdd(&P.stats[STAT_CULLED], (unsigned long long)c);
    }
}

// Any-hit queries traced by a wave as a STREAM OF (ray, node) PAIRS.  Two uses:
//   shadow pass (PRIMARY = false): one query per queued (hit, light sample) — testVisibilityLightSample,
//     reference src/light.cpp:68-89; the answer is the visibility bit;
//   primary pass (PRIMARY = true): "does this camera ray meet any primitive at all", for the rays that gave up
//     in k_primary (see there).  Rays that do not are final; the others go to k_closest.
//
// The answer of an any-hit query ("is any primitive accepted before the light") is the OR over the nodes the
// ray enters, and entering a node depends only on the ray's own box tests on the way down.  So the pairs
// (ray, node it has entered) of a packet of rays are independent pieces of work: any lane can take any pair,
// test the node's two child boxes against that ray and emit up to two new pairs, in any order, and every ray
// still performs exactly the box and primitive tests the reference's traversal makes for it before its first
// accepted primitive.  The wave keeps the pairs of its packet on ONE stack in LDS and every step pops the top
// 64: all lanes work as long as 64 pairs exist, however differently long the individual rays' traversals
// are, and control flow is wave-uniform.  Pairs whose node is a leaf go to a second pool; a step that tests
// primitives runs when 64 of those have gathered (or nothing else is left), so box tests and primitive tests
// never share a step.  A lane works for different rays every step; what it needs of them comes from a ray table in
// LDS.  A ray that has been blocked gets a flag there, and its remaining pairs are dropped when popped.
//
// Ray groups (G = 1 or 2).  The queries of a packet are consecutive work items, i.e. mostly light samples of the
// same hit: rays with one origin and nearly one direction, which enter mostly the same nodes.  A stack entry is
// therefore (group of G consecutive rays, node, which rays of the group have entered it): ONE popped entry, one
// staged node record, and G box tests of that record, each ray deciding for itself exactly as it would alone (a ray
// that has not entered the node cannot enter a child).  On the headline frame: 35.1 M box steps with G = 1, 24.8 M with
// G = 2 (1.46 of 2 rays share an entry on average).  G = 2 is the faster by about 7 %; groups of four (18.8 M steps, 2.0
// of 4 rays per entry) were slower than either and are no longer built.
//
// Stack bound.  The top k <= 64 entries are popped, the topmost by lane k-1, the one below it by lane k-2, and so
// on; the children are pushed in lane order (those of lane 0 lowest).  By induction the tree levels of the
// entries never decrease from the bottom of the stack to the top: the popped entries are the deepest ones, their
// levels do not decrease with the lane index, and every child is deeper than everything left below.  Hence a step pops
// level-l entries only together with ALL deeper entries, the level l+1 then consists of the new children
// alone, and no level ever holds more than the 128 children of one step: 128 entries per inner level suffice.
//
// Persistent waves: a wave whose packet has ended takes the next one from a global cursor and stops once
// the cursor has passed the packet count.
constexpr int STREAM_RAYS = 64;                           // rays per packet
__device__ __forceinline__ bool lane_in(unsigned long long mask);
__host__ __device__ constexpr uint32_t stream_ref_limit(int group) { return group == 1 ? (1u << 25) - 1u : (1u << 24) - 1u; }

// Stack / leaf-pool entries of the single-ray (G = 1) and the two-ray (G = 2) stream.  `ref` is a child reference as the
// node record holds it: >= 0 an inner node's index, < 0 = ~(primitive slot of a leaf).
//   G = 1:  (ref << 6) | ray
//   G = 2:  code << 30 | (ref & 0xffffff) << 6 | group, group = ray / 2, code = which rays of the group have entered the
//           node: 1 = the first only, 2 = both, 3 = the second only (so each "has entered" is ONE unsigned comparison)
template <int G> struct PairEntry {
    static __device__ __forceinline__ uint32_t root(int32_t ref, uint32_t group, uint32_t raysOfGroup /* bit per ray */)
    {
        if (G == 1) return (uint32_t(ref) << 6) | group;
        const uint32_t code = raysOfGroup == 3u ? 2u : (raysOfGroup == 1u ? 1u : 3u);
        return (code << 30) | ((uint32_t(ref) & 0xffffffu) << 6) | group;
    }
    // byte offset of the group's first ray in a table of 16-byte rows
    static __device__ __forceinline__ uint32_t ray_bytes(uint32_t e) { return G == 1 ? (e << 4) & 0x3f0u : (e << 5) & 0x3e0u; }
    static __device__ __forceinline__ uint32_t first_ray(uint32_t e) { return G == 1 ? e & 63u : (e & 31u) << 1; }
    // byte offset of the 64-byte record of an inner node entry
    static __device__ __forceinline__ uint32_t record_bytes(uint32_t e) { return G == 1 ? e & ~63u : e & 0x3fffffc0u; }
    // primitive slot of a leaf entry
    static __device__ __forceinline__ uint32_t slot(uint32_t e) { return G == 1 ? uint32_t(~(int32_t(e) >> 6)) : ~(e >> 6) & 0xffffffu; }
    // lanes whose entry says that ray g of the group has entered the node
    static __device__ __forceinline__ unsigned long long entered(uint32_t e, int g)
    {
        if (G == 1) return ~0ull;
        return g == 0 ? __builtin_amdgcn_uicmp(e, 0xc0000000u, 36 /* < */) : __builtin_amdgcn_uicmp(e, 0x80000000u, 35 /* >= */);
    }
    static __device__ __forceinline__ uint32_t entered_bits(uint32_t e) { return G == 1 ? 1u : (e < 0xc0000000u ? 1u : 0u) | (e >= 0x80000000u ? 2u : 0u); }
    // entry of the child `ref` for the rays that enter it (lane masks first / second; second = first for G = 1)
    static __device__ __forceinline__ uint32_t child(uint32_t e, int32_t ref, unsigned long long first, unsigned long long second)
    {
        if (G == 1) return (uint32_t(ref) << 6) | (e & 63u);
        uint32_t code = lane_in(second) ? 0xc0000000u : 0x40000000u;
        code = lane_in(first & second) ? 0x80000000u : code;
        // (code & top) | (rest & ~top) as one bit-field insert; the compiler would otherwise spend an AND and an OR on it
        const uint32_t rest = (uint32_t(ref) << 6) | (e & 31u);
        uint32_t entry;
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(entry) : "s"(0xc0000000u), "v"(code), "v"(rest));
        return entry;
    }
};
constexpr int STREAM_PER_LEVEL = 128;                     // stack entries per inner tree level
constexpr int STREAM_LEAF_CAP = 192;                      // < 64 waiting + 128 from one step
#ifndef RT_XCD_PACKETS
#define RT_XCD_PACKETS 1
#endif
#ifndef RT_LEAF_TRIGGER
#define RT_LEAF_TRIGGER 64
#endif
constexpr uint32_t STREAM_LEAF_TRIGGER = RT_LEAF_TRIGGER; // a primitive step runs once this many leaf entries wait (<= 64)

constexpr uint32_t RAY_BLOCKED = 8u;                      // flag bit next to the three zero-direction bits

constexpr int STREAM_STAGE_STRIDE = 1024 + 16;            // bytes between the four quarter-wave images of the record stage

__host__ __device__ constexpr int stream_stack_entries(int levels) { return STREAM_PER_LEVEL * (levels < 2 ? 1 : levels - 1); }
__host__ __device__ constexpr size_t stream_lds_bytes(int stackEntriesInLds)
{
    // record stage, ray table (origin + flags, reciprocal direction: 2 x 16 bytes per ray), leaf pool, pair stack
    return 4 * STREAM_STAGE_STRIDE + size_t(STREAM_RAYS) * 32 + size_t(STREAM_LEAF_CAP) * 4 + size_t(stackEntriesInLds) * 4;
}
constexpr uint32_t STREAM_MAX_WAVES = 256u * 32u; // every wave slot of the chip

// The pair stack of a wave: the first `ldsEntries` entries in LDS, the rest of the proven bound in a spill
// area of the wave's own in device memory.  The spill is written and read by different lanes of the wave,
// so it bypasses the per-CU cache (device-scope atomics); a wave rarely gets that deep.
struct PairStack {
    uint32_t* lds;
    uint32_t* spill;
    uint32_t ldsEntries;
    __device__ __forceinline__ uint32_t get(uint32_t i) const
    {
        if (i < ldsEntries) return lds[i];
        return __hip_atomic_load(spill + (i - ldsEntries), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ void put(uint32_t i, uint32_t v) const
    {
        if (i < ldsEntries) lds[i] = v;
        else __hip_atomic_store(spill + (i - ldsEntries), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

// One 64-byte record per lane (byte offset `off` from `base`, any valid offset for lanes that need nothing),
// fetched so that a load instruction touches 16 cache lines instead of 64: the four lanes of a quad fetch the
// four 16-byte quarters of ONE record per load (the record of quad lane 0, then 1, 2, 3).  The data go to the
// LDS stage directly (destination = uniform base + lane * 16, which puts the quarters of a record next to each
// other), from where every lane reads its own record back once stage_wait() has passed.  The images of the
// four loads are 16 bytes apart modulo the bank width so that those reads do not collide.
typedef __attribute__((address_space(3))) char* lds_bytes;
__device__ __forceinline__ void stage_records(const void* base, uint32_t off, lds_bytes stage, uint32_t quarter)
{
    const char* g = static_cast<const char*>(base);
    __builtin_amdgcn_global_load_lds(g + size_t(uint32_t(__builtin_amdgcn_update_dpp(0, int(off), 0x00, 0xf, 0xf, true)) + quarter), stage, 16, 0, 0);
    __builtin_amdgcn_global_load_lds(g + size_t(uint32_t(__builtin_amdgcn_update_dpp(0, int(off), 0x55, 0xf, 0xf, true)) + quarter), stage + STREAM_STAGE_STRIDE, 16, 0, 0);
    __builtin_amdgcn_global_load_lds(g + size_t(uint32_t(__builtin_amdgcn_update_dpp(0, int(off), 0xaa, 0xf, 0xf, true)) + quarter), stage + 2 * STREAM_STAGE_STRIDE, 16, 0, 0);
    __builtin_amdgcn_global_load_lds(g + size_t(uint32_t(__builtin_amdgcn_update_dpp(0, int(off), 0xff, 0xf, 0xf, true)) + quarter), stage + 3 * STREAM_STAGE_STRIDE, 16, 0, 0);
}
__device__ __forceinline__ void stage_wait()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// LDS operations of a wave execute in order; this only keeps the compiler from moving them across (one wave per block)
__device__ __forceinline__ void lds_fence()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}


// "entered at all" of both child boxes of a node — box_entry() followed by the caller's `t < FLT_MAX` — for a ray of
// the FAST CLASS (RayD::fast) in a bounded scene (SceneView::bounded).  As box_pair_fast() explains, the parameter the
// reference reports is then always below FLT_MAX and never a NaN: its last comparison is always true and the choice
// between t_in and t_out that feeds it is not needed.  What is left of reference src/intersect.cpp:128-139 is the three
// rejections.
// a, b, c are the twelve consecutive floats of a node record (two boxes, slab by slab): each slab is one packed
// subtract and one packed multiply with the ray's origin component and reciprocal broadcast to both halves.
// A ray that has been answered is poisoned in the ray table (origin x = +inf): its x slab products are both -inf or
// both +inf, which fails `t_out <= 0` or `t_in > t_out`, so it enters nothing without any flag test here.
// The answers are lane masks (one bit per lane, in scalar registers): the three rejections of each box are three vector
// compares whose results are combined by scalar instructions, and the masks feed the prefix counts, the population
// counts and the execution masks of the pushes directly, without a round trip through vector registers.
// (compare codes of __builtin_amdgcn_fcmpf / sicmp: 10 = unordered or greater, 13 = unordered or less-or-equal,
// 39 = signed greater-or-equal)
__device__ __forceinline__ void boxes_entered_masks(f32x4 a, f32x4 b, f32x4 c, f32x4 o, f32x4 i, unsigned long long& left, unsigned long long& right)
{
    const f32x2 oX { o.x, o.x }, oY { o.y, o.y }, oZ { o.z, o.z }, iX { i.x, i.x }, iY { i.y, i.y }, iZ { i.z, i.z };
    const f32x2 p0 = (a.xy - oX) * iX; // left  x slab
    const f32x2 p1 = (a.zw - oY) * iY; // left  y slab
    const f32x2 p2 = (b.xy - oZ) * iZ; // left  z slab
    const f32x2 p3 = (b.zw - oX) * iX; // right x slab
    const f32x2 p4 = (c.xy - oY) * iY; // right y slab
    const f32x2 p5 = (c.zw - oZ) * iZ; // right z slab
    const float tinL = fmaxf(fmaxf(fminf(p0.x, p0.y), fminf(p1.x, p1.y)), fminf(p2.x, p2.y));
    const float toutL = fminf(fminf(fmaxf(p0.x, p0.y), fmaxf(p1.x, p1.y)), fmaxf(p2.x, p2.y));
    const float tinR = fmaxf(fmaxf(fminf(p3.x, p3.y), fminf(p4.x, p4.y)), fminf(p5.x, p5.y));
    const float toutR = fminf(fminf(fmaxf(p3.x, p3.y), fmaxf(p4.x, p4.y)), fmaxf(p5.x, p5.y));
    // !(t_in > t_out) && !(t_out <= 0) && !(|t_in| <= 1e-6)
    left = __builtin_amdgcn_fcmpf(tinL, toutL, 13) & __builtin_amdgcn_fcmpf(toutL, 0.0f, 10) & __builtin_amdgcn_fcmpf(fabsf(tinL), 0.000001f, 10);
    right = __builtin_amdgcn_fcmpf(tinR, toutR, 13) & __builtin_amdgcn_fcmpf(toutR, 0.0f, 10) & __builtin_amdgcn_fcmpf(fabsf(tinR), 0.000001f, 10);
}
// number of set bits of `mask` below this lane, plus `base`
__device__ __forceinline__ uint32_t mask_prefix(unsigned long long mask, uint32_t base)
{
    return __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), base));
}
// "this lane's bit of `mask` is set", as the execution mask of the branch (no vector instruction)
__device__ __forceinline__ bool lane_in(unsigned long long mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }

// value of `v` in lane `src` (any lane may ask for any lane's value; all lanes must be active)
__device__ __forceinline__ float lane_value(uint32_t src, float v)
{
    return __int_as_float(__builtin_amdgcn_ds_bpermute(int(src << 2), __float_as_int(v)));
}

// Timing build (-DRT_STREAM_TIMING): the shadow stream adds up, per wave, the clock ticks between marks inside its steps
// and leaves the sums in the debug slots of the statistics block.
#ifdef RT_STREAM_TIMING
#define STREAM_MARK(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tick[i] += now_ - mark; mark = now_; } while (0)
#else
#define STREAM_MARK(i) do { } while (0)
#endif
template <bool COUNT, bool PRIMARY, int G>
__global__ void __launch_bounds__(WAVE, 5) k_stream(const FrameParams P)
{
#ifdef RT_STREAM_TIMING
    unsigned long long tick[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, mark = __builtin_amdgcn_s_memtime();
#endif
    // Ray i of the packet is set up by lane i.  What a box step needs of it (origin, reciprocal direction, flags) goes
    // to a table in LDS; the direction, which only the rare primitive steps need, stays in lane i's registers and is
    // fetched from there with lane permutes.
    using E = PairEntry<G>;
    extern __shared__ f32x4 streamLds[];
    char* stage = reinterpret_cast<char*>(streamLds);
    // Ray table: rayO[i] = (origin, flag word), rayI[i] = (reciprocal direction, unused).  A lane that works on a pair
    // of ray i reads both with one 16-byte LDS read each (lane permutes of the six values cost several times as much
    // of the LDS pipeline, which is what this kernel runs out of).  The flag word holds the zero-direction classes and
    // "blocked"; other lanes set the latter with an LDS atomic.
    f32x4* rayO = reinterpret_cast<f32x4*>(stage + 4 * STREAM_STAGE_STRIDE);
    f32x4* rayI = rayO + STREAM_RAYS;
    uint32_t* leafPool = reinterpret_cast<uint32_t*>(rayI + STREAM_RAYS);
    auto flag_word = [&](uint32_t i) { return reinterpret_cast<uint32_t*>(rayO + i) + 3; };
    const PairStack pool { leafPool + STREAM_LEAF_CAP, P.pairSpill + size_t(blockIdx.x) * P.pairSpillEntries, uint32_t(P.pairStackLds) };
    const int lane = int(threadIdx.x);
    Counts cnt;
    uint32_t nPacketsDone = 0, nBoxSteps = 0, nPrimSteps = 0, nEntries = 0; // (test-counting build only; wave-uniform)
    const SceneView& S = P.scene;
    const uint32_t perHit = uint32_t(P.shadowSamplesPerHit);
    unsigned long long* const counters = PRIMARY ? P.primaryCounters : P.queueCounters;
    const unsigned long long queued = counters[PRIMARY ? Q_GAVE_UP : Q_SHADOW_ITEMS];
    // camera rays that gave up are long by construction: a quarter of the slots at most gives the stream enough pairs
    // and spreads the (often few) rays over more waves ...
    uint32_t raysPerPacket = STREAM_RAYS;
    if (PRIMARY) {
        // ... down to one ray per packet when there are fewer rays than waves (a single such ray can have thousands of
        // pairs; a wave that holds sixteen of them is the last one running)
        const unsigned long long perWave = queued / (2ull * gridDim.x);
        raysPerPacket = perWave < 1ull ? 1u : (perWave > STREAM_RAYS / 4 ? STREAM_RAYS / 4 : uint32_t(perWave));
    }
    const unsigned long long nPackets = (queued + raysPerPacket - 1) / raysPerPacket;
    const uint32_t nPix = uint32_t(size_t(P.rowsLocal) * size_t(P.width));
    const float tBound = PRIMARY ? FLT_MAX : 1.0f; // a shadow ray ends at the light sample (t = 1)
    const bool sceneBounded = S.bounded != 0;
    const uint32_t lastNode = S.numNodes > 0 ? uint32_t(S.numNodes - 1) : 0u, lastPrim = S.numPrims > 0 ? uint32_t(S.numPrims - 1) : 0u;
    // (byte offset of the last node record, wave-uniform: what a staged fetch is clamped to)
    const uint32_t lastNodeBytes = uint32_t(__builtin_amdgcn_readfirstlane(int(lastNode * uint32_t(sizeof(NodeRec)))));
    const lds_bytes stageLds = (lds_bytes)stage;
    const f32x4* staged = reinterpret_cast<const f32x4*>(stage + (lane & 3) * STREAM_STAGE_STRIDE + (lane >> 2) * 64);
    const uint32_t quarter = uint32_t(lane & 3) * 16u;
    // Packets of 64 consecutive queued queries come from a cursor.  The shadow pass has eight of them, one per eighth of
    // the queue, and a wave starts with the eighth of its own XCD: the queue is in hit order, i.e. in image order, so an
    // XCD's waves work on one region of the scene and its L2 cache holds that region's part of the hierarchy instead of
    // a slice of everything.  A wave whose eighth has run out goes on with the next one, so nobody idles.
    const unsigned long long perEighth = (nPackets + 7ull) / 8ull;
    // (hardware register XCC_ID, bits 3:0: the XCD this wave runs on)
    const uint32_t home = (PRIMARY || !RT_XCD_PACKETS) ? 0u : uint32_t(__builtin_amdgcn_s_getreg((3 << 11) | 20)) & 7u;
    uint32_t eighthsDone = 0;
    while (true) {
        unsigned long long pk = nPackets;
        if (PRIMARY || !RT_XCD_PACKETS) {
            unsigned long long grabbed = 0;
            if (lane == 0) grabbed = atomicAdd(&counters[PRIMARY ? Q_GAVE_UP_PACKETS : Q_SHADOW_PACKETS], 1ull);
            // (the same in every lane: kept in scalar registers)
            pk = (unsigned long long)uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(grabbed))))
                | ((unsigned long long)uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(grabbed >> 32)))) << 32);
        } else {
            while (eighthsDone < 8u) {
                const uint32_t x = (home + eighthsDone) & 7u;
                unsigned long long grabbed = 0;
                if (lane == 0) grabbed = atomicAdd(&counters[Q_SHADOW_XCD + 16 * x], 1ull);
                const unsigned long long g = (unsigned long long)uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(grabbed))))
                    | ((unsigned long long)uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(grabbed >> 32)))) << 32);
                if (g < perEighth && x * perEighth + g < nPackets) {
                    pk = x * perEighth + g;
                    break;
                }
                eighthsDone++;
            }
        }
        if (pk >= nPackets) break;
        if (COUNT) nPacketsDone++;
        // ---- this lane's query; the ones that pass the root gate start as pairs (ray, root)
        const unsigned long long qi = pk * raysPerPacket + uint32_t(lane);
        const bool have = uint32_t(lane) < raysPerPacket && qi < queued;
        uint32_t work = 0;
        RayD r = make_ray(splat3(0.0f), splat3(1.0f));
        bool traced = false;
        if (have) {
            if (PRIMARY) {
                // a camera ray that gave up in k_primary
                work = P.gaveUpQueue[qi];
                HitEntry e;
                e.work = work;
                Rng rng;
                size_t pix;
                int sidx;
                hit_ray(P, 0u, e, nPix, rng, r, pix, sidx); // (a camera ray: never a wavefront node)
            } else {
                work = P.shadowQueue[P.shadowQueueReversed ? (unsigned long long)P.shadowQueueItems - 1ull - qi : qi];
                const HitEntry e = P.queue[work / perHit];
                const int j = int(work % perHit);
                Rng rng;
                RayD cam;
                size_t pix;
                int sidx;
                hit_ray(P, work / perHit, e, nPix, rng, cam, pix, sidx);
                // which light and which of its samples: the last light whose first shadow sample is <= j
                vec3 position, color;
                auto sample_of = [&](auto home) {
                    constexpr bool TABLE = decltype(home)::value;
                    int li = 0;
                    for (int i = 0; i < P.nLights; i++)
                        if (light_in<TABLE>(P, i).shadowBase >= 0 && light_in<TABLE>(P, i).shadowBase <= j) li = i;
                    const LightDev& L = light_in<TABLE>(P, li);
                    const int k = j - L.shadowBase;
                    light_sample(L, k, rng.pixelKey, rng.sample, rng.randCounter + uint32_t(L.drawBase) + uint32_t(k) * light_draws_per_sample(L.type), position, color);
                };
                if (P.lightTable) sample_of(std::true_type()); else sample_of(std::false_type());
                const float epsilon = 0.00001f;
                const vec3 origin = cam.o + (1 - epsilon) * e.t * cam.d;
                r = make_ray(origin, position - origin);
                cnt.shadow++;
            }
            float tRoot;
            if (COUNT) cnt.box++;
            // root gate against the query's own bound (…hierarchy.cpp:296-301)
            traced = S.numPrims > 0 && box_entry(r, S.rootLo, S.rootHi, tRoot) && tRoot < tBound;
        }
        const uint32_t zeroBits = (r.zx ? 1u : 0u) | (r.zy ? 2u : 0u) | (r.zz ? 4u : 0u);
        rayO[lane] = f32x4 { r.o.x, r.o.y, r.o.z, __uint_as_float(zeroBits) };
        rayI[lane] = f32x4 { r.inv.x, r.inv.y, r.inv.z, 0.0f };
        const unsigned long long m = __builtin_amdgcn_ballot_w64(traced);
        // The packet takes the general box test when some ray of it is outside the fast class of boxes_entered_masks(): a
        // direction component in the zero class or not finite, or an origin component beyond the bound (or NaN).
        // (passed through a scalar register so that the branches on it stay scalar branches)
        const bool zeroClass = __builtin_amdgcn_readfirstlane(int(!sceneBounded || __builtin_amdgcn_ballot_w64(traced && !r.fast) != 0)) != 0;
        // one entry per group of G consecutive rays that has a ray past the root gate: (group, those rays, root)
        uint32_t count = 0, leafCount = 0;
        {
            constexpr int LOG_G = G == 1 ? 0 : 1;
            const uint32_t myGroup = uint32_t(lane) >> LOG_G;
            const uint32_t groupMask = uint32_t(m >> (myGroup << LOG_G)) & ((1u << G) - 1u);
            const bool leader = (uint32_t(lane) & (uint32_t(G) - 1u)) == 0u && groupMask != 0u;
            const unsigned long long lm = __builtin_amdgcn_ballot_w64(leader);
            if (leader) {
                const uint32_t at = lanes_below(lm);
                const uint32_t entry = E::root(S.rootRef, myGroup, groupMask);
                if (S.rootRef >= 0) pool.put(at, entry);
                else leafPool[at] = entry;
            }
            if (S.rootRef >= 0) count = uint32_t(__popcll(lm));
            else leafCount = uint32_t(__popcll(lm));
        }
        lds_fence();
        STREAM_MARK(0); // packet set-up
        // ---- the stream
        while (count | leafCount) {
#ifdef RT_STREAM_TIMING
            const bool wasBoxStep = count && leafCount < STREAM_LEAF_TRIGGER;
#endif
            if (count && leafCount < STREAM_LEAF_TRIGGER) {
                // Box step of the single-ray and the two-ray stream, written for the fewest vector instructions per step:
                // votes, prefix counts and push predicates live in scalar registers as lane masks; the entry formats
                // (PairEntry) make the record's byte offset one AND and a child's entry two to four operations, the same
                // for inner and leaf children.
                const uint32_t k = count < 64u ? count : 64u;
                if (COUNT) { nBoxSteps++; nEntries += k; }
                const unsigned long long got = k >= 64u ? ~0ull : (1ull << k) - 1ull;
                count -= k;
                const bool inLds = count + 128u <= pool.ldsEntries; // this step's pops and pushes stay clear of the spill
                // (lanes without an entry read whatever lies above the top, possibly never written: the record offset is
                // clamped below and their votes are masked out)
                uint32_t e;
                if (inLds) e = pool.lds[count + uint32_t(lane)];
                else e = lane_in(got) ? pool.get(count + uint32_t(lane)) : 0u;
                // the rays' values are read before the node loads are issued so that both are in flight together
                const char* const rayAt = reinterpret_cast<const char*>(rayO) + E::ray_bytes(e);
                f32x4 ro[G], ri[G];
#pragma unroll
                for (int g = 0; g < G; g++) {
                    ro[g] = *reinterpret_cast<const f32x4*>(rayAt + g * 16);
                    ri[g] = *reinterpret_cast<const f32x4*>(rayAt + STREAM_RAYS * 16 + g * 16);
                }
                asm volatile("" ::: "memory");
                stage_records(S.nodes, min(E::record_bytes(e), lastNodeBytes), stageLds, quarter);
                STREAM_MARK(1); // pop, ray reads and fetch issued
                // the ray values are named as operands of the wait so that their LDS reads are issued ahead of it
                if (G == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ro[0]), "+v"(ri[0])::"memory");
                else asm volatile("s_waitcnt vmcnt(0)" : "+v"(ro[0]), "+v"(ri[0]), "+v"(ro[G > 1 ? 1 : 0]), "+v"(ri[G > 1 ? 1 : 0])::"memory");
                STREAM_MARK(2); // records arrived
                const f32x4 a = staged[0], b = staged[1], c = staged[2];
                const f32x2 refs = *reinterpret_cast<const f32x2*>(staged + 3);
                const int32_t refL = __float_as_int(refs.x), refR = __float_as_int(refs.y);
                // per ray of the entry: has it entered the entry's node (always, for a single-ray entry), and which of the
                // two child boxes does it enter
                unsigned long long gL[G], gR[G];
#pragma unroll
                for (int g = 0; g < G; g++) {
                    const unsigned long long in = E::entered(e, g) & got;
                    unsigned long long mL, mR;
                    if (!zeroClass) {
                        boxes_entered_masks(a, b, c, ro[g], ri[g], mL, mR);
                        if (COUNT && lane_in(in) && !(__float_as_uint(ro[g].w) & RAY_BLOCKED)) cnt.box += 2;
                    } else {
                        const uint32_t flags = __float_as_uint(ro[g].w);
                        const bool live = lane_in(in) && !(flags & RAY_BLOCKED);
                        if (COUNT && live) cnt.box += 2;
                        RayD q;
                        q.o = make3(ro[g].x, ro[g].y, ro[g].z);
                        q.inv = make3(ri[g].x, ri[g].y, ri[g].z);
                        q.zx = flags & 1u; q.zy = flags & 2u; q.zz = flags & 4u;
                        float tL, tR;
                        mL = __builtin_amdgcn_ballot_w64(live && box_entry(q, a.x, a.z, b.x, a.y, a.w, b.y, tL) && tL < FLT_MAX);
                        mR = __builtin_amdgcn_ballot_w64(live && box_entry(q, b.z, c.x, c.z, b.w, c.y, c.w, tR) && tR < FLT_MAX);
                    }
                    gL[g] = mL & in;
                    gR[g] = mR & in;
                }
                const unsigned long long anyL = G == 1 ? gL[0] : gL[0] | gL[G - 1], anyR = G == 1 ? gR[0] : gR[0] | gR[G - 1];
                const unsigned long long innerL = __builtin_amdgcn_sicmp(refL, 0, 39), innerR = __builtin_amdgcn_sicmp(refR, 0, 39);
                uint32_t entryL = E::child(e, refL, gL[0], gL[G - 1]), entryR = E::child(e, refR, gR[0], gR[G - 1]);
                // children that are inner nodes go back on the stack, those of higher lanes higher (left below right)
                const unsigned long long iL = anyL & innerL, iR = anyR & innerR;
                const uint32_t at = mask_prefix(iR, mask_prefix(iL, 0u));
#ifdef RT_STREAM_TIMING
                asm volatile("" : "+v"(entryL), "+v"(entryR));
#endif
                STREAM_MARK(3); // stage read back, box tests, entries
                if (inLds) {
                    uint32_t* const top = pool.lds + count;
                    if (lane_in(iL)) top[at] = entryL;
                    if (lane_in(iR & iL)) top[at + 1u] = entryR;
                    if (lane_in(iR & ~iL)) top[at] = entryR;
                } else {
                    if (lane_in(iL)) pool.put(count + at, entryL);
                    if (lane_in(iR)) pool.put(count + at + (lane_in(iL) ? 1u : 0u), entryR);
                }
                count += uint32_t(__popcll(iL)) + uint32_t(__popcll(iR));
                // children that are leaves: to the primitive pool (any order), the left ones first
                const unsigned long long lL = anyL & ~innerL, lR = anyR & ~innerR;
                const uint32_t nLL = uint32_t(__popcll(lL));
                if (lane_in(lL)) (leafPool + leafCount)[mask_prefix(lL, 0u)] = entryL;
                if (lane_in(lR)) (leafPool + leafCount + nLL)[mask_prefix(lR, 0u)] = entryR;
                leafCount += nLL + uint32_t(__popcll(lR));
            } else {
                // primitive step: up to 64 (group, leaf) entries
                const uint32_t k = leafCount < 64u ? leafCount : 64u;
                if (COUNT) nPrimSteps++;
                const bool got = uint32_t(lane) < k;
                const uint32_t e = got ? leafPool[leafCount - 1u - uint32_t(lane)] : 0u;
                leafCount -= k;
                // (single-ray entries hold the leaf's reference as the node record does: ~slot)
                const uint32_t first = E::first_ray(e);
                const uint32_t mask = got ? E::entered_bits(e) : 0u;
                const uint32_t firstSlot = E::slot(e);
                asm volatile("" ::: "memory");
                stage_records(S.prims, min(firstSlot, lastPrim) * uint32_t(sizeof(PrimRec)), stageLds, quarter);
                stage_wait();
                // The rays of the group one after the other, as a loop that is NOT unrolled: one copy of the triangle test
                // in the code, and only one ray's values in registers at a time.
#pragma nounroll
                for (int g = 0; g < G; g++) {
                    const uint32_t ray = first + uint32_t(g);
                    const f32x4 o = rayO[ray];
                    RayD q;
                    q.o = make3(o.x, o.y, o.z);
                    q.d = make3(lane_value(ray, r.d.x), lane_value(ray, r.d.y), lane_value(ray, r.d.z));
                    bool live = ((mask >> g) & 1u) && !(__float_as_uint(o.w) & RAY_BLOCKED);
                    if (live) {
                        int32_t slot = int32_t(firstSlot);
                        int32_t more;
                        bool firstRec = true;
                        do {
                            // the leaf's first record is staged; further ones (leaves cut off by the level limit) come one by one
                            Rec64 p;
                            if (firstRec) {
                                p.a = staged[0];
                                p.b = staged[1];
                                p.c = staged[2];
                                p.d = staged[3];
                                firstRec = false;
                            } else {
                                p = load_rec(&S.prims[slot]);
                            }
                            more = prim_more(p);
                            if (COUNT) cnt.tri++;
                            float t;
                            if (triangle_hit(q, p, tBound, t)) {
                                // answered: flagged, and poisoned for the fast box test (see boxes_entered_masks)
                                atomicOr(flag_word(ray), RAY_BLOCKED);
                                *reinterpret_cast<float*>(rayO + ray) = __builtin_inff();
                                live = false;
                            }
                            slot++;
                        } while (more > 0 && live);
                    }
                }
            }
            __syncthreads(); // one wave per block: orders this step's LDS and spill traffic before the next step's
            STREAM_MARK(wasBoxStep ? 4 : 5); // pushes + wait / the whole primitive step
        }
        // ---- answers
        const bool found = *flag_word(uint32_t(lane)) & RAY_BLOCKED;
        if (!PRIMARY) {
            if (have) P.visibility[work] = found ? 0 : 1;
        } else {
            // a camera ray that met no primitive is final (sky or black); the others queue up for k_closest (one
            // atomic per wave)
            if (have && !found) {
                if (P.f.depth_of_field && P.sampleBegin + int(work / nPix) == 0) P.dofFlags[work % nPix] = 0;
                const vec3 c = (P.f.environment_mapping && P.p.enabled_skybox) ? skybox_color(P.scene, r.d) : splat3(0.0f);
                float* out = sample_slot(P, size_t(work % nPix), P.sampleBegin + int(work / nPix));
                out[0] = c.x;
                out[1] = c.y;
                out[2] = c.z;
            }
            const unsigned long long hits = __builtin_amdgcn_ballot_w64(have && found);
            if (hits) {
                const int leader = __ffsll((long long)hits) - 1;
                unsigned int base = 0;
                if (lane == leader) base = (unsigned int)atomicAdd(&P.primaryCounters[Q_HITS], (unsigned long long)__popcll(hits));
                base = __shfl(base, leader, 64);
                if (have && found) {
                    HitEntry e;
                    e.work = work;
                    e.t = 0.0f;
                    e.slot = -1;
                    e.pad = 0;
                    P.queue[base + lanes_below(hits)] = e;
                }
            }
        }
        lds_fence();
        STREAM_MARK(6); // answers
    }
#ifdef RT_STREAM_TIMING
    if (!PRIMARY && G <= 2 && P.stats && lane == 0)
        for (int i = 0; i < 7; i++) atomicAdd(&P.stats[STAT_DEBUG0 + i], tick[i]);
#endif
    flush_counts(cnt, P.stats, !PRIMARY);
    if (COUNT && !PRIMARY && P.stats && lane == 0) {
        atomicAdd(&P.stats[STAT_STREAM_PACKETS], (unsigned long long)nPacketsDone);
        atomicAdd(&P.stats[STAT_STREAM_BOX_STEPS], (unsigned long long)nBoxSteps);
        atomicAdd(&P.stats[STAT_STREAM_PRIM_STEPS], (unsigned long long)nPrimSteps);
        atomicAdd(&P.stats[STAT_STREAM_ENTRIES], (unsigned long long)nEntries);
    }
}

// Shade pass: one lane per queued hit — direct lighting with the visibility bits of the shadow pass
// (reference src/light.cpp:124-220 for a hit whose colour is Lo, src/render.cpp:64-67).  Eight waves per SIMD (64
// registers): the pass waits on the hit, material and visibility reads of its lanes (0.33 -> 0.30 ms).
template <bool TABLE>
__global__ void __launch_bounds__(BLOCK, 8) k_shade(const FrameParams P)
{
    const unsigned int qBegin = (unsigned int)P.queueCounters[Q_RANGE_BEGIN], nHits = (unsigned int)P.queueCounters[Q_RANGE_END];
    const uint32_t nPix = uint32_t(size_t(P.rowsLocal) * size_t(P.width));
    // with a cull pass before it, the kernel walks the list of hits that pass left to shade (full waves); otherwise the
    // round's part of the hit queue
    const bool listed = P.shadeList != nullptr && P.shadowSamplesPerHit > 0;
    const unsigned int first = listed ? 0u : qBegin, count = listed ? (unsigned int)P.queueCounters[Q_SHADE_ITEMS] : nHits;
    for (unsigned int i = first + blockIdx.x * BLOCK + threadIdx.x; i < count; i += gridDim.x * BLOCK) {
        const unsigned int q = listed ? P.shadeList[P.shadowQueueReversed ? uint32_t(P.queueCapacity) - 1u - i : i] : i;
        const HitEntry e = P.queue[q];
        if (e.slot < 0 || e.pad != 0) continue; // nothing found after all / colour already written by k_cull
        Rng rng;
        RayD r;
        size_t pix;
        int sidx;
        hit_ray(P, q, e, nPix, rng, r, pix, sidx);
        const Surface s = surface_at(P, r, e.t, e.slot);
        const uint8_t* vis = P.visibility + size_t(q) * size_t(P.shadowSamplesPerHit);
        vec3 shading = splat3(0.0f);
        for (int li = 0; li < P.nLights; li++) {
            const LightDev& L = light_in<TABLE>(P, li);
            if (!light_enabled(P, L.type)) continue;
            const bool shadowed = light_shadowed(P, L.type);
            const int total = L.nA * L.nB;
            const uint32_t per = light_draws_per_sample(L.type);
            for (int k = 0; k < total; k++) {
                vec3 position, color;
                light_sample(L, k, rng.pixelKey, rng.sample, rng.randCounter + uint32_t(L.drawBase) + uint32_t(k) * per, position, color);
                const float lighted = shadowed ? float(vis[L.shadowBase + k]) : 1.0f;
                const vec3 shade = P.f.shading ? phong(position, color, r, e.t, s) : s.kd;
                shading = shading + light_weight(L, shade) * lighted;
            }
        }
        float* out = hit_colour_slot(P, q, pix, sidx);
        out[0] = shading.x;
        out[1] = shading.y;
        out[2] = shading.z;
    }
}

// Sums the sample planes of each pixel in sample order and divides like the reference does:
// supersampling: sum / float(n*n) (multipleRays.cpp:50); depth of field: (centre + lens rays) / (n + 1) when
// the pixel's flag is set, else the centre ray alone (render.cpp:109-118).
__global__ void __launch_bounds__(BLOCK) k_resolve(const FrameParams P)
{
    const size_t nPix = size_t(P.rowsLocal) * size_t(P.width);
    const size_t pix = size_t(blockIdx.x) * BLOCK + threadIdx.x;
    if (pix >= nPix) return;
    const int S = P.samplesPerPixel;
    vec3 color = load3(P.samples + pix * 3);
    if (P.f.multiple_rays) {
        color = splat3(0.0f) + color;
        for (int s = 1; s < S; s++) color = color + load3(P.samples + (size_t(s) * nPix + pix) * 3);
        color = color / float(S);
    } else if (P.dofFlags[pix]) {
        for (int s = 1; s < S; s++) color = color + load3(P.samples + (size_t(s) * nPix + pix) * 3);
        color = color / float(S);
    }
    P.out[pix * 3] = color.x;
    P.out[pix * 3 + 1] = color.y;
    P.out[pix * 3 + 2] = color.z;
}

template <bool PRUNE, bool SPH>
__global__ void __launch_bounds__(BLOCK) k_intersect(const FrameParams P, const rt_ray* __restrict__ rays, uint64_t n, rt_hit* __restrict__ hits)
{
    extern __shared__ uint2 lds_stack[];
    const uint64_t i = uint64_t(blockIdx.x) * BLOCK + threadIdx.x;
    Stack st { lds_stack + threadIdx.x, uint32_t(BLOCK) };
    Counts cnt;
    if (i < n) {
        const rt_ray ray = rays[i];
        const RayD r = make_ray(load3(ray.origin), load3(ray.direction));
        float t = ray.t;
        Attr attr;
        const int slot = intersect<PRUNE, false, true, SPH>(P, r, t, st, cnt, attr);
        rt_hit h;
        h.hit = slot >= 0;
        h.t = t;
        h.mesh = -1;
        h.triangle = -1;
        h.normal[0] = h.normal[1] = h.normal[2] = 0.0f;
        h.bary[0] = h.bary[1] = h.bary[2] = 0.0f;
        h.kd[0] = h.kd[1] = h.kd[2] = 0.0f;
        if (slot >= 0) {
            const PrimRec& p = P.scene.prims[slot];
            const Surface s = surface_at(P, r, attr.t, attr.slot);
            h.mesh = p.mesh;
            h.triangle = p.prim;
            h.normal[0] = s.normal.x; h.normal[1] = s.normal.y; h.normal[2] = s.normal.z;
            h.bary[0] = s.bary.x; h.bary[1] = s.bary.y; h.bary[2] = s.bary.z;
            h.kd[0] = s.kd.x; h.kd[1] = s.kd.y; h.kd[2] = s.kd.z;
        }
        hits[i] = h;
    }
    flush_counts(cnt, P.stats);
}

template <bool PRUNE, bool SPH>
__global__ void __launch_bounds__(BLOCK) k_visibility(const FrameParams P, const rt_ray* __restrict__ rays, uint64_t n, const float* __restrict__ samples, int nSamples, uint8_t* __restrict__ out)
{
    extern __shared__ uint2 lds_stack[];
    const uint64_t i = uint64_t(blockIdx.x) * BLOCK + threadIdx.x;
    Stack st { lds_stack + threadIdx.x, uint32_t(BLOCK) };
    Counts cnt;
    if (i < n * uint64_t(nSamples)) {
        const uint64_t ri = i / uint64_t(nSamples);
        const int si = int(i % uint64_t(nSamples));
        const rt_ray ray = rays[ri];
        const RayD r = make_ray(load3(ray.origin), load3(ray.direction));
        out[i] = visibility<PRUNE, false, SPH>(P, load3(samples + 3 * si), r, ray.t, st, cnt) != 0.0f ? 1 : 0;
    }
}

// Bloom, reference src/bloom.cpp:52-114: thresholded copy of the frame filtered with a (2k+1)^2 kernel whose
// vertical taps are strided by HEIGHT (bloom.cpp:66), bounds-checked on the flat index and on the column of
// the centre pixel only.  Taps are accumulated column-offset-major like the reference's loops.
__global__ void __launch_bounds__(BLOCK) k_bloom(const float* __restrict__ src, float* __restrict__ dst, int width, int height, const BloomParams bp)
{
    __shared__ float weights[BLOOM_MAX_TAPS * BLOOM_MAX_TAPS];
    const int k = bp.size, size = 2 * k + 1;
    // gaussianKernel, reference src/bloom.cpp:26-46: weights[a*size+b] = gaus[a][b] = (valy(b) * valx(a)) / denom
    if (bp.gaussian)
        for (int e = int(threadIdx.x); e < size * size; e += BLOCK) weights[e] = (bp.f[e % size] * bp.f[e / size]) / bp.denom;
    __syncthreads();
    const long n = long(width) * long(height);
    const long idx = long(blockIdx.x) * BLOCK + threadIdx.x;
    if (idx >= n) return;
    const float boxDiv = float(size * size);
    const int col = int(idx % width);
    vec3 avg = splat3(0.0f);
    for (int i = 0; i < size; i++)
        for (int j = 0; j < size; j++) {
            const int dx = i - k;
            const long at = idx + long(j - k) * height + dx;
            const int x = dx + col;
            if (x >= 0 && x < width && at >= 0 && at < n) {
                vec3 c = load3(src + at * 3);
                if (!(c.x > bp.threshold || c.y > bp.threshold || c.z > bp.threshold)) c = splat3(0.0f);
                if (bp.gaussian) avg = avg + c * weights[j * size + i];
                else avg = avg + c / boxDiv;
            }
        }
    const vec3 add = avg * bp.scale;
    const vec3 base = bp.debug ? splat3(0.0f) : load3(src + idx * 3);
    const vec3 o = bp.debug ? add : base + add;
    dst[idx * 3] = o.x;
    dst[idx * 3 + 1] = o.y;
    dst[idx * 3 + 2] = o.z;
}

// ---------------------------------------------------------------------------------------------------
// Wavefront evaluation of frames with bounce features (mirror reflection, see-through continuation, glossy
// reflection, reflection map): the reference's recursion (src/render.cpp:18-80) laid out as a tree of nodes per pixel
// sample and evaluated one generation at a time, so that every ray of a generation — however many calls a pixel
// sample has spawned — gets a lane of its own:
//   k_wf_trace    closest hit of every pending ray of a generation (the first launch of a chunk: of its camera rays,
//                 generated on the fly), then the branch of getFinalColor the call takes:
//                 what kind of node it becomes, the nodes of its children (allocated with one atomic per wave, so the
//                 children of a wave's hits are adjacent), and, where its value needs the direct lighting Lo, an
//                 entry in the hit queue;
//   k_wf_spawn    the sample rays of the generation's glossy calls, one lane per (call, sample);
//   (the queue passes k_cull / k_stream / k_shade: Lo of every queued hit, all generations at once)
//   k_wf_combine  values bottom-up, one generation per launch, with the reference's operations in its order; the
//                 roots write the pixel samples.
// A call's random draws do not depend on the evaluation order: light jitter is indexed from the node's randBase
// (its depth-first position on the rand() stream, see k_wf_count / k_wf_index), glossy deviates from a hash of the
// reflection ray.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wf_pixel(const FrameParams& P, uint32_t work, uint32_t nPix, size_t& pix, int& sidx, int& x, int& y)
{
    sidx = P.sampleBegin + int(work / nPix);
    pix = size_t(work % nPix);
    x = int(pix % size_t(P.width));
    y = local_row_y(P, int(pix / size_t(P.width)));
}

// ROOTS = true: the lanes are the pixel samples of the chunk themselves (work ids wfBegin .. wfEnd - 1 counted from
// wfRootBegin); their camera rays are generated here and only the samples whose call spawns another call get a node
// (most end at their first hit or miss: those write their pixel sample, or queue their hit for the lighting passes,
// directly).  ROOTS = false: the lanes are the nodes wfBegin .. wfEnd - 1 of a generation.
template <bool PRUNE, bool COUNT, bool ROOTS>
__global__ void __launch_bounds__(WAVE) k_wf_trace(const FrameParams P)
{
    extern __shared__ uint2 lds_stack[];
    Stack st { lds_stack + threadIdx.x, uint32_t(WAVE) };
    Counts cnt;
    const int lane = int(threadIdx.x);
    const uint32_t nPix = uint32_t(size_t(P.rowsLocal) * size_t(P.width));
    // the generation's node range: known to the host (roots; test paths) or read from the table the device keeps
    const uint32_t begin = (!ROOTS && P.wfGen >= 0) ? uint32_t(P.queueCounters[Q_WF_GEN + P.wfGen]) : P.wfBegin;
    const uint32_t end = (!ROOTS && P.wfGen >= 0) ? uint32_t(P.queueCounters[Q_WF_GEN + P.wfGen + 1]) : P.wfEnd;
    const uint32_t n = end - begin;
    const uint32_t rounded = (n + 63u) & ~63u; // whole waves stay in the loop together
    const bool reflMap = P.f.environment_mapping && P.p.enabled_refl_map;
    const uint32_t filter = uint32_t(P.p.glossy_filter_size);
    for (uint32_t i = blockIdx.x * WAVE + threadIdx.x; i < rounded; i += gridDim.x * WAVE) {
        uint32_t node = begin + i; // ROOTS: assigned below, and only if the sample's call has children
        uint32_t kind = WF_DEAD, nChildren = 0, depth = 0, blendDepth = 0, randBase = 0;
        bool pending = false, wantsLo = false;
        RayD r = make_ray(splat3(0.0f), splat3(1.0f));
        float t = FLT_MAX;
        int slot = -1;
        Surface s;
        uint32_t work = 0;
        size_t pix = 0;
        int sidx = 0;
        if (i < n) {
            if (ROOTS) {
                work = P.wfRootBegin + begin + i;
                int x, y;
                wf_pixel(P, work, nPix, pix, sidx, x, y);
                // depth of field: the lens rays of a pixel exist only when its centre ray (an earlier pass) said so
                pending = !(P.f.depth_of_field && sidx > 0 && P.dofFlags[pix] == 0);
                if (pending) {
                    Rng rng;
                    vec3 o, d;
                    sample_camera_ray(P, x, y, sidx, rng, o, d);
                    r = make_ray(o, d);
                    randBase = rng.randCounter;
                    depth = uint32_t(P.p.ray_depth);
                    cnt.primary++;
                }
            } else {
                const uint32_t meta = P.wfMeta[node];
                pending = wf_kind(meta) == WF_PENDING;
                if (pending) {
                    const WfRay w = P.wfRays[node];
                    work = w.work;
                    r = make_ray(load3(w.o), load3(w.d));
                    depth = wf_depth(meta);
                    blendDepth = wf_blend_depth(meta);
                    randBase = P.wfNodes[node].randBase;
                    cnt.bounce++;
                }
            }
            if (pending) {
                Attr attr;
                slot = traverse<PRUNE, false, COUNT, false>(P.scene, r, t, st, cnt, attr);
                if (ROOTS && P.f.depth_of_field && sidx == 0) P.dofFlags[pix] = dof_flag(P, r.d, slot >= 0 ? t : FLT_MAX) ? 1 : 0;
                if (slot < 0) {
                    kind = WF_MISS;
                } else {
                    // the branch of getFinalColor this call takes (render.cpp:25-67)
                    s = surface_at(P, r, t, slot);
                    const bool blend = !reflMap && P.f.transparency && s.transparency != 1.0f;
                    const bool specular = !reflMap && !blend && (P.f.recursive || P.f.glossy) && depth > 0 && (s.ks.x != 0 || s.ks.y != 0 || s.ks.z != 0);
                    if (reflMap) {
                        kind = WF_VALUE;
                    } else if (blend) {
                        if (int(blendDepth) >= P.p.max_transparency_depth) {
                            cnt.overflow++;
                            kind = WF_LO;
                        } else {
                            kind = WF_BLEND;
                            nChildren = 1;
                        }
                        wantsLo = true;
                    } else if (specular) {
                        kind = P.f.glossy ? WF_GLOSSY : WF_MIRROR;
                        nChildren = kind == WF_GLOSSY ? filter : 1u;
                    } else {
                        kind = WF_LO;
                        wantsLo = true;
                    }
                }
            }
        }
        // a glossy call without samples is ks * 0 / 0 in the reference (render.cpp:59) and here: it keeps a node
        const bool hasNode = !ROOTS || nChildren > 0 || kind == WF_GLOSSY;
        const uint32_t need = nChildren + ((ROOTS && hasNode && pending) ? 1u : 0u);
        // ---- nodes for the children (and for a root that has any): one atomic per wave
        uint32_t prefix = need; // inclusive scan over the wave
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(prefix, off, 64);
            if (lane >= off) prefix += v;
        }
        const uint32_t total = __shfl(prefix, 63, 64);
        unsigned long long base = 0;
        if (lane == 63 && total) base = atomicAdd(&P.queueCounters[Q_WF_NODES], (unsigned long long)total);
        base = __shfl(base, 63, 64);
        uint32_t firstChild = uint32_t(base) + prefix - need;
        if (ROOTS && need) node = firstChild++;
        bool fitsNodes = base + total <= (unsigned long long)P.wfCapacity;
        if (total && !fitsNodes && lane == 63) P.queueCounters[Q_WF_OVERFLOW] = 1ull;
        // ---- glossy calls of this generation: one list entry each
        const unsigned long long glossyVotes = __ballot(kind == WF_GLOSSY && nChildren > 0);
        uint32_t glossyAt = 0;
        if (glossyVotes) {
            const int leader = __ffsll((long long)glossyVotes) - 1;
            unsigned long long gb = 0;
            if (lane == leader) gb = atomicAdd(&P.queueCounters[Q_WF_GLOSSY], (unsigned long long)__popcll(glossyVotes));
            gb = __shfl(gb, leader, 64);
            if (gb + (unsigned long long)__popcll(glossyVotes) > (unsigned long long)P.wfGlossyCapacity) {
                fitsNodes = false;
                if (lane == leader) P.queueCounters[Q_WF_OVERFLOW] = 1ull;
            }
            glossyAt = uint32_t(gb) + lanes_below(glossyVotes);
            // A list entry that was reserved but cannot be used (this wave's nodes or the list itself did not fit) must
            // not keep what an earlier chunk left there: k_wf_spawn runs before the host learns of the overflow.
            if (!fitsNodes && kind == WF_GLOSSY && nChildren > 0 && glossyAt < P.wfGlossyCapacity) P.wfGlossy[glossyAt].node = 0xffffffffu;
        }
        // ---- hit queue entries of the calls whose value needs Lo (the queue holds an entry per root and per node)
        const unsigned long long loVotes = __ballot(wantsLo);
        uint32_t hitAt = 0;
        if (loVotes) {
            const int leader = __ffsll((long long)loVotes) - 1;
            unsigned long long hb = 0;
            if (lane == leader) hb = atomicAdd(&P.queueCounters[Q_HITS], (unsigned long long)__popcll(loVotes));
            hb = __shfl(hb, leader, 64);
            hitAt = uint32_t(hb) + lanes_below(loVotes);
        }
        if (!pending) continue;
        if (!fitsNodes && need) {
            // the chunk is abandoned and rendered again in smaller pieces; leave nothing half-written behind
            kind = WF_MISS;
            nChildren = 0;
            wantsLo = false;
        }
        const bool nodeWritten = !ROOTS || (hasNode && fitsNodes);
        if (wantsLo) {
            HitEntry e;
            e.work = work;
            e.t = t;
            e.slot = slot;
            e.pad = 0;
            P.queue[hitAt] = e;
            P.wfHitNode[hitAt] = nodeWritten ? node : 0xffffffffu; // no node: a camera ray's hit, coloured into its pixel sample
        }
        if (kind == WF_MISS || kind == WF_VALUE) {
            const vec3 c = (kind == WF_VALUE || (P.f.environment_mapping && P.p.enabled_skybox)) ? skybox_color(P.scene, r.d) : splat3(0.0f);
            float* v = nodeWritten ? P.wfValue + 3 * size_t(node) : sample_slot(P, pix, sidx);
            v[0] = c.x; v[1] = c.y; v[2] = c.z;
        }
        if (!nodeWritten) continue;
        WfNode& nd = P.wfNodes[node];
        nd.t = t;
        nd.slot = slot;
        nd.firstChild = firstChild;
        P.wfMeta[node] = wf_meta(kind, depth, blendDepth) | (ROOTS ? WF_ROOT : 0u);
        if (ROOTS) {
            nd.randBase = randBase;
            WfRay& w = P.wfRays[node];
            w.o[0] = r.o.x; w.o[1] = r.o.y; w.o[2] = r.o.z;
            w.d[0] = r.d.x; w.d[1] = r.d.y; w.d[2] = r.d.z;
            w.work = work;
            w.parent = 0xffffffffu;
        }
        const uint32_t childRand = randBase + uint32_t(P.drawsPerHit); // the call's lighting draws come before its child's
        if (kind == WF_BLEND) {
            nd.a = s.transparency;
            const vec3 o = (t + 0.000001f) * r.d + r.o; // render.cpp:40
            WfRay& w = P.wfRays[firstChild];
            w.o[0] = o.x; w.o[1] = o.y; w.o[2] = o.z;
            w.d[0] = r.d.x; w.d[1] = r.d.y; w.d[2] = r.d.z;
            w.work = work;
            w.parent = node;
            P.wfMeta[firstChild] = wf_meta(WF_PENDING, depth, blendDepth + 1u);
            P.wfNodes[firstChild].randBase = childRand;
        } else if (kind == WF_MIRROR || kind == WF_GLOSSY) {
            nd.coef[0] = s.ks.x; nd.coef[1] = s.ks.y; nd.coef[2] = s.ks.z;
            // computeReflectionRay, reference src/shading.cpp:42-51
            const float epsilon = 0.00001f;
            const vec3 rd = r.d - (2 * dot(r.d, s.normal)) * s.normal;
            const vec3 ro = t * r.d * (1 - epsilon) + r.o;
            if (kind == WF_MIRROR) {
                WfRay& w = P.wfRays[firstChild];
                w.o[0] = ro.x; w.o[1] = ro.y; w.o[2] = ro.z;
                w.d[0] = rd.x; w.d[1] = rd.y; w.d[2] = rd.z;
                w.work = work;
                w.parent = node;
                P.wfMeta[firstChild] = wf_meta(WF_PENDING, depth - 1u, blendDepth);
                P.wfNodes[firstChild].randBase = childRand;
            } else if (nChildren) {
                WfGlossy& g = P.wfGlossy[glossyAt];
                g.origin[0] = ro.x; g.origin[1] = ro.y; g.origin[2] = ro.z;
                g.dir[0] = rd.x; g.dir[1] = rd.y; g.dir[2] = rd.z;
                g.normal[0] = s.normal.x; g.normal[1] = s.normal.y; g.normal[2] = s.normal.z;
                g.side = P.p.glossy_sigma / (s.shininess * 3.0f);
                g.node = node;
                g.key = rng_ray_key(ro, rd);
            }
        }
    }
    flush_counts(cnt, P.stats);
}

// Closes generation `gen` of a chunk on the device (gen 0 = the camera rays), so that the next one can be launched
// without the host knowing how many calls it has.  The generation table T (Q_WF_GEN) holds the node count at every
// generation boundary: T[0] = 0, generation g >= 1 traces the nodes [T[g - 1], T[g]) and this kernel, launched behind
// its trace and sample generation, sets T[g + 1] to the nodes allocated so far.  After an overflow the following
// generations are left empty.  The list of glossy calls starts again, and the host's copy of the chunk's state (pinned
// memory, read after the one wait per chunk) is brought up to date.
__global__ void k_wf_next(const FrameParams P, int gen)
{
    unsigned long long* c = P.queueCounters;
    const unsigned long long nodes = c[Q_WF_NODES];
    const bool overflow = c[Q_WF_OVERFLOW] != 0ull;
    const unsigned long long now = overflow ? c[Q_WF_GEN + gen] : (nodes < (unsigned long long)P.wfCapacity ? nodes : (unsigned long long)P.wfCapacity);
    c[Q_WF_GEN + gen + 1] = now;
    c[Q_WF_GLOSSY] = 0ull;
    if (P.wfReport) {
        P.wfReport[0] = nodes;
        P.wfReport[1] = overflow ? 1ull : 0ull;
        P.wfReport[2 + gen + 1] = now;
    }
}

// dst = src (add = 0) or dst += src, src = 0 (add = 1): the ray counters of a chunk are kept apart until the chunk is
// known to stand, and the counters before a pass are kept in case the pass is rendered again another way
__global__ void k_stats(unsigned long long* dst, unsigned long long* src, int add)
{
    const int i = int(threadIdx.x);
    if (i >= STAT_COUNT) return;
    if (add) {
        dst[i] += src[i];
        src[i] = 0ull;
    } else {
        dst[i] = src[i];
    }
}

// One lane per (glossy call of the generation, sample): the perturbed reflection ray of reference src/gloss.cpp:14-51,
// same arithmetic as the glossy frames of final_color.
__global__ void __launch_bounds__(BLOCK) k_wf_spawn(const FrameParams P)
{
    const uint32_t filter = uint32_t(P.p.glossy_filter_size);
    // (after an overflow the counter can exceed the list, and entries of waves whose nodes did not fit are void)
    unsigned long long calls = P.queueCounters[Q_WF_GLOSSY];
    if (calls > (unsigned long long)P.wfGlossyCapacity) calls = P.wfGlossyCapacity;
    const unsigned long long n = calls * filter;
    const uint32_t nPix = uint32_t(size_t(P.rowsLocal) * size_t(P.width));
    for (unsigned long long i = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * BLOCK) {
        const WfGlossy g = P.wfGlossy[i / filter];
        const uint32_t k = uint32_t(i % filter);
        if (g.node >= P.wfCapacity) continue; // void entry
        const WfNode& parent = P.wfNodes[g.node];
        const uint32_t parentMeta = P.wfMeta[g.node];
        const uint32_t child = parent.firstChild + k;
        if (child >= P.wfCapacity) continue;
        const uint32_t work = P.wfRays[g.node].work;
        size_t pix;
        int sidx, x, y;
        wf_pixel(P, work, nPix, pix, sidx, x, y);
        Rng rng;
        rng.pixelKey = rng_pixel_key(P.p.seed, uint32_t(y * P.width + x));
        rng.sample = uint32_t(sidx);
        const vec3 dir = load3(g.dir), normal = load3(g.normal);
        const vec3 w = normalize(dir);
        const float awx = fabsf(w.x), awy = fabsf(w.y), awz = fabsf(w.z);
        const int minTerm = (awx <= awy && awx <= awz) ? 0 : ((awy <= awx && awy <= awz) ? 1 : 2);
        vec3 tt = w;
        if (minTerm == 0) tt.x = 1; else if (minTerm == 1) tt.y = 1; else tt.z = 1;
        const vec3 u = normalize(cross(tt, w));
        const vec3 v = cross(w, u);
        const float offset = -g.side / 2.0f;
        const uint32_t idx = g.key + 2u * k;
        const double n0 = rng.gauss(idx), n1 = rng.gauss(idx + 1u);
        const vec3 rp = (dir + float(double(offset) + n0 * double(g.side)) * u) + float(double(offset) + n1 * double(g.side)) * v;
        WfNode& nd = P.wfNodes[child];
        if (dot(rp, normal) > 0) {
            const vec3 d = normalize(rp);
            WfRay& ray = P.wfRays[child];
            ray.o[0] = g.origin[0]; ray.o[1] = g.origin[1]; ray.o[2] = g.origin[2];
            ray.d[0] = d.x; ray.d[1] = d.y; ray.d[2] = d.z;
            ray.work = work;
            ray.parent = g.node;
            P.wfMeta[child] = wf_meta(WF_PENDING, wf_depth(parentMeta) - 1u, wf_blend_depth(parentMeta));
            nd.randBase = parent.randBase + uint32_t(P.drawsPerHit);
        } else {
            P.wfMeta[child] = wf_meta(WF_DEAD, 0, 0);
        }
    }
}

// Position of every call on the rand() stream when calls fan out (glossy reflection) AND the lights are jittered: the
// reference draws in the order its recursion runs, so a call's first draw comes after the draws of everything the
// recursion evaluated before it — its ancestors' lighting and the whole subtrees of its earlier siblings.  The shape
// of the tree does not depend on those draws (they only jitter light samples), so it is expanded first; then
//   k_wf_count  bottom-up: calls in every node's subtree that evaluate the lighting (every call that hit something);
//   k_wf_index  top-down: a parent hands each child the stream position after its own draws and its earlier
//               children's subtrees.
// Chains of single children need neither: k_wf_trace gives the child its position directly.
__global__ void __launch_bounds__(BLOCK) k_wf_count(const FrameParams P)
{
    const uint32_t n = P.wfEnd - P.wfBegin;
    const uint32_t filter = uint32_t(P.p.glossy_filter_size);
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const uint32_t node = P.wfBegin + i;
        const uint32_t meta = P.wfMeta[node];
        if (meta & WF_ROOT) continue; // nothing asks for a pixel sample's own total
        const uint32_t kind = wf_kind(meta);
        uint32_t hits = (kind == WF_DEAD || kind == WF_MISS || kind == WF_PENDING) ? 0u : 1u;
        if (kind == WF_BLEND || kind == WF_MIRROR) hits += P.wfHits[P.wfNodes[node].firstChild];
        else if (kind == WF_GLOSSY)
            for (uint32_t k = 0; k < filter; k++) hits += P.wfHits[P.wfNodes[node].firstChild + k];
        P.wfHits[node] = hits;
    }
}
__global__ void __launch_bounds__(BLOCK) k_wf_index(const FrameParams P)
{
    const uint32_t n = P.wfEnd - P.wfBegin;
    const uint32_t filter = uint32_t(P.p.glossy_filter_size);
    const uint32_t draws = uint32_t(P.drawsPerHit);
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const uint32_t node = P.wfBegin + i;
        const uint32_t meta = P.wfMeta[node];
        const uint32_t kind = wf_kind(meta);
        if (((meta & WF_ROOT) != 0u) != (P.wfRootsTurn != 0) || (kind != WF_BLEND && kind != WF_MIRROR && kind != WF_GLOSSY)) continue;
        const WfNode& nd = P.wfNodes[node];
        uint32_t at = nd.randBase + draws; // this call's own lighting draws come first (render.cpp:25)
        const uint32_t children = kind == WF_GLOSSY ? filter : 1u;
        for (uint32_t k = 0; k < children; k++) {
            const uint32_t c = nd.firstChild + k;
            P.wfNodes[c].randBase = at;
            at += draws * P.wfHits[c];
        }
    }
}

// Values of one generation from those of the next (launched from the deepest generation upwards).  The nodes of the
// pixel samples themselves sit in the first generation's range next to their children: they are skipped there and
// evaluated by a last launch of their own (wfRootsTurn), which also writes the pixel samples.
__global__ void __launch_bounds__(BLOCK) k_wf_combine(const FrameParams P)
{
    const uint32_t n = P.wfEnd - P.wfBegin;
    const uint32_t rounded = (n + 63u) & ~63u; // whole waves stay in the loop together (the glossy sums are a wave's job)
    const uint32_t nPix = uint32_t(size_t(P.rowsLocal) * size_t(P.width));
    const uint32_t filter = uint32_t(P.p.glossy_filter_size);
    const int lane = int(threadIdx.x) & 63;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < rounded; i += gridDim.x * BLOCK) {
        const uint32_t node = P.wfBegin + i;
        uint32_t kind = WF_DEAD;
        bool root = false;
        if (i < n) {
            const uint32_t meta = P.wfMeta[node];
            root = (meta & WF_ROOT) != 0u;
            // (the value of a call that ends at its hit or miss was written where it was found: nothing to combine)
            if (root == (P.wfRootsTurn != 0)) kind = wf_kind(meta);
        }
        vec3 result = splat3(0.0f);
        uint32_t firstChild = 0;
        vec3 coef = splat3(0.0f);
        if (kind == WF_BLEND || kind == WF_MIRROR || kind == WF_GLOSSY) {
            const WfNode& nd = P.wfNodes[node];
            firstChild = nd.firstChild;
            coef = load3(nd.coef);
            if (kind == WF_BLEND) result = nd.a * coef + (1.0f - nd.a) * load3(P.wfValue + 3 * size_t(firstChild)); // render.cpp:43
            else if (kind == WF_MIRROR) result = coef * load3(P.wfValue + 3 * size_t(firstChild));                   // render.cpp:61
        }
        // Glossy calls: the children's values are summed in sample order like the reference does (render.cpp:56-58).  A
        // lane walking its own 64 children would touch 64 scattered cache lines per step of the wave; instead the wave
        // takes the calls of its lanes one at a time, loads a call's children with one coalesced read (child k by lane
        // k) and runs the ordered sum on values handed round from the lanes' registers.
        unsigned long long todo = __ballot(kind == WF_GLOSSY);
        while (todo) {
            const int owner = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const uint32_t first = uint32_t(__builtin_amdgcn_readlane(int(firstChild), owner));
            vec3 sum = splat3(0.0f);
            for (uint32_t base = 0; base < filter; base += 64u) {
                const uint32_t k = base + uint32_t(lane);
                bool alive = false;
                vec3 v = splat3(0.0f);
                if (k < filter) {
                    alive = wf_kind(P.wfMeta[first + k]) != WF_DEAD;
                    if (alive) v = load3(P.wfValue + 3 * size_t(first + k));
                }
                const unsigned long long live = __ballot(alive);
                const uint32_t count = filter - base < 64u ? filter - base : 64u;
                for (uint32_t j = 0; j < count; j++) {
                    if (!((live >> j) & 1ull)) continue;
                    sum = sum + make3(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.x), int(j))),
                                      __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.y), int(j))),
                                      __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.z), int(j))));
                }
            }
            if (lane == owner) result = coef * sum / float(filter); // render.cpp:59
        }
        if (kind != WF_BLEND && kind != WF_MIRROR && kind != WF_GLOSSY) continue;
        if (root) {
            size_t pix;
            int sidx, x, y;
            wf_pixel(P, P.wfRays[node].work, nPix, pix, sidx, x, y);
            float* out = sample_slot(P, pix, sidx);
            out[0] = result.x; out[1] = result.y; out[2] = result.z;
        } else {
            float* v = P.wfValue + 3 * size_t(node);
            v[0] = result.x; v[1] = result.y; v[2] = result.z;
        }
    }
}

size_t stack_bytes(const FrameParams& P, int threads)
{
    int depth = P.scene.stackDepth < 1 ? 1 : P.scene.stackDepth;
    return size_t(threads) * size_t(depth) * sizeof(uint2);
}

// Blocks of `kernel` that can be resident on the whole chip at once (persistent kernels are sized to this).
template <class K>
uint32_t resident_blocks(K kernel, int threads, size_t lds)
{
    // The answer depends on (kernel, threads, lds, device).  The render kernels all have the same function type, so
    // the kernel's address is part of the key; each thread remembers its recent queries (a frame alternates between
    // a handful of them).
    struct Entry { const void* kernel; int threads, dev; size_t lds; uint32_t blocks; };
    constexpr int SLOTS = 16;
    static thread_local Entry cache[SLOTS] = {};
    static thread_local int next = 0;
    int perCu = 0, dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256u * 8u;
    const void* key = reinterpret_cast<const void*>(kernel);
    for (const Entry& e : cache)
        if (e.kernel == key && e.dev == dev && e.threads == threads && e.lds == lds) return e.blocks;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return 256u * 8u;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, kernel, threads, lds) != hipSuccess || perCu < 1) perCu = 4;
    cache[next] = Entry { key, threads, dev, lds, uint32_t(perCu) * uint32_t(cus) };
    next = (next + 1) % SLOTS;
    return uint32_t(perCu) * uint32_t(cus);
}

// Launches a render kernel with one-wave blocks: as many as are resident at once, or fewer if the pass is smaller.
template <class K>
void launch_walkers(K kernel, const FrameParams& P, size_t lds, hipStream_t stream)
{
    const uint32_t cap = resident_blocks(kernel, WAVE, lds);
    hipLaunchKernelGGL(kernel, dim3(P.logicalBlocks < cap ? P.logicalBlocks : cap), dim3(WAVE), lds, stream, P);
}

template <bool PRUNE, bool GLOSSY>
hipError_t launch_render_t(const FrameParams& P, size_t lds, hipStream_t stream)
{
    if (P.scene.hasSpheres) {
        if (P.countTests) launch_walkers(k_render<PRUNE, GLOSSY, true, true>, P, lds, stream);
        else launch_walkers(k_render<PRUNE, GLOSSY, false, true>, P, lds, stream);
    } else {
        if (P.countTests) launch_walkers(k_render<PRUNE, GLOSSY, true, false>, P, lds, stream);
        else launch_walkers(k_render<PRUNE, GLOSSY, false, false>, P, lds, stream);
    }
    return hipGetLastError();
}

// The stream kernel on every resident wave (how many queries there are is only known on the device).
template <bool COUNT, bool PRIMARY, int G>
void launch_stream_g(const FrameParams& P, uint32_t wanted, size_t lds, hipStream_t stream)
{
    uint32_t cap = resident_blocks(k_stream<COUNT, PRIMARY, G>, WAVE, lds);
    if (P.pairSpillEntries > 0 && cap > P.pairSpillWaves) cap = P.pairSpillWaves;
    hipLaunchKernelGGL((k_stream<COUNT, PRIMARY, G>), dim3(wanted < cap ? wanted : cap), dim3(WAVE), lds, stream, P);
}
template <bool PRIMARY>
void launch_stream(const FrameParams& P, uint32_t wanted, size_t lds, hipStream_t stream)
{
    // camera rays that gave up are unrelated rays: no groups there
    const int group = PRIMARY ? 1 : P.streamGroup;
    if (P.countTests) {
        if (group == 2) launch_stream_g<true, PRIMARY, PRIMARY ? 1 : 2>(P, wanted, lds, stream);
        else launch_stream_g<true, PRIMARY, 1>(P, wanted, lds, stream);
    } else {
        if (group == 2) launch_stream_g<false, PRIMARY, PRIMARY ? 1 : 2>(P, wanted, lds, stream);
        else launch_stream_g<false, PRIMARY, 1>(P, wanted, lds, stream);
    }
}

// The queue passes of one sample range.  The hits of a pass come in two lots: those k_primary queues itself (nearly
// all) and those of the camera rays that gave up, which k_stream (primary mode) and k_closest finish — two kernels
// that run a handful of long rays each and leave the chip almost empty for most of a millisecond.  With `aux` (a
// second stream of the library's own, a second set of counters and spill space) the second lot gets a lighting round
// of its own, and that round together with the two kernels runs on the second stream beside the first lot's
// lighting: round A = entries [0, a) of the hit queue, shadow items stored from the front of their queue; round
// B = entries [a, end), items stored from the back; the visibility bytes are indexed by hit and never overlap.
template <bool PRUNE>
hipError_t launch_split_t(const FrameParams& P, dim3 gridDirect, size_t lds, hipStream_t stream, hipEvent_t* ev, const SplitAux* aux)
{
    const size_t streamLds = stream_lds_bytes(P.pairStackLds);
    {
        // enough chunks for every resident wave to come back to the cursor several times
        const bool sky = P.f.environment_mapping != 0 || P.f.depth_of_field != 0; // (the kernel's variants with more live registers)
        const auto kernel = P.countTests ? (sky ? k_primary<PRUNE, true, 7> : k_primary<PRUNE, true, 8>) : (sky ? k_primary<PRUNE, false, 7> : k_primary<PRUNE, false, 8>);
        const uint32_t waves = resident_blocks(kernel, WAVE, size_t(WAVE) * size_t(P.primaryStack) * sizeof(uint2));
        uint32_t chunk = P.logicalBlocks / (waves * 8u);
        chunk = chunk < 1u ? 1u : (chunk > PRIMARY_CHUNK_MAX ? PRIMARY_CHUNK_MAX : chunk);
        FrameParams Q = P;
        Q.primaryChunk = chunk;
        const size_t ldsPrimary = size_t(WAVE) * size_t(P.primaryStack) * sizeof(uint2);
        // (A smaller budget for small passes — a rank's share of a frame — was tried and dropped: it shortens this kernel
        // and lengthens the chain of kernels behind it by more: slowest share of 2 / 4 ranks 6.1 / 3.7 -> 5.3 / 3.3 ms
        // without it, of 8 ranks the same.)
        launch_walkers(kernel, Q, ldsPrimary, stream);
    }
    auto finish_gave_up = [&](hipStream_t st) {
        // how many rays gave up, and how many of those meet a primitive, is only known on the device: both kernels
        // get a grid that covers every resident wave (on the second stream the stream kernel spills to that stream's
        // space: the first round's shadow pass may be running)
        FrameParams G = P;
        if (aux) {
            G.pairSpill = aux->pairSpill;
            G.pairSpillWaves = aux->pairSpillWaves;
        }
        launch_stream<true>(G, gridDirect.x, streamLds, st);
        // (on the second stream: stacks in device memory, no LDS, so that the kernel fits beside the first round's
        // shadow pass instead of waiting for its waves to end)
        const bool deviceStacks = aux && aux->laneStacks;
        if (deviceStacks) {
            G.laneStacks = aux->laneStacks;
            G.laneStackWaves = aux->laneStackWaves;
        }
        const size_t ldsClosest = deviceStacks ? 0 : lds;
        uint32_t walkers = P.countTests ? resident_blocks(k_closest<PRUNE, true>, WAVE, ldsClosest) : resident_blocks(k_closest<PRUNE, false>, WAVE, ldsClosest);
        if (deviceStacks && walkers > aux->laneStackWaves) walkers = aux->laneStackWaves;
        const dim3 gc(gridDirect.x < walkers ? gridDirect.x : walkers);
        if (P.countTests) hipLaunchKernelGGL((k_closest<PRUNE, true>), gc, dim3(WAVE), ldsClosest, st, G);
        else hipLaunchKernelGGL((k_closest<PRUNE, false>), gc, dim3(WAVE), ldsClosest, st, G);
    };
    auto lighting = [&](const FrameParams& R, hipStream_t st, hipEvent_t* marks, hipEvent_t* shadowOnly = nullptr) {
        if (R.shadowSamplesPerHit > 0) hipLaunchKernelGGL(R.lightTable ? k_cull<true> : k_cull<false>, dim3(256 * 8), dim3(BLOCK), 0, st, R);
        if (marks) (void)hipEventRecord(marks[1], st);
        if (shadowOnly) (void)hipEventRecord(shadowOnly[0], st);
        if (R.shadowSamplesPerHit > 0) launch_stream<false>(R, gridDirect.x, streamLds, st);
        if (shadowOnly) (void)hipEventRecord(shadowOnly[1], st);
        if (marks) (void)hipEventRecord(marks[2], st);
        hipLaunchKernelGGL(R.lightTable ? k_shade<true> : k_shade<false>, dim3(256 * 8), dim3(BLOCK), 0, st, R);
        if (marks) (void)hipEventRecord(marks[3], st);
    };
    if (!aux) {
        finish_gave_up(stream);
        hipLaunchKernelGGL(k_range, dim3(1), dim3(1), 0, stream, P.queueCounters, (const unsigned long long*)nullptr, (const unsigned long long*)&P.primaryCounters[Q_HITS]);
        if (ev) (void)hipEventRecord(ev[0], stream);
        lighting(P, stream, ev);
        return hipGetLastError();
    }
    hipError_t e;
    // round A: what k_primary queued
    hipLaunchKernelGGL(k_range, dim3(1), dim3(1), 0, stream, P.queueCounters, (const unsigned long long*)nullptr, (const unsigned long long*)&P.primaryCounters[Q_HITS]);
    if (ev) (void)hipEventRecord(ev[0], stream);
    if ((e = hipEventRecord(aux->fork, stream)) != hipSuccess) return e;
    if ((e = hipStreamWaitEvent(aux->stream, aux->fork, 0)) != hipSuccess) return e;
    // second stream: the rays that gave up, then round B
    finish_gave_up(aux->stream);
    FrameParams B = P;
    B.queueCounters = aux->counters;
    B.shadowQueueReversed = 1;
    B.streamGroup = 1; // the second round's few queries belong to scattered hits: nothing to share an entry
    B.pairSpill = aux->pairSpill;
    B.pairSpillWaves = aux->pairSpillWaves;
    hipLaunchKernelGGL(k_range, dim3(1), dim3(1), 0, aux->stream, B.queueCounters, (const unsigned long long*)&P.queueCounters[Q_RANGE_END], (const unsigned long long*)&P.primaryCounters[Q_HITS]);
    lighting(B, aux->stream, nullptr, ev ? aux->shadowMarks : nullptr);
    if ((e = hipEventRecord(aux->join, aux->stream)) != hipSuccess) return e;
    lighting(P, stream, ev);
    if ((e = hipStreamWaitEvent(stream, aux->join, 0)) != hipSuccess) return e;
    return hipGetLastError();
}


// Renders the pixel samples [sBegin, sEnd) of the call's rows as wavefront passes, in chunks of roots sized so that
// the call trees of a chunk fit the node buffer.  A chunk that turns out not to fit (the fan-out of glossy calls is
// only known once they are traced) is abandoned and rendered again as smaller chunks; what a node computes does not
// depend on the chunking.
// The generations of a chunk are launched without the host knowing their sizes: every trace kernel reads its node range
// from the generation table on the device (k_wf_next) and runs on a resident-size grid.  The host launches as many
// generations as the previous chunk needed, then waits ONCE per chunk and reads, from pinned memory, the table and the
// overflow flag — which it needs to size the value passes and to decide whether the chunk stands; if the tree turns out
// deeper than launched, it launches two more generations and waits again.  The ray counters of a chunk go to a block of
// their own (P.stats + STAT_COUNT) and are added to the call's only when the chunk stands.
template <bool PRUNE>
hipError_t wavefront_pass(FrameParams& P, int sBegin, int sEnd, size_t lds, hipStream_t stream, unsigned long long& nodesEvaluated)
{
    static_assert(STAT_COUNT <= 64, "k_stats runs one wave");
    hipError_t e;
    if (!P.wfReport) return hipErrorInvalidValue;
    P.wfReport[2] = 0ull; // T[0]
    P.sampleBegin = sBegin;
    P.sampleEnd = sEnd;
    const unsigned long long totalRoots = (unsigned long long)P.rowsLocal * (unsigned long long)P.width * (unsigned long long)(sEnd - sBegin);
    unsigned long long chunk = P.wfCapacity; // (the hit queue holds an entry per root and per node)
    if (chunk < 1) chunk = 1;
    unsigned long long done = 0;
    unsigned long long* const callStats = P.stats;
    unsigned long long* const chunkStats = P.stats ? P.stats + STAT_COUNT : nullptr;
    struct RestoreStats {
        FrameParams& P;
        unsigned long long* stats;
        ~RestoreStats() { P.stats = stats; }
    } restore { P, callStats };
    P.stats = chunkStats;
    // the deepest tree the features allow: a specular bounce per unit of ray_depth, and see-through continuations
    int maxGenerations = 1 + ((P.f.recursive || P.f.glossy) ? P.p.ray_depth : 0) + (P.f.transparency ? P.p.max_transparency_depth : 0);
    if (maxGenerations > WF_MAX_GENERATIONS) maxGenerations = WF_MAX_GENERATIONS;
    static thread_local int generationsHint = 3; // generations the last chunk needed beyond its camera rays
    std::vector<uint32_t> gens;
    const uint32_t capTrace = P.countTests ? resident_blocks(k_wf_trace<PRUNE, true, false>, WAVE, lds) : resident_blocks(k_wf_trace<PRUNE, false, false>, WAVE, lds);
    while (done < totalRoots) {
        const uint32_t nRoots = uint32_t(totalRoots - done < chunk ? totalRoots - done : chunk);
        if ((e = hipMemsetAsync(P.queueCounters, 0, Q_COUNT * sizeof(unsigned long long), stream)) != hipSuccess) return e;
        if (chunkStats && (e = hipMemsetAsync(chunkStats, 0, STAT_COUNT * sizeof(unsigned long long), stream)) != hipSuccess) return e;
        P.wfRootBegin = uint32_t(done);
        P.wfBegin = 0;
        P.wfEnd = nRoots;
        P.wfGen = -1;
        P.wfLighting = 0;
        P.wfRootsTurn = 0;
        {
            const uint32_t waves = (nRoots + WAVE - 1) / WAVE;
            const uint32_t cap = P.countTests ? resident_blocks(k_wf_trace<PRUNE, true, true>, WAVE, lds) : resident_blocks(k_wf_trace<PRUNE, false, true>, WAVE, lds);
            const dim3 grid(waves < cap ? waves : cap);
            if (P.countTests) hipLaunchKernelGGL((k_wf_trace<PRUNE, true, true>), grid, dim3(WAVE), lds, stream, P);
            else hipLaunchKernelGGL((k_wf_trace<PRUNE, false, true>), grid, dim3(WAVE), lds, stream, P);
            if (P.f.glossy) hipLaunchKernelGGL(k_wf_spawn, dim3(256 * 8), dim3(BLOCK), 0, stream, P);
            hipLaunchKernelGGL(k_wf_next, dim3(1), dim3(1), 0, stream, P, 0);
        }
        // generations 1 .. : launched ahead, sized on the device
        int launched = 0, wanted = generationsHint < 1 ? 1 : generationsHint;
        bool overflow = false;
        int complete = -1; // first generation that made no call
        while (true) {
            if (wanted > maxGenerations) wanted = maxGenerations;
            for (; launched < wanted; launched++) {
                P.wfGen = launched; // generation launched + 1 traces [T[launched], T[launched + 1])
                if (P.countTests) hipLaunchKernelGGL((k_wf_trace<PRUNE, true, false>), dim3(capTrace), dim3(WAVE), lds, stream, P);
                else hipLaunchKernelGGL((k_wf_trace<PRUNE, false, false>), dim3(capTrace), dim3(WAVE), lds, stream, P);
                if (P.f.glossy) hipLaunchKernelGGL(k_wf_spawn, dim3(256 * 8), dim3(BLOCK), 0, stream, P);
                hipLaunchKernelGGL(k_wf_next, dim3(1), dim3(1), 0, stream, P, launched + 1);
            }
            P.wfGen = -1;
            if ((e = hipGetLastError()) != hipSuccess) return e;
            if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e; // the one wait of the chunk
            const volatile unsigned long long* report = P.wfReport;
            if (report[1] != 0ull) {
                overflow = true;
                break;
            }
            for (int g = 0; g <= launched && complete < 0; g++)
                if (report[2 + g + 1] == report[2 + g]) complete = g;
            if (complete >= 0 || launched >= maxGenerations) break;
            wanted = launched + 2;
        }
        if (overflow) {
            if (nRoots <= 1) return hipErrorOutOfMemory; // one pixel sample's call tree exceeds the node buffer
            chunk = nRoots / 2u;
            continue; // (the chunk's ray counters are dropped with it)
        }
        if (complete < 0) complete = launched; // (cannot happen: the features bound the tree's depth)
        generationsHint = complete < 1 ? 1 : complete;
        gens.clear();
        for (int g = 0; g <= complete; g++) gens.push_back(uint32_t(P.wfReport[2 + g]));
        if (P.f.glossy && P.drawsPerHit > 0 && gens.size() > 1) {
            auto blocks_for = [](uint32_t n) { const uint32_t b = (n + BLOCK - 1) / BLOCK; return dim3(b < 256u * 16u ? b : 256u * 16u); };
            for (size_t g = gens.size() - 1; g-- > 0;) { // subtree totals, deepest generation first
                P.wfBegin = gens[g];
                P.wfEnd = gens[g + 1];
                hipLaunchKernelGGL(k_wf_count, blocks_for(P.wfEnd - P.wfBegin), dim3(BLOCK), 0, stream, P);
            }
            for (size_t g = 0; g + 1 < gens.size(); g++) { // stream positions, top-down; the pixel samples' own nodes first
                P.wfBegin = gens[g];
                P.wfEnd = gens[g + 1];
                if (g == 0) {
                    P.wfRootsTurn = 1;
                    hipLaunchKernelGGL(k_wf_index, blocks_for(P.wfEnd - P.wfBegin), dim3(BLOCK), 0, stream, P);
                    P.wfRootsTurn = 0;
                }
                hipLaunchKernelGGL(k_wf_index, blocks_for(P.wfEnd - P.wfBegin), dim3(BLOCK), 0, stream, P);
            }
        }
        // direct lighting of every call that needs it, all generations at once
        hipLaunchKernelGGL(k_range, dim3(1), dim3(1), 0, stream, P.queueCounters, (const unsigned long long*)nullptr, (const unsigned long long*)&P.queueCounters[Q_HITS]);
        P.wfLighting = 1;
        if (P.shadowSamplesPerHit > 0) {
            hipLaunchKernelGGL(P.lightTable ? k_cull<true> : k_cull<false>, dim3(256 * 8), dim3(BLOCK), 0, stream, P);
            launch_stream<false>(P, 256u * 32u, stream_lds_bytes(P.pairStackLds), stream);
        }
        hipLaunchKernelGGL(P.lightTable ? k_shade<true> : k_shade<false>, dim3(256 * 8), dim3(BLOCK), 0, stream, P);
        P.wfLighting = 0;
        // values, deepest generation first; the pixel samples' own nodes (in the first range) last
        for (int turn = 0; turn < 2; turn++) {
            P.wfRootsTurn = turn;
            for (size_t g = gens.size() - 1; g-- > 0;) {
                if (turn == 1 && g != 0) continue;
                P.wfBegin = gens[g];
                P.wfEnd = gens[g + 1];
                const uint32_t n = P.wfEnd - P.wfBegin;
                const uint32_t blocks = (n + BLOCK - 1) / BLOCK;
                hipLaunchKernelGGL(k_wf_combine, dim3(blocks < 256u * 16u ? blocks : 256u * 16u), dim3(BLOCK), 0, stream, P);
            }
        }
        P.wfRootsTurn = 0;
        if (chunkStats) hipLaunchKernelGGL(k_stats, dim3(1), dim3(64), 0, stream, callStats, chunkStats, 1);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        nodesEvaluated += (unsigned long long)nRoots + gens.back();
        done += nRoots;
        // next chunk: as many roots as fill 60 % of the node buffer at the node-per-root rate just seen (a chunk that had
        // to be halved because of a few heavy samples grows back afterwards)
        const double perRoot = double(gens.back()) / double(nRoots);
        const double fit = 0.6 * double(P.wfCapacity) / (perRoot > 1e-3 ? perRoot : 1e-3);
        chunk = fit >= double(P.wfCapacity) ? P.wfCapacity : (fit < 1.0 ? 1ull : (unsigned long long)fit);
    }
    return hipSuccess;
}

} // namespace

void size_pair_stack(FrameParams& P)
{
    const int bound = stream_stack_entries(P.scene.stackDepth);
    if (P.pairStackLds > bound) P.pairStackLds = bound;
    if (P.pairStackLds < 64) P.pairStackLds = bound < 64 ? bound : 64;
    P.pairSpillEntries = uint32_t(bound - P.pairStackLds);
    P.pairSpillWaves = STREAM_MAX_WAVES;
    P.pairSpill = nullptr;
}

bool direct_only(const FrameParams& P)
{
    return !(P.f.recursive || P.f.glossy || P.f.transparency || (P.f.environment_mapping && P.p.enabled_refl_map));
}

hipError_t launch_render(const FrameParams& Pin, hipStream_t stream, hipEvent_t afterPasses, hipEvent_t* splitEvents, bool* splitRecorded,
    unsigned long long* wavefrontNodes, const SplitAux* aux, bool* shadowMarksRecorded)
{
    if (splitRecorded) *splitRecorded = false;
    if (shadowMarksRecorded) *shadowMarksRecorded = false;
    unsigned long long nodesEvaluated = 0;
    struct Report {
        unsigned long long* to;
        unsigned long long& from;
        ~Report() { if (to) *to = from; }
    } report { wavefrontNodes, nodesEvaluated };
    FrameParams P = Pin;
    const int rows = P.rowsLocal;
    if (rows <= 0 || P.width <= 0) return hipSuccess;
    const uint32_t tiles = uint32_t((P.width + TILE - 1) / TILE) * uint32_t((rows + TILE - 1) / TILE);
    const size_t lds = stack_bytes(P, WAVE);
    const bool prune = !P.f.recursive && !P.f.transparency;
    const bool glossy = P.f.glossy != 0;
    // the shadow pass packs a node index or primitive slot into STREAM_REF_BITS bits (inner nodes are fewer than primitives)
    if (P.streamGroup != 1 && P.streamGroup != 2) P.streamGroup = 2;
    if (P.streamGroup != 1 && uint32_t(P.scene.numPrims) > stream_ref_limit(P.streamGroup)) P.streamGroup = 1; // (entries of groups have fewer reference bits)
    const bool fits = uint32_t(P.scene.numPrims) <= stream_ref_limit(P.streamGroup) && P.scene.stackDepth < 64;
    const bool split = P.queue != nullptr && P.f.accel_structure && !P.scene.hasSpheres && direct_only(P) && fits;
    // bounce features as wavefront passes
    const bool wavefront = P.wfNodes != nullptr && P.f.accel_structure && !P.scene.hasSpheres && !direct_only(P) && fits;
    hipError_t e;
    bool wavefrontGaveUp = false;
    auto pass = [&](int sBegin, int sEnd) -> hipError_t {
        if (wavefront && !wavefrontGaveUp) {
            // (the ray counters as they stand are kept on the device, in case the pass has to be done again)
            unsigned long long* const backup = P.stats ? P.stats + 2 * STAT_COUNT : nullptr;
            if (backup) hipLaunchKernelGGL(k_stats, dim3(1), dim3(64), 0, stream, backup, P.stats, 0);
            const unsigned long long nodesBefore = nodesEvaluated;
            e = prune ? wavefront_pass<true>(P, sBegin, sEnd, lds, stream, nodesEvaluated) : wavefront_pass<false>(P, sBegin, sEnd, lds, stream, nodesEvaluated);
            if (e != hipErrorOutOfMemory) return e;
            // The call tree of a single pixel sample does not fit the node buffer (glossy_filter_size ^ ray_depth nodes at
            // worst): this pass, and what is left of the frame, goes to the per-lane kernel, which needs no memory per
            // call and gives the same frame.  What the abandoned pass has written is written again.
            wavefrontGaveUp = true;
            nodesEvaluated = nodesBefore;
            if (backup) hipLaunchKernelGGL(k_stats, dim3(1), dim3(64), 0, stream, P.stats, backup, 0);
        }
        P.sampleBegin = sBegin;
        P.sampleEnd = sEnd;
        P.logicalBlocks = tiles * uint32_t(sEnd - sBegin);
        if (split) {
            // the counter blocks of the pass (primary pass, lighting round A, lighting round B) are one allocation
            if ((e = hipMemsetAsync(P.primaryCounters, 0, (aux ? 3 : 1) * Q_COUNT * sizeof(unsigned long long), stream)) != hipSuccess) return e;
            const uint32_t waves = uint32_t((size_t(rows) * size_t(P.width) * size_t(sEnd - sBegin) + WAVE - 1) / WAVE);
            const dim3 gridDirect(waves < 256u * 32u ? waves : 256u * 32u); // at most every wave slot of the chip
            hipEvent_t* ev = (splitEvents && sBegin == 0 && sEnd == P.samplesPerPixel) ? splitEvents : nullptr;
            if (ev && splitRecorded) *splitRecorded = true;
            if (ev && aux && aux->shadowMarks && shadowMarksRecorded) *shadowMarksRecorded = true;
            return prune ? launch_split_t<true>(P, gridDirect, lds, stream, ev, aux) : launch_split_t<false>(P, gridDirect, lds, stream, ev, aux);
        }
        if (prune) return glossy ? launch_render_t<true, true>(P, lds, stream) : launch_render_t<true, false>(P, lds, stream);
        return glossy ? launch_render_t<false, true>(P, lds, stream) : launch_render_t<false, false>(P, lds, stream);
    };
    if (P.f.depth_of_field && !P.f.multiple_rays && P.samplesPerPixel > 1) {
        // the centre rays decide per pixel whether the lens rays are traced at all
        if ((e = pass(0, 1)) != hipSuccess) return e;
        int chunk = P.samplesPerPixel - 1;
        if (split) {
            // bound the hit queue, as below
            chunk = int(P.queueCapacity / (size_t(rows) * size_t(P.width)));
            if (chunk < 1) chunk = 1;
        }
        for (int sb = 1; sb < P.samplesPerPixel; sb += chunk)
            if ((e = pass(sb, sb + chunk < P.samplesPerPixel ? sb + chunk : P.samplesPerPixel)) != hipSuccess) return e;
    } else if (split) {
        // bound the hit queue: at most queueCapacity (pixel, sample) pairs per pass
        const size_t nPix = size_t(rows) * size_t(P.width);
        int chunk = int(P.queueCapacity / nPix);
        if (chunk < 1) chunk = 1;
        for (int sb = 0; sb < P.samplesPerPixel; sb += chunk)
            if ((e = pass(sb, sb + chunk < P.samplesPerPixel ? sb + chunk : P.samplesPerPixel)) != hipSuccess) return e;
    } else {
        if ((e = pass(0, P.samplesPerPixel)) != hipSuccess) return e;
    }
    if (afterPasses && (e = hipEventRecord(afterPasses, stream)) != hipSuccess) return e;
    if (P.samplesPerPixel > 1) {
        const size_t nPix = size_t(rows) * size_t(P.width);
        hipLaunchKernelGGL(k_resolve, dim3(uint32_t((nPix + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, stream, P);
        return hipGetLastError();
    }
    return hipSuccess;
}

hipError_t launch_intersect(const FrameParams& P, const rt_ray* rays, uint64_t n, rt_hit* hits, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const dim3 grid(uint32_t((n + BLOCK - 1) / BLOCK));
    const size_t lds = stack_bytes(P, BLOCK);
    const bool prune = !P.f.recursive && !P.f.transparency;
    if (P.scene.hasSpheres) {
        if (prune) hipLaunchKernelGGL((k_intersect<true, true>), grid, dim3(BLOCK), lds, stream, P, rays, n, hits);
        else hipLaunchKernelGGL((k_intersect<false, true>), grid, dim3(BLOCK), lds, stream, P, rays, n, hits);
    } else {
        if (prune) hipLaunchKernelGGL((k_intersect<true, false>), grid, dim3(BLOCK), lds, stream, P, rays, n, hits);
        else hipLaunchKernelGGL((k_intersect<false, false>), grid, dim3(BLOCK), lds, stream, P, rays, n, hits);
    }
    return hipGetLastError();
}

hipError_t launch_visibility(const FrameParams& P, const rt_ray* rays, uint64_t n, const float* samples, int nSamples, uint8_t* out, hipStream_t stream)
{
    const uint64_t total = n * uint64_t(nSamples);
    if (total == 0) return hipSuccess;
    const dim3 grid(uint32_t((total + BLOCK - 1) / BLOCK));
    const size_t lds = stack_bytes(P, BLOCK);
    const bool prune = !P.f.recursive && !P.f.transparency;
    if (P.scene.hasSpheres) {
        if (prune) hipLaunchKernelGGL((k_visibility<true, true>), grid, dim3(BLOCK), lds, stream, P, rays, n, samples, nSamples, out);
        else hipLaunchKernelGGL((k_visibility<false, true>), grid, dim3(BLOCK), lds, stream, P, rays, n, samples, nSamples, out);
    } else {
        if (prune) hipLaunchKernelGGL((k_visibility<true, false>), grid, dim3(BLOCK), lds, stream, P, rays, n, samples, nSamples, out);
        else hipLaunchKernelGGL((k_visibility<false, false>), grid, dim3(BLOCK), lds, stream, P, rays, n, samples, nSamples, out);
    }
    return hipGetLastError();
}

hipError_t launch_bloom(const float* src, float* dst, int width, int height, const BloomParams& bp, hipStream_t stream)
{
    const long n = long(width) * long(height);
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_bloom, dim3(uint32_t((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, stream, src, dst, width, height, bp);
    return hipGetLastError();
}

} // namespace rt
