// Multi-GPU assembly of a frame under the C ABI: the row-block shares of rt_render_blocks travel from the GPUs that
// rendered them straight into their places in the root GPU's frame, as grouped RCCL sends and receives over xGMI
// (replaces the shared-memory Screen the reference's row threads write into, reference src/render.cpp:86-89, and the
// per-camera threads of src/main.cpp:533-552 when the cameras are dealt to GPUs).
//
// RCCL is bound at run time (dlopen), so the library loads on machines without it and a process that already holds an
// RCCL (PyTorch ships its own) can name the one to use: RT_RCCL_LIBRARY, default librccl.so.1.
#include "rt_amd.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

rt_status rt_set_error(rt_status code, const std::string& msg); // (rt_api.cpp: the calling thread's error text)

struct rt_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

namespace {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) getUniqueId = nullptr;
    decltype(&ncclCommInitRank) commInitRank = nullptr;
    decltype(&ncclCommInitAll) commInitAll = nullptr;
    decltype(&ncclCommDestroy) commDestroy = nullptr;
    decltype(&ncclGroupStart) groupStart = nullptr;
    decltype(&ncclGroupEnd) groupEnd = nullptr;
    decltype(&ncclSend) send = nullptr;
    decltype(&ncclRecv) recv = nullptr;
    decltype(&ncclGetErrorString) errorString = nullptr;
    std::string problem;
};

const Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* named = std::getenv("RT_RCCL_LIBRARY");
        const char* candidates[] = { named, "librccl.so.1", "librccl.so" };
        for (const char* name : candidates) {
            if (!name || !*name) continue;
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) {
            r.problem = "cannot load RCCL (librccl.so.1; RT_RCCL_LIBRARY names another)";
            return;
        }
        auto sym = [&](const char* name) {
            void* p = dlsym(r.lib, name);
            if (!p && r.problem.empty()) r.problem = std::string("RCCL library lacks ") + name;
            return p;
        };
        r.getUniqueId = reinterpret_cast<decltype(r.getUniqueId)>(sym("ncclGetUniqueId"));
        r.commInitRank = reinterpret_cast<decltype(r.commInitRank)>(sym("ncclCommInitRank"));
        r.commInitAll = reinterpret_cast<decltype(r.commInitAll)>(sym("ncclCommInitAll"));
        r.commDestroy = reinterpret_cast<decltype(r.commDestroy)>(sym("ncclCommDestroy"));
        r.groupStart = reinterpret_cast<decltype(r.groupStart)>(sym("ncclGroupStart"));
        r.groupEnd = reinterpret_cast<decltype(r.groupEnd)>(sym("ncclGroupEnd"));
        r.send = reinterpret_cast<decltype(r.send)>(sym("ncclSend"));
        r.recv = reinterpret_cast<decltype(r.recv)>(sym("ncclRecv"));
        r.errorString = reinterpret_cast<decltype(r.errorString)>(sym("ncclGetErrorString"));
    });
    return r;
}

rt_status nccl_fail(const Rccl& r, ncclResult_t e, const char* what)
{
    return rt_set_error(RT_ERR_HIP, std::string(what) + ": " + (r.errorString ? r.errorString(e) : "RCCL error"));
}
#define RT_NCCL(call, what)                                     \
    do {                                                        \
        const ncclResult_t e_ = (call);                         \
        if (e_ != ncclSuccess) return nccl_fail(r, e_, what);   \
    } while (0)
#define RT_HIPC(call)                                                                              \
    do {                                                                                           \
        const hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) return rt_set_error(RT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

} // namespace

extern "C" {

rt_status rt_comm_unique_id(uint8_t* id)
{
    if (!id) return rt_set_error(RT_ERR_INVALID, "rt_comm_unique_id: null argument");
    const Rccl& r = rccl();
    if (!r.problem.empty()) return rt_set_error(RT_ERR_NO_DEVICE, r.problem);
    static_assert(RT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    ncclUniqueId u;
    RT_NCCL(r.getUniqueId(&u), "ncclGetUniqueId");
    std::memcpy(id, u.internal, RT_COMM_ID_BYTES);
    return RT_OK;
}

rt_status rt_comm_init_rank(const uint8_t* id, int32_t rank, int32_t world, rt_comm** out)
{
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return rt_set_error(RT_ERR_INVALID, "rt_comm_init_rank: bad argument");
    *out = nullptr;
    const Rccl& r = rccl();
    if (!r.problem.empty()) return rt_set_error(RT_ERR_NO_DEVICE, r.problem);
    ncclUniqueId u;
    std::memcpy(u.internal, id, RT_COMM_ID_BYTES);
    rt_comm* c = new (std::nothrow) rt_comm;
    if (!c) return rt_set_error(RT_ERR_INVALID, "out of host memory");
    c->rank = rank;
    c->world = world;
    if (hipGetDevice(&c->device) != hipSuccess) {
        delete c;
        return rt_set_error(RT_ERR_NO_DEVICE, "rt_comm_init_rank: no current HIP device");
    }
    const ncclResult_t e = r.commInitRank(&c->comm, world, u, rank);
    if (e != ncclSuccess) {
        delete c;
        return nccl_fail(r, e, "ncclCommInitRank");
    }
    *out = c;
    return RT_OK;
}

rt_status rt_comm_init_all(int32_t n, const int32_t* devices, rt_comm** out)
{
    if (n < 1 || !out) return rt_set_error(RT_ERR_INVALID, "rt_comm_init_all: bad argument");
    for (int i = 0; i < n; i++) out[i] = nullptr;
    const Rccl& r = rccl();
    if (!r.problem.empty()) return rt_set_error(RT_ERR_NO_DEVICE, r.problem);
    std::vector<ncclComm_t> comms(size_t(n), nullptr);
    std::vector<int> devs(size_t(n), 0);
    for (int i = 0; i < n; i++) devs[size_t(i)] = devices ? devices[i] : i;
    RT_NCCL(r.commInitAll(comms.data(), n, devs.data()), "ncclCommInitAll");
    for (int i = 0; i < n; i++) {
        rt_comm* c = new (std::nothrow) rt_comm;
        if (!c) return rt_set_error(RT_ERR_INVALID, "out of host memory");
        c->comm = comms[size_t(i)];
        c->rank = i;
        c->world = n;
        c->device = devs[size_t(i)];
        out[i] = c;
    }
    return RT_OK;
}

int32_t rt_comm_rank(const rt_comm* c) { return c ? c->rank : 0; }
int32_t rt_comm_world(const rt_comm* c) { return c ? c->world : 0; }

void rt_comm_destroy(rt_comm* c)
{
    if (!c) return;
    const Rccl& r = rccl();
    if (c->comm && r.commDestroy) (void)r.commDestroy(c->comm);
    delete c;
}

rt_status rt_gather_blocks(rt_comm* c, int32_t root, const float* share, float* frame, int32_t width, int32_t height, int32_t rowBlock, void* stream)
{
    if (!c || width <= 0 || height <= 0 || rowBlock <= 0 || root < 0 || root >= c->world) return rt_set_error(RT_ERR_INVALID, "rt_gather_blocks: bad argument");
    if (c->rank == root && !frame) return rt_set_error(RT_ERR_INVALID, "rt_gather_blocks: the root needs a frame");
    const Rccl& r = rccl();
    if (!r.problem.empty()) return rt_set_error(RT_ERR_NO_DEVICE, r.problem);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nBlocks = (height + rowBlock - 1) / rowBlock;
    const size_t rowFloats = size_t(width) * 3;
    auto block_rows = [&](int b) { return size_t((b + 1) * rowBlock < height ? rowBlock : height - b * rowBlock); };
    bool mine = false;
    for (int b = c->rank; b < nBlocks; b += c->world) mine = true;
    if (mine && !share) return rt_set_error(RT_ERR_INVALID, "rt_gather_blocks: null share");
    RT_HIPC(hipSetDevice(c->device));
    // Block b of the frame is contiguous both in the frame and in the share of rank b % world (where it is block
    // b / world): one send / receive per block, all of them in one group, the data landing where it belongs.
    RT_NCCL(r.groupStart(), "ncclGroupStart");
    ncclResult_t e = ncclSuccess;
    if (c->rank == root) {
        for (int b = 0; b < nBlocks && e == ncclSuccess; b++) {
            const int owner = b % c->world;
            if (owner == root) continue;
            e = r.recv(frame + size_t(b) * size_t(rowBlock) * rowFloats, block_rows(b) * rowFloats, ncclFloat, owner, c->comm, s);
        }
    } else {
        size_t at = 0;
        for (int b = c->rank; b < nBlocks && e == ncclSuccess; b += c->world) {
            e = r.send(share + at, block_rows(b) * rowFloats, ncclFloat, root, c->comm, s);
            at += block_rows(b) * rowFloats;
        }
    }
    const ncclResult_t g = r.groupEnd();
    if (e != ncclSuccess) return nccl_fail(r, e, "ncclSend / ncclRecv");
    if (g != ncclSuccess) return nccl_fail(r, g, "ncclGroupEnd");
    if (c->rank == root) {
        // the root's own blocks: device-to-device copies on the same stream
        size_t at = 0;
        for (int b = root; b < nBlocks; b += c->world) {
            const size_t count = block_rows(b) * rowFloats;
            RT_HIPC(hipMemcpyAsync(frame + size_t(b) * size_t(rowBlock) * rowFloats, share + at, count * sizeof(float), hipMemcpyDeviceToDevice, s));
            at += count;
        }
    }
    return RT_OK;
}

} // extern "C"
