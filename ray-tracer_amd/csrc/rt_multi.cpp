// rt_multi.cpp -- one frame over several devices of a node, driven from ONE host process through
// the C ABI (rt_multi_* in include/rt_mi355x.h).
//
// The reference fans a frame out over THREAD_SPAWN_COUNT host threads, one column strip each
// (TraceScene, Source/main.cpp:647-659).  Here the fan-out is over GPUs.  Every device owns an
// ordinary rt_context; the _device entry points only enqueue, so one host thread keeps all devices
// busy.  A frame is split either by sample slices (each device adds its samples of every pixel into
// its own full-frame sums; the sums are sum-REDUCED to the first device, which divides and packs)
// or by interleaved 8x8 tiles (each device divides and packs its own tiles into a tile-major block;
// the blocks are GATHERED to the first device, which writes them into the frame).  Sums are
// integers and every path is keyed on (seed, pixel, sample), so the frame is bit-identical to the
// single-device one for any number of devices and either split.
//
// The exchange runs over RCCL (ncclReduce; grouped ncclSend/ncclRecv), loaded with dlopen when a
// multi-device object asks for it, or as direct peer copies over xGMI (the first device reads the
// other devices' sums in an add kernel; blocks are hipMemcpyPeerAsync'ed into the gather buffer).
// Peer copies are also what several logical ranks on ONE device use.  The RCCL branch is the same code
// for one device as for several -- every rank, the first included, sends its block and reduces its sums
// -- so a one-device object over RCCL executes every RCCL call of this file (with itself as the only
// peer).  What has NOT run anywhere is RCCL between distinct devices: no multi-GPU machine was available
// to this build, which is why RT_EXCHANGE_AUTO takes peer copies whenever the first device can address
// all others, and RCCL only where it cannot.
//
// Frames are double-buffered: every rank has a compute and an exchange stream and two sets of
// buffers, so the exchange of frame k runs beside the tracing of frame k + 1.
#include "rt_device.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

using namespace rtamd;

namespace {

constexpr uint32_t kMaxRanks = 16; // also the add kernel's operand limit
// Buffer sets per rank: a call of g frames takes g of the first max(2, 2g), in rotation, so that two
// calls' worth of frames can be in flight.  Only the sets a multi-device object has used exist.
constexpr int kSlots = 2 * RT_MAX_JOBS;

// The handful of RCCL entry points used, resolved at run time: a process that never asks for a
// multi-device exchange never loads the library.
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;

    bool load(std::string &why)
    {
        if (handle)
            return true;
        const char *names[] = { "librccl.so.1", "librccl.so" };
        for (const char *n : names) // a copy the process already holds (PyTorch bundles one) wins
            if (!handle)
                handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const char *n : names)
            if (!handle)
                handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!handle) {
            why = std::string("cannot load librccl: ") + dlerror();
            return false;
        }
#define RT_SYM(field, name)                                                        \
    field = reinterpret_cast<decltype(field)>(dlsym(handle, name));                \
    if (!field) {                                                                  \
        why = std::string("librccl lacks ") + name;                                \
        return false;                                                              \
    }
        RT_SYM(CommInitAll, "ncclCommInitAll")
        RT_SYM(CommDestroy, "ncclCommDestroy")
        RT_SYM(Reduce, "ncclReduce")
        RT_SYM(Send, "ncclSend")
        RT_SYM(Recv, "ncclRecv")
        RT_SYM(GroupStart, "ncclGroupStart")
        RT_SYM(GroupEnd, "ncclGroupEnd")
        RT_SYM(GetErrorString, "ncclGetErrorString")
#undef RT_SYM
        return true;
    }
};

struct Rank {
    int device = 0;
    rt_context *ctx = nullptr;
    hipStream_t compute = nullptr, comm = nullptr;
    uint32_t *sums[kSlots] = {};
    uint8_t *block[kSlots] = {}; // ranks >= 1: this rank's tile block
    hipEvent_t traced[kSlots] = {};    // the rank's share of the slot's frame is in its buffers
    hipEvent_t exchanged[kSlots] = {}; // ... and has left them again
    bool exchanged_valid[kSlots] = {};
    ncclComm_t nccl = nullptr;
};

class DeviceGuard {
public:
    explicit DeviceGuard() { have_ = hipGetDevice(&prev_) == hipSuccess; }
    ~DeviceGuard()
    {
        if (have_)
            (void)hipSetDevice(prev_);
    }

private:
    int prev_ = 0;
    bool have_ = false;
};

} // namespace

struct rt_multi {
    std::vector<Rank> ranks;
    bool use_rccl = false;
    RcclApi rccl;
    size_t sums_bytes = 0;   // per slot and rank, as allocated
    size_t block_bytes = 0;  // per slot and rank, as allocated
    int n_slots = 0;         // buffer sets allocated
    uint8_t *blocks[kSlots] = {}; // first device: gather buffer, n ranks x block_bytes
    uint8_t *d_rgb = nullptr; // first device: frame of rt_multi_render
    size_t rgb_bytes = 0;
    uint64_t frames = 0;
    std::string error;
    std::recursive_mutex mtx;
};

namespace {

int fail(rt_multi *m, int status, const std::string &msg)
{
    if (m)
        m->error = msg;
    return status;
}

#define RTM_HIP(m, call)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail((m), RT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));     \
    } while (0)

#define RTM_NCCL(m, call)                                                                             \
    do {                                                                                              \
        ncclResult_t r_ = (call);                                                                     \
        if (r_ != ncclSuccess)                                                                        \
            return fail((m), RT_ERR_HIP, std::string(#call) + ": " + (m)->rccl.GetErrorString(r_));   \
    } while (0)

#define RTM_RT(m, rank, call)                                                                                    \
    do {                                                                                                         \
        int s_ = (call);                                                                                         \
        if (s_ != RT_OK)                                                                                         \
            return fail((m), s_, std::string(#call) + " on device " + std::to_string((rank).device) + ": " +    \
                                     rt_last_error((rank).ctx));                                                 \
    } while (0)

void free_buffers(rt_multi *m)
{
    for (Rank &r : m->ranks) {
        if (!r.ctx)
            continue;
        (void)hipSetDevice(r.device);
        for (int s = 0; s < kSlots; ++s) {
            if (r.sums[s]) (void)hipFree(r.sums[s]);
            if (r.block[s]) (void)hipFree(r.block[s]);
            r.sums[s] = nullptr;
            r.block[s] = nullptr;
        }
    }
    if (!m->ranks.empty() && m->ranks[0].ctx) {
        (void)hipSetDevice(m->ranks[0].device);
        for (int s = 0; s < kSlots; ++s) {
            if (m->blocks[s]) (void)hipFree(m->blocks[s]);
            m->blocks[s] = nullptr;
        }
    }
    m->sums_bytes = m->block_bytes = 0;
    m->n_slots = 0;
}

// `n_slots` frame buffer sets of every rank, grown when a larger frame or a longer call comes (after all
// enqueued work is done).
int ensure_buffers(rt_multi *m, size_t sums_bytes, size_t block_bytes, int n_slots)
{
    if (m->sums_bytes >= sums_bytes && m->block_bytes >= block_bytes && m->n_slots >= n_slots)
        return RT_OK;
    sums_bytes = std::max(sums_bytes, m->sums_bytes);
    block_bytes = std::max(block_bytes, m->block_bytes);
    n_slots = std::max(n_slots, m->n_slots);
    for (Rank &r : m->ranks) {
        RTM_HIP(m, hipSetDevice(r.device));
        RTM_HIP(m, hipDeviceSynchronize());
        for (bool &v : r.exchanged_valid)
            v = false;
    }
    free_buffers(m);
    const size_t n = m->ranks.size();
    for (size_t i = 0; i < n; ++i) {
        Rank &r = m->ranks[i];
        RTM_HIP(m, hipSetDevice(r.device));
        for (int s = 0; s < n_slots; ++s) {
            if (hipMalloc((void **)&r.sums[s], sums_bytes) != hipSuccess)
                return fail(m, RT_ERR_OUT_OF_MEMORY, "hipMalloc failed for a rank's sums");
            // (over RCCL the first rank sends its block like every other; with peer copies it packs
            // straight into the gather buffer)
            if ((i > 0 || m->use_rccl) && hipMalloc((void **)&r.block[s], block_bytes) != hipSuccess)
                return fail(m, RT_ERR_OUT_OF_MEMORY, "hipMalloc failed for a rank's tile block");
        }
    }
    RTM_HIP(m, hipSetDevice(m->ranks[0].device));
    for (int s = 0; s < n_slots; ++s) {
        if (hipMalloc((void **)&m->blocks[s], n * block_bytes) != hipSuccess)
            return fail(m, RT_ERR_OUT_OF_MEMORY, "hipMalloc failed for the gather buffer");
    }
    m->sums_bytes = sums_bytes;
    m->block_bytes = block_bytes;
    m->n_slots = n_slots;
    return RT_OK;
}

// Contiguous, as-even-as-possible split of [0, total) over n ranks.
void sample_slice(uint32_t total, uint32_t rank, uint32_t n, uint32_t &begin, uint32_t &count)
{
    const uint32_t base = total / n, extra = total % n;
    begin = rank * base + (rank < extra ? rank : extra);
    count = base + (rank < extra ? 1u : 0u);
}

} // namespace

extern "C" {

const char *rt_multi_last_error(const rt_multi *m) { return m ? m->error.c_str() : ""; }

uint32_t rt_multi_device_count(const rt_multi *m) { return m ? (uint32_t)m->ranks.size() : 0u; }

rt_context *rt_multi_context(rt_multi *m, uint32_t index)
{
    return (m && index < m->ranks.size()) ? m->ranks[index].ctx : nullptr;
}

int rt_multi_uses_rccl(const rt_multi *m) { return (m && m->use_rccl) ? 1 : 0; }

int rt_multi_destroy(rt_multi *m)
{
    if (!m)
        return RT_ERR_INVALID_ARGUMENT;
    DeviceGuard guard;
    for (Rank &r : m->ranks) {
        if (!r.ctx) // creation stopped before this rank: nothing of it exists
            continue;
        (void)hipSetDevice(r.device);
        (void)hipDeviceSynchronize();
    }
    for (Rank &r : m->ranks)
        if (r.nccl && m->rccl.CommDestroy)
            (void)m->rccl.CommDestroy(r.nccl);
    free_buffers(m);
    if (!m->ranks.empty() && m->ranks[0].ctx) {
        (void)hipSetDevice(m->ranks[0].device);
        if (m->d_rgb) (void)hipFree(m->d_rgb);
    }
    for (Rank &r : m->ranks) {
        if (!r.ctx)
            continue;
        (void)hipSetDevice(r.device);
        for (int s = 0; s < kSlots; ++s) {
            if (r.traced[s]) (void)hipEventDestroy(r.traced[s]);
            if (r.exchanged[s]) (void)hipEventDestroy(r.exchanged[s]);
        }
        if (r.compute) (void)hipStreamDestroy(r.compute);
        if (r.comm) (void)hipStreamDestroy(r.comm);
        if (r.ctx) (void)rt_destroy(r.ctx);
    }
    delete m;
    return RT_OK;
}

int rt_multi_create(const int *device_ids, uint32_t n_devices, int exchange, rt_multi **out)
{
    if (!out)
        return RT_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (!device_ids || n_devices == 0 || n_devices > kMaxRanks)
        return RT_ERR_INVALID_ARGUMENT;
    if (exchange != RT_EXCHANGE_AUTO && exchange != RT_EXCHANGE_RCCL && exchange != RT_EXCHANGE_PEER_COPY)
        return RT_ERR_INVALID_ARGUMENT;
    bool distinct = true;
    for (uint32_t i = 0; i < n_devices; ++i)
        for (uint32_t j = 0; j < i; ++j)
            if (device_ids[i] == device_ids[j])
                distinct = false;
    if (exchange == RT_EXCHANGE_RCCL && !distinct)
        return RT_ERR_INVALID_ARGUMENT; // RCCL takes one rank per device
    rt_multi *m = new (std::nothrow) rt_multi;
    if (!m)
        return RT_ERR_OUT_OF_MEMORY;
    DeviceGuard guard;
    m->ranks.resize(n_devices);
    int st = RT_OK;
    for (uint32_t i = 0; i < n_devices && st == RT_OK; ++i) {
        Rank &r = m->ranks[i];
        r.device = device_ids[i];
        st = rt_create(r.device, &r.ctx);
        if (st != RT_OK)
            break;
        if (hipSetDevice(r.device) != hipSuccess ||
            hipStreamCreateWithFlags(&r.compute, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&r.comm, hipStreamNonBlocking) != hipSuccess)
            st = RT_ERR_HIP;
        for (int s = 0; s < kSlots && st == RT_OK; ++s)
            if (hipEventCreateWithFlags(&r.traced[s], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&r.exchanged[s], hipEventDisableTiming) != hipSuccess)
                st = RT_ERR_HIP;
    }
    // the first device reads (peer copies) or receives (RCCL) the others' memory
    bool all_peers = true; // the first device can address every other rank's memory
    for (uint32_t i = 1; i < n_devices && st == RT_OK; ++i) {
        if (m->ranks[i].device == m->ranks[0].device)
            continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, m->ranks[0].device, m->ranks[i].device) == hipSuccess && can) {
            (void)hipSetDevice(m->ranks[0].device);
            const hipError_t e = hipDeviceEnablePeerAccess(m->ranks[i].device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
                st = RT_ERR_HIP;
            (void)hipGetLastError();
        } else {
            all_peers = false;
            if (exchange == RT_EXCHANGE_PEER_COPY)
                st = RT_ERR_HIP; // no peer path between the devices
        }
    }
    // AUTO: peer copies wherever the first device reaches all others (the transport every test of this
    // file has run over), RCCL only where it does not
    const bool want_rccl = exchange == RT_EXCHANGE_RCCL || (exchange == RT_EXCHANGE_AUTO && !all_peers);
    if (st == RT_OK && distinct && want_rccl) {
        std::string why;
        if (m->rccl.load(why)) {
            std::vector<ncclComm_t> comms(n_devices);
            const ncclResult_t r = m->rccl.CommInitAll(comms.data(), (int)n_devices, device_ids);
            if (r == ncclSuccess) {
                for (uint32_t i = 0; i < n_devices; ++i)
                    m->ranks[i].nccl = comms[i];
                m->use_rccl = true;
            } else if (exchange == RT_EXCHANGE_RCCL) {
                st = RT_ERR_HIP;
            }
        } else if (exchange == RT_EXCHANGE_RCCL) {
            st = RT_ERR_HIP;
        }
    }
    // without RCCL the first device's add kernel and copies read the other devices' buffers in place
    if (st == RT_OK && !m->use_rccl && !all_peers)
        st = RT_ERR_HIP;
    if (st != RT_OK) {
        rt_multi_destroy(m);
        return st;
    }
    *out = m;
    return RT_OK;
}

int rt_multi_set_scene(rt_multi *m, const rt_sphere *spheres, uint32_t n)
{
    if (!m)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::recursive_mutex> lock(m->mtx);
    // the first rank builds the hierarchy (on the host: seconds for very long lists); the others take it over
    for (Rank &r : m->ranks)
        RTM_RT(m, r, set_scene_sharing_hierarchy(r.ctx, spheres, n, &r == &m->ranks[0] ? nullptr : m->ranks[0].ctx));
    return RT_OK;
}

int rt_multi_set_option(rt_multi *m, int option, int64_t value)
{
    if (!m)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::recursive_mutex> lock(m->mtx);
    for (Rank &r : m->ranks)
        RTM_RT(m, r, rt_set_option(r.ctx, option, value));
    return RT_OK;
}

int rt_multi_synchronize(rt_multi *m)
{
    if (!m)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::recursive_mutex> lock(m->mtx);
    DeviceGuard guard;
    for (Rank &r : m->ranks) {
        RTM_HIP(m, hipSetDevice(r.device));
        RTM_HIP(m, hipStreamSynchronize(r.compute));
        RTM_HIP(m, hipStreamSynchronize(r.comm));
    }
    return RT_OK;
}

} // extern "C"

namespace {

int enqueue_frames(rt_multi *m, const rt_camera *cams, const rt_params *params, uint32_t n_frames, int partition,
                   uint8_t *const *d_rgb);

// Enqueues; on an error part of the call may be enqueued on some devices and not on others: wait for
// what is, and forget every slot's "left the buffers" event, so that the next frame starts from idle
// devices instead of on top of buffers something may still read.  (The error message is kept.)
int enqueue_or_drain(rt_multi *m, const rt_camera *cams, const rt_params *params, uint32_t n_frames, int partition,
                     uint8_t *const *d_rgb)
{
    const int st = enqueue_frames(m, cams, params, n_frames, partition, d_rgb);
    if (st != RT_OK) {
        const std::string why = m->error;
        for (Rank &r : m->ranks) {
            if (hipSetDevice(r.device) == hipSuccess) {
                (void)hipStreamSynchronize(r.compute);
                (void)hipStreamSynchronize(r.comm);
            }
            for (bool &v : r.exchanged_valid)
                v = false;
        }
        (void)hipGetLastError();
        m->error = why;
    }
    return st;
}

} // namespace

extern "C" {

int rt_multi_render_device(rt_multi *m, const rt_camera *cam, const rt_params *params, int partition, uint8_t *d_rgb)
{
    if (!m)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::recursive_mutex> lock(m->mtx);
    DeviceGuard guard;
    return enqueue_or_drain(m, cam, params, 1u, partition, &d_rgb);
}

int rt_multi_render_frames_device(rt_multi *m, const rt_camera *cams, const rt_params *params, uint32_t n_frames,
                                  int partition, uint8_t *const *d_rgb)
{
    if (!m)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::recursive_mutex> lock(m->mtx);
    if (n_frames == 0u)
        return RT_OK;
    if (n_frames > RT_MAX_JOBS)
        return fail(m, RT_ERR_INVALID_ARGUMENT, "more than RT_MAX_JOBS frames in one call");
    DeviceGuard guard;
    return enqueue_or_drain(m, cams, params, n_frames, partition, d_rgb);
}

} // extern "C"

namespace {

// The frames of one call: every rank traces its shares of all of them in ONE launch (the launch's end,
// paths running out with the work queue dry, is paid once instead of per frame), then the frames are
// exchanged and finished one by one.
int enqueue_frames(rt_multi *m, const rt_camera *cams, const rt_params *params, uint32_t n_frames, int partition,
                   uint8_t *const *d_rgb)
{
    if (!cams || !params || !d_rgb)
        return fail(m, RT_ERR_INVALID_ARGUMENT, "null camera, params or d_rgb");
    if (partition != RT_PARTITION_SAMPLES && partition != RT_PARTITION_TILES)
        return fail(m, RT_ERR_INVALID_ARGUMENT, "partition must be RT_PARTITION_SAMPLES or RT_PARTITION_TILES");
    const uint32_t n = (uint32_t)m->ranks.size();
    size_t sums_bytes = 0u, bb_max = 0u;
    for (uint32_t f = 0; f < n_frames; ++f) {
        const rt_params &p = params[f];
        if (!d_rgb[f])
            return fail(m, RT_ERR_INVALID_ARGUMENT, "null camera, params or d_rgb");
        if (p.sample_begin || p.sample_count || p.tile_first || p.tile_stride)
            return fail(m, RT_ERR_INVALID_ARGUMENT, "rt_multi_render splits the frame itself: sample_* and tile_* must be 0");
        if (p.width == 0 || p.height == 0 || p.spp == 0 || (uint64_t)p.width * p.height > (1ull << 30))
            return fail(m, RT_ERR_INVALID_ARGUMENT, "bad frame size or spp");
        sums_bytes = std::max(sums_bytes, (size_t)p.width * p.height * 3u * sizeof(uint32_t));
        bb_max = std::max(bb_max, (size_t)rt_tile_block_bytes(p.width, p.height, 0u, n)); // every rank's block padded to the first's
    }
    const int ring = std::max(2, 2 * (int)n_frames);
    int st = ensure_buffers(m, sums_bytes, bb_max, ring);
    if (st != RT_OK)
        return st;
    int slots[RT_MAX_JOBS];
    for (uint32_t f = 0; f < n_frames; ++f)
        slots[f] = (int)((m->frames + f) % (uint64_t)ring);
    Rank &root = m->ranks[0];
    const bool tiles = partition == RT_PARTITION_TILES;

    // ---- every rank: zero, trace its shares, (tiles) divide and pack -- on its compute stream
    for (uint32_t i = 0; i < n; ++i) {
        Rank &r = m->ranks[i];
        RTM_HIP(m, hipSetDevice(r.device));
        rt_job jobs[RT_MAX_JOBS];
        uint32_t n_jobs = 0u;
        for (uint32_t f = 0; f < n_frames; ++f) {
            const int slot = slots[f];
            // the slot's previous frame must have left these buffers: with RCCL the rank's own exchange
            // stream knows, with peer copies the first device's (it pulled them)
            Rank &freed_by = m->use_rccl ? r : root;
            if (freed_by.exchanged_valid[slot])
                RTM_HIP(m, hipStreamWaitEvent(r.compute, freed_by.exchanged[slot], 0));
            RTM_HIP(m, hipMemsetAsync(r.sums[slot], 0, (size_t)params[f].width * params[f].height * 3u * sizeof(uint32_t),
                                      r.compute));
            rt_params p = params[f];
            if (tiles) {
                p.tile_first = i;
                p.tile_stride = n;
            } else {
                sample_slice(p.spp, i, n, p.sample_begin, p.sample_count);
                if (p.sample_count == 0u)
                    continue; // more ranks than samples: nothing to add
                if (n == 1u)
                    p.sample_begin = p.sample_count = 0u;
            }
            rt_job &j = jobs[n_jobs++];
            memset(&j, 0, sizeof j);
            j.cam = cams[f];
            j.params = p;
            j.xsize = (int)p.width;
            j.ysize = (int)p.height;
            j.d_sums = r.sums[slot];
        }
        if (n_jobs == 1u) // (one job: the entry point that also takes calls too large for one launch)
            RTM_RT(m, r, rt_accumulate_device(r.ctx, &jobs[0].cam, &jobs[0].params, 0, 0, jobs[0].xsize, jobs[0].ysize,
                                              jobs[0].d_sums, r.compute));
        else if (n_jobs > 1u)
            RTM_RT(m, r, rt_accumulate_batch_device(r.ctx, jobs, n_jobs, r.compute));
        for (uint32_t f = 0; f < n_frames; ++f) {
            const int slot = slots[f];
            const rt_params &p = params[f];
            if (tiles) {
                uint8_t *dst = (i == 0 && !m->use_rccl) ? m->blocks[slot] : r.block[slot];
                RTM_RT(m, r, rt_pack_tiles_device(r.ctx, r.sums[slot], p.width, p.height, p.spp, p.flags & RT_FLAG_GAMMA2,
                                                  i, n, dst, r.compute));
            }
            RTM_HIP(m, hipEventRecord(r.traced[slot], r.compute));
        }
    }

    // ---- frame by frame: the exchange, on the exchange streams, and the first device's last step
    for (uint32_t f = 0; f < n_frames; ++f) {
        const int slot = slots[f];
        const rt_params &p = params[f];
        const uint32_t W = p.width, H = p.height;
        const size_t count = (size_t)W * H * 3u;
        const size_t bb = (size_t)rt_tile_block_bytes(W, H, 0u, n);
        for (uint32_t i = 0; i < n; ++i) {
            Rank &r = m->ranks[i];
            RTM_HIP(m, hipSetDevice(r.device));
            RTM_HIP(m, hipStreamWaitEvent(r.comm, r.traced[slot], 0));
        }
        if (m->use_rccl) {
            // One group for all ranks' calls (one thread drives every communicator).  Sample slices: every
            // rank's sums reduced in place into the first rank's.  Tiles: every rank, the first included,
            // sends its block to the first, which receives block j at offset j of the gather buffer -- with
            // one device this is the first rank sending to itself, the same calls as with eight.
            RTM_NCCL(m, m->rccl.GroupStart());
            for (uint32_t i = 0; i < n; ++i) {
                Rank &r = m->ranks[i];
                ncclResult_t e = ncclSuccess;
                if (!tiles) {
                    e = m->rccl.Reduce(r.sums[slot], r.sums[slot], count, ncclUint32, ncclSum, 0, r.nccl, r.comm);
                } else {
                    if (i == 0)
                        for (uint32_t j = 0; j < n && e == ncclSuccess; ++j)
                            e = m->rccl.Recv(m->blocks[slot] + (size_t)j * bb, bb, ncclUint8, (int)j, r.nccl, r.comm);
                    if (e == ncclSuccess)
                        e = m->rccl.Send(r.block[slot], bb, ncclUint8, 0, r.nccl, r.comm);
                }
                if (e != ncclSuccess) {
                    (void)m->rccl.GroupEnd();
                    return fail(m, RT_ERR_HIP, std::string("RCCL exchange: ") + m->rccl.GetErrorString(e));
                }
            }
            RTM_NCCL(m, m->rccl.GroupEnd());
        } else if (n > 1u) {
            // peer copies: the first device waits for every rank's share, then pulls it
            RTM_HIP(m, hipSetDevice(root.device));
            for (uint32_t i = 1; i < n; ++i)
                RTM_HIP(m, hipStreamWaitEvent(root.comm, m->ranks[i].traced[slot], 0));
            if (tiles) {
                for (uint32_t i = 1; i < n; ++i) {
                    uint8_t *dst = m->blocks[slot] + (size_t)i * bb;
                    if (m->ranks[i].device == root.device)
                        RTM_HIP(m, hipMemcpyAsync(dst, m->ranks[i].block[slot], bb, hipMemcpyDeviceToDevice, root.comm));
                    else
                        RTM_HIP(m, hipMemcpyPeerAsync(dst, root.device, m->ranks[i].block[slot], m->ranks[i].device, bb,
                                                      root.comm));
                }
            } else {
                AddArgs a;
                memset(&a, 0, sizeof a);
                a.dst = root.sums[slot];
                a.count = count;
                a.n_src = n - 1u;
                for (uint32_t i = 1; i < n; ++i)
                    a.src[i - 1u] = m->ranks[i].sums[slot];
                RTM_HIP(m, launch_add_sums(a, root.comm));
            }
        }

        // ---- the first device: FinalColor / SPP and PutPixel (Source/main.cpp:638-642)
        RTM_HIP(m, hipSetDevice(root.device));
        if (tiles)
            RTM_RT(m, root, rt_unpack_tiles_device(root.ctx, m->blocks[slot], n, bb, W, H, d_rgb[f], root.comm));
        else
            RTM_RT(m, root, rt_resolve_device(root.ctx, root.sums[slot], W, H, p.spp, p.flags & RT_FLAG_GAMMA2,
                                              0, 0, (int)W, (int)H, d_rgb[f], root.comm));
        RTM_HIP(m, hipEventRecord(root.exchanged[slot], root.comm));
        root.exchanged_valid[slot] = true;
        for (uint32_t i = 1; i < n && m->use_rccl; ++i) {
            Rank &r = m->ranks[i];
            RTM_HIP(m, hipSetDevice(r.device));
            RTM_HIP(m, hipEventRecord(r.exchanged[slot], r.comm));
            r.exchanged_valid[slot] = true;
        }
    }
    m->frames += n_frames;
    return RT_OK;
}

} // namespace

extern "C" {

int rt_multi_render(rt_multi *m, const rt_camera *cam, const rt_params *params, int partition, uint8_t *rgb)
{
    if (!m)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::recursive_mutex> lock(m->mtx);
    if (!rgb || !params)
        return fail(m, RT_ERR_INVALID_ARGUMENT, "null rgb or params");
    if (params->width == 0 || params->height == 0 || (uint64_t)params->width * params->height > (1ull << 30))
        return fail(m, RT_ERR_INVALID_ARGUMENT, "bad frame size");
    DeviceGuard guard;
    Rank &root = m->ranks[0];
    const size_t bytes = (size_t)params->width * params->height * 3u;
    RTM_HIP(m, hipSetDevice(root.device));
    if (m->rgb_bytes < bytes) {
        RTM_HIP(m, hipDeviceSynchronize());
        if (m->d_rgb) (void)hipFree(m->d_rgb);
        m->d_rgb = nullptr;
        m->rgb_bytes = 0;
        if (hipMalloc((void **)&m->d_rgb, bytes) != hipSuccess)
            return fail(m, RT_ERR_OUT_OF_MEMORY, "hipMalloc failed for the frame");
        m->rgb_bytes = bytes;
    }
    const int st = rt_multi_render_device(m, cam, params, partition, m->d_rgb);
    if (st != RT_OK)
        return st;
    RTM_HIP(m, hipSetDevice(root.device));
    RTM_HIP(m, hipMemcpyAsync(rgb, m->d_rgb, bytes, hipMemcpyDeviceToHost, root.comm));
    RTM_HIP(m, hipStreamSynchronize(root.comm));
    return RT_OK;
}

int rt_multi_get_stats(rt_multi *m, rt_stats *stats)
{
    if (!m)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::recursive_mutex> lock(m->mtx);
    if (!stats)
        return fail(m, RT_ERR_INVALID_ARGUMENT, "null stats");
    memset(stats, 0, sizeof *stats);
    for (Rank &r : m->ranks) {
        rt_stats s;
        RTM_RT(m, r, rt_get_stats(r.ctx, &s));
        stats->rays += s.rays;
        stats->sphere_tests += s.sphere_tests;
        stats->wave_passes += s.wave_passes;
        stats->node_visits += s.node_visits;
        stats->exact_tests += s.exact_tests;
        stats->root_evaluations += s.root_evaluations;
    }
    return RT_OK;
}

int rt_multi_reset_stats(rt_multi *m)
{
    if (!m)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::recursive_mutex> lock(m->mtx);
    for (Rank &r : m->ranks)
        RTM_RT(m, r, rt_reset_stats(r.ctx));
    return RT_OK;
}

} // extern "C"
