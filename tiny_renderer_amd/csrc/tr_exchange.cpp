// tr_exchange.cpp -- the hand-tuned alternative to the RCCL all-gather of the frame buffer
// ("7 concurrent peer copies ... measure both").
//
// One process per GPU.  Every rank owns up to 64 full-size frame buffers ("slots": one device
// allocation) and exports them (hipIpcGetMemHandle); after the handles have been exchanged -- by
// whatever rendezvous the host has: torch.distributed in bench.py -- every rank holds all its
// peers' slots mapped.  An all-gather of slot b is then, on rank r:
//
//   1. tell every peer "my slot b is open for generation g" (a 4-byte store into the peer's flag block)
//   2. per peer p, on a stream of its own: wait for p's "open", copy r's band into p's slot b with the
//      DMA engines (hipMemcpyAsync device-to-device over xGMI: no compute unit is involved, so the
//      copies run beside the next frame's tile kernel, which an RCCL kernel cannot: a machine-filling
//      kernel starves another queue's workgroups, profiles/r02_notes.md), then store "arrived, g"
//   3. on the caller's stream: join the copy streams, wait until every peer's "arrived" says g
//
// That is the PUSH form, for callers whose ranges differ from call to call.  A copy engine cannot be predicated on
// the error word: after a wait has timed out the push still lands in a slot its owner never opened.  Once the ranks'
// byte ranges are known to everybody (tr_exchange_set_ranges: tr_band_rows' bands) the dense exchange therefore PULLS:
// 1. "my band of generation g is ready" to every peer; 2. per peer p: wait for p's "ready", copy p's band out of p's
// mapped slot into the OWN slot, tell p "pulled, g"; 3. wait until every peer has pulled.  Nothing but 4-byte flags
// is ever written into another rank's memory: a peer that is late or gone costs this rank its own frame (error word)
// and nobody else anything.  The sparse tile push (k_push_tiles) is a kernel and is predicated.
//
// Flags are generation counters in uncached device memory (hipDeviceMallocUncached), written across
// GPUs by one-lane kernels with system-scope stores and awaited by one-lane-per-flag kernels that spin
// with s_sleep and give up after ten seconds (TR_E_EXCHANGE instead of a hung GPU).  All n - 1 band
// copies leave at once: xGMI is point to point, each of the 7 links carries one band.
//
// Second backend, TR_EXCHANGE_RCCL ("ncclAllGather(sendbuf = band, recvbuf = full frame) in
// place"; north_star's RCCL all-gather for a host that is not Python): the same calls, but connect() builds an RCCL
// communicator -- rank 0's record carries the ncclUniqueId -- and all_gather is one in-place ncclAllGather on the
// caller's stream.  librccl is opened with dlopen when such an exchange is created: a single-GPU user of the
// library needs no RCCL.
//
// Not in the reference (single process, single thread); replaces nothing of it.  Selected with
// `bench.py --exchange peer`; covered on one GPU by two processes sharing the device
// (tests/test_gpu_parity.py::test_peer_exchange_two_processes_one_gpu) -- cross-GPU runs are the
// driver's (8-GPU node).
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "tiny_renderer.h"
#include "tr_error.h"
#include "tr_kernels.h"

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return tr::fail(TR_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

namespace {
constexpr uint32_t MAX_SLOTS = 64;
constexpr uint32_t MAX_RANKS = 64;

// what a rank publishes
struct Blob {
    uint32_t magic, rank, n_ranks, n_slots;
    uint64_t frame_bytes;
    union {
        struct {
            hipIpcMemHandle_t frames;  // all slots: one allocation, slot b at b * slot_stride
            hipIpcMemHandle_t flags;
        };
        ncclUniqueId rccl_id;  // TR_EXCHANGE_RCCL: rank 0's record carries the communicator's id
    };
};

// librccl, resolved when the first RCCL exchange is created
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
} g_rccl;

int load_rccl()
{
    if (g_rccl.lib) return TR_OK;
    void *lib = nullptr;
    for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" })
        if ((lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!lib) return tr::fail(TR_E_RCCL, std::string("librccl could not be loaded: ") + dlerror());
    Rccl r;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(lib, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(lib, "ncclCommDestroy");
    r.CommGetAsyncError = (decltype(r.CommGetAsyncError))dlsym(lib, "ncclCommGetAsyncError");
    r.AllGather = (decltype(r.AllGather))dlsym(lib, "ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.CommGetAsyncError || !r.AllGather || !r.GetErrorString)
        return tr::fail(TR_E_RCCL, "librccl lacks an expected symbol");
    r.lib = lib;
    g_rccl = r;
    return TR_OK;
}

#define RCCL_TRY(expr)                                                                                      \
    do {                                                                                                    \
        ncclResult_t r_ = (expr);                                                                           \
        if (r_ != ncclSuccess) return tr::fail(TR_E_RCCL, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); \
    } while (0)
static_assert(sizeof(Blob) <= TR_EXCHANGE_HANDLE_BYTES, "blob must fit the published size");

// flag block of one rank (uncached): [kind][slot][peer] generation counters, plus an error word
struct FlagIndex {
    static uint32_t open(uint32_t slot, uint32_t peer) { return (0 * MAX_SLOTS + slot) * MAX_RANKS + peer; }
    static uint32_t arrived(uint32_t slot, uint32_t peer) { return (1 * MAX_SLOTS + slot) * MAX_RANKS + peer; }
    static uint32_t error() { return 2 * MAX_SLOTS * MAX_RANKS; }
    static uint32_t words() { return 2 * MAX_SLOTS * MAX_RANKS + 16; }
};
}  // namespace

struct tr_exchange {
    int device = 0;
    uint32_t n_ranks = 0, rank = 0, n_slots = 0;
    size_t frame_bytes = 0, slot_stride = 0;
    uint8_t *frames = nullptr;              // the slots' allocation
    uint8_t *frame[MAX_SLOTS] = {};
    std::vector<uint8_t *> peer_base;       // [rank]: the peer's allocation mapped here
    std::vector<size_t> range_offset, range_bytes;  // tr_exchange_set_ranges: every rank's part of a frame (pull form)
    uint32_t *flags = nullptr;              // this rank's block (peers write into it)
    std::vector<uint8_t *> peer_frame[MAX_SLOTS];  // [slot][rank], own entry = own pointer
    std::vector<uint32_t *> peer_flags;            // [rank]
    std::vector<hipStream_t> copy_stream;          // [rank]
    std::vector<hipEvent_t> copy_done;             // [rank]
    hipEvent_t fork = nullptr;
    uint32_t generation[MAX_SLOTS] = {};
    uint64_t bytes_sent = 0;  // bytes this rank has pushed to its peers so far (dense calls)
    // sparse calls (tr_exchange_all_gather_tiles): per slot, for each peer, which tiles of THIS rank's band hold zeros in
    // the peer's copy -- [peer][tile], 0 = unknown -- and the device's count of the bytes those calls pushed
    uint32_t *remote_clean[MAX_SLOTS] = {};
    uint32_t remote_tiles[MAX_SLOTS] = {};
    unsigned long long *d_tile_bytes = nullptr;
    uint32_t **d_wait_list = nullptr;  // device array of flag pointers for the arrival wait: [slot][peer]
    uint32_t **d_open_list = nullptr;  // ... and of the peers' "open" flags this rank stores into: [slot][peer]
    bool connected = false;
    int backend = TR_EXCHANGE_PEER;
    ncclComm_t comm = nullptr;     // TR_EXCHANGE_RCCL
    ncclUniqueId rccl_id = {};     // ... rank 0's, published with its record
    uint64_t timeout_ticks = 1000000000ull;  // 10 s of the 100 MHz wall clock; TR_EXCHANGE_TIMEOUT_MS overrides (tests)
};

extern "C" {

int tr_exchange_create(int device, uint32_t n_ranks, uint32_t rank, uint32_t n_slots, size_t frame_bytes, tr_exchange **out)
{
    return tr_exchange_create_backend(device, n_ranks, rank, n_slots, frame_bytes, TR_EXCHANGE_PEER, out);
}

int tr_exchange_create_backend(int device, uint32_t n_ranks, uint32_t rank, uint32_t n_slots, size_t frame_bytes, int backend,
                               tr_exchange **out)
{
    if (!out) return tr::fail(TR_E_INVALID, "null out pointer");
    *out = nullptr;
    if (backend != TR_EXCHANGE_PEER && backend != TR_EXCHANGE_RCCL) return tr::fail(TR_E_INVALID, "unknown exchange backend");
    if (backend == TR_EXCHANGE_RCCL) {
        int st = load_rccl();
        if (st != TR_OK) return st;
    }
    if (n_ranks == 0 || n_ranks > MAX_RANKS || rank >= n_ranks || n_slots == 0 || n_slots > MAX_SLOTS || frame_bytes == 0)
        return tr::fail(TR_E_INVALID, "tr_exchange_create: need rank < n_ranks <= 64, 1..64 slots, a non-empty frame");
    if (device >= 0) HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipGetDevice(&device));
    tr_exchange *x = new tr_exchange();
    x->device = device;
    x->n_ranks = n_ranks;
    x->rank = rank;
    x->n_slots = n_slots;
    x->frame_bytes = frame_bytes;
    x->backend = backend;
    if (const char *ms = getenv("TR_EXCHANGE_TIMEOUT_MS")) {
        const long v = atol(ms);
        if (v > 0) x->timeout_ticks = (uint64_t)v * 100000ull;
    }
    x->slot_stride = (frame_bytes + 4095u) & ~(size_t)4095u;
    hipError_t e = hipMalloc((void **)&x->frames, x->slot_stride * n_slots);
    if (e == hipSuccess) e = hipMemset(x->frames, 0, x->slot_stride * n_slots);
    for (uint32_t b = 0; b < n_slots && e == hipSuccess; b++) x->frame[b] = x->frames + (size_t)b * x->slot_stride;
    if (e == hipSuccess) e = hipExtMallocWithFlags((void **)&x->flags, FlagIndex::words() * 4, hipDeviceMallocUncached);
    if (e == hipSuccess) e = hipMemset(x->flags, 0, FlagIndex::words() * 4);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&x->fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        tr_exchange_destroy(x);
        return tr::fail(TR_E_HIP, std::string("tr_exchange_create: ") + hipGetErrorString(e));
    }
    if (backend == TR_EXCHANGE_RCCL && rank == 0u) {
        ncclResult_t r = g_rccl.GetUniqueId(&x->rccl_id);
        if (r != ncclSuccess) {
            tr_exchange_destroy(x);
            return tr::fail(TR_E_RCCL, std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r));
        }
    }
    *out = x;
    return TR_OK;
}

void *tr_exchange_frame(tr_exchange *x, uint32_t slot) { return (x && slot < x->n_slots) ? x->frame[slot] : nullptr; }

int tr_exchange_export(tr_exchange *x, void *blob)
{
    if (!x || !blob) return tr::fail(TR_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(x->device));
    Blob b;
    memset(&b, 0, sizeof b);
    b.magic = 0x54524558u;  // "TREX"
    b.rank = x->rank;
    b.n_ranks = x->n_ranks;
    b.n_slots = x->n_slots;
    b.frame_bytes = x->frame_bytes;
    if (x->backend == TR_EXCHANGE_RCCL) {
        b.magic = 0x54525243u;  // "TRRC"
        b.rccl_id = x->rccl_id;  // (meaningful in rank 0's record only)
    } else {
        HIP_TRY(hipIpcGetMemHandle(&b.frames, x->frames));
        HIP_TRY(hipIpcGetMemHandle(&b.flags, x->flags));
    }
    memset(blob, 0, TR_EXCHANGE_HANDLE_BYTES);
    memcpy(blob, &b, sizeof b);
    return TR_OK;
}

int tr_exchange_connect(tr_exchange *x, const void *blobs)
{
    if (!x || !blobs) return tr::fail(TR_E_INVALID, "null argument");
    if (x->connected) return tr::fail(TR_E_INVALID, "tr_exchange_connect: already connected");
    HIP_TRY(hipSetDevice(x->device));
    if (x->backend == TR_EXCHANGE_RCCL) {
        // every rank's record must describe the same exchange; rank 0's carries the id.  Collective: all ranks
        // are inside ncclCommInitRank together.
        ncclUniqueId id = {};
        for (uint32_t p = 0; p < x->n_ranks; p++) {
            Blob b;
            memcpy(&b, (const uint8_t *)blobs + (size_t)p * TR_EXCHANGE_HANDLE_BYTES, sizeof b);
            if (b.magic != 0x54525243u || b.rank != p || b.n_ranks != x->n_ranks || b.n_slots != x->n_slots ||
                b.frame_bytes != x->frame_bytes)
                return tr::fail(TR_E_INVALID, "tr_exchange_connect: the peers' records do not describe the same RCCL exchange");
            if (p == 0u) id = b.rccl_id;
        }
        RCCL_TRY(g_rccl.CommInitRank(&x->comm, (int)x->n_ranks, id, (int)x->rank));
        x->connected = true;
        return TR_OK;
    }
    for (uint32_t s = 0; s < x->n_slots; s++) x->peer_frame[s].assign(x->n_ranks, nullptr);
    x->peer_flags.assign(x->n_ranks, nullptr);
    x->peer_base.assign(x->n_ranks, nullptr);
    x->copy_stream.assign(x->n_ranks, nullptr);
    x->copy_done.assign(x->n_ranks, nullptr);
    for (uint32_t p = 0; p < x->n_ranks; p++) {
        Blob b;
        memcpy(&b, (const uint8_t *)blobs + (size_t)p * TR_EXCHANGE_HANDLE_BYTES, sizeof b);
        if (b.magic != 0x54524558u || b.rank != p || b.n_ranks != x->n_ranks || b.n_slots != x->n_slots ||
            b.frame_bytes != x->frame_bytes)
            return tr::fail(TR_E_INVALID, "tr_exchange_connect: the peers' records do not describe the same exchange");
        if (p == x->rank) {
            for (uint32_t s = 0; s < x->n_slots; s++) x->peer_frame[s][p] = x->frame[s];
            x->peer_flags[p] = x->flags;
            continue;
        }
        HIP_TRY(hipIpcOpenMemHandle((void **)&x->peer_base[p], b.frames, hipIpcMemLazyEnablePeerAccess));
        for (uint32_t s = 0; s < x->n_slots; s++) x->peer_frame[s][p] = x->peer_base[p] + (size_t)s * x->slot_stride;
        HIP_TRY(hipIpcOpenMemHandle((void **)&x->peer_flags[p], b.flags, hipIpcMemLazyEnablePeerAccess));
        HIP_TRY(hipStreamCreateWithFlags(&x->copy_stream[p], hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&x->copy_done[p], hipEventDisableTiming));
    }
    // the arrival wait's list of flag addresses, per slot: [slot][peer] -> &own flags[arrived(slot, peer)]
    std::vector<uint32_t *> list((size_t)MAX_SLOTS * x->n_ranks, nullptr);
    for (uint32_t s = 0; s < x->n_slots; s++)
        for (uint32_t p = 0; p < x->n_ranks; p++) list[(size_t)s * x->n_ranks + p] = x->flags + FlagIndex::arrived(s, p);
    HIP_TRY(hipMalloc((void **)&x->d_wait_list, list.size() * sizeof(uint32_t *)));
    HIP_TRY(hipMemcpy(x->d_wait_list, list.data(), list.size() * sizeof(uint32_t *), hipMemcpyHostToDevice));
    for (uint32_t s = 0; s < x->n_slots; s++)
        for (uint32_t p = 0; p < x->n_ranks; p++)
            list[(size_t)s * x->n_ranks + p] = x->peer_flags[p] + FlagIndex::open(s, x->rank);
    HIP_TRY(hipMalloc((void **)&x->d_open_list, list.size() * sizeof(uint32_t *)));
    HIP_TRY(hipMemcpy(x->d_open_list, list.data(), list.size() * sizeof(uint32_t *), hipMemcpyHostToDevice));
    x->connected = true;
    return TR_OK;
}

namespace {
// The peer transport's exchange of slot `slot`: this rank's part goes to every peer either as the byte range
// [offset, offset + bytes) through the DMA engines (tiles == nullptr) or tile by tile through k_push_tiles.
int peer_all_gather(tr_exchange *x, uint32_t slot, size_t offset, size_t bytes, const tr_band_tiles *tiles, hipStream_t stream)
{
    const uint32_t g = x->generation[slot] + 1u;
    const uint32_t r = x->rank;
    // dense, and everybody's range is known: pull (nothing but flags is written into a peer's memory)
    const bool pull = !tiles && !x->range_bytes.empty();
    if (pull && (offset != x->range_offset[r] || bytes != x->range_bytes[r]))
        return tr::fail(TR_E_INVALID, "tr_exchange_all_gather: this rank's range differs from the one declared with tr_exchange_set_ranges");
    tr::DevFrame frame = {};
    if (tiles) {
        frame.width = tiles->width;
        frame.height = tiles->height;
        frame.band_y0 = tiles->band_y0;
        frame.band_y1 = tiles->band_y1;
        frame.ntx = tiles->tiles_x;
        frame.nty = tiles->tiles_y;
        frame.ty_base = tiles->first_tile_row;
        const uint32_t n_tiles = frame.ntx * frame.nty;
        if (x->remote_tiles[slot] != n_tiles) {  // first sparse call on the slot (or another grid): nothing is known
            if (x->remote_clean[slot]) {
                HIP_TRY(hipDeviceSynchronize());
                HIP_TRY(hipFree(x->remote_clean[slot]));
                x->remote_clean[slot] = nullptr;
            }
            HIP_TRY(hipMalloc((void **)&x->remote_clean[slot], (size_t)x->n_ranks * (n_tiles ? n_tiles : 1u) * 4u));
            // a slot that has never been exchanged holds zeros on every rank (tr_exchange_create fills the slots, and
            // nobody but this rank writes its band of the peers' copies): the record starts as "zeros there"; after any
            // exchange on the slot, with whatever grid, nothing is known
            HIP_TRY(hipMemsetAsync(x->remote_clean[slot], x->generation[slot] == 0u ? 1 : 0,
                                   (size_t)x->n_ranks * (n_tiles ? n_tiles : 1u) * 4u, stream));
            x->remote_tiles[slot] = n_tiles;
        }
        if (!x->d_tile_bytes) {
            HIP_TRY(hipMalloc((void **)&x->d_tile_bytes, sizeof(unsigned long long)));
            HIP_TRY(hipMemsetAsync(x->d_tile_bytes, 0, sizeof(unsigned long long), stream));
        }
    } else if (x->remote_clean[slot]) {
        // a dense exchange rewrites the peers' copies of the band whatever they held: the record starts over
        HIP_TRY(hipMemsetAsync(x->remote_clean[slot], 0, (size_t)x->n_ranks * (x->remote_tiles[slot] ? x->remote_tiles[slot] : 1u) * 4u,
                               stream));
    }
    // 1. my slot is open for generation g -- ordered after everything the caller queued on `stream`
    //    (its consumer of the slot's previous content, and the render of this band)
    if (x->n_ranks > 1) {
        int rc = tr::launch_flags_store_all(x->d_open_list + (size_t)slot * x->n_ranks, x->n_ranks, r, g, stream);
        if (rc) return tr::fail(TR_E_HIP, "flag store launch failed");
    }
    HIP_TRY(hipEventRecord(x->fork, stream));
    // 2. per peer: wait for its "open", send my band (DMA engines, or the tile kernel), say "arrived"
    for (uint32_t p = 0; p < x->n_ranks; p++) {
        if (p == r) continue;
        hipStream_t c = x->copy_stream[p];
        HIP_TRY(hipStreamWaitEvent(c, x->fork, 0));
        int rc = tr::launch_flag_wait(x->flags + FlagIndex::open(slot, p), g, x->flags + FlagIndex::error(), x->timeout_ticks, c);
        if (rc) return tr::fail(TR_E_HIP, "flag wait launch failed");
        if (tiles) {
            // (a peer that never opened its slot: the wait has set the error word and the kernel writes nothing)
            rc = tr::launch_push_tiles(x->frame[slot], x->peer_frame[slot][p], tiles->clean_device,
                                       x->remote_clean[slot] + (size_t)p * x->remote_tiles[slot], frame,
                                       x->flags + FlagIndex::error(), x->d_tile_bytes, c);
            if (rc) return tr::fail(TR_E_HIP, "tile push launch failed");
        } else if (pull) {
            // p's band out of p's slot into mine (flag "open" read as "p's band is ready", "arrived" as "I have pulled")
            if (x->range_bytes[p])
                HIP_TRY(hipMemcpyAsync(x->frame[slot] + x->range_offset[p], x->peer_frame[slot][p] + x->range_offset[p], x->range_bytes[p],
                                       hipMemcpyDeviceToDevice, c));
        } else if (bytes) {
            // (push: after a timed-out wait this copy still lands -- see the file comment; declare the ranges to pull)
            HIP_TRY(hipMemcpyAsync(x->peer_frame[slot][p] + offset, x->frame[slot] + offset, bytes, hipMemcpyDeviceToDevice, c));
        }
        // (after a timeout the peer is not told "arrived": tr_exchange_status reports TR_E_EXCHANGE on both sides)
        rc = tr::launch_flag_store_unless(x->peer_flags[p] + FlagIndex::arrived(slot, r), g, x->flags + FlagIndex::error(), c);
        if (rc) return tr::fail(TR_E_HIP, "flag store launch failed");
        HIP_TRY(hipEventRecord(x->copy_done[p], c));
    }
    // 3. the caller's stream continues when my copies have left and every peer's band has arrived
    for (uint32_t p = 0; p < x->n_ranks; p++)
        if (p != r) HIP_TRY(hipStreamWaitEvent(stream, x->copy_done[p], 0));
    if (x->n_ranks > 1) {
        int rc = tr::launch_flags_wait_all(x->d_wait_list + (size_t)slot * x->n_ranks, x->n_ranks, r, g,
                                           x->flags + FlagIndex::error(), x->timeout_ticks, stream);
        if (rc) return tr::fail(TR_E_HIP, "flag wait launch failed");
    }
    // (only now: a call that failed half way has not used up the generation its peers are waiting for -- the
    // exchange is unusable after such a failure, but the next call does not pretend to be a later generation)
    x->generation[slot] = g;
    if (!tiles) x->bytes_sent += (uint64_t)bytes * (x->n_ranks - 1u);  // (pulled or pushed: what leaves this rank per call)
    return TR_OK;
}
}  // namespace

int tr_exchange_set_ranges(tr_exchange *x, const size_t *offsets, const size_t *bytes)
{
    if (!x) return tr::fail(TR_E_INVALID, "null exchange");
    if (!offsets || !bytes) {  // back to the push form
        x->range_offset.clear();
        x->range_bytes.clear();
        return TR_OK;
    }
    for (uint32_t p = 0; p < x->n_ranks; p++) {
        if (offsets[p] > x->frame_bytes || bytes[p] > x->frame_bytes - offsets[p])
            return tr::fail(TR_E_INVALID, "tr_exchange_set_ranges: a range lies outside the frame");
        for (uint32_t q = 0; q < p; q++)
            if (bytes[p] && bytes[q] && offsets[p] < offsets[q] + bytes[q] && offsets[q] < offsets[p] + bytes[p])
                return tr::fail(TR_E_INVALID, "tr_exchange_set_ranges: the ranks' ranges overlap");
    }
    x->range_offset.assign(offsets, offsets + x->n_ranks);
    x->range_bytes.assign(bytes, bytes + x->n_ranks);
    return TR_OK;
}

int tr_exchange_all_gather(tr_exchange *x, uint32_t slot, size_t offset, size_t bytes, void *stream_)
{
    if (!x || !x->connected) return tr::fail(TR_E_INVALID, "tr_exchange_all_gather: not connected");
    if (slot >= x->n_slots || offset > x->frame_bytes || bytes > x->frame_bytes - offset)
        return tr::fail(TR_E_INVALID, "tr_exchange_all_gather: slot or byte range outside the frame");
    hipStream_t stream = (hipStream_t)stream_;
    HIP_TRY(hipSetDevice(x->device));
    if (x->backend == TR_EXCHANGE_RCCL) {
        // in place: rank r's `bytes` at `offset` are piece r of n equal pieces that start at offset - r * bytes
        const size_t before = (size_t)x->rank * bytes;
        if (offset < before || offset - before + (size_t)x->n_ranks * bytes > x->frame_bytes)
            return tr::fail(TR_E_INVALID, "tr_exchange_all_gather (RCCL): the ranks' ranges must be equal pieces of one range of "
                                          "the frame, in rank order (tr_band_rows with a height the ranks divide)");
        RCCL_TRY(g_rccl.AllGather(x->frame[slot] + offset, x->frame[slot] + (offset - before), bytes, ncclUint8, x->comm, stream));
        x->bytes_sent += (uint64_t)bytes * (x->n_ranks - 1u);
        return TR_OK;
    }
    return peer_all_gather(x, slot, offset, bytes, nullptr, stream);
}

int tr_exchange_all_gather_tiles(tr_exchange *x, uint32_t slot, const tr_band_tiles *tiles, void *stream_)
{
    if (!x || !x->connected || !tiles) return tr::fail(TR_E_INVALID, "tr_exchange_all_gather_tiles: not connected, or null tiles");
    if (slot >= x->n_slots || tiles->frame_buffer_device != x->frame[slot])
        return tr::fail(TR_E_INVALID, "tr_exchange_all_gather_tiles: the tiles must describe the slot's own frame buffer");
    if ((size_t)tiles->width * tiles->height * 3u != x->frame_bytes || tiles->band_y0 < 0 || tiles->band_y0 > tiles->band_y1 ||
        (uint32_t)tiles->band_y1 > tiles->height)
        return tr::fail(TR_E_INVALID, "tr_exchange_all_gather_tiles: the frame does not match the exchange's");
    // the band's rows in the buffer (row 0 = top): [height - y1, height - y0)
    const size_t row = (size_t)tiles->width * 3u;
    const size_t offset = (size_t)(tiles->height - (uint32_t)tiles->band_y1) * row;
    const size_t bytes = (size_t)(tiles->band_y1 - tiles->band_y0) * row;
    if (x->backend == TR_EXCHANGE_RCCL || tiles->width % 16u != 0u) return tr_exchange_all_gather(x, slot, offset, bytes, stream_);
    HIP_TRY(hipSetDevice(x->device));
    return peer_all_gather(x, slot, offset, bytes, tiles, (hipStream_t)stream_);
}

uint64_t tr_exchange_bytes_sent(tr_exchange *x)
{
    if (!x) return 0u;
    unsigned long long tiles = 0;
    if (x->d_tile_bytes) {  // (what the sparse calls pushed is counted on the device: waits for it)
        (void)hipSetDevice(x->device);
        if (hipMemcpy(&tiles, x->d_tile_bytes, sizeof tiles, hipMemcpyDeviceToHost) != hipSuccess) tiles = 0;
    }
    return x->bytes_sent + (uint64_t)tiles;
}

int tr_exchange_status(tr_exchange *x)
{
    if (!x) return tr::fail(TR_E_INVALID, "null exchange");
    HIP_TRY(hipSetDevice(x->device));
    if (x->backend == TR_EXCHANGE_RCCL) {
        ncclResult_t async = ncclSuccess;
        if (x->comm) RCCL_TRY(g_rccl.CommGetAsyncError(x->comm, &async));
        if (async != ncclSuccess) return tr::fail(TR_E_RCCL, std::string("RCCL communicator: ") + g_rccl.GetErrorString(async));
        return TR_OK;
    }
    uint32_t err = 0;
    HIP_TRY(hipMemcpy(&err, x->flags + FlagIndex::error(), 4, hipMemcpyDeviceToHost));
    if (err) return tr::fail(TR_E_EXCHANGE, "a peer's band did not arrive in time (the rank is gone or out of step)");
    return TR_OK;
}

int tr_exchange_read(tr_exchange *x, uint32_t slot, void *host, size_t bytes)
{
    if (!x || !host || slot >= x->n_slots || bytes > x->frame_bytes) return tr::fail(TR_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(x->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host, x->frame[slot], bytes, hipMemcpyDeviceToHost));
    return tr_exchange_status(x);
}

int tr_exchange_disconnect(tr_exchange *x)
{
    if (!x) return tr::fail(TR_E_INVALID, "null exchange");
    HIP_TRY(hipSetDevice(x->device));
    HIP_TRY(hipDeviceSynchronize());
    for (uint32_t p = 0; p < x->peer_flags.size(); p++) {
        if (p == x->rank) continue;
        if (x->peer_base.size() > p && x->peer_base[p]) (void)hipIpcCloseMemHandle(x->peer_base[p]);
        if (x->peer_flags[p]) (void)hipIpcCloseMemHandle(x->peer_flags[p]);
        if (x->peer_base.size() > p) x->peer_base[p] = nullptr;
        x->peer_flags[p] = nullptr;
        for (uint32_t s = 0; s < MAX_SLOTS; s++)
            if (x->peer_frame[s].size() > p) x->peer_frame[s][p] = nullptr;
    }
    x->connected = false;
    return TR_OK;
}

void tr_exchange_destroy(tr_exchange *x)
{
    if (!x) return;
    (void)hipSetDevice(x->device);
    (void)hipDeviceSynchronize();
    if (x->comm) (void)g_rccl.CommDestroy(x->comm);
    for (uint32_t p = 0; p < x->peer_flags.size(); p++) {
        if (p == x->rank) continue;
        if (x->peer_base.size() > p && x->peer_base[p]) (void)hipIpcCloseMemHandle(x->peer_base[p]);
        if (x->peer_flags[p]) (void)hipIpcCloseMemHandle(x->peer_flags[p]);
        if (x->copy_stream[p]) (void)hipStreamDestroy(x->copy_stream[p]);
        if (x->copy_done[p]) (void)hipEventDestroy(x->copy_done[p]);
    }
    if (x->fork) (void)hipEventDestroy(x->fork);
    if (x->d_wait_list) (void)hipFree(x->d_wait_list);
    if (x->d_open_list) (void)hipFree(x->d_open_list);
    for (uint32_t k = 0; k < MAX_SLOTS; k++)
        if (x->remote_clean[k]) (void)hipFree(x->remote_clean[k]);
    if (x->d_tile_bytes) (void)hipFree(x->d_tile_bytes);
    if (x->frames) (void)hipFree(x->frames);
    if (x->flags) (void)hipFree(x->flags);
    delete x;
}

}  // extern "C"
