// Vote exchange over RCCL behind the C ABI (SURVEY section 8(b) "vote_allgather", 8(e)): frames shard across
// ranks with no data-path collective; the only exchange is one ncclAllGather per wave of a few fixed-size
// records per rank, after which every rank replays the votes in frame order.
//
// librccl is opened lazily (dlopen) the first time a communicator is asked for, so the library loads - and every
// other entry point works - on a host without RCCL, and a process that already carries an RCCL (PyTorch's) reuses
// that copy instead of loading a second one.
#include <dlfcn.h>

#include "dfd_common.h"

using namespace dfd;

namespace {

typedef struct ncclComm* ncclComm_t;
struct NcclId { char internal[DFD_COMM_ID_BYTES]; };
enum { kNcclSuccess = 0, kNcclInt8 = 0 };

struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(NcclId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, NcclId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
};

Rccl* rccl() {
    static Rccl R;
    if (R.lib || !R.err.empty()) return &R;
    const char* env = getenv("DFD_RCCL_LIB");
    const char* names[] = {env, "librccl.so.1", "librccl.so"};
    for (int pass = 0; pass < 2 && !R.lib; ++pass)             // pass 0: a copy this process already loaded
        for (const char* nm : names) {
            if (!nm) continue;
            R.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (R.lib) break;
        }
    if (!R.lib) { R.err = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?"); return &R; }
    R.GetUniqueId = reinterpret_cast<decltype(R.GetUniqueId)>(dlsym(R.lib, "ncclGetUniqueId"));
    R.CommInitRank = reinterpret_cast<decltype(R.CommInitRank)>(dlsym(R.lib, "ncclCommInitRank"));
    R.CommDestroy = reinterpret_cast<decltype(R.CommDestroy)>(dlsym(R.lib, "ncclCommDestroy"));
    R.AllGather = reinterpret_cast<decltype(R.AllGather)>(dlsym(R.lib, "ncclAllGather"));
    R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(dlsym(R.lib, "ncclGetErrorString"));
    if (!R.GetUniqueId || !R.CommInitRank || !R.CommDestroy || !R.AllGather) {
        R.err = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather";
        R.lib = nullptr;
    }
    return &R;
}

const char* nerr(Rccl* R, int code) { return R->GetErrorString ? R->GetErrorString(code) : "nccl error"; }

}  // namespace

namespace dfd {

struct CommState {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    DevBuf send, recv;
};

void comm_destroy(dfd_handle* h) {
    if (!h->comm) return;
    if (h->comm->comm && rccl()->CommDestroy) rccl()->CommDestroy(h->comm->comm);
    delete h->comm;
    h->comm = nullptr;
}

}  // namespace dfd

extern "C" {

int dfd_comm_unique_id(void* id_out) {
    if (!id_out) return fail(nullptr, DFD_ERR_ARG, "comm_unique_id: null pointer");
    Rccl* R = rccl();
    if (!R->lib) return fail(nullptr, DFD_ERR_STATE, "%s", R->err.c_str());
    NcclId id;
    const int rc = R->GetUniqueId(&id);
    if (rc != kNcclSuccess) return fail(nullptr, DFD_ERR_HIP, "ncclGetUniqueId: %s", nerr(R, rc));
    memcpy(id_out, id.internal, DFD_COMM_ID_BYTES);
    return DFD_OK;
}

int dfd_comm_init(dfd_handle* h, const void* id, int rank, int world) {
    if (!h) return DFD_ERR_ARG;
    if (!id || world <= 0 || rank < 0 || rank >= world) return fail(h, DFD_ERR_ARG, "comm_init: bad id / rank %d / world %d", rank, world);
    if (h->comm) return fail(h, DFD_ERR_STATE, "comm_init: the handle already has a communicator");
    Rccl* R = rccl();
    if (!R->lib) return fail(h, DFD_ERR_STATE, "%s", R->err.c_str());
    DFD_HIP_TRY(h, hipSetDevice(h->device));
    CommState* C = new (std::nothrow) CommState();
    if (!C) return fail(h, DFD_ERR_ARG, "out of host memory");
    NcclId nid;
    memcpy(nid.internal, id, DFD_COMM_ID_BYTES);
    const int rc = R->CommInitRank(&C->comm, world, nid, rank);
    if (rc != kNcclSuccess) {
        delete C;
        return fail(h, DFD_ERR_HIP, "ncclCommInitRank(rank %d of %d): %s", rank, world, nerr(R, rc));
    }
    C->rank = rank;
    C->world = world;
    h->comm = C;
    return DFD_OK;
}

int dfd_comm_destroy(dfd_handle* h) {
    if (!h) return DFD_ERR_ARG;
    if (h->stream) stream_sync(h);
    comm_destroy(h);
    return DFD_OK;
}

int dfd_comm_info(const dfd_handle* h, int* rank, int* world) {
    if (!h) return DFD_ERR_ARG;
    if (rank) *rank = h->comm ? h->comm->rank : 0;
    if (world) *world = h->comm ? h->comm->world : 0;
    return DFD_OK;
}

int dfd_vote_allgather(dfd_handle* h, const void* local_records, size_t bytes_per_rank, void* all_records_out) {
    if (!h) return DFD_ERR_ARG;
    if (!local_records || !all_records_out || bytes_per_rank == 0) return fail(h, DFD_ERR_ARG, "vote_allgather: null pointer or empty record block");
    if (!h->comm) return fail(h, DFD_ERR_STATE, "vote_allgather: no communicator (dfd_comm_init)");
    CommState& C = *h->comm;
    Rccl* R = rccl();
    DFD_HIP_TRY(h, hipSetDevice(h->device));
    int rc;
    if ((rc = ensure(h, &C.send, bytes_per_rank))) return rc;
    if ((rc = ensure(h, &C.recv, bytes_per_rank * C.world))) return rc;
    DFD_HIP_TRY(h, hipMemcpyAsync(C.send.p, local_records, bytes_per_rank, hipMemcpyHostToDevice, h->stream));
    const int nrc = R->AllGather(C.send.p, C.recv.p, bytes_per_rank, kNcclInt8, C.comm, h->stream);
    if (nrc != kNcclSuccess) return fail(h, DFD_ERR_HIP, "ncclAllGather: %s", nerr(R, nrc));
    DFD_HIP_TRY(h, hipMemcpyAsync(all_records_out, C.recv.p, bytes_per_rank * C.world, hipMemcpyDeviceToHost, h->stream));
    DFD_HIP_TRY(h, stream_sync(h));
    return DFD_OK;
}

// The record blocks of `waves` consecutive waves in one call: ONE upload, one ncclAllGather per wave (wave w gathers
// [world][bytes_per_rank] into slot w - the collective the frame order needs stays per wave), ONE download and ONE
// stream wait.  With look-ahead batching (streams.py) a rank has the records of its next L waves before the first
// exchange, so the per-wave upload + wait of dfd_vote_allgather (two PCIe latencies and a drained stream per wave)
// is paid once per L waves; the gathered bytes and their order are the same.
int dfd_vote_allgather_waves(dfd_handle* h, const void* local_records, int waves, size_t bytes_per_rank, void* all_records_out) {
    if (!h) return DFD_ERR_ARG;
    if (!local_records || !all_records_out || bytes_per_rank == 0 || waves <= 0)
        return fail(h, DFD_ERR_ARG, "vote_allgather_waves: null pointer, no waves or empty record block");
    if (!h->comm) return fail(h, DFD_ERR_STATE, "vote_allgather_waves: no communicator (dfd_comm_init)");
    CommState& C = *h->comm;
    Rccl* R = rccl();
    DFD_HIP_TRY(h, hipSetDevice(h->device));
    int rc;
    const size_t out_per_wave = bytes_per_rank * C.world;
    if ((rc = ensure(h, &C.send, bytes_per_rank * waves))) return rc;
    if ((rc = ensure(h, &C.recv, out_per_wave * waves))) return rc;
    DFD_HIP_TRY(h, hipMemcpyAsync(C.send.p, local_records, bytes_per_rank * waves, hipMemcpyHostToDevice, h->stream));
    for (int w = 0; w < waves; ++w) {
        const int nrc = R->AllGather(static_cast<const char*>(C.send.p) + (size_t)w * bytes_per_rank,
                                     static_cast<char*>(C.recv.p) + (size_t)w * out_per_wave, bytes_per_rank, kNcclInt8, C.comm, h->stream);
        if (nrc != kNcclSuccess) return fail(h, DFD_ERR_HIP, "ncclAllGather (wave %d of %d): %s", w, waves, nerr(R, nrc));
    }
    DFD_HIP_TRY(h, hipMemcpyAsync(all_records_out, C.recv.p, out_per_wave * waves, hipMemcpyDeviceToHost, h->stream));
    DFD_HIP_TRY(h, stream_sync(h));
    return DFD_OK;
}

}  // extern "C"
