// RCCL transport of the Z-slab multi-GPU run (include/mgps.h, struct mgps_comm).  The reference is
// single-process shared memory, so nothing here has a reference counterpart: the ghost-plane
// exchange is one ncclSend/ncclRecv pair per Z-neighbour inside a single group call, enqueued on
// the solver's stream, so it is ordered with the kernels around it without any host
// synchronisation; over xGMI each GPU talks to at most two peers, each on its own direct link.
//
// librccl is opened with dlopen at communicator creation: a single-GPU user never loads it, and the
// library links (and its CPU tests run) on machines without RCCL.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "mgps_internal.h"

using namespace mgps;

namespace {

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi gApi;
std::once_flag gApiOnce;
std::string gApiError;

template <class F>
bool sym(void *lib, const char *name, F &fn)
{
    fn = reinterpret_cast<F>(dlsym(lib, name));
    if (!fn) gApiError = std::string("librccl: missing symbol ") + name;
    return fn != nullptr;
}

bool loadRccl()
{
    std::call_once(gApiOnce, [] {
        // a process that already holds an RCCL (torch bundles one with the same SONAME) gets that copy
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            gApi.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (gApi.lib) break;
        }
        if (!gApi.lib) {
            gApiError = std::string("cannot open librccl: ") + dlerror();
            return;
        }
        bool ok = sym(gApi.lib, "ncclGetUniqueId", gApi.GetUniqueId) && sym(gApi.lib, "ncclCommInitRank", gApi.CommInitRank) &&
                  sym(gApi.lib, "ncclCommDestroy", gApi.CommDestroy) && sym(gApi.lib, "ncclGroupStart", gApi.GroupStart) &&
                  sym(gApi.lib, "ncclGroupEnd", gApi.GroupEnd) && sym(gApi.lib, "ncclSend", gApi.Send) &&
                  sym(gApi.lib, "ncclRecv", gApi.Recv) && sym(gApi.lib, "ncclAllReduce", gApi.AllReduce) &&
                  sym(gApi.lib, "ncclGetErrorString", gApi.GetErrorString);
        if (!ok) gApi.lib = nullptr;
    });
    return gApi.lib != nullptr;
}

struct RcclState {
    ncclComm_t comm = nullptr;
    int rank = 0, size = 1, device = 0;
    double *scratchDev = nullptr;  // all-reduce staging
    double *scratchHost = nullptr;
    hipStream_t own = nullptr;     // stream of the scalar all-reduces
    // The solver uses the communicator from more than one stream (the scalar all-reduces above; since round 4 the exchanges that
    // follow a sweep go on a transfer stream beside the sweep's interior part).  Every rank issues its calls in the same
    // order, which is what the library asks for; so that two operations of one communicator never run side by side on the
    // device -- a combination this code has never been run with on more than one GPU -- each operation waits for the event
    // the previous one left when it was queued on another stream.  Compute beside a transfer is unaffected.
    hipEvent_t lastDone = nullptr;
    hipStream_t lastStream = nullptr;
    bool used = false;
};
int commBegin(RcclState *s, hipStream_t st)
{
    if (s->used && s->lastStream != st && s->lastDone && hipStreamWaitEvent(st, s->lastDone, 0) != hipSuccess) return 1;
    return 0;
}
int commEnd(RcclState *s, hipStream_t st)
{
    if (!s->lastDone && hipEventCreateWithFlags(&s->lastDone, hipEventDisableTiming) != hipSuccess) return 1;
    if (hipEventRecord(s->lastDone, st) != hipSuccess) return 1;
    s->lastStream = st;
    s->used = true;
    return 0;
}
#define COMM_ORDER_BEGIN(s, st)                                                         \
    do {                                                                                \
        if (commBegin(s, st)) {                                                         \
            setLastGlobalError("mgps rccl transport: hipStreamWaitEvent failed");       \
            return 1;                                                                   \
        }                                                                               \
    } while (0)
#define COMM_ORDER_END(s, st)                                                           \
    do {                                                                                \
        if (commEnd(s, st)) {                                                           \
            setLastGlobalError("mgps rccl transport: hipEventRecord failed");           \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

#define NCCL_TRY(call)                                                                              \
    do {                                                                                            \
        ncclResult_t r_ = (call);                                                                   \
        if (r_ != ncclSuccess) {                                                                    \
            setLastGlobalError(std::string(#call) + ": " + gApi.GetErrorString(r_));                \
            return 1;                                                                               \
        }                                                                                           \
    } while (0)
#define HIP_TRY(call)                                                                \
    do {                                                                             \
        hipError_t e_ = (call);                                                      \
        if (e_ != hipSuccess) {                                                      \
            setLastGlobalError(std::string(#call) + ": " + hipGetErrorString(e_));   \
            return 1;                                                                \
        }                                                                            \
    } while (0)

int rcclExchange(void *user, const void *sendLo, size_t sendLoBytes, void *recvLo, size_t recvLoBytes,
                 const void *sendHi, size_t sendHiBytes, void *recvHi, size_t recvHiBytes, void *stream)
{
    auto *s = static_cast<RcclState *>(user);
    hipStream_t st = static_cast<hipStream_t>(stream);
    COMM_ORDER_BEGIN(s, st);
    NCCL_TRY(gApi.GroupStart());
    if (sendLo && sendLoBytes) NCCL_TRY(gApi.Send(sendLo, sendLoBytes, ncclChar, s->rank - 1, s->comm, st));
    if (recvLo && recvLoBytes) NCCL_TRY(gApi.Recv(recvLo, recvLoBytes, ncclChar, s->rank - 1, s->comm, st));
    if (sendHi && sendHiBytes) NCCL_TRY(gApi.Send(sendHi, sendHiBytes, ncclChar, s->rank + 1, s->comm, st));
    if (recvHi && recvHiBytes) NCCL_TRY(gApi.Recv(recvHi, recvHiBytes, ncclChar, s->rank + 1, s->comm, st));
    NCCL_TRY(gApi.GroupEnd());
    COMM_ORDER_END(s, st);
    return 0;
}

int rcclExchange2(void *user, const mgps_xfer2 *sendLo, const mgps_xfer2 *recvLo, const mgps_xfer2 *sendHi, const mgps_xfer2 *recvHi, void *stream)
{
    auto *s = static_cast<RcclState *>(user);
    hipStream_t st = static_cast<hipStream_t>(stream);
    COMM_ORDER_BEGIN(s, st);
    NCCL_TRY(gApi.GroupStart());
    for (int q = 0; q < 2; ++q) {  // segment 0 of every message, then segment 1: the same order on both ends of a link
        if (sendLo && sendLo->bytes[q]) NCCL_TRY(gApi.Send(sendLo->ptr[q], sendLo->bytes[q], ncclChar, s->rank - 1, s->comm, st));
        if (recvLo && recvLo->bytes[q]) NCCL_TRY(gApi.Recv(recvLo->ptr[q], recvLo->bytes[q], ncclChar, s->rank - 1, s->comm, st));
        if (sendHi && sendHi->bytes[q]) NCCL_TRY(gApi.Send(sendHi->ptr[q], sendHi->bytes[q], ncclChar, s->rank + 1, s->comm, st));
        if (recvHi && recvHi->bytes[q]) NCCL_TRY(gApi.Recv(recvHi->ptr[q], recvHi->bytes[q], ncclChar, s->rank + 1, s->comm, st));
    }
    NCCL_TRY(gApi.GroupEnd());
    COMM_ORDER_END(s, st);
    return 0;
}

int rcclAllreduce(void *user, double *values, int count, int op)
{
    auto *s = static_cast<RcclState *>(user);
    if (count > 64) return 1;
    std::memcpy(s->scratchHost, values, size_t(count) * sizeof(double));
    HIP_TRY(hipMemcpyAsync(s->scratchDev, s->scratchHost, size_t(count) * sizeof(double), hipMemcpyHostToDevice, s->own));
    COMM_ORDER_BEGIN(s, s->own);
    NCCL_TRY(gApi.AllReduce(s->scratchDev, s->scratchDev, size_t(count), ncclDouble, op == 0 ? ncclSum : ncclMax, s->comm, s->own));
    COMM_ORDER_END(s, s->own);
    HIP_TRY(hipMemcpyAsync(s->scratchHost, s->scratchDev, size_t(count) * sizeof(double), hipMemcpyDeviceToHost, s->own));
    HIP_TRY(hipStreamSynchronize(s->own));
    std::memcpy(values, s->scratchHost, size_t(count) * sizeof(double));
    return 0;
}

// the same on device doubles, on the caller's stream: ordered with the kernels around it, no host hop
int rcclAllreduceDevice(void *user, double *valuesDev, int count, int op, void *stream)
{
    auto *s = static_cast<RcclState *>(user);
    hipStream_t st = static_cast<hipStream_t>(stream);
    COMM_ORDER_BEGIN(s, st);
    NCCL_TRY(gApi.AllReduce(valuesDev, valuesDev, size_t(count), ncclDouble, op == 0 ? ncclSum : ncclMax, s->comm, st));
    COMM_ORDER_END(s, st);
    return 0;
}

int rcclGather(void *user, const void *send, void *recv, size_t bytes, int root, void *stream)
{
    auto *s = static_cast<RcclState *>(user);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (s->rank == root) HIP_TRY(hipMemcpyAsync(static_cast<char *>(recv) + size_t(root) * bytes, send, bytes, hipMemcpyDeviceToDevice, st));
    COMM_ORDER_BEGIN(s, st);
    NCCL_TRY(gApi.GroupStart());
    if (s->rank == root) {
        for (int r = 0; r < s->size; ++r)
            if (r != root) NCCL_TRY(gApi.Recv(static_cast<char *>(recv) + size_t(r) * bytes, bytes, ncclChar, r, s->comm, st));
    } else
        NCCL_TRY(gApi.Send(send, bytes, ncclChar, root, s->comm, st));
    NCCL_TRY(gApi.GroupEnd());
    COMM_ORDER_END(s, st);
    return 0;
}

int rcclGatherv(void *user, const void *send, size_t sendBytes, void *recv, const size_t *counts, const size_t *displs, int root, void *stream)
{
    auto *s = static_cast<RcclState *>(user);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (s->rank == root && sendBytes)
        HIP_TRY(hipMemcpyAsync(static_cast<char *>(recv) + displs[root], send, sendBytes, hipMemcpyDeviceToDevice, st));
    COMM_ORDER_BEGIN(s, st);
    NCCL_TRY(gApi.GroupStart());
    if (s->rank == root) {
        for (int r = 0; r < s->size; ++r)
            if (r != root && counts[r]) NCCL_TRY(gApi.Recv(static_cast<char *>(recv) + displs[r], counts[r], ncclChar, r, s->comm, st));
    } else if (sendBytes)
        NCCL_TRY(gApi.Send(send, sendBytes, ncclChar, root, s->comm, st));
    NCCL_TRY(gApi.GroupEnd());
    COMM_ORDER_END(s, st);
    return 0;
}

int rcclScatterv(void *user, const void *send, const size_t *counts, const size_t *displs, void *recv, size_t recvBytes, int root, void *stream)
{
    auto *s = static_cast<RcclState *>(user);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (s->rank == root && recvBytes)
        HIP_TRY(hipMemcpyAsync(recv, static_cast<const char *>(send) + displs[root], recvBytes, hipMemcpyDeviceToDevice, st));
    COMM_ORDER_BEGIN(s, st);
    NCCL_TRY(gApi.GroupStart());
    if (s->rank == root) {
        for (int r = 0; r < s->size; ++r)
            if (r != root && counts[r]) NCCL_TRY(gApi.Send(static_cast<const char *>(send) + displs[r], counts[r], ncclChar, r, s->comm, st));
    } else if (recvBytes)
        NCCL_TRY(gApi.Recv(recv, recvBytes, ncclChar, root, s->comm, st));
    NCCL_TRY(gApi.GroupEnd());
    COMM_ORDER_END(s, st);
    return 0;
}

int rcclScatter(void *user, const void *send, void *recv, size_t bytes, int root, void *stream)
{
    auto *s = static_cast<RcclState *>(user);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (s->rank == root) HIP_TRY(hipMemcpyAsync(recv, static_cast<const char *>(send) + size_t(root) * bytes, bytes, hipMemcpyDeviceToDevice, st));
    COMM_ORDER_BEGIN(s, st);
    NCCL_TRY(gApi.GroupStart());
    if (s->rank == root) {
        for (int r = 0; r < s->size; ++r)
            if (r != root) NCCL_TRY(gApi.Send(static_cast<const char *>(send) + size_t(r) * bytes, bytes, ncclChar, r, s->comm, st));
    } else
        NCCL_TRY(gApi.Recv(recv, bytes, ncclChar, root, s->comm, st));
    NCCL_TRY(gApi.GroupEnd());
    COMM_ORDER_END(s, st);
    return 0;
}

void rcclDestroy(void *user)
{
    auto *s = static_cast<RcclState *>(user);
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->lastDone) (void)hipEventDestroy(s->lastDone);
    if (s->comm) gApi.CommDestroy(s->comm);
    (void)hipFree(s->scratchDev);
    if (s->scratchHost) (void)hipHostFree(s->scratchHost);
    if (s->own) (void)hipStreamDestroy(s->own);
    delete s;
}

}  // namespace

extern "C" {

int mgps_rccl_unique_id(unsigned char out_id[128])
try {
    static_assert(sizeof(ncclUniqueId) == 128, "mgps_rccl_unique_id ships 128 bytes");
    if (!out_id) {
        setLastGlobalError("mgps_rccl_unique_id: NULL");
        return MGPS_ERR_INVALID_ARGUMENT;
    }
    if (!loadRccl()) {
        setLastGlobalError(gApiError);
        return MGPS_ERR_COMM;
    }
    ncclUniqueId id;
    if (gApi.GetUniqueId(&id) != ncclSuccess) {
        setLastGlobalError("ncclGetUniqueId failed");
        return MGPS_ERR_COMM;
    }
    std::memcpy(out_id, &id, 128);
    return MGPS_OK;
}
MGPS_API_CATCH(nullptr)

int mgps_comm_create_rccl(mgps_comm *out, int rank, int size, const unsigned char id[128], int device)
try {
    if (!out || !id || size < 1 || rank < 0 || rank >= size) {
        setLastGlobalError("mgps_comm_create_rccl: bad arguments");
        return MGPS_ERR_INVALID_ARGUMENT;
    }
    std::memset(out, 0, sizeof(*out));
    if (!loadRccl()) {
        setLastGlobalError(gApiError);
        return MGPS_ERR_COMM;
    }
    if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;
    if (hipSetDevice(device) != hipSuccess) {
        setLastGlobalError("mgps_comm_create_rccl: hipSetDevice failed");
        return MGPS_ERR_NO_DEVICE;
    }
    auto *s = new RcclState();
    s->rank = rank;
    s->size = size;
    s->device = device;
    ncclUniqueId uid;
    std::memcpy(&uid, id, 128);
    ncclResult_t r = gApi.CommInitRank(&s->comm, size, uid, rank);
    if (r != ncclSuccess) {
        setLastGlobalError(std::string("ncclCommInitRank: ") + gApi.GetErrorString(r));
        delete s;
        return MGPS_ERR_COMM;
    }
    if (hipMalloc(reinterpret_cast<void **>(&s->scratchDev), 64 * sizeof(double)) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **>(&s->scratchHost), 64 * sizeof(double)) != hipSuccess ||
        hipStreamCreateWithFlags(&s->own, hipStreamNonBlocking) != hipSuccess) {
        setLastGlobalError("mgps_comm_create_rccl: scratch allocation failed");
        rcclDestroy(s);
        return MGPS_ERR_ALLOC;
    }
    out->struct_size = int(sizeof(mgps_comm));
    out->rank = rank;
    out->size = size;
    out->user = s;
    out->exchange = rcclExchange;
    out->allreduce = rcclAllreduce;
    out->gather = rcclGather;
    out->scatter = rcclScatter;
    out->destroy = rcclDestroy;
    out->gatherv = rcclGatherv;
    out->scatterv = rcclScatterv;
    out->allreduce_device = rcclAllreduceDevice;
    out->exchange2 = rcclExchange2;
    return MGPS_OK;
}
MGPS_API_CATCH(nullptr)

int mgps_comm_rccl_selftest(mgps_comm *comm, size_t floats)
try {
    // ncclSend + ncclRecv to this rank itself inside one group, the call shape of the ghost exchange: checks on
    // any box (a single GPU is enough) that librccl's point-to-point path launches, completes in stream order and
    // delivers the bytes.  Only valid for transports made by mgps_comm_create_rccl.
    if (!comm || comm->exchange != rcclExchange || floats == 0) {
        setLastGlobalError("mgps_comm_rccl_selftest: not an RCCL transport");
        return MGPS_ERR_INVALID_ARGUMENT;
    }
    auto *s = static_cast<RcclState *>(comm->user);
    float *src = nullptr, *dst = nullptr;
    std::vector<float> host(floats);
    for (size_t i = 0; i < floats; ++i) host[i] = float(i % 977) + 0.5f;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&src), floats * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&dst), floats * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(src, host.data(), floats * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(dst, 0, floats * sizeof(float));
    int rc = MGPS_OK;
    if (e == hipSuccess) {
        ncclResult_t r = gApi.GroupStart();
        if (r == ncclSuccess) r = gApi.Send(src, floats * sizeof(float), ncclChar, s->rank, s->comm, s->own);
        if (r == ncclSuccess) r = gApi.Recv(dst, floats * sizeof(float), ncclChar, s->rank, s->comm, s->own);
        if (r == ncclSuccess) r = gApi.GroupEnd();
        if (r != ncclSuccess) {
            setLastGlobalError(std::string("mgps_comm_rccl_selftest: ") + gApi.GetErrorString(r));
            rc = MGPS_ERR_COMM;
        }
        if (rc == MGPS_OK) e = hipStreamSynchronize(s->own);
        std::vector<float> back(floats, -1.f);
        if (rc == MGPS_OK && e == hipSuccess) e = hipMemcpy(back.data(), dst, floats * sizeof(float), hipMemcpyDeviceToHost);
        if (rc == MGPS_OK && e == hipSuccess && back != host) {
            setLastGlobalError("mgps_comm_rccl_selftest: received data differ from the data sent");
            rc = MGPS_ERR_COMM;
        }
        // the call shape of exchange2: two segments per message, all four operations in one group, segment by segment
        if (rc == MGPS_OK && e == hipSuccess && floats >= 2) {
            e = hipMemset(dst, 0, floats * sizeof(float));
            const size_t a = floats / 3 + 1, b = floats - a;  // uneven halves
            if (e == hipSuccess) {
                r = gApi.GroupStart();
                if (r == ncclSuccess) r = gApi.Send(src, a * sizeof(float), ncclChar, s->rank, s->comm, s->own);
                if (r == ncclSuccess) r = gApi.Recv(dst, a * sizeof(float), ncclChar, s->rank, s->comm, s->own);
                if (r == ncclSuccess) r = gApi.Send(src + a, b * sizeof(float), ncclChar, s->rank, s->comm, s->own);
                if (r == ncclSuccess) r = gApi.Recv(dst + a, b * sizeof(float), ncclChar, s->rank, s->comm, s->own);
                if (r == ncclSuccess) r = gApi.GroupEnd();
                if (r != ncclSuccess) {
                    setLastGlobalError(std::string("mgps_comm_rccl_selftest (two segments): ") + gApi.GetErrorString(r));
                    rc = MGPS_ERR_COMM;
                }
                if (rc == MGPS_OK) e = hipStreamSynchronize(s->own);
                if (rc == MGPS_OK && e == hipSuccess) e = hipMemcpy(back.data(), dst, floats * sizeof(float), hipMemcpyDeviceToHost);
                if (rc == MGPS_OK && e == hipSuccess && back != host) {
                    setLastGlobalError("mgps_comm_rccl_selftest: two-segment messages arrived out of order or incomplete");
                    rc = MGPS_ERR_COMM;
                }
            }
        }
    }
    (void)hipFree(src);
    (void)hipFree(dst);
    if (e != hipSuccess) {
        setLastGlobalError(std::string("mgps_comm_rccl_selftest: ") + hipGetErrorString(e));
        return MGPS_ERR_HIP;
    }
    return rc;
}
MGPS_API_CATCH(nullptr)

// A check of ANY transport (the vtable alone: RCCL, or the test-suite's torch.distributed one) before a run trusts it: every rank
// pushes rank-stamped data through exchange, exchange2, allreduce_device, allreduce and gather / scatter and verifies what arrives
// from its neighbours.  Every rank walks through all steps whatever it finds, and the verdicts are joined at the end (a rank that
// left early would leave the others waiting).  *ranks_seen = the sum over ranks of 1, as the device all-reduce delivered it.
int mgps_comm_preflight(const mgps_comm *comm, size_t floats, int *ranks_seen)
try {
    if (!comm || !comm->exchange || !comm->allreduce || !comm->gather || !comm->scatter || floats < 4) {
        setLastGlobalError("mgps_comm_preflight: incomplete mgps_comm");
        return MGPS_ERR_INVALID_ARGUMENT;
    }
    const int rank = comm->rank, size = comm->size;
    const bool lo = rank > 0, hi = rank < size - 1;
    std::string what;
    auto note = [&](const std::string &m) {
        if (what.empty()) what = m;
    };
    auto stamp = [](int r, int side, size_t i) { return float(r) + 0.25f * float(side) + 1e-3f * float(i % 101); };
    float *dev = nullptr;  // send lo | send hi | recv lo | recv hi
    double *dd = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&dev), 4 * floats * sizeof(float)) != hipSuccess || hipMalloc(reinterpret_cast<void **>(&dd), 2 * sizeof(double)) != hipSuccess) {
        (void)hipGetLastError();
        if (dev) (void)hipFree(dev);
        setLastGlobalError("mgps_comm_preflight: allocation failed");
        return MGPS_ERR_ALLOC;
    }
    std::vector<float> host(4 * floats, -1.f), back(2 * floats);
    for (size_t i = 0; i < floats; ++i) {
        host[i] = stamp(rank, 1, i);           // what goes down
        host[floats + i] = stamp(rank, 2, i);  // what goes up
    }
    auto upload = [&] { return hipMemcpy(dev, host.data(), 4 * floats * sizeof(float), hipMemcpyHostToDevice) == hipSuccess; };
    auto verify = [&](const char *step) {
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(back.data(), dev + 2 * floats, 2 * floats * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
            note(std::string(step) + ": reading the received data failed");
            return;
        }
        for (size_t i = 0; i < floats; ++i) {
            if (lo && back[i] != stamp(rank - 1, 2, i)) {
                note(std::string(step) + ": data from the lower neighbour differ from what it sent");
                return;
            }
            if (hi && back[floats + i] != stamp(rank + 1, 1, i)) {
                note(std::string(step) + ": data from the upper neighbour differ from what it sent");
                return;
            }
        }
    };
    const size_t bytes = floats * sizeof(float);
    if (!upload()) note("upload failed");
    if (comm->exchange(comm->user, lo ? dev : nullptr, bytes, lo ? dev + 2 * floats : nullptr, bytes, hi ? dev + floats : nullptr, bytes, hi ? dev + 3 * floats : nullptr, bytes, nullptr) != 0)
        note("exchange returned an error");
    verify("exchange");
    if (comm->exchange2) {
        if (!upload()) note("upload failed");
        const size_t a = floats / 3 + 1, b = floats - a;  // two uneven segments per message
        mgps_xfer2 seg[4] = {{{dev, dev + a}, {a * sizeof(float), b * sizeof(float)}},
                             {{dev + 2 * floats, dev + 2 * floats + a}, {a * sizeof(float), b * sizeof(float)}},
                             {{dev + floats, dev + floats + a}, {a * sizeof(float), b * sizeof(float)}},
                             {{dev + 3 * floats, dev + 3 * floats + a}, {a * sizeof(float), b * sizeof(float)}}};
        if (comm->exchange2(comm->user, lo ? &seg[0] : nullptr, lo ? &seg[1] : nullptr, hi ? &seg[2] : nullptr, hi ? &seg[3] : nullptr, nullptr) != 0)
            note("exchange2 returned an error");
        verify("exchange2");
    }
    double seen = 0.0;
    if (comm->allreduce_device) {
        const double ones[2] = {1.0, double(rank)};
        if (hipMemcpy(dd, ones, sizeof(ones), hipMemcpyHostToDevice) != hipSuccess) note("upload failed");
        if (comm->allreduce_device(comm->user, dd, 1, 0, nullptr) != 0 || comm->allreduce_device(comm->user, dd + 1, 1, 1, nullptr) != 0) note("allreduce_device returned an error");
        double got[2] = {0, 0};
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(got, dd, sizeof(got), hipMemcpyDeviceToHost) != hipSuccess) note("allreduce_device: reading the result failed");
        if (got[0] != double(size) || got[1] != double(size - 1)) note("allreduce_device: sum / max over the ranks is wrong");
        seen = got[0];
    }
    {
        double v[2] = {1.0, double(rank)};
        if (comm->allreduce(comm->user, v, 1, 0) != 0 || comm->allreduce(comm->user, v + 1, 1, 1) != 0) note("allreduce returned an error");
        if (v[0] != double(size) || v[1] != double(size - 1)) note("allreduce: sum / max over the ranks is wrong");
        if (!comm->allreduce_device) seen = v[0];
    }
    {  // gather to rank 0 and scatter back (the collapse's collectives): a stamp per rank
        const size_t n = std::min<size_t>(floats, 64), gb = n * sizeof(float);
        float *all = nullptr;
        if (rank == 0 && hipMalloc(reinterpret_cast<void **>(&all), gb * size_t(size)) != hipSuccess) note("allocation failed");
        if (comm->gather(comm->user, dev, rank == 0 ? all : nullptr, gb, 0, nullptr) != 0) note("gather returned an error");
        if (rank == 0 && all) {
            std::vector<float> g(n * size_t(size));
            if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(g.data(), all, gb * size_t(size), hipMemcpyDeviceToHost) != hipSuccess) note("gather: reading the result failed");
            for (int r = 0; r < size; ++r)
                for (size_t i = 0; i < n; ++i)
                    if (g[size_t(r) * n + i] != stamp(r, 1, i)) note("gather: rank " + std::to_string(r) + "'s share is wrong");
        }
        if (comm->scatter(comm->user, rank == 0 ? all : nullptr, dev + 2 * floats, gb, 0, nullptr) != 0) note("scatter returned an error");
        std::vector<float> mine(n);
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(mine.data(), dev + 2 * floats, gb, hipMemcpyDeviceToHost) != hipSuccess) note("scatter: reading the result failed");
        for (size_t i = 0; i < n; ++i)
            if (mine[i] != stamp(rank, 1, i)) note("scatter: this rank's share is wrong");
        if (all) (void)hipFree(all);
    }
    (void)hipFree(dev);
    (void)hipFree(dd);
    double failed = what.empty() ? 0.0 : 1.0;
    (void)comm->allreduce(comm->user, &failed, 1, 1);
    if (ranks_seen) *ranks_seen = int(seen);
    if (!what.empty()) {
        setLastGlobalError("mgps_comm_preflight (rank " + std::to_string(rank) + "): " + what);
        return MGPS_ERR_COMM;
    }
    if (failed != 0.0) {
        setLastGlobalError("mgps_comm_preflight: another rank found the transport broken");
        return MGPS_ERR_COMM;
    }
    return MGPS_OK;
}
MGPS_API_CATCH(nullptr)


int mgps_comm_rccl_selfbench(mgps_comm *comm, size_t floats, int reps, double *us_per_exchange)
try {
    // device time of `reps` back-to-back send-to-self + receive-from-self groups of `floats` floats: what one
    // ghost exchange costs on this box before any link is involved (launch + RCCL's point-to-point kernel)
    if (!comm || comm->exchange != rcclExchange || floats == 0 || reps < 1 || !us_per_exchange) {
        setLastGlobalError("mgps_comm_rccl_selfbench: bad arguments");
        return MGPS_ERR_INVALID_ARGUMENT;
    }
    auto *s = static_cast<RcclState *>(comm->user);
    float *src = nullptr, *dst = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&src), floats * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&dst), floats * sizeof(float));
    if (e == hipSuccess) e = hipMemset(src, 0, floats * sizeof(float));
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    int rc = MGPS_OK;
    for (int pass = 0; pass < 2 && e == hipSuccess && rc == MGPS_OK; ++pass) {  // pass 0 warms up
        if (pass == 1) e = hipEventRecord(e0, s->own);
        for (int q = 0; q < (pass ? reps : 3) && rc == MGPS_OK; ++q) {
            ncclResult_t r = gApi.GroupStart();
            if (r == ncclSuccess) r = gApi.Send(src, floats * sizeof(float), ncclChar, s->rank, s->comm, s->own);
            if (r == ncclSuccess) r = gApi.Recv(dst, floats * sizeof(float), ncclChar, s->rank, s->comm, s->own);
            if (r == ncclSuccess) r = gApi.GroupEnd();
            if (r != ncclSuccess) {
                setLastGlobalError(std::string("mgps_comm_rccl_selfbench: ") + gApi.GetErrorString(r));
                rc = MGPS_ERR_COMM;
            }
        }
        if (pass == 1 && e == hipSuccess) e = hipEventRecord(e1, s->own);
        if (e == hipSuccess) e = hipStreamSynchronize(s->own);
    }
    float ms = 0.f;
    if (e == hipSuccess && rc == MGPS_OK) e = hipEventElapsedTime(&ms, e0, e1);
    *us_per_exchange = double(ms) * 1e3 / reps;
    (void)hipFree(src);
    (void)hipFree(dst);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (e != hipSuccess) {
        setLastGlobalError(std::string("mgps_comm_rccl_selfbench: ") + hipGetErrorString(e));
        return MGPS_ERR_HIP;
    }
    return rc;
}
MGPS_API_CATCH(nullptr)

void mgps_comm_destroy(mgps_comm *comm)
{
    if (comm && comm->destroy) comm->destroy(comm->user);
    if (comm) std::memset(comm, 0, sizeof(*comm));
}

}  // extern "C"
