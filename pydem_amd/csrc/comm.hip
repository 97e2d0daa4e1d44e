// comm.hip -- RCCL transport for the cross-tile edge strips of the directory flow.
//
// The reference moves edge strips between worker processes through a shared on-disk zarr store
// (pydem/process_manager.py:243-255, :362-381).  Here every rank packs the lines it owns into a
// zero-initialised device buffer laid out identically on all ranks and ONE in-place
// ncclAllReduce(sum) over xGMI leaves the full set of lines on every rank (disjoint fills: x + 0 is
// exact, NaN stays NaN).  Strips are KiB-sized, so the exchange is latency-bound; a single
// collective per step beats a fan of point-to-point messages on the xGMI mesh.
#include "internal.h"
#include <rccl/rccl.h>
#include <string.h>

struct pydem_comm {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int world = 1, rank = 0, device = 0;
    double *buf = nullptr;      // device staging buffer
    size_t cap = 0;             // in doubles
    hipEvent_t ev_clear = nullptr, ev_packed = nullptr;   // staging buffer cleared (comm stream) / lines packed (tile stream)
};

#define NCCL_TRY(expr)                                                                             \
    do {                                                                                           \
        ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess) {                                                                   \
            pydem_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, ncclGetErrorString(r_)); \
            return -6;                                                                             \
        }                                                                                          \
    } while (0)

namespace {

template <typename T>
__global__ void k_pack_line(const T *__restrict__ src, int64_t stride, int64_t count, double *__restrict__ dst)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += (int64_t)gridDim.x * blockDim.x)
        dst[k] = (double)src[k * stride];
}

}  // namespace

extern "C" {

int pydem_comm_unique_id(char *out128)
{
    static_assert(sizeof(ncclUniqueId) <= 128, "ncclUniqueId larger than expected");
    ncclUniqueId id;
    NCCL_TRY(ncclGetUniqueId(&id));
    memset(out128, 0, 128);
    memcpy(out128, &id, sizeof(id));
    return 0;
}

int pydem_comm_create(int world, int rank, const char *uid128, int device, pydem_comm **out)
{
    HIP_TRY(hipSetDevice(device));
    pydem_comm *c = new pydem_comm();
    c->world = world; c->rank = rank; c->device = device;
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_clear, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_packed, hipEventDisableTiming));
    ncclUniqueId id;
    memcpy(&id, uid128, sizeof(id));
    NCCL_TRY(ncclCommInitRank(&c->comm, world, id, rank));
    *out = c;
    return 0;
}

int pydem_comm_destroy(pydem_comm *c)
{
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->buf) (void)hipFree(c->buf);
    if (c->ev_clear) (void)hipEventDestroy(c->ev_clear);
    if (c->ev_packed) (void)hipEventDestroy(c->ev_packed);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

// make room for n doubles and clear them
int pydem_comm_begin(pydem_comm *c, int64_t n_doubles)
{
    HIP_TRY(hipSetDevice(c->device));
    if ((size_t)n_doubles > c->cap) {
        if (c->buf) HIP_TRY(hipFree(c->buf));
        HIP_TRY(hipMalloc((void **)&c->buf, (size_t)n_doubles * 8));
        c->cap = (size_t)n_doubles;
    }
    HIP_TRY(hipMemsetAsync(c->buf, 0, (size_t)n_doubles * 8, c->stream));
    HIP_TRY(hipEventRecord(c->ev_clear, c->stream));      // the packs wait for this on their own streams
    return 0;
}

static int pack_one(pydem_comm *c, pydem_tile *t, int field, int axis, int64_t index, int64_t offset)
{
    HIP_TRY(hipSetDevice(t->device));
    if (t->device != c->device) { pydem_set_error("pydem_comm_pack_line: tile and communicator live on different devices"); return -2; }
    const int64_t lim = axis == 0 ? t->n : t->m;
    if (index < 0) index += lim;
    if (index < 0 || index >= lim) { pydem_set_error("line index out of range"); return -2; }
    const int64_t count = axis == 0 ? t->m : t->n;
    if ((size_t)(offset + count) > c->cap) { pydem_set_error("pydem_comm_pack_line: staging buffer too small"); return -2; }
    const int64_t stride = axis == 0 ? 1 : t->m;
    const int64_t first = axis == 0 ? index * t->m : index;
    const int g = (int)(cdiv(count, 256) < 64 ? cdiv(count, 256) : 64);
    double *dst = c->buf + offset;
    switch (field) {
        case PYDEM_ELEV: case PYDEM_MAG: case PYDEM_DIRECTION: case PYDEM_PROPORTION: case PYDEM_UCA: case PYDEM_TWI: {
            const double *src = field == PYDEM_ELEV ? t->elev : field == PYDEM_MAG ? t->mag : field == PYDEM_DIRECTION ? t->dir
                              : field == PYDEM_PROPORTION ? t->prop : field == PYDEM_UCA ? t->uca : t->twi;
            if (!src || !t->have[field]) { pydem_set_error("field %d not available", field); return -3; }
            hipLaunchKernelGGL(k_pack_line<double>, dim3(g), dim3(256), 0, t->stream, src + first, stride, count, dst);
            break;
        }
        case PYDEM_FLATS: case PYDEM_EDGE_TODO: case PYDEM_EDGE_DONE: {
            const uint8_t *src = field == PYDEM_FLATS ? t->flats : field == PYDEM_EDGE_TODO ? t->edge_todo : t->edge_done;
            if (!src || !t->have[field]) { pydem_set_error("field %d not available", field); return -3; }
            hipLaunchKernelGGL(k_pack_line<uint8_t>, dim3(g), dim3(256), 0, t->stream, src + first, stride, count, dst);
            break;
        }
        default: pydem_set_error("pydem_comm_pack_line: unsupported field %d", field); return -2;
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// one row / column of a resident field -> doubles at buf[offset ...] (device to device)
int pydem_comm_pack_line(pydem_comm *c, pydem_tile *t, int field, int axis, int64_t index, int64_t offset)
{
    HIP_TRY(hipSetDevice(t->device));
    HIP_TRY(hipStreamWaitEvent(t->stream, c->ev_clear, 0));
    PYDEM_TRY(pack_one(c, t, field, axis, index, offset));
    HIP_TRY(hipStreamSynchronize(t->stream));
    return 0;
}

// several lines of one tile, no host synchronisation: the pack kernels wait (event) for the clear of the staging
// buffer on the communicator's stream, and the next collective on that stream waits (event) for the packs
int pydem_comm_pack_lines(pydem_comm *c, pydem_tile *t, int count, const int *fields, const int *axes, const int64_t *indices,
                          const int64_t *offsets)
{
    HIP_TRY(hipSetDevice(t->device));
    HIP_TRY(hipStreamWaitEvent(t->stream, c->ev_clear, 0));
    for (int k = 0; k < count; k++) PYDEM_TRY(pack_one(c, t, fields[k], axes[k], indices[k], offsets[k]));
    HIP_TRY(hipEventRecord(c->ev_packed, t->stream));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_packed, 0));
    return 0;
}

// in-place all-reduce of the first n doubles of the staging buffer (op 0 = sum, 1 = max), result copied to host
int pydem_comm_allreduce(pydem_comm *c, int64_t n_doubles, int op, double *host_out)
{
    HIP_TRY(hipSetDevice(c->device));
    if ((size_t)n_doubles > c->cap) { pydem_set_error("pydem_comm_allreduce: staging buffer too small"); return -2; }
    NCCL_TRY(ncclAllReduce(c->buf, c->buf, (size_t)n_doubles, ncclDouble, op == 1 ? ncclMax : ncclSum, c->comm, c->stream));   // also for world == 1
    if (host_out) HIP_TRY(hipMemcpyAsync(host_out, c->buf, (size_t)n_doubles * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

// host values -> staging buffer (for scalar reductions such as the timing max / barrier)
int pydem_comm_put(pydem_comm *c, const double *host_in, int64_t n_doubles, int64_t offset)
{
    HIP_TRY(hipSetDevice(c->device));
    if ((size_t)(offset + n_doubles) > c->cap) { pydem_set_error("pydem_comm_put: staging buffer too small"); return -2; }
    HIP_TRY(hipMemcpyAsync(c->buf + offset, host_in, (size_t)n_doubles * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

}  // extern "C"
