// tile.hip -- handle management, spacing tables, upload/download and the extern "C" surface.
#include "internal.h"
#include <cmath>
#include <math.h>
#include <stdarg.h>
#include <string.h>
#include <stdlib.h>

static thread_local char g_err[1024] = "";

void pydem_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- planes of destroyed tiles are kept for the next tile of the same shape ------------------------------------------
// A fresh DEMProcessor per elevation file is how the reference is used, and a 16384^2 tile is a dozen 0.25-2 GiB planes:
// mapping them anew costs the drop-in call more than its kernels (measured: ~240 ms of hipMalloc inside a 54-ms calc_twi).
// Blocks of 1 MiB and more that tile_alloc handed out go on a per-device free list when their tile is destroyed and are
// handed out again on an exact size match (the contents are whatever the last owner left: no stage relies on fresh
// memory).  PYDEM_PLANE_CACHE_GB bounds what is kept per device (default 32, 0 = off; the least recently returned blocks are evicted first); pydem_hip_release_scratch and a
// failing hipMalloc empty the lists.
#include <map>
#include <mutex>
#include <unordered_map>
namespace {
struct FreeBlock { void *p; uint64_t stamp; };      // stamp: when the block was given back (the oldest is evicted first)
struct PlaneCache {
    std::mutex mu;
    std::multimap<size_t, FreeBlock> free_blocks;
    uint64_t clock = 0;
    std::unordered_map<void *, size_t> live;          // blocks handed out by tile_alloc (size known at destroy time)
    size_t kept = 0;
};
std::mutex g_pc_table;
std::map<int, PlaneCache *> g_pc;
PlaneCache *plane_cache(int device)
{
    std::lock_guard<std::mutex> g(g_pc_table);
    PlaneCache *&c = g_pc[device];
    if (!c) c = new PlaneCache();
    return c;
}
size_t plane_cache_limit()
{
    // default 32 GiB: the planes of one 16384^2 tile (24.7 GB) -- what the drop-in pattern "a fresh DEMProcessor per file" reuses;
    // fractions of a GB are honoured, a negative value means 0 (off)
    static const size_t lim = [] {
        const char *e = getenv("PYDEM_PLANE_CACHE_GB");
        const double gb = e ? atof(e) : 32.0;
        return (size_t)((gb > 0.0 ? gb : 0.0) * (double)((size_t)1 << 30));
    }();
    return lim;
}
// (c->mu held) take the least recently returned blocks off the lists until `room` more bytes fit under the limit; the caller
// frees them AFTER it has let go of the lock (hipFree synchronises the device)
void plane_cache_evict(PlaneCache *c, size_t room, std::vector<void *> &victims)
{
    const size_t lim = plane_cache_limit();
    while (!c->free_blocks.empty() && c->kept + room > lim) {
        auto old = c->free_blocks.begin();
        for (auto it = c->free_blocks.begin(); it != c->free_blocks.end(); ++it) if (it->second.stamp < old->second.stamp) old = it;
        victims.push_back(old->second.p);
        c->kept -= old->first;
        c->free_blocks.erase(old);
    }
}
constexpr size_t PLANE_MIN = (size_t)1 << 20;
void plane_cache_flush(int device)
{
    PlaneCache *c = plane_cache(device);
    std::lock_guard<std::mutex> g(c->mu);
    for (auto &kv : c->free_blocks) (void)hipFree(kv.second.p);
    c->free_blocks.clear(); c->kept = 0;
}
}  // namespace

void *plane_take(int device, size_t bytes)
{
    std::vector<void *> victims;                     // (PYDEM_PLANE_EVICT_ON_MISS only; freed once the lock is released)
    if (bytes >= PLANE_MIN && plane_cache_limit()) {
        PlaneCache *c = plane_cache(device);
        std::lock_guard<std::mutex> g(c->mu);
        auto it = c->free_blocks.find(bytes);
        if (it != c->free_blocks.end()) {
            void *q = it->second.p;
            c->free_blocks.erase(it); c->kept -= bytes;
            c->live[q] = bytes;
            return q;
        }
        // (a size the lists do not hold: nothing is evicted here -- the blocks on the lists are those of the tile destroyed
        // last, i.e. what the tile being built asks for next; dead sizes leave when plane_give needs their room, and a failing
        // hipMalloc empties the lists.  PYDEM_PLANE_EVICT_ON_MISS=1: the round-5 behaviour, for A/B runs)
        static const bool on_miss = [] { const char *e = getenv("PYDEM_PLANE_EVICT_ON_MISS"); return e && atoi(e) > 0; }();
        if (on_miss) plane_cache_evict(c, bytes, victims);
    }
    for (void *v : victims) (void)hipFree(v);
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, bytes);
    {   // PYDEM_PLANE_DEBUG=1: one line per block that is mapped anew (a steady state maps nothing)
        static const bool dbg = [] { const char *d = getenv("PYDEM_PLANE_DEBUG"); return d && atoi(d) > 0; }();
        if (dbg) fprintf(stderr, "[pydem] plane_take: hipMalloc(%zu)\n", bytes);
    }
    if (e != hipSuccess) {                          // the free lists may hold what is missing
        (void)hipGetLastError();
        plane_cache_flush(device);
        e = hipMalloc(&q, bytes);
    }
    if (e != hipSuccess) { pydem_set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e)); return nullptr; }
    if (bytes >= PLANE_MIN && plane_cache_limit()) {
        PlaneCache *c = plane_cache(device);
        std::lock_guard<std::mutex> g(c->mu);
        c->live[q] = bytes;
    }
    return q;
}

void plane_give(int device, void *q)
{
    if (!q) return;
    PlaneCache *c = plane_cache(device);
    std::vector<void *> victims;
    bool kept = false;
    {
        std::lock_guard<std::mutex> g(c->mu);
        auto it = c->live.find(q);
        if (it != c->live.end()) {
            const size_t bytes = it->second;
            c->live.erase(it);
            if (bytes <= plane_cache_limit()) {
                plane_cache_evict(c, bytes, victims);
                c->free_blocks.emplace(bytes, FreeBlock{q, ++c->clock}); c->kept += bytes;
                kept = true;
            }
        }
    }
    for (void *v : victims) (void)hipFree(v);
    if (!kept) (void)hipFree(q);
}

// hipMalloc for everything else the library maps on a device (scratch, arenas, record buffers): when it fails the free lists
// of the current device are emptied and the call repeated
hipError_t dev_malloc(void **p, size_t bytes)
{
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
        int device = 0;
        (void)hipGetLastError();
        if (hipGetDevice(&device) == hipSuccess) { plane_cache_flush(device); e = hipMalloc(p, bytes); }
    }
    return e;
}

template <typename T>
int tile_alloc(pydem_tile *t, T **p, size_t count)
{
    if (*p) return 0;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);
    void *q = plane_take(t->device, bytes);
    if (!q) return -1;
    t->device_bytes += (int64_t)bytes;
    *p = (T *)q;
    return 0;
}
template int tile_alloc<double>(pydem_tile *, double **, size_t);
template int tile_alloc<uint8_t>(pydem_tile *, uint8_t **, size_t);
template int tile_alloc<int8_t>(pydem_tile *, int8_t **, size_t);
template int tile_alloc<int32_t>(pydem_tile *, int32_t **, size_t);
template int tile_alloc<RowTab>(pydem_tile *, RowTab **, size_t);
template int tile_alloc<uint16_t>(pydem_tile *, uint16_t **, size_t);

// ---- per-device scratch arena of the conditioning stages (internal.h)
#include <map>
#include <time.h>
#include <mutex>
namespace {
struct DevArena { std::mutex busy; void *p = nullptr; size_t bytes = 0; };
std::mutex g_arena_table;
std::map<int, DevArena *> g_arenas;
DevArena *arena_of(int device)
{
    std::lock_guard<std::mutex> g(g_arena_table);
    DevArena *&a = g_arenas[device];
    if (!a) a = new DevArena();
    return a;
}
}  // namespace

int arena_acquire(int device, ArenaLease *L)
{
    DevArena *a = arena_of(device);
    a->busy.lock();
    L->device = device; L->arena = a; L->base = (char *)a->p; L->bytes = a->bytes; L->off = 0; L->want = 0; L->held = true;
    return 0;
}

void *arena_take(ArenaLease *L, size_t bytes)
{
    const size_t need = (bytes + 255) & ~(size_t)255;
    L->want += need ? need : 256;
    if (L->base && L->off + need <= L->bytes) { void *q = L->base + L->off; L->off += need ? need : 256; return q; }
    void *q = nullptr;
    const hipError_t e = dev_malloc(&q, need ? need : 256);
    if (e != hipSuccess) { pydem_set_error("hipMalloc(%zu bytes) failed: %s", need, hipGetErrorString(e)); return nullptr; }
    L->extra.push_back(q);
    return q;
}

ArenaLease::~ArenaLease()
{
    if (!held) return;
    DevArena *a = static_cast<DevArena *>(arena);
    (void)hipSetDevice(device);
    for (void *q : extra) (void)hipFree(q);
    if (want > a->bytes) {              // next time everything fits
        if (a->p) (void)hipFree(a->p);
        a->p = nullptr; a->bytes = 0;
        void *q = nullptr;
        if (dev_malloc(&q, want + want / 8) == hipSuccess) { a->p = q; a->bytes = want + want / 8; }
    }
    a->busy.unlock();
}

namespace {

template <typename T>
__global__ void k_line_gather(const T *__restrict__ src, int64_t stride, int64_t count, T *__restrict__ dst)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += (int64_t)gridDim.x * blockDim.x) dst[k] = src[k * stride];
}
template <typename T>
__global__ void k_line_scatter(const T *__restrict__ src, int64_t stride, int64_t count, T *__restrict__ dst)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += (int64_t)gridDim.x * blockDim.x) dst[k * stride] = src[k];
}

__global__ void k_restore_pit_slopes(const int32_t *__restrict__ src, int64_t n, double *mag)
{
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
        if (src[e] >= 0) mag[src[e]] = -1.0;      // unused output slots hold -1
}

template <typename S>
__global__ void k_convert_to_f64(const S *__restrict__ src, double *__restrict__ dst, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = (double)src[i];
}

__global__ void k_find_flats(const double *__restrict__ mag, uint8_t *__restrict__ flats, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) flats[i] = (mag[i] == -1.0);    // dem_processing.py:305-306
}

size_t dtype_size(int dt)
{
    switch (dt) {
        case PYDEM_F64: return 8;
        case PYDEM_F32: return 4;
        case PYDEM_I16: return 2;
        case PYDEM_I32: return 4;
        case PYDEM_U8: return 1;
        case PYDEM_I8: return 1;
    }
    return 0;
}

int field_ptr(pydem_tile *t, int field, void ***pp, size_t *elem)
{
    switch (field) {
        case PYDEM_ELEV: *pp = (void **)&t->elev; *elem = 8; return 0;
        case PYDEM_MAG: *pp = (void **)&t->mag; *elem = 8; return 0;
        case PYDEM_DIRECTION: *pp = (void **)&t->dir; *elem = 8; return 0;
        case PYDEM_FLATS: *pp = (void **)&t->flats; *elem = 1; return 0;
        case PYDEM_SECTION: *pp = (void **)&t->section; *elem = 1; return 0;
        case PYDEM_PROPORTION: *pp = (void **)&t->prop; *elem = 8; return 0;
        case PYDEM_UCA: *pp = (void **)&t->uca; *elem = 8; return 0;
        case PYDEM_TWI: *pp = (void **)&t->twi; *elem = 8; return 0;
        case PYDEM_EDGE_TODO: *pp = (void **)&t->edge_todo; *elem = 1; return 0;
        case PYDEM_EDGE_DONE: *pp = (void **)&t->edge_done; *elem = 1; return 0;
    }
    pydem_set_error("unknown field id %d", field);
    return -2;
}

int ensure_field(pydem_tile *t, int field)
{
    void **pp; size_t elem;
    PYDEM_TRY(field_ptr(t, field, &pp, &elem));
    if (*pp) return 0;
    if (elem == 8) return tile_alloc(t, (double **)pp, (size_t)t->NN);
    return tile_alloc(t, (uint8_t **)pp, (size_t)t->NN);
}

}  // namespace

int tile_pinned(pydem_tile *t, size_t bytes, void **out)
{
    if (t->h_stage_bytes < bytes) {
        if (t->h_stage) { (void)hipHostFree(t->h_stage); t->h_stage = nullptr; t->h_stage_bytes = 0; }
        const size_t want = bytes + bytes / 4 + 4096;
        HIP_TRY(hipHostMalloc(&t->h_stage, want, hipHostMallocDefault));
        t->h_stage_bytes = want;
    }
    *out = t->h_stage;
    return 0;
}

// ---- whole-plane transfers between pageable host arrays and the device ------------------------------------------------
// A plain hipMemcpy of a pageable array pins the caller's pages on the fly: measured here at ~3 GB/s for a fresh 2 GiB array
// (0.7 s for the 16384^2 elevations).  Planes of XFER_MIN bytes or more go through per-device pinned chunks instead: XFER_T
// host threads each copy their chunks between the caller's array and their two pinned buffers while their own stream moves
// the previous chunk over PCIe (the threads also take the first-touch page faults of a fresh destination array in
// parallel).  PYDEM_XFER_THREADS=0 restores the plain copy.
#include <thread>
#include <sys/mman.h>
namespace {
constexpr size_t XFER_CHUNK = (size_t)8 << 20;
constexpr size_t XFER_MIN = (size_t)32 << 20;
constexpr int XFER_TMAX = 16;
struct XferLane { void *pin[2] = {nullptr, nullptr}; hipStream_t stream = nullptr; hipEvent_t ev[2] = {nullptr, nullptr}; };
struct XferPool { std::mutex busy; XferLane lane[XFER_TMAX]; int ready = 0; };
std::mutex g_xfer_table;
std::map<int, XferPool *> g_xfer;

int xfer_threads()
{
    static const int n = [] {
        const char *e = getenv("PYDEM_XFER_THREADS");
        int v = e ? atoi(e) : 8;
        return v < 0 ? 0 : (v > XFER_TMAX ? XFER_TMAX : v);
    }();
    return n;
}

int xfer_pool(int device, int T, XferPool **out)
{
    XferPool *P;
    {
        std::lock_guard<std::mutex> g(g_xfer_table);
        XferPool *&p = g_xfer[device];
        if (!p) p = new XferPool();
        P = p;
    }
    *out = P;
    return 0;
}

int xfer_prepare(XferPool *P, int T)      // under P->busy
{
    for (int k = P->ready; k < T; ++k) {
        XferLane &L = P->lane[k];
        for (int b = 0; b < 2; ++b) {
            if (!L.pin[b]) HIP_TRY(hipHostMalloc(&L.pin[b], XFER_CHUNK, hipHostMallocDefault));
            if (!L.ev[b]) HIP_TRY(hipEventCreateWithFlags(&L.ev[b], hipEventDisableTiming));
        }
        if (!L.stream) HIP_TRY(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
        P->ready = k + 1;
    }
    return 0;
}

// one lane's share of the plane: chunks k, k + T, ...
hipError_t xfer_lane_run(int device, XferLane &L, char *dev, char *host, size_t bytes, int k, int T, bool to_host)
{
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return e;
    const size_t nchunk = (bytes + XFER_CHUNK - 1) / XFER_CHUNK;
    size_t prev_off = 0, prev_len = 0; int prev_b = -1;
    int it = 0;
    for (size_t c = (size_t)k; c < nchunk; c += (size_t)T, ++it) {
        const int b = it & 1;
        const size_t off = c * XFER_CHUNK, len = (bytes - off < XFER_CHUNK) ? bytes - off : XFER_CHUNK;
        if (!to_host) {
            if (it >= 2 && (e = hipEventSynchronize(L.ev[b])) != hipSuccess) return e;     // the buffer's previous chunk has left
            memcpy(L.pin[b], host + off, len);
            if ((e = hipMemcpyAsync(dev + off, L.pin[b], len, hipMemcpyHostToDevice, L.stream)) != hipSuccess) return e;
            if ((e = hipEventRecord(L.ev[b], L.stream)) != hipSuccess) return e;
        } else {
            if ((e = hipMemcpyAsync(L.pin[b], dev + off, len, hipMemcpyDeviceToHost, L.stream)) != hipSuccess) return e;
            if ((e = hipEventRecord(L.ev[b], L.stream)) != hipSuccess) return e;
            if (prev_b >= 0) {
                if ((e = hipEventSynchronize(L.ev[prev_b])) != hipSuccess) return e;
                memcpy(host + prev_off, L.pin[prev_b], prev_len);
            }
            prev_b = b; prev_off = off; prev_len = len;
        }
    }
    if (to_host && prev_b >= 0) {
        if ((e = hipEventSynchronize(L.ev[prev_b])) != hipSuccess) return e;
        memcpy(host + prev_off, L.pin[prev_b], prev_len);
    }
    return hipStreamSynchronize(L.stream);
}
}  // namespace

// copy `bytes` between a device plane and a (pageable) host array; the tile's stream is drained first and the call returns
// with the transfer complete
int tile_plane_copy(pydem_tile *t, void *dev, void *host, size_t bytes, bool to_host)
{
    const int T = xfer_threads();
    if (T == 0 || bytes < XFER_MIN) {
        if (to_host) HIP_TRY(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, t->stream));
        else HIP_TRY(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        return 0;
    }
    HIP_TRY(hipStreamSynchronize(t->stream));          // what the plane holds (or what still reads it) is settled
    if (to_host) {
        // a fresh destination array is faulted in by the copy threads: ask for huge pages where the kernel gives them on request
        const uintptr_t a = ((uintptr_t)host + 4095) & ~(uintptr_t)4095, b = ((uintptr_t)host + bytes) & ~(uintptr_t)4095;
        if (b > a) (void)madvise((void *)a, b - a, MADV_HUGEPAGE);
    }
    XferPool *P;
    PYDEM_TRY(xfer_pool(t->device, T, &P));
    std::lock_guard<std::mutex> g(P->busy);
    PYDEM_TRY(xfer_prepare(P, T));
    hipError_t err[XFER_TMAX];
    std::thread th[XFER_TMAX];
    for (int k = 1; k < T; ++k)
        th[k] = std::thread([&, k] { err[k] = xfer_lane_run(t->device, P->lane[k], (char *)dev, (char *)host, bytes, k, T, to_host); });
    err[0] = xfer_lane_run(t->device, P->lane[0], (char *)dev, (char *)host, bytes, 0, T, to_host);
    for (int k = 1; k < T; ++k) th[k].join();
    for (int k = 0; k < T; ++k)
        if (err[k] != hipSuccess) { pydem_set_error("plane transfer: %s", hipGetErrorString(err[k])); return -1; }
    return 0;
}

int ensure_fields(pydem_tile *t, std::initializer_list<int> fields)
{
    for (int f : fields) PYDEM_TRY(ensure_field(t, f));
    return 0;
}

extern "C" {

const char *pydem_hip_last_error(void) { return g_err; }

int pydem_hip_release_scratch(void)
{
    // lock order: never `busy` under the table lock (a lease holds `busy` for the length of a conditioning stage; records are
    // never deleted, so the snapshot stays valid)
    std::vector<std::pair<int, DevArena *>> all;
    {
        std::lock_guard<std::mutex> g(g_arena_table);
        for (auto &kv : g_arenas) all.push_back(kv);
    }
    for (auto &kv : all) {
        std::lock_guard<std::mutex> b(kv.second->busy);
        if (kv.second->p) { (void)hipSetDevice(kv.first); (void)hipFree(kv.second->p); kv.second->p = nullptr; kv.second->bytes = 0; }
    }
    std::vector<int> devs;
    {
        std::lock_guard<std::mutex> g(g_pc_table);
        for (auto &kv : g_pc) devs.push_back(kv.first);
    }
    for (int d : devs) { (void)hipSetDevice(d); plane_cache_flush(d); }
    // ... and the pinned chunks / streams of the whole-plane transfers (they come back with the next large transfer)
    std::vector<std::pair<int, XferPool *>> pools;
    {
        std::lock_guard<std::mutex> g(g_xfer_table);
        for (auto &kv : g_xfer) pools.push_back(kv);
    }
    for (auto &kv : pools) {
        std::lock_guard<std::mutex> b(kv.second->busy);
        (void)hipSetDevice(kv.first);
        for (int k = 0; k < kv.second->ready; k++) {
            XferLane &L = kv.second->lane[k];
            for (int q = 0; q < 2; q++) {
                if (L.pin[q]) { (void)hipHostFree(L.pin[q]); L.pin[q] = nullptr; }
                if (L.ev[q]) { (void)hipEventDestroy(L.ev[q]); L.ev[q] = nullptr; }
            }
            if (L.stream) { (void)hipStreamDestroy(L.stream); L.stream = nullptr; }
        }
        kv.second->ready = 0;
    }
    return 0;
}

int pydem_hip_device_count(int *count)
{
    HIP_TRY(hipGetDeviceCount(count));
    return 0;
}

int pydem_hip_device_memory(int device, int64_t *free_bytes, int64_t *total_bytes)
{
    size_t f = 0, tot = 0;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemGetInfo(&f, &tot));
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)tot;
    return 0;
}

int pydem_hip_device_name(int device, char *buf, int buflen)
{
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, device));
    snprintf(buf, buflen, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return 0;
}

int pydem_tile_create(int64_t n_rows, int64_t n_cols, int device, pydem_tile **out)
{
    if (n_rows < 3 || n_cols < 3) { pydem_set_error("tile must be at least 3x3 (got %lld x %lld)", (long long)n_rows, (long long)n_cols); return -2; }
    if (n_rows * n_cols >= (int64_t)INT32_MAX) { pydem_set_error("tile too large for int32 cell ids"); return -2; }
    HIP_TRY(hipSetDevice(device));
    pydem_tile *t = new pydem_tile();
    t->n = n_rows; t->m = n_cols; t->NN = n_rows * n_cols; t->device = device;
    HIP_TRY(hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&t->stream2, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&t->ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&t->ev_join, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&t->ev_snap, hipEventDisableTiming));
    for (int i = 0; i < 8; i++) HIP_TRY(hipEventCreate(&t->ev[i]));
    PYDEM_TRY(tile_alloc(t, &t->counters, 64));
    HIP_TRY(hipHostMalloc((void **)&t->h_counters, 64 * sizeof(int32_t), hipHostMallocDefault));
    *out = t;
    return 0;
}

int pydem_tile_destroy(pydem_tile *t)
{
    if (!t) return 0;
    (void)hipSetDevice(t->device);
    (void)hipStreamSynchronize(t->stream);
    // (a stage that returned early between the fork and the join of the side stream may have left kernels there that still
    // write edge_todo / todo_work / prop: the planes go to the free lists -- i.e. to the next tile -- only once both streams are idle)
    if (t->stream2) (void)hipStreamSynchronize(t->stream2);
    void *ptrs[] = {t->elev, t->mag, t->dir, t->prop, t->uca, t->twi, t->flats, t->edge_todo, t->edge_done,
                    t->flat0, t->section, t->dX, t->dY, t->dX2, t->dY2, t->rowtab, t->sec_theta, t->row_area, t->inmask,
                    t->gflags, t->todo_work, t->indeg, t->queue[0], t->queue[1], t->labels, t->flatlist,
                    t->counters, t->scratch, t->pits.src, t->pits.dst, t->pits.w, t->pits.in_src,
                    t->pits.in_dst, t->pits.in_w, t->pits.raw_src, t->pits.raw_dst, t->pits.raw_w,
                    t->estamp, t->edelta, t->p_delta, t->s_data, t->p_flags, t->s_flags, t->line_stage, t->contrib,
                    t->eseed, t->lines_stage, t->pits.sort_buf, t->nd_rec, t->cond_mem, t->cb_mem[0], t->cb_mem[1], t->cb_mem[2]};
    for (void *p : ptrs) if (p) plane_give(t->device, p);          // (blocks that did not come from tile_alloc are freed)
    if (t->h_counters) (void)hipHostFree(t->h_counters);
    if (t->h_strip_d) (void)hipHostFree(t->h_strip_d);
    if (t->h_stage) (void)hipHostFree(t->h_stage);
    if (t->h_strip_f) (void)hipHostFree(t->h_strip_f);
    for (int i = 0; i < 8; i++) if (t->ev[i]) (void)hipEventDestroy(t->ev[i]);
    if (t->ev_fork) (void)hipEventDestroy(t->ev_fork);
    if (t->ev_join) (void)hipEventDestroy(t->ev_join);
    if (t->ev_snap) (void)hipEventDestroy(t->ev_snap);
    if (t->stream2) { (void)hipStreamSynchronize(t->stream2); (void)hipStreamDestroy(t->stream2); }
    if (t->stream) (void)hipStreamDestroy(t->stream);
    delete t;
    return 0;
}

int pydem_tile_set_spacing(pydem_tile *t, const double *dX, const double *dY, const double *dX2, const double *dY2)
{
    HIP_TRY(hipSetDevice(t->device));
    const int64_t n = t->n;
    // the facet tests compare slopes by cross-multiplication with the spacings and divide with host reciprocals: both
    // assume finite, strictly positive cell sizes (the reference would run on with negative or zero ones and return
    // mirrored / infinite slopes -- not reproduced, so refuse)
    for (int64_t r = 0; r < n - 1; r++)
        if (!(dX[r] > 0 && dY[r] > 0 && std::isfinite(dX[r]) && std::isfinite(dY[r]))) {
            pydem_set_error("pydem_tile_set_spacing: dX / dY must be finite and > 0 (row %lld: %g, %g)", (long long)r, dX[r], dY[r]);
            return -2;
        }
    for (int64_t r = 0; r < n; r++)      // the sweep carries the edge_todo taint in the sign of a cell's shares: areas must be positive
        if (!(dX2[r] * dY2[r] > 0 && std::isfinite(dX2[r] * dY2[r]))) {
            pydem_set_error("pydem_tile_set_spacing: dX2 * dY2 (cell area) must be finite and > 0 (row %lld: %g, %g)", (long long)r, dX2[r], dY2[r]);
            return -2;
        }
    t->h_dX.assign(dX, dX + n - 1); t->h_dY.assign(dY, dY + n - 1);
    t->h_dX2.assign(dX2, dX2 + n); t->h_dY2.assign(dY2, dY2 + n);
    std::vector<RowTab> tab((size_t)(n - 1));
    int exotic = 0;
    for (int64_t r = 0; r < n - 1; r++) {
        RowTab &e = tab[(size_t)r];
        e.dX = dX[r]; e.dY = dY[r];
        e.hyp = sqrt(dX[r] * dX[r] + dY[r] * dY[r]);   // np.sqrt(d1**2 + d2**2), dem_processing.py:1962
        e.thA = atan2(dY[r], dX[r]);                    // np.arctan2(d2, d1), :1936 (host libm == numpy)
        e.thB = atan2(dX[r], dY[r]);
        e.rdX = 1.0 / e.dX; e.rdY = 1.0 / e.dY; e.rhyp = 1.0 / e.hyp;
        // the stencil's mask path assumes that no slope quotient can overflow into a NaN: elevations are tested per row,
        // the spacings here
        if (!(e.dX > 0x1p-500 && e.dX < 0x1p500 && e.dY > 0x1p-500 && e.dY < 0x1p500 && e.hyp < 0x1p500)) exotic = 1;
    }
    { const char *ev = getenv("PYDEM_STENCIL_EXACT"); if (ev && *ev == '1') exotic = 1; }
    t->stencil_exact_only = exotic;
    // theta per row for section/proportion: facet-0 spacing of rows 1..n-2 with the first and last
    // entries duplicated (dem_processing.py:1031-1033)
    std::vector<double> st((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        int64_t r = i - 1;
        if (r < 0) r = 0;
        if (r > n - 3) r = n - 3;
        st[(size_t)i] = atan2(dY[r], dX[r]);
    }
    PYDEM_TRY(tile_alloc(t, &t->dX, (size_t)n)); PYDEM_TRY(tile_alloc(t, &t->dY, (size_t)n));
    PYDEM_TRY(tile_alloc(t, &t->dX2, (size_t)n)); PYDEM_TRY(tile_alloc(t, &t->dY2, (size_t)n));
    PYDEM_TRY(tile_alloc(t, &t->rowtab, (size_t)n)); PYDEM_TRY(tile_alloc(t, &t->sec_theta, (size_t)n));
    HIP_TRY(hipMemcpyAsync(t->dX, dX, (n - 1) * 8, hipMemcpyHostToDevice, t->stream));
    HIP_TRY(hipMemcpyAsync(t->dY, dY, (n - 1) * 8, hipMemcpyHostToDevice, t->stream));
    HIP_TRY(hipMemcpyAsync(t->dX2, dX2, n * 8, hipMemcpyHostToDevice, t->stream));
    HIP_TRY(hipMemcpyAsync(t->dY2, dY2, n * 8, hipMemcpyHostToDevice, t->stream));
    HIP_TRY(hipMemcpyAsync(t->rowtab, tab.data(), (n - 1) * sizeof(RowTab), hipMemcpyHostToDevice, t->stream));
    HIP_TRY(hipMemcpyAsync(t->sec_theta, st.data(), n * 8, hipMemcpyHostToDevice, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    t->spacing_set = true;
    return 0;
}

int pydem_tile_upload(pydem_tile *t, int field, const void *src, int dtype)
{
    HIP_TRY(hipSetDevice(t->device));
    // incremental edge rounds keep the deltas of cells below unresolved inlets pending until the flush: an upload between
    // two rounds must not discard them silently (a new UCA plane replaces what they would have been added to)
    if (t->einc_ready && field != PYDEM_UCA) PYDEM_TRY(stage_edge_flush(t));
    t->edge_clean = false;
    t->einc_ready = false;
    void **pp; size_t elem;
    PYDEM_TRY(field_ptr(t, field, &pp, &elem));
    PYDEM_TRY(ensure_field(t, field));
    const size_t ds = dtype_size(dtype);
    if (!ds) { pydem_set_error("unknown dtype %d", dtype); return -2; }
    if (ds == elem && (elem == 1 || dtype == PYDEM_F64)) {
        PYDEM_TRY(tile_plane_copy(t, *pp, const_cast<void *>(src), (size_t)t->NN * elem, false));
    } else {
        if (elem != 8) { pydem_set_error("dtype conversion is only supported for float64 fields"); return -2; }
        const size_t bytes = (size_t)t->NN * ds;
        if (t->scratch_bytes < bytes) {
            if (t->scratch) { HIP_TRY(hipFree(t->scratch)); t->device_bytes -= (int64_t)t->scratch_bytes; }
            HIP_TRY(dev_malloc(&t->scratch, bytes));
            t->scratch_bytes = bytes; t->device_bytes += (int64_t)bytes;
        }
        PYDEM_TRY(tile_plane_copy(t, t->scratch, const_cast<void *>(src), bytes, false));
        const int grid = (int)(cdiv(t->NN, 256) < 8192 ? cdiv(t->NN, 256) : 8192);
        double *dst = (double *)*pp;
        switch (dtype) {
            case PYDEM_F32: hipLaunchKernelGGL(k_convert_to_f64<float>, dim3(grid), dim3(256), 0, t->stream, (const float *)t->scratch, dst, t->NN); break;
            case PYDEM_I16: hipLaunchKernelGGL(k_convert_to_f64<int16_t>, dim3(grid), dim3(256), 0, t->stream, (const int16_t *)t->scratch, dst, t->NN); break;
            case PYDEM_I32: hipLaunchKernelGGL(k_convert_to_f64<int32_t>, dim3(grid), dim3(256), 0, t->stream, (const int32_t *)t->scratch, dst, t->NN); break;
            case PYDEM_U8: hipLaunchKernelGGL(k_convert_to_f64<uint8_t>, dim3(grid), dim3(256), 0, t->stream, (const uint8_t *)t->scratch, dst, t->NN); break;
            case PYDEM_I8: hipLaunchKernelGGL(k_convert_to_f64<int8_t>, dim3(grid), dim3(256), 0, t->stream, (const int8_t *)t->scratch, dst, t->NN); break;
            default: pydem_set_error("bad dtype"); return -2;
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(t->stream));
    }
    t->have[field] = true;
    if (field == PYDEM_ELEV) { t->elev_f32 = (dtype == PYDEM_F32); t->elev_dtype = dtype; }
    if (field == PYDEM_ELEV || field == PYDEM_MAG || field == PYDEM_DIRECTION || field == PYDEM_FLATS) t->graph_valid = false;
    return 0;
}

int pydem_tile_download(pydem_tile *t, int field, void *dst)
{
    HIP_TRY(hipSetDevice(t->device));
    void **pp; size_t elem;
    PYDEM_TRY(field_ptr(t, field, &pp, &elem));
    if (!*pp || !t->have[field]) { pydem_set_error("field %d has not been computed or uploaded", field); return -3; }
    if (field == PYDEM_UCA) PYDEM_TRY(stage_edge_flush(t));     // incremental edge rounds: settle what is still waiting upstream
    if (field == PYDEM_EDGE_DONE || field == PYDEM_EDGE_TODO) PYDEM_TRY(stage_edge_catchup(t));   // condensed rounds: the interior masks catch up
    PYDEM_TRY(tile_plane_copy(t, *pp, dst, (size_t)t->NN * elem, true));
    return 0;
}

// one row (axis 0) or one column (axis 1) of a field: the strips the directory flow exchanges
// between neighbouring tiles (reference process_manager.py:131-145, :252-255).  Columns are
// gathered / scattered by a kernel through a contiguous staging buffer (a strided 2-D memcpy of
// 1- or 8-byte rows costs one DMA descriptor per element).
static int line_copy(pydem_tile *t, int field, int axis, int64_t index, void *host, bool to_host)
{
    HIP_TRY(hipSetDevice(t->device));
    void **pp; size_t elem;
    PYDEM_TRY(field_ptr(t, field, &pp, &elem));
    if (!*pp || !t->have[field]) { pydem_set_error("field %d has not been computed or uploaded", field); return -3; }
    const int64_t lim = axis == 0 ? t->n : t->m;
    if (index < 0) index += lim;
    if (index < 0 || index >= lim || (axis != 0 && axis != 1)) { pydem_set_error("line index out of range"); return -2; }
    if (to_host && (field == PYDEM_UCA || field == PYDEM_EDGE_DONE || field == PYDEM_EDGE_TODO) && !tile_line_watched(t, axis, index))
        PYDEM_TRY(stage_edge_catchup(t));                       // condensed edge rounds only keep the watched lines current
    char *base = (char *)*pp;
    // the caller's array is pageable: the transfer goes through the tile's pinned staging buffer (tile_pinned)
    const size_t nbytes = (size_t)(axis == 0 ? t->m : t->n) * elem;
    void *pin = nullptr;
    PYDEM_TRY(tile_pinned(t, nbytes, &pin));
    void *user = host;
    host = pin;
    if (!to_host) memcpy(pin, user, nbytes);
    if (axis == 0) {
        char *row = base + (size_t)index * t->m * elem;
        if (to_host) HIP_TRY(hipMemcpyAsync(host, row, (size_t)t->m * elem, hipMemcpyDeviceToHost, t->stream));
        else HIP_TRY(hipMemcpyAsync(row, host, (size_t)t->m * elem, hipMemcpyHostToDevice, t->stream));
    } else {
        const int64_t cnt = t->n;
        if (!t->line_stage) PYDEM_TRY(tile_alloc(t, &t->line_stage, (size_t)(t->n > t->m ? t->n : t->m)));
        const int g = (int)(cdiv(cnt, 256) < 64 ? cdiv(cnt, 256) : 64);
        char *col = base + (size_t)index * elem;
        if (to_host) {
            if (elem == 8) hipLaunchKernelGGL(k_line_gather<double>, dim3(g), dim3(256), 0, t->stream, (const double *)col, t->m, cnt, t->line_stage);
            else hipLaunchKernelGGL(k_line_gather<uint8_t>, dim3(g), dim3(256), 0, t->stream, (const uint8_t *)col, t->m, cnt, (uint8_t *)t->line_stage);
            HIP_TRY(hipMemcpyAsync(host, t->line_stage, (size_t)cnt * elem, hipMemcpyDeviceToHost, t->stream));
        } else {
            HIP_TRY(hipMemcpyAsync(t->line_stage, host, (size_t)cnt * elem, hipMemcpyHostToDevice, t->stream));
            if (elem == 8) hipLaunchKernelGGL(k_line_scatter<double>, dim3(g), dim3(256), 0, t->stream, (const double *)t->line_stage, t->m, cnt, (double *)col);
            else hipLaunchKernelGGL(k_line_scatter<uint8_t>, dim3(g), dim3(256), 0, t->stream, (const uint8_t *)t->line_stage, t->m, cnt, (uint8_t *)col);
        }
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(t->stream));
    if (to_host) memcpy(user, pin, nbytes);
    if (!to_host && (field == PYDEM_ELEV || field == PYDEM_MAG || field == PYDEM_DIRECTION || field == PYDEM_FLATS)) t->graph_valid = false;
    return 0;
}

int pydem_tile_get_line(pydem_tile *t, int field, int axis, int64_t index, void *dst) { return line_copy(t, field, axis, index, dst, true); }

// several lines with ONE synchronisation (an edge round of the directory flow reads ~14 strips of the
// tile that just ran; one launch + copy + sync per strip is ~30 us each)
int pydem_tile_get_lines(pydem_tile *t, int count, const int *fields, const int *axes, const int64_t *indices, void *const *dsts)
{
    HIP_TRY(hipSetDevice(t->device));
    const size_t L = (size_t)(t->n > t->m ? t->n : t->m);
    if (count > 0 && t->lines_cap < count) {
        if (t->lines_stage) { HIP_TRY(hipFree(t->lines_stage)); t->device_bytes -= (int64_t)((size_t)t->lines_cap * L * 8); }
        const int cap = count < 16 ? 16 : count;
        HIP_TRY(dev_malloc(&t->lines_stage, (size_t)cap * L * 8));
        t->lines_cap = cap; t->device_bytes += (int64_t)((size_t)cap * L * 8);
    }
    void *pin_v = nullptr;                                  // (pinned staging: the callers' arrays are pageable)
    PYDEM_TRY(tile_pinned(t, (size_t)(count > 0 ? count : 1) * L * 8, &pin_v));
    char *pin = (char *)pin_v;
    std::vector<size_t> nbytes((size_t)(count > 0 ? count : 0));
    for (int k = 0; k < count; k++) {
        void **pp; size_t elem;
        PYDEM_TRY(field_ptr(t, fields[k], &pp, &elem));
        if (!*pp || !t->have[fields[k]]) { pydem_set_error("field %d has not been computed or uploaded", fields[k]); return -3; }
        const int64_t lim = axes[k] == 0 ? t->n : t->m;
        int64_t index = indices[k];
        if (index < 0) index += lim;
        if (index < 0 || index >= lim || (axes[k] != 0 && axes[k] != 1)) { pydem_set_error("line index out of range"); return -2; }
        if ((fields[k] == PYDEM_UCA || fields[k] == PYDEM_EDGE_DONE || fields[k] == PYDEM_EDGE_TODO) && !tile_line_watched(t, axes[k], index))
            PYDEM_TRY(stage_edge_catchup(t));
        char *base = (char *)*pp;
        if (axes[k] == 0) {
            nbytes[(size_t)k] = (size_t)t->m * elem;
            HIP_TRY(hipMemcpyAsync(pin + (size_t)k * L * 8, base + (size_t)index * t->m * elem, (size_t)t->m * elem, hipMemcpyDeviceToHost, t->stream));
        } else {
            const int64_t cnt = t->n;
            const int g = (int)(cdiv(cnt, 256) < 64 ? cdiv(cnt, 256) : 64);
            char *col = base + (size_t)index * elem;
            double *stage = (double *)t->lines_stage + (size_t)k * L;
            if (elem == 8) hipLaunchKernelGGL(k_line_gather<double>, dim3(g), dim3(256), 0, t->stream, (const double *)col, t->m, cnt, stage);
            else hipLaunchKernelGGL(k_line_gather<uint8_t>, dim3(g), dim3(256), 0, t->stream, (const uint8_t *)col, t->m, cnt, (uint8_t *)stage);
            nbytes[(size_t)k] = (size_t)cnt * elem;
            HIP_TRY(hipMemcpyAsync(pin + (size_t)k * L * 8, stage, (size_t)cnt * elem, hipMemcpyDeviceToHost, t->stream));
        }
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(t->stream));
    for (int k = 0; k < count; k++) memcpy(dsts[k], pin + (size_t)k * L * 8, nbytes[(size_t)k]);
    return 0;
}
int pydem_tile_set_line(pydem_tile *t, int field, int axis, int64_t index, const void *src) { return line_copy(t, field, axis, index, (void *)src, false); }

int pydem_tile_synchronize(pydem_tile *t)
{
    HIP_TRY(hipSetDevice(t->device));
    HIP_TRY(hipStreamSynchronize(t->stream));
    return 0;
}

int pydem_tile_timings(pydem_tile *t, pydem_timings *out) { *out = t->tm; return 0; }
int64_t pydem_tile_device_bytes(pydem_tile *t) { return t->device_bytes; }

int pydem_tile_synth_fractal(pydem_tile *t, uint32_t seed, int64_t row0, int64_t col0, int n_octaves,
                             int top_shift, double zmin, double zrange)
{
    HIP_TRY(hipSetDevice(t->device));
    t->graph_valid = false;
    PYDEM_TRY(ensure_field(t, PYDEM_ELEV));
    PYDEM_TRY(stage_synth(t, seed, row0, col0, n_octaves, top_shift, zmin, zrange));
    t->have[PYDEM_ELEV] = true;
    t->elev_f32 = false; t->elev_dtype = PYDEM_F64;
    for (int f = PYDEM_MAG; f < PYDEM_FIELD_COUNT; f++) t->have[f] = false;
    return 0;
}

static int need(pydem_tile *t, int field, const char *what)
{
    if (!t->have[field]) { pydem_set_error("%s: required input field %d is missing", what, field); return -3; }
    return 0;
}

int pydem_slopes_directions(pydem_tile *t)
{
    HIP_TRY(hipSetDevice(t->device));
    if (t->einc_ready) PYDEM_TRY(stage_edge_flush(t));      // (the resident UCA plane survives this stage: pending deltas first)
    t->edge_clean = false;      // these stages reuse the edge-round work lists
    t->einc_ready = false;
    PYDEM_TRY(need(t, PYDEM_ELEV, "pydem_slopes_directions"));
    if (!t->spacing_set) { pydem_set_error("pydem_slopes_directions: call pydem_tile_set_spacing first"); return -3; }
    PYDEM_TRY(ensure_fields(t, {PYDEM_MAG, PYDEM_DIRECTION, PYDEM_FLATS}));
    PYDEM_TRY(tile_alloc(t, &t->flat0, (size_t)t->NN));
    t->graph_valid = false;
    PYDEM_TRY(stage_stencil(t));
    PYDEM_TRY(stage_flats(t));
    t->have[PYDEM_MAG] = t->have[PYDEM_DIRECTION] = t->have[PYDEM_FLATS] = true;
    return 0;
}

int pydem_fill_flats(pydem_tile *t, double max_pit_area, int below_sea, double source_tol, int peaks, int pits, int artefacts_only,
                     int *needs_host)
{
    t->edge_clean = false;
    t->einc_ready = false;
    HIP_TRY(hipSetDevice(t->device));
    PYDEM_TRY(need(t, PYDEM_ELEV, "pydem_fill_flats"));
    t->graph_valid = false;
    const int r = stage_fill_flats(t, max_pit_area, below_sea, source_tol, peaks, pits, artefacts_only);
    if (r < 0) return r;
    if (needs_host) *needs_host = r;
    for (int f = PYDEM_MAG; f < PYDEM_FIELD_COUNT; f++) t->have[f] = false;     // the work planes of the conditioning
    return 0;
}

int pydem_pit_candidates(pydem_tile *t, int below_sea, int64_t *npits)
{
    HIP_TRY(hipSetDevice(t->device));
    PYDEM_TRY(need(t, PYDEM_ELEV, "pydem_pit_candidates"));
    return stage_pit_candidates(t, below_sea, npits);
}

int pydem_pit_candidates_read(pydem_tile *t, int64_t npits, int32_t *cells, double *elev)
{
    HIP_TRY(hipSetDevice(t->device));
    return stage_pit_candidates_read(t, npits, cells, elev);
}

int pydem_pit_paths(pydem_tile *t, const int32_t *order, int64_t npits, int max_iter, int max_dist, double max_dist_XY,
                    int64_t *n_failed, int64_t *iter_used, int64_t *rounds, int *needs_host)
{
    t->edge_clean = false;
    t->einc_ready = false;
    HIP_TRY(hipSetDevice(t->device));
    PYDEM_TRY(need(t, PYDEM_ELEV, "pydem_pit_paths"));
    // the path values get the dtype of the array the reference edits (:539): integer surfaces truncate them, float32 ones round
    const int dtype_mode = t->elev_dtype == PYDEM_F64 ? 0 : (t->elev_dtype == PYDEM_F32 ? 2 : 1);
    t->graph_valid = false;
    const int r = stage_pit_paths(t, order, npits, max_iter, max_dist, max_dist_XY, n_failed, iter_used, rounds, dtype_mode);
    if (r < 0) return r;
    if (needs_host) *needs_host = r;
    for (int f = PYDEM_MAG; f < PYDEM_FIELD_COUNT; f++) t->have[f] = false;
    return 0;
}

int pydem_find_flats(pydem_tile *t)
{
    t->edge_clean = false;      // these stages reuse the edge-round work lists
    HIP_TRY(hipSetDevice(t->device));
    PYDEM_TRY(need(t, PYDEM_MAG, "pydem_find_flats"));
    PYDEM_TRY(ensure_field(t, PYDEM_FLATS));
    const int grid = (int)(cdiv(t->NN, 256) < 8192 ? cdiv(t->NN, 256) : 8192);
    hipLaunchKernelGGL(k_find_flats, dim3(grid), dim3(256), 0, t->stream, t->mag, t->flats, t->NN);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(t->stream));
    t->have[PYDEM_FLATS] = true;
    return 0;
}

int pydem_uca(pydem_tile *t, pydem_options *opt)
{
    t->edge_clean = false;      // these stages reuse the edge-round work lists
    t->einc_ready = false;
    HIP_TRY(hipSetDevice(t->device));
    PYDEM_TRY(need(t, PYDEM_ELEV, "pydem_uca"));
    PYDEM_TRY(need(t, PYDEM_MAG, "pydem_uca"));
    PYDEM_TRY(need(t, PYDEM_DIRECTION, "pydem_uca"));
    PYDEM_TRY(need(t, PYDEM_FLATS, "pydem_uca"));
    if (!t->spacing_set) { pydem_set_error("pydem_uca: call pydem_tile_set_spacing first"); return -3; }
    PYDEM_TRY(ensure_fields(t, {PYDEM_SECTION, PYDEM_PROPORTION, PYDEM_UCA, PYDEM_EDGE_TODO, PYDEM_EDGE_DONE}));
    PYDEM_TRY(stage_section_graph(t, opt));
    t->graph_valid = true;
    PYDEM_TRY(stage_sweep(t, opt));
    // record minimum area (dem_processing.py:897-899)
    double mn = opt->twi_min_area;
    for (int64_t i = 0; i < t->n; i++) {
        const double a = t->h_dX2[(size_t)i] * t->h_dY2[(size_t)i];
        if (a < mn) mn = a;
    }
    opt->twi_min_area = mn;
    t->have[PYDEM_SECTION] = t->have[PYDEM_PROPORTION] = t->have[PYDEM_UCA] = true;
    t->have[PYDEM_EDGE_TODO] = t->have[PYDEM_EDGE_DONE] = true;
    return 0;
}

// a tile rebuilt from stored elev / aspect / slope (process_manager.calc_uca_ec :227-240, or a resumed directory job):
// the flow graph is built once, the first time an edge round needs it
static int ensure_graph(pydem_tile *t, pydem_options *opt, const char *who)
{
    if (t->graph_valid) return 0;
    PYDEM_TRY(need(t, PYDEM_ELEV, who));
    PYDEM_TRY(need(t, PYDEM_MAG, who));
    PYDEM_TRY(need(t, PYDEM_DIRECTION, who));
    if (!t->spacing_set) { pydem_set_error("%s: call pydem_tile_set_spacing first", who); return -3; }
    PYDEM_TRY(ensure_fields(t, {PYDEM_SECTION, PYDEM_PROPORTION}));
    PYDEM_TRY(stage_section_graph(t, opt));
    t->graph_valid = true;
    t->have[PYDEM_SECTION] = t->have[PYDEM_PROPORTION] = true;
    return 0;
}

int pydem_build_graph(pydem_tile *t, pydem_options *opt)
{
    HIP_TRY(hipSetDevice(t->device));
    PYDEM_TRY(need(t, PYDEM_FLATS, "pydem_build_graph"));
    PYDEM_TRY(ensure_fields(t, {PYDEM_EDGE_TODO, PYDEM_EDGE_DONE}));
    return ensure_graph(t, opt, "pydem_build_graph");
}

int pydem_uca_edge_update(pydem_tile *t, pydem_options *opt, const double *const data[4],
                          const uint8_t *const done[4], const uint8_t *const todo[4])
{
    HIP_TRY(hipSetDevice(t->device));
    PYDEM_TRY(need(t, PYDEM_UCA, "pydem_uca_edge_update"));
    PYDEM_TRY(need(t, PYDEM_FLATS, "pydem_uca_edge_update"));
    PYDEM_TRY(ensure_fields(t, {PYDEM_EDGE_TODO, PYDEM_EDGE_DONE}));
    PYDEM_TRY(ensure_graph(t, opt, "pydem_uca_edge_update"));
    PYDEM_TRY(stage_edge_update(t, opt, data, done, todo));
    t->have[PYDEM_EDGE_TODO] = t->have[PYDEM_EDGE_DONE] = true;
    return 0;
}

int pydem_uca_edge_round_inc(pydem_tile *t, pydem_options *opt, const double *const data[4],
                             const uint8_t *const done[4], const uint8_t *const todo[4])
{
    HIP_TRY(hipSetDevice(t->device));
    PYDEM_TRY(need(t, PYDEM_UCA, "pydem_uca_edge_round_inc"));
    PYDEM_TRY(need(t, PYDEM_FLATS, "pydem_uca_edge_round_inc"));
    PYDEM_TRY(need(t, PYDEM_EDGE_TODO, "pydem_uca_edge_round_inc"));
    PYDEM_TRY(need(t, PYDEM_EDGE_DONE, "pydem_uca_edge_round_inc"));
    PYDEM_TRY(ensure_graph(t, opt, "pydem_uca_edge_round_inc"));
    PYDEM_TRY(stage_edge_round_inc(t, opt, data, done, todo));
    return 0;
}

int pydem_uca_edge_round_inc_dev(pydem_tile *t, pydem_options *opt)
{
    HIP_TRY(hipSetDevice(t->device));
    PYDEM_TRY(need(t, PYDEM_UCA, "pydem_uca_edge_round_inc_dev"));
    PYDEM_TRY(need(t, PYDEM_FLATS, "pydem_uca_edge_round_inc_dev"));
    PYDEM_TRY(need(t, PYDEM_EDGE_TODO, "pydem_uca_edge_round_inc_dev"));
    PYDEM_TRY(need(t, PYDEM_EDGE_DONE, "pydem_uca_edge_round_inc_dev"));
    PYDEM_TRY(ensure_graph(t, opt, "pydem_uca_edge_round_inc_dev"));
    if (!t->s_data || !t->s_flags) { pydem_set_error("pydem_uca_edge_round_inc_dev: no strips (pydem_board_set_desc + pydem_board_eval first)"); return -3; }
    PYDEM_TRY(stage_edge_round_inc(t, opt, nullptr, nullptr, nullptr));
    return 0;
}

int pydem_uca_edge_flush(pydem_tile *t)
{
    HIP_TRY(hipSetDevice(t->device));
    return stage_edge_flush(t);
}

int pydem_twi(pydem_tile *t, pydem_options *opt)
{
    HIP_TRY(hipSetDevice(t->device));
    PYDEM_TRY(stage_edge_flush(t));
    PYDEM_TRY(need(t, PYDEM_UCA, "pydem_twi"));
    PYDEM_TRY(need(t, PYDEM_MAG, "pydem_twi"));
    PYDEM_TRY(ensure_field(t, PYDEM_TWI));
    PYDEM_TRY(stage_twi(t, opt));
    t->have[PYDEM_TWI] = true;
    return 0;
}

int pydem_tile_pit_edges(pydem_tile *t, int64_t *n, int32_t *src, int32_t *dst, double *w)
{
    HIP_TRY(hipSetDevice(t->device));
    const int64_t nr = t->pits.n_raw;          // slots handed out, including unused ones (src == -1)
    std::vector<int32_t> hs((size_t)nr), hd((size_t)nr);
    std::vector<double> hw((size_t)nr);
    if (nr) {
        HIP_TRY(hipMemcpyAsync(hs.data(), t->pits.raw_src, nr * 4, hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipMemcpyAsync(hd.data(), t->pits.raw_dst, nr * 4, hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipMemcpyAsync(hw.data(), t->pits.raw_w, nr * 8, hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
    }
    int64_t k = 0;
    for (int64_t e = 0; e < nr; e++) {
        if (hs[(size_t)e] < 0) continue;
        if (src) { src[k] = hs[(size_t)e]; dst[k] = hd[(size_t)e]; w[k] = hw[(size_t)e]; }
        k++;
    }
    *n = k;
    return 0;
}

// the static half of the packed graph word of every cell (uca.hip): bits 0-7 in-mask (NW N NE W E SW S SE), 8 / 9 regular
// out-edge to the facet's first / second neighbour, 10 / 11 pit out- / in-edges, 12-14 facet index
int pydem_tile_graph_words(pydem_tile *t, uint32_t *out)
{
    HIP_TRY(hipSetDevice(t->device));
    if (!t->graph_valid || !t->indeg) { pydem_set_error("pydem_tile_graph_words: no flow graph on this tile (pydem_uca / pydem_build_graph first)"); return -3; }
    HIP_TRY(hipMemcpyAsync(out, t->indeg, (size_t)t->NN * 4, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    for (int64_t c = 0; c < t->NN; c++) out[c] &= 0x7FFFu;
    return 0;
}

int pydem_tile_restore_pit_slopes(pydem_tile *t)
{
    t->edge_clean = false;      // these stages reuse the edge-round work lists
    HIP_TRY(hipSetDevice(t->device));
    if (t->pits.n_raw == 0) return 0;
    const int64_t g = cdiv(t->pits.n_raw, 256);
    hipLaunchKernelGGL(k_restore_pit_slopes, dim3((unsigned)(g < 1024 ? g : 1024)), dim3(256), 0, t->stream, t->pits.raw_src,
                       t->pits.n_raw, t->mag);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(t->stream));
    return 0;
}

int pydem_bench_stencil(pydem_tile *t, int iters, double *avg_ms)
{
    t->edge_clean = false;      // these stages reuse the edge-round work lists
    t->einc_ready = false;
    HIP_TRY(hipSetDevice(t->device));
    PYDEM_TRY(need(t, PYDEM_ELEV, "pydem_bench_stencil"));
    if (!t->spacing_set) { pydem_set_error("pydem_bench_stencil: call pydem_tile_set_spacing first"); return -3; }
    PYDEM_TRY(ensure_fields(t, {PYDEM_MAG, PYDEM_DIRECTION}));
    PYDEM_TRY(tile_alloc(t, &t->flat0, (size_t)t->NN));
    return bench_stencil(t, iters, avg_ms);
}

}  // extern "C"
