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
#include <time.h>
#include <algorithm>
#include <vector>

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

// Wait for stream s, but not for ever when a collective may be in it: after PYDEM_EDGE_TIMEOUT seconds (default 300, 0 = wait for
// ever) the call fails with a message that names what it waited for and aborts the communicator -- a rank that never arrived
// (or ranks whose call sequences differ) would otherwise leave the others in ncclAllReduce without a word.
static int collective_wait(hipStream_t s, pydem_comm *c, const char *what)
{
    static double limit_s = -1.0;
    if (limit_s < 0.0) { const char *e = getenv("PYDEM_EDGE_TIMEOUT"); limit_s = e ? atof(e) : 300.0; if (limit_s < 0.0) limit_s = 0.0; }
    if (limit_s == 0.0 || !c) { HIP_TRY(hipStreamSynchronize(s)); return 0; }
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    unsigned spins = 0;
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e == hipSuccess) return 0;
        if (e != hipErrorNotReady) { pydem_set_error("%s: %s", what, hipGetErrorString(e)); return -1; }
        if ((++spins & 1023u) == 0) {
            clock_gettime(CLOCK_MONOTONIC, &t1);
            const double dt = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
            if (dt > limit_s) {
                pydem_set_error("%s: the collective did not come back within %.0f s (PYDEM_EDGE_TIMEOUT) on rank %d of %d -- a rank left the job or the ranks "
                                "issue different collectives; the RCCL communicator is aborted", what, limit_s, c->rank, c->world);
                if (c->comm) { (void)ncclCommAbort(c->comm); c->comm = nullptr; }
                return -7;
            }
        }
    }
}

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
        HIP_TRY(dev_malloc((void **)&c->buf, (size_t)n_doubles * 8));
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
    if ((field == PYDEM_UCA || field == PYDEM_EDGE_DONE || field == PYDEM_EDGE_TODO) && !tile_line_watched(t, axis, index))
        PYDEM_TRY(stage_edge_catchup(t));                       // condensed edge rounds only keep the watched lines current
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
    if (!c->comm) { pydem_set_error("pydem_comm_allreduce: the communicator was aborted"); return -6; }
    NCCL_TRY(ncclAllReduce(c->buf, c->buf, (size_t)n_doubles, ncclDouble, op == 1 ? ncclMax : ncclSum, c->comm, c->stream));   // also for world == 1
    if (host_out) HIP_TRY(hipMemcpyAsync(host_out, c->buf, (size_t)n_doubles * 8, hipMemcpyDeviceToHost, c->stream));
    PYDEM_TRY(collective_wait(c->stream, c, "pydem_comm_allreduce"));
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

// =============================================================================================
// Edge board: the strips of the cross-tile fix-up stay on the device.
//
// The reference's edge workers read their neighbours' lines from the shared zarr store, apply the corner rules
// and hand the strips to calc_uca (pydem/process_manager.py:243-279); its manager recomputes the metrics of the
// tiles around every finished worker (calc_uca_ec_metrics :199-221, update_uca_edge_metrics :1061-1088).  With
// one tile per GPU and KiB-sized strips that bookkeeping -- not the flow routing -- sets the pace, so here it
// never leaves the device: every rank keeps a replicated BOARD of all the lines any tile reads from any other
// (or from itself), refreshed after each wave by ONE collective (pack -> ncclAllReduce -> scatter), and one
// kernel evaluates a tile from the board: it writes the strips (data, done, todo) straight into the buffers the
// tile's round reads and reduces the numbers the schedule needs (metric counts, dropped 'todo' pixels, a hash
// of the strips) into a few words per tile, the only thing the host reads per wave.
// The rules are those of pydem_amd/process_manager.py (_edge_inputs, _todo_dropped, _adopt_finished,
// _tile_metric), which stays the specification: the CPU test tier runs it with the oracle-backed processor
// and tests/test_gpu_process_manager.py checks that both produce the same waves, rounds and masks.
struct pydem_board_desc {           // one per tile; offsets in doubles into the board, -1 = no such line
    int32_t n, m;
    int64_t own_todo[4], own_done[4];             // sides: left, right, top, bottom
    int64_t nb_uca[4], nb_done[4], nb_todo[4];
    int32_t nb_self[4];                           // the edge table points the tile at its own line (mosaic border)
    int64_t cnr_done[4], cnr_uca[4];              // diagonal neighbour's pixel for the corners tl, tr, bl, br
    int32_t cnr_1ov[4];                           // check_1overlap (:286-293) of that corner
    double *s_data; uint8_t *s_flags; int32_t L;  // strips of the tile's next round (tiles resident on this device)
};

struct pydem_pack_line { const void *src; int64_t stride, count, rel; int32_t bytes; };   // bytes: 8 = double, 1 = mask

struct pydem_board_list { int n; int tile[64]; int full[64]; };
struct pydem_board_segs { int n; int64_t src[64], dst[64], cnt[64]; };

struct pydem_board {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev = nullptr;
    double *mb = nullptr; int64_t cap = 0;        // the board
    double *wb = nullptr; int64_t wcap = 0;       // wave staging (what the collective carries)
    pydem_board_desc *desc = nullptr; int n_tiles = 0;
    std::vector<pydem_board_desc> h_desc;
    unsigned long long *scal = nullptr, *h_scal = nullptr;   // [n_tiles][8]
    // per tile: where its lines live in the board, and (tiles of this rank) how to gather them from the tile
    struct TileLines { int64_t mb_start = 0, size = 0; pydem_tile *tile = nullptr; int count = 0; pydem_pack_line *lines = nullptr;
                       std::vector<pydem_pack_line> h_lines;
                       struct Lay { int64_t rel, count; int32_t bytes; };
                       std::vector<Lay> layout;          // every line of the tile (also of another rank's tile): where on the board, how long, 8 = area / 1 = mask
                       std::vector<std::pair<int, int64_t>> where; };      // (axis, index) of the lines: the tile's condensed edge rounds watch them
    std::vector<TileLines> tl;
    // queued waves (pydem_board_run_waves): the schedule's state on the device + a pinned copy
    unsigned long long *sched = nullptr, *h_sched = nullptr, *scal_tb = nullptr;
    void *q_tiles = nullptr; void *q_lines = nullptr; void *q_unpack = nullptr; int q_count = 0, q_nlines = 0, q_nunpack = 0; int64_t q_nper = 1;   // tables of the queued rounds / gathers / the unpacking after the collective
    int64_t q_stage_bytes = 0; bool q_staged_ok = false;   // staging layout of the queued collective (areas as doubles, masks as bytes), known for every tile
    hipGraphExec_t wave_exec[2] = {nullptr, nullptr};   // one wave, captured: before / after the collective (one graph without one)
    unsigned long long wave_ok = 0; bool wave_comm = false; hipStream_t wave_stream = nullptr;   // what the graphs were captured for
    bool graph_failed = false, tables_valid = false;
    pydem_comm *last_comm = nullptr;                     // communicator of the last refresh (the evaluation that follows waits for its collective: with the watchdog)
    bool prepared = false; unsigned long long prepared_ok = 0; bool prepared_staged = false;   // pydem_board_prepare_waves passed for this set of tiles
    std::vector<int> q_mine;                             // this rank's tiles of the prepared set
    int g_eval = 1;
    unsigned long long *h_progress = nullptr;            // pinned, written by the selection kernel: [0] waves selected so far in the batch, [1] members of the last one
    unsigned char *h_stage = nullptr; size_t h_stage_bytes = 0;   // pinned copy of the staging buffer (batches whose sum over the ranks is the caller's: run_waves_ex)
};

// ---- queued waves: layout of the schedule's state (64-bit words; the same table on the host, pydem_amd/_ffi.py) ----
// [0] tiles whose rounds may be queued (bit per tile)   [1] out: why the batch stopped (0 = it did not)
// [2] out: waves run                                    [3] waves allowed
// [4] tiles whose metric is read again before the next selection (the last wave + its side neighbours)
// [5] members of the current wave                       [6] tiles the current wave's lines are read by
// [8 + a] metric numerator / [72 + a] denominator of tile a as the host schedule holds them (a diagonal neighbour keeps its
// old numbers, process_manager.py:1116-1136)            [136 + a] strip hash of a's last round, [200 + a] whether it has one
// [264 + a] tiles that read a line of a                 [328 + a] a and its four side neighbours
// [392 + w] members of wave w of the batch
// [456 + a] the tile's round counter when the batch began (seed stamps)   [7] out: the waves ran as captured graphs
// [520] the tile whose strips are evaluated again with rule :274 everywhere before this wave (tie-break), as a bit
// [521] out: tie-breaks of the batch   [522] out: bit w = wave w of the batch was a tie-break
enum { SCH_OK = 0, SCH_STOP = 1, SCH_NWAVES = 2, SCH_LIMIT = 3, SCH_CHECK = 4, SCH_WAVE = 5, SCH_AFFECTED = 6, SCH_GRAPH = 7, SCH_ND = 8,
       SCH_PD = 72, SCH_HASH = 136, SCH_HAS = 200, SCH_READERS = 264, SCH_NBRS = 328, SCH_LOG = 392, SCH_ROUND = 456, SCH_TB = 520,
       SCH_NTB = 521, SCH_TBLOG = 522, SCH_WORDS = 528 };

static void board_drop_graphs(pydem_board *b)
{
    for (auto &e : b->wave_exec) if (e) { (void)hipGraphExecDestroy(e); e = nullptr; }
}

namespace {

__device__ __forceinline__ unsigned long long mix64(unsigned long long x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

// one thread per (side, position) of one tile
__global__ __launch_bounds__(1024) void k_board_eval(const double *__restrict__ mb, const pydem_board_desc *__restrict__ descs,
                                                    pydem_board_list Lst, unsigned long long *__restrict__ scal,
                                                    const unsigned long long *__restrict__ gate)
{
    const int tile = Lst.tile[blockIdx.y], full = Lst.full[blockIdx.y];
    if (gate && !((*gate >> tile) & 1ull)) return;      // queued waves: only the tiles that read a line of the wave
    const pydem_board_desc D = descs[tile];
    const int n = D.n, m = D.m;
    const int64_t total = 2 * (int64_t)n + 2 * (int64_t)m;
    unsigned long long a_ndone = 0, a_pdone = 0, a_dself = 0, a_dfull = 0, a_seeds = 0, a_hash = 0;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        int k; int64_t p;
        if (q < n) { k = 0; p = q; } else if (q < 2 * (int64_t)n) { k = 1; p = q - n; }
        else if (q < 2 * (int64_t)n + m) { k = 2; p = q - 2 * (int64_t)n; } else { k = 3; p = q - 2 * (int64_t)n - m; }
        const int64_t len = k < 2 ? n : m;
        auto rd = [&](int64_t off, int64_t pos) -> double { return off < 0 ? 0.0 : mb[off + pos]; };
        const bool own_td = rd(D.own_todo[k], p) != 0.0;
        const bool own_dn = rd(D.own_done[k], p) != 0.0;
        double data = rd(D.nb_uca[k], p);
        const bool nb_dn = rd(D.nb_done[k], p) != 0.0;
        const bool nb_td = rd(D.nb_todo[k], p) != 0.0;
        bool done = nb_dn;
        // ---- corner rules (:258-270): the top / bottom strip gives its corner up when both strips are finished
        // there; with a one-pixel overlap both take the diagonal neighbour's value when that one is finished
        const bool at_lo = p == 0, at_hi = p == len - 1;
        int ctb = -1, clr = -1;
        if (at_lo || at_hi) {
            if (k >= 2) { ctb = k; clr = at_lo ? 0 : 1; }
            else { clr = k; ctb = at_lo ? 2 : 3; }
        }
        // (a 1-wide or 1-high tile has both ends in one position: the lower end's rule wins like the dict order)
        if (ctb >= 0) {
            const int64_t pos_tb = clr == 0 ? 0 : m - 1;      // corner on the top / bottom strip
            const int64_t pos_lr = ctb == 2 ? 0 : n - 1;      // corner on the left / right strip
            const bool d_tb = rd(D.nb_done[ctb], pos_tb) != 0.0;
            const bool d_lr = rd(D.nb_done[clr], pos_lr) != 0.0;
            if (d_tb && d_lr) {
                if (k >= 2) done = false;                                                           // :266
                const int key = (ctb == 2 ? 0 : 2) + (clr == 0 ? 0 : 1);                            // tl, tr, bl, br
                if (D.cnr_1ov[key] && D.cnr_done[key] >= 0 && mb[D.cnr_done[key]] != 0.0) data = mb[D.cnr_uca[key]];   // :268-270
            }
        }
        // ---- rule :274 (full), or only where the neighbour line is the tile's own (mosaic border)
        const bool drop_self = own_td && nb_td && D.nb_self[k];
        const bool drop_full = own_td && nb_td;
        const bool td = own_td && !(full ? drop_full : drop_self);
        const bool td_adopt = td || (done && !own_dn);                                             // _adopt_finished
        if (D.s_data) {
            D.s_data[(size_t)k * D.L + p] = data;
            D.s_flags[(size_t)k * D.L + p] = done;
            D.s_flags[(size_t)(4 + k) * D.L + p] = td_adopt;
        }
        a_seeds += done && td_adopt;
        // ---- metric (:199-221): 'todo' cells facing a finished neighbour cell, all eight keys
        a_ndone += own_td && nb_dn;
        a_pdone += own_td;
        if (k >= 2 && (at_lo || at_hi)) {
            const int key = (k == 2 ? 0 : 2) + (at_lo ? 0 : 1);
            const bool edn = D.cnr_done[key] >= 0 && mb[D.cnr_done[key]] != 0.0;
            a_ndone += own_td && edn;
            a_pdone += own_td;
            if (at_lo && at_hi) {   // m == 1: the same pixel is the other corner too
                const int key2 = (k == 2 ? 0 : 2) + 1;
                const bool edn2 = D.cnr_done[key2] >= 0 && mb[D.cnr_done[key2]] != 0.0;
                a_ndone += own_td && edn2; a_pdone += own_td;
            }
        }
        // ---- dropped 'todo' pixels (_todo_dropped): a corner pixel survives if either of its strips keeps it
        if (k < 2 || !(at_lo || at_hi)) {
            bool keep_s = own_td && !drop_self, keep_f = own_td && !drop_full;
            if (k < 2 && (at_lo || at_hi)) {
                const int64_t pos_tb = k == 0 ? 0 : m - 1;
                const bool tb_own = rd(D.own_todo[ctb], pos_tb) != 0.0, tb_nb = rd(D.nb_todo[ctb], pos_tb) != 0.0;
                keep_s = keep_s || (tb_own && !(tb_nb && D.nb_self[ctb]));
                keep_f = keep_f || (tb_own && !tb_nb);
            }
            a_dself += own_td && !keep_s;
            a_dfull += own_td && !keep_f;
        }
        unsigned long long h = (unsigned long long)__double_as_longlong(done ? data : 0.0);
        h = mix64(h ^ ((unsigned long long)q * 0x9E3779B97F4A7C15ull) ^ ((unsigned long long)done << 1) ^ (unsigned long long)td_adopt);
        a_hash += h;
    }
    // reduction: lanes of a wavefront by shuffles, the wavefronts of the block through LDS, one atomic per block and value
    // (blocks of 1024 threads, at most 16 per tile: with 64-256 small blocks the same-address atomics of a tile were half of the
    // kernel's time)
    __shared__ unsigned long long red[6][16];
    unsigned long long v[6] = {a_ndone, a_pdone, a_dself, a_dfull, a_seeds, a_hash};
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        unsigned long long x = v[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
        if (lane == 0) red[j][wv] = x;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        unsigned long long x = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); w++) x += red[threadIdx.x][w];
        atomicAdd(&scal[(size_t)tile * 8 + threadIdx.x], x);
    }
}

// all lines of one tile -> wave staging (blockIdx.y = line)
__global__ void k_board_pack(const pydem_pack_line *__restrict__ lines, int nlines, double *__restrict__ wb)
{
    for (int l = blockIdx.y; l < nlines; l += gridDim.y) {
        const pydem_pack_line P = lines[l];
        double *dst = wb + P.rel;
        if (P.bytes == 8) {
            const double *src = (const double *)P.src;
            for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < P.count; k += (int64_t)gridDim.x * blockDim.x) dst[k] = src[k * P.stride];
        } else {
            const uint8_t *src = (const uint8_t *)P.src;
            for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < P.count; k += (int64_t)gridDim.x * blockDim.x) dst[k] = (double)src[k * P.stride];
        }
    }
}

// wave staging -> board (blockIdx.y = tile of the wave)
__global__ void k_board_scatter(const double *__restrict__ wb, double *__restrict__ mb, pydem_board_segs S)
{
    for (int s = blockIdx.y; s < S.n; s += gridDim.y) {
        const int64_t src = S.src[s], dst = S.dst[s], cnt = S.cnt[s];
        for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < cnt; k += (int64_t)gridDim.x * blockDim.x) mb[dst + k] = wb[src + k];
    }
}

// ---- queued waves -------------------------------------------------------------------------------------------------
// One wave of ProcessManager._process_uca_edges_pool_device chosen on the device (one thread per tile, one workgroup): the
// candidates are the tiles whose metric is positive or that still drop 'todo' pixels on the mosaic border, and whose
// strips changed since their last round; with at most 2 * n_workers tiles the ranking selects every candidate, so the
// wave IS the candidate set.  The batch stops when there is no candidate (the host decides about the tie-break rule), when a
// candidate's round cannot be queued (its first round builds the fix-up state on the host) or at the wave limit.
__global__ void k_sched_select(unsigned long long *__restrict__ S, unsigned long long *__restrict__ scal, int n_tiles, unsigned long long *progress)
{
    const int a = threadIdx.x;
    __shared__ unsigned long long s_wave, s_aff, s_chk, s_drop[64];
    if (a == 0) { s_wave = 0; s_aff = 0; s_chk = 0; }
    s_drop[a] = 0;
    __syncthreads();
    const bool live = S[SCH_STOP] == 0;
    bool cand = false;
    if (live && a < n_tiles) {
        if ((S[SCH_CHECK] >> a) & 1ull) { S[SCH_ND + a] = scal[(size_t)a * 8 + 0]; S[SCH_PD + a] = scal[(size_t)a * 8 + 1]; }
        cand = (S[SCH_ND + a] > 0 || scal[(size_t)a * 8 + 2] != 0) && !(S[SCH_HAS + a] && S[SCH_HASH + a] == scal[(size_t)a * 8 + 5]);
        if (cand) atomicOr(&s_wave, 1ull << a);
        s_drop[a] = scal[(size_t)a * 8 + 3];
    }
    __syncthreads();
    unsigned long long wave = s_wave;
    int stop = 0, tb = -1;
    if (live) {
        if (S[SCH_NWAVES] >= S[SCH_LIMIT]) stop = 3;
        else if (wave == 0) {
            // no candidate: the tile that would drop the most 'todo' pixels under rule :274 everywhere runs alone, its strips
            // evaluated with that rule (lowest index first); none: the fix-up is over
            unsigned long long best = 0;
            for (int q = 0; q < n_tiles; q++) if (s_drop[q] > best) { best = s_drop[q]; tb = q; }
            if (tb < 0) stop = 1;
            else if (!((S[SCH_OK] >> tb) & 1ull)) stop = 2;
            else wave = 1ull << tb;
        }
        else if (wave & ~S[SCH_OK]) stop = 2;
    }
    const bool run = live && stop == 0;
    const bool member = run && a < n_tiles && ((wave >> a) & 1ull);
    if (member) {
        if (tb < 0) { S[SCH_HASH + a] = scal[(size_t)a * 8 + 5]; S[SCH_HAS + a] = 1; }
        else S[SCH_HAS + a] = 0;                               // (a tie-break forgets the tile's last strips)
        atomicOr(&s_aff, S[SCH_READERS + a]);
        atomicOr(&s_chk, S[SCH_NBRS + a]);
    }
    __syncthreads();
    // the evaluation kernel accumulates into the scalars of the tiles it visits
    if (run && a < n_tiles && ((s_aff >> a) & 1ull))
        for (int j = 0; j < 8; j++) scal[(size_t)a * 8 + j] = 0ull;
    if (a == 0) {
        if (run) {
            if (tb >= 0) { S[SCH_TBLOG] |= 1ull << S[SCH_NWAVES]; S[SCH_NTB] += 1; }
            S[SCH_LOG + S[SCH_NWAVES]] = wave; S[SCH_NWAVES] += 1; S[SCH_WAVE] = wave; S[SCH_AFFECTED] = s_aff; S[SCH_CHECK] = s_chk;
            S[SCH_TB] = tb >= 0 ? wave : 0ull;
            // (host memory: what the watchdog of pydem_board_run_waves names when a batch does not come back)
            if (progress) { __hip_atomic_store(&progress[1], wave, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __hip_atomic_store(&progress[0], S[SCH_NWAVES], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
        } else { S[SCH_WAVE] = 0; S[SCH_AFFECTED] = 0; S[SCH_CHECK] = 0; S[SCH_TB] = 0; if (live) S[SCH_STOP] = (unsigned long long)stop; }
    }
}

// end of a batch: the metrics of the last wave's neighbourhood, as the next selection would read them
__global__ void k_sched_settle(unsigned long long *__restrict__ S, const unsigned long long *__restrict__ scal, int n_tiles)
{
    const int a = threadIdx.x;
    if (a < n_tiles && ((S[SCH_CHECK] >> a) & 1ull)) { S[SCH_ND + a] = scal[(size_t)a * 8 + 0]; S[SCH_PD + a] = scal[(size_t)a * 8 + 1]; }
    __syncthreads();
    if (a == 0) S[SCH_CHECK] = 0;
}

// the lines of all tiles of this rank -> staging at the board's own offsets, for the members of the wave (blockIdx.y = line)
struct pydem_pack_line_q { pydem_pack_line P; int64_t abs, st; int32_t tile, pad; };      // abs: board offset (doubles); st: staging offset (bytes)
// staged == 0: straight to the board (no collective); staged == 1: into the staging buffer the collective carries -- areas as
// doubles, masks as BYTES (a sum over ranks of disjoint fills is exact byte by byte as well: x + 0)
__global__ void k_board_pack_gated(const pydem_pack_line_q *__restrict__ lines, int nlines, double *__restrict__ mb, unsigned char *__restrict__ wb,
                                   int staged, const unsigned long long *__restrict__ gate)
{
    const unsigned long long wave = *gate;
    for (int l = blockIdx.y; l < nlines; l += gridDim.y) {
        const pydem_pack_line_q Q = lines[l];
        if (!((wave >> Q.tile) & 1ull)) continue;
        const pydem_pack_line &P = Q.P;
        const int64_t k0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, dk = (int64_t)gridDim.x * blockDim.x;
        if (P.bytes == 8) {
            const double *src = (const double *)P.src;
            double *dst = staged ? reinterpret_cast<double *>(wb + Q.st) : mb + Q.abs;
            for (int64_t k = k0; k < P.count; k += dk) dst[k] = src[k * P.stride];
        } else {
            const uint8_t *src = (const uint8_t *)P.src;
            if (staged) { unsigned char *dst = wb + Q.st; for (int64_t k = k0; k < P.count; k += dk) dst[k] = src[k * P.stride] != 0; }
            else { double *dst = mb + Q.abs; for (int64_t k = k0; k < P.count; k += dk) dst[k] = (double)src[k * P.stride]; }
        }
    }
}

// staging -> board after the collective: the lines of ALL tiles that are members of the wave (blockIdx.y = line)
struct pydem_unpack_line { int64_t st, abs, count; int32_t bytes, tile; };
__global__ void k_board_unpack_gated(const pydem_unpack_line *__restrict__ lines, int nlines, const unsigned char *__restrict__ wb,
                                     double *__restrict__ mb, const unsigned long long *__restrict__ gate)
{
    const unsigned long long wave = *gate;
    for (int l = blockIdx.y; l < nlines; l += gridDim.y) {
        const pydem_unpack_line U = lines[l];
        if (!((wave >> U.tile) & 1ull)) continue;
        const int64_t k0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, dk = (int64_t)gridDim.x * blockDim.x;
        double *dst = mb + U.abs;
        if (U.bytes == 8) { const double *src = reinterpret_cast<const double *>(wb + U.st); for (int64_t k = k0; k < U.count; k += dk) dst[k] = src[k]; }
        else { const unsigned char *src = wb + U.st; for (int64_t k = k0; k < U.count; k += dk) dst[k] = (double)src[k]; }
    }
}

__global__ void k_board_zero(unsigned long long *scal, pydem_board_list Lst)
{
    const int k = threadIdx.x >> 3, j = threadIdx.x & 7;
    if (k < Lst.n) scal[(size_t)Lst.tile[k] * 8 + j] = 0ull;
}

}  // namespace

extern "C" {

int pydem_board_create(int device, int n_tiles, int64_t n_doubles, pydem_board **out)
{
    HIP_TRY(hipSetDevice(device));
    pydem_board *b = new pydem_board();
    b->device = device; b->n_tiles = n_tiles; b->cap = n_doubles;
    HIP_TRY(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&b->ev, hipEventDisableTiming));
    HIP_TRY(dev_malloc((void **)&b->mb, (size_t)(n_doubles > 0 ? n_doubles : 1) * 8));
    HIP_TRY(hipMemsetAsync(b->mb, 0, (size_t)(n_doubles > 0 ? n_doubles : 1) * 8, b->stream));
    HIP_TRY(hipMalloc((void **)&b->desc, (size_t)n_tiles * sizeof(pydem_board_desc)));
    HIP_TRY(hipMalloc((void **)&b->scal, (size_t)n_tiles * 8 * sizeof(unsigned long long)));
    HIP_TRY(hipHostMalloc((void **)&b->h_scal, (size_t)n_tiles * 8 * sizeof(unsigned long long)));
    b->h_desc.resize((size_t)n_tiles);
    *out = b;
    return 0;
}

int pydem_board_destroy(pydem_board *b)
{
    if (!b) return 0;
    (void)hipSetDevice(b->device);
    (void)hipStreamSynchronize(b->stream);
    for (void *p : {(void *)b->mb, (void *)b->wb, (void *)b->desc, (void *)b->scal}) if (p) (void)hipFree(p);
    for (auto &T : b->tl) if (T.lines) (void)hipFree(T.lines);
    if (b->h_scal) (void)hipHostFree(b->h_scal);
    if (b->sched) (void)hipFree(b->sched);
    if (b->scal_tb) (void)hipFree(b->scal_tb);
    if (b->h_sched) (void)hipHostFree(b->h_sched);
    if (b->h_progress) (void)hipHostFree(b->h_progress);
    if (b->h_stage) (void)hipHostFree(b->h_stage);
    board_drop_graphs(b);
    if (b->q_tiles) (void)hipFree(b->q_tiles);
    if (b->q_lines) (void)hipFree(b->q_lines);
    if (b->q_unpack) (void)hipFree(b->q_unpack);
    if (b->ev) (void)hipEventDestroy(b->ev);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
    return 0;
}

// descriptor of one tile: 4+4+12+8 offsets and the flags, in the order of pydem_board_desc (see _ffi.Board.set_desc);
// `tile` = the resident tile whose strip buffers the evaluation fills, or NULL for a tile of another rank / device
int pydem_board_set_desc(pydem_board *b, int index, int32_t n, int32_t m, const int64_t *offsets28, const int32_t *flags8, pydem_tile *tile)
{
    HIP_TRY(hipSetDevice(b->device));
    if (index < 0 || index >= b->n_tiles) { pydem_set_error("pydem_board_set_desc: tile index out of range"); return -2; }
    pydem_board_desc &D = b->h_desc[(size_t)index];
    D.n = n; D.m = m;
    const int64_t *o = offsets28;
    for (int k = 0; k < 4; k++) { D.own_todo[k] = o[k]; D.own_done[k] = o[4 + k]; D.nb_uca[k] = o[8 + k]; D.nb_done[k] = o[12 + k]; D.nb_todo[k] = o[16 + k];
                                  D.cnr_done[k] = o[20 + k]; D.cnr_uca[k] = o[24 + k]; D.nb_self[k] = flags8[k]; D.cnr_1ov[k] = flags8[4 + k]; }
    for (int k = 0; k < 28; k++) if (o[k] >= b->cap) { pydem_set_error("pydem_board_set_desc: offset beyond the board"); return -2; }
    D.s_data = nullptr; D.s_flags = nullptr; D.L = n > m ? n : m;
    if (tile) {
        if (tile->device != b->device) { pydem_set_error("pydem_board_set_desc: tile lives on another device"); return -2; }
        if (tile->n != n || tile->m != m) { pydem_set_error("pydem_board_set_desc: tile shape mismatch"); return -2; }
        PYDEM_TRY(tile_alloc(tile, &tile->s_data, (size_t)D.L * 4));
        PYDEM_TRY(tile_alloc(tile, &tile->s_flags, (size_t)D.L * 8));
        D.s_data = tile->s_data; D.s_flags = tile->s_flags;
    }
    HIP_TRY(hipMemcpyAsync(b->desc + index, &D, sizeof(D), hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return 0;
}

// which lines of the board belong to tile `index` (contiguous: mb_start .. mb_start + size) and, for a tile of this rank,
// how to gather them: line k = fields[k] / axes[k] / indices[k] of `tile`, rel_offsets[k] doubles after mb_start
int pydem_board_set_lines(pydem_board *b, int index, int64_t mb_start, int64_t size, pydem_tile *tile, int count,
                          const int *fields, const int *axes, const int64_t *indices, const int64_t *rel_offsets)
{
    HIP_TRY(hipSetDevice(b->device));
    if (index < 0 || index >= b->n_tiles) { pydem_set_error("pydem_board_set_lines: tile index out of range"); return -2; }
    if (mb_start < 0 || mb_start + size > b->cap) { pydem_set_error("pydem_board_set_lines: lines beyond the board"); return -2; }
    if ((int)b->tl.size() != b->n_tiles) b->tl.resize((size_t)b->n_tiles);
    pydem_board::TileLines &T = b->tl[(size_t)index];
    T.mb_start = mb_start; T.size = size; T.tile = tile; T.count = 0; T.where.clear(); T.layout.clear();
    // the layout of the tile's lines, for tiles of other ranks too when the caller lists them (the queued waves carry masks as
    // bytes through the collective and need to know which lines are masks)
    for (int k = 0; k < count; k++) {
        const pydem_board_desc &D = b->h_desc[(size_t)index];
        if (axes[k] != 0 && axes[k] != 1) { pydem_set_error("pydem_board_set_lines: axis"); return -2; }
        pydem_board::TileLines::Lay L;
        L.rel = rel_offsets[k]; L.count = axes[k] == 0 ? D.m : D.n; L.bytes = fields[k] == PYDEM_UCA ? 8 : 1;
        if (L.rel < 0 || L.rel + L.count > size) { pydem_set_error("pydem_board_set_lines: line beyond the tile's board segment"); return -2; }
        T.layout.push_back(L);
    }
    if (!tile) return 0;
    if (tile->device != b->device) { pydem_set_error("pydem_board_set_lines: tile lives on another device"); return -2; }
    std::vector<pydem_pack_line> h((size_t)count);
    for (int k = 0; k < count; k++) {
        void **pp; size_t elem;
        const int axis = axes[k];
        int64_t idx = indices[k];
        const int64_t lim = axis == 0 ? tile->n : tile->m;
        if (idx < 0) idx += lim;
        if (idx < 0 || idx >= lim || (axis != 0 && axis != 1)) { pydem_set_error("pydem_board_set_lines: line index out of range"); return -2; }
        const int f = fields[k];
        const void *base = nullptr;
        if (f == PYDEM_UCA) { base = tile->uca; elem = 8; }
        else if (f == PYDEM_EDGE_TODO) { base = tile->edge_todo; elem = 1; }
        else if (f == PYDEM_EDGE_DONE) { base = tile->edge_done; elem = 1; }
        else { pydem_set_error("pydem_board_set_lines: field %d is not an edge field", f); return -2; }
        (void)pp;
        if (!base || !tile->have[f]) { pydem_set_error("pydem_board_set_lines: field %d is not resident", f); return -3; }
        tile_watch_line(tile, axis, idx);          // the condensed edge rounds keep this line current between two rounds
        T.where.emplace_back(axis, idx);
        pydem_pack_line &P = h[(size_t)k];
        P.count = axis == 0 ? tile->m : tile->n;
        P.stride = axis == 0 ? 1 : tile->m;
        P.src = (const char *)base + (size_t)(axis == 0 ? idx * tile->m : idx) * elem;
        P.rel = rel_offsets[k]; P.bytes = (int32_t)elem;
        if (P.rel < 0 || P.rel + P.count > size) { pydem_set_error("pydem_board_set_lines: line beyond the tile's board segment"); return -2; }
    }
    if (T.lines) HIP_TRY(hipFree(T.lines));
    HIP_TRY(hipMalloc((void **)&T.lines, (size_t)(count > 0 ? count : 1) * sizeof(pydem_pack_line)));
    HIP_TRY(hipMemcpyAsync(T.lines, h.data(), (size_t)count * sizeof(pydem_pack_line), hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    T.count = count;
    T.h_lines = h;
    return 0;
}

// Refresh the board with the lines of the tiles that ran in a wave (the same list on every rank, at most 64 tiles per
// call): the tiles of this rank gather their lines into the wave staging buffer (one kernel each, on the tile's own
// stream behind the round it just ran), with a communicator the staging buffer is summed over the ranks (disjoint
// fills), then one kernel copies it into the board.
// the staging buffer of a wave: segment table, capacity, zero fill (when the buffer is summed over ranks) and the pack kernels
// of this rank's tiles, ordered against the board's stream by events
static int board_stage(pydem_board *b, int n_wave, const int *wave_tiles, bool summed, pydem_board_segs &S, int64_t &total)
{
    if (n_wave > 64) { pydem_set_error("pydem_board_refresh: at most 64 tiles per call"); return -2; }
    S.n = n_wave;
    total = 0;
    for (int k = 0; k < n_wave; k++) {
        const int i = wave_tiles[k];
        if (i < 0 || i >= (int)b->tl.size()) { pydem_set_error("pydem_board_refresh: tile without lines (pydem_board_set_lines)"); return -2; }
        S.src[k] = total; S.dst[k] = b->tl[(size_t)i].mb_start; S.cnt[k] = b->tl[(size_t)i].size;
        total += b->tl[(size_t)i].size;
    }
    if (total > b->wcap) {
        if (b->wb) HIP_TRY(hipFree(b->wb));
        HIP_TRY(dev_malloc((void **)&b->wb, (size_t)total * 8));
        b->wcap = total;
    }
    if (summed) HIP_TRY(hipMemsetAsync(b->wb, 0, (size_t)total * 8, b->stream));
    HIP_TRY(hipEventRecord(b->ev, b->stream));
    for (int k = 0; k < n_wave; k++) {
        pydem_board::TileLines &T = b->tl[(size_t)wave_tiles[k]];
        if (!T.tile || T.count == 0) continue;
        pydem_tile *t = T.tile;
        for (const auto &ln : T.where)             // (a line registered after the tile's condensed graph was built: the interior first)
            if (!tile_line_watched(t, ln.first, ln.second)) { PYDEM_TRY(stage_edge_catchup(t)); break; }
        HIP_TRY(hipStreamWaitEvent(t->stream, b->ev, 0));
        hipLaunchKernelGGL(k_board_pack, dim3(8, T.count), dim3(256), 0, t->stream, T.lines, T.count, b->wb + S.src[k]);
        HIP_TRY(hipEventRecord(t->ev_snap, t->stream));
        HIP_TRY(hipStreamWaitEvent(b->stream, t->ev_snap, 0));
    }
    return 0;
}

int pydem_board_refresh(pydem_board *b, pydem_comm *c, int n_wave, const int *wave_tiles)
{
    HIP_TRY(hipSetDevice(b->device));
    if (n_wave <= 0) return 0;
    pydem_board_segs S;
    int64_t total = 0;
    PYDEM_TRY(board_stage(b, n_wave, wave_tiles, c != nullptr, S, total));
    b->last_comm = c;
    if (c && !c->comm) { pydem_set_error("pydem_board_refresh: the communicator was aborted"); return -6; }
    if (c) NCCL_TRY(ncclAllReduce(b->wb, b->wb, (size_t)total, ncclDouble, ncclSum, c->comm, b->stream));   // also for world == 1
    hipLaunchKernelGGL(k_board_scatter, dim3(16, n_wave), dim3(256), 0, b->stream, b->wb, b->mb, S);
    HIP_TRY(hipGetLastError());
    return 0;
}

// The same refresh with the sum over ranks done by the caller on the host (transports without RCCL: the torch.distributed
// fallback of pydem_amd/parallel.py): `stage` packs this rank's lines of the wave into the zeroed staging buffer and
// returns it, the caller sums the buffers of all ranks, `unstage` files the result on the board.
int pydem_board_refresh_stage(pydem_board *b, int n_wave, const int *wave_tiles, double *host_out, int64_t cap, int64_t *n_doubles)
{
    HIP_TRY(hipSetDevice(b->device));
    *n_doubles = 0;
    if (n_wave <= 0) return 0;
    pydem_board_segs S;
    int64_t total = 0;
    PYDEM_TRY(board_stage(b, n_wave, wave_tiles, true, S, total));
    if (total > cap) { pydem_set_error("pydem_board_refresh_stage: the wave needs %lld doubles, the buffer holds %lld", (long long)total, (long long)cap); return -2; }
    HIP_TRY(hipMemcpyAsync(host_out, b->wb, (size_t)total * 8, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    *n_doubles = total;
    return 0;
}

int pydem_board_refresh_unstage(pydem_board *b, int n_wave, const int *wave_tiles, const double *host_in, int64_t n_doubles)
{
    HIP_TRY(hipSetDevice(b->device));
    if (n_wave <= 0) return 0;
    if (n_wave > 64) { pydem_set_error("pydem_board_refresh: at most 64 tiles per call"); return -2; }
    pydem_board_segs S;
    S.n = n_wave;
    int64_t total = 0;
    for (int k = 0; k < n_wave; k++) {
        const int i = wave_tiles[k];
        if (i < 0 || i >= (int)b->tl.size()) { pydem_set_error("pydem_board_refresh: tile without lines (pydem_board_set_lines)"); return -2; }
        S.src[k] = total; S.dst[k] = b->tl[(size_t)i].mb_start; S.cnt[k] = b->tl[(size_t)i].size;
        total += b->tl[(size_t)i].size;
    }
    if (total != n_doubles || total > b->wcap) { pydem_set_error("pydem_board_refresh_unstage: buffer does not belong to this wave"); return -2; }
    HIP_TRY(hipMemcpyAsync(b->wb, host_in, (size_t)total * 8, hipMemcpyHostToDevice, b->stream));
    hipLaunchKernelGGL(k_board_scatter, dim3(16, n_wave), dim3(256), 0, b->stream, b->wb, b->mb, S);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(b->stream));      // (host_in is the caller's pageable memory)
    return 0;
}

// evaluate `count` tiles (tiles[k], full[k]: rule :274 everywhere instead of on the mosaic border only); the scalars
// of ALL tiles of the board are returned (8 words per tile: n_done, p_done, dropped_self, dropped_full, seeds, hash, -, -;
// tiles that were not evaluated keep their previous values)
int pydem_board_eval(pydem_board *b, int count, const int *tiles, const int *full, unsigned long long *out)
{
    HIP_TRY(hipSetDevice(b->device));
    for (int k0 = 0; k0 < count; k0 += 64) {
        pydem_board_list Lst;
        Lst.n = count - k0 < 64 ? count - k0 : 64;
        int64_t most = 1;
        for (int k = 0; k < Lst.n; k++) {
            const int i = tiles[k0 + k];
            if (i < 0 || i >= b->n_tiles) { pydem_set_error("pydem_board_eval: tile index out of range"); return -2; }
            Lst.tile[k] = i; Lst.full[k] = full[k0 + k];
            const pydem_board_desc &D = b->h_desc[(size_t)i];
            const int64_t total = 2 * (int64_t)D.n + 2 * (int64_t)D.m;
            if (total > most) most = total;
        }
        const int g = (int)(cdiv(most, 4096) < 16 ? cdiv(most, 4096) : 16);
        hipLaunchKernelGGL(k_board_zero, dim3(1), dim3(512), 0, b->stream, b->scal, Lst);
        hipLaunchKernelGGL(k_board_eval, dim3(g, Lst.n), dim3(1024), 0, b->stream, b->mb, b->desc, Lst, b->scal, (const unsigned long long *)nullptr);
    }
    HIP_TRY(hipMemcpyAsync(b->h_scal, b->scal, (size_t)b->n_tiles * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipGetLastError());
    PYDEM_TRY(collective_wait(b->stream, b->last_comm, "pydem_board_eval (behind pydem_board_refresh)"));
    memcpy(out, b->h_scal, (size_t)b->n_tiles * 8 * sizeof(unsigned long long));
    return 0;
}

// Up to `k_waves` waves of the fix-up queued back to back without a look from the host (process_manager.py:1214-1246 as
// deterministic waves; for a mosaic of at most 2 * n_workers tiles, where the ranking selects every candidate): per wave
// the selection kernel, for every tile of this rank a condensed round and the gather of its lines (both gated by the
// wave's member word, on the tile's stream), the sum over the ranks when `c` is given (the WHOLE staging buffer: its size
// must not depend on the wave), the copy to the board and the evaluation of the tiles that read the wave's lines.
// `state` (SCH_WORDS 64-bit words, layout above) carries the schedule in and out; `scal_out` as in pydem_board_eval.
//
// Two calls, because a batch contains collectives: pydem_board_prepare_waves does everything that can FAIL on one rank alone
// (the tiles' fix-up state, watched lines, staging layout, the device tables and their allocations) and enqueues nothing;
// the caller lets the ranks agree on its verdict (one allreduce) and only then calls pydem_board_run_waves, which -- for
// the prepared set of tiles -- only enqueues.  A rank that found a problem after the others had entered the batch's first
// ncclAllReduce would leave them there for good.
static int board_prepare(pydem_board *b, bool staged, unsigned long long ok)
{
    if (b->n_tiles > 64) { pydem_set_error("pydem_board_run_waves: at most 64 tiles"); return -2; }
    if ((int)b->tl.size() != b->n_tiles) { pydem_set_error("pydem_board_run_waves: pydem_board_set_lines first"); return -2; }
    b->prepared = false;
    if (!b->sched) {
        HIP_TRY(hipMalloc((void **)&b->sched, SCH_WORDS * sizeof(unsigned long long)));
        HIP_TRY(hipHostMalloc((void **)&b->h_sched, SCH_WORDS * sizeof(unsigned long long)));
        HIP_TRY(hipMalloc((void **)&b->scal_tb, 64 * 8 * sizeof(unsigned long long)));
        HIP_TRY(hipHostMalloc((void **)&b->h_progress, 2 * sizeof(unsigned long long)));
        b->h_progress[0] = b->h_progress[1] = 0;
    }
    if (b->cap > b->wcap) {
        if (b->wb) HIP_TRY(hipFree(b->wb));
        HIP_TRY(dev_malloc((void **)&b->wb, (size_t)b->cap * 8));
        b->wcap = b->cap;
        board_drop_graphs(b); b->tables_valid = false;   // (they hold the old staging buffer)
    }
    int64_t most = 1;
    std::vector<int> mine;
    for (int i = 0; i < b->n_tiles; i++) {
        pydem_board::TileLines &T = b->tl[(size_t)i];
        const pydem_board_desc &D = b->h_desc[(size_t)i];
        most = std::max<int64_t>(most, 2 * (int64_t)D.n + 2 * (int64_t)D.m);
        if (!T.tile || !((ok >> i) & 1ull)) continue;
        mine.push_back(i);
        if (!tile_edge_queue_ready(T.tile)) { pydem_set_error("pydem_board_run_waves: tile %d cannot queue its rounds", i); return -3; }
        for (const auto &ln : T.where)             // (a line registered after the tile's condensed graph was built)
            if (!tile_line_watched(T.tile, ln.first, ln.second)) { pydem_set_error("pydem_board_run_waves: tile %d: a board line is not watched", i); return -3; }
    }
    static int eval_blocks = -1;                     // PYDEM_EVAL_BLOCKS: blocks of 1024 threads per tile in the evaluation (default 16)
    if (eval_blocks < 0) { const char *e = getenv("PYDEM_EVAL_BLOCKS"); eval_blocks = e ? std::max(1, std::min(atoi(e), 256)) : 16; }
    b->g_eval = (int)std::min<int64_t>(cdiv(most, 1024), eval_blocks);
    hipStream_t bs = b->stream;
    const bool rebuild = !b->tables_valid || b->wave_ok != ok;
    if (rebuild) {
        HIP_TRY(hipStreamSynchronize(bs));
        board_drop_graphs(b);
        b->tables_valid = false;
        if (b->q_tiles) { HIP_TRY(hipFree(b->q_tiles)); b->q_tiles = nullptr; }
        if (b->q_lines) { HIP_TRY(hipFree(b->q_lines)); b->q_lines = nullptr; }
        if (b->q_unpack) { HIP_TRY(hipFree(b->q_unpack)); b->q_unpack = nullptr; }
        // staging layout of the collective: every line of every tile at a byte offset (8-byte aligned), areas as doubles, masks as
        // bytes -- a fifth of the board's size.  Needs the line list of the other ranks' tiles too (pydem_board_set_lines).
        std::vector<std::vector<int64_t>> st_off((size_t)b->n_tiles);
        std::vector<pydem_unpack_line> hu;
        int64_t st = 0;
        b->q_staged_ok = true;
        for (int i = 0; i < b->n_tiles; i++) {
            const pydem_board::TileLines &T = b->tl[(size_t)i];
            if (T.size > 0 && T.layout.empty()) b->q_staged_ok = false;
            for (const auto &L : T.layout) {
                st_off[(size_t)i].push_back(st);
                pydem_unpack_line U; U.st = st; U.abs = T.mb_start + L.rel; U.count = L.count; U.bytes = L.bytes; U.tile = i;
                hu.push_back(U);
                st += (L.count * L.bytes + 7) & ~(int64_t)7;
            }
        }
        b->q_stage_bytes = st;
        if (st > b->wcap * 8) b->q_staged_ok = false;
        const size_t qb = tile_edge_queue_desc_bytes();
        std::vector<char> hq(qb * std::max<size_t>(mine.size(), 1));
        std::vector<pydem_pack_line_q> hl;
        b->q_nper = 1;
        for (size_t k = 0; k < mine.size(); k++) {
            const int i = mine[k];
            pydem_board::TileLines &T = b->tl[(size_t)i];
            int64_t nper = 0;
            PYDEM_TRY(tile_edge_queue_desc(T.tile, hq.data() + k * qb, b->sched + SCH_WAVE, i, b->sched + SCH_ROUND + i, b->sched + SCH_NWAVES, &nper));
            b->q_nper = std::max(b->q_nper, nper);
            if (T.h_lines.size() != T.layout.size()) b->q_staged_ok = false;
            for (size_t l = 0; l < T.h_lines.size(); l++) {
                const pydem_pack_line &P = T.h_lines[l];
                pydem_pack_line_q Q; Q.P = P; Q.abs = T.mb_start + P.rel; Q.tile = i; Q.pad = 0;
                Q.st = l < st_off[(size_t)i].size() ? st_off[(size_t)i][l] : 0;
                hl.push_back(Q);
            }
        }
        b->q_count = (int)mine.size(); b->q_nlines = (int)hl.size(); b->q_nunpack = (int)hu.size();
        if (b->q_count) {
            HIP_TRY(hipMalloc(&b->q_tiles, hq.size()));
            HIP_TRY(hipMemcpy(b->q_tiles, hq.data(), hq.size(), hipMemcpyHostToDevice));
        }
        if (b->q_nlines) {
            HIP_TRY(hipMalloc(&b->q_lines, hl.size() * sizeof(pydem_pack_line_q)));
            HIP_TRY(hipMemcpy(b->q_lines, hl.data(), hl.size() * sizeof(pydem_pack_line_q), hipMemcpyHostToDevice));
        }
        if (b->q_nunpack) {
            HIP_TRY(hipMalloc(&b->q_unpack, hu.size() * sizeof(pydem_unpack_line)));
            HIP_TRY(hipMemcpy(b->q_unpack, hu.data(), hu.size() * sizeof(pydem_unpack_line), hipMemcpyHostToDevice));
        }
        b->tables_valid = true; b->wave_ok = ok;
    }
    if (staged && !b->q_staged_ok) { pydem_set_error("pydem_board_run_waves: with a communicator every tile needs its line list (pydem_board_set_lines, also for tiles of other ranks)"); return -3; }
    b->q_mine = mine;
    b->prepared = true; b->prepared_ok = ok; b->prepared_staged = staged;
    return 0;
}

// `staged` != 0: the batch will sum its staging buffer over the ranks (a communicator, or the caller's exchange)
int pydem_board_prepare_waves(pydem_board *b, int staged, unsigned long long ok_tiles)
{
    HIP_TRY(hipSetDevice(b->device));
    return board_prepare(b, staged != 0, ok_tiles);
}

// wait for the batch on stream s, at most PYDEM_EDGE_TIMEOUT seconds (default 300; 0 = for ever): a rank whose schedule
// disagrees with the others' would otherwise sit in a collective without a word
static int board_wait(pydem_board *b, hipStream_t s, pydem_comm *c)
{
    static double limit_s = -1.0;
    if (limit_s < 0.0) { const char *e = getenv("PYDEM_EDGE_TIMEOUT"); limit_s = e ? atof(e) : 300.0; if (limit_s < 0.0) limit_s = 0.0; }
    if (limit_s == 0.0) { HIP_TRY(hipStreamSynchronize(s)); return 0; }
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    unsigned spins = 0;
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e == hipSuccess) return 0;
        if (e != hipErrorNotReady) { pydem_set_error("pydem_board_run_waves: %s", hipGetErrorString(e)); return -1; }
        if ((++spins & 1023u) == 0) {
            clock_gettime(CLOCK_MONOTONIC, &t1);
            const double dt = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
            if (dt > limit_s) {
                const unsigned long long nw = b->h_progress ? b->h_progress[0] : 0ull, wave = b->h_progress ? b->h_progress[1] : 0ull;
                pydem_set_error("pydem_board_run_waves: the batch did not come back within %.0f s (PYDEM_EDGE_TIMEOUT); the last wave selected on this rank was "
                                "wave %llu of the batch, members 0x%llx%s", limit_s, nw, wave,
                                c ? " -- the ranks' schedules disagree or a rank left the job; the RCCL communicator is aborted" : "");
                if (c && c->comm) { (void)ncclCommAbort(c->comm); c->comm = nullptr; }
                return -7;
            }
        }
    }
}

// exchange(ctx, op, buf, n): op 0 = sum the n BYTES at buf over the ranks, byte by byte, in place; op 1 = maximum of the n
// doubles at buf over the ranks, in place.  Returns 0.  (The staging buffer of a batch summed by the caller instead of RCCL:
// processes that share one GPU -- RCCL refuses two ranks on one device -- and transports without RCCL.  One host look per
// wave: this is the tested restatement of the queued path, not the fast one.)
int pydem_board_run_waves_ex(pydem_board *b, pydem_comm *c, int k_waves, unsigned long long *state, unsigned long long *scal_out,
                             pydem_exchange_fn exchange, void *ctx)
{
    HIP_TRY(hipSetDevice(b->device));
    if (k_waves < 1 || k_waves > SCH_ROUND - SCH_LOG) { pydem_set_error("pydem_board_run_waves: 1..64 waves per batch"); return -2; }
    if (c && exchange) { pydem_set_error("pydem_board_run_waves: a communicator or an exchange function, not both"); return -2; }
    if (c && !c->comm) { pydem_set_error("pydem_board_run_waves: the communicator was aborted"); return -6; }
    const bool staged = c != nullptr || exchange != nullptr;
    const unsigned long long ok = state[SCH_OK];
    // (callers that did not prepare -- a single process -- get it here; with several ranks the verdict must have been agreed on)
    if (!(b->prepared && b->prepared_ok == ok && b->prepared_staged == staged && b->tables_valid)) PYDEM_TRY(board_prepare(b, staged, ok));
    pydem_board_list Lst, LstFull;
    Lst.n = b->n_tiles; LstFull.n = b->n_tiles;
    for (int i = 0; i < b->n_tiles; i++) { Lst.tile[i] = i; Lst.full[i] = 0; LstFull.tile[i] = i; LstFull.full[i] = 1; state[SCH_ROUND + i] = 0; }
    const std::vector<int> &mine = b->q_mine;
    for (int i : mine) state[SCH_ROUND + i] = tile_edge_round_counter(b->tl[(size_t)i].tile);
    const int g_eval = b->g_eval;
    // The whole batch runs on ONE stream, the board's (the tiles' streams are idle: every host-driven wave ends with a
    // synchronised evaluation), and a wave is a fixed handful of launches whatever the number of tiles: the rounds and the
    // gathers go through device tables (entry = tile), gated by the wave's member word.  One tile per rank and eight tiles of
    // one process take the same path (the second is what a single GPU can test).
    hipStream_t bs = b->stream;
    // One wave, in two parts around the collective.  Everything a launch needs is read on the device (member words, round
    // stamps), so the parts can be captured once and replayed wave after wave.
    auto issue = [&](int part) -> int {
        if (part == 0) {
            hipLaunchKernelGGL(k_sched_select, dim3(1), dim3(64), 0, bs, b->sched, b->scal, b->n_tiles, b->h_progress);
            // (a tie-break wave: the strips of its tile once more, rule :274 everywhere; the numbers go to a scratch row)
            hipLaunchKernelGGL(k_board_eval, dim3(g_eval, b->n_tiles), dim3(1024), 0, bs, b->mb, b->desc, LstFull, b->scal_tb, b->sched + SCH_TB);
            if (staged) HIP_TRY(hipMemsetAsync(b->wb, 0, (size_t)b->q_stage_bytes, bs));
            PYDEM_TRY(stage_edge_rounds_queued(bs, b->q_tiles, b->q_count, b->q_nper));
            // (without a collective the lines go straight to the board: the staging buffer exists to be summed over the ranks)
            if (b->q_nlines > 0)
                hipLaunchKernelGGL(k_board_pack_gated, dim3(16, b->q_nlines), dim3(256), 0, bs, (const pydem_pack_line_q *)b->q_lines, b->q_nlines,
                                   b->mb, reinterpret_cast<unsigned char *>(b->wb), staged ? 1 : 0, b->sched + SCH_WAVE);
        } else {
            if (staged && b->q_nunpack > 0)
                hipLaunchKernelGGL(k_board_unpack_gated, dim3(16, b->q_nunpack), dim3(256), 0, bs, (const pydem_unpack_line *)b->q_unpack, b->q_nunpack,
                                   reinterpret_cast<const unsigned char *>(b->wb), b->mb, b->sched + SCH_WAVE);
            hipLaunchKernelGGL(k_board_eval, dim3(g_eval, b->n_tiles), dim3(1024), 0, bs, b->mb, b->desc, Lst, b->scal, b->sched + SCH_AFFECTED);
        }
        return 0;
    };
    // ---- the wave as a captured graph: one launch per wave instead of seven.  PYDEM_EDGE_GRAPH=0 / 1 forces plain launches /
    // graphs; the default is graphs WITHOUT a collective and plain launches with one: measured equal once a wave was down to
    // six launches (27.5 against 26.0 ms for 126 waves), and graph capture around ncclAllReduce is one more thing that has
    // never run with more than one rank
    static int graph_env = -2;
    if (graph_env == -2) { const char *e = getenv("PYDEM_EDGE_GRAPH"); graph_env = e ? (atoi(e) != 0) : -1; }
    const int use_graph = exchange ? 0 : (graph_env >= 0 ? graph_env : (c ? 0 : 1));
    const int n_parts = c ? 2 : 1;                   // (without a collective both parts are one graph)
    if (use_graph && !b->graph_failed && (!b->wave_exec[0] || b->wave_comm != (c != nullptr) || b->wave_stream != bs)) {
        board_drop_graphs(b);
        HIP_TRY(hipStreamSynchronize(bs));
        for (int part = 0; part < n_parts && !b->graph_failed; part++) {
            hipGraph_t g = nullptr;
            if (hipStreamBeginCapture(bs, hipStreamCaptureModeRelaxed) != hipSuccess) { b->graph_failed = true; break; }
            int rc = issue(part);
            if (rc == 0 && !c) rc = issue(1);
            const hipError_t ec = hipStreamEndCapture(bs, &g);
            if (rc != 0 || ec != hipSuccess || !g || hipGraphInstantiate(&b->wave_exec[part], g, nullptr, nullptr, 0) != hipSuccess)
                b->graph_failed = true;
            if (g) (void)hipGraphDestroy(g);
        }
        if (b->graph_failed) { (void)hipGetLastError(); board_drop_graphs(b); }
        b->wave_comm = c != nullptr; b->wave_stream = bs;
    }
    const bool graphs = use_graph && !b->graph_failed && b->wave_exec[0];
    state[SCH_STOP] = 0; state[SCH_NWAVES] = 0; state[SCH_CHECK] = 0; state[SCH_WAVE] = 0; state[SCH_AFFECTED] = 0;
    state[SCH_TB] = 0; state[SCH_NTB] = 0; state[SCH_TBLOG] = 0;
    if (state[SCH_LIMIT] > (unsigned long long)k_waves) state[SCH_LIMIT] = (unsigned long long)k_waves;
    memcpy(b->h_sched, state, SCH_WORDS * sizeof(unsigned long long));
    b->h_progress[0] = b->h_progress[1] = 0;
    HIP_TRY(hipStreamSynchronize(b->stream));       // (the evaluations of the host-driven waves ran there)
    HIP_TRY(hipMemcpyAsync(b->sched, b->h_sched, SCH_WORDS * sizeof(unsigned long long), hipMemcpyHostToDevice, bs));
    if (exchange && b->h_stage_bytes < (size_t)b->q_stage_bytes + 64) {
        if (b->h_stage) { HIP_TRY(hipHostFree(b->h_stage)); b->h_stage = nullptr; b->h_stage_bytes = 0; }
        HIP_TRY(hipHostMalloc((void **)&b->h_stage, (size_t)b->q_stage_bytes + 64));
        b->h_stage_bytes = (size_t)b->q_stage_bytes + 64;
    }
    for (int w = 0; w < k_waves; w++) {
        if (graphs) HIP_TRY(hipGraphLaunch(b->wave_exec[0], bs)); else PYDEM_TRY(issue(0));
        if (c) NCCL_TRY(ncclAllReduce(b->wb, b->wb, (size_t)b->q_stage_bytes, ncclUint8, ncclSum, c->comm, bs));   // (disjoint fills: x + 0 byte by byte)
        bool stop_now = false;
        if (exchange) {
            // the caller's sum, and before it the check RCCL cannot make: every rank must have selected the SAME wave (the
            // schedule runs on every rank from replicated numbers) -- maximum and minimum of (members, wave count, stop) agree
            HIP_TRY(hipMemcpyAsync(b->h_stage, b->wb, (size_t)b->q_stage_bytes, hipMemcpyDeviceToHost, bs));
            HIP_TRY(hipMemcpyAsync(b->h_sched, b->sched, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, bs));
            PYDEM_TRY(board_wait(b, bs, nullptr));
            const unsigned long long wv = b->h_sched[SCH_WAVE], nwv = b->h_sched[SCH_NWAVES], stp = b->h_sched[SCH_STOP];
            double probe[8] = {(double)(wv & 0xFFFFFFFFull), -(double)(wv & 0xFFFFFFFFull), (double)(wv >> 32), -(double)(wv >> 32),
                               (double)nwv, -(double)nwv, (double)stp, -(double)stp};
            if (exchange(ctx, 1, probe, 8) != 0) { pydem_set_error("pydem_board_run_waves: the exchange function failed"); return -6; }
            for (int q = 0; q < 8; q += 2)
                if (probe[q] != -probe[q + 1]) {
                    pydem_set_error("pydem_board_run_waves: the ranks disagree on wave %d of the batch (this rank: members 0x%llx, %llu waves so far, stop %llu)",
                                    w, wv, nwv, stp);
                    return -8;
                }
            if (exchange(ctx, 0, b->h_stage, b->q_stage_bytes) != 0) { pydem_set_error("pydem_board_run_waves: the exchange function failed"); return -6; }
            HIP_TRY(hipMemcpyAsync(b->wb, b->h_stage, (size_t)b->q_stage_bytes, hipMemcpyHostToDevice, bs));
            stop_now = stp != 0;                     // (agreed on above: every rank leaves the batch here)
        }
        if (graphs) { if (c) HIP_TRY(hipGraphLaunch(b->wave_exec[1], bs)); }
        else PYDEM_TRY(issue(1));
        if (stop_now) break;
    }
    // the metrics of the last wave's neighbourhood, as the next selection would read them
    hipLaunchKernelGGL(k_sched_settle, dim3(1), dim3(64), 0, bs, b->sched, b->scal, b->n_tiles);
    HIP_TRY(hipMemcpyAsync(b->h_sched, b->sched, SCH_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost, bs));
    HIP_TRY(hipMemcpyAsync(b->h_scal, b->scal, (size_t)b->n_tiles * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, bs));
    HIP_TRY(hipGetLastError());
    PYDEM_TRY(board_wait(b, bs, c));
    memcpy(state, b->h_sched, SCH_WORDS * sizeof(unsigned long long));
    memcpy(scal_out, b->h_scal, (size_t)b->n_tiles * 8 * sizeof(unsigned long long));
    for (int i : mine) tile_edge_rounds_ran(b->tl[(size_t)i].tile, (int)state[SCH_NWAVES]);
    state[SCH_GRAPH] = graphs ? 1 : 0;
    b->prepared = false;                             // (a tile's state may change before the next batch: the caller prepares again or run_waves does)
    return 0;
}

int pydem_board_run_waves(pydem_board *b, pydem_comm *c, int k_waves, unsigned long long *state, unsigned long long *scal_out)
{
    return pydem_board_run_waves_ex(b, c, k_waves, state, scal_out, nullptr, nullptr);
}

// RCCL ranks behind a communicator (1 for a single process)
int pydem_comm_count(pydem_comm *c, int *count)
{
    if (!c || !c->comm) { pydem_set_error("pydem_comm_count: no communicator"); return -2; }
    NCCL_TRY(ncclCommCount(c->comm, count));
    return 0;
}

// may the rounds of this tile be queued (pydem_board_run_waves)?  1 / 0
int pydem_tile_edge_queue_ready(pydem_tile *t) { return t && tile_edge_queue_ready(t) ? 1 : 0; }

// debugging / tests: a copy of the board
int pydem_board_download(pydem_board *b, double *out)
{
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipMemcpyAsync(out, b->mb, (size_t)b->cap * 8, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return 0;
}

}  // extern "C"
