// internal.h -- shared declarations of libpydem_hip.so (gfx950 only; HIP, no compatibility layers)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>
#include <utility>
#include <initializer_list>
#include "../../include/pydem_hip.h"

// ---- error plumbing -----------------------------------------------------------------------
void pydem_set_error(const char *fmt, ...);
#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            pydem_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
            return -1;                                                                         \
        }                                                                                      \
    } while (0)
#define PYDEM_TRY(expr)            \
    do {                           \
        int r_ = (expr);           \
        if (r_ != 0) return r_;    \
    } while (0)

// ---- facet tables (reference: dem_processing.py:173-193) ------------------------------------
// facet k: first neighbour e1 (cardinal), second neighbour e2 (diagonal); (row offset, col offset)
__host__ __device__ constexpr int fe1r(int k) { return k == 1 || k == 2 ? -1 : (k == 5 || k == 6 ? 1 : 0); }
__host__ __device__ constexpr int fe1c(int k) { return k == 0 || k == 7 ? 1 : (k == 3 || k == 4 ? -1 : 0); }
__host__ __device__ constexpr int fe2r(int k) { return k <= 3 ? -1 : 1; }
__host__ __device__ constexpr int fe2c(int k) { return (k == 0 || k == 1 || k == 6 || k == 7) ? 1 : -1; }

// per-spacing-row table entry: row r of dX/dY (r = 0..n-2)
struct RowTab {
    double dX, dY;    // fence-grid spacing of row r
    double hyp;       // sqrt(dX*dX + dY*dY)            (dem_processing.py:1962 denominator)
    double thA;       // atan2(dY, dX): facets 0,3,4,7  (dem_processing.py:1936)
    double thB;       // atan2(dX, dY): facets 1,2,5,6
    double rdX, rdY, rhyp;   // correctly rounded reciprocals of dX, dY, hyp (Markstein division in the marching stencil)
};

// pit -> drain edges: raw triplets as emitted (the reference's pit_i, pit_j, pit_prop) and two
// sorted views of the edges that survive the adjacency keep-filter
struct PitGraph {
    int32_t *raw_src = nullptr, *raw_dst = nullptr; double *raw_w = nullptr;
    int64_t raw_cap = 0, n_raw = 0;
    int64_t n_edges = 0, sorted_cap = 0;       // kept edges
    int32_t *src = nullptr, *dst = nullptr; double *w = nullptr;            // sorted by (src, dst)
    int32_t *in_src = nullptr, *in_dst = nullptr; double *in_w = nullptr;   // sorted by (dst, src)
    void *sort_buf = nullptr; size_t sort_bytes = 0;                        // keys / indices / radix-sort scratch, kept across calls
};

struct pydem_tile {
    int64_t n = 0, m = 0, NN = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;  // side stream: memory-bound graph kernels that run beside the issue-bound pit search
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_snap = nullptr;
    hipEvent_t ev[8] = {};
    // fields
    double *elev = nullptr, *mag = nullptr, *dir = nullptr, *prop = nullptr, *uca = nullptr, *twi = nullptr;
    uint8_t *flats = nullptr, *edge_todo = nullptr, *edge_done = nullptr, *flat0 = nullptr;
    int8_t *section = nullptr;
    bool have[PYDEM_FIELD_COUNT] = {};
    // spacing
    double *dX = nullptr, *dY = nullptr, *dX2 = nullptr, *dY2 = nullptr;   // device copies
    RowTab *rowtab = nullptr;                                              // [n-1]
    double *sec_theta = nullptr;                                           // [n] theta per row for section/proportion
    double *row_area = nullptr;                                            // [n] dX2*dY2
    std::vector<double> h_dX, h_dY, h_dX2, h_dY2;
    bool spacing_set = false;
    int stencil_exact_only = 0;    // a spacing outside [2^-500, 2^500] (or PYDEM_STENCIL_EXACT=1): the marching stencil keeps to its exact path
    bool elev_f32 = false;          // the resident elevation was uploaded as float32: differences are float32 subtractions (stencil, pit drops)
    int elev_dtype = 0;             // pydem_dtype of the last elevation upload (the conditioning keeps the array's dtype where the reference does)
    // graph / sweep scratch
    uint8_t *inmask = nullptr, *gflags = nullptr, *todo_work = nullptr;   // inmask/gflags: unused since the cinfo word
    double *contrib = nullptr;     // [2*NN] outgoing contributions per cell (double2)
    int32_t *indeg = nullptr, *queue[2] = {nullptr, nullptr}, *labels = nullptr, *flatlist = nullptr;
    int32_t *counters = nullptr;       // device scalars
    int32_t *h_counters = nullptr;     // pinned host mirror
    PitGraph pits;
    // edge-resolution rounds
    int32_t *estamp = nullptr; double *edelta = nullptr, *p_delta = nullptr, *s_data = nullptr;
    uint8_t *p_flags = nullptr, *s_flags = nullptr;
    int32_t eepoch = 0;
    int32_t *eseed = nullptr;       // seed frontier of the current edge round (cell, graph word) pairs
    bool edge_clean = false;        // edge flags / counts are zero and the masks only differ from their defaults on etodo_prev cells
    bool einc_compact = false;      // ... in the compact record form (one 128-byte record per not-done cell)
    void *nd_rec = nullptr; int64_t nd_cap = 0; int32_t nd = 0;
    int64_t circular_cells = -1;    // cells the last sweep found on / below a drainage loop (-1: no sweep ran on this handle)
    int64_t einc_round = 0;         // incremental edge rounds run on this tile so far (stamps of the NaN flood)
    bool einc_ready = false;        // incremental edge rounds: counts / deltas / FINAL flags are live (uca.hip K7i)
    // condensed form of the incremental rounds (uca_cond.inl): the cascade of a round runs on the watched cells only
    std::vector<std::pair<int, int64_t>> watch;     // lines other tiles read (axis, index >= 0); the perimeter is always watched
    size_t watch_built = 0;         // how many of them the live condensed graph covers
    bool cond_live = false;         // (only meaningful while einc_ready)
    bool cond_pending = false;      // rounds ran since the interior last caught up
    int32_t cond_nw = 0, cond_nan_cap = 0;
    void *cond_mem = nullptr; size_t cond_bytes = 0;
    void *cond_node = nullptr, *cond_edge = nullptr; double *cond_slot = nullptr;
    int32_t *cond_q0 = nullptr, *cond_q1 = nullptr, *cond_nanq = nullptr, *cond_cnt = nullptr;
    void *cb_mem[3] = {nullptr, nullptr, nullptr}; size_t cb_bytes[3] = {0, 0, 0};   // scratch of the device build of that graph (uca_cbuild.inl), kept with the tile
    double *h_strip_d = nullptr; uint8_t *h_strip_f = nullptr; size_t h_strip_cap = 0;   // pinned strip staging
    void *h_stage = nullptr; size_t h_stage_bytes = 0;      // pinned host staging of the conditioning stages (tile_pinned)
    int32_t etodo_prev = 0;         // cells whose edge_done byte the previous round cleared (tlist = flatlist)
    double *line_stage = nullptr;   // max(n, m) doubles: staging for column get/set
    void *lines_stage = nullptr; int lines_cap = 0;   // staging for pydem_tile_get_lines
    bool graph_valid = false;   // inmask/gflags/section/prop/pit lists match the resident elev/dir/flats
    void *scratch = nullptr; size_t scratch_bytes = 0;
    int64_t device_bytes = 0;
    pydem_timings tm = {};
};

template <typename T>
int tile_alloc(pydem_tile *t, T **p, size_t count);

// Scratch of the conditioning stages (region records of pydem_fill_flats, reservation planes / footprints / trails of
// pydem_pit_paths: 6 GB + up to 17 GB for the large-window simulations of a 8192^2 tile).  Allocating and freeing that
// per call costs hundreds of milliseconds in the driver, so every device keeps ONE arena that the stages lease for the
// duration of a call (a mutex: conditioning stages of two tiles on one device run one after the other); it grows to the
// largest request seen and is released by pydem_hip_release_scratch().
struct ArenaLease {
    int device = -1;
    void *arena = nullptr;              // the device's arena record (kept here: the destructor must not take the table lock while it holds `busy`)
    char *base = nullptr; size_t bytes = 0, off = 0, want = 0;
    std::vector<void *> extra;          // what did not fit this time (plain allocations, freed when the lease ends)
    bool held = false;
    ~ArenaLease();
};
int arena_acquire(int device, ArenaLease *L);
void *arena_take(ArenaLease *L, size_t bytes);       // 256-byte aligned; nullptr on allocation failure (error set)

int ensure_fields(pydem_tile *t, std::initializer_list<int> fields);
// pinned host memory of at least `bytes` that lives with the tile (grown on demand): every per-round transfer of the
// conditioning stages goes through it -- asynchronous copies from / to pageable memory make the runtime pin and unpin
// the pages behind the caller's back, and the NEXT GPU call then waits ~20 ms for that housekeeping
// device blocks through the per-device free lists of tile.hip (planes of destroyed tiles are reused)
void *plane_take(int device, size_t bytes);
void plane_give(int device, void *q);
hipError_t dev_malloc(void **p, size_t bytes);      // hipMalloc; on failure the free lists are emptied and the call repeated
int tile_pinned(pydem_tile *t, size_t bytes, void **out);

// stage entry points implemented in the .hip files
int stage_stencil(pydem_tile *t);
int stage_flats(pydem_tile *t);
int stage_section_graph(pydem_tile *t, const pydem_options *opt);
int stage_pits(pydem_tile *t, const pydem_options *opt);
int stage_sweep(pydem_tile *t, const pydem_options *opt);
int stage_twi(pydem_tile *t, const pydem_options *opt);
int stage_edge_update(pydem_tile *t, const pydem_options *opt, const double *const data[4], const uint8_t *const done[4],
                      const uint8_t *const todo[4]);
int stage_edge_round_inc(pydem_tile *t, const pydem_options *opt, const double *const data[4], const uint8_t *const done[4],
                         const uint8_t *const todo[4]);
int stage_edge_flush(pydem_tile *t);
// condensed incremental rounds: the interior of the tile catches up with the watched lines (no-op otherwise); reads of the
// edge fields that are not watched lines call it first
int stage_edge_catchup(pydem_tile *t);
// queued waves of the fix-up (comm.hip: pydem_board_run_waves): a condensed round gated by a device word
bool tile_edge_queue_ready(const pydem_tile *t);
size_t tile_edge_queue_desc_bytes();
int tile_edge_queue_desc(pydem_tile *t, void *out, const unsigned long long *gate, int bit, const unsigned long long *round_base,
                         const unsigned long long *round_add, int64_t *nper);
int stage_edge_rounds_queued(hipStream_t s, const void *d_q, int count, int64_t max_nper);
unsigned long long tile_edge_round_counter(const pydem_tile *t);
void tile_edge_rounds_ran(pydem_tile *t, int waves);
void tile_watch_line(pydem_tile *t, int axis, int64_t index);
bool tile_line_watched(const pydem_tile *t, int axis, int64_t index);
int stage_fill_flats(pydem_tile *t, double max_pit_area, int below_sea, double source_tol, int peaks, int pits, int artefacts_only);
int stage_pit_candidates(pydem_tile *t, int below_sea, int64_t *npits);
int stage_pit_candidates_read(pydem_tile *t, int64_t npits, int32_t *cells, double *elev);
int stage_pit_paths(pydem_tile *t, const int32_t *order_host, int64_t npits, int max_iter, int max_dist, double max_dist_XY,
                    int64_t *n_failed, int64_t *iter_used, int64_t *rounds_out, int dtype_mode);
int stage_synth(pydem_tile *t, uint32_t seed, int64_t row0, int64_t col0, int n_oct, int top_shift,
                double zmin, double zrange);
int bench_stencil(pydem_tile *t, int iters, double *avg_ms);

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
