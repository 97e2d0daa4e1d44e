// uca.hip -- K3 section/proportion, K4 implicit flow graph, K5 frontier sweep, K6 TWI.
//
// Replaces, for one tile (reference pydem/dem_processing.py unless noted):
//   K3  _calc_uca_section_proportion :1021-1070                      (int8 facet index: bit-exact)
//   K4  _mk_adjacency_matrix :1072-1153 + _mk_connectivity :1155-1267 -- the reference builds an
//       NN x NN scipy CSC matrix (and its CSR twin, :879); here the graph stays implicit: every
//       cell has <= 2 regular out-edges given by (section, proportion) and the keep-filter of
//       :1136-1137, so a cell's regular in-edges are one bit per 8-neighbour (`inmask`); the few
//       non-adjacent pit->drain edges live in a small side list (PitGraph).
//   K5  _calc_uca_chunk :864-987 + the native loop cyutils._drain_area (pydem/cyfuncs/cyutils.pyx
//       :119-187): level-synchronous topological sweep.  The reference pushes area[i]*w along
//       out-edges and re-scans all N cells four times per round; here each frontier cell PULLS
//       a0 + sum(area[u]*w(u->c)) over its in-edges in a fixed order (no floating-point atomics,
//       so results are run-to-run deterministic) and then decrements its targets' in-degree;
//       the next frontier is compacted with a wavefront ballot/popcount prefix and one atomic
//       per wave.  On a DAG both formulations compute the same fixed point; only the order of
//       the additions differs (<= a few ulp; tolerance 1e-6 relative per BASELINE.json).
//   K6  calc_twi :1647-1677.
// Everything here is bounded by HBM (or by launch/atomic latency in the sweep's long tail).
#include "internal.h"
#include <math.h>

#define PI_D 3.141592653589793

namespace {

enum : uint8_t { GF_OUT1 = 1, GF_OUT2 = 2, GF_PIT_OUT = 4, GF_PIT_IN = 8 };

// 8-neighbour offsets in ascending cell-id order: NW N NE W E SW S SE
__device__ __constant__ const int NB_DI[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
__device__ __constant__ const int NB_DJ[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
// a neighbour at offset d drains into the centre iff its section is one of these two facets
// (its e1 -- for cardinal offsets -- or e2 -- for diagonal offsets -- points back at the centre)
__device__ __constant__ const int NB_S0[8] = {6, 5, 4, 0, 3, 0, 1, 2};
__device__ __constant__ const int NB_S1[8] = {7, 6, 5, 7, 4, 1, 2, 3};

// ------------------------------------------------------------------------------- K3
__global__ __launch_bounds__(256) void k_section_proportion(const double *__restrict__ dir,
                                                            const uint8_t *__restrict__ flats,
                                                            const double *__restrict__ sec_theta, int64_t NN, int m,
                                                            int8_t *__restrict__ section, double *__restrict__ prop)
{
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < NN; c += (int64_t)gridDim.x * blockDim.x) {
        const double theta = sec_theta[c / m];
        const double d = dir[c];
        int sec0 = (int)(int8_t)(int)floor(d / PI_D * 2.0);                      // :1035
        const double quadrant = d - PI_D / 2.0 * (double)sec0;                   // :1037
        const int mod2 = sec0 & 1;                                               // python % 2
        int sec = sec0 * 2 + ((quadrant > theta) && mod2 == 0) + ((quadrant > (PI_D / 2 - theta)) && mod2 == 1);  // :1040-1043
        sec = (int)(int8_t)sec;
        double p = NAN;
        const bool I1 = sec == 0 || sec == 1 || sec == 4 || sec == 5;           // :1050
        const double cth = PI_D / 2 - theta;
        if (I1 && quadrant <= theta) p = quadrant / theta;                       // :1052-1053
        if (I1 && quadrant > theta) p = (quadrant - theta) / cth;                // :1054-1056
        if (!I1 && quadrant <= cth) p = quadrant / cth;                          // :1057-1059
        if (!I1 && quadrant > cth) p = (quadrant - cth) / theta;                 // :1060-1062
        if (flats[c]) { sec = -1; p = NAN; }                                     // :1064-1065
        if (sec == 8) sec = 0;                                                   // :1067
        const int a = (sec & 1) ? -1 : 1;                                        // adjust[section], negative wraps
        prop[c] = (1 + a) / 2.0 - (double)a * p;                                 // :1068
        section[c] = (int8_t)sec;
    }
}

// ------------------------------------------------------------------------------- K4
// keep-filter of _mk_adjacency_matrix (:1136-1137)
__device__ __forceinline__ bool keep_edge(double w, double z_to, double z_from)
{
    return !isnan(w) && (w > 1e-8) && (z_to <= z_from);
}

__global__ __launch_bounds__(256) void k_build_graph(const int8_t *__restrict__ section, const double *__restrict__ prop,
                                                     const double *__restrict__ elev, int n, int m,
                                                     uint8_t *__restrict__ inmask, uint8_t *__restrict__ gflags,
                                                     int32_t *__restrict__ level, uint8_t *__restrict__ todo0,
                                                     uint8_t *__restrict__ todo_work, double *__restrict__ corner_sums)
{
    const int64_t NN = (int64_t)n * m;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < NN; c += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(c / m), j = (int)(c - (int64_t)i * m);
        const int s = section[c];
        const double p = prop[c], z = elev[c];
        uint8_t gf = 0;
        double outsum = 0.0;
        if (s >= 0 && s <= 7) {
            const int i1 = i + fe1r(s), j1 = j + fe1c(s), i2 = i + fe2r(s), j2 = j + fe2c(s);
            const double w2 = 1 - p;                                             // :1082
            if (i1 >= 0 && i1 < n && j1 >= 0 && j1 < m && keep_edge(p, elev[(int64_t)i1 * m + j1], z)) { gf |= GF_OUT1; outsum += p; }
            if (i2 >= 0 && i2 < n && j2 >= 0 && j2 < m && keep_edge(w2, elev[(int64_t)i2 * m + j2], z)) { gf |= GF_OUT2; outsum += w2; }
        }
        uint8_t im = 0;
        double insum = 0.0;
#pragma unroll
        for (int d = 0; d < 8; d++) {
            const int ui = i + NB_DI[d], uj = j + NB_DJ[d];
            if (ui < 0 || ui >= n || uj < 0 || uj >= m) continue;
            const int64_t u = (int64_t)ui * m + uj;
            const int su = section[u];
            if (su != NB_S0[d] && su != NB_S1[d]) continue;
            const bool cardinal = (NB_DI[d] == 0) || (NB_DJ[d] == 0);
            const double pu = prop[u];
            const double w = cardinal ? pu : 1 - pu;
            if (keep_edge(w, z, elev[u])) { im |= (uint8_t)(1u << d); insum += w; }
        }
        inmask[c] = im;
        gflags[c] = gf;
        level[c] = im ? 0x7fffffff : 0;   // sources (no in-edges) are round 0
        // inlet-edge detection (_calc_uca_chunk :909-930); interior cells are never 'todo'
        const bool top = i == 0, bot = i == n - 1, left = j == 0, right = j == m - 1;
        if (top || bot || left || right) {
            const double TOL = 1e-2;
            bool td = false;
            const bool has_out = outsum > TOL;
            // assignment order of the reference: left, right, top, bottom (later overwrite earlier)
            if (left) td = has_out && (s == 6 || s == 7 || s == 0 || s == 1);
            if (right) td = has_out && (s == 2 || s == 3 || s == 4 || s == 5);
            if (top) td = has_out && (s == 4 || s == 5 || s == 6 || s == 7);
            if (bot) td = has_out && (s == 0 || s == 1 || s == 2 || s == 3);
            if ((top || bot) && (left || right)) {
                const int q = (top ? 0 : 2) + (left ? 0 : 1);
                corner_sums[q * 3 + 0] = outsum;
                corner_sums[q * 3 + 1] = insum;
                corner_sums[q * 3 + 2] = td ? 1.0 : 0.0;
            }
            if (isnan(z)) td = false;                                            // :935
            todo0[c] = td;
            todo_work[c] = td;
        }
    }
}

// pit edges contribute to in-degrees, flags and the corner sums
__global__ void k_graph_add_pits(const int32_t *__restrict__ src, const int32_t *__restrict__ dst,
                                 const double *__restrict__ w, int64_t ne, int n, int m,
                                 uint8_t *gflags, int32_t *level, double *corner_sums)
{
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += (int64_t)gridDim.x * blockDim.x) {
        const int32_t s = src[e], d = dst[e];
        level[d] = 0x7fffffff;   // a pit drains here: not a source
        atomicOr((unsigned *)(gflags + (s & ~3)), (unsigned)GF_PIT_OUT << (8 * (s & 3)));
        atomicOr((unsigned *)(gflags + (d & ~3)), (unsigned)GF_PIT_IN << (8 * (d & 3)));
        const int corners[4] = {0, m - 1, (n - 1) * m, (n - 1) * m + m - 1};
        for (int q = 0; q < 4; q++) {
            if (s == corners[q]) atomicAdd(&corner_sums[q * 3 + 0], w[e]);
            if (d == corners[q]) atomicAdd(&corner_sums[q * 3 + 1], w[e]);
        }
    }
}

// corner pass-through rule (:920-930): todo |= (outsum > TOL) | (insum < TOL)
__global__ void k_corner_todo(const double *__restrict__ corner_sums, const double *__restrict__ elev, int n, int m,
                              uint8_t *todo0, uint8_t *todo_work)
{
    const int q = threadIdx.x;
    if (q >= 4) return;
    const int64_t corners[4] = {0, m - 1, (int64_t)(n - 1) * m, (int64_t)(n - 1) * m + m - 1};
    const int64_t c = corners[q];
    bool td = corner_sums[q * 3 + 2] != 0.0;
    td = td || (corner_sums[q * 3 + 0] > 1e-2) || (corner_sums[q * 3 + 1] < 1e-2);
    if (isnan(elev[c])) td = false;
    todo0[c] = td;
    todo_work[c] = td;
}

// ------------------------------------------------------------------------------- K5
// Frontier bookkeeping without per-edge atomics.  level[c] is the round in which cell c is
// processed (0 for sources, LEVEL_INF while unknown).  When cell u (level r) is final it looks at
// each target t: t is ready for round r+1 iff every upstream cell of t has level <= r, and exactly
// one of t's upstream cells with level == r -- the one with the largest cell id -- appends t to
// the next frontier and stamps level[t] = r+1.  All level-r stamps were written by the previous
// launch, so the test reads only settled values; the in-degree counters (and their ~1.4 device
// atomics per cell) of a textbook Kahn sweep disappear.  The frontier itself is appended through
// an LDS staging buffer: wavefront ballot + popcount prefix, one LDS atomic per wave, and one
// global atomic per ~1.5k cells when the buffer is flushed.
constexpr int32_t LEVEL_INF = 0x7fffffff;
constexpr int STAGE_CAP = 8192;      // LDS staging entries per block (32 KiB)
constexpr int STAGE_FLUSH = STAGE_CAP - 512;    // flush when fewer than 2*256 slots remain

struct SweepArgs {
    const uint8_t *inmask, *gflags;
    const int8_t *section;
    const double *prop, *a0;     // a0[i] = dX2[i]*dY2[i]
    double *area;
    uint8_t *todo_work;
    int32_t *level;
    int n, m;
    // pit side lists
    const int32_t *pit_src, *pit_dst;   // out-edges sorted by src
    int64_t n_pit;
    const int32_t *pin_dst, *pin_src; const double *pin_w;   // in-edges sorted by (dst, src)
};

__device__ __forceinline__ int64_t lower_bound_i32(const int32_t *a, int64_t n, int32_t key)
{
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}

struct Stage {
    int32_t buf[STAGE_CAP];
    int32_t cnt;
    int32_t base;
};

// wave-aggregated append into the block's LDS staging buffer
__device__ __forceinline__ void stage_push(Stage &S, bool pred, int32_t cell)
{
    const unsigned long long bal = __ballot(pred);
    if (bal == 0ull) return;
    const int lane = (int)__lane_id();
    const int leader = __ffsll((long long)bal) - 1;
    int32_t base = 0;
    if (lane == leader) base = atomicAdd(&S.cnt, (int32_t)__popcll(bal));
    base = __shfl(base, leader);
    if (pred) S.buf[base + __popcll(bal & ((1ull << lane) - 1ull))] = cell;
}

// block-wide: move the staged cells to the global frontier (call from uniform control flow)
__device__ __forceinline__ void stage_flush(Stage &S, int32_t *__restrict__ qn, int32_t *cn, bool force)
{
    __syncthreads();
    const int32_t c = S.cnt;
    if (c > 0 && (force || c > STAGE_FLUSH)) {
        if (threadIdx.x == 0) S.base = atomicAdd(cn, c);
        __syncthreads();
        const int32_t b = S.base;
        for (int32_t k = threadIdx.x; k < c; k += blockDim.x) qn[b + k] = S.buf[k];
        __syncthreads();
        if (threadIdx.x == 0) S.cnt = 0;
    }
    __syncthreads();
}

// does upstream cell u (level r) own the hand-off of target t to round r+1?
__device__ __forceinline__ bool owns_target(const SweepArgs &A, int32_t t, int32_t u, int32_t r)
{
    const uint8_t im = A.inmask[t];
    int32_t owner = -1;
    bool ready = true;
#pragma unroll
    for (int d = 0; d < 8; d++) {
        if (im & (1u << d)) {
            const int32_t v = t + NB_DI[d] * A.m + NB_DJ[d];
            const int32_t lv = A.level[v];
            ready = ready && (lv <= r);
            if (lv == r) owner = v > owner ? v : owner;
        }
    }
    if (A.gflags[t] & GF_PIT_IN) {
        for (int64_t e = lower_bound_i32(A.pin_dst, A.n_pit, t); e < A.n_pit && A.pin_dst[e] == t; e++) {
            const int32_t v = A.pin_src[e];
            const int32_t lv = A.level[v];
            ready = ready && (lv <= r);
            if (lv == r) owner = v > owner ? v : owner;
        }
    }
    return ready && owner == u;
}

// after cell c (level r) is final: hand its ready targets to the next frontier
__device__ __forceinline__ void release_targets(const SweepArgs &A, Stage &S, bool active, int32_t c, uint8_t gf, int s,
                                                int32_t r, int32_t *__restrict__ qn, int32_t *cn)
{
    const int m = A.m;
    int32_t t1 = -1, t2 = -1;
    if (active && (gf & GF_OUT1)) t1 = c + fe1r(s) * m + fe1c(s);
    if (active && (gf & GF_OUT2)) t2 = c + fe2r(s) * m + fe2c(s);
    const bool r1 = t1 >= 0 && owns_target(A, t1, c, r);
    const bool r2 = t2 >= 0 && owns_target(A, t2, c, r);
    if (r1) A.level[t1] = r + 1;
    if (r2) A.level[t2] = r + 1;
    stage_push(S, r1, t1);
    stage_push(S, r2, t2);
    if (active && (gf & GF_PIT_OUT)) {                                           // rare: drained pit
        for (int64_t e = lower_bound_i32(A.pit_src, A.n_pit, c); e < A.n_pit && A.pit_src[e] == c; e++) {
            const int32_t t = A.pit_dst[e];
            if (owns_target(A, t, c, r)) { A.level[t] = r + 1; qn[atomicAdd(cn, 1)] = t; }
        }
    }
}

// round 0: every cell without in-edges is a source (ids = colsum == 0, :882-883): area = dX2*dY2
__global__ __launch_bounds__(256) void k_sweep_sources(SweepArgs A, int32_t *__restrict__ qn, int32_t *cn, int32_t *nsrc)
{
    __shared__ Stage S;
    if (threadIdx.x == 0) S.cnt = 0;
    __syncthreads();
    const int64_t NN = (int64_t)A.n * A.m;
    int32_t mine = 0;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < NN; base += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = base + threadIdx.x;
        bool src = false;
        uint8_t gf = 0; int s = -1;
        if (c < NN) {
            src = A.level[c] == 0;
            if (src) { gf = A.gflags[c]; A.area[c] = A.a0[c / A.m]; s = A.section[c]; mine++; }
        }
        release_targets(A, S, src, (int32_t)c, gf, s, 0, qn, cn);
        stage_flush(S, qn, cn, false);
    }
    stage_flush(S, qn, cn, true);
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(nsrc, mine);
}

// rounds >= 1: pull, store, release
__device__ __forceinline__ void process_cell(const SweepArgs &A, Stage &S, bool active, int32_t c, int32_t r,
                                             int32_t *__restrict__ qn, int32_t *cn)
{
    uint8_t gf = 0; int s = -1;
    if (active) {
        const int m = A.m;
        const int i = c / m;
        const uint8_t im = A.inmask[c];
        gf = A.gflags[c];
        s = A.section[c];
        double acc = A.a0[i];                                                   // :885, :901
        uint8_t td = A.todo_work[c];
        // regular in-edges in ascending source id: NW N NE W E SW S SE
#pragma unroll
        for (int d = 0; d < 8; d++) {
            if (im & (1u << d)) {
                const int32_t u = c + NB_DI[d] * m + NB_DJ[d];
                const bool cardinal = (NB_DI[d] == 0) || (NB_DJ[d] == 0);
                const double pu = A.prop[u];
                const double w = cardinal ? pu : 1 - pu;
                acc += A.area[u] * w;                                           // cyutils.pyx:163
                td |= A.todo_work[u];                                           // :165 (float taint -> bool)
            }
        }
        if (gf & GF_PIT_IN) {
            for (int64_t e = lower_bound_i32(A.pin_dst, A.n_pit, c); e < A.n_pit && A.pin_dst[e] == c; e++) {
                acc += A.area[A.pin_src[e]] * A.pin_w[e];
                td |= A.todo_work[A.pin_src[e]];
            }
        }
        A.area[c] = acc;
        A.todo_work[c] = td;
    }
    release_targets(A, S, active, c, gf, s, r, qn, cn);
}

// one frontier round; counters rotate over 3 slots: in = r%3, out = (r+1)%3, (r+2)%3 is cleared
__global__ __launch_bounds__(256) void k_sweep_round(SweepArgs A, const int32_t *__restrict__ qc, int32_t *__restrict__ qn,
                                                     int32_t *cnt3, int r, int32_t *total)
{
    __shared__ Stage S;
    const int32_t nq = cnt3[r % 3];
    if (blockIdx.x == 0 && threadIdx.x == 0) { cnt3[(r + 2) % 3] = 0; if (nq) { atomicAdd(total, nq); atomicAdd(total + 2, 1); } }
    if (nq == 0) return;
    if (threadIdx.x == 0) S.cnt = 0;
    __syncthreads();
    int32_t *cn = &cnt3[(r + 1) % 3];
    for (int32_t base = blockIdx.x * blockDim.x; base < nq; base += gridDim.x * blockDim.x) {
        const int32_t q = base + threadIdx.x;
        const bool active = q < nq;
        const int32_t c = active ? qc[q] : 0;
        process_cell(A, S, active, c, r, qn, cn);
        stage_flush(S, qn, cn, false);
    }
    stage_flush(S, qn, cn, true);
}

__global__ void k_row_area(const double *__restrict__ dX2, const double *__restrict__ dY2, int n, double *a0)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a0[i] = dX2[i] * dY2[i];                                          // :885
}

// finalisation of _calc_uca_chunk (:966-980): NaN on flats, edge_done = ~edge_todo etc.
__global__ __launch_bounds__(256) void k_uca_finalize(double *__restrict__ uca, const uint8_t *__restrict__ flats,
                                                      const uint8_t *__restrict__ todo_work, const double *__restrict__ elev,
                                                      uint8_t *__restrict__ edge_done, int64_t NN, int apply_limit, double limit)
{
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < NN; c += (int64_t)gridDim.x * blockDim.x) {
        double a = uca[c];
        if (flats[c]) { a = NAN; uca[c] = a; }                                   // :972
        bool dn = !todo_work[c];                                                 // :974
        if (isnan(elev[c])) dn = true;                                           // :975
        if (apply_limit && a > limit) dn = true;                                 // :977-980
        edge_done[c] = dn;
    }
}

// ------------------------------------------------------------------------------- K6
__global__ __launch_bounds__(256) void k_twi(const double *__restrict__ uca, const double *__restrict__ mag,
                                             double *__restrict__ twi, int64_t NN, double min_slope, int lim_uca,
                                             double uca_cap, int lim_twi, double twi_cap)
{
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < NN; c += (int64_t)gridDim.x * blockDim.x) {
        double t = uca[c];
        if (lim_uca && t > uca_cap) t = uca_cap;                                 // :1663-1665
        t = log(t / (mag[c] + min_slope));                                       // :1667
        if (lim_twi && t > twi_cap) t = twi_cap;                                 // :1669-1672
        twi[c] = t;
    }
}

int grid_for(int64_t work, int cap) { const int64_t g = cdiv(work, 256); return (int)(g < cap ? (g > 0 ? g : 1) : cap); }

}  // namespace

int stage_section_graph(pydem_tile *t, const pydem_options *opt)
{
    const int n = (int)t->n, m = (int)t->m;
    PYDEM_TRY(tile_alloc(t, &t->inmask, (size_t)t->NN));
    PYDEM_TRY(tile_alloc(t, &t->gflags, (size_t)t->NN + 4));
    PYDEM_TRY(tile_alloc(t, &t->todo_work, (size_t)t->NN));
    PYDEM_TRY(tile_alloc(t, &t->indeg, (size_t)t->NN));
    HIP_TRY(hipEventRecord(t->ev[0], t->stream));
    const int big = grid_for(t->NN, 8192);
    hipLaunchKernelGGL(k_section_proportion, dim3(big), dim3(256), 0, t->stream, t->dir, t->flats, t->sec_theta, t->NN, m,
                       t->section, t->prop);
    HIP_TRY(hipEventRecord(t->ev[1], t->stream));
    t->tm.n_pit_edges = 0; t->tm.n_pits_undrained = 0; t->tm.pits_ms = 0;
    t->pits.n_edges = 0; t->pits.n_raw = 0;
    if (opt->drain_pits) PYDEM_TRY(stage_pits(t, opt));
    HIP_TRY(hipEventRecord(t->ev[2], t->stream));
    double *corner_sums = (double *)(t->counters + 16);                          // 12 doubles inside the counter block
    HIP_TRY(hipMemsetAsync(t->counters, 0, 64 * sizeof(int32_t), t->stream));
    HIP_TRY(hipMemsetAsync(t->edge_todo, 0, (size_t)t->NN, t->stream));
    HIP_TRY(hipMemsetAsync(t->todo_work, 0, (size_t)t->NN, t->stream));
    hipLaunchKernelGGL(k_build_graph, dim3(big), dim3(256), 0, t->stream, t->section, t->prop, t->elev, n, m, t->inmask,
                       t->gflags, t->indeg, t->edge_todo, t->todo_work, corner_sums);
    if (t->pits.n_edges > 0)
        hipLaunchKernelGGL(k_graph_add_pits, dim3(grid_for(t->pits.n_edges, 1024)), dim3(256), 0, t->stream, t->pits.src,
                           t->pits.dst, t->pits.w, t->pits.n_edges, n, m, t->gflags, t->indeg, corner_sums);
    hipLaunchKernelGGL(k_corner_todo, dim3(1), dim3(64), 0, t->stream, corner_sums, t->elev, n, m, t->edge_todo, t->todo_work);
    HIP_TRY(hipEventRecord(t->ev[3], t->stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventSynchronize(t->ev[3]));
    float a = 0, b = 0;
    HIP_TRY(hipEventElapsedTime(&a, t->ev[0], t->ev[1]));
    HIP_TRY(hipEventElapsedTime(&b, t->ev[2], t->ev[3]));
    t->tm.graph_ms = a + b;
    return 0;
}

int stage_sweep(pydem_tile *t, const pydem_options *opt)
{
    const int n = (int)t->n, m = (int)t->m;
    PYDEM_TRY(tile_alloc(t, &t->queue[0], (size_t)t->NN));
    PYDEM_TRY(tile_alloc(t, &t->queue[1], (size_t)t->NN));
    PYDEM_TRY(tile_alloc(t, &t->row_area, (size_t)t->n));
    int32_t *cnt3 = t->counters;        // [0..2] rotating frontier sizes
    int32_t *total = t->counters + 3;   // cells processed by rounds >= 1
    int32_t *nsrc = t->counters + 4;    // source cells (round 0)
    int32_t *nrounds = t->counters + 5; // rounds with a non-empty frontier
    HIP_TRY(hipEventRecord(t->ev[0], t->stream));
    HIP_TRY(hipMemsetAsync(t->counters, 0, 16 * sizeof(int32_t), t->stream));
    hipLaunchKernelGGL(k_row_area, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, t->stream, t->dX2, t->dY2, n, t->row_area);
    SweepArgs A;
    A.inmask = t->inmask; A.gflags = t->gflags; A.section = t->section; A.prop = t->prop; A.a0 = t->row_area;
    A.area = t->uca; A.todo_work = t->todo_work; A.level = t->indeg; A.n = n; A.m = m;
    A.pit_src = t->pits.src; A.pit_dst = t->pits.dst; A.n_pit = t->pits.n_edges;
    A.pin_dst = t->pits.in_dst; A.pin_src = t->pits.in_src; A.pin_w = t->pits.in_w;
    hipLaunchKernelGGL(k_sweep_sources, dim3(grid_for(t->NN, 4096)), dim3(256), 0, t->stream, A, t->queue[1], &cnt3[1], nsrc);
    int64_t launches = 1;
    int r = 1;
    int64_t last = t->NN;   // size of the frontier the next round will read (upper bound until first readback)
    for (;;) {
        const int batch = last > 262144 ? 2 : (last > 4096 ? 8 : 64);
        const int grid = grid_for(last, 2048);
        for (int b = 0; b < batch; b++, r++) {
            hipLaunchKernelGGL(k_sweep_round, dim3(grid), dim3(256), 0, t->stream, A, t->queue[r % 2], t->queue[(r + 1) % 2],
                               cnt3, r, total);
            launches++;
        }
        HIP_TRY(hipMemcpyAsync(t->h_counters, t->counters, 16 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        last = t->h_counters[r % 3];
        if (last == 0) break;
        if (r > (1 << 24)) { pydem_set_error("frontier sweep did not terminate"); return -5; }
    }
    const int64_t processed = (int64_t)t->h_counters[3] + t->h_counters[4];
    t->tm.n_unresolved = t->NN - processed;
    t->tm.sweep_kernel_launches = launches;
    double min_area = INFINITY;
    for (int64_t i = 0; i < t->n; i++) { const double a = t->h_dX2[(size_t)i] * t->h_dY2[(size_t)i]; if (a < min_area) min_area = a; }
    hipLaunchKernelGGL(k_uca_finalize, dim3(grid_for(t->NN, 8192)), dim3(256), 0, t->stream, t->uca, t->flats, t->todo_work,
                       t->elev, t->edge_done, t->NN, opt->apply_uca_limit_edges, opt->uca_saturation_limit * 2 * min_area);
    HIP_TRY(hipEventRecord(t->ev[1], t->stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventSynchronize(t->ev[1]));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, t->ev[0], t->ev[1]));
    t->tm.sweep_ms = ms;
    t->tm.sweep_rounds = t->h_counters[5];
    return 0;
}

int stage_twi(pydem_tile *t, const pydem_options *opt)
{
    HIP_TRY(hipEventRecord(t->ev[0], t->stream));
    const double uca_cap = opt->uca_saturation_limit * opt->twi_min_area;
    const double twi_cap = log(opt->uca_saturation_limit * opt->twi_min_area / opt->twi_min_slope);
    hipLaunchKernelGGL(k_twi, dim3(grid_for(t->NN, 8192)), dim3(256), 0, t->stream, t->uca, t->mag, t->twi, t->NN,
                       opt->twi_min_slope, opt->apply_twi_limits_on_uca, uca_cap, opt->apply_twi_limits, twi_cap);
    HIP_TRY(hipEventRecord(t->ev[1], t->stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventSynchronize(t->ev[1]));
    float a = 0;
    HIP_TRY(hipEventElapsedTime(&a, t->ev[0], t->ev[1]));
    t->tm.twi_ms = a;
    return 0;
}
