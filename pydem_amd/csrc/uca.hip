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
//       :119-187): topological sweep.  The reference pushes area[i]*w along out-edges and re-scans
//       all N cells four times per round; here each cell PULLS a0 + sum(area[u]*w(u->c)) over its
//       in-edges in a fixed order (no floating-point atomics, so results are run-to-run
//       deterministic).  Default schedule: tile passes (K5b) -- one wavefront per 32x32 tile runs
//       as many level-synchronous rounds as it can on its CU, tiles are re-listed when a finished
//       cell drains into them.  Alternative (PYDEM_SWEEP_MODE=queue): frontier queue rounds with a
//       level-ownership hand-off instead of in-degree atomics.  On a DAG all formulations compute
//       the same fixed point; only the order of the additions differs from the reference's push
//       (<= a few ulp; tolerance 1e-6 relative per BASELINE.json).
//   K7  edge-resolution rounds: calc_uca(uca_init=, edge_init_data=) :724-771 (count-based Kahn on
//       the cells downstream of the seeds).
//   K6  calc_twi :1647-1677.
// The pointwise kernels are bounded by HBM; the sweeps by dependent memory latency times the tiles / cells in flight.
#include "internal.h"
#include <hipcub/hipcub.hpp>      // scans / radix sorts of the device operator build (uca_cbuild.inl)
#include <algorithm>
#include <functional>
#include <memory>
#include <thread>
#include <vector>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define PI_D 3.141592653589793

namespace {

// Per-cell graph word `cinfo` (one 32-bit load tells a thread everything static about a cell and its
// sweep state; the sweep is bound by the number of distinct cache lines it touches per cell):
//   bits 0-7   inmask: which of the 8 neighbours (NW N NE W E SW S SE) drain into this cell
//   bit  8/9   regular out-edge to the facet's first / second neighbour survives the keep-filter
//   bit  10/11 cell has pit out-edges / pit in-edges (side lists)
//   bits 12-14 facet index (section) when bit 8 or 9 is set
//   bits 15-31 level: sweep round in which the cell is processed (CI_LEVEL_INF = not yet known)
constexpr uint32_t CI_OUT1 = 1u << 8, CI_OUT2 = 1u << 9, CI_PIT_OUT = 1u << 10, CI_PIT_IN = 1u << 11;
constexpr int CI_SEC_SHIFT = 12, CI_LEVEL_SHIFT = 15;
constexpr uint32_t CI_LEVEL_INF = 0x1FFFFu, CI_STATIC_MASK = 0x7FFFu;
__device__ __forceinline__ uint32_t ci_level(uint32_t w) { return w >> CI_LEVEL_SHIFT; }
__device__ __forceinline__ int ci_section(uint32_t w) { return (int)((w >> CI_SEC_SHIFT) & 7u); }
__device__ __forceinline__ uint32_t ci_with_level(uint32_t w, uint32_t lv) { return (w & CI_STATIC_MASK) | (lv << CI_LEVEL_SHIFT); }

// 8-neighbour offsets in ascending cell-id order: NW N NE W E SW S SE
__device__ __constant__ const int NB_DI[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
__device__ __constant__ const int NB_DJ[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
// a neighbour at offset d drains into the centre iff its section is one of these two facets
// (its e1 -- for cardinal offsets -- or e2 -- for diagonal offsets -- points back at the centre)
__device__ __constant__ const int NB_S0[8] = {6, 5, 4, 0, 3, 0, 1, 2};
__device__ __constant__ const int NB_S1[8] = {7, 6, 5, 7, 4, 1, 2, 3};

// keep-filter of _mk_adjacency_matrix (:1136-1137)
__device__ __forceinline__ bool keep_edge(double w, double z_to, double z_from)
{
    return !isnan(w) && (w > 1e-8) && (z_to <= z_from);
}

// ------------------------------------------------------------------------------- K3 (+ first half of K4)
// section / proportion, and while both are in registers the cell's two regular out-edges (keep-filter of
// :1136-1137): the graph word leaves this kernel with its out flags and facet index; the in-mask follows
// in k_graph_inmask from the neighbours' words
__global__ __launch_bounds__(256) void k_section_proportion(const double *__restrict__ dir,
                                                            const uint8_t *__restrict__ flats,
                                                            const double *__restrict__ sec_theta, int64_t NN, int n, int m,
                                                            const double *__restrict__ elev,
                                                            int8_t *__restrict__ section, double *__restrict__ prop,
                                                            uint32_t *__restrict__ cinfo)
{
    // rows come from blockIdx.y (grid-stride), columns from blockIdx.x: no 64-bit division per cell
    (void)NN;
    for (int i = blockIdx.y; i < n; i += gridDim.y)
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < m; j += gridDim.x * blockDim.x) {
        const int64_t c = (int64_t)i * m + j;
        const double theta = sec_theta[i];
        const double d = dir[c];
        int sec0 = (int)(int8_t)(int)floor(d / PI_D * 2.0);                      // :1035
        const double quadrant = d - PI_D / 2.0 * (double)sec0;                   // :1037
        const int mod2 = sec0 & 1;                                               // python % 2
        int sec = sec0 * 2 + ((quadrant > theta) && mod2 == 0) + ((quadrant > (PI_D / 2 - theta)) && mod2 == 1);  // :1040-1043
        sec = (int)(int8_t)sec;
        const bool I1 = sec == 0 || sec == 1 || sec == 4 || sec == 5;           // :1050
        const double cth = PI_D / 2 - theta;
        // the four cases of :1052-1062 -- I1: quadrant / theta or (quadrant - theta) / cth, else: quadrant / cth or (quadrant - cth) / theta --
        // exclude each other: operands selected first, ONE division (a wavefront holds cells of all four cases, and the four
        // predicated fp64 divisions were most of this kernel's instructions); a NaN direction fails both tests and stays NaN
        const double t1 = I1 ? theta : cth, t2 = I1 ? cth : theta;
        const bool lo = quadrant <= t1, hi = quadrant > t1;
        const double num = lo ? quadrant : quadrant - t1, den = lo ? t1 : t2;
        double p = (lo || hi) ? num / den : NAN;
        if (flats[c]) { sec = -1; p = NAN; }                                     // :1064-1065
        if (sec == 8) sec = 0;                                                   // :1067
        const int a = (sec & 1) ? -1 : 1;                                        // adjust[section], negative wraps
        const double pf = (1 + a) / 2.0 - (double)a * p;                         // :1068
        prop[c] = pf;
        section[c] = (int8_t)sec;
        uint32_t gf = 0;
        if (sec >= 0 && sec <= 7) {
            const double z = elev[c];
            const int i1 = i + fe1r(sec), j1 = j + fe1c(sec), i2 = i + fe2r(sec), j2 = j + fe2c(sec);
            if (i1 >= 0 && i1 < n && j1 >= 0 && j1 < m && keep_edge(pf, elev[(int64_t)i1 * m + j1], z)) gf |= CI_OUT1;
            if (i2 >= 0 && i2 < n && j2 >= 0 && j2 < m && keep_edge(1 - pf, elev[(int64_t)i2 * m + j2], z)) gf |= CI_OUT2;   // weights [prop, 1 - prop] :1082
        }
        cinfo[c] = gf | ((uint32_t)(sec & 7) << CI_SEC_SHIFT);
    }
}

// ------------------------------------------------------------------------------- K4
// second half of K4: a neighbour drains into this cell iff its facet points here and that out-edge survived the
// keep-filter -- both are in the neighbour's graph word (4 B, row-coalesced) instead of its section, proportion
// and elevation
__global__ __launch_bounds__(256) void k_graph_inmask(const double *__restrict__ prop, const double *__restrict__ elev,
                                                      int n, int m, uint32_t *__restrict__ cinfo, uint8_t *__restrict__ todo0,
                                                      uint8_t *__restrict__ todo_work, double *__restrict__ corner_sums)
{
    for (int i = blockIdx.y; i < n; i += gridDim.y)
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < m; j += gridDim.x * blockDim.x) {
        const int64_t c = (int64_t)i * m + j;
        const uint32_t cw = cinfo[c] & (CI_OUT1 | CI_OUT2 | (7u << CI_SEC_SHIFT));
        uint32_t im = 0;
#pragma unroll
        for (int d = 0; d < 8; d++) {
            const int ui = i + NB_DI[d], uj = j + NB_DJ[d];
            if (ui < 0 || ui >= n || uj < 0 || uj >= m) continue;
            const uint32_t wu = cinfo[(int64_t)ui * m + uj];
            const int su = (int)((wu >> CI_SEC_SHIFT) & 7u);
            const bool cardinal = (NB_DI[d] == 0) || (NB_DJ[d] == 0);
            if ((su == NB_S0[d] || su == NB_S1[d]) && (wu & (cardinal ? CI_OUT1 : CI_OUT2))) im |= 1u << d;
        }
        // sources (no in-edges) are round 0
        cinfo[c] = im | cw | ((im ? CI_LEVEL_INF : 0u) << CI_LEVEL_SHIFT);
        // inlet-edge detection (_calc_uca_chunk :909-930); interior cells are never 'todo'
        const bool top = i == 0, bot = i == n - 1, left = j == 0, right = j == m - 1;
        if (top || bot || left || right) {
            const double TOL = 1e-2;
            const double p = prop[c], z = elev[c];
            double outsum = 0.0;
            if (cw & CI_OUT1) outsum += p;
            if (cw & CI_OUT2) outsum += 1 - p;
            const int s = (cw & (CI_OUT1 | CI_OUT2)) ? ci_section(cw) : -1;     // only cells with an out-edge can be inlets
            bool td = false;
            const bool has_out = outsum > TOL;
            // assignment order of the reference: left, right, top, bottom (later overwrite earlier)
            if (left) td = has_out && (s == 6 || s == 7 || s == 0 || s == 1);
            if (right) td = has_out && (s == 2 || s == 3 || s == 4 || s == 5);
            if (top) td = has_out && (s == 4 || s == 5 || s == 6 || s == 7);
            if (bot) td = has_out && (s == 0 || s == 1 || s == 2 || s == 3);
            if ((top || bot) && (left || right)) {
                double insum = 0.0;
                for (int d = 0; d < 8; d++)
                    if (im & (1u << d)) {
                        const int64_t u = (int64_t)(i + NB_DI[d]) * m + (j + NB_DJ[d]);
                        const bool cardinal = (NB_DI[d] == 0) || (NB_DJ[d] == 0);
                        insum += cardinal ? prop[u] : 1 - prop[u];
                    }
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
                                 uint32_t *cinfo, double *corner_sums)
{
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += (int64_t)gridDim.x * blockDim.x) {
        const int32_t s = src[e], d = dst[e];
        atomicOr(&cinfo[s], CI_PIT_OUT);
        atomicOr(&cinfo[d], CI_PIT_IN | (CI_LEVEL_INF << CI_LEVEL_SHIFT));   // a pit drains here: not a source
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
constexpr int STAGE_CAP = 4096;      // LDS staging entries per block (8 B each: 32 KiB)
constexpr int STAGE_FLUSH = STAGE_CAP - 512;    // flush when fewer than 2*256 slots remain

struct SweepArgs {
    uint32_t *cinfo;
    const double *prop, *a0;     // a0[i] = dX2[i]*dY2[i]
    double *area;
    double2 *contrib;            // per cell: (area*w1, area*w2), negated when the cell carries edge_todo taint
    uint8_t *todo_work;
    int n, m;
    // pit side lists: out-edges sorted by (src, dst), in-edges sorted by (dst, src), block start tables
    const int32_t *pit_src, *pit_dst;
    const int32_t *pin_dst, *pin_src; const double *pin_w;
    int64_t n_pit;
    int dbg;                     // timing experiments only (PYDEM_TILE_DEBUG)
    int32_t qcap;                // frontier queue capacity (entries)
    int32_t *err;                // queue overflow counter
    int32_t *tile_open;          // per 32x32 tile: cells still open after its last visit (INT_MAX pattern: not visited yet)
};

// A frontier entry carries the cell AND its graph word: the round that processes it starts its
// gathers straight from the queue load (one dependent memory round trip less per round -- the long
// tail of the sweep is nothing but such round trips).
struct QE { int32_t c; uint32_t cw; };

struct Stage {
    QE buf[STAGE_CAP];
    int32_t cnt;
    int32_t base;
};

// wave-aggregated append into the block's LDS staging buffer
__device__ __forceinline__ void stage_push(Stage &S, bool pred, int32_t cell, uint32_t cw)
{
    const unsigned long long bal = __ballot(pred);
    if (bal == 0ull) return;
    const int lane = (int)__lane_id();
    const int leader = __ffsll((long long)bal) - 1;
    int32_t base = 0;
    if (lane == leader) base = atomicAdd(&S.cnt, (int32_t)__popcll(bal));
    base = __shfl(base, leader);
    if (pred) { QE q; q.c = cell; q.cw = cw; S.buf[base + __popcll(bal & ((1ull << lane) - 1ull))] = q; }
}

// block-wide: move the staged cells to the global frontier (call from uniform control flow)
__device__ __forceinline__ void stage_flush(const SweepArgs &A, Stage &S, QE *__restrict__ qn, int32_t *cn, bool force)
{
    __syncthreads();
    const int32_t c = S.cnt;
    if (c > 0 && (force || c > STAGE_FLUSH)) {
        if (threadIdx.x == 0) S.base = atomicAdd(cn, c);
        __syncthreads();
        const int32_t b = S.base;
        if ((int64_t)b + c <= A.qcap) { for (int32_t k = threadIdx.x; k < c; k += blockDim.x) qn[b + k] = S.buf[k]; }
        else if (threadIdx.x == 0) atomicAdd(A.err, 1);
        __syncthreads();
        if (threadIdx.x == 0) S.cnt = 0;
    }
    __syncthreads();
}

// Pit side lists without a search: until a cell is processed its slot of the area array is unused, so
// it carries {first in-edge, first out-edge} of the cell (k_pit_stash); the edges of one cell are
// contiguous in the sorted lists and end where the key changes.  Reading the slot costs one load that
// travels with the gathers instead of the block-table lookup plus key scan (3+ dependent loads).
__device__ __forceinline__ int2 pit_stash(const SweepArgs &A, int32_t c) { return reinterpret_cast<const int2 *>(A.area)[c]; }

__global__ void k_pit_stash(const int32_t *__restrict__ pin_dst, const int32_t *__restrict__ pit_src, int64_t ne, double *area)
{
    int32_t *slots = reinterpret_cast<int32_t *>(area);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += (int64_t)gridDim.x * blockDim.x) {
        if (e == 0 || pin_dst[e - 1] != pin_dst[e]) slots[2 * (int64_t)pin_dst[e]] = (int32_t)e;
        if (e == 0 || pit_src[e - 1] != pit_src[e]) slots[2 * (int64_t)pit_src[e] + 1] = (int32_t)e;
    }
}

// graph words of the 8 neighbours of t (row ti, column tj); out-of-tile neighbours read t itself (never used)
__device__ __forceinline__ void load_upstream_words(const SweepArgs &A, int32_t t, int ti, int tj, uint32_t w[8])
{
#pragma unroll
    for (int d = 0; d < 8; d++) {
        const int ii = ti + NB_DI[d], jj = tj + NB_DJ[d];
        const bool ok = ii >= 0 && ii < A.n && jj >= 0 && jj < A.m;
        w[d] = A.cinfo[ok ? t + NB_DI[d] * A.m + NB_DJ[d] : t];
    }
}

// Frontier bookkeeping without per-edge atomics.  The level field of cinfo[c] is the round in which
// cell c is processed (0 for sources, CI_LEVEL_INF while unknown).  When cell u (level r) is final it
// looks at each target t: t is ready for round r+1 iff every upstream cell of t has level <= r, and
// exactly one of t's upstream cells with level == r -- the one with the largest cell id -- appends t
// to the next frontier and stamps level r+1.  All level-r stamps were written by the previous
// launch, so the test reads only settled values; the in-degree counters (and their ~1.4 device
// atomics per cell) of a textbook Kahn sweep disappear.
// `ct` = graph word of t, `w` = graph words of its neighbours (SPEC: all eight were loaded up front;
// otherwise only those in t's in-mask are fetched here), `in_start` = first pit in-edge of t.
template <bool SPEC>
__device__ __forceinline__ bool owns_target(const SweepArgs &A, int32_t t, int32_t u, uint32_t r, uint32_t ct, const uint32_t w[8],
                                            int32_t in_start)
{
    int32_t owner = -1;
    bool ready = true;
#pragma unroll
    for (int d = 0; d < 8; d++) {
        if (ct & (1u << d)) {
            const int32_t v = t + NB_DI[d] * A.m + NB_DJ[d];
            const uint32_t lv = (v == u) ? r : ci_level(SPEC ? w[d] : A.cinfo[v]);
            ready = ready && (lv <= r);
            if (lv == r) owner = v > owner ? v : owner;
        }
    }
    if (ct & CI_PIT_IN) {
        for (int32_t e = in_start;; e += 4) {
            int32_t d4[4], s4[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int64_t idx = e + k < A.n_pit ? e + k : A.n_pit - 1;
                d4[k] = A.pin_dst[idx]; s4[k] = A.pin_src[idx];
            }
            uint32_t l4[4];
            bool v4[4];
            bool more = true;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                more = more && e + k < A.n_pit && d4[k] == t;
                v4[k] = more;
                l4[k] = (more && s4[k] != u) ? ci_level(A.cinfo[s4[k]]) : r;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (v4[k]) {
                    ready = ready && (l4[k] <= r);
                    if (l4[k] == r) owner = s4[k] > owner ? s4[k] : owner;
                }
            }
            if (!more) break;
        }
    }
    return ready && owner == u;
}

// publish a finished cell: area plus the two outgoing contributions (sign carries the todo taint)
__device__ __forceinline__ void publish_cell(const SweepArgs &A, int32_t c, uint32_t cw, double acc, bool td)
{
    A.area[c] = acc;
    double2 o = make_double2(0.0, 0.0);
    if (cw & (CI_OUT1 | CI_OUT2)) {
        const double p = A.prop[c];
        if (cw & CI_OUT1) o.x = acc * p;                                        // area[i] * factor, cyutils.pyx:163
        if (cw & CI_OUT2) o.y = acc * (1 - p);
        if (td) { o.x = -o.x; o.y = -o.y; }
    }
    A.contrib[c] = o;
    if (td) A.todo_work[c] = 1;
}

// sum over the pit in-edges of cell c (sorted by source id), four edges per batch of loads
__device__ __forceinline__ void gather_pit_edges(const SweepArgs &A, int32_t c, int32_t in_start, double &acc, bool &td)
{
    for (int32_t e = in_start;; e += 4) {
        int32_t d4[4], s4[4]; double w4[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int64_t idx = e + k < A.n_pit ? e + k : A.n_pit - 1;
            d4[k] = A.pin_dst[idx]; s4[k] = A.pin_src[idx]; w4[k] = A.pin_w[idx];
        }
        double a4[4]; uint8_t t4[4];
        bool more = true;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            more = more && e + k < A.n_pit && d4[k] == c;
            a4[k] = more ? A.area[s4[k]] : 0.0;
            t4[k] = more ? A.todo_work[s4[k]] : (uint8_t)0;
        }
        more = true;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            more = more && e + k < A.n_pit && d4[k] == c;
            if (more) { acc += a4[k] * w4[k]; td = td || (t4[k] != 0); }
        }
        if (!more) break;
    }
}

// rounds >= 1: pull, store, release.  LOWLAT (small frontiers): every load that only needs the queue
// entry -- in-edge contributions, the targets' graph words, the graph words of ALL their neighbours,
// the pit slots -- is issued in one batch, so a regular cell costs two dependent memory round trips
// per round (queue entry, batch).  Large frontiers are bound by the lines they touch, not by latency,
// and fetch only the neighbours in each target's in-mask.
// push2(pred, t, ct): called by all lanes together for the two regular targets; pushp(t, ct): called
// by single lanes for the (rare) pit targets.
template <bool LOWLAT, typename Push2, typename PushP>
__device__ __forceinline__ void process_cell(const SweepArgs &A, bool active, QE q, uint32_t r, Push2 push2, PushP pushp)
{
    const int32_t c = q.c;
    const uint32_t cw = q.cw;
    int32_t t1 = -1, t2 = -1;
    uint32_t c1 = 0, c2 = 0;
    bool r1 = false, r2 = false;
    int32_t out_start = -1;
    if (active) {
        const int m = A.m;
        const int i = c / m, j = c - i * m;
        const int sct = ci_section(cw);
        uint32_t l1[8], l2[8];
        int2 s1 = make_int2(-1, -1), s2 = make_int2(-1, -1), sc = make_int2(-1, -1);
        if (cw & CI_OUT1) {
            const int dr = fe1r(sct), dc = fe1c(sct);
            t1 = c + dr * m + dc; c1 = A.cinfo[t1];
            if (LOWLAT) { load_upstream_words(A, t1, i + dr, j + dc, l1); s1 = pit_stash(A, t1); }
        }
        if (cw & CI_OUT2) {
            const int dr = fe2r(sct), dc = fe2c(sct);
            t2 = c + dr * m + dc; c2 = A.cinfo[t2];
            if (LOWLAT) { load_upstream_words(A, t2, i + dr, j + dc, l2); s2 = pit_stash(A, t2); }
        }
        if (cw & (CI_PIT_IN | CI_PIT_OUT)) sc = pit_stash(A, c);
        double acc = A.a0[i];                                                   // :885, :901
        // only inlet cells on the tile edge start tainted (:909-930); interior bytes are written later
        bool td = (i == 0 || i == A.n - 1 || j == 0 || j == m - 1) && A.todo_work[c] != 0;
        // regular in-edges in ascending source id: NW N NE W E SW S SE; a cardinal neighbour feeds us
        // through its first slot, a diagonal one through its second
#pragma unroll
        for (int d = 0; d < 8; d++) {
            if (cw & (1u << d)) {
                const int32_t u = c + NB_DI[d] * m + NB_DJ[d];
                const bool cardinal = (NB_DI[d] == 0) || (NB_DJ[d] == 0);
                const double x = cardinal ? A.contrib[u].x : A.contrib[u].y;
                acc += fabs(x);                                                 // cyutils.pyx:163
                td = td || (x < 0);                                             // :165 (float taint -> bool)
            }
        }
        if (cw & CI_PIT_IN) gather_pit_edges(A, c, sc.x, acc, td);
        out_start = sc.y;
        publish_cell(A, c, cw, acc, td);
        if (t1 >= 0) {
            if (!LOWLAT && (c1 & CI_PIT_IN)) s1 = pit_stash(A, t1);
            r1 = owns_target<LOWLAT>(A, t1, c, r, c1, l1, s1.x);
        }
        if (t2 >= 0) {
            if (!LOWLAT && (c2 & CI_PIT_IN)) s2 = pit_stash(A, t2);
            r2 = owns_target<LOWLAT>(A, t2, c, r, c2, l2, s2.x);
        }
        if (r1) A.cinfo[t1] = ci_with_level(c1, r + 1);
        if (r2) A.cinfo[t2] = ci_with_level(c2, r + 1);
    }
    push2(r1, t1, c1);
    push2(r2, t2, c2);
    if (active && (cw & CI_PIT_OUT)) {                                           // rare: drained pit
        for (int32_t e = out_start;; e += 4) {
            int32_t s4[4], t4[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int64_t idx = e + k < A.n_pit ? e + k : A.n_pit - 1;
                s4[k] = A.pit_src[idx]; t4[k] = A.pit_dst[idx];
            }
            bool more = true;
            for (int k = 0; k < 4 && more; k++) {
                more = e + k < A.n_pit && s4[k] == c;
                if (!more) break;
                const int32_t t = t4[k];
                const uint32_t ct = A.cinfo[t];
                const int2 st = pit_stash(A, t);
                uint32_t lw[8];
                load_upstream_words(A, t, t / A.m, t % A.m, lw);
                if (owns_target<true>(A, t, c, r, ct, lw, st.x)) {
                    A.cinfo[t] = ci_with_level(ct, r + 1);
                    pushp(t, ct);
                }
            }
            if (!more) break;
        }
    }
}

#ifdef PYDEM_SWEEP_QUEUE      // the frontier-queue schedule: a diagnostic build only (python -m pydem_amd.build with PYDEM_HIPCC_FLAGS=-DPYDEM_SWEEP_QUEUE)
// one frontier round; counters rotate over 3 slots: in = r%3, out = (r+1)%3, (r+2)%3 is cleared
template <bool LOWLAT>
__global__ __launch_bounds__(256) void k_sweep_round(SweepArgs A, const QE *__restrict__ qc, QE *__restrict__ qn,
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
        QE qe; qe.c = 0; qe.cw = 0;
        if (active) qe = qc[q];
        process_cell<LOWLAT>(A, active, qe, (uint32_t)r,
                             [&](bool pred, int32_t t, uint32_t ct) { stage_push(S, pred, t, ct); },
                             [&](int32_t t, uint32_t ct) {
                                 const int32_t slot = atomicAdd(cn, 1);
                                 if (slot < A.qcap) { QE e; e.c = t; e.cw = ct; qn[slot] = e; }
                                 else atomicAdd(A.err, 1);
                             });
        stage_flush(A, S, qn, cn, false);
    }
    stage_flush(A, S, qn, cn, true);
}

// Small frontiers: ONE workgroup runs round after round without going back to the host.  A kernel
// boundary makes every first access of the next round a trip across the fabric (the per-XCD L2s are
// only coherent at kernel boundaries), ~2 us per dependent load and 4+ of them per round; inside one
// workgroup the frontier, the graph words and the contributions it wrote a round ago come from its
// own CU's caches and __syncthreads() is the only synchronisation.  The last ~600 rounds of a
// 16384^2 tile are rivers of a few hundred cells.  Stops when the frontier is empty or outgrows
// SWEEP_SMALL_CAP; reports the round it stopped at and what it processed.
constexpr int SWEEP_SMALL_CAP = 1024;

__global__ __launch_bounds__(1024) void k_sweep_small(SweepArgs A, QE *q0, QE *q1, int32_t *cnt3, int r_start, int r_max,
                                                      int32_t *total, int32_t *state)
{
    __shared__ int s_next;
    int r = r_start;
    int32_t nq = cnt3[r % 3];
    int32_t done = 0, rounds = 0;
    while (nq > 0 && nq <= SWEEP_SMALL_CAP && r < r_max) {
        if (threadIdx.x == 0) s_next = 0;
        __syncthreads();
        const QE *qc = (r % 2) ? q1 : q0;
        QE *qn = (r % 2) ? q0 : q1;
        for (int32_t base = 0; base < nq; base += blockDim.x) {
            const int32_t q = base + threadIdx.x;
            const bool active = q < nq;
            QE qe; qe.c = 0; qe.cw = 0;
            if (active) qe = qc[q];
            auto push = [&](int32_t t, uint32_t ct) {
                const int32_t slot = atomicAdd(&s_next, 1);
                if (slot < A.qcap) { QE e; e.c = t; e.cw = ct; qn[slot] = e; }
                else atomicAdd(A.err, 1);
            };
            process_cell<true>(A, active, qe, (uint32_t)r, [&](bool pred, int32_t t, uint32_t ct) { if (pred) push(t, ct); }, push);
        }
        __syncthreads();
        done += nq; rounds++;
        nq = s_next;
        r++;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        cnt3[r % 3] = nq; cnt3[(r + 1) % 3] = 0; cnt3[(r + 2) % 3] = 0;
        atomicAdd(total, done); atomicAdd(total + 2, rounds);
        state[0] = r;
    }
}

#endif  // PYDEM_SWEEP_QUEUE

// ------------------------------------------------------------------------------- K5b
// Tile passes.  A queue round moves every flow path by ONE cell per kernel boundary and re-streams
// scattered 64 B lines for every cell it touches.  Here ONE WAVEFRONT owns a 32x32 tile for a pass and
// runs as many level-synchronous rounds as it can without leaving the CU: a cell is final once all
// its upstream cells are final -- those inside the tile become so during the pass, cells of the
// one-cell halo only if an earlier pass finished them.  Only the BOOKKEEPING lives in LDS (one word
// per cell -- graph bits + open-upstream count --, a final-cell bitmap for the setup, the ready ring: 4.9 KB per tile, so 32
// tiles are in flight per CU in the full passes and hide each other's latency); areas and contributions go straight to their global arrays, where
// the same wavefront finds them again in its caches a round later (a workgroup-scope fence orders a
// round's stores before the next round's loads).  A pass is either over all tiles (pass 1) or over
// the tiles LISTED by the previous pass: whenever a finished cell drains into another tile, that
// tile is listed for the next pass.  Flow paths advance by one tile per pass instead of one cell
// per round.  Cells finished in pass p carry level p.  The arithmetic per cell (gather order,
// products) is identical to process_cell().
#ifndef PYDEM_TH
#define PYDEM_TH 32
#endif
constexpr int TT = 32;                  // tile width (cells per row: two rows per wavefront load, a 34-bit row of the final bitmap)
constexpr int TH = PYDEM_TH, HH = TH + 2;  // tile height (32 or 64 rows)
static_assert(TH == 32 || TH == 64, "tile height");
// sp half-word of a cell: bits 14-15 state (0 open, 1 final / outside, 2 finished in this pass), bit 13 "waits for
// a cell another tile must finish", bits 0-12 upstream cells of this tile that are still open
constexpr uint32_t SP_STATE_SHIFT = 14, SP_BLOCKED = 1u << 13;
#ifndef PYDEM_STG_B
#define PYDEM_STG_B 4
#endif
constexpr int TILE_RING = TH == 32 ? 256 : 512;  // ready-list ring (a push that does not fit is dropped: the cell stays open with a zero
                                // count, the tile lists itself and the next pass picks the cell up in its setup)

struct TileW {
    uint32_t cs[TH * TT];           // per cell of the tile: high half = static graph bits, low half = sp (state / blocked / open-upstream
                                    // count); one LDS word per cell: a count-down is a plain 32-bit atomic, setup and rounds read both
                                    // halves at once.  The halo ring only exists as bits of `fin`.
    union {
        unsigned long long fin[HH]; // staging and setup: bit lj of row li (halo coordinates) = the cell was final before this pass
                                    // (or lies outside the grid)
        double a0[TH];              // rounds: cell area of the tile's rows
    };
    uint16_t list[TILE_RING];       // ready cells in the order they became ready
    int tail;                       // list end (monotonic; slot = index % TILE_RING)
    int limit;                      // first list index whose push was dropped (INT_MAX: none): the pass stops there
};                                  // 4888 bytes: eight 4-tile workgroups per CU

__device__ __forceinline__ uint16_t &sp_of(TileW &L, int idx) { return reinterpret_cast<uint16_t *>(L.cs)[2 * idx]; }
__device__ __forceinline__ uint32_t sp_state(const TileW &L, int idx) { return (L.cs[idx] >> SP_STATE_SHIFT) & 3u; }

// count-down of cell idx; returns the old half-word (a count that is decremented is >= 1: no borrow into the graph bits)
__device__ __forceinline__ uint32_t sp_dec(TileW &L, int idx) { return atomicSub(&L.cs[idx], 1u) & 0xFFFFu; }

__device__ __forceinline__ void tile_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// orders LDS traffic of the wavefront only (the rounds have no global stores to wait for)
__device__ __forceinline__ void tile_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

constexpr int TILE_PEND = 48;    // woken tiles a wavefront collects before it appends them to the next pass's list
struct TileNext {            // LISTED passes: tiles that must run again in the next pass
    int32_t *flag;           // per tile: last pass it was listed for
    int32_t *list, *count;
};

// contribution of in-edge d (0..7 = NW N NE W E SW S SE, the order of the in-mask bits) into cell c: one 8-byte load,
// the cardinal neighbours hand over their first share, the diagonal ones their second
__device__ __forceinline__ const double *in_edge_ptr(const SweepArgs &A, int32_t c, int m, int d)
{
    const int q = d + (d >> 2);                   // 0..8 with the centre (4) skipped
    const int di = (q * 11) >> 5;                 // q / 3
    const int32_t u = c + (di - 1) * m + (q - 3 * di - 1);
    return reinterpret_cast<const double *>(A.contrib + u) + (((0x5A >> d) & 1) ? 0 : 1);
}
__device__ __forceinline__ double in_edge(const SweepArgs &A, int32_t c, int m, int d) { return *in_edge_ptr(A, c, m, d); }

// Addresses of a tile visit.  The visits issue half of their cycles (profiles/r04_pmc_sq_hot.csv), and a third of a round's
// vector instructions were 64-bit address arithmetic: cell id (a quarter-rate 32-bit multiply) -> sign extension -> shift ->
// 64-bit add, per access.  The tile is wavefront-uniform, so the five planes get one SCALAR base each -- the address of the
// halo's first cell (i0 - 1, j0 - 1), which for a tile on the grid's edge lies in front of the plane and is only ever
// used with offsets of cells that exist -- and a lane addresses a cell by `h` = halo row * m + halo column (24-bit
// multiply: full rate; stage_sweep refuses tiles wider than 2^24 columns), scaled to a 32-bit unsigned byte offset: the
// loads and stores take the SGPR base + VGPR offset form.
struct TileBase {
    const char *cinfo, *area, *contrib, *prop, *todo;
    int32_t org;             // cell id of the halo's first cell (negative on the first row / column of the grid)
};
__device__ __forceinline__ TileBase tile_base(const SweepArgs &A, int i0, int j0)
{
    TileBase B;
    const int64_t org = (int64_t)(i0 - 1) * A.m + (j0 - 1);
    B.org = (int32_t)org;
    B.cinfo = reinterpret_cast<const char *>(A.cinfo) + org * 4;
    B.area = reinterpret_cast<const char *>(A.area) + org * 8;
    B.contrib = reinterpret_cast<const char *>(A.contrib) + org * 16;
    B.prop = reinterpret_cast<const char *>(A.prop) + org * 8;
    B.todo = reinterpret_cast<const char *>(A.todo_work) + org;
    return B;
}
template <typename T>
__device__ __forceinline__ T ld_off(const char *base, uint32_t off) { return *reinterpret_cast<const T *>(base + (size_t)off); }
template <typename T>
__device__ __forceinline__ void st_off(const char *base, uint32_t off, T v) { *reinterpret_cast<T *>(const_cast<char *>(base) + (size_t)off) = v; }
__device__ __forceinline__ uint32_t halo_cell(int li, int lj, int m) { return __umul24((uint32_t)li, (uint32_t)m) + (uint32_t)lj; }

// byte offsets of the eight in-edges relative to a cell's entry of the contribution plane (one LDS table per workgroup,
// filled at kernel start): neighbour d (0..7 = NW N NE W E SW S SE), the cardinal ones hand over their first share, the
// diagonal ones their second
__device__ __forceinline__ void fill_nbr16(int32_t *nbr16, int m)
{
    if (threadIdx.x < 8) {
        const int d = threadIdx.x, q = d + (d >> 2), di = q / 3, dj = q - 3 * di;
        nbr16[d] = ((di - 1) * m + (dj - 1)) * 16 + (((0x5A >> d) & 1) ? 0 : 8);
    }
}
__device__ __forceinline__ double in_edge_h(const TileBase &B, const int32_t *nbr16, uint32_t h, int d)
{
#ifdef PYDEM_EXP_AREAONLY      // timing experiment (taint is lost: results differ): the share formed from the upstream cell's area and proportion
    const uint32_t off16 = h * 16u + (uint32_t)nbr16[d];
    const uint32_t uo = (off16 >> 4) * 8u;
    const double ua = ld_off<double>(B.area, uo), up = ld_off<double>(B.prop, uo);
    return (off16 & 8u) ? ua * (1 - up) : ua * up;
#else
    return ld_off<double>(B.contrib, h * 16u + (uint32_t)nbr16[d]);        // (the neighbour lies in the halo: the sum is >= 0)
#endif
}

// staging of a tile visit: the graph words of the tile (high half of L.cs, state bit "final before this pass" in the low
// half) and the final bitmap of tile + halo
// (`hw`, when given: the graph words of the halo ring as well -- the symbolic visit K5f looks for its inlets there; ring
// position: top row 0..TT+1, bottom row TT+2.., left column 2(TT+2).., right column 2(TT+2)+TH..)
__device__ __forceinline__ void tile_stage(const SweepArgs &A, const TileBase &B, TileW &L, uint32_t pass, int i0, int j0, int lane,
                                           uint32_t *hw = nullptr)
{
    const int n = A.n, m = A.m;
    const int half = lane >> 5, l32 = lane & 31;
    constexpr int NSET = TH * TT / 64;
    // ---- stage the graph words of tile + halo by rows (two 32-cell rows per wavefront load, the halo ring in three
    // loads), a few loads of a lane in flight at a time: the passes are latency-bound.  The "final before this pass"
    // bits come out of wave ballots, one 64-bit word per row.
    auto stage_word = [&](int gi, int gj) -> uint32_t {                  // 0xFFFFFFFF: outside the grid
        return (gi >= 0 && gi < n && gj >= 0 && gj < m) ? A.cinfo[(int64_t)gi * m + gj] : 0xFFFFFFFFu;
    };
    auto final_before = [&](uint32_t w) -> bool {                        // outside the grid: nothing drains from there
        if (w == 0xFFFFFFFFu) return true;
        const uint32_t lv = ci_level(w);
        return lv >= 1 && lv < pass;
    };
    unsigned long long colL, colR;                                                  // bit r: the halo cell left / right of tile row r is final
    {
        const uint32_t wr = stage_word(half ? i0 + TH : i0 - 1, j0 + l32);          // top / bottom halo row
        uint32_t wk = 0xFFFFFFFFu;
        if (lane < 4) wk = stage_word((lane & 2) ? i0 + TH : i0 - 1, (lane & 1) ? j0 + TT : j0 - 1);   // corners
        if (hw) {
            hw[(half ? TT + 2 : 0) + l32 + 1] = wr;
            if (lane < 4) hw[((lane & 2) ? TT + 2 : 0) + ((lane & 1) ? TT + 1 : 0)] = wk;
        }
        if (TH == 32) {
            const uint32_t wc = stage_word(i0 + l32, half ? j0 + TT : j0 - 1);      // left / right halo column
            if (hw) hw[2 * (TT + 2) + (half ? TH : 0) + l32] = wc;
            const unsigned long long bc = __ballot(final_before(wc));
            colL = bc & 0xFFFFFFFFull; colR = bc >> 32;
        } else {
            const uint32_t wl = stage_word(i0 + lane, j0 - 1), wq = stage_word(i0 + lane, j0 + TT);
            colL = __ballot(final_before(wl)); colR = __ballot(final_before(wq));
        }
        const bool fr = final_before(wr), fk = lane < 4 && final_before(wk);
        const unsigned long long br = __ballot(fr), bk = __ballot(fk);
        if (lane == 0) {
            L.fin[0] = ((br & 0xFFFFFFFFull) << 1) | (bk & 1ull) | (((bk >> 1) & 1ull) << 33);
            L.fin[HH - 1] = ((br >> 32) << 1) | ((bk >> 2) & 1ull) | (((bk >> 3) & 1ull) << 33);
        }
    }
    constexpr int STG_B = PYDEM_STG_B;
#pragma unroll 1
    for (int kb = 0; kb < NSET; kb += STG_B) {
        uint32_t wst[STG_B];
#pragma unroll
        for (int k = 0; k < STG_B; k++) wst[k] = stage_word(i0 + 2 * (kb + k) + half, j0 + l32);
#pragma unroll
        for (int k = 0; k < STG_B; k++) {
            const int r = 2 * (kb + k);                                  // this load: local rows r + 1 (lanes 0-31), r + 2
            const bool f = final_before(wst[k]);
            L.cs[lane + 64 * (kb + k)] = ((wst[k] == 0xFFFFFFFFu ? 0u : (wst[k] & CI_STATIC_MASK)) << 16) | ((f ? 1u : 0u) << SP_STATE_SHIFT);
            const unsigned long long b = __ballot(f);
            if (lane == 0) {
                L.fin[r + 1] = ((b & 0xFFFFFFFFull) << 1) | ((colL >> r) & 1ull) | (((colR >> r) & 1ull) << 33);
                L.fin[r + 2] = ((b >> 32) << 1) | ((colL >> (r + 1)) & 1ull) | (((colR >> (r + 1)) & 1ull) << 33);
            }
        }
    }
}

template <bool LISTED>
__device__ __forceinline__ void sweep_one_tile(const SweepArgs &A, TileW &L, uint32_t pass, int tiles_x, int tid, int lane,
                                               uint8_t *__restrict__ tile_done, int32_t &n_final, const TileNext &N,
                                               int32_t *pend, int &npend, const int32_t *nbr16)
{
    const int by = tid / tiles_x, bx = tid - by * tiles_x;
    const int i0 = by * TH, j0 = bx * TT, n = A.n, m = A.m;
    const TileBase B = tile_base(A, i0, j0);
    const int half = lane >> 5, l32 = lane & 31;         // interior cell k of a lane: row 2k + half, column l32 of the tile
    // ready-list push; `consumed` = entries already processed (ring occupancy = tail - consumed)
    auto push_ready = [&](int cell, int consumed) {
        const int slot = atomicAdd(&L.tail, 1);
        if (slot - consumed < TILE_RING) L.list[slot % TILE_RING] = (uint16_t)cell;
        else atomicMin(&L.limit, slot);
    };
#ifdef PYDEM_TILE_PROF
    const bool prof = (A.dbg & 4) != 0 && (int)pass >= (A.dbg >> 8);      // PYDEM_TILE_DEBUG = 4 + 256 * first pass to account
#else
    constexpr bool prof = false;       // (the phase timers cost registers: build with -DPYDEM_TILE_PROF to use PYDEM_TILE_DEBUG=4)
#endif
    long long tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0;
    int nrounds = 0;
    if (prof) tk0 = wall_clock64();
    if (lane == 0) { L.tail = 0; L.limit = INT32_MAX; }
    const double a0_row = (lane < TH && i0 + lane < n) ? A.a0[i0 + lane] : 0.0;       // lands in LDS once the final bitmap is no longer needed
    constexpr int NSET = TH * TT / 64;
    tile_stage(A, B, L, pass, i0, j0, lane);
    tile_wave_sync();
    if (prof) tk1 = wall_clock64();
    // ---- per-cell setup: how many upstream cells are still open = in-mask bits whose neighbour is not final (three row
    // words of the final bitmap, LDS broadcasts) ...
    uint32_t pitmask = 0;
    int32_t n_open = 0;
#pragma unroll 2
    for (int k = 0; k < NSET; k++) {
        const int li = 2 * k + half + 1, idx = lane + 64 * k;
        const uint32_t w = L.cs[idx];
        if (!((w >> SP_STATE_SHIFT) & 3u)) {                        // open (cells outside the grid were staged as final)
            const unsigned long long f0 = L.fin[li - 1], f1 = L.fin[li], f2 = L.fin[li + 1];
            const uint32_t nf = ((uint32_t)(f0 >> l32) & 7u) | (((uint32_t)(f1 >> l32) & 1u) << 3) |
                                (((uint32_t)(f1 >> (l32 + 2)) & 1u) << 4) | (((uint32_t)(f2 >> l32) & 7u) << 5);
            const uint32_t pend = __popc((w >> 16) & 0xFFu & ~nf);
            sp_of(L, idx) = (uint16_t)pend;                         // the state bits of an open cell are zero: the half-word is the count
            n_open++;
            if (w & (CI_PIT_IN << 16)) pitmask |= 1u << k;
            else if (pend == 0) push_ready(lane + 64 * k, 0);
        }
    }
    // ... then the pit in-edges of the lane's drains: their list offsets in one batch, the edges two loads at a time
    if (pitmask) {
        constexpr int PB = 4;
#pragma unroll 1
        for (int kb = 0; kb < NSET; kb += PB) {
            if (!((pitmask >> kb) & ((1u << PB) - 1u))) continue;
            int32_t e0[PB];
#pragma unroll
            for (int k = 0; k < PB; k++) {
                e0[k] = 0;
                if (pitmask & (1u << (kb + k))) {
                    const int cell = lane + 64 * (kb + k);
                    e0[k] = ld_off<int2>(B.area, halo_cell((cell >> 5) + 1, (cell & 31) + 1, m) * 8u).x;
                }
            }
#pragma unroll
            for (int k = 0; k < PB; k++) {
                if (!(pitmask & (1u << (kb + k)))) continue;
                const int cell = lane + 64 * (kb + k);
                const int32_t c = B.org + (int32_t)halo_cell((cell >> 5) + 1, (cell & 31) + 1, m);
                const int idx = cell;
                uint32_t pend = sp_of(L, idx);
                for (int32_t e = e0[k];; e += 2) {                  // pit -> drain edges are short: most sources sit in this tile
                    const int64_t ia = e < A.n_pit ? e : A.n_pit - 1, ib = e + 1 < A.n_pit ? e + 1 : A.n_pit - 1;
                    const int32_t da = A.pin_dst[ia], db = A.pin_dst[ib], sa = A.pin_src[ia], sb = A.pin_src[ib];
                    const bool va = e < A.n_pit && da == c, vb = va && e + 1 < A.n_pit && db == c;
                    auto blocked = [&](int32_t sc) -> uint32_t {
                        const int si = sc / m - i0, sj = sc % m - j0;
                        if (si >= 0 && si < TH && sj >= 0 && sj < TT)
                            return sp_state(L, si * TT + sj) ? 0u : 1u;          // released on chip when the pit finishes
                        const uint32_t lv = ci_level(A.cinfo[sc]);
                        return (lv >= 1 && lv < pass) ? 0u : SP_BLOCKED;                   // another tile's business: blocked for this pass
                    };
                    uint32_t ba = 0, bb = 0;
                    if (va) ba = blocked(sa);
                    if (vb) bb = blocked(sb);
                    pend = (pend + (ba & 1u) + (bb & 1u)) | ((ba | bb) & SP_BLOCKED);
                    if (!vb) break;
                }
                sp_of(L, idx) = (uint16_t)pend;
                if (pend == 0) push_ready(cell, 0);
            }
        }
    }
    tile_wave_sync();
    if (lane < TH) L.a0[lane] = a0_row;                  // (the final bitmap is dead now)
    tile_wave_sync();
    // ---- rounds: the ready cells [head, tail) are processed, the targets they release are appended
    if (prof) tk2 = wall_clock64();
    int32_t finalized = 0;
    uint32_t wake = 0;                 // bit (dti + 1) * 3 + dtj + 1: a finished cell drains into that neighbour tile
    int head = 0;
    // A lane that finishes a cell and thereby makes one of its targets ready keeps that target for itself ("chain"):
    // the next iteration needs neither the ready list (three dependent LDS operations) nor a load of the contribution it
    // has just stored (a load behind a pending write-through store waits for the store's round trip) -- the share travels
    // in a register.  Along a river this is the whole critical path of a tile visit.  Lanes without a chained cell take
    // the next entries of the ready list.
    int mycell = -1, cd = -1;          // the lane's cell (tile-local id), the in-edge (bit of the in-mask) the chained share arrives by
    double cv = 0.0;                   // the chained share
    for (;;) {
        const int tail = L.tail < L.limit ? L.tail : L.limit;
        const bool need = mycell < 0;
        const unsigned long long nb = __ballot(need);
        const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(nb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nb, 0u));
        const int avail = tail - head, wanted = __popcll(nb);
        if (need && rank < avail) { mycell = L.list[(head + rank) % TILE_RING]; cd = -1; }
        head += wanted < avail ? wanted : avail;
        if (!__ballot(mycell >= 0)) break;
        nrounds++;
        int next = -1, nd = -1;
        double nv = 0.0;
        if (mycell >= 0) {
            const int cell = mycell;
            const int li = (cell >> 5) + 1, lj = (cell & 31) + 1, idx = cell;
            const int gi = i0 + li - 1, gj = j0 + lj - 1;
            const uint32_t h = halo_cell(li, lj, m);                 // the cell in halo coordinates: offsets into the five planes
            const int32_t c = B.org + (int32_t)h;
            const uint32_t cw = L.cs[idx] >> 16;
            // everything that only needs (c, cw): proportion, pit slots, the in-edge contributions (most cells have one
            // or two in-edges: walk the set bits, in ascending order like the reference's pull, four loads in flight)
            double pv = 0.0;
            if (cw & (CI_OUT1 | CI_OUT2)) pv = ld_off<double>(B.prop, h * 8u);
            int2 po = make_int2(0, 0);
            if (cw & (CI_PIT_IN | CI_PIT_OUT)) po = ld_off<int2>(B.area, h * 8u);
            uint32_t mm = cw & 0xFFu;
            double xs[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                xs[q] = 0.0;
                if (mm) {
                    const int d = __ffs(mm) - 1; mm &= mm - 1u;
                    if (d == cd) xs[q] = cv; else xs[q] = in_edge_h(B, nbr16, h, d);
                }
            }
            double a = L.a0[li - 1];
            bool td = (gi == 0 || gi == n - 1 || gj == 0 || gj == m - 1) && ld_off<uint8_t>(B.todo, h) != 0;
#pragma unroll
            for (int q = 0; q < 4; q++) { a += fabs(xs[q]); td = td || (xs[q] < 0); }       // (+0.0 leaves the positive sum as it is)
            while (mm) {                                                                    // five and more in-edges: rare
                const int d = __ffs(mm) - 1; mm &= mm - 1u;
                const double x = (d == cd) ? cv : in_edge_h(B, nbr16, h, d);
                a += fabs(x); td = td || (x < 0);
            }
            if (cw & CI_PIT_IN)
                for (int32_t e = po.x; e < A.n_pit && A.pin_dst[e] == c; e++) {
                    const int32_t sc = A.pin_src[e];
                    a += A.area[sc] * A.pin_w[e];
                    td = td || (A.todo_work[sc] != 0);
                }
            double2 o = make_double2(0.0, 0.0);
            if (cw & CI_OUT1) o.x = a * pv;
            if (cw & CI_OUT2) o.y = a * (1 - pv);
            if (td) { o.x = -o.x; o.y = -o.y; }
            st_off<double>(B.area, h * 8u, a);
#ifndef PYDEM_EXP_NOCONTRIB      // timing experiment: the tile visits without their 16-byte contribution stores
            st_off<double2>(B.contrib, h * 16u, o);
#endif
            if (LISTED) st_off<uint32_t>(B.cinfo, h * 4u, ci_with_level(cw, pass));    // finished in this pass (other tiles treat levels < their pass as final)
            if (td) st_off<uint8_t>(B.todo, h, (uint8_t)1);
            sp_of(L, idx) = (uint16_t)(2u << SP_STATE_SHIFT);
            finalized++;
            // release the targets inside the tile; targets in other tiles may be ready now: their tiles run in the
            // next pass (listing a tile whose cell still waits for somebody else costs one idle staging; whoever
            // finishes last lists it again)
            auto release = [&](int ti, int tj, double share, bool chainable) {      // tile-local coordinates 1..TT when inside
                if (ti >= 1 && ti <= TH && tj >= 1 && tj <= TT) {
                    const int tcell = (ti - 1) * TT + (tj - 1);
                    if (sp_dec(L, tcell) == 1u) {
                        if (chainable && next < 0) {
                            const int q = (li - ti + 1) * 3 + (lj - tj + 1);          // where this cell sits in the target's 3x3 window
                            next = tcell; nd = q - (q > 4 ? 1 : 0); nv = share;
                        } else push_ready(tcell, head);
                    }
                } else if (LISTED) {
                    // (out-edges only exist towards cells of the grid.)  The eight neighbour tiles are woken once, after
                    // the rounds: two dependent returning atomics per cell would sit on every round's critical path
                    const int dti = ti < 1 ? -1 : (ti > TH ? 1 : 0), dtj = tj < 1 ? -1 : (tj > TT ? 1 : 0);
                    if (ti >= 1 - TH && ti <= 2 * TH && tj >= 1 - TT && tj <= 2 * TT) wake |= 1u << ((dti + 1) * 3 + dtj + 1);
                    else {                                            // a pit draining further away than the next tile
                        const int tt = ((i0 + ti - 1) / TH) * tiles_x + (j0 + tj - 1) / TT;
                        if (atomicExch(&N.flag[tt], (int32_t)pass + 1) != (int32_t)pass + 1) N.list[atomicAdd(N.count, 1)] = tt;
                    }
                }
            };
            const int sct = ci_section(cw);
            if (cw & CI_OUT1) release(li + fe1r(sct), lj + fe1c(sct), o.x, true);
            if (cw & CI_OUT2) release(li + fe2r(sct), lj + fe2c(sct), o.y, true);
            if (cw & CI_PIT_OUT)
                for (int32_t e = po.y; e < A.n_pit && A.pit_src[e] == c; e++) {
                    const int32_t dc = A.pit_dst[e];
                    const int ti = dc / m - i0 + 1, tj = dc % m - j0 + 1;
                    // a drain inside the tile that an earlier pass already finished cannot exist (it waits for this pit)
                    release(ti, tj, 0.0, false);
                }
        }
        mycell = next; cd = nd; cv = nv;
        tile_wave_sync();
    }
    // ---- the tile is done when every cell that was open at setup has been finished.  A pass that revisits a tile
    // finishes few cells and stamps their level as it goes; the first pass finishes most of the tile and stamps in rows
    if (prof) tk3 = wall_clock64();
    if (!LISTED)
        for (int k = 0; k < NSET; k++) {
            const int li = 2 * k + half + 1, idx = lane + 64 * k;
            const uint32_t w = L.cs[idx];
            if (((w >> SP_STATE_SHIFT) & 3u) == 2u) A.cinfo[(int64_t)(i0 + li - 1) * m + j0 + l32] = ci_with_level(w >> 16, pass);
        }
    for (int off = 32; off > 0; off >>= 1) { finalized += __shfl_down(finalized, off); n_open += __shfl_down(n_open, off); }
    if (LISTED) {
        // the tiles this visit wakes (lanes 0-8: the neighbours a finished cell drains into; lane 9: the tile itself when
        // its ready ring overflowed) go to the wavefront's pending list; the kernel appends the pending lists to the
        // global one with one add per wavefront / workgroup (tile_list_flush): the list counter is a single address too
        for (int off = 32; off > 0; off >>= 1) wake |= __shfl_xor(wake, off);
        bool win = false;
        int tt = 0;
        if (lane < 9 && ((wake >> lane) & 1u)) { tt = tid + (lane / 3 - 1) * tiles_x + (lane % 3 - 1); win = true; }
        if (lane == 9 && L.limit != INT32_MAX) { tt = tid; win = true; }
        if (win) win = atomicExch(&N.flag[tt], (int32_t)pass + 1) != (int32_t)pass + 1;
        const unsigned long long bw = __ballot(win);
        if (win) pend[npend + __popcll(bw & ((1ull << lane) - 1ull))] = tt;
        npend += __popcll(bw);
    }
    if (lane == 0) {
        n_final += finalized;          // (lane 0's running count: the kernel adds it to the global counter once, see there)
        if (finalized == n_open) tile_done[tid] = 1;
        A.tile_open[tid] = n_open - finalized;
        if (prof) {      // cycles per phase, summed over tiles (PYDEM_TILE_DEBUG=4)
            const long long tk4 = wall_clock64();
            unsigned long long *acc = reinterpret_cast<unsigned long long *>(A.err + 1 + 16);   // counters[32..] region: see stage_sweep
            atomicAdd(acc + 0, (unsigned long long)(tk1 - tk0)); atomicAdd(acc + 1, (unsigned long long)(tk2 - tk1));
            atomicAdd(acc + 2, (unsigned long long)(tk3 - tk2)); atomicAdd(acc + 3, (unsigned long long)(tk4 - tk3));
            atomicAdd(acc + 4, (unsigned long long)nrounds); atomicAdd(acc + 5, 1ull);
            if (finalized == 0) atomicAdd(acc + 6, 1ull);
        }
    }
    tile_wave_sync();
}

#include "uca_sym.inl"

// every tile that is not done yet (LISTED: also lists the tiles of the next pass).  Persistent wavefronts, XCD-contiguous
// bands of tiles: workgroup b runs on XCD b % 8 and its wavefronts take the next tile of that XCD's band from the band's
// counter -- a visit lasts 20 .. 100 rounds, and with four fixed tiles per workgroup the LDS of a workgroup (an eighth of
// the CU's) sat behind its slowest visit (PYDEM_SWEEP_STATIC build: the round-3 mapping, for A/B runs)
#ifndef PYDEM_SWEEP_STEAL
#define PYDEM_SWEEP_STEAL 1          // bands a workgroup works on: its own XCD (8 = the others afterwards: measured 31.1 vs 29.3 ms of sweep -- slower)
#endif
#ifndef PYDEM_FULL_WPB
#define PYDEM_FULL_WPB 4            // wavefronts per workgroup of the two full passes
#endif
constexpr int FWPB = PYDEM_FULL_WPB;
template <bool LISTED>
__global__ __launch_bounds__(64 * FWPB, 32 / FWPB) void k_sweep_tiles(SweepArgs A, uint32_t pass, int tiles_x, int tiles_total,
                                                     uint8_t *__restrict__ tile_done, int32_t *n_final, TileNext N, int32_t *work8)
{
    __shared__ TileW L[FWPB];
    __shared__ int32_t s_pend[FWPB][TILE_PEND];
    __shared__ int32_t s_nbr16[8];
    fill_nbr16(s_nbr16, A.m);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int32_t fin = 0;
    int npend = 0;
    auto flush = [&]() {
        tile_wave_sync();
        int32_t base = 0;
        if (lane == 0) base = atomicAdd(N.count, npend);
        base = __shfl(base, 0);
        if (lane < npend) N.list[base + lane] = s_pend[wave][lane];
        npend = 0;
        tile_wave_sync();
    };
#ifdef PYDEM_SWEEP_STATIC
    (void)work8;
    const int per4 = (gridDim.x >> 3) * 4;
    for (int tid = (blockIdx.x & 7) * per4 + (blockIdx.x >> 3) * 4 + wave, once = 0; once < 1; once++) {
        if (tid < tiles_total && !tile_done[tid])
            sweep_one_tile<LISTED>(A, L[wave], pass, tiles_x, tid, lane, tile_done, fin, N, s_pend[wave], npend, s_nbr16);
    }
#else
    const int per = (tiles_total + 7) >> 3;
    // own band first, then the other XCDs' bands (their tails: a band of rough terrain takes longer than its neighbours)
    for (int st = 0; st < PYDEM_SWEEP_STEAL; st++) {
        const int xcd = (int)((blockIdx.x + st) & 7);
        for (;;) {
            int32_t q = 0;
            if (lane == 0) q = atomicAdd(&work8[xcd], 1);
            q = __builtin_amdgcn_readfirstlane(__shfl(q, 0));
            const int tid = xcd * per + q;
            if (q >= per || tid >= tiles_total) break;
            if (tile_done[tid]) continue;
            sweep_one_tile<LISTED>(A, L[wave], pass, tiles_x, tid, lane, tile_done, fin, N, s_pend[wave], npend, s_nbr16);
            if (LISTED && npend > TILE_PEND - 10) flush();      // (a visit adds at most ten)
        }
    }
#endif
    if (LISTED && npend) flush();
    // the counter of finished cells is a single address: one add per wavefront
    if (lane == 0 && fin) atomicAdd(n_final, fin);
}

// later passes: only the listed tiles (those a finished cell of the previous pass drains into); lists the next ones.  The
// wavefronts take the list entries from a counter (work3[pass % 3]; the counter of the next pass is cleared here).
#ifndef PYDEM_LISTED_OCC
#define PYDEM_LISTED_OCC 6
#endif
#ifndef PYDEM_LISTED_WPB
#define PYDEM_LISTED_WPB 2          // wavefronts (= tiles in flight) per workgroup of the listed passes (same-box A/B: 4: 30.36, 1: 29.83, 2: 29.68 ms of sweep)
#endif
constexpr int LWPB = PYDEM_LISTED_WPB;
__global__ __launch_bounds__(64 * LWPB, PYDEM_LISTED_OCC) void k_sweep_tiles_listed(SweepArgs A, uint32_t pass, int tiles_x, const int32_t *__restrict__ list_in,
                                                            const int32_t *n_in, uint8_t *__restrict__ tile_done, int32_t *n_final,
                                                            TileNext N, int32_t *clear_count, int32_t *work3, SymArgs Y)
{
    __shared__ TileW L[LWPB];
    __shared__ SymLight SL[LWPB];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int32_t nt = *n_in;
#ifdef PYDEM_LISTED_DYNAMIC
    if (blockIdx.x == 0 && threadIdx.x < 8) work3[((pass + 1) % 3) * 8 + threadIdx.x] = 0;         // the next pass's work counters
#endif
    if (blockIdx.x == 0 && threadIdx.x == 0) *clear_count = 0;                                     // the list of the pass after the next one
    // (the tile id is the same for the whole wavefront: as a scalar it keeps the tile's coordinates and every address
    // derived from them out of the vector registers)
    __shared__ int32_t s_pend[LWPB][TILE_PEND];
    __shared__ int32_t s_nbr16[8];
    fill_nbr16(s_nbr16, A.m);
    __syncthreads();
    int32_t fin = 0;               // finished cells of all tiles of this wavefront: one add at the end
    int npend = 0;                 // tiles woken by this wavefront's visits that are not on the global list yet
    auto flush = [&]() {
        tile_wave_sync();
        int32_t base = 0;
        if (lane == 0) base = atomicAdd(N.count, npend);
        base = __shfl(base, 0);
        if (lane < npend) N.list[base + lane] = s_pend[wave][lane];
        npend = 0;
        tile_wave_sync();
    };
#ifndef PYDEM_LISTED_DYNAMIC
    // (taking the list entries from counters like the full passes do was measured and is slower here: 2.74 / 2.18 / 1.68 ms
    // against 2.41 / 1.79 / 1.26 ms for passes 4-6 -- -DPYDEM_LISTED_DYNAMIC keeps the variant)
    (void)work3;
    for (int32_t k = __builtin_amdgcn_readfirstlane(blockIdx.x * LWPB + wave); k < nt; k += gridDim.x * LWPB) {
#else
    // eight counters (one per XCD's workgroups, each over an eighth of the list): a single one would see a grab per visit
    // from every wavefront of the chip -- 100 k returning atomics on one address per pass, ~10 ns each
    const int xcd = blockIdx.x & 7;
    const int32_t per = (nt + 7) >> 3;
    int32_t *wk = &work3[(pass % 3) * 8 + xcd];
    for (;;) {
        int32_t k = 0;
        if (lane == 0) k = atomicAdd(wk, 1);
        k = __builtin_amdgcn_readfirstlane(__shfl(k, 0));
        (void)per;
        k = k * 8 + xcd;                   // interleaved: the list is in wake order, contiguous eighths would be regions of unequal work
        if (k >= nt) break;
#endif
        const int tid = __builtin_amdgcn_readfirstlane(list_in[k]);
        const uint32_t blk = Y.tile_sym ? __builtin_amdgcn_readfirstlane(Y.tile_sym[tid]) : SYM_NONE;
        if (blk != SYM_NONE) sym_light_visit(A, Y, SL[wave], pass, tiles_x, tid, lane, blk, fin, N, s_pend[wave], npend);     // (a tile that went symbolic, K5f)
        else sweep_one_tile<true>(A, L[wave], pass, tiles_x, tid, lane, tile_done, fin, N, s_pend[wave], npend, s_nbr16);
        if (npend > TILE_PEND - 10) flush();            // (a visit adds at most ten)
    }
    if (npend) flush();
    if (lane == 0 && fin) atomicAdd(n_final, fin);
}

// ------------------------------------------------------------------------------- K5d
// Dense level prologue (PYDEM_SWEEP_DENSE=k, measured alternative -- see DESIGN.md section 4 "Round 4"): k full-grid
// streaming kernels ahead of the tile passes.  Level p finishes every open cell whose sources all finished in a level
// < p: one lane per cell, no lists, no atomics besides the count, the same gather order as the tile visits (NW ... SE,
// then the pit in-edges by source), so the values are bit-identical to theirs.  A stamp of THIS level reads as "open"
// (levels < p count), so concurrent stamps never let a cell read a value written in the same launch.
constexpr int DENSE_BAND = 64;
__global__ __launch_bounds__(256) void k_sweep_dense_level(SweepArgs A, uint32_t p, int32_t *n_final)
{
    const int n = A.n, m = A.m;
    int32_t fin = 0;
    // a workgroup walks a band of DENSE_BAND consecutive rows of its 256 columns: the neighbour rows it reads were touched by
    // itself a row earlier (L1 / the XCD's L2), and the count of finished cells costs one atomic per band and column block
    // (one per row and block kept the counter's L2 channel busy for 12 ms per level)
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i_end = min(n, (int)(blockIdx.y + 1) * DENSE_BAND);
    if (j < m)
    for (int i = blockIdx.y * DENSE_BAND; i < i_end; i++) {
        const int32_t c = i * m + j;
        const uint32_t w = A.cinfo[c];
        const uint32_t lv = ci_level(w);
        if (lv >= 1 && lv != CI_LEVEL_INF) continue;                     // finished in an earlier level
        const uint32_t cw = w & CI_STATIC_MASK;
        bool ready = true;
#pragma unroll
        for (int d = 0; d < 8; d++)
            if (cw & (1u << d)) {
                const uint32_t lu = ci_level(A.cinfo[c + NB_DI[d] * m + NB_DJ[d]]);
                ready = ready && lu >= 1 && lu < p;
            }
        if (!ready) continue;
        int2 po = make_int2(0, 0);
        if (cw & CI_PIT_IN) {
            po = pit_stash(A, c);
            for (int32_t e = po.x; e < A.n_pit && A.pin_dst[e] == c; e++) {
                const uint32_t lu = ci_level(A.cinfo[A.pin_src[e]]);
                ready = ready && lu >= 1 && lu < p;
            }
            if (!ready) continue;
        }
        double a = A.a0[i];
        bool td = (i == 0 || i == n - 1 || j == 0 || j == m - 1) && A.todo_work[c] != 0;
#pragma unroll
        for (int d = 0; d < 8; d++)
            if (cw & (1u << d)) { const double x = in_edge(A, c, m, d); a += fabs(x); td = td || (x < 0); }
        if (cw & CI_PIT_IN)
            for (int32_t e = po.x; e < A.n_pit && A.pin_dst[e] == c; e++) {
                const int32_t sc = A.pin_src[e];
                a += A.area[sc] * A.pin_w[e];
                td = td || (A.todo_work[sc] != 0);
            }
        double2 o = make_double2(0.0, 0.0);
        if (cw & (CI_OUT1 | CI_OUT2)) {
            const double pv = A.prop[c];
            if (cw & CI_OUT1) o.x = a * pv;
            if (cw & CI_OUT2) o.y = a * (1 - pv);
            if (td) { o.x = -o.x; o.y = -o.y; }
        }
        A.area[c] = a;
        A.contrib[c] = o;
        A.cinfo[c] = ci_with_level(cw, p);
        if (td) A.todo_work[c] = 1;
        fin++;
    }
    for (int off = 32; off > 0; off >>= 1) fin += __shfl_down(fin, off);
    __shared__ int32_t s_fin;
    if (threadIdx.x == 0) s_fin = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && fin) atomicAdd(&s_fin, fin);
    __syncthreads();
    if (threadIdx.x == 0 && s_fin) atomicAdd(n_final, s_fin);
}

// ------------------------------------------------------------------------------- K5e
// Resident visits for the river passes.  After a dozen passes the listed tiles have a few dozen open cells each, and a
// pass lasts as long as its longest dependency chain: rounds per visit x time per round.  The generic round goes
// through global memory for every cell (proportion, in-edge shares, results, a fence that waits for the stores); here
// the open cells of the tile get SLOTS in LDS -- (K, p) = constant part of the area (cell area + shares of the upstream
// cells that are final already) and proportion, fetched for all open cells at once before the rounds --, the rounds
// only touch LDS, and the results leave the CU once after the last round.  A tile with more open cells than slots
// takes the generic visit (sweep_one_tile).  The sum of a cell's in-edges is taken over the final ones first and the
// ones finished in this visit second, both ascending (the generic visit: ascending over all of them).
constexpr int RCAP = 256;
constexpr uint32_t RS_TAINT = 1u << 19, RS_FIN = 1u << 20;      // slot word: bits 0-10 cell, 11-18 in-edges from cells open at setup
constexpr int RS_OPEN_SHIFT = 11, RS_FINAL_SHIFT = 21;          //            bits 21-28 in-edges from cells final at setup
constexpr uint32_t RS_CELL = (1u << RS_OPEN_SHIFT) - 1u, RS_CELL_OPEN = (1u << 19) - 1u;

struct TileR {
    TileW W;                    // staging, final bitmap, count-downs as in the generic visit; W.list = ready SLOTS (each enters once)
    uint16_t map[TH * TT];      // cell -> slot (only read for cells that were open at setup)
    double Kd[RCAP];            // constant part while the cell is open, its area once it is finished
    double Pd[RCAP];            // proportion
    uint32_t sm[RCAP];          // slot word
};

// tile-local id of the neighbour in-edge d (0..7 = NW N NE W E SW S SE) comes from
__device__ __forceinline__ int nb_local(int cell, int d)
{
    const int q = d + (d >> 2);
    const int di = (q * 11) >> 5;
    return cell + (di - 1) * TT + (q - 3 * di - 1);
}

__device__ __forceinline__ void sweep_tile_resident(const SweepArgs &A, TileR &R, uint32_t pass, int tiles_x, int tid, int lane,
                                                    uint8_t *__restrict__ tile_done, int32_t &n_final, const TileNext &N,
                                                    int32_t *pend, int &npend)
{
    TileW &L = R.W;
    const int by = tid / tiles_x, bx = tid - by * tiles_x;
    const int i0 = by * TH, j0 = bx * TT, n = A.n, m = A.m;
    const int half = lane >> 5, l32 = lane & 31;
    constexpr int NSET = TH * TT / 64;
    if (lane == 0) L.tail = 0;
    const double a0_row = (lane < TH && i0 + lane < n) ? A.a0[i0 + lane] : 0.0;
    tile_stage(A, tile_base(A, i0, j0), L, pass, i0, j0, lane);
    tile_wave_sync();
    // ---- setup: open-upstream counts (as in the generic visit) and a slot per open cell
    uint32_t pitmask = 0;
    int nslot = 0;
#pragma unroll 2
    for (int k = 0; k < NSET; k++) {
        const int li = 2 * k + half + 1, idx = lane + 64 * k;
        const uint32_t w = L.cs[idx];
        const bool open = !((w >> SP_STATE_SHIFT) & 3u);
        const unsigned long long bo = __ballot(open);
        if (open) {
            const unsigned long long f0 = L.fin[li - 1], f1 = L.fin[li], f2 = L.fin[li + 1];
            const uint32_t nf = ((uint32_t)(f0 >> l32) & 7u) | (((uint32_t)(f1 >> l32) & 1u) << 3) |
                                (((uint32_t)(f1 >> (l32 + 2)) & 1u) << 4) | (((uint32_t)(f2 >> l32) & 7u) << 5);
            const uint32_t im = (w >> 16) & 0xFFu;
            const uint32_t cnt = __popc(im & ~nf);
            sp_of(L, idx) = (uint16_t)cnt;
            const int s = nslot + __popcll(bo & ((1ull << lane) - 1ull));
            R.map[idx] = (uint16_t)s;
            R.sm[s] = (uint32_t)idx | ((im & ~nf) << RS_OPEN_SHIFT) | ((im & nf) << RS_FINAL_SHIFT);
            if (w & (CI_PIT_IN << 16)) pitmask |= 1u << k;
            else if (cnt == 0) L.list[atomicAdd(&L.tail, 1)] = (uint16_t)s;
        }
        nslot += __popcll(bo);
    }
    // pit in-edges of the lane's drains: sources open in this tile are released on chip, sources another tile has not
    // finished block the drain for this pass
    if (pitmask) {
#pragma unroll 1
        for (int k = 0; k < NSET; k++) {
            if (!(pitmask & (1u << k))) continue;
            const int cell = lane + 64 * k;
            const int32_t c = (i0 + (cell >> 5)) * m + j0 + (cell & 31);
            uint32_t cnt = sp_of(L, cell);
            for (int32_t e = pit_stash(A, c).x; e < A.n_pit && A.pin_dst[e] == c; e++) {
                const int32_t sc = A.pin_src[e];
                const int si = sc / m - i0, sj = sc % m - j0;
                if (si >= 0 && si < TH && sj >= 0 && sj < TT) cnt += sp_state(L, si * TT + sj) ? 0u : 1u;
                else { const uint32_t lv = ci_level(A.cinfo[sc]); if (!(lv >= 1 && lv < pass)) cnt |= SP_BLOCKED; }
            }
            sp_of(L, cell) = (uint16_t)cnt;
            if (cnt == 0) L.list[atomicAdd(&L.tail, 1)] = R.map[cell];
        }
    }
    tile_wave_sync();
    if (lane < TH) L.a0[lane] = a0_row;                  // (the final bitmap is dead now)
    tile_wave_sync();
    // ---- the constant part of every open cell: all its loads in flight together
    for (int s = lane; s < nslot; s += 64) {
        uint32_t smv = R.sm[s];
        const int cell = smv & RS_CELL;
        const int gi = i0 + (cell >> 5), gj = j0 + (cell & 31);
        const int32_t c = gi * m + gj;
        const uint32_t cw = L.cs[cell] >> 16;
        double pv = 0.0;
        if (cw & (CI_OUT1 | CI_OUT2)) pv = A.prop[c];
        bool td = (gi == 0 || gi == n - 1 || gj == 0 || gj == m - 1) && A.todo_work[c] != 0;
        uint32_t mm = (smv >> RS_FINAL_SHIFT) & 0xFFu;
        double xs[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            xs[q] = 0.0;
            if (mm) { const int d = __ffs(mm) - 1; mm &= mm - 1u; xs[q] = in_edge(A, c, m, d); }
        }
        double K = L.a0[cell >> 5];
#pragma unroll
        for (int q = 0; q < 4; q++) { K += fabs(xs[q]); td = td || (xs[q] < 0); }
        while (mm) { const int d = __ffs(mm) - 1; mm &= mm - 1u; const double x = in_edge(A, c, m, d); K += fabs(x); td = td || (x < 0); }
        R.Kd[s] = K; R.Pd[s] = pv;
        R.sm[s] = (smv & RS_CELL_OPEN) | (td ? RS_TAINT : 0u);
    }
    tile_lds_sync();
    // ---- rounds on LDS
    uint32_t wake = 0;
    int head = 0;
    for (;;) {
        const int tail = L.tail;
        if (head >= tail) break;
        const int idx = head + lane;
        if (idx < tail) {
            // (a lane that went on at once with a target its cell released -- no list, no barrier -- was measured: slower,
            // the other branch of a braided river waits in the list meanwhile)
            const int s = L.list[idx];
            const uint32_t smv = R.sm[s];
            const int cell = smv & RS_CELL, li = cell >> 5, lj = cell & 31;
            const uint32_t cw = L.cs[cell] >> 16;
            double a = R.Kd[s];
            bool td = (smv & RS_TAINT) != 0u;
            uint32_t mm = (smv >> RS_OPEN_SHIFT) & 0xFFu;       // sources of this tile that were open at setup: finished in this visit
            while (mm) {
                const int d = __ffs(mm) - 1; mm &= mm - 1u;
                const int ss = R.map[nb_local(cell, d)];
                const double as = R.Kd[ss], ps = R.Pd[ss];
                a += ((0x5A >> d) & 1) ? as * ps : as * (1 - ps);
                td = td || (R.sm[ss] & RS_TAINT);
            }
            const int32_t c = (i0 + li) * m + j0 + lj;
            int2 po = make_int2(0, 0);
            if (cw & (CI_PIT_IN | CI_PIT_OUT)) po = pit_stash(A, c);
            if (cw & CI_PIT_IN)
                for (int32_t e = po.x; e < A.n_pit && A.pin_dst[e] == c; e++) {
                    const int32_t sc = A.pin_src[e];
                    const int si = sc / m - i0, sj = sc % m - j0;
                    bool here = false;
                    if (si >= 0 && si < TH && sj >= 0 && sj < TT) here = sp_state(L, si * TT + sj) == 2u;      // finished in this visit
                    if (here) { const int ss = R.map[si * TT + sj]; a += R.Kd[ss] * A.pin_w[e]; td = td || (R.sm[ss] & RS_TAINT); }
                    else { a += A.area[sc] * A.pin_w[e]; td = td || (A.todo_work[sc] != 0); }
                }
            R.Kd[s] = a;
            R.sm[s] = (smv & RS_CELL) | (td ? RS_TAINT : 0u) | RS_FIN;
            sp_of(L, cell) = (uint16_t)(2u << SP_STATE_SHIFT);
            auto release = [&](int ti, int tj) {            // tile-local coordinates 0..TT-1 when inside
                if (ti >= 0 && ti < TH && tj >= 0 && tj < TT) {
                    const int tcell = ti * TT + tj;
                    if (sp_dec(L, tcell) == 1u) L.list[atomicAdd(&L.tail, 1)] = R.map[tcell];
                } else {
                    const int dti = ti < 0 ? -1 : (ti >= TH ? 1 : 0), dtj = tj < 0 ? -1 : (tj >= TT ? 1 : 0);
                    if (ti >= -TH && ti < 2 * TH && tj >= -TT && tj < 2 * TT) wake |= 1u << ((dti + 1) * 3 + dtj + 1);
                    else {
                        const int tt = ((i0 + ti) / TH) * tiles_x + (j0 + tj) / TT;
                        if (atomicExch(&N.flag[tt], (int32_t)pass + 1) != (int32_t)pass + 1) N.list[atomicAdd(N.count, 1)] = tt;
                    }
                }
            };
            const int sct = ci_section(cw);
            if (cw & CI_OUT1) release(li + fe1r(sct), lj + fe1c(sct));
            if (cw & CI_OUT2) release(li + fe2r(sct), lj + fe2c(sct));
            if (cw & CI_PIT_OUT)
                for (int32_t e = po.y; e < A.n_pit && A.pit_src[e] == c; e++) {
                    const int32_t dc = A.pit_dst[e];
                    release(dc / m - i0, dc % m - j0);
                }
        }
        head = tail < head + 64 ? tail : head + 64;
        tile_lds_sync();
    }
    // ---- results of the finished cells
    int nfin = 0;
    for (int s = lane; s < nslot; s += 64) {
        const uint32_t smv = R.sm[s];
        if (!(smv & RS_FIN)) continue;
        const int cell = smv & RS_CELL;
        const int32_t c = (i0 + (cell >> 5)) * m + j0 + (cell & 31);
        const uint32_t cw = L.cs[cell] >> 16;
        const double a = R.Kd[s], pv = R.Pd[s];
        double2 o = make_double2(0.0, 0.0);
        if (cw & CI_OUT1) o.x = a * pv;
        if (cw & CI_OUT2) o.y = a * (1 - pv);
        if (smv & RS_TAINT) { o.x = -o.x; o.y = -o.y; A.todo_work[c] = 1; }
        A.area[c] = a;
        A.contrib[c] = o;
        A.cinfo[c] = ci_with_level(cw, pass);
        nfin++;
    }
    for (int off = 32; off > 0; off >>= 1) { nfin += __shfl_down(nfin, off); wake |= __shfl_xor(wake, off); }
    bool win = false;
    int tt = 0;
    if (lane < 9 && ((wake >> lane) & 1u)) { tt = tid + (lane / 3 - 1) * tiles_x + (lane % 3 - 1); win = true; }
    if (win) win = atomicExch(&N.flag[tt], (int32_t)pass + 1) != (int32_t)pass + 1;
    const unsigned long long bw = __ballot(win);
    if (win) pend[npend + __popcll(bw & ((1ull << lane) - 1ull))] = tt;
    npend += __popcll(bw);
    if (lane == 0) {
        n_final += nfin;
        A.tile_open[tid] = nslot - nfin;
        if (nfin == nslot) tile_done[tid] = 1;
    }
    tile_wave_sync();
}

// listed passes with few tiles: one wavefront (= one workgroup) per tile, resident visit when the tile's open cells fit
__global__ __launch_bounds__(64) void k_sweep_tiles_resident(SweepArgs A, uint32_t pass, int tiles_x, const int32_t *__restrict__ list_in,
                                                             const int32_t *n_in, uint8_t *__restrict__ tile_done, int32_t *n_final,
                                                             TileNext N, int32_t *clear_count, SymArgs Y)
{
    __shared__ TileR R;
    __shared__ SymLight SL;
    __shared__ int32_t s_pend[TILE_PEND];
    __shared__ int32_t s_nbr16[8];
    fill_nbr16(s_nbr16, A.m);
    __syncthreads();
    const int lane = threadIdx.x;
    const int32_t nt = *n_in;
    if (blockIdx.x == 0 && lane == 0) *clear_count = 0;
    int32_t fin = 0;
    int npend = 0;
    auto flush = [&]() {
        tile_wave_sync();
        int32_t base = 0;
        if (lane == 0) base = atomicAdd(N.count, npend);
        base = __shfl(base, 0);
        if (lane < npend) N.list[base + lane] = s_pend[lane];
        npend = 0;
        tile_wave_sync();
    };
    for (int32_t k = blockIdx.x; k < nt; k += gridDim.x) {
        const int tid = __builtin_amdgcn_readfirstlane(list_in[k]);
        const uint32_t blk = Y.tile_sym ? __builtin_amdgcn_readfirstlane(Y.tile_sym[tid]) : SYM_NONE;
        if (blk != SYM_NONE) sym_light_visit(A, Y, SL, pass, tiles_x, tid, lane, blk, fin, N, s_pend, npend);
        else if (__builtin_amdgcn_readfirstlane(A.tile_open[tid]) <= RCAP)
            sweep_tile_resident(A, R, pass, tiles_x, tid, lane, tile_done, fin, N, s_pend, npend);
        else
            sweep_one_tile<true>(A, R.W, pass, tiles_x, tid, lane, tile_done, fin, N, s_pend, npend, s_nbr16);
        if (npend > TILE_PEND - 10) flush();
    }
    if (npend) flush();
    if (lane == 0 && fin) atomicAdd(n_final, fin);
}

// ------------------------------------------------------------------------------- K5a
// The first pass over a tile is special: nothing is final yet, so a cell can only be finished if its whole upstream
// closure lies inside the tile (and involves no pit edge).  Most cells are like that (~2/3 of a fractal tile), and for
// them nothing but the area has to leave the CU before the visit ends: one WORKGROUP of four wavefronts owns the tile,
// every thread OWNS 4 cells (row 8k + t / 32, column t % 32), and per cell LDS holds one word (graph bits + count-down of the open upstream
// cells) and one 16-byte slot that carries the cell's proportion until the cell is finished and its two contributions
// afterwards (20 KB per tile: eight tiles per CU).  A round: every thread reads the words of its own open cells,
// finishes those whose count reached zero -- in-edge gather from LDS in the same ascending order
// as process_cell(), arithmetic identical -- and counts its in-tile targets down with non-returning LDS atomics (the
// owner sees the zero in the next round).  At the end the rows of the tile are written once, coalesced: contribution
// 16 B + graph word 4 B per finished cell; the generic pass stored them in dependency order and paid ~47 B of write
// traffic per finished cell for 28 (profiles/README.md).  Cells with pit edges, cells fed from another tile and
// everything downstream of them are left to pass 2.
constexpr uint32_t FC_COUNT = 0xFu, FC_BLOCKED = 1u << 8, FC_DONE = 1u << 9, FC_TODO = 1u << 10;   // low half of the cell word

struct TileFirst {
    double2 slot[TT * TT];      // .x = proportion while the cell is open; (share 1, share 2) once it is finished
    uint32_t cs[TT * TT];       // high half: static graph bits, low half: FC_*
    double a0[TT];
};

// workgroup barrier that orders LDS traffic only: the global stores of a round (areas, fire-and-forget) are not waited for
__device__ __forceinline__ void lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__global__ __launch_bounds__(256) void k_sweep_first(SweepArgs A, int tiles_x, int tiles_total, uint8_t *__restrict__ tile_done,
                                                     int32_t *n_final)
{
    __shared__ TileFirst L;
    __shared__ int32_t s_fin[2], s_any[2];
    const int t = threadIdx.x;
    const int per = gridDim.x >> 3;                 // workgroup b runs on XCD b % 8: one contiguous band of tiles per XCD
    const int tid = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (tid >= tiles_total) return;
    const int by = tid / tiles_x, bx = tid - by * tiles_x;
    const int i0 = by * TT, j0 = bx * TT, n = A.n, m = A.m;
    const int r0 = t >> 5, l32 = t & 31;            // the thread's cell k: row 8k + r0, column l32 (idx = t + 256 k)
    const int gj = j0 + l32;
    constexpr int NSET = TT * TT / 256;
    uint32_t ingrid = 0;                             // bit k: the thread's cell k exists
    if (t < 2) { s_fin[t] = 0; s_any[t] = 0; }
    {
        // ---- stage: graph words and proportions of the thread's cells, all loads in flight at once
        uint32_t cw[NSET];
        double pv[NSET];
#pragma unroll
        for (int k = 0; k < NSET; k++) {
            const int gi = i0 + 8 * k + r0;
            const bool ok = gi < n && gj < m;
            const int64_t c = (int64_t)gi * m + gj;
            cw[k] = ok ? A.cinfo[c] : 0xFFFFFFFFu;
            pv[k] = ok ? A.prop[c] : 0.0;
        }
        if (t < TT) L.a0[t] = (i0 + t < n) ? A.a0[i0 + t] : 0.0;
        const uint32_t out_col = (l32 == 0 ? 0x29u : 0u) | (l32 == TT - 1 ? 0x94u : 0u);      // in-edges that would come from another tile
        const bool edge_col = gj == 0 || gj == m - 1;
#pragma unroll
        for (int k = 0; k < NSET; k++) {
            const int r = 8 * k + r0, gi = i0 + r;
            const uint32_t w = cw[k];
            uint32_t word = FC_BLOCKED;
            if (w != 0xFFFFFFFFu) {
                ingrid |= 1u << k;
                const uint32_t outside = out_col | (r == 0 ? 0x07u : 0u) | (r == TT - 1 ? 0xE0u : 0u);
                const bool blocked = ((w & 0xFFu & outside) != 0u) || (w & (CI_PIT_IN | CI_PIT_OUT));
                word = ((w & CI_STATIC_MASK) << 16) | __popc(w & 0xFFu) | (blocked ? FC_BLOCKED : 0u);
                if (!blocked && (edge_col || gi == 0 || gi == n - 1) && A.todo_work[(int64_t)gi * m + gj] != 0) word |= FC_TODO;   // inlet cells of the tile's edge
            }
            L.cs[t + 256 * k] = word;
            L.slot[t + 256 * k].x = pv[k];
        }
    }
    lds_sync();
    int ph = 0;
    // ---- rounds.  A thread that counts a target down to zero goes on with that target itself ("chain": a river is
    // walked by one thread within one round instead of one cell per round); whatever else becomes ready waits for
    // its owner's next look
    for (;;) {
        uint32_t ready = 0;
#pragma unroll
        for (int k = 0; k < NSET; k++)
            if ((L.cs[t + 256 * k] & (FC_COUNT | FC_BLOCKED | FC_DONE)) == 0u) ready |= 1u << k;
        if (ready) s_any[ph] = 1;
        lds_sync();                                    // (also: every thread has read its words before anybody counts down)
        const bool go = s_any[ph] != 0;
        if (t == 0) s_any[ph ^ 1] = 0;
        ph ^= 1;
        if (!go) break;
        while (ready) {
            const int k = __ffs(ready) - 1; ready &= ready - 1u;
            int cur = t + 256 * k;
            do {
                const int r = cur >> 5, cl = cur & 31;
                const uint32_t word = L.cs[cur], w = word >> 16;
                const double pv = L.slot[cur].x;
                double a = L.a0[r];
                bool td = (word & FC_TODO) != 0u;
                uint32_t mm = w & 0xFFu;
                while (mm) {                                        // ascending neighbour order, like the reference's pull
                    const int d = __ffs(mm) - 1; mm &= mm - 1u;
                    const int q = d + (d >> 2), di = (q * 11) >> 5;            // q = 0..8 with the centre skipped, di = q / 3
                    const double2 u = L.slot[cur + (di - 1) * TT + (q - 3 * di - 1)];
                    const double x = ((0x5A >> d) & 1) ? u.x : u.y;             // cardinal neighbours hand over their first share
                    a += fabs(x); td = td || (x < 0);
                }
                double2 o = make_double2(0.0, 0.0);
                if (w & CI_OUT1) o.x = a * pv;
                if (w & CI_OUT2) o.y = a * (1 - pv);
                if (td) { o.x = -o.x; o.y = -o.y; }
                L.slot[cur] = o;
                L.cs[cur] = word | FC_DONE | (td ? FC_TODO : 0u);
                A.area[(int64_t)(i0 + r) * m + j0 + cl] = a;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");      // the shares are in place before a count can reach zero
                const int sct = ci_section(w);
                int next = -1;
                if (w & CI_OUT1) {
                    const int tr = r + fe1r(sct), tc = cl + fe1c(sct);
                    if (tr >= 0 && tr < TT && tc >= 0 && tc < TT &&
                        (atomicSub(&L.cs[tr * TT + tc], 1u) & (FC_COUNT | FC_BLOCKED)) == 1u) next = tr * TT + tc;
                }
                if (w & CI_OUT2) {
                    const int tr = r + fe2r(sct), tc = cl + fe2c(sct);
                    if (tr >= 0 && tr < TT && tc >= 0 && tc < TT &&
                        (atomicSub(&L.cs[tr * TT + tc], 1u) & (FC_COUNT | FC_BLOCKED)) == 1u && next < 0) next = tr * TT + tc;
                }
                cur = next;
            } while (cur >= 0);
        }
        lds_sync();
    }
    // ---- write the finished cells, row by row
    uint32_t done = 0;
#pragma unroll
    for (int k = 0; k < NSET; k++) {
        const uint32_t word = L.cs[t + 256 * k];
        if (!(word & FC_DONE)) continue;
        done |= 1u << k;
        const int64_t c = (int64_t)(i0 + 8 * k + r0) * m + gj;
        A.contrib[c] = L.slot[t + 256 * k];
        A.cinfo[c] = ci_with_level(word >> 16, 1u);
        if (word & FC_TODO) A.todo_work[c] = 1;
    }
    const int32_t finalized = __popc(done), n_cells = __popc(ingrid);
    if (finalized) atomicAdd(&s_fin[0], finalized);
    if (n_cells) atomicAdd(&s_fin[1], n_cells);
    __syncthreads();
    if (t == 0) {
        if (s_fin[0]) atomicAdd(n_final, s_fin[0]);
        if (s_fin[0] == s_fin[1]) tile_done[tid] = 1;
    }
}

#ifdef PYDEM_SWEEP_QUEUE
// switch from queue rounds to listed tile passes: the tiles that hold the current frontier
__global__ void k_tiles_of_frontier(const QE *__restrict__ q, const int32_t *nq, int m, int tiles_x, int32_t stamp, TileNext N)
{
    const int32_t n = *nq;
    for (int32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const int32_t c = q[k].c;
        const int tt = (c / m / TH) * tiles_x + (c % m) / TT;
        if (atomicExch(&N.flag[tt], stamp) != stamp) N.list[atomicAdd(N.count, 1)] = tt;
    }
}

// after the tile passes: cells that are not final but whose upstream cells all are form the first
// queue frontier (level `r`)
__global__ __launch_bounds__(256) void k_sweep_rebuild_frontier(SweepArgs A, uint32_t r, QE *__restrict__ qn, int32_t *cn)
{
    __shared__ Stage S;
    if (threadIdx.x == 0) S.cnt = 0;
    __syncthreads();
    const int64_t NN = (int64_t)A.n * A.m;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < NN; base += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = base + threadIdx.x;
        bool push = false;
        uint32_t cw = 0;
        if (c < NN) {
            cw = A.cinfo[c];
            const uint32_t lv = ci_level(cw);
            if (lv == 0 || lv == CI_LEVEL_INF) {
                bool ready = true;
#pragma unroll
                for (int d = 0; d < 8; d++)
                    if (cw & (1u << d)) {
                        const uint32_t lu = ci_level(A.cinfo[c + NB_DI[d] * A.m + NB_DJ[d]]);
                        ready = ready && (lu >= 1 && lu < r);
                    }
                if (ready && (cw & CI_PIT_IN))
                    for (int32_t e = pit_stash(A, (int32_t)c).x; e < A.n_pit && A.pin_dst[e] == (int32_t)c; e++) {
                        const uint32_t lu = ci_level(A.cinfo[A.pin_src[e]]);
                        ready = ready && (lu >= 1 && lu < r);
                    }
                if (ready) { A.cinfo[c] = ci_with_level(cw, r); push = true; }
            }
        }
        stage_push(S, push, (int32_t)c, cw);
        stage_flush(A, S, qn, cn, false);
    }
    stage_flush(A, S, qn, cn, true);
}

#endif  // PYDEM_SWEEP_QUEUE

__global__ void k_row_area(const double *__restrict__ dX2, const double *__restrict__ dY2, int n, double *a0)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a0[i] = dX2[i] * dY2[i];                                          // :885
}

// ------------------------------------------------------------------------------- K5c
// Circular drainage (rare: e.g. the overlap-1 patch of a tile's edge aspects can close a two-cell loop).  When the tile
// passes stall, the cells on and below the loop are unfinished; the reference then re-seeds its push sweep from the
// unfinished cells within 1 % of the highest unfinished elevation (dem_processing.py:951-964) and its Cython loop
// (cyutils.pyx:119-187) keeps going round by round -- pushing IN PLACE in ascending cell order, queueing a cell again
// whenever all its sources are done, skipping pushes into finished cells on the tile edge.  Those rules are order
// dependent, so the handful of unfinished cells is replayed here exactly like that by ONE thread (the oracle's
// oracle_uca_chunk / oracle_drain_area are the line-by-line model); everything else of the tile is already final.
struct ReseedCell { int32_t c; int32_t pin_first; int32_t pout_first; int32_t pad; };

// not finished by any pass: sources start with level 0, every other cell with "not yet known"
__device__ __forceinline__ bool ci_unfinished(uint32_t w) { const uint32_t lv = ci_level(w); return lv == 0 || lv == CI_LEVEL_INF; }

// area / taint an unfinished cell has received so far: the shares of its FINISHED upstream cells (graph word level >= 1)
__device__ void gather_finished(const SweepArgs &A, int32_t c, uint32_t cw, int32_t pin_first, double &a, bool &td)
{
    const int m = A.m, gi = c / m, gj = c - gi * m;
    a = A.a0[gi];
    td = (gi == 0 || gi == A.n - 1 || gj == 0 || gj == m - 1) && A.todo_work[c] != 0;
    for (int d = 0; d < 8; d++)
        if (cw & (1u << d)) {
            const int32_t u = c + NB_DI[d] * m + NB_DJ[d];
            if (ci_unfinished(A.cinfo[u])) continue;
            const bool cardinal = (NB_DI[d] == 0) || (NB_DJ[d] == 0);
            const double x = cardinal ? A.contrib[u].x : A.contrib[u].y;
            a += fabs(x); td = td || (x < 0);
        }
    if (cw & CI_PIT_IN)
        for (int32_t e = pin_first; e < A.n_pit && A.pin_dst[e] == c; e++) {
            const int32_t sc = A.pin_src[e];
            if (ci_unfinished(A.cinfo[sc])) continue;
            a += A.area[sc] * A.pin_w[e];
            td = td || (A.todo_work[sc] != 0);
        }
}

// the unfinished cells, in any order (the host sorts the few of them); their pit-list offsets are saved because the area
// slot that holds them is about to carry the area
__global__ __launch_bounds__(256) void k_reseed_collect(SweepArgs A, ReseedCell *__restrict__ list, int32_t *count, int32_t cap,
                                                        const double *__restrict__ elev, int32_t *nan_flag)
{
    const int64_t NN = (int64_t)A.n * A.m;
    for (int64_t c64 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c64 < NN; c64 += (int64_t)gridDim.x * blockDim.x) {
        const int32_t c = (int32_t)c64;
        if (elev[c] != elev[c]) *nan_flag = 1;              // numpy's max over the tile propagates NaN (:963): then nothing is re-seeded
        const uint32_t w = A.cinfo[c];
        if (!ci_unfinished(w)) continue;
        int2 po = make_int2(0, 0);
        if (w & (CI_PIT_IN | CI_PIT_OUT)) po = pit_stash(A, c);
        const int32_t slot = atomicAdd(count, 1);
        if (slot < cap) { list[slot].c = c; list[slot].pin_first = po.x; list[slot].pout_first = po.y; list[slot].pad = 0; }
    }
}

__device__ __forceinline__ int reseed_find(const ReseedCell *U, int32_t nU, int32_t c)
{
    int lo = 0, hi = nU - 1;
    while (lo <= hi) { const int mid = (lo + hi) >> 1; if (U[mid].c == c) return mid; if (U[mid].c < c) lo = mid + 1; else hi = mid - 1; }
    return -1;
}

// ONE thread.  st[k]: bit 0 done, bit 1 in the current frontier, bit 2 in the previous frontier; tdf[k]: taint
__global__ void k_reseed_replay(SweepArgs A, const ReseedCell *__restrict__ U, int32_t nU, const double *__restrict__ elev,
                                const double *__restrict__ pit_w, uint8_t *__restrict__ st, uint8_t *__restrict__ tdf,
                                int tile_has_nan, int maxcount, uint32_t pass, int32_t *n_final, int32_t *n_done_out)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const int n = A.n, m = A.m;
    auto on_edge = [&](int32_t c) { const int i = c / m, j = c - i * m; return i == 0 || i == n - 1 || j == 0 || j == m - 1; };
    auto is_done = [&](int32_t c) -> bool {            // finished before the stall, or in the replay
        if (!ci_unfinished(A.cinfo[c])) return true;
        const int k = reseed_find(U, nU, c);
        return k >= 0 && (st[k] & 1u);
    };
    // what the unfinished cells have received so far
    for (int k = 0; k < nU; k++) {
        double a; bool td;
        gather_finished(A, U[k].c, A.cinfo[U[k].c] & CI_STATIC_MASK, U[k].pin_first, a, td);
        A.area[U[k].c] = a; tdf[k] = td; st[k] = 0;
    }
    int32_t n_done = 0, done_prev = -1;
    for (int count = 2; n_done < nU && count < maxcount && n_done != done_prev; count++) {     // :951-952 (the first drain was the tile passes)
        done_prev = n_done;
        // ---- the new frontier: ids[((data * ~done - max_elev) / max_elev > -0.01)] (:962-964); finished cells count as 0
        double mx = 0.0;
        for (int k = 0; k < nU; k++) if (!(st[k] & 1u) && elev[U[k].c] > mx) mx = elev[U[k].c];
        if (tile_has_nan) mx = NAN;
        for (int k = 0; k < nU; k++) {
            const double v = (st[k] & 1u) ? 0.0 : elev[U[k].c];
            st[k] = (uint8_t)((st[k] & 1u) | (((v - mx) / mx > -0.01) ? 2u : 0u));
        }
        // ---- cyutils._drain_area (:119-187)
        for (int64_t guard = 0; guard < 8 * (int64_t)nU + 64; guard++) {
            for (int k = 0; k < nU; k++) if (st[k] & 2u) { if (!(st[k] & 1u)) n_done++; st[k] |= 1u; }     // :138-140
            for (int k = 0; k < nU; k++) st[k] = (uint8_t)((st[k] & 1u) | ((st[k] & 2u) ? 4u : 0u));         // swap, zero the new frontier
            for (int k = 0; k < nU; k++) {
                if (!(st[k] & 4u)) continue;
                const int32_t i = U[k].c;
                const uint32_t cw = A.cinfo[i] & CI_STATIC_MASK;
                // the column of i: targets in ascending cell order (scipy sorts the indices of a CSC column)
                int32_t tg[2]; double fc[2]; int nt = 0;
                if (cw & (CI_OUT1 | CI_OUT2)) {
                    const int sct = ci_section(cw);
                    const double pv = A.prop[i];
                    if (cw & CI_OUT1) { tg[nt] = i + fe1r(sct) * m + fe1c(sct); fc[nt] = pv; nt++; }
                    if (cw & CI_OUT2) { tg[nt] = i + fe2r(sct) * m + fe2c(sct); fc[nt] = 1 - pv; nt++; }
                    if (nt == 2 && tg[1] < tg[0]) { const int32_t tt = tg[0]; tg[0] = tg[1]; tg[1] = tt; const double ff = fc[0]; fc[0] = fc[1]; fc[1] = ff; }
                }
                int32_t e = (cw & CI_PIT_OUT) ? U[k].pout_first : 0;
                for (int q = 0;; q++) {
                    int32_t row; double factor;
                    if (cw & CI_PIT_OUT) { if (!(e < A.n_pit && A.pit_src[e] == i)) break; row = A.pit_dst[e]; factor = pit_w[e]; e++; }
                    else { if (q >= nt) break; row = tg[q]; factor = fc[q]; }
                    const int kr = reseed_find(U, nU, row);
                    if (kr < 0) continue;                                   // (cannot happen: everything below an unfinished cell is unfinished)
                    if ((st[kr] & 1u) && on_edge(row)) continue;            // :159-161
                    A.area[row] += A.area[i] * factor;                      // :163
                    if (tdf[k]) tdf[kr] = 1;                                // edge_todo[row] += edge_todo[i] * factor (positive factors)
                    bool wait = false;                                      // :173-179
                    const uint32_t cwr = A.cinfo[row] & CI_STATIC_MASK;
                    for (int d = 0; d < 8 && !wait; d++)
                        if ((cwr & (1u << d)) && !is_done(row + NB_DI[d] * m + NB_DJ[d])) wait = true;
                    if (!wait && (cwr & CI_PIT_IN))
                        for (int32_t e2 = U[kr].pin_first; e2 < A.n_pit && A.pin_dst[e2] == row; e2++)
                            if (!is_done(A.pin_src[e2])) { wait = true; break; }
                    if (!wait) st[kr] |= 2u;
                }
            }
            bool changed = false;                                           // :187
            for (int k = 0; k < nU; k++) if (((st[k] >> 1) & 1u) != ((st[k] >> 2) & 1u)) changed = true;
            if (!changed) break;
        }
        for (int k = 0; k < nU; k++) st[k] &= 1u;                           // ids[:] = False (:962)
    }
    // finished cells join the others (level stamp, taint); the rest keep what they have received, like the reference
    for (int k = 0; k < nU; k++) {
        const int32_t c = U[k].c;
        if (tdf[k]) A.todo_work[c] = 1;
        if (st[k] & 1u) A.cinfo[c] = ci_with_level(A.cinfo[c], pass);
    }
    atomicAdd(n_final, n_done);
    *n_done_out = n_done;
}

// Host replay (more unfinished cells than the one-thread replay above is meant for): what the host needs of an unfinished
// cell -- its static graph word, elevation, proportion, pit-list offsets and what it has received from its finished
// upstream cells -- gathered in parallel; the results come back through k_reseed_scatter.
struct ReseedHost { int32_t c; uint32_t cw; int32_t pin_first, pout_first; double area, elev, prop; int32_t td, pad; };
__global__ __launch_bounds__(256) void k_reseed_gather(SweepArgs A, const ReseedCell *__restrict__ U, int32_t nU, const double *__restrict__ elev,
                                                       ReseedHost *__restrict__ out)
{
    for (int32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < nU; k += gridDim.x * blockDim.x) {
        ReseedHost h;
        h.c = U[k].c; h.cw = A.cinfo[h.c] & CI_STATIC_MASK; h.pin_first = U[k].pin_first; h.pout_first = U[k].pout_first;
        double a; bool td;
        gather_finished(A, h.c, h.cw, h.pin_first, a, td);
        h.area = a; h.td = td; h.pad = 0; h.elev = elev[h.c]; h.prop = (h.cw & (CI_OUT1 | CI_OUT2)) ? A.prop[h.c] : 0.0;
        out[k] = h;
    }
}
struct ReseedBack { int32_t c; int32_t flags; double area; };      // flags: bit 0 finished by the replay, bit 1 taint
__global__ __launch_bounds__(256) void k_reseed_scatter(SweepArgs A, const ReseedBack *__restrict__ B, int32_t nU, uint32_t pass)
{
    for (int32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < nU; k += gridDim.x * blockDim.x) {
        const int32_t c = B[k].c;
        A.area[c] = B[k].area;
        if (B[k].flags & 2) A.todo_work[c] = 1;
        if (B[k].flags & 1) A.cinfo[c] = ci_with_level(A.cinfo[c], pass);
    }
}

// finalisation of _calc_uca_chunk (:966-980): NaN on flats, edge_done = ~edge_todo etc.
// (16 cells per thread: the byte masks travel as 16-byte words; uca and elev are only touched where the masks ask for
// them -- flats, cells still on `todo` -- unless the saturation limit needs every value: 3 instead of 19 bytes per cell)
__device__ __forceinline__ bool finalize_cell(double *__restrict__ uca, const double *__restrict__ elev, int64_t c, uint32_t f, uint32_t td,
                                              int apply_limit, double limit)
{
    if (f) uca[c] = NAN;                                                         // :972
    bool dn = !td;                                                               // :974
    if (td && isnan(elev[c])) dn = true;                                         // :975
    if (apply_limit && !f && uca[c] > limit) dn = true;                          // :977-980 (NaN > limit is false)
    return dn;
}

__global__ __launch_bounds__(256) void k_uca_finalize(double *__restrict__ uca, const uint8_t *__restrict__ flats,
                                                      const uint8_t *__restrict__ todo_work, const double *__restrict__ elev,
                                                      uint8_t *__restrict__ edge_done, int64_t NN, int apply_limit, double limit)
{
    const int64_t nvec = NN >> 4, stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t v = t0; v < nvec; v += stride) {
        const uint4 f4 = reinterpret_cast<const uint4 *>(flats)[v], t4 = reinterpret_cast<const uint4 *>(todo_work)[v];
        const uint32_t fw[4] = {f4.x, f4.y, f4.z, f4.w}, tw[4] = {t4.x, t4.y, t4.z, t4.w};
        uint32_t dw[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t d = 0;
#pragma unroll
            for (int b = 0; b < 4; b++)
                d |= (finalize_cell(uca, elev, v * 16 + q * 4 + b, (fw[q] >> (8 * b)) & 0xFFu, (tw[q] >> (8 * b)) & 0xFFu, apply_limit, limit) ? 1u : 0u) << (8 * b);
            dw[q] = d;
        }
        reinterpret_cast<uint4 *>(edge_done)[v] = make_uint4(dw[0], dw[1], dw[2], dw[3]);
    }
    for (int64_t c = (nvec << 4) + t0; c < NN; c += stride)
        edge_done[c] = finalize_cell(uca, elev, c, flats[c], todo_work[c], apply_limit, limit);
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

// ------------------------------------------------------------------------------- K7
// Edge-resolution round for one tile: DEMProcessor.calc_uca(uca_init=..., edge_init_data=...)
// (reference pydem/dem_processing.py:720-771) and _calc_uca_chunk_update (:778-862) with the
// native floods cyutils.drain_connections (cyutils.pyx:35-72) and drain_area (:78-187).
// The reference rebuilds section/proportion/adjacency on every call (:787-793); here the graph
// built by pydem_uca is still resident and is reused.  Only cells downstream of the seeds are
// touched: stamp[c] == epoch marks membership, so nothing of size NN is cleared per round except
// the two byte masks that are outputs.
// Per-cell state of a round lives in two words that are ZERO between rounds (the cells a round touched
// are on its lists and are wiped at its end, so nothing of size NN is cleared per round):
//   flag[c]   EF_S reached from a seed (or a seed), EF_SEED, EF_T visited by the todo flood, EF_DONE swept
//   cinfo[c]  the level field (unused after the main sweep) counts the in-edges of c that come from
//             reached cells; bit 31 marks the seeds.  The reach flood increments it once per edge it
//             walks, the sweep decrements it once per edge it has pulled over: a cell is ready when
//             its count returns to zero -- textbook Kahn, but only on the few thousand cells downstream
//             of an edge, and every step of a level is ONE batch of independent loads/atomics.
constexpr uint32_t EF_S = 1u, EF_SEED = 2u, EF_T = 4u, EF_DONE = 8u;
constexpr uint32_t CI_ESEED = 1u << 31, CI_EONE = 1u << CI_LEVEL_SHIFT;
__device__ __forceinline__ uint32_t ci_ecount(uint32_t w) { return (w >> CI_LEVEL_SHIFT) & 0xFFFFu; }

struct EdgeArgs {
    SweepArgs G;             // graph
    uint32_t *flag;          // [NN]
    double *delta;           // [NN] area delta of this round (valid where EF_DONE)
    const uint8_t *flats;
    uint8_t *edge_done;      // output mask
    // perimeter tables, index p: top row (m), bottom row (m), left col rows 1..n-2, right col rows 1..n-2
    uint8_t *p_done, *p_seed;
    double *p_delta;
    int32_t *rlist, *rcount; // reached cells (seeds first): their uca is updated at the end
    int32_t *tlist, *tcount; // cells whose edge_done byte this round cleared
    const int2 *pit_off;     // per cell {first pit in-edge, first pit out-edge} (valid where the graph word says so)
};

__device__ __forceinline__ int64_t perim_index(int i, int j, int n, int m)
{
    if (i == 0) return j;
    if (i == n - 1) return (int64_t)m + j;
    if (j == 0) return 2 * (int64_t)m + (i - 1);
    if (j == m - 1) return 2 * (int64_t)m + (n - 2) + (i - 1);
    return -1;
}

// base value of a cell's delta: edge cells initialised from a finished neighbour start from
// (neighbour value - own uca) (:806-809); flats are NaN (:815); everything else 0 (:802)
__device__ __forceinline__ double edge_base(const EdgeArgs &E, int32_t c)
{
    if (E.flats[c]) return NAN;
    const int i = c / E.G.m, j = c - i * E.G.m;
    const int64_t p = perim_index(i, j, E.G.n, E.G.m);
    if (p >= 0 && E.p_done[p]) return E.p_delta[p];
    return 0.0;
}

// one list slot per calling lane, one atomic per wavefront (works in divergent code: the ballot is
// over the lanes that are executing the call)
__device__ __forceinline__ int32_t agg_slot(int32_t *count)
{
    const unsigned long long bal = __ballot(true);
    const int lane = (int)__lane_id();
    const int leader = __ffsll((long long)bal) - 1;
    int32_t base = 0;
    if (lane == leader) base = atomicAdd(count, (int32_t)__popcll(bal));
    base = __shfl(base, leader);
    return base + __popcll(bal & ((1ull << lane) - 1ull));
}

__global__ void k_edge_clear_levels(uint32_t *__restrict__ cinfo, int64_t NN)
{
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < NN; c += (int64_t)gridDim.x * blockDim.x)
        cinfo[c] &= CI_STATIC_MASK;
}

// undo the previous round's edge_done = 0 bytes (the mask is rebuilt from all-True every round, :812)
__global__ void k_edge_restore(const int32_t *__restrict__ tlist, int32_t n, uint8_t *__restrict__ edge_done)
{
    for (int32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) edge_done[tlist[k]] = 1;
}

// strips -> per-perimeter-cell state (:726-739, :798-809); seeds start the reach flood, cells
// that stay 'todo' start the todo flood
__global__ void k_edge_init(EdgeArgs E, const double *__restrict__ sdata, const uint8_t *__restrict__ sdone,
                            const uint8_t *__restrict__ stodo, int L, const double *__restrict__ uca,
                            uint8_t *__restrict__ edge_todo, QE *q_flood, int32_t *n_flood, QE *q_seed, int32_t *n_seed)
{
    const int n = E.G.n, m = E.G.m;
    const int64_t nper = 2 * (int64_t)m + 2 * (int64_t)(n - 2);
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nper) return;
    int i, j;
    if (p < m) { i = 0; j = (int)p; }
    else if (p < 2 * (int64_t)m) { i = n - 1; j = (int)(p - m); }
    else if (p < 2 * (int64_t)m + (n - 2)) { i = (int)(p - 2 * (int64_t)m) + 1; j = 0; }
    else { i = (int)(p - 2 * (int64_t)m - (n - 2)) + 1; j = m - 1; }
    const int32_t c = i * m + j;
    bool dn = false, td = false;
    double init = 0.0;
    // dict order of the reference: left, right, top, bottom
    if (j == 0) { dn |= sdone[0 * L + i] != 0; init += sdata[0 * L + i] * (double)(sdone[0 * L + i] != 0); td |= stodo[0 * L + i] != 0; }
    if (j == m - 1) { dn |= sdone[1 * L + i] != 0; init += sdata[1 * L + i] * (double)(sdone[1 * L + i] != 0); td |= stodo[1 * L + i] != 0; }
    if (i == 0) { dn |= sdone[2 * L + j] != 0; init += sdata[2 * L + j] * (double)(sdone[2 * L + j] != 0); td |= stodo[2 * L + j] != 0; }
    if (i == n - 1) { dn |= sdone[3 * L + j] != 0; init += sdata[3 * L + j] * (double)(sdone[3 * L + j] != 0); td |= stodo[3 * L + j] != 0; }
    if (!dn) init = 0.0;                                                         // :738-739
    const bool seed = dn && td;                                                  // :798
    const bool todo_out = td && !dn;                                             // :799
    E.p_done[p] = dn;
    E.p_seed[p] = seed;
    E.p_delta[p] = dn ? init - uca[c] : 0.0;                                     // :806-809
    edge_todo[c] = todo_out;                                                     // returned as edge_todo_i (:817, :862)
    if (todo_out) {
        E.edge_done[c] = 0;
        E.flag[c] = EF_T;
        E.tlist[agg_slot(E.tcount)] = c;
        QE q; q.c = c; q.cw = (E.G.cinfo[c] & CI_STATIC_MASK) | (1u << 31);     // bit 31 of a flood entry: todo flood
        q_flood[agg_slot(n_flood)] = q;
    }
    if (seed) {
        const uint32_t cw = E.G.cinfo[c] & CI_STATIC_MASK;
        E.flag[c] = EF_S | EF_SEED;
        E.G.cinfo[c] = cw | CI_ESEED;
        E.rlist[agg_slot(E.rcount)] = c;
        QE q; q.c = c; q.cw = cw;
        q_flood[agg_slot(n_flood)] = q;
        q.cw = cw | CI_ESEED;                                                    // bit 31 of a sweep entry: seed
        q_seed[agg_slot(n_seed)] = q;
    }
}

// Both floods in one breadth-first loop (entry bit 31: 0 = reach flood from the seeds, 1 = todo flood).
// Reach (:820-825): every edge walked bumps the target's count; the first visitor lists the target and
// expands it next level.  Todo (:848-853): edge_done = False downstream of the cells that stay 'todo'.
template <typename Push>
__device__ __forceinline__ void edge_flood_cell(const EdgeArgs &E, QE q, int32_t *rcount, int32_t *tcount, Push push)
{
    const SweepArgs &A = E.G;
    const int32_t u = q.c;
    const uint32_t cw = q.cw;
    const bool todo = (cw >> 31) != 0;
    const int s = ci_section(cw);
    int32_t pe = 0;
    if (cw & CI_PIT_OUT) pe = E.pit_off[u].y;
    auto visit = [&](int32_t t) {
        const uint32_t ct = A.cinfo[t];
        if (!todo) {
            const uint32_t old = atomicOr(&E.flag[t], EF_S);
            atomicAdd(&A.cinfo[t], CI_EONE);
            if (!(old & EF_S)) {
                E.rlist[agg_slot(rcount)] = t;
                push(t, ct & CI_STATIC_MASK);
            }
        } else {
            const uint32_t old = atomicOr(&E.flag[t], EF_T);
            if (!(old & EF_T)) {
                E.edge_done[t] = 0;                                              // edge_done = ~edge_todo (:856)
                E.tlist[agg_slot(tcount)] = t;
                push(t, (ct & CI_STATIC_MASK) | (1u << 31));
            }
        }
    };
    if (cw & CI_OUT1) visit(u + fe1r(s) * A.m + fe1c(s));
    if (cw & CI_OUT2) visit(u + fe2r(s) * A.m + fe2c(s));
    if (cw & CI_PIT_OUT)
        for (int32_t e = pe; e < A.n_pit && A.pit_src[e] == u; e++) visit(A.pit_dst[e]);
}

// Seeded sweep (drain_area with skip_edge=False on the flooded sub-graph, :836-842).  Seeds keep the
// edge value itself (a done cell on the tile edge never receives, cyutils.pyx:159-161); every other
// cell pulls from its reached upstream cells in the fixed neighbour order.  All loads and the
// count-down atomics on the targets depend only on the queue entry: one memory round trip per level.
template <typename Push>
__device__ __forceinline__ void edge_sweep_cell(const EdgeArgs &E, QE q, Push push)
{
    const SweepArgs &A = E.G;
    const int32_t c = q.c;
    const uint32_t cw = q.cw;
    const int m = A.m;
    const bool seed = (cw & CI_ESEED) != 0;
    // (a workgroup's memory pipeline moves about one scattered access per ns: only the neighbours in the
    // in-mask are fetched -- all in one batch, the uses come later)
    uint32_t f[8]; double dl[8], pr[8];
#pragma unroll
    for (int d = 0; d < 8; d++) {
        f[d] = 0; dl[d] = 0.0; pr[d] = 0.0;
        if (!seed && (cw & (1u << d))) {
            const int32_t u = c + NB_DI[d] * m + NB_DJ[d];
            f[d] = E.flag[u]; dl[d] = E.delta[u]; pr[d] = A.prop[u];
        }
    }
    int2 po = make_int2(0, 0);
    if (cw & (CI_PIT_IN | CI_PIT_OUT)) po = E.pit_off[c];
    const int s = ci_section(cw);
    int32_t t1 = -1, t2 = -1;
    uint32_t o1 = 0, o2 = 0;
    if (cw & CI_OUT1) { t1 = c + fe1r(s) * m + fe1c(s); o1 = atomicSub(&A.cinfo[t1], CI_EONE); }
    if (cw & CI_OUT2) { t2 = c + fe2r(s) * m + fe2c(s); o2 = atomicSub(&A.cinfo[t2], CI_EONE); }
    double acc = edge_base(E, c);
    if (!seed) {
#pragma unroll
        for (int d = 0; d < 8; d++) {
            if ((cw & (1u << d)) && (f[d] & EF_S)) {
                const bool cardinal = (NB_DI[d] == 0) || (NB_DJ[d] == 0);
                acc += dl[d] * (cardinal ? pr[d] : 1 - pr[d]);
            }
        }
        if (cw & CI_PIT_IN)
            for (int32_t e = po.x; e < A.n_pit && A.pin_dst[e] == c; e++)
                if (E.flag[A.pin_src[e]] & EF_S) acc += E.delta[A.pin_src[e]] * A.pin_w[e];
    }
    E.delta[c] = acc;
    atomicOr(&E.flag[c], EF_DONE);
    if (t1 >= 0 && ci_ecount(o1) == 1u && !(o1 & CI_ESEED)) push(t1, o1 & CI_STATIC_MASK);
    if (t2 >= 0 && ci_ecount(o2) == 1u && !(o2 & CI_ESEED)) push(t2, o2 & CI_STATIC_MASK);
    if (cw & CI_PIT_OUT)
        for (int32_t e = po.y; e < A.n_pit && A.pit_src[e] == c; e++) {
            const int32_t t = A.pit_dst[e];
            const uint32_t o = atomicSub(&A.cinfo[t], CI_EONE);
            if (ci_ecount(o) == 1u && !(o & CI_ESEED)) push(t, o & CI_STATIC_MASK);
        }
}

// one level, many workgroups (large frontiers); counters rotate over 3 slots as in the main sweep
template <int WHICH>   // 0 floods, 1 sweep
__global__ __launch_bounds__(256) void k_edge_level(EdgeArgs E, const QE *__restrict__ qc, QE *__restrict__ qn, int32_t *cnt3, int r)
{
    const int32_t nq = cnt3[r % 3];
    if (blockIdx.x == 0 && threadIdx.x == 0) cnt3[(r + 2) % 3] = 0;
    if (nq == 0) return;
    int32_t *cn = &cnt3[(r + 1) % 3];
    for (int32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < nq; k += gridDim.x * blockDim.x) {
        auto push = [&](int32_t t, uint32_t ct) { QE e; e.c = t; e.cw = ct; qn[agg_slot(cn)] = e; };
        if (WHICH == 0) edge_flood_cell(E, qc[k], E.rcount, E.tcount, push);
        else edge_sweep_cell(E, qc[k], push);
    }
}

// Small frontiers: ONE workgroup runs level after level without going back to the host (a kernel
// boundary costs a launch plus a trip across the fabric for every first access; the floods and sweeps
// downstream of an edge are hundreds of levels of a few cells).  Stops when the frontier is empty
// or outgrows SMALL_CAP and reports where it stopped.
constexpr int SMALL_CAP = 4096;

template <int WHICH>
__global__ __launch_bounds__(1024) void k_edge_small(EdgeArgs E, QE *q0, QE *q1, int32_t *cnt3, int r_start, int32_t *state)
{
    // the frontier lives in LDS (and is mirrored to the global queues, stores nobody waits for, so that a
    // frontier that outgrows the cap can be handed back); list counters are LDS copies for the same reason
    __shared__ QE s_q[2][SMALL_CAP];
    __shared__ int32_t s_next, s_rcount, s_tcount;
    int r = r_start;
    int32_t nq = cnt3[r % 3];
    if (threadIdx.x == 0) { s_rcount = *E.rcount; s_tcount = *E.tcount; }
    if (nq > 0 && nq <= SMALL_CAP) {
        const QE *qc = (r % 2) ? q1 : q0;
        for (int32_t k = threadIdx.x; k < nq; k += blockDim.x) s_q[r % 2][k] = qc[k];
    }
    __syncthreads();
    while (nq > 0 && nq <= SMALL_CAP) {
        if (threadIdx.x == 0) s_next = 0;
        __syncthreads();
        QE *qn = (r % 2) ? q0 : q1;
        QE *ln = s_q[(r + 1) % 2];
        for (int32_t k = threadIdx.x; k < nq; k += blockDim.x) {
            auto push = [&](int32_t t, uint32_t ct) {
                QE e; e.c = t; e.cw = ct;
                const int32_t slot = agg_slot(&s_next);
                if (slot < SMALL_CAP) ln[slot] = e;
                qn[slot] = e;
            };
            if (WHICH == 0) edge_flood_cell(E, s_q[r % 2][k], &s_rcount, &s_tcount, push);
            else edge_sweep_cell(E, s_q[r % 2][k], push);
        }
        __syncthreads();
        nq = s_next;
        r++;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        cnt3[r % 3] = nq; cnt3[(r + 1) % 3] = 0; cnt3[(r + 2) % 3] = 0;
        *E.rcount = s_rcount; *E.tcount = s_tcount;
        state[0] = r;
    }
}

// pit edge offsets per cell for the edge rounds (the main sweep keeps them in the area slots it is
// about to overwrite; afterwards the contribution array is free and holds them for good)
__global__ void k_pit_offsets(const int32_t *__restrict__ pin_dst, const int32_t *__restrict__ pit_src, int64_t ne, int2 *off)
{
    int32_t *slots = reinterpret_cast<int32_t *>(off);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += (int64_t)gridDim.x * blockDim.x) {
        if (e == 0 || pin_dst[e - 1] != pin_dst[e]) slots[2 * (int64_t)pin_dst[e]] = (int32_t)e;
        if (e == 0 || pit_src[e - 1] != pit_src[e]) slots[2 * (int64_t)pit_src[e] + 1] = (int32_t)e;
    }
}

// self.uca += area (:769) on the reached cells
__global__ void k_edge_apply(EdgeArgs E, double *__restrict__ uca, const int32_t *nr)
{
    const int32_t n = *nr;
    for (int32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) {
        const int32_t c = E.rlist[q];
        // cells the sweep never reached (cyclic drainage) keep their initial value, like the reference
        uca[c] += (E.flag[c] & EF_DONE) ? E.delta[c] : edge_base(E, c);
    }
}

// ... plus the finished edge cells the flood never reached
__global__ void k_edge_apply_perimeter(EdgeArgs E, double *__restrict__ uca)
{
    const int n = E.G.n, m = E.G.m;
    const int64_t nper = 2 * (int64_t)m + 2 * (int64_t)(n - 2);
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nper) return;
    int i, j;
    if (p < m) { i = 0; j = (int)p; }
    else if (p < 2 * (int64_t)m) { i = n - 1; j = (int)(p - m); }
    else if (p < 2 * (int64_t)m + (n - 2)) { i = (int)(p - 2 * (int64_t)m) + 1; j = 0; }
    else { i = (int)(p - 2 * (int64_t)m - (n - 2)) + 1; j = m - 1; }
    const int32_t c = i * m + j;
    if (E.p_done[p] && !(E.flag[c] & EF_S)) uca[c] += edge_base(E, c);
}

// wipe the per-round state of every cell this round touched
__global__ void k_edge_cleanup(EdgeArgs E, const int32_t *nr, const int32_t *nt)
{
    const int32_t n_r = *nr, n_t = *nt;
    for (int32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < n_r + n_t; q += gridDim.x * blockDim.x) {
        const int32_t c = q < n_r ? E.rlist[q] : E.tlist[q - n_r];
        E.flag[c] = 0;
        if (q < n_r) E.G.cinfo[c] &= CI_STATIC_MASK;
    }
}


// ------------------------------------------------------------------------------- K7i
// Incremental edge rounds (the pool schedule of the ProcessManager).  A round of the kind above walks
// everything downstream of its seeds -- whole rivers -- although most of those cells stay downstream of
// another unresolved inlet and are of no use to anybody until that one resolves too; a river that runs along a
// tile border hands a seed across it dozens of times and each hand-over re-walks the rest of the river
// (measured: 247 rounds, 3-7 ms each, for an 8-tile mosaic).  Here the state of the fix-up persists between
// rounds instead:
//   * ND, the cells that are not 'done' (edge_done == 0: downstream of an unresolved inlet), is closed under
//     "downstream of"; the count field of the graph word holds, for every cell, the number of its in-edges
//     that come from ND cells, plus ONE for the outside of the tile while the cell is an unresolved inlet
//     (edge_todo == 1);
//   * a cell is 'done' when its count reaches zero (nothing unresolved is left upstream of it): it then PULLS
//     the deltas of its upstream cells in the fixed neighbour order (deterministic, no floating-point
//     atomics), adds the sum to its area and counts its targets down.  A seed adopts the neighbour's finished
//     value when the strip arrives (it never receives, cyutils.pyx:159-161) and loses its outside edge; it is
//     'done' like any other cell, when its count reaches zero, and only then lets go of its targets;
//   * deltas wait in the FINAL upstream cells (delta[]) until the cell below them becomes final: every cell is
//     processed exactly once in the whole fix-up, and a round only costs the chain of cells it finishes.
// When the fix-up ends, the cells that are still not done (their inlet never resolved) pull what their FINAL
// upstream cells hold (pydem_uca_edge_flush): the reference propagates those partial sums round by round
// (:836-842), the areas agree up to the order of the additions.  'done' / 'todo' masks after every round are
// the reference's: edge_done = not downstream of a remaining 'todo' inlet (:848-856), edge_todo = the inlets
// that stay 'todo' (:817).
constexpr uint32_t EF_FINAL = 16u, EF_NAN = 32u;      // EF_NAN: flooded by a NaN seed (k_einc_nan_flood)

struct IncArgs {
    SweepArgs G;
    uint32_t *flag;          // [NN] EF_FINAL
    double *delta;           // [NN] valid where EF_FINAL
    const uint8_t *flats;
    uint8_t *edge_done, *edge_todo;
    double *uca;
    const int2 *pit_off;
    int set_done;            // 0 in the final flush: the cells stay 'not done'
    int32_t *nanq, *n_nan;   // cells whose seed value is NaN (k_einc_nan_flood)
    uint32_t round16;        // this round's number (mod 2^16, never 0); flag bits 16-31 = round that last seeded the cell
    int32_t *prof;           // -DPYDEM_EINC_PROF: levels / 10 ns ticks by frontier width (<=8, <=64, <=512, more)
};

// once per fix-up: counts of the ND sub-graph
__global__ __launch_bounds__(256) void k_einc_prepare(IncArgs E, int64_t NN)
{
    const SweepArgs &A = E.G;
    for (int64_t c64 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c64 < NN; c64 += (int64_t)gridDim.x * blockDim.x) {
        if (E.edge_done[c64]) continue;
        const int32_t u = (int32_t)c64;
        E.delta[u] = 0.0;
        const uint32_t cw = A.cinfo[u];
        const int s = ci_section(cw);
        if (cw & CI_OUT1) atomicAdd(&A.cinfo[u + fe1r(s) * A.m + fe1c(s)], CI_EONE);
        if (cw & CI_OUT2) atomicAdd(&A.cinfo[u + fe2r(s) * A.m + fe2c(s)], CI_EONE);
        if (cw & CI_PIT_OUT)
            for (int32_t e = E.pit_off[u].y; e < A.n_pit && A.pit_src[e] == u; e++) atomicAdd(&A.cinfo[A.pit_dst[e]], CI_EONE);
        if (E.edge_todo[u]) atomicAdd(&A.cinfo[u], CI_EONE);                     // the outside of the tile
    }
}

__device__ __forceinline__ void perim_cell(int64_t p, int n, int m, int &i, int &j)
{
    if (p < m) { i = 0; j = (int)p; }
    else if (p < 2 * (int64_t)m) { i = n - 1; j = (int)(p - m); }
    else if (p < 2 * (int64_t)m + (n - 2)) { i = (int)(p - 2 * (int64_t)m) + 1; j = 0; }
    else { i = (int)(p - 2 * (int64_t)m - (n - 2)) + 1; j = m - 1; }
}

// strips -> events on the perimeter (:726-739, :798-809).  Queue entries are cells whose count reached zero.
__global__ void k_einc_seed(IncArgs E, const double *__restrict__ sdata, const uint8_t *__restrict__ sdone,
                            const uint8_t *__restrict__ stodo, int L, QE *q, int32_t *nq)
{
    const int n = E.G.n, m = E.G.m;
    const int64_t nper = 2 * (int64_t)m + 2 * (int64_t)(n - 2);
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nper) return;
    int i, j;
    perim_cell(p, n, m, i, j);
    const int32_t c = i * m + j;
    bool dn = false, td = false;
    double init = 0.0;
    // dict order of the reference: left, right, top, bottom
    if (j == 0) { dn |= sdone[0 * L + i] != 0; init += sdata[0 * L + i] * (double)(sdone[0 * L + i] != 0); td |= stodo[0 * L + i] != 0; }
    if (j == m - 1) { dn |= sdone[1 * L + i] != 0; init += sdata[1 * L + i] * (double)(sdone[1 * L + i] != 0); td |= stodo[1 * L + i] != 0; }
    if (i == 0) { dn |= sdone[2 * L + j] != 0; init += sdata[2 * L + j] * (double)(sdone[2 * L + j] != 0); td |= stodo[2 * L + j] != 0; }
    if (i == n - 1) { dn |= sdone[3 * L + j] != 0; init += sdata[3 * L + j] * (double)(sdone[3 * L + j] != 0); td |= stodo[3 * L + j] != 0; }
    const bool own_todo = E.edge_todo[c] != 0;
    const bool own_done = E.edge_done[c] != 0;
    const uint32_t cw = E.G.cinfo[c] & CI_STATIC_MASK;
    if (dn) {
        const double d = E.flats[c] ? NAN : init - E.uca[c];                     // :806-809, :815
        if (!own_done) E.flag[c] = (E.flag[c] & 0xFFFFu) | (E.round16 << 16);     // a seed of this round: upstream values do not enter it
        if (!(E.flag[c] & EF_FINAL) && !own_done) {
            // a seed (:798), or a cell below one of the tile's own unresolved inlets whose neighbour copy is finished:
            // it adopts the finished value (it will not pull) and holds the difference for the cells below it.  It is
            // 'done' -- and lets go of its targets -- once nothing unresolved is left upstream of it inside the tile
            E.uca[c] += d;
            E.delta[c] = d;
            E.flag[c] = (E.flag[c] & (EF_NAN | 0xFFFF0000u)) | EF_FINAL;
            if (d != d) E.nanq[atomicAdd(E.n_nan, 1)] = c;
            E.edge_todo[c] = 0;
            if (own_todo) {
                const uint32_t old = atomicSub(&E.G.cinfo[c], CI_EONE);          // the outside of the tile
                if (ci_ecount(old) == 1u) { QE e; e.c = c; e.cw = cw; q[agg_slot(nq)] = e; }
            }
        } else {
            E.uca[c] += d;                                                      // finished on both sides: re-synchronised
            E.edge_todo[c] = 0;
            if (!own_done) {                                                    // ... or a seed of an earlier round that still waits: the
                E.delta[c] += d;                                                // difference joins what it holds for its targets
                if (d != d) E.nanq[atomicAdd(E.n_nan, 1)] = c;
            }
        }
    } else if (own_todo && !td) {
        // the 'todo' flag was dropped without a value (rule :274 / the mosaic border): the outside edge goes away
        E.edge_todo[c] = 0;
        const uint32_t old = atomicSub(&E.G.cinfo[c], CI_EONE);
        if (ci_ecount(old) == 1u) { QE e; e.c = c; e.cw = cw; q[agg_slot(nq)] = e; }
    }
}

// the final flush: the remaining inlets let go of the outside (their flags stay)
__global__ void k_einc_release_todo(IncArgs E, QE *q, int32_t *nq)
{
    const int n = E.G.n, m = E.G.m;
    const int64_t nper = 2 * (int64_t)m + 2 * (int64_t)(n - 2);
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nper) return;
    int i, j;
    perim_cell(p, n, m, i, j);
    const int32_t c = i * m + j;
    if (!E.edge_todo[c] || (E.flag[c] & EF_FINAL)) return;
    const uint32_t old = atomicSub(&E.G.cinfo[c], CI_EONE);
    if (ci_ecount(old) == 1u) { QE e; e.c = c; e.cw = E.G.cinfo[c] & CI_STATIC_MASK; q[agg_slot(nq)] = e; }
}

template <typename Push>
__device__ __forceinline__ void einc_cell(const IncArgs &E, QE q, Push push)
{
    const SweepArgs &A = E.G;
    const int32_t c = q.c;
    const uint32_t cw = q.cw;
    const int m = A.m;
    // every load of the cell in ONE batch (own flag / area, the in-neighbours' flag / delta / proportion): the cascade
    // is a chain of dependent memory round trips and nothing else
    int2 po = make_int2(0, 0);
    if (cw & (CI_PIT_IN | CI_PIT_OUT)) po = E.pit_off[c];
    const uint32_t own_flag = E.flag[c];
    const double own_uca = E.uca[c];
    const bool flat = E.flats[c] != 0;
    uint32_t f[8]; double dl[8], pr[8];
#pragma unroll
    for (int d = 0; d < 8; d++) {
        f[d] = 0; dl[d] = 0.0; pr[d] = 0.0;
        if (cw & (1u << d)) {
            const int32_t u = c + NB_DI[d] * m + NB_DJ[d];
            f[d] = E.flag[u]; dl[d] = E.delta[u]; pr[d] = A.prop[u];
        }
    }
    if (!(own_flag & EF_FINAL)) {                                                // (seeds keep the value they adopted)
        double acc = flat ? NAN : 0.0;                                           // :815
#pragma unroll
        for (int d = 0; d < 8; d++) {
            if ((cw & (1u << d)) && (f[d] & EF_FINAL)) {
                const bool cardinal = (NB_DI[d] == 0) || (NB_DJ[d] == 0);
                acc += dl[d] * (cardinal ? pr[d] : 1 - pr[d]);
            }
        }
        if (cw & CI_PIT_IN)
            for (int32_t e = po.x; e < A.n_pit && A.pin_dst[e] == c; e++)
                if (E.flag[A.pin_src[e]] & EF_FINAL) acc += E.delta[A.pin_src[e]] * A.pin_w[e];
        E.delta[c] = acc;
        E.uca[c] = own_uca + acc;
        E.flag[c] = (own_flag & EF_NAN) | EF_FINAL;
    }
    if (E.set_done) E.edge_done[c] = 1;
    const int s = ci_section(cw);
    auto release = [&](int32_t t) {
        const uint32_t old = atomicSub(&A.cinfo[t], CI_EONE);
        if (ci_ecount(old) == 1u) push(t, old & CI_STATIC_MASK);
    };
    if (cw & CI_OUT1) release(c + fe1r(s) * m + fe1c(s));
    if (cw & CI_OUT2) release(c + fe2r(s) * m + fe2c(s));
    if (cw & CI_PIT_OUT)
        for (int32_t e = po.y; e < A.n_pit && A.pit_src[e] == c; e++) release(A.pit_dst[e]);
}

// NaN is absorbing in the reference's rounds and floods everything below the seed in the round it arrives (see
// k_cinc_nan_flood for the argument); cell-indexed form: breadth first over the out-edges of the graph words
__global__ __launch_bounds__(1024) void k_einc_nan_flood(IncArgs E)
{
    __shared__ int32_t s_tail;
    const SweepArgs &A = E.G;
    if (threadIdx.x == 0) s_tail = *E.n_nan;
    __syncthreads();
    int32_t head = 0, tail = s_tail;
    const int32_t n_origin = tail;             // the NaN seeds themselves
    while (head < tail) {
        for (int32_t q = head + threadIdx.x; q < tail; q += blockDim.x) {
            const int32_t c = E.nanq[q];
            if (q >= n_origin && (E.flag[c] >> 16) == E.round16) continue;       // a seed of this round keeps its value
            if (atomicOr(&E.flag[c], EF_NAN) & EF_NAN) continue;
            E.uca[c] = NAN;
            const uint32_t cw = A.cinfo[c];
            const int s = ci_section(cw);
            auto visit = [&](int32_t t) { if (!(E.flag[t] & EF_NAN)) E.nanq[atomicAdd(&s_tail, 1)] = t; };
            if (cw & CI_OUT1) visit(c + fe1r(s) * A.m + fe1c(s));
            if (cw & CI_OUT2) visit(c + fe2r(s) * A.m + fe2c(s));
            if (cw & CI_PIT_OUT)
                for (int32_t e = E.pit_off[c].y; e < A.n_pit && A.pit_src[e] == c; e++) visit(A.pit_dst[e]);
        }
        __syncthreads();
        head = tail; tail = s_tail;
        __syncthreads();
    }
    if (threadIdx.x == 0) *E.n_nan = 0;
}

__global__ __launch_bounds__(256) void k_einc_level(IncArgs E, const QE *__restrict__ qc, QE *__restrict__ qn, int32_t *cnt3, int r)
{
    const int32_t nq = cnt3[r % 3];
    if (blockIdx.x == 0 && threadIdx.x == 0) cnt3[(r + 2) % 3] = 0;
    if (nq == 0) return;
    int32_t *cn = &cnt3[(r + 1) % 3];
    for (int32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < nq; k += gridDim.x * blockDim.x) {
        auto push = [&](int32_t t, uint32_t ct) { QE e; e.c = t; e.cw = ct; qn[agg_slot(cn)] = e; };
        einc_cell(E, qc[k], push);
    }
}

// small frontiers: one workgroup, level after level (see k_edge_small)
__global__ __launch_bounds__(1024) void k_einc_small(IncArgs E, QE *q0, QE *q1, int32_t *cnt3, int r_start, int32_t *state)   // (launched with 64..1024 threads)
{
    __shared__ QE s_q[2][SMALL_CAP];
    __shared__ int32_t s_next;
    int r = r_start;
    int32_t nq = cnt3[r % 3];
    if (nq > 0 && nq <= SMALL_CAP) {
        const QE *qc = (r % 2) ? q1 : q0;
        for (int32_t k = threadIdx.x; k < nq; k += blockDim.x) s_q[r % 2][k] = qc[k];
    }
#ifdef PYDEM_EINC_PROF
    long long prof_t[4] = {0, 0, 0, 0}; int prof_n[4] = {0, 0, 0, 0};
#endif
    __syncthreads();
    while (nq > 0 && nq <= SMALL_CAP) {
#ifdef PYDEM_EINC_PROF
        const long long t0 = wall_clock64();
        const int cls = nq <= 8 ? 0 : (nq <= 64 ? 1 : (nq <= 512 ? 2 : 3));
#endif
        if (threadIdx.x == 0) s_next = 0;
        __syncthreads();
        QE *qn = (r % 2) ? q0 : q1;
        QE *ln = s_q[(r + 1) % 2];
        for (int32_t k = threadIdx.x; k < nq; k += blockDim.x) {
            auto push = [&](int32_t t, uint32_t ct) {
                QE e; e.c = t; e.cw = ct;
                const int32_t slot = agg_slot(&s_next);
                if (slot < SMALL_CAP) ln[slot] = e;
                qn[slot] = e;
            };
            einc_cell(E, s_q[r % 2][k], push);
        }
        __syncthreads();
        nq = s_next;
        r++;
        __syncthreads();
#ifdef PYDEM_EINC_PROF
        prof_t[cls] += wall_clock64() - t0; prof_n[cls]++;
#endif
    }
    if (threadIdx.x == 0) {
        cnt3[r % 3] = nq; cnt3[(r + 1) % 3] = 0; cnt3[(r + 2) % 3] = 0;
        state[0] = r;
#ifdef PYDEM_EINC_PROF
        for (int k = 0; k < 4; k++) { atomicAdd(&E.prof[k], prof_n[k]); atomicAdd(&E.prof[4 + k], (int)prof_t[k]); }
#endif
    }
}


// ---- compact form of the incremental rounds ----------------------------------------------------------------
// The cascade above is a chain of dependent accesses into seven tile-sized arrays: a level costs 5-7 us, most of
// it address translation and HBM misses (measured: 35 k levels = 230 ms for an 8-tile fix-up at 16384^2) although
// the cells it will ever touch -- ND, 50-70 k per tile -- would fit the L2.  So the fix-up state moves into ONE
// 128-byte record per ND cell (compact id k: the cell, its graph word, the compact ids and weights of its
// in-edges, the ids of its two targets, count, flags, delta); a finished cell writes its contribution into its
// targets' in-slots (plain stores, one slot per edge: the sum stays in the fixed neighbour order) and counts them
// down, so a cell that becomes ready needs nothing but its own record -- ONE dependent access per level plus the
// count-down atomics.  The cascade runs on records only, and the
// areas / masks of the tile are updated from the records by a streaming kernel after the cascade (nothing in the
// chain waits for the big arrays).  Tiles whose ND set is too large for that (a tile that is one single
// catchment below its inlet edge) keep the cell-indexed form.
struct __attribute__((aligned(128))) NDRec {
    int32_t cell;
    uint32_t cw;
    int32_t out_id[2];       // compact ids of the two targets (-1: no such edge)
    uint8_t out_slot[2];     // which in-slot of the target this cell feeds (the target's neighbour index NW..SE)
    uint16_t seed_round;     // round (mod 2^16) in which the strips last initialised the cell: a NaN flood of that round stops here
    int32_t cnt;             // unresolved in-edges (+1 for the outside of the tile while the cell is a 'todo' inlet)
    uint32_t flag;
    int32_t wid;             // condensed form (uca_cond.inl): node of a watched cell, -1 otherwise
    double delta;
    double out_w[2];         // proportion, 1 - proportion (:1082)
    double in_delta[8];      // what the finished in-neighbour NW..SE has handed over (0 until then)
};
static_assert(sizeof(NDRec) == 128, "one cache line per ND cell");
constexpr uint32_t NF_FINAL = 1u, NF_DONE = 2u, NF_APPLIED = 4u, NF_SEED = 8u, NF_NAN = 16u;
constexpr uint32_t ND_FLAT = 1u << 16;            // in the record's graph word: the cell is a flat (its delta is NaN, :815)
constexpr int64_t ND_COMPACT_MAX = 6 << 20;      // records (768 MiB)

struct CIncArgs {
    SweepArgs G;
    NDRec *rec; int32_t nd;
    int32_t *cid;            // [NN] compact id + 1 (0: not an ND cell)
    const int2 *pit_off;
    const uint8_t *flats;
    uint8_t *edge_done, *edge_todo;
    double *uca;
    int set_done;
    int32_t *prof;
    int32_t *nanq, *n_nan;   // records whose delta is NaN (k_cinc_nan_flood)
    uint32_t round16;        // this round's number (mod 2^16, never 0)
};

// bytes of a 32-bit word that are zero, exactly (0x80 per zero byte)
__device__ __forceinline__ uint32_t zero_bytes(uint32_t x)
{
    const uint32_t y = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    return ~(y | x | 0x7F7F7F7Fu);
}

// (both kernels read the mask sixteen cells per load: the not-done cells are ~0.1 % of a tile, and one byte per thread made these
// two passes over a 268 MB plane 0.43 + 0.69 ms of a round-1 fix-up -- round 6)
__global__ __launch_bounds__(256) void k_nd_count(const uint8_t *__restrict__ edge_done, int64_t NN, unsigned long long *count)
{
    unsigned long long c = 0;
    const int64_t n16 = NN / 16;
    const uint4 *v = reinterpret_cast<const uint4 *>(edge_done);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) {
        const uint4 x = v[i];
        c += __popc(zero_bytes(x.x)) + __popc(zero_bytes(x.y)) + __popc(zero_bytes(x.z)) + __popc(zero_bytes(x.w));
    }
    if (blockIdx.x == 0) for (int64_t i = n16 * 16 + threadIdx.x; i < NN; i += blockDim.x) c += edge_done[i] == 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}

__device__ __forceinline__ void nd_assign_cell(const CIncArgs &E, int64_t c64, int32_t k)
{
    E.cid[c64] = k + 1;
    NDRec &R = E.rec[k];
    R.cell = (int32_t)c64;
    R.cw = (E.G.cinfo[c64] & CI_STATIC_MASK) | (E.flats[c64] ? ND_FLAT : 0u);
    R.flag = 0; R.delta = 0.0; R.seed_round = 0; R.wid = -1;
}

__global__ __launch_bounds__(256) void k_nd_assign(CIncArgs E, int64_t NN, int32_t *counter)
{
    const int64_t n16 = NN / 16;
    const uint4 *v = reinterpret_cast<const uint4 *>(E.edge_done);
    const int lane = threadIdx.x & 63;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x; i0 < n16; i0 += (int64_t)gridDim.x * blockDim.x) {      // (uniform per workgroup)
        const int64_t i = i0 + threadIdx.x;
        const uint4 x = i < n16 ? v[i] : make_uint4(~0u, ~0u, ~0u, ~0u);
        const uint32_t w[4] = {zero_bytes(x.x), zero_bytes(x.y), zero_bytes(x.z), zero_bytes(x.w)};
        const int nz = __popc(w[0]) + __popc(w[1]) + __popc(w[2]) + __popc(w[3]);
        if (__ballot(nz > 0) == 0) continue;                           // (almost always: nothing to do for these 1024 cells)
        // record ids for the wavefront's cells with ONE atomic: inclusive scan of the counts over the lanes
        int incl = nz;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        int32_t base = 0;
        if (lane == 63) base = atomicAdd(counter, incl);
        int32_t k = __shfl(base, 63) + incl - nz;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t z = w[q];
            while (z) { const int b = __ffs((int)z) - 1; z &= z - 1; nd_assign_cell(E, i * 16 + q * 4 + (b >> 3), k++); }
        }
    }
    if (blockIdx.x == 0) for (int64_t c64 = n16 * 16 + threadIdx.x; c64 < NN; c64 += blockDim.x) if (!E.edge_done[c64]) nd_assign_cell(E, c64, atomicAdd(counter, 1));
}

__global__ __launch_bounds__(256) void k_nd_link(CIncArgs E)
{
    const SweepArgs &A = E.G;
    const int m = A.m;
    if (blockIdx.x == 0 && threadIdx.x == 0) {                                   // the sink of the missing edges
        NDRec &S = E.rec[E.nd];
        S.cell = -1; S.cw = 0; S.out_id[0] = S.out_id[1] = -1; S.out_slot[0] = S.out_slot[1] = 0; S.cnt = 1 << 30; S.flag = 0; S.delta = 0.0; S.wid = -1;
    }
    for (int32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < E.nd; k += gridDim.x * blockDim.x) {
        NDRec &R = E.rec[k];
        const int32_t c = R.cell;
        const uint32_t cw = R.cw;
        int32_t cnt = 0;
#pragma unroll
        for (int d = 0; d < 8; d++) {
            if (cw & (1u << d)) cnt += E.cid[c + NB_DI[d] * m + NB_DJ[d]] != 0;
            R.in_delta[d] = 0.0;
        }
        if (cw & CI_PIT_IN)
            for (int32_t e = E.pit_off[c].x; e < A.n_pit && A.pin_dst[e] == c; e++) cnt += E.cid[A.pin_src[e]] != 0;
        const int s = ci_section(cw);
        const double p = A.prop[c];
        const int dr[2] = {fe1r(s), fe2r(s)}, dc[2] = {fe1c(s), fe2c(s)};
        const uint32_t has[2] = {cw & CI_OUT1, cw & CI_OUT2};
        for (int j = 0; j < 2; j++) {
            int32_t id = -1; int slot = 0;
            if (has[j]) {
                id = E.cid[c + dr[j] * m + dc[j]] - 1;
                // seen from the target, this cell sits at (-dr, -dc): its index in the neighbour order NW..SE
                for (int d = 0; d < 8; d++) if (NB_DI[d] == -dr[j] && NB_DJ[d] == -dc[j]) slot = d;
            }
            R.out_id[j] = id; R.out_slot[j] = (uint8_t)slot;
        }
        R.out_w[0] = p; R.out_w[1] = 1 - p;
        if (E.edge_todo[c]) cnt += 1;                                            // the outside of the tile
        R.cnt = cnt;
    }
}

__global__ void k_cinc_seed(CIncArgs E, const double *__restrict__ sdata, const uint8_t *__restrict__ sdone,
                            const uint8_t *__restrict__ stodo, int L, QE *q, int32_t *nq)
{
    const int n = E.G.n, m = E.G.m;
    const int64_t nper = 2 * (int64_t)m + 2 * (int64_t)(n - 2);
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nper) return;
    int i, j;
    perim_cell(p, n, m, i, j);
    const int32_t c = i * m + j;
    bool dn = false, td = false;
    double init = 0.0;
    if (j == 0) { dn |= sdone[0 * L + i] != 0; init += sdata[0 * L + i] * (double)(sdone[0 * L + i] != 0); td |= stodo[0 * L + i] != 0; }
    if (j == m - 1) { dn |= sdone[1 * L + i] != 0; init += sdata[1 * L + i] * (double)(sdone[1 * L + i] != 0); td |= stodo[1 * L + i] != 0; }
    if (i == 0) { dn |= sdone[2 * L + j] != 0; init += sdata[2 * L + j] * (double)(sdone[2 * L + j] != 0); td |= stodo[2 * L + j] != 0; }
    if (i == n - 1) { dn |= sdone[3 * L + j] != 0; init += sdata[3 * L + j] * (double)(sdone[3 * L + j] != 0); td |= stodo[3 * L + j] != 0; }
    const bool own_todo = E.edge_todo[c] != 0;
    const bool own_done = E.edge_done[c] != 0;
    const int32_t k = E.cid[c] - 1;
    if (dn) {
        const double d = E.flats[c] ? NAN : init - E.uca[c];
        E.uca[c] += d;
        E.edge_todo[c] = 0;
        if (k >= 0 && !own_done) E.rec[k].seed_round = (uint16_t)E.round16;     // a seed of this round (:798): upstream values do not enter it
        if (k >= 0 && !own_done && !(E.rec[k].flag & NF_FINAL)) {
            NDRec &R = E.rec[k];
            R.delta = d;
            R.flag = (R.flag & NF_NAN) | NF_FINAL | NF_SEED;
            if (d != d) E.nanq[atomicAdd(E.n_nan, 1)] = k;                       // (k_cinc_nan_flood)
            if (own_todo) {
                const int32_t old = atomicSub(&R.cnt, 1);
                if (old == 1) { QE e; e.c = k; e.cw = 0; q[agg_slot(nq)] = e; }
            }
        } else if (k >= 0 && !own_done) {
            // a seed of an earlier round that still waits for its own upstream cells, and the neighbour's copy has moved
            // on since: the reference re-initialises it in every round (`area_edges - uca`, :806-809) and lets the
            // difference run down; here it joins what the cell holds for its targets
            E.rec[k].delta += d;
            if (d != d) E.nanq[atomicAdd(E.n_nan, 1)] = k;
        }
    } else if (own_todo && !td) {
        E.edge_todo[c] = 0;
        if (k >= 0) {
            const int32_t old = atomicSub(&E.rec[k].cnt, 1);
            if (old == 1) { QE e; e.c = k; e.cw = 0; q[agg_slot(nq)] = e; }
        }
    }
}

__global__ void k_cinc_release_todo(CIncArgs E, QE *q, int32_t *nq)
{
    const int n = E.G.n, m = E.G.m;
    const int64_t nper = 2 * (int64_t)m + 2 * (int64_t)(n - 2);
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nper) return;
    int i, j;
    perim_cell(p, n, m, i, j);
    const int32_t c = i * m + j;
    const int32_t k = E.cid[c] - 1;
    if (k < 0 || !E.edge_todo[c] || (E.rec[k].flag & NF_FINAL)) return;
    const int32_t old = atomicSub(&E.rec[k].cnt, 1);
    if (old == 1) { QE e; e.c = k; e.cw = 0; q[agg_slot(nq)] = e; }
}

template <typename Push>
__device__ __forceinline__ void cinc_cell(const CIncArgs &E, QE q, Push push)
{
    // (letting the lane walk on along the chain it releases -- one count-down atomic plus one record load per step, no
    // queue, no barrier -- was measured and is slower: 323 instead of 209 ms of rounds for the 8-tile fix-up at
    // 16384^2; the side branches a walking lane pushes wait for the whole walk)
    const int32_t k = q.c;
    NDRec &R = E.rec[k];
    // the only dependent access of a level: the cell's own record, one cache line, loaded whole
    const uint4 *line = reinterpret_cast<const uint4 *>(&R);
    uint4 L[8];
#pragma unroll
    for (int i = 0; i < 8; i++) L[i] = line[i];
    NDRec V;
    __builtin_memcpy(&V, L, sizeof(V));
    const uint32_t cw = V.cw;
    double delta = V.delta;
    if (!(V.flag & NF_FINAL)) {
        double acc = (cw & ND_FLAT) ? NAN : 0.0;                                 // :815
#pragma unroll
        for (int d = 0; d < 8; d++) acc += V.in_delta[d];                        // fixed order NW..SE; untouched slots are 0
        if (cw & CI_PIT_IN) {
            // (fetching the pits eight at a time -- list entries, record ids, records each together -- was measured in round 6 and is
            // slower: most drains have one or two pits, and the flush went from 6.2-7.6 to 6.7-8.3 ms per tile)
            const SweepArgs &A = E.G;
            for (int32_t e = E.pit_off[V.cell].x; e < A.n_pit && A.pin_dst[e] == V.cell; e++) {
                const int32_t ks = E.cid[A.pin_src[e]] - 1;
                if (ks >= 0 && (E.rec[ks].flag & NF_FINAL)) acc += E.rec[ks].delta * A.pin_w[e];
            }
        }
        delta = acc;
        R.delta = acc;
        R.flag = (V.flag & NF_NAN) | NF_FINAL | NF_DONE;
    } else {
        R.flag = V.flag | NF_DONE;                                               // a seed keeps the value it adopted
    }
    // hand the contribution over, then count the targets down.  A missing edge points at the sink record rec[nd]
    // (its count never reaches zero), so both stores and both atomics are issued unconditionally, back to back
    const int32_t o0 = V.out_id[0] >= 0 ? V.out_id[0] : E.nd, o1 = V.out_id[1] >= 0 ? V.out_id[1] : E.nd;
    E.rec[o0].in_delta[V.out_slot[0]] = delta * V.out_w[0];
    E.rec[o1].in_delta[V.out_slot[1]] = delta * V.out_w[1];
    const int32_t old0 = atomicSub(&E.rec[o0].cnt, 1);
    const int32_t old1 = atomicSub(&E.rec[o1].cnt, 1);
    if (old0 == 1) push(o0, 0u);
    if (old1 == 1) push(o1, 0u);
    if (cw & CI_PIT_OUT) {
        const SweepArgs &A = E.G;
        for (int32_t e = E.pit_off[V.cell].y; e < A.n_pit && A.pit_src[e] == V.cell; e++) {
            const int32_t kt = E.cid[A.pit_dst[e]] - 1;
            if (kt >= 0 && atomicSub(&E.rec[kt].cnt, 1) == 1) push(kt, 0u);
        }
    }
}

// A round of the reference propagates whatever a seed carries through ALL cells below it at once (:826-840), and NaN is
// absorbing there: a cell that received NaN in one round stays NaN when a later round re-initialises it from a finished
// neighbour (`area_edges - uca`, :806-809).  The incremental rounds hold deltas back until a cell's last upstream cell is
// done, and a cell that adopts a neighbour's value in the meantime drops them -- harmless for numbers (the adopted value
// contains them), wrong for NaN.  So a NaN seed floods its NaN through everything downstream in the round it arrives --
// except the other seeds of that round, which are 'done' from the start in the reference's sweep and take nothing from
// upstream (:826-829): one workgroup, breadth first over the out-links of the records, each record claimed once.
__global__ __launch_bounds__(1024) void k_cinc_nan_flood(CIncArgs E)
{
    __shared__ int32_t s_tail;
    if (threadIdx.x == 0) s_tail = *E.n_nan;
    __syncthreads();
    int32_t head = 0, tail = s_tail;
    const int32_t n_origin = tail;             // the NaN seeds themselves
    while (head < tail) {
        for (int32_t q = head + threadIdx.x; q < tail; q += blockDim.x) {
            const int32_t k = E.nanq[q];
            NDRec &R = E.rec[k];
            if (q >= n_origin && R.seed_round == (uint16_t)E.round16) continue;  // a seed of this round keeps its value
            if (atomicOr(&R.flag, NF_NAN) & NF_NAN) continue;                     // flooded in an earlier round
            E.uca[R.cell] = NAN;
            for (int o = 0; o < 2; o++) {
                const int32_t t = R.out_id[o];
                if (t >= 0 && !(E.rec[t].flag & NF_NAN)) E.nanq[atomicAdd(&s_tail, 1)] = t;
            }
            if (R.cw & CI_PIT_OUT) {
                const SweepArgs &A = E.G;
                for (int32_t e = E.pit_off[R.cell].y; e < A.n_pit && A.pit_src[e] == R.cell; e++) {
                    const int32_t t = E.cid[A.pit_dst[e]] - 1;
                    if (t >= 0 && !(E.rec[t].flag & NF_NAN)) E.nanq[atomicAdd(&s_tail, 1)] = t;
                }
            }
        }
        __syncthreads();
        head = tail; tail = s_tail;
        __syncthreads();
    }
    if (threadIdx.x == 0) *E.n_nan = 0;
}

__global__ __launch_bounds__(256) void k_cinc_level(CIncArgs E, const QE *__restrict__ qc, QE *__restrict__ qn, int32_t *cnt3, int r)
{
    const int32_t nq = cnt3[r % 3];
    if (blockIdx.x == 0 && threadIdx.x == 0) cnt3[(r + 2) % 3] = 0;
    if (nq == 0) return;
    int32_t *cn = &cnt3[(r + 1) % 3];
    for (int32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < nq; k += gridDim.x * blockDim.x) {
        auto push = [&](int32_t t, uint32_t ct) { QE e; e.c = t; e.cw = ct; qn[agg_slot(cn)] = e; };
        cinc_cell(E, qc[k], push);
    }
}

__global__ __launch_bounds__(1024) void k_cinc_small(CIncArgs E, QE *q0, QE *q1, int32_t *cnt3, int r_start, int32_t *state, int cap)
{
    __shared__ QE s_q[2][SMALL_CAP];
    __shared__ int32_t s_cnt[3];     // pushes of level r go to s_cnt[(r + 1) % 3]; s_cnt[(r + 2) % 3] is zeroed meanwhile: ONE barrier per level
    int r = r_start;
    int32_t nq = cnt3[r % 3];
    if (nq > 0 && nq <= cap) {
        const QE *qc = (r % 2) ? q1 : q0;
        for (int32_t k = threadIdx.x; k < nq; k += blockDim.x) s_q[r % 2][k] = qc[k];
    }
    if (threadIdx.x == 0) { s_cnt[0] = s_cnt[1] = s_cnt[2] = 0; }
#ifdef PYDEM_EINC_PROF
    long long prof_t[4] = {0, 0, 0, 0}; int prof_n[4] = {0, 0, 0, 0};
#endif
    __syncthreads();
    while (nq > 0 && nq <= cap) {
#ifdef PYDEM_EINC_PROF
        const long long t0 = wall_clock64();
        const int cls = nq <= 8 ? 0 : (nq <= 64 ? 1 : (nq <= 512 ? 2 : 3));
#endif
        int32_t *cn = &s_cnt[(r + 1) % 3];
        if (threadIdx.x == 0) s_cnt[(r + 2) % 3] = 0;
        QE *qn = (r % 2) ? q0 : q1;
        QE *ln = s_q[(r + 1) % 2];
        for (int32_t k = threadIdx.x; k < nq; k += blockDim.x) {
            auto push = [&](int32_t t, uint32_t ct) {
                QE e; e.c = t; e.cw = ct;
                const int32_t slot = agg_slot(cn);
                if (slot < SMALL_CAP) ln[slot] = e;
                else qn[slot] = e;                           // beyond the LDS queue: straight to the global one
            };
            cinc_cell(E, s_q[r % 2][k], push);
        }
        __syncthreads();
        nq = *cn;
        r++;
#ifdef PYDEM_EINC_PROF
        prof_t[cls] += wall_clock64() - t0; prof_n[cls]++;
#endif
    }
    if (nq > cap && r > r_start) {
        // the frontier outgrew the workgroup: the level kernels take over from the global queue, whose head is still in LDS
        QE *qg = (r % 2) ? q1 : q0;
        for (int32_t k = threadIdx.x; k < SMALL_CAP; k += blockDim.x) qg[k] = s_q[r % 2][k];
    }
    if (threadIdx.x == 0) {
        cnt3[r % 3] = nq; cnt3[(r + 1) % 3] = 0; cnt3[(r + 2) % 3] = 0;
        state[0] = r;
#ifdef PYDEM_EINC_PROF
        for (int k = 0; k < 4; k++) { atomicAdd(&E.prof[k], prof_n[k]); atomicAdd(&E.prof[4 + k], (int)prof_t[k]); }
#endif
    }
}

// records -> tile: areas and masks of the cells the last cascade finished
__global__ __launch_bounds__(256) void k_cinc_apply(CIncArgs E)
{
    for (int32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < E.nd; k += gridDim.x * blockDim.x) {
        NDRec &R = E.rec[k];
        const uint32_t f = R.flag;
        if ((f & NF_DONE) && !(f & NF_APPLIED)) {
            if (!(f & NF_SEED)) E.uca[R.cell] += R.delta;                       // (a seed took its value when the strip arrived)
            if (E.set_done) E.edge_done[R.cell] = 1;
            R.flag = f | NF_APPLIED;
        }
    }
}

int grid_for(int64_t work, int cap) { const int64_t g = cdiv(work, 256); return (int)(g < cap ? (g > 0 ? g : 1) : cap); }

#include "uca_cond.inl"
#include "uca_cbuild.inl"

}  // namespace

int stage_section_graph(pydem_tile *t, const pydem_options *opt)
{
    const int n = (int)t->n, m = (int)t->m;
    t->edge_clean = false;
    t->einc_ready = false;
    PYDEM_TRY(tile_alloc(t, &t->todo_work, (size_t)t->NN));
    PYDEM_TRY(tile_alloc(t, &t->indeg, (size_t)t->NN));          // the cinfo words
    const dim3 grid2((unsigned)(cdiv(m, 256) < 64 ? cdiv(m, 256) : 64), (unsigned)(n < 16384 ? n : 16384));
    t->tm.n_pit_edges = 0; t->tm.n_pits_undrained = 0; t->tm.pits_ms = 0;
    t->pits.n_edges = 0; t->pits.n_raw = 0;
    // The section / out-flag kernel and the in-mask kernel are streaming kernels that share nothing with the pit search
    // (bound by instruction issue, little memory traffic) except the flats mask: the reference derives section and
    // proportion from the mask as it is BEFORE the pits are patched (:1021-1070 runs ahead of _mk_adjacency_matrix).  A
    // snapshot of the mask (the todo_work bytes are idle until the in-mask kernel clears them) lets both kernels run on
    // the side stream while the main stream searches the pits and patches flats / mag.
    double *corner_sums = (double *)(t->counters + 16);                          // 12 doubles inside the counter block
    HIP_TRY(hipEventRecord(t->ev_fork, t->stream));
    HIP_TRY(hipStreamWaitEvent(t->stream2, t->ev_fork, 0));
    HIP_TRY(hipMemcpyAsync(t->todo_work, t->flats, (size_t)t->NN, hipMemcpyDeviceToDevice, t->stream2));
    HIP_TRY(hipEventRecord(t->ev_snap, t->stream2));
    HIP_TRY(hipStreamWaitEvent(t->stream, t->ev_snap, 0));                       // the pit search may patch flats from here on
    hipLaunchKernelGGL(k_section_proportion, grid2, dim3(256), 0, t->stream2, t->dir, (const uint8_t *)t->todo_work, t->sec_theta,
                       t->NN, n, m, t->elev, t->section, t->prop, (uint32_t *)t->indeg);
    HIP_TRY(hipMemsetAsync(t->counters + 16, 0, 24 * sizeof(int32_t), t->stream2));
    HIP_TRY(hipMemsetAsync(t->edge_todo, 0, (size_t)t->NN, t->stream2));
    HIP_TRY(hipMemsetAsync(t->todo_work, 0, (size_t)t->NN, t->stream2));
    hipLaunchKernelGGL(k_graph_inmask, grid2, dim3(256), 0, t->stream2, t->prop, t->elev, n, m,
                       (uint32_t *)t->indeg, t->edge_todo, t->todo_work, corner_sums);
    HIP_TRY(hipEventRecord(t->ev_join, t->stream2));
    static int graph_serial = -1;      // PYDEM_GRAPH_SERIAL=1 (measurements): the graph kernels before the pit search instead of beside it
    if (graph_serial < 0) { const char *e = getenv("PYDEM_GRAPH_SERIAL"); graph_serial = e ? atoi(e) : 0; }
    if (graph_serial) HIP_TRY(hipStreamWaitEvent(t->stream, t->ev_join, 0));
    HIP_TRY(hipEventRecord(t->ev[0], t->stream));
    if (opt->drain_pits) PYDEM_TRY(stage_pits(t, opt));
    HIP_TRY(hipEventRecord(t->ev[2], t->stream));
    HIP_TRY(hipStreamWaitEvent(t->stream, t->ev_join, 0));
    if (t->pits.n_edges > 0) {
        hipLaunchKernelGGL(k_graph_add_pits, dim3(grid_for(t->pits.n_edges, 1024)), dim3(256), 0, t->stream, t->pits.src,
                           t->pits.dst, t->pits.w, t->pits.n_edges, n, m, (uint32_t *)t->indeg, corner_sums);
    }
    hipLaunchKernelGGL(k_corner_todo, dim3(1), dim3(64), 0, t->stream, corner_sums, t->elev, n, m, t->edge_todo, t->todo_work);
    HIP_TRY(hipEventRecord(t->ev[3], t->stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventSynchronize(t->ev[3]));
    float a = 0, b = 0;
    HIP_TRY(hipEventElapsedTime(&a, t->ev[0], t->ev[2]));      // the pit search, with both graph kernels beside it
    HIP_TRY(hipEventElapsedTime(&b, t->ev[2], t->ev[3]));      // what is left of them after it + pit flags + corners
    (void)a;
    t->tm.graph_ms = b;
    return 0;
}

static void fill_sweep_args(pydem_tile *t, SweepArgs &A)
{
    A.cinfo = (uint32_t *)t->indeg; A.prop = t->prop; A.a0 = t->row_area; A.area = t->uca;
    A.contrib = (double2 *)t->contrib; A.todo_work = t->todo_work; A.n = (int)t->n; A.m = (int)t->m;
    A.pit_src = t->pits.src; A.pit_dst = t->pits.dst; A.n_pit = t->pits.n_edges;
    A.pin_dst = t->pits.in_dst; A.pin_src = t->pits.in_src; A.pin_w = t->pits.in_w;
    A.qcap = (int32_t)(t->NN / 2 < INT32_MAX ? t->NN / 2 : INT32_MAX);      // queue buffers hold NN ints = NN/2 entries
    A.err = t->counters + 15;
    A.tile_open = nullptr;           // (stage_sweep points it at its scratch)
    { const char *e = getenv("PYDEM_TILE_DEBUG"); A.dbg = e ? atoi(e) : 0; }
}

// The re-seed loop of the reference (dem_processing.py:951-964 around cyutils._drain_area, cyutils.pyx:119-187) over the
// unfinished cells on the HOST: the same rules as k_reseed_replay, line by line, with the frontiers kept as sorted lists
// instead of flag scans (a round costs its frontier, not the whole list).  H is sorted by cell.  Pit lists: out-edges
// (src-sorted: pit_src / pit_dst / pit_w), in-edges (dst-sorted: pin_dst / pin_src).
static int64_t reseed_replay_host(std::vector<ReseedHost> &H, std::vector<uint8_t> &done, int n, int m, int maxcount, bool tile_has_nan,
                                  const std::vector<int32_t> &pit_src, const std::vector<int32_t> &pit_dst, const std::vector<double> &pit_w,
                                  const std::vector<int32_t> &pin_dst, const std::vector<int32_t> &pin_src)
{
    const int32_t nU = (int32_t)H.size();
    const int64_t n_pit = (int64_t)pit_src.size();
    static const int DI[8] = {-1, -1, -1, 0, 0, 1, 1, 1}, DJ[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
    auto find = [&](int32_t c) -> int32_t {
        int32_t lo = 0, hi = nU - 1;
        while (lo <= hi) { const int32_t mid = (lo + hi) >> 1; if (H[(size_t)mid].c == c) return mid; if (H[(size_t)mid].c < c) lo = mid + 1; else hi = mid - 1; }
        return -1;
    };
    auto on_edge = [&](int32_t c) { const int i = c / m, j = c - i * m; return i == 0 || i == n - 1 || j == 0 || j == m - 1; };
    auto is_done = [&](int32_t c) -> bool { const int32_t k = find(c); return k < 0 || done[(size_t)k]; };   // not listed = finished by the passes
    done.assign((size_t)nU, 0);
    std::vector<uint8_t> in_cur((size_t)nU, 0);
    std::vector<int32_t> cur, prev;
    int64_t n_done = 0, done_prev = -1;
    for (int count = 2; n_done < nU && count < maxcount && n_done != done_prev; count++) {          // :951-952
        done_prev = n_done;
        double mx = 0.0;                                                                            // :962-964
        for (int32_t k = 0; k < nU; k++) if (!done[(size_t)k] && H[(size_t)k].elev > mx) mx = H[(size_t)k].elev;
        if (tile_has_nan) mx = NAN;
        cur.clear();
        for (int32_t k = 0; k < nU; k++) {
            const double v = done[(size_t)k] ? 0.0 : H[(size_t)k].elev;
            if ((v - mx) / mx > -0.01) cur.push_back(k);
        }
        for (int64_t guard = 0; guard < 8 * (int64_t)nU + 64; guard++) {
            for (int32_t k : cur) { if (!done[(size_t)k]) n_done++; done[(size_t)k] = 1; }          // cyutils.pyx:138-140
            prev.swap(cur); cur.clear();
            for (int32_t k : prev) {                                                                // ascending cell order
                const ReseedHost &S = H[(size_t)k];
                const int32_t i = S.c;
                int32_t tg[2]; double fc[2]; int nt = 0;
                if (S.cw & (CI_OUT1 | CI_OUT2)) {
                    const int sct = (int)((S.cw >> CI_SEC_SHIFT) & 7u);
                    if (S.cw & CI_OUT1) { tg[nt] = i + fe1r(sct) * m + fe1c(sct); fc[nt] = S.prop; nt++; }
                    if (S.cw & CI_OUT2) { tg[nt] = i + fe2r(sct) * m + fe2c(sct); fc[nt] = 1 - S.prop; nt++; }
                    if (nt == 2 && tg[1] < tg[0]) { std::swap(tg[0], tg[1]); std::swap(fc[0], fc[1]); }
                }
                int64_t e = (S.cw & CI_PIT_OUT) ? S.pout_first : 0;
                for (int q = 0;; q++) {
                    int32_t row; double factor;
                    if (S.cw & CI_PIT_OUT) { if (!(e < n_pit && pit_src[(size_t)e] == i)) break; row = pit_dst[(size_t)e]; factor = pit_w[(size_t)e]; e++; }
                    else { if (q >= nt) break; row = tg[q]; factor = fc[q]; }
                    const int32_t kr = find(row);
                    if (kr < 0) continue;
                    if (done[(size_t)kr] && on_edge(row)) continue;                                  // :159-161
                    H[(size_t)kr].area += S.area * factor;                                          // :163
                    if (S.td) H[(size_t)kr].td = 1;
                    bool wait = false;                                                              // :173-179
                    const uint32_t cwr = H[(size_t)kr].cw;
                    for (int d = 0; d < 8 && !wait; d++)
                        if ((cwr & (1u << d)) && !is_done(row + DI[d] * m + DJ[d])) wait = true;
                    if (!wait && (cwr & CI_PIT_IN))
                        for (int64_t e2 = H[(size_t)kr].pin_first; e2 < n_pit && pin_dst[(size_t)e2] == row; e2++)
                            if (!is_done(pin_src[(size_t)e2])) { wait = true; break; }
                    if (!wait && !in_cur[(size_t)kr]) { in_cur[(size_t)kr] = 1; cur.push_back(kr); }
                }
            }
            std::sort(cur.begin(), cur.end());
            for (int32_t k : cur) in_cur[(size_t)k] = 0;
            if (cur == prev) break;                                                                 // :187 (the frontier did not change)
        }
        cur.clear();
    }
    return n_done;
}

int stage_sweep(pydem_tile *t, const pydem_options *opt)
{
    const int n = (int)t->n, m = (int)t->m;
    // (the tile visits address a tile + halo by 32-bit byte offsets from a scalar base: 34 rows of 16-byte entries)
    if (m >= (1 << 22)) { pydem_set_error("uca: tiles wider than 4 194 303 columns are not supported (this one has %d)", m); return -2; }
    PYDEM_TRY(tile_alloc(t, &t->queue[0], (size_t)t->NN));
    PYDEM_TRY(tile_alloc(t, &t->queue[1], (size_t)t->NN));
    PYDEM_TRY(tile_alloc(t, &t->row_area, (size_t)t->n));
    PYDEM_TRY(tile_alloc(t, &t->contrib, (size_t)t->NN * 2));
    t->circular_cells = 0;
    int32_t *total = t->counters + 3;   // cells processed so far
#ifdef PYDEM_SWEEP_QUEUE
    int32_t *cnt3 = t->counters;        // [0..2] rotating frontier sizes
    int32_t *nsrc = t->counters + 4;    // source cells (round 0)
    int64_t done_prev = 0;
#endif
    HIP_TRY(hipEventRecord(t->ev[0], t->stream));
    HIP_TRY(hipMemsetAsync(t->counters, 0, 16 * sizeof(int32_t), t->stream));
    hipLaunchKernelGGL(k_row_area, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, t->stream, t->dX2, t->dY2, n, t->row_area);
    SweepArgs A;
    fill_sweep_args(t, A);
    if (A.n_pit > 0)   // the unused area slots of pit sources / drains carry their edge-list offsets until they are processed
        hipLaunchKernelGGL(k_pit_stash, dim3(grid_for(A.n_pit, 2048)), dim3(256), 0, t->stream, A.pin_dst, A.pit_src, A.n_pit, t->uca);
    // ---- tile-local passes until they stop paying, then the queue rounds take over
    const int tiles_x = (int)cdiv(m, TT), tiles_total = tiles_x * (int)cdiv(n, TH);
    // scratch: tile_done bytes | per-tile "listed for pass" stamps | two tile lists | open cells per tile
    const size_t tiles_pad = ((size_t)tiles_total + 255) & ~(size_t)255;
    const size_t scratch_need = tiles_pad * (1 + 4 + 4 + 4 + 4 + 4) + 64 * sizeof(int32_t);     // ... | block offsets of the symbolic tiles | K5f counters
    if (t->scratch_bytes < scratch_need) {
        if (t->scratch) { HIP_TRY(hipFree(t->scratch)); t->device_bytes -= (int64_t)t->scratch_bytes; }
        HIP_TRY(dev_malloc((void **)&t->scratch, scratch_need));
        t->scratch_bytes = scratch_need; t->device_bytes += (int64_t)scratch_need;
    }
    uint8_t *tile_done = (uint8_t *)t->scratch;
    int32_t *tile_flag = (int32_t *)(tile_done + tiles_pad);
    int32_t *tile_list[2] = {tile_flag + tiles_pad, tile_flag + 2 * tiles_pad};
    int32_t *cntT = t->counters + 56;          // rotating tile-list sizes
    A.tile_open = tile_flag + 3 * tiles_pad;
    // K5f (two-level solve, uca_sym.inl): at the first look of the host that finds at most PYDEM_SWEEP_SYM tiles listed (0: never) ONE
    // symbolic visit per unfinished tile replaces the repeated numeric visits of the later passes.  The pool of coefficients lives in the
    // two queue buffers (idle in this schedule; the re-seed replay takes them over after k_sym_finish).
    static int64_t sym_switch = -1;
    if (sym_switch < 0) { const char *e = getenv("PYDEM_SWEEP_SYM"); sym_switch = e ? atoll(e) : 4096; if (sym_switch < 0) sym_switch = 0; }
    SymArgs Y;
    memset(&Y, 0, sizeof(Y));
    int32_t *symc = tile_flag + 4 * tiles_pad + tiles_pad;      // 64 counters: [0..15] pool regions, [16..23] / [24..31] work counters of the two launches, [32..] statistics
    const bool sym_on = sym_switch > 0 && TH == 32 && t->NN / 2 < ((int64_t)1 << 31);
    if (sym_on) {
        Y.pool0 = (double *)t->queue[0]; Y.pool1 = (double *)t->queue[1];
        Y.nreg = t->NN >= ((int64_t)1 << 22) ? 16 : 2;
        Y.reg_cap = (uint32_t)((t->NN / 2) / (Y.nreg / 2));
        Y.ctr = symc; Y.tile_sym = (uint32_t *)(tile_flag + 4 * tiles_pad); Y.stat = symc + 32;
        HIP_TRY(hipMemsetAsync(Y.tile_sym, 0xff, tiles_pad * 4, t->stream));
        HIP_TRY(hipMemsetAsync(symc, 0, 64 * sizeof(int32_t), t->stream));
    }
    HIP_TRY(hipMemsetAsync(tile_done, 0, tiles_pad * 5, t->stream));      // done bytes + stamps
    HIP_TRY(hipMemsetAsync(A.tile_open, 0x7f, tiles_pad * 4, t->stream));
    HIP_TRY(hipMemsetAsync(cntT, 0, 4 * sizeof(int32_t), t->stream));
    if (A.dbg & 4) HIP_TRY(hipMemsetAsync(t->counters + 32, 0, 24 * sizeof(int32_t), t->stream));
    int64_t launches = 0;
    uint32_t pass = 0;
    int32_t *work16 = t->counters + 16, *work3 = t->counters + 16;      // band counters of the two full passes / rotating counters of the listed ones
    HIP_TRY(hipMemsetAsync(work16, 0, 16 * sizeof(int32_t), t->stream));
    static int lds_pad = -1;        // occupancy experiments only: extra dynamic LDS per workgroup (PYDEM_TILE_LDS_PAD)
    if (lds_pad < 0) { const char *e = getenv("PYDEM_TILE_LDS_PAD"); lds_pad = e ? atoi(e) : 0; }
    // listed tile passes from pass p on (the list of pass p is in tile_list[p % 2] / cntT[p % 3]); returns the next pass number
    static int res_switch = -1;     // listed tiles at or below which the passes use resident visits (K5e)
    if (res_switch < 0) { const char *e = getenv("PYDEM_SWEEP_RESIDENT"); res_switch = e ? atoi(e) : 4096; }
    int64_t listed_left = 0;        // tiles listed for the pass run_listed returned (> 0 only when it stopped at `stop_at`)
    auto run_listed = [&](int p, int64_t ntiles, int64_t stop_at) -> int {
        TileNext N;
        N.flag = tile_flag;
        listed_left = ntiles;
        while (ntiles > stop_at) {
            static int dbg_each = -1;                        // PYDEM_SWEEP_DEBUG=2: a look (and a line) after every pass
            if (dbg_each < 0) { const char *e = getenv("PYDEM_SWEEP_DEBUG"); dbg_each = (e && atoi(e) >= 2) ? 1 : 0; }
            const int batch = dbg_each ? 1 : ntiles < 2048 ? 16 : 8;    // passes between two looks at the list size (a look idles the GPU for ~30 us; the grid only shrinks below 8192 listed tiles)
#ifndef PYDEM_LISTED_DYNAMIC
            const int grid = (int)(ntiles < 8192 ? (ntiles > 64 ? ntiles : 64) : 8192) * (4 / LWPB);
#else
            const int64_t want = cdiv(ntiles, 4);            // persistent wavefronts: no more workgroups than the chip holds at once
            const int grid = (int)(want < 256 * PYDEM_LISTED_OCC ? (want > 16 ? want : 16) : 256 * PYDEM_LISTED_OCC);
#endif
            const bool resident = ntiles <= res_switch;
            for (int b = 0; b < batch; b++, p++) {
                N.list = tile_list[(p + 1) % 2]; N.count = &cntT[(p + 1) % 3];
                if (resident)
                    hipLaunchKernelGGL(k_sweep_tiles_resident, dim3((unsigned)(ntiles > 256 ? ntiles : 256)), dim3(64), 0, t->stream, A, (uint32_t)p, tiles_x,
                                       (const int32_t *)tile_list[p % 2], (const int32_t *)&cntT[p % 3], tile_done, total, N,
                                       &cntT[(p + 2) % 3], Y);
                else
                    hipLaunchKernelGGL(k_sweep_tiles_listed, dim3(grid), dim3(64 * LWPB), (size_t)lds_pad, t->stream, A, (uint32_t)p, tiles_x,
                                       (const int32_t *)tile_list[p % 2], (const int32_t *)&cntT[p % 3], tile_done, total, N,
                                       &cntT[(p + 2) % 3], work3, Y);
                launches++;
            }
            if (hipMemcpyAsync(t->h_counters, t->counters, 64 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream) != hipSuccess) return -1;
            if (hipStreamSynchronize(t->stream) != hipSuccess) return -1;
            ntiles = t->h_counters[56 + p % 3];
            listed_left = ntiles;
            if (getenv("PYDEM_SWEEP_DEBUG")) fprintf(stderr, "listed tile pass %d: %lld tiles listed next, processed %d\n", p, (long long)ntiles, t->h_counters[3]);
            if (p > (int)CI_LEVEL_INF - 256) return -2;
        }
        return p;
    };
    // ---- circular drainage: replay of the reference's re-seed loop over the unfinished cells (K5c), after either schedule
    auto replay_unfinished = [&](uint32_t pass) -> int {
        HIP_TRY(hipMemcpyAsync(t->h_counters, t->counters, 64 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        if ((int64_t)t->h_counters[3] >= t->NN) return 0;
        {
            // (the counter says how many cells the schedule processed; what is unfinished is decided by the level stamps:
            // the queue schedule does not count every cell it settles)
            ReseedCell *U = (ReseedCell *)t->queue[0];                       // scratch: the queue buffers are idle by now
            const int64_t cap64 = std::min<int64_t>(t->NN * 4 / (int64_t)sizeof(ReseedCell), (int64_t)1 << 22);
            uint8_t *stf = (uint8_t *)t->queue[1];                           // state bytes, then taint bytes
            int32_t *rc = t->counters + 60;                                  // [60] collected, [61] NaN flag, [62] finished by the replay
            HIP_TRY(hipMemsetAsync(rc, 0, 3 * sizeof(int32_t), t->stream));
            hipLaunchKernelGGL(k_reseed_collect, dim3(grid_for(t->NN, 4096)), dim3(256), 0, t->stream, A, U, rc, (int32_t)cap64,
                               (const double *)t->elev, rc + 1);
            HIP_TRY(hipMemcpyAsync(t->h_counters, t->counters, 64 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
            HIP_TRY(hipStreamSynchronize(t->stream));
            const int64_t unfinished = (int64_t)t->h_counters[60];
            if (unfinished == 0) {
                // every cell carries a level stamp although the schedule did not process all of them: seen with the queue
                // schedule on tiles with circular drainage (a queued cell of a loop is never processed).  Not silently.
                pydem_set_error("sweep: %lld cells were scheduled but never processed (PYDEM_SWEEP_MODE=queue does not support this tile's "
                                "circular drainage; use the default schedule)", (long long)(t->NN - (int64_t)t->h_counters[3]));
                return -5;
            }
            t->circular_cells = unfinished;
            static int64_t host_above = -1;        // PYDEM_RESEED_HOST_ABOVE (tests): unfinished cells above which the host replays
            if (host_above < 0) { const char *e = getenv("PYDEM_RESEED_HOST_ABOVE"); host_above = e ? atoll(e) : ((int64_t)1 << 22); }
            if (unfinished > cap64 || unfinished > host_above) {
                // more cells than the one-thread replay is meant for (a loop at the head of a long river): the reference finishes
                // such a tile, slowly (:951-964) -- so does the host here, with a note on stderr
                fprintf(stderr, "pydem: circular drainage with %lld unfinished cells: the re-seed loop runs on the host\n", (long long)unfinished);
                if (unfinished > INT32_MAX / 2) { pydem_set_error("circular drainage: %lld unfinished cells", (long long)unfinished); return -5; }
                const int32_t nU = (int32_t)unfinished;
                void *d_tmp = nullptr;
                HIP_TRY(dev_malloc((void **)&d_tmp, (size_t)nU * (sizeof(ReseedCell) + sizeof(ReseedHost))));
                ReseedCell *U2 = (ReseedCell *)d_tmp;
                ReseedHost *dH = (ReseedHost *)((char *)d_tmp + (size_t)nU * sizeof(ReseedCell));
                HIP_TRY(hipMemsetAsync(rc, 0, 3 * sizeof(int32_t), t->stream));
                hipLaunchKernelGGL(k_reseed_collect, dim3(grid_for(t->NN, 4096)), dim3(256), 0, t->stream, A, U2, rc, nU, (const double *)t->elev, rc + 1);
                hipLaunchKernelGGL(k_reseed_gather, dim3(grid_for(nU, 1024)), dim3(256), 0, t->stream, A, (const ReseedCell *)U2, nU, (const double *)t->elev, dH);
                std::vector<ReseedHost> H((size_t)nU);
                HIP_TRY(hipMemcpyAsync(H.data(), dH, (size_t)nU * sizeof(ReseedHost), hipMemcpyDeviceToHost, t->stream));
                HIP_TRY(hipMemcpyAsync(t->h_counters, t->counters, 64 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
                HIP_TRY(hipStreamSynchronize(t->stream));
                const bool has_nan = t->h_counters[61] != 0;
                std::sort(H.begin(), H.end(), [](const ReseedHost &a, const ReseedHost &b) { return a.c < b.c; });
                const size_t np = (size_t)A.n_pit;
                std::vector<int32_t> h_ps(np), h_pd(np), h_is(np), h_id(np); std::vector<double> h_pw(np);
                if (np) {
                    HIP_TRY(hipMemcpy(h_ps.data(), A.pit_src, np * 4, hipMemcpyDeviceToHost));
                    HIP_TRY(hipMemcpy(h_pd.data(), A.pit_dst, np * 4, hipMemcpyDeviceToHost));
                    HIP_TRY(hipMemcpy(h_pw.data(), t->pits.w, np * 8, hipMemcpyDeviceToHost));
                    HIP_TRY(hipMemcpy(h_id.data(), A.pin_dst, np * 4, hipMemcpyDeviceToHost));
                    HIP_TRY(hipMemcpy(h_is.data(), A.pin_src, np * 4, hipMemcpyDeviceToHost));
                }
                std::vector<uint8_t> dn;
                const int64_t n_done = reseed_replay_host(H, dn, n, m, (int)opt->circular_ref_maxcount, has_nan, h_ps, h_pd, h_pw, h_id, h_is);
                std::vector<ReseedBack> back((size_t)nU);
                for (int32_t k = 0; k < nU; k++) { back[(size_t)k].c = H[(size_t)k].c; back[(size_t)k].area = H[(size_t)k].area; back[(size_t)k].flags = (dn[(size_t)k] ? 1 : 0) | (H[(size_t)k].td ? 2 : 0); }
                static_assert(sizeof(ReseedBack) <= sizeof(ReseedHost), "the gather buffer is reused for the results");
                HIP_TRY(hipMemcpy(dH, back.data(), (size_t)nU * sizeof(ReseedBack), hipMemcpyHostToDevice));
                hipLaunchKernelGGL(k_reseed_scatter, dim3(grid_for(nU, 1024)), dim3(256), 0, t->stream, A, (const ReseedBack *)dH, nU, pass);
                HIP_TRY(hipStreamSynchronize(t->stream));
                HIP_TRY(hipFree(d_tmp));
                t->h_counters[3] += (int32_t)n_done;
                HIP_TRY(hipMemcpy(t->counters + 3, t->h_counters + 3, sizeof(int32_t), hipMemcpyHostToDevice));
                if (getenv("PYDEM_SWEEP_DEBUG")) fprintf(stderr, "circular drainage: %lld unfinished cells, %lld finished by the re-seed replay on the host\n", (long long)unfinished, (long long)n_done);
                return 0;
            }
            std::vector<ReseedCell> hu((size_t)unfinished);
            HIP_TRY(hipMemcpyAsync(hu.data(), U, hu.size() * sizeof(ReseedCell), hipMemcpyDeviceToHost, t->stream));
            HIP_TRY(hipStreamSynchronize(t->stream));
            std::sort(hu.begin(), hu.end(), [](const ReseedCell &a, const ReseedCell &b) { return a.c < b.c; });
            HIP_TRY(hipMemcpyAsync(U, hu.data(), hu.size() * sizeof(ReseedCell), hipMemcpyHostToDevice, t->stream));
            hipLaunchKernelGGL(k_reseed_replay, dim3(1), dim3(64), 0, t->stream, A, (const ReseedCell *)U, (int32_t)unfinished,
                               (const double *)t->elev, (const double *)t->pits.w, stf, stf + unfinished, t->h_counters[61],
                               (int)opt->circular_ref_maxcount, pass, total, rc + 2);
            launches += 2;
            HIP_TRY(hipMemcpyAsync(t->h_counters, t->counters, 64 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
            HIP_TRY(hipStreamSynchronize(t->stream));
            if (getenv("PYDEM_SWEEP_DEBUG")) fprintf(stderr, "circular drainage: %lld unfinished cells, %d finished by the re-seed replay\n", (long long)unfinished, t->h_counters[62]);
        }
        return 0;
    };
    // 0: tile passes only (the product schedule), 1: tile pass + queue rounds + listed tail.  The queue schedule refuses
    // some valid inputs (circular drainage it cannot finish, frontiers above NN/2 entries): it only exists in builds with
    // -DPYDEM_SWEEP_QUEUE, a product build ignores PYDEM_SWEEP_MODE
#ifdef PYDEM_SWEEP_QUEUE
    static int sweep_mode = -1;
    if (sweep_mode < 0) { const char *e = getenv("PYDEM_SWEEP_MODE"); sweep_mode = (e && !strcmp(e, "queue")) ? 1 : 0; }
#else
    constexpr int sweep_mode = 0;
#endif
#ifdef PYDEM_SWEEP_STATIC
    const unsigned full_grid = (unsigned)(((tiles_total + 31) / 32) * 8);
#else
    const unsigned full_grid = (unsigned)std::min<int64_t>(((tiles_total + 31) / 32) * 8, 256 * (32 / FWPB));    // persistent: 32 wavefronts per CU
#endif

    if (sweep_mode == 0) {
        // pass 1 over every tile, pass 2 over every tile that is not done (it also lists the tiles of pass 3),
        // then only the listed tiles until no tile is listed any more
        // PYDEM_SWEEP_FIRST=lds: pass 1 by the LDS-resident kernel (K5a).  Measured at 16384^2: 9.9 ms against 7.8 ms of the
        // generic kernel (eight 20-KB tiles per CU instead of 32 bookkeeping-only ones: the dependency depth of a tile times
        // the LDS latency per cell is not hidden any more), and pass 2 inherits the cells with pit edges -- not the default
        static int first_kind = -1;
        if (first_kind < 0) { const char *e = getenv("PYDEM_SWEEP_FIRST"); first_kind = (e && !strcmp(e, "lds")) ? 1 : 0; }
        // PYDEM_SWEEP_DENSE=k: k dense level kernels (K5d) first; the tile passes then start at pass k + 1.  Measured, not
        // the default (profiles/r04_sweep_passes_dense*.txt)
        static int dense_levels = -1;
        if (dense_levels < 0) { const char *e = getenv("PYDEM_SWEEP_DENSE"); dense_levels = e ? atoi(e) : 0; if (dense_levels < 0) dense_levels = 0; }
        for (int lv = 1; lv <= dense_levels; lv++) {
            hipLaunchKernelGGL(k_sweep_dense_level, dim3((unsigned)cdiv(m, 256), (unsigned)cdiv(n, DENSE_BAND)), dim3(256), 0, t->stream,
                               A, (uint32_t)lv, total);
            launches++;
        }
        const uint32_t pb = (uint32_t)dense_levels;       // passes so far
        if (first_kind == 1 && TH == 32 && pb == 0)
            hipLaunchKernelGGL(k_sweep_first, dim3((unsigned)(((tiles_total + 7) / 8) * 8)), dim3(256), 0, t->stream, A, tiles_x, tiles_total, tile_done, total);
        else {
            TileNext N0; N0.flag = nullptr; N0.list = nullptr; N0.count = nullptr;
            hipLaunchKernelGGL(k_sweep_tiles<false>, dim3(full_grid), dim3(64 * FWPB), (size_t)lds_pad, t->stream, A, pb + 1u, tiles_x, tiles_total, tile_done, total, N0, work16);
        }
        TileNext N; N.flag = tile_flag; N.list = tile_list[(pb + 3) % 2]; N.count = &cntT[(pb + 3) % 3];
        hipLaunchKernelGGL(k_sweep_tiles<true>, dim3(full_grid), dim3(64 * FWPB), (size_t)lds_pad, t->stream, A, pb + 2u, tiles_x, tiles_total, tile_done, total, N, work16 + 8);
        HIP_TRY(hipMemsetAsync(work3, 0, 24 * sizeof(int32_t), t->stream));       // (the listed passes reuse the band counters' words: 3 x 8, rotating)
        launches += 2;
        HIP_TRY(hipMemcpyAsync(t->h_counters, t->counters, 64 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        if (getenv("PYDEM_SWEEP_DEBUG")) fprintf(stderr, "tile passes %u-%u: %d cells of %lld, %d tiles listed\n", pb + 1, pb + 2, t->h_counters[3], (long long)t->NN, t->h_counters[56 + (pb + 3) % 3]);
        int p_end = run_listed((int)pb + 3, t->h_counters[56 + (pb + 3) % 3], sym_on ? sym_switch : 0);
        if (sym_on && p_end > 0 && listed_left > 0) {
            // pass p_end = the symbolic visit of every tile that is not done (K5f (a)): a superset of the tiles listed for it
            const uint32_t ps = (uint32_t)p_end;
            TileNext N3; N3.flag = tile_flag; N3.list = tile_list[(ps + 1) % 2]; N3.count = &cntT[(ps + 1) % 3];
            HIP_TRY(hipMemsetAsync(&cntT[(ps + 2) % 3], 0, sizeof(int32_t), t->stream));      // (a numeric pass clears the count of the list after the next one)
            int32_t *cand = tile_list[ps % 2];      // (the list of this pass is not used: the candidates take its place)
            hipLaunchKernelGGL(k_sym_candidates, dim3((unsigned)cdiv(tiles_total, 256)), dim3(256), 0, t->stream, (const uint8_t *)tile_done, (const int32_t *)A.tile_open,
                               tiles_total, 256, cand, symc + 16);
            // the three size classes side by side (each kernel lasts as long as its longest visits: 0.7 + 0.4 + 0.4 ms one after the other)
            HIP_TRY(hipEventRecord(t->ev_fork, t->stream));
            HIP_TRY(hipStreamWaitEvent(t->stream2, t->ev_fork, 0));
            hipLaunchKernelGGL(k_sweep_sym<256>, dim3(256 * 32), dim3(64), 0, t->stream, A, Y, ps, tiles_x, tiles_total, tile_done, total, N3, (const int32_t *)cand, (const int32_t *)(symc + 16), 0, 0);
            hipLaunchKernelGGL(k_sweep_sym<1024>, dim3(256 * 16), dim3(64), 0, t->stream2, A, Y, ps, tiles_x, tiles_total, tile_done, total, N3, (const int32_t *)cand, (const int32_t *)(symc + 17), 1, 256);
            HIP_TRY(hipEventRecord(t->ev_join, t->stream2));
            HIP_TRY(hipStreamWaitEvent(t->stream, t->ev_join, 0));
            launches += 4;
            HIP_TRY(hipMemcpyAsync(t->h_counters, t->counters, 64 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
            HIP_TRY(hipStreamSynchronize(t->stream));
            if (getenv("PYDEM_SWEEP_DEBUG")) fprintf(stderr, "symbolic pass %u: %d cells of %lld, %d tiles listed\n", ps, t->h_counters[3], (long long)t->NN, t->h_counters[56 + (ps + 1) % 3]);
            p_end = run_listed((int)ps + 1, t->h_counters[56 + (ps + 1) % 3], 0);
        }
        if (p_end == -1) { pydem_set_error("HIP error in the listed tile passes"); return -4; }
        if (p_end == -2) { pydem_set_error("flow paths longer than %u passes are not supported", CI_LEVEL_INF); return -5; }
        pass = (uint32_t)p_end;
        t->tm.sweep_tile_passes = (int64_t)pass;
        if (sym_on) {
            // K5f (c): the cells that still carry a symbolic header
            hipLaunchKernelGGL(k_sym_finish, dim3((unsigned)std::min<int64_t>(tiles_total, 256 * 16)), dim3(64), 0, t->stream, A, Y, pass, tiles_x, tiles_total, total);
            launches++;
            if (getenv("PYDEM_SWEEP_DEBUG")) {
                int32_t hs[64];
                HIP_TRY(hipMemcpyAsync(hs, symc, sizeof(hs), hipMemcpyDeviceToHost, t->stream));
                HIP_TRY(hipStreamSynchronize(t->stream));
                int64_t used = 0;
                for (int r = 0; r < Y.nreg; r++) used += hs[r];
                fprintf(stderr, "two-level solve: %d symbolic tiles (%d fell back to numeric visits), %d outlets, %d symbolic cells, pool %lld of %lld doubles\n",
                        hs[32], hs[33], hs[34], hs[35], (long long)used, (long long)Y.reg_cap * Y.nreg);
#ifdef PYDEM_SYM_PROF
                {
                    const unsigned long long *acc = (const unsigned long long *)(hs + 40);
                    const double v = acc[8] ? (double)acc[8] : 1.0;
                    fprintf(stderr, "symbolic visits: %llu, %.1f open cells, %.1f rounds, %.1f pool entries each; us per visit: stage %.1f, inlets %.1f, set-up %.1f, constants %.1f, "
                            "rounds %.1f (%.2f us per round), end %.1f\n", acc[8], acc[9] / v, acc[6] / v, acc[7] / v, acc[0] / v / 100, acc[1] / v / 100, acc[2] / v / 100,
                            acc[3] / v / 100, acc[4] / v / 100, acc[6] ? acc[4] / 100.0 / (double)acc[6] : 0.0, acc[5] / v / 100);
                }
#endif
            }
        }
        HIP_TRY(hipMemcpyAsync(t->h_counters, t->counters, 64 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        PYDEM_TRY(replay_unfinished(pass));
        if (A.dbg & 4) {
            const unsigned long long *acc = (const unsigned long long *)(t->h_counters + 32);
            fprintf(stderr, "tile phases (10 ns ticks summed over %llu tile runs, %llu of them finished nothing): stage %llu, setup %llu, rounds %llu (%llu rounds), stamp %llu\n",
                    acc[5], acc[6], acc[0], acc[1], acc[2], acc[4], acc[3]);
        }
    }
#ifdef PYDEM_SWEEP_QUEUE
    if (sweep_mode == 1) {
    // every pass re-stages all tiles that still have an open cell (rivers cross most tiles), so after the
    // first pass (which finishes ~3/4 of the grid) the queue rounds are cheaper than another pass
    static int max_passes = -1;
    if (max_passes < 0) { const char *e = getenv("PYDEM_TILE_PASSES"); max_passes = e ? atoi(e) : 1; if (max_passes < 1) max_passes = 1; }
    for (;;) {
        pass++;
        { TileNext N0; N0.flag = nullptr; N0.list = nullptr; N0.count = nullptr;
          HIP_TRY(hipMemsetAsync(work16, 0, 16 * sizeof(int32_t), t->stream));
          hipLaunchKernelGGL(k_sweep_tiles<false>, dim3(full_grid), dim3(256), 0, t->stream, A, pass, tiles_x, tiles_total, tile_done, total, N0, work16); }
        launches++;
        HIP_TRY(hipMemcpyAsync(t->h_counters, t->counters, 16 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        const int64_t done_now = t->h_counters[3];
        const int64_t gained = done_now - done_prev;
        if (getenv("PYDEM_SWEEP_DEBUG")) fprintf(stderr, "tile pass %u: +%lld cells (%.2f%%), total %.2f%%\n", pass, (long long)gained, 100.0 * gained / t->NN, 100.0 * done_now / t->NN);
        done_prev = done_now;
        // a pass costs about one streaming read of the unfinished tiles; stop when it finishes < 1.5 % of the grid
        if (done_now >= t->NN || gained * 64 < t->NN || (int)pass >= max_passes) break;
    }
    t->tm.sweep_tile_passes = (int64_t)pass;
    int r = (int)pass + 1;
    int64_t last = 0;
    if (done_prev < t->NN) {
        hipLaunchKernelGGL(k_sweep_rebuild_frontier, dim3(grid_for(t->NN, 4096)), dim3(256), 0, t->stream, A, (uint32_t)r,
                           (QE *)t->queue[r % 2], &cnt3[r % 3]);
        launches++;
        HIP_TRY(hipMemcpyAsync(t->h_counters, t->counters, 16 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        last = t->h_counters[r % 3];
        if (getenv("PYDEM_SWEEP_DEBUG")) fprintf(stderr, "after rebuild: frontier %lld, processed %d of %lld\n", (long long)last, t->h_counters[3], (long long)t->NN);
    }
    (void)nsrc;
    static int small_cap = -1, tile_switch = -1;
    if (small_cap < 0) { const char *e = getenv("PYDEM_SWEEP_SMALL"); small_cap = e ? atoi(e) : SWEEP_SMALL_CAP; }
    if (tile_switch < 0) { const char *e = getenv("PYDEM_SWEEP_TILE_SWITCH"); tile_switch = e ? atoi(e) : 20000; }
    while (last > 0) {
        if (last <= tile_switch) {
            // ---- the rivers: listed tile passes.  A queue round moves every river by ONE cell per kernel
            // boundary; a listed pass moves it through a whole tile (the on-chip rounds) for the same boundary.
            int p = r;
            TileNext N;
            N.flag = tile_flag; N.list = tile_list[p % 2]; N.count = &cntT[p % 3];
            hipLaunchKernelGGL(k_tiles_of_frontier, dim3(grid_for(last, 256)), dim3(256), 0, t->stream, (const QE *)t->queue[r % 2],
                               (const int32_t *)&cnt3[r % 3], m, tiles_x, (int32_t)p, N);
            launches++;
            int64_t ntiles = last;      // upper bound for the first batch
            while (ntiles > 0) {
                const int batch = 8;
                const int grid = (int)(ntiles * 2 < 2048 ? (ntiles * 2 > 64 ? ntiles * 2 : 64) : 2048);
                for (int b = 0; b < batch; b++, p++) {
                    N.list = tile_list[(p + 1) % 2]; N.count = &cntT[(p + 1) % 3];
                    hipLaunchKernelGGL(k_sweep_tiles_listed, dim3(grid), dim3(256), 0, t->stream, A, (uint32_t)p, tiles_x,
                                       (const int32_t *)tile_list[p % 2], (const int32_t *)&cntT[p % 3], tile_done, total, N,
                                       &cntT[(p + 2) % 3], work3, Y);
                    launches++;
                }
                HIP_TRY(hipMemcpyAsync(t->h_counters, t->counters, 64 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
                HIP_TRY(hipStreamSynchronize(t->stream));
                ntiles = t->h_counters[56 + p % 3];
                if (getenv("PYDEM_SWEEP_DEBUG")) fprintf(stderr, "listed tile pass %d: %lld tiles listed next, processed %d\n", p, (long long)ntiles, t->h_counters[3]);
                if (p > (int)CI_LEVEL_INF - 256) { pydem_set_error("flow paths longer than %u passes are not supported", CI_LEVEL_INF); return -5; }
            }
            t->tm.sweep_tile_passes += p - r;
            r = p;
            last = 0;
            break;
        }
        if (last <= small_cap && last <= SWEEP_SMALL_CAP) {
            // one workgroup, many rounds (until the frontier is empty or grows past SWEEP_SMALL_CAP)
            hipLaunchKernelGGL(k_sweep_small, dim3(1), dim3(1024), 0, t->stream, A, (QE *)t->queue[0], (QE *)t->queue[1], cnt3, r,
                               (int)CI_LEVEL_INF - 256, total, t->counters + 14);
            launches++;
            HIP_TRY(hipMemcpyAsync(t->h_counters, t->counters, 16 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
            HIP_TRY(hipStreamSynchronize(t->stream));
            const int r_new = t->h_counters[14];
            if (getenv("PYDEM_SWEEP_DEBUG")) fprintf(stderr, "single-workgroup rounds %d..%d, frontier now %d\n", r, r_new, t->h_counters[r_new % 3]);
            if (r_new == r) { pydem_set_error("sweep made no progress at round %d", r); return -5; }
            r = r_new;
            last = t->h_counters[r % 3];
            if (r >= (int)CI_LEVEL_INF - 256) { pydem_set_error("flow paths longer than %u rounds are not supported", CI_LEVEL_INF); return -5; }
            continue;
        }
        const int batch = last > 262144 ? 2 : (last > 4096 ? 8 : 64);
        const int grid = grid_for(last, 2048);
        const bool lowlat = last <= 262144;       // latency-bound rounds: speculative batched loads
        for (int b = 0; b < batch; b++, r++) {
            if (lowlat)
                hipLaunchKernelGGL(k_sweep_round<true>, dim3(grid), dim3(256), 0, t->stream, A, (const QE *)t->queue[r % 2],
                                   (QE *)t->queue[(r + 1) % 2], cnt3, r, total);
            else
                hipLaunchKernelGGL(k_sweep_round<false>, dim3(grid), dim3(256), 0, t->stream, A, (const QE *)t->queue[r % 2],
                                   (QE *)t->queue[(r + 1) % 2], cnt3, r, total);
            launches++;
        }
        HIP_TRY(hipMemcpyAsync(t->h_counters, t->counters, 16 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        last = t->h_counters[r % 3];
        if (getenv("PYDEM_SWEEP_DEBUG")) fprintf(stderr, "round %d: frontier %lld, processed %d\n", r, (long long)last, t->h_counters[3]);
        if (r > (int)CI_LEVEL_INF - 256) { pydem_set_error("flow paths longer than %u rounds are not supported", CI_LEVEL_INF); return -5; }
    }
        PYDEM_TRY(replay_unfinished((uint32_t)r));
    }   // sweep_mode == 1
#endif  // PYDEM_SWEEP_QUEUE
    if (t->h_counters[15] > 0) { pydem_set_error("sweep frontier exceeded the queue capacity (%lld entries)", (long long)A.qcap); return -5; }
    const int64_t processed = (int64_t)t->h_counters[3];     // tile passes + queue rounds ([4], [10]: tile-pass statistics)
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

static double host_now_ms()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

int stage_edge_update(pydem_tile *t, const pydem_options *opt, const double *const data[4], const uint8_t *const done[4],
                      const uint8_t *const todo[4])
{
    (void)opt;
    if (t->einc_ready) PYDEM_TRY(stage_edge_flush(t));      // incremental rounds left deltas waiting: settle them first
    const double t_begin = host_now_ms();
    const int n = (int)t->n, m = (int)t->m;
    PYDEM_TRY(tile_alloc(t, &t->row_area, (size_t)t->n));
    const int L = n > m ? n : m;
    const int64_t nper = 2 * (int64_t)m + 2 * (int64_t)(n - 2);
    PYDEM_TRY(tile_alloc(t, &t->estamp, (size_t)t->NN));
    PYDEM_TRY(tile_alloc(t, &t->edelta, (size_t)t->NN));
    PYDEM_TRY(tile_alloc(t, &t->labels, (size_t)t->NN));        // rlist
    PYDEM_TRY(tile_alloc(t, &t->flatlist, (size_t)t->NN));      // tlist (kept until the next round restores the mask)
    PYDEM_TRY(tile_alloc(t, &t->queue[0], (size_t)t->NN));
    PYDEM_TRY(tile_alloc(t, &t->queue[1], (size_t)t->NN));
    PYDEM_TRY(tile_alloc(t, &t->eseed, (size_t)(nper > 0 ? nper : 1) * 2));
    PYDEM_TRY(tile_alloc(t, &t->p_delta, (size_t)nper));
    PYDEM_TRY(tile_alloc(t, &t->p_flags, (size_t)nper * 2));
    PYDEM_TRY(tile_alloc(t, &t->s_data, (size_t)L * 4));
    PYDEM_TRY(tile_alloc(t, &t->s_flags, (size_t)L * 8));
    EdgeArgs E;
    SweepArgs &A = E.G;
    fill_sweep_args(t, A);
    E.flag = (uint32_t *)t->estamp; E.delta = t->edelta; E.flats = t->flats; E.edge_done = t->edge_done;
    E.p_done = t->p_flags; E.p_seed = t->p_flags + nper; E.p_delta = t->p_delta;
    E.rlist = t->labels; E.rcount = t->counters + 6;
    E.tlist = t->flatlist; E.tcount = t->counters + 7;
    // int2 per cell = half of a double2 slot: the offsets use the first NN * 8 bytes of the contribution array
    PYDEM_TRY(tile_alloc(t, &t->contrib, (size_t)t->NN * 2));
    E.pit_off = reinterpret_cast<const int2 *>(t->contrib);
    if (!t->edge_clean) {
        if (A.n_pit > 0)
            hipLaunchKernelGGL(k_pit_offsets, dim3(grid_for(A.n_pit, 2048)), dim3(256), 0, t->stream, A.pin_dst, A.pit_src, A.n_pit,
                               reinterpret_cast<int2 *>(t->contrib));
        // first round after the graph was (re)built: flags and counts to zero, masks to their defaults (:812, :817)
        HIP_TRY(hipMemsetAsync(t->estamp, 0, (size_t)t->NN * 4, t->stream));
        hipLaunchKernelGGL(k_edge_clear_levels, dim3(grid_for(t->NN, 8192)), dim3(256), 0, t->stream, A.cinfo, t->NN);
        HIP_TRY(hipMemsetAsync(t->edge_todo, 0, (size_t)t->NN, t->stream));
        HIP_TRY(hipMemsetAsync(t->edge_done, 1, (size_t)t->NN, t->stream));
        t->edge_clean = true;
    } else if (t->etodo_prev > 0) {
        hipLaunchKernelGGL(k_edge_restore, dim3(grid_for(t->etodo_prev, 1024)), dim3(256), 0, t->stream, t->flatlist, t->etodo_prev,
                           t->edge_done);
    }
    // strips -> device (left, right, top, bottom), padded to L entries each (data == NULL: they are there already, written
    // by the edge board)
    if (data) {
        // (pinned staging like the incremental rounds: asynchronous copies from pageable memory make the runtime pin and unpin pages
        // behind the caller's back, and the next GPU call waits for that)
        if (t->h_strip_cap < (size_t)L) {
            if (t->h_strip_d) { (void)hipHostFree(t->h_strip_d); (void)hipHostFree(t->h_strip_f); }
            HIP_TRY(hipHostMalloc((void **)&t->h_strip_d, (size_t)L * 4 * sizeof(double)));
            HIP_TRY(hipHostMalloc((void **)&t->h_strip_f, (size_t)L * 8));
            t->h_strip_cap = (size_t)L;
        }
        double *hd = t->h_strip_d;
        uint8_t *hf = t->h_strip_f;
        memset(hd, 0, (size_t)L * 4 * sizeof(double));
        memset(hf, 0, (size_t)L * 8);
        for (int s = 0; s < 4; s++) {
            const int len = s < 2 ? n : m;
            for (int k = 0; k < len; k++) {
                hd[(size_t)s * L + k] = data[s][k];
                hf[(size_t)s * L + k] = done[s][k] != 0;
                hf[(size_t)(4 + s) * L + k] = todo[s][k] != 0;
            }
        }
        HIP_TRY(hipMemcpyAsync(t->s_data, hd, (size_t)L * 4 * 8, hipMemcpyHostToDevice, t->stream));
        HIP_TRY(hipMemcpyAsync(t->s_flags, hf, (size_t)L * 8, hipMemcpyHostToDevice, t->stream));
    }
    HIP_TRY(hipMemsetAsync(t->counters, 0, 16 * sizeof(int32_t), t->stream));
    int32_t *cnt3 = t->counters;      // rotating frontier sizes; level r reads queue[r % 2] / cnt3[r % 3]
    int32_t *n_seed = t->counters + 8;
    QE *q0 = (QE *)t->queue[0], *q1 = (QE *)t->queue[1];
    hipLaunchKernelGGL(k_edge_init, dim3((unsigned)cdiv(nper, 128)), dim3(128), 0, t->stream, E, t->s_data, t->s_flags,
                       t->s_flags + (size_t)4 * L, L, t->uca, t->edge_todo, q0, &cnt3[0], (QE *)t->eseed, n_seed);
    HIP_TRY(hipMemcpyAsync(t->h_counters, t->counters, 16 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    const int32_t nflood = t->h_counters[0], nseed = t->h_counters[8];
    int dbg_rounds[2] = {0, 0}, dbg_wide[2] = {0, 0};
    auto run_levels = [&](int which, int32_t first) -> int {
        // which: 0 floods, 1 seeded sweep.  The frontier of level 0 is in queue[0] / cnt3[0].
        int r = 0;
        int32_t last = first;
        int32_t *state = t->counters + 12;
        static int small_cap = -1;
        if (small_cap < 0) { const char *e = getenv("PYDEM_EDGE_SMALL_CAP"); small_cap = e ? atoi(e) : SMALL_CAP; if (small_cap > SMALL_CAP) small_cap = SMALL_CAP; }
        while (last > 0) {
            if (last <= small_cap) {
                if (which == 0) hipLaunchKernelGGL(k_edge_small<0>, dim3(1), dim3(1024), 0, t->stream, E, q0, q1, cnt3, r, state);
                else hipLaunchKernelGGL(k_edge_small<1>, dim3(1), dim3(1024), 0, t->stream, E, q0, q1, cnt3, r, state);
                HIP_TRY(hipMemcpyAsync(t->h_counters, t->counters, 16 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
                HIP_TRY(hipStreamSynchronize(t->stream));
                r = t->h_counters[12];
                last = t->h_counters[r % 3];
                dbg_rounds[which] = r;
                continue;
            }
            const int batch = last > 65536 ? 4 : 16;
            const int grid = grid_for(last, 1024);
            dbg_wide[which] += batch;
            for (int b = 0; b < batch; b++, r++) {
                if (which == 0) hipLaunchKernelGGL(k_edge_level<0>, dim3(grid), dim3(256), 0, t->stream, E, (r % 2) ? q1 : q0, (r % 2) ? q0 : q1, cnt3, r);
                else hipLaunchKernelGGL(k_edge_level<1>, dim3(grid), dim3(256), 0, t->stream, E, (r % 2) ? q1 : q0, (r % 2) ? q0 : q1, cnt3, r);
            }
            HIP_TRY(hipMemcpyAsync(t->h_counters, t->counters, 16 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
            HIP_TRY(hipStreamSynchronize(t->stream));
            last = t->h_counters[r % 3];
            dbg_rounds[which] = r;
            if (r > (1 << 24)) { pydem_set_error("edge update: flow paths too long"); return -5; }
        }
        return 0;
    };
    if (nflood > 0) PYDEM_TRY(run_levels(0, nflood));
    if (nseed > 0) {
        int32_t three[3] = {nseed, 0, 0};
        HIP_TRY(hipMemcpyAsync(q0, t->eseed, (size_t)nseed * sizeof(QE), hipMemcpyDeviceToDevice, t->stream));
        HIP_TRY(hipMemcpyAsync(cnt3, three, sizeof(three), hipMemcpyHostToDevice, t->stream));
        PYDEM_TRY(run_levels(1, nseed));
        hipLaunchKernelGGL(k_edge_apply, dim3(grid_for(t->NN < (1 << 20) ? t->NN : (1 << 20), 1024)), dim3(256), 0, t->stream, E, t->uca,
                           E.rcount);
    }
    hipLaunchKernelGGL(k_edge_apply_perimeter, dim3((unsigned)cdiv(nper, 128)), dim3(128), 0, t->stream, E, t->uca);
    hipLaunchKernelGGL(k_edge_cleanup, dim3(256), dim3(256), 0, t->stream, E, (const int32_t *)E.rcount, (const int32_t *)E.tcount);
    HIP_TRY(hipMemcpyAsync(t->h_counters, t->counters, 16 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(t->stream));
    t->etodo_prev = t->h_counters[7];
    if (getenv("PYDEM_EDGE_DEBUG"))
        fprintf(stderr, "edge round: %d seeds, %d todo cells, %d cells reached; levels: floods %d (%d wide), sweep %d (%d wide); %.3f ms\n",
                nseed, t->h_counters[7], t->h_counters[6], dbg_rounds[0], dbg_wide[0], dbg_rounds[1], dbg_wide[1], host_now_ms() - t_begin);
    return 0;
}

// ---- incremental edge rounds: host side ----------------------------------------------------------------
static int einc_args(pydem_tile *t, IncArgs &E)
{
    PYDEM_TRY(tile_alloc(t, &t->row_area, (size_t)t->n));
    PYDEM_TRY(tile_alloc(t, &t->estamp, (size_t)t->NN));
    PYDEM_TRY(tile_alloc(t, &t->edelta, (size_t)t->NN));
    PYDEM_TRY(tile_alloc(t, &t->queue[0], (size_t)t->NN));
    PYDEM_TRY(tile_alloc(t, &t->queue[1], (size_t)t->NN));
    PYDEM_TRY(tile_alloc(t, &t->contrib, (size_t)t->NN * 2));
    fill_sweep_args(t, E.G);
    E.flag = (uint32_t *)t->estamp; E.delta = t->edelta; E.flats = t->flats; E.edge_done = t->edge_done; E.edge_todo = t->edge_todo;
    E.uca = t->uca; E.pit_off = reinterpret_cast<const int2 *>(t->contrib); E.set_done = 1;
    E.prof = t->counters + 40;
    E.nanq = reinterpret_cast<int32_t *>(t->queue[1]); E.n_nan = t->counters + 53;   // (queue 1 is empty until the cascade's first level)
    E.round16 = (uint32_t)(t->einc_round % 65535) + 1u;
    return 0;
}

// run the cascade whose first frontier is in queue[0] / counters[0]; ONE host synchronisation when the frontier stays small
static int einc_cascade(pydem_tile *t, const IncArgs &E, int *levels)
{
    int32_t *cnt3 = t->counters;
    int32_t *state = t->counters + 12;
    QE *q0 = (QE *)t->queue[0], *q1 = (QE *)t->queue[1];
    int r = 0;
    for (;;) {
        static int einc_block = -1;
        if (einc_block < 0) { const char *e = getenv("PYDEM_EINC_BLOCK"); einc_block = e ? atoi(e) : 1024; }
        hipLaunchKernelGGL(k_einc_small, dim3(1), dim3(einc_block), 0, t->stream, E, q0, q1, cnt3, r, state);
        HIP_TRY(hipMemcpyAsync(t->h_counters, t->counters, 16 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        r = t->h_counters[12];
        int32_t last = t->h_counters[r % 3];
        if (last == 0) break;
        // the frontier outgrew one workgroup: level kernels until it is small again
        while (last > SMALL_CAP) {
            const int batch = last > 65536 ? 4 : 16;
            const int grid = grid_for(last, 1024);
            for (int b = 0; b < batch; b++, r++)
                hipLaunchKernelGGL(k_einc_level, dim3(grid), dim3(256), 0, t->stream, E, (r % 2) ? q1 : q0, (r % 2) ? q0 : q1, cnt3, r);
            HIP_TRY(hipMemcpyAsync(t->h_counters, t->counters, 16 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
            HIP_TRY(hipStreamSynchronize(t->stream));
            last = t->h_counters[r % 3];
            if (r > (1 << 24)) { pydem_set_error("edge update: flow paths too long"); return -5; }
        }
        if (last == 0) break;
    }
    if (levels) *levels = r;
    HIP_TRY(hipGetLastError());
    return 0;
}


static int cinc_args(pydem_tile *t, CIncArgs &E)
{
    fill_sweep_args(t, E.G);
    E.rec = (NDRec *)t->nd_rec; E.nd = t->nd; E.cid = t->estamp;
    E.pit_off = reinterpret_cast<const int2 *>(t->contrib);
    E.flats = t->flats; E.edge_done = t->edge_done; E.edge_todo = t->edge_todo; E.uca = t->uca; E.set_done = 1;
    E.prof = t->counters + 40;
    E.nanq = reinterpret_cast<int32_t *>(t->edelta); E.n_nan = t->counters + 53;     // (the delta plane is idle in the compact form)
    E.round16 = (uint32_t)(t->einc_round % 65535) + 1u;
    return 0;
}

static int cinc_cascade(pydem_tile *t, const CIncArgs &E, int *levels)
{
    int32_t *cnt3 = t->counters;
    int32_t *state = t->counters + 12;
    QE *q0 = (QE *)t->queue[0], *q1 = (QE *)t->queue[1];
    int r = 0;
    for (;;) {
        static int cinc_block = -1;
        if (cinc_block < 0) { const char *e = getenv("PYDEM_EINC_BLOCK"); cinc_block = e ? atoi(e) : 1024; }
        // (PYDEM_CINC_SMALL: the frontier width up to which ONE workgroup walks the levels; wider levels are launches over the chip)
        static int cinc_cap = -1;
        if (cinc_cap < 0) { const char *e = getenv("PYDEM_CINC_SMALL"); cinc_cap = e ? std::max(1, std::min(atoi(e), SMALL_CAP)) : 1024; }      // (measured on the flush of 8 x 16384^2: 4096 -> 1024 -0.8 ms per tile, 256 the same)
        hipLaunchKernelGGL(k_cinc_small, dim3(1), dim3(cinc_block), 0, t->stream, E, q0, q1, cnt3, r, state, cinc_cap);
        // (the usual case: the frontier stayed small and the cascade is over -- the records go to the tile right away,
        // ONE host synchronisation per round; cells finished so far are applied either way)
        hipLaunchKernelGGL(k_cinc_apply, dim3(grid_for(E.nd, 1024)), dim3(256), 0, t->stream, E);
        HIP_TRY(hipMemcpyAsync(t->h_counters, t->counters, 16 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        r = t->h_counters[12];
        int32_t last = t->h_counters[r % 3];
        if (last == 0) break;
        bool wide = false;
        while (last > cinc_cap) {
            wide = true;
            const int batch = last > 65536 ? 4 : 16;
            const int grid = grid_for(last, 1024);
            for (int b = 0; b < batch; b++, r++)
                hipLaunchKernelGGL(k_cinc_level, dim3(grid), dim3(256), 0, t->stream, E, (r % 2) ? q1 : q0, (r % 2) ? q0 : q1, cnt3, r);
            HIP_TRY(hipMemcpyAsync(t->h_counters, t->counters, 16 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
            HIP_TRY(hipStreamSynchronize(t->stream));
            last = t->h_counters[r % 3];
            if (r > (1 << 24)) { pydem_set_error("edge update: flow paths too long"); return -5; }
        }
        if (last == 0) {
            if (wide) { hipLaunchKernelGGL(k_cinc_apply, dim3(grid_for(E.nd, 1024)), dim3(256), 0, t->stream, E); HIP_TRY(hipStreamSynchronize(t->stream)); }
            break;
        }
    }
    if (levels) *levels = r;
    HIP_TRY(hipGetLastError());
    return 0;
}

// first incremental round after the graph was (re)built: count the cells that are not done and choose the form
static int einc_prepare(pydem_tile *t, IncArgs &E)
{
    if (E.G.n_pit > 0)
        hipLaunchKernelGGL(k_pit_offsets, dim3(grid_for(E.G.n_pit, 2048)), dim3(256), 0, t->stream, E.G.pin_dst, E.G.pit_src, E.G.n_pit,
                           reinterpret_cast<int2 *>(t->contrib));
    HIP_TRY(hipMemsetAsync(t->estamp, 0, (size_t)t->NN * 4, t->stream));
    unsigned long long *cnt64 = reinterpret_cast<unsigned long long *>(t->counters + 48);
    HIP_TRY(hipMemsetAsync(t->counters + 48, 0, 8 * sizeof(int32_t), t->stream));       // ([53]: NaN seeds of a round)
    hipLaunchKernelGGL(k_nd_count, dim3(grid_for(t->NN, 4096)), dim3(256), 0, t->stream, t->edge_done, t->NN, cnt64);
    HIP_TRY(hipMemcpyAsync(t->h_counters + 48, t->counters + 48, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    const int64_t nd = (int64_t)*reinterpret_cast<unsigned long long *>(t->h_counters + 48);
    int64_t compact_max = ND_COMPACT_MAX;            // (read per fix-up: the tests switch the form)
    { const char *e = getenv("PYDEM_EINC_COMPACT_MAX"); if (e) compact_max = atoll(e); }
    t->einc_compact = nd <= compact_max;
    if (t->einc_compact) {
        if (nd > t->nd_cap) {
            if (t->nd_rec) { HIP_TRY(hipFree(t->nd_rec)); t->device_bytes -= t->nd_cap * (int64_t)sizeof(NDRec); }
            const int64_t cap = nd + nd / 8 + 1024;
            HIP_TRY(dev_malloc((void **)&t->nd_rec, (size_t)cap * sizeof(NDRec)));
            t->nd_cap = cap; t->device_bytes += cap * (int64_t)sizeof(NDRec);
        }
        t->nd = (int32_t)nd;
        CIncArgs C;
        PYDEM_TRY(cinc_args(t, C));
        HIP_TRY(hipMemsetAsync(t->counters + 50, 0, sizeof(int32_t), t->stream));
        if (nd > 0) {
            hipLaunchKernelGGL(k_nd_assign, dim3(grid_for(t->NN, 4096)), dim3(256), 0, t->stream, C, t->NN, t->counters + 50);
            hipLaunchKernelGGL(k_nd_link, dim3(grid_for(nd, 1024)), dim3(256), 0, t->stream, C);
        }
    } else {
        hipLaunchKernelGGL(k_edge_clear_levels, dim3(grid_for(t->NN, 8192)), dim3(256), 0, t->stream, E.G.cinfo, t->NN);
        hipLaunchKernelGGL(k_einc_prepare, dim3(grid_for(t->NN, 8192)), dim3(256), 0, t->stream, E, t->NN);
    }
    HIP_TRY(hipGetLastError());
    t->einc_ready = true;
    t->edge_clean = false;          // the classic rounds find their zeroed state gone
    return 0;
}

// ---- condensed incremental rounds: host side (kernels in uca_cond.inl) ----------------------------------------
void tile_watch_line(pydem_tile *t, int axis, int64_t index)
{
    const int64_t lim = axis == 0 ? t->n : t->m;
    if (index < 0) index += lim;
    if (index == 0 || index == lim - 1) return;                    // perimeter: always watched
    for (const auto &w : t->watch) if (w.first == axis && w.second == index) return;
    t->watch.emplace_back(axis, index);
}

bool tile_line_watched(const pydem_tile *t, int axis, int64_t index)
{
    if (!(t->einc_ready && t->cond_live)) return true;             // nothing is deferred
    const int64_t lim = axis == 0 ? t->n : t->m;
    if (index < 0) index += lim;
    if (index == 0 || index == lim - 1) return true;
    for (size_t k = 0; k < t->watch_built && k < t->watch.size(); k++)
        if (t->watch[k].first == axis && t->watch[k].second == index) return true;
    return false;
}

static int cond_args(pydem_tile *t, CondArgsE &X)
{
    PYDEM_TRY(cinc_args(t, X.C));
    X.node = (CNode *)t->cond_node; X.nw = t->cond_nw; X.edge = (const CEdge *)t->cond_edge; X.slot = t->cond_slot;
    X.q0 = t->cond_q0; X.q1 = t->cond_q1; X.nanq = t->cond_nanq; X.nan_cap = t->cond_nan_cap; X.cnt = t->cond_cnt;
    X.gate = nullptr; X.gate_bit = 0; X.round_base = nullptr; X.round_add = nullptr;
    return 0;
}

// Build the condensed graph of the watched cells from the compact records (just linked by einc_prepare) ON THE HOST: the
// build of rounds 4-5, since round 6 the fall-back and the checker of the device build (cond_build_device below).  Returns 0
// and leaves cond_live false when the tile does not qualify (a cycle among the records, a pathological fan).
static thread_local const char *g_cond_host_gave_up = "";       // why the host build left the tile to the plain cascade (PYDEM_COND_BUILD=check reports it)
static int cond_build_host(pydem_tile *t)
{
    t->cond_live = false; t->cond_pending = false;
    g_cond_host_gave_up = "";
    const double t_begin = host_now_ms();
    const int32_t nd = t->nd;
    const int n = (int)t->n, m = (int)t->m;
    CIncArgs C;
    PYDEM_TRY(cinc_args(t, C));
    // ---- watched records: the perimeter and the lines other tiles read
    auto mark = [&](int axis, int64_t index) {
        const int64_t count = axis == 0 ? m : n;
        hipLaunchKernelGGL(k_cond_mark, dim3((unsigned)std::min<int64_t>(cdiv(count, 256), 64)), dim3(256), 0, t->stream, C, axis, index);
    };
    mark(0, 0); mark(0, n - 1); mark(1, 0); mark(1, m - 1);
    for (const auto &w : t->watch) mark(w.first, w.second);
    // ---- pit -> drain edges between records and the records' graph fields (scratch: the two queue buffers, idle until the
    // first cascade), through pinned staging
    CPitEdge *d_pe = reinterpret_cast<CPitEdge *>(t->queue[1]);
    const int32_t pe_cap = (int32_t)std::min<int64_t>(t->NN / 4, (int64_t)1 << 24);
    int32_t *d_npe = t->counters + 54;
    HIP_TRY(hipMemsetAsync(d_npe, 0, sizeof(int32_t), t->stream));
    if (C.G.n_pit > 0)
        hipLaunchKernelGGL(k_cond_pit_edges, dim3(grid_for(nd, 1024)), dim3(256), 0, t->stream, C, (const double *)t->pits.w, d_pe, d_npe, pe_cap);
    CRecH *d_hr = reinterpret_cast<CRecH *>(t->queue[0]);
    void *d_tmp = nullptr;
    if ((int64_t)nd * (int64_t)sizeof(CRecH) > t->NN * 4) {       // (small tiles that are mostly 'not done': the queue buffer is too short)
        HIP_TRY(dev_malloc((void **)&d_tmp, (size_t)nd * sizeof(CRecH)));
        d_hr = reinterpret_cast<CRecH *>(d_tmp);
    }
    hipLaunchKernelGGL(k_cond_extract, dim3(grid_for(nd, 1024)), dim3(256), 0, t->stream, C, d_hr);
    void *pin_v = nullptr;
    // (pinned staging: the records' extract, and behind it room for the nodes -- at most one per cell of a watched line)
    const size_t hr_bytes = (((size_t)nd * sizeof(CRecH) + 64) + 127) & ~(size_t)127;
    const size_t nw_bound = (size_t)std::min<int64_t>((int64_t)nd, (int64_t)(4 + t->watch.size()) * (int64_t)std::max(n, m));
    PYDEM_TRY(tile_pinned(t, hr_bytes + nw_bound * sizeof(CNode), &pin_v));
    const CRecH *hr = reinterpret_cast<const CRecH *>((char *)pin_v + 64);
    int32_t *h_npe = reinterpret_cast<int32_t *>(pin_v);
    HIP_TRY(hipMemcpyAsync(h_npe, d_npe, sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipMemcpyAsync((void *)hr, d_hr, (size_t)nd * sizeof(CRecH), hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    if (d_tmp) HIP_TRY(hipFree(d_tmp));
    const int32_t npe = *h_npe;
    if (npe > pe_cap) { g_cond_host_gave_up = "more pit edges among the records than its scratch holds"; return 0; }
    std::vector<CPitEdge> pe((size_t)npe);
    if (npe) HIP_TRY(hipMemcpy(pe.data(), d_pe, (size_t)npe * sizeof(CPitEdge), hipMemcpyDeviceToHost));
    const double t_copied = host_now_ms();
    std::sort(pe.begin(), pe.end(), [](const CPitEdge &a, const CPitEdge &b) { return a.src != b.src ? a.src < b.src : a.dst < b.dst; });
    // The order of the watched nodes is formed on a thread of its own beside the adjacency.  The counting passes of the
    // adjacency CAN run on several threads (PYDEM_COND_THREADS=<n>), but the default is one: on the two-socket hosts of the
    // GPU boxes the arrays the workers touch first land on their memory nodes, and the reverse sweep that follows (one thread,
    // random access into exactly those arrays) loses more (12-13 -> 14-25 ms) than the passes gain (8 -> 5 ms).
    // (Also measured and not kept: the sweep as a depth-first post-order over the out-edges alone, without predecessor lists --
    // adjacency 8 -> 5 ms, sweep 12.5 -> 17 ms: every record is visited twice and its targets' states once more.)
    static int n_thr = -1;
    if (n_thr < 0) {
        const char *e = getenv("PYDEM_COND_THREADS");
        const int hw = (int)std::thread::hardware_concurrency();
        n_thr = e ? atoi(e) : 1;
        n_thr = std::max(1, std::min(n_thr, std::min(16, hw > 0 ? hw : 1)));
    }
    const int T = nd < 20000 ? 1 : n_thr;
    const bool order_thread = nd >= 20000;
    auto par_for = [&](int64_t count, const std::function<void(int64_t, int64_t, int)> &fn) {
        if (T == 1 || count < 4096) { fn(0, count, 0); return; }
        std::vector<std::thread> th;
        for (int q = 1; q < T; q++) th.emplace_back(fn, count * q / T, count * (q + 1) / T, q);
        fn(0, count / T, 0);
        for (auto &x : th) x.join();
    };
    // ---- watched records in ascending cell order (on its own thread beside the adjacency)
    std::vector<int32_t> wrec, wid((size_t)nd, -1);
    auto node_order = [&]() {
        std::vector<std::pair<int32_t, int32_t>> key;
        for (int32_t k = 0; k < nd; k++) if (hr[k].wid == -2) key.emplace_back(hr[k].cell, k);
        std::sort(key.begin(), key.end());
        wrec.resize(key.size());
        for (size_t w = 0; w < key.size(); w++) { wrec[w] = key[w].second; wid[(size_t)key[w].second] = (int32_t)w; }
    };
    std::thread th_order;
    if (order_thread) th_order = std::thread(node_order);
    struct JoinGuard { std::thread &t; ~JoinGuard() { if (t.joinable()) t.join(); } } guard_order{th_order};
    // ---- out-edges per record (regular ones first, then the pit edges), in-degrees, predecessor lists
    std::vector<int32_t> ob((size_t)nd + 1, 0);
    par_for(nd, [&](int64_t k0, int64_t k1, int) { for (int64_t k = k0; k < k1; k++) ob[(size_t)k + 1] = (hr[k].out_id[0] >= 0) + (hr[k].out_id[1] >= 0); });
    for (const auto &e : pe) ob[(size_t)e.src + 1]++;
    for (int32_t k = 0; k < nd; k++) ob[(size_t)k + 1] += ob[(size_t)k];
    const int64_t n_out = ob[(size_t)nd];
    std::vector<int32_t> ot((size_t)n_out); std::vector<double> ow((size_t)n_out);
    std::vector<int32_t> indeg((size_t)nd, 0), pb((size_t)nd + 1, 0);
    bool bad_target = false;
    par_for(nd, [&](int64_t k0, int64_t k1, int) {
        const CPitEdge *q = std::lower_bound(pe.data(), pe.data() + pe.size(), (int32_t)k0, [](const CPitEdge &a, int32_t v) { return a.src < v; });
        const CPitEdge *qe = pe.data() + pe.size();
        for (int64_t k = k0; k < k1; k++) {
            int32_t f = ob[(size_t)k];
            for (int j = 0; j < 2; j++)
                if (hr[k].out_id[j] >= 0) { ot[(size_t)f] = hr[k].out_id[j]; ow[(size_t)f++] = hr[k].out_w[j]; }
            for (; q != qe && q->src == (int32_t)k; q++) { ot[(size_t)f] = q->dst; ow[(size_t)f++] = q->w; }
            for (int32_t e = ob[(size_t)k]; e < f; e++) {
                const int32_t tg = ot[(size_t)e];
                if (tg < 0 || tg >= nd) { bad_target = true; continue; }
                __atomic_fetch_add(&indeg[(size_t)tg], 1, __ATOMIC_RELAXED);
            }
        }
    });
    if (bad_target) { g_cond_host_gave_up = "an out-edge that leaves the records"; return 0; }
    for (int32_t k = 0; k < nd; k++) pb[(size_t)k + 1] = pb[(size_t)k] + indeg[(size_t)k];
    std::vector<int32_t> pred((size_t)n_out);
    {
        std::vector<int32_t> fill(pb.begin(), pb.end() - 1);
        par_for(nd, [&](int64_t k0, int64_t k1, int) {
            for (int64_t k = k0; k < k1; k++)
                for (int32_t e = ob[(size_t)k]; e < ob[(size_t)k + 1]; e++)
                    pred[(size_t)__atomic_fetch_add(&fill[(size_t)ot[(size_t)e]], 1, __ATOMIC_RELAXED)] = (int32_t)k;
        });
    }
    const double t_csr = host_now_ms();
    if (order_thread) th_order.join(); else node_order();
    const int32_t nw = (int32_t)wrec.size();
    const double t_wsort = host_now_ms();
    // ---- reverse topological order: X(k) = the watched cells the water of k reaches next, with the path weights, kept as
    // scale[k] * V(rep[k]): a cell with ONE out-edge shares the vector of its target (rep < 0: the unit vector of watched
    // node -1 - rep), only the cells where the flow splits merge two (sorted) vectors into a new one.  Vectors live in
    // chunks that never move; vref[id] = where vector id is.
    typedef std::pair<int32_t, double> Ent;
    struct VecRef { const Ent *p; int64_t n; };
    std::unique_ptr<VecRef[]> vref(new VecRef[(size_t)nd + 1]);
    int32_t n_vec = 0;
    int64_t n_ent = 0;
    std::vector<int32_t> rep((size_t)nd, INT32_MIN);     // INT32_MIN: the empty vector (the water ends inside the tile)
    std::vector<double> scale((size_t)nd, 0.0);
    // (the sweep itself stays on one thread: the graph of the records is a bundle of rivers, narrow and thousands of records
    // deep -- a Kahn pass shared by 4 / 8 threads over a common ready list measured 100-150 ms against 12: every record then
    // costs a few cache-line transfers between cores.  Last in, first out: a river is walked while its lines are warm.)
    std::vector<int32_t> out_left((size_t)nd), stack;
    for (int32_t k = 0; k < nd; k++) { out_left[(size_t)k] = ob[(size_t)k + 1] - ob[(size_t)k]; if (!out_left[(size_t)k]) stack.push_back(k); }
    constexpr size_t CHUNK = (size_t)1 << 18;
    std::vector<std::unique_ptr<Ent[]>> chunks;
    Ent *cur = nullptr; size_t cur_left = 0;
    int64_t processed = 0;
    std::vector<Ent> acc, nxt;
    // the vector of target tg as seen through an edge of weight w: (rep, factor)
    auto through = [&](int32_t tg, double w, int32_t &r, double &f) {
        if (wid[(size_t)tg] >= 0) { r = -1 - wid[(size_t)tg]; f = w; }
        else { r = rep[(size_t)tg]; f = w * scale[(size_t)tg]; }
    };
    auto add_into = [&](int32_t r, double f) {            // acc += f * V(r), both sorted by node
        if (r == INT32_MIN) return;
        Ent unit(-1 - r, 1.0);
        const Ent *vb = r < 0 ? &unit : vref[(size_t)r].p, *ve = r < 0 ? &unit + 1 : vref[(size_t)r].p + vref[(size_t)r].n;
        nxt.clear();
        size_t i = 0;
        for (const Ent *p = vb; p != ve; p++) {
            while (i < acc.size() && acc[i].first < p->first) nxt.push_back(acc[i++]);
            if (i < acc.size() && acc[i].first == p->first) { nxt.emplace_back(p->first, acc[i].second + f * p->second); i++; }
            else nxt.emplace_back(p->first, f * p->second);
        }
        while (i < acc.size()) nxt.push_back(acc[i++]);
        acc.swap(nxt);
    };
    while (!stack.empty()) {
        const int32_t k = stack.back(); stack.pop_back();
        processed++;
        const int32_t e0 = ob[(size_t)k], e1 = ob[(size_t)k + 1];
        if (e1 - e0 == 1) through(ot[(size_t)e0], ow[(size_t)e0], rep[(size_t)k], scale[(size_t)k]);
        else if (e1 - e0 >= 2) {
            acc.clear();
            for (int32_t e = e0; e < e1; e++) { int32_t r; double f; through(ot[(size_t)e], ow[(size_t)e], r, f); add_into(r, f); }
            if (!acc.empty()) {
                if (acc.size() > cur_left) {
                    const size_t sz = std::max(CHUNK, acc.size());
                    chunks.emplace_back(new Ent[sz]);
                    cur = chunks.back().get(); cur_left = sz;
                }
                std::copy(acc.begin(), acc.end(), cur);
                vref[(size_t)n_vec].p = cur; vref[(size_t)n_vec].n = (int64_t)acc.size();
                cur += acc.size(); cur_left -= acc.size();
                rep[(size_t)k] = n_vec++; scale[(size_t)k] = 1.0;
                n_ent += (int64_t)acc.size();
                if (n_ent > ((int64_t)1 << 27)) { g_cond_host_gave_up = "a pathological fan"; return 0; }    // (keep the cell-by-cell rounds)
            }
        }
        for (int32_t e = pb[(size_t)k]; e < pb[(size_t)k + 1]; e++) if (--out_left[(size_t)pred[(size_t)e]] == 0) stack.push_back(pred[(size_t)e]);
    }
    if (processed != nd) { g_cond_host_gave_up = "a cycle among the records"; return 0; }       // not a DAG: plain cascade
    const double t_swept = host_now_ms();
    // ---- nodes, edges, slots
    if ((size_t)nw > nw_bound) { pydem_set_error("condensed edge rounds: %d watched nodes, expected at most %zu", nw, nw_bound); return -5; }
    CNode *nodes = reinterpret_cast<CNode *>((char *)pin_v + hr_bytes);
    auto vsize = [&](int32_t r) -> int64_t { return r == INT32_MIN ? 0 : (r < 0 ? 1 : vref[(size_t)r].n); };
    int64_t ne_all = 0;
    for (int32_t w = 0; w < nw; w++) ne_all += vsize(rep[(size_t)wrec[(size_t)w]]);
    if (ne_all > INT32_MAX / 2) { g_cond_host_gave_up = "too many edges"; return 0; }
    // all edges in source order first (dst, weight), in-degrees; then the split into inline / array parts
    std::vector<int32_t> e_dst((size_t)ne_all); std::vector<double> e_w((size_t)ne_all);
    std::vector<int32_t> ebeg((size_t)nw + 1, 0), n_in((size_t)nw, 0);
    {
        int64_t e = 0;
        for (int32_t w = 0; w < nw; w++) {
            const int32_t k = wrec[(size_t)w];
            const int32_t r = rep[(size_t)k];
            const double f = scale[(size_t)k];
            if (r != INT32_MIN && r < 0) { e_dst[(size_t)e] = -1 - r; e_w[(size_t)e] = f; e++; }
            else if (r != INT32_MIN)
                for (int64_t q = 0; q < vref[(size_t)r].n; q++) { e_dst[(size_t)e] = vref[(size_t)r].p[q].first; e_w[(size_t)e] = f * vref[(size_t)r].p[q].second; e++; }
            ebeg[(size_t)w + 1] = (int32_t)e;
        }
        for (int64_t q = 0; q < ne_all; q++) n_in[(size_t)e_dst[(size_t)q]]++;
    }
    std::vector<int32_t> in_base((size_t)nw + 1, 0), out_base((size_t)nw + 1, 0);
    for (int32_t w = 0; w < nw; w++) {
        in_base[(size_t)w + 1] = in_base[(size_t)w] + std::max(0, n_in[(size_t)w] - 2);
        out_base[(size_t)w + 1] = out_base[(size_t)w] + std::max(0, ebeg[(size_t)w + 1] - ebeg[(size_t)w] - 2);
    }
    const int64_t ne = out_base[(size_t)nw], nslot = in_base[(size_t)nw];      // array parts
    std::vector<CEdge> edges((size_t)std::max<int64_t>(ne, 1));
    std::vector<int32_t> fill((size_t)nw, 0);                                    // next in-slot of a node (sources ascend with the edge order)
    for (int32_t w = 0; w < nw; w++) {
        CNode &N = nodes[(size_t)w];
        memset(&N, 0, sizeof(N));
        const int32_t k = wrec[(size_t)w];
        N.rec = k; N.cell = hr[k].cell; N.cw = hr[k].cw;
        N.n_in = n_in[(size_t)w]; N.n_out = ebeg[(size_t)w + 1] - ebeg[(size_t)w];
        N.in_base = in_base[(size_t)w]; N.out_base = out_base[(size_t)w];
        for (int e = 0; e < N.n_out; e++) {
            const int64_t q = (int64_t)ebeg[(size_t)w] + e;
            CEdge ed;
            ed.dst = e_dst[(size_t)q]; ed.w = e_w[(size_t)q];
            const int32_t sl = fill[(size_t)ed.dst]++;
            ed.slot = sl < 2 ? -1 - sl : in_base[(size_t)ed.dst] + sl - 2;
            if (e < 2) N.e_inl[e] = ed; else edges[(size_t)(N.out_base + e - 2)] = ed;
        }
        const int32_t outside = hr[k].cnt - indeg[(size_t)k];                    // +1 while the cell is a 'todo' inlet (k_nd_link)
        if (outside != 0 && outside != 1) { pydem_set_error("condensed edge rounds: inconsistent count of record %d", k); return -5; }
        N.cnt = N.n_in + outside;
    }
    // ---- device copy (one allocation: nodes | edges | slots | two queues | NaN list | counters)
    const size_t nan_cap = (size_t)ne_all + (size_t)nw + 64;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_node = take((size_t)nw * sizeof(CNode)), o_edge = take((size_t)(ne + 1) * sizeof(CEdge)), o_slot = take((size_t)(nslot + 1) * 8),
                 o_q0 = take((size_t)nw * 4), o_q1 = take((size_t)nw * 4), o_nan = take(nan_cap * 4), o_cnt = take(64);
    if (off > t->cond_bytes) {
        if (t->cond_mem) { HIP_TRY(hipFree(t->cond_mem)); t->device_bytes -= (int64_t)t->cond_bytes; t->cond_mem = nullptr; t->cond_bytes = 0; }
        HIP_TRY(dev_malloc((void **)&t->cond_mem, off + off / 8));
        t->cond_bytes = off + off / 8; t->device_bytes += (int64_t)t->cond_bytes;
    }
    char *base = (char *)t->cond_mem;
    t->cond_node = base + o_node; t->cond_edge = base + o_edge; t->cond_slot = (double *)(base + o_slot);
    t->cond_q0 = (int32_t *)(base + o_q0); t->cond_q1 = (int32_t *)(base + o_q1); t->cond_nanq = (int32_t *)(base + o_nan);
    t->cond_cnt = (int32_t *)(base + o_cnt); t->cond_nw = nw; t->cond_nan_cap = (int32_t)std::min<size_t>(nan_cap, (size_t)INT32_MAX);
    HIP_TRY(hipMemsetAsync(base + o_slot, 0, off - o_slot, t->stream));
    if (nw) HIP_TRY(hipMemcpyAsync(t->cond_node, nodes, (size_t)nw * sizeof(CNode), hipMemcpyHostToDevice, t->stream));
    if (ne) HIP_TRY(hipMemcpyAsync(t->cond_edge, edges.data(), (size_t)ne * sizeof(CEdge), hipMemcpyHostToDevice, t->stream));
    CondArgsE X;
    t->cond_live = true;                  // (cond_args reads the fields set above)
    PYDEM_TRY(cond_args(t, X));
    if (nw) hipLaunchKernelGGL(k_cond_attach, dim3(grid_for(nw, 256)), dim3(256), 0, t->stream, X);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(t->stream));     // (nodes / edges are host vectors about to go out of scope)
    t->watch_built = t->watch.size();
    if (getenv("PYDEM_EDGE_DEBUG"))
        fprintf(stderr, "condensed edge rounds: %d records -> %d watched nodes, %lld edges (%d pit edges among the records); %.2f ms "
                "(copy %.2f, adjacency %.2f, node order %.2f, reverse sweep %.2f [%zu vectors, %zu entries], nodes + upload %.2f)\n",
                nd, nw, (long long)ne_all, npe, host_now_ms() - t_begin, t_copied - t_begin, t_csr - t_copied, t_wsort - t_csr, t_swept - t_wsort,
                (size_t)n_vec, (size_t)n_ent, host_now_ms() - t_swept);
    return 0;
}

// ---- the same graph built on the device (kernels in uca_cbuild.inl) ---------------------------------------------------
static int cb_reserve(pydem_tile *t, int which, size_t bytes)
{
    if (t->cb_bytes[which] >= bytes) return 0;
    if (t->cb_mem[which]) { HIP_TRY(hipStreamSynchronize(t->stream)); HIP_TRY(hipFree(t->cb_mem[which])); t->device_bytes -= (int64_t)t->cb_bytes[which]; t->cb_mem[which] = nullptr; t->cb_bytes[which] = 0; }
    const size_t want = (bytes + bytes / 4 + ((size_t)1 << 20)) & ~(((size_t)1 << 20) - 1);     // (headroom + 1 MiB steps: run-to-run sizes move by a few records)
    HIP_TRY(dev_malloc(&t->cb_mem[which], want));
    t->cb_bytes[which] = want; t->device_bytes += (int64_t)want;
    return 0;
}

struct CBump {
    char *base; size_t off = 0;
    explicit CBump(void *b) : base((char *)b) {}
    template <typename T> T *take(size_t count) { T *p = base ? (T *)(base + off) : nullptr; off += (count * sizeof(T) + 255) & ~(size_t)255; return p; }
};

// *status: 1 built (cond_live), 0 the tile does not qualify (a cycle among the records: the host build would say the same),
// -1 the device build gave up (a vector of more than CB_RUN entries, pool overflow): try the host build
static int cond_build_device(pydem_tile *t, int *status)
{
    *status = -1;
    t->cond_live = false; t->cond_pending = false;
    const double t_begin = host_now_ms();
    const int32_t nd = t->nd;
    const int n = (int)t->n, m = (int)t->m;
    CBArgs B;
    memset(&B, 0, sizeof(B));
    PYDEM_TRY(cinc_args(t, B.C));
    const CIncArgs &C = B.C;
    auto mark = [&](int axis, int64_t index) {
        const int64_t count = axis == 0 ? m : n;
        hipLaunchKernelGGL(k_cond_mark, dim3((unsigned)std::min<int64_t>(cdiv(count, 256), 64)), dim3(256), 0, t->stream, C, axis, index);
    };
    mark(0, 0); mark(0, n - 1); mark(1, 0); mark(1, m - 1);
    for (const auto &w : t->watch) mark(w.first, w.second);
    const size_t nw_bound = (size_t)std::min<int64_t>((int64_t)nd, (int64_t)(4 + t->watch.size()) * (int64_t)std::max(n, m));
    const int nd1 = nd + 1;
    size_t tmp_scan = 0, tmp_sortw = 0;
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_scan, (int32_t *)nullptr, (int32_t *)nullptr, nd1, t->stream));
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_sortw, (int32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr,
                                               (int)nw_bound, 0, 32, t->stream));
    const size_t tmp1 = std::max(tmp_scan, tmp_sortw) + 256;
    // ---- phase 1: per-record state, counts
    void *tmp_a = nullptr;
    auto lay1 = [&](void *base) {
        CBump A(base);
        B.rv = A.take<CBVal>((size_t)nd); B.ri = A.take<CBRec>((size_t)nd);
        B.pred_cnt = A.take<int32_t>((size_t)nd1); B.pit_cnt = A.take<int32_t>((size_t)nd1);
        B.pred_beg = A.take<int32_t>((size_t)nd1); B.pit_beg = A.take<int32_t>((size_t)nd1);
        B.q0 = A.take<int32_t>((size_t)nd * CB_NQ); B.q1 = A.take<int32_t>((size_t)nd * CB_NQ);     // (CB_NQ sub-queues each: any of them may hold a whole level)
        B.qcnt = A.take<int32_t>((size_t)3 * CB_NQ * CB_PAD); B.poolc = A.take<int32_t>((size_t)CB_NQ * CB_PAD);
        B.wcell = A.take<int32_t>(nw_bound); B.wrec = A.take<int32_t>(nw_bound);
        B.wcell_s = A.take<int32_t>(nw_bound); B.wrec_s = A.take<int32_t>(nw_bound);
        B.ctr = A.take<int32_t>(CBC_WORDS);
        tmp_a = A.take<char>(tmp1);
        return A.off;
    };
    PYDEM_TRY(cb_reserve(t, 0, lay1(nullptr)));
    lay1(t->cb_mem[0]);
    B.w_cap = (int32_t)nw_bound;
    B.w_sorted = t->pits.w;
    HIP_TRY(hipMemsetAsync(B.ctr, 0, CBC_WORDS * sizeof(int32_t), t->stream));
    HIP_TRY(hipMemsetAsync(B.qcnt, 0, (size_t)4 * CB_NQ * CB_PAD * sizeof(int32_t), t->stream));      // (qcnt and, behind it, poolc)
    B.qcap = nd;
    const int g_nd = grid_for(nd, 1024);
    hipLaunchKernelGGL(k_cb_count, dim3(g_nd), dim3(256), 0, t->stream, B);
    { size_t tb = tmp1; HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp_a, tb, B.pred_cnt, B.pred_beg, nd1, t->stream)); }
    { size_t tb = tmp1; HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp_a, tb, B.pit_cnt, B.pit_beg, nd1, t->stream)); }
    int32_t *h = t->h_counters;
    HIP_TRY(hipMemcpyAsync(h, B.ctr, CBC_WORDS * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipMemcpyAsync(h + 16, B.pred_beg + nd, sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipMemcpyAsync(h + 17, B.pit_beg + nd, sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(t->stream));
    const int32_t nw = h[CBC_NW], n_pred = h[16], n_pit = h[17];
    if (h[CBC_FAIL] & 2) { pydem_set_error("condensed edge rounds: inconsistent in-edge count of a record"); return -5; }
    if ((size_t)nw > nw_bound) { pydem_set_error("condensed edge rounds: %d watched nodes, expected at most %zu", nw, nw_bound); return -5; }
    const double t_counted = host_now_ms();
    // ---- phase 2: lists, node order, the reverse sweep
    const int64_t pool_cap = std::min<int64_t>((int64_t)nd + 16384, (int64_t)1 << 24);       // entries per region (CB_NQ regions: 16 x nd in all, ~8 x what the merges of a 16384^2 tile take)
    const int nw1 = nw + 1;
    auto lay2 = [&](void *base) {
        CBump A(base);
        B.pred = A.take<int32_t>((size_t)n_pred + 1); B.pit = A.take<CBPit>((size_t)n_pit + 1);
        B.pool = A.take<CBEnt>((size_t)pool_cap * CB_NQ);
        B.nout_c = A.take<int32_t>((size_t)nw1); B.nout = A.take<int32_t>((size_t)nw1);
        B.n_in = A.take<int32_t>((size_t)nw1); B.in_first = A.take<int32_t>((size_t)nw1);
        B.exc_in_c = A.take<int32_t>((size_t)nw1); B.exc_in = A.take<int32_t>((size_t)nw1);
        B.exc_out_c = A.take<int32_t>((size_t)nw1); B.exc_out = A.take<int32_t>((size_t)nw1);
        return A.off;
    };
    PYDEM_TRY(cb_reserve(t, 1, lay2(nullptr)));
    lay2(t->cb_mem[1]);
    B.pool_cap = (int32_t)pool_cap; B.nw = nw;
    hipLaunchKernelGGL(k_cb_fill, dim3(g_nd), dim3(256), 0, t->stream, B);
    int cell_bits = 1;
    while (((int64_t)1 << cell_bits) < t->NN) cell_bits++;
    if (nw > 0) {
        size_t tb = tmp1;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp_a, tb, B.wcell, B.wcell_s, B.wrec, B.wrec_s, nw, 0, cell_bits, t->stream));
        hipLaunchKernelGGL(k_cb_wid, dim3(grid_for(nw, 256)), dim3(256), 0, t->stream, B);
    }
    // the levels: one launch per level over the whole chip, in batches; one look from the host per batch (the launches behind the
    // end of the sweep find empty sub-queues and return at once)
    static int cb_grid = -1, cb_chain = -1;
    if (cb_grid < 0) { const char *e = getenv("PYDEM_CB_GRID"); cb_grid = e ? std::max(1, std::min(atoi(e), 65536)) : CB_GRID; cb_grid = ((cb_grid + CB_NQ - 1) / CB_NQ) * CB_NQ; }
    if (cb_chain < 0) { const char *e = getenv("PYDEM_CB_CHAIN"); cb_chain = e ? std::max(0, atoi(e)) : 2; }
    B.max_chain = cb_chain;
    int32_t *d_dbg = nullptr;                              // PYDEM_CB_DEBUG=1: per-level statistics of the sweep to stderr (diagnostic, one extra allocation)
    const int dbg_levels = 4096;
    if (getenv("PYDEM_CB_DEBUG")) { HIP_TRY(hipMalloc((void **)&d_dbg, (size_t)dbg_levels * 4 * sizeof(int32_t))); HIP_TRY(hipMemsetAsync(d_dbg, 0, (size_t)dbg_levels * 4 * sizeof(int32_t), t->stream)); }
    B.dbg = d_dbg; B.level = 0;
    int levels_run = 0;
    void *pin_q = nullptr;
    PYDEM_TRY(tile_pinned(t, (size_t)3 * CB_NQ * CB_PAD * sizeof(int32_t), &pin_q));
    const int32_t *hq = (const int32_t *)pin_q;
    for (;;) {
        for (int b = 0; b < 64; b++, levels_run++) {
            B.level = levels_run < dbg_levels ? levels_run : dbg_levels - 1;
            hipLaunchKernelGGL(k_cb_level, dim3(cb_grid), dim3(CB_LANES), 0, t->stream, B, levels_run);
        }
        HIP_TRY(hipMemcpyAsync(h, B.ctr, CBC_WORDS * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipMemcpyAsync(pin_q, B.qcnt, (size_t)3 * CB_NQ * CB_PAD * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        int64_t left = 0;
        for (int q = 0; q < CB_NQ; q++) left += hq[((levels_run % 3) * CB_NQ + q) * CB_PAD];
        if (left == 0 || (h[CBC_FAIL] & 1)) break;
        if (levels_run > (1 << 22)) { pydem_set_error("condensed edge rounds: flow paths too long"); return -5; }
    }
    if (d_dbg) {
        std::vector<int32_t> hd((size_t)dbg_levels * 4);
        HIP_TRY(hipMemcpy(hd.data(), d_dbg, hd.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipFree(d_dbg));
        fprintf(stderr, "cb levels (frontier / largest merge / entries merged / deepest chain):");
        for (int l = 0; l < levels_run && l < dbg_levels; l++) { if (l % 8 == 0) fprintf(stderr, "\n  %4d:", l); fprintf(stderr, " %d/%d/%d/%d", hd[4 * l], hd[4 * l + 1], hd[4 * l + 2], hd[4 * l + 3]); }
        fprintf(stderr, "\n");
        B.dbg = nullptr;
    }
    hipLaunchKernelGGL(k_cb_nout, dim3(grid_for(nw1, 256)), dim3(256), 0, t->stream, B);
    { size_t tb = tmp1; HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp_a, tb, B.nout_c, B.nout, nw1, t->stream)); }
    HIP_TRY(hipMemcpyAsync(h, B.ctr, CBC_WORDS * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipMemcpyAsync(h + 16, B.nout + nw, sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(t->stream));
    const double t_swept = host_now_ms();
    const int32_t ne_all = h[16], levels = h[CBC_LEVELS], n_slow = h[CBC_SLOW];
    if (h[CBC_FAIL] & 1) return 0;                              // (*status == -1: the host build takes over)
    { const char *e = getenv("PYDEM_CB_FORCE_FALLBACK"); if (e && atoi(e) > 0) return 0; }      // (tests: the hand-over to the host build after a finished sweep)
    if (h[CBC_PROC] != nd) { *status = 0; return 0; }           // a cycle among the records: not a DAG, plain cascade
    if (ne_all > INT32_MAX / 2) { *status = 0; return 0; }
    // ---- phase 3: edges, slots, nodes -- straight into the round's arrays (one allocation: nodes | edges | slots | two queues |
    // NaN list | counters; edge / slot arrays sized by the bound ne_all)
    size_t tmp_sorte = 0, tmp_scanw = 0;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_sorte, (int32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr,
                                               (int)std::max(ne_all, 1), 0, 32, t->stream));
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_scanw, (int32_t *)nullptr, (int32_t *)nullptr, nw1, t->stream));
    const size_t tmp3 = std::max(tmp_sorte, tmp_scanw) + 256;
    void *tmp_c = nullptr;
    auto lay3 = [&](void *base) {
        CBump A(base);
        const size_t ne1 = (size_t)ne_all + 1;
        B.e_dst = A.take<int32_t>(ne1); B.e_q = A.take<int32_t>(ne1); B.e_dst_s = A.take<int32_t>(ne1); B.e_q_s = A.take<int32_t>(ne1);
        B.e_slot = A.take<int32_t>(ne1); B.e_w = A.take<double>(ne1);
        tmp_c = A.take<char>(tmp3);
        return A.off;
    };
    PYDEM_TRY(cb_reserve(t, 2, lay3(nullptr)));
    lay3(t->cb_mem[2]);
    const size_t nan_cap = (size_t)ne_all + (size_t)nw + 64;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_node = take((size_t)nw * sizeof(CNode)), o_edge = take((size_t)(ne_all + 1) * sizeof(CEdge)), o_slot = take((size_t)(ne_all + 1) * 8),
                 o_q0 = take((size_t)nw * 4), o_q1 = take((size_t)nw * 4), o_nan = take(nan_cap * 4), o_cnt = take(64);
    if (off > t->cond_bytes) {
        if (t->cond_mem) { HIP_TRY(hipFree(t->cond_mem)); t->device_bytes -= (int64_t)t->cond_bytes; t->cond_mem = nullptr; t->cond_bytes = 0; }
        const size_t want = (off + off / 4 + ((size_t)1 << 20)) & ~(((size_t)1 << 20) - 1);
        HIP_TRY(dev_malloc((void **)&t->cond_mem, want));
        t->cond_bytes = want; t->device_bytes += (int64_t)t->cond_bytes;
    }
    char *base = (char *)t->cond_mem;
    t->cond_node = base + o_node; t->cond_edge = base + o_edge; t->cond_slot = (double *)(base + o_slot);
    t->cond_q0 = (int32_t *)(base + o_q0); t->cond_q1 = (int32_t *)(base + o_q1); t->cond_nanq = (int32_t *)(base + o_nan);
    t->cond_cnt = (int32_t *)(base + o_cnt); t->cond_nw = nw; t->cond_nan_cap = (int32_t)std::min<size_t>(nan_cap, (size_t)INT32_MAX);
    B.node = (CNode *)t->cond_node; B.edge = (CEdge *)t->cond_edge;
    HIP_TRY(hipMemsetAsync(base + o_slot, 0, off - o_slot, t->stream));
    if (nw > 0) {
        const int g_nw = grid_for(nw1, 256);
        hipLaunchKernelGGL(k_cb_edges, dim3(g_nw), dim3(256), 0, t->stream, B);
        hipLaunchKernelGGL(k_cb_excess, dim3(g_nw), dim3(256), 0, t->stream, B);
        { size_t tb = tmp3; HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp_c, tb, B.n_in, B.in_first, nw1, t->stream)); }
        { size_t tb = tmp3; HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp_c, tb, B.exc_in_c, B.exc_in, nw1, t->stream)); }
        { size_t tb = tmp3; HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp_c, tb, B.exc_out_c, B.exc_out, nw1, t->stream)); }
        if (ne_all > 0) {
            int node_bits = 1;
            while (((int64_t)1 << node_bits) < nw) node_bits++;
            size_t tb = tmp3;
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp_c, tb, B.e_dst, B.e_dst_s, B.e_q, B.e_q_s, ne_all, 0, node_bits, t->stream));
            hipLaunchKernelGGL(k_cb_slots, dim3(grid_for(ne_all, 256)), dim3(256), 0, t->stream, B, ne_all);
        }
        hipLaunchKernelGGL(k_cb_nodes, dim3(g_nw), dim3(256), 0, t->stream, B);
    }
    CondArgsE X;
    t->cond_live = true;                  // (cond_args reads the fields set above)
    PYDEM_TRY(cond_args(t, X));
    if (nw) hipLaunchKernelGGL(k_cond_attach, dim3(grid_for(nw, 256)), dim3(256), 0, t->stream, X);
    HIP_TRY(hipGetLastError());
    t->watch_built = t->watch.size();
    *status = 1;
    if (getenv("PYDEM_EDGE_DEBUG")) {
        HIP_TRY(hipStreamSynchronize(t->stream));
        fprintf(stderr, "condensed edge rounds (device build): %d records -> %d watched nodes, %d edges (%d pit edges among the records); %.2f ms "
                "(lists %.2f, reverse sweep %.2f [%d levels, %d launches, %d merges from the pool], nodes %.2f)\n",
                nd, nw, ne_all, n_pit, host_now_ms() - t_begin, t_counted - t_begin, t_swept - t_counted, levels, levels_run, n_slow, host_now_ms() - t_swept);
    }
    return 0;
}

// PYDEM_COND_BUILD=check: the device build against the host build, node by node (same nodes, counts, edges and slots;
// weights to 1e-12 relative: the host sorts the pit edges of one pit by record id, the device by drain cell)
static int cond_build_check(pydem_tile *t)
{
    int st = -1;
    PYDEM_TRY(cond_build_device(t, &st));
    if (st != 1) { PYDEM_TRY(cond_build_host(t)); if (st == 0 && t->cond_live) { pydem_set_error("condensed build check: the device build found a cycle, the host build none"); return -5; } return 0; }
    HIP_TRY(hipStreamSynchronize(t->stream));
    const int32_t nw = t->cond_nw;
    std::vector<CNode> dn((size_t)std::max(nw, 1));
    std::vector<CEdge> de;
    auto grab = [&](std::vector<CNode> &nodes, std::vector<CEdge> &edges) -> int {
        nodes.resize((size_t)std::max(t->cond_nw, 1));
        if (t->cond_nw) HIP_TRY(hipMemcpy(nodes.data(), t->cond_node, (size_t)t->cond_nw * sizeof(CNode), hipMemcpyDeviceToHost));
        int64_t ne = 0;
        for (int32_t w = 0; w < t->cond_nw; w++) ne = std::max<int64_t>(ne, (int64_t)nodes[(size_t)w].out_base + std::max(0, nodes[(size_t)w].n_out - 2));
        edges.resize((size_t)std::max<int64_t>(ne, 1));
        if (ne) HIP_TRY(hipMemcpy(edges.data(), t->cond_edge, (size_t)ne * sizeof(CEdge), hipMemcpyDeviceToHost));
        return 0;
    };
    PYDEM_TRY(grab(dn, de));
    // (the device build has attached the nodes to their records: undo that before the host build reads them again)
    std::vector<CNode> hn; std::vector<CEdge> he;
    {
        CondArgsE X; PYDEM_TRY(cond_args(t, X));
        if (nw) hipLaunchKernelGGL(k_cond_detach, dim3(grid_for(nw, 256)), dim3(256), 0, t->stream, X);
    }
    PYDEM_TRY(cond_build_host(t));
    if (!t->cond_live) {
        // (a capacity limit of the host build -- the pit edges among the records go through a scratch of NN / 4 entries -- is not a
        // difference: the device build has no such limit and its operator stands; anything else is)
        if (!strcmp(g_cond_host_gave_up, "more pit edges among the records than its scratch holds")) {
            if (getenv("PYDEM_EDGE_DEBUG")) fprintf(stderr, "condensed build check: host build skipped (%s); device operator rebuilt and kept\n", g_cond_host_gave_up);
            int st2 = -1;
            PYDEM_TRY(cond_build_device(t, &st2));
            if (st2 != 1) { pydem_set_error("condensed build check: the device build did not repeat itself"); return -5; }
            return 0;
        }
        pydem_set_error("condensed build check: the host build gave up (%s) where the device build did not", g_cond_host_gave_up);
        return -5;
    }
    HIP_TRY(hipStreamSynchronize(t->stream));
    PYDEM_TRY(grab(hn, he));
    if (t->cond_nw != nw) { pydem_set_error("condensed build check: %d nodes on the device, %d on the host", nw, t->cond_nw); return -5; }
    double worst = 0.0;
    for (int32_t w = 0; w < nw; w++) {
        const CNode &a = dn[(size_t)w], &b = hn[(size_t)w];
        if (a.cell != b.cell || a.cw != b.cw || a.cnt != b.cnt || a.n_in != b.n_in || a.n_out != b.n_out || a.in_base != b.in_base || a.out_base != b.out_base) {
            pydem_set_error("condensed build check: node %d (cell %d / %d): cnt %d / %d, in %d / %d, out %d / %d, bases %d %d / %d %d", w, a.cell, b.cell, a.cnt, b.cnt,
                            a.n_in, b.n_in, a.n_out, b.n_out, a.in_base, a.out_base, b.in_base, b.out_base);
            return -5;
        }
        for (int e = 0; e < a.n_out; e++) {
            const CEdge &x = e < 2 ? a.e_inl[e] : de[(size_t)(a.out_base + e - 2)], &y = e < 2 ? b.e_inl[e] : he[(size_t)(b.out_base + e - 2)];
            if (x.dst != y.dst || x.slot != y.slot) { pydem_set_error("condensed build check: node %d edge %d: dst %d / %d, slot %d / %d", w, e, x.dst, y.dst, x.slot, y.slot); return -5; }
            const double d = fabs(x.w - y.w) / (fabs(y.w) > 0 ? fabs(y.w) : 1.0);
            if (!(d <= 1e-12)) { pydem_set_error("condensed build check: node %d edge %d: weight %.17g / %.17g", w, e, x.w, y.w); return -5; }
            worst = std::max(worst, d);
        }
    }
    if (getenv("PYDEM_EDGE_DEBUG")) {
        int64_t h_out[6] = {0, 0, 0, 0, 0, 0}, h_in[6] = {0, 0, 0, 0, 0, 0};      // <= 2, 3-4, 5-8, 9-16, 17-64, more
        int mx_out = 0, mx_in = 0;
        auto cls = [](int n) { return n <= 2 ? 0 : (n <= 4 ? 1 : (n <= 8 ? 2 : (n <= 16 ? 3 : (n <= 64 ? 4 : 5)))); };
        for (int32_t w = 0; w < nw; w++) { h_out[cls(dn[(size_t)w].n_out)]++; h_in[cls(dn[(size_t)w].n_in)]++; mx_out = std::max(mx_out, dn[(size_t)w].n_out); mx_in = std::max(mx_in, dn[(size_t)w].n_in); }
        fprintf(stderr, "condensed build check: %d nodes identical, weights within %.3g relative; out-edges per node <=2 / 3-4 / 5-8 / 9-16 / 17-64 / more: %lld %lld %lld %lld %lld %lld (max %d); "
                "in-edges: %lld %lld %lld %lld %lld %lld (max %d)\n", nw, worst, (long long)h_out[0], (long long)h_out[1], (long long)h_out[2], (long long)h_out[3], (long long)h_out[4], (long long)h_out[5], mx_out,
                (long long)h_in[0], (long long)h_in[1], (long long)h_in[2], (long long)h_in[3], (long long)h_in[4], (long long)h_in[5], mx_in);
    }
    return 0;
}

// Build the condensed graph of the watched cells from the compact records (just linked by einc_prepare).  Returns 0 and
// leaves cond_live false when the tile does not qualify (switched off, too many records, a cycle among the records).
static int cond_build(pydem_tile *t)
{
    t->cond_live = false; t->cond_pending = false;
    static int enabled = -1;
    if (enabled < 0) { const char *e = getenv("PYDEM_EDGE_COND"); enabled = e ? atoi(e) : 1; }
    int64_t max_nd = 1 << 20;
    { const char *e = getenv("PYDEM_EDGE_COND_MAX"); if (e) max_nd = atoll(e); }
    if (!enabled || !t->einc_compact || t->nd <= 0 || t->nd > max_nd) return 0;
    const char *how = getenv("PYDEM_COND_BUILD");           // (read per build: the tests switch it) device (default) | host | check
    if (how && !strcmp(how, "host")) return cond_build_host(t);
    if (how && !strcmp(how, "check")) return cond_build_check(t);
    int st = -1;
    PYDEM_TRY(cond_build_device(t, &st));
    if (st < 0) return cond_build_host(t);
    return 0;
}

// the interior catches up: done watched nodes -> their records, the NaN flood below the nodes it passed, ONE cascade
static int cond_catchup(pydem_tile *t, int set_done)
{
    if (!(t->einc_ready && t->cond_live)) return 0;
    CondArgsE X;
    PYDEM_TRY(cond_args(t, X));
    X.C.set_done = set_done;
    HIP_TRY(hipMemsetAsync(t->counters, 0, 16 * sizeof(int32_t), t->stream));
    if (X.nw > 0) {
        hipLaunchKernelGGL(k_cond_release, dim3(grid_for(X.nw, 256)), dim3(256), 0, t->stream, X, (QE *)t->queue[0], &t->counters[0]);
        hipLaunchKernelGGL(k_cond_nan_interior, dim3(1), dim3(1024), 0, t->stream, X);
    }
    PYDEM_TRY(cinc_cascade(t, X.C, nullptr));
    t->cond_pending = false;
    return 0;
}

int stage_edge_catchup(pydem_tile *t)
{
    if (!(t->einc_ready && t->cond_live && t->cond_pending)) return 0;
    HIP_TRY(hipSetDevice(t->device));
    return cond_catchup(t, 1);
}

int stage_edge_round_inc(pydem_tile *t, const pydem_options *opt, const double *const data[4], const uint8_t *const done[4],
                         const uint8_t *const todo[4])
{
    if (opt->apply_uca_limit_edges) {
        // edge_done is then more than "not downstream of a 'todo' inlet" (:977-980): the counts below would be wrong
        pydem_set_error("incremental edge rounds do not support apply_uca_limit_edges; use pydem_uca_edge_update");
        return -6;
    }
    // The counts of the incremental form assume that the cells that are not done form a DAG.  A tile with circular drainage
    // (the re-seed replay ran in its sweep: cells on and below a loop never count down to zero, while the reference's masks
    // only ask whether a 'todo' inlet lies upstream) runs the plain round instead; so does a tile resumed from the store,
    // whose sweep did not run in this process.
    if (t->circular_cells != 0) return stage_edge_update(t, opt, data, done, todo);
    const double t_begin = host_now_ms();
    const int n = (int)t->n, m = (int)t->m;
    const int L = n > m ? n : m;
    const int64_t nper = 2 * (int64_t)m + 2 * (int64_t)(n - 2);
    t->einc_round++;
    IncArgs E;
    PYDEM_TRY(einc_args(t, E));
    PYDEM_TRY(tile_alloc(t, &t->s_data, (size_t)L * 4));
    PYDEM_TRY(tile_alloc(t, &t->s_flags, (size_t)L * 8));
    if (!t->einc_ready) { PYDEM_TRY(einc_prepare(t, E)); PYDEM_TRY(cond_build(t)); }
    // strips -> device (left, right, top, bottom), padded to L entries each (pinned staging: the copies are asynchronous);
    // data == NULL: the edge board's evaluation kernel has already written them (comm.hip)
    if (data) {
        if (t->h_strip_cap < (size_t)L) {
            if (t->h_strip_d) { (void)hipHostFree(t->h_strip_d); (void)hipHostFree(t->h_strip_f); }
            HIP_TRY(hipHostMalloc((void **)&t->h_strip_d, (size_t)L * 4 * sizeof(double)));
            HIP_TRY(hipHostMalloc((void **)&t->h_strip_f, (size_t)L * 8));
            t->h_strip_cap = (size_t)L;
        }
        double *hd = t->h_strip_d;
        uint8_t *hf = t->h_strip_f;
        for (int s = 0; s < 4; s++) {
            const int len = s < 2 ? n : m;
            for (int k = 0; k < len; k++) {
                hd[(size_t)s * L + k] = data[s][k];
                hf[(size_t)s * L + k] = done[s][k] != 0;
                hf[(size_t)(4 + s) * L + k] = todo[s][k] != 0;
            }
        }
        HIP_TRY(hipMemcpyAsync(t->s_data, hd, (size_t)L * 4 * 8, hipMemcpyHostToDevice, t->stream));
        HIP_TRY(hipMemcpyAsync(t->s_flags, hf, (size_t)L * 8, hipMemcpyHostToDevice, t->stream));
    }
    HIP_TRY(hipMemsetAsync(t->counters, 0, 16 * sizeof(int32_t), t->stream));
#ifdef PYDEM_EINC_PROF
    HIP_TRY(hipMemsetAsync(t->counters + 40, 0, 8 * sizeof(int32_t), t->stream));
#endif
    int levels = 0;
    if (t->einc_compact && t->cond_live) {
        // condensed form: seeds + NaN flood + cascade on the watched nodes, two launches and no host look (the edge board's
        // pack kernels follow on the same stream); the interior catches up later (stage_edge_catchup / the flush)
        CondArgsE X;
        PYDEM_TRY(cond_args(t, X));
        hipLaunchKernelGGL(k_cond_seed, dim3((unsigned)cdiv(nper, 128)), dim3(128), 0, t->stream, X, t->s_data, t->s_flags,
                           t->s_flags + (size_t)4 * L, L);
        hipLaunchKernelGGL(k_cond_run, dim3(1), dim3(COND_THREADS), 0, t->stream, X);
        t->cond_pending = true;
        HIP_TRY(hipGetLastError());
        static int sync_rounds = -1;       // PYDEM_EDGE_SYNC=1: wait for the round (per-round timings of tools/pm_multitile_timing.py)
        if (sync_rounds < 0) { const char *e = getenv("PYDEM_EDGE_SYNC"); sync_rounds = e ? atoi(e) : 0; }
        if (data || sync_rounds || getenv("PYDEM_EDGE_DEBUG")) {
            HIP_TRY(hipStreamSynchronize(t->stream));            // (host strips: the pinned staging is reused by the next round)
            if (getenv("PYDEM_EDGE_DEBUG")) {
                int32_t lv[3];
                HIP_TRY(hipMemcpy(lv, t->cond_cnt, sizeof(lv), hipMemcpyDeviceToHost));
                fprintf(stderr, "condensed edge round: %d levels on %d nodes; %.3f ms\n", lv[2], t->cond_nw, host_now_ms() - t_begin);
            }
        }
        return 0;
    }
    if (t->einc_compact) {
        CIncArgs C;
        PYDEM_TRY(cinc_args(t, C));
        hipLaunchKernelGGL(k_cinc_seed, dim3((unsigned)cdiv(nper, 128)), dim3(128), 0, t->stream, C, t->s_data, t->s_flags,
                           t->s_flags + (size_t)4 * L, L, (QE *)t->queue[0], &t->counters[0]);
        hipLaunchKernelGGL(k_cinc_nan_flood, dim3(1), dim3(1024), 0, t->stream, C);
        PYDEM_TRY(cinc_cascade(t, C, &levels));
    } else {
        hipLaunchKernelGGL(k_einc_seed, dim3((unsigned)cdiv(nper, 128)), dim3(128), 0, t->stream, E, t->s_data, t->s_flags,
                           t->s_flags + (size_t)4 * L, L, (QE *)t->queue[0], &t->counters[0]);
        hipLaunchKernelGGL(k_einc_nan_flood, dim3(1), dim3(1024), 0, t->stream, E);
        PYDEM_TRY(einc_cascade(t, E, &levels));
    }
    if (getenv("PYDEM_EDGE_DEBUG"))
        fprintf(stderr, "incremental edge round: %d levels; %.3f ms\n", levels, host_now_ms() - t_begin);
#ifdef PYDEM_EINC_PROF
    if (getenv("PYDEM_EDGE_DEBUG")) {
        int32_t pr[8];
        HIP_TRY(hipMemcpy(pr, t->counters + 40, sizeof(pr), hipMemcpyDeviceToHost));
        fprintf(stderr, "   levels by frontier width <=8 / <=64 / <=512 / more: %d %d %d %d; us: %.0f %.0f %.0f %.0f\n", pr[0], pr[1], pr[2], pr[3],
                pr[4] * 0.01, pr[5] * 0.01, pr[6] * 0.01, pr[7] * 0.01);
    }
#endif
    return 0;
}

// Queued waves of the fix-up (pydem_board_run_waves, comm.hip): can this tile's next round be queued without the host knowing
// whether it will run?  Only the condensed form qualifies (two launches, no host look), after the tile's first round has
// built it.
bool tile_edge_queue_ready(const pydem_tile *t)
{
    return t->einc_ready && t->einc_compact && t->cond_live && t->circular_cells == 0 && t->s_data && t->s_flags;
}

// Entry of the queued waves' tile table (opaque to comm.hip): the condensed round of tile t, run only while bit `bit` of the
// device word *gate is set (the members of a queued wave are chosen on the device); the strips are in the tile's buffers
// (written by the board's evaluation kernel).  The seed stamp is *round_base + *round_add + 1 (mod 65535), both read on the
// device: the launches can be captured in a graph and replayed wave after wave.  The caller advances the tile's round
// counter by the waves it ran (tile_edge_rounds_ran).
size_t tile_edge_queue_desc_bytes() { return sizeof(QTile); }

int tile_edge_queue_desc(pydem_tile *t, void *out, const unsigned long long *gate, int bit, const unsigned long long *round_base,
                         const unsigned long long *round_add, int64_t *nper)
{
    if (!tile_edge_queue_ready(t)) { pydem_set_error("queued edge round: the tile's condensed fix-up state is not built"); return -3; }
    const int n = (int)t->n, m = (int)t->m;
    QTile q;
    memset(&q, 0, sizeof(q));
    PYDEM_TRY(cond_args(t, q.X));
    q.X.gate = gate; q.X.gate_bit = bit; q.X.round_base = round_base; q.X.round_add = round_add;
    q.sdata = t->s_data; q.sflags = t->s_flags; q.L = n > m ? n : m;
    q.nper = 2 * (int64_t)m + 2 * (int64_t)(n - 2);
    *nper = q.nper;
    memcpy(out, &q, sizeof(q));
    return 0;
}

// the rounds of `count` tiles (device table d_q), two launches on stream s
int stage_edge_rounds_queued(hipStream_t s, const void *d_q, int count, int64_t max_nper)
{
    if (count <= 0) return 0;
    hipLaunchKernelGGL(k_cond_seed_q, dim3((unsigned)cdiv(max_nper, 128), (unsigned)count), dim3(128), 0, s, (const QTile *)d_q);
    hipLaunchKernelGGL(k_cond_run_q, dim3((unsigned)count), dim3(COND_THREADS), 0, s, (const QTile *)d_q);
    return 0;
}

unsigned long long tile_edge_round_counter(const pydem_tile *t) { return (unsigned long long)t->einc_round; }

void tile_edge_rounds_ran(pydem_tile *t, int waves)
{
    t->einc_round += waves;
    if (waves > 0) t->cond_pending = true;
}

int stage_edge_flush(pydem_tile *t)
{
    if (!t->einc_ready) return 0;
    const int n = (int)t->n, m = (int)t->m;
    const int64_t nper = 2 * (int64_t)m + 2 * (int64_t)(n - 2);
    if (t->einc_compact && t->cond_live) {
        // condensed form: the interior catches up with what is done, the remaining inlets let go on the watched graph
        // (nothing becomes 'done' any more), and the interior follows once more
        const double t_flush0 = host_now_ms();
        PYDEM_TRY(cond_catchup(t, 1));
        CondArgsE X;
        PYDEM_TRY(cond_args(t, X));
        X.C.set_done = 0;
        hipLaunchKernelGGL(k_cond_release_todo, dim3((unsigned)cdiv(nper, 128)), dim3(128), 0, t->stream, X);
        hipLaunchKernelGGL(k_cond_run, dim3(1), dim3(COND_THREADS), 0, t->stream, X);
        PYDEM_TRY(cond_catchup(t, 0));
        t->einc_ready = false; t->cond_live = false;
        if (getenv("PYDEM_EDGE_DEBUG")) {
            HIP_TRY(hipStreamSynchronize(t->stream));
            int32_t tot[6];
            HIP_TRY(hipMemcpy(tot, t->cond_cnt, sizeof(tot), hipMemcpyDeviceToHost));
            fprintf(stderr, "condensed edge rounds: flush (interior cascade) %.3f ms; %d rounds ran on the watched graph, %d levels, %d nodes finished\n",
                    host_now_ms() - t_flush0, tot[4], tot[3], tot[5]);
        }
        return 0;
    }
    HIP_TRY(hipMemsetAsync(t->counters, 0, 16 * sizeof(int32_t), t->stream));
    if (t->einc_compact) {
        CIncArgs C;
        PYDEM_TRY(cinc_args(t, C));
        C.set_done = 0;
        hipLaunchKernelGGL(k_cinc_release_todo, dim3((unsigned)cdiv(nper, 128)), dim3(128), 0, t->stream, C, (QE *)t->queue[0], &t->counters[0]);
        PYDEM_TRY(cinc_cascade(t, C, nullptr));
    } else {
        IncArgs E;
        PYDEM_TRY(einc_args(t, E));
        E.set_done = 0;
        hipLaunchKernelGGL(k_einc_release_todo, dim3((unsigned)cdiv(nper, 128)), dim3(128), 0, t->stream, E, (QE *)t->queue[0], &t->counters[0]);
        PYDEM_TRY(einc_cascade(t, E, nullptr));
    }
    t->einc_ready = false;              // counts and deltas are spent: the next incremental round starts from the masks again
    return 0;
}
