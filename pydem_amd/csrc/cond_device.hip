// cond_device.hip -- elevation conditioning on the device, part 1: quantisation artefacts and flats.
//
// Replaces, for a tile whose elevation is resident in HBM (reference creare-com/pydem v1.2.1):
//   pydem/dem_processing.py:396-426   calc_fill_pit_artifacts
//   pydem/dem_processing.py:551-579   calc_fill_flats, with _fill_flat :308-394 and the helpers
//   pydem/utils.py:342-370 get_border_mask, :374-402 get_distance, :450-468 find_centroid
// The reference labels the candidate cells with scipy.ndimage.label and then runs a Python loop over the
// labelled regions, each inside its bounding box grown by one pixel: 155 k regions on a 2048^2 SRTM-like tile,
// a handful of scipy.ndimage calls each.  No region reads what another region writes (`roi` is the unmodified
// surface, `out` only receives the region's own cells, :308-394), so all regions are processed at once:
//   * candidate masks by a 3x3 minimum (scipy.ndimage.minimum_filter with its 'reflect' border = minimum over the
//     neighbours that exist);
//   * regions by the union-find labelling of ccl.h (a component of cells that are <= all their neighbours has ONE
//     elevation, so "the region's level" is the elevation of any of its cells);
//   * per-region facts (size, bounding box, rim tests, lowest uphill rim value, coordinate sums) by atomics on a
//     record per region;
//   * the two (1, sqrt 2) chamfer distances of _fill_flat as Jacobi sweeps over ALL flat cells of all regions,
//     every region frozen at the sweep at which utils.get_distance would have stopped for it -- as soon as all of
//     its cells have SOME finite value, not at convergence (:392-401): the sweep number is part of the result.
// Every floating-point expression keeps numpy's operation order (-ffp-contract=off); the host implementation
// (cond_host.cpp behind pydem_amd/conditioning.py, pinned bit for bit by tests/golden/g5_* and g7_*) is the
// twin the tests compare with.  Tiles with no-data (NaN) cells run here too: the candidate masks replay scipy's ring
// filter value for value where a NaN is near (k_cond_ring_*), everything else compares like numpy (NaN: false).
#include "internal.h"
#include <math.h>

namespace {

#include "ccl.h"

constexpr double SQRT2 = 1.4142135623730951;      // np.sqrt(2.0)

// one list slot per calling lane, one atomic per wavefront
__device__ __forceinline__ int32_t agg_slot_c(int32_t *count)
{
    const unsigned long long bal = __ballot(true);
    const int lane = (int)__lane_id();
    const int leader = __ffsll((long long)bal) - 1;
    int32_t base = 0;
    if (lane == leader) base = atomicAdd(count, (int32_t)__popcll(bal));
    base = __shfl(base, leader);
    return base + __popcll(bal & ((1ull << lane) - 1ull));
}

struct CondArgs {
    int n, m;
    int64_t NN;
    double *elev;            // the surface being conditioned (input of the step)
    double *built;           // output surface of fill_flats
    uint8_t *mask;           // candidate mask of the step
    const int32_t *list;     // compacted mask cells
    const int32_t *count;
    int32_t *labels;         // root cell per mask cell
    int32_t *rid;            // [NN] region index of a ROOT cell
    int32_t *creg;           // [NN] region index of every mask cell (one hop instead of two in the per-cell kernels)
    int f32;                 // the elevations are float32 values: `rim - 1` rounds in float32 (:424)
    int below_sea;
    int nan_tile;            // the tile has no-data cells: a region may then hold cells of DIFFERENT heights (a NaN shields a lower
                             // neighbour from scipy's filter), and "the region's level" is what the reference takes: its first cell
};

// the level the reference works with (:322, :418: `roi[region][0]`, the region's first cell in raster order = the root of the
// min-index union-find).  Without no-data cells every cell of a region has that height and the extra load is skipped.
__device__ __forceinline__ double region_level(const CondArgs &A, int32_t c) { return A.nan_tile ? A.elev[A.labels[c]] : A.elev[c]; }

// one record per region (struct of arrays, sized by the number of mask cells)
struct Regions {
    int32_t *size, *n_edge, *i0, *i1, *j0, *j1, *flags, *centre, *done_hi, *done_lo, *rem_hi, *rem_lo;
    unsigned long long *sum_i, *sum_j, *lowest_bits, *cdist_bits;
    double *top;
};
constexpr int32_t RF_BAD = 1, RF_SOURCE = 2, RF_DRAIN = 4, RF_GENERAL = 8, RF_SRC_CENTRE = 16, RF_DRN_CENTRE = 32, RF_DRN_EDGE = 64,
                  RF_INTERP = 128, RF_NEED_CENTRE = 256;

__device__ __forceinline__ bool sea_ok(double v, int below_sea) { return below_sea ? (v != 0.0) : (v > 0.0); }

// order-preserving map double -> uint64 (for atomicMin on values of either sign)
__device__ __forceinline__ unsigned long long dkey(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double dunkey(unsigned long long k)
{
    const unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)b);
}

// candidate mask: minimum_filter(elev, 3x3) >= elev, and above (or not at) sea level (:410-412, :565-568)
__global__ __launch_bounds__(256) void k_cond_mask(CondArgs A, int corners_off, int32_t *nan_count)
{
    const int n = A.n, m = A.m;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < A.NN; c += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(c / m), j = (int)(c - (int64_t)i * m);
        const double z = A.elev[c];
        if (isnan(z)) { *nan_count = 1; A.mask[c] = 0; continue; }     // (a flag: the host only asks whether any cell is NaN -- no atomic on one address per no-data cell)
        bool low = true;
        for (int di = -1; di <= 1 && low; di++) {
            const int ii = i + di;
            if (ii < 0 || ii >= n) continue;
            for (int dj = -1; dj <= 1; dj++) {
                const int jj = j + dj;
                if (jj < 0 || jj >= m) continue;
                if (A.elev[(int64_t)ii * m + jj] < z) { low = false; break; }
            }
        }
        bool v = low && sea_ok(z, A.below_sea);
        if (corners_off && (i == 0 || i == n - 1) && (j == 0 || j == m - 1)) v = false;          // :569-572
        A.mask[c] = v;
    }
}

// ---- no-data (NaN) cells: scipy's minimum_filter, exactly ---------------------------------------------------
// scipy.ndimage.minimum_filter(x, (3, 3)) is separable: NI_MinOrMaxFilter1D along axis 0, then along axis 1 on the result,
// and that 1-D filter is a ring of (value, death) pairs (ni_filters.c): a new value that is <= the front REPLACES the
// whole ring, otherwise it removes the entries >= itself from the back and is appended; the front is reported and leaves
// after three steps.  Without NaN that is the running minimum; with NaN every comparison is false: a NaN is appended,
// shields the entries in front of it from later (smaller) values and is reported itself while it is the front.  The masks
// `minimum_filter(z) >= z` of the reference (:410-412, :565-568) depend on it, so on tiles with no-data cells the two
// passes are replayed value for value (checked against scipy on random arrays: oracle/ref_harness / tests).  A clean
// window needs no replay: three values without NaN -> their minimum, three NaN -> NaN.  Otherwise the ring is replayed
// from the last point where its state is known: the line start, or three consecutive values of one kind (the ring then
// holds just what those three leave).  RING_BACK bounds the look-back; a line that alternates longer than that sets the
// give-up flag (the host path takes the tile).
constexpr int RING_BACK = 48;

template <typename Get>
__device__ double ring3_at(Get get, int p, int L, int32_t *giveup)
{
    // stream index ll = 0 .. L + 1, value pv(ll) = x[clamp(ll - 1)] ('reflect' with one element on either side); the
    // output of position p is the front after step ll = p + 2
    auto pv = [&](int ll) -> double { const int q = ll - 1; return get(q < 0 ? 0 : (q >= L ? L - 1 : q)); };
    const double a = pv(p), b = pv(p + 1), c = pv(p + 2);
    const bool na = a != a, nb = b != b, nc = c != c;
    if (!na && !nb && !nc) { double mn = a < b ? a : b; return c < mn ? c : mn; }
    if (na && nb && nc) return a;
    // ---- find the start: the latest s <= p + 1 with pv(s - 2 .. s) all of one kind, or the stream start
    int s = -1;
    for (int e = p + 1; e >= 2 && e >= p + 2 - RING_BACK; e--) {
        const double u = pv(e - 2), v = pv(e - 1), w = pv(e);
        const bool nu = u != u, nv = v != v, nw = w != w;
        if (nu == nv && nv == nw) { s = e; break; }
    }
    int first;
    if (s < 0) {
        if (p + 2 - RING_BACK > 0) { *giveup = 1; return a; }
        first = 0;                                   // from the stream start
    } else first = s - 2;
    // ---- replay: ring as arrays (at most 3 live entries)
    double rv[4]; int rd[4]; int cnt = 1;
    rv[0] = pv(first); rd[0] = first + 3;
    for (int ll = first + 1; ll <= p + 2; ll++) {
        const double val = pv(ll);
        if (rd[0] == ll) { for (int k = 1; k < cnt; k++) { rv[k - 1] = rv[k]; rd[k - 1] = rd[k]; } cnt--; }
        if (cnt == 0 || val <= rv[0]) { rv[0] = val; rd[0] = ll + 3; cnt = 1; }
        else {
            while (cnt > 0 && rv[cnt - 1] >= val) cnt--;
            rv[cnt] = val; rd[cnt] = ll + 3; cnt++;
        }
    }
    return rv[0];
}

// pass 1 (axis 0): tmp[i][j] = ring filter of column j at i
__global__ __launch_bounds__(256) void k_cond_ring_cols(CondArgs A, double *__restrict__ tmp, int32_t *giveup)
{
    const int n = A.n, m = A.m;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < A.NN; c += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(c / m), j = (int)(c - (int64_t)i * m);
        const double *col = A.elev + j;
        tmp[c] = ring3_at([&](int q) { return col[(int64_t)q * m]; }, i, n, giveup);
    }
}

// pass 2 (axis 1) + the mask: minimum_filter(elev, 3x3) >= elev, above (or not at) sea level
__global__ __launch_bounds__(256) void k_cond_ring_rows_mask(CondArgs A, const double *__restrict__ tmp, int corners_off, int32_t *giveup)
{
    const int n = A.n, m = A.m;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < A.NN; c += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(c / m), j = (int)(c - (int64_t)i * m);
        const double z = A.elev[c];
        const double *row = tmp + (int64_t)i * m;
        const double mn = ring3_at([&](int q) { return row[q]; }, j, m, giveup);
        bool v = (mn >= z) && sea_ok(z, A.below_sea);
        if (corners_off && (i == 0 || i == n - 1) && (j == 0 || j == m - 1)) v = false;
        A.mask[c] = v;
    }
}

// region numbers for the roots (any order): 8 list entries per thread, ONE add to the region counter per workgroup trip --
// the counter is a single address, and one add per wavefront of roots kept its L2 channel busy for the whole kernel
// (140 k adds of ~10 ns: 1.4 ms for the 8.9 M flat cells of the 8192^2 SRTM-like tile)
__global__ __launch_bounds__(256) void k_region_index(CondArgs A, int32_t *nreg)
{
    constexpr int PER = 8;
    __shared__ int32_t wave_tot[4];
    __shared__ int32_t blk_base;
    const int32_t nf = *A.count;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t base = (int64_t)blockIdx.x * (256 * PER); base < nf; base += (int64_t)gridDim.x * (256 * PER)) {
        int32_t cell[PER];
        uint32_t root = 0;
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const int64_t q = base + j * 256 + threadIdx.x;
            cell[j] = q < nf ? A.list[q] : -1;
        }
#pragma unroll
        for (int j = 0; j < PER; j++)
            if (cell[j] >= 0 && A.labels[cell[j]] == cell[j]) root |= 1u << j;
        const int32_t mine = __popc(root);
        int32_t incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int32_t o = __shfl_up(incl, off); if (lane >= off) incl += o; }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int32_t tot = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
            blk_base = tot ? atomicAdd(nreg, tot) : 0;
        }
        __syncthreads();
        int32_t o = blk_base + incl - mine;
        for (int k = 0; k < wave; k++) o += wave_tot[k];
#pragma unroll
        for (int j = 0; j < PER; j++)
            if (root & (1u << j)) A.rid[cell[j]] = o++;
        __syncthreads();
    }
}

__global__ void k_cell_region(CondArgs A)
{
    const int32_t nf = *A.count;
    for (int32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nf; q += gridDim.x * blockDim.x) {
        const int32_t c = A.list[q];
        A.creg[c] = A.rid[A.labels[c]];
    }
}

__global__ void k_region_init(Regions R, int32_t nreg, int n, int m)
{
    for (int32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < nreg; r += gridDim.x * blockDim.x) {
        R.size[r] = 0; R.n_edge[r] = 0; R.i0[r] = n; R.i1[r] = -1; R.j0[r] = m; R.j1[r] = -1; R.flags[r] = 0; R.centre[r] = 0x7FFFFFFF;
        R.done_hi[r] = 0x7FFFFFFF; R.done_lo[r] = 0x7FFFFFFF; R.rem_hi[r] = 0; R.rem_lo[r] = 0;
        R.sum_i[r] = 0; R.sum_j[r] = 0; R.lowest_bits[r] = ~0ull; R.cdist_bits[r] = ~0ull; R.top[r] = 0.0;
    }
}


// ---- per-region accumulation.  A lake puts millions of consecutive list entries on one region record; 64 atomics of a
// wavefront on the same address are 64 serial round trips.  When every entry of a wavefront belongs to the same region
// (the loops below keep all 64 lanes in step, `valid` marks the entries past the end) the lanes combine first and lane 0
// speaks for all; mixed wavefronts (the small regions) keep their per-lane atomics.
__device__ __forceinline__ int wave_min(int v) { for (int o = 32; o; o >>= 1) { const int w = __shfl_xor(v, o); v = w < v ? w : v; } return v; }
__device__ __forceinline__ int wave_max(int v) { for (int o = 32; o; o >>= 1) { const int w = __shfl_xor(v, o); v = w > v ? w : v; } return v; }
__device__ __forceinline__ int wave_or(int v) { for (int o = 32; o; o >>= 1) v |= __shfl_xor(v, o); return v; }
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) { for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o); return v; }
__device__ __forceinline__ unsigned long long wave_min(unsigned long long v)
{
    for (int o = 32; o; o >>= 1) { const unsigned long long w = __shfl_xor(v, o); v = w < v ? w : v; }
    return v;
}
// all valid entries of the wavefront carry region r0 (lane 0 is valid whenever any lane is: the entries are consecutive)
__device__ __forceinline__ bool one_region(bool valid, int32_t r, int32_t &r0)
{
    r0 = __shfl(r, 0);
    return __ballot(valid && r != r0) == 0ull;
}

// Mixed wavefronts: the list is in raster order, so the entries of one region that a wavefront holds sit next to each other
// in RUNS (a row segment of the region).  A segmented reduction over the runs leaves every run's total in its first lane,
// and only those lanes talk to the region records: the small regions of an integer surface are 3-4 cells, i.e. two or
// three atomics saved out of every three or four, ten record fields each (k_flat_scan: 89 M atomics for 8.9 M flat cells).
struct Runs { unsigned long long heads; bool head; };
__device__ __forceinline__ Runs wave_runs(bool valid, int32_t r)
{
    const int lane = (int)__lane_id();
    const int32_t key = valid ? r : -1;
    const int32_t prev = __shfl_up(key, 1);
    Runs q;
    q.head = lane == 0 || prev != key;
    q.heads = __ballot(q.head);
    q.head = q.head && valid;
    return q;
}
template <typename T, typename Op>
__device__ __forceinline__ T run_reduce(T v, const Runs &q, Op op)
{
    const int lane = (int)__lane_id();
    const unsigned long long after = lane < 63 ? q.heads >> (lane + 1) : 0ull;      // bit b: lane + 1 + b starts a run
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const T o = __shfl_down(v, off);
        if (lane + off < 64 && (after & ((1ull << off) - 1ull)) == 0ull) v = op(v, o);
    }
    return v;
}
struct OpAdd { template <typename T> __device__ T operator()(T a, T b) const { return a + b; } };
struct OpMin { template <typename T> __device__ T operator()(T a, T b) const { return b < a ? b : a; } };
struct OpMax { template <typename T> __device__ T operator()(T a, T b) const { return b > a ? b : a; } };
struct OpOr { template <typename T> __device__ T operator()(T a, T b) const { return a | b; } };

// ---- quantisation artefacts (:396-426) ---------------------------------------------------------------------
// a region is raised by one unit when it is small, lies strictly inside the array and its whole rim is exactly
// one unit higher
__global__ void k_art_scan(CondArgs A, Regions R, double max_area)
{
    const int32_t nf = *A.count;
    const int n = A.n, m = A.m;
    for (int32_t q = blockIdx.x * blockDim.x + threadIdx.x;; q += gridDim.x * blockDim.x) {
        const bool valid = q < nf;
        if (!__ballot(valid)) break;
        int32_t r = -1;
        bool bad = false;
        if (valid) {
            const int32_t c = A.list[q];
            r = A.creg[c];
            const int i = c / m, j = c - i * m;
            bad = (i == 0 || j == 0 || i == n - 1 || j == m - 1);                 // the one-pixel rim must lie inside (:414-415)
            const double level = region_level(A, c);
            if (!bad) {
                for (int d = 0; d < 9; d++) {
                    if (d == 4) continue;
                    const int32_t nb = c + (d / 3 - 1) * m + (d % 3 - 1);
                    if (A.mask[nb]) continue;                                     // same region
                    const double v = A.elev[nb];
                    const bool ok = A.f32 ? ((float)v - 1.0f == (float)level) : (v - 1 == level);    // :424
                    if (!ok) { bad = true; break; }
                }
            }
        }
        const Runs rq = wave_runs(valid, r);
        const int cnt = run_reduce(valid ? 1 : 0, rq, OpAdd()), anybad = run_reduce(bad ? 1 : 0, rq, OpOr());
        if (rq.head) {
            atomicAdd(&R.size[r], cnt);
            if (anybad) atomicOr(&R.flags[r], RF_BAD);
        }
    }
}

__global__ void k_art_apply(CondArgs A, Regions R, double max_area)
{
    const int32_t nf = *A.count;
    for (int32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nf; q += gridDim.x * blockDim.x) {
        const int32_t c = A.list[q];
        const int32_t r = A.creg[c];
        if ((R.flags[r] & RF_BAD) || (double)R.size[r] > max_area) continue;     // :419-420
        const double v = A.elev[c];
        A.elev[c] = A.f32 ? (double)((float)v + 1.0f) : v + 1;                   // :425 (in the array's dtype)
    }
}

// ---- flats (:551-579, _fill_flat :308-394) -------------------------------------------------------------------
// facts about every region: size, bounding box, cells on the tile edge, coordinate sums, and what its rim offers
// (a cell of the rim is 8-adjacent to the region and not part of it: get_border_mask, utils.py:342-370)
__global__ void k_flat_scan(CondArgs A, Regions R)
{
    const int32_t nf = *A.count;
    const int n = A.n, m = A.m;
    for (int32_t q = blockIdx.x * blockDim.x + threadIdx.x;; q += gridDim.x * blockDim.x) {
        const bool valid = q < nf;
        if (!__ballot(valid)) break;
        int32_t r = -1, fl = 0;
        int i = 0, j = 0;
        bool edge = false;
        unsigned long long low = ~0ull;
        if (valid) {
            const int32_t c = A.list[q];
            r = A.creg[c];
            i = c / m; j = c - i * m;
            edge = (i == 0 || j == 0 || i == n - 1 || j == m - 1);
            const double level = region_level(A, c);
            for (int d = 0; d < 9; d++) {
                if (d == 4) continue;
                const int ii = i + d / 3 - 1, jj = j + d % 3 - 1;
                if (ii < 0 || ii >= n || jj < 0 || jj >= m) continue;
                const int32_t nb = ii * m + jj;
                if (A.mask[nb]) continue;
                const double v = A.elev[nb];
                if (v == level) fl |= RF_DRAIN;                                  // :337
                else if (v > level) { fl |= RF_SOURCE; const unsigned long long k = dkey(v); low = k < low ? k : low; }   // :338, :344
            }
        }
        const Runs rq = wave_runs(valid, r);
        const int cnt = run_reduce(valid ? 1 : 0, rq, OpAdd()), n_edge = run_reduce(edge ? 1 : 0, rq, OpAdd());
        const int i0 = run_reduce(i, rq, OpMin()), i1 = run_reduce(i, rq, OpMax()), j0 = run_reduce(j, rq, OpMin()), j1 = run_reduce(j, rq, OpMax());
        const int si = run_reduce(i, rq, OpAdd()), sj = run_reduce(j, rq, OpAdd());       // (at most 64 coordinates below 2^24 each)
        const unsigned long long lw = run_reduce(low, rq, OpMin());
        const int flw = run_reduce(fl, rq, OpOr());
        if (rq.head) {
            atomicAdd(&R.size[r], cnt);
            atomicMin(&R.i0[r], i0); atomicMax(&R.i1[r], i1); atomicMin(&R.j0[r], j0); atomicMax(&R.j1[r], j1);
            if (n_edge) atomicAdd(&R.n_edge[r], n_edge);
            atomicAdd(&R.sum_i[r], (unsigned long long)si); atomicAdd(&R.sum_j[r], (unsigned long long)sj);
            if (lw != ~0ull) atomicMin(&R.lowest_bits[r], lw);
            if (flw) atomicOr(&R.flags[r], flw);
        }
    }
}

// window of a region: its bounding box grown by one pixel inside the array (:575-577)
__device__ __forceinline__ void region_window(const Regions &R, int32_t r, int n, int m, int &wi0, int &wi1, int &wj0, int &wj1)
{
    wi0 = R.i0[r] > 0 ? R.i0[r] - 1 : 0; wi1 = R.i1[r] + 2 < n ? R.i1[r] + 2 : n;
    wj0 = R.j0[r] > 0 ? R.j0[r] - 1 : 0; wj1 = R.j1[r] + 2 < m ? R.j1[r] + 2 : m;
}

// the decisions of _fill_flat that need the whole region (:340-371); single pixels are done on the spot (:312-325)
__global__ void k_flat_plan(CondArgs A, Regions R, int32_t nreg, const int32_t *root_of, double source_tol, int peaks, int pits)
{
    const int n = A.n, m = A.m;
    for (int32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < nreg; r += gridDim.x * blockDim.x) {
        const int32_t root = root_of[r];
        const double level = A.elev[root];
        int wi0, wi1, wj0, wj1;
        region_window(R, r, n, m, wi0, wi1, wj0, wj1);
        const int64_t wsize = (int64_t)(wi1 - wi0) * (wj1 - wj0);
        int32_t fl = R.flags[r];
        if (wsize <= 9 && R.size[r] == 1) {
            // "a single pixel": raise it towards its lowest higher neighbour, unless every neighbour is higher (:312-325)
            int n_high = 0; double low_high = INFINITY;
            for (int ii = wi0; ii < wi1; ii++)
                for (int jj = wj0; jj < wj1; jj++) {
                    const double v = A.elev[(int64_t)ii * m + jj];
                    if (v > level) { n_high++; low_high = v < low_high ? v : low_high; }
                }
            if (n_high == wsize - 1) { }
            else if (n_high > 0) { const double d = low_high - level; A.built[root] = A.built[root] + ((d < 1.0 ? d : 1.0) - 0.01); }
            else if (peaks) A.built[root] = A.built[root] + 0.5;
            continue;
        }
        bool go = true;
        if (fl & RF_SOURCE) {
            const double lowest = dunkey(R.lowest_bits[r]);
            R.top[r] = (level + 1.0 < lowest) ? level + 1.0 : lowest;            // min(level + 1.0, lowest) :346
        } else if (peaks) {
            R.top[r] = level + 0.5;                                               // :349
            fl |= RF_SRC_CENTRE | RF_NEED_CENTRE;
        } else go = false;
        if (go) {
            if (fl & RF_DRAIN) { }
            else if (R.n_edge[r] > 0) {                                           // :362-366
                fl |= RF_DRN_EDGE;
                if (R.n_edge[r] == R.size[r]) go = false;
            } else if (pits) fl |= RF_DRN_CENTRE | RF_NEED_CENTRE;                // :367-371
            else go = false;
        }
        if (go) fl |= RF_GENERAL | RF_INTERP;
        R.flags[r] = fl;
    }
}

// centre cell of the regions that need one (find_centroid, utils.py:450-468): the region cell closest to the
// centre of mass, first in raster order among equals.  Coordinates are window coordinates like the reference's.
__device__ __forceinline__ double centre_dist(const Regions &R, int32_t r, int i, int j, int n, int m)
{
    int wi0, wi1, wj0, wj1;
    region_window(R, r, n, m, wi0, wi1, wj0, wj1);
    const double cnt = (double)R.size[r];
    const double cy = (double)(long long)(R.sum_i[r] - (unsigned long long)R.size[r] * (unsigned long long)wi0) / cnt;
    const double cx = (double)(long long)(R.sum_j[r] - (unsigned long long)R.size[r] * (unsigned long long)wj0) / cnt;
    const double dy = (double)(i - wi0) - cy, dx = (double)(j - wj0) - cx;
    return sqrt(dy * dy + dx * dx);
}

__global__ void k_centre_pass(CondArgs A, Regions R, int pass)
{
    const int32_t nf = *A.count;
    const int n = A.n, m = A.m;
    for (int32_t q = blockIdx.x * blockDim.x + threadIdx.x;; q += gridDim.x * blockDim.x) {
        const bool valid = q < nf;
        if (!__ballot(valid)) break;
        int32_t r = -1, c = 0x7FFFFFFF;
        unsigned long long k = ~0ull;
        bool want = false;
        if (valid) {
            c = A.list[q];
            r = A.creg[c];
            want = (R.flags[r] & RF_NEED_CENTRE) != 0;
            if (want) { const int i = c / m, j = c - i * m; k = dkey(centre_dist(R, r, i, j, n, m)); }
        }
        if (!__ballot(want)) continue;
        const Runs rq = wave_runs(valid, r);
        if (pass == 0) {
            const unsigned long long kw = run_reduce(want ? k : ~0ull, rq, OpMin());
            if (rq.head && kw != ~0ull) atomicMin(&R.cdist_bits[r], kw);
        } else {
            const int cw = run_reduce((want && k == R.cdist_bits[r]) ? c : 0x7FFFFFFF, rq, OpMin());
            if (rq.head && cw != 0x7FFFFFFF) atomicMin(&R.centre[r], cw);
        }
    }
}

// seeds inside the regions and the number of cells that still wait for a distance
__global__ void k_flat_seed(CondArgs A, Regions R, double *dh, double *dl)
{
    const int32_t nf = *A.count;
    const int n = A.n, m = A.m;
    for (int32_t q = blockIdx.x * blockDim.x + threadIdx.x;; q += gridDim.x * blockDim.x) {
        const bool valid = q < nf;
        if (!__ballot(valid)) break;
        int32_t r = -1;
        bool wait_hi = false, wait_lo = false;
        if (valid) {
            const int32_t c = A.list[q];
            r = A.creg[c];
            const int32_t fl = R.flags[r];
            const int i = c / m, j = c - i * m;
            const bool on_edge = (i == 0 || j == 0 || i == n - 1 || j == m - 1);
            const bool is_centre = R.centre[r] == c;
            if ((fl & RF_SRC_CENTRE) && is_centre) A.built[c] = R.top[r];        // out[ci] = top, whatever follows (:351)
            if (fl & RF_GENERAL) {
                const bool seed_hi = (fl & RF_SRC_CENTRE) && is_centre;
                const bool seed_lo = ((fl & RF_DRN_EDGE) && on_edge) || ((fl & RF_DRN_CENTRE) && is_centre);
                dh[c] = seed_hi ? 0.0 : INFINITY;
                dl[c] = seed_lo ? 0.0 : INFINITY;
                wait_hi = !seed_hi; wait_lo = !seed_lo;
            }
        }
        if (!(__ballot(wait_hi) | __ballot(wait_lo))) continue;
        const Runs rq = wave_runs(valid, r);
        const int nh = run_reduce(wait_hi ? 1 : 0, rq, OpAdd()), nl = run_reduce(wait_lo ? 1 : 0, rq, OpAdd());
        if (rq.head) {
            if (nh) atomicAdd(&R.rem_hi[r], nh);
            if (nl) atomicAdd(&R.rem_lo[r], nl);
        }
    }
}

__global__ void k_flat_seed_done(Regions R, int32_t nreg)
{
    for (int32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < nreg; r += gridDim.x * blockDim.x) {
        if (!(R.flags[r] & RF_GENERAL)) continue;
        if (R.rem_hi[r] == 0) R.done_hi[r] = 0;
        if (R.rem_lo[r] == 0) R.done_lo[r] = 0;
    }
}

// "This cell got its first finite value": count-down of the region's open cells, the sweep that brings it to zero stops
// the region.  A front that crosses a lake is thousands of cells of ONE region per sweep -- one atomic each on the same
// address kept its L2 channel busy for the whole sweep (~10 ns per add: 30-50 us for a front of 5000 cells).  The lanes
// of a wavefront that report for the same region subtract once.  Call from the converged part of the caller's loop body.
__device__ __forceinline__ void region_arrivals(int32_t *rem, int32_t *done, int32_t r, bool arrived, int sweep)
{
    unsigned long long pend = __ballot(arrived);
    while (pend) {
        const int leader = __ffsll((long long)pend) - 1;
        const int32_t lr = __shfl(r, leader);
        const unsigned long long same = __ballot(arrived && r == lr);
        if ((int)__lane_id() == leader) {
            const int k = __popcll(same);
            if (atomicSub(&rem[lr], k) == k) done[lr] = sweep;
        }
        pend &= ~same;
    }
}

// One Jacobi sweep of utils.get_distance (:392-401) for both distances of every region that has not stopped yet:
//   d = min(d, min(d over the cell and its 4 cardinal neighbours) + 1, min(d over the 3x3) + sqrt 2)
// with d = 0 on the seeds of the rim and "no value yet" everywhere else outside the region.  A region stops
// after the sweep in which its last cell got a value (done_* = number of that sweep).
__global__ __launch_bounds__(256) void k_flat_sweep(CondArgs A, Regions R, const int32_t *__restrict__ alist, int32_t na,
                                                    const double *__restrict__ dh0, double *__restrict__ dh1,
                                                    const double *__restrict__ dl0, double *__restrict__ dl1, int sweep, double source_tol)
{
    const int n = A.n, m = A.m;
    for (int32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < na; q += gridDim.x * blockDim.x) {
        const int32_t c = alist[q];
        const int32_t r = A.creg[c];
        const int32_t fl = R.flags[r];
        const bool act_hi = R.done_hi[r] >= sweep, act_lo = R.done_lo[r] >= sweep;
        const double oh = dh0[c], ol = dl0[c];
        double nh = oh, nl = ol;
        if (act_hi || act_lo) {
            const int i = c / m, j = c - i * m;
            const double level = region_level(A, c);
            const double src_max = (fl & RF_SOURCE) ? dunkey(R.lowest_bits[r]) + source_tol : 0.0;   // lowest + tol (:347)
            double card_h = oh, all_h = oh, card_l = ol, all_l = ol;
            for (int d = 0; d < 9; d++) {
                if (d == 4) continue;
                const int ii = i + d / 3 - 1, jj = j + d % 3 - 1;
                if (ii < 0 || ii >= n || jj < 0 || jj >= m) continue;
                const int32_t nb = ii * m + jj;
                double vh, vl;
                if (A.mask[nb]) { vh = dh0[nb]; vl = dl0[nb]; }
                else {
                    const double z = A.elev[nb];
                    vh = ((fl & RF_SOURCE) && z > level && z <= src_max) ? 0.0 : INFINITY;
                    vl = ((fl & RF_DRAIN) && z == level) ? 0.0 : INFINITY;
                }
                const bool cardinal = (d == 1 || d == 3 || d == 5 || d == 7);
                if (cardinal) { card_h = vh < card_h ? vh : card_h; card_l = vl < card_l ? vl : card_l; }
                all_h = vh < all_h ? vh : all_h; all_l = vl < all_l ? vl : all_l;
            }
            if (act_hi) {
                const double s = card_h + 1, g = all_h + SQRT2;
                const double best = s < g ? s : g;
                nh = best < oh ? best : oh;
            }
            if (act_lo) {
                const double s = card_l + 1, g = all_l + SQRT2;
                const double best = s < g ? s : g;
                nl = best < ol ? best : ol;
            }
        }
        region_arrivals(R.rem_hi, R.done_hi, r, act_hi && isinf(oh) && !isinf(nh), sweep);
        region_arrivals(R.rem_lo, R.done_lo, r, act_lo && isinf(ol) && !isinf(nl), sweep);
        dh1[c] = nh; dl1[c] = nl;
    }
}

// The same sweep over a WORK LIST: behind the front of first arrivals a region's distances settle after a few
// sweeps, and a lake that takes 1500 sweeps to cross has millions of settled cells.  A cell is looked at in sweep s + 1
// only if it or one of its region neighbours changed in sweep s (a cell that changed is listed itself, so that its new
// value reaches the other buffer of the ping-pong pair); cells that are not listed have the same value in both buffers.
// (`wl_lds` / `lds_cap`: the single-workgroup kernel keeps the first lds_cap entries of the next list in LDS and only what does not
// fit there in the global list; the multi-workgroup kernel passes lds_cap = 0)
__device__ __forceinline__ void flat_sweep_entry(const CondArgs &A, const Regions &R, int32_t c, int32_t *__restrict__ wl_out, int32_t *n_out,
                                                 int32_t *stamp, const double *__restrict__ dh0, double *__restrict__ dh1,
                                                 const double *__restrict__ dl0, double *__restrict__ dl1, int sweep, double source_tol,
                                                 int32_t *wl_lds = nullptr, int32_t lds_cap = 0)
{
    const int n = A.n, m = A.m;
    const int i = c / m, j = c - i * m;
    // all loads of the 8 neighbours in one batch (mask, both distances, elevation), issued BEFORE the region record is known
    // (its index is a load of its own): the sweeps are chains of dependent round trips, a few thousand cells per sweep, and
    // nearly every listed cell belongs to a region that is still sweeping
    uint8_t nmask[9]; double ndh[9], ndl[9], nz[9]; bool inb[9];
#pragma unroll
    for (int d = 0; d < 9; d++) {
        const int ii = i + d / 3 - 1, jj = j + d % 3 - 1;
        inb[d] = d != 4 && ii >= 0 && ii < n && jj >= 0 && jj < m;
        const int32_t nb = inb[d] ? ii * m + jj : c;
        nmask[d] = A.mask[nb]; ndh[d] = dh0[nb]; ndl[d] = dl0[nb]; nz[d] = A.elev[nb];
    }
    const int32_t r = A.creg[c];
    const int32_t fl = R.flags[r];
    const bool act_hi = R.done_hi[r] >= sweep, act_lo = R.done_lo[r] >= sweep;
    const unsigned long long lowest_bits = R.lowest_bits[r];
    const double oh = dh0[c], ol = dl0[c];
    double nh = oh, nl = ol;
    if (act_hi || act_lo) {
        const double level = region_level(A, c);
        const double src_max = (fl & RF_SOURCE) ? dunkey(lowest_bits) + source_tol : 0.0;   // lowest + tol (:347)
        double card_h = oh, all_h = oh, card_l = ol, all_l = ol;
#pragma unroll
        for (int d = 0; d < 9; d++) {
            if (!inb[d]) continue;
            double vh, vl;
            if (nmask[d]) { vh = ndh[d]; vl = ndl[d]; }
            else {
                const double z = nz[d];
                vh = ((fl & RF_SOURCE) && z > level && z <= src_max) ? 0.0 : INFINITY;
                vl = ((fl & RF_DRAIN) && z == level) ? 0.0 : INFINITY;
            }
            const bool cardinal = (d == 1 || d == 3 || d == 5 || d == 7);
            if (cardinal) { card_h = vh < card_h ? vh : card_h; card_l = vl < card_l ? vl : card_l; }
            all_h = vh < all_h ? vh : all_h; all_l = vl < all_l ? vl : all_l;
        }
        if (act_hi) {
            const double sv = card_h + 1, g = all_h + SQRT2;
            const double best = sv < g ? sv : g;
            nh = best < oh ? best : oh;
        }
        if (act_lo) {
            const double sv = card_l + 1, g = all_l + SQRT2;
            const double best = sv < g ? sv : g;
            nl = best < ol ? best : ol;
        }
    }
    region_arrivals(R.rem_hi, R.done_hi, r, act_hi && isinf(oh) && !isinf(nh), sweep);
    region_arrivals(R.rem_lo, R.done_lo, r, act_lo && isinf(ol) && !isinf(nl), sweep);
    dh1[c] = nh; dl1[c] = nl;
    int32_t nbs[9];
    bool take[9];
    int cnt = 0;
#pragma unroll
    for (int d = 0; d < 9; d++) take[d] = false;
    if (nh != oh || nl != ol) {
        // the cell and its region neighbours are looked at in the next sweep: all nine stamps travel together
#pragma unroll
        for (int d = 0; d < 9; d++) {
            const int ii = i + d / 3 - 1, jj = j + d % 3 - 1;
            nbs[d] = ii * m + jj;
            take[d] = ii >= 0 && ii < n && jj >= 0 && jj < m && A.mask[nbs[d]];
        }
#pragma unroll
        for (int d = 0; d < 9; d++) take[d] = take[d] && atomicExch(&stamp[nbs[d]], sweep + 1) != sweep + 1;
#pragma unroll
        for (int d = 0; d < 9; d++) cnt += take[d] ? 1 : 0;
    }
    // list slots: ONE atomic per wavefront (the counter is a single address: a few thousand changed cells adding to it one
    // by one keep its L2 channel busy for ~10 ns each); the lanes that are in this call -- the others sit out the caller's
    // loop -- share their counts (0..9) bit by bit through ballots
    {
        const unsigned long long act = __ballot(true);
        const unsigned long long b0 = __ballot((cnt & 1) != 0), b1 = __ballot((cnt & 2) != 0), b2 = __ballot((cnt & 4) != 0), b3 = __ballot((cnt & 8) != 0);
        if (b0 | b1 | b2 | b3) {
            const int lane = (int)__lane_id();
            const unsigned long long lt = (1ull << lane) - 1ull;
            const int excl = __popcll(b0 & lt) + 2 * __popcll(b1 & lt) + 4 * __popcll(b2 & lt) + 8 * __popcll(b3 & lt);
            const int tot = __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2) + 8 * __popcll(b3);
            const int leader = __ffsll((long long)act) - 1;
            int32_t base = 0;
            if (lane == leader) base = atomicAdd(n_out, tot);
            base = __shfl(base, leader);
            int32_t at = base + excl;
#pragma unroll
            for (int d = 0; d < 9; d++)
                if (take[d]) { if (at < lds_cap) wl_lds[at] = nbs[d]; else wl_out[at] = nbs[d]; at++; }
        }
    }
}

__global__ __launch_bounds__(256) void k_flat_sweep_wl(CondArgs A, Regions R, const int32_t *__restrict__ wl_in, int32_t *__restrict__ wl_out,
                                                       int32_t *cnt3, int32_t *stamp, const double *__restrict__ dh0, double *__restrict__ dh1,
                                                       const double *__restrict__ dl0, double *__restrict__ dl1, int sweep, double source_tol)
{
    const int32_t na = cnt3[sweep % 3];
    int32_t *n_out = &cnt3[(sweep + 1) % 3];
    if (blockIdx.x == 0 && threadIdx.x == 0) cnt3[(sweep + 2) % 3] = 0;
    for (int32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < na; q += gridDim.x * blockDim.x)
        flat_sweep_entry(A, R, wl_in[q], wl_out, n_out, stamp, dh0, dh1, dl0, dl1, sweep, source_tol);
}

// The tail of the sweeps -- the front of first arrivals crossing a large lake, a few hundred cells per sweep for a
// thousand sweeps and more -- is nothing but kernel boundaries when every sweep is a launch.  One workgroup runs
// `nsweeps` sweeps in a row: the lists, distances and region records it touches were written by itself (visible
// through the CU's L1 after a workgroup barrier); the list counters and stamps are only ever touched by atomics.
constexpr int FS_CAP = 12288;        // entries of a work list kept in LDS by the single-workgroup sweeps (two lists: 96 KB)
__global__ __launch_bounds__(1024) void k_flat_sweep_small(CondArgs A, Regions R, int32_t *al0, int32_t *al1, int32_t *cnt3, int32_t *stamp,
                                                           double *dhA, double *dhB, double *dlA, double *dlB, int sweep0, int nsweeps,
                                                           double source_tol, int32_t *sweeps_done)
{
    // The work lists live in LDS: a sweep of the tail is a chain of dependent round trips (list length, list entry, region
    // record, neighbours, stamps, list append); with the lists and their counters on chip three of them are gone.  Entries
    // beyond FS_CAP go to the global list; if that happens the kernel completes the sweep, writes the LDS part behind them
    // and hands back to the multi-workgroup kernel.
    __shared__ int32_t s_list[2][FS_CAP];
    __shared__ int32_t s_n[3];       // list lengths, rotating: [ci] the list being read, [ci + 1] the one being written, [ci + 2] cleared
    const int t = threadIdx.x;
    int32_t na = cnt3[sweep0 % 3];
    {
        const int32_t *wl_in = ((sweep0 - 1) & 1) ? al1 : al0;
        for (int32_t q = t; q < na; q += blockDim.x) s_list[0][q] = wl_in[q];      // (the host enters with na <= small_cap <= FS_CAP)
        if (t < 3) s_n[t] = 0;
    }
    __syncthreads();
    int in = 0, ci = 0, s = sweep0;
    for (; s < sweep0 + nsweeps; s++) {
        const int cur = (s - 1) & 1;
        int32_t *wl_out = cur ? al0 : al1;
        const double *dh0 = cur ? dhB : dhA, *dl0 = cur ? dlB : dlA;
        double *dh1 = cur ? dhA : dhB, *dl1 = cur ? dlA : dlB;
        for (int32_t q = t; q < na; q += blockDim.x)
            flat_sweep_entry(A, R, s_list[in][q], wl_out, &s_n[(ci + 1) % 3], stamp, dh0, dh1, dl0, dl1, s, source_tol, s_list[in ^ 1], FS_CAP);
        // the counter the NEXT sweep appends to is cleared before the barrier: it was last read (as `na`) two barriers ago, and
        // nobody adds to it in this sweep (a store after the barrier would race with the next sweep's LDS atomics)
        if (t == 0) s_n[(ci + 2) % 3] = 0;
        __syncthreads();
        ci = (ci + 1) % 3;
        na = s_n[ci];
        in ^= 1;
        if (na > FS_CAP) { s++; break; }            // the next list continues in global memory: back to the launches per sweep
    }
    // the list of sweep `s` (the next one to run) goes back to global memory with its length; the counter of the sweep after it is zero
    {
        int32_t *wl = ((s - 1) & 1) ? al1 : al0;
        const int32_t keep = na < FS_CAP ? na : FS_CAP;
        for (int32_t q = t; q < keep; q += blockDim.x) wl[q] = s_list[in][q];
        if (t == 0) { cnt3[s % 3] = na; cnt3[(s + 1) % 3] = 0; cnt3[(s + 2) % 3] = 0; *sweeps_done = s - sweep0; }
    }
}

// Between the two: lists of a few thousand to a few ten thousand entries for hundreds of sweeps (the fronts crossing the
// big plateaus of an integer surface).  A launch per sweep costs ~48 us there, the single workgroup ~27 us (four entries
// per thread, one after the other).  Here a few dozen workgroups stay resident and separate the sweeps by a barrier of
// their own: an arrival counter in global memory, agent-scope release / acquire around it.  `xcd_only`: only the
// workgroups the dispatcher places on one XCD take part (workgroup b goes to XCD b % 8 -- the others exit at once), so
// that lists, distances and the counter stay in ONE L2; correctness does not depend on the placement.
__device__ __forceinline__ void coop_barrier(int32_t *bar, int nwg, int &phase)
{
    // EVERY thread releases its own stores at agent scope before the workgroup barrier: a workgroup-scope barrier does not
    // wait for the other wavefronts' global stores (vmcnt), so a fence in thread 0 alone would let a remote workgroup read
    // stale distances / list entries of wavefronts 1-3 after it sees the arrival
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) {
        phase++;
        __hip_atomic_fetch_add(bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase * nwg) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

__global__ __launch_bounds__(256) void k_flat_sweep_coop(CondArgs A, Regions R, int32_t *al0, int32_t *al1, int32_t *cnt3, int32_t *stamp,
                                                         double *dhA, double *dhB, double *dlA, double *dlB, int sweep0, int nsweeps,
                                                         double source_tol, int32_t *sweeps_done, int32_t *bar, int nwg, int cap, int xcd_only)
{
    int wg = blockIdx.x;
    if (xcd_only) { if (blockIdx.x & 7) return; wg = blockIdx.x >> 3; }
    int phase = 0, s = sweep0;
    for (; s < sweep0 + nsweeps; s++) {
        const int32_t na = __hip_atomic_load(&cnt3[s % 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (nobody writes it during sweep s)
        if (na == 0 || na > cap) break;
        if (wg == 0 && threadIdx.x == 0) __hip_atomic_store(&cnt3[(s + 2) % 3], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int cur = (s - 1) & 1;
        const int32_t *wl_in = cur ? al1 : al0;
        int32_t *wl_out = cur ? al0 : al1;
        const double *dh0 = cur ? dhB : dhA, *dl0 = cur ? dlB : dlA;
        double *dh1 = cur ? dhA : dhB, *dl1 = cur ? dlA : dlB;
        for (int32_t q = wg * 256 + threadIdx.x; q < na; q += nwg * 256)
            flat_sweep_entry(A, R, wl_in[q], wl_out, &cnt3[(s + 1) % 3], stamp, dh0, dh1, dl0, dl1, s, source_tol);
        coop_barrier(bar, nwg, phase);
    }
    if (wg == 0 && threadIdx.x == 0) *sweeps_done = s - sweep0;
}

// cells of regions that are still sweeping -> next active list.  Eight list entries per thread, ONE add to the list counter
// per workgroup trip (a single address: one add per wavefront of a 8.9 M-cell list was 140 k serial atomics = 1.4 ms)
__global__ __launch_bounds__(256) void k_flat_active(CondArgs A, Regions R, const int32_t *__restrict__ alist, int32_t na, int sweep, int32_t *out, int32_t *nout)
{
    constexpr int PER = 8;
    __shared__ int32_t wave_tot[4];
    __shared__ int32_t blk_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t base = (int64_t)blockIdx.x * (256 * PER); base < na; base += (int64_t)gridDim.x * (256 * PER)) {
        int32_t cell[PER];
        uint32_t keep = 0;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int64_t q = base + k * 256 + threadIdx.x;
            cell[k] = 0;
            if (q < na) {
                cell[k] = alist[q];
                const int32_t r = A.creg[cell[k]];
                // (a region that stopped in the last sweep stays for one more batch: those sweeps copy its final values
                // into the other buffer of the ping-pong pair, so that both agree when the region leaves the list)
                if ((R.flags[r] & RF_GENERAL) && (R.done_hi[r] >= sweep - 1 || R.done_lo[r] >= sweep - 1)) keep |= 1u << k;
            }
        }
        const int mine = __popc(keep);
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off); if (lane >= off) incl += o; }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int32_t tot = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
            blk_base = tot ? atomicAdd(nout, tot) : 0;
        }
        __syncthreads();
        int32_t o = blk_base + incl - mine;
        for (int w = 0; w < wave; w++) o += wave_tot[w];
#pragma unroll
        for (int k = 0; k < PER; k++) if (keep & (1u << k)) out[o++] = cell[k];
        __syncthreads();
    }
}

// ---- several sweeps per pass: blocks with a halo ----------------------------------------------------------------
// The tail of the sweeps is a front crossing the big plateaus: a few thousand listed cells per sweep, more than a thousand
// sweeps, each of them a chain of dependent round trips to L2 plus a launch or a device-wide barrier (13-36 us per sweep).
// A Jacobi sweep reads a 3x3 window, so a 32 x 32 block loaded with a halo of FT cells can run FT sweeps from LDS before it
// needs anybody else's results: the values it writes for its interior are the ones FT single sweeps would have left there
// (same operands, same operations, in the same order).  The blocks of a pass = those with a listed cell nearby; a block
// whose interior changed lists itself and its eight neighbours for the next pass (and is thereby rewritten once more, so that
// both buffers of the ping-pong pair agree on it when it leaves the list).
// What a block cannot know is the sweep at which a region STOPS -- its last cell may get its first value anywhere.  A pass
// runs as if no region stopped before its last sweep and counts the first arrivals per (region, sweep of the pass);
// k_flat_accept then books them, and when a region turns out to have stopped in mid-pass the same pass runs once more over
// the same input -- now with the stopping sweeps known -- for the blocks that hold cells of such a region.
#ifndef PYDEM_FLAT_T
#define PYDEM_FLAT_T 16
#endif
constexpr int FB = 32, FT = PYDEM_FLAT_T, FW = FB + 2 * FT;       // block edge, sweeps per pass = halo (8: 81 KB of LDS, 16: 144 KB), edge of the loaded window
static_assert(FT >= 1 && FT <= 16, "sweeps per pass");
constexpr int FB_THREADS = FT > 8 ? 1024 : 512;
constexpr uint32_t FI_NOSLOT = 0x3FFFu;                // (table rows: 14 bits)
constexpr uint32_t FI_MASK = 1u, FI_CARD_H = 2u, FI_ALL_H = 4u, FI_CARD_L = 8u, FI_ALL_L = 16u, FI_GEN = 32u;
struct FlatBatch {
    const double *dh_in, *dl_in;
    double *dh_out, *dl_out;
    const int32_t *blocks; const int32_t *nblk;      // (the count is read on the device: passes are queued without a look from the host)
    int nbi, nbj;                // blocks per column / row of the tile
    int s0, T;                   // first sweep of the pass, sweeps in it (<= FT)
    int first;                   // 1: the run that counts arrivals and lists the next blocks; 0: the repeat after k_flat_accept
    const int32_t *again;        // the repeat only runs when k_flat_accept raised this flag
    const int32_t *slot_of;      // per region: row of the arrival table (-1: none)
    int32_t *arr;                // [slot][2][FT] first arrivals of the pass
    int32_t *bstamp, *next_list, *next_count; int32_t pass_id;
    double source_tol;
    int32_t *err;
};

struct FlatLds {
    double dh[2][FW * FW], dl[2][FW * FW];
    uint32_t info[FW * FW];      // FI_* bits, bits 8-12 / 13-17: sweeps of this pass the cell's region takes part in (hi / lo), bits 18-31 table row
    int flag;
};

__device__ __forceinline__ void flat_batch_blocks(const CondArgs &A, const Regions &R, const FlatBatch &P, FlatLds &S, int wg, int nwg)
{
    double (*s_dh)[FW * FW] = S.dh, (*s_dl)[FW * FW] = S.dl;
    uint32_t *s_info = S.info;
    int &s_flag = S.flag;
    double *s_z = s_dh[1];                    // elevations of the window while the seeds are worked out
    const int n = A.n, m = A.m, tid = (int)threadIdx.x, T = P.T;
    if (!P.first && !*P.again) return;
    const int32_t nblk = *P.nblk;
    for (int bq = wg; bq < nblk; bq += nwg) {
        const int b = P.blocks[bq];
        const int bi = b / P.nbj, bj = b - bi * P.nbj;
        const int i0 = bi * FB - FT, j0 = bj * FB - FT;
        if (tid == 0) s_flag = 0;
        for (int idx = tid; idx < FW * FW; idx += FB_THREADS) {
            const int li = idx / FW, lj = idx - li * FW;
            const int gi = i0 + li, gj = j0 + lj;
            const bool ing = gi >= 0 && gi < n && gj >= 0 && gj < m;
            const int32_t c = ing ? gi * m + gj : 0;
            s_z[idx] = ing ? A.elev[c] : NAN;                     // (NaN: never a seed, like a neighbour that does not exist)
            s_info[idx] = (ing && A.mask[c]) ? FI_MASK : 0u;
        }
        __syncthreads();
        bool needfix = false;
        for (int idx = tid; idx < FW * FW; idx += FB_THREADS) {
            if (!(s_info[idx] & FI_MASK)) continue;
            const int li = idx / FW, lj = idx - li * FW;
            const int32_t c = (i0 + li) * m + j0 + lj;
            const int32_t r = A.creg[c];
            const int32_t fl = R.flags[r];
            uint32_t info = FI_MASK, jlh = 0, jll = 0, slot = FI_NOSLOT;
            if (fl & RF_GENERAL) {
                info |= FI_GEN;
                const int32_t dH = R.done_hi[r], dL = R.done_lo[r];
                jlh = dH >= P.s0 + T - 1 ? (uint32_t)T : (dH < P.s0 ? 0u : (uint32_t)(dH - P.s0 + 1));
                jll = dL >= P.s0 + T - 1 ? (uint32_t)T : (dL < P.s0 ? 0u : (uint32_t)(dL - P.s0 + 1));
                if (jlh | jll) {
                    const int32_t sl = P.slot_of[r];
                    slot = sl >= 0 ? (uint32_t)sl : FI_NOSLOT;
                    const double level = region_level(A, c);
                    const double src_max = (fl & RF_SOURCE) ? dunkey(R.lowest_bits[r]) + P.source_tol : 0.0;
#pragma unroll
                    for (int d = 0; d < 9; d++) {
                        if (d == 4) continue;
                        const int ni = li + d / 3 - 1, nj = lj + d % 3 - 1;
                        if (ni < 0 || ni >= FW || nj < 0 || nj >= FW) continue;
                        const int nidx = ni * FW + nj;
                        if (s_info[nidx] & FI_MASK) continue;
                        const double z = s_z[nidx];
                        const bool sh = (fl & RF_SOURCE) && z > level && z <= src_max, sl2 = (fl & RF_DRAIN) && z == level;
                        const bool cardinal = (d == 1 || d == 3 || d == 5 || d == 7);
                        if (sh) info |= FI_ALL_H | (cardinal ? FI_CARD_H : 0u);
                        if (sl2) info |= FI_ALL_L | (cardinal ? FI_CARD_L : 0u);
                    }
                    const bool interior = li >= FT && li < FT + FB && lj >= FT && lj < FT + FB;
                    if (interior && ((jlh > 0 && (int)jlh < T) || (jll > 0 && (int)jll < T))) needfix = true;
                }
                s_dh[0][idx] = P.dh_in[c]; s_dl[0][idx] = P.dl_in[c];
            }
            // (the FI_MASK bit the neighbours look at does not change; the word is complete before the barrier)
            atomicExch(&s_info[idx], info | (jlh << 8) | (jll << 13) | (slot << 18));
        }
        if (!P.first && needfix) s_flag = 1;
        __syncthreads();
        if (!P.first && !s_flag) { __syncthreads(); continue; }          // the repeat: nothing of a region that stopped in mid-pass in here
        bool changed = false;
        for (int j = 0; j < T; j++) {
            const int p = j & 1, lo = j + 1, hi = FW - 2 - j;
            for (int base = 0; base < FW * FW; base += FB_THREADS) {      // (the trip count is the same for all lanes of a wavefront: ballots below)
                const int idx = base + tid;
                bool arr_h = false, arr_l = false;
                uint32_t slot = FI_NOSLOT;
                if (idx < FW * FW) {
                    const int li = idx / FW, lj = idx - li * FW;
                    const uint32_t info = s_info[idx];
                    if (li >= lo && li <= hi && lj >= lo && lj <= hi && (info & FI_GEN)) {
                        const bool act_h = j < (int)((info >> 8) & 31u), act_l = j < (int)((info >> 13) & 31u);
                        const double oh = s_dh[p][idx], ol = s_dl[p][idx];
                        double nh = oh, nl = ol;
                        if (act_h || act_l) {
                            double card_h = oh, all_h = oh, card_l = ol, all_l = ol;
#pragma unroll
                            for (int d = 0; d < 9; d++) {
                                if (d == 4) continue;
                                const int nidx = idx + (d / 3 - 1) * FW + (d % 3 - 1);        // (li, lj in 1 .. FW - 2: inside the window)
                                if (!(s_info[nidx] & FI_MASK)) continue;
                                const double vh = s_dh[p][nidx], vl = s_dl[p][nidx];
                                const bool cardinal = (d == 1 || d == 3 || d == 5 || d == 7);
                                if (cardinal) { card_h = vh < card_h ? vh : card_h; card_l = vl < card_l ? vl : card_l; }
                                all_h = vh < all_h ? vh : all_h; all_l = vl < all_l ? vl : all_l;
                            }
                            if (info & FI_CARD_H) card_h = 0.0 < card_h ? 0.0 : card_h;
                            if (info & FI_ALL_H) all_h = 0.0 < all_h ? 0.0 : all_h;
                            if (info & FI_CARD_L) card_l = 0.0 < card_l ? 0.0 : card_l;
                            if (info & FI_ALL_L) all_l = 0.0 < all_l ? 0.0 : all_l;
                            if (act_h) {
                                const double sv = card_h + 1, g = all_h + SQRT2;
                                const double best = sv < g ? sv : g;
                                nh = best < oh ? best : oh;
                            }
                            if (act_l) {
                                const double sv = card_l + 1, g = all_l + SQRT2;
                                const double best = sv < g ? sv : g;
                                nl = best < ol ? best : ol;
                            }
                            if (li >= FT && li < FT + FB && lj >= FT && lj < FT + FB) {
                                changed = changed || nh != oh || nl != ol;
                                arr_h = act_h && isinf(oh) && !isinf(nh);
                                arr_l = act_l && isinf(ol) && !isinf(nl);
                                slot = info >> 18;
                            }
                        }
                        s_dh[p ^ 1][idx] = nh; s_dl[p ^ 1][idx] = nl;
                    }
                }
                if (P.first) {
                    // first arrivals per (region, sweep of the pass): the lanes that report for the same region add once
                    for (int f = 0; f < 2; f++) {
                        const bool arrived = f ? arr_l : arr_h;
                        unsigned long long pend = __ballot(arrived);
                        while (pend) {
                            const int leader = __ffsll((long long)pend) - 1;
                            const uint32_t ls = (uint32_t)__shfl((int)slot, leader);
                            const unsigned long long same = __ballot(arrived && slot == ls);
                            if ((int)__lane_id() == leader) {
                                if (ls == FI_NOSLOT) *P.err = 1;                        // a region without a table row moved: the caller gives up
                                else atomicAdd(&P.arr[((int64_t)ls * 2 + f) * FT + j], (int32_t)__popcll(same));
                            }
                            pend &= ~same;
                        }
                    }
                }
            }
            __syncthreads();
        }
        const int pf = T & 1;
        for (int q = tid; q < FB * FB; q += FB_THREADS) {
            const int li = FT + q / FB, lj = FT + q % FB, idx = li * FW + lj;
            if (s_info[idx] & FI_GEN) {
                const int32_t c = (i0 + li) * m + j0 + lj;
                P.dh_out[c] = s_dh[pf][idx]; P.dl_out[c] = s_dl[pf][idx];
            }
        }
        const int any = __syncthreads_or(changed ? 1 : 0);
        if (P.first && any && tid < 9) {
            const int ni = bi + tid / 3 - 1, nj = bj + tid % 3 - 1;
            if (ni >= 0 && ni < P.nbi && nj >= 0 && nj < P.nbj) {
                const int nb = ni * P.nbj + nj;
                if (atomicExch(&P.bstamp[nb], P.pass_id + 1) != P.pass_id + 1) P.next_list[atomicAdd(P.next_count, 1)] = nb;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(FB_THREADS) void k_flat_batch(CondArgs A, Regions R, FlatBatch P)
{
    __shared__ FlatLds S;
    flat_batch_blocks(A, R, P, S, (int)blockIdx.x, (int)gridDim.x);
}

// the regions that are still sweeping get a row of the arrival table (from the cells of the current work list: a region that
// still moves has cells on it)
__global__ __launch_bounds__(256) void k_flat_table(CondArgs A, Regions R, const int32_t *__restrict__ wl, int32_t na, int sweep, int32_t *slot_of,
                                                    int32_t *ar_id, int32_t *count, int32_t cap)
{
    for (int32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < na; q += gridDim.x * blockDim.x) {
        const int32_t r = A.creg[wl[q]];
        if (!(R.flags[r] & RF_GENERAL) || (R.done_hi[r] < sweep && R.done_lo[r] < sweep)) continue;
        if (atomicCAS(&slot_of[r], -1, -2) == -1) {
            const int32_t k = atomicAdd(count, 1);
            if (k < cap) { ar_id[k] = r; slot_of[r] = k; }
        }
    }
}

// blocks of the first pass: every block with a listed cell within FT cells of its interior
__global__ __launch_bounds__(256) void k_flat_blocks(const int32_t *__restrict__ wl, int32_t na, int n, int m, int nbj, int32_t *bstamp, int32_t pass_id,
                                                     int32_t *list, int32_t *count)
{
    for (int32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < na; q += gridDim.x * blockDim.x) {
        const int32_t c = wl[q];
        const int i = c / m, j = c - i * m;
        int seen[4]; int ns = 0;
        for (int k = 0; k < 4; k++) {
            int ii = i + ((k & 1) ? FT : -FT), jj = j + ((k & 2) ? FT : -FT);
            ii = ii < 0 ? 0 : (ii >= n ? n - 1 : ii); jj = jj < 0 ? 0 : (jj >= m ? m - 1 : jj);
            const int b = (ii / FB) * nbj + jj / FB;
            bool dup = false;
            for (int u = 0; u < ns; u++) dup = dup || seen[u] == b;
            if (dup) continue;
            seen[ns++] = b;
            if (atomicExch(&bstamp[b], pass_id) != pass_id) list[atomicAdd(count, 1)] = b;
        }
    }
}

// books the first arrivals of a pass: a region whose last cell arrived in sweep s0 + j stops there (done = that sweep, like
// region_arrivals); out[0] = 1 when that happened before the last sweep of the pass (the pass is repeated for its blocks)
__device__ __forceinline__ void flat_accept_rows(const Regions &R, const int32_t *__restrict__ ar_id, int32_t nt, int32_t *arr, int s0, int T, int32_t *out,
                                                 int first, int stride)
{
    for (int32_t k = first; k < nt; k += stride) {
        const int32_t r = ar_id[k];
        for (int f = 0; f < 2; f++) {
            int32_t *rem = f ? R.rem_lo : R.rem_hi, *done = f ? R.done_lo : R.done_hi;
            int32_t *a = arr + ((int64_t)k * 2 + f) * FT;
            int32_t left = rem[r];
            for (int j = 0; j < T; j++) {
                const int32_t got = a[j];
                a[j] = 0;
                if (!got) continue;
                left -= got;
                if (left == 0 && done[r] >= s0) { done[r] = s0 + j; if (j < T - 1) out[0] = 1; }
            }
            rem[r] = left;
        }
    }
}

// (queued behind every pass: clears the block counter and the repeat flag of the pass after the next / the next one, counts
// the passes that had blocks)
__global__ __launch_bounds__(256) void k_flat_accept(Regions R, const int32_t *__restrict__ ar_id, const int32_t *__restrict__ count, int32_t cap,
                                                     int32_t *arr, int s0, int T, int32_t *again, int32_t *again_next, const int32_t *nblk,
                                                     int32_t *clear_count, int32_t *stats)
{
    if (*nblk == 0) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) { *clear_count = 0; *again_next = 0; stats[0] += 1; stats[2] += *nblk; }
    const int32_t nt = *count < cap ? *count : cap;
    flat_accept_rows(R, ar_id, nt, arr, s0, T, again, (int)(blockIdx.x * blockDim.x + threadIdx.x), (int)(gridDim.x * blockDim.x));
}

// the new surface between the uphill rim and the outlet (:376-380)
__global__ void k_flat_interp(CondArgs A, Regions R, const double *__restrict__ dh, const double *__restrict__ dl)
{
    const int32_t nf = *A.count;
    const int n = A.n, m = A.m;
    for (int32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nf; q += gridDim.x * blockDim.x) {
        const int32_t c = A.list[q];
        const int32_t r = A.creg[c];
        const int32_t fl = R.flags[r];
        if (!(fl & RF_INTERP)) continue;
        const int i = c / m, j = c - i * m;
        // cells whose value is set, not interpolated: the LAST `pinned` assignment of _fill_flat wins (:353, :364, :370)
        bool pinned = false;
        if (fl & RF_DRN_EDGE) pinned = (i == 0 || j == 0 || i == n - 1 || j == m - 1);
        else if (fl & RF_DRN_CENTRE) pinned = R.centre[r] == c;
        else if (fl & RF_SRC_CENTRE) pinned = R.centre[r] == c;
        if (pinned) continue;
        const double level = region_level(A, c), top = R.top[r];
        const double h = dh[c], l = dl[c];
        A.built[c] = (top * (l * l) + level * (h * h)) / ((l * l) + (h * h));
    }
}

__global__ void k_roots(CondArgs A, int32_t *root_of)
{
    const int32_t nf = *A.count;
    for (int32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nf; q += gridDim.x * blockDim.x) {
        const int32_t c = A.list[q];
        if (A.labels[c] == c) root_of[A.rid[c]] = c;
    }
}

int grid_of(int64_t work, int cap) { const int64_t g = cdiv(work, 256); return (int)(g < cap ? (g > 0 ? g : 1) : cap); }

struct Scratch {            // the region records: one block of the device's conditioning arena, leased for the call
    ArenaLease lease;
    void *p = nullptr;
    int device = 0;
};

// carve the region records out of one allocation
int alloc_regions(Scratch &S, Regions &R, int32_t **root_of, int32_t **alist0, int32_t **alist1, int64_t nf)
{
    const size_t n4 = (size_t)nf * 4, n8 = (size_t)nf * 8;
    const size_t total = 15 * n4 + 5 * n8 + 256;
    if (!S.lease.held) PYDEM_TRY(arena_acquire(S.device, &S.lease));
    S.p = arena_take(&S.lease, total);
    if (!S.p) return -1;
    char *p = (char *)S.p;
    auto take4 = [&]() { int32_t *q = (int32_t *)p; p += n4; return q; };
    auto take8 = [&]() { unsigned long long *q = (unsigned long long *)p; p += n8; return q; };
    R.sum_i = take8(); R.sum_j = take8(); R.lowest_bits = take8(); R.cdist_bits = take8(); R.top = (double *)take8();
    R.size = take4(); R.n_edge = take4(); R.i0 = take4(); R.i1 = take4(); R.j0 = take4(); R.j1 = take4(); R.flags = take4();
    R.centre = take4(); R.done_hi = take4(); R.done_lo = take4(); R.rem_hi = take4(); R.rem_lo = take4();
    *root_of = take4(); *alist0 = take4(); *alist1 = take4();
    return 0;
}

// mask -> list -> labels -> region index; returns the number of mask cells and regions
int label_regions(pydem_tile *t, CondArgs &A, int32_t *nf_out, int32_t *nreg_out)
{
    int32_t *cnt = t->counters;
    HIP_TRY(hipMemsetAsync(cnt, 0, 16 * sizeof(int32_t), t->stream));
    const int big = grid_of(t->NN, 4096);
    hipLaunchKernelGGL(k_compact_flats, dim3(big), dim3(256), 0, t->stream, A.mask, t->NN, t->flatlist, cnt);
    HIP_TRY(hipMemcpyAsync(t->h_counters, cnt, sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    const int32_t nf = t->h_counters[0];
    *nf_out = nf; *nreg_out = 0;
    if (nf == 0) return 0;
    const int g1 = grid_of(nf, 2048);
    hipLaunchKernelGGL(k_label_init, dim3(g1), dim3(256), 0, t->stream, t->flatlist, cnt, t->labels, A.m);
    hipLaunchKernelGGL(k_label_union, dim3(g1), dim3(256), 0, t->stream, t->flatlist, cnt, A.mask, t->labels, A.n, A.m);
    hipLaunchKernelGGL(k_label_flatten, dim3(g1), dim3(256), 0, t->stream, t->flatlist, cnt, t->labels);
    hipLaunchKernelGGL(k_region_index, dim3(g1), dim3(256), 0, t->stream, A, cnt + 1);
    hipLaunchKernelGGL(k_cell_region, dim3(g1), dim3(256), 0, t->stream, A);
    HIP_TRY(hipMemcpyAsync(t->h_counters, cnt, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    *nreg_out = t->h_counters[1];
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace

// calc_fill_flats (:551-579) on the tile's resident elevation: artefact pits first when max_pit_area > 0 (:560-561),
// then every flat is re-surfaced.  Returns 1 (and leaves the tile untouched) only when a tile with no-data cells defeats the
// bounded replay of scipy's filter (k_cond_ring_*: salt-and-pepper no-data along a whole line).
int stage_fill_flats(pydem_tile *t, double max_pit_area, int below_sea, double source_tol, int peaks, int pits, int artefacts_only)
{
    const int n = (int)t->n, m = (int)t->m;
    PYDEM_TRY(tile_alloc(t, &t->flat0, (size_t)t->NN));
    PYDEM_TRY(tile_alloc(t, &t->labels, (size_t)t->NN));
    PYDEM_TRY(tile_alloc(t, &t->flatlist, (size_t)t->NN));
    PYDEM_TRY(tile_alloc(t, &t->queue[0], (size_t)t->NN));
    PYDEM_TRY(tile_alloc(t, &t->queue[1], (size_t)t->NN));
    CondArgs A;
    A.n = n; A.m = m; A.NN = t->NN; A.elev = t->elev; A.built = nullptr; A.mask = t->flat0; A.list = t->flatlist; A.count = t->counters;
    A.labels = t->labels; A.rid = t->queue[0]; A.creg = t->queue[1]; A.f32 = t->elev_f32 ? 1 : 0; A.below_sea = below_sea; A.nan_tile = 0;
    const int big = grid_of(t->NN, 8192);
    int32_t nf = 0, nreg = 0;
    // a tile with no-data cells: the candidate mask once more, with scipy's NaN behaviour replayed exactly (k_cond_ring_*;
    // the `mag` plane is scratch here).  Returns 1 when a line defeats the bounded replay (the host path takes the tile).
    auto nan_mask = [&](int corners_off) -> int {
        PYDEM_TRY(tile_alloc(t, &t->mag, (size_t)t->NN));
        HIP_TRY(hipMemsetAsync(t->counters + 9, 0, sizeof(int32_t), t->stream));
        hipLaunchKernelGGL(k_cond_ring_cols, dim3(big), dim3(256), 0, t->stream, A, t->mag, t->counters + 9);
        hipLaunchKernelGGL(k_cond_ring_rows_mask, dim3(big), dim3(256), 0, t->stream, A, (const double *)t->mag, corners_off, t->counters + 9);
        HIP_TRY(hipMemcpyAsync(t->h_counters + 9, t->counters + 9, sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(t->stream));
        return t->h_counters[9] > 0 ? 1 : 0;
    };
    // ---- quantisation artefacts
    if (max_pit_area > 0) {
        HIP_TRY(hipMemsetAsync(t->counters + 8, 0, sizeof(int32_t), t->stream));
        hipLaunchKernelGGL(k_cond_mask, dim3(big), dim3(256), 0, t->stream, A, 0, t->counters + 8);
        HIP_TRY(hipMemcpyAsync(t->h_counters + 8, t->counters + 8, sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        if (t->h_counters[8] > 0) { const int r = nan_mask(0); if (r) return r; A.nan_tile = 1; }
        PYDEM_TRY(label_regions(t, A, &nf, &nreg));
        if (nf > 0) {
            Scratch S; S.device = t->device; Regions R; int32_t *root_of, *al0, *al1;
            PYDEM_TRY(alloc_regions(S, R, &root_of, &al0, &al1, nf));
            const int g1 = grid_of(nf, 2048), gr = grid_of(nreg, 2048);
            hipLaunchKernelGGL(k_region_init, dim3(gr), dim3(256), 0, t->stream, R, nreg, n, m);
            hipLaunchKernelGGL(k_art_scan, dim3(g1), dim3(256), 0, t->stream, A, R, max_pit_area);
            hipLaunchKernelGGL(k_art_apply, dim3(g1), dim3(256), 0, t->stream, A, R, max_pit_area);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(t->stream));
        }
    }
    if (artefacts_only) return 0;
    // ---- flats: the surface becomes float64 (:562), `built` starts as a copy of it
    PYDEM_TRY(tile_alloc(t, &t->mag, (size_t)t->NN));
    PYDEM_TRY(tile_alloc(t, &t->dir, (size_t)t->NN));
    PYDEM_TRY(tile_alloc(t, &t->uca, (size_t)t->NN));
    PYDEM_TRY(tile_alloc(t, &t->twi, (size_t)t->NN));
    PYDEM_TRY(tile_alloc(t, &t->prop, (size_t)t->NN));
    A.f32 = 0;
    A.built = t->prop;
    HIP_TRY(hipMemsetAsync(t->counters + 8, 0, sizeof(int32_t), t->stream));
    hipLaunchKernelGGL(k_cond_mask, dim3(big), dim3(256), 0, t->stream, A, 1, t->counters + 8);
    HIP_TRY(hipMemcpyAsync(t->h_counters + 8, t->counters + 8, sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipMemcpyAsync(A.built, t->elev, (size_t)t->NN * 8, hipMemcpyDeviceToDevice, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    if (t->h_counters[8] > 0) { const int r = nan_mask(1); if (r) return r; A.nan_tile = 1; }
    PYDEM_TRY(label_regions(t, A, &nf, &nreg));
    if (nf > 0) {
        Scratch S; S.device = t->device; Regions R; int32_t *root_of, *al0, *al1;
        PYDEM_TRY(alloc_regions(S, R, &root_of, &al0, &al1, nf));
        const int g1 = grid_of(nf, 2048), gr = grid_of(nreg, 2048);
        double *dh[2] = {t->mag, t->dir}, *dl[2] = {t->uca, t->twi};
        hipLaunchKernelGGL(k_region_init, dim3(gr), dim3(256), 0, t->stream, R, nreg, n, m);
        hipLaunchKernelGGL(k_roots, dim3(g1), dim3(256), 0, t->stream, A, root_of);
        hipLaunchKernelGGL(k_flat_scan, dim3(g1), dim3(256), 0, t->stream, A, R);
        hipLaunchKernelGGL(k_flat_plan, dim3(gr), dim3(256), 0, t->stream, A, R, nreg, root_of, source_tol, peaks, pits);
        hipLaunchKernelGGL(k_centre_pass, dim3(g1), dim3(256), 0, t->stream, A, R, 0);
        hipLaunchKernelGGL(k_centre_pass, dim3(g1), dim3(256), 0, t->stream, A, R, 1);
        hipLaunchKernelGGL(k_flat_seed, dim3(g1), dim3(256), 0, t->stream, A, R, dh[0], dl[0]);
        hipLaunchKernelGGL(k_flat_seed_done, dim3(gr), dim3(256), 0, t->stream, R, nreg);
        // sweep 1 looks at every cell of a region in the general case (it also fills the second buffer of the pair); from
        // then on the work lists carry the cells whose neighbourhood changed.  The host looks at the list length every 32
        // sweeps; sweeps over an empty list are no-ops.
        int32_t *cnt = t->counters;
        PYDEM_TRY(tile_alloc(t, &t->indeg, (size_t)t->NN));
        int32_t *stamp = t->indeg;
        HIP_TRY(hipMemsetAsync(stamp, 0, (size_t)t->NN * 4, t->stream));
        HIP_TRY(hipMemsetAsync(cnt + 2, 0, 6 * sizeof(int32_t), t->stream));
        hipLaunchKernelGGL(k_flat_active, dim3(g1), dim3(256), 0, t->stream, A, R, t->flatlist, nf, 1, al0, cnt + 2);
        // (k_flat_active counted into cnt[2]; the sweep kernels rotate over cnt3 = cnt + 4 .. cnt + 6: sweep s reads cnt3[s % 3])
        HIP_TRY(hipMemcpyAsync(cnt + 4 + 1, cnt + 2, sizeof(int32_t), hipMemcpyDeviceToDevice, t->stream));
        HIP_TRY(hipMemcpyAsync(t->h_counters + 2, cnt + 2, sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        int32_t na = t->h_counters[2];
        int32_t *al[2] = {al0, al1};
        int cur = 0, sweep = 1, pp = 0;
        const int64_t sweep_cap = (int64_t)n * m + 2;
        int64_t cell_sweeps = 0;
        const int gw = grid_of(nf, 2048);
        static int small_cap = -1;          // lists up to this length are swept by one workgroup, many sweeps per launch
        if (small_cap < 0) { const char *e = getenv("PYDEM_FLAT_SMALL"); small_cap = e ? atoi(e) : 4096; if (small_cap > FS_CAP) small_cap = FS_CAP; }
        static int coop_cap = -1, coop_wg = 32, coop_xcd = 0, coop_min = 0;      // lists up to coop_cap: resident workgroups, a barrier per sweep
        if (coop_cap < 0) {
            const char *e = getenv("PYDEM_FLAT_COOP"); coop_cap = e ? atoi(e) : 8192;
            if ((e = getenv("PYDEM_FLAT_COOP_WG"))) coop_wg = atoi(e) > 0 ? atoi(e) : 32;
            if ((e = getenv("PYDEM_FLAT_COOP_XCD"))) coop_xcd = atoi(e);
            if ((e = getenv("PYDEM_FLAT_COOP_MIN"))) coop_min = atoi(e);
        }
        static int cond_debug = -1;
        if (cond_debug < 0) { const char *e = getenv("PYDEM_COND_DEBUG"); cond_debug = e ? atoi(e) : 0; }
        // several sweeps per pass from LDS (k_flat_batch) once the list is short enough that its blocks fit the device a few times over
        static int batch_cells = -1, batch_regions = 4096, batch_T = FT;
        if (batch_cells < 0) {
            const char *e = getenv("PYDEM_FLAT_BATCH"); batch_cells = e ? atoi(e) : 16384;
            if ((e = getenv("PYDEM_FLAT_BATCH_T"))) { batch_T = atoi(e); if (batch_T < 1 || batch_T > FT) batch_T = FT; }
            if ((e = getenv("PYDEM_FLAT_BATCH_REGIONS"))) { batch_regions = atoi(e); if (batch_regions < 1 || batch_regions >= (int)FI_NOSLOT) batch_regions = (int)FI_NOSLOT - 1; }
        }
        const int nbi = (n + FB - 1) / FB, nbj = (m + FB - 1) / FB;
        int32_t *b_slot = nullptr, *b_arid = nullptr, *b_arr = nullptr, *b_stamp = nullptr, *b_list[2] = {nullptr, nullptr};
        int batch_limit = batch_cells;
        int64_t batch_passes = 0, batch_blocks = 0;
        while (na > 0) {
            bool small_run = false;
            if (batch_limit > 0 && sweep >= 2 && na <= batch_limit) {
                if (!b_slot) {
                    const size_t nblocks = (size_t)nbi * nbj;
                    char *q = (char *)arena_take(&S.lease, (size_t)nreg * 4 + (size_t)batch_regions * 4 + (size_t)batch_regions * 2 * FT * 4 + 3 * nblocks * 4 + 1024);
                    if (!q) return -1;
                    b_slot = (int32_t *)q; q += (size_t)nreg * 4;
                    b_arid = (int32_t *)q; q += (size_t)batch_regions * 4;
                    b_arr = (int32_t *)q; q += (size_t)batch_regions * 2 * FT * 4;
                    b_stamp = (int32_t *)q; q += nblocks * 4;
                    b_list[0] = (int32_t *)q; q += nblocks * 4;
                    b_list[1] = (int32_t *)q;
                }
                // cnt[16]: table rows, cnt[18]: a region without a row moved (the rest of cnt[16..31]: see below)
                const int q0 = (sweep - 1) & 1;                            // sweep `sweep` reads dh[q0] (the list al[q0])
                HIP_TRY(hipMemsetAsync(b_slot, 0xFF, (size_t)nreg * 4, t->stream));
                HIP_TRY(hipMemsetAsync(b_arr, 0, (size_t)batch_regions * 2 * FT * 4, t->stream));
                HIP_TRY(hipMemsetAsync(b_stamp, 0, (size_t)nbi * nbj * 4, t->stream));
                HIP_TRY(hipMemsetAsync(cnt + 16, 0, 16 * sizeof(int32_t), t->stream));
                hipLaunchKernelGGL(k_flat_table, dim3(grid_of(na, 1024)), dim3(256), 0, t->stream, A, R, al[q0], na, sweep, b_slot, b_arid, cnt + 16, batch_regions);
                hipLaunchKernelGGL(k_flat_blocks, dim3(grid_of(na, 1024)), dim3(256), 0, t->stream, al[q0], na, n, m, nbj, b_stamp, 1, b_list[0], cnt + 20);
                HIP_TRY(hipMemcpyAsync(t->h_counters + 16, cnt + 16, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
                HIP_TRY(hipStreamSynchronize(t->stream));
                if (cond_debug > 1) fprintf(stderr, "fill_flats: sweep %d, list %d: %d regions still sweeping, %d blocks\n", sweep, na, t->h_counters[16], t->h_counters[20]);
                if (t->h_counters[16] > batch_regions) batch_limit = na / 2;      // too many regions still sweeping: more single sweeps first
                else {
                    // cnt[20..22]: block counts, rotating (pass k reads [k % 3], appends to [(k + 1) % 3], its k_flat_accept clears [(k + 2) % 3]);
                    // cnt[28 + k % 2]: the repeat flag of pass k; cnt[24..26]: passes with blocks / (unused) / block visits.
                    // The passes are queued CH at a time without a look from the host (a pass without blocks does nothing).
                    int bin = q0;
                    FlatBatch P;
                    P.nbi = nbi; P.nbj = nbj; P.T = batch_T; P.slot_of = b_slot; P.arr = b_arr; P.bstamp = b_stamp; P.source_tol = source_tol; P.err = cnt + 18;
                    const int CH = 16, gb = 256;
                    const int sweep_first = sweep;
                    int k = 0;
                    for (;;) {
                        for (int c = 0; c < CH; c++, k++) {
                            const int pbin = bin ^ (k & 1);
                            P.dh_in = dh[pbin]; P.dl_in = dl[pbin]; P.dh_out = dh[pbin ^ 1]; P.dl_out = dl[pbin ^ 1];
                            P.blocks = b_list[k & 1]; P.nblk = cnt + 20 + k % 3; P.s0 = sweep_first + k * batch_T; P.first = 1; P.again = cnt + 28 + (k & 1);
                            P.next_list = b_list[(k & 1) ^ 1]; P.next_count = cnt + 20 + (k + 1) % 3; P.pass_id = 1 + k;
                            hipLaunchKernelGGL(k_flat_batch, dim3(gb), dim3(FB_THREADS), 0, t->stream, A, R, P);
                            hipLaunchKernelGGL(k_flat_accept, dim3(grid_of(t->h_counters[16], 64)), dim3(256), 0, t->stream, R, b_arid, cnt + 16, batch_regions, b_arr,
                                               P.s0, batch_T, cnt + 28 + (k & 1), cnt + 28 + ((k + 1) & 1), cnt + 20 + k % 3, cnt + 20 + (k + 2) % 3, cnt + 24);
                            P.first = 0;
                            hipLaunchKernelGGL(k_flat_batch, dim3(gb), dim3(FB_THREADS), 0, t->stream, A, R, P);
                        }
                        HIP_TRY(hipMemcpyAsync(t->h_counters + 17, cnt + 17, 14 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
                        HIP_TRY(hipGetLastError());
                        HIP_TRY(hipStreamSynchronize(t->stream));
                        if (t->h_counters[18]) { pydem_set_error("fill_flats: a region outside the arrival table moved"); return -5; }
                        if (t->h_counters[20 + k % 3] == 0) break;
                        if (sweep_first + (int64_t)k * batch_T > sweep_cap + 256) { pydem_set_error("fill_flats: distance sweeps did not terminate"); return -5; }
                    }
                    batch_passes = t->h_counters[24]; batch_blocks = t->h_counters[26];
                    sweep = sweep_first + (int)batch_passes * batch_T; bin ^= (int)(batch_passes & 1);
                    pp = bin; na = 0;
                    if (cond_debug) fprintf(stderr, "fill_flats: %lld passes of %d sweeps, %lld block visits\n", (long long)batch_passes, batch_T, (long long)batch_blocks);
                    break;
                }
            }
            if (na <= coop_cap && na > coop_min) {
                HIP_TRY(hipMemsetAsync(cnt + 48, 0, sizeof(int32_t), t->stream));      // the barrier's arrival counter: a cache line of its own
                hipLaunchKernelGGL(k_flat_sweep_coop, dim3((unsigned)(coop_xcd ? coop_wg * 8 : coop_wg)), dim3(256), 0, t->stream, A, R, al[0], al[1], cnt + 4, stamp,
                                   dh[0], dh[1], dl[0], dl[1], sweep, 512, source_tol, cnt + 7, cnt + 48, coop_wg, coop_cap, coop_xcd);
                small_run = true;
            } else if (na <= small_cap) {
                const int ns = 256;
                hipLaunchKernelGGL(k_flat_sweep_small, dim3(1), dim3(1024), 0, t->stream, A, R, al[0], al[1], cnt + 4, stamp, dh[0], dh[1], dl[0], dl[1],
                                   sweep, ns, source_tol, cnt + 7);
                small_run = true;
            } else {
                for (int b = 0; b < 32; b++, sweep++) {
                    const int q = (sweep - 1) & 1;
                    hipLaunchKernelGGL(k_flat_sweep_wl, dim3(gw), dim3(256), 0, t->stream, A, R, al[q], al[q ^ 1], cnt + 4, stamp, dh[q], dh[q ^ 1],
                                       dl[q], dl[q ^ 1], sweep, source_tol);
                }
            }
            HIP_TRY(hipMemcpyAsync(t->h_counters + 4, cnt + 4, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
            HIP_TRY(hipStreamSynchronize(t->stream));
            if (small_run) sweep += t->h_counters[7];           // (it stops early when a list outgrows its LDS buffer)
            pp = cur = (sweep - 1) & 1;
            na = t->h_counters[4 + sweep % 3];
            cell_sweeps += na;
            if (cond_debug > 1) fprintf(stderr, "fill_flats: sweep %d, list %d\n", sweep, na);
            if (na > 0 && sweep > sweep_cap + 256) { pydem_set_error("fill_flats: distance sweeps did not terminate"); return -5; }
        }
        if (getenv("PYDEM_COND_DEBUG"))
            fprintf(stderr, "fill_flats: %d flat cells in %d regions, %d sweeps, %lld cell-sweeps\n", nf, nreg, sweep - 1, (long long)cell_sweeps);
        // (every region took part in at least one sweep after the one that stopped it: both buffers hold its final values)
        hipLaunchKernelGGL(k_flat_interp, dim3(g1), dim3(256), 0, t->stream, A, R, dh[pp], dl[pp]);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(t->stream));
    }
    // built -> elev
    HIP_TRY(hipMemcpyAsync(t->elev, A.built, (size_t)t->NN * 8, hipMemcpyDeviceToDevice, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    t->elev_f32 = false; t->elev_dtype = PYDEM_F64;      // float64 from here on, like the reference's data.astype('float64') (:562)
    return 0;
}
