// cond_paths.hip -- elevation conditioning on the device, part 2: pit drain paths.
//
// Replaces DEMProcessor.calc_pit_drain_paths (reference pydem/dem_processing.py:428-548; helpers
// utils.get_border_index pydem/utils.py:313-340, _get_dX_mean :1993-1997) for a surface resident in HBM (float64 values;
// the dtype the reference would edit the array in decides how the path values round, PathArgs::dtype_mode).
// The reference visits the strict local minima in ascending elevation (np.argsort, :450-452), grows a region from
// each through its lowest rim cells until a rim cell lies below the pit, prunes the visiting order to an
// 8-connected chain and rewrites the elevations along it IN PLACE (:535-539): pit k sees the paths of pits 0..k-1.
//
// That order is kept exactly, but not by running the pits one after the other.  A round takes the first pending
// pits of the order (one wavefront each) and lets all of them SIMULATE on the current surface; every simulation
// reserves the cells it read (rown = smallest order among the readers) and the cells it would write (wown).  Pit k
// then commits iff no earlier pending pit writes a cell k read and no earlier pending pit read a cell k writes -- its
// simulation saw exactly the surface the sequential loop would have shown it.  Everybody else simulates again next
// round.  The earliest pending pit always commits, so the loop terminates; on fractal terrain most of a window
// commits in its first round (pits interact only through shared channels).
// One thing the reservations cannot see: a pit that has to wait may, after an earlier path lowered its rim, grow in a
// new direction and meet a cell that a LATER pit has already rewritten.  Committed cells carry the order of their
// writer; a simulation that reads a cell written by a later pit raises a flag and the caller falls back to the host
// loop (cond_host.cpp: pydem_cond_pit_paths) on the untouched surface.  The soak tools count how often
// that happens.
//
// A pit that has to wait is NOT simulated again as long as nothing it read has changed: its footprint, chain and chain
// values stay where they are (storage belongs to the pit, not to its place in the round's window), every commit stamps the
// cells it rewrites with the round (wround), and a waiting pit whose footprint holds no stamp >= the round of its simulation
// only renews its reservations (kept_simulation) -- on the 8192^2 SRTM-like tile 2.5 simulations per pit otherwise.
//
// A simulation is a wavefront: membership of region + rim in a window bitmap in LDS (64 x 64 cells; pits that leave
// it are re-run in a 256 x 256 window, those that leave that one too in a 640 x 640 window -- 300 iterations cannot
// leave that one; a pit remembers the window it needs, and the medium / large-window simulations of such pits run on
// the side stream beside the small-window kernel of the round), the rim as a list with holes, the minimum by a
// wave reduction, the cells at the minimum in ascending cell order by setting their bits in a second bitmap and
// scanning it (the reference sorts them, :470), the trail in visiting order.  The arithmetic of the path (index and
// metric reach, np.sum's pairwise order for the dY sums, np.linspace) follows the host implementation line by line.
#include "internal.h"
#include <hipcub/hipcub.hpp>
#include <math.h>
#include <string.h>
#include <algorithm>
#include <time.h>

namespace {

#ifdef PYDEM_PATHS_PROF
// profiling build (tools/gpu_c5prof.sh): 10 ns ticks of the phases of a simulation, summed per window class (0 small, 1 medium / large)
// in 64 banks (the sums of a simulation leave in one go when it ends): 0 clear, 1 min scan, 2 fresh pass, 3 emit, 4 ring, 5 footprint,
// 6 path (one lane), 7 iterations, 8 whole simulation, 9 reservations, 10 simulations, 11 kept checks (ticks), 12 kept checks (count),
// 13 the longest simulation
__device__ unsigned long long g_paths_prof[2][64][16];
struct PProf {
    unsigned long long v[16]; int cls, bank, lane;
    __device__ PProf(int cls_, int bank_, int lane_) : cls(cls_), bank(bank_ & 63), lane(lane_) { for (int i = 0; i < 16; i++) v[i] = 0; }
    __device__ ~PProf() { if (lane == 0) { for (int i = 0; i < 16; i++) if (v[i] && i != 13) atomicAdd(&g_paths_prof[cls][bank][i], v[i]); atomicMax(&g_paths_prof[cls][bank][13], v[8]); } }
};
#define PPROF(i, expr) do { const long long t0_ = wall_clock64(); expr; pp_.v[i] += (unsigned long long)(wall_clock64() - t0_); } while (0)
#else
#define PPROF(i, expr) do { expr; } while (0)
#endif

constexpr int ST_PENDING = 0, ST_FAILED = 1, ST_PATH = 2, ST_OVERFLOW = 3, ST_TOOBIG = 4;

struct PathArgs {
    double *e;               // the surface (read by the simulations, written by the commits)
    int n, m;
    const int32_t *order;    // pit cells in processing order
    const int32_t *window;   // [nw] indices into `order` of the pits of this round (ascending)
    int nw;
    const double *dX, *dY; int ndX;
    int max_iter, max_dist; double max_dist_XY;
    unsigned long long mmagic;   // ceil(2^63 / m): cell / m = umul64hi(2 * cell, mmagic), exact for every cell id (the simulations decode
                             // cell ids all the time and a 32-bit division by a run-time divisor is ~40 instructions)
    int dtype_mode;          // the dtype the reference edits the array in (:535-539): 0 float64, 1 integer (path values truncate), 2 float32 (they round)
    int32_t *rown, *wown;    // [NN] smallest order among the pending readers / writers of a cell (INT_MAX: none)
    int32_t *bown;           // [NN] smallest order among the pits that read the cell and stay pending after this round
    int32_t *wstamp, *rstamp;   // [NN] largest order among the committed writers / readers of a cell (-1: none)
    int32_t *tent;           // per slot: passed the first two commit rules
    // per window slot
    int32_t *status, *nF, *nC, *iters;
    int32_t **Fp, **Cp; double **CVp;      // where the slot's footprint / chain / chain values live
    int32_t *fcap, *ccap;
    int32_t *flags;          // [0] a simulation met a cell written by a later pit, [1] capacity exceeded, [2] first pit of the order whose
                             // simulation left the window of its tier in this round (nobody from it on may commit; INT_MAX: none)
    const int32_t *tier;     // per slot: 0 small window, 1 medium, 2 large (what earlier rounds learned about the pit)
    // simulations that outlive their round
    int round;               // 1, 2, ...
    int count_kept;          // debug: count the kept simulations (flags[4]) and the stale ones (flags[5])
    int32_t *wround;         // [NN] round of the last commit that rewrote the cell (0: never)
    int32_t *simround;       // per slot: round of the simulation whose results the slot holds (<= 0: none)
    const int32_t *home;     // per slot: the pit's block of the small-window footprint / chain arrays (it keeps it while it waits)
    const int32_t *src;      // per slot: the slot the pit had in the previous round (-1: new, or it sat that round out)
    const int32_t *o_status, *o_nF, *o_nC, *o_iters, *o_simround;      // the previous round's per-slot state (carried over by k_paths_slots)
};

// The slot holds a finished simulation and no commit has rewritten a cell it read since: the simulation is what a new one
// would compute, cell for cell.  Its reservations (released at the end of every round) are made again.
__device__ bool kept_simulation(const PathArgs &A, int slot, int lane)
{
    const int sr = A.simround[slot], st = A.status[slot];
    if (sr <= 0 || (st != ST_FAILED && st != ST_PATH)) return false;
    const int32_t *F = A.Fp[slot];
    const int nF = A.nF[slot];
    bool stale = false;
    for (int q = lane; q < nF; q += 64) if (A.wround[F[q]] >= sr) stale = true;
    if (__any(stale)) { if (A.count_kept && lane == 0) atomicAdd(&A.flags[5], 1); return false; }
    if (A.count_kept && lane == 0) atomicAdd(&A.flags[4], 1);
    const int k = A.window[slot];
    for (int q = lane; q < nF; q += 64) atomicMin(&A.rown[F[q]], k);
    if (st == ST_PATH) {
        const int32_t *C = A.Cp[slot];
        const int nC = A.nC[slot];
        for (int q = lane; q < nC; q += 64) atomicMin(&A.wown[C[q]], k);
    }
    return true;
}

__device__ __forceinline__ double wave_min(double v)
{
    for (int o = 32; o > 0; o >>= 1) { const double w = __shfl_xor(v, o); v = w < v ? w : v; }
    return v;
}

// numpy's pairwise sum of a contiguous vector (np.add.reduce), n <= a few hundred here
__device__ double np_sum_dev(const double *a, int64_t n)
{
    if (n < 8) { double r = 0.; for (int64_t i = 0; i < n; i++) r += a[i]; return r; }
    if (n <= 128) {
        double r[8];
        for (int k = 0; k < 8; k++) r[k] = a[k];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_sum_dev(a, n2) + np_sum_dev(a + n2, n - n2);
}

__device__ double pit_reach(const PathArgs &A, int pi, int pj, int32_t t)
{
    const int m = A.m;
    const int64_t ti = t / m, tj = t % m;
    const int64_t lo = pi < ti ? pi : ti, hi = pi < ti ? ti : pi;
    double dxm;
    if (pi == ti) dxm = A.dX[pi < A.ndX - 1 ? pi : A.ndX - 1];                 // _get_dX_mean :1993-1997
    else dxm = np_sum_dev(A.dX + lo, hi - lo) / (double)(hi - lo);
    const double run = dxm * (double)(pj - tj);
    const double rise = np_sum_dev(A.dY + lo, hi - lo);
    return sqrt(run * run + rise * rise);
}

// the same for the small window, whose rows of dX / dY were staged in LDS when the simulation began (rows[0 .. 63] = dX[oi ..],
// rows[64 .. 127] = dY[oi ..]): the outlet lies in the window, so every row between it and the pit does; at most 32 addends, numpy's
// pairwise order (np_sum_dev without its recursion above 128), no 64-bit division
__device__ __forceinline__ double np_sum_le128(const double *a, int n)
{
    if (n < 8) { double r = 0.; for (int i = 0; i < n; i++) r += a[i]; return r; }
    double r[8];
    for (int k = 0; k < 8; k++) r[k] = a[k];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int k = 0; k < 8; k++) r[k] += a[i + k];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}
__device__ double pit_reach_rows(const PathArgs &A, int pi, int pj, int ti, int tj, const double *rows, int oi)
{
    const int lo = pi < ti ? pi : ti, hi = pi < ti ? ti : pi;
    double dxm;
    if (pi == ti) dxm = rows[(pi < A.ndX - 1 ? pi : A.ndX - 1) - oi];           // _get_dX_mean :1993-1997
    else dxm = np_sum_le128(rows + (lo - oi), hi - lo) / (double)(hi - lo);
    const double run = dxm * (double)(pj - tj);
    const double rise = np_sum_le128(rows + 64 + (lo - oi), hi - lo);
    return sqrt(run * run + rise * rise);
}

// One pit, one wavefront.  WIN: window edge, RCAP: rim capacity; the trail lives in LDS for the small window and in
// global scratch for the large one.
template <int WIN, int RCAP>
__device__ void simulate_pit(const PathArgs &A, int slot, uint32_t *seen, uint32_t *freshmap, int32_t *rim, double *rimz, uint16_t *holes, int32_t *flist, int32_t *trail, int tcap,
                             double *rows = nullptr)
{
    const int lane = (int)(threadIdx.x & 63);
    const int k = A.window[slot];
    const int32_t pit = A.order[k];
    const int n = A.n, m = A.m;
    const int pi = pit / m, pj = pit - pi * m;
    const int oi = pi - WIN / 2, oj = pj - WIN / 2;          // window origin (may be negative)
    constexpr int WORDS = WIN * WIN / 32;
#ifdef PYDEM_PATHS_PROF
    PProf pp_(WIN > 64 ? 1 : 0, slot, lane);
    pp_.v[10] = 1;
    const long long tsim0_ = wall_clock64();
    struct SimEnd { PProf &p; long long t0; __device__ ~SimEnd() { p.v[8] += (unsigned long long)(wall_clock64() - t0); } } simend_{pp_, tsim0_};
#endif
    PPROF(0, for (int w = lane; w < WORDS; w += 64) { seen[w] = 0; freshmap[w] = 0; });
    if (lane == 0) A.simround[slot] = A.round;
    if (WIN == 64 && rows) {                                 // the window's rows of dX / dY (the choice of the outlet sums over them)
        const int r = oi + lane;
        const bool in = r >= 0 && r < A.ndX;
        rows[lane] = in ? A.dX[r] : 0.0; rows[64 + lane] = in ? A.dY[r] : 0.0;
    }
    __builtin_amdgcn_wave_barrier();
    // the rim is an unordered list with HOLES: a cell that leaves it (promoted into the region) frees its slot (rim = -1,
    // height +inf), the slot goes on a stack and the next cell that joins takes it -- nothing is compacted per iteration.
    // nrim = slots in use incl. holes, nh = holes on the stack, n_alive = cells on the rim.  Nothing depends on the order of
    // the list: the cells at the lowest height are sorted by cell id before they join the trail (:470).
    int nrim = 0, ntrail = 0, nh = 0, n_alive = 0;
    bool overflow = false, later = false;
    auto row_of = [&](int32_t c) -> int { return (int)__umul64hi((unsigned long long)(uint32_t)c << 1, A.mmagic); };
    auto bit_of = [&](int32_t c, int &word, uint32_t &mask) -> bool {
        const int ci = row_of(c);
        const int i = ci - oi, j = c - ci * m - oj;
        if (i < 0 || i >= WIN || j < 0 || j >= WIN) return false;
        const int b = i * WIN + j;
        word = b >> 5; mask = 1u << (b & 31);
        return true;
    };
    // the 8 neighbours of the cells src[0..cnt) that have not been seen yet join the rim
    auto add_ring = [&](const int32_t *src, int cnt) {
        for (int base = 0; base < cnt * 8; base += 64) {
            const int q = base + lane;
            bool add = false; int32_t t = -1;
            if (q < cnt * 8) {
                const int32_t c = src[q >> 3];
                int d = q & 7; d += (d >= 4);
                const int ci = row_of(c);
                const int ii = ci + d / 3 - 1, jj = c - ci * m + d % 3 - 1;
                if (ii >= 0 && ii < n && jj >= 0 && jj < m) {
                    t = ii * m + jj;
                    const int wi = ii - oi, wj = jj - oj;                            // (window coordinates straight from ii, jj: no second decode)
                    if (wi < 0 || wi >= WIN || wj < 0 || wj >= WIN) overflow = true;
                    else {
                        const int b = wi * WIN + wj;
                        const uint32_t mask = 1u << (b & 31);
                        add = !(atomicOr(&seen[b >> 5], mask) & mask);
                    }
                }
            }
            const unsigned long long bal = __ballot(add);
            const int cnt = __popcll(bal), rk = __popcll(bal & ((1ull << lane) - 1ull));
            const int grow = cnt > nh ? cnt - nh : 0;                                  // slots taken beyond the list end
            if (add) {
                // the cell's elevation is read ONCE, when it joins the rim (nothing changes the surface while the round simulates)
                const int pos = rk < nh ? (int)holes[nh - 1 - rk] : nrim + (rk - nh);
                if (pos < RCAP) { rim[pos] = t; rimz[pos] = A.e[t]; if (A.wstamp[t] > k) later = true; }
                else overflow = true;
            }
            nh -= cnt - grow; nrim += grow; n_alive += cnt;
            __builtin_amdgcn_wave_barrier();                                           // (the stack entries are spent before the next batch reads it)
        }
        overflow = __any(overflow);
        if (nrim > RCAP) nrim = RCAP;
        __builtin_amdgcn_wave_barrier();
    };
    // bits of a window bitmap (optionally AND NOT a second one) -> cells in raster = ascending cell order
    auto emit_bits = [&](const uint32_t *map, const uint32_t *minus, bool clear, int32_t *dst, int start, int cap, int stop_at, int first_word = 0) -> int {
        int emitted = start;
        bool spill = false;
        for (int base = first_word & ~63; base < WORDS; base += 64) {
            const int w = base + lane;
            uint32_t bits = (w < WORDS) ? (minus ? (map[w] & ~minus[w]) : map[w]) : 0u;
            const int cntw = __popc(bits);
            int incl = cntw;
            for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
            int pos = emitted + incl - cntw;
            while (bits) {
                const int b = __ffs((int)bits) - 1; bits &= bits - 1;
                const int idx = w * 32 + b;
                if (pos < cap) dst[pos] = (oi + idx / WIN) * m + (oj + idx % WIN); else spill = true;
                pos++;
            }
            if (clear && w < WORDS) const_cast<uint32_t *>(map)[w] = 0;
            emitted += __shfl(incl, 63);
            if (stop_at >= 0 && emitted - start >= stop_at) break;           // (the rest of the map is empty)
        }
        __builtin_amdgcn_wave_barrier();
        return __any(spill) ? -1 : emitted;
    };
    { int word = 0; uint32_t mask = 0; bit_of(pit, word, mask); if (lane == 0) { seen[word] |= mask; trail[0] = pit; } }
    ntrail = 1;
    __builtin_amdgcn_wave_barrier();
    add_ring(trail, 1);
    const double floor_ = A.e[pit];
    if (A.wstamp[pit] > k) later = true;
    later = __any(later);
    bool found = false;
    int it = 0, n_out = 0;
    double lowest_out = 0.0;     // the height of the outlet candidates
    int32_t *outlet = rim;       // the outlet candidates overwrite the rim list once the growth is over
    if (!overflow) {
        for (it = 0; it < A.max_iter; it++) {
            if (n_alive == 0) break;
            double lowest = INFINITY; bool has_nan = false;
            PPROF(1, for (int q = lane; q < nrim; q += 64) {
                const double z = rimz[q];
                if (z < lowest) lowest = z;
                if (isnan(z)) has_nan = true;
            }
            lowest = wave_min(lowest));
            if (__any(has_nan)) break;                       // np.min propagates NaN: the pit fails (:468)
            // cells at the lowest height leave the rim (their slots become holes): the first 64 go to a list (sorted in registers
            // below), any further ones into the `freshmap` bitmap
            int nfresh = 0, wlow = WORDS;          // (wlow: the first bitmap word that holds one of them -- the scan below starts there)
#ifdef PYDEM_PATHS_PROF
            const long long tf0_ = wall_clock64();
#endif
            for (int base = 0; base < nrim; base += 64) {
                const int q = base + lane;
                bool is_fresh = false; int32_t t = -1;
                if (q < nrim) { t = rim[q]; is_fresh = t >= 0 && rimz[q] == lowest; }
                const unsigned long long bf = __ballot(is_fresh);
                if (!bf) continue;
                const int rkf = __popcll(bf & ((1ull << lane) - 1ull));
                const int pf = nfresh + rkf;
                if (is_fresh) {
                    int word = 0; uint32_t mask = 0; bit_of(t, word, mask);
                    if (word < wlow) wlow = word;
                    if (pf < 64) flist[pf] = t;
                    else atomicOr(&freshmap[word], mask);
                    rim[q] = -1; rimz[q] = INFINITY; holes[nh + rkf] = (uint16_t)q;
                }
                nh += __popcll(bf);
                nfresh += __popcll(bf);
            }
            n_alive -= nfresh;
            __builtin_amdgcn_wave_barrier();
#ifdef PYDEM_PATHS_PROF
            pp_.v[2] += (unsigned long long)(wall_clock64() - tf0_);
#endif
            // ascending cell order (:470): up to 64 cells by rank (one cell per lane), more through the window bitmap
            auto emit_fresh = [&](int32_t *dst, int start, int cap) -> int {
                if (nfresh <= 64) {
                    if (start + nfresh > cap) return -1;
                    const int32_t mine = lane < nfresh ? flist[lane] : 0x7FFFFFFF;
                    int rank = 0;
                    for (int o = 0; o < nfresh; o++) rank += __shfl(mine, o) < mine;
                    __builtin_amdgcn_wave_barrier();
                    if (lane < nfresh) dst[start + rank] = mine;
                    __builtin_amdgcn_wave_barrier();
                    return start + nfresh;
                }
                for (int q = lane; q < 64; q += 64) { int word = 0; uint32_t mask = 0; bit_of(flist[q], word, mask); atomicOr(&freshmap[word], mask); }
                __builtin_amdgcn_wave_barrier();
                // plateau pits promote a whole ring per iteration: the scan of the 640 x 640 bitmap starts at the ring's first row
                // and stops after its last cell instead of walking 100 x 64 words from the window's top every time
                int w0 = wlow;
                for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(w0, o); w0 = v < w0 ? v : w0; }
                return emit_bits(freshmap, nullptr, true, dst, start, cap, nfresh, w0);
            };
            if (lowest < floor_) {                           // the first lower rim cells: outlet candidates (:471-473)
                const int got = emit_fresh(outlet, 0, RCAP);
                if (got < 0) overflow = true;
                n_out = nfresh; found = true; lowest_out = lowest;
                break;
            }
            if (nfresh > tcap - ntrail) { overflow = true; break; }
            PPROF(3, emit_fresh(trail, ntrail, tcap));
            const int first = ntrail;
            ntrail += nfresh;
            PPROF(4, add_ring(trail + first, nfresh));
            if (overflow) break;
#ifdef PYDEM_PATHS_PROF
            pp_.v[7] += 1ull;
#endif
        }
    }
    if (__any(later) && lane == 0) atomicOr(&A.flags[0], 1);
    if (overflow) { if (lane == 0) { A.status[slot] = ST_OVERFLOW; A.nF[slot] = 0; A.nC[slot] = 0; A.iters[slot] = 0; } return; }
    if (lane == 0) A.iters[slot] = found ? it + 1 : 0;
    // ---- what the simulation read: region + everything that was ever on the rim = the `seen` map
    int32_t *F = A.Fp[slot];
    const int fcap = A.fcap[slot];
    int nF = 0;
    PPROF(5, nF = emit_bits(seen, nullptr, false, F, 0, fcap, -1));
    if (nF < 0) { if (lane == 0) { A.status[slot] = ST_OVERFLOW; A.nF[slot] = 0; A.nC[slot] = 0; } return; }
    // ---- the path: one lane does what the host loop does (:485-539)
    int32_t *C = A.Cp[slot];
    double *CV = A.CVp[slot];
    const int ccap = A.ccap[slot];
    int nC = 0, st = found ? ST_PATH : ST_FAILED;
    int32_t end = -1;
#ifdef PYDEM_PATHS_PROF
    const long long tpath0_ = wall_clock64();
#endif
    if (lane == 0 && found) {
        int no = n_out;
        if (A.max_dist) {                                    // index-space reach (:485-493)
            int w = 0;
            for (int q = 0; q < no; q++) {
                const int32_t t = outlet[q];
                const int tr = row_of(t);
                const double di = (double)(pi - tr), dj = (double)(pj - (t - tr * m));
                if (sqrt(di * di + dj * dj) <= (double)A.max_dist) outlet[w++] = t;
            }
            no = w;
        }
        const bool use_xy = A.max_dist_XY != 0 && !isnan(A.max_dist_XY);
        int valid = 0; double best = INFINITY;
        auto reach = [&](int32_t t) -> double {
            if (WIN == 64 && rows) { const int tr = row_of(t); return pit_reach_rows(A, pi, pj, tr, t - tr * m, rows, oi); }
            return pit_reach(A, pi, pj, t);
        };
        for (int q = 0; q < no; q++) {                       // metric reach (:494-512)
            const double r = reach(outlet[q]);
            if (use_xy && !(r <= A.max_dist_XY)) continue;
            if (valid == 0) end = outlet[q];
            valid++;
            if (r < best) best = r;
        }
        if (valid == 0) st = ST_FAILED;
        else if (valid > 1) {
            for (int q = 0; q < no; q++) {
                const double r = reach(outlet[q]);
                if (use_xy && !(r <= A.max_dist_XY)) continue;
                if (r == best) { end = outlet[q]; break; }
            }
        }
#ifdef PYDEM_PATHS_PROF
        pp_.v[14] += (unsigned long long)(wall_clock64() - tpath0_);
#endif
        if (st == ST_PATH) {
            // prune the trail, walking back from the outlet, to an 8-connected chain (:516-532): an entry stays when it touches the
            // entry kept after it.  The walk is one lane's and stays in the trail's own storage (LDS for the small window): the w-th
            // entry kept goes to trail[ntrail - w] -- never below the entry being read, since at most ntrail - q entries are kept
            // when entry q is read -- so the chain ends up in path order at trail[ntrail - nC + 1 .. ntrail].  (It was written
            // backwards into the slot's chain in global memory and reversed there, load after store, by the same lane: a quarter of
            // a simulation's time.)
            if (ntrail + 1 > ccap) st = ST_TOOBIG;
            else {
                int w = 0;
                trail[ntrail - w++] = end;
                int bi = row_of(end), bj = end - bi * m;                 // the entry kept last
                for (int q = ntrail - 1; q >= 1; q--) {
                    const int32_t a = trail[q];
                    const int ai = row_of(a), aj = a - ai * m;
                    const int dii = ai > bi ? ai - bi : bi - ai, djj = aj > bj ? aj - bj : bj - aj;
                    if (dii <= 1 && djj <= 1) { trail[ntrail - w++] = a; bi = ai; bj = aj; }
                }
                trail[ntrail - w++] = pit;
                nC = w;
            }
        }
    }
    nC = __shfl(nC, 0); st = __shfl(st, 0);
#ifdef PYDEM_PATHS_PROF
    pp_.v[15] += (unsigned long long)(wall_clock64() - tpath0_);
#endif
    if (st == ST_PATH) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // elevations fall linearly along the chain (:535-539), every lane its entries.  The outlet was taken off the rim at the
        // height `lowest` < the pit's height `floor_` (both read from the surface this round simulates on): the reference's
        // "the pit lies below the outlet" branch (:536) cannot be taken here.
        const double base = floor_, e_end = lowest_out;
        // (a float32 surface takes the difference in float32, :537; the stored values get the array's dtype, :539)
        const double drop = A.dtype_mode == 2 ? (double)((float)e_end - (float)base) : e_end - base;
        const double step = 1.0 / (double)(nC - 1);  // np.linspace(0, 1, L): arange * step, last = 1
        const int32_t *chain = trail + (ntrail - nC + 1);
        for (int q = lane; q < nC; q += 64) {
            const double f = (q == nC - 1) ? 1.0 : (double)q * step;
            double v = base + f * drop;
            if (A.dtype_mode == 1) v = trunc(v);
            else if (A.dtype_mode == 2) v = (double)(float)v;
            C[q] = chain[q];
            CV[q] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
#ifdef PYDEM_PATHS_PROF
    pp_.v[6] += (unsigned long long)(wall_clock64() - tpath0_);
#endif
    if (lane == 0) {
        if (st == ST_TOOBIG) { st = ST_OVERFLOW; }
        A.status[slot] = st; A.nF[slot] = st == ST_OVERFLOW ? 0 : nF; A.nC[slot] = st == ST_PATH ? nC : 0;
    }
    st = __shfl(st, 0);
    __builtin_amdgcn_wave_barrier();
    if (st == ST_OVERFLOW) return;
    PPROF(9, {
    for (int q = lane; q < nF; q += 64) atomicMin(&A.rown[F[q]], k);
    if (st == ST_PATH) for (int q = lane; q < nC; q += 64) atomicMin(&A.wown[C[q]], k);
    });
}

// small window: rim / trail capacities and the wavefronts per SIMD the kernel is compiled for.  Measured on the 8192^2
// SRTM-like tile (989 k small-window simulations): 512 / 1024 / 3 (49 KB of LDS per four simulations, 140 VGPRs) 34.9 ms;
// 384 / 896 / 4 28.2; 256 / 512 / 4 23.2 (+2400 pits that go on to the medium window: +1 ms there); 192 / 384 and 128 / 256
// lose more to the medium window and to extra rounds than they gain
#ifndef PYDEM_SRCAP
#define PYDEM_SRCAP 256
#define PYDEM_STCAP 512
#define PYDEM_SMALL_WAVES 4
#endif
#ifndef PYDEM_PATHS_CHUNK
#define PYDEM_PATHS_CHUNK 4
#endif
constexpr int SWIN = 64, SRCAP = PYDEM_SRCAP, STCAP = PYDEM_STCAP;
#ifndef PYDEM_MWIN
#define PYDEM_MWIN 256
#define PYDEM_MRCAP 1024
#endif
constexpr int MWIN = PYDEM_MWIN, MRCAP = PYDEM_MRCAP;       // 16 + 14 KB of LDS: five medium-window simulations per CU (measured: 192^2 and 128^2
                                                           // windows or a rim of 512 send too many pits on to the large window; 320^2 / 2048 cost occupancy)
constexpr int BWIN = 640, BRCAP = 4096;                    // 102 + 56 KB: one per CU

__global__ __launch_bounds__(256, PYDEM_SMALL_WAVES) void k_paths_small(PathArgs A, int nslots)
{
    __shared__ uint32_t s_seen[4][SWIN * SWIN / 32], s_fresh[4][SWIN * SWIN / 32];
    __shared__ int32_t s_rim[4][SRCAP], s_trail[4][STCAP + 1];      // (+1: the chain is pruned in place and ends one past the trail)
    __shared__ double s_rimz[4][SRCAP];
    __shared__ int32_t s_flist[4][64];
    __shared__ uint16_t s_holes[4][SRCAP];
    __shared__ double s_rows[4][2 * SWIN];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#ifdef PYDEM_PATHS_STATIC
    const int q = blockIdx.x * 4 + wv;
    if (q >= nslots) return;
    if (A.tier[q] > 0) return;       // known to leave the small window: its medium / large-window simulation runs beside this kernel
    if (kept_simulation(A, q, lane)) return;
    simulate_pit<SWIN, SRCAP>(A, q, s_seen[wv], s_fresh[wv], s_rim[wv], s_rimz[wv], s_holes[wv], s_flist[wv], s_trail[wv], STCAP, s_rows[wv]);
#else
    // persistent wavefronts take the next slot from a counter (flags[3], cleared by k_paths_slots): a simulation lasts 1 .. 300
    // iterations, and with four fixed slots per workgroup its LDS waits for the longest of the four
    // (PYDEM_PATHS_CHUNK slots per claim: the counter is ONE address, and from the second round on most slots are a look at a kept
    // simulation -- a claim per slot made the counter the bound of those rounds)
    for (;;) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&A.flags[3], PYDEM_PATHS_CHUNK);
        base = __shfl(base, 0);
        if (base >= nslots) break;
        const int stop = base + PYDEM_PATHS_CHUNK < nslots ? base + PYDEM_PATHS_CHUNK : nslots;
        for (int q = base; q < stop; q++) {
            if (A.tier[q] > 0) continue;     // known to leave the small window: its medium / large-window simulation runs beside this kernel
#ifdef PYDEM_PATHS_PROF
            {
                const long long tk0_ = wall_clock64();
                const bool kept_ = kept_simulation(A, q, lane);
                if (lane == 0) { atomicAdd(&g_paths_prof[0][q & 63][11], (unsigned long long)(wall_clock64() - tk0_)); atomicAdd(&g_paths_prof[0][q & 63][12], 1ull); }
                if (kept_) continue;
            }
#else
            if (kept_simulation(A, q, lane)) continue;
#endif
            simulate_pit<SWIN, SRCAP>(A, q, s_seen[wv], s_fresh[wv], s_rim[wv], s_rimz[wv], s_holes[wv], s_flist[wv], s_trail[wv], STCAP, s_rows[wv]);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
    }
#endif
}

// pits that left the small window: one wavefront per workgroup, the window in dynamic LDS, the trail in global scratch.
// Two sizes: the large window holds whatever the reference's 300 iterations can reach and fills the LDS of a CU; most
// plateau pits need far less, and the medium window lets three of them share a CU.  A simulation that leaves the medium
// window closes the round for every pit from it on (flags[2]) and runs in the large one from the next round.
template <int WIN, int RCAP>
__global__ __launch_bounds__(64) void k_paths_big(PathArgs A, const int32_t *__restrict__ slots, const int32_t *__restrict__ qidx, int nslots,
                                                 int32_t *bigtrail, int64_t trail_cap, int close_round)
{
    extern __shared__ uint32_t dyn[];
    uint32_t *seen = dyn, *fresh = dyn + WIN * WIN / 32;
    double *rimz = (double *)(fresh + WIN * WIN / 32);
    int32_t *rim = (int32_t *)(rimz + RCAP);
    int32_t *flist = rim + RCAP;
    uint16_t *holes = (uint16_t *)(flist + 64);
    const int q = blockIdx.x;
    if (q >= nslots) return;
    const int slot = slots[q];
    if (kept_simulation(A, slot, (int)threadIdx.x)) return;
    simulate_pit<WIN, RCAP>(A, slot, seen, fresh, rim, rimz, holes, flist, bigtrail + (int64_t)qidx[q] * trail_cap, (int)trail_cap);
    if (close_round && threadIdx.x == 0 && A.status[slot] == ST_OVERFLOW) atomicMin(&A.flags[2], A.window[slot]);
}

// Which pits commit.  Pit k saw what the sequential loop would have shown it when
//   (1) no earlier pending pit writes a cell k read, and (2) no earlier pending pit read a cell k writes (k_paths_tentative);
//   (3) no earlier pit that stays pending after this round has a footprint that overlaps k's (k_paths_blocked marks them,
//       k_paths_commit checks): such a pit simulates again on a changed surface and may then write where it only read before.
// k_limit: the first pit of the order that could not be simulated in this round -- nobody after it may commit.
// What the rules cannot exclude is caught when it happens: a simulation that reads a cell written by a later pit
// (simulate_pit), or a commit that writes a cell a later pit has already read (here), raises flags[0].
__global__ __launch_bounds__(256) void k_paths_tentative(PathArgs A, int k_limit)
{
    const int lane = (int)(threadIdx.x & 63);
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= A.nw) return;
    const int k = A.window[slot];
    const int st = A.status[slot];
    bool ok = (st == ST_FAILED || st == ST_PATH) && k < k_limit && k < A.flags[2];
    if (ok) {
        const int32_t *F = A.Fp[slot];
        const int32_t *C = A.Cp[slot];
        const int nF = A.nF[slot], nC = A.nC[slot];
        for (int q = lane; q < nF; q += 64) if (A.wown[F[q]] < k) ok = false;
        for (int q = lane; q < nC; q += 64) if (A.rown[C[q]] != k) ok = false;
        ok = !__any(!ok);
    }
    if (lane == 0) A.tent[slot] = ok;
}

__global__ __launch_bounds__(256) void k_paths_blocked(PathArgs A)
{
    const int lane = (int)(threadIdx.x & 63);
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= A.nw) return;
    const int st = A.status[slot];
    if ((st != ST_FAILED && st != ST_PATH) || A.tent[slot]) return;
    const int k = A.window[slot];
    const int32_t *F = A.Fp[slot];
    for (int q = lane; q < A.nF[slot]; q += 64) atomicMin(&A.bown[F[q]], k);
}

__global__ __launch_bounds__(256) void k_paths_commit(PathArgs A, int32_t *done, int32_t *counts)
{
    const int lane = (int)(threadIdx.x & 63);
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= A.nw) return;
    if (!A.tent[slot]) return;
    const int k = A.window[slot];
    const int st = A.status[slot];
    const int32_t *F = A.Fp[slot];
    const int32_t *C = A.Cp[slot];
    const int nF = A.nF[slot], nC = A.nC[slot];
    bool ok = true, clash = false;
    for (int q = lane; q < nF; q += 64) if (A.bown[F[q]] < k) ok = false;
    ok = !__any(!ok);
    if (!ok) return;
    for (int q = lane; q < nC; q += 64) if (A.rstamp[C[q]] > k) clash = true;     // a later pit has read what this one rewrites
    if (__any(clash)) { if (lane == 0) atomicOr(&A.flags[0], 1); return; }
    if (st == ST_PATH) {
        const double *CV = A.CVp[slot];
        for (int q = lane; q < nC; q += 64) { A.e[C[q]] = CV[q]; atomicMax(&A.wstamp[C[q]], k); A.wround[C[q]] = A.round; }
    }
    for (int q = lane; q < nF; q += 64) atomicMax(&A.rstamp[F[q]], k);
    if (lane == 0) {
        done[slot] = 1;
        // (single addresses: one atomic per committed pit on each of them kept their L2 channel busy for the length of the kernel
        // -- 48 k commits per round.  Nobody reads a count of commits; the maximum is only sent when it would change what a
        // plain load sees)
        if (st == ST_FAILED) atomicAdd(&counts[1], 1);
        const int32_t its = A.iters[slot];
        if (its > __hip_atomic_load(&counts[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&counts[2], its);
    }
}

__global__ __launch_bounds__(256) void k_paths_release(PathArgs A)
{
    const int lane = (int)(threadIdx.x & 63);
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= A.nw) return;
    const int st = A.status[slot];
    if (st != ST_FAILED && st != ST_PATH) return;
    const int32_t *F = A.Fp[slot];
    const int32_t *C = A.Cp[slot];
    for (int q = lane; q < A.nF[slot]; q += 64) { A.rown[F[q]] = 0x7FFFFFFF; A.bown[F[q]] = 0x7FFFFFFF; }
    for (int q = lane; q < A.nC[slot]; q += 64) A.wown[C[q]] = 0x7FFFFFFF;
}

// slot tables of a round: small slots point into the per-slot arrays
__global__ void k_paths_slots(PathArgs A, int32_t *F, int32_t *C, double *CV, int fcap, int ccap)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) { A.flags[2] = 0x7FFFFFFF; A.flags[3] = 0; }
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < A.nw; s += gridDim.x * blockDim.x) {
        const int64_t h = A.home[s];
        A.Fp[s] = F + h * fcap; A.Cp[s] = C + h * ccap; A.CVp[s] = CV + h * ccap;
        A.fcap[s] = fcap; A.ccap[s] = ccap;
        const int src = A.src[s];
        if (src >= 0) {
            A.status[s] = A.o_status[src]; A.nF[s] = A.o_nF[src]; A.nC[s] = A.o_nC[src]; A.iters[s] = A.o_iters[src];
            A.simround[s] = A.o_simround[src];
        } else {
            A.status[s] = ST_PENDING; A.nF[s] = 0; A.nC[s] = 0; A.iters[s] = 0; A.simround[s] = -1;
        }
    }
}

// keep: the blocks belong to their pits (medium window) -- a slot that carries a finished simulation keeps its state
__global__ void k_paths_bigslots(PathArgs A, const int32_t *slots, const int32_t *qidx, int nb, int32_t *F, int32_t *C, double *CV, int64_t fcap, int64_t ccap, int keep)
{
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nb; q += gridDim.x * blockDim.x) {
        const int s = slots[q];
        const int64_t blk = qidx[q];             // block of the scratch arrays
        A.Fp[s] = F + blk * fcap; A.Cp[s] = C + blk * ccap; A.CVp[s] = CV + blk * ccap;
        A.fcap[s] = (int32_t)fcap; A.ccap[s] = (int32_t)ccap;
        if (!(keep && A.simround[s] > 0)) { A.status[s] = ST_PENDING; A.simround[s] = -1; }
    }
}

// strict local minima above (or not at) sea level (:444-449): minimum_filter over the 8 neighbours > e
__global__ __launch_bounds__(256) void k_paths_pits(const double *__restrict__ e, int n, int m, int below_sea, uint8_t *mask, int32_t *nan_count)
{
    const int64_t NN = (int64_t)n * m;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < NN; c += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(c / m), j = (int)(c - (int64_t)i * m);
        const double z = e[c];
        (void)nan_count;
        // scipy's 'reflect' border mirrors the cell itself into the footprint on the array edge: e > e is false there.
        // The footprint filter (NI_MinOrMaxFilter) starts with its FIRST element -- the north-west neighbour -- and replaces
        // it by every later one that compares smaller: a NaN in first place stays (NaN > e is false: no pit), a NaN anywhere
        // else is never taken.  The same loop, value for value (no-data tiles; without NaN it is the plain minimum).
        bool low = false;
        if (!(i == 0 || j == 0 || i == n - 1 || j == m - 1)) {
            double tmp = e[c - m - 1];
            for (int d = 1; d < 9; d++) {
                if (d == 4) continue;
                const double v = e[c + (d / 3 - 1) * m + (d % 3 - 1)];
                if (v < tmp) tmp = v;
            }
            low = tmp > z;
        }
        mask[c] = low && (below_sea ? (z != 0.0) : (z > 0.0));
    }
}

__global__ void k_fill_i32(int32_t *p, int64_t n, int32_t v)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

__global__ void k_gather_f64(const double *__restrict__ e, const int32_t *__restrict__ ids, int32_t n, double *out)
{
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = e[ids[i]];
}

#include "ccl.h"

// a buffer carved out of the device's conditioning arena for the duration of the call (internal.h: ArenaLease)
struct Buf {
    void *p = nullptr;
    int get(ArenaLease &L, size_t bytes) { p = arena_take(&L, bytes ? bytes : 8); return p ? 0 : -1; }
};

int gridp(int64_t work, int cap) { const int64_t g = cdiv(work, 256); return (int)(g < cap ? (g > 0 ? g : 1) : cap); }

}  // namespace

// step 1: the pits of the resident surface.  *npits = number of strict minima (no-data cells compare like numpy's NaN).
// The caller reads their cells and elevations (stage_pit_paths_candidates), sorts them like the reference
// (np.argsort) and passes the order to stage_pit_paths.
int stage_pit_candidates(pydem_tile *t, int below_sea, int64_t *npits)
{
    const int n = (int)t->n, m = (int)t->m;
    PYDEM_TRY(tile_alloc(t, &t->flat0, (size_t)t->NN));
    PYDEM_TRY(tile_alloc(t, &t->flatlist, (size_t)t->NN));
    HIP_TRY(hipMemsetAsync(t->counters, 0, 16 * sizeof(int32_t), t->stream));
    hipLaunchKernelGGL(k_paths_pits, dim3(gridp(t->NN, 8192)), dim3(256), 0, t->stream, t->elev, n, m, below_sea, t->flat0, t->counters + 8);
    hipLaunchKernelGGL(k_compact_flats, dim3(gridp(t->NN, 4096)), dim3(256), 0, t->stream, t->flat0, t->NN, t->flatlist, t->counters);
    HIP_TRY(hipMemcpyAsync(t->h_counters, t->counters, 16 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(t->stream));
    *npits = t->h_counters[0];
    return 0;
}

// cells (ascending) and elevations of the candidates found by stage_pit_candidates
int stage_pit_candidates_read(pydem_tile *t, int64_t npits, int32_t *cells, double *elev)
{
    if (npits <= 0) return 0;
    ArenaLease lease;
    PYDEM_TRY(arena_acquire(t->device, &lease));
    // the compaction emits blocks out of order: the ids are sorted on the device (they were sorted on the host: 8-10 ms of
    // std::sort and two more transfers for the 341 k candidates of the 8192^2 SRTM-like tile), the elevations gathered in that
    // order, and both leave through the tile's pinned staging buffer in one go
    Buf sorted, vals, scratch;
    size_t sort_bytes = 0;
    HIP_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, sort_bytes, (const int32_t *)nullptr, (int32_t *)nullptr, (int)npits, 0, 32, t->stream));
    PYDEM_TRY(sorted.get(lease, (size_t)npits * 4)); PYDEM_TRY(vals.get(lease, (size_t)npits * 8)); PYDEM_TRY(scratch.get(lease, sort_bytes + 256));
    HIP_TRY(hipcub::DeviceRadixSort::SortKeys(scratch.p, sort_bytes, (const int32_t *)t->flatlist, (int32_t *)sorted.p, (int)npits, 0, 32, t->stream));
    hipLaunchKernelGGL(k_gather_f64, dim3(gridp(npits, 2048)), dim3(256), 0, t->stream, t->elev, (const int32_t *)sorted.p, (int32_t)npits, (double *)vals.p);
    void *pin = nullptr;
    PYDEM_TRY(tile_pinned(t, (size_t)npits * 12, &pin));
    HIP_TRY(hipMemcpyAsync(pin, vals.p, (size_t)npits * 8, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipMemcpyAsync((char *)pin + (size_t)npits * 8, sorted.p, (size_t)npits * 4, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(t->stream));
    memcpy(cells, (char *)pin + (size_t)npits * 8, (size_t)npits * 4);
    memcpy(elev, pin, (size_t)npits * 8);
    return 0;
}

// step 2: the paths, in the given order.  Returns 0 (done), 1 (the caller must use the host loop on the ORIGINAL surface,
// which this call has restored) or a negative error.
int stage_pit_paths(pydem_tile *t, const int32_t *order_host, int64_t npits, int max_iter, int max_dist, double max_dist_XY,
                    int64_t *n_failed, int64_t *iter_used, int64_t *rounds_out, int dtype_mode)
{
    const int n = (int)t->n, m = (int)t->m;
    *n_failed = 0; *iter_used = 0; if (rounds_out) *rounds_out = 0;
    if (npits == 0) return 0;
    if (npits >= (1ll << 30)) { pydem_set_error("too many pits"); return -2; }
    if (!t->spacing_set) { pydem_set_error("pit drain paths: call pydem_tile_set_spacing first"); return -3; }
    if (max_iter > 300) return 1;                             // the large window is sized for the reference's 300 iterations
    // speculation window and large-window simulations per round: measured on the 8192^2 SRTM-like tile (341 090 pits, 4885 of
    // them plateau pits): 32768 / 256 -> 56 rounds, 247 ms; 131072 / 2048 -> 26 rounds, 170 ms; larger does not pay.  With the medium
    // window (its scratch blocks are 65 k entries instead of 410 k): 131072 / 4096 is 3 ms faster than / 2048, 8192 no better.
    // Round 5, with simulations that outlive their round (a waiting pit costs a look at its footprint, not a simulation): the
    // whole order in the first window, 524288 / 8192 -> 24 rounds, 55-57 ms against 60-62 with 131072 / 4096 on the same box
    static int win_cap = -1, big_env = -1;
    if (win_cap < 0) { const char *e = getenv("PYDEM_PATHS_WINDOW"); win_cap = e ? atoi(e) : 524288; if (win_cap < 64) win_cap = 64; }
    if (big_env < 0) { const char *e = getenv("PYDEM_PATHS_BIG"); big_env = e ? atoi(e) : 8192; if (big_env < 1) big_env = 1; }
    int big_max = big_env;
    struct timespec ts0; clock_gettime(CLOCK_MONOTONIC, &ts0);
    const int W = (int)(npits < win_cap ? npits : win_cap);
    const int FCAP = STCAP + SRCAP, CCAP = STCAP + 1;
    const int64_t BIGF = std::min<int64_t>((int64_t)BWIN * BWIN, t->NN);   // footprint / trail capacity of a large-window simulation
    if ((int64_t)big_max > npits) big_max = (int)npits;
    const int64_t MIDF = std::min<int64_t>((int64_t)MWIN * MWIN, t->NN);    // ... of a medium-window one
    ArenaLease lease;
    PYDEM_TRY(arena_acquire(t->device, &lease));
    Buf b_bown, b_rstamp, b_tent, b_tier, b_wround, b_home, b_src, b_status2, b_nF2, b_nC2, b_iters2, b_sim, b_sim2;
    Buf b_order, b_window, b_rown, b_wown, b_stamp, b_status, b_nF, b_nC, b_iters, b_F, b_C, b_CV, b_flags, b_done, b_counts, b_slots,
        b_backup, b_bigtrail, b_bigF, b_bigC, b_bigCV, b_midtrail, b_midF, b_midC, b_midCV, b_Fp, b_Cp, b_CVp, b_fcap, b_ccap;
    PYDEM_TRY(b_order.get(lease, (size_t)npits * 4)); PYDEM_TRY(b_window.get(lease, (size_t)W * 4));
    PYDEM_TRY(b_rown.get(lease, (size_t)t->NN * 4)); PYDEM_TRY(b_wown.get(lease, (size_t)t->NN * 4)); PYDEM_TRY(b_stamp.get(lease, (size_t)t->NN * 4));
    PYDEM_TRY(b_bown.get(lease, (size_t)t->NN * 4)); PYDEM_TRY(b_rstamp.get(lease, (size_t)t->NN * 4)); PYDEM_TRY(b_tent.get(lease, (size_t)W * 4)); PYDEM_TRY(b_tier.get(lease, (size_t)W * 4));
    PYDEM_TRY(b_status.get(lease, (size_t)W * 4)); PYDEM_TRY(b_nF.get(lease, (size_t)W * 4)); PYDEM_TRY(b_nC.get(lease, (size_t)W * 4)); PYDEM_TRY(b_iters.get(lease, (size_t)W * 4));
    PYDEM_TRY(b_status2.get(lease, (size_t)W * 4)); PYDEM_TRY(b_nF2.get(lease, (size_t)W * 4)); PYDEM_TRY(b_nC2.get(lease, (size_t)W * 4)); PYDEM_TRY(b_iters2.get(lease, (size_t)W * 4));
    PYDEM_TRY(b_sim.get(lease, (size_t)W * 4)); PYDEM_TRY(b_sim2.get(lease, (size_t)W * 4)); PYDEM_TRY(b_home.get(lease, (size_t)W * 4)); PYDEM_TRY(b_src.get(lease, (size_t)W * 4));
    PYDEM_TRY(b_wround.get(lease, (size_t)t->NN * 4));
    PYDEM_TRY(b_F.get(lease, (size_t)W * FCAP * 4)); PYDEM_TRY(b_C.get(lease, (size_t)W * CCAP * 4)); PYDEM_TRY(b_CV.get(lease, (size_t)W * CCAP * 8));
    PYDEM_TRY(b_Fp.get(lease, (size_t)W * 8)); PYDEM_TRY(b_Cp.get(lease, (size_t)W * 8)); PYDEM_TRY(b_CVp.get(lease, (size_t)W * 8));
    PYDEM_TRY(b_fcap.get(lease, (size_t)W * 4)); PYDEM_TRY(b_ccap.get(lease, (size_t)W * 4));
    PYDEM_TRY(b_flags.get(lease, 32)); PYDEM_TRY(b_done.get(lease, (size_t)W * 4)); PYDEM_TRY(b_counts.get(lease, 16)); PYDEM_TRY(b_slots.get(lease, (size_t)W * 16 + 64));
    PYDEM_TRY(b_backup.get(lease, (size_t)t->NN * 8));
    HIP_TRY(hipMemcpyAsync(b_backup.p, t->elev, (size_t)t->NN * 8, hipMemcpyDeviceToDevice, t->stream));
    // pinned staging: [order | window | status | done | tiers | (slot, block) lists of the medium / large simulations: known, fresh, re-run]
    void *pin_v = nullptr;
    PYDEM_TRY(tile_pinned(t, ((size_t)npits + 10 * (size_t)W + 64) * 4, &pin_v));
    int32_t *pin_order = (int32_t *)pin_v, *pin_window = pin_order + npits, *pin_status = pin_window + W, *pin_done = pin_status + W, *pin_tier = pin_done + W,
            *pin_home = pin_tier + W, *pin_src = pin_home + W, *pin_slots = pin_src + W;
    memcpy(pin_order, order_host, (size_t)npits * 4);
    HIP_TRY(hipMemcpyAsync(b_order.p, pin_order, (size_t)npits * 4, hipMemcpyHostToDevice, t->stream));
    const int gN = gridp(t->NN, 8192);
    hipLaunchKernelGGL(k_fill_i32, dim3(gN), dim3(256), 0, t->stream, (int32_t *)b_rown.p, t->NN, 0x7FFFFFFF);
    hipLaunchKernelGGL(k_fill_i32, dim3(gN), dim3(256), 0, t->stream, (int32_t *)b_wown.p, t->NN, 0x7FFFFFFF);
    hipLaunchKernelGGL(k_fill_i32, dim3(gN), dim3(256), 0, t->stream, (int32_t *)b_stamp.p, t->NN, -1);
    hipLaunchKernelGGL(k_fill_i32, dim3(gN), dim3(256), 0, t->stream, (int32_t *)b_rstamp.p, t->NN, -1);
    hipLaunchKernelGGL(k_fill_i32, dim3(gN), dim3(256), 0, t->stream, (int32_t *)b_bown.p, t->NN, 0x7FFFFFFF);
    HIP_TRY(hipMemsetAsync(b_flags.p, 0, 32, t->stream));
    HIP_TRY(hipMemsetAsync(b_counts.p, 0, 16, t->stream));
    HIP_TRY(hipMemsetAsync(b_wround.p, 0, (size_t)t->NN * 4, t->stream));
    PathArgs A;
    A.e = t->elev; A.n = n; A.m = m; A.order = (const int32_t *)b_order.p; A.window = (const int32_t *)b_window.p; A.nw = 0;
    A.dX = t->dX; A.dY = t->dY; A.ndX = n - 1; A.max_iter = max_iter; A.max_dist = max_dist; A.max_dist_XY = max_dist_XY;
    A.dtype_mode = dtype_mode;
    A.mmagic = (((unsigned long long)1 << 63) + (unsigned long long)m - 1) / (unsigned long long)m;
    A.rown = (int32_t *)b_rown.p; A.wown = (int32_t *)b_wown.p; A.wstamp = (int32_t *)b_stamp.p;
    A.bown = (int32_t *)b_bown.p; A.rstamp = (int32_t *)b_rstamp.p; A.tent = (int32_t *)b_tent.p;
    A.status = (int32_t *)b_status.p; A.nF = (int32_t *)b_nF.p; A.nC = (int32_t *)b_nC.p; A.iters = (int32_t *)b_iters.p;
    A.Fp = (int32_t **)b_Fp.p; A.Cp = (int32_t **)b_Cp.p; A.CVp = (double **)b_CVp.p; A.fcap = (int32_t *)b_fcap.p; A.ccap = (int32_t *)b_ccap.p;
    A.flags = (int32_t *)b_flags.p; A.tier = (const int32_t *)b_tier.p;
    A.wround = (int32_t *)b_wround.p; A.home = (const int32_t *)b_home.p; A.src = (const int32_t *)b_src.p; A.round = 0; A.count_kept = getenv("PYDEM_PATHS_DEBUG") != nullptr;
    // per-slot state in two sets: a round's set is filled from the previous round's (k_paths_slots) for the pits that wait
    int32_t *const set_status[2] = {(int32_t *)b_status.p, (int32_t *)b_status2.p}, *const set_nF[2] = {(int32_t *)b_nF.p, (int32_t *)b_nF2.p},
                  *const set_nC[2] = {(int32_t *)b_nC.p, (int32_t *)b_nC2.p}, *const set_iters[2] = {(int32_t *)b_iters.p, (int32_t *)b_iters2.p},
                  *const set_sim[2] = {(int32_t *)b_sim.p, (int32_t *)b_sim2.p};
    static int keep_env = -1;         // PYDEM_PATHS_KEEP=0: every waiting pit is simulated again every round (the behaviour up to round 4)
    if (keep_env < 0) { const char *e = getenv("PYDEM_PATHS_KEEP"); keep_env = e ? atoi(e) : 1; }
    // what belongs to a pending pit (parallel to `pending`): its block of the small-window arrays, its block of the medium-window
    // pool (-1: none), its slot in the previous round (-1: none)
    // The host lists of the rounds live as long as the thread: 341 k pits make each of them 1.3 MB, and a fresh allocation of that
    // size is a mapping the first round pays for page by page (1 ms of zero fill in round 1, more in the setup).
    static thread_local std::vector<int32_t> p_home, p_block, p_src, w_home, w_block, w_src, free_home, free_block, pending, win, h_status, h_done, big;
    static thread_local std::vector<uint8_t> tier, proven;
    for (std::vector<int32_t> *v : {&pending, &win, &p_home, &p_block, &p_src, &w_home, &w_block, &w_src, &free_block, &big}) { v->clear(); v->reserve((size_t)W); }
    free_home.resize((size_t)W); h_status.resize((size_t)W); h_done.resize((size_t)W);
    for (int h = 0; h < W; h++) free_home[(size_t)h] = W - 1 - h;
    tier.assign((size_t)npits, 0);                          // what earlier rounds learned: 1 = the pit leaves the small window, 2 = the medium one too
    proven.assign((size_t)npits, 0);                        // a medium-window simulation of the pit has completed
    int64_t next = 0;                 // first pit of the order that has not entered a window yet
    int64_t rounds = 0, big_runs = 0, small_runs = 0;
    double ms_small = 0, ms_big = 0, ms_commit = 0, ms_prep = 0, ms_post = 0;
    const bool prof = getenv("PYDEM_PATHS_DEBUG") != nullptr;
    auto now_ms = []() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    bool fallback = false;
    const size_t big_lds = (size_t)2 * (BWIN * BWIN / 8) + (size_t)BRCAP * 12 + 64 * 4 + (size_t)BRCAP * 2;     // + the hole stack
    const size_t mid_lds = (size_t)2 * (MWIN * MWIN / 8) + (size_t)MRCAP * 12 + 64 * 4 + (size_t)MRCAP * 2;
    static int use_mid = -1;          // PYDEM_PATHS_MID=0: every pit that leaves the small window goes straight to the large one
    if (use_mid < 0) { const char *e = getenv("PYDEM_PATHS_MID"); use_mid = e ? atoi(e) : 1; }
    static int large_env = -1;        // large-window simulations per round when the medium window exists (they are rare: 77 of 14 k on the SRTM-like tile)
    if (large_env < 0) { const char *e = getenv("PYDEM_PATHS_LARGE"); large_env = e ? atoi(e) : 512; if (large_env < 1) large_env = 1; }
    const int large_max = use_mid ? (big_max < large_env ? big_max : large_env) : big_max;
    int64_t mid_runs = 0;
    bool big_ready = false;
    HIP_TRY(hipStreamSynchronize(t->stream));
    const double ms_setup = now_ms() - (ts0.tv_sec * 1e3 + ts0.tv_nsec * 1e-6);
    while (!pending.empty() || next < npits) {
        const double t_p = now_ms();
        if ((int)pending.size() < W && next < npits) {          // (in bulk: a push_back per list and pit was 4.4 ms of the first round)
            const size_t have = pending.size(), room = (size_t)W - have, left = (size_t)(npits - next), add = room < left ? room : left;
            pending.resize(have + add); p_home.resize(have + add); p_block.resize(have + add, -1); p_src.resize(have + add, -1);
            int32_t *pd = pending.data() + have, *ph = p_home.data() + have;
            const int32_t *fh = free_home.data() + free_home.size() - 1;
            for (size_t i = 0; i < add; i++) { pd[i] = (int32_t)(next + (int64_t)i); ph[i] = *(fh - i); }
            free_home.resize(free_home.size() - add);
            next += (int64_t)add;
        }
        const double t_p1 = now_ms();
        // only as many large-window pits as one launch holds can take part in a round, and nobody after the first one
        // left out may commit (k_limit below): the pits behind it are not worth simulating this round
        int nw = (int)pending.size();
        {
            int seen_big = 0;
            for (int s2 = 0; s2 < nw; s2++)
                if (tier[(size_t)pending[(size_t)s2]] && ++seen_big > big_max) { nw = s2; break; }
        }
        A.nw = nw;
        A.round = (int)rounds + 1;
        {
            const int cur = (int)(rounds & 1);
            A.status = set_status[cur]; A.nF = set_nF[cur]; A.nC = set_nC[cur]; A.iters = set_iters[cur]; A.simround = set_sim[cur];
            A.o_status = set_status[cur ^ 1]; A.o_nF = set_nF[cur ^ 1]; A.o_nC = set_nC[cur ^ 1]; A.o_iters = set_iters[cur ^ 1]; A.o_simround = set_sim[cur ^ 1];
        }
        // a pit that sits this round out gives its medium-window block back (the round's participants need at most big_max)
        for (size_t s2 = (size_t)nw; s2 < pending.size(); s2++)
            if (p_block[s2] >= 0) { free_block.push_back(p_block[s2]); p_block[s2] = -1; }
        const double t_p2 = now_ms();
        memcpy(pin_window, pending.data(), (size_t)nw * 4);
        memcpy(pin_home, p_home.data(), (size_t)nw * 4);
        {
            const int32_t *pd = pending.data(), *ps = p_src.data();
            const uint8_t *tr = tier.data();
            for (int s2 = 0; s2 < nw; s2++) { pin_tier[s2] = tr[pd[s2]]; pin_src[s2] = keep_env ? ps[s2] : -1; }
        }
        const double t_p3 = now_ms();
        HIP_TRY(hipMemcpyAsync(b_window.p, pin_window, (size_t)nw * 4, hipMemcpyHostToDevice, t->stream));
        HIP_TRY(hipMemcpyAsync(b_tier.p, pin_tier, (size_t)nw * 4, hipMemcpyHostToDevice, t->stream));
        HIP_TRY(hipMemcpyAsync(b_home.p, pin_home, (size_t)nw * 4, hipMemcpyHostToDevice, t->stream));
        HIP_TRY(hipMemcpyAsync(b_src.p, pin_src, (size_t)nw * 4, hipMemcpyHostToDevice, t->stream));
        HIP_TRY(hipMemsetAsync(b_done.p, 0, (size_t)nw * 4, t->stream));
        const double t_a = now_ms();
        ms_prep += t_a - t_p;
        if (prof && rounds < 4) fprintf(stderr, "    host before: fill %.2f, cut %.2f, staging %.2f, copies issued %.2f ms\n", t_p1 - t_p, t_p2 - t_p1, t_p3 - t_p2, t_a - t_p3);
        small_runs += nw;
        hipLaunchKernelGGL(k_paths_slots, dim3(gridp(nw, 256)), dim3(256), 0, t->stream, A, (int32_t *)b_F.p, (int32_t *)b_C.p, (double *)b_CV.p, FCAP, CCAP);
        // the medium / large-window simulations of a round: entry q of `big` owns block q of the scratch arrays.  Those of
        // the pits whose tier is known run on the side stream BESIDE the small-window kernel (all simulations of a round read
        // the same committed surface); what the small-window kernel newly sends on follows on the main stream.
        int k_limit = 0x7FFFFFFF, large_used = 0;      // (blocks of the large-window pool handed out in this round; a medium-window block belongs to its pit)
        auto launch_large = [&](const std::vector<int> &qs, size_t stage_at, hipStream_t st, int close_round) -> int {
            if (!big_ready) {
                // scratch blocks (trail, footprint, chain, chain values) sized by window: one pool per tier
                PYDEM_TRY(b_bigtrail.get(lease, (size_t)large_max * BIGF * 4));
                PYDEM_TRY(b_bigF.get(lease, (size_t)large_max * BIGF * 4));
                PYDEM_TRY(b_bigC.get(lease, (size_t)large_max * (BIGF + 1) * 4));
                PYDEM_TRY(b_bigCV.get(lease, (size_t)large_max * (BIGF + 1) * 8));
                if (use_mid) {
                    for (int bq = big_max - 1; bq >= 0; bq--) free_block.push_back(bq);
                    PYDEM_TRY(b_midtrail.get(lease, (size_t)big_max * MIDF * 4));
                    PYDEM_TRY(b_midF.get(lease, (size_t)big_max * MIDF * 4));
                    PYDEM_TRY(b_midC.get(lease, (size_t)big_max * (MIDF + 1) * 4));
                    PYDEM_TRY(b_midCV.get(lease, (size_t)big_max * (MIDF + 1) * 8));
                }
                HIP_TRY(hipFuncSetAttribute((const void *)k_paths_big<BWIN, BRCAP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)big_lds));
                HIP_TRY(hipFuncSetAttribute((const void *)k_paths_big<MWIN, MRCAP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mid_lds));
                big_ready = true;
            }
            const int nb = (int)qs.size();
            if (nb == 0) return 0;
            // staging: [(slot, block) of the medium ones | (slot, block) of the large ones | (slot, 0) of the large ones left out]
            // A round has room for large_max large-window simulations; a pit left out is not simulated and closes the round for
            // everybody from it on (k_limit), like the pits beyond big_max.
            int nm = 0, nl = 0, nd = 0;
            for (int q : qs) if (tier[(size_t)pending[(size_t)big[(size_t)q]]] == 1) nm++;
            int nl_room = large_max - large_used; if (nl_room < 0) nl_room = 0;
            const int nl_all = nb - nm, nl_take = nl_all < nl_room ? nl_all : nl_room;
            int32_t *ms = pin_slots + 2 * stage_at, *mq = ms + nm, *ls = mq + nm, *lq = ls + nl_take, *ds = lq + nl_take, *dq = ds + (nl_all - nl_take);
            nm = 0;
            for (int q : qs) {
                const int slot = big[(size_t)q], kk = pending[(size_t)slot];
                if (tier[(size_t)kk] == 1) {
                    if (p_block[(size_t)slot] < 0) {
                        if (free_block.empty()) { pydem_set_error("pit drain paths: medium-window pool exhausted"); return -5; }
                        p_block[(size_t)slot] = free_block.back(); free_block.pop_back();
                    }
                    ms[nm] = slot; mq[nm++] = p_block[(size_t)slot];
                }
                else if (nl < nl_take) { ls[nl] = slot; lq[nl++] = large_used++; }
                else { ds[nd] = slot; dq[nd++] = 0; if (kk < k_limit) k_limit = kk; }
            }
            int32_t *d_ms = (int32_t *)b_slots.p + 2 * stage_at;
            HIP_TRY(hipMemcpyAsync(d_ms, ms, (size_t)nb * 8, hipMemcpyHostToDevice, st));
            const int32_t *d_mq = d_ms + nm, *d_ls = d_mq + nm, *d_lq = d_ls + nl, *d_ds = d_lq + nl, *d_dq = d_ds + nd;
            if (nm) hipLaunchKernelGGL(k_paths_bigslots, dim3(gridp(nm, 64)), dim3(256), 0, st, A, (const int32_t *)d_ms, d_mq, nm, (int32_t *)b_midF.p,
                                       (int32_t *)b_midC.p, (double *)b_midCV.p, MIDF, MIDF + 1, 1);
            if (nl) hipLaunchKernelGGL(k_paths_bigslots, dim3(gridp(nl, 64)), dim3(256), 0, st, A, d_ls, d_lq, nl, (int32_t *)b_bigF.p,
                                       (int32_t *)b_bigC.p, (double *)b_bigCV.p, BIGF, BIGF + 1, 0);
            if (nd) hipLaunchKernelGGL(k_paths_bigslots, dim3(gridp(nd, 64)), dim3(256), 0, st, A, d_ds, d_dq, nd, (int32_t *)b_bigF.p,      // (state "pending": not simulated)
                                       (int32_t *)b_bigC.p, (double *)b_bigCV.p, BIGF, BIGF + 1, 0);
            if (nl) hipLaunchKernelGGL((k_paths_big<BWIN, BRCAP>), dim3(nl), dim3(64), big_lds, st, A, d_ls, d_lq, nl, (int32_t *)b_bigtrail.p, BIGF, 0);
            if (nm) hipLaunchKernelGGL((k_paths_big<MWIN, MRCAP>), dim3(nm), dim3(64), mid_lds, st, A, (const int32_t *)d_ms, d_mq, nm, (int32_t *)b_midtrail.p, MIDF,
                                       close_round);
            big_runs += nl; mid_runs += nm;
            return 0;
        };
        big.clear();
        for (int s = 0; s < nw; s++) if (tier[(size_t)pending[(size_t)s]]) big.push_back(s);      // (at most big_max: the window was cut there)
        const int n_known = (int)big.size();
        // A pit whose medium-window simulation has never completed may leave that window too.  While such pits take part the
        // host looks at the states after the simulations and re-runs those in the large window IN THIS ROUND (one more
        // synchronisation; the first rounds); once every medium-window pit of a round is proven, a simulation that leaves the
        // window anyway (its surroundings changed) closes the round behind it (flags[2]) and moves up for the next one.
        auto unproven = [&](int q) { const int kk = pending[(size_t)big[(size_t)q]]; return tier[(size_t)kk] == 1 && !proven[(size_t)kk]; };
        bool check = false;
        for (int q = 0; q < n_known; q++) check = check || unproven(q);
        std::vector<int> qs;
        if (n_known) {
            for (int q = 0; q < n_known; q++) qs.push_back(q);
            HIP_TRY(hipEventRecord(t->ev_fork, t->stream));
            HIP_TRY(hipStreamWaitEvent(t->stream2, t->ev_fork, 0));
            // (whether the round is checked is only known after the small-window kernel; the known ones never close the round when unproven ones are among them)
            PYDEM_TRY(launch_large(qs, 0, t->stream2, check ? 0 : 1));
            HIP_TRY(hipEventRecord(t->ev_join, t->stream2));
        }
#ifdef PYDEM_PATHS_STATIC
        hipLaunchKernelGGL(k_paths_small, dim3((unsigned)cdiv(nw, 4)), dim3(256), 0, t->stream, A, nw);
#else
        hipLaunchKernelGGL(k_paths_small, dim3((unsigned)std::min<int64_t>(cdiv(nw, 4), 256 * PYDEM_SMALL_WAVES)), dim3(256), 0, t->stream, A, nw);
#endif
        HIP_TRY(hipMemcpyAsync(pin_status, A.status, (size_t)nw * 4, hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(t->stream));
        memcpy(h_status.data(), pin_status, (size_t)nw * 4);
        const double t_b = now_ms();
        ms_small += t_b - t_a;
        for (int s = 0; s < nw; s++) {               // (the status of a known pit is in the making on the side stream: not looked at here)
            uint8_t &tr = tier[(size_t)pending[(size_t)s]];
            if (tr == 0 && h_status[(size_t)s] == ST_OVERFLOW) { big.push_back(s); tr = use_mid ? 1 : 2; }
        }
        // as many medium / large-window simulations as one launch holds take part in this round; the first one left out
        // closes the round for everybody after it: a pit commits only when every earlier pending pit was simulated
        if ((int)big.size() > big_max) {
            for (size_t q = (size_t)big_max; q < big.size(); q++) { const int kk = pending[(size_t)big[q]]; if (kk < k_limit) k_limit = kk; }
            big.resize((size_t)big_max);
        }
        const int nb_all = (int)big.size();
        if (nb_all > n_known) {
            qs.clear();
            for (int q = n_known; q < nb_all; q++) { qs.push_back(q); check = check || unproven(q); }
            PYDEM_TRY(launch_large(qs, (size_t)n_known, t->stream, 0));      // (fresh ones are unproven: the round is checked)
        }
        if (n_known) HIP_TRY(hipStreamWaitEvent(t->stream, t->ev_join, 0));
        if (check) {
            HIP_TRY(hipMemcpyAsync(pin_status, A.status, (size_t)nw * 4, hipMemcpyDeviceToHost, t->stream));
            HIP_TRY(hipStreamSynchronize(t->stream));
            qs.clear();
            for (int q = 0; q < nb_all; q++) {
                const int kk = pending[(size_t)big[(size_t)q]];
                if (tier[(size_t)kk] != 1) continue;
                if (pin_status[big[(size_t)q]] == ST_OVERFLOW) {
                    tier[(size_t)kk] = 2; qs.push_back(q);
                    int32_t &blk = p_block[(size_t)big[(size_t)q]];
                    if (blk >= 0) { free_block.push_back(blk); blk = -1; }
                }
                else proven[(size_t)kk] = 1;
            }
            PYDEM_TRY(launch_large(qs, (size_t)nb_all, t->stream, 0));
        }
        if (prof) { HIP_TRY(hipStreamSynchronize(t->stream)); }
        const double t_c = now_ms();
        ms_big += t_c - t_b;
        hipLaunchKernelGGL(k_paths_tentative, dim3((unsigned)cdiv(nw, 4)), dim3(256), 0, t->stream, A, k_limit);
        hipLaunchKernelGGL(k_paths_blocked, dim3((unsigned)cdiv(nw, 4)), dim3(256), 0, t->stream, A);
        hipLaunchKernelGGL(k_paths_commit, dim3((unsigned)cdiv(nw, 4)), dim3(256), 0, t->stream, A, (int32_t *)b_done.p, (int32_t *)b_counts.p);
        hipLaunchKernelGGL(k_paths_release, dim3((unsigned)cdiv(nw, 4)), dim3(256), 0, t->stream, A);
        HIP_TRY(hipMemcpyAsync(pin_done, b_done.p, (size_t)nw * 4, hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipMemcpyAsync(pin_status, A.status, (size_t)nw * 4, hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipMemcpyAsync(t->h_counters, b_flags.p, 32, hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(t->stream));
        memcpy(h_done.data(), pin_done, (size_t)nw * 4);
        memcpy(h_status.data(), pin_status, (size_t)nw * 4);
        rounds++;
        const double t_d = now_ms();
        ms_commit += t_d - t_c;
        if (prof) {
            int ncommit = 0;
            for (int s2 = 0; s2 < nw; s2++) ncommit += h_done[(size_t)s2] ? 1 : 0;
            fprintf(stderr, "  round %lld: window %d, medium + large-window %d, committed %d; ms small %.2f large %.2f commit %.2f; kept so far %d, found stale %d\n", (long long)rounds, nw, (int)big.size(),
                    ncommit, t_b - t_a, t_c - t_b, now_ms() - t_c, t->h_counters[4], t->h_counters[5]);
        }
#ifdef PYDEM_PATHS_PROF
        {
            static unsigned long long h[2][64][16], last[2][16];
            if (rounds == 1) memset(last, 0, sizeof(last));
            HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_paths_prof), sizeof(h)));
            for (int c = 0; c < 2; c++) {
                unsigned long long v[16] = {0};
                for (int b = 0; b < 64; b++) for (int i = 0; i < 16; i++) { if (i == 13) v[i] = h[c][b][i] > v[i] ? h[c][b][i] : v[i]; else v[i] += h[c][b][i]; }
                fprintf(stderr, "    %s: %llu simulations, %llu iterations, %.1f ms of wavefront time (path %.1f, ring %.1f), kept checks %llu in %.1f ms; longest simulation so far %.3f ms\n", c ? "medium / large" : "small",
                        v[10] - last[c][10], v[7] - last[c][7], (v[8] - last[c][8]) * 1e-5, (v[6] - last[c][6]) * 1e-5, (v[4] - last[c][4]) * 1e-5, v[12] - last[c][12], (v[11] - last[c][11]) * 1e-5, v[13] * 1e-5);
                memcpy(last[c], v, sizeof(v));
            }
        }
#endif
        if (getenv("PYDEM_PATHS_DEBUG") && atoi(getenv("PYDEM_PATHS_DEBUG")) >= 2) {
            std::vector<int32_t> h_it((size_t)nw);
            HIP_TRY(hipMemcpy(h_it.data(), A.iters, (size_t)nw * 4, hipMemcpyDeviceToHost));
            for (int s2 = 0; s2 < nw; s2++)
                if (h_done[(size_t)s2]) fprintf(stderr, "  commit k=%d cell=%d status=%d iters=%d\n", pending[(size_t)s2], order_host[pending[(size_t)s2]],
                                                 h_status[(size_t)s2], h_it[(size_t)s2]);
        }
        if (t->h_counters[0] || t->h_counters[1]) { fallback = true; break; }
        bool stuck = false, escalated = false;
        for (int s : big)
            if (h_status[(size_t)s] == ST_OVERFLOW) {
                uint8_t &tr = tier[(size_t)pending[(size_t)s]];
                if (tr == 1) {                                   // left the medium window: the large one from the next round on
                    tr = 2; escalated = true;
                    if (p_block[(size_t)s] >= 0) { free_block.push_back(p_block[(size_t)s]); p_block[(size_t)s] = -1; }
                }
                else stuck = true;                               // does not even fit the large window
            }
        if (stuck) { fallback = true; break; }
        const double t_q1 = now_ms();
        win.resize(pending.size()); w_home.resize(pending.size()); w_block.resize(pending.size()); w_src.resize(pending.size());
        const double t_q2 = now_ms();
        size_t nk = 0;                                         // pits that stay
        {
            const int32_t *hd = h_done.data(), *pd = pending.data(), *ph = p_home.data(), *pb = p_block.data();
            int32_t *ow = win.data(), *oh = w_home.data(), *ob = w_block.data(), *os = w_src.data();
            // (every entry is written where the next survivor goes and the index moves on only for a survivor: the 40 / 60 branch on
            // "committed" was 4.5 ns per slot; homes go back on their stack only while pits still wait to enter a window)
            for (int s = 0; s < nw; s++) {
                const int32_t d = hd[s];
                ow[nk] = pd[s]; oh[nk] = ph[s]; ob[nk] = pb[s]; os[nk] = s;
                nk += d ? 0 : 1;
                if (d && pb[s] >= 0) free_block.push_back(pb[s]);
            }
            if (next < npits) for (int s = 0; s < nw; s++) if (hd[s]) free_home.push_back(ph[s]);
        }
        win.resize(nk); w_home.resize(nk); w_block.resize(nk); w_src.resize(nk);
        if ((int)win.size() == nw && !escalated) { fallback = true; break; }          // (cannot happen: the first pit always commits)
        for (size_t s = (size_t)nw; s < pending.size(); s++) {                        // the part of the window that sat this round out
            win.push_back(pending[s]); w_home.push_back(p_home[s]); w_block.push_back(p_block[s]); w_src.push_back(-1);
        }
        pending.swap(win); p_home.swap(w_home); p_block.swap(w_block); p_src.swap(w_src);
        ms_post += now_ms() - t_d;
        if (prof && rounds <= 4) fprintf(stderr, "    host after: checks %.2f, resize %.2f, compaction %.2f ms\n", t_q1 - t_d, t_q2 - t_q1, now_ms() - t_q2);
    }
    if (rounds_out) *rounds_out = rounds;
#ifdef PYDEM_PATHS_PROF
    {
        static unsigned long long h[2][64][16];
        HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_paths_prof), sizeof(h)));
        for (int c = 0; c < 2; c++) {
            unsigned long long v[16] = {0};
            for (int b = 0; b < 64; b++) for (int i = 0; i < 16; i++) v[i] += h[c][b][i];
            fprintf(stderr, "%s simulations: %llu, %llu iterations; 10 ns ticks summed: whole %llu = clear %llu, min scan %llu, fresh pass %llu, emit %llu, ring %llu, footprint %llu, path %llu, "
                            "reservations %llu; kept checks %llu in %llu ticks; of the path: outlet choice %llu, + pruning %llu\n", c ? "medium / large-window" : "small-window", v[10], v[7], v[8], v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[9], v[12], v[11], v[14], v[15]);
        }
        memset(h, 0, sizeof(h));
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_paths_prof), h, sizeof(h)));
    }
#endif
    if (getenv("PYDEM_PATHS_DEBUG"))
        fprintf(stderr, "pit drain paths: %lld pits, %lld rounds, %lld small-window, %lld medium-window and %lld large-window simulations; ms: small %.1f, medium + large %.1f, "
                        "commit %.1f, buffers %.1f, host before / after a round %.1f / %.1f, whole call %.1f%s\n", (long long)npits, (long long)rounds, (long long)small_runs, (long long)mid_runs, (long long)big_runs, ms_small, ms_big, ms_commit,
                ms_setup, ms_prep, ms_post, now_ms() - (ts0.tv_sec * 1e3 + ts0.tv_nsec * 1e-6), fallback ? " -> host loop" : "");
    if (fallback) {
        HIP_TRY(hipMemcpyAsync(t->elev, b_backup.p, (size_t)t->NN * 8, hipMemcpyDeviceToDevice, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        return 1;
    }
    HIP_TRY(hipMemcpyAsync(t->h_counters, b_counts.p, 16, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    *n_failed = t->h_counters[1];
    *iter_used = t->h_counters[2];
    return 0;
}
