// pits.hip -- pit -> drain assignment on the device.
//
// Replaces _mk_connectivity_pits (reference pydem/dem_processing.py:1269-1382) with its helpers
// utils.get_border_index (pydem/utils.py:313-340) and _get_dX_mean (:1993-1997).  The reference
// walks the pits one by one in Python (402 k border recomputations with np.setdiff1d at 1024^2);
// each pit is independent of the others, so they are solved in parallel in four tiers of growing
// window / border capacity (a region can grow by at most one cell of Chebyshev radius per iteration):
// a LANE per pit (16x16 window, 89 % of the pits), a WAVEFRONT per pit (128x128, then 256x256 for
// plateau terrain) and a WORKGROUP per pit with a 640x640 bitmap window -- large enough for
// drain_pits_max_iter <= 300 always.  Region and border live in LDS (bitmap + unordered border list),
// minima are lane-serial / DPP reductions, border updates use LDS atomics.  Integer work (which cells
// drain where) is exact; weights use numpy's pairwise summation order so they match the reference
// bit for bit.  The raw (pit, drain, weight) triplets are then filtered like _mk_adjacency_matrix
// (:1136-1137) and radix-sorted into the two side lists of the implicit graph.
#include "internal.h"
#include <hipcub/hipcub.hpp>
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

namespace {

// numpy pairwise sum (np.add.reduce on a contiguous float64 vector), see oracle/pydem_oracle.c.
// numpy recurses on halves above 128 elements; a recursive device function would give the kernels a
// dynamic stack (scratch, low occupancy), so the recursion is unrolled as a template of bounded depth
// (128 * 2^9 elements cover any tile height; slices here are <= drain_pits_max_dist = 32 by default).
__device__ __forceinline__ double np_pairwise_leaf(const double *a, int n)
{
    if (n < 8) {
        double res = 0.;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    }
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
        r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += a[i];
    return res;
}

// Slices longer than 128 (only possible when drain_pits_max_dist is disabled) follow numpy's
// recursive halving with an explicit stack kept in LDS (`stk`: 16 frames of {offset, length, stage,
// left value}); no device recursion, no dynamic stack.
struct PwFrame { int off, len, stage, pad; double lval; };

__device__ __forceinline__ double np_pairwise_sum(const double *a, int n, PwFrame *stk)
{
    if (n <= 128) return np_pairwise_leaf(a, n);
    int sp = 0;
    stk[0].off = 0; stk[0].len = n; stk[0].stage = 0;
    double ret = 0.0;
    while (sp >= 0) {
        PwFrame &f = stk[sp];
        if (f.len <= 128) {
            ret = np_pairwise_leaf(a + f.off, f.len);
            sp--;
            // hand the value to the parents
            while (sp >= 0) {
                PwFrame &p = stk[sp];
                int n2 = p.len / 2; n2 -= n2 % 8;
                if (p.stage == 1) {            // left child done: start the right child
                    p.lval = ret; p.stage = 2;
                    stk[sp + 1].off = p.off + n2; stk[sp + 1].len = p.len - n2; stk[sp + 1].stage = 0;
                    sp++;
                    break;
                }
                ret = p.lval + ret;            // both children done
                sp--;
            }
        } else {                               // stage 0: descend into the left child
            int n2 = f.len / 2; n2 -= n2 % 8;
            f.stage = 1;
            stk[sp + 1].off = f.off; stk[sp + 1].len = n2; stk[sp + 1].stage = 0;
            sp++;
        }
    }
    return ret;
}

constexpr int OUT_CHUNK = 64;   // output slots reserved per global atomic

struct PitParams {
    const double *elev;
    const uint8_t *pitmask;     // flats & (elev > 0), frozen before any pit is patched (:1284)
    const double *dX, *dY;      // fence spacing, n-1 entries
    double *mag;                // patched: mag[pit] = mean(s)  (:1370)
    uint8_t *flats;             // patched: flats[pit] = False   (:1371)
    int n, m;
    int max_iter, max_dist, min_border;
    int elev_f32;               // float32 DEM: |e[pit]-e[drain]| is a float32 subtraction in the reference (:1361)
    double max_dist_XY;         // NaN = None
    // raw output triplets (pit, drain, weight), appended per pit in ascending drain order
    int32_t *out_src, *out_dst; double *out_w;
    int32_t *out_count;         // [0] edges, [1] pits without drain, [2] overflow pits, [3] capacity errors
    int32_t out_cap;
    int32_t *overflow_list;     // pits to re-run with the next larger window
    int32_t *overflow_count;
    int32_t *lane_overflow;     // pits the lane version hands to the wavefront version (count: out_count[4])
    uint32_t *lane_state;       // 12 words per handed-over pit (same index as lane_overflow): iterations done, epit_border, the REGION as a
                                // 16 x 16 bitmap -- the wavefront version starts from the lane version's region instead of from the pit
    int32_t *work_next;         // next unclaimed entry of the pit list (lane version)
    int32_t *row_overflow;      // entries of the hand-over list the row version (pits_row.inl) passes on to the wavefront version
    int32_t *row_overflow_count;
    int rw_target;              // row version: entries the head is refilled with
    int2 *row_rec;              // row version: per entry of the hand-over list (first output slot, candidate drains) for k_pits_row_finish
    const int32_t *wave_index;  // wavefront version: entry q of its input is entry wave_index[q] of the hand-over list (nullptr: q itself)
    int32_t *dbg;               // PYDEM_PITS_DEBUG=2: per-pit {rounds, last border size, hand-over reason, drains}
    unsigned long long *prof;   // PYDEM_PITS_DEBUG=3: cycles per phase of the lane pass
};

__device__ __forceinline__ double pit_drop(const PitParams &P, double epit, double edrain)
{
    if (P.elev_f32) return (double)fabsf((float)epit - (float)edrain);
    return fabs(epit - edrain);
}

// group = the threads that own one pit: a wavefront (NT = 64) or a whole workgroup (NT = 256)
template <int NT>
__device__ __forceinline__ void group_sync()
{
    if (NT == 64) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}

template <int NT>
__device__ __forceinline__ double group_min(double v, double *red, int gl)
{
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off));
    if (NT == 64) return v;
    group_sync<NT>();
    if ((gl & 63) == 0) red[gl >> 6] = v;
    group_sync<NT>();
    double r = red[0];
    for (int k = 1; k < NT / 64; k++) r = fmin(r, red[k]);
    group_sync<NT>();
    return r;
}

template <int NT>
__device__ __forceinline__ int group_sum(int v, int *red, int gl, int *excl)
{
    // inclusive wave scan, then (for workgroups) offsets across waves; returns total, *excl = exclusive prefix
    const int lane = gl & 63;
    int incl = v;
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off); if (lane >= off) incl += o; }
    if (NT == 64) { *excl = incl - v; return __shfl(incl, 63); }
    group_sync<NT>();
    if (lane == 63) red[gl >> 6] = incl;
    group_sync<NT>();
    int base = 0, tot = 0;
    for (int k = 0; k < NT / 64; k++) { if (k < (gl >> 6)) base += red[k]; tot += red[k]; }
    group_sync<NT>();
    *excl = base + incl - v;
    return tot;
}

// filters and weights of one pit: serial on one thread (drain counts are tiny), numpy operation order
__device__ void finish_pit(const PitParams &P, int32_t pit, int ipit, int jpit, double epit, int ndrain, int32_t *dlist,
                           double *dxy, double *sv, int32_t &chunk_base, int32_t &chunk_left, PwFrame *stk)
{
    const int n = P.n, m = P.m;
    int nd = ndrain;
    if (P.max_dist) {                                                        // :1335-1343
        int keep = 0;
        for (int t = 0; t < nd; t++) {
            const int di = ipit - dlist[t] / m, dj = jpit - dlist[t] % m;
            const double dij = sqrt((double)(di * di + dj * dj));
            if (dij <= (double)P.max_dist) dlist[keep++] = dlist[t];
        }
        nd = keep;
    }
    if (nd > 0) {
        const int ndX = n - 1;
        for (int t = 0; t < nd; t++) {                                       // :1346-1349
            const int idr = dlist[t] / m, jdr = dlist[t] % m;
            double dxm;
            if (ipit == idr) dxm = P.dX[ipit < ndX - 1 ? ipit : ndX - 1];    // _get_dX_mean :1994-1995
            else {
                const int a = ipit < idr ? ipit : idr, b = ipit < idr ? idr : ipit;
                dxm = np_pairwise_sum(P.dX + a, b - a, stk) / (double)(b - a);   // .mean() :1997
            }
            const double dx = dxm * (double)(jpit - jdr);
            const int a = ipit < idr ? ipit : idr, b = ipit < idr ? idr : ipit;
            const double dy = np_pairwise_sum(P.dY + a, b - a, stk);
            dxy[t] = sqrt(dx * dx + dy * dy);
        }
        if (!isnan(P.max_dist_XY) && P.max_dist_XY != 0) {                   // :1352-1358
            int keep = 0;
            for (int t = 0; t < nd; t++)
                if (dxy[t] <= P.max_dist_XY) { dlist[keep] = dlist[t]; dxy[keep] = dxy[t]; keep++; }
            nd = keep;
        }
    }
    if (nd == 0) { atomicAdd(&P.out_count[1], 1); }
    else {
        for (int t = 0; t < nd; t++) sv[t] = pit_drop(P, epit, P.elev[dlist[t]]) / dxy[t];   // :1361
        const double ssum = np_pairwise_sum(sv, nd, stk);
        // output slots come in chunks (one global atomic per ~30 pits instead of one per pit: 3.5 M
        // atomics on a single address cost ~40 ms); unused slots keep src = -1 and are dropped later
        if (nd > chunk_left) {
            const int32_t grab = nd > OUT_CHUNK ? nd : OUT_CHUNK;
            chunk_base = atomicAdd(&P.out_count[0], grab);
            chunk_left = grab;
        }
        if (chunk_base + nd <= P.out_cap) {
            for (int t = 0; t < nd; t++) {                                   // :1365-1367
                P.out_src[chunk_base + t] = pit; P.out_dst[chunk_base + t] = dlist[t]; P.out_w[chunk_base + t] = sv[t] / ssum;
            }
        } else atomicAdd(&P.out_count[3], 1);
        chunk_base += nd; chunk_left -= nd;
        P.mag[pit] = ssum / (double)nd;                                      // np.mean(s) :1370
        P.flats[pit] = 0;                                                    // :1371
    }
}

// Wavefront version of finish_pit: lane t owns drain t (at most 64 drains, slices of at most 63 rows,
// so every numpy sum is a single pairwise leaf).  Filters compact with ballots (order preserved),
// the distance / slope arithmetic of the drains runs in parallel, and only the numpy-ordered sum of
// the slopes is serial.  chunk_base / chunk_left stay wave-uniform.
__device__ __forceinline__ void finish_pit_wave(const PitParams &P, int32_t pit, int ipit, int jpit, double epit, int ndrain,
                                                int32_t *dlist, double *dxy, double *sv, int32_t &chunk_base,
                                                int32_t &chunk_left, int lane)
{
    const int n = P.n, m = P.m;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int nd = ndrain;
    bool live = lane < nd;
    int32_t cell = live ? dlist[lane] : 0;
    if (P.max_dist) {                                                        // :1335-1343
        if (live) {
            const int di = ipit - cell / m, dj = jpit - cell % m;
            live = sqrt((double)(di * di + dj * dj)) <= (double)P.max_dist;
        }
        const unsigned long long bal = __ballot(live);
        __builtin_amdgcn_wave_barrier();
        if (live) dlist[__popcll(bal & lt)] = cell;
        nd = __popcll(bal);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        live = lane < nd;
        cell = live ? dlist[lane] : 0;
    }
    double d = 0.0;
    if (live) {                                                              // :1346-1349
        const int ndX = n - 1;
        const int idr = cell / m, jdr = cell % m;
        const int a = ipit < idr ? ipit : idr, b = ipit < idr ? idr : ipit;
        double dxm;
        if (ipit == idr) dxm = P.dX[ipit < ndX - 1 ? ipit : ndX - 1];        // _get_dX_mean :1994-1995
        else dxm = np_pairwise_leaf(P.dX + a, b - a) / (double)(b - a);      // .mean() :1997
        const double dx = dxm * (double)(jpit - jdr);
        const double dy = np_pairwise_leaf(P.dY + a, b - a);
        d = sqrt(dx * dx + dy * dy);
    }
    if (!isnan(P.max_dist_XY) && P.max_dist_XY != 0) {                       // :1352-1358
        const bool keep = live && d <= P.max_dist_XY;
        const unsigned long long bal = __ballot(keep);
        __builtin_amdgcn_wave_barrier();
        if (keep) { const int k = __popcll(bal & lt); dlist[k] = cell; dxy[k] = d; }
        nd = __popcll(bal);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        live = lane < nd;
        cell = live ? dlist[lane] : 0;
        d = live ? dxy[lane] : 0.0;
    }
    if (nd == 0) { if (lane == 0) atomicAdd(&P.out_count[1], 1); return; }
    double s = 0.0;
    if (live) { s = pit_drop(P, epit, P.elev[cell]) / d; sv[lane] = s; }           // :1361
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double ssum = 0.0;
    if (lane == 0) ssum = np_pairwise_leaf(sv, nd);
    ssum = __shfl(ssum, 0);
    if (nd > chunk_left) {                                                   // see finish_pit
        const int32_t grab = nd > OUT_CHUNK ? nd : OUT_CHUNK;
        int32_t g = 0;
        if (lane == 0) g = atomicAdd(&P.out_count[0], grab);
        chunk_base = __shfl(g, 0);
        chunk_left = grab;
    }
    if (chunk_base + nd <= P.out_cap) {
        if (live) {                                                          // :1365-1367
            P.out_src[chunk_base + lane] = pit; P.out_dst[chunk_base + lane] = cell; P.out_w[chunk_base + lane] = s / ssum;
        }
    } else if (lane == 0) atomicAdd(&P.out_count[3], 1);
    chunk_base += nd; chunk_left -= nd;
    if (lane == 0) {
        P.mag[pit] = ssum / (double)nd;                                      // np.mean(s) :1370
        P.flats[pit] = 0;                                                    // :1371
    }
}

// One pit.  W = window edge (multiple of 32), MAXD = drain list capacity.
// region / border / promote: W*W-bit bitmaps; dlist/dxy/sv: drain scratch.
template <int NT, int W, int MAXD>
__device__ void solve_pit(const PitParams &P, int32_t pit, int gl, uint32_t *region, uint32_t *border, uint32_t *promote,
                          int32_t *dlist, double *dxy, double *sv, double *redd, int *redi, int *flag,
                          int32_t &chunk_base, int32_t &chunk_left, PwFrame *stk)
{
    constexpr int WORDS = W * W / 32;
    constexpr int WPR = W / 32;                    // words per window row
    const int n = P.n, m = P.m;
    const int ipit = pit / m, jpit = pit - ipit * m;
    // window origin: centred on the pit, clipped to the tile
    int r0 = ipit - W / 2, c0 = jpit - W / 2;
    if (r0 > n - W) r0 = n - W;
    if (c0 > m - W) c0 = m - W;
    if (r0 < 0) r0 = 0;
    if (c0 < 0) c0 = 0;
    for (int w = gl; w < WORDS; w += NT) { region[w] = 0; border[w] = 0; promote[w] = 0; }
    if (gl == 0) { flag[0] = 0; flag[1] = 0; flag[2] = W; flag[3] = -1; }     // [2],[3]: first/last window row in use
    group_sync<NT>();
    const double epit = P.elev[pit];
    // pit_area = [pit]; border = its 8 neighbours inside the tile (:1289-1292)
    if (gl == 0) {
        const int wr = ipit - r0, wc = jpit - c0;
        region[wr * WPR + (wc >> 5)] |= 1u << (wc & 31);
        for (int di = -1; di <= 1; di++)
            for (int dj = -1; dj <= 1; dj++) {
                if (!di && !dj) continue;
                const int ii = ipit + di, jj = jpit + dj;
                if (ii < 0 || ii >= n || jj < 0 || jj >= m) continue;
                const int r = ii - r0, c = jj - c0;
                if (r < 0 || r >= W || c < 0 || c >= W) { flag[0] = 1; continue; }
                border[r * WPR + (c >> 5)] |= 1u << (c & 31);
                if (r < flag[2]) flag[2] = r;
                if (r > flag[3]) flag[3] = r;
            }
    }
    group_sync<NT>();
    double epit_border = epit;
    if (P.min_border) {                                                          // :1294-1295
        double mn = INFINITY;
        for (int w = gl; w < WORDS; w += NT) {
            uint32_t b = border[w];
            while (b) {
                const int k = __ffs((int)b) - 1; b &= b - 1;
                const int r = w / WPR, c = (w % WPR) * 32 + k;
                mn = fmin(mn, P.elev[(int64_t)(r0 + r) * m + (c0 + c)]);
            }
        }
        epit_border = group_min<NT>(mn, redd, gl);
    }
    int ndrain = -1;        // -1: none found
    for (int it = 0; it < P.max_iter; it++) {                                    // :1300
        if (flag[0]) break;                                                      // left the window
        const int w_first = flag[2] * WPR, w_last = (flag[3] + 1) * WPR;         // rows that hold border / region bits
        // --- scan the border: minima of all / non-pit / pit cells
        double mn = INFINITY, mn_np = INFINITY, mn_p = INFINITY, nan_mark = 1.0;
        for (int w = w_first + gl; w < w_last; w += NT) {
            uint32_t b = border[w];
            while (b) {
                const int k = __ffs((int)b) - 1; b &= b - 1;
                const int r = w / WPR, c = (w % WPR) * 32 + k;
                const int64_t cell = (int64_t)(r0 + r) * m + (c0 + c);
                const double e = P.elev[cell];
                if (e != e) nan_mark = -1.0;
                mn = fmin(mn, e);
                if (P.pitmask[cell]) mn_p = fmin(mn_p, e); else mn_np = fmin(mn_np, e);
            }
        }
        mn = group_min<NT>(mn, redd, gl);
        mn_np = group_min<NT>(mn_np, redd, gl);
        mn_p = group_min<NT>(mn_p, redd, gl);
        // numpy's min propagates NaN: with a nodata cell on the border there is no non-pit drain and no growth
        const bool has_nan = group_min<NT>(nan_mark, redd, gl) < 0;
        if (mn == INFINITY && !has_nan) break;                                   // empty border (:1304-1305)
        int mode = 0;                                                            // 1: non-pit drains, 2: pit drains
        if (!has_nan && mn_np < epit_border) mode = 1;                           // :1312-1316
        else if (mn_p < epit) mode = 2;                                          // :1317-1320
        if (!mode && has_nan) break;
        if (mode) {
            // collect drains in ascending cell order (window scan order == ascending id)
            int cnt = 0;
            // each thread owns a contiguous run of words so the concatenation over threads is ordered
            const int per = (w_last - w_first + NT - 1) / NT;
            const int w_lo = w_first + gl * per, w_hi = (w_lo + per < w_last) ? w_lo + per : w_last;
            for (int w = w_lo; w < w_hi; w++) {
                uint32_t b = border[w];
                while (b) {
                    const int k = __ffs((int)b) - 1; b &= b - 1;
                    const int r = w / WPR, c = (w % WPR) * 32 + k;
                    const int64_t cell = (int64_t)(r0 + r) * m + (c0 + c);
                    const double e = P.elev[cell];
                    const bool isp = P.pitmask[cell];
                    if (mode == 1 ? (!isp && e < epit_border) : (isp && e < epit)) cnt++;
                }
            }
            int excl;
            const int tot = group_sum<NT>(cnt, redi, gl, &excl);
            if (tot > MAXD) { if (gl == 0) flag[0] = 1; group_sync<NT>(); break; }
            int pos = excl;
            for (int w = w_lo; w < w_hi; w++) {
                uint32_t b = border[w];
                while (b) {
                    const int k = __ffs((int)b) - 1; b &= b - 1;
                    const int r = w / WPR, c = (w % WPR) * 32 + k;
                    const int64_t cell = (int64_t)(r0 + r) * m + (c0 + c);
                    const double e = P.elev[cell];
                    const bool isp = P.pitmask[cell];
                    if (mode == 1 ? (!isp && e < epit_border) : (isp && e < epit)) dlist[pos++] = (int32_t)cell;
                }
            }
            ndrain = tot;
            group_sync<NT>();
            break;
        }
        // --- grow: pit_area += border[eborder == emin] (:1322-1323)
        for (int w = w_first + gl; w < w_last; w += NT) {
            uint32_t b = border[w], pr = 0;
            while (b) {
                const int k = __ffs((int)b) - 1; b &= b - 1;
                const int r = w / WPR, c = (w % WPR) * 32 + k;
                if (P.elev[(int64_t)(r0 + r) * m + (c0 + c)] == mn) pr |= 1u << k;
            }
            promote[w] = pr;
            if (pr) region[w] |= pr;        // word w is owned by this thread in this phase
        }
        group_sync<NT>();
        for (int w = w_first + gl; w < w_last; w += NT) {
            uint32_t pr = promote[w];
            while (pr) {
                const int k = __ffs((int)pr) - 1; pr &= pr - 1;
                const int r = w / WPR, c = (w % WPR) * 32 + k;
                for (int di = -1; di <= 1; di++)
                    for (int dj = -1; dj <= 1; dj++) {
                        if (!di && !dj) continue;
                        const int ii = r0 + r + di, jj = c0 + c + dj;
                        if (ii < 0 || ii >= n || jj < 0 || jj >= m) continue;
                        const int rr = r + di, cc = c + dj;
                        if (rr < 0 || rr >= W || cc < 0 || cc >= W) { flag[0] = 1; continue; }
                        const int ww = rr * WPR + (cc >> 5);
                        const uint32_t bit = 1u << (cc & 31);
                        if (!((region[ww] | border[ww]) & bit)) {
                            atomicOr(&border[ww], bit);
                            atomicMin(&flag[2], rr); atomicMax(&flag[3], rr);
                        }
                    }
            }
        }
        group_sync<NT>();
        for (int w = w_first + gl; w < w_last; w += NT) {
            const uint32_t pr = promote[w];
            if (pr) border[w] &= ~pr;
        }
        group_sync<NT>();
    }
    group_sync<NT>();
    if (flag[0]) {                                                               // hand over to the large-window pass
        if (gl == 0) {
            if (P.overflow_list) P.overflow_list[atomicAdd(P.overflow_count, 1)] = pit;
            else atomicAdd(&P.out_count[3], 1);
        }
        return;
    }
    if (ndrain < 0) { if (gl == 0) atomicAdd(&P.out_count[1], 1); return; }      // :1327-1329
    if (gl == 0) finish_pit(P, pit, ipit, jpit, epit, ndrain, dlist, dxy, sv, chunk_base, chunk_left, stk);

}

// ---------------------------------------------------------------------------------------------
// Wavefront version (second pass: the ~10 % of the pits the lane version hands over; they average
// 66 rounds with borders of 50-250 cells).  The kernel is bound by instruction issue (a wave64
// instruction occupies its SIMD for 4 cycles and a round used to be ~600 of them), so a round is
// kept short: the border is an incremental, unordered LIST in LDS (elevation, window position, pit
// flag) that is read ONCE per round into registers -- minimum (DPP reduction), then a ballot per
// 64 slots takes the cells equal to it out of the list; their slots are recycled for the cells that
// enter next, so the list stays as short as the border; then one 64-lane step per 8 promoted cells
// tests their 8 neighbours against the region|border bitmap (LDS atomicOr: exactly one lane wins a
// new cell), loads the new elevations in parallel and files them by ballot rank.  As in the lane
// version the drain tests are evaluated when a cell enters the border.  Window 128x128 cells; pits
// that leave it (or exceed the list / drain capacity) go to the workgroup version.
// ---------------------------------------------------------------------------------------------
#ifndef PYDEM_WV_OCC
#define PYDEM_WV_OCC 6
#endif
#ifndef PYDEM_WV_CAP
#define PYDEM_WV_CAP 256
#endif
constexpr int WV_MAXD = 64;       // drain list capacity
constexpr int LN_W16 = 16;        // window edge of the lane version (LN_W below; its hand-over records are read here)

// Two instances: <128, 256, uint16_t> for the bulk (positions fit 14 bits, bit 14 = pit flag; 6.4 KB of LDS and 80 VGPRs:
// six wavefronts per SIMD instead of five with a 384-cell list -- 18 of 397 678 pits of the 16384^2 bench tile then need the
// large instance) and
// <256, 2048, uint32_t> for the pits of plateau terrain whose border outgrows 384 cells (integer DEMs: a
// quarter of the candidates at 4096^2) -- one wavefront per workgroup there, 44 KB of LDS each
template <int WW, int CAP, typename PosT>
struct WaveLds {
    static constexpr PosT HOLE = (PosT)~(PosT)0;
    static constexpr PosT PITBIT = (PosT)((PosT)1 << (sizeof(PosT) * 8 - 2));
    uint32_t seen[WW * WW / 32];
    double le[CAP];
    PosT lpos[CAP];               // window position | PITBIT; HOLE = free slot
    uint16_t holes[CAP];          // free slots below the list end
    union {
        PosT pq[CAP];             // cells promoted in the current round
        struct { int32_t dl[WV_MAXD]; double dxy[WV_MAXD], sv[WV_MAXD]; } fin;   // drain scratch (after the rounds)
    } u;
};

// fmin without the canonicalising v_max_f64 the compiler puts in front of every llvm.minnum operand (one extra fp64
// instruction per minimum in kernels that are bound by instruction issue): v_min_f64 already returns the other operand
// when one is a (quiet) NaN, which is what the border scans rely on
__device__ __forceinline__ double min_f64(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int CTRL>
__device__ __forceinline__ double dpp_fmin(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    // (every lane has a source lane under these permutations: with bound_ctrl the move needs no tied old value, i.e. no copy of
    // the operand in front of it -- two v_mov per call otherwise)
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return min_f64(v, __hiloint2double(hi2, lo2));
}
// minimum over the wavefront: butterflies inside each row of 16 lanes on the DPP crossbar, then the four
// row values through scalar registers
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "wave_min uses the gfx9 DPP row broadcasts (row_bcast:15 / row_bcast:31, removed in gfx10+): this library is built for gfx950 only"
#endif
__device__ __forceinline__ double wave_min(double v)
{
    v = dpp_fmin<0xB1>(v);        // quad_perm [1,0,3,2]
    v = dpp_fmin<0x4E>(v);        // quad_perm [2,3,0,1]
    v = dpp_fmin<0x141>(v);       // row_half_mirror
    v = dpp_fmin<0x140>(v);       // row_mirror: every lane of a row holds the row's minimum
    // across the rows with the gfx9 row broadcasts: lane 15 of row k-1 into row k, then lane 31 into rows 2 and 3 -- lane 63
    // ends up with the minimum of all four (the zeros bound_ctrl puts where there is no source land in lanes that are not
    // read); two readlanes instead of eight and no minimum on values that came back through scalar registers
    v = dpp_fmin<0x142>(v);       // row_bcast:15
    v = dpp_fmin<0x143>(v);       // row_bcast:31
    // (after the two broadcasts ONLY lane 63 holds the minimum -- rows 0 and 1 took min(v, 0) from the lanes without a source;
    // the function hands out lane 63's value as a wavefront-uniform scalar, so no caller can pick up another lane's)
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_readlane(hi, 63), __builtin_amdgcn_readlane(lo, 63));
}
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int W2, int W2_CAP, typename PosT>
__device__ void solve_pit_wave(const PitParams &P, int32_t pit, int lane, WaveLds<W2, W2_CAP, PosT> &L, int32_t &chunk_base,
                               int32_t &chunk_left, const uint32_t *state = nullptr)
{
    constexpr int W2_SLOTS = W2_CAP / 64;
    constexpr PosT W2_HOLE = WaveLds<W2, W2_CAP, PosT>::HOLE, W2_PITBIT = WaveLds<W2, W2_CAP, PosT>::PITBIT;
    const int n = P.n, m = P.m;
    const int ipit = pit / m, jpit = pit - ipit * m;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int r0 = ipit - W2 / 2, c0 = jpit - W2 / 2;
    if (r0 > n - W2) r0 = n - W2;
    if (c0 > m - W2) c0 = m - W2;
    if (r0 < 0) r0 = 0;
    if (c0 < 0) c0 = 0;
    for (int w = lane; w < W2 * W2 / 32; w += 64) L.seen[w] = 0;
    const double epit = P.elev[pit];
    double epit_border = epit;
    int nb = 0, nh = 0, n_alive = 0;        // list end / free slots below it / live entries (wave-uniform)
    bool has_np = false, has_p = false, has_nan = false;
    int over = 0;                           // 1: left the window, 2: list capacity, 3: drain capacity
    // the unseen neighbours of the nq cells in L.u.pq join the border
    auto expand = [&](int nq) {
        for (int base = 0; base < nq * 8; base += 64) {
            const int idx = base + lane;
            bool isnew = false, out = false;
            double e = 0.0; uint32_t pm = 0; int npos = 0;
            if (idx < nq * 8) {
                const int pos = L.u.pq[idx >> 3], d = idx & 7;
                const int di = d < 3 ? -1 : (d < 5 ? 0 : 1);
                const int dj = d < 3 ? d - 1 : (d == 3 ? -1 : (d == 4 ? 1 : d - 6));
                const int r = pos / W2, c = pos % W2;
                const int ii = r0 + r + di, jj = c0 + c + dj;
                if (ii >= 0 && ii < n && jj >= 0 && jj < m) {
                    const int rr = r + di, cc = c + dj;
                    if (rr < 0 || rr >= W2 || cc < 0 || cc >= W2) out = true;
                    else {
                        npos = rr * W2 + cc;
                        const uint32_t bit = 1u << (npos & 31);
                        const uint32_t old = atomicOr(&L.seen[npos >> 5], bit);
                        if (!(old & bit)) {
                            isnew = true;
                            const int64_t cell = (int64_t)ii * m + jj;
                            e = P.elev[cell]; pm = P.pitmask[cell];
                        }
                    }
                }
            }
            if (__ballot(out)) over = 1;
            const unsigned long long bal = __ballot(isnew);
            const int cnt = __popcll(bal);
            const int fresh = cnt > nh ? cnt - nh : 0;                           // slots taken beyond the list end
            if (nb + fresh > W2_CAP) { over = 2; break; }
            if (isnew) {
                const int rk = __popcll(bal & lt);
                const int k = rk < nh ? (int)L.holes[nh - 1 - rk] : nb + (rk - nh);
                L.le[k] = e; L.lpos[k] = (PosT)((PosT)npos | (pm ? W2_PITBIT : (PosT)0));
            }
            if (__ballot(isnew && pm && e < epit)) has_p = true;
            if (__ballot(isnew && !pm && e < epit_border)) has_np = true;
            if (__ballot(isnew && e != e)) has_nan = true;                       // nodata on the border (see below)
            nh -= cnt - fresh; nb += fresh; n_alive += cnt;
        }
        wave_sync();
    };
    int it0 = 0, nq0 = 1;
    if (state) {
        // the lane version grew this pit for `it0` iterations before its 16 x 16 window or its 32-cell list overflowed: the
        // region it reached is marked, and ONE expansion of all its cells rebuilds the border (the drain tests of the border
        // cells are evaluated as they enter, like always) -- instead of replaying those iterations one round each
        it0 = (int)state[0];
        epit_border = __hiloint2double((int)state[3], (int)state[2]);
        int lr0 = ipit - LN_W16 / 2, lc0 = jpit - LN_W16 / 2;                   // the lane version's window (same clipping)
        if (lr0 > n - LN_W16) lr0 = n - LN_W16;
        if (lc0 > m - LN_W16) lc0 = m - LN_W16;
        if (lr0 < 0) lr0 = 0;
        if (lc0 < 0) lc0 = 0;
        const int dpos = (lr0 - r0) * W2 + (lc0 - c0);
        nq0 = 0;
#pragma unroll 1
        for (int j = 0; j < 4; j++) {
            const int cidx = lane + 64 * j;                                      // cell of the 16 x 16 window
            const bool in = (state[4 + 2 * j + (lane >> 5)] >> (lane & 31)) & 1u;
            const unsigned long long bal = __ballot(in);
            if (in) {
                const int pos = dpos + (cidx >> 4) * W2 + (cidx & 15);
                atomicOr(&L.seen[pos >> 5], 1u << (pos & 31));
                L.u.pq[nq0 + __popcll(bal & lt)] = (PosT)pos;
            }
            nq0 += __popcll(bal);
        }
    } else if (lane == 0) {                                                      // pit_area = [pit] (:1289-1292)
        const int pos = (ipit - r0) * W2 + (jpit - c0);
        L.seen[pos >> 5] = 1u << (pos & 31);
        L.u.pq[0] = (PosT)pos;
    }
    wave_sync();
    expand(nq0);
    if (!state && P.min_border) {                                                // :1294-1295
        double mn = INFINITY;
        for (int k = lane; k < nb; k += 64) mn = min_f64(mn, L.le[k]);
        mn = wave_min(mn);
        if (nb) epit_border = mn;
        has_np = false;                                                          // nothing is below the minimum
    }
    int mode = 0, ndrain = -1, it_used = it0;
    for (int it = it0; it < P.max_iter; it++) {                                  // :1300
        if (over) break;
        if (n_alive == 0) break;                                                 // :1304-1305
        // numpy's min propagates NaN: with a nodata cell on the border there is no non-pit drain and no growth
        if (has_nan) { if (has_p) mode = 2; break; }
        if (has_np) { mode = 1; break; }                                         // :1312-1316
        if (has_p) { mode = 2; break; }                                          // :1317-1320
        it_used = it + 1;
        // one read of the list: the minimum ...
        double e[W2_SLOTS];
        double mn = INFINITY;
#pragma unroll
        for (int j = 0; j < W2_SLOTS; j++) {
            const int k = lane + 64 * j;
            e[j] = INFINITY;
            if (j * 64 < nb) {                                                   // (wave-uniform: half of the rounds have a border of <= 64 cells)
                if (k < nb) e[j] = L.le[k];                                      // free slots hold +inf
                mn = min_f64(mn, e[j]);
            }
        }
        mn = wave_min(mn);
        // ... and pit_area += border[eborder == emin] (:1322-1323): out of the list, slots recycled
        int nq = 0;
#pragma unroll
        for (int j = 0; j < W2_SLOTS; j++) {
            if (j * 64 >= nb) break;
            const int k = lane + 64 * j;
            bool match = k < nb && e[j] == mn;
            PosT ps = 0;
            if (match) { ps = L.lpos[k]; match = ps != W2_HOLE; }
            const unsigned long long bal = __ballot(match);
            if (!bal) continue;
            if (match) {
                const int r = nq + __popcll(bal & lt);
                L.u.pq[r] = (PosT)(ps & (PosT)(W2_PITBIT - 1));
                L.holes[nh + r] = (uint16_t)k;
                L.le[k] = INFINITY; L.lpos[k] = W2_HOLE;
            }
            nq += __popcll(bal);
        }
        nh += nq; n_alive -= nq;
        wave_sync();
        expand(nq);
    }
    if (!over && mode) {
        // drains: ballot-compacted, then rank-sorted into ascending cell order (the order of setdiff1d)
        int nd = 0;
        for (int base = 0; base < nb; base += 64) {
            const int k = base + lane;
            bool pred = false;
            int32_t cell = 0;
            if (k < nb && L.lpos[k] != W2_HOLE) {
                const int pos = (int)(L.lpos[k] & (PosT)(W2_PITBIT - 1));
                const bool pm = (L.lpos[k] & W2_PITBIT) != 0;
                const double ev = L.le[k];
                pred = mode == 1 ? (!pm && ev < epit_border) : (pm && ev < epit);
                cell = (int32_t)((int64_t)(r0 + pos / W2) * m + (c0 + pos % W2));
            }
            const unsigned long long bal = __ballot(pred);
            const int rank = nd + __popcll(bal & lt);
            wave_sync();                                                         // (u.pq is dead: the drain scratch may be written)
            if (pred && rank < WV_MAXD) L.u.fin.dl[rank] = cell;
            nd += __popcll(bal);
        }
        if (nd > WV_MAXD) over = 3;
        else {
            wave_sync();
            const int32_t key = lane < nd ? L.u.fin.dl[lane] : INT32_MAX;
            int rank = 0;
            for (int t = 0; t < nd; t++) rank += L.u.fin.dl[t] < key;
            wave_sync();
            if (lane < nd) L.u.fin.dl[rank] = key;
            wave_sync();
            ndrain = nd;
        }
    }
    if (lane == 0 && P.dbg) {                                                    // statistics (debug only)
        const int idx = atomicAdd(&P.out_count[6], 1);
        P.dbg[4 * idx] = it_used; P.dbg[4 * idx + 1] = n_alive; P.dbg[4 * idx + 2] = over; P.dbg[4 * idx + 3] = ndrain;
    }
    if (over) {                                                                  // hand over to the next larger pass
        if (lane == 0) P.overflow_list[atomicAdd(P.overflow_count, 1)] = pit;
        return;
    }
    if (ndrain < 0) { if (lane == 0) atomicAdd(&P.out_count[1], 1); return; }    // :1327-1329
    finish_pit_wave(P, pit, ipit, jpit, epit, ndrain, L.u.fin.dl, L.u.fin.dxy, L.u.fin.sv, chunk_base, chunk_left, lane);
}

#include "pits_row.inl"

// ---------------------------------------------------------------------------------------------
// Lane version (first pass over ALL pits).  The wavefront version spends ~700 VALU issues per round
// with 64 lanes serving a border of ~10-20 cells, so it is instruction-bound with most lanes idle.
// Here every LANE owns one pit: the border is an unordered list (elevation + window position) in
// LDS, region|border membership is a 16x16-cell bitmap in LDS, and the per-round work is a scan of
// the list for the minimum plus the 8 neighbours of each promoted cell.  The drain test of the
// reference ("any non-pit border cell below the pit / any pit cell below the pit", :1312-1320) uses
// fixed thresholds, so it is evaluated once per cell when the cell ENTERS the border.  Lanes
// reconverge after their pit; output slots are then allocated with ONE atomic per wavefront
// (exact prefix sum of the drain counts: no holes).  Pits that leave the window or the list
// capacity go to the wavefront version.
// ---------------------------------------------------------------------------------------------
constexpr int LN_W = 16;          // window edge
static_assert(LN_W == LN_W16, "hand-over records");
#ifndef PYDEM_LN_B
#define PYDEM_LN_B 32
#endif
constexpr int LN_B = PYDEM_LN_B;  // border list capacity (one bit per slot in a 32-bit register)
#ifndef PYDEM_LN_T
#define PYDEM_LN_T 64
#endif
constexpr int LN_T = PYDEM_LN_T;  // threads (= pits) per workgroup: one wavefront, 22.5 KB of LDS -> 7 workgroups per CU (128 threads: 6 wavefronts per CU)

__device__ __forceinline__ double np_pairwise_leaf_strided(const double *a, int stride, int n)
{
    if (n < 8) {
        double res = 0.;
        for (int i = 0; i < n; i++) res += a[i * stride];
        return res;
    }
    double r0 = a[0], r1 = a[stride], r2 = a[2 * stride], r3 = a[3 * stride], r4 = a[4 * stride], r5 = a[5 * stride],
           r6 = a[6 * stride], r7 = a[7 * stride];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        r0 += a[i * stride]; r1 += a[(i + 1) * stride]; r2 += a[(i + 2) * stride]; r3 += a[(i + 3) * stride];
        r4 += a[(i + 4) * stride]; r5 += a[(i + 5) * stride]; r6 += a[(i + 6) * stride]; r7 += a[(i + 7) * stride];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += a[i * stride];
    return res;
}

// np_pairwise_leaf for slices of at most 15 elements with all loads issued up front (indices clamped to the slice):
// the lane version sums dX / dY over the rows between a pit and its drain, a chain of dependent global loads otherwise
__device__ __forceinline__ double np_pairwise_leaf15(const double *__restrict__ a, int n)
{
    double v[15];
#pragma unroll
    for (int k = 0; k < 15; k++) v[k] = a[k < n ? k : (n > 0 ? n - 1 : 0)];
    if (n < 8) {
        double res = 0.;
#pragma unroll
        for (int k = 0; k < 7; k++) if (k < n) res += v[k];
        return res;
    }
    double res = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
#pragma unroll
    for (int k = 8; k < 15; k++) if (k < n) res += v[k];
    return res;
}

// Lanes are re-armed as soon as their pit is finished: a wavefront keeps a private range of the pit list
// (LN_CHUNK pits per global atomic) and every loop trip (a) hands new pits to idle lanes, (b) runs ONE round
// of every busy lane -- or its drain selection when the growth has ended -- and (c) writes the output of the
// lanes that just finished with one slot allocation per wavefront.  With one pit per lane per launch a
// wavefront would run as long as its slowest pit (~40 rounds) while the average pit needs 11.
constexpr int LN_CHUNK = 256;
constexpr int LN_FIN_BATCH = 32;   // lanes that wait before the drain selection runs (12: 6.6 ms, 20: 6.0, 32: 5.6, 44: 5.8 at 16384^2)

__global__ __launch_bounds__(LN_T) void k_pits_lane(PitParams P, const int32_t *__restrict__ pits, const int32_t *npits)
{
    __shared__ double s_e[LN_B * LN_T];          // [slot][lane]: border elevations, later the drain slopes
    __shared__ uint8_t s_p[LN_B * LN_T];         // [slot][lane]: window position r*16+c
    __shared__ uint8_t s_q[LN_B * LN_T];         // [slot][lane]: cells promoted in the current round
    __shared__ uint32_t s_seen[8 * LN_T];        // [word][lane]: region | border bitmap of the window
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int n = P.n, m = P.m;
    const int32_t np = *npits;
    double *const le = s_e + tid;
    uint8_t *const lp = s_p + tid, *const lq = s_q + tid;
    uint32_t *const seen = s_seen + tid;
    // per-lane state of the pit in progress
    bool running = false, over = false, has_np = false, has_p = false, has_nan = false;
    int pending = 0;            // growth has ended: 1 / 2 drain mode, 3 no drain, 4 hand over; the lane waits for the next batch
    int32_t pit = 0;
    int r0 = 0, c0 = 0, ipit = 0, jpit = 0, nb = 0, it = 0;
    uint32_t pitbits = 0;
    double epit = 0.0, epit_border = 0.0;
    // the wavefront's range of the pit list
    int32_t chunk_next = 0, chunk_end = 0;
    bool more = true;
    // The unseen neighbours of the cells promoted in a round join the border in two steps: first every promoted cell
    // files the window positions of its new neighbours (LDS only), then the elevations / pit flags of ALL new cells are
    // fetched in one batch -- one memory round trip per round instead of one per promoted cell (the 64 pits of a
    // wavefront advance in lock-step, so a round used to last as long as the lane with the most promoted cells).
    auto discover = [&](int r, int c) {
        // the 3x3 neighbourhood of (r, c) lives in two words of the window bitmap (a word holds two 16-cell rows; rows
        // r - 1 and r + 1 are in consecutive words, row r in one of them): two independent reads, the eight bit tests in
        // registers, two writes -- instead of eight dependent read-modify-write round trips to LDS
        const int wa = (r > 0 ? r - 1 : 0) >> 1, wb = (r < LN_W - 1 ? r + 1 : LN_W - 1) >> 1;
        uint32_t va = seen[wa * LN_T], vb = seen[wb * LN_T];
        if (wb == wa) vb = va;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int di = k < 3 ? -1 : (k < 5 ? 0 : 1);
            const int dj = k < 3 ? k - 1 : (k == 3 ? -1 : (k == 4 ? 1 : k - 6));
            const int ii = r0 + r + di, jj = c0 + c + dj;
            const int rr = r + di, cc = c + dj;
            if (ii < 0 || ii >= n || jj < 0 || jj >= m) continue;
            if (rr < 0 || rr >= LN_W || cc < 0 || cc >= LN_W) { over = true; continue; }
            const int pos = rr * LN_W + cc;
            const uint32_t bit = 1u << (pos & 31);
            const bool in_a = (pos >> 5) == wa;
            const uint32_t wd = in_a ? va : vb;
            if (wd & bit) continue;
            if (nb == LN_B) { over = true; continue; }        // (not marked: the bitmap stays "region + listed border" for the hand-over)
            if (in_a) va |= bit; else vb |= bit;
            if (wb == wa) vb = va;
            lp[nb * LN_T] = (uint8_t)pos;
            nb++;
        }
        seen[wa * LN_T] = va;
        if (wb != wa) seen[wb * LN_T] = vb;
    };
    auto fetch_new = [&](int nb0) {
        constexpr int FB = 8;
        for (int base = nb0; base < nb; base += FB) {
            double e8[FB]; uint8_t pm8[FB];
#pragma unroll
            for (int k = 0; k < FB; k++) {
                e8[k] = 0.0; pm8[k] = 0;
                if (base + k < nb) {
                    const int pos = lp[(base + k) * LN_T];
                    const int64_t cell = (int64_t)(r0 + (pos >> 4)) * m + (c0 + (pos & 15));
                    e8[k] = P.elev[cell]; pm8[k] = P.pitmask[cell];
                }
            }
#pragma unroll
            for (int k = 0; k < FB; k++) {
                const int idx = base + k;
                if (idx >= nb) continue;
                le[idx * LN_T] = e8[k];
                if (e8[k] != e8[k]) has_nan = true;          // nodata on the border: see the drain rules below
                if (pm8[k]) { pitbits |= 1u << idx; if (e8[k] < epit) has_p = true; }
                else { pitbits &= ~(1u << idx); if (e8[k] < epit_border) has_np = true; }
            }
        }
    };
    const bool prof = P.prof != nullptr;
    long long acc_a = 0, acc_b = 0, acc_c = 0, acc_g = 0, trips = 0, busy = 0;
    for (;;) {
        long long tk0 = 0, tk1 = 0, tk2 = 0, tkg = 0;
        if (prof) tk0 = clock64();
        // ---- (a) new pits for idle lanes
        const unsigned long long idle = __ballot(!running);
        if (idle && more) {
            if (chunk_next == chunk_end) {
                int32_t b = 0;
                if (lane == 0) b = atomicAdd(P.work_next, LN_CHUNK);
                b = __shfl(b, 0);
                chunk_next = b < np ? b : np;
                chunk_end = b + LN_CHUNK < np ? b + LN_CHUNK : np;
                if (chunk_next >= chunk_end) more = false;
            }
            const int avail = chunk_end - chunk_next, want = __popcll(idle);
            const int rank = __popcll(idle & lt);
            if (!running && rank < avail) {
                pit = pits[chunk_next + rank];
                ipit = pit / m; jpit = pit - ipit * m;
                r0 = ipit - LN_W / 2; c0 = jpit - LN_W / 2;
                if (r0 > n - LN_W) r0 = n - LN_W;
                if (c0 > m - LN_W) c0 = m - LN_W;
                if (r0 < 0) r0 = 0;
                if (c0 < 0) c0 = 0;
#pragma unroll
                for (int w = 0; w < 8; w++) seen[w * LN_T] = 0;
                epit = P.elev[pit];
                epit_border = epit;
                nb = 0; pitbits = 0; it = 0;
                has_np = false; has_p = false; has_nan = false; over = false;
                {                                                                // pit_area = [pit] (:1289-1292)
                    const int pos = (ipit - r0) * LN_W + (jpit - c0);
                    seen[(pos >> 5) * LN_T] = 1u << (pos & 31);
                    discover(ipit - r0, jpit - c0);
                    fetch_new(0);
                }
                if (P.min_border) {                                              // :1294-1295
                    double mn = INFINITY;
                    for (int k = 0; k < nb; k++) mn = min_f64(mn, le[k * LN_T]);
                    if (nb) epit_border = mn;
                    has_np = false;                                              // nothing is below the minimum
                }
                running = true;
            }
            chunk_next += want < avail ? want : avail;
        }
        if (!__ballot(running)) { if (more) continue; break; }
        if (prof) { tk1 = clock64(); busy += __popcll(__ballot(running)); trips++; }
        // ---- (b) one step of every busy lane
        int nd = 0;                 // drains of a pit that finished in this trip, sorted, in slots [0, nd)
        int status = 0;             // 1: drained, 2: no drain, 3: hand over to the wavefront version
        double ssum = 0.0;
        if (running && !pending) {
            if (over) pending = 4;
            else if (it >= P.max_iter || nb == 0) pending = 3;                   // :1300, :1304-1305
            // numpy's min propagates NaN: with a nodata cell on the border neither `eborder_nopits.min() < epit_border`
            // nor `eborder == emin` can hold -- no non-pit drain, no growth; only a lower pit cell can still drain it
            else if (has_nan) pending = has_p ? 2 : 3;
            else if (has_np) pending = 1;                                        // :1312-1316
            else if (has_p) pending = 2;                                         // :1317-1320
            else {
                // the minimum, then the entries equal to it: two passes over the list with eight independent LDS reads in
                // flight (a one-entry-at-a-time loop waits out the LDS latency 2 x nb times per round)
                double mn = INFINITY;
                for (int k0 = 0; k0 < nb; k0 += 8) {
                    double v[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) v[i] = k0 + i < nb ? le[(k0 + i) * LN_T] : INFINITY;
#pragma unroll
                    for (int i = 0; i < 8; i++) mn = min_f64(mn, v[i]);
                }
                uint32_t eq = 0;
                for (int k0 = 0; k0 < nb; k0 += 8) {
                    double v[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) v[i] = k0 + i < nb ? le[(k0 + i) * LN_T] : INFINITY;
#pragma unroll
                    for (int i = 0; i < 8; i++) if (k0 + i < nb && v[i] == mn) eq |= 1u << (k0 + i);
                }
                // pit_area += border[eborder == emin] (:1322-1323): take them out of the list first (highest slot first: the
                // entry that fills a hole then always comes from the kept tail) ...
                int nq = 0;
                while (eq) {
                    const int k = 31 - __clz((int)eq);
                    eq &= ~(1u << k);
                    lq[nq * LN_T] = lp[k * LN_T]; nq++;
                    nb--;
                    if (k != nb) {
                        le[k * LN_T] = le[nb * LN_T]; lp[k * LN_T] = lp[nb * LN_T];
                        pitbits = (pitbits & ~(1u << k)) | (((pitbits >> nb) & 1u) << k);
                    }
                }
                // ... then the new border cells around them
                const int nb0 = nb;
                for (int j = 0; j < nq; j++) { const int pos = lq[j * LN_T]; discover(pos >> 4, pos & 15); }
                fetch_new(nb0);
                it++;
            }
        }
        // the drain selection / slope arithmetic is a long divergent path: it runs for a batch of waiting lanes at once
        // (when a third of the wavefront waits, or nobody is growing any more) instead of in every trip
        const unsigned long long waiting = __ballot(pending != 0), growing = __ballot(running && !pending);
        if (prof) { tkg = clock64(); acc_g += tkg - tk1; }
        if (!(__popcll(waiting) >= LN_FIN_BATCH || (waiting && !growing))) continue;
        if (pending) {
            const int mode = pending <= 2 ? pending : 0;
            status = pending == 4 ? 3 : 2;
            if (mode) {
                // drains to the front of the list in ascending cell order (window order == cell order): one pass marks
                // them (eight LDS reads in flight), the marked entries are swapped to the front in slot order and the
                // handful of drains is then insertion-sorted by position
                uint32_t mt = 0;
                for (int k0 = 0; k0 < nb; k0 += 8) {
                    double v[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) v[i] = k0 + i < nb ? le[(k0 + i) * LN_T] : INFINITY;
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const bool isp = (pitbits >> (k0 + i)) & 1u;
                        if (k0 + i < nb && (mode == 1 ? (!isp && v[i] < epit_border) : (isp && v[i] < epit))) mt |= 1u << (k0 + i);
                    }
                }
                while (mt) {
                    const int k = __ffs((int)mt) - 1;
                    mt &= mt - 1u;
                    if (k != nd) {                                               // slot nd holds an entry that is not a drain
                        const double ek = le[k * LN_T], en = le[nd * LN_T];
                        const uint8_t pk = lp[k * LN_T], pn = lp[nd * LN_T];
                        const uint32_t bk = (pitbits >> k) & 1u, bn = (pitbits >> nd) & 1u;
                        le[k * LN_T] = en; lp[k * LN_T] = pn; le[nd * LN_T] = ek; lp[nd * LN_T] = pk;
                        pitbits = (pitbits & ~((1u << k) | (1u << nd))) | (bn << k) | (bk << nd);
                    }
                    nd++;
                }
                for (int a = 1; a < nd; a++) {                                   // insertion sort by window position
                    const double ea = le[a * LN_T]; const uint8_t pa = lp[a * LN_T]; const uint32_t ba = (pitbits >> a) & 1u;
                    int b = a - 1;
                    while (b >= 0 && lp[b * LN_T] > pa) {
                        le[(b + 1) * LN_T] = le[b * LN_T]; lp[(b + 1) * LN_T] = lp[b * LN_T];
                        pitbits = (pitbits & ~(1u << (b + 1))) | (((pitbits >> b) & 1u) << (b + 1));
                        b--;
                    }
                    le[(b + 1) * LN_T] = ea; lp[(b + 1) * LN_T] = pa;
                    pitbits = (pitbits & ~(1u << (b + 1))) | (ba << (b + 1));
                }
                // filters and slopes (see finish_pit; every slice is shorter than 16 rows)
                const int ndX = n - 1;
                const bool xy = !isnan(P.max_dist_XY) && P.max_dist_XY != 0;
                int keep = 0;
                for (int t = 0; t < nd; t++) {
                    const int pos = lp[t * LN_T];
                    const int idr = r0 + (pos >> 4), jdr = c0 + (pos & 15);
                    if (P.max_dist) {                                            // :1335-1343
                        const int di = ipit - idr, dj = jpit - jdr;
                        if (!(sqrt((double)(di * di + dj * dj)) <= (double)P.max_dist)) continue;
                    }
                    const int a = ipit < idr ? ipit : idr, b = ipit < idr ? idr : ipit;
                    double dxm;
                    if (ipit == idr) dxm = P.dX[ipit < ndX - 1 ? ipit : ndX - 1];    // _get_dX_mean :1994-1995
                    else dxm = np_pairwise_leaf15(P.dX + a, b - a) / (double)(b - a);    // .mean() :1997 (window: b - a <= 15)
                    const double dx = dxm * (double)(jpit - jdr);
                    const double dy = np_pairwise_leaf15(P.dY + a, b - a);
                    const double d = sqrt(dx * dx + dy * dy);
                    if (xy && !(d <= P.max_dist_XY)) continue;                   // :1352-1358
                    le[keep * LN_T] = pit_drop(P, epit, le[t * LN_T]) / d;             // :1361
                    lp[keep * LN_T] = (uint8_t)pos;
                    keep++;
                }
                nd = keep;
                if (nd > 0) {
                    ssum = np_pairwise_leaf_strided(le, LN_T, nd);
                    P.mag[pit] = ssum / (double)nd;                              // np.mean(s) :1370
                    P.flats[pit] = 0;                                            // :1371
                    status = 1;
                }
            }
            running = false; pending = 0;
        }
        // ---- (c) all lanes together: one slot allocation / counter update per wavefront for the pits that just ended
        if (prof) { tk2 = clock64(); acc_a += tk1 - tk0; acc_b += tk2 - tkg; }
        if (!__ballot(status != 0)) continue;
        int incl = nd;
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off); if (lane >= off) incl += o; }
        const int tot = __shfl(incl, 63);
        int32_t wbase = 0;
        if (lane == 0 && tot) wbase = atomicAdd(&P.out_count[0], tot);
        wbase = __shfl(wbase, 0);
        if (tot) {
            if ((int64_t)wbase + tot <= P.out_cap) {
                const int32_t o = wbase + incl - nd;
                for (int t = 0; t < nd; t++) {                                   // :1365-1367
                    const int pos = lp[t * LN_T];
                    P.out_src[o + t] = pit;
                    P.out_dst[o + t] = (int32_t)((int64_t)(r0 + (pos >> 4)) * m + (c0 + (pos & 15)));
                    P.out_w[o + t] = le[t * LN_T] / ssum;
                }
            } else if (lane == 0) atomicAdd(&P.out_count[3], 1);
        }
        const unsigned long long b_un = __ballot(status == 2), b_ov = __ballot(status == 3);
        if (lane == 0 && b_un) atomicAdd(&P.out_count[1], __popcll(b_un));       // :1327-1329
        if (b_ov) {
            int32_t obase = 0;
            if (lane == 0) obase = atomicAdd(&P.out_count[4], __popcll(b_ov));
            obase = __shfl(obase, 0);
            if (status == 3) {
                const int32_t slot = obase + __popcll(b_ov & lt);
                P.lane_overflow[slot] = pit;
                if (P.lane_state) {
                // the region = bitmap minus the cells still on the border list.  The round that overflowed had taken its
                // minimum cells out of the list already (they belong to the region) and has been counted in `it`; the border
                // is recomputed by the wavefront version as "unseen neighbours of the region"
                for (int k = 0; k < nb; k++) { const int pos = lp[k * LN_T]; seen[(pos >> 5) * LN_T] &= ~(1u << (pos & 31)); }
                uint32_t *rec = P.lane_state + (size_t)slot * 12;
                rec[0] = (uint32_t)it; rec[1] = 0;                                 // (the round that overflowed has been counted)
                rec[2] = (uint32_t)__double2loint(epit_border); rec[3] = (uint32_t)__double2hiint(epit_border);
#pragma unroll
                for (int w = 0; w < 8; w++) rec[4 + w] = seen[w * LN_T];
                }
            }
        }
        if (prof) acc_c += clock64() - tk2;
    }
    if (prof && lane == 0) {    // cycles per phase, summed over wavefronts (PYDEM_PITS_DEBUG=3)
        atomicAdd(P.prof + 0, (unsigned long long)acc_a); atomicAdd(P.prof + 1, (unsigned long long)acc_b);
        atomicAdd(P.prof + 2, (unsigned long long)acc_c); atomicAdd(P.prof + 3, (unsigned long long)busy);
        atomicAdd(P.prof + 4, (unsigned long long)trips); atomicAdd(P.prof + 5, (unsigned long long)acc_g);
    }
}

constexpr int W_LARGE = 640, MAXD_LARGE = 2048;

// wave-per-pit: 4 pits per 256-thread block
template <int CAP, int OCC>
__global__ __launch_bounds__(256, OCC) void k_pits_wave(PitParams P, const int32_t *__restrict__ pits, const int32_t *npits)
{
    __shared__ WaveLds<128, CAP, uint16_t> s_l[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int32_t np = *npits;
    int32_t chunk_base = 0, chunk_left = 0;
    // persistent wavefronts take the next pit from a counter: a pit needs 5 .. 300 rounds, and with a fixed stride the four
    // wavefronts of a workgroup (one LDS allocation) wait for the one that drew the long pits
    for (;;) {
        int32_t q = 0;
        if (lane == 0) q = atomicAdd(P.work_next, 1);
        q = __shfl(q, 0);
        if (q >= np) break;
        if (P.wave_index) q = P.wave_index[q];
        solve_pit_wave(P, pits[q], lane, s_l[wave], chunk_base, chunk_left, P.lane_state ? P.lane_state + (size_t)q * 12 : nullptr);
        wave_sync();
    }
}

// the same with a 256x256 window and room for 2048 border cells: one pit per 64-thread workgroup
__global__ __launch_bounds__(64) void k_pits_wave_big(PitParams P, const int32_t *__restrict__ pits, const int32_t *npits)
{
    __shared__ WaveLds<256, 2048, uint32_t> s_l;
    const int lane = threadIdx.x;
    const int32_t np = *npits;
    int32_t chunk_base = 0, chunk_left = 0;
    for (int32_t q = blockIdx.x; q < np; q += gridDim.x) {
        solve_pit_wave(P, pits[q], lane, s_l, chunk_base, chunk_left);
        wave_sync();
    }
}

// workgroup-per-pit with the full-radius window in dynamic LDS (3 * 640*640/8 = 153.6 KB)
// (threads per pit: the tier runs a few dozen plateau pits, one per CU, each for up to 300 rounds of bitmap scans over a window of
// up to 12 800 words with a load per border cell -- the rounds are as long as a thread's share of the scan)
#ifndef PYDEM_BLOCK_NT
#define PYDEM_BLOCK_NT 512
#endif
constexpr int BLOCK_NT = PYDEM_BLOCK_NT;
__global__ __launch_bounds__(BLOCK_NT) void k_pits_block(PitParams P, const int32_t *__restrict__ pits, const int32_t *npits,
                                                    int32_t *g_dl, double *g_dxy, double *g_sv)
{
    constexpr int WORDS = W_LARGE * W_LARGE / 32;
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn[];
    __shared__ double redd[BLOCK_NT / 64];
    __shared__ int redi[BLOCK_NT / 64];
    __shared__ int flag[4];
    __shared__ PwFrame stk[16];
    int32_t chunk_base = 0, chunk_left = 0;
    const int32_t np = *npits;
    for (int32_t q = blockIdx.x; q < np; q += gridDim.x) {
        solve_pit<BLOCK_NT, W_LARGE, MAXD_LARGE>(P, pits[q], threadIdx.x, dyn, dyn + WORDS, dyn + 2 * WORDS,
                                             g_dl + (size_t)blockIdx.x * MAXD_LARGE, g_dxy + (size_t)blockIdx.x * MAXD_LARGE,
                                             g_sv + (size_t)blockIdx.x * MAXD_LARGE, redd, redi, flag, chunk_base, chunk_left, stk);
        __syncthreads();
    }
}

// (16 cells per thread; the elevation is only read under a flat)
__global__ void k_pitmask(const uint8_t *__restrict__ flats, const double *__restrict__ elev, int64_t NN, uint8_t *pitmask)
{
    const int64_t nvec = NN >> 4, stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t v = t0; v < nvec; v += stride) {
        const uint4 f4 = reinterpret_cast<const uint4 *>(flats)[v];
        const uint32_t fw[4] = {f4.x, f4.y, f4.z, f4.w};
        uint32_t pw[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t p = 0;
            if (fw[q])
#pragma unroll
                for (int b = 0; b < 4; b++)
                    if (((fw[q] >> (8 * b)) & 0xFFu) && elev[v * 16 + q * 4 + b] > 0) p |= 1u << (8 * b);      // :1284
            pw[q] = p;
        }
        reinterpret_cast<uint4 *>(pitmask)[v] = make_uint4(pw[0], pw[1], pw[2], pw[3]);
    }
    for (int64_t c = (nvec << 4) + t0; c < NN; c += stride) pitmask[c] = flats[c] && (elev[c] > 0);
}

// same block-aggregated compaction as the flats stage (mask -> list of cell ids)
__global__ __launch_bounds__(256) void k_compact_mask(const uint8_t *__restrict__ mask, int64_t NN,
                                                       int32_t *__restrict__ list, int32_t *__restrict__ count)
{
    // 4 x 64 cells per thread (sixteen 16 B loads), 64 Ki cells per block trip and ONE atomic for them: the counter is a
    // single address, whose returning atomics the L2 serialises at ~12 ns each (16 Ki-cell trips: 16384 atomics = 200 us
    // at 16384^2, four times the time the 1 B/cell takes to read)
    __shared__ int32_t wave_tot[4][4];
    __shared__ int32_t blk_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t base = (int64_t)blockIdx.x * 65536; base < NN; base += (int64_t)gridDim.x * 65536) {
        unsigned long long bits[4];   // bit k of bits[j] set <=> cell base + j * 16384 + threadIdx.x * 64 + k is set
        int32_t mine[4], incl[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int64_t c0 = base + j * 16384 + (int64_t)threadIdx.x * 64;
            bits[j] = 0;
            if (c0 + 64 <= NN) {
                uint4 v[4];
#pragma unroll
                for (int q = 0; q < 4; q++) v[q] = *reinterpret_cast<const uint4 *>(mask + c0 + 16 * q);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t w[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
                    for (int k = 0; k < 16; k++)
                        bits[j] |= (unsigned long long)(((w[k >> 2] >> (8 * (k & 3))) & 0xffu) ? 1u : 0u) << (16 * q + k);
                }
            } else {
                for (int k = 0; k < 64; k++)
                    if (c0 + k < NN && mask[c0 + k]) bits[j] |= 1ull << k;
            }
            mine[j] = __popcll(bits[j]);
            incl[j] = mine[j];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int32_t o = __shfl_up(incl[j], off);
                if (lane >= off) incl[j] += o;
            }
            if (lane == 63) wave_tot[j][wave] = incl[j];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int32_t tot = 0;
            for (int j = 0; j < 4; j++) for (int k = 0; k < 4; k++) tot += wave_tot[j][k];
            blk_base = tot ? atomicAdd(count, tot) : 0;
        }
        __syncthreads();
        int32_t off = blk_base;
#pragma unroll
        for (int j = 0; j < 4; j++) {           // ascending cell order within the trip: sub-chunk, wavefront, lane, bit
            int32_t o = off + incl[j] - mine[j];
            for (int k = 0; k < wave; k++) o += wave_tot[j][k];
            const int64_t c0 = base + j * 16384 + (int64_t)threadIdx.x * 64;
            unsigned long long b = bits[j];
            while (b) {
                const int k = __ffsll((long long)b) - 1;
                b &= b - 1;
                list[o++] = (int32_t)(c0 + k);
            }
            off += wave_tot[j][0] + wave_tot[j][1] + wave_tot[j][2] + wave_tot[j][3];
        }
        __syncthreads();
    }
}

// keep-filter of _mk_adjacency_matrix applied to the pit edges (:1136-1137) + 64-bit sort keys, COMPACTED: unused output
// slots (every wavefront of the pit tiers leaves a partly used chunk behind: a third of the raw slots at 16384^2) and
// dropped edges never reach the sorts.  The order of the compacted entries does not matter (the sorts define it); one
// atomic per 1024 raw slots.
__global__ __launch_bounds__(256) void k_pit_keys(const int32_t *__restrict__ src, const int32_t *__restrict__ dst, const double *__restrict__ w,
                                                  const double *__restrict__ elev, int32_t ne, uint64_t *key_out, uint64_t *key_in, int32_t *idx,
                                                  int32_t *count, int kb)
{
    // keys of 2 * kb bits (kb = bits of a cell id): the radix sorts run over 2 * kb bits instead of 64 (seven 8-bit passes
    // instead of eight at 16384^2)
    constexpr int PER = 4;
    __shared__ int32_t wave_tot[4];
    __shared__ int32_t blk_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t base = (int64_t)blockIdx.x * (256 * PER); base < ne; base += (int64_t)gridDim.x * (256 * PER)) {
        uint32_t keep = 0;
        int32_t sv[PER], dv[PER];
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const int64_t e = base + j * 256 + threadIdx.x;
            sv[j] = -1; dv[j] = 0;
            if (e < ne) { sv[j] = src[e]; dv[j] = dst[e]; }
        }
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const int64_t e = base + j * 256 + threadIdx.x;
            if (sv[j] >= 0) {
                const double we = w[e];
                if (!isnan(we) && we > 1e-8 && elev[dv[j]] <= elev[sv[j]]) keep |= 1u << j;
            }
        }
        const int32_t mine = __popc(keep);
        int32_t incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int32_t o = __shfl_up(incl, off); if (lane >= off) incl += o; }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int32_t tot = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
            blk_base = tot ? atomicAdd(count, tot) : 0;
        }
        __syncthreads();
        int32_t o = blk_base + incl - mine;
        for (int k = 0; k < wave; k++) o += wave_tot[k];
#pragma unroll
        for (int j = 0; j < PER; j++)
            if (keep & (1u << j)) {
                key_out[o] = ((uint64_t)(uint32_t)sv[j] << kb) | (uint32_t)dv[j];
                key_in[o] = ((uint64_t)(uint32_t)dv[j] << kb) | (uint32_t)sv[j];
                idx[o] = (int32_t)(base + j * 256 + threadIdx.x);
                o++;
            }
        __syncthreads();
    }
}

__global__ void k_pit_gather(const uint64_t *__restrict__ keys, const int32_t *__restrict__ idx, const double *__restrict__ w,
                             int32_t ne, int swap, int32_t *a, int32_t *b, double *wo, int32_t *dup, int kb)
{
    for (int32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += gridDim.x * blockDim.x) {
        const uint64_t k = keys[e];
        // a pit's drains are a set and every pit is solved once, so (src, dst) is unique and the sorted order does not depend on
        // the (scheduling-dependent) order of the compacted entries; counted so that a violation is an error, not a silent
        // change of the summation order
        if (dup && e > 0 && keys[e - 1] == k) atomicAdd(dup, 1);
        const int32_t hi = (int32_t)(k >> kb), lo = (int32_t)(k & ((1ull << kb) - 1ull));
        a[e] = swap ? lo : hi;      // a = src, b = dst in both views
        b[e] = swap ? hi : lo;
        wo[e] = w[idx[e]];
    }
}

template <typename T>
int dev_realloc(pydem_tile *t, T **p, size_t count)
{
    if (*p) { plane_give(t->device, *p); *p = nullptr; }      // (back to the device's free list: tile.hip)
    return tile_alloc(t, p, count);
}

}  // namespace

int stage_pits(pydem_tile *t, const pydem_options *opt)
{
    const int n = (int)t->n, m = (int)t->m;
    if (opt->drain_pits_max_iter > 300) {
        pydem_set_error("drain_pits_max_iter > 300 is not supported by the device window (got %d)", opt->drain_pits_max_iter);
        return -2;
    }
    HIP_TRY(hipEventRecord(t->ev[4], t->stream));
    PYDEM_TRY(tile_alloc(t, &t->flat0, (size_t)t->NN));       // reused as the frozen pit mask
    PYDEM_TRY(tile_alloc(t, &t->flatlist, (size_t)t->NN));    // reused as the pit list
    PYDEM_TRY(tile_alloc(t, &t->labels, (size_t)t->NN));      // reused as the overflow list
    PYDEM_TRY(tile_alloc(t, &t->queue[0], (size_t)t->NN));    // (sweep queue, idle here) lane -> wavefront hand-over list
    int32_t *cnt = t->counters + 40;                            // [0] npits, [1..4] out_count, scratch
    HIP_TRY(hipMemsetAsync(cnt, 0, 16 * sizeof(int32_t), t->stream));
    const int big = (int)(cdiv(t->NN, 256) < 8192 ? cdiv(t->NN, 256) : 8192);
    hipLaunchKernelGGL(k_pitmask, dim3(big), dim3(256), 0, t->stream, t->flats, t->elev, t->NN, t->flat0);
    hipLaunchKernelGGL(k_compact_mask, dim3((unsigned)(cdiv(t->NN, 4096) < 4096 ? cdiv(t->NN, 4096) : 4096)), dim3(256), 0,
                       t->stream, t->flat0, t->NN, t->flatlist, cnt);
    HIP_TRY(hipMemcpyAsync(t->h_counters, cnt, sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    const int32_t npits = t->h_counters[0];
    t->tm.n_pit_edges = 0; t->tm.n_pits_undrained = 0;
    t->tm.n_pits = npits; t->tm.n_pits_row = 0; t->tm.n_pits_wave = 0; t->tm.n_pits_big = 0;
    t->pits.n_edges = 0;
    if (npits == 0) { t->tm.pits_ms = 0; return 0; }

    // raw triplets; capacity grows until everything fits
    int64_t cap = (int64_t)npits * 2 + 16384 * 4 * OUT_CHUNK + 1024;   // + one partly used chunk per wavefront
    for (int attempt = 0;; attempt++) {
        if (t->pits.raw_cap < cap) {
            PYDEM_TRY(dev_realloc(t, &t->pits.raw_src, (size_t)cap));
            PYDEM_TRY(dev_realloc(t, &t->pits.raw_dst, (size_t)cap));
            PYDEM_TRY(dev_realloc(t, &t->pits.raw_w, (size_t)cap));
            t->pits.raw_cap = cap;
        }
        HIP_TRY(hipMemsetAsync(cnt + 1, 0, 8 * sizeof(int32_t), t->stream));
        PitParams P;
        P.dbg = nullptr; P.prof = nullptr;
        const char *dbg_env = getenv("PYDEM_PITS_DEBUG");
        if (dbg_env && atoi(dbg_env) == 3) { HIP_TRY(hipMalloc(&P.prof, 128)); HIP_TRY(hipMemsetAsync(P.prof, 0, 128, t->stream)); }
        if (dbg_env && atoi(dbg_env) >= 2) HIP_TRY(hipMalloc(&P.dbg, (size_t)npits * 16));
        HIP_TRY(hipMemsetAsync(t->pits.raw_src, 0xFF, (size_t)t->pits.raw_cap * 4, t->stream));   // -1 = unused slot
        P.elev = t->elev; P.pitmask = t->flat0; P.dX = t->dX; P.dY = t->dY; P.mag = t->mag; P.flats = t->flats;
        P.n = n; P.m = m; P.max_iter = opt->drain_pits_max_iter; P.max_dist = opt->drain_pits_max_dist;
        P.min_border = opt->drain_pits_min_border; P.max_dist_XY = opt->drain_pits_max_dist_XY;
        P.elev_f32 = t->elev_f32 ? 1 : 0;
        P.out_src = t->pits.raw_src; P.out_dst = t->pits.raw_dst; P.out_w = t->pits.raw_w;
        P.out_count = cnt + 1; P.out_cap = (int32_t)(t->pits.raw_cap < INT32_MAX ? t->pits.raw_cap : INT32_MAX);
        P.overflow_list = t->labels; P.overflow_count = cnt + 3;
        P.lane_overflow = t->queue[0];
        {   // hand-over records: 12 words per pit the lane version gives up on (at most every pit); PYDEM_PITS_HANDOVER=0: restart from the pit
            static int handover = -1;
            if (handover < 0) { const char *e = getenv("PYDEM_PITS_HANDOVER"); handover = e ? atoi(e) : 1; }
            P.lane_state = nullptr;
            if (handover && (int64_t)npits * 12 <= t->NN) { PYDEM_TRY(tile_alloc(t, &t->queue[1], (size_t)t->NN)); P.lane_state = (uint32_t *)t->queue[1]; }
        }
        // pass 1: a lane per pit (16x16 window); pass 2: a wavefront per pit it handed over (64x64)
        HIP_TRY(hipMemsetAsync(cnt + 10, 0, sizeof(int32_t), t->stream));
        P.work_next = cnt + 10;
        const int gl_cap = 3072 * 128 / LN_T;
        const int gl = (int)(cdiv(npits, LN_T) < gl_cap ? cdiv(npits, LN_T) : gl_cap);      // persistent: 3 workgroups per CU, 4 deep
        hipLaunchKernelGGL(k_pits_lane, dim3(gl), dim3(LN_T), 0, t->stream, P, t->flatlist, cnt);
        HIP_TRY(hipMemcpyAsync(t->h_counters, cnt, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        const int32_t n_lane_over = t->h_counters[5];
        if (P.prof) {
            unsigned long long h[8];
            HIP_TRY(hipMemcpy(h, P.prof, 64, hipMemcpyDeviceToHost));
            fprintf(stderr, "pits/lane: %llu loop trips over all wavefronts, %.1f busy lanes per trip; cycles: refill %llu, growth %llu, drain selection %llu, output %llu\n",
                    h[4], h[4] ? (double)h[3] / (double)h[4] : 0.0, h[0], h[5], h[1], h[2]);
        }
        // pass 1b: a ROW of 16 lanes per handed-over pit (64x64 window, bucketed border: pits_row.inl) + the arithmetic of its drains a
        // lane per pit; pass 2: a wavefront per pit that outgrew the row (index list; 128x128 window, 512 border cells: only a few
        // thousand pits come this far, room matters more than occupancy); pass 3: 256x256 / 2048 cells.  The three passes read their
        // counts on the device: launched back to back, ONE host synchronisation behind them.
        int32_t n_row_over = -1, n_wave_over = 0, n_over = 0;
        P.wave_index = nullptr; P.row_overflow = nullptr; P.row_overflow_count = nullptr; P.row_rec = nullptr; P.rw_target = PYDEM_RW_TARGET;
        // PYDEM_PITS_ROW: 0 never, 2 whenever there is a pit for it, 1 (default) when the pits outnumber the wavefronts the
        // chip holds: four pits per wavefront buy throughput, and a few thousand pits (plateau terrain: config 5 hands 5362 pits
        // on, 5263 of which outgrow the row pass) all run at once either way -- the pass would only add its serial chain
        // (same-box, config 5: pit search 9.8 ms without, 11.3 ms with it).  Read per call: the tests switch it.
        const char *row_env = getenv("PYDEM_PITS_ROW");
        const int row_mode = row_env ? atoi(row_env) : 1;
        const bool use_row = row_mode == 2 || (row_mode == 1 && n_lane_over >= 4 * 256 * PYDEM_WV_OCC);
        if (use_row && n_lane_over > 0 && 8 * (int64_t)n_lane_over + 64 <= t->NN) {
            HIP_TRY(hipMemsetAsync(cnt + 9, 0, 4 * sizeof(int32_t), t->stream));                           // [9] pass 3's hand-over count, [10] work counter, [12] pass 1b's
            P.row_overflow = t->labels; P.row_overflow_count = cnt + 12;
            P.row_rec = (int2 *)(t->labels + (((size_t)t->NN / 2) & ~(size_t)63));                         // (the lists of passes 1b / 2 stay below NN / 4)
            const char *tgt_env = getenv("PYDEM_RW_TARGET");
            P.rw_target = tgt_env && atoi(tgt_env) > 0 ? atoi(tgt_env) : PYDEM_RW_TARGET;
            // (16 + 192 border cells, 4 wavefronts per SIMD, one wavefront per workgroup: same-box 16384^2, row + wavefront pass:
            // 16 + 128 cells 4.1 + 2.2 ms, 16 + 160 4.5 + 1.0, 16 + 192 4.9 + 0.5; 128-thread workgroups 0.1 ms slower)
            constexpr int per = RW_NT / 16, gr_cap = 256 * (RW_OCC * 256 / RW_NT);                         // resident: every CU full
            const int gr = (int)(cdiv(n_lane_over, per) < gr_cap ? cdiv(n_lane_over, per) : gr_cap);
            hipLaunchKernelGGL((k_pits_row<RW_CAP, RW_OCC, RW_NT>), dim3(gr), dim3(RW_NT), 0, t->stream, P, t->queue[0], cnt + 5);
            hipLaunchKernelGGL(k_pits_row_finish, dim3((unsigned)(cdiv(n_lane_over, 256) < 2048 ? cdiv(n_lane_over, 256) : 2048)), dim3(256), 0,
                               t->stream, P, t->queue[0], cnt + 5, P.row_rec);
            if (P.prof) {
                unsigned long long h[16];
                HIP_TRY(hipMemcpy(h, P.prof, 128, hipMemcpyDeviceToHost));
                fprintf(stderr, "pits/row: %llu loop trips over all wavefronts, %.2f busy / %.2f growing rows per trip, %llu refills; cycles: refill %llu, rounds %llu, drain selection %llu, output %llu\n",
                        h[12], h[12] ? (double)h[13] / (double)h[12] : 0.0, h[12] ? (double)h[14] / (double)h[12] : 0.0, h[15], h[8], h[9], h[10], h[11]);
            }
            P.wave_index = t->labels;
            P.overflow_list = t->labels + (((size_t)n_lane_over + 63) & ~(size_t)63);                      // (pass 2's own hand-over list behind the index list)
            HIP_TRY(hipMemsetAsync(cnt + 10, 0, sizeof(int32_t), t->stream));
            const int gw = (int)(cdiv(n_lane_over, 4) < 256 * 4 ? cdiv(n_lane_over, 4) : 256 * 4);
            hipLaunchKernelGGL((k_pits_wave<512, 4>), dim3(gw), dim3(256), 0, t->stream, P, t->queue[0], cnt + 12);
            const int32_t *big_in = P.overflow_list;
            P.overflow_list = t->queue[0]; P.overflow_count = cnt + 9;                                     // (the hand-over list of pass 1 is dead by then)
            P.wave_index = nullptr;
            hipLaunchKernelGGL(k_pits_wave_big, dim3(256), dim3(64), 0, t->stream, P, big_in, cnt + 3);
            HIP_TRY(hipMemcpyAsync(t->h_counters, cnt, 16 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
            HIP_TRY(hipStreamSynchronize(t->stream));
            n_row_over = t->h_counters[12]; n_wave_over = t->h_counters[3]; n_over = t->h_counters[9];
        } else {
            if (n_lane_over > 0) {
                const int gw = (int)(cdiv(n_lane_over, 4) < 256 * PYDEM_WV_OCC ? cdiv(n_lane_over, 4) : 256 * PYDEM_WV_OCC);     // resident: every CU full
                HIP_TRY(hipMemsetAsync(cnt + 10, 0, sizeof(int32_t), t->stream));
                hipLaunchKernelGGL((k_pits_wave<PYDEM_WV_CAP, PYDEM_WV_OCC>), dim3(gw), dim3(256), 0, t->stream, P, t->queue[0], cnt + 5);
                HIP_TRY(hipMemcpyAsync(t->h_counters, cnt, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
                HIP_TRY(hipStreamSynchronize(t->stream));
            }
            n_wave_over = t->h_counters[3];
            if (n_wave_over > 0) {
                // pass 3: the wavefront solver with a 256x256 window / 2048 border cells for what outgrew pass 2
                HIP_TRY(hipMemsetAsync(cnt + 9, 0, sizeof(int32_t), t->stream));
                P.overflow_list = t->queue[0]; P.overflow_count = cnt + 9;
                hipLaunchKernelGGL(k_pits_wave_big, dim3(n_wave_over < 4096 ? n_wave_over : 4096), dim3(64), 0, t->stream, P, t->labels, cnt + 3);
                HIP_TRY(hipMemcpyAsync(t->h_counters, cnt, 16 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
                HIP_TRY(hipStreamSynchronize(t->stream));
                n_over = t->h_counters[9];
            }
        }
        if (P.prof) { (void)hipFree(P.prof); P.prof = nullptr; }
        t->tm.n_pits_row = n_row_over >= 0 ? n_lane_over : 0;
        t->tm.n_pits_wave = n_row_over >= 0 ? n_row_over : n_lane_over;
        t->tm.n_pits_big = n_wave_over;
        if (dbg_env) fprintf(stderr, "pits: %d candidates, %d left the 16x16 lane window, %d left the 64x64 row pass (-1: not run), %d left the 128x128 wavefront pass, %d left the 256x256 / 2048-cell pass, %d edge slots, %d undrained\n", npits, n_lane_over, n_row_over, n_wave_over, n_over, t->h_counters[1], t->h_counters[2]);
        if (P.dbg) {
            const int nrec = t->h_counters[7];
            std::vector<int32_t> rec((size_t)nrec * 4);
            HIP_TRY(hipMemcpy(rec.data(), P.dbg, rec.size() * 4, hipMemcpyDeviceToHost));
            (void)hipFree(P.dbg); P.dbg = nullptr;
            std::vector<int> rounds, border; int reason[4] = {0, 0, 0, 0}; long long tot_rounds = 0;
            for (int i = 0; i < nrec; i++) {
                rounds.push_back(rec[4 * i]); border.push_back(rec[4 * i + 1]); reason[rec[4 * i + 2] & 3]++; tot_rounds += rec[4 * i];
            }
            {   // who spends the rounds: pits by round count, how many of each class end with drains / without / handed on
                const int edges[7] = {16, 32, 64, 128, 200, 299, 1 << 30};
                long long cr[7] = {0}, cn[7] = {0}, cd[7] = {0}, cu[7] = {0}, co[7] = {0};
                for (int i = 0; i < nrec; i++) {
                    int b = 0; while (rec[4 * i] > edges[b]) b++;
                    cr[b] += rec[4 * i]; cn[b]++;
                    if (rec[4 * i + 2] & 3) co[b]++; else if (rec[4 * i + 3] >= 0) cd[b]++; else cu[b]++;
                }
                for (int b = 0; b < 7; b++)
                    fprintf(stderr, "pits/wave: rounds <= %d: %lld pits, %lld rounds (%.1f %%); %lld drain, %lld end without a drain, %lld handed on\n",
                            edges[b] == (1 << 30) ? 300 : edges[b], cn[b], cr[b], 100.0 * cr[b] / (tot_rounds ? tot_rounds : 1), cd[b], cu[b], co[b]);
            }
            std::sort(rounds.begin(), rounds.end()); std::sort(border.begin(), border.end());
            auto pct = [&](std::vector<int> &v, double q) { return v.empty() ? 0 : v[(size_t)(q * (v.size() - 1))]; };
            fprintf(stderr, "pits/wave: %d pits, %lld rounds; rounds p50 %d p90 %d p99 %d max %d; last border p50 %d p90 %d p99 %d max %d; "
                    "ok %d, window exit %d, border cap %d, drain cap %d\n", nrec, tot_rounds, pct(rounds, .5), pct(rounds, .9),
                    pct(rounds, .99), pct(rounds, 1.), pct(border, .5), pct(border, .9), pct(border, .99), pct(border, 1.),
                    reason[0], reason[1], reason[2], reason[3]);
        }
        if (n_over > 0) {
            // second pass: workgroup per pit, 640x640 window in LDS
            const int gb = n_over < 1024 ? n_over : 1024;
            const size_t dyn = (size_t)3 * W_LARGE * W_LARGE / 8;
            // (function attributes are per device, and tiles of several devices / threads come through here: set it
            // every time -- the call is cheap next to a 640 x 640 window search)
            HIP_TRY(hipFuncSetAttribute((const void *)k_pits_block, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
            const size_t need = (size_t)gb * MAXD_LARGE * (4 + 8 + 8);
            if (t->scratch_bytes < need) {
                if (t->scratch) { HIP_TRY(hipFree(t->scratch)); t->device_bytes -= (int64_t)t->scratch_bytes; }
                HIP_TRY(dev_malloc((void **)&t->scratch, need));
                t->scratch_bytes = need; t->device_bytes += (int64_t)need;
            }
            double *g_dxy = (double *)t->scratch;
            double *g_sv = g_dxy + (size_t)gb * MAXD_LARGE;
            int32_t *g_dl = (int32_t *)(g_sv + (size_t)gb * MAXD_LARGE);
            P.overflow_list = nullptr; P.overflow_count = nullptr;
            hipLaunchKernelGGL(k_pits_block, dim3(gb), dim3(BLOCK_NT), dyn, t->stream, P, t->queue[0], cnt + 9, g_dl, g_dxy, g_sv);
            HIP_TRY(hipMemcpyAsync(t->h_counters, cnt, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
            HIP_TRY(hipStreamSynchronize(t->stream));
        }
        HIP_TRY(hipGetLastError());
        const int64_t ne = t->h_counters[1];
        if (t->h_counters[4] > 0 && ne <= t->pits.raw_cap) {
            pydem_set_error("pit->drain: %d pits exceeded the device window/drain capacity", t->h_counters[4]);
            return -5;
        }
        if (ne > t->pits.raw_cap) {       // not everything fitted: the patches are idempotent, rerun with room
            if (attempt > 3) { pydem_set_error("pit edge buffer keeps overflowing"); return -5; }
            cap = ne + 1024;
            continue;
        }
        t->pits.n_raw = ne;
        t->tm.n_pits_undrained = t->h_counters[2];
        break;
    }
    // sorted views of the kept edges: by (src,dst) for releasing targets, by (dst,src) for the pull
    const int32_t ne = (int32_t)t->pits.n_raw;
    t->tm.n_pit_edges = ne;
    if (ne > 0) {
        if (t->pits.sorted_cap < ne) {
            // `ne` counts the partly used chunks of the wavefronts as well and moves by a few hundred from run to run: sized in
            // steps of 64 Ki entries, or every run of the same tile would ask the plane cache (exact sizes, tile.hip) for
            // blocks of a size it has not seen
            const size_t sc = ((size_t)ne + 65535) & ~(size_t)65535;
            PYDEM_TRY(dev_realloc(t, &t->pits.src, sc)); PYDEM_TRY(dev_realloc(t, &t->pits.dst, sc));
            PYDEM_TRY(dev_realloc(t, &t->pits.w, sc)); PYDEM_TRY(dev_realloc(t, &t->pits.in_src, sc));
            PYDEM_TRY(dev_realloc(t, &t->pits.in_dst, sc)); PYDEM_TRY(dev_realloc(t, &t->pits.in_w, sc));
            t->pits.sorted_cap = (int64_t)sc;
        }
        // one persistent scratch block (allocating and freeing eight buffers per call costs more than the sorts)
        size_t tmp_bytes = 0;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, (uint64_t *)nullptr, (uint64_t *)nullptr, (int32_t *)nullptr,
                                                   (int32_t *)nullptr, ne, 0, 64, t->stream));
        const size_t ne8 = ((size_t)ne * 8 + 255) & ~(size_t)255, ne4 = ((size_t)ne * 4 + 255) & ~(size_t)255;
        const size_t tmp_al = (tmp_bytes + 255) & ~(size_t)255;
        const size_t need_sort = 4 * ne8 + 3 * ne4 + 2 * tmp_al + 256;          // (two scratch areas: the sorts run side by side)
        if (t->pits.sort_bytes < need_sort) {
            if (t->pits.sort_buf) { HIP_TRY(hipFree(t->pits.sort_buf)); t->device_bytes -= (int64_t)t->pits.sort_bytes; }
            HIP_TRY(dev_malloc((void **)&t->pits.sort_buf, need_sort + need_sort / 4));
            t->pits.sort_bytes = need_sort + need_sort / 4; t->device_bytes += (int64_t)t->pits.sort_bytes;
        }
        char *sb = (char *)t->pits.sort_buf;
        uint64_t *k1 = (uint64_t *)sb, *k2 = (uint64_t *)(sb + ne8), *k1s = (uint64_t *)(sb + 2 * ne8), *k2s = (uint64_t *)(sb + 3 * ne8);
        int32_t *idx = (int32_t *)(sb + 4 * ne8), *i1 = (int32_t *)(sb + 4 * ne8 + ne4), *i2 = (int32_t *)(sb + 4 * ne8 + 2 * ne4);
        void *tmp = sb + 4 * ne8 + 3 * ne4;
        int kb = 1;
        while (((int64_t)1 << kb) < t->NN) kb++;          // bits of a cell id
        HIP_TRY(hipMemsetAsync(cnt + 10, 0, sizeof(int32_t), t->stream));
        hipLaunchKernelGGL(k_pit_keys, dim3((unsigned)(cdiv(ne, 1024) < 8192 ? cdiv(ne, 1024) : 8192)), dim3(256), 0, t->stream, t->pits.raw_src,
                           t->pits.raw_dst, t->pits.raw_w, t->elev, ne, k1, k2, idx, cnt + 10, kb);
        HIP_TRY(hipMemcpyAsync(t->h_counters, cnt + 10, sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        const int32_t nk = t->h_counters[0];        // kept edges
        if (nk > 0) {
            // (the temporary storage of both sorts is the tail of the persistent block, sized for ne >= nk entries)
            const int g = (int)(cdiv(nk, 256) < 1024 ? cdiv(nk, 256) : 1024);
            // the two sorts share nothing but their read-only inputs (the host has just waited for k_pit_keys): the (dst, src) one and its
            // gather go to the side stream -- behind the graph kernels if those still run, which the main stream waits for anyway --
            // and the main stream waits for them before the host looks at the duplicate counter
            void *tmp2 = (char *)tmp + tmp_al;
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp2, tmp_bytes, k2, k2s, idx, i2, nk, 0, 2 * kb, t->stream2));
            hipLaunchKernelGGL(k_pit_gather, dim3(g), dim3(256), 0, t->stream2, k2s, i2, t->pits.raw_w, nk, 1, t->pits.in_src,
                               t->pits.in_dst, t->pits.in_w, (int32_t *)nullptr, kb);
            HIP_TRY(hipEventRecord(t->ev_snap, t->stream2));
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, k1, k1s, idx, i1, nk, 0, 2 * kb, t->stream));
            HIP_TRY(hipMemsetAsync(cnt + 11, 0, sizeof(int32_t), t->stream));
            hipLaunchKernelGGL(k_pit_gather, dim3(g), dim3(256), 0, t->stream, k1s, i1, t->pits.raw_w, nk, 0, t->pits.src, t->pits.dst,
                               t->pits.w, cnt + 11, kb);
            HIP_TRY(hipStreamWaitEvent(t->stream, t->ev_snap, 0));
            HIP_TRY(hipMemcpyAsync(t->h_counters, cnt + 11, sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
            HIP_TRY(hipStreamSynchronize(t->stream));
            if (t->h_counters[0] != 0) { pydem_set_error("pit search: %d duplicate pit -> drain edges", t->h_counters[0]); return -5; }
        }
        HIP_TRY(hipGetLastError());
        t->pits.n_edges = nk;
        t->tm.n_pit_edges = nk;
    }
    HIP_TRY(hipEventRecord(t->ev[5], t->stream));
    HIP_TRY(hipEventSynchronize(t->ev[5]));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, t->ev[4], t->ev[5]));
    t->tm.pits_ms = ms;
    return 0;
}
