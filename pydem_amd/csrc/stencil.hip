// stencil.hip -- K1: Tarboton D-infinity slope magnitude + direction (3x3 stencil, 8 facets).
//
// Replaces _tarboton_slopes_directions (reference pydem/dem_processing.py:1753-1903) with its
// helpers _get_d1_d2 (:1905-1938) and _calc_direction (:1942-1991).  The reference makes 8
// full-array numpy passes (one per facet) plus 4 edge and 4 corner passes; here one kernel
// handles every interior cell (all 8 facets in registers, elevation staged through an LDS halo
// tile) and a second O(perimeter) kernel applies the edge/corner rules.
//
// Bounded by HBM: algorithmic traffic 24 B/cell (read elev 8, write mag 8 + direction 8), plus
// 1 B/cell for the flat0 mask consumed by the flats stage.  No MFMA: there is no contraction.
// Compiled with -ffp-contract=off: the facet arithmetic must round exactly like numpy's
// separate multiply/add ufuncs or exact ties on integer DEMs break differently.
#include "internal.h"
#include <float.h>
#include <stdlib.h>
#include <string.h>

#define PI_D 3.141592653589793

namespace {

constexpr int TX = 64;    // tile columns (one wavefront wide: 512 B rows)
constexpr int TY = 32;    // tile rows
constexpr int RPT = 8;    // rows marched per thread (256 threads = 64 x 4)
constexpr int LW = TX + 2;

// winner bookkeeping across the 8 facets: `rad2 > mag` with strict '>' so the first facet wins
// exact ties (dem_processing.py:1986-1989)
struct Best {
    double rad2;   // best squared magnitude so far (-1 = none)
    double s1, s2; // slopes of the winning facet (for the final atan2)
    double theta;  // table angle of the winning facet
    int k;         // winning facet
    int kind;      // 1: r = 0 (cardinal), 2: r = theta (diagonal), 3: r = atan2(s2, s1)
};

// One facet of _calc_direction (:1958-1989).  s1, s2, sd are the three slopes; d1, d2 the
// spacings (only used to decide r > theta by cross-multiplication instead of an arctangent:
// for s1 > 0, s2 > 0: atan2(s2, s1) > atan2(d2, d1)  <=>  s2*d1 > s1*d2; when the two
// products agree to within a few ulps the comparison is re-done with atan2 exactly as the
// reference does, so exact ties (s1 == s2, d1 == d2 on integer DEMs) resolve identically).
__device__ __forceinline__ void facet(double s1, double s2, double sd, double d1, double d2,
                                      double theta, int k, Best &b)
{
    const double s1_2 = s1 * s1;
    const bool s1le = s1 <= 0, s2le = s2 <= 0, s1gt = s1 > 0, s2gt = s2 > 0;
    double rad2 = s1_2 + s2 * s2;
    int kind = 3;
    bool rgt = false;
    if (s1gt && s2gt) {
        const double a = s2 * d1, c = s1 * d2;
        rgt = a > c;
        if (fabs(a - c) <= 8.0 * DBL_EPSILON * fmax(a, c)) rgt = atan2(s2, s1) > theta;
    }
    if ((s1le && s2gt) || rgt) { rad2 = sd * sd; kind = 2; }           // I1 :1973-1976
    if (s1gt && s2le) { rad2 = s1_2; kind = 1; }                       // I2 :1978-1981 (r < 0 implies s2 < 0)
    if (s1le && (s2le || (s2gt && sd <= 0))) rad2 = -1.0;              // I3 :1983-1984
    if (rad2 > b.rad2) {                                               // I4 :1986-1989
        b.rad2 = rad2; b.s1 = s1; b.s2 = s2; b.theta = theta; b.k = k; b.kind = kind;
    }
}

// atan2(y, x) for y > 0, x > 0 (the only case an unclamped facet can produce, see facet()).
// ocml's general atan2 costs ~130 fp64 operations per cell -- more than a quarter of the whole
// stencil; this one is ~45: q = min/max in (0, 1], c = nearest multiple of 1/16,
// atan(q) = atan(c) + atan(u), u = (q - c)/(1 + q*c), |u| <= 1/32, odd Taylor series to u^11
// (truncation < 1e-19 relative), table of correctly rounded atan(k/16).  Error <= ~1.5 ulp.
__device__ __constant__ const double ATAN_16[17] = {
    0, 0.06241880999595735, 0.12435499454676144, 0.18534794999569476, 0.24497866312686414, 0.30288486837497142,
    0.35877067027057225, 0.41241044159738732, 0.46364760900080609, 0.51238946031073773, 0.55859931534356244,
    0.60228734613496415, 0.64350110879328437, 0.68231655487474807, 0.71882999962162453, 0.75315128096219441,
    0.78539816339744828};

__device__ __forceinline__ double atan2_pos(double y, double x)
{
    // atan(num/den), num <= den: c = nearest multiple of 1/16 to the quotient (a raw reciprocal is good enough to pick
    // it), u = (num/den - c) / (1 + c num/den) = (num - c den) / (den + c num): one division
    const bool swap = y > x;
    const double num = swap ? x : y, den = swap ? y : x;
    const double kf = rint(num * __builtin_amdgcn_rcp(den) * 16.0);
    const double c = kf * 0.0625;
    const double u = __builtin_fma(-c, den, num) / __builtin_fma(c, num, den);
    const double w = u * u;
    double p = __builtin_fma(w, -1.0 / 11, 1.0 / 9);
    p = __builtin_fma(w, p, -1.0 / 7);
    p = __builtin_fma(w, p, 1.0 / 5);
    p = __builtin_fma(w, p, -1.0 / 3);
    p = __builtin_fma(w * u, p, u);
    const double a = ATAN_16[(int)kf] + p;
    return swap ? PI_D / 2 - a : a;
}

// direction of the winner: r * ang[1] + ang[0] * pi / 2 (:1989); ang_adj table :184-193
__device__ __forceinline__ double direction_of(int k, int kind, double s1, double s2, double theta)
{
    if (k < 0) return -1.0;
    const int a0 = (k + 1) >> 1;                   // 0,1,1,2,2,3,3,4
    const double a1 = (k & 1) ? -1.0 : 1.0;        // 1,-1,1,-1,...
    double r = 0.0;
    if (kind == 2) r = theta;
    else if (kind == 3) r = atan2_pos(s2, s1);
    return r * a1 + (double)a0 * PI_D / 2;
}

__device__ __forceinline__ double winner_direction(const Best &b) { return direction_of(b.k, b.kind, b.s1, b.s2, b.theta); }

// Lean facet for the marching kernel: only (rad2, code = 4*k + kind) is tracked; the winner's
// slopes are re-selected once at the end.  Same decisions as facet() except that `r > theta` is the
// plain cross-multiplication (exact ties compare equal on both sides and stay unclamped, like
// atan2(s, s) == atan2(d, d) in the reference).
struct Acc { double rad2; int code; };

// squares are passed in: an edge quotient enters up to two facets of a cell, its square is computed once
__device__ __forceinline__ void facet_lean(double s1, double s2, double sd, double s1sq, double s2sq, double sdsq,
                                           double d1, double d2, int k, Acc &acc)
{
    const bool s1gt = s1 > 0, s1le = s1 <= 0, s2gt = s2 > 0, s2le = s2 <= 0;
    const bool rgt = s1gt && s2gt && (s2 * d1 > s1 * d2);
    const bool diag = (s1le && s2gt) || rgt;                    // I1 :1973-1976
    const bool card = s1gt && s2le;                             // I2 :1978-1981
    const bool none = s1le && (s2le || (s2gt && sd <= 0));      // I3 :1983-1984 (rad2 = -1 never beats the running maximum)
    const double cand = card ? s1sq : (diag ? sdsq : s1sq + s2sq);
    const bool upd = !none && cand > acc.rad2;                  // I4 :1986-1989
    acc.rad2 = upd ? cand : acc.rad2;
    acc.code = upd ? 4 * k + (card ? 1 : (diag ? 2 : 3)) : acc.code;
}

// x / d for a per-row spacing d with r = RN(1/d) from the host: q = RN(x*r), one exact remainder, one
// correction -- Markstein's sequence returns the correctly rounded quotient (checked against 4e8
// IEEE divisions on the host, oracle/ and tests compare the results bit for bit), 3 full-rate
// operations instead of ~11 with a quarter-rate v_rcp_f64
__device__ __forceinline__ double div_row(double x, double d, double r)
{
    const double q = x * r;
    const double rem = __builtin_fma(-q, d, x);
    return __builtin_fma(rem, r, q);
}

// elevation difference in the dtype the reference subtracts in: numpy keeps a float32 DEM float32 through
// `data[slc0] - data[slc1]` and only the division by the float64 spacing promotes (_calc_direction :1958-1962), so a
// float32 tile (uploaded PYDEM_F32, conditioning off) rounds every difference to 24 bits first.  The casts are exact
// (the device copy holds the float32 values), RN subtraction is sign-symmetric in either width.
template <bool F32>
__device__ __forceinline__ double zsub(double a, double b)
{
    if (F32) return (double)((float)a - (float)b);
    return a - b;
}

// all 8 facets of an interior cell.  tn = spacing row i-1 (facets 0-3), ts = row i (facets 4-7)
// (_get_d1_d2 :1912-1924: facets 0,3,4,7 use d1 = dX, d2 = dY; facets 1,2,5,6 d1 = dY, d2 = dX)
template <bool F32>
__device__ __forceinline__ void eight_facets(double z0, double zN, double zS, double zE, double zW,
                                             double zNE, double zNW, double zSW, double zSE,
                                             const RowTab &tn, const RowTab &ts, Best &b)
{
    const double sdNE = zsub<F32>(z0, zNE) / tn.hyp, sdNW = zsub<F32>(z0, zNW) / tn.hyp;
    const double sdSW = zsub<F32>(z0, zSW) / ts.hyp, sdSE = zsub<F32>(z0, zSE) / ts.hyp;
    const double s1N = zsub<F32>(z0, zN) / tn.dY, s1S = zsub<F32>(z0, zS) / ts.dY;
    facet(zsub<F32>(z0, zE) / tn.dX, zsub<F32>(zE, zNE) / tn.dY, sdNE, tn.dX, tn.dY, tn.thA, 0, b);
    facet(s1N, zsub<F32>(zN, zNE) / tn.dX, sdNE, tn.dY, tn.dX, tn.thB, 1, b);
    facet(s1N, zsub<F32>(zN, zNW) / tn.dX, sdNW, tn.dY, tn.dX, tn.thB, 2, b);
    facet(zsub<F32>(z0, zW) / tn.dX, zsub<F32>(zW, zNW) / tn.dY, sdNW, tn.dX, tn.dY, tn.thA, 3, b);
    facet(zsub<F32>(z0, zW) / ts.dX, zsub<F32>(zW, zSW) / ts.dY, sdSW, ts.dX, ts.dY, ts.thA, 4, b);
    facet(s1S, zsub<F32>(zS, zSW) / ts.dX, sdSW, ts.dY, ts.dX, ts.thB, 5, b);
    facet(s1S, zsub<F32>(zS, zSE) / ts.dX, sdSE, ts.dY, ts.dX, ts.thB, 6, b);
    facet(zsub<F32>(z0, zE) / ts.dX, zsub<F32>(zE, zSE) / ts.dY, sdSE, ts.dX, ts.dY, ts.thA, 7, b);
}

__device__ __forceinline__ Best best_init()
{
    Best b; b.rad2 = -1.0; b.s1 = 0; b.s2 = 0; b.theta = 0; b.k = -1; b.kind = 0;
    return b;
}

// ---------------------------------------------------------------------------------------------
// interior kernel: block = 64 x 4 threads, tile = 64 columns x 32 rows, LDS halo tile 66 x 34.
// Tiles are column-aligned to multiples of 64 so each wavefront stores whole 512 B row segments.
// XCD-aware tile order: consecutive block ids land on different XCDs (b % 8), so block b is
// remapped to tile (b % 8) * tiles_per_xcd + b / 8: each XCD walks a contiguous band of tiles
// and neighbouring tiles share halo lines in the same L2.
// ---------------------------------------------------------------------------------------------
template <bool XCD_SWIZZLE, bool F32>
__global__ __launch_bounds__(256) void k_stencil_interior(const double *__restrict__ elev, int n, int m,
                                                          const RowTab *__restrict__ rowtab,
                                                          double *__restrict__ mag, double *__restrict__ dir,
                                                          uint8_t *__restrict__ flat0, int tiles_x, int tiles_total)
{
    __shared__ double tile[(TY + 2) * LW];
    int tid = blockIdx.x;
    if (XCD_SWIZZLE) {
        const int per = (tiles_total + 7) >> 3;
        tid = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
        if (tid >= tiles_total) return;
    }
    const int by = tid / tiles_x, bx = tid - by * tiles_x;
    const int j0 = bx * TX, i0 = by * TY;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;

    for (int idx = threadIdx.x; idx < (TY + 2) * LW; idx += 256) {
        const int lr = idx / LW, lc = idx - lr * LW;
        const int gi = i0 - 1 + lr, gj = j0 - 1 + lc;
        double v = 0.0;
        if (gi >= 0 && gi < n && gj >= 0 && gj < m) v = elev[(size_t)gi * m + gj];
        tile[idx] = v;
    }
    __syncthreads();

    const int j = j0 + tx;
    const int lc = tx + 1;
    int lr = ty * RPT + 1;                       // LDS row of the first cell of this thread
    // 3x3 window, rolled down the column
    double aW = tile[(lr - 1) * LW + lc - 1], a0 = tile[(lr - 1) * LW + lc], aE = tile[(lr - 1) * LW + lc + 1];
    double cW = tile[lr * LW + lc - 1], c0 = tile[lr * LW + lc], cE = tile[lr * LW + lc + 1];
#pragma unroll
    for (int q = 0; q < RPT; q++, lr++) {
        const int i = i0 + ty * RPT + q;
        const double bW = tile[(lr + 1) * LW + lc - 1], b0 = tile[(lr + 1) * LW + lc], bE = tile[(lr + 1) * LW + lc + 1];
        if (i >= 1 && i < n - 1 && j >= 1 && j < m - 1) {
            const RowTab tn = rowtab[i - 1], ts = rowtab[i];
            Best b = best_init();
            eight_facets<F32>(c0, a0, b0, cE, cW, aE, aW, bW, bE, tn, ts, b);
            const size_t c = (size_t)i * m + j;
            mag[c] = b.rad2 > 0 ? sqrt(b.rad2) : b.rad2;           // :1901
            dir[c] = winner_direction(b);
            flat0[c] = (b.rad2 == -1.0);
        }
        aW = cW; a0 = c0; aE = cE;
        cW = bW; c0 = b0; cE = bE;
    }
}

// ---------------------------------------------------------------------------------------------
// Variant 2 -- "march": one lane per column, rows marched in registers, quotients shared.
// Every slope the 8 facets need is an EDGE quotient: (z_a - z_b)/d for two neighbouring cells and
// a row spacing.  Per cell there are only 5 distinct ones (E-edge over the north and the south
// spacing, the S-edge, the SE and SW diagonals); the other 13 of the 18 divisions of variant 1 are
// the same numbers seen from the neighbouring cell (IEEE subtraction and division are sign-
// symmetric, so -(a-b)/d == (b-a)/d bit for bit).  Each lane computes its 5 quotients when a row
// enters its 3-row window and fetches the neighbours' copies with wavefront lane shifts; the
// arithmetic that reaches `facet()` is bit-identical to variant 1.  A wavefront covers 64 columns
// and produces 62 (lanes 0 and 63 are halo), rows are 512 B coalesced loads, no LDS.
// ---------------------------------------------------------------------------------------------
// lane shifts as DPP moves (v_mov_b32_dpp wave_shr:1 / wave_shl:1 -- gfx9 wavefront shifts; 2 VALU
// moves per double, no LDS crossbar round trip).  The edge lanes keep their own value (halo lanes).
__device__ __forceinline__ double lane_prev(double x)    // value held by lane-1 (column j-1)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_next(double x)    // value held by lane+1 (column j+1)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

constexpr int MARCH_ROWS = 128;   // output rows per wavefront on large tiles (fewer on small ones: the chip wants >= ~10 k wavefronts)

template <bool F32>
__global__ __launch_bounds__(256) void k_stencil_march(const double *__restrict__ elev, int n, int m,
                                                       const RowTab *__restrict__ rowtab,
                                                       double *__restrict__ mag, double *__restrict__ dir,
                                                       uint8_t *__restrict__ flat0, int strips, int chunks, int rows_per_wave)
{
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));   // wavefront id (scalar: row tables via s_load)
    if (wid >= strips * chunks) return;
    const int chunk = wid / strips, strip = wid - chunk * strips;  // consecutive waves walk along a row band
    const int j = strip * 62 + lane;                                // this lane's column (lane 0 / 63 = halo)
    const int i0 = 1 + chunk * rows_per_wave;                       // first output row
    const int i1 = (i0 + rows_per_wave < n - 1) ? i0 + rows_per_wave : n - 1;   // one past the last output row
    const bool colok = j < m;
    const int jc = colok ? j : m - 1;
    const double *col = elev + jc;

    // window rows i-1, i (already "entered"), then row i+1 enters at every step
    double zN = col[(size_t)(i0 - 1) * m], z0 = col[(size_t)i0 * m];
    // quantities of the rows already in the window, as if they had entered one by one
    double zE0 = lane_next(z0), zW0 = lane_prev(z0);
    const RowTab t0 = rowtab[i0 - 1];
    double hEs_N = div_row(zsub<F32>(zN, lane_next(zN)), t0.dX, t0.rdX);   // E-edge of row i-1 over its south spacing dX[i-1]
    double hEn_0 = div_row(zsub<F32>(z0, zE0), t0.dX, t0.rdX);             // E-edge of row i over its north spacing dX[i-1]
    double v_N = div_row(zsub<F32>(zN, z0), t0.dY, t0.rdY);                // vertical edge (i-1) -> i
    double dSE_N = div_row(zsub<F32>(zN, zE0), t0.hyp, t0.rhyp);           // diagonal (i-1,j) -> (i,j+1)
    double dSW_N = div_row(zsub<F32>(zN, zW0), t0.hyp, t0.rhyp);           // diagonal (i-1,j) -> (i,j-1)
    double hEs_0 = (i0 <= n - 2) ? div_row(zsub<F32>(z0, zE0), rowtab[i0].dX, rowtab[i0].rdX) : 0.0;   // E-edge of row i over its south spacing dX[i]
    // neighbours' copies
    double hEs_N_L = lane_prev(hEs_N), dSE_N_L = lane_prev(dSE_N), v_N_L = lane_prev(v_N), v_N_R = lane_next(v_N);
    double dSW_N_R = lane_next(dSW_N), hEn_0_L = lane_prev(hEn_0), hEs_0_L = lane_prev(hEs_0);

    for (int i = i0; i < i1; i++) {
        const RowTab tn = rowtab[i - 1], ts = rowtab[i];
        // ---- row i+1 enters
        const double zS = col[(size_t)(i + 1) * m];
        const double zES = lane_next(zS), zWS = lane_prev(zS);
        const double hEn_S = div_row(zsub<F32>(zS, zES), ts.dX, ts.rdX);   // E-edge of row i+1 over its north spacing dX[i]
        const double v_0 = div_row(zsub<F32>(z0, zS), ts.dY, ts.rdY);      // vertical edge i -> i+1
        const double dSE_0 = div_row(zsub<F32>(z0, zES), ts.hyp, ts.rhyp);
        const double dSW_0 = div_row(zsub<F32>(z0, zWS), ts.hyp, ts.rhyp);
        const double hEs_S = (i + 1 <= n - 2) ? div_row(zsub<F32>(zS, zES), rowtab[i + 1].dX, rowtab[i + 1].rdX) : 0.0;
        const double hEn_S_L = lane_prev(hEn_S), v_0_L = lane_prev(v_0), v_0_R = lane_next(v_0);
        const double dSE_0_L = lane_prev(dSE_0), dSW_0_R = lane_next(dSW_0), hEs_S_L = lane_prev(hEs_S);
        // ---- the 8 facets of cell (i, j)
        Acc acc; acc.rad2 = -1.0; acc.code = -4;
        const double sdNE = -dSW_N_R, sdNW = -dSE_N_L;
        const double s1_0 = hEn_0, s2_0 = -v_N_R;          // s1=(z0-zE)/dXn  s2=(zE-zNE)/dYn
        const double s1_1 = -v_N, s2_1 = hEs_N;            // s1=(z0-zN)/dYn  s2=(zN-zNE)/dXn
        const double s2_2 = -hEs_N_L;                      //                 s2=(zN-zNW)/dXn
        const double s1_3 = -hEn_0_L, s2_3 = -v_N_L;       // s1=(z0-zW)/dXn  s2=(zW-zNW)/dYn
        const double s1_4 = -hEs_0_L, s2_4 = v_0_L;        // s1=(z0-zW)/dXs  s2=(zW-zSW)/dYs
        const double s1_5 = v_0, s2_5 = -hEn_S_L;          // s1=(z0-zS)/dYs  s2=(zS-zSW)/dXs
        const double s2_6 = hEn_S;                         //                 s2=(zS-zSE)/dXs
        const double s1_7 = hEs_0, s2_7 = v_0_R;           // s1=(z0-zE)/dXs  s2=(zE-zSE)/dYs
        // (x*x == (-x)*(-x): the squares are taken from the un-negated quotients)
        const double qNE = dSW_N_R * dSW_N_R, qNW = dSE_N_L * dSE_N_L, qSW = dSW_0 * dSW_0, qSE = dSE_0 * dSE_0;
        const double q1_1 = v_N * v_N, q1_5 = v_0 * v_0;
        facet_lean(s1_0, s2_0, sdNE, hEn_0 * hEn_0, v_N_R * v_N_R, qNE, tn.dX, tn.dY, 0, acc);
        facet_lean(s1_1, s2_1, sdNE, q1_1, hEs_N * hEs_N, qNE, tn.dY, tn.dX, 1, acc);
        facet_lean(s1_1, s2_2, sdNW, q1_1, hEs_N_L * hEs_N_L, qNW, tn.dY, tn.dX, 2, acc);
        facet_lean(s1_3, s2_3, sdNW, hEn_0_L * hEn_0_L, v_N_L * v_N_L, qNW, tn.dX, tn.dY, 3, acc);
        facet_lean(s1_4, s2_4, dSW_0, hEs_0_L * hEs_0_L, v_0_L * v_0_L, qSW, ts.dX, ts.dY, 4, acc);
        facet_lean(s1_5, s2_5, dSW_0, q1_5, hEn_S_L * hEn_S_L, qSW, ts.dY, ts.dX, 5, acc);
        facet_lean(s1_5, s2_6, dSE_0, q1_5, hEn_S * hEn_S, qSE, ts.dY, ts.dX, 6, acc);
        facet_lean(s1_7, s2_7, dSE_0, hEs_0 * hEs_0, v_0_R * v_0_R, qSE, ts.dX, ts.dY, 7, acc);
        if (lane >= 1 && lane <= 62 && j >= 1 && j < m - 1) {
            const int k = acc.code >> 2, kind = acc.code & 3;
            // slopes / table angle of the winning facet
            double w1 = s1_0, w2 = s2_0;
            w1 = k == 1 ? s1_1 : w1; w2 = k == 1 ? s2_1 : w2;
            w1 = k == 2 ? s1_1 : w1; w2 = k == 2 ? s2_2 : w2;
            w1 = k == 3 ? s1_3 : w1; w2 = k == 3 ? s2_3 : w2;
            w1 = k == 4 ? s1_4 : w1; w2 = k == 4 ? s2_4 : w2;
            w1 = k == 5 ? s1_5 : w1; w2 = k == 5 ? s2_5 : w2;
            w1 = k == 6 ? s1_5 : w1; w2 = k == 6 ? s2_6 : w2;
            w1 = k == 7 ? s1_7 : w1; w2 = k == 7 ? s2_7 : w2;
            const bool kindA = (k == 0) || (k == 3) || (k == 4) || (k == 7);
            const double th = (k < 4) ? (kindA ? tn.thA : tn.thB) : (kindA ? ts.thA : ts.thB);
            const size_t c = (size_t)i * m + j;
            mag[c] = acc.rad2 > 0 ? sqrt(acc.rad2) : acc.rad2;         // :1901
            dir[c] = direction_of(k, kind, w1, w2, th);
            flat0[c] = (acc.rad2 == -1.0);
        }
        // ---- roll the window
        zN = z0; z0 = zS;
        hEs_N = hEs_0; hEs_N_L = hEs_0_L;
        hEn_0 = hEn_S; hEn_0_L = hEn_S_L;
        hEs_0 = hEs_S; hEs_0_L = hEs_S_L;
        v_N = v_0; v_N_L = v_0_L; v_N_R = v_0_R;
        dSE_N_L = dSE_0_L; dSW_N_R = dSW_0_R;
    }
}

// ---------------------------------------------------------------------------------------------
// perimeter kernel: one thread per edge/corner cell (dem_processing.py:1779-1899).
// ---------------------------------------------------------------------------------------------
template <bool F32>
__device__ void interior_cell(const double *elev, int n, int m, const RowTab *rowtab, int i, int j,
                              double &rad2, double &d)
{
    // pre-sqrt magnitude and direction of interior cell (i, j) recomputed from global memory;
    // cells that are not interior still hold the initial -1 at the time of the copy rules
    rad2 = -1.0; d = -1.0;
    if (i < 1 || i > n - 2 || j < 1 || j > m - 2) return;
    const double *r0 = elev + (size_t)(i - 1) * m + j, *r1 = r0 + m, *r2 = r1 + m;
    Best b = best_init();
    eight_facets<F32>(r1[0], r0[0], r2[0], r1[1], r1[-1], r0[1], r0[-1], r2[-1], r2[1], rowtab[i - 1], rowtab[i], b);
    rad2 = b.rad2; d = winner_direction(b);
}

// one facet of an edge cell; mode 0: per-row spacing (left/right edges, topbot == None),
// mode 1: fixed spacing row `sr` ('top' -> 0, 'bot' -> n-2) (:1925-1934)
template <bool F32>
__device__ void edge_facet(const double *elev, int n, int m, const RowTab *rowtab, int i, int j, int k,
                           int mode, int sr, Best &b)
{
    int r;
    if (mode == 1) r = sr;
    else r = (k <= 3) ? i - 1 : i;   // facets 0-3 reference spacing row i-1, 4-7 row i (:1914-1921)
    const RowTab t = rowtab[r];
    const bool A = (k == 0 || k == 3 || k == 4 || k == 7);
    const double d1 = A ? t.dX : t.dY, d2 = A ? t.dY : t.dX, th = A ? t.thA : t.thB;
    const double z0 = elev[(size_t)i * m + j];
    const double z1 = elev[(size_t)(i + fe1r(k)) * m + (j + fe1c(k))];
    const double z2 = elev[(size_t)(i + fe2r(k)) * m + (j + fe2c(k))];
    facet(zsub<F32>(z0, z1) / d1, zsub<F32>(z1, z2) / d2, zsub<F32>(z0, z2) / t.hyp, d1, d2, th, k, b);
}

template <bool F32>
__global__ void k_stencil_perimeter(const double *__restrict__ elev, int n, int m,
                                    const RowTab *__restrict__ rowtab,
                                    double *__restrict__ mag, double *__restrict__ dir, uint8_t *__restrict__ flat0)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nper = 2 * (int64_t)m + 2 * (int64_t)(n - 2);
    if (p >= nper) return;
    int i, j;
    if (p < m) { i = 0; j = (int)p; }
    else if (p < 2 * (int64_t)m) { i = n - 1; j = (int)(p - m); }
    else if (p < 2 * (int64_t)m + (n - 2)) { i = (int)(p - 2 * (int64_t)m) + 1; j = 0; }
    else { i = (int)(p - 2 * (int64_t)m - (n - 2)) + 1; j = m - 1; }

    const double HP = PI_D / 2, P32 = 3 * PI_D / 2, TWOPI = 2 * PI_D;
    double rad2 = -1.0, d = -1.0;
    // --- copy-from-interior rules in the reference's order: left, right, top, bottom (:1782-1795).
    // For a corner the value can arrive through two hops (side copy on row 1 / n-2, then the
    // top / bottom copy).
    const bool top = (i == 0), bot = (i == n - 1), left = (j == 0), right = (j == m - 1);
    if ((left || right) && !top && !bot) {
        double r2, dd;
        interior_cell<F32>(elev, n, m, rowtab, i, left ? 1 : m - 2, r2, dd);
        const bool take = left ? (dd > HP && dd < P32) : (dd < HP || dd > P32);
        if (take) { rad2 = r2; d = dd; }
    } else {
        // row 0 copies from row 1, row n-1 from row n-2; on the corner columns row 1 / n-2 itself
        // holds whatever the side copy put there
        const int ii = top ? 1 : n - 2;
        double r2 = -1.0, dd = -1.0;
        if (left || right) {
            double r3, d3;
            interior_cell<F32>(elev, n, m, rowtab, ii, left ? 1 : m - 2, r3, d3);
            const bool take1 = left ? (d3 > HP && d3 < P32) : (d3 < HP || d3 > P32);
            if (take1) { r2 = r3; dd = d3; }
        } else {
            interior_cell<F32>(elev, n, m, rowtab, ii, j, r2, dd);
        }
        const bool take = top ? (dd > 0 && dd < PI_D) : (dd > PI_D && dd < TWOPI);
        if (take) { rad2 = r2; d = dd; }
    }
    // --- inward facets (:1800-1899); the copied value competes through the same strict '>'
    Best b = best_init();
    b.rad2 = rad2;
    int copied = (rad2 > -1.0) || (d != -1.0);
    (void)copied;
    const int sr = top ? 0 : n - 2;
    if (top && left) { edge_facet<F32>(elev, n, m, rowtab, i, j, 6, 1, sr, b); edge_facet<F32>(elev, n, m, rowtab, i, j, 7, 1, sr, b); }
    else if (top && right) { edge_facet<F32>(elev, n, m, rowtab, i, j, 4, 1, sr, b); edge_facet<F32>(elev, n, m, rowtab, i, j, 5, 1, sr, b); }
    else if (bot && left) { edge_facet<F32>(elev, n, m, rowtab, i, j, 0, 1, sr, b); edge_facet<F32>(elev, n, m, rowtab, i, j, 1, 1, sr, b); }
    else if (bot && right) { edge_facet<F32>(elev, n, m, rowtab, i, j, 2, 1, sr, b); edge_facet<F32>(elev, n, m, rowtab, i, j, 3, 1, sr, b); }
    else if (left) { const int ks[4] = {0, 1, 6, 7}; for (int q = 0; q < 4; q++) edge_facet<F32>(elev, n, m, rowtab, i, j, ks[q], 0, 0, b); }
    else if (right) { const int ks[4] = {2, 3, 4, 5}; for (int q = 0; q < 4; q++) edge_facet<F32>(elev, n, m, rowtab, i, j, ks[q], 0, 0, b); }
    else if (top) { const int ks[4] = {4, 5, 6, 7}; for (int q = 0; q < 4; q++) edge_facet<F32>(elev, n, m, rowtab, i, j, ks[q], 1, sr, b); }
    else { const int ks[4] = {0, 1, 2, 3}; for (int q = 0; q < 4; q++) edge_facet<F32>(elev, n, m, rowtab, i, j, ks[q], 1, sr, b); }
    const double dout = (b.k >= 0) ? winner_direction(b) : d;   // no facet beat the copied value
    const size_t c = (size_t)i * m + j;
    mag[c] = b.rad2 > 0 ? sqrt(b.rad2) : b.rad2;
    dir[c] = dout;
    flat0[c] = (b.rad2 == -1.0);
}

template <bool SW>
int launch_interior(pydem_tile *t)
{
    const int tiles_x = (int)cdiv(t->m, TX), tiles_y = (int)cdiv(t->n, TY);
    const int total = tiles_x * tiles_y;
    const int grid = SW ? ((total + 7) / 8) * 8 : total;
    if (t->elev_f32)
        hipLaunchKernelGGL((k_stencil_interior<SW, true>), dim3(grid), dim3(256), 0, t->stream, t->elev, (int)t->n, (int)t->m,
                           t->rowtab, t->mag, t->dir, t->flat0, tiles_x, total);
    else
        hipLaunchKernelGGL((k_stencil_interior<SW, false>), dim3(grid), dim3(256), 0, t->stream, t->elev, (int)t->n, (int)t->m,
                           t->rowtab, t->mag, t->dir, t->flat0, tiles_x, total);
    return 0;
}

}  // namespace

static int stencil_variant()
{
    static int v = -1;
    if (v < 0) { const char *e = getenv("PYDEM_STENCIL"); v = (e && !strcmp(e, "tile")) ? 1 : 2; }
    return v;
}

static void launch_stencil(pydem_tile *t)
{
    if (stencil_variant() == 1) { launch_interior<true>(t); return; }
    const int strips = (int)cdiv(t->m - 2, 62);
    int rows = MARCH_ROWS;            // every chunk re-reads two halo rows: long chunks on big tiles, enough wavefronts on small ones
    while (rows > 16 && (int64_t)strips * cdiv(t->n - 2, rows) < 12288) rows >>= 1;
    const int chunks = (int)cdiv(t->n - 2, rows);
    const int waves = strips * chunks;
    if (t->elev_f32)
        hipLaunchKernelGGL(k_stencil_march<true>, dim3((unsigned)cdiv(waves, 4)), dim3(256), 0, t->stream, t->elev, (int)t->n, (int)t->m,
                           t->rowtab, t->mag, t->dir, t->flat0, strips, chunks, rows);
    else
        hipLaunchKernelGGL(k_stencil_march<false>, dim3((unsigned)cdiv(waves, 4)), dim3(256), 0, t->stream, t->elev, (int)t->n, (int)t->m,
                           t->rowtab, t->mag, t->dir, t->flat0, strips, chunks, rows);
}

int stage_stencil(pydem_tile *t)
{
    HIP_TRY(hipEventRecord(t->ev[0], t->stream));
    launch_stencil(t);
    HIP_TRY(hipEventRecord(t->ev[1], t->stream));
    const int64_t nper = 2 * t->m + 2 * (t->n - 2);
    if (t->elev_f32)
        hipLaunchKernelGGL(k_stencil_perimeter<true>, dim3((unsigned)cdiv(nper, 128)), dim3(128), 0, t->stream, t->elev,
                           (int)t->n, (int)t->m, t->rowtab, t->mag, t->dir, t->flat0);
    else
        hipLaunchKernelGGL(k_stencil_perimeter<false>, dim3((unsigned)cdiv(nper, 128)), dim3(128), 0, t->stream, t->elev,
                           (int)t->n, (int)t->m, t->rowtab, t->mag, t->dir, t->flat0);
    HIP_TRY(hipEventRecord(t->ev[2], t->stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventSynchronize(t->ev[2]));
    float a = 0, b = 0;
    HIP_TRY(hipEventElapsedTime(&a, t->ev[0], t->ev[1]));
    HIP_TRY(hipEventElapsedTime(&b, t->ev[0], t->ev[2]));
    t->tm.stencil_kernel_ms = a;
    t->tm.slopes_directions_ms = b;
    return 0;
}

int bench_stencil(pydem_tile *t, int iters, double *avg_ms)
{
    launch_stencil(t);   // warm-up
    HIP_TRY(hipEventRecord(t->ev[0], t->stream));
    for (int q = 0; q < iters; q++) launch_stencil(t);
    HIP_TRY(hipEventRecord(t->ev[1], t->stream));
    HIP_TRY(hipEventSynchronize(t->ev[1]));
    HIP_TRY(hipGetLastError());
    float a = 0;
    HIP_TRY(hipEventElapsedTime(&a, t->ev[0], t->ev[1]));
    *avg_ms = (double)a / iters;
    return 0;
}
