// pits_row.inl -- a ROW of 16 lanes per pit: the tier between the lane version and the wavefront version
// (included by pits.hip inside its anonymous namespace; reference pydem/dem_processing.py:1269-1382, utils.py:313-340).
//
// The wavefront version serves a border of 10-60 cells with 64 lanes and pays ~170 vector and ~135 scalar instructions per
// round of ONE pit: both issue ports are 75-80 % busy and most lanes idle.  Here a wavefront carries FOUR pits, one per DPP
// row of 16 lanes: the minimum is four row-local DPP steps (no cross-row broadcast, no readlane), ballots are cut into
// 16-bit row masks, a round's expansion step covers 2 promoted cells x 8 neighbours, and everything that was wavefront-uniform
// (list end, free slots, iteration count, thresholds) lives in vector registers.  Rows are re-armed from the hand-over list
// as soon as their pit ends (persistent wavefronts, like the lane version).
//
// The border is BUCKETED: the reference takes ALL border cells at the minimum elevation per iteration (:1300-1323), and a
// list scan per round costs the same per pit whether 64 or 16 lanes do it.  So the list has a HEAD of 16 slots (one per lane of
// the row) and an unordered TAIL, with a threshold T: every tail entry is > T.  A round reads only the head: its minimum mn is the
// minimum of the whole border whenever mn <= T, and then every entry equal to mn is in the head too.  Cells that enter the border
// go to the head when they are <= T and a head slot is free, to the tail otherwise (T drops below them if they were <= T).  When
// the head is empty or its minimum is above T, the head is poured back and the ~12 smallest entries are selected again with a new
// T (three passes over the tail) -- every 10-20 rounds instead of two passes per round.  The rounds, their order and their cells
// are those of the plain scan: same integer results, the drains in ascending cell order, the weights in numpy's summation order.
//
// Window 64 x 64 around the pit, 16 + 192 border cells, 16 drains, at most 16 cells at the minimum; what outgrows any of them
// goes on to the wavefront version, which starts again from the lane version's hand-over record.  The arithmetic of the drains
// (filters, slopes, weights) is not done here: k_pits_row_finish, a lane per pit.
#ifndef PYDEM_RW_W
#define PYDEM_RW_W 64
#endif
#ifndef PYDEM_RW_CAP
#define PYDEM_RW_CAP 208
#endif
#ifndef PYDEM_RW_NT
#define PYDEM_RW_NT 64
#endif
#ifndef PYDEM_RW_OCC
#define PYDEM_RW_OCC 4
#endif
#ifndef PYDEM_RW_TARGET
#define PYDEM_RW_TARGET 12
#endif
constexpr int RW_W = PYDEM_RW_W, RW_CAP = PYDEM_RW_CAP, RW_NT = PYDEM_RW_NT, RW_OCC = PYDEM_RW_OCC;
constexpr int RW_H = 16;          // head slots = lanes of a row
constexpr int RW_D = 16;          // drains (lane t of the row owns drain t)
constexpr int RW_CHUNK = 16;      // pits per global atomic on the work counter
constexpr int RW_OUT_CHUNK = 128; // output slots per global atomic (a trip needs at most 4 x RW_D)
constexpr uint16_t RW_HOLE = 0xFFFFu, RW_PITBIT = 0x4000u;
static_assert(RW_W * RW_W <= 0x4000 && (RW_W & (RW_W - 1)) == 0 && RW_W >= 32, "window positions are 14-bit, rows are whole words");
static_assert(RW_W > LN_W16 + 2, "the lane version's window and its rim lie inside this one");

template <int RCAP>                       // slots: RW_H of the head, then the tail
struct RowLds {
    double le[RCAP];                      // border elevations (+inf: free slot)
    uint32_t seen[RW_W * RW_W / 32];      // region | border; after the rounds: drain scratch
    uint16_t lpos[RCAP];                  // window position | RW_PITBIT; RW_HOLE: free slot
    uint16_t pq[RW_H];                    // cells promoted in this round
    uint8_t holes[RCAP];                  // free tail slots below the list end
    uint8_t hholes[RW_H];                 // free head slots
};
static_assert(sizeof(uint32_t) * (RW_W * RW_W / 32) >= RW_D * (4 + 8 + 8), "drain scratch fits the bitmap");

__device__ __forceinline__ uint32_t row_ballot(bool p, int lane)
{
    return (uint32_t)(__ballot(p) >> (lane & 48)) & 0xFFFFu;
}
__device__ __forceinline__ double row_min(double v)
{
    v = dpp_fmin<0xB1>(v);        // quad_perm [1,0,3,2]
    v = dpp_fmin<0x4E>(v);        // quad_perm [2,3,0,1]
    v = dpp_fmin<0x141>(v);       // row_half_mirror
    v = dpp_fmin<0x140>(v);       // row_mirror: all 16 lanes of the row hold its minimum
    return v;
}
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
__device__ __forceinline__ int row_sum(int v)
{
    v += dpp_i32<0xB1>(v); v += dpp_i32<0x4E>(v); v += dpp_i32<0x141>(v); v += dpp_i32<0x140>(v);
    return v;
}
__device__ __forceinline__ int row_scan_incl(int v)               // row_shr:1,2,4,8 (zeros shifted in)
{
    v += dpp_i32<0x111>(v); v += dpp_i32<0x112>(v); v += dpp_i32<0x114>(v); v += dpp_i32<0x118>(v);
    return v;
}
__device__ __forceinline__ double row_first(double v, int lane)   // the value of the row's lane 0
{
    return __shfl(v, lane & 48);
}

// np_pairwise_leaf for the short slices between a pit and its drain with the loads of every group issued together and indices
// clamped to the slice (a one-load-at-a-time loop waits out the memory latency per element while three other pits stand still)
__device__ __forceinline__ double np_pairwise_leaf_pre(const double *__restrict__ a, int n)
{
    const int last = n > 0 ? n - 1 : 0;
    if (n < 8) {
        double v[7];
#pragma unroll
        for (int k = 0; k < 7; k++) v[k] = a[k < n ? k : last];
        double res = 0.;
#pragma unroll
        for (int k = 0; k < 7; k++) if (k < n) res += v[k];
        return res;
    }
    double r[8];
#pragma unroll
    for (int k = 0; k < 8; k++) r[k] = a[k];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = a[i + k];
#pragma unroll
        for (int k = 0; k < 8; k++) r[k] += v[k];
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    double v[7];
#pragma unroll
    for (int k = 0; k < 7; k++) v[k] = a[i + k < n ? i + k : last];
#pragma unroll
    for (int k = 0; k < 7; k++) if (i + k < n) res += v[k];
    return res;
}

template <int RCAP, int OCC, int NT>
__global__ __launch_bounds__(NT, OCC) void k_pits_row(PitParams P, const int32_t *__restrict__ pits, const int32_t *npits)
{
    static_assert(RCAP <= 256 && RCAP % 16 == 0 && RCAP > RW_H, "free-slot indices are bytes");
    __shared__ RowLds<RCAP> s_l[NT / 16];
    const int lane = threadIdx.x & 63, sub = lane & 15;
    RowLds<RCAP> &L = s_l[threadIdx.x >> 4];
    const uint32_t lt16 = (1u << sub) - 1u;
    const unsigned long long ltrow = (1ull << (lane & 48)) - 1ull;
    const int n = P.n, m = P.m;
    const int32_t np = *npits;
    // the neighbour this lane tests in an expansion step (lanes 0-7: first cell of the pair, 8-15: second)
    const int dd = sub & 7;
    const int ndi = dd < 3 ? -1 : (dd < 5 ? 0 : 1);
    const int ndj = dd < 3 ? dd - 1 : (dd == 3 ? -1 : (dd == 4 ? 1 : dd - 6));
    // state of the row's pit (the same value in its 16 lanes, except `fl`)
    bool running = false;
    int pending = 0;
    int32_t q = 0, pit = 0;
    int ipit = 0, jpit = 0, r0 = 0, c0 = 0;
    double epit = 0.0, epit_border = 0.0;
    double T = -INFINITY;         // every tail entry is > T
    int nb = RW_H, nh = 0, hnh = RW_H, n_alive = 0, it = 0;   // tail end / free tail slots below it / free head slots / live entries
    int over = 0;                 // 2 list capacity or more than 16 cells at the minimum, 3 drain capacity (1: left the window, in fl)
    uint32_t fl = 0;              // what THIS lane saw: 1 a non-pit cell below the threshold entered, 2 a lower pit cell, 4 nodata, 8 window left
    int32_t chunk_next = 0, chunk_end = 0;
    int32_t oc_base = 0, oc_left = 0;     // the wavefront's chunk of output slots
    bool more = true;

    // a new border cell of this lane: the drain tests are fixed thresholds, evaluated once, on entry (:1312-1320)
    auto flag_entry = [&](double e, uint32_t pm) {
        if (pm) { if (e < epit) fl |= 2u; }
        else if (e < epit_border) fl |= 1u;
        if (e != e) fl |= 4u;                                                    // nodata on the border (see the rules below)
    };
    // the unseen neighbours of the nq cells in L.pq join the border, two cells per step
    auto expand = [&](int nq) {
        for (int base = 0; base < nq; base += 2) {
            const int c = base + (sub >> 3);
            bool isnew = false;
            double e = 0.0; uint32_t pm = 0; int npos = 0;
            if (c < nq) {
                const uint32_t pos = L.pq[c];
                const int rr = (int)(pos / RW_W) + ndi, cc = (int)(pos % RW_W) + ndj;
                const int ii = r0 + rr, jj = c0 + cc;
                if (ii >= 0 && ii < n && jj >= 0 && jj < m) {
                    if (rr < 0 || rr >= RW_W || cc < 0 || cc >= RW_W) fl |= 8u;
                    else {
                        npos = rr * RW_W + cc;
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
            const uint32_t rb = row_ballot(isnew, lane);
            if (!rb) continue;
            bool toh = isnew && e <= T;
            uint32_t hb = row_ballot(toh, lane);
            if (__popc(hb) > hnh) {
                // the head cannot take them: they go to the tail, and the threshold drops below the smallest of them
                T = nextafter(row_min(toh ? e : INFINITY), -INFINITY);
                toh = false; hb = 0;
            }
            const uint32_t tb = rb & ~hb;
            const int ch = __popc(hb), ct = __popc(tb);
            const int fresh = ct > nh ? ct - nh : 0;                             // tail slots taken beyond the list end
            if (nb + fresh > RCAP) { over = 2; break; }
            if (isnew) {
                int k;
                if (toh) k = L.hholes[hnh - 1 - __popc(hb & lt16)];
                else { const int rk = __popc(tb & lt16); k = rk < nh ? (int)L.holes[nh - 1 - rk] : nb + (rk - nh); }
                L.le[k] = e; L.lpos[k] = (uint16_t)(npos | (pm ? RW_PITBIT : 0));
                flag_entry(e, pm);
            }
            hnh -= ch; nh -= ct - fresh; nb += fresh; n_alive += ch + ct;
        }
        wave_sync();
    };
    // the head is poured back into the tail, and the ~RW_TARGET smallest entries are selected again (sets T; over = 2 if the
    // entries do not fit or more than 16 of them share the minimum)
    auto refill = [&]() {
        {
            const uint16_t hp = L.lpos[sub];
            const bool live = hp != RW_HOLE;
            const uint32_t rb = row_ballot(live, lane);
            const int cnt = __popc(rb);
            const int fresh = cnt > nh ? cnt - nh : 0;
            if (nb + fresh > RCAP) { over = 2; return; }
            if (live) {
                const int rk = __popc(rb & lt16);
                const int k = rk < nh ? (int)L.holes[nh - 1 - rk] : nb + (rk - nh);
                L.le[k] = L.le[sub]; L.lpos[k] = hp;
                L.le[sub] = INFINITY; L.lpos[sub] = RW_HOLE;
            }
            nh -= cnt - fresh; nb += fresh; hnh = RW_H;
        }
        wave_sync();
        double lo = INFINITY, nhi = INFINITY;                                    // minimum, -maximum of the live entries
        for (int k = RW_H + sub; k < nb; k += 16) {
            const double v = L.le[k];
            lo = min_f64(lo, v);
            nhi = min_f64(nhi, v == INFINITY ? INFINITY : -v);
        }
        lo = row_min(lo);
        if (!(lo < INFINITY)) { over = 2; return; }                              // (nothing finite on the border: not this tier's case)
        const double hi = -row_min(nhi);
        const int alive_t = nb - RW_H - nh;
        double t = alive_t > RW_H ? lo + (hi - lo) * ((double)P.rw_target / (double)alive_t) : hi;
        if (!(t >= lo)) t = lo;
        int c = 0;
        for (;;) {
            int mine = 0;
            for (int k = RW_H + sub; k < nb; k += 16) mine += L.le[k] <= t;
            c = row_sum(mine);
            if (c <= RW_H) break;
            if (t == lo) { over = 2; return; }                                   // more than 16 cells at the minimum
            const double t2 = lo + (t - lo) * 0.5;
            t = t2 < t ? t2 : lo;
        }
        T = t;
        int nm = 0;
        for (int k0 = RW_H; k0 < nb; k0 += 16) {
            const int k = k0 + sub;
            const bool mv = k < nb && L.le[k] <= t;
            const uint32_t rb = row_ballot(mv, lane);
            if (!rb) continue;
            if (mv) {
                const int r = nm + __popc(rb & lt16);
                L.le[r] = L.le[k]; L.lpos[r] = L.lpos[k];
                L.le[k] = INFINITY; L.lpos[k] = RW_HOLE;
                L.holes[nh + r] = (uint8_t)k;
            }
            nm += __popc(rb);
        }
        nh += nm; hnh = RW_H - nm;
        if (sub < hnh) L.hholes[sub] = (uint8_t)(nm + sub);
        wave_sync();
    };

    const bool prof = P.prof != nullptr;
    long long acc_a = 0, acc_b = 0, acc_c = 0, acc_d = 0, trips = 0, busy = 0, grow = 0, refills = 0;
    for (;;) {
        long long tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0;
        if (prof) tk0 = clock64();
        // ---- (a) new pits for idle rows
        const unsigned long long idle = __ballot(!running);
        if (idle && more) {
            if (chunk_next == chunk_end) {
                int32_t b = 0;
                if (lane == 0) b = atomicAdd(P.work_next, RW_CHUNK);
                b = __shfl(b, 0);
                chunk_next = b < np ? b : np;
                chunk_end = b + RW_CHUNK < np ? b + RW_CHUNK : np;
                if (chunk_next >= chunk_end) more = false;
            }
            const int avail = chunk_end - chunk_next, want = __popcll(idle) >> 4;
            const int rank = __popcll(idle & ltrow) >> 4;
            if (!running && rank < avail) {
                q = chunk_next + rank;
                pit = pits[q];
                ipit = pit / m; jpit = pit - ipit * m;
                r0 = ipit - RW_W / 2; c0 = jpit - RW_W / 2;
                if (r0 > n - RW_W) r0 = n - RW_W;
                if (c0 > m - RW_W) c0 = m - RW_W;
                if (r0 < 0) r0 = 0;
                if (c0 < 0) c0 = 0;
                for (int w = sub; w < RW_W * RW_W / 32; w += 16) L.seen[w] = 0;
                L.le[sub] = INFINITY; L.lpos[sub] = RW_HOLE; L.hholes[sub] = (uint8_t)sub;
                epit = P.elev[pit];
                epit_border = epit;
                T = -INFINITY;
                nb = RW_H; nh = 0; hnh = RW_H; n_alive = 0; it = 0; over = 0; fl = 0; pending = 0;
                wave_sync();
                if (P.lane_state) {
                    // the lane version grew this pit for `it` rounds before its 16 x 16 window or its 32-cell list overflowed.  Its
                    // region R comes as a bitmap, one row per lane here: the border is dilate(R) & ~R on the 18 x 18 rim of that
                    // window (word operations + the rows above / below by DPP); the cells are filed in the tail, then their
                    // elevations are loaded 16 at a time (drain tests on entry, like always)
                    const uint32_t *st = P.lane_state + (size_t)q * 12;
                    it = (int)st[0];
                    epit_border = __hiloint2double((int)st[3], (int)st[2]);
                    int lr0 = ipit - LN_W16 / 2, lc0 = jpit - LN_W16 / 2;       // the lane version's window (same clipping)
                    if (lr0 > n - LN_W16) lr0 = n - LN_W16;
                    if (lc0 > m - LN_W16) lc0 = m - LN_W16;
                    if (lr0 < 0) lr0 = 0;
                    if (lc0 < 0) lc0 = 0;
                    const uint32_t R = (st[4 + (sub >> 1)] >> ((sub & 1) * 16)) & 0xFFFFu;   // row `sub` of its bitmap
                    const uint32_t X = R << 1;                                   // bit b = column b - 1 of that window
                    const uint32_t H = (X | (X << 1) | (X >> 1)) & 0x3FFFFu;
                    const uint32_t Hup = (uint32_t)dpp_i32<0x111>((int)H), Hdn = (uint32_t)dpp_i32<0x101>((int)H);   // rows sub - 1, sub + 1
                    // columns of the rim that lie in the tile and in this window
                    uint32_t colok = 0x3FFFFu;
                    {
                        const int jlo = lc0 - 1, wlo = lc0 - 1 - c0;             // tile column / window column of bit 0
                        const int cut_lo = jlo < 0 || wlo < 0 ? 1 : 0;           // (at most the rim column itself is outside)
                        if (cut_lo) colok &= ~1u;
                        int hi_t = m - jlo, hi_w = RW_W - wlo;                   // first bit outside the tile / the window
                        int hi_b = hi_t < hi_w ? hi_t : hi_w;
                        if (hi_b < 18) colok &= (1u << (hi_b < 0 ? 0 : hi_b)) - 1u;
                    }
                    const int wr = lr0 - r0 + sub;                               // window row of this lane's bitmap row
                    uint32_t D = (Hup | H | Hdn) & ~X & colok;
                    if (lr0 + sub >= n) D = 0;
                    // lanes 0 and 15 also own the rim rows above and below
                    uint32_t E = 0; int wre = 0;
                    if (sub == 0 && lr0 - 1 >= 0 && wr - 1 >= 0) { E = H & colok; wre = wr - 1; }
                    if (sub == 15 && lr0 + 16 < n && wr + 1 < RW_W) { E = H & colok; wre = wr + 1; }
                    const int mine = __popc(D) + __popc(E);
                    const int incl = row_scan_incl(mine);
                    const int total = __shfl(incl, (lane & 48) + 15);
                    if (RW_H + total > RCAP) over = 2;
                    else {
                        // the bitmap: region and rim of a row are one or two words
                        const int wc0 = lc0 - 1 - c0;                            // window column of bit 0 (may be -1: that bit is cut)
                        {
                            uint32_t bits = X | D; int col = wc0;
                            if (col < 0) { bits >>= 1; col = 0; }
                            const int p = wr * RW_W + col;
                            const unsigned long long sh = (unsigned long long)bits << (p & 31);
                            if ((uint32_t)sh) atomicOr(&L.seen[p >> 5], (uint32_t)sh);
                            if (sh >> 32) atomicOr(&L.seen[(p >> 5) + 1], (uint32_t)(sh >> 32));
                        }
                        if (E) {
                            uint32_t bits = E; int col = wc0;
                            if (col < 0) { bits >>= 1; col = 0; }
                            const int p = wre * RW_W + col;
                            const unsigned long long sh = (unsigned long long)bits << (p & 31);
                            if ((uint32_t)sh) atomicOr(&L.seen[p >> 5], (uint32_t)sh);
                            if (sh >> 32) atomicOr(&L.seen[(p >> 5) + 1], (uint32_t)(sh >> 32));
                        }
                        int k = RW_H + incl - mine;
                        for (uint32_t b = D; b; b &= b - 1u) L.lpos[k++] = (uint16_t)(wr * RW_W + wc0 + (__ffs((int)b) - 1));
                        for (uint32_t b = E; b; b &= b - 1u) L.lpos[k++] = (uint16_t)(wre * RW_W + wc0 + (__ffs((int)b) - 1));
                        nb = RW_H + total; n_alive = total;
                        wave_sync();
                        for (int k2 = RW_H + sub; k2 < nb; k2 += 16) {
                            const uint32_t pos = L.lpos[k2];
                            const int64_t cell = (int64_t)(r0 + (int)(pos / RW_W)) * m + (c0 + (int)(pos % RW_W));
                            const double e = P.elev[cell];
                            const uint32_t pm = P.pitmask[cell];
                            L.le[k2] = e;
                            if (pm) L.lpos[k2] = (uint16_t)(pos | RW_PITBIT);
                            flag_entry(e, pm);
                        }
                        wave_sync();
                    }
                } else {                                                         // pit_area = [pit] (:1289-1292)
                    const int pos = (ipit - r0) * RW_W + (jpit - c0);
                    if (sub == 0) { L.seen[pos >> 5] = 1u << (pos & 31); L.pq[0] = (uint16_t)pos; }
                    wave_sync();
                    expand(1);
                    if (P.min_border) {                                          // :1294-1295
                        double mn = INFINITY;
                        for (int k = RW_H + sub; k < nb; k += 16) mn = min_f64(mn, L.le[k]);
                        mn = row_min(mn);
                        if (n_alive) epit_border = mn;
                        fl &= ~1u;                                               // nothing is below the minimum
                    }
                }
                running = true;
            }
            chunk_next += want < avail ? want : avail;
        }
        if (!__ballot(running)) { if (more) continue; break; }
        if (prof) { tk1 = clock64(); acc_a += tk1 - tk0; trips++; busy += __popcll(__ballot(running)) >> 4; }

        // ---- (b) one round of every growing row
        if (running && !pending) {
            uint32_t f_np = 0, f_p = 0, f_nan = 0;
            if (row_ballot(fl != 0, lane)) {
                f_np = row_ballot((fl & 1u) != 0, lane); f_p = row_ballot((fl & 2u) != 0, lane); f_nan = row_ballot((fl & 4u) != 0, lane);
                if (row_ballot((fl & 8u) != 0, lane) && !over) over = 1;
            }
            // :1300, :1304-1305 no drain; numpy's min propagates NaN: with a nodata cell on the border there is no non-pit drain
            // and no growth, only a lower pit cell can still drain it; :1312-1316 non-pit drains; :1317-1320 pit drains
            // (one chain of selects: as nested branches every level copies the row's state)
            const int by_flags = f_nan ? (f_p ? 2 : 3) : (f_np ? 1 : (f_p ? 2 : 0));
            pending = over ? 4 : ((it >= P.max_iter || n_alive == 0) ? 3 : by_flags);
        }
        {
            if (running && !pending) {
                // the minimum of the head is the minimum of the border while it is <= T (free slots hold +inf)
                double he = L.le[sub];
                double mn = row_min(he);
                if (__ballot(!(mn <= T) || hnh == RW_H)) {                       // (any row: the others refill with it at no cost)
                    if (prof) refills++;
                    refill();
                    he = L.le[sub];
                    mn = row_min(he);
                }
                if (!over) {
                    // pit_area += border[eborder == emin] (:1322-1323): out of the head, slots recycled
                    bool match = false;
                    uint16_t ps = 0;
                    if (he == mn) { ps = L.lpos[sub]; match = ps != RW_HOLE; }
                    const uint32_t rb = row_ballot(match, lane);
                    const int nq = __popc(rb);
                    if (match) {
                        const int r = __popc(rb & lt16);
                        L.pq[r] = (uint16_t)(ps & (RW_PITBIT - 1));
                        L.hholes[hnh + r] = (uint8_t)sub;
                        L.le[sub] = INFINITY; L.lpos[sub] = RW_HOLE;
                    }
                    hnh += nq; n_alive -= nq;
                    wave_sync();
                    expand(nq);
                    it++;
                }
            }
        }

        if (prof) { tk2 = clock64(); acc_b += tk2 - tk1; grow += __popcll(__ballot(running && !pending)) >> 4; }
        // ---- (c) rows whose growth has ended: the drains, in ascending cell order.  Filters, slopes and weights are NOT computed
        // here (a divergent section of one row stalls the three other pits of the wavefront: square roots, divisions, chains of
        // loads from dX / dY): the cells go to the output slots as they are and k_pits_row_finish does the arithmetic a lane per pit
        int status = 0;             // 1: drains selected, 2: no drain, 3: hand over to the wavefront version
        int nd = 0;
        int32_t cell = 0;
        if (pending) {
            const int mode = pending <= 2 ? pending : 0;
            status = pending == 4 ? 3 : 2;
            int32_t *dl = (int32_t *)L.seen;                                     // (the bitmap is dead)
            if (mode) {
                // ballot-compacted, then rank-sorted (the order of setdiff1d)
                for (int k0 = 0; k0 < nb; k0 += 16) {
                    const int k = k0 + sub;
                    bool pred = false;
                    int32_t cl = 0;
                    if (k < nb) {
                        const uint16_t ps = L.lpos[k];
                        if (ps != RW_HOLE) {
                            const uint32_t pos = ps & (RW_PITBIT - 1);
                            const bool pm = (ps & RW_PITBIT) != 0;
                            const double ev = L.le[k];
                            pred = mode == 1 ? (!pm && ev < epit_border) : (pm && ev < epit);
                            cl = (int32_t)((int64_t)(r0 + (int)(pos / RW_W)) * m + (c0 + (int)(pos % RW_W)));
                        }
                    }
                    const uint32_t rb = row_ballot(pred, lane);
                    const int rank = nd + __popc(rb & lt16);
                    if (pred && rank < RW_D) dl[rank] = cl;
                    nd += __popc(rb);
                }
                wave_sync();
                if (nd > RW_D) { status = 3; over = 3; nd = 0; }
                else if (nd > 0) {
                    const int32_t key = sub < nd ? dl[sub] : INT32_MAX;
                    int rank = 0;
                    for (int t = 0; t < nd; t++) rank += dl[t] < key;
                    wave_sync();
                    if (sub < nd) dl[rank] = key;
                    wave_sync();
                    cell = sub < nd ? dl[sub] : 0;
                    status = 1;
                }
            }
            if (P.dbg && sub == 0) {                                             // statistics (debug only)
                const int idx = atomicAdd(&P.out_count[6], 1);
                P.dbg[4 * idx] = it; P.dbg[4 * idx + 1] = n_alive; P.dbg[4 * idx + 2] = over; P.dbg[4 * idx + 3] = status == 1 ? nd : -1;
            }
            running = false; pending = 0;
        }

        if (prof) { tk3 = clock64(); acc_c += tk3 - tk2; }
        // ---- (d) all rows together: output slots from the wavefront's chunk (one returning atomic per RW_OUT_CHUNK slots; the
        // slots a chunk leaves unused keep src = -1 and are dropped later), one record per pit for the finishing kernel
        if (!__ballot(status != 0)) continue;
        const int ndo = status == 1 ? nd : 0;
        const int nd0 = __builtin_amdgcn_readlane(ndo, 0), nd1 = __builtin_amdgcn_readlane(ndo, 16),
                  nd2 = __builtin_amdgcn_readlane(ndo, 32), nd3 = __builtin_amdgcn_readlane(ndo, 48);
        const int tot = nd0 + nd1 + nd2 + nd3;
        int32_t o = 0;
        bool fits = false;
        if (tot) {
            if (tot > oc_left) {
                int32_t b = 0;
                if (lane == 0) b = atomicAdd(&P.out_count[0], RW_OUT_CHUNK);
                oc_base = __builtin_amdgcn_readfirstlane(b); oc_left = RW_OUT_CHUNK;
            }
            fits = (int64_t)oc_base + tot <= P.out_cap;
            if (fits) {
                const int row = lane >> 4;
                o = oc_base + (row > 0 ? nd0 : 0) + (row > 1 ? nd1 : 0) + (row > 2 ? nd2 : 0);
                if (status == 1 && sub < nd) { P.out_src[o + sub] = pit; P.out_dst[o + sub] = cell; }
            } else if (lane == 0) atomicAdd(&P.out_count[3], 1);
            oc_base += tot; oc_left -= tot;
        }
        if (status && sub == 0) P.row_rec[q] = make_int2(o, status == 1 && fits ? nd : 0);
        const unsigned long long b_un = __ballot(status == 2 && sub == 0);
        if (lane == 0 && b_un) atomicAdd(&P.out_count[1], __popcll(b_un));       // :1327-1329
        if (status == 3 && sub == 0) P.row_overflow[atomicAdd(P.row_overflow_count, 1)] = q;   // (entry of the hand-over list)
        if (prof) acc_d += clock64() - tk3;
    }
    if (prof && lane == 0) {    // cycles per phase, summed over wavefronts (PYDEM_PITS_DEBUG=3)
        atomicAdd(P.prof + 8, (unsigned long long)acc_a); atomicAdd(P.prof + 9, (unsigned long long)acc_b);
        atomicAdd(P.prof + 10, (unsigned long long)acc_c); atomicAdd(P.prof + 11, (unsigned long long)acc_d);
        atomicAdd(P.prof + 12, (unsigned long long)trips); atomicAdd(P.prof + 13, (unsigned long long)busy);
        atomicAdd(P.prof + 14, (unsigned long long)grow); atomicAdd(P.prof + 15, (unsigned long long)refills);
    }
}

// Filters, slopes and weights of the pits the row version drained (reference :1335-1371; the arithmetic of finish_pit, in place
// in the output slots): a lane per pit.  rec[q] = (first slot, number of candidate drains in ascending cell order; 0: nothing to do).
__global__ __launch_bounds__(256) void k_pits_row_finish(PitParams P, const int32_t *__restrict__ pits, const int32_t *npits, const int2 *__restrict__ rec)
{
    const int n = P.n, m = P.m;
    const int32_t np = *npits;
    const bool xy = !isnan(P.max_dist_XY) && P.max_dist_XY != 0;
    for (int32_t q = blockIdx.x * 256 + threadIdx.x; q < np; q += gridDim.x * 256) {
        const int2 r = rec[q];
        const int nd = r.y, o = r.x;
        if (nd == 0) continue;
        const int32_t pit = pits[q];
        const int ipit = pit / m, jpit = pit - ipit * m;
        const double epit = P.elev[pit];
        const int ndX = n - 1;
        int keep = 0;
        for (int t = 0; t < nd; t++) {
            const int32_t cell = P.out_dst[o + t];
            const int idr = cell / m, jdr = cell - idr * m;
            if (P.max_dist) {                                                    // :1335-1343
                const int di = ipit - idr, dj = jpit - jdr;
                if (!(sqrt((double)(di * di + dj * dj)) <= (double)P.max_dist)) continue;
            }
            const int a = ipit < idr ? ipit : idr, b = ipit < idr ? idr : ipit;
            double dxm;
            if (ipit == idr) dxm = P.dX[ipit < ndX - 1 ? ipit : ndX - 1];        // _get_dX_mean :1994-1995
            else dxm = np_pairwise_leaf_pre(P.dX + a, b - a) / (double)(b - a);  // .mean() :1997 (b - a < RW_W: one leaf)
            const double dx = dxm * (double)(jpit - jdr);
            const double dy = np_pairwise_leaf_pre(P.dY + a, b - a);
            const double d = sqrt(dx * dx + dy * dy);
            if (xy && !(d <= P.max_dist_XY)) continue;                           // :1352-1358
            P.out_dst[o + keep] = cell;
            P.out_w[o + keep] = pit_drop(P, epit, P.elev[cell]) / d;             // :1361
            keep++;
        }
        for (int t = keep; t < nd; t++) P.out_src[o + t] = -1;                   // (unused slot)
        if (keep == 0) { atomicAdd(&P.out_count[1], 1); continue; }              // :1327-1329
        const double ssum = np_pairwise_leaf(P.out_w + o, keep);
        for (int t = 0; t < keep; t++) P.out_w[o + t] = P.out_w[o + t] / ssum;   // :1365-1367
        P.mag[pit] = ssum / (double)keep;                                        // np.mean(s) :1370
        P.flats[pit] = 0;                                                        // :1371
    }
}
