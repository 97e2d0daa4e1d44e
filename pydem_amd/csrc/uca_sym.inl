// uca_sym.inl -- K5f: two-level (tile -> perimeter) solve for the tail of the UCA sweep.  Included by uca.hip inside its
// anonymous namespace, after sweep_one_tile().
//
// Replaces, like the tile passes it shortens: _calc_uca_chunk (pydem/dem_processing.py:864-987) around the native loop
// cyutils._drain_area (pydem/cyfuncs/cyutils.pyx:119-187).
//
// The tile passes advance a flow path by one TILE per pass: a visit can only finish what its neighbours released in an
// earlier pass, so after the two full passes 15 % of the cells (the plumes below every tile crossing) take 80 more
// passes in which the same tiles are staged again and again (profiles/r04_sweep_passes_dense0.txt: 18.4 ms of 30.9).
// The area of a cell is LINEAR in the areas of the cells upstream of it, so ONE visit per tile can do all of the tile's
// work at once if the open cells of the halo -- the tile's INLETS -- are carried as symbols:
//   (a) pass 3, k_sweep_sym: every open cell of the tile is finished as  K + sum_j coef_j * area(inlet j)  (K: the cell
//       area + everything that arrives from finished cells).  A cell that no inlet reaches is finished for good (this is
//       what an ordinary pass 3 would do); the others keep a header (K, where their coefficients are) in their -- still
//       unused -- contribution slot, the level stamp CI_LEVEL_SYM, and the coefficients in a pool.  The cells whose flow
//       leaves the tile (OUTLETS) are listed in the tile's block.
//   (b) passes 4..: a listed tile that went symbolic gets a LIGHT visit (sym_light_visit, inside the listed / resident
//       kernels): the inlets that are final by now are loaded, every outlet all of whose inlets are final is evaluated,
//       written and stamped like a finished cell, and the tiles it drains into are listed -- the same protocol as the
//       numeric visits (level stamps < pass count as final), so tiles that stayed numeric (too many open cells, inlets or
//       outlets, pool exhausted) take part with their ordinary visits.  The depth in passes is unchanged; a pass costs
//       two dependent loads per tile instead of a staging, a set-up and a dozen rounds.
//   (c) k_sym_finish, after the last pass: every cell that still carries CI_LEVEL_SYM is evaluated from its header in one
//       flat launch (cells below a drainage loop go back to "unfinished" and take the re-seed replay K5c like before).
// Sums are re-associated (K + sum coef * x instead of the cascade): uca within ~1e-13 relative of the tile passes; the
// `edge_todo` taint is an OR over the same dependency sets and stays exact.
// Model and sizes: tools/sim_two_level.py, profiles/r05_sim_two_level_*.txt.

constexpr uint32_t CI_LEVEL_SYM = CI_LEVEL_INF - 1u;       // open; the contribution slot holds the symbolic header
constexpr uint32_t SYM_NONE = 0xFFFFFFFFu;
constexpr int SYM_CHUNK = 512;                             // doubles a wavefront takes from its pool region at a time
constexpr int SYM_MAXIN = 64;                              // inlets per tile (bits of the dependency mask)
constexpr int NHALO = 2 * (TT + 2) + 2 * TH;               // cells of the halo ring

struct SymArgs {
    double *pool0, *pool1;       // two buffers (the idle queue buffers of the tile)
    uint32_t reg_cap;            // doubles per region
    int nreg;                    // regions (power of two; region r lies in buffer r & 1 at (r >> 1) * reg_cap)
    int32_t *ctr;                // [nreg] bump counters
    uint32_t *tile_sym;          // per tile: pool offset of its block, SYM_NONE = the tile stays numeric
    int32_t *stat;               // [0] symbolic tiles, [1] tiles that fell back, [2] outlets, [3] symbolic cells
};

// a pool offset: bit 31 = buffer, bits 0-30 = index in doubles
__device__ __forceinline__ double *sym_ptr(const SymArgs &Y, uint32_t off)
{
    return ((off >> 31) ? Y.pool1 : Y.pool0) + (off & 0x7FFFFFFFu);
}

struct SymAlloc { uint32_t cur, end; int reg; };           // wavefront-uniform

// wavefront-collective bump allocation: every lane asks for `need` doubles (0: nothing) and gets its own contiguous
// block; SYM_NONE for ALL lanes when the pool is exhausted
__device__ __forceinline__ uint32_t sym_alloc(const SymArgs &Y, SymAlloc &AL, uint32_t need, int lane, uint32_t &incl, uint32_t &total)
{
    incl = need;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d); if (lane >= d) incl += v; }
    total = __builtin_amdgcn_readfirstlane(__shfl(incl, 63));
    if (total == 0) return 0u;
    if (AL.cur + total > AL.end) {
        const uint32_t grab = total > (uint32_t)SYM_CHUNK ? total : (uint32_t)SYM_CHUNK;
        uint32_t got = SYM_NONE;
        for (int tr = 0; tr < Y.nreg && got == SYM_NONE; tr++) {
            const int r = (AL.reg + tr) & (Y.nreg - 1);
            uint32_t b = 0;
            if (lane == 0) {
                b = (uint32_t)atomicAdd(&Y.ctr[r], (int32_t)grab);
                if (b + grab > Y.reg_cap) atomicSub(&Y.ctr[r], (int32_t)grab);        // (keeps a full region's counter bounded)
            }
            b = __builtin_amdgcn_readfirstlane(__shfl(b, 0));
            if (b + grab <= Y.reg_cap) { got = ((uint32_t)(r & 1) << 31) | ((uint32_t)(r >> 1) * Y.reg_cap + b); AL.reg = r; }
        }
        if (got == SYM_NONE) return SYM_NONE;
        AL.cur = got; AL.end = got + grab;
    }
    const uint32_t my = AL.cur + incl - need;
    AL.cur += total;
    return my;
}

// slot word of the symbolic visit: bits 0-9 cell, 10-17 in-edges from cells that were open at set-up (cells of the tile
// or inlets), 18 taint, 19 finished; bits 20-27: in-edges from final cells while the constant parts are gathered,
// afterwards bits 20-28 = the neighbour tiles the cell drains into, bit 29 = it has a pit edge that goes further
constexpr uint32_t SS_CELL = 0x3FFu, SS_TAINT = 1u << 18, SS_FIN = 1u << 19, SS_FAR = 1u << 29;
constexpr int SS_OPEN_SHIFT = 10, SS_FINAL_SHIFT = 20, SS_WAKE_SHIFT = 20;

template <int RC>
struct TileS {
    TileW W;                          // staging, final bitmap, count-downs as in the generic visit (W.list is not used)
    uint16_t map[TH * TT];            // cell -> slot
    uint16_t list[RC];                // ready slots in the order they became ready (each enters once)
    double Kd[RC], Pd[RC];            // constant part (the area, for a cell no inlet reaches); proportion
    unsigned long long mk[RC];        // inlets the cell depends on
    uint32_t off[RC], sm[RC];         // pool offset of (mask, coefficients); slot word
    uint32_t hw[NHALO];               // graph words of the halo ring
    int32_t in_id[SYM_MAXIN];         // inlets: cell ids (halo cells first, then pit sources further away)
    double in_p[SYM_MAXIN];           // proportion of the halo inlets
    uint8_t hidx[NHALO];              // ring position -> inlet index (0xFF: none)
    uint16_t rt_incl[64], rt_slot[64];    // the round's table: per ready cell the inclusive prefix of (1 + entries) and its slot
    // pit edges of the open cells, read once at set-up (a walk of the global lists per round -- offsets, destinations, sources,
    // their level stamps: three or four dependent trips -- was half of a round's 6.7 us):
    static constexpr int PE = RC / 2, PO = RC / 2, LP = RC >= 1024 ? RC : 3 * RC;
    uint16_t pe_code[PE];             // in-edges from sources that were open at set-up: slot of the source, or 0x8000 | inlet
    double pe_w[PE];                  // their weights
    uint16_t po_code[PO];             // out-edges: cell of this tile (< 1024), 1024 + neighbour tile (0..8), 0x8000 | index into far_tile
    uint16_t pe_first[RC], po_first[RC];
    uint8_t pe_n[RC], po_n[RC];
    int32_t far_tile[32];
    double lpool[LP];                 // the tile's first coefficients stay on chip until the visit ends (offsets with SYM_LOCAL); what
                                      // does not fit goes straight to the global pool
    int tail, nin, fail, npe, npo, nfar;
};
constexpr uint32_t SYM_LOCAL = 1u << 30;                   // offset into TileS::lpool (only while the visit runs)

// ring position of a halo cell (halo coordinates: rows 0..HH-1, columns 0..TT+1)
__device__ __forceinline__ int halo_pos(int hi, int hj)
{
    if (hi == 0) return hj;
    if (hi == HH - 1) return (TT + 2) + hj;
    return 2 * (TT + 2) + (hj == 0 ? 0 : TH) + hi - 1;
}

// in-edge directions (bits of the in-mask) whose neighbour lies outside the tile, for the cell in row r / column c
__device__ __forceinline__ uint32_t outside_dirs(int r, int c)
{
    return (r == 0 ? 0x07u : 0u) | (r == TH - 1 ? 0xE0u : 0u) | (c == 0 ? 0x29u : 0u) | (c == TT - 1 ? 0x94u : 0u);
}

// offset of the neighbour of in-edge d (0..7 = NW N NE W E SW S SE)
__device__ __forceinline__ void nb_delta(int d, int &di, int &dj)
{
    const int q = d + (d >> 2);
    di = ((q * 11) >> 5) - 1;
    dj = q - 3 * (di + 1) - 1;
}

// tile-local id of the neighbour in-edge d comes from
__device__ __forceinline__ int sym_nb_local(int cell, int d)
{
    int di, dj;
    nb_delta(d, di, dj);
    return cell + di * TT + dj;
}

// ONE symbolic visit (see the head of the file).  Returns false when the tile does not fit (inlets, slots, outlets, pool):
// nothing but unused pool space has been written then, and the caller runs the numeric visit of this pass instead.
template <int RC>
__device__ bool sym_visit(const SweepArgs &A, const SymArgs &Y, SymAlloc &AL, TileS<RC> &S, uint32_t pass, int tiles_x, int tid, int lane,
                          uint8_t *__restrict__ tile_done, int32_t &n_final, const TileNext &N, int32_t *pend, int &npend)
{
    if (TH != 32) return false;        // (the slot word holds 10 bits of cell id; stage_sweep only turns the two-level solve on for 32-row tiles)
    TileW &L = S.W;
    const int by = tid / tiles_x, bx = tid - by * tiles_x;
    const int i0 = by * TH, j0 = bx * TT, n = A.n, m = A.m;
    const int half = lane >> 5, l32 = lane & 31;
    constexpr int NSET = TH * TT / 64;
    if (lane == 0) { S.tail = 0; S.nin = 0; S.fail = 0; S.npe = 0; S.npo = 0; S.nfar = 0; }
#ifdef PYDEM_SYM_PROF           // phase timers of the symbolic visit (10 ns ticks summed over the visits into Y.stat[8..]; a diagnostic build)
    long long tk[8]; int nrounds = 0, nentries = 0;
    tk[0] = wall_clock64();
#define SYM_TICK(i) tk[i] = wall_clock64()
#else
#define SYM_TICK(i)
#endif
    const double a0_row = (lane < TH && i0 + lane < n) ? A.a0[i0 + lane] : 0.0;
    tile_stage(A, tile_base(A, i0, j0), L, pass, i0, j0, lane, S.hw);
    tile_wave_sync();
    SYM_TICK(1);
    // ---- inlets: open cells of the halo ring with an edge into the tile
    int nin = 0;
    for (int q0 = 0; q0 < NHALO; q0 += 64) {
        const int q = q0 + lane;
        bool is_in = false;
        int hi = 0, hj = 0;
        if (q < NHALO) {
            if (q < TT + 2) { hi = 0; hj = q; }
            else if (q < 2 * (TT + 2)) { hi = HH - 1; hj = q - (TT + 2); }
            else if (q < 2 * (TT + 2) + TH) { hi = q - 2 * (TT + 2) + 1; hj = 0; }
            else { hi = q - 2 * (TT + 2) - TH + 1; hj = TT + 1; }
            const uint32_t w = S.hw[q];
            const uint32_t lv = ci_level(w);
            if (w != 0xFFFFFFFFu && !(lv >= 1 && lv < pass)) {
                const int sct = ci_section(w);
                if (w & CI_OUT1) { const int ti = hi + fe1r(sct), tj = hj + fe1c(sct); is_in = is_in || (ti >= 1 && ti <= TH && tj >= 1 && tj <= TT); }
                if (w & CI_OUT2) { const int ti = hi + fe2r(sct), tj = hj + fe2c(sct); is_in = is_in || (ti >= 1 && ti <= TH && tj >= 1 && tj <= TT); }
            }
        }
        const unsigned long long b = __ballot(is_in);
        const int idx = nin + __popcll(b & ((1ull << lane) - 1ull));
        if (q < NHALO) S.hidx[q] = (uint8_t)(is_in ? idx : 0xFF);
        if (is_in && idx < SYM_MAXIN) S.in_id[idx] = (i0 + hi - 1) * m + j0 + hj - 1;
        nin += __popcll(b);
    }
    if (nin > SYM_MAXIN) return false;
    if (lane == 0) S.nin = nin;
    tile_wave_sync();
    if (lane < nin) S.in_p[lane] = A.prop[S.in_id[lane]];
    SYM_TICK(2);
    // ---- set-up: a slot per open cell, count = open upstream cells INSIDE the tile (an open inlet does not block)
    uint32_t pitmask = 0, poutmask = 0;
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
            const uint32_t cnt = __popc(im & ~nf & ~outside_dirs(li - 1, l32));
            sp_of(L, idx) = (uint16_t)cnt;
            const int s = nslot + __popcll(bo & ((1ull << lane) - 1ull));
            if (s < RC) {
                S.map[idx] = (uint16_t)s;
                S.sm[s] = (uint32_t)idx | ((im & ~nf) << SS_OPEN_SHIFT) | ((im & nf) << SS_FINAL_SHIFT);
                if (w & (CI_PIT_OUT << 16)) poutmask |= 1u << k;
                if (w & (CI_PIT_IN << 16)) pitmask |= 1u << k;               // (joins the ready list after its pit edges have been read)
                else if (cnt == 0) S.list[atomicAdd(&S.tail, 1)] = (uint16_t)s;
            }
        }
        nslot += __popcll(bo);
    }
    if (nslot > RC) return false;
    tile_wave_sync();
    if (lane < TH) L.a0[lane] = a0_row;                  // (the final bitmap is dead now)
    tile_wave_sync();
    SYM_TICK(3);
    // ---- the constant part of every open cell: all its loads in flight together
    for (int s = lane; s < nslot; s += 64) {
        const uint32_t smv = S.sm[s];
        const int cell = smv & SS_CELL;
        const int gi = i0 + (cell >> 5), gj = j0 + (cell & 31);
        const int32_t c = gi * m + gj;
        const uint32_t cw = L.cs[cell] >> 16;
        double pv = 0.0;
        if (cw & (CI_OUT1 | CI_OUT2)) pv = A.prop[c];
        bool td = (gi == 0 || gi == n - 1 || gj == 0 || gj == m - 1) && A.todo_work[c] != 0;
        uint32_t mm = (smv >> SS_FINAL_SHIFT) & 0xFFu;
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
        S.Kd[s] = K; S.Pd[s] = pv; S.mk[s] = 0ull; S.off[s] = 0u;
        S.sm[s] = (smv & (SS_CELL | (0xFFu << SS_OPEN_SHIFT))) | (td ? SS_TAINT : 0u);
        S.pe_n[s] = 0; S.po_n[s] = 0;
    }
    tile_wave_sync();
    // ---- pit edges of the open cells.  In-edges of the lane's drains: a source that is final adds to the constant part now; a
    // source that is open in this tile is released on chip (entry: its slot); an open source elsewhere is an inlet
    if (pitmask) {
#pragma unroll 1
        for (int k = 0; k < NSET; k++) {
            if (!(pitmask & (1u << k))) continue;
            const int cell = lane + 64 * k;
            const int s = S.map[cell];
            const int32_t c = (i0 + (cell >> 5)) * m + j0 + (cell & 31);
            const int32_t e0 = pit_stash(A, c).x;
            uint32_t cnt = sp_of(L, cell);
            int ne = 0;
            double kadd = 0.0;
            bool td = false;
            for (int32_t e = e0; e < A.n_pit && A.pin_dst[e] == c; e++) {
                const int32_t sc = A.pin_src[e];
                const int si = sc / m - i0, sj = sc % m - j0;
                const bool inside = si >= 0 && si < TH && sj >= 0 && sj < TT;
                bool open_src;
                if (inside) { open_src = sp_state(L, si * TT + sj) == 0u; if (open_src) cnt++; }
                else { const uint32_t lv = ci_level(A.cinfo[sc]); open_src = !(lv >= 1 && lv < pass); }
                if (open_src) ne++;
                else { kadd += A.area[sc] * A.pin_w[e]; td = td || (A.todo_work[sc] != 0); }
            }
            int base = 0;
            if (ne) {
                base = atomicAdd(&S.npe, ne);
                if (base + ne > TileS<RC>::PE || ne > 255) { S.fail = 1; ne = 0; }
            }
            int q = 0;
            if (ne)
                for (int32_t e = e0; e < A.n_pit && A.pin_dst[e] == c; e++) {
                    const int32_t sc = A.pin_src[e];
                    const int si = sc / m - i0, sj = sc % m - j0;
                    uint32_t code;
                    if (si >= 0 && si < TH && sj >= 0 && sj < TT) {
                        if (sp_state(L, si * TT + sj) != 0u) continue;
                        code = S.map[si * TT + sj];
                    } else {
                        const uint32_t lv = ci_level(A.cinfo[sc]);
                        if (lv >= 1 && lv < pass) continue;
                        const int j = atomicAdd(&S.nin, 1);
                        if (j < SYM_MAXIN) S.in_id[j] = sc; else S.fail = 1;
                        code = 0x8000u | (uint32_t)(j & 63);
                    }
                    S.pe_code[base + q] = (uint16_t)code; S.pe_w[base + q] = A.pin_w[e]; q++;
                }
            S.pe_first[s] = (uint16_t)base; S.pe_n[s] = (uint8_t)ne;
            S.Kd[s] += kadd;
            if (td) S.sm[s] |= SS_TAINT;
            sp_of(L, cell) = (uint16_t)cnt;
            if (cnt == 0) S.list[atomicAdd(&S.tail, 1)] = (uint16_t)s;
        }
    }
    // out-edges of the lane's open pits: where they drain to
    if (poutmask) {
#pragma unroll 1
        for (int k = 0; k < NSET; k++) {
            if (!(poutmask & (1u << k))) continue;
            const int cell = lane + 64 * k;
            const int s = S.map[cell];
            const int32_t c = (i0 + (cell >> 5)) * m + j0 + (cell & 31);
            const int32_t e0 = pit_stash(A, c).y;
            int ne = 0;
            for (int32_t e = e0; e < A.n_pit && A.pit_src[e] == c; e++) ne++;
            int base = 0;
            if (ne) {
                base = atomicAdd(&S.npo, ne);
                if (base + ne > TileS<RC>::PO || ne > 255) { S.fail = 1; ne = 0; }
            }
            for (int q = 0; q < ne; q++) {
                const int32_t dc = A.pit_dst[e0 + q];
                const int ti = dc / m - i0, tj = dc % m - j0;
                uint32_t code;
                if (ti >= 0 && ti < TH && tj >= 0 && tj < TT) code = (uint32_t)(ti * TT + tj);
                else if (ti >= -TH && ti < 2 * TH && tj >= -TT && tj < 2 * TT) {
                    const int dti = ti < 0 ? -1 : (ti >= TH ? 1 : 0), dtj = tj < 0 ? -1 : (tj >= TT ? 1 : 0);
                    code = 1024u + (uint32_t)((dti + 1) * 3 + dtj + 1);
                } else {
                    const int f = atomicAdd(&S.nfar, 1);
                    if (f < 32) S.far_tile[f] = (dc / m / TH) * tiles_x + (dc % m) / TT; else S.fail = 1;
                    code = 0x8000u | (uint32_t)(f & 31);
                }
                S.po_code[base + q] = (uint16_t)code;
            }
            S.po_first[s] = (uint16_t)base; S.po_n[s] = (uint8_t)ne;
        }
    }
    tile_wave_sync();
    if (S.fail) return false;
    const int nin_all = S.nin < SYM_MAXIN ? S.nin : SYM_MAXIN;
    SYM_TICK(4);
    // ---- rounds: counts and slots in LDS, coefficients in the pool (written in one round, read by this wavefront in a later one)
    uint32_t wake = 0;
    int head = 0;
    bool exhausted = false, any_global = false;
    uint32_t lused = 0;                                    // doubles of the LDS pool in use (wavefront-uniform)
    auto coef_at = [&](uint32_t o, int k) -> double {      // entry k of the block at offset o
        return (o & SYM_LOCAL) ? S.lpool[(o & (SYM_LOCAL - 1u)) + (uint32_t)k] : sym_ptr(Y, o)[k];
    };
    for (;;) {
        const int tail = S.tail;
        if (head >= tail) break;
        const int idx = head + lane;
        const bool act = idx < tail;
        int s = 0, cell = 0;
        uint32_t smv = 0, cw = 0, opn = 0, om = 0;
        double a = 0.0;
        bool td = false;
        unsigned long long mask = 0ull;
        if (act) {
            s = S.list[idx]; smv = S.sm[s]; cell = smv & SS_CELL;
            cw = L.cs[cell] >> 16;
            a = S.Kd[s]; td = (smv & SS_TAINT) != 0u;
            const int r = cell >> 5, cc = cell & 31;
            om = outside_dirs(r, cc);
            opn = (smv >> SS_OPEN_SHIFT) & 0xFFu;
            uint32_t mm = opn;
            while (mm) {
                const int d = __ffs(mm) - 1; mm &= mm - 1u;
                if ((om >> d) & 1u) {
                    int di, dj; nb_delta(d, di, dj);
                    mask |= 1ull << S.hidx[halo_pos(r + 1 + di, cc + 1 + dj)];
                } else {
                    const int ss = S.map[sym_nb_local(cell, d)];
                    const double as = S.Kd[ss], ps = S.Pd[ss];
                    a += ((0x5A >> d) & 1) ? as * ps : as * (1 - ps);
                    td = td || (S.sm[ss] & SS_TAINT);
                    mask |= S.mk[ss];
                }
            }
            for (int q = 0, pf = S.pe_first[s], pn = S.pe_n[s]; q < pn; q++) {          // pit in-edges from sources open at set-up
                const uint32_t code = S.pe_code[pf + q];
                if (code & 0x8000u) mask |= 1ull << (code & 63u);
                else { a += S.Kd[code] * S.pe_w[pf + q]; td = td || (S.sm[code] & SS_TAINT); mask |= S.mk[code]; }
            }
        }
        const int nent = __popcll(mask);
        uint32_t incl = 0, total = 0, my;
        {
            const uint32_t need = (act && nent) ? (uint32_t)nent + 1u : 0u;
            uint32_t inc2 = need;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(inc2, d); if (lane >= d) inc2 += v; }
            const uint32_t tot2 = __builtin_amdgcn_readfirstlane(__shfl(inc2, 63));
            if (lused + tot2 <= (uint32_t)TileS<RC>::LP) { incl = inc2; total = tot2; my = SYM_LOCAL | (lused + inc2 - need); lused += tot2; }
            else { my = sym_alloc(Y, AL, need, lane, incl, total); any_global = true; }
        }
        if (my == SYM_NONE) { exhausted = true; break; }
        S.rt_incl[lane] = (uint16_t)incl;
        if (act) {
            S.rt_slot[lane] = (uint16_t)s;
            S.Kd[s] = a; S.mk[s] = mask; S.off[s] = my;
            // (the open in-edges stay in the slot word: the lanes that compute this cell's coefficients below read them)
            uint32_t smw = (smv & (SS_CELL | (0xFFu << SS_OPEN_SHIFT))) | (td ? SS_TAINT : 0u) | SS_FIN;
            sp_of(L, cell) = (uint16_t)(2u << SP_STATE_SHIFT);
            uint32_t wk = 0;
            auto release = [&](int ti, int tj) {            // tile-local coordinates 0..TT-1 when inside
                if (ti >= 0 && ti < TH && tj >= 0 && tj < TT) {
                    const int tcell = ti * TT + tj;
                    if (sp_dec(L, tcell) == 1u) S.list[atomicAdd(&S.tail, 1)] = S.map[tcell];
                } else {
                    const int dti = ti < 0 ? -1 : (ti >= TH ? 1 : 0), dtj = tj < 0 ? -1 : (tj >= TT ? 1 : 0);
                    if (ti >= -TH && ti < 2 * TH && tj >= -TT && tj < 2 * TT) wk |= 1u << ((dti + 1) * 3 + dtj + 1);
                    else if (nent) wk |= 1u << 9;
                    else {
                        const int tt = ((i0 + ti) / TH) * tiles_x + (j0 + tj) / TT;
                        if (atomicExch(&N.flag[tt], (int32_t)pass + 1) != (int32_t)pass + 1) N.list[atomicAdd(N.count, 1)] = tt;
                    }
                }
            };
            const int sct = ci_section(cw);
            const int li = cell >> 5, lj = cell & 31;
            if (cw & CI_OUT1) release(li + fe1r(sct), lj + fe1c(sct));
            if (cw & CI_OUT2) release(li + fe2r(sct), lj + fe2c(sct));
            for (int q = 0, pf = S.po_first[s], pn = S.po_n[s]; q < pn; q++) {          // pit out-edges
                const uint32_t code = S.po_code[pf + q];
                if (code < 1024u) { if (sp_dec(L, (int)code) == 1u) S.list[atomicAdd(&S.tail, 1)] = S.map[code]; }
                else if (code < 0x8000u) wk |= 1u << (code - 1024u);
                else if (nent) wk |= 1u << 9;
                else {
                    const int tt = S.far_tile[code & 31u];
                    if (atomicExch(&N.flag[tt], (int32_t)pass + 1) != (int32_t)pass + 1) N.list[atomicAdd(N.count, 1)] = tt;
                }
            }
            if (nent) smw |= wk << SS_WAKE_SHIFT; else wake |= wk & 0x1FFu;
            S.sm[s] = smw;
        }
        // (the stores of the previous round's coefficients must have landed before this round reads them -- only once the
        // global pool is in play; the wait then overlaps with the LDS work above, the end of a round only orders LDS traffic)
        if (any_global) tile_wave_sync(); else tile_lds_sync();
        // ---- the coefficients of the round's cells, ONE lane per (cell, inlet): a cell's entries are sums over its open
        // in-edges of weight x the source's coefficient for the same inlet -- every lane has its own few loads in flight, the
        // round costs one trip to the pool however many entries its cells have (a lane walking its cell's entries one
        // after the other paid a trip per entry: 20 ms for this pass at 16384^2)
        for (uint32_t e = (uint32_t)lane; e < total; e += 64u) {
            int lo = 0, hi = 63;                               // owner: the first lane whose inclusive prefix exceeds e
            while (lo < hi) { const int mid = (lo + hi) >> 1; if ((uint32_t)S.rt_incl[mid] > e) hi = mid; else lo = mid + 1; }
            const int so = S.rt_slot[lo];
            const unsigned long long mo = S.mk[so];
            const int k = (int)(e - ((uint32_t)S.rt_incl[lo] - (uint32_t)__popcll(mo) - 1u));
            const uint32_t oo = S.off[so];
            double *dst = (oo & SYM_LOCAL) ? &S.lpool[oo & (SYM_LOCAL - 1u)] : sym_ptr(Y, oo);
            if (k == 0) { dst[0] = __longlong_as_double((long long)mo); continue; }
            unsigned long long mj = mo;
            for (int q = 1; q < k; q++) mj &= mj - 1ull;
            const int j = __ffsll((long long)mj) - 1;
            const unsigned long long below = (1ull << j) - 1ull;
            const uint32_t smo = S.sm[so];
            const int ocell = smo & SS_CELL, r = ocell >> 5, cc = ocell & 31;
            const uint32_t oom = outside_dirs(r, cc);
            double cf = 0.0;
            uint32_t mm = (smo >> SS_OPEN_SHIFT) & 0xFFu;
            while (mm) {
                const int d = __ffs(mm) - 1; mm &= mm - 1u;
                if ((oom >> d) & 1u) {
                    int di, dj; nb_delta(d, di, dj);
                    const int jj = S.hidx[halo_pos(r + 1 + di, cc + 1 + dj)];
                    if (jj == j) { const double pj = S.in_p[jj]; cf += ((0x5A >> d) & 1) ? pj : 1 - pj; }
                } else {
                    const int ss = S.map[sym_nb_local(ocell, d)];
                    const unsigned long long ms = S.mk[ss];
                    if ((ms >> j) & 1ull) {
                        const double ps = S.Pd[ss];
                        cf += (((0x5A >> d) & 1) ? ps : 1 - ps) * coef_at(S.off[ss], 1 + __popcll(ms & below));
                    }
                }
            }
            for (int q = 0, pf = S.pe_first[so], pn = S.pe_n[so]; q < pn; q++) {
                const uint32_t code = S.pe_code[pf + q];
                if (code & 0x8000u) { if ((int)(code & 63u) == j) cf += S.pe_w[pf + q]; }
                else {
                    const unsigned long long ms = S.mk[code];
                    if ((ms >> j) & 1ull) cf += S.pe_w[pf + q] * coef_at(S.off[code], 1 + __popcll(ms & below));
                }
            }
            dst[k] = cf;
        }
        head = tail < head + 64 ? tail : head + 64;
        tile_lds_sync();
#ifdef PYDEM_SYM_PROF
        nrounds++; nentries += (int)total;
#endif
    }
    if (exhausted) return false;
    SYM_TICK(5);
    // ---- the tile's block: inlets and outlets (a symbolic cell whose flow leaves the tile)
    int nout = 0, nsym = 0;
    for (int s0 = 0; s0 < nslot; s0 += 64) {
        const int s = s0 + lane;
        const uint32_t smv = s < nslot ? S.sm[s] : 0u;
        const bool sym = (smv & SS_FIN) && S.mk[s < nslot ? s : 0] != 0ull;
        nout += __popcll(__ballot(sym && ((smv >> SS_WAKE_SHIFT) & 0x3FFu)));
        nsym += __popcll(__ballot(sym));
    }
    if (nout > 64) return false;
    uint32_t blk = SYM_NONE, lbase = 0;
    if (nsym) {
        const uint32_t hdr = 2u + (uint32_t)((nin_all + 1) / 2) + 4u * (uint32_t)nout;
        uint32_t incl_b = 0, total_b = 0;
        blk = sym_alloc(Y, AL, lane == 0 ? hdr + lused : 0u, lane, incl_b, total_b);     // the block, then the coefficients that stayed on chip
        blk = __builtin_amdgcn_readfirstlane(__shfl(blk, 0));
        if (blk == SYM_NONE) return false;
        lbase = blk + hdr;
        double *G = sym_ptr(Y, lbase);
        for (uint32_t q = (uint32_t)lane; q < lused; q += 64u) G[q] = S.lpool[q];
        double *B = sym_ptr(Y, blk);
        if (lane == 0) {
            B[0] = __longlong_as_double((long long)((unsigned long long)nin_all | ((unsigned long long)nout << 8)));
            B[1] = __longlong_as_double(0ll);                                 // outlets resolved so far
        }
        int32_t *ids = reinterpret_cast<int32_t *>(B + 2);
        if (lane < nin_all) ids[lane] = S.in_id[lane];
        if (lane == nin_all && (nin_all & 1)) ids[lane] = 0;
    }
    // ---- results: cells no inlet reaches are finished; the others leave their header in the contribution slot
    int nfin = 0, obase = 0;
    double *R = blk != SYM_NONE ? sym_ptr(Y, blk) + 2 + (nin_all + 1) / 2 : nullptr;
    for (int s0 = 0; s0 < nslot; s0 += 64) {
        const int s = s0 + lane;
        const uint32_t smv = s < nslot ? S.sm[s] : 0u;
        const bool fin = (smv & SS_FIN) != 0u;
        const unsigned long long mask = fin ? S.mk[s] : 0ull;
        const bool is_out = fin && mask != 0ull && ((smv >> SS_WAKE_SHIFT) & 0x3FFu);
        const unsigned long long bo = __ballot(is_out);
        if (fin) {
            const int cell = smv & SS_CELL;
            const int32_t c = (i0 + (cell >> 5)) * m + j0 + (cell & 31);
            const uint32_t cw = L.cs[cell] >> 16;
            const double a = S.Kd[s], pv = S.Pd[s];
            if (mask == 0ull) {
                double2 o = make_double2(0.0, 0.0);
                if (cw & CI_OUT1) o.x = a * pv;
                if (cw & CI_OUT2) o.y = a * (1 - pv);
                if (smv & SS_TAINT) { o.x = -o.x; o.y = -o.y; A.todo_work[c] = 1; }
                A.area[c] = a;
                A.contrib[c] = o;
                A.cinfo[c] = ci_with_level(cw, pass);
                nfin++;
            } else {
                const uint32_t goff = (S.off[s] & SYM_LOCAL) ? lbase + (S.off[s] & (SYM_LOCAL - 1u)) : S.off[s];
                const unsigned long long hb = (unsigned long long)goff | ((smv & SS_TAINT) ? (1ull << 32) : 0ull);
                A.contrib[c] = make_double2(a, __longlong_as_double((long long)hb));
                A.cinfo[c] = ci_with_level(cw, CI_LEVEL_SYM);
                if (is_out) {
                    double *rec = R + 4 * (obase + __popcll(bo & ((1ull << lane) - 1ull)));
                    const unsigned long long wk = (smv >> SS_WAKE_SHIFT) & 0x3FFu;
                    rec[0] = __longlong_as_double((long long)((unsigned long long)cell | (wk << 16) | ((unsigned long long)goff << 32)));
                    rec[1] = a;
                    rec[2] = pv;
                    int32_t pout = 0;
                    if (cw & CI_PIT_OUT) pout = pit_stash(A, c).y;
                    rec[3] = __longlong_as_double((long long)((unsigned long long)(uint32_t)pout | ((unsigned long long)cw << 32) |
                                                              ((smv & SS_TAINT) ? (1ull << 47) : 0ull)));
                }
            }
        }
        obase += __popcll(bo);
    }
    for (int off = 32; off > 0; off >>= 1) { nfin += __shfl_down(nfin, off); wake |= __shfl_xor(wake, off); }
    bool win = false;
    int tt = 0;
    if (lane < 9 && ((wake >> lane) & 1u)) { tt = tid + (lane / 3 - 1) * tiles_x + (lane % 3 - 1); win = true; }
    if (win) win = atomicExch(&N.flag[tt], (int32_t)pass + 1) != (int32_t)pass + 1;
    const unsigned long long bw = __ballot(win);
    if (win) pend[npend + __popcll(bw & ((1ull << lane) - 1ull))] = tt;
    npend += __popcll(bw);
#ifdef PYDEM_SYM_PROF
    if (lane == 0) {
        SYM_TICK(6);
        unsigned long long *acc = reinterpret_cast<unsigned long long *>(Y.stat + 8);
        for (int q = 0; q < 6; q++) atomicAdd(acc + q, (unsigned long long)(tk[q + 1] - tk[q]));
        atomicAdd(acc + 6, (unsigned long long)nrounds); atomicAdd(acc + 7, (unsigned long long)nentries);
        atomicAdd(acc + 8, 1ull); atomicAdd(acc + 9, (unsigned long long)nslot);
    }
#endif
    if (lane == 0) {
        n_final += nfin;
        A.tile_open[tid] = nslot - nfin;
        if (nfin == nslot) tile_done[tid] = 1;
        Y.tile_sym[tid] = blk;
        if (blk != SYM_NONE) { atomicAdd(&Y.stat[0], 1); atomicAdd(&Y.stat[2], nout); atomicAdd(&Y.stat[3], nsym); }
    }
    tile_wave_sync();
    return true;
}

// LDS of a light visit / of the final evaluation: values and taints of the tile's inlets
struct SymLight { double x[SYM_MAXIN]; uint8_t t[SYM_MAXIN]; };

// the inlets of a block that are final for a visit of pass `pass` (bit j of the result), their areas / taints into LDS
__device__ __forceinline__ unsigned long long sym_load_inlets(const SweepArgs &A, const double *B, int J, SymLight &S, uint32_t pass, int lane)
{
    const int32_t *ids = reinterpret_cast<const int32_t *>(B + 2);
    bool fin = false;
    if (lane < J) {
        const int32_t u = ids[lane];
        const uint32_t lv = ci_level(A.cinfo[u]);
        fin = lv >= 1 && lv < pass;
        if (fin) { S.x[lane] = A.area[u]; S.t[lane] = A.todo_work[u]; }
    }
    return __ballot(fin);
}

// (b) light visit of a symbolic tile: every outlet all of whose inlets are final by now is evaluated and written like a
// finished cell; the tiles it drains into are listed for the next pass
__device__ __forceinline__ void sym_light_visit(const SweepArgs &A, const SymArgs &Y, SymLight &S, uint32_t pass, int tiles_x, int tid, int lane,
                                                uint32_t blk, int32_t &n_final, const TileNext &N, int32_t *pend, int &npend)
{
    double *B = sym_ptr(Y, blk);
    const unsigned long long h0 = (unsigned long long)__double_as_longlong(B[0]);
    const int J = (int)(h0 & 0xFFull), O = (int)((h0 >> 8) & 0xFFull);
    const unsigned long long res = (unsigned long long)__double_as_longlong(B[1]);
    const unsigned long long F = sym_load_inlets(A, B, J, S, pass, lane);
    tile_wave_sync();
    const double *R = B + 2 + (J + 1) / 2;
    const int by = tid / tiles_x, bx = tid - by * tiles_x;
    const int i0 = by * TH, j0 = bx * TT, m = A.m;
    bool did = false;
    uint32_t wake = 0;
    if (lane < O && !((res >> lane) & 1ull)) {
        const unsigned long long r0 = (unsigned long long)__double_as_longlong(R[4 * lane]);
        const double *E = sym_ptr(Y, (uint32_t)(r0 >> 32));
        unsigned long long mask = (unsigned long long)__double_as_longlong(E[0]);
        if (!(mask & ~F)) {
            const unsigned long long r3 = (unsigned long long)__double_as_longlong(R[4 * lane + 3]);
            double a = R[4 * lane + 1];
            const double pv = R[4 * lane + 2];
            bool td = ((r3 >> 47) & 1ull) != 0ull;
            for (int k = 1; mask; k++) {
                const int j = __ffsll((long long)mask) - 1; mask &= mask - 1ull;
                a += E[k] * S.x[j];
                td = td || (S.t[j] != 0);
            }
            const int cell = (int)(r0 & 0x3FFull);
            const int32_t c = (i0 + (cell >> 5)) * m + j0 + (cell & 31);
            const uint32_t cw = (uint32_t)(r3 >> 32) & CI_STATIC_MASK;
            double2 o = make_double2(0.0, 0.0);
            if (cw & CI_OUT1) o.x = a * pv;
            if (cw & CI_OUT2) o.y = a * (1 - pv);
            if (td) { o.x = -o.x; o.y = -o.y; A.todo_work[c] = 1; }
            A.area[c] = a;
            A.contrib[c] = o;
            A.cinfo[c] = ci_with_level(cw, pass);
            did = true;
            const uint32_t wk = (uint32_t)(r0 >> 16) & 0x3FFu;
            wake = wk & 0x1FFu;
            if (wk & 0x200u)                                  // a pit that drains further away than the next tile
                for (int32_t e = (int32_t)(uint32_t)r3; e < A.n_pit && A.pit_src[e] == c; e++) {
                    const int32_t dc = A.pit_dst[e];
                    const int ti = dc / m - i0, tj = dc % m - j0;
                    if (ti >= -TH && ti < 2 * TH && tj >= -TT && tj < 2 * TT) continue;
                    const int tt = (dc / m / TH) * tiles_x + (dc % m) / TT;
                    if (atomicExch(&N.flag[tt], (int32_t)pass + 1) != (int32_t)pass + 1) N.list[atomicAdd(N.count, 1)] = tt;
                }
        }
    }
    const unsigned long long bd = __ballot(did);
    for (int off = 32; off > 0; off >>= 1) wake |= __shfl_xor(wake, off);
    bool win = false;
    int tt = 0;
    if (lane < 9 && lane != 4 && ((wake >> lane) & 1u)) { tt = tid + (lane / 3 - 1) * tiles_x + (lane % 3 - 1); win = true; }
    if (win) win = atomicExch(&N.flag[tt], (int32_t)pass + 1) != (int32_t)pass + 1;
    const unsigned long long bw = __ballot(win);
    if (win) pend[npend + __popcll(bw & ((1ull << lane) - 1ull))] = tt;
    npend += __popcll(bw);
    if (lane == 0 && bd) { B[1] = __longlong_as_double((long long)(res | bd)); n_final += __popcll(bd); }
    tile_wave_sync();
}

// the tiles of the symbolic pass: every tile that is not done, by class -- at most `split` open cells from the front of
// `list`, the others from its back (one array of tiles_total entries holds both)
__global__ __launch_bounds__(256) void k_sym_candidates(const uint8_t *__restrict__ tile_done, const int32_t *__restrict__ tile_open, int tiles_total,
                                                        int split, int32_t *__restrict__ list, int32_t *cnt2)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool cand = tid < tiles_total && !tile_done[tid];
    const bool small = cand && tile_open[tid] <= split;
    const unsigned long long ba = __ballot(small), bb = __ballot(cand && !small);
    int32_t base_a = 0, base_b = 0;
    if (lane == 0) { if (ba) base_a = atomicAdd(&cnt2[0], __popcll(ba)); if (bb) base_b = atomicAdd(&cnt2[1], __popcll(bb)); }
    base_a = __shfl(base_a, 0); base_b = __shfl(base_b, 0);
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (small) list[base_a + __popcll(ba & lt)] = tid;
    else if (cand) list[tiles_total - 1 - (base_b + __popcll(bb & lt))] = tid;
}

// (a) the symbolic pass over one class of the unfinished tiles (k_sym_candidates): symbolic visit when the tile's open
// cells fit the slots of this instantiation (lo < open cells <= RC; 1024 slots hold any tile), the numeric visit of this pass
// for a tile whose symbolic visit gave up (inlets, outlets, pool).  One wavefront per workgroup, the list entries strided over the workgroups.
template <int RC>
__global__ __launch_bounds__(64) void k_sweep_sym(SweepArgs A, SymArgs Y, uint32_t pass, int tiles_x, int tiles_total, uint8_t *__restrict__ tile_done,
                                                  int32_t *n_final, TileNext N, const int32_t *__restrict__ list, const int32_t *cnt, int from_back, int lo)
{
    __shared__ TileS<RC> S;
    __shared__ int32_t s_pend[TILE_PEND];
    __shared__ int32_t s_nbr16[8];
    fill_nbr16(s_nbr16, A.m);
    __syncthreads();
    const int lane = threadIdx.x;
    int32_t fin = 0;
    int npend = 0;
    SymAlloc AL;
    AL.cur = 0; AL.end = 0; AL.reg = (int)(blockIdx.x & (unsigned)(Y.nreg - 1));
    auto flush = [&]() {
        tile_wave_sync();
        int32_t base = 0;
        if (lane == 0) base = atomicAdd(N.count, npend);
        base = __shfl(base, 0);
        if (lane < npend) N.list[base + lane] = s_pend[lane];
        npend = 0;
        tile_wave_sync();
    };
    const int32_t nt = *cnt;
    for (int32_t k = blockIdx.x; k < nt; k += gridDim.x) {
        const int tid = __builtin_amdgcn_readfirstlane(list[from_back ? tiles_total - 1 - k : k]);
        const int open = __builtin_amdgcn_readfirstlane(A.tile_open[tid]);
        if (open <= lo || open > RC) continue;              // (another instantiation's tile)
        const bool numeric = !sym_visit<RC>(A, Y, AL, S, pass, tiles_x, tid, lane, tile_done, fin, N, s_pend, npend);
        if (numeric) {
            if (lane == 0) atomicAdd(&Y.stat[1], 1);
            tile_wave_sync();
            sweep_one_tile<true>(A, S.W, pass, tiles_x, tid, lane, tile_done, fin, N, s_pend, npend, s_nbr16);
        }
        if (npend > TILE_PEND - 10) flush();
    }
    if (npend) flush();
    if (lane == 0 && fin) atomicAdd(n_final, fin);
}

// (c) after the last pass: every cell that still carries a symbolic header is evaluated (all inlets final) or, below a
// drainage loop, handed back to the unfinished cells (level "not yet known": the re-seed replay K5c reads that)
__global__ __launch_bounds__(64) void k_sym_finish(SweepArgs A, SymArgs Y, uint32_t pass, int tiles_x, int tiles_total, int32_t *n_final)
{
    __shared__ SymLight S;
    const int lane = threadIdx.x;
    const int n = A.n, m = A.m;
    int32_t fin = 0;
    for (int tid = blockIdx.x; tid < tiles_total; tid += gridDim.x) {
        const uint32_t blk = __builtin_amdgcn_readfirstlane(Y.tile_sym[tid]);
        if (blk == SYM_NONE) continue;
        const double *B = sym_ptr(Y, blk);
        const int J = (int)((unsigned long long)__double_as_longlong(B[0]) & 0xFFull);
        const unsigned long long F = sym_load_inlets(A, B, J, S, pass, lane);
        tile_wave_sync();
        const int by = tid / tiles_x, bx = tid - by * tiles_x;
        const int i0 = by * TH, j0 = bx * TT;
        constexpr int NSET = TH * TT / 64;
        uint32_t w[4];
#pragma unroll 1
        for (int kb = 0; kb < NSET; kb += 4) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int idx = lane + 64 * (kb + k), gi = i0 + (idx >> 5), gj = j0 + (idx & 31);
                w[k] = (gi < n && gj < m) ? A.cinfo[gi * m + gj] : 0u;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (ci_level(w[k]) != CI_LEVEL_SYM) continue;
                const int idx = lane + 64 * (kb + k);
                const int32_t c = (i0 + (idx >> 5)) * m + j0 + (idx & 31);
                const uint32_t cw = w[k] & CI_STATIC_MASK;
                const double2 h = A.contrib[c];
                const unsigned long long hb = (unsigned long long)__double_as_longlong(h.y);
                const double *E = sym_ptr(Y, (uint32_t)hb);
                unsigned long long mask = (unsigned long long)__double_as_longlong(E[0]);
                if (mask & ~F) { A.cinfo[c] = ci_with_level(cw, CI_LEVEL_INF); continue; }
                double a = h.x;
                bool td = ((hb >> 32) & 1ull) != 0ull;
                for (int q = 1; mask; q++) {
                    const int j = __ffsll((long long)mask) - 1; mask &= mask - 1ull;
                    a += E[q] * S.x[j];
                    td = td || (S.t[j] != 0);
                }
                double2 o = make_double2(0.0, 0.0);
                if (cw & (CI_OUT1 | CI_OUT2)) {
                    const double pv = A.prop[c];
                    if (cw & CI_OUT1) o.x = a * pv;
                    if (cw & CI_OUT2) o.y = a * (1 - pv);
                }
                if (td) { o.x = -o.x; o.y = -o.y; A.todo_work[c] = 1; }
                A.area[c] = a;
                A.contrib[c] = o;
                A.cinfo[c] = ci_with_level(cw, pass);
                fin++;
            }
        }
        tile_wave_sync();
    }
    for (int off = 32; off > 0; off >>= 1) fin += __shfl_down(fin, off);
    if (lane == 0 && fin) atomicAdd(n_final, fin);
}
