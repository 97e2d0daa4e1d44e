// uca_cond.inl -- included by uca.hip (inside its anonymous namespace, behind the compact incremental rounds K7i).
//
// Condensed ("transfer operator") form of the incremental edge rounds.
//
// Reference: the multi-worker fix-up pydem/process_manager.py:224-284, 1090-1246 (worker calc_uca_ec, the schedule) and
// the round itself, pydem/dem_processing.py:778-862.  The compact incremental round (NDRec, above) walks its cascade cell
// by cell: a round costs as many dependent levels as the rivers below its seeds are long (300-900 levels of ~2.2 us at
// 16384^2, 127 waves per mosaic).  But between two rounds nobody looks at the interior of a tile: the schedule only reads
// the WATCHED lines -- the perimeter (where strips enter) and the lines other tiles read (the edge board's interest
// lines).  So the not-done sub-graph ND is condensed once per fix-up into a graph on its watched cells W alone:
//
//     edge p -> q  (p, q in W)  with weight  T(q, p) = sum over the flow paths p -> q whose inner cells are not in W of
//                                                      the products of the edge weights (pit -> drain edges included)
//
// A round is then the SAME cascade (counts, seeds that adopt a finished value, deltas handed downstream, NaN flood, the
// split FINAL / DONE) on a few thousand nodes with a depth of a handful of levels; `done` of a watched cell means
// what it meant before (every 'todo' inlet upstream of it has been released: by induction over the paths, that is "all
// its W-predecessors are done"), its delta is the same linear combination of the seeds' deltas, summed in a fixed order
// (slot per in-edge, ascending source): deterministic, equal to the cell-by-cell cascade up to the rounding of a
// re-associated sum.  The interior catches up when somebody needs it (pydem_uca_edge_flush, a download of the masks, a
// line that is not watched): the done W nodes are released into the compact records and ONE ordinary cascade finishes
// everything below them (stage_edge_catchup); the NaN flood continues below the nodes it passed.
//
// The operator is built on the host from a copy of the records (reverse topological order: the vector a cell carries is
// "which watched cells does my water reach next, with what weight" -- flow converges, those vectors stay short; the
// other direction, "which inlets feed me", grows to thousands of entries along a river).
#include <algorithm>
#include <utility>

constexpr uint32_t CF_RELEASED = 1u << 8, CF_NANPASS = 1u << 9, CF_NANINT = 1u << 10;
constexpr int COND_QCAP = 4096;
constexpr int COND_THREADS = 1024;    // the cascade's workgroup (the first rounds of a fix-up are thousands of nodes wide; 256 threads: same median, slower maximum)
constexpr int32_t W_BARRIER = 1 << 30;        // added to the full-graph count of a watched record: the interior cascade never fires it

struct CEdge { int32_t dst, slot; double w; };      // slot >= 0: index into the slot array; < 0: inline slot -1 - slot of node dst
// One node = one 128-byte line: the first two in-slots and the first two out-edges (the common case: a watched cell has two
// neighbours on the watched graph) travel with the node, so a level of the cascade is ONE dependent load per node, the
// hand-over stores and the returning count-down; the rest lives in the slot / edge arrays.
struct __attribute__((aligned(128))) CNode {
    int32_t rec, cell;
    int32_t cnt;                 // unresolved condensed in-edges (+1: the outside of the tile while the cell is a 'todo' inlet)
    uint32_t flag;               // NF_* like the records, CF_*
    double delta;
    uint32_t cw;                 // graph word of the cell (ND_FLAT)
    uint32_t seed_round;
    int32_t n_in, n_out;         // in-edges (one slot each, ascending source) / out-edges
    int32_t in_base, out_base;   // where slots 2.. / edges 2.. live
    double in_inl[2];
    CEdge e_inl[2];
    int32_t pad[8];
};
static_assert(sizeof(CNode) == 128, "one node per cache line");

struct CondArgsE {
    CIncArgs C;
    CNode *node; int32_t nw;
    const CEdge *edge;
    double *slot;
    int32_t *q0, *q1;            // level queues (nw entries each)
    int32_t *nanq; int32_t nan_cap;   // NaN flood list (every edge can enter it once per flood)
    int32_t *cnt;                // [0] first level, [1] NaN seeds
    // queued waves (pydem_board_run_waves): the round runs only when bit `gate_bit` of *gate is set -- the wave's members are
    // chosen on the device, the host has queued the round for every tile (gate == nullptr: an ordinary round)
    const unsigned long long *gate; int32_t gate_bit;
    // ... and its seed stamp is read on the device as well (the queued wave is a captured graph: no per-wave kernel argument):
    // the tile's round counter when the batch began + the number of the wave in the batch
    const unsigned long long *round_base, *round_add;
};

__device__ __forceinline__ bool cond_gated_off(const CondArgsE &X)
{
    return X.gate != nullptr && !((*X.gate >> X.gate_bit) & 1ull);
}
__device__ __forceinline__ uint32_t cond_round16(const CondArgsE &X)
{
    return X.round_base ? (uint32_t)((*X.round_base + *X.round_add) % 65535ull) + 1u : X.C.round16;
}


__device__ __forceinline__ int32_t cond_wid(const CIncArgs &E, int32_t c)
{
    const int32_t k = E.cid[c] - 1;
    return k >= 0 ? E.rec[k].wid : -1;
}

__device__ __forceinline__ CEdge cond_edge(const CondArgsE &X, const CNode &N, int e) { return e < 2 ? N.e_inl[e] : X.edge[N.out_base + e - 2]; }

// strips -> events on the perimeter, exactly k_cinc_seed with the watched nodes in place of the records
__device__ __forceinline__ void cond_seed_cell(const CondArgsE &X, const double *__restrict__ sdata, const uint8_t *__restrict__ sdone,
                                               const uint8_t *__restrict__ stodo, int L, int64_t p)
{
    const CIncArgs &E = X.C;
    const uint32_t round16 = cond_round16(X);
    const int n = E.G.n, m = E.G.m;
    const int64_t nper = 2 * (int64_t)m + 2 * (int64_t)(n - 2);
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
    const int32_t w = cond_wid(E, c);
    if (dn) {
        const double d = E.flats[c] ? NAN : init - E.uca[c];
        E.uca[c] += d;
        E.edge_todo[c] = 0;
        if (w >= 0 && !own_done) X.node[w].seed_round = round16;
        if (w >= 0 && !own_done && !(X.node[w].flag & NF_FINAL)) {
            CNode &N = X.node[w];
            N.delta = d;
            N.flag = (N.flag & (NF_NAN | CF_NANPASS | CF_NANINT)) | NF_FINAL | NF_SEED;
            if (d != d) X.nanq[atomicAdd(&X.cnt[1], 1)] = w;
            if (own_todo) {
                const int32_t old = atomicSub(&N.cnt, 1);
                if (old == 1) X.q0[agg_slot(&X.cnt[0])] = w;
            }
        } else if (w >= 0 && !own_done) {
            X.node[w].delta += d;
            if (d != d) X.nanq[atomicAdd(&X.cnt[1], 1)] = w;
        }
    } else if (own_todo && !td) {
        E.edge_todo[c] = 0;
        if (w >= 0) {
            const int32_t old = atomicSub(&X.node[w].cnt, 1);
            if (old == 1) X.q0[agg_slot(&X.cnt[0])] = w;
        }
    }
}

__global__ void k_cond_seed(CondArgsE X, const double *__restrict__ sdata, const uint8_t *__restrict__ sdone,
                            const uint8_t *__restrict__ stodo, int L)
{
    cond_seed_cell(X, sdata, sdone, stodo, L, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// Queued waves (pydem_board_run_waves): the rounds of ALL tiles of this rank in one launch each -- blockIdx.y (seeds) /
// blockIdx.x (cascade) = entry of a device table, a tile takes part when its bit of the wave's member word is set.
struct QTile { CondArgsE X; const double *sdata; const uint8_t *sflags; int64_t nper; int32_t L, pad; };

__global__ void k_cond_seed_q(const QTile *__restrict__ Q)
{
    const QTile q = Q[blockIdx.y];
    if (cond_gated_off(q.X)) return;
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= q.nper) return;
    cond_seed_cell(q.X, q.sdata, q.sflags, q.sflags + (size_t)4 * q.L, q.L, p);
}

// the final flush: the remaining inlets let go of the outside (k_cinc_release_todo)
__global__ void k_cond_release_todo(CondArgsE X)
{
    const CIncArgs &E = X.C;
    const int n = E.G.n, m = E.G.m;
    const int64_t nper = 2 * (int64_t)m + 2 * (int64_t)(n - 2);
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nper) return;
    int i, j;
    perim_cell(p, n, m, i, j);
    const int32_t c = i * m + j;
    const int32_t w = cond_wid(E, c);
    if (w < 0 || !E.edge_todo[c] || (X.node[w].flag & NF_FINAL)) return;
    const int32_t old = atomicSub(&X.node[w].cnt, 1);
    if (old == 1) X.q0[agg_slot(&X.cnt[0])] = w;
}

// ONE workgroup: the NaN flood of the round (k_cinc_nan_flood on the condensed graph), then the cascade level after level
// (cinc_cell on nodes).  No host look in between: a wave of the fix-up is seed kernel + this kernel + the board's pack.
__device__ __forceinline__ void cond_run_block(const CondArgsE &X)
{
    const CIncArgs &E = X.C;
    const uint32_t round16 = cond_round16(X);
    __shared__ int32_t s_tail, s_cnt[3];
    // ---- NaN flood
    if (threadIdx.x == 0) s_tail = X.cnt[1];
    __syncthreads();
    {
        int32_t head = 0, tail = s_tail;
        const int32_t n_origin = tail;
        while (head < tail) {
            for (int32_t q = head + threadIdx.x; q < tail; q += blockDim.x) {
                CNode &N = X.node[X.nanq[q]];
                if (q >= n_origin && N.seed_round == round16) continue;            // a seed of this round keeps its value
                if (atomicOr(&N.flag, NF_NAN | CF_NANPASS) & NF_NAN) continue;          // flooded in an earlier round
                E.uca[N.cell] = NAN;
                for (int e = 0; e < N.n_out; e++) {
                    const int32_t t = cond_edge(X, N, e).dst;
                    // (every node enters the list at most once per flood: claimed through a flag of its own)
                    if (!(X.node[t].flag & NF_NAN)) { const int32_t s = atomicAdd(&s_tail, 1); if (s < X.nan_cap) X.nanq[s] = t; }
                }
            }
            __syncthreads();
            head = tail; tail = s_tail < X.nan_cap ? s_tail : X.nan_cap;
            __syncthreads();
        }
    }
    // ---- cascade: the level queues live in LDS (entries beyond COND_QCAP in the global queues)
    __shared__ int32_t s_q[2][COND_QCAP];
    __shared__ int32_t s_done;
    if (threadIdx.x == 0) { s_cnt[0] = X.cnt[0]; s_cnt[1] = 0; s_cnt[2] = 0; s_done = 0; }
    __syncthreads();
    int r = 0;
    int32_t nq = s_cnt[0];
    for (int32_t k = threadIdx.x; k < nq && k < COND_QCAP; k += blockDim.x) s_q[0][k] = X.q0[k];
    __syncthreads();
    while (nq > 0) {
        const int32_t *qc = (r & 1) ? X.q1 : X.q0;
        int32_t *qn = (r & 1) ? X.q0 : X.q1;
        const int32_t *lc = s_q[r & 1];
        int32_t *ln = s_q[(r + 1) & 1];
        int32_t *cn = &s_cnt[(r + 1) % 3];
        if (threadIdx.x == 0) s_cnt[(r + 2) % 3] = 0;
        for (int32_t k = threadIdx.x; k < nq; k += blockDim.x) {
            CNode &N = X.node[k < COND_QCAP ? lc[k] : qc[k]];
            // the node's line, loaded whole: the only dependent access of a level
            const uint4 *line = reinterpret_cast<const uint4 *>(&N);
            uint4 Lw[6];
#pragma unroll
            for (int i = 0; i < 6; i++) Lw[i] = line[i];
            CNode V;
            __builtin_memcpy(&V, Lw, 96);
            const uint32_t f = V.flag;
            double delta = V.delta;
            if (!(f & NF_FINAL)) {
                double acc = (V.cw & ND_FLAT) ? NAN : 0.0;                           // :815
                acc += V.in_inl[0];                                                  // fixed order: ascending source (empty slots hold 0)
                acc += V.in_inl[1];
                // (slots 2.. in batches of eight loads, then added in slot order: a watched cell below a confluence has up to ~200
                // in-edges, and one dependent load per slot made such a node the whole level's duration)
                for (int s2 = 2; s2 < V.n_in; s2 += 8) {
                    double tv[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) tv[u] = s2 + u < V.n_in ? X.slot[V.in_base + s2 + u - 2] : 0.0;
#pragma unroll
                    for (int u = 0; u < 8; u++) if (s2 + u < V.n_in) acc += tv[u];
                }
                delta = acc;
                N.delta = acc;
                N.flag = (f & (NF_NAN | CF_NANPASS | CF_NANINT)) | NF_FINAL | NF_DONE | NF_APPLIED;
            } else {
                N.flag = f | NF_DONE | NF_APPLIED;                                   // a seed took its value when the strip arrived
            }
            // (the cell's area and mask are written after the cascade, all cells at once: a read-modify-write of the area
            // plane inside the level would make every level's barrier wait for an HBM round trip)
            X.nanq[agg_slot(&s_done)] = (k < COND_QCAP ? lc[k] : qc[k]) | ((f & NF_FINAL) ? (int32_t)0x40000000 : 0);
            // the two edges that travel with the node: both hand-overs, then both count-downs back to back (a count-down is a
            // round trip to the L2; tested one after the other they were half of a narrow level), then the releases
            int32_t old_cnt[2] = {0, 0};
#pragma unroll
            for (int e = 0; e < 2; e++)
                if (e < V.n_out) {
                    const CEdge ed = V.e_inl[e];
                    if (ed.slot < 0) X.node[ed.dst].in_inl[-1 - ed.slot] = delta * ed.w;
                    else X.slot[ed.slot] = delta * ed.w;
                }
#pragma unroll
            for (int e = 0; e < 2; e++)
                if (e < V.n_out) old_cnt[e] = atomicSub(&X.node[V.e_inl[e].dst].cnt, 1);
#pragma unroll
            for (int e = 0; e < 2; e++)
                if (e < V.n_out && old_cnt[e] == 1) {
                    const int32_t sl = agg_slot(cn);
                    if (sl < COND_QCAP) ln[sl] = V.e_inl[e].dst; else qn[sl] = V.e_inl[e].dst;
                }
            // the other out-edges (a quarter of the nodes have some, a few hundred per tile 17-70) eight at a time: the edges loaded
            // together, the hand-overs stored, the count-downs issued back to back, then the releases -- two dependent round trips per
            // eight edges instead of two per edge
            for (int e = 2; e < V.n_out; e += 8) {
                CEdge ex[8];
                int32_t oldx[8];
#pragma unroll
                for (int u = 0; u < 8; u++) if (e + u < V.n_out) ex[u] = X.edge[V.out_base + e + u - 2];
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (e + u < V.n_out) {
                        if (ex[u].slot < 0) X.node[ex[u].dst].in_inl[-1 - ex[u].slot] = delta * ex[u].w;
                        else X.slot[ex[u].slot] = delta * ex[u].w;
                    }
#pragma unroll
                for (int u = 0; u < 8; u++) oldx[u] = e + u < V.n_out ? atomicSub(&X.node[ex[u].dst].cnt, 1) : 0;
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (e + u < V.n_out && oldx[u] == 1) {
                        const int32_t sl = agg_slot(cn);
                        if (sl < COND_QCAP) ln[sl] = ex[u].dst; else qn[sl] = ex[u].dst;
                    }
            }
        }
        __syncthreads();
        nq = *cn;
        r++;
    }
    // ---- areas and masks of the cells this round finished (the NaN list is idle after the flood: it holds them)
    for (int32_t k = threadIdx.x; k < s_done; k += blockDim.x) {
        const int32_t e = X.nanq[k];
        const CNode &N = X.node[e & 0x3FFFFFFF];
        if (!(e & 0x40000000)) E.uca[N.cell] += N.delta;                             // (a seed took its value when the strip arrived)
        if (E.set_done) E.edge_done[N.cell] = 1;
    }
    if (threadIdx.x == 0) { X.cnt[0] = 0; X.cnt[1] = 0; X.cnt[2] = r; X.cnt[3] += r; X.cnt[4] += 1; X.cnt[5] += s_done; }   // ([3..5]: levels / rounds / nodes finished so far, PYDEM_EDGE_DEBUG)
}

__global__ __launch_bounds__(COND_THREADS) void k_cond_run(CondArgsE X)
{
    cond_run_block(X);
}

__global__ __launch_bounds__(COND_THREADS) void k_cond_run_q(const QTile *__restrict__ Q)
{
    const CondArgsE X = Q[blockIdx.x].X;
    if (cond_gated_off(X)) return;
    cond_run_block(X);
}

// catch-up, step 1: the watched nodes that are done hand their state to their compact records and enter the records'
// cascade as its first frontier (a FINAL record releases its targets with the delta it holds, cinc_cell)
__global__ void k_cond_release(CondArgsE X, QE *q, int32_t *nq)
{
    for (int32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < X.nw; w += gridDim.x * blockDim.x) {
        CNode &N = X.node[w];
        const uint32_t f = N.flag;
        if (!(f & NF_DONE) || (f & CF_RELEASED)) continue;
        NDRec &R = X.C.rec[N.rec];
        R.delta = N.delta;
        R.flag = (f & (NF_NAN | NF_SEED)) | NF_FINAL | NF_APPLIED;                   // (cinc_cell adds NF_DONE)
        N.flag = f | CF_RELEASED;
        QE e; e.c = N.rec; e.cw = 0;
        q[agg_slot(nq)] = e;
    }
}

// catch-up, step 2: the NaN flood goes on below the watched nodes it passed, through interior records only (the watched
// ones were handled in their round: a seed of that round stopped it, everything else let it through)
__global__ __launch_bounds__(1024) void k_cond_nan_interior(CondArgsE X)
{
    const CIncArgs &E = X.C;
    __shared__ int32_t s_tail;
    int32_t *list = E.nanq;
    if (threadIdx.x == 0) s_tail = 0;
    __syncthreads();
    for (int32_t w = threadIdx.x; w < X.nw; w += blockDim.x) {
        CNode &N = X.node[w];
        if ((N.flag & CF_NANPASS) && !(N.flag & CF_NANINT)) { N.flag |= CF_NANINT; list[atomicAdd(&s_tail, 1)] = N.rec; }
    }
    __syncthreads();
    int32_t head = 0, tail = s_tail;
    const int32_t n_origin = tail;
    while (head < tail) {
        for (int32_t q = head + threadIdx.x; q < tail; q += blockDim.x) {
            NDRec &R = E.rec[list[q]];
            if (q >= n_origin) {
                if (atomicOr(&R.flag, NF_NAN) & NF_NAN) continue;
                E.uca[R.cell] = NAN;
            }
            for (int o = 0; o < 2; o++) {
                const int32_t t = R.out_id[o];
                if (t >= 0 && E.rec[t].wid < 0 && !(E.rec[t].flag & NF_NAN)) list[atomicAdd(&s_tail, 1)] = t;
            }
            if (R.cw & CI_PIT_OUT) {
                const SweepArgs &A = E.G;
                for (int32_t e = E.pit_off[R.cell].y; e < A.n_pit && A.pit_src[e] == R.cell; e++) {
                    const int32_t t = E.cid[A.pit_dst[e]] - 1;
                    if (t >= 0 && E.rec[t].wid < 0 && !(E.rec[t].flag & NF_NAN)) list[atomicAdd(&s_tail, 1)] = t;
                }
            }
        }
        __syncthreads();
        head = tail; tail = s_tail;
        __syncthreads();
    }
}

// ---- build ------------------------------------------------------------------------------------------------------
// watched lines -> records
__global__ void k_cond_mark(CIncArgs E, int axis, int64_t index)
{
    const int n = E.G.n, m = E.G.m;
    const int64_t count = axis == 0 ? m : n;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < count; p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = axis == 0 ? index * m + p : p * m + index;
        const int32_t k = E.cid[c] - 1;
        if (k >= 0) E.rec[k].wid = -2;
    }
}

struct CPitEdge { int32_t src, dst; double w; };
// pit -> drain edges between two ND records (k, kt, weight); w_sorted = the weights in (src, dst) order
__global__ void k_cond_pit_edges(CIncArgs E, const double *__restrict__ w_sorted, CPitEdge *out, int32_t *count, int32_t cap)
{
    const SweepArgs &A = E.G;
    for (int32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < E.nd; k += gridDim.x * blockDim.x) {
        const NDRec &R = E.rec[k];
        if (!(R.cw & CI_PIT_OUT)) continue;
        for (int32_t e = E.pit_off[R.cell].y; e < A.n_pit && A.pit_src[e] == R.cell; e++) {
            const int32_t kt = E.cid[A.pit_dst[e]] - 1;
            if (kt < 0) continue;
            const int32_t s = atomicAdd(count, 1);
            if (s < cap) { out[s].src = k; out[s].dst = kt; out[s].w = w_sorted[e]; }
        }
    }
}

// what the host build needs of a record (40 bytes instead of the record's 128: the copy is a third of the build)
struct CRecH { int32_t cell; uint32_t cw; int32_t out_id[2]; double out_w[2]; int32_t cnt, wid; };
__global__ void k_cond_extract(CIncArgs E, CRecH *out)
{
    for (int32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < E.nd; k += gridDim.x * blockDim.x) {
        const NDRec &R = E.rec[k];
        CRecH h;
        h.cell = R.cell; h.cw = R.cw; h.out_id[0] = R.out_id[0]; h.out_id[1] = R.out_id[1]; h.out_w[0] = R.out_w[0]; h.out_w[1] = R.out_w[1];
        h.cnt = R.cnt; h.wid = R.wid;
        out[k] = h;
    }
}

__global__ void k_cond_attach(CondArgsE X)
{
    for (int32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < X.nw; w += gridDim.x * blockDim.x) {
        NDRec &R = X.C.rec[X.node[w].rec];
        R.wid = w;
        R.cnt += W_BARRIER;
    }
}

// (PYDEM_COND_BUILD=check: the host build reads the records again after the device build attached its nodes)
__global__ void k_cond_detach(CondArgsE X)
{
    for (int32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < X.nw; w += gridDim.x * blockDim.x) {
        NDRec &R = X.C.rec[X.node[w].rec];
        R.wid = -1;
        R.cnt -= W_BARRIER;
    }
}
