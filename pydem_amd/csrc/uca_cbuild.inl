// uca_cbuild.inl -- included by uca.hip behind uca_cond.inl (inside its anonymous namespace).
//
// The condensed operator of uca_cond.inl built ON THE DEVICE.
//
// Reference: the fix-up worker pydem/process_manager.py:224-284 and the round pydem/dem_processing.py:778-862; what is
// built is the graph uca_cond.inl describes (edge p -> q between watched cells, weight = sum over the flow paths p -> q
// through unwatched cells of the products of the edge weights).  Until round 6 the build ran on ONE host core from a copy
// of the records (cond_build_host in uca.hip: 25-30 ms per 16384^2 tile -- adjacency 9, reverse sweep 12-20, nodes 3 -- the
// largest single item on the critical path of a wave-1 round, and 8 tiles of one process built one after the other).
// Here the records never leave HBM:
//
//   1. k_cb_count / k_cb_fill: per record its predecessors (the in-bits of the graph word + the pit -> drain edges that
//      end in it) and its pit out-edges as compact lists (two exclusive scans), the watched records listed and sorted by
//      cell (node order = ascending cell, like the host build);
//   2. k_cb_level: the records in reverse topological order, one launch per level over the whole chip (the levels are wide):
//      X(k) = "the watched cells the water of k reaches next, with the path weights", kept as
//      scale[k] * V(rep[k]) -- a record with ONE out-edge shares its target's vector, only where the flow splits two or
//      more SORTED vectors are merged into a new one (bump allocation in a pool; inputs staged in LDS so that a merge is
//      one round trip to the pool, not one per entry).  The sums are formed in the fixed order of the out-edges (first
//      target, second target, pit edges by drain cell): deterministic, equal to the host build's up to the order in which
//      THAT sorted the pit edges of one pit (by record id, which differs from run to run);
//   3. k_cb_nout .. k_cb_nodes: the watched records' vectors become the nodes' out-edges; in-slots in ascending source
//      order through one stable radix sort by target.
//
// The host build stays as the fall-back (a vector of more than CB_RUN entries, a pool region that overflows) and as the
// checker (PYDEM_COND_BUILD=check builds both and compares node by node; =host forces it).
// (hipcub is included at the top of uca.hip: this file sits inside its anonymous namespace)

constexpr int32_t CB_EMPTY = INT32_MIN;      // the empty vector: the water ends inside the tile
constexpr int32_t CB_CHAIN = INT32_MIN;      // bit 31 of a predecessor entry: that record has one out-edge
constexpr int CB_GRID = 8192;                // workgroups (= wavefronts = records in flight) of a level launch (PYDEM_CB_GRID)
// Every frontier is CB_NQ sub-queues and the pool CB_NQ regions, each with a counter on a line of its own: with one queue and one
// pool the 3000 wave-aggregated atomics of a 6000-record level queued up at two addresses of the fabric (~25 ns each: 90 us
// for the level, 13 ns per record at every width).  A workgroup consumes, fills and allocates from sub-queue / region blockIdx % CB_NQ.
constexpr int CB_NQ = 16, CB_PAD = 32;
constexpr int CB_LANES = 64;                 // a workgroup of the level kernel is one wavefront

struct CBVal { int32_t rep, vbeg, vn, wid; double scale; int32_t out_left, pad; };     // rep >= 0: owner record of the shared vector (vbeg, vn copied); < 0: unit vector of node -1 - rep
static_assert(sizeof(CBVal) == 32, "four records per line");
struct CBRec { int32_t pred_beg, pred_cnt, pit_beg, pit_cnt; };
struct CBEnt { int32_t node, pad; double w; };
struct CBPit { int32_t dst, pad; double w; };
enum { CBC_NW = 1, CBC_FAIL = 5, CBC_PROC = 6, CBC_LEVELS = 7, CBC_SLOW = 9, CBC_WORDS = 16 };

struct CBArgs {
    CIncArgs C;
    CBVal *rv; CBRec *ri;
    int32_t *pred_cnt, *pit_cnt, *pred_beg, *pit_beg;      // [nd + 1]: counts, their exclusive sums
    int32_t *pred; CBPit *pit;
    CBEnt *pool; int32_t pool_cap;       // pool_cap: entries of ONE of the CB_NQ regions of the pool
    int32_t *q0, *q1; int32_t qcap;      // CB_NQ sub-queues of qcap entries each
    int32_t *qcnt, *poolc;               // [3][CB_NQ][CB_PAD] rotating sub-queue sizes; [CB_NQ][CB_PAD] bump counters of the pool regions
    int32_t *wcell, *wrec, *wcell_s, *wrec_s; int32_t w_cap;
    int32_t *ctr;
    const double *w_sorted;              // pit weights in (src, dst) order (PitGraph::w)
    int32_t nw;
    int32_t max_chain;                   // links of a chain one lane finishes in passing
    int32_t *dbg;                        // PYDEM_CB_DEBUG: per level (frontier, largest merge, entries merged, deepest chain)
    int32_t level;
    int32_t *nout_c, *nout, *n_in, *in_first, *exc_in_c, *exc_in, *exc_out_c, *exc_out;      // [nw + 1]: counts / exclusive sums
    int32_t *e_dst, *e_q, *e_dst_s, *e_q_s, *e_slot; double *e_w;
    CNode *node; CEdge *edge;
};

// ---- 1. predecessor / pit lists ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cb_count(CBArgs B)
{
    const CIncArgs &E = B.C;
    const SweepArgs &A = E.G;
    const int m = A.m;
    if (blockIdx.x == 0 && threadIdx.x == 0) { B.pred_cnt[E.nd] = 0; B.pit_cnt[E.nd] = 0; }
    for (int32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < E.nd; k += gridDim.x * blockDim.x) {
        const NDRec &R = E.rec[k];
        const int32_t c = R.cell;
        const uint32_t cw = R.cw;
        int32_t np = 0, npit = 0;
#pragma unroll
        for (int d = 0; d < 8; d++)
            if (cw & (1u << d)) np += E.cid[c + NB_DI[d] * m + NB_DJ[d]] != 0;
        if (cw & CI_PIT_IN)
            for (int32_t e = E.pit_off[c].x; e < A.n_pit && A.pin_dst[e] == c; e++) np += E.cid[A.pin_src[e]] != 0;
        if (cw & CI_PIT_OUT)
            for (int32_t e = E.pit_off[c].y; e < A.n_pit && A.pit_src[e] == c; e++) npit += E.cid[A.pit_dst[e]] != 0;
        B.pred_cnt[k] = np; B.pit_cnt[k] = npit;
        CBVal v;
        v.rep = CB_EMPTY; v.vbeg = 0; v.vn = 0; v.wid = -1; v.scale = 0.0; v.pad = 0;
        v.out_left = (R.out_id[0] >= 0) + (R.out_id[1] >= 0) + npit;
        B.rv[k] = v;
        const int32_t outside = R.cnt - np;                                  // +1 while the cell is a 'todo' inlet (k_nd_link)
        if (outside != 0 && outside != 1) atomicOr(&B.ctr[CBC_FAIL], 2);
        if (R.wid == -2) {
            const int32_t s = agg_slot(&B.ctr[CBC_NW]);
            if (s < B.w_cap) { B.wcell[s] = c; B.wrec[s] = k; }
        }
    }
}

__global__ __launch_bounds__(256) void k_cb_fill(CBArgs B)
{
    const CIncArgs &E = B.C;
    const SweepArgs &A = E.G;
    const int m = A.m;
    for (int32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < E.nd; k += gridDim.x * blockDim.x) {
        const NDRec &R = E.rec[k];
        const int32_t c = R.cell;
        const uint32_t cw = R.cw;
        CBRec ri;
        ri.pred_beg = B.pred_beg[k]; ri.pred_cnt = B.pred_beg[k + 1] - ri.pred_beg;
        ri.pit_beg = B.pit_beg[k]; ri.pit_cnt = B.pit_beg[k + 1] - ri.pit_beg;
        B.ri[k] = ri;
        // (bit 31 of an entry: the predecessor has ONE out-edge -- this record is its only target, and whoever releases it can
        // finish it on the spot, cb_process)
        int32_t f = ri.pred_beg;
#pragma unroll
        for (int d = 0; d < 8; d++)
            if (cw & (1u << d)) { const int32_t p = E.cid[c + NB_DI[d] * m + NB_DJ[d]] - 1; if (p >= 0) B.pred[f++] = p | (B.rv[p].out_left == 1 ? CB_CHAIN : 0); }
        if (cw & CI_PIT_IN)
            for (int32_t e = E.pit_off[c].x; e < A.n_pit && A.pin_dst[e] == c; e++) { const int32_t p = E.cid[A.pin_src[e]] - 1; if (p >= 0) B.pred[f++] = p | (B.rv[p].out_left == 1 ? CB_CHAIN : 0); }
        f = ri.pit_beg;
        if (cw & CI_PIT_OUT)
            for (int32_t e = E.pit_off[c].y; e < A.n_pit && A.pit_src[e] == c; e++) {
                const int32_t kt = E.cid[A.pit_dst[e]] - 1;
                if (kt >= 0) { CBPit q; q.dst = kt; q.pad = 0; q.w = B.w_sorted[e]; B.pit[f++] = q; }
            }
        if ((R.out_id[0] >= 0) + (R.out_id[1] >= 0) + ri.pit_cnt == 0) {                                       // a sink: first level
            const int sq = blockIdx.x % CB_NQ;
            B.q0[(size_t)sq * B.qcap + agg_slot(&B.qcnt[sq * CB_PAD])] = k;
        }
    }
}

__global__ void k_cb_wid(CBArgs B)
{
    for (int32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < B.nw; w += gridDim.x * blockDim.x) B.rv[B.wrec_s[w]].wid = w;
}

// ---- 2. reverse topological sweep -----------------------------------------------------------------------------------
// One launch per level over the whole chip, ONE WAVEFRONT PER RECORD.  What was measured on the way (8 x 16384^2, 322 k records in
// 340-590 levels per tile; profiles/r06_cb_levels_16384.txt): one workgroup walking all levels 23-47 ms (one CU's memory
// parallelism against ~1000 records x 8 dependent accesses per level); one lane per record, 64 records per wavefront, one launch
// per level 12-20 ms -- the lanes of a wavefront run in lock step, so every wavefront paid for its longest merge and its longest
// chain, and ONE frontier counter + ONE pool counter took ~3000 wave-aggregated atomics per level at ~25 ns each; 16 sub-queues /
// pool regions and 1-4 records per wavefront 6.5-12.8 ms, a level now as long as its LARGEST merge walked by one lane (60-130
// entries, ~24 us).  Here the wavefront shares one record's work: the predecessors count down one per lane, the entries of the
// vectors to merge are fetched lane-strided into LDS (one round trip), two sorted runs are merged by RANK (every lane finds its
// element's place in the other run by binary search; equal nodes are summed, acc + f * entry like the host build's add_into,
// edge after edge), the result leaves lane-strided.
#ifndef PYDEM_CB_RUN
#define PYDEM_CB_RUN 512
#endif
constexpr int CB_RUN = PYDEM_CB_RUN;         // entries a merge may stage (inputs; two intermediate runs of the same size): 18 KB of LDS, 8 wavefronts per CU
                                             // (the largest merge of the 16384^2 bench mosaic stages 136; more than CB_RUN: the host build takes the tile)
struct CBWaveLds { int32_t id[3 * CB_RUN]; double v[3 * CB_RUN]; int32_t off[CB_LANES + 1]; };

__device__ __forceinline__ void cb_wave_sync() { __syncthreads(); }          // (a workgroup is one wavefront)

// merged = A (+) Bv, both sorted by node with distinct nodes inside a run; to LDS at x0 (pool == nullptr) or to the pool; returns the count
__device__ __forceinline__ int32_t cb_merge_wave(CBWaveLds &S, int32_t a0, int32_t na, int32_t b0, int32_t nb, int32_t x0, CBEnt *pool)
{
    const int lane = threadIdx.x;
    const unsigned long long lt = (1ull << lane) - 1ull;
    auto lower = [&](int32_t base, int32_t n, int32_t key) { int32_t lo = 0, hi = n; while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (S.id[base + mid] < key) lo = mid + 1; else hi = mid; } return lo; };
    auto emit = [&](int32_t pos, int32_t id, double v) {
        if (pool) { CBEnt t; t.node = id; t.pad = 0; t.w = v; pool[pos] = t; }
        else { S.id[x0 + pos] = id; S.v[x0 + pos] = v; }
    };
    int32_t dups = 0;
    for (int32_t base = 0; base < na; base += CB_LANES) {                       // the elements of A: their own index + the elements of B in front of them that are not their twins
        const int32_t i = base + lane;
        const bool valid = i < na;
        int32_t ida = 0, cb = 0; bool twin = false; double v = 0.0;
        if (valid) {
            ida = S.id[a0 + i]; v = S.v[a0 + i];
            cb = lower(b0, nb, ida);
            twin = cb < nb && S.id[b0 + cb] == ida;
            if (twin) v = v + S.v[b0 + cb];                                    // acc + f * entry
        }
        const unsigned long long tm = __ballot(twin);
        if (valid) emit(i + cb - (dups + __popcll(tm & lt)), ida, v);
        dups += __popcll(tm);
    }
    int32_t dups_b = 0;
    for (int32_t base = 0; base < nb; base += CB_LANES) {                       // the elements of B without a twin in A
        const int32_t j = base + lane;
        const bool valid = j < nb;
        int32_t idb = 0, ca = 0; bool twin = false;
        if (valid) { idb = S.id[b0 + j]; ca = lower(a0, na, idb); twin = ca < na && S.id[a0 + ca] == idb; }
        const unsigned long long tm = __ballot(twin);
        if (valid && !twin) emit(j + ca - (dups_b + __popcll(tm & lt)), idb, S.v[b0 + j]);
        dups_b += __popcll(tm);
    }
    return na + nb - dups;
}

// the predecessors of a record count down, one per lane; those this releases enter the next frontier, except ONE with a single
// out-edge (bit 31 of its entry), which is returned: the caller finishes it on the spot
__device__ __forceinline__ int32_t cb_countdown_wave(const CBArgs &B, const CBRec &rr, int32_t *qn, int32_t *cn)
{
    const int lane = threadIdx.x;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int32_t next = -1;
    for (int32_t base = 0; base < rr.pred_cnt; base += CB_LANES) {
        const int32_t e = base + lane;
        int32_t p = -1; bool released = false, chain = false;
        if (e < rr.pred_cnt) {
            const int32_t pe = B.pred[rr.pred_beg + e];
            p = pe & ~CB_CHAIN; chain = (pe & CB_CHAIN) != 0;
            released = atomicSub(&B.rv[p].out_left, 1) == 1;
        }
        const unsigned long long cm = __ballot(released && chain);
        int keep = -1;
        if (next < 0 && cm) { keep = __ffsll((long long)cm) - 1; next = __shfl(p, keep); }
        const bool topush = released && lane != keep;
        const unsigned long long pm = __ballot(topush);
        if (pm) {
            const int leader = __ffsll((long long)pm) - 1;
            int32_t slot0 = 0;
            if (lane == leader) slot0 = atomicAdd(cn, (int32_t)__popcll(pm));
            slot0 = __shfl(slot0, leader);
            if (topush) qn[slot0 + __popcll(pm & lt)] = p;
        }
    }
    return next;
}

__device__ __forceinline__ void cb_record_wave(const CBArgs &B, int32_t r, CBWaveLds &S, int32_t *poolc, int32_t pool_base, int32_t *qn, int32_t *cn, int32_t &n_chain)
{
    const int lane = threadIdx.x;
    const NDRec &R = B.C.rec[r];
    const int32_t o0 = R.out_id[0], o1 = R.out_id[1];
    const double w0 = R.out_w[0], w1 = R.out_w[1];
    const CBRec ri = B.ri[r];
    const int deg = (o0 >= 0) + (o1 >= 0) + ri.pit_cnt;
    // the count-downs first: whoever they release runs in a later launch and finds this record finished (see the file's head)
    int32_t next = cb_countdown_wave(B, ri, qn, cn);
    int32_t rep = CB_EMPTY, vbeg = 0, vn = 0;
    double scale = 0.0;
    auto edge = [&](int e, int32_t &tg, double &w) {                 // the e-th out-edge in the fixed order: first target, second target, pit edges by drain cell
        if (o0 >= 0) { if (e == 0) { tg = o0; w = w0; return; } e--; }
        if (o1 >= 0) { if (e == 0) { tg = o1; w = w1; return; } e--; }
        const CBPit p = B.pit[ri.pit_beg + e];
        tg = p.dst; w = p.w;
    };
    if (deg == 1) {
        int32_t tg; double w;
        edge(0, tg, w);
        const CBVal T = B.rv[tg];
        if (T.wid >= 0) { rep = -1 - T.wid; scale = w; }
        else { rep = T.rep; scale = w * T.scale; vbeg = T.vbeg; vn = T.vn; }
    } else if (deg > CB_LANES) {
        if (lane == 0) atomicOr(&B.ctr[CBC_FAIL], 1);
    } else if (deg >= 2) {
        // ---- the runs side by side in LDS, every product formed (f * entry, the host build's operands)
        int32_t o = 0;
        bool fits = true;
        for (int e = 0; e < deg; e++) {
            int32_t tg; double w;
            edge(e, tg, w);
            const CBVal T = B.rv[tg];
            if (lane == 0) S.off[e] = o;
            if (T.wid >= 0) { if (lane == 0 && o < CB_RUN) { S.id[o] = T.wid; S.v[o] = w; } o += 1; }                                 // (f * 1.0)
            else if (T.rep == CB_EMPTY) { }
            else if (T.rep < 0) { if (lane == 0 && o < CB_RUN) { S.id[o] = -1 - T.rep; S.v[o] = w * T.scale; } o += 1; }
            else {
                const double f = w * T.scale;
                if (o + T.vn <= CB_RUN)
                    for (int32_t q = lane; q < T.vn; q += CB_LANES) { const CBEnt t = B.pool[T.vbeg + q]; S.id[o + q] = t.node; S.v[o + q] = f * t.w; }
                o += T.vn;
            }
            if (o > CB_RUN) fits = false;
        }
        if (lane == 0) S.off[deg] = o;
        const int32_t total = o;
        if (B.dbg && lane == 0) { atomicMax(&B.dbg[4 * B.level + 1], total); atomicAdd(&B.dbg[4 * B.level + 2], total); }
        if (!fits) { if (lane == 0) { atomicOr(&B.ctr[CBC_FAIL], 1); atomicAdd(&B.ctr[CBC_SLOW], 1); } }       // (a vector of more than CB_RUN entries: the host build)
        else if (total > 0) {
            int32_t base_rel = 0;
            if (lane == 0) base_rel = atomicAdd(poolc, total);
            base_rel = __shfl(base_rel, 0);
            if ((int64_t)base_rel + total > (int64_t)B.pool_cap) { if (lane == 0) atomicOr(&B.ctr[CBC_FAIL], 1); }
            else {
                cb_wave_sync();
                // acc = run 0, then acc = merge(acc, run e) edge after edge; the last merge writes the pool
                int32_t a0 = S.off[0], na = S.off[1] - S.off[0], no = 0;
                bool flip = false;
                for (int e = 1; e < deg; e++) {
                    const int32_t b0 = S.off[e], nb = S.off[e + 1] - S.off[e];
                    const bool last = e == deg - 1;
                    const int32_t x0 = (flip ? 2 : 1) * CB_RUN;
                    const int32_t n = cb_merge_wave(S, a0, na, b0, nb, x0, last ? B.pool + pool_base + base_rel : nullptr);
                    cb_wave_sync();
                    if (last) no = n; else { a0 = x0; na = n; flip = !flip; }
                }
                if (no > 0) { rep = r; scale = 1.0; vbeg = pool_base + base_rel; vn = no; }
            }
        }
    }
    // ---- the record is finished; then the chain it released, link by link (each link: its vector is the finished record's, scaled)
    int32_t wid_r = B.rv[r].wid;
    int n_links = 0;
    for (;;) {
        if (lane == 0) { CBVal &V = B.rv[r]; V.rep = rep; V.vbeg = vbeg; V.vn = vn; V.scale = scale; }       // (wid / out_left stay)
        if (next < 0) break;
        if (n_links >= B.max_chain) {                                  // (the rest of a long chain enters the next level as ordinary records)
            if (lane == 0) qn[atomicAdd(cn, 1)] = next;
            break;
        }
        n_links++;
        const NDRec &Rn = B.C.rec[next];
        const CBRec rin = B.ri[next];
        const int32_t wid_n = B.rv[next].wid;
        const int32_t after = cb_countdown_wave(B, rin, qn, cn);
        double w;
        if (Rn.out_id[0] >= 0) w = Rn.out_w[0]; else if (Rn.out_id[1] >= 0) w = Rn.out_w[1]; else w = B.pit[rin.pit_beg].w;
        if (wid_r >= 0) { rep = -1 - wid_r; scale = w; vbeg = 0; vn = 0; }
        else scale = w * scale;                                        // (rep, vbeg, vn: r's)
        r = next; wid_r = wid_n; next = after;
        n_chain++;
    }
    if (B.dbg && lane == 0) atomicMax(&B.dbg[4 * B.level + 3], n_links);
    cb_wave_sync();                                                    // (the staging area is the next record's)
}

// level r: the frontier is the sub-queues of q[r & 1] (sizes qcnt[r % 3]), the next one goes to q[(r + 1) & 1]; gridDim.x is a multiple of CB_NQ
__global__ __launch_bounds__(CB_LANES) void k_cb_level(CBArgs B, int r)
{
    __shared__ CBWaveLds S;
    const int sq = blockIdx.x % CB_NQ, part = blockIdx.x / CB_NQ, nparts = gridDim.x / CB_NQ;
    const int32_t nq = B.qcnt[((r % 3) * CB_NQ + sq) * CB_PAD];
    // (the size of the level after the next is cleared by EVERY launch, also by those behind the end of the sweep: the counters
    // rotate, and a launch that returned without clearing would leave the size of level r - 1 where level r + 2 looks)
    if (part == 0 && threadIdx.x == 0) B.qcnt[(((r + 2) % 3) * CB_NQ + sq) * CB_PAD] = 0;
    if (nq == 0) return;
    if (part == 0 && threadIdx.x == 0) { atomicAdd(&B.ctr[CBC_PROC], nq); B.ctr[CBC_LEVELS] = r + 1; if (B.dbg) atomicAdd(&B.dbg[4 * r], nq); }
    const int32_t *qc = ((r & 1) ? B.q1 : B.q0) + (size_t)sq * B.qcap;
    int32_t *qn = ((r & 1) ? B.q0 : B.q1) + (size_t)sq * B.qcap;
    int32_t *cn = &B.qcnt[(((r + 1) % 3) * CB_NQ + sq) * CB_PAD];
    int32_t n_chain = 0;
    for (int32_t k = part; k < nq; k += nparts)
        cb_record_wave(B, qc[k], S, &B.poolc[sq * CB_PAD], sq * B.pool_cap, qn, cn, n_chain);
    if (n_chain && threadIdx.x == 0) atomicAdd(&B.ctr[CBC_PROC], n_chain);      // (records finished in passing never enter a frontier)
}

// ---- 3. nodes -----------------------------------------------------------------------------------------------------------
__global__ void k_cb_nout(CBArgs B)
{
    for (int32_t w = blockIdx.x * blockDim.x + threadIdx.x; w <= B.nw; w += gridDim.x * blockDim.x) {
        int32_t n = 0;
        if (w < B.nw) { const CBVal V = B.rv[B.wrec_s[w]]; n = V.rep == CB_EMPTY ? 0 : (V.rep < 0 ? 1 : V.vn); }
        B.nout_c[w] = n;          // (its exclusive sum `nout` = the first edge of node w)
        B.n_in[w] = 0;
    }
}

// all edges in source order (dst, weight) and the in-degrees; nout holds the scanned offsets by now
__global__ void k_cb_edges(CBArgs B)
{
    for (int32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < B.nw; w += gridDim.x * blockDim.x) {
        const CBVal V = B.rv[B.wrec_s[w]];
        int32_t q = B.nout[w];
        if (V.rep == CB_EMPTY) continue;
        if (V.rep < 0) { const int32_t d = -1 - V.rep; B.e_dst[q] = d; B.e_q[q] = q; B.e_w[q] = V.scale; atomicAdd(&B.n_in[d], 1); continue; }
        for (int32_t i = 0; i < V.vn; i++, q++) {
            const CBEnt t = B.pool[V.vbeg + i];
            B.e_dst[q] = t.node; B.e_q[q] = q; B.e_w[q] = V.scale * t.w;
            atomicAdd(&B.n_in[t.node], 1);
        }
    }
}

__global__ void k_cb_excess(CBArgs B)
{
    for (int32_t w = blockIdx.x * blockDim.x + threadIdx.x; w <= B.nw; w += gridDim.x * blockDim.x) {
        const int32_t ni = w < B.nw ? B.n_in[w] : 0, no = w < B.nw ? B.nout[w + 1] - B.nout[w] : 0;
        if (w == B.nw) B.n_in[w] = 0;
        B.exc_in_c[w] = ni > 2 ? ni - 2 : 0;
        B.exc_out_c[w] = no > 2 ? no - 2 : 0;
    }
}

// the edges sorted by target (stable: sources ascend within a target): an edge's in-slot is its rank among them
__global__ void k_cb_slots(CBArgs B, int32_t ne_all)
{
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ne_all; i += gridDim.x * blockDim.x)
        B.e_slot[B.e_q_s[i]] = i - B.in_first[B.e_dst_s[i]];
}

__global__ void k_cb_nodes(CBArgs B)
{
    for (int32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < B.nw; w += gridDim.x * blockDim.x) {
        const int32_t k = B.wrec_s[w];
        const NDRec &R = B.C.rec[k];
        CNode N;
        __builtin_memset(&N, 0, sizeof(N));
        N.rec = k; N.cell = R.cell; N.cw = R.cw;
        N.n_in = B.n_in[w]; N.n_out = B.nout[w + 1] - B.nout[w];
        N.in_base = B.exc_in[w]; N.out_base = B.exc_out[w];                  // (scanned)
        for (int e = 0; e < N.n_out; e++) {
            const int32_t q = B.nout[w] + e;
            CEdge ed;
            ed.dst = B.e_dst[q]; ed.w = B.e_w[q];
            const int32_t sl = B.e_slot[q];
            ed.slot = sl < 2 ? -1 - sl : B.exc_in[ed.dst] + sl - 2;
            if (e < 2) N.e_inl[e] = ed; else B.edge[N.out_base + e - 2] = ed;
        }
        N.cnt = N.n_in + (R.cnt - B.ri[k].pred_cnt);
        B.node[w] = N;
    }
}
