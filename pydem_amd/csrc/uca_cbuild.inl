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
// The host build stays as the fall-back (a record with more than CB_MAXD out-edges, a pool that overflows) and as the
// checker (PYDEM_COND_BUILD=check builds both and compares node by node; =host forces it).
// (hipcub is included at the top of uca.hip: this file sits inside its anonymous namespace)

constexpr int32_t CB_EMPTY = INT32_MIN;      // the empty vector: the water ends inside the tile
constexpr int32_t CB_CHAIN = INT32_MIN;      // bit 31 of a predecessor entry: that record has one out-edge
constexpr int CB_MAXD = 16;                  // out-edges of a record the device merge handles (2 + pit edges)
constexpr int CB_GRID = 256;                 // workgroups of a level launch (grid-stride over the frontier)
constexpr int CB_LANES = 64;                 // a workgroup of the level kernel is one wavefront
constexpr int CB_POOL_LEVEL = 2560;          // its LDS staging for the inputs of the lanes' merges (entries, 30 KB)
constexpr int CB_POOL_SWEEP = 4700;          // ... of the one-workgroup kernel (56 KB)

struct CBVal { int32_t rep, vbeg, vn, wid; double scale; int32_t out_left, pad; };     // rep >= 0: owner record of the shared vector (vbeg, vn copied); < 0: unit vector of node -1 - rep
static_assert(sizeof(CBVal) == 32, "four records per line");
struct CBRec { int32_t pred_beg, pred_cnt, pit_beg, pit_cnt; };
struct CBEnt { int32_t node, pad; double w; };
struct CBPit { int32_t dst, pad; double w; };
enum { CBC_NW = 1, CBC_POOL = 4, CBC_FAIL = 5, CBC_PROC = 6, CBC_LEVELS = 7, CBC_SLOW = 9, CBC_Q = 10 /* .. 12: rotating frontier sizes */, CBC_WORDS = 16 };

struct CBArgs {
    CIncArgs C;
    CBVal *rv; CBRec *ri;
    int32_t *pred_cnt, *pit_cnt, *pred_beg, *pit_beg;      // [nd + 1]: counts, their exclusive sums
    int32_t *pred; CBPit *pit;
    CBEnt *pool; int32_t pool_cap;
    int32_t *q0, *q1;
    int32_t *wcell, *wrec, *wcell_s, *wrec_s; int32_t w_cap;
    int32_t *ctr;
    const double *w_sorted;              // pit weights in (src, dst) order (PitGraph::w)
    int32_t nw;
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
        if ((R.out_id[0] >= 0) + (R.out_id[1] >= 0) + ri.pit_cnt == 0) B.q0[agg_slot(&B.ctr[CBC_Q])] = k;      // a sink: first level
    }
}

__global__ void k_cb_wid(CBArgs B)
{
    for (int32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < B.nw; w += gridDim.x * blockDim.x) B.rv[B.wrec_s[w]].wid = w;
}

// ---- 2. reverse topological sweep -----------------------------------------------------------------------------------
// One launch per level over the whole chip: the levels are WIDE (322 k records in 450-850 levels at 16384^2, thousands in the
// first ones) -- one workgroup walking them (the first build) took 23-47 ms, longer than the host.  A workgroup is ONE
// wavefront with an LDS staging area for the inputs of its lanes' merges: the entries of the vectors a lane merges are fetched
// in one round trip (independent loads), the products formed, and the merge of the sorted runs reads LDS; only when the area
// is full does a merge walk the pool entry by entry.
template <int N> struct CBStageLds { static constexpr int CAP = N; int32_t id[N]; double v[N]; int32_t used; };

// returns false when the record has to wait for room in the staging area (DEFER: the caller runs it again after the level's
// other records; nothing has been allocated or written by then)
template <bool DEFER, typename Stage, typename Push>
__device__ __forceinline__ bool cb_process(const CBArgs &B, int32_t r, Stage &S, Push push, int32_t &n_chain)
{
    constexpr int CB_POOL_LDS = Stage::CAP;
    const NDRec &R = B.C.rec[r];
    const int32_t o0 = R.out_id[0], o1 = R.out_id[1];
    const double w0 = R.out_w[0], w1 = R.out_w[1];
    CBRec ri = B.ri[r];
    const int deg = (o0 >= 0) + (o1 >= 0) + ri.pit_cnt;
    int32_t rep = CB_EMPTY, vbeg = 0, vn = 0;
    double scale = 0.0;
    auto edge = [&](int e, int32_t &tg, double &w) {                 // the e-th out-edge in the fixed order
        if (o0 >= 0) { if (e == 0) { tg = o0; w = w0; return; } e--; }
        if (o1 >= 0) { if (e == 0) { tg = o1; w = w1; return; } e--; }
        const CBPit p = B.pit[ri.pit_beg + e];
        tg = p.dst; w = p.w;
    };
    bool generic = deg > 4 && deg <= CB_MAXD;
    if (deg == 1) {
        int32_t tg; double w;
        edge(0, tg, w);
        const CBVal T = B.rv[tg];
        if (T.wid >= 0) { rep = -1 - T.wid; scale = w; }
        else { rep = T.rep; scale = w * T.scale; vbeg = T.vbeg; vn = T.vn; }
    } else if (deg > CB_MAXD) {
        atomicOr(&B.ctr[CBC_FAIL], 1);
    } else if (deg >= 2 && deg <= 4) {
        // The common case, entirely in registers (arrays indexed by unrolled constants; the generic path below keeps its cursors
        // in scratch memory, a memory round trip per access: the first build spent 40 us per level there).  The host build's
        // own order: acc = f0 * V0, then acc = merge(acc, f_e * V_e) edge after edge, sums formed as acc + f_e * entry.
        int32_t eb[4], en[4];                                        // eb < 0: the unit vector of node -1 - eb
        double ef[4];
        int total = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            eb[e] = 0; en[e] = 0; ef[e] = 0.0;
            if (e < deg) {
                int32_t tg; double w;
                edge(e, tg, w);
                const CBVal T = B.rv[tg];
                if (T.wid >= 0) { eb[e] = -1 - T.wid; en[e] = 1; ef[e] = w; }
                else if (T.rep == CB_EMPTY) { }
                else if (T.rep < 0) { eb[e] = T.rep; en[e] = 1; ef[e] = w * T.scale; }
                else { eb[e] = T.vbeg; en[e] = T.vn; ef[e] = w * T.scale; }
                total += en[e];
            }
        }
        const int need = deg == 2 ? total : 3 * total;               // the inputs (+ two intermediate runs when there is more than one merge)
        const int32_t so = total == 0 ? 0 : (need <= CB_POOL_LDS ? atomicAdd(&S.used, need) : CB_POOL_LDS);
        if (total > 0 && so + need > CB_POOL_LDS) {                  // no room in the staging area
            if (DEFER && need <= CB_POOL_LDS) return false;          // ... this time
            generic = true;                                          // ... ever: the generic path, from the pool
        } else if (total > 0) {
            const int32_t base = atomicAdd(&B.ctr[CBC_POOL], total);
            if ((int64_t)base + total > (int64_t)B.pool_cap) atomicOr(&B.ctr[CBC_FAIL], 1);
            else {
                // the inputs side by side in LDS, every product formed (one round trip to the pool: the loads are independent)
                int32_t rb[5];
                int o = so;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    rb[e] = o;
                    if (en[e] == 1 && eb[e] < 0) { S.id[o] = -1 - eb[e]; S.v[o] = ef[e]; o++; }       // (f * 1.0)
                    else for (int q = 0; q < en[e]; q++, o++) { const CBEnt t = B.pool[eb[e] + q]; S.id[o] = t.node; S.v[o] = ef[e] * t.w; }
                }
                rb[4] = o;
                int32_t a0 = rb[0], a1 = rb[1];
                int32_t no = 0;
                bool flip = false;
#pragma unroll
                for (int e = 1; e < 4; e++) {
                    if (e < deg) {
                        const bool last = e == deg - 1;
                        const int32_t ob = so + (flip ? 2 : 1) * total;
                        int32_t oo = ob;
                        int32_t i = a0, j = rb[e];
                        const int32_t j1 = rb[e + 1];
                        int32_t ida = i < a1 ? S.id[i] : INT32_MAX, idb = j < j1 ? S.id[j] : INT32_MAX;
                        while (ida != INT32_MAX || idb != INT32_MAX) {
                            int32_t id; double v;
                            if (ida < idb) { id = ida; v = S.v[i]; i++; ida = i < a1 ? S.id[i] : INT32_MAX; }
                            else if (idb < ida) { id = idb; v = S.v[j]; j++; idb = j < j1 ? S.id[j] : INT32_MAX; }
                            else { id = ida; v = S.v[i] + S.v[j]; i++; j++; ida = i < a1 ? S.id[i] : INT32_MAX; idb = j < j1 ? S.id[j] : INT32_MAX; }
                            if (last) { CBEnt t; t.node = id; t.pad = 0; t.w = v; B.pool[base + no++] = t; }
                            else { S.id[oo] = id; S.v[oo] = v; oo++; }
                        }
                        if (!last) { a0 = ob; a1 = oo; flip = !flip; }
                    }
                }
                if (no > 0) { rep = r; scale = 1.0; vbeg = base; vn = no; }
            }
        }
    }
    if (generic) {
        int32_t eb[CB_MAXD], en[CB_MAXD];                            // eb < 0: the unit vector of node -1 - eb
        double ef[CB_MAXD];
        int total = 0;
        for (int e = 0; e < deg; e++) {
            int32_t tg; double w;
            edge(e, tg, w);
            const CBVal T = B.rv[tg];
            if (T.wid >= 0) { eb[e] = -1 - T.wid; en[e] = 1; ef[e] = w; }
            else if (T.rep == CB_EMPTY) { eb[e] = 0; en[e] = 0; ef[e] = 0.0; }
            else if (T.rep < 0) { eb[e] = T.rep; en[e] = 1; ef[e] = w * T.scale; }
            else { eb[e] = T.vbeg; en[e] = T.vn; ef[e] = w * T.scale; }
            total += en[e];
        }
        if (total > 0) {
            const int32_t base = atomicAdd(&B.ctr[CBC_POOL], total);
            if ((int64_t)base + total > (int64_t)B.pool_cap) atomicOr(&B.ctr[CBC_FAIL], 1);
            else {
                int32_t no = 0;
                // room in the workgroup's LDS staging (bump allocation per launch: few lanes of a wavefront merge, so one of them
                // may take hundreds of entries)
                const int32_t so = total <= CB_POOL_LDS ? atomicAdd(&S.used, total) : CB_POOL_LDS;
                if (so + total <= CB_POOL_LDS) {
                    // the inputs side by side in LDS, every product formed (f * entry, the host build's operands), then a merge of
                    // the deg sorted runs
                    int32_t off[CB_MAXD + 1];
                    int o = so;
                    for (int e = 0; e < deg; e++) {
                        off[e] = o;
                        if (en[e] == 1 && eb[e] < 0) { S.id[o] = -1 - eb[e]; S.v[o] = ef[e]; o++; }       // (f * 1.0)
                        else for (int q = 0; q < en[e]; q++, o++) { const CBEnt t = B.pool[eb[e] + q]; S.id[o] = t.node; S.v[o] = ef[e] * t.w; }
                    }
                    off[deg] = o;
                    int32_t cur[CB_MAXD];
                    for (int e = 0; e < deg; e++) cur[e] = off[e];
                    for (;;) {
                        int32_t mn = INT32_MAX;
                        for (int e = 0; e < deg; e++) if (cur[e] < off[e + 1]) { const int32_t id = S.id[cur[e]]; mn = id < mn ? id : mn; }
                        if (mn == INT32_MAX) break;
                        double acc = 0.0; bool first = true;
                        for (int e = 0; e < deg; e++)
                            if (cur[e] < off[e + 1] && S.id[cur[e]] == mn) { const double v = S.v[cur[e]]; acc = first ? v : acc + v; first = false; cur[e]++; }
                        CBEnt t; t.node = mn; t.pad = 0; t.w = acc;
                        B.pool[base + no++] = t;
                    }
                } else {
                    // no room: straight from the pool, one head per run
                    atomicAdd(&B.ctr[CBC_SLOW], 1);
                    int32_t cur[CB_MAXD], hid[CB_MAXD];
                    double hv[CB_MAXD];
                    auto head = [&](int e) {
                        if (cur[e] >= en[e]) { hid[e] = INT32_MAX; return; }
                        if (eb[e] < 0) { hid[e] = -1 - eb[e]; hv[e] = ef[e]; }
                        else { const CBEnt t = B.pool[eb[e] + cur[e]]; hid[e] = t.node; hv[e] = ef[e] * t.w; }
                    };
                    for (int e = 0; e < deg; e++) { cur[e] = 0; head(e); }
                    for (;;) {
                        int32_t mn = INT32_MAX;
                        for (int e = 0; e < deg; e++) mn = hid[e] < mn ? hid[e] : mn;
                        if (mn == INT32_MAX) break;
                        double acc = 0.0; bool first = true;
                        for (int e = 0; e < deg; e++)
                            if (hid[e] == mn) { acc = first ? hv[e] : acc + hv[e]; first = false; cur[e]++; head(e); }
                        CBEnt t; t.node = mn; t.pad = 0; t.w = acc;
                        B.pool[base + no++] = t;
                    }
                }
                if (no > 0) { rep = r; scale = 1.0; vbeg = base; vn = no; }
            }
        }
    }
    // ---- the record is finished; its predecessors count down.  A predecessor with ONE out-edge that this releases is finished
    // on the spot (its vector is this record's, scaled: no merge, nothing to wait for) and the walk goes on from there -- a
    // chain of such records costs one dependent load per link instead of one launch per link.
    int32_t wid_r = B.rv[r].wid;
    for (;;) {
        CBVal &V = B.rv[r];
        V.rep = rep; V.vbeg = vbeg; V.vn = vn; V.scale = scale;       // (wid / out_left stay)
        int32_t next = -1;
        for (int32_t e = 0; e < ri.pred_cnt; e++) {
            const int32_t pe = B.pred[ri.pred_beg + e];
            const int32_t p = pe & ~CB_CHAIN;
            if (atomicSub(&B.rv[p].out_left, 1) == 1) {
                if ((pe & CB_CHAIN) && next < 0) next = p;
                else push(p);
            }
        }
        if (next < 0) break;
        // `next` has one out-edge, and it ends in r: through(r, w)
        const NDRec &Rn = B.C.rec[next];
        const CBRec rin = B.ri[next];
        const int32_t wid_n = B.rv[next].wid;
        double w;
        if (Rn.out_id[0] >= 0) w = Rn.out_w[0]; else if (Rn.out_id[1] >= 0) w = Rn.out_w[1]; else w = B.pit[rin.pit_beg].w;
        if (wid_r >= 0) { rep = -1 - wid_r; scale = w; vbeg = 0; vn = 0; }
        else scale = w * scale;                                        // (rep, vbeg, vn: r's)
        r = next; ri = rin; wid_r = wid_n;
        n_chain++;
    }
    return true;
}

// level r: the frontier is q[r & 1][0 .. ctr[CBC_Q + r % 3]), the next one goes to q[(r + 1) & 1]
__global__ __launch_bounds__(CB_LANES) void k_cb_level(CBArgs B, int r)
{
    __shared__ CBStageLds<CB_POOL_LEVEL> S;
    if (threadIdx.x == 0) S.used = 0;
    __syncthreads();
    const int32_t nq = B.ctr[CBC_Q + r % 3];
    // (the size of the level after the next is cleared by EVERY launch, also by those behind the end of the sweep: the counters
    // rotate, and a launch that returned without clearing would leave the size of level r - 1 where level r + 2 looks)
    if (blockIdx.x == 0 && threadIdx.x == 0) B.ctr[CBC_Q + (r + 2) % 3] = 0;
    if (nq == 0) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) { atomicAdd(&B.ctr[CBC_PROC], nq); B.ctr[CBC_LEVELS] = r + 1; }
    const int32_t *qc = (r & 1) ? B.q1 : B.q0;
    int32_t *qn = (r & 1) ? B.q0 : B.q1;
    int32_t *cn = &B.ctr[CBC_Q + (r + 1) % 3];
    int32_t n_chain = 0;
    auto push = [&](int32_t p) { qn[agg_slot(cn)] = p; };
    for (int32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < nq; k += gridDim.x * blockDim.x)
        (void)cb_process<false>(B, qc[k], S, push, n_chain);
    if (n_chain) atomicAdd(&B.ctr[CBC_PROC], n_chain);               // (records finished in passing never enter a frontier)
}

// The narrow remainder: ONE workgroup, level after level without a launch in between (the cascade's own pattern: level queues in
// LDS, one barrier per level).  A launch boundary leaves every XCD's L2 cold for what the others wrote -- each of a level's six
// or seven dependent round trips then goes to the fabric (measured: 30 us per level with one launch per level, whatever its
// width) -- while one CU keeps reading its own XCD's L2.  Starts at level r0 with the frontier in q[r0 & 1] / ctr[CBC_Q + r0 % 3].
constexpr int CB_QCAP = 1024;
__global__ __launch_bounds__(1024) void k_cb_sweep(CBArgs B, int r0)
{
    __shared__ int32_t s_q[2][CB_QCAP];
    __shared__ CBStageLds<CB_POOL_SWEEP> S;
    __shared__ int32_t s_cnt[3], s_proc, s_retry[2];
    int r = r0;
    if (threadIdx.x == 0) { s_cnt[r % 3] = B.ctr[CBC_Q + r % 3]; s_cnt[(r + 1) % 3] = 0; s_cnt[(r + 2) % 3] = 0; S.used = 0; s_proc = 0; s_retry[0] = 0; s_retry[1] = 0; }
    __syncthreads();
    int32_t nq = s_cnt[r % 3];
    {
        const int32_t *q = (r & 1) ? B.q1 : B.q0;
        for (int32_t k = threadIdx.x; k < nq && k < CB_QCAP; k += blockDim.x) s_q[r & 1][k] = q[k];
    }
    __syncthreads();
    int32_t n_chain = 0;
    while (nq > 0) {
        const int32_t *qc = (r & 1) ? B.q1 : B.q0;
        int32_t *qn = (r & 1) ? B.q0 : B.q1;
        const int32_t *lc = s_q[r & 1];
        int32_t *ln = s_q[(r + 1) & 1];
        int32_t *cn = &s_cnt[(r + 1) % 3];
        if (threadIdx.x == 0) { s_cnt[(r + 2) % 3] = 0; s_proc += nq; }
        auto push = [&](int32_t p) { const int32_t sl = agg_slot(cn); if (sl < CB_QCAP) ln[sl] = p; else qn[sl] = p; };
        // records whose merge found no room in the staging area wait in a list (behind the frontier queues' nd entries: q0 / q1
        // have 2 * nd) and run again, with the area handed out anew, after the others
        int32_t *rl[2] = {B.q0 + B.C.nd, B.q1 + B.C.nd};
        for (int32_t k = threadIdx.x; k < nq; k += blockDim.x) {
            const int32_t rec = k < CB_QCAP ? lc[k] : qc[k];
            if (!cb_process<true>(B, rec, S, push, n_chain)) rl[0][agg_slot(&s_retry[0])] = rec;
        }
        __syncthreads();
        int pass = 0;
        while (s_retry[pass & 1] > 0) {                               // (uniform: read between two barriers)
            const int32_t nr = s_retry[pass & 1];
            __syncthreads();
            if (threadIdx.x == 0) { S.used = 0; s_retry[(pass + 1) & 1] = 0; }
            __syncthreads();
            for (int32_t k = threadIdx.x; k < nr; k += blockDim.x) {
                const int32_t rec = rl[pass & 1][k];
                if (!cb_process<true>(B, rec, S, push, n_chain)) rl[(pass + 1) & 1][agg_slot(&s_retry[(pass + 1) & 1])] = rec;
            }
            __syncthreads();
            pass++;
        }
        nq = *cn;
        r++;
        // (the staging area is handed out anew every level; its users of this level are past the barrier)
        __syncthreads();
        if (threadIdx.x == 0) { S.used = 0; s_retry[0] = 0; s_retry[1] = 0; }
        __syncthreads();
    }
    if (n_chain) atomicAdd(&B.ctr[CBC_PROC], n_chain);
    if (threadIdx.x == 0) { atomicAdd(&B.ctr[CBC_PROC], s_proc); B.ctr[CBC_LEVELS] = r; B.ctr[CBC_Q] = 0; B.ctr[CBC_Q + 1] = 0; B.ctr[CBC_Q + 2] = 0; }
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
