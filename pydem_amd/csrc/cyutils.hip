// cyutils.hip -- the reference's ONLY native boundary, kept callable as it is:
//   pydem/cyfuncs/cyutils.pyx:35-46   drain_connections(arr, ids, indptr, indices, set_to)
//   pydem/cyfuncs/cyutils.pyx:78-116  drain_area(area, done, ids, col_indptr, col_indices, col_data,
//                                                row_indptr, row_indices, n_rows, n_cols,
//                                                edge_todo, edge_todo_no_mask, skip_edge)
// over a GENERIC sparse graph in scipy CSC/CSR form (the DEMProcessor path above never builds that
// matrix; this shim exists so code written against the Cython module keeps working).  Semantics follow
// the Cython loops (:49-72, :119-187) round for round: level-synchronous PUSH, `done` marked before a
// round's pushes, a target that is already done and lies on the tile edge is skipped, a target becomes
// part of the next frontier when all its CSR upstream cells are done, and the loop ends when the
// frontier repeats itself (normally: is empty).  Differences: frontiers are lists (the Cython code
// rescans all N cells four times per round).  The additions into one target within a round keep the
// Cython order -- ascending source id -- without floating-point atomics: a round first collects its
// targets, then every target PULLS over its CSR row (scipy's tocsr keeps the column indices sorted)
// from the sources that are in the round's frontier; results are bit-identical run to run.
#include "internal.h"
#include <string.h>

namespace {

__device__ __forceinline__ bool cy_on_edge(int64_t id, int64_t n_rows, int64_t n_cols)   // cyutils.pyx:207-226
{
    return id < n_cols || id >= n_cols * n_rows - n_cols || id % n_cols == 0 || id % n_cols == n_cols - 1;
}

__global__ void k_cy_mark_done(const int32_t *__restrict__ front, int32_t nf, uint8_t *done, int32_t *mark, int32_t tag)
{
    for (int32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nf; q += gridDim.x * blockDim.x) {
        if (done) done[front[q]] = 1;      // :138-140
        mark[front[q]] = tag - 1;          // members of the current frontier (to recognise a repeating frontier)
    }
}

// a list slot for every lane that is active here, ONE atomic per wavefront: the list counters are single addresses, and
// one returning atomic per pushed node serialises in the L2 (~10 ns each)
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

// phase 1 of a round: the distinct targets of the frontier (cyutils.pyx:155-161)
__global__ void k_cy_targets(const int32_t *__restrict__ front, int32_t nf, const uint8_t *__restrict__ done,
                             const int32_t *__restrict__ col_indptr, const int32_t *__restrict__ col_indices, int64_t n_rows,
                             int64_t n_cols, int skip_edge, int32_t *tmark, int32_t tag, int32_t *targets, int32_t *n_targets)
{
    for (int32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nf; q += gridDim.x * blockDim.x) {
        const int32_t i = front[q];
        for (int32_t j = col_indptr[i]; j < col_indptr[i + 1]; j++) {
            const int32_t row = col_indices[j];
            if ((skip_edge || done[row]) && cy_on_edge(row, n_rows, n_cols)) continue;          // :159-161
            if (atomicExch(&tmark[row], tag) != tag) targets[agg_slot(n_targets)] = row;
        }
    }
}

// phase 2: every target adds what the frontier sends it, in ascending source order (:163-168), then checks
// whether all of its upstream cells are done (:173-179)
__global__ void k_cy_pull(const int32_t *__restrict__ targets, int32_t nt, double *area, const uint8_t *__restrict__ done,
                          const int32_t *__restrict__ col_indptr, const int32_t *__restrict__ col_indices,
                          const double *__restrict__ col_data, const int32_t *__restrict__ row_indptr,
                          const int32_t *__restrict__ row_indices, double *edge_todo, double *edge_todo_nm,
                          int32_t *mark, int32_t tag, int32_t *next, int32_t *n_next, int32_t *n_repeat)
{
    for (int32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nt; q += gridDim.x * blockDim.x) {
        const int32_t row = targets[q];
        double a = area[row];
        double t = edge_todo ? edge_todo[row] : 0.0, tn = edge_todo_nm ? edge_todo_nm[row] : 0.0;
        bool wait = false;
        for (int32_t k = row_indptr[row]; k < row_indptr[row + 1]; k++) {
            const int32_t src = row_indices[k];
            if (mark[src] == tag - 1) {                      // a member of this round's frontier: its push, :163
                for (int32_t j = col_indptr[src]; j < col_indptr[src + 1]; j++) {
                    if (col_indices[j] != row) continue;
                    const double w = col_data[j];
                    a += area[src] * w;
                    if (edge_todo) t += edge_todo[src] * w;
                    if (edge_todo_nm) tn += edge_todo_nm[src] * w;
                }
            }
            if (!done[src]) wait = true;
        }
        area[row] = a;
        if (edge_todo) edge_todo[row] = t;
        if (edge_todo_nm) edge_todo_nm[row] = tn;
        if (!wait) {
            const int32_t old = atomicExch(&mark[row], tag);
            if (old != tag) {
                next[agg_slot(n_next)] = row;
                if (old == tag - 1) atomicAdd(n_repeat, 1);                                      // was in the previous frontier too
            }
        }
    }
}

__global__ void k_cy_flood(const int32_t *__restrict__ front, int32_t nf, int32_t *arr, const int32_t *__restrict__ indptr,
                           const int32_t *__restrict__ indices, int32_t set_to, int32_t *mark, int32_t tag, int32_t *next,
                           int32_t *n_next, int32_t *n_repeat)
{
    for (int32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nf; q += gridDim.x * blockDim.x) {
        const int32_t i = front[q];
        for (int32_t j = indptr[i]; j < indptr[i + 1]; j++) {
            const int32_t row = indices[j];
            if (atomicExch(&arr[row], set_to) != set_to) {                                       // :69-70
                const int32_t old = atomicExch(&mark[row], tag);
                if (old != tag) { next[agg_slot(n_next)] = row; if (old == tag - 1) atomicAdd(n_repeat, 1); }
            }
        }
    }
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) { if (bytes == 0) bytes = 8; HIP_TRY(hipMalloc(&p, bytes)); return 0; }
    template <typename T> T *as() { return (T *)p; }
};

int front_from_mask(const uint8_t *mask, int64_t n, std::vector<int32_t> &out)
{
    out.clear();
    for (int64_t i = 0; i < n; i++) if (mask[i]) out.push_back((int32_t)i);
    return 0;
}

}  // namespace

extern "C" {

int pydem_drain_area(double *area, uint8_t *done, uint8_t *ids, const int32_t *col_indptr, const int32_t *col_indices,
                     const double *col_data, const int32_t *row_indptr, const int32_t *row_indices, int64_t n_rows,
                     int64_t n_cols, double *edge_todo, double *edge_todo_no_mask, int skip_edge, int device)
{
    HIP_TRY(hipSetDevice(device));
    const int64_t N = n_rows * n_cols;
    if (N >= INT32_MAX) { pydem_set_error("graph too large for int32 ids"); return -2; }
    const int64_t nnz = col_indptr[N];
    DevBuf d_area, d_done, d_cp, d_ci, d_cd, d_rp, d_ri, d_et, d_etn, d_mark, d_q0, d_q1, d_cnt, d_tmark, d_targets;
    PYDEM_TRY(d_area.alloc(N * 8)); PYDEM_TRY(d_done.alloc(N)); PYDEM_TRY(d_cp.alloc((N + 1) * 4)); PYDEM_TRY(d_ci.alloc(nnz * 4));
    PYDEM_TRY(d_cd.alloc(nnz * 8)); PYDEM_TRY(d_rp.alloc((N + 1) * 4)); PYDEM_TRY(d_ri.alloc(nnz * 4));
    PYDEM_TRY(d_mark.alloc(N * 4)); PYDEM_TRY(d_q0.alloc(N * 4)); PYDEM_TRY(d_q1.alloc(N * 4)); PYDEM_TRY(d_cnt.alloc(16));
    PYDEM_TRY(d_tmark.alloc(N * 4)); PYDEM_TRY(d_targets.alloc(N * 4));
    if (edge_todo) PYDEM_TRY(d_et.alloc(N * 8));
    if (edge_todo_no_mask) PYDEM_TRY(d_etn.alloc(N * 8));
    HIP_TRY(hipMemcpy(d_area.p, area, N * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_done.p, done, N, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_cp.p, col_indptr, (N + 1) * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_ci.p, col_indices, nnz * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_cd.p, col_data, nnz * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_rp.p, row_indptr, (N + 1) * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_ri.p, row_indices, nnz * 4, hipMemcpyHostToDevice));
    if (edge_todo) HIP_TRY(hipMemcpy(d_et.p, edge_todo, N * 8, hipMemcpyHostToDevice));
    if (edge_todo_no_mask) HIP_TRY(hipMemcpy(d_etn.p, edge_todo_no_mask, N * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(d_mark.p, 0, N * 4));
    HIP_TRY(hipMemset(d_tmark.p, 0, N * 4));
    std::vector<int32_t> front;
    front_from_mask(ids, N, front);
    int32_t nf = (int32_t)front.size();
    HIP_TRY(hipMemcpy(d_q0.p, front.data(), (size_t)nf * 4, hipMemcpyHostToDevice));
    int32_t *q[2] = {d_q0.as<int32_t>(), d_q1.as<int32_t>()};
    int32_t tag = 2;      // marks: tag of the round in which a cell entered a frontier
    int cur = 0;
    // (the initial frontier carries tag 1 so that "same frontier again" can be recognised in round 1)
    for (int64_t r = 0;; r++) {
        if (nf > 0) {
            const int g = (int)(cdiv(nf, 256) < 2048 ? cdiv(nf, 256) : 2048);
            hipLaunchKernelGGL(k_cy_mark_done, dim3(g), dim3(256), 0, 0, q[cur], nf, d_done.as<uint8_t>(), d_mark.as<int32_t>(), tag);
            HIP_TRY(hipMemset(d_cnt.p, 0, 16));
            hipLaunchKernelGGL(k_cy_targets, dim3(g), dim3(256), 0, 0, q[cur], nf, d_done.as<uint8_t>(), d_cp.as<int32_t>(), d_ci.as<int32_t>(),
                               n_rows, n_cols, skip_edge, d_tmark.as<int32_t>(), tag, d_targets.as<int32_t>(), d_cnt.as<int32_t>() + 2);
            int32_t nt = 0;
            HIP_TRY(hipMemcpy(&nt, d_cnt.as<int32_t>() + 2, 4, hipMemcpyDeviceToHost));
            if (nt > 0) {
                const int gt = (int)(cdiv(nt, 256) < 2048 ? cdiv(nt, 256) : 2048);
                hipLaunchKernelGGL(k_cy_pull, dim3(gt), dim3(256), 0, 0, d_targets.as<int32_t>(), nt, d_area.as<double>(), d_done.as<uint8_t>(),
                                   d_cp.as<int32_t>(), d_ci.as<int32_t>(), d_cd.as<double>(), d_rp.as<int32_t>(), d_ri.as<int32_t>(),
                                   edge_todo ? d_et.as<double>() : nullptr, edge_todo_no_mask ? d_etn.as<double>() : nullptr,
                                   d_mark.as<int32_t>(), tag, q[1 - cur], d_cnt.as<int32_t>(), d_cnt.as<int32_t>() + 1);
            }
            HIP_TRY(hipGetLastError());
        } else {
            HIP_TRY(hipMemset(d_cnt.p, 0, 16));
        }
        int32_t h[2];
        HIP_TRY(hipMemcpy(h, d_cnt.p, 8, hipMemcpyDeviceToHost));
        const int32_t n_next = h[0], n_rep = h[1];
        // keep_going = (ids != ids_old)  (:187): stop when the new frontier equals the old one
        const bool same = (n_next == nf) && (n_rep == n_next);
        cur = 1 - cur; nf = n_next; tag++;
        if (same) break;
        if (r > (1ll << 31)) { pydem_set_error("drain_area did not terminate"); return -5; }
    }
    HIP_TRY(hipMemcpy(area, d_area.p, N * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(done, d_done.p, N, hipMemcpyDeviceToHost));
    if (edge_todo) HIP_TRY(hipMemcpy(edge_todo, d_et.p, N * 8, hipMemcpyDeviceToHost));
    if (edge_todo_no_mask) HIP_TRY(hipMemcpy(edge_todo_no_mask, d_etn.p, N * 8, hipMemcpyDeviceToHost));
    // final frontier back into ids (callers overwrite it, dem_processing.py:962)
    memset(ids, 0, (size_t)N);
    if (nf > 0) {
        front.resize((size_t)nf);
        HIP_TRY(hipMemcpy(front.data(), q[cur], (size_t)nf * 4, hipMemcpyDeviceToHost));
        for (int32_t v : front) ids[v] = 1;
    }
    return 0;
}

int pydem_drain_connections(uint8_t *arr, uint8_t *ids, const int32_t *indptr, const int32_t *indices, int64_t n,
                            uint8_t set_to, int device)
{
    HIP_TRY(hipSetDevice(device));
    if (n >= INT32_MAX) { pydem_set_error("graph too large for int32 ids"); return -2; }
    const int64_t nnz = indptr[n];
    DevBuf d_arr, d_p, d_i, d_mark, d_q0, d_q1, d_cnt;
    PYDEM_TRY(d_arr.alloc(n * 4)); PYDEM_TRY(d_p.alloc((n + 1) * 4)); PYDEM_TRY(d_i.alloc(nnz * 4)); PYDEM_TRY(d_mark.alloc(n * 4));
    PYDEM_TRY(d_q0.alloc(n * 4)); PYDEM_TRY(d_q1.alloc(n * 4)); PYDEM_TRY(d_cnt.alloc(16));
    std::vector<int32_t> a32((size_t)n);
    for (int64_t k = 0; k < n; k++) a32[(size_t)k] = arr[k];
    HIP_TRY(hipMemcpy(d_arr.p, a32.data(), n * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_p.p, indptr, (n + 1) * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_i.p, indices, nnz * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(d_mark.p, 0, n * 4));
    std::vector<int32_t> front;
    front_from_mask(ids, n, front);
    int32_t nf = (int32_t)front.size();
    HIP_TRY(hipMemcpy(d_q0.p, front.data(), (size_t)nf * 4, hipMemcpyHostToDevice));
    int32_t *q[2] = {d_q0.as<int32_t>(), d_q1.as<int32_t>()};
    int32_t tag = 2;
    int cur = 0;
    for (;;) {
        HIP_TRY(hipMemset(d_cnt.p, 0, 16));
        if (nf > 0) {
            const int g = (int)(cdiv(nf, 256) < 2048 ? cdiv(nf, 256) : 2048);
            hipLaunchKernelGGL(k_cy_mark_done, dim3(g), dim3(256), 0, 0, q[cur], nf, (uint8_t *)nullptr, d_mark.as<int32_t>(), tag);
            hipLaunchKernelGGL(k_cy_flood, dim3(g), dim3(256), 0, 0, q[cur], nf, d_arr.as<int32_t>(), d_p.as<int32_t>(),
                               d_i.as<int32_t>(), (int32_t)set_to, d_mark.as<int32_t>(), tag, q[1 - cur], d_cnt.as<int32_t>(),
                               d_cnt.as<int32_t>() + 1);
            HIP_TRY(hipGetLastError());
        }
        int32_t h[2];
        HIP_TRY(hipMemcpy(h, d_cnt.p, 8, hipMemcpyDeviceToHost));
        const bool same = (h[0] == nf) && (h[1] == h[0]);
        cur = 1 - cur; nf = h[0]; tag++;
        if (same) break;
    }
    HIP_TRY(hipMemcpy(a32.data(), d_arr.p, n * 4, hipMemcpyDeviceToHost));
    for (int64_t k = 0; k < n; k++) arr[k] = (uint8_t)a32[(size_t)k];
    memset(ids, 0, (size_t)n);
    return 0;
}

}  // extern "C"
