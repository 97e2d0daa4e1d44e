// flats.hip -- K2: flats mask with its one-pixel downstream extension.
//
// Replaces _find_flats_edges (reference pydem/dem_processing.py:657-680, helper
// utils.get_adjacent_index pydem/utils.py:270-311) and the epilogue of calc_slopes_directions
// (:610-613).  The reference labels the 8-connected regions of `mag == -1` with
// scipy.ndimage.label and, region by region in label order, overwrites the mask on every
// 8-neighbour J of the region's cells with `elev[J] == elev[first cell of the region]`.
// scipy numbers regions by the raster position of their first cell, and later regions
// overwrite earlier ones, so the final value of J is decided by the neighbouring region whose
// first (= minimum-index) cell is largest.  Here: union-find labelling with min-index roots
// over the compacted list of flat cells, then every neighbour of a flat cell takes
// `elev[J] == elev[root]` of the max-root region touching it.  Integer/bool work: bit-exact.
#include "internal.h"

namespace {

#include "ccl.h"

// For every neighbour J of every flat cell: flats[J] = (elev[J] == elev[root of the max-root flat
// region adjacent to J]).  Several threads may compute the same J; they all write the same value.
__global__ void k_flats_extend(const int32_t *__restrict__ list, const int32_t *__restrict__ count,
                               const uint8_t *__restrict__ flat0, const int32_t *__restrict__ labels,
                               const double *__restrict__ elev, uint8_t *__restrict__ flats, int n, int m)
{
    const int32_t nf = *count;
    for (int32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nf * 8; q += gridDim.x * blockDim.x) {
        const int32_t c = list[q >> 3];
        int d = q & 7;
        d += (d >= 4);                                   // skip the centre of the 3x3
        const int i = c / m + d / 3 - 1, j = c % m + d % 3 - 1;
        if (i < 0 || i >= n || j < 0 || j >= m) continue;
        const int32_t J = i * m + j;
        // A cell that is not flat itself can only become true through a neighbouring region whose root has ITS elevation, and the
        // threads of that region's cells get past this test: a thread whose own region cannot set J leaves J (false from the copy
        // of flat0) to them -- on fractal terrain that is nearly every thread, and it saves the scan of J's eight neighbours.
        // (A flat J always takes the full path: its value is rewritten as the reference rewrites it.)
        if (!flat0[J] && !(elev[J] == elev[labels[c]])) continue;
        int32_t best = -1;
        for (int dd = 0; dd < 9; dd++) {
            if (dd == 4) continue;
            const int ii = i + dd / 3 - 1, jj = j + dd % 3 - 1;
            if (ii < 0 || ii >= n || jj < 0 || jj >= m) continue;
            const int32_t nb = ii * m + jj;
            if (flat0[nb]) { const int32_t r = labels[nb]; best = r > best ? r : best; }
        }
        flats[J] = (elev[J] == elev[best]);
    }
}

// direction[flats] = mag[flats] = -1 (dem_processing.py:611-612), only near flat cells
__global__ void k_flats_patch(const int32_t *__restrict__ list, const int32_t *__restrict__ count,
                              const uint8_t *__restrict__ flats, double *__restrict__ mag, double *__restrict__ dir,
                              int n, int m)
{
    const int32_t nf = *count;
    for (int32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nf * 9; q += gridDim.x * blockDim.x) {
        const int32_t c = list[q / 9];
        const int d = q % 9;
        const int i = c / m + d / 3 - 1, j = c % m + d % 3 - 1;
        if (i < 0 || i >= n || j < 0 || j >= m) continue;
        const int32_t J = i * m + j;
        if (flats[J]) { mag[J] = -1.0; dir[J] = -1.0; }
    }
}

__global__ __launch_bounds__(256) void k_count_flats(const uint8_t *__restrict__ flats, int64_t NN, int32_t *count)
{
    int32_t local = 0;
    const int64_t nvec = NN >> 4, stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t v = t0; v < nvec; v += stride) {                // 16 mask bytes per load
        const uint4 f4 = reinterpret_cast<const uint4 *>(flats)[v];
        const uint32_t fw[4] = {f4.x, f4.y, f4.z, f4.w};
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int b = 0; b < 4; b++) local += ((fw[q] >> (8 * b)) & 0xFFu) != 0;
    }
    for (int64_t c = (nvec << 4) + t0; c < NN; c += stride) local += flats[c] != 0;
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off);
    __shared__ int32_t s_w[4];                                   // one add per workgroup: the counter is a single address
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) { const int32_t tot = s_w[0] + s_w[1] + s_w[2] + s_w[3]; if (tot) atomicAdd(count, tot); }
}

}  // namespace

int stage_flats(pydem_tile *t)
{
    const int n = (int)t->n, m = (int)t->m;
    PYDEM_TRY(tile_alloc(t, &t->labels, (size_t)t->NN));
    PYDEM_TRY(tile_alloc(t, &t->flatlist, (size_t)t->NN));
    HIP_TRY(hipEventRecord(t->ev[3], t->stream));
    HIP_TRY(hipMemsetAsync(t->counters, 0, 64 * sizeof(int32_t), t->stream));
    int32_t *cnt = t->counters;          // [0] number of flat0 cells, [1] final flats count
    const int big = (int)(cdiv(t->NN, 256) < 4096 ? cdiv(t->NN, 256) : 4096);
    hipLaunchKernelGGL(k_compact_flats, dim3(big), dim3(256), 0, t->stream, t->flat0, t->NN, t->flatlist, cnt);
    HIP_TRY(hipMemcpyAsync(t->flats, t->flat0, (size_t)t->NN, hipMemcpyDeviceToDevice, t->stream));
    HIP_TRY(hipMemcpyAsync(t->h_counters, cnt, sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    const int32_t nf = t->h_counters[0];
    if (nf > 0) {
        const int g1 = (int)(cdiv(nf, 256) < 2048 ? cdiv(nf, 256) : 2048);
        const int g8 = (int)(cdiv((int64_t)nf * 9, 256) < 4096 ? cdiv((int64_t)nf * 9, 256) : 4096);
        hipLaunchKernelGGL(k_label_init, dim3(g1), dim3(256), 0, t->stream, t->flatlist, cnt, t->labels, m);
        hipLaunchKernelGGL(k_label_union, dim3(g1), dim3(256), 0, t->stream, t->flatlist, cnt, t->flat0, t->labels, n, m);
        hipLaunchKernelGGL(k_label_flatten, dim3(g1), dim3(256), 0, t->stream, t->flatlist, cnt, t->labels);
        hipLaunchKernelGGL(k_flats_extend, dim3(g8), dim3(256), 0, t->stream, t->flatlist, cnt, t->flat0, t->labels,
                           t->elev, t->flats, n, m);
        hipLaunchKernelGGL(k_flats_patch, dim3(g8), dim3(256), 0, t->stream, t->flatlist, cnt, t->flats, t->mag, t->dir, n, m);
    }
    hipLaunchKernelGGL(k_count_flats, dim3(big), dim3(256), 0, t->stream, t->flats, t->NN, cnt + 1);
    HIP_TRY(hipMemcpyAsync(t->h_counters, cnt, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipEventRecord(t->ev[4], t->stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventSynchronize(t->ev[4]));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, t->ev[3], t->ev[4]));
    t->tm.flats_ms = ms;
    t->tm.n_flats = t->h_counters[1];
    return 0;
}
