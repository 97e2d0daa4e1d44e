// ccl.h -- compaction of a byte mask into a cell list and 8-connected component labelling of the listed cells
// (lock-free union-find, the label of a component is its smallest cell id = its first cell in raster order,
// which is also how scipy.ndimage.label orders its labels).  Shared by flats.hip (flats of the slope stencil,
// reference _find_flats_edges, pydem/dem_processing.py:657-680) and cond_device.hip (flats / depressions of the
// elevation conditioning, :396-426 and :551-579).  Include inside an anonymous namespace.
#pragma once

#define RLX_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

// stream compaction of flat0 != 0 into a list of cell ids.  Each thread reads 16 mask bytes with
// one 16 B load; ranks come from a wavefront prefix (shuffles) plus a per-block LDS prefix over
// the 4 waves, and the block reserves its output range with ONE global atomic per 64 Ki cells
// (a first version issued one atomic per wavefront on a single address: 22 ms at 16384^2).
__global__ __launch_bounds__(256) void k_compact_flats(const uint8_t *__restrict__ flat0, int64_t NN,
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
                for (int q = 0; q < 4; q++) v[q] = *reinterpret_cast<const uint4 *>(flat0 + c0 + 16 * q);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t w[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
                    for (int k = 0; k < 16; k++)
                        bits[j] |= (unsigned long long)(((w[k >> 2] >> (8 * (k & 3))) & 0xffu) ? 1u : 0u) << (16 * q + k);
                }
            } else {
                for (int k = 0; k < 64; k++)
                    if (c0 + k < NN && flat0[c0 + k]) bits[j] |= 1ull << k;
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

// Initial labels: a cell points at the first cell of its horizontal RUN as far as the wavefront sees it (consecutive list
// entries c - 1, c of one row: the list is in raster order within a compaction trip), not at itself.  A lake is rows of
// thousands of cells; with self-labels every union along a row lengthens a chain that every later find walks.
__global__ void k_label_init(const int32_t *__restrict__ list, const int32_t *__restrict__ count, int32_t *labels, int m)
{
    const int32_t nf = *count;
    const int lane = threadIdx.x & 63;
    for (int32_t q = blockIdx.x * blockDim.x + threadIdx.x;; q += gridDim.x * blockDim.x) {
        const bool valid = q < nf;
        if (!__ballot(valid)) break;
        const int32_t c = valid ? list[q] : -1;
        const int32_t prev = __shfl_up(c, 1);
        const bool head = lane == 0 || !valid || prev != c - 1 || c % m == 0;       // (lane 0: the union pass joins it to the run before it)
        const unsigned long long heads = __ballot(head);
        const int s = 63 - __clzll((long long)(heads & ((2ull << lane) - 1ull)));   // the last head at or before this lane
        const int32_t start = __shfl(c, s);
        if (valid) labels[c] = start;
    }
}

// root of x; a walk of two hops or more leaves x pointing at the root (labels only ever decrease, and the root is the
// smallest id on the path: atomicMin keeps whatever another thread put there in the meantime if that is smaller still)
__device__ __forceinline__ int32_t uf_find(int32_t *L, int32_t x)
{
    const int32_t x0 = x;
    int32_t p = RLX_LOAD(&L[x]);
    int hops = 0;
    while (p != x) { x = p; p = RLX_LOAD(&L[x]); hops++; }
    if (hops >= 2) atomicMin(&L[x0], x);
    return x;
}

// lock-free union with min-index roots (labels only ever decrease)
__device__ __forceinline__ void uf_union(int32_t *L, int32_t a, int32_t b)
{
    for (;;) {
        a = uf_find(L, a);
        b = uf_find(L, b);
        if (a == b) return;
        if (a > b) { const int32_t t = a; a = b; b = t; }
        const int32_t old = atomicMin(&L[b], a);     // link the larger root under the smaller
        if (old == b) return;
        b = old;                                      // someone else moved b meanwhile: retry
    }
}

__global__ void k_label_union(const int32_t *__restrict__ list, const int32_t *__restrict__ count,
                              const uint8_t *__restrict__ flat0, int32_t *labels, int n, int m)
{
    const int32_t nf = *count;
    for (int32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nf; q += gridDim.x * blockDim.x) {
        const int32_t c = list[q];
        const int i = c / m, j = c - i * m;
        // forward half of the 8-neighbourhood: E, SW, S, SE (each pair is visited once)
        if (j + 1 < m && flat0[c + 1]) uf_union(labels, c, c + 1);
        if (i + 1 < n) {
            if (j > 0 && flat0[c + m - 1]) uf_union(labels, c, c + m - 1);
            if (flat0[c + m]) uf_union(labels, c, c + m);
            if (j + 1 < m && flat0[c + m + 1]) uf_union(labels, c, c + m + 1);
        }
    }
}

__global__ void k_label_flatten(const int32_t *__restrict__ list, const int32_t *__restrict__ count, int32_t *labels)
{
    const int32_t nf = *count;
    for (int32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nf; q += gridDim.x * blockDim.x) {
        const int32_t c = list[q];
        const int32_t r = uf_find(labels, c);
        if (r != c) __hip_atomic_store(&labels[c], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

