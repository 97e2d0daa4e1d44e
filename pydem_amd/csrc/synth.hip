// synth.hip -- deterministic value-noise DEM generator (inputs only; not part of the reference).
// Third copy of pydem_amd/synth.py:fractal_unit and oracle/pydem_oracle.c:oracle_synth_fractal:
// same integer hash, same IEEE double operations in the same order (-ffp-contract=off), so the
// three produce bit-identical tiles.
#include "internal.h"

namespace {

__device__ __forceinline__ double hash01(uint32_t ix, uint32_t iy, uint32_t seed)
{
    uint32_t h = (ix * 0x9E3779B1u) ^ (iy * 0x85EBCA77u) ^ (seed * 0xC2B2AE3Du);
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return (double)h * 0x1p-32;
}

__global__ __launch_bounds__(256) void k_synth(double *__restrict__ z, int n, int m, uint32_t seed, int64_t row0,
                                               int64_t col0, int n_oct, int top_shift, double zmin, double zrange)
{
    const int64_t NN = (int64_t)n * m;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < NN; c += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = c / m, j = c - i * m;
        const uint32_t gi = (uint32_t)(i + row0), gj = (uint32_t)(j + col0);
        double acc = 0.0, amp = 1.0, norm = 0.0;
        for (int o = 0; o < n_oct; o++) {
            const int s = top_shift - o;
            const uint32_t mask = (1u << s) - 1u;
            const double inv = __longlong_as_double((long long)(1023 - s) << 52);   // 2^-s
            const uint32_t iy = gi >> s, ix = gj >> s;
            const double fy = (double)(gi & mask) * inv, fx = (double)(gj & mask) * inv;
            const double ty = (fy * fy) * (3.0 - 2.0 * fy), tx = (fx * fx) * (3.0 - 2.0 * fx);
            const uint32_t sd = (uint32_t)(((uint64_t)seed * 1000003ull + (uint64_t)o) & 0xFFFFFFFFull);
            const double v00 = hash01(ix, iy, sd), v10 = hash01(ix + 1, iy, sd);
            const double v01 = hash01(ix, iy + 1, sd), v11 = hash01(ix + 1, iy + 1, sd);
            const double a = v00 + tx * (v10 - v00), b = v01 + tx * (v11 - v01);
            const double nse = a + ty * (b - a);
            acc = acc + amp * nse;
            norm = norm + amp;
            amp = amp * 0.57;
        }
        z[c] = zmin + zrange * (acc / norm);
    }
}

}  // namespace

int stage_synth(pydem_tile *t, uint32_t seed, int64_t row0, int64_t col0, int n_oct, int top_shift, double zmin, double zrange)
{
    if (n_oct < 0 || top_shift < n_oct - 1 + 0 || top_shift > 30) { pydem_set_error("bad synth parameters"); return -2; }
    const int64_t g = cdiv(t->NN, 256);
    hipLaunchKernelGGL(k_synth, dim3((unsigned)(g < 16384 ? g : 16384)), dim3(256), 0, t->stream, t->elev, (int)t->n, (int)t->m,
                       seed, row0, col0, n_oct, top_shift, zmin, zrange);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(t->stream));
    return 0;
}
