// micro-benchmark: per-level cost components of a single-workgroup level loop on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(1024) void k_sync_only(int levels, int *out)
{
    __shared__ int s;
    int acc = 0;
    for (int r = 0; r < levels; r++) {
        if (threadIdx.x == 0) s = r;
        __syncthreads();
        acc += s;
        __syncthreads();
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = acc;
}

// each level: every active thread does ONE dependent random load (pointer chase), then barrier
__global__ __launch_bounds__(1024) void k_chase(const int *__restrict__ next, int levels, int active, int *out)
{
    int p = threadIdx.x * 9973 + 17;
    for (int r = 0; r < levels; r++) {
        if ((int)threadIdx.x < active) p = next[p];
        __syncthreads();
    }
    if ((int)threadIdx.x < active) out[threadIdx.x] = p;
}

// each level: one returning atomic on a random address
__global__ __launch_bounds__(1024) void k_atomic(int *__restrict__ buf, const int *__restrict__ next, int levels, int active, int *out)
{
    int p = threadIdx.x * 9973 + 17;
    for (int r = 0; r < levels; r++) {
        if ((int)threadIdx.x < active) { const int o = atomicAdd(&buf[p], 1); p = (p * 31 + o + 7) & ((1 << 26) - 1); }
        __syncthreads();
    }
    if ((int)threadIdx.x < active) out[threadIdx.x] = p;
}

// pointer chase with no barrier, one wave
__global__ void k_chase_wave(const int *__restrict__ next, int levels, int *out)
{
    int p = threadIdx.x * 9973 + 17;
    for (int r = 0; r < levels; r++) p = next[p];
    out[threadIdx.x] = p;
}

int main()
{
    const int N = 1 << 26;   // 256 MB of ints
    std::vector<int> h(N);
    unsigned long long x = 88172645463325252ull;
    for (int i = 0; i < N; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (int)(x % N); }
    int *d, *out, *buf;
    CK(hipMalloc(&d, (size_t)N * 4)); CK(hipMalloc(&out, 4096 * 4)); CK(hipMalloc(&buf, (size_t)N * 4));
    CK(hipMemcpy(d, h.data(), (size_t)N * 4, hipMemcpyHostToDevice));
    CK(hipMemset(buf, 0, (size_t)N * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int L = 2000;
    auto timeit = [&](const char *name, auto launch) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-40s %.3f us per level\n", name, ms * 1e3 / L);
    };
    timeit("3 barriers, 1024 threads", [&] { hipLaunchKernelGGL(k_sync_only, dim3(1), dim3(1024), 0, 0, L, out); });
    for (int act : {1, 64, 1024})
        timeit(act == 1 ? "random load + barrier (1 lane)" : act == 64 ? "random load + barrier (64 lanes)" : "random load + barrier (1024 lanes)",
               [&] { hipLaunchKernelGGL(k_chase, dim3(1), dim3(1024), 0, 0, d, L, act, out); });
    for (int act : {64, 1024})
        timeit(act == 64 ? "random atomic + barrier (64 lanes)" : "random atomic + barrier (1024 lanes)",
               [&] { hipLaunchKernelGGL(k_atomic, dim3(1), dim3(1024), 0, 0, buf, d, L, act, out); });
    timeit("random load chain, one wave, no barrier", [&] { hipLaunchKernelGGL(k_chase_wave, dim3(1), dim3(64), 0, 0, d, L, out); });
    return 0;
}
