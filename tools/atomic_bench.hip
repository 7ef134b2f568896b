// tools/atomic_bench.hip -- what do 3.4 x 10^7 64-bit atomic adds to a table of N counters cost (the scan kernels' per-record
// adds when the files do not fit LDS counters): agent scope on ONE table (resolved beyond the XCD-private L2s) against
// workgroup scope on a table per XCD (row = the hardware's XCC_ID: all adders of a row share its L2, where the add is done).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_bench tools/atomic_bench.hip && /tmp/atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned long long u64;
__device__ __forceinline__ unsigned rnd(unsigned x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }
template <int MODE>   // 0: agent scope, one table; 1: workgroup scope, table per XCD
__global__ __launch_bounds__(1024) void k_add(u64 *tab, int n, size_t stride, int per, int *xccSeen)
{
    unsigned s = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
    u64 *row = tab;
    if (MODE == 1) {
        const int xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | 20) & 15;   // HW_REG_XCC_ID, bits 3:0
        row = tab + (size_t)xcc * stride;
        if (threadIdx.x == 0) xccSeen[blockIdx.x] = xcc;
    }
    for (int i = 0; i < per; i++) {
        s = rnd(s);
        const unsigned k = s % (unsigned)n;
        if (MODE == 0) __hip_atomic_fetch_add(&row[k], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(&row[k], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
__global__ void k_sum(const u64 *tab, size_t n, u64 *out)
{
    u64 s = 0;
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += tab[i];
    atomicAdd(out, s);
}
int main()
{
    const int grid = 512, per = 64;                 // 512 x 1024 x 64 = 3.36e7 adds
    int *xs; hipMalloc(&xs, grid * 4);
    u64 *out; hipMalloc(&out, 8);
    for (int n : {1900, 16000, 30000, 60000}) {
        const size_t stride = ((size_t)n + 31) & ~(size_t)31;
        u64 *tab; hipMalloc(&tab, 16 * stride * 8);
        for (int mode = 0; mode < 2; mode++) {
            float best = 1e9f;
            u64 total = 0;
            for (int rep = 0; rep < 4; rep++) {
                hipMemset(tab, 0, 16 * stride * 8); hipMemset(out, 0, 8);
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0);
                if (mode == 0) k_add<0><<<grid, 1024>>>(tab, n, stride, per, xs); else k_add<1><<<grid, 1024>>>(tab, n, stride, per, xs);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
                k_sum<<<256, 256>>>(tab, 16 * stride, out);
                hipMemcpy(&total, out, 8, hipMemcpyDeviceToHost);
            }
            std::vector<int> h(grid); hipMemcpy(h.data(), xs, grid * 4, hipMemcpyDeviceToHost);
            int mx = 0; for (int v : h) mx = v > mx ? v : mx;
            printf("N %6d  %s  %8.1f us  sum %llu (%s)%s\n", n, mode ? "workgroup scope, table per XCD" : "agent scope, one table        ", best * 1e3,
                   total, total == (u64)grid * 1024 * per ? "exact" : "LOST UPDATES", mode ? (mx == 7 ? "  xcc ids 0..7" : "  xcc ids odd") : "");
        }
        hipFree(tab);
    }
    return 0;
}
