// igd_sortscan.hpp -- device building blocks shared by igd_create.hip and igd_hip.hip (Seqpare):
// an exclusive scan with int64 results and a STABLE LSD byte radix sort of (uint32 key, uint32
// payload) pairs.  Everything is `static`: each translation unit gets its own copy of the kernels.
#ifndef IGD_SORTSCAN_HPP
#define IGD_SORTSCAN_HPP
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WAVE 64
#define SCAN_WG 256
#define SCAN_PER 8
#define SCAN_TILE (SCAN_WG * SCAN_PER)
#define RS_WG 256
#define RS_WAVES (RS_WG / WAVE)
#define RS_STRIPS 8                                  // strips of 64 per wave
#define RS_BLOCK (RS_WG * RS_STRIPS)                 // 2048 pairs per workgroup

// ---------------------------------------------------------------------------------------------
// exclusive scan, int64 result (three launches: tile sums, scan of sums by one workgroup, apply)
template <typename T>
static __global__ void __launch_bounds__(SCAN_WG) ss_scan_sums(const T *__restrict__ in, int64_t n, int64_t *__restrict__ sums)
{
    __shared__ int64_t red[SCAN_WG / WAVE];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER; k++) {
        const int64_t i = base + k * SCAN_WG + threadIdx.x;
        if (i < n) s += (int64_t)in[i];
    }
    for (int o = WAVE / 2; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & (WAVE - 1)) == 0) red[threadIdx.x / WAVE] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t t = 0;
        for (int w = 0; w < SCAN_WG / WAVE; w++) t += red[w];
        sums[blockIdx.x] = t;
    }
}

static __global__ void __launch_bounds__(1024) ss_scan_of_sums(int64_t *__restrict__ sums, int64_t nb, int64_t *__restrict__ total)
{
    __shared__ int64_t wsum[16];
    __shared__ int64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    for (int64_t base = 0; base < nb; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const int64_t v = i < nb ? sums[i] : 0;
        int64_t x = v;
        for (int o = 1; o < WAVE; o <<= 1) {
            const int64_t y = __shfl_up(x, o);
            if (lane >= o) x += y;
        }
        if (lane == WAVE - 1) wsum[w] = x;
        __syncthreads();
        int64_t pre = carry;
        for (int k = 0; k < w; k++) pre += wsum[k];
        if (i < nb) sums[i] = pre + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = pre + x;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total) *total = carry;
}

template <typename T, typename O>
static __global__ void __launch_bounds__(SCAN_WG) ss_scan_apply(const T *__restrict__ in, int64_t n, const int64_t *__restrict__ sums, O *__restrict__ out)
{
    __shared__ int64_t wsum[SCAN_WG / WAVE];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_PER;
    int64_t v[SCAN_PER], s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER; k++) {
        v[k] = base + k < n ? (int64_t)in[base + k] : 0;
        s += v[k];
    }
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    int64_t x = s;
    for (int o = 1; o < WAVE; o <<= 1) {
        const int64_t y = __shfl_up(x, o);
        if (lane >= o) x += y;
    }
    if (lane == WAVE - 1) wsum[w] = x;
    __syncthreads();
    int64_t pre = sums[blockIdx.x] + x - s;
    for (int k = 0; k < w; k++) pre += wsum[k];
#pragma unroll
    for (int k = 0; k < SCAN_PER; k++) {
        if (base + k < n) out[base + k] = (O)pre;
        pre += v[k];
    }
}

template <typename T, typename O>
static hipError_t exclusive_scan(const T *in, int64_t n, O *out, int64_t *sums, int64_t *d_total, hipStream_t st)
{
    if (n <= 0) return hipMemsetAsync(d_total, 0, 8, st);
    const int64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    ss_scan_sums<T><<<(unsigned)nb, SCAN_WG, 0, st>>>(in, n, sums);
    ss_scan_of_sums<<<1, 1024, 0, st>>>(sums, nb, d_total);
    ss_scan_apply<T, O><<<(unsigned)nb, SCAN_WG, 0, st>>>(in, n, sums, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Stable LSD radix sort by a uint32 key (tile number in `create`, group in Seqpare), one byte per pass.
// A workgroup owns RS_BLOCK consecutive pairs; wave w owns the w-th quarter, in strips of 64, so
// "earlier in the input" = (lower block, lower wave, lower strip, lower lane).
static __global__ void __launch_bounds__(RS_WG) ss_rs_hist(const uint32_t *__restrict__ keys, int64_t n, int shift,
                                                   uint32_t *__restrict__ hist, int64_t nBlocks)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_BLOCK;
    uint32_t kv[RS_STRIPS];
#pragma unroll
    for (int k = 0; k < RS_STRIPS; k++) {                 // (all of the thread's keys asked for at once: a load inside `if (i < n)` is waited for before the next one goes out)
        const int64_t i = base + k * RS_WG + threadIdx.x;
        kv[k] = keys[i < n ? i : base];
    }
#pragma unroll
    for (int k = 0; k < RS_STRIPS; k++) {
        const int64_t i = base + k * RS_WG + threadIdx.x;
        if (i < n) atomicAdd(&h[(kv[k] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(int64_t)threadIdx.x * nBlocks + blockIdx.x] = h[threadIdx.x];      // digit-major: one scan gives every base
}

static __global__ void __launch_bounds__(RS_WG) ss_rs_scatter(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                      int64_t n, int shift, const int64_t *__restrict__ digitBase,
                                                      int64_t nBlocks, uint32_t *__restrict__ keysOut,
                                                      uint32_t *__restrict__ valsOut)
{
    __shared__ int64_t woff[RS_WAVES][256];
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    const int64_t base = (int64_t)blockIdx.x * RS_BLOCK + (int64_t)w * (RS_STRIPS * WAVE);
    for (int k = 0; k < RS_WAVES; k++) woff[k][threadIdx.x] = 0;
    __syncthreads();
    uint32_t key[RS_STRIPS], val[RS_STRIPS];
#pragma unroll
    for (int k = 0; k < RS_STRIPS; k++) {
        const int64_t i = base + k * WAVE + lane;
        key[k] = i < n ? keys[i] : 0xffffffffu;
        val[k] = i < n ? vals[i] : 0u;
        if (i < n) atomicAdd((unsigned long long *)&woff[w][(key[k] >> shift) & 255u], 1ull);
    }
    __syncthreads();
    {   // digit d (= threadIdx.x): global base of this block, then the waves in order
        int64_t g = digitBase[(int64_t)threadIdx.x * nBlocks + blockIdx.x];
        for (int k = 0; k < RS_WAVES; k++) {
            const int64_t t = woff[k][threadIdx.x];
            woff[k][threadIdx.x] = g;
            g += t;
        }
    }
    __syncthreads();
    const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int k = 0; k < RS_STRIPS; k++) {
        const int64_t i = base + k * WAVE + lane;
        const bool valid = i < n;
        const uint32_t d = (key[k] >> shift) & 255u;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1u;
            const uint64_t m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        if (valid) {
            const int rank = __popcll(peers & lt);
            const int64_t pos = woff[w][d] + rank;
            keysOut[pos] = key[k];
            valsOut[pos] = val[k];
        }
        __builtin_amdgcn_wave_barrier();
        if (valid && (peers & lt) == 0) woff[w][d] += __popcll(peers);      // one leader per digit
        __builtin_amdgcn_wave_barrier();
    }
}


// sorts (kA, vA) by the low `bits` bits of the key; kB/vB are scratch of the same size, hist holds
// 256 * nBlocks uint32, digitBase as many int64, sums ceil(256*nBlocks / SCAN_TILE) + 1 int64.
// On return *kA/*vA point at the sorted arrays (the buffers may have been swapped).
static inline int64_t rs_blocks(int64_t n) { return (n + RS_BLOCK - 1) / RS_BLOCK; }
static hipError_t radix_sort_pairs(uint32_t **kA, uint32_t **vA, uint32_t **kB, uint32_t **vB, int64_t n, int bits,
                                   uint32_t *hist, int64_t *digitBase, int64_t *sums, int64_t *d_total, hipStream_t st)
{
    const int64_t nBlocks = rs_blocks(n), nh = nBlocks * 256;
    hipError_t e = hipSuccess;
    for (int shift = 0; shift < bits && e == hipSuccess; shift += 8) {
        ss_rs_hist<<<(unsigned)nBlocks, RS_WG, 0, st>>>(*kA, n, shift, hist, nBlocks);
        e = exclusive_scan<uint32_t, int64_t>(hist, nh, digitBase, sums, d_total, st);
        if (e != hipSuccess) break;
        ss_rs_scatter<<<(unsigned)nBlocks, RS_WG, 0, st>>>(*kA, *vA, n, shift, digitBase, nBlocks, *kB, *vB);
        e = hipGetLastError();
        uint32_t *t = *kA; *kA = *kB; *kB = t;
        t = *vA; *vA = *vB; *vB = t;
    }
    return e;
}
#endif
