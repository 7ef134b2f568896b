// tools/vmem_bench.hip -- how fast can a persistent, two-units-in-flight wave loop stream a ~320 MB record image,
// as a function of the load INSTRUCTIONS used (the scan kernel's question: 5 x dword + 5 x ushort per unit, or
// fewer, wider loads from a lane-major layout)?  Standalone: hipcc -O3 --offload-arch=gfx950 tools/vmem_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#define UNITS 190000
#define RECS 280                 // records per unit (of 320 slots)
#define WG 1024

template <int MODE>
__global__ __launch_bounds__(WG, 8) void k(const uint32_t *__restrict__ pse, const uint16_t *__restrict__ px,
                                             const uint32_t *__restrict__ blk, int nUnits, uint32_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int gwave = blockIdx.x * (WG / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nwaves = gridDim.x * (WG / 64);
    uint32_t acc = 0;
    uint32_t a[2][8];
    auto issue = [&](int u, int b) {
        const size_t off = (size_t)u * RECS;             // MODE 0/1/3: natural record order, units back to back
        if (u >= nUnits) { for (int r = 0; r < 8; r++) a[b][r] = 0; return; }
        if (MODE == 0 || MODE == 3) {                    // slot-major: 5 dword + 5 ushort (3: dwords only)
            const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)(pse + off), 0, RECS * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void *)(px + off), 0, RECS * 2, 0x00020000);
            uint32_t x = 0;
#pragma unroll
            for (int r = 0; r < 5; r++) {
                a[b][r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsA, lane * 4, r * 256, 0);
                if (MODE == 0) x ^= (uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsX, lane * 2, r * 128, 0);
            }
            a[b][5] = x; a[b][6] = 0; a[b][7] = 0;
        } else if (MODE == 1) {                          // lane-major, natural order: pse x4 + x1 at lane*20, px as 12 B/lane (x3)
            const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)(pse + off), 0, RECS * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void *)(blk + (size_t)u * 192), 0, ((RECS + 4) / 5) * 12, 0x00020000);
            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
            typedef uint32_t u3 __attribute__((ext_vector_type(3)));
            const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rsA, lane * 20, 0, 0);
            a[b][4] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsA, lane * 20, 16, 0);
            const u3 w = __builtin_amdgcn_raw_buffer_load_b96(rsX, lane * 12, 0, 0);
            a[b][0] = v.x; a[b][1] = v.y; a[b][2] = v.z; a[b][3] = v.w; a[b][5] = w.x; a[b][6] = w.y; a[b][7] = w.z;
        } else if (MODE == 4) {                          // slot-major as MODE 0, but a unit's dataset numbers right behind its record words (one 6 x RECS byte block per unit)
            const unsigned char *ub = (const unsigned char *)blk + (size_t)u * (RECS * 6);
            const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)ub, 0, RECS * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void *)(ub + RECS * 4), 0, RECS * 2, 0x00020000);
            uint32_t x = 0;
#pragma unroll
            for (int r = 0; r < 5; r++) {
                a[b][r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsA, lane * 4, r * 256, 0);
                x ^= (uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsX, lane * 2, r * 128, 0);
            }
            a[b][5] = x; a[b][6] = 0; a[b][7] = 0;
        } else if (MODE == 5) {                          // two records per lane: words as dwordx2 + dwordx2 + dword, dataset numbers as dword + dword + ushort (6 loads/unit)
            const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)(pse + off), 0, RECS * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void *)(px + off), 0, RECS * 2, 0x00020000);
            typedef uint32_t u2 __attribute__((ext_vector_type(2)));
            const u2 v0 = __builtin_amdgcn_raw_buffer_load_b64(rsA, lane * 8, 0, 0);
            const u2 v1 = __builtin_amdgcn_raw_buffer_load_b64(rsA, lane * 8, 512, 0);
            a[b][4] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsA, lane * 4, 1024, 0);
            a[b][5] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsX, lane * 4, 0, 0);
            a[b][6] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsX, lane * 4, 256, 0);
            a[b][7] = (uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsX, lane * 2, 512, 0);
            a[b][0] = v0.x; a[b][1] = v0.y; a[b][2] = v1.x; a[b][3] = v1.y;
        } else {                                         // MODE 2: one 32-byte block per lane: 2 x dwordx4
            const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void *)(blk + (size_t)u * 512), 0, ((RECS + 4) / 5) * 32, 0x00020000);
            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
            const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rsB, lane * 32, 0, 0);
            const u4 w = __builtin_amdgcn_raw_buffer_load_b128(rsB, lane * 32, 16, 0);
            a[b][0] = v.x; a[b][1] = v.y; a[b][2] = v.z; a[b][3] = v.w; a[b][4] = w.x; a[b][5] = w.y; a[b][6] = w.z; a[b][7] = w.w;
        }
    };
    issue(gwave, 0);
    for (int u = gwave; u < nUnits; u += 2 * nwaves) {
        issue(u + nwaves, 1);
#pragma unroll
        for (int r = 0; r < 8; r++) acc ^= a[0][r];
        issue(u + 2 * nwaves, 0);
#pragma unroll
        for (int r = 0; r < 8; r++) acc += a[1][r];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main()
{
    const size_t nrec = (size_t)UNITS * RECS + 4096;
    uint32_t *pse, *blk, *out;
    uint16_t *px;
    hipMalloc(&pse, nrec * 4); hipMalloc(&px, nrec * 2); hipMalloc(&blk, (size_t)UNITS * 2048 + 4096); hipMalloc(&out, 64);
    hipMemset(pse, 1, nrec * 4); hipMemset(px, 1, nrec * 2); hipMemset(blk, 1, (size_t)UNITS * 2048 + 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes[6] = {UNITS * RECS * 6.0, UNITS * (RECS * 4.0 + ((RECS + 4) / 5) * 12.0), UNITS * ((RECS + 4) / 5) * 32.0, UNITS * RECS * 4.0, UNITS * RECS * 6.0, UNITS * RECS * 6.0};
    const char *name[6] = {"slot-major 5 x dword + 5 x ushort (10 loads/unit)", "lane-major dwordx4 + dword + dwordx3 (3 loads/unit)",
                           "lane-major 32-B blocks, 2 x dwordx4 (2 loads/unit)", "slot-major 5 x dword only (5 loads/unit)",
                           "slot-major, words and dataset numbers of a unit adjacent",
                           "two records per lane: 2 x dwordx2 + dword, 2 x dword + ushort (6 loads/unit)"};
    for (int rep = 0; rep < 2; rep++)
        for (int m = 0; m < 6; m++) {
            float best = 1e9f;
            for (int it = 0; it < 12; it++) {
                hipEventRecord(e0);
                if (m == 0) k<0><<<512, WG>>>(pse, px, blk, UNITS, out);
                else if (m == 1) k<1><<<512, WG>>>(pse, px, blk, UNITS, out);
                else if (m == 2) k<2><<<512, WG>>>(pse, px, blk, UNITS, out);
                else if (m == 3) k<3><<<512, WG>>>(pse, px, blk, UNITS, out);
                else if (m == 4) k<4><<<512, WG>>>(pse, px, blk, UNITS, out);
                else k<5><<<512, WG>>>(pse, px, blk, UNITS, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (it >= 2 && ms < best) best = ms;
            }
            if (rep) printf("%-58s %7.1f us  %6.2f MB  %6.0f GB/s\n", name[m], best * 1e3, bytes[m] / 1e6, bytes[m] / (best * 1e-3) / 1e9);
        }
    hipError_t e = hipGetLastError();
    printf("status: %s\n", hipGetErrorString(e));
    return 0;
}
