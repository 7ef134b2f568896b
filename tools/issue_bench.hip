// tools/issue_bench.hip -- what a gfx950 CU issues per cycle: scalar ALU, vector ALU, and both from the same waves,
// at 1 .. 8 waves per SIMD.  hipcc --offload-arch=gfx950 -O2 tools/issue_bench.hip -o /tmp/issue_bench && /tmp/issue_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP8(x) x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(256) void k(int iters, unsigned *out, unsigned long long *cyc)
{
    unsigned v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3, s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {   // 64 scalar adds, four independent chains
            asm volatile(REP8(REP8("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n")) : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)::"scc");
        } else if (MODE == 1) {   // 64 x 4 vector adds
            asm volatile(REP8(REP8("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n")) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        } else if (MODE == 2) {   // alternating: 128 scalar + 128 vector
            asm volatile(REP8(REP8("s_add_u32 %0, %0, 1\n v_add_u32 %4, %4, 1\n s_add_u32 %1, %1, 1\n v_add_u32 %5, %5, 1\n")) : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+v"(v0), "+v"(v1)::"scc");
        } else if (MODE == 3) {   // dependent scalar chain
            asm volatile(REP8(REP8("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n")) : "+s"(s0)::"scc");
        } else if (MODE == 4) {   // readlane + dependent scalar + vector (the compare loop's mix)
            asm volatile(REP8(REP8("s_ff1_i32_b32 %1, %0\n v_readlane_b32 %2, %4, %1\n s_bitset0_b32 %0, %1\n v_add_u32 %5, %5, %2\n")) : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+v"(v0), "+v"(v1)::"scc");
        } else if (MODE == 5) {   // v_readlane only (independent, constant lane)
            asm volatile(REP8(REP8("v_readlane_b32 %0, %4, 1\n v_readlane_b32 %1, %5, 2\n v_readlane_b32 %2, %4, 3\n v_readlane_b32 %3, %5, 4\n")) : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+v"(v0), "+v"(v1));
        } else if (MODE == 6) {   // v_cmp into an SGPR pair + v_addc reading it (2 wait states between)
            asm volatile(REP8(REP8("v_cmp_eq_u32_e64 s[20:21], %0, %1\n v_cmp_eq_u32_e64 s[22:23], %1, %0\n s_nop 0\n v_addc_co_u32_e64 %2, s[20:21], 0, %2, s[20:21]\n v_addc_co_u32_e64 %3, s[22:23], 0, %3, s[22:23]\n")) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)::"s20", "s21", "s22", "s23");
        } else if (MODE == 7) {   // v_pk_max_u16 + v_cmp to VCC + v_addc (vcc)
            asm volatile(REP8(REP8("v_pk_max_u16 %2, %0, %1\n s_nop 0\n v_cmp_eq_u32_e32 vcc, %0, %2\n s_nop 1\n v_addc_co_u32_e32 %3, vcc, 0, %3, vcc\n")) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)::"vcc");
        } else if (MODE == 8) {   // the compare loop's body as shipped (no branch)
            asm volatile(REP8(REP8("s_ff1_i32_b64 %1, vcc\n v_readlane_b32 %2, %4, %1\n s_bitset0_b64 vcc, %1\n s_nop 0\n v_pk_max_u16 %5, %4, %2\n s_nop 0\n v_cmp_eq_u32_e64 s[20:21], %4, %5\n s_nop 1\n v_addc_co_u32_e64 %6, s[20:21], 0, %6, s[20:21]\n")) : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+v"(v0), "+v"(v1), "+v"(v2)::"vcc", "s20", "s21");
        } else if (MODE == 9) {   // scalar loads from the kernel arguments' page (SMEM issue)
            asm volatile(REP8(REP8("s_load_dword %0, %4, 0x0\n s_load_dword %1, %4, 0x4\n s_load_dword %2, %4, 0x8\n s_load_dword %3, %4, 0xc\n")) "s_waitcnt lgkmcnt(0)\n" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "s"(out));
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + s0 + s1 + s2 + s3;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = c1 - c0; cyc[1] = t1 - t0; }
}

template <int MODE>
static void run(const char *name, int perInstr, int wavesPerSimd, int cus)
{
    const int iters = 2000;
    unsigned *out; unsigned long long *cyc;
    hipMalloc(&out, (size_t)cus * wavesPerSimd * 256 * 4); hipMalloc(&cyc, 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<cus * wavesPerSimd, 256>>>(10, out, cyc);
    hipEventRecord(a);
    k<MODE><<<cus * wavesPerSimd, 256>>>(iters, out, cyc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    const double instrPerWave = (double)iters * 64;             // passes of the 4- (or 5-, 9-) instruction group, per wave
    const double perCU = instrPerWave * wavesPerSimd * 4;        // all waves of a CU
    // clock: s_memtime ticks per microsecond from the realtime counter (100 MHz)
    const double us = h[1] / 100.0, mhz = h[0] / us;
    printf("%-50s %d waves/SIMD: %8.1f us | per CU %5.2f instr/clk, per SIMD one per %5.2f clk (clock %.0f MHz) | oldest wave: one per %5.1f clk\n", name, wavesPerSimd, ms * 1e3,
           perInstr * perCU / (ms * 1e3 * mhz), ms * 1e3 * mhz * 4 / (perInstr * perCU), mhz, h[0] / (instrPerWave * perInstr));
    hipFree(out); hipFree(cyc);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs\n", p.name, cus);
    for (int w : {1, 2, 4, 8}) {
        printf("--\n");
        run<0>("scalar adds, 4 independent chains", 4, w, cus);
        run<3>("scalar adds, one dependent chain", 4, w, cus);
        run<1>("vector adds, 4 independent chains", 4, w, cus);
        run<2>("scalar and vector adds alternating", 4, w, cus);
        run<4>("s_ff1, v_readlane, s_bitset0, v_add (dependent)", 4, w, cus);
        run<5>("v_readlane x 4 (independent)", 4, w, cus);
        run<6>("2 x v_cmp -> SGPR pair, 2 x v_addc from it", 4, w, cus);
        run<7>("v_pk_max, v_cmp -> vcc, v_addc (3 VALU + nops)", 3, w, cus);
        run<8>("compare loop body as shipped (3 SALU + 4 VALU)", 7, w, cus);
        run<9>("s_load_dword x 4", 4, w, cus);
    }
    return 0;
}
