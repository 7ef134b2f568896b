// igd_create.hip -- `igd create` on MI355X (gfx950): intervals -> tiles of an .igd, bit for bit
// what the reference writes.
//
// Reference path replaced (databio/IGD, /root/reference):
//   igd_add        src/igd_base.c:118-169   replicate an interval into tiles start/nbp..(end-1)/nbp
//   igd_saveT      src/igd_base.c:333-364   append the batch to one temp file per tile (input order)
//   igd_save       src/igd_base.c:396-461   per tile: read back, radix_sort_intv, append to the .igd
//   radix_sort_*   src/igd_base.h:196-249   klib's in-place MSD byte radix sort ("American flag"
//                                           permutation; insertion sort for <= 64 records)
// The host side (BED parsing, contig dictionary, _index.tsv, file writing) is igd_create.c.
//
// Why "bit for bit" shapes the design: the reference's sort is UNSTABLE, so the order of records
// with equal start inside a tile -- which `igd search -f` prints -- is whatever that exact algorithm
// leaves.  The displacement cycles of its permutation are inherently sequential, but only within
// one radix bucket of one tile, and a .igd has 10^5..10^6 tiles: so the device runs the same
// algorithm with ONE WAVE PER TILE -- counting, prefix, small-bucket ordering and the record
// gather use all 64 lanes, the cycle-following runs on lane 0 over LDS -- and thousands of tiles
// in flight.  Everything before it is ordinary data-parallel work:
//
//   k_span          per interval: tile span, per-contig tile count (max), replica count; the
//                   interval as one 16-byte record (so that the gather at the end is ONE access)
//   scan            replica offsets (exclusive scan, int64)
//   k_expand        (tile, interval) pairs in input order + per-tile counts
//   scan            tile offsets
//   ss_rs_hist / ss_rs_scatter   STABLE LSD byte radix sort of the pairs by tile number: afterwards
//                   every tile holds its intervals in input order -- what the reference's temp
//                   files hold (files in glob order, lines in file order)
//   k_tile_sort     the reference's sort per tile, then gather {idx,start,end,value} -> AoS records
//
// HBM traffic (R replicas, P radix passes = ceil(log256 nTiles)): expand 8R, each pass 4R + 16R,
// tile sort 8R + ~3 sector-granular gathers + 16R out: ~100 B per replica, i.e. ~5 GB for the
// 5.3e7-replica benchmark database -- about a millisecond at HBM speed.  The sequential part of
// the tile sort (a dependent LDS chain per moved record) is what bounds the kernel, not HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <errno.h>
#include <unistd.h>

#include "igd_hip.h"

extern "C" void igd_hip_set_error_(const char *msg);       // igd_hip.hip

#define CHK(call)                                                                     \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess) {                                                       \
            char b_[400];                                                             \
            snprintf(b_, sizeof b_, "%s: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            igd_hip_set_error_(b_);                                                   \
            rc = IGD_HIP_ERR_DEVICE;                                                  \
            goto done;                                                                \
        }                                                                             \
    } while (0)

#include <chrono>
static double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
// IGD_TIMING=1: wall time of each phase (the stream is drained at every mark, so the sum is a little
// above an untimed run)
#define MARK(what)                                                                                   \
    do {                                                                                             \
        if (timing) {                                                                                \
            (void)hipStreamSynchronize(st);                                                          \
            const double t_ = now_ms();                                                              \
            fprintf(stderr, "[igd_hip_create] %-26s %9.3f ms\n", what, t_ - tmark);                  \
            tmark = t_;                                                                              \
        }                                                                                            \
    } while (0)

#include "igd_sortscan.hpp"

#define TS_CAP 1024                                  // tile records sorted in LDS; larger tiles in HBM scratch
#define CTG_LDS 2048                                 // contigs whose tile count is reduced in LDS

// ---------------------------------------------------------------------------------------------
// igd_add, src/igd_base.c:124-126,131,145-147: tile span of every interval; mTiles[c] = 1 + max n2
__global__ void __launch_bounds__(256) k_span(const int32_t *__restrict__ ctg, const int32_t *__restrict__ start,
                                              const int32_t *__restrict__ end, const int32_t *__restrict__ value,
                                              const int32_t *__restrict__ file, int64_t n, int32_t nbp, int32_t nCtg,
                                              uint32_t *__restrict__ span, int32_t *__restrict__ mTiles,
                                              int4 *__restrict__ rec4)
{
    __shared__ int32_t lmax[CTG_LDS];
    const bool in_lds = nCtg <= CTG_LDS;
    if (in_lds) {
        for (int c = threadIdx.x; c < nCtg; c += blockDim.x) lmax[c] = 0;
        __syncthreads();
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t s = start[i], e = end[i];
        const int32_t n1 = s / nbp, n2 = (e - 1) / nbp;
        span[i] = (uint32_t)(n2 - n1 + 1);
        rec4[i] = make_int4(file[i], s, e, value ? value[i] : 0);      // the record as the .igd stores it: ONE gather later
        const int32_t c = ctg[i];
        if (in_lds) {
            if (lmax[c] < n2 + 1) atomicMax(&lmax[c], n2 + 1);
        } else atomicMax(&mTiles[c], n2 + 1);
    }
    if (in_lds) {
        __syncthreads();
        for (int c = threadIdx.x; c < nCtg; c += blockDim.x)
            if (lmax[c] > 0) atomicMax(&mTiles[c], lmax[c]);
    }
}

// one (tile, interval) pair per replica, in input order (interval-major, tiles ascending); tile counts
__global__ void __launch_bounds__(256) k_expand(const int32_t *__restrict__ ctg, const int32_t *__restrict__ start,
                                                const uint32_t *__restrict__ span, const int64_t *__restrict__ roff,
                                                int64_t n, int32_t nbp, const int64_t *__restrict__ tbase,
                                                uint32_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                uint32_t *__restrict__ tileCnt)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t t0 = (uint32_t)(tbase[ctg[i]] + start[i] / nbp);
        const uint32_t m = span[i];
        const int64_t o = roff[i];
        for (uint32_t j = 0; j < m; j++) {
            keys[o + j] = t0 + j;
            vals[o + j] = (uint32_t)i;
            atomicAdd(&tileCnt[t0 + j], 1u);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The reference's per-tile sort (src/igd_base.h:196-249) + record gather.  One wave per tile.
struct KV { int32_t key; uint32_t rec; };

struct TileArgs {
    const int64_t *tileOff;                       // [nTiles+1]
    const uint32_t *vals;                         // intervals of every tile, input order
    const int4 *rec4;                             // {idx,start,end,value} per interval
    void *out;                                    // AoS records
    KV *scratch;                                  // R entries: tiles larger than TS_CAP sort here
    int2 *gstack;                                 // R/64 + nTiles entries (segments of big tiles)
    int8_t *gshift;
    unsigned int *next;                           // dynamic tile counter
    int64_t nTiles;
    int32_t gType;
};

__device__ __forceinline__ void emit_record(const TileArgs &a, int64_t at, KV e)
{
    const int4 r = a.rec4[e.rec];                 // the line was fetched when the key was read: an L2 hit
    if (a.gType == 0) {
        int32_t *o = (int32_t *)a.out + at * 3;
        o[0] = r.x; o[1] = r.y; o[2] = r.z;
    } else ((int4 *)a.out)[at] = r;
}

__global__ void __launch_bounds__(WAVE) k_tile_sort(TileArgs a)
{
    __shared__ KV lkv[TS_CAP];
    __shared__ int32_t lo[256], hi[256], first[256];
    __shared__ int2 lstack[TS_CAP / 64 + 2];
    __shared__ int8_t lshift[TS_CAP / 64 + 2];
    __shared__ int top;
    __shared__ unsigned int tileShared;
    const int lane = threadIdx.x;

    for (;;) {
        if (lane == 0) tileShared = atomicAdd(a.next, 1u);
        __syncthreads();
        const int64_t t = tileShared;
        __syncthreads();
        if (t >= a.nTiles) return;
        const int64_t off = a.tileOff[t];
        const int64_t n64 = a.tileOff[t + 1] - off;
        if (n64 == 0) continue;
        const int32_t n = (int32_t)n64;
        const bool small = n <= TS_CAP;
        KV *A = small ? lkv : a.scratch + off;                      // flat pointer: LDS or HBM
        int2 *stack = small ? lstack : a.gstack + (off / 64 + t);
        int8_t *sshift = small ? lshift : a.gshift + (off / 64 + t);

        for (int32_t i = lane; i < n; i += WAVE) {
            const uint32_t rec = a.vals[off + i];
            KV e; e.key = a.rec4[rec].y; e.rec = rec;
            A[i] = e;
        }
        __syncthreads();

        if (n <= 64) {                                              // radix_sort: insertion sort of the whole tile
            if (lane < n) {
                const KV e = A[lane];
                int rank = 0;
                for (int q = 0; q < n; q++) {
                    const int32_t kq = A[q].key;
                    rank += (kq < e.key) || (kq == e.key && q < lane);
                }
                emit_record(a, off + rank, e);
            }
            __syncthreads();
            continue;
        }

        if (lane == 0) { stack[0] = make_int2(0, n); sshift[0] = 24; top = 1; }
        __syncthreads();
        while (top > 0) {
            const int2 seg = stack[top - 1];
            const int shift = sshift[top - 1];
            __syncthreads();
            if (lane == 0) top = top - 1;
            const int32_t beg = seg.x, end = seg.y, m = end - beg;
            // count
            for (int k = lane; k < 256; k += WAVE) hi[k] = 0;
            __syncthreads();
            for (int32_t p = beg + lane; p < end; p += WAVE) atomicAdd(&hi[(A[p].key >> shift) & 255], 1);
            __syncthreads();
            // prefix over the 256 buckets: lane l owns buckets 4l..4l+3
            int32_t c0 = hi[4 * lane], c1 = hi[4 * lane + 1], c2 = hi[4 * lane + 2], c3 = hi[4 * lane + 3];
            const int32_t mine = c0 + c1 + c2 + c3;
            int32_t x = mine;
            for (int o = 1; o < WAVE; o <<= 1) {
                const int32_t y = __shfl_up(x, o);
                if (lane >= o) x += y;
            }
            const bool single = __ballot(c0 == m || c1 == m || c2 == m || c3 == m) != 0;
            __syncthreads();
            int32_t b = beg + x - mine;
            lo[4 * lane] = first[4 * lane] = b; b += c0; hi[4 * lane] = b;
            lo[4 * lane + 1] = first[4 * lane + 1] = b; b += c1; hi[4 * lane + 1] = b;
            lo[4 * lane + 2] = first[4 * lane + 2] = b; b += c2; hi[4 * lane + 2] = b;
            lo[4 * lane + 3] = first[4 * lane + 3] = b; b += c3; hi[4 * lane + 3] = b;
            __syncthreads();
            // the displacement cycles (sequential by nature); nothing moves when one bucket holds everything
            if (!single && lane == 0) {
                for (int k = 0; k < 256;) {
                    const int32_t pk = lo[k];
                    if (pk == hi[k]) { k++; continue; }
                    KV carry = A[pk];
                    int d = (carry.key >> shift) & 255;
                    if (d == k) { lo[k] = pk + 1; continue; }
                    do {
                        const int32_t pd = lo[d];
                        const KV tmp = A[pd];
                        A[pd] = carry;
                        lo[d] = pd + 1;
                        carry = tmp;
                        d = (carry.key >> shift) & 255;
                    } while (d != k);
                    A[pk] = carry;
                    lo[k] = pk + 1;
                }
            }
            __syncthreads();
            // buckets: final (last byte, or one record), ordered in place (<= 64), or another level
            for (int32_t p0 = beg; p0 < end; p0 += WAVE) {
                const int32_t p = p0 + lane;
                if (p < end) {
                    const KV e = A[p];
                    const int k = (e.key >> shift) & 255;
                    const int32_t b0 = first[k], e0 = hi[k], sz = e0 - b0;
                    if (shift == 0 || sz == 1) emit_record(a, off + p, e);
                    else if (sz <= 64) {
                        int rank = 0;
                        for (int32_t q = b0; q < e0; q++) {
                            const int32_t kq = A[q].key;
                            rank += (kq < e.key) || (kq == e.key && q < p);
                        }
                        emit_record(a, off + b0 + rank, e);
                    } else if (p == b0) {
                        const int at = atomicAdd(&top, 1);
                        stack[at] = make_int2(b0, e0);
                        sshift[at] = (int8_t)(shift > 8 ? shift - 8 : 0);
                    }
                }
            }
            __syncthreads();
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
static int write_all(int fd, const void *p, size_t n)
{
    const char *c = (const char *)p;
    while (n > 0) {
        const ssize_t w = write(fd, c, n > ((size_t)1 << 30) ? ((size_t)1 << 30) : n);
        if (w < 0) { if (errno == EINTR) continue; return -1; }
        c += w; n -= (size_t)w;
    }
    return 0;
}

extern "C" void igd_hip_created_free(igd_hip_created *c)
{
    if (!c) return;
    free(c->nTile);
    free(c->nCnt);
    if (c->records) (void)hipHostFree(c->records);
    memset(c, 0, sizeof *c);
}

extern "C" int igd_hip_create(const igd_hip_create_desc *d, int device, igd_hip_created *out)
{
    int rc = IGD_HIP_OK;
    if (!d || !out || d->nbp <= 0 || d->nCtg < 0 || d->n < 0 || (d->gType != 0 && d->gType != 1) ||
        (d->n > 0 && (!d->ctg || !d->start || !d->end || !d->file)) || d->n >= 0xffffffffLL) {
        igd_hip_set_error_("igd_hip_create: bad descriptor");
        return IGD_HIP_ERR_ARG;
    }
    memset(out, 0, sizeof *out);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        igd_hip_set_error_("igd_hip_create: no HIP device");
        return IGD_HIP_ERR_DEVICE;
    }
    if (device < 0 || device >= ndev) {
        igd_hip_set_error_("igd_hip_create: device out of range");
        return IGD_HIP_ERR_ARG;
    }
    const int64_t n = d->n;
    const int32_t nCtg = d->nCtg;
    const size_t recBytes = d->gType == 0 ? 12 : 16;
    hipStream_t st = nullptr;
    int32_t *dc = nullptr, *ds = nullptr, *de = nullptr, *dv = nullptr, *df = nullptr, *dmT = nullptr;
    uint32_t *dspan = nullptr, *kA = nullptr, *vA = nullptr, *kB = nullptr, *vB = nullptr, *dcnt = nullptr, *dhist = nullptr;
    int64_t *droff = nullptr, *dsums = nullptr, *dtot = nullptr, *dtbase = nullptr, *dtoff = nullptr, *ddig = nullptr;
    KV *dscr = nullptr;
    int4 *drec = nullptr;
    int2 *dgst = nullptr;
    int8_t *dgsh = nullptr;
    unsigned int *dnext = nullptr;
    void *dout = nullptr;
    int64_t *tbase = nullptr;
    int64_t nTiles = 0, R = 0, maxTile = 0;
    int cus = 256;
    const bool timing = getenv("IGD_TIMING") != nullptr;
    double tmark = now_ms();
    {
        CHK(hipSetDevice(device));
        CHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));   // not hipGetDeviceProperties (~30 ms)
        if (cus <= 0) cus = 256;
        CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        out->nTile = (int32_t *)calloc((size_t)(nCtg > 0 ? nCtg : 1), sizeof(int32_t));
        tbase = (int64_t *)calloc((size_t)nCtg + 1, sizeof(int64_t));
        if (!out->nTile || !tbase) { rc = IGD_HIP_ERR_NOMEM; goto done; }
        const size_t nb4 = (size_t)(n > 0 ? n : 1) * 4;
        CHK(hipMalloc(&dc, nb4)); CHK(hipMalloc(&ds, nb4)); CHK(hipMalloc(&de, nb4)); CHK(hipMalloc(&df, nb4));
        CHK(hipMalloc(&dspan, nb4)); CHK(hipMalloc(&droff, nb4 * 2)); CHK(hipMalloc(&drec, nb4 * 4));
        CHK(hipMalloc(&dmT, (size_t)(nCtg > 0 ? nCtg : 1) * 4));
        CHK(hipMalloc(&dtbase, ((size_t)nCtg + 1) * 8));
        CHK(hipMalloc(&dtot, 64));
        CHK(hipMalloc(&dnext, 64));
        CHK(hipMemcpyAsync(dc, d->ctg, (size_t)n * 4, hipMemcpyHostToDevice, st));
        CHK(hipMemcpyAsync(ds, d->start, (size_t)n * 4, hipMemcpyHostToDevice, st));
        CHK(hipMemcpyAsync(de, d->end, (size_t)n * 4, hipMemcpyHostToDevice, st));
        CHK(hipMemcpyAsync(df, d->file, (size_t)n * 4, hipMemcpyHostToDevice, st));
        if (d->value && d->gType == 1) {
            CHK(hipMalloc(&dv, nb4));
            CHK(hipMemcpyAsync(dv, d->value, (size_t)n * 4, hipMemcpyHostToDevice, st));
        }
        CHK(hipMemsetAsync(dmT, 0, (size_t)(nCtg > 0 ? nCtg : 1) * 4, st));
        MARK("init + malloc + H2D");
        const int64_t nbScanN = (n + SCAN_TILE - 1) / SCAN_TILE + 1;
        // 1. spans, tiles per contig
        if (n > 0) {
            k_span<<<cus * 8, 256, 0, st>>>(dc, ds, de, dv, df, n, d->nbp, nCtg, dspan, dmT, drec);
            CHK(hipGetLastError());
        }
        CHK(hipMemcpyAsync(out->nTile, dmT, (size_t)nCtg * 4, hipMemcpyDeviceToHost, st));
        CHK(hipStreamSynchronize(st));
        for (int32_t c = 0; c < nCtg; c++) { tbase[c] = nTiles; nTiles += out->nTile[c]; }
        tbase[nCtg] = nTiles;
        if (nTiles >= 0xffffffffLL) { igd_hip_set_error_("igd_hip_create: too many tiles"); rc = IGD_HIP_ERR_ARG; goto done; }
        CHK(hipMemcpyAsync(dtbase, tbase, ((size_t)nCtg + 1) * 8, hipMemcpyHostToDevice, st));
        // 2. replica offsets
        CHK(hipMalloc(&dsums, (size_t)nbScanN * 8));
        CHK((exclusive_scan<uint32_t, int64_t>(dspan, n, droff, dsums, dtot, st)));
        CHK(hipMemcpyAsync(&R, dtot, 8, hipMemcpyDeviceToHost, st));
        CHK(hipStreamSynchronize(st));
        out->nTiles = nTiles;
        out->nRecords = R;
        out->nCnt = (int32_t *)calloc((size_t)(nTiles > 0 ? nTiles : 1), sizeof(int32_t));
        if (!out->nCnt) { rc = IGD_HIP_ERR_NOMEM; goto done; }
        if (R == 0) goto done;
        MARK("span + scan");
        // 3. pairs + tile counts
        const size_t rb4 = (size_t)R * 4;
        CHK(hipMalloc(&kA, rb4)); CHK(hipMalloc(&vA, rb4)); CHK(hipMalloc(&kB, rb4)); CHK(hipMalloc(&vB, rb4));
        CHK(hipMalloc(&dcnt, (size_t)(nTiles + 1) * 4));
        CHK(hipMalloc(&dtoff, (size_t)(nTiles + 1) * 8));
        CHK(hipMemsetAsync(dcnt, 0, (size_t)(nTiles + 1) * 4, st));
        k_expand<<<cus * 8, 256, 0, st>>>(dc, ds, dspan, droff, n, d->nbp, dtbase, kA, vA, dcnt);
        CHK(hipGetLastError());
        CHK(hipMemcpyAsync(out->nCnt, dcnt, (size_t)nTiles * 4, hipMemcpyDeviceToHost, st));
        // 4. tile offsets (nTiles + 1 entries: the extra zero count gives the end of the last tile)
        {
            const int64_t nbT = (nTiles + 1 + SCAN_TILE - 1) / SCAN_TILE + 1;
            int64_t *sumsT = nullptr;
            CHK(hipMalloc(&sumsT, (size_t)nbT * 8));
            hipError_t e = exclusive_scan<uint32_t, int64_t>(dcnt, nTiles + 1, dtoff, sumsT, dtot, st);
            hipError_t e2 = hipStreamSynchronize(st);
            (void)hipFree(sumsT);
            CHK(e); CHK(e2);
        }
        for (int64_t t = 0; t < nTiles; t++) if (out->nCnt[t] > maxTile) maxTile = out->nCnt[t];
        MARK("expand + tile offsets");
        // 5. stable radix sort of the pairs by tile number
        {
            const int64_t nBlocks = (R + RS_BLOCK - 1) / RS_BLOCK;
            const int64_t nh = nBlocks * 256;
            CHK(hipMalloc(&dhist, (size_t)nh * 4));
            CHK(hipMalloc(&ddig, (size_t)nh * 8));
            int64_t *sumsH = nullptr;
            CHK(hipMalloc(&sumsH, (size_t)((nh + SCAN_TILE - 1) / SCAN_TILE + 1) * 8));
            int bits = 0;
            while (bits < 32 && (nTiles - 1) >> bits) bits++;
            hipError_t e = radix_sort_pairs(&kA, &vA, &kB, &vB, R, bits, dhist, ddig, sumsH, dtot, st);
            hipError_t e2 = hipStreamSynchronize(st);
            (void)hipFree(sumsH);
            CHK(e); CHK(e2);
        }
        MARK("radix sort by tile");
        // 6. the reference's sort inside every tile + gather into records
        (void)hipFree(kB); kB = nullptr;
        (void)hipFree(vB); vB = nullptr;
        (void)hipFree(kA); kA = nullptr;
        CHK(hipMalloc(&dout, (size_t)R * recBytes));
        if (maxTile > TS_CAP) {
            CHK(hipMalloc(&dscr, (size_t)R * sizeof(KV)));
            CHK(hipMalloc(&dgst, (size_t)(R / 64 + nTiles + 2) * sizeof(int2)));
            CHK(hipMalloc(&dgsh, (size_t)(R / 64 + nTiles + 2)));
        }
        CHK(hipMemsetAsync(dnext, 0, 4, st));
        {
            TileArgs a;
            a.tileOff = dtoff; a.vals = vA; a.rec4 = drec;
            a.out = dout; a.scratch = dscr; a.gstack = dgst; a.gshift = dgsh; a.next = dnext;
            a.nTiles = nTiles; a.gType = d->gType;
            const int64_t want = nTiles < (int64_t)cus * 14 ? nTiles : (int64_t)cus * 14;
            k_tile_sort<<<(unsigned)(want > 0 ? want : 1), WAVE, 0, st>>>(a);
            CHK(hipGetLastError());
        }
        MARK("tile sort + gather");
        if (d->out_fd < 0) {
            CHK(hipHostMalloc(&out->records, (size_t)R * recBytes, hipHostMallocDefault));
            MARK("pinned alloc");
            CHK(hipMemcpyAsync(out->records, dout, (size_t)R * recBytes, hipMemcpyDeviceToHost, st));
            CHK(hipStreamSynchronize(st));
            MARK("D2H records");
        }
    }
done:
    if (rc == IGD_HIP_OK && d->out_fd >= 0) {
        // the .igd: header, then the tiles streamed through two pinned buffers
        const size_t hdr = 12 + 4 * (size_t)nCtg + 4 * (size_t)nTiles + 40 * (size_t)nCtg;
        char *h = (char *)calloc(1, hdr);
        if (!h) rc = IGD_HIP_ERR_NOMEM;
        else {
            memcpy(h, &d->nbp, 4); memcpy(h + 4, &d->gType, 4); memcpy(h + 8, &nCtg, 4);
            memcpy(h + 12, out->nTile, 4 * (size_t)nCtg);
            memcpy(h + 12 + 4 * (size_t)nCtg, out->nCnt, 4 * (size_t)nTiles);
            char *nm = h + 12 + 4 * (size_t)nCtg + 4 * (size_t)nTiles;
            for (int32_t i = 0; i < nCtg; i++)
                if (d->ctgName && d->ctgName[i]) strncpy(nm + 40 * (size_t)i, d->ctgName[i], 39);
            if (write_all(d->out_fd, h, hdr) != 0) { igd_hip_set_error_("igd_hip_create: write failed"); rc = IGD_HIP_ERR_ARG; }
            free(h);
        }
        const size_t total = (size_t)R * recBytes, CH = (size_t)32 << 20;
        char *stage[2] = { nullptr, nullptr };
        hipEvent_t ev[2] = { nullptr, nullptr };
        if (rc == IGD_HIP_OK && total > 0) {
            bool ok = hipHostMalloc((void **)&stage[0], CH, hipHostMallocDefault) == hipSuccess &&
                      hipHostMalloc((void **)&stage[1], CH, hipHostMallocDefault) == hipSuccess &&
                      hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) == hipSuccess &&
                      hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) == hipSuccess;
            const size_t nch = (total + CH - 1) / CH;
            for (size_t k = 0; ok && k <= nch; k++) {
                if (k < nch) {
                    const size_t len = k + 1 < nch ? CH : total - k * CH;
                    ok = hipMemcpyAsync(stage[k & 1], (const char *)dout + k * CH, len, hipMemcpyDeviceToHost, st) == hipSuccess &&
                         hipEventRecord(ev[k & 1], st) == hipSuccess;
                }
                if (ok && k > 0) {                       // chunk k-1 goes to the file while chunk k is in flight
                    const size_t j = k - 1, len = j + 1 < nch ? CH : total - j * CH;
                    ok = hipEventSynchronize(ev[j & 1]) == hipSuccess && write_all(d->out_fd, stage[j & 1], len) == 0;
                }
            }
            if (!ok) { igd_hip_set_error_("igd_hip_create: streaming the records to the file failed"); rc = IGD_HIP_ERR_DEVICE; }
            if (timing) { const double t_ = now_ms(); fprintf(stderr, "[igd_hip_create] %-26s %9.3f ms\n", "D2H + write (overlapped)", t_ - tmark); tmark = t_; }
        }
        if (stage[0]) (void)hipHostFree(stage[0]);
        if (stage[1]) (void)hipHostFree(stage[1]);
        if (ev[0]) (void)hipEventDestroy(ev[0]);
        if (ev[1]) (void)hipEventDestroy(ev[1]);
    }
    (void)hipFree(dc); (void)hipFree(ds); (void)hipFree(de); (void)hipFree(dv); (void)hipFree(df); (void)hipFree(dmT);
    (void)hipFree(dspan); (void)hipFree(kA); (void)hipFree(vA); (void)hipFree(kB); (void)hipFree(vB);
    (void)hipFree(dcnt); (void)hipFree(dhist); (void)hipFree(droff); (void)hipFree(dsums); (void)hipFree(dtot);
    (void)hipFree(dtbase); (void)hipFree(dtoff); (void)hipFree(ddig); (void)hipFree(dscr); (void)hipFree(dgst);
    (void)hipFree(dgsh); (void)hipFree(dnext); (void)hipFree(dout); (void)hipFree(drec);
    if (st) (void)hipStreamDestroy(st);
    free(tbase);
    if (rc != IGD_HIP_OK) igd_hip_created_free(out);
    return rc;
}
