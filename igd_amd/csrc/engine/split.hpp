// engine/split.hpp -- part of igd_hip.hip (included there once; not a stand-alone header).
// unordered batches: two-level split into tile buckets (k_split_local / k_split_fine*), scans, scatter
// ------------------------------------------------------------------------------------------
// Bucket path without global atomics ("split"): 10^6 random atomicAdds on the per-tile counters cost
// ~40 us each way on this part (device-scope atomics are resolved beyond the XCD-private L2s), so the
// (query, tile) pairs are grouped in two levels with LDS atomics only:
//   k_split_local   a workgroup takes 4096 queries, counts their pairs per COARSE bucket (tile >> shift,
//                   <= 1024 buckets) in LDS, and writes them, grouped by bucket, into its own region
//                   + one table row (offset | count << 16 per bucket);
//   k_split_fine    one workgroup per bucket collects the bucket's segments from all regions, counts per
//                   tile in LDS (the bucket spans 2^shift tiles), writes pairN/pairPos of its tiles and
//                   the pairs, tile by tile, into `pairs`.
// Output = exactly what count/scan/scatter leave (pairN, pairPos = END of each tile's range, pairs).
// Round 5 (LABNOTES R5-11): neither kernel gathers from memory any more -- k_split_local keeps one bit per tile ("holds
// records") and the contig tables in LDS, puts its region together in LDS and writes it out side by side; the fine kernel
// lays a bucket's segments out flat, fetches every tuple once (eight loads in flight per thread, none behind a branch) and
// places the pairs from LDS.  Each stage has the older form behind it for what does not fit (IGD_HIP_SPLIT_NO* switches).
#ifndef SP_WG
#define SP_WG 1024
#endif
#ifndef SP_PER
#define SP_PER 4
#endif
#define SP_Q (SP_WG * SP_PER) // queries per workgroup of k_split_local: more queries = longer segments per (workgroup, bucket)
#ifndef SPF_WG
#define SPF_WG 256           // threads of k_split_fine
#endif
#define SP_CAP (SP_Q * IGD_SHORT_TILES)
#define SP_LONG (SP_Q / 4)     // pairs of one k_split_local workgroup in one coarse bucket from which the bucket counts as piled up
#ifndef SP_MAXC
#define SP_MAXC 1024
#endif
#ifndef SP_MINSHIFT
#define SP_MINSHIFT 8          // log2 of the smallest coarse bucket (tiles)
#endif

// FAST: the usual database (its own power-of-two tiles, tile bits and contig tables staged in LDS, coarse buckets of >= 4 tiles):
// the per-query part written without branches -- both kinds of kernel spend their time issuing instructions (800 vector + 490
// scalar per wave in the general form, whose four unrolled queries each carry the re-tiled copy's gate, C division and the
// tile-by-tile loops).
template <bool FAST>
__global__ __launch_bounds__(SP_WG) void k_split_local(DbView db, const int32_t *__restrict__ ichr,
                                                       const int32_t *__restrict__ qs, const int32_t *__restrict__ qe,
                                                       int nq, int rule, int packed, int shift, int nCoarse,
                                                       uint32_t *__restrict__ table, SpTuple *__restrict__ reg,
                                                       int2 *__restrict__ longList, int32_t *__restrict__ ctl, int gate,
                                                       int epoch, u64 *__restrict__ zeroHits, u64 *__restrict__ zeroTotal,
                                                       int32_t *__restrict__ bucketLong, int bitsWords, int ctgStaged, int stageCap)
{
    {
        const int gi = blockIdx.x * SP_WG + threadIdx.x;
        if (zeroHits) for (int f = gi; f < db.nFiles; f += gridDim.x * SP_WG) zeroHits[f] = 0;   // IGD_HIP_FLAG_ZERO_FIRST
        if (zeroTotal && gi == 0) *zeroTotal = 0;
        if (gate == 0 && gi == 0) {
            ctl[CTL_NLONG + ((epoch + 1) & 1)] = 0;
            ctl[CTL_NFIX + ((epoch + 1) & 1)] = 0;
            ctl[CTL_BUDGET + ((epoch + 1) & 1)] = 0;
        ctl[CTL_NHEAVY + ((epoch + 1) & 1)] = 0;
        ctl[CTL_NHEAVYS + ((epoch + 1) & 1)] = 0;
        ctl[CTL_NFAR + ((epoch + 1) & 1)] = 0;
        }
    }
    if (gate != 0 && __builtin_amdgcn_readfirstlane(ctl[CTL_UNSORTED]) != gate) return;
    __shared__ uint32_t hist[SP_MAXC], wsum[SP_WG / IGD_WAVE];
    uint32_t *cur = hist;                                 // (a bucket's count is read, then its cursor written, by the one thread that owns the bucket)
    for (int b = threadIdx.x; b < nCoarse; b += SP_WG) hist[b] = 0;
    // "does tile t hold records" is asked once or twice per query, of tiles all over the genome: out of tileCnt[] that is a
    // 4-byte gather that moves a cache line per query (10^6 queries: 128 MB through the L2s -- 3/4 of this kernel's time went
    // into waiting for it); the same answers as one bit per tile are 24 KB for hg38 in 16 kbp tiles, staged in LDS
    // ... and so are the two contig tables (ctgStaged = nCtg when they fit): with them no load is left between a query's
    // arrival and its LDS atomics, and the queries themselves are all asked for up front -- behind the branches below the
    // thread's four would arrive one after the other
    extern __shared__ uint32_t sl_bits[];
    int32_t *sCtgN = (int32_t *)(sl_bits + bitsWords), *sCtgB = sCtgN + ctgStaged;
    int c_[SP_PER], qs_[SP_PER], qe_[SP_PER];
#pragma unroll
    for (int k = 0; k < SP_PER; k++) {                    // (asked for first: they come from memory, the tables below from an L2)
        const int i = blockIdx.x * SP_Q + k * SP_WG + threadIdx.x;
        const bool in = i < nq;
        c_[k] = in ? ichr[i] : -1; qs_[k] = in ? qs[i] : 0; qe_[k] = in ? qe[i] : 0;
    }
    for (int w = threadIdx.x; w < bitsWords; w += SP_WG) sl_bits[w] = db.tileBits[w];
    for (int c = threadIdx.x; c < ctgStaged; c += SP_WG) { sCtgN[c] = db.ctgNTile[c]; sCtgB[c] = db.ctgBase[c]; }
    __syncthreads();
    auto has_records = [&](int t) -> bool { return bitsWords ? ((sl_bits[t >> 5] >> (t & 31)) & 1u) != 0u : db.tileCnt[t] > 0; };
    // (query_span's last test -- rule NEST: an empty first tile ends the query, :468 -- is made here, from the bits)
    const int spanRule = (rule & ~0xff) | IGD_HIP_RULE_FLAT;
    const bool nestTest = (rule & 0xff) == IGD_HIP_RULE_NEST && db.vshift < 0;
    int gt0[SP_PER], ntl[SP_PER];
    if (FAST) {
        const int sh = db.shift;
#pragma unroll
        for (int k = 0; k < SP_PER; k++) {
            const int c = c_[k], s = qs_[k], e = qe_[k];
            const bool valid = (unsigned)c < (unsigned)db.nCtg;      // (i >= nq: c = -1)
            const int cc = valid ? c : 0;
            const int mT = sCtgN[cc] - 1;
            const int n1 = tile_shift(s, sh);
            int n2 = tile_shift((int)((unsigned)e - 1u), sh);
            bool ok = valid && n1 >= 0 && n1 <= mT;           // (:459-462)
            n2 = n2 > mT ? mT : n2;
            const int n = n2 > n1 ? n2 - n1 + 1 : 1;
            const int g = ok ? sCtgB[cc] + n1 : 0;
            // the four tiles from g on, one bit each (two words of the staged bits; one word of slack behind the last)
            const uint32_t w0 = sl_bits[g >> 5], w1 = sl_bits[(g >> 5) + 1];
            const uint32_t four = (uint32_t)(((((unsigned long long)w1) << 32) | w0) >> (g & 31)) & 0xFu;
            if ((rule & 0xff) == IGD_HIP_RULE_NEST) ok = ok && (four & 1u);     // (:468)
            const bool longQ = n > IGD_SHORT_TILES;
            const bool rare = ok && (longQ || (packed && e <= (int)((unsigned)n1 << sh)));   // walk_kind: WALK_ALL / WALK_FIRST
            const uint32_t live = (ok && !rare) ? (four & ((1u << (n & 7)) - 1u)) : 0u;      // (n <= 4 here)
            gt0[k] = g; ntl[k] = (int)live;
            if (rare) {
                const int i = blockIdx.x * SP_Q + k * SP_WG + threadIdx.x;
                longList[atomicAdd(&ctl[CTL_NLONG + (epoch & 1)], 1)] = make_int2(i, longQ ? WALK_ALL : WALK_FIRST);
                if (longQ) cover_tiles(db, ctl, 1, epoch, g + 1, g + n - 1);
            }
            if (live) {      // a coarse bucket holds >= 256 tiles: the query's <= 4 tiles lie in one bucket, or in two neighbours
                const int b0 = g >> shift, jb = ((b0 + 1) << shift) - g;
                const uint32_t lo = live & ((jb < 4 ? 1u << jb : 16u) - 1u);
                if (lo) atomicAdd(&hist[b0], (uint32_t)__popc(lo));
                if (live != lo) atomicAdd(&hist[b0 + 1], (uint32_t)__popc(live ^ lo));
            }
        }
    } else
#pragma unroll
    for (int k = 0; k < SP_PER; k++) {
        const int i = blockIdx.x * SP_Q + k * SP_WG + threadIdx.x;
        ntl[k] = 0; gt0[k] = 0;
        const int c = c_[k];
        if (c >= 0 && c < db.nCtg) {                      // (i >= nq: c = -1)
            const int s = qs_[k], e = qe_[k];
            int g, n;
            if (query_span_at(db, c, ctgStaged ? sCtgN[c] : db.ctgNTile[c], ctgStaged ? sCtgB[c] : db.ctgBase[c], s, e, spanRule, g, n) &&
                !(nestTest && !has_records(g))) {
                const int kind = walk_kind(db, s, e, n, packed);
                if (kind >= 0) {
                    longList[atomicAdd(&ctl[CTL_NLONG + (epoch & 1)], 1)] = make_int2(i, kind);
                    if (kind == WALK_ALL) cover_tiles(db, ctl, 1, epoch, g + 1, g + n - 1);   // first and last tile by the walk, the rest covered
                } else {
                    int live = 0;                         // bit j: tile g+j is not empty
                    for (int j = 0; j < n; j++) live |= (int)has_records(g + j) << j;
                    gt0[k] = g; ntl[k] = live;
                    for (int j = 0; j < n; j++)
                        if ((live >> j) & 1) atomicAdd(&hist[(g + j) >> shift], 1u);
                }
            }
        }
    }
    __syncthreads();
    {   // exclusive prefix over the buckets: thread t owns buckets 4t .. 4t+3 (SP_MAXC = 4 * SP_WG)
        const int b0 = threadIdx.x * (SP_MAXC / SP_WG);
        uint32_t c[SP_MAXC / SP_WG], sum = 0;
#pragma unroll
        for (int k = 0; k < SP_MAXC / SP_WG; k++) { c[k] = b0 + k < nCoarse ? hist[b0 + k] : 0u; sum += c[k]; }
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        uint32_t x = sum;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = (uint32_t)__shfl_up((int)x, o);
            if (lane >= o) x += y;
        }
        if (lane == 63) wsum[w] = x;
        __syncthreads();
        uint32_t run = x - sum;
        for (int k = 0; k < w; k++) run += wsum[k];
#pragma unroll
        for (int k = 0; k < SP_MAXC / SP_WG; k++) {
            if (b0 + k < nCoarse) {
                table[(size_t)blockIdx.x * nCoarse + b0 + k] = run | (c[k] << 16);
                if (c[k] >= SP_LONG && bucketLong) { bucketLong[b0 + k] = epoch; ctl[CTL_PILED] = epoch; }   // a quarter of this workgroup's queries in ONE bucket: a piled-up batch
                cur[b0 + k] = run;
            }
            run += c[k];
        }
    }
    __syncthreads();
    const size_t rb = (size_t)blockIdx.x * SP_CAP;
    // The tuples go to their places in the workgroup's region -- grouped by bucket, i.e. 64 lanes to 64 places: a store
    // instruction of 64 separate 12-byte pieces, 16 000 of them per 10^6 queries, was a third of this kernel.  The region is
    // put together in LDS (in the area of the tile bits, which nobody reads any more) and written out as it lies, 16 bytes
    // per lane; a workgroup with more pairs than the area holds (long queries: up to 4 pairs each) stores them one by one.
    uint32_t nPairs = 0;
    for (int k = 0; k < SP_WG / IGD_WAVE; k++) nPairs += wsum[k];
    const bool staged = nPairs <= (uint32_t)stageCap;
    SpTuple *stg = (SpTuple *)sl_bits;
    if (FAST) {
#pragma unroll
        for (int k = 0; k < SP_PER; k++) {
            const uint32_t live = (uint32_t)ntl[k];
            if (live) {
                const int g = gt0[k], b0 = g >> shift, jb = ((b0 + 1) << shift) - g;
                const uint32_t lo = live & ((jb < 4 ? 1u << jb : 16u) - 1u);
                uint32_t pLo = lo ? atomicAdd(&cur[b0], (uint32_t)__popc(lo)) : 0u;
                uint32_t pHi = live != lo ? atomicAdd(&cur[b0 + 1], (uint32_t)__popc(live ^ lo)) : 0u;
#pragma unroll
                for (int j = 0; j < IGD_SHORT_TILES; j++)
                    if ((live >> j) & 1u) {
                        const uint32_t pos = ((lo >> j) & 1u) ? pLo++ : pHi++;
                        SpTuple tu; tu.t = g + j; tu.s = qs_[k]; tu.e = qe_[k];
                        if (staged) stg[pos] = tu; else reg[rb + pos] = tu;
                    }
            }
        }
    } else
#pragma unroll
    for (int k = 0; k < SP_PER; k++)
        for (int live = ntl[k], j = 0; live; live >>= 1, j++)
            if (live & 1) {
                const int t = gt0[k] + j;
                const uint32_t pos = atomicAdd(&cur[t >> shift], 1u);
                SpTuple tu; tu.t = t; tu.s = qs_[k]; tu.e = qe_[k];
                if (staged) stg[pos] = tu; else reg[rb + pos] = tu;
            }
    if (staged) {
        __syncthreads();
        const uint4 *src = (const uint4 *)sl_bits;
        uint4 *dst = (uint4 *)(reg + rb);                 // (a region starts at a multiple of SP_CAP tuples: 16-byte aligned; the last piece may carry up to 12 bytes beyond the pairs -- inside the area, inside the region, never read)
        for (uint32_t k = threadIdx.x; k < (nPairs * 3 + 3) / 4; k += SP_WG) dst[k] = src[k];
    }
}

#define SP_ROWS 4     // table rows a thread keeps in flight
#ifndef SPF_FLY
#define SPF_FLY 8     // tuples a thread of the staged path keeps in flight
#endif
__device__ __forceinline__ void split_fine_whole(int b, int nT, int shift, int nCoarse, int nWG, const uint32_t *__restrict__ table,
                                                      const SpTuple *__restrict__ reg,
                                                      int32_t *__restrict__ pairN, int32_t *__restrict__ pairPos,
                                                      int2 *__restrict__ pairs, const int32_t *__restrict__ ctl, int gate,
                                                      int32_t *__restrict__ ctlw, int epoch, int32_t *__restrict__ heavy, int cap = 0)
{
    extern __shared__ uint32_t sp_lds[];
    const int F = 1 << shift;
    uint32_t *cnt = sp_lds, *start = sp_lds + F;
    __shared__ uint32_t wsum[SPF_WG / IGD_WAVE], baseSh;
    const int t0 = b << shift;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int f = threadIdx.x; f < F; f += SPF_WG) cnt[f] = 0;
    // The usual bucket (round 5): all of its segments are short and its pairs fit the workgroup's staging area (cap tuples
    // of LDS behind the counters).  One thread per segment walked its tuples one dependent load after the other, twice
    // (count, then place): ~11 memory round trips in a row at 3 waves per SIMD.  Now the segments are laid out flat -- every
    // thread expands its own into a list of tuple addresses in LDS -- and thread k fetches tuple k, k + 256, ... (SPF_FLY loads
    // in flight), counts it and keeps it in LDS, from where the second pass places it: the tuples are read from memory once,
    // in one or two round trips.
    int nStaged = -1;
    uint32_t *stT = start + F, *stS = stT + cap, *stE = stS + cap;   // staged tuples: tile - t0 (first: the tuple's address), qs, qe
    if (cap > 0 && nWG <= SPF_WG * SP_ROWS) {
        __shared__ uint32_t wsumN[SPF_WG / IGD_WAVE], wsumO[SPF_WG / IGD_WAVE];
        uint32_t e[SP_ROWS], mine = 0, offs = 0;
        int anyLong = 0;
#pragma unroll
        for (int r = 0; r < SP_ROWS; r++) { const int w = (int)threadIdx.x + r * SPF_WG; e[r] = w < nWG ? table[(size_t)w * nCoarse + b] : 0u; }
#pragma unroll
        for (int r = 0; r < SP_ROWS; r++) { mine += e[r] >> 16; offs += e[r] & 0xFFFFu; anyLong |= (e[r] >> 16) >= IGD_WAVE; }
        const uint32_t incl = (uint32_t)wave_inclusive_sum((int)mine);
        for (int o = 32; o > 0; o >>= 1) offs += (uint32_t)__shfl_xor((int)offs, o);
        if (lane == 63) { wsumN[wv] = incl; wsumO[wv] = offs; }
        anyLong = __syncthreads_or(anyLong);
        uint32_t at0 = incl - mine, N = 0, O = 0;
        for (int k = 0; k < SPF_WG / IGD_WAVE; k++) { if (k < wv) at0 += wsumN[k]; N += wsumN[k]; O += wsumO[k]; }
        if (!anyLong && N <= (uint32_t)cap) {
            nStaged = (int)N;
            if (threadIdx.x == 0) baseSh = O;
#pragma unroll
            for (int r = 0; r < SP_ROWS; r++) {
                const unsigned at = (unsigned)((int)threadIdx.x + r * SPF_WG) * (unsigned)SP_CAP + (e[r] & 0xFFFFu);
                for (int j = 0, c = (int)(e[r] >> 16); j < c; j++) stT[at0++] = at + (unsigned)j;
            }
            __syncthreads();
            // (eight in flight: a bucket of the benchmark's batch holds ~1400 tuples, 5-6 per thread -- in rounds of four the second
            // round trip, to tuples another XCD has just written, was a fifth of the workgroup's time)
            for (int k0 = 0; k0 < nStaged; k0 += SPF_FLY * SPF_WG) {
                SpTuple tu[SPF_FLY];
#pragma unroll
                for (int u = 0; u < SPF_FLY; u++) {       // (no branch around a load: behind one the compiler waits for each load before it asks for the next)
                    const int k = k0 + u * SPF_WG + (int)threadIdx.x;
                    tu[u] = reg[stT[k < nStaged ? k : 0]];
                }
#pragma unroll
                for (int u = 0; u < SPF_FLY; u++) {
                    const int k = k0 + u * SPF_WG + (int)threadIdx.x;
                    if (k < nStaged) { const uint32_t f = (uint32_t)(tu[u].t - t0); atomicAdd(&cnt[f], 1u); stT[k] = f; stS[k] = (uint32_t)tu[u].s; stE[k] = (uint32_t)tu[u].e; }
                }
            }
        }
    }
    if (nStaged < 0)
    {   // first pair of this bucket = pairs of all earlier buckets = the sum over the table's rows of each row's own
        // exclusive prefix at this column (the low half of the entries this workgroup reads anyway): no kernel of column sums
        uint32_t x = 0;
        for (int w = threadIdx.x; w < nWG; w += SPF_WG) x += table[(size_t)w * nCoarse + b] & 0xFFFFu;
        for (int o = 32; o > 0; o >>= 1) x += (uint32_t)__shfl_down((int)x, o);
        if (lane == 0) wsum[wv] = x;
    }
    __syncthreads();
    if (nStaged < 0 && threadIdx.x == 0) { uint32_t t = 0; for (int k = 0; k < SPF_WG / IGD_WAVE; k++) t += wsum[k]; baseSh = t; }
    // Every (workgroup of k_split_local, this bucket) segment of pairs is walked: short ones by the thread that looked
    // them up, long ones (>= 64 pairs: queries that come in sorted runs put a workgroup's 4096 queries into one or two
    // buckets, and ONE thread walked them all -- 5 x the time of scattered queries) by the whole wave.
    auto walk = [&](auto fn) {
        for (int wb = 0; wb < nWG; wb += SPF_WG * SP_ROWS) {
            const int w0 = wb + (int)threadIdx.x;
            uint32_t e[SP_ROWS];
#pragma unroll
            for (int r = 0; r < SP_ROWS; r++) { const int w = w0 + r * SPF_WG; e[r] = w < nWG ? table[(size_t)w * nCoarse + b] : 0u; }
#pragma unroll
            for (int r = 0; r < SP_ROWS; r++) {
                const unsigned at = (unsigned)(w0 + r * SPF_WG) * (unsigned)SP_CAP + (e[r] & 0xFFFFu);   // (< 2^24 queries x 4 pairs: fits 32 bits)
                const int c = (int)(e[r] >> 16);
                const bool longSeg = c >= IGD_WAVE;
                if (!longSeg) for (int j = 0; j < c; j++) fn(reg[(size_t)at + j]);
                unsigned long long m = __ballot(longSeg);
                while (m) {
                    const int src = __builtin_ctzll(m);
                    m &= m - 1;
                    const unsigned at2 = (unsigned)__builtin_amdgcn_readlane((int)at, src);
                    const int c2 = __builtin_amdgcn_readlane(c, src);
                    // four tuples in flight per lane (one workgroup owns the bucket: 10^6 queries piled up in a few tiles are
                    // ALL its pairs, and a load waited for per tuple made that 2.6 ms)
                    int j = lane;
                    for (; j + 3 * IGD_WAVE < c2; j += 4 * IGD_WAVE) {
                        const SpTuple a0 = reg[(size_t)at2 + j], a1 = reg[(size_t)at2 + j + IGD_WAVE];
                        const SpTuple a2 = reg[(size_t)at2 + j + 2 * IGD_WAVE], a3 = reg[(size_t)at2 + j + 3 * IGD_WAVE];
                        fn(a0); fn(a1); fn(a2); fn(a3);
                    }
                    for (; j < c2; j += IGD_WAVE) fn(reg[(size_t)at2 + j]);
                }
            }
        }
    };
    if (nStaged < 0) walk([&](const SpTuple &tu) { atomicAdd(&cnt[tu.t - t0], 1u); });
    __syncthreads();
    {   // exclusive prefix over the bucket's tiles: thread t owns F/SPF_WG consecutive tiles
        const int per0 = F >= SPF_WG ? F / SPF_WG : 1, f0 = threadIdx.x * per0;
        const int per = f0 < F ? per0 : 0;               // (more threads than tiles: the rest own none)
        uint32_t sum = 0;
        for (int k = 0; k < per; k++) sum += cnt[f0 + k];
        uint32_t x = sum;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = (uint32_t)__shfl_up((int)x, o);
            if (lane >= o) x += y;
        }
        __syncthreads();
        if (lane == 63) wsum[wv] = x;
        __syncthreads();
        uint32_t run = baseSh + x - sum;
        for (int k = 0; k < wv; k++) run += wsum[k];
        for (int k = 0; k < per; k++) {
            const uint32_t c = cnt[f0 + k];
            start[f0 + k] = run;
            if (t0 + f0 + k < nT) {
                pairPos[t0 + f0 + k] = (int32_t)(run + c);
                int32_t pn = (int32_t)c;
                if (heavy && c > IGD_HEAVY_PAIRS) {       // too many pairs for one wave: heavy_bucket_body shares the tile out
                    const int at = atomicAdd(&ctlw[CTL_NHEAVY + (epoch & 1)], 1);
                    if (at < IGD_HEAVY_MAX) { heavy[at] = t0 + f0 + k; pn = -pn; }   // negative: "not yours" for igd_scan_tiles
                }
                pairN[t0 + f0 + k] = pn;
            }
            run += c;
        }
    }
    __syncthreads();
    if (nStaged >= 0) {       // (placed straight from LDS; putting the bucket's pairs in order in LDS first and writing them out side by side
                              // -- what pays in k_split_local -- cost 3 us here: the places of one bucket lie within a dozen KB)
        for (int k = (int)threadIdx.x; k < nStaged; k += SPF_WG) {
            const uint32_t pos = atomicAdd(&start[stT[k]], 1u);
            pairs[pos] = make_int2((int)stS[k], (int)stE[k]);
        }
        return;
    }
    walk([&](const SpTuple &tu) {
        const uint32_t pos = atomicAdd(&start[tu.t - t0], 1u);
        pairs[pos] = make_int2(tu.s, tu.e);
    });
}

__global__ __launch_bounds__(SPF_WG) void k_split_fine(int nT, int shift, int nCoarse, int nWG, const uint32_t *__restrict__ table,
                                                      const SpTuple *__restrict__ reg,
                                                      int32_t *__restrict__ pairN, int32_t *__restrict__ pairPos,
                                                      int2 *__restrict__ pairs, const int32_t *__restrict__ ctl, int gate,
                                                      int32_t *__restrict__ ctlw, int epoch, int32_t *__restrict__ heavy, int cap)
{
    if (gate != 0 && __builtin_amdgcn_readfirstlane(ctl[CTL_UNSORTED]) != gate) return;
    split_fine_whole((int)blockIdx.x, nT, shift, nCoarse, nWG, table, reg, pairN, pairPos, pairs, ctl, gate, ctlw, epoch, heavy, cap);
}

// Several workgroups per coarse bucket for the buckets of a PILED-UP batch (round 4).  One workgroup per bucket walks ALL pairs
// of the bucket twice: an unordered batch piled up in a few tiles puts 10^6 pairs into one bucket, and that one workgroup
// streams 12 MB of tuples at what a single CU keeps in flight -- 1.7 ms where the whole batch otherwise takes 0.13.  A bucket
// in which k_split_local has seen a long segment (bucketLong[b] == epoch) is shared by SPF_S workgroups of SPF_W waves, each
// wave taking every (SPF_S * SPF_W)-th segment, and the step that needs all of them -- where a tile's pairs start -- sits
// between two kernels: _a counts (bucket, share, tile), _b adds the shares up, places its own and scatters.  Every other
// bucket is grouped by its first workgroup in _a exactly as before (split_fine_whole) and costs the other seven one load.
#ifndef SPF_S
#define SPF_S 8                // workgroups per piled-up coarse bucket ...
#endif
#define SPF_W (SPF_WG / IGD_WAVE)   // ... of this many waves each
template <typename FN>
__device__ __forceinline__ void split_walk_share(int nWG, int nCoarse, int b, int part, int lane, const uint32_t *__restrict__ table,
                                                 const SpTuple *__restrict__ reg, FN fn)
{
    constexpr int P = SPF_S * SPF_W;
    for (int wb = part; wb < nWG; wb += P * IGD_WAVE) {
        const int w = wb + lane * P;
        const uint32_t e = w < nWG ? table[(size_t)w * nCoarse + b] : 0u;
        const unsigned at = (unsigned)w * (unsigned)SP_CAP + (e & 0xFFFFu);
        const int c = (int)(e >> 16);
        const bool longSeg = c >= IGD_WAVE;
        if (!longSeg) for (int j = 0; j < c; j++) fn(reg[(size_t)at + j]);
        unsigned long long m = __ballot(longSeg);
        while (m) {
            const int src = __builtin_ctzll(m);
            m &= m - 1;
            const unsigned at2 = (unsigned)__builtin_amdgcn_readlane((int)at, src);
            const int c2 = __builtin_amdgcn_readlane(c, src);
            int j = lane;
            for (; j + 3 * IGD_WAVE < c2; j += 4 * IGD_WAVE) {
                const SpTuple a0 = reg[(size_t)at2 + j], a1 = reg[(size_t)at2 + j + IGD_WAVE];
                const SpTuple a2 = reg[(size_t)at2 + j + 2 * IGD_WAVE], a3 = reg[(size_t)at2 + j + 3 * IGD_WAVE];
                fn(a0); fn(a1); fn(a2); fn(a3);
            }
            for (; j < c2; j += IGD_WAVE) fn(reg[(size_t)at2 + j]);
        }
    }
}

__global__ __launch_bounds__(SPF_WG) void k_split_fine_a(int nT, int shift, int nCoarse, int nWG, const uint32_t *__restrict__ table,
                                                        const SpTuple *__restrict__ reg, uint32_t *__restrict__ sub,
                                                        uint32_t *__restrict__ bucketBase, const int32_t *__restrict__ bucketLong,
                                                        int32_t *__restrict__ pairN, int32_t *__restrict__ pairPos,
                                                        int2 *__restrict__ pairs, const int32_t *__restrict__ ctl, int gate,
                                                        int32_t *__restrict__ ctlw, int epoch, int32_t *__restrict__ heavy, int cap)
{
    if (gate != 0 && __builtin_amdgcn_readfirstlane(ctl[CTL_UNSORTED]) != gate) return;
    const int b = blockIdx.x % nCoarse, share = blockIdx.x / nCoarse;   // (workgroups go round-robin to the 8 XCDs: share = blockIdx & 7 put every first share -- all the work of an ordinary batch -- on ONE of them: 121 instead of 25 us)
    if (__builtin_amdgcn_readfirstlane(bucketLong[b]) != epoch) {      // the usual bucket: one workgroup, one kernel
        if (share == 0) split_fine_whole(b, nT, shift, nCoarse, nWG, table, reg, pairN, pairPos, pairs, ctl, gate, ctlw, epoch, heavy, cap);
        return;
    }
    extern __shared__ uint32_t sp_lds[];
    __shared__ uint32_t wsumA[SPF_W];
    const int F = 1 << shift;
    uint32_t *cnt = sp_lds;
    const int t0 = b << shift, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int f = threadIdx.x; f < F; f += SPF_WG) cnt[f] = 0;
    if (share == 0) {           // pairs of all earlier buckets: the sum over the table's rows of each row's own exclusive prefix at this column
        uint32_t x = 0;
        for (int w = threadIdx.x; w < nWG; w += SPF_WG) x += table[(size_t)w * nCoarse + b] & 0xFFFFu;
        for (int o = 32; o > 0; o >>= 1) x += (uint32_t)__shfl_xor((int)x, o);
        if (lane == 0) wsumA[wv] = x;
    }
    __syncthreads();
    if (share == 0 && threadIdx.x == 0) { uint32_t t = 0; for (int k = 0; k < SPF_W; k++) t += wsumA[k]; bucketBase[b] = t; }
    split_walk_share(nWG, nCoarse, b, share * SPF_W + wv, lane, table, reg, [&](const SpTuple &tu) { atomicAdd(&cnt[tu.t - t0], 1u); });
    __syncthreads();
    uint32_t *mine = sub + (((size_t)b * SPF_S + share) << shift);
    for (int f = threadIdx.x; f < F; f += SPF_WG) mine[f] = cnt[f];
}

__global__ __launch_bounds__(SPF_WG) void k_split_fine_b(int nT, int shift, int nCoarse, int nWG, const uint32_t *__restrict__ table,
                                                        const SpTuple *__restrict__ reg, const uint32_t *__restrict__ sub,
                                                        const uint32_t *__restrict__ bucketBase, const int32_t *__restrict__ bucketLong,
                                                        int32_t *__restrict__ pairN, int32_t *__restrict__ pairPos,
                                                        int2 *__restrict__ pairs, const int32_t *__restrict__ ctl, int gate,
                                                        int32_t *__restrict__ ctlw, int epoch, int32_t *__restrict__ heavy)
{
    if (gate != 0 && __builtin_amdgcn_readfirstlane(ctl[CTL_UNSORTED]) != gate) return;
    if (__builtin_amdgcn_readfirstlane(ctl[CTL_PILED]) != epoch) return;  // no piled-up bucket in this batch (the usual case: an empty launch of a few hundred workgroups)
    extern __shared__ uint32_t sp_lds[];
    __shared__ uint32_t wsumB[SPF_W], carry;
    const int F = 1 << shift;
    uint32_t *start = sp_lds;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int item = blockIdx.x; item < nCoarse * SPF_S; item += gridDim.x) {
    const int b = item % nCoarse, share = item / nCoarse;
    if (__builtin_amdgcn_readfirstlane(bucketLong[b]) != epoch) continue;  // done by k_split_fine_a
    const int t0 = b << shift;
    __syncthreads();                                     // (the previous item's LDS is done with)
    if (threadIdx.x == 0) carry = bucketBase[b];
    __syncthreads();
    // per tile: all shares' pairs (where the next tile starts) and those of the shares before this one (where this one's go)
    for (int f0 = 0; f0 < F; f0 += SPF_WG) {
        const int f = f0 + (int)threadIdx.x;
        uint32_t all = 0, before = 0;
        if (f < F) {
            const uint32_t *col = sub + (((size_t)b * SPF_S) << shift) + f;
#pragma unroll
            for (int k = 0; k < SPF_S; k++) { const uint32_t c = col[(size_t)k << shift]; all += c; before += k < share ? c : 0u; }
        }
        uint32_t x = all;                                 // inclusive prefix over the round's tiles: inside the wave, then over the waves
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)x, o); if (lane >= o) x += y; }
        if (lane == 63) wsumB[wv] = x;
        __syncthreads();
        uint32_t first = carry + x - all;
        for (int k = 0; k < wv; k++) first += wsumB[k];
        if (f < F) {
            start[f] = first + before;
            if (share == 0 && t0 + f < nT) {
                pairPos[t0 + f] = (int32_t)(first + all);
                int32_t pn = (int32_t)all;
                if (heavy && all > IGD_HEAVY_PAIRS) {     // too many pairs for one wave: heavy_bucket_body shares the tile out
                    const int at = atomicAdd(&ctlw[CTL_NHEAVY + (epoch & 1)], 1);
                    if (at < IGD_HEAVY_MAX) { heavy[at] = t0 + f; pn = -pn; }
                }
                pairN[t0 + f] = pn;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) { uint32_t t = carry; for (int k = 0; k < SPF_W; k++) t += wsumB[k]; carry = t; }
        __syncthreads();
    }
    split_walk_share(nWG, nCoarse, b, share * SPF_W + wv, lane, table, reg, [&](const SpTuple &tu) {
        const uint32_t pos = atomicAdd(&start[tu.t - t0], 1u);
        pairs[pos] = make_int2(tu.s, tu.e);
    });
    }
}

// step 2: exclusive scan of pairCnt -> pairPos (two kernels, no inter-block protocol).  The
// apply kernel also moves the counts to pairN and leaves pairCnt zeroed for the next batch.
__global__ __launch_bounds__(IGD_SCAN_BLOCK) void k_scan_block_sums(const int32_t *__restrict__ in,
                                                                    int n, int32_t *__restrict__ blockSums,
                                                                    const int32_t *__restrict__ ctl, int gate)
{
    if (gate != 0 && ctl[CTL_UNSORTED] != gate) return;
    __shared__ int32_t red[IGD_SCAN_BLOCK / IGD_WAVE];
    int base = blockIdx.x * IGD_SCAN_TILE + threadIdx.x * IGD_SCAN_ITEMS;
    int s = 0;
#pragma unroll
    for (int k = 0; k < IGD_SCAN_ITEMS; k++)
        if (base + k < n) s += in[base + k];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < IGD_SCAN_BLOCK / IGD_WAVE; w++) t += red[w];
        blockSums[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(IGD_SCAN_BLOCK) void k_scan_apply(int32_t *__restrict__ in, int n,
                                                               const int32_t *__restrict__ blockSums,
                                                               int32_t *__restrict__ out,
                                                               int32_t *__restrict__ copy,
                                                               const int32_t *__restrict__ ctl, int gate)
{
    if (gate != 0 && ctl[CTL_UNSORTED] != gate) return;
    __shared__ int32_t red[IGD_SCAN_BLOCK / IGD_WAVE];
    __shared__ int32_t wsum[IGD_SCAN_BLOCK / IGD_WAVE];
    // prefix of the earlier blocks' sums (every block recomputes it; a few hundred values)
    int pre = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += IGD_SCAN_BLOCK) pre += blockSums[b];
    for (int o = 32; o > 0; o >>= 1) pre += __shfl_down(pre, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = pre;
    // local items
    int base = blockIdx.x * IGD_SCAN_TILE + threadIdx.x * IGD_SCAN_ITEMS;
    int v[IGD_SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int k = 0; k < IGD_SCAN_ITEMS; k++) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        s += v[k];
    }
    // inclusive scan of s across the wave
    int inc = s;
    int lane = threadIdx.x & 63;
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    int blockPre = 0;
    for (int w = 0; w < IGD_SCAN_BLOCK / IGD_WAVE; w++) blockPre += red[w];
    int wavePre = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) wavePre += wsum[w];
    int run = blockPre + wavePre + inc - s;
#pragma unroll
    for (int k = 0; k < IGD_SCAN_ITEMS; k++) {
        if (base + k < n) {
            out[base + k] = run;
            if (copy) { copy[base + k] = v[k]; in[base + k] = 0; }
        }
        run += v[k];
    }
}

// step 3: scatter (qs,qe) of every pair to its tile's slot range.  After this kernel
// pairPos[t] is the END of tile t's range.
__global__ void k_scatter_pairs(DbView db, const int32_t *__restrict__ ichr,
                                const int32_t *__restrict__ qs, const int32_t *__restrict__ qe,
                                int nq, int rule, int packed, int32_t *__restrict__ pairPos,
                                void *__restrict__ pairs, const int32_t *__restrict__ ctl, int gate)
{
    if (gate != 0 && ctl[CTL_UNSORTED] != gate) return;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    int gt0, ntl;
    int s = qs[i], e = qe[i];
    if (!query_span(db, ichr[i], s, e, rule, gt0, ntl)) return;
    if (walk_kind(db, s, e, ntl, packed) >= 0) return;
    for (int k = 0; k < ntl; k++) {
        if (db.tileCnt[gt0 + k] > 0) {
            int p = atomicAdd(&pairPos[gt0 + k], 1);
            ((int2 *)pairs)[p] = make_int2(s, e);
        }
    }
}
