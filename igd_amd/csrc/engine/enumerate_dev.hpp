// engine/enumerate_dev.hpp -- part of igd_hip.hip (included there once; not a stand-alone header).
// `-f` enumeration kernels
// ------------------------------------------------------------------------------------------
// `-f` enumeration (get_overlaps_f1/_f0, src/igd_search.c:537-620,114-200).  A wave walks the tiles
// of its query in ascending order and each tile from its LAST 64 records to the first, so that
// record indices come out descending, as the reverse scans at :575-579 and :608-612 emit them.
//   pass COUNT: qcount[q] = overlaps of query q
//   scan      : qoff = exclusive scan of qcount
//   pass FILL : the rank-th overlap of query q goes to out[qoff[q] + rank]
// SEQ (Seqpare, src/igd_search.c:253-352): the record's place is taken by what seq_overlaps stores for it:
// start <- idx_g (index of the record inside its tile), end <- the bits of the float similarity
// sm = st / (qlen + rlen - st), computed in single precision in the reference's order of operations.
__device__ __forceinline__ int seq_similarity_bits(int qs, int qe, int s, int e)
{
    const float qlen = (float)(qe - qs);
    const float st = (float)((qe < e ? qe : e) - (qs > s ? qs : s));
    const float rlen = (float)(e - s);
    return __float_as_int(__fdiv_rn(st, __fsub_rn(__fadd_rn(qlen, rlen), st)));
}

// Query-major: one wave per query, so that a contiguous range of queries is a contiguous range of the
// output -- which is what lets the host side stream the result out in chunks while later chunks are
// still being produced (the path is bound by the 16 bytes per overlap that cross PCIe, not by these
// kernels).  No grouping step at all: with queries in any order a tile's records are simply re-read from
// L2 / HBM (2.2 KB per (query, tile) pair; < 1 ms per 10^6 queries either way).
//   COUNT (FILL = false): qcount[q] = overlaps of query q            (all its tiles)
//   FILL                : out[qoff[q] - base0 + rank] = the overlaps of queries [qa, qb), reference order
//   PACK8 (with FILL)   : 8 bytes per overlap -- start | (end - start) << idxBits | idx (igd_hip_hit8; `out` is an array of those)
template <bool FILL, bool SEQ = false, bool PACK8 = false>
__global__ __launch_bounds__(256) void igd_enum_queries(
    DbView db, const int32_t *__restrict__ q_ichr, const int32_t *__restrict__ q_qs,
    const int32_t *__restrict__ q_qe, int qa, int qb, int64_t *__restrict__ qcount,
    const int64_t *__restrict__ qoff, int64_t base0, igd_hip_hit *__restrict__ out, int idxBits = 0)
{
    const int lane = threadIdx.x & 63;
    const int gwave = blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const u64 above = (lane == 63) ? 0ull : (~0ull << (lane + 1));   // lanes with a higher record index

    for (int q = qa + gwave; q < qb; q += nwaves) {
        const int qs = __builtin_amdgcn_readfirstlane(q_qs[q]);
        const int qe = __builtin_amdgcn_readfirstlane(q_qe[q]);
        const int cc = __builtin_amdgcn_readfirstlane(q_ichr[q]);
        int64_t cnt = 0;
        int gt0, ntl;
        if (query_span(db, cc, qs, qe, IGD_HIP_RULE_NEST, gt0, ntl)) {
            gt0 = __builtin_amdgcn_readfirstlane(gt0);
            ntl = __builtin_amdgcn_readfirstlane(ntl);
            const int64_t base = FILL ? qoff[q] - base0 : 0;
            for (int k = 0; k < ntl; k++) {
                const int t = gt0 + k;
                const int tcnt = __builtin_amdgcn_readfirstlane(db.tileCnt[t]);
                if (tcnt == 0) continue;
                const int lob = (k == 0) ? INT_MIN : __builtin_amdgcn_readfirstlane(db.tileBd[t]);
                const int64_t toff = db.tileOff[t];
                // records from the end of the tile towards the front, 64 at a time
                for (int hi = tcnt; hi > 0; hi -= IGD_WAVE) {
                    const int i = hi - IGD_WAVE + lane;           // lane 63 = highest index of this step
                    const bool ok = i >= 0;
                    const int s = ok ? db.start[toff + i] : INT_MAX;
                    const int e = ok ? db.end[toff + i] : INT_MIN;
                    const bool hit = (s < qe) & (s >= lob) & (e > qs);
                    const u64 m = __ballot(hit);
                    if (FILL && PACK8 && hit) {
                        const uint32_t hi32 = ((uint32_t)(e - s) << idxBits) | (uint32_t)db.idx[toff + i];
                        ((uint2 *)out)[base + cnt + __popcll(m & above)] = make_uint2((uint32_t)s, hi32);
                    } else
                    if (FILL && hit) {
                        igd_hip_hit h;
                        h.q = q; h.idx = db.idx[toff + i]; h.start = s; h.end = e;
                        if (SEQ) { h.start = i; h.end = seq_similarity_bits(qs, qe, s, e); }
                        out[base + cnt + __popcll(m & above)] = h;
                    }
                    cnt += __popcll(m);
                    // all starts in this step are below lob => so is everything before it
                    const int smax = __builtin_amdgcn_readlane(s, 63);
                    if (smax < lob) break;
                }
            }
        }
        if (!FILL && lane == 0) qcount[q] = cnt;
    }
}

// exclusive scan of int64 per-query counts -> qoff[0..n] (two kernels, like the tile scan)
__global__ __launch_bounds__(IGD_SCAN_BLOCK) void k_scan64_sums(const int64_t *__restrict__ in, int n,
                                                                int64_t *__restrict__ blockSums)
{
    __shared__ int64_t red[IGD_SCAN_BLOCK / IGD_WAVE];
    const int base = blockIdx.x * IGD_SCAN_TILE + threadIdx.x * IGD_SCAN_ITEMS;
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < IGD_SCAN_ITEMS; k++)
        if (base + k < n) s += in[base + k];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t t = 0;
        for (int w = 0; w < IGD_SCAN_BLOCK / IGD_WAVE; w++) t += red[w];
        blockSums[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(IGD_SCAN_BLOCK) void k_scan64_apply(const int64_t *__restrict__ in, int n,
                                                                 const int64_t *__restrict__ blockSums,
                                                                 int64_t *__restrict__ out /* n+1 */)
{
    __shared__ int64_t red[IGD_SCAN_BLOCK / IGD_WAVE];
    __shared__ int64_t wsum[IGD_SCAN_BLOCK / IGD_WAVE];
    int64_t pre = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += IGD_SCAN_BLOCK) pre += blockSums[b];
    for (int o = 32; o > 0; o >>= 1) pre += __shfl_down(pre, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = pre;
    const int base = blockIdx.x * IGD_SCAN_TILE + threadIdx.x * IGD_SCAN_ITEMS;
    int64_t v[IGD_SCAN_ITEMS];
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < IGD_SCAN_ITEMS; k++) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        s += v[k];
    }
    int64_t inc = s;
    const int lane = threadIdx.x & 63;
    for (int o = 1; o < 64; o <<= 1) {
        int64_t t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    int64_t run = inc - s;
    for (int w = 0; w < IGD_SCAN_BLOCK / IGD_WAVE; w++) run += red[w];
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) run += wsum[w];
#pragma unroll
    for (int k = 0; k < IGD_SCAN_ITEMS; k++) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
        if (base + k == n - 1) out[n] = run;             // the grand total closes the offsets
    }
}

// the largest (uint32_t)(end - start) over all records: whether (length, idx) fit one word of igd_hip_hit8 (-f in 8 bytes per overlap)
__global__ __launch_bounds__(256) void k_max_len(const int32_t *__restrict__ start, const int32_t *__restrict__ end, int64_t n, unsigned int *__restrict__ out)
{
    unsigned int m = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const unsigned int d = (unsigned int)(end[i] - start[i]);
        m = d > m ? d : m;
    }
    for (int o = 32; o > 0; o >>= 1) { const unsigned int t = __shfl_down(m, o); m = t > m ? t : m; }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
