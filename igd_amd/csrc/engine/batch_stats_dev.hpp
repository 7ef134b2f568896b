// engine/batch_stats_dev.hpp -- part of igd_hip.hip (included there once; not a stand-alone header).
// instrumentation: terms of the algorithmic byte model
// ------------------------------------------------------------------------------------------
// instrumentation: exact terms of the algorithmic byte model, thread per query.
__device__ __forceinline__ int lower_bound_start(const int32_t *s, int n, int key)
{
    int lo = 0, hi = n;          // first index with s[i] >= key
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (s[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ void k_batch_stats(DbView db, const int32_t *__restrict__ ichr,
                              const int32_t *__restrict__ qs, const int32_t *__restrict__ qe, int nq,
                              int rule, u64 *__restrict__ acc /* queries,pairs,S,B */)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    u64 nqv = 0, pairs = 0, S = 0, B = 0;
    if (i < nq) {
        int gt0, ntl;
        int c = ichr[i];
        // "reached the tile logic": valid contig and n1 in range (before the NEST test)
        if (c >= 0 && c < db.nCtg) {
            int n1 = qs[i] / db.nbp;
            if (n1 >= 0 && n1 <= db.ctgNTile[c] - 1) nqv = 1;
        }
        if (query_span(db, c, qs[i], qe[i], rule, gt0, ntl)) {
            for (int k = 0; k < ntl; k++) {
                int t = gt0 + k;
                int cnt = db.tileCnt[t];
                if (cnt == 0) continue;
                const int32_t *s = db.start + db.tileOff[t];
                if (!(qe[i] > s[0])) continue;
                int hi = lower_bound_start(s, cnt, qe[i]);
                int lo = (k == 0) ? 0 : lower_bound_start(s, cnt, db.tileBd[t]);
                pairs++;
                S += hi > lo ? (u64)(hi - lo) : 0;
                int b = 0;
                while ((1ll << b) < (long long)cnt + 1) b++;
                B += b;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        nqv += __shfl_down(nqv, o); pairs += __shfl_down(pairs, o);
        S += __shfl_down(S, o); B += __shfl_down(B, o);
    }
    if ((threadIdx.x & 63) == 0) {
        if (nqv) atomicAdd(&acc[0], nqv);
        if (pairs) atomicAdd(&acc[1], pairs);
        if (S) atomicAdd(&acc[2], S);
        if (B) atomicAdd(&acc[3], B);
    }
}
