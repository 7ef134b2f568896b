// engine/hitmap_dev.hpp -- part of igd_hip.hip (included there once; not a stand-alone header).
// `-m` hit map kernel
// ------------------------------------------------------------------------------------------
// `-m`: dataset x dataset hit map (getMap src/igd_search.c:772-826, getMap_v :829-886;
// SURVEY 8f row f3).  Tile by tile, every record j is a query against its own tile:
//     hitmap[idx_j][idx_i]++   for every i with  start_i < end_j && end_i > start_j
//                                                && (start_j >= bd || start_i >= bd)   [&& value_j,value_i > v]
// (the last clause is the reference's tS skip, :803-804: two records that both begin before the
// tile were already paired in an earlier tile; its maxE early exit, :791-796/:811, only shortens
// the scan).  One workgroup per tile; each thread owns a record j of a 256-record slice and walks
// the tile's records, staged 256 at a time in LDS (broadcast reads), leaving as soon as the sorted
// starts pass every end of the slice.  Counters are the reference's uint32.
#define IGD_MAP_WG 256
template <bool USE_V>
__global__ __launch_bounds__(IGD_MAP_WG) void igd_hitmap_tiles(DbView db, int v, uint32_t *__restrict__ hitmap,
                                                               u64 *__restrict__ total)
{
    __shared__ int32_t sS[IGD_MAP_WG], sE[IGD_MAP_WG], sX[IGD_MAP_WG], sV[IGD_MAP_WG];
    __shared__ u64 red[IGD_MAP_WG / IGD_WAVE];
    u64 found = 0;
    for (int t = blockIdx.x; t < db.nT; t += gridDim.x) {
        const int cnt = db.tileCnt[t];
        if (cnt == 0) continue;
        const int64_t off = db.tileOff[t];
        const int tb = db.tileBd[t];
        const int bd = tb == INT_MIN ? 0 : tb;                       // getMap uses nbp*n1, also for tile 0
        for (int jb = 0; jb < cnt; jb += IGD_MAP_WG) {
            const int j = jb + (int)threadIdx.x;
            bool act = j < cnt;
            const int qs = act ? db.start[off + j] : 0;
            const int qe = act ? db.end[off + j] : INT_MIN;
            const int jj = act ? db.idx[off + j] : 0;
            if (USE_V && act) act = db.value[off + j] > v;
            const bool prefix = qs < bd;
            uint32_t *row = hitmap + (size_t)jj * (size_t)db.nFiles;
            for (int ib = 0; ib < cnt; ib += IGD_MAP_WG) {
                const int i = ib + (int)threadIdx.x;
                __syncthreads();
                if (i < cnt) {
                    sS[threadIdx.x] = db.start[off + i];
                    sE[threadIdx.x] = db.end[off + i];
                    sX[threadIdx.x] = db.idx[off + i];
                    if (USE_V) sV[threadIdx.x] = db.value[off + i];
                }
                __syncthreads();
                // sorted by start: once the first start of this stage is >= every end of the slice, done
                if (!__syncthreads_or(act && qe > sS[0])) break;
                const int nB = cnt - ib < IGD_MAP_WG ? cnt - ib : IGD_MAP_WG;
                if (act) {
                    for (int k = 0; k < nB; k++) {
                        const int s = sS[k];
                        if (s >= qe) break;                          // this thread's partners end here
                        bool hit = sE[k] > qs && (!prefix || s >= bd);
                        if (USE_V) hit = hit && sV[k] > v;
                        if (hit) { atomicAdd(&row[sX[k]], 1u); found++; }
                    }
                }
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) found += __shfl_down(found, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = found;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 s = 0;
        for (int w = 0; w < IGD_MAP_WG / IGD_WAVE; w++) s += red[w];
        if (s) atomicAdd(total, s);
    }
}
