// engine/compact_image.hpp -- part of igd_hip.hip (included there once; not a stand-alone header).
// k_pack_units: the 6-byte tile-relative image and its unit descriptors
// ------------------------------------------------------------------------------------------
// Compact image.  Everything the COUNT of a (query, tile) pair depends on is relative to the
// tile: with T = tile start, W = tile width, a record of tile j is reduced to
//     s' = start < T ? 0 : start - T + 1        in [0, W]    (0: "starts before this tile")
//     e' = min(end - T, W)                       in [1, W]    (W: "reaches the tile's end")
// and a query visiting the tile to
//     qe' = min(qe - T, W) + 1,   qs' = first ? max(qs - T + 1, 1) : 1,   lob' = first ? 0 : 1
// so that   lob' <= s' < qe'  &&  e' >= qs'   <=>   lob <= start < qe  &&  end > qs
// for every query with qe > T (the conditions of SURVEY App. B.3; proof in DESIGN.md).
// Stored word:  (65535 - s') | e' << 16.  With the query word  (65536 - qe') | qs' << 16  the
// test  s' < qe' && e' >= qs'  is "both 16-bit halves >= the query's halves": one v_pk_max_u16
// and one compare.  The remaining condition s' >= lob' only excludes records that start before
// the tile (s' = 0) from queries for which this is not the first tile; since EVERY such query
// matches EVERY such record on the other two conditions, it is applied once per unit as a
// correction (hits -= number of non-first queries) instead of once per record and query.  A
// first-tile query with qe <= T (an inverted query reaching back over the tile start) is the one
// case that needs the exact starts: the grouping kernels list it for k_exact_walk (WALK_FIRST).
// 6 bytes per record (4 + 2; 8 with the 16-bit value) instead of 12 (16).
// Slot summaries: the scan kernel reads a unit as IGD_SLOTS slots of 64 consecutive records.  The
// component-wise maximum of a slot's words, (65535 - min s') | max e' << 16, passes the query
// test exactly when SOME word in the slot COULD pass it, so a query whose word fails against the
// summary skips the slot.  The low half is read from the slot's first record (the tile is sorted by
// start); the whole summary word of every slot is kept in the unit descriptor (Unit::W), so the scan
// kernel can prune a unit's queries before -- or without -- waiting for the unit's records.
__global__ __launch_bounds__(256) void k_pack_units(DbView db, Unit *__restrict__ unitsOut, uint32_t *__restrict__ pse,
                                                    uint16_t *__restrict__ px, uint32_t *__restrict__ pv,
                                                    int32_t *__restrict__ flag /* bit 0: a value needs > 16 bits; bit 1: malformed tile */)
{
    const int lane = threadIdx.x & 63;
    const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nw = (gridDim.x * blockDim.x) >> 6;
    int wide = 0;
    for (int ui = gw; ui < db.nUnits; ui += nw) {
        const Unit u = unitsOut[ui];
        const int T = (int)((unsigned)UNIT_J(u) * (unsigned)db.nbp);
        unsigned mx[6] = {0u, 0u, 0u, 0u, 0u, 0u};        // summary words of the unit's slots
        int npre = 0;
        for (int i0 = 0; i0 < u.n; i0 += IGD_WAVE) {
            const int i = i0 + lane;
            unsigned edv = 0u, spv = 65535u;
            if (i < u.n) {
            const int64_t r = u.off + i;
            const int st = db.start[r], en = db.end[r];
            const unsigned sp = st < T ? 0u : (unsigned)(st - T) + 1u;
            long long ed = (long long)en - T;
            if (ed > db.nbp) ed = db.nbp;
            if (ed < 1) ed = 1;
            edv = (unsigned)ed;
            spv = sp;
            pse[r] = (65535u - sp) | ((unsigned)ed << 16);
            px[r] = (uint16_t)db.idx[r];
            if (pv) {
                const int v = db.value[r];
                wide |= (v < -32768) | (v > 32767);
                pv[r] = (uint32_t)(uint16_t)db.idx[r] | ((uint32_t)(uint16_t)(int16_t)v << 16);
            }
            // a record that does not belong to its tile (malformed file): keep the exact path
            if (!(st < T + db.nbp && en > T)) wide |= 2;
            }
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned y = (unsigned)__shfl_xor((int)edv, o);
                edv = y > edv ? y : edv;
            }
            npre += __popcll(__ballot(spv == 0u));
            const unsigned s0 = (unsigned)__shfl((int)spv, 0);      // the tile is sorted by start: the slot's first record has its smallest s'
            if (i0 / IGD_WAVE < 6) mx[i0 / IGD_WAVE] = (65535u - s0) | (edv << 16);
        }
        if (lane == 0) {
            for (int r = 0; r < 6; r++) unitsOut[ui].W[r] = mx[r];
            unitsOut[ui].pre = npre;
        }
    }
    if (wide) atomicOr(flag, wide);
}
