// engine/seqpare.hpp -- part of igd_hip.hip (included there once; not a stand-alone header).
// Seqpare `-s`: kernels and host
// ------------------------------------------------------------------------------------------
// Seqpare (`search -q f.bed -s`), seqOverlaps src/igd_search.c:354-451.
// For every dataset m and every contig of the query file the reference repeatedly takes the best
// remaining (query, record) pair -- strict '>' while scanning queries in order, pairs of a query in
// discovery order -- and drops the pair's query ("row") and record ("column" = (index in tile, first
// tile of the QUERY): the reference stores idx_t = n1 for every tile, :291,:337).  That is greedy
// matching by the total order (similarity descending, position ascending), and the groups
// (query contig, dataset) are independent: entries are made by the `-f` enumeration kernel, brought
// together per group and put in that total order by two stable radix sorts (igd_sortscan.hpp: by
// similarity descending, then by group -- ties keep the reference's scan order), and each group is
// resolved by ONE WAVE, 64 candidates at a time, with hash sets of the taken rows and columns.
// Per dataset the accepted similarities are finally added up in double, contig by contig in the
// query file's order and in acceptance order -- the reference's order of additions.
#define SQ_CAP 1024                  // groups up to this size keep their hash sets in LDS
#define SQ_TAB (2 * SQ_CAP)
struct SeqArgs {
    const igd_hip_hit *ent;          // (q, idx_f, idx_g, sm bits), query-major, discovery order
    const uint32_t *vals;            // entry numbers: grouped, inside a group by (similarity desc, position asc)
    const int64_t *goff;             // [nGroups*nFiles + 1]
    const int32_t *q_qs;             // query starts (n1 = qs / nbp)
    float *sel;                      // accepted similarities, at goff[g] + i
    int32_t *nsel;                   // [nGroups*nFiles]
    int32_t *x_rows; unsigned long long *x_cols;   // HBM hash sets of groups > SQ_CAP: 4 slots per entry, preset to ~0
    unsigned int *next;
    int64_t nG;                      // number of groups
    int32_t nbp;
};

__device__ __forceinline__ uint32_t sq_hash32(uint32_t x) { x *= 0x9E3779B1u; return x ^ (x >> 15); }
__device__ __forceinline__ uint32_t sq_hash64(unsigned long long x)
{
    x *= 0x9E3779B97F4A7C15ull;
    return (uint32_t)(x >> 32) ^ (uint32_t)x;
}

// One wave per group.  Candidates arrive in the greedy order; 64 at a time: every lane looks its
// row and column up in the hash sets of what was accepted before this batch, then the batch is
// settled lane by lane (an accepted earlier lane knocks out later lanes that share its row or
// column), and the survivors are inserted and written out in lane order = acceptance order.
__global__ void __launch_bounds__(64) k_seq_greedy(SeqArgs a)
{
    __shared__ int32_t l_rows[SQ_TAB];
    __shared__ unsigned long long l_cols[SQ_TAB];
    __shared__ unsigned int gShared;
    const int lane = threadIdx.x;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (;;) {
        if (lane == 0) gShared = atomicAdd(a.next, 1u);
        __syncthreads();
        const int64_t g = gShared;
        __syncthreads();
        if (g >= a.nG) return;
        const int64_t off = a.goff[g];
        const int64_t n = a.goff[g + 1] - off;
        if (n == 0) { if (lane == 0) a.nsel[g] = 0; continue; }
        const bool small = n <= SQ_CAP;
        int32_t *rows = small ? l_rows : a.x_rows + 4 * off;
        unsigned long long *cols = small ? l_cols : a.x_cols + 4 * off;
        uint32_t mask = SQ_TAB - 1;
        if (small) {
            for (int i = lane; i < SQ_TAB; i += 64) { l_rows[i] = -1; l_cols[i] = ~0ull; }
        } else {
            uint32_t t = 1;
            while ((int64_t)t < 2 * n) t <<= 1;               // <= 4n slots, preset by the host
            mask = t - 1;
        }
        __syncthreads();
        int32_t nacc = 0;
        for (int64_t base = 0; base < n; base += 64) {
            const int64_t p = base + lane;
            bool valid = p < n;
            igd_hip_hit h;
            h.q = 0; h.idx = 0; h.start = 0; h.end = 0;
            if (valid) h = a.ent[a.vals[off + p]];
            const float x = __int_as_float(h.end);
            valid = valid && (x > 0.0f);
            if (__ballot(valid) == 0) break;                    // sorted: nothing positive is left
            const int32_t r = h.q;
            const unsigned long long k = ((unsigned long long)(uint32_t)(a.q_qs[h.q] / a.nbp) << 32) | (uint32_t)h.start;
            bool out = !valid;
            if (valid) {                                        // accepted in an earlier batch?
                for (uint32_t s = sq_hash32((uint32_t)r) & mask;; s = (s + 1) & mask) {
                    const int32_t v = __hip_atomic_load(&rows[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // not from a stale L1 line
                    if (v == r) { out = true; break; }
                    if (v == -1) break;
                }
                if (!out)
                    for (uint32_t s = sq_hash64(k) & mask;; s = (s + 1) & mask) {
                        const unsigned long long v = __hip_atomic_load(&cols[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (v == k) { out = true; break; }
                        if (v == ~0ull) break;
                    }
            }
            const int lim = (n - base) < 64 ? (int)(n - base) : 64;
            for (int j = 0; j < lim; j++) {                     // settle the batch in candidate order
                const int alive = __builtin_amdgcn_readlane((int)!out, j);
                if (!alive) continue;
                const int32_t rj = __builtin_amdgcn_readlane(r, j);
                const uint32_t klo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)k, j);
                const uint32_t khi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(k >> 32), j);
                const unsigned long long kj = ((unsigned long long)khi << 32) | klo;
                if (lane > j && (r == rj || k == kj)) out = true;
            }
            const unsigned long long accm = __ballot(!out);
            if (!out) {
                for (uint32_t s = sq_hash32((uint32_t)r) & mask;; s = (s + 1) & mask)
                    if (atomicCAS(&rows[s], -1, r) == -1) break;
                for (uint32_t s = sq_hash64(k) & mask;; s = (s + 1) & mask)
                    if (atomicCAS(&cols[s], ~0ull, k) == ~0ull) break;
                a.sel[off + nacc + __popcll(accm & lt)] = x;
            }
            nacc += __popcll(accm);
            __syncthreads();
        }
        if (lane == 0) a.nsel[g] = nacc;
        __syncthreads();
    }
}

// first sort key: similarity descending (positive floats order like their bit patterns)
__global__ void k_seq_key1(const igd_hip_hit *__restrict__ ent, int64_t n, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        keys[i] = ~(uint32_t)ent[i].end;
        vals[i] = (uint32_t)i;
    }
}

// second sort key: the group (query contig * nFiles + dataset) of the entry now at position i; group histogram
__global__ void k_seq_key2(const igd_hip_hit *__restrict__ ent, const uint32_t *__restrict__ vals, int64_t n,
                           const int32_t *__restrict__ qgrp, int32_t nFiles, uint32_t *__restrict__ keys, uint32_t *__restrict__ gcnt)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const igd_hip_hit h = ent[vals[i]];
        const uint32_t k = (uint32_t)qgrp[h.q] * (uint32_t)nFiles + (uint32_t)h.idx;
        keys[i] = k;
        atomicAdd(&gcnt[k], 1u);
    }
}

// sm[m] += maxf in the reference's order: contigs of the query file outermost, acceptance order inside
__global__ void k_seq_accumulate(const float *__restrict__ sel, const int32_t *__restrict__ nsel, const int64_t *__restrict__ goff,
                                 int32_t nGroups, int32_t nFiles, double *__restrict__ sums)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= nFiles) return;
    double acc = sums[m];                                     // 0, or the running sum of the earlier contigs (igd_hip_seqpare_add)
    for (int32_t c = 0; c < nGroups; c++) {
        const int64_t g = (int64_t)c * nFiles + m;
        const int64_t o = goff[g];
        const int32_t k = nsel[g];
        for (int32_t i = 0; i < k; i++) acc += (double)sel[o + i];
    }
    sums[m] = acc;
}

static int seqpare_core(igd_hip_db *db, const int32_t *ichr, const int32_t *qs, const int32_t *qe, int64_t nq,
                        const int32_t *qgroup, int32_t nGroups, double *sums, bool carry)
{
    if (!db || !sums || nq < 0 || nq > max_batch() || nGroups < 0 || (nq > 0 && (!ichr || !qs || !qe || !qgroup)) ||
        (int64_t)nGroups * db->nFiles >= 0x7fffffffLL) {
        snprintf(g_err, sizeof g_err, "igd_hip_seqpare: bad argument (batch limit %lld queries)", (long long)max_batch());
        return IGD_HIP_ERR_ARG;
    }
    if (db->gType != 1) {
        snprintf(g_err, sizeof g_err, "igd_hip_seqpare: needs a gType-1 database (seq_overlaps reads 16-byte records)");
        return IGD_HIP_ERR_ARG;
    }
    if (!carry) for (int32_t m = 0; m < db->nFiles; m++) sums[m] = 0.0;
    if (nq == 0 || db->nT == 0 || nGroups == 0 || db->nFiles == 0) return IGD_HIP_OK;
    HIPCHK(hipSetDevice(db->device));
    hipStream_t st = db->stream;
    int rc = ensure_qstage(db, nq);
    if (rc != IGD_HIP_OK) return rc;
    const int64_t nG = (int64_t)nGroups * db->nFiles;
    int32_t *d_qgrp = nullptr, *d_nsel = nullptr, *x_rows = nullptr;
    int64_t *d_qcount = nullptr, *d_qoff = nullptr, *d_bsum = nullptr, *d_goff = nullptr, *d_dig = nullptr,
            *d_sums = nullptr, *d_tot = nullptr;
    unsigned long long *x_cols = nullptr;
    uint32_t *kA = nullptr, *vA = nullptr, *kB = nullptr, *vB = nullptr, *d_gcnt = nullptr, *d_hist = nullptr;
    float *d_sel = nullptr;
    double *d_out = nullptr;
    unsigned int *d_next = nullptr;
    igd_hip_hit *d_ent = nullptr;
    uint32_t *h_gcnt = nullptr;
    auto cleanup = [&]() {
        void *ps[] = { d_qgrp, d_nsel, x_rows, d_qcount, d_qoff, d_bsum, d_goff, d_dig, d_sums,
                       d_tot, x_cols, kA, vA, kB, vB, d_gcnt, d_hist, d_sel, d_out, d_next, d_ent };
        for (void *p : ps) if (p) (void)hipFree(p);
        free(h_gcnt);
    };
#define EH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_err(#x, e_, __FILE__, __LINE__); cleanup(); return IGD_HIP_ERR_DEVICE; } } while (0)
#define DA(p, n) do { if ((rc = dalloc(&(p), (size_t)(n), nullptr)) != IGD_HIP_OK) { cleanup(); return rc; } } while (0)
    DA(d_qcount, nq); DA(d_qoff, nq + 1);
    DA(d_bsum, nq / IGD_SCAN_TILE + 2); DA(d_qgrp, nq);
    EH(hipMemcpyAsync(db->d_qc, ichr, (size_t)nq * 4, hipMemcpyHostToDevice, st));
    EH(hipMemcpyAsync(db->d_qs, qs, (size_t)nq * 4, hipMemcpyHostToDevice, st));
    EH(hipMemcpyAsync(db->d_qe, qe, (size_t)nq * 4, hipMemcpyHostToDevice, st));
    EH(hipMemcpyAsync(d_qgrp, qgroup, (size_t)nq * 4, hipMemcpyHostToDevice, st));
    igd_enum_queries<false, true><<<db->grid * 4, 256, 0, st>>>(db->v, db->d_qc, db->d_qs, db->d_qe, 0, (int)nq, d_qcount, nullptr, 0, nullptr);
    {
        const int sb = (int)((nq + IGD_SCAN_TILE - 1) / IGD_SCAN_TILE);
        k_scan64_sums<<<sb, IGD_SCAN_BLOCK, 0, st>>>(d_qcount, (int)nq, d_bsum);
        k_scan64_apply<<<sb, IGD_SCAN_BLOCK, 0, st>>>(d_qcount, (int)nq, d_bsum, d_qoff);
    }
    int64_t E = 0;
    EH(hipMemcpyAsync(&E, d_qoff + nq, 8, hipMemcpyDeviceToHost, st));
    EH(hipStreamSynchronize(st));
    if (E == 0) { cleanup(); return IGD_HIP_OK; }
    if (E >= 0xffffffffLL) { cleanup(); snprintf(g_err, sizeof g_err, "igd_hip_seqpare: %lld overlaps exceed one batch", (long long)E); return IGD_HIP_ERR_ARG; }
    DA(d_ent, E);
    igd_enum_queries<true, true><<<db->grid * 4, 256, 0, st>>>(db->v, db->d_qc, db->d_qs, db->d_qe, 0, (int)nq, nullptr, d_qoff, 0, d_ent);
    // order the entries: stable radix sort by similarity (descending), then by group (query contig * nFiles +
    // dataset) -> inside a group: similarity descending, ties in the reference's scan order
    DA(kA, E); DA(vA, E); DA(kB, E); DA(vB, E); DA(d_gcnt, nG + 1); DA(d_goff, nG + 1); DA(d_tot, 8);
    EH(hipMemsetAsync(d_gcnt, 0, (size_t)(nG + 1) * 4, st));
    {
        const int64_t nh = rs_blocks(E) * 256;
        const int64_t nsum = ((nh > nG + 1 ? nh : nG + 1) + SCAN_TILE - 1) / SCAN_TILE + 1;
        DA(d_hist, nh); DA(d_dig, nh); DA(d_sums, nsum);
        k_seq_key1<<<db->grid * 4, 256, 0, st>>>(d_ent, E, kA, vA);
        EH(radix_sort_pairs(&kA, &vA, &kB, &vB, E, 32, d_hist, d_dig, d_sums, d_tot, st));
        k_seq_key2<<<db->grid * 4, 256, 0, st>>>(d_ent, vA, E, d_qgrp, db->nFiles, kA, d_gcnt);
        EH((exclusive_scan<uint32_t, int64_t>(d_gcnt, nG + 1, d_goff, d_sums, d_tot, st)));
        int bits = 0;
        while (bits < 32 && ((uint64_t)(nG - 1) >> bits)) bits++;
        EH(radix_sort_pairs(&kA, &vA, &kB, &vB, E, bits, d_hist, d_dig, d_sums, d_tot, st));
    }
    // largest group decides whether the HBM hash sets of k_seq_greedy are needed
    h_gcnt = (uint32_t *)malloc((size_t)nG * 4);
    if (!h_gcnt) { cleanup(); return IGD_HIP_ERR_NOMEM; }
    EH(hipMemcpyAsync(h_gcnt, d_gcnt, (size_t)nG * 4, hipMemcpyDeviceToHost, st));
    EH(hipStreamSynchronize(st));
    uint32_t maxG = 0;
    for (int64_t g = 0; g < nG; g++) if (h_gcnt[g] > maxG) maxG = h_gcnt[g];
    if (maxG > SQ_CAP) {
        DA(x_rows, 4 * E); DA(x_cols, 4 * E);
        EH(hipMemsetAsync(x_rows, 0xff, (size_t)E * 16, st));
        EH(hipMemsetAsync(x_cols, 0xff, (size_t)E * 32, st));
    }
    DA(d_sel, E); DA(d_nsel, nG); DA(d_next, 16); DA(d_out, db->nFiles);
    EH(hipMemsetAsync(d_next, 0, 4, st));
    if (carry) EH(hipMemcpyAsync(d_out, sums, (size_t)db->nFiles * 8, hipMemcpyHostToDevice, st));   // the running sums continue
    else EH(hipMemsetAsync(d_out, 0, (size_t)db->nFiles * 8, st));
    {
        SeqArgs a;
        a.ent = d_ent; a.vals = vA; a.goff = d_goff; a.q_qs = db->d_qs; a.sel = d_sel; a.nsel = d_nsel;
        a.x_rows = x_rows; a.x_cols = x_cols;
        a.next = d_next; a.nG = nG; a.nbp = db->nbp;
        const int64_t want = nG < (int64_t)db->grid * 3 ? nG : (int64_t)db->grid * 3;
        k_seq_greedy<<<(unsigned)want, 64, 0, st>>>(a);
        k_seq_accumulate<<<(db->nFiles + 63) / 64, 64, 0, st>>>(d_sel, d_nsel, d_goff, nGroups, db->nFiles, d_out);
    }
    EH(hipGetLastError());
    EH(hipMemcpyAsync(sums, d_out, (size_t)db->nFiles * 8, hipMemcpyDeviceToHost, st));
    EH(hipStreamSynchronize(st));
#undef EH
#undef DA
    cleanup();
    return IGD_HIP_OK;
}

extern "C" int igd_hip_seqpare(igd_hip_db *db, const int32_t *ichr, const int32_t *qs, const int32_t *qe, int64_t nq,
                               const int32_t *qgroup, int32_t nGroups, double *sums)
{
    return seqpare_core(db, ichr, qs, qe, nq, qgroup, nGroups, sums, false);
}
extern "C" int igd_hip_seqpare_add(igd_hip_db *db, const int32_t *ichr, const int32_t *qs, const int32_t *qe, int64_t nq,
                                   const int32_t *qgroup, int32_t nGroups, double *sums)
{
    return seqpare_core(db, ichr, qs, qe, nq, qgroup, nGroups, sums, true);
}
