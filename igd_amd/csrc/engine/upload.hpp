// engine/upload.hpp -- part of igd_hip.hip (included there once; not a stand-alone header).
// AoS -> SoA transposes of the .igd records at open
// ------------------------------------------------------------------------------------------
// upload: AoS (file order) -> SoA.  One record per lane: a 16-byte gdata_t is one dwordx4
// load (src/igd_base.h:41-46: idx,start,end,value); 12-byte gdata0_t three dword loads.
__global__ void k_aos_to_soa16(const int4 *__restrict__ aos, int64_t n, int32_t *__restrict__ start,
                               int32_t *__restrict__ end, int32_t *__restrict__ idx,
                               int32_t *__restrict__ value)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        int4 r = aos[i];
        idx[i] = r.x; start[i] = r.y; end[i] = r.z; value[i] = r.w;
    }
}
__global__ void k_aos_to_soa12(const int32_t *__restrict__ aos, int64_t n, int32_t *__restrict__ start,
                               int32_t *__restrict__ end, int32_t *__restrict__ idx)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        idx[i] = aos[3 * i]; start[i] = aos[3 * i + 1]; end[i] = aos[3 * i + 2];
    }
}

// hits[] is indexed by idx without a bounds check in the reference (src/igd_search.c:491); on
// the GPU an out-of-range idx would corrupt LDS, so the image is validated once at open.
__global__ void k_idx_range(const int32_t *__restrict__ idx, int64_t n, int32_t nFiles, int32_t *__restrict__ bad)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int b = 0;
    for (; i < n; i += stride) b |= (idx[i] < 0) | (idx[i] >= nFiles);
    if (b) atomicOr(bad, 1);
}
