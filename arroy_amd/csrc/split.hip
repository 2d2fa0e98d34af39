// split.hip — single-node build-side entry points: the margin loop over one id list
// (src/writer.rs:1201-1207,1424-1431,1494-1501) and one `D::create_split` (src/writer.rs:1200).
// The whole-forest, level-synchronous versions of the same device code are in forest.hip.
#include "common.h"
#include "split_device.h"

namespace ah {

static constexpr int kBlock = 256;
static constexpr int kMaxBlocks = 2048;

// One octet per item (f32 metrics) — same streaming pattern as the distance scan.  Side bits are packed
// LSB-first into 32-bit words with atomicOr (integer, order-free); the left count is an integer reduction.
template <int METRIC>
__global__ __launch_bounds__(kBlock) void k_split_sides_f32(DataView dv, const float *__restrict__ nvec,
                                                            const float *__restrict__ nhdr,
                                                            const uint32_t *__restrict__ ids, uint64_t n,
                                                            uint32_t *__restrict__ bits, unsigned long long *n_left,
                                                            float *__restrict__ margins, uint32_t *err,
                                                            int row_is_normal) {
    extern __shared__ float4 s_n4[];
    for (uint32_t i = threadIdx.x; i < (dv.pitch >> 2); i += blockDim.x)
        s_n4[i] = reinterpret_cast<const float4 *>(nvec)[i];
    __syncthreads();
    const float *s_n = reinterpret_cast<const float *>(s_n4);
    const LeafHdr nh = {nhdr[0], nhdr[1]};
    const uint32_t j = threadIdx.x & 7u;
    const uint64_t n_octets = ((uint64_t)gridDim.x * blockDim.x) >> 3;
    uint32_t my_left = 0;
    for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; i < n; i += n_octets) {
        uint64_t row = i;
        if (ids) {
            row = row_of_id(dv, ids[i]);
            if (row == ~0ull) {
                if (j == 0) atomicOr(err, 1u);
                continue;
            }
        }
        // row_is_normal: the dataset rows are split-plane normals and the broadcast vector is the query leaf
        // (`D::margin(&normal, query_leaf)`, src/reader.rs:366-369): the bias then comes from the ROW's header.
        LeafHdr h = nh;
        if (row_is_normal && (METRIC == AH_EUCLIDEAN || METRIC == AH_MANHATTAN)) h.h0 = dv.headers[row];
        const float m = margin_f32<METRIC>(dv, s_n, h, row, j);
        if (j == 0) {
            const uint32_t side = side_of_margin(m);
            if (bits) {
                if (side) atomicOr(&bits[i >> 5], 1u << (i & 31));
                else my_left++;
            }
            if (margins) margins[i] = m;
        }
    }
    // wave reduction of the per-lane left counts, one atomic per wave
    for (int off = 32; off > 0; off >>= 1) my_left += __shfl_down(my_left, off);
    if ((threadIdx.x & 63u) == 0 && my_left && n_left) atomicAdd(n_left, (unsigned long long)my_left);
}

__global__ __launch_bounds__(kBlock) void k_split_sides_bq(DataView dv, const uint64_t *__restrict__ nvec,
                                                           const float *__restrict__ nhdr,
                                                           const uint32_t *__restrict__ ids, uint64_t n,
                                                           uint32_t *__restrict__ bits, unsigned long long *n_left,
                                                           float *__restrict__ margins, uint32_t *err,
                                                           int row_is_normal) {
    extern __shared__ uint64_t s_nw[];
    for (uint32_t i = threadIdx.x; i < dv.pitch; i += blockDim.x) s_nw[i] = nvec[i];
    __syncthreads();
    const LeafHdr nh = {nhdr[0], nhdr[1]};
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t my_left = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t row = i;
        if (ids) {
            row = row_of_id(dv, ids[i]);
            if (row == ~0ull) {
                atomicOr(err, 1u);
                continue;
            }
        }
        LeafHdr h = nh;
        if (row_is_normal) h.h0 = dv.headers[row];
        const float m = margin_bq(dv, s_nw, h, row);
        const uint32_t side = side_of_margin(m);
        if (bits) {
            if (side) atomicOr(&bits[i >> 5], 1u << (i & 31));
            else my_left++;
        }
        if (margins) margins[i] = m;
    }
    for (int off = 32; off > 0; off >>= 1) my_left += __shfl_down(my_left, off);
    if ((threadIdx.x & 63u) == 0 && my_left && n_left) atomicAdd(n_left, (unsigned long long)my_left);
}

int launch_split_sides(const DataView &dv, const void *d_nvec, const float *d_nhdr, const uint32_t *d_ids, uint64_t n,
                       uint8_t *d_side_bits, unsigned long long *d_n_left, float *d_margins, uint32_t *d_err,
                       hipStream_t s, int row_is_normal) {
    if (n == 0) return AH_OK;
    uint32_t *bits = reinterpret_cast<uint32_t *>(d_side_bits);
    if (metric_is_bq(dv.metric)) {
        uint64_t b = (n + kBlock - 1) / kBlock;
        if (b > kMaxBlocks) b = kMaxBlocks;
        hipLaunchKernelGGL(k_split_sides_bq, dim3((unsigned)b), dim3(kBlock), dv.pitch * 8, s, dv,
                           (const uint64_t *)d_nvec, d_nhdr, d_ids, n, bits, d_n_left, d_margins, d_err, row_is_normal);
    } else {
        uint64_t b = (n + (kBlock / 8) - 1) / (kBlock / 8);
        if (b > kMaxBlocks) b = kMaxBlocks;
        const size_t sh = (size_t)dv.pitch * 4;
#define AH_LAUNCH(M)                                                                                               \
    hipLaunchKernelGGL((k_split_sides_f32<M>), dim3((unsigned)b), dim3(kBlock), sh, s, dv, (const float *)d_nvec, \
                       d_nhdr, d_ids, n, bits, d_n_left, d_margins, d_err, row_is_normal)
        switch (dv.metric) {
        case AH_EUCLIDEAN: AH_LAUNCH(AH_EUCLIDEAN); break;
        case AH_MANHATTAN: AH_LAUNCH(AH_MANHATTAN); break;
        case AH_COSINE: AH_LAUNCH(AH_COSINE); break;
        default: AH_LAUNCH(AH_DOT_PRODUCT); break;
        }
#undef AH_LAUNCH
    }
    AH_HIP(hipGetLastError());
    return AH_OK;
}

// One wave per create_split; three centroid-sized LDS buffers.
__global__ __launch_bounds__(64) void k_create_split(DataView dv, const uint32_t *__restrict__ sample_rows,
                                                     void *out_vec, float *out_hdr) {
    extern __shared__ float4 s_buf4[];
    float *s_buf = reinterpret_cast<float *>(s_buf4);
    const uint32_t fpitch = f32_space_pitch(dv.metric, dv.dims);
    __shared__ uint32_t s_rows[AH_SPLIT_SAMPLES];
    if (threadIdx.x < AH_SPLIT_SAMPLES) s_rows[threadIdx.x] = sample_rows[threadIdx.x];
    __syncthreads();
    wave_create_split_any(dv, s_rows, s_buf, s_buf + fpitch, s_buf + 2 * fpitch, out_vec, out_hdr, threadIdx.x);
}

int launch_create_split(const DataView &dv, const uint32_t *d_sample_rows, void *d_out_vec, float *d_out_hdr,
                        hipStream_t s) {
    const size_t sh = (size_t)f32_space_pitch(dv.metric, dv.dims) * 4 * 3;
    AH_REQUIRE(sh <= 150 * 1024, AH_ERR_INVALID_DIMENSION, "dimensions %u too large for the LDS-resident two-means",
               dv.dims);
    if (sh > 48 * 1024)
        AH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_create_split),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    hipLaunchKernelGGL(k_create_split, dim3(1), dim3(64), sh, s, dv, d_sample_rows, d_out_vec, d_out_hdr);
    AH_HIP(hipGetLastError());
    return AH_OK;
}

}  // namespace ah
