// lsq_norms.hip -- quantize_norms on the device: the link between the encode and the search.
//
// Reference: quantize_norms(B, C, cbnorms) (src/utils.jl:6-31) with reconstruct (src/utils.jl:34-45 style: CB = C[1][:, B[1,:]]; CB += C[i][:, B[i,:]],
// i ascending).  Per database vector: the reconstruction's squared norm (f32, dimensions ascending, the square rounded before the add) and the
// index of the nearest of the h scalar centroids, (norm - cb[j])^2 in f32, first minimum (findmin).  The search consumes cbnorms[index]
// (demos/demo_lsq_gpu.jl:57-60), which this kernel can emit in the same pass.  PARITY UNPINNED against the reference: its norm loop is `@simd`
// (reassociated by the compiler); this is the sequential order, the same as the Python mirror (reference_api.quantize_norms), bit for bit.
#include "lsq_internal.h"

#pragma clang fp contract(off)

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// one thread per vector: m codeword rows streamed along the dimension (L2-resident codebooks), the running reconstruction in registers 4 dims at a time
__global__ __launch_bounds__(256) void quantize_norms_kernel(const uint8_t *__restrict__ codes, int stride, const float *__restrict__ K,
                                                             const float *__restrict__ cb, int ncb, int64_t n, int d, int m,
                                                             uint8_t *__restrict__ idx0, int16_t *__restrict__ idx1, float *__restrict__ dbnorms,
                                                             float *__restrict__ norms) {
    __shared__ float cbs[LSQ_H];
    for (int t = threadIdx.x; t < ncb; t += 256) cbs[t] = cb[t];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *row[LSQ_MAX_M];
    for (int j = 0; j < m; ++j) row[j] = K + ((int64_t)j * LSQ_H + codes[i * stride + j]) * d;
    float nrm = 0.0f;
    const bool vec = (d & 3) == 0 && (((uintptr_t)K & 15) == 0);
    if (vec) {
        for (int t = 0; t < d; t += 4) {
            f32x4 acc = *reinterpret_cast<const f32x4 *>(row[0] + t);
            for (int j = 1; j < m; ++j) acc = acc + *reinterpret_cast<const f32x4 *>(row[j] + t);      // codebooks ascending
            nrm = nrm + acc.x * acc.x;                                                               // dimensions ascending; square, then add
            nrm = nrm + acc.y * acc.y;
            nrm = nrm + acc.z * acc.z;
            nrm = nrm + acc.w * acc.w;
        }
    } else {
        for (int t = 0; t < d; ++t) {
            float acc = row[0][t];
            for (int j = 1; j < m; ++j) acc = acc + row[j][t];
            nrm = nrm + acc * acc;
        }
    }
    int best = 0;
    float bd = (nrm - cbs[0]) * (nrm - cbs[0]);
    for (int j = 1; j < ncb; ++j) {
        const float df = nrm - cbs[j], dd = df * df;
        if (dd < bd) { bd = dd; best = j; }                                                          // strict <: the first minimum (findmin)
    }
    if (idx0) idx0[i] = (uint8_t)best;
    if (idx1) idx1[i] = (int16_t)(best + 1);
    if (dbnorms) dbnorms[i] = cbs[best];
    if (norms) norms[i] = nrm;
}

}  // namespace

int lsq_launch_quantize_norms(hipStream_t s, const uint8_t *codes, int stride, const float *K, const float *cb, int ncb, int64_t n, int d, int m,
                              uint8_t *idx0, int16_t *idx1, float *dbnorms, float *norms) {
    if (n <= 0) return LSQ_OK;
    hipLaunchKernelGGL(quantize_norms_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, codes, stride, K, cb, ncb, n, d, m, idx0, idx1,
                       dbnorms, norms);
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}
