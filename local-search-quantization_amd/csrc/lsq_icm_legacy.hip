// lsq_icm_legacy.hip -- the three earlier ICM schedules (0: per-node L2 gathers, 1: fused sweeps with register-resident
// unaries, 2: LDS slices + combine).  TUNING BUILD ONLY (liblsq_mi355x_tuning.so, -DLSQ_TUNING): they are the measured
// alternatives DESIGN.md section 4.2 cites and independent implementations the parity tests cross-check; the shipped
// library carries the LDS-walk kernel (lsq_icm.hip) only.  Same arithmetic as the walk kernel: conditioning adds in
// ascending k (plain f32 adds), lowest-index argmin (encode_icm.jl:76-119).
#include <mutex>

#include "lsq_wave.h"

namespace {

// ---- ICM node update, one launch per node (schedule 0) -----------------------------------------
// Streams U_j (1 KiB/vector, non-temporal) from HBM, gathers (M-1) 1 KiB columns of block-row j
// from L2, writes one code byte.  encode_icm.jl:76-119 for all vectors of the chunk.
template <int M>
__global__ __launch_bounds__(256) void icm_node_kernel(const float *__restrict__ Uj, const float *__restrict__ Tj,
                                                       uint8_t *__restrict__ rec, int64_t n, int j) {
    constexpr int CS = (M <= 8) ? 8 : 16;
    const int lane = threadIdx.x & 63;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    int64_t i = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    for (; i < n; i += nwaves) {
        const CodeRec cr = load_rec<CS>(rec, i);
        f32x4 s = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(Uj + i * LSQ_H) + lane);
        f32x4 c[M > 1 ? M - 1 : 1];
#pragma unroll
        for (int kk = 0; kk < M - 1; ++kk) {
            const int k = kk + (kk >= j ? 1 : 0);
            const float *col = Tj + ((int64_t)(k * LSQ_H) + cr.get(k)) * LSQ_H;
            c[kk] = reinterpret_cast<const f32x4 *>(col)[lane];
        }
#pragma unroll
        for (int kk = 0; kk < M - 1; ++kk) s = s + c[kk];      // ascending k, plain f32 adds
        const int best = wave_first_argmin(s, lane);
        if (lane == 0) rec[i * CS + j] = (uint8_t)best;
    }
}

// ---- fused sweeps (schedule 1): unaries register-resident, all nsweeps*M node updates ---------
struct NodeOrder { int v[LSQ_MAX_M]; };

template <int M, int J>
__device__ inline void fused_node(const f32x4 (&u)[M], const float *__restrict__ T, CodeRec &cr, int lane) {
    f32x4 s = u[J];
    f32x4 c[M > 1 ? M - 1 : 1];
    const float *Tj = T + (int64_t)J * M * LSQ_H * LSQ_H;
#pragma unroll
    for (int kk = 0; kk < M - 1; ++kk) {
        constexpr int dummy = 0; (void)dummy;
        const int k = kk + (kk >= J ? 1 : 0);
        const float *col = Tj + ((int64_t)(k * LSQ_H) + cr.get(k)) * LSQ_H;
        c[kk] = reinterpret_cast<const f32x4 *>(col)[lane];
    }
#pragma unroll
    for (int kk = 0; kk < M - 1; ++kk) s = s + c[kk];
    cr.set(J, (uint32_t)wave_first_argmin(s, lane));
}

template <int M, int J>
struct FusedDispatch {
    __device__ static inline void run(int j, const f32x4 (&u)[M], const float *T, CodeRec &cr, int lane) {
        if (j == J) fused_node<M, J>(u, T, cr, lane);
        else FusedDispatch<M, J + 1>::run(j, u, T, cr, lane);
    }
};
template <int M>
struct FusedDispatch<M, M> {
    __device__ static inline void run(int, const f32x4 (&)[M], const float *, CodeRec &, int) {}
};

template <int M>
__global__ __launch_bounds__(256) void icm_fused_kernel(const float *__restrict__ U, const float *__restrict__ T,
                                                        uint8_t *__restrict__ rec, int64_t n, NodeOrder order, int nsweeps) {
    constexpr int CS = (M <= 8) ? 8 : 16;
    const int lane = threadIdx.x & 63;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    int64_t i = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    for (; i < n; i += nwaves) {
        CodeRec cr = load_rec<CS>(rec, i);
        f32x4 u[M];
#pragma unroll
        for (int j = 0; j < M; ++j)
            u[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(U + ((int64_t)j * n + i) * LSQ_H) + lane);
        for (int sw = 0; sw < nsweeps; ++sw)
#pragma unroll 1
            for (int q = 0; q < M; ++q) FusedDispatch<M, 0>::run(order.v[q], u, T, cr, lane);
        if (lane == 0) {
            uint64_t *p = reinterpret_cast<uint64_t *>(rec + i * CS);
            p[0] = cr.lo;
            if (CS == 16) p[1] = cr.hi;
        }
    }
}

// ---- LDS-slice schedule (schedule 2) ---------------------------------------------------------------
// Why: on gfx950 the HBM-miss stream of U_j and the L2-hit table gathers of icm_node_kernel do not
// overlap -- their times ADD (measured: 164 us + 240 us -> 470 us per 10^6-vector launch; tools/
// ubench_icm.hip).  So the vector-memory path is given to the U stream alone and the
// table columns come from LDS:
//   * a 1024-thread block owns one SLICE of SL candidates (16 for m <= 10, 8 above) of node j and
//     stages T_j[k][b][a0..a0+SL) for all k != j, b into LDS: (m-1)*256*SL*4 B (112 KiB at m = 8);
//   * U_j is stored slice-major, Us[slice][i][SL], so one wave load = 1 KiB contiguous = 64/(SL/4)
//     vectors x SL candidates (lane = (SL/4)*v + q: candidates 4q..4q+3 of vector v);
//   * per vector the block emits the partial (min, index-in-slice) of its SL candidates; a second
//     tiny kernel (icm_combine_kernel) takes the lowest-index global minimum over the 256/SL slices.
// Conditioning order, plain f32 adds and first-index argmin are exactly those of icm_node_kernel.
template <int M, int SL>
__global__ __launch_bounds__(1024) void icm_slice_kernel(const float *__restrict__ Usj, const float *__restrict__ Tj,
                                                         const uint8_t *__restrict__ rec, float2 *__restrict__ part,
                                                         int64_t n, int j, int nranges) {
    constexpr int CS = (M <= 8) ? 8 : 16;
    constexpr int NS = LSQ_H / SL;          // slices
    constexpr int LPV = SL / 4;             // lanes per vector
    constexpr int VPW = 64 / LPV;           // vectors per wave iteration
    constexpr int CW = (M - 1 + 3) / 4;     // compacted code words (conditioning codes in ascending k, j skipped)
    constexpr int RW = CS / 4;              // record words
    extern __shared__ f32x4 lds_tab[];      // [(M-1)*256][LPV]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int slice = blockIdx.x % NS, range = blockIdx.x / NS;

    for (int e = threadIdx.x; e < (M - 1) * LSQ_H * LPV; e += 1024) {
        const int q = e % LPV, eb = e / LPV, kk = eb >> 8, b = eb & 255;
        const int k = kk + (kk >= j ? 1 : 0);
        lds_tab[e] = *reinterpret_cast<const f32x4 *>(Tj + ((int64_t)(k * LSQ_H) + b) * LSQ_H + slice * SL + q * 4);
    }
    // v_perm_b32 selectors: compact word w takes bytes k(4w..4w+3) - 4w (0..4) of record words (w, w+1)
    uint32_t sel[CW > 0 ? CW : 1];
#pragma unroll
    for (int w = 0; w < CW; ++w) {
        uint32_t sv = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int kk = 4 * w + t;
            const int k = kk + (kk >= j ? 1 : 0);
            sv |= (uint32_t)((kk < M - 1 ? k - 4 * w : 0) & 7) << (8 * t);
        }
        sel[w] = sv;
    }
    __syncthreads();

    const int64_t per = (n + nranges - 1) / nranges;
    const int64_t lo = range * per, hi = (lo + per < n) ? lo + per : n;
    const int v = lane / LPV, q = lane % LPV;
    const float *Ub = Usj + (int64_t)slice * n * SL;
    const int64_t step = 16 * VPW;

    struct Item { f32x4 u; uint32_t r[RW]; };
    auto load_item = [&](int64_t i0, Item &it) {
        const int64_t i = i0 + v;
        if (i < hi) {
            it.u = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(Ub + i * SL) + q);
            const uint32_t *rp = reinterpret_cast<const uint32_t *>(rec + i * CS);
#pragma unroll
            for (int w = 0; w < RW; ++w) it.r[w] = rp[w];
        } else {
            it.u = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < RW; ++w) it.r[w] = 0u;
        }
    };

    int64_t i0 = lo + (int64_t)wave * VPW;
    Item a, b;
    load_item(i0, a);
    load_item(i0 + step, b);
    for (; i0 < hi; i0 += step) {
        const Item cur = a;
        a = b;
        load_item(i0 + 2 * step, b);          // two iterations of U in flight per wave

        f32x4 s = cur.u;
#pragma unroll
        for (int w = 0; w < CW; ++w) {
            const uint32_t hiw = (w + 1 < RW) ? cur.r[w + 1] : 0u;
            const uint32_t cw = __builtin_amdgcn_perm(hiw, cur.r[w], sel[w]);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int kk = 4 * w + t;
                if (kk < M - 1) {
                    const uint32_t code = (cw >> (8 * t)) & 0xffu;
                    s = s + lds_tab[(kk * LSQ_H + code) * LPV + q];      // ascending k, plain f32 add
                }
            }
        }
        // partial first-argmin over this vector's SL candidates: in-lane 4, then across its LPV lanes
        float lm = fminf(fminf(s.x, s.y), fminf(s.z, s.w));
        int li = ((s.x == lm) ? 0 : (s.y == lm) ? 1 : (s.z == lm) ? 2 : 3) + 4 * q;
        if (lm != lm) { lm = __builtin_inff(); li = 1000; }              // all-NaN lane: never wins
        if (slice == 0 && q == 0 && s.x != s.x) { lm = -__builtin_inff(); li = 0; }   // s[0] NaN: strict-< scan keeps index 0
        {
            float ov = dpp_self<DPP_XOR1, 0xf>(lm);
            int oi = __builtin_amdgcn_update_dpp(li, li, DPP_XOR1, 0xf, 0xf, false);
            if (ov < lm || (ov == lm && oi < li)) { lm = ov; li = oi; }
            if (LPV == 4) {
                ov = dpp_self<DPP_XOR2, 0xf>(lm);
                oi = __builtin_amdgcn_update_dpp(li, li, DPP_XOR2, 0xf, 0xf, false);
                if (ov < lm || (ov == lm && oi < li)) { lm = ov; li = oi; }
            }
        }
        if (q == 0 && i0 + v < hi) part[(int64_t)slice * n + i0 + v] = make_float2(lm, __int_as_float(li));
    }
}

template <int SL>
__global__ __launch_bounds__(256) void icm_combine_kernel(const float2 *__restrict__ part, uint8_t *__restrict__ rec, int64_t n, int cs, int j) {
    constexpr int NS = LSQ_H / SL;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float2 p = part[i];
    float best = p.x;
    int bi = __float_as_int(p.y);
#pragma unroll
    for (int sl = 1; sl < NS; ++sl) {
        p = part[(int64_t)sl * n + i];
        if (p.x < best) { best = p.x; bi = SL * sl + __float_as_int(p.y); }      // strict <: lowest slice wins ties
    }
    rec[i * cs + j] = (uint8_t)(bi > 255 ? 0 : bi);
}

inline unsigned wave_grid(int64_t n) {      // persistent grid: 4 waves per block, <= 8 blocks per CU on 256 CUs
    int64_t blocks = (n + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

}  // namespace

#define LSQ_DISPATCH_M(m, EXPR)                                                              \
    switch (m) {                                                                             \
        case 1: { constexpr int M_ = 1; EXPR; } break;   case 2: { constexpr int M_ = 2; EXPR; } break;   \
        case 3: { constexpr int M_ = 3; EXPR; } break;   case 4: { constexpr int M_ = 4; EXPR; } break;   \
        case 5: { constexpr int M_ = 5; EXPR; } break;   case 6: { constexpr int M_ = 6; EXPR; } break;   \
        case 7: { constexpr int M_ = 7; EXPR; } break;   case 8: { constexpr int M_ = 8; EXPR; } break;   \
        case 9: { constexpr int M_ = 9; EXPR; } break;   case 10: { constexpr int M_ = 10; EXPR; } break; \
        case 11: { constexpr int M_ = 11; EXPR; } break; case 12: { constexpr int M_ = 12; EXPR; } break; \
        case 13: { constexpr int M_ = 13; EXPR; } break; case 14: { constexpr int M_ = 14; EXPR; } break; \
        case 15: { constexpr int M_ = 15; EXPR; } break; case 16: { constexpr int M_ = 16; EXPR; } break; \
        default: lsq_set_error("m = %d out of range 1..16", m); return LSQ_EINVAL;          \
    }

int lsq_launch_icm_node(hipStream_t s, const float *Uj, const float *T, uint8_t *rec, int64_t n, int m, int j) {
    if (n <= 0) return LSQ_OK;
    const float *Tj = T + (int64_t)j * m * LSQ_H * LSQ_H;
    LSQ_DISPATCH_M(m, hipLaunchKernelGGL(icm_node_kernel<M_>, dim3(wave_grid(n)), dim3(256), 0, s, Uj, Tj, rec, n, j));
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

int lsq_launch_icm_fused(hipStream_t s, const float *U, const float *T, uint8_t *rec, int64_t n, int m,
                         const int32_t *order_host, int nsweeps) {
    if (n <= 0) return LSQ_OK;
    NodeOrder o;
    for (int q = 0; q < LSQ_MAX_M; ++q) o.v[q] = q < m ? order_host[q] : 0;
    LSQ_DISPATCH_M(m, hipLaunchKernelGGL(icm_fused_kernel<M_>, dim3(wave_grid(n)), dim3(256), 0, s, U, T, rec, n, o, nsweeps));
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

template <int M, int SL>
static int launch_slice_t(hipStream_t s, const float *Usj, const float *Tj, uint8_t *rec, float2 *part, int64_t n, int j) {
    constexpr int NS = LSQ_H / SL;
    constexpr int LDS_BYTES = (M - 1) * LSQ_H * SL * 4;
    static std::mutex mu;                  // > 64 KiB of dynamic LDS needs the opt-in; cheap enough to repeat per launch here
    {
        std::lock_guard<std::mutex> lock(mu);
        LSQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&icm_slice_kernel<M, SL>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    }
    const int64_t per_iter = 16 * (64 / (SL / 4));            // vectors per block iteration
    int64_t nranges = (n + 4 * per_iter - 1) / (4 * per_iter);
    const int64_t max_ranges = 512 / NS;                       // ~2 blocks per CU over the launch (1 resident: LDS)
    if (nranges > max_ranges) nranges = max_ranges;
    if (nranges < 1) nranges = 1;
    hipLaunchKernelGGL((icm_slice_kernel<M, SL>), dim3((unsigned)(NS * nranges)), dim3(1024), LDS_BYTES, s, Usj, Tj, rec, part, n, j, (int)nranges);
    LSQ_HIP(hipGetLastError());
    const int cs = (M <= 8) ? 8 : 16;
    hipLaunchKernelGGL((icm_combine_kernel<SL>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, rec, n, cs, j);
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

int lsq_launch_icm_slice(hipStream_t s, const float *Usj, const float *T, uint8_t *rec, float2 *part, int64_t n, int m, int j) {
    if (n <= 0) return LSQ_OK;
    const float *Tj = T + (int64_t)j * m * LSQ_H * LSQ_H;
    switch (m) {
        case 1: return launch_slice_t<1, 16>(s, Usj, Tj, rec, part, n, j);
        case 2: return launch_slice_t<2, 16>(s, Usj, Tj, rec, part, n, j);
        case 3: return launch_slice_t<3, 16>(s, Usj, Tj, rec, part, n, j);
        case 4: return launch_slice_t<4, 16>(s, Usj, Tj, rec, part, n, j);
        case 5: return launch_slice_t<5, 16>(s, Usj, Tj, rec, part, n, j);
        case 6: return launch_slice_t<6, 16>(s, Usj, Tj, rec, part, n, j);
        case 7: return launch_slice_t<7, 16>(s, Usj, Tj, rec, part, n, j);
        case 8: return launch_slice_t<8, 16>(s, Usj, Tj, rec, part, n, j);
        case 9: return launch_slice_t<9, 16>(s, Usj, Tj, rec, part, n, j);
        case 10: return launch_slice_t<10, 16>(s, Usj, Tj, rec, part, n, j);
        case 11: return launch_slice_t<11, 8>(s, Usj, Tj, rec, part, n, j);
        case 12: return launch_slice_t<12, 8>(s, Usj, Tj, rec, part, n, j);
        case 13: return launch_slice_t<13, 8>(s, Usj, Tj, rec, part, n, j);
        case 14: return launch_slice_t<14, 8>(s, Usj, Tj, rec, part, n, j);
        case 15: return launch_slice_t<15, 8>(s, Usj, Tj, rec, part, n, j);
        case 16: return launch_slice_t<16, 8>(s, Usj, Tj, rec, part, n, j);
        default: lsq_set_error("m = %d out of range 1..16", m); return LSQ_EINVAL;
    }
}
