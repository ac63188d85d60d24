// lsq_linscan.hip -- ADC linear scan for codes + separately stored database norms (HOST code).
//
// SURVEY 8(f)-1 / north_star: "linscan asymmetric-distance table sums stay on the host but call into the
// same C-ABI".  Replaces linscan_aqd_query_extra_byte of the reference
// (src/linscan/cpp/linscan_aqd_pairwise_byte.cpp:14-104, bound at src/linscan/Linscan.jl:63-69), written from
// scratch: same arithmetic order (so distances are bit-identical to the reference build in oracle/_ref and the
// ids agree including ties), different machinery -- a bounded max-heap per query instead of materialising
// 10^7 (dist, id) pairs and partial_sort-ing them, std::thread workers instead of OpenMP (no second OpenMP
// runtime next to torch's).
//
//   table[j]   = ((0 - (2 q0) c_j0) - (2 q1) c_j1) - ...           (f32, k ascending; no FMA: -ffp-contract=off)
//   dist(i)    = (((0 + table[0*h + b_i0]) + table[1*h + b_i1]) + ...) + dbnorms[i]
//   result     = the nn smallest (dist, id) pairs in lexicographic order, ids 1-BASED like the reference (:75).
#include <algorithm>
#include <thread>
#include <utility>
#include <vector>

#include "lsq_internal.h"

namespace {

typedef std::pair<float, int> DistId;      // lexicographic < : distance, then id -- std::pair's own order

void scan_queries(float *dists, int *idx, const unsigned char *codes, const float *queries, const float *codebooks,
                  const float *dbnorms, int q0, int q1, int ncodes, int m, int h, int d, int nn) {
    const int total = m * h;
    std::vector<float> table((size_t)total);
    std::vector<DistId> heap;
    heap.reserve((size_t)nn + 1);
    for (int q = q0; q < q1; ++q) {
        const float *query = queries + (size_t)q * d;
        for (int j = 0; j < total; ++j) {
            const float *c = codebooks + (size_t)j * d;
            float t = 0.0f;
            for (int k = 0; k < d; ++k) t -= 2 * query[k] * c[k];      // (2*q)*c, then one rounded subtract
            table[(size_t)j] = t;
        }
        heap.clear();
        const unsigned char *code = codes;
        for (int i = 0; i < ncodes; ++i, code += m) {
            float acc = 0.0f;
            for (int k = 0; k < m; ++k) acc += table[(size_t)h * k + code[k]];
            acc += dbnorms[i];
            const DistId cand(acc, i + 1);
            if ((int)heap.size() < nn) {
                heap.push_back(cand);
                std::push_heap(heap.begin(), heap.end());               // max-heap on (dist, id)
            } else if (cand < heap.front()) {
                std::pop_heap(heap.begin(), heap.end());
                heap.back() = cand;
                std::push_heap(heap.begin(), heap.end());
            }
        }
        std::sort_heap(heap.begin(), heap.end());                       // ascending (dist, id)
        for (int r = 0; r < nn; ++r) {
            dists[(size_t)q * nn + r] = heap[(size_t)r].first;
            idx[(size_t)q * nn + r] = heap[(size_t)r].second;
        }
    }
}

}  // namespace

extern "C" int lsq_linscan_aqd_query_extra_byte(float *dists, int *idx, const unsigned char *codes, const float *queries,
                                                const float *codebooks, const float *dbnorms, int nqueries, int ncodes,
                                                int m, int h, int d, int nn, int nthreads) {
    if (nqueries < 0 || ncodes < 0 || m < 1 || h < 1 || h > 256 || d < 1 || nn < 1) {
        lsq_set_error("lsq_linscan_aqd_query_extra_byte: bad shape nq=%d n=%d m=%d h=%d d=%d nn=%d", nqueries, ncodes, m, h, d, nn);
        return LSQ_EINVAL;
    }
    if (nn > ncodes) { lsq_set_error("lsq_linscan_aqd_query_extra_byte: nn=%d exceeds the database size %d", nn, ncodes); return LSQ_EINVAL; }
    if (nqueries == 0) return LSQ_OK;
    if (!dists || !idx || !codes || !queries || !codebooks || !dbnorms) { lsq_set_error("lsq_linscan_aqd_query_extra_byte: null pointer"); return LSQ_EINVAL; }
    int nt = nthreads > 0 ? nthreads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > nqueries) nt = nqueries;
    std::vector<std::thread> pool;
    pool.reserve((size_t)nt);
    for (int t = 0; t < nt; ++t) {
        const int q0 = (int)((int64_t)nqueries * t / nt), q1 = (int)((int64_t)nqueries * (t + 1) / nt);
        pool.emplace_back(scan_queries, dists, idx, codes, queries, codebooks, dbnorms, q0, q1, ncodes, m, h, d, nn);
    }
    for (auto &th : pool) th.join();
    return LSQ_OK;
}
