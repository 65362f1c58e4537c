// mild_histogram_matching of the inference notebook (test.ipynb c1:69-85) on the device, per sample:
//   s_vals, bin_idx, s_counts = np.unique(source);  t_vals, t_counts = np.unique(reference)
//   s_q = cumsum(s_counts) / sum;  t_q = cumsum(t_counts) / sum            (float64)
//   matched = np.interp(s_q, t_q, t_vals)[bin_idx]
//   adjusted = (1 - weight) * source + weight * matched                     (float64 result)
// Restated on sorted arrays instead of unique ones: for an element of value v, s_q = (#source elements <= v) / n_s
// = upper_bound(v) / n_s; the unique target value of rank j ends its run at a position e_j of the sorted target and
// t_q[j] = e_j / n_t, so np.interp's bracket [t_q[j], t_q[j+1]) is found by locating the largest run end e with
// e / n_t <= s_q.  All quantiles are the same IEEE double divisions numpy performs, so the bracket decisions are
// identical.  Sorting: hipcub::DeviceRadixSort (rocPRIM) -- the one library primitive of this file.
#include <hipcub/hipcub.hpp>

#include "common.h"
#include "../../include/gandanet.h"

namespace {

__global__ void iota_kernel(unsigned int* idx, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) idx[i] = (unsigned int)i;
}

__device__ __forceinline__ long upper_bound(const float* a, long n, float v) {   // first index with a[i] > v
    long lo = 0, hi = n;
    while (lo < hi) {
        const long mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ long lower_bound(const float* a, long n, float v) {   // first index with a[i] >= v
    long lo = 0, hi = n;
    while (lo < hi) {
        const long mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void hist_match_kernel(const float* __restrict__ skeys, const unsigned int* __restrict__ sidx,
                                                        long ns, const float* __restrict__ tkeys, long nt, double weight,
                                                        double* __restrict__ out) {
#pragma clang fp contract(off)      // numpy rounds the product and the sum separately
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < ns; i += (long)gridDim.x * 256) {
        const float v = skeys[i];
        const double x = (double)upper_bound(skeys, ns, v) / (double)ns;          // s_q of this value
        // largest count c in [0, nt] with c / nt <= x
        long c = (long)floor(x * (double)nt);
        if (c > nt) c = nt;
        if (c < 0) c = 0;
        while (c < nt && (double)(c + 1) / (double)nt <= x) ++c;
        while (c > 0 && (double)c / (double)nt > x) --c;
        double matched;
        long e = 0;                                                               // largest run end <= c
        if (c > 0) {
            const float val = tkeys[c - 1];
            e = upper_bound(tkeys, nt, val) == c ? c : lower_bound(tkeys, nt, val);
        }
        if (e == 0) {
            matched = (double)tkeys[0];                                           // x < t_q[0]
        } else if (e == nt) {
            matched = (double)tkeys[nt - 1];                                      // last unique value
        } else {
            const double fpj = (double)tkeys[e - 1], xpj = (double)e / (double)nt;
            if (xpj == x) {
                matched = fpj;
            } else {
                const float nxt = tkeys[e];
                const double xpn = (double)upper_bound(tkeys, nt, nxt) / (double)nt;
                const double slope = ((double)nxt - fpj) / (xpn - xpj);
                matched = slope * (x - xpj) + fpj;
            }
        }
        const float t1 = (float)(1.0 - weight) * v;                       // float32 array times a Python scalar
        out[sidx[i]] = (double)t1 + weight * matched;
    }
}

}  // namespace

// workspace: sorted source keys + indices (in and out), sorted target keys (in copy and out) + the radix sort's own temp
static size_t sort_temp_bytes(long ns, long nt) {
    size_t a = 0, b = 0;
    hipcub::DeviceRadixSort::SortPairs(nullptr, a, (const float*)nullptr, (float*)nullptr, (const unsigned int*)nullptr,
                                       (unsigned int*)nullptr, (int)ns);
    hipcub::DeviceRadixSort::SortKeys(nullptr, b, (const float*)nullptr, (float*)nullptr, (int)nt);
    return a > b ? a : b;
}
static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" size_t gd_hist_match_ws_bytes(long ns, long nt) {
    if (ns <= 0 || nt <= 0 || ns > 0x7fffffffL || nt > 0x7fffffffL) return 0;
    return align256(ns * 4) * 3 + align256(nt * 4) + align256(sort_temp_bytes(ns, nt)) + 256;
}

extern "C" int gd_hist_match(const float* src, const float* ref, int B, long ns, long nt, double weight, double* out,
                             void* ws, size_t ws_bytes, void* stream) {
    GD_CHECK_ARG(src && ref && out && ws && B > 0 && ns > 0 && nt > 0, "gd_hist_match: bad arguments");
    GD_CHECK_ARG(ns <= 0x7fffffffL && nt <= 0x7fffffffL, "gd_hist_match: samples of more than 2^31 elements");
    GD_CHECK_ARG(ws_bytes >= gd_hist_match_ws_bytes(ns, nt), "gd_hist_match: workspace smaller than gd_hist_match_ws_bytes");
    hipStream_t s = (hipStream_t)stream;
    char* p = (char*)(((size_t)ws + 255) & ~(size_t)255);
    float* skeys = (float*)p;           p += align256(ns * 4);
    unsigned int* iidx = (unsigned int*)p;  p += align256(ns * 4);
    unsigned int* sidx = (unsigned int*)p;  p += align256(ns * 4);
    float* tkeys = (float*)p;           p += align256(nt * 4);
    size_t temp = sort_temp_bytes(ns, nt);
    void* tmp = p;
    hipLaunchKernelGGL(iota_kernel, dim3(gd_cdiv(ns, 256) > 2048 ? 2048 : gd_cdiv(ns, 256)), dim3(256), 0, s, iidx, ns);
    for (int b = 0; b < B; ++b) {
        if (hipcub::DeviceRadixSort::SortPairs(tmp, temp, src + (long)b * ns, skeys, iidx, sidx, (int)ns, 0, 32, s) != hipSuccess ||
            hipcub::DeviceRadixSort::SortKeys(tmp, temp, ref + (long)b * nt, tkeys, (int)nt, 0, 32, s) != hipSuccess) {
            gd_set_error("gd_hist_match: radix sort failed");
            return -2;
        }
        hipLaunchKernelGGL(hist_match_kernel, dim3(gd_cdiv(ns, 256) > 2048 ? 2048 : gd_cdiv(ns, 256)), dim3(256), 0, s, skeys,
                           sidx, ns, tkeys, nt, weight, out + (long)b * ns);
    }
    GD_LAUNCH_CHECK();
    return 0;
}
