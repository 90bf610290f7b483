// Top-k of the predictive means on the device (reference ital/retrieval_base.py:64-75 `top_results`:
// np.argsort(rel_mean)[::-1][:k]) -- exact radix select, no N-sized transfer to the host.
//
// Order: value descending, a NaN before every number (np.argsort puts NaNs last, the reversal first), equal values by
// descending index (what reversing a stable ascending sort gives).  64-bit order-preserving keys; eight passes of
// 8 bits narrow the k-th largest key down (a histogram kernel over the values that still match the prefix, then a
// one-workgroup pick); a collect pass gathers everything above the threshold plus the ties; one workgroup sorts the
// k survivors in LDS.  The values stay L2-resident between the passes (8 MB at N = 1M).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ital_hip.h"
#include "ital_internal.h"

namespace ital {

constexpr int TOPK_MAX = ITAL_TOPK_MAX;      // survivors one workgroup sorts in LDS
constexpr int TOPK_EQ_CAP = 4096;            // ties at the threshold that are listed (more: the ordered scan below)

struct TopkWork {
    unsigned long long prefix;   // bits of the threshold key fixed so far (left aligned)
    unsigned long long k_rem;    // how many of the values that match the prefix are still wanted
    unsigned int n_gt;           // values above the threshold written so far
    unsigned int n_eq;           // values equal to the threshold listed so far
    unsigned int hist[256];
    unsigned long long eq_idx[TOPK_EQ_CAP];
};

__device__ __forceinline__ unsigned long long order_key(double x) {
    if (x != x) return ~0ull;                               // NaN: above every number
    if (x == 0.0) return 0x8000000000000000ull;             // -0 and +0 compare equal (numpy orders them by index)
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);    // ascending doubles <-> ascending keys
}

__global__ void topk_init_kernel(TopkWork* w, int k) {
    if (threadIdx.x == 0) { w->prefix = 0; w->k_rem = (unsigned long long)k; w->n_gt = 0; w->n_eq = 0; }
    if (threadIdx.x < 256) w->hist[threadIdx.x] = 0;
}

__global__ __launch_bounds__(256) void topk_hist_kernel(const double* __restrict__ v, int64_t n, int pass, TopkWork* w) {
    __shared__ unsigned int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const unsigned long long prefix = w->prefix;
    const int shift = 56 - 8 * pass;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long key = order_key(v[i]);
        const bool match = pass == 0 || ((key ^ prefix) >> (shift + 8)) == 0;
        if (match) atomicAdd(&h[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&w->hist[threadIdx.x], h[threadIdx.x]);
}

// The bin of this pass that holds the k_rem-th largest matching key; bins above it are taken whole.
__global__ __launch_bounds__(256) void topk_pick_kernel(TopkWork* w, int pass) {
    __shared__ unsigned long long above[257];
    if (threadIdx.x == 0) {
        unsigned long long acc = 0;
        above[256] = 0;
        for (int b = 255; b >= 0; b--) { above[b + 1] = acc; acc += w->hist[b]; }
        above[0] = acc;     // [b + 1]: matching keys in bins above b
    }
    __syncthreads();
    const unsigned long long k_rem = w->k_rem;
    const int b = threadIdx.x;
    const unsigned long long ab = above[b + 1], incl = ab + w->hist[b];
    __syncthreads();
    if (ab < k_rem && k_rem <= incl) {
        w->prefix |= (unsigned long long)b << (56 - 8 * pass);
        w->k_rem = k_rem - ab;
    }
    w->hist[b] = 0;
}

__global__ __launch_bounds__(256) void topk_collect_kernel(const double* __restrict__ v, int64_t n, int64_t index_offset,
                                                           TopkWork* w, double* __restrict__ out_vals,
                                                           int64_t* __restrict__ out_idx) {
    const unsigned long long thr = w->prefix;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double x = v[i];
        const unsigned long long key = order_key(x);
        if (key > thr) {
            const unsigned int at = atomicAdd(&w->n_gt, 1u);
            out_vals[at] = x;
            out_idx[at] = index_offset + i;
        } else if (key == thr) {
            const unsigned int at = atomicAdd(&w->n_eq, 1u);
            if (at < TOPK_EQ_CAP) w->eq_idx[at] = (unsigned long long)i;
        }
    }
}

// One workgroup: completes the survivors with the k_rem ties of largest index, then sorts all k by (key, index) descending.
__global__ __launch_bounds__(1024) void topk_finish_kernel(const double* __restrict__ v, int64_t n, int64_t index_offset,
                                                            TopkWork* w, int k, double* __restrict__ out_vals,
                                                            int64_t* __restrict__ out_idx) {
    extern __shared__ unsigned long long sm[];
    unsigned long long* skey = sm;                 // [P]
    long long* sidx = (long long*)(sm + TOPK_MAX);  // [P]
    const int n_gt = (int)w->n_gt;
    const int k_rem = (int)w->k_rem;
    const unsigned long long thr = w->prefix;
    const unsigned int n_eq = w->n_eq;
    int P = 1;
    while (P < k) P <<= 1;
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
        if (i < n_gt) { skey[i] = order_key(out_vals[i]); sidx[i] = out_idx[i]; }
        else { skey[i] = 0; sidx[i] = -1; }       // below every real entry
    }
    __syncthreads();
    if (n_eq <= TOPK_EQ_CAP) {
        // the listed ties: keep the k_rem largest indices (rank by counting, the list is short)
        for (unsigned int e = threadIdx.x; e < n_eq; e += blockDim.x) {
            const unsigned long long mine = w->eq_idx[e];
            int rank = 0;
            for (unsigned int f = 0; f < n_eq; f++) rank += (w->eq_idx[f] > mine) ? 1 : 0;
            if (rank < k_rem) { skey[n_gt + rank] = thr; sidx[n_gt + rank] = index_offset + (long long)mine; }
        }
    } else if (threadIdx.x < 64) {
        // degenerate input (thousands of equal values at the threshold): ordered scan from the highest index down, one wave
        int taken = 0;
        for (int64_t base = n; base > 0 && taken < k_rem; base -= 64) {
            const int64_t i = base - 1 - threadIdx.x;
            const bool hit = i >= 0 && order_key(v[i]) == thr;
            const unsigned long long m = __ballot(hit);
            const int before = __popcll(m & ((1ull << threadIdx.x) - 1ull));
            if (hit && taken + before < k_rem) { skey[n_gt + taken + before] = thr; sidx[n_gt + taken + before] = index_offset + i; }
            taken += __popcll(m);
        }
    }
    __syncthreads();
    // bitonic sort, descending by (key, index)
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = threadIdx.x; i < P / 2; i += blockDim.x) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const unsigned long long ka = skey[lo], kb = skey[hi];
                const long long ia = sidx[lo], ib = sidx[hi];
                const bool a_first = ka > kb || (ka == kb && ia > ib);
                if (a_first != desc) { skey[lo] = kb; skey[hi] = ka; sidx[lo] = ib; sidx[hi] = ia; }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        const long long gi = sidx[i];
        out_idx[i] = gi;
        out_vals[i] = gi >= 0 ? v[gi - index_offset] : 0.0;
    }
}

}  // namespace ital

using namespace ital;

extern "C" int64_t ital_topk_workspace(void) { return (int64_t)sizeof(TopkWork); }

extern "C" int ital_topk(const double* v, int64_t n, int64_t index_offset, int k, double* out_vals, int64_t* out_idx,
                         void* work, hipStream_t stream) {
    if (k < 1 || k > TOPK_MAX) return ital_fail(-22, "ital_topk: k must be in 1..ITAL_TOPK_MAX");
    if (n < k) return ital_fail(-22, "ital_topk: fewer values than k");
    if (!work) return ital_fail(-22, "ital_topk: workspace missing (ital_topk_workspace bytes)");
    TopkWork* w = static_cast<TopkWork*>(work);
    int blocks = (int)((n + 1023) / 1024);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    ITAL_LAUNCH(topk_init_kernel, dim3(1), dim3(256), 0, stream, w, k);
    for (int pass = 0; pass < 8; pass++) {
        ITAL_LAUNCH(topk_hist_kernel, dim3(blocks), dim3(256), 0, stream, v, n, pass, w);
        ITAL_LAUNCH(topk_pick_kernel, dim3(1), dim3(256), 0, stream, w, pass);
    }
    ITAL_LAUNCH(topk_collect_kernel, dim3(blocks), dim3(256), 0, stream, v, n, index_offset, w, out_vals, out_idx);
    static ItalLdsFlags lds_flags;
    const size_t lds = (size_t)TOPK_MAX * 16;
    if (const int rc = ital_raise_lds_limit(reinterpret_cast<const void*>(&topk_finish_kernel), (int)lds, lds_flags, "ital_topk"))
        return rc;
    ITAL_LAUNCH(topk_finish_kernel, dim3(1), dim3(1024), lds, stream, v, n, index_offset, w, k, out_vals, out_idx);
    return ital_check_launch("ital_topk");
}
