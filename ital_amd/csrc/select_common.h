// Arg-extreme, selection record and batch bookkeeping shared by the selection kernels (select.hip) and by the scorers that
// end a greedy step with the selection themselves (score.hip: one rank, small problems -- every launch saved is ~6 us of a
// 3 ms round).  np.argmax / np.argmin semantics of the reference (ital/ital.py:130, ital/mcmi.py:77): the FIRST extreme in
// candidate-list order wins, and a NaN beats every number (first NaN wins).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ital_hip.h"

namespace ital {

struct Best {
    double val;
    int64_t pos;  // global list position, < 0: nothing
    int64_t loc;  // local position on the rank that holds it
};

// true if a precedes b under "first extreme, NaN wins"
__device__ __forceinline__ bool better(const Best& a, const Best& b, int mode) {
    if (a.pos < 0) return false;
    if (b.pos < 0) return true;
    const bool an = isnan(a.val), bn = isnan(b.val);
    if (an != bn) return an;
    if (an) return a.pos < b.pos;
    if (a.val != b.val) return mode == 0 ? a.val > b.val : a.val < b.val;
    return a.pos < b.pos;
}

__device__ __forceinline__ Best wave_best(Best v, int mode) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        Best o;
        o.val = __shfl_xor(v.val, off, 64);
        o.pos = __shfl_xor(v.pos, off, 64);
        o.loc = __shfl_xor(v.loc, off, 64);
        if (better(o, v, mode)) v = o;
    }
    return v;
}

static __device__ Best block_best(Best v, int mode) {
    __shared__ double sval[16];
    __shared__ int64_t spos[16], sloc[16];
    v = wave_best(v, mode);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) { sval[wave] = v.val; spos[wave] = v.pos; sloc[wave] = v.loc; }
    __syncthreads();
    Best r = {0.0, -1, 0};
    if (wave == 0) {
        if (lane < nw) { r.val = sval[lane]; r.pos = spos[lane]; r.loc = sloc[lane]; }
        r = wave_best(r, mode);
    }
    return r;  // valid in wave 0
}

// Best live position of the strided range first, first + step, ...: four positions per trip, their flags and values
// loaded before any is compared (the single-workgroup selections are a chain of memory round trips otherwise).
__device__ __forceinline__ Best scan_best(const double* __restrict__ mi, const uint8_t* __restrict__ alive, int64_t n_cand,
                                          int64_t first, int64_t step, int64_t pos_offset, const int64_t* __restrict__ gpos,
                                          int mode) {
    Best v = {0.0, -1, 0};
    for (int64_t p = first; p < n_cand; p += 4 * step) {
        bool live[4];
        double val[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int64_t q = p + u * step;
            const bool in = q < n_cand;
            live[u] = in && alive[in ? q : 0] != 0;
            val[u] = mi[in ? q : 0];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (!live[u]) continue;
            const int64_t q = p + u * step;
            Best c = {val[u], gpos ? gpos[q] : pos_offset + q, q};
            if (better(c, v, mode)) v = c;
        }
    }
    return v;
}

struct RecordArgs {
    const int32_t* cand; int64_t pos_offset; const int64_t* gpos; int64_t row_offset; int rank, mode, nparts;
    const double *mu, *s2, *X, *xnorm; int ldx; const double* V; int64_t ldv; int m, ldw;
    const double* C; int64_t ldc; int nprev, kmax;
    const double* work; double* record; const int* status;
};

// Packs the record of the winner `v` (valid in thread 0 of the block).
static __device__ void record_body(const RecordArgs& a, Best v) {
    __shared__ int64_t s_pos, s_loc;
    __shared__ double s_val;
    if (threadIdx.x == 0) { s_pos = v.pos; s_val = v.val; s_loc = v.loc; }
    __syncthreads();
    const int64_t gpos = s_pos;
    double* rec = a.record;
    const int rec_len = ITAL_REC_HEADER + a.ldx + a.ldw + a.kmax;
    const double status = a.status ? (double)*a.status : 0.0;   // the rank's status word travels with its record
    if (gpos < 0) {
        for (int i = threadIdx.x; i < rec_len; i += blockDim.x) rec[i] = (i == 1) ? -1.0 : (i == 8 ? status : 0.0);
        return;
    }
    const int64_t lp = s_loc;
    const int row = a.cand[lp];
    if (threadIdx.x == 0) {
        rec[0] = s_val;
        rec[1] = (double)gpos;
        rec[2] = (double)(a.row_offset + row);
        rec[3] = a.mu[row];
        rec[4] = a.s2[row];
        rec[5] = a.xnorm[row];
        rec[6] = (double)a.rank;
        rec[7] = (double)lp;
        rec[8] = status;
        rec[9] = 0.0;
    }
    for (int k = threadIdx.x; k < a.ldx; k += blockDim.x) rec[ITAL_REC_HEADER + k] = a.X[(int64_t)row * a.ldx + k];
    for (int r = threadIdx.x; r < a.ldw; r += blockDim.x)
        rec[ITAL_REC_HEADER + a.ldx + r] = r < a.m ? a.V[(int64_t)r * a.ldv + row] : 0.0;
    for (int b = threadIdx.x; b < a.kmax; b += blockDim.x)
        rec[ITAL_REC_HEADER + a.ldx + a.ldw + b] = b < a.nprev ? a.C[(int64_t)b * a.ldc + row] : 0.0;
}

static __device__ void resolve_body(const double* __restrict__ records, int world, int rec_len, int rank, int mode, int slot,
                             ital_batch b, uint8_t* __restrict__ alive, int64_t* __restrict__ ret) {
    __shared__ int s_win;
    if (threadIdx.x == 0) {
        Best v = {0.0, -1, 0};
        int win = -1;
        long long status = 0;
        for (int w = 0; w < world; w++) {
            const double* r = records + (int64_t)w * rec_len;
            Best c = {r[0], (int64_t)r[1], 0};
            if (better(c, v, mode)) { v = c; win = w; }
            status |= (long long)r[8];
        }
        s_win = win;
        ret[b.kmax] |= status;   // every rank sees the OR of all ranks' status words: fall-backs are decided alike
    }
    __syncthreads();
    const int win = s_win;
    if (win < 0) {
        if (threadIdx.x == 0) ret[slot] = -1;
        return;
    }
    const double* r = records + (int64_t)win * rec_len;
    const int64_t gidx = (int64_t)r[2];
    if (threadIdx.x == 0) {
        b.bidx[slot] = gidx;
        b.bgpos[slot] = (int64_t)r[1];
        b.bmu[slot] = r[3];
        b.XBn[slot] = r[5];
        b.sig[slot * b.kmax + slot] = r[4];
        for (int q = 0; q < slot; q++) {
            const double c = r[ITAL_REC_HEADER + b.ldx + b.ldw + q];
            b.sig[slot * b.kmax + q] = c;
            b.sig[q * b.kmax + slot] = c;
        }
        // insert the new member into the order-by-data-index list
        int at = slot;
        while (at > 0 && b.bidx[b.bsort[at - 1]] > gidx) { b.bsort[at] = b.bsort[at - 1]; at--; }
        b.bsort[at] = slot;
        ret[slot] = gidx;
        if ((int)r[6] == rank) alive[(int64_t)r[7]] = 0;
    }
    for (int k = threadIdx.x; k < b.ldx; k += blockDim.x) b.XB[(int64_t)slot * b.ldx + k] = r[ITAL_REC_HEADER + k];
    for (int q = threadIdx.x; q < b.ldw; q += blockDim.x) b.VB[(int64_t)slot * b.ldw + q] = r[ITAL_REC_HEADER + b.ldx + q];
}


// ---- selection as the tail of a scoring kernel ("last block" pattern) ------------------------------------------------
// Every block of the scoring launch reduces the candidates it scored to one partial (value, list position, local
// position); the block that finishes last -- a ticket counter tells -- reduces the partials of the whole step, packs the
// winner's record and appends it to the batch state: what ital_select_fused does in a launch of its own.
struct SelectTail {
    RecordArgs rec;          // rec.record: scratch of ITAL_REC_HEADER + ldx + ldw + kmax doubles
    int slot;
    ital_batch b;
    uint8_t* alive;
    int64_t* ret;            // nullptr: stop after the record (what ital_select_local does)
    double* parts;           // [3 * nparts] block partials
    unsigned int* counter;   // ticket counter, zero before the launch; reset by the finishing block
    int enabled;
};

// v: the block's best, valid in thread 0.  part: index of this block among the `nparts` partials of the step.  `finishing`:
// this launch is the step's last one (its `nblocks` blocks take tickets); earlier launches (slabs) only leave partials.
static __device__ void select_tail(const SelectTail& s, Best v, int part, int nparts, bool finishing, int nblocks) {
    __shared__ int s_last;
    if (threadIdx.x == 0) {
        volatile double* pw = s.parts + 3 * (int64_t)part;
        pw[0] = v.val;
        pw[1] = (double)v.pos;
        pw[2] = (double)v.loc;
        int last = 0;
        if (finishing) {
            __threadfence();
            last = atomicAdd(s.counter, 1u) == (unsigned)(nblocks - 1);
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    Best w = {0.0, -1, 0};
    const volatile double* pr = s.parts;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) {
        Best c = {pr[3 * i], (int64_t)pr[3 * i + 1], (int64_t)pr[3 * i + 2]};
        if (better(c, w, s.rec.mode)) w = c;
    }
    w = block_best(w, s.rec.mode);
    record_body(s.rec, w);
    if (!s.ret) {                       // several ranks: the record goes into the exchange, ital_select_resolve follows
        if (threadIdx.x == 0) *s.counter = 0;
        return;
    }
    __threadfence_block();
    __syncthreads();
    const int rec_len = ITAL_REC_HEADER + s.rec.ldx + s.rec.ldw + s.rec.kmax;
    resolve_body(s.rec.record, 1, rec_len, s.rec.rank, s.rec.mode, s.slot, s.b, s.alive, s.ret);
    if (threadIdx.x == 0) *s.counter = 0;
}

}  // namespace ital
