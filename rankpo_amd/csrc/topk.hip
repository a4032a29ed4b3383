// Exact top-k selection over score rows, merged chunk by chunk ("next" row f3 of SURVEY.md §8: replaces the k-selection of
// faiss.IndexFlatIP.search, reference src/utils.py:58-80, which runs on the CPU there).  HBM-bound integer / compare work:
// every score is read exactly once (4 or 2 bytes per (query, corpus row) pair), nothing but the k winners is written.
//
// Order: value descending, ties by the smaller corpus index (= a stable argsort of -score, the oracle's definition).
// One block per query row.  The row's current winners live in LDS; the block streams the row in segments of 1024
// columns, collecting every element that beats the current k-th winner into an LDS candidate list (atomic counter), and
// re-selects (bitonic sort of winners + candidates, 4096 slots) whenever the list could overflow during the next segment.
// After the first few segments the k-th winner is high and almost nothing passes the filter, so the kernel is one
// coalesced pass over the scores.
#include "common.hpp"

namespace {

constexpr int kTopkThreads = 256;
constexpr int kTopkMaxK = 1024;      // winners kept per row
constexpr int kTopkCap = 2048;       // candidate list
constexpr int kTopkSeg = 1024;       // columns per segment (<= kTopkCap / 2)
constexpr int kTopkSlots = 4096;     // bitonic sort size >= kTopkMaxK + kTopkCap

__device__ __forceinline__ bool before(float va, long long ia, float vb, long long ib) {
    return va > vb || (va == vb && ia < ib);
}

template <typename T>
__device__ __forceinline__ float load_score(const T* p);
template <>
__device__ __forceinline__ float load_score<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float load_score<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }
template <>
__device__ __forceinline__ float load_score<f16_t>(const f16_t* p) { return (float)*p; }

// sorts the first n (a power of two, <= kTopkSlots) (value, index) slots: best first
template <int NT>
__device__ void bitonic_sort_desc(float* val, long long* idx, int tid, int n = kTopkSlots) {
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = tid; t < n / 2; t += NT) {
                const int lo = 2 * t - (t & (stride - 1));       // index with bit `stride` clear
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;                 // this run sorts best-first, the next one worst-first
                const float va = val[lo], vb = val[hi];
                const long long ia = idx[lo], ib = idx[hi];
                const bool a_first = before(va, ia, vb, ib);
                if (a_first != up) {
                    val[lo] = vb; val[hi] = va;
                    idx[lo] = ib; idx[hi] = ia;
                }
            }
        }
    }
    __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(kTopkThreads) void topk_merge_kernel(const T* __restrict__ scores, int64_t ld, int64_t cols,
                                                                  int64_t col0, int k, float* __restrict__ best_val,
                                                                  long long* __restrict__ best_idx, int first, int split, int64_t seg) {
    __shared__ float s_val[kTopkSlots];
    __shared__ long long s_idx[kTopkSlots];
    __shared__ int s_count;
    const int tid = threadIdx.x;
    // split > 1: block = (score row, column segment): LIST blockIdx.x = row * split + s holds the winners of columns
    // [s seg, (s + 1) seg) of row blockIdx.x / split (rpo_topk_merge_split); split == 1: one list per row
    const int64_t row = blockIdx.x;
    const int sgm = (int)(row % split);
    const T* srow = scores + (row / split) * ld + sgm * seg;
    col0 += sgm * seg;
    cols = cols - sgm * seg < 0 ? 0 : (cols - sgm * seg < seg ? cols - sgm * seg : seg);
    // slots [0, k): winners so far; [kTopkMaxK, kTopkMaxK + count): candidates; everything else: (-inf, max index)
    for (int i = tid; i < kTopkSlots; i += kTopkThreads) {
        float v = -INFINITY;
        long long ix = 0x7fffffffffffffffLL;
        if (i < k && !first) {
            v = best_val[row * k + i];
            ix = best_idx[row * k + i];
        }
        s_val[i] = v;
        s_idx[i] = ix;
    }
    if (tid == 0) s_count = 0;
    __syncthreads();
    float tv = s_val[k - 1];
    long long ti = s_idx[k - 1];
    for (int64_t c0 = 0; c0 < cols; c0 += kTopkSeg) {
        const int64_t cend = c0 + kTopkSeg < cols ? c0 + kTopkSeg : cols;
        for (int64_t c = c0 + tid; c < cend; c += kTopkThreads) {
            const float v = load_score<T>(srow + c);
            const long long ix = col0 + c;
            if (before(v, ix, tv, ti)) {
                const int pos = atomicAdd(&s_count, 1);          // < kTopkCap by construction
                s_val[kTopkMaxK + pos] = v;
                s_idx[kTopkMaxK + pos] = ix;
            }
        }
        __syncthreads();
        const int count = s_count;
        const bool last = cend == cols;
        if (count > kTopkCap - kTopkSeg || (last && count > 0)) {
            bitonic_sort_desc<kTopkThreads>(s_val, s_idx, tid);   // winners + candidates (+ padding) -> best first
            for (int i = k + tid; i < kTopkSlots; i += kTopkThreads) {   // drop everything past the k-th
                s_val[i] = -INFINITY;
                s_idx[i] = 0x7fffffffffffffffLL;
            }
            if (tid == 0) s_count = 0;
            __syncthreads();
            tv = s_val[k - 1];
            ti = s_idx[k - 1];
        }
    }
    for (int i = tid; i < k; i += kTopkThreads) {
        best_val[row * k + i] = s_val[i];
        best_idx[row * k + i] = s_idx[i];
    }
}

// ---- streaming path: 1024 threads per list, one 16-byte vector per thread and segment, kFastDepth segments in flight --------
// Requires 16-byte aligned rows (ld and the chunk base multiples of the vector length).  A segment is 1024 vectors
// (4096 f32 / 8192 bf16 columns).  Round 6 (the search of bench.py --workload encode spent 2.9 of its 6.7 ms here at 0.7 TB/s):
// the kernel was bound by its BARRIERS, not by memory -- two per segment plus a 78-stage bitonic sort of all 4096 slots per
// re-selection, at least one per call.  Now: the candidates sit right behind the k winners (slots [k, k + count)), a re-selection
// sorts only the next power of two of k + count slots (k = 100 with a few dozen candidates: 128 slots, 28 stages), and a GROUP of
// kFastDepth segments is filtered under one threshold between ONE pair of barriers.  A group that overflows the list (only while
// the k-th winner is still weak: the first columns of the first chunk) is replayed from its registers, segment by segment in
// quarters, with a re-selection after each.
constexpr int kFastThreads = 1024;
constexpr int kFastDepth = 4;

template <typename T>
__global__ __launch_bounds__(kFastThreads) void topk_merge_fast_kernel(const T* __restrict__ scores, int64_t ld, int64_t cols,
                                                                       int64_t col0, int k, float* __restrict__ best_val,
                                                                       long long* __restrict__ best_idx, int first, int split,
                                                                       int64_t seg) {
    constexpr int V = Elem<T>::kVec;
    constexpr int SEG = kFastThreads * V;
    __shared__ float s_val[kTopkSlots];
    __shared__ long long s_idx[kTopkSlots];
    __shared__ int s_count;
    const int tid = threadIdx.x;
    const int64_t row = blockIdx.x;                        // the LIST: (score row, column segment), see topk_merge_kernel
    const int sgm = (int)(row % split);
    const T* srow = scores + (row / split) * ld + sgm * seg;
    col0 += sgm * seg;
    cols = cols - sgm * seg < 0 ? 0 : (cols - sgm * seg < seg ? cols - sgm * seg : seg);
    // slots [0, k): winners so far; [k, k + count): candidates; everything behind: (-inf, max index) -- an invariant of every step
    for (int i = tid; i < kTopkSlots; i += kFastThreads) {
        float v = -INFINITY;
        long long ix = 0x7fffffffffffffffLL;
        if (i < k && !first) {
            v = best_val[row * k + i];
            ix = best_idx[row * k + i];
        }
        s_val[i] = v;
        s_idx[i] = ix;
    }
    if (tid == 0) s_count = 0;
    __syncthreads();
    float tv = s_val[k - 1];
    long long ti = s_idx[k - 1];
    const int cap = kTopkSlots - k < kTopkCap ? kTopkSlots - k : kTopkCap;      // candidates the list takes
    auto reselect = [&]() {                                   // winners + candidates -> k winners, list emptied (block-uniform call)
        const int count = s_count < cap ? s_count : cap;
        int n = 64;
        while (n < k + count) n <<= 1;
        bitonic_sort_desc<kFastThreads>(s_val, s_idx, tid, n);
        for (int i = k + tid; i < n; i += kFastThreads) {      // drop everything past the k-th
            s_val[i] = -INFINITY;
            s_idx[i] = 0x7fffffffffffffffLL;
        }
        if (tid == 0) s_count = 0;
        __syncthreads();
        tv = s_val[k - 1];
        ti = s_idx[k - 1];
    };
    auto offer = [&](float v, long long ix) {                  // append (v, ix) if it beats the k-th winner; false: the list is full
        if (before(v, ix, tv, ti)) {
            const int pos = atomicAdd(&s_count, 1);
            if (pos < cap) {
                s_val[k + pos] = v;
                s_idx[k + pos] = ix;
            }
        }
    };
    const int64_t nseg = (cols + SEG - 1) / SEG;
    Vec16<T> reg[kFastDepth];
    auto fetch = [&](int64_t sg, Vec16<T>& r) {
        const int64_t c = sg * SEG + (int64_t)tid * V;
        if (c + V <= cols) {
            r.load_nt(srow + c);
        } else {                                              // ragged tail: element-wise, -inf / never-selected padding
#pragma unroll
            for (int j = 0; j < V; ++j) r.v[j] = c + j < cols ? load_score<T>(srow + c + j) : -INFINITY;
        }
    };
    for (int64_t s0 = 0; s0 < nseg; s0 += kFastDepth) {
#pragma unroll
        for (int u = 0; u < kFastDepth; ++u)
            if (s0 + u < nseg) fetch(s0 + u, reg[u]);
        // Bootstrap: while the list holds fewer than k winners (k-th = -inf: the first columns of the first chunk) EVERYTHING passes
        // the filter and a whole group would overflow the list 16 times over.  The first quarter of the group's first segment
        // (<= 256 V <= cap elements) goes in alone and is settled; the k-th winner of those already turns away ~95 % of what
        // follows (k = 100), so the rest of the group takes the common path.  (Before: the group overflowed and was replayed in 16
        // quarters with a full 4096-slot sort each: 0.5 ms for the first chunk of a search against 0.09 for the others.)
        bool boot = tv == -INFINITY && ti == 0x7fffffffffffffffLL;            // block-uniform
        if (boot && first && s0 == 0) {
            // An EMPTY list (the first group of a search's first chunk; round 6: the fused search step left this call as the only
            // selection pass of a search, and 0.36 ms of its 0.36 were the two big sorts of the quarter bootstrap below): the k-th
            // largest of the 1024 threads' own maxima over the group bounds the group's k-th largest element from below -- k
            // elements, one per thread, reach it.  One 1024-slot sort of the maxima gives a threshold that lets ~k of the group's
            // 32 K elements through (a thread's maximum of 32 beats it with probability k / 1024) instead of the 5 % the quarter's own
            // k-th winner admits.  Ties at the threshold all pass (ti = max index); a flood of them overflows into the replay path,
            // which restarts from an empty threshold.  Fewer than k finite maxima (a chunk of < k columns): the quarter bootstrap.
            float m = -INFINITY;
#pragma unroll
            for (int u = 0; u < kFastDepth; ++u) {
                if (u >= nseg) break;
                const int64_t c = u * (int64_t)SEG + (int64_t)tid * V;
#pragma unroll
                for (int j = 0; j < V; ++j)
                    if (c + j < cols) m = fmaxf(m, (float)reg[u].v[j]);
            }
            s_val[tid] = m;                                   // (slots [0, 1024): all (-inf, max index) in an empty list)
            bitonic_sort_desc<kFastThreads>(s_val, s_idx, tid, kFastThreads);      // equal indices: the order of equal maxima is immaterial
            const float t0 = s_val[k - 1];
            __syncthreads();
            s_val[tid] = -INFINITY;                           // the invariant again
            __syncthreads();
            if (t0 > -INFINITY) {
                tv = t0;
                boot = false;                                 // the group takes the common path under this threshold
            }
        }
        if (boot) {
            if (tid < 256) {
                const int64_t c = s0 * SEG + (int64_t)tid * V;
#pragma unroll
                for (int j = 0; j < V; ++j)
                    if (c + j < cols) offer(reg[0].v[j], col0 + c + j);
            }
            __syncthreads();
            reselect();
        }
        const int before_count = s_count;                     // uniform: read after the barrier that ended the last step
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kFastDepth; ++u) {
            if (s0 + u >= nseg) break;
            if (boot && u == 0 && tid < 256) continue;        // already in
            const int64_t c = (s0 + u) * SEG + (int64_t)tid * V;
#pragma unroll
            for (int j = 0; j < V; ++j)
                if (c + j < cols) offer(reg[u].v[j], col0 + c + j);
        }
        __syncthreads();
        const int count = s_count;
        if (count > cap) {
            // overflow: drop this group's appends, settle what was there, replay the group segment by segment in quarters
            __syncthreads();
            if (tid == 0) s_count = before_count;
            __syncthreads();
            for (int i = k + before_count + tid; i < k + cap; i += kFastThreads) {
                s_val[i] = -INFINITY;
                s_idx[i] = 0x7fffffffffffffffLL;
            }
            __syncthreads();
            reselect();
#pragma unroll
            for (int u = 0; u < kFastDepth; ++u) {
                if (s0 + u >= nseg) break;
                const int64_t c = (s0 + u) * SEG + (int64_t)tid * V;
                for (int part = 0; part < 4; ++part) {
                    if ((tid >> 8) == part && !(boot && u == 0 && part == 0)) {     // (the bootstrap quarter is in already)
#pragma unroll
                        for (int j = 0; j < V; ++j)
                            if (c + j < cols) offer(reg[u].v[j], col0 + c + j);      // <= 256 V <= cap appends
                    }
                    __syncthreads();
                    if (s_count > 0) reselect();
                }
            }
        } else if (count > cap / 2) {
            reselect();
        }
    }
    __syncthreads();
    if (s_count > 0) reselect();
    for (int i = tid; i < k; i += kFastThreads) {
        best_val[row * k + i] = s_val[i];
        best_idx[row * k + i] = s_idx[i];
    }
}

// ---- the fused search step's second half (round 6): the candidates rpo_sim_topk_filter appended for a row + the row's k winners
// -> the row's k winners.  One block per row; sorts the next power of two of k + count slots (a few dozen candidates per corpus
// chunk once the winners have seen one chunk: 128 or 256 slots).  A list that overflowed (count > cap: its tail was dropped) raises
// *overflow and the caller redoes the chunk through the score matrix; the counter is reset for the next chunk either way.
__global__ __launch_bounds__(kTopkThreads) void topk_merge_cand_kernel(const float* __restrict__ cand_val,
                                                                       const long long* __restrict__ cand_idx, int* __restrict__ cand_cnt,
                                                                       int cap, int k, float* __restrict__ best_val,
                                                                       long long* __restrict__ best_idx, int* __restrict__ overflow) {
    __shared__ float s_val[kTopkSlots];
    __shared__ long long s_idx[kTopkSlots];
    const int tid = threadIdx.x;
    const int64_t row = blockIdx.x;
    int count = cand_cnt[row];
    if (count == 0) return;                      // uniform: nothing beat this row's k-th winner in the chunk
    __syncthreads();                             // everybody has read the counter
    if (tid == 0) {
        cand_cnt[row] = 0;
        if (count > cap) *overflow = 1;
    }
    count = count < cap ? count : cap;
    int n = 64;
    while (n < k + count) n <<= 1;
    for (int i = tid; i < n; i += kTopkThreads) {
        float v = -INFINITY;
        long long ix = 0x7fffffffffffffffLL;
        if (i < k) {
            v = best_val[row * k + i];
            ix = best_idx[row * k + i];
        } else if (i < k + count) {
            v = cand_val[row * cap + (i - k)];
            ix = cand_idx[row * cap + (i - k)];
        }
        s_val[i] = v;
        s_idx[i] = ix;
    }
    bitonic_sort_desc<kTopkThreads>(s_val, s_idx, tid, n);
    for (int i = tid; i < k; i += kTopkThreads) {
        best_val[row * k + i] = s_val[i];
        best_idx[row * k + i] = s_idx[i];
    }
}

}  // namespace

extern "C" int rpo_topk_merge_candidates(const float* cand_val, const int64_t* cand_idx, int32_t* cand_cnt, int64_t rows, int cap,
                                         int k, float* best_val, int64_t* best_idx, int32_t* overflow, rpo_stream_t stream) {
    if (!cand_val || !cand_idx || !cand_cnt || !best_val || !best_idx || !overflow || rows <= 0 || cap <= 0 || k <= 0)
        return RPO_ERR_INVALID_ARG;
    if (k > kTopkMaxK || k + cap > kTopkSlots || rows > INT32_MAX) return RPO_ERR_UNSUPPORTED;
    RPO_LAUNCH(topk_merge_cand_kernel, dim3((unsigned)rows), dim3(kTopkThreads), 0, (hipStream_t)stream, cand_val,
               (const long long*)cand_idx, (int*)cand_cnt, cap, k, best_val, (long long*)best_idx, (int*)overflow);
    return rpo_launch_status();
}

static int topk_launch(const void* scores, int64_t ld, int64_t rows, int64_t cols, int64_t col0, int k, int dtype, int split,
                       float* best_val, int64_t* best_idx, int first, rpo_stream_t stream) {
    if (!scores || !best_val || !best_idx || rows <= 0 || cols <= 0 || ld < cols || k <= 0 || col0 < 0 || split <= 0)
        return RPO_ERR_INVALID_ARG;
    if (k > kTopkMaxK || rows * split > INT32_MAX) return RPO_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (!rpo_dtype_ok(dtype)) return RPO_ERR_INVALID_ARG;
    const int V = 16 / rpo_elem_size(dtype);
    // a list's columns: ceil(cols / split) rounded up to whole 16-byte vectors, so that every segment starts vector-aligned
    const int64_t seg = split == 1 ? cols : rpo_cdiv(rpo_cdiv(cols, split), V) * V;
    const dim3 grid((unsigned)(rows * split));
    if (ld % V == 0 && rpo_aligned16(scores) && seg >= 4096) {
        if (dtype == RPO_DT_F32)
            RPO_LAUNCH(topk_merge_fast_kernel<float>, grid, dim3(kFastThreads), 0, st, (const float*)scores,
                       ld, cols, col0, k, best_val, (long long*)best_idx, first, split, seg);
        else if (dtype == RPO_DT_F16)
            RPO_LAUNCH(topk_merge_fast_kernel<f16_t>, grid, dim3(kFastThreads), 0, st,
                       (const f16_t*)scores, ld, cols, col0, k, best_val, (long long*)best_idx, first, split, seg);
        else
            RPO_LAUNCH(topk_merge_fast_kernel<bf16_t>, grid, dim3(kFastThreads), 0, st,
                       (const bf16_t*)scores, ld, cols, col0, k, best_val, (long long*)best_idx, first, split, seg);
        return rpo_launch_status();
    }
    if (dtype == RPO_DT_F32)
        RPO_LAUNCH(topk_merge_kernel<float>, grid, dim3(kTopkThreads), 0, st, (const float*)scores, ld, cols,
                   col0, k, best_val, (long long*)best_idx, first, split, seg);
    else if (dtype == RPO_DT_BF16)
        RPO_LAUNCH(topk_merge_kernel<bf16_t>, grid, dim3(kTopkThreads), 0, st, (const bf16_t*)scores, ld,
                   cols, col0, k, best_val, (long long*)best_idx, first, split, seg);
    else
        RPO_LAUNCH(topk_merge_kernel<f16_t>, grid, dim3(kTopkThreads), 0, st, (const f16_t*)scores, ld,
                   cols, col0, k, best_val, (long long*)best_idx, first, split, seg);
    return rpo_launch_status();
}

extern "C" int rpo_topk_merge(const void* scores, int64_t ld, int64_t rows, int64_t cols, int64_t col0, int k, int dtype,
                              float* best_val, int64_t* best_idx, int first, rpo_stream_t stream) {
    return topk_launch(scores, ld, rows, cols, col0, k, dtype, 1, best_val, best_idx, first, stream);
}

extern "C" int rpo_topk_merge_split(const void* scores, int64_t ld, int64_t rows, int64_t cols, int64_t col0, int k, int dtype,
                                    int split, float* best_val, int64_t* best_idx, int first, rpo_stream_t stream) {
    return topk_launch(scores, ld, rows, cols, col0, k, dtype, split, best_val, best_idx, first, stream);
}
