// (1) last-token / CLS pooling fused with L2 normalisation, forward and backward.
// Reference: modeling.py:224-236, rankpo_trainer.py:409-417, modeling.py:523-534.
//
// HBM-bound, tiny: per sample the forward reads one int64 mask row (L * 8 B) and ONE hidden row
// (d * s B) and writes d * s B; the backward writes the dense [N, L, d] gradient exactly once.
#include "common.hpp"

namespace {

constexpr int kPoolThreads = 256;
constexpr int kPoolWaves = kPoolThreads / RPO_WAVE;
constexpr int kFillPerThread = 4;

// First index of the minimum of mask[0..L) (torch.argmin semantics: ties -> smallest index).
__device__ __forceinline__ int block_argmin_first(const int64_t* __restrict__ m, int64_t L, int64_t* s_val,
                                                  int* s_idx) {
    int64_t best = INT64_MAX;
    int bidx = INT32_MAX;
    const int tid = threadIdx.x;
    if ((L & 1) == 0 && rpo_aligned16_dev(m)) {
        const longlong2* m2 = reinterpret_cast<const longlong2*>(m);
        for (int64_t i = tid; i < (L >> 1); i += kPoolThreads) {
            longlong2 v = m2[i];
            // strict '<' keeps the earliest index inside this thread's ascending scan
            if (v.x < best) { best = v.x; bidx = (int)(2 * i); }
            if (v.y < best) { best = v.y; bidx = (int)(2 * i + 1); }
        }
    } else {
        for (int64_t i = tid; i < L; i += kPoolThreads) {
            int64_t v = m[i];
            if (v < best) { best = v; bidx = (int)i; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        int64_t ov = __shfl_xor(best, o, 64);
        int oi = __shfl_xor(bidx, o, 64);
        if (ov < best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    const int w = tid >> 6;
    if ((tid & 63) == 0) { s_val[w] = best; s_idx[w] = bidx; }
    __syncthreads();
    best = s_val[0];
    bidx = s_idx[0];
#pragma unroll
    for (int i = 1; i < kPoolWaves; ++i) {
        int64_t ov = s_val[i];
        int oi = s_idx[i];
        if (ov < best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    return bidx;
}

template <typename T>
__global__ __launch_bounds__(kPoolThreads) void pool_normalize_fwd_kernel(
    const T* __restrict__ h, int64_t sn, int64_t sl, const int64_t* __restrict__ mask, int64_t L, int64_t d,
    int pool_mode, int normalize, float eps, T* __restrict__ out, int32_t* __restrict__ idx_out,
    float* __restrict__ norm_out) {
    __shared__ int64_t s_val[kPoolWaves];
    __shared__ int s_idx[kPoolWaves];
    __shared__ float s_red[kPoolWaves];
    const int64_t n = blockIdx.x;
    int idx = 0;
    if (pool_mode == RPO_POOL_LAST) {
        int am = block_argmin_first(mask + n * L, L, s_val, s_idx);
        idx = (int)(((int64_t)am - 1 + L) % L);   // (argmin - 1) mod L, python semantics
    }
    const T* row = h + n * sn + (int64_t)idx * sl;
    T* o = out + n * d;
    constexpr int V = Elem<T>::kVec;
    const bool vec = (d % V == 0) && rpo_aligned16_dev(row) && rpo_aligned16_dev(o);
    float ss = 0.f;
    if (vec && d <= (int64_t)kPoolThreads * V) {
        // the whole row fits one 16-byte vector per thread (d <= 2048 bf16 / 1024 f32: every encoder the reference names): ONE
        // load, the row stays in registers through the norm reduction -- one dependent memory round trip less per sample (round 5:
        // at an encode()-scale batch, N = 4096 x L = 512 x d = 2048, the kernel is a chain of round trips, not bandwidth)
        const int64_t c = (int64_t)threadIdx.x * V;
        Vec16<T> x;
#pragma unroll
        for (int k = 0; k < V; ++k) x.v[k] = 0.f;
        if (c < d) x.load(row + c);
        if (normalize) {
#pragma unroll
            for (int k = 0; k < V; ++k) ss += x.v[k] * x.v[k];
            ss = block_sum<kPoolWaves>(ss, s_red);
        }
        const float nrm1 = sqrtf(ss);
        if (c < d) {
            if (normalize) {
#pragma unroll
                for (int k = 0; k < V; ++k) x.v[k] = x.v[k] / fmaxf(nrm1, eps);
            }
            x.store(o + c);
        }
        if (threadIdx.x == 0) {
            idx_out[n] = idx;
            norm_out[n] = nrm1;
        }
        return;
    }
    if (normalize) {
        if (vec) {
            for (int64_t c = (int64_t)threadIdx.x * V; c < d; c += (int64_t)kPoolThreads * V) {
                Vec16<T> x;
                x.load(row + c);
#pragma unroll
                for (int k = 0; k < V; ++k) ss += x.v[k] * x.v[k];
            }
        } else {
            for (int64_t c = threadIdx.x; c < d; c += kPoolThreads) {
                float x = Elem<T>::ld(row + c);
                ss += x * x;
            }
        }
        ss = block_sum<kPoolWaves>(ss, s_red);
    }
    const float nrm = sqrtf(ss);
    if (vec) {
        for (int64_t c = (int64_t)threadIdx.x * V; c < d; c += (int64_t)kPoolThreads * V) {
            Vec16<T> x;
            x.load(row + c);   // second touch of the row: served by L1/L2
            if (normalize) {
#pragma unroll
                for (int k = 0; k < V; ++k) x.v[k] = x.v[k] / fmaxf(nrm, eps);
            }
            x.store(o + c);
        }
    } else {
        for (int64_t c = threadIdx.x; c < d; c += kPoolThreads) {
            float x = Elem<T>::ld(row + c);
            Elem<T>::st(o + c, normalize ? x / fmaxf(nrm, eps) : x);
        }
    }
    if (threadIdx.x == 0) {
        idx_out[n] = idx;
        norm_out[n] = nrm;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Forward, ONE WAVE PER SAMPLE (round 6).  The block-per-sample kernel above is a chain of dependent round trips with two
// block-wide barriers in it: at an encode()-scale batch (N = 4096 samples x L = 512 x d = 2048 bf16: 50 MB of algorithmic traffic)
// it moved 2.0 TB/s, a quarter of the HBM peak (profiles/r05_pmc_traffic.md).  Here a wave owns a sample: the whole mask row is
// requested in ONE round of 16-byte loads per lane (up to four in flight per lane: L <= 512; longer rows go round again), the
// argmin is a register scan + six wave shuffles (no LDS, no barrier), the pooled row is requested straight behind it (NV 16-byte
// loads per lane in flight, the row stays in registers through the norm), and four samples share a block so that a CU holds up to
// 32 samples in flight.  Rows of up to 8 vectors per lane (d <= 4096 bf16 / fp16, 2048 f32); anything else -- and rows that are not
// 16-byte friendly -- takes the block kernel.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kPoolSamplesPerBlock = kPoolThreads / RPO_WAVE;

__device__ __forceinline__ void argmin_take(int64_t v, int i, int64_t& best, int& bidx) {
    if (v < best) { best = v; bidx = i; }              // strict '<': the earliest index of a lane's ascending scan wins ties
}

// First index of the minimum of m[0..L) by one wave (torch.argmin: ties -> smallest index).
__device__ __forceinline__ int wave_argmin_first(const int64_t* __restrict__ m, int64_t L, int lane) {
    int64_t best = INT64_MAX;
    int bidx = INT32_MAX;
    if ((L & 1) == 0 && rpo_aligned16_dev(m)) {
        const longlong2* m2 = reinterpret_cast<const longlong2*>(m);
        const int64_t np = L >> 1;                      // 16-byte pieces
        for (int64_t p0 = 0; p0 < np; p0 += 4 * RPO_WAVE) {
            longlong2 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {               // four independent loads in flight
                const int64_t p = p0 + j * RPO_WAVE + lane;
                v[j] = p < np ? m2[p] : longlong2{INT64_MAX, INT64_MAX};
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = (int)(2 * (p0 + j * RPO_WAVE + lane));
                argmin_take(v[j].x, i, best, bidx);
                argmin_take(v[j].y, i + 1, best, bidx);
            }
        }
    } else {
        for (int64_t i = lane; i < L; i += RPO_WAVE) argmin_take(m[i], (int)i, best, bidx);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int64_t ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bidx, o, 64);
        if (ov < best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    return bidx;
}

template <typename T, int NV>
__global__ __launch_bounds__(kPoolThreads) void pool_normalize_fwd_wave_kernel(
    const T* __restrict__ h, int64_t sn, int64_t sl, const int64_t* __restrict__ mask, int64_t N, int64_t L, int64_t d,
    int pool_mode, int normalize, float eps, T* __restrict__ out, int32_t* __restrict__ idx_out, float* __restrict__ norm_out) {
    const int lane = threadIdx.x & (RPO_WAVE - 1);
    const int64_t n = (int64_t)blockIdx.x * kPoolSamplesPerBlock + (threadIdx.x >> 6);
    if (n >= N) return;                                 // (whole waves: no barrier anywhere in this kernel)
    int idx = 0;
    if (pool_mode == RPO_POOL_LAST) {
        const int am = wave_argmin_first(mask + n * L, L, lane);
        idx = (int)(((int64_t)am - 1 + L) % L);         // (argmin - 1) mod L, python semantics
    }
    constexpr int V = Elem<T>::kVec;
    const T* row = h + n * sn + (int64_t)idx * sl;
    T* o = out + n * d;
    Vec16<T> x[NV];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {                      // NV loads in flight; vector j of lane l = elements (j * 64 + l) * V ...
        const int64_t c = ((int64_t)j * RPO_WAVE + lane) * V;
        if (c < d) x[j].load_nt(row + c);               // (read once: streaming)
        else {
#pragma unroll
            for (int k = 0; k < V; ++k) x[j].v[k] = 0.f;
        }
    }
    if (normalize) {
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int k = 0; k < V; ++k) ss += x[j].v[k] * x[j].v[k];
        ss = wave_sum(ss);
    }
    const float nrm = sqrtf(ss);
    const float den = fmaxf(nrm, eps);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int64_t c = ((int64_t)j * RPO_WAVE + lane) * V;
        if (c < d) {
            if (normalize) {
#pragma unroll
                for (int k = 0; k < V; ++k) x[j].v[k] = x[j].v[k] / den;
            }
            x[j].store(o + c);
        }
    }
    if (lane == 0) {
        idx_out[n] = idx;
        norm_out[n] = nrm;
    }
}

// Backward.  grid = (bx, N): sample n = blockIdx.y.  Block x == 0 of every sample first produces dx_n
// (written into dh / drow); then all blocks of the sample zero-fill the rest of dh[n] with 16-byte
// non-temporal stores, skipping the pooled row (no 64-bit index arithmetic in the fill loop).
template <typename T>
__global__ __launch_bounds__(kPoolThreads) void pool_normalize_bwd_kernel(
    const T* __restrict__ g, const T* __restrict__ y, const int32_t* __restrict__ idx,
    const float* __restrict__ norm, int64_t L, int64_t d, int normalize, float eps, T* __restrict__ dh,
    T* __restrict__ drow, int vec_fill) {
    __shared__ float s_red[kPoolWaves];
    const int64_t n = blockIdx.y;
    const int my_idx = idx[n];
    if (blockIdx.x == 0) {
        const T* gn = g + n * d;
        const T* yn = y + n * d;
        const float nrm = norm[n];
        float proj = 0.f;
        const bool through_norm = normalize && (nrm >= eps);
        if (through_norm) {
            for (int64_t c = threadIdx.x; c < d; c += kPoolThreads) proj += Elem<T>::ld(gn + c) * Elem<T>::ld(yn + c);
            proj = block_sum<kPoolWaves>(proj, s_red);
        }
        const float inv = normalize ? 1.0f / fmaxf(nrm, eps) : 1.0f;
        T* dst_h = dh ? dh + (n * L + my_idx) * d : nullptr;
        T* dst_r = drow ? drow + n * d : nullptr;
        for (int64_t c = threadIdx.x; c < d; c += kPoolThreads) {
            float gv = Elem<T>::ld(gn + c);
            float v = through_norm ? (gv - Elem<T>::ld(yn + c) * proj) * inv : gv * inv;
            if (dst_h) Elem<T>::st(dst_h + c, v);
            if (dst_r) Elem<T>::st(dst_r + c, v);
        }
    }
    if (!dh) return;
    T* base = dh + n * L * d;
    if (vec_fill) {
        constexpr int V = Elem<T>::kVec;
        const unsigned cpr = (unsigned)(d / V);               // 16-byte chunks per row
        const unsigned total = (unsigned)L * cpr;              // host guarantees < 2^31
        const uint4_t z = {0u, 0u, 0u, 0u};
        uint4_t* dst = reinterpret_cast<uint4_t*>(base);
        // block b owns the contiguous chunks [b * 4 * 256, (b + 1) * 4 * 256): 4 streaming stores per thread
#pragma unroll
        for (int k = 0; k < kFillPerThread; ++k) {
            const unsigned i = (blockIdx.x * kFillPerThread + k) * kPoolThreads + threadIdx.x;
            if (i < total && (int)(i / cpr) != my_idx) __builtin_nontemporal_store(z, dst + i);
        }
    } else {
        const int64_t total = L * d;
        for (int64_t i = (int64_t)blockIdx.x * kPoolThreads + threadIdx.x; i < total;
             i += (int64_t)gridDim.x * kPoolThreads) {
            if ((int)(i / d) != my_idx) Elem<T>::st(base + i, 0.f);
        }
    }
}

template <typename T>
int launch_fwd(const void* h, int64_t sn, int64_t sl, const int64_t* mask, int64_t N, int64_t L, int64_t d,
               int pool_mode, int normalize, float eps, void* out, int32_t* idx_out, float* norm_out,
               hipStream_t st) {
    constexpr int V = Elem<T>::kVec;
    const int64_t vecs = rpo_cdiv(d, (int64_t)V * RPO_WAVE);           // 16-byte vectors per lane of a wave that owns a row
    // every row 16-byte friendly: base pointers aligned and all strides multiples of the vector (the pooled row of sample n sits at
    // h + n sn + idx sl for an idx only the kernel knows)
    // ... and only where it measured faster (profiles/r06_pool_normalize.md, kernel durations from rocprofv3's trace on cold data):
    // last-token pooling of MANY samples with mask rows a wave reads in one or two rounds -- 4096 x 512 x 2048: 15.2 vs 15.7 us,
    // 16384 x 128 x 4096: 61.5 vs 70.6.  One long mask row per wave loses to 256 threads on it (64 x 4096: 10.9 vs 7.1 us), and
    // without a mask (CLS / packed rows) the two forms are the same two round trips (4096 rows: 11.1 vs 10.2).
    const bool wave_ok = d % V == 0 && vecs <= 8 && rpo_aligned16(h) && rpo_aligned16(out) && sn % V == 0 && sl % V == 0 &&
                         pool_mode == RPO_POOL_LAST && N >= 512 && L <= 1024;
    if (wave_ok) {
        const dim3 grid((unsigned)rpo_cdiv(N, kPoolSamplesPerBlock)), block(kPoolThreads);
#define RPO_POOL_WAVE(NV)                                                                                          \
        RPO_LAUNCH((pool_normalize_fwd_wave_kernel<T, NV>), grid, block, 0, st, (const T*)h, sn, sl, mask, N, L, d, pool_mode, \
                   normalize, eps, (T*)out, idx_out, norm_out)
        if (vecs <= 1) RPO_POOL_WAVE(1);
        else if (vecs <= 2) RPO_POOL_WAVE(2);
        else if (vecs <= 4) RPO_POOL_WAVE(4);
        else RPO_POOL_WAVE(8);
#undef RPO_POOL_WAVE
        return rpo_launch_status();
    }
    RPO_LAUNCH(pool_normalize_fwd_kernel<T>, dim3((unsigned)N), dim3(kPoolThreads), 0, st,
                       (const T*)h, sn, sl, mask, L, d, pool_mode, normalize, eps, (T*)out, idx_out, norm_out);
    return rpo_launch_status();
}

template <typename T>
int launch_bwd(const void* g, const void* y, const int32_t* idx, const float* norm, int64_t N, int64_t L,
               int64_t d, int normalize, float eps, void* dh, void* drow, hipStream_t st) {
    constexpr int V = Elem<T>::kVec;
    const bool vec_ok = dh && (d % V == 0) && rpo_aligned16(dh) && (L * (d / V) < (int64_t)1 << 31);
    int64_t bx = 1;
    if (dh) {
        const int64_t per_sample = vec_ok ? L * (d / V) : L * d;
        bx = rpo_cdiv(per_sample, (int64_t)kPoolThreads * (vec_ok ? kFillPerThread : 8));
        if (bx < 1) bx = 1;
    }
    RPO_LAUNCH(pool_normalize_bwd_kernel<T>, dim3((unsigned)bx, (unsigned)N), dim3(kPoolThreads), 0, st,
                       (const T*)g, (const T*)y, idx, norm, L, d, normalize, eps, (T*)dh, (T*)drow,
                       vec_ok ? 1 : 0);
    return rpo_launch_status();
}

}  // namespace

extern "C" int rpo_pool_normalize_fwd(const void* h, int64_t h_stride_n, int64_t h_stride_l, const int64_t* mask,
                                      int64_t N, int64_t L, int64_t d, int dtype, int pool_mode, int normalize,
                                      float eps, void* out, int32_t* idx_out, float* norm_out,
                                      rpo_stream_t stream) {
    if (!h || !out || !idx_out || !norm_out || N <= 0 || L <= 0 || d <= 0) return RPO_ERR_INVALID_ARG;
    if (pool_mode != RPO_POOL_LAST && pool_mode != RPO_POOL_CLS) return RPO_ERR_INVALID_ARG;
    if (pool_mode == RPO_POOL_LAST && !mask) return RPO_ERR_INVALID_ARG;
    if (L > INT32_MAX || N > INT32_MAX) return RPO_ERR_UNSUPPORTED;          // (the backward's grid holds N in its y dimension: 65535 there)
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case RPO_DT_F32:
            return launch_fwd<float>(h, h_stride_n, h_stride_l, mask, N, L, d, pool_mode, normalize, eps, out,
                                     idx_out, norm_out, st);
        case RPO_DT_BF16:
            return launch_fwd<bf16_t>(h, h_stride_n, h_stride_l, mask, N, L, d, pool_mode, normalize, eps, out,
                                      idx_out, norm_out, st);
        case RPO_DT_F16:
            return launch_fwd<f16_t>(h, h_stride_n, h_stride_l, mask, N, L, d, pool_mode, normalize, eps, out,
                                     idx_out, norm_out, st);
        default:
            return RPO_ERR_INVALID_ARG;
    }
}

extern "C" int rpo_pool_normalize_bwd(const void* grad_out, const void* out, const int32_t* idx,
                                      const float* norm, int64_t N, int64_t L, int64_t d, int dtype,
                                      int normalize, float eps, void* dh, void* drow, rpo_stream_t stream) {
    if (!grad_out || !out || !idx || !norm || N <= 0 || L <= 0 || d <= 0) return RPO_ERR_INVALID_ARG;
    if (!dh && !drow) return RPO_ERR_INVALID_ARG;
    if (L > INT32_MAX || N > 65535) return RPO_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case RPO_DT_F32:
            return launch_bwd<float>(grad_out, out, idx, norm, N, L, d, normalize, eps, dh, drow, st);
        case RPO_DT_BF16:
            return launch_bwd<bf16_t>(grad_out, out, idx, norm, N, L, d, normalize, eps, dh, drow, st);
        case RPO_DT_F16:
            return launch_bwd<f16_t>(grad_out, out, idx, norm, N, L, d, normalize, eps, dh, drow, st);
        default:
            return RPO_ERR_INVALID_ARG;
    }
}
