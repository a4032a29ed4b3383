// (3) RankPO paired scoring + loss + metrics, forward and backward.
// Reference: rankpo_trainer.py:436-443 (scores), 545-566 (rankpo_loss), 482-520 (loss mix + metrics).
//
// B is the per-device batch (8 in the reference scripts): the whole problem is 3*B*d*s bytes (98 KB at
// d = 2048 bf16), i.e. latency-bound.  Two launches forward (dots spread over B*G waves, then one block
// for the scalar work), one launch backward.
#include "common.hpp"

namespace {

constexpr int kDotThreads = 256;   // 4 waves
constexpr int kFinThreads = 256;

// scores[b, g] = <q_b, p_{b*G+g}>  (f32, unscaled).  grid = B, one wave per g (looping when G > 4).
template <typename T>
__global__ __launch_bounds__(kDotThreads) void grouped_dots_kernel(const T* __restrict__ q, const T* __restrict__ p,
                                                                    int64_t G, int64_t d, float* __restrict__ out) {
    const int64_t b = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const T* qb = q + b * d;
    constexpr int V = Elem<T>::kVec;
    const bool vec = (d % V == 0) && rpo_aligned16_dev(q) && rpo_aligned16_dev(p);
    for (int64_t g = wave; g < G; g += kDotThreads / 64) {
        const T* pg = p + (b * G + g) * d;
        float acc = 0.f;
        if (vec) {
            for (int64_t c = (int64_t)lane * V; c < d; c += 64 * V) {
                Vec16<T> a, x;
                a.load(qb + c);
                x.load(pg + c);
#pragma unroll
                for (int k = 0; k < V; ++k) acc = fmaf(a.v[k], x.v[k], acc);
            }
        } else {
            for (int64_t c = lane; c < d; c += 64) acc = fmaf(Elem<T>::ld(qb + c), Elem<T>::ld(pg + c), acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) out[b * G + g] = acc;
    }
}

__device__ __forceinline__ float log_sigmoid(float x) {
    // min(x,0) - log1p(exp(-|x|))
    return fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
}
__device__ __forceinline__ float sigmoidf(float x) {
    const float e = expf(-fabsf(x));
    return x >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
}

__global__ __launch_bounds__(kFinThreads) void rankpo_finalize_kernel(
    const float* __restrict__ scores, const float* __restrict__ ref_c, const float* __restrict__ ref_r, int64_t B,
    rpo_rankpo_params prm, float* __restrict__ losses_out, float* __restrict__ loss_out,
    float* __restrict__ metrics_out, float* __restrict__ dscores_out) {
    __shared__ float s_red[kFinThreads / 64];
    float acc[RPO_NUM_METRICS];
#pragma unroll
    for (int k = 0; k < RPO_NUM_METRICS; ++k) acc[k] = 0.f;
    const float invB = 1.0f / (float)B;
    const float T = prm.temperature;
    for (int64_t b = threadIdx.x; b < B; b += kFinThreads) {
        const float c = scores[2 * b], r = scores[2 * b + 1];
        const float rc = ref_c ? ref_c[b] : 0.f;
        const float rr = ref_r ? ref_r[b] : 0.f;
        float ds0 = 0.f, ds1 = 0.f, lb = 0.f;
        if (prm.rankpo_weight > 0.f) {
            float adv = c - r;                                           // :545
            if (!prm.reference_free) adv -= (rc - rr);                    // :546-548
            adv = adv / T;                                               // :550
            const float z = adv - prm.gamma_beta_ratio;                  // :554
            const float bz = prm.beta * z;
            float dz;
            if (prm.loss_type == RPO_LOSS_SIGMOID) {                      // :556-560
                lb = -log_sigmoid(bz) * (1.f - prm.label_smoothing) - log_sigmoid(-bz) * prm.label_smoothing;
                dz = -prm.beta * (1.f - prm.label_smoothing) * sigmoidf(-bz) +
                     prm.beta * prm.label_smoothing * sigmoidf(bz);
            } else {                                                     // :561-562
                lb = fmaxf(1.f - bz, 0.f);
                dz = (1.f - bz > 0.f) ? -prm.beta : 0.f;
            }
            dz = dz * (prm.rankpo_weight * invB) / T;
            ds0 += dz;
            ds1 -= dz;
            acc[RPO_METRIC_RANKPO_LOSS] += lb;
        }
        if (prm.sft_weight > 0.f) {                                      // :499-505
            const float t0 = c / T, t1 = r / T;
            const float m = fmaxf(t0, t1);
            const float lse = m + logf(expf(t0 - m) + expf(t1 - m));
            acc[RPO_METRIC_SFT_LOSS] += lse - t0;
            const float w = prm.sft_weight * invB / T;
            ds0 += (expf(t0 - lse) - 1.f) * w;
            ds1 += expf(t1 - lse) * w;
        }
        const float cr = prm.beta * (c - rc), rw = prm.beta * (r - rr);  // :509-510
        acc[RPO_METRIC_REWARDS_CHOSEN] += cr;
        acc[RPO_METRIC_REWARDS_REJECTED] += rw;
        acc[RPO_METRIC_REWARDS_ACCURACIES] += (cr > rw) ? 1.f : 0.f;
        acc[RPO_METRIC_REWARDS_MARGINS] += cr - rw;
        acc[RPO_METRIC_SCORES_CHOSEN] += c;
        acc[RPO_METRIC_SCORES_REJECTED] += r;
        acc[RPO_METRIC_SCORES_MARGINS] += c - r;
        losses_out[b] = lb;
        dscores_out[2 * b] = ds0;
        dscores_out[2 * b + 1] = ds1;
    }
    float tot[RPO_NUM_METRICS];
#pragma unroll
    for (int k = 0; k < RPO_NUM_METRICS; ++k) tot[k] = block_sum<kFinThreads / 64>(acc[k], s_red) * invB;
    if (threadIdx.x == 0) {
        float loss = 0.f;
        if (prm.rankpo_weight > 0.f) loss += prm.rankpo_weight * tot[RPO_METRIC_RANKPO_LOSS];   // :494
        if (prm.sft_weight > 0.f) loss += prm.sft_weight * tot[RPO_METRIC_SFT_LOSS];            // :504
        loss_out[0] = loss;
#pragma unroll
        for (int k = 0; k < RPO_NUM_METRICS; ++k) metrics_out[k] = tot[k];
    }
}

// grid = B.  dq_b = gl (ds0 p_{2b} + ds1 p_{2b+1});  dp_{2b+g} = gl ds_g q_b.
template <typename T>
__global__ __launch_bounds__(kDotThreads) void rankpo_bwd_kernel(const T* __restrict__ q, const T* __restrict__ p,
                                                                  const float* __restrict__ ds,
                                                                  const float* __restrict__ grad_loss, int64_t d,
                                                                  T* __restrict__ dq, T* __restrict__ dp) {
    const int64_t b = blockIdx.x;
    const float gl = grad_loss[0];
    const float s0 = gl * ds[2 * b], s1 = gl * ds[2 * b + 1];
    const T* qb = q + b * d;
    const T* p0 = p + (2 * b) * d;
    const T* p1 = p0 + d;
    constexpr int V = Elem<T>::kVec;
    const bool vec = (d % V == 0) && rpo_aligned16_dev(q) && rpo_aligned16_dev(p) &&
                     (!dq || rpo_aligned16_dev(dq)) && (!dp || rpo_aligned16_dev(dp));
    if (vec) {
        for (int64_t c = (int64_t)threadIdx.x * V; c < d; c += (int64_t)kDotThreads * V) {
            Vec16<T> a, x0, x1, o;
            a.load(qb + c);
            x0.load(p0 + c);
            x1.load(p1 + c);
            if (dq) {
#pragma unroll
                for (int k = 0; k < V; ++k) o.v[k] = s0 * x0.v[k] + s1 * x1.v[k];
                o.store(dq + b * d + c);
            }
            if (dp) {
#pragma unroll
                for (int k = 0; k < V; ++k) o.v[k] = s0 * a.v[k];
                o.store(dp + (2 * b) * d + c);
#pragma unroll
                for (int k = 0; k < V; ++k) o.v[k] = s1 * a.v[k];
                o.store(dp + (2 * b + 1) * d + c);
            }
        }
    } else {
        for (int64_t c = threadIdx.x; c < d; c += kDotThreads) {
            const float a = Elem<T>::ld(qb + c), x0 = Elem<T>::ld(p0 + c), x1 = Elem<T>::ld(p1 + c);
            if (dq) Elem<T>::st(dq + b * d + c, s0 * x0 + s1 * x1);
            if (dp) {
                Elem<T>::st(dp + (2 * b) * d + c, s0 * a);
                Elem<T>::st(dp + (2 * b + 1) * d + c, s1 * a);
            }
        }
    }
}

}  // namespace

// Shared with infonce.hip (RPO_TARGET_FIRST mode): raw grouped dot products.
int rpo_launch_grouped_dots(const void* q, const void* p, int64_t B, int64_t G, int64_t d, int dtype, float* out,
                            hipStream_t st) {
    if (dtype == RPO_DT_F32)
        RPO_LAUNCH(grouped_dots_kernel<float>, dim3((unsigned)B), dim3(kDotThreads), 0, st, (const float*)q,
                           (const float*)p, G, d, out);
    else if (dtype == RPO_DT_BF16)
        RPO_LAUNCH(grouped_dots_kernel<bf16_t>, dim3((unsigned)B), dim3(kDotThreads), 0, st,
                           (const bf16_t*)q, (const bf16_t*)p, G, d, out);
    else if (dtype == RPO_DT_F16)
        RPO_LAUNCH(grouped_dots_kernel<f16_t>, dim3((unsigned)B), dim3(kDotThreads), 0, st,
                           (const f16_t*)q, (const f16_t*)p, G, d, out);
    else
        return RPO_ERR_INVALID_ARG;
    return rpo_launch_status();
}

extern "C" int rpo_rankpo_fwd(const void* q, const void* p, const float* ref_chosen, const float* ref_rejected,
                              int64_t B, int64_t d, int dtype, const rpo_rankpo_params* params, float* scores_out,
                              float* losses_out, float* loss_out, float* metrics_out, float* dscores_out,
                              rpo_stream_t stream) {
    if (!q || !p || !params || !scores_out || !losses_out || !loss_out || !metrics_out || !dscores_out)
        return RPO_ERR_INVALID_ARG;
    if (B <= 0 || d <= 0 || B > INT32_MAX) return RPO_ERR_INVALID_ARG;
    if (!rpo_dtype_ok(dtype)) return RPO_ERR_INVALID_ARG;
    if (params->loss_type != RPO_LOSS_SIGMOID && params->loss_type != RPO_LOSS_HINGE) return RPO_ERR_INVALID_ARG;
    if (!(params->temperature > 0.f)) return RPO_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    int rc = rpo_launch_grouped_dots(q, p, B, 2, d, dtype, scores_out, st);
    if (rc != RPO_OK) return rc;
    RPO_LAUNCH(rankpo_finalize_kernel, dim3(1), dim3(kFinThreads), 0, st, scores_out, ref_chosen,
                       ref_rejected, B, *params, losses_out, loss_out, metrics_out, dscores_out);
    return rpo_launch_status();
}

extern "C" int rpo_rankpo_bwd(const void* q, const void* p, const float* dscores, const float* grad_loss, int64_t B,
                              int64_t d, int dtype, void* dq_out, void* dp_out, rpo_stream_t stream) {
    if (!q || !p || !dscores || !grad_loss || (!dq_out && !dp_out)) return RPO_ERR_INVALID_ARG;
    if (B <= 0 || d <= 0 || B > INT32_MAX) return RPO_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == RPO_DT_F32)
        RPO_LAUNCH(rankpo_bwd_kernel<float>, dim3((unsigned)B), dim3(kDotThreads), 0, st, (const float*)q,
                           (const float*)p, dscores, grad_loss, d, (float*)dq_out, (float*)dp_out);
    else if (dtype == RPO_DT_BF16)
        RPO_LAUNCH(rankpo_bwd_kernel<bf16_t>, dim3((unsigned)B), dim3(kDotThreads), 0, st, (const bf16_t*)q,
                           (const bf16_t*)p, dscores, grad_loss, d, (bf16_t*)dq_out, (bf16_t*)dp_out);
    else if (dtype == RPO_DT_F16)
        RPO_LAUNCH(rankpo_bwd_kernel<f16_t>, dim3((unsigned)B), dim3(kDotThreads), 0, st, (const f16_t*)q,
                           (const f16_t*)p, dscores, grad_loss, d, (f16_t*)dq_out, (f16_t*)dp_out);
    else
        return RPO_ERR_INVALID_ARG;
    return rpo_launch_status();
}
