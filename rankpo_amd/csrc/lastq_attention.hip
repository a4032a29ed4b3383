// One-query-per-sequence attention: the attention of the LAST block of a last-token-pooled encoder (reference: the encoder
// forward behind modeling.py:219 followed by the pooling of modeling.py:224-230 -- only the last real token of every sequence
// is read downstream, so the last block needs its attention output for that ONE query per sequence; rankpo_amd/encoder.py,
// `LlamaLayer.forward_last_rows`).  The query is the last token of its sequence: it sees every key, no mask.
//
// Shape of the problem (cfg 2): 56 sequences x 32 q heads, <= 4096 keys each, 8 kv heads: 0.16 GFLOP against 317 MB of K / V --
// HBM-bound (AI = 0.5 FLOP/B), so the kernels stream K / V exactly once: one block per (sequence, kv head) serves all
// REP = num_heads / num_kv_heads query heads of the group from the same K / V rows, 16-byte loads, HD / 8 lanes per key row,
// keys dealt to the 4 waves x (64 / lanes-per-key) lane groups, U key passes of loads in flight per wave before the first use.
// Softmax is online per lane group (f32, exp2 with the scale folded into q); the groups' partial (m, l, acc) are merged through
// LDS in fixed order: deterministic, no atomics.  The backward makes ONE pass as well: every key row belongs to exactly one lane
// group, which writes its dK / dV rows (P is recomputed from the saved lse), dQ is summed over the groups like the forward's acc.
#include "common.hpp"

namespace {

constexpr int kLqThreads = 256;
constexpr int kLqWaves = 4;
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

__device__ __forceinline__ void unpack8(const uint4& t, float* f) {
    const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(w[i] << 16);
        f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (unsigned)f32_to_bf16(f[2 * i]) | ((unsigned)f32_to_bf16(f[2 * i + 1]) << 16);
    return make_uint4(w[0], w[1], w[2], w[3]);
}
// sum over the LPK lanes that share a key row (consecutive lanes); every lane of the group gets the total
template <int LPK>
__device__ __forceinline__ float group_sum(float x) {
#pragma unroll
    for (int o = 1; o < LPK; o <<= 1) x += __shfl_xor(x, o, 64);
    return x;
}

template <int HD, int REP, int U>
__global__ __launch_bounds__(kLqThreads, 4) void lastq_fwd_kernel(      // 4 blocks per CU = 128 registers: hipcc took 129-130 without
    const bf16_t* __restrict__ q, int64_t q_stride, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, int64_t k_stride,
    int64_t v_stride, const int* __restrict__ cu, int nkv, float scale, bf16_t* __restrict__ out, int64_t out_stride,
    float* __restrict__ lse) {
    constexpr int LPK = HD / 8, KPW = 64 / LPK, G = KPW * kLqWaves;
    __shared__ float red_acc[G][REP][HD];
    __shared__ float red_m[G][REP], red_l[G][REP];
    const int seq = blockIdx.x / nkv, kvh = blockIdx.x - seq * nkv;
    const int t0 = cu[seq], len = cu[seq + 1] - t0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot = lane / LPK, sl = lane - slot * LPK, g = wave * KPW + slot;
    const int nh = nkv * REP;
    float qf[REP][8], m[REP], l[REP], acc[REP][8];
#pragma unroll
    for (int h = 0; h < REP; ++h) {
        const uint4 t = *reinterpret_cast<const uint4*>(q + (int64_t)seq * q_stride + (kvh * REP + h) * HD + sl * 8);
        unpack8(t, qf[h]);
        m[h] = -INFINITY;
        l[h] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            qf[h][e] *= scale * kLog2e;
            acc[h][e] = 0.f;
        }
    }
    const bf16_t* kb = k + (int64_t)t0 * k_stride + kvh * HD + sl * 8;
    const bf16_t* vb = v + (int64_t)t0 * v_stride + kvh * HD + sl * 8;
    for (int base = g; base < len; base += G * U) {
        uint4 kr[U], vr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int key = min(base + u * G, len - 1);       // clamped: a row past the end is loaded but never used
            kr[u] = *reinterpret_cast<const uint4*>(kb + (int64_t)key * k_stride);
            vr[u] = *reinterpret_cast<const uint4*>(vb + (int64_t)key * v_stride);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (base + u * G < len) {                         // uniform over the lanes of a group
                float kf[8], vf[8];
                unpack8(kr[u], kf);
                unpack8(vr[u], vf);
#pragma unroll
                for (int h = 0; h < REP; ++h) {
                    float s = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) s = fmaf(qf[h][e], kf[e], s);
                    s = group_sum<LPK>(s);
                    const float mn = fmaxf(m[h], s);
                    const float alpha = __builtin_amdgcn_exp2f(m[h] - mn);
                    const float p = __builtin_amdgcn_exp2f(s - mn);
                    l[h] = fmaf(l[h], alpha, p);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[h][e] = fmaf(acc[h][e], alpha, p * vf[e]);
                    m[h] = mn;
                }
            }
        }
    }
#pragma unroll
    for (int h = 0; h < REP; ++h) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red_acc[g][h][sl * 8 + e] = acc[h][e];
        if (sl == 0) {
            red_m[g][h] = m[h];
            red_l[g][h] = l[h];
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < REP * HD / 2; idx += kLqThreads) {      // two adjacent output elements per thread
        const int h = idx / (HD / 2), e = (idx - h * (HD / 2)) * 2;
        float M = -INFINITY;
#pragma unroll 4
        for (int j = 0; j < G; ++j) M = fmaxf(M, red_m[j][h]);
        float L = 0.f, o0 = 0.f, o1 = 0.f;
        if (len > 0) {
#pragma unroll 4
            for (int j = 0; j < G; ++j) {                                          // fixed order: deterministic
                const float w = __builtin_amdgcn_exp2f(red_m[j][h] - M);           // a group that saw no key: exp2(-inf) = 0
                L = fmaf(red_l[j][h], w, L);
                o0 = fmaf(red_acc[j][h][e], w, o0);
                o1 = fmaf(red_acc[j][h][e + 1], w, o1);
            }
        }
        const float inv = L > 0.f ? 1.0f / L : 0.f;
        const unsigned pk = (unsigned)f32_to_bf16(o0 * inv) | ((unsigned)f32_to_bf16(o1 * inv) << 16);
        *reinterpret_cast<unsigned*>(out + (int64_t)seq * out_stride + (kvh * REP + h) * HD + e) = pk;
        if (e == 0) lse[(int64_t)seq * nh + kvh * REP + h] = L > 0.f ? (M + log2f(L)) * kLn2 : -INFINITY;
    }
}

template <int HD, int REP, int U>
__global__ __launch_bounds__(kLqThreads) void lastq_bwd_kernel(
    const bf16_t* __restrict__ q, int64_t q_stride, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, int64_t k_stride,
    int64_t v_stride, const int* __restrict__ cu, int nkv, float scale, const bf16_t* __restrict__ out, int64_t out_stride,
    const bf16_t* __restrict__ dout, int64_t dout_stride, const float* __restrict__ lse, bf16_t* __restrict__ dq,
    int64_t dq_stride, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, int64_t dk_stride, int64_t dv_stride) {
    constexpr int LPK = HD / 8, KPW = 64 / LPK, G = KPW * kLqWaves;
    __shared__ float red_acc[G][REP][HD];
    const int seq = blockIdx.x / nkv, kvh = blockIdx.x - seq * nkv;
    const int t0 = cu[seq], len = cu[seq + 1] - t0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot = lane / LPK, sl = lane - slot * LPK, g = wave * KPW + slot;
    const int nh = nkv * REP;
    float qf[REP][8], dof[REP][8], dqa[REP][8], delta[REP], lse2[REP];
#pragma unroll
    for (int h = 0; h < REP; ++h) {
        const int col = (kvh * REP + h) * HD + sl * 8;
        float of[8];
        unpack8(*reinterpret_cast<const uint4*>(q + (int64_t)seq * q_stride + col), qf[h]);
        unpack8(*reinterpret_cast<const uint4*>(dout + (int64_t)seq * dout_stride + col), dof[h]);
        unpack8(*reinterpret_cast<const uint4*>(out + (int64_t)seq * out_stride + col), of);
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            d = fmaf(dof[h][e], of[e], d);
            qf[h][e] *= scale * kLog2e;
            dqa[h][e] = 0.f;
        }
        delta[h] = group_sum<LPK>(d);                        // rowsum(dO o O) of this query head
        lse2[h] = lse[(int64_t)seq * nh + kvh * REP + h] * kLog2e;
    }
    const int64_t col0 = kvh * HD + sl * 8;
    const bf16_t* kb = k + (int64_t)t0 * k_stride + col0;
    const bf16_t* vb = v + (int64_t)t0 * v_stride + col0;
    bf16_t* dkb = dk + (int64_t)t0 * dk_stride + col0;
    bf16_t* dvb = dv + (int64_t)t0 * dv_stride + col0;
    for (int base = g; base < len; base += G * U) {
        uint4 kr[U], vr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int key = min(base + u * G, len - 1);
            kr[u] = *reinterpret_cast<const uint4*>(kb + (int64_t)key * k_stride);
            vr[u] = *reinterpret_cast<const uint4*>(vb + (int64_t)key * v_stride);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int key = base + u * G;
            if (key < len) {
                float kf[8], vf[8], dkf[8], dvf[8];
                unpack8(kr[u], kf);
                unpack8(vr[u], vf);
#pragma unroll
                for (int e = 0; e < 8; ++e) dkf[e] = dvf[e] = 0.f;
#pragma unroll
                for (int h = 0; h < REP; ++h) {
                    float s = 0.f, dp = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        s = fmaf(qf[h][e], kf[e], s);
                        dp = fmaf(dof[h][e], vf[e], dp);
                    }
                    s = group_sum<LPK>(s);
                    dp = group_sum<LPK>(dp);
                    const float p = __builtin_amdgcn_exp2f(s - lse2[h]);
                    const float ds = p * (dp - delta[h]);             // d loss / d (scale q.k)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        dvf[e] = fmaf(p, dof[h][e], dvf[e]);
                        dkf[e] = fmaf(ds, qf[h][e], dkf[e]);           // qf = q scale log2e: the log2e is divided out below
                        dqa[h][e] = fmaf(ds, kf[e], dqa[h][e]);
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) dkf[e] *= kLn2;
                *reinterpret_cast<uint4*>(dkb + (int64_t)key * dk_stride) = pack8(dkf);
                *reinterpret_cast<uint4*>(dvb + (int64_t)key * dv_stride) = pack8(dvf);
            }
        }
    }
#pragma unroll
    for (int h = 0; h < REP; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) red_acc[g][h][sl * 8 + e] = dqa[h][e];
    __syncthreads();
    for (int idx = threadIdx.x; idx < REP * HD / 2; idx += kLqThreads) {
        const int h = idx / (HD / 2), e = (idx - h * (HD / 2)) * 2;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll 4
        for (int j = 0; j < G; ++j) {
            a0 += red_acc[j][h][e];
            a1 += red_acc[j][h][e + 1];
        }
        const unsigned pk = (unsigned)f32_to_bf16(a0 * scale) | ((unsigned)f32_to_bf16(a1 * scale) << 16);
        *reinterpret_cast<unsigned*>(dq + (int64_t)seq * dq_stride + (kvh * REP + h) * HD + e) = pk;
    }
}

bool lastq_args_ok(int64_t N, int64_t nh, int64_t nkv, int64_t hd) {
    if (N <= 0 || nh <= 0 || nkv <= 0 || nh % nkv != 0 || N * nkv > 0x7fffffff) return false;
    return true;
}
bool lastq_supported(int64_t nh, int64_t nkv, int64_t hd) {
    const int64_t rep = nh / nkv;
    return (hd == 64 || hd == 128) && (rep == 1 || rep == 2 || rep == 4);
}

}  // namespace

#define RPO_LQ_DISPATCH(KERNEL, U64, U128, ...)                                                                            \
    do {                                                                                                                    \
        const int rep = (int)(num_heads / num_kv_heads);                                                                    \
        const dim3 grid((unsigned)(num_seqs * num_kv_heads)), block(kLqThreads);                                            \
        hipStream_t st = (hipStream_t)stream;                                                                               \
        if (head_dim == 64) {                                                                                               \
            if (rep == 1) RPO_LAUNCH((KERNEL<64, 1, U64>), grid, block, 0, st, __VA_ARGS__);                                \
            else if (rep == 2) RPO_LAUNCH((KERNEL<64, 2, U64>), grid, block, 0, st, __VA_ARGS__);                           \
            else RPO_LAUNCH((KERNEL<64, 4, U64>), grid, block, 0, st, __VA_ARGS__);                                         \
        } else {                                                                                                            \
            if (rep == 1) RPO_LAUNCH((KERNEL<128, 1, U128>), grid, block, 0, st, __VA_ARGS__);                              \
            else if (rep == 2) RPO_LAUNCH((KERNEL<128, 2, U128>), grid, block, 0, st, __VA_ARGS__);                         \
            else RPO_LAUNCH((KERNEL<128, 4, U128>), grid, block, 0, st, __VA_ARGS__);                                       \
        }                                                                                                                   \
    } while (0)

extern "C" int rpo_lastq_attn_fwd(const void* q, int64_t q_stride, const void* k, const void* v, int64_t k_stride,
                                  int64_t v_stride, const int* cu_seqlens, int64_t num_seqs, int64_t num_heads,
                                  int64_t num_kv_heads, int64_t head_dim, float scale, void* out, int64_t out_stride,
                                  float* lse, rpo_stream_t stream) {
    if (!q || !k || !v || !cu_seqlens || !out || !lse || !lastq_args_ok(num_seqs, num_heads, num_kv_heads, head_dim))
        return RPO_ERR_INVALID_ARG;
    if (!lastq_supported(num_heads, num_kv_heads, head_dim)) return RPO_ERR_UNSUPPORTED;
    if (!rpo_aligned16(q) || !rpo_aligned16(k) || !rpo_aligned16(v) || !rpo_aligned16(out) || q_stride % 8 || k_stride % 8 ||
        v_stride % 8 || out_stride % 8)
        return RPO_ERR_UNSUPPORTED;
    RPO_LQ_DISPATCH(lastq_fwd_kernel, 4, 4, (const bf16_t*)q, q_stride, (const bf16_t*)k, (const bf16_t*)v, k_stride, v_stride,
                    cu_seqlens, (int)num_kv_heads, scale, (bf16_t*)out, out_stride, lse);
    return rpo_launch_status();
}

extern "C" int rpo_lastq_attn_bwd(const void* q, int64_t q_stride, const void* k, const void* v, int64_t k_stride,
                                  int64_t v_stride, const int* cu_seqlens, int64_t num_seqs, int64_t num_heads,
                                  int64_t num_kv_heads, int64_t head_dim, float scale, const void* out, int64_t out_stride,
                                  const void* dout, int64_t dout_stride, const float* lse, void* dq, int64_t dq_stride, void* dk,
                                  void* dv, int64_t dk_stride, int64_t dv_stride, rpo_stream_t stream) {
    if (!q || !k || !v || !cu_seqlens || !out || !dout || !lse || !dq || !dk || !dv ||
        !lastq_args_ok(num_seqs, num_heads, num_kv_heads, head_dim))
        return RPO_ERR_INVALID_ARG;
    if (!lastq_supported(num_heads, num_kv_heads, head_dim)) return RPO_ERR_UNSUPPORTED;
    if (!rpo_aligned16(q) || !rpo_aligned16(k) || !rpo_aligned16(v) || !rpo_aligned16(out) || !rpo_aligned16(dout) ||
        !rpo_aligned16(dq) || !rpo_aligned16(dk) || !rpo_aligned16(dv) || q_stride % 8 || k_stride % 8 || v_stride % 8 ||
        out_stride % 8 || dout_stride % 8 || dq_stride % 8 || dk_stride % 8 || dv_stride % 8)
        return RPO_ERR_UNSUPPORTED;
    RPO_LQ_DISPATCH(lastq_bwd_kernel, 2, 2, (const bf16_t*)q, q_stride, (const bf16_t*)k, (const bf16_t*)v, k_stride, v_stride,
                    cu_seqlens, (int)num_kv_heads, scale, (const bf16_t*)out, out_stride, (const bf16_t*)dout, dout_stride, lse,
                    (bf16_t*)dq, dq_stride, (bf16_t*)dk, (bf16_t*)dv, dk_stride, dv_stride);
    return rpo_launch_status();
}
