// Causal variable-length flash attention for head_dim 64 (bf16), grouped-query heads: forward.
// Encoder-side kernel (the encoder's GEMMs stay under PyTorch-ROCm); replaces the AOTriton varlen kernel the packed
// encoder path used (profiles/r01d: attn_fwd ~ 250 TFLOP/s at head_dim 64).
//
// Everything is computed TRANSPOSED so that a query lives on a lane (lane & 15) through the whole pipeline:
//   S^T = K Q^T      A = K tile rows (ds_read_b128, XOR-swizzled), B = Q fragments held in registers
//   softmax over keys = over the accumulator registers of a lane + two shfl_xor (16, 32); running max / sum per lane
//   O^T = V^T P^T    B = P^T: the S^T accumulators of two 16-key tiles, converted to bf16, ARE the B fragment
//                    (k-slot j of lane group g = key 4g + j of the first tile, 16 + 4g + (j - 4) of the second);
//                    A = V^T read from the row-major V tile with ds_read_b64_tr_b16 in the SAME key order.
// Block = 4 waves x 32 queries = 128 queries of one (sequence, head); key tiles of 64; K/V tiles staged through
// registers (global loads for tile t+1 are issued before the MFMAs of tile t, written to LDS after them).
#include "common.hpp"

namespace {

constexpr int kFaThreads = 256, kFaBM = 128, kFaBN = 64, kFaHD = 64;

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

__device__ __forceinline__ u32x2 lds_tr_read(unsigned addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    return (unsigned)f32_to_bf16(a) | ((unsigned)f32_to_bf16(b) << 16);
}

// tiles: int32 [ntiles][2] = (sequence id, first query row inside the sequence), heaviest tiles first.
__global__ __launch_bounds__(kFaThreads, 2) void fa_fwd_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, int64_t sq, int64_t sk,
    int64_t sv, const int* __restrict__ cu, const int* __restrict__ tiles, int nh, int nkv, float scale_log2e,
    float scale, bf16_t* __restrict__ o, int64_t so, float* __restrict__ lse, int64_t lse_seq_stride,
    int64_t lse_head_stride, int lse_packed) {
    __shared__ __attribute__((aligned(16))) char smem[2 * kFaBN * 128];   // K tile | V tile, 128-byte rows
    char* Ks = smem;
    char* Vs = smem + kFaBN * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, fr = lane & 15;
    const int seq = tiles[2 * blockIdx.x], q0 = tiles[2 * blockIdx.x + 1];
    const int h = blockIdx.y, hk = h / (nh / nkv);
    const int64_t t0 = cu[seq];
    const int len = cu[seq + 1] - (int)t0;
    const int qw = q0 + 32 * wave;                       // this wave's first query row (inside the sequence)

    // Q^T fragments (B operand): lane = query fr of tile n, k = hd 32 ks + 8 g .. + 7
    short8_t bq[2][2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int qi = qw + 16 * n + fr;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (qi < len) bq[n][ks] = *reinterpret_cast<const short8_t*>(q + (t0 + qi) * sq + h * kFaHD + 32 * ks + 8 * g);
            else bq[n][ks] = short8_t{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    float4_t oacc[4][2];                                 // O^T: [hd tile c][query tile n], rows = hd 16c + 4g + r
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int n = 0; n < 2; ++n) oacc[c][n] = float4_t{0.f, 0.f, 0.f, 0.f};
    float mrun[2] = {-1e30f, -1e30f}, lrun[2] = {0.f, 0.f};

    const int last_q = min(q0 + kFaBM - 1, len - 1);
    const int nkt = last_q / kFaBN + 1;                  // key tiles 0 .. nkt-1 (causal)
    // staging: thread t moves chunks t and t + 256 of K and of V (64 rows x 8 chunks of 16 bytes each)
    uint4 kreg[2], vreg[2];
    auto stage_load = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int id = tid + 256 * i, row = id >> 3, ch = id & 7;
            const int key = kt * kFaBN + row;
            if (key < len) {
                kreg[i] = *reinterpret_cast<const uint4*>(k + (t0 + key) * sk + hk * kFaHD + ch * 8);
                vreg[i] = *reinterpret_cast<const uint4*>(v + (t0 + key) * sv + hk * kFaHD + ch * 8);
            } else {
                kreg[i] = make_uint4(0, 0, 0, 0);
                vreg[i] = make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto stage_write = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int id = tid + 256 * i, row = id >> 3, ch = id & 7;
            *reinterpret_cast<uint4*>(Ks + row * 128 + ((ch ^ (row & 7)) << 4)) = kreg[i];
            *reinterpret_cast<uint4*>(Vs + row * 128 + ((ch ^ (((row >> 1) & 3) << 1)) << 4)) = vreg[i];
        }
    };
    stage_load(0);
    stage_write();
    __syncthreads();
    const unsigned vs_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)Vs;
    const int qq = fr >> 2, pp = fr & 3;                 // tr-read address roles inside a 16-lane group

    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) stage_load(kt + 1);
        // this wave's queries may all lie before this key tile (upper waves of the last tiles): nothing to do
        const bool active = (kt * kFaBN <= qw + 31) && (qw < len);
        if (active) {
            // ---- S^T = K Q^T
            float4_t s[4][2];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) s[m][n] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int row = 16 * m + fr;
                    const short8_t a = *reinterpret_cast<const short8_t*>(Ks + row * 128 + (((ks * 4 + g) ^ (row & 7)) << 4));
#pragma unroll
                    for (int n = 0; n < 2; ++n) s[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bq[n][ks], s[m][n], 0, 0, 0);
                }
            }
            // ---- causal / length mask (only tiles that touch the diagonal or the sequence end)
            const int kbase = kt * kFaBN + 4 * g;
            const bool need_mask = (kt * kFaBN + kFaBN - 1 > qw) || (kt * kFaBN + kFaBN > len);
            if (need_mask) {
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int qi = qw + 16 * n + fr;
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int key = kbase + 16 * m + r;
                            if (key > qi || key >= len) s[m][n][r] = -1e30f;
                        }
                }
            }
            // ---- online softmax (per query = per lane column), P^T fragments
            short8_t pfrag[2][2];                        // [k-step s (32 keys)][query tile n]
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                float mx = s[0][n][0];
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[m][n][r]);
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float mnew = fmaxf(mrun[n], mx);
                const float alpha = __builtin_amdgcn_exp2f((mrun[n] - mnew) * scale_log2e);
                const float mls = mnew * scale_log2e;
                float sum = 0.f;
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(fmaf(s[m][n][r], scale_log2e, -mls));
                        s[m][n][r] = pv;
                        sum += pv;
                    }
                sum += __shfl_xor(sum, 16, 64);
                sum += __shfl_xor(sum, 32, 64);
                lrun[n] = lrun[n] * alpha + sum;
                mrun[n] = mnew;
#pragma unroll
                for (int c = 0; c < 4; ++c) oacc[c][n] *= alpha;
#pragma unroll
                for (int sI = 0; sI < 2; ++sI) {
                    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
                    u32x4 w;
                    w[0] = pack_bf16(s[2 * sI][n][0], s[2 * sI][n][1]);
                    w[1] = pack_bf16(s[2 * sI][n][2], s[2 * sI][n][3]);
                    w[2] = pack_bf16(s[2 * sI + 1][n][0], s[2 * sI + 1][n][1]);
                    w[3] = pack_bf16(s[2 * sI + 1][n][2], s[2 * sI + 1][n][3]);
                    pfrag[sI][n] = __builtin_bit_cast(short8_t, w);
                }
            }
            // ---- O^T += V^T P^T   (A = V^T via transposed LDS reads, same key order as the P fragments).
            // The 8 transposed reads of a 32-key step and their wait are ONE asm statement: hipcc does not model
            // inline-asm LDS reads, so the MFMAs that consume them must not be schedulable above the wait.
#pragma unroll
            for (int sI = 0; sI < 2; ++sI) {
                unsigned ad[8];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int r0 = 32 * sI + 4 * g + qq, r1 = r0 + 16;
                    const int lc = 2 * c + (pp >> 1);
                    ad[2 * c] = vs_base + r0 * 128 + ((lc ^ (((r0 >> 1) & 3) << 1)) << 4) + 8 * (pp & 1);
                    ad[2 * c + 1] = vs_base + r1 * 128 + ((lc ^ (((r1 >> 1) & 3) << 1)) << 4) + 8 * (pp & 1);
                }
                u32x2 t0v, t1v, t2v, t3v, t4v, t5v, t6v, t7v;
                asm volatile(
                    "ds_read_b64_tr_b16 %0, %8\n\t"
                    "ds_read_b64_tr_b16 %1, %9\n\t"
                    "ds_read_b64_tr_b16 %2, %10\n\t"
                    "ds_read_b64_tr_b16 %3, %11\n\t"
                    "ds_read_b64_tr_b16 %4, %12\n\t"
                    "ds_read_b64_tr_b16 %5, %13\n\t"
                    "ds_read_b64_tr_b16 %6, %14\n\t"
                    "ds_read_b64_tr_b16 %7, %15\n\t"
                    "s_waitcnt lgkmcnt(0)"
                    : "=&v"(t0v), "=&v"(t1v), "=&v"(t2v), "=&v"(t3v), "=&v"(t4v), "=&v"(t5v), "=&v"(t6v), "=&v"(t7v)
                    : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7])
                    : "memory");
                typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
                const u32x4 w0 = {t0v[0], t0v[1], t1v[0], t1v[1]}, w1 = {t2v[0], t2v[1], t3v[0], t3v[1]};
                const u32x4 w2 = {t4v[0], t4v[1], t5v[0], t5v[1]}, w3 = {t6v[0], t6v[1], t7v[0], t7v[1]};
                const short8_t av[4] = {__builtin_bit_cast(short8_t, w0), __builtin_bit_cast(short8_t, w1),
                                        __builtin_bit_cast(short8_t, w2), __builtin_bit_cast(short8_t, w3)};
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        oacc[c][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[c], pfrag[sI][n], oacc[c][n], 0, 0, 0);
            }
        }
        __syncthreads();
        if (kt + 1 < nkt) {
            stage_write();
            __syncthreads();
        }
    }
    // ---- epilogue: O[q][16c + 4g + r] = O^T / l ;  lse = scale m + ln l
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int qi = qw + 16 * n + fr;
        if (qi >= len) continue;
        const float inv = 1.0f / lrun[n];
        bf16_t* orow = o + (t0 + qi) * so + h * kFaHD;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint2 w;
            w.x = pack_bf16(oacc[c][n][0] * inv, oacc[c][n][1] * inv);
            w.y = pack_bf16(oacc[c][n][2] * inv, oacc[c][n][3] * inv);
            *reinterpret_cast<uint2*>(orow + 16 * c + 4 * g) = w;
        }
        if (g == 0)
            lse[(int64_t)seq * lse_seq_stride + (int64_t)h * lse_head_stride + (lse_packed ? t0 : 0) + qi] =
                mrun[n] * scale + logf(lrun[n]);
    }
}

}  // namespace

extern "C" int rpo_flash_attn_fwd(const void* q, const void* k, const void* v, int64_t q_stride, int64_t k_stride,
                                  int64_t v_stride, const int* cu_seqlens, const int* tiles, int64_t ntiles,
                                  int64_t total_tokens, int64_t num_heads, int64_t num_kv_heads, int64_t head_dim,
                                  float scale, void* out, int64_t out_stride, float* lse, int64_t lse_max_len,
                                  rpo_stream_t stream) {
    if (!q || !k || !v || !cu_seqlens || !tiles || !out || !lse || ntiles <= 0 || total_tokens <= 0)
        return RPO_ERR_INVALID_ARG;
    if (head_dim != kFaHD || num_heads <= 0 || num_kv_heads <= 0 || num_heads % num_kv_heads != 0 || num_heads > 65535)
        return RPO_ERR_UNSUPPORTED;
    if (q_stride % 8 || k_stride % 8 || v_stride % 8 || out_stride % 4 || !rpo_aligned16(q) || !rpo_aligned16(k) ||
        !rpo_aligned16(v) || (reinterpret_cast<uintptr_t>(out) & 7))
        return RPO_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const float log2e = 1.4426950408889634f;
    RPO_LAUNCH(fa_fwd_kernel, dim3((unsigned)ntiles, (unsigned)num_heads), dim3(kFaThreads), 0, st, (const bf16_t*)q,
               (const bf16_t*)k, (const bf16_t*)v, q_stride, k_stride, v_stride, cu_seqlens, tiles, (int)num_heads,
               (int)num_kv_heads, scale * log2e, scale, (bf16_t*)out, out_stride, lse,
               lse_max_len > 0 ? num_heads * lse_max_len : 0, lse_max_len > 0 ? lse_max_len : total_tokens,
               lse_max_len > 0 ? 0 : 1);
    return rpo_launch_status();
}
