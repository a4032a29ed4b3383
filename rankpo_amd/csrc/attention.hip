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
#include <cstdlib>
#include <cstring>

namespace {

constexpr int kFaThreads = 256, kFaBM = 128, kFaBN = 64, kFaHD = 64;

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;


// two f32 -> one dword of two bf16 (round to nearest even) in ONE instruction; written as `f32_to_bf16(a) | f32_to_bf16(b) << 16`
// hipcc converts each value on its own and merges them with a third instruction.
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// two accumulator tiles (rows 4g + r of each) -> one 8 x bf16 operand fragment: k-slots {4g + j, 16 + 4g + (j - 4)}
__device__ __forceinline__ short8_t pack_frag(const float4_t& lo, const float4_t& hi) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    const u32x4 w = {pack_bf16(lo[0], lo[1]), pack_bf16(lo[2], lo[3]), pack_bf16(hi[0], hi[1]), pack_bf16(hi[2], hi[3])};
    return __builtin_bit_cast(short8_t, w);
}

// K / V tiles (64 keys x 128 B each) go global -> LDS by 16-byte global_load_lds into a ring of three (K | V) images
// (chunk ^= row & 7 on the source side: conflict-free for the ds_read_b128 row reads of K and for the transposed
// ds_read_b64_tr_b16 reads of V): tile kt + 2 is issued while tile kt is consumed, ONE raw barrier per tile, counted
// vmcnt.  Keys past the end of the sequence are clamped to its last row (their scores are masked).
constexpr int kKvTile = 2 * kFaBN * 128;                          // 16 KiB

#define RPO_TR4(OUT0, OUT1, OUT2, OUT3, ADDR, OFF0, OFF1, OFF2, OFF3)                                               \
    asm volatile("ds_read_b64_tr_b16 %0, %4 offset:" #OFF0 "\n\tds_read_b64_tr_b16 %1, %4 offset:" #OFF1 "\n\t"      \
                 "ds_read_b64_tr_b16 %2, %4 offset:" #OFF2 "\n\tds_read_b64_tr_b16 %3, %4 offset:" #OFF3             \
                 : "=&v"(OUT0), "=&v"(OUT1), "=&v"(OUT2), "=&v"(OUT3)                                                \
                 : "v"(ADDR)                                                                                         \
                 : "memory")

__device__ __forceinline__ short8_t join_tr(const u32x2& lo, const u32x2& hi) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    const u32x4 w = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(short8_t, w);
}

// tiles: int32 [ntiles][2] = (sequence id, first query row inside the sequence), heaviest tiles first.
__global__ __launch_bounds__(kFaThreads, 2) void fa_fwd_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, int64_t sq, int64_t sk,
    int64_t sv, const int* __restrict__ cu, const int* __restrict__ tiles, int nh, int nkv, float scale_log2e,
    float scale, bf16_t* __restrict__ o, int64_t so, float* __restrict__ lse, int64_t lse_seq_stride,
    int64_t lse_head_stride, int lse_packed) {
    __shared__ __attribute__((aligned(16))) char smem[3 * kKvTile];       // ring of (K tile | V tile), 128-byte rows
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const int last_q = min(q0 + kFaBM - 1, len - 1);
    const int nkt = last_q / kFaBN + 1;                  // key tiles 0 .. nkt-1 (causal)
    // staging: DMA instruction u = 2 * wave + i (i = 0, 1) fills tile rows 8u .. 8u + 7 of K and of V; lane l carries
    // row 8u + (l >> 3), physical chunk l & 7 = logical chunk (l & 7) ^ (row & 7)
    const int srow = lane >> 3, lchunk = (lane & 7) ^ srow;
    const char* ksrc = reinterpret_cast<const char*>(k + t0 * sk + hk * kFaHD);
    const char* vsrc = reinterpret_cast<const char*>(v + t0 * sv + hk * kFaHD);
    const unsigned skb = (unsigned)sk * 2u, svb = (unsigned)sv * 2u;
    auto stage = [&](int kt, int buf) {
        char* base = smem + buf * kKvTile;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int u = 2 * wave + i;
            const unsigned row = (unsigned)min(kt * kFaBN + 8 * u + srow, len - 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ksrc + (row * skb + lchunk * 16)),
                                             (__attribute__((address_space(3))) void*)(base + u * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vsrc + (row * svb + lchunk * 16)),
                                             (__attribute__((address_space(3))) void*)(base + kFaBN * 128 + u * 1024), 16, 0, 0);
        }
    };
    stage(0, 0);
    if (nkt > 1) stage(1, 1);
    // the Q fragments are ordinary loads issued BEFORE the DMAs: touching them here puts hipcc's wait for them in front
    // of the loop (inside it, it would be a vmcnt(0) that makes every tile's DMA synchronous)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(bq[n][ks]));

    float4_t oacc[4][2];                                 // O^T: [hd tile c][query tile n], rows = hd 16c + 4g + r
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int n = 0; n < 2; ++n) oacc[c][n] = float4_t{0.f, 0.f, 0.f, 0.f};
    float mrun[2] = {-1e30f, -1e30f}, lrun[2] = {0.f, 0.f};

    // per-lane offsets inside an image.  K row reads: row 16 m + fr, chunk (4 ks + g) ^ (fr & 7).  V transposed reads:
    // lane (g, qq, pp) addresses row 32 sI + 16 h + 4 g + qq, hd columns 16 c + 4 pp .. + 3 = chunk (2 c) ^ x with
    // x = (pp >> 1) ^ (4 (g & 1) + qq); h and sI are immediate offsets (2048, 4096).
    const int qq = fr >> 2, pp = fr & 3;                 // tr-read address roles inside a 16-lane group
    const int xs = (pp >> 1) ^ (4 * (g & 1) + qq);
    unsigned tr_off[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) tr_off[c] = kFaBN * 128 + (4 * g + qq) * 128 + (((2 * c) ^ xs) << 4) + 8 * (pp & 1);
    unsigned row_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) row_off[ks] = fr * 128 + (((4 * ks + g) ^ (fr & 7)) << 4);
    const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // tile kt landed; tile kt + 1 may fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nkt) stage(kt + 2, cur == 0 ? 2 : cur - 1);               // the buffer read in iteration kt - 1
        // this wave's queries may all lie before this key tile (upper waves of the last tiles): nothing to do
        const bool active = (kt * kFaBN <= qw + 31) && (qw < len);
        if (active) {
            const char* Ks = smem + cur * kKvTile;
            const unsigned tb = smem_base + cur * kKvTile;
            // V^T fragments (A operands of O^T += V^T P^T): issued now, consumed after the softmax arithmetic
            u32x2 x0, x1, x2, x3, x4, x5, x6, x7, y0, y1, y2, y3, y4, y5, y6, y7;
            {
                const unsigned a0 = tb + tr_off[0], a1 = tb + tr_off[1], a2 = tb + tr_off[2], a3 = tb + tr_off[3];
                RPO_TR4(x0, x1, y0, y1, a0, 0, 2048, 4096, 6144);
                RPO_TR4(x2, x3, y2, y3, a1, 0, 2048, 4096, 6144);
                RPO_TR4(x4, x5, y4, y5, a2, 0, 2048, 4096, 6144);
                RPO_TR4(x6, x7, y6, y7, a3, 0, 2048, 4096, 6144);
            }
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
                    const short8_t a = *reinterpret_cast<const short8_t*>(Ks + row_off[ks] + m * 2048);
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
            float mnew[2];
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                float mx[4];
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    mx[m] = fmaxf(fmaxf(s[m][n][0], s[m][n][1]), fmaxf(s[m][n][2], s[m][n][3]));
                float mm = fmaxf(fmaxf(mx[0], mx[1]), fmaxf(mx[2], mx[3]));
                mm = fmaxf(mm, __shfl_xor(mm, 16, 64));
                mm = fmaxf(mm, __shfl_xor(mm, 32, 64));
                mnew[n] = fmaxf(mrun[n], mm);
            }
            // the running maximum rarely moves after the first tiles: skip the rescale of O (32 multiplies) when no lane's did
            if (__builtin_amdgcn_ballot_w64(mnew[0] != mrun[0] || mnew[1] != mrun[1]) != 0) {
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const float alpha = __builtin_amdgcn_exp2f((mrun[n] - mnew[n]) * scale_log2e);
                    lrun[n] *= alpha;
                    mrun[n] = mnew[n];
#pragma unroll
                    for (int c = 0; c < 4; ++c) oacc[c][n] *= alpha;
                }
            }
            short8_t pfrag[2][2];                        // [k-step s (32 keys)][query tile n]
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const float mls = mnew[n] * scale_log2e;
                float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(fmaf(s[m][n][r], scale_log2e, -mls));
                        s[m][n][r] = pv;
                        sum[r] += pv;
                    }
                float st = (sum[0] + sum[1]) + (sum[2] + sum[3]);
                st += __shfl_xor(st, 16, 64);
                st += __shfl_xor(st, 32, 64);
                lrun[n] += st;
                pfrag[0][n] = pack_frag(s[0][n], s[1][n]);
                pfrag[1][n] = pack_frag(s[2][n], s[3][n]);
            }
            // ---- O^T += V^T P^T   (A = V^T via the transposed LDS reads, same key order as the P fragments).
            // hipcc does not model inline-asm LDS reads: naming every destination in the wait keeps the MFMAs below it.
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(y0),
                           "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7)
                         :
                         : "memory");
            const short8_t vt0[4] = {join_tr(x0, x1), join_tr(x2, x3), join_tr(x4, x5), join_tr(x6, x7)};
            const short8_t vt1[4] = {join_tr(y0, y1), join_tr(y2, y3), join_tr(y4, y5), join_tr(y6, y7)};
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    oacc[c][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt0[c], pfrag[0][n], oacc[c][n], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    oacc[c][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt1[c], pfrag[1][n], oacc[c][n], 0, 0, 0);
        }
        cur = cur == 2 ? 0 : cur + 1;
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

// ------------------------------------------------------------------------------------------------------------------
// Backward.  Three launches, no atomics, deterministic:
//   fa_delta_kernel      delta[h][t] = sum_d dO[t,h,d] O[t,h,d]
//   fa_bwd_dq_kernel     block = 128 queries of one (sequence, head), loop over key tiles <= diagonal:
//                        S^T = K Q^T, dP^T = V dO^T, dS^T = P (dP - delta) scale, dQ^T += K^T dS^T
//                        (query on the lane: lse / delta are per-lane scalars; dS^T accumulators are the B fragments)
//   fa_bwd_dkdv_kernel   block = 64 keys of one (sequence, kv head), loop over the q heads of the group and the query
//                        tiles >= the key tile: S = Q K^T, dP = dO V^T (key on the lane, K / V fragments stay in
//                        registers), dV^T += dO^T P, dK^T += Q^T dS with Q^T / dO^T read transposed from the same LDS
//                        images that serve the row reads (chunk ^= row & 7 is conflict-free for both kinds of read).
// P is recomputed from the saved lse: P = exp(scale s - lse).
// ------------------------------------------------------------------------------------------------------------------
// Writes the two per-(head, query) row constants of the backward, NEGATED so that they can be the initial accumulators of the
// S and dP MFMA chains:  nd[h][t] = -delta = -sum_d dO O   and   nl[h][t] = -lse / scale, so that S' = Q K^T - lse / scale
// gives p = exp2(scale log2(e) S') with no subtraction, and dP' = dO V^T - delta is dS / (p scale) as it leaves the chain.
__global__ __launch_bounds__(256) void fa_delta_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout,
                                                       int64_t so, int64_t sdo, int nh, int64_t T,
                                                       const float* __restrict__ lse, float inv_scale,
                                                       float* __restrict__ nd, float* __restrict__ nl) {
    // one wave per (token, 8 heads): lane = (head in group of 8, 16-byte chunk)
    const int64_t t = blockIdx.x;
    for (int hc = threadIdx.x; hc < nh * 8; hc += 256) {
        const int h = hc >> 3, ch = hc & 7;
        Vec16<bf16_t> a, b;
        a.load(o + t * so + h * kFaHD + ch * 8);
        b.load(dout + t * sdo + h * kFaHD + ch * 8);
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = fmaf(a.v[e], b.v[e], acc);
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        if (ch == 0) {
            nd[(int64_t)h * T + t] = -acc;
            nl[(int64_t)h * T + t] = -lse[(int64_t)h * T + t] * inv_scale;
        }
    }
}



// K / V tiles: the same LDS-DMA ring as the forward kernel.
constexpr int kDqTile = kKvTile;

__global__ __launch_bounds__(kFaThreads, 2) void fa_bwd_dq_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const bf16_t* __restrict__ dout, int64_t sq, int64_t sk, int64_t sv, int64_t sdo, const int* __restrict__ cu,
    const int* __restrict__ tiles, int nh, int nkv, float scale_log2e, float scale, const float* __restrict__ lse,
    const float* __restrict__ delta, int64_t T, bf16_t* __restrict__ dq, int64_t sdq) {
    __shared__ __attribute__((aligned(16))) char smem[3 * kDqTile];      // ring of (K tile | V tile), chunk ^= row & 7
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, fr = lane & 15;
    const int seq = tiles[2 * blockIdx.x], q0 = tiles[2 * blockIdx.x + 1];
    const int h = blockIdx.y, hk = h / (nh / nkv);
    const int64_t t0 = cu[seq];
    const int len = cu[seq + 1] - (int)t0;
    const int qw = q0 + 32 * wave;
    short8_t bq[2][2], bdo[2][2];
    float lq[2], dl[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int qi = qw + 16 * n + fr;
        const bool ok = qi < len;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bq[n][ks] = ok ? *reinterpret_cast<const short8_t*>(q + (t0 + qi) * sq + h * kFaHD + 32 * ks + 8 * g)
                           : short8_t{0, 0, 0, 0, 0, 0, 0, 0};
            bdo[n][ks] = ok ? *reinterpret_cast<const short8_t*>(dout + (t0 + qi) * sdo + h * kFaHD + 32 * ks + 8 * g)
                            : short8_t{0, 0, 0, 0, 0, 0, 0, 0};
        }
        lq[n] = ok ? lse[(int64_t)h * T + t0 + qi] : 0.f;      // -lse / scale (fa_delta_kernel)
        dl[n] = ok ? delta[(int64_t)h * T + t0 + qi] : 0.f;    // -delta
    }
    const int last_q = min(q0 + kFaBM - 1, len - 1);
    const int nkt = last_q / kFaBN + 1;
    // staging: DMA instruction u = 2 * wave + i (i = 0, 1) fills tile rows 8u .. 8u + 7 of K and of V; lane l carries
    // row 8u + (l >> 3), physical chunk l & 7 = logical chunk (l & 7) ^ (row & 7)
    const int srow = lane >> 3, lchunk = (lane & 7) ^ srow;
    const char* ksrc = reinterpret_cast<const char*>(k + t0 * sk + hk * kFaHD);
    const char* vsrc = reinterpret_cast<const char*>(v + t0 * sv + hk * kFaHD);
    const unsigned skb = (unsigned)sk * 2u, svb = (unsigned)sv * 2u;
    auto stage = [&](int kt, int buf) {
        char* base = smem + buf * kDqTile;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int u = 2 * wave + i;
            const unsigned row = (unsigned)min(kt * kFaBN + 8 * u + srow, len - 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ksrc + (row * skb + lchunk * 16)),
                                             (__attribute__((address_space(3))) void*)(base + u * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vsrc + (row * svb + lchunk * 16)),
                                             (__attribute__((address_space(3))) void*)(base + kFaBN * 128 + u * 1024), 16, 0, 0);
        }
    };
    stage(0, 0);
    if (nkt > 1) stage(1, 1);
    // the ordinary loads above were issued BEFORE the DMAs: touching their results here puts hipcc's wait for them in
    // front of the loop (inside it, it would be a vmcnt(0) that makes every tile's DMA synchronous)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(bq[n][ks]), "+v"(bdo[n][ks]));
        asm volatile("" : "+v"(lq[n]), "+v"(dl[n]));
    }
    float4_t acc[4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[c][n] = float4_t{0.f, 0.f, 0.f, 0.f};
    // per-lane offsets inside a K (or V) image.  Row reads: row 16 m + fr, chunk (4 ks + g) ^ (fr & 7).  Transposed reads:
    // lane (g, qq, pp) addresses row 32 sI + 16 h + 4 g + qq, hd columns 16 c + 4 pp .. + 3 = chunk (2 c) ^ x with
    // x = (pp >> 1) ^ (4 (g & 1) + qq); h and sI are immediate offsets (2048, 4096).
    const int qq = fr >> 2, pp = fr & 3;
    const int xs = (pp >> 1) ^ (4 * (g & 1) + qq);
    unsigned tr_off[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) tr_off[c] = (4 * g + qq) * 128 + (((2 * c) ^ xs) << 4) + 8 * (pp & 1);
    unsigned row_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) row_off[ks] = fr * 128 + (((4 * ks + g) ^ (fr & 7)) << 4);
    const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // tile kt landed; tile kt + 1 may fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nkt) stage(kt + 2, cur == 0 ? 2 : cur - 1);               // the buffer read in iteration kt - 1
        const bool active = (kt * kFaBN <= qw + 31) && (qw < len);
        if (active) {
            const char* Ks = smem + cur * kDqTile;
            const char* Vs = Ks + kFaBN * 128;
            const unsigned tb = smem_base + cur * kDqTile;
            // K^T fragments (A operands of dQ^T += K^T dS^T): issued now, consumed after the softmax arithmetic
            u32x2 x0, x1, x2, x3, x4, x5, x6, x7, y0, y1, y2, y3, y4, y5, y6, y7;
            {
                const unsigned a0 = tb + tr_off[0], a1 = tb + tr_off[1], a2 = tb + tr_off[2], a3 = tb + tr_off[3];
                RPO_TR4(x0, x1, y0, y1, a0, 0, 2048, 4096, 6144);
                RPO_TR4(x2, x3, y2, y3, a1, 0, 2048, 4096, 6144);
                RPO_TR4(x4, x5, y4, y5, a2, 0, 2048, 4096, 6144);
                RPO_TR4(x6, x7, y6, y7, a3, 0, 2048, 4096, 6144);
            }
            // the row constants (query = lane column: one scalar per lane and query tile) are the initial accumulators
            float4_t s[4][2], dp[4][2];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    s[m][n] = float4_t{lq[n], lq[n], lq[n], lq[n]};
                    dp[m][n] = float4_t{dl[n], dl[n], dl[n], dl[n]};
                }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const short8_t ak = *reinterpret_cast<const short8_t*>(Ks + row_off[ks] + m * 2048);
                    const short8_t av = *reinterpret_cast<const short8_t*>(Vs + row_off[ks] + m * 2048);
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        s[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ak, bq[n][ks], s[m][n], 0, 0, 0);
                        dp[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bdo[n][ks], dp[m][n], 0, 0, 0);
                    }
                }
            }
            const int kbase = kt * kFaBN + 4 * g;
            const bool need_mask = (kt * kFaBN + kFaBN - 1 > qw) || (kt * kFaBN + kFaBN > len);
            short8_t dsf[2][2];
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int qi = qw + 16 * n + fr;
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float pv = __builtin_amdgcn_exp2f(s[m][n][r] * scale_log2e);
                        if (need_mask) {
                            const int key = kbase + 16 * m + r;
                            if (key > qi || key >= len || qi >= len) pv = 0.f;
                        }
                        s[m][n][r] = pv * dp[m][n][r];                    // dS / scale (scale: epilogue)
                    }
                dsf[0][n] = pack_frag(s[0][n], s[1][n]);
                dsf[1][n] = pack_frag(s[2][n], s[3][n]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(y0),
                           "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7)
                         :
                         : "memory");
            const short8_t kt0[4] = {join_tr(x0, x1), join_tr(x2, x3), join_tr(x4, x5), join_tr(x6, x7)};
            const short8_t kt1[4] = {join_tr(y0, y1), join_tr(y2, y3), join_tr(y4, y5), join_tr(y6, y7)};
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[c][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt0[c], dsf[0][n], acc[c][n], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[c][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt1[c], dsf[1][n], acc[c][n], 0, 0, 0);
        }
        cur = cur == 2 ? 0 : cur + 1;
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int qi = qw + 16 * n + fr;
        if (qi >= len) continue;
        bf16_t* row = dq + (t0 + qi) * sdq + h * kFaHD;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint2 w;
            w.x = pack_bf16(acc[c][n][0] * scale, acc[c][n][1] * scale);
            w.y = pack_bf16(acc[c][n][2] * scale, acc[c][n][3] * scale);
            *reinterpret_cast<uint2*>(row + 16 * c + 4 * g) = w;
        }
    }
}

constexpr int kFaDkdvThreads = 512;

// ---- dK / dV ---------------------------------------------------------------------------------------------------------
// Block = 8 waves = 2 key halves (32 keys each) x 4 query groups (32 queries each): 64 keys per block; the Q / dO tile of
// an iteration (128 queries of one q head) is shared by all 8 waves.
// ktiles: int32 [n][3] = (sequence id, kv head, first key of a 64-key tile), sorted by (sequence, head, key): all key
// tiles of one (sequence, kv head) re-read the same Q / dO rows, so they should run at the same time on ONE XCD (shared
// L2; measured 94 % L2 hit rate, HBM traffic = 1.2 x the unique bytes).  Blocks b, b + 8, ... share an XCD: block b
// takes entry (b % 8) * ceil(n / 8) + b / 8, i.e. every XCD walks its own contiguous eighth of the table.
// The Q / dO tiles go global -> LDS by 16-byte global_load_lds (no VGPR staging, no ds_write: a register-staged version
// spent ~415 LDS-store cycles per 1024 MFMA cycles on ds_write_b128 and needed two barriers per tile), three tiles
// deep: tile it + 2 is issued while tile it is consumed, ONE raw barrier per tile, counted vmcnt.
// LDS image per tile: Q 128 rows x 128 B | dO 128 x 128 B | lse 128 f32 | delta 128 f32 (33 KiB); rows are unpadded
// with the 16-byte chunk index XORed by (row & 7) on the SOURCE side (the DMA destination is lane-linear), which keeps
// both the ds_read_b128 row reads and the ds_read_b64_tr_b16 transposed reads conflict-free.
// Rows past the end of the sequence are clamped to its last row: their P and dS are masked to exactly 0.
constexpr int kDmaTile = 2 * kFaBM * 128 + 2 * kFaBM * 4;        // 33792 B
constexpr int kDmaLds = 3 * kDmaTile;                             // 101376 B (one block per CU)

#define RPO_TR2(OUT0, OUT1, ADDR, OFF0, OFF1)                                                                       \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:" #OFF0 "\n\tds_read_b64_tr_b16 %1, %2 offset:" #OFF1             \
                 : "=&v"(OUT0), "=&v"(OUT1)                                                                          \
                 : "v"(ADDR)                                                                                         \
                 : "memory")

#ifdef RPO_FA_STAMP   // diagnostic build only (tools/exp): in-kernel cycle stamps of the dK/dV loop, summed per wave role
__device__ unsigned long long g_fa_stamp[8 * 8];
#define RPO_STAMP(VAR)                                                                     \
    do {                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(VAR)::"memory");        \
        __builtin_amdgcn_sched_barrier(0);                                                 \
    } while (0)
#define RPO_STAMP_ADD(I, A, B) st_acc[I] += (B) - (A)
#else
#define RPO_STAMP(VAR)
#define RPO_STAMP_ADD(I, A, B)
#endif

__global__ __launch_bounds__(kFaDkdvThreads, 1) void fa_bwd_dkdv_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const bf16_t* __restrict__ dout, int64_t sq, int64_t sk, int64_t sv, int64_t sdo, const int* __restrict__ cu,
    const int* __restrict__ ktiles, int nh, int nkv, float scale_log2e, float scale, const float* __restrict__ lse,
    const float* __restrict__ delta, int64_t T, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, int64_t sdk,
    int64_t sdv, int n_ktiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave8 >> 2, wave = wave8 & 3;            // key half, query group
    const int g = lane >> 4, fr = lane & 15;
    const int per = (n_ktiles + 7) >> 3;
    const int entry = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (entry >= n_ktiles) return;
    const int seq = ktiles[3 * entry], hk = ktiles[3 * entry + 1], kb0 = ktiles[3 * entry + 2];
    const int k0 = kb0 + 32 * kh;                          // this wave's 32 keys
    const int group = nh / nkv;
    const int64_t t0 = cu[seq];
    const int len = cu[seq + 1] - (int)t0;
    const int qt0 = (kb0 / kFaBM) * kFaBM;                 // first query tile that can see the block's keys
    const int nqt = (len - qt0 + kFaBM - 1) / kFaBM;
    const int niter = nqt * group;

    // staging: DMA instruction u = 2 * wave8 + i (i = 0, 1) fills tile rows 8u .. 8u + 7 of Q and of dO;
    // lane l carries row 8u + (l >> 3), physical chunk l & 7 = logical chunk (l & 7) ^ (row & 7).
    const int srow = lane >> 3, lchunk = (lane & 7) ^ srow;
    const int lrow = 64 * (wave8 & 1) + lane;              // lse / delta row of this lane (waves 4..7 repeat 0..3)
    const float* ld_src = ((wave8 >> 1) & 1) ? delta : lse;   // -delta : -lse / scale (fa_delta_kernel)
    // addresses = wave-uniform base of (sequence, q head) + a 32-bit lane offset inside the sequence (< 2^31: a sequence
    // is at most 2^31 / row-stride-bytes rows, checked by the host wrapper)
    const unsigned sqb = (unsigned)sq * 2u, sdob = (unsigned)sdo * 2u;
    auto stage = [&](int it, int buf) {
        const int hq = hk * group + it / nqt, qb = qt0 + (it % nqt) * kFaBM;
        char* base = smem + buf * kDmaTile;
        const char* qsrc = reinterpret_cast<const char*>(q + t0 * sq + hq * kFaHD);
        const char* dsrc = reinterpret_cast<const char*>(dout + t0 * sdo + hq * kFaHD);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int u = 2 * wave8 + i;
            const unsigned row = (unsigned)min(qb + 8 * u + srow, len - 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qsrc + (row * sqb + lchunk * 16)),
                                             (__attribute__((address_space(3))) void*)(base + u * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dsrc + (row * sdob + lchunk * 16)),
                                             (__attribute__((address_space(3))) void*)(base + kFaBM * 128 + u * 1024), 16, 0, 0);
        }
        const float* lsrc = ld_src + (int64_t)hq * T + t0;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(lsrc + (unsigned)min(qb + lrow, len - 1)),
                                         (__attribute__((address_space(3))) void*)(base + 2 * kFaBM * 128 + ((wave8 >> 1) & 1) * 512 + (wave8 & 1) * 256),
                                         4, 0, 0);
    };
    short8_t bk[2][2], bv[2][2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int key = k0 + 16 * n + fr;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bk[n][ks] = key < len ? *reinterpret_cast<const short8_t*>(k + (t0 + key) * sk + hk * kFaHD + 32 * ks + 8 * g)
                                  : short8_t{0, 0, 0, 0, 0, 0, 0, 0};
            bv[n][ks] = key < len ? *reinterpret_cast<const short8_t*>(v + (t0 + key) * sv + hk * kFaHD + 32 * ks + 8 * g)
                                  : short8_t{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    stage(0, 0);
    if (niter > 1) stage(1, 1);
    // The K / V fragments are ordinary loads issued BEFORE the DMAs.  Touching them here makes hipcc place their
    // (counted) wait in front of the loop; left to their first use it would sit inside the loop as vmcnt(0) and
    // turn every tile's DMA synchronous.
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(bk[n][ks]), "+v"(bv[n][ks]));
    float4_t dka[4][2], dva[4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            dka[c][n] = float4_t{0.f, 0.f, 0.f, 0.f};
            dva[c][n] = float4_t{0.f, 0.f, 0.f, 0.f};
        }
    // per-lane LDS offsets inside a tile image.  Row reads: row 32 wave + 16 m + fr, chunk (4 ks + g) ^ (fr & 7).
    // Transposed reads: lane (g, qq, pp) addresses row 32 wave + 16 h + 4 g + qq, hd columns 16 c + 4 pp .. + 3, i.e.
    // chunk (2 c + (pp >> 1)) ^ (row & 7) = (2 c) ^ x with x = (pp >> 1) ^ (4 (g & 1) + qq); h is an immediate offset.
    const int qq = fr >> 2, pp = fr & 3;
    const int xs = (pp >> 1) ^ (4 * (g & 1) + qq);
    unsigned tr_off[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) tr_off[c] = (32 * wave + 4 * g + qq) * 128 + (((2 * c) ^ xs) << 4) + 8 * (pp & 1);
    unsigned row_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) row_off[ks] = (32 * wave + fr) * 128 + (((4 * ks + g) ^ (fr & 7)) << 4);
    const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

#ifdef RPO_FA_STAMP
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ta, tb_, tc, td, te, tf, tg;
#endif
    int cur = 0;
    for (int it = 0; it < niter; ++it) {
        RPO_STAMP(ta);
        if (it + 1 < niter) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");   // tile it landed; tile it + 1 may fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RPO_STAMP(tb_);
        __builtin_amdgcn_s_barrier();
        RPO_STAMP(tc);
        if (it + 2 < niter) stage(it + 2, cur == 0 ? 2 : cur - 1);               // the buffer read in iteration it - 1
        RPO_STAMP(td);
        RPO_STAMP_ADD(0, ta, tb_);
        RPO_STAMP_ADD(1, tb_, tc);
        RPO_STAMP_ADD(2, tc, td);
        const char* Qs = smem + cur * kDmaTile;
        const char* Ds = Qs + kFaBM * 128;
        const float* Ls = reinterpret_cast<const float*>(Qs + 2 * kFaBM * 128);
        const float* Dl = Ls + kFaBM;
        const int qb = qt0 + (it % nqt) * kFaBM;
        const int qw = qb + 32 * wave;                       // this wave's 32 queries
        const bool active = (qw + 31 >= k0) && (qw < len) && (k0 < len);
        if (active) {
            const unsigned tb = smem_base + cur * kDmaTile;
            u32x2 d0, d1, d2, d3, d4, d5, d6, d7, e0, e1, e2, e3, e4, e5, e6, e7;
            {   // dO^T fragments: image offset 16384, second query half + 16 rows = 2048 B
                const unsigned a0 = tb + tr_off[0], a1 = tb + tr_off[1], a2 = tb + tr_off[2], a3 = tb + tr_off[3];
                RPO_TR2(d0, d1, a0, 16384, 18432);
                RPO_TR2(d2, d3, a1, 16384, 18432);
                RPO_TR2(d4, d5, a2, 16384, 18432);
                RPO_TR2(d6, d7, a3, 16384, 18432);
            }
            // the row constants -lse / scale and -delta (rows = queries 4g + r of the accumulator tile) are the initial
            // accumulators of the S and dP chains: no subtraction per element afterwards
            float4_t s[2][2], dp[2][2];                       // [query tile m][key tile n]; rows = queries 4g + r
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const float4_t lr = *reinterpret_cast<const float4_t*>(Ls + 32 * wave + 16 * m + 4 * g);
                const float4_t dr = *reinterpret_cast<const float4_t*>(Dl + 32 * wave + 16 * m + 4 * g);
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    s[m][n] = lr;
                    dp[m][n] = dr;
                }
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const short8_t aq = *reinterpret_cast<const short8_t*>(Qs + row_off[ks] + m * 2048);
                    const short8_t ad = *reinterpret_cast<const short8_t*>(Ds + row_off[ks] + m * 2048);
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        s[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq, bk[n][ks], s[m][n], 0, 0, 0);
                        dp[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ad, bv[n][ks], dp[m][n], 0, 0, 0);
                    }
                }
            }
            RPO_STAMP(te);
            RPO_STAMP_ADD(3, td, te);
            {   // Q^T: lands under the exp / mask arithmetic below
                const unsigned a0 = tb + tr_off[0], a1 = tb + tr_off[1], a2 = tb + tr_off[2], a3 = tb + tr_off[3];
                RPO_TR2(e0, e1, a0, 0, 2048);
                RPO_TR2(e2, e3, a1, 0, 2048);
                RPO_TR2(e4, e5, a2, 0, 2048);
                RPO_TR2(e6, e7, a3, 0, 2048);
            }
            const bool need_mask = (qw < k0 + 31) || (qw + 32 > len) || (k0 + 32 > len);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int qr0 = qw + 16 * m + 4 * g;
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int key = k0 + 16 * n + fr;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float pv = __builtin_amdgcn_exp2f(s[m][n][r] * scale_log2e);
                        if (need_mask && (key > qr0 + r || key >= len || qr0 + r >= len)) pv = 0.f;
                        s[m][n][r] = pv;                                       // P
                        dp[m][n][r] = pv * dp[m][n][r];                         // dS / scale (scale: epilogue)
                    }
                }
            }
            short8_t pf[2], dsf[2];
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                pf[n] = pack_frag(s[0][n], s[1][n]);      // k-slots = queries {4g + j, 16 + 4g + (j - 4)} of the wave's 32
                dsf[n] = pack_frag(dp[0][n], dp[1][n]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(e0),
                           "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7)
                         :
                         : "memory");
            RPO_STAMP(tf);
            RPO_STAMP_ADD(4, te, tf);
            const short8_t atd[4] = {join_tr(d0, d1), join_tr(d2, d3), join_tr(d4, d5), join_tr(d6, d7)};
            const short8_t atq[4] = {join_tr(e0, e1), join_tr(e2, e3), join_tr(e4, e5), join_tr(e6, e7)};
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    dva[c][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(atd[c], pf[n], dva[c][n], 0, 0, 0);
                    dka[c][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(atq[c], dsf[n], dka[c][n], 0, 0, 0);
                }
            RPO_STAMP(tg);
            RPO_STAMP_ADD(5, tf, tg);
#ifdef RPO_FA_STAMP
            st_acc[6] += 1;
#endif
        }
#ifdef RPO_FA_STAMP
        st_acc[7] += 1;
#endif
        cur = cur == 2 ? 0 : cur + 1;
    }
#ifdef RPO_FA_STAMP
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&g_fa_stamp[wave8 * 8 + i], st_acc[i]);
#endif
    __syncthreads();                                        // every wave is done with the tile images
    float4_t* red = reinterpret_cast<float4_t*>(smem);       // [wave8][c in pair][n][lane]
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int cp = (pass & 1) * 2;                      // first hd tile of this pass
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int n = 0; n < 2; ++n)
                red[(wave8 * 4 + cc * 2 + n) * 64 + lane] = pass < 2 ? dka[cp + cc][n] : dva[cp + cc][n];
        __syncthreads();
        {   // wave (kh, qg) finishes tile (cc = qg >> 1, n = qg & 1) of its key half
            const int cc = wave >> 1, n = wave & 1;
            float4_t tot = red[((kh * 4 + 0) * 4 + cc * 2 + n) * 64 + lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) tot += red[((kh * 4 + w) * 4 + cc * 2 + n) * 64 + lane];
            if (pass < 2) tot *= scale;                         // dK = scale * Q^T (P (dP - delta))
            const int key = k0 + 16 * n + fr;
            if (key < len) {
                bf16_t* dst = (pass < 2 ? dk + (t0 + key) * sdk : dv + (t0 + key) * sdv) + hk * kFaHD + 16 * (cp + cc) + 4 * g;
                uint2 w2;
                w2.x = pack_bf16(tot[0], tot[1]);
                w2.y = pack_bf16(tot[2], tot[3]);
                *reinterpret_cast<uint2*>(dst) = w2;
            }
        }
        __syncthreads();
    }
}

}  // namespace

#ifdef RPO_FA_STAMP
extern "C" int rpo_debug_fa_stamps(unsigned long long* out64, int reset) {
    if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_fa_stamp), sizeof(unsigned long long) * 64) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[64] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_fa_stamp), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

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

extern "C" int rpo_flash_attn_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout,
                                  int64_t q_stride, int64_t k_stride, int64_t v_stride, int64_t out_stride,
                                  int64_t dout_stride, const int* cu_seqlens, const int* q_tiles, int64_t n_q_tiles,
                                  const int* k_tiles, int64_t n_k_tiles, int64_t total_tokens, int64_t num_heads,
                                  int64_t num_kv_heads, int64_t head_dim, float scale, const float* lse, float* delta,
                                  void* dq, void* dk, void* dv, int64_t dq_stride, int64_t dk_stride, int64_t dv_stride,
                                  rpo_stream_t stream) {
    if (!q || !k || !v || !out || !dout || !cu_seqlens || !q_tiles || !k_tiles || !lse || !delta || !dq || !dk || !dv)
        return RPO_ERR_INVALID_ARG;
    if (n_q_tiles <= 0 || n_k_tiles <= 0 || total_tokens <= 0) return RPO_ERR_INVALID_ARG;
    if (head_dim != kFaHD || num_heads <= 0 || num_kv_heads <= 0 || num_heads % num_kv_heads != 0 || num_heads > 65535)
        return RPO_ERR_UNSUPPORTED;
    if (q_stride % 8 || k_stride % 8 || v_stride % 8 || out_stride % 8 || dout_stride % 8 || dq_stride % 4 ||
        dk_stride % 4 || dv_stride % 4 || !rpo_aligned16(q) || !rpo_aligned16(k) || !rpo_aligned16(v) ||
        !rpo_aligned16(out) || !rpo_aligned16(dout))
        return RPO_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const float log2e = 1.4426950408889634f;
    float* nd = delta;                                       // scratch [2][num_heads][T]: -delta | -lse / scale
    float* nl = delta + num_heads * total_tokens;
    RPO_LAUNCH(fa_delta_kernel, dim3((unsigned)total_tokens), dim3(256), 0, st, (const bf16_t*)out, (const bf16_t*)dout,
               out_stride, dout_stride, (int)num_heads, total_tokens, lse, 1.0f / scale, nd, nl);
    int rc = rpo_launch_status();
    if (rc != RPO_OK) return rc;
    RPO_LAUNCH(fa_bwd_dq_kernel, dim3((unsigned)n_q_tiles, (unsigned)num_heads), dim3(kFaThreads), 0, st,
               (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)dout, q_stride, k_stride, v_stride,
               dout_stride, cu_seqlens, q_tiles, (int)num_heads, (int)num_kv_heads, scale * log2e, scale, nl, nd,
               total_tokens, (bf16_t*)dq, dq_stride);
    rc = rpo_launch_status();
    if (rc != RPO_OK) return rc;
    const unsigned dkdv_grid = (unsigned)(((n_k_tiles + 7) / 8) * 8);
    static const bool attr_set = [] {
        (void)hipFuncSetAttribute((const void*)fa_bwd_dkdv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kDmaLds);
        return true;
    }();
    (void)attr_set;
    RPO_LAUNCH(fa_bwd_dkdv_kernel, dim3(dkdv_grid), dim3(kFaDkdvThreads), kDmaLds, st, (const bf16_t*)q, (const bf16_t*)k,
               (const bf16_t*)v, (const bf16_t*)dout, q_stride, k_stride, v_stride, dout_stride, cu_seqlens, k_tiles,
               (int)num_heads, (int)num_kv_heads, scale * log2e, scale, nl, nd, total_tokens, (bf16_t*)dk,
               (bf16_t*)dv, dk_stride, dv_stride, (int)n_k_tiles);
    return rpo_launch_status();
}
