// (2) similarity GEMM + temperature + InfoNCE cross-entropy, fused.  Reference: modeling.py:292-314, :321.
//
// Forward kernels (all write the temperature-scaled scores in the storage dtype and, per score row and
// per block of passage columns, an online-softmax partial (max, sum exp)):
//   sim_tile_kernel    Q > 64: 128(p) x 128(q) x 128-byte MFMA tile, LDS double-buffered, filled by
//                      16-byte global_load_lds with an XOR-swizzled source (conflict-free ds_read_b128),
//                      XCD-aware block order.  bf16: v_mfma_f32_16x16x32_bf16; f32: v_mfma_f32_16x16x4_f32.
//   sim_skinny_kernel  Q <= 64 (every shape the reference scripts produce: 8x48 ... 64x384): 16 queries x 16..64 passage
//                      rows per pass, fragments loaded straight into registers, K split over the 8 waves; small problems
//                      run as ONE block that also finishes lse and the loss (one launch).
//   sim_rowwise_kernel any d / alignment (scalar loads), one block per score row.
//   ce_finalize_kernel combines the partials -> lse[Q], loss (fixed summation order; a ticket counter with
//                      agent-scope release/acquire picks the block that adds up the per-block sums).
// The MFMA computes scores^T tiles (A = passages, B = queries) so that each lane ends up with 4 consecutive
// passage columns of ONE query row: 8/16-byte score stores and an almost lane-local row reduction.
//
// Backward: dS = grad * (softmax(S) - onehot) / (Q T) is recomputed from the stored scores and lse;
//   dq = dS p (own q rows), dp = dS^T q (own p rows).
#include "common.hpp"
#include <algorithm>
#include <atomic>
#include <cstdlib>

int rpo_launch_grouped_dots(const void* q, const void* p, int64_t B, int64_t G, int64_t d, int dtype, float* out,
                            hipStream_t st);

namespace {

#define RPO_NEG_INF (-__builtin_huge_valf())

// ------------------------------------------------------------------------------------------------
// MFMA fragment op per storage type.  A fragment is 16 bytes per lane; lane group g = lane>>4 holds the
// 16-byte chunk g of a 64-byte K segment of row (lane & 15).
//   bf16: chunk g = k 8g..8g+7  -> exactly the v_mfma_f32_16x16x32_bf16 operand layout.
//   f32 : chunk g = k 4g..4g+3  -> four v_mfma_f32_16x16x4_f32, MFMA t consuming element t of every
//         lane's chunk (k = 4g + t on lane group g, for A and B alike), i.e. a permuted but complete K sum.
// ------------------------------------------------------------------------------------------------
template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    typedef short8_t Frag;
    __device__ static __forceinline__ void mma(const Frag& a, const Frag& b, float4_t& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    __device__ static __forceinline__ Frag zero() { return Frag{0, 0, 0, 0, 0, 0, 0, 0}; }
};
template <> struct Mma<f16_t> {          // f16: the same operand layout, v_mfma_f32_16x16x32_f16
    typedef half8_t Frag;
    __device__ static __forceinline__ void mma(const Frag& a, const Frag& b, float4_t& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    __device__ static __forceinline__ Frag zero() { return Frag{0, 0, 0, 0, 0, 0, 0, 0}; }
};
template <> struct Mma<float> {
    typedef float4_t Frag;
    __device__ static __forceinline__ void mma(const Frag& a, const Frag& b, float4_t& c) {
#pragma unroll
        for (int t = 0; t < 4; ++t) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[t], c, 0, 0, 0);
    }
    __device__ static __forceinline__ Frag zero() { return Frag{0.f, 0.f, 0.f, 0.f}; }
};

// scores = round(round(dot) / T) in the storage dtype (the reference's rounding points for bf16; identity roundings
// for f32).  T == 1 (eval branch) skips the division.  The quotient is x * (1/T) refined by one FMA pair
// (r = x - q T; q += r / T): correctly rounded except in rare last-bit cases, at 3 VALU slots instead of the ~10 of
// the IEEE division sequence -- the epilogue is 128 elements per lane in the 256 x 256 kernel.
template <typename T>
__device__ __forceinline__ float finish_score(float acc, float temperature, float inv_temperature, bool scale) {
    float x = Elem<T>::round(acc);
    if (scale) {
        float qv = x * inv_temperature;
        const float r = fmaf(-qv, temperature, x);
        qv = fmaf(r, inv_temperature, qv);
        x = Elem<T>::round(qv);
    }
    return x;
}

// two f32 -> one dword of two bf16 (round to nearest even, NaN kept) in ONE v_cvt_pk_bf16_f32; written as
// `f32_to_bf16(a) | f32_to_bf16(b) << 16` hipcc converts each value on its own and merges them with a third instruction.
__device__ __forceinline__ unsigned pack2_bf16(float a, float b) {     // a vector conversion, not asm: hipcc must see the instruction (hazards)
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_));
}

// exp(v - m) as one FMA + one v_exp_f32
#define RPO_LOG2E 1.4426950408889634f
__device__ __forceinline__ float exp_sub(float v, float m_log2e) { return __builtin_amdgcn_exp2f(fmaf(v, RPO_LOG2E, -m_log2e)); }

__device__ __forceinline__ void softmax_merge(float& m, float& l, float om, float ol) {
    const float M = fmaxf(m, om);
    const float a = (m == RPO_NEG_INF) ? 0.f : l * __expf(m - M);
    const float b = (om == RPO_NEG_INF) ? 0.f : ol * __expf(om - M);
    m = M;
    l = a + b;
}

// Store 4 consecutive scores of one row.
template <typename T>
__device__ __forceinline__ void store_scores4(T* row_ptr, int64_t col, int64_t ncols, const float v[4], bool vec_ok) {
    if (vec_ok && col + 3 < ncols) {
        if constexpr (__is_same(T, bf16_t)) {
            uint2 w;
            w.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
            w.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
            *reinterpret_cast<uint2*>(row_ptr + col) = w;
        } else if constexpr (__is_same(T, f16_t)) {
            typedef __attribute__((ext_vector_type(4))) _Float16 half4_;
            const half4_ h = {(f16_t)v[0], (f16_t)v[1], (f16_t)v[2], (f16_t)v[3]};
            *reinterpret_cast<half4_*>(row_ptr + col) = h;
        } else {
            *reinterpret_cast<float4*>(row_ptr + col) = make_float4(v[0], v[1], v[2], v[3]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (col + j < ncols) Elem<T>::st(row_ptr + col + j, v[j]);
    }
}

// ------------------------------------------------------------------------------------------------
// Big tile kernel
// ------------------------------------------------------------------------------------------------
// Tile = TP passages x TQ queries, 4 waves as 2 x 2, TP / 2 x TQ / 2 scores^T per wave.  128 x 128 is the efficient shape
// (16 MFMAs per 8 LDS fragment reads); a grid of few such tiles leaves most of the 256 CUs idle -- Q = P = 1024 is 64 tiles --
// so the host picks 128 x 64 or 64 x 64 when the larger tile would not give every CU two blocks (round 4: SURVEY 8d's
// 1024^2 sweep points and every mid-size shape between the skinny kernel and the 256 x 256 kernel).
constexpr int kTileP = 128, kTileRowBytes = 128;   // kTileP: the LARGEST tile edge
constexpr int kTileThreads = 256;
// K loop: a ring of S stages of (A | B) K-steps filled by LDS-DMA, S - 1 of them in flight, ONE raw barrier per K-step, counted
// s_waitcnt vmcnt.  A small tile's K-step is ~100 ns of MFMAs against ~1 us of load latency, and the grids that get small tiles
// have one block per CU, so nothing else hides it: with the two-stage ring of rounds 1-3 the 64 x 64 tile ran at one K-step per
// memory round trip (29 us for 32 K-steps at Q = P = 1024, d = 2048).  S = 2 at 128 x 128 (64 KiB, two blocks per CU, as before),
// 3 at 128 x 64 (72 KiB, two blocks), 8 at 64 x 64 (128 KiB, one block).
constexpr int tile_lds_bytes(int TP, int TQ, int S) { return S * (TP + TQ) * kTileRowBytes; }

template <int N>
__device__ __forceinline__ void wait_vmcnt_imm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }
// all but the `later` youngest STAGES of this wave have landed (IPS DMA instructions per stage); later <= J
template <int J, int IPS>
struct WaitStages {
    static __device__ __forceinline__ void go(int later) {
        if (later >= J) wait_vmcnt_imm<J * IPS>();
        else WaitStages<J - 1, IPS>::go(later);
    }
};
template <int IPS>
struct WaitStages<0, IPS> {
    static __device__ __forceinline__ void go(int) { wait_vmcnt_imm<0>(); }
};

template <typename T, int TP, int TQ, int S>
__global__ __launch_bounds__(kTileThreads, 2) void sim_tile_kernel(
    const T* __restrict__ q, const T* __restrict__ p, int64_t Q, int64_t P, int64_t d, float temperature,
    int scale, int do_stats, T* __restrict__ scores, float2* __restrict__ partial, int nPt, int nQt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Mma<T>::Frag Frag;
    constexpr int KE = kTileRowBytes / (int)sizeof(T);   // K elements per tile row (64 bf16 / 32 f32)
    constexpr int CE = 16 / (int)sizeof(T);              // elements per 16-byte chunk
    constexpr int MA = TP / 32, NB = TQ / 32;            // 16-row fragments per wave along passages / queries
    constexpr int kABuf = TP * kTileRowBytes, kStage = (TP + TQ) * kTileRowBytes;
    static_assert((S - 1) * (MA + NB) <= 63, "vmcnt is a 6-bit counter");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave >> 1, wq = wave & 1;
    const int g = lane >> 4;

    // XCD-aware order: blocks b and b+8 share an XCD (speed only); give each XCD a contiguous run of
    // tiles, walked in groups of 8 passage tiles so that co-resident blocks share q / p panels in L2.
    const int nwg = nPt * nQt;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    constexpr int GROUP = 8;
    const int width = GROUP * nQt;
    const int group_id = wg / width;
    const int first_p = group_id * GROUP;
    const int gsz = min(nPt - first_p, GROUP);
    const int pt = first_p + (wg % width) % gsz;
    const int qt = (wg % width) / gsz;
    const int64_t p0 = (int64_t)pt * TP, q0 = (int64_t)qt * TQ;

    // staging: instruction i of wave w fills tile rows (4i + w)*8 .. +7 (1 KiB, lane-linear in LDS);
    // lane l carries row (l >> 3), physical chunk (l & 7) = logical chunk (l & 7) ^ (row & 7).
    const int srow = lane >> 3;
    const int lchunk = (lane & 7) ^ srow;
    const T* a_src[MA];
    const T* b_src[NB];
#pragma unroll
    for (int i = 0; i < MA; ++i)          // clamp: edge rows are masked later
        a_src[i] = p + min(p0 + (4 * i + wave) * 8 + srow, P - 1) * d + lchunk * CE;
#pragma unroll
    for (int i = 0; i < NB; ++i) b_src[i] = q + min(q0 + (4 * i + wave) * 8 + srow, Q - 1) * d + lchunk * CE;
    auto stage = [&](int t, int buf) {
        const int64_t k0 = (int64_t)t * KE;
#pragma unroll
        for (int i = 0; i < MA; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + k0),
                                             (__attribute__((address_space(3))) void*)(smem + buf * kStage + (4 * i + wave) * 1024),
                                             16, 0, 0);
#pragma unroll
        for (int i = 0; i < NB; ++i)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(b_src[i] + k0),
                (__attribute__((address_space(3))) void*)(smem + buf * kStage + kABuf + (4 * i + wave) * 1024), 16, 0, 0);
    };

    float4_t acc[MA][NB];
#pragma unroll
    for (int m = 0; m < MA; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[m][n] = float4_t{0.f, 0.f, 0.f, 0.f};

    const int nk = (int)(d / KE);
    const int frow = lane & 15;
#pragma unroll
    for (int t = 0; t < S - 1; ++t)
        if (t < nk) stage(t, t);
    int cur = 0, nxt = S - 1;                                // ring slots of K-step t and of K-step t + S - 1
    for (int t = 0; t < nk; ++t) {
        // K-step t has landed: this wave issued min(S - 2, nk - 1 - t) younger stages; every wave says so at the barrier, which
        // also tells that everybody is done reading K-step t - 1, whose slot the next DMA overwrites
        WaitStages<S - 2, MA + NB>::go(min(S - 2, nk - 1 - t));
        __builtin_amdgcn_s_barrier();
        if (t + S - 1 < nk) stage(t + S - 1, nxt);
        const char* Ab = smem + cur * kStage + (wp * (TP / 2) + frow) * kTileRowBytes;
        const char* Bb = smem + cur * kStage + kABuf + (wq * (TQ / 2) + frow) * kTileRowBytes;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int coff = (((ks * 4 + g) ^ (lane & 7)) << 4);
            Frag a[MA], b[NB];
#pragma unroll
            for (int m = 0; m < MA; ++m) a[m] = *reinterpret_cast<const Frag*>(Ab + m * 16 * kTileRowBytes + coff);
#pragma unroll
            for (int n = 0; n < NB; ++n) b[n] = *reinterpret_cast<const Frag*>(Bb + n * 16 * kTileRowBytes + coff);
#pragma unroll
            for (int m = 0; m < MA; ++m)
#pragma unroll
                for (int n = 0; n < NB; ++n) Mma<T>::mma(a[m], b[n], acc[m][n]);
        }
        cur = cur + 1 == S ? 0 : cur + 1;
        nxt = nxt + 1 == S ? 0 : nxt + 1;
    }
    __syncthreads();                                         // the ring is free: the epilogue reuses its first bytes

    // ---- epilogue: acc[m][n][j] = <p_{pbase + 16m + 4g + j}, q_{qbase + 16n + (lane&15)}>
    const int64_t pbase = p0 + wp * (TP / 2) + g * 4;
    const int64_t qbase = q0 + wq * (TQ / 2) + frow;
    const bool vec_ok = (P % 4 == 0) && rpo_aligned16_dev(scores);
    const float inv_t = 1.0f / temperature;
    float2* s_stat = reinterpret_cast<float2*>(smem);   // [wq][TQ / 2] from the wp == 1 waves (LDS is free now)
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int64_t qi = qbase + 16 * n;
        const bool qv = qi < Q;
        float mx = RPO_NEG_INF;
        float v[MA][4];
#pragma unroll
        for (int m = 0; m < MA; ++m) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[m][j] = finish_score<T>(acc[m][n][j], temperature, inv_t, scale);
                if (pbase + 16 * m + j < P) mx = fmaxf(mx, v[m][j]);
            }
            if (qv) store_scores4<T>(scores + qi * P, pbase + 16 * m, P, v[m], vec_ok);
        }
        if (do_stats) {
            float sum = 0.f;
#pragma unroll
            for (int m = 0; m < MA; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (pbase + 16 * m + j < P) sum += exp_sub(v[m][j], mx * RPO_LOG2E);
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {
                const float om = __shfl_xor(mx, o, 64), ol = __shfl_xor(sum, o, 64);
                softmax_merge(mx, sum, om, ol);
            }
            if (wp == 1 && g == 0) s_stat[wq * (TQ / 2) + 16 * n + frow] = make_float2(mx, sum);
            acc[0][n][0] = mx;   // keep for the cross-wave merge below
            acc[0][n][1] = sum;
        }
    }
    if (do_stats) {
        __syncthreads();
        if (wp == 0 && g == 0) {
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const int64_t qi = qbase + 16 * n;
                float mx = acc[0][n][0], sum = acc[0][n][1];
                const float2 o = s_stat[wq * (TQ / 2) + 16 * n + frow];
                softmax_merge(mx, sum, o.x, o.y);
                if (qi < Q) partial[(int64_t)pt * Q + qi] = make_float2(mx, sum);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 256 x 256 tile kernel (bf16, Q and P >= 512): 8 waves as 2 (passage halves) x 4 (query quarters), 128 x 64 scores^T
// per wave = 32 accumulator fragments (128 VGPRs), K-step 64, ONE block per CU, LDS 128 KiB = 2 buffers x 4 units of
// 16 KiB.  A K-step is cut into 4 phases; each phase issues the ds_reads of one operand sub-tile, the 16-byte
// global_load_lds of ONE unit of the NEXT K-step, a COUNTED s_waitcnt vmcnt(4) (two units stay in flight across the
// raw s_barrier), and 16 MFMAs (one quadrant of the wave's tile over the whole K-step); the two wave groups
// (wp = 0 / 1, SIMD partners) run half a phase apart so that MFMAs of one overlap the loads of the other:
//
//   phase   ds_read (this K-step)            glds (next K-step)   MFMAs (acc quadrant)
//     0     A rows sub0 (8) + B sub0 (4)     unit 0 = A sub0 rows  m 0-3 x n 0-1
//     1     B sub1 (4)                       unit 1 = B sub0 rows  m 0-3 x n 2-3
//     2     A rows sub1 (8)                  unit 2 = B sub1 rows  m 4-7 x n 2-3
//     3     -                                unit 3 = A sub1 rows  m 4-7 x n 0-1
//
// Units are laid out by FIRST USE (unit 0/1 are needed in phase 0, unit 2 in phase 1, unit 3 in phase 2), so that the
// wait that retires a unit always sits one full phase (one barrier) before its first read, and a unit is overwritten
// four phases after its last read.  LDS rows are 128 bytes, lane-linear per 1 KiB DMA piece, XOR-swizzled
// (chunk ^= row & 7) on the per-lane SOURCE address and on the ds_read_b128 address.
// ------------------------------------------------------------------------------------------------
constexpr int kBigTile = 256, kBigThreads = 512, kBigUnitBytes = 128 * kTileRowBytes;   // 16 KiB
constexpr int kBigBufBytes = 4 * kBigUnitBytes;                                         // 64 KiB per buffer
constexpr int kBigStageRowBytes = 272, kBigStageWaveBytes = 64 * kBigStageRowBytes;     // epilogue score staging
constexpr int kBigLdsBytes = 8 * kBigStageWaveBytes + 4 * 64 * 8;                       // 141,312 B >= 2 buffers
constexpr int64_t kBigPersistBlocks = 256;                                              // one per CU (a multiple of 8: XCD affinity)

// tile row (0..255) of unit-local row u (0..127)
__device__ __forceinline__ int big_unit_row(int unit, int u) {
    switch (unit) {
        case 0: return (u & 63) + ((u >> 6) << 7);              // A: rows   0-63, 128-191
        case 3: return 64 + (u & 63) + ((u >> 6) << 7);         // A: rows 64-127, 192-255
        case 1: return ((u >> 5) << 6) + (u & 31);              // B: rows 64 wq + 0..31
        default: return ((u >> 5) << 6) + 32 + (u & 31);        // B: rows 64 wq + 32..63
    }
}

// PERSISTENT since round 5 (DESIGN.md section 8.3 named it in round 2): one block per CU walks its tiles (launched with one block
// per TILE it is the kernel of rounds 1-4: `have_next` is never true; that launch is the A/B arm, -DRPO_SIM_PERSIST=0).
// What it buys: the 8 LDS-DMA instructions of a tile's first K-step are issued BEFORE the previous tile's epilogue (its address
// arithmetic, the 1-2 us flight of the first units and the block relaunch hide under ~1500 vector instructions of epilogue), and the
// score stores of tile i drain under the first K-step of tile i + 1: that K-step's counted waits are widened by the 16 store
// instructions a wave issued behind the DMAs (CDNA4's vmcnt counts stores too) -- which is why the interior epilogue's stores
// are asm here (an exact count), and why an edge tile, whose store count varies, ends in a full drain.  The epilogue stages through
// the ring's SECOND buffer only (16 rows per wave at a time), so that the first buffer can take the next tile meanwhile.
// EPI = 0: the scoring forward (scores = <q, p> / T with the reference's two bf16 roundings, per-tile softmax partials).
// EPI = 1 (round 5): the same frame as a plain NT GEMM, C[Q, P] = q[Q, K] p[P, K]^T rounded to bf16 once -- the two products of the
// scoring BACKWARD at sweep sizes, dq = dS p_all and dp = dS^T q_all, with the reduction operand transposed beforehand so that
// both operands are contiguous along the reduction like the forward's (rpo_sim_gemm_nt).  lda / ldb / ldc: row strides (elements)
// of p, q and the output (EPI 0: d, d, P).
// EPI = 2 (round 6): the exact SEARCH step (rpo_sim_topk_filter).  The score matrix never reaches HBM: a score (one rounding to bf16,
// the eval similarity's) is compared with its query row's current k-th winner in the accumulator registers and only a score that
// beats it is appended to the row's candidate list (atomic slot counter in global memory; a few dozen per row and corpus chunk once
// the winners have seen one chunk).  The tile's 256 thresholds sit in the 10 KiB of LDS behind the ring, loaded one tile ahead.
// No counted stores: every tile ends in a full drain, like an edge tile of EPI 0.
// EPI = 3: EPI 2 on the UNROUNDED f32 accumulators -- the search step of an f32 index whose embeddings are exactly representable in
// bf16 (what an encoder that computes in bf16 hands over): the bf16 products are exact, the sums are f32 sums, i.e. an f32 inner
// product in this frame's summation order at 16 x the f32 MFMA rate.
//          (F16 = true: the same for values exact in fp16 -- an fp16 encoder's output -- with the f16 MFMA.)
// EPI = 4: the score matrix of the same problem as f32 (`scores` is a float*, ldc its row stride in elements), direct 16-byte
// stores: the first chunk of such a search (small), so that every score of it comes out of ONE summation order.
struct SimFilter {
    const float* best_val;        // [Q, k] winners so far, best first: row r's threshold is (best_val, best_idx)[r k + k - 1]
    const long long* best_idx;
    float* cand_val;              // [Q, cap]
    long long* cand_idx;          // [Q, cap]
    int* cand_cnt;                // [Q]: candidates appended (may exceed cap: the merge reports the overflow)
    int64_t col0;                 // corpus index of p's row 0
    int k, cap;
};
constexpr int kBigThrOff = 2 * kBigBufBytes;      // EPI 2: float tv[256] | long long ti[256] (3 KiB of the 10 KiB behind the ring)

// F16 (EPI 3 / 4 only: the outputs are f32): the operands are fp16 -- the same 2-byte staging and fragment layout, v_mfma_f32_16x16x32_f16.
template <bool F16>
__device__ __forceinline__ float4_t big_mma(const short8_t a, const short8_t b, const float4_t c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

template <int EPI, bool F16 = false>
__global__ __launch_bounds__(kBigThreads, 2) void sim_tile256_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ p, int64_t Q, int64_t P, int64_t d, int64_t lda, int64_t ldb,
    int64_t ldc, float temperature, int scale, int do_stats_arg, bf16_t* __restrict__ scores, float2* __restrict__ partial,
    int nPt, int nQt, int stagger, int dbg, const SimFilter flt) {
    const int do_stats = EPI == 0 ? do_stats_arg : 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef short8_t Frag;
    constexpr int KE = 64;   // bf16 elements per K-step

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave >> 2, wq = wave & 3;
    const int g = lane >> 4, frow = lane & 15;

    const int nwg = nPt * nQt;
    // virtual block vb -> tile: the one-tile kernel's XCD-aware order (blocks that share an XCD walk neighbouring tiles); this block
    // takes vb = blockIdx.x, + gridDim.x, ... (gridDim.x is a multiple of 8: all of them on this block's XCD)
    auto tile_of = [&](int vb, int64_t& p0_, int64_t& q0_, int& pt_) {
        const int xcd = vb & 7, q8 = nwg >> 3, r8 = nwg & 7;
        const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (vb >> 3);
        constexpr int GROUP = 4;
        const int width = GROUP * nQt;
        const int first_p = (wg / width) * GROUP;
        const int gsz = min(nPt - first_p, GROUP);
        pt_ = first_p + (wg % width) % gsz;
        p0_ = (int64_t)pt_ * kBigTile;
        q0_ = (int64_t)((wg % width) / gsz) * kBigTile;
    };
    int64_t p0, q0;
    int pt;
    tile_of((int)blockIdx.x, p0, q0, pt);

    // staging: a unit is 16 DMA pieces of 1 KiB (8 rows); wave w issues pieces w and 8 + w of every unit.
    const int srow = lane >> 3;
    const int lchunk = (lane & 7) ^ srow;
    // per-lane BYTE OFFSETS of this wave's 8 pieces from the two operand bases (32 bits each: the host admits operands below 4 GB);
    // pointers would be 16 registers that have to live through the previous tile's epilogue, where the pressure peaks
    unsigned soff[4][2];
    auto set_src = [&](int64_t p0_, int64_t q0_) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ur = (8 * j + wave) * 8 + srow;           // unit-local row
                const int tr = big_unit_row(u, ur);                 // tile row
                const bool isA = (u == 0 || u == 3);
                const int64_t gr = isA ? min(p0_ + tr, P - 1) : min(q0_ + tr, Q - 1);
                soff[u][j] = (unsigned)((gr * (isA ? lda : ldb) + lchunk * 8) * 2);
            }
    };
    set_src(p0, q0);
#define RPO_BIG_STAGE(U, T, BUF)                                                                                   \
    do {                                                                                                           \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                                         \
            char* dst_ = smem + (BUF) * kBigBufBytes + (U) * kBigUnitBytes + (8 * j_ + wave) * 1024;               \
            const char* base_ = reinterpret_cast<const char*>(((U) == 0 || (U) == 3) ? p : q);                     \
            __builtin_amdgcn_global_load_lds(                                                                      \
                (const __attribute__((address_space(1))) void*)(base_ + (soff[U][j_] + (unsigned)(T) * (KE * 2))), \
                (__attribute__((address_space(3))) void*)dst_, 16, 0, 0);                                          \
        }                                                                                                          \
    } while (0)

    const int nk = (int)(d / KE);
    // per-lane read offsets inside a unit (row part); the chunk part depends on the k-step
    const int a_off = (wp * 64 + frow) * kTileRowBytes;         // + m*16 rows, units 0 / 3
    const int b_off = (wq * 32 + frow) * kTileRowBytes;         // + n*16 rows, units 1 / 2
    const int c0 = ((0 * 4 + g) ^ (lane & 7)) << 4, c1 = ((1 * 4 + g) ^ (lane & 7)) << 4;
    const float inv_t = 1.0f / temperature;
    const bool vec_ok = (ldc % 4 == 0) && rpo_aligned16_dev(scores);
    const bool staged = (ldc % 8 == 0) && rpo_aligned16_dev(scores) && !(dbg & 2);
    // epilogue staging: 16 query rows x 128 passages of bf16 per wave at a time, in the ring's SECOND buffer
    char* wstage = smem + kBigBufBytes + wave * (16 * kBigStageRowBytes);
    float2* s_stat = reinterpret_cast<float2*>(smem + 8 * kBigStageWaveBytes);   // [wq][64] from the wp == 1 waves

    float* s_tv = reinterpret_cast<float*>(smem + kBigThrOff);
    long long* s_ti = reinterpret_cast<long long*>(smem + kBigThrOff + kBigTile * 4);
    // thread t fetches the threshold of query row q0_ + (t & 255), clamped into the matrix (the epilogue does not use a row outside
    // it).  UNCONDITIONAL, and so is the wait + LDS write that follows at the tile's end: with either under a condition hipcc sees a
    // path on which the loads are still pending at the K loop's head and puts a vmcnt(0) there -- the end of every counted wait.
    auto load_thr = [&](int64_t q0_, float& tv_, long long& ti_) {
        const int64_t r = min(q0_ + (tid & (kBigTile - 1)), Q - 1);
        tv_ = flt.best_val[r * flt.k + flt.k - 1];
        ti_ = flt.best_idx[r * flt.k + flt.k - 1];
    };
    if constexpr (EPI == 2 || EPI == 3) {      // BEFORE the first DMAs (nothing in flight yet); visible to the epilogue through the loop's last barrier
        float tv_;
        long long ti_;
        load_thr(q0, tv_, ti_);
        s_tv[tid & (kBigTile - 1)] = tv_;
        s_ti[tid & (kBigTile - 1)] = ti_;
    }
    RPO_BIG_STAGE(0, 0, 0);
    RPO_BIG_STAGE(1, 0, 0);
    RPO_BIG_STAGE(2, 0, 0);
    RPO_BIG_STAGE(3, 0, 0);
    bool carry = false;      // 16 score stores of the previous tile were issued BEHIND this tile's first DMAs and may still fly
    for (int vb = (int)blockIdx.x; vb < nwg; vb += (int)gridDim.x) {
    float4_t acc[8][4];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = float4_t{0.f, 0.f, 0.f, 0.f};
    // units 0, 1 of K-step 0 have landed (this wave's pieces): all but units 2, 3 (4 instructions) and the carried stores
    if (carry) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // ... everybody's; and every wave is through the previous tile's epilogue (its LDS staging area
                                       // lies in the buffer K-step 0 stages K-step 1 into)

    // Two barriers per phase (load part | MFMA part) and the two wave groups offset by ONE barrier: while waves 0-3
    // (wp = 0) run their 16 MFMAs, waves 4-7 -- their SIMD partners -- issue ds_reads / LDS-DMA, and vice versa, so
    // the matrix pipe of every SIMD always has one wave in its MFMA part.  A unit retired by the vmcnt of phase g is
    // read in phase g + 1 at the earliest: by then both groups' waits and one more barrier have passed.
    if (stagger && wp == 1) __builtin_amdgcn_s_barrier();
    Frag a[4][2], b0[2][2], b1[2][2];                           // [rep][k-step half]
    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1, nxt = cur ^ 1;
        const bool more = t + 1 < nk;
        const char* base = smem + cur * kBigBufBytes;
        // ---------------- phase 0
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            b0[n][0] = *reinterpret_cast<const Frag*>(base + 1 * kBigUnitBytes + b_off + n * 16 * kTileRowBytes + c0);
            b0[n][1] = *reinterpret_cast<const Frag*>(base + 1 * kBigUnitBytes + b_off + n * 16 * kTileRowBytes + c1);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            a[m][0] = *reinterpret_cast<const Frag*>(base + 0 * kBigUnitBytes + a_off + m * 16 * kTileRowBytes + c0);
            a[m][1] = *reinterpret_cast<const Frag*>(base + 0 * kBigUnitBytes + a_off + m * 16 * kTileRowBytes + c1);
        }
        if (more) {
            RPO_BIG_STAGE(0, t + 1, nxt);
            if (carry && t == 0) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");   // (+ the 16 stores between the tile's first DMAs and these)
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");    // retires unit 2 of this K-step (read in phase 1)
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m][n] = big_mma<F16>(a[m][h], b0[n][h], acc[m][n]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
        // ---------------- phase 1
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            b1[n][0] = *reinterpret_cast<const Frag*>(base + 2 * kBigUnitBytes + b_off + n * 16 * kTileRowBytes + c0);
            b1[n][1] = *reinterpret_cast<const Frag*>(base + 2 * kBigUnitBytes + b_off + n * 16 * kTileRowBytes + c1);
        }
        if (more) {
            RPO_BIG_STAGE(1, t + 1, nxt);
            if (carry && t == 0) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");    // retires unit 3 of this K-step (read in phase 2)
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m][2 + n] = big_mma<F16>(a[m][h], b1[n][h], acc[m][2 + n]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
        // ---------------- phase 2
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            a[m][0] = *reinterpret_cast<const Frag*>(base + 3 * kBigUnitBytes + a_off + m * 16 * kTileRowBytes + c0);
            a[m][1] = *reinterpret_cast<const Frag*>(base + 3 * kBigUnitBytes + a_off + m * 16 * kTileRowBytes + c1);
        }
        if (more) RPO_BIG_STAGE(2, t + 1, nxt);                 // nothing to retire here: units 0', 1' are due in phase 3
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[4 + m][2 + n] = big_mma<F16>(a[m][h], b1[n][h], acc[4 + m][2 + n]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
        // ---------------- phase 3
        if (more) {
            RPO_BIG_STAGE(3, t + 1, nxt);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");    // retires units 0', 1' of the next K-step (read in its phase 0)
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[4 + m][n] = big_mma<F16>(a[m][h], b0[n][h], acc[4 + m][n]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
    }
    if (stagger && wp == 0) __builtin_amdgcn_s_barrier();
    __syncthreads();   // every wave is done with the ring: the second buffer becomes the epilogue's staging area, the first one
                       // takes the next tile's first K-step NOW (its units fly under the epilogue)
    const int64_t p0e = p0, q0e = q0;      // this tile's origin, for the epilogue
    const int pte = pt;
    const bool have_next = vb + (int)gridDim.x < nwg;
    if (have_next) {
        tile_of(vb + (int)gridDim.x, p0, q0, pt);
        set_src(p0, q0);
        RPO_BIG_STAGE(0, 0, 0);
        RPO_BIG_STAGE(1, 0, 0);
        RPO_BIG_STAGE(2, 0, 0);
        RPO_BIG_STAGE(3, 0, 0);
    }
    float next_tv = 0.f;
    long long next_ti = 0;
    if constexpr (EPI == 2 || EPI == 3) load_thr(q0, next_tv, next_ti);   // (q0: the next tile's, or still this one's) in registers through the filter

    // ---- epilogue: acc[m][n][j] = <p_{pbase + 16m + j}, q_{qbase + 16n}>
    const int64_t pbase = p0e + wp * 128 + g * 4;
    const int64_t qbase = q0e + wq * 64 + frow;
    // Scores leave through LDS: each wave parks its 64 (q) x 128 (p) bf16 sub-tile in its own 64 x 272-byte image
    // (8-byte ds_writes of 4 consecutive p) and streams it out as whole 256-byte row segments with 16-byte stores,
    // instead of 8-byte stores that touch a quarter of a 128-byte line each (measured: the direct stores cost 14 %
    // of the kernel at Q = P = 16384).
    const bool interior = staged && p0e + kBigTile <= P && q0e + kBigTile <= Q;
    if constexpr (EPI == 4) {
        float* out = reinterpret_cast<float*>(scores);
        const bool v4 = (ldc % 4 == 0) && rpo_aligned16_dev(out);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int64_t qi = qbase + 16 * n;
            if (qi >= Q) continue;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int64_t pi = pbase + 16 * m;
                float* dst = out + qi * ldc + pi;
                if (v4 && pi + 3 < P) {
                    *reinterpret_cast<float4_t*>(dst) = acc[m][n];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (pi + j < P) dst[j] = acc[m][n][j];
                }
            }
        }
    }
    if constexpr (EPI == 2 || EPI == 3) {
        // Filter, in two passes over the lane's 4 rows x 32 scores.  Pass 1 marks the survivors (a bit per score); a lane then
        // reserves its slots with ONE atomic per row -- the (up to) four issued back to back, one wait for all -- and pass 2 writes
        // them.  (An atomic per survivor with its own wait -- the first form -- cost 23 % of the kernel at ~500 survivors per row
        // and chunk: every wait also drains the next tile's first DMAs.)
        unsigned mask[4];
        float tvn[4];
        long long tin[4];
        auto rounded = [&](int m, int n, float (&x)[4]) {       // the score as the search defines it: bf16 once (EPI 2) / the f32 sum (EPI 3)
            if constexpr (EPI == 3) {
#pragma unroll
                for (int j = 0; j < 4; ++j) x[j] = acc[m][n][j];
            } else {
                const unsigned w01 = pack2_bf16(acc[m][n][0], acc[m][n][1]), w23 = pack2_bf16(acc[m][n][2], acc[m][n][3]);
                x[0] = __uint_as_float(w01 << 16);
                x[1] = __uint_as_float(w01 & 0xffff0000u);
                x[2] = __uint_as_float(w23 << 16);
                x[3] = __uint_as_float(w23 & 0xffff0000u);
            }
        };
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int lr = wq * 64 + frow + 16 * n;             // the row inside the tile
            const bool qv = qbase + 16 * n < Q;
            tvn[n] = qv ? s_tv[lr] : INFINITY;                  // a row outside the matrix: nothing passes
            tin[n] = qv ? s_ti[lr] : -1;
            mask[n] = 0;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                float x[4];
                rounded(m, n, x);
                // the common case (none of the four reaches the threshold value) costs one max3 pair and one compare
                if (max3_raw(max3_raw(x[0], x[1], x[2]), x[3], x[3]) >= tvn[n]) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int64_t pi = pbase + 16 * m + j;
                        if (pi < P && (x[j] > tvn[n] || (x[j] == tvn[n] && flt.col0 + pi < tin[n]))) mask[n] |= 1u << (4 * m + j);
                    }
                }
            }
        }
        if (mask[0] | mask[1] | mask[2] | mask[3]) {
            int slot[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) slot[n] = mask[n] ? atomicAdd(flt.cand_cnt + (qbase + 16 * n), __popc(mask[n])) : 0;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                if (!mask[n]) continue;
                const int64_t qi = qbase + 16 * n;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    if (!((mask[n] >> (4 * m)) & 15u)) continue;
                    float x[4];
                    rounded(m, n, x);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if ((mask[n] >> (4 * m + j)) & 1u) {
                            const int pos = slot[n]++;
                            if (pos < flt.cap) {
                                flt.cand_val[qi * flt.cap + pos] = x[j];
                                flt.cand_idx[qi * flt.cap + pos] = flt.col0 + pbase + 16 * m + j;
                            }
                        }
                }
            }
        }
    }
#pragma unroll
    for (int n = 0; n < (EPI >= 2 ? 0 : 4); ++n) {
        const int64_t qi = qbase + 16 * n;
        const bool qv = qi < Q;
        float mx = RPO_NEG_INF, sum = 0.f;
        if constexpr (EPI == 1) {
            // plain GEMM epilogue: one rounding to bf16, parked in the wave's LDS image, streamed out below as 256-byte segments
            // (the host admits this instantiation only when the staged path applies: ldc % 8 == 0, 16-byte aligned output)
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                uint2 w;
                w.x = pack2_bf16(acc[m][n][0], acc[m][n][1]);
                w.y = pack2_bf16(acc[m][n][2], acc[m][n][3]);
                *reinterpret_cast<uint2*>(wstage + frow * kBigStageRowBytes + (16 * m + 4 * g) * 2) = w;
            }
        } else if (interior) {
            // interior tile (no row or column outside the matrix): no bounds tests, both roundings as packed conversions
            // (two scores per v_cvt_pk_bf16_f32, whose result IS the store payload): ~10 VALU instructions per score
            // instead of ~18 -- the epilogue is VALU-bound (128 scores per lane) and was 14 % of the kernel at d = 2048
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                uint2 w;
                w.x = pack2_bf16(acc[m][n][0], acc[m][n][1]);               // the reference's bf16 matmul output
                w.y = pack2_bf16(acc[m][n][2], acc[m][n][3]);
                float x[4] = {__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16),
                              __uint_as_float(w.y & 0xffff0000u)};
                if (scale) {                                                // x / T as finish_score computes it, then bf16
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float q0v = x[j] * inv_t;
                        const float r = fmaf(-q0v, temperature, x[j]);
                        x[j] = fmaf(r, inv_t, q0v);
                    }
                    w.x = pack2_bf16(x[0], x[1]);
                    w.y = pack2_bf16(x[2], x[3]);
                    x[0] = __uint_as_float(w.x << 16);
                    x[1] = __uint_as_float(w.x & 0xffff0000u);
                    x[2] = __uint_as_float(w.y << 16);
                    x[3] = __uint_as_float(w.y & 0xffff0000u);
                }
                *reinterpret_cast<uint2*>(wstage + frow * kBigStageRowBytes + (16 * m + 4 * g) * 2) = w;
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[m][n][j] = x[j];
                mx = max3_raw(max3_raw(mx, x[0], x[1]), x[2], x[3]);
            }
            if (do_stats) {
                const float ml = mx * RPO_LOG2E;
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int j = 0; j < 4; ++j) sum += exp_sub(acc[m][n][j], ml);
            }
        } else {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = finish_score<bf16_t>(acc[m][n][j], temperature, inv_t, scale);
                    acc[m][n][j] = v[j];
                    if (pbase + 16 * m + j < P) mx = fmaxf(mx, v[j]);
                }
                if (staged) {
                    uint2 w;
                    w.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                    w.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                    *reinterpret_cast<uint2*>(wstage + frow * kBigStageRowBytes + (16 * m + 4 * g) * 2) = w;
                } else if (qv && !(dbg & 1)) {
                    store_scores4<bf16_t>(scores + qi * ldc, pbase + 16 * m, P, v, vec_ok);
                }
            }
            if (do_stats) {
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (pbase + 16 * m + j < P) sum += exp_sub(acc[m][n][j], mx * RPO_LOG2E);
            }
        }
        if (do_stats) {
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {
                const float om = __shfl_xor(mx, o, 64), ol = __shfl_xor(sum, o, 64);
                softmax_merge(mx, sum, om, ol);
            }
            if (wp == 1 && g == 0) s_stat[wq * 64 + 16 * n + frow] = make_float2(mx, sum);
            acc[0][n][0] = mx;
            acc[0][n][1] = sum;
        }
        if (staged && !(dbg & 1)) {
            // stream this quarter (16 query rows) out now: its stores drain while the next quarter is being computed
            const int64_t prow0 = p0e + wp * 128 + (lane & 15) * 8;         // first of this lane's 8 passage columns
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 16 * n + 4 * i + (lane >> 4);
                const int64_t qi = q0e + wq * 64 + r;
                const uint4_t w = *reinterpret_cast<const uint4_t*>(wstage + (r & 15) * kBigStageRowBytes + (lane & 15) * 16);
                if (interior) {
                    // exactly ONE store instruction per (n, i) and wave: the next tile's first K-step counts them in its waits
                    bf16_t* dst = scores + qi * ldc + prow0;
                    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(dst), "v"(w) : "memory");
                } else if (qi < Q) {
                    bf16_t* dst = scores + qi * ldc + prow0;
                    if (prow0 + 7 < P) {
                        if (dbg & 4) *reinterpret_cast<uint4_t*>(dst) = w;
                        else __builtin_nontemporal_store(w, reinterpret_cast<uint4_t*>(dst));
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (prow0 + e < P) dst[e] = (bf16_t)(w[e >> 1] >> ((e & 1) * 16));
                    }
                }
            }
        }
    }
    if (do_stats) {
        __syncthreads();
        if (wp == 0 && g == 0) {
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int64_t qi = qbase + 16 * n;
                float mx = acc[0][n][0], sum = acc[0][n][1];
                const float2 o = s_stat[wq * 64 + 16 * n + frow];
                softmax_merge(mx, sum, o.x, o.y);
                if (qi < Q) partial[(int64_t)pte * Q + qi] = make_float2(mx, sum);
            }
        }
    }
    // an edge tile's stores are not counted (their number depends on the bounds): drain them, and with them the next tile's first
    // units, before the next tile starts; an interior tile leaves its 16 stores in flight
    if constexpr (EPI == 4) {
        if (have_next) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // uncounted stores: a full drain, like an edge tile
    } else if constexpr (EPI == 2 || EPI == 3) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // candidate stores, the next tile's first units, its thresholds
        __syncthreads();                                        // every wave has read this tile's thresholds
        s_tv[tid & (kBigTile - 1)] = next_tv;                   // (threads t and t + 256 write the same value)
        s_ti[tid & (kBigTile - 1)] = next_ti;
    } else {
        carry = have_next && interior && !(dbg & 1);
        if (have_next && !carry) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    }
#undef RPO_BIG_STAGE
}


// ------------------------------------------------------------------------------------------------
// Skinny kernel: Q <= 64 (every shape the reference scripts produce: 8 x 48 ... 64 x 384).
//   A pass = one group of 16 queries x NG groups of 16 passage rows (NG = 4 for Q <= 16, where q is small and is
//   amortised over four passage groups; NG = 1 above: a block that re-reads 64 queries per 16 passages spends 4/5 of its
//   bytes on q).  Block = 8 waves; block b takes passes b, b + gridDim.x, ...
//   Fragments go straight from global memory into registers (every byte is used once per pass).  K is split over the
//   8 waves, wave w taking the 64-byte segments w, w + 8, ...; a wave issues the loads of EIGHT of its segments
//   (8 (NG + 1) 16-byte loads per lane, up to ~200 VGPRs: 8 waves per block leave them the whole register file; a 16-wave
//   version spilled) before the first MFMA, so a cfg-2 block (8 x 48 x 2048: 224 KB) has its whole operand set in flight
//   at once and pays ONE memory round trip (the round-1 kernel walked 16 dependent load -> MFMA steps per wave).  The 8
//   partial tiles are summed through LDS in wave order (deterministic), one wave per tile, which then rounds / scales /
//   stores its 16 x 16 scores and reduces its rows.
//   gridDim.x == 1 (small problems, <= 384 KB: every single-GPU shape of the reference): the block walks all passes, then
//   merges the per-group softmax partials, writes lse and the mean loss itself -- ONE launch, no workspace traffic.
//   Otherwise partial[group][row] goes to the workspace and ce_finalize_kernel follows.
//   Measured anatomy of the one-block launch (rocprofv3, MI355X): 4.2 us for a near-empty problem (dispatch, kernel
//   arguments, one memory round trip, LDS sum, two barriers, end-of-kernel write-back) + ~29 us per MB: fragment-shaped
//   loads (16 rows x 64 B per wave instruction) bring one CU ~35 GB/s.
// ------------------------------------------------------------------------------------------------
constexpr int kSkinnyThreads = 512, kSkinnyWaves = 8, kSkinnyUnroll = 8, kSkinnyMaxFusedGroups = 16;

// One finished 16 (passages of group pg) x 16 (queries) tile, held as the MFMA accumulator layout (lane = query column
// lane & 15, registers = passages 4 (lane >> 4) + j): round / scale / store the scores, reduce the rows' softmax partial over
// the tile's 16 columns, and either park (max, sum) + the positive's score in LDS (fused single-block finalize) or write the
// partial to the workspace.
template <typename T>
__device__ __forceinline__ void skinny_tile_epilogue(const float4_t& t, int pg, int ngroups, int64_t qr, bool bv, int g,
                                                     int64_t Q, int64_t P, float temperature, int scale, int do_stats,
                                                     bool fused, int64_t group, T* __restrict__ scores,
                                                     float2* __restrict__ partial, float2 (*s_part)[64], float* s_tgt) {
    if (pg >= ngroups) return;
    const int64_t pbase = (int64_t)pg * 16 + g * 4;
    const bool vec_ok = (P % 4 == 0) && rpo_aligned16_dev(scores);
    const float inv_t = 1.0f / temperature;
    float v[4];
    float mx = RPO_NEG_INF;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[j] = finish_score<T>(t[j], temperature, inv_t, scale);
        if (pbase + j < P) mx = fmaxf(mx, v[j]);
    }
    if (bv) store_scores4<T>(scores + qr * P, pbase, P, v, vec_ok);
    if (!do_stats) return;
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (pbase + j < P) sum += exp_sub(v[j], mx * RPO_LOG2E);
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
        const float om = __shfl_xor(mx, o, 64), ol = __shfl_xor(sum, o, 64);
        softmax_merge(mx, sum, om, ol);
    }
    if (!bv) return;
    if (fused) {
        if (g == 0) s_part[pg][qr] = make_float2(mx, sum);
        const int64_t tgt = qr * group;                  // the positive's column (modeling.py:301-302)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (pbase + j == tgt) s_tgt[qr] = v[j];
    } else if (g == 0) {
        partial[(int64_t)pg * Q + qr] = make_float2(mx, sum);
    }
}

// lse / loss of a single-block launch from the per-group partials parked in LDS (thread i < Q owns row i)
__device__ __forceinline__ void skinny_fused_finalize(int tid, int64_t Q, int ngroups, float2 (*s_part)[64], const float* s_tgt,
                                                      float* s_red, float* __restrict__ lse_out, float* __restrict__ loss_out) {
    float rowloss = 0.f;
    if (tid < Q) {
        float m = RPO_NEG_INF, l = 0.f;
        for (int b2 = 0; b2 < ngroups; ++b2) softmax_merge(m, l, s_part[b2][tid].x, s_part[b2][tid].y);
        const float lse = m + logf(l);
        lse_out[tid] = lse;
        rowloss = lse - s_tgt[tid];
    }
    const float tot = block_sum<8>(rowloss, s_red);
    if (tid == 0) loss_out[0] = tot / (float)Q;
}

// Arrival counters of the multi-block launches: one 64-bit slot per launch, handed out round-robin by the host together with an
// EPOCH (the launch's sequence number / slot count + 1, never 0): word = epoch << 32 | blocks arrived.  A block whose epoch is
// not the word's starts the count over, so a slot needs neither a memset in front of the launch nor a reset behind it, and a
// launch that never finished (a fault, an abort) cannot leave a count behind that a later launch on the slot would inherit
// (round 3's plain counters did: the slot's next launches then never saw `gridDim.x - 1` and left lse / loss unwritten without
// an error -- advisor, round 3).  Two launches on ONE slot in flight at the same time would still disturb each other: that takes
// kSkinnyTicketSlots launches of this kernel in flight at once.
constexpr int kSkinnyTicketSlots = 1024;
__device__ unsigned long long g_skinny_ticket[kSkinnyTicketSlots];
// the launches' sequence numbers: ONE counter for the whole library (a function-local static of the templated launcher would be
// one per storage dtype, and an f32 and a bf16 launch would then share a (slot, epoch) pair)
static std::atomic<unsigned long long> g_skinny_next_launch{0};

template <typename T, int NG>
__global__ __launch_bounds__(kSkinnyThreads) void sim_skinny_kernel(
    const T* __restrict__ q, const T* __restrict__ p, int64_t Q, int64_t P, int64_t d, float temperature,
    int scale, int do_stats, int64_t group, T* __restrict__ scores, float2* __restrict__ partial,
    float* __restrict__ lse_out, float* __restrict__ loss_out, int ticket_slot, unsigned ticket_epoch) {
    typedef typename Mma<T>::Frag Frag;
    constexpr int CE = 16 / (int)sizeof(T);   // elements per chunk
    constexpr int SE = 4 * CE;                // elements per 64-byte K segment
    constexpr int U = kSkinnyUnroll;
    __shared__ float4_t s_acc[kSkinnyWaves][NG][64];
    __shared__ float2 s_part[kSkinnyMaxFusedGroups][64];
    __shared__ float s_tgt[64];
    __shared__ float s_red[kSkinnyWaves];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, frow = lane & 15;
    const int ngroups = (int)((P + 15) / 16);              // passage groups
    const int npchunk = (ngroups + NG - 1) / NG;
    const int npass = npchunk * (int)((Q + 15) / 16);
    const bool fused = gridDim.x == 1 && do_stats;
    const int64_t nseg = (d + SE - 1) / SE;
    for (int pass = (int)blockIdx.x; pass < npass; pass += (int)gridDim.x) {
        const int qg = pass / npchunk, pg0 = (pass % npchunk) * NG;
        const int64_t qr = 16 * qg + frow;
        const bool bv = qr < Q;
        const T* b_row = q + (bv ? qr : 0) * d + g * CE;
        const T* a_row[NG];
        bool av[NG];
#pragma unroll
        for (int n = 0; n < NG; ++n) {
            const int64_t pr = (int64_t)(pg0 + n) * 16 + frow;
            av[n] = pr < P;
            a_row[n] = p + (av[n] ? pr : 0) * d + g * CE;
        }
        float4_t acc[NG];
#pragma unroll
        for (int n = 0; n < NG; ++n) acc[n] = float4_t{0.f, 0.f, 0.f, 0.f};
        for (int64_t s0 = wave; s0 < nseg; s0 += kSkinnyWaves * U) {
            Frag a[U][NG], b[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {          // all loads of the batch first: one memory round trip per batch
                const int64_t k = (s0 + (int64_t)u * kSkinnyWaves) * SE;
                const bool kv = k + g * CE < d;       // d % CE == 0 is guaranteed by the host
#pragma unroll
                for (int n = 0; n < NG; ++n) {
                    a[u][n] = Mma<T>::zero();
                    if (av[n] && kv) a[u][n] = *reinterpret_cast<const Frag*>(a_row[n] + k);
                }
                b[u] = Mma<T>::zero();
                if (bv && kv) b[u] = *reinterpret_cast<const Frag*>(b_row + k);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int n = 0; n < NG; ++n) Mma<T>::mma(a[u][n], b[u], acc[n]);
        }
#pragma unroll
        for (int n = 0; n < NG; ++n) s_acc[wave][n][lane] = acc[n];
        __syncthreads();
        if (wave < NG) {                              // wave t finishes the tile of passage group pg0 + t
            float4_t t = s_acc[0][wave][lane];
#pragma unroll
            for (int w = 1; w < kSkinnyWaves; ++w) t += s_acc[w][wave][lane];      // fixed order: wave 0 + 1 + ... + 7
            skinny_tile_epilogue<T>(t, pg0 + wave, ngroups, qr, bv, g, Q, P, temperature, scale, do_stats, fused, group, scores,
                                    partial, s_part, s_tgt);
        }
        __syncthreads();                              // s_acc is rewritten by the next pass
    }
    if (fused) {
        skinny_fused_finalize(tid, Q, ngroups, s_part, s_tgt, s_red, lse_out, loss_out);
        return;
    }
    if (!do_stats || ticket_slot < 0) return;
    // Several blocks, still ONE launch (the W = 8 scoring shape, 64 x 384): every block publishes its partials and scores,
    // then takes a ticket; the block that arrives last merges the per-group partials of each row in group order -- the order
    // ce_finalize_kernel uses, so lse and loss are bit-identical to the two-launch form -- and writes lse / loss.
    __shared__ int s_last;
    __syncthreads();                                  // the block's stores happen-before thread 0's release below
    if (tid == 0) {
        unsigned long long* ticket = g_skinny_ticket + ticket_slot;
        // release: the block's stores (ordered before this by the barrier) are visible to whoever reads the ticket; acquire: the
        // last arriver sees every other block's.  (A compare-and-swap loop instead of one fetch-add: the stale-epoch case has to
        // replace the word, not add to it; the loop retries only when another block arrived in between.)
        unsigned long long seen = __hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), want;
        do {
            const unsigned arrived = (unsigned)(seen >> 32) == ticket_epoch ? (unsigned)seen : 0u;
            want = ((unsigned long long)ticket_epoch << 32) | (arrived + 1u);
        } while (!__hip_atomic_compare_exchange_strong(ticket, &seen, want, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT));
        s_last = ((unsigned)want == gridDim.x);
    }
    __syncthreads();
    if (!s_last) return;
    float rowloss = 0.f;
    if (tid < Q) {
        float m = RPO_NEG_INF, l = 0.f;
        const unsigned long long* pw = reinterpret_cast<const unsigned long long*>(partial);
        for (int b2 = 0; b2 < ngroups; ++b2) {
            const unsigned long long w = __hip_atomic_load(pw + (int64_t)b2 * Q + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            softmax_merge(m, l, __uint_as_float((unsigned)w), __uint_as_float((unsigned)(w >> 32)));
        }
        const float lse = m + logf(l);
        lse_out[tid] = lse;
        // the positive's score as stored (other blocks wrote it: read past this CU's caches)
        const T* sp = scores + (int64_t)tid * P + (int64_t)tid * group;
        float tgt;
        if constexpr (sizeof(T) == 2) {
            const unsigned short w = __hip_atomic_load(reinterpret_cast<const unsigned short*>(sp), __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT);
            tgt = Elem<T>::ld(reinterpret_cast<const T*>(&w));
        } else {
            tgt = __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(sp), __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT));
        }
        rowloss = lse - tgt;
    }
    const float tot = block_sum<8>(rowloss, s_red);
    if (tid == 0) loss_out[0] = tot / (float)Q;
}

// ------------------------------------------------------------------------------------------------
// Small kernel: the reference's single-GPU shapes (Q <= 16, P <= 128, e.g. 8 x 48 x 2048) as ONE block with COALESCED loads.
//   Why: the skinny kernel's fragment-shaped loads (16 rows x 64 B per wave instruction) bring one CU ~35 GB/s, so its
//   one-block launch spends 6.4 of its ~8 us loading 224 KB.  Here every wave instruction loads 1 KiB of ONE row
//   (whole 128-byte lines), all (P + Q) rows x 2 K-halves are in flight at once (<= 32 x 16 B per lane), and the MFMA
//   fragments come out of an LDS image: rows of KCB + 32 bytes (the 32-byte skew puts the 16 rows of a ds_read_b128
//   fragment on distinct bank groups), queries in rows 0..15, passage group pg in rows 16 + 16 pg ...  The row length is
//   processed in two chunks of KCB bytes because (P + 16) x 4 KiB exceeds LDS.  Rows / K-tails that do not exist are
//   zero (K-tail) or never leave their own output row / column (missing rows), which is masked at the stores.
//   K is split over the 8 waves per 64-byte segment as in the skinny kernel; the 8 partial tiles per passage group are
//   summed through LDS in wave order, then wave t finishes group t and the block finalizes lse / loss.
// ------------------------------------------------------------------------------------------------
constexpr int kSmallThreads = 512, kSmallMaxGroups = 8, kSmallQPieces = 4, kSmallPPieces = 14;   // pieces per wave and chunk

template <typename T>
__global__ __launch_bounds__(kSmallThreads) void sim_small_kernel(
    const T* __restrict__ q, const T* __restrict__ p, int64_t Q, int64_t P, int64_t d, float temperature, int scale,
    int do_stats, int64_t group, T* __restrict__ scores, float* __restrict__ lse_out, float* __restrict__ loss_out,
    int kcb, int stage_bytes) {
    typedef typename Mma<T>::Frag Frag;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    __shared__ float2 s_part[kSmallMaxGroups][64];
    __shared__ float s_tgt[64];
    __shared__ float s_red[8];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, frow = lane & 15;
    const int ngroups = (int)((P + 15) / 16);
    const int row_bytes = (int)(d * (int64_t)sizeof(T));
    const int ldsrow = kcb + 32;
    // A chunk of a row is 1 << pshift pieces of 1 KiB.  Wave w owns piece j = w & pmask of rows (w >> pshift) + i * rstep:
    // per load only an offset increment is left (the first version recomputed row / piece / base per load: ~55
    // instructions each, 1750 before the first wait).  Offsets are 32-bit against the uniform q / p base pointers.
    const int pshift = 31 - __builtin_clz((unsigned)(kcb >> 10));
    const int pmask = (1 << pshift) - 1;
    const int rstep = 8 >> pshift, row0 = wave >> pshift;
    const unsigned jbyte = (unsigned)((wave & pmask) * 1024 + lane * 16);
    const unsigned rb = (unsigned)row_bytes, gstep = (unsigned)rstep * rb, lstep = (unsigned)(rstep * ldsrow);
    const unsigned char* qb = reinterpret_cast<const unsigned char*>(q);
    const unsigned char* pb = reinterpret_cast<const unsigned char*>(p);
    constexpr int NPQ = kSmallQPieces, NPP = kSmallPPieces;
    uint4_t bq[2][NPQ], bp[2][NPP];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int i = 0; i < NPQ; ++i) bq[c][i] = uint4_t{0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < NPP; ++i) bp[c][i] = uint4_t{0u, 0u, 0u, 0u};
        const unsigned koff = (unsigned)(c * kcb) + jbyte;
        if (koff < rb) {                                   // the K tail of a short row stays zero
            unsigned off = (unsigned)row0 * rb + koff;
#pragma unroll
            for (int i = 0; i < NPQ; ++i, off += gstep)
                if (row0 + i * rstep < (int)Q) bq[c][i] = *reinterpret_cast<const uint4_t*>(qb + off);
            off = (unsigned)row0 * rb + koff;
#pragma unroll
            for (int i = 0; i < NPP; ++i, off += gstep)
                if (row0 + i * rstep < (int)P) bp[c][i] = *reinterpret_cast<const uint4_t*>(pb + off);
        }
    }
    float4_t acc[kSmallMaxGroups];
#pragma unroll
    for (int n = 0; n < kSmallMaxGroups; ++n) acc[n] = float4_t{0.f, 0.f, 0.f, 0.f};
    const int nseg = kcb >> 6;                            // 64-byte K segments per chunk
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        if (c * kcb >= row_bytes) break;                   // short rows fit one chunk
        unsigned lo = (unsigned)(row0 * ldsrow) + jbyte;
#pragma unroll
        for (int i = 0; i < NPQ; ++i, lo += lstep)
            if (row0 + i * rstep < (int)Q) *reinterpret_cast<uint4_t*>(s_raw + lo) = bq[c][i];
        lo = (unsigned)((16 + row0) * ldsrow) + jbyte;
#pragma unroll
        for (int i = 0; i < NPP; ++i, lo += lstep)
            if (row0 + i * rstep < (int)P) *reinterpret_cast<uint4_t*>(s_raw + lo) = bp[c][i];
        __syncthreads();
        for (int sgm = wave; sgm < nseg; sgm += 8) {
            const int koff = sgm * 64 + g * 16;
            const Frag b = *reinterpret_cast<const Frag*>(s_raw + frow * ldsrow + koff);
#pragma unroll
            for (int n = 0; n < kSmallMaxGroups; ++n)
                if (n < ngroups) {
                    const Frag a = *reinterpret_cast<const Frag*>(s_raw + (16 + 16 * n + frow) * ldsrow + koff);
                    Mma<T>::mma(a, b, acc[n]);
                }
        }
        __syncthreads();                                  // the image is overwritten by the next chunk / by the partial sums
    }
    float4_t (*s_acc)[kSmallMaxGroups][64] = reinterpret_cast<float4_t (*)[kSmallMaxGroups][64]>(s_raw);   // [8][8][64]
#pragma unroll
    for (int n = 0; n < kSmallMaxGroups; ++n)
        if (n < ngroups) s_acc[wave][n][lane] = acc[n];
    __syncthreads();
    const bool fused = do_stats != 0;
    if (wave < ngroups) {
        float4_t t = s_acc[0][wave][lane];
#pragma unroll
        for (int w = 1; w < 8; ++w) t += s_acc[w][wave][lane];          // fixed order: wave 0 + 1 + ... + 7
        skinny_tile_epilogue<T>(t, wave, ngroups, frow, frow < Q, g, Q, P, temperature, scale, do_stats, fused, group, scores,
                                nullptr, s_part, s_tgt);
    }
    if (!fused) return;
    __syncthreads();
    skinny_fused_finalize(tid, Q, ngroups, s_part, s_tgt, s_red, lse_out, loss_out);
}

// ------------------------------------------------------------------------------------------------
// Row-wise kernel: any d, any alignment.  grid = Q, 4 waves stride over the passages.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void sim_rowwise_kernel(const T* __restrict__ q, const T* __restrict__ p,
                                                           int64_t Q, int64_t P, int64_t d, float temperature,
                                                           int scale, int do_stats, T* __restrict__ scores,
                                                           float2* __restrict__ partial) {
    __shared__ float2 s_part[4];
    const int64_t i = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const T* qi = q + i * d;
    float mx = RPO_NEG_INF, sum = 0.f;
    for (int64_t j = wave; j < P; j += 4) {
        const T* pj = p + j * d;
        float a = 0.f;
        for (int64_t c = lane; c < d; c += 64) a = fmaf(Elem<T>::ld(qi + c), Elem<T>::ld(pj + c), a);
        a = wave_sum(a);
        const float v = finish_score<T>(a, temperature, 1.0f / temperature, scale);
        if (lane == 0) Elem<T>::st(scores + i * P + j, v);
        softmax_merge(mx, sum, v, 1.0f);
    }
    if (!do_stats) return;
    if (lane == 0) s_part[wave] = make_float2(mx, sum);
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = s_part[0].x, l = s_part[0].y;
        for (int w = 1; w < 4; ++w) softmax_merge(m, l, s_part[w].x, s_part[w].y);
        partial[i] = make_float2(m, l);
    }
}

// ------------------------------------------------------------------------------------------------
// CE finalize: thread per score row.  partial is [nPb][Q].  loss = mean_i (lse_i - S[i, i * group]).
// ------------------------------------------------------------------------------------------------
constexpr int kFinThreads = 256;

template <typename T>
__global__ __launch_bounds__(kFinThreads) void ce_finalize_kernel(const float2* __restrict__ partial,
                                                                   const T* __restrict__ scores, int64_t Q,
                                                                   int64_t P, int nPb, int64_t group,
                                                                   float* __restrict__ lse_out,
                                                                   float* __restrict__ loss_out,
                                                                   float* __restrict__ blocksum,
                                                                   unsigned* __restrict__ ticket) {
    __shared__ float s_red[kFinThreads / 64];
    __shared__ int s_last;
    const int64_t i = (int64_t)blockIdx.x * kFinThreads + threadIdx.x;
    float rowloss = 0.f;
    if (i < Q) {
        float m = RPO_NEG_INF, l = 0.f;
        for (int b = 0; b < nPb; ++b) {
            const float2 o = partial[(int64_t)b * Q + i];
            softmax_merge(m, l, o.x, o.y);
        }
        const float lse = m + logf(l);
        lse_out[i] = lse;
        rowloss = lse - Elem<T>::ld(scores + i * P + i * group);
    }
    const float bs = block_sum<kFinThreads / 64>(rowloss, s_red);
    if (gridDim.x == 1) {
        if (threadIdx.x == 0) loss_out[0] = bs / (float)Q;
        return;
    }
    // several blocks: publish the block sum, take a ticket; the last arriver adds them up in index order.
    if (threadIdx.x == 0) {
        blocksum[blockIdx.x] = bs;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == gridDim.x - 1);
        if (s_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            float tot = 0.f;
            for (unsigned b = 0; b < gridDim.x; ++b)
                tot += __hip_atomic_load(blocksum + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            loss_out[0] = tot / (float)Q;
        }
    }
}

// RPO_TARGET_FIRST: raw [B, G] dots -> scores, lse, loss.  One block, thread per row.
template <typename T>
__global__ __launch_bounds__(kFinThreads) void first_finalize_kernel(const float* __restrict__ raw, int64_t B,
                                                                      int64_t G, float temperature, int scale,
                                                                      T* __restrict__ scores,
                                                                      float* __restrict__ lse_out,
                                                                      float* __restrict__ loss_out) {
    __shared__ float s_red[kFinThreads / 64];
    float rowloss = 0.f;
    for (int64_t b = threadIdx.x; b < B; b += kFinThreads) {
        float m = RPO_NEG_INF, l = 0.f, s0 = 0.f;
        for (int64_t gi = 0; gi < G; ++gi) {
            const float v = finish_score<T>(raw[b * G + gi], temperature, 1.0f / temperature, scale);
            Elem<T>::st(scores + b * G + gi, v);
            if (gi == 0) s0 = v;
            softmax_merge(m, l, v, 1.0f);
        }
        if (lse_out) {
            const float lse = m + logf(l);
            lse_out[b] = lse;
            rowloss += lse - s0;
        }
    }
    if (loss_out) {
        const float tot = block_sum<kFinThreads / 64>(rowloss, s_red);
        if (threadIdx.x == 0) loss_out[0] = tot / (float)B;
    }
}

// ------------------------------------------------------------------------------------------------
// Backward, VALU form (one wave per output row x column chunk).
//   blocks [0, q_rows*nchunk): dq row;  blocks [q_rows*nchunk, +p_rows*nchunk): dp row.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void infonce_bwd_valu_kernel(
    const T* __restrict__ q, const T* __restrict__ p, const T* __restrict__ scores, const float* __restrict__ lse,
    const float* __restrict__ grad_loss, int64_t Q, int64_t P, int64_t d, float temperature, int64_t group,
    int64_t q_row0, int64_t q_rows, int64_t p_row0, int64_t p_rows, T* __restrict__ dq, T* __restrict__ dp,
    int nchunk) {
    // block = one output row x one chunk of 64 x V columns; its 4 waves split the reduction range (the P passages of a dq
    // row, the Q queries of a dp row) into contiguous quarters and each wave issues the row loads of 8 terms before their
    // FMAs, so a row costs a few memory round trips instead of one per term (round 1: one wave per row walking 48
    // dependent load -> FMA steps, 19 us at 8 x 48 x 2048).  The four partial sums are added in wave order through LDS.
    constexpr int V = Elem<T>::kVec;
    constexpr int U = 8;
    __shared__ float s_sum[3][64][V];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float coef = grad_loss[0] / ((float)Q * temperature);
    int64_t b = blockIdx.x;
    const int64_t ndq = dq ? q_rows * nchunk : 0;
    const bool is_dq = b < ndq;
    if (!is_dq) b -= ndq;
    const int64_t row = b / nchunk;
    const int chunk = (int)(b % nchunk);
    float accv[V];
#pragma unroll
    for (int k = 0; k < V; ++k) accv[k] = 0.f;
    const int64_t c0 = ((int64_t)chunk * 64 + lane) * V;
    const bool cv = c0 < d;
    // dq row gi:  sum_j w[gi, j] p_j   ;   dp row gj:  sum_i w[i, gj] q_i     (w = coef (softmax(S) - onehot))
    const int64_t R = is_dq ? P : Q;                       // reduction length
    const int64_t per = (R + 3) / 4;
    const int64_t r_begin = wave * per, r_end = min(R, r_begin + per);
    const int64_t gi = q_row0 + row, gj = p_row0 + row;
    const T* src = is_dq ? p : q;
    for (int64_t r0 = r_begin; r0 < r_end; r0 += 64) {
        const int64_t r = r0 + lane;
        float w = 0.f;
        if (r < r_end) {
            if (is_dq) w = coef * (__expf(Elem<T>::ld(scores + gi * P + r) - lse[gi]) - (r == gi * group ? 1.f : 0.f));
            else       w = coef * (__expf(Elem<T>::ld(scores + r * P + gj) - lse[r]) - (gj == r * group ? 1.f : 0.f));
        }
        const int rn = (int)min((int64_t)64, r_end - r0);  // wave-uniform
        for (int t0 = 0; t0 < rn; t0 += U) {
            Vec16<T> x[U];
            float wt[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {                  // loads of the batch first
                const int t = min(t0 + u, rn - 1);         // clamped: a repeated row gets weight 0 below
                wt[u] = (t0 + u < rn) ? __shfl(w, t, 64) : 0.f;
                if (cv) x[u].load(src + (r0 + t) * d + c0);
            }
            if (cv) {
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int k = 0; k < V; ++k) accv[k] = fmaf(wt[u], x[u].v[k], accv[k]);
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int k = 0; k < V; ++k) s_sum[wave - 1][lane][k] = accv[k];
    }
    __syncthreads();
    if (wave != 0 || !cv) return;
    Vec16<T> o;
#pragma unroll
    for (int k = 0; k < V; ++k) o.v[k] = ((accv[k] + s_sum[0][lane][k]) + s_sum[1][lane][k]) + s_sum[2][lane][k];
    o.store((is_dq ? dq : dp) + row * d + c0);
}

// Scalar backward for shapes the vector form cannot take (d % V != 0 or unaligned): thread per column.
template <typename T>
__global__ __launch_bounds__(64) void infonce_bwd_scalar_kernel(
    const T* __restrict__ q, const T* __restrict__ p, const T* __restrict__ scores, const float* __restrict__ lse,
    const float* __restrict__ grad_loss, int64_t Q, int64_t P, int64_t d, float temperature, int64_t group,
    int64_t q_row0, int64_t q_rows, int64_t p_row0, int64_t p_rows, T* __restrict__ dq, T* __restrict__ dp,
    int nchunk) {
    const float coef = grad_loss[0] / ((float)Q * temperature);
    int64_t b = blockIdx.x;
    const int64_t ndq = dq ? q_rows * nchunk : 0;
    const bool is_dq = b < ndq;
    if (!is_dq) b -= ndq;
    const int64_t row = b / nchunk;
    const int64_t c = (b % nchunk) * 64 + threadIdx.x;
    if (c >= d) return;
    float acc = 0.f;
    if (is_dq) {
        const int64_t gi = q_row0 + row;
        const float lse_i = lse[gi];
        for (int64_t j = 0; j < P; ++j) {
            const float w = coef * (__expf(Elem<T>::ld(scores + gi * P + j) - lse_i) - (j == gi * group ? 1.f : 0.f));
            acc = fmaf(w, Elem<T>::ld(p + j * d + c), acc);
        }
        Elem<T>::st(dq + row * d + c, acc);
    } else {
        const int64_t gj = p_row0 + row;
        for (int64_t i = 0; i < Q; ++i) {
            const float w = coef * (__expf(Elem<T>::ld(scores + i * P + gj) - lse[i]) - (gj == i * group ? 1.f : 0.f));
            acc = fmaf(w, Elem<T>::ld(q + i * d + c), acc);
        }
        Elem<T>::st(dp + row * d + c, acc);
    }
}

// dS for the GEMM form of the backward (large Q*P): dS[i,j] = coef (exp(S[i,j] - lse_i) - [j == i*group]).
//   ds  [q_rows, P] = dS[q_row0 + r, :]        (row-major: the A operand of dq = dS p)
//   dst [p_rows, Q] = dS[:, p_row0 + r]^T      (row-major: the A operand of dp = dS^T q)
// 64x64 tiles over the (row, col) range that is needed; the transposed copy goes through LDS so that both outputs
// are written with coalesced rows.  grid = (ceil(P/64), ceil(Q/64)).
template <typename T>
__global__ __launch_bounds__(256) void infonce_ds_kernel(const T* __restrict__ scores, const float* __restrict__ lse,
                                                         const float* __restrict__ grad_loss, int64_t Q, int64_t P,
                                                         float temperature, int64_t group, int64_t q_row0,
                                                         int64_t q_rows, int64_t p_row0, int64_t p_rows,
                                                         T* __restrict__ ds, T* __restrict__ dst) {
    __shared__ float tile[64][65];
    const int64_t i0 = (int64_t)blockIdx.y * 64, j0 = (int64_t)blockIdx.x * 64;
    const bool need_ds = ds && i0 < q_row0 + q_rows && i0 + 64 > q_row0;
    const bool need_dst = dst && j0 < p_row0 + p_rows && j0 + 64 > p_row0;
    if (!need_ds && !need_dst) return;
    const float coef = grad_loss[0] / ((float)Q * temperature);
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 columns x 4 row phases
    for (int r = ty; r < 64; r += 4) {
        const int64_t i = i0 + r, j = j0 + tx;
        float w = 0.f;
        if (i < Q && j < P) w = coef * (__expf(Elem<T>::ld(scores + i * P + j) - lse[i]) - (j == i * group ? 1.f : 0.f));
        tile[r][tx] = w;
        if (need_ds && i < Q && j < P && i >= q_row0 && i < q_row0 + q_rows) Elem<T>::st(ds + (i - q_row0) * P + j, w);
    }
    if (!need_dst) return;
    __syncthreads();
    for (int c = ty; c < 64; c += 4) {            // output row = passage column j0 + c, output col = query i0 + tx
        const int64_t j = j0 + c, i = i0 + tx;
        if (j < P && i < Q && j >= p_row0 && j < p_row0 + p_rows) Elem<T>::st(dst + (j - p_row0) * Q + i, tile[tx][c]);
    }
}

// RPO_TARGET_FIRST backward: grid = own q rows (b) ; w[b,g] = coef (softmax(s_b)[g] - [g == 0]).
//   dq_b = sum_g w[b,g] p_{bG+g}   ;   dp_{bG+g} = w[b,g] q_b   (only rows inside the own ranges are written)
template <typename T>
__global__ __launch_bounds__(256) void infonce_first_bwd_kernel(
    const T* __restrict__ q, const T* __restrict__ p, const T* __restrict__ scores, const float* __restrict__ lse,
    const float* __restrict__ grad_loss, int64_t B, int64_t G, int64_t d, float temperature, int64_t q_row0,
    int64_t q_rows, int64_t p_row0, int64_t p_rows, T* __restrict__ dq, T* __restrict__ dp) {
    const int64_t b = blockIdx.x;   // all B rows; a row may own its q, some of its p's, or neither
    const float coef = grad_loss[0] / ((float)B * temperature);
    const float lse_b = lse[b];
    const bool own_q = dq && b >= q_row0 && b < q_row0 + q_rows;
    for (int64_t c = threadIdx.x; c < d; c += 256) {
        float aq = 0.f;
        const float qv = Elem<T>::ld(q + b * d + c);
        for (int64_t g = 0; g < G; ++g) {
            const float w = coef * (__expf(Elem<T>::ld(scores + b * G + g) - lse_b) - (g == 0 ? 1.f : 0.f));
            const int64_t pj = b * G + g;
            aq = fmaf(w, Elem<T>::ld(p + pj * d + c), aq);
            if (dp && pj >= p_row0 && pj < p_row0 + p_rows) Elem<T>::st(dp + (pj - p_row0) * d + c, w * qv);
        }
        if (own_q) Elem<T>::st(dq + (b - q_row0) * d + c, aq);
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
enum FwdPath { PATH_TILE = 0, PATH_SKINNY = 1, PATH_ROWWISE = 2, PATH_TILE256 = 3 };

struct Plan {
    int path;
    int nPb;          // passage column blocks (partials per row)
    int nPt, nQt;     // tile grid
    int tp, tq;       // PATH_TILE: tile shape (passages x queries)
    int nFin;         // finalize blocks
    size_t off_partial, off_blocksum, off_raw, total;
};

template <typename T> constexpr int row_elems() { return kTileRowBytes / (int)sizeof(T); }

static Plan make_plan(int64_t Q, int64_t P, int64_t d, int dtype, bool aligned) {
    Plan pl{};
    const int es = rpo_elem_size(dtype);
    const int CE = 16 / es, KE = kTileRowBytes / es;
    const bool chunk_ok = aligned && (d % CE == 0);
    if (!chunk_ok) {
        pl.path = PATH_ROWWISE;
        pl.nPb = 1;
    } else if (Q <= 64 || (d % KE != 0)) {
        if (Q <= 64) {
            pl.path = PATH_SKINNY;
            pl.nPb = (int)rpo_cdiv(P, 16);
        } else {
            pl.path = PATH_ROWWISE;
            pl.nPb = 1;
        }
    } else {
        // the 256 x 256 kernel runs one block per CU: take it only when its grid can fill most of the 256 CUs
        const bool big = dtype == RPO_DT_BF16 && rpo_cdiv(P, kBigTile) * rpo_cdiv(Q, kBigTile) >= 192 &&
                         P * d * 2 < ((int64_t)1 << 32) && Q * d * 2 < ((int64_t)1 << 32);     // 32-bit piece offsets
        pl.path = big ? PATH_TILE256 : PATH_TILE;
        pl.tp = pl.tq = big ? kBigTile : kTileP;
        if (!big) {
            // the largest tile that still gives every CU two blocks (the kernel's occupancy); else the smallest.  The query side
            // shrinks first: a wave's row segments of the score matrix stay 64 passages = 128 bytes (bf16) long
            const int64_t want = 2 * 256;
            if (rpo_cdiv(P, 128) * rpo_cdiv(Q, 128) < want) {
                pl.tq = 64;
                if (rpo_cdiv(P, 128) * rpo_cdiv(Q, 64) < want) pl.tp = 64;
            }
        }
        pl.nPt = (int)rpo_cdiv(P, pl.tp);
        pl.nQt = (int)rpo_cdiv(Q, pl.tq);
        pl.nPb = pl.nPt;
    }
    pl.nFin = (int)rpo_cdiv(Q, kFinThreads);
    size_t off = 256;                                   // [0,16): ticket counter
    pl.off_partial = off;
    off += (size_t)pl.nPb * (size_t)Q * sizeof(float2);
    off = (off + 255) & ~(size_t)255;
    pl.off_blocksum = off;
    off += (size_t)pl.nFin * sizeof(float);
    off = (off + 255) & ~(size_t)255;
    pl.off_raw = off;                                   // RPO_TARGET_FIRST raw dots [Q, P/Q]
    off += (size_t)P * sizeof(float);
    pl.total = (off + 255) & ~(size_t)255;
    return pl;
}

template <typename T>
int fwd_impl(const void* q, const void* p, int64_t Q, int64_t P, int64_t d, float temperature, int target_mode,
             void* scores_out, float* lse_out, float* loss_out, void* ws, size_t ws_bytes, hipStream_t st) {
    const int dtype = rpo_dtype_of<T>();
    const bool do_stats = lse_out != nullptr;
    const int scale = temperature != 1.0f;
    const bool aligned = rpo_aligned16(q) && rpo_aligned16(p);
    Plan pl = make_plan(Q, P, d, dtype, aligned);
    if (do_stats || target_mode == RPO_TARGET_FIRST) {
        if (!ws || ws_bytes < pl.total || (reinterpret_cast<uintptr_t>(ws) & 255)) return RPO_ERR_WORKSPACE;
    }
    char* wsb = (char*)ws;
    if (target_mode == RPO_TARGET_FIRST) {
        const int64_t G = P / Q;
        float* raw = (float*)(wsb + pl.off_raw);
        int rc = rpo_launch_grouped_dots(q, p, Q, G, d, dtype, raw, st);
        if (rc != RPO_OK) return rc;
        RPO_LAUNCH(first_finalize_kernel<T>, dim3(1), dim3(kFinThreads), 0, st, raw, Q, G, temperature,
                           scale, (T*)scores_out, lse_out, loss_out);
        return rpo_launch_status();
    }
    float2* partial = do_stats ? (float2*)(wsb + pl.off_partial) : nullptr;
    bool fused_finalize = false;      // the forward kernel wrote lse and loss itself (single-block skinny launch)
    if (pl.path == PATH_TILE) {
        static bool attr_set = false;   // idempotent; a race only repeats the same call
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)sim_tile_kernel<T, 128, 128, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      tile_lds_bytes(128, 128, 2));
            (void)hipFuncSetAttribute((const void*)sim_tile_kernel<T, 128, 64, 3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      tile_lds_bytes(128, 64, 3));
            (void)hipFuncSetAttribute((const void*)sim_tile_kernel<T, 64, 64, 8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      tile_lds_bytes(64, 64, 8));
            attr_set = true;
        }
        const dim3 grid((unsigned)(pl.nPt * pl.nQt)), block(kTileThreads);
        if (pl.tp == 128 && pl.tq == 128)
            RPO_LAUNCH((sim_tile_kernel<T, 128, 128, 2>), grid, block, tile_lds_bytes(128, 128, 2), st, (const T*)q, (const T*)p, Q, P,
                       d, temperature, scale, do_stats ? 1 : 0, (T*)scores_out, partial, pl.nPt, pl.nQt);
        else if (pl.tp == 128)
            RPO_LAUNCH((sim_tile_kernel<T, 128, 64, 3>), grid, block, tile_lds_bytes(128, 64, 3), st, (const T*)q, (const T*)p, Q, P,
                       d, temperature, scale, do_stats ? 1 : 0, (T*)scores_out, partial, pl.nPt, pl.nQt);
        else
            RPO_LAUNCH((sim_tile_kernel<T, 64, 64, 8>), grid, block, tile_lds_bytes(64, 64, 8), st, (const T*)q, (const T*)p, Q, P, d,
                       temperature, scale, do_stats ? 1 : 0, (T*)scores_out, partial, pl.nPt, pl.nQt);
    } else if (pl.path == PATH_TILE256) {
        if constexpr (__is_same(T, bf16_t)) {          // make_plan picks this path for bf16 only
            static bool attr_set256 = false;
            if (!attr_set256) {
                (void)hipFuncSetAttribute((const void*)sim_tile256_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          kBigLdsBytes);
                attr_set256 = true;
            }
#ifndef RPO_SIM_PERSIST
#define RPO_SIM_PERSIST 1       // 0: one block per tile (rounds 1-4), the A/B arm of profiles/r05_sim_tile256_persistent_ab.txt
#endif
            const int64_t ntile = (int64_t)pl.nPt * pl.nQt;
            RPO_LAUNCH(sim_tile256_kernel<0>, dim3((unsigned)(RPO_SIM_PERSIST ? std::min<int64_t>(ntile, kBigPersistBlocks) : ntile)),
                       dim3(kBigThreads), kBigLdsBytes, st, (const bf16_t*)q, (const bf16_t*)p, Q, P, d, d, d, P, temperature, scale,
                       do_stats ? 1 : 0, (bf16_t*)scores_out, partial, pl.nPt, pl.nQt, /*stagger=*/1, /*dbg=*/0, SimFilter{});
        }
    } else if (pl.path == PATH_SKINNY) {
        const int ng = Q <= 16 ? 4 : 1;
        // the reference's single-GPU shapes: one block, coalesced loads through an LDS image (sim_small_kernel)
        {
            const int64_t row_bytes = d * (int64_t)sizeof(T);
            int64_t kcb = 1024;                                   // bytes of a row per chunk: a power-of-two number of KiB
            while (2 * kcb < row_bytes) kcb *= 2;
            const int64_t stage = (16 + 16 * (int64_t)pl.nPb) * (kcb + 32);
            const int64_t accb = 8 * (int64_t)kSmallMaxGroups * 64 * 16;
            const int64_t lds = stage > accb ? stage : accb;
            if (Q <= 16 && pl.nPb <= kSmallMaxGroups && kcb <= 8192 && Q * (kcb / 1024) <= 8 * kSmallQPieces &&
                P * (kcb / 1024) <= 8 * kSmallPPieces && lds <= 150 * 1024 && row_bytes % 16 == 0) {
                static bool attr_set_small = false;
                if (!attr_set_small) {
                    (void)hipFuncSetAttribute((const void*)sim_small_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                              150 * 1024);
                    attr_set_small = true;
                }
                RPO_LAUNCH(sim_small_kernel<T>, dim3(1), dim3(kSmallThreads), (size_t)lds, st, (const T*)q, (const T*)p, Q, P, d,
                           temperature, scale, do_stats ? 1 : 0, P / Q, (T*)scores_out, lse_out, loss_out, (int)kcb, (int)lds);
                return rpo_launch_status();
            }
        }
        const int64_t npass = rpo_cdiv(pl.nPb, ng) * rpo_cdiv(Q, 16);
        // small problems (every single-GPU shape of the reference: 8 x 48 x 2048 bf16 is 224 KB) run as ONE block that
        // walks all passes and finishes lse / loss itself: a second launch costs more than the one CU loses in bandwidth
        const int64_t bytes = (Q + P) * d * (int64_t)sizeof(T);
        const bool one_block = pl.nPb <= kSkinnyMaxFusedGroups && bytes <= 384 * 1024;
        const unsigned nblk = one_block ? 1u : (unsigned)npass;
        // several blocks: the last block to arrive finalizes (one launch; needs the scores in memory for the positives' column)
        int slot = -1;
        unsigned epoch = 0;
        if (nblk > 1 && do_stats && scores_out != nullptr) {
            const unsigned long long seq = g_skinny_next_launch.fetch_add(1, std::memory_order_relaxed);
            slot = (int)(seq % (unsigned)kSkinnyTicketSlots);
            epoch = (unsigned)(seq / (unsigned)kSkinnyTicketSlots) + 1u;       // never 0: a zero-initialised slot matches no launch
        }
        fused_finalize = do_stats && (nblk == 1 || slot >= 0);
        const dim3 grid(nblk), block(kSkinnyThreads);
        if (ng == 4)
            RPO_LAUNCH((sim_skinny_kernel<T, 4>), grid, block, 0, st, (const T*)q, (const T*)p, Q, P, d, temperature, scale,
                       do_stats ? 1 : 0, P / Q, (T*)scores_out, partial, lse_out, loss_out, slot, epoch);
        else
            RPO_LAUNCH((sim_skinny_kernel<T, 1>), grid, block, 0, st, (const T*)q, (const T*)p, Q, P, d, temperature, scale,
                       do_stats ? 1 : 0, P / Q, (T*)scores_out, partial, lse_out, loss_out, slot, epoch);
    } else {
        RPO_LAUNCH(sim_rowwise_kernel<T>, dim3((unsigned)Q), dim3(256), 0, st, (const T*)q, (const T*)p, Q,
                           P, d, temperature, scale, do_stats ? 1 : 0, (T*)scores_out, partial);
    }
    int rc = rpo_launch_status();
    if (rc != RPO_OK || !do_stats || fused_finalize) return rc;
    unsigned* ticket = (unsigned*)wsb;
    if (pl.nFin > 1) (void)hipMemsetAsync(ticket, 0, 16, st);
    RPO_LAUNCH(ce_finalize_kernel<T>, dim3((unsigned)pl.nFin), dim3(kFinThreads), 0, st, partial,
                       (const T*)scores_out, Q, P, pl.nPb, P / Q, lse_out, loss_out,
                       (float*)(wsb + pl.off_blocksum), ticket);
    return rpo_launch_status();
}

template <typename T>
int bwd_impl(const void* q, const void* p, const void* scores, const float* lse, const float* grad_loss, int64_t Q,
             int64_t P, int64_t d, float temperature, int target_mode, int64_t q_row0, int64_t q_rows,
             int64_t p_row0, int64_t p_rows, void* dq, void* dp, hipStream_t st) {
    if (target_mode == RPO_TARGET_FIRST) {
        RPO_LAUNCH(infonce_first_bwd_kernel<T>, dim3((unsigned)Q), dim3(256), 0, st, (const T*)q,
                           (const T*)p, (const T*)scores, lse, grad_loss, Q, P / Q, d, temperature, q_row0, q_rows,
                           p_row0, p_rows, (T*)dq, (T*)dp);
        return rpo_launch_status();
    }
    constexpr int V = Elem<T>::kVec;
    const bool vec = (d % V == 0) && rpo_aligned16(q) && rpo_aligned16(p) && (!dq || rpo_aligned16(dq)) &&
                     (!dp || rpo_aligned16(dp));
    const int64_t group = P / Q;
    if (vec) {
        const int nchunk = (int)rpo_cdiv(d, 64 * V);
        const int64_t blocks = ((dq ? q_rows : 0) + (dp ? p_rows : 0)) * nchunk;
        if (blocks <= 0) return RPO_OK;
        RPO_LAUNCH(infonce_bwd_valu_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, st, (const T*)q,
                           (const T*)p, (const T*)scores, lse, grad_loss, Q, P, d, temperature, group, q_row0,
                           q_rows, p_row0, p_rows, (T*)dq, (T*)dp, nchunk);
    } else {
        const int nchunk = (int)rpo_cdiv(d, 64);
        const int64_t blocks = ((dq ? q_rows : 0) + (dp ? p_rows : 0)) * nchunk;
        if (blocks <= 0) return RPO_OK;
        RPO_LAUNCH(infonce_bwd_scalar_kernel<T>, dim3((unsigned)blocks), dim3(64), 0, st, (const T*)q,
                           (const T*)p, (const T*)scores, lse, grad_loss, Q, P, d, temperature, group, q_row0,
                           q_rows, p_row0, p_rows, (T*)dq, (T*)dp, nchunk);
    }
    return rpo_launch_status();
}

static int check_common(const void* q, const void* p, int64_t Q, int64_t P, int64_t d, int dtype, float temperature,
                        int target_mode) {
    if (!q || !p || Q <= 0 || P <= 0 || d <= 0) return RPO_ERR_INVALID_ARG;
    if (!rpo_dtype_ok(dtype)) return RPO_ERR_INVALID_ARG;
    if (target_mode != RPO_TARGET_INBATCH && target_mode != RPO_TARGET_FIRST) return RPO_ERR_INVALID_ARG;
    if (!(temperature > 0.f)) return RPO_ERR_INVALID_ARG;
    if (target_mode == RPO_TARGET_FIRST && P % Q != 0) return RPO_ERR_INVALID_ARG;  // .view(Q, G, -1)
    if (Q > INT32_MAX || P > INT32_MAX || d > INT32_MAX) return RPO_ERR_UNSUPPORTED;
    if (rpo_cdiv(P, 64) * rpo_cdiv(Q, 64) > INT32_MAX) return RPO_ERR_UNSUPPORTED;
    return RPO_OK;
}

}  // namespace

namespace {
__global__ void skinny_ticket_poison_kernel(unsigned long long word) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < kSkinnyTicketSlots) g_skinny_ticket[i] = word;
}
}  // namespace

extern "C" int rpo_infonce_debug_poison_tickets(uint64_t word, rpo_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    RPO_LAUNCH(skinny_ticket_poison_kernel, dim3((kSkinnyTicketSlots + 255) / 256), dim3(256), 0, st, (unsigned long long)word);
    return rpo_launch_status();
}

extern "C" size_t rpo_infonce_workspace_bytes(int64_t Q, int64_t P, int64_t d, int dtype) {
    if (Q <= 0 || P <= 0 || d <= 0) return 0;
    // sized for the path with the most partials so that the answer does not depend on pointer alignment
    Plan a = make_plan(Q, P, d, dtype, true), b = make_plan(Q, P, d, dtype, false);
    return a.total > b.total ? a.total : b.total;
}

extern "C" int rpo_infonce_fwd(const void* q, const void* p, int64_t Q, int64_t P, int64_t d, int dtype,
                               float temperature, int target_mode, void* scores_out, float* lse_out,
                               float* loss_out, void* workspace, size_t workspace_bytes, rpo_stream_t stream) {
    int rc = check_common(q, p, Q, P, d, dtype, temperature, target_mode);
    if (rc != RPO_OK) return rc;
    if (!scores_out) return RPO_ERR_INVALID_ARG;
    if ((lse_out == nullptr) != (loss_out == nullptr)) return RPO_ERR_INVALID_ARG;
    if (lse_out && P < Q) return RPO_ERR_INVALID_ARG;   // target_i = i * (P // Q) needs group_size >= 1
    hipStream_t st = (hipStream_t)stream;
    if (dtype == RPO_DT_F32)
        return fwd_impl<float>(q, p, Q, P, d, temperature, target_mode, scores_out, lse_out, loss_out, workspace,
                               workspace_bytes, st);
    if (dtype == RPO_DT_F16)
        return fwd_impl<f16_t>(q, p, Q, P, d, temperature, target_mode, scores_out, lse_out, loss_out, workspace,
                               workspace_bytes, st);
    return fwd_impl<bf16_t>(q, p, Q, P, d, temperature, target_mode, scores_out, lse_out, loss_out, workspace,
                            workspace_bytes, st);
}

extern "C" int rpo_infonce_bwd(const void* q, const void* p, const void* scores, const float* lse,
                               const float* grad_loss, int64_t Q, int64_t P, int64_t d, int dtype,
                               float temperature, int target_mode, int64_t q_row0, int64_t q_rows, int64_t p_row0,
                               int64_t p_rows, void* dq_out, void* dp_out, void* workspace, size_t workspace_bytes,
                               rpo_stream_t stream) {
    (void)workspace;
    (void)workspace_bytes;
    int rc = check_common(q, p, Q, P, d, dtype, temperature, target_mode);
    if (rc != RPO_OK) return rc;
    if (!scores || !lse || !grad_loss || P < Q) return RPO_ERR_INVALID_ARG;
    if (!dq_out && !dp_out) return RPO_ERR_INVALID_ARG;
    if (q_row0 < 0 || q_rows < 0 || q_row0 + q_rows > Q || p_row0 < 0 || p_rows < 0 || p_row0 + p_rows > P)
        return RPO_ERR_INVALID_ARG;
    if (dq_out && q_rows == 0) dq_out = nullptr;
    if (dp_out && p_rows == 0) dp_out = nullptr;
    if (!dq_out && !dp_out) return RPO_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == RPO_DT_F32)
        return bwd_impl<float>(q, p, scores, lse, grad_loss, Q, P, d, temperature, target_mode, q_row0, q_rows,
                               p_row0, p_rows, dq_out, dp_out, st);
    if (dtype == RPO_DT_F16)
        return bwd_impl<f16_t>(q, p, scores, lse, grad_loss, Q, P, d, temperature, target_mode, q_row0, q_rows,
                               p_row0, p_rows, dq_out, dp_out, st);
    return bwd_impl<bf16_t>(q, p, scores, lse, grad_loss, Q, P, d, temperature, target_mode, q_row0, q_rows, p_row0,
                            p_rows, dq_out, dp_out, st);
}

extern "C" int rpo_infonce_ds(const void* scores, const float* lse, const float* grad_loss, int64_t Q, int64_t P,
                              int dtype, float temperature, int64_t q_row0, int64_t q_rows, int64_t p_row0,
                              int64_t p_rows, void* ds_out, void* dst_out, rpo_stream_t stream) {
    if (!scores || !lse || !grad_loss || Q <= 0 || P < Q || !(temperature > 0.f)) return RPO_ERR_INVALID_ARG;
    if (!rpo_dtype_ok(dtype)) return RPO_ERR_INVALID_ARG;
    if (q_row0 < 0 || q_rows < 0 || q_row0 + q_rows > Q || p_row0 < 0 || p_rows < 0 || p_row0 + p_rows > P)
        return RPO_ERR_INVALID_ARG;
    if (q_rows == 0) ds_out = nullptr;
    if (p_rows == 0) dst_out = nullptr;
    if (!ds_out && !dst_out) return RPO_OK;
    if (rpo_cdiv(Q, 64) > 65535) return RPO_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)rpo_cdiv(P, 64), (unsigned)rpo_cdiv(Q, 64));
    if (dtype == RPO_DT_F32)
        RPO_LAUNCH(infonce_ds_kernel<float>, grid, dim3(256), 0, st, (const float*)scores, lse, grad_loss, Q, P,
                   temperature, P / Q, q_row0, q_rows, p_row0, p_rows, (float*)ds_out, (float*)dst_out);
    else if (dtype == RPO_DT_F16)
        RPO_LAUNCH(infonce_ds_kernel<f16_t>, grid, dim3(256), 0, st, (const f16_t*)scores, lse, grad_loss, Q, P,
                   temperature, P / Q, q_row0, q_rows, p_row0, p_rows, (f16_t*)ds_out, (f16_t*)dst_out);
    else
        RPO_LAUNCH(infonce_ds_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)scores, lse, grad_loss, Q, P,
                   temperature, P / Q, q_row0, q_rows, p_row0, p_rows, (bf16_t*)ds_out, (bf16_t*)dst_out);
    return rpo_launch_status();
}

// C [rows_b, rows_a] = B [rows_b, K] A [rows_a, K]^T (bf16 in, f32 accumulate, one rounding to bf16): see sim_tile256_kernel<1>.
extern "C" int rpo_sim_gemm_nt(const void* a, int64_t rows_a, int64_t lda, const void* b, int64_t rows_b, int64_t ldb, int64_t K,
                               void* c, int64_t ldc, rpo_stream_t stream) {
    if (!a || !b || !c || rows_a <= 0 || rows_b <= 0 || K <= 0) return RPO_ERR_INVALID_ARG;
    // the frame's K-step is 64 elements and its operands travel as 16-byte LDS-DMA pieces; the epilogue streams 256-byte row
    // segments out of LDS (the `staged` path of the forward): everything else is left to the caller's library GEMM
    if (K % 64 != 0 || lda % 8 != 0 || ldb % 8 != 0 || ldc % 8 != 0 || lda < K || ldb < K || ldc < rows_a || !rpo_aligned16(a) ||
        !rpo_aligned16(b) || !rpo_aligned16(c))
        return RPO_ERR_UNSUPPORTED;
    const int64_t nPt = rpo_cdiv(rows_a, kBigTile), nQt = rpo_cdiv(rows_b, kBigTile);
    if (nPt * nQt > 0x7fffffff) return RPO_ERR_UNSUPPORTED;
    // the kernel addresses its LDS-DMA pieces by 32-bit byte offsets from the operand bases
    if (rows_a * lda * 2 >= ((int64_t)1 << 32) || rows_b * ldb * 2 >= ((int64_t)1 << 32)) return RPO_ERR_UNSUPPORTED;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)sim_tile256_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kBigLdsBytes);
        attr_set = true;
    }
    RPO_LAUNCH(sim_tile256_kernel<1>, dim3((unsigned)(RPO_SIM_PERSIST ? std::min<int64_t>(nPt * nQt, kBigPersistBlocks) : nPt * nQt)),
               dim3(kBigThreads), kBigLdsBytes, (hipStream_t)stream, (const bf16_t*)b, (const bf16_t*)a, rows_b, rows_a, K, lda, ldb, ldc,
               1.0f, 0, 0, (bf16_t*)c, (float2*)nullptr, (int)nPt, (int)nQt, /*stagger=*/1, /*dbg=*/0, SimFilter{});
    return rpo_launch_status();
}

// One step of the exact search (retrieval.FlatIPIndex.search; the k-selection of faiss.IndexFlatIP.search, reference
// src/utils.py:58-80): the query block against one corpus chunk, survivors of the rows' current k-th winners appended to the rows'
// candidate lists -- sim_tile256_kernel<2>; rpo_topk_merge_candidates (topk.hip) folds the lists into the winners.
// 1 when rpo_infonce_fwd scores a bf16 [Q, d] x [P, d] problem with the 256 x 256 kernel: exactly the shapes rpo_sim_topk_filter
// takes, so that the fused step returns bit for bit what the score-matrix path returns (same kernel frame, same f32 summation order).
extern "C" int rpo_sim_topk_filter_ok(int64_t Q, int64_t P, int64_t d) {
    if (Q <= 0 || P <= 0 || d <= 0) return 0;
    return make_plan(Q, P, d, RPO_DT_BF16, /*aligned=*/true).path == PATH_TILE256 ? 1 : 0;
}

extern "C" int rpo_sim_topk_filter(const void* q, const void* p, int64_t Q, int64_t P, int64_t d, int dtype, int64_t col0, int k,
                                   int round_scores, const float* best_val, const int64_t* best_idx, float* cand_val,
                                   int64_t* cand_idx, int32_t* cand_cnt, int cap, rpo_stream_t stream) {
    if (!q || !p || !best_val || !best_idx || !cand_val || !cand_idx || !cand_cnt || Q <= 0 || P <= 0 || d <= 0 || col0 < 0 ||
        k <= 0 || cap <= 0 || (dtype != RPO_DT_BF16 && dtype != RPO_DT_F16))
        return RPO_ERR_INVALID_ARG;
    if (dtype == RPO_DT_F16 && round_scores) return RPO_ERR_UNSUPPORTED;        // fp16 operands: f32 scores only
    if (!rpo_aligned16(q) || !rpo_aligned16(p) || !rpo_sim_topk_filter_ok(Q, P, d)) return RPO_ERR_UNSUPPORTED;
    const int64_t nPt = rpo_cdiv(P, kBigTile), nQt = rpo_cdiv(Q, kBigTile);
    if (nPt * nQt > 0x7fffffff) return RPO_ERR_UNSUPPORTED;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)sim_tile256_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, kBigLdsBytes);
        (void)hipFuncSetAttribute((const void*)sim_tile256_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, kBigLdsBytes);
        (void)hipFuncSetAttribute((const void*)sim_tile256_kernel<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kBigLdsBytes);
        attr_set = true;
    }
    SimFilter flt{best_val, (const long long*)best_idx, cand_val, (long long*)cand_idx, cand_cnt, col0, k, cap};
    const dim3 grid((unsigned)std::min<int64_t>(nPt * nQt, kBigPersistBlocks));
#define RPO_FILTER_LAUNCH(...)                                                                                                   \
    RPO_LAUNCH((sim_tile256_kernel<__VA_ARGS__>), grid, dim3(kBigThreads), kBigLdsBytes, (hipStream_t)stream, (const bf16_t*)q, \
               (const bf16_t*)p, Q, P, d, d, d, P, 1.0f, 0, 0, (bf16_t*)nullptr, (float2*)nullptr, (int)nPt, (int)nQt,          \
               /*stagger=*/1, /*dbg=*/0, flt)
    if (dtype == RPO_DT_F16) RPO_FILTER_LAUNCH(3, true);
    else if (round_scores) RPO_FILTER_LAUNCH(2);
    else RPO_FILTER_LAUNCH(3);
#undef RPO_FILTER_LAUNCH
    return rpo_launch_status();
}

// scores f32 [Q, P] (row stride ldc) = q [Q, d] p [P, d]^T, bf16 or fp16 operands, f32 sums UNROUNDED, in the 256 x 256 frame's
// summation order (sim_tile256_kernel<4>): the score matrix that goes with rpo_sim_topk_filter(round_scores = 0); same shapes.
extern "C" int rpo_sim_scores_f32(const void* q, const void* p, int64_t Q, int64_t P, int64_t d, int dtype, float* scores, int64_t ldc,
                                  rpo_stream_t stream) {
    if (!q || !p || !scores || Q <= 0 || P <= 0 || d <= 0 || ldc < P || (dtype != RPO_DT_BF16 && dtype != RPO_DT_F16))
        return RPO_ERR_INVALID_ARG;
    if (!rpo_aligned16(q) || !rpo_aligned16(p) || !rpo_sim_topk_filter_ok(Q, P, d)) return RPO_ERR_UNSUPPORTED;
    const int64_t nPt = rpo_cdiv(P, kBigTile), nQt = rpo_cdiv(Q, kBigTile);
    if (nPt * nQt > 0x7fffffff) return RPO_ERR_UNSUPPORTED;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)sim_tile256_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, kBigLdsBytes);
        (void)hipFuncSetAttribute((const void*)sim_tile256_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kBigLdsBytes);
        attr_set = true;
    }
    const dim3 grid((unsigned)std::min<int64_t>(nPt * nQt, kBigPersistBlocks));
    if (dtype == RPO_DT_F16)
        RPO_LAUNCH((sim_tile256_kernel<4, true>), grid, dim3(kBigThreads), kBigLdsBytes, (hipStream_t)stream, (const bf16_t*)q,
                   (const bf16_t*)p, Q, P, d, d, d, ldc, 1.0f, 0, 0, (bf16_t*)scores, (float2*)nullptr, (int)nPt, (int)nQt,
                   /*stagger=*/1, /*dbg=*/0, SimFilter{});
    else
        RPO_LAUNCH((sim_tile256_kernel<4>), grid, dim3(kBigThreads), kBigLdsBytes, (hipStream_t)stream, (const bf16_t*)q,
                   (const bf16_t*)p, Q, P, d, d, d, ldc, 1.0f, 0, 0, (bf16_t*)scores, (float2*)nullptr, (int)nPt, (int)nQt,
                   /*stagger=*/1, /*dbg=*/0, SimFilter{});
    return rpo_launch_status();
}

thread_local int rpo_tls_last_hip_error = 0;

extern "C" int rpo_version(void) { return 100; }

extern "C" const char* rpo_last_hip_error(void) {
    return hipGetErrorString((hipError_t)rpo_tls_last_hip_error);
}

extern "C" const char* rpo_status_string(int status) {
    switch (status) {
        case RPO_OK: return "ok";
        case RPO_ERR_INVALID_ARG: return "invalid argument";
        case RPO_ERR_UNSUPPORTED: return "unsupported shape / dtype / alignment";
        case RPO_ERR_WORKSPACE: return "workspace missing, misaligned or too small";
        case RPO_ERR_LAUNCH: return "HIP launch error";
        default: return "unknown status";
    }
}
