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

#ifndef RPO_FA_STAGE_POS
#define RPO_FA_STAGE_POS 0      // where the forward kernel issues the LDS-DMA of tile kt + 2 (experiments: 1, 2)
#endif

#ifdef RPO_FA_STAMP   // diagnostic build only (tools/exp): in-kernel cycle stamps of a kernel's loop segments, summed per wave role
                      // (-DRPO_FA_STAMP: the 8-wave dK/dV kernel; + -DRPO_FA_STAMP_FWD: the forward kernel instead)
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
#ifdef RPO_FA_LADDER  // diagnostic build only (tools/exp/build_variant.sh ladder -DRPO_FA_LADDER; tools/fa_ladder64.py reads it): ONE record per
                      // block of the three head_dim-64 kernels -- s_memrealtime (100 MHz, the same clock on every CU) at the block's entry,
                      // at the start and the end of its key-tile / slice loop, after its last store was ISSUED and after it has LANDED,
                      // + where it ran (XCC id, HW id) and how many iterations its loop had.  Region K of the buffer = kernel K
                      // (0 forward, 1 dQ, 2 dK/dV); nothing of this exists in the shipped library.
constexpr int kLadMax = 1 << 17;
__device__ unsigned long long g_fa_ladder[3 * kLadMax * 8];
#define RPO_LAD_DECL unsigned long long lad_[5] = {0, 0, 0, 0, 0}
#define RPO_LAD(I)                                                                                      \
    do {                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(lad_[I])::"memory");             \
        __builtin_amdgcn_sched_barrier(0);                                                              \
    } while (0)
#define RPO_LAD_END(K, ITERS)                                                                           \
    do {                                                                                                \
        RPO_LAD(3);                                                                                     \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                \
        RPO_LAD(4);                                                                                     \
        const unsigned bid_ = blockIdx.x + gridDim.x * blockIdx.y;                                      \
        if (threadIdx.x == 0 && bid_ < (unsigned)kLadMax) {                                             \
            unsigned xcc_, hw_;                                                                         \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                         \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                           \
            unsigned long long* r_ = g_fa_ladder + ((size_t)(K) * kLadMax + bid_) * 8;                  \
            r_[0] = lad_[0]; r_[1] = lad_[1]; r_[2] = lad_[2]; r_[3] = lad_[3]; r_[4] = lad_[4];        \
            r_[5] = ((unsigned long long)xcc_ << 32) | hw_;                                             \
            r_[6] = (unsigned long long)(ITERS);                                                        \
            r_[7] = 1;                                                                                  \
        }                                                                                               \
    } while (0)
#else
#define RPO_LAD_DECL
#define RPO_LAD(I)
#define RPO_LAD_END(K, ITERS)
#endif
#if defined(RPO_FA_STAMP) && defined(RPO_FA_STAMP_FWD)
#define RPO_FSTAMP(VAR) RPO_STAMP(VAR)
#define RPO_FSTAMP_ADD(I, A, B) RPO_STAMP_ADD(I, A, B)
#else
#define RPO_FSTAMP(VAR)
#define RPO_FSTAMP_ADD(I, A, B)
#endif


// two f32 -> one dword of two bf16 (round to nearest even) in ONE v_cvt_pk_bf16_f32; written as `f32_to_bf16(a) | f32_to_bf16(b) << 16`
// hipcc converts each value on its own and merges them with a third instruction.  A vector conversion, NOT an asm statement: the
// instruction must be visible to hipcc's hazard recognizer -- a VALU instruction that reads the result of the v_exp_f32 right in
// front of it needs one wait state (transcendental forwarding), which hipcc cannot insert in front of an opaque asm.  Rounds 1-2
// had the asm form; with the forward kernels' row sums moved to the matrix pipe (round 3) nothing stood between the last v_exp_f32
// of a key tile and the conversion any more, and the head_dim-128 kernel packed the PRE-exp score (-1e30 on masked keys) into P.
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_));
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

#define RPO_TR2(OUT0, OUT1, ADDR, OFF0, OFF1)                                                                       \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:" #OFF0 "\n\tds_read_b64_tr_b16 %1, %2 offset:" #OFF1             \
                 : "=&v"(OUT0), "=&v"(OUT1)                                                                          \
                 : "v"(ADDR)                                                                                         \
                 : "memory")

__device__ __forceinline__ short8_t join_tr(const u32x2& lo, const u32x2& hi) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    const u32x4 w = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(short8_t, w);
}

// Inverse rotary embedding folded into the backward epilogues (round 2): the attention backward is asked for the gradients
// w.r.t. the PRE-rotary q / k (y1 = x1 cos - x2 sin, y2 = x2 cos + x1 sin with (x1, x2) = (x[j], x[j + hd / 2]), HF layout) ->
// dx1 = dy1 cos + dy2 sin, dx2 = dy2 cos - dy1 sin.  In every dQ^T / dK^T accumulator layout of this file a lane holds rows j
// and j + hd / 2 of the same token (hd tiles c and c + 2 at head_dim 64, c and c + 4 at 128), so the rotation is lane-local,
// in f32, BEFORE the one rounding to bf16 (the separate rpo_rope pass re-read the rounded gradient and rounded it again, and
// moved 0.28 ms of HBM traffic per block on cfg 2).  cos / sin: f32 [period][hd / 2], token t uses row t % period.
__device__ __forceinline__ void inv_rope4(float4_t& lo, float4_t& hi, const float4_t& c, const float4_t& sn) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float a = lo[r], b = hi[r];
        lo[r] = fmaf(a, c[r], b * sn[r]);
        hi[r] = fmaf(b, c[r], -(a * sn[r]));
    }
}

// Forward rotary of a block's OWN Q fragments (round 2): the forward kernel loads its 128 queries x 1 head once anyway, so it
// rotates them in registers (same arithmetic as rope_kernel: y1 = x1 cos - x2 sin, y2 = x2 cos + x1 sin, f32, one rounding) and
// writes them back in place -- every q element belongs to exactly one block, and the backward kernels read the rotated q from
// memory.  The separate rpo_rope pass then only has the k heads left (8 of 40 on cfg 2).  lo / hi = the lane's 8 elements of
// the low / high half of the head (hd j .. j + 7 and j + hd / 2 ..), c0 / c1 / s0 / s1 = cos / sin of j .. j + 3, j + 4 .. j + 7.
__device__ __forceinline__ void rope_frag(short8_t& lo, short8_t& hi, const float4_t& c0, const float4_t& c1,
                                          const float4_t& s0, const float4_t& s1) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    float y1[8], y2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float cc = e < 4 ? c0[e & 3] : c1[e & 3], ss = e < 4 ? s0[e & 3] : s1[e & 3];
        const float a = bf16_to_f32((bf16_t)lo[e]), b = bf16_to_f32((bf16_t)hi[e]);
        y1[e] = a * cc - b * ss;
        y2[e] = b * cc + a * ss;
    }
    const u32x4 wl = {pack_bf16(y1[0], y1[1]), pack_bf16(y1[2], y1[3]), pack_bf16(y1[4], y1[5]), pack_bf16(y1[6], y1[7])};
    const u32x4 wh = {pack_bf16(y2[0], y2[1]), pack_bf16(y2[2], y2[3]), pack_bf16(y2[4], y2[5]), pack_bf16(y2[6], y2[7])};
    lo = __builtin_bit_cast(short8_t, wl);
    hi = __builtin_bit_cast(short8_t, wh);
}

// Cross-group reductions of the forward kernels (a query's scores sit in lanes fr, fr + 16, fr + 32, fr + 48): gfx950's row swaps
// (v_permlane16_swap / v_permlane32_swap: two VALU instructions) instead of __shfl_xor's ds_bpermute round trips through the LDS
// queue, which sat in the dependent chain max -> exchange -> max -> exchange -> exp of every key tile.
__device__ __forceinline__ float max_xor16(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return max2_raw(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// max(v, v of lane ^ 32, also)
__device__ __forceinline__ float max_xor32_and(float v, float also) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return max3_raw(__uint_as_float(r[0]), __uint_as_float(r[1]), also);
}

// Query-tile work list, two formats (`tcols`, an argument of the C entry points):
//   2: int32 [n][2] = (sequence id, first query row), heaviest tiles first; grid = (n, heads), the head is blockIdx.y.
//   3: int32 [n][3] = (sequence id, first query row, head), n % 8 == 0, grid = (n): block b takes entry (b & 7) * n / 8 + (b >> 3),
//      i.e. the blocks that share an XCD (round-robin dispatch: b and b + 8) walk ONE eighth of the list in order.  The host
//      puts all query tiles x all q heads of one (sequence, kv head) into one eighth, next to each other: the ~100 blocks an
//      XCD runs at a time then stream the SAME K / V rows (<= 1 MB) through its 4 MB L2, instead of the K / V of every
//      sequence at once (format 2: FETCH traffic 6.3 x the algorithmic bytes, 50 % L2 hit rate, profiles/r01_fa_pmc.md).
//      Padding entries have first query row >= 2^30.  Placement is a speed matter only.
struct FaTile { int seq, q0, h; };
__device__ __forceinline__ FaTile fa_tile(const int* __restrict__ tiles, int tcols) {
    if (tcols == 3) {
        const int per = (int)(gridDim.x >> 3);
        const int e = 3 * ((int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3));
        return FaTile{tiles[e], tiles[e + 1], tiles[e + 2]};
    }
    return FaTile{tiles[2 * blockIdx.x], tiles[2 * blockIdx.x + 1], (int)blockIdx.y};
}

__global__ __launch_bounds__(kFaThreads, 2) void fa_fwd_kernel(
    const bf16_t* q /* no __restrict__: q_rw below aliases it when the rotary fold writes q back */, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, int64_t sq, int64_t sk,
    int64_t sv, const int* __restrict__ cu, const int* __restrict__ tiles, int tcols, int nh, int nkv, float scale_log2e,
    float scale, bf16_t* __restrict__ o, int64_t so, float* __restrict__ lse, int64_t lse_seq_stride,
    int64_t lse_head_stride, int lse_packed, const float* __restrict__ rcos, const float* __restrict__ rsin, int64_t rperiod,
    bf16_t* q_rw) {
    __shared__ __attribute__((aligned(16))) char smem[3 * kKvTile];       // ring of (K tile | V tile), 128-byte rows
    RPO_LAD_DECL;
    RPO_LAD(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, fr = lane & 15;
    const FaTile ft = fa_tile(tiles, tcols);
    if (ft.q0 >= (1 << 30)) return;                      // padding entry of the XCD-dealt list
    const int seq = ft.seq, q0 = ft.q0;
    const int h = ft.h, hk = h / (nh / nkv);
    const int64_t t0 = cu[seq];
    const int len = cu[seq + 1] - (int)t0;
    const int qw = q0 + 32 * wave;                       // this wave's first query row (inside the sequence)

    // Q^T fragments (B operand): lane = query fr of tile n, k = hd 32 ks + 8 g .. + 7
    short8_t bq[2][2];
    float4_t rc[2][2], rs[2][2];                         // rotary tables of the lane's 8 frequencies (only when rcos is given)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int qi = qw + 16 * n + fr;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (qi < len) bq[n][ks] = *reinterpret_cast<const short8_t*>(q + (t0 + qi) * sq + h * kFaHD + 32 * ks + 8 * g);
            else bq[n][ks] = short8_t{0, 0, 0, 0, 0, 0, 0, 0};
        }
        if (rcos && qi < len) {
            const int64_t tr = ((t0 + qi) % rperiod) * (kFaHD / 2) + 8 * g;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                rc[n][i] = *reinterpret_cast<const float4_t*>(rcos + tr + 4 * i);
                rs[n][i] = *reinterpret_cast<const float4_t*>(rsin + tr + 4 * i);
            }
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
    if (rcos) {                                          // rotary on the block's own Q, written back in place (rope_frag)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int qi = qw + 16 * n + fr;
            if (qi >= len) continue;
            rope_frag(bq[n][0], bq[n][1], rc[n][0], rc[n][1], rs[n][0], rs[n][1]);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                *reinterpret_cast<short8_t*>(q_rw + (t0 + qi) * sq + h * kFaHD + 32 * ks + 8 * g) = bq[n][ks];
        }
    }

    float4_t oacc[4][2];                                 // O^T: [hd tile c][query tile n], rows = hd 16c + 4g + r
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int n = 0; n < 2; ++n) oacc[c][n] = float4_t{0.f, 0.f, 0.f, 0.f};
    float mrun[2] = {-1e30f, -1e30f};
    // The softmax denominators come out of the matrix pipe: one more A tile of ONES beside V^T gives sum_keys P^T[key][q] in every row
    // of a 16 x 16 accumulator (all four registers of lane (g, fr) = the row sum of query fr, summed over the key groups of all four
    // g) -- 4 MFMAs per key tile instead of 32 v_add + 2 cross-group exchanges per query tile in a VALU-bound loop.  The sum is
    // taken over the bf16-rounded probabilities, the very operand O^T is accumulated from.
    float4_t lacc[2] = {float4_t{0.f, 0.f, 0.f, 0.f}, float4_t{0.f, 0.f, 0.f, 0.f}};
    short8_t ones;                                       // bf16 1.0 x 8, pinned in four registers for the whole kernel
    {
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_;
        u32x4_ w;
        asm volatile("v_mov_b32 %0, 0x3f803f80\n\tv_mov_b32 %1, 0x3f803f80\n\tv_mov_b32 %2, 0x3f803f80\n\tv_mov_b32 %3, 0x3f803f80"
                     : "=v"(w[0]), "=v"(w[1]), "=v"(w[2]), "=v"(w[3]));
        ones = __builtin_bit_cast(short8_t, w);
    }

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

#if defined(RPO_FA_STAMP) && defined(RPO_FA_STAMP_FWD)
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ta, tb_, tc, td, te, tf, tg;
#endif
    int cur = 0;
    RPO_LAD(1);
    for (int kt = 0; kt < nkt; ++kt) {
        RPO_FSTAMP(ta);
        if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // tile kt landed; tile kt + 1 may fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RPO_FSTAMP(tb_);
        __builtin_amdgcn_s_barrier();
        RPO_FSTAMP(tc);
#if RPO_FA_STAGE_POS == 0
        if (kt + 2 < nkt) stage(kt + 2, cur == 0 ? 2 : cur - 1);               // the buffer read in iteration kt - 1
#endif
        RPO_FSTAMP(td);
        RPO_FSTAMP_ADD(0, ta, tb_);
        RPO_FSTAMP_ADD(1, tb_, tc);
        RPO_FSTAMP_ADD(2, tc, td);
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
#ifdef RPO_FA_EXP_SETPRIO      // A/B build (round 4: round 3's in-step measurement of this went through a --lib that switched nothing)
            __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const short8_t a = *reinterpret_cast<const short8_t*>(Ks + row_off[ks] + m * 2048);
#pragma unroll
                    for (int n = 0; n < 2; ++n) s[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bq[n][ks], s[m][n], 0, 0, 0);
                }
            }
#ifdef RPO_FA_EXP_SETPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            RPO_FSTAMP(te);
            RPO_FSTAMP_ADD(3, td, te);
#if RPO_FA_STAGE_POS == 1
            if (kt + 2 < nkt) stage(kt + 2, cur == 0 ? 2 : cur - 1);           // behind the S MFMAs: their results are still in the pipe
#endif
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
#if defined(RPO_FA_EXP_NOSOFTMAX)
            // TIMING ABLATION (round 4, never shipped; tools/exp/build_variant.sh fwdnosm -DRPO_FA_EXP_NOSOFTMAX): the key-tile loop
            // without its vector arithmetic -- no row maximum, no exp2, no rescale; the scores are packed as they leave the MFMA.
            // What is left is the MFMA + LDS + DMA + barrier skeleton: the time a PERFECT overlap of the softmax under the matrix
            // pipe (a software pipeline across key tiles) could reach at best.  Results are garbage.
            short8_t pfrag[2][2];
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                pfrag[0][n] = pack_frag(s[0][n], s[1][n]);
                pfrag[1][n] = pack_frag(s[2][n], s[3][n]);
            }
#else
            float mnew[2];
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                float mm = max3_known(s[0][n][0], s[0][n][1], s[0][n][2]);          // 16 scores of this lane: 8 v_max3_f32
                mm = max3_known(mm, s[0][n][3], s[1][n][0]);
                mm = max3_known(mm, s[1][n][1], s[1][n][2]);
                mm = max3_known(mm, s[1][n][3], s[2][n][0]);
                mm = max3_known(mm, s[2][n][1], s[2][n][2]);
                mm = max3_known(mm, s[2][n][3], s[3][n][0]);
                mm = max3_known(mm, s[3][n][1], s[3][n][2]);
                mm = max2_known(mm, s[3][n][3]);
                mm = max_xor32_and(max_xor16(mm), mrun[n]);
                mnew[n] = mm;
            }
            // the running maximum rarely moves after the first tiles: skip the rescale of O (32 multiplies) when no lane's did
            if (__builtin_amdgcn_ballot_w64(mnew[0] != mrun[0] || mnew[1] != mrun[1]) != 0) {
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const float alpha = __builtin_amdgcn_exp2f((mrun[n] - mnew[n]) * scale_log2e);
                    lacc[n] *= alpha;
                    mrun[n] = mnew[n];
#pragma unroll
                    for (int c = 0; c < 4; ++c) oacc[c][n] *= alpha;
                }
            }
            short8_t pfrag[2][2];                        // [k-step s (32 keys)][query tile n]
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const float mls = mnew[n] * scale_log2e;
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[m][n][r] = __builtin_amdgcn_exp2f(fmaf(s[m][n][r], scale_log2e, -mls));
                pfrag[0][n] = pack_frag(s[0][n], s[1][n]);
                pfrag[1][n] = pack_frag(s[2][n], s[3][n]);
            }
#endif
            RPO_FSTAMP(tf);
            RPO_FSTAMP_ADD(4, te, tf);
            // ---- O^T += V^T P^T   (A = V^T via the transposed LDS reads, same key order as the P fragments).
            // hipcc does not model inline-asm LDS reads: naming every destination in the wait keeps the MFMAs below it.
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(y0),
                           "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7)
                         :
                         : "memory");
            const short8_t vt0[4] = {join_tr(x0, x1), join_tr(x2, x3), join_tr(x4, x5), join_tr(x6, x7)};
            const short8_t vt1[4] = {join_tr(y0, y1), join_tr(y2, y3), join_tr(y4, y5), join_tr(y6, y7)};
#ifdef RPO_FA_EXP_SETPRIO
            __builtin_amdgcn_s_setprio(1);
#endif
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
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                lacc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pfrag[0][n], lacc[n], 0, 0, 0);
                lacc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pfrag[1][n], lacc[n], 0, 0, 0);
            }
#ifdef RPO_FA_EXP_SETPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            RPO_FSTAMP(tg);
            RPO_FSTAMP_ADD(5, tf, tg);
#if defined(RPO_FA_STAMP) && defined(RPO_FA_STAMP_FWD)
            st_acc[6] += 1;
#endif
        }
#if RPO_FA_STAGE_POS == 1
        else if (kt + 2 < nkt) stage(kt + 2, cur == 0 ? 2 : cur - 1);
#endif
#if RPO_FA_STAGE_POS == 2
        if (kt + 2 < nkt) stage(kt + 2, cur == 0 ? 2 : cur - 1);               // at the end of the iteration: the waves have drifted apart
#endif
#if defined(RPO_FA_STAMP) && defined(RPO_FA_STAMP_FWD)
        st_acc[7] += 1;
#endif
        cur = cur == 2 ? 0 : cur + 1;
    }
#if defined(RPO_FA_STAMP) && defined(RPO_FA_STAMP_FWD)
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&g_fa_stamp[wave * 8 + i], st_acc[i]);
#endif
    RPO_LAD(2);
    // ---- epilogue: O[q][16c + 4g + r] = O^T / l ;  lse = scale m + ln l
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int qi = qw + 16 * n + fr;
        if (qi >= len) continue;
        const float l = lacc[n][0];                      // every row of the ones-tile accumulator holds the row sum
        const float inv = 1.0f / l;
        bf16_t* orow = o + (t0 + qi) * so + h * kFaHD;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint2 w;
            w.x = pack_bf16(oacc[c][n][0] * inv, oacc[c][n][1] * inv);
            w.y = pack_bf16(oacc[c][n][2] * inv, oacc[c][n][3] * inv);
            *reinterpret_cast<uint2*>(orow + 16 * c + 4 * g) = w;
        }
        if (g == 0 && lse)                               // lse NULL: forward-only caller (no backward will read it)
            lse[(int64_t)seq * lse_seq_stride + (int64_t)h * lse_head_stride + (lse_packed ? t0 : 0) + qi] =
                mrun[n] * scale + logf(l);
    }
    RPO_LAD_END(0, nkt);
}

// ------------------------------------------------------------------------------------------------------------------
// head_dim 128 (the Llama-3-8B architecture, BASELINE configs[4]): the same forward with the tile re-proportioned so that the
// LDS ring, the DMA count and the MFMAs per barrier stay what they are at head_dim 64 -- key tiles of 32 (32 keys x 256 B =
// the same 8 KiB image as 64 keys x 128 B), 4 k-steps for S^T = K Q^T (16 MFMAs), 8 hd tiles for O^T = V^T P^T (16 MFMAs), half
// the softmax arithmetic per MFMA.  LDS rows are 256 bytes = the whole bank width, so the swizzle is chunk ^= 2 (row & 7):
// conflict-free for the ds_read_b128 row reads of K (16 rows x one chunk) AND for the ds_read_b64_tr_b16 reads of V^T
// (8 rows x 32 bytes per half-wave), checked by enumeration.  A 1-KiB DMA piece is 4 rows.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kFa128HD = 128, kFa128BN = 32, kFa128Row = 256;

#ifndef RPO_F128_EXP
#define RPO_F128_EXP 0   // timing-only ablations (tools/exp/build_variant.sh NAME -DRPO_F128_EXP=mask; results are WRONG by design):
#endif                   // 1 no softmax arithmetic, 2 no LDS-DMA staging inside the loop, 4 no barrier, 8 no waits on the DMA ring

// A block = WQ x HEADS waves: WQ waves x 32 queries = 32 WQ queries of HEADS consecutive q heads (which share ONE kv head: HEADS
// divides nh / nkv), all on the same staged K / V tile; SUB sub-tiles of 32 keys per staged tile, i.e. per barrier, per counted
// wait and per burst of DMA issues (round 5).  Every wave runs the SAME arithmetic on its 32 queries in the same order whatever
// the instantiation (the online softmax advances 32 keys at a time), so all of them are bit-identical; what changes is how often
// the waves meet and how many LDS-DMA pieces each of them issues per MFMA -- the timing ablations (RPO_F128_EXP, cfg 5's shape,
// profiles/r05_fa_fwd128_ablation.txt) price the staging at 21 % of the kernel, the softmax arithmetic at 27 %, the barrier at 6 %:
//   <4, 1, 1>  rounds 2-4: 128 queries x 1 head, barrier + 4 DMA pieces per wave per 32 keys (34 MFMAs)
//   <8, 1, 2>  256 queries share the tile (2 pieces per wave per 32 keys), one barrier per 64 keys; the block's early waves idle
//              through its last 7 key tiles (causal)
//   <4, 2, 2>  128 queries x 2 heads share the tile: the same halving of the staging without that idling
//   KF = true  all eight K row fragments of a sub-tile are read FIRST (one wait, then the 16 S^T MFMAs back to back) and the V^T
//              fragments are read behind those MFMAs, under the softmax arithmetic; false (rounds 2-4): V^T reads first, K fragments
//              two at a time with a wait in front of every group of four MFMAs
template <int WQ, int HEADS, int SUB, bool KF>
__global__ __launch_bounds__(64 * WQ * HEADS, WQ * HEADS == 4 ? 2 : 1) void fa_fwd128_kernel(
    const bf16_t* q /* no __restrict__: q_rw below aliases it when the rotary fold writes q back */, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, int64_t sq, int64_t sk,
    int64_t sv, const int* __restrict__ cu, const int* __restrict__ tiles, int tcols, int nh, int nkv, float scale_log2e,
    float scale, bf16_t* __restrict__ o, int64_t so, float* __restrict__ lse, int64_t lse_seq_stride,
    int64_t lse_head_stride, int lse_packed, const float* __restrict__ rcos, const float* __restrict__ rsin, int64_t rperiod,
    bf16_t* q_rw) {
    constexpr int WAVES = WQ * HEADS;
    constexpr int BM = 32 * WQ;                          // queries per block (of each of its HEADS heads)
    constexpr int BN = kFa128BN * SUB;                   // keys per staged tile
    constexpr int kTile = 2 * BN * kFa128Row;            // (K tile | V tile), 256-byte rows
    constexpr int kVOff = BN * kFa128Row;                // V image behind the K image
    constexpr int PPW = (BN / 4) / WAVES;                // 1-KiB DMA pieces (4 rows) per operand and wave
    static_assert(PPW >= 1 && PPW * WAVES * 4 == BN, "the waves split a tile's rows evenly");
    constexpr int RING = 3;
    __shared__ __attribute__((aligned(16))) char smem[RING * kTile];      // ring of three tiles
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, fr = lane & 15;
    const FaTile ft = fa_tile(tiles, tcols);
    if (ft.q0 >= (1 << 30)) return;
    const int seq = ft.seq, q0 = ft.q0;
    const int h_first = tcols == 3 ? ft.h : ft.h * HEADS;          // the FIRST of the block's HEADS heads (format 2: blockIdx.y counts blocks)
    const int h = h_first + wave / WQ, hk = h_first / (nh / nkv);
    const int64_t t0 = cu[seq];
    const int len = cu[seq + 1] - (int)t0;
    const int qw = q0 + 32 * (wave % WQ);

    short8_t bq[2][4];                                   // Q^T fragments: lane = query fr of tile n, k = hd 32 ks + 8 g .. + 7
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int qi = qw + 16 * n + fr;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (qi < len) bq[n][ks] = *reinterpret_cast<const short8_t*>(q + (t0 + qi) * sq + h * kFa128HD + 32 * ks + 8 * g);
            else bq[n][ks] = short8_t{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    const int last_q = min(q0 + BM - 1, len - 1);
    const int nkt = last_q / BN + 1;
    // staging: DMA piece u = PPW * wave + i fills tile rows 4u .. 4u + 3 of K and of V; lane l carries row 4u + (l >> 4),
    // physical chunk l & 15 = logical chunk (l & 15) ^ 2 (row & 7)
    const int srow = lane >> 4;
    const char* ksrc = reinterpret_cast<const char*>(k + t0 * sk + hk * kFa128HD);
    const char* vsrc = reinterpret_cast<const char*>(v + t0 * sv + hk * kFa128HD);
    const unsigned skb = (unsigned)sk * 2u, svb = (unsigned)sv * 2u;
    auto stage = [&](int kt, int buf) {
        char* base = smem + buf * kTile;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int u = PPW * wave + i;
            const int trow = 4 * u + srow;                                         // tile row 0 .. BN - 1
            const unsigned lchunk = (unsigned)((lane & 15) ^ (2 * (trow & 7)));
            const unsigned row = (unsigned)min(kt * BN + trow, len - 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ksrc + (row * skb + lchunk * 16)),
                                             (__attribute__((address_space(3))) void*)(base + u * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vsrc + (row * svb + lchunk * 16)),
                                             (__attribute__((address_space(3))) void*)(base + kVOff + u * 1024), 16, 0, 0);
        }
    };
    stage(0, 0);
    if (nkt > 1) stage(1, 1);
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(bq[n][ks]));
    if (rcos) {                                          // rotary on the block's own Q, written back in place (rope_frag): the
#pragma unroll                                           // lane's hd 32 ks + 8g + e pairs with 32 (ks + 2) + 8g + e
        for (int n = 0; n < 2; ++n) {
            const int qi = qw + 16 * n + fr;
            if (qi >= len) continue;
            const int64_t tr = ((t0 + qi) % rperiod) * (kFa128HD / 2) + 8 * g;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                rope_frag(bq[n][ks], bq[n][ks + 2], *reinterpret_cast<const float4_t*>(rcos + tr + 32 * ks),
                          *reinterpret_cast<const float4_t*>(rcos + tr + 32 * ks + 4),
                          *reinterpret_cast<const float4_t*>(rsin + tr + 32 * ks),
                          *reinterpret_cast<const float4_t*>(rsin + tr + 32 * ks + 4));
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                *reinterpret_cast<short8_t*>(q_rw + (t0 + qi) * sq + h * kFa128HD + 32 * ks + 8 * g) = bq[n][ks];
        }
    }

    float4_t oacc[8][2];                                 // O^T: [hd tile c][query tile n], rows = hd 16c + 4g + r
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int n = 0; n < 2; ++n) oacc[c][n] = float4_t{0.f, 0.f, 0.f, 0.f};
    float mrun[2] = {-1e30f, -1e30f};
    float4_t lacc[2] = {float4_t{0.f, 0.f, 0.f, 0.f}, float4_t{0.f, 0.f, 0.f, 0.f}};     // softmax denominators from the matrix pipe (fa_fwd_kernel)
    short8_t ones;                                       // bf16 1.0 x 8, pinned in four registers for the whole kernel
    {
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_;
        u32x4_ w;
        asm volatile("v_mov_b32 %0, 0x3f803f80\n\tv_mov_b32 %1, 0x3f803f80\n\tv_mov_b32 %2, 0x3f803f80\n\tv_mov_b32 %3, 0x3f803f80"
                     : "=v"(w[0]), "=v"(w[1]), "=v"(w[2]), "=v"(w[3]));
        ones = __builtin_bit_cast(short8_t, w);
    }

    // K row reads: row 16 m + fr, chunk (4 ks + g) ^ 2 (fr & 7).  V transposed reads: lane (g, qq, pp) addresses row 16 h + 4 g + qq,
    // hd columns 16 c + 4 pp .. + 3 = chunk (2 c + (pp >> 1)) ^ 2 (4 (g & 1) + qq), byte 8 (pp & 1) inside it; h = immediate 4096.
    // Sub-tile j of a staged tile: + 32 rows = 8192 bytes in both images (row & 7 and with it the swizzle repeat every 8 rows).
    const int qq = fr >> 2, pp = fr & 3;
    const int vsw = 2 * (4 * (g & 1) + qq);
    unsigned tr_off[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
        tr_off[c] = kVOff + (4 * g + qq) * kFa128Row + (((2 * c + (pp >> 1)) ^ vsw) << 4) + 8 * (pp & 1);
    unsigned row_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) row_off[ks] = fr * kFa128Row + (((4 * ks + g) ^ (2 * (fr & 7))) << 4);
    const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
#if !(RPO_F128_EXP & 8)
        if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");    // tile kt landed; tile kt + 1 may fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#if !(RPO_F128_EXP & 4)
        __builtin_amdgcn_s_barrier();
#endif
#if !(RPO_F128_EXP & 2)
        if (kt + 2 < nkt) stage(kt + 2, cur == 0 ? 2 : cur - 1);
#endif
#pragma unroll
        for (int sub = 0; sub < SUB; ++sub) {
            const int k0 = kt * BN + kFa128BN * sub;                 // first key of this 32-key sub-tile
            const bool active = (k0 <= qw + 31) && (qw < len);
            if (!active) continue;
            const char* Ks = smem + cur * kTile + sub * (kFa128BN * kFa128Row);
            const unsigned tb = smem_base + cur * kTile + sub * (kFa128BN * kFa128Row);
            // V^T fragments of the 32 keys: x[c] = keys 4g .. 4g + 3 of rows 0-15, y[c] = of rows 16-31, for hd tile c
            u32x2 x0, x1, x2, x3, x4, x5, x6, x7, y0, y1, y2, y3, y4, y5, y6, y7;
#define RPO_TR2V(OUT0, OUT1, ADDR)                                                                              \
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:4096" : "=&v"(OUT0), "=&v"(OUT1) : "v"(ADDR) : "memory")
#define RPO_TR_ALL()                   \
    RPO_TR2V(x0, y0, tb + tr_off[0]);  \
    RPO_TR2V(x1, y1, tb + tr_off[1]);  \
    RPO_TR2V(x2, y2, tb + tr_off[2]);  \
    RPO_TR2V(x3, y3, tb + tr_off[3]);  \
    RPO_TR2V(x4, y4, tb + tr_off[4]);  \
    RPO_TR2V(x5, y5, tb + tr_off[5]);  \
    RPO_TR2V(x6, y6, tb + tr_off[6]);  \
    RPO_TR2V(x7, y7, tb + tr_off[7])
            if constexpr (!KF) { RPO_TR_ALL(); }
            // ---- S^T = K Q^T: 2 key sub-tiles x 2 query tiles, 4 k-steps
            float4_t s[2][2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) s[m][n] = float4_t{0.f, 0.f, 0.f, 0.f};
            if constexpr (KF) {
                short8_t ak[4][2];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
                        ak[ks][m] = *reinterpret_cast<const short8_t*>(Ks + row_off[ks] + m * 16 * kFa128Row);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
                            s[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ak[ks][m], bq[n][ks], s[m][n], 0, 0, 0);
                // every score register is named here so that the transposed reads below are issued behind the MFMAs, not above
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) asm volatile("" : "+v"(s[m][n]));
                RPO_TR_ALL();
            } else {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const short8_t a = *reinterpret_cast<const short8_t*>(Ks + row_off[ks] + m * 16 * kFa128Row);
#pragma unroll
                        for (int n = 0; n < 2; ++n) s[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bq[n][ks], s[m][n], 0, 0, 0);
                    }
                }
            }
#undef RPO_TR_ALL
#undef RPO_TR2V
            const int kbase = k0 + 4 * g;
            const bool need_mask = (k0 + kFa128BN - 1 > qw) || (k0 + kFa128BN > len);
            if (need_mask) {
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int qi = qw + 16 * n + fr;
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int key = kbase + 16 * m + r;
                            if (key > qi || key >= len) s[m][n][r] = -1e30f;
                        }
                }
            }
            short8_t pfrag[2];                           // [query tile n]: k-slots = keys {4g + j, 16 + 4g + (j - 4)} of the tile
#if (RPO_F128_EXP & 1)
#pragma unroll
            for (int n = 0; n < 2; ++n) pfrag[n] = pack_frag(s[0][n], s[1][n]);
#else
            float mnew[2];
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                float mm = max3_known(s[0][n][0], s[0][n][1], s[0][n][2]);
                mm = max3_known(mm, s[0][n][3], s[1][n][0]);
                mm = max3_known(mm, s[1][n][1], s[1][n][2]);
                mm = max2_known(mm, s[1][n][3]);
                mm = max_xor32_and(max_xor16(mm), mrun[n]);
                mnew[n] = mm;
            }
            if (__builtin_amdgcn_ballot_w64(mnew[0] != mrun[0] || mnew[1] != mrun[1]) != 0) {
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const float alpha = __builtin_amdgcn_exp2f((mrun[n] - mnew[n]) * scale_log2e);
                    lacc[n] *= alpha;
                    mrun[n] = mnew[n];
#pragma unroll
                    for (int c = 0; c < 8; ++c) oacc[c][n] *= alpha;
                }
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const float mls = mnew[n] * scale_log2e;
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[m][n][r] = __builtin_amdgcn_exp2f(fmaf(s[m][n][r], scale_log2e, -mls));
                pfrag[n] = pack_frag(s[0][n], s[1][n]);
            }
#endif
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(y0),
                           "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7)
                         :
                         : "memory");
            const short8_t vt[8] = {join_tr(x0, y0), join_tr(x1, y1), join_tr(x2, y2), join_tr(x3, y3),
                                    join_tr(x4, y4), join_tr(x5, y5), join_tr(x6, y6), join_tr(x7, y7)};
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    oacc[c][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt[c], pfrag[n], oacc[c][n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < 2; ++n) lacc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pfrag[n], lacc[n], 0, 0, 0);
        }
        cur = cur == 2 ? 0 : cur + 1;
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int qi = qw + 16 * n + fr;
        if (qi >= len) continue;
        const float l = lacc[n][0];                      // every row of the ones-tile accumulator holds the row sum
        const float inv = 1.0f / l;
        bf16_t* orow = o + (t0 + qi) * so + h * kFa128HD;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            uint2 w;
            w.x = pack_bf16(oacc[c][n][0] * inv, oacc[c][n][1] * inv);
            w.y = pack_bf16(oacc[c][n][2] * inv, oacc[c][n][3] * inv);
            *reinterpret_cast<uint2*>(orow + 16 * c + 4 * g) = w;
        }
        if (g == 0 && lse)                               // lse NULL: forward-only caller (no backward will read it)
            lse[(int64_t)seq * lse_seq_stride + (int64_t)h * lse_head_stride + (lse_packed ? t0 : 0) + qi] =
                mrun[n] * scale + logf(l);
    }
}

// (Round 5 also built this forward as a software pipeline inside the wave -- S^T of key tile kt + 1 issued before the softmax of tile
// kt, ring of four tiles, chains and exponentials in one basic block under sched_group_barrier -- bit-identical, and 0.4-1.7 % SLOWER
// in same-process A/Bs: with two waves per SIMD the partner wave already fills what the pipeline would; commit 1bc251a holds the
// kernel, profiles/r05_fa_fwd128_ablation.txt the numbers.)

// ------------------------------------------------------------------------------------------------------------------
// head_dim 128, ONE WAVE PER SIMD (round 5; rpo_flash_attn_fwd's q_block = 64): block = 64 queries x the FOUR q heads of one kv head,
// one head per wave (one staged K / V tile serves all four: half the K / V traffic per query row of a 128-query block, and every
// wave walks the same number of key tiles).  The wave's O^T (128 registers), softmax denominators (16), Q^T fragments (64) and the K
// fragments of the tile in flight (32) live in the accumulator file, the scores, V^T and P^T fragments in literal VGPRs; the
// key-tile loop is two generated asm statements per 32-key tile (tools/gen/gen_fwd128w_body.py -> attention_fwd128w_gen.inc, register
// map and pipeline in its docstring): P1 = the S^T chains of the NEXT tile with this tile's exponentials in their gaps, P2 = this
// tile's O^T / l products with the next tile's exponents (e = c s - scale, in place) and the lane's largest e in theirs, plus the
// wave's four LDS-DMA pieces of the tiles staged this iteration.  The stream is ISSUE-bound (a vector instruction holds the SIMD's
// port 4 cycles, v_exp_f32 8, a 16x16x32 MFMA 8 of its 16), so the work that counts is the instruction count per tile:
//   * the softmax scale is DEFERRED: it starts at tile 0's row maximum (statement FIRST) and moves only when some lane's exponent
//     exceeds kFwDefer (one ballot between P2 and the next P1, then RESCALE: the cross-lane row maximum, O^T and l through VGPRs) --
//     p <= 2^kFwDefer, no accuracy cost in floating point, but not bit-identical to the exact-maximum kernel;
//   * no row-maximum reduction in the steady state (no lane above the threshold means no row above it);
//   * round 5's ladder, cfg-5-like batch, stand-alone: 0.304 (first correct) -> 0.340 (no hazard padding between statements, K / V
//     rings apart, slot immediates) -> 0.357 (64 queries x 4 heads) -> 0.365 (DMA in the stream, interleaved maxima) -> 0.415
//     (exponent in P2, lane-only maximum, exponentials split 3 : 1 over P1 / P2 by issue cost) -> 0.436 (Q in / O out as whole rows
//     through LDS) against 0.333-0.337 for fa_fwd128_kernel on the same boxes (profiles/r05_fa_fwd128w_ladder.md).
// K / V: 8-KiB images (32 keys x 256 bytes, chunk ^= 2 (row & 7)) in a ring of four K tiles and a ring of four V tiles (tile j's K
// rows are read two iterations before its V rows), one barrier per tile.  Q and O: a 16-KiB region per wave, whole 256-byte rows
// between LDS and memory (per-lane fragment accesses at a row stride touch 16 lines of 32-64 bytes per instruction, and one wave
// per SIMD has nobody to hide that behind).  Same fragment layouts and rotary fold as fa_fwd128_kernel.
// ------------------------------------------------------------------------------------------------------------------
#if defined(RPO_FW_VARIANT_INC)                      // timing experiments: the generator's output under its GEN_* switches (results may be
#include "attention_fwd128w_gen_variant.inc"         // wrong); tools/exp/build_variant.sh writes the file into its build directory
#else
#include "attention_fwd128w_gen.inc"
#endif
#ifdef RPO_ONEWAVE64                                 // `make ONEWAVE64=1`: the head_dim-64 one-wave kernels (fa_fwd64w_kernel, fa_bwd_dq64w_kernel).
#include "attention_fwd64w_gen.inc"                  // Built, tested and measured in round 5; they LOSE 4-10 % to the two-waves-per-SIMD kernels
#endif                                               // at head_dim 64, so the default library leaves their ~6000 generated lines out
#ifndef RPO_FW_EXP
#define RPO_FW_EXP 0     // timing-only ablations: 2 no LDS-DMA staging inside the loop, 4 no barrier / ring wait
#endif
#ifndef RPO_FW_NT
#define RPO_FW_NT 0      // A/B: 1 the Q pieces (read once) by non-temporal LDS-DMA, 2 the O rows by non-temporal stores, 4 the rotated Q rows
#endif
#ifndef RPO_FW_DEFER
#define RPO_FW_DEFER 8.0f
#endif
constexpr float kFwDefer = RPO_FW_DEFER;

#define FWW_HD 128
#define FWW(X) RPO_FW_##X
#define FWW_KERNEL fa_fwd128w_kernel
#define FWW_TRS trv[0], trv[1], trv[2], trv[3], trv[4], trv[5], trv[6], trv[7]
#define FWW_KRS krow[0], krow[1], krow[2], krow[3]
#include "attention_fwdw_kernel.inc"
#undef FWW_HD
#undef FWW
#undef FWW_KERNEL
#undef FWW_TRS
#undef FWW_KRS
#ifdef RPO_ONEWAVE64
// head_dim 64 (round 5): the same kernel source; 4-KiB images (rings of 16 + 16 KiB), 8-KiB Q / O regions, two DMA pieces per wave
// and tile.  Half the matrix work per exponential: 16 + 20 MFMAs per tile around the same 32 exponentials per lane.
#define FWW_HD 64
#define FWW(X) RPO_FW64_##X
#define FWW_KERNEL fa_fwd64w_kernel
#define FWW_TRS trv[0], trv[1], trv[2], trv[3]
#define FWW_KRS krow[0], krow[1]
#include "attention_fwdw_kernel.inc"
#undef FWW_HD
#undef FWW
#undef FWW_KERNEL
#undef FWW_TRS
#undef FWW_KRS
#endif  // RPO_ONEWAVE64

// ------------------------------------------------------------------------------------------------------------------
// Backward.  Two launches, no atomics, deterministic:
//   fa_bwd_dq_kernel     block = 128 queries of one (sequence, head): delta[h][t] = sum_d dO[t,h,d] O[t,h,d] in the prologue,
//                        then the loop over key tiles <= diagonal:
//                        S^T = K Q^T, dP^T = V dO^T, dS^T = P (dP - delta) scale, dQ^T += K^T dS^T
//                        (query on the lane: lse / delta are per-lane scalars; dS^T accumulators are the B fragments)
//   fa_bwd_dkdv_kernel   block = 64 keys of one (sequence, kv head), loop over the q heads of the group and the query
//                        tiles >= the key tile: S = Q K^T, dP = dO V^T (key on the lane, K / V fragments stay in
//                        registers), dV^T += dO^T P, dK^T += Q^T dS with Q^T / dO^T read transposed from the same LDS
//                        images that serve the row reads (chunk ^= row & 7 is conflict-free for both kinds of read).
// P is recomputed from the saved lse: P = exp(scale s - lse).
// ------------------------------------------------------------------------------------------------------------------
// The two per-(head, query) row constants of the backward, NEGATED so that they can be the initial accumulators of the S and dP
// MFMA chains: nd[h][t] = -delta = -sum_d dO O and nl[h][t] = -lse / scale, so that S' = Q K^T - lse / scale gives
// p = exp2(scale log2(e) S') with no subtraction, and dP' = dO V^T - delta is dS / (p scale) as it leaves the chain.  The dQ kernel
// computes them in its prologue and writes them for the dK/dV kernel (round 1: a separate fa_delta_kernel, 0.32 ms per call).
// (Tried and dropped: block-resident Q fragments pre-multiplied by scale log2(e) and re-rounded to bf16, so that the chain's
// result is the exponent itself -- one multiplication per score less, backward 7.45 -> 7.34 ms, but the extra rounding of Q
// doubles the error of dQ at a logit spread of 3: 0.0066 instead of 0.0033 relative L2, `tools/fa_accuracy.py` (round 2; git history).)

// K / V tiles: the same LDS-DMA ring as the forward kernel.
constexpr int kDqTile = kKvTile;

__global__ __launch_bounds__(kFaThreads, 2) void fa_bwd_dq_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const bf16_t* __restrict__ dout, int64_t sq, int64_t sk, int64_t sv, int64_t sdo, const int* __restrict__ cu,
    const int* __restrict__ tiles, int tcols, int nh, int nkv, float scale_log2e, float scale,
    const float* __restrict__ lse, const bf16_t* __restrict__ o, int64_t so, float* __restrict__ nl_out,
    float* __restrict__ nd_out, int64_t T, bf16_t* __restrict__ dq, int64_t sdq, const float* __restrict__ rcos,
    const float* __restrict__ rsin, int64_t rperiod) {
    __shared__ __attribute__((aligned(16))) char smem[3 * kDqTile];      // ring of (K tile | V tile), chunk ^= row & 7
    RPO_LAD_DECL;
    RPO_LAD(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, fr = lane & 15;
    const FaTile ft = fa_tile(tiles, tcols);
    if (ft.q0 >= (1 << 30)) return;
    const int seq = ft.seq, q0 = ft.q0;
    const int h = ft.h, hk = h / (nh / nkv);
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
        // The two row constants of the backward, NEGATED (they are the initial accumulators of the S and dP chains):
        // -lse / scale and -delta = -sum_d dO O.  Computed HERE (round 1 ran a separate fa_delta_kernel over O and dO: 0.32 ms
        // per call): the block holds its dO fragments anyway, the O fragments are 4 more 16-byte loads per lane; lane (g, fr)
        // sums its 16 hd columns of query fr, two shfl_xor finish the row.  Lane group 0 also writes both for the dK/dV kernel.
        float part = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const short8_t ov = ok ? *reinterpret_cast<const short8_t*>(o + (t0 + qi) * so + h * kFaHD + 32 * ks + 8 * g)
                                   : short8_t{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 8; ++e)
                part = fmaf(bf16_to_f32((bf16_t)bdo[n][ks][e]), bf16_to_f32((bf16_t)ov[e]), part);
        }
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        lq[n] = ok ? -lse[(int64_t)h * T + t0 + qi] / scale : 0.f;
        dl[n] = ok ? -part : 0.f;
        if (ok && g == 0) {
            nl_out[(int64_t)h * T + t0 + qi] = lq[n];
            nd_out[(int64_t)h * T + t0 + qi] = dl[n];
        }
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
    RPO_LAD(1);
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
            // two copies of the arithmetic, the branch OUTSIDE: written as `if (need_mask)` inside the element loop hipcc kept a
            // scalar branch, a key-index add and a compare per element on the unmasked path as well (6 instructions per score
            // instead of 3: ~100 per key tile)
            if (need_mask) {
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int qi = qw + 16 * n + fr;
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float pv = __builtin_amdgcn_exp2f(s[m][n][r] * scale_log2e);
                            const int key = kbase + 16 * m + r;
                            pv = (key > qi || key >= len || qi >= len) ? 0.f : pv;
                            s[m][n][r] = pv * dp[m][n][r];                // dS / scale (scale: epilogue)
                        }
                }
            } else {
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            s[m][n][r] = __builtin_amdgcn_exp2f(s[m][n][r] * scale_log2e) * dp[m][n][r];
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
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
    RPO_LAD(2);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int qi = qw + 16 * n + fr;
        if (qi >= len) continue;
        bf16_t* row = dq + (t0 + qi) * sdq + h * kFaHD;
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c][n] *= scale;
        if (rcos) {                                         // rows 16c + 4g + r pair with 16 (c + 2) + 4g + r
            const int64_t tr = ((t0 + qi) % rperiod) * (kFaHD / 2) + 4 * g;
#pragma unroll
            for (int c = 0; c < 2; ++c)
                inv_rope4(acc[c][n], acc[c + 2][n], *reinterpret_cast<const float4_t*>(rcos + tr + 16 * c),
                          *reinterpret_cast<const float4_t*>(rsin + tr + 16 * c));
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint2 w;
            w.x = pack_bf16(acc[c][n][0], acc[c][n][1]);
            w.y = pack_bf16(acc[c][n][2], acc[c][n][3]);
            *reinterpret_cast<uint2*>(row + 16 * c + 4 * g) = w;
        }
    }
    RPO_LAD_END(1, nkt);
}

// ------------------------------------------------------------------------------------------------------------------
// dQ at head_dim 64, ONE WAVE PER SIMD (round 5; rpo_flash_attn_bwd's q_block = 64): the construction of fa_fwd128w_kernel applied to
// fa_bwd_dq_kernel.  Block = 64 queries x the four q heads of one kv head (one head per wave: the staged K / V tiles serve four waves,
// the K / V fragments four query tiles instead of two: half the LDS reads per MFMA); dQ^T (64 registers), Q^T and dO^T fragments
// (32 + 32) and the K / V row fragments of the tile in flight live in the accumulator file; the key-tile loop is two generated asm
// statements per 32-key tile (tools/gen/gen_dq64w_body.py -> attention_dq64w_gen.inc, pipeline and register map in its docstring).
// No online softmax here (p = exp2(c S - lse log2e) from the saved lse), so no rescale statements: X1 / X2 and a mask.
// Q, dO and O come in as whole 128-byte rows through a 24-KiB LDS region per wave (delta = sum dO O from the fragments, as
// fa_bwd_dq_kernel computes it; both row constants are written for the dK/dV kernel); dQ goes out through the same region.
// K ring: eight 4-KiB images (a K tile is read by rows two iterations before its transposed reads); V ring: four.
// ------------------------------------------------------------------------------------------------------------------
#ifdef RPO_ONEWAVE64
#include "attention_dq64w_gen.inc"

__global__ __launch_bounds__(256, 1) void fa_bwd_dq64w_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const bf16_t* __restrict__ dout, int64_t sq, int64_t sk, int64_t sv, int64_t sdo, const int* __restrict__ cu,
    const int* __restrict__ tiles, int tcols, int nh, int nkv, float scale_log2e, float scale,
    const float* __restrict__ lse, const bf16_t* __restrict__ o, int64_t so, float* __restrict__ nl_out,
    float* __restrict__ nd_out, int64_t T, bf16_t* __restrict__ dq, int64_t sdq, const float* __restrict__ rcos,
    const float* __restrict__ rsin, int64_t rperiod) {
#if defined(RPO_FA_STAMP) && defined(RPO_FA_STAMP_DQ)
    unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ts0, ts1, ts2, ts3, ts4;
    RPO_STAMP(ts0);
#define DQS(VAR) RPO_STAMP(VAR)
#define DQS_ADD(I, A, B) st_acc[I] += (B) - (A)
#else
#define DQS(VAR)
#define DQS_ADD(I, A, B)
#endif
    constexpr int BN = 32, ROW_ = 2 * kFaHD, kImg = BN * ROW_;           // 4-KiB images
    constexpr int kVRing = 8 * kImg;                                     // K ring: slots 0-7 at byte 0; V ring: slots 0-3 behind it
    constexpr int kIo = 64 * ROW_;                                       // 64 rows of one of Q / dO / O (8 KiB)
    __shared__ __attribute__((aligned(16))) char smem[12 * kImg + 4 * 3 * kIo];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, fr = lane & 15;
    const FaTile ft = fa_tile(tiles, tcols);
    if (ft.q0 >= (1 << 30)) return;
    const int seq = ft.seq, q0 = ft.q0;
    const int h = (tcols == 3 ? ft.h : ft.h * 4) + wave, hk = h / (nh / nkv);
    const int64_t t0 = cu[seq];
    const int len = cu[seq + 1] - (int)t0;
    const int qw = q0;
    const int nkt = min(q0 + 63, len - 1) / BN + 1;                      // key tiles the block walks (every wave the same)
#define DQ_SWZ(R) ((R) & 7)
    // staging: a 4-KiB image = 4 pieces of 1 KiB (8 rows of 128 bytes); wave w carries piece w of the K and of the V image; lane l
    // carries row 8 w + (l >> 3), physical chunk l & 7 = logical chunk (l & 7) ^ (row & 7)
    const int srow = lane >> 3, lcol = lane & 7;
    const char* ksrc = reinterpret_cast<const char*>(k + t0 * sk + hk * kFaHD);
    const char* vsrc = reinterpret_cast<const char*>(v + t0 * sv + hk * kFaHD);
    const unsigned skb = (unsigned)sk * 2u, svb = (unsigned)sv * 2u;
    auto stage = [&](const char* src, unsigned stride_b, int kt, char* image) {
        const int trow = 8 * wave + srow;
        const unsigned lchunk = (unsigned)(lcol ^ DQ_SWZ(trow));
        const unsigned row = (unsigned)min(kt * BN + trow, len - 1);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (row * stride_b + lchunk * 16)),
                                         (__attribute__((address_space(3))) void*)(image + wave * 1024), 16, 0, 0);
    };
    // Q, dO, O: the wave's 64 rows of each (clamped to the sequence) as 8 + 8 + 8 LDS-DMA pieces, then K tiles 0-3 and V tiles 0-3
    char* const io = smem + 12 * kImg + wave * (3 * kIo);
    {
        const char* src[3] = {reinterpret_cast<const char*>(q + t0 * sq + h * kFaHD), reinterpret_cast<const char*>(dout + t0 * sdo + h * kFaHD),
                              reinterpret_cast<const char*>(o + t0 * so + h * kFaHD)};
        const unsigned strb[3] = {(unsigned)sq * 2u, (unsigned)sdo * 2u, (unsigned)so * 2u};
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int trow = 8 * u + srow;
                const unsigned lchunk = (unsigned)(lcol ^ DQ_SWZ(trow));
                const unsigned row = (unsigned)min(qw + trow, len - 1);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[a] + ((size_t)row * strb[a] + lchunk * 16)),
                                                 (__attribute__((address_space(3))) void*)(io + a * kIo + u * 1024), 16, 0, 0);
            }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j < nkt) { stage(ksrc, skb, j, smem + j * kImg); stage(vsrc, svb, j, smem + kVRing + j * kImg); }
    // the saved lse of the lane's four queries (clamped rows: duplicates that are never stored)
    float lq[4], dl[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) lq[n] = lse[(int64_t)h * T + t0 + min(qw + 16 * n + fr, len - 1)];
    RPO_DQ_INIT_ACC();                                                    // dQ^T = 0 under the loads' latency
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                         // (the first K / V tiles are everybody's)
    // fragments: lane (g, fr) = query 16 n + fr, hd 32 ks + 8 g .. + 7; delta = sum_d dO O over the row: the lane's 16 columns, then
    // the four lanes of the row
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        float part = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const unsigned fo = (16 * n + fr) * ROW_ + (((4 * ks + g) ^ DQ_SWZ(fr)) << 4);
            const short8_t bq = *reinterpret_cast<const short8_t*>(io + fo);
            const short8_t bdo = *reinterpret_cast<const short8_t*>(io + kIo + fo);
            const short8_t ov = *reinterpret_cast<const short8_t*>(io + 2 * kIo + fo);
#pragma unroll
            for (int e = 0; e < 8; ++e) part = fmaf(bf16_to_f32((bf16_t)bdo[e]), bf16_to_f32((bf16_t)ov[e]), part);
            const uint4_t wq = __builtin_bit_cast(uint4_t, bq), wd = __builtin_bit_cast(uint4_t, bdo);
            RPO_DQ_Q_TO_ACC(n, ks, wq);
            RPO_DQ_DO_TO_ACC(n, ks, wd);
        }
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        dl[n] = -part;
        const int qi = qw + 16 * n + fr;
        if (qi < len && g == 0) {
            nl_out[(int64_t)h * T + t0 + qi] = -lq[n] / scale;
            nd_out[(int64_t)h * T + t0 + qi] = dl[n];
        }
        lq[n] *= 1.4426950408889634f;                                     // p = exp2(c S - lse log2e)
    }
    // loop-invariant per-lane LDS addresses (slot 0 of either ring): K / V rows of k-step ks (row fr, chunk (4 ks + g) ^ (fr & 7)), K^T
    // blocks of hd tile c (row 4 g + qq, chunk (2 c + (pp >> 1)) ^ (row & 7))
    const int qq = fr >> 2, pp = fr & 3;
    const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    unsigned trk[4], krow[2], vrow[2];
#pragma unroll
    for (int c = 0; c < 4; ++c) trk[c] = smem_base + (4 * g + qq) * ROW_ + (((2 * c + (pp >> 1)) ^ DQ_SWZ(4 * g + qq)) << 4) + 8 * (pp & 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        krow[ks] = smem_base + fr * ROW_ + (((4 * ks + g) ^ DQ_SWZ(fr)) << 4);
        vrow[ks] = krow[ks] + kVRing;
    }
    unsigned pk, pv;                                                      // the wave's piece inside a tile: byte offsets from the tile's first row
    {
        const int trow = 8 * wave + srow;
        const unsigned lchunk = (unsigned)(lcol ^ DQ_SWZ(trow));
        pk = (unsigned)trow * skb + lchunk * 16;
        pv = (unsigned)trow * svb + lchunk * 16;
        asm volatile("" : "+v"(pk), "+v"(pv));
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(trk[c]));
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(krow[ks]), "+v"(vrow[ks]));
    auto needs_mask = [&](int j) { return (j * BN + BN - 1 > qw) || (j * BN + BN > len); };
#define DQ_MASK(GEN, J)                                                                                            \
    do {                                                                                                           \
        const int k0_ = (J) * BN;                                                                                  \
        const int d0_ = min(qw + fr, len - 1) - k0_ - 4 * g, d1_ = min(qw + 16 + fr, len - 1) - k0_ - 4 * g;       \
        const int d2_ = min(qw + 32 + fr, len - 1) - k0_ - 4 * g, d3_ = min(qw + 48 + fr, len - 1) - k0_ - 4 * g;  \
        RPO_DQ_MASK_##GEN(d0_, d1_, d2_, d3_);                                                                     \
    } while (0)
    // from here on v[64:215] and a[0:159] belong to the generated statements
    RPO_DQ_INIT(lq[0], lq[1], lq[2], lq[3], dl[0], dl[1], dl[2], dl[3]);
    RPO_DQ_READ0(krow[0], krow[1], vrow[0], vrow[1]);
    RPO_DQ_CHAIN0();
    if (needs_mask(0)) DQ_MASK(A, 0);
    RPO_DQ_PRE0(krow[0], krow[1], scale_log2e);
#define DQ_ITER(S, NXT, KT)                                                                                        \
    do {                                                                                                           \
        const int kt_ = (KT);                                                                                      \
        DQS(ts1);                                                                                                  \
        /* K(kt + 2) and V(kt + 1) landed (staged two / three iterations ago); the previous iteration's two pieces may fly.  The  */ \
        /* barrier also orders this iteration's DMA (V(kt + 4) -> the slot of V(kt)) behind every wave's reads of V(kt)           */ \
        if (kt_ + 3 < nkt) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                        \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                      \
        DQS(ts2);                                                                                                  \
        DQS_ADD(0, ts1, ts2);                                             /* 0: ring wait */                        \
        __builtin_amdgcn_s_barrier();                                                                              \
        DQS(ts3);                                                                                                  \
        DQS_ADD(1, ts2, ts3);                                             /* 1: barrier */                          \
        const bool instream_ = (kt_ + 5) * BN <= len && kt_ + 4 < nkt && kt_ + 1 < nkt;                            \
        if (!instream_ && kt_ + 4 < nkt) {                                                                         \
            stage(ksrc, skb, kt_ + 4, smem + ((kt_ + 4) & 7) * kImg);                                              \
            stage(vsrc, svb, kt_ + 4, smem + kVRing + (kt_ & 3) * kImg);                                           \
        }                                                                                                          \
        DQS(ts2);                                                                                                  \
        DQS_ADD(2, ts3, ts2);                                             /* 2: hipcc staging */                    \
        if (kt_ + 1 < nkt) {                                                                                       \
            RPO_DQ_X1_S##S(trk[0], trk[1], trk[2], trk[3], vrow[0], vrow[1]);                                      \
            DQS(ts3);                                                                                              \
            DQS_ADD(3, ts2, ts3);                                         /* 3: X1 */                               \
            if (needs_mask(kt_ + 1)) DQ_MASK(NXT, kt_ + 1);                                                        \
            DQS(ts4);                                                                                              \
            DQS_ADD(4, ts3, ts4);                                         /* 4: mask */                             \
            if (instream_) {                                                                                       \
                const char* sk_ = ksrc + (size_t)(kt_ + 4) * BN * skb;                                             \
                const char* sv_ = vsrc + (size_t)(kt_ + 4) * BN * svb;                                             \
                const unsigned mk_ = smem_base + ((kt_ + 4) & 7) * kImg + wave * 1024;                             \
                const unsigned mv_ = smem_base + kVRing + (kt_ & 3) * kImg + wave * 1024;                          \
                RPO_DQ_X2D_S##S(krow[0], krow[1], scale_log2e, pk, pv, sk_, sv_, mk_, mv_);                        \
            } else {                                                                                               \
                RPO_DQ_X2_S##S(krow[0], krow[1], scale_log2e);                                                     \
            }                                                                                                      \
            DQS(ts3);                                                                                              \
            DQS_ADD(5, ts4, ts3);                                         /* 5: X2 */                               \
            DQS_ADD(7, 0, 1);                                             /* 7: full iterations */                  \
        } else {                                                                                                   \
            RPO_DQ_X1L_S##S(trk[0], trk[1], trk[2], trk[3], vrow[0], vrow[1]);                                     \
            RPO_DQ_X2L();                                                                                          \
        }                                                                                                          \
    } while (0)
    DQS(ts1);
    DQS_ADD(6, ts0, ts1);                                                 /* 6: prologue */
    for (int kt = 0; kt < nkt; kt += 8) {
        DQ_ITER(0, B, kt);
        if (kt + 1 < nkt) DQ_ITER(1, A, kt + 1);
        if (kt + 2 < nkt) DQ_ITER(2, B, kt + 2);
        if (kt + 3 < nkt) DQ_ITER(3, A, kt + 3);
        if (kt + 4 < nkt) DQ_ITER(4, B, kt + 4);
        if (kt + 5 < nkt) DQ_ITER(5, A, kt + 5);
        if (kt + 6 < nkt) DQ_ITER(6, B, kt + 6);
        if (kt + 7 < nkt) DQ_ITER(7, A, kt + 7);
    }
#undef DQ_ITER
#undef DQ_MASK
    DQS(ts4);
    // ---- epilogue: dQ[q][16c + 4g + r] = scale dQ^T (inverse rotation: rows 16 c + 4 g + r pair with 16 (c + 2) + 4 g + r) into the
    // wave's LDS region, whole rows out of it
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int qi = qw + 16 * n + fr;
        float4_t acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float x0, x1, x2, x3;
            RPO_DQ_READ_DQ(c, n, x0, x1, x2, x3);
            acc[c] = float4_t{x0 * scale, x1 * scale, x2 * scale, x3 * scale};
        }
        if (rcos) {
            const int64_t tr = ((t0 + min(qi, len - 1)) % rperiod) * (kFaHD / 2) + 4 * g;
#pragma unroll
            for (int c = 0; c < 2; ++c)
                inv_rope4(acc[c], acc[c + 2], *reinterpret_cast<const float4_t*>(rcos + tr + 16 * c),
                          *reinterpret_cast<const float4_t*>(rsin + tr + 16 * c));
        }
        char* lrow = io + (16 * n + fr) * ROW_ + 8 * (g & 1);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint2_t w;
            w.x = pack_bf16(acc[c][0], acc[c][1]);
            w.y = pack_bf16(acc[c][2], acc[c][3]);
            *reinterpret_cast<uint2_t*>(lrow + (((2 * c + (g >> 1)) ^ DQ_SWZ(fr)) << 4)) = w;
        }
    }
    {
        bf16_t* dbase = dq + (t0 + qw) * sdq + h * kFaHD + 8 * lcol;
        int srow_e = lane >> 3;
        asm volatile("" : "+v"(srow_e));
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int trow = 8 * u + srow_e;
            const uint4_t w = *reinterpret_cast<const uint4_t*>(io + trow * ROW_ + ((lcol ^ DQ_SWZ(trow)) << 4));
            if (qw + trow < len) *reinterpret_cast<uint4_t*>(dbase + (int64_t)trow * sdq) = w;
        }
    }
#if defined(RPO_FA_STAMP) && defined(RPO_FA_STAMP_DQ)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DQS(ts3);
    DQS_ADD(8, ts4, ts3);                                                 /* 8: epilogue, stores landed */
    if (lane == 0)
        for (int i = 0; i < 16; ++i) atomicAdd(&g_fa_stamp[wave * 16 + i], st_acc[i]);
#endif
#undef DQ_SWZ
#undef DQS
#undef DQS_ADD
}
#endif  // RPO_ONEWAVE64

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

// ---- dK / dV, one wave per SIMD ------------------------------------------------------------------------------------------
// Block = 4 waves = 256 keys of one (sequence, kv head); wave w owns keys [kb0 + 64 w, + 64) entirely: its K / V fragments
// (64 VGPRs) and its dK^T / dV^T accumulators (128 registers) stay in the register file for the whole sweep over the
// group's q heads x 32-row query slices, so there is no cross-wave sum and nothing but the 8 KB Q / dO slice (+ its 64 row
// constants) is shared.  One wave per SIMD (up to 512 registers): the two waves of the 8-wave kernel above leave every
// barrier together and then want the same unit at the same time (profiles/r01_fa_dkdv_stamps.md); here a SIMD's MFMA, VALU
// and LDS work belong to ONE instruction stream, and the bytes through the L1 -> LDS path per MFMA drop 4x.
// Slices arrive by global_load_lds into a ring of 8 images, 4 slices ahead, one raw barrier per slice, counted vmcnt.
// Accumulator-file map (literal registers in the asm statements; hipcc must not touch AGPRs in this kernel -- its
// resource line must read 0 spills and the .s no v_accvgpr_* outside ASMSTART / ASMEND):
//   dva[c][n] = a[16c + 4n ..+3] (0..63), dka[c][n] = a[64 + 16c + 4n ..+3], bk[n][ks] = a[128 + 8n + 4ks ..+3],
//   bv[n][ks] = a[160 + 8n + 4ks ..+3].
constexpr int kSl = 32;                                      // query rows per slice
constexpr int kSlImg = 2 * kSl * 128 + 2 * kSl * 4;          // Q | dO | -lse/scale | -delta = 8448 B
constexpr int kSlRing = 8, kSlAhead = 4;
constexpr int kDkdv4Lds = kSlRing * kSlImg;                  // 67584 B

// MFMA statements of the one-wave-per-SIMD dK/dV kernel with literal accumulator-file registers (map: see the kernel)
#define RPO_D4_M1_KS0(M, AQ, AD)                                                                              \
    asm volatile(                                                                                            \
        "s_nop 1\n\t" \
        "v_mfma_f32_16x16x32_bf16 %0, %8, a[128:131], %0\n\t" \
        "v_mfma_f32_16x16x32_bf16 %4, %9, a[160:163], %4\n\t" \
        "v_mfma_f32_16x16x32_bf16 %1, %8, a[136:139], %1\n\t" \
        "v_mfma_f32_16x16x32_bf16 %5, %9, a[168:171], %5\n\t" \
        "v_mfma_f32_16x16x32_bf16 %2, %8, a[144:147], %2\n\t" \
        "v_mfma_f32_16x16x32_bf16 %6, %9, a[176:179], %6\n\t" \
        "v_mfma_f32_16x16x32_bf16 %3, %8, a[152:155], %3\n\t" \
        "v_mfma_f32_16x16x32_bf16 %7, %9, a[184:187], %7" \
        : "+v"(s[M][0]), "+v"(s[M][1]), "+v"(s[M][2]), "+v"(s[M][3]), "+v"(dp[M][0]), "+v"(dp[M][1]),         \
          "+v"(dp[M][2]), "+v"(dp[M][3])                                                                        \
        : "v"(AQ), "v"(AD)                                                                                   \
        : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191")
#define RPO_D4_M1_KS1(M, AQ, AD)                                                                              \
    asm volatile(                                                                                            \
        "s_nop 1\n\t" \
        "v_mfma_f32_16x16x32_bf16 %0, %8, a[132:135], %0\n\t" \
        "v_mfma_f32_16x16x32_bf16 %4, %9, a[164:167], %4\n\t" \
        "v_mfma_f32_16x16x32_bf16 %1, %8, a[140:143], %1\n\t" \
        "v_mfma_f32_16x16x32_bf16 %5, %9, a[172:175], %5\n\t" \
        "v_mfma_f32_16x16x32_bf16 %2, %8, a[148:151], %2\n\t" \
        "v_mfma_f32_16x16x32_bf16 %6, %9, a[180:183], %6\n\t" \
        "v_mfma_f32_16x16x32_bf16 %3, %8, a[156:159], %3\n\t" \
        "v_mfma_f32_16x16x32_bf16 %7, %9, a[188:191], %7" \
        : "+v"(s[M][0]), "+v"(s[M][1]), "+v"(s[M][2]), "+v"(s[M][3]), "+v"(dp[M][0]), "+v"(dp[M][1]),         \
          "+v"(dp[M][2]), "+v"(dp[M][3])                                                                        \
        : "v"(AQ), "v"(AD)                                                                                   \
        : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191")
#define RPO_D4_M2_N0(ATD, ATQ, PF, DS)                                                                        \
    asm volatile(                                                                                            \
        "s_nop 1\n\t" \
        "v_mfma_f32_16x16x32_bf16 a[0:3], %0, %8, a[0:3]\n\t" \
        "v_mfma_f32_16x16x32_bf16 a[64:67], %4, %9, a[64:67]\n\t" \
        "v_mfma_f32_16x16x32_bf16 a[16:19], %1, %8, a[16:19]\n\t" \
        "v_mfma_f32_16x16x32_bf16 a[80:83], %5, %9, a[80:83]\n\t" \
        "v_mfma_f32_16x16x32_bf16 a[32:35], %2, %8, a[32:35]\n\t" \
        "v_mfma_f32_16x16x32_bf16 a[96:99], %6, %9, a[96:99]\n\t" \
        "v_mfma_f32_16x16x32_bf16 a[48:51], %3, %8, a[48:51]\n\t" \
        "v_mfma_f32_16x16x32_bf16 a[112:115], %7, %9, a[112:115]" \
        :                                                                                                    \
        : "v"(ATD[0]), "v"(ATD[1]), "v"(ATD[2]), "v"(ATD[3]), "v"(ATQ[0]), "v"(ATQ[1]), "v"(ATQ[2]), "v"(ATQ[3]), \
          "v"(PF[0]), "v"(DS[0])                                                                            \
        : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191")
#define RPO_D4_M2_N1(ATD, ATQ, PF, DS)                                                                        \
    asm volatile(                                                                                            \
        "s_nop 1\n\t" \
        "v_mfma_f32_16x16x32_bf16 a[4:7], %0, %8, a[4:7]\n\t" \
        "v_mfma_f32_16x16x32_bf16 a[68:71], %4, %9, a[68:71]\n\t" \
        "v_mfma_f32_16x16x32_bf16 a[20:23], %1, %8, a[20:23]\n\t" \
        "v_mfma_f32_16x16x32_bf16 a[84:87], %5, %9, a[84:87]\n\t" \
        "v_mfma_f32_16x16x32_bf16 a[36:39], %2, %8, a[36:39]\n\t" \
        "v_mfma_f32_16x16x32_bf16 a[100:103], %6, %9, a[100:103]\n\t" \
        "v_mfma_f32_16x16x32_bf16 a[52:55], %3, %8, a[52:55]\n\t" \
        "v_mfma_f32_16x16x32_bf16 a[116:119], %7, %9, a[116:119]" \
        :                                                                                                    \
        : "v"(ATD[0]), "v"(ATD[1]), "v"(ATD[2]), "v"(ATD[3]), "v"(ATQ[0]), "v"(ATQ[1]), "v"(ATQ[2]), "v"(ATQ[3]), \
          "v"(PF[1]), "v"(DS[1])                                                                            \
        : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191")
#define RPO_D4_M2_N2(ATD, ATQ, PF, DS)                                                                        \
    asm volatile(                                                                                            \
        "s_nop 1\n\t" \
        "v_mfma_f32_16x16x32_bf16 a[8:11], %0, %8, a[8:11]\n\t" \
        "v_mfma_f32_16x16x32_bf16 a[72:75], %4, %9, a[72:75]\n\t" \
        "v_mfma_f32_16x16x32_bf16 a[24:27], %1, %8, a[24:27]\n\t" \
        "v_mfma_f32_16x16x32_bf16 a[88:91], %5, %9, a[88:91]\n\t" \
        "v_mfma_f32_16x16x32_bf16 a[40:43], %2, %8, a[40:43]\n\t" \
        "v_mfma_f32_16x16x32_bf16 a[104:107], %6, %9, a[104:107]\n\t" \
        "v_mfma_f32_16x16x32_bf16 a[56:59], %3, %8, a[56:59]\n\t" \
        "v_mfma_f32_16x16x32_bf16 a[120:123], %7, %9, a[120:123]" \
        :                                                                                                    \
        : "v"(ATD[0]), "v"(ATD[1]), "v"(ATD[2]), "v"(ATD[3]), "v"(ATQ[0]), "v"(ATQ[1]), "v"(ATQ[2]), "v"(ATQ[3]), \
          "v"(PF[2]), "v"(DS[2])                                                                            \
        : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191")
#define RPO_D4_M2_N3(ATD, ATQ, PF, DS)                                                                        \
    asm volatile(                                                                                            \
        "s_nop 1\n\t" \
        "v_mfma_f32_16x16x32_bf16 a[12:15], %0, %8, a[12:15]\n\t" \
        "v_mfma_f32_16x16x32_bf16 a[76:79], %4, %9, a[76:79]\n\t" \
        "v_mfma_f32_16x16x32_bf16 a[28:31], %1, %8, a[28:31]\n\t" \
        "v_mfma_f32_16x16x32_bf16 a[92:95], %5, %9, a[92:95]\n\t" \
        "v_mfma_f32_16x16x32_bf16 a[44:47], %2, %8, a[44:47]\n\t" \
        "v_mfma_f32_16x16x32_bf16 a[108:111], %6, %9, a[108:111]\n\t" \
        "v_mfma_f32_16x16x32_bf16 a[60:63], %3, %8, a[60:63]\n\t" \
        "v_mfma_f32_16x16x32_bf16 a[124:127], %7, %9, a[124:127]" \
        :                                                                                                    \
        : "v"(ATD[0]), "v"(ATD[1]), "v"(ATD[2]), "v"(ATD[3]), "v"(ATQ[0]), "v"(ATQ[1]), "v"(ATQ[2]), "v"(ATQ[3]), \
          "v"(PF[3]), "v"(DS[3])                                                                            \
        : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191")
#ifdef RPO_D4_EXP_DQ_ATOMICS
// PRICING EXPERIMENT (round 4, never shipped; tools/exp/build_variant.sh dqatomics -DRPO_D4_EXP_DQ_ATOMICS): what the dQ sum of a
// one-kernel, five-product backward (cdna_hip_programming.md, Appendix B 'Attention backward': dQ by global_atomic_add_f32 from the
// 256-key block) would add to THIS kernel before a single extra MFMA: after every slice each wave issues the block's share of
// the f32 adds -- 8 query rows x 256 contiguous bytes (one dQ row of the head) -- into a [T, nh, 64] float buffer, adding 0.0
// (dK / dV stay what they are, the parity tests still pass); the LDS-DMA ring's counted waits are widened by the 24 atomics
// that sit between a stage and its wait.
__device__ float* g_exp_dq32 = nullptr;
#endif

// generated by tools/gen/gen_dkdv4_body.py (register map and operand list there); 238 instructions
#define RPO_D4_SLICE_BODY_LOAD(RA0, RA1, LRD, TP0, TP1, TP2, TP3, SCL, NRA0, NRA1, NLRD)                    \
    asm volatile(                                                                                               \
        "ds_read_b128 v[160:163], %2\n\t"                                                                             \
        "ds_read_b128 v[168:171], %2 offset:128\n\t"                                                                  \
        "ds_read_b128 v[128:131], %0\n\t"                                                                             \
        "ds_read_b128 v[132:135], %0 offset:4096\n\t"                                                                 \
        "ds_read_b128 v[164:167], %2 offset:64\n\t"                                                                   \
        "ds_read_b128 v[172:175], %2 offset:192\n\t"                                                                  \
        "ds_read_b128 v[136:139], %0 offset:2048\n\t"                                                                 \
        "ds_read_b128 v[140:143], %0 offset:6144\n\t"                                                                 \
        "ds_read_b128 v[144:147], %1\n\t"                                                                             \
        "ds_read_b128 v[148:151], %1 offset:4096\n\t"                                                                 \
        "ds_read_b128 v[152:155], %1 offset:2048\n\t"                                                                 \
        "ds_read_b128 v[156:159], %1 offset:6144\n\t"                                                                 \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                    \
        "v_mfma_f32_16x16x32_bf16 v[64:67], v[128:131], a[128:131], v[160:163]\n\t"                                   \
        "ds_read_b64_tr_b16 v[176:177], %3 offset:4096\n\t"                                                           \
        "v_mfma_f32_16x16x32_bf16 v[80:83], v[136:139], a[128:131], v[164:167]\n\t"                                   \
        "ds_read_b64_tr_b16 v[178:179], %3 offset:6144\n\t"                                                           \
        "v_mfma_f32_16x16x32_bf16 v[96:99], v[132:135], a[160:163], v[168:171]\n\t"                                   \
        "ds_read_b64_tr_b16 v[192:193], %3\n\t"                                                                       \
        "v_mfma_f32_16x16x32_bf16 v[112:115], v[140:143], a[160:163], v[172:175]\n\t"                                 \
        "ds_read_b64_tr_b16 v[194:195], %3 offset:2048\n\t"                                                           \
        "v_mfma_f32_16x16x32_bf16 v[64:67], v[144:147], a[132:135], v[64:67]\n\t"                                     \
        "ds_read_b64_tr_b16 v[180:181], %4 offset:4096\n\t"                                                           \
        "v_mfma_f32_16x16x32_bf16 v[80:83], v[152:155], a[132:135], v[80:83]\n\t"                                     \
        "ds_read_b64_tr_b16 v[182:183], %4 offset:6144\n\t"                                                           \
        "v_mfma_f32_16x16x32_bf16 v[96:99], v[148:151], a[164:167], v[96:99]\n\t"                                     \
        "ds_read_b64_tr_b16 v[196:197], %4\n\t"                                                                       \
        "v_mfma_f32_16x16x32_bf16 v[112:115], v[156:159], a[164:167], v[112:115]\n\t"                                 \
        "ds_read_b64_tr_b16 v[198:199], %4 offset:2048\n\t"                                                           \
        "v_mfma_f32_16x16x32_bf16 v[68:71], v[128:131], a[136:139], v[160:163]\n\t"                                   \
        "ds_read_b64_tr_b16 v[184:185], %5 offset:4096\n\t"                                                           \
        "v_mfma_f32_16x16x32_bf16 v[84:87], v[136:139], a[136:139], v[164:167]\n\t"                                   \
        "ds_read_b64_tr_b16 v[186:187], %5 offset:6144\n\t"                                                           \
        "v_mfma_f32_16x16x32_bf16 v[100:103], v[132:135], a[168:171], v[168:171]\n\t"                                 \
        "ds_read_b64_tr_b16 v[200:201], %5\n\t"                                                                       \
        "v_mul_f32 v64, %7, v64\n\t"                                                                                  \
        "v_mul_f32 v65, %7, v65\n\t"                                                                                  \
        "v_mul_f32 v66, %7, v66\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 v[116:119], v[140:143], a[168:171], v[172:175]\n\t"                                 \
        "ds_read_b64_tr_b16 v[202:203], %5 offset:2048\n\t"                                                           \
        "v_mul_f32 v67, %7, v67\n\t"                                                                                  \
        "v_mul_f32 v80, %7, v80\n\t"                                                                                  \
        "v_mul_f32 v81, %7, v81\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 v[68:71], v[144:147], a[140:143], v[68:71]\n\t"                                     \
        "ds_read_b64_tr_b16 v[188:189], %6 offset:4096\n\t"                                                           \
        "v_mul_f32 v82, %7, v82\n\t"                                                                                  \
        "v_mul_f32 v83, %7, v83\n\t"                                                                                  \
        "v_exp_f32 v64, v64\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[84:87], v[152:155], a[140:143], v[84:87]\n\t"                                     \
        "ds_read_b64_tr_b16 v[190:191], %6 offset:6144\n\t"                                                           \
        "v_exp_f32 v65, v65\n\t"                                                                                      \
        "v_exp_f32 v66, v66\n\t"                                                                                      \
        "v_exp_f32 v67, v67\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[100:103], v[148:151], a[172:175], v[100:103]\n\t"                                 \
        "ds_read_b64_tr_b16 v[204:205], %6\n\t"                                                                       \
        "v_exp_f32 v80, v80\n\t"                                                                                      \
        "v_exp_f32 v81, v81\n\t"                                                                                      \
        "v_exp_f32 v82, v82\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[116:119], v[156:159], a[172:175], v[116:119]\n\t"                                 \
        "ds_read_b64_tr_b16 v[206:207], %6 offset:2048\n\t"                                                           \
        "v_exp_f32 v83, v83\n\t"                                                                                      \
        "v_mul_f32 v96, v64, v96\n\t"                                                                                 \
        "v_mul_f32 v97, v65, v97\n\t"                                                                                 \
        "v_mfma_f32_16x16x32_bf16 v[72:75], v[128:131], a[144:147], v[160:163]\n\t"                                   \
        "v_mul_f32 v98, v66, v98\n\t"                                                                                 \
        "v_mul_f32 v99, v67, v99\n\t"                                                                                 \
        "v_mul_f32 v112, v80, v112\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 v[88:91], v[136:139], a[144:147], v[164:167]\n\t"                                   \
        "v_mul_f32 v113, v81, v113\n\t"                                                                               \
        "v_mul_f32 v114, v82, v114\n\t"                                                                               \
        "v_mul_f32 v115, v83, v115\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 v[104:107], v[132:135], a[176:179], v[168:171]\n\t"                                 \
        "v_cvt_pk_bf16_f32 v208, v64, v65\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v209, v66, v67\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v210, v80, v81\n\t"                                                                        \
        "v_mfma_f32_16x16x32_bf16 v[120:123], v[140:143], a[176:179], v[172:175]\n\t"                                 \
        "v_cvt_pk_bf16_f32 v211, v82, v83\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v224, v96, v97\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v225, v98, v99\n\t"                                                                        \
        "v_mfma_f32_16x16x32_bf16 v[72:75], v[144:147], a[148:151], v[72:75]\n\t"                                     \
        "v_cvt_pk_bf16_f32 v226, v112, v113\n\t"                                                                      \
        "v_cvt_pk_bf16_f32 v227, v114, v115\n\t"                                                                      \
        "v_mul_f32 v68, %7, v68\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 v[88:91], v[152:155], a[148:151], v[88:91]\n\t"                                     \
        "v_mul_f32 v69, %7, v69\n\t"                                                                                  \
        "v_mul_f32 v70, %7, v70\n\t"                                                                                  \
        "v_mul_f32 v71, %7, v71\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 v[104:107], v[148:151], a[180:183], v[104:107]\n\t"                                 \
        "v_mul_f32 v84, %7, v84\n\t"                                                                                  \
        "v_mul_f32 v85, %7, v85\n\t"                                                                                  \
        "v_mul_f32 v86, %7, v86\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 v[120:123], v[156:159], a[180:183], v[120:123]\n\t"                                 \
        "v_mul_f32 v87, %7, v87\n\t"                                                                                  \
        "v_exp_f32 v68, v68\n\t"                                                                                      \
        "v_exp_f32 v69, v69\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[76:79], v[128:131], a[152:155], v[160:163]\n\t"                                   \
        "v_exp_f32 v70, v70\n\t"                                                                                      \
        "v_exp_f32 v71, v71\n\t"                                                                                      \
        "v_exp_f32 v84, v84\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[92:95], v[136:139], a[152:155], v[164:167]\n\t"                                   \
        "v_exp_f32 v85, v85\n\t"                                                                                      \
        "v_exp_f32 v86, v86\n\t"                                                                                      \
        "v_exp_f32 v87, v87\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[108:111], v[132:135], a[184:187], v[168:171]\n\t"                                 \
        "v_mul_f32 v100, v68, v100\n\t"                                                                               \
        "v_mul_f32 v101, v69, v101\n\t"                                                                               \
        "v_mul_f32 v102, v70, v102\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 v[124:127], v[140:143], a[184:187], v[172:175]\n\t"                                 \
        "v_mul_f32 v103, v71, v103\n\t"                                                                               \
        "v_mul_f32 v116, v84, v116\n\t"                                                                               \
        "v_mul_f32 v117, v85, v117\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 v[76:79], v[144:147], a[156:159], v[76:79]\n\t"                                     \
        "v_mul_f32 v118, v86, v118\n\t"                                                                               \
        "v_mul_f32 v119, v87, v119\n\t"                                                                               \
        "v_cvt_pk_bf16_f32 v212, v68, v69\n\t"                                                                        \
        "v_mfma_f32_16x16x32_bf16 v[92:95], v[152:155], a[156:159], v[92:95]\n\t"                                     \
        "v_cvt_pk_bf16_f32 v213, v70, v71\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v214, v84, v85\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v215, v86, v87\n\t"                                                                        \
        "v_mfma_f32_16x16x32_bf16 v[108:111], v[148:151], a[188:191], v[108:111]\n\t"                                 \
        "v_cvt_pk_bf16_f32 v228, v100, v101\n\t"                                                                      \
        "v_cvt_pk_bf16_f32 v229, v102, v103\n\t"                                                                      \
        "v_cvt_pk_bf16_f32 v230, v116, v117\n\t"                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[124:127], v[156:159], a[188:191], v[124:127]\n\t"                                 \
        "v_cvt_pk_bf16_f32 v231, v118, v119\n\t"                                                                      \
        "v_mul_f32 v72, %7, v72\n\t"                                                                                  \
        "v_mul_f32 v73, %7, v73\n\t"                                                                                  \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                    \
        "s_nop 1\n\t"                                                                                                 \
        "v_mfma_f32_16x16x32_bf16 a[0:3], v[176:179], v[208:211], a[0:3]\n\t"                                         \
        "ds_read_b128 v[160:163], %10\n\t"                                                                            \
        "v_mul_f32 v74, %7, v74\n\t"                                                                                  \
        "v_mul_f32 v75, %7, v75\n\t"                                                                                  \
        "v_mul_f32 v88, %7, v88\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 a[64:67], v[192:195], v[224:227], a[64:67]\n\t"                                     \
        "ds_read_b128 v[168:171], %10 offset:128\n\t"                                                                 \
        "v_mul_f32 v89, %7, v89\n\t"                                                                                  \
        "v_mul_f32 v90, %7, v90\n\t"                                                                                  \
        "v_mul_f32 v91, %7, v91\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 a[16:19], v[180:183], v[208:211], a[16:19]\n\t"                                     \
        "ds_read_b128 v[128:131], %8\n\t"                                                                             \
        "v_exp_f32 v72, v72\n\t"                                                                                      \
        "v_exp_f32 v73, v73\n\t"                                                                                      \
        "v_exp_f32 v74, v74\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[80:83], v[196:199], v[224:227], a[80:83]\n\t"                                     \
        "ds_read_b128 v[132:135], %8 offset:4096\n\t"                                                                 \
        "v_exp_f32 v75, v75\n\t"                                                                                      \
        "v_exp_f32 v88, v88\n\t"                                                                                      \
        "v_exp_f32 v89, v89\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[32:35], v[184:187], v[208:211], a[32:35]\n\t"                                     \
        "ds_read_b128 v[164:167], %10 offset:64\n\t"                                                                  \
        "v_exp_f32 v90, v90\n\t"                                                                                      \
        "v_exp_f32 v91, v91\n\t"                                                                                      \
        "v_mul_f32 v104, v72, v104\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 a[96:99], v[200:203], v[224:227], a[96:99]\n\t"                                     \
        "ds_read_b128 v[172:175], %10 offset:192\n\t"                                                                 \
        "v_mul_f32 v105, v73, v105\n\t"                                                                               \
        "v_mul_f32 v106, v74, v106\n\t"                                                                               \
        "v_mul_f32 v107, v75, v107\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 a[48:51], v[188:191], v[208:211], a[48:51]\n\t"                                     \
        "ds_read_b128 v[136:139], %8 offset:2048\n\t"                                                                 \
        "v_mul_f32 v120, v88, v120\n\t"                                                                               \
        "v_mul_f32 v121, v89, v121\n\t"                                                                               \
        "v_mul_f32 v122, v90, v122\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 a[112:115], v[204:207], v[224:227], a[112:115]\n\t"                                 \
        "ds_read_b128 v[140:143], %8 offset:6144\n\t"                                                                 \
        "v_mul_f32 v123, v91, v123\n\t"                                                                               \
        "v_cvt_pk_bf16_f32 v216, v72, v73\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v217, v74, v75\n\t"                                                                        \
        "s_nop 1\n\t"                                                                                                 \
        "v_mfma_f32_16x16x32_bf16 a[4:7], v[176:179], v[212:215], a[4:7]\n\t"                                         \
        "ds_read_b128 v[144:147], %9\n\t"                                                                             \
        "v_cvt_pk_bf16_f32 v218, v88, v89\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v219, v90, v91\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v232, v104, v105\n\t"                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[68:71], v[192:195], v[228:231], a[68:71]\n\t"                                     \
        "ds_read_b128 v[148:151], %9 offset:4096\n\t"                                                                 \
        "v_cvt_pk_bf16_f32 v233, v106, v107\n\t"                                                                      \
        "v_cvt_pk_bf16_f32 v234, v120, v121\n\t"                                                                      \
        "v_cvt_pk_bf16_f32 v235, v122, v123\n\t"                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[20:23], v[180:183], v[212:215], a[20:23]\n\t"                                     \
        "ds_read_b128 v[152:155], %9 offset:2048\n\t"                                                                 \
        "v_mul_f32 v76, %7, v76\n\t"                                                                                  \
        "v_mul_f32 v77, %7, v77\n\t"                                                                                  \
        "v_mul_f32 v78, %7, v78\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 a[84:87], v[196:199], v[228:231], a[84:87]\n\t"                                     \
        "ds_read_b128 v[156:159], %9 offset:6144\n\t"                                                                 \
        "v_mul_f32 v79, %7, v79\n\t"                                                                                  \
        "v_mul_f32 v92, %7, v92\n\t"                                                                                  \
        "v_mul_f32 v93, %7, v93\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 a[36:39], v[184:187], v[212:215], a[36:39]\n\t"                                     \
        "v_mul_f32 v94, %7, v94\n\t"                                                                                  \
        "v_mul_f32 v95, %7, v95\n\t"                                                                                  \
        "v_exp_f32 v76, v76\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[100:103], v[200:203], v[228:231], a[100:103]\n\t"                                 \
        "v_exp_f32 v77, v77\n\t"                                                                                      \
        "v_exp_f32 v78, v78\n\t"                                                                                      \
        "v_exp_f32 v79, v79\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[52:55], v[188:191], v[212:215], a[52:55]\n\t"                                     \
        "v_exp_f32 v92, v92\n\t"                                                                                      \
        "v_exp_f32 v93, v93\n\t"                                                                                      \
        "v_exp_f32 v94, v94\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[116:119], v[204:207], v[228:231], a[116:119]\n\t"                                 \
        "v_exp_f32 v95, v95\n\t"                                                                                      \
        "v_mul_f32 v108, v76, v108\n\t"                                                                               \
        "v_mul_f32 v109, v77, v109\n\t"                                                                               \
        "s_nop 1\n\t"                                                                                                 \
        "v_mfma_f32_16x16x32_bf16 a[8:11], v[176:179], v[216:219], a[8:11]\n\t"                                       \
        "v_mul_f32 v110, v78, v110\n\t"                                                                               \
        "v_mul_f32 v111, v79, v111\n\t"                                                                               \
        "v_mul_f32 v124, v92, v124\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 a[72:75], v[192:195], v[232:235], a[72:75]\n\t"                                     \
        "v_mul_f32 v125, v93, v125\n\t"                                                                               \
        "v_mul_f32 v126, v94, v126\n\t"                                                                               \
        "v_mul_f32 v127, v95, v127\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 a[24:27], v[180:183], v[216:219], a[24:27]\n\t"                                     \
        "v_cvt_pk_bf16_f32 v220, v76, v77\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v221, v78, v79\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v222, v92, v93\n\t"                                                                        \
        "v_mfma_f32_16x16x32_bf16 a[88:91], v[196:199], v[232:235], a[88:91]\n\t"                                     \
        "v_cvt_pk_bf16_f32 v223, v94, v95\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v236, v108, v109\n\t"                                                                      \
        "v_cvt_pk_bf16_f32 v237, v110, v111\n\t"                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[40:43], v[184:187], v[216:219], a[40:43]\n\t"                                     \
        "v_cvt_pk_bf16_f32 v238, v124, v125\n\t"                                                                      \
        "v_cvt_pk_bf16_f32 v239, v126, v127\n\t"                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[104:107], v[200:203], v[232:235], a[104:107]\n\t"                                 \
        "v_mfma_f32_16x16x32_bf16 a[56:59], v[188:191], v[216:219], a[56:59]\n\t"                                     \
        "v_mfma_f32_16x16x32_bf16 a[120:123], v[204:207], v[232:235], a[120:123]\n\t"                                 \
        "s_nop 1\n\t"                                                                                                 \
        "v_mfma_f32_16x16x32_bf16 a[12:15], v[176:179], v[220:223], a[12:15]\n\t"                                     \
        "v_mfma_f32_16x16x32_bf16 a[76:79], v[192:195], v[236:239], a[76:79]\n\t"                                     \
        "v_mfma_f32_16x16x32_bf16 a[28:31], v[180:183], v[220:223], a[28:31]\n\t"                                     \
        "v_mfma_f32_16x16x32_bf16 a[92:95], v[196:199], v[236:239], a[92:95]\n\t"                                     \
        "v_mfma_f32_16x16x32_bf16 a[44:47], v[184:187], v[220:223], a[44:47]\n\t"                                     \
        "v_mfma_f32_16x16x32_bf16 a[108:111], v[200:203], v[236:239], a[108:111]\n\t"                                 \
        "v_mfma_f32_16x16x32_bf16 a[60:63], v[188:191], v[220:223], a[60:63]\n\t"                                     \
        "v_mfma_f32_16x16x32_bf16 a[124:127], v[204:207], v[236:239], a[124:127]"                                     \
        :                                                                                                           \
        : "v"(RA0), "v"(RA1), "v"(LRD), "v"(TP0), "v"(TP1), "v"(TP2), "v"(TP3), "s"(SCL), "v"(NRA0), "v"(NRA1),   \
          "v"(NLRD)                                                                                \
        : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "memory")
// generated by tools/gen/gen_dkdv4_body.py (register map and operand list there); 226 instructions
#define RPO_D4_SLICE_BODY_HOT(RA0, RA1, LRD, TP0, TP1, TP2, TP3, SCL, NRA0, NRA1, NLRD)                    \
    asm volatile(                                                                                               \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                    \
        "v_mfma_f32_16x16x32_bf16 v[64:67], v[128:131], a[128:131], v[160:163]\n\t"                                   \
        "ds_read_b64_tr_b16 v[176:177], %3 offset:4096\n\t"                                                           \
        "v_mfma_f32_16x16x32_bf16 v[80:83], v[136:139], a[128:131], v[164:167]\n\t"                                   \
        "ds_read_b64_tr_b16 v[178:179], %3 offset:6144\n\t"                                                           \
        "v_mfma_f32_16x16x32_bf16 v[96:99], v[132:135], a[160:163], v[168:171]\n\t"                                   \
        "ds_read_b64_tr_b16 v[192:193], %3\n\t"                                                                       \
        "v_mfma_f32_16x16x32_bf16 v[112:115], v[140:143], a[160:163], v[172:175]\n\t"                                 \
        "ds_read_b64_tr_b16 v[194:195], %3 offset:2048\n\t"                                                           \
        "v_mfma_f32_16x16x32_bf16 v[64:67], v[144:147], a[132:135], v[64:67]\n\t"                                     \
        "ds_read_b64_tr_b16 v[180:181], %4 offset:4096\n\t"                                                           \
        "v_mfma_f32_16x16x32_bf16 v[80:83], v[152:155], a[132:135], v[80:83]\n\t"                                     \
        "ds_read_b64_tr_b16 v[182:183], %4 offset:6144\n\t"                                                           \
        "v_mfma_f32_16x16x32_bf16 v[96:99], v[148:151], a[164:167], v[96:99]\n\t"                                     \
        "ds_read_b64_tr_b16 v[196:197], %4\n\t"                                                                       \
        "v_mfma_f32_16x16x32_bf16 v[112:115], v[156:159], a[164:167], v[112:115]\n\t"                                 \
        "ds_read_b64_tr_b16 v[198:199], %4 offset:2048\n\t"                                                           \
        "v_mfma_f32_16x16x32_bf16 v[68:71], v[128:131], a[136:139], v[160:163]\n\t"                                   \
        "ds_read_b64_tr_b16 v[184:185], %5 offset:4096\n\t"                                                           \
        "v_mfma_f32_16x16x32_bf16 v[84:87], v[136:139], a[136:139], v[164:167]\n\t"                                   \
        "ds_read_b64_tr_b16 v[186:187], %5 offset:6144\n\t"                                                           \
        "v_mfma_f32_16x16x32_bf16 v[100:103], v[132:135], a[168:171], v[168:171]\n\t"                                 \
        "ds_read_b64_tr_b16 v[200:201], %5\n\t"                                                                       \
        "v_mul_f32 v64, %7, v64\n\t"                                                                                  \
        "v_mul_f32 v65, %7, v65\n\t"                                                                                  \
        "v_mul_f32 v66, %7, v66\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 v[116:119], v[140:143], a[168:171], v[172:175]\n\t"                                 \
        "ds_read_b64_tr_b16 v[202:203], %5 offset:2048\n\t"                                                           \
        "v_mul_f32 v67, %7, v67\n\t"                                                                                  \
        "v_mul_f32 v80, %7, v80\n\t"                                                                                  \
        "v_mul_f32 v81, %7, v81\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 v[68:71], v[144:147], a[140:143], v[68:71]\n\t"                                     \
        "ds_read_b64_tr_b16 v[188:189], %6 offset:4096\n\t"                                                           \
        "v_mul_f32 v82, %7, v82\n\t"                                                                                  \
        "v_mul_f32 v83, %7, v83\n\t"                                                                                  \
        "v_exp_f32 v64, v64\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[84:87], v[152:155], a[140:143], v[84:87]\n\t"                                     \
        "ds_read_b64_tr_b16 v[190:191], %6 offset:6144\n\t"                                                           \
        "v_exp_f32 v65, v65\n\t"                                                                                      \
        "v_exp_f32 v66, v66\n\t"                                                                                      \
        "v_exp_f32 v67, v67\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[100:103], v[148:151], a[172:175], v[100:103]\n\t"                                 \
        "ds_read_b64_tr_b16 v[204:205], %6\n\t"                                                                       \
        "v_exp_f32 v80, v80\n\t"                                                                                      \
        "v_exp_f32 v81, v81\n\t"                                                                                      \
        "v_exp_f32 v82, v82\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[116:119], v[156:159], a[172:175], v[116:119]\n\t"                                 \
        "ds_read_b64_tr_b16 v[206:207], %6 offset:2048\n\t"                                                           \
        "v_exp_f32 v83, v83\n\t"                                                                                      \
        "v_mul_f32 v96, v64, v96\n\t"                                                                                 \
        "v_mul_f32 v97, v65, v97\n\t"                                                                                 \
        "v_mfma_f32_16x16x32_bf16 v[72:75], v[128:131], a[144:147], v[160:163]\n\t"                                   \
        "v_mul_f32 v98, v66, v98\n\t"                                                                                 \
        "v_mul_f32 v99, v67, v99\n\t"                                                                                 \
        "v_mul_f32 v112, v80, v112\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 v[88:91], v[136:139], a[144:147], v[164:167]\n\t"                                   \
        "v_mul_f32 v113, v81, v113\n\t"                                                                               \
        "v_mul_f32 v114, v82, v114\n\t"                                                                               \
        "v_mul_f32 v115, v83, v115\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 v[104:107], v[132:135], a[176:179], v[168:171]\n\t"                                 \
        "v_cvt_pk_bf16_f32 v208, v64, v65\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v209, v66, v67\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v210, v80, v81\n\t"                                                                        \
        "v_mfma_f32_16x16x32_bf16 v[120:123], v[140:143], a[176:179], v[172:175]\n\t"                                 \
        "v_cvt_pk_bf16_f32 v211, v82, v83\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v224, v96, v97\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v225, v98, v99\n\t"                                                                        \
        "v_mfma_f32_16x16x32_bf16 v[72:75], v[144:147], a[148:151], v[72:75]\n\t"                                     \
        "v_cvt_pk_bf16_f32 v226, v112, v113\n\t"                                                                      \
        "v_cvt_pk_bf16_f32 v227, v114, v115\n\t"                                                                      \
        "v_mul_f32 v68, %7, v68\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 v[88:91], v[152:155], a[148:151], v[88:91]\n\t"                                     \
        "v_mul_f32 v69, %7, v69\n\t"                                                                                  \
        "v_mul_f32 v70, %7, v70\n\t"                                                                                  \
        "v_mul_f32 v71, %7, v71\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 v[104:107], v[148:151], a[180:183], v[104:107]\n\t"                                 \
        "v_mul_f32 v84, %7, v84\n\t"                                                                                  \
        "v_mul_f32 v85, %7, v85\n\t"                                                                                  \
        "v_mul_f32 v86, %7, v86\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 v[120:123], v[156:159], a[180:183], v[120:123]\n\t"                                 \
        "v_mul_f32 v87, %7, v87\n\t"                                                                                  \
        "v_exp_f32 v68, v68\n\t"                                                                                      \
        "v_exp_f32 v69, v69\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[76:79], v[128:131], a[152:155], v[160:163]\n\t"                                   \
        "v_exp_f32 v70, v70\n\t"                                                                                      \
        "v_exp_f32 v71, v71\n\t"                                                                                      \
        "v_exp_f32 v84, v84\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[92:95], v[136:139], a[152:155], v[164:167]\n\t"                                   \
        "v_exp_f32 v85, v85\n\t"                                                                                      \
        "v_exp_f32 v86, v86\n\t"                                                                                      \
        "v_exp_f32 v87, v87\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[108:111], v[132:135], a[184:187], v[168:171]\n\t"                                 \
        "v_mul_f32 v100, v68, v100\n\t"                                                                               \
        "v_mul_f32 v101, v69, v101\n\t"                                                                               \
        "v_mul_f32 v102, v70, v102\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 v[124:127], v[140:143], a[184:187], v[172:175]\n\t"                                 \
        "v_mul_f32 v103, v71, v103\n\t"                                                                               \
        "v_mul_f32 v116, v84, v116\n\t"                                                                               \
        "v_mul_f32 v117, v85, v117\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 v[76:79], v[144:147], a[156:159], v[76:79]\n\t"                                     \
        "v_mul_f32 v118, v86, v118\n\t"                                                                               \
        "v_mul_f32 v119, v87, v119\n\t"                                                                               \
        "v_cvt_pk_bf16_f32 v212, v68, v69\n\t"                                                                        \
        "v_mfma_f32_16x16x32_bf16 v[92:95], v[152:155], a[156:159], v[92:95]\n\t"                                     \
        "v_cvt_pk_bf16_f32 v213, v70, v71\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v214, v84, v85\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v215, v86, v87\n\t"                                                                        \
        "v_mfma_f32_16x16x32_bf16 v[108:111], v[148:151], a[188:191], v[108:111]\n\t"                                 \
        "v_cvt_pk_bf16_f32 v228, v100, v101\n\t"                                                                      \
        "v_cvt_pk_bf16_f32 v229, v102, v103\n\t"                                                                      \
        "v_cvt_pk_bf16_f32 v230, v116, v117\n\t"                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[124:127], v[156:159], a[188:191], v[124:127]\n\t"                                 \
        "v_cvt_pk_bf16_f32 v231, v118, v119\n\t"                                                                      \
        "v_mul_f32 v72, %7, v72\n\t"                                                                                  \
        "v_mul_f32 v73, %7, v73\n\t"                                                                                  \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                    \
        "s_nop 1\n\t"                                                                                                 \
        "v_mfma_f32_16x16x32_bf16 a[0:3], v[176:179], v[208:211], a[0:3]\n\t"                                         \
        "ds_read_b128 v[160:163], %10\n\t"                                                                            \
        "v_mul_f32 v74, %7, v74\n\t"                                                                                  \
        "v_mul_f32 v75, %7, v75\n\t"                                                                                  \
        "v_mul_f32 v88, %7, v88\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 a[64:67], v[192:195], v[224:227], a[64:67]\n\t"                                     \
        "ds_read_b128 v[168:171], %10 offset:128\n\t"                                                                 \
        "v_mul_f32 v89, %7, v89\n\t"                                                                                  \
        "v_mul_f32 v90, %7, v90\n\t"                                                                                  \
        "v_mul_f32 v91, %7, v91\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 a[16:19], v[180:183], v[208:211], a[16:19]\n\t"                                     \
        "ds_read_b128 v[128:131], %8\n\t"                                                                             \
        "v_exp_f32 v72, v72\n\t"                                                                                      \
        "v_exp_f32 v73, v73\n\t"                                                                                      \
        "v_exp_f32 v74, v74\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[80:83], v[196:199], v[224:227], a[80:83]\n\t"                                     \
        "ds_read_b128 v[132:135], %8 offset:4096\n\t"                                                                 \
        "v_exp_f32 v75, v75\n\t"                                                                                      \
        "v_exp_f32 v88, v88\n\t"                                                                                      \
        "v_exp_f32 v89, v89\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[32:35], v[184:187], v[208:211], a[32:35]\n\t"                                     \
        "ds_read_b128 v[164:167], %10 offset:64\n\t"                                                                  \
        "v_exp_f32 v90, v90\n\t"                                                                                      \
        "v_exp_f32 v91, v91\n\t"                                                                                      \
        "v_mul_f32 v104, v72, v104\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 a[96:99], v[200:203], v[224:227], a[96:99]\n\t"                                     \
        "ds_read_b128 v[172:175], %10 offset:192\n\t"                                                                 \
        "v_mul_f32 v105, v73, v105\n\t"                                                                               \
        "v_mul_f32 v106, v74, v106\n\t"                                                                               \
        "v_mul_f32 v107, v75, v107\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 a[48:51], v[188:191], v[208:211], a[48:51]\n\t"                                     \
        "ds_read_b128 v[136:139], %8 offset:2048\n\t"                                                                 \
        "v_mul_f32 v120, v88, v120\n\t"                                                                               \
        "v_mul_f32 v121, v89, v121\n\t"                                                                               \
        "v_mul_f32 v122, v90, v122\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 a[112:115], v[204:207], v[224:227], a[112:115]\n\t"                                 \
        "ds_read_b128 v[140:143], %8 offset:6144\n\t"                                                                 \
        "v_mul_f32 v123, v91, v123\n\t"                                                                               \
        "v_cvt_pk_bf16_f32 v216, v72, v73\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v217, v74, v75\n\t"                                                                        \
        "s_nop 1\n\t"                                                                                                 \
        "v_mfma_f32_16x16x32_bf16 a[4:7], v[176:179], v[212:215], a[4:7]\n\t"                                         \
        "ds_read_b128 v[144:147], %9\n\t"                                                                             \
        "v_cvt_pk_bf16_f32 v218, v88, v89\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v219, v90, v91\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v232, v104, v105\n\t"                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[68:71], v[192:195], v[228:231], a[68:71]\n\t"                                     \
        "ds_read_b128 v[148:151], %9 offset:4096\n\t"                                                                 \
        "v_cvt_pk_bf16_f32 v233, v106, v107\n\t"                                                                      \
        "v_cvt_pk_bf16_f32 v234, v120, v121\n\t"                                                                      \
        "v_cvt_pk_bf16_f32 v235, v122, v123\n\t"                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[20:23], v[180:183], v[212:215], a[20:23]\n\t"                                     \
        "ds_read_b128 v[152:155], %9 offset:2048\n\t"                                                                 \
        "v_mul_f32 v76, %7, v76\n\t"                                                                                  \
        "v_mul_f32 v77, %7, v77\n\t"                                                                                  \
        "v_mul_f32 v78, %7, v78\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 a[84:87], v[196:199], v[228:231], a[84:87]\n\t"                                     \
        "ds_read_b128 v[156:159], %9 offset:6144\n\t"                                                                 \
        "v_mul_f32 v79, %7, v79\n\t"                                                                                  \
        "v_mul_f32 v92, %7, v92\n\t"                                                                                  \
        "v_mul_f32 v93, %7, v93\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 a[36:39], v[184:187], v[212:215], a[36:39]\n\t"                                     \
        "v_mul_f32 v94, %7, v94\n\t"                                                                                  \
        "v_mul_f32 v95, %7, v95\n\t"                                                                                  \
        "v_exp_f32 v76, v76\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[100:103], v[200:203], v[228:231], a[100:103]\n\t"                                 \
        "v_exp_f32 v77, v77\n\t"                                                                                      \
        "v_exp_f32 v78, v78\n\t"                                                                                      \
        "v_exp_f32 v79, v79\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[52:55], v[188:191], v[212:215], a[52:55]\n\t"                                     \
        "v_exp_f32 v92, v92\n\t"                                                                                      \
        "v_exp_f32 v93, v93\n\t"                                                                                      \
        "v_exp_f32 v94, v94\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[116:119], v[204:207], v[228:231], a[116:119]\n\t"                                 \
        "v_exp_f32 v95, v95\n\t"                                                                                      \
        "v_mul_f32 v108, v76, v108\n\t"                                                                               \
        "v_mul_f32 v109, v77, v109\n\t"                                                                               \
        "s_nop 1\n\t"                                                                                                 \
        "v_mfma_f32_16x16x32_bf16 a[8:11], v[176:179], v[216:219], a[8:11]\n\t"                                       \
        "v_mul_f32 v110, v78, v110\n\t"                                                                               \
        "v_mul_f32 v111, v79, v111\n\t"                                                                               \
        "v_mul_f32 v124, v92, v124\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 a[72:75], v[192:195], v[232:235], a[72:75]\n\t"                                     \
        "v_mul_f32 v125, v93, v125\n\t"                                                                               \
        "v_mul_f32 v126, v94, v126\n\t"                                                                               \
        "v_mul_f32 v127, v95, v127\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 a[24:27], v[180:183], v[216:219], a[24:27]\n\t"                                     \
        "v_cvt_pk_bf16_f32 v220, v76, v77\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v221, v78, v79\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v222, v92, v93\n\t"                                                                        \
        "v_mfma_f32_16x16x32_bf16 a[88:91], v[196:199], v[232:235], a[88:91]\n\t"                                     \
        "v_cvt_pk_bf16_f32 v223, v94, v95\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v236, v108, v109\n\t"                                                                      \
        "v_cvt_pk_bf16_f32 v237, v110, v111\n\t"                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[40:43], v[184:187], v[216:219], a[40:43]\n\t"                                     \
        "v_cvt_pk_bf16_f32 v238, v124, v125\n\t"                                                                      \
        "v_cvt_pk_bf16_f32 v239, v126, v127\n\t"                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[104:107], v[200:203], v[232:235], a[104:107]\n\t"                                 \
        "v_mfma_f32_16x16x32_bf16 a[56:59], v[188:191], v[216:219], a[56:59]\n\t"                                     \
        "v_mfma_f32_16x16x32_bf16 a[120:123], v[204:207], v[232:235], a[120:123]\n\t"                                 \
        "s_nop 1\n\t"                                                                                                 \
        "v_mfma_f32_16x16x32_bf16 a[12:15], v[176:179], v[220:223], a[12:15]\n\t"                                     \
        "v_mfma_f32_16x16x32_bf16 a[76:79], v[192:195], v[236:239], a[76:79]\n\t"                                     \
        "v_mfma_f32_16x16x32_bf16 a[28:31], v[180:183], v[220:223], a[28:31]\n\t"                                     \
        "v_mfma_f32_16x16x32_bf16 a[92:95], v[196:199], v[236:239], a[92:95]\n\t"                                     \
        "v_mfma_f32_16x16x32_bf16 a[44:47], v[184:187], v[220:223], a[44:47]\n\t"                                     \
        "v_mfma_f32_16x16x32_bf16 a[108:111], v[200:203], v[236:239], a[108:111]\n\t"                                 \
        "v_mfma_f32_16x16x32_bf16 a[60:63], v[188:191], v[220:223], a[60:63]\n\t"                                     \
        "v_mfma_f32_16x16x32_bf16 a[124:127], v[204:207], v[236:239], a[124:127]"                                     \
        :                                                                                                           \
        : "v"(RA0), "v"(RA1), "v"(LRD), "v"(TP0), "v"(TP1), "v"(TP2), "v"(TP3), "s"(SCL), "v"(NRA0), "v"(NRA1),   \
          "v"(NLRD)                                                                                \
        : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "memory")
// generated by tools/gen/gen_dkdv4_body.py (register map and operand list there); 304 instructions
#define RPO_D4_DIAG_BODY_LOAD(RA0, RA1, LRD, TP0, TP1, TP2, TP3, SCL, NRA0, NRA1, NLRD, DLANE)                    \
    asm volatile(                                                                                               \
        "ds_read_b128 v[160:163], %2\n\t"                                                                             \
        "ds_read_b128 v[168:171], %2 offset:128\n\t"                                                                  \
        "ds_read_b128 v[128:131], %0\n\t"                                                                             \
        "ds_read_b128 v[132:135], %0 offset:4096\n\t"                                                                 \
        "ds_read_b128 v[164:167], %2 offset:64\n\t"                                                                   \
        "ds_read_b128 v[172:175], %2 offset:192\n\t"                                                                  \
        "ds_read_b128 v[136:139], %0 offset:2048\n\t"                                                                 \
        "ds_read_b128 v[140:143], %0 offset:6144\n\t"                                                                 \
        "ds_read_b128 v[144:147], %1\n\t"                                                                             \
        "ds_read_b128 v[148:151], %1 offset:4096\n\t"                                                                 \
        "ds_read_b128 v[152:155], %1 offset:2048\n\t"                                                                 \
        "ds_read_b128 v[156:159], %1 offset:6144\n\t"                                                                 \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                    \
        "v_mov_b32 v239, 0xf149f2ca\n\t"                                                                              \
        "v_cmp_ge_i32 vcc, 0, %11\n\t"                                                                                \
        "v_cndmask_b32 v64, v239, v160, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, 1, %11\n\t"                                                                                \
        "v_cndmask_b32 v65, v239, v161, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, 2, %11\n\t"                                                                                \
        "v_cndmask_b32 v66, v239, v162, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, 3, %11\n\t"                                                                                \
        "v_cndmask_b32 v67, v239, v163, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, 16, %11\n\t"                                                                               \
        "v_cndmask_b32 v80, v239, v164, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, 17, %11\n\t"                                                                               \
        "v_cndmask_b32 v81, v239, v165, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, 18, %11\n\t"                                                                               \
        "v_cndmask_b32 v82, v239, v166, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, 19, %11\n\t"                                                                               \
        "v_cndmask_b32 v83, v239, v167, vcc\n\t"                                                                      \
        "s_nop 1\n\t"                                                                                                 \
        "v_mfma_f32_16x16x32_bf16 v[64:67], v[128:131], a[128:131], v[64:67]\n\t"                                     \
        "ds_read_b64_tr_b16 v[176:177], %3 offset:4096\n\t"                                                           \
        "v_cmp_ge_i32 vcc, -16, %11\n\t"                                                                              \
        "v_cndmask_b32 v68, v239, v160, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, -15, %11\n\t"                                                                              \
        "v_cndmask_b32 v69, v239, v161, vcc\n\t"                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[80:83], v[136:139], a[128:131], v[80:83]\n\t"                                     \
        "ds_read_b64_tr_b16 v[178:179], %3 offset:6144\n\t"                                                           \
        "v_cmp_ge_i32 vcc, -14, %11\n\t"                                                                              \
        "v_cndmask_b32 v70, v239, v162, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, -13, %11\n\t"                                                                              \
        "v_cndmask_b32 v71, v239, v163, vcc\n\t"                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[96:99], v[132:135], a[160:163], v[168:171]\n\t"                                   \
        "ds_read_b64_tr_b16 v[192:193], %3\n\t"                                                                       \
        "v_cmp_ge_i32 vcc, 0, %11\n\t"                                                                                \
        "v_cndmask_b32 v84, v239, v164, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, 1, %11\n\t"                                                                                \
        "v_cndmask_b32 v85, v239, v165, vcc\n\t"                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[112:115], v[140:143], a[160:163], v[172:175]\n\t"                                 \
        "ds_read_b64_tr_b16 v[194:195], %3 offset:2048\n\t"                                                           \
        "v_cmp_ge_i32 vcc, 2, %11\n\t"                                                                                \
        "v_cndmask_b32 v86, v239, v166, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, 3, %11\n\t"                                                                                \
        "v_cndmask_b32 v87, v239, v167, vcc\n\t"                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[64:67], v[144:147], a[132:135], v[64:67]\n\t"                                     \
        "ds_read_b64_tr_b16 v[180:181], %4 offset:4096\n\t"                                                           \
        "v_mfma_f32_16x16x32_bf16 v[80:83], v[152:155], a[132:135], v[80:83]\n\t"                                     \
        "ds_read_b64_tr_b16 v[182:183], %4 offset:6144\n\t"                                                           \
        "v_mfma_f32_16x16x32_bf16 v[96:99], v[148:151], a[164:167], v[96:99]\n\t"                                     \
        "ds_read_b64_tr_b16 v[196:197], %4\n\t"                                                                       \
        "v_mfma_f32_16x16x32_bf16 v[112:115], v[156:159], a[164:167], v[112:115]\n\t"                                 \
        "ds_read_b64_tr_b16 v[198:199], %4 offset:2048\n\t"                                                           \
        "v_mfma_f32_16x16x32_bf16 v[68:71], v[128:131], a[136:139], v[68:71]\n\t"                                     \
        "ds_read_b64_tr_b16 v[184:185], %5 offset:4096\n\t"                                                           \
        "v_cmp_ge_i32 vcc, -32, %11\n\t"                                                                              \
        "v_cndmask_b32 v72, v239, v160, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, -31, %11\n\t"                                                                              \
        "v_cndmask_b32 v73, v239, v161, vcc\n\t"                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[84:87], v[136:139], a[136:139], v[84:87]\n\t"                                     \
        "ds_read_b64_tr_b16 v[186:187], %5 offset:6144\n\t"                                                           \
        "v_cmp_ge_i32 vcc, -30, %11\n\t"                                                                              \
        "v_cndmask_b32 v74, v239, v162, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, -29, %11\n\t"                                                                              \
        "v_cndmask_b32 v75, v239, v163, vcc\n\t"                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[100:103], v[132:135], a[168:171], v[168:171]\n\t"                                 \
        "ds_read_b64_tr_b16 v[200:201], %5\n\t"                                                                       \
        "v_cmp_ge_i32 vcc, -16, %11\n\t"                                                                              \
        "v_cndmask_b32 v88, v239, v164, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, -15, %11\n\t"                                                                              \
        "v_cndmask_b32 v89, v239, v165, vcc\n\t"                                                                      \
        "v_mul_f32 v64, %7, v64\n\t"                                                                                  \
        "v_mul_f32 v65, %7, v65\n\t"                                                                                  \
        "v_mul_f32 v66, %7, v66\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 v[116:119], v[140:143], a[168:171], v[172:175]\n\t"                                 \
        "ds_read_b64_tr_b16 v[202:203], %5 offset:2048\n\t"                                                           \
        "v_cmp_ge_i32 vcc, -14, %11\n\t"                                                                              \
        "v_cndmask_b32 v90, v239, v166, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, -13, %11\n\t"                                                                              \
        "v_cndmask_b32 v91, v239, v167, vcc\n\t"                                                                      \
        "v_mul_f32 v67, %7, v67\n\t"                                                                                  \
        "v_mul_f32 v80, %7, v80\n\t"                                                                                  \
        "v_mul_f32 v81, %7, v81\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 v[68:71], v[144:147], a[140:143], v[68:71]\n\t"                                     \
        "ds_read_b64_tr_b16 v[188:189], %6 offset:4096\n\t"                                                           \
        "v_mul_f32 v82, %7, v82\n\t"                                                                                  \
        "v_mul_f32 v83, %7, v83\n\t"                                                                                  \
        "v_exp_f32 v64, v64\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[84:87], v[152:155], a[140:143], v[84:87]\n\t"                                     \
        "ds_read_b64_tr_b16 v[190:191], %6 offset:6144\n\t"                                                           \
        "v_exp_f32 v65, v65\n\t"                                                                                      \
        "v_exp_f32 v66, v66\n\t"                                                                                      \
        "v_exp_f32 v67, v67\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[100:103], v[148:151], a[172:175], v[100:103]\n\t"                                 \
        "ds_read_b64_tr_b16 v[204:205], %6\n\t"                                                                       \
        "v_exp_f32 v80, v80\n\t"                                                                                      \
        "v_exp_f32 v81, v81\n\t"                                                                                      \
        "v_exp_f32 v82, v82\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[116:119], v[156:159], a[172:175], v[116:119]\n\t"                                 \
        "ds_read_b64_tr_b16 v[206:207], %6 offset:2048\n\t"                                                           \
        "v_exp_f32 v83, v83\n\t"                                                                                      \
        "v_mul_f32 v96, v64, v96\n\t"                                                                                 \
        "v_mul_f32 v97, v65, v97\n\t"                                                                                 \
        "v_mfma_f32_16x16x32_bf16 v[72:75], v[128:131], a[144:147], v[72:75]\n\t"                                     \
        "v_cmp_ge_i32 vcc, -48, %11\n\t"                                                                              \
        "v_cndmask_b32 v76, v239, v160, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, -47, %11\n\t"                                                                              \
        "v_cndmask_b32 v77, v239, v161, vcc\n\t"                                                                      \
        "v_mul_f32 v98, v66, v98\n\t"                                                                                 \
        "v_mul_f32 v99, v67, v99\n\t"                                                                                 \
        "v_mul_f32 v112, v80, v112\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 v[88:91], v[136:139], a[144:147], v[88:91]\n\t"                                     \
        "v_cmp_ge_i32 vcc, -46, %11\n\t"                                                                              \
        "v_cndmask_b32 v78, v239, v162, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, -45, %11\n\t"                                                                              \
        "v_cndmask_b32 v79, v239, v163, vcc\n\t"                                                                      \
        "v_mul_f32 v113, v81, v113\n\t"                                                                               \
        "v_mul_f32 v114, v82, v114\n\t"                                                                               \
        "v_mul_f32 v115, v83, v115\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 v[104:107], v[132:135], a[176:179], v[168:171]\n\t"                                 \
        "v_cmp_ge_i32 vcc, -32, %11\n\t"                                                                              \
        "v_cndmask_b32 v92, v239, v164, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, -31, %11\n\t"                                                                              \
        "v_cndmask_b32 v93, v239, v165, vcc\n\t"                                                                      \
        "v_cvt_pk_bf16_f32 v208, v64, v65\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v209, v66, v67\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v210, v80, v81\n\t"                                                                        \
        "v_mfma_f32_16x16x32_bf16 v[120:123], v[140:143], a[176:179], v[172:175]\n\t"                                 \
        "v_cmp_ge_i32 vcc, -30, %11\n\t"                                                                              \
        "v_cndmask_b32 v94, v239, v166, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, -29, %11\n\t"                                                                              \
        "v_cndmask_b32 v95, v239, v167, vcc\n\t"                                                                      \
        "v_cvt_pk_bf16_f32 v211, v82, v83\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v224, v96, v97\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v225, v98, v99\n\t"                                                                        \
        "v_mfma_f32_16x16x32_bf16 v[72:75], v[144:147], a[148:151], v[72:75]\n\t"                                     \
        "v_cvt_pk_bf16_f32 v226, v112, v113\n\t"                                                                      \
        "v_cvt_pk_bf16_f32 v227, v114, v115\n\t"                                                                      \
        "v_mul_f32 v68, %7, v68\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 v[88:91], v[152:155], a[148:151], v[88:91]\n\t"                                     \
        "v_mul_f32 v69, %7, v69\n\t"                                                                                  \
        "v_mul_f32 v70, %7, v70\n\t"                                                                                  \
        "v_mul_f32 v71, %7, v71\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 v[104:107], v[148:151], a[180:183], v[104:107]\n\t"                                 \
        "v_mul_f32 v84, %7, v84\n\t"                                                                                  \
        "v_mul_f32 v85, %7, v85\n\t"                                                                                  \
        "v_mul_f32 v86, %7, v86\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 v[120:123], v[156:159], a[180:183], v[120:123]\n\t"                                 \
        "v_mul_f32 v87, %7, v87\n\t"                                                                                  \
        "v_exp_f32 v68, v68\n\t"                                                                                      \
        "v_exp_f32 v69, v69\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[76:79], v[128:131], a[152:155], v[76:79]\n\t"                                     \
        "v_exp_f32 v70, v70\n\t"                                                                                      \
        "v_exp_f32 v71, v71\n\t"                                                                                      \
        "v_exp_f32 v84, v84\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[92:95], v[136:139], a[152:155], v[92:95]\n\t"                                     \
        "v_exp_f32 v85, v85\n\t"                                                                                      \
        "v_exp_f32 v86, v86\n\t"                                                                                      \
        "v_exp_f32 v87, v87\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[108:111], v[132:135], a[184:187], v[168:171]\n\t"                                 \
        "v_mul_f32 v100, v68, v100\n\t"                                                                               \
        "v_mul_f32 v101, v69, v101\n\t"                                                                               \
        "v_mul_f32 v102, v70, v102\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 v[124:127], v[140:143], a[184:187], v[172:175]\n\t"                                 \
        "v_mul_f32 v103, v71, v103\n\t"                                                                               \
        "v_mul_f32 v116, v84, v116\n\t"                                                                               \
        "v_mul_f32 v117, v85, v117\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 v[76:79], v[144:147], a[156:159], v[76:79]\n\t"                                     \
        "v_mul_f32 v118, v86, v118\n\t"                                                                               \
        "v_mul_f32 v119, v87, v119\n\t"                                                                               \
        "v_cvt_pk_bf16_f32 v212, v68, v69\n\t"                                                                        \
        "v_mfma_f32_16x16x32_bf16 v[92:95], v[152:155], a[156:159], v[92:95]\n\t"                                     \
        "v_cvt_pk_bf16_f32 v213, v70, v71\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v214, v84, v85\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v215, v86, v87\n\t"                                                                        \
        "v_mfma_f32_16x16x32_bf16 v[108:111], v[148:151], a[188:191], v[108:111]\n\t"                                 \
        "v_cvt_pk_bf16_f32 v228, v100, v101\n\t"                                                                      \
        "v_cvt_pk_bf16_f32 v229, v102, v103\n\t"                                                                      \
        "v_cvt_pk_bf16_f32 v230, v116, v117\n\t"                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[124:127], v[156:159], a[188:191], v[124:127]\n\t"                                 \
        "v_cvt_pk_bf16_f32 v231, v118, v119\n\t"                                                                      \
        "v_mul_f32 v72, %7, v72\n\t"                                                                                  \
        "v_mul_f32 v73, %7, v73\n\t"                                                                                  \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                    \
        "s_nop 1\n\t"                                                                                                 \
        "v_mfma_f32_16x16x32_bf16 a[0:3], v[176:179], v[208:211], a[0:3]\n\t"                                         \
        "ds_read_b128 v[160:163], %10\n\t"                                                                            \
        "v_mul_f32 v74, %7, v74\n\t"                                                                                  \
        "v_mul_f32 v75, %7, v75\n\t"                                                                                  \
        "v_mul_f32 v88, %7, v88\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 a[64:67], v[192:195], v[224:227], a[64:67]\n\t"                                     \
        "ds_read_b128 v[168:171], %10 offset:128\n\t"                                                                 \
        "v_mul_f32 v89, %7, v89\n\t"                                                                                  \
        "v_mul_f32 v90, %7, v90\n\t"                                                                                  \
        "v_mul_f32 v91, %7, v91\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 a[16:19], v[180:183], v[208:211], a[16:19]\n\t"                                     \
        "ds_read_b128 v[128:131], %8\n\t"                                                                             \
        "v_exp_f32 v72, v72\n\t"                                                                                      \
        "v_exp_f32 v73, v73\n\t"                                                                                      \
        "v_exp_f32 v74, v74\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[80:83], v[196:199], v[224:227], a[80:83]\n\t"                                     \
        "ds_read_b128 v[132:135], %8 offset:4096\n\t"                                                                 \
        "v_exp_f32 v75, v75\n\t"                                                                                      \
        "v_exp_f32 v88, v88\n\t"                                                                                      \
        "v_exp_f32 v89, v89\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[32:35], v[184:187], v[208:211], a[32:35]\n\t"                                     \
        "ds_read_b128 v[164:167], %10 offset:64\n\t"                                                                  \
        "v_exp_f32 v90, v90\n\t"                                                                                      \
        "v_exp_f32 v91, v91\n\t"                                                                                      \
        "v_mul_f32 v104, v72, v104\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 a[96:99], v[200:203], v[224:227], a[96:99]\n\t"                                     \
        "ds_read_b128 v[172:175], %10 offset:192\n\t"                                                                 \
        "v_mul_f32 v105, v73, v105\n\t"                                                                               \
        "v_mul_f32 v106, v74, v106\n\t"                                                                               \
        "v_mul_f32 v107, v75, v107\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 a[48:51], v[188:191], v[208:211], a[48:51]\n\t"                                     \
        "ds_read_b128 v[136:139], %8 offset:2048\n\t"                                                                 \
        "v_mul_f32 v120, v88, v120\n\t"                                                                               \
        "v_mul_f32 v121, v89, v121\n\t"                                                                               \
        "v_mul_f32 v122, v90, v122\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 a[112:115], v[204:207], v[224:227], a[112:115]\n\t"                                 \
        "ds_read_b128 v[140:143], %8 offset:6144\n\t"                                                                 \
        "v_mul_f32 v123, v91, v123\n\t"                                                                               \
        "v_cvt_pk_bf16_f32 v216, v72, v73\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v217, v74, v75\n\t"                                                                        \
        "s_nop 1\n\t"                                                                                                 \
        "v_mfma_f32_16x16x32_bf16 a[4:7], v[176:179], v[212:215], a[4:7]\n\t"                                         \
        "ds_read_b128 v[144:147], %9\n\t"                                                                             \
        "v_cvt_pk_bf16_f32 v218, v88, v89\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v219, v90, v91\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v232, v104, v105\n\t"                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[68:71], v[192:195], v[228:231], a[68:71]\n\t"                                     \
        "ds_read_b128 v[148:151], %9 offset:4096\n\t"                                                                 \
        "v_cvt_pk_bf16_f32 v233, v106, v107\n\t"                                                                      \
        "v_cvt_pk_bf16_f32 v234, v120, v121\n\t"                                                                      \
        "v_cvt_pk_bf16_f32 v235, v122, v123\n\t"                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[20:23], v[180:183], v[212:215], a[20:23]\n\t"                                     \
        "ds_read_b128 v[152:155], %9 offset:2048\n\t"                                                                 \
        "v_mul_f32 v76, %7, v76\n\t"                                                                                  \
        "v_mul_f32 v77, %7, v77\n\t"                                                                                  \
        "v_mul_f32 v78, %7, v78\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 a[84:87], v[196:199], v[228:231], a[84:87]\n\t"                                     \
        "ds_read_b128 v[156:159], %9 offset:6144\n\t"                                                                 \
        "v_mul_f32 v79, %7, v79\n\t"                                                                                  \
        "v_mul_f32 v92, %7, v92\n\t"                                                                                  \
        "v_mul_f32 v93, %7, v93\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 a[36:39], v[184:187], v[212:215], a[36:39]\n\t"                                     \
        "v_mul_f32 v94, %7, v94\n\t"                                                                                  \
        "v_mul_f32 v95, %7, v95\n\t"                                                                                  \
        "v_exp_f32 v76, v76\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[100:103], v[200:203], v[228:231], a[100:103]\n\t"                                 \
        "v_exp_f32 v77, v77\n\t"                                                                                      \
        "v_exp_f32 v78, v78\n\t"                                                                                      \
        "v_exp_f32 v79, v79\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[52:55], v[188:191], v[212:215], a[52:55]\n\t"                                     \
        "v_exp_f32 v92, v92\n\t"                                                                                      \
        "v_exp_f32 v93, v93\n\t"                                                                                      \
        "v_exp_f32 v94, v94\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[116:119], v[204:207], v[228:231], a[116:119]\n\t"                                 \
        "v_exp_f32 v95, v95\n\t"                                                                                      \
        "v_mul_f32 v108, v76, v108\n\t"                                                                               \
        "v_mul_f32 v109, v77, v109\n\t"                                                                               \
        "s_nop 1\n\t"                                                                                                 \
        "v_mfma_f32_16x16x32_bf16 a[8:11], v[176:179], v[216:219], a[8:11]\n\t"                                       \
        "v_mul_f32 v110, v78, v110\n\t"                                                                               \
        "v_mul_f32 v111, v79, v111\n\t"                                                                               \
        "v_mul_f32 v124, v92, v124\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 a[72:75], v[192:195], v[232:235], a[72:75]\n\t"                                     \
        "v_mul_f32 v125, v93, v125\n\t"                                                                               \
        "v_mul_f32 v126, v94, v126\n\t"                                                                               \
        "v_mul_f32 v127, v95, v127\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 a[24:27], v[180:183], v[216:219], a[24:27]\n\t"                                     \
        "v_cvt_pk_bf16_f32 v220, v76, v77\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v221, v78, v79\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v222, v92, v93\n\t"                                                                        \
        "v_mfma_f32_16x16x32_bf16 a[88:91], v[196:199], v[232:235], a[88:91]\n\t"                                     \
        "v_cvt_pk_bf16_f32 v223, v94, v95\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v236, v108, v109\n\t"                                                                      \
        "v_cvt_pk_bf16_f32 v237, v110, v111\n\t"                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[40:43], v[184:187], v[216:219], a[40:43]\n\t"                                     \
        "v_cvt_pk_bf16_f32 v238, v124, v125\n\t"                                                                      \
        "v_cvt_pk_bf16_f32 v239, v126, v127\n\t"                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[104:107], v[200:203], v[232:235], a[104:107]\n\t"                                 \
        "v_mfma_f32_16x16x32_bf16 a[56:59], v[188:191], v[216:219], a[56:59]\n\t"                                     \
        "v_mfma_f32_16x16x32_bf16 a[120:123], v[204:207], v[232:235], a[120:123]\n\t"                                 \
        "s_nop 1\n\t"                                                                                                 \
        "v_mfma_f32_16x16x32_bf16 a[12:15], v[176:179], v[220:223], a[12:15]\n\t"                                     \
        "v_mfma_f32_16x16x32_bf16 a[76:79], v[192:195], v[236:239], a[76:79]\n\t"                                     \
        "v_mfma_f32_16x16x32_bf16 a[28:31], v[180:183], v[220:223], a[28:31]\n\t"                                     \
        "v_mfma_f32_16x16x32_bf16 a[92:95], v[196:199], v[236:239], a[92:95]\n\t"                                     \
        "v_mfma_f32_16x16x32_bf16 a[44:47], v[184:187], v[220:223], a[44:47]\n\t"                                     \
        "v_mfma_f32_16x16x32_bf16 a[108:111], v[200:203], v[236:239], a[108:111]\n\t"                                 \
        "v_mfma_f32_16x16x32_bf16 a[60:63], v[188:191], v[220:223], a[60:63]\n\t"                                     \
        "v_mfma_f32_16x16x32_bf16 a[124:127], v[204:207], v[236:239], a[124:127]"                                     \
        :                                                                                                           \
        : "v"(RA0), "v"(RA1), "v"(LRD), "v"(TP0), "v"(TP1), "v"(TP2), "v"(TP3), "s"(SCL), "v"(NRA0), "v"(NRA1),   \
          "v"(NLRD), "v"(DLANE)                                                                                \
        : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "vcc", "memory")
// generated by tools/gen/gen_dkdv4_body.py (register map and operand list there); 292 instructions
#define RPO_D4_DIAG_BODY_HOT(RA0, RA1, LRD, TP0, TP1, TP2, TP3, SCL, NRA0, NRA1, NLRD, DLANE)                    \
    asm volatile(                                                                                               \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                    \
        "v_mov_b32 v239, 0xf149f2ca\n\t"                                                                              \
        "v_cmp_ge_i32 vcc, 0, %11\n\t"                                                                                \
        "v_cndmask_b32 v64, v239, v160, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, 1, %11\n\t"                                                                                \
        "v_cndmask_b32 v65, v239, v161, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, 2, %11\n\t"                                                                                \
        "v_cndmask_b32 v66, v239, v162, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, 3, %11\n\t"                                                                                \
        "v_cndmask_b32 v67, v239, v163, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, 16, %11\n\t"                                                                               \
        "v_cndmask_b32 v80, v239, v164, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, 17, %11\n\t"                                                                               \
        "v_cndmask_b32 v81, v239, v165, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, 18, %11\n\t"                                                                               \
        "v_cndmask_b32 v82, v239, v166, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, 19, %11\n\t"                                                                               \
        "v_cndmask_b32 v83, v239, v167, vcc\n\t"                                                                      \
        "s_nop 1\n\t"                                                                                                 \
        "v_mfma_f32_16x16x32_bf16 v[64:67], v[128:131], a[128:131], v[64:67]\n\t"                                     \
        "ds_read_b64_tr_b16 v[176:177], %3 offset:4096\n\t"                                                           \
        "v_cmp_ge_i32 vcc, -16, %11\n\t"                                                                              \
        "v_cndmask_b32 v68, v239, v160, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, -15, %11\n\t"                                                                              \
        "v_cndmask_b32 v69, v239, v161, vcc\n\t"                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[80:83], v[136:139], a[128:131], v[80:83]\n\t"                                     \
        "ds_read_b64_tr_b16 v[178:179], %3 offset:6144\n\t"                                                           \
        "v_cmp_ge_i32 vcc, -14, %11\n\t"                                                                              \
        "v_cndmask_b32 v70, v239, v162, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, -13, %11\n\t"                                                                              \
        "v_cndmask_b32 v71, v239, v163, vcc\n\t"                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[96:99], v[132:135], a[160:163], v[168:171]\n\t"                                   \
        "ds_read_b64_tr_b16 v[192:193], %3\n\t"                                                                       \
        "v_cmp_ge_i32 vcc, 0, %11\n\t"                                                                                \
        "v_cndmask_b32 v84, v239, v164, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, 1, %11\n\t"                                                                                \
        "v_cndmask_b32 v85, v239, v165, vcc\n\t"                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[112:115], v[140:143], a[160:163], v[172:175]\n\t"                                 \
        "ds_read_b64_tr_b16 v[194:195], %3 offset:2048\n\t"                                                           \
        "v_cmp_ge_i32 vcc, 2, %11\n\t"                                                                                \
        "v_cndmask_b32 v86, v239, v166, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, 3, %11\n\t"                                                                                \
        "v_cndmask_b32 v87, v239, v167, vcc\n\t"                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[64:67], v[144:147], a[132:135], v[64:67]\n\t"                                     \
        "ds_read_b64_tr_b16 v[180:181], %4 offset:4096\n\t"                                                           \
        "v_mfma_f32_16x16x32_bf16 v[80:83], v[152:155], a[132:135], v[80:83]\n\t"                                     \
        "ds_read_b64_tr_b16 v[182:183], %4 offset:6144\n\t"                                                           \
        "v_mfma_f32_16x16x32_bf16 v[96:99], v[148:151], a[164:167], v[96:99]\n\t"                                     \
        "ds_read_b64_tr_b16 v[196:197], %4\n\t"                                                                       \
        "v_mfma_f32_16x16x32_bf16 v[112:115], v[156:159], a[164:167], v[112:115]\n\t"                                 \
        "ds_read_b64_tr_b16 v[198:199], %4 offset:2048\n\t"                                                           \
        "v_mfma_f32_16x16x32_bf16 v[68:71], v[128:131], a[136:139], v[68:71]\n\t"                                     \
        "ds_read_b64_tr_b16 v[184:185], %5 offset:4096\n\t"                                                           \
        "v_cmp_ge_i32 vcc, -32, %11\n\t"                                                                              \
        "v_cndmask_b32 v72, v239, v160, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, -31, %11\n\t"                                                                              \
        "v_cndmask_b32 v73, v239, v161, vcc\n\t"                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[84:87], v[136:139], a[136:139], v[84:87]\n\t"                                     \
        "ds_read_b64_tr_b16 v[186:187], %5 offset:6144\n\t"                                                           \
        "v_cmp_ge_i32 vcc, -30, %11\n\t"                                                                              \
        "v_cndmask_b32 v74, v239, v162, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, -29, %11\n\t"                                                                              \
        "v_cndmask_b32 v75, v239, v163, vcc\n\t"                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[100:103], v[132:135], a[168:171], v[168:171]\n\t"                                 \
        "ds_read_b64_tr_b16 v[200:201], %5\n\t"                                                                       \
        "v_cmp_ge_i32 vcc, -16, %11\n\t"                                                                              \
        "v_cndmask_b32 v88, v239, v164, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, -15, %11\n\t"                                                                              \
        "v_cndmask_b32 v89, v239, v165, vcc\n\t"                                                                      \
        "v_mul_f32 v64, %7, v64\n\t"                                                                                  \
        "v_mul_f32 v65, %7, v65\n\t"                                                                                  \
        "v_mul_f32 v66, %7, v66\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 v[116:119], v[140:143], a[168:171], v[172:175]\n\t"                                 \
        "ds_read_b64_tr_b16 v[202:203], %5 offset:2048\n\t"                                                           \
        "v_cmp_ge_i32 vcc, -14, %11\n\t"                                                                              \
        "v_cndmask_b32 v90, v239, v166, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, -13, %11\n\t"                                                                              \
        "v_cndmask_b32 v91, v239, v167, vcc\n\t"                                                                      \
        "v_mul_f32 v67, %7, v67\n\t"                                                                                  \
        "v_mul_f32 v80, %7, v80\n\t"                                                                                  \
        "v_mul_f32 v81, %7, v81\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 v[68:71], v[144:147], a[140:143], v[68:71]\n\t"                                     \
        "ds_read_b64_tr_b16 v[188:189], %6 offset:4096\n\t"                                                           \
        "v_mul_f32 v82, %7, v82\n\t"                                                                                  \
        "v_mul_f32 v83, %7, v83\n\t"                                                                                  \
        "v_exp_f32 v64, v64\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[84:87], v[152:155], a[140:143], v[84:87]\n\t"                                     \
        "ds_read_b64_tr_b16 v[190:191], %6 offset:6144\n\t"                                                           \
        "v_exp_f32 v65, v65\n\t"                                                                                      \
        "v_exp_f32 v66, v66\n\t"                                                                                      \
        "v_exp_f32 v67, v67\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[100:103], v[148:151], a[172:175], v[100:103]\n\t"                                 \
        "ds_read_b64_tr_b16 v[204:205], %6\n\t"                                                                       \
        "v_exp_f32 v80, v80\n\t"                                                                                      \
        "v_exp_f32 v81, v81\n\t"                                                                                      \
        "v_exp_f32 v82, v82\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[116:119], v[156:159], a[172:175], v[116:119]\n\t"                                 \
        "ds_read_b64_tr_b16 v[206:207], %6 offset:2048\n\t"                                                           \
        "v_exp_f32 v83, v83\n\t"                                                                                      \
        "v_mul_f32 v96, v64, v96\n\t"                                                                                 \
        "v_mul_f32 v97, v65, v97\n\t"                                                                                 \
        "v_mfma_f32_16x16x32_bf16 v[72:75], v[128:131], a[144:147], v[72:75]\n\t"                                     \
        "v_cmp_ge_i32 vcc, -48, %11\n\t"                                                                              \
        "v_cndmask_b32 v76, v239, v160, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, -47, %11\n\t"                                                                              \
        "v_cndmask_b32 v77, v239, v161, vcc\n\t"                                                                      \
        "v_mul_f32 v98, v66, v98\n\t"                                                                                 \
        "v_mul_f32 v99, v67, v99\n\t"                                                                                 \
        "v_mul_f32 v112, v80, v112\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 v[88:91], v[136:139], a[144:147], v[88:91]\n\t"                                     \
        "v_cmp_ge_i32 vcc, -46, %11\n\t"                                                                              \
        "v_cndmask_b32 v78, v239, v162, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, -45, %11\n\t"                                                                              \
        "v_cndmask_b32 v79, v239, v163, vcc\n\t"                                                                      \
        "v_mul_f32 v113, v81, v113\n\t"                                                                               \
        "v_mul_f32 v114, v82, v114\n\t"                                                                               \
        "v_mul_f32 v115, v83, v115\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 v[104:107], v[132:135], a[176:179], v[168:171]\n\t"                                 \
        "v_cmp_ge_i32 vcc, -32, %11\n\t"                                                                              \
        "v_cndmask_b32 v92, v239, v164, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, -31, %11\n\t"                                                                              \
        "v_cndmask_b32 v93, v239, v165, vcc\n\t"                                                                      \
        "v_cvt_pk_bf16_f32 v208, v64, v65\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v209, v66, v67\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v210, v80, v81\n\t"                                                                        \
        "v_mfma_f32_16x16x32_bf16 v[120:123], v[140:143], a[176:179], v[172:175]\n\t"                                 \
        "v_cmp_ge_i32 vcc, -30, %11\n\t"                                                                              \
        "v_cndmask_b32 v94, v239, v166, vcc\n\t"                                                                      \
        "v_cmp_ge_i32 vcc, -29, %11\n\t"                                                                              \
        "v_cndmask_b32 v95, v239, v167, vcc\n\t"                                                                      \
        "v_cvt_pk_bf16_f32 v211, v82, v83\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v224, v96, v97\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v225, v98, v99\n\t"                                                                        \
        "v_mfma_f32_16x16x32_bf16 v[72:75], v[144:147], a[148:151], v[72:75]\n\t"                                     \
        "v_cvt_pk_bf16_f32 v226, v112, v113\n\t"                                                                      \
        "v_cvt_pk_bf16_f32 v227, v114, v115\n\t"                                                                      \
        "v_mul_f32 v68, %7, v68\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 v[88:91], v[152:155], a[148:151], v[88:91]\n\t"                                     \
        "v_mul_f32 v69, %7, v69\n\t"                                                                                  \
        "v_mul_f32 v70, %7, v70\n\t"                                                                                  \
        "v_mul_f32 v71, %7, v71\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 v[104:107], v[148:151], a[180:183], v[104:107]\n\t"                                 \
        "v_mul_f32 v84, %7, v84\n\t"                                                                                  \
        "v_mul_f32 v85, %7, v85\n\t"                                                                                  \
        "v_mul_f32 v86, %7, v86\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 v[120:123], v[156:159], a[180:183], v[120:123]\n\t"                                 \
        "v_mul_f32 v87, %7, v87\n\t"                                                                                  \
        "v_exp_f32 v68, v68\n\t"                                                                                      \
        "v_exp_f32 v69, v69\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[76:79], v[128:131], a[152:155], v[76:79]\n\t"                                     \
        "v_exp_f32 v70, v70\n\t"                                                                                      \
        "v_exp_f32 v71, v71\n\t"                                                                                      \
        "v_exp_f32 v84, v84\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[92:95], v[136:139], a[152:155], v[92:95]\n\t"                                     \
        "v_exp_f32 v85, v85\n\t"                                                                                      \
        "v_exp_f32 v86, v86\n\t"                                                                                      \
        "v_exp_f32 v87, v87\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[108:111], v[132:135], a[184:187], v[168:171]\n\t"                                 \
        "v_mul_f32 v100, v68, v100\n\t"                                                                               \
        "v_mul_f32 v101, v69, v101\n\t"                                                                               \
        "v_mul_f32 v102, v70, v102\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 v[124:127], v[140:143], a[184:187], v[172:175]\n\t"                                 \
        "v_mul_f32 v103, v71, v103\n\t"                                                                               \
        "v_mul_f32 v116, v84, v116\n\t"                                                                               \
        "v_mul_f32 v117, v85, v117\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 v[76:79], v[144:147], a[156:159], v[76:79]\n\t"                                     \
        "v_mul_f32 v118, v86, v118\n\t"                                                                               \
        "v_mul_f32 v119, v87, v119\n\t"                                                                               \
        "v_cvt_pk_bf16_f32 v212, v68, v69\n\t"                                                                        \
        "v_mfma_f32_16x16x32_bf16 v[92:95], v[152:155], a[156:159], v[92:95]\n\t"                                     \
        "v_cvt_pk_bf16_f32 v213, v70, v71\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v214, v84, v85\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v215, v86, v87\n\t"                                                                        \
        "v_mfma_f32_16x16x32_bf16 v[108:111], v[148:151], a[188:191], v[108:111]\n\t"                                 \
        "v_cvt_pk_bf16_f32 v228, v100, v101\n\t"                                                                      \
        "v_cvt_pk_bf16_f32 v229, v102, v103\n\t"                                                                      \
        "v_cvt_pk_bf16_f32 v230, v116, v117\n\t"                                                                      \
        "v_mfma_f32_16x16x32_bf16 v[124:127], v[156:159], a[188:191], v[124:127]\n\t"                                 \
        "v_cvt_pk_bf16_f32 v231, v118, v119\n\t"                                                                      \
        "v_mul_f32 v72, %7, v72\n\t"                                                                                  \
        "v_mul_f32 v73, %7, v73\n\t"                                                                                  \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                    \
        "s_nop 1\n\t"                                                                                                 \
        "v_mfma_f32_16x16x32_bf16 a[0:3], v[176:179], v[208:211], a[0:3]\n\t"                                         \
        "ds_read_b128 v[160:163], %10\n\t"                                                                            \
        "v_mul_f32 v74, %7, v74\n\t"                                                                                  \
        "v_mul_f32 v75, %7, v75\n\t"                                                                                  \
        "v_mul_f32 v88, %7, v88\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 a[64:67], v[192:195], v[224:227], a[64:67]\n\t"                                     \
        "ds_read_b128 v[168:171], %10 offset:128\n\t"                                                                 \
        "v_mul_f32 v89, %7, v89\n\t"                                                                                  \
        "v_mul_f32 v90, %7, v90\n\t"                                                                                  \
        "v_mul_f32 v91, %7, v91\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 a[16:19], v[180:183], v[208:211], a[16:19]\n\t"                                     \
        "ds_read_b128 v[128:131], %8\n\t"                                                                             \
        "v_exp_f32 v72, v72\n\t"                                                                                      \
        "v_exp_f32 v73, v73\n\t"                                                                                      \
        "v_exp_f32 v74, v74\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[80:83], v[196:199], v[224:227], a[80:83]\n\t"                                     \
        "ds_read_b128 v[132:135], %8 offset:4096\n\t"                                                                 \
        "v_exp_f32 v75, v75\n\t"                                                                                      \
        "v_exp_f32 v88, v88\n\t"                                                                                      \
        "v_exp_f32 v89, v89\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[32:35], v[184:187], v[208:211], a[32:35]\n\t"                                     \
        "ds_read_b128 v[164:167], %10 offset:64\n\t"                                                                  \
        "v_exp_f32 v90, v90\n\t"                                                                                      \
        "v_exp_f32 v91, v91\n\t"                                                                                      \
        "v_mul_f32 v104, v72, v104\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 a[96:99], v[200:203], v[224:227], a[96:99]\n\t"                                     \
        "ds_read_b128 v[172:175], %10 offset:192\n\t"                                                                 \
        "v_mul_f32 v105, v73, v105\n\t"                                                                               \
        "v_mul_f32 v106, v74, v106\n\t"                                                                               \
        "v_mul_f32 v107, v75, v107\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 a[48:51], v[188:191], v[208:211], a[48:51]\n\t"                                     \
        "ds_read_b128 v[136:139], %8 offset:2048\n\t"                                                                 \
        "v_mul_f32 v120, v88, v120\n\t"                                                                               \
        "v_mul_f32 v121, v89, v121\n\t"                                                                               \
        "v_mul_f32 v122, v90, v122\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 a[112:115], v[204:207], v[224:227], a[112:115]\n\t"                                 \
        "ds_read_b128 v[140:143], %8 offset:6144\n\t"                                                                 \
        "v_mul_f32 v123, v91, v123\n\t"                                                                               \
        "v_cvt_pk_bf16_f32 v216, v72, v73\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v217, v74, v75\n\t"                                                                        \
        "s_nop 1\n\t"                                                                                                 \
        "v_mfma_f32_16x16x32_bf16 a[4:7], v[176:179], v[212:215], a[4:7]\n\t"                                         \
        "ds_read_b128 v[144:147], %9\n\t"                                                                             \
        "v_cvt_pk_bf16_f32 v218, v88, v89\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v219, v90, v91\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v232, v104, v105\n\t"                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[68:71], v[192:195], v[228:231], a[68:71]\n\t"                                     \
        "ds_read_b128 v[148:151], %9 offset:4096\n\t"                                                                 \
        "v_cvt_pk_bf16_f32 v233, v106, v107\n\t"                                                                      \
        "v_cvt_pk_bf16_f32 v234, v120, v121\n\t"                                                                      \
        "v_cvt_pk_bf16_f32 v235, v122, v123\n\t"                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[20:23], v[180:183], v[212:215], a[20:23]\n\t"                                     \
        "ds_read_b128 v[152:155], %9 offset:2048\n\t"                                                                 \
        "v_mul_f32 v76, %7, v76\n\t"                                                                                  \
        "v_mul_f32 v77, %7, v77\n\t"                                                                                  \
        "v_mul_f32 v78, %7, v78\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 a[84:87], v[196:199], v[228:231], a[84:87]\n\t"                                     \
        "ds_read_b128 v[156:159], %9 offset:6144\n\t"                                                                 \
        "v_mul_f32 v79, %7, v79\n\t"                                                                                  \
        "v_mul_f32 v92, %7, v92\n\t"                                                                                  \
        "v_mul_f32 v93, %7, v93\n\t"                                                                                  \
        "v_mfma_f32_16x16x32_bf16 a[36:39], v[184:187], v[212:215], a[36:39]\n\t"                                     \
        "v_mul_f32 v94, %7, v94\n\t"                                                                                  \
        "v_mul_f32 v95, %7, v95\n\t"                                                                                  \
        "v_exp_f32 v76, v76\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[100:103], v[200:203], v[228:231], a[100:103]\n\t"                                 \
        "v_exp_f32 v77, v77\n\t"                                                                                      \
        "v_exp_f32 v78, v78\n\t"                                                                                      \
        "v_exp_f32 v79, v79\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[52:55], v[188:191], v[212:215], a[52:55]\n\t"                                     \
        "v_exp_f32 v92, v92\n\t"                                                                                      \
        "v_exp_f32 v93, v93\n\t"                                                                                      \
        "v_exp_f32 v94, v94\n\t"                                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[116:119], v[204:207], v[228:231], a[116:119]\n\t"                                 \
        "v_exp_f32 v95, v95\n\t"                                                                                      \
        "v_mul_f32 v108, v76, v108\n\t"                                                                               \
        "v_mul_f32 v109, v77, v109\n\t"                                                                               \
        "s_nop 1\n\t"                                                                                                 \
        "v_mfma_f32_16x16x32_bf16 a[8:11], v[176:179], v[216:219], a[8:11]\n\t"                                       \
        "v_mul_f32 v110, v78, v110\n\t"                                                                               \
        "v_mul_f32 v111, v79, v111\n\t"                                                                               \
        "v_mul_f32 v124, v92, v124\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 a[72:75], v[192:195], v[232:235], a[72:75]\n\t"                                     \
        "v_mul_f32 v125, v93, v125\n\t"                                                                               \
        "v_mul_f32 v126, v94, v126\n\t"                                                                               \
        "v_mul_f32 v127, v95, v127\n\t"                                                                               \
        "v_mfma_f32_16x16x32_bf16 a[24:27], v[180:183], v[216:219], a[24:27]\n\t"                                     \
        "v_cvt_pk_bf16_f32 v220, v76, v77\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v221, v78, v79\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v222, v92, v93\n\t"                                                                        \
        "v_mfma_f32_16x16x32_bf16 a[88:91], v[196:199], v[232:235], a[88:91]\n\t"                                     \
        "v_cvt_pk_bf16_f32 v223, v94, v95\n\t"                                                                        \
        "v_cvt_pk_bf16_f32 v236, v108, v109\n\t"                                                                      \
        "v_cvt_pk_bf16_f32 v237, v110, v111\n\t"                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[40:43], v[184:187], v[216:219], a[40:43]\n\t"                                     \
        "v_cvt_pk_bf16_f32 v238, v124, v125\n\t"                                                                      \
        "v_cvt_pk_bf16_f32 v239, v126, v127\n\t"                                                                      \
        "v_mfma_f32_16x16x32_bf16 a[104:107], v[200:203], v[232:235], a[104:107]\n\t"                                 \
        "v_mfma_f32_16x16x32_bf16 a[56:59], v[188:191], v[216:219], a[56:59]\n\t"                                     \
        "v_mfma_f32_16x16x32_bf16 a[120:123], v[204:207], v[232:235], a[120:123]\n\t"                                 \
        "s_nop 1\n\t"                                                                                                 \
        "v_mfma_f32_16x16x32_bf16 a[12:15], v[176:179], v[220:223], a[12:15]\n\t"                                     \
        "v_mfma_f32_16x16x32_bf16 a[76:79], v[192:195], v[236:239], a[76:79]\n\t"                                     \
        "v_mfma_f32_16x16x32_bf16 a[28:31], v[180:183], v[220:223], a[28:31]\n\t"                                     \
        "v_mfma_f32_16x16x32_bf16 a[92:95], v[196:199], v[236:239], a[92:95]\n\t"                                     \
        "v_mfma_f32_16x16x32_bf16 a[44:47], v[184:187], v[220:223], a[44:47]\n\t"                                     \
        "v_mfma_f32_16x16x32_bf16 a[108:111], v[200:203], v[236:239], a[108:111]\n\t"                                 \
        "v_mfma_f32_16x16x32_bf16 a[60:63], v[188:191], v[220:223], a[60:63]\n\t"                                     \
        "v_mfma_f32_16x16x32_bf16 a[124:127], v[204:207], v[236:239], a[124:127]"                                     \
        :                                                                                                           \
        : "v"(RA0), "v"(RA1), "v"(LRD), "v"(TP0), "v"(TP1), "v"(TP2), "v"(TP3), "s"(SCL), "v"(NRA0), "v"(NRA1),   \
          "v"(NLRD), "v"(DLANE)                                                                                \
        : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "vcc", "memory")

template <bool DOWN>
__global__ __launch_bounds__(256, 1) void fa_bwd_dkdv4_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const bf16_t* __restrict__ dout, int64_t sq, int64_t sk, int64_t sv, int64_t sdo, const int* __restrict__ cu,
    const int* __restrict__ ktiles, int nh, int nkv, float scale_log2e, float scale, const float* __restrict__ nl,
    const float* __restrict__ nd, int64_t T, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, int64_t sdk,
    int64_t sdv, int n_ktiles, int gshift, const float* __restrict__ rcos, const float* __restrict__ rsin, int64_t rperiod) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    RPO_LAD_DECL;
    RPO_LAD(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, fr = lane & 15;
    const int per = (n_ktiles + 7) >> 3;
    const int entry = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (entry >= n_ktiles) return;
    const int seq = ktiles[3 * entry], hk = ktiles[3 * entry + 1], kb0 = ktiles[3 * entry + 2];
    const int k0 = kb0 + 64 * wave;                        // this wave's 64 keys
    const int group = nh / nkv;
    const int64_t t0 = cu[seq];
    const int len = cu[seq + 1] - (int)t0;
    if (kb0 >= len) return;                                // padding entry of the table
    const int qt0 = kb0;                                   // first query row that can see the block's keys (multiple of 32)
    const int nsl = (len - qt0 + kSl - 1) / kSl;
    const int niter = nsl * group;

    short8_t bk[4][2], bv[4][2];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int key = k0 + 16 * n + fr;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bk[n][ks] = key < len ? *reinterpret_cast<const short8_t*>(k + (t0 + key) * sk + hk * kFaHD + 32 * ks + 8 * g)
                                  : short8_t{0, 0, 0, 0, 0, 0, 0, 0};
            bv[n][ks] = key < len ? *reinterpret_cast<const short8_t*>(v + (t0 + key) * sv + hk * kFaHD + 32 * ks + 8 * g)
                                  : short8_t{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    // staging: wave w moves rows 8w .. 8w + 7 of the Q and of the dO slice (one 1-KiB DMA each; lane l carries row
    // 8w + (l >> 3), physical chunk l & 7 = logical chunk (l & 7) ^ (row & 7)); wave 0 also moves the 64 row constants
    // (lanes 0..31: -lse / scale, lanes 32..63: -delta).
    const int srow = 8 * wave + (lane >> 3), lchunk = (lane & 7) ^ (lane >> 3);
    const unsigned sqb = (unsigned)sq * 2u, sdob = (unsigned)sdo * 2u;
    const float* rc_src = lane < 32 ? nl : nd;
    auto stage = [&](int hq, int qb, int buf) {
        char* base = smem + buf * kSlImg;
        const char* qsrc = reinterpret_cast<const char*>(q + t0 * sq + hq * kFaHD);
        const char* dsrc = reinterpret_cast<const char*>(dout + t0 * sdo + hq * kFaHD);
        const unsigned row = (unsigned)min(qb + srow, len - 1);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qsrc + (row * sqb + lchunk * 16)),
                                         (__attribute__((address_space(3))) void*)(base + wave * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dsrc + (row * sdob + lchunk * 16)),
                                         (__attribute__((address_space(3))) void*)(base + kSl * 128 + wave * 1024), 16, 0, 0);
        if (wave == 0) {
            const float* src = rc_src + (int64_t)hq * T + t0 + (unsigned)min(qb + (lane & 31), len - 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(base + 2 * kSl * 128), 4, 0, 0);
        }
    };
    // (q head, slice) of the next stage to issue, advanced incrementally.  Two sweep orders:
    //   sweep_down = 0: head-major, slices ascending (round 1);
    //   sweep_down = 1: slice-major from the LAST query slice downwards, the group's q heads innermost.  Every key block of a
    //   (sequence, kv head) then starts at the same slice and walks the same (slice, head) sequence: launched side by side on
    //   one XCD (the group-ordered work list) they read each Q / dO slice at about the same time, once from HBM and the rest
    //   from L2 (round 1's order: 10 % L2 hits, 8.9 GB of HBM traffic per launch for 1.2 GB of operands).
    //   (the downward order needs a power-of-two group: stage / iteration n is slice nsl - 1 - (n >> gshift), head n & (group - 1),
    //   so it carries LESS loop state than the ascending one -- this kernel has no register to spare: hipcc's resource line
    //   must show 0 scratch, see below)
    int st_h = hk * group, st_s = 0, st_buf = 0, st_n = 0;
    auto stage_next = [&]() {
        if constexpr (DOWN) {
            stage(hk * group + (st_n & (group - 1)), qt0 + (nsl - 1 - (st_n >> gshift)) * kSl, st_buf);
        } else {
            stage(st_h, qt0 + st_s * kSl, st_buf);
            if (++st_s == nsl) { st_s = 0; ++st_h; }
        }
        ++st_n;
        st_buf = (st_buf + 1) & (kSlRing - 1);
    };
#pragma unroll 1
    for (int i = 0; i < kSlAhead && i < niter; ++i) stage_next();
    // the K / V fragments move to a[128:191] (this also places hipcc's wait for their loads in front of the loop)
    {
        const uint4_t w_ = __builtin_bit_cast(uint4_t, bk[0][0]);
        asm volatile("v_accvgpr_write_b32 a128, %0\n\tv_accvgpr_write_b32 a129, %1\n\t"
                     "v_accvgpr_write_b32 a130, %2\n\tv_accvgpr_write_b32 a131, %3"
                     :
                     : "v"(w_[0]), "v"(w_[1]), "v"(w_[2]), "v"(w_[3])
                     : "a128", "a129", "a130", "a131");
    }
    {
        const uint4_t w_ = __builtin_bit_cast(uint4_t, bk[0][1]);
        asm volatile("v_accvgpr_write_b32 a132, %0\n\tv_accvgpr_write_b32 a133, %1\n\t"
                     "v_accvgpr_write_b32 a134, %2\n\tv_accvgpr_write_b32 a135, %3"
                     :
                     : "v"(w_[0]), "v"(w_[1]), "v"(w_[2]), "v"(w_[3])
                     : "a132", "a133", "a134", "a135");
    }
    {
        const uint4_t w_ = __builtin_bit_cast(uint4_t, bk[1][0]);
        asm volatile("v_accvgpr_write_b32 a136, %0\n\tv_accvgpr_write_b32 a137, %1\n\t"
                     "v_accvgpr_write_b32 a138, %2\n\tv_accvgpr_write_b32 a139, %3"
                     :
                     : "v"(w_[0]), "v"(w_[1]), "v"(w_[2]), "v"(w_[3])
                     : "a136", "a137", "a138", "a139");
    }
    {
        const uint4_t w_ = __builtin_bit_cast(uint4_t, bk[1][1]);
        asm volatile("v_accvgpr_write_b32 a140, %0\n\tv_accvgpr_write_b32 a141, %1\n\t"
                     "v_accvgpr_write_b32 a142, %2\n\tv_accvgpr_write_b32 a143, %3"
                     :
                     : "v"(w_[0]), "v"(w_[1]), "v"(w_[2]), "v"(w_[3])
                     : "a140", "a141", "a142", "a143");
    }
    {
        const uint4_t w_ = __builtin_bit_cast(uint4_t, bk[2][0]);
        asm volatile("v_accvgpr_write_b32 a144, %0\n\tv_accvgpr_write_b32 a145, %1\n\t"
                     "v_accvgpr_write_b32 a146, %2\n\tv_accvgpr_write_b32 a147, %3"
                     :
                     : "v"(w_[0]), "v"(w_[1]), "v"(w_[2]), "v"(w_[3])
                     : "a144", "a145", "a146", "a147");
    }
    {
        const uint4_t w_ = __builtin_bit_cast(uint4_t, bk[2][1]);
        asm volatile("v_accvgpr_write_b32 a148, %0\n\tv_accvgpr_write_b32 a149, %1\n\t"
                     "v_accvgpr_write_b32 a150, %2\n\tv_accvgpr_write_b32 a151, %3"
                     :
                     : "v"(w_[0]), "v"(w_[1]), "v"(w_[2]), "v"(w_[3])
                     : "a148", "a149", "a150", "a151");
    }
    {
        const uint4_t w_ = __builtin_bit_cast(uint4_t, bk[3][0]);
        asm volatile("v_accvgpr_write_b32 a152, %0\n\tv_accvgpr_write_b32 a153, %1\n\t"
                     "v_accvgpr_write_b32 a154, %2\n\tv_accvgpr_write_b32 a155, %3"
                     :
                     : "v"(w_[0]), "v"(w_[1]), "v"(w_[2]), "v"(w_[3])
                     : "a152", "a153", "a154", "a155");
    }
    {
        const uint4_t w_ = __builtin_bit_cast(uint4_t, bk[3][1]);
        asm volatile("v_accvgpr_write_b32 a156, %0\n\tv_accvgpr_write_b32 a157, %1\n\t"
                     "v_accvgpr_write_b32 a158, %2\n\tv_accvgpr_write_b32 a159, %3"
                     :
                     : "v"(w_[0]), "v"(w_[1]), "v"(w_[2]), "v"(w_[3])
                     : "a156", "a157", "a158", "a159");
    }
    {
        const uint4_t w_ = __builtin_bit_cast(uint4_t, bv[0][0]);
        asm volatile("v_accvgpr_write_b32 a160, %0\n\tv_accvgpr_write_b32 a161, %1\n\t"
                     "v_accvgpr_write_b32 a162, %2\n\tv_accvgpr_write_b32 a163, %3"
                     :
                     : "v"(w_[0]), "v"(w_[1]), "v"(w_[2]), "v"(w_[3])
                     : "a160", "a161", "a162", "a163");
    }
    {
        const uint4_t w_ = __builtin_bit_cast(uint4_t, bv[0][1]);
        asm volatile("v_accvgpr_write_b32 a164, %0\n\tv_accvgpr_write_b32 a165, %1\n\t"
                     "v_accvgpr_write_b32 a166, %2\n\tv_accvgpr_write_b32 a167, %3"
                     :
                     : "v"(w_[0]), "v"(w_[1]), "v"(w_[2]), "v"(w_[3])
                     : "a164", "a165", "a166", "a167");
    }
    {
        const uint4_t w_ = __builtin_bit_cast(uint4_t, bv[1][0]);
        asm volatile("v_accvgpr_write_b32 a168, %0\n\tv_accvgpr_write_b32 a169, %1\n\t"
                     "v_accvgpr_write_b32 a170, %2\n\tv_accvgpr_write_b32 a171, %3"
                     :
                     : "v"(w_[0]), "v"(w_[1]), "v"(w_[2]), "v"(w_[3])
                     : "a168", "a169", "a170", "a171");
    }
    {
        const uint4_t w_ = __builtin_bit_cast(uint4_t, bv[1][1]);
        asm volatile("v_accvgpr_write_b32 a172, %0\n\tv_accvgpr_write_b32 a173, %1\n\t"
                     "v_accvgpr_write_b32 a174, %2\n\tv_accvgpr_write_b32 a175, %3"
                     :
                     : "v"(w_[0]), "v"(w_[1]), "v"(w_[2]), "v"(w_[3])
                     : "a172", "a173", "a174", "a175");
    }
    {
        const uint4_t w_ = __builtin_bit_cast(uint4_t, bv[2][0]);
        asm volatile("v_accvgpr_write_b32 a176, %0\n\tv_accvgpr_write_b32 a177, %1\n\t"
                     "v_accvgpr_write_b32 a178, %2\n\tv_accvgpr_write_b32 a179, %3"
                     :
                     : "v"(w_[0]), "v"(w_[1]), "v"(w_[2]), "v"(w_[3])
                     : "a176", "a177", "a178", "a179");
    }
    {
        const uint4_t w_ = __builtin_bit_cast(uint4_t, bv[2][1]);
        asm volatile("v_accvgpr_write_b32 a180, %0\n\tv_accvgpr_write_b32 a181, %1\n\t"
                     "v_accvgpr_write_b32 a182, %2\n\tv_accvgpr_write_b32 a183, %3"
                     :
                     : "v"(w_[0]), "v"(w_[1]), "v"(w_[2]), "v"(w_[3])
                     : "a180", "a181", "a182", "a183");
    }
    {
        const uint4_t w_ = __builtin_bit_cast(uint4_t, bv[3][0]);
        asm volatile("v_accvgpr_write_b32 a184, %0\n\tv_accvgpr_write_b32 a185, %1\n\t"
                     "v_accvgpr_write_b32 a186, %2\n\tv_accvgpr_write_b32 a187, %3"
                     :
                     : "v"(w_[0]), "v"(w_[1]), "v"(w_[2]), "v"(w_[3])
                     : "a184", "a185", "a186", "a187");
    }
    {
        const uint4_t w_ = __builtin_bit_cast(uint4_t, bv[3][1]);
        asm volatile("v_accvgpr_write_b32 a188, %0\n\tv_accvgpr_write_b32 a189, %1\n\t"
                     "v_accvgpr_write_b32 a190, %2\n\tv_accvgpr_write_b32 a191, %3"
                     :
                     : "v"(w_[0]), "v"(w_[1]), "v"(w_[2]), "v"(w_[3])
                     : "a188", "a189", "a190", "a191");
    }
    // dK^T / dV^T accumulators: a[0:127], zeroed here (AGPR map in the comment above the kernel)
    asm volatile(
        "v_accvgpr_write_b32 a0, 0\n\t"
        "v_accvgpr_write_b32 a1, 0\n\t"
        "v_accvgpr_write_b32 a2, 0\n\t"
        "v_accvgpr_write_b32 a3, 0\n\t"
        "v_accvgpr_write_b32 a4, 0\n\t"
        "v_accvgpr_write_b32 a5, 0\n\t"
        "v_accvgpr_write_b32 a6, 0\n\t"
        "v_accvgpr_write_b32 a7, 0\n\t"
        "v_accvgpr_write_b32 a8, 0\n\t"
        "v_accvgpr_write_b32 a9, 0\n\t"
        "v_accvgpr_write_b32 a10, 0\n\t"
        "v_accvgpr_write_b32 a11, 0\n\t"
        "v_accvgpr_write_b32 a12, 0\n\t"
        "v_accvgpr_write_b32 a13, 0\n\t"
        "v_accvgpr_write_b32 a14, 0\n\t"
        "v_accvgpr_write_b32 a15, 0\n\t"
        "v_accvgpr_write_b32 a16, 0\n\t"
        "v_accvgpr_write_b32 a17, 0\n\t"
        "v_accvgpr_write_b32 a18, 0\n\t"
        "v_accvgpr_write_b32 a19, 0\n\t"
        "v_accvgpr_write_b32 a20, 0\n\t"
        "v_accvgpr_write_b32 a21, 0\n\t"
        "v_accvgpr_write_b32 a22, 0\n\t"
        "v_accvgpr_write_b32 a23, 0\n\t"
        "v_accvgpr_write_b32 a24, 0\n\t"
        "v_accvgpr_write_b32 a25, 0\n\t"
        "v_accvgpr_write_b32 a26, 0\n\t"
        "v_accvgpr_write_b32 a27, 0\n\t"
        "v_accvgpr_write_b32 a28, 0\n\t"
        "v_accvgpr_write_b32 a29, 0\n\t"
        "v_accvgpr_write_b32 a30, 0\n\t"
        "v_accvgpr_write_b32 a31, 0\n\t"
        "v_accvgpr_write_b32 a32, 0\n\t"
        "v_accvgpr_write_b32 a33, 0\n\t"
        "v_accvgpr_write_b32 a34, 0\n\t"
        "v_accvgpr_write_b32 a35, 0\n\t"
        "v_accvgpr_write_b32 a36, 0\n\t"
        "v_accvgpr_write_b32 a37, 0\n\t"
        "v_accvgpr_write_b32 a38, 0\n\t"
        "v_accvgpr_write_b32 a39, 0\n\t"
        "v_accvgpr_write_b32 a40, 0\n\t"
        "v_accvgpr_write_b32 a41, 0\n\t"
        "v_accvgpr_write_b32 a42, 0\n\t"
        "v_accvgpr_write_b32 a43, 0\n\t"
        "v_accvgpr_write_b32 a44, 0\n\t"
        "v_accvgpr_write_b32 a45, 0\n\t"
        "v_accvgpr_write_b32 a46, 0\n\t"
        "v_accvgpr_write_b32 a47, 0\n\t"
        "v_accvgpr_write_b32 a48, 0\n\t"
        "v_accvgpr_write_b32 a49, 0\n\t"
        "v_accvgpr_write_b32 a50, 0\n\t"
        "v_accvgpr_write_b32 a51, 0\n\t"
        "v_accvgpr_write_b32 a52, 0\n\t"
        "v_accvgpr_write_b32 a53, 0\n\t"
        "v_accvgpr_write_b32 a54, 0\n\t"
        "v_accvgpr_write_b32 a55, 0\n\t"
        "v_accvgpr_write_b32 a56, 0\n\t"
        "v_accvgpr_write_b32 a57, 0\n\t"
        "v_accvgpr_write_b32 a58, 0\n\t"
        "v_accvgpr_write_b32 a59, 0\n\t"
        "v_accvgpr_write_b32 a60, 0\n\t"
        "v_accvgpr_write_b32 a61, 0\n\t"
        "v_accvgpr_write_b32 a62, 0\n\t"
        "v_accvgpr_write_b32 a63, 0\n\t"
        "v_accvgpr_write_b32 a64, 0\n\t"
        "v_accvgpr_write_b32 a65, 0\n\t"
        "v_accvgpr_write_b32 a66, 0\n\t"
        "v_accvgpr_write_b32 a67, 0\n\t"
        "v_accvgpr_write_b32 a68, 0\n\t"
        "v_accvgpr_write_b32 a69, 0\n\t"
        "v_accvgpr_write_b32 a70, 0\n\t"
        "v_accvgpr_write_b32 a71, 0\n\t"
        "v_accvgpr_write_b32 a72, 0\n\t"
        "v_accvgpr_write_b32 a73, 0\n\t"
        "v_accvgpr_write_b32 a74, 0\n\t"
        "v_accvgpr_write_b32 a75, 0\n\t"
        "v_accvgpr_write_b32 a76, 0\n\t"
        "v_accvgpr_write_b32 a77, 0\n\t"
        "v_accvgpr_write_b32 a78, 0\n\t"
        "v_accvgpr_write_b32 a79, 0\n\t"
        "v_accvgpr_write_b32 a80, 0\n\t"
        "v_accvgpr_write_b32 a81, 0\n\t"
        "v_accvgpr_write_b32 a82, 0\n\t"
        "v_accvgpr_write_b32 a83, 0\n\t"
        "v_accvgpr_write_b32 a84, 0\n\t"
        "v_accvgpr_write_b32 a85, 0\n\t"
        "v_accvgpr_write_b32 a86, 0\n\t"
        "v_accvgpr_write_b32 a87, 0\n\t"
        "v_accvgpr_write_b32 a88, 0\n\t"
        "v_accvgpr_write_b32 a89, 0\n\t"
        "v_accvgpr_write_b32 a90, 0\n\t"
        "v_accvgpr_write_b32 a91, 0\n\t"
        "v_accvgpr_write_b32 a92, 0\n\t"
        "v_accvgpr_write_b32 a93, 0\n\t"
        "v_accvgpr_write_b32 a94, 0\n\t"
        "v_accvgpr_write_b32 a95, 0\n\t"
        "v_accvgpr_write_b32 a96, 0\n\t"
        "v_accvgpr_write_b32 a97, 0\n\t"
        "v_accvgpr_write_b32 a98, 0\n\t"
        "v_accvgpr_write_b32 a99, 0\n\t"
        "v_accvgpr_write_b32 a100, 0\n\t"
        "v_accvgpr_write_b32 a101, 0\n\t"
        "v_accvgpr_write_b32 a102, 0\n\t"
        "v_accvgpr_write_b32 a103, 0\n\t"
        "v_accvgpr_write_b32 a104, 0\n\t"
        "v_accvgpr_write_b32 a105, 0\n\t"
        "v_accvgpr_write_b32 a106, 0\n\t"
        "v_accvgpr_write_b32 a107, 0\n\t"
        "v_accvgpr_write_b32 a108, 0\n\t"
        "v_accvgpr_write_b32 a109, 0\n\t"
        "v_accvgpr_write_b32 a110, 0\n\t"
        "v_accvgpr_write_b32 a111, 0\n\t"
        "v_accvgpr_write_b32 a112, 0\n\t"
        "v_accvgpr_write_b32 a113, 0\n\t"
        "v_accvgpr_write_b32 a114, 0\n\t"
        "v_accvgpr_write_b32 a115, 0\n\t"
        "v_accvgpr_write_b32 a116, 0\n\t"
        "v_accvgpr_write_b32 a117, 0\n\t"
        "v_accvgpr_write_b32 a118, 0\n\t"
        "v_accvgpr_write_b32 a119, 0\n\t"
        "v_accvgpr_write_b32 a120, 0\n\t"
        "v_accvgpr_write_b32 a121, 0\n\t"
        "v_accvgpr_write_b32 a122, 0\n\t"
        "v_accvgpr_write_b32 a123, 0\n\t"
        "v_accvgpr_write_b32 a124, 0\n\t"
        "v_accvgpr_write_b32 a125, 0\n\t"
        "v_accvgpr_write_b32 a126, 0\n\t"
        "v_accvgpr_write_b32 a127, 0"
        :
        :
        : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127");
    const int qq = fr >> 2, pp = fr & 3;
    const int xs = (pp >> 1) ^ (4 * (g & 1) + qq);
    unsigned tr_off[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) tr_off[c] = (4 * g + qq) * 128 + (((2 * c) ^ xs) << 4) + 8 * (pp & 1);
    unsigned row_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) row_off[ks] = fr * 128 + (((4 * ks + g) ^ (fr & 7)) << 4);
    const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int per_stage = wave == 0 ? 3 : 2;               // DMA instructions this wave issues per slice

    // Loop: one raw barrier per slice, slice it + 4 issued behind it, then
    //   common case (slice fully visible to the wave's keys, no masking): RPO_D4_SLICE_BODY_*, ONE hand-placed instruction
    //   stream = M1 key-tile major, the exp2 / dS arithmetic of key tile n in the issue gaps of the MFMAs that follow its
    //   chains, M2 key-tile major behind it, and the LDS reads of the NEXT slice's row operands under M2;
    //   diagonal / last slices: the same steps as separate statements with hipcc's code for the masked arithmetic.
    // hipcc must not spill into a[0:191] (it does not know them to be occupied): its resource line must show 0 scratch, and
    // any v_accvgpr_* outside ASMSTART / ASMEND may only name a192 and up.
    auto wait_landed = [&](int upto) {                        // every stage <= upto of THIS wave has landed
        const int later = (st_n - 1 - upto) * per_stage;     // DMA instructions issued after it: 0 .. 9
        if (later >= 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else if (later == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (later == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (later == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (later == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    bool hot = false;      // v[128:175] hold the row fragments / row constants of slice `it` (prefetched by the previous body)
    int cur = 0, sl = 0;
#ifdef RPO_D4_EXP_DQ_ATOMICS
    int exp_hq = hk * group;
#endif
    RPO_LAD(1);
    for (int it = 0; it < niter; ++it) {
        if constexpr (DOWN) sl = nsl - 1 - (it >> gshift);
        // slices <= it + 1 have landed (the body prefetches from the next image).  Steady state (slices still being staged):
        // exactly two later stages are in flight, one compare instead of the general ladder
        if (st_n < niter) {
#ifdef RPO_D4_EXP_DQ_ATOMICS
            // + the 8 atomics of each of the iterations it - 3, it - 2, it - 1 (younger than the stage that must have landed)
            if (it >= 3) {
                if (wave == 0) asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
            } else
#endif
            if (wave == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            wait_landed(it + 1 < niter ? it + 1 : it);
        }
        __builtin_amdgcn_s_barrier();
        if (st_n < niter) stage_next();                      // slice it + 4 -> the image of slice it - 4
        const int qb = qt0 + sl * kSl;
        const bool active = (qb + kSl - 1 >= k0) && (k0 < len);
        // three kinds of active slices (round 3): plain; the two DIAGONAL ones of the wave (the causal boundary crosses them, at
        // different slices for each wave of the block: the same hand-placed stream with the mask folded into the S' chains'
        // initial accumulators, so that the block does not wait at the next barrier for one wave on hipcc's slower path); the
        // tail (sequence end inside the slice or inside the wave's keys: the select-based code below)
        const bool tail = (qb + kSl > len) || (k0 + 64 > len);
        const bool diag = qb < k0 + 63;
        const unsigned img = smem_base + cur * kSlImg;
        const unsigned next_img = smem_base + ((cur + 1) & (kSlRing - 1)) * kSlImg;
        if (active && !tail) {
            if (!diag) {
                if (hot)
                    RPO_D4_SLICE_BODY_HOT(img + row_off[0], img + row_off[1], img + 2 * kSl * 128 + 16 * g, img + tr_off[0],
                                          img + tr_off[1], img + tr_off[2], img + tr_off[3], scale_log2e,
                                          next_img + row_off[0], next_img + row_off[1], next_img + 2 * kSl * 128 + 16 * g);
                else
                    RPO_D4_SLICE_BODY_LOAD(img + row_off[0], img + row_off[1], img + 2 * kSl * 128 + 16 * g, img + tr_off[0],
                                           img + tr_off[1], img + tr_off[2], img + tr_off[3], scale_log2e,
                                           next_img + row_off[0], next_img + row_off[1], next_img + 2 * kSl * 128 + 16 * g);
            } else {
                const int dlane = (k0 + fr) - (qb + 4 * g);      // the lane's key (tile 0) minus its first query row
                if (hot)
                    RPO_D4_DIAG_BODY_HOT(img + row_off[0], img + row_off[1], img + 2 * kSl * 128 + 16 * g, img + tr_off[0],
                                         img + tr_off[1], img + tr_off[2], img + tr_off[3], scale_log2e,
                                         next_img + row_off[0], next_img + row_off[1], next_img + 2 * kSl * 128 + 16 * g, dlane);
                else
                    RPO_D4_DIAG_BODY_LOAD(img + row_off[0], img + row_off[1], img + 2 * kSl * 128 + 16 * g, img + tr_off[0],
                                          img + tr_off[1], img + tr_off[2], img + tr_off[3], scale_log2e,
                                          next_img + row_off[0], next_img + row_off[1], next_img + 2 * kSl * 128 + 16 * g, dlane);
            }
            hot = true;
        } else if (active) {
            hot = false;
            // M1: S' = Q K^T - lse / scale, dP' = dO V^T - delta (row constants = initial accumulators)
            float4_t s[2][4], dp[2][4];                       // rows = queries 16 m + 4 g + r, col = key 16 n + fr
            const char* Qs = smem + cur * kSlImg;
            const char* Ds = Qs + kSl * 128;
            const float* Ls = reinterpret_cast<const float*>(Qs + 2 * kSl * 128);
            const short8_t aq00 = *reinterpret_cast<const short8_t*>(Qs + row_off[0]);
            const short8_t ad00 = *reinterpret_cast<const short8_t*>(Ds + row_off[0]);
            const float4_t lr0 = *reinterpret_cast<const float4_t*>(Ls + 4 * g);
            const float4_t dr0 = *reinterpret_cast<const float4_t*>(Ls + kSl + 4 * g);
            const short8_t aq01 = *reinterpret_cast<const short8_t*>(Qs + row_off[0] + 2048);
            const short8_t ad01 = *reinterpret_cast<const short8_t*>(Ds + row_off[0] + 2048);
            const float4_t lr1 = *reinterpret_cast<const float4_t*>(Ls + 16 + 4 * g);
            const float4_t dr1 = *reinterpret_cast<const float4_t*>(Ls + kSl + 16 + 4 * g);
            const short8_t aq10 = *reinterpret_cast<const short8_t*>(Qs + row_off[1]);
            const short8_t ad10 = *reinterpret_cast<const short8_t*>(Ds + row_off[1]);
            const short8_t aq11 = *reinterpret_cast<const short8_t*>(Qs + row_off[1] + 2048);
            const short8_t ad11 = *reinterpret_cast<const short8_t*>(Ds + row_off[1] + 2048);
            // transposed fragments of the slice (A operands of M2): dO^T at image offset 4096, queries 16..31 at + 2048
            u32x2 d0, d1, d2, d3, d4, d5, d6, d7, e0, e1, e2, e3, e4, e5, e6, e7;
            {
                const unsigned a0 = img + tr_off[0], a1 = img + tr_off[1], a2 = img + tr_off[2], a3 = img + tr_off[3];
                RPO_TR2(d0, d1, a0, 4096, 6144);
                RPO_TR2(d2, d3, a1, 4096, 6144);
                RPO_TR2(d4, d5, a2, 4096, 6144);
                RPO_TR2(d6, d7, a3, 4096, 6144);
                RPO_TR2(e0, e1, a0, 0, 2048);
                RPO_TR2(e2, e3, a1, 0, 2048);
                RPO_TR2(e4, e5, a2, 0, 2048);
                RPO_TR2(e6, e7, a3, 0, 2048);
            }
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                s[0][n] = lr0;
                dp[0][n] = dr0;
                s[1][n] = lr1;
                dp[1][n] = dr1;
            }
            RPO_D4_M1_KS0(0, aq00, ad00);
            RPO_D4_M1_KS0(1, aq01, ad01);
            RPO_D4_M1_KS1(0, aq10, ad10);
            RPO_D4_M1_KS1(1, aq11, ad11);
            // an MFMA result is readable by the VALU 12 states after issue
            asm volatile("s_nop 11"
                         : "+v"(s[0][0]), "+v"(s[0][1]), "+v"(s[0][2]), "+v"(s[0][3]), "+v"(s[1][0]), "+v"(s[1][1]),
                           "+v"(s[1][2]), "+v"(s[1][3]), "+v"(dp[0][0]), "+v"(dp[0][1]), "+v"(dp[0][2]), "+v"(dp[0][3]),
                           "+v"(dp[1][0]), "+v"(dp[1][1]), "+v"(dp[1][2]), "+v"(dp[1][3]));
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int qr0 = qb + 16 * m + 4 * g;
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const int key = k0 + 16 * n + fr;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float pv = __builtin_amdgcn_exp2f(s[m][n][r] * scale_log2e);
                        pv = (key > qr0 + r || key >= len || qr0 + r >= len) ? 0.f : pv;      // select, no branch
                        s[m][n][r] = pv;
                        dp[m][n][r] = pv * dp[m][n][r];
                    }
                }
            }
            short8_t pf[4], dsf[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                pf[n] = pack_frag(s[0][n], s[1][n]);      // k-slots = queries {4g + j, 16 + 4g + (j - 4)} of the slice
                dsf[n] = pack_frag(dp[0][n], dp[1][n]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(e0),
                           "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7)
                         :
                         : "memory");
            const short8_t atd[4] = {join_tr(d0, d1), join_tr(d2, d3), join_tr(d4, d5), join_tr(d6, d7)};
            const short8_t atq[4] = {join_tr(e0, e1), join_tr(e2, e3), join_tr(e4, e5), join_tr(e6, e7)};
            RPO_D4_M2_N0(atd, atq, pf, dsf);
            RPO_D4_M2_N1(atd, atq, pf, dsf);
            RPO_D4_M2_N2(atd, atq, pf, dsf);
            RPO_D4_M2_N3(atd, atq, pf, dsf);
        } else {
            hot = false;
        }
#ifdef RPO_D4_EXP_DQ_ATOMICS
        {
            const int hq_it = DOWN ? hk * group + (it & (group - 1)) : exp_hq;
            const int row0 = min(qb + 8 * wave, len > 8 ? len - 8 : 0);
            uint64_t dst = (uint64_t)(g_exp_dq32 + ((t0 + row0) * (int64_t)nh + hq_it) * kFaHD);
            const unsigned lane4 = 4u * (unsigned)lane;
            const float zero = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                asm volatile("global_atomic_add_f32 %0, %1, %2" : : "v"(lane4), "v"(zero), "s"(dst) : "memory");
                dst += (uint64_t)nh * kFaHD * 4u;
            }
        }
#endif
        cur = (cur + 1) & (kSlRing - 1);
        if constexpr (!DOWN) {
#ifdef RPO_D4_EXP_DQ_ATOMICS
            if (++sl == nsl) { sl = 0; ++exp_hq; }
#else
            if (++sl == nsl) sl = 0;
#endif
        }
    }
    RPO_LAD(2);
    // epilogue: dK[key][16 c + 4 g + r] = scale * dka, dV likewise (unscaled); the wave owns its keys: no reduction
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");       // the last MFMAs' results are readable
    float4_t dka[4][4], dva[4][4];
    {
        float f0_, f1_, f2_, f3_;
        asm volatile("v_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1\n\t"
                     "v_accvgpr_read_b32 %2, a2\n\tv_accvgpr_read_b32 %3, a3"
                     : "=v"(f0_), "=v"(f1_), "=v"(f2_), "=v"(f3_));
        dva[0][0] = float4_t{f0_, f1_, f2_, f3_};
    }
    {
        float f0_, f1_, f2_, f3_;
        asm volatile("v_accvgpr_read_b32 %0, a64\n\tv_accvgpr_read_b32 %1, a65\n\t"
                     "v_accvgpr_read_b32 %2, a66\n\tv_accvgpr_read_b32 %3, a67"
                     : "=v"(f0_), "=v"(f1_), "=v"(f2_), "=v"(f3_));
        dka[0][0] = float4_t{f0_, f1_, f2_, f3_};
    }
    {
        float f0_, f1_, f2_, f3_;
        asm volatile("v_accvgpr_read_b32 %0, a4\n\tv_accvgpr_read_b32 %1, a5\n\t"
                     "v_accvgpr_read_b32 %2, a6\n\tv_accvgpr_read_b32 %3, a7"
                     : "=v"(f0_), "=v"(f1_), "=v"(f2_), "=v"(f3_));
        dva[0][1] = float4_t{f0_, f1_, f2_, f3_};
    }
    {
        float f0_, f1_, f2_, f3_;
        asm volatile("v_accvgpr_read_b32 %0, a68\n\tv_accvgpr_read_b32 %1, a69\n\t"
                     "v_accvgpr_read_b32 %2, a70\n\tv_accvgpr_read_b32 %3, a71"
                     : "=v"(f0_), "=v"(f1_), "=v"(f2_), "=v"(f3_));
        dka[0][1] = float4_t{f0_, f1_, f2_, f3_};
    }
    {
        float f0_, f1_, f2_, f3_;
        asm volatile("v_accvgpr_read_b32 %0, a8\n\tv_accvgpr_read_b32 %1, a9\n\t"
                     "v_accvgpr_read_b32 %2, a10\n\tv_accvgpr_read_b32 %3, a11"
                     : "=v"(f0_), "=v"(f1_), "=v"(f2_), "=v"(f3_));
        dva[0][2] = float4_t{f0_, f1_, f2_, f3_};
    }
    {
        float f0_, f1_, f2_, f3_;
        asm volatile("v_accvgpr_read_b32 %0, a72\n\tv_accvgpr_read_b32 %1, a73\n\t"
                     "v_accvgpr_read_b32 %2, a74\n\tv_accvgpr_read_b32 %3, a75"
                     : "=v"(f0_), "=v"(f1_), "=v"(f2_), "=v"(f3_));
        dka[0][2] = float4_t{f0_, f1_, f2_, f3_};
    }
    {
        float f0_, f1_, f2_, f3_;
        asm volatile("v_accvgpr_read_b32 %0, a12\n\tv_accvgpr_read_b32 %1, a13\n\t"
                     "v_accvgpr_read_b32 %2, a14\n\tv_accvgpr_read_b32 %3, a15"
                     : "=v"(f0_), "=v"(f1_), "=v"(f2_), "=v"(f3_));
        dva[0][3] = float4_t{f0_, f1_, f2_, f3_};
    }
    {
        float f0_, f1_, f2_, f3_;
        asm volatile("v_accvgpr_read_b32 %0, a76\n\tv_accvgpr_read_b32 %1, a77\n\t"
                     "v_accvgpr_read_b32 %2, a78\n\tv_accvgpr_read_b32 %3, a79"
                     : "=v"(f0_), "=v"(f1_), "=v"(f2_), "=v"(f3_));
        dka[0][3] = float4_t{f0_, f1_, f2_, f3_};
    }
    {
        float f0_, f1_, f2_, f3_;
        asm volatile("v_accvgpr_read_b32 %0, a16\n\tv_accvgpr_read_b32 %1, a17\n\t"
                     "v_accvgpr_read_b32 %2, a18\n\tv_accvgpr_read_b32 %3, a19"
                     : "=v"(f0_), "=v"(f1_), "=v"(f2_), "=v"(f3_));
        dva[1][0] = float4_t{f0_, f1_, f2_, f3_};
    }
    {
        float f0_, f1_, f2_, f3_;
        asm volatile("v_accvgpr_read_b32 %0, a80\n\tv_accvgpr_read_b32 %1, a81\n\t"
                     "v_accvgpr_read_b32 %2, a82\n\tv_accvgpr_read_b32 %3, a83"
                     : "=v"(f0_), "=v"(f1_), "=v"(f2_), "=v"(f3_));
        dka[1][0] = float4_t{f0_, f1_, f2_, f3_};
    }
    {
        float f0_, f1_, f2_, f3_;
        asm volatile("v_accvgpr_read_b32 %0, a20\n\tv_accvgpr_read_b32 %1, a21\n\t"
                     "v_accvgpr_read_b32 %2, a22\n\tv_accvgpr_read_b32 %3, a23"
                     : "=v"(f0_), "=v"(f1_), "=v"(f2_), "=v"(f3_));
        dva[1][1] = float4_t{f0_, f1_, f2_, f3_};
    }
    {
        float f0_, f1_, f2_, f3_;
        asm volatile("v_accvgpr_read_b32 %0, a84\n\tv_accvgpr_read_b32 %1, a85\n\t"
                     "v_accvgpr_read_b32 %2, a86\n\tv_accvgpr_read_b32 %3, a87"
                     : "=v"(f0_), "=v"(f1_), "=v"(f2_), "=v"(f3_));
        dka[1][1] = float4_t{f0_, f1_, f2_, f3_};
    }
    {
        float f0_, f1_, f2_, f3_;
        asm volatile("v_accvgpr_read_b32 %0, a24\n\tv_accvgpr_read_b32 %1, a25\n\t"
                     "v_accvgpr_read_b32 %2, a26\n\tv_accvgpr_read_b32 %3, a27"
                     : "=v"(f0_), "=v"(f1_), "=v"(f2_), "=v"(f3_));
        dva[1][2] = float4_t{f0_, f1_, f2_, f3_};
    }
    {
        float f0_, f1_, f2_, f3_;
        asm volatile("v_accvgpr_read_b32 %0, a88\n\tv_accvgpr_read_b32 %1, a89\n\t"
                     "v_accvgpr_read_b32 %2, a90\n\tv_accvgpr_read_b32 %3, a91"
                     : "=v"(f0_), "=v"(f1_), "=v"(f2_), "=v"(f3_));
        dka[1][2] = float4_t{f0_, f1_, f2_, f3_};
    }
    {
        float f0_, f1_, f2_, f3_;
        asm volatile("v_accvgpr_read_b32 %0, a28\n\tv_accvgpr_read_b32 %1, a29\n\t"
                     "v_accvgpr_read_b32 %2, a30\n\tv_accvgpr_read_b32 %3, a31"
                     : "=v"(f0_), "=v"(f1_), "=v"(f2_), "=v"(f3_));
        dva[1][3] = float4_t{f0_, f1_, f2_, f3_};
    }
    {
        float f0_, f1_, f2_, f3_;
        asm volatile("v_accvgpr_read_b32 %0, a92\n\tv_accvgpr_read_b32 %1, a93\n\t"
                     "v_accvgpr_read_b32 %2, a94\n\tv_accvgpr_read_b32 %3, a95"
                     : "=v"(f0_), "=v"(f1_), "=v"(f2_), "=v"(f3_));
        dka[1][3] = float4_t{f0_, f1_, f2_, f3_};
    }
    {
        float f0_, f1_, f2_, f3_;
        asm volatile("v_accvgpr_read_b32 %0, a32\n\tv_accvgpr_read_b32 %1, a33\n\t"
                     "v_accvgpr_read_b32 %2, a34\n\tv_accvgpr_read_b32 %3, a35"
                     : "=v"(f0_), "=v"(f1_), "=v"(f2_), "=v"(f3_));
        dva[2][0] = float4_t{f0_, f1_, f2_, f3_};
    }
    {
        float f0_, f1_, f2_, f3_;
        asm volatile("v_accvgpr_read_b32 %0, a96\n\tv_accvgpr_read_b32 %1, a97\n\t"
                     "v_accvgpr_read_b32 %2, a98\n\tv_accvgpr_read_b32 %3, a99"
                     : "=v"(f0_), "=v"(f1_), "=v"(f2_), "=v"(f3_));
        dka[2][0] = float4_t{f0_, f1_, f2_, f3_};
    }
    {
        float f0_, f1_, f2_, f3_;
        asm volatile("v_accvgpr_read_b32 %0, a36\n\tv_accvgpr_read_b32 %1, a37\n\t"
                     "v_accvgpr_read_b32 %2, a38\n\tv_accvgpr_read_b32 %3, a39"
                     : "=v"(f0_), "=v"(f1_), "=v"(f2_), "=v"(f3_));
        dva[2][1] = float4_t{f0_, f1_, f2_, f3_};
    }
    {
        float f0_, f1_, f2_, f3_;
        asm volatile("v_accvgpr_read_b32 %0, a100\n\tv_accvgpr_read_b32 %1, a101\n\t"
                     "v_accvgpr_read_b32 %2, a102\n\tv_accvgpr_read_b32 %3, a103"
                     : "=v"(f0_), "=v"(f1_), "=v"(f2_), "=v"(f3_));
        dka[2][1] = float4_t{f0_, f1_, f2_, f3_};
    }
    {
        float f0_, f1_, f2_, f3_;
        asm volatile("v_accvgpr_read_b32 %0, a40\n\tv_accvgpr_read_b32 %1, a41\n\t"
                     "v_accvgpr_read_b32 %2, a42\n\tv_accvgpr_read_b32 %3, a43"
                     : "=v"(f0_), "=v"(f1_), "=v"(f2_), "=v"(f3_));
        dva[2][2] = float4_t{f0_, f1_, f2_, f3_};
    }
    {
        float f0_, f1_, f2_, f3_;
        asm volatile("v_accvgpr_read_b32 %0, a104\n\tv_accvgpr_read_b32 %1, a105\n\t"
                     "v_accvgpr_read_b32 %2, a106\n\tv_accvgpr_read_b32 %3, a107"
                     : "=v"(f0_), "=v"(f1_), "=v"(f2_), "=v"(f3_));
        dka[2][2] = float4_t{f0_, f1_, f2_, f3_};
    }
    {
        float f0_, f1_, f2_, f3_;
        asm volatile("v_accvgpr_read_b32 %0, a44\n\tv_accvgpr_read_b32 %1, a45\n\t"
                     "v_accvgpr_read_b32 %2, a46\n\tv_accvgpr_read_b32 %3, a47"
                     : "=v"(f0_), "=v"(f1_), "=v"(f2_), "=v"(f3_));
        dva[2][3] = float4_t{f0_, f1_, f2_, f3_};
    }
    {
        float f0_, f1_, f2_, f3_;
        asm volatile("v_accvgpr_read_b32 %0, a108\n\tv_accvgpr_read_b32 %1, a109\n\t"
                     "v_accvgpr_read_b32 %2, a110\n\tv_accvgpr_read_b32 %3, a111"
                     : "=v"(f0_), "=v"(f1_), "=v"(f2_), "=v"(f3_));
        dka[2][3] = float4_t{f0_, f1_, f2_, f3_};
    }
    {
        float f0_, f1_, f2_, f3_;
        asm volatile("v_accvgpr_read_b32 %0, a48\n\tv_accvgpr_read_b32 %1, a49\n\t"
                     "v_accvgpr_read_b32 %2, a50\n\tv_accvgpr_read_b32 %3, a51"
                     : "=v"(f0_), "=v"(f1_), "=v"(f2_), "=v"(f3_));
        dva[3][0] = float4_t{f0_, f1_, f2_, f3_};
    }
    {
        float f0_, f1_, f2_, f3_;
        asm volatile("v_accvgpr_read_b32 %0, a112\n\tv_accvgpr_read_b32 %1, a113\n\t"
                     "v_accvgpr_read_b32 %2, a114\n\tv_accvgpr_read_b32 %3, a115"
                     : "=v"(f0_), "=v"(f1_), "=v"(f2_), "=v"(f3_));
        dka[3][0] = float4_t{f0_, f1_, f2_, f3_};
    }
    {
        float f0_, f1_, f2_, f3_;
        asm volatile("v_accvgpr_read_b32 %0, a52\n\tv_accvgpr_read_b32 %1, a53\n\t"
                     "v_accvgpr_read_b32 %2, a54\n\tv_accvgpr_read_b32 %3, a55"
                     : "=v"(f0_), "=v"(f1_), "=v"(f2_), "=v"(f3_));
        dva[3][1] = float4_t{f0_, f1_, f2_, f3_};
    }
    {
        float f0_, f1_, f2_, f3_;
        asm volatile("v_accvgpr_read_b32 %0, a116\n\tv_accvgpr_read_b32 %1, a117\n\t"
                     "v_accvgpr_read_b32 %2, a118\n\tv_accvgpr_read_b32 %3, a119"
                     : "=v"(f0_), "=v"(f1_), "=v"(f2_), "=v"(f3_));
        dka[3][1] = float4_t{f0_, f1_, f2_, f3_};
    }
    {
        float f0_, f1_, f2_, f3_;
        asm volatile("v_accvgpr_read_b32 %0, a56\n\tv_accvgpr_read_b32 %1, a57\n\t"
                     "v_accvgpr_read_b32 %2, a58\n\tv_accvgpr_read_b32 %3, a59"
                     : "=v"(f0_), "=v"(f1_), "=v"(f2_), "=v"(f3_));
        dva[3][2] = float4_t{f0_, f1_, f2_, f3_};
    }
    {
        float f0_, f1_, f2_, f3_;
        asm volatile("v_accvgpr_read_b32 %0, a120\n\tv_accvgpr_read_b32 %1, a121\n\t"
                     "v_accvgpr_read_b32 %2, a122\n\tv_accvgpr_read_b32 %3, a123"
                     : "=v"(f0_), "=v"(f1_), "=v"(f2_), "=v"(f3_));
        dka[3][2] = float4_t{f0_, f1_, f2_, f3_};
    }
    {
        float f0_, f1_, f2_, f3_;
        asm volatile("v_accvgpr_read_b32 %0, a60\n\tv_accvgpr_read_b32 %1, a61\n\t"
                     "v_accvgpr_read_b32 %2, a62\n\tv_accvgpr_read_b32 %3, a63"
                     : "=v"(f0_), "=v"(f1_), "=v"(f2_), "=v"(f3_));
        dva[3][3] = float4_t{f0_, f1_, f2_, f3_};
    }
    {
        float f0_, f1_, f2_, f3_;
        asm volatile("v_accvgpr_read_b32 %0, a124\n\tv_accvgpr_read_b32 %1, a125\n\t"
                     "v_accvgpr_read_b32 %2, a126\n\tv_accvgpr_read_b32 %3, a127"
                     : "=v"(f0_), "=v"(f1_), "=v"(f2_), "=v"(f3_));
        dka[3][3] = float4_t{f0_, f1_, f2_, f3_};
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int key = k0 + 16 * n + fr;
        if (key >= len) continue;
        bf16_t* krow = dk + (t0 + key) * sdk + hk * kFaHD + 4 * g;
        bf16_t* vrow = dv + (t0 + key) * sdv + hk * kFaHD + 4 * g;
#pragma unroll
        for (int c = 0; c < 4; ++c) dka[c][n] *= scale;
        if (rcos) {                                         // dK w.r.t. the pre-rotary k (inv_rope4)
            const int64_t tr = ((t0 + key) % rperiod) * (kFaHD / 2) + 4 * g;
#pragma unroll
            for (int c = 0; c < 2; ++c)
                inv_rope4(dka[c][n], dka[c + 2][n], *reinterpret_cast<const float4_t*>(rcos + tr + 16 * c),
                          *reinterpret_cast<const float4_t*>(rsin + tr + 16 * c));
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint2 w;
            w.x = pack_bf16(dka[c][n][0], dka[c][n][1]);
            w.y = pack_bf16(dka[c][n][2], dka[c][n][3]);
            *reinterpret_cast<uint2*>(krow + 16 * c) = w;
            w.x = pack_bf16(dva[c][n][0], dva[c][n][1]);
            w.y = pack_bf16(dva[c][n][2], dva[c][n][3]);
            *reinterpret_cast<uint2*>(vrow + 16 * c) = w;
        }
    }
    RPO_LAD_END(2, niter);
}

// ------------------------------------------------------------------------------------------------------------------
// head_dim 128 backward (round 2; the Llama-3-8B architecture, BASELINE configs[4]).  Same two-launch, atomic-free scheme as at
// head_dim 64, on the forward's 256-byte LDS rows (chunk ^= 2 (row & 7), 1-KiB DMA piece = 4 rows):
//   fa_bwd_dq128_kernel    block = 128 queries of one (sequence, head), key tiles of 32 (the forward's (K | V) ring):
//                          S^T = K Q^T - lse / scale, dP^T = V dO^T - delta, dS^T = exp2(scale log2(e) S^T) dP^T,
//                          dQ^T += K^T dS^T (K^T by transposed reads of the SAME K image); 48 MFMAs per tile.
//                          Prologue: delta and the two row constants, written for the dK/dV kernel.
//   fa_bwd_dkdv128_kernel  block = 4 waves = 128 keys of one (sequence, kv head), ONE wave per SIMD: wave w owns keys
//                          [32 w, 32 w + 32) -- its K and V fragments (64 registers) and the dK^T / dV^T
//                          accumulators (128) never leave the register file, no cross-wave sum -- and the block sweeps the
//                          group's q heads x 32-row query slices (Q | dO | 64 row constants = 16.25 KiB by LDS-DMA, ring of 3):
//                          S = Q K^T, dP = dO V^T, dV^T += dO^T P, dK^T += Q^T dS; 64 MFMAs per slice and wave.
// Both are hipcc-scheduled (no hand-placed stream): 2.0x PyTorch's op on cfg 5's shape, see DESIGN.md.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kFaThreads, 2) void fa_bwd_dq128_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const bf16_t* __restrict__ dout, int64_t sq, int64_t sk, int64_t sv, int64_t sdo, const int* __restrict__ cu,
    const int* __restrict__ tiles, int tcols, int nh, int nkv, float scale_log2e, float scale,
    const float* __restrict__ lse, const bf16_t* __restrict__ o, int64_t so, float* __restrict__ nl_out,
    float* __restrict__ nd_out, int64_t T, bf16_t* __restrict__ dq, int64_t sdq, const float* __restrict__ rcos,
    const float* __restrict__ rsin, int64_t rperiod) {
    __shared__ __attribute__((aligned(16))) char smem[3 * kKvTile];      // ring of (K tile | V tile), 256-byte rows, 32 keys
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, fr = lane & 15;
    const FaTile ft = fa_tile(tiles, tcols);
    if (ft.q0 >= (1 << 30)) return;
    const int seq = ft.seq, q0 = ft.q0;
    const int h = ft.h, hk = h / (nh / nkv);
    const int64_t t0 = cu[seq];
    const int len = cu[seq + 1] - (int)t0;
    const int qw = q0 + 32 * wave;
    short8_t bq[2][4], bdo[2][4];
    float lq[2], dl[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int qi = qw + 16 * n + fr;
        const bool ok = qi < len;
        float part = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int64_t col = h * kFa128HD + 32 * ks + 8 * g;
            const short8_t z = short8_t{0, 0, 0, 0, 0, 0, 0, 0};
            bq[n][ks] = ok ? *reinterpret_cast<const short8_t*>(q + (t0 + qi) * sq + col) : z;
            bdo[n][ks] = ok ? *reinterpret_cast<const short8_t*>(dout + (t0 + qi) * sdo + col) : z;
            const short8_t ov = ok ? *reinterpret_cast<const short8_t*>(o + (t0 + qi) * so + col) : z;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                part = fmaf(bf16_to_f32((bf16_t)bdo[n][ks][e]), bf16_to_f32((bf16_t)ov[e]), part);
        }
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        const float lse_q = ok ? lse[(int64_t)h * T + t0 + qi] : 0.f;
        lq[n] = -lse_q / scale;
        dl[n] = ok ? -part : 0.f;
        if (ok && g == 0) {
            nl_out[(int64_t)h * T + t0 + qi] = lq[n];
            nd_out[(int64_t)h * T + t0 + qi] = dl[n];
        }
    }
    const int last_q = min(q0 + kFaBM - 1, len - 1);
    const int nkt = last_q / kFa128BN + 1;
    const int srow = lane >> 4;
    const char* ksrc = reinterpret_cast<const char*>(k + t0 * sk + hk * kFa128HD);
    const char* vsrc = reinterpret_cast<const char*>(v + t0 * sv + hk * kFa128HD);
    const unsigned skb = (unsigned)sk * 2u, svb = (unsigned)sv * 2u;
    auto stage = [&](int kt, int buf) {
        char* base = smem + buf * kKvTile;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int u = 2 * wave + i;
            const int trow = 4 * u + srow;
            const unsigned lchunk = (unsigned)((lane & 15) ^ (2 * (trow & 7)));
            const unsigned row = (unsigned)min(kt * kFa128BN + trow, len - 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ksrc + (row * skb + lchunk * 16)),
                                             (__attribute__((address_space(3))) void*)(base + u * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vsrc + (row * svb + lchunk * 16)),
                                             (__attribute__((address_space(3))) void*)(base + kFa128BN * kFa128Row + u * 1024), 16, 0, 0);
        }
    };
    stage(0, 0);
    if (nkt > 1) stage(1, 1);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(bq[n][ks]), "+v"(bdo[n][ks]));
        asm volatile("" : "+v"(lq[n]), "+v"(dl[n]));
    }
    float4_t acc[8][2];                                  // dQ^T: [hd tile c][query tile n], rows = hd 16c + 4g + r
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[c][n] = float4_t{0.f, 0.f, 0.f, 0.f};
    const int qq = fr >> 2, pp = fr & 3;
    const int vsw = 2 * (4 * (g & 1) + qq);
    // K^T: the forward's V^T addressing on the K image.  tr_off[c] = row part + (((2 c + (pp >> 1)) ^ vsw) << 4) + 8 (pp & 1); vsw is
    // even and everything else of the offset lies below bit 5, so tr_off[c] = tr_off[0] ^ (c << 5): ONE register and an XOR in front
    // of each read instead of eight registers live across the loop -- round 2's kernel (256 VGPRs, 2 blocks per CU) spilled 7 dwords
    // to scratch and RELOADED two of them inside the key-tile loop (scratch loads count in vmcnt, next to the LDS-DMA ring's
    // counted waits).  The XOR is an asm statement so that hipcc does not hoist the eight values back out of the loop.
    const unsigned tr0 = (4 * g + qq) * kFa128Row + (((pp >> 1) ^ vsw) << 4) + 8 * (pp & 1);
    unsigned row_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) row_off[ks] = fr * kFa128Row + (((4 * ks + g) ^ (2 * (fr & 7))) << 4);
    const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

#define RPO_TR2H(OUT0, OUT1, ADDR)                                                                              \
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:4096" : "=&v"(OUT0), "=&v"(OUT1) : "v"(ADDR) : "memory")
#define RPO_TR2H_X(OUT0, OUT1, TB, C)                                                                           \
    do {                                                                                                        \
        unsigned a_;                                                                                            \
        asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a_) : "n"((C) << 5), "v"(tr0));                               \
        RPO_TR2H(OUT0, OUT1, (TB) + a_);                                                                         \
    } while (0)
    int cur = 0;
#ifndef RPO_F128_EXP
#define RPO_F128_EXP 0   // timing-only ablations (tools/exp/build_variant.sh NAME -DRPO_F128_EXP=mask; results are WRONG by design):
#endif                   // 1 no softmax arithmetic, 2 no LDS-DMA staging inside the loop, 4 no barrier, 8 no waits on the DMA ring
    for (int kt = 0; kt < nkt; ++kt) {
#if !(RPO_F128_EXP & 8)
        if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#if !(RPO_F128_EXP & 4)
        __builtin_amdgcn_s_barrier();
#endif
#if !(RPO_F128_EXP & 2)
        if (kt + 2 < nkt) stage(kt + 2, cur == 0 ? 2 : cur - 1);
#endif
        const bool active = (kt * kFa128BN <= qw + 31) && (qw < len);
        if (active) {
            const char* Ks = smem + cur * kKvTile;
            const char* Vs = Ks + kFa128BN * kFa128Row;
            const unsigned tb = smem_base + cur * kKvTile;
            u32x2 x0, x1, x2, x3, x4, x5, x6, x7, y0, y1, y2, y3, y4, y5, y6, y7;
            RPO_TR2H(x0, y0, tb + tr0);
            RPO_TR2H_X(x1, y1, tb, 1);
            RPO_TR2H_X(x2, y2, tb, 2);
            RPO_TR2H_X(x3, y3, tb, 3);
            RPO_TR2H_X(x4, y4, tb, 4);
            RPO_TR2H_X(x5, y5, tb, 5);
            RPO_TR2H_X(x6, y6, tb, 6);
            RPO_TR2H_X(x7, y7, tb, 7);
            float4_t s[2][2], dp[2][2];                   // [key sub-tile m][query tile n], rows = keys 16m + 4g + r
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    s[m][n] = float4_t{lq[n], lq[n], lq[n], lq[n]};
                    dp[m][n] = float4_t{dl[n], dl[n], dl[n], dl[n]};
                }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const short8_t ak = *reinterpret_cast<const short8_t*>(Ks + row_off[ks] + m * 16 * kFa128Row);
                    const short8_t av = *reinterpret_cast<const short8_t*>(Vs + row_off[ks] + m * 16 * kFa128Row);
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        s[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ak, bq[n][ks], s[m][n], 0, 0, 0);
                        dp[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bdo[n][ks], dp[m][n], 0, 0, 0);
                    }
                }
            }
            const int kbase = kt * kFa128BN + 4 * g;
            const bool need_mask = (kt * kFa128BN + kFa128BN - 1 > qw) || (kt * kFa128BN + kFa128BN > len);
            if (need_mask) {
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int qi = qw + 16 * n + fr;
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float pv = __builtin_amdgcn_exp2f(s[m][n][r] * scale_log2e);
                            const int key = kbase + 16 * m + r;
                            pv = (key > qi || key >= len || qi >= len) ? 0.f : pv;
                            s[m][n][r] = pv * dp[m][n][r];
                        }
                }
            } else {
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) s[m][n][r] = __builtin_amdgcn_exp2f(s[m][n][r] * scale_log2e) * dp[m][n][r];
            }
            short8_t dsf[2];                              // k-slots = keys {4g + j, 16 + 4g + (j - 4)} of the tile
#pragma unroll
            for (int n = 0; n < 2; ++n) dsf[n] = pack_frag(s[0][n], s[1][n]);
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(y0),
                           "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7)
                         :
                         : "memory");
            const short8_t ktf[8] = {join_tr(x0, y0), join_tr(x1, y1), join_tr(x2, y2), join_tr(x3, y3),
                                     join_tr(x4, y4), join_tr(x5, y5), join_tr(x6, y6), join_tr(x7, y7)};
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[c][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf[c], dsf[n], acc[c][n], 0, 0, 0);
        }
        cur = cur == 2 ? 0 : cur + 1;
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int qi = qw + 16 * n + fr;
        if (qi >= len) continue;
        bf16_t* row = dq + (t0 + qi) * sdq + h * kFa128HD;
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c][n] *= scale;
        if (rcos) {                                         // rows 16c + 4g + r pair with 16 (c + 4) + 4g + r
            const int64_t tr = ((t0 + qi) % rperiod) * (kFa128HD / 2) + 4 * g;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                inv_rope4(acc[c][n], acc[c + 4][n], *reinterpret_cast<const float4_t*>(rcos + tr + 16 * c),
                          *reinterpret_cast<const float4_t*>(rsin + tr + 16 * c));
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            uint2 w;
            w.x = pack_bf16(acc[c][n][0], acc[c][n][1]);
            w.y = pack_bf16(acc[c][n][2], acc[c][n][3]);
            *reinterpret_cast<uint2*>(row + 16 * c + 4 * g) = w;
        }
    }
}

#undef RPO_TR2H_X
#undef RPO_TR2H
constexpr int kD128Sl = 32;                                        // query rows per slice
constexpr int kD128Img = 2 * kD128Sl * kFa128Row + 256;            // Q | dO | 32 x -lse / scale | 32 x -delta = 16640 B
constexpr int kD128Keys = 128;                                     // keys per block (4 waves x 32)
#ifndef RPO_D128_AHEAD
#define RPO_D128_AHEAD 3
#endif
// slice images: ring of 4, 3 slices ahead of the one consumed.  Deeper does not pay (measured in one process, backward entry
// point on cfg 5's shape: 3 / 4 / 6 / 7 ahead in a ring of 8 = 7.727 / 7.725 / 7.724 / 7.723 ms): the ring hides the latency
// already; what the Q / dO stream still costs (1.0 ms of the kernel's 4.3: 0.2 for issuing the DMA instructions, 0.8 that
// vanish when every stage re-reads the SAME slice) is throughput of the L2 -> L1 -> LDS path, which no queue depth buys back.
constexpr int kD128Ring = 4, kD128Ahead = RPO_D128_AHEAD;
constexpr int kD128Lds = kD128Ring * kD128Img;                     // 66560 B (dynamic)
static_assert(kD128Ahead >= 2 && kD128Ahead < kD128Ring, "the image of stage it + Ahead must not be one of the two being read");

#include "attention_dkdv128_gen.inc"

// Round 3: the slice body is ONE hand-placed instruction stream (tools/gen/gen_dkdv128_body.py -> attention_dkdv128_gen.inc), as
// at head_dim 64: M1 (S', dP' chains, key-tile major) -> the exp2 / dS arithmetic and bf16 packing of key tile n in the issue
// gaps of the MFMAs that follow its chains -> M2 (dV^T, dK^T) with the LDS reads of the NEXT slice's row operands under it.
// Register split: v[0:63] hipcc (addresses, loop state), v[64:255] the stream's; a[0:127] dV^T / dK^T accumulators, a[128:191]
// the wave's K / V fragments (they were 64 VGPRs in round 2's hipcc-scheduled kernel).  Masked slices (diagonal, sequence end)
// run the same steps as separate statements around hipcc's code for the masked arithmetic.  hipcc must not spill (it does not
// know a[0:191] to be occupied): its resource line must show 0 scratch and no v_accvgpr_* outside ASMSTART / ASMEND.
// DOWN (sweep_down of the C entry point, power-of-two group): slice-major from the LAST query slice downwards, the group's q
// heads innermost -- every key block of a (sequence, kv head) then starts at the same slice and walks the same (slice, head)
// sequence, so blocks launched side by side on one XCD (the group-ordered work list) read each Q / dO slice at about the same
// time: once from HBM, the rest from L2.
template <bool DOWN>
__global__ __launch_bounds__(256, 1) void fa_bwd_dkdv128_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const bf16_t* __restrict__ dout, int64_t sq, int64_t sk, int64_t sv, int64_t sdo, const int* __restrict__ cu,
    const int* __restrict__ ktiles, int nh, int nkv, float scale_log2e, float scale, const float* __restrict__ nl,
    const float* __restrict__ nd, int64_t T, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, int64_t sdk,
    int64_t sdv, int n_ktiles, int gshift, const float* __restrict__ rcos, const float* __restrict__ rsin, int64_t rperiod) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, fr = lane & 15;
    const int per = (n_ktiles + 7) >> 3;
    const int entry = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (entry >= n_ktiles) return;
    const int seq = ktiles[3 * entry], hk = ktiles[3 * entry + 1], kb0 = ktiles[3 * entry + 2];
    const int k0 = kb0 + 32 * wave;                        // this wave's 32 keys
    const int group = nh / nkv;
    const int64_t t0 = cu[seq];
    const int len = cu[seq + 1] - (int)t0;
    if (kb0 >= len) return;                                // padding entry of the table
    const int qt0 = kb0;                                   // first query row that can see the block's keys (multiple of 32)
    const int nsl = (len - qt0 + kD128Sl - 1) / kD128Sl;
    const int niter = nsl * group;

    {
        short8_t bk[2][4], bv[2][4];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int key = k0 + 16 * n + fr;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const short8_t z = short8_t{0, 0, 0, 0, 0, 0, 0, 0};
                bk[n][ks] = key < len ? *reinterpret_cast<const short8_t*>(k + (t0 + key) * sk + hk * kFa128HD + 32 * ks + 8 * g) : z;
                bv[n][ks] = key < len ? *reinterpret_cast<const short8_t*>(v + (t0 + key) * sv + hk * kFa128HD + 32 * ks + 8 * g) : z;
            }
        }
        // the K / V fragments move to a[128:191] (this also places hipcc's wait for their loads in front of the DMA ring)
        RPO_D128_KV_TO_ACC(bk, bv);
    }
    // staging: wave w moves rows 8w .. 8w + 7 of the Q and of the dO slice (two 1-KiB pieces of 4 rows each); wave 0 also the 64
    // row constants (lanes 0..31: -lse / scale, lanes 32..63: -delta)
    const int srow = lane >> 4;
    const unsigned sqb = (unsigned)sq * 2u, sdob = (unsigned)sdo * 2u;
    const float* rc_src = lane < 32 ? nl : nd;
    auto stage = [&](int hq, int qb, int buf) {
        char* base = smem + buf * kD128Img;
        const char* qsrc = reinterpret_cast<const char*>(q + t0 * sq + hq * kFa128HD);
        const char* dsrc = reinterpret_cast<const char*>(dout + t0 * sdo + hq * kFa128HD);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int u = 2 * wave + i;
            const int trow = 4 * u + srow;
            const unsigned lchunk = (unsigned)((lane & 15) ^ (2 * (trow & 7)));
            const unsigned row = (unsigned)min(qb + trow, len - 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qsrc + (row * sqb + lchunk * 16)),
                                             (__attribute__((address_space(3))) void*)(base + u * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dsrc + (row * sdob + lchunk * 16)),
                                             (__attribute__((address_space(3))) void*)(base + kD128Sl * kFa128Row + u * 1024), 16, 0, 0);
        }
        if (wave == 0) {
            const float* src = rc_src + (int64_t)hq * T + t0 + (unsigned)min(qb + (lane & 31), len - 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(base + 2 * kD128Sl * kFa128Row), 4, 0, 0);
        }
    };
    int st_h = hk * group, st_s = 0, st_buf = 0, st_n = 0;     // (q head, slice) of the next stage, head-major, slices ascending
    auto stage_next = [&]() {
#if defined(RPO_D128_EXP) && RPO_D128_EXP == 3                    // (timing experiment: every stage re-reads the SAME slice: L2 hits)
        stage(hk * group, qt0, st_buf);
#else
        if constexpr (DOWN) {
            stage(hk * group + (st_n & (group - 1)), qt0 + (nsl - 1 - (st_n >> gshift)) * kD128Sl, st_buf);
        } else {
            stage(st_h, qt0 + st_s * kD128Sl, st_buf);
            if (++st_s == nsl) { st_s = 0; ++st_h; }
        }
#endif
        ++st_n;
        st_buf = (st_buf + 1) & (kD128Ring - 1);
    };
#pragma unroll 1
    for (int i = 0; i < kD128Ahead && i < niter; ++i) stage_next();
    // dV^T / dK^T accumulators [hd tile c][key tile n] (rows = hd 16c + 4g + r, col = key 16n + fr): LITERAL accumulator
    // registers for the whole kernel, dV^T[c][n] = a[8c + 4n : + 3], dK^T[c][n] = a[64 + 8c + 4n : + 3]
    RPO_D128_ZERO_ACC();
    const int qq = fr >> 2, pp = fr & 3;
    const int vsw = 2 * (4 * (g & 1) + qq);
    unsigned tr_off[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
        tr_off[c] = (4 * g + qq) * kFa128Row + (((2 * c + (pp >> 1)) ^ vsw) << 4) + 8 * (pp & 1);
    unsigned row_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) row_off[ks] = fr * kFa128Row + (((4 * ks + g) ^ (2 * (fr & 7))) << 4);
    const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned rc_off = 2 * kD128Sl * kFa128Row + 16 * g;   // row constants of the lane's 4 rows (tile m at + 64, -delta at + 128)

#define RPO_TR2H(OUT0, OUT1, ADDR)                                                                              \
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:4096" : "=&v"(OUT0), "=&v"(OUT1) : "v"(ADDR) : "memory")
    // s_waitcnt takes an immediate: one statement per (number of later stages in flight, DMA instructions per stage)
#define RPO_D128_WAIT(N)                                                                                            \
    do {                                                                                                            \
        if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * (N)) : "memory");                               \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (N)) : "memory");                                         \
    } while (0)
    auto wait_landed = [&](int upto) {                        // every stage <= upto of THIS wave has landed
        const int later = st_n - 1 - upto;                   // stages issued after it: 0 .. Ahead - 1
        if (later >= 6) RPO_D128_WAIT(6);
        else if (later == 5) RPO_D128_WAIT(5);
        else if (later == 4) RPO_D128_WAIT(4);
        else if (later == 3) RPO_D128_WAIT(3);
        else if (later == 2) RPO_D128_WAIT(2);
        else if (later == 1) RPO_D128_WAIT(1);
        else RPO_D128_WAIT(0);
    };
    bool hot = false;      // v[96:175] hold the row fragments / row constants of slice `it` (prefetched by the previous body)
    int cur = 0, sl = 0;
    for (int it = 0; it < niter; ++it) {
        if constexpr (DOWN) sl = nsl - 1 - (it >> gshift);
        // slices <= it + 1 have landed (the body prefetches from the next image).  Steady state (slices still being staged):
        // stages it + 2 .. it + Ahead - 1 are in flight
        if (st_n < niter) {
            RPO_D128_WAIT(kD128Ahead - 2);
        } else {
            wait_landed(it + 1 < niter ? it + 1 : it);
        }
#if !defined(RPO_D128_EXP) || RPO_D128_EXP != 1               // (timing experiments: tools/exp/build_d128_variant.sh)
        __builtin_amdgcn_s_barrier();
#endif
#if !defined(RPO_D128_EXP) || RPO_D128_EXP != 2
        if (st_n < niter) stage_next();                      // slice it + Ahead -> an image whose slice (<= it - 2) is done
#else
        if (st_n < niter) ++st_n;
#endif
        const int qb = qt0 + sl * kD128Sl;
        const bool active = (qb + kD128Sl - 1 >= k0) && (k0 < len);
        // three kinds of active slices: plain (every key of the wave visible to every query of the slice), the DIAGONAL one (the
        // causal boundary crosses it: one per wave and q head, at a different slice for each wave -- the same hand-placed stream
        // with the mask folded into the S' chains' initial accumulators, so that the block does not wait at the next barrier for
        // one wave on a slow path), and the tail (sequence end inside the slice or inside the wave's keys: hipcc's select-based
        // code below; the last slice of a head for all four waves at once)
        const bool tail = (qb + kD128Sl > len) || (k0 + 32 > len);
        const bool diag = qb < k0 + 31;
        const unsigned img = smem_base + cur * kD128Img;
        const unsigned next_img = smem_base + ((cur + 1) & (kD128Ring - 1)) * kD128Img;
        if (active && !tail) {
            if (!diag) {
                if (hot)
                    RPO_D128_SLICE_BODY_HOT(img + row_off[0], img + row_off[1], img + row_off[2], img + row_off[3], img + rc_off,
                                            img + tr_off[0], img + tr_off[1], img + tr_off[2], img + tr_off[3], img + tr_off[4],
                                            img + tr_off[5], img + tr_off[6], img + tr_off[7], scale_log2e, next_img + row_off[0],
                                            next_img + row_off[1], next_img + row_off[2], next_img + row_off[3], next_img + rc_off);
                else
                    RPO_D128_SLICE_BODY_LOAD(img + row_off[0], img + row_off[1], img + row_off[2], img + row_off[3], img + rc_off,
                                             img + tr_off[0], img + tr_off[1], img + tr_off[2], img + tr_off[3], img + tr_off[4],
                                             img + tr_off[5], img + tr_off[6], img + tr_off[7], scale_log2e, next_img + row_off[0],
                                             next_img + row_off[1], next_img + row_off[2], next_img + row_off[3], next_img + rc_off);
            } else {
                const int dlane = (k0 + fr) - (qb + 4 * g);      // the lane's key (tile 0) minus its first query row
                if (hot)
                    RPO_D128_DIAG_BODY_HOT(img + row_off[0], img + row_off[1], img + row_off[2], img + row_off[3], img + rc_off,
                                           img + tr_off[0], img + tr_off[1], img + tr_off[2], img + tr_off[3], img + tr_off[4],
                                           img + tr_off[5], img + tr_off[6], img + tr_off[7], scale_log2e, next_img + row_off[0],
                                           next_img + row_off[1], next_img + row_off[2], next_img + row_off[3], next_img + rc_off,
                                           dlane);
                else
                    RPO_D128_DIAG_BODY_LOAD(img + row_off[0], img + row_off[1], img + row_off[2], img + row_off[3], img + rc_off,
                                            img + tr_off[0], img + tr_off[1], img + tr_off[2], img + tr_off[3], img + tr_off[4],
                                            img + tr_off[5], img + tr_off[6], img + tr_off[7], scale_log2e, next_img + row_off[0],
                                            next_img + row_off[1], next_img + row_off[2], next_img + row_off[3], next_img + rc_off,
                                            dlane);
            }
            hot = true;
        } else if (active) {
            hot = false;
            const char* Qs = smem + cur * kD128Img;
            const char* Ds = Qs + kD128Sl * kFa128Row;
            const float* Ls = reinterpret_cast<const float*>(Qs + 2 * kD128Sl * kFa128Row);
            const unsigned tb = img;
            float4_t s[2][2], dp[2][2];                       // [query tile m][key tile n]; rows = queries 16m + 4g + r, col = key 16n + fr
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const float4_t lr = *reinterpret_cast<const float4_t*>(Ls + 16 * m + 4 * g);
                const float4_t dr = *reinterpret_cast<const float4_t*>(Ls + kD128Sl + 16 * m + 4 * g);
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    s[m][n] = lr;
                    dp[m][n] = dr;
                }
            }
#define RPO_D128_M1_STEP(KS)                                                                                        \
            {                                                                                                       \
                const short8_t aq0_ = *reinterpret_cast<const short8_t*>(Qs + row_off[KS]);                          \
                const short8_t ad0_ = *reinterpret_cast<const short8_t*>(Ds + row_off[KS]);                          \
                const short8_t aq1_ = *reinterpret_cast<const short8_t*>(Qs + row_off[KS] + 16 * kFa128Row);         \
                const short8_t ad1_ = *reinterpret_cast<const short8_t*>(Ds + row_off[KS] + 16 * kFa128Row);         \
                RPO_D128_M1_KS##KS(0, aq0_, ad0_);                                                                   \
                RPO_D128_M1_KS##KS(1, aq1_, ad1_);                                                                   \
            }
            RPO_D128_M1_STEP(0)
            RPO_D128_M1_STEP(1)
            RPO_D128_M1_STEP(2)
            RPO_D128_M1_STEP(3)
#undef RPO_D128_M1_STEP
            // 8-pass MFMA result -> VALU read: 11 wait states, which hipcc cannot count for asm; the operands pin the statement
            // between the chains and their first use
            asm volatile("s_nop 15"
                         : "+v"(s[0][0]), "+v"(s[0][1]), "+v"(s[1][0]), "+v"(s[1][1]), "+v"(dp[0][0]), "+v"(dp[0][1]),
                           "+v"(dp[1][0]), "+v"(dp[1][1]));
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int qr0 = qb + 16 * m + 4 * g;
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int key = k0 + 16 * n + fr;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float pv = __builtin_amdgcn_exp2f(s[m][n][r] * scale_log2e);
                        pv = (key > qr0 + r || key >= len || qr0 + r >= len) ? 0.f : pv;
                        s[m][n][r] = pv;
                        dp[m][n][r] = pv * dp[m][n][r];
                    }
                }
            }
            short8_t pf[2], dsf[2];                         // k-slots = queries {4g + j, 16 + 4g + (j - 4)} of the slice
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                pf[n] = pack_frag(s[0][n], s[1][n]);
                dsf[n] = pack_frag(dp[0][n], dp[1][n]);
            }
            // M2 one hd tile at a time: dO^T / Q^T fragments by transposed reads (d = queries 4g .. 4g + 3 of rows 0-15, dd = of 16-31)
#define RPO_D128_M2_STEP(C)                                                                                         \
            {                                                                                                       \
                u32x2 d_, dd_, e_, ee_;                                                                              \
                RPO_TR2H(d_, dd_, tb + kD128Sl * kFa128Row + tr_off[C]);                                             \
                RPO_TR2H(e_, ee_, tb + tr_off[C]);                                                                   \
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d_), "+v"(dd_), "+v"(e_), "+v"(ee_) : : "memory");        \
                const short8_t atd_ = join_tr(d_, dd_), atq_ = join_tr(e_, ee_);                                     \
                RPO_D128_M2_C##C(atd_, atq_, pf, dsf);                                                               \
            }
            RPO_D128_M2_STEP(0)
            RPO_D128_M2_STEP(1)
            RPO_D128_M2_STEP(2)
            RPO_D128_M2_STEP(3)
            RPO_D128_M2_STEP(4)
            RPO_D128_M2_STEP(5)
            RPO_D128_M2_STEP(6)
            RPO_D128_M2_STEP(7)
#undef RPO_D128_M2_STEP
        } else {
            hot = false;
        }
        cur = (cur + 1) & (kD128Ring - 1);
        if constexpr (!DOWN) sl = sl + 1 == nsl ? 0 : sl + 1;
    }
#undef RPO_TR2H
    // the last MFMAs have to leave the pipe before their accumulators are read (hipcc cannot see the dependency)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#define RPO_D128_RD(B0, B1, B2, B3, X)                                                                              \
    do {                                                                                                            \
        float t0_, t1_, t2_, t3_;                                                                                   \
        asm volatile("v_accvgpr_read_b32 %0, a" #B0 "\n\tv_accvgpr_read_b32 %1, a" #B1 "\n\t"                        \
                     "v_accvgpr_read_b32 %2, a" #B2 "\n\tv_accvgpr_read_b32 %3, a" #B3                              \
                     : "=v"(t0_), "=v"(t1_), "=v"(t2_), "=v"(t3_));                                                   \
        X = float4_t{t0_, t1_, t2_, t3_};                                                                           \
    } while (0)
    auto store4 = [](bf16_t* dst, const float4_t& x) {
        uint2 w;
        w.x = pack_bf16(x[0], x[1]);
        w.y = pack_bf16(x[2], x[3]);
        *reinterpret_cast<uint2*>(dst) = w;
    };
    {
        float4_t kq[8], vq[8];                          // dK^T / dV^T quads of key tile 0: rows = hd 16c + 4g + r
        RPO_D128_RD(0, 1, 2, 3, vq[0]);
        RPO_D128_RD(64, 65, 66, 67, kq[0]);
        RPO_D128_RD(8, 9, 10, 11, vq[1]);
        RPO_D128_RD(72, 73, 74, 75, kq[1]);
        RPO_D128_RD(16, 17, 18, 19, vq[2]);
        RPO_D128_RD(80, 81, 82, 83, kq[2]);
        RPO_D128_RD(24, 25, 26, 27, vq[3]);
        RPO_D128_RD(88, 89, 90, 91, kq[3]);
        RPO_D128_RD(32, 33, 34, 35, vq[4]);
        RPO_D128_RD(96, 97, 98, 99, kq[4]);
        RPO_D128_RD(40, 41, 42, 43, vq[5]);
        RPO_D128_RD(104, 105, 106, 107, kq[5]);
        RPO_D128_RD(48, 49, 50, 51, vq[6]);
        RPO_D128_RD(112, 113, 114, 115, kq[6]);
        RPO_D128_RD(56, 57, 58, 59, vq[7]);
        RPO_D128_RD(120, 121, 122, 123, kq[7]);
        const int key = k0 + 0 + fr;
        if (key < len) {
            bf16_t* krow = dk + (t0 + key) * sdk + hk * kFa128HD + 4 * g;
            bf16_t* vrow = dv + (t0 + key) * sdv + hk * kFa128HD + 4 * g;
#pragma unroll
            for (int c = 0; c < 8; ++c) kq[c] *= scale;                          // dK = scale * dS^T Q
            if (rcos) {                                                          // ... w.r.t. the pre-rotary k (inv_rope4)
                const int64_t tr = ((t0 + key) % rperiod) * (kFa128HD / 2) + 4 * g;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    inv_rope4(kq[c], kq[c + 4], *reinterpret_cast<const float4_t*>(rcos + tr + 16 * c),
                              *reinterpret_cast<const float4_t*>(rsin + tr + 16 * c));
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                store4(krow + 16 * c, kq[c]);
                store4(vrow + 16 * c, vq[c]);
            }
        }
    }
    {
        float4_t kq[8], vq[8];                          // dK^T / dV^T quads of key tile 1: rows = hd 16c + 4g + r
        RPO_D128_RD(4, 5, 6, 7, vq[0]);
        RPO_D128_RD(68, 69, 70, 71, kq[0]);
        RPO_D128_RD(12, 13, 14, 15, vq[1]);
        RPO_D128_RD(76, 77, 78, 79, kq[1]);
        RPO_D128_RD(20, 21, 22, 23, vq[2]);
        RPO_D128_RD(84, 85, 86, 87, kq[2]);
        RPO_D128_RD(28, 29, 30, 31, vq[3]);
        RPO_D128_RD(92, 93, 94, 95, kq[3]);
        RPO_D128_RD(36, 37, 38, 39, vq[4]);
        RPO_D128_RD(100, 101, 102, 103, kq[4]);
        RPO_D128_RD(44, 45, 46, 47, vq[5]);
        RPO_D128_RD(108, 109, 110, 111, kq[5]);
        RPO_D128_RD(52, 53, 54, 55, vq[6]);
        RPO_D128_RD(116, 117, 118, 119, kq[6]);
        RPO_D128_RD(60, 61, 62, 63, vq[7]);
        RPO_D128_RD(124, 125, 126, 127, kq[7]);
        const int key = k0 + 16 + fr;
        if (key < len) {
            bf16_t* krow = dk + (t0 + key) * sdk + hk * kFa128HD + 4 * g;
            bf16_t* vrow = dv + (t0 + key) * sdv + hk * kFa128HD + 4 * g;
#pragma unroll
            for (int c = 0; c < 8; ++c) kq[c] *= scale;                          // dK = scale * dS^T Q
            if (rcos) {                                                          // ... w.r.t. the pre-rotary k (inv_rope4)
                const int64_t tr = ((t0 + key) % rperiod) * (kFa128HD / 2) + 4 * g;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    inv_rope4(kq[c], kq[c + 4], *reinterpret_cast<const float4_t*>(rcos + tr + 16 * c),
                              *reinterpret_cast<const float4_t*>(rsin + tr + 16 * c));
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                store4(krow + 16 * c, kq[c]);
                store4(vrow + 16 * c, vq[c]);
            }
        }
    }
#undef RPO_D128_RD
}

}  // namespace

#ifdef RPO_FA_LADDER
// host copy of region `kernel` (0 forward, 1 dQ, 2 dK/dV) of the per-block records: out = nblocks x 8 uint64; reset: zero the region
extern "C" int rpo_debug_fa_ladder(unsigned long long* out, int kernel, int nblocks, int reset) {
    if (kernel < 0 || kernel > 2 || nblocks <= 0 || nblocks > kLadMax) return -1;
    const size_t off = (size_t)kernel * kLadMax * 8 * sizeof(unsigned long long), n = (size_t)nblocks * 8 * sizeof(unsigned long long);
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fa_ladder), n, off) != hipSuccess) return -1;
    if (reset) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_fa_ladder)) != hipSuccess) return -1;
        if (hipMemset((char*)p + off, 0, n) != hipSuccess) return -1;
    }
    return 0;
}
#endif

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

extern "C" int rpo_build_flags(void) {
#ifdef RPO_ONEWAVE64
    return RPO_BUILD_ONEWAVE64;
#else
    return 0;
#endif
}

extern "C" int rpo_flash_attn_fwd(const void* q, const void* k, const void* v, int64_t q_stride, int64_t k_stride,
                                  int64_t v_stride, const int* cu_seqlens, const int* tiles, int64_t ntiles,
                                  int64_t tile_cols, int64_t total_tokens, int64_t num_heads, int64_t num_kv_heads, int64_t head_dim,
                                  float scale, void* out, int64_t out_stride, float* lse, int64_t lse_max_len,
                                  const float* rope_cos, const float* rope_sin, int64_t rope_period, int64_t q_block,
                                  rpo_stream_t stream) {
    if (!q || !k || !v || !cu_seqlens || !tiles || !out || ntiles <= 0 || total_tokens <= 0)      // lse may be NULL (forward only)
        return RPO_ERR_INVALID_ARG;
    // q_block: query rows per work-list entry.  128 (or 0): every kernel.  64: head_dim 128 only -- an entry is 64 queries x the FOUR
    // consecutive q heads that start at the entry's head (format 2: head = 4 x the launch's y index), all of one kv head:
    // fa_fwd128w_kernel, one wave per SIMD
    if (q_block == 0) q_block = 128;
    if (q_block != 128 && !(q_block == 64 && (head_dim == kFa128HD || head_dim == kFaHD) && num_kv_heads > 0 &&
                            num_heads % num_kv_heads == 0 && (num_heads / num_kv_heads) % 4 == 0))
        return RPO_ERR_UNSUPPORTED;
    // rope_cos / rope_sin (both or neither): q arrives UN-rotated and is rotated IN PLACE by the block that owns it (k must
    // arrive rotated: every query block reads it)
    if ((rope_cos == nullptr) != (rope_sin == nullptr) || (rope_cos && rope_period <= 0)) return RPO_ERR_INVALID_ARG;
    if (rope_cos && (!rpo_aligned16(rope_cos) || !rpo_aligned16(rope_sin))) return RPO_ERR_UNSUPPORTED;
    if (!(tile_cols == 2 || (tile_cols == 3 && ntiles % 8 == 0))) return RPO_ERR_UNSUPPORTED;
    if ((head_dim != kFaHD && head_dim != kFa128HD) || num_heads <= 0 || num_kv_heads <= 0 || num_heads % num_kv_heads != 0 ||
        num_heads > 65535)
        return RPO_ERR_UNSUPPORTED;
    if (q_stride % 8 || k_stride % 8 || v_stride % 8 || out_stride % 4 || !rpo_aligned16(q) || !rpo_aligned16(k) ||
        !rpo_aligned16(v) || (reinterpret_cast<uintptr_t>(out) & 7))
        return RPO_ERR_UNSUPPORTED;
    if (q_block == 64 && (out_stride % 8 || !rpo_aligned16(out))) return RPO_ERR_UNSUPPORTED;      // whole-row stores, 16 bytes per lane
    hipStream_t st = (hipStream_t)stream;
    const float log2e = 1.4426950408889634f;
    dim3 grid((unsigned)ntiles, tile_cols == 3 ? 1u : (unsigned)num_heads);
#ifndef RPO_F128_WQ
#define RPO_F128_WQ 4
#endif
#ifndef RPO_F128_HEADS
#define RPO_F128_HEADS 1
#endif
#ifndef RPO_F128_SUB
#define RPO_F128_SUB 1
#endif
#ifndef RPO_F128_KF
#define RPO_F128_KF false
#endif
    if (head_dim == kFa128HD && tile_cols != 3) grid.y = (unsigned)(num_heads / RPO_F128_HEADS);
#define RPO_F128_KERNEL fa_fwd128_kernel<RPO_F128_WQ, RPO_F128_HEADS, RPO_F128_SUB, RPO_F128_KF>
#ifndef RPO_ONEWAVE64
    if (q_block == 64 && head_dim == kFaHD) return RPO_ERR_UNSUPPORTED;       // not in this build (make ONEWAVE64=1; rpo_build_flags())
#else
    if (q_block == 64 && head_dim == kFaHD) {
        if (tile_cols != 3) grid.y = (unsigned)(num_heads / 4);
        RPO_LAUNCH(fa_fwd64w_kernel, grid, dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, q_stride,
                   k_stride, v_stride, cu_seqlens, tiles, (int)tile_cols, (int)num_heads, (int)num_kv_heads, scale * log2e, scale,
                   (bf16_t*)out, out_stride, lse, lse_max_len > 0 ? num_heads * lse_max_len : 0,
                   lse_max_len > 0 ? lse_max_len : total_tokens, lse_max_len > 0 ? 0 : 1, rope_cos, rope_sin, rope_period,
                   (bf16_t*)const_cast<void*>(q));
        return rpo_launch_status();
    }
#endif
    if (q_block == 64) {
        if (tile_cols != 3) grid.y = (unsigned)(num_heads / 4);
        RPO_LAUNCH(fa_fwd128w_kernel, grid, dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, q_stride,
                   k_stride, v_stride, cu_seqlens, tiles, (int)tile_cols, (int)num_heads, (int)num_kv_heads, scale * log2e, scale,
                   (bf16_t*)out, out_stride, lse, lse_max_len > 0 ? num_heads * lse_max_len : 0,
                   lse_max_len > 0 ? lse_max_len : total_tokens, lse_max_len > 0 ? 0 : 1, rope_cos, rope_sin, rope_period,
                   (bf16_t*)const_cast<void*>(q));
        return rpo_launch_status();
    }
    if (head_dim == kFa128HD)
        RPO_LAUNCH((RPO_F128_KERNEL), grid, dim3(64 * RPO_F128_WQ * RPO_F128_HEADS), 0, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, q_stride,
                   k_stride, v_stride, cu_seqlens, tiles, (int)tile_cols, (int)num_heads, (int)num_kv_heads, scale * log2e, scale,
                   (bf16_t*)out, out_stride, lse, lse_max_len > 0 ? num_heads * lse_max_len : 0,
                   lse_max_len > 0 ? lse_max_len : total_tokens, lse_max_len > 0 ? 0 : 1, rope_cos, rope_sin, rope_period,
                   (bf16_t*)const_cast<void*>(q));
    else
        RPO_LAUNCH(fa_fwd_kernel, grid, dim3(kFaThreads), 0, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, q_stride,
                   k_stride, v_stride, cu_seqlens, tiles, (int)tile_cols, (int)num_heads, (int)num_kv_heads, scale * log2e, scale,
                   (bf16_t*)out, out_stride, lse, lse_max_len > 0 ? num_heads * lse_max_len : 0,
                   lse_max_len > 0 ? lse_max_len : total_tokens, lse_max_len > 0 ? 0 : 1, rope_cos, rope_sin, rope_period,
                   (bf16_t*)const_cast<void*>(q));
    return rpo_launch_status();
}

extern "C" int rpo_flash_attn_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout,
                                  int64_t q_stride, int64_t k_stride, int64_t v_stride, int64_t out_stride,
                                  int64_t dout_stride, const int* cu_seqlens, const int* q_tiles, int64_t n_q_tiles,
                                  int64_t q_tile_cols, const int* k_tiles, int64_t n_k_tiles, int64_t key_block,
                                  int64_t sweep_down, int64_t total_tokens, int64_t num_heads, int64_t num_kv_heads, int64_t head_dim, float scale, const float* lse, float* delta,
                                  void* dq, void* dk, void* dv, int64_t dq_stride, int64_t dk_stride, int64_t dv_stride,
                                  const float* rope_cos, const float* rope_sin, int64_t rope_period, int64_t q_block,
                                  rpo_stream_t stream) {
    if (!q || !k || !v || !out || !dout || !cu_seqlens || !q_tiles || !k_tiles || !lse || !delta || !dq || !dk || !dv)
        return RPO_ERR_INVALID_ARG;
    // q_block: query rows per entry of q_tiles.  128 (or 0): fa_bwd_dq_kernel / fa_bwd_dq128_kernel.  64: head_dim 64 only -- an entry
    // is 64 queries x the four consecutive q heads that start at the entry's head (rpo_flash_attn_fwd's q_block = 64 format):
    // fa_bwd_dq64w_kernel, one wave per SIMD; dq must be 16-byte aligned with dq_stride % 8 == 0 there
    if (q_block == 0) q_block = 128;
    if (q_block != 128 && !(q_block == 64 && head_dim == kFaHD && num_kv_heads > 0 && num_heads % num_kv_heads == 0 &&
                            (num_heads / num_kv_heads) % 4 == 0 && dq_stride % 8 == 0 && rpo_aligned16(dq)))
        return RPO_ERR_UNSUPPORTED;
#ifndef RPO_ONEWAVE64
    if (q_block == 64) return RPO_ERR_UNSUPPORTED;                             // not in this build (make ONEWAVE64=1)
#endif
    // rope_cos / rope_sin (both or neither): dq / dk leave as gradients w.r.t. the PRE-rotary q / k (inverse rotation in the
    // epilogues); not offered by the 8-wave dK/dV kernel (key_block 64), whose epilogue splits a row's halves over two passes
    if ((rope_cos == nullptr) != (rope_sin == nullptr) || (rope_cos && rope_period <= 0)) return RPO_ERR_INVALID_ARG;
    if (rope_cos && (key_block == 64 || !rpo_aligned16(rope_cos) || !rpo_aligned16(rope_sin))) return RPO_ERR_UNSUPPORTED;
    if (n_q_tiles <= 0 || n_k_tiles <= 0 || total_tokens <= 0) return RPO_ERR_INVALID_ARG;
    // `key_block` states what the entries of `k_tiles` mean: blocks of 256 keys (one-wave-per-SIMD dK/dV kernel) or of 64
    // keys (the 8-wave kernel).  The caller that built the table says so; nothing is read from the environment.
    // head_dim 128: blocks of 128 keys (fa_bwd_dkdv128_kernel), nothing else.
    if (head_dim == kFa128HD ? key_block != kD128Keys : (key_block != 256 && key_block != 64)) return RPO_ERR_UNSUPPORTED;
    if (!(q_tile_cols == 2 || (q_tile_cols == 3 && n_q_tiles % 8 == 0))) return RPO_ERR_UNSUPPORTED;
    if ((head_dim != kFaHD && head_dim != kFa128HD) || num_heads <= 0 || num_kv_heads <= 0 || num_heads % num_kv_heads != 0 ||
        num_heads > 65535)
        return RPO_ERR_UNSUPPORTED;
    if (q_stride % 8 || k_stride % 8 || v_stride % 8 || out_stride % 8 || dout_stride % 8 || dq_stride % 4 ||
        dk_stride % 4 || dv_stride % 4 || !rpo_aligned16(q) || !rpo_aligned16(k) || !rpo_aligned16(v) ||
        !rpo_aligned16(out) || !rpo_aligned16(dout))
        return RPO_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const float log2e = 1.4426950408889634f;
    float* nd = delta;                                       // scratch [2][num_heads][T]: -delta | -lse / scale
    float* nl = delta + num_heads * total_tokens;
    // (round 1 launched fa_delta_kernel here; the dQ kernel now computes and writes both row constants itself)
    if (head_dim == kFa128HD) {
        RPO_LAUNCH(fa_bwd_dq128_kernel, dim3((unsigned)n_q_tiles, q_tile_cols == 3 ? 1u : (unsigned)num_heads), dim3(kFaThreads),
                   0, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)dout, q_stride, k_stride, v_stride,
                   dout_stride, cu_seqlens, q_tiles, (int)q_tile_cols, (int)num_heads, (int)num_kv_heads, scale * log2e, scale,
                   lse, (const bf16_t*)out, out_stride, nl, nd, total_tokens, (bf16_t*)dq, dq_stride, rope_cos, rope_sin, rope_period);
        const int rc128 = rpo_launch_status();
        if (rc128 != RPO_OK) return rc128;
        static const bool attr128_set = [] {
            (void)hipFuncSetAttribute((const void*)fa_bwd_dkdv128_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kD128Lds);
            (void)hipFuncSetAttribute((const void*)fa_bwd_dkdv128_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kD128Lds);
            return true;
        }();
        (void)attr128_set;
        const int group128 = (int)(num_heads / num_kv_heads);
        int gshift128 = 0;
        while ((1 << gshift128) < group128) ++gshift128;
        const dim3 grid128((unsigned)(((n_k_tiles + 7) / 8) * 8));
        if (sweep_down && (1 << gshift128) == group128)
            RPO_LAUNCH(fa_bwd_dkdv128_kernel<true>, grid128, dim3(256), kD128Lds, st, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, (const bf16_t*)dout, q_stride, k_stride, v_stride, dout_stride, cu_seqlens, k_tiles,
                       (int)num_heads, (int)num_kv_heads, scale * log2e, scale, nl, nd, total_tokens, (bf16_t*)dk, (bf16_t*)dv,
                       dk_stride, dv_stride, (int)n_k_tiles, gshift128, rope_cos, rope_sin, rope_period);
        else
            RPO_LAUNCH(fa_bwd_dkdv128_kernel<false>, grid128, dim3(256), kD128Lds, st, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, (const bf16_t*)dout, q_stride, k_stride, v_stride, dout_stride, cu_seqlens, k_tiles,
                       (int)num_heads, (int)num_kv_heads, scale * log2e, scale, nl, nd, total_tokens, (bf16_t*)dk, (bf16_t*)dv,
                       dk_stride, dv_stride, (int)n_k_tiles, gshift128, rope_cos, rope_sin, rope_period);
        return rpo_launch_status();
    }
#ifdef RPO_ONEWAVE64
    if (q_block == 64)
        RPO_LAUNCH(fa_bwd_dq64w_kernel, dim3((unsigned)n_q_tiles, q_tile_cols == 3 ? 1u : (unsigned)(num_heads / 4)), dim3(256), 0,
                   st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)dout, q_stride, k_stride, v_stride,
                   dout_stride, cu_seqlens, q_tiles, (int)q_tile_cols, (int)num_heads, (int)num_kv_heads, scale * log2e, scale, lse,
                   (const bf16_t*)out, out_stride, nl, nd, total_tokens, (bf16_t*)dq, dq_stride, rope_cos, rope_sin, rope_period);
    else
#endif
        RPO_LAUNCH(fa_bwd_dq_kernel, dim3((unsigned)n_q_tiles, q_tile_cols == 3 ? 1u : (unsigned)num_heads), dim3(kFaThreads), 0,
                   st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)dout, q_stride, k_stride, v_stride,
                   dout_stride, cu_seqlens, q_tiles, (int)q_tile_cols, (int)num_heads, (int)num_kv_heads, scale * log2e, scale, lse,
                   (const bf16_t*)out, out_stride, nl, nd, total_tokens, (bf16_t*)dq, dq_stride, rope_cos, rope_sin, rope_period);
    int rc = rpo_launch_status();
    if (rc != RPO_OK) return rc;
    const unsigned dkdv_grid = (unsigned)(((n_k_tiles + 7) / 8) * 8);
    const bool use_v1 = key_block == 64;
#ifdef RPO_D4_EXP_DQ_ATOMICS
    {
        static float* exp_buf = nullptr;                       // experiment only: one buffer for the largest batch the tools use
        if (!exp_buf) {
            (void)hipMalloc((void**)&exp_buf, ((size_t)262144 + 64) * 32 * 64 * 4);
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_exp_dq32), &exp_buf, sizeof(exp_buf));
        }
        if ((size_t)total_tokens > 262144 || num_heads > 32) return RPO_ERR_UNSUPPORTED;
    }
#endif
    static const bool attr_set = [] {
        (void)hipFuncSetAttribute((const void*)fa_bwd_dkdv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kDmaLds);
        (void)hipFuncSetAttribute((const void*)fa_bwd_dkdv4_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kDkdv4Lds);
        (void)hipFuncSetAttribute((const void*)fa_bwd_dkdv4_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kDkdv4Lds);
        return true;
    }();
    (void)attr_set;
    if (use_v1) {
        RPO_LAUNCH(fa_bwd_dkdv_kernel, dim3(dkdv_grid), dim3(kFaDkdvThreads), kDmaLds, st, (const bf16_t*)q, (const bf16_t*)k,
                   (const bf16_t*)v, (const bf16_t*)dout, q_stride, k_stride, v_stride, dout_stride, cu_seqlens, k_tiles,
                   (int)num_heads, (int)num_kv_heads, scale * log2e, scale, nl, nd, total_tokens, (bf16_t*)dk,
                   (bf16_t*)dv, dk_stride, dv_stride, (int)n_k_tiles);
    } else {
        const int group = (int)(num_heads / num_kv_heads);
        int gshift = 0;
        while ((1 << gshift) < group) ++gshift;
        // the downward sweep indexes (slice, head) by shift / mask: a group that is not a power of two keeps the ascending order
        if (sweep_down && (1 << gshift) == group)
            RPO_LAUNCH(fa_bwd_dkdv4_kernel<true>, dim3(dkdv_grid), dim3(256), kDkdv4Lds, st, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, (const bf16_t*)dout, q_stride, k_stride, v_stride, dout_stride, cu_seqlens, k_tiles,
                       (int)num_heads, (int)num_kv_heads, scale * log2e, scale, nl, nd, total_tokens, (bf16_t*)dk,
                       (bf16_t*)dv, dk_stride, dv_stride, (int)n_k_tiles, gshift, rope_cos, rope_sin, rope_period);
        else
            RPO_LAUNCH(fa_bwd_dkdv4_kernel<false>, dim3(dkdv_grid), dim3(256), kDkdv4Lds, st, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, (const bf16_t*)dout, q_stride, k_stride, v_stride, dout_stride, cu_seqlens, k_tiles,
                       (int)num_heads, (int)num_kv_heads, scale * log2e, scale, nl, nd, total_tokens, (bf16_t*)dk,
                       (bf16_t*)dv, dk_stride, dv_stride, (int)n_k_tiles, gshift, rope_cos, rope_sin, rope_period);
    }
    return rpo_launch_status();
}
