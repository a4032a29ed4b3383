// Fused elementwise pieces of the Llama block that PyTorch runs as several kernels each (profiles/r01_*: mul / add /
// silu / cat / neg passes are ~17 % of the step).  HBM-bound, 16-byte accesses, one pass.
//   swiglu_fwd : out = silu(g) * u                         (3 n s bytes)
//   swiglu_bwd : dg = dout * u * silu'(g), du = dout * silu(g)   (5 n s bytes)
//   rope       : rotate pairs (j, j + hd/2) of every head by the row's angle (x_in may equal x); backward = inverse
// The encoder itself stays under PyTorch-ROCm (north star); these calls are used by rankpo_amd/encoder.py only.
#include "common.hpp"

namespace {

constexpr int kEwThreads = 256;

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }

// 2-D form: g, u: [rows, cols] with row stride ld_gu (they may be the two halves of one fused gate|up projection
// output), out: [rows, cols] with row stride ld_out.  Block (bx, row-group): one 16-byte vector per thread.
template <typename T>
__global__ __launch_bounds__(kEwThreads) void swiglu_fwd_kernel(const T* __restrict__ g, const T* __restrict__ u,
                                                                T* __restrict__ out, int64_t rows, int vec_per_row,
                                                                int64_t ld_gu, int64_t ld_out) {
    // ONE 16-byte vector per thread, consecutive blocks own consecutive vectors, streaming (non-temporal) loads and
    // stores: measured 6.6 TB/s on MI355X vs 4.6-4.9 TB/s for a capped grid-stride loop
    // (tools/exp/exp_stream.hip).
    constexpr int V = Elem<T>::kVec;
    const int64_t i = (int64_t)blockIdx.x * kEwThreads + threadIdx.x;
    const int64_t r = i / vec_per_row;
    if (r >= rows) return;
    const int c = (int)(i - r * vec_per_row) * V;
    Vec16<T> a, b, o;
    a.load_nt(g + r * ld_gu + c);
    b.load_nt(u + r * ld_gu + c);
#pragma unroll
    for (int k = 0; k < V; ++k) o.v[k] = a.v[k] * sigmoid_f(a.v[k]) * b.v[k];
    o.store_nt(out + r * ld_out + c);
}

template <typename T>
__global__ __launch_bounds__(kEwThreads) void swiglu_bwd_kernel(const T* g, const T* u, const T* dout, T* dg, T* du,
                                                                T* prod, int64_t rows, int vec_per_row, int64_t ld_gu,
                                                                int64_t ld_dout, int64_t ld_dgu,
                                                                int64_t ld_prod) {   // outputs may alias inputs
    constexpr int V = Elem<T>::kVec;
    const int64_t i = (int64_t)blockIdx.x * kEwThreads + threadIdx.x;
    const int64_t r = i / vec_per_row;
    if (r >= rows) return;
    const int c = (int)(i - r * vec_per_row) * V;
    Vec16<T> a, b, d, og, ou, op;
    a.load_nt(g + r * ld_gu + c);
    b.load_nt(u + r * ld_gu + c);
    d.load_nt(dout + r * ld_dout + c);
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const float s = sigmoid_f(a.v[k]);
        const float silu = a.v[k] * s;
        og.v[k] = d.v[k] * b.v[k] * (s + silu * (1.0f - s));   // silu' = s (1 + g (1 - s))
        ou.v[k] = d.v[k] * silu;
        op.v[k] = silu * b.v[k];                               // the forward product, same arithmetic as swiglu_fwd_kernel
    }
    og.store_nt(dg + r * ld_dgu + c);
    ou.store_nt(du + r * ld_dgu + c);
    if (prod) op.store_nt(prod + r * ld_prod + c);
}

// x: [rows, H, hd] (row stride = row_stride elements, heads contiguous), cos/sin: f32 [period, hd/2];
// row r uses table row r % period.  One block per row; thread handles 8 (bf16) / 4 (f32) consecutive dims of the
// low half and the matching dims of the high half.  sign = +1 forward, -1 backward.
template <typename T>
__global__ __launch_bounds__(kEwThreads) void rope_kernel(const T* xin, T* x, int64_t row_stride,
                                                          const float* __restrict__ cs, const float* __restrict__ sn,
                                                          int64_t rows, int H, int hd, int64_t period, float sign) {
    constexpr int V = Elem<T>::kVec;
    const int half = hd >> 1;
    const int vec_per_head = half / V;
    const int total = H * vec_per_head;            // (low, high) vector pairs of one row
    // A row with few pairs (round 3: the rotary fold leaves this kernel the K heads only, 8 heads x 4 pairs = 32 of the block's
    // 256 threads) shares the block with its neighbours: rpb rows per block, thread -> (row tid / total, pair tid % total)
    const int rpb = total <= kEwThreads / 2 ? kEwThreads / total : 1;
    const int rin = rpb > 1 ? threadIdx.x / total : 0;
    const int i0 = rpb > 1 ? threadIdx.x - rin * total : threadIdx.x;
    if (rin >= rpb) return;                        // (threads beyond rpb * total when total does not divide the block)
    for (int64_t r = (int64_t)blockIdx.x * rpb + rin; r < rows; r += (int64_t)gridDim.x * rpb) {
        T* xr = x + r * row_stride;
        const T* xi = xin + r * row_stride;
        const float* c = cs + (r % period) * half;
        const float* s = sn + (r % period) * half;
        for (int i = i0; i < total; i += (rpb > 1 ? total : kEwThreads)) {
            const int h = i / vec_per_head, j = (i - h * vec_per_head) * V;
            T* lo = xr + h * hd + j;
            T* hi = lo + half;
            Vec16<T> a, b, oa, ob;
            a.load_nt(xi + h * hd + j);
            b.load_nt(xi + h * hd + j + half);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float cc = c[j + k], ss = sign * s[j + k];
                oa.v[k] = a.v[k] * cc - b.v[k] * ss;      // x1 cos - x2 sin
                ob.v[k] = b.v[k] * cc + a.v[k] * ss;      // x2 cos + x1 sin
            }
            oa.store_nt(lo);
            ob.store_nt(hi);
        }
    }
}

// The same backward pass with the recomputed product written TRANSPOSED, prod_t [cols, rows] (round 4).  Why: the weight gradient of
// the down projection, dW [d, ff] = dY[T, d]^T prod[T, ff], runs at 1.37 PFLOP/s in the layout round 2 gave it (dY^T contiguous
// along the tokens, prod strided) and at 1.57 with BOTH operands contiguous along the token reduction (tools/probe_wgrad.py: 3.70 ->
// 3.23 ms per block at T = 151 552) -- which needs prod^T, and this kernel is the one that writes the product anyway: the same
// bytes, stored through an LDS tile as whole 128-byte lines of the transposed matrix (the 64 x 64 tile of transpose_kernel,
// measured the faster one there).  dg / du stay row-major (the input-gradient GEMM reads them that way).
template <typename T, int TC, bool DGU_T>
__global__ __launch_bounds__(256) void swiglu_bwd_t_kernel(const T* __restrict__ g, const T* __restrict__ u, const T* __restrict__ dout,
                                                           T* __restrict__ dg, T* __restrict__ du, T* __restrict__ prod_t,
                                                           T* __restrict__ dgu_t, int64_t rows, int64_t cols, int64_t ld_gu,
                                                           int64_t ld_dout, int64_t ld_dgu, int64_t ld_pt) {
    // tile = TR tokens x TC columns: the five row-major streams (g, u, dout in; dg, du out) move TC * sizeof(T) contiguous bytes
    // per token row (512 B at TC = 256: a 64 x 64 tile's 128-byte pieces on ALL six streams measured 5.0 TB/s, 2.95 ms against
    // the streaming kernel's 2.27 -- more than the GEMM wins back); only the transposed outputs are written in TR-token pieces.
    // DGU_T: dg and du are ALSO written transposed, dgu_t [2 cols, rows] (dg^T in rows 0 .. cols - 1, du^T behind it), for the
    // weight gradient of the fused gate|up projection; the three transposed tiles go through the ONE LDS tile one after the
    // other, the packed results waiting in registers.
    constexpr int V = Elem<T>::kVec;
    constexpr int TR = 64;
    constexpr int LDT = TC + 8 / (int)sizeof(T);
    constexpr int VPR = TC / V;                       // vectors per tile row
    constexpr int VPC = TR / V;                       // vectors per output row (one column of the tile)
    constexpr int NI = TR * VPR / 256;                // vectors per thread
    __shared__ __attribute__((aligned(16))) T tile[TR][LDT];
    const int64_t r0 = (int64_t)blockIdx.y * TR, c0 = (int64_t)blockIdx.x * TC;
    const int t = threadIdx.x;
    // Tile rows are LDT * sizeof(T) = 520 bytes (bf16): odd rows are 8-byte aligned only, so a 16-byte vector goes in as two
    // 8-byte stores (a `ds_write_b128` there is a misaligned access: advisor, round 4; two `ds_write_b64` cost the same 12-13
    // cycles).  Rows of a 16-byte multiple would make the transposed 2-byte reads below 4-way bank-conflicted instead of 2-way
    // (8 rows apart = 8 * stride dwords = 0 or 32 mod 64 banks for every 16-byte-multiple stride).
    auto st_tile = [](T* dst, const uint4_t v) {
        uint2_t* d2 = reinterpret_cast<uint2_t*>(dst);
        d2[0] = uint2_t{v.x, v.y};
        d2[1] = uint2_t{v.z, v.w};
    };
    uint4_t pk_g[DGU_T ? NI : 1], pk_u[DGU_T ? NI : 1];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int idx = t + 256 * i;
        const int row = idx / VPR, cv = (idx % VPR) * V;
        const int64_t r = r0 + row, c = c0 + cv;
        __attribute__((aligned(16))) T pv[V];
        __attribute__((aligned(16))) T gv[V];
        __attribute__((aligned(16))) T uv[V];
        if (r < rows && c < cols) {                      // cols is a multiple of V: a vector is inside or outside as a whole
            Vec16<T> a, b, d, og, ou;
            a.load_nt(g + r * ld_gu + c);
            b.load_nt(u + r * ld_gu + c);
            d.load_nt(dout + r * ld_dout + c);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float sg = sigmoid_f(a.v[k]);
                const float silu = a.v[k] * sg;
                og.v[k] = d.v[k] * b.v[k] * (sg + silu * (1.0f - sg));
                ou.v[k] = d.v[k] * silu;
                Elem<T>::st(&pv[k], silu * b.v[k]);            // the forward product, same arithmetic and rounding as swiglu_fwd_kernel
                if constexpr (DGU_T) {
                    Elem<T>::st(&gv[k], og.v[k]);
                    Elem<T>::st(&uv[k], ou.v[k]);
                }
            }
            og.store_nt(dg + r * ld_dgu + c);
            ou.store_nt(du + r * ld_dgu + c);
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) pv[k] = gv[k] = uv[k] = T(0);
        }
        st_tile(&tile[row][cv], *reinterpret_cast<const uint4_t*>(pv));
        if constexpr (DGU_T) {
            pk_g[i] = *reinterpret_cast<const uint4_t*>(gv);
            pk_u[i] = *reinterpret_cast<const uint4_t*>(uv);
        }
    }
    auto flush = [&](T* __restrict__ out) {                 // the LDS tile, transposed, to out[c0 + oc][r0 ..]
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TC * VPC / 256; ++i) {
            const int idx = t + 256 * i;
            const int oc = idx / VPC, rv = (idx % VPC) * V;      // output row = column c0 + oc; tokens r0 + rv .. + V - 1
            const int64_t c = c0 + oc, r = r0 + rv;
            if (c >= cols || r >= rows) continue;
            __attribute__((aligned(16))) T v[V];
#pragma unroll
            for (int e = 0; e < V; ++e) v[e] = tile[rv + e][oc];
            if (r + V <= rows) {
                __builtin_nontemporal_store(*reinterpret_cast<const uint4_t*>(v), reinterpret_cast<uint4_t*>(out + c * ld_pt + r));
            } else {
#pragma unroll
                for (int e = 0; e < V; ++e)
                    if (r + e < rows) out[c * ld_pt + r + e] = v[e];
            }
        }
    };
    flush(prod_t);
    if constexpr (DGU_T) {
        auto refill = [&](const uint4_t* pk) {
            __syncthreads();                                 // everybody has read the previous tile
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int idx = t + 256 * i;
                st_tile(&tile[idx / VPR][(idx % VPR) * V], pk[i]);
            }
        };
        refill(pk_g);
        flush(dgu_t);
        refill(pk_u);
        flush(dgu_t + cols * ld_pt);
    }
}

inline unsigned ew_grid(int64_t nvec) { return (unsigned)rpo_cdiv(nvec, kEwThreads); }

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// 2-D transpose  out[C, R] = in[R, C]^T  (both row-major, row strides ld_in / ld_out elements).
// Why it exists: hipBLASLt's weight-gradient GEMM dW = dY^T X with BOTH operands strided along the token reduction runs at
// 0.9-1.2 PFLOP/s; with ONE operand contiguous along it, 1.36-1.39 (tools/probe_wgrad.py), and the input-gradient GEMM wants
// W^T.  PyTorch's `.t().contiguous()` moves 0.35 TB/s on these shapes (3.4 ms for a [151552, 2048] bf16 operand); this kernel
// is HBM-bound.  64 x 64 tile through LDS: 16-byte global loads along the input rows, 2-byte (bf16) / 4-byte (f32) LDS reads
// down the tile columns, 16-byte global stores along the output rows: whole 128-byte lines on both sides.
// (Round 3, measured and dropped: a 128 x 128 tile -- 256-byte pieces on both sides -- moved 5.03 TB/s on [151552, 2048] against
// 5.83 for this one, 5.49 against 5.79 on [151552, 4096]: tools/transpose_ab.py (round 3; git history).)
// ------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t R, int64_t C,
                                                        int64_t ld_in, int64_t ld_out, int vec_in, int vec_out) {
    constexpr int V = 16 / (int)sizeof(T);            // elements per 16-byte vector
    constexpr int TS = 64;                            // tile side
    constexpr int LDT = TS + 8 / (int)sizeof(T);      // 136-byte (bf16) / 264-byte (f32) LDS rows: 8-byte aligned, banks spread
    constexpr int VPR = TS / V;                       // vectors per tile row
    __shared__ __attribute__((aligned(16))) T tile[TS][LDT];
    const int64_t r0 = (int64_t)blockIdx.y * TS, c0 = (int64_t)blockIdx.x * TS;
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < TS * VPR / 256; ++i) {
        const int idx = t + 256 * i;
        const int row = idx / VPR, cv = (idx % VPR) * V;
        const int64_t r = r0 + row, c = c0 + cv;
        __attribute__((aligned(16))) T v[V];
        if (r < R && vec_in && c + V <= C) {
            *reinterpret_cast<uint4_t*>(v) = __builtin_nontemporal_load(reinterpret_cast<const uint4_t*>(in + r * ld_in + c));
        } else {
#pragma unroll
            for (int e = 0; e < V; ++e) v[e] = (r < R && c + e < C) ? in[r * ld_in + c + e] : T(0);
        }
#pragma unroll
        for (int e = 0; e < V; ++e) tile[row][cv + e] = v[e];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TS * VPR / 256; ++i) {
        const int idx = t + 256 * i;
        const int oc = idx / VPR, rv = (idx % VPR) * V;          // output row = input column c0 + oc; input rows r0 + rv ..
        const int64_t c = c0 + oc, r = r0 + rv;
        if (c >= C) continue;
        __attribute__((aligned(16))) T v[V];
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] = tile[rv + e][oc];
        if (vec_out && r + V <= R) {
            __builtin_nontemporal_store(*reinterpret_cast<const uint4_t*>(v), reinterpret_cast<uint4_t*>(out + c * ld_out + r));
        } else {
#pragma unroll
            for (int e = 0; e < V; ++e)
                if (r + e < R) out[c * ld_out + r + e] = v[e];
        }
    }
}

extern "C" int rpo_transpose(const void* in, void* out, int64_t rows, int64_t cols, int64_t ld_in, int64_t ld_out, int dtype,
                             rpo_stream_t stream) {
    if (!in || !out || rows <= 0 || cols <= 0 || ld_in < cols || ld_out < rows) return RPO_ERR_INVALID_ARG;
    if (!rpo_dtype_ok(dtype)) return RPO_ERR_INVALID_ARG;
    if (dtype == RPO_DT_F16) dtype = RPO_DT_BF16;               // a transpose moves bits: any 2-byte element
    const int64_t gx = rpo_cdiv(cols, 64), gy = rpo_cdiv(rows, 64);
    if (gy > 65535) return RPO_ERR_UNSUPPORTED;                 // 4 M rows
    const int es = dtype == RPO_DT_BF16 ? 2 : 4, V = 16 / es;
    const int vec_in = rpo_aligned16(in) && (ld_in % V == 0), vec_out = rpo_aligned16(out) && (ld_out % V == 0);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)gx, (unsigned)gy);
    if (dtype == RPO_DT_BF16)
        RPO_LAUNCH(transpose_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)in, (bf16_t*)out, rows, cols, ld_in, ld_out,
                   vec_in, vec_out);
    else
        RPO_LAUNCH(transpose_kernel<float>, grid, dim3(256), 0, st, (const float*)in, (float*)out, rows, cols, ld_in, ld_out,
                   vec_in, vec_out);
    return rpo_launch_status();
}

extern "C" int rpo_swiglu_fwd(const void* g, const void* u, void* out, int64_t rows, int64_t cols, int64_t ld_gu,
                              int64_t ld_out, int dtype, rpo_stream_t stream) {
    if (!g || !u || !out || rows <= 0 || cols <= 0) return RPO_ERR_INVALID_ARG;
    const int V = dtype == RPO_DT_BF16 ? 8 : 4;
    if (cols % V != 0 || ld_gu % V != 0 || ld_out % V != 0 || ld_gu < cols || ld_out < cols || !rpo_aligned16(g) ||
        !rpo_aligned16(u) || !rpo_aligned16(out))
        return RPO_ERR_UNSUPPORTED;
    const int64_t nvec = rows * (cols / V);
    if (nvec / kEwThreads >= INT32_MAX) return RPO_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == RPO_DT_BF16)
        RPO_LAUNCH(swiglu_fwd_kernel<bf16_t>, dim3(ew_grid(nvec)), dim3(kEwThreads), 0, st, (const bf16_t*)g,
                   (const bf16_t*)u, (bf16_t*)out, rows, (int)(cols / V), ld_gu, ld_out);
    else if (dtype == RPO_DT_F32)
        RPO_LAUNCH(swiglu_fwd_kernel<float>, dim3(ew_grid(nvec)), dim3(kEwThreads), 0, st, (const float*)g,
                   (const float*)u, (float*)out, rows, (int)(cols / V), ld_gu, ld_out);
    else
        return RPO_ERR_INVALID_ARG;
    return rpo_launch_status();
}

extern "C" int rpo_swiglu_bwd(const void* g, const void* u, const void* dout, void* dg, void* du, void* prod_out,
                              int64_t rows, int64_t cols, int64_t ld_gu, int64_t ld_dout, int64_t ld_dgu, int64_t ld_prod,
                              int dtype, rpo_stream_t stream) {
    if (!g || !u || !dout || !dg || !du || rows <= 0 || cols <= 0) return RPO_ERR_INVALID_ARG;
    const int V = dtype == RPO_DT_BF16 ? 8 : 4;
    if (cols % V != 0 || ld_gu % V != 0 || ld_dout % V != 0 || ld_dgu % V != 0 || !rpo_aligned16(g) ||
        !rpo_aligned16(u) || !rpo_aligned16(dout) || !rpo_aligned16(dg) || !rpo_aligned16(du) ||
        (prod_out && (ld_prod % V != 0 || !rpo_aligned16(prod_out))))
        return RPO_ERR_UNSUPPORTED;
    const int64_t nvec = rows * (cols / V);
    if (nvec / kEwThreads >= INT32_MAX) return RPO_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == RPO_DT_BF16)
        RPO_LAUNCH(swiglu_bwd_kernel<bf16_t>, dim3(ew_grid(nvec)), dim3(kEwThreads), 0, st, (const bf16_t*)g,
                   (const bf16_t*)u, (const bf16_t*)dout, (bf16_t*)dg, (bf16_t*)du, (bf16_t*)prod_out, rows, (int)(cols / V),
                   ld_gu, ld_dout, ld_dgu, ld_prod);
    else if (dtype == RPO_DT_F32)
        RPO_LAUNCH(swiglu_bwd_kernel<float>, dim3(ew_grid(nvec)), dim3(kEwThreads), 0, st, (const float*)g,
                   (const float*)u, (const float*)dout, (float*)dg, (float*)du, (float*)prod_out, rows, (int)(cols / V), ld_gu,
                   ld_dout, ld_dgu, ld_prod);
    else
        return RPO_ERR_INVALID_ARG;
    return rpo_launch_status();
}

extern "C" int rpo_swiglu_bwd_t(const void* g, const void* u, const void* dout, void* dg, void* du, void* prod_t_out,
                                void* dgu_t_out, int64_t rows, int64_t cols, int64_t ld_gu, int64_t ld_dout, int64_t ld_dgu,
                                int64_t ld_t, int dtype, rpo_stream_t stream) {
    if (!g || !u || !dout || !dg || !du || !prod_t_out || rows <= 0 || cols <= 0) return RPO_ERR_INVALID_ARG;
    const int V = dtype == RPO_DT_BF16 ? 8 : 4;
    if (cols % V != 0 || ld_gu % V != 0 || ld_dout % V != 0 || ld_dgu % V != 0 || ld_t % V != 0 || ld_t < rows ||
        !rpo_aligned16(g) || !rpo_aligned16(u) || !rpo_aligned16(dout) || !rpo_aligned16(dg) || !rpo_aligned16(du) ||
        !rpo_aligned16(prod_t_out) || (dgu_t_out && !rpo_aligned16(dgu_t_out)))
        return RPO_ERR_UNSUPPORTED;
    // the transposed outputs may not overlay the inputs (a tile's product is written after OTHER tiles may still read dout)
    constexpr int kTC = 256;
    const int64_t gy = rpo_cdiv(rows, 64);
    if (gy > 65535) return RPO_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
#define RPO_SWT(T, TCV, FLAG)                                                                                               \
    RPO_LAUNCH((swiglu_bwd_t_kernel<T, TCV, FLAG>), dim3((unsigned)rpo_cdiv(cols, TCV), (unsigned)gy), dim3(256), 0, st,    \
               (const T*)g, (const T*)u, (const T*)dout, (T*)dg, (T*)du, (T*)prod_t_out, (T*)dgu_t_out, rows, cols, ld_gu,   \
               ld_dout, ld_dgu, ld_t)
    if (dtype == RPO_DT_BF16) {
        if (dgu_t_out) RPO_SWT(bf16_t, kTC, true);
        else RPO_SWT(bf16_t, kTC, false);
    } else if (dtype == RPO_DT_F32) {
        if (dgu_t_out) RPO_SWT(float, kTC / 2, true);
        else RPO_SWT(float, kTC / 2, false);
    } else {
        return RPO_ERR_INVALID_ARG;
    }
#undef RPO_SWT
    return rpo_launch_status();
}

extern "C" int rpo_rope(const void* x_in, void* x, int64_t row_stride, const float* cos_tab, const float* sin_tab, int64_t rows,
                                int64_t heads, int64_t head_dim, int64_t period, int dtype, int backward,
                                rpo_stream_t stream) {
    if (!x_in || !x || !cos_tab || !sin_tab || rows <= 0 || heads <= 0 || head_dim <= 0 || period <= 0)
        return RPO_ERR_INVALID_ARG;
    const int V = dtype == RPO_DT_BF16 ? 8 : 4;
    if ((head_dim / 2) % V != 0 || head_dim % 2 != 0 || row_stride % V != 0 || !rpo_aligned16(x) || !rpo_aligned16(x_in))
        return RPO_ERR_UNSUPPORTED;
    const int64_t pairs = heads * ((head_dim / 2) / V);
    const int64_t rpb = pairs <= kEwThreads / 2 ? kEwThreads / pairs : 1;      // rows per block (see the kernel)
    const int64_t nblk = rpo_cdiv(rows, rpb);
    int64_t grid = nblk < INT32_MAX ? nblk : INT32_MAX;
    hipStream_t st = (hipStream_t)stream;
    const float sign = backward ? -1.0f : 1.0f;
    if (dtype == RPO_DT_BF16)
        RPO_LAUNCH(rope_kernel<bf16_t>, dim3((unsigned)grid), dim3(kEwThreads), 0, st, (const bf16_t*)x_in, (bf16_t*)x, row_stride, cos_tab,
                   sin_tab, rows, (int)heads, (int)head_dim, period, sign);
    else if (dtype == RPO_DT_F32)
        RPO_LAUNCH(rope_kernel<float>, dim3((unsigned)grid), dim3(kEwThreads), 0, st, (const float*)x_in, (float*)x, row_stride, cos_tab,
                   sin_tab, rows, (int)heads, (int)head_dim, period, sign);
    else
        return RPO_ERR_INVALID_ARG;
    return rpo_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// Residual add + RMSNorm, fused (replaces `x = x + delta; y = rms_norm(x) * w` = 2-3 PyTorch kernels forward and
// cuComputeGradInput + cuComputePartGradGammaBeta + an add kernel backward: ~7.5 % of the cfg-2 step).
// One wave per row, lane l owns the 16-byte vectors l, l + 64, ... of the row; each wave walks a contiguous chunk
// of rows.  forward:  x_new = x + delta (rounded to the storage dtype), rstd = rsqrt(mean(x_new^2) + eps),
//                     y = x_new * rstd * w.
// backward: g = dy * w; c = mean(g * xhat); dx = (g - xhat * c) * rstd + dres;  dw partial sums per wave (f32,
//           fixed order) -> dw_partial[wave][d], summed by the caller.
// ------------------------------------------------------------------------------------------------------------------
namespace {

constexpr int kNormThreads = 256, kNormWaves = kNormThreads / 64;

// 16 bytes of T kept PACKED in registers (4 VGPRs) and widened element by element at the point of use: half the registers
// of Vec16<bf16_t> for operands that are only read once per row (weight, incoming residual gradient)
template <typename T> struct Raw16 {
    uint4_t r;
    __device__ __forceinline__ void load_nt(const T* p) { r = __builtin_nontemporal_load(reinterpret_cast<const uint4_t*>(p)); }
    __device__ __forceinline__ void load(const T* p) { r = *reinterpret_cast<const uint4_t*>(p); }
    __device__ __forceinline__ float get(int e) const;
};
template <> __device__ __forceinline__ float Raw16<float>::get(int e) const { return __uint_as_float(r[e]); }
template <> __device__ __forceinline__ float Raw16<bf16_t>::get(int e) const {
    return (e & 1) ? __uint_as_float(r[e >> 1] & 0xffff0000u) : __uint_as_float(r[e >> 1] << 16);
}

template <typename T, int KMAX>
__global__ __launch_bounds__(kNormThreads) void add_rmsnorm_fwd_kernel(const T* __restrict__ x, const T* __restrict__ delta,
                                                                       const T* __restrict__ w, float eps,
                                                                       T* __restrict__ x_out, T* __restrict__ y,
                                                                       float* __restrict__ rstd_out, int64_t rows,
                                                                       int d, int64_t rows_per_wave) {
    constexpr int V = Elem<T>::kVec;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * kNormWaves + (threadIdx.x >> 6);
    const int nvec = d / V;
    float wv[KMAX][V];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int vi = lane + 64 * k;
        if (vi < nvec) {
            Vec16<T> t;
            t.load(w + vi * V);
#pragma unroll
            for (int e = 0; e < V; ++e) wv[k][e] = t.v[e];
        }
    }
    const int64_t r0 = wave * rows_per_wave, r1 = r0 + rows_per_wave < rows ? r0 + rows_per_wave : rows;
    const float inv_d = 1.0f / (float)d;
    for (int64_t r = r0; r < r1; ++r) {
        Vec16<T> xv[KMAX];
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int vi = lane + 64 * k;
            if (vi < nvec) {
                xv[k].load_nt(x + r * d + vi * V);
                if (delta) {
                    Vec16<T> dv;
                    dv.load_nt(delta + r * d + vi * V);
#pragma unroll
                    for (int e = 0; e < V; ++e) xv[k].v[e] = Elem<T>::round(xv[k].v[e] + dv.v[e]);
                    xv[k].store(x_out + r * d + vi * V);        // read again by the next kernels: keep it cacheable
                }
#pragma unroll
                for (int e = 0; e < V; ++e) ss = fmaf(xv[k].v[e], xv[k].v[e], ss);
            }
        }
        ss = wave_sum(ss);
        const float rstd = rsqrtf(ss * inv_d + eps);
        if (lane == 0) rstd_out[r] = rstd;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int vi = lane + 64 * k;
            if (vi < nvec) {
                Vec16<T> o;
#pragma unroll
                for (int e = 0; e < V; ++e) o.v[e] = xv[k].v[e] * rstd * wv[k][e];
                o.store(y + r * d + vi * V);
            }
        }
    }
}

template <typename T, int KMAX>
__global__ __launch_bounds__(kNormThreads) void add_rmsnorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ xn,
                                                                       const T* __restrict__ w,
                                                                       const float* __restrict__ rstd_in,
                                                                       const T* __restrict__ dres, T* __restrict__ dx,
                                                                       float* __restrict__ dw_partial, int64_t rows,
                                                                       int d, int64_t rows_per_wave) {
    constexpr int V = Elem<T>::kVec;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * kNormWaves + (threadIdx.x >> 6);
    const int nvec = d / V;
    float dwa[KMAX][V];
    Raw16<T> wv[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int vi = lane + 64 * k;
#pragma unroll
        for (int e = 0; e < V; ++e) dwa[k][e] = 0.f;
        if (vi < nvec) wv[k].load(w + vi * V);
    }
    const int64_t r0 = wave * rows_per_wave, r1 = r0 + rows_per_wave < rows ? r0 + rows_per_wave : rows;
    const float inv_d = 1.0f / (float)d;
    for (int64_t r = r0; r < r1; ++r) {
        const float rstd = rstd_in[r];
        Raw16<T> gv[KMAX], xv[KMAX], rv[KMAX];                // dy, x_new, incoming residual gradient: packed
        float c = 0.f;
        // all three streams of the row are requested before anything is consumed: ONE memory round trip per row
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int vi = lane + 64 * k;
            if (vi < nvec) {
                gv[k].load_nt(dy + r * d + vi * V);
                xv[k].load_nt(xn + r * d + vi * V);
                if (dres) rv[k].load_nt(dres + r * d + vi * V);
            }
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int vi = lane + 64 * k;
            if (vi < nvec) {
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const float xh = xv[k].get(e) * rstd;             // xhat
                    const float dyv = gv[k].get(e);
                    dwa[k][e] = fmaf(dyv, xh, dwa[k][e]);
                    c = fmaf(dyv * wv[k].get(e), xh, c);              // g = dy * w
                }
            }
        }
        c = wave_sum(c) * inv_d;
        // keep the operands PACKED across the reduction: without this hipcc carries the widened x-hat / g / w values of the
        // first pass over to the second one (196 VGPRs, 2 waves per SIMD instead of 3)
#pragma unroll
        for (int k = 0; k < KMAX; ++k) asm volatile("" : "+v"(gv[k].r), "+v"(xv[k].r), "+v"(wv[k].r));
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int vi = lane + 64 * k;
            if (vi < nvec) {
                Vec16<T> o;
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const float xh = xv[k].get(e) * rstd;
                    const float g = gv[k].get(e) * wv[k].get(e);
                    o.v[e] = (g - xh * c) * rstd;
                }
                if (dres) {
#pragma unroll
                    for (int e = 0; e < V; ++e) o.v[e] += rv[k].get(e);
                }
                o.store(dx + r * d + vi * V);
            }
        }
    }
    if (r0 < rows || true) {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int vi = lane + 64 * k;
            if (vi < nvec) {
                Vec16<float> o;
                // V floats = V/4 16-byte stores
#pragma unroll
                for (int h = 0; h < V / 4; ++h) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o.v[e] = dwa[k][4 * h + e];
                    o.store(dw_partial + wave * d + vi * V + 4 * h);
                }
            }
        }
    }
}

template <typename T>
int launch_norm_fwd(const void* x, const void* delta, const void* w, float eps, void* x_out, void* y, float* rstd,
                    int64_t rows, int64_t d, int nwaves, hipStream_t st) {
    constexpr int V = Elem<T>::kVec;
    const int kk = (int)rpo_cdiv(d / V, 64);
    const int64_t rpw = rpo_cdiv(rows, nwaves);
    const dim3 grid((unsigned)rpo_cdiv(nwaves, kNormWaves)), block(kNormThreads);
#define RPO_NF(K)                                                                                               \
    RPO_LAUNCH((add_rmsnorm_fwd_kernel<T, K>), grid, block, 0, st, (const T*)x, (const T*)delta, (const T*)w, eps, \
               (T*)x_out, (T*)y, rstd, rows, (int)d, rpw)
    if (kk <= 1) RPO_NF(1);
    else if (kk <= 2) RPO_NF(2);
    else if (kk <= 4) RPO_NF(4);
    else if (kk <= 8) RPO_NF(8);
    else if (kk <= 16) RPO_NF(16);
    else return RPO_ERR_UNSUPPORTED;
#undef RPO_NF
    return rpo_launch_status();
}

template <typename T>
int launch_norm_bwd(const void* dy, const void* xn, const void* w, const float* rstd, const void* dres, void* dx,
                    float* dwp, int64_t rows, int64_t d, int nwaves, hipStream_t st) {
    constexpr int V = Elem<T>::kVec;
    const int kk = (int)rpo_cdiv(d / V, 64);
    const int64_t rpw = rpo_cdiv(rows, nwaves);
    const dim3 grid((unsigned)rpo_cdiv(nwaves, kNormWaves)), block(kNormThreads);
#define RPO_NB(K)                                                                                                \
    RPO_LAUNCH((add_rmsnorm_bwd_kernel<T, K>), grid, block, 0, st, (const T*)dy, (const T*)xn, (const T*)w, rstd,  \
               (const T*)dres, (T*)dx, dwp, rows, (int)d, rpw)
    if (kk <= 1) RPO_NB(1);
    else if (kk <= 2) RPO_NB(2);
    else if (kk <= 4) RPO_NB(4);
    else if (kk <= 8) RPO_NB(8);
    else return RPO_ERR_UNSUPPORTED;
#undef RPO_NB
    return rpo_launch_status();
}

}  // namespace

#ifndef RPO_NORM_FWD_CAP
#define RPO_NORM_FWD_CAP 8192
#endif
#ifndef RPO_NORM_BWD_CAP
#define RPO_NORM_BWD_CAP 8192
#endif
static int norm_waves(int64_t rows, int64_t cap) {
    int64_t w = rows < cap ? rows : cap;
    w = (w + 3) / 4 * 4;
    return (int)(w < 4 ? 4 : w);
}

extern "C" int rpo_add_rmsnorm_waves(int64_t rows) {
    // number of waves (= rows of dw_partial) the BACKWARD kernel uses for `rows` rows: a multiple of 4, <= the cap
    return norm_waves(rows, RPO_NORM_BWD_CAP);
}

extern "C" int rpo_add_rmsnorm_fwd(const void* x, const void* delta, const void* weight, float eps, void* x_out,
                                   void* y_out, float* rstd_out, int64_t rows, int64_t d, int dtype,
                                   rpo_stream_t stream) {
    if (!x || !weight || !y_out || !rstd_out || rows <= 0 || d <= 0) return RPO_ERR_INVALID_ARG;
    if (delta && !x_out) return RPO_ERR_INVALID_ARG;
    const int V = dtype == RPO_DT_BF16 ? 8 : 4;
    if (d % V != 0 || !rpo_aligned16(x) || !rpo_aligned16(y_out) || !rpo_aligned16(weight) ||
        (delta && (!rpo_aligned16(delta) || !rpo_aligned16(x_out))))
        return RPO_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    // (round 3, measured and dropped: ONE row per wave, as many waves as rows -- the forward has no per-wave output that would
    // bound its wave count -- moved 5.71 TB/s at d = 2048 against 5.98 for this capped loop, 5.06 against 4.96 at d = 4096:
    // tools/norm_ab.py (round 4; git history))
    const int nw = norm_waves(rows, RPO_NORM_FWD_CAP);       // (the forward writes no per-wave partials: its count is its own)
    if (dtype == RPO_DT_BF16) return launch_norm_fwd<bf16_t>(x, delta, weight, eps, x_out, y_out, rstd_out, rows, d, nw, st);
    if (dtype == RPO_DT_F32) return launch_norm_fwd<float>(x, delta, weight, eps, x_out, y_out, rstd_out, rows, d, nw, st);
    return RPO_ERR_INVALID_ARG;
}

extern "C" int rpo_add_rmsnorm_bwd(const void* dy, const void* x_new, const void* weight, const float* rstd,
                                   const void* dres, void* dx_out, float* dw_partial, int64_t rows, int64_t d,
                                   int dtype, rpo_stream_t stream) {
    if (!dy || !x_new || !weight || !rstd || !dx_out || !dw_partial || rows <= 0 || d <= 0) return RPO_ERR_INVALID_ARG;
    const int V = dtype == RPO_DT_BF16 ? 8 : 4;
    if (d % V != 0 || !rpo_aligned16(dy) || !rpo_aligned16(x_new) || !rpo_aligned16(dx_out) || !rpo_aligned16(weight) ||
        !rpo_aligned16(dw_partial) || (dres && !rpo_aligned16(dres)))
        return RPO_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int nw = rpo_add_rmsnorm_waves(rows);
    if (dtype == RPO_DT_BF16) return launch_norm_bwd<bf16_t>(dy, x_new, weight, rstd, dres, dx_out, dw_partial, rows, d, nw, st);
    if (dtype == RPO_DT_F32) return launch_norm_bwd<float>(dy, x_new, weight, rstd, dres, dx_out, dw_partial, rows, d, nw, st);
    return RPO_ERR_INVALID_ARG;
}
