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

template <typename T>
__global__ __launch_bounds__(kEwThreads) void swiglu_fwd_kernel(const T* __restrict__ g, const T* __restrict__ u,
                                                                T* __restrict__ out, int64_t nvec) {
    // ONE 16-byte vector per thread, block b owns the contiguous vectors [256 b, 256 b + 256), streaming
    // (non-temporal) loads and stores: measured 6.6 TB/s on MI355X vs 4.6-4.9 TB/s for a capped grid-stride loop
    // (tools/exp/exp_stream.hip).
    constexpr int V = Elem<T>::kVec;
    const int64_t i = (int64_t)blockIdx.x * kEwThreads + threadIdx.x;
    if (i >= nvec) return;
    Vec16<T> a, b, o;
    a.load_nt(g + i * V);
    b.load_nt(u + i * V);
#pragma unroll
    for (int k = 0; k < V; ++k) o.v[k] = a.v[k] * sigmoid_f(a.v[k]) * b.v[k];
    o.store_nt(out + i * V);
}

template <typename T>
__global__ __launch_bounds__(kEwThreads) void swiglu_bwd_kernel(const T* g, const T* u, const T* dout, T* dg, T* du,
                                                                int64_t nvec) {   // dg / du may alias g / u / dout
    constexpr int V = Elem<T>::kVec;
    const int64_t i = (int64_t)blockIdx.x * kEwThreads + threadIdx.x;
    if (i >= nvec) return;
    Vec16<T> a, b, d, og, ou;
    a.load_nt(g + i * V);
    b.load_nt(u + i * V);
    d.load_nt(dout + i * V);
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const float s = sigmoid_f(a.v[k]);
        const float silu = a.v[k] * s;
        og.v[k] = d.v[k] * b.v[k] * (s + silu * (1.0f - s));   // silu' = s (1 + g (1 - s))
        ou.v[k] = d.v[k] * silu;
    }
    og.store_nt(dg + i * V);
    ou.store_nt(du + i * V);
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
    const int total = H * vec_per_head;
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        T* xr = x + r * row_stride;
        const T* xi = xin + r * row_stride;
        const float* c = cs + (r % period) * half;
        const float* s = sn + (r % period) * half;
        for (int i = threadIdx.x; i < total; i += kEwThreads) {
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

inline unsigned ew_grid(int64_t nvec) { return (unsigned)rpo_cdiv(nvec, kEwThreads); }

}  // namespace

extern "C" int rpo_swiglu_fwd(const void* g, const void* u, void* out, int64_t n, int dtype, rpo_stream_t stream) {
    if (!g || !u || !out || n <= 0) return RPO_ERR_INVALID_ARG;
    const int V = dtype == RPO_DT_BF16 ? 8 : 4;
    if (n % V != 0 || !rpo_aligned16(g) || !rpo_aligned16(u) || !rpo_aligned16(out)) return RPO_ERR_UNSUPPORTED;
    if (n / V / kEwThreads >= INT32_MAX) return RPO_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == RPO_DT_BF16)
        RPO_LAUNCH(swiglu_fwd_kernel<bf16_t>, dim3(ew_grid(n / V)), dim3(kEwThreads), 0, st, (const bf16_t*)g,
                   (const bf16_t*)u, (bf16_t*)out, n / V);
    else if (dtype == RPO_DT_F32)
        RPO_LAUNCH(swiglu_fwd_kernel<float>, dim3(ew_grid(n / V)), dim3(kEwThreads), 0, st, (const float*)g,
                   (const float*)u, (float*)out, n / V);
    else
        return RPO_ERR_INVALID_ARG;
    return rpo_launch_status();
}

extern "C" int rpo_swiglu_bwd(const void* g, const void* u, const void* dout, void* dg, void* du, int64_t n,
                              int dtype, rpo_stream_t stream) {
    if (!g || !u || !dout || !dg || !du || n <= 0) return RPO_ERR_INVALID_ARG;
    const int V = dtype == RPO_DT_BF16 ? 8 : 4;
    if (n % V != 0 || !rpo_aligned16(g) || !rpo_aligned16(u) || !rpo_aligned16(dout) || !rpo_aligned16(dg) ||
        !rpo_aligned16(du))
        return RPO_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == RPO_DT_BF16)
        RPO_LAUNCH(swiglu_bwd_kernel<bf16_t>, dim3(ew_grid(n / V)), dim3(kEwThreads), 0, st, (const bf16_t*)g,
                   (const bf16_t*)u, (const bf16_t*)dout, (bf16_t*)dg, (bf16_t*)du, n / V);
    else if (dtype == RPO_DT_F32)
        RPO_LAUNCH(swiglu_bwd_kernel<float>, dim3(ew_grid(n / V)), dim3(kEwThreads), 0, st, (const float*)g,
                   (const float*)u, (const float*)dout, (float*)dg, (float*)du, n / V);
    else
        return RPO_ERR_INVALID_ARG;
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
    int64_t grid = rows < INT32_MAX ? rows : INT32_MAX;   // one block per row: contiguous 2*heads*head_dim bytes
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
