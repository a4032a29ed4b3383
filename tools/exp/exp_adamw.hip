// Scratch experiment: access idioms for the 7-stream AdamW update (bf16 param/grad, f32 master/m/v).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) unsigned int u4;
typedef __attribute__((ext_vector_type(2))) unsigned int u2;

template <bool NT> __device__ __forceinline__ u4 ld4(const u4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st4(u4* p, u4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
template <bool NT> __device__ __forceinline__ u2 ld2(const u2* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st2(u2* p, u2 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

template <bool NTL, bool NTS>
__device__ __forceinline__ void upd(int64_t i, unsigned short* param, float* master, const unsigned short* grad, float* m, float* v) {
    u2 gb = ld2<NTL>((const u2*)(grad + 4 * i));
    u4 mm = ld4<NTL>((const u4*)(m + 4 * i)), vv = ld4<NTL>((const u4*)(v + 4 * i)), ww = ld4<NTL>((const u4*)(master + 4 * i));
    float g[4] = {__uint_as_float(gb[0] << 16), __uint_as_float(gb[0] & 0xffff0000u), __uint_as_float(gb[1] << 16), __uint_as_float(gb[1] & 0xffff0000u)};
    u4 om, ov, ow; unsigned short pb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float w = __uint_as_float(ww[k]), a = __uint_as_float(mm[k]), b = __uint_as_float(vv[k]);
        a = 0.9f * a + 0.1f * g[k]; b = 0.999f * b + 0.001f * g[k] * g[k];
        w -= 1e-5f * (a / (sqrtf(b) + 1e-8f));
        om[k] = __float_as_uint(a); ov[k] = __float_as_uint(b); ow[k] = __float_as_uint(w);
        __bf16 h = (__bf16)w; pb[k] = __builtin_bit_cast(unsigned short, h);
    }
    st4<NTS>((u4*)(m + 4 * i), om); st4<NTS>((u4*)(v + 4 * i), ov); st4<NTS>((u4*)(master + 4 * i), ow);
    u2 po = {(unsigned)pb[0] | ((unsigned)pb[1] << 16), (unsigned)pb[2] | ((unsigned)pb[3] << 16)};
    st2<NTS>((u2*)(param + 4 * i), po);
}

template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_stride(unsigned short* param, float* master, const unsigned short* grad, float* m, float* v, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) upd<NTL, NTS>(i, param, master, grad, m, v);
}
template <bool NTL, bool NTS, int U>
__global__ __launch_bounds__(256) void k_chunk(unsigned short* param, float* master, const unsigned short* grad, float* m, float* v, int64_t n4) {
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const int64_t i = ((int64_t)blockIdx.x * U + j) * 256 + threadIdx.x;
        if (i < n4) upd<NTL, NTS>(i, param, master, grad, m, v);
    }
}

#define TIME(name, ...)                                                                  \
    do {                                                                                    \
        for (int w_ = 0; w_ < 2; ++w_) { __VA_ARGS__; }                                          \
        hipEventRecord(e0);                                                                 \
        for (int r_ = 0; r_ < 5; ++r_) { __VA_ARGS__; }                                          \
        hipEventRecord(e1); hipEventSynchronize(e1);                                        \
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;                                 \
        printf("%-34s %8.1f us  %7.1f GB/s\n", name, ms * 1e3, 28.0 * n / ms / 1e6);        \
    } while (0)

int main() {
    const int64_t n = 1235828736LL, n4 = n / 4;
    unsigned short *param, *grad; float *master, *m, *v;
    hipMalloc(&param, n * 2); hipMalloc(&grad, n * 2); hipMalloc(&master, n * 4); hipMalloc(&m, n * 4); hipMalloc(&v, n * 4);
    hipMemset(grad, 0x3c, n * 2); hipMemset(master, 0, n * 4); hipMemset(m, 0, n * 4); hipMemset(v, 0, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {2048, 8192, 65536}) {
        char nm[64];
        snprintf(nm, 64, "stride plain, %d blocks", blocks);
        TIME(nm, hipLaunchKernelGGL((k_stride<false, false>), dim3(blocks), dim3(256), 0, 0, param, master, grad, m, v, n4));
        snprintf(nm, 64, "stride nt-store, %d blocks", blocks);
        TIME(nm, hipLaunchKernelGGL((k_stride<false, true>), dim3(blocks), dim3(256), 0, 0, param, master, grad, m, v, n4));
        snprintf(nm, 64, "stride nt-both, %d blocks", blocks);
        TIME(nm, hipLaunchKernelGGL((k_stride<true, true>), dim3(blocks), dim3(256), 0, 0, param, master, grad, m, v, n4));
    }
    TIME("chunk U1 plain", hipLaunchKernelGGL((k_chunk<false, false, 1>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, param, master, grad, m, v, n4));
    TIME("chunk U1 nt-both", hipLaunchKernelGGL((k_chunk<true, true, 1>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, param, master, grad, m, v, n4));
    TIME("chunk U1 nt-load", hipLaunchKernelGGL((k_chunk<true, false, 1>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, param, master, grad, m, v, n4));
    TIME("chunk U2 plain", hipLaunchKernelGGL((k_chunk<false, false, 2>), dim3((unsigned)((n4 + 511) / 512)), dim3(256), 0, 0, param, master, grad, m, v, n4));
    TIME("chunk U2 nt-both", hipLaunchKernelGGL((k_chunk<true, true, 2>), dim3((unsigned)((n4 + 511) / 512)), dim3(256), 0, 0, param, master, grad, m, v, n4));
    TIME("chunk U4 plain", hipLaunchKernelGGL((k_chunk<false, false, 4>), dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, 0, param, master, grad, m, v, n4));
    TIME("chunk U4 nt-both", hipLaunchKernelGGL((k_chunk<true, true, 4>), dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, 0, param, master, grad, m, v, n4));
    return 0;
}
