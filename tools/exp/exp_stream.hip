// Scratch experiment (not part of the library): which streaming idioms get closest to the HBM roofline on gfx950 for
// a 2-read / 1-write bf16 elementwise kernel (the SwiGLU forward shape).  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) unsigned int u4;

__device__ __forceinline__ float sig(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ u4 op(u4 a, u4 b) {
    u4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float a0 = __uint_as_float(a[i] << 16), a1 = __uint_as_float(a[i] & 0xffff0000u);
        float b0 = __uint_as_float(b[i] << 16), b1 = __uint_as_float(b[i] & 0xffff0000u);
        float r0 = a0 * sig(a0) * b0, r1 = a1 * sig(a1) * b1;
        __bf16 h0 = (__bf16)r0, h1 = (__bf16)r1;
        o[i] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
    }
    return o;
}

template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k(const u4* __restrict__ g, const u4* __restrict__ u, u4* __restrict__ o, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride * U) {
        u4 a[U], b[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int64_t idx = i + j * stride;
            if (idx < n) {
                a[j] = NTL ? __builtin_nontemporal_load(g + idx) : g[idx];
                b[j] = NTL ? __builtin_nontemporal_load(u + idx) : u[idx];
            }
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int64_t idx = i + j * stride;
            if (idx < n) {
                u4 r = op(a[j], b[j]);
                if (NTS) __builtin_nontemporal_store(r, o + idx); else o[idx] = r;
            }
        }
    }
}

template <int U, bool NTL, bool NTS>
void run(const char* name, const u4* g, const u4* u, u4* o, int64_t n, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<U, NTL, NTS>), dim3(blocks), dim3(256), 0, 0, g, u, o, n);
    hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<U, NTL, NTS>), dim3(blocks), dim3(256), 0, 0, g, u, o, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("%-28s blocks %6d  %8.1f us  %7.1f GB/s\n", name, blocks, ms * 1e3, 3.0 * n * 16 / ms / 1e6);
}

// contiguous chunk per block: block b owns vectors [b*256*U, (b+1)*256*U)
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void kc(const u4* __restrict__ g, const u4* __restrict__ u, u4* __restrict__ o, int64_t n) {
    const int64_t base = (int64_t)blockIdx.x * 256 * U + threadIdx.x;
    u4 a[U], b[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const int64_t idx = base + j * 256;
        if (idx < n) {
            a[j] = NTL ? __builtin_nontemporal_load(g + idx) : g[idx];
            b[j] = NTL ? __builtin_nontemporal_load(u + idx) : u[idx];
        }
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const int64_t idx = base + j * 256;
        if (idx < n) {
            u4 r = op(a[j], b[j]);
            if (NTS) __builtin_nontemporal_store(r, o + idx); else o[idx] = r;
        }
    }
}
template <int U, bool NTL, bool NTS>
void runc(const char* name, const u4* g, const u4* u, u4* o, int64_t n) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int64_t blocks = (n + 256 * U - 1) / (256 * U);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((kc<U, NTL, NTS>), dim3((unsigned)blocks), dim3(256), 0, 0, g, u, o, n);
    hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((kc<U, NTL, NTS>), dim3((unsigned)blocks), dim3(256), 0, 0, g, u, o, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("%-28s blocks %7lld  %8.1f us  %7.1f GB/s\n", name, (long long)blocks, ms * 1e3, 3.0 * n * 16 / ms / 1e6);
}

int main() {
    const int64_t n = 150000LL * 8192 / 8;   // 16-byte vectors
    u4 *g, *u, *o;
    hipMalloc(&g, n * 16); hipMalloc(&u, n * 16); hipMalloc(&o, n * 16);
    hipMemset(g, 0x3c, n * 16); hipMemset(u, 0x3d, n * 16);
    runc<1, false, false>("chunk U1", g, u, o, n);
    runc<1, true, true>("chunk U1 nt", g, u, o, n);
    runc<2, false, false>("chunk U2", g, u, o, n);
    runc<2, true, true>("chunk U2 nt", g, u, o, n);
    runc<4, false, false>("chunk U4", g, u, o, n);
    runc<4, true, true>("chunk U4 nt", g, u, o, n);
    runc<4, true, false>("chunk U4 nt-load", g, u, o, n);
    runc<8, false, false>("chunk U8", g, u, o, n);
    runc<8, true, true>("chunk U8 nt", g, u, o, n);
    for (int blocks : {65536, 262144}) {
        run<1, false, false>("U1", g, u, o, n, blocks);
        run<1, false, true>("U1 nt-store", g, u, o, n, blocks);
        run<1, true, true>("U1 nt-load nt-store", g, u, o, n, blocks);
        run<2, false, false>("U2", g, u, o, n, blocks);
        run<2, true, true>("U2 nt-load nt-store", g, u, o, n, blocks);
        run<4, false, true>("U4 nt-store", g, u, o, n, blocks);
        run<4, true, true>("U4 nt-load nt-store", g, u, o, n, blocks);
    }
    return 0;
}
