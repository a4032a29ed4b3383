// Probe: lane -> element mapping of ds_read_b64_tr_b16 on gfx950 (feeds the MFMA B-operand recipe of DESIGN.md).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(2))) unsigned int u2;
__global__ void k(unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 64];   // [row][col], value = row*64+col
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x;
    // within each 16-lane group: lane 4q+p supplies the address of row q, columns 4p..4p+3 of a 4x16 block
    const int grp = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int row = grp * 4 + q, col = 4 * p;              // group g: block rows 4g..4g+3, cols 0..15
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned short*)(&lds[row * 64 + col]);
    u2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[lane * 4 + 0] = (unsigned short)(v[0] & 0xffff);
    out[lane * 4 + 1] = (unsigned short)(v[0] >> 16);
    out[lane * 4 + 2] = (unsigned short)(v[1] & 0xffff);
    out[lane * 4 + 3] = (unsigned short)(v[1] >> 16);
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int e = 0; e < 4; ++e) printf(" (r%d,c%d)", h[l * 4 + e] / 64, h[l * 4 + e] % 64);
        printf("\n");
    }
    return 0;
}
