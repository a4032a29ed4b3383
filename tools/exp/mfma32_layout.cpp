// Probe of v_mfma_f32_32x32x16_bf16's operand layouts on gfx950 (build: hipcc --offload-arch=gfx950 -O2 -o mfma32_layout mfma32_layout.cpp).
// Assumed: A (32 x 16): lane l holds row l % 32, k = 8 (l / 32) + e (e = 0..7, 4 VGPRs); B (16 x 32): lane l holds column l % 32, the
// same k; D (32 x 32 f32, 16 VGPRs): register r of lane l = row 8 (r / 4) + 4 (l / 32) + r % 4, column l % 32.  One-hot probes.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f16v;
__global__ void probe(int I0, int K0, int J0, float* out) {
    const int l = threadIdx.x;
    bf8 a, b;
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * (l / 32) + e;
        a[e] = (__bf16)((l % 32 == I0 && k == K0) ? 1.0f : 0.0f);
        b[e] = (__bf16)((l % 32 == J0 && k == K0) ? 1.0f : 0.0f);
    }
    f16v c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[l * 16 + r] = c[r];
}
int main() {
    float* d; hipMalloc(&d, 64 * 16 * 4);
    float h[64 * 16];
    int bad = 0, n = 0;
    for (int I0 = 0; I0 < 32; I0 += 5)
        for (int K0 = 0; K0 < 16; K0 += 3)
            for (int J0 = 0; J0 < 32; J0 += 7) {
                hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, I0, K0, J0, d);
                hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
                for (int l = 0; l < 64; ++l)
                    for (int r = 0; r < 16; ++r) {
                        const int row = 8 * (r / 4) + 4 * (l / 32) + r % 4, col = l % 32;
                        const float want = (row == I0 && col == J0) ? 1.0f : 0.0f;
                        if (h[l * 16 + r] != want) { if (bad < 5) printf("I0 %d K0 %d J0 %d: lane %d reg %d = %g want %g\n", I0, K0, J0, l, r, h[l * 16 + r], want); ++bad; }
                        ++n;
                    }
            }
    printf("mfma 32x32x16 bf16 layout probe: %d values checked, %d wrong\n", n, bad);
    return bad != 0;
}
