// Shared device helpers for librankpo_hip (gfx950 / CDNA4 only: 64-wide waves, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rankpo_hip.h"

#define RPO_WAVE 64

typedef unsigned short bf16_t;  // raw bf16 bits
typedef _Float16 f16_t;         // IEEE half (the reference's fp16 arithmetic: configs/ds_zero1_config_bge.json:2-11, modeling.py:453-454 use_fp16)
typedef __attribute__((ext_vector_type(8))) _Float16 half8_t; // 8 f16 = one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(8))) short short8_t;   // 8 bf16 = one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float float4_t;   // 16x16 MFMA accumulator fragment
typedef __attribute__((ext_vector_type(4))) unsigned int uint4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int uint2_t;

// ---- bf16 <-> f32 ------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_t b) { return __uint_as_float(((unsigned)b) << 16); }

// Round to nearest even; NaN stays NaN (plain cast -> v_cvt_pk_bf16_f32 on gfx950).
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ float round_to_bf16(float f) { return bf16_to_f32(f32_to_bf16(f)); }

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int kVec = 4;  // elements per 16-byte access
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
    __device__ static __forceinline__ float round(float v) { return v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int kVec = 8;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
    __device__ static __forceinline__ float round(float v) { return round_to_bf16(v); }
};

template <> struct Elem<f16_t> {
    static constexpr int kVec = 8;
    __device__ static __forceinline__ float ld(const f16_t* p) { return (float)*p; }
    __device__ static __forceinline__ void st(f16_t* p, float v) { *p = (f16_t)v; }          // round to nearest even; overflow -> inf, as torch's .half()
    __device__ static __forceinline__ float round(float v) { return (float)(f16_t)v; }
};
// storage dtype code of a kernel's element type, and element size of a code (host side)
template <typename T> constexpr int rpo_dtype_of() { return sizeof(T) == 4 ? RPO_DT_F32 : (__is_same(T, bf16_t) ? RPO_DT_BF16 : RPO_DT_F16); }
static inline int rpo_elem_size(int dtype) { return dtype == RPO_DT_F32 ? 4 : 2; }
static inline bool rpo_dtype_ok(int dtype) { return dtype == RPO_DT_F32 || dtype == RPO_DT_BF16 || dtype == RPO_DT_F16; }

// 16-byte vector of T held as f32 in registers
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    float v[4];
    __device__ __forceinline__ void load(const float* p) {
        float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    __device__ __forceinline__ void store(float* p) const {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
    // streaming (non-temporal) forms for data touched once
    __device__ __forceinline__ void load_nt(const float* p) {
        const uint4_t t = __builtin_nontemporal_load(reinterpret_cast<const uint4_t*>(p));
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = __uint_as_float(t[i]);
    }
    __device__ __forceinline__ void store_nt(float* p) const {
        uint4_t t;
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = __float_as_uint(v[i]);
        __builtin_nontemporal_store(t, reinterpret_cast<uint4_t*>(p));
    }
};
template <> struct Vec16<bf16_t> {
    float v[8];
    __device__ __forceinline__ void load(const bf16_t* p) {
        uint4 t = *reinterpret_cast<const uint4*>(p);
        unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    __device__ __forceinline__ void store(bf16_t* p) const {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            w[i] = (unsigned)f32_to_bf16(v[2 * i]) | ((unsigned)f32_to_bf16(v[2 * i + 1]) << 16);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    __device__ __forceinline__ void load_nt(const bf16_t* p) {
        const uint4_t t = __builtin_nontemporal_load(reinterpret_cast<const uint4_t*>(p));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(t[i] << 16);
            v[2 * i + 1] = __uint_as_float(t[i] & 0xffff0000u);
        }
    }
    __device__ __forceinline__ void store_nt(bf16_t* p) const {
        uint4_t t;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            t[i] = (unsigned)f32_to_bf16(v[2 * i]) | ((unsigned)f32_to_bf16(v[2 * i + 1]) << 16);
        __builtin_nontemporal_store(t, reinterpret_cast<uint4_t*>(p));
    }
};

template <> struct Vec16<f16_t> {
    float v[8];
    __device__ __forceinline__ void unpack(const uint4_t t) {
        const half8_t h = __builtin_bit_cast(half8_t, t);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)h[i];
    }
    __device__ __forceinline__ uint4_t pack() const {
        half8_t h;
#pragma unroll
        for (int i = 0; i < 8; ++i) h[i] = (f16_t)v[i];
        return __builtin_bit_cast(uint4_t, h);
    }
    __device__ __forceinline__ void load(const f16_t* p) { unpack(*reinterpret_cast<const uint4_t*>(p)); }
    __device__ __forceinline__ void store(f16_t* p) const { *reinterpret_cast<uint4_t*>(p) = pack(); }
    __device__ __forceinline__ void load_nt(const f16_t* p) { unpack(__builtin_nontemporal_load(reinterpret_cast<const uint4_t*>(p))); }
    __device__ __forceinline__ void store_nt(f16_t* p) const { __builtin_nontemporal_store(pack(), reinterpret_cast<uint4_t*>(p)); }
};

__device__ __forceinline__ bool rpo_aligned16_dev(const void* p) {
    return (reinterpret_cast<uintptr_t>(p) & 15) == 0;
}

// max of MFMA results without hipcc's canonicalising `v_max_f32 x, x, x` in front of every fmaxf operand (an MFMA output is
// not known to be a quiet number to the compiler: 32 extra VALU instructions per key tile in the attention forward, 490 in
// the 256 x 256 similarity kernel's epilogue).  NaN behaviour = the instruction's (IEEE maxNum: a quiet NaN operand is
// ignored), the same as fmaxf.
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float max2_raw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// the same on values hipcc KNOWS to be quiet numbers (MFMA results it has not lost track of): plain fmaxf, which then compiles to
// v_max3_f32 / v_max_f32 without the canonicalising instruction, stays visible to hipcc's hazard recognizer and scheduler, and
// does not cost the `s_nop 0` hipcc puts behind every asm statement whose result the next instruction reads.  The forward attention
// kernels' score maxima use these: an asm instruction as the FIRST reader of an MFMA result is not covered by hipcc's wait states
// for MFMA -> VALU reads (measured: 1 % slower than the asm form, two canonicalising instructions per chain survive; bit-identical)
__device__ __forceinline__ float max3_known(float a, float b, float c) {
    return __builtin_fmaxf(__builtin_fmaxf(a, b), c);
}
__device__ __forceinline__ float max2_known(float a, float b) {
    return __builtin_fmaxf(a, b);
}

// ---- wave / block reductions (64-wide waves) -----------------------------------------------------
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}
__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
    return x;
}
// Sum over a block of NW waves; every thread gets the total.  `scratch` holds >= NW floats.
// Fixed summation order -> bitwise reproducible.
template <int NW>
__device__ __forceinline__ float block_sum(float x, float* scratch) {
    x = wave_sum(x);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[w] = x;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) t += scratch[i];
    return t;
}

// ---- host-side helpers
// hipGetLastError() is sticky per host thread and also reports benign codes left behind by OTHER users of the
// runtime (e.g. hipErrorNotReady from PyTorch's event queries): clear it right before our launch so that the
// status we return describes this launch only.
#define RPO_LAUNCH(...)                   \
    do {                                  \
        (void)hipGetLastError();          \
        hipLaunchKernelGGL(__VA_ARGS__);  \
    } while (0)
extern thread_local int rpo_tls_last_hip_error;   // defined in infonce.hip; read by rpo_last_hip_error()
static inline int rpo_launch_status() {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return RPO_OK;
    rpo_tls_last_hip_error = (int)e;
    return RPO_ERR_LAUNCH;
}
static inline bool rpo_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline int64_t rpo_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
