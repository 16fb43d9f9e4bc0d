// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the Show-and-Tell hot path.
// wave = 64 lanes; MFMA fragment maps follow the CDNA4 ISA (32x32x16 bf16 / 32x32x2 f32 / 16x16x4 f32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SAT_OK 0
#define SAT_ERR_ARG 1001          // bad shape / alignment / null pointer
#define SAT_ERR_WORKSPACE 1002    // workspace too small
#define SAT_ERR_UNSUPPORTED 1003

#define SAT_LAUNCH_CHECK()                         \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

typedef __bf16 bf16_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

enum { SAT_F32 = 0, SAT_BF16 = 1 };

static inline int sat_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// accurate (ocml) exp/tanh: these sit on the serial LSTM chain and in CE, never on a throughput path
__device__ __forceinline__ float sat_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float sat_tanh(float x) { return tanhf(x); }

// 16-byte global store, write-through to memory at system scope (sc0 sc1).  For a kernel's bulk OUTPUT that the NEXT launch reads:
// every XCD has its own L2 and a launch ends with a write-back of the dirty lines its workgroups left there; lines that went
// through already shorten that tail (conv output tiles: 19.8 -> 18.7 us for the 25.7 MB of a layer-3 conv3, -2 % on the step).
__device__ __forceinline__ void store16_wt(void* dst, u32x4 v) {
    // s_nop: a VMEM store of more than 64 bits needs wait states before its data VGPRs may be overwritten; the compiler's hazard
    // recognizer does not look inside inline asm
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }
