// Shared device/host helpers for libocta_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/octa_hip.h"

typedef unsigned short bf16_t;  // raw bf16 bits

// ---------------------------------------------------------------- error reporting
void octa_set_error(const char* fmt, ...);
#define OCTA_FAIL(code, ...) do { octa_set_error(__VA_ARGS__); return (code); } while (0)
#define OCTA_REQUIRE(cond, ...) do { if (!(cond)) { octa_set_error(__VA_ARGS__); return OCTA_ERR_BAD_ARG; } } while (0)
#define OCTA_CHECK_LAUNCH(name) do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) { \
    octa_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); return OCTA_ERR_LAUNCH; } } while (0)

bool octa_wgrad_fold_begin(hipStream_t st, float* ws, int64_t ws_floats);      // conv.hip: fold session of the partial-store weight gradients over the CALLER's scratch (true: this call opened it)
int octa_rev_walk();                             // api.cpp: first-pass reductions walk their tensor end first (octa_tuning_set(7, .))
bool octa_deterministic();                        // api.cpp: octa_tuning_set(5, 1) / OCTA_DETERMINISTIC=1: every cross-workgroup sum in a fixed order
int octa_wgrad_fold_end();
float* octa_wgrad_fold_reserve(hipStream_t st, float* dw, float* dbias, const int64_t* strides, int Ntot, int Kpad, int Cg, int CgReal, int KW,
                               int split, int64_t* slice_out);
void octa_note_conv_kernel(const char* name);   // conv.hip: name reported by octa_last_conv_kernel()

// ---------------------------------------------------------------- in-kernel clock stamps (DIAGNOSTIC build only)
// build.sh diag compiles the library once more with -DOCTA_DIAG_STAMPS into libocta_hip_diag.so: the three MFMA-bound kernels then
// stamp s_memtime (shader cycles) and s_memrealtime (100 MHz) around their main loop into a __device__ array of the code object that
// nothing else reads (MI355X_MICROARCH.md, DVFS give-back item 6: in-kernel clock = d memtime / d memrealtime x 100 MHz).  The
// shipped library contains no stamp: the macros expand to nothing.
#ifdef OCTA_DIAG_STAMPS
#define OCTA_STAMP_DECL unsigned long long stamp_t0 = 0, stamp_r0 = 0
#define OCTA_STAMP_BEGIN asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_t0), "=s"(stamp_r0) :: "memory")
#define OCTA_STAMP_END(ARR)                                                                                                   \
    {                                                                                                                         \
        unsigned long long t1_, r1_;                                                                                          \
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1_), "=s"(r1_) :: "memory");          \
        if (threadIdx.x == 0) {                                                                                               \
            unsigned long long* d_ = ARR[blockIdx.x & 4095];                                                                  \
            d_[0] = stamp_t0; d_[1] = stamp_r0; d_[2] = t1_; d_[3] = r1_;                                                     \
        }                                                                                                                     \
    }
#else
#define OCTA_STAMP_DECL
#define OCTA_STAMP_BEGIN
#define OCTA_STAMP_END(ARR)
#endif

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------- bf16 <-> f32
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
// round-to-nearest-even, NaN stays NaN: gfx950's v_cvt_pk_bf16_f32 (the bit-twiddled form cost 6 VALU ops per element in every
// epilogue; same results for every finite input)
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
typedef __bf16 octa_bf2_t __attribute__((ext_vector_type(2)));
typedef float octa_f2_t __attribute__((ext_vector_type(2)));

struct f16_t { unsigned short v; };   // raw IEEE half bits; a distinct type (bf16_t is a plain unsigned short) so that templates can specialise
__device__ __forceinline__ float h2f(unsigned short v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ unsigned short f2h(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }   // round-to-nearest-even, saturates to inf

template <typename T> struct DT;
template <> struct DT<float> {
    static constexpr int EPC = 4;  // elements per 16-byte chunk
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct DT<bf16_t> {
    static constexpr int EPC = 8;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

template <> struct DT<f16_t> {
    static constexpr int EPC = 8;
    __device__ static __forceinline__ float ld(const f16_t* p) { return h2f(p->v); }
    __device__ static __forceinline__ void st(f16_t* p, float v) { p->v = f2h(v); }
};
// two fp32 values -> one dword of two 16-bit elements of T
template <typename T> __device__ __forceinline__ unsigned pack2(float a, float b);
template <> __device__ __forceinline__ unsigned pack2<bf16_t>(float a, float b) {
    const octa_f2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, octa_bf2_t));
}
template <> __device__ __forceinline__ unsigned pack2<f16_t>(float a, float b) { return (unsigned)f2h(a) | ((unsigned)f2h(b) << 16); }
template <typename T> struct DTName;
template <> struct DTName<float> { static constexpr const char* v = "f32"; };
template <> struct DTName<bf16_t> { static constexpr const char* v = "bf16"; };
template <> struct DTName<f16_t> { static constexpr const char* v = "f16"; };

// unpack a 16-byte chunk into EPC floats / pack back
template <typename T> __device__ __forceinline__ void unpack16(const uint4& c, float* f);
template <> __device__ __forceinline__ void unpack16<float>(const uint4& c, float* f) {
    f[0] = __uint_as_float(c.x); f[1] = __uint_as_float(c.y); f[2] = __uint_as_float(c.z); f[3] = __uint_as_float(c.w);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const uint4& c, float* f) {
    f[0] = __uint_as_float(c.x << 16); f[1] = __uint_as_float(c.x & 0xffff0000u);
    f[2] = __uint_as_float(c.y << 16); f[3] = __uint_as_float(c.y & 0xffff0000u);
    f[4] = __uint_as_float(c.z << 16); f[5] = __uint_as_float(c.z & 0xffff0000u);
    f[6] = __uint_as_float(c.w << 16); f[7] = __uint_as_float(c.w & 0xffff0000u);
}
template <> __device__ __forceinline__ void unpack16<f16_t>(const uint4& c, float* f) {
    f[0] = h2f((unsigned short)(c.x & 0xffffu)); f[1] = h2f((unsigned short)(c.x >> 16));
    f[2] = h2f((unsigned short)(c.y & 0xffffu)); f[3] = h2f((unsigned short)(c.y >> 16));
    f[4] = h2f((unsigned short)(c.z & 0xffffu)); f[5] = h2f((unsigned short)(c.z >> 16));
    f[6] = h2f((unsigned short)(c.w & 0xffffu)); f[7] = h2f((unsigned short)(c.w >> 16));
}
template <typename T> __device__ __forceinline__ uint4 pack16(const float* f);
template <> __device__ __forceinline__ uint4 pack16<float>(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
}
template <> __device__ __forceinline__ uint4 pack16<bf16_t>(const float* f) {
    uint4 c;
    c.x = pack2<bf16_t>(f[0], f[1]);
    c.y = pack2<bf16_t>(f[2], f[3]);
    c.z = pack2<bf16_t>(f[4], f[5]);
    c.w = pack2<bf16_t>(f[6], f[7]);
    return c;
}

template <> __device__ __forceinline__ uint4 pack16<f16_t>(const float* f) {
    return make_uint4(pack2<f16_t>(f[0], f[1]), pack2<f16_t>(f[2], f[3]), pack2<f16_t>(f[4], f[5]), pack2<f16_t>(f[6], f[7]));
}
// three-way dtype dispatch: `using T` inside the braces
#define OCTA_DISPATCH3(dtype, NAME, ...)                                                     \
    if ((dtype) == OCTA_F32) { using T = float; __VA_ARGS__ }                                \
    else if ((dtype) == OCTA_BF16) { using T = bf16_t; __VA_ARGS__ }                         \
    else if ((dtype) == OCTA_F16) { using T = f16_t; __VA_ARGS__ }                           \
    else OCTA_FAIL(OCTA_ERR_BAD_ARG, NAME ": bad dtype %d", (int)(dtype));
#define OCTA_DTYPE_OK(dtype) ((dtype) == OCTA_F32 || (dtype) == OCTA_BF16 || (dtype) == OCTA_F16)

// ---------------------------------------------------------------- reductions (wave = 64)
// The four steps inside a 16-lane row are DPP moves (VALU rate); only the two cross-row steps go through the LDS crossbar
// (ds_bpermute, what __shfl_xor compiles to - six of those per sum made the small reduction kernels latency-bound).
__device__ __forceinline__ float wave_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]: lane ^ 1
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]: lane ^ 2
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x124, 0xF, 0xF, true));   // row_ror:4
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x128, 0xF, 0xF, true));   // row_ror:8
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
template <int CTRL> __device__ __forceinline__ double octa_dpp_d(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)u, CTRL, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double wave_sum_d(double v) {
    v += octa_dpp_d<0xB1>(v);
    v += octa_dpp_d<0x4E>(v);
    v += octa_dpp_d<0x124>(v);
    v += octa_dpp_d<0x128>(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
// block-wide sum of NV values per thread; result valid in thread 0 (and broadcast via smem[0..NV))
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* smem /* >= NV * 16 floats */) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) smem[i * 16 + wid] = v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float s = 0.f;
            for (int w = 0; w < nw; ++w) s += smem[i * 16 + w];
            v[i] = s;
        }
    }
}

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case OCTA_ACT_RELU: return v > 0.f ? v : 0.f;
        case OCTA_ACT_LEAKY02: return v > 0.f ? v : 0.2f * v;
        case OCTA_ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
        case OCTA_ACT_TANH: return tanhf(v);
        default: return v;
    }
}

// Activation of a whole accumulator tile behind ONE decision.  act_apply() inside the fully unrolled epilogue loops costs a
// scalar compare-and-branch chain PER ELEMENT (the switch is on a kernel argument; nothing hoists it out of straight-line
// code): ~3,000 SALU instructions per tile of the persistent conv kernel, more time than its MFMAs.
template <typename V4, int NI, int NJ>
__device__ __forceinline__ void act_tile(V4 (&acc)[NI][NJ], int act) {
    if (act == OCTA_ACT_NONE) return;
#define OCTA_ACT_TILE(EXPR)                                                                  \
    _Pragma("unroll") for (int i = 0; i < NI; ++i) _Pragma("unroll") for (int j = 0; j < NJ; ++j) \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) { const float v = acc[i][j][e]; acc[i][j][e] = (EXPR); }
    if (act == OCTA_ACT_RELU) { OCTA_ACT_TILE(v > 0.f ? v : 0.f) }
    else if (act == OCTA_ACT_LEAKY02) { OCTA_ACT_TILE(v > 0.f ? v : 0.2f * v) }
    else if (act == OCTA_ACT_SIGMOID) { OCTA_ACT_TILE(1.f / (1.f + __expf(-v))) }
    else if (act == OCTA_ACT_TANH) { OCTA_ACT_TILE(tanhf(v)) }
#undef OCTA_ACT_TILE
}

// Stream-ordered zero fill as a KERNEL.  hipMemsetAsync is not used anywhere in this library: captured into a hipGraph it
// becomes a memset node, and memset nodes were observed (ROCm 7.2, gfx950) to lose their ordering against the neighbouring
// kernel nodes when graphs are replayed back-to-back without host synchronisation - accumulators were cleared AFTER the
// atomics that fill them (spectral-norm v = 0 -> 0/0).  A kernel node keeps the stream order.
__global__ static void octa_zero_words_kernel(uint32_t* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static inline hipError_t octa_zero_async(void* p, size_t bytes /* multiple of 4 */, hipStream_t st) {
    const size_t n = bytes / 4;
    if (n == 0) return hipSuccess;
    size_t nb = (n + 255) / 256;
    if (nb > 2048) nb = 2048;
    octa_zero_words_kernel<<<(unsigned)nb, 256, 0, st>>>((uint32_t*)p, n);
    return hipGetLastError();
}
