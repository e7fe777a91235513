// Split-attention glue (radix 2) and the adversarial attention gate / 1x1 head.  HBM-bound streaming
// kernels: 16-byte chunks, wavefront shuffles for the per-pixel and per-channel sums.
#include "common.hpp"
#include <algorithm>
#include <stdlib.h>

// slab counts of the streaming passes (workgroups per sample along HW): swept in situ (round 5, tools/ab_envs.sh): fewer, longer
// workgroups amortise a workgroup's fixed cost (coefficient loads, block reduction, partial store)
static int splat_reduce_slabs() { static const int v = getenv("OCTA_SPLAT_SLABS") ? std::max(1, atoi(getenv("OCTA_SPLAT_SLABS"))) : 24; return v; }
static int splat_apply_slabs() { static const int v = getenv("OCTA_SPLAT_APPLY_SLABS") ? std::max(1, atoi(getenv("OCTA_SPLAT_APPLY_SLABS"))) : 128; return v; }
static int splat_dx_slabs(int B) { static const int v = getenv("OCTA_SPLAT_DX_SLABS") ? std::max(1, atoi(getenv("OCTA_SPLAT_DX_SLABS"))) : 64; return std::max(1, std::min(v, 1024 / std::max(1, B))); }

static inline int ew_blocks(int64_t n) { int64_t b = cdiv64(n, 256); return (int)(b > 131072 ? 131072 : (b < 1 ? 1 : b)); }   // see norm.hip

// =========================================================================================== SplAt
// gap[b][c] += (1/HW) sum_hw (x[b,hw,c] + x[b,hw,C+c]);  grid (colblocks, hw splits, B)
template <typename T>
__global__ __launch_bounds__(256) void splat_gap_kernel(const T* __restrict__ x, float* __restrict__ gap, int HW, int C, int TX, int rpb) {
    constexpr int EPC = DT<T>::EPC;
    extern __shared__ float red[];   // [RY][TX*EPC]
    const int RY = 256 / TX;
    const int cx = threadIdx.x % TX, ry = threadIdx.x / TX;
    const int col = blockIdx.x * TX + cx, cpr = C / EPC;
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * rpb, r1 = min(HW, r0 + rpb);
    float s[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s[e] = 0.f;
    if (col < cpr) {
        const T* base = x + (int64_t)b * HW * 2 * C + col * EPC;
        for (int r = r0 + ry; r < r1; r += RY) {
            float u[EPC], v[EPC];
            unpack16<T>(*(const uint4*)(base + (int64_t)r * 2 * C), u);
            unpack16<T>(*(const uint4*)(base + (int64_t)r * 2 * C + C), v);
#pragma unroll
            for (int e = 0; e < EPC; ++e) s[e] += u[e] + v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) red[((size_t)ry * TX + cx) * EPC + e] = s[e];
    __syncthreads();
    const float inv = 1.f / (float)HW;
    for (int ch = threadIdx.x; ch < TX * EPC; ch += 256) {
        const int c = blockIdx.x * TX * EPC + ch;
        if (c >= C) continue;
        float a = 0.f;
        for (int yy = 0; yy < RY; ++yy) a += red[(size_t)yy * TX * EPC + ch];
        atomicAdd(gap + (int64_t)b * C + c, a * inv);
    }
}
static void splat_map(int cpr, int& TX, int& gridx) {
    if (cpr >= 256) TX = 256; else if (256 % cpr == 0) TX = cpr; else TX = 64;
    gridx = cdiv(cpr, TX);
}
extern "C" int octa_splat_gap(const void* x, float* gap, int B, int HW, int C, int dtype, int prezeroed, octa_stream_t stream) {
    OCTA_REQUIRE(x && gap && B > 0 && HW > 0 && C % 8 == 0, "octa_splat_gap: bad arguments (C %% 8)");
    hipStream_t st = (hipStream_t)stream;
    if (!prezeroed && octa_zero_async(gap, (size_t)B * C * sizeof(float), st) != hipSuccess) OCTA_FAIL(OCTA_ERR_LAUNCH, "octa_splat_gap: memset failed");
    const int epc = dtype == OCTA_F32 ? 4 : 8;
    int TX, gx;
    splat_map(C / epc, TX, gx);
    const int RY = 256 / TX;
    int rpb = cdiv(HW, splat_reduce_slabs());
    if (rpb < RY * 8) rpb = RY * 8;
    if (octa_deterministic()) rpb = HW;                    // one workgroup per (sample, column block): one add per address, fixed order
    dim3 grid(gx, cdiv(HW, rpb), B);
    const size_t sh = (size_t)256 * epc * sizeof(float);
    if (dtype == OCTA_F32) splat_gap_kernel<float><<<grid, 256, sh, st>>>((const float*)x, gap, HW, C, TX, rpb);
    else if (dtype == OCTA_BF16) splat_gap_kernel<bf16_t><<<grid, 256, sh, st>>>((const bf16_t*)x, gap, HW, C, TX, rpb);
    else if (dtype == OCTA_F16) splat_gap_kernel<f16_t><<<grid, 256, sh, st>>>((const f16_t*)x, gap, HW, C, TX, rpb);
    else OCTA_FAIL(OCTA_ERR_BAD_ARG, "octa_splat_gap: bad dtype");
    OCTA_CHECK_LAUNCH("splat_gap");
    return OCTA_OK;
}

// out = a0*x[:, :C] + a1*x[:, C:]; grid (hw blocks, B); attention for the sample staged in LDS
template <typename T>
__global__ __launch_bounds__(256) void splat_apply_kernel(const T* __restrict__ x, const float* __restrict__ logits, T* __restrict__ out, int HW,
                                                          int C, int relu, int rpb) {
    constexpr int EPC = DT<T>::EPC;
    extern __shared__ float a0s[];   // [C]
    const int b = blockIdx.y;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float l0 = logits[(int64_t)b * 2 * C + c], l1 = logits[(int64_t)b * 2 * C + C + c];
        a0s[c] = 1.f / (1.f + expf(l1 - l0));
    }
    __syncthreads();
    const int cpr = C / EPC;
    const int r0 = blockIdx.x * rpb, r1 = min(HW, r0 + rpb);
    const int64_t total = (int64_t)(r1 - r0) * cpr;
    for (int64_t i = threadIdx.x; i < total; i += 256) {
        const int r = r0 + (int)(i / cpr);
        const int c0 = (int)(i % cpr) * EPC;
        float u[EPC], v[EPC];
        const T* px = x + ((int64_t)b * HW + r) * 2 * C + c0;
        unpack16<T>(*(const uint4*)px, u);
        unpack16<T>(*(const uint4*)(px + C), v);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float a0 = a0s[c0 + e];
            float o = a0 * u[e] + (1.f - a0) * v[e];
            if (relu) o = o > 0.f ? o : 0.f;
            u[e] = o;
        }
        *(uint4*)(out + ((int64_t)b * HW + r) * C + c0) = pack16<T>(u);
    }
}
extern "C" int octa_splat_apply(const void* x, const float* logits, void* out, int B, int HW, int C, int dtype, int relu,
                                octa_stream_t stream) {
    OCTA_REQUIRE(x && logits && out && C % 8 == 0 && C <= 8192, "octa_splat_apply: bad arguments");
    int rpb = cdiv(HW, splat_apply_slabs());
    if (rpb < 8) rpb = 8;
    dim3 grid(cdiv(HW, rpb), B);
    const size_t sh = (size_t)C * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == OCTA_F32) splat_apply_kernel<float><<<grid, 256, sh, st>>>((const float*)x, logits, (float*)out, HW, C, relu, rpb);
    else if (dtype == OCTA_BF16) splat_apply_kernel<bf16_t><<<grid, 256, sh, st>>>((const bf16_t*)x, logits, (bf16_t*)out, HW, C, relu, rpb);
    else if (dtype == OCTA_F16) splat_apply_kernel<f16_t><<<grid, 256, sh, st>>>((const f16_t*)x, logits, (f16_t*)out, HW, C, relu, rpb);
    else OCTA_FAIL(OCTA_ERR_BAD_ARG, "octa_splat_apply: bad dtype");
    OCTA_CHECK_LAUNCH("splat_apply");
    return OCTA_OK;
}

// phase 0a: da[b][r*C+c] += sum_hw dout' * x_r   (dout' masked by out > 0 when relu)
template <typename T>
__global__ __launch_bounds__(256) void splat_bwd_reduce_kernel(const T* __restrict__ dout, const T* __restrict__ x, const T* __restrict__ outp,
                                                               float* __restrict__ da, int HW, int C, int TX, int rpb, int relu) {
    constexpr int EPC = DT<T>::EPC;
    extern __shared__ float red[];   // [RY][TX*EPC][2]
    const int RY = 256 / TX;
    const int cx = threadIdx.x % TX, ry = threadIdx.x / TX;
    const int col = blockIdx.x * TX + cx, cpr = C / EPC;
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * rpb, r1 = min(HW, r0 + rpb);
    float s0[EPC], s1[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { s0[e] = 0.f; s1[e] = 0.f; }
    if (col < cpr) {
        for (int r = r0 + ry; r < r1; r += RY) {
            float d[EPC], o[EPC], u[EPC], v[EPC];
            const int64_t po = ((int64_t)b * HW + r) * C + col * EPC;
            unpack16<T>(*(const uint4*)(dout + po), d);
            if (relu) unpack16<T>(*(const uint4*)(outp + po), o);
            const T* px = x + ((int64_t)b * HW + r) * 2 * C + col * EPC;
            unpack16<T>(*(const uint4*)px, u);
            unpack16<T>(*(const uint4*)(px + C), v);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float dd = (relu && !(o[e] > 0.f)) ? 0.f : d[e];
                s0[e] += dd * u[e]; s1[e] += dd * v[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) { red[(((size_t)ry * TX + cx) * EPC + e) * 2] = s0[e]; red[(((size_t)ry * TX + cx) * EPC + e) * 2 + 1] = s1[e]; }
    __syncthreads();
    for (int ch = threadIdx.x; ch < TX * EPC; ch += 256) {
        const int c = blockIdx.x * TX * EPC + ch;
        if (c >= C) continue;
        float a = 0.f, bb = 0.f;
        for (int yy = 0; yy < RY; ++yy) { a += red[((size_t)yy * TX * EPC + ch) * 2]; bb += red[((size_t)yy * TX * EPC + ch) * 2 + 1]; }
        atomicAdd(da + (int64_t)b * 2 * C + c, a);
        atomicAdd(da + (int64_t)b * 2 * C + C + c, bb);
    }
}
// phase 0b: radix-2 softmax backward in place: dl0 = a0 a1 (da0 - da1), dl1 = -dl0
__global__ void splat_softmax_bwd_kernel(const float* __restrict__ logits, float* __restrict__ da, int B, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i % C;
    const float l0 = logits[(int64_t)b * 2 * C + c], l1 = logits[(int64_t)b * 2 * C + C + c];
    const float a0 = 1.f / (1.f + expf(l1 - l0));
    const float g = a0 * (1.f - a0) * (da[(int64_t)b * 2 * C + c] - da[(int64_t)b * 2 * C + C + c]);
    da[(int64_t)b * 2 * C + c] = g;
    da[(int64_t)b * 2 * C + C + c] = -g;
}
// phase 1: dx_r = a_r * dout' + dgap / HW
template <typename T>
__global__ __launch_bounds__(256) void splat_bwd_apply_kernel(const T* __restrict__ dout, const T* __restrict__ outp, const float* __restrict__ logits,
                                                              const float* __restrict__ dgap, T* __restrict__ dx, int HW, int C, int relu,
                                                              int rpb) {
    constexpr int EPC = DT<T>::EPC;
    extern __shared__ float sm[];   // a0[C], dg[C]
    float* a0s = sm;
    float* dgs = sm + C;
    const int b = blockIdx.y;
    const float inv = 1.f / (float)HW;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float l0 = logits[(int64_t)b * 2 * C + c], l1 = logits[(int64_t)b * 2 * C + C + c];
        a0s[c] = 1.f / (1.f + expf(l1 - l0));
        dgs[c] = dgap ? dgap[(int64_t)b * C + c] * inv : 0.f;
    }
    __syncthreads();
    const int cpr = C / EPC;
    const int r0 = blockIdx.x * rpb, r1 = min(HW, r0 + rpb);
    const int64_t total = (int64_t)(r1 - r0) * cpr;
    for (int64_t i = threadIdx.x; i < total; i += 256) {
        const int r = r0 + (int)(i / cpr);
        const int c0 = (int)(i % cpr) * EPC;
        const int64_t po = ((int64_t)b * HW + r) * C + c0;
        float d[EPC], o[EPC], u[EPC], v[EPC];
        unpack16<T>(*(const uint4*)(dout + po), d);
        if (relu) unpack16<T>(*(const uint4*)(outp + po), o);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float dd = (relu && !(o[e] > 0.f)) ? 0.f : d[e];
            const float a0 = a0s[c0 + e];
            u[e] = a0 * dd + dgs[c0 + e];
            v[e] = (1.f - a0) * dd + dgs[c0 + e];
        }
        T* px = dx + ((int64_t)b * HW + r) * 2 * C + c0;
        *(uint4*)px = pack16<T>(u);
        *(uint4*)(px + C) = pack16<T>(v);
    }
}
extern "C" int octa_splat_bwd(const void* dout, const void* x, const float* logits, const void* out, const float* dgap, void* dx,
                              float* dlogits, int B, int HW, int C, int dtype, int relu, int phase, int prezeroed, octa_stream_t stream) {
    OCTA_REQUIRE(dout && logits && C % 8 == 0 && C <= 4096, "octa_splat_bwd: bad arguments");
    OCTA_REQUIRE(!relu || out, "octa_splat_bwd: relu needs the forward output");
    OCTA_REQUIRE(OCTA_DTYPE_OK(dtype), "octa_splat_bwd: bad dtype");
    hipStream_t st = (hipStream_t)stream;
    const int epc = dtype == OCTA_F32 ? 4 : 8;
    if (phase == 0) {
        OCTA_REQUIRE(x && dlogits, "octa_splat_bwd(phase 0): null pointer");
        if (!prezeroed && octa_zero_async(dlogits, (size_t)B * 2 * C * sizeof(float), st) != hipSuccess) OCTA_FAIL(OCTA_ERR_LAUNCH, "octa_splat_bwd: memset failed");
        int TX, gx;
        splat_map(C / epc, TX, gx);
        const int RY = 256 / TX;
        int rpb = cdiv(HW, splat_reduce_slabs());
        if (rpb < RY * 8) rpb = RY * 8;
        if (octa_deterministic()) rpb = HW;
        dim3 grid(gx, cdiv(HW, rpb), B);
        const size_t sh = (size_t)256 * epc * 2 * sizeof(float);
        if (dtype == OCTA_F32) splat_bwd_reduce_kernel<float><<<grid, 256, sh, st>>>((const float*)dout, (const float*)x, (const float*)out, dlogits, HW, C, TX, rpb, relu);
        else if (dtype == OCTA_BF16) splat_bwd_reduce_kernel<bf16_t><<<grid, 256, sh, st>>>((const bf16_t*)dout, (const bf16_t*)x, (const bf16_t*)out, dlogits, HW, C, TX, rpb, relu);
        else splat_bwd_reduce_kernel<f16_t><<<grid, 256, sh, st>>>((const f16_t*)dout, (const f16_t*)x, (const f16_t*)out, dlogits, HW, C, TX, rpb, relu);
        OCTA_CHECK_LAUNCH("splat_bwd_reduce");
        splat_softmax_bwd_kernel<<<cdiv(B * C, 256), 256, 0, st>>>(logits, dlogits, B, C);
        OCTA_CHECK_LAUNCH("splat_softmax_bwd");
    } else {
        OCTA_REQUIRE(dx, "octa_splat_bwd(phase 1): null pointer");
        int rpb = cdiv(HW, splat_apply_slabs());
        if (rpb < 8) rpb = 8;
        dim3 grid(cdiv(HW, rpb), B);
        const size_t sh = (size_t)2 * C * sizeof(float);
        if (dtype == OCTA_F32) splat_bwd_apply_kernel<float><<<grid, 256, sh, st>>>((const float*)dout, (const float*)out, logits, dgap, (float*)dx, HW, C, relu, rpb);
        else if (dtype == OCTA_BF16) splat_bwd_apply_kernel<bf16_t><<<grid, 256, sh, st>>>((const bf16_t*)dout, (const bf16_t*)out, logits, dgap, (bf16_t*)dx, HW, C, relu, rpb);
        else splat_bwd_apply_kernel<f16_t><<<grid, 256, sh, st>>>((const f16_t*)dout, (const f16_t*)out, logits, dgap, (f16_t*)dx, HW, C, relu, rpb);
        OCTA_CHECK_LAUNCH("splat_bwd_apply");
    }
    return OCTA_OK;
}

// =========================================================================================== SplAt with bn0 + ReLU on the fly
// extra/resnest.py:99-103,106-138: conv -> bn0 -> relu -> split attention.  The unfused path materialises y = relu(bn0(x)) (a
// BatchNorm apply pass: read x, write y) and reads y twice (GAP, weighted sum); its backward materialises dy and runs the
// BatchNorm backward over it (write dy, read dy + x twice, write dx).  Here x is the RAW conv output and every kernel recomputes
// y = max(x * sc + sh, 0) (sc = gamma * invstd, sh = beta - mean * sc) in registers: 2 of 6.5 forward passes and 3 of 7 backward
// passes over the 2C-channel tensor disappear, y and dy are never stored.
template <int EPC>
__device__ __forceinline__ void bn_co(const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                      const float* __restrict__ beta, int ch, float (&sc)[EPC], float (&sh)[EPC]) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        const float k = gamma[ch + e] * invstd[ch + e];
        sc[e] = k;
        sh[e] = beta[ch + e] - mean[ch + e] * k;
    }
}
struct SplatBn { const float* mean; const float* invstd; const float* gamma; const float* beta; };

// gap[b][c] += (1/HW) sum_hw (y[b,hw,c] + y[b,hw,C+c]);  grid (colblocks, hw splits, B)
template <typename T>
__global__ __launch_bounds__(256) void splat_gap_bn_kernel(const T* __restrict__ x, SplatBn bn, float* __restrict__ gap, int HW, int C, int TX, int rpb) {
    constexpr int EPC = DT<T>::EPC;
    extern __shared__ float red[];   // [RY][TX*EPC]
    const int RY = 256 / TX;
    const int cx = threadIdx.x % TX, ry = threadIdx.x / TX;
    const int col = blockIdx.x * TX + cx, cpr = C / EPC;
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * rpb, r1 = min(HW, r0 + rpb);
    float s[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s[e] = 0.f;
    if (col < cpr) {
        float scu[EPC], shu[EPC], scv[EPC], shv[EPC];
        bn_co<EPC>(bn.mean, bn.invstd, bn.gamma, bn.beta, col * EPC, scu, shu);
        bn_co<EPC>(bn.mean, bn.invstd, bn.gamma, bn.beta, C + col * EPC, scv, shv);
        const T* base = x + (int64_t)b * HW * 2 * C + col * EPC;
        int r = r0 + ry;
        for (; r + RY < r1; r += 2 * RY) {               // two rows (four loads) in flight
            const uint4 qa = *(const uint4*)(base + (int64_t)r * 2 * C), qb = *(const uint4*)(base + (int64_t)r * 2 * C + C);
            const uint4 qc = *(const uint4*)(base + (int64_t)(r + RY) * 2 * C), qd = *(const uint4*)(base + (int64_t)(r + RY) * 2 * C + C);
            float u[EPC], v[EPC], u2[EPC], v2[EPC];
            unpack16<T>(qa, u); unpack16<T>(qb, v); unpack16<T>(qc, u2); unpack16<T>(qd, v2);
#pragma unroll
            for (int e = 0; e < EPC; ++e)
                s[e] += (fmaxf(fmaf(u[e], scu[e], shu[e]), 0.f) + fmaxf(fmaf(v[e], scv[e], shv[e]), 0.f)) +
                        (fmaxf(fmaf(u2[e], scu[e], shu[e]), 0.f) + fmaxf(fmaf(v2[e], scv[e], shv[e]), 0.f));
        }
        for (; r < r1; r += RY) {
            float u[EPC], v[EPC];
            unpack16<T>(*(const uint4*)(base + (int64_t)r * 2 * C), u);
            unpack16<T>(*(const uint4*)(base + (int64_t)r * 2 * C + C), v);
#pragma unroll
            for (int e = 0; e < EPC; ++e) s[e] += fmaxf(fmaf(u[e], scu[e], shu[e]), 0.f) + fmaxf(fmaf(v[e], scv[e], shv[e]), 0.f);
        }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) red[((size_t)ry * TX + cx) * EPC + e] = s[e];
    __syncthreads();
    const float inv = 1.f / (float)HW;
    for (int ch = threadIdx.x; ch < TX * EPC; ch += 256) {
        const int c = blockIdx.x * TX * EPC + ch;
        if (c >= C) continue;
        float a = 0.f;
        for (int yy = 0; yy < RY; ++yy) a += red[(size_t)yy * TX * EPC + ch];
        atomicAdd(gap + (int64_t)b * C + c, a * inv);
    }
}
// out = a0*y[:, :C] + a1*y[:, C:]; grid (hw blocks, B); attention and the BatchNorm coefficients of the sample staged in LDS
template <typename T>
__global__ __launch_bounds__(256) void splat_apply_bn_kernel(const T* __restrict__ x, SplatBn bn, const float* __restrict__ logits, T* __restrict__ out,
                                                             int HW, int C, int relu, int rpb) {
    constexpr int EPC = DT<T>::EPC;
    extern __shared__ float sm[];   // a0[C], sc[2C], sh[2C]
    float* a0s = sm;
    float* scs = sm + C;
    float* shs = sm + 3 * C;
    const int b = blockIdx.y;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float l0 = logits[(int64_t)b * 2 * C + c], l1 = logits[(int64_t)b * 2 * C + C + c];
        a0s[c] = 1.f / (1.f + expf(l1 - l0));
    }
    for (int c = threadIdx.x; c < 2 * C; c += 256) {
        const float k = bn.gamma[c] * bn.invstd[c];
        scs[c] = k;
        shs[c] = bn.beta[c] - bn.mean[c] * k;
    }
    __syncthreads();
    const int cpr = C / EPC;
    const int r0 = blockIdx.x * rpb, r1 = min(HW, r0 + rpb);
    const int64_t total = (int64_t)(r1 - r0) * cpr;
    for (int64_t i = threadIdx.x; i < total; i += 256) {
        const int r = r0 + (int)(i / cpr);
        const int c0 = (int)(i % cpr) * EPC;
        float u[EPC], v[EPC];
        const T* px = x + ((int64_t)b * HW + r) * 2 * C + c0;
        unpack16<T>(*(const uint4*)px, u);
        unpack16<T>(*(const uint4*)(px + C), v);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float a0 = a0s[c0 + e];
            const float yu = fmaxf(fmaf(u[e], scs[c0 + e], shs[c0 + e]), 0.f), yv = fmaxf(fmaf(v[e], scs[C + c0 + e], shs[C + c0 + e]), 0.f);
            float o = a0 * yu + (1.f - a0) * yv;
            if (relu) o = o > 0.f ? o : 0.f;
            u[e] = o;
        }
        *(uint4*)(out + ((int64_t)b * HW + r) * C + c0) = pack16<T>(u);
    }
}
// backward, logits: da[b][r*C+c] += sum_hw dout' * y_r   (dout' masked by out > 0 when relu)
template <typename T>
__global__ __launch_bounds__(256) void splat_bwd_reduce_bn_kernel(const T* __restrict__ dout, const T* __restrict__ x, SplatBn bn, const T* __restrict__ outp,
                                                                  float* __restrict__ da, int HW, int C, int TX, int rpb, int relu) {
    constexpr int EPC = DT<T>::EPC;
    extern __shared__ float red[];   // [RY][TX*EPC][2]
    const int RY = 256 / TX;
    const int cx = threadIdx.x % TX, ry = threadIdx.x / TX;
    const int col = blockIdx.x * TX + cx, cpr = C / EPC;
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * rpb, r1 = min(HW, r0 + rpb);
    float s0[EPC], s1[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { s0[e] = 0.f; s1[e] = 0.f; }
    if (col < cpr) {
        float scu[EPC], shu[EPC], scv[EPC], shv[EPC];
        bn_co<EPC>(bn.mean, bn.invstd, bn.gamma, bn.beta, col * EPC, scu, shu);
        bn_co<EPC>(bn.mean, bn.invstd, bn.gamma, bn.beta, C + col * EPC, scv, shv);
        int r = r0 + ry;
        for (; r + RY < r1; r += 2 * RY) {                 // two rows per iteration, loads first
            const int64_t pa = ((int64_t)b * HW + r) * C + col * EPC, pb = ((int64_t)b * HW + r + RY) * C + col * EPC;
            const T* xa = x + ((int64_t)b * HW + r) * 2 * C + col * EPC;
            const T* xb = x + ((int64_t)b * HW + r + RY) * 2 * C + col * EPC;
            const uint4 qd0 = *(const uint4*)(dout + pa), qd1 = *(const uint4*)(dout + pb);
            uint4 qo0 = make_uint4(0, 0, 0, 0), qo1 = qo0;
            if (relu) { qo0 = *(const uint4*)(outp + pa); qo1 = *(const uint4*)(outp + pb); }
            const uint4 qu0 = *(const uint4*)xa, qv0 = *(const uint4*)(xa + C), qu1 = *(const uint4*)xb, qv1 = *(const uint4*)(xb + C);
            float d0[EPC], d1[EPC], o0[EPC], o1[EPC], u0[EPC], v0[EPC], u1[EPC], v1[EPC];
            unpack16<T>(qd0, d0); unpack16<T>(qd1, d1); unpack16<T>(qu0, u0); unpack16<T>(qv0, v0); unpack16<T>(qu1, u1); unpack16<T>(qv1, v1);
            if (relu) { unpack16<T>(qo0, o0); unpack16<T>(qo1, o1); }
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float da = (relu && !(o0[e] > 0.f)) ? 0.f : d0[e], db = (relu && !(o1[e] > 0.f)) ? 0.f : d1[e];
                s0[e] += da * fmaxf(fmaf(u0[e], scu[e], shu[e]), 0.f) + db * fmaxf(fmaf(u1[e], scu[e], shu[e]), 0.f);
                s1[e] += da * fmaxf(fmaf(v0[e], scv[e], shv[e]), 0.f) + db * fmaxf(fmaf(v1[e], scv[e], shv[e]), 0.f);
            }
        }
        for (; r < r1; r += RY) {
            float d[EPC], o[EPC], u[EPC], v[EPC];
            const int64_t po = ((int64_t)b * HW + r) * C + col * EPC;
            unpack16<T>(*(const uint4*)(dout + po), d);
            if (relu) unpack16<T>(*(const uint4*)(outp + po), o);
            const T* px = x + ((int64_t)b * HW + r) * 2 * C + col * EPC;
            unpack16<T>(*(const uint4*)px, u);
            unpack16<T>(*(const uint4*)(px + C), v);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float dd = (relu && !(o[e] > 0.f)) ? 0.f : d[e];
                s0[e] += dd * fmaxf(fmaf(u[e], scu[e], shu[e]), 0.f);
                s1[e] += dd * fmaxf(fmaf(v[e], scv[e], shv[e]), 0.f);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) { red[(((size_t)ry * TX + cx) * EPC + e) * 2] = s0[e]; red[(((size_t)ry * TX + cx) * EPC + e) * 2 + 1] = s1[e]; }
    __syncthreads();
    for (int ch = threadIdx.x; ch < TX * EPC; ch += 256) {
        const int c = blockIdx.x * TX * EPC + ch;
        if (c >= C) continue;
        float a = 0.f, bb = 0.f;
        for (int yy = 0; yy < RY; ++yy) { a += red[((size_t)yy * TX * EPC + ch) * 2]; bb += red[((size_t)yy * TX * EPC + ch) * 2 + 1]; }
        atomicAdd(da + (int64_t)b * 2 * C + c, a);
        atomicAdd(da + (int64_t)b * 2 * C + C + c, bb);
    }
}
// backward, data: the gradient that reaches y is dy_r = a_r * dout' + dgap / HW (never stored); dyp = dy where y > 0.
//   PASS 0: BatchNorm-backward sums  partial[by][0][ch] = sum dyp, partial[by][1][ch] = sum dyp * xhat   (by = b * gridDim.y + slab)
//   PASS 1: dx = gamma * invstd * (dyp - fin0 - xhat * fin1)      (fin = sums / (B * HW), bn_bwd_finalize_kernel's output)
template <typename T, int PASS>
__global__ __launch_bounds__(256) void splat_bn_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ outp, const T* __restrict__ x, SplatBn bn,
                                                           const float* __restrict__ logits, const float* __restrict__ dgap, const float* __restrict__ fin,
                                                           float* __restrict__ partial, T* __restrict__ dx, int HW, int C, int TX, int rpb, int relu) {
    constexpr int EPC = DT<T>::EPC;
    extern __shared__ float red[];   // PASS 0: [RY][TX*EPC][4]
    const int RY = 256 / TX;
    const int cx = threadIdx.x % TX, ry = threadIdx.x / TX;
    const int col = blockIdx.x * TX + cx, cpr = C / EPC;
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * rpb, r1 = min(HW, r0 + rpb);
    const int C2 = 2 * C;
    float s[4][EPC];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < EPC; ++e) s[k][e] = 0.f;
    if (col < cpr) {
        const int cu = col * EPC, cv = C + col * EPC;
        float scu[EPC], shu[EPC], scv[EPC], shv[EPC], a0[EPC], dg[EPC];
        bn_co<EPC>(bn.mean, bn.invstd, bn.gamma, bn.beta, cu, scu, shu);
        bn_co<EPC>(bn.mean, bn.invstd, bn.gamma, bn.beta, cv, scv, shv);
        const float inv = 1.f / (float)HW;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float l0 = logits[(int64_t)b * C2 + cu + e], l1 = logits[(int64_t)b * C2 + cv + e];
            a0[e] = 1.f / (1.f + expf(l1 - l0));
            dg[e] = dgap[(int64_t)b * C + cu + e] * inv;
        }
        // PASS 0: xhat = (x - mu) * is;  PASS 1: dx = gi * dyp - P * x - Q with P = gi * fin1 * is, Q = gi * fin0 - P * mu
        float m0[EPC], m1[EPC], n0[EPC], n1[EPC], gu[EPC], gv[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float muu = bn.mean[cu + e], isu = bn.invstd[cu + e], muv = bn.mean[cv + e], isv = bn.invstd[cv + e];
            if (PASS == 0) { m0[e] = muu; n0[e] = isu; m1[e] = muv; n1[e] = isv; gu[e] = 0.f; gv[e] = 0.f; }
            else {
                gu[e] = bn.gamma[cu + e] * isu; gv[e] = bn.gamma[cv + e] * isv;
                const float pu = gu[e] * fin[C2 + cu + e] * isu, pv = gv[e] * fin[C2 + cv + e] * isv;
                m0[e] = pu; n0[e] = gu[e] * fin[cu + e] - pu * muu;
                m1[e] = pv; n1[e] = gv[e] * fin[cv + e] - pv * muv;
            }
        }
        auto row = [&](int r, const uint4& qd, const uint4& qo, const uint4& qu, const uint4& qv) {
            float d[EPC], o[EPC], u[EPC], v[EPC], ou[EPC], ov[EPC];
            unpack16<T>(qd, d); unpack16<T>(qu, u); unpack16<T>(qv, v);
            if (relu) unpack16<T>(qo, o);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float dd = (relu && !(o[e] > 0.f)) ? 0.f : d[e];
                const float du = fmaf(u[e], scu[e], shu[e]) > 0.f ? fmaf(a0[e], dd, dg[e]) : 0.f;
                const float dv = fmaf(v[e], scv[e], shv[e]) > 0.f ? fmaf(1.f - a0[e], dd, dg[e]) : 0.f;
                if (PASS == 0) {
                    s[0][e] += du; s[1][e] += du * (u[e] - m0[e]) * n0[e];
                    s[2][e] += dv; s[3][e] += dv * (v[e] - m1[e]) * n1[e];
                } else {
                    ou[e] = gu[e] * du - m0[e] * u[e] - n0[e];
                    ov[e] = gv[e] * dv - m1[e] * v[e] - n1[e];
                }
            }
            if (PASS == 1) {
                T* pd = dx + ((int64_t)b * HW + r) * C2 + cu;
                *(uint4*)pd = pack16<T>(ou);
                *(uint4*)(pd + C) = pack16<T>(ov);
            }
        };
        const uint4 z4 = make_uint4(0, 0, 0, 0);
        int r = r0 + ry;
        for (; r + RY < r1; r += 2 * RY) {                 // two rows per iteration: eight loads in flight, then the arithmetic
            const int64_t pa = ((int64_t)b * HW + r) * C + cu, pb = ((int64_t)b * HW + r + RY) * C + cu;
            const T* xa = x + ((int64_t)b * HW + r) * C2 + cu;
            const T* xb = x + ((int64_t)b * HW + r + RY) * C2 + cu;
            const uint4 qd0 = *(const uint4*)(dout + pa), qd1 = *(const uint4*)(dout + pb);
            const uint4 qo0 = relu ? *(const uint4*)(outp + pa) : z4, qo1 = relu ? *(const uint4*)(outp + pb) : z4;
            const uint4 qu0 = *(const uint4*)xa, qv0 = *(const uint4*)(xa + C), qu1 = *(const uint4*)xb, qv1 = *(const uint4*)(xb + C);
            row(r, qd0, qo0, qu0, qv0);
            row(r + RY, qd1, qo1, qu1, qv1);
        }
        for (; r < r1; r += RY) {
            const int64_t po = ((int64_t)b * HW + r) * C + cu;
            const T* px = x + ((int64_t)b * HW + r) * C2 + cu;
            row(r, *(const uint4*)(dout + po), relu ? *(const uint4*)(outp + po) : z4, *(const uint4*)px, *(const uint4*)(px + C));
        }
    }
    if (PASS == 1) return;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < EPC; ++e) red[(((size_t)ry * TX + cx) * EPC + e) * 4 + k] = s[k][e];
    __syncthreads();
    const int by = b * gridDim.y + blockIdx.y;
    for (int ch = threadIdx.x; ch < TX * EPC; ch += 256) {
        const int c = blockIdx.x * TX * EPC + ch;
        if (c >= C) continue;
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        for (int yy = 0; yy < RY; ++yy)
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] += red[((size_t)yy * TX * EPC + ch) * 4 + k];
        partial[((size_t)by * 2 + 0) * C2 + c] = a[0];
        partial[((size_t)by * 2 + 1) * C2 + c] = a[1];
        partial[((size_t)by * 2 + 0) * C2 + C + c] = a[2];
        partial[((size_t)by * 2 + 1) * C2 + C + c] = a[3];
    }
}
// ---- round 4: bn0's backward sums WITHOUT a pass of their own.  The gradient that reaches y_r is dy_r = a_r dout' + dgap / HW, and
// a_r, dgap are constants of a (sample, channel): with m_r = [y_r > 0]
//     sum_hw m_r dy_r        = a_r P_r  + (dgap / HW) M_r          P_r  = sum_hw m_r dout'        M_r  = sum_hw m_r
//     sum_hw m_r dy_r xhat_r = a_r Px_r + (dgap / HW) Mx_r         Px_r = sum_hw m_r dout' xhat_r   Mx_r = sum_hw m_r xhat_r
// P, Px, M, Mx do not depend on the micro-net's backward, so the pass that computes the logit gradients (it reads dout, out and x
// anyway) takes them along: aux[b][k][c], k = 0..3 for the first radix half (P, Px, M, Mx), 4..7 for the second.  After the
// micro-net's backward ONE tiny kernel assembles the per-sample sums and folds them over the batch (splat_bn_assemble_finalize_kernel),
// and the dx pass follows: three passes over (dout, out, x) become two
// (tools/bn_ledger.py: 8.2 GB -> 5.9 GB per step at B = 16, 400 x 400).
template <typename T>
__global__ __launch_bounds__(256) void splat_bwd_reduce_bn2_kernel(const T* __restrict__ dout, const T* __restrict__ x, SplatBn bn, const T* __restrict__ outp,
                                                                   float* __restrict__ da, float* __restrict__ aux, int HW, int C, int TX, int rpb, int relu, int rev) {
    constexpr int EPC = DT<T>::EPC;
    extern __shared__ float red[];   // [RY][TX*EPC][5], used twice
    const int RY = 256 / TX;
    const int cx = threadIdx.x % TX, ry = threadIdx.x / TX;
    const int col = blockIdx.x * TX + cx, cpr = C / EPC;
    // (rev: end first -- the last samples / rows are what the upstream gradient kernel wrote last, see bn_reduce_kernel)
    const int b = rev ? (int)(gridDim.z - 1 - blockIdx.z) : (int)blockIdx.z;
    const int r0 = (rev ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y) * rpb, r1 = min(HW, r0 + rpb);
    float s[10][EPC];                // 0: da_u, 1..4: P Px M Mx (u); 5: da_v, 6..9: P Px M Mx (v)
#pragma unroll
    for (int k = 0; k < 10; ++k)
#pragma unroll
        for (int e = 0; e < EPC; ++e) s[k][e] = 0.f;
    if (col < cpr) {
        const int cu = col * EPC, cv = C + col * EPC;
        float scu[EPC], shu[EPC], scv[EPC], shv[EPC], muu[EPC], isu[EPC], muv[EPC], isv[EPC];
        bn_co<EPC>(bn.mean, bn.invstd, bn.gamma, bn.beta, cu, scu, shu);
        bn_co<EPC>(bn.mean, bn.invstd, bn.gamma, bn.beta, cv, scv, shv);
#pragma unroll
        for (int e = 0; e < EPC; ++e) { muu[e] = bn.mean[cu + e]; isu[e] = bn.invstd[cu + e]; muv[e] = bn.mean[cv + e]; isv[e] = bn.invstd[cv + e]; }
        auto row = [&](const uint4& qd, const uint4& qo, const uint4& qu, const uint4& qv) {
            float d[EPC], o[EPC], u[EPC], v[EPC];
            unpack16<T>(qd, d); unpack16<T>(qu, u); unpack16<T>(qv, v);
            if (relu) unpack16<T>(qo, o);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float dd = (relu && !(o[e] > 0.f)) ? 0.f : d[e];
                const float yu = fmaf(u[e], scu[e], shu[e]), yv = fmaf(v[e], scv[e], shv[e]);
                const float mu_ = yu > 0.f ? 1.f : 0.f, mv_ = yv > 0.f ? 1.f : 0.f;
                const float xu = (u[e] - muu[e]) * isu[e], xv = (v[e] - muv[e]) * isv[e];
                const float du = mu_ * dd, dv = mv_ * dd;
                s[0][e] += du * yu;                       // = dd * max(yu, 0)
                s[1][e] += du; s[2][e] += du * xu; s[3][e] += mu_; s[4][e] += mu_ * xu;
                s[5][e] += dv * yv;
                s[6][e] += dv; s[7][e] += dv * xv; s[8][e] += mv_; s[9][e] += mv_ * xv;
            }
        };
        const uint4 z4 = make_uint4(0, 0, 0, 0);
        int r = r0 + ry;
        for (; r + RY < r1; r += 2 * RY) {                 // two rows per iteration, loads first
            const int64_t pa = ((int64_t)b * HW + r) * C + cu, pb = ((int64_t)b * HW + r + RY) * C + cu;
            const T* xa = x + ((int64_t)b * HW + r) * 2 * C + cu;
            const T* xb = x + ((int64_t)b * HW + r + RY) * 2 * C + cu;
            const uint4 qd0 = *(const uint4*)(dout + pa), qd1 = *(const uint4*)(dout + pb);
            const uint4 qo0 = relu ? *(const uint4*)(outp + pa) : z4, qo1 = relu ? *(const uint4*)(outp + pb) : z4;
            const uint4 qu0 = *(const uint4*)xa, qv0 = *(const uint4*)(xa + C), qu1 = *(const uint4*)xb, qv1 = *(const uint4*)(xb + C);
            row(qd0, qo0, qu0, qv0);
            row(qd1, qo1, qu1, qv1);
        }
        for (; r < r1; r += RY) {
            const int64_t po = ((int64_t)b * HW + r) * C + cu;
            const T* px = x + ((int64_t)b * HW + r) * 2 * C + cu;
            row(*(const uint4*)(dout + po), relu ? *(const uint4*)(outp + po) : z4, *(const uint4*)px, *(const uint4*)(px + C));
        }
    }
    // block reduction over the RY row groups, five sums at a time (one radix half per round)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();
#pragma unroll
        for (int k = 0; k < 5; ++k)
#pragma unroll
            for (int e = 0; e < EPC; ++e) red[(((size_t)ry * TX + cx) * EPC + e) * 5 + k] = s[half * 5 + k][e];
        __syncthreads();
        for (int ch = threadIdx.x; ch < TX * EPC; ch += 256) {
            const int c = blockIdx.x * TX * EPC + ch;
            if (c >= C) continue;
            float a[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
            for (int yy = 0; yy < RY; ++yy)
#pragma unroll
                for (int k = 0; k < 5; ++k) a[k] += red[((size_t)yy * TX * EPC + ch) * 5 + k];
            atomicAdd(da + (int64_t)b * 2 * C + half * C + c, a[0]);
#pragma unroll
            for (int k = 0; k < 4; ++k) atomicAdd(aux + ((int64_t)b * 8 + half * 4 + k) * C + c, a[1 + k]);
        }
    }
}
// bn0's backward sums assembled from aux and folded over the samples in ONE launch (the assemble + bn_bwd_finalize pair was two
// 5-6 us launches on the critical path of every split-attention block): thread = channel ch of the 2C tensor, loop over b <= 32,
// double accumulation like bn_bwd_finalize_kernel.  fin[0][ch] = sum m dy / N, fin[1][ch] = sum m dy xhat / N, dbeta / dgamma +=.
__global__ __launch_bounds__(256) void splat_bn_assemble_finalize_kernel(const float* __restrict__ aux, const float* __restrict__ logits,
                                                                         const float* __restrict__ dgap, float* __restrict__ fin, float* __restrict__ dgamma,
                                                                         float* __restrict__ dbeta, int B, int C, float inv_hw, double inv_rows) {
    const int ch = blockIdx.x * 256 + threadIdx.x, C2 = 2 * C;
    if (ch >= C2) return;
    const int half = ch >= C ? 1 : 0, c = ch - half * C;
    double s = 0.0, ss = 0.0;
    // eight samples' operands in flight at once (56 loads): in the step they are cold, and one sample per trip made this a chain of B memory
    // round trips in 1-8 workgroups (9.2 us per launch); the sums run in the same order as before
    for (int b0 = 0; b0 < B; b0 += 8) {
        float t0[8], t1[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bool ok = b0 + k < B;
            const int b = ok ? b0 + k : B - 1;
            const float l0 = logits[(int64_t)b * C2 + c], l1 = logits[(int64_t)b * C2 + C + c];
            const float dgv = dgap[(int64_t)b * C + c];
            const float* au = aux + ((int64_t)b * 8 + half * 4) * C + c;
            const float u0 = au[0], u1 = au[(int64_t)C], u2 = au[2 * (int64_t)C], u3 = au[3 * (int64_t)C];
            const float a0 = 1.f / (1.f + expf(l1 - l0)), a = half ? 1.f - a0 : a0, dg = dgv * inv_hw;
            t0[k] = ok ? fmaf(a, u0, dg * u2) : 0.f;
            t1[k] = ok ? fmaf(a, u1, dg * u3) : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (b0 + k < B) { s += (double)t0[k]; ss += (double)t1[k]; }
        }
    }
    if (dbeta) dbeta[ch] += (float)s;
    if (dgamma) dgamma[ch] += (float)ss;
    fin[ch] = (float)(s * inv_rows);
    fin[C2 + ch] = (float)(ss * inv_rows);
}
int octa_bn_bwd_finalize_launch(const float* partial, int nby, int C, int64_t rows, float* fin, float* dgamma, float* dbeta, hipStream_t st);   // norm.hip

#define OCTA_SPLAT_BN_ARGS                                                                                                             \
    OCTA_REQUIRE(x && mean && invstd && gamma && beta && B > 0 && HW > 0 && C % 8 == 0 && C <= 4096, "octa_splat_bn: bad arguments (C %% 8, C <= 4096)"); \
    OCTA_REQUIRE(OCTA_DTYPE_OK(dtype), "octa_splat_bn: bad dtype");                                                                    \
    hipStream_t st = (hipStream_t)stream;                                                                                              \
    const int epc = dtype == OCTA_F32 ? 4 : 8;                                                                                         \
    SplatBn bn;                                                                                                                        \
    bn.mean = mean; bn.invstd = invstd; bn.gamma = gamma; bn.beta = beta

extern "C" int octa_splat_bn_gap(const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta, float* gap, int B, int HW,
                                 int C, int dtype, int prezeroed, octa_stream_t stream) {
    OCTA_SPLAT_BN_ARGS;
    OCTA_REQUIRE(gap, "octa_splat_bn_gap: null pointer");
    if (!prezeroed && octa_zero_async(gap, (size_t)B * C * sizeof(float), st) != hipSuccess) OCTA_FAIL(OCTA_ERR_LAUNCH, "octa_splat_bn_gap: memset failed");
    int TX, gx;
    splat_map(C / epc, TX, gx);
    const int RY = 256 / TX;
    int rpb = cdiv(HW, splat_reduce_slabs());
    if (rpb < RY * 8) rpb = RY * 8;
    if (octa_deterministic()) rpb = HW;                    // one workgroup per (sample, column block): one add per address, fixed order
    dim3 grid(gx, cdiv(HW, rpb), B);
    const size_t sh = (size_t)256 * epc * sizeof(float);
    if (dtype == OCTA_F32) splat_gap_bn_kernel<float><<<grid, 256, sh, st>>>((const float*)x, bn, gap, HW, C, TX, rpb);
    else if (dtype == OCTA_BF16) splat_gap_bn_kernel<bf16_t><<<grid, 256, sh, st>>>((const bf16_t*)x, bn, gap, HW, C, TX, rpb);
    else splat_gap_bn_kernel<f16_t><<<grid, 256, sh, st>>>((const f16_t*)x, bn, gap, HW, C, TX, rpb);
    OCTA_CHECK_LAUNCH("splat_gap_bn");
    return OCTA_OK;
}
extern "C" int octa_splat_bn_apply(const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta, const float* logits,
                                   void* out, int B, int HW, int C, int dtype, int relu, octa_stream_t stream) {
    OCTA_SPLAT_BN_ARGS;
    OCTA_REQUIRE(logits && out, "octa_splat_bn_apply: null pointer");
    int rpb = cdiv(HW, splat_apply_slabs());
    if (rpb < 8) rpb = 8;
    dim3 grid(cdiv(HW, rpb), B);
    const size_t sh = (size_t)5 * C * sizeof(float);
    if (dtype == OCTA_F32) splat_apply_bn_kernel<float><<<grid, 256, sh, st>>>((const float*)x, bn, logits, (float*)out, HW, C, relu, rpb);
    else if (dtype == OCTA_BF16) splat_apply_bn_kernel<bf16_t><<<grid, 256, sh, st>>>((const bf16_t*)x, bn, logits, (bf16_t*)out, HW, C, relu, rpb);
    else splat_apply_bn_kernel<f16_t><<<grid, 256, sh, st>>>((const f16_t*)x, bn, logits, (f16_t*)out, HW, C, relu, rpb);
    OCTA_CHECK_LAUNCH("splat_apply_bn");
    return OCTA_OK;
}
extern "C" int octa_splat_bn_bwd_logits(const void* dout, const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                        const float* logits, const void* out, float* dlogits, int B, int HW, int C, int dtype, int relu, int prezeroed,
                                        octa_stream_t stream) {
    OCTA_SPLAT_BN_ARGS;
    OCTA_REQUIRE(dout && logits && dlogits && (!relu || out), "octa_splat_bn_bwd_logits: null pointer (relu needs the forward output)");
    if (!prezeroed && octa_zero_async(dlogits, (size_t)B * 2 * C * sizeof(float), st) != hipSuccess) OCTA_FAIL(OCTA_ERR_LAUNCH, "octa_splat_bn_bwd_logits: memset failed");
    int TX, gx;
    splat_map(C / epc, TX, gx);
    const int RY = 256 / TX;
    int rpb = cdiv(HW, splat_reduce_slabs());
    if (rpb < RY * 8) rpb = RY * 8;
    if (octa_deterministic()) rpb = HW;                    // one workgroup per (sample, column block): one add per address, fixed order
    dim3 grid(gx, cdiv(HW, rpb), B);
    const size_t sh = (size_t)256 * epc * 2 * sizeof(float);
    if (dtype == OCTA_F32) splat_bwd_reduce_bn_kernel<float><<<grid, 256, sh, st>>>((const float*)dout, (const float*)x, bn, (const float*)out, dlogits, HW, C, TX, rpb, relu);
    else if (dtype == OCTA_BF16) splat_bwd_reduce_bn_kernel<bf16_t><<<grid, 256, sh, st>>>((const bf16_t*)dout, (const bf16_t*)x, bn, (const bf16_t*)out, dlogits, HW, C, TX, rpb, relu);
    else splat_bwd_reduce_bn_kernel<f16_t><<<grid, 256, sh, st>>>((const f16_t*)dout, (const f16_t*)x, bn, (const f16_t*)out, dlogits, HW, C, TX, rpb, relu);
    OCTA_CHECK_LAUNCH("splat_bwd_reduce_bn");
    splat_softmax_bwd_kernel<<<cdiv(B * C, 256), 256, 0, st>>>(logits, dlogits, B, C);
    OCTA_CHECK_LAUNCH("splat_softmax_bwd");
    return OCTA_OK;
}
static int splat_bn_bwd_logits2_impl(const void* dout, const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                         const float* logits, const void* out, float* dlogits, float* aux, int B, int HW, int C, int dtype, int relu,
                                         int prezeroed, int raw_da, octa_stream_t stream) {
    OCTA_SPLAT_BN_ARGS;
    OCTA_REQUIRE(dout && logits && dlogits && aux && (!relu || out), "octa_splat_bn_bwd_logits2: null pointer (relu needs the forward output)");
    if (!prezeroed) {
        if (octa_zero_async(dlogits, (size_t)B * 2 * C * sizeof(float), st) != hipSuccess || octa_zero_async(aux, (size_t)B * 8 * C * sizeof(float), st) != hipSuccess)
            OCTA_FAIL(OCTA_ERR_LAUNCH, "octa_splat_bn_bwd_logits2: memset failed");
    }
    int TX, gx;
    splat_map(C / epc, TX, gx);
    const int RY = 256 / TX;
    int rpb = cdiv(HW, splat_reduce_slabs());
    if (rpb < RY * 8) rpb = RY * 8;
    if (octa_deterministic()) rpb = HW;                    // one workgroup per (sample, column block): one add per address, fixed order
    dim3 grid(gx, cdiv(HW, rpb), B);
    const size_t sh = (size_t)256 * epc * 5 * sizeof(float);
    if (dtype == OCTA_F32) splat_bwd_reduce_bn2_kernel<float><<<grid, 256, sh, st>>>((const float*)dout, (const float*)x, bn, (const float*)out, dlogits, aux, HW, C, TX, rpb, relu, octa_rev_walk());
    else if (dtype == OCTA_BF16) splat_bwd_reduce_bn2_kernel<bf16_t><<<grid, 256, sh, st>>>((const bf16_t*)dout, (const bf16_t*)x, bn, (const bf16_t*)out, dlogits, aux, HW, C, TX, rpb, relu, octa_rev_walk());
    else splat_bwd_reduce_bn2_kernel<f16_t><<<grid, 256, sh, st>>>((const f16_t*)dout, (const f16_t*)x, bn, (const f16_t*)out, dlogits, aux, HW, C, TX, rpb, relu, octa_rev_walk());
    OCTA_CHECK_LAUNCH("splat_bwd_reduce_bn2");
    if (!raw_da) {
        splat_softmax_bwd_kernel<<<cdiv(B * C, 256), 256, 0, st>>>(logits, dlogits, B, C);
        OCTA_CHECK_LAUNCH("splat_softmax_bwd");
    }
    return OCTA_OK;
}
extern "C" int octa_splat_bn_bwd_logits2(const void* dout, const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                         const float* logits, const void* out, float* dlogits, float* aux, int B, int HW, int C, int dtype, int relu,
                                         int prezeroed, octa_stream_t stream) {
    return splat_bn_bwd_logits2_impl(dout, x, mean, invstd, gamma, beta, logits, out, dlogits, aux, B, HW, C, dtype, relu, prezeroed, 0, stream);
}
extern "C" int octa_splat_bn_bwd_da2(const void* dout, const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                     const float* logits, const void* out, float* da, float* aux, int B, int HW, int C, int dtype, int relu,
                                     int prezeroed, octa_stream_t stream) {
    return splat_bn_bwd_logits2_impl(dout, x, mean, invstd, gamma, beta, logits, out, da, aux, B, HW, C, dtype, relu, prezeroed, 1, stream);
}
extern "C" int octa_splat_bn_bwd_dx2(const void* dout, const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                     const float* logits, const void* out, const float* dgap, const float* aux, void* dx, float* dgamma, float* dbeta,
                                     float* ws, int B, int HW, int C, int dtype, int relu, octa_stream_t stream) {
    OCTA_SPLAT_BN_ARGS;
    OCTA_REQUIRE(dout && logits && dgap && aux && dx && ws && (!relu || out), "octa_splat_bn_bwd_dx2: null pointer (relu needs the forward output)");
    OCTA_REQUIRE(B <= 1024, "octa_splat_bn_bwd_dx2: batch %d too large for the partial-sum workspace", B);
    int TX, gx;
    splat_map(C / epc, TX, gx);
    const int RY = 256 / TX;
    int nslab = splat_dx_slabs(B);
    if (nslab < 1) nslab = 1;
    int rpb = cdiv(HW, nslab);
    if (rpb < RY * 8) rpb = RY * 8;
    nslab = cdiv(HW, rpb);
    dim3 grid(gx, nslab, B);
    const int C2 = 2 * C;
    float* fin = ws + (size_t)1026 * 2 * C2;
    splat_bn_assemble_finalize_kernel<<<cdiv(C2, 256), 256, 0, st>>>(aux, logits, dgap, fin, dgamma, dbeta, B, C, 1.f / (float)HW, 1.0 / ((double)B * HW));
    OCTA_CHECK_LAUNCH("splat_bn_assemble_finalize");
#define OCTA_SPLAT_BN_BWD1(TT)                                                                                                                   \
    splat_bn_bwd_kernel<TT, 1><<<grid, 256, 0, st>>>((const TT*)dout, (const TT*)out, (const TT*)x, bn, logits, dgap, fin, ws, (TT*)dx, HW, C, TX, rpb, relu)
    if (dtype == OCTA_F32) { OCTA_SPLAT_BN_BWD1(float); }
    else if (dtype == OCTA_BF16) { OCTA_SPLAT_BN_BWD1(bf16_t); }
    else { OCTA_SPLAT_BN_BWD1(f16_t); }
#undef OCTA_SPLAT_BN_BWD1
    OCTA_CHECK_LAUNCH("splat_bn_bwd(dx)");
    return OCTA_OK;
}
extern "C" int octa_splat_bn_bwd_dx(const void* dout, const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                    const float* logits, const void* out, const float* dgap, void* dx, float* dgamma, float* dbeta, float* ws, int B,
                                    int HW, int C, int dtype, int relu, octa_stream_t stream) {
    OCTA_SPLAT_BN_ARGS;
    OCTA_REQUIRE(dout && logits && dgap && dx && ws && (!relu || out), "octa_splat_bn_bwd_dx: null pointer (relu needs the forward output)");
    int TX, gx;
    splat_map(C / epc, TX, gx);
    const int RY = 256 / TX;
    int nslab = splat_dx_slabs(B);                                  // partial rows: B * nslab <= 1024 (octa_bn_workspace_floats)
    if (nslab < 1) nslab = 1;
    int rpb = cdiv(HW, nslab);
    if (rpb < RY * 8) rpb = RY * 8;
    nslab = cdiv(HW, rpb);
    OCTA_REQUIRE(B * nslab <= 1024, "octa_splat_bn_bwd_dx: batch %d too large for the partial-sum workspace", B);
    dim3 grid(gx, nslab, B);
    const int C2 = 2 * C;
    float* fin = ws + (size_t)1026 * 2 * C2;
    const size_t sh = (size_t)256 * epc * 4 * sizeof(float);
#define OCTA_SPLAT_BN_BWD(TT, PASS, SH)                                                                                                          \
    splat_bn_bwd_kernel<TT, PASS><<<grid, 256, SH, st>>>((const TT*)dout, (const TT*)out, (const TT*)x, bn, logits, dgap, fin, ws, (TT*)dx, HW, C, TX, rpb, relu)
    if (dtype == OCTA_F32) { OCTA_SPLAT_BN_BWD(float, 0, sh); }
    else if (dtype == OCTA_BF16) { OCTA_SPLAT_BN_BWD(bf16_t, 0, sh); }
    else { OCTA_SPLAT_BN_BWD(f16_t, 0, sh); }
    OCTA_CHECK_LAUNCH("splat_bn_bwd(sums)");
    const int rc = octa_bn_bwd_finalize_launch(ws, B * nslab, C2, (int64_t)B * HW, fin, dgamma, dbeta, st);
    if (rc != OCTA_OK) return rc;
    if (dtype == OCTA_F32) { OCTA_SPLAT_BN_BWD(float, 1, 0); }
    else if (dtype == OCTA_BF16) { OCTA_SPLAT_BN_BWD(bf16_t, 1, 0); }
    else { OCTA_SPLAT_BN_BWD(f16_t, 1, 0); }
#undef OCTA_SPLAT_BN_BWD
    OCTA_CHECK_LAUNCH("splat_bn_bwd(dx)");
    return OCTA_OK;
}
#undef OCTA_SPLAT_BN_ARGS

// =========================================================================================== SplAt attention micro-net
// fc1 (grouped 1x1) -> bn1 over the batch -> relu -> fc2 (grouped 1x1) on (B, C) vectors (resnest.py:118-125):
// a few MFLOP, so instead of ~25 generic launches per split-attention block each direction is a handful of
// wave-per-output-row kernels in exact fp32 (one wavefront = one output channel, lanes stride the input
// channels, B <= 32 accumulators per lane, wavefront shuffles for the sums).
#define SPLAT_MAXB 32

// Thread layout of the three reduction-shaped kernels (fwd1, fwd2, bwdA): lane bits = batch index bl (BT = 16 or 32 lanes),
// the remaining bits = slice s of the reduced axis.  One accumulator per thread, the slices meet through one or two cross-row
// shuffles and a 4-entry LDS column, and the batch statistics are a DPP reduction over the BT lanes.  (The first version kept
// 32 accumulators per thread and ran a full 64-lane wave_sum per batch entry: ~100 LDS-crossbar shuffles per block, 12-16 us
// for a few kFLOP.)
template <int BT>
__device__ __forceinline__ float splat_row_sum(float v) {       // sum over the BT lanes that share the slice; every lane gets it
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x124, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x128, 0xF, 0xF, true));
    if (BT == 32) v += __shfl_xor(v, 16, 64);
    return v;
}
template <int BT>
__device__ __forceinline__ float splat_slice_sum(float v, float (*red)[32]) {   // sum over the 256 / BT slices; valid in threads < BT
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (BT == 16) v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (lane < BT) red[wave][lane] = v;
    __syncthreads();
    return (threadIdx.x < BT) ? (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]) : 0.f;
}

// block per fc1 output channel j
template <int BT>
__global__ __launch_bounds__(256) void splat_mlp_fwd1_kernel(const float* __restrict__ gap, const float* __restrict__ w1, const float* __restrict__ b1,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ rm,
                                                             float* __restrict__ rv, float momentum, float eps, int training, float* __restrict__ h1,
                                                             float* __restrict__ h2, float* __restrict__ mean, float* __restrict__ invstd, int B, int C,
                                                             int inter, int groups) {
    __shared__ float red[4][32];
    constexpr int NS = 256 / BT;
    const int j = blockIdx.x, tid = threadIdx.x, bl = tid & (BT - 1), sl = tid / BT;
    const int Cg = C / groups, grp = j / (inter / groups);
    const float* wr = w1 + (size_t)j * Cg;
    const float* gr = gap + (size_t)(bl < B ? bl : 0) * C + grp * Cg;
    float acc = 0.f;
    if (bl < B)
        for (int cb = sl; cb < Cg; cb += 8 * NS) {           // eight steps' operands in flight at once (cold in the step); same sum order
            float wv[8], gv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int c = cb + k * NS, cc = c < Cg ? c : sl; wv[k] = wr[cc]; gv[k] = gr[cc]; }
#pragma unroll
            for (int k = 0; k < 8; ++k) if (cb + k * NS < Cg) acc += wv[k] * gv[k];
        }
    const float hsum = splat_slice_sum<BT>(acc, red);
    if (tid >= BT) return;                                  // threads 0 .. BT-1 (one row of wave 0) finish the channel
    const bool live = bl < B;
    const float h = live ? hsum + (b1 ? b1[j] : 0.f) : 0.f;
    float mu, is;
    if (training) {
        mu = splat_row_sum<BT>(h) / (float)B;
        const float d = live ? h - mu : 0.f;
        const float var = splat_row_sum<BT>(d * d) / (float)B;
        is = 1.f / sqrtf(var + eps);
        if (tid == 0) {
            if (rm) rm[j] = (1.f - momentum) * rm[j] + momentum * mu;
            if (rv) rv[j] = (1.f - momentum) * rv[j] + momentum * (B > 1 ? var * (float)B / (float)(B - 1) : var);
        }
    } else {
        mu = rm[j];
        is = 1.f / sqrtf(rv[j] + eps);
    }
    if (tid == 0) { mean[j] = mu; invstd[j] = is; }
    if (live) {
        h1[(size_t)bl * inter + j] = h;
        const float o = (h - mu) * is * gamma[j] + beta[j];
        h2[(size_t)bl * inter + j] = o > 0.f ? o : 0.f;
    }
}
// block per fc2 output channel n: logits[b][n]
template <int BT>
__global__ __launch_bounds__(256) void splat_mlp_fwd2_kernel(const float* __restrict__ h2, const float* __restrict__ w2, const float* __restrict__ b2,
                                                             float* __restrict__ logits, int B, int inter, int N, int groups) {
    __shared__ float red[4][32];
    constexpr int NS = 256 / BT;
    const int n = blockIdx.x, tid = threadIdx.x, bl = tid & (BT - 1), sl = tid / BT;
    const int Ig = inter / groups, grp = n / (N / groups);
    const float* wr = w2 + (size_t)n * Ig;
    const float* hr = h2 + (size_t)(bl < B ? bl : 0) * inter + grp * Ig;
    float acc = 0.f;
    if (bl < B)
        for (int jb = sl; jb < Ig; jb += 8 * NS) {
            float wv[8], hv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int j = jb + k * NS, jj = j < Ig ? j : sl; wv[k] = wr[jj]; hv[k] = hr[jj]; }
#pragma unroll
            for (int k = 0; k < 8; ++k) if (jb + k * NS < Ig) acc += wv[k] * hv[k];
        }
    const float v = splat_slice_sum<BT>(acc, red);
    if (tid < BT && bl < B) logits[(size_t)bl * N + n] = v + (b2 ? b2[n] : 0.f);
}
extern "C" int octa_splat_mlp_fwd(const float* gap, const float* w1, const float* b1, const float* gamma, const float* beta, float* rm, float* rv,
                                  float momentum, float eps, int training, const float* w2, const float* b2, float* h1, float* h2, float* mean,
                                  float* invstd, float* logits, int B, int C, int inter, int groups, octa_stream_t stream) {
    OCTA_REQUIRE(gap && w1 && gamma && beta && w2 && h1 && h2 && mean && invstd && logits, "octa_splat_mlp_fwd: null pointer");
    OCTA_REQUIRE(B >= 1 && B <= SPLAT_MAXB && groups >= 1 && C % groups == 0 && inter % groups == 0, "octa_splat_mlp_fwd: needs 1 <= B <= 32 (got %d)", B);
    OCTA_REQUIRE(training ? B > 1 : (rm && rv), "octa_splat_mlp_fwd: batch statistics need B > 1, eval needs running stats");
    hipStream_t st = (hipStream_t)stream;
    if (B <= 16) splat_mlp_fwd1_kernel<16><<<inter, 256, 0, st>>>(gap, w1, b1, gamma, beta, rm, rv, momentum, eps, training, h1, h2, mean, invstd, B, C, inter, groups);
    else splat_mlp_fwd1_kernel<32><<<inter, 256, 0, st>>>(gap, w1, b1, gamma, beta, rm, rv, momentum, eps, training, h1, h2, mean, invstd, B, C, inter, groups);
    OCTA_CHECK_LAUNCH("splat_mlp_fwd1");
    if (B <= 16) splat_mlp_fwd2_kernel<16><<<2 * C, 256, 0, st>>>(h2, w2, b2, logits, B, inter, 2 * C, groups);
    else splat_mlp_fwd2_kernel<32><<<2 * C, 256, 0, st>>>(h2, w2, b2, logits, B, inter, 2 * C, groups);
    OCTA_CHECK_LAUNCH("splat_mlp_fwd2");
    return OCTA_OK;
}

// backward A: block per j.  dh2[b][j] = sum_n dl[b][n] W2[n][j]  -> relu mask -> bn1 backward -> dh1[b][j]; dgamma/dbeta +=
// In the step every operand is cold (40-400 MB of activations pass between the forward and this launch), so the launch is a chain of memory round
// trips and nothing else.  Thread t takes rows n = t, t + 256, ... of W2's column j (one 4-byte piece of each row: all of them in flight at once)
// against dl[.][n] of all the batch entries (coalesced over t); the BT accumulators per thread meet through an LDS transpose (thread (bl, sl) sums
// the BT rows of slice sl for batch entry bl) and the usual slice sum.  The operands of the BatchNorm part travel with the first round trip.
// (Before: thread = (batch entry, slice of n), 32-64 dependent-latency iterations per thread: 16.4 us per launch in situ.)
// dl[b][n] of the micro-net backward.  RAW: `dl` holds the raw attention gradients da (what splat_bwd_reduce_bn2 summed) and the radix-2 softmax
// backward of splat_softmax_bwd_kernel -- dl0 = a0 (1 - a0) (da0 - da1), dl1 = -dl0, a0 = 1 / (1 + exp(l1 - l0)) -- is applied where the value is
// read (round 5: that kernel was a 4.7 us launch of B x C threads in front of every micro-net backward, 21 per step; here it is three more
// loads of the same round trip and one exp per value)
template <bool RAW>
__device__ __forceinline__ float splat_dl_at(const float* __restrict__ dl, const float* __restrict__ logits, int b, int n, int N) {
    if (!RAW) return dl[(size_t)b * N + n];
    const int C = N >> 1, c = n < C ? n : n - C;
    const float* lr = logits + (size_t)b * N;
    const float* dr = dl + (size_t)b * N;
    const float l0 = lr[c], l1 = lr[C + c], d0 = dr[c], d1 = dr[C + c];
    const float a0 = 1.f / (1.f + expf(l1 - l0));
    const float g = a0 * (1.f - a0) * (d0 - d1);
    return n < C ? g : -g;
}
template <int BT, bool RAW>
__device__ __forceinline__ void splat_mlp_bwdA_body(int j, const float* __restrict__ dl, const float* __restrict__ logits, const float* __restrict__ w2, const float* __restrict__ h1,
                                                            const float* __restrict__ h2, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ gamma, float* __restrict__ dh1,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ db1, int B,
                                                            int inter, int N, int groups, float (*red)[32], float* __restrict__ tr) {
    const int tid = threadIdx.x, bl = tid & (BT - 1), sl = tid / BT;
    const int Ig = inter / groups, Ng = N / groups, grp = j / Ig, jl = j - grp * Ig;
    const bool fin = tid < BT, live = fin && bl < B;
    float mu = 0.f, is = 0.f, g = 0.f, h1v = 0.f, h2v = 0.f;
    if (fin) { mu = mean[j]; is = invstd[j]; g = gamma[j]; }
    if (live) { h1v = h1[(size_t)bl * inter + j]; h2v = h2[(size_t)bl * inter + j]; }
    const float* wc = w2 + (size_t)grp * Ng * Ig + jl;
    float acc[BT];
#pragma unroll
    for (int b = 0; b < BT; ++b) acc[b] = 0.f;
    for (int nn = tid; nn < Ng; nn += 256) {
        const float w = wc[(size_t)nn * Ig];
#pragma unroll
        for (int b = 0; b < BT; ++b) acc[b] += b < B ? w * splat_dl_at<RAW>(dl, logits, b, grp * Ng + nn, N) : 0.f;
    }
#pragma unroll
    for (int b = 0; b < BT; ++b) tr[tid * (BT + 1) + b] = acc[b];
    __syncthreads();
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < BT; ++r) v += tr[(sl * BT + r) * (BT + 1) + bl];
    float d = splat_slice_sum<BT>(v, red);
    if (!fin) return;
    const float xh = live ? (h1v - mu) * is : 0.f;
    if (!live || !(h2v > 0.f)) d = 0.f;
    const float s1 = splat_row_sum<BT>(d), s2 = splat_row_sum<BT>(d * xh);
    const float o = live ? g * is * (d - s1 / (float)B - xh * (s2 / (float)B)) : 0.f;
    const float sdh = splat_row_sum<BT>(o);
    if (live) dh1[(size_t)bl * inter + j] = o;
    if (tid == 0) {
        dgamma[j] += s2;
        dbeta[j] += s1;
        if (db1) db1[j] += sdh;
    }
}
// backward B: block per n.  dW2[n][j] += sum_b dl[b][n] h2[b][j];  db2[n] += sum_b dl[b][n]
template <int BT, bool RAW>
__device__ __forceinline__ void splat_mlp_bwdB_body(int n, const float* __restrict__ dl, const float* __restrict__ logits, const float* __restrict__ h2, float* __restrict__ dw2,
                                                            float* __restrict__ db2, int B, int inter, int N, int groups) {
    const int Ig = inter / groups, grp = n / (N / groups);
    float d[BT];
    float sb = 0.f;
#pragma unroll
    for (int b = 0; b < BT; ++b) { d[b] = b < B ? splat_dl_at<RAW>(dl, logits, b, n, N) : 0.f; sb += d[b]; }     // uniform addresses: scalar loads
    for (int j = threadIdx.x; j < Ig; j += 256) {
        float a = 0.f;
#pragma unroll
        for (int b = 0; b < BT; ++b) a += d[b] * h2[(size_t)(b < B ? b : 0) * inter + grp * Ig + j];
        dw2[(size_t)n * Ig + j] += a;
    }
    if (threadIdx.x == 0 && db2) db2[n] += sb;
}
// backward C: dgap[b][c] += sum_j dh1[b][j] W1[j][c].  Block = 64 channels c (lanes) x 4 waves; the j range of the
// channel's group is split over gridDim.y blocks and the block's 4 waves (coalesced W1 rows), partial sums meet in
// LDS and leave with one atomic per (b, c).  dgap is zeroed by the host wrapper.
template <int BT>
__device__ __forceinline__ void splat_mlp_bwdC_body(int bx, int by, int ny, float (*red)[BT][64], const float* __restrict__ dh1, const float* __restrict__ w1, float* __restrict__ dgap, int B,
                                                             int C, int inter, int groups) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = bx * 64 + lane;
    const int Cg = C / groups, Ig = inter / groups;
    const bool live = c < C;
    // the 64 channels of a block may straddle two groups only when Cg % 64 != 0; the group is per lane, the dh1 row index is
    // uniform whenever it is not (the common case), so those loads become scalar
    const int grp = live ? c / Cg : 0, cl = live ? c - grp * Cg : 0;
    float acc[BT];
#pragma unroll
    for (int b = 0; b < BT; ++b) acc[b] = 0.f;
    const int per = (Ig + ny - 1) / ny;
    const int j0 = by * per, j1 = min(Ig, j0 + per);
    if (live)
        for (int jb = j0 + wave; jb < j1; jb += 16) {            // four rows of W1 in flight at once; same sum order
            float wv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int jj = jb + 4 * k; wv[k] = w1[(size_t)(grp * Ig + (jj < j1 ? jj : jb)) * Cg + cl]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int jj = jb + 4 * k;
                if (jj >= j1) continue;
                const int j = grp * Ig + jj;
#pragma unroll
                for (int b = 0; b < BT; ++b) acc[b] += wv[k] * dh1[(size_t)(b < B ? b : 0) * inter + j];
            }
        }
#pragma unroll
    for (int b = 0; b < BT; ++b) red[wave][b][lane] = acc[b];
    __syncthreads();
    for (int b = wave; b < B; b += 4)
        if (live) atomicAdd(dgap + (size_t)b * C + c, red[0][b][lane] + red[1][b][lane] + red[2][b][lane] + red[3][b][lane]);
}
// backward D: block per j.  dW1[j][c] += sum_b dh1[b][j] gap[b][c]
template <int BT>
__device__ __forceinline__ void splat_mlp_bwdD_body(int j, const float* __restrict__ dh1, const float* __restrict__ gap, float* __restrict__ dw1, int B,
                                                            int C, int inter, int groups) {
    const int Cg = C / groups, grp = j / (inter / groups);
    float d[BT];
#pragma unroll
    for (int b = 0; b < BT; ++b) d[b] = b < B ? dh1[(size_t)b * inter + j] : 0.f;
    for (int c = threadIdx.x; c < Cg; c += 256) {
        float a = 0.f;
#pragma unroll
        for (int b = 0; b < BT; ++b) a += d[b] * gap[(size_t)(b < B ? b : 0) * C + grp * Cg + c];
        dw1[(size_t)j * Cg + c] += a;
    }
}
template <int BT, bool RAW>
__global__ __launch_bounds__(256) void splat_mlp_bwdAB_kernel(const float* __restrict__ dl, const float* __restrict__ logits, const float* __restrict__ w2, const float* __restrict__ h1,
                                                             const float* __restrict__ h2, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma, float* __restrict__ dh1, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, float* __restrict__ db1, float* __restrict__ dw2,
                                                             float* __restrict__ db2, int B, int inter, int N, int groups) {
    __shared__ float red[4][32];
    __shared__ float tr[256 * (BT + 1)];
    if ((int)blockIdx.x < inter) splat_mlp_bwdA_body<BT, RAW>(blockIdx.x, dl, logits, w2, h1, h2, mean, invstd, gamma, dh1, dgamma, dbeta, db1, B, inter, N, groups, red, tr);
    else splat_mlp_bwdB_body<BT, RAW>(blockIdx.x - inter, dl, logits, h2, dw2, db2, B, inter, N, groups);
}
template <int BT>
__global__ __launch_bounds__(256) void splat_mlp_bwdCD_kernel(const float* __restrict__ dh1, const float* __restrict__ w1, const float* __restrict__ gap,
                                                             float* __restrict__ dgap, float* __restrict__ dw1, int B, int C, int inter, int groups,
                                                             int nx, int ny) {
    __shared__ float red[4][BT][64];
    const int nC = nx * ny;
    if ((int)blockIdx.x < nC) splat_mlp_bwdC_body<BT>(blockIdx.x % nx, blockIdx.x / nx, ny, red, dh1, w1, dgap, B, C, inter, groups);
    else splat_mlp_bwdD_body<BT>(blockIdx.x - nC, dh1, gap, dw1, B, C, inter, groups);
}
static int splat_mlp_bwd_impl(const float* dlogits, const float* raw_logits, const float* gap, const float* w1, const float* w2, const float* h1, const float* h2,
                                  const float* mean, const float* invstd, const float* gamma, float* dh1_ws, float* dgap, float* dw1, float* db1,
                                  float* dgamma, float* dbeta, float* dw2, float* db2, int B, int C, int inter, int groups, int prezeroed,
                                  octa_stream_t stream) {
    OCTA_REQUIRE(dlogits && gap && w1 && w2 && h1 && h2 && mean && invstd && gamma && dh1_ws && dgap && dw1 && dgamma && dbeta && dw2,
                 "octa_splat_mlp_bwd: null pointer");
    OCTA_REQUIRE(B > 1 && B <= SPLAT_MAXB && groups >= 1, "octa_splat_mlp_bwd: needs 2 <= B <= 32");
    hipStream_t st = (hipStream_t)stream;
    // two launches instead of four (+ a zero fill): A and B only read dlogits, C and D only read dh1 -- each pair shares a grid,
    // the block index selects the role
    if (raw_logits) {
        if (B <= 16) splat_mlp_bwdAB_kernel<16, true><<<inter + 2 * C, 256, 0, st>>>(dlogits, raw_logits, w2, h1, h2, mean, invstd, gamma, dh1_ws, dgamma, dbeta, db1, dw2, db2, B, inter, 2 * C, groups);
        else splat_mlp_bwdAB_kernel<32, true><<<inter + 2 * C, 256, 0, st>>>(dlogits, raw_logits, w2, h1, h2, mean, invstd, gamma, dh1_ws, dgamma, dbeta, db1, dw2, db2, B, inter, 2 * C, groups);
    } else if (B <= 16) splat_mlp_bwdAB_kernel<16, false><<<inter + 2 * C, 256, 0, st>>>(dlogits, nullptr, w2, h1, h2, mean, invstd, gamma, dh1_ws, dgamma, dbeta, db1, dw2, db2, B, inter, 2 * C, groups);
    else splat_mlp_bwdAB_kernel<32, false><<<inter + 2 * C, 256, 0, st>>>(dlogits, nullptr, w2, h1, h2, mean, invstd, gamma, dh1_ws, dgamma, dbeta, db1, dw2, db2, B, inter, 2 * C, groups);
    OCTA_CHECK_LAUNCH("splat_mlp_bwdAB");
    if (!prezeroed && octa_zero_async(dgap, (size_t)B * C * sizeof(float), st) != hipSuccess) OCTA_FAIL(OCTA_ERR_LAUNCH, "octa_splat_mlp_bwd: memset failed");
    {
        const int Ig = inter / groups;
        int js = Ig / 32;                 // >= 32 rows of W1 per block
        if (js < 1) js = 1;
        if (js > 8) js = 8;
        if (octa_deterministic()) js = 1;      // one workgroup per 64 channels of dgap: one add per address
        const int nC = cdiv(C, 64) * js;
        if (B <= 16) splat_mlp_bwdCD_kernel<16><<<nC + inter, 256, 0, st>>>(dh1_ws, w1, gap, dgap, dw1, B, C, inter, groups, cdiv(C, 64), js);
        else splat_mlp_bwdCD_kernel<32><<<nC + inter, 256, 0, st>>>(dh1_ws, w1, gap, dgap, dw1, B, C, inter, groups, cdiv(C, 64), js);
    }
    OCTA_CHECK_LAUNCH("splat_mlp_bwdCD");
    return OCTA_OK;
}
extern "C" int octa_splat_mlp_bwd(const float* dlogits, const float* gap, const float* w1, const float* w2, const float* h1, const float* h2,
                                  const float* mean, const float* invstd, const float* gamma, float* dh1_ws, float* dgap, float* dw1, float* db1,
                                  float* dgamma, float* dbeta, float* dw2, float* db2, int B, int C, int inter, int groups, int prezeroed,
                                  octa_stream_t stream) {
    return splat_mlp_bwd_impl(dlogits, nullptr, gap, w1, w2, h1, h2, mean, invstd, gamma, dh1_ws, dgap, dw1, db1, dgamma, dbeta, dw2, db2, B, C, inter, groups,
                              prezeroed, stream);
}
extern "C" int octa_splat_mlp_bwd_da(const float* da, const float* logits, const float* gap, const float* w1, const float* w2, const float* h1,
                                     const float* h2, const float* mean, const float* invstd, const float* gamma, float* dh1_ws, float* dgap, float* dw1,
                                     float* db1, float* dgamma, float* dbeta, float* dw2, float* db2, int B, int C, int inter, int groups, int prezeroed,
                                     octa_stream_t stream) {
    OCTA_REQUIRE(logits, "octa_splat_mlp_bwd_da: null logits");
    return splat_mlp_bwd_impl(da, logits, gap, w1, w2, h1, h2, mean, invstd, gamma, dh1_ws, dgap, dw1, db1, dgamma, dbeta, dw2, db2, B, C, inter, groups,
                              prezeroed, stream);
}

// =========================================================================================== AAG / head
// LPP lanes cooperate on one pixel; each lane owns CPL chunks (channels (lp + j*LPP)*EPC ..).
template <typename T, int K, int CPL>
__global__ __launch_bounds__(256) void aag_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                      T* __restrict__ masked, float* __restrict__ y, int64_t npix, int HW, int C, int LPP, int mode) {
    constexpr int EPC = DT<T>::EPC;
    extern __shared__ float ws[];   // [K][C]
    for (int i = threadIdx.x; i < K * C; i += 256) ws[i] = w[i];
    __syncthreads();
    const int lp = threadIdx.x % LPP;
    const int ppb = 256 / LPP;          // pixels per block iteration
    const int cpr = C / EPC;
    for (int64_t pix = (int64_t)blockIdx.x * ppb + threadIdx.x / LPP; pix < npix + ppb; pix += (int64_t)gridDim.x * ppb) {
        const bool live = pix < npix;   // keep whole waves in the shuffles
        float xv[CPL][EPC];
        float acc[K];
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = 0.f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const int ch = lp + j * LPP;
            if (live && ch < cpr) {
                unpack16<T>(*(const uint4*)(x + pix * C + ch * EPC), xv[j]);
#pragma unroll
                for (int k = 0; k < K; ++k)
#pragma unroll
                    for (int e = 0; e < EPC; ++e) acc[k] += xv[j][e] * ws[k * C + ch * EPC + e];
            } else {
#pragma unroll
                for (int e = 0; e < EPC; ++e) xv[j][e] = 0.f;
            }
        }
        for (int o = LPP >> 1; o > 0; o >>= 1)
#pragma unroll
            for (int k = 0; k < K; ++k) acc[k] += __shfl_xor(acc[k], o, 64);
        if (!live) continue;
        float l[K];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < K; ++k) { l[k] = acc[k] + bias[k]; mx = fmaxf(mx, l[k]); }
        const int64_t b = pix / HW, hw = pix % HW;
        if (mode == 1) {
            if (lp == 0)
#pragma unroll
                for (int k = 0; k < K; ++k) y[(b * K + k) * HW + hw] = l[k];
            continue;
        }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) { l[k] = expf(l[k] - mx); s += l[k]; }
        const float inv = 1.f / s;
        float mask = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) { l[k] *= inv; if (k >= 1) mask += l[k]; }
        if (lp == 0)
#pragma unroll
            for (int k = 0; k < K; ++k) y[(b * K + k) * HW + hw] = l[k];
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const int ch = lp + j * LPP;
            if (ch < cpr) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) xv[j][e] *= mask;
                *(uint4*)(masked + pix * C + ch * EPC) = pack16<T>(xv[j]);
            }
        }
    }
}

template <typename T, int K, int CPL>
__global__ __launch_bounds__(256) void aag_bwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ y,
                                                      const T* __restrict__ dmasked, const float* __restrict__ dy, T* __restrict__ dx,
                                                      float* __restrict__ dw, float* __restrict__ dbias, int64_t npix, int HW, int C, int LPP,
                                                      int mode, float* __restrict__ part /* [gridDim.x][K*C + K] or null */, int det) {
    constexpr int EPC = DT<T>::EPC;
    extern __shared__ float sm[];   // ws[K][C] then dwred[K][C + 1]
    float* ws = sm;
    float* dwred = sm + K * C;
    for (int i = threadIdx.x; i < K * C; i += 256) { ws[i] = w[i]; dwred[i] = 0.f; }
    if (threadIdx.x < K) dwred[K * C + threadIdx.x] = 0.f;
    __syncthreads();
    const int lp = threadIdx.x % LPP;
    const int ppb = 256 / LPP;
    const int cpr = C / EPC;
    float dwa[K][CPL][EPC];
    float dba[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        dba[k] = 0.f;
#pragma unroll
        for (int j = 0; j < CPL; ++j)
#pragma unroll
            for (int e = 0; e < EPC; ++e) dwa[k][j][e] = 0.f;
    }
    // two pixel rounds per iteration: all loads of both rounds are issued before the first dependent shuffle (the loop is
    // latency-bound otherwise); block-uniform trip count keeps whole waves in the shuffles; 32-bit pixel decode
    const int64_t G = (int64_t)gridDim.x * ppb;
    for (int64_t base = (int64_t)blockIdx.x * ppb; base < npix; base += 2 * G) {
        int64_t pixs[2];
        bool lives[2];
        float xv[2][CPL][EPC], dm[2][CPL][EPC];
        float dmask[2], ykv[2][K], dyv[2][K];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            pixs[u] = base + u * G + threadIdx.x / LPP;
            lives[u] = pixs[u] < npix;
            dmask[u] = 0.f;
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                const int ch = lp + j * LPP;
                if (lives[u] && ch < cpr) {
                    unpack16<T>(*(const uint4*)(x + pixs[u] * C + ch * EPC), xv[u][j]);
                    if (mode == 0) unpack16<T>(*(const uint4*)(dmasked + pixs[u] * C + ch * EPC), dm[u][j]);
                } else {
#pragma unroll
                    for (int e = 0; e < EPC; ++e) { xv[u][j][e] = 0.f; dm[u][j][e] = 0.f; }
                }
            }
            const unsigned p32 = lives[u] ? (unsigned)pixs[u] : 0u;
            const unsigned b = p32 / (unsigned)HW, hw = p32 - b * (unsigned)HW;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const size_t idx = ((size_t)b * K + k) * HW + hw;
                ykv[u][k] = (mode == 0 && lives[u]) ? y[idx] : 0.f;
                dyv[u][k] = (dy && lives[u]) ? dy[idx] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (mode == 0) {
#pragma unroll
                for (int j = 0; j < CPL; ++j)
#pragma unroll
                    for (int e = 0; e < EPC; ++e) dmask[u] += dm[u][j][e] * xv[u][j][e];
                for (int o = LPP >> 1; o > 0; o >>= 1) dmask[u] += __shfl_xor(dmask[u], o, 64);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!lives[u]) continue;
            const int64_t pix = pixs[u];
            float dl[K];
            float mask = 1.f;
            if (mode == 1) {
#pragma unroll
                for (int k = 0; k < K; ++k) dl[k] = dyv[u][k];
            } else {
                float g[K];
                float dot = 0.f;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    g[k] = dyv[u][k] + (k >= 1 ? dmask[u] : 0.f);
                    dot += ykv[u][k] * g[k];
                }
                mask = 1.f - ykv[u][0];
#pragma unroll
                for (int k = 0; k < K; ++k) dl[k] = ykv[u][k] * (g[k] - dot);
            }
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                const int ch = lp + j * LPP;
                if (ch < cpr) {
                    float o[EPC];
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        float v = mode == 0 ? dm[u][j][e] * mask : 0.f;
#pragma unroll
                        for (int k = 0; k < K; ++k) { v += dl[k] * ws[k * C + ch * EPC + e]; dwa[k][j][e] += dl[k] * xv[u][j][e]; }
                        o[e] = v;
                    }
                    *(uint4*)(dx + pix * C + ch * EPC) = pack16<T>(o);
                }
            }
            if (lp == 0)
#pragma unroll
                for (int k = 0; k < K; ++k) dba[k] += dl[k];
        }
    }
    // block reduction of the weight gradient through LDS atomics, then one global atomic per element
    if (det) {
        // deterministic mode: the 256 / LPP pixel slots (and the four waves' bias sums) add in slot order
        for (int slot = 0; slot < ppb; ++slot) {
            if ((int)threadIdx.x / LPP == slot) {
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    const int ch = lp + j * LPP;
                    if (ch < cpr)
#pragma unroll
                        for (int k = 0; k < K; ++k)
#pragma unroll
                            for (int e = 0; e < EPC; ++e) dwred[k * C + ch * EPC + e] += dwa[k][j][e];
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float s = wave_sum(dba[k]);
            for (int wv = 0; wv < 4; ++wv) {
                if ((int)threadIdx.x == wv * 64) dwred[K * C + k] += s;
                __syncthreads();
            }
        }
    } else {
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        const int ch = lp + j * LPP;
        if (ch < cpr)
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int e = 0; e < EPC; ++e) atomicAdd(&dwred[k * C + ch * EPC + e], dwa[k][j][e]);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float s = wave_sum(dba[k]);
        if ((threadIdx.x & 63) == 0) atomicAdd(&dwred[K * C + k], s);
    }
    }
    __syncthreads();
    if (part) {
        // per-block partials, summed by aag_partial_reduce_kernel: 1024 blocks adding into the same few cache lines with global
        // float atomics cost 50-100 us per launch (measured: 944 -> 513 us/step over the six launches without them)
        float* my = part + (size_t)blockIdx.x * (K * C + K);
        for (int i = threadIdx.x; i < K * C + K; i += 256) my[i] = dwred[i];
    } else {
        for (int i = threadIdx.x; i < K * C; i += 256) atomicAdd(dw + i, dwred[i]);
        if (threadIdx.x < K) atomicAdd(dbias + threadIdx.x, dwred[K * C + threadIdx.x]);
    }
}

// dw[i] += sum_b part[b][i] (i < nw), dbias[i - nw] += ... (nw <= i < n): 64 elements x 16 partial ranges per block
__global__ __launch_bounds__(1024) void aag_partial_reduce_kernel(const float* __restrict__ part, int nb, int n, int nw, float* __restrict__ dw,
                                                                  float* __restrict__ dbias) {
    __shared__ float red[16][64];
    const int il = threadIdx.x & 63, pr = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + il;
    const int per = (nb + 15) / 16, b0 = pr * per, b1 = min(nb, b0 + per);
    float a = 0.f;
    if (i < n) {
#pragma unroll 8
        for (int b = b0; b < b1; ++b) a += part[(size_t)b * n + i];
    }
    red[pr][il] = a;
    __syncthreads();
    if (pr == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[q][il];
        if (i < nw) dw[i] += t; else dbias[i - nw] += t;
    }
}

static int aag_lpp(int cpr, int& cpl) {
    int lpp = 1;
    while (lpp < cpr && lpp < 64) lpp <<= 1;
    cpl = cdiv(cpr, lpp);
    return lpp;
}

template <typename T, int K>
static int aag_fwd_launch(const void* x, const float* w, const float* bias, void* masked, float* y, int64_t npix, int HW, int C, int mode,
                          hipStream_t st) {
    int cpl;
    const int lpp = aag_lpp(C / DT<T>::EPC, cpl);
    const int ppb = 256 / lpp;
    int64_t nb = cdiv64(npix, ppb);
    static const int fcap = getenv("OCTA_AAG_FWD_BLOCKS") ? std::max(64, atoi(getenv("OCTA_AAG_FWD_BLOCKS"))) : 4096;
    if (nb > fcap) nb = fcap;
    const size_t sh = (size_t)K * C * sizeof(float);
#define AAG_F(CPLV) aag_fwd_kernel<T, K, CPLV><<<(int)nb, 256, sh, st>>>((const T*)x, w, bias, (T*)masked, y, npix, HW, C, lpp, mode)
    if (cpl <= 1) AAG_F(1); else if (cpl <= 2) AAG_F(2); else if (cpl <= 4) AAG_F(4);
    else OCTA_FAIL(OCTA_ERR_UNSUPPORTED, "octa_aag_fwd: C=%d too large", C);
#undef AAG_F
    OCTA_CHECK_LAUNCH("aag_fwd");
    return OCTA_OK;
}
static int64_t aag_bwd_blocks(int64_t npix, int ppb) {
    int64_t nb = cdiv64(npix, (int64_t)ppb * 16);
    // wide, low-resolution gates (1024 channels at 25 x 25: 4 pixels per block round) came out at 157 workgroups on 256 CUs with
    // 16 dependent rounds each: at least one workgroup per CU while a workgroup still has two rounds (one loop iteration)
    if (nb < 256) { const int64_t nb2 = cdiv64(npix, (int64_t)ppb * 2); nb = nb2 < 256 ? nb2 : 256; }
    static const int cap = getenv("OCTA_AAG_BWD_BLOCKS") ? std::max(64, atoi(getenv("OCTA_AAG_BWD_BLOCKS"))) : 1024;
    if (nb > cap) nb = cap;
    return nb < 1 ? 1 : nb;
}
template <typename T, int K>
static int aag_bwd_launch(const void* x, const float* w, const float* y, const void* dmasked, const float* dy, void* dx, float* dw, float* dbias,
                          int64_t npix, int HW, int C, int mode, float* part, hipStream_t st) {
    int cpl;
    const int lpp = aag_lpp(C / DT<T>::EPC, cpl);
    const int ppb = 256 / lpp;
    const int det = octa_deterministic() ? 1 : 0;
    const int64_t nb = (det && !part) ? 1 : aag_bwd_blocks(npix, ppb);      // (deterministic without the partials workspace: one workgroup)
    const size_t sh = (size_t)(2 * K * C + K) * sizeof(float);
#define AAG_B(CPLV) aag_bwd_kernel<T, K, CPLV><<<(int)nb, 256, sh, st>>>((const T*)x, w, y, (const T*)dmasked, dy, (T*)dx, dw, dbias, npix, HW, C, lpp, mode, part, det)
    if (cpl <= 1) AAG_B(1); else if (cpl <= 2) AAG_B(2); else if (cpl <= 4) AAG_B(4);
    else OCTA_FAIL(OCTA_ERR_UNSUPPORTED, "octa_aag_bwd: C=%d too large", C);
#undef AAG_B
    OCTA_CHECK_LAUNCH("aag_bwd");
    if (part) {
        const int n = K * C + K;
        aag_partial_reduce_kernel<<<cdiv(n, 64), 1024, 0, st>>>(part, (int)nb, n, K * C, dw, dbias);
        OCTA_CHECK_LAUNCH("aag_partial_reduce");
    }
    return OCTA_OK;
}

extern "C" int octa_aag_fwd(const void* x, const float* w, const float* bias, void* masked, float* y, int64_t B, int HW, int C, int K, int dtype,
                            int mode, octa_stream_t stream) {
    OCTA_REQUIRE(x && w && bias && y && (mode == 1 || masked), "octa_aag_fwd: null pointer");
    OCTA_REQUIRE(C % 8 == 0 && K >= 2 && K <= 4, "octa_aag_fwd: needs C %% 8 == 0 and 2 <= num_classes <= 4 (got C=%d K=%d)", C, K);
    OCTA_REQUIRE(OCTA_DTYPE_OK(dtype), "octa_aag_fwd: bad dtype");
    hipStream_t st = (hipStream_t)stream;
    const int64_t npix = B * HW;
#define AAG_K(KV) (dtype == OCTA_F32 ? aag_fwd_launch<float, KV>(x, w, bias, masked, y, npix, HW, C, mode, st) \
                   : dtype == OCTA_BF16 ? aag_fwd_launch<bf16_t, KV>(x, w, bias, masked, y, npix, HW, C, mode, st) \
                                        : aag_fwd_launch<f16_t, KV>(x, w, bias, masked, y, npix, HW, C, mode, st))
    if (K == 2) return AAG_K(2);
    if (K == 3) return AAG_K(3);
    return AAG_K(4);
#undef AAG_K
}
extern "C" size_t octa_aag_workspace_floats(int C, int K) { return (size_t)1024 * (size_t)(K * C + K); }
extern "C" int octa_aag_bwd(const void* x, const float* w, const float* y, const void* dmasked, const float* dy, void* dx, float* dw,
                            float* dbias, int64_t B, int HW, int C, int K, int dtype, int mode, float* part, octa_stream_t stream) {
    OCTA_REQUIRE(x && w && dx && dw && dbias && (mode == 1 || (y && dmasked)), "octa_aag_bwd: null pointer");
    OCTA_REQUIRE(C % 8 == 0 && K >= 2 && K <= 4, "octa_aag_bwd: needs C %% 8 == 0 and 2 <= num_classes <= 4");
    OCTA_REQUIRE(OCTA_DTYPE_OK(dtype), "octa_aag_bwd: bad dtype");
    hipStream_t st = (hipStream_t)stream;
    const int64_t npix = B * HW;
#define AAG_K(KV) (dtype == OCTA_F32 ? aag_bwd_launch<float, KV>(x, w, y, dmasked, dy, dx, dw, dbias, npix, HW, C, mode, part, st) \
                   : dtype == OCTA_BF16 ? aag_bwd_launch<bf16_t, KV>(x, w, y, dmasked, dy, dx, dw, dbias, npix, HW, C, mode, part, st) \
                                        : aag_bwd_launch<f16_t, KV>(x, w, y, dmasked, dy, dx, dw, dbias, npix, HW, C, mode, part, st))
    if (K == 2) return AAG_K(2);
    if (K == 3) return AAG_K(3);
    return AAG_K(4);
#undef AAG_K
}
