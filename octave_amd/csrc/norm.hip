// Training-mode BatchNorm2d on NHWC activations: statistics, apply (+residual, +ReLU), backward.
// HBM-bound: every kernel moves 16-byte chunks, consecutive lanes on consecutive chunks.
#include "common.hpp"
#include <stdlib.h>

// thread t of a 256-thread block owns chunk column (blockIdx.x*TX + t%TX) and row lane t/TX
struct ColMap { int TX, RY, gridx; };
static ColMap col_map(int cpr) {
    ColMap m;
    if (cpr >= 256) m.TX = 256;
    else if (256 % cpr == 0) m.TX = cpr;
    else m.TX = 64;
    m.RY = 256 / m.TX;
    m.gridx = cdiv(cpr, m.TX);
    return m;
}
static int rows_per_block(int64_t rows, int RY) {
    int64_t rpb = cdiv64(rows, 1024);
    if (rpb < (int64_t)RY * 16) rpb = (int64_t)RY * 16;
    return (int)rpb;
}

extern "C" size_t octa_bn_workspace_floats(int64_t rows, int C) {
    // partials [nby][2][C] with nby <= 1024 (+1 slack), then 2C finalised values
    return (size_t)(1026) * 2 * (size_t)C + 4 * (size_t)C;
}

// partial[by][0][c] = sum a, partial[by][1][c] = sum b over the block's rows, where
//   MODE 0 (stats): a = x, b = x*x
//   MODE 1 (bwd)  : a = dy', b = dy' * xhat   (dy' = dy masked by y > 0 when relu)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const T* __restrict__ x, int ldx, int xoff, const T* __restrict__ dy, int lddy,
                                                        int dyoff, const T* __restrict__ y, int ldy, int yoff,
                                                        const float* __restrict__ mean, const float* __restrict__ invstd, int relu,
                                                        const uint8_t* __restrict__ rmask, int64_t rows, int C, int TX, int rpb,
                                                        float* __restrict__ partial) {
    constexpr int EPC = DT<T>::EPC;
    extern __shared__ float red[];   // [RY][TX*EPC][2 or 3]
    const int RY = 256 / TX;
    const int cx = threadIdx.x % TX, ry = threadIdx.x / TX;
    const int col = blockIdx.x * TX + cx;
    const int cpr = C / EPC;
    const int64_t r0 = (int64_t)blockIdx.y * rpb, r1 = min(rows, r0 + rpb);
    float sa[EPC], sb[EPC], mu[EPC], is[EPC];
    int cnt = 0;
#pragma unroll
    for (int e = 0; e < EPC; ++e) { sa[e] = 0.f; sb[e] = 0.f; mu[e] = 0.f; is[e] = 1.f; }
    if (col < cpr) {
        if (MODE == 1) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) { mu[e] = mean[col * EPC + e]; is[e] = invstd[col * EPC + e]; }
        }
        for (int64_t r = r0 + ry; r < r1; r += RY) {
            float xv[EPC];
            unpack16<T>(*(const uint4*)(x + r * ldx + xoff + col * EPC), xv);
            if (MODE == 0) {
                // shifted sums (shift = this thread's first sample): no E[x^2]-E[x]^2 cancellation
                if (cnt == 0) {
#pragma unroll
                    for (int e = 0; e < EPC; ++e) mu[e] = xv[e];
                }
#pragma unroll
                for (int e = 0; e < EPC; ++e) { const float d = xv[e] - mu[e]; sa[e] += d; sb[e] += d * d; }
                ++cnt;
            } else {
                float dv[EPC], yv[EPC];
                unpack16<T>(*(const uint4*)(dy + r * lddy + dyoff + col * EPC), dv);
                if (relu && rmask) {
                    // one byte per 16-byte chunk written by bn_apply: bit e = "output e passed the ReLU" (a 1/16-size read instead of y)
                    const unsigned mb = rmask[r * cpr + col];
#pragma unroll
                    for (int e = 0; e < EPC; ++e) yv[e] = (mb >> e) & 1u ? 1.f : 0.f;
                } else if (relu) unpack16<T>(*(const uint4*)(y + r * ldy + yoff + col * EPC), yv);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float d = (relu && !(yv[e] > 0.f)) ? 0.f : dv[e];
                    sa[e] += d; sb[e] += d * (xv[e] - mu[e]) * is[e];
                }
            }
        }
    }
    if (MODE == 0) {
        // thread partial -> (mean_t, M2_t); the count rides along in a third slot
        float* myred = red + ((size_t)ry * TX + cx) * EPC * 3;
        const float fn = (float)cnt;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float m1 = cnt ? sa[e] / fn : 0.f;
            myred[3 * e] = mu[e] + m1;
            myred[3 * e + 1] = cnt ? sb[e] - sa[e] * m1 : 0.f;
            myred[3 * e + 2] = fn;
        }
        __syncthreads();
        for (int ch = threadIdx.x; ch < TX * EPC; ch += 256) {
            const int c = blockIdx.x * TX * EPC + ch;
            if (c >= C) continue;
            double n = 0.0, mean_b = 0.0, m2 = 0.0;      // Chan et al. pairwise merge
            for (int yy = 0; yy < RY; ++yy) {
                const float* pr = red + ((size_t)yy * TX * EPC + ch) * 3;
                const double nb = (double)pr[2];
                if (nb > 0.0) {
                    const double delta = (double)pr[0] - mean_b, tot = n + nb;
                    mean_b += delta * nb / tot;
                    m2 += (double)pr[1] + delta * delta * n * nb / tot;
                    n = tot;
                }
            }
            partial[((size_t)blockIdx.y * 2 + 0) * C + c] = (float)mean_b;
            partial[((size_t)blockIdx.y * 2 + 1) * C + c] = (float)m2;
        }
        return;
    }
    float* myred = red + ((size_t)ry * TX + cx) * EPC * 2;
#pragma unroll
    for (int e = 0; e < EPC; ++e) { myred[2 * e] = sa[e]; myred[2 * e + 1] = sb[e]; }
    __syncthreads();
    // threads 0 .. TX*EPC-1 finish one channel each
    for (int ch = threadIdx.x; ch < TX * EPC; ch += 256) {
        const int c = blockIdx.x * TX * EPC + ch;
        if (c >= C) continue;
        float a = 0.f, b = 0.f;
        for (int yy = 0; yy < RY; ++yy) { a += red[((size_t)yy * TX * EPC + ch) * 2]; b += red[((size_t)yy * TX * EPC + ch) * 2 + 1]; }
        partial[((size_t)blockIdx.y * 2 + 0) * C + c] = a;
        partial[((size_t)blockIdx.y * 2 + 1) * C + c] = b;
    }
}

// one workgroup per channel.  The block partials (n_b, mean_b, M2_b) are re-expressed around a common reference r (the first
// block's mean): S1 = sum n_b (mean_b - r), S2 = sum M2_b + n_b (mean_b - r)^2, after which mean = r + S1/n and
// M2 = S2 - S1^2/n (parallel-axis form of the Chan merge, evaluated in double: r is within a few sigma/sqrt(n_b) of the mean,
// so nothing cancels).  Plain sums reduce with DPP adds; the pairwise-merge butterfly this replaces spent 36 LDS-crossbar
// shuffles and 9 double divisions on the critical path of a kernel that runs 70 times per step.
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const float* __restrict__ partial, int nby, int rpb, int C, int64_t rows, float eps,
                                                               float momentum, float* __restrict__ mean, float* __restrict__ invstd,
                                                               float* __restrict__ rm, float* __restrict__ rv) {
    __shared__ double wred[4][2];
    const int c = blockIdx.x;
    const double r = (double)partial[c];                 // block 0's mean
    double s1 = 0.0, s2 = 0.0;
    for (int b = threadIdx.x; b < nby; b += 256) {
        const int64_t lo = (int64_t)b * rpb;
        const double nb = (double)((rows - lo) < rpb ? (rows - lo) : rpb);
        const double d = (double)partial[((size_t)b * 2) * C + c] - r;
        s1 += nb * d;
        s2 += (double)partial[((size_t)b * 2 + 1) * C + c] + nb * d * d;
    }
    s1 = wave_sum_d(s1);
    s2 = wave_sum_d(s2);
    if ((threadIdx.x & 63) == 0) { wred[threadIdx.x >> 6][0] = s1; wred[threadIdx.x >> 6][1] = s2; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    s1 = (wred[0][0] + wred[1][0]) + (wred[2][0] + wred[3][0]);
    s2 = (wred[0][1] + wred[1][1]) + (wred[2][1] + wred[3][1]);
    const double n = (double)rows;
    const double m = r + s1 / n;
    double var = (s2 - s1 * s1 / n) / n;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rm) rm[c] = (1.f - momentum) * rm[c] + momentum * (float)m;
    if (rv) rv[c] = (1.f - momentum) * rv[c] + momentum * (float)(rows > 1 ? var * n / (n - 1.0) : var);
}

extern "C" int octa_bn_stats(const void* x, int64_t rows, int C, int ld, int off, int dtype, float eps, float momentum, float* mean,
                             float* invstd, float* running_mean, float* running_var, float* ws, octa_stream_t stream) {
    OCTA_REQUIRE(x && mean && invstd && ws, "octa_bn_stats: null pointer");
    OCTA_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && ld % 8 == 0 && off % 8 == 0, "octa_bn_stats: C/ld/off must be multiples of 8 (C=%d ld=%d off=%d)", C, ld, off);
    OCTA_REQUIRE(OCTA_DTYPE_OK(dtype), "octa_bn_stats: bad dtype");
    hipStream_t st = (hipStream_t)stream;
    const int epc = dtype == OCTA_F32 ? 4 : 8;
    const ColMap cm = col_map(C / epc);
    const int rpb = rows_per_block(rows, cm.RY);
    const int nby = (int)cdiv64(rows, rpb);
    dim3 grid(cm.gridx, nby);
    const size_t sh = (size_t)256 * epc * 3 * sizeof(float);
    if (dtype == OCTA_F32)
        bn_reduce_kernel<float, 0><<<grid, 256, sh, st>>>((const float*)x, ld, off, nullptr, 0, 0, nullptr, 0, 0, nullptr, nullptr, 0, nullptr, rows, C, cm.TX, rpb, ws);
    else if (dtype == OCTA_BF16)
        bn_reduce_kernel<bf16_t, 0><<<grid, 256, sh, st>>>((const bf16_t*)x, ld, off, nullptr, 0, 0, nullptr, 0, 0, nullptr, nullptr, 0, nullptr, rows, C, cm.TX, rpb, ws);
    else
        bn_reduce_kernel<f16_t, 0><<<grid, 256, sh, st>>>((const f16_t*)x, ld, off, nullptr, 0, 0, nullptr, 0, 0, nullptr, nullptr, 0, nullptr, rows, C, cm.TX, rpb, ws);
    OCTA_CHECK_LAUNCH("bn_reduce(stats)");
    bn_stats_finalize_kernel<<<C, 256, 0, st>>>(ws, nby, rpb, C, rows, eps, momentum, mean, invstd, running_mean, running_var);
    OCTA_CHECK_LAUNCH("bn_stats_finalize");
    return OCTA_OK;
}

// Column-aligned grid stride: the launch makes (gridDim.x * 256) a multiple of the chunks per row, so a thread keeps ONE channel
// chunk for its whole life and loads that chunk's coefficients once.  (The one-chunk-per-thread form fetched 128 bytes of
// per-channel parameters for every 16 bytes of activations: the L1 request rate, not HBM, set its speed.)
static inline int ew_blocks_aligned(int64_t total, int cpr) {
    int64_t b = cdiv64(total, 256);
    const int64_t cap = 256 * 10;                       // ~10 resident workgroups per CU
    if (b > cap) b = cap;
    int a = cpr, g = 256;                               // unit = cpr / gcd(cpr, 256)
    while (g) { const int t = a % g; a = g; g = t; }
    const int unit = cpr / a;
    b = (b + unit - 1) / unit * unit;
    return (int)(b < 1 ? 1 : b);
}

template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, int ldx, int xoff, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const T* __restrict__ res, int ldr, int roff,
                                                       T* __restrict__ y, int ldy, int yoff, int64_t rows, int cpr, int relu,
                                                       uint8_t* __restrict__ rmask) {
    constexpr int EPC = DT<T>::EPC;
    const int64_t total = rows * cpr;
    const int64_t stride = (int64_t)gridDim.x * 256;     // multiple of cpr (ew_blocks_aligned)
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t rstep = stride / cpr;
    int64_t r = i0 / cpr;
    const int c0 = (int)(i0 - r * cpr) * EPC;
    float mu[EPC], sc[EPC], be[EPC];
#pragma unroll
    for (int k = 0; k < EPC / 4; ++k) {
        const float4 m4 = *(const float4*)(mean + c0 + 4 * k), i4 = *(const float4*)(invstd + c0 + 4 * k);
        const float4 g4 = *(const float4*)(gamma + c0 + 4 * k), b4 = *(const float4*)(beta + c0 + 4 * k);
        mu[4 * k] = m4.x; mu[4 * k + 1] = m4.y; mu[4 * k + 2] = m4.z; mu[4 * k + 3] = m4.w;
        sc[4 * k] = g4.x * i4.x; sc[4 * k + 1] = g4.y * i4.y; sc[4 * k + 2] = g4.z * i4.z; sc[4 * k + 3] = g4.w * i4.w;
        be[4 * k] = b4.x; be[4 * k + 1] = b4.y; be[4 * k + 2] = b4.z; be[4 * k + 3] = b4.w;
    }
    // block-uniform trip count: the mask bytes of 4 neighbouring lanes leave as ONE dword (byte stores ran 5x slower)
    for (int64_t base = (int64_t)blockIdx.x * 256; base < total; base += stride, r += rstep) {
        const int64_t i = base + threadIdx.x;
        const bool live = i < total;
        const int64_t rr_ = live ? r : 0;
        float v[EPC], rr[EPC];
        unsigned mb = 0u;
        unpack16<T>(*(const uint4*)(x + rr_ * ldx + xoff + c0), v);
        if (res) unpack16<T>(*(const uint4*)(res + rr_ * ldr + roff + c0), rr);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            float o = (v[e] - mu[e]) * sc[e] + be[e];
            if (res) o += rr[e];
            if (relu) { if (o > 0.f) mb |= 1u << e; else o = 0.f; }
            v[e] = o;
        }
        if (live) *(uint4*)(y + rr_ * ldy + yoff + c0) = pack16<T>(v);
        if (rmask) {
            unsigned w = live ? (mb << (8 * (threadIdx.x & 3))) : 0u;
            w |= (unsigned)__builtin_amdgcn_mov_dpp((int)w, 0xB1, 0xF, 0xF, true);      // lane ^ 1
            w |= (unsigned)__builtin_amdgcn_mov_dpp((int)w, 0x4E, 0xF, 0xF, true);      // lane ^ 2
            if ((threadIdx.x & 3) == 0 && live) *(unsigned*)(rmask + i) = w;     // buffer is padded to a multiple of 4 bytes
        }
    }
}

extern "C" int octa_bn_apply(const void* x, int ldx, int xoff, const float* mean, const float* invstd, const float* gamma,
                             const float* beta, const void* residual, int ldr, int roff, void* y, int ldy, int yoff, int64_t rows,
                             int C, int dtype, int relu, uint8_t* relu_mask, octa_stream_t stream) {
    OCTA_REQUIRE(x && y && mean && invstd && gamma && beta, "octa_bn_apply: null pointer");
    OCTA_REQUIRE(C % 8 == 0 && ldx % 8 == 0 && xoff % 8 == 0 && ldy % 8 == 0 && yoff % 8 == 0 && (!residual || (ldr % 8 == 0 && roff % 8 == 0)),
                 "octa_bn_apply: C/ld/off must be multiples of 8");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == OCTA_F32)
        bn_apply_kernel<float><<<ew_blocks_aligned(rows * (C / 4), C / 4), 256, 0, st>>>((const float*)x, ldx, xoff, mean, invstd, gamma, beta, (const float*)residual, ldr, roff, (float*)y, ldy, yoff, rows, C / 4, relu, relu_mask);
    else if (dtype == OCTA_BF16)
        bn_apply_kernel<bf16_t><<<ew_blocks_aligned(rows * (C / 8), C / 8), 256, 0, st>>>((const bf16_t*)x, ldx, xoff, mean, invstd, gamma, beta, (const bf16_t*)residual, ldr, roff, (bf16_t*)y, ldy, yoff, rows, C / 8, relu, relu_mask);
    else if (dtype == OCTA_F16)
        bn_apply_kernel<f16_t><<<ew_blocks_aligned(rows * (C / 8), C / 8), 256, 0, st>>>((const f16_t*)x, ldx, xoff, mean, invstd, gamma, beta, (const f16_t*)residual, ldr, roff, (f16_t*)y, ldy, yoff, rows, C / 8, relu, relu_mask);
    else OCTA_FAIL(OCTA_ERR_BAD_ARG, "octa_bn_apply: bad dtype");
    OCTA_CHECK_LAUNCH("bn_apply");
    return OCTA_OK;
}

// ws layout after finalize: fin[0][c] = sum dy' / N ; fin[1][c] = sum dy' xhat / N
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nby, int C, int64_t rows, float* __restrict__ fin,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ double wred[4][2];
    const int c = blockIdx.x;
    double s = 0.0, ss = 0.0;
    for (int b = threadIdx.x; b < nby; b += 256) { s += (double)partial[((size_t)b * 2) * C + c]; ss += (double)partial[((size_t)b * 2 + 1) * C + c]; }
    s = wave_sum_d(s);
    ss = wave_sum_d(ss);
    if ((threadIdx.x & 63) == 0) { wred[threadIdx.x >> 6][0] = s; wred[threadIdx.x >> 6][1] = ss; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    s = (wred[0][0] + wred[1][0]) + (wred[2][0] + wred[3][0]);
    ss = (wred[0][1] + wred[1][1]) + (wred[2][1] + wred[3][1]);
    if (dbeta) dbeta[c] += (float)s;
    if (dgamma) dgamma[c] += (float)ss;
    fin[c] = (float)(s / (double)rows);
    fin[C + c] = (float)(ss / (double)rows);
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, int lddy, int dyoff, const T* __restrict__ x, int ldx,
                                                           int xoff, const T* __restrict__ y, int ldy, int yoff,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ fin,
                                                           const uint8_t* __restrict__ rmask, T* __restrict__ dx, int lddx, int dxoff, T* __restrict__ dres, int lddr,
                                                           int droff, int64_t rows, int C, int relu) {
    constexpr int EPC = DT<T>::EPC;
    const int cpr = C / EPC;
    const int64_t total = rows * cpr;
    const int64_t stride = (int64_t)gridDim.x * 256;     // multiple of cpr: one channel chunk per thread (see bn_apply_kernel)
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t rstep = stride / cpr;
    int64_t r = i0 / cpr;
    const int c0 = (int)(i0 - r * cpr) * EPC;
    float mu[EPC], isd[EPC], gi[EPC], f0[EPC], f1[EPC];
#pragma unroll
    for (int k = 0; k < EPC / 4; ++k) {
        const float4 m4 = *(const float4*)(mean + c0 + 4 * k), i4 = *(const float4*)(invstd + c0 + 4 * k), g4 = *(const float4*)(gamma + c0 + 4 * k);
        const float4 a4 = *(const float4*)(fin + c0 + 4 * k), b4 = *(const float4*)(fin + C + c0 + 4 * k);
        mu[4 * k] = m4.x; mu[4 * k + 1] = m4.y; mu[4 * k + 2] = m4.z; mu[4 * k + 3] = m4.w;
        isd[4 * k] = i4.x; isd[4 * k + 1] = i4.y; isd[4 * k + 2] = i4.z; isd[4 * k + 3] = i4.w;
        gi[4 * k] = g4.x * i4.x; gi[4 * k + 1] = g4.y * i4.y; gi[4 * k + 2] = g4.z * i4.z; gi[4 * k + 3] = g4.w * i4.w;
        f0[4 * k] = a4.x; f0[4 * k + 1] = a4.y; f0[4 * k + 2] = a4.z; f0[4 * k + 3] = a4.w;
        f1[4 * k] = b4.x; f1[4 * k + 1] = b4.y; f1[4 * k + 2] = b4.z; f1[4 * k + 3] = b4.w;
    }
    for (int64_t i = i0; i < total; i += stride, r += rstep) {
        float dv[EPC], xv[EPC], yv[EPC], o[EPC];
        unpack16<T>(*(const uint4*)(dy + r * lddy + dyoff + c0), dv);
        unpack16<T>(*(const uint4*)(x + r * ldx + xoff + c0), xv);
        if (relu && rmask) {
            const unsigned mb = rmask[i];
#pragma unroll
            for (int e = 0; e < EPC; ++e) yv[e] = (mb >> e) & 1u ? 1.f : 0.f;
        } else if (relu) unpack16<T>(*(const uint4*)(y + r * ldy + yoff + c0), yv);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float d = (relu && !(yv[e] > 0.f)) ? 0.f : dv[e];
            dv[e] = d;
            const float xh = (xv[e] - mu[e]) * isd[e];
            o[e] = gi[e] * (d - f0[e] - xh * f1[e]);
        }
        *(uint4*)(dx + r * lddx + dxoff + c0) = pack16<T>(o);
        if (dres) *(uint4*)(dres + r * lddr + droff + c0) = pack16<T>(dv);
    }
}

extern "C" int octa_bn_bwd(const void* dy, int lddy, int dyoff, const void* x, int ldx, int xoff, const void* y, int ldy, int yoff,
                           const float* mean, const float* invstd, const float* gamma, void* dx, int lddx, int dxoff, void* dres,
                           int lddr, int droff, float* dgamma, float* dbeta, int64_t rows, int C, int dtype, int relu,
                           const uint8_t* relu_mask, float* ws, octa_stream_t stream) {
    OCTA_REQUIRE(dy && x && mean && invstd && gamma && dx && ws, "octa_bn_bwd: null pointer");
    OCTA_REQUIRE(!relu || y || relu_mask, "octa_bn_bwd: relu needs the forward output or the mask octa_bn_apply wrote");
    OCTA_REQUIRE(C % 8 == 0 && lddy % 8 == 0 && dyoff % 8 == 0 && ldx % 8 == 0 && xoff % 8 == 0 && lddx % 8 == 0 && dxoff % 8 == 0,
                 "octa_bn_bwd: C/ld/off must be multiples of 8");
    OCTA_REQUIRE(OCTA_DTYPE_OK(dtype), "octa_bn_bwd: bad dtype");
    hipStream_t st = (hipStream_t)stream;
    const int epc = dtype == OCTA_F32 ? 4 : 8;
    const ColMap cm = col_map(C / epc);
    const int rpb = rows_per_block(rows, cm.RY);
    const int nby = (int)cdiv64(rows, rpb);
    dim3 grid(cm.gridx, nby);
    const size_t sh = (size_t)256 * epc * 2 * sizeof(float);
    float* fin = ws + (size_t)1026 * 2 * C;
    if (dtype == OCTA_F32)
        bn_reduce_kernel<float, 1><<<grid, 256, sh, st>>>((const float*)x, ldx, xoff, (const float*)dy, lddy, dyoff, (const float*)y, ldy, yoff, mean, invstd, relu, relu_mask, rows, C, cm.TX, rpb, ws);
    else if (dtype == OCTA_BF16)
        bn_reduce_kernel<bf16_t, 1><<<grid, 256, sh, st>>>((const bf16_t*)x, ldx, xoff, (const bf16_t*)dy, lddy, dyoff, (const bf16_t*)y, ldy, yoff, mean, invstd, relu, relu_mask, rows, C, cm.TX, rpb, ws);
    else
        bn_reduce_kernel<f16_t, 1><<<grid, 256, sh, st>>>((const f16_t*)x, ldx, xoff, (const f16_t*)dy, lddy, dyoff, (const f16_t*)y, ldy, yoff, mean, invstd, relu, relu_mask, rows, C, cm.TX, rpb, ws);
    OCTA_CHECK_LAUNCH("bn_reduce(bwd)");
    bn_bwd_finalize_kernel<<<C, 256, 0, st>>>(ws, nby, C, rows, fin, dgamma, dbeta);
    OCTA_CHECK_LAUNCH("bn_bwd_finalize");
    if (dtype == OCTA_F32)
        bn_bwd_apply_kernel<float><<<ew_blocks_aligned(rows * (C / 4), C / 4), 256, 0, st>>>((const float*)dy, lddy, dyoff, (const float*)x, ldx, xoff, (const float*)y, ldy, yoff, mean, invstd, gamma, fin, relu_mask, (float*)dx, lddx, dxoff, (float*)dres, lddr, droff, rows, C, relu);
    else if (dtype == OCTA_BF16)
        bn_bwd_apply_kernel<bf16_t><<<ew_blocks_aligned(rows * (C / 8), C / 8), 256, 0, st>>>((const bf16_t*)dy, lddy, dyoff, (const bf16_t*)x, ldx, xoff, (const bf16_t*)y, ldy, yoff, mean, invstd, gamma, fin, relu_mask, (bf16_t*)dx, lddx, dxoff, (bf16_t*)dres, lddr, droff, rows, C, relu);
    else
        bn_bwd_apply_kernel<f16_t><<<ew_blocks_aligned(rows * (C / 8), C / 8), 256, 0, st>>>((const f16_t*)dy, lddy, dyoff, (const f16_t*)x, ldx, xoff, (const f16_t*)y, ldy, yoff, mean, invstd, gamma, fin, relu_mask, (f16_t*)dx, lddx, dxoff, (f16_t*)dres, lddr, droff, rows, C, relu);
    OCTA_CHECK_LAUNCH("bn_bwd_apply");
    return OCTA_OK;
}
