// Pooling, layout conversion, channel copies, elementwise helpers, fused Adam.  All HBM-bound.
#include "common.hpp"

static inline int ew_blocks(int64_t n) { int64_t b = cdiv64(n, 256); return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }
#define DISPATCH_T(dtype, NAME, ...)                                                         \
    if ((dtype) == OCTA_F32) { using T = float; __VA_ARGS__ }                                \
    else if ((dtype) == OCTA_BF16) { using T = bf16_t; __VA_ARGS__ }                         \
    else if ((dtype) == OCTA_F16) { using T = f16_t; __VA_ARGS__ }                           \
    else OCTA_FAIL(OCTA_ERR_BAD_ARG, NAME ": bad dtype %d", (int)(dtype));

// ------------------------------------------------------------------------------------------ maxpool 3x3 s2 p1
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ am, int B, int H,
                                                          int W, int C, int OH, int OW) {
    constexpr int EPC = DT<T>::EPC;
    const int cpr = C / EPC;
    const int64_t total = (int64_t)B * OH * OW * cpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c0 = (int)(i % cpr) * EPC;
        int64_t p = i / cpr;
        const int ow = (int)(p % OW); p /= OW;
        const int oh = (int)(p % OH);
        const int b = (int)(p / OH);
        float best[EPC];
        int bi[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) { best[e] = -INFINITY; bi[e] = -1; }
        // scan order kh, kw; first strict maximum wins (ATen max_pool2d CPU tie rule)
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * 2 - 1 + kh;
            if ((unsigned)ih >= (unsigned)H) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = ow * 2 - 1 + kw;
                if ((unsigned)iw >= (unsigned)W) continue;
                float v[EPC];
                unpack16<T>(*(const uint4*)(x + ((int64_t)(b * H + ih) * W + iw) * C + c0), v);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    if (bi[e] < 0 || v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bi[e] = kh * 3 + kw; }
                }
            }
        }
        const int64_t o = ((int64_t)(b * OH + oh) * OW + ow) * C + c0;
        *(uint4*)(y + o) = pack16<T>(best);
#pragma unroll
        for (int e = 0; e < EPC; ++e) am[o + e] = (uint8_t)bi[e];
    }
}
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ am, T* __restrict__ dx, int B,
                                                          int H, int W, int C, int OH, int OW, const T* __restrict__ add, int ldadd) {
    constexpr int EPC = DT<T>::EPC;
    const int cpr = C / EPC;
    const int64_t total = (int64_t)B * H * W * cpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c0 = (int)(i % cpr) * EPC;
        int64_t p = i / cpr;
        const int iw = (int)(p % W); p /= W;
        const int ih = (int)(p % H);
        const int b = (int)(p / H);
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
        // fan-out gradient sum: the other consumer's gradient of the pooled tensor (any per-pixel stride) starts the accumulator
        if (add) unpack16<T>(*(const uint4*)(add + ((int64_t)(b * H + ih) * W + iw) * ldadd + c0), acc);
        // windows (oh, ow) with ih = 2*oh - 1 + kh
        for (int kh = 0; kh < 3; ++kh) {
            const int th = ih + 1 - kh;
            if (th < 0 || (th & 1)) continue;
            const int oh = th >> 1;
            if (oh >= OH) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int tw = iw + 1 - kw;
                if (tw < 0 || (tw & 1)) continue;
                const int ow = tw >> 1;
                if (ow >= OW) continue;
                const int64_t o = ((int64_t)(b * OH + oh) * OW + ow) * C + c0;
                float d[EPC];
                unpack16<T>(*(const uint4*)(dy + o), d);
#pragma unroll
                for (int e = 0; e < EPC; ++e) if (am[o + e] == kh * 3 + kw) acc[e] += d[e];
            }
        }
        *(uint4*)(dx + ((int64_t)(b * H + ih) * W + iw) * C + c0) = pack16<T>(acc);
    }
}
extern "C" int octa_maxpool3s2_fwd(const void* x, void* y, uint8_t* argmax, int B, int H, int W, int C, int OH, int OW, int dtype,
                                   octa_stream_t stream) {
    OCTA_REQUIRE(x && y && argmax && C % 8 == 0, "octa_maxpool3s2_fwd: bad arguments (C %% 8)");
    OCTA_REQUIRE(OH == (H + 2 - 3) / 2 + 1 && OW == (W + 2 - 3) / 2 + 1, "octa_maxpool3s2_fwd: OH/OW mismatch");
    DISPATCH_T(dtype, "octa_maxpool3s2_fwd", maxpool_fwd_kernel<T><<<ew_blocks((int64_t)B * OH * OW * C / DT<T>::EPC), 256, 0, (hipStream_t)stream>>>((const T*)x, (T*)y, argmax, B, H, W, C, OH, OW);)
    OCTA_CHECK_LAUNCH("maxpool_fwd");
    return OCTA_OK;
}
extern "C" int octa_maxpool3s2_bwd_add(const void* dy, const uint8_t* argmax, void* dx, const void* addend, int ld_addend, int B, int H, int W, int C,
                                       int OH, int OW, int dtype, octa_stream_t stream) {
    OCTA_REQUIRE(dy && dx && argmax && C % 8 == 0, "octa_maxpool3s2_bwd_add: bad arguments");
    OCTA_REQUIRE(!addend || (ld_addend >= C && ld_addend % 8 == 0 && ((uintptr_t)addend & 15) == 0), "octa_maxpool3s2_bwd_add: addend stride / alignment");
    DISPATCH_T(dtype, "octa_maxpool3s2_bwd_add", maxpool_bwd_kernel<T><<<ew_blocks((int64_t)B * H * W * C / DT<T>::EPC), 256, 0, (hipStream_t)stream>>>((const T*)dy, argmax, (T*)dx, B, H, W, C, OH, OW, (const T*)addend, ld_addend);)
    OCTA_CHECK_LAUNCH("maxpool_bwd");
    return OCTA_OK;
}
extern "C" int octa_maxpool3s2_bwd(const void* dy, const uint8_t* argmax, void* dx, int B, int H, int W, int C, int OH, int OW, int dtype,
                                   octa_stream_t stream) {
    OCTA_REQUIRE(dy && dx && argmax && C % 8 == 0, "octa_maxpool3s2_bwd: bad arguments");
    DISPATCH_T(dtype, "octa_maxpool3s2_bwd", maxpool_bwd_kernel<T><<<ew_blocks((int64_t)B * H * W * C / DT<T>::EPC), 256, 0, (hipStream_t)stream>>>((const T*)dy, argmax, (T*)dx, B, H, W, C, OH, OW, (const T*)nullptr, 0);)
    OCTA_CHECK_LAUNCH("maxpool_bwd");
    return OCTA_OK;
}

// ------------------------------------------------------------------------------------------ avgpool (ATen divisor rules)
__device__ __forceinline__ float avg_divisor(int o, int k, int s, int p, int L, int cip, int& lo, int& hi) {
    int st = o * s - p;
    int en = min(st + k, L + p);
    const int pool = en - st;
    lo = max(st, 0);
    hi = min(en, L);
    return (float)(cip ? pool : (hi - lo));
}
template <typename T>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, int OH, int OW,
                                                          int k, int s, int p, int cip) {
    constexpr int EPC = DT<T>::EPC;
    const int cpr = C / EPC;
    const int64_t total = (int64_t)B * OH * OW * cpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c0 = (int)(i % cpr) * EPC;
        int64_t q = i / cpr;
        const int ow = (int)(q % OW); q /= OW;
        const int oh = (int)(q % OH);
        const int b = (int)(q / OH);
        int h0, h1, w0, w1;
        const float dh = avg_divisor(oh, k, s, p, H, cip, h0, h1);
        const float dw = avg_divisor(ow, k, s, p, W, cip, w0, w1);
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
        for (int ih = h0; ih < h1; ++ih)
            for (int iw = w0; iw < w1; ++iw) {
                float v[EPC];
                unpack16<T>(*(const uint4*)(x + ((int64_t)(b * H + ih) * W + iw) * C + c0), v);
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc[e] += v[e];
            }
        const float inv = 1.f / (dh * dw);
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] *= inv;
        *(uint4*)(y + ((int64_t)(b * OH + oh) * OW + ow) * C + c0) = pack16<T>(acc);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B, int H, int W, int C, int OH, int OW,
                                                          int k, int s, int p, int cip, const T* __restrict__ add, int ldadd) {
    constexpr int EPC = DT<T>::EPC;
    const int cpr = C / EPC;
    const int64_t total = (int64_t)B * H * W * cpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c0 = (int)(i % cpr) * EPC;
        int64_t q = i / cpr;
        const int iw = (int)(q % W); q /= W;
        const int ih = (int)(q % H);
        const int b = (int)(q / H);
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
        if (add) unpack16<T>(*(const uint4*)(add + ((int64_t)(b * H + ih) * W + iw) * ldadd + c0), acc);     // (see maxpool_bwd_kernel)
        // windows containing (ih, iw): oh in [ceil((ih+p-k+1)/s), floor((ih+p)/s)]
        const int oh_hi = min((ih + p) / s, OH - 1), ow_hi = min((iw + p) / s, OW - 1);
        const int th = ih + p - k + 1, tw = iw + p - k + 1;
        const int oh_lo = th > 0 ? (th + s - 1) / s : 0, ow_lo = tw > 0 ? (tw + s - 1) / s : 0;
        for (int oh = oh_lo; oh <= oh_hi; ++oh) {
            int h0, h1;
            const float dh = avg_divisor(oh, k, s, p, H, cip, h0, h1);
            for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                int w0, w1;
                const float dw = avg_divisor(ow, k, s, p, W, cip, w0, w1);
                float d[EPC];
                unpack16<T>(*(const uint4*)(dy + ((int64_t)(b * OH + oh) * OW + ow) * C + c0), d);
                const float inv = 1.f / (dh * dw);
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc[e] += d[e] * inv;
            }
        }
        *(uint4*)(dx + ((int64_t)(b * H + ih) * W + iw) * C + c0) = pack16<T>(acc);
    }
}
extern "C" int octa_avgpool_fwd(const void* x, void* y, int B, int H, int W, int C, int OH, int OW, int k, int stride, int pad,
                                int count_include_pad, int dtype, octa_stream_t stream) {
    OCTA_REQUIRE(x && y && C % 8 == 0 && k > 0 && stride > 0, "octa_avgpool_fwd: bad arguments");
    DISPATCH_T(dtype, "octa_avgpool_fwd", avgpool_fwd_kernel<T><<<ew_blocks((int64_t)B * OH * OW * C / DT<T>::EPC), 256, 0, (hipStream_t)stream>>>((const T*)x, (T*)y, B, H, W, C, OH, OW, k, stride, pad, count_include_pad);)
    OCTA_CHECK_LAUNCH("avgpool_fwd");
    return OCTA_OK;
}
extern "C" int octa_avgpool_bwd_add(const void* dy, void* dx, const void* addend, int ld_addend, int B, int H, int W, int C, int OH, int OW, int k,
                                    int stride, int pad, int count_include_pad, int dtype, octa_stream_t stream) {
    OCTA_REQUIRE(dy && dx && C % 8 == 0 && k > 0 && stride > 0, "octa_avgpool_bwd_add: bad arguments");
    OCTA_REQUIRE(!addend || (ld_addend >= C && ld_addend % 8 == 0 && ((uintptr_t)addend & 15) == 0), "octa_avgpool_bwd_add: addend stride / alignment");
    DISPATCH_T(dtype, "octa_avgpool_bwd_add", avgpool_bwd_kernel<T><<<ew_blocks((int64_t)B * H * W * C / DT<T>::EPC), 256, 0, (hipStream_t)stream>>>((const T*)dy, (T*)dx, B, H, W, C, OH, OW, k, stride, pad, count_include_pad, (const T*)addend, ld_addend);)
    OCTA_CHECK_LAUNCH("avgpool_bwd");
    return OCTA_OK;
}
extern "C" int octa_avgpool_bwd(const void* dy, void* dx, int B, int H, int W, int C, int OH, int OW, int k, int stride, int pad,
                                int count_include_pad, int dtype, octa_stream_t stream) {
    OCTA_REQUIRE(dy && dx && C % 8 == 0 && k > 0 && stride > 0, "octa_avgpool_bwd: bad arguments");
    DISPATCH_T(dtype, "octa_avgpool_bwd", avgpool_bwd_kernel<T><<<ew_blocks((int64_t)B * H * W * C / DT<T>::EPC), 256, 0, (hipStream_t)stream>>>((const T*)dy, (T*)dx, B, H, W, C, OH, OW, k, stride, pad, count_include_pad, (const T*)nullptr, 0);)
    OCTA_CHECK_LAUNCH("avgpool_bwd");
    return OCTA_OK;
}

// ------------------------------------------------------------------------------------------ layout conversion
// 32x32 LDS tile transpose between the (c, hw) planes of NCHW and the (hw, c) rows of NHWC.
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, int64_t sb, int64_t sc, int64_t sh, int64_t sw, T* __restrict__ dst,
                                    int C, int H, int W, int ld, int off, int cpad) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int HW = H * W;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += 8) {        // read: x -> pixel (contiguous in NCHW)
        const int c = c0 + j, p = p0 + threadIdx.x;
        float v = 0.f;
        if (c < C && p < HW) v = src[b * sb + c * sc + (int64_t)(p / W) * sh + (int64_t)(p % W) * sw];
        tile[j][threadIdx.x] = v;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) {        // write: x -> channel (contiguous in NHWC)
        const int p = p0 + j, c = c0 + threadIdx.x;
        if (p < HW && c < cpad) DT<T>::st(dst + ((int64_t)b * HW + p) * ld + off + c, c < C ? tile[threadIdx.x][j] : 0.f);
    }
}
// few channels (the 3-channel image, the 2-channel class maps of the discriminator): one thread per pixel reads its C planes
// (consecutive threads = consecutive pixels) and writes the pixel's cpad channels; the 32 x 32 transpose above spends a 256-thread
// workgroup on 32 pixels there (79 us for the 16 x 3 x 400 x 400 input)
template <typename T, int CP>
__global__ __launch_bounds__(256) void nchw_to_nhwc_few_kernel(const float* __restrict__ src, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                                                               T* __restrict__ dst, int C, int H, int W, int ld, int off, int cpad, int64_t npix) {
    const int HW = H * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
        const int b = (int)(i / HW), p = (int)(i - (int64_t)b * HW);
        const float* s = src + b * sb + (int64_t)(p / W) * sh + (int64_t)(p % W) * sw;
        float v[CP];
#pragma unroll
        for (int c = 0; c < CP; ++c) v[c] = c < C ? s[c * sc] : 0.f;
        T* d = dst + i * ld + off;
        if (sizeof(T) == 2 && CP == 8 && cpad == 8 && (((uintptr_t)d) & 15) == 0) *(uint4*)d = pack16<T>(v);
        else {
#pragma unroll
            for (int c = 0; c < CP; ++c) if (c < cpad) DT<T>::st(d + c, v[c]);
        }
    }
}
extern "C" int octa_nchw_to_nhwc(const float* src, int64_t sb, int64_t sc, int64_t sh, int64_t sw, void* dst, int B, int C, int H, int W,
                                 int ld, int off, int cpad, int dtype, octa_stream_t stream) {
    OCTA_REQUIRE(src && dst && B > 0 && C > 0 && cpad >= C && off + cpad <= ld, "octa_nchw_to_nhwc: bad arguments");
    if (cpad <= 8) {
        const int64_t npix = (int64_t)B * H * W;
        const int blocks = ew_blocks(npix);
        DISPATCH_T(dtype, "octa_nchw_to_nhwc", nchw_to_nhwc_few_kernel<T, 8><<<blocks, 256, 0, (hipStream_t)stream>>>(src, sb, sc, sh, sw, (T*)dst, C, H, W, ld, off, cpad, npix);)
        OCTA_CHECK_LAUNCH("nchw_to_nhwc(few)");
        return OCTA_OK;
    }
    dim3 grid(cdiv(H * W, 32), cdiv(cpad, 32), B), block(32, 8);
    DISPATCH_T(dtype, "octa_nchw_to_nhwc", nchw_to_nhwc_kernel<T><<<grid, block, 0, (hipStream_t)stream>>>(src, sb, sc, sh, sw, (T*)dst, C, H, W, ld, off, cpad);)
    OCTA_CHECK_LAUNCH("nchw_to_nhwc");
    return OCTA_OK;
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, int ld, int off, float* __restrict__ dst, int C, int HW, int accumulate) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int p = p0 + j, c = c0 + threadIdx.x;
        tile[j][threadIdx.x] = (p < HW && c < C) ? DT<T>::ld(src + ((int64_t)b * HW + p) * ld + off + c) : 0.f;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int c = c0 + j, p = p0 + threadIdx.x;
        if (c < C && p < HW) {
            float* d = dst + ((int64_t)b * C + c) * HW + p;
            *d = accumulate ? (*d + tile[threadIdx.x][j]) : tile[threadIdx.x][j];
        }
    }
}
// few channels: one thread per pixel, C plane stores (consecutive threads = consecutive pixels of a plane)
template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_few_kernel(const T* __restrict__ src, int ld, int off, float* __restrict__ dst, int C, int HW,
                                                               int accumulate, int64_t npix) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
        const int b = (int)(i / HW), p = (int)(i - (int64_t)b * HW);
        const T* s = src + i * ld + off;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (c >= C) break;
            float* d = dst + ((int64_t)b * C + c) * HW + p;
            const float v = DT<T>::ld(s + c);
            *d = accumulate ? (*d + v) : v;
        }
    }
}
extern "C" int octa_nhwc_to_nchw(const void* src, int ld, int off, int dtype, float* dst, int B, int C, int H, int W, int accumulate,
                                 octa_stream_t stream) {
    OCTA_REQUIRE(src && dst && B > 0 && C > 0, "octa_nhwc_to_nchw: bad arguments");
    if (C <= 8) {
        const int64_t npix = (int64_t)B * H * W;
        DISPATCH_T(dtype, "octa_nhwc_to_nchw", nhwc_to_nchw_few_kernel<T><<<ew_blocks(npix), 256, 0, (hipStream_t)stream>>>((const T*)src, ld, off, dst, C, H * W, accumulate, npix);)
        OCTA_CHECK_LAUNCH("nhwc_to_nchw(few)");
        return OCTA_OK;
    }
    dim3 grid(cdiv(H * W, 32), cdiv(C, 32), B), block(32, 8);
    DISPATCH_T(dtype, "octa_nhwc_to_nchw", nhwc_to_nchw_kernel<T><<<grid, block, 0, (hipStream_t)stream>>>((const T*)src, ld, off, dst, C, H * W, accumulate);)
    OCTA_CHECK_LAUNCH("nhwc_to_nchw");
    return OCTA_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void copy_channels_kernel(const T* __restrict__ src, int Hs, int Ws, int lds, int soff, T* __restrict__ dst,
                                                            int Hd, int Wd, int ldd, int doff, int B, int cpr, int accumulate) {
    constexpr int EPC = DT<T>::EPC;
    const int64_t total = (int64_t)B * Hd * Wd * cpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c0 = (int)(i % cpr) * EPC;
        int64_t q = i / cpr;
        const int w = (int)(q % Wd); q /= Wd;
        const int h = (int)(q % Hd);
        const int b = (int)(q / Hd);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (h < Hs && w < Ws) v = *(const uint4*)(src + ((int64_t)(b * Hs + h) * Ws + w) * lds + soff + c0);
        T* d = dst + ((int64_t)(b * Hd + h) * Wd + w) * ldd + doff + c0;
        if (accumulate) {
            float a[EPC], c[EPC];
            unpack16<T>(v, a);
            unpack16<T>(*(const uint4*)d, c);
#pragma unroll
            for (int e = 0; e < EPC; ++e) a[e] += c[e];
            v = pack16<T>(a);
        }
        *(uint4*)d = v;
    }
}
extern "C" int octa_copy_channels(const void* src, int Hs, int Ws, int lds, int soff, void* dst, int Hd, int Wd, int ldd, int doff, int B,
                                  int C, int dtype, int accumulate, octa_stream_t stream) {
    OCTA_REQUIRE(src && dst && C % 8 == 0 && lds % 8 == 0 && soff % 8 == 0 && ldd % 8 == 0 && doff % 8 == 0, "octa_copy_channels: C/ld/off %% 8");
    DISPATCH_T(dtype, "octa_copy_channels", copy_channels_kernel<T><<<ew_blocks((int64_t)B * Hd * Wd * C / DT<T>::EPC), 256, 0, (hipStream_t)stream>>>((const T*)src, Hs, Ws, lds, soff, (T*)dst, Hd, Wd, ldd, doff, B, C / DT<T>::EPC, accumulate);)
    OCTA_CHECK_LAUNCH("copy_channels");
    return OCTA_OK;
}

// ------------------------------------------------------------------------------------------ elementwise
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_kernel(const T* __restrict__ y, const T* __restrict__ dy, T* __restrict__ dx, int64_t n, int act) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float yy = DT<T>::ld(y + i), d = DT<T>::ld(dy + i);
        float g;
        switch (act) {
            case OCTA_ACT_RELU: g = yy > 0.f ? d : 0.f; break;
            case OCTA_ACT_LEAKY02: g = yy > 0.f ? d : 0.2f * d; break;
            case OCTA_ACT_SIGMOID: g = d * yy * (1.f - yy); break;
            case OCTA_ACT_TANH: g = d * (1.f - yy * yy); break;
            default: g = d;
        }
        DT<T>::st(dx + i, g);
    }
}
extern "C" int octa_act_bwd(const void* y, const void* dy, void* dx, int64_t n, int act, int dtype, octa_stream_t stream) {
    OCTA_REQUIRE(y && dy && dx && n > 0, "octa_act_bwd: bad arguments");
    DISPATCH_T(dtype, "octa_act_bwd", act_bwd_kernel<T><<<ew_blocks(n), 256, 0, (hipStream_t)stream>>>((const T*)y, (const T*)dy, (T*)dx, n, act);)
    OCTA_CHECK_LAUNCH("act_bwd");
    return OCTA_OK;
}
template <typename T>
__global__ __launch_bounds__(256) void relu_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = DT<T>::ld(x + i);
        DT<T>::st(y + i, v > 0.f ? v : 0.f);
    }
}
extern "C" int octa_relu_fwd(const void* x, void* y, int64_t n, int dtype, octa_stream_t stream) {
    OCTA_REQUIRE(x && y && n > 0, "octa_relu_fwd: bad arguments");
    DISPATCH_T(dtype, "octa_relu_fwd", relu_kernel<T><<<ew_blocks(n), 256, 0, (hipStream_t)stream>>>((const T*)x, (T*)y, n);)
    OCTA_CHECK_LAUNCH("relu");
    return OCTA_OK;
}
template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) DT<T>::st(o + i, DT<T>::ld(a + i) + DT<T>::ld(b + i));
}
extern "C" int octa_add(const void* a, const void* b, void* out, int64_t n, int dtype, octa_stream_t stream) {
    OCTA_REQUIRE(a && b && out && n > 0, "octa_add: bad arguments");
    DISPATCH_T(dtype, "octa_add", add_kernel<T><<<ew_blocks(n), 256, 0, (hipStream_t)stream>>>((const T*)a, (const T*)b, (T*)out, n);)
    OCTA_CHECK_LAUNCH("add");
    return OCTA_OK;
}
template <typename S, typename D>
__global__ __launch_bounds__(256) void cast_kernel(const S* __restrict__ s, D* __restrict__ d, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) DT<D>::st(d + i, DT<S>::ld(s + i));
}
extern "C" int octa_cast(const void* src, int sd, void* dst, int dd, int64_t n, octa_stream_t stream) {
    OCTA_REQUIRE(src && dst && n > 0, "octa_cast: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (sd == OCTA_F32 && dd == OCTA_BF16) cast_kernel<float, bf16_t><<<ew_blocks(n), 256, 0, st>>>((const float*)src, (bf16_t*)dst, n);
    else if (sd == OCTA_BF16 && dd == OCTA_F32) cast_kernel<bf16_t, float><<<ew_blocks(n), 256, 0, st>>>((const bf16_t*)src, (float*)dst, n);
    else if (sd == OCTA_F32 && dd == OCTA_F32) cast_kernel<float, float><<<ew_blocks(n), 256, 0, st>>>((const float*)src, (float*)dst, n);
    else if (sd == OCTA_BF16 && dd == OCTA_BF16) cast_kernel<bf16_t, bf16_t><<<ew_blocks(n), 256, 0, st>>>((const bf16_t*)src, (bf16_t*)dst, n);
    else if (sd == OCTA_F32 && dd == OCTA_F16) cast_kernel<float, f16_t><<<ew_blocks(n), 256, 0, st>>>((const float*)src, (f16_t*)dst, n);
    else if (sd == OCTA_F16 && dd == OCTA_F32) cast_kernel<f16_t, float><<<ew_blocks(n), 256, 0, st>>>((const f16_t*)src, (float*)dst, n);
    else if (sd == OCTA_F16 && dd == OCTA_F16) cast_kernel<f16_t, f16_t><<<ew_blocks(n), 256, 0, st>>>((const f16_t*)src, (f16_t*)dst, n);
    else OCTA_FAIL(OCTA_ERR_BAD_ARG, "octa_cast: bad dtypes");
    OCTA_CHECK_LAUNCH("cast");
    return OCTA_OK;
}

// ------------------------------------------------------------------------------------------ Adam (torch.optim.Adam semantics)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   int64_t n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2s, float gs,
                                                   const int* __restrict__ step_dev, const float* __restrict__ ls_state, int ls_flag) {
    if (ls_state) {                               // loss scaling: the scale lives on the device, a flagged step is skipped
        if (ls_state[ls_flag] != 0.f) return;
        gs /= ls_state[0];
    }
    if (step_dev) {                               // t = updates applied so far + 1, kept on the device: a captured graph advances by itself
        const double t = (double)(*step_dev + 1); // and a skipped step does not count (octa_step_end commits the counter)
        bc1 = (float)(1.0 - pow((double)b1, t));
        bc2s = (float)sqrt(1.0 - pow((double)b2, t));
    }
    // 16-byte accesses when the four arrays allow it (the arenas do): a quarter of the memory instructions of the scalar loop
    int64_t done = 0;
    if (((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0) {
        const int64_t n4 = n >> 2;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
            const float4 g4 = ((const float4*)g)[i], p4 = ((const float4*)p)[i], m4 = ((const float4*)m)[i], v4 = ((const float4*)v)[i];
            const float ga[4] = {g4.x, g4.y, g4.z, g4.w}, pa[4] = {p4.x, p4.y, p4.z, p4.w}, ma[4] = {m4.x, m4.y, m4.z, m4.w}, va[4] = {v4.x, v4.y, v4.z, v4.w};
            float po[4], mo[4], vo[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float gg = ga[e] * gs;
                if (wd != 0.f) gg += wd * pa[e];
                mo[e] = b1 * ma[e] + (1.f - b1) * gg;
                vo[e] = b2 * va[e] + (1.f - b2) * gg * gg;
                const float denom = sqrtf(vo[e]) / bc2s + eps;
                po[e] = pa[e] - (lr / bc1) * (mo[e] / denom);
            }
            ((float4*)m)[i] = make_float4(mo[0], mo[1], mo[2], mo[3]);
            ((float4*)v)[i] = make_float4(vo[0], vo[1], vo[2], vo[3]);
            ((float4*)p)[i] = make_float4(po[0], po[1], po[2], po[3]);
        }
        done = n4 << 2;
    }
    for (int64_t i = done + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float gg = g[i] * gs;
        const float pp = p[i];
        if (wd != 0.f) gg += wd * pp;
        const float mm = b1 * m[i] + (1.f - b1) * gg;
        const float vv = b2 * v[i] + (1.f - b2) * gg * gg;
        m[i] = mm; v[i] = vv;
        const float denom = sqrtf(vv) / bc2s + eps;
        p[i] = pp - (lr / bc1) * (mm / denom);
    }
}
extern "C" int octa_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                              float weight_decay, int step, float grad_scale, const int* step_dev, const float* ls_state, int ls_flag,
                              octa_stream_t stream) {
    OCTA_REQUIRE(p && g && m && v && n > 0 && (step >= 1 || step_dev), "octa_adam_step: bad arguments");
    OCTA_REQUIRE(!ls_state || (ls_flag >= 2 && ls_flag < 8), "octa_adam_step: ls_flag indexes the found-inf slots 2..7 of the loss-scale state");
    if (step < 1) step = 1;
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    adam_kernel<<<ew_blocks((n + 3) / 4), 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale, step_dev, ls_state, ls_flag);
    OCTA_CHECK_LAUNCH("adam");
    return OCTA_OK;
}

// ------------------------------------------------------------------------------------------ dynamic loss scaling (fp16)
// state[0] = loss scale, state[1] = clean steps since the last change, state[2..7] = "a non-finite gradient was found" flags
// (one per optimiser).  Everything stays on the device, so the scheme is hipGraph-capturable and needs no host sync.
__global__ __launch_bounds__(256) void nonfinite_flag_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ flag) {
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const unsigned u = __float_as_uint(g[i]) & 0x7fffffffu;
        bad |= u >= 0x7f800000u;                     // inf or nan
    }
    if (bad) *flag = 1.f;                            // benign race: every writer stores the same value
}
extern "C" int octa_nonfinite_flag(const float* g, int64_t n, float* flag, octa_stream_t stream) {
    OCTA_REQUIRE(g && flag && n > 0, "octa_nonfinite_flag: bad arguments");
    int64_t b = cdiv64(n, 256 * 8);
    nonfinite_flag_kernel<<<(int)(b > 4096 ? 4096 : b), 256, 0, (hipStream_t)stream>>>(g, n, flag);
    OCTA_CHECK_LAUNCH("nonfinite_flag");
    return OCTA_OK;
}
__global__ void loss_scale_update_kernel(float* state, int nflags, float growth, float backoff, int interval) {
    bool found = false;
    for (int k = 0; k < nflags; ++k) { found |= state[2 + k] != 0.f; state[2 + k] = 0.f; }
    if (found) { state[0] = fmaxf(state[0] * backoff, 1.f); state[1] = 0.f; }
    else {
        const float t = state[1] + 1.f;
        if (t >= (float)interval) { state[0] = fminf(state[0] * growth, 16777216.f); state[1] = 0.f; }
        else state[1] = t;
    }
}
extern "C" int octa_loss_scale_update(float* state, int nflags, float growth, float backoff, int interval, octa_stream_t stream) {
    OCTA_REQUIRE(state && nflags >= 1 && nflags <= 6 && growth >= 1.f && backoff > 0.f && backoff <= 1.f && interval >= 1, "octa_loss_scale_update: bad arguments");
    loss_scale_update_kernel<<<1, 1, 0, (hipStream_t)stream>>>(state, nflags, growth, backoff, interval);
    OCTA_CHECK_LAUNCH("loss_scale_update");
    return OCTA_OK;
}

// End of an optimiser step in ONE single-thread launch (it replaces loss_scale_update + host_tick and keeps Adam's step
// counters on the device): (1) the applied-update counter of optimiser k advances unless its found-inf flag state[2+k] is
// set; (2) the loss-scale update above; (3) the step tick (extras.hip: host_tick_kernel).
__global__ void step_end_kernel(float* state, int nflags, float growth, float backoff, int interval, int* c0, int* c1, int* tick_dev, int* tick_host) {
    int* const cs[2] = {c0, c1};
    for (int k = 0; k < 2; ++k)
        if (cs[k] && !(state && k < nflags && state[2 + k] != 0.f)) *cs[k] += 1;
    if (state) {
        bool found = false;
        for (int k = 0; k < nflags; ++k) { found |= state[2 + k] != 0.f; state[2 + k] = 0.f; }
        if (growth == 1.f && backoff == 1.f) { state[1] = 0.f; }        // static scale: the value the caller set never moves (no clamps either)
        else if (found) { state[0] = fmaxf(state[0] * backoff, 1.f); state[1] = 0.f; }
        else {
            const float t = state[1] + 1.f;
            if (t >= (float)interval) { state[0] = fminf(state[0] * growth, 16777216.f); state[1] = 0.f; }
            else state[1] = t;
        }
    }
    if (tick_dev) {
        const int v = *tick_dev + 1;
        *tick_dev = v;
        if (tick_host) __hip_atomic_store(tick_host, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
extern "C" int octa_step_end(float* ls_state, int nflags, float growth, float backoff, int interval, int* step_dev0, int* step_dev1,
                             int* tick_dev, int* tick_host, octa_stream_t stream) {
    OCTA_REQUIRE(!ls_state || (nflags >= 1 && nflags <= 6 && growth >= 1.f && backoff > 0.f && backoff <= 1.f && interval >= 1), "octa_step_end: bad loss-scale arguments");
    OCTA_REQUIRE(ls_state || step_dev0 || step_dev1 || tick_dev, "octa_step_end: nothing to do");
    step_end_kernel<<<1, 1, 0, (hipStream_t)stream>>>(ls_state, nflags, growth, backoff, interval, step_dev0, step_dev1, tick_dev, tick_host);
    OCTA_CHECK_LAUNCH("step_end");
    return OCTA_OK;
}
