// Discriminator-specific kernels: instance noise + clip, spectral normalisation, full-extent conv.
#include "common.hpp"

struct Strides4 { int64_t b, c, h, w; };

// ------------------------------------------------------------------------------------------ InstanceNoise + clip
template <typename T>
__global__ __launch_bounds__(256) void noise_clip_fwd_kernel(const float* __restrict__ src, Strides4 s, const float* __restrict__ noise,
                                                             T* __restrict__ dst, uint8_t* __restrict__ mask, int B, int C, int H, int W, int ld,
                                                             int cpad, int clip) {
    const int64_t total = (int64_t)B * H * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int w = (int)(i % W);
        const int h = (int)((i / W) % H);
        const int b = (int)(i / ((int64_t)W * H));
        const float nz = noise ? noise[h * W + w] : 0.f;
        if constexpr (sizeof(T) == 2) {
            if (cpad == 8 && (ld & 7) == 0) {
                // the usual shape (2 class maps padded to one 16-byte channel vector): ONE store per pixel instead of eight 2-byte ones
                // (61 MB per call at 400 x 400: 34 -> 12 us)
                float v8[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    float v = 0.f;
                    if (c < C) {
                        v = src[b * s.b + c * s.c + h * s.h + w * s.w] + nz;
                        uint8_t m = 1;
                        if (clip) { m = (v >= 0.f && v <= 1.f) ? 1 : 0; v = fminf(fmaxf(v, 0.f), 1.f); }
                        if (mask) mask[(((int64_t)b * C + c) * H + h) * W + w] = m;
                    }
                    v8[c] = v;
                }
                *(uint4*)(dst + i * ld) = pack16<T>(v8);
                continue;
            }
        }
        for (int c = 0; c < cpad; ++c) {
            float v = 0.f;
            if (c < C) {
                v = src[b * s.b + c * s.c + h * s.h + w * s.w] + nz;
                uint8_t m = 1;
                if (clip) {   // torch.clip backward passes the gradient where min <= x <= max
                    m = (v >= 0.f && v <= 1.f) ? 1 : 0;
                    v = fminf(fmaxf(v, 0.f), 1.f);
                }
                if (mask) mask[(((int64_t)b * C + c) * H + h) * W + w] = m;
            }
            DT<T>::st(dst + i * ld + c, v);
        }
    }
}
extern "C" int octa_noise_clip_fwd(const float* src, const int64_t* ss, const float* noise, void* dst, uint8_t* mask, int B, int C, int H, int W,
                                   int ld, int cpad, int dtype, int clip, octa_stream_t stream) {
    OCTA_REQUIRE(src && ss && dst && cpad >= C && cpad <= ld, "octa_noise_clip_fwd: bad arguments");
    Strides4 s{ss[0], ss[1], ss[2], ss[3]};
    const int64_t total = (int64_t)B * H * W;
    const int blocks = (int)(cdiv64(total, 256) > 4096 ? 4096 : cdiv64(total, 256));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == OCTA_F32) noise_clip_fwd_kernel<float><<<blocks, 256, 0, st>>>(src, s, noise, (float*)dst, mask, B, C, H, W, ld, cpad, clip);
    else if (dtype == OCTA_BF16) noise_clip_fwd_kernel<bf16_t><<<blocks, 256, 0, st>>>(src, s, noise, (bf16_t*)dst, mask, B, C, H, W, ld, cpad, clip);
    else if (dtype == OCTA_F16) noise_clip_fwd_kernel<f16_t><<<blocks, 256, 0, st>>>(src, s, noise, (f16_t*)dst, mask, B, C, H, W, ld, cpad, clip);
    else OCTA_FAIL(OCTA_ERR_BAD_ARG, "octa_noise_clip_fwd: bad dtype");
    OCTA_CHECK_LAUNCH("noise_clip_fwd");
    return OCTA_OK;
}
// dsrc (NCHW dense fp32) = mask * ddst (NHWC)
template <typename T>
__global__ __launch_bounds__(256) void noise_clip_bwd_kernel(const T* __restrict__ ddst, int ld, const uint8_t* __restrict__ mask,
                                                             float* __restrict__ dsrc, int B, int C, int H, int W) {
    const int64_t total = (int64_t)B * C * H * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int w = (int)(i % W);
        const int h = (int)((i / W) % H);
        const int c = (int)((i / ((int64_t)W * H)) % C);
        const int b = (int)(i / ((int64_t)W * H * C));
        const float d = DT<T>::ld(ddst + (((int64_t)b * H + h) * W + w) * ld + c);
        dsrc[i] = (!mask || mask[i]) ? d : 0.f;
    }
}
extern "C" int octa_noise_clip_bwd(const void* ddst, int ld, const uint8_t* mask, float* dsrc, int B, int C, int H, int W, int dtype,
                                   octa_stream_t stream) {
    OCTA_REQUIRE(ddst && dsrc, "octa_noise_clip_bwd: bad arguments");
    const int64_t total = (int64_t)B * C * H * W;
    const int blocks = (int)(cdiv64(total, 256) > 4096 ? 4096 : cdiv64(total, 256));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == OCTA_F32) noise_clip_bwd_kernel<float><<<blocks, 256, 0, st>>>((const float*)ddst, ld, mask, dsrc, B, C, H, W);
    else if (dtype == OCTA_BF16) noise_clip_bwd_kernel<bf16_t><<<blocks, 256, 0, st>>>((const bf16_t*)ddst, ld, mask, dsrc, B, C, H, W);
    else if (dtype == OCTA_F16) noise_clip_bwd_kernel<f16_t><<<blocks, 256, 0, st>>>((const f16_t*)ddst, ld, mask, dsrc, B, C, H, W);
    else OCTA_FAIL(OCTA_ERR_BAD_ARG, "octa_noise_clip_bwd: bad dtype");
    OCTA_CHECK_LAUNCH("noise_clip_bwd");
    return OCTA_OK;
}

// ---- InstanceNoise + clip, written SPACE-TO-DEPTH for the k4 s2 p1 conv that follows (round 5): that conv on [H][W][C] is a k2 s1 p0
// conv on dst[b][Y][X][(dy*2+dx)*C + c] = v[b][c][2Y+dy-1][2X+dx-1] (0 outside the image), Y <= H/2, X <= W/2 -- for C = 2 the eight
// channels of a pixel are all real (the plain layout pads 2 channels to 8: 4x the bytes and 4x the MFMA K of the first discriminator conv)
template <typename T>
__global__ __launch_bounds__(256) void noise_clip_s2d_fwd_kernel(const float* __restrict__ src, Strides4 s, const float* __restrict__ noise,
                                                                 T* __restrict__ dst, uint8_t* __restrict__ mask, int B, int C, int H, int W, int ld, int clip) {
    const int H2 = H / 2 + 1, W2 = W / 2 + 1;
    const int64_t total = (int64_t)B * H2 * W2 * 4;            // thread = (pixel of the s2d image, quadrant)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int qd = (int)(i & 3);
        int64_t p = i >> 2;
        const int X = (int)(p % W2); p /= W2;
        const int Y = (int)(p % H2);
        const int b = (int)(p / H2);
        const int h = 2 * Y + (qd >> 1) - 1, w = 2 * X + (qd & 1) - 1;
        const bool in = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
        const float nz = (noise && in) ? noise[h * W + w] : 0.f;
        T* o = dst + (((int64_t)b * H2 + Y) * W2 + X) * ld + qd * C;
        for (int c = 0; c < C; ++c) {
            float v = 0.f;
            if (in) {
                v = src[b * s.b + c * s.c + h * s.h + w * s.w] + nz;
                uint8_t m = 1;
                if (clip) { m = (v >= 0.f && v <= 1.f) ? 1 : 0; v = fminf(fmaxf(v, 0.f), 1.f); }
                if (mask) mask[(((int64_t)b * C + c) * H + h) * W + w] = m;
            }
            DT<T>::st(o + c, v);
        }
        if (qd == 3) for (int c = 4 * C; c < ld; ++c) DT<T>::st(dst + (((int64_t)b * H2 + Y) * W2 + X) * ld + c, 0.f);
    }
}
extern "C" int octa_noise_clip_s2d_fwd(const float* src, const int64_t* ss, const float* noise, void* dst, uint8_t* mask, int B, int C, int H, int W,
                                       int ld, int dtype, int clip, octa_stream_t stream) {
    OCTA_REQUIRE(src && ss && dst && C > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && ld >= 4 * C, "octa_noise_clip_s2d_fwd: even H / W, ld >= 4 C");
    Strides4 s{ss[0], ss[1], ss[2], ss[3]};
    const int64_t total = (int64_t)B * (H / 2 + 1) * (W / 2 + 1) * 4;
    const int blocks = (int)(cdiv64(total, 256) > 8192 ? 8192 : cdiv64(total, 256));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == OCTA_F32) noise_clip_s2d_fwd_kernel<float><<<blocks, 256, 0, st>>>(src, s, noise, (float*)dst, mask, B, C, H, W, ld, clip);
    else if (dtype == OCTA_BF16) noise_clip_s2d_fwd_kernel<bf16_t><<<blocks, 256, 0, st>>>(src, s, noise, (bf16_t*)dst, mask, B, C, H, W, ld, clip);
    else if (dtype == OCTA_F16) noise_clip_s2d_fwd_kernel<f16_t><<<blocks, 256, 0, st>>>(src, s, noise, (f16_t*)dst, mask, B, C, H, W, ld, clip);
    else OCTA_FAIL(OCTA_ERR_BAD_ARG, "octa_noise_clip_s2d_fwd: bad dtype");
    OCTA_CHECK_LAUNCH("noise_clip_s2d_fwd");
    return OCTA_OK;
}
// dsrc[b,c,h,w] (dense NCHW fp32) = mask ? ddst[b][(h+1)/2][(w+1)/2][(((h+1)&1)*2 + ((w+1)&1))*C + c] : 0
template <typename T>
__global__ __launch_bounds__(256) void noise_clip_s2d_bwd_kernel(const T* __restrict__ ddst, int ld, const uint8_t* __restrict__ mask,
                                                                 float* __restrict__ dsrc, int B, int C, int H, int W) {
    const int H2 = H / 2 + 1, W2 = W / 2 + 1;
    const int64_t total = (int64_t)B * C * H * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int w = (int)(i % W);
        const int h = (int)((i / W) % H);
        const int c = (int)((i / ((int64_t)W * H)) % C);
        const int b = (int)(i / ((int64_t)W * H * C));
        const int Y = (h + 1) >> 1, X = (w + 1) >> 1, qd = (((h + 1) & 1) << 1) | ((w + 1) & 1);
        const float d = DT<T>::ld(ddst + (((int64_t)b * H2 + Y) * W2 + X) * ld + qd * C + c);
        dsrc[i] = (!mask || mask[i]) ? d : 0.f;
    }
}
extern "C" int octa_noise_clip_s2d_bwd(const void* ddst, int ld, const uint8_t* mask, float* dsrc, int B, int C, int H, int W, int dtype,
                                       octa_stream_t stream) {
    OCTA_REQUIRE(ddst && dsrc && H % 2 == 0 && W % 2 == 0 && ld >= 4 * C, "octa_noise_clip_s2d_bwd: bad arguments");
    const int64_t total = (int64_t)B * C * H * W;
    const int blocks = (int)(cdiv64(total, 256) > 4096 ? 4096 : cdiv64(total, 256));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == OCTA_F32) noise_clip_s2d_bwd_kernel<float><<<blocks, 256, 0, st>>>((const float*)ddst, ld, mask, dsrc, B, C, H, W);
    else if (dtype == OCTA_BF16) noise_clip_s2d_bwd_kernel<bf16_t><<<blocks, 256, 0, st>>>((const bf16_t*)ddst, ld, mask, dsrc, B, C, H, W);
    else if (dtype == OCTA_F16) noise_clip_s2d_bwd_kernel<f16_t><<<blocks, 256, 0, st>>>((const f16_t*)ddst, ld, mask, dsrc, B, C, H, W);
    else OCTA_FAIL(OCTA_ERR_BAD_ARG, "octa_noise_clip_s2d_bwd: bad dtype");
    OCTA_CHECK_LAUNCH("noise_clip_s2d_bwd");
    return OCTA_OK;
}

// ------------------------------------------------------------------------------------------ spectral norm (single block)
__device__ float block_total(float v, float* red) {   // all threads get the block-wide sum
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < (int)((blockDim.x + 63) >> 6); ++i) s += red[i];
    return s;
}
// Spectral norm, multi-block: the matrix (up to 1024 x 240 fp32 = 1 MB) is streamed by ~64 workgroups instead of one.
//   A: v_raw[j] += sum_{i in block rows} W[i][j] u[i]                      (atomics into a zeroed K-vector)
//   B: v = v_raw / max(|v_raw|, eps);  wv[i] = W[i] . v  for the block's rows; block 0 stores v
//   C: u = wv / max(|wv|, eps); sigma = u . wv;  w_sn = W / sigma; block 0 stores u and sigma
// Every block recomputes the tiny norms (K resp. Cout values) itself, so no grid-wide sync is needed.
#define SN_ROWS 16
__global__ __launch_bounds__(256) void spectral_A_kernel(const float* __restrict__ w, const float* __restrict__ u, float* __restrict__ vraw, int Cout, int K, int det) {
    // det (deterministic mode): workgroup 0 walks all the rows, so every vraw[j] receives ONE add
    if (det && blockIdx.x != 0) return;
    const int r0 = det ? 0 : blockIdx.x * SN_ROWS, r1 = det ? Cout : min(Cout, r0 + SN_ROWS);
    for (int j = threadIdx.x; j < K; j += 256) {
        float s = 0.f;
        for (int i = r0; i < r1; ++i) s += w[(int64_t)i * K + j] * u[i];
        atomicAdd(vraw + j, s);
    }
}
__device__ float block_total256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void spectral_B_kernel(const float* __restrict__ w, const float* __restrict__ vraw, float* __restrict__ v, float* __restrict__ wv,
                                                         int Cout, int K, int iter, float eps, float* __restrict__ vsave) {
    extern __shared__ float sm[];   // vs[K], red[4]
    float* vs = sm;
    float* red = sm + K;
    float nrm = 0.f;
    for (int j = threadIdx.x; j < K; j += 256) { const float x = iter ? vraw[j] : v[j]; vs[j] = x; nrm += x * x; }
    nrm = sqrtf(block_total256(nrm, red));
    if (iter) {
        const float d = fmaxf(nrm, eps);
        for (int j = threadIdx.x; j < K; j += 256) vs[j] /= d;
        __syncthreads();
        if (blockIdx.x == 0) for (int j = threadIdx.x; j < K; j += 256) v[j] = vs[j];
    }
    if (vsave && blockIdx.x == 0) {                  // the v this step used, kept for the backward pass
        __syncthreads();
        for (int j = threadIdx.x; j < K; j += 256) vsave[j] = vs[j];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * SN_ROWS, r1 = min(Cout, r0 + SN_ROWS);
    for (int i = r0 + wave; i < r1; i += 4) {
        float s = 0.f;
        for (int j = lane; j < K; j += 64) s += w[(int64_t)i * K + j] * vs[j];
        s = wave_sum(s);
        if (lane == 0) wv[i] = s;
    }
}
__global__ __launch_bounds__(256) void spectral_C_kernel(const float* __restrict__ w, const float* __restrict__ wv, float* __restrict__ u, float* __restrict__ sigma,
                                                         float* __restrict__ wsn, int Cout, int K, int iter, float eps, float* __restrict__ usave) {
    __shared__ float red[4];
    float a = 0.f, b = 0.f;     // |wv|^2 and u_old . wv
    for (int i = threadIdx.x; i < Cout; i += 256) { const float x = wv[i]; a += x * x; b += u[i] * x; }
    a = block_total256(a, red);
    b = block_total256(b, red);
    float sg;
    if (iter) {
        const float d = fmaxf(sqrtf(a), eps);
        sg = a / d;                                   // u_new . wv with u_new = wv / d
        if (blockIdx.x == 0) for (int i = threadIdx.x; i < Cout; i += 256) { const float un = wv[i] / d; u[i] = un; if (usave) usave[i] = un; }
    } else {
        sg = b;
        if (usave && blockIdx.x == 0) for (int i = threadIdx.x; i < Cout; i += 256) usave[i] = u[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) sigma[0] = sg;
    const float inv = 1.f / sg;
    const int64_t n = (int64_t)Cout * K;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) wsn[i] = w[i] * inv;
}
extern "C" int octa_spectral_norm_fwd(const float* w, float* u, float* v, int Cout, int K, int do_power_iter, float eps, float* sigma,
                                      float* w_sn, float* ws, float* uv_saved, int ws_prezeroed, octa_stream_t stream) {
    OCTA_REQUIRE(w && u && v && sigma && w_sn && ws && Cout > 0 && K > 0, "octa_spectral_norm_fwd: bad arguments (ws: K + Cout floats)");
    hipStream_t st = (hipStream_t)stream;
    float* vraw = ws;
    float* wv = ws + K;
    const int nb = cdiv(Cout, SN_ROWS);
    if (do_power_iter) {
        if (!ws_prezeroed && octa_zero_async(vraw, (size_t)K * sizeof(float), st) != hipSuccess) OCTA_FAIL(OCTA_ERR_LAUNCH, "octa_spectral_norm_fwd: memset failed");
        spectral_A_kernel<<<nb, 256, 0, st>>>(w, u, vraw, Cout, K, octa_deterministic() ? 1 : 0);
        OCTA_CHECK_LAUNCH("spectral_A");
    }
    spectral_B_kernel<<<nb, 256, (size_t)(K + 4) * sizeof(float), st>>>(w, vraw, v, wv, Cout, K, do_power_iter, eps, uv_saved ? uv_saved + Cout : nullptr);
    OCTA_CHECK_LAUNCH("spectral_B");
    int nc = (int)cdiv64((int64_t)Cout * K, 256 * 8);
    if (nc < 1) nc = 1;
    spectral_C_kernel<<<nc, 256, 0, st>>>(w, wv, u, sigma, w_sn, Cout, K, do_power_iter, eps, uv_saved);
    OCTA_CHECK_LAUNCH("spectral_C");
    return OCTA_OK;
}
// backward: dot = sum(dw_sn * w_sn) (atomics into ws[0]), then dw += (dw_sn - dot * u v^T) / sigma
// dw_sn may be channels-last ([Cout][KH][KW][Cin], what the weight-gradient kernels prefer): khw = KH*KW then, 0 for dense OIHW
__device__ __forceinline__ int64_t sn_src(int64_t i, int K, int khw) {
    if (!khw) return i;
    const int r = (int)(i / K), c = (int)(i % K), cin = K / khw;
    return (int64_t)r * K + (int64_t)(c % khw) * cin + c / khw;
}
__global__ __launch_bounds__(256) void spectral_bwd_dot_kernel(const float* __restrict__ dwsn, const float* __restrict__ wsn, int64_t n, float* __restrict__ dot,
                                                               int K, int khw) {
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += dwsn[sn_src(i, K, khw)] * wsn[i];
    s = block_total256(s, red);
    if (threadIdx.x == 0) atomicAdd(dot, s);
}
__global__ __launch_bounds__(256) void spectral_bwd_apply_kernel(const float* __restrict__ dwsn, const float* __restrict__ u, const float* __restrict__ v,
                                                                 const float* __restrict__ sigma, const float* __restrict__ dot, int K, int64_t n,
                                                                 float* __restrict__ dw, int accumulate, int khw) {
    const float inv = 1.f / sigma[0], s = dot[0];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / K), c = (int)(i % K);
        const float g = (dwsn[sn_src(i, K, khw)] - s * u[r] * v[c]) * inv;
        dw[i] = accumulate ? dw[i] + g : g;
    }
}
extern "C" int octa_spectral_norm_bwd(const float* dw_sn, const float* w_sn, const float* u, const float* v, const float* sigma, int Cout, int K,
                                      float* dw, float* ws, int accumulate, int ws_prezeroed, int dwsn_khw, octa_stream_t stream) {
    OCTA_REQUIRE(dw_sn && w_sn && u && v && sigma && dw && ws && dwsn_khw >= 0 && (dwsn_khw == 0 || K % dwsn_khw == 0), "octa_spectral_norm_bwd: bad arguments (ws: 1 float; dwsn_khw divides K)");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)Cout * K;
    int nb = (int)cdiv64(n, 256 * 8);
    if (nb < 1 || octa_deterministic()) nb = 1;            // (deterministic mode: one workgroup, one add into the dot product)
    if (!ws_prezeroed && octa_zero_async(ws, sizeof(float), st) != hipSuccess) OCTA_FAIL(OCTA_ERR_LAUNCH, "octa_spectral_norm_bwd: memset failed");
    spectral_bwd_dot_kernel<<<nb, 256, 0, st>>>(dw_sn, w_sn, n, ws, K, dwsn_khw);
    OCTA_CHECK_LAUNCH("spectral_bwd_dot");
    spectral_bwd_apply_kernel<<<nb, 256, 0, st>>>(dw_sn, u, v, sigma, ws, K, n, dw, accumulate, dwsn_khw);
    OCTA_CHECK_LAUNCH("spectral_bwd_apply");
    return OCTA_OK;
}

// ------------------------------------------------------------------------------------------ spectral norm, several layers per launch
// A discriminator call normalises four conv weights (blocks.py:97-110), each three launches of a few microseconds in a chain
// that is pure launch latency.  The layers are independent: one launch per kernel covers all of them (blockIdx.x -> (layer,
// block of the layer) through a prefix table passed by value).
#define SN_MAXJOBS 8
struct SnBatch {
    int n;
    int first[SN_MAXJOBS + 1];                // first block of job j (A / B: row blocks, C: element blocks, bwd: element blocks)
    const float* w[SN_MAXJOBS]; float* u[SN_MAXJOBS]; float* v[SN_MAXJOBS]; float* sigma[SN_MAXJOBS]; float* wsn[SN_MAXJOBS];
    float* ws[SN_MAXJOBS]; float* uvs[SN_MAXJOBS];
    int Cout[SN_MAXJOBS], K[SN_MAXJOBS];
    // optional packed conv operands of w_sn, written by the C kernel beside w_sn (saves the per-call pack launches of a normalised weight):
    void* pk_fwd[SN_MAXJOBS]; void* pk_dgt[SN_MAXJOBS];
    int pk_khw[SN_MAXJOBS], pk_kw[SN_MAXJOBS], pk_cin[SN_MAXJOBS], pk_cinp[SN_MAXJOBS], pk_coutp[SN_MAXJOBS], pk_dtype[SN_MAXJOBS];
};
template <typename T> __device__ __forceinline__ void sn_pack(const SnBatch& b, int j, const float* __restrict__ w, float inv, int lb, int nblk) {
    const int Cout = b.Cout[j], K = b.K[j], khw = b.pk_khw[j], KW = b.pk_kw[j], Cin = b.pk_cin[j], cinp = b.pk_cinp[j], coutp = b.pk_coutp[j];
    if (b.pk_fwd[j]) {            // forward operand [Cout][KH][KW][cin_pad]
        T* __restrict__ out = (T*)b.pk_fwd[j];
        const int64_t tot = (int64_t)Cout * khw * cinp;
        for (int64_t i = (int64_t)lb * 256 + threadIdx.x; i < tot; i += (int64_t)nblk * 256) {
            const int ci = (int)(i % cinp);
            const int64_t t = i / cinp;
            const int tap = (int)(t % khw), co = (int)(t / khw);
            float v = ci < Cin ? w[(int64_t)co * K + ci * khw + tap] * inv : 0.f;
            asm volatile("" : "+v"(v));     // the fp32 product, rounded as w_sn stores it, THEN converted (no fused mul + f16 convert)
            DT<T>::st(out + i, v);
        }
    }
    if (b.pk_dgt[j]) {            // tap-major data-gradient operand [KH*KW][cin_pad][cout_pad]
        T* __restrict__ out = (T*)b.pk_dgt[j];
        const int64_t tot = (int64_t)khw * cinp * coutp;
        for (int64_t i = (int64_t)lb * 256 + threadIdx.x; i < tot; i += (int64_t)nblk * 256) {
            const int co = (int)(i % coutp);
            const int64_t t = i / coutp;
            const int ci = (int)(t % cinp), tap = (int)(t / cinp);
            float v = (ci < Cin && co < Cout) ? w[(int64_t)co * K + ci * khw + tap] * inv : 0.f;
            asm volatile("" : "+v"(v));
            DT<T>::st(out + i, v);
        }
    }
    (void)KW;
}
__device__ __forceinline__ int sn_job_of(const SnBatch& b, int blk, int& local) {
    int j = 0;
#pragma unroll
    for (int i = 1; i < SN_MAXJOBS; ++i) if (i < b.n && blk >= b.first[i]) j = i;
    local = blk - b.first[j];
    return j;
}
__global__ __launch_bounds__(256) void spectral_A_batch_kernel(const SnBatch b, int det) {
    int lb;
    const int j = sn_job_of(b, blockIdx.x, lb);
    if (det && lb != 0) return;                // deterministic mode: the job's first workgroup walks all its rows
    const float* __restrict__ w = b.w[j];
    const float* __restrict__ u = b.u[j];
    float* __restrict__ vraw = b.ws[j];
    const int Cout = b.Cout[j], K = b.K[j];
    const int r0 = det ? 0 : lb * SN_ROWS, r1 = det ? Cout : min(Cout, r0 + SN_ROWS);
    for (int c = threadIdx.x; c < K; c += 256) {
        float s = 0.f;
        for (int i = r0; i < r1; ++i) s += w[(int64_t)i * K + c] * u[i];
        atomicAdd(vraw + c, s);
    }
}
__global__ __launch_bounds__(256) void spectral_B_batch_kernel(const SnBatch b, int iter, float eps) {
    extern __shared__ float sm[];   // vs[Kmax], red[4]
    int lb;
    const int j = sn_job_of(b, blockIdx.x, lb);
    const float* __restrict__ w = b.w[j];
    const int Cout = b.Cout[j], K = b.K[j];
    float* __restrict__ v = b.v[j];
    const float* __restrict__ vraw = b.ws[j];
    float* __restrict__ wv = b.ws[j] + K;
    float* vsave = b.uvs[j] ? b.uvs[j] + Cout : nullptr;
    float* vs = sm;
    float* red = sm + K;
    float nrm = 0.f;
    for (int c = threadIdx.x; c < K; c += 256) { const float x = iter ? vraw[c] : v[c]; vs[c] = x; nrm += x * x; }
    nrm = sqrtf(block_total256(nrm, red));
    if (iter) {
        const float d = fmaxf(nrm, eps);
        for (int c = threadIdx.x; c < K; c += 256) vs[c] /= d;
        __syncthreads();
        if (lb == 0) for (int c = threadIdx.x; c < K; c += 256) v[c] = vs[c];
    }
    if (vsave && lb == 0) {
        __syncthreads();
        for (int c = threadIdx.x; c < K; c += 256) vsave[c] = vs[c];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = lb * SN_ROWS, r1 = min(Cout, r0 + SN_ROWS);
    for (int i = r0 + wave; i < r1; i += 4) {
        float s = 0.f;
        for (int c = lane; c < K; c += 64) s += w[(int64_t)i * K + c] * vs[c];
        s = wave_sum(s);
        if (lane == 0) wv[i] = s;
    }
}
__global__ __launch_bounds__(256) void spectral_C_batch_kernel(const SnBatch b, int iter, float eps) {
    __shared__ float red[4];
    int lb;
    const int j = sn_job_of(b, blockIdx.x, lb);
    const int nblk = b.first[j + 1] - b.first[j];
    const float* __restrict__ w = b.w[j];
    const int Cout = b.Cout[j], K = b.K[j];
    float* __restrict__ u = b.u[j];
    const float* __restrict__ wv = b.ws[j] + K;
    float* __restrict__ wsn = b.wsn[j];
    float* usave = b.uvs[j];
    float a = 0.f, bb = 0.f;     // |wv|^2 and u_old . wv
    for (int i = threadIdx.x; i < Cout; i += 256) { const float x = wv[i]; a += x * x; bb += u[i] * x; }
    a = block_total256(a, red);
    bb = block_total256(bb, red);
    float sg;
    if (iter) {
        const float d = fmaxf(sqrtf(a), eps);
        sg = a / d;
        // every block has read the old u above (and needs it for nothing else): block 0 may overwrite it only after the
        // OTHER blocks of this layer have read it too -- they do not read it when iter != 0 beyond the sum above, whose value
        // (bb) is unused in this branch, so the order does not matter
        if (lb == 0) for (int i = threadIdx.x; i < Cout; i += 256) { const float un = wv[i] / d; u[i] = un; if (usave) usave[i] = un; }
    } else {
        sg = bb;
        if (usave && lb == 0) for (int i = threadIdx.x; i < Cout; i += 256) usave[i] = u[i];
    }
    if (lb == 0 && threadIdx.x == 0) b.sigma[j][0] = sg;
    const float inv = 1.f / sg;
    const int64_t n = (int64_t)Cout * K;
    for (int64_t i = (int64_t)lb * 256 + threadIdx.x; i < n; i += (int64_t)nblk * 256) wsn[i] = w[i] * inv;
    if (b.pk_fwd[j] || b.pk_dgt[j]) {
        if (b.pk_dtype[j] == OCTA_F32) sn_pack<float>(b, j, w, inv, lb, nblk);
        else if (b.pk_dtype[j] == OCTA_BF16) sn_pack<bf16_t>(b, j, w, inv, lb, nblk);
        else sn_pack<f16_t>(b, j, w, inv, lb, nblk);
    }
}
extern "C" int octa_spectral_norm_fwd_batch(const octa_sn_job* jobs, int n, int do_power_iter, float eps, int ws_prezeroed, octa_stream_t stream) {
    OCTA_REQUIRE(jobs && n >= 1 && n <= SN_MAXJOBS, "octa_spectral_norm_fwd_batch: 1..%d jobs", SN_MAXJOBS);
    hipStream_t st = (hipStream_t)stream;
    SnBatch rb, cb;
    rb.n = cb.n = n;
    int rtot = 0, ctot = 0, kmax = 0;
    for (int j = 0; j < n; ++j) {
        const octa_sn_job& q = jobs[j];
        OCTA_REQUIRE(q.w && q.u && q.v && q.sigma && q.w_sn && q.ws && q.Cout > 0 && q.K > 0, "octa_spectral_norm_fwd_batch: job %d: bad arguments (ws: K + Cout floats)", j);
        rb.first[j] = rtot; cb.first[j] = ctot;
        rtot += cdiv(q.Cout, SN_ROWS);
        int nc = (int)cdiv64((int64_t)q.Cout * q.K, 256 * 8);
        ctot += nc < 1 ? 1 : nc;
        rb.w[j] = cb.w[j] = q.w; rb.u[j] = cb.u[j] = q.u; rb.v[j] = cb.v[j] = q.v; rb.sigma[j] = cb.sigma[j] = q.sigma;
        rb.wsn[j] = cb.wsn[j] = q.w_sn; rb.ws[j] = cb.ws[j] = q.ws; rb.uvs[j] = cb.uvs[j] = q.uv_saved;
        rb.Cout[j] = cb.Cout[j] = q.Cout; rb.K[j] = cb.K[j] = q.K;
        cb.pk_fwd[j] = q.packed_fwd; cb.pk_dgt[j] = q.packed_dgrad_taps;
        cb.pk_khw[j] = q.KH * q.KW; cb.pk_kw[j] = q.KW; cb.pk_cin[j] = q.Cin; cb.pk_cinp[j] = (q.Cin + 7) / 8 * 8; cb.pk_coutp[j] = (q.Cout + 7) / 8 * 8;
        cb.pk_dtype[j] = q.pack_dtype;
        if (q.packed_fwd || q.packed_dgrad_taps)
            OCTA_REQUIRE(q.KH > 0 && q.KW > 0 && q.Cin > 0 && q.Cin * q.KH * q.KW == q.K && OCTA_DTYPE_OK(q.pack_dtype),
                         "octa_spectral_norm_fwd_batch: job %d: packed operands need KH, KW, Cin with Cin*KH*KW == K and a dtype", j);
        if (q.K > kmax) kmax = q.K;
        if (do_power_iter && !ws_prezeroed && octa_zero_async(q.ws, (size_t)q.K * sizeof(float), st) != hipSuccess)
            OCTA_FAIL(OCTA_ERR_LAUNCH, "octa_spectral_norm_fwd_batch: memset failed");
    }
    rb.first[n] = rtot; cb.first[n] = ctot;
    if (do_power_iter) {
        spectral_A_batch_kernel<<<rtot, 256, 0, st>>>(rb, octa_deterministic() ? 1 : 0);
        OCTA_CHECK_LAUNCH("spectral_A(batch)");
    }
    spectral_B_batch_kernel<<<rtot, 256, (size_t)(kmax + 4) * sizeof(float), st>>>(rb, do_power_iter, eps);
    OCTA_CHECK_LAUNCH("spectral_B(batch)");
    spectral_C_batch_kernel<<<ctot, 256, 0, st>>>(cb, do_power_iter, eps);
    OCTA_CHECK_LAUNCH("spectral_C(batch)");
    return OCTA_OK;
}
struct SnBwdBatch {
    int n;
    int first[SN_MAXJOBS + 1];
    const float* dwsn[SN_MAXJOBS]; const float* wsn[SN_MAXJOBS]; const float* u[SN_MAXJOBS]; const float* v[SN_MAXJOBS];
    const float* sigma[SN_MAXJOBS]; float* dw[SN_MAXJOBS]; float* dot[SN_MAXJOBS];
    int K[SN_MAXJOBS], khw[SN_MAXJOBS], acc[SN_MAXJOBS];
    int64_t nel[SN_MAXJOBS];
};
template <int PASS>
__global__ __launch_bounds__(256) void spectral_bwd_batch_kernel(const SnBwdBatch b) {
    __shared__ float red[4];
    int lb = blockIdx.x, j = 0;
#pragma unroll
    for (int i = 1; i < SN_MAXJOBS; ++i) if (i < b.n && (int)blockIdx.x >= b.first[i]) j = i;
    lb -= b.first[j];
    const int nblk = b.first[j + 1] - b.first[j];
    const float* __restrict__ dwsn = b.dwsn[j];
    const int K = b.K[j], khw = b.khw[j];
    const int64_t n = b.nel[j];
    if (PASS == 0) {
        const float* __restrict__ wsn = b.wsn[j];
        float s = 0.f;
        for (int64_t i = (int64_t)lb * 256 + threadIdx.x; i < n; i += (int64_t)nblk * 256) s += dwsn[sn_src(i, K, khw)] * wsn[i];
        s = block_total256(s, red);
        if (threadIdx.x == 0) atomicAdd(b.dot[j], s);
    } else {
        const float* __restrict__ u = b.u[j];
        const float* __restrict__ v = b.v[j];
        float* __restrict__ dw = b.dw[j];
        const float inv = 1.f / b.sigma[j][0], s = b.dot[j][0];
        const int accumulate = b.acc[j];
        for (int64_t i = (int64_t)lb * 256 + threadIdx.x; i < n; i += (int64_t)nblk * 256) {
            const int r = (int)(i / K), c = (int)(i % K);
            const float g = (dwsn[sn_src(i, K, khw)] - s * u[r] * v[c]) * inv;
            dw[i] = accumulate ? dw[i] + g : g;
        }
    }
}
extern "C" int octa_spectral_norm_bwd_batch(const octa_sn_bwd_job* jobs, int n, int ws_prezeroed, octa_stream_t stream) {
    OCTA_REQUIRE(jobs && n >= 1 && n <= SN_MAXJOBS, "octa_spectral_norm_bwd_batch: 1..%d jobs", SN_MAXJOBS);
    hipStream_t st = (hipStream_t)stream;
    SnBwdBatch b;
    b.n = n;
    int tot = 0;
    for (int j = 0; j < n; ++j) {
        const octa_sn_bwd_job& q = jobs[j];
        OCTA_REQUIRE(q.dw_sn && q.w_sn && q.u && q.v && q.sigma && q.dw && q.ws && q.Cout > 0 && q.K > 0 && q.dwsn_khw >= 0 &&
                         (q.dwsn_khw == 0 || q.K % q.dwsn_khw == 0),
                     "octa_spectral_norm_bwd_batch: job %d: bad arguments (ws: 1 float; dwsn_khw divides K)", j);
        b.first[j] = tot;
        const int64_t nel = (int64_t)q.Cout * q.K;
        int nb = (int)cdiv64(nel, 256 * 8);
        if (octa_deterministic()) nb = 1;
        tot += nb < 1 ? 1 : nb;
        b.dwsn[j] = q.dw_sn; b.wsn[j] = q.w_sn; b.u[j] = q.u; b.v[j] = q.v; b.sigma[j] = q.sigma; b.dw[j] = q.dw; b.dot[j] = q.ws;
        b.K[j] = q.K; b.khw[j] = q.dwsn_khw; b.acc[j] = q.accumulate; b.nel[j] = nel;
        if (!ws_prezeroed && octa_zero_async(q.ws, sizeof(float), st) != hipSuccess) OCTA_FAIL(OCTA_ERR_LAUNCH, "octa_spectral_norm_bwd_batch: memset failed");
    }
    b.first[n] = tot;
    spectral_bwd_batch_kernel<0><<<tot, 256, 0, st>>>(b);
    OCTA_CHECK_LAUNCH("spectral_bwd_dot(batch)");
    spectral_bwd_batch_kernel<1><<<tot, 256, 0, st>>>(b);
    OCTA_CHECK_LAUNCH("spectral_bwd_apply(batch)");
    return OCTA_OK;
}

// ------------------------------------------------------------------------------------------ full-extent conv = dot product
template <typename T>
__global__ __launch_bounds__(256) void fullconv_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, float* __restrict__ out, int64_t n,
                                                           float sign, const float* __restrict__ sign_dev, int nblk, const float* __restrict__ bias) {
    __shared__ float red[16];
    if (sign_dev) sign *= sign_dev[0];
    const int b = blockIdx.y;
    float acc[1] = {0.f};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)nblk * 256) acc[0] += DT<T>::ld(x + b * n + i) * w[i];
    block_sum<1>(acc, red);
    if (threadIdx.x == 0) atomicAdd(out + b, sign * (acc[0] + ((bias && blockIdx.x == 0) ? bias[0] : 0.f)));
}
__global__ void fullconv_init_kernel(const float* __restrict__ bias, float* __restrict__ out, int B, float sign, const float* __restrict__ sign_dev) {
    if (sign_dev) sign *= sign_dev[0];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) out[i] = sign * (bias ? bias[0] : 0.f);
}
extern "C" int octa_fullconv_fwd(const void* x, const float* w, const float* bias, float* out, int B, int64_t n, int dtype, float sign,
                                 const float* sign_dev, int out_prezeroed, octa_stream_t stream) {
    OCTA_REQUIRE(x && w && out && B > 0 && n > 0, "octa_fullconv_fwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (!out_prezeroed) {
        fullconv_init_kernel<<<cdiv(B, 64), 64, 0, st>>>(bias, out, B, sign, sign_dev);
        OCTA_CHECK_LAUNCH("fullconv_init");
    }
    const float* badd = out_prezeroed ? bias : nullptr;      // zeroed `out`: the first workgroup of every sample adds the bias itself
    int nblk = (int)(cdiv64(n, 256 * 8) > 64 ? 64 : cdiv64(n, 256 * 8));
    if (nblk < 1 || octa_deterministic()) nblk = 1;
    dim3 grid(nblk, B);
    if (dtype == OCTA_F32) fullconv_fwd_kernel<float><<<grid, 256, 0, st>>>((const float*)x, w, out, n, sign, sign_dev, nblk, badd);
    else if (dtype == OCTA_BF16) fullconv_fwd_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)x, w, out, n, sign, sign_dev, nblk, badd);
    else if (dtype == OCTA_F16) fullconv_fwd_kernel<f16_t><<<grid, 256, 0, st>>>((const f16_t*)x, w, out, n, sign, sign_dev, nblk, badd);
    else OCTA_FAIL(OCTA_ERR_BAD_ARG, "octa_fullconv_fwd: bad dtype");
    OCTA_CHECK_LAUNCH("fullconv_fwd");
    return OCTA_OK;
}
// dx[b,i] = sign*dout[b]*w[i];  dw[i] += sign * sum_b dout[b]*x[b,i];  dbias += sign * sum_b dout[b]
template <typename T>
__global__ __launch_bounds__(256) void fullconv_bwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ dout,
                                                           T* __restrict__ dx, float* __restrict__ dw, float* __restrict__ dbias, int B, int64_t n,
                                                           float sign, const float* __restrict__ sign_dev, int dw_c, int gate_act) {
    if (sign_dev) sign *= sign_dev[0];
    const int64_t hw = dw_c > 0 ? n / dw_c : 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float wi = w[i];
        float g = 0.f;
        for (int b = 0; b < B; ++b) {
            const float d = sign * dout[b];
            const float xv = DT<T>::ld(x + b * n + i);
            // gate_act: x is the output of that activation and dx the gradient that reaches it -- times f'(x) here (octa_fullconv_bwd_gated)
            float gt = 1.f;
            if (gate_act == OCTA_ACT_TANH) gt = 1.f - xv * xv;
            else if (gate_act == OCTA_ACT_SIGMOID) gt = xv * (1.f - xv);
            else if (gate_act == OCTA_ACT_LEAKY02) gt = xv > 0.f ? 1.f : 0.2f;
            else if (gate_act == OCTA_ACT_RELU) gt = xv > 0.f ? 1.f : 0.f;
            if (dx) DT<T>::st(dx + b * n + i, d * wi * gt);
            g += d * xv;
        }
        if (dw) dw[dw_c > 0 ? (i % dw_c) * hw + i / dw_c : i] += g;      // dw_c: dw is [C][HW] (the parameter's own OIHW order)
    }
    if (dbias && blockIdx.x == 0 && threadIdx.x == 0) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += sign * dout[b];
        dbias[0] += s;
    }
}
extern "C" int octa_fullconv_bwd_gated(const void* x, const float* w, const float* dout, void* dx, float* dw, float* dbias, int B, int64_t n,
                                       int dtype, float sign, const float* sign_dev, int dw_c, int gate_act, octa_stream_t stream);
extern "C" int octa_fullconv_bwd(const void* x, const float* w, const float* dout, void* dx, float* dw, float* dbias, int B, int64_t n,
                                 int dtype, float sign, const float* sign_dev, int dw_c, octa_stream_t stream) {
    return octa_fullconv_bwd_gated(x, w, dout, dx, dw, dbias, B, n, dtype, sign, sign_dev, dw_c, 0, stream);
}
extern "C" int octa_fullconv_bwd_gated(const void* x, const float* w, const float* dout, void* dx, float* dw, float* dbias, int B, int64_t n,
                                       int dtype, float sign, const float* sign_dev, int dw_c, int gate_act, octa_stream_t stream) {
    OCTA_REQUIRE(x && w && dout && B > 0 && n > 0 && dw_c >= 0 && (dw_c == 0 || n % dw_c == 0), "octa_fullconv_bwd: bad arguments");
    OCTA_REQUIRE(gate_act >= 0 && gate_act <= OCTA_ACT_TANH, "octa_fullconv_bwd_gated: activation code");
    hipStream_t st = (hipStream_t)stream;
    const int blocks = (int)(cdiv64(n, 256) > 2048 ? 2048 : cdiv64(n, 256));
    if (dtype == OCTA_F32) fullconv_bwd_kernel<float><<<blocks, 256, 0, st>>>((const float*)x, w, dout, (float*)dx, dw, dbias, B, n, sign, sign_dev, dw_c, gate_act);
    else if (dtype == OCTA_BF16) fullconv_bwd_kernel<bf16_t><<<blocks, 256, 0, st>>>((const bf16_t*)x, w, dout, (bf16_t*)dx, dw, dbias, B, n, sign, sign_dev, dw_c, gate_act);
    else if (dtype == OCTA_F16) fullconv_bwd_kernel<f16_t><<<blocks, 256, 0, st>>>((const f16_t*)x, w, dout, (f16_t*)dx, dw, dbias, B, n, sign, sign_dev, dw_c, gate_act);
    else OCTA_FAIL(OCTA_ERR_BAD_ARG, "octa_fullconv_bwd: bad dtype");
    OCTA_CHECK_LAUNCH("fullconv_bwd");
    return OCTA_OK;
}
