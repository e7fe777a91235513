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
                                   int ld, int cpad, int dtype, octa_stream_t stream) {
    OCTA_REQUIRE(src && ss && dst && cpad >= C && cpad <= ld, "octa_noise_clip_fwd: bad arguments");
    Strides4 s{ss[0], ss[1], ss[2], ss[3]};
    const int64_t total = (int64_t)B * H * W;
    const int blocks = (int)(cdiv64(total, 256) > 4096 ? 4096 : cdiv64(total, 256));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == OCTA_F32) noise_clip_fwd_kernel<float><<<blocks, 256, 0, st>>>(src, s, noise, (float*)dst, mask, B, C, H, W, ld, cpad, 1);
    else if (dtype == OCTA_BF16) noise_clip_fwd_kernel<bf16_t><<<blocks, 256, 0, st>>>(src, s, noise, (bf16_t*)dst, mask, B, C, H, W, ld, cpad, 1);
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
    else OCTA_FAIL(OCTA_ERR_BAD_ARG, "octa_noise_clip_bwd: bad dtype");
    OCTA_CHECK_LAUNCH("noise_clip_bwd");
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
__global__ __launch_bounds__(1024) void spectral_fwd_kernel(const float* __restrict__ w, float* __restrict__ u, float* __restrict__ v, int Cout,
                                                            int K, int iter, float eps, float* __restrict__ sigma, float* __restrict__ wsn) {
    extern __shared__ float sm[];   // us[Cout], vs[K], wv[Cout], red[16]
    float* us = sm;
    float* vs = us + Cout;
    float* wv = vs + K;
    float* red = wv + Cout;
    const int t = threadIdx.x, nt = blockDim.x;
    for (int i = t; i < Cout; i += nt) us[i] = u[i];
    for (int j = t; j < K; j += nt) vs[j] = v[j];
    __syncthreads();
    if (iter) {
        // v = normalize(W^T u): column j is owned by the threads {j, j+K', ...}: each sums a strided subset of rows
        // (consecutive threads read consecutive columns), partial sums meet in LDS
        const int KP = (K + 63) / 64 * 64;            // columns padded to whole waves
        const int G = nt / KP > 0 ? nt / KP : 1;      // row groups
        const int j = t % KP, gidx = t / KP;
        float s = 0.f;
        if (j < K && gidx < G)
            for (int i = gidx; i < Cout; i += G) s += w[(int64_t)i * K + j] * us[i];
        __syncthreads();
        if (gidx == 0 && j < K) vs[j] = 0.f;
        __syncthreads();
        if (j < K && gidx < G) atomicAdd(&vs[j], s);
        __syncthreads();
        float nrm = 0.f;
        for (int jj = t; jj < K; jj += nt) nrm += vs[jj] * vs[jj];
        nrm = sqrtf(block_total(nrm, red));
        const float dv = fmaxf(nrm, eps);
        for (int jj = t; jj < K; jj += nt) vs[jj] /= dv;
        __syncthreads();
    }
    // wv = W v  (one wave per row, lanes over K)
    const int lane = t & 63, wid = t >> 6, nw = nt >> 6;
    for (int i = wid; i < Cout; i += nw) {
        float s = 0.f;
        for (int j = lane; j < K; j += 64) s += w[(int64_t)i * K + j] * vs[j];
        s = wave_sum(s);
        if (lane == 0) wv[i] = s;
    }
    __syncthreads();
    if (iter) {
        float nrm = 0.f;
        for (int i = t; i < Cout; i += nt) nrm += wv[i] * wv[i];
        nrm = sqrtf(block_total(nrm, red));
        const float du = fmaxf(nrm, eps);
        for (int i = t; i < Cout; i += nt) us[i] = wv[i] / du;
        __syncthreads();
    }
    float sg = 0.f;
    for (int i = t; i < Cout; i += nt) sg += us[i] * wv[i];
    sg = block_total(sg, red);
    if (t == 0) sigma[0] = sg;
    if (iter) {
        for (int i = t; i < Cout; i += nt) u[i] = us[i];
        for (int j = t; j < K; j += nt) v[j] = vs[j];
    }
    const float inv = 1.f / sg;
    for (int64_t i = t; i < (int64_t)Cout * K; i += nt) wsn[i] = w[i] * inv;
}
extern "C" int octa_spectral_norm_fwd(const float* w, float* u, float* v, int Cout, int K, int do_power_iter, float eps, float* sigma,
                                      float* w_sn, octa_stream_t stream) {
    OCTA_REQUIRE(w && u && v && sigma && w_sn && Cout > 0 && K > 0 && K <= 1024, "octa_spectral_norm_fwd: bad arguments (K <= 1024)");
    const size_t sh = (size_t)(2 * Cout + K + 16) * sizeof(float);
    OCTA_REQUIRE(sh <= 60000, "octa_spectral_norm_fwd: matrix too large for the single-block kernel");
    spectral_fwd_kernel<<<1, 1024, sh, (hipStream_t)stream>>>(w, u, v, Cout, K, do_power_iter, eps, sigma, w_sn);
    OCTA_CHECK_LAUNCH("spectral_fwd");
    return OCTA_OK;
}
// dw += (dw_sn - (sum dw_sn * w_sn) u v^T) / sigma
__global__ __launch_bounds__(1024) void spectral_bwd_kernel(const float* __restrict__ dwsn, const float* __restrict__ wsn, const float* __restrict__ u,
                                                            const float* __restrict__ v, const float* __restrict__ sigma, int Cout, int K,
                                                            float* __restrict__ dw) {
    __shared__ float red[16];
    const int64_t n = (int64_t)Cout * K;
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += dwsn[i] * wsn[i];
    s = block_total(s, red);
    const float inv = 1.f / sigma[0];
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const int r = (int)(i / K), c = (int)(i % K);
        dw[i] += (dwsn[i] - s * u[r] * v[c]) * inv;
    }
}
extern "C" int octa_spectral_norm_bwd(const float* dw_sn, const float* w_sn, const float* u, const float* v, const float* sigma, int Cout, int K,
                                      float* dw, float* ws, octa_stream_t stream) {
    (void)ws;
    OCTA_REQUIRE(dw_sn && w_sn && u && v && sigma && dw, "octa_spectral_norm_bwd: bad arguments");
    spectral_bwd_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(dw_sn, w_sn, u, v, sigma, Cout, K, dw);
    OCTA_CHECK_LAUNCH("spectral_bwd");
    return OCTA_OK;
}

// ------------------------------------------------------------------------------------------ full-extent conv = dot product
template <typename T>
__global__ __launch_bounds__(256) void fullconv_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, float* __restrict__ out, int64_t n,
                                                           float sign, const float* __restrict__ sign_dev, int nblk) {
    __shared__ float red[16];
    if (sign_dev) sign *= sign_dev[0];
    const int b = blockIdx.y;
    float acc[1] = {0.f};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)nblk * 256) acc[0] += DT<T>::ld(x + b * n + i) * w[i];
    block_sum<1>(acc, red);
    if (threadIdx.x == 0) atomicAdd(out + b, sign * acc[0]);
}
__global__ void fullconv_init_kernel(const float* __restrict__ bias, float* __restrict__ out, int B, float sign, const float* __restrict__ sign_dev) {
    if (sign_dev) sign *= sign_dev[0];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) out[i] = sign * (bias ? bias[0] : 0.f);
}
extern "C" int octa_fullconv_fwd(const void* x, const float* w, const float* bias, float* out, int B, int64_t n, int dtype, float sign,
                                 const float* sign_dev, octa_stream_t stream) {
    OCTA_REQUIRE(x && w && out && B > 0 && n > 0, "octa_fullconv_fwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    fullconv_init_kernel<<<cdiv(B, 64), 64, 0, st>>>(bias, out, B, sign, sign_dev);
    OCTA_CHECK_LAUNCH("fullconv_init");
    int nblk = (int)(cdiv64(n, 256 * 8) > 64 ? 64 : cdiv64(n, 256 * 8));
    if (nblk < 1) nblk = 1;
    dim3 grid(nblk, B);
    if (dtype == OCTA_F32) fullconv_fwd_kernel<float><<<grid, 256, 0, st>>>((const float*)x, w, out, n, sign, sign_dev, nblk);
    else if (dtype == OCTA_BF16) fullconv_fwd_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)x, w, out, n, sign, sign_dev, nblk);
    else OCTA_FAIL(OCTA_ERR_BAD_ARG, "octa_fullconv_fwd: bad dtype");
    OCTA_CHECK_LAUNCH("fullconv_fwd");
    return OCTA_OK;
}
// dx[b,i] = sign*dout[b]*w[i];  dw[i] += sign * sum_b dout[b]*x[b,i];  dbias += sign * sum_b dout[b]
template <typename T>
__global__ __launch_bounds__(256) void fullconv_bwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ dout,
                                                           T* __restrict__ dx, float* __restrict__ dw, float* __restrict__ dbias, int B, int64_t n,
                                                           float sign, const float* __restrict__ sign_dev) {
    if (sign_dev) sign *= sign_dev[0];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float wi = w[i];
        float g = 0.f;
        for (int b = 0; b < B; ++b) {
            const float d = sign * dout[b];
            if (dx) DT<T>::st(dx + b * n + i, d * wi);
            g += d * DT<T>::ld(x + b * n + i);
        }
        if (dw) dw[i] += g;
    }
    if (dbias && blockIdx.x == 0 && threadIdx.x == 0) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += sign * dout[b];
        dbias[0] += s;
    }
}
extern "C" int octa_fullconv_bwd(const void* x, const float* w, const float* dout, void* dx, float* dw, float* dbias, int B, int64_t n,
                                 int dtype, float sign, const float* sign_dev, octa_stream_t stream) {
    OCTA_REQUIRE(x && w && dout && B > 0 && n > 0, "octa_fullconv_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int blocks = (int)(cdiv64(n, 256) > 2048 ? 2048 : cdiv64(n, 256));
    if (dtype == OCTA_F32) fullconv_bwd_kernel<float><<<blocks, 256, 0, st>>>((const float*)x, w, dout, (float*)dx, dw, dbias, B, n, sign, sign_dev);
    else if (dtype == OCTA_BF16) fullconv_bwd_kernel<bf16_t><<<blocks, 256, 0, st>>>((const bf16_t*)x, w, dout, (bf16_t*)dx, dw, dbias, B, n, sign, sign_dev);
    else OCTA_FAIL(OCTA_ERR_BAD_ARG, "octa_fullconv_bwd: bad dtype");
    OCTA_CHECK_LAUNCH("fullconv_bwd");
    return OCTA_OK;
}
