// Loss kernels: fused (softmax +) WeightedPartialCE + Dice, interlayer KL with the nearest
// up-sampling folded in, LS-GAN.  Algorithmic traffic = one read of each class map; sums are
// wavefront shuffles -> block LDS -> per-block partials, finalised in double by one block.
#include "common.hpp"

#define KMAX 8

struct Strides4 { int64_t b, c, h, w; };

extern "C" size_t octa_loss_workspace_floats(int B, int K) {
    // partials [B][NBLK_MAX][2K+2] + final block (2K weights/counts + 2B dice terms + 8)
    return (size_t)B * 256 * (2 * KMAX + 2) + 4 * KMAX + 2 * (size_t)B + 16;
}
static inline float* loss_final(float* ws, int B) { return ws + (size_t)B * 256 * (2 * KMAX + 2); }
static inline const float* loss_final(const float* ws, int B) { return ws + (size_t)B * 256 * (2 * KMAX + 2); }

// per pixel: p = softmax(in) or in; partial sums per block:
//   [0..K)  n_c   = sum ys_c
//   [K..2K) S_c   = sum ys_c * log(p~_c + 1e-12),  p~ = p*ys (or p when full)
//   [2K]    inter = sum p*ys      [2K+1] card = sum (p + ys)
template <int K>
__global__ __launch_bounds__(256) void wpce_dice_partial_kernel(const float* __restrict__ in, Strides4 si, const float* __restrict__ ys, Strides4 st,
                                                                int H, int W, int from_logits, int full, float* __restrict__ partial, int nblk) {
    __shared__ float red[(2 * K + 2) * 16];
    const int b = blockIdx.y;
    const int HW = H * W;
    float acc[2 * K + 2];
#pragma unroll
    for (int i = 0; i < 2 * K + 2; ++i) acc[i] = 0.f;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += nblk * 256) {
        const int h = p / W, w = p % W;
        float pv[K], tv[K];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            pv[k] = in[b * si.b + k * si.c + h * si.h + w * si.w];
            tv[k] = ys[b * st.b + k * st.c + h * st.h + w * st.w];
            mx = fmaxf(mx, pv[k]);
        }
        if (from_logits) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) { pv[k] = expf(pv[k] - mx); s += pv[k]; }
            const float inv = 1.f / s;
#pragma unroll
            for (int k = 0; k < K; ++k) pv[k] *= inv;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            acc[k] += tv[k];
            const float pm = full ? pv[k] : pv[k] * tv[k];
            acc[K + k] += tv[k] * logf(pm + 1e-12f);
            acc[2 * K] += pv[k] * tv[k];
            acc[2 * K + 1] += pv[k] + tv[k];
        }
    }
    block_sum<2 * K + 2>(acc, red);
    if (threadIdx.x == 0) {
        float* dst = partial + ((size_t)b * nblk + blockIdx.x) * (2 * K + 2);
#pragma unroll
        for (int i = 0; i < 2 * K + 2; ++i) dst[i] = acc[i];
    }
}

// final[0..K) = w_c ; final[K..2K) = n_c ; final[2K + 2b] = inter_b ; final[2K + 2b + 1] = card_b
template <int K>
__global__ __launch_bounds__(64) void wpce_dice_final_kernel(const float* __restrict__ partial, int B, int nblk, int64_t npix, int reduction_sum,
                                                             float* __restrict__ fin, float* __restrict__ out) {
    // one wavefront: lanes stride over the block partials of a sample, wave_sum_d folds them
    const int lane = threadIdx.x;
    double n[K], S[K];
    for (int k = 0; k < K; ++k) { n[k] = 0.0; S[k] = 0.0; }
    double dice = 0.0;
    for (int b = 0; b < B; ++b) {
        double inter = 0.0, card = 0.0, nb[K], Sb[K];
        for (int k = 0; k < K; ++k) { nb[k] = 0.0; Sb[k] = 0.0; }
        for (int j = lane; j < nblk; j += 64) {
            const float* p = partial + ((size_t)b * nblk + j) * (2 * K + 2);
            for (int k = 0; k < K; ++k) { nb[k] += (double)p[k]; Sb[k] += (double)p[K + k]; }
            inter += (double)p[2 * K]; card += (double)p[2 * K + 1];
        }
        inter = wave_sum_d(inter); card = wave_sum_d(card);
        for (int k = 0; k < K; ++k) { n[k] += wave_sum_d(nb[k]); S[k] += wave_sum_d(Sb[k]); }
        if (lane == 0) { fin[2 * K + 2 * b] = (float)inter; fin[2 * K + 2 * b + 1] = (float)card; }
        dice += 1.0 - 2.0 * inter / (card + 1e-12);
    }
    if (lane != 0) return;
    double ntot = 0.0;
    for (int k = 0; k < K; ++k) ntot += n[k];
    double l = 0.0;
    for (int k = 0; k < K; ++k) {
        const float wk = (float)ntot / ((float)n[k] + 1e-12f);   // fp32 like the reference (losses.py:38)
        fin[k] = wk;
        fin[K + k] = (float)n[k];
        l -= (double)wk * S[k];
    }
    out[0] = (float)(reduction_sum ? l : l / (double)npix);
    out[1] = (float)(dice / (double)B);
}

template <int K>
__global__ __launch_bounds__(256) void wpce_dice_bwd_kernel(const float* __restrict__ in, Strides4 si, const float* __restrict__ ys, Strides4 st,
                                                            int B, int H, int W, int from_logits, int full, int reduction_sum,
                                                            const float* __restrict__ gw, const float* __restrict__ gd,
                                                            const float* __restrict__ fin, float* __restrict__ din, Strides4 sd) {
    const int64_t total = (int64_t)B * H * W;
    const float g_w = gw ? gw[0] : 0.f, g_d = gd ? gd[0] : 0.f;
    const float scale_w = reduction_sum ? g_w : g_w / (float)total;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int w = (int)(i % W);
        const int h = (int)((i / W) % H);
        const int b = (int)(i / ((int64_t)W * H));
        float pv[K], tv[K], g[K];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            pv[k] = in[b * si.b + k * si.c + h * si.h + w * si.w];
            tv[k] = ys[b * st.b + k * st.c + h * st.h + w * st.w];
            mx = fmaxf(mx, pv[k]);
        }
        if (from_logits) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) { pv[k] = expf(pv[k] - mx); s += pv[k]; }
            const float inv = 1.f / s;
#pragma unroll
            for (int k = 0; k < K; ++k) pv[k] *= inv;
        }
        const float inter = fin[2 * K + 2 * b], card = fin[2 * K + 2 * b + 1] + 1e-12f;
        const float dscale = -g_d * 2.f / (float)B / (card * card);
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            // wpce: -w_k t_k d/dp log(p t + eps) = -w_k t_k t / (p t + eps)   (full: -w_k t / (p + eps))
            const float den = full ? (pv[k] + 1e-12f) : (pv[k] * tv[k] + 1e-12f);
            const float num = full ? tv[k] : tv[k] * tv[k];
            g[k] = -scale_w * fin[k] * num / den + dscale * (tv[k] * card - inter);
            dot += g[k] * pv[k];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) din[b * sd.b + k * sd.c + h * sd.h + w * sd.w] = from_logits ? pv[k] * (g[k] - dot) : g[k];
    }
}

#define LOSS_K_SWITCH(K, ...) switch (K) { case 2: { constexpr int KK = 2; __VA_ARGS__ } break; case 3: { constexpr int KK = 3; __VA_ARGS__ } break; \
    case 4: { constexpr int KK = 4; __VA_ARGS__ } break; default: OCTA_FAIL(OCTA_ERR_UNSUPPORTED, "loss: num_classes %d not in 2..4", K); }

extern "C" int octa_wpce_dice_fwd(const float* in, const int64_t* is, const float* ys, const int64_t* yst, int B, int K, int H, int W,
                                  int from_logits, int full, int reduction_sum, float* out, float* ws, octa_stream_t stream) {
    OCTA_REQUIRE(in && is && ys && yst && out && ws && B > 0 && H > 0 && W > 0, "octa_wpce_dice_fwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    int nblk = cdiv(H * W, 256 * 4);
    if (nblk > 256) nblk = 256;
    if (nblk < 1) nblk = 1;
    Strides4 si{is[0], is[1], is[2], is[3]}, sy{yst[0], yst[1], yst[2], yst[3]};
    LOSS_K_SWITCH(K,
        wpce_dice_partial_kernel<KK><<<dim3(nblk, B), 256, 0, st>>>(in, si, ys, sy, H, W, from_logits, full, ws, nblk);
        OCTA_CHECK_LAUNCH("wpce_dice_partial");
        wpce_dice_final_kernel<KK><<<1, 64, 0, st>>>(ws, B, nblk, (int64_t)B * H * W, reduction_sum, loss_final(ws, B), out);
        OCTA_CHECK_LAUNCH("wpce_dice_final");)
    return OCTA_OK;
}
extern "C" int octa_wpce_dice_bwd(const float* in, const int64_t* is, const float* ys, const int64_t* yst, int B, int K, int H, int W,
                                  int from_logits, int full, int reduction_sum, const float* g_wpce, const float* g_dice, const float* ws,
                                  float* din, const int64_t* ds, octa_stream_t stream) {
    OCTA_REQUIRE(in && is && ys && yst && ws && din && ds, "octa_wpce_dice_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    Strides4 si{is[0], is[1], is[2], is[3]}, sy{yst[0], yst[1], yst[2], yst[3]}, sd{ds[0], ds[1], ds[2], ds[3]};
    const int64_t total = (int64_t)B * H * W;
    int blocks = (int)(cdiv64(total, 256) > 4096 ? 4096 : cdiv64(total, 256));
    LOSS_K_SWITCH(K,
        wpce_dice_bwd_kernel<KK><<<blocks, 256, 0, st>>>(in, si, ys, sy, B, H, W, from_logits, full, reduction_sum, g_wpce, g_dice, loss_final(ws, B), din, sd);
        OCTA_CHECK_LAUNCH("wpce_dice_bwd");)
    return OCTA_OK;
}

// ------------------------------------------------------------------------------------------ interlayer KL
#define KL_MAX_MAPS 8
struct KlMaps { const float* p[KL_MAX_MAPS]; float* d[KL_MAX_MAPS]; int shift[KL_MAX_MAPS]; float w[KL_MAX_MAPS]; int n; float wsum; };
// (optional fan-out addends of the backward kernels: the gradient another consumer of a map produced, same dense fp32 layout)

template <int K>
__global__ __launch_bounds__(256) void kl_fwd_kernel(const float* __restrict__ basis, KlMaps mp, int B, int H, int W, float* __restrict__ partial) {
    __shared__ float red[16];
    // 32-bit pixel decode (the host checks B*K*H*W < 2^31): the 64-bit divisions and the index chains of this loop were as
    // expensive as its 12 logarithms per pixel (60 us for 16 x 2 x 400 x 400; the bytes would take 12)
    const int total = B * H * W, HW = H * W;
    float acc[1] = {0.f};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int b = i / HW, r = i - b * HW;
        const int h = r / W, w = r - h * W;
        float m[K];
#pragma unroll
        for (int k = 0; k < K; ++k) m[k] = 0.f;
        for (int j = 0; j < mp.n; ++j) {
            const int s = mp.shift[j];
            const int hs = H >> s, wsz = W >> s;
            const float* q = mp.p[j] + ((size_t)b * K * hs + (h >> s)) * wsz + (w >> s);
            const float wj = mp.w[j];
#pragma unroll
            for (int k = 0; k < K; ++k) m[k] += logf(q[(size_t)k * hs * wsz] * wj + 1e-12f);
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float P = basis[((size_t)b * K + k) * HW + r];
            acc[0] += P * (logf(P + 1e-12f) - m[k] / mp.wsum);
        }
    }
    block_sum<1>(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc[0];
}
__global__ void kl_final_kernel(const float* __restrict__ partial, int n, int64_t npix, float* __restrict__ out) {
    double s = 0.0;                       // one wavefront: a single thread walking 1024 partials took 43 us
    for (int i = threadIdx.x; i < n; i += 64) s += (double)partial[i];
    s = wave_sum_d(s);
    if (threadIdx.x != 0) return;
    const float l = (float)(s / (double)npix);
    out[0] = l;
    out[1] = (l != l) ? 1.f : 0.f;
}
extern "C" int octa_interlayer_kl_fwd(const float* basis, const float* const* maps, const int* shifts, const float* weights, int n_maps,
                                      float wsum, int B, int K, int H, int W, float* out, float* ws, octa_stream_t stream) {
    OCTA_REQUIRE(basis && maps && shifts && weights && out && ws && n_maps >= 1 && n_maps <= KL_MAX_MAPS, "octa_interlayer_kl_fwd: bad arguments");
    KlMaps mp;
    mp.n = n_maps; mp.wsum = wsum;
    for (int j = 0; j < n_maps; ++j) {
        OCTA_REQUIRE(maps[j] && shifts[j] >= 0 && ((H >> shifts[j]) << shifts[j]) == H && ((W >> shifts[j]) << shifts[j]) == W,
                     "octa_interlayer_kl_fwd: map %d is not an integer power-of-two reduction of the basis", j);
        mp.p[j] = maps[j]; mp.d[j] = nullptr; mp.shift[j] = shifts[j]; mp.w[j] = weights[j];
    }
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = (int64_t)B * H * W;
    OCTA_REQUIRE(total * K < (1ll << 31), "octa_interlayer_kl_fwd: B*K*H*W must be below 2^31");
    int blocks = (int)(cdiv64(total, 256 * 4) > 1024 ? 1024 : cdiv64(total, 256 * 4));
    if (blocks < 1) blocks = 1;
    LOSS_K_SWITCH(K, kl_fwd_kernel<KK><<<blocks, 256, 0, st>>>(basis, mp, B, H, W, ws); OCTA_CHECK_LAUNCH("kl_fwd");)
    kl_final_kernel<<<1, 64, 0, st>>>(ws, blocks, total, out);
    OCTA_CHECK_LAUNCH("kl_final");
    return OCTA_OK;
}

// d basis: g/N * (log(P+eps) - m + P/(P+eps))
template <int K>
__global__ __launch_bounds__(256) void kl_bwd_basis_kernel(const float* __restrict__ basis, KlMaps mp, int B, int H, int W, const float* __restrict__ g,
                                                           float* __restrict__ dbasis, const float* __restrict__ add) {
    const int total = B * K * H * W, HW = H * W;          // (< 2^31: checked by the host; 32-bit pixel decode as in kl_fwd_kernel)
    const float gs = g[0] / (float)((int64_t)B * H * W);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int bk = i / HW, r = i - bk * HW;
        const int h = r / W, w = r - h * W;
        const float P = basis[i];
        float m = 0.f;
        for (int j = 0; j < mp.n; ++j) {
            const int s = mp.shift[j];
            m += logf(mp.p[j][((size_t)bk * (H >> s) + (h >> s)) * (W >> s) + (w >> s)] * mp.w[j] + 1e-12f);
        }
        const float d = gs * (logf(P + 1e-12f) - m / mp.wsum + P / (P + 1e-12f));
        dbasis[i] = add ? d + add[i] : d;
    }
}
// d map j (gather form, one thread per source pixel): -g/N * w/(w Q + eps)/wsum * sum_{block} P
__global__ __launch_bounds__(256) void kl_bwd_map_kernel(const float* __restrict__ basis, const float* __restrict__ q, float wgt, float wsum, int shift,
                                                         int64_t BK, int H, int W, int64_t npix, const float* __restrict__ g, float* __restrict__ dq,
                                                         const float* __restrict__ add) {
    // f = 2^shift adjacent lanes per source pixel, one basis row of the f x f block each, then a butterfly over the f lanes (the
    // coarse maps were 20 000 threads walking 256 strided values each: 30 us for the 25 x 25 map)
    const int hs = H >> shift, wsz = W >> shift, f = 1 << shift;
    const int64_t total = BK * hs * wsz * f;
    const float gs = -g[0] / (float)npix / wsum;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < ((total + 255) / 256) * 256; t += (int64_t)gridDim.x * 256) {
        const bool live = t < total;
        const int64_t i = (live ? t : total - 1) >> shift;
        const int dy = (int)((live ? t : total - 1) & (f - 1));
        const int w = (int)(i % wsz);
        const int h = (int)((i / wsz) % hs);
        const int64_t bk = i / ((int64_t)wsz * hs);
        const float* row = basis + (bk * H + (h << shift) + dy) * W + (w << shift);
        float s = 0.f;
        for (int dx = 0; dx < f; ++dx) s += row[dx];
        for (int o = 1; o < f; o <<= 1) s += __shfl_xor(s, o);          // f <= 64 divides the wave: the f lanes of a pixel are adjacent
        if (live && dy == 0) {
            const float d = gs * s * wgt / (q[i] * wgt + 1e-12f);
            dq[i] = add ? d + add[i] : d;
        }
    }
}
extern "C" int octa_interlayer_kl_bwd_add(const float* basis, const float* const* maps, const int* shifts, const float* weights, int n_maps,
                                          float wsum, int B, int K, int H, int W, const float* g, float* dbasis, float* const* dmaps,
                                          const float* basis_addend, const float* const* map_addends, octa_stream_t stream) {
    OCTA_REQUIRE(basis && maps && shifts && weights && g && dmaps && n_maps >= 1 && n_maps <= KL_MAX_MAPS, "octa_interlayer_kl_bwd: bad arguments");
    KlMaps mp;
    mp.n = n_maps; mp.wsum = wsum;
    for (int j = 0; j < n_maps; ++j) { mp.p[j] = maps[j]; mp.d[j] = dmaps[j]; mp.shift[j] = shifts[j]; mp.w[j] = weights[j]; }
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = (int64_t)B * K * H * W;
    OCTA_REQUIRE(total < (1ll << 31), "octa_interlayer_kl_bwd: B*K*H*W must be below 2^31");
    if (dbasis) {
        int blocks = (int)(cdiv64(total, 256) > 4096 ? 4096 : cdiv64(total, 256));
        LOSS_K_SWITCH(K, kl_bwd_basis_kernel<KK><<<blocks, 256, 0, st>>>(basis, mp, B, H, W, g, dbasis, basis_addend); OCTA_CHECK_LAUNCH("kl_bwd_basis");)
    }
    for (int j = 0; j < n_maps; ++j) {
        if (!dmaps[j]) continue;
        OCTA_REQUIRE(shifts[j] <= 6, "octa_interlayer_kl_bwd: maps down to 1/64 of the basis (2^shift lanes of one wave per source pixel)");
        const int64_t tj = (total >> (2 * shifts[j])) << shifts[j];            // 2^shift lanes per source pixel
        int blocks = (int)(cdiv64(tj, 256) > 4096 ? 4096 : cdiv64(tj, 256));
        if (blocks < 1) blocks = 1;
        kl_bwd_map_kernel<<<blocks, 256, 0, st>>>(basis, maps[j], weights[j], wsum, shifts[j], (int64_t)B * K, H, W, (int64_t)B * H * W, g, dmaps[j],
                                                     map_addends ? map_addends[j] : nullptr);
        OCTA_CHECK_LAUNCH("kl_bwd_map");
    }
    return OCTA_OK;
}
extern "C" int octa_interlayer_kl_bwd(const float* basis, const float* const* maps, const int* shifts, const float* weights, int n_maps,
                                      float wsum, int B, int K, int H, int W, const float* g, float* dbasis, float* const* dmaps,
                                      octa_stream_t stream) {
    return octa_interlayer_kl_bwd_add(basis, maps, shifts, weights, n_maps, wsum, B, K, H, W, g, dbasis, dmaps, nullptr, nullptr, stream);
}

// ------------------------------------------------------------------------------------------ LS-GAN
__global__ void lsgan_fwd_kernel(const float* __restrict__ real, const float* __restrict__ fake, int nr, int nf, int mode, float* __restrict__ out) {
    __shared__ float red[32];
    float acc[2] = {0.f, 0.f};
    if (mode == 1) for (int i = threadIdx.x; i < nr; i += blockDim.x) { const float d = real[i] - 1.f; acc[0] += d * d; }
    for (int i = threadIdx.x; i < nf; i += blockDim.x) { const float d = fake[i] + (mode == 1 ? 1.f : -1.f); acc[1] += d * d; }
    block_sum<2>(acc, red);
    if (threadIdx.x == 0) out[0] = (mode == 1 ? 0.5f * acc[0] / (float)nr : 0.f) + 0.5f * acc[1] / (float)nf;
}
__global__ void lsgan_bwd_kernel(const float* __restrict__ real, const float* __restrict__ fake, int nr, int nf, int mode, const float* __restrict__ g,
                                 float* __restrict__ dr, float* __restrict__ df) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float gs = g[0];
    if (mode == 1 && dr && i < nr) dr[i] = gs * (real[i] - 1.f) / (float)nr;
    if (df && i < nf) df[i] = gs * (fake[i] + (mode == 1 ? 1.f : -1.f)) / (float)nf;
}
extern "C" int octa_lsgan_fwd(const float* real, const float* fake, int n_real, int n_fake, int mode, float* out, octa_stream_t stream) {
    OCTA_REQUIRE(fake && out && n_fake > 0 && (mode == 0 || (real && n_real > 0)), "octa_lsgan_fwd: bad arguments");
    lsgan_fwd_kernel<<<1, 256, 0, (hipStream_t)stream>>>(real, fake, n_real, n_fake, mode, out);
    OCTA_CHECK_LAUNCH("lsgan_fwd");
    return OCTA_OK;
}
extern "C" int octa_lsgan_bwd(const float* real, const float* fake, int n_real, int n_fake, int mode, const float* g, float* d_real,
                              float* d_fake, octa_stream_t stream) {
    OCTA_REQUIRE(fake && g && n_fake > 0, "octa_lsgan_bwd: bad arguments");
    const int n = n_real > n_fake ? n_real : n_fake;
    lsgan_bwd_kernel<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(real, fake, n_real, n_fake, mode, g, d_real, d_fake);
    OCTA_CHECK_LAUNCH("lsgan_bwd");
    return OCTA_OK;
}

// ------------------------------------------------------------------------------------------ combining the loss terms
// total = ((a0 wa0 + a1 wa1) + b0 wb0 + b1 wb1) + c0 wc0 (terms of weight 0 are skipped, not multiplied: a NaN flag rides in such a slot), scaled = total x scale:
// what `l[0] + l[1] + kl_w * kl[0] + adv_w * g_adv` and `* loss_scale` did in six ATen launches forward and a dozen backward (selects, fills, copies, adds
// of 4-byte tensors, each a 5 us slot on the critical path between the forward and the backward pass).  Same operations in the same order, unfused.
__global__ void loss_combine_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c, float wa0, float wa1, float wb0,
                                        float wb1, float wc0, const float* __restrict__ scale_dev, float scale_host, float* __restrict__ total,
                                        float* __restrict__ scaled) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float t = 0.f;
    bool any = false;
    auto add = [&](const float* p, int i, float w) {
        if (!p || w == 0.f) return;
        const float v = w == 1.f ? p[i] : __fmul_rn(w, p[i]);
        t = any ? __fadd_rn(t, v) : v;
        any = true;
    };
    add(a, 0, wa0); add(a, 1, wa1); add(b, 0, wb0); add(b, 1, wb1); add(c, 0, wc0);
    total[0] = t;
    scaled[0] = __fmul_rn(t, scale_dev ? scale_dev[0] : scale_host);
}
__global__ void loss_combine_bwd_kernel(const float* __restrict__ g, float wa0, float wa1, float wb0, float wb1, float wc0, const float* __restrict__ scale_dev,
                                        float scale_host, float* __restrict__ da, float* __restrict__ db, float* __restrict__ dc) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float gs = __fmul_rn(g[0], scale_dev ? scale_dev[0] : scale_host);
    auto w_ = [&](float w) { return w == 1.f ? gs : (w == 0.f ? 0.f : __fmul_rn(gs, w)); };
    if (da) { da[0] = w_(wa0); da[1] = w_(wa1); }
    if (db) { db[0] = w_(wb0); db[1] = w_(wb1); }
    if (dc) dc[0] = w_(wc0);
}
extern "C" int octa_loss_combine_fwd(const float* a, const float* b, const float* c, float wa0, float wa1, float wb0, float wb1, float wc0,
                                     const float* scale_dev, float scale_host, float* total, float* scaled, octa_stream_t stream) {
    OCTA_REQUIRE(total && scaled && (a || b || c), "octa_loss_combine_fwd: bad arguments");
    loss_combine_fwd_kernel<<<1, 64, 0, (hipStream_t)stream>>>(a, b, c, wa0, wa1, wb0, wb1, wc0, scale_dev, scale_host, total, scaled);
    OCTA_CHECK_LAUNCH("loss_combine_fwd");
    return OCTA_OK;
}
extern "C" int octa_loss_combine_bwd(const float* g, float wa0, float wa1, float wb0, float wb1, float wc0, const float* scale_dev, float scale_host,
                                     float* da, float* db, float* dc, octa_stream_t stream) {
    OCTA_REQUIRE(g && (da || db || dc), "octa_loss_combine_bwd: bad arguments");
    loss_combine_bwd_kernel<<<1, 64, 0, (hipStream_t)stream>>>(g, wa0, wa1, wb0, wb1, wc0, scale_dev, scale_host, da, db, dc);
    OCTA_CHECK_LAUNCH("loss_combine_bwd");
    return OCTA_OK;
}

// ------------------------------------------------------------------------------------------ per-pixel class softmax
// nn.Softmax(dim=1) on the (B, classes, H, W) logits (segmentor/compose.py:192): strided in, dense NCHW out.
template <int K>
__global__ __launch_bounds__(256) void class_softmax_fwd_kernel(const float* __restrict__ in, Strides4 si, float* __restrict__ out, int B, int H, int W) {
    const int64_t total = (int64_t)B * H * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int w = (int)(i % W);
        const int h = (int)((i / W) % H);
        const int b = (int)(i / ((int64_t)W * H));
        float v[K];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < K; ++k) { v[k] = in[b * si.b + k * si.c + h * si.h + w * si.w]; mx = fmaxf(mx, v[k]); }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) { v[k] = expf(v[k] - mx); s += v[k]; }
        const float inv = 1.f / s;
#pragma unroll
        for (int k = 0; k < K; ++k) out[(((int64_t)b * K + k) * H + h) * W + w] = v[k] * inv;
    }
}
template <int K>
__global__ __launch_bounds__(256) void class_softmax_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dp, Strides4 sd, float* __restrict__ din,
                                                                int B, int H, int W) {
    const int64_t total = (int64_t)B * H * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int w = (int)(i % W);
        const int h = (int)((i / W) % H);
        const int b = (int)(i / ((int64_t)W * H));
        float pv[K], g[K];
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            pv[k] = p[(((int64_t)b * K + k) * H + h) * W + w];
            g[k] = dp[b * sd.b + k * sd.c + h * sd.h + w * sd.w];
            dot += pv[k] * g[k];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) din[(((int64_t)b * K + k) * H + h) * W + w] = pv[k] * (g[k] - dot);
    }
}
extern "C" int octa_class_softmax_fwd(const float* in, const int64_t* is, float* out, int B, int K, int H, int W, octa_stream_t stream) {
    OCTA_REQUIRE(in && is && out && B > 0, "octa_class_softmax_fwd: bad arguments");
    Strides4 si{is[0], is[1], is[2], is[3]};
    const int64_t total = (int64_t)B * H * W;
    int blocks = (int)(cdiv64(total, 256) > 4096 ? 4096 : cdiv64(total, 256));
    hipStream_t st = (hipStream_t)stream;
    LOSS_K_SWITCH(K, class_softmax_fwd_kernel<KK><<<blocks, 256, 0, st>>>(in, si, out, B, H, W); OCTA_CHECK_LAUNCH("class_softmax_fwd");)
    return OCTA_OK;
}
extern "C" int octa_class_softmax_bwd(const float* p, const float* dp, const int64_t* ds, float* din, int B, int K, int H, int W, octa_stream_t stream) {
    OCTA_REQUIRE(p && dp && ds && din && B > 0, "octa_class_softmax_bwd: bad arguments");
    Strides4 sd{ds[0], ds[1], ds[2], ds[3]};
    const int64_t total = (int64_t)B * H * W;
    int blocks = (int)(cdiv64(total, 256) > 4096 ? 4096 : cdiv64(total, 256));
    hipStream_t st = (hipStream_t)stream;
    LOSS_K_SWITCH(K, class_softmax_bwd_kernel<KK><<<blocks, 256, 0, st>>>(p, dp, sd, din, B, H, W); OCTA_CHECK_LAUNCH("class_softmax_bwd");)
    return OCTA_OK;
}
