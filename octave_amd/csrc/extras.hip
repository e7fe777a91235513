// The rest of the reference's public surface around the hot path, all HBM / latency-bound streaming kernels:
//   - InterlayerDivergence JSD branch (segmentor/losses.py:154-169), nearest up-sampling fused like the KLD kernels
//   - WeightedPartialCE's nn.CrossEntropyLoss (manual=False) and nn.BCEWithLogitsLoss (num_classes == 1) branches (:40-60)
//   - LabelNoise mode 'label' (discriminator/blocks.py:172-177)
//   - ResnestUNet.predict post-processing: sigmoid / one-hot(argmax) (segmentor/compose.py:189-199), Dice coefficient
//   - nn.AdaptiveAvgPool2d of the classification head (segmentor/compose.py:88-98)
//   - device-side synthetic OCTA batches and the discriminator's real-mask pyramid (discriminator/blocks.py:114-125)
#include "common.hpp"
#include <stdlib.h>

struct XStrides4 { int64_t b, c, h, w; };
static inline int x_blocks(int64_t n, int cap = 4096) { int64_t b = cdiv64(n, 256); return (int)(b > cap ? cap : (b < 1 ? 1 : b)); }
#define X_K_SWITCH(K, NAME, ...) switch (K) { case 1: { constexpr int KK = 1; __VA_ARGS__ } break; case 2: { constexpr int KK = 2; __VA_ARGS__ } break; \
    case 3: { constexpr int KK = 3; __VA_ARGS__ } break; case 4: { constexpr int KK = 4; __VA_ARGS__ } break; \
    default: OCTA_FAIL(OCTA_ERR_UNSUPPORTED, NAME ": num_classes %d not in 1..4", K); }

__global__ void x_final_mean_kernel(const float* __restrict__ partial, int n, double denom, float* __restrict__ out) {
    if (threadIdx.x != 0) return;
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += (double)partial[i];
    const float l = (float)(s / denom);
    out[0] = l;
    out[1] = (l != l) ? 1.f : 0.f;
}

// ------------------------------------------------------------------------------------------ interlayer JSD
#define JS_MAX_MAPS 8
struct JsMaps { const float* p[JS_MAX_MAPS]; int shift[JS_MAX_MAPS]; float w[JS_MAX_MAPS]; int n; };

// mean_q = (1/n) sum_j w_j Q_j (nearest-resized); M = (P + mean_q)/2
// loss = mean_pix sum_k [ P/2 (log(P+1e-12) - log(M+eps)) + mean_q/2 (log(mean_q+1e-12) - log(M+eps)) ]
template <int K>
__global__ __launch_bounds__(256) void jsd_fwd_kernel(const float* __restrict__ basis, JsMaps mp, float eps, int B, int H, int W, float* __restrict__ partial) {
    __shared__ float red[16];
    const int64_t total = (int64_t)B * H * W;
    float acc[1] = {0.f};
    const float invn = 1.f / (float)mp.n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int w = (int)(i % W);
        const int h = (int)((i / W) % H);
        const int b = (int)(i / ((int64_t)W * H));
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float P = basis[(((int64_t)b * K + k) * H + h) * W + w];
            float mq = 0.f;
            for (int j = 0; j < mp.n; ++j) {
                const int s = mp.shift[j];
                mq += mp.w[j] * mp.p[j][(((int64_t)b * K + k) * (H >> s) + (h >> s)) * (W >> s) + (w >> s)];
            }
            mq *= invn;
            const float lM = logf(0.5f * (P + mq) + eps);
            acc[0] += 0.5f * P * (logf(P + 1e-12f) - lM) + 0.5f * mq * (logf(mq + 1e-12f) - lM);
        }
    }
    block_sum<1>(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc[0];
}
// dbasis = g/N dL/dP ; gq = g/N dL/dmean_q / n  (per fine pixel; folded over the nearest blocks by jsd_bwd_map_kernel)
template <int K>
__global__ __launch_bounds__(256) void jsd_bwd_fine_kernel(const float* __restrict__ basis, JsMaps mp, float eps, int B, int H, int W,
                                                           const float* __restrict__ g, float* __restrict__ dbasis, float* __restrict__ gq) {
    const int64_t total = (int64_t)B * K * H * W;
    const float gs = g[0] / (float)((int64_t)B * H * W);
    const float invn = 1.f / (float)mp.n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int w = (int)(i % W);
        const int h = (int)((i / W) % H);
        const int64_t bk = i / ((int64_t)W * H);
        const float P = basis[i];
        float mq = 0.f;
        for (int j = 0; j < mp.n; ++j) {
            const int s = mp.shift[j];
            mq += mp.w[j] * mp.p[j][(bk * (H >> s) + (h >> s)) * (W >> s) + (w >> s)];
        }
        mq *= invn;
        const float M = 0.5f * (P + mq) + eps;
        const float lM = logf(M);
        const float common = 0.25f * (P + mq) / M;
        if (dbasis) dbasis[i] = gs * (0.5f * (logf(P + 1e-12f) - lM + P / (P + 1e-12f)) - common);
        gq[i] = gs * invn * (0.5f * (logf(mq + 1e-12f) - lM + mq / (mq + 1e-12f)) - common);
    }
}
__global__ __launch_bounds__(256) void jsd_bwd_map_kernel(const float* __restrict__ gq, float wgt, int shift, int64_t BK, int H, int W, float* __restrict__ dq) {
    const int hs = H >> shift, wsz = W >> shift, f = 1 << shift;
    const int64_t total = BK * hs * wsz;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int w = (int)(i % wsz);
        const int h = (int)((i / wsz) % hs);
        const int64_t bk = i / ((int64_t)wsz * hs);
        float s = 0.f;
        for (int dy = 0; dy < f; ++dy)
            for (int dx = 0; dx < f; ++dx) s += gq[(bk * H + (h << shift) + dy) * W + (w << shift) + dx];
        dq[i] = wgt * s;
    }
}
static int js_fill(JsMaps& mp, const float* const* maps, const int* shifts, const float* weights, int n_maps, int H, int W, const char* who) {
    OCTA_REQUIRE(maps && shifts && weights && n_maps >= 1 && n_maps <= JS_MAX_MAPS, "%s: bad map list", who);
    mp.n = n_maps;
    for (int j = 0; j < n_maps; ++j) {
        OCTA_REQUIRE(maps[j] && shifts[j] >= 0 && ((H >> shifts[j]) << shifts[j]) == H && ((W >> shifts[j]) << shifts[j]) == W,
                     "%s: map %d is not an integer power-of-two reduction of the basis", who, j);
        mp.p[j] = maps[j]; mp.shift[j] = shifts[j]; mp.w[j] = weights[j];
    }
    return OCTA_OK;
}
extern "C" int octa_interlayer_jsd_fwd(const float* basis, const float* const* maps, const int* shifts, const float* weights, int n_maps, float eps,
                                       int B, int K, int H, int W, float* out, float* ws, octa_stream_t stream) {
    OCTA_REQUIRE(basis && out && ws, "octa_interlayer_jsd_fwd: null pointer");
    JsMaps mp;
    int rc = js_fill(mp, maps, shifts, weights, n_maps, H, W, "octa_interlayer_jsd_fwd");
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = (int64_t)B * H * W;
    const int blocks = x_blocks(cdiv64(total, 4), 1024);
    X_K_SWITCH(K, "octa_interlayer_jsd_fwd", jsd_fwd_kernel<KK><<<blocks, 256, 0, st>>>(basis, mp, eps, B, H, W, ws); OCTA_CHECK_LAUNCH("jsd_fwd");)
    x_final_mean_kernel<<<1, 64, 0, st>>>(ws, blocks, (double)total, out);
    OCTA_CHECK_LAUNCH("jsd_final");
    return OCTA_OK;
}
extern "C" int octa_interlayer_jsd_bwd(const float* basis, const float* const* maps, const int* shifts, const float* weights, int n_maps, float eps,
                                       int B, int K, int H, int W, const float* g, float* dbasis, float* gq_ws, float* const* dmaps,
                                       octa_stream_t stream) {
    OCTA_REQUIRE(basis && g && gq_ws && dmaps, "octa_interlayer_jsd_bwd: null pointer");
    JsMaps mp;
    int rc = js_fill(mp, maps, shifts, weights, n_maps, H, W, "octa_interlayer_jsd_bwd");
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = (int64_t)B * K * H * W;
    X_K_SWITCH(K, "octa_interlayer_jsd_bwd", jsd_bwd_fine_kernel<KK><<<x_blocks(total), 256, 0, st>>>(basis, mp, eps, B, H, W, g, dbasis, gq_ws); OCTA_CHECK_LAUNCH("jsd_bwd_fine");)
    for (int j = 0; j < n_maps; ++j) {
        if (!dmaps[j]) continue;
        jsd_bwd_map_kernel<<<x_blocks(total >> (2 * shifts[j])), 256, 0, st>>>(gq_ws, weights[j], shifts[j], (int64_t)B * K, H, W, dmaps[j]);
        OCTA_CHECK_LAUNCH("jsd_bwd_map");
    }
    return OCTA_OK;
}

// ------------------------------------------------------------------------------------------ pixel CE / BCE
// mode 0 (K == 2): nn.CrossEntropyLoss()(z, t), z_k = y_hat_k * ys_k (or y_hat_k when full), t = long(ys_1)
// mode 1 (K == 1): nn.BCEWithLogitsLoss()(z, ys)
template <int K>
__global__ __launch_bounds__(256) void pixel_ce_fwd_kernel(const float* __restrict__ in, XStrides4 si, const float* __restrict__ ys, XStrides4 st,
                                                           int B, int H, int W, int full, int mode, float* __restrict__ partial) {
    __shared__ float red[16];
    const int64_t total = (int64_t)B * H * W;
    float acc[1] = {0.f};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int w = (int)(i % W);
        const int h = (int)((i / W) % H);
        const int b = (int)(i / ((int64_t)W * H));
        float z[K], t[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            t[k] = ys[b * st.b + k * st.c + h * st.h + w * st.w];
            const float p = in[b * si.b + k * si.c + h * si.h + w * si.w];
            z[k] = full ? p : p * t[k];
        }
        if (mode == 1) acc[0] += fmaxf(z[0], 0.f) - z[0] * t[0] + log1pf(expf(-fabsf(z[0])));
        else {
            const int tgt = (int)(long)t[K - 1];        // K == 2: class index of the pixel
            float mx = z[0];
#pragma unroll
            for (int k = 1; k < K; ++k) mx = fmaxf(mx, z[k]);
            float s = 0.f, zt = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) { s += expf(z[k] - mx); if (k == tgt) zt = z[k]; }
            acc[0] += mx + logf(s) - zt;
        }
    }
    block_sum<1>(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc[0];
}
template <int K>
__global__ __launch_bounds__(256) void pixel_ce_bwd_kernel(const float* __restrict__ in, XStrides4 si, const float* __restrict__ ys, XStrides4 st,
                                                           int B, int H, int W, int full, int mode, const float* __restrict__ g, float* __restrict__ din) {
    const int64_t total = (int64_t)B * H * W;
    const float gs = g[0] / (float)(mode == 1 ? total * K : total);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int w = (int)(i % W);
        const int h = (int)((i / W) % H);
        const int b = (int)(i / ((int64_t)W * H));
        float z[K], t[K], dz[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            t[k] = ys[b * st.b + k * st.c + h * st.h + w * st.w];
            const float p = in[b * si.b + k * si.c + h * si.h + w * si.w];
            z[k] = full ? p : p * t[k];
        }
        if (mode == 1) dz[0] = 1.f / (1.f + expf(-z[0])) - t[0];
        else {
            const int tgt = (int)(long)t[K - 1];
            float mx = z[0];
#pragma unroll
            for (int k = 1; k < K; ++k) mx = fmaxf(mx, z[k]);
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) { dz[k] = expf(z[k] - mx); s += dz[k]; }
#pragma unroll
            for (int k = 0; k < K; ++k) dz[k] = dz[k] / s - (k == tgt ? 1.f : 0.f);
        }
#pragma unroll
        for (int k = 0; k < K; ++k) din[(((int64_t)b * K + k) * H + h) * W + w] = gs * dz[k] * (full ? 1.f : t[k]);
    }
}
extern "C" int octa_pixel_ce_fwd(const float* in, const int64_t* is, const float* ys, const int64_t* yst, int B, int K, int H, int W, int full, int mode,
                                 float* out, float* ws, octa_stream_t stream) {
    OCTA_REQUIRE(in && is && ys && yst && out && ws && B > 0 && H > 0 && W > 0, "octa_pixel_ce_fwd: bad arguments");
    OCTA_REQUIRE((mode == 0 && K == 2) || (mode == 1 && K == 1), "octa_pixel_ce_fwd: mode 0 needs 2 classes, mode 1 needs 1 (got mode %d, K %d)", mode, K);
    XStrides4 si{is[0], is[1], is[2], is[3]}, sy{yst[0], yst[1], yst[2], yst[3]};
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = (int64_t)B * H * W;
    const int blocks = x_blocks(cdiv64(total, 4), 1024);
    X_K_SWITCH(K, "octa_pixel_ce_fwd", pixel_ce_fwd_kernel<KK><<<blocks, 256, 0, st>>>(in, si, ys, sy, B, H, W, full, mode, ws); OCTA_CHECK_LAUNCH("pixel_ce_fwd");)
    x_final_mean_kernel<<<1, 64, 0, st>>>(ws, blocks, (double)total * (mode == 1 ? K : 1), out);
    OCTA_CHECK_LAUNCH("pixel_ce_final");
    return OCTA_OK;
}
extern "C" int octa_pixel_ce_bwd(const float* in, const int64_t* is, const float* ys, const int64_t* yst, int B, int K, int H, int W, int full, int mode,
                                 const float* g, float* din, octa_stream_t stream) {
    OCTA_REQUIRE(in && is && ys && yst && g && din, "octa_pixel_ce_bwd: bad arguments");
    OCTA_REQUIRE((mode == 0 && K == 2) || (mode == 1 && K == 1), "octa_pixel_ce_bwd: bad mode / class count");
    XStrides4 si{is[0], is[1], is[2], is[3]}, sy{yst[0], yst[1], yst[2], yst[3]};
    X_K_SWITCH(K, "octa_pixel_ce_bwd", pixel_ce_bwd_kernel<KK><<<x_blocks((int64_t)B * H * W), 256, 0, (hipStream_t)stream>>>(in, si, ys, sy, B, H, W, full, mode, g, din);
               OCTA_CHECK_LAUNCH("pixel_ce_bwd");)
    return OCTA_OK;
}

// ------------------------------------------------------------------------------------------ LabelNoise 'label': y = |1 - x|
__global__ void abs1m_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float d = 1.f - x[i];
        if (dy) out[i] = (d > 0.f ? -1.f : (d < 0.f ? 1.f : 0.f)) * dy[i];     // backward: d|1-x|/dx
        else out[i] = fabsf(d);
    }
}
extern "C" int octa_abs1m(const float* x, const float* dy, float* out, int64_t n, octa_stream_t stream) {
    OCTA_REQUIRE(x && out && n > 0, "octa_abs1m: bad arguments");
    abs1m_kernel<<<x_blocks(n), 256, 0, (hipStream_t)stream>>>(x, dy, out, n);
    OCTA_CHECK_LAUNCH("abs1m");
    return OCTA_OK;
}

// ------------------------------------------------------------------------------------------ predict post-processing
// mode 0: out_f32[b,k,h,w] = sigmoid(logits);  mode 1: out_i64[b,k,h,w] = (k == argmax_k logits) with the FIRST maximum winning
// (torch.argmax), maxclass[0] = max over pixels of the argmax (F.one_hot without num_classes sizes its output by it)
template <int K>
__global__ __launch_bounds__(256) void predict_map_kernel(const float* __restrict__ in, XStrides4 si, int B, int H, int W, int mode, float* __restrict__ outf,
                                                          long long* __restrict__ outi, int* __restrict__ maxclass) {
    const int64_t total = (int64_t)B * H * W;
    int mymax = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int w = (int)(i % W);
        const int h = (int)((i / W) % H);
        const int b = (int)(i / ((int64_t)W * H));
        float v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = in[b * si.b + k * si.c + h * si.h + w * si.w];
        if (mode == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) outf[(((int64_t)b * K + k) * H + h) * W + w] = 1.f / (1.f + expf(-v[k]));
        } else {
            int am = 0;
            float best = v[0];
#pragma unroll
            for (int k = 1; k < K; ++k) if (v[k] > best || (v[k] != v[k] && best == best)) { best = v[k]; am = k; }
#pragma unroll
            for (int k = 0; k < K; ++k) outi[(((int64_t)b * K + k) * H + h) * W + w] = (k == am) ? 1ll : 0ll;
            mymax = max(mymax, am);
        }
    }
    if (mode == 1 && maxclass && mymax > 0) atomicMax(maxclass, mymax);
}
extern "C" int octa_predict_map(const float* logits, const int64_t* ls, int B, int K, int H, int W, int mode, void* out, int* maxclass_zeroed,
                                octa_stream_t stream) {
    OCTA_REQUIRE(logits && ls && out && (mode == 0 || mode == 1), "octa_predict_map: bad arguments");
    XStrides4 si{ls[0], ls[1], ls[2], ls[3]};
    X_K_SWITCH(K, "octa_predict_map", predict_map_kernel<KK><<<x_blocks((int64_t)B * H * W), 256, 0, (hipStream_t)stream>>>(
                   logits, si, B, H, W, mode, (float*)out, (long long*)out, maxclass_zeroed);
               OCTA_CHECK_LAUNCH("predict_map");)
    return OCTA_OK;
}
// Dice coefficient terms: out[b][k][0] = sum_hw pred*target, out[b][k][1] = sum_hw (pred + target); block per (b, k)
__global__ __launch_bounds__(256) void dice_terms_kernel(const float* __restrict__ p, XStrides4 sp, const float* __restrict__ t, XStrides4 st, int K, int H, int W,
                                                         float* __restrict__ out) {
    __shared__ float red[32];
    const int b = blockIdx.x / K, k = blockIdx.x % K;
    float acc[2] = {0.f, 0.f};
    for (int i = threadIdx.x; i < H * W; i += 256) {
        const int h = i / W, w = i % W;
        const float a = p[b * sp.b + k * sp.c + h * sp.h + w * sp.w], c = t[b * st.b + k * st.c + h * st.h + w * st.w];
        acc[0] += a * c; acc[1] += a + c;
    }
    block_sum<2>(acc, red);
    if (threadIdx.x == 0) { out[(size_t)blockIdx.x * 2] = acc[0]; out[(size_t)blockIdx.x * 2 + 1] = acc[1]; }
}
extern "C" int octa_dice_terms(const float* pred, const int64_t* ps, const float* target, const int64_t* ts, int B, int K, int H, int W, float* out,
                               octa_stream_t stream) {
    OCTA_REQUIRE(pred && ps && target && ts && out && B > 0 && K > 0, "octa_dice_terms: bad arguments");
    XStrides4 sp{ps[0], ps[1], ps[2], ps[3]}, st{ts[0], ts[1], ts[2], ts[3]};
    dice_terms_kernel<<<B * K, 256, 0, (hipStream_t)stream>>>(pred, sp, target, st, K, H, W, out);
    OCTA_CHECK_LAUNCH("dice_terms");
    return OCTA_OK;
}

// ------------------------------------------------------------------------------------------ adaptive average pooling (NCHW fp32)
// window of output o along an axis of length L split into O parts: [floor(o L / O), ceil((o + 1) L / O))   (ATen rule)
__global__ __launch_bounds__(256) void adaptive_avgpool_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ out, int64_t BC,
                                                               int H, int W, int OH, int OW) {
    if (!dy) {
        const int64_t total = BC * OH * OW;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
            const int ow = (int)(i % OW), oh = (int)((i / OW) % OH);
            const int64_t bc = i / ((int64_t)OW * OH);
            const int h0 = (oh * H) / OH, h1 = ((oh + 1) * H + OH - 1) / OH, w0 = (ow * W) / OW, w1 = ((ow + 1) * W + OW - 1) / OW;
            float s = 0.f;
            for (int h = h0; h < h1; ++h)
                for (int w = w0; w < w1; ++w) s += x[(bc * H + h) * W + w];
            out[i] = s / (float)((h1 - h0) * (w1 - w0));
        }
    } else {   // backward, gather form: dx[h][w] = sum over the windows that contain (h, w) (candidates around h*OH/H, exact test)
        const int64_t total = BC * H * W;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
            const int w = (int)(i % W), h = (int)((i / W) % H);
            const int64_t bc = i / ((int64_t)W * H);
            const int oh_lo = max(0, (h * OH) / H - 1), oh_hi = min(OH - 1, ((h + 1) * OH + H - 1) / H);
            const int ow_lo = max(0, (w * OW) / W - 1), ow_hi = min(OW - 1, ((w + 1) * OW + W - 1) / W);
            float s = 0.f;
            for (int oh = oh_lo; oh <= oh_hi; ++oh) {
                const int h0 = (oh * H) / OH, h1 = ((oh + 1) * H + OH - 1) / OH;
                if (h < h0 || h >= h1) continue;
                for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                    const int w0 = (ow * W) / OW, w1 = ((ow + 1) * W + OW - 1) / OW;
                    if (w < w0 || w >= w1) continue;
                    s += dy[(bc * OH + oh) * OW + ow] / (float)((h1 - h0) * (w1 - w0));
                }
            }
            out[i] = s;
        }
    }
}
extern "C" int octa_adaptive_avgpool(const float* x, const float* dy, float* out, int64_t BC, int H, int W, int OH, int OW, octa_stream_t stream) {
    OCTA_REQUIRE((x || dy) && out && BC > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "octa_adaptive_avgpool: bad arguments");
    const int64_t total = dy ? BC * H * W : BC * OH * OW;
    adaptive_avgpool_kernel<<<x_blocks(total), 256, 0, (hipStream_t)stream>>>(x, dy, out, BC, H, W, OH, OW);
    OCTA_CHECK_LAUNCH("adaptive_avgpool");
    return OCTA_OK;
}

// ------------------------------------------------------------------------------------------ synthetic OCTA batches
// Counter-based generator (one 64-bit mix per sample: no state, any launch shape gives the same stream).
__device__ __forceinline__ float x_uniform(unsigned long long seed, unsigned long long idx) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (idx + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.f / 16777216.f);     // 24 random bits -> [0, 1)
}
// x (B,3,H,W): grayscale plane replicated to 3 channels.  vessel == 0: U[0,1); vessel != 0: a curvilinear
// "vessel" field (sum of three sinusoidal ridges with per-image random phase) + noise.
// ys (B,2,H,W): scribbles, ~5 % of the pixels labelled class 1 and ~5 % class 0, the rest unlabelled (all zero).
// real (B,2,H,W): dense binary mask, one-hot (20 % foreground; the thresholded vessel field when vessel != 0).
__global__ __launch_bounds__(256) void synth_octa_kernel(unsigned long long seed, int B, int H, int W, int vessel, float* __restrict__ x,
                                                         float* __restrict__ ys, float* __restrict__ real) {
    const int64_t HW = (int64_t)H * W, total = (int64_t)B * HW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / HW, p = i - b * HW;
        const int h = (int)(p / W), w = (int)(p % W);
        const float u0 = x_uniform(seed, (unsigned long long)i * 4ull), u1 = x_uniform(seed, (unsigned long long)i * 4ull + 1ull),
                    u2 = x_uniform(seed, (unsigned long long)i * 4ull + 2ull);
        float img = u0;
        bool fg = u2 > 0.8f;
        if (vessel) {
            const float ph0 = 6.2831853f * x_uniform(seed ^ 0x5555ull, (unsigned long long)b * 8ull), ph1 = 6.2831853f * x_uniform(seed ^ 0x5555ull, (unsigned long long)b * 8ull + 1ull),
                        ph2 = 6.2831853f * x_uniform(seed ^ 0x5555ull, (unsigned long long)b * 8ull + 2ull);
            const float fy = (float)h / (float)H, fx = (float)w / (float)W;
            const float r0 = fabsf(__sinf(19.f * fx + 7.f * __sinf(5.f * fy + ph0) + ph1)), r1 = fabsf(__sinf(23.f * fy + 5.f * __sinf(7.f * fx + ph2) + ph0)),
                        r2 = fabsf(__sinf(13.f * (fx + fy) + ph2));
            const float ridge = fminf(fminf(r0, r1), r2);        // ~0 on the curves
            fg = ridge < 0.12f;
            img = fminf(1.f, (fg ? 0.65f : 0.15f) + 0.35f * u0);
        }
        const float l1 = (u1 < 0.05f) ? 1.f : 0.f, l0 = (u1 > 0.5f && u1 < 0.55f) ? 1.f : 0.f;
        float s1 = l1, s0 = l0;
        if (vessel) { s1 = (fg && u1 < 0.25f) ? 1.f : 0.f; s0 = (!fg && u1 < 0.06f) ? 1.f : 0.f; }
        x[(b * 3 + 0) * HW + p] = img; x[(b * 3 + 1) * HW + p] = img; x[(b * 3 + 2) * HW + p] = img;
        ys[(b * 2 + 0) * HW + p] = s0; ys[(b * 2 + 1) * HW + p] = s1;
        real[(b * 2 + 0) * HW + p] = fg ? 0.f : 1.f; real[(b * 2 + 1) * HW + p] = fg ? 1.f : 0.f;
    }
}
extern "C" int octa_synth_octa(int64_t seed, int B, int H, int W, int vessel, float* x, float* ys, float* real, octa_stream_t stream) {
    OCTA_REQUIRE(x && ys && real && B > 0 && H > 0 && W > 0, "octa_synth_octa: bad arguments");
    synth_octa_kernel<<<x_blocks((int64_t)B * H * W, 8192), 256, 0, (hipStream_t)stream>>>((unsigned long long)seed, B, H, W, vessel, x, ys, real);
    OCTA_CHECK_LAUNCH("synth_octa");
    return OCTA_OK;
}
// levels 1 .. n-1 of the discriminator's real pyramid: nearest down-sampling by 2^l of a dense (B,C,H,W) map, all in ONE launch
struct PyrOut { float* p[8]; int n; };
__global__ __launch_bounds__(256) void mask_pyramid_kernel(const float* __restrict__ src, PyrOut o, int64_t BC, int H, int W) {
    for (int l = 1; l < o.n; ++l) {
        const int hs = H >> l, wsz = W >> l;
        const int64_t total = BC * hs * wsz;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
            const int w = (int)(i % wsz), h = (int)((i / wsz) % hs);
            const int64_t bc = i / ((int64_t)wsz * hs);
            o.p[l][i] = src[(bc * H + ((int64_t)h << l)) * W + ((int64_t)w << l)];
        }
    }
}
extern "C" int octa_mask_pyramid(const float* src, float* const* levels_host, int n_levels, int64_t BC, int H, int W, octa_stream_t stream) {
    OCTA_REQUIRE(src && levels_host && n_levels >= 2 && n_levels <= 8, "octa_mask_pyramid: bad arguments");
    PyrOut o;
    o.n = n_levels;
    for (int l = 0; l < n_levels; ++l) {
        OCTA_REQUIRE(l == 0 || levels_host[l], "octa_mask_pyramid: null level %d", l);
        OCTA_REQUIRE(((H >> l) << l) == H && ((W >> l) << l) == W, "octa_mask_pyramid: %dx%d is not divisible by 2^%d", H, W, l);
        o.p[l] = levels_host[l];
    }
    mask_pyramid_kernel<<<x_blocks(BC * (H >> 1) * (W >> 1)), 256, 0, (hipStream_t)stream>>>(src, o, BC, H, W);
    OCTA_CHECK_LAUNCH("mask_pyramid");
    return OCTA_OK;
}

// ------------------------------------------------------------------------------------------ step tick
// The last kernel of a training step bumps a device counter and publishes it to a word of pinned HOST memory (system-scope
// store).  The host paces hipGraph replays by reading that word: event queries proved unusable for this (an event behind a
// few queued graph launches was reported 70-100 ms after it had actually completed), an ordinary load from coherent pinned
// memory has no such latency and needs no HIP call.
__global__ void host_tick_kernel(int* __restrict__ dev_counter, int* host_flag) {
    const int v = *dev_counter + 1;
    *dev_counter = v;
    __hip_atomic_store(host_flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
extern "C" int octa_host_tick(int* dev_counter, int* host_flag, octa_stream_t stream) {
    OCTA_REQUIRE(dev_counter && host_flag, "octa_host_tick: null pointer");
    host_tick_kernel<<<1, 1, 0, (hipStream_t)stream>>>(dev_counter, host_flag);
    OCTA_CHECK_LAUNCH("host_tick");
    return OCTA_OK;
}

// Slot (*dev_counter % slots) of a ring in pinned HOST memory -> device buffer, as a kernel.  The per-step host updates
// (discriminator noise drawn on the CPU, Adam bias corrections) reach the device through this instead of hipMemcpyAsync: a
// pinned H2D copy queued between hipGraph launches made the device wait for the host runtime (replayed steps stalled for
// ~3 step times whenever the host was not inside a HIP call), a kernel that reads host memory has no such dependency.
__global__ __launch_bounds__(256) void ring_fetch_kernel(const unsigned* __restrict__ ring, int64_t slot_words, int slots,
                                                         const int* __restrict__ dev_counter, unsigned* __restrict__ dst) {
    const int slot = (int)((unsigned)(*dev_counter) % (unsigned)slots);
    const unsigned* src = ring + (int64_t)slot * slot_words;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < slot_words; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}
extern "C" int octa_ring_fetch(const void* ring_host, int64_t slot_bytes, int slots, const int* dev_counter, void* dst, octa_stream_t stream) {
    OCTA_REQUIRE(ring_host && dev_counter && dst && slots > 0 && slot_bytes > 0 && slot_bytes % 4 == 0, "octa_ring_fetch: bad arguments (slot_bytes %% 4)");
    const int64_t words = slot_bytes / 4;
    int blocks = (int)((words + 255) / 256);
    if (blocks > 512) blocks = 512;
    ring_fetch_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>((const unsigned*)ring_host, words, slots, dev_counter, (unsigned*)dst);
    OCTA_CHECK_LAUNCH("ring_fetch");
    return OCTA_OK;
}


// ------------------------------------------------------------------------------------------ comm-stream stand-in (tools/comm_pressure.py)
__global__ __launch_bounds__(256) void probe_stream_load_kernel(float4* __restrict__ buf, int64_t n4, int reps) {
    for (int rp = 0; rp < reps; ++rp)
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
            float4 v = buf[i];
            v.x += 0.f; v.y += 0.f; v.z += 0.f; v.w += 0.f;
            asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));      // (keeps the round trip from being optimised away)
            buf[i] = v;
        }
}
extern "C" int octa_probe_stream_load(float* buf, int64_t bytes, int nblocks, int reps, octa_stream_t stream) {
    OCTA_REQUIRE(buf && bytes >= 16 && ((uintptr_t)buf & 15) == 0 && nblocks >= 1 && nblocks <= 4096 && reps >= 1, "octa_probe_stream_load: bad arguments");
    probe_stream_load_kernel<<<nblocks, 256, 0, (hipStream_t)stream>>>((float4*)buf, bytes / 16, reps);
    OCTA_CHECK_LAUNCH("probe_stream_load");
    return OCTA_OK;
}
