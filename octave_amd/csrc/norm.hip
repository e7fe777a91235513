// Training-mode BatchNorm2d on NHWC activations: statistics, apply (+residual, +ReLU), backward.
// HBM-bound: every kernel moves 16-byte chunks, consecutive lanes on consecutive chunks.
#include "common.hpp"
#include <algorithm>
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
    // slabs of the reduction passes: 1024 until round 5; 640 measured -0.1 ms per step in situ (tools/ab_envs.sh: fewer, longer workgroups
    // amortise a workgroup's fixed cost -- coefficient loads, block reduction, partial store); <= 1024: the partial buffer
    static const int nby_max = getenv("OCTA_BN_NBY") ? std::max(16, std::min(1024, atoi(getenv("OCTA_BN_NBY")))) : 640;
    int64_t rpb = cdiv64(rows, nby_max);
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
                                                        float* __restrict__ partial, int rev) {
    constexpr int EPC = DT<T>::EPC;
    extern __shared__ float red[];   // [RY][TX*EPC][2 or 3]
    const int RY = 256 / TX;
    const int cx = threadIdx.x % TX, ry = threadIdx.x / TX;
    const int col = blockIdx.x * TX + cx;
    const int cpr = C / EPC;
    // row slab of this workgroup.  REVERSED (octa_tuning_set(7, .), default on): the workgroups are dispatched in blockIdx order, so
    // slab gridDim.y - 1 runs first and the pass walks the tensor END FIRST -- the end is what its producer (a conv's epilogue, the
    // upstream gradient kernel) wrote last and what the 256 MB memory-side cache still holds when the tensor is larger than it; the
    // apply pass that follows walks forward again and starts on the lines this pass read last
    const int slab = rev ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y;
    const int64_t r0 = (int64_t)slab * rpb, r1 = min(rows, r0 + rpb);
    float sa[EPC], sb[EPC], mu[EPC], is[EPC];
    int cnt = 0;
#pragma unroll
    for (int e = 0; e < EPC; ++e) { sa[e] = 0.f; sb[e] = 0.f; mu[e] = 0.f; is[e] = 1.f; }
    if (col < cpr) {
        if (MODE == 1) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) { mu[e] = mean[col * EPC + e]; is[e] = invstd[col * EPC + e]; }
        }
        int64_t rbeg = r0 + ry;
        if (MODE == 0 && rbeg < r1) {
            // shifted sums (shift = this thread's first sample): no E[x^2]-E[x]^2 cancellation.  Four independent row loads in
            // flight per thread: the one-load-per-iteration loop read at 4.0 TB/s where the other streaming kernels reach 5-6
            unpack16<T>(*(const uint4*)(x + rbeg * ldx + xoff + col * EPC), mu);
            const T* __restrict__ xp = x + xoff + col * EPC;
            int64_t r = rbeg;
            // eight row loads in flight per thread first (a slab is only 16-64 row steps long: with four, every step waited for a
            // full memory latency and the pass read 2 TB/s on the 40 MB layers; round 5, tools/bn_reduce_micro.py)
            for (; r + 7 * RY < r1; r += 8 * RY) {
                uint4 q[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) q[u] = *(const uint4*)(xp + (r + (int64_t)u * RY) * ldx);
#pragma unroll
                for (int u = 0; u < 8; u += 2) {
                    float x0[EPC], x1[EPC];
                    unpack16<T>(q[u], x0); unpack16<T>(q[u + 1], x1);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        const float d0 = x0[e] - mu[e], d1 = x1[e] - mu[e];
                        sa[e] += d0 + d1;
                        sb[e] += d0 * d0 + d1 * d1;
                    }
                }
                cnt += 8;
            }
            for (; r + 3 * RY < r1; r += 4 * RY) {
                const uint4 q0 = *(const uint4*)(xp + r * ldx), q1 = *(const uint4*)(xp + (r + RY) * ldx);
                const uint4 q2 = *(const uint4*)(xp + (r + 2 * RY) * ldx), q3 = *(const uint4*)(xp + (r + 3 * RY) * ldx);
                float x0[EPC], x1[EPC], x2[EPC], x3[EPC];
                unpack16<T>(q0, x0); unpack16<T>(q1, x1); unpack16<T>(q2, x2); unpack16<T>(q3, x3);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float d0 = x0[e] - mu[e], d1 = x1[e] - mu[e], d2 = x2[e] - mu[e], d3 = x3[e] - mu[e];
                    sa[e] += (d0 + d1) + (d2 + d3);
                    sb[e] += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                }
                cnt += 4;
            }
            for (; r < r1; r += RY) {
                float xv[EPC];
                unpack16<T>(*(const uint4*)(xp + r * ldx), xv);
#pragma unroll
                for (int e = 0; e < EPC; ++e) { const float d = xv[e] - mu[e]; sa[e] += d; sb[e] += d * d; }
                ++cnt;
            }
            rbeg = r1;
        }
        if (MODE == 1) {
            // two rows per iteration, their (up to six) loads issued before any arithmetic
            const bool um = relu && rmask, uy = relu && !rmask;
            // four rows (8-12 loads) in flight first: same reason as the statistics pass
            for (; rbeg + 3 * RY < r1; rbeg += 4 * RY) {
                uint4 xq[4], dq[4], yq[4];
                unsigned mq[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t rr = rbeg + (int64_t)u * RY;
                    xq[u] = *(const uint4*)(x + rr * ldx + xoff + col * EPC);
                    dq[u] = *(const uint4*)(dy + rr * lddy + dyoff + col * EPC);
                    mq[u] = 0xffu; yq[u] = make_uint4(0, 0, 0, 0);
                    if (um) mq[u] = rmask[rr * cpr + col];
                    if (uy) yq[u] = *(const uint4*)(y + rr * ldy + yoff + col * EPC);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float xv[EPC], dv[EPC], yv[EPC];
                    unpack16<T>(xq[u], xv); unpack16<T>(dq[u], dv);
                    if (uy) unpack16<T>(yq[u], yv);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        const bool k = uy ? (yv[e] > 0.f) : ((mq[u] >> e) & 1u) != 0;
                        const float g = k ? dv[e] : 0.f;
                        sa[e] += g;
                        sb[e] += g * (xv[e] - mu[e]) * is[e];
                    }
                }
            }
            for (; rbeg + RY < r1; rbeg += 2 * RY) {
                const int64_t ra = rbeg, rb2 = rbeg + RY;
                const uint4 xa = *(const uint4*)(x + ra * ldx + xoff + col * EPC), xb = *(const uint4*)(x + rb2 * ldx + xoff + col * EPC);
                const uint4 da = *(const uint4*)(dy + ra * lddy + dyoff + col * EPC), db = *(const uint4*)(dy + rb2 * lddy + dyoff + col * EPC);
                unsigned ma = 0xffu, mb = 0xffu;
                uint4 ya = make_uint4(0, 0, 0, 0), yb = ya;
                if (um) { ma = rmask[ra * cpr + col]; mb = rmask[rb2 * cpr + col]; }
                if (uy) { ya = *(const uint4*)(y + ra * ldy + yoff + col * EPC); yb = *(const uint4*)(y + rb2 * ldy + yoff + col * EPC); }
                float x0[EPC], x1[EPC], d0[EPC], d1[EPC], y0[EPC], y1[EPC];
                unpack16<T>(xa, x0); unpack16<T>(xb, x1); unpack16<T>(da, d0); unpack16<T>(db, d1);
                if (uy) { unpack16<T>(ya, y0); unpack16<T>(yb, y1); }
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const bool k0 = uy ? (y0[e] > 0.f) : ((ma >> e) & 1u) != 0, k1 = uy ? (y1[e] > 0.f) : ((mb >> e) & 1u) != 0;
                    const float g0 = k0 ? d0[e] : 0.f, g1 = k1 ? d1[e] : 0.f;
                    sa[e] += g0 + g1;
                    sb[e] += (g0 * (x0[e] - mu[e]) + g1 * (x1[e] - mu[e])) * is[e];
                }
            }
        }
        for (int64_t r = rbeg; r < r1; r += RY) {
            float xv[EPC];
            unpack16<T>(*(const uint4*)(x + r * ldx + xoff + col * EPC), xv);
            if (MODE == 0) {
                // (unreachable: the statistics pass ran above)
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
            // merge of the RY thread partials (n_t, mean_t, M2_t) around a common reference (row group 0's mean, which always has
            // samples): S1 = sum n_t (mean_t - ref), S2 = sum M2_t + n_t (mean_t - ref)^2; mean = ref + S1 / n, M2 = S2 - S1^2 / n --
            // plain sums and ONE division (the pairwise Chan merge this replaces divided twice per row group in double precision, a
            // serial chain of up to 64 x 2 divisions at the end of every workgroup); ref is within a few sigma / sqrt(n_t) of the mean,
            // so nothing cancels
            const double ref = (double)red[(size_t)ch * 3];
            double n = 0.0, s1 = 0.0, s2 = 0.0;
            for (int yy = 0; yy < RY; ++yy) {
                const float* pr = red + ((size_t)yy * TX * EPC + ch) * 3;
                const double nb = (double)pr[2], d = (double)pr[0] - ref;
                n += nb;
                s1 += nb * d;
                s2 += (double)pr[1] + nb * d * d;
            }
            const double m1 = n > 0.0 ? s1 / n : 0.0;
            partial[((size_t)slab * 2 + 0) * C + c] = (float)(ref + m1);
            partial[((size_t)slab * 2 + 1) * C + c] = (float)(s2 - s1 * m1);
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
        partial[((size_t)slab * 2 + 0) * C + c] = a;
        partial[((size_t)slab * 2 + 1) * C + c] = b;
    }
}

// ----------------------------------------------------------------------------------------------------------------------------
// Small tensors (up to ~16 rows per thread with 32 row slabs: the 13x13 .. 26x26 stages): the three launches of a BatchNorm pass are latency, not bytes.  Two launches instead:
//   bn_small_reduce : a workgroup owns TX chunk columns (8 -> 64 channels of a 16-bit tensor: one 128-byte line per row) x a slab
//                     of rows and writes ONE partial per channel; at most 32 row slabs, so a column block has <= 32 partials;
//   bn_small_apply / bn_small_bwd_apply : same column-block geometry; the workgroup's prologue merges the <= 32 partials of ITS
//                     channels (a few KB, 256 threads) into LDS coefficients, then streams its rows.  The workgroups of row
//                     slab 0 also write what the separate finalize launch used to (mean / invstd / running statistics,
//                     resp. dgamma / dbeta).
// No finalize launch, no atomics, no cross-workgroup ordering inside a kernel.  (A single-launch reduce + finalize with a
// last-arriver ticket was measured first: the device-scope fences it needs cost more than the launch they save.)
struct BnSmallGeo { int TX, gridx, nby, rpb; };
static bool bn_small_geometry(int64_t rows, int C, int epc, BnSmallGeo& g) {
    static int64_t max_bytes = -1;
    static int max_rpt = -1, max_nby = -1;
    if (max_nby < 0) { const char* e = getenv("OCTA_BN_SMALL_MAX_NBY"); max_nby = e ? atoi(e) : 32; if (max_nby < 1) max_nby = 1; if (max_nby > 1024) max_nby = 1024; }
    if (max_bytes < 0) { const char* e = getenv("OCTA_BN_SMALL_MAX_MB"); max_bytes = (int64_t)(e ? atoi(e) : 96) << 20; }
    if (max_rpt < 0) { const char* e = getenv("OCTA_BN_SMALL_MAX_RPT"); max_rpt = e ? atoi(e) : 16; }
    const int cpr = C / epc;
    if (cpr % 4 != 0 || rows * C * (int64_t)(epc == 4 ? 4 : 2) > max_bytes) return false;
    g.TX = (cpr % 8 == 0 && cpr >= 64) ? 8 : 4;
    g.gridx = cpr / g.TX;
    const int RY = 256 / g.TX;
    int64_t nby = cdiv64(rows, (int64_t)RY * 2);            // at least two rows per thread
    if (nby > max_nby) nby = max_nby;
    if (nby < 1) nby = 1;
    g.rpb = (int)cdiv64(rows, nby);
    g.nby = (int)cdiv64(rows, g.rpb);
    // rows per thread; beyond ~16 the many-slab kernels stream better (in-situ sweep: 8 / 12 / 16 / 32 / 48 rows and 32 / 64 /
    // 128 slabs; 16 x 32 won, 48 rows lost everything again)
    return cdiv64(g.rpb, RY) <= max_rpt;
}

template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_small_reduce_kernel(const T* __restrict__ x, int ldx, int xoff, const T* __restrict__ dy, int lddy,
                                                              int dyoff, const T* __restrict__ y, int ldy, int yoff,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd, int relu,
                                                              const uint8_t* __restrict__ rmask, int64_t rows, int C, int TX, int rpb,
                                                              float* __restrict__ partial) {
    constexpr int EPC = DT<T>::EPC;
    constexpr int NS = MODE == 0 ? 3 : 2;
    __shared__ float red[256 * EPC * NS];              // [RY][CH][NS]
    const int RY = 256 / TX, CH = TX * EPC;
    const int cx = threadIdx.x % TX, ry = threadIdx.x / TX;
    const int col = blockIdx.x * TX + cx;
    const int cpr = C / EPC;
    const int64_t r0 = (int64_t)blockIdx.y * rpb, r1 = min(rows, r0 + rpb);
    float sa[EPC], sb[EPC], mu[EPC], is[EPC];
    int cnt = 0;
#pragma unroll
    for (int e = 0; e < EPC; ++e) { sa[e] = 0.f; sb[e] = 0.f; mu[e] = 0.f; is[e] = 1.f; }
    if (MODE == 1) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) { mu[e] = mean[col * EPC + e]; is[e] = invstd[col * EPC + e]; }
    }
    int64_t rbeg = r0 + ry;
    // (these tensors give a thread <= 16 rows: with one load per iteration the kernel is 16 dependent round trips to memory;
    // four resp. two rows' loads are issued together)
    if (MODE == 0 && rbeg < r1) {
        const T* __restrict__ xp = x + xoff + col * EPC;
        unpack16<T>(*(const uint4*)(xp + rbeg * ldx), mu);
        int64_t r = rbeg;
        for (; r + 3 * RY < r1; r += 4 * RY) {
            const uint4 q0 = *(const uint4*)(xp + r * ldx), q1 = *(const uint4*)(xp + (r + RY) * ldx);
            const uint4 q2 = *(const uint4*)(xp + (r + 2 * RY) * ldx), q3 = *(const uint4*)(xp + (r + 3 * RY) * ldx);
            float x0[EPC], x1[EPC], x2[EPC], x3[EPC];
            unpack16<T>(q0, x0); unpack16<T>(q1, x1); unpack16<T>(q2, x2); unpack16<T>(q3, x3);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float d0 = x0[e] - mu[e], d1 = x1[e] - mu[e], d2 = x2[e] - mu[e], d3 = x3[e] - mu[e];
                sa[e] += (d0 + d1) + (d2 + d3);
                sb[e] += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
            cnt += 4;
        }
        for (; r < r1; r += RY) {
            float xv[EPC];
            unpack16<T>(*(const uint4*)(xp + r * ldx), xv);
#pragma unroll
            for (int e = 0; e < EPC; ++e) { const float d = xv[e] - mu[e]; sa[e] += d; sb[e] += d * d; }
            ++cnt;
        }
        rbeg = r1;
    }
    if (MODE == 1) {
        const bool um = relu && rmask, uy = relu && !rmask;
        for (; rbeg + RY < r1; rbeg += 2 * RY) {
            const int64_t ra = rbeg, rb2 = rbeg + RY;
            const uint4 xa = *(const uint4*)(x + ra * ldx + xoff + col * EPC), xb = *(const uint4*)(x + rb2 * ldx + xoff + col * EPC);
            const uint4 da = *(const uint4*)(dy + ra * lddy + dyoff + col * EPC), db = *(const uint4*)(dy + rb2 * lddy + dyoff + col * EPC);
            unsigned ma = 0xffu, mb = 0xffu;
            uint4 ya = make_uint4(0, 0, 0, 0), yb = ya;
            if (um) { ma = rmask[ra * cpr + col]; mb = rmask[rb2 * cpr + col]; }
            if (uy) { ya = *(const uint4*)(y + ra * ldy + yoff + col * EPC); yb = *(const uint4*)(y + rb2 * ldy + yoff + col * EPC); }
            float x0[EPC], x1[EPC], d0[EPC], d1[EPC], y0[EPC], y1[EPC];
            unpack16<T>(xa, x0); unpack16<T>(xb, x1); unpack16<T>(da, d0); unpack16<T>(db, d1);
            if (uy) { unpack16<T>(ya, y0); unpack16<T>(yb, y1); }
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const bool k0 = uy ? (y0[e] > 0.f) : ((ma >> e) & 1u) != 0, k1 = uy ? (y1[e] > 0.f) : ((mb >> e) & 1u) != 0;
                const float g0 = k0 ? d0[e] : 0.f, g1 = k1 ? d1[e] : 0.f;
                sa[e] += g0 + g1;
                sb[e] += (g0 * (x0[e] - mu[e]) + g1 * (x1[e] - mu[e])) * is[e];
            }
        }
    }
    for (int64_t r = rbeg; r < r1; r += RY) {
        float xv[EPC];
        unpack16<T>(*(const uint4*)(x + r * ldx + xoff + col * EPC), xv);
        if (MODE == 0) {
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
    float* my = red + ((size_t)ry * CH + cx * EPC) * NS;
    if (MODE == 0) {
        const float fn = (float)cnt;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float m1 = cnt ? sa[e] / fn : 0.f;
            my[e * 3] = mu[e] + m1;                           // thread mean
            my[e * 3 + 1] = cnt ? sb[e] - sa[e] * m1 : 0.f;   // thread M2
            my[e * 3 + 2] = fn;
        }
    } else {
#pragma unroll
        for (int e = 0; e < EPC; ++e) { my[e * 2] = sa[e]; my[e * 2 + 1] = sb[e]; }
    }
    __syncthreads();
    if (threadIdx.x >= CH) return;
    const int ch = threadIdx.x, c = blockIdx.x * CH + ch;
    if (MODE == 0) {
        // row lanes re-expressed around row lane 0's mean (it always holds a sample): plain double sums, no divisions
        const double r = (double)red[ch * 3];
        double n = 0.0, s1 = 0.0, s2 = 0.0;
        for (int yy = 0; yy < RY; ++yy) {
            const float* pr = red + ((size_t)yy * CH + ch) * 3;
            const double nb = (double)pr[2], d = (double)pr[0] - r;
            n += nb; s1 += nb * d; s2 += (double)pr[1] + nb * d * d;
        }
        partial[((size_t)blockIdx.y * 2 + 0) * C + c] = (float)(r + s1 / n);
        partial[((size_t)blockIdx.y * 2 + 1) * C + c] = (float)(s2 - s1 * s1 / n);
    } else {
        float a = 0.f, b = 0.f;
        for (int yy = 0; yy < RY; ++yy) { a += red[((size_t)yy * CH + ch) * 2]; b += red[((size_t)yy * CH + ch) * 2 + 1]; }
        partial[((size_t)blockIdx.y * 2 + 0) * C + c] = a;
        partial[((size_t)blockIdx.y * 2 + 1) * C + c] = b;
    }
}

// prologue shared by the two apply kernels: S1[ch], S2[ch] (doubles in LDS) of this column block from its nbp partials
//   MODE 0: S1 = sum n_b (mean_b - r), S2 = sum M2_b + n_b (mean_b - r)^2, r = block 0's mean      MODE 1: plain sums
template <int MODE>
__device__ __forceinline__ void bn_small_merge(const float* __restrict__ partial, int nbp, int rpbp, int64_t rows, int C, int CH, int c0,
                                               double (*fred)[2] /* [256][2] */, double (*S)[2] /* [64][2] */) {
    const int ch = threadIdx.x % CH, part = threadIdx.x / CH, NP = 256 / CH;
    const int c = c0 + ch;
    double s1 = 0.0, s2 = 0.0;
    // the partials of eight slabs in flight at once (and the reference beside them): one slab per trip was a chain of up to eight memory round trips at
    // the head of a 13 us kernel.  Same sums in the same order.
    const float rf = MODE == 0 ? partial[c] : 0.f;
    for (int b0 = part; b0 < nbp; b0 += 8 * NP) {
        float p1[8], p2[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int b = b0 + k * NP, bb = b < nbp ? b : part;
            p1[k] = partial[((size_t)bb * 2) * C + c]; p2[k] = partial[((size_t)bb * 2 + 1) * C + c];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int b = b0 + k * NP;
            if (b >= nbp) continue;
            if (MODE == 0) {
                const int64_t lo = (int64_t)b * rpbp;
                const double nb = (double)((rows - lo) < rpbp ? (rows - lo) : rpbp);
                const double d = (double)p1[k] - (double)rf;
                s1 += nb * d;
                s2 += (double)p2[k] + nb * d * d;
            } else { s1 += (double)p1[k]; s2 += (double)p2[k]; }
        }
    }
    fred[threadIdx.x][0] = s1; fred[threadIdx.x][1] = s2;
    __syncthreads();
    if (threadIdx.x < CH) {
        s1 = 0.0; s2 = 0.0;
        for (int q = 0; q < NP; ++q) { s1 += fred[q * CH + ch][0]; s2 += fred[q * CH + ch][1]; }
        S[ch][0] = s1; S[ch][1] = s2;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_small_apply_kernel(const T* __restrict__ x, int ldx, int xoff, const float* __restrict__ partial, int nbp,
                                                             int rpbp, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const T* __restrict__ res, int ldr, int roff, T* __restrict__ y, int ldy, int yoff,
                                                             int64_t rows, int C, int TX, int rpb, int relu, uint8_t* __restrict__ rmask,
                                                             float eps, float momentum, float* __restrict__ mean_out,
                                                             float* __restrict__ invstd_out, float* __restrict__ rm, float* __restrict__ rv) {
    constexpr int EPC = DT<T>::EPC;
    __shared__ double fred[256][2];
    __shared__ double S[64][2];
    __shared__ float coef[64][3];                      // mean, gamma * invstd, beta
    const int RY = 256 / TX, CH = TX * EPC;
    const int c0 = blockIdx.x * CH;
    const int cx = threadIdx.x % TX, ry = threadIdx.x / TX;
    const int col = blockIdx.x * TX + cx, cpr = C / EPC;
    const int64_t r0 = (int64_t)blockIdx.y * rpb, r1 = min(rows, r0 + rpb);
    // the per-channel operands and the first row pair travel beside the merge of the partials (everything but the partials is cold in the step)
    float gv = 0.f, bv = 0.f, rmv = 0.f, rvv = 0.f;
    if (threadIdx.x < CH) {
        gv = gamma[c0 + threadIdx.x]; bv = beta[c0 + threadIdx.x];
        if (blockIdx.y == 0) { if (rm) rmv = rm[c0 + threadIdx.x]; if (rv) rvv = rv[c0 + threadIdx.x]; }
    }
    const bool pre = r0 + RY < r1;
    uint4 pxa = make_uint4(0, 0, 0, 0), pxc = pxa, psa = pxa, psc = pxa;
    if (pre) {
        const int64_t ra = r0 + ry, rc = r0 + RY + ry;
        const int64_t qa = ra < r1 ? ra : r0, qc = rc < r1 ? rc : r0;
        pxa = *(const uint4*)(x + qa * ldx + xoff + col * EPC); pxc = *(const uint4*)(x + qc * ldx + xoff + col * EPC);
        if (res) { psa = *(const uint4*)(res + qa * ldr + roff + col * EPC); psc = *(const uint4*)(res + qc * ldr + roff + col * EPC); }
    }
    bn_small_merge<0>(partial, nbp, rpbp, rows, C, CH, c0, fred, S);
    if (threadIdx.x < CH) {
        const int ch = threadIdx.x, c = c0 + ch;
        const double n = (double)rows, r = (double)partial[c], s1 = S[ch][0], s2 = S[ch][1];
        const double m = r + s1 / n;
        double var = (s2 - s1 * s1 / n) / n;
        if (var < 0.0) var = 0.0;
        const float isd = (float)(1.0 / sqrt(var + (double)eps));
        coef[ch][0] = (float)m; coef[ch][1] = gv * isd; coef[ch][2] = bv;
        if (blockIdx.y == 0) {
            mean_out[c] = (float)m;
            invstd_out[c] = isd;
            if (rm) rm[c] = (1.f - momentum) * rmv + momentum * (float)m;
            if (rv) rv[c] = (1.f - momentum) * rvv + momentum * (float)(rows > 1 ? var * n / (n - 1.0) : var);
        }
    }
    __syncthreads();
    float mu[EPC], sc[EPC], be[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { mu[e] = coef[cx * EPC + e][0]; sc[e] = coef[cx * EPC + e][1]; be[e] = coef[cx * EPC + e][2]; }
    // block-uniform trip count (the DPP mask combine below needs every lane of a quad in the loop)
    int64_t rb = r0;
    for (; rb + RY < r1; rb += 2 * RY) {                 // two row groups per iteration, their loads issued together
        const int64_t ra = rb + ry, rc = rb + RY + ry;
        const bool la = ra < r1, lc = rc < r1;
        const int64_t qa = la ? ra : r0, qc = lc ? rc : r0;
        uint4 xa = pxa, xc = pxc, sa = psa, sc2 = psc;
        if (rb != r0) {
            xa = *(const uint4*)(x + qa * ldx + xoff + col * EPC); xc = *(const uint4*)(x + qc * ldx + xoff + col * EPC);
            if (res) { sa = *(const uint4*)(res + qa * ldr + roff + col * EPC); sc2 = *(const uint4*)(res + qc * ldr + roff + col * EPC); }
        }
        float va[EPC], vc[EPC], ea[EPC], ec[EPC];
        unpack16<T>(xa, va); unpack16<T>(xc, vc);
        if (res) { unpack16<T>(sa, ea); unpack16<T>(sc2, ec); }
        unsigned mba = 0u, mbc = 0u;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            float oa = (va[e] - mu[e]) * sc[e] + be[e], oc = (vc[e] - mu[e]) * sc[e] + be[e];
            if (res) { oa += ea[e]; oc += ec[e]; }
            if (relu) {
                if (oa > 0.f) mba |= 1u << e; else oa = 0.f;
                if (oc > 0.f) mbc |= 1u << e; else oc = 0.f;
            }
            va[e] = oa; vc[e] = oc;
        }
        if (la) *(uint4*)(y + qa * ldy + yoff + col * EPC) = pack16<T>(va);
        if (lc) *(uint4*)(y + qc * ldy + yoff + col * EPC) = pack16<T>(vc);
        if (rmask) {
            unsigned wa = la ? (mba << (8 * (threadIdx.x & 3))) : 0u, wc = lc ? (mbc << (8 * (threadIdx.x & 3))) : 0u;
            wa |= (unsigned)__builtin_amdgcn_mov_dpp((int)wa, 0xB1, 0xF, 0xF, true);
            wc |= (unsigned)__builtin_amdgcn_mov_dpp((int)wc, 0xB1, 0xF, 0xF, true);
            wa |= (unsigned)__builtin_amdgcn_mov_dpp((int)wa, 0x4E, 0xF, 0xF, true);
            wc |= (unsigned)__builtin_amdgcn_mov_dpp((int)wc, 0x4E, 0xF, 0xF, true);
            if ((threadIdx.x & 3) == 0 && la) *(unsigned*)(rmask + qa * cpr + col) = wa;
            if ((threadIdx.x & 3) == 0 && lc) *(unsigned*)(rmask + qc * cpr + col) = wc;
        }
    }
    for (; rb < r1; rb += RY) {
        const int64_t r = rb + ry;
        const bool live = r < r1;
        const int64_t rr_ = live ? r : r0;
        float v[EPC], rr[EPC];
        unsigned mb = 0u;
        unpack16<T>(*(const uint4*)(x + rr_ * ldx + xoff + col * EPC), v);
        if (res) unpack16<T>(*(const uint4*)(res + rr_ * ldr + roff + col * EPC), rr);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            float o = (v[e] - mu[e]) * sc[e] + be[e];
            if (res) o += rr[e];
            if (relu) { if (o > 0.f) mb |= 1u << e; else o = 0.f; }
            v[e] = o;
        }
        if (live) *(uint4*)(y + rr_ * ldy + yoff + col * EPC) = pack16<T>(v);
        if (rmask) {
            // 4 neighbouring lanes = 4 neighbouring chunk columns of one row (TX % 4 == 0, cpr % 4 == 0): one dword store
            unsigned w = live ? (mb << (8 * (threadIdx.x & 3))) : 0u;
            w |= (unsigned)__builtin_amdgcn_mov_dpp((int)w, 0xB1, 0xF, 0xF, true);
            w |= (unsigned)__builtin_amdgcn_mov_dpp((int)w, 0x4E, 0xF, 0xF, true);
            if ((threadIdx.x & 3) == 0 && live) *(unsigned*)(rmask + rr_ * cpr + col) = w;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_small_bwd_apply_kernel(const T* __restrict__ dy, int lddy, int dyoff, const T* __restrict__ x, int ldx,
                                                                 int xoff, const T* __restrict__ y, int ldy, int yoff,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 const float* __restrict__ gamma, const float* __restrict__ partial, int nbp,
                                                                 const uint8_t* __restrict__ rmask, T* __restrict__ dx, int lddx, int dxoff,
                                                                 T* __restrict__ dres, int lddr, int droff, int64_t rows, int C, int TX, int rpb,
                                                                 int relu, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    constexpr int EPC = DT<T>::EPC;
    __shared__ double fred[256][2];
    __shared__ double S[64][2];
    __shared__ float coef[64][5];                      // mean, invstd, gamma * invstd, sum dy' / N, sum dy' xhat / N
    const int RY = 256 / TX, CH = TX * EPC;
    const int c0 = blockIdx.x * CH;
    const int cx = threadIdx.x % TX, ry = threadIdx.x / TX;
    const int col = blockIdx.x * TX + cx, cpr = C / EPC;
    const int64_t r0 = (int64_t)blockIdx.y * rpb, r1 = min(rows, r0 + rpb);
    int64_t rbeg = r0 + ry;
    const bool um = relu && rmask, uy = relu && !rmask;
    // the per-channel operands and the first row pair travel beside the merge of the partials (see bn_small_apply_kernel)
    float mv = 0.f, iv = 0.f, gv = 0.f, dbv = 0.f, dgv = 0.f;
    if (threadIdx.x < CH) {
        const int c = c0 + threadIdx.x;
        mv = mean[c]; iv = invstd[c]; gv = gamma[c];
        if (blockIdx.y == 0) { if (dbeta) dbv = dbeta[c]; if (dgamma) dgv = dgamma[c]; }
    }
    const bool pre = rbeg + RY < r1;
    uint4 pxa = make_uint4(0, 0, 0, 0), pxb = pxa, pda = pxa, pdb = pxa, pya = pxa, pyb = pxa;
    unsigned pma = 0xffu, pmb = 0xffu;
    if (pre) {
        const int64_t ra = rbeg, rb2 = rbeg + RY;
        pxa = *(const uint4*)(x + ra * ldx + xoff + col * EPC); pxb = *(const uint4*)(x + rb2 * ldx + xoff + col * EPC);
        pda = *(const uint4*)(dy + ra * lddy + dyoff + col * EPC); pdb = *(const uint4*)(dy + rb2 * lddy + dyoff + col * EPC);
        if (um) { pma = rmask[ra * cpr + col]; pmb = rmask[rb2 * cpr + col]; }
        if (uy) { pya = *(const uint4*)(y + ra * ldy + yoff + col * EPC); pyb = *(const uint4*)(y + rb2 * ldy + yoff + col * EPC); }
    }
    bn_small_merge<1>(partial, nbp, 0, rows, C, CH, c0, fred, S);
    if (threadIdx.x < CH) {
        const int ch = threadIdx.x, c = c0 + ch;
        const double n = (double)rows, s = S[ch][0], ss = S[ch][1];
        coef[ch][0] = mv; coef[ch][1] = iv; coef[ch][2] = gv * iv;
        coef[ch][3] = (float)(s / n); coef[ch][4] = (float)(ss / n);
        if (blockIdx.y == 0) {
            if (dbeta) dbeta[c] = dbv + (float)s;
            if (dgamma) dgamma[c] = dgv + (float)ss;
        }
    }
    __syncthreads();
    float mu[EPC], isd[EPC], gi[EPC], f0[EPC], f1[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        const float* k = coef[cx * EPC + e];
        mu[e] = k[0]; isd[e] = k[1]; gi[e] = k[2]; f0[e] = k[3]; f1[e] = k[4];
    }
    {   // two rows per iteration, loads first (see bn_small_reduce_kernel)
        const int64_t rfirst = rbeg;
        for (; rbeg + RY < r1; rbeg += 2 * RY) {
            const int64_t ra = rbeg, rb2 = rbeg + RY;
            uint4 xa = pxa, xb = pxb, da = pda, db = pdb;
            if (rbeg != rfirst) {
                xa = *(const uint4*)(x + ra * ldx + xoff + col * EPC); xb = *(const uint4*)(x + rb2 * ldx + xoff + col * EPC);
                da = *(const uint4*)(dy + ra * lddy + dyoff + col * EPC); db = *(const uint4*)(dy + rb2 * lddy + dyoff + col * EPC);
            }
            unsigned ma = pma, mb = pmb;
            uint4 ya = pya, yb = pyb;
            if (rbeg != rfirst) {
                if (um) { ma = rmask[ra * cpr + col]; mb = rmask[rb2 * cpr + col]; }
                if (uy) { ya = *(const uint4*)(y + ra * ldy + yoff + col * EPC); yb = *(const uint4*)(y + rb2 * ldy + yoff + col * EPC); }
            }
            float x0[EPC], x1[EPC], d0[EPC], d1[EPC], y0[EPC], y1[EPC], o0[EPC], o1[EPC];
            unpack16<T>(xa, x0); unpack16<T>(xb, x1); unpack16<T>(da, d0); unpack16<T>(db, d1);
            if (uy) { unpack16<T>(ya, y0); unpack16<T>(yb, y1); }
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const bool k0 = uy ? (y0[e] > 0.f) : ((ma >> e) & 1u) != 0, k1 = uy ? (y1[e] > 0.f) : ((mb >> e) & 1u) != 0;
                d0[e] = k0 ? d0[e] : 0.f; d1[e] = k1 ? d1[e] : 0.f;
                o0[e] = gi[e] * (d0[e] - f0[e] - (x0[e] - mu[e]) * isd[e] * f1[e]);
                o1[e] = gi[e] * (d1[e] - f0[e] - (x1[e] - mu[e]) * isd[e] * f1[e]);
            }
            *(uint4*)(dx + ra * lddx + dxoff + col * EPC) = pack16<T>(o0);
            *(uint4*)(dx + rb2 * lddx + dxoff + col * EPC) = pack16<T>(o1);
            if (dres) { *(uint4*)(dres + ra * lddr + droff + col * EPC) = pack16<T>(d0); *(uint4*)(dres + rb2 * lddr + droff + col * EPC) = pack16<T>(d1); }
        }
    }
    for (int64_t r = rbeg; r < r1; r += RY) {
        float dv[EPC], xv[EPC], yv[EPC], o[EPC];
        unpack16<T>(*(const uint4*)(dy + r * lddy + dyoff + col * EPC), dv);
        unpack16<T>(*(const uint4*)(x + r * ldx + xoff + col * EPC), xv);
        if (relu && rmask) {
            const unsigned mb = rmask[r * cpr + col];
#pragma unroll
            for (int e = 0; e < EPC; ++e) yv[e] = (mb >> e) & 1u ? 1.f : 0.f;
        } else if (relu) unpack16<T>(*(const uint4*)(y + r * ldy + yoff + col * EPC), yv);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float d = (relu && !(yv[e] > 0.f)) ? 0.f : dv[e];
            dv[e] = d;
            const float xh = (xv[e] - mu[e]) * isd[e];
            o[e] = gi[e] * (d - f0[e] - xh * f1[e]);
        }
        *(uint4*)(dx + r * lddx + dxoff + col * EPC) = pack16<T>(o);
        if (dres) *(uint4*)(dres + r * lddr + droff + col * EPC) = pack16<T>(dv);
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
        bn_reduce_kernel<float, 0><<<grid, 256, sh, st>>>((const float*)x, ld, off, nullptr, 0, 0, nullptr, 0, 0, nullptr, nullptr, 0, nullptr, rows, C, cm.TX, rpb, ws, octa_rev_walk());
    else if (dtype == OCTA_BF16)
        bn_reduce_kernel<bf16_t, 0><<<grid, 256, sh, st>>>((const bf16_t*)x, ld, off, nullptr, 0, 0, nullptr, 0, 0, nullptr, nullptr, 0, nullptr, rows, C, cm.TX, rpb, ws, octa_rev_walk());
    else
        bn_reduce_kernel<f16_t, 0><<<grid, 256, sh, st>>>((const f16_t*)x, ld, off, nullptr, 0, 0, nullptr, 0, 0, nullptr, nullptr, 0, nullptr, rows, C, cm.TX, rpb, ws, octa_rev_walk());
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
    static const int per_cu = getenv("OCTA_EW_WG_PER_CU") ? std::max(1, atoi(getenv("OCTA_EW_WG_PER_CU"))) : 10;
    const int64_t cap = 256 * per_cu;                   // ~10 resident workgroups per CU
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

// Training-mode forward in one call: statistics + apply.  Small tensors take the two-launch path above, the rest
// octa_bn_stats + octa_bn_apply (three launches).
extern "C" int octa_bn_train_fwd(const void* x, int ldx, int xoff, const float* gamma, const float* beta, const void* residual, int ldr,
                                 int roff, void* y, int ldy, int yoff, int64_t rows, int C, int dtype, float eps, float momentum, int relu,
                                 float* mean, float* invstd, float* running_mean, float* running_var, uint8_t* relu_mask, float* ws,
                                 octa_stream_t stream) {
    OCTA_REQUIRE(x && y && gamma && beta && mean && invstd && ws, "octa_bn_train_fwd: null pointer");
    OCTA_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0 && xoff % 8 == 0 && ldy % 8 == 0 && yoff % 8 == 0 &&
                     (!residual || (ldr % 8 == 0 && roff % 8 == 0)),
                 "octa_bn_train_fwd: C/ld/off must be multiples of 8");
    OCTA_REQUIRE(OCTA_DTYPE_OK(dtype), "octa_bn_train_fwd: bad dtype");
    hipStream_t st = (hipStream_t)stream;
    const int epc = dtype == OCTA_F32 ? 4 : 8;
    BnSmallGeo sg;
    if (!bn_small_geometry(rows, C, epc, sg)) {
        const int rc = octa_bn_stats(x, rows, C, ldx, xoff, dtype, eps, momentum, mean, invstd, running_mean, running_var, ws, stream);
        if (rc != OCTA_OK) return rc;
        return octa_bn_apply(x, ldx, xoff, mean, invstd, gamma, beta, residual, ldr, roff, y, ldy, yoff, rows, C, dtype, relu, relu_mask, stream);
    }
    dim3 g(sg.gridx, sg.nby);
#define OCTA_BN_SMALL_FWD(TT)                                                                                                                    \
    bn_small_reduce_kernel<TT, 0><<<g, 256, 0, st>>>((const TT*)x, ldx, xoff, nullptr, 0, 0, nullptr, 0, 0, nullptr, nullptr, 0, nullptr, rows, C, \
                                                     sg.TX, sg.rpb, ws);                                                                          \
    OCTA_CHECK_LAUNCH("bn_small_reduce(stats)");                                                                                                  \
    bn_small_apply_kernel<TT><<<g, 256, 0, st>>>((const TT*)x, ldx, xoff, ws, sg.nby, sg.rpb, gamma, beta, (const TT*)residual, ldr, roff, (TT*)y,  \
                                                 ldy, yoff, rows, C, sg.TX, sg.rpb, relu, relu_mask, eps, momentum, mean, invstd, running_mean,    \
                                                 running_var)
    if (dtype == OCTA_F32) { OCTA_BN_SMALL_FWD(float); }
    else if (dtype == OCTA_BF16) { OCTA_BN_SMALL_FWD(bf16_t); }
    else { OCTA_BN_SMALL_FWD(f16_t); }
#undef OCTA_BN_SMALL_FWD
    OCTA_CHECK_LAUNCH("bn_small_apply");
    return OCTA_OK;
}

// Training-mode forward when the producing conv already summed its output (octa_conv2d_fwd_stats): sums[r][0][c] = sum (x - shift),
// sums[r][1][c] = sum (x - shift)^2 over replica r's share of the rows, shift = the running mean BEFORE this update (NULL: 0).
// One thread per channel merges the replicas in double; then the ordinary apply launch.  Two launches instead of three (two for
// the small tensors either way), and the statistics pass over the activation is gone.
__global__ __launch_bounds__(256) void bn_sums_finalize_kernel(const float* __restrict__ sums, int R, int C, int64_t rows, float eps, float momentum,
                                                              float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ rm,
                                                              float* __restrict__ rv) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int r = 0; r < R; ++r) { s1 += (double)sums[((size_t)r * 2) * C + c]; s2 += (double)sums[((size_t)r * 2 + 1) * C + c]; }
    const double n = (double)rows, sh = rm ? (double)rm[c] : 0.0;
    const double m = sh + s1 / n;
    double var = (s2 - s1 * s1 / n) / n;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rm) rm[c] = (1.f - momentum) * rm[c] + momentum * (float)m;
    if (rv) rv[c] = (1.f - momentum) * rv[c] + momentum * (float)(rows > 1 ? var * n / (n - 1.0) : var);
}
extern "C" int octa_bn_train_fwd_sums(const void* x, int ldx, int xoff, const float* sums, int replicas, const float* gamma, const float* beta,
                                      const void* residual, int ldr, int roff, void* y, int ldy, int yoff, int64_t rows, int C, int dtype,
                                      float eps, float momentum, int relu, float* mean, float* invstd, float* running_mean, float* running_var,
                                      uint8_t* relu_mask, octa_stream_t stream) {
    OCTA_REQUIRE(x && y && sums && gamma && beta && mean && invstd && replicas >= 1, "octa_bn_train_fwd_sums: null pointer");
    bn_sums_finalize_kernel<<<cdiv(C, 256), 256, 0, (hipStream_t)stream>>>(sums, replicas, C, rows, eps, momentum, mean, invstd, running_mean, running_var);
    OCTA_CHECK_LAUNCH("bn_sums_finalize");
    return octa_bn_apply(x, ldx, xoff, mean, invstd, gamma, beta, residual, ldr, roff, y, ldy, yoff, rows, C, dtype, relu, relu_mask, stream);
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

// (used by splat_aag.hip: split attention with bn0 recomputed on the fly)
int octa_bn_bwd_finalize_launch(const float* partial, int nby, int C, int64_t rows, float* fin, float* dgamma, float* dbeta, hipStream_t st) {
    bn_bwd_finalize_kernel<<<C, 256, 0, st>>>(partial, nby, C, rows, fin, dgamma, dbeta);
    OCTA_CHECK_LAUNCH("bn_bwd_finalize");
    return OCTA_OK;
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
    BnSmallGeo sg;
    if (bn_small_geometry(rows, C, epc, sg)) {
        dim3 g(sg.gridx, sg.nby);
#define OCTA_BN_SMALL_BWD(TT)                                                                                                                   \
        bn_small_reduce_kernel<TT, 1><<<g, 256, 0, st>>>((const TT*)x, ldx, xoff, (const TT*)dy, lddy, dyoff, (const TT*)y, ldy, yoff, mean, invstd, \
                                                         relu, relu_mask, rows, C, sg.TX, sg.rpb, ws);                                           \
        OCTA_CHECK_LAUNCH("bn_small_reduce(bwd)");                                                                                               \
        bn_small_bwd_apply_kernel<TT><<<g, 256, 0, st>>>((const TT*)dy, lddy, dyoff, (const TT*)x, ldx, xoff, (const TT*)y, ldy, yoff, mean, invstd, \
                                                         gamma, ws, sg.nby, relu_mask, (TT*)dx, lddx, dxoff, (TT*)dres, lddr, droff, rows, C,    \
                                                         sg.TX, sg.rpb, relu, dgamma, dbeta)
        if (dtype == OCTA_F32) { OCTA_BN_SMALL_BWD(float); }
        else if (dtype == OCTA_BF16) { OCTA_BN_SMALL_BWD(bf16_t); }
        else { OCTA_BN_SMALL_BWD(f16_t); }
#undef OCTA_BN_SMALL_BWD
        OCTA_CHECK_LAUNCH("bn_small_bwd_apply");
        return OCTA_OK;
    }
    {
    if (dtype == OCTA_F32)
        bn_reduce_kernel<float, 1><<<grid, 256, sh, st>>>((const float*)x, ldx, xoff, (const float*)dy, lddy, dyoff, (const float*)y, ldy, yoff, mean, invstd, relu, relu_mask, rows, C, cm.TX, rpb, ws, octa_rev_walk());
    else if (dtype == OCTA_BF16)
        bn_reduce_kernel<bf16_t, 1><<<grid, 256, sh, st>>>((const bf16_t*)x, ldx, xoff, (const bf16_t*)dy, lddy, dyoff, (const bf16_t*)y, ldy, yoff, mean, invstd, relu, relu_mask, rows, C, cm.TX, rpb, ws, octa_rev_walk());
    else
        bn_reduce_kernel<f16_t, 1><<<grid, 256, sh, st>>>((const f16_t*)x, ldx, xoff, (const f16_t*)dy, lddy, dyoff, (const f16_t*)y, ldy, yoff, mean, invstd, relu, relu_mask, rows, C, cm.TX, rpb, ws, octa_rev_walk());
    OCTA_CHECK_LAUNCH("bn_reduce(bwd)");
    bn_bwd_finalize_kernel<<<C, 256, 0, st>>>(ws, nby, C, rows, fin, dgamma, dbeta);
    OCTA_CHECK_LAUNCH("bn_bwd_finalize");
    }
    if (dtype == OCTA_F32)
        bn_bwd_apply_kernel<float><<<ew_blocks_aligned(rows * (C / 4), C / 4), 256, 0, st>>>((const float*)dy, lddy, dyoff, (const float*)x, ldx, xoff, (const float*)y, ldy, yoff, mean, invstd, gamma, fin, relu_mask, (float*)dx, lddx, dxoff, (float*)dres, lddr, droff, rows, C, relu);
    else if (dtype == OCTA_BF16)
        bn_bwd_apply_kernel<bf16_t><<<ew_blocks_aligned(rows * (C / 8), C / 8), 256, 0, st>>>((const bf16_t*)dy, lddy, dyoff, (const bf16_t*)x, ldx, xoff, (const bf16_t*)y, ldy, yoff, mean, invstd, gamma, fin, relu_mask, (bf16_t*)dx, lddx, dxoff, (bf16_t*)dres, lddr, droff, rows, C, relu);
    else
        bn_bwd_apply_kernel<f16_t><<<ew_blocks_aligned(rows * (C / 8), C / 8), 256, 0, st>>>((const f16_t*)dy, lddy, dyoff, (const f16_t*)x, ldx, xoff, (const f16_t*)y, ldy, yoff, mean, invstd, gamma, fin, relu_mask, (f16_t*)dx, lddx, dxoff, (f16_t*)dres, lddr, droff, rows, C, relu);
    OCTA_CHECK_LAUNCH("bn_bwd_apply");
    return OCTA_OK;
}
