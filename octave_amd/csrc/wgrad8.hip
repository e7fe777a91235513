// Batched weight-gradient engine (gfx950, bf16/f16): dW[n][k] += sum_m dy[m][n] * im2col(x)[m][k] for MANY layers per launch.
//
// Why batched: the weight gradients are off the critical path of the backward pass (nothing downstream reads them
// before the optimiser), so the host queues them per stage and flushes the queue as ONE launch.  A launch then holds
// hundreds of output tiles, the split over the pixel axis (and with it the fp32 reduction traffic) shrinks by the
// number of layers batched, and ~60 launches per step disappear.
//
// Kernel shape (one workgroup = 512 threads = 8 waves, one workgroup per CU):
//   output slab 256(N) x 128(K) or 128(N) x 256(K), every wave owns a 64 x 64 sub-tile (16 MFMA 16x16x32 accumulators);
//   the contraction index is the PIXEL: 64 pixel rows per stage, 3-stage LDS ring (3 x 48 KB) filled by LDS-DMA
//   (global_load_lds_dwordx4, 6 wave-instructions per wave and stage) with ONE raw s_barrier per stage and a counted
//   vmcnt that keeps the next stage in flight across it; no register staging, no ds_write.
//   Both operands have the contraction index as their slow memory axis, so the MFMA fragments are read transposed with
//   ds_read_b64_tr_b16.  LDS rows are 256/512 B; the 32-byte pair index of a row is XOR-ed with
//   f(m) = (m & 3) | ((m >> 3) & 1) << 2, which makes the 32-lane halves of a transposed read hit 8 distinct 32-byte
//   slots (conflict-free); the LDS-DMA image is lane-linear, so the same involution is applied to the per-lane SOURCE chunk.
//   Out-of-image taps, the M tail and channel padding are sourced from a 16-byte zero page.
#include "common.hpp"
#include <vector>
#include <algorithm>
#include <functional>
#include <map>
#include <mutex>

typedef __attribute__((ext_vector_type(8))) __bf16 wg_bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 wg_f16x8_t;
typedef __attribute__((ext_vector_type(4))) float wg_f32x4_t;
typedef __attribute__((ext_vector_type(4))) short wg_s16x4_t;

__device__ __attribute__((aligned(16))) unsigned int wg8_zero_page[4] = {0u, 0u, 0u, 0u};

struct WgProb {
    const unsigned short* x; const unsigned short* dy; float* dw; float* dbias;
    int H, W, OH, OW;
    int Cg, CgReal, Ng;
    int KH, KW, stride, pad;
    int ldx, xoff, ldy, yoff;
    int M, Kpad;
    int s_o, s_i, s_h, s_w;
    unsigned magicOW, magicOH;      // floor(2^32 / d) + 1: exact quotient for n * d < 2^32
    int tilesN, tilesK, groups, splitM, mPerSplit;
    int blockStart;
    int nblocks;                    // tilesN * tilesK * groups * splitM (wgrad9x: the XCD-interleaved schedule)
    // partial-store mode (round 4, conv.hip: wgrad_fold_kernel): part != NULL -> M-split sp stores its raw tile to
    // part[sp * part_slice + (g * Ng + n) * Kpad + k] (bias sums behind the groups * Ng * Kpad block) instead of float atomics into dw
    float* part;
    long part_slice;
};
#define WG_MAXP 20
struct WgBatch { WgProb p[WG_MAXP]; int n; };

template <int N> __device__ __forceinline__ void wg_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wg_glds16(const void* gsrc, unsigned lds_base /* wave-uniform */) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_base) : "memory");
}
__device__ __forceinline__ unsigned wg_lds_addr(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) void*)p; }

// transposed 8-byte LDS read with an immediate row offset (inline asm: hipcc does not fold integer LDS address arithmetic into
// the offset field and spent 2 VALU per read on it).  Not counted by hipcc: the caller waits with wg_wait_frags.
typedef __attribute__((ext_vector_type(2))) unsigned wg_u32x2_t;
template <int OFF> __device__ __forceinline__ wg_u32x2_t wg_tr(unsigned addr) {
    wg_u32x2_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int PRB, int QRB, int HS>
__device__ __forceinline__ void wg_load_half(const unsigned (&pa)[4], const unsigned (&qa)[4], wg_u32x2_t (&pf)[4][2], wg_u32x2_t (&qf)[4][2]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { pf[i][0] = wg_tr<HS * 32 * PRB>(pa[i]); pf[i][1] = wg_tr<HS * 32 * PRB + 4 * PRB>(pa[i]); }
#pragma unroll
    for (int i = 0; i < 4; ++i) { qf[i][0] = wg_tr<HS * 32 * QRB>(qa[i]); qf[i][1] = wg_tr<HS * 32 * QRB + 4 * QRB>(qa[i]); }
}
// wait for every outstanding LDS read; naming the 16 destinations pins the wait between the reads and their consumers
__device__ __forceinline__ void wg_wait_frags(wg_u32x2_t (&pf)[4][2], wg_u32x2_t (&qf)[4][2]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(pf[0][0]), "+v"(pf[0][1]), "+v"(pf[1][0]), "+v"(pf[1][1]), "+v"(pf[2][0]), "+v"(pf[2][1]), "+v"(pf[3][0]), "+v"(pf[3][1]),
                   "+v"(qf[0][0]), "+v"(qf[0][1]), "+v"(qf[1][0]), "+v"(qf[1][1]), "+v"(qf[2][0]), "+v"(qf[2][1]), "+v"(qf[3][0]), "+v"(qf[3][1])
                 :: "memory");
}

template <int F16> struct WgMma;
template <> struct WgMma<0> {
    __device__ static __forceinline__ void run(const uint4& a, const uint4& b, wg_f32x4_t& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wg_bf16x8_t, a), __builtin_bit_cast(wg_bf16x8_t, b), c, 0, 0, 0);
    }
    __device__ static __forceinline__ float cvt(unsigned short v) { return __uint_as_float(((unsigned)v) << 16); }
};
template <> struct WgMma<1> {
    __device__ static __forceinline__ void run(const uint4& a, const uint4& b, wg_f32x4_t& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(wg_f16x8_t, a), __builtin_bit_cast(wg_f16x8_t, b), c, 0, 0, 0);
    }
    __device__ static __forceinline__ float cvt(unsigned short v) { return (float)__builtin_bit_cast(_Float16, v); }
};

template <int WN, int WK, int F16, int STAGGER>
__global__ __launch_bounds__(512) void wgrad8_kernel(const WgBatch batch) {
    constexpr int BN = WN * 64, BK = WK * 64, MT = 64, STAGES = 3;
    constexpr int PRB = BN * 2, QRB = BK * 2;                      // LDS row bytes of the dy / x stage images
    constexpr int PBYTES = MT * PRB, QBYTES = MT * QRB, SBYTES = PBYTES + QBYTES;
    constexpr int P_LPR = PRB / 16, P_RPI = 64 / P_LPR, P_IPW = MT / P_RPI / 8;   // lanes per row, rows per DMA instruction, instr per wave
    constexpr int Q_LPR = QRB / 16, Q_RPI = 64 / Q_LPR, Q_IPW = MT / Q_RPI / 8;
    constexpr int LPT = P_IPW + Q_IPW;                             // DMA instructions per wave and stage (6)
    static_assert(WN * WK == 8 && LPT == 6, "8 waves of 64x64");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[STAGES * SBYTES + BN * 4];
    float* const sBias = (float*)(smem + STAGES * SBYTES);

    // ---- which problem / tile is this workgroup (XCD-contiguous order over the whole launch)
    const int total = gridDim.x, Lb = blockIdx.x;
    const int xcd = Lb & 7, jq = Lb >> 3, qn = total >> 3, rn = total & 7;
    const int Lp = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + jq;
    int pi = 0;
    for (int i = 1; i < batch.n; ++i) if (batch.p[i].blockStart <= Lp) pi = i;
    // every field the loop needs is copied to registers once (the kernel-argument struct is not re-read in the loop)
    const WgProb& Pk = batch.p[pi];
    const int tilesN = Pk.tilesN, tilesK = Pk.tilesK, groups = Pk.groups, mPerSplit = Pk.mPerSplit, Mtot = Pk.M;
    const int Ng = Pk.Ng, Kpad = Pk.Kpad, Cg = Pk.Cg, CgReal = Pk.CgReal, KW = Pk.KW;
    const int H = Pk.H, W = Pk.W, OH = Pk.OH, OW = Pk.OW, stride = Pk.stride, pad = Pk.pad;
    const int ldx = Pk.ldx, ldy = Pk.ldy;
    const unsigned magicOW = Pk.magicOW, magicOH = Pk.magicOH;
    const unsigned short* const xbase = Pk.x;
    const unsigned short* const dybase = Pk.dy;
    int bid = Lp - Pk.blockStart;
    const int nt = bid % tilesN; bid /= tilesN;
    const int kt = bid % tilesK; bid /= tilesK;
    const int g = bid % groups;
    const int sp = bid / groups;
    const int n0 = nt * BN, k0 = kt * BK;
    const int mbeg = sp * mPerSplit;
    const int mend = min(Mtot, mbeg + mPerSplit);
    const int nsteps = (mend - mbeg + MT - 1) / MT;
    if (nsteps <= 0) return;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wn = wave / WK, wk = wave % WK;
    const int r = lane & 15, q = lane >> 4;
    const unsigned long zaddr = (unsigned long)(const void*)wg8_zero_page;

    // ---- DMA roles.  Instruction i of this wave fills LDS rows (i*8 + wave)*RPI + lane/LPR: rows of one lane differ by a
    // multiple of 16, so f(row) -- and with it the data chunk this lane fetches -- is the same for all of them.
    // VALU budget: this loop is VALU-issue bound (two waves share a SIMD's vector pipe with their own MFMA issue), so the
    // per-row address work is kept to 32-bit full-rate instructions: dy rows are one 64-bit add per stage (invalid lanes
    // point at the zero page with a zero step), x rows track (ih, iw, input pixel) incrementally with single-wrap updates.
    const int prow = wave * P_RPI + lane / P_LPR;
    const int ppos = lane % P_LPR;
    const int pf_ = (prow & 3) | (((prow >> 3) & 1) << 2);
    const int pchunk = (((ppos >> 1) ^ pf_) << 1) | (ppos & 1);
    const bool pvalid = (n0 + pchunk * 8) < Ng;
    const unsigned long pstep = pvalid ? (unsigned long)((long)MT * ldy * 2) : 0ul;
    unsigned long pptr[P_IPW];
#pragma unroll
    for (int i = 0; i < P_IPW; ++i)
        pptr[i] = pvalid ? (unsigned long)(dybase + ((long)(mbeg + prow + i * 8 * P_RPI) * ldy + Pk.yoff + g * Ng + n0 + pchunk * 8)) : zaddr;

    const int qrow = wave * Q_RPI + lane / Q_LPR;
    const int qpos = lane % Q_LPR;
    const int qf_ = (qrow & 3) | (((qrow >> 3) & 1) << 2);
    const int qchunk = (((qpos >> 1) ^ qf_) << 1) | (qpos & 1);
    const int kel = k0 + qchunk * 8;
    const bool kvalid = kel < Kpad;
    const int qtap = kel / Cg, qcc = kel - qtap * Cg;
    const int qkh = qtap / KW, qkw = qtap - qkh * KW;
    const int qdh = qkh - pad, qdw = qkw - pad;
    // byte address of channel chunk qcc of input pixel 0 for this lane's group
    const unsigned long qbase = (unsigned long)(xbase + (Pk.xoff + g * CgReal + qcc));
    const bool plain = (Pk.KH == 1 && KW == 1 && pad == 0 && stride == 1);   // 1x1: input pixel == output pixel
    const int ldx2 = ldx * 2;
    // incremental pixel state of the Q_IPW rows this lane fetches (general path)
    const int dq = MT / OW, dr = MT - dq * OW;               // a stage advances the output pixel by dq rows + dr columns
    const int sdr = stride * dr, sdq = stride * dq, OWs = OW * stride, OHs = OH * stride;
    const int thrW = OWs + qdw, thrH = OHs + qdh;
    const int dpix = sdq * W + sdr, cW = stride * W - OWs, cH = H * W - OHs * W;
    int qih[Q_IPW], qiw[Q_IPW], qpix[Q_IPW];
#pragma unroll
    for (int i = 0; i < Q_IPW; ++i) {
        const int m = mbeg + qrow + i * 8 * Q_RPI;
        const int ow = m % OW, tq = m / OW, oh = tq % OH, b = tq / OH;
        qih[i] = oh * stride + qdh; qiw[i] = ow * stride + qdw;
        qpix[i] = plain ? m : (b * H + qih[i]) * W + qiw[i];
    }

    const unsigned sbase = wg_lds_addr(smem);
    auto issue = [&](int stage, int mcur) {
        const unsigned pb = sbase + (unsigned)(stage * SBYTES), qb = pb + PBYTES;
        const bool full = (mcur + MT) <= mend;      // uniform: no per-row tail test on full stages
#pragma unroll
        for (int i = 0; i < P_IPW; ++i) {
            unsigned long src = pptr[i];
            if (!full) src = ((mcur + prow + i * 8 * P_RPI) < mend) ? src : zaddr;
            wg_glds16((const void*)src, __builtin_amdgcn_readfirstlane(pb + (unsigned)((i * 8 + wave) * 1024)));
            pptr[i] += pstep;
        }
#pragma unroll
        for (int i = 0; i < Q_IPW; ++i) {
            bool ok = kvalid;
            if (!full) ok = ok & ((mcur + qrow + i * 8 * Q_RPI) < mend);
            if (!plain) ok = ok & ((unsigned)qih[i] < (unsigned)H) & ((unsigned)qiw[i] < (unsigned)W);
            const unsigned off = (unsigned)__mul24(qpix[i], ldx2);          // pixel index < 2^23, ldx2 < 2^17: exact in 24 x 24 bits
            const unsigned long a = qbase + (unsigned long)off;
            const unsigned long src = ok ? a : zaddr;
            wg_glds16((const void*)src, __builtin_amdgcn_readfirstlane(qb + (unsigned)((i * 8 + wave) * 1024)));
            if (plain) qpix[i] += MT;
            else {
                qiw[i] += sdr;
                const bool c1 = qiw[i] >= thrW;
                qiw[i] -= c1 ? OWs : 0;
                qih[i] += sdq + (c1 ? stride : 0);
                const bool c2 = qih[i] >= thrH;
                qih[i] -= c2 ? OHs : 0;
                qpix[i] += dpix + (c1 ? cW : 0) + (c2 ? cH : 0);
            }
        }
    };

    wg_f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (wg_f32x4_t){0.f, 0.f, 0.f, 0.f};

    // ---- transposed fragment addressing: lane (r, q) reads 8 bytes at rows mrow and mrow + 4, f is a lane constant.
    // LDS byte addresses of stage 0 are kept in 8 registers; a stage adds one scalar, rows are immediate offsets.
    const int mrow = 8 * q + (r >> 2);
    const int fr = (mrow & 3) | (((mrow >> 3) & 1) << 2);
    const int cb = (r & 3) * 8;                  // byte inside the 32-byte pair
    unsigned pad0[4], qad0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        pad0[i] = sbase + (unsigned)(mrow * PRB + ((((wn * 4 + i)) ^ fr) << 5) + cb);
        qad0[i] = sbase + (unsigned)(PBYTES + mrow * QRB + ((((wk * 4 + i)) ^ fr) << 5) + cb);
    }

    // ---- fused bias gradient (k-tile 0 only): thread = (chunk position, 4 or 2 rows with equal f) of the dy stage image
    float* const dbias = Pk.dbias;
    const bool do_bias = (dbias != nullptr) && (kt == 0);
    constexpr int B_RG = 512 / P_LPR, B_NJ = MT / B_RG;
    const int brow = t / P_LPR, bpos = t % P_LPR;
    const int bf_ = (brow & 3) | (((brow >> 3) & 1) << 2);
    const int bchunk = (((bpos >> 1) ^ bf_) << 1) | (bpos & 1);
    float bsum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[e] = 0.f;
    if (do_bias && t < BN) sBias[t] = 0.f;

    static_assert(STAGES == 3, "ring depth");
    const bool late = (wave >= 4) && (STAGGER != 0);
    issue(0, mbeg);
    if (nsteps > 1) issue(1, mbeg + MT);
    for (int it = 0; it < nsteps; ++it) {
        if (it + 1 < nsteps) wg_wait_vmcnt<LPT>(); else wg_wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // every ds_read of the previous stage has returned (WAR on the ring)
        __builtin_amdgcn_s_barrier();
        // the two waves of a SIMD (w and w + 4) take turns: waves 0-3 issue the next DMA stage (address VALU + 6 LDS-DMA) and
        // then run their MFMAs, waves 4-7 run their MFMAs first and issue afterwards, so one of the pair feeds the matrix pipe
        // while the other one feeds the memory pipe
        if (!late && it + 2 < nsteps) issue((it + 2) % STAGES, mbeg + (it + 2) * MT);
        const unsigned char* sP = smem + (it % STAGES) * SBYTES;
        const unsigned char* sQ = sP + PBYTES;
        if (do_bias) {
#pragma unroll
            for (int jj = 0; jj < B_NJ; ++jj) {
                const uint4 v = *(const uint4*)(sP + (brow + jj * B_RG) * PRB + bpos * 16);
                const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bsum[2 * e] += WgMma<F16>::cvt((unsigned short)(w4[e] & 0xffffu));
                    bsum[2 * e + 1] += WgMma<F16>::cvt((unsigned short)(w4[e] >> 16));
                }
            }
        }
        const unsigned so = (unsigned)((it % STAGES) * SBYTES);
        unsigned pa[4], qa[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { pa[i] = pad0[i] + so; qa[i] = qad0[i] + so; }
        {
            wg_u32x2_t pf0[4][2], qf0[4][2], pf1[4][2], qf1[4][2];
            wg_load_half<PRB, QRB, 0>(pa, qa, pf0, qf0);
            wg_wait_frags(pf0, qf0);
            wg_load_half<PRB, QRB, 1>(pa, qa, pf1, qf1);          // in flight under the first 16 MFMAs
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    WgMma<F16>::run(make_uint4(pf0[i][0].x, pf0[i][0].y, pf0[i][1].x, pf0[i][1].y),
                                    make_uint4(qf0[j][0].x, qf0[j][0].y, qf0[j][1].x, qf0[j][1].y), acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
            wg_wait_frags(pf1, qf1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    WgMma<F16>::run(make_uint4(pf1[i][0].x, pf1[i][0].y, pf1[i][1].x, pf1[i][1].y),
                                    make_uint4(qf1[j][0].x, qf1[j][0].y, qf1[j][1].x, qf1[j][1].y), acc[i][j]);
        }
        if (late && it + 2 < nsteps) issue((it + 2) % STAGES, mbeg + (it + 2) * MT);
    }

    float* const part = Pk.part ? Pk.part + (long)sp * Pk.part_slice : nullptr;
    if (do_bias) {
        // the row groups' column sums meet through the (now idle) stage ring and are added in row-group order: no LDS float
        // atomics, so the bias gradient does not depend on the order in which the waves arrive (deterministic mode relies on it)
        __syncthreads();
        float* const bred = (float*)smem;                  // [B_RG][BN]
#pragma unroll
        for (int e = 0; e < 8; ++e) bred[brow * BN + bchunk * 8 + e] = bsum[e];
        __syncthreads();
        if (t < BN) {
            float v = 0.f;
            for (int rg = 0; rg < B_RG; ++rg) v += bred[rg * BN + t];
            sBias[t] = v;
        }
        if (t < BN && n0 + t < Ng) {
            if (part) part[(long)groups * Ng * Kpad + g * Ng + n0 + t] = sBias[t];
            else atomicAdd(dbias + g * Ng + n0 + t, sBias[t]);
        }
    }
    if (part) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + (wk * 4 + j) * 16 + r;
            if (k >= Kpad) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = n0 + (wn * 4 + i) * 16 + q * 4 + e;
                    if (n < Ng) part[(long)(g * Ng + n) * Kpad + k] = acc[i][j][e];
                }
            }
        }
        return;
    }
    // ---- epilogue: D[row = n (4q + e)][col = k (r)], fp32 atomics into the (channels-last) gradient tensor
    float* const dw = Pk.dw;
    const long s_o = Pk.s_o, s_i = Pk.s_i, s_h = Pk.s_h, s_w = Pk.s_w;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = k0 + (wk * 4 + j) * 16 + r;
        if (k >= Kpad) continue;
        const int tap = k / Cg, ci = k - tap * Cg;
        if (ci >= CgReal) continue;
        const int kh = tap / KW, kw = tap - kh * KW;
        const long koff = (long)ci * s_i + (long)kh * s_h + (long)kw * s_w;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = n0 + (wn * 4 + i) * 16 + q * 4 + e;
                if (n < Ng) atomicAdd(dw + (long)(g * Ng + n) * s_o + koff, acc[i][j][e]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
#ifdef OCTA_DIAG_STAMPS
__device__ unsigned long long octa_diag_stamps_wgrad9[4096][4];
extern "C" int octa_diag_stamps_read_wgrad9(void* host, int clear) {
    if (clear) { static unsigned long long z[4096][4]; return hipMemcpyToSymbol(HIP_SYMBOL(octa_diag_stamps_wgrad9), z, sizeof(z)) == hipSuccess ? 0 : -3; }
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(octa_diag_stamps_wgrad9), sizeof(octa_diag_stamps_wgrad9)) == hipSuccess ? 0 : -3;
}
#endif
// wgrad9: 256(N) x 256(K) output tile per workgroup (round 3).  Same batch interface, DMA roles, zero page and epilogue as
// wgrad8; what changed, and why:
//   * the tile is twice as big and a stage is 32 pixel rows (4 ring slots of 32 KB): per FLOP a wave issues 25 % fewer
//     transposed LDS reads and 33 % fewer LDS-DMA instructions -- wgrad8 was bound by instruction ISSUE (two waves share a
//     SIMD's vector issue port with their own MFMAs), not by the matrix pipe;
//   * v_mfma_f32_32x32x16: half as many MFMA instructions for the same tile, i.e. half the issue slots they hold, and the
//     natural software-pipelining unit becomes a k16 half-stage of 8 MFMAs whose 24 fragment registers are double-buffered
//     ACROSS the stage barrier (the first fragments of stage s+1 are fetched under the last MFMAs of stage s), so no LDS
//     latency is exposed after the barrier; accumulators 128 + fragments 48 registers per lane;
//   * a 3x3 layer now has 18 (not 36) K-tiles per pixel range, which fit the 32 CUs of one XCD: the tiles that stream the same
//     dy rows and the same (tap-shifted) x rows run side by side behind one L2.
// LDS image: rows of 512 B (256 channels); the 32-byte pair index of row m is XOR-ed with f(m) = (m & 3) << 1, which sends the
// 4 rows x 2 pairs that one 32-lane half of a 32x32x16 transposed read touches to 8 distinct 32-byte slots of the 256-byte
// bank row (conflict-free); the same involution is applied to the per-lane SOURCE chunk of the lane-linear LDS-DMA image.
typedef __attribute__((ext_vector_type(16))) float wg_f32x16_t;
template <int F16> struct WgMma32;
template <> struct WgMma32<0> {
    __device__ static __forceinline__ void run(const uint4& a, const uint4& b, wg_f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wg_bf16x8_t, a), __builtin_bit_cast(wg_bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct WgMma32<1> {
    __device__ static __forceinline__ void run(const uint4& a, const uint4& b, wg_f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wg_f16x8_t, a), __builtin_bit_cast(wg_f16x8_t, b), c, 0, 0, 0);
    }
};
// fragments of one k16 half-stage: 4 A blocks (32 output channels each) + 2 B blocks (32 taps x channels each), two transposed
// 8-byte reads per block (pixel rows 8h .. 8h+3 and 8h+4 .. 8h+7 of the half-stage, h = lane >> 5)
template <int HS>
__device__ __forceinline__ void wg9_load_half(const unsigned (&aa)[4], const unsigned (&ba)[2], wg_u32x2_t (&af)[4][2], wg_u32x2_t (&bf)[2][2]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { af[i][0] = wg_tr<HS * 8192>(aa[i]); af[i][1] = wg_tr<HS * 8192 + 2048>(aa[i]); }
#pragma unroll
    for (int j = 0; j < 2; ++j) { bf[j][0] = wg_tr<HS * 8192>(ba[j]); bf[j][1] = wg_tr<HS * 8192 + 2048>(ba[j]); }
}
__device__ __forceinline__ void wg9_wait_frags(wg_u32x2_t (&af)[4][2], wg_u32x2_t (&bf)[2][2]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(af[2][0]), "+v"(af[2][1]), "+v"(af[3][0]), "+v"(af[3][1]),
                   "+v"(bf[0][0]), "+v"(bf[0][1]), "+v"(bf[1][0]), "+v"(bf[1][1])
                 :: "memory");
}
template <int F16>
__device__ __forceinline__ void wg9_mma_half(const wg_u32x2_t (&af)[4][2], const wg_u32x2_t (&bf)[2][2], wg_f32x16_t (&acc)[4][2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            WgMma32<F16>::run(make_uint4(af[i][0].x, af[i][0].y, af[i][1].x, af[i][1].y), make_uint4(bf[j][0].x, bf[j][0].y, bf[j][1].x, bf[j][1].y), acc[i][j]);
}

// ABL: timing-only ablation builds (tools/wgrad_micro.py "ablate"; results are wrong by construction): 1 = no epilogue atomics,
// 2 = no LDS-DMA after the prologue, 4 = no MFMAs, 8 = no transposed LDS reads
template <int F16, int STAGGER, int ABL = 0>
__global__ __launch_bounds__(512) void wgrad9_kernel(const WgBatch batch) {
    constexpr int BN = 256, BK = 256, MT = 32, SLOTS = 4;
    constexpr int RB = 512, IMG = MT * RB, SBYTES = 2 * IMG;       // LDS row bytes, bytes of one stage image (dy or x), bytes of a ring slot
    constexpr int LPR = 32, RPI = 2, IPW = 2;                      // lanes per row, rows per DMA instruction, instructions per wave and image
    constexpr int LPT = 2 * IPW;                                   // DMA instructions per wave and stage
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SLOTS * SBYTES + BN * 4];
    float* const sBias = (float*)(smem + SLOTS * SBYTES);

    // ---- which problem / tile is this workgroup (XCD-contiguous order over the whole launch; within a problem the K-tiles of
    // one pixel range are consecutive, so the tiles that share dy / x rows sit on one XCD)
    const int total = gridDim.x, Lb = blockIdx.x;
    const int xcd = Lb & 7, jq = Lb >> 3, qn = total >> 3, rn = total & 7;
    const int Lp = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + jq;
    int pi = 0;
    for (int i = 1; i < batch.n; ++i) if (batch.p[i].blockStart <= Lp) pi = i;
    const WgProb& Pk = batch.p[pi];
    const int tilesN = Pk.tilesN, tilesK = Pk.tilesK, groups = Pk.groups, mPerSplit = Pk.mPerSplit, Mtot = Pk.M;
    const int Ng = Pk.Ng, Kpad = Pk.Kpad, Cg = Pk.Cg, CgReal = Pk.CgReal, KW = Pk.KW;
    const int H = Pk.H, W = Pk.W, OH = Pk.OH, OW = Pk.OW, stride = Pk.stride, pad = Pk.pad;
    const int ldx = Pk.ldx, ldy = Pk.ldy;
    const unsigned short* const xbase = Pk.x;
    const unsigned short* const dybase = Pk.dy;
    int bid = Lp - Pk.blockStart;
    const int nt = bid % tilesN; bid /= tilesN;
    const int kt = bid % tilesK; bid /= tilesK;
    const int g = bid % groups;
    const int sp = bid / groups;
    const int n0 = nt * BN, k0 = kt * BK;
    const int mbeg = (ABL & 16) ? 0 : sp * mPerSplit;              // ABL 16: every workgroup streams pixel range 0 (all L2 hits)
    const int mend = min(Mtot, mbeg + mPerSplit);
    const int nsteps = (mend - mbeg + MT - 1) / MT;
    if (nsteps <= 0) return;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wn = wave >> 2, wk = wave & 3;                       // 2 x 4 waves of 128(N) x 64(K)
    const unsigned long zaddr = (unsigned long)(const void*)wg8_zero_page;

    // ---- DMA roles (both images have the same geometry): instruction i of this wave fills LDS rows (i*8 + wave)*2 + lane/32
    const int drow = wave * RPI + lane / LPR;
    const int dpos = lane % LPR;
    const int df_ = (drow & 3) << 1;
    const int dchunk = (((dpos >> 1) ^ df_) << 1) | (dpos & 1);
    const bool pvalid = (n0 + dchunk * 8) < Ng;
    const unsigned long pstep = pvalid ? (unsigned long)((long)MT * ldy * 2) : 0ul;
    unsigned long pptr[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i)
        pptr[i] = pvalid ? (unsigned long)(dybase + ((long)(mbeg + drow + i * 8 * RPI) * ldy + Pk.yoff + g * Ng + n0 + dchunk * 8)) : zaddr;
    const int kel = k0 + dchunk * 8;
    const bool kvalid = kel < Kpad;
    const int qtap = kel / Cg, qcc = kel - qtap * Cg;
    const int qkh = qtap / KW, qkw = qtap - qkh * KW;
    const int qdh = qkh - pad, qdw = qkw - pad;
    const unsigned long qbase = (unsigned long)(xbase + (Pk.xoff + g * CgReal + qcc));
    const bool plain = (Pk.KH == 1 && KW == 1 && pad == 0 && stride == 1);   // 1x1: input pixel == output pixel
    const int ldx2 = ldx * 2;
    const int dq = MT / OW, dr = MT - dq * OW;               // a stage advances the output pixel by dq rows + dr columns
    const int sdr = stride * dr, sdq = stride * dq, OWs = OW * stride, OHs = OH * stride;
    const int thrW = OWs + qdw, thrH = OHs + qdh;
    const int dpix = sdq * W + sdr, cW = stride * W - OWs, cH = H * W - OHs * W;
    int qih[IPW], qiw[IPW], qpix[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int m = mbeg + drow + i * 8 * RPI;
        const int ow = m % OW, tq = m / OW, oh = tq % OH, b = tq / OH;
        qih[i] = oh * stride + qdh; qiw[i] = ow * stride + qdw;
        qpix[i] = plain ? m : (b * H + qih[i]) * W + qiw[i];
    }

    const unsigned sbase = wg_lds_addr(smem);
    // one LDS-DMA instruction each, branch-free (invalid rows / taps / channels read the zero page): P = dy rows, Q = x rows of this lane's tap
    auto issueP = [&](int i, int slot, int mcur) {
        const bool ok = (mcur + drow + i * 8 * RPI) < mend;
        const unsigned long src = ok ? pptr[i] : zaddr;
        wg_glds16((const void*)src, __builtin_amdgcn_readfirstlane(sbase + (unsigned)(slot * SBYTES + (i * 8 + wave) * 1024)));
        pptr[i] += pstep;
    };
    auto issueQ = [&](int i, int slot, int mcur) {
        const bool ok = kvalid & ((mcur + drow + i * 8 * RPI) < mend) & ((unsigned)qih[i] < (unsigned)H) & ((unsigned)qiw[i] < (unsigned)W);
        const unsigned off = (unsigned)__mul24(qpix[i], ldx2);              // pixel index < 2^23, ldx2 < 2^17: exact in 24 x 24 bits
        const unsigned long a = qbase + (unsigned long)off;
        const unsigned long src = ok ? a : zaddr;
        wg_glds16((const void*)src, __builtin_amdgcn_readfirstlane(sbase + (unsigned)(slot * SBYTES + IMG + (i * 8 + wave) * 1024)));
        qiw[i] += sdr;
        const bool c1 = qiw[i] >= thrW;
        qiw[i] -= c1 ? OWs : 0;
        qih[i] += sdq + (c1 ? stride : 0);
        const bool c2 = qih[i] >= thrH;
        qih[i] -= c2 ? OHs : 0;
        qpix[i] += dpix + (c1 ? cW : 0) + (c2 ? cH : 0);
    };
    auto issue = [&](int slot, int mcur) { issueP(0, slot, mcur); issueP(1, slot, mcur); issueQ(0, slot, mcur); issueQ(1, slot, mcur); };

    wg_f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- transposed fragment addressing (slot 0, half-stage 0): 16-lane group gq = lane >> 4 reads pixel rows 8 (gq >> 1) + (r >> 2)
    // (+ 4 for the second read) of channel pair 2 * block + (gq & 1); f(row) = ((r >> 2) & 3) << 1 is a lane constant
    const int r = lane & 15, gq = lane >> 4;
    const int frow = 8 * (gq >> 1) + (r >> 2);
    const int fr = ((r >> 2) & 3) << 1;
    const int cb = (r & 3) * 8;
    unsigned abase[4], bbase[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) abase[i] = sbase + (unsigned)(frow * RB + ((((wn * 4 + i) * 2 + (gq & 1)) ^ fr) << 5) + cb);
#pragma unroll
    for (int j = 0; j < 2; ++j) bbase[j] = sbase + (unsigned)(IMG + frow * RB + ((((wk * 2 + j) * 2 + (gq & 1)) ^ fr) << 5) + cb);

    // ---- fused bias gradient (k-tile 0 only): thread = (16-byte chunk position, rows brow and brow + 16) of the dy stage image
    float* const dbias = Pk.dbias;
    const bool do_bias = (dbias != nullptr) && (kt == 0);
    const int brow = t / LPR, bpos = t % LPR;
    const int bf_ = (brow & 3) << 1;
    const int bchunk = (((bpos >> 1) ^ bf_) << 1) | (bpos & 1);
    float bsum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[e] = 0.f;
    if (do_bias && t < BN) sBias[t] = 0.f;

    issue(0, mbeg);
    if (nsteps > 1) issue(1, mbeg + MT);
    if (nsteps > 2) issue(2, mbeg + 2 * MT);
    if (nsteps > 2) wg_wait_vmcnt<2 * LPT>(); else if (nsteps > 1) wg_wait_vmcnt<LPT>(); else wg_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    wg_u32x2_t afX[4][2], bfX[2][2], afY[4][2], bfY[2][2];
    if (ABL & 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { afY[i][0] = (wg_u32x2_t){0x3f803f80u, 0x3f803f80u}; afY[i][1] = afY[i][0]; }
#pragma unroll
        for (int j = 0; j < 2; ++j) { bfY[j][0] = (wg_u32x2_t){0x3f803f80u, 0x3f803f80u}; bfY[j][1] = bfY[j][0]; }
    }
    {
        unsigned aa[4], ba[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) aa[i] = abase[i];
#pragma unroll
        for (int j = 0; j < 2; ++j) ba[j] = bbase[j];
        wg9_load_half<0>(aa, ba, afX, bfX);
        wg9_wait_frags(afX, bfX);
    }
    // One MFMA, then the instructions that ride in its shadow (a 32x32x16 MFMA occupies the matrix pipe for 32 cycles and holds the
    // vector issue port for 8 of them): two transposed reads of the NEXT half-stage's fragments, or one LDS-DMA instruction of
    // stage it + 3 with its address arithmetic.  The order is pinned (sched_barrier): issuing all 12 reads in front of the 8
    // MFMAs, as the first version did, left the matrix pipe idle for ~150 cycles per half-stage.
#define WG9_MMA(AF, BF, i, j)                                                                                                              \
    if (!(ABL & 4)) WgMma32<F16>::run(make_uint4(AF[i][0].x, AF[i][0].y, AF[i][1].x, AF[i][1].y), make_uint4(BF[j][0].x, BF[j][0].y, BF[j][1].x, BF[j][1].y), acc[i][j])
#define WG9_TR(F, i, HS, ad)                                                                                                               \
    if (!(ABL & 8)) { F[i][0] = wg_tr<HS * 8192>(ad); F[i][1] = wg_tr<HS * 8192 + 2048>(ad); }
#define WG9_SB __builtin_amdgcn_sched_barrier(0)
    OCTA_STAMP_DECL;
    OCTA_STAMP_BEGIN;
    for (int it = 0; it < nsteps; ++it) {
        const int rem = nsteps - 1 - it;                            // stages after this one
        const bool more = !(ABL & 2) && rem >= 3;                   // stage it + 3 goes into the slot of stage it - 1 (its reads ended before the last barrier)
        const int s3 = (it + 3) & (SLOTS - 1), m3 = mbeg + (it + 3) * MT;
        const unsigned so = (unsigned)((it & (SLOTS - 1)) * SBYTES);
        const unsigned sn = (unsigned)(((it + 1) & (SLOTS - 1)) * SBYTES);
        if (do_bias) {
            const unsigned char* sP = smem + so;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const uint4 v = *(const uint4*)(sP + (brow + jj * 16) * RB + bpos * 16);
                const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bsum[2 * e] += WgMma<F16>::cvt((unsigned short)(w4[e] & 0xffffu));
                    bsum[2 * e + 1] += WgMma<F16>::cvt((unsigned short)(w4[e] >> 16));
                }
            }
        }
        WG9_SB;
        // ---- half-stage 0 of stage it (fragments X); fetch half-stage 1 (Y); first half of the DMA
        WG9_MMA(afX, bfX, 0, 0); WG9_TR(afY, 0, 1, abase[0] + so); WG9_SB;
        WG9_MMA(afX, bfX, 1, 0); WG9_TR(afY, 1, 1, abase[1] + so); WG9_SB;
        WG9_MMA(afX, bfX, 2, 0); WG9_TR(afY, 2, 1, abase[2] + so); WG9_SB;
        WG9_MMA(afX, bfX, 3, 0); WG9_TR(afY, 3, 1, abase[3] + so); WG9_SB;
        WG9_MMA(afX, bfX, 0, 1); WG9_TR(bfY, 0, 1, bbase[0] + so); WG9_SB;
        WG9_MMA(afX, bfX, 1, 1); WG9_TR(bfY, 1, 1, bbase[1] + so); WG9_SB;
        WG9_MMA(afX, bfX, 2, 1); if (more) issueP(0, s3, m3); WG9_SB;
        WG9_MMA(afX, bfX, 3, 1); if (more) issueP(1, s3, m3); WG9_SB;
        if (ABL & 2) wg_wait_vmcnt<0>();
        else if (rem >= 3) wg_wait_vmcnt<LPT + 2>(); else if (rem == 2) wg_wait_vmcnt<LPT>(); else wg_wait_vmcnt<0>();   // stage it + 1 has landed (this wave's part)
        wg9_wait_frags(afY, bfY);                                   // ... and every LDS read of stage it has returned
        __builtin_amdgcn_s_barrier();
        WG9_SB;
        // ---- half-stage 1 (fragments Y); fetch half-stage 0 of stage it + 1 (X); second half of the DMA
        WG9_MMA(afY, bfY, 0, 0); WG9_TR(afX, 0, 0, abase[0] + sn); WG9_SB;
        WG9_MMA(afY, bfY, 1, 0); WG9_TR(afX, 1, 0, abase[1] + sn); WG9_SB;
        WG9_MMA(afY, bfY, 2, 0); WG9_TR(afX, 2, 0, abase[2] + sn); WG9_SB;
        WG9_MMA(afY, bfY, 3, 0); WG9_TR(afX, 3, 0, abase[3] + sn); WG9_SB;
        WG9_MMA(afY, bfY, 0, 1); WG9_TR(bfX, 0, 0, bbase[0] + sn); WG9_SB;
        WG9_MMA(afY, bfY, 1, 1); WG9_TR(bfX, 1, 0, bbase[1] + sn); WG9_SB;
        WG9_MMA(afY, bfY, 2, 1); if (more) issueQ(0, s3, m3); WG9_SB;
        WG9_MMA(afY, bfY, 3, 1); if (more) issueQ(1, s3, m3); WG9_SB;
        wg9_wait_frags(afX, bfX);
        WG9_SB;
    }
#undef WG9_MMA
#undef WG9_TR
#undef WG9_SB
    OCTA_STAMP_END(octa_diag_stamps_wgrad9)

    float* const part = Pk.part ? Pk.part + (long)sp * Pk.part_slice : nullptr;
    if (do_bias) {
        // the row groups' column sums meet through the (now idle) stage ring and are added in row-group order: no LDS float
        // atomics, so the bias gradient does not depend on the order in which the waves arrive (deterministic mode relies on it)
        __syncthreads();
        float* const bred = (float*)smem;                  // [512 / LPR][BN]
#pragma unroll
        for (int e = 0; e < 8; ++e) bred[brow * BN + bchunk * 8 + e] = bsum[e];
        __syncthreads();
        if (t < BN) {
            float v = 0.f;
            for (int rg = 0; rg < 512 / LPR; ++rg) v += bred[rg * BN + t];
            sBias[t] = v;
        }
        if (t < BN && n0 + t < Ng) {
            if (part) part[(long)groups * Ng * Kpad + g * Ng + n0 + t] = sBias[t];
            else atomicAdd(dbias + g * Ng + n0 + t, sBias[t]);
        }
    }
    if (part) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = k0 + wk * 64 + j * 32 + (lane & 31);
            if (k >= Kpad) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int n = n0 + wn * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    if (n < Ng) part[(long)(g * Ng + n) * Kpad + k] = acc[i][j][e];
                }
            }
        }
        return;
    }
    // ---- epilogue: 32x32 accumulator block (i, j): column c = lane & 31, row n = (e & 3) + 8 (e >> 2) + 4 (lane >> 5); one register
    // of all lanes = two 128-byte runs along Cin of the channels-last gradient tensor (full-rate float atomics)
    float* const dw = Pk.dw;
    const long s_o = Pk.s_o, s_i = Pk.s_i, s_h = Pk.s_h, s_w = Pk.s_w;
    const int lc = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = k0 + wk * 64 + j * 32 + lc;
        if (k >= Kpad) continue;
        const int tap = k / Cg, ci = k - tap * Cg;
        if (ci >= CgReal) continue;
        const int kh = tap / KW, kw = tap - kh * KW;
        const long koff = (long)ci * s_i + (long)kh * s_h + (long)kw * s_w;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + wn * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (ABL & 1) { asm volatile("" :: "v"(acc[i][j][e])); continue; }
                if (n < Ng) atomicAdd(dw + (long)(g * Ng + n) * s_o + koff, acc[i][j][e]);
            }
        }
    }
}


// wgrad9x: the wgrad9 tile loop behind an XCD-interleaved, optionally persistent schedule (round 5).
// What the launch log of a training step showed (OCTA_WG_LOG): a wgrad9 launch is 3-5 lock-step rounds of 256 workgroups, every
// round ends in one burst of 64 MB of float atomics (~45 us at the chip's 1.3 TB/s, the CUs idle meanwhile), and one split length
// for all problems of a batch quantises the rounds.  Here
//   * every problem carries its own split (host: per-class search over a model of this schedule), and its blocks -- pixel range
//     slowest, tiles fastest, as before -- are dealt to the eight XCDs in eighths, so that every XCD gets the same mix;
//   * an XCD's sequence is cyclic and XCD x starts at position sched.off[x]: the XCDs are in different problems at any time, their
//     rounds end at different times and the atomics of one XCD's epilogues drain beside the main loops of the others;
//   * PERSIST = 1: gridDim.x = 8 * SL resident workgroups; workgroup (xcd, slot) takes the positions slot, slot + SL, ... of its
//     XCD's sequence (no launch ramp between blocks; the static deal equals the hardware's when a round's blocks are equally long);
//     PERSIST = 0: one position per workgroup, gridDim.x = 8 * (longest XCD sequence).
// blockIdx & 7 is used as the XCD label (a performance assumption only: every position is processed exactly once whatever the
// placement).
// wgrad9a: wgrad9 with FOUR waves of 128(N) x 128(K) per workgroup -- one wave per SIMD, 256 accumulator registers in the unified file
// (round 5; octa_tuning_set(9, 2)).  Why it was tried: per 32-pixel stage a workgroup of wgrad9 reads 96 KB of fragments from LDS (96 B / clk
// at full MFMA rate: 37 % of the LDS array) and issues those reads between its MFMAs.  A 128 x 128 wave tile needs (4 + 4) fragment blocks per
// 16 MFMAs instead of (4 + 2) per 8: -25 % fragment reads per FLOP, one read pair per MFMA.  Measured 9-22 % SLOWER: with one
// wave per SIMD nothing covers a wave's barrier and vmcnt waits.  Same LDS image, DMA image, ring, barrier protocol and epilogue
// mapping as wgrad9; a wave now issues 8 LDS-DMA instructions per stage and interleaves one transposed read pair with every MFMA.
__device__ __forceinline__ void wg9s_wait16(wg_u32x2_t (&a)[4][2], wg_u32x2_t (&b)[4][2]);
template <int HS>
__device__ __forceinline__ void wg9a_load_half(const unsigned (&aa)[4], const unsigned (&ba)[4], wg_u32x2_t (&af)[4][2], wg_u32x2_t (&bf)[4][2]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { af[i][0] = wg_tr<HS * 8192>(aa[i]); af[i][1] = wg_tr<HS * 8192 + 2048>(aa[i]); }
#pragma unroll
    for (int j = 0; j < 4; ++j) { bf[j][0] = wg_tr<HS * 8192>(ba[j]); bf[j][1] = wg_tr<HS * 8192 + 2048>(ba[j]); }
}

template <int F16>
__global__ __launch_bounds__(256) void wgrad9a_kernel(const WgBatch batch) {
    constexpr int BN = 256, BK = 256, MT = 32, SLOTS = 4;
    constexpr int RB = 512, IMG = MT * RB, SBYTES = 2 * IMG;
    constexpr int LPR = 32, RPI = 2, IPW = 4;                      // lanes per row, rows per DMA instruction, instructions per wave and image
    constexpr int LPT = 2 * IPW;                                   // DMA instructions per wave and stage (8)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SLOTS * SBYTES + BN * 4];
    float* const sBias = (float*)(smem + SLOTS * SBYTES);

    const int total = gridDim.x, Lb = blockIdx.x;
    const int xcd = Lb & 7, jq = Lb >> 3, qn = total >> 3, rn = total & 7;
    const int Lp = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + jq;
    int pi = 0;
    for (int i = 1; i < batch.n; ++i) if (batch.p[i].blockStart <= Lp) pi = i;
    const WgProb& Pk = batch.p[pi];
    const int tilesN = Pk.tilesN, tilesK = Pk.tilesK, groups = Pk.groups, mPerSplit = Pk.mPerSplit, Mtot = Pk.M;
    const int Ng = Pk.Ng, Kpad = Pk.Kpad, Cg = Pk.Cg, CgReal = Pk.CgReal, KW = Pk.KW;
    const int H = Pk.H, W = Pk.W, OH = Pk.OH, OW = Pk.OW, stride = Pk.stride, pad = Pk.pad;
    const int ldx = Pk.ldx, ldy = Pk.ldy;
    const unsigned short* const xbase = Pk.x;
    const unsigned short* const dybase = Pk.dy;
    int bid = Lp - Pk.blockStart;
    const int nt = bid % tilesN; bid /= tilesN;
    const int kt = bid % tilesK; bid /= tilesK;
    const int g = bid % groups;
    const int sp = bid / groups;
    const int n0 = nt * BN, k0 = kt * BK;
    const int mbeg = sp * mPerSplit;
    const int mend = min(Mtot, mbeg + mPerSplit);
    const int nsteps = (mend - mbeg + MT - 1) / MT;
    if (nsteps <= 0) return;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wn = wave >> 1, wk = wave & 1;                       // 2 x 2 waves of 128(N) x 128(K)
    const unsigned long zaddr = (unsigned long)(const void*)wg8_zero_page;

    // ---- DMA roles: instruction i of this wave fills LDS rows (i*4 + wave)*2 + lane/32 of an image (a lane's rows differ by 8: same f)
    const int drow = wave * RPI + lane / LPR;
    const int dpos = lane % LPR;
    const int df_ = (drow & 3) << 1;
    const int dchunk = (((dpos >> 1) ^ df_) << 1) | (dpos & 1);
    const bool pvalid = (n0 + dchunk * 8) < Ng;
    const unsigned long pstep = pvalid ? (unsigned long)((long)MT * ldy * 2) : 0ul;
    unsigned long pptr[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i)
        pptr[i] = pvalid ? (unsigned long)(dybase + ((long)(mbeg + drow + i * 4 * RPI) * ldy + Pk.yoff + g * Ng + n0 + dchunk * 8)) : zaddr;
    const int kel = k0 + dchunk * 8;
    const bool kvalid = kel < Kpad;
    const int qtap = kel / Cg, qcc = kel - qtap * Cg;
    const int qkh = qtap / KW, qkw = qtap - qkh * KW;
    const int qdh = qkh - pad, qdw = qkw - pad;
    const unsigned long qbase = (unsigned long)(xbase + (Pk.xoff + g * CgReal + qcc));
    const bool plain = (Pk.KH == 1 && KW == 1 && pad == 0 && stride == 1);
    const int ldx2 = ldx * 2;
    const int dq = MT / OW, dr = MT - dq * OW;
    const int sdr = stride * dr, sdq = stride * dq, OWs = OW * stride, OHs = OH * stride;
    const int thrW = OWs + qdw, thrH = OHs + qdh;
    const int dpix = sdq * W + sdr, cW = stride * W - OWs, cH = H * W - OHs * W;
    int qih[IPW], qiw[IPW], qpix[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int m = mbeg + drow + i * 4 * RPI;
        const int ow = m % OW, tq = m / OW, oh = tq % OH, b = tq / OH;
        qih[i] = oh * stride + qdh; qiw[i] = ow * stride + qdw;
        qpix[i] = plain ? m : (b * H + qih[i]) * W + qiw[i];
    }
    const unsigned sbase = wg_lds_addr(smem);
    auto issueP = [&](int i, int slot, int mcur) {
        const bool ok = (mcur + drow + i * 4 * RPI) < mend;
        const unsigned long src = ok ? pptr[i] : zaddr;
        wg_glds16((const void*)src, __builtin_amdgcn_readfirstlane(sbase + (unsigned)(slot * SBYTES + (i * 4 + wave) * 1024)));
        pptr[i] += pstep;
    };
    auto issueQ = [&](int i, int slot, int mcur) {
        const bool ok = kvalid & ((mcur + drow + i * 4 * RPI) < mend) & ((unsigned)qih[i] < (unsigned)H) & ((unsigned)qiw[i] < (unsigned)W);
        const unsigned off = (unsigned)__mul24(qpix[i], ldx2);
        const unsigned long a = qbase + (unsigned long)off;
        const unsigned long src = ok ? a : zaddr;
        wg_glds16((const void*)src, __builtin_amdgcn_readfirstlane(sbase + (unsigned)(slot * SBYTES + IMG + (i * 4 + wave) * 1024)));
        qiw[i] += sdr;
        const bool c1 = qiw[i] >= thrW;
        qiw[i] -= c1 ? OWs : 0;
        qih[i] += sdq + (c1 ? stride : 0);
        const bool c2 = qih[i] >= thrH;
        qih[i] -= c2 ? OHs : 0;
        qpix[i] += dpix + (c1 ? cW : 0) + (c2 ? cH : 0);
    };
    auto issue = [&](int slot, int mcur) {
        issueP(0, slot, mcur); issueP(1, slot, mcur); issueP(2, slot, mcur); issueP(3, slot, mcur);
        issueQ(0, slot, mcur); issueQ(1, slot, mcur); issueQ(2, slot, mcur); issueQ(3, slot, mcur);
    };

    wg_f32x16_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- transposed fragment addressing as in wgrad9: block b = 32 channels = pairs 2 b, 2 b + 1
    const int r = lane & 15, gq = lane >> 4;
    const int frow = 8 * (gq >> 1) + (r >> 2);
    const int fr = ((r >> 2) & 3) << 1;
    const int cb = (r & 3) * 8;
    unsigned abase[4], bbase[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) abase[i] = sbase + (unsigned)(frow * RB + ((((wn * 4 + i) * 2 + (gq & 1)) ^ fr) << 5) + cb);
#pragma unroll
    for (int j = 0; j < 4; ++j) bbase[j] = sbase + (unsigned)(IMG + frow * RB + ((((wk * 4 + j) * 2 + (gq & 1)) ^ fr) << 5) + cb);

    // ---- fused bias gradient (k-tile 0 only): thread = (16-byte chunk position, rows brow + 8 jj) of the dy stage image
    float* const dbias = Pk.dbias;
    const bool do_bias = (dbias != nullptr) && (kt == 0);
    const int brow = t / LPR, bpos = t % LPR;                      // brow 0 .. 7
    const int bf_ = (brow & 3) << 1;
    const int bchunk = (((bpos >> 1) ^ bf_) << 1) | (bpos & 1);
    float bsum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[e] = 0.f;
    if (do_bias && t < BN) sBias[t] = 0.f;

    issue(0, mbeg);
    if (nsteps > 1) issue(1, mbeg + MT);
    if (nsteps > 2) issue(2, mbeg + 2 * MT);
    if (nsteps > 2) wg_wait_vmcnt<2 * LPT>(); else if (nsteps > 1) wg_wait_vmcnt<LPT>(); else wg_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    wg_u32x2_t afX[4][2], bfX[4][2], afY[4][2], bfY[4][2];
    wg9a_load_half<0>(abase, bbase, afX, bfX);
    wg9s_wait16(afX, bfX);
#define WG9A_MMA(AF, BF, i, j)                                                                                                             \
    WgMma32<F16>::run(make_uint4(AF[i][0].x, AF[i][0].y, AF[i][1].x, AF[i][1].y), make_uint4(BF[j][0].x, BF[j][0].y, BF[j][1].x, BF[j][1].y), acc[i][j])
#define WG9A_TR(F, i, HS, ad) { F[i][0] = wg_tr<HS * 8192>(ad); F[i][1] = wg_tr<HS * 8192 + 2048>(ad); }
#define WG9A_SB __builtin_amdgcn_sched_barrier(0)
    OCTA_STAMP_DECL;
    OCTA_STAMP_BEGIN;
    for (int it = 0; it < nsteps; ++it) {
        const int rem = nsteps - 1 - it;
        const bool more = rem >= 3;
        const int s3 = (it + 3) & (SLOTS - 1), m3 = mbeg + (it + 3) * MT;
        const unsigned so = (unsigned)((it & (SLOTS - 1)) * SBYTES);
        const unsigned sn = (unsigned)(((it + 1) & (SLOTS - 1)) * SBYTES);
        if (do_bias) {
            const unsigned char* sP = smem + so;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const uint4 v = *(const uint4*)(sP + (brow + jj * 8) * RB + bpos * 16);
                const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bsum[2 * e] += WgMma<F16>::cvt((unsigned short)(w4[e] & 0xffffu));
                    bsum[2 * e + 1] += WgMma<F16>::cvt((unsigned short)(w4[e] >> 16));
                }
            }
        }
        WG9A_SB;
        // ---- half-stage 0 (fragments X): 16 MFMAs; behind them the 8 fragment blocks of half-stage 1 (Y) and the dy half of the DMA
        WG9A_MMA(afX, bfX, 0, 0); WG9A_TR(afY, 0, 1, abase[0] + so); WG9A_SB;
        WG9A_MMA(afX, bfX, 1, 0); WG9A_TR(afY, 1, 1, abase[1] + so); WG9A_SB;
        WG9A_MMA(afX, bfX, 2, 0); WG9A_TR(afY, 2, 1, abase[2] + so); WG9A_SB;
        WG9A_MMA(afX, bfX, 3, 0); WG9A_TR(afY, 3, 1, abase[3] + so); WG9A_SB;
        WG9A_MMA(afX, bfX, 0, 1); WG9A_TR(bfY, 0, 1, bbase[0] + so); WG9A_SB;
        WG9A_MMA(afX, bfX, 1, 1); WG9A_TR(bfY, 1, 1, bbase[1] + so); WG9A_SB;
        WG9A_MMA(afX, bfX, 2, 1); WG9A_TR(bfY, 2, 1, bbase[2] + so); WG9A_SB;
        WG9A_MMA(afX, bfX, 3, 1); WG9A_TR(bfY, 3, 1, bbase[3] + so); WG9A_SB;
        WG9A_MMA(afX, bfX, 0, 2); if (more) issueP(0, s3, m3); WG9A_SB;
        WG9A_MMA(afX, bfX, 1, 2); WG9A_SB;
        WG9A_MMA(afX, bfX, 2, 2); if (more) issueP(1, s3, m3); WG9A_SB;
        WG9A_MMA(afX, bfX, 3, 2); WG9A_SB;
        WG9A_MMA(afX, bfX, 0, 3); if (more) issueP(2, s3, m3); WG9A_SB;
        WG9A_MMA(afX, bfX, 1, 3); WG9A_SB;
        WG9A_MMA(afX, bfX, 2, 3); if (more) issueP(3, s3, m3); WG9A_SB;
        WG9A_MMA(afX, bfX, 3, 3); WG9A_SB;
        if (rem >= 3) wg_wait_vmcnt<LPT + IPW>(); else if (rem == 2) wg_wait_vmcnt<LPT>(); else wg_wait_vmcnt<0>();   // stage it + 1 has landed (this wave's part)
        wg9s_wait16(afY, bfY);
        __builtin_amdgcn_s_barrier();
        WG9A_SB;
        // ---- half-stage 1 (fragments Y); fetch half-stage 0 of stage it + 1 (X); the x half of the DMA
        WG9A_MMA(afY, bfY, 0, 0); WG9A_TR(afX, 0, 0, abase[0] + sn); WG9A_SB;
        WG9A_MMA(afY, bfY, 1, 0); WG9A_TR(afX, 1, 0, abase[1] + sn); WG9A_SB;
        WG9A_MMA(afY, bfY, 2, 0); WG9A_TR(afX, 2, 0, abase[2] + sn); WG9A_SB;
        WG9A_MMA(afY, bfY, 3, 0); WG9A_TR(afX, 3, 0, abase[3] + sn); WG9A_SB;
        WG9A_MMA(afY, bfY, 0, 1); WG9A_TR(bfX, 0, 0, bbase[0] + sn); WG9A_SB;
        WG9A_MMA(afY, bfY, 1, 1); WG9A_TR(bfX, 1, 0, bbase[1] + sn); WG9A_SB;
        WG9A_MMA(afY, bfY, 2, 1); WG9A_TR(bfX, 2, 0, bbase[2] + sn); WG9A_SB;
        WG9A_MMA(afY, bfY, 3, 1); WG9A_TR(bfX, 3, 0, bbase[3] + sn); WG9A_SB;
        WG9A_MMA(afY, bfY, 0, 2); if (more) issueQ(0, s3, m3); WG9A_SB;
        WG9A_MMA(afY, bfY, 1, 2); WG9A_SB;
        WG9A_MMA(afY, bfY, 2, 2); if (more) issueQ(1, s3, m3); WG9A_SB;
        WG9A_MMA(afY, bfY, 3, 2); WG9A_SB;
        WG9A_MMA(afY, bfY, 0, 3); if (more) issueQ(2, s3, m3); WG9A_SB;
        WG9A_MMA(afY, bfY, 1, 3); WG9A_SB;
        WG9A_MMA(afY, bfY, 2, 3); if (more) issueQ(3, s3, m3); WG9A_SB;
        WG9A_MMA(afY, bfY, 3, 3); WG9A_SB;
        wg9s_wait16(afX, bfX);
        WG9A_SB;
        // (the 256 accumulators fill the AGPR half of the register file exactly: naming them here keeps the allocator from routing two
        // blocks through scratch around the back edge)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+a"(acc[i][j]));
    }
#undef WG9A_MMA
#undef WG9A_TR
#undef WG9A_SB
    OCTA_STAMP_END(octa_diag_stamps_wgrad9)

    float* const part = Pk.part ? Pk.part + (long)sp * Pk.part_slice : nullptr;
    if (do_bias) {
        __syncthreads();
        float* const bred = (float*)smem;                  // [256 / LPR][BN]
#pragma unroll
        for (int e = 0; e < 8; ++e) bred[brow * BN + bchunk * 8 + e] = bsum[e];
        __syncthreads();
        if (t < BN) {
            float v = 0.f;
            for (int rg = 0; rg < 256 / LPR; ++rg) v += bred[rg * BN + t];
            sBias[t] = v;
        }
        if (t < BN && n0 + t < Ng) {
            if (part) part[(long)groups * Ng * Kpad + g * Ng + n0 + t] = sBias[t];
            else atomicAdd(dbias + g * Ng + n0 + t, sBias[t]);
        }
    }
    if (part) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + wk * 128 + j * 32 + (lane & 31);
            if (k >= Kpad) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_sched_barrier(0);          // one block at a time: the accumulators leave the AGPRs 16 at a time, not all 256 at once
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int n = n0 + wn * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    if (n < Ng) part[(long)(g * Ng + n) * Kpad + k] = acc[i][j][e];
                }
            }
        }
        return;
    }
    float* const dw = Pk.dw;
    const long s_o = Pk.s_o, s_i = Pk.s_i, s_h = Pk.s_h, s_w = Pk.s_w;
    const int lc = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = k0 + wk * 128 + j * 32 + lc;
        if (k >= Kpad) continue;
        const int tap = k / Cg, ci = k - tap * Cg;
        if (ci >= CgReal) continue;
        const int kh = tap / KW, kw = tap - kh * KW;
        const long koff = (long)ci * s_i + (long)kh * s_h + (long)kw * s_w;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_barrier(0);              // (as above)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + wn * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (n < Ng) atomicAdd(dw + (long)(g * Ng + n) * s_o + koff, acc[i][j][e]);
            }
        }
    }
}

// wgrad9s: wgrad9 on v_mfma_f32_16x16x32 (round 5; octa_tuning_set(9, 1)).  3.12 showed a long wgrad9 launch to be power-limited
// (in-kernel clock 1.35-1.6 GHz): under that cap the 16x16x32 shape is the one the guide measures 1.12-1.15x ahead of 32x32x16
// (MI355X_MICROARCH.md, DVFS give-back 7).  Same batch interface, tile (256 x 256, 8 waves of 128(N) x 64(K)), 32-pixel stages, 4-slot
// ring, DMA roles, zero page and epilogue protocol as wgrad9; what differs:
//   * a stage is ONE k32 step: 8 A blocks x 4 B blocks of 16 channels = 32 MFMAs per wave, cut into two sub-steps by the A blocks
//     (A0 = blocks 0-3, A1 = 4-7) around the stage barrier.  Sub-step 0 runs A0 x B and fetches A1 (8 transposed reads) and the dy
//     half of the DMA; sub-step 1 runs A1 x B and fetches A0 and B of stage it + 1 (16 reads) and the x half of the DMA.  B is
//     double-buffered across stages (the loop is unrolled by two): 64 fragment registers + 128 accumulators;
//   * lane (r = lane & 15, gq = lane >> 4) reads pixel rows 8 gq + (r >> 2) (+ 4) of a 16-channel block, so one 32-lane half touches
//     rows {0..3, 8..11}: the 32-byte pair index is XOR-ed with f(m) = (m & 3) | ((m >> 3) & 1) << 2 (wgrad8's involution) instead of
//     wgrad9's (m & 3) << 1, in the fragment addresses, the DMA source chunks and the bias reads alike;
//   * a 16x16 accumulator register is 4 rows x 64 bytes of the gradient tensor (the 32x32 form: 2 x 128 bytes).
__device__ __forceinline__ void wg9s_wait16(wg_u32x2_t (&a)[4][2], wg_u32x2_t (&b)[4][2]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[2][0]), "+v"(a[2][1]), "+v"(a[3][0]), "+v"(a[3][1]),
                   "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[2][0]), "+v"(b[2][1]), "+v"(b[3][0]), "+v"(b[3][1])
                 :: "memory");
}
__device__ __forceinline__ void wg9s_wait8(wg_u32x2_t (&a)[4][2]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[2][0]), "+v"(a[2][1]), "+v"(a[3][0]), "+v"(a[3][1])
                 :: "memory");
}

template <int F16>
__global__ __launch_bounds__(512) void wgrad9s_kernel(const WgBatch batch) {
    constexpr int BN = 256, BK = 256, MT = 32, SLOTS = 4;
    constexpr int RB = 512, IMG = MT * RB, SBYTES = 2 * IMG;
    constexpr int LPR = 32, RPI = 2, IPW = 2;
    constexpr int LPT = 2 * IPW;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SLOTS * SBYTES + BN * 4];
    float* const sBias = (float*)(smem + SLOTS * SBYTES);

    const int total = gridDim.x, Lb = blockIdx.x;
    const int xcd = Lb & 7, jq = Lb >> 3, qn = total >> 3, rn = total & 7;
    const int Lp = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + jq;
    int pi = 0;
    for (int i = 1; i < batch.n; ++i) if (batch.p[i].blockStart <= Lp) pi = i;
    const WgProb& Pk = batch.p[pi];
    const int tilesN = Pk.tilesN, tilesK = Pk.tilesK, groups = Pk.groups, mPerSplit = Pk.mPerSplit, Mtot = Pk.M;
    const int Ng = Pk.Ng, Kpad = Pk.Kpad, Cg = Pk.Cg, CgReal = Pk.CgReal, KW = Pk.KW;
    const int H = Pk.H, W = Pk.W, OH = Pk.OH, OW = Pk.OW, stride = Pk.stride, pad = Pk.pad;
    const int ldx = Pk.ldx, ldy = Pk.ldy;
    const unsigned short* const xbase = Pk.x;
    const unsigned short* const dybase = Pk.dy;
    int bid = Lp - Pk.blockStart;
    const int nt = bid % tilesN; bid /= tilesN;
    const int kt = bid % tilesK; bid /= tilesK;
    const int g = bid % groups;
    const int sp = bid / groups;
    const int n0 = nt * BN, k0 = kt * BK;
    const int mbeg = sp * mPerSplit;
    const int mend = min(Mtot, mbeg + mPerSplit);
    const int nsteps = (mend - mbeg + MT - 1) / MT;
    if (nsteps <= 0) return;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wn = wave >> 2, wk = wave & 3;                       // 2 x 4 waves of 128(N) x 64(K)
    const unsigned long zaddr = (unsigned long)(const void*)wg8_zero_page;

    // ---- DMA roles: as wgrad9, with the 16x16x32 involution (a lane's rows differ by 16: same f)
    const int drow = wave * RPI + lane / LPR;
    const int dpos = lane % LPR;
    const int df_ = (drow & 3) | (((drow >> 3) & 1) << 2);
    const int dchunk = (((dpos >> 1) ^ df_) << 1) | (dpos & 1);
    const bool pvalid = (n0 + dchunk * 8) < Ng;
    const unsigned long pstep = pvalid ? (unsigned long)((long)MT * ldy * 2) : 0ul;
    unsigned long pptr[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i)
        pptr[i] = pvalid ? (unsigned long)(dybase + ((long)(mbeg + drow + i * 8 * RPI) * ldy + Pk.yoff + g * Ng + n0 + dchunk * 8)) : zaddr;
    const int kel = k0 + dchunk * 8;
    const bool kvalid = kel < Kpad;
    const int qtap = kel / Cg, qcc = kel - qtap * Cg;
    const int qkh = qtap / KW, qkw = qtap - qkh * KW;
    const int qdh = qkh - pad, qdw = qkw - pad;
    const unsigned long qbase = (unsigned long)(xbase + (Pk.xoff + g * CgReal + qcc));
    const bool plain = (Pk.KH == 1 && KW == 1 && pad == 0 && stride == 1);
    const int ldx2 = ldx * 2;
    const int dq = MT / OW, dr = MT - dq * OW;
    const int sdr = stride * dr, sdq = stride * dq, OWs = OW * stride, OHs = OH * stride;
    const int thrW = OWs + qdw, thrH = OHs + qdh;
    const int dpix = sdq * W + sdr, cW = stride * W - OWs, cH = H * W - OHs * W;
    int qih[IPW], qiw[IPW], qpix[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int m = mbeg + drow + i * 8 * RPI;
        const int ow = m % OW, tq = m / OW, oh = tq % OH, b = tq / OH;
        qih[i] = oh * stride + qdh; qiw[i] = ow * stride + qdw;
        qpix[i] = plain ? m : (b * H + qih[i]) * W + qiw[i];
    }
    const unsigned sbase = wg_lds_addr(smem);
    auto issueP = [&](int i, int slot, int mcur) {
        const bool ok = (mcur + drow + i * 8 * RPI) < mend;
        const unsigned long src = ok ? pptr[i] : zaddr;
        wg_glds16((const void*)src, __builtin_amdgcn_readfirstlane(sbase + (unsigned)(slot * SBYTES + (i * 8 + wave) * 1024)));
        pptr[i] += pstep;
    };
    auto issueQ = [&](int i, int slot, int mcur) {
        const bool ok = kvalid & ((mcur + drow + i * 8 * RPI) < mend) & ((unsigned)qih[i] < (unsigned)H) & ((unsigned)qiw[i] < (unsigned)W);
        const unsigned off = (unsigned)__mul24(qpix[i], ldx2);
        const unsigned long a = qbase + (unsigned long)off;
        const unsigned long src = ok ? a : zaddr;
        wg_glds16((const void*)src, __builtin_amdgcn_readfirstlane(sbase + (unsigned)(slot * SBYTES + IMG + (i * 8 + wave) * 1024)));
        qiw[i] += sdr;
        const bool c1 = qiw[i] >= thrW;
        qiw[i] -= c1 ? OWs : 0;
        qih[i] += sdq + (c1 ? stride : 0);
        const bool c2 = qih[i] >= thrH;
        qih[i] -= c2 ? OHs : 0;
        qpix[i] += dpix + (c1 ? cW : 0) + (c2 ? cH : 0);
    };
    auto issue = [&](int slot, int mcur) { issueP(0, slot, mcur); issueP(1, slot, mcur); issueQ(0, slot, mcur); issueQ(1, slot, mcur); };

    wg_f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (wg_f32x4_t){0.f, 0.f, 0.f, 0.f};

    // ---- transposed fragment addressing (slot 0): lane (r, gq) reads pixel rows 8 gq + (r >> 2) and + 4 of 16-channel block b
    const int r = lane & 15, gq = lane >> 4;
    const int frow = 8 * gq + (r >> 2);
    const int fr = (frow & 3) | (((frow >> 3) & 1) << 2);
    const int cb = (r & 3) * 8;
    unsigned abase[8], bbase[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) abase[i] = sbase + (unsigned)(frow * RB + (((wn * 8 + i) ^ fr) << 5) + cb);
#pragma unroll
    for (int j = 0; j < 4; ++j) bbase[j] = sbase + (unsigned)(IMG + frow * RB + (((wk * 4 + j) ^ fr) << 5) + cb);

    float* const dbias = Pk.dbias;
    const bool do_bias = (dbias != nullptr) && (kt == 0);
    const int brow = t / LPR, bpos = t % LPR;
    const int bf_ = (brow & 3) | (((brow >> 3) & 1) << 2);
    const int bchunk = (((bpos >> 1) ^ bf_) << 1) | (bpos & 1);
    float bsum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[e] = 0.f;
    if (do_bias && t < BN) sBias[t] = 0.f;

    issue(0, mbeg);
    if (nsteps > 1) issue(1, mbeg + MT);
    if (nsteps > 2) issue(2, mbeg + 2 * MT);
    if (nsteps > 2) wg_wait_vmcnt<2 * LPT>(); else if (nsteps > 1) wg_wait_vmcnt<LPT>(); else wg_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    wg_u32x2_t a0[4][2], a1[4][2], bE[4][2], bO[4][2];
#define WG9S_TR(F, i, ad) { F[i][0] = wg_tr<0>(ad); F[i][1] = wg_tr<2048>(ad); }
#pragma unroll
    for (int i = 0; i < 4; ++i) WG9S_TR(a0, i, abase[i]);
#pragma unroll
    for (int j = 0; j < 4; ++j) WG9S_TR(bE, j, bbase[j]);
    wg9s_wait16(a0, bE);
#define WG9S_MMA(AF, BF, i, ai, j)                                                                                                         \
    WgMma<F16>::run(make_uint4(AF[i][0].x, AF[i][0].y, AF[i][1].x, AF[i][1].y), make_uint4(BF[j][0].x, BF[j][0].y, BF[j][1].x, BF[j][1].y), acc[ai][j])
#define WG9S_SB __builtin_amdgcn_sched_barrier(0)
    // one stage; BC: this stage's B fragments, BN_: the set the next stage's are fetched into
#define WG9S_STAGE(BC, BN_)                                                                                                                \
    {                                                                                                                                      \
        const int rem = nsteps - 1 - it;                                                                                                   \
        const bool more = rem >= 3;                                                                                                        \
        const int s3 = (it + 3) & (SLOTS - 1), m3 = mbeg + (it + 3) * MT;                                                                  \
        const unsigned so = (unsigned)((it & (SLOTS - 1)) * SBYTES);                                                                       \
        const unsigned sn = (unsigned)(((it + 1) & (SLOTS - 1)) * SBYTES);                                                                 \
        if (do_bias) {                                                                                                                     \
            const unsigned char* sP = smem + so;                                                                                           \
            _Pragma("unroll")                                                                                                              \
            for (int jj = 0; jj < 2; ++jj) {                                                                                               \
                const uint4 v = *(const uint4*)(sP + (brow + jj * 16) * RB + bpos * 16);                                                   \
                const unsigned w4[4] = {v.x, v.y, v.z, v.w};                                                                               \
                _Pragma("unroll")                                                                                                          \
                for (int e = 0; e < 4; ++e) {                                                                                              \
                    bsum[2 * e] += WgMma<F16>::cvt((unsigned short)(w4[e] & 0xffffu));                                                     \
                    bsum[2 * e + 1] += WgMma<F16>::cvt((unsigned short)(w4[e] >> 16));                                                     \
                }                                                                                                                          \
            }                                                                                                                              \
        }                                                                                                                                  \
        WG9S_SB;                                                                                                                           \
        /* sub-step 0: A0 x B; fetch A1 (blocks 4-7) of this stage; the dy half of the DMA */                                              \
        WG9S_MMA(a0, BC, 0, 0, 0); WG9S_TR(a1, 0, abase[4] + so); WG9S_SB;                                                                 \
        WG9S_MMA(a0, BC, 1, 1, 0); WG9S_SB;                                                                                                \
        WG9S_MMA(a0, BC, 2, 2, 0); WG9S_TR(a1, 1, abase[5] + so); WG9S_SB;                                                                 \
        WG9S_MMA(a0, BC, 3, 3, 0); WG9S_SB;                                                                                                \
        WG9S_MMA(a0, BC, 0, 0, 1); WG9S_TR(a1, 2, abase[6] + so); WG9S_SB;                                                                 \
        WG9S_MMA(a0, BC, 1, 1, 1); WG9S_SB;                                                                                                \
        WG9S_MMA(a0, BC, 2, 2, 1); WG9S_TR(a1, 3, abase[7] + so); WG9S_SB;                                                                 \
        WG9S_MMA(a0, BC, 3, 3, 1); WG9S_SB;                                                                                                \
        WG9S_MMA(a0, BC, 0, 0, 2); WG9S_SB;                                                                                                \
        WG9S_MMA(a0, BC, 1, 1, 2); if (more) issueP(0, s3, m3); WG9S_SB;                                                                   \
        WG9S_MMA(a0, BC, 2, 2, 2); WG9S_SB;                                                                                                \
        WG9S_MMA(a0, BC, 3, 3, 2); WG9S_SB;                                                                                                \
        WG9S_MMA(a0, BC, 0, 0, 3); WG9S_SB;                                                                                                \
        WG9S_MMA(a0, BC, 1, 1, 3); if (more) issueP(1, s3, m3); WG9S_SB;                                                                   \
        WG9S_MMA(a0, BC, 2, 2, 3); WG9S_SB;                                                                                                \
        WG9S_MMA(a0, BC, 3, 3, 3); WG9S_SB;                                                                                                \
        if (rem >= 3) wg_wait_vmcnt<LPT + 2>(); else if (rem == 2) wg_wait_vmcnt<LPT>(); else wg_wait_vmcnt<0>();                          \
        wg9s_wait8(a1);                                                                                                                    \
        __builtin_amdgcn_s_barrier();                                                                                                      \
        WG9S_SB;                                                                                                                           \
        /* sub-step 1: A1 x B; fetch A0 and B of stage it + 1 (its slot has landed: the barrier above); the x half of the DMA */           \
        WG9S_MMA(a1, BC, 0, 4, 0); WG9S_TR(a0, 0, abase[0] + sn); WG9S_SB;                                                                 \
        WG9S_MMA(a1, BC, 1, 5, 0); WG9S_TR(a0, 1, abase[1] + sn); WG9S_SB;                                                                 \
        WG9S_MMA(a1, BC, 2, 6, 0); WG9S_TR(a0, 2, abase[2] + sn); WG9S_SB;                                                                 \
        WG9S_MMA(a1, BC, 3, 7, 0); WG9S_TR(a0, 3, abase[3] + sn); WG9S_SB;                                                                 \
        WG9S_MMA(a1, BC, 0, 4, 1); WG9S_TR(BN_, 0, bbase[0] + sn); WG9S_SB;                                                                \
        WG9S_MMA(a1, BC, 1, 5, 1); WG9S_SB;                                                                                                \
        WG9S_MMA(a1, BC, 2, 6, 1); WG9S_TR(BN_, 1, bbase[1] + sn); WG9S_SB;                                                                \
        WG9S_MMA(a1, BC, 3, 7, 1); WG9S_SB;                                                                                                \
        WG9S_MMA(a1, BC, 0, 4, 2); WG9S_TR(BN_, 2, bbase[2] + sn); WG9S_SB;                                                                \
        WG9S_MMA(a1, BC, 1, 5, 2); WG9S_SB;                                                                                                \
        WG9S_MMA(a1, BC, 2, 6, 2); WG9S_TR(BN_, 3, bbase[3] + sn); WG9S_SB;                                                                \
        WG9S_MMA(a1, BC, 3, 7, 2); WG9S_SB;                                                                                                \
        WG9S_MMA(a1, BC, 0, 4, 3); WG9S_SB;                                                                                                \
        WG9S_MMA(a1, BC, 1, 5, 3); if (more) issueQ(0, s3, m3); WG9S_SB;                                                                   \
        WG9S_MMA(a1, BC, 2, 6, 3); WG9S_SB;                                                                                                \
        WG9S_MMA(a1, BC, 3, 7, 3); if (more) issueQ(1, s3, m3); WG9S_SB;                                                                   \
        wg9s_wait16(a0, BN_);                                                                                                              \
        WG9S_SB;                                                                                                                           \
    }
    OCTA_STAMP_DECL;
    OCTA_STAMP_BEGIN;
    for (int it = 0; it < nsteps; ++it) {
        WG9S_STAGE(bE, bO)
        if (++it >= nsteps) break;
        WG9S_STAGE(bO, bE)
    }
#undef WG9S_STAGE
#undef WG9S_MMA
#undef WG9S_TR
#undef WG9S_SB
    OCTA_STAMP_END(octa_diag_stamps_wgrad9)

    float* const part = Pk.part ? Pk.part + (long)sp * Pk.part_slice : nullptr;
    if (do_bias) {
        __syncthreads();
        float* const bred = (float*)smem;                  // [512 / LPR][BN]
#pragma unroll
        for (int e = 0; e < 8; ++e) bred[brow * BN + bchunk * 8 + e] = bsum[e];
        __syncthreads();
        if (t < BN) {
            float v = 0.f;
            for (int rg = 0; rg < 512 / LPR; ++rg) v += bred[rg * BN + t];
            sBias[t] = v;
        }
        if (t < BN && n0 + t < Ng) {
            if (part) part[(long)groups * Ng * Kpad + g * Ng + n0 + t] = sBias[t];
            else atomicAdd(dbias + g * Ng + n0 + t, sBias[t]);
        }
    }
    // D[row = n (4 gq + e)][col = k (r)] of 16x16 block (i, j)
    if (part) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + wk * 64 + j * 16 + r;
            if (k >= Kpad) continue;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = n0 + wn * 128 + i * 16 + gq * 4 + e;
                    if (n < Ng) part[(long)(g * Ng + n) * Kpad + k] = acc[i][j][e];
                }
            }
        }
        return;
    }
    float* const dw = Pk.dw;
    const long s_o = Pk.s_o, s_i = Pk.s_i, s_h = Pk.s_h, s_w = Pk.s_w;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = k0 + wk * 64 + j * 16 + r;
        if (k >= Kpad) continue;
        const int tap = k / Cg, ci = k - tap * Cg;
        if (ci >= CgReal) continue;
        const int kh = tap / KW, kw = tap - kh * KW;
        const long koff = (long)ci * s_i + (long)kh * s_h + (long)kw * s_w;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = n0 + wn * 128 + i * 16 + gq * 4 + e;
                if (n < Ng) atomicAdd(dw + (long)(g * Ng + n) * s_o + koff, acc[i][j][e]);
            }
        }
    }
}

#ifdef OCTA_DIAG_STAMPS
// timeline of the wgrad9x workgroups (diagnostic build only): [workgroup][0] = items stamped, then per item 4 x s_memrealtime (100 MHz):
// item begin, main loop begin, main loop end, epilogue drained (the diagnostic build waits for its atomics there), and s_memtime (shader
// clock) at main loop begin / end
#define WG9X_MAXIT 12
__device__ unsigned long long octa_diag_timeline_wg9x[2048][1 + 6 * WG9X_MAXIT];
extern "C" int octa_diag_timeline_read_wgrad9x(void* host, int clear) {
    if (clear) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(octa_diag_timeline_wg9x)) != hipSuccess) return -3;
        return hipMemset(p, 0, sizeof(octa_diag_timeline_wg9x)) == hipSuccess ? 0 : -3;
    }
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(octa_diag_timeline_wg9x), sizeof(octa_diag_timeline_wg9x)) == hipSuccess ? 0 : -3;
}
#define WG9X_STAMP(k)                                                                                                                      \
    {                                                                                                                                      \
        unsigned long long r_;                                                                                                             \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r_) :: "memory");                                                   \
        if (threadIdx.x == 0 && blockIdx.x < 2048 && tl_item < WG9X_MAXIT) octa_diag_timeline_wg9x[blockIdx.x][1 + 6 * tl_item + (k)] = r_;  \
    }
#define WG9X_STAMP_CLK(k)                                                                                                                  \
    {                                                                                                                                      \
        unsigned long long r_;                                                                                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r_) :: "memory");                                                       \
        if (threadIdx.x == 0 && blockIdx.x < 2048 && tl_item < WG9X_MAXIT) octa_diag_timeline_wg9x[blockIdx.x][1 + 6 * tl_item + (k)] = r_;  \
    }
#else
#define WG9X_STAMP(k)
#define WG9X_STAMP_CLK(k)
#endif
struct WgSched { int len[8], off[8]; };   // blocks in XCD x's sequence; the position it starts at (a multiple of the slots per XCD)

template <int F16, int PERSIST>
__global__ __launch_bounds__(512) void wgrad9x_kernel(const WgBatch batch, const WgSched sched) {
    constexpr int BN = 256, BK = 256, MT = 32, SLOTS = 4;
    constexpr int RB = 512, IMG = MT * RB, SBYTES = 2 * IMG;
    constexpr int LPR = 32, RPI = 2, IPW = 2;
    constexpr int LPT = 2 * IPW;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SLOTS * SBYTES + BN * 4];
    float* const sBias = (float*)(smem + SLOTS * SBYTES);

    const int Lb = blockIdx.x, xcd = Lb & 7, SL = (int)(gridDim.x >> 3);
    const int nprob = batch.n, seqlen = sched.len[xcd], seqoff = sched.off[xcd];
    int pos = Lb >> 3;
#ifdef OCTA_DIAG_STAMPS
    int tl_item = 0;
#endif

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wn = wave >> 2, wk = wave & 3;
    const unsigned long zaddr = (unsigned long)(const void*)wg8_zero_page;
    const int drow = wave * RPI + lane / LPR;
    const int dpos = lane % LPR;
    const int df_ = (drow & 3) << 1;
    const int dchunk = (((dpos >> 1) ^ df_) << 1) | (dpos & 1);
    const unsigned sbase = wg_lds_addr(smem);
    const int r = lane & 15, gq = lane >> 4;
    const int frow = 8 * (gq >> 1) + (r >> 2);
    const int fr = ((r >> 2) & 3) << 1;
    const int cb = (r & 3) * 8;
    unsigned abase[4], bbase[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) abase[i] = sbase + (unsigned)(frow * RB + ((((wn * 4 + i) * 2 + (gq & 1)) ^ fr) << 5) + cb);
#pragma unroll
    for (int j = 0; j < 2; ++j) bbase[j] = sbase + (unsigned)(IMG + frow * RB + ((((wk * 2 + j) * 2 + (gq & 1)) ^ fr) << 5) + cb);
    const int brow = t / LPR, bpos = t % LPR;
    const int bf_ = (brow & 3) << 1;
    const int bchunk = (((bpos >> 1) ^ bf_) << 1) | (bpos & 1);

    wg_f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    for (;;) {
        // ---- the problem and block of position pos in this XCD's (cyclic) sequence: problem p contributes its x-th eighth
        if (pos >= seqlen) return;
        WG9X_STAMP(0)
        int q = pos + seqoff; q -= q >= seqlen ? seqlen : 0;
        pos += SL;
        int pi = 0, first = 0;
        for (; pi < nprob; ++pi) {
            const long NB = batch.p[pi].nblocks;
            first = (int)((NB * xcd) >> 3);
            const int cnt = (int)((NB * (xcd + 1)) >> 3) - first;
            if (q < cnt) break;
            q -= cnt;
        }
        if (pi >= nprob) return;               // (cannot happen: seqlen is the sum of the counts)
        const WgProb& Pk = batch.p[pi];
        const int tilesN = Pk.tilesN, tilesK = Pk.tilesK, groups = Pk.groups, mPerSplit = Pk.mPerSplit, Mtot = Pk.M;
        const int Ng = Pk.Ng, Kpad = Pk.Kpad, Cg = Pk.Cg, CgReal = Pk.CgReal, KW = Pk.KW;
        const int H = Pk.H, W = Pk.W, OH = Pk.OH, OW = Pk.OW, stride = Pk.stride, pad = Pk.pad;
        const int ldx = Pk.ldx, ldy = Pk.ldy;
        const unsigned short* const xbase = Pk.x;
        const unsigned short* const dybase = Pk.dy;
        int bid = first + q;
        const int nt = bid % tilesN; bid /= tilesN;
        const int kt = bid % tilesK; bid /= tilesK;
        const int g = bid % groups;
        const int sp = bid / groups;
        const int n0 = nt * BN, k0 = kt * BK;
        const int mbeg = sp * mPerSplit;
        const int mend = min(Mtot, mbeg + mPerSplit);
        const int nsteps = (mend - mbeg + MT - 1) / MT;
        if (nsteps <= 0) { if (PERSIST) continue; else return; }

        const bool pvalid = (n0 + dchunk * 8) < Ng;
        const unsigned long pstep = pvalid ? (unsigned long)((long)MT * ldy * 2) : 0ul;
        unsigned long pptr[IPW];
#pragma unroll
        for (int i = 0; i < IPW; ++i)
            pptr[i] = pvalid ? (unsigned long)(dybase + ((long)(mbeg + drow + i * 8 * RPI) * ldy + Pk.yoff + g * Ng + n0 + dchunk * 8)) : zaddr;
        const int kel = k0 + dchunk * 8;
        const bool kvalid = kel < Kpad;
        const int qtap = kel / Cg, qcc = kel - qtap * Cg;
        const int qkh = qtap / KW, qkw = qtap - qkh * KW;
        const int qdh = qkh - pad, qdw = qkw - pad;
        const unsigned long qbase = (unsigned long)(xbase + (Pk.xoff + g * CgReal + qcc));
        const bool plain = (Pk.KH == 1 && KW == 1 && pad == 0 && stride == 1);
        const int ldx2 = ldx * 2;
        const int dq = MT / OW, dr = MT - dq * OW;
        const int sdr = stride * dr, sdq = stride * dq, OWs = OW * stride, OHs = OH * stride;
        const int thrW = OWs + qdw, thrH = OHs + qdh;
        const int dpix = sdq * W + sdr, cW = stride * W - OWs, cH = H * W - OHs * W;
        int qih[IPW], qiw[IPW], qpix[IPW];
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int m = mbeg + drow + i * 8 * RPI;
            const int ow = m % OW, tq = m / OW, oh = tq % OH, b = tq / OH;
            qih[i] = oh * stride + qdh; qiw[i] = ow * stride + qdw;
            qpix[i] = plain ? m : (b * H + qih[i]) * W + qiw[i];
        }
        auto issueP = [&](int i, int slot, int mcur) {
            const bool ok = (mcur + drow + i * 8 * RPI) < mend;
            const unsigned long src = ok ? pptr[i] : zaddr;
            wg_glds16((const void*)src, __builtin_amdgcn_readfirstlane(sbase + (unsigned)(slot * SBYTES + (i * 8 + wave) * 1024)));
            pptr[i] += pstep;
        };
        auto issueQ = [&](int i, int slot, int mcur) {
            const bool ok = kvalid & ((mcur + drow + i * 8 * RPI) < mend) & ((unsigned)qih[i] < (unsigned)H) & ((unsigned)qiw[i] < (unsigned)W);
            const unsigned off = (unsigned)__mul24(qpix[i], ldx2);
            const unsigned long a = qbase + (unsigned long)off;
            const unsigned long src = ok ? a : zaddr;
            wg_glds16((const void*)src, __builtin_amdgcn_readfirstlane(sbase + (unsigned)(slot * SBYTES + IMG + (i * 8 + wave) * 1024)));
            qiw[i] += sdr;
            const bool c1 = qiw[i] >= thrW;
            qiw[i] -= c1 ? OWs : 0;
            qih[i] += sdq + (c1 ? stride : 0);
            const bool c2 = qih[i] >= thrH;
            qih[i] -= c2 ? OHs : 0;
            qpix[i] += dpix + (c1 ? cW : 0) + (c2 ? cH : 0);
        };
        auto issue = [&](int slot, int mcur) { issueP(0, slot, mcur); issueP(1, slot, mcur); issueQ(0, slot, mcur); issueQ(1, slot, mcur); };

        float* const dbias = Pk.dbias;
        const bool do_bias = (dbias != nullptr) && (kt == 0);
        float bsum[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bsum[e] = 0.f;

        issue(0, mbeg);
        if (nsteps > 1) issue(1, mbeg + MT);
        if (nsteps > 2) issue(2, mbeg + 2 * MT);
        if (nsteps > 2) wg_wait_vmcnt<2 * LPT>(); else if (nsteps > 1) wg_wait_vmcnt<LPT>(); else wg_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        wg_u32x2_t afX[4][2], bfX[2][2], afY[4][2], bfY[2][2];
        {
            unsigned aa[4], ba[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) aa[i] = abase[i];
#pragma unroll
            for (int j = 0; j < 2; ++j) ba[j] = bbase[j];
            wg9_load_half<0>(aa, ba, afX, bfX);
            wg9_wait_frags(afX, bfX);
        }
#define WG9_MMA(AF, BF, i, j)                                                                                                              \
    WgMma32<F16>::run(make_uint4(AF[i][0].x, AF[i][0].y, AF[i][1].x, AF[i][1].y), make_uint4(BF[j][0].x, BF[j][0].y, BF[j][1].x, BF[j][1].y), acc[i][j])
#define WG9_TR(F, i, HS, ad) { F[i][0] = wg_tr<HS * 8192>(ad); F[i][1] = wg_tr<HS * 8192 + 2048>(ad); }
#define WG9_SB __builtin_amdgcn_sched_barrier(0)
        WG9X_STAMP(1)
        WG9X_STAMP_CLK(4)
        for (int it = 0; it < nsteps; ++it) {
            const int rem = nsteps - 1 - it;
            const bool more = rem >= 3;
            const int s3 = (it + 3) & (SLOTS - 1), m3 = mbeg + (it + 3) * MT;
            const unsigned so = (unsigned)((it & (SLOTS - 1)) * SBYTES);
            const unsigned sn = (unsigned)(((it + 1) & (SLOTS - 1)) * SBYTES);
            if (do_bias) {
                const unsigned char* sP = smem + so;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const uint4 v = *(const uint4*)(sP + (brow + jj * 16) * RB + bpos * 16);
                    const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        bsum[2 * e] += WgMma<F16>::cvt((unsigned short)(w4[e] & 0xffffu));
                        bsum[2 * e + 1] += WgMma<F16>::cvt((unsigned short)(w4[e] >> 16));
                    }
                }
            }
            WG9_SB;
            WG9_MMA(afX, bfX, 0, 0); WG9_TR(afY, 0, 1, abase[0] + so); WG9_SB;
            WG9_MMA(afX, bfX, 1, 0); WG9_TR(afY, 1, 1, abase[1] + so); WG9_SB;
            WG9_MMA(afX, bfX, 2, 0); WG9_TR(afY, 2, 1, abase[2] + so); WG9_SB;
            WG9_MMA(afX, bfX, 3, 0); WG9_TR(afY, 3, 1, abase[3] + so); WG9_SB;
            WG9_MMA(afX, bfX, 0, 1); WG9_TR(bfY, 0, 1, bbase[0] + so); WG9_SB;
            WG9_MMA(afX, bfX, 1, 1); WG9_TR(bfY, 1, 1, bbase[1] + so); WG9_SB;
            WG9_MMA(afX, bfX, 2, 1); if (more) issueP(0, s3, m3); WG9_SB;
            WG9_MMA(afX, bfX, 3, 1); if (more) issueP(1, s3, m3); WG9_SB;
            if (rem >= 3) wg_wait_vmcnt<LPT + 2>(); else if (rem == 2) wg_wait_vmcnt<LPT>(); else wg_wait_vmcnt<0>();
            wg9_wait_frags(afY, bfY);
            __builtin_amdgcn_s_barrier();
            WG9_SB;
            WG9_MMA(afY, bfY, 0, 0); WG9_TR(afX, 0, 0, abase[0] + sn); WG9_SB;
            WG9_MMA(afY, bfY, 1, 0); WG9_TR(afX, 1, 0, abase[1] + sn); WG9_SB;
            WG9_MMA(afY, bfY, 2, 0); WG9_TR(afX, 2, 0, abase[2] + sn); WG9_SB;
            WG9_MMA(afY, bfY, 3, 0); WG9_TR(afX, 3, 0, abase[3] + sn); WG9_SB;
            WG9_MMA(afY, bfY, 0, 1); WG9_TR(bfX, 0, 0, bbase[0] + sn); WG9_SB;
            WG9_MMA(afY, bfY, 1, 1); WG9_TR(bfX, 1, 0, bbase[1] + sn); WG9_SB;
            WG9_MMA(afY, bfY, 2, 1); if (more) issueQ(0, s3, m3); WG9_SB;
            WG9_MMA(afY, bfY, 3, 1); if (more) issueQ(1, s3, m3); WG9_SB;
            wg9_wait_frags(afX, bfX);
            WG9_SB;
        }
#undef WG9_MMA
#undef WG9_TR
#undef WG9_SB
        WG9X_STAMP(2)
        WG9X_STAMP_CLK(5)

        float* const part = Pk.part ? Pk.part + (long)sp * Pk.part_slice : nullptr;
        if (do_bias) {
            __syncthreads();
            float* const bred = (float*)smem;                  // [512 / LPR][BN]
#pragma unroll
            for (int e = 0; e < 8; ++e) bred[brow * BN + bchunk * 8 + e] = bsum[e];
            __syncthreads();
            if (t < BN) {
                float v = 0.f;
                for (int rg = 0; rg < 512 / LPR; ++rg) v += bred[rg * BN + t];
                sBias[t] = v;
            }
            if (t < BN && n0 + t < Ng) {
                if (part) part[(long)groups * Ng * Kpad + g * Ng + n0 + t] = sBias[t];
                else atomicAdd(dbias + g * Ng + n0 + t, sBias[t]);
            }
        }
        if (part) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k = k0 + wk * 64 + j * 32 + (lane & 31);
                if (k >= Kpad) continue;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int n = n0 + wn * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                        if (n < Ng) part[(long)(g * Ng + n) * Kpad + k] = acc[i][j][e];
                    }
                }
            }
        } else {
            float* const dw = Pk.dw;
            const long s_o = Pk.s_o, s_i = Pk.s_i, s_h = Pk.s_h, s_w = Pk.s_w;
            const int lc = lane & 31, lh = lane >> 5;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k = k0 + wk * 64 + j * 32 + lc;
                if (k >= Kpad) continue;
                const int tap = k / Cg, ci = k - tap * Cg;
                if (ci >= CgReal) continue;
                const int kh = tap / KW, kw = tap - kh * KW;
                const long koff = (long)ci * s_i + (long)kh * s_h + (long)kw * s_w;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int n = n0 + wn * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                        if (n < Ng) atomicAdd(dw + (long)(g * Ng + n) * s_o + koff, acc[i][j][e]);
                    }
                }
            }
        }
#ifdef OCTA_DIAG_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WG9X_STAMP(3)
        if (threadIdx.x == 0 && blockIdx.x < 2048 && tl_item < WG9X_MAXIT) octa_diag_timeline_wg9x[blockIdx.x][0] = tl_item + 1;
        ++tl_item;
#endif
        if (!PERSIST) return;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        __syncthreads();            // every wave is done with the ring (fragments, bias scratch) before the next block's DMA lands in it
    }
}

// ------------------------------------------------------------------------------------------ host side
static unsigned wg_magic(int d) { return (unsigned)((1ull << 32) / (unsigned)d) + 1u; }
static int g_wg8_fold = 0;            // octa_tuning_set(4, 0 / 1)

// a problem whose M axis is split takes private slices for its partial tiles when the batch's fold session has room (conv.hip)
static void wg_take_fold(WgProb& p, hipStream_t st) {
    p.part = nullptr; p.part_slice = 0;
    // Off by default: measured +0.1 ms per step (profiles/r04_ab_wgrad8_fold.txt).  Unlike the few-channel layers, the M-splits of
    // these kernels add into DIFFERENT tiles' addresses at the full atomic rate, and the slices cost a write and a read more.
    // octa_tuning_set(4, 1) / OCTA_WGRAD8_FOLD=1 turns it on (tests, A/B runs).
    static const bool env_on = getenv("OCTA_WGRAD8_FOLD") != nullptr && atoi(getenv("OCTA_WGRAD8_FOLD")) != 0;
    if (!(env_on || g_wg8_fold || octa_deterministic()) || p.splitM < 2) return;
    const int64_t strides[4] = {p.s_o, p.s_i, p.s_h, p.s_w};
    int64_t slice = 0;
    float* ws = octa_wgrad_fold_reserve(st, p.dw, p.dbias, strides, p.groups * p.Ng, p.Kpad, p.Cg, p.CgReal, p.KW, p.splitM, &slice);
    if (ws) { p.part = ws; p.part_slice = (long)slice; }
}

static int g_wg8_min_ng = 128;        // octa_tuning_set(3, n): smallest Cout / groups the batched kernels accept
// can the batched kernel take this job?  (everything else goes to the single-problem kernels of conv.hip)
static bool wg8_eligible(const octa_wgrad_job& j) {
    const octa_conv_desc& d = j.d;
    if (d.dtype != OCTA_BF16 && d.dtype != OCTA_F16) return false;
    if (d.upshuffle || !j.x || !j.dy || !j.dw) return false;
    const int Ng = d.Cout / d.groups;
    if (Ng < g_wg8_min_ng) return false;                     // tall-skinny problems stay on the slab kernels
    if (d.OW < 2 || d.OH < 2 || d.OW > 511 || d.OH > 511 || 64 / d.OW + 1 > d.OH) return false;   // single-wrap pixel updates
    if ((int64_t)d.B * d.H * d.W >= (1 << 23) || d.ldx * 2 >= (1 << 23)) return false;              // 24-bit multiply operands
    const int64_t M = (int64_t)d.B * d.OH * d.OW;
    if (M >= (1 << 23) || M < 64) return false;
    if ((int64_t)d.B * d.H * d.W * d.ldx >= (1ll << 31) || M * d.ldy >= (1ll << 31)) return false;
    for (int i = 0; i < 4; ++i) if (j.dw_strides[i] >= (1ll << 31) || j.dw_strides[i] < 0) return false;
    if ((int64_t)d.Cout * j.dw_strides[0] >= (1ll << 31)) return false;
    if (d.ldy % 8 || d.yoff % 8 || d.ldx % 8 || d.xoff % 8) return false;
    if (d.groups > 1 && (Ng % 8 || (d.Cin / d.groups) % 8)) return false;
    return true;
}

struct WgPlan { WgProb p; int64_t steps; int variant; };

template <int F16>
static int wg8_launch(std::vector<WgPlan>& plans, int variant, hipStream_t st) {
    // steps per workgroup: as few splits as still give every CU ~3 workgroups, never fewer than 8 stages of 64 pixels each
    size_t i0 = 0;
    while (i0 < plans.size()) {
        const size_t i1 = std::min(plans.size(), i0 + (size_t)WG_MAXP);
        int64_t tiles = 0, maxsteps = 1;
        for (size_t i = i0; i < i1; ++i) { tiles += (int64_t)plans[i].p.tilesN * plans[i].p.tilesK * plans[i].p.groups; maxsteps = std::max(maxsteps, plans[i].steps); }
        // steps per workgroup S: minimise  rounds(S) * (S + E)  with rounds = ceil(workgroups / 256 CUs) (one workgroup per CU) and
        // E ~ the prologue + atomic epilogue of a workgroup in units of a 64-pixel stage
        static const int E = getenv("OCTA_WG8_EPI") ? atoi(getenv("OCTA_WG8_EPI")) : 6;
        static const int minsteps = getenv("OCTA_WG8_MINSTEPS") ? atoi(getenv("OCTA_WG8_MINSTEPS")) : 4;
        auto blocks_at = [&](int64_t s) { int64_t b = 0; for (size_t i = i0; i < i1; ++i) b += (int64_t)plans[i].p.tilesN * plans[i].p.tilesK * plans[i].p.groups * ((plans[i].steps + s - 1) / s); return b; };
        int64_t S = maxsteps, best = -1;
        for (int64_t s = maxsteps; s >= minsteps && !octa_deterministic(); s = (s > 64 ? s - s / 32 : s - 1)) {      // (deterministic mode: no M-split, one workgroup per output tile)
            const int64_t rounds = (blocks_at(s) + 255) / 256;
            const int64_t cost = rounds * (s + E);
            if (best < 0 || cost < best) { best = cost; S = s; }
        }
        (void)tiles;
        // long workgroups first: the tail of the launch is then made of short ones
        std::stable_sort(plans.begin() + i0, plans.begin() + i1, [&](const WgPlan& a, const WgPlan& b) { return std::min(a.steps, S) > std::min(b.steps, S); });
        WgBatch batch;
        batch.n = (int)(i1 - i0);
        int64_t nblk = 0;
        for (size_t i = i0; i < i1; ++i) {
            WgProb& p = plans[i].p;
            const int64_t split = (plans[i].steps + S - 1) / S;
            const int64_t sps = (plans[i].steps + split - 1) / split;      // balanced stages per split
            p.mPerSplit = (int)(sps * 64);
            p.splitM = (int)((p.M + p.mPerSplit - 1) / p.mPerSplit);
            wg_take_fold(p, st);
            p.blockStart = (int)nblk;
            nblk += (int64_t)p.tilesN * p.tilesK * p.groups * p.splitM;
            batch.p[i - i0] = p;
        }
        if (nblk <= 0 || nblk >= (1ll << 30)) OCTA_FAIL(OCTA_ERR_BAD_ARG, "octa_conv2d_wgrad_batch: bad grid %lld", (long long)nblk);
        static const bool stagger = getenv("OCTA_WG8_NOSTAGGER") == nullptr;
        if (stagger) {
            if (variant == 0) wgrad8_kernel<4, 2, F16, 1><<<(unsigned)nblk, 512, 0, st>>>(batch);
            else wgrad8_kernel<2, 4, F16, 1><<<(unsigned)nblk, 512, 0, st>>>(batch);
        } else {
            if (variant == 0) wgrad8_kernel<4, 2, F16, 0><<<(unsigned)nblk, 512, 0, st>>>(batch);
            else wgrad8_kernel<2, 4, F16, 0><<<(unsigned)nblk, 512, 0, st>>>(batch);
        }
        OCTA_CHECK_LAUNCH("wgrad8");
        octa_note_conv_kernel(variant == 0 ? (F16 ? "wgrad8_kernel<f16,256x128>" : "wgrad8_kernel<bf16,256x128>")
                                           : (F16 ? "wgrad8_kernel<f16,128x256>" : "wgrad8_kernel<bf16,128x256>"));
        i0 = i1;
    }
    return OCTA_OK;
}

static int g_wg9_shape16 = 0;          // octa_tuning_set(9, 0 / 1): 1 = the 256 x 256 kernel on v_mfma_f32_16x16x32 (wgrad9s)
static int wg9_shape16() {
    static const int env = getenv("OCTA_WG9_SHAPE16") ? atoi(getenv("OCTA_WG9_SHAPE16")) : -1;
    return env >= 0 ? env : g_wg9_shape16;
}
static int g_wg9_ablate = 0;           // octa_tuning_set(2, mask): timing-only ablation build of wgrad9 (tools/wgrad_micro.py)
// 256 x 256 tiles (wgrad9): one launch per batch, 32-pixel stages
template <int F16>
static int wg9_launch(std::vector<WgPlan>& plans, hipStream_t st) {
    size_t i0 = 0;
    while (i0 < plans.size()) {
        const size_t i1 = std::min(plans.size(), i0 + (size_t)WG_MAXP);
        int64_t maxsteps = 1;
        for (size_t i = i0; i < i1; ++i) maxsteps = std::max(maxsteps, plans[i].steps);
        // stages per workgroup S: minimise rounds(S) * (S + E), rounds = ceil(workgroups / 256 CUs), E ~ prologue + 256 KB of float
        // atomics per workgroup in units of a 32-pixel stage
        static const int E = getenv("OCTA_WG9_EPI") ? atoi(getenv("OCTA_WG9_EPI")) : 40;
        static const int minsteps = getenv("OCTA_WG9_MINSTEPS") ? atoi(getenv("OCTA_WG9_MINSTEPS")) : 24;
        auto blocks_at = [&](int64_t s) { int64_t b = 0; for (size_t i = i0; i < i1; ++i) b += (int64_t)plans[i].p.tilesN * plans[i].p.tilesK * plans[i].p.groups * ((plans[i].steps + s - 1) / s); return b; };
        int64_t S = maxsteps, best = -1;
        for (int64_t s = maxsteps; s >= minsteps && !octa_deterministic(); s = (s > 64 ? s - s / 32 : s - 1)) {      // (deterministic mode: no M-split)
            const int64_t rounds = (blocks_at(s) + 255) / 256;
            const int64_t cost = rounds * (s + E);
            if (best < 0 || cost < best) { best = cost; S = s; }
        }
        std::stable_sort(plans.begin() + i0, plans.begin() + i1, [&](const WgPlan& a, const WgPlan& b) { return std::min(a.steps, S) > std::min(b.steps, S); });
        WgBatch batch;
        batch.n = (int)(i1 - i0);
        int64_t nblk = 0;
        for (size_t i = i0; i < i1; ++i) {
            WgProb& p = plans[i].p;
            const int64_t split = (plans[i].steps + S - 1) / S;
            const int64_t sps = (plans[i].steps + split - 1) / split;      // balanced stages per split
            p.mPerSplit = (int)(sps * 32);
            p.splitM = (int)((p.M + p.mPerSplit - 1) / p.mPerSplit);
            wg_take_fold(p, st);
            p.blockStart = (int)nblk;
            nblk += (int64_t)p.tilesN * p.tilesK * p.groups * p.splitM;
            batch.p[i - i0] = p;
        }
        if (nblk <= 0 || nblk >= (1ll << 30)) OCTA_FAIL(OCTA_ERR_BAD_ARG, "octa_conv2d_wgrad_batch: bad grid %lld", (long long)nblk);
        static const bool logit = getenv("OCTA_WG_LOG") != nullptr;      // tools/wgrad_sched_sim.py reads these lines
        if (logit) {
            fprintf(stderr, "wg9 launch S %lld nblk %lld :", (long long)S, (long long)nblk);
            for (int i = 0; i < batch.n; ++i) {
                const WgProb& p = batch.p[i];
                fprintf(stderr, " [Ng %d Kpad %d M %d tiles %d steps %d split %d]", p.Ng, p.Kpad, p.M, p.tilesN * p.tilesK * p.groups, (p.M + 31) / 32, p.splitM);
            }
            fprintf(stderr, "\n");
        }
        static const bool stagger = getenv("OCTA_WG8_NOSTAGGER") == nullptr;
        if (g_wg9_ablate && !F16) {
            switch (g_wg9_ablate) {
                case 1: wgrad9_kernel<0, 1, 1><<<(unsigned)nblk, 512, 0, st>>>(batch); break;
                case 2: wgrad9_kernel<0, 1, 2><<<(unsigned)nblk, 512, 0, st>>>(batch); break;
                case 4: wgrad9_kernel<0, 1, 4><<<(unsigned)nblk, 512, 0, st>>>(batch); break;
                case 8: wgrad9_kernel<0, 1, 8><<<(unsigned)nblk, 512, 0, st>>>(batch); break;
                case 3: wgrad9_kernel<0, 1, 3><<<(unsigned)nblk, 512, 0, st>>>(batch); break;
                case 11: wgrad9_kernel<0, 1, 11><<<(unsigned)nblk, 512, 0, st>>>(batch); break;
                case 7: wgrad9_kernel<0, 1, 7><<<(unsigned)nblk, 512, 0, st>>>(batch); break;
                case 16: wgrad9_kernel<0, 1, 16><<<(unsigned)nblk, 512, 0, st>>>(batch); break;
                case 21: wgrad9_kernel<0, 1, 21><<<(unsigned)nblk, 512, 0, st>>>(batch); break;
                case 13: wgrad9_kernel<0, 1, 13><<<(unsigned)nblk, 512, 0, st>>>(batch); break;
                case 29: wgrad9_kernel<0, 1, 29><<<(unsigned)nblk, 512, 0, st>>>(batch); break;
                default: OCTA_FAIL(OCTA_ERR_BAD_ARG, "wgrad9 ablation %d is not built", g_wg9_ablate);
            }
        } else if (wg9_shape16() == 2) wgrad9a_kernel<F16><<<(unsigned)nblk, 256, 0, st>>>(batch);
        else if (wg9_shape16()) wgrad9s_kernel<F16><<<(unsigned)nblk, 512, 0, st>>>(batch);
        else if (stagger) wgrad9_kernel<F16, 1><<<(unsigned)nblk, 512, 0, st>>>(batch);
        else wgrad9_kernel<F16, 0><<<(unsigned)nblk, 512, 0, st>>>(batch);
        OCTA_CHECK_LAUNCH("wgrad9");
        octa_note_conv_kernel(wg9_shape16() == 2 ? (F16 ? "wgrad9a_kernel<f16,256x256,4 waves>" : "wgrad9a_kernel<bf16,256x256,4 waves>")
                              : wg9_shape16() ? (F16 ? "wgrad9s_kernel<f16,256x256,16x16x32>" : "wgrad9s_kernel<bf16,256x256,16x16x32>")
                                            : (F16 ? "wgrad9_kernel<f16,256x256>" : "wgrad9_kernel<bf16,256x256>"));
        i0 = i1;
    }
    return OCTA_OK;
}

// ---- wgrad9x: per-class splits, XCD-interleaved sequences, optionally persistent (kernel comment above)
static int g_wg9_sched = 0;             // octa_tuning_set(8, mode): 0 = wgrad9 (rounds of one split length), 1 = wgrad9x one block per workgroup, 2 = wgrad9x persistent
static int wg9_sched_mode() {
    static const int env = getenv("OCTA_WG9_SCHED") ? atoi(getenv("OCTA_WG9_SCHED")) : -1;
    return env >= 0 ? env : g_wg9_sched;
}
static int wg_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}
#include "wgrad2d.hpp"
// Stage-units until the last XCD is done.  Every XCD runs its own sequence (problem p contributes the x-th eighth of its nb[p] blocks
// of len[p] stages, + E for prologue and epilogue) on SL slots: dealt statically (persistent) or to the slot that frees first.
static int64_t wg9x_model(const std::vector<int64_t>& nb, const std::vector<int64_t>& len, const int* off, int SL, bool persist, int E) {
    int64_t worst = 0;
    std::vector<int64_t> seq, slot((size_t)SL);
    for (int x = 0; x < 8; ++x) {
        seq.clear();
        for (size_t p = 0; p < nb.size(); ++p) {
            const int64_t cnt = ((nb[p] * (x + 1)) >> 3) - ((nb[p] * x) >> 3);
            seq.insert(seq.end(), (size_t)cnt, len[p] + E);
        }
        if (seq.empty()) continue;
        if (off && off[x] > 0 && off[x] < (int)seq.size()) std::rotate(seq.begin(), seq.begin() + off[x], seq.end());
        std::fill(slot.begin(), slot.end(), 0);
        if (persist) {
            for (size_t i = 0; i < seq.size(); ++i) slot[i % SL] += seq[i];
        } else {
            // list scheduling; the heap is the SL finish times
            std::make_heap(slot.begin(), slot.end(), std::greater<int64_t>());
            for (size_t i = 0; i < seq.size(); ++i) {
                std::pop_heap(slot.begin(), slot.end(), std::greater<int64_t>());
                slot.back() += seq[i];
                std::push_heap(slot.begin(), slot.end(), std::greater<int64_t>());
            }
        }
        worst = std::max(worst, *std::max_element(slot.begin(), slot.end()));
    }
    return worst;
}
static void wg9x_offsets(const std::vector<int64_t>& nb, int SL, int rot, int* len8, int* off8) {
    for (int x = 0; x < 8; ++x) {
        int64_t L = 0;
        for (size_t p = 0; p < nb.size(); ++p) L += ((nb[p] * (x + 1)) >> 3) - ((nb[p] * x) >> 3);
        len8[x] = (int)L;
        const int64_t rounds = (L + SL - 1) / SL;
        off8[x] = rot ? (int)(((rounds * x) / 8) * SL) : 0;     // XCD x starts x/8 of the way into its rounds
        if (off8[x] >= len8[x]) off8[x] = 0;
    }
}

template <int F16>
static int wg9x_launch(std::vector<WgPlan>& plans, hipStream_t st, int mode) {
    static const int E = getenv("OCTA_WG9X_EPI") ? atoi(getenv("OCTA_WG9X_EPI")) : 16;
    static const int minlen = getenv("OCTA_WG9X_MINLEN") ? atoi(getenv("OCTA_WG9X_MINLEN")) : 24;
    static const int rot = getenv("OCTA_WG9X_ROT") ? atoi(getenv("OCTA_WG9X_ROT")) : 1;
    static const bool logit = getenv("OCTA_WG_LOG") != nullptr;
    const bool persist = mode == 2;
    const int SLmax = std::max(1, wg_num_cus() / 8);
    size_t i0 = 0;
    while (i0 < plans.size()) {
        const size_t i1 = std::min(plans.size(), i0 + (size_t)WG_MAXP);
        const size_t n = i1 - i0;
        // long pixel axes first; problems with the same number of stages form a class and share a split
        std::stable_sort(plans.begin() + i0, plans.begin() + i1, [](const WgPlan& a, const WgPlan& b) { return a.steps > b.steps; });
        std::vector<int> cls(n), cfirst;
        for (size_t i = 0; i < n; ++i) {
            if (i == 0 || plans[i0 + i].steps != plans[i0 + i - 1].steps) cfirst.push_back((int)i);
            cls[i] = (int)cfirst.size() - 1;
        }
        const int nc = (int)cfirst.size();
        std::vector<int64_t> tiles(n), nb(n), len(n);
        for (size_t i = 0; i < n; ++i) tiles[i] = (int64_t)plans[i0 + i].p.tilesN * plans[i0 + i].p.tilesK * plans[i0 + i].p.groups;
        std::vector<int> split((size_t)nc, 1);
        int len8[8], off8[8];
        auto cost = [&]() {
            for (size_t i = 0; i < n; ++i) {
                const int64_t sp = split[cls[i]], steps = plans[i0 + i].steps;
                const int64_t sps = (steps + sp - 1) / sp;
                nb[i] = tiles[i] * ((steps + sps - 1) / sps);
                len[i] = sps;
            }
            wg9x_offsets(nb, SLmax, rot, len8, off8);
            return wg9x_model(nb, len, off8, SLmax, persist, E);
        };
        // the search is remembered per batch signature (a training step asks for the same few batches every time)
        static std::mutex memo_mu;
        static std::map<std::vector<int64_t>, std::vector<int>> memo;
        std::vector<int64_t> key{(int64_t)persist, (int64_t)SLmax, (int64_t)E, (int64_t)minlen, (int64_t)rot, (int64_t)octa_deterministic()};
        for (size_t i = 0; i < n; ++i) { key.push_back(tiles[i]); key.push_back(plans[i0 + i].steps); }
        bool known = false;
        {
            std::lock_guard<std::mutex> lk(memo_mu);
            auto it = memo.find(key);
            if (it != memo.end()) { split = it->second; known = true; }
        }
        if (!known && !octa_deterministic()) {          // (deterministic mode: no M-split, one block per output tile)
            // start from the best common target length (every class split to at most that many stages), then move one class at a time
            int64_t best = cost();
            std::vector<int> bsplit = split;
            std::vector<int64_t> targets;
            for (int c = 0; c < nc; ++c) {
                const int64_t steps = plans[i0 + cfirst[c]].steps;
                for (int64_t k = 1; k <= 64 && steps / k >= minlen; ++k) targets.push_back((steps + k - 1) / k);
            }
            std::sort(targets.begin(), targets.end());
            targets.erase(std::unique(targets.begin(), targets.end()), targets.end());
            for (int64_t Lt : targets) {
                for (int c = 0; c < nc; ++c) {
                    const int64_t steps = plans[i0 + cfirst[c]].steps;
                    split[c] = (int)std::max<int64_t>(1, std::min<int64_t>((steps + Lt - 1) / Lt, std::max<int64_t>(1, std::min<int64_t>(64, steps / minlen))));
                }
                const int64_t v = cost();
                if (v < best) { best = v; bsplit = split; }
            }
            split = bsplit;
            for (int pass = 0; pass < 3; ++pass) {
                bool moved = false;
                for (int c = 0; c < nc; ++c) {
                    const int64_t steps = plans[i0 + cfirst[c]].steps;
                    const int smax = (int)std::max<int64_t>(1, std::min<int64_t>(64, steps / minlen));
                    int bs = split[c];
                    for (int sp = 1; sp <= smax; ++sp) {
                        split[c] = sp;
                        const int64_t v = cost();
                        if (v < best) { best = v; bs = sp; moved = true; }
                    }
                    split[c] = bs;
                }
                if (!moved) break;
            }
        }
        if (!known) { std::lock_guard<std::mutex> lk(memo_mu); memo[key] = split; }
        (void)cost();                          // nb / len / len8 / off8 of the chosen splits
        WgBatch batch;
        WgSched sched;
        batch.n = (int)n;
        int maxlen = 0;
        for (int x = 0; x < 8; ++x) { sched.len[x] = len8[x]; sched.off[x] = off8[x]; maxlen = std::max(maxlen, len8[x]); }
        for (size_t i = 0; i < n; ++i) {
            WgProb& p = plans[i0 + i].p;
            p.mPerSplit = (int)(len[i] * 32);
            p.splitM = (int)((p.M + p.mPerSplit - 1) / p.mPerSplit);
            p.nblocks = (int)nb[i];
            if ((int64_t)p.tilesN * p.tilesK * p.groups * p.splitM != nb[i]) OCTA_FAIL(OCTA_ERR_BAD_ARG, "octa_conv2d_wgrad_batch: internal: block count of problem %d", (int)i);
            wg_take_fold(p, st);
            p.blockStart = 0;
            batch.p[i] = p;
        }
        if (maxlen <= 0 || maxlen >= (1 << 27)) OCTA_FAIL(OCTA_ERR_BAD_ARG, "octa_conv2d_wgrad_batch: bad grid %d", maxlen);
        const int SL = persist ? std::min(SLmax, maxlen) : maxlen;
        if (persist && SL < SLmax) {           // fewer blocks than slots: the start positions were computed for SLmax slots
            std::vector<int64_t> nbv(nb.begin(), nb.end());
            wg9x_offsets(nbv, SL, rot, sched.len, sched.off);
        }
        if (logit) {
            fprintf(stderr, "wg9x launch mode %d slots %d model %lld :", mode, SL, (long long)wg9x_model(nb, len, sched.off, persist ? SL : SLmax, persist, E));
            for (size_t i = 0; i < n; ++i) fprintf(stderr, " [tiles %lld steps %lld split %d len %lld]", (long long)tiles[i], (long long)plans[i0 + i].steps, batch.p[i].splitM, (long long)len[i]);
            fprintf(stderr, "\n");
        }
        if (persist) wgrad9x_kernel<F16, 1><<<(unsigned)(8 * SL), 512, 0, st>>>(batch, sched);
        else wgrad9x_kernel<F16, 0><<<(unsigned)(8 * SL), 512, 0, st>>>(batch, sched);
        OCTA_CHECK_LAUNCH("wgrad9x");
        octa_note_conv_kernel(persist ? (F16 ? "wgrad9x_kernel<f16,256x256,persistent>" : "wgrad9x_kernel<bf16,256x256,persistent>")
                                      : (F16 ? "wgrad9x_kernel<f16,256x256>" : "wgrad9x_kernel<bf16,256x256>"));
        i0 = i1;
    }
    return OCTA_OK;
}

static int g_wgrad_families = 3;     // bit 0: 256x128 / 128x256 tiles (wgrad8), bit 1: 256x256 tiles (wgrad9); octa_tuning_set(1, mask)
void octa_set_deterministic(int on);   // api.cpp
void octa_set_halo8_packed(int on);    // conv.hip
void octa_set_rev_walk(int on);        // api.cpp
extern "C" int octa_tuning_set(int key, int value) {
    if (key == 10) { OCTA_REQUIRE(value >= 0 && value <= 2, "octa_tuning_set: key 10 = 3x3 stride-1 weight gradients of exact 5 x 25 geometries on the 2-D patch kernel (0 never / 1 every eligible geometry / 2 where it pays)"); g_wg2d = value; return OCTA_OK; }
    if (key == 9) { OCTA_REQUIRE(value >= 0 && value <= 2, "octa_tuning_set: key 9 = MFMA shape of the 256 x 256 weight-gradient kernel: 0 = 32x32x16 (wgrad9), 1 = 16x16x32 (wgrad9s), 2 = 32x32x16 with four waves of 128 x 128 (wgrad9a)"); g_wg9_shape16 = value; return OCTA_OK; }
    if (key == 8) { OCTA_REQUIRE(value >= 0 && value <= 2, "octa_tuning_set: key 8 = wgrad9 schedule: 0 rounds of one split length, 1 per-class splits + XCD-interleaved sequences, 2 the same, persistent"); g_wg9_sched = value; return OCTA_OK; }
    if (key == 7) { OCTA_REQUIRE(value == 0 || value == 1, "octa_tuning_set: key 7 = first-pass reductions walk their tensor end first (0 / 1)"); octa_set_rev_walk(value); return OCTA_OK; }
    if (key == 6) { OCTA_REQUIRE(value == 0 || value == 1, "octa_tuning_set: key 6 = halo8 patch image: 1 packed (bank-conflict-free), 0 linear"); octa_set_halo8_packed(value); return OCTA_OK; }
    if (key == 5) { OCTA_REQUIRE(value == 0 || value == 1, "octa_tuning_set: key 5 = deterministic mode (0 / 1)"); octa_set_deterministic(value); return OCTA_OK; }
    if (key == 2) { g_wg9_ablate = value; return OCTA_OK; }
    if (key == 4) { OCTA_REQUIRE(value == 0 || value == 1, "octa_tuning_set: key 4 = partial tiles + fold for the batched weight-gradient kernels (0 / 1)"); g_wg8_fold = value; return OCTA_OK; }
    if (key == 3) { OCTA_REQUIRE(value >= 8 && value <= 4096, "octa_tuning_set: key 3 = minimum Cout / groups of the batched weight-gradient kernels"); g_wg8_min_ng = value; return OCTA_OK; }
    OCTA_REQUIRE(key == 1 && value >= 1 && value <= 3, "octa_tuning_set: key 1 = batched weight-gradient tile families (mask 1..3)");
    g_wgrad_families = value;
    return OCTA_OK;
}

static int wg8_variant(const octa_wgrad_job& j) {
    const int Ng = j.d.Cout / j.d.groups, Kpad = j.d.KH * j.d.KW * j.d.cin_g_pad;
    // 256 x 256 tiles when they waste at most 15 % of their MFMA work on padding (variant 2)
    const int64_t w2 = (int64_t)cdiv(Ng, 256) * 256 * cdiv(Kpad, 256) * 256;
    if ((g_wgrad_families & 2) && (w2 * 100 <= (int64_t)Ng * Kpad * 115 || !(g_wgrad_families & 1))) return 2;
    // slab orientation: the one that wastes less padded MFMA work; ties go to 256(N) x 128(K)
    const int64_t w0 = (int64_t)cdiv(Ng, 256) * 256 * cdiv(Kpad, 128) * 128, w1 = (int64_t)cdiv(Ng, 128) * 128 * cdiv(Kpad, 256) * 256;
    return (w1 < w0) ? 1 : 0;
}
extern "C" size_t octa_wgrad_job_class(const octa_wgrad_job* job) {
    static const bool off = getenv("OCTA_NO_WGRAD8") != nullptr;
    if (!job || off) return 0;
    if (wg2d_eligible(*job)) return 4;            // the 2-D patch kernel: its own flush group (one launch per job)
    if (!wg8_eligible(*job)) return 0;
    return 1 + wg8_variant(*job);
}

extern "C" int octa_conv2d_wgrad_batch(const octa_wgrad_job* jobs, int n, float* ws, int64_t ws_bytes, octa_stream_t stream) {
    OCTA_REQUIRE(jobs != nullptr && n >= 0, "octa_conv2d_wgrad_batch: bad arguments");
    OCTA_REQUIRE(ws_bytes >= 0 && (ws || ws_bytes == 0) && ((uintptr_t)ws & 15) == 0, "octa_conv2d_wgrad_batch: ws must be a 16-byte aligned buffer of ws_bytes, or NULL / 0");
    static const bool off = getenv("OCTA_NO_WGRAD8") != nullptr;
    hipStream_t st = (hipStream_t)stream;
    std::vector<WgPlan> plans[2][3];   // [f16][variant]
    // the per-layer jobs of the batch share one fold session: their partial tiles are summed by ONE fold launch (conv.hip)
    struct FoldSession {
        bool mine;
        FoldSession(hipStream_t s, float* w, int64_t nfl) : mine(octa_wgrad_fold_begin(s, w, nfl)) {}
        ~FoldSession() { if (mine) octa_wgrad_fold_end(); }
        int close() { const bool m = mine; mine = false; return m ? octa_wgrad_fold_end() : OCTA_OK; }
    } fold(st, ws_bytes > 0 ? ws : nullptr, ws_bytes / 4);
    std::vector<const octa_wgrad_job*> patch2d[2];          // [f16]: the jobs of the 2-D patch kernel, one launch per four
    for (int i = 0; i < n; ++i) {
        const octa_wgrad_job& j = jobs[i];
        if (!off && wg2d_eligible(j)) { patch2d[j.d.dtype == OCTA_F16 ? 1 : 0].push_back(&j); continue; }
        if (off || !wg8_eligible(j)) {
            const int rc = octa_conv2d_wgrad(&j.d, j.x, j.dy, j.dw, j.dw_strides, j.dbias, stream);
            if (rc) return rc;
            continue;
        }
        const octa_conv_desc& d = j.d;
        WgPlan pl;
        WgProb& p = pl.p;
        p.x = (const unsigned short*)j.x; p.dy = (const unsigned short*)j.dy; p.dw = j.dw; p.dbias = j.dbias;
        p.H = d.H; p.W = d.W; p.OH = d.OH; p.OW = d.OW;
        p.Cg = d.cin_g_pad; p.CgReal = d.Cin / d.groups; p.Ng = d.Cout / d.groups;
        p.KH = d.KH; p.KW = d.KW; p.stride = d.stride; p.pad = d.pad;
        p.ldx = d.ldx; p.xoff = d.xoff; p.ldy = d.ldy; p.yoff = d.yoff;
        p.M = d.B * d.OH * d.OW; p.Kpad = d.KH * d.KW * p.Cg;
        p.s_o = (int)j.dw_strides[0]; p.s_i = (int)j.dw_strides[1]; p.s_h = (int)j.dw_strides[2]; p.s_w = (int)j.dw_strides[3];
        p.magicOW = wg_magic(d.OW); p.magicOH = wg_magic(d.OH);
        p.groups = d.groups;
        pl.variant = wg8_variant(j);
        p.tilesN = cdiv(p.Ng, pl.variant == 1 ? 128 : 256);
        p.tilesK = cdiv(p.Kpad, pl.variant == 0 ? 128 : 256);
        pl.steps = pl.variant == 2 ? (p.M + 31) / 32 : (p.M + 63) / 64;
        p.splitM = 1; p.mPerSplit = 0; p.blockStart = 0; p.part = nullptr; p.part_slice = 0;
        plans[d.dtype == OCTA_F16 ? 1 : 0][pl.variant].push_back(pl);
    }
    for (int f = 0; f < 2; ++f)
        if (!patch2d[f].empty()) { const int rc = wg2d_launch(patch2d[f].data(), (int)patch2d[f].size(), f, st); if (rc) return rc; }
    for (int v = 0; v < 2; ++v) {
        if (!plans[0][v].empty()) { const int rc = wg8_launch<0>(plans[0][v], v, st); if (rc) return rc; }
        if (!plans[1][v].empty()) { const int rc = wg8_launch<1>(plans[1][v], v, st); if (rc) return rc; }
    }
    const int sched_mode = wg9_sched_mode();
    if (!plans[0][2].empty()) { const int rc = sched_mode ? wg9x_launch<0>(plans[0][2], st, sched_mode) : wg9_launch<0>(plans[0][2], st); if (rc) return rc; }
    if (!plans[1][2].empty()) { const int rc = sched_mode ? wg9x_launch<1>(plans[1][2], st, sched_mode) : wg9_launch<1>(plans[1][2], st); if (rc) return rc; }
    return fold.close();          // ONE fold launch (per 16 jobs) behind every kernel of the batch that stored partial tiles
}
