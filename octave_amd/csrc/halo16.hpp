// The 2-D patch 3x3 kernel of halo8.hpp with v_mfma_f32_16x16x32 instead of 32x32x16 (round 5), same tile (256 patch pixels x 128
// output channels), same 64 x 64 per wave, same LDS image, DMA roles, weight ring and barriers; included by conv.hip behind halo8.hpp.
//
// Why.  halo8 keeps the matrix pipe busy 64-70 % of the cycles and the chip answers with its clock (1.6 GHz instead of 2.4:
// profiles/r04_halo8_pmc.txt, r05_clock_stamps.txt).  MI355X_MICROARCH.md "DVFS give-back" item 7 / cdna_hip_programming.md rule 28:
// at equal cycles per FLOP the chip holds a higher clock on the 16x16x32 shape (1.12-1.15x the FLOP/s of 32x32x16 in bare loops with
// operands re-read from LDS).  Same LDS bytes per FLOP: a k32-step of a wave is 4 pixel + 4 weight ds_read_b128 for 16 MFMAs.
//
// Pipeline (per wave; a stage = one tap of one 64-channel slice = TWO k32-steps of 16 MFMAs; two fragment sets X / Y of 8 reads):
//     step 0: 16 MFMAs on X = (st, 0); behind the first eight, one by one, the reads of (st, 1) -> Y
//     s_waitcnt lgkmcnt(0): Y is there and every LDS read of stage st has returned; counted vmcnt: this wave's part of B(st + 1)
//             [and of the next slice's patch] has landed; ONE raw s_barrier: ... for every wave, and ring slot tap % 3 is free
//     step 1: 16 MFMAs on Y; behind the first eight the reads of (st + 1, 0) -> X, behind the next four the DMA instructions of
//             B(st + 3) and (taps 2..4) a pair of the next slice's patch
//     s_waitcnt lgkmcnt(0)
// Every read has at least eight MFMAs (128+ cycles) to return before its wait, so no counted lgkmcnt is needed.
// Bank conflicts: the weights' 16-row blocks are aligned (conflict-free); a pixel block is 16 consecutive tile pixels, which the
// packed patch image (H8Lines) maps to 16 consecutive rows mod 16 -- conflict-free for the taps whose row offset is even, one 2-way
// conflict per lane group for the odd ones (a ds_read_b128 group mixes the k-chunks c and c + 1; tools: /tmp-free derivation in
// DESIGN.md 3.11).
#pragma once

__device__ __forceinline__ void h16_wait8(ig8_u32x4_t (&p)[4], ig8_u32x4_t (&w)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]) :: "memory");
}

template <typename T, int MODE>
__global__ __launch_bounds__(512) void conv_halo16_kernel(const ConvArgs a, int PH, int PW, int tiles_x, int tiles_y, const H8Lines lines) {
    constexpr int WNW = 2, MI = 4, NJ = 4, BN = 128, BST = 3;
    constexpr int B_BYTES = BN * 128, B_IPW = BN / 64;
    constexpr int A_OFF = BST * B_BYTES, SCRATCH = A_OFF + 2 * H8_A_BYTES;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SCRATCH + 1024 + H8_MAXLINES * 4];
    unsigned* const ltab = (unsigned*)(smem + SCRATCH + 1024);
    for (int i = 0; i < lines.n; ++i) if (threadIdx.x == 0) ltab[i] = lines.base[i];
    __syncthreads();

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WNW, wn = wave % WNW;
    int g = blockIdx.z, mt, nt, part = -1, tail = 0;
    if (a.sk_parts > 1) {
        const int bx = blockIdx.x;
        int Lp;
        if (bx < a.sk_full) Lp = xcd_remap(bx, a.sk_full);
        else { const int bb = bx - a.sk_full; tail = bb / a.sk_parts; part = bb - tail * a.sk_parts; Lp = a.sk_full + tail; }
        g = Lp / a.sk_tpg;
        Lp -= g * a.sk_tpg;
        nt = Lp % a.sk_gy;
        mt = Lp / a.sk_gy;
    } else xcd_tile(gridDim.x, gridDim.y, mt, nt);
    const int tx = mt % tiles_x;
    const int ty = (mt / tiles_x) % tiles_y;
    const int b = mt / (tiles_x * tiles_y);
    const int y0 = ty * PH, x0 = tx * PW, n0 = nt * BN;
    const int PW2 = PW + 2, NI = 2 * ((lines.rows + 15) >> 4), NPIX = PH * PW, NLINES = lines.n;
    const int Cg = a.Cg;
    int s0 = 0, s1 = Cg >> 6;
    if (part >= 0) { const int nsl = s1; s0 = part * nsl / a.sk_parts; s1 = (part + 1) * nsl / a.sk_parts; }
    const unsigned Cg2 = (unsigned)Cg * 2u;
    const unsigned long zaddr = (unsigned long)(const void*)octa_zero_page;
    const unsigned long xbase = (unsigned long)((const T*)a.x + a.xoff + g * a.CgStride);
    const unsigned long wbase = (unsigned long)((const T*)a.w + (size_t)g * a.Ng * (size_t)(9 * Cg));
    const unsigned sbase = lds_addr(smem);

    // ---- DMA roles (as halo8): instruction I covers lines 4 I .. 4 I + 3; lane l writes slot l & 15 of line L = 4 I + (l >> 4)
    const int Lq = lane >> 4, hq = (lane >> 3) & 1;
    const int L7 = (4 * (wave & 1) + Lq) & 7;
    const int chunk = (lane & 7) ^ L7;
    unsigned aoff[H8_A_IPW];
    unsigned amask = 0;
#pragma unroll
    for (int j = 0; j < H8_A_IPW; ++j) {
        const int I = wave + 8 * j;
        const int L = 4 * I + Lq;
        const int R = (L & 7) + 8 * hq + 16 * (L >> 3);
        int ry = -1, rx = 0;
        for (int q = 0; q < NLINES; ++q) { const unsigned d = (unsigned)R - ltab[q]; if (d < (unsigned)PW2) { ry = q; rx = (int)d; } }
        const int iy = y0 - 1 + ry, ix = x0 - 1 + rx;
        const bool ok = (I < NI) && (ry >= 0) && ((unsigned)iy < (unsigned)a.H) && ((unsigned)ix < (unsigned)a.W);
        aoff[j] = ok ? (unsigned)(((b * a.H + iy) * a.W + ix) * a.ldx + chunk * 8) * 2u : 0u;
        amask |= (ok ? 1u : 0u) << j;
    }
    unsigned long wptr[B_IPW];
    unsigned wkm[B_IPW];
#pragma unroll
    for (int j = 0; j < B_IPW; ++j) {
        const int I = wave + 8 * j;
        const int L = 4 * I + Lq;
        const int nloc = (L & 7) + 8 * hq + 16 * (L >> 3);
        const bool ok = n0 + nloc < a.Ng;
        wptr[j] = ok ? wbase + (unsigned long)((unsigned)((n0 + nloc) * (9 * Cg) + chunk * 8) * 2u) : zaddr;
        wkm[j] = ok ? 0xffffffffu : 0u;
    }
    auto dmaA = [&](int j, int slice) {
        const int I = wave + 8 * j;
        const unsigned long src = ((amask >> j) & 1u) ? xbase + aoff[j] + (unsigned)(slice * 128) : zaddr;
        const unsigned dst = I < NI ? sbase + (unsigned)(A_OFF + (slice & 1) * H8_A_BYTES + I * 1024) : sbase + (unsigned)SCRATCH;
        glds16_fast((const void*)src, __builtin_amdgcn_readfirstlane(dst));
    };
    auto dmaB = [&](int j, unsigned koff, int slot) {
        const unsigned long src = wptr[j] + (unsigned long)(koff & wkm[j]);
        glds16_fast((const void*)src, __builtin_amdgcn_readfirstlane(sbase + (unsigned)(slot * B_BYTES + (wave + 8 * j) * 1024)));
    };

    // ---- fragment addresses.  16x16x32: lane (r = lane & 15, q = lane >> 4) supplies 8 consecutive k of row r: chunk 4 ks + q of
    // k32-step ks; slot = chunk ^ (R & 7), i.e. address(ks = 1) = address(0) ^ 64.  The four 16-row weight blocks of a wave are 8 lines
    // (2048 bytes) apart: one address register, immediate offsets (ring slot + block).
    const int r = lane & 15, q = lane >> 4;
    unsigned bfw;
    {
        const int nrow = wn * 64 + r;
        bfw = sbase + h8_rowbase(nrow) + (unsigned)((q ^ (nrow & 7)) << 4);
    }
    // pixels: [tap][block pair], k32-step 0, as 16-bit byte offsets into a patch buffer (< 45056), two blocks per register: 36 full
    // addresses beside 64 accumulator and 64 fragment registers spilled
    unsigned ta[9][MI / 2];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int i2 = 0; i2 < MI / 2; ++i2) ta[tp][i2] = 0u;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        int p = wm * 64 + 16 * i + r;
        p = p < NPIX ? p : NPIX - 1;
        const int py = p / PW, px = p - py * PW;
        unsigned lb[3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) lb[dy] = ltab[py + dy];
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
            const int kh = tp / 3, kw = tp - kh * 3;
            const int R = (int)lb[MODE == 0 ? kh : 2 - kh] + px + 1 + (MODE == 0 ? (kw - 1) : (1 - kw));
            ta[tp][i >> 1] |= (h8_rowbase(R) + (unsigned)((q ^ (R & 7)) << 4)) << (16 * (i & 1));
        }
    }
    const unsigned abase = sbase + (unsigned)A_OFF;
#define H16_TA(TP, I_) (((I_) & 1) ? (ta[TP][(I_) >> 1] >> 16) : (ta[TP][(I_) >> 1] & 0xffffu))

    f32x4_t acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jn = 0; jn < NJ; ++jn) acc[i][jn] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

#define H16_SB __builtin_amdgcn_sched_barrier(0)
    OCTA_STAMP_DECL;
    OCTA_STAMP_BEGIN;
#pragma unroll
    for (int j = 0; j < H8_A_IPW; ++j) dmaA(j, s0);
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int j = 0; j < B_IPW; ++j) dmaB(j, (unsigned)u * Cg2 + (unsigned)(s0 * 128), u);
    wait_vmcnt<2 * B_IPW>();
    __builtin_amdgcn_s_barrier();
    ig8_u32x4_t xp[4], xw[4], yp[4], yw[4];              // X = (stage, step 0), Y = (stage, step 1): pixels, weights
    unsigned cur[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) cur[i] = H16_TA(0, i) + abase + (unsigned)((s0 & 1) * H8_A_BYTES);
#pragma unroll
    for (int i = 0; i < MI; ++i) xp[i] = h8_rd(cur[i]);
    xw[0] = h8_rdo<0>(bfw); xw[1] = h8_rdo<2048>(bfw); xw[2] = h8_rdo<4096>(bfw); xw[3] = h8_rdo<6144>(bfw);
    h16_wait8(xp, xw);
#define H16_MM(PF, WF, I_, J_) Mma8<T>::run(WF[J_], PF[I_], acc[I_][J_]);
    // One stage.  MFMA order inside a step: (0,0) (0,1) (1,0) (1,1) (0,2) (0,3) (1,2) (1,3) (2,0) ... -- blocks of 2 x 2.
#define H16_STAGE(TAP)                                                                                         \
    {                                                                                                          \
        constexpr int SLOT = (TAP) % 3, NSLOT = ((TAP) + 1) % 3, NTAP = ((TAP) + 1) % 9, TAP3 = ((TAP) + 3) % 9; \
        constexpr unsigned BO = (unsigned)(SLOT * B_BYTES), NBO = (unsigned)(NSLOT * B_BYTES);                 \
        const unsigned bfw1 = bfw ^ 64u;                                                                       \
        H16_SB;                                                                                                \
        /* step 0: X; reads of (st, 1) -> Y */                                                                 \
        H16_MM(xp, xw, 0, 0) yp[0] = h8_rd(cur[0] ^ 64u); H16_SB;                                              \
        H16_MM(xp, xw, 0, 1) yp[1] = h8_rd(cur[1] ^ 64u); H16_SB;                                              \
        H16_MM(xp, xw, 1, 0) yw[0] = h8_rdo<BO>(bfw1); H16_SB;                                                 \
        H16_MM(xp, xw, 1, 1) yw[1] = h8_rdo<BO + 2048>(bfw1); H16_SB;                                          \
        H16_MM(xp, xw, 0, 2) yp[2] = h8_rd(cur[2] ^ 64u); H16_SB;                                              \
        H16_MM(xp, xw, 0, 3) yp[3] = h8_rd(cur[3] ^ 64u); H16_SB;                                              \
        H16_MM(xp, xw, 1, 2) yw[2] = h8_rdo<BO + 4096>(bfw1); H16_SB;                                          \
        H16_MM(xp, xw, 1, 3) yw[3] = h8_rdo<BO + 6144>(bfw1); H16_SB;                                          \
        H16_MM(xp, xw, 2, 0) H16_SB; H16_MM(xp, xw, 2, 1) H16_SB; H16_MM(xp, xw, 3, 0) H16_SB; H16_MM(xp, xw, 3, 1) H16_SB; \
        H16_MM(xp, xw, 2, 2) H16_SB; H16_MM(xp, xw, 2, 3) H16_SB; H16_MM(xp, xw, 3, 2) H16_SB; H16_MM(xp, xw, 3, 3) H16_SB; \
        h16_wait8(yp, yw);                                                                                     \
        if ((TAP) >= 7 && lastslice) wait_vmcnt<0>();                                                          \
        else if ((TAP) >= 3 && (TAP) <= 5 && !lastslice) wait_vmcnt<B_IPW + 2>();                              \
        else wait_vmcnt<B_IPW>();                                                                              \
        __builtin_amdgcn_s_barrier();                                                                          \
        H16_SB;                                                                                                \
        if ((TAP) == 8 && lastslice) {     /* the last stage: nothing left to fetch */                         \
            _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int jn = 0; jn < NJ; ++jn) { H16_MM(yp, yw, i, jn) } \
        } else {                                                                                               \
            /* step 1: Y; reads of (st + 1, 0) -> X (their addresses replace this stage's in cur[]), then the DMA instructions */ \
            _Pragma("unroll") for (int i = 0; i < MI; ++i) cur[i] = H16_TA(NTAP, i) + ((TAP) == 8 ? naoffs : aoffs); \
            H16_SB;                                                                                            \
            H16_MM(yp, yw, 0, 0) xp[0] = h8_rd(cur[0]); H16_SB;                                                \
            H16_MM(yp, yw, 0, 1) xp[1] = h8_rd(cur[1]); H16_SB;                                                 \
            H16_MM(yp, yw, 1, 0) xw[0] = h8_rdo<NBO>(bfw); H16_SB;                                             \
            H16_MM(yp, yw, 1, 1) xw[1] = h8_rdo<NBO + 2048>(bfw); H16_SB;                                      \
            H16_MM(yp, yw, 0, 2) xp[2] = h8_rd(cur[2]); H16_SB;                                                \
            H16_MM(yp, yw, 0, 3) xp[3] = h8_rd(cur[3]); H16_SB;                                                 \
            H16_MM(yp, yw, 1, 2) xw[2] = h8_rdo<NBO + 4096>(bfw); H16_SB;                                      \
            H16_MM(yp, yw, 1, 3) xw[3] = h8_rdo<NBO + 6144>(bfw); H16_SB;                                      \
            H16_MM(yp, yw, 2, 0)                                                                               \
            if (!((TAP) >= 6 && lastslice)) {                                                                  \
                const unsigned koff3 = (unsigned)TAP3 * Cg2 + kslice + ((TAP) >= 6 ? 128u : 0u);               \
                dmaB(0, koff3, SLOT);                                                                          \
            }                                                                                                  \
            H16_SB;                                                                                            \
            H16_MM(yp, yw, 2, 1)                                                                               \
            if (!((TAP) >= 6 && lastslice)) {                                                                  \
                const unsigned koff3 = (unsigned)TAP3 * Cg2 + kslice + ((TAP) >= 6 ? 128u : 0u);               \
                dmaB(1, koff3, SLOT);                                                                          \
            }                                                                                                  \
            H16_SB;                                                                                            \
            H16_MM(yp, yw, 3, 0)                                                                               \
            if ((TAP) >= 2 && (TAP) <= 4 && !lastslice) dmaA(2 * ((TAP) - 2), slice + 1);                      \
            H16_SB;                                                                                            \
            H16_MM(yp, yw, 3, 1)                                                                               \
            if ((TAP) >= 2 && (TAP) <= 4 && !lastslice) dmaA(2 * ((TAP) - 2) + 1, slice + 1);                  \
            H16_SB;                                                                                            \
            H16_MM(yp, yw, 2, 2) H16_SB; H16_MM(yp, yw, 2, 3) H16_SB; H16_MM(yp, yw, 3, 2) H16_SB; H16_MM(yp, yw, 3, 3) H16_SB; \
            h16_wait8(xp, xw);                                                                                 \
        }                                                                                                      \
    }
    for (int slice = s0; slice < s1; ++slice) {
        const bool lastslice = slice + 1 == s1;
        const unsigned aoffs = abase + (unsigned)((slice & 1) * H8_A_BYTES), naoffs = abase + (unsigned)(((slice + 1) & 1) * H8_A_BYTES);
        const unsigned kslice = (unsigned)(slice * 128);
        // (opaque re-definition of the loop-invariant addresses: otherwise their XOR-ed / offset variants are hoisted out of the loop
        // into dozens of registers and the accumulators spill, as in halo8)
        asm volatile("" : "+v"(bfw));
        asm volatile("" : "+v"(aoff[0]), "+v"(aoff[1]), "+v"(aoff[2]), "+v"(aoff[3]), "+v"(aoff[4]), "+v"(aoff[5]));      // (keeps the 64-bit patch source pointers from being hoisted: 12 registers)
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) asm volatile("" : "+v"(ta[tp][0]), "+v"(ta[tp][1]));
        H16_STAGE(0) H16_STAGE(1) H16_STAGE(2) H16_STAGE(3) H16_STAGE(4) H16_STAGE(5) H16_STAGE(6) H16_STAGE(7) H16_STAGE(8)
    }
#undef H16_STAGE
#undef H16_TA
#undef H16_MM
#undef H16_SB
    OCTA_STAMP_END(octa_diag_stamps_halo8)

    // ---- epilogue.  D[n][m] of block (i, jn): lane (r, q) holds pixel m = r of pixel block i and channels 4 q .. 4 q + 3 of weight block jn
    if (part >= 0) {
        float* __restrict__ wsp = a.sk_ws + ((size_t)tail * a.sk_parts + part) * (size_t)(256 * BN);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int p = wm * 64 + 16 * i + r;
#pragma unroll
            for (int jn = 0; jn < NJ; ++jn) *(f32x4_t*)(wsp + (size_t)p * BN + wn * 64 + 16 * jn + 4 * q) = acc[i][jn];
        }
        return;
    }
    T* __restrict__ yb = (T*)a.y + a.yoff;
    const int act = a.act;
#pragma unroll
    for (int jn = 0; jn < NJ; ++jn) {
        const int nb = n0 + wn * 64 + 16 * jn + 4 * q;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.bias)
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[e] = (nb + e < a.Ng) ? a.bias[g * a.Ng + nb + e] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][jn][e] += bv[e];
    }
#define H16_ACT(EXPR) _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int jn = 0; jn < NJ; ++jn) \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) { const float v = acc[i][jn][e]; acc[i][jn][e] = (EXPR); }
    if (act == OCTA_ACT_RELU) { H16_ACT(v > 0.f ? v : 0.f) }
    else if (act == OCTA_ACT_LEAKY02) { H16_ACT(v > 0.f ? v : 0.2f * v) }
    else if (act == OCTA_ACT_SIGMOID) { H16_ACT(1.f / (1.f + __expf(-v))) }
    else if (act == OCTA_ACT_TANH) { H16_ACT(tanhf(v)) }
#undef H16_ACT
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int p = wm * 64 + 16 * i + r;
        const int py = p / PW, px = p - py * PW;
        const int oy = y0 + py, ox = x0 + px;
        if (p >= NPIX || oy >= a.H || ox >= a.W) continue;
        const size_t pix = ((size_t)b * a.H + oy) * a.W + ox;
#pragma unroll
        for (int jn = 0; jn < NJ; ++jn) {
            const int nb = n0 + wn * 64 + 16 * jn + 4 * q;
            if (nb >= a.NgSt) continue;
            T* dst = yb + pix * a.ldy + g * a.Ng + nb;
            if (a.vec_store && nb + 3 < a.Ng) {
                *(uint2*)dst = make_uint2(pack2<T>(acc[i][jn][0], acc[i][jn][1]), pack2<T>(acc[i][jn][2], acc[i][jn][3]));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (nb + e < a.NgSt) DT<T>::st(dst + e, nb + e < a.Ng ? acc[i][jn][e] : 0.f);
            }
        }
    }
}
