// halo16 with a PERSISTENT tile loop (round 5): one workgroup per CU walks work items (whole tiles, then the parts of the tail-split
// tiles) and the stage pipeline never drains between them; included by conv.hip behind halo16.hpp.
//
// Why.  In-kernel stamps (profiles/r05_clock_stamps.txt): a one-tile-per-workgroup launch of the 512 -> 256 layer at 100 x 100 gives
// each tile 67 us of its CU, of which the main loop is 52.6 us -- 14.6 us per tile (22 % of the forward, 39 % of the data gradient
// with its 36-stage tiles) are dispatch ramp, address set-up, the DMA prologue (44 KB patch + 48 KB weights before the first MFMA)
// and the store epilogue.  Here an item change is a slice change: during the last slice of an item the patch prefetch (taps 2..4) and
// the weight prefetch (taps 6..8) fetch the FIRST slice of the next item, the last stage reads the next item's first fragments, and
// only the register epilogue (bias, activation, 16 stores per lane) sits between two items -- with the next item's first weight and
// patch stages already in flight or landed.  Per item only the DMA source offsets change (the LDS image, the fragment addresses and the
// DMA roles are the same for every tile): ~60 VALU + some SALU, done at the top of the item's last slice.
//
// Items: i in [0, nfull) = whole tile i; i >= nfull: tile nfull + (i - nfull) / parts, part (i - nfull) % parts of its 64-channel
// slices (raw fp32 partial tile -> halo8_splitk_fix_kernel).  Workgroup w takes items w, w + G, w + 2 G, ...; inside a round of G whole
// tiles the tile of workgroup w is xcd_remap(w): the workgroups of one XCD work on neighbouring tiles of one weight slab.
#pragma once

template <typename T, int MODE>
__global__ __launch_bounds__(512) void conv_halo16p_kernel(const ConvArgs a, int PH, int PW, int tiles_x, int tiles_y, const H8Lines lines, int nitems,
                                                           int nfull, int parts, int gy, int tpg) {
    constexpr int WNW = 2, MI = 4, NJ = 4, BN = 128, BST = 3;
    constexpr int B_BYTES = BN * 128, B_IPW = BN / 64;
    constexpr int A_OFF = BST * B_BYTES, SCRATCH = A_OFF + 2 * H8_A_BYTES;
    constexpr int TATAB = SCRATCH + 1024 + 512;            // per (wm, lane): 9 taps x 2 packed pixel fragment addresses = 72 bytes (18 KB)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[TATAB + 4 * 64 * 72];
    unsigned* const ltab = (unsigned*)(smem + SCRATCH + 1024);
    for (int i = 0; i < lines.n; ++i) if (threadIdx.x == 0) ltab[i] = lines.base[i];
    __syncthreads();

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WNW, wn = wave % WNW;
    const int G = gridDim.x;
    const int PW2 = PW + 2, NI = 2 * ((lines.rows + 15) >> 4), NPIX = PH * PW, NLINES = lines.n;
    const int Cg = a.Cg, NSL = Cg >> 6;
    const unsigned Cg2 = (unsigned)Cg * 2u;
    const unsigned long zaddr = (unsigned long)(const void*)octa_zero_page;
    const unsigned sbase = lds_addr(smem);

    // ---- work item -> (tile, part) -> (group, n-tile, image, patch origin, slice range); all wave-uniform
    struct Item { int g, n0, b, y0, x0, s0, s1, part, tail; };
    auto decode = [&](int it) -> Item {
        Item q;
        int Lp;
        q.part = -1; q.tail = 0;
        if (it < nfull) {
            const int round = it / G, w = it - round * G;
            const int cnt = min(G, nfull - round * G);
            Lp = round * G + xcd_remap(w, cnt);
        } else {
            const int j = it - nfull;
            q.tail = j / parts; q.part = j - q.tail * parts;
            Lp = nfull + q.tail;
        }
        q.g = Lp / tpg;
        Lp -= q.g * tpg;
        const int nt = Lp % gy, mt = Lp / gy;
        const int tx = mt % tiles_x, ty = (mt / tiles_x) % tiles_y;
        q.b = mt / (tiles_x * tiles_y);
        q.y0 = ty * PH; q.x0 = tx * PW; q.n0 = nt * BN;
        q.s0 = 0; q.s1 = NSL;
        if (q.part >= 0) { q.s0 = q.part * NSL / parts; q.s1 = (q.part + 1) * NSL / parts; }
        return q;
    };

    // ---- DMA roles (tile-invariant part): instruction I covers lines 4 I .. 4 I + 3; lane l writes slot l & 15 of line L = 4 I + (l >> 4).
    // Per instruction j of this wave: the patch row (ry, rx) its lane fetches (packed, bit 15 = valid), resp. the weight row nloc.
    const int Lq = lane >> 4, hq = (lane >> 3) & 1;
    const int L7 = (4 * (wave & 1) + Lq) & 7;
    const int chunk = (lane & 7) ^ L7;
    unsigned rowyx[H8_A_IPW / 2];                          // two instructions per register: (valid << 15 | ry << 8 | rx) x 2
#pragma unroll
    for (int j2 = 0; j2 < H8_A_IPW / 2; ++j2) rowyx[j2] = 0u;
#pragma unroll
    for (int j = 0; j < H8_A_IPW; ++j) {
        const int I = wave + 8 * j;
        const int L = 4 * I + Lq;
        const int R = (L & 7) + 8 * hq + 16 * (L >> 3);
        int ry = -1, rx = 0;
        for (int q = 0; q < NLINES; ++q) { const unsigned d = (unsigned)R - ltab[q]; if (d < (unsigned)PW2) { ry = q; rx = (int)d; } }
        const unsigned v = (I < NI && ry >= 0) ? (0x8000u | ((unsigned)ry << 8) | (unsigned)rx) : 0u;
        rowyx[j >> 1] |= v << (16 * (j & 1));
    }
    unsigned long wptr[B_IPW];
    unsigned wkm[B_IPW];
    auto set_weights = [&](const Item& q) {
        const unsigned long wbase = (unsigned long)((const T*)a.w + (size_t)q.g * a.Ng * (size_t)(9 * Cg));
#pragma unroll
        for (int j = 0; j < B_IPW; ++j) {
            const int I = wave + 8 * j;
            const int L = 4 * I + Lq;
            const int nloc = (L & 7) + 8 * hq + 16 * (L >> 3);
            const bool ok = q.n0 + nloc < a.Ng;
            wptr[j] = ok ? wbase + (unsigned long)((unsigned)((q.n0 + nloc) * (9 * Cg) + chunk * 8) * 2u) : zaddr;
            wkm[j] = ok ? 0xffffffffu : 0u;
        }
    };
    // patch instruction j of slice `slice` of item q into patch buffer `pb`.  The source offset is rebuilt from the packed (ry, rx) of the
    // role (a dozen VALU instructions, six times per slice) instead of living in six registers across the stages
    auto dmaA = [&](int j, const Item& q, int slice, int pb) {
        const int I = wave + 8 * j;
        const unsigned v = (rowyx[j >> 1] >> (16 * (j & 1))) & 0xffffu;
        const int ry = (int)((v >> 8) & 0x7fu), rx = (int)(v & 0xffu);
        const int iy = q.y0 - 1 + ry, ix = q.x0 - 1 + rx;
        const bool ok = (v & 0x8000u) && ((unsigned)iy < (unsigned)a.H) && ((unsigned)ix < (unsigned)a.W);
        const unsigned long xb = (unsigned long)((const T*)a.x + a.xoff + q.g * a.CgStride);
        const unsigned off = (unsigned)(((q.b * a.H + iy) * a.W + ix) * a.ldx + chunk * 8) * 2u + (unsigned)(slice * 128);
        const unsigned long src = ok ? xb + off : zaddr;
        const unsigned dst = I < NI ? sbase + (unsigned)(A_OFF + pb * H8_A_BYTES + I * 1024) : sbase + (unsigned)SCRATCH;
        glds16_fast((const void*)src, __builtin_amdgcn_readfirstlane(dst));
    };
    auto dmaB = [&](int j, unsigned koff, int slot) {
        const unsigned long src = wptr[j] + (unsigned long)(koff & wkm[j]);
        glds16_fast((const void*)src, __builtin_amdgcn_readfirstlane(sbase + (unsigned)(slot * B_BYTES + (wave + 8 * j) * 1024)));
    };

    // ---- fragment addresses (tile-invariant), as halo16
    const int r = lane & 15, q4 = lane >> 4;
    unsigned bfw;
    {
        const int nrow = wn * 64 + r;
        bfw = sbase + h8_rowbase(nrow) + (unsigned)((q4 ^ (nrow & 7)) << 4);
    }
    // pixel fragment addresses of k32-step 0, [tap][block pair] as 16-bit byte offsets into a patch buffer, two blocks per dword: the
    // eighteen dwords of a lane live in LDS (TATAB; the two waves of a wave row share them) and a stage fetches its tap's pair with one
    // ds_read_b64 -- in registers they were the difference between 256 VGPRs with spills inside the stage loop and none
    const unsigned tabase = sbase + (unsigned)(TATAB + (wm * 64 + lane) * 72);
    {
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
                ta[tp][i >> 1] |= (h8_rowbase(R) + (unsigned)((q4 ^ (R & 7)) << 4)) << (16 * (i & 1));
            }
        }
        if (wn == 0) {
            unsigned* dst = (unsigned*)(smem + TATAB + (wm * 64 + lane) * 72);
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) { dst[2 * tp] = ta[tp][0]; dst[2 * tp + 1] = ta[tp][1]; }
        }
    }
    __syncthreads();
    const unsigned abase = sbase + (unsigned)A_OFF;
    unsigned long long tpk;                                 // the tap pair being unpacked: (blocks 0, 1 | blocks 2, 3)
#define H16_RDTA(TP) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(tpk) : "v"(tabase), "n"((TP) * 8))
#define H16_TA(I_) ((unsigned)(((I_) & 2 ? (unsigned)(tpk >> 32) : (unsigned)tpk) >> (16 * ((I_) & 1))) & 0xffffu)

    f32x4_t acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jn = 0; jn < NJ; ++jn) acc[i][jn] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    int it = blockIdx.x;
    if (it >= nitems) return;
    Item cur = decode(it), nxt = cur;
    bool has_next = it + G < nitems;
    if (has_next) nxt = decode(it + G);

#define H16_SB __builtin_amdgcn_sched_barrier(0)
    OCTA_STAMP_DECL;
    OCTA_STAMP_BEGIN;
    // prologue of the FIRST item only: patch (s0) -> buffer 0, B(0..2); X = (stage 0, step 0)
    set_weights(cur);
    int pb = 0;                                            // patch buffer of the slice being computed
#pragma unroll
    for (int j = 0; j < H8_A_IPW; ++j) dmaA(j, cur, cur.s0, 0);
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int j = 0; j < B_IPW; ++j) dmaB(j, (unsigned)u * Cg2 + (unsigned)(cur.s0 * 128), u);
    H16_RDTA(0);
    wait_vmcnt<2 * B_IPW>();
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tpk) :: "memory");
    __builtin_amdgcn_s_barrier();
    ig8_u32x4_t xp[4], xw[4], yp[4], yw[4];
    unsigned curad[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) curad[i] = H16_TA(i) + abase;
#pragma unroll
    for (int i = 0; i < MI; ++i) xp[i] = h8_rd(curad[i]);
    xw[0] = h8_rdo<0>(bfw); xw[1] = h8_rdo<2048>(bfw); xw[2] = h8_rdo<4096>(bfw); xw[3] = h8_rdo<6144>(bfw);
    h16_wait8(xp, xw);
#define H16_MM(PF, WF, I_, J_) Mma8<T>::run(WF[J_], PF[I_], acc[I_][J_]);
    // One stage.  `more`: a slice follows this one in the workgroup's flattened (item, slice) sequence; nslice / npb: that slice's
    // index within ITS item and its patch buffer; at tap 6 of an item's last slice the weight pointers move on to the next item.
#define H16P_STAGE(TAP)                                                                                        \
    {                                                                                                          \
        constexpr int SLOT = (TAP) % 3, NSLOT = ((TAP) + 1) % 3, NTAP = ((TAP) + 1) % 9, TAP3 = ((TAP) + 3) % 9; \
        constexpr unsigned BO = (unsigned)(SLOT * B_BYTES), NBO = (unsigned)(NSLOT * B_BYTES);                 \
        const unsigned bfw1 = bfw ^ 64u;                                                                       \
        if ((TAP) == 6 && lastslice && more) set_weights(nxt);                                                 \
        H16_SB;                                                                                                \
        H16_MM(xp, xw, 0, 0) yp[0] = h8_rd(curad[0] ^ 64u); H16_SB;                                            \
        H16_MM(xp, xw, 0, 1) yp[1] = h8_rd(curad[1] ^ 64u); H16_SB;                                            \
        H16_MM(xp, xw, 1, 0) yw[0] = h8_rdo<BO>(bfw1); H16_SB;                                                 \
        H16_MM(xp, xw, 1, 1) yw[1] = h8_rdo<BO + 2048>(bfw1); H16_SB;                                          \
        H16_MM(xp, xw, 0, 2) yp[2] = h8_rd(curad[2] ^ 64u); H16_SB;                                            \
        H16_MM(xp, xw, 0, 3) yp[3] = h8_rd(curad[3] ^ 64u); H16_SB;                                            \
        H16_MM(xp, xw, 1, 2) yw[2] = h8_rdo<BO + 4096>(bfw1); H16_SB;                                          \
        H16_MM(xp, xw, 1, 3) yw[3] = h8_rdo<BO + 6144>(bfw1); H16_SB;                                          \
        H16_MM(xp, xw, 2, 0) H16_RDTA(NTAP); H16_SB;                                                           \
        H16_MM(xp, xw, 2, 1) H16_SB; H16_MM(xp, xw, 3, 0) H16_SB; H16_MM(xp, xw, 3, 1) H16_SB; \
        H16_MM(xp, xw, 2, 2) H16_SB; H16_MM(xp, xw, 2, 3) H16_SB; H16_MM(xp, xw, 3, 2) H16_SB; H16_MM(xp, xw, 3, 3) H16_SB; \
        h16_wait8(yp, yw);                                                                                     \
        asm volatile("" : "+v"(tpk));                                                                          \
        if ((TAP) >= 7 && !more) wait_vmcnt<0>();                                                              \
        else if ((TAP) >= 3 && (TAP) <= 5 && more) wait_vmcnt<B_IPW + 2>();                                    \
        else wait_vmcnt<B_IPW>();                                                                              \
        __builtin_amdgcn_s_barrier();                                                                          \
        H16_SB;                                                                                                \
        if ((TAP) == 8 && !more) {         /* the very last stage of this workgroup */                         \
            _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int jn = 0; jn < NJ; ++jn) { H16_MM(yp, yw, i, jn) } \
        } else {                                                                                               \
            _Pragma("unroll") for (int i = 0; i < MI; ++i) curad[i] = H16_TA(i) + ((TAP) == 8 ? naoffs : aoffs); \
            H16_SB;                                                                                            \
            H16_MM(yp, yw, 0, 0) xp[0] = h8_rd(curad[0]); H16_SB;                                              \
            H16_MM(yp, yw, 0, 1) xp[1] = h8_rd(curad[1]); H16_SB;                                              \
            H16_MM(yp, yw, 1, 0) xw[0] = h8_rdo<NBO>(bfw); H16_SB;                                             \
            H16_MM(yp, yw, 1, 1) xw[1] = h8_rdo<NBO + 2048>(bfw); H16_SB;                                      \
            H16_MM(yp, yw, 0, 2) xp[2] = h8_rd(curad[2]); H16_SB;                                              \
            H16_MM(yp, yw, 0, 3) xp[3] = h8_rd(curad[3]); H16_SB;                                              \
            H16_MM(yp, yw, 1, 2) xw[2] = h8_rdo<NBO + 4096>(bfw); H16_SB;                                      \
            H16_MM(yp, yw, 1, 3) xw[3] = h8_rdo<NBO + 6144>(bfw); H16_SB;                                      \
            H16_MM(yp, yw, 2, 0)                                                                               \
            if (!((TAP) >= 6 && !more)) {                                                                      \
                const unsigned koff3 = (unsigned)TAP3 * Cg2 + ((TAP) >= 6 ? nkslice : kslice);                 \
                dmaB(0, koff3, SLOT);                                                                          \
            }                                                                                                  \
            H16_SB;                                                                                            \
            H16_MM(yp, yw, 2, 1)                                                                               \
            if (!((TAP) >= 6 && !more)) {                                                                      \
                const unsigned koff3 = (unsigned)TAP3 * Cg2 + ((TAP) >= 6 ? nkslice : kslice);                 \
                dmaB(1, koff3, SLOT);                                                                          \
            }                                                                                                  \
            H16_SB;                                                                                            \
            H16_MM(yp, yw, 3, 0)                                                                               \
            if ((TAP) >= 2 && (TAP) <= 4 && more) dmaA(2 * ((TAP) - 2), lastslice ? nxt : cur, nslice, pb ^ 1); \
            H16_SB;                                                                                            \
            H16_MM(yp, yw, 3, 1)                                                                               \
            if ((TAP) >= 2 && (TAP) <= 4 && more) dmaA(2 * ((TAP) - 2) + 1, lastslice ? nxt : cur, nslice, pb ^ 1); \
            H16_SB;                                                                                            \
            H16_MM(yp, yw, 2, 2) H16_SB; H16_MM(yp, yw, 2, 3) H16_SB; H16_MM(yp, yw, 3, 2) H16_SB; H16_MM(yp, yw, 3, 3) H16_SB; \
            h16_wait8(xp, xw);                                                                                 \
        }                                                                                                      \
    }

    for (;;) {
        for (int slice = cur.s0; slice < cur.s1; ++slice) {
            const bool lastslice = slice + 1 == cur.s1;
            const bool more = !lastslice || has_next;
            const int nslice = lastslice ? nxt.s0 : slice + 1;
            const unsigned aoffs = abase + (unsigned)(pb * H8_A_BYTES), naoffs = abase + (unsigned)((pb ^ 1) * H8_A_BYTES);
            const unsigned kslice = (unsigned)(slice * 128), nkslice = (unsigned)(nslice * 128);
            asm volatile("" : "+v"(bfw), "+v"(rowyx[0]), "+v"(rowyx[1]), "+v"(rowyx[2]));      // (opaque: keeps address variants from being hoisted into registers)
            H16P_STAGE(0) H16P_STAGE(1) H16P_STAGE(2) H16P_STAGE(3) H16P_STAGE(4) H16P_STAGE(5) H16P_STAGE(6) H16P_STAGE(7) H16P_STAGE(8)
            pb ^= 1;
        }
        // ---- epilogue of item `cur` (the next item's first stages are in flight / landed meanwhile).
        // D[n][m] of block (i, jn): lane (r, q4) holds pixel m = r of pixel block i and channels 4 q4 .. 4 q4 + 3 of weight block jn
        if (cur.part >= 0) {
            float* __restrict__ wsp = a.sk_ws + ((size_t)cur.tail * parts + cur.part) * (size_t)(256 * BN);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int p = wm * 64 + 16 * i + r;
#pragma unroll
                for (int jn = 0; jn < NJ; ++jn) *(f32x4_t*)(wsp + (size_t)p * BN + wn * 64 + 16 * jn + 4 * q4) = acc[i][jn];
            }
        } else {
            T* __restrict__ yb = (T*)a.y + a.yoff;
            const int act = a.act;
#pragma unroll
            for (int jn = 0; jn < NJ; ++jn) {
                const int nb = cur.n0 + wn * 64 + 16 * jn + 4 * q4;
                float bv[4] = {0.f, 0.f, 0.f, 0.f};
                if (a.bias)
#pragma unroll
                    for (int e = 0; e < 4; ++e) bv[e] = (nb + e < a.Ng) ? a.bias[cur.g * a.Ng + nb + e] : 0.f;
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
                const int oy = cur.y0 + py, ox = cur.x0 + px;
                if (p >= NPIX || oy >= a.H || ox >= a.W) continue;
                const size_t pix = ((size_t)cur.b * a.H + oy) * a.W + ox;
#pragma unroll
                for (int jn = 0; jn < NJ; ++jn) {
                    const int nb = cur.n0 + wn * 64 + 16 * jn + 4 * q4;
                    if (nb >= a.NgSt) continue;
                    T* dst = yb + pix * a.ldy + cur.g * a.Ng + nb;
                    if (a.vec_store && nb + 3 < a.Ng) {
                        *(uint2*)dst = make_uint2(pack2<T>(acc[i][jn][0], acc[i][jn][1]), pack2<T>(acc[i][jn][2], acc[i][jn][3]));
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (nb + e < a.NgSt) DT<T>::st(dst + e, nb + e < a.Ng ? acc[i][jn][e] : 0.f);
                    }
                }
            }
        }
        if (!has_next) break;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int jn = 0; jn < NJ; ++jn) acc[i][jn] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        it += G;
        cur = nxt;
        has_next = it + G < nitems;
        if (has_next) nxt = decode(it + G);
    }
#undef H16P_STAGE
#undef H16_TA
#undef H16_RDTA
#undef H16_MM
#undef H16_SB
    OCTA_STAMP_END(octa_diag_stamps_halo8)
}

// launch: whole tiles + the parts of the tail tiles as work items over one workgroup per CU
template <typename T>
static bool launch_halo16p(const ConvArgs& a, int groups, hipStream_t st) {
    if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.upshuffle || a.addend || a.stats) return false;
    if (a.H != a.OH || a.W != a.OW || a.Cg % 64 != 0) return false;
    if ((int64_t)a.B * a.H * a.W * (int64_t)a.ldx >= (1ll << 30)) return false;
    if ((int64_t)groups * a.Ng * 9 * a.Cg >= (1ll << 30)) return false;
    int PH, PW;
    halo8_patch(a.H, a.W, PH, PW);
    if (PH + 2 > H8_MAXLINES || PW + 2 > 255) return false;
    H8Lines lines;
    halo8_lines(PH, PW, lines);
    const int tiles_y = cdiv(a.H, PH), tiles_x = cdiv(a.W, PW);
    const int gx = a.B * tiles_y * tiles_x, gy = cdiv(a.Ng, 128);
    const int tiles = gx * gy * groups;
    const int ncu = octa_num_cus();
    const int parts = (groups == 1 || a.NgSt == a.Ng) ? halo8_parts(a, tiles, a.Cg / 64) : 1;
    const int ntail = parts > 1 ? tiles % ncu : 0;
    const int nfull = tiles - ntail;
    const int nitems = nfull + ntail * parts;
    const int G = nitems < ncu ? nitems : ncu;
    ConvArgs b = a;
    if (parts > 1) { b.sk_parts = parts; b.sk_full = nfull; b.sk_gy = gy; b.sk_tpg = gx * gy; }
    if (a.mode == 0) conv_halo16p_kernel<T, 0><<<G, 512, 0, st>>>(b, PH, PW, tiles_x, tiles_y, lines, nitems, nfull, parts > 1 ? parts : 1, gy, gx * gy);
    else conv_halo16p_kernel<T, 1><<<G, 512, 0, st>>>(b, PH, PW, tiles_x, tiles_y, lines, nitems, nfull, parts > 1 ? parts : 1, gy, gx * gy);
    if (parts > 1) halo8_splitk_fix_kernel<T><<<dim3(ntail, 8), 256, 0, st>>>(b, PH, PW, tiles_x, tiles_y);
    note_kernel<T>("conv_halo16p_kernel", 256, 128);
    if (parts > 1) { const size_t l = strlen(g_last_kernel); snprintf(g_last_kernel + l, sizeof(g_last_kernel) - l, "+tail%dx%d", ntail, parts); }
    return true;
}
