// 8-wave implicit-GEMM convolution kernel (forward and stride-1 data gradient), bf16 / f16, included by conv.hip.
//
// C[m][n] = sum_k im2col(x)[m][k] * W[n][k]: one workgroup (512 threads, one per CU) owns a 256(M) x 128(N) or 128 x 256
// output slab, every wave a 64 x 64 sub-tile (16 MFMA 16x16x32 accumulators).  K advances 64 elements (8 chunks of 16 B =
// one 128-byte line per row) per stage through a 3-stage LDS ring (3 x 48 KB) filled by LDS-DMA, one raw s_barrier per
// stage, counted vmcnt.
// LDS image of a stage: rows of 128 bytes, rows R and R + 8 share one 256-byte line (line = (R & 7) + 8 (R >> 4), half =
// (R >> 3) & 1), and the 16-byte chunk c of row R sits at slot 8 half + (c ^ (R & 7)).  (a) One LDS-DMA wave-instruction
// (1 KiB = 4 lines) therefore fetches 8 FULL 128-byte rows from global memory -- a plane-per-chunk image made every
// instruction touch 64 different cache lines and ran 30 % slower than the 4-wave kernels; (b) the 16 lanes of a
// ds_read_b128 group (rows r = 0..15, one logical chunk) hit 16 distinct slots of the 256-byte bank row: conflict-free.
// Fragment reads are inline-asm ds_read_b128 with immediate offsets (4 address registers per wave).
// The loop is VALU-issue bound like wgrad8's: waves 0-3 issue the next stage before their MFMAs, waves 4-7 after.
#pragma once

typedef __attribute__((ext_vector_type(4))) unsigned ig8_u32x4_t;
#ifdef OCTA_DIAG_STAMPS
__device__ unsigned long long octa_diag_stamps_halo8[4096][4];
__device__ unsigned long long octa_diag_stamps_igemm8[4096][4];
#endif
template <int OFF> __device__ __forceinline__ ig8_u32x4_t ig8_rd(unsigned addr) {
    ig8_u32x4_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
__device__ __forceinline__ void ig8_wait8(ig8_u32x4_t (&a)[4], ig8_u32x4_t (&b)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) :: "memory");
}
// the four 16-row sub-tiles of a wave's 64 rows are 8 lines (2048 bytes) apart
__device__ __forceinline__ void ig8_load_sub(unsigned abase, unsigned bbase, ig8_u32x4_t (&xf)[4], ig8_u32x4_t (&wf)[4]) {
    xf[0] = ig8_rd<0>(abase); xf[1] = ig8_rd<2048>(abase); xf[2] = ig8_rd<4096>(abase); xf[3] = ig8_rd<6144>(abase);
    wf[0] = ig8_rd<0>(bbase); wf[1] = ig8_rd<2048>(bbase); wf[2] = ig8_rd<4096>(bbase); wf[3] = ig8_rd<6144>(bbase);
}

template <typename T> struct Mma8;
template <> struct Mma8<f16_t> {
    __device__ static __forceinline__ void run(const ig8_u32x4_t& a, const ig8_u32x4_t& b, f32x4_t& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct Mma8<bf16_t> {
    __device__ static __forceinline__ void run(const ig8_u32x4_t& a, const ig8_u32x4_t& b, f32x4_t& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};

// MODE 0: forward gather (any stride); 1: data-gradient gather, stride 1
// WM x WN waves of 64 x 64: 4 x 2 / 2 x 4 (8 waves, one workgroup per CU, 3-stage ring) or 2 x 2 (4 waves, 128 x 128 slab,
// 2-stage ring = 64 KB so that TWO workgroups share a CU: the mid-size layers - 25x25 / 50x50 1x1 convs, grouped 3x3 - have
// too few 256-wide slabs for 256 CUs and were left to the generic 64x64 kernel, whose loop is address arithmetic around 4 MFMAs)
template <typename T, int WM, int WN, int MODE>
__global__ __launch_bounds__(WM * WN * 64) void conv_igemm8_kernel(const ConvArgs a) {
    constexpr int EPC = 8;
    constexpr int NW = WM * WN;
    constexpr int BM = WM * 64, BN = WN * 64, KP = 8, STAGES = NW == 8 ? 3 : 2, PF = STAGES - 1;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, SBYTES = A_BYTES + B_BYTES;
    constexpr int RPR = NW * 8;                            // rows one round of DMA instructions (one per wave) covers
    constexpr int A_IPW = BM / RPR, B_IPW = BN / RPR;      // DMA instructions per wave and stage (8 rows each)
    constexpr int LPT = A_IPW + B_IPW;
    static_assert(NW == 8 || NW == 4, "8 or 4 waves of 64x64");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[STAGES * SBYTES];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int g = blockIdx.z;
    int mt, nt;
    // tail split-K (8 waves): a 1-D launch of sk_full whole tiles followed by sk_parts workgroups per remaining tile (tiles are
    // numbered group-major: sk_tpg tiles per group).
    // A launch of, say, 316 tiles on 256 CUs otherwise runs a second round that is 23 % full; here its 60 tiles become 240
    // workgroups of a quarter of the K range each.  part >= 0: this workgroup owns slices [s0, s1) and ends in a raw fp32 store.
    int part = -1, tail = 0;
    if (NW == 8 && a.sk_parts > 1) {
        const int b = blockIdx.x;
        int Lp;
        if (b < a.sk_full) Lp = xcd_remap(b, a.sk_full);
        else { const int bb = b - a.sk_full; tail = bb / a.sk_parts; part = bb - tail * a.sk_parts; Lp = a.sk_full + tail; }
        g = Lp / a.sk_tpg;
        Lp -= g * a.sk_tpg;
        nt = Lp % a.sk_gy;
        mt = Lp / a.sk_gy;
    } else xcd_tile(gridDim.x, gridDim.y, mt, nt);
    const int m0 = mt * BM, n0 = nt * BN;
    const int Kc = a.Kc, CgC = a.Cg / EPC, KW = a.KW, Wimg = a.W, ldx = a.ldx;
    const unsigned long zaddr = (unsigned long)(const void*)octa_zero_page;

    // ---- DMA roles.  Instruction j of this wave fills lines 4 (wave + NW j) .. + 3 of the stage image: lane l writes slot
    // l & 15 of line Lc + 4 NW j, i.e. row rowc + RPR j, chunk (l & 7) ^ (Lc & 7) -- row offset and chunk are lane constants.
    const int Lc = 4 * wave + (lane >> 4);
    const int rowc = (Lc & 7) + 8 * ((lane >> 3) & 1) + 16 * (Lc >> 3);
    const int chunk = (lane & 7) ^ (Lc & 7);
    // A: the A_IPW pixels this lane gathers (byte address of tap (0,0), channel 0; in-image tap mask)
    unsigned long abase[A_IPW];
    unsigned rmask[A_IPW];
#pragma unroll
    for (int j = 0; j < A_IPW; ++j) {
        const int m = m0 + rowc + RPR * j;
        const bool rvalid = m < a.M;
        const int mm = rvalid ? m : 0;
        const int ow = mm % a.OW;
        const int tq = mm / a.OW;
        const int oh = tq % a.OH;
        const int b = tq / a.OH;
        int rh, rw;
        if (MODE == 0) { rh = oh * a.stride - a.pad; rw = ow * a.stride - a.pad; }
        else { rh = oh + a.pad; rw = ow + a.pad; }
        unsigned mk = 0;
        if (rvalid) {
            const int ntaps = a.KH * KW;
            for (int tp = 0; tp < ntaps; ++tp) {
                const int kh_ = tp / KW, kw_ = tp - kh_ * KW;
                bool ok;
                if (MODE == 0) ok = ((unsigned)(rh + kh_) < (unsigned)a.H) && ((unsigned)(rw + kw_) < (unsigned)Wimg);
                else ok = ((unsigned)(rh - kh_) < (unsigned)a.H) && ((unsigned)(rw - kw_) < (unsigned)Wimg);
                mk |= (ok ? 1u : 0u) << tp;
            }
        }
        rmask[j] = mk;
        const long roff = ((long)(b * a.H + rh) * Wimg + rw) * ldx + a.xoff + g * a.CgStride;
        abase[j] = (unsigned long)((const T*)a.x + roff);
    }
    // B: the B_IPW weight rows this lane fetches (zero page when out of range); the K offset of a stage is added per stage
    const size_t Kelem = (size_t)Kc * EPC;
    unsigned long wptr[B_IPW];
    bool wvalid[B_IPW];
#pragma unroll
    for (int j = 0; j < B_IPW; ++j) {
        const int n = n0 + rowc + RPR * j;
        wvalid[j] = n < a.Ng;
        wptr[j] = wvalid[j] ? (unsigned long)((const T*)a.w + ((size_t)g * a.Ng + n) * Kelem + (size_t)chunk * EPC) : zaddr;
    }
    // K order: a stage is ONE tap x 64 channels (Cg % 64 == 0).  The stages walk the taps of one 64-channel slice before moving
    // to the next slice (slice-major), not the slices of one tap: the nine taps of a 3x3 layer re-read the same activation
    // lines (shifted by a pixel / an image row), and with the taps 8 .. 32 stages apart those lines had left the XCD's 4 MB
    // L2 in between -- rocprofv3 FETCH_SIZE showed 3-4x the algorithmic bytes for this kernel (profiles/r03_a_pmc_traffic.json).
    // Per-lane state of the stage to be issued next: tap (kh, kw), channel chunk cc = 8 * slice + chunk inside the tap.
    const int ntaps = a.KH * KW;
    int cc = chunk, tap = 0, kh = 0, kw = 0;
    int nk = (Kc + KP - 1) / KP;
    if (part >= 0) {
        const int nsl = CgC / KP;                           // 64-channel slices (Cg % 64 == 0)
        const int s0 = part * nsl / a.sk_parts, s1 = (part + 1) * nsl / a.sk_parts;
        cc += KP * s0;
        nk = (s1 - s0) * ntaps;
    }

    const unsigned sbase = lds_addr(smem);
    auto advanceK = [&]() {                               // next tap of the slice; after the last tap, the next slice
        tap += 1; kw += 1;
        const bool wrap2 = kw >= KW;
        kw = wrap2 ? 0 : kw;
        kh += wrap2 ? 1 : 0;
        const bool wrap = tap >= ntaps;
        tap = wrap ? 0 : tap; kh = wrap ? 0 : kh; kw = wrap ? 0 : kw;
        cc += wrap ? KP : 0;
    };
    auto issue = [&](int stage) {
        const unsigned ab = sbase + (unsigned)(stage * SBYTES), bb = ab + A_BYTES;
        const int tpix = kh * Wimg + kw;
        const long soff = (long)(((MODE == 0 ? tpix : -tpix) * ldx + cc * EPC) * 2);      // byte offset of (tap, channel chunk)
        const unsigned long woff = (unsigned long)((tap * CgC + (cc - chunk)) * (EPC * 2));    // (tap, slice) inside the packed weight row
        const bool kin = cc < CgC;
        const unsigned tbit = kin ? (1u << tap) : 0u;
#pragma unroll
        for (int j = 0; j < A_IPW; ++j) {
            const bool ok = (rmask[j] & tbit) != 0;
            const unsigned long src = ok ? (abase[j] + soff) : zaddr;
            glds16_fast((const void*)src, __builtin_amdgcn_readfirstlane(ab + (unsigned)((wave + NW * j) * 1024)));
        }
#pragma unroll
        for (int j = 0; j < B_IPW; ++j) {
            const unsigned long src = (kin && wvalid[j]) ? (wptr[j] + woff) : zaddr;
            glds16_fast((const void*)src, __builtin_amdgcn_readfirstlane(bb + (unsigned)((wave + NW * j) * 1024)));
        }
        advanceK();
    };

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // fragment addresses: lane (r, q), k-substep s reads logical chunk 4 s + q of row (wave tile row + 16 i + r):
    // line (r & 7) + 8 (4 w + i), slot 8 ((r >> 3) & 1) + ((4 s + q) ^ (r & 7)); i is an immediate offset (2048 i)
    const int r = lane & 15, q = lane >> 4;
    const int lrow = (r & 7) * 256 + ((r >> 3) & 1) * 128;
    const unsigned afrag0 = sbase + (unsigned)(wm * 4 * 2048 + lrow + (((0 + q) ^ (r & 7)) << 4));
    const unsigned afrag1 = sbase + (unsigned)(wm * 4 * 2048 + lrow + (((4 + q) ^ (r & 7)) << 4));
    const unsigned bfrag0 = sbase + (unsigned)(A_BYTES + wn * 4 * 2048 + lrow + (((0 + q) ^ (r & 7)) << 4));
    const unsigned bfrag1 = sbase + (unsigned)(A_BYTES + wn * 4 * 2048 + lrow + (((4 + q) ^ (r & 7)) << 4));
    if constexpr (NW == 4) {
    // 4-wave variant (two workgroups per CU cover each other's stalls): stage-at-a-time loop, waves 0-1 issue the next stage
    // before their MFMAs, waves 2-3 after
    const bool late = wave >= NW / 2;
    issue(0);
    if (PF > 1 && nk > 1) issue(1);
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt has landed once only the younger stages (PF - 1 of them, none in the 2-stage ring) are outstanding
        if (PF > 1 && kt + 1 < nk) wait_vmcnt<(PF - 1) * LPT>(); else wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (!late && kt + PF < nk) issue((kt + PF) % STAGES);
        const unsigned so = (unsigned)((kt % STAGES) * SBYTES);
        {
            ig8_u32x4_t xf0[4], wf0[4], xf1[4], wf1[4];
            ig8_load_sub(afrag0 + so, bfrag0 + so, xf0, wf0);
            ig8_wait8(xf0, wf0);
            ig8_load_sub(afrag1 + so, bfrag1 + so, xf1, wf1);       // in flight under the first 16 MFMAs
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) Mma8<T>::run(wf0[i], xf0[j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
            ig8_wait8(xf1, wf1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) Mma8<T>::run(wf1[i], xf1[j], acc[i][j]);
        }
        if (late && kt + PF < nk) issue((kt + PF) % STAGES);
    }
    } else {
    // The K loop (8 waves, one workgroup per CU).  A stage is two k32 sub-steps of 16 MFMAs; the 8 fragment reads of the NEXT sub-step and the LDS-DMA
    // instructions of a later stage are issued one at a time BETWEEN the MFMAs of the current one (an MFMA 16x16x32 keeps the
    // matrix pipe busy for 16 cycles and the vector issue port for 8: one ds_read_b128 or one DMA instruction rides in that
    // shadow), and the first sub-step of stage kt + 1 is fetched across the stage barrier under the last MFMAs of stage kt.
    // The first version read 8 fragments, waited for them, and only then started its MFMAs: ~150 idle cycles per sub-step
    // (measured on wgrad9: -21 % time from this reordering alone).
    // DMA placement: the slot of stage kt + 2 (= the slot of stage kt - 1) is free during the whole iteration kt; all LPT
    // instructions go into the first half, so that every one of them has at least a full iteration to land before the
    // barrier in the middle of iteration kt + 1 certifies the stage.
    auto issueA = [&](int j, int stage, long soff, unsigned tbit) {
        const bool ok = (rmask[j] & tbit) != 0;
        const unsigned long src = ok ? (abase[j] + soff) : zaddr;
        glds16_fast((const void*)src, __builtin_amdgcn_readfirstlane(sbase + (unsigned)(stage * SBYTES + (wave + NW * j) * 1024)));
    };
    auto issueB = [&](int j, int stage, bool kin, unsigned long woff) {
        const unsigned long src = (kin && wvalid[j]) ? (wptr[j] + woff) : zaddr;
        glds16_fast((const void*)src, __builtin_amdgcn_readfirstlane(sbase + (unsigned)(stage * SBYTES + A_BYTES + (wave + NW * j) * 1024)));
    };
    issue(0);
    if (nk > 1) issue(1);
    if (nk > 1) wait_vmcnt<LPT>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    OCTA_STAMP_DECL;
    OCTA_STAMP_BEGIN;
    ig8_u32x4_t xfX[4], wfX[4], xfY[4], wfY[4];
    ig8_load_sub(afrag0, bfrag0, xfX, wfX);
    ig8_wait8(xfX, wfX);
#define IG8_SB __builtin_amdgcn_sched_barrier(0)
#define IG8_MMA(WF, XF, i, j) Mma8<T>::run(WF[i], XF[j], acc[i][j])
    // DMA slots of one iteration: instruction index d in [0, LPT) (A rows first, then B rows); first half / second half split
    static_assert(LPT <= 8, "the DMA instructions of a stage ride behind MFMAs 8..15 of the first sub-step");
    for (int kt = 0; kt < nk; ++kt) {
        const int rem = nk - 1 - kt;
        const bool more = rem >= 2;                            // stage kt + 2 exists
        const int s2 = (kt + 2) % STAGES;
        const unsigned so = (unsigned)((kt % STAGES) * SBYTES), sn = (unsigned)(((kt + 1) % STAGES) * SBYTES);
        // (tap, channel chunk) state of stage kt + 2, frozen for this iteration's DMA instructions
        const int tpix = kh * Wimg + kw;
        const long soff = (long)(((MODE == 0 ? tpix : -tpix) * ldx + cc * EPC) * 2);
        const unsigned long woff = (unsigned long)((tap * CgC + (cc - chunk)) * (EPC * 2));
        const bool kin = cc < CgC;
        const unsigned tbit = kin ? (1u << tap) : 0u;
        auto dma = [&](int d) { if (d < A_IPW) issueA(d, s2, soff, tbit); else issueB(d - A_IPW, s2, kin, woff); };
        IG8_SB;
        // ---- sub-step 0 of stage kt (X); fetch sub-step 1 (Y)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                IG8_MMA(wfX, xfX, i, j);
                const int m = i * 4 + j;
                if (m < 4) xfY[m] = m == 0 ? ig8_rd<0>(afrag1 + so) : m == 1 ? ig8_rd<2048>(afrag1 + so) : m == 2 ? ig8_rd<4096>(afrag1 + so) : ig8_rd<6144>(afrag1 + so);
                else if (m < 8) wfY[m - 4] = m == 4 ? ig8_rd<0>(bfrag1 + so) : m == 5 ? ig8_rd<2048>(bfrag1 + so) : m == 6 ? ig8_rd<4096>(bfrag1 + so) : ig8_rd<6144>(bfrag1 + so);
                else if (more && (m - 8) < LPT) dma(m - 8);
                IG8_SB;
            }
        }
        if (more) wait_vmcnt<LPT>(); else wait_vmcnt<0>();       // stage kt + 1 has landed (this wave's part); only stage kt + 2 may be in flight
        ig8_wait8(xfY, wfY);
        __builtin_amdgcn_s_barrier();
        IG8_SB;
        // ---- sub-step 1 (Y); fetch sub-step 0 of stage kt + 1 (X)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                IG8_MMA(wfY, xfY, i, j);
                const int m = i * 4 + j;
                if (m < 4) xfX[m] = m == 0 ? ig8_rd<0>(afrag0 + sn) : m == 1 ? ig8_rd<2048>(afrag0 + sn) : m == 2 ? ig8_rd<4096>(afrag0 + sn) : ig8_rd<6144>(afrag0 + sn);
                else if (m < 8) wfX[m - 4] = m == 4 ? ig8_rd<0>(bfrag0 + sn) : m == 5 ? ig8_rd<2048>(bfrag0 + sn) : m == 6 ? ig8_rd<4096>(bfrag0 + sn) : ig8_rd<6144>(bfrag0 + sn);
                IG8_SB;
            }
        }
        if (more) advanceK();
        ig8_wait8(xfX, wfX);
        IG8_SB;
    }
#undef IG8_SB
#undef IG8_MMA
    OCTA_STAMP_END(octa_diag_stamps_igemm8)
    }

    if (NW == 8 && part >= 0) {
        // partial tile: raw fp32 accumulators, [m local][n local], to this (tile, part)'s slot of the workspace
        float* __restrict__ wsp = a.sk_ws + ((size_t)tail * a.sk_parts + part) * (size_t)(BM * BN);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *(f32x4_t*)(wsp + (size_t)((wm * 4 + j) * 16 + r) * BN + (wn * 4 + i) * 16 + q * 4) = acc[i][j];
        return;
    }
    // epilogue: lane holds, per (tn, tm), 4 consecutive output channels (rows of D) of pixel column r
    T* __restrict__ yb = (T*)a.y + a.yoff;
    bias_act_tile(acc, a, n0 + wn * 4 * 16 + q * 4, g);
    addend_tile<T>(acc, a, m0 + wm * 4 * 16 + r, n0 + wn * 4 * 16 + q * 4, g);
    if (a.stats) {
        __syncthreads();                                   // every wave has left the K loop: the ring may be overwritten
        stats_tile<BN>(acc, a, n0 + wn * 4 * 16 + q * 4, n0, m0 + wm * 4 * 16 + r, g, (float*)smem, blockIdx.x + blockIdx.y * gridDim.x);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + (wm * 4 + j) * 16 + r;
        if (m >= a.M) continue;
        size_t pix = (size_t)m;
        int ow = 0, oh = 0, bb = 0;
        if (a.upshuffle) { ow = m % a.OW; const int tq = m / a.OW; oh = tq % a.OH; bb = tq / a.OH; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int nb = n0 + (wn * 4 + i) * 16 + q * 4;
            if (nb >= a.NgSt) continue;
            int chan = g * a.Ng + nb;
            int bidx = chan;
            if (a.upshuffle) {
                const int dd = nb / a.CoutT;
                chan = nb - dd * a.CoutT;
                bidx = chan;
                pix = ((size_t)(bb * 2 * a.OH + 2 * oh + (dd >> 1)) * (2 * a.OW) + 2 * ow + (dd & 1));
            }
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e];
            T* dst = yb + pix * a.ldy + chan;
            if (a.vec_store && nb + 3 < a.Ng) {
                *(uint2*)dst = make_uint2(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (nb + e < a.NgSt) DT<T>::st(dst + e, nb + e < a.Ng ? v[e] : 0.f);
            }
        }
    }
}

// Finishes the split tiles: y = act(sum of the parts + bias) [+ addend], stored like the kernel's own epilogue does.
// One thread per (pixel row, 4 channels); blockIdx.x = tail tile, blockIdx.y = 16-row strip.
template <typename T, int BM, int BN>
__global__ __launch_bounds__(256) void igemm8_splitk_fix_kernel(const ConvArgs a) {
    int Lp = a.sk_full + blockIdx.x;
    const int g = Lp / a.sk_tpg;
    Lp -= g * a.sk_tpg;
    const int nt = Lp % a.sk_gy, mt = Lp / a.sk_gy;
    const int m0 = mt * BM, n0 = nt * BN;
    constexpr int TPR = BN / 4;                            // threads per row
    constexpr int RPB = 256 / TPR;                         // rows per pass
    const int tc = threadIdx.x % TPR, tr = threadIdx.x / TPR;
    const int nb = n0 + tc * 4;
    if (nb >= a.NgSt) return;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.bias)
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = (nb + e < a.Ng) ? a.bias[g * a.Ng + nb + e] : 0.f;
    const float* __restrict__ wsp = a.sk_ws + (size_t)blockIdx.x * a.sk_parts * (size_t)(BM * BN);
    T* __restrict__ yb = (T*)a.y + a.yoff;
    const T* __restrict__ ad = (const T*)a.addend;
    const int rows = BM / gridDim.y;
    for (int rr = blockIdx.y * rows + tr; rr < (int)(blockIdx.y + 1) * rows; rr += RPB) {
        const int m = m0 + rr;
        if (m >= a.M) break;
        f32x4_t v = *(const f32x4_t*)(wsp + (size_t)rr * BN + tc * 4);
        for (int p = 1; p < a.sk_parts; ++p) {
            const f32x4_t u = *(const f32x4_t*)(wsp + (size_t)p * (BM * BN) + (size_t)rr * BN + tc * 4);
            v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
        }
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = act_apply(v[e] + bv[e], a.act);
            if (ad && nb + e < a.Ng) o[e] += DT<T>::ld(ad + (size_t)m * a.ldadd + g * a.Ng + nb + e);
        }
        T* dst = yb + (size_t)m * a.ldy + g * a.Ng + nb;
        if (a.vec_store && nb + 3 < a.Ng) *(uint2*)dst = make_uint2(pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3]));
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (nb + e < a.NgSt) DT<T>::st(dst + e, nb + e < a.Ng ? o[e] : 0.f);
        }
    }
}

// How many parts the tiles behind the last full round of 256 are split into (1 = no split), for a grid of `tiles` tiles whose K
// range has `nsl` 64-channel slices of `ntaps` stages each.
static int splitk_parts(const ConvArgs& a, int tiles, int nsl, int ntaps, int64_t tile_floats) {
    static const bool off = getenv("OCTA_NO_SPLITK") != nullptr;
    if (off || !a.sk_ws) return 1;
    static const int maxtail = getenv("OCTA_SK_MAXTAIL") ? atoi(getenv("OCTA_SK_MAXTAIL")) : 128;
    static const int minst = getenv("OCTA_SK_MINSTAGES") ? atoi(getenv("OCTA_SK_MINSTAGES")) : 12;
    const int ncu = octa_num_cus();                        // one workgroup per CU: a round is ncu tiles (256 on MI355X)
    const int tail = tiles % ncu;
    if (tail == 0 || tail > maxtail * ncu / 256) return 1; // the last round is more than half full: nothing worth the extra pass
    int parts = ncu / tail;
    if (parts > 8) parts = 8;
    if (parts > nsl) parts = nsl;
    while (parts > 1 && (nsl / parts) * ntaps < minst) --parts;    // at least a dozen stages per part: prologue + ring fill are ~4
    if (parts < 2) return 1;
    if ((int64_t)tail * parts * tile_floats * 4 > a.sk_cap) return 1;
    return parts;
}

// eligibility + launch; variant 0: 256(M) x 128(N), 1: 128(M) x 256(N), 2: 128 x 128 with 4 waves.  Returns false when the
// legacy kernels must run.
template <typename T>
static bool launch_igemm8(const ConvArgs& a, int groups, int variant, hipStream_t st) {
    if (a.KH * a.KW > 32) return false;
    if (a.mode == 1 && a.stride != 1) return false;
    if ((int64_t)a.B * a.H * a.W * (int64_t)a.ldx >= (1ll << 30)) return false;
    if (a.Cg % 64 != 0) return false;                     // whole 128-byte lines per tap: at most one tap wrap per K step
    if (variant == 2) {
        dim3 grid(cdiv(a.M, 128), cdiv(a.Ng, 128), groups);
        if (a.mode == 0) conv_igemm8_kernel<T, 2, 2, 0><<<grid, 256, 0, st>>>(a);
        else conv_igemm8_kernel<T, 2, 2, 1><<<grid, 256, 0, st>>>(a);
        note_kernel<T>("conv_igemm8_kernel", 128, 128);
    } else {
        const int BM = variant == 0 ? 256 : 128, BN = variant == 0 ? 128 : 256;
        dim3 grid(cdiv(a.M, BM), cdiv(a.Ng, BN), groups);
        ConvArgs b = a;
        const int tiles = grid.x * grid.y * groups;
        // (grouped layers: NgSt == Ng, i.e. no pad channels to zero-fill between the groups)
        const int parts = (!a.upshuffle && !a.stats && (groups == 1 || a.NgSt == a.Ng)) ? splitk_parts(a, tiles, a.Cg / 64, a.KH * a.KW, (int64_t)BM * BN) : 1;
        const int ntail = tiles % octa_num_cus();             // tiles behind the last full round
        if (parts > 1) {
            b.sk_parts = parts; b.sk_full = tiles - ntail; b.sk_gy = grid.y; b.sk_tpg = grid.x * grid.y;
            grid = dim3(b.sk_full + ntail * parts, 1, 1);
        }
        if (variant == 0) {
            if (a.mode == 0) conv_igemm8_kernel<T, 4, 2, 0><<<grid, 512, 0, st>>>(b);
            else conv_igemm8_kernel<T, 4, 2, 1><<<grid, 512, 0, st>>>(b);
            if (parts > 1) igemm8_splitk_fix_kernel<T, 256, 128><<<dim3(ntail, 8), 256, 0, st>>>(b);
            note_kernel<T>("conv_igemm8_kernel", 256, 128);
            if (parts > 1) { const size_t l = strlen(g_last_kernel); snprintf(g_last_kernel + l, sizeof(g_last_kernel) - l, "+tail%dx%d", ntail, parts); }
        } else {
            if (a.mode == 0) conv_igemm8_kernel<T, 2, 4, 0><<<grid, 512, 0, st>>>(b);
            else conv_igemm8_kernel<T, 2, 4, 1><<<grid, 512, 0, st>>>(b);
            if (parts > 1) igemm8_splitk_fix_kernel<T, 128, 256><<<dim3(ntail, 8), 256, 0, st>>>(b);
            note_kernel<T>("conv_igemm8_kernel", 128, 256);
            if (parts > 1) { const size_t l = strlen(g_last_kernel); snprintf(g_last_kernel + l, sizeof(g_last_kernel) - l, "+tail%dx%d", ntail, parts); }
        }
    }
    return true;
}
