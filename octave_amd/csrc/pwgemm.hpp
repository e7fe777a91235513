// Persistent pointwise GEMM (round 4): 1x1 / stride 1 / no padding convolutions -- forward AND data gradient, three of the
// four convs of every bottleneck, the shortcut convs, the conv-transpose up-shuffles -- as  Y[m][n] = act(sum_k X[m][k] W[n][k] + b[n])
// (+ addend), bf16 / f16, included by conv.hip behind igemm8.hpp (whose LDS image, fragment reads and MFMA wrappers it reuses).
//
// Why a second kernel.  These layers have SHORT K loops (K = 128 .. 1024, i.e. 2 .. 16 stages of 64) and are bound by memory
// and latency, not by the matrix pipe.  On the tile-per-workgroup kernels a tile is: address set-up, two DMA round trips to
// fill the ring, a handful of stages, a burst of stores -- the phases of one tile never overlap, and the 64 x 64 tiles the
// heuristic picks for occupancy move M N K 2 (1/64 + 1/64) bytes through L2 (163 MB for the 25 x 25 256 -> 1024 layer, whose
// operands are 26 MB).  Here ONE workgroup per CU walks a contiguous range of 128 x 256 / 256 x 128 / 128 x 128 tiles and the
// stage stream never drains between them: while tile i is in its last stages the LDS-DMA instructions of tile i + 1 are already
// in flight, and its epilogue stores leave while the next tile's MFMAs run.  No taps, no bounds masks: a lane's DMA source is
// one running pointer per row (zero page with a zero step for rows beyond M / N).
//
// Tile order: n fastest inside an M strip, each XCD owns a contiguous range of tiles (the strip's activations are fetched into
// ONE L2), the workgroups of an XCD split its range contiguously.
#pragma once

// bias + activation + optional addend + store of one 64 x 64 wave tile, exactly like conv_igemm8_kernel's epilogue
template <typename T>
__device__ __forceinline__ void pw_store_tile(f32x4_t (&acc)[4][4], const ConvArgs& a, int m0, int n0, int wm, int wn, int r, int q) {
    T* __restrict__ yb = (T*)a.y + a.yoff;
    bias_act_tile(acc, a, n0 + wn * 64 + q * 4, 0);
    addend_tile<T>(acc, a, m0 + wm * 64 + r, n0 + wn * 64 + q * 4, 0);
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
            int chan = nb;
            if (a.upshuffle) {
                const int dd = nb / a.CoutT;
                chan = nb - dd * a.CoutT;
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

// WM x WN waves of 64 x 64 (8 waves: 128 x 256 / 256 x 128; 4 waves: 128 x 128); ntn = n-tiles per M strip, ntiles = all tiles
template <typename T, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64) void pwgemm_kernel(const ConvArgs a, int ntn, int ntiles) {
    constexpr int NW = WM * WN;
    constexpr int BM = WM * 64, BN = WN * 64, STAGES = 3;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, SBYTES = A_BYTES + B_BYTES;
    constexpr int RPR = NW * 8;                            // rows one round of DMA instructions (one per wave) covers
    constexpr int A_IPW = BM / RPR, B_IPW = BN / RPR;      // DMA instructions per wave and stage (8 rows each)
    constexpr int LPT = A_IPW + B_IPW;
    static_assert(NW == 8 || NW == 4, "8 or 4 waves of 64x64");
    static_assert(LPT <= 8, "the DMA instructions of a stage ride behind MFMAs 8..15 of the first sub-step");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[STAGES * SBYTES];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // ---- this workgroup's tile range: XCD x (= blockIdx.x & 7, workgroups are dealt round-robin over the XCDs) owns tiles
    // [T x / 8, T (x + 1) / 8), its workgroups (blockIdx.x >> 3 = 0 .. nwx - 1) split that range contiguously
    int t0, t1;
    {
        const int G = gridDim.x, NX = G < 8 ? G : 8;          // (fewer than 8 workgroups: one range each)
        const int xcd = blockIdx.x % NX, i = blockIdx.x / NX;
        const int nwx = (G - xcd + NX - 1) / NX;
        const long x0 = (long)ntiles * xcd / NX, x1 = (long)ntiles * (xcd + 1) / NX;
        t0 = (int)(x0 + (x1 - x0) * i / nwx);
        t1 = (int)(x0 + (x1 - x0) * (i + 1) / nwx);
    }
    if (t0 >= t1) return;                                  // (the whole workgroup: no barrier is left behind)
    const int nk = a.Cg >> 6;                              // stages of 64 channels per tile (Cg % 64 == 0)
    const int S = (t1 - t0) * nk;                          // this workgroup's stage stream
    const unsigned long zaddr = (unsigned long)(const void*)octa_zero_page;

    // ---- DMA roles (igemm8.hpp): instruction j of this wave fills lines 4 (wave + NW j) .. + 3 of the stage image; lane l
    // writes slot l & 15 of line Lc + 4 NW j, i.e. row rowc + RPR j, chunk (l & 7) ^ (Lc & 7)
    const int Lc = 4 * wave + (lane >> 4);
    const int rowc = (Lc & 7) + 8 * ((lane >> 3) & 1) + 16 * (Lc >> 3);
    const int chunk = (lane & 7) ^ (Lc & 7);
    const size_t Kelem = (size_t)a.Kc * 8;
    unsigned long ap[A_IPW], wp[B_IPW];
    unsigned astep[A_IPW], wstep[B_IPW];
    int td = t0, kd = 0;                                   // (tile, stage inside the tile) of the NEXT stage to be issued
    auto setup = [&](int tile) {
        const int mt = tile / ntn, nt = tile - mt * ntn;
#pragma unroll
        for (int j = 0; j < A_IPW; ++j) {
            const int m = mt * BM + rowc + RPR * j;
            const bool ok = m < a.M;
            ap[j] = ok ? (unsigned long)((const T*)a.x + ((size_t)m * a.ldx + a.xoff + chunk * 8)) : zaddr;
            astep[j] = ok ? 128u : 0u;
        }
#pragma unroll
        for (int j = 0; j < B_IPW; ++j) {
            const int n = nt * BN + rowc + RPR * j;
            const bool ok = n < a.Ng;
            wp[j] = ok ? (unsigned long)((const T*)a.w + ((size_t)n * Kelem + chunk * 8)) : zaddr;
            wstep[j] = ok ? 128u : 0u;
        }
    };
    auto advance = [&]() {                                 // the stage after the one just issued
        kd += 1;
        if (kd == nk) {
            kd = 0; td += 1;
            if (td < t1) setup(td);
        } else {
#pragma unroll
            for (int j = 0; j < A_IPW; ++j) ap[j] += astep[j];
#pragma unroll
            for (int j = 0; j < B_IPW; ++j) wp[j] += wstep[j];
        }
    };
    const unsigned sbase = lds_addr(smem);
    auto dma = [&](int d, int stage) {
        if (d < A_IPW) glds16_fast((const void*)ap[d], __builtin_amdgcn_readfirstlane(sbase + (unsigned)(stage * SBYTES + (wave + NW * d) * 1024)));
        else glds16_fast((const void*)wp[d - A_IPW], __builtin_amdgcn_readfirstlane(sbase + (unsigned)(stage * SBYTES + A_BYTES + (wave + NW * (d - A_IPW)) * 1024)));
    };
    auto issue = [&](int stage) {
#pragma unroll
        for (int d = 0; d < LPT; ++d) dma(d, stage);
        advance();
    };

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int r = lane & 15, q = lane >> 4;
    const int lrow = (r & 7) * 256 + ((r >> 3) & 1) * 128;
    const unsigned afrag0 = sbase + (unsigned)(wm * 4 * 2048 + lrow + (((0 + q) ^ (r & 7)) << 4));
    const unsigned afrag1 = sbase + (unsigned)(wm * 4 * 2048 + lrow + (((4 + q) ^ (r & 7)) << 4));
    const unsigned bfrag0 = sbase + (unsigned)(A_BYTES + wn * 4 * 2048 + lrow + (((0 + q) ^ (r & 7)) << 4));
    const unsigned bfrag1 = sbase + (unsigned)(A_BYTES + wn * 4 * 2048 + lrow + (((4 + q) ^ (r & 7)) << 4));

    setup(t0);
    issue(0);
    if (S > 1) issue(1);
    if (S > 1) wait_vmcnt<LPT>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    ig8_u32x4_t xfX[4], wfX[4], xfY[4], wfY[4];
    ig8_load_sub(afrag0, bfrag0, xfX, wfX);
    ig8_wait8(xfX, wfX);
#define PW_SB __builtin_amdgcn_sched_barrier(0)
#define PW_MMA(WF, XF, i, j) Mma8<T>::run(WF[i], XF[j], acc[i][j])
    int tc = t0, kc = 0;                                   // (tile, stage inside the tile) being multiplied
    for (int s = 0; s < S; ++s) {
        const bool more = (S - 1 - s) >= 2;                // stage s + 2 exists
        const int s2 = (s + 2) % STAGES;
        const unsigned so = (unsigned)((s % STAGES) * SBYTES), sn = (unsigned)(((s + 1) % STAGES) * SBYTES);
        PW_SB;
        // ---- sub-step 0 of stage s (X); fetch sub-step 1 (Y); the DMA instructions of stage s + 2 behind MFMAs 8 ..
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                PW_MMA(wfX, xfX, i, j);
                const int m = i * 4 + j;
                if (m < 4) xfY[m] = m == 0 ? ig8_rd<0>(afrag1 + so) : m == 1 ? ig8_rd<2048>(afrag1 + so) : m == 2 ? ig8_rd<4096>(afrag1 + so) : ig8_rd<6144>(afrag1 + so);
                else if (m < 8) wfY[m - 4] = m == 4 ? ig8_rd<0>(bfrag1 + so) : m == 5 ? ig8_rd<2048>(bfrag1 + so) : m == 6 ? ig8_rd<4096>(bfrag1 + so) : ig8_rd<6144>(bfrag1 + so);
                else if (more && (m - 8) < LPT) dma(m - 8, s2);
                PW_SB;
            }
        }
        if (more) wait_vmcnt<LPT>(); else wait_vmcnt<0>();   // stage s + 1 has landed (this wave's part); only stage s + 2 may be in flight
        ig8_wait8(xfY, wfY);
        __builtin_amdgcn_s_barrier();
        PW_SB;
        // ---- sub-step 1 (Y); fetch sub-step 0 of stage s + 1 (X)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                PW_MMA(wfY, xfY, i, j);
                const int m = i * 4 + j;
                if (m < 4) xfX[m] = m == 0 ? ig8_rd<0>(afrag0 + sn) : m == 1 ? ig8_rd<2048>(afrag0 + sn) : m == 2 ? ig8_rd<4096>(afrag0 + sn) : ig8_rd<6144>(afrag0 + sn);
                else if (m < 8) wfX[m - 4] = m == 4 ? ig8_rd<0>(bfrag0 + sn) : m == 5 ? ig8_rd<2048>(bfrag0 + sn) : m == 6 ? ig8_rd<4096>(bfrag0 + sn) : ig8_rd<6144>(bfrag0 + sn);
                PW_SB;
            }
        }
        if (more) advance();
        ig8_wait8(xfX, wfX);
        PW_SB;
        // ---- end of a tile: its stores leave while the next tile's stages (already in flight) land
        kc += 1;
        if (kc == nk) {
            const int mt = tc / ntn, nt = tc - mt * ntn;
            pw_store_tile<T>(acc, a, mt * BM, nt * BN, wm, wn, r, q);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            kc = 0; tc += 1;
        }
    }
#undef PW_SB
#undef PW_MMA
}

// eligibility + launch; variant 0: 256(M) x 128(N), 1: 128 x 256, 2: 128 x 128 with 4 waves.  Returns false when another kernel must run.
template <typename T>
static bool launch_pwgemm(const ConvArgs& a, int groups, int variant, hipStream_t st) {
    if (groups != 1 || a.KH != 1 || a.KW != 1 || a.stride != 1 || a.pad != 0 || a.H != a.OH || a.W != a.OW) return false;
    if (a.Cg % 64 != 0 || a.stats || a.sk_parts > 1) return false;
    const int BM = variant == 0 ? 256 : 128, BN = variant == 1 ? 256 : 128;
    const int ntm = cdiv(a.M, BM), ntn = cdiv(a.Ng, BN);
    const long tiles = (long)ntm * ntn;
    if (tiles >= (1l << 30)) return false;
    const int grid = (int)(tiles < octa_num_cus() ? tiles : octa_num_cus());
    if (variant == 0) { pwgemm_kernel<T, 4, 2><<<grid, 512, 0, st>>>(a, ntn, (int)tiles); note_kernel<T>("pwgemm_kernel", 256, 128); }
    else if (variant == 1) { pwgemm_kernel<T, 2, 4><<<grid, 512, 0, st>>>(a, ntn, (int)tiles); note_kernel<T>("pwgemm_kernel", 128, 256); }
    else { pwgemm_kernel<T, 2, 2><<<grid, 256, 0, st>>>(a, ntn, (int)tiles); note_kernel<T>("pwgemm_kernel", 128, 128); }
    return true;
}
