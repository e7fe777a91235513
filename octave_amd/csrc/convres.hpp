// Resident-weight, persistent convolution for the wide, shallow layers (Cin/group 32 or 64, <= 256 output channels) at
// the full-resolution end of the U-Net.  Those launches are HBM-bound on paper (0.2-0.5 GB of activations, < 64 KB of
// weights) but ran at 20-25 % of HBM speed in the tile-per-workgroup kernels: a tile there is 9..18 MFMA steps, far too
// short to hide the latency of the weight stream it re-fetches per tile (profiles/r02_layer_times.txt).
//
// Here ONE 8-wave workgroup per CU keeps the whole packed weight operand in LDS for its lifetime and walks a contiguous
// range of 256-pixel tiles.  Per tile only the activation patch moves: LDS-DMA (global_load_lds_dwordx4) into an
// NST-deep ring, issued NST-1 tiles ahead, one counted s_waitcnt + one s_barrier per tile, no barrier inside a tile.
//   TAPS = 9: 3x3 / stride 1 / pad 1, tile = 16x16 pixels, patch = 18x18 pixels fetched once (zero outside the image);
//             MODE 0 forward (tap reads (y-1+kh, x-1+kw)), MODE 1 data gradient ([ci][kh][kw][co] operand, mirrored taps)
//   TAPS = 1: 1x1 / stride 1 (also the conv-transpose "upshuffle" GEMM), tile = 256 consecutive pixels.
// Patch rows are pixel-major (RB = 64*NCH bytes per pixel, whole 64/128-byte runs per DMA lane group - the coalesced
// shape, cf. igemm8.hpp) with the 16-byte slot XOR-swizzled by the row so that the MFMA fragment reads (16 consecutive
// rows, one slot) are bank-conflict-free: slot' = slot ^ ((row >> 2) & 3) for 64-byte rows, slot ^ (row & 7) for 128.
// vmcnt counts loads and stores in issue order on gfx9-family parts, so the counted wait at the top of a tile may leave
// the previous tiles' output stores in flight - but only when their number is known: `exact` (every lane stores every
// fragment with one vector store) and the tiles were full; otherwise the wait is stricter (never looser).

template <int NCH> __device__ __forceinline__ int res_swz(int p) { return NCH == 1 ? ((p >> 2) & 3) : (p & 7); }

// NH: output-channel passes per tile.  TN = 16 n-tiles at once need 128 accumulator + 128 weight-fragment registers and spill; with
// NH = 2 a tile is two passes of TN = 8 over the SAME resident patch (the 64 -> 256 pointwise / up-shuffle layers).
template <typename T, int TAPS, int NCH, int TN, int MODE, int NH = 1>
__global__ __launch_bounds__(512) void conv_res_kernel(const ConvArgs a, int ntiles, int tiles_x, int tiles_y, int exact) {
    constexpr int EPC = DT<T>::EPC;
    static_assert(EPC == 8, "16-bit element types only");
    constexpr int RB = 64 * NCH;                  // bytes per patch row (= Cg * 2)
    constexpr int SPR = 4 * NCH;                  // 16-byte slots per row
    constexpr int RPI = 64 / SPR;                 // patch rows per DMA instruction
    constexpr int PW = TAPS == 9 ? 18 : 16;
    constexpr int PROWS = TAPS == 9 ? 18 * 18 : 256;
    constexpr int LP = (PROWS + RPI * 8 - 1) / (RPI * 8);   // DMA instructions per wave per patch
    // COMPACT (3x3, 64 input channels, 64 output channels: 72 KB of weights): a ring stage holds only the NINSTR instructions that carry
    // patch rows (41 KB instead of 48 KB, so that two stages + the weights fit the 160 KB); the other instructions of the fixed
    // per-wave count land in a scratch KiB (the vmcnt waits count instructions), and the outputs leave by direct stores (no strips)
    constexpr bool COMPACT = TAPS == 9 && NCH == 2 && TN == 4;
    constexpr int NINSTR = (PROWS + RPI - 1) / RPI;
    constexpr int SROWS = COMPACT ? NINSTR * RPI : LP * 8 * RPI;           // rows of one ring stage
    constexpr int NST = NCH == 1 ? (TAPS == 9 ? 3 : 4) : 2;
    constexpr int BNR = TN * NH * 16;             // all output channels of the tile (every pass)
    constexpr int NSTEP = TAPS * NCH;
    constexpr int TM = 2;
    constexpr int NP = TN >= 2 ? TN / 2 : 1;      // 16-byte output stores per pixel row of a lane (fragment pairs)
    constexpr int NS = TM * NP * NH;              // output store instructions per wave per full tile (exact mode)
    constexpr int WAIT_BASE = (NST - 2) * LP;     // the younger patches
    constexpr int WAIT_ST = WAIT_BASE + (NST - 1) * NS > 60 ? 60 : WAIT_BASE + (NST - 1) * NS;
    constexpr int W_INSTR = NSTEP * 4 * BNR / 64; // weight DMA instructions per workgroup
    __shared__ uint4 sW[NSTEP * 4 * BNR];         // [step][plane q][n]
    __shared__ uint4 sP[NST * SROWS * SPR];       // [stage][row][slot]
    __shared__ float sBias[BNR];
    __shared__ uint4 sT[(TN >= 2 && TN <= 4 && !COMPACT) ? 8 * TM * 16 * (TN / 2) * 4 : 1];   // output transpose strips, one per wave
    __shared__ uint4 sScr[COMPACT ? 64 : 1];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 15, q = lane >> 4;
    // grouped layers (round 4): blockIdx.y = group, every group its own set of gridDim.x persistent workgroups with its own resident
    // operand; workgroup x of every group walks the same tile range and sits on the same XCD (gridDim.x % 8 == 0), so the
    // groups' slices of one activation line are fetched into that L2 once
    // (an output-channel SPLIT of a group -- a.Ng = 64 of its 128 / 256 channels per pseudo-group, launch_res -- shares the group's input
    // slice: bits 16.. of `exact` = log2 of the pseudo-groups per group)
    const int grp = blockIdx.y;
    const int Kelem = TAPS * a.Cg;
    const T* __restrict__ xg = (const T*)a.x + a.xoff + (grp >> ((exact >> 16) & 7)) * a.CgStride;
    const T* __restrict__ wg = (const T*)a.w + (size_t)grp * a.Ng * Kelem;
    const T* zero = (const T*)octa_zero_page;

    // this workgroup's contiguous tile range
    const int G = gridDim.x;
    // tile sequence of this workgroup: contiguous range, or (strided) every G-th tile so that the chip works on one compact
    // window of G consecutive tiles at a time
    const bool strided = ((exact >> 8) & 255) & 8;
    const int t0 = strided ? (int)blockIdx.x : (int)((int64_t)blockIdx.x * ntiles / G);
    const int tstep = strided ? G : 1;
    const int nmine = strided ? (ntiles - (int)blockIdx.x + G - 1) / G : (int)((int64_t)(blockIdx.x + 1) * ntiles / G) - t0;

    // ---- per-lane patch DMA roles (tile independent)
    int dpy[LP], dpx[LP], dsl[LP];
    bool dok[LP];
#pragma unroll
    for (int i = 0; i < LP; ++i) {
        const int I = i * 8 + wave;
        const int p = I * RPI + lane / SPR;
        const int ps = lane % SPR;
        dsl[i] = (ps ^ res_swz<NCH>(p)) * EPC;      // element offset of the logical slot this lane fetches
        dok[i] = p < PROWS;
        if (TAPS == 9) { dpy[i] = p / PW; dpx[i] = p - dpy[i] * PW; }
        else { dpy[i] = 0; dpx[i] = p; }
    }
    const unsigned sP_base = lds_addr(sP);
    auto issue_patch = [&](int tl, int stage, bool live) {
        int b = 0, y0 = 0, x0 = 0;
        if (TAPS == 9) {
            const int tx = tl % tiles_x;
            const int tq = tl / tiles_x;
            b = tq / tiles_y;
            y0 = (tq - b * tiles_y) * 16;
            x0 = tx * 16;
        }
        const unsigned base = sP_base + (unsigned)(stage * SROWS * RB);
#pragma unroll
        for (int i = 0; i < LP; ++i) {
            bool ok = live & dok[i];
            int off;
            if (TAPS == 9) {
                const int iy = y0 - 1 + dpy[i], ix = x0 - 1 + dpx[i];
                ok = ok & ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W);
                off = ((b * a.H + iy) * a.W + ix) * a.ldx + dsl[i];
            } else {
                const int m = tl * 256 + dpx[i];
                ok = ok & (m < a.M);
                off = m * a.ldx + dsl[i];
            }
            const T* src = ok ? (xg + off) : zero;
            const unsigned dst = (COMPACT && i * 8 + wave >= NINSTR) ? lds_addr(sScr) : base + (unsigned)((i * 8 + wave) * 1024);
            glds16_fast(src, __builtin_amdgcn_readfirstlane(dst));
        }
    };

    // ---- prologue: bias -> LDS, weights -> LDS (once), the first NST-1 patches
    if (t < BNR) sBias[t] = (a.bias && t < (a.upshuffle ? a.CoutT : a.Ng)) ? a.bias[grp * a.Ng + t] : 0.f;    // upshuffle (never grouped): one bias per transposed-conv channel
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {
        const unsigned sW_base = lds_addr(sW);
        for (int e = wave; e < W_INSTR; e += 8) {
            const int E = e * 64 + lane;
            const int step = E / (4 * BNR);
            const int rem = E - step * (4 * BNR);
            const int plane = rem / BNR, sl = rem - plane * BNR;
            // fragment pair (2i, 2i+1), D row rho <-> channel 32*i + 8*(rho >> 2) + 4*(i & 1) + (rho & 3): a lane's two fragments
            // then hold 8 CONSECUTIVE channels of its pixel = one 16-byte store (see the epilogue)
            const int n = TN >= 2 ? ((sl & ~31) + ((sl & 15) >> 2) * 8 + ((sl >> 4) & 1) * 4 + (sl & 3)) : sl;
            const int tap = step / NCH, kc = step - tap * NCH;
            const T* src = (n < a.Ng) ? (wg + (size_t)n * Kelem + tap * a.Cg + kc * 32 + plane * EPC) : zero;
            glds16_fast(src, __builtin_amdgcn_readfirstlane(sW_base + (unsigned)(e * 1024)));
        }
    }
#pragma unroll
    for (int k = 0; k < NST - 1; ++k) issue_patch(t0 + k * tstep, k, k < nmine);

    // fragment row bases of this lane: tile rows wave*2 + j
    int prow0[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) prow0[j] = (wave * TM + j) * PW + r;
    const uint4* const pB0 = sW + q * BNR + r;
    T* __restrict__ yb = (T*)a.y + a.yoff + grp * a.Ng;

    const int dbg = (exact >> 8) & 255;            // tools/convres_micro.py ablations: 1 no stores, 2 no MFMA loop, 4 no patch DMA
    exact &= 255;
    int hist = 0;                                  // bit k: tile (current - 1 - k) issued an unknown number of stores
    // Deferred output stores.  A burst of stores at the end of a tile blocks the wave at store ISSUE until the write path
    // has drained it (measured: store time and MFMA time simply added up).  The packed outputs of tile k-1 therefore stay
    // in registers and go out one instruction at a time between the MFMA steps of tile k.
    constexpr bool DEFER = TN >= 2 && TN <= 4 && !COMPACT;       // NS <= 4 deferred stores (RES_FLUSH below is written out for 4)
    const bool defer = DEFER && a.vec16 && a.Ng % 8 == 0 && a.NgSt == a.Ng && !a.upshuffle;
    // Named scalars, not arrays: anything the optimiser cannot prove constant-indexed is demoted to scratch memory, whose
    // loads come with s_waitcnt vmcnt(0) and drain the whole prefetch ring (measured: 3 us per tile).  Slot n = 2 * row + half.
    uint4 dr0 = {}, dr1 = {}, dr2 = {}, dr3 = {};
    unsigned df0 = ~0u, df1 = ~0u, df2 = ~0u, df3 = ~0u;   // byte offset of the 16-byte piece, ~0u = nothing to store
    bool have = false;
#define RES_ST(DR, DF) { if (DF != ~0u) *(uint4*)((char*)yb + (size_t)DF) = DR; }
#define RES_FLUSH_ALL() { RES_ST(dr0, df0) if constexpr (NP == 2) RES_ST(dr1, df1) RES_ST(dr2, df2) if constexpr (NP == 2) RES_ST(dr3, df3) }
    // store k (of NS) of the previous tile goes out after MFMA step (k + 1) * NSTEP / (NS + 1)
#define RES_AT(k, step) (((k) + 1) * NSTEP / (NS + 1) == (step))
#define RES_FLUSH_STEP(step)                                                     \
    {                                                                            \
        if (RES_AT(0, step)) RES_ST(dr0, df0)                                    \
        if constexpr (NP == 2) {                                                 \
            if (RES_AT(1, step)) RES_ST(dr1, df1)                                \
            if (RES_AT(2, step)) RES_ST(dr2, df2)                                \
            if (RES_AT(3, step)) RES_ST(dr3, df3)                                \
        } else {                                                                 \
            if (RES_AT(1, step)) RES_ST(dr2, df2)                                \
        }                                                                        \
    }
    for (int tl = t0, it = 0; it < nmine; tl += tstep, ++it) {
        const int stage = it % NST;
        // patch `it` has landed once at most the younger patches (+ the known stores) are outstanding
        if (it >= NST && exact && hist == 0) wait_vmcnt<WAIT_ST>();
        else wait_vmcnt<WAIT_BASE>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (!(dbg & 4)) issue_patch(tl + (NST - 1) * tstep, (it + NST - 1) % NST, it + NST - 1 < nmine);

        const char* const pst = (const char*)sP + stage * SROWS * RB;
        // tile geometry (once per tile; the passes below share it)
        int b = 0, y0 = 0, x0 = 0;
        bool full;
        if (TAPS == 9) {
            const int tx = tl % tiles_x;
            const int tq = tl / tiles_x;
            b = tq / tiles_y;
            y0 = (tq - b * tiles_y) * 16;
            x0 = tx * 16;
            full = (y0 + 16 <= a.OH) & (x0 + 16 <= a.OW);
        } else {
            full = tl * 256 + 256 <= a.M;
        }
        hist = ((hist << 1) | (full ? 0 : 1)) & ((1 << NST) - 1);
#pragma unroll
      for (int nh = 0; nh < NH; ++nh) {
        f32x4_t acc[TN][TM];
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (!(dbg & 2)) {
            // fragments of step s+1 are fetched before the MFMAs of step s (two register sets)
            uint4 xf[2][TM], wf[2][TN];
            auto fetch = [&](int step, int buf) {
                const int tap = step / NCH, kc = step - tap * NCH;
                const int kh = tap / 3, kw = tap - kh * 3;
                const int dy = (TAPS == 1) ? 0 : (MODE == 0 ? kh : 2 - kh), dx = (TAPS == 1) ? 0 : (MODE == 0 ? kw : 2 - kw);
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    const int p = prow0[j] + dy * PW + dx;
                    xf[buf][j] = *(const uint4*)(pst + p * RB + (((kc * 4 + q) ^ res_swz<NCH>(p)) << 4));
                }
                const uint4* pB = pB0 + step * 4 * BNR;
#pragma unroll
                for (int i = 0; i < TN; ++i) wf[buf][i] = pB[(nh * TN + i) * 16];
            };
            fetch(0, 0);
#pragma unroll
            for (int step = 0; step < NSTEP; ++step) {
                if (step + 1 < NSTEP) fetch(step + 1, (step + 1) & 1);
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) Mma<T>::run(wf[step & 1][i], xf[step & 1][j], acc[i][j]);
                if constexpr (DEFER) {
                    if (have) RES_FLUSH_STEP(step)
                }
            }
        }
        if (DEFER && (dbg & 2) && have) RES_FLUSH_ALL()

        // epilogue: lane holds, per (i, j), 4 consecutive output channels of one pixel
        if (dbg & 1) { if (acc[0][0][0] == 123.456f) yb[0] = T{}; continue; }
        // bias and activation over the whole accumulator tile first (one activation decision per tile, see act_tile)
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int nb = (TN >= 2 ? (i >> 1) * 32 + q * 8 + (i & 1) * 4 : q * 4) + nh * TN * 16;
            const int bidx = (TAPS == 1 && a.upshuffle) ? nb % a.CoutT : nb;
            const float4 bv = *(const float4*)(sBias + bidx);
#pragma unroll
            for (int j = 0; j < TM; ++j) { acc[i][j][0] += bv.x; acc[i][j][1] += bv.y; acc[i][j][2] += bv.z; acc[i][j][3] += bv.w; }
        }
        act_tile(acc, a.act);
        if constexpr (DEFER) {
            have = defer;
            if (defer) {
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    uint4 own[NP];
#pragma unroll
                    for (int ip = 0; ip < NP; ++ip) {
                        const int chan = ip * 32 + q * 8;
                        float v[8];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const f32x4_t c = acc[2 * ip + h][j];
                            v[4 * h + 0] = c[0]; v[4 * h + 1] = c[1]; v[4 * h + 2] = c[2]; v[4 * h + 3] = c[3];
                        }
                        own[ip] = make_uint4(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]), pack2<T>(v[4], v[5]), pack2<T>(v[6], v[7]));
                    }
                    // transpose through a wave-private LDS strip so that lane L of store h holds bytes [16 L, 16 L + 16) of a
                    // contiguous run of the pixel row (64 lanes x 16 B = 1 KiB): the only store shape that ran near copy speed
                    // (fragment-shaped stores, 16 pixels x 64 B per instruction: 2.8 TB/s; whole lines, lanes 128 B apart: 1.4)
                    constexpr int CPP = NP * 4;                       // 16-byte pieces per pixel
                    if (dbg & 16) {       // experiment: direct fragment-shaped 16-byte stores, no transpose
                        size_t pix; bool ok;
                        if (TAPS == 9) { const int oy = y0 + wave * TM + j, ox = x0 + r; ok = (oy < a.OH) & (ox < a.OW); pix = ((size_t)b * a.OH + oy) * a.OW + ox; }
                        else { const int m = tl * 256 + (wave * TM + j) * 16 + r; ok = m < a.M; pix = (size_t)m; }
                        const unsigned o0 = (ok && q * 8 < a.Ng) ? (unsigned)((pix * a.ldy + q * 8) * 2) : ~0u;
                        const unsigned o1 = (ok && 32 + q * 8 < a.Ng) ? (unsigned)((pix * a.ldy + 32 + q * 8) * 2) : ~0u;
                        if (j == 0) { dr0 = own[0]; df0 = o0; if constexpr (NP == 2) { dr1 = own[1]; df1 = o1; } }
                        else { dr2 = own[0]; df2 = o0; if constexpr (NP == 2) { dr3 = own[1]; df3 = o1; } }
                        continue;
                    }
                    char* const strip = (char*)sT + wave * (TM * 16 * CPP * 16) + j * (16 * CPP * 16);
#pragma unroll
                    for (int ip = 0; ip < NP; ++ip)
                        *(uint4*)(strip + r * (CPP * 16) + (((ip * 4 + q) ^ res_swz<NP>(r)) << 4)) = own[ip];
                }
                if (dbg & 16) continue;
                // read back lane-linear: lane L of piece H of row J is bytes [16 (64 H + L), +16) of that pixel row
#define RES_READBACK(J, H, DR, DF)                                                                                  \
                if constexpr ((H) < NP) {                                                                           \
                    size_t row_pix;                                                                                 \
                    int row_left;                    /* pixels of this row inside the image / the tensor */        \
                    if (TAPS == 9) {                                                                                \
                        const int oy = y0 + wave * TM + (J);                                                        \
                        row_pix = ((size_t)b * a.OH + oy) * a.OW + x0;                                              \
                        row_left = oy < a.OH ? a.OW - x0 : 0;                                                       \
                    } else {                                                                                        \
                        const int m = tl * 256 + (wave * TM + (J)) * 16;                                            \
                        row_pix = (size_t)m;                                                                        \
                        row_left = a.M - m;                                                                         \
                    }                                                                                               \
                    const int Lh = (H) * 64 + lane;                                                                 \
                    const int p = Lh / CPP, c = (Lh % CPP) ^ res_swz<NP>(p);                                        \
                    DR = *(const uint4*)((char*)sT + wave * (TM * 16 * CPP * 16) + (J) * (16 * CPP * 16) + Lh * 16); \
                    DF = (p < row_left && c * 8 < a.Ng) ? (unsigned)(((row_pix + p) * a.ldy + c * 8) * 2) : ~0u;    \
                }
                {
                    constexpr int CPP = NP * 4;
                    RES_READBACK(0, 0, dr0, df0) RES_READBACK(0, 1, dr1, df1) RES_READBACK(1, 0, dr2, df2) RES_READBACK(1, 1, dr3, df3)
                }
                continue;
            }
        }
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            size_t pix;
            int ow = 0, oh = 0, bb = 0;
            if (TAPS == 9) {
                const int oy = y0 + wave * TM + j, ox = x0 + r;
                if (oy >= a.OH || ox >= a.OW) continue;
                pix = ((size_t)b * a.OH + oy) * a.OW + ox;
            } else {
                const int m = tl * 256 + (wave * TM + j) * 16 + r;
                if (m >= a.M) continue;
                pix = (size_t)m;
                if (a.upshuffle) { ow = m % a.OW; const int tq = m / a.OW; oh = tq % a.OH; bb = tq / a.OH; }
            }
#pragma unroll
            for (int ip = 0; ip < NP; ++ip) {
                constexpr int CW = TN >= 2 ? 8 : 4;             // channels per lane per store
                const int nb = (TN >= 2 ? ip * 32 + q * 8 : q * 4) + nh * TN * 16;
                if (nb >= a.NgSt) continue;
                int chan = nb;
                if (TAPS == 1 && a.upshuffle) {
                    const int dd = nb / a.CoutT;
                    chan = nb - dd * a.CoutT;
                    pix = ((size_t)(bb * 2 * a.OH + 2 * oh + (dd >> 1)) * (2 * a.OW) + 2 * ow + (dd & 1));
                }
                float v[CW];
#pragma unroll
                for (int h = 0; h < CW / 4; ++h) {
                    const f32x4_t c = acc[TN >= 2 ? 2 * ip + h : 0][j];
                    v[4 * h + 0] = c[0]; v[4 * h + 1] = c[1]; v[4 * h + 2] = c[2]; v[4 * h + 3] = c[3];
                }
                T* dst = yb + pix * a.ldy + chan;
                if (a.vec16 && nb + CW - 1 < a.Ng) {
                    if constexpr (CW == 8) *(uint4*)dst = make_uint4(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]), pack2<T>(v[4], v[5]), pack2<T>(v[6], v[7]));
                    else *(uint2*)dst = make_uint2(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]));
                } else {
#pragma unroll
                    for (int e = 0; e < CW; ++e) if (nb + e < a.NgSt) DT<T>::st(dst + e, nb + e < a.Ng ? v[e] : 0.f);
                }
            }
        }
      }     // nh
    }
    if constexpr (DEFER) {
        if (have) RES_FLUSH_ALL()
    }
#undef RES_ST
#undef RES_FLUSH_ALL
#undef RES_AT
#undef RES_FLUSH_STEP
#undef RES_READBACK
    wait_vmcnt<0>();        // the look-ahead patches of tiles past the range are still writing LDS
}

// LDS bytes of one instantiation (host mirror of the constants above)
static inline int res_lds_bytes(int taps, int nch, int tn) {
    const int rpi = 64 / (4 * nch), prows = taps == 9 ? 324 : 256;
    const bool compact = taps == 9 && nch == 2 && tn == 4;
    const int lp = (prows + rpi * 8 - 1) / (rpi * 8), srows = compact ? (prows + rpi - 1) / rpi * rpi : lp * 8 * rpi, nst = nch == 1 ? (taps == 9 ? 3 : 4) : 2;
    return taps * nch * 4 * tn * 16 * 16 + nst * srows * 64 * nch + tn * 16 * 4 + (((tn == 2 || tn == 4) && !compact) ? 8 * 2 * 16 * (tn / 2) * 4 * 16 : 16) + (compact ? 1024 : 16);
}

template <typename T, int TAPS, int NCH, int TN, int MODE, int NH = 1>
static void launch_res_one(const ConvArgs& a, int ntiles, int tx, int ty, int exact, hipStream_t st, int groups = 1) {
    const int per_cu = res_lds_bytes(TAPS, NCH, TN * NH) * 2 <= 160 * 1024 ? 2 : 1;
    int grid = octa_num_cus() * per_cu / groups / 8 * 8;      // per (pseudo-)group, a multiple of 8: workgroup x of every group on the same XCD
    if (grid < 8) grid = 8;
    if (const char* e = getenv("OCTA_CONVRES_GRID")) { const int g = atoi(e); if (g > 0) grid = g; }    // tests: many tiles per workgroup
    if (grid > ntiles) grid = ntiles;
    if (const char* e = getenv("OCTA_CONVRES_DBG")) exact |= atoi(e) << 8;
    conv_res_kernel<T, TAPS, NCH, TN, MODE, NH><<<dim3(grid, groups), 512, 0, st>>>(a, ntiles, tx, ty, exact);
}

// eligibility + launch.  Returns false when another kernel must run.
template <typename T>
static bool launch_res(const ConvArgs& a_in, int groups, hipStream_t st, bool forced = true) {
    if constexpr (sizeof(T) != 2) return false;
    else {
        ConvArgs a = a_in;
        if (a.Cg != 32 && a.Cg != 64) return false;
        const bool k3 = a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && !a.upshuffle && a.H == a.OH && a.W == a.OW;
        const bool k1 = a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0;
        if (!k3 && !k1) return false;
        // 3x3 with 128 / 256 output channels per group: every 64-channel slice of the outputs is a pseudo-group of its own (own resident
        // operand, the group's input slice shared): the packed rows, the bias and the output channels of pseudo-group p start at p * 64
        int nsh = 0;
        if (k3 && a.Ng > 64 && (a.Ng == 128 || a.Ng == 256) && a.NgSt == a.Ng && (groups == 1 || a.CgStride == a.Cg)) {
            nsh = a.Ng == 128 ? 1 : 2;
            groups <<= nsh;
            a.Ng = a.NgSt = 64;
        }
        static const bool no_grouped = getenv("OCTA_NO_GROUPED_RES") != nullptr;      // A/B switch
        if (groups != 1 && no_grouped) return false;
        // grouped: whole 64 / 128-byte channel slices per group (the DMA rows and the 16-byte stores stay aligned), 2 .. 32 (pseudo-)groups
        if (groups != 1 && (a.upshuffle || (nsh == 0 && a.CgStride != a.Cg) || a.Ng % 8 != 0 || a.NgSt != a.Ng || groups > 32 || (groups & (groups - 1)) != 0)) return false;
        if ((int64_t)a.B * a.H * a.W * (int64_t)a.ldx >= (1ll << 31)) return false;
        if ((int64_t)a.M * (a.upshuffle ? 4 : 1) * (int64_t)a.ldy >= (1ll << 31)) return false;      // 32-bit byte offsets of the deferred stores
        if (a.upshuffle && a.CoutT % 8 != 0) return false;
        const int nch = a.Cg / 32;
        const int need = a.NgSt;                    // channels the epilogue must cover
        int tn;
        if (k3) {
            tn = need <= 32 ? 2 : 4;
            if (need > 64) return false;
            static const bool no_compact = getenv("OCTA_NO_RES_COMPACT") != nullptr;      // A/B switch
            if (nch == 2 && tn == 4 && (no_compact || !a.vec16)) return false;      // 72 KB of weights + two compact 41 KB patches: the COMPACT instantiation
        } else {
            tn = need <= 16 ? 1 : need <= 32 ? 2 : need <= 64 ? 4 : need <= 128 ? 8 : 16;
            if (need > 256 || (nch == 1 && (tn == 1 || tn >= 8))) return false;
            if (tn >= 8 && !forced) return false;      // (two passes of 8 n-tiles, NH = 2) only when asked for: the autotuner decides
        }
        const int exact = (a.vec16 && a.Ng == tn * 16 && a.NgSt == a.Ng) | (nsh << 16);
        int ntiles, tx = 0, ty = 0;
        if (k3) { tx = cdiv(a.W, 16); ty = cdiv(a.H, 16); ntiles = a.B * tx * ty; }
        else ntiles = cdiv(a.M, 256);
        if (k3) {
            if (a.mode == 0) {
                if (nch == 1 && tn == 4) launch_res_one<T, 9, 1, 4, 0>(a, ntiles, tx, ty, exact, st, groups);
                else if (nch == 1) launch_res_one<T, 9, 1, 2, 0>(a, ntiles, tx, ty, exact, st, groups);
                else if (tn == 4) launch_res_one<T, 9, 2, 4, 0>(a, ntiles, tx, ty, exact, st, groups);
                else launch_res_one<T, 9, 2, 2, 0>(a, ntiles, tx, ty, exact, st, groups);
            } else {
                if (nch == 1 && tn == 4) launch_res_one<T, 9, 1, 4, 1>(a, ntiles, tx, ty, exact, st, groups);
                else if (nch == 1) launch_res_one<T, 9, 1, 2, 1>(a, ntiles, tx, ty, exact, st, groups);
                else if (tn == 4) launch_res_one<T, 9, 2, 4, 1>(a, ntiles, tx, ty, exact, st, groups);
                else launch_res_one<T, 9, 2, 2, 1>(a, ntiles, tx, ty, exact, st, groups);
            }
        } else if (nch == 2) {
            if (tn == 1) launch_res_one<T, 1, 2, 1, 0>(a, ntiles, tx, ty, exact, st, groups);
            else if (tn == 2) launch_res_one<T, 1, 2, 2, 0>(a, ntiles, tx, ty, exact, st, groups);
            else if (tn == 4) launch_res_one<T, 1, 2, 4, 0>(a, ntiles, tx, ty, exact, st, groups);
            else if (tn == 8) launch_res_one<T, 1, 2, 8, 0>(a, ntiles, tx, ty, exact, st, groups);
            else launch_res_one<T, 1, 2, 8, 0, 2>(a, ntiles, tx, ty, exact, st, groups);      // 16 n-tiles as two passes of 8 (no spills)
        } else {
            if (tn == 2) launch_res_one<T, 1, 1, 2, 0>(a, ntiles, tx, ty, exact, st, groups);
            else launch_res_one<T, 1, 1, 4, 0>(a, ntiles, tx, ty, exact, st, groups);
        }
        note_kernel<T>(k3 ? "conv_res3x3_kernel" : "conv_res1x1_kernel", 256, tn * 16);
        return true;
    }
}
