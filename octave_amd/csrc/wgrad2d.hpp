// wgrad2d: weight gradient of 3x3 stride-1 pad-1 layers with a 2-D PIXEL PATCH shared by the nine taps (round 5; included by wgrad8.hip).
//
// dW[co][kh][kw][ci] = sum over output pixels of dy[pixel][co] * x[pixel + (kh - 1, kw - 1)][ci].  wgrad9 streams, per 32-pixel stage and 256 x 256
// tile, 16 KB of dy and 16 KB of tap-shifted x rows: 7.6 KB of L2 -> LDS fill per MFLOP, its second roofline.  Here a tile is
// 256 output channels x (9 taps x 32 input channels) and a stage is a 5 x 25 patch of output pixels: 64 KB of dy rows + the 7 x 27 input patch of the
// 32-channel slice (12 KB), shared by all nine taps: 4.0 KB per MFLOP.
//
// 768 threads = 12 waves (three per SIMD, <= 168 registers): wave (wm, kh) owns output channels 64 wm .. 64 wm + 63 (two 32-row MFMA blocks) of kernel
// row kh (three taps = three 32-column blocks): 6 accumulators of v_mfma_f32_32x32x16.  A stage is 8 k16 steps (16 tile pixels each; the tile's
// 125 pixels are padded to 128 with zero rows); per step a wave reads 2 dy^T blocks + 3 x^T blocks with ds_read_b64_tr_b16 (10 reads for 6 MFMAs),
// double-buffered one step ahead.  Tile pixel p sits in row p of the dy image (wgrad9's 512-byte rows and XOR involution) and reads patch row
// prow(p) + 27 kh + kw of the x image (64-byte rows, lane-linear: the four rows x two 32-byte halves a 32-lane half touches cover the 64 banks once
// whenever the four rows are consecutive, i.e. always except across a tile-row wrap); prow(p) comes from a per-lane byte table (4 registers), kw is an
// immediate offset.  Two LDS slots of 76 KB: the DMA of stage s + 1 is issued right behind the barrier that opens stage s (one barrier per stage:
// a stage is ~4 600 cycles of MFMA per SIMD, the fill's latency hides behind it).
// Only exact geometries: H % 5 == 0, W % 25 == 0 (25 / 50 / 100: the decoder 3x3 layers), Cin/groups % 32 == 0, no fused bias.
// Measured (profiles/r05_wgrad2d.txt): the MFMA + read loop runs 1.16 PFLOP/s, the LDS-DMA alone takes longer than wgrad9's whole launch -- with two
// slots the DMA queue is empty at every stage boundary and the path needs ~1 us from issue to landing: 526 us against 421 on the 50 x 50 layer.
// Off by default (octa_tuning_set(10, 1) / OCTA_WGRAD2D=1); the fix is a ring of four half-images of dy (lead 5).
struct Wg2dArgs {
    const unsigned short* x; const unsigned short* dy; float* dw;
    int B, H, W;
    int Cg, Ng, groups;
    int ldx, xoff, ldy, yoff;
    long s_o, s_i, s_h, s_w;
    int tilesN, tilesC;
    int ty, tx, npatch;
    int parts, ppp;            // M-split: workgroup part handles patches [part * ppp, min(npatch, (part + 1) * ppp))
    float* partws; long part_slice;      // optional private partial tiles (deterministic mode / fold): part p stores to partws + p * part_slice
    int Kpad;                  // 9 * Cg (layout of the partial tiles: [g * Ng + n][Kpad])
    int abl;                   // timing-only ablations (OCTA_WG2D_ABL; results are wrong): 1 = no DMA after the prologue, 2 = no MFMA, 4 = no fragment reads, 8 = no epilogue
};

template <int F16>
__global__ __launch_bounds__(768) void wgrad2d_kernel(const Wg2dArgs a) {
    constexpr int PH = 5, PW = 25, NPIX = PH * PW, PR = PW + 2, PROWS = (PH + 2) * PR;     // 125 tile pixels, 27-pixel patch rows, 189 patch rows
    constexpr int DYIMG = 128 * 512, PIMG = 192 * 64, SLOT = DYIMG + PIMG;                 // 64 KB + 12 KB
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * SLOT];

    const int total = gridDim.x, Lb = blockIdx.x;
    const int xcd = Lb & 7, jq = Lb >> 3, qn = total >> 3, rn = total & 7;
    int bid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + jq;
    const int tc = bid % a.tilesC; bid /= a.tilesC;
    const int tn = bid % a.tilesN; bid /= a.tilesN;
    const int g = bid % a.groups;
    const int part = bid / a.groups;
    const int n0 = tn * 256, c0 = tc * 32;
    const int pbeg = part * a.ppp, pend = min(a.npatch, pbeg + a.ppp);
    if (pbeg >= pend) return;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave & 3, kh = wave >> 2;
    const unsigned long zaddr = (unsigned long)(const void*)wg8_zero_page;
    const unsigned sbase = wg_lds_addr(smem);
    const int H = a.H, W = a.W, ldx = a.ldx, ldy = a.ldy, Ng = a.Ng;

    // ---- DMA roles.  dy image: instruction I = wave + 12 k (k = 0 .. 5, I < 64) fills rows 2 I, 2 I + 1 (32 lanes of 16 bytes each);
    // (2 I + lane / 32) & 3 does not depend on k, so the involution's source chunk is one value per lane.
    const int dhalf = lane >> 5, dpos = lane & 31;
    const int p0 = 2 * wave + dhalf;                                   // tile pixel (= image row) of instruction k = 0; + 24 per k
    const int df_ = (p0 & 3) << 1;
    const int dchunk = (((dpos >> 1) ^ df_) << 1) | (dpos & 1);
    const bool nvalid = (n0 + dchunk * 8) < Ng;
    const unsigned long dyb = (unsigned long)(a.dy + (a.yoff + g * Ng + n0 + dchunk * 8));
    // x patch: instruction `wave` fills patch rows 16 wave + lane / 4 (4 lanes of 16 bytes = the 32-channel slice)
    const int q = 16 * wave + (lane >> 2);
    const int qy = q / PR, qx = q - qy * PR;
    const bool qrow_ok = q < PROWS;
    const unsigned long xb = (unsigned long)(a.x + (a.xoff + g * a.Cg + c0 + (lane & 3) * 8));

    int pidx = pbeg;
    int pb = pidx / (a.ty * a.tx);
    int prem = pidx - pb * (a.ty * a.tx);
    int pyi = prem / a.tx, pxi = prem - pyi * a.tx;
    // One DMA instruction each.  The dy instructions of a stage are issued one per k16 step of the PREVIOUS stage (issue_dy, k = 0 .. 5) and the
    // patch instruction behind step 6, not as one burst behind the barrier: the LDS-DMA path of a CU sustains ~25-40 GB/s with a ~1 us
    // issue-to-landed latency (MI355X_MICROARCH.md, ldsdma-fill), and a 76 KB burst followed by a wait left it idle most of the stage
    // (4.3 us per stage with nothing else in the loop).
    int nbase = 0, npy = 0, npx = 0;                                   // next stage: first pixel of its patch; (row, column) of this lane's next dy row
    auto begin_next = [&](int b, int yi, int xi) { nbase = (b * H + yi * PH) * W + xi * PW; npy = p0 / PW; npx = p0 - (p0 / PW) * PW; };
    auto issue_dy = [&](int k, int slot) {
        if (wave + 12 * k < 64) {
            const bool ok = nvalid && (p0 + 24 * k) < NPIX;
            const unsigned long src = ok ? dyb + (unsigned long)((unsigned)(nbase + npy * W + npx) * (unsigned)(ldy * 2)) : zaddr;   // (host: tensor < 4 GB)
            wg_glds16((const void*)src, __builtin_amdgcn_readfirstlane(sbase + (unsigned)(slot * SLOT + (wave + 12 * k) * 1024)));
        }
        if (npx >= 1) { npy += 1; npx -= 1; } else { npx += 24; }      // + 24 pixels: one row down, one column back
    };
    auto issue_patch = [&](int slot, int b, int yi, int xi) {
        const int y = yi * PH - 1 + qy, xx = xi * PW - 1 + qx;
        const bool ok = qrow_ok && (unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W;
        const unsigned long src = ok ? xb + (unsigned long)((unsigned)((b * H + y) * W + xx) * (unsigned)(ldx * 2)) : zaddr;
        wg_glds16((const void*)src, __builtin_amdgcn_readfirstlane(sbase + (unsigned)(slot * SLOT + DYIMG + wave * 1024)));
    };

    // ---- fragment addressing (slot 0, step 0)
    const int r = lane & 15, gq = lane >> 4;
    const int frow = 8 * (gq >> 1) + (r >> 2);
    const int fr = ((r >> 2) & 3) << 1;
    const int cb = (r & 3) * 8;
    unsigned abase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) abase[i] = sbase + (unsigned)(frow * 512 + ((((wm * 2 + i) * 2 + (gq & 1)) ^ fr) << 5) + cb);
    const unsigned pbase = sbase + (unsigned)(DYIMG + kh * PR * 64 + (gq & 1) * 32 + cb);
    // patch row of the tile pixels this lane reads: step s, read h -> pixel 16 s + frow + 4 h; one byte each.  Pixels >= 125 (the padding of the
    // last step) read patch row 0: their dy rows are zero, and row 0 + 27 kh + kw is real, finite data (a row index past the image would not be)
    unsigned tab[4];
#pragma unroll
    for (int w4 = 0; w4 < 4; ++w4) {
        unsigned v = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = w4 * 4 + e, s = idx >> 1, h = idx & 1;
            const int p = 16 * s + frow + 4 * h;
            const int pr = p < NPIX ? (p / PW) * PR + (p % PW) : 0;
            v |= (unsigned)pr << (8 * e);
        }
        tab[w4] = v;
    }

    wg_f32x16_t acc[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    begin_next(pb, pyi, pxi);
#pragma unroll
    for (int k = 0; k < 6; ++k) issue_dy(k, 0);
    issue_patch(0, pb, pyi, pxi);
    wg_wait_vmcnt<0>();

    wg_u32x2_t aX[2][2], bX[3][2], aY[2][2], bY[3][2];
#define WG2D_LOAD(AF, BF, S, so)                                                                                                            \
    if (!(a.abl & 4)) {                                                                                                                     \
        const unsigned tw_ = tab[(S) >> 1];                                                                                                 \
        const unsigned r0_ = (tw_ >> (16 * ((S) & 1))) & 0xffu, r1_ = (tw_ >> (16 * ((S) & 1) + 8)) & 0xffu;                                \
        const unsigned b0_ = pbase + (so) + r0_ * 64u, b1_ = pbase + (so) + r1_ * 64u;                                                      \
        AF[0][0] = wg_tr<(S) * 8192>(abase[0] + (so)); AF[0][1] = wg_tr<(S) * 8192 + 2048>(abase[0] + (so));                                \
        AF[1][0] = wg_tr<(S) * 8192>(abase[1] + (so)); AF[1][1] = wg_tr<(S) * 8192 + 2048>(abase[1] + (so));                                \
        BF[0][0] = wg_tr<0>(b0_); BF[0][1] = wg_tr<0>(b1_);                                                                                 \
        BF[1][0] = wg_tr<64>(b0_); BF[1][1] = wg_tr<64>(b1_);                                                                               \
        BF[2][0] = wg_tr<128>(b0_); BF[2][1] = wg_tr<128>(b1_);                                                                             \
    }
#define WG2D_WAIT(AF, BF)                                                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                                     \
                 : "+v"(AF[0][0]), "+v"(AF[0][1]), "+v"(AF[1][0]), "+v"(AF[1][1]), "+v"(BF[0][0]), "+v"(BF[0][1]), "+v"(BF[1][0]), "+v"(BF[1][1]), \
                   "+v"(BF[2][0]), "+v"(BF[2][1]) :: "memory")
#define WG2D_MMA1(AF, BF, i, j)                                                                                                             \
    if (!(a.abl & 2)) WgMma32<F16>::run(make_uint4(AF[i][0].x, AF[i][0].y, AF[i][1].x, AF[i][1].y), make_uint4(BF[j][0].x, BF[j][0].y, BF[j][1].x, BF[j][1].y), acc[i][j])
#define WG2D_MMA6(AF, BF)                                                                                                                   \
    _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                                                           \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) WG2D_MMA1(AF, BF, i, j)
#define WG2D_SB __builtin_amdgcn_sched_barrier(0)
    for (int st = 0;; ++st) {
        const unsigned so = (unsigned)((st & 1) * SLOT);
        const int ns = (st + 1) & 1;
        __builtin_amdgcn_s_barrier();                  // stage st has landed (every wave waited for its part), stage st - 1 is read out
        ++pidx;
        const bool has_next = pidx < pend && !(a.abl & 1);
        if (++pxi == a.tx) { pxi = 0; if (++pyi == a.ty) { pyi = 0; ++pb; } }
        if (has_next) begin_next(pb, pyi, pxi);
        WG2D_SB;
        // (reads of step s + 1, then the six MFMAs of step s: interleaving one read pair behind every MFMA, wgrad9's recipe, measured 15 % SLOWER
        // here -- three waves per SIMD already fill each other's gaps)
        WG2D_LOAD(aX, bX, 0, so); WG2D_WAIT(aX, bX); WG2D_SB;
        WG2D_LOAD(aY, bY, 1, so); if (has_next) issue_dy(0, ns); WG2D_SB; WG2D_MMA6(aX, bX); WG2D_SB; WG2D_WAIT(aY, bY); WG2D_SB;
        WG2D_LOAD(aX, bX, 2, so); if (has_next) issue_dy(1, ns); WG2D_SB; WG2D_MMA6(aY, bY); WG2D_SB; WG2D_WAIT(aX, bX); WG2D_SB;
        WG2D_LOAD(aY, bY, 3, so); if (has_next) issue_dy(2, ns); WG2D_SB; WG2D_MMA6(aX, bX); WG2D_SB; WG2D_WAIT(aY, bY); WG2D_SB;
        WG2D_LOAD(aX, bX, 4, so); if (has_next) issue_dy(3, ns); WG2D_SB; WG2D_MMA6(aY, bY); WG2D_SB; WG2D_WAIT(aX, bX); WG2D_SB;
        WG2D_LOAD(aY, bY, 5, so); if (has_next) issue_dy(4, ns); WG2D_SB; WG2D_MMA6(aX, bX); WG2D_SB; WG2D_WAIT(aY, bY); WG2D_SB;
        WG2D_LOAD(aX, bX, 6, so); if (has_next) issue_dy(5, ns); WG2D_SB; WG2D_MMA6(aY, bY); WG2D_SB; WG2D_WAIT(aX, bX); WG2D_SB;
        WG2D_LOAD(aY, bY, 7, so); if (has_next) issue_patch(ns, pb, pyi, pxi); WG2D_SB; WG2D_MMA6(aX, bX); WG2D_SB; WG2D_WAIT(aY, bY); WG2D_SB;
        WG2D_MMA6(aY, bY); WG2D_SB;
        wg_wait_vmcnt<0>();                            // this wave's part of stage st + 1
        if (pidx >= pend) break;
    }
#undef WG2D_MMA1
#undef WG2D_LOAD
#undef WG2D_WAIT
#undef WG2D_MMA6
#undef WG2D_SB

    // ---- epilogue: block (i, j): column ci = lane & 31 of tap (kh, j), rows co = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
    if (a.abl & 8) { if (acc[0][0][0] == 123.456f) a.dw[0] = 1.f; return; }
    const int lc = lane & 31, lh = lane >> 5;
    const int ci = c0 + lc;
    if (a.partws) {
        float* const part_ = a.partws + (long)part * a.part_slice;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int k = (kh * 3 + j) * a.Cg + ci;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int n = n0 + (wm * 2 + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    if (n < Ng) part_[(long)(g * Ng + n) * a.Kpad + k] = acc[i][j][e];
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const long koff = (long)ci * a.s_i + (long)kh * a.s_h + (long)j * a.s_w;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + (wm * 2 + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (n < Ng) atomicAdd(a.dw + (long)(g * Ng + n) * a.s_o + koff, acc[i][j][e]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ host side
static int g_wg2d = 0;                  // octa_tuning_set(10, 0 / 1): 3x3 stride-1 weight gradients of exact 5 x 25 geometries on wgrad2d
static int wg2d_on() {
    static const int env = getenv("OCTA_WGRAD2D") ? atoi(getenv("OCTA_WGRAD2D")) : -1;
    return env >= 0 ? env : g_wg2d;
}
static bool wg2d_eligible(const octa_wgrad_job& j) {
    const octa_conv_desc& d = j.d;
    if (!wg2d_on() || octa_deterministic()) return false;
    if (d.dtype != OCTA_BF16 && d.dtype != OCTA_F16) return false;
    if (d.upshuffle || !j.x || !j.dy || !j.dw || j.dbias) return false;
    if (d.KH != 3 || d.KW != 3 || d.stride != 1 || d.pad != 1 || d.OH != d.H || d.OW != d.W) return false;
    if (d.H % 5 || d.W % 25) return false;
    const int Cg = d.Cin / d.groups, Ng = d.Cout / d.groups;
    if (Cg % 32 || d.cin_g_pad != Cg || Ng % 8 || Ng < 128) return false;
    if (d.ldy % 8 || d.yoff % 8 || d.ldx % 8 || d.xoff % 8) return false;
    const int64_t px = (int64_t)d.B * d.H * d.W;
    if (px * d.ldx * 2 >= (1ll << 32) || px * d.ldy * 2 >= (1ll << 32)) return false;      // 32-bit byte offsets in the DMA roles
    for (int i = 0; i < 4; ++i) if (j.dw_strides[i] < 0) return false;
    return true;
}
static int wg2d_launch(const octa_wgrad_job& j, hipStream_t st) {
    const octa_conv_desc& d = j.d;
    Wg2dArgs a;
    a.x = (const unsigned short*)j.x; a.dy = (const unsigned short*)j.dy; a.dw = j.dw;
    a.B = d.B; a.H = d.H; a.W = d.W;
    a.groups = d.groups; a.Cg = d.Cin / d.groups; a.Ng = d.Cout / d.groups;
    a.ldx = d.ldx; a.xoff = d.xoff; a.ldy = d.ldy; a.yoff = d.yoff;
    a.s_o = (long)j.dw_strides[0]; a.s_i = (long)j.dw_strides[1]; a.s_h = (long)j.dw_strides[2]; a.s_w = (long)j.dw_strides[3];
    a.tilesN = cdiv(a.Ng, 256); a.tilesC = a.Cg / 32;
    a.ty = d.H / 5; a.tx = d.W / 25; a.npatch = d.B * a.ty * a.tx;
    a.partws = nullptr; a.part_slice = 0; a.Kpad = 9 * a.Cg;
    a.abl = getenv("OCTA_WG2D_ABL") ? atoi(getenv("OCTA_WG2D_ABL")) : 0;
    const int64_t tiles = (int64_t)a.groups * a.tilesN * a.tilesC;
    // M-split: minimise rounds x (stages + E) over the number of parts; E ~ the epilogue (96 KB of float atomics per workgroup) + prologue in stages
    static const int E = getenv("OCTA_WG2D_EPI") ? atoi(getenv("OCTA_WG2D_EPI")) : 8;
    const int ncu = wg_num_cus();
    int best_parts = 1;
    int64_t best = -1;
    for (int parts = 1; parts <= a.npatch && parts <= 256; ++parts) {
        const int ppp = cdiv(a.npatch, parts);
        if (ppp < 8 && parts > 1) break;
        const int64_t rounds = (tiles * cdiv(a.npatch, ppp) + ncu - 1) / ncu;
        const int64_t cost = rounds * (ppp + E);
        if (best < 0 || cost < best) { best = cost; best_parts = parts; }
    }
    a.ppp = cdiv(a.npatch, best_parts);
    a.parts = cdiv(a.npatch, a.ppp);
    const int64_t nblk = tiles * a.parts;
    if (nblk <= 0 || nblk >= (1ll << 30)) OCTA_FAIL(OCTA_ERR_BAD_ARG, "wgrad2d: bad grid %lld", (long long)nblk);
    if (d.dtype == OCTA_F16) wgrad2d_kernel<1><<<(unsigned)nblk, 768, 0, st>>>(a);
    else wgrad2d_kernel<0><<<(unsigned)nblk, 768, 0, st>>>(a);
    OCTA_CHECK_LAUNCH("wgrad2d");
    octa_note_conv_kernel(d.dtype == OCTA_F16 ? "wgrad2d_kernel<f16,256x9x32>" : "wgrad2d_kernel<bf16,256x9x32>");
    return OCTA_OK;
}
