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
    int rot, rot2;             // start offset (patches) per ci-tile / per co-tile inside the part
};

// up to four layers per launch: blocks [start[j], start[j] + nblk of job j) work on job j (start[j] is a multiple of 8, so blockIdx & 7 labels the
// same XCD inside every job); one resident workgroup per CU means a second launch could only begin when the first one's last workgroup had ended --
// in one launch the next job's workgroups follow on each CU as it frees up
#define WG2D_MAXJOBS 4
struct Wg2dBatch { int n; int start[WG2D_MAXJOBS + 1]; int nblk[WG2D_MAXJOBS]; Wg2dArgs job[WG2D_MAXJOBS]; };
template <int F16>
__global__ __launch_bounds__(768) void wgrad2d_kernel(const Wg2dBatch batch) {
    int jb = 0;
#pragma unroll
    for (int k = 1; k < WG2D_MAXJOBS; ++k) if (k < batch.n && (int)blockIdx.x >= batch.start[k]) jb = k;
    const Wg2dArgs& a = batch.job[jb];
    constexpr int PH = 5, PW = 25, NPIX = PH * PW, PR = PW + 2, PROWS = (PH + 2) * PR;     // 125 tile pixels, 27-pixel patch rows, 189 patch rows
    constexpr int DYHALF = 64 * 512, PIMG = 192 * 64, POFF = 4 * DYHALF;                   // ring of four 32 KB half-images of dy, then two 12 KB patch slots
    __shared__ __attribute__((aligned(1024))) unsigned char smem[4 * DYHALF + 2 * PIMG];

    const int total = batch.nblk[jb], Lb = (int)blockIdx.x - batch.start[jb];
    if (Lb >= total) return;                                           // (padding blocks between two jobs)
    const int xcd = Lb & 7, jq = Lb >> 3, qn = total >> 3, rn = total & 7;
    int bid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + jq;
    const int tc = bid % a.tilesC; bid /= a.tilesC;
    const int tn = bid % a.tilesN; bid /= a.tilesN;
    const int g = bid % a.groups;
    const int part = bid / a.groups;
    const int n0 = tn * 256, c0 = tc * 32;
    const int pbeg = part * a.ppp, pend = min(a.npatch, pbeg + a.ppp);
    const int nst = pend - pbeg;                                       // stages (= patches) of this workgroup
    if (nst <= 0) return;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave & 3, kh = wave >> 2;
    const bool dywave = wave < 8;                                      // waves 0-7 fill the dy ring, waves 8-11 the patch slots
    const unsigned long zaddr = (unsigned long)(const void*)wg8_zero_page;
    const unsigned sbase = wg_lds_addr(smem);
    const int H = a.H, W = a.W, ldx = a.ldx, ldy = a.ldy, Ng = a.Ng;

    // ---- DMA roles: one register set for both (a spill here is reloaded by scratch_load, whose compiler-placed s_waitcnt vmcnt(0) would wait for every
    // LDS-DMA in flight).  dy half-image (64 rows): instruction I = wave + 8 k (k = 0 .. 3) of a dy wave fills local rows 2 I, 2 I + 1;
    // (2 I + lane / 32) & 3 does not depend on k, so the involution's source chunk is one value per lane.  x patch: instruction I = (wave - 8) + 4 j
    // (j = 0 .. 2) of a patch wave fills patch rows 16 I + lane / 4 (4 lanes of 16 bytes = the slice)
    unsigned long gbase;               // the lane's first byte of row 0 of its tensor
    int r0;                            // its row in instruction 0: + 16 per k (dy), + 64 per j (patch)
    bool lane_ok = true;
    if (dywave) {
        const int dpos = lane & 31;
        r0 = 2 * wave + (lane >> 5);
        const int df_ = (r0 & 3) << 1;
        const int dchunk = (((dpos >> 1) ^ df_) << 1) | (dpos & 1);
        lane_ok = (n0 + dchunk * 8) < Ng;
        gbase = (unsigned long)(a.dy + (a.yoff + g * Ng + n0 + dchunk * 8));
    } else {
        r0 = 16 * (wave & 3) + (lane >> 2);
        gbase = (unsigned long)(a.x + (a.xoff + g * a.Cg + c0 + (lane & 3) * 8));
    }
    const unsigned ldy2 = (unsigned)ldy * 2u, ldx2 = (unsigned)ldx * 2u;

    // The workgroups of one part read the same dy tiles and patches; walking them in lockstep puts every CU of the XCD on the same L2 lines at the
    // same moment, so each ci-tile starts `rot` patches further on and wraps round inside the part
    struct Pc { int b, y, x, i; };
    auto at = [&](int i) { Pc c; c.i = i; c.b = i / (a.ty * a.tx); const int rem = i - c.b * (a.ty * a.tx); c.y = rem / a.tx; c.x = rem - c.y * a.tx; return c; };
    const Pc cbeg = at(pbeg);
    auto adv = [&](Pc c) {
        if (++c.i == pend) return cbeg;
        if (++c.x == a.tx) { c.x = 0; if (++c.y == a.ty) { c.y = 0; ++c.b; } }
        return c;
    };
    const Pc c0p = at(pbeg + (tc * a.rot + tn * a.rot2) % nst);
    Pc c1 = adv(c0p), c2 = adv(c1);

    // one half-image of dy: 4 instructions of this (dy) wave.  p / 25 = (p * 41) >> 10 for p < 128
    auto issue_dy_half = [&](Pc c, int h, int slot) {
        const int base = (c.b * H + c.y * PH) * W + c.x * PW;
        int rr = r0;
        asm volatile("" : "+v"(rr));                                   // recompute the rows' offsets here: hoisted out of the loop they cost 8 registers, and spills
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = 64 * h + 16 * k + rr, py = (p * 41) >> 10, px = p - py * PW;
            const bool ok = lane_ok && p < NPIX;
            const unsigned long src = ok ? gbase + (unsigned long)((unsigned)(base + py * W + px) * ldy2) : zaddr;            // (host: tensor < 4 GB)
            wg_glds16((const void*)src, __builtin_amdgcn_readfirstlane(sbase + (unsigned)(slot * DYHALF + (wave + 8 * k) * 1024)));
        }
    };
    // the 7 x 27 patch of a stage: 3 instructions of this (patch) wave.  q / 27 = (q * 19) >> 9 for q < 192
    auto issue_patch = [&](Pc c, int slot) {
        int rr = r0;
        asm volatile("" : "+v"(rr));
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int q = rr + 64 * j, qy = (q * 19) >> 9, qx = q - qy * PR;
            const int y = c.y * PH - 1 + qy, xx = c.x * PW - 1 + qx;
            const bool ok = q < PROWS && (unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W;
            const unsigned long src = ok ? gbase + (unsigned long)((unsigned)((c.b * H + y) * W + xx) * ldx2) : zaddr;
            wg_glds16((const void*)src, __builtin_amdgcn_readfirstlane(sbase + (unsigned)(POFF + slot * PIMG + ((wave & 3) + 4 * j) * 1024)));
        }
    };

    // ---- fragment addressing (ring slot 0 / patch slot 0, step 0 of a half)
    const int r = lane & 15, gq = lane >> 4;
    const int frow = 8 * (gq >> 1) + (r >> 2);
    const int fr = ((r >> 2) & 3) << 1;
    const int cb = (r & 3) * 8;
    unsigned abase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) abase[i] = sbase + (unsigned)(frow * 512 + ((((wm * 2 + i) * 2 + (gq & 1)) ^ fr) << 5) + cb);
    const unsigned pbase = sbase + (unsigned)(POFF + kh * PR * 64 + (gq & 1) * 32 + cb);
    // patch row of the tile pixels this lane reads: step s (0 .. 7 over the stage), read h -> pixel 16 s + frow + 4 h; one byte each.  Pixels >= 125
    // (the padding of the last step) read patch row 0: their dy rows are zero, and row 0 + 27 kh + kw is real, finite data
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

    // ---- prologue: halves (0, 0), (0, 1), (1, 0) and the patch of stage 0; half (0, 0) and the patch must have landed
    if (dywave) {
        issue_dy_half(c0p, 0, 0);
        issue_dy_half(c0p, 1, 1);
        if (nst > 1) { issue_dy_half(c1, 0, 2); wg_wait_vmcnt<8>(); } else wg_wait_vmcnt<4>();
    } else {
        issue_patch(c0p, 0);
        wg_wait_vmcnt<0>();
    }

    wg_u32x2_t aX[2][2], bX[3][2], aY[2][2], bY[3][2];
    // S8: step of the stage (0 .. 7, selects the patch rows); the dy rows are those of step S8 & 3 of the half-image at byte offset `dof`
#define WG2D_LOAD(AF, BF, S8, dof, pof)                                                                                                      \
    {                                                                                                                                       \
        const unsigned tw_ = tab[(S8) >> 1];                                                                                                \
        const unsigned r0_ = (tw_ >> (16 * ((S8) & 1))) & 0xffu, r1_ = (tw_ >> (16 * ((S8) & 1) + 8)) & 0xffu;                              \
        const unsigned b0_ = pbase + (pof) + r0_ * 64u, b1_ = pbase + (pof) + r1_ * 64u;                                                    \
        AF[0][0] = wg_tr<((S8) & 3) * 8192>(abase[0] + (dof)); AF[0][1] = wg_tr<((S8) & 3) * 8192 + 2048>(abase[0] + (dof));                \
        AF[1][0] = wg_tr<((S8) & 3) * 8192>(abase[1] + (dof)); AF[1][1] = wg_tr<((S8) & 3) * 8192 + 2048>(abase[1] + (dof));                \
        BF[0][0] = wg_tr<0>(b0_); BF[0][1] = wg_tr<0>(b1_);                                                                                 \
        BF[1][0] = wg_tr<64>(b0_); BF[1][1] = wg_tr<64>(b1_);                                                                               \
        BF[2][0] = wg_tr<128>(b0_); BF[2][1] = wg_tr<128>(b1_);                                                                             \
    }
#define WG2D_WAIT(AF, BF)                                                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                                     \
                 : "+v"(AF[0][0]), "+v"(AF[0][1]), "+v"(AF[1][0]), "+v"(AF[1][1]), "+v"(BF[0][0]), "+v"(BF[0][1]), "+v"(BF[1][0]), "+v"(BF[1][1]), \
                   "+v"(BF[2][0]), "+v"(BF[2][1]) :: "memory")
#define WG2D_MMA6(AF, BF)                                                                                                                   \
    _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                                                           \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                                       \
            WgMma32<F16>::run(make_uint4(AF[i][0].x, AF[i][0].y, AF[i][1].x, AF[i][1].y), make_uint4(BF[j][0].x, BF[j][0].y, BF[j][1].x, BF[j][1].y), acc[i][j])
#define WG2D_SB __builtin_amdgcn_sched_barrier(0)

    OCTA_STAMP_DECL;
    OCTA_STAMP_BEGIN;
    for (int S = 0; S < nst; ++S) {
        const unsigned d0 = (unsigned)(((2 * S) & 3) * DYHALF), d1 = d0 + DYHALF;       // ring slots of this stage's halves ({0, 1} or {2, 3})
        const unsigned po = (unsigned)((S & 1) * PIMG);
        // ---- half 0.  Barrier: half (S, 0) and the patch of S have landed (every wave waited for its part); half (S - 1, 1) is read out, so its
        // ring slot takes half (S + 1, 1) and the other patch slot the patch of S + 1
        __builtin_amdgcn_s_barrier();
        if (S + 1 < nst) {
            if (dywave) issue_dy_half(c1, 1, (2 * S + 3) & 3);
            else issue_patch(c1, (S + 1) & 1);
        }
        WG2D_SB;
        WG2D_LOAD(aX, bX, 0, d0, po); WG2D_WAIT(aX, bX); WG2D_SB;
        WG2D_LOAD(aY, bY, 1, d0, po); WG2D_SB; WG2D_MMA6(aX, bX); WG2D_SB; WG2D_WAIT(aY, bY); WG2D_SB;
        WG2D_LOAD(aX, bX, 2, d0, po); WG2D_SB; WG2D_MMA6(aY, bY); WG2D_SB; WG2D_WAIT(aX, bX); WG2D_SB;
        WG2D_LOAD(aY, bY, 3, d0, po); WG2D_SB; WG2D_MMA6(aX, bX); WG2D_SB; WG2D_WAIT(aY, bY); WG2D_SB;
        WG2D_MMA6(aY, bY); WG2D_SB;
        // half (S, 1) must have landed: behind it in this wave's queue are the halves (S + 1, 0) and (S + 1, 1), four instructions each
        if (dywave) { if (S + 1 < nst) wg_wait_vmcnt<8>(); else wg_wait_vmcnt<0>(); }
        // ---- half 1.  Barrier: half (S, 0) is read out: its ring slot takes half (S + 2, 0)
        __builtin_amdgcn_s_barrier();
        if (dywave && S + 2 < nst) issue_dy_half(c2, 0, (2 * S + 4) & 3);
        WG2D_SB;
        WG2D_LOAD(aX, bX, 4, d1, po); WG2D_WAIT(aX, bX); WG2D_SB;
        WG2D_LOAD(aY, bY, 5, d1, po); WG2D_SB; WG2D_MMA6(aX, bX); WG2D_SB; WG2D_WAIT(aY, bY); WG2D_SB;
        WG2D_LOAD(aX, bX, 6, d1, po); WG2D_SB; WG2D_MMA6(aY, bY); WG2D_SB; WG2D_WAIT(aX, bX); WG2D_SB;
        WG2D_LOAD(aY, bY, 7, d1, po); WG2D_SB; WG2D_MMA6(aX, bX); WG2D_SB; WG2D_WAIT(aY, bY); WG2D_SB;
        WG2D_MMA6(aY, bY); WG2D_SB;
        // half (S + 1, 0) and the patch of S + 1 must have landed: behind the former are the halves (S + 1, 1) and (S + 2, 0)
        if (dywave) {
            if (S + 2 < nst) wg_wait_vmcnt<8>(); else if (S + 1 < nst) wg_wait_vmcnt<4>(); else wg_wait_vmcnt<0>();
        } else wg_wait_vmcnt<0>();
        c1 = c2; c2 = adv(c2);
    }
    OCTA_STAMP_END(octa_diag_stamps_wgrad9);
#undef WG2D_LOAD
#undef WG2D_WAIT
#undef WG2D_MMA6
#undef WG2D_SB

    // ---- epilogue: block (i, j): column ci = lane & 31 of tap (kh, j), rows co = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
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
// octa_tuning_set(10, v): 3x3 stride-1 weight gradients of exact 5 x 25 geometries on wgrad2d: 0 = never, 1 = every geometry the kernel takes,
// 2 (default) = where taking the layer out of the batched wgrad9 launch is measured to pay (ungrouped, >= 256 input channels, OCTA_WG2D_MINC, and
// >= 256 output channels: the decoder's three big 3x3 layers, in ONE launch of their own; profiles/r05_wgrad2d_ring.txt (f), (h))
static int g_wg2d = 2;
static int wg2d_on() {
    static const int env = getenv("OCTA_WGRAD2D") ? atoi(getenv("OCTA_WGRAD2D")) : -1;
    return env >= 0 ? env : g_wg2d;
}
static bool wg2d_eligible(const octa_wgrad_job& j) {
    const octa_conv_desc& d = j.d;
    const int mode = wg2d_on();
    if (!mode || octa_deterministic()) return false;
    static const int minc = getenv("OCTA_WG2D_MINC") ? atoi(getenv("OCTA_WG2D_MINC")) : 256;
    if (mode == 2 && (d.groups != 1 || d.Cout < 256 || d.Cin < minc)) return false;
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
static int wg2d_plan(const octa_wgrad_job& j, Wg2dArgs& a, int& nblk_out) {
    const octa_conv_desc& d = j.d;
    a.x = (const unsigned short*)j.x; a.dy = (const unsigned short*)j.dy; a.dw = j.dw;
    a.B = d.B; a.H = d.H; a.W = d.W;
    a.groups = d.groups; a.Cg = d.Cin / d.groups; a.Ng = d.Cout / d.groups;
    a.ldx = d.ldx; a.xoff = d.xoff; a.ldy = d.ldy; a.yoff = d.yoff;
    a.s_o = (long)j.dw_strides[0]; a.s_i = (long)j.dw_strides[1]; a.s_h = (long)j.dw_strides[2]; a.s_w = (long)j.dw_strides[3];
    a.tilesN = cdiv(a.Ng, 256); a.tilesC = a.Cg / 32;
    a.ty = d.H / 5; a.tx = d.W / 25; a.npatch = d.B * a.ty * a.tx;
    a.partws = nullptr; a.part_slice = 0; a.Kpad = 9 * a.Cg;
    static const int rot = getenv("OCTA_WG2D_ROT") ? atoi(getenv("OCTA_WG2D_ROT")) : 1, rot2 = getenv("OCTA_WG2D_ROT2") ? atoi(getenv("OCTA_WG2D_ROT2")) : 0;
    a.rot = rot; a.rot2 = rot2;
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
    if (nblk <= 0 || nblk >= (1ll << 26)) OCTA_FAIL(OCTA_ERR_BAD_ARG, "wgrad2d: bad grid %lld", (long long)nblk);
    nblk_out = (int)nblk;
    return OCTA_OK;
}
// the eligible jobs of a batch (one dtype), WG2D_MAXJOBS per launch
static int wg2d_launch(const octa_wgrad_job* const* jobs, int n, int f16, hipStream_t st) {
    for (int i0 = 0; i0 < n; i0 += WG2D_MAXJOBS) {
        Wg2dBatch b;
        b.n = n - i0 < WG2D_MAXJOBS ? n - i0 : WG2D_MAXJOBS;
        int at = 0;
        for (int k = 0; k < WG2D_MAXJOBS; ++k) {
            b.start[k] = at;
            b.nblk[k] = 0;
            if (k < b.n) {
                const int rc = wg2d_plan(*jobs[i0 + k], b.job[k], b.nblk[k]);
                if (rc) return rc;
                at += (b.nblk[k] + 7) & ~7;
            } else b.job[k] = b.job[0];
        }
        b.start[WG2D_MAXJOBS] = at;
        if (f16) wgrad2d_kernel<1><<<(unsigned)at, 768, 0, st>>>(b);
        else wgrad2d_kernel<0><<<(unsigned)at, 768, 0, st>>>(b);
        OCTA_CHECK_LAUNCH("wgrad2d");
    }
    octa_note_conv_kernel(f16 ? "wgrad2d_kernel<f16,256x9x32>" : "wgrad2d_kernel<bf16,256x9x32>");
    return OCTA_OK;
}
