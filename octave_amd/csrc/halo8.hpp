// 8-wave 3x3 convolution with a 2-D pixel patch per tile (round 4), bf16 / f16, forward and stride-1 data gradient; included by
// conv.hip behind igemm8.hpp (LDS image of a row, LDS-DMA helpers) -- the big decoder layers and the split-attention 3x3s.
//
// Why.  conv_igemm8_kernel walks a LINEAR tile of 128 pixels through (tap, 64-channel slice) stages: per stage it moves 16 KB of
// activations and 32 KB of weights from L2 into LDS for 4.2 MFLOP, 11.5 KB per MFLOP -- 4.3 GB per launch for the 512 -> 256 layer at
// 100 x 100, i.e. 260 us at the 16.5 TB/s the L2 -> LDS path delivers, against 200 us of MFMA time: the kernel is
// bound by its LDS fill, and the nine tap-shifted copies of the same activation rows are most of what it fetches.  Here a tile is
// a PH x PW patch of output pixels (PH PW <= 256, e.g. 10 x 25) x 128 output channels:
//   * the (PH + 2) x (PW + 2) input patch of a 64-channel slice is staged ONCE (<= 44 KB) and the nine taps read it at shifted
//     LDS rows -- 9x less activation traffic;
//   * 256 pixel rows per tile instead of 128 -- half the weight traffic per FLOP.  4.4 KB per MFLOP in all (2.6x less);
//   * v_mfma_f32_32x32x16, 64 x 64 per wave: 4 ds_read_b128 per 4 MFMAs.
// LDS: a 3-slot weight ring (3 x 16 KB) + two patch buffers (2 x 44 KB) + 1 KB scratch = 137 KB, one workgroup per CU.
// One raw s_barrier per (slice, tap) stage.  Row image of both operands = igemm8's (128-byte rows paired into 256-byte lines, XOR swizzle): a
// DMA instruction fetches 8 full rows, fragment reads of 16 consecutive rows are conflict-free (patch-line wraps cost a 2-way
// conflict now and then).
#pragma once

typedef __attribute__((ext_vector_type(16))) float h8_f32x16_t;

template <typename T> struct Mma32;
template <> struct Mma32<bf16_t> {
    __device__ static __forceinline__ void run(const ig8_u32x4_t& a, const ig8_u32x4_t& b, h8_f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct Mma32<f16_t> {
    __device__ static __forceinline__ void run(const ig8_u32x4_t& a, const ig8_u32x4_t& b, h8_f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};
__device__ __forceinline__ ig8_u32x4_t h8_rd(unsigned addr) {
    ig8_u32x4_t v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
template <unsigned OFF> __device__ __forceinline__ ig8_u32x4_t h8_rdo(unsigned addr) {
    ig8_u32x4_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N> __device__ __forceinline__ void h8_wait(ig8_u32x4_t (&f)[N]) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(f[i]));
}
// counted form: returns once at most CNT LDS operations are outstanding.  LDS operations return in order, so with the CNT youngest
// reads still in flight every older read -- the set `f` -- has arrived (a scalar load sharing the counter only makes it wait longer)
template <int CNT, int N> __device__ __forceinline__ void h8_waitc(ig8_u32x4_t (&f)[N]) {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(CNT) : "memory");
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(f[i]));
}
// byte offset of row R of an image (without the chunk slot) and the slot of chunk c
__device__ __forceinline__ unsigned h8_rowbase(int R) { return (unsigned)(((R >> 4) << 11) + ((R & 7) << 8) + (((R >> 3) & 1) << 7)); }

constexpr int H8_AROWS = 352;                          // patch rows an A buffer holds: (PH + 2) (PW + 2) <= 352
// Where the patch's lines live in an A buffer (round 5).  Row (ry, rx) of the (PH + 2) x (PW + 2) patch sits at image row
// base[ry] + rx.  A linear image (base[ry] = ry (PW + 2)) makes every fragment read of 32 consecutive tile pixels cross a patch line,
// whose +2 row jump puts two of a ds_read_b128 lane group's 16 rows on the same (row mod 16) bank quad: 4 extra cycles on every
// 4-cycle pixel read, 40 % of the kernel's LDS cycles (profiles/r04_halo8_pmc.txt).  The host therefore places line ry at a row
// = PW ry (mod 16): then image row = tile pixel index + a per-tap constant (mod 16) whatever lines the 32 pixels span, and the
// reads are conflict-free.  The lines are packed greedily in residue order (10 x 25: 333 rows instead of 324); the DMA roles and
// fragment addresses are per-lane precomputed values anyway, so the placement costs nothing per stage.
constexpr int H8_MAXLINES = 68;
struct H8Lines { int n, rows; unsigned base[H8_MAXLINES]; };
constexpr int H8_A_BYTES = H8_AROWS * 128;             // 45056 = 44 DMA instructions of 1 KiB
constexpr int H8_A_IPW = 6;                            // patch DMA instructions per wave and slice (44 / 8, rounded up)

// 8 waves as 4 (M) x 2 (N), 64 x 64 per wave: tile = 256 patch pixels x 128 output channels.  MODE 0 forward, 1 data gradient
// (taps flipped).  (A 2 x 4 arrangement with 128 x 64 per wave -- 256 channels per tile -- needs 128 accumulator + 48 fragment
// registers and spilled; it moves the same bytes per FLOP anyway: the patch is shared by the nine taps, so the weights dominate
// the fill and their bytes per FLOP depend on the PIXEL count of a tile only.)
//
// Pipeline (per wave; a stage = one tap of one 64-channel slice = 4 k16-steps of 4 MFMAs; stage st = 9 slice + tap):
//   * fragments are fetched TWO k16-steps ahead into three rotating register sets (set of (st, ks) = (tap + ks) % 3) and waited
//     for with COUNTED lgkmcnt, so every ds_read has a whole step to return;
//   * one raw s_barrier per stage, in the middle of step 2: before it the wave waits for ITS part of B(st + 1) [and, at tap 8,
//     of the next slice's patch]; behind it every wave has finished reading stage st, so slot tap % 3 of the weight ring is free
//     and receives B(st + 3) during step 3; the next slice's six patch instructions leave in pairs at taps 2, 3, 4;
//   * the nine taps are unrolled: ring slot, fragment set, tap offsets and DMA roles are compile-time, the per-stage scalar and
//     vector bookkeeping is ~40 instructions (the first version spent 160 around its 16 MFMAs and was issue-bound:
//     SQ_ACTIVE_INST_ANY 38 % of the wave cycles against 12 % of MFMA issue, profiles/r04_halo8_pmc.txt).
template <typename T, int MODE>
__global__ __launch_bounds__(512) void conv_halo8_kernel(const ConvArgs a, int PH, int PW, int tiles_x, int tiles_y, const H8Lines lines) {
    constexpr int WNW = 2, MI = 2, NJ = 2, BN = 128, BST = 3;
    constexpr int B_BYTES = BN * 128, B_IPW = BN / 64;
    constexpr int A_OFF = BST * B_BYTES, SCRATCH = A_OFF + 2 * H8_A_BYTES;       // weight ring first: its slot offsets fit ds_read's 16-bit offset field
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SCRATCH + 1024 + H8_MAXLINES * 4];
    unsigned* const ltab = (unsigned*)(smem + SCRATCH + 1024);                    // line table: image row of patch line ry
    for (int i = 0; i < lines.n; ++i) if (threadIdx.x == 0) ltab[i] = lines.base[i];      // (uniform index: scalar loads of the kernel argument)
    __syncthreads();

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WNW, wn = wave % WNW;
    // tail split (as conv_igemm8_kernel's): a 1-D launch of sk_full whole tiles followed by sk_parts workgroups per remaining tile, each
    // over a contiguous range of the 64-channel slices, ending in a raw fp32 store; halo8_splitk_fix_kernel finishes those tiles
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
    int s0 = 0, s1 = Cg >> 6;                              // this workgroup's 64-channel slices
    if (part >= 0) { const int nsl = s1; s0 = part * nsl / a.sk_parts; s1 = (part + 1) * nsl / a.sk_parts; }
    const unsigned Cg2 = (unsigned)Cg * 2u;
    const unsigned long zaddr = (unsigned long)(const void*)octa_zero_page;
    const unsigned long xbase = (unsigned long)((const T*)a.x + a.xoff + g * a.CgStride);
    const unsigned long wbase = (unsigned long)((const T*)a.w + (size_t)g * a.Ng * (size_t)(9 * Cg));
    const unsigned sbase = lds_addr(smem);

    // ---- DMA roles: instruction I of an image covers lines 4 I .. 4 I + 3; lane l writes slot l & 15 of line L = 4 I + (l >> 4),
    // i.e. row (L & 7) + 8 ((l >> 3) & 1) + 16 (L >> 3), chunk (l & 7) ^ (L & 7).  This wave issues I = wave + 8 j.
    const int Lq = lane >> 4, hq = (lane >> 3) & 1;
    const int L7 = (4 * (wave & 1) + Lq) & 7;              // (L & 7): the same for every j (8 j keeps I's parity)
    const int chunk = (lane & 7) ^ L7;
    unsigned aoff[H8_A_IPW];                               // byte offset of (pixel, chunk) from xbase; invalid -> amask bit clear
    unsigned amask = 0;
#pragma unroll
    for (int j = 0; j < H8_A_IPW; ++j) {
        const int I = wave + 8 * j;
        const int L = 4 * I + Lq;
        const int R = (L & 7) + 8 * hq + 16 * (L >> 3);
        int ry = -1, rx = 0;                               // the patch line that owns image row R (none: a hole of the packing)
        for (int q = 0; q < NLINES; ++q) { const unsigned d = (unsigned)R - ltab[q]; if (d < (unsigned)PW2) { ry = q; rx = (int)d; } }
        const int iy = y0 - 1 + ry, ix = x0 - 1 + rx;
        const bool ok = (I < NI) && (ry >= 0) && ((unsigned)iy < (unsigned)a.H) && ((unsigned)ix < (unsigned)a.W);
        aoff[j] = ok ? (unsigned)(((b * a.H + iy) * a.W + ix) * a.ldx + chunk * 8) * 2u : 0u;
        amask |= (ok ? 1u : 0u) << j;
    }
    // weight rows: 64-bit pointer of (row, chunk) at K offset 0, or the zero page with a zero K mask
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
    // patch instruction j of slice `slice` into buffer slice & 1 (instructions beyond the image land in the scratch KiB, so that
    // every wave issues exactly H8_A_IPW of them per slice: the vmcnt waits below count instructions)
    auto dmaA = [&](int j, int slice) {
        const int I = wave + 8 * j;
        const unsigned long src = ((amask >> j) & 1u) ? xbase + aoff[j] + (unsigned)(slice * 128) : zaddr;
        const unsigned dst = I < NI ? sbase + (unsigned)(A_OFF + (slice & 1) * H8_A_BYTES + I * 1024) : sbase + (unsigned)SCRATCH;
        glds16_fast((const void*)src, __builtin_amdgcn_readfirstlane(dst));
    };
    // weight instruction j of the stage at K byte offset koff into ring slot `slot`
    auto dmaB = [&](int j, unsigned koff, int slot) {
        const unsigned long src = wptr[j] + (unsigned long)(koff & wkm[j]);
        glds16_fast((const void*)src, __builtin_amdgcn_readfirstlane(sbase + (unsigned)(slot * B_BYTES + (wave + 8 * j) * 1024)));
    };

    // ---- fragment addresses.  32x32x16: lane (r5 = lane & 31, kq = lane >> 5) supplies 8 consecutive k of row r5: chunk 2 ks + kq of
    // k16-step ks; slot = chunk ^ (R & 7) = ((kq ^ (R & 7)) ^ (ks << 1)), i.e. address(ks) = address(0) ^ (ks << 5).
    const int r5 = lane & 31, kq = lane >> 5;
    unsigned bf0[NJ];                                      // weights: [block], k16-step 0, slot 0 of the ring (slots by immediate offset)
#pragma unroll
    for (int jn = 0; jn < NJ; ++jn) {
        const int nrow = wn * 64 + 32 * jn + r5;
        bf0[jn] = sbase + h8_rowbase(nrow) + (unsigned)((kq ^ (nrow & 7)) << 4);
    }
    unsigned ta[9][MI];                                    // pixels: [tap][block], k16-step 0, patch buffer 0
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        int p = wm * (32 * MI) + 32 * i + r5;
        p = p < NPIX ? p : NPIX - 1;                       // padding rows of the tile: any valid row, never stored
        const int py = p / PW, px = p - py * PW;
        unsigned lb[3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) lb[dy] = ltab[py + dy];         // image rows of patch lines py .. py + 2
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
            const int kh = tp / 3, kw = tp - kh * 3;
            const int R = (int)lb[MODE == 0 ? kh : 2 - kh] + px + 1 + (MODE == 0 ? (kw - 1) : (1 - kw));
            ta[tp][i] = sbase + (unsigned)A_OFF + h8_rowbase(R) + (unsigned)((kq ^ (R & 7)) << 4);
        }
    }

    h8_f32x16_t acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jn = 0; jn < NJ; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;

#define H8_SB __builtin_amdgcn_sched_barrier(0)
    OCTA_STAMP_DECL;
    OCTA_STAMP_BEGIN;
    // prologue: patch 0, B(0), B(1), B(2); then wait for the patch and B(0), fetch the fragments of (0, step 0) and (0, step 1)
#pragma unroll
    for (int j = 0; j < H8_A_IPW; ++j) dmaA(j, s0);
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int j = 0; j < B_IPW; ++j) dmaB(j, (unsigned)u * Cg2 + (unsigned)(s0 * 128), u);      // stages 0..2 = taps 0..2 of the first slice
    wait_vmcnt<2 * B_IPW>();
    __builtin_amdgcn_s_barrier();
    // fragments of a k16-step: [0..MI) pixels, [MI..MI+NJ) weights; three sets
    ig8_u32x4_t f0[4], f1[4], f2[4];
    unsigned cur[MI];                                      // this stage's pixel fragment addresses (step 0)
#pragma unroll
    for (int i = 0; i < MI; ++i) cur[i] = ta[0][i] + (unsigned)((s0 & 1) * H8_A_BYTES);
#pragma unroll
    for (int i = 0; i < MI; ++i) f0[i] = h8_rd(cur[i]);
#pragma unroll
    for (int jn = 0; jn < NJ; ++jn) f0[MI + jn] = h8_rd(bf0[jn]);
#pragma unroll
    for (int i = 0; i < MI; ++i) f1[i] = h8_rd(cur[i] ^ 32u);
#pragma unroll
    for (int jn = 0; jn < NJ; ++jn) f1[MI + jn] = h8_rd(bf0[jn] ^ 32u);
    h8_wait(f0); h8_wait(f1);
    // One k16-step: the 4 MFMAs of fragment set FC; behind the first two the 4 reads of the step two ahead into FN
#define H8_MM(FC, I_, J_) Mma32<T>::run(FC[MI + J_], FC[I_], acc[I_][J_]); 
#define H8_STAGE(TAP, F0, F1, F2)                                                             \
    {                                                                                         \
        constexpr int SLOT = (TAP) % 3, NSLOT = ((TAP) + 1) % 3, NTAP = ((TAP) + 1) % 9, TAP3 = ((TAP) + 3) % 9; \
        constexpr unsigned BO = (unsigned)(SLOT * B_BYTES), NBO = (unsigned)(NSLOT * B_BYTES); \
        unsigned nx[MI];                   /* (st + 1, step 0) pixel addresses */              \
        _Pragma("unroll") for (int i = 0; i < MI; ++i) nx[i] = ta[NTAP][i] + ((TAP) == 8 ? naoffs : aoffs); \
        H8_SB;                                                                                \
        /* step 0: MFMAs on F0 = (st, 0); reads (st, 2) -> F2 */                              \
        H8_MM(F0, 0, 0) F2[0] = h8_rd(cur[0] ^ 64u); F2[1] = h8_rd(cur[1] ^ 64u); H8_SB;     \
        H8_MM(F0, 0, 1) F2[2] = h8_rdo<BO>((bf0[0] ^ 64u)); F2[3] = h8_rdo<BO>((bf0[1] ^ 64u)); H8_SB; \
        H8_MM(F0, 1, 0) H8_SB; H8_MM(F0, 1, 1) H8_SB;                                         \
        h8_waitc<4>(F1); H8_SB;           /* (st, 1), issued during the previous step; this step's 4 reads stay in flight */ \
        /* step 1: MFMAs on F1; reads (st, 3) -> F0 */                                        \
        H8_MM(F1, 0, 0) F0[0] = h8_rd(cur[0] ^ 96u); F0[1] = h8_rd(cur[1] ^ 96u); H8_SB;     \
        H8_MM(F1, 0, 1) F0[2] = h8_rdo<BO>((bf0[0] ^ 96u)); F0[3] = h8_rdo<BO>((bf0[1] ^ 96u)); H8_SB; \
        H8_MM(F1, 1, 0) H8_SB; H8_MM(F1, 1, 1) H8_SB;                                         \
        h8_waitc<4>(F2); H8_SB;           /* (st, 2) */                                       \
        /* step 2: MFMAs on F2.  In its middle: every read of stage st has returned (lgkmcnt 0), this wave's part of B(st + 1) -- and, */ \
        /* whenever it matters, of the next slice's patch -- has landed (younger: B(st + 2), the patch pair of stage st - 1), barrier: */ \
        /* for every wave; slot tap % 3 of the weight ring is free.  Behind it the reads of (st + 1, 0) -> F1. */ \
        H8_MM(F2, 0, 0) H8_SB; H8_MM(F2, 0, 1) H8_SB;                                         \
        h8_wait(F0);                                                                          \
        if ((TAP) >= 7 && lastslice) wait_vmcnt<0>();                                         \
        else if ((TAP) >= 3 && (TAP) <= 5 && !lastslice) wait_vmcnt<B_IPW + 2>();             \
        else wait_vmcnt<B_IPW>();                                                             \
        __builtin_amdgcn_s_barrier();                                                         \
        H8_SB;                                                                                \
        if ((TAP) == 8 && lastslice) {     /* the last stage: nothing left to fetch */        \
            H8_MM(F2, 1, 0) H8_SB; H8_MM(F2, 1, 1) H8_SB;                                     \
            H8_MM(F0, 0, 0) H8_SB; H8_MM(F0, 0, 1) H8_SB; H8_MM(F0, 1, 0) H8_SB; H8_MM(F0, 1, 1) H8_SB; \
        } else {                                                                              \
            H8_MM(F2, 1, 0) F1[0] = h8_rd(nx[0]); F1[1] = h8_rd(nx[1]); H8_SB;                \
            H8_MM(F2, 1, 1) F1[2] = h8_rdo<NBO>(bf0[0]); F1[3] = h8_rdo<NBO>(bf0[1]); H8_SB; \
            /* step 3: MFMAs on F0 = (st, 3); reads (st + 1, 1) -> F2; the DMA instructions of B(st + 3) and the patch pair */ \
            H8_MM(F0, 0, 0) F2[0] = h8_rd(nx[0] ^ 32u); F2[1] = h8_rd(nx[1] ^ 32u); H8_SB;    \
            H8_MM(F0, 0, 1) F2[2] = h8_rdo<NBO>((bf0[0] ^ 32u)); F2[3] = h8_rdo<NBO>((bf0[1] ^ 32u)); H8_SB; \
            H8_MM(F0, 1, 0)                                                                   \
            if (!((TAP) >= 6 && lastslice)) {                                                 \
                const unsigned koff3 = (unsigned)TAP3 * Cg2 + kslice + ((TAP) >= 6 ? 128u : 0u); \
                dmaB(0, koff3, SLOT); dmaB(1, koff3, SLOT);                                   \
            }                                                                                 \
            H8_SB;                                                                            \
            H8_MM(F0, 1, 1)                                                                   \
            if ((TAP) >= 2 && (TAP) <= 4 && !lastslice) { dmaA(2 * ((TAP) - 2), slice + 1); dmaA(2 * ((TAP) - 2) + 1, slice + 1); } \
            H8_SB;                                                                            \
            h8_waitc<4>(F1); H8_SB;       /* (st + 1, 0) for the next stage's first step; (st + 1, 1) stays in flight */ \
        }                                                                                     \
        _Pragma("unroll") for (int i = 0; i < MI; ++i) cur[i] = nx[i];                        \
    }
    for (int slice = s0; slice < s1; ++slice) {
        const bool lastslice = slice + 1 == s1;
        const unsigned aoffs = (unsigned)((slice & 1) * H8_A_BYTES), naoffs = (unsigned)(((slice + 1) & 1) * H8_A_BYTES);
        const unsigned kslice = (unsigned)(slice * 128);
        // (opaque re-definition: otherwise the XOR-ed / offset variants of these 20 loop-invariant addresses are hoisted out of the
        // loop into ~60 more registers and the accumulators spill)
        asm volatile("" : "+v"(bf0[0]), "+v"(bf0[1]), "+v"(ta[0][0]), "+v"(ta[0][1]), "+v"(ta[1][0]), "+v"(ta[1][1]), "+v"(ta[2][0]), "+v"(ta[2][1]),
                          "+v"(ta[3][0]), "+v"(ta[3][1]), "+v"(ta[4][0]), "+v"(ta[4][1]), "+v"(ta[5][0]), "+v"(ta[5][1]), "+v"(ta[6][0]), "+v"(ta[6][1]),
                          "+v"(ta[7][0]), "+v"(ta[7][1]), "+v"(ta[8][0]), "+v"(ta[8][1]));
        // fragment set of (tap, step ks) = (tap + ks) % 3; ring slot = tap % 3 (9 taps per slice: both independent of the slice)
        H8_STAGE(0, f0, f1, f2) H8_STAGE(1, f1, f2, f0) H8_STAGE(2, f2, f0, f1)
        H8_STAGE(3, f0, f1, f2) H8_STAGE(4, f1, f2, f0) H8_STAGE(5, f2, f0, f1)
        H8_STAGE(6, f0, f1, f2) H8_STAGE(7, f1, f2, f0) H8_STAGE(8, f2, f0, f1)
    }
#undef H8_STAGE
#undef H8_MM
#undef H8_SB
    OCTA_STAMP_END(octa_diag_stamps_halo8)

    // ---- epilogue.  D[n][m]: lane l holds pixel m = l & 31 of block i and, for v = 0 .. 15, channel 8 (v >> 2) + 4 (l >> 5) + (v & 3)
    // of block jn: four groups of 4 consecutive channels (8-byte stores).
    if (part >= 0) {
        // partial tile: raw fp32 accumulators [pixel row of the tile][channel of the tile] into this (tile, part)'s slot of the workspace
        float* __restrict__ wsp = a.sk_ws + ((size_t)tail * a.sk_parts + part) * (size_t)(256 * BN);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int p = wm * (32 * MI) + 32 * i + r5;
#pragma unroll
            for (int jn = 0; jn < NJ; ++jn)
#pragma unroll
                for (int vg = 0; vg < 4; ++vg) {
                    const int c = wn * 64 + 32 * jn + 8 * vg + 4 * kq;
                    *(f32x4_t*)(wsp + (size_t)p * BN + c) = (f32x4_t){acc[i][jn][4 * vg], acc[i][jn][4 * vg + 1], acc[i][jn][4 * vg + 2], acc[i][jn][4 * vg + 3]};
                }
        }
        return;
    }
    T* __restrict__ yb = (T*)a.y + a.yoff;
    const int act = a.act;
#pragma unroll
    for (int jn = 0; jn < NJ; ++jn) {
#pragma unroll
        for (int vg = 0; vg < 4; ++vg) {
            const int nb = n0 + wn * 64 + 32 * jn + 8 * vg + 4 * kq;
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.bias)
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[e] = (nb + e < a.Ng) ? a.bias[g * a.Ng + nb + e] : 0.f;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][jn][4 * vg + e] += bv[e];
        }
    }
    // one activation decision per tile (a switch per element compiles to a compare-and-branch chain per element, common.hpp)
#define H8_ACT(EXPR) _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int jn = 0; jn < NJ; ++jn) \
        _Pragma("unroll") for (int e = 0; e < 16; ++e) { const float v = acc[i][jn][e]; acc[i][jn][e] = (EXPR); }
    if (act == OCTA_ACT_RELU) { H8_ACT(v > 0.f ? v : 0.f) }
    else if (act == OCTA_ACT_LEAKY02) { H8_ACT(v > 0.f ? v : 0.2f * v) }
    else if (act == OCTA_ACT_SIGMOID) { H8_ACT(1.f / (1.f + __expf(-v))) }
    else if (act == OCTA_ACT_TANH) { H8_ACT(tanhf(v)) }
#undef H8_ACT
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int p = wm * (32 * MI) + 32 * i + r5;
        const int py = p / PW, px = p - py * PW;
        const int oy = y0 + py, ox = x0 + px;
        if (p >= NPIX || oy >= a.H || ox >= a.W) continue;
        const size_t pix = ((size_t)b * a.H + oy) * a.W + ox;
#pragma unroll
        for (int jn = 0; jn < NJ; ++jn) {
#pragma unroll
            for (int vg = 0; vg < 4; ++vg) {
                const int nb = n0 + wn * 64 + 32 * jn + 8 * vg + 4 * kq;
                if (nb >= a.NgSt) continue;
                T* dst = yb + pix * a.ldy + g * a.Ng + nb;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][jn][4 * vg + e];
                if (a.vec_store && nb + 3 < a.Ng) {
                    *(uint2*)dst = make_uint2(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (nb + e < a.NgSt) DT<T>::st(dst + e, nb + e < a.Ng ? v[e] : 0.f);
                }
            }
        }
    }
}

// Patch shape for an H x W image: PH PW <= 256, (PH + 2) (PW + 2) <= H8_AROWS, fewest tiles (ties: the squarer patch, less halo)
static void halo8_patch(int H, int W, int& PH, int& PW) {
    long best = -1;
    PH = 16; PW = 16;
    for (int pw = 4; pw <= 64; ++pw) {
        int ph = 256 / pw;
        while (ph > 1 && (ph + 2) * (pw + 2) > H8_AROWS) --ph;
        if (ph > H) ph = H;
        const int pwe = pw > W ? W : pw;
        const long tiles = (long)cdiv(H, ph) * cdiv(W, pwe);
        const long halo = (long)(ph + 2) * (pwe + 2);
        const long score = tiles * 100000 + halo;
        if (best < 0 || score < best) { best = score; PH = ph; PW = pwe; }
    }
}

// Line placement of the patch image (see H8Lines): packed = line ry at a row = PW ry (mod 16), lines laid out greedily in the order
// that wastes the fewest rows; falls back to the linear image when the packing does not fit an A buffer (patch widths that are
// multiples of 16 need a pitch of PW + 16) or when octa_tuning_set(6, 0) asks for the old image (A/B runs).
static int g_h8_packed = 1;
static void halo8_lines(int PH, int PW, H8Lines& t) {
    const int nl = PH + 2, PW2 = PW + 2;
    t.n = nl;
    static const bool env_lin = getenv("OCTA_H8_LINEAR") != nullptr;
    if (g_h8_packed && !env_lin && nl <= H8_MAXLINES) {
        bool used[H8_MAXLINES] = {};
        int pos = 0;
        for (int k = 0; k < nl; ++k) {
            int best = -1, bgap = 16;
            for (int ry = 0; ry < nl; ++ry) {
                if (used[ry]) continue;
                const int gap = ((PW * ry - pos) % 16 + 16) % 16;
                if (gap < bgap) { bgap = gap; best = ry; }
            }
            used[best] = true;
            t.base[best] = (unsigned)(pos + bgap);
            pos += bgap + PW2;
        }
        t.rows = pos;
        if (pos <= H8_AROWS) return;
    }
    for (int ry = 0; ry < nl; ++ry) t.base[ry] = (unsigned)(ry * PW2);
    t.rows = nl * PW2;
}

// Finishes the split tiles: y = act(sum of the parts + bias), stored at the patch's pixels.  One thread per (tile pixel row, 4 channels);
// blockIdx.x = tail tile, blockIdx.y = strip of 32 pixel rows.
template <typename T>
__global__ __launch_bounds__(256) void halo8_splitk_fix_kernel(const ConvArgs a, int PH, int PW, int tiles_x, int tiles_y) {
    constexpr int BN = 128;
    int Lp = a.sk_full + blockIdx.x;
    const int g = Lp / a.sk_tpg;
    Lp -= g * a.sk_tpg;
    const int nt = Lp % a.sk_gy, mt = Lp / a.sk_gy;
    const int tx = mt % tiles_x, ty = (mt / tiles_x) % tiles_y, b = mt / (tiles_x * tiles_y);
    const int y0 = ty * PH, x0 = tx * PW, n0 = nt * BN;
    const int tc = threadIdx.x & 31, tr = threadIdx.x >> 5;              // 32 threads x 4 channels per row, 8 rows per pass
    const int nb = n0 + tc * 4;
    if (nb >= a.NgSt) return;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.bias)
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = (nb + e < a.Ng) ? a.bias[g * a.Ng + nb + e] : 0.f;
    const float* __restrict__ wsp = a.sk_ws + (size_t)blockIdx.x * a.sk_parts * (size_t)(256 * BN);
    T* __restrict__ yb = (T*)a.y + a.yoff;
    for (int p = blockIdx.y * 32 + tr; p < (int)(blockIdx.y + 1) * 32; p += 8) {
        const int py = p / PW, px = p - py * PW;
        const int oy = y0 + py, ox = x0 + px;
        if (p >= PH * PW || oy >= a.H || ox >= a.W) continue;
        f32x4_t v = *(const f32x4_t*)(wsp + (size_t)p * BN + tc * 4);
        for (int q = 1; q < a.sk_parts; ++q) {
            const f32x4_t u = *(const f32x4_t*)(wsp + (size_t)q * (256 * BN) + (size_t)p * BN + tc * 4);
            v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
        }
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = act_apply(v[e] + bv[e], a.act);
        T* dst = yb + (((size_t)b * a.H + oy) * a.W + ox) * a.ldy + g * a.Ng + nb;
        if (a.vec_store && nb + 3 < a.Ng) *(uint2*)dst = make_uint2(pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3]));
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (nb + e < a.NgSt) DT<T>::st(dst + e, nb + e < a.Ng ? o[e] : 0.f);
        }
    }
}

// How many parts the tiles behind the last full round are split into (1 = no split): halo8's stages are a (slice, tap) pair, a part
// is a contiguous range of slices and should keep at least two of them (18 stages) to amortise its prologue
static int halo8_parts(const ConvArgs& a, int tiles, int nsl) {
    static const bool off = getenv("OCTA_NO_SPLITK") != nullptr;
    if (off || !a.sk_ws) return 1;
    const int ncu = octa_num_cus();
    const int tail = tiles % ncu;
    if (tail == 0 || tail > ncu / 2) return 1;
    int parts = ncu / tail;
    if (parts > 8) parts = 8;
    if (parts > nsl / 2) parts = nsl / 2;
    if (parts < 2) return 1;
    if ((int64_t)tail * parts * 256 * 128 * 4 > a.sk_cap) return 1;
    return parts;
}

// eligibility + launch.  Returns false when another kernel must run.
template <typename T, int MODE> __global__ void conv_halo16_kernel(const ConvArgs a, int PH, int PW, int tiles_x, int tiles_y, const H8Lines lines);   // halo16.hpp
// shape16: the v_mfma_f32_16x16x32 form of the kernel (halo16.hpp, algo 13) instead of 32x32x16 (algo 12)
template <typename T>
static bool launch_halo8(const ConvArgs& a, int groups, hipStream_t st, bool shape16 = false) {
    if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.upshuffle || a.addend || a.stats) return false;
    if (a.H != a.OH || a.W != a.OW || a.Cg % 64 != 0) return false;
    if ((int64_t)a.B * a.H * a.W * (int64_t)a.ldx >= (1ll << 30)) return false;      // 32-bit byte offsets of the patch rows
    if ((int64_t)groups * a.Ng * 9 * a.Cg >= (1ll << 30)) return false;
    int PH, PW;
    halo8_patch(a.H, a.W, PH, PW);
    static const int env_pw = getenv("OCTA_H8_PW") ? atoi(getenv("OCTA_H8_PW")) : 0, env_ph = getenv("OCTA_H8_PH") ? atoi(getenv("OCTA_H8_PH")) : 0;
    if (env_pw > 0 && env_ph > 0 && env_pw * env_ph <= 256 && (env_pw + 2) * (env_ph + 2) <= H8_AROWS) { PW = env_pw; PH = env_ph; }     // (experiments)
    if (PH + 2 > H8_MAXLINES) return false;
    H8Lines lines;
    halo8_lines(PH, PW, lines);
    const int tiles_y = cdiv(a.H, PH), tiles_x = cdiv(a.W, PW);
    dim3 grid(a.B * tiles_y * tiles_x, cdiv(a.Ng, 128), groups);
    ConvArgs b = a;
    const int tiles = grid.x * grid.y * groups;
    const int parts = (groups == 1 || a.NgSt == a.Ng) ? halo8_parts(a, tiles, a.Cg / 64) : 1;
    const int ntail = tiles % octa_num_cus();
    if (parts > 1) {
        b.sk_parts = parts; b.sk_full = tiles - ntail; b.sk_gy = grid.y; b.sk_tpg = grid.x * grid.y;
        grid = dim3(b.sk_full + ntail * parts, 1, 1);
    }
    if (shape16) {
        if (a.mode == 0) conv_halo16_kernel<T, 0><<<grid, 512, 0, st>>>(b, PH, PW, tiles_x, tiles_y, lines);
        else conv_halo16_kernel<T, 1><<<grid, 512, 0, st>>>(b, PH, PW, tiles_x, tiles_y, lines);
    } else if (a.mode == 0) conv_halo8_kernel<T, 0><<<grid, 512, 0, st>>>(b, PH, PW, tiles_x, tiles_y, lines);
    else conv_halo8_kernel<T, 1><<<grid, 512, 0, st>>>(b, PH, PW, tiles_x, tiles_y, lines);
    if (parts > 1) halo8_splitk_fix_kernel<T><<<dim3(ntail, 8), 256, 0, st>>>(b, PH, PW, tiles_x, tiles_y);
    note_kernel<T>(shape16 ? "conv_halo16_kernel" : "conv_halo8_kernel", 256, 128);
    if (parts > 1) { const size_t l = strlen(g_last_kernel); snprintf(g_last_kernel + l, sizeof(g_last_kernel) - l, "+tail%dx%d", ntail, parts); }
    return true;
}
