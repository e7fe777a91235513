// Fill rate of a CU's LDS from global memory: LDS-DMA (global_load_lds_dwordx4) against register staging (global_load_dwordx4 + ds_write_b128),
// one workgroup per CU, W waves, every wave streaming 1 KB pieces (two 512-byte rows, the dy-row shape of the weight-gradient kernels) of a buffer of F
// bytes that all workgroups of an XCD walk together (F = 2 MB: served by the XCD's L2; 64 MB: beyond it).
// build: hipcc --offload-arch=gfx950 -O3 -o ldsdma_rate tools/micro/ldsdma_rate.hip ; run: ./ldsdma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void glds16(const void* g, unsigned lds) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

// MODE 0: LDS-DMA, DEPTH groups of 4 instructions in flight per wave.  MODE 1: register staging, 8 loads then 8 ds_write_b128, two groups in flight.
template <int MODE, int DEPTH>
__global__ __launch_bounds__(1024) void fill_kernel(const unsigned char* __restrict__ src, size_t fbytes, int iters, int stagger, unsigned* sink) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const unsigned sbase = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
    // piece p of the walk = 1 KB at byte p * 1024 (lane: row lane / 32 of 512 bytes, 16 bytes each); the workgroups of an XCD (blockIdx & 7) walk the same
    // pieces, `stagger` pieces apart
    const size_t npieces = fbytes >> 10;
    size_t p = ((size_t)(blockIdx.x >> 3) * stagger * nw + wave) % npieces;
    const unsigned char* base = src + (size_t)(blockIdx.x & 7) * fbytes + lane * 16;
    if (MODE == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                glds16(base + (p << 10), __builtin_amdgcn_readfirstlane(sbase + (unsigned)(((it % DEPTH) * 4 + k) * nw + wave) * 1024u));
                p += nw; if (p >= npieces) p -= npieces;
            }
            if (it >= DEPTH - 1) wait_vm<(DEPTH - 1) * 4>();
        }
        wait_vm<0>();
    } else {
        uint4 va[8], vb[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { va[k] = *(const uint4*)(base + (p << 10)); p += nw; if (p >= npieces) p -= npieces; }
        for (int it = 0; it < iters; it += 2) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { vb[k] = *(const uint4*)(base + (p << 10)); p += nw; if (p >= npieces) p -= npieces; }
#pragma unroll
            for (int k = 0; k < 8; ++k) *(uint4*)(smem + ((k * nw + wave) * 1024 + lane * 16)) = va[k];
#pragma unroll
            for (int k = 0; k < 8; ++k) { va[k] = *(const uint4*)(base + (p << 10)); p += nw; if (p >= npieces) p -= npieces; }
#pragma unroll
            for (int k = 0; k < 8; ++k) *(uint4*)(smem + (((8 + k) * nw + wave) * 1024 + lane * 16)) = vb[k];
        }
    }
    __syncthreads();
    if (sink && smem[threadIdx.x * 16] == 0x5a && iters < 0) sink[0] = 1;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    const size_t maxf = 64ull << 20;
    unsigned char* buf; CK(hipMalloc(&buf, 8 * maxf)); CK(hipMemset(buf, 1, 8 * maxf));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("LDS fill rate per CU, %d CUs, one workgroup per CU; bytes per wave-iteration: 4 KB (LDS-DMA) / 8 KB (register staging)\n", ncu);
    for (size_t f : {size_t(2) << 20, size_t(64) << 20}) {
        for (int stagger : {0, 7}) {
            for (int waves : {4, 8, 12, 16}) {
                for (int mode = 0; mode < 4; ++mode) {
                    // mode 0 / 1 / 2: LDS-DMA with 2 / 4 / 8 groups of 4 instructions in flight per wave; 3: register staging
                    const int depth = mode == 0 ? 2 : mode == 1 ? 4 : 8;
                    const size_t lds = mode == 3 ? (size_t)16 * waves * 1024 : (size_t)depth * 4 * waves * 1024;
                    if (lds > 160 * 1024) { continue; }
                    const int iters = mode == 3 ? 512 : 1024;                 // 4 MB per wave either way
                    auto launch = [&]() {
                        if (mode == 0) fill_kernel<0, 2><<<ncu, waves * 64, lds>>>(buf, f, iters, stagger, nullptr);
                        else if (mode == 1) fill_kernel<0, 4><<<ncu, waves * 64, lds>>>(buf, f, iters, stagger, nullptr);
                        else if (mode == 2) fill_kernel<0, 8><<<ncu, waves * 64, lds>>>(buf, f, iters, stagger, nullptr);
                        else fill_kernel<1, 2><<<ncu, waves * 64, lds>>>(buf, f, iters, stagger, nullptr);
                    };
                    if (mode == 0) { CK(hipFuncSetAttribute((const void*)fill_kernel<0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); }
                    if (mode == 1) { CK(hipFuncSetAttribute((const void*)fill_kernel<0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); }
                    if (mode == 2) { CK(hipFuncSetAttribute((const void*)fill_kernel<0, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); }
                    if (mode == 3) { CK(hipFuncSetAttribute((const void*)fill_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); }
                    launch(); CK(hipDeviceSynchronize());
                    float best = 1e30f;
                    for (int r = 0; r < 3; ++r) {
                        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
                    }
                    CK(hipGetLastError());
                    const double bytes = (double)waves * (mode == 3 ? (iters + 1) * 8192.0 : iters * 4096.0);
                    printf("footprint %3zu MB per XCD, stagger %d, %2d waves, %-28s in flight %3zu KB: %6.1f GB/s per CU (%5.2f TB/s chip)\n", f >> 20, stagger, waves,
                           mode == 3 ? "register staging (8+8 loads)" : mode == 0 ? "LDS-DMA depth 2x4" : mode == 1 ? "LDS-DMA depth 4x4" : "LDS-DMA depth 8x4",
                           mode == 3 ? (size_t)16 * waves : (size_t)depth * 4 * waves, bytes / (best * 1e-3) / 1e9, bytes * ncu / (best * 1e-3) / 1e12);
                }
            }
        }
    }
    return 0;
}
