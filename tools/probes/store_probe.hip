// How fast can ONE workgroup per CU stream stores?  (MI355X; explains the conv epilogue's 2.8 TB/s)
// Build: hipcc --offload-arch=gfx950 -O3 -o store_probe store_probe.hip ; run: ./store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// pattern 0: every wave-instruction writes 1 KiB contiguous (lane-linear 16 B); consecutive instructions of a wave are consecutive KiB
// pattern 1: same, but the 8 waves of a block interleave at 1 KiB granularity (block writes a contiguous region cooperatively)
// pattern 2: fragment-shaped: lane (r = lane & 15, q = lane >> 4) writes 16 B at pixel r * 128 + q * 16 (+64 for odd instructions)
// pattern 3: 4 B per lane, 256 B contiguous per instruction
template <int PATTERN, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void store_kernel(char* __restrict__ out, size_t bytes_per_block, int lds_pad) {
    extern __shared__ char pad[];
    if (lds_pad < 0) pad[threadIdx.x] = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* base = out + (size_t)blockIdx.x * bytes_per_block;
    const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
    if (PATTERN == 0) {
        const size_t per_wave = bytes_per_block / WAVES;
        char* p = base + wave * per_wave + lane * 16;
        for (size_t o = 0; o < per_wave; o += 1024) *(uint4*)(p + o) = v;
    } else if (PATTERN == 1) {
        char* p = base + wave * 1024 + lane * 16;
        for (size_t o = 0; o < bytes_per_block; o += WAVES * 1024) *(uint4*)(p + o) = v;
    } else if (PATTERN == 2) {
        const int r = lane & 15, q = lane >> 4;
        char* p = base + wave * 2048 + r * 128 + q * 16;
        for (size_t o = 0; o < bytes_per_block; o += WAVES * 2048) { *(uint4*)(p + o) = v; *(uint4*)(p + o + 64) = v; }
    } else {
        char* p = base + wave * 256 + lane * 4;
        for (size_t o = 0; o < bytes_per_block; o += WAVES * 256) *(unsigned*)(p + o) = v.x;
    }
}

template <int PATTERN, int WAVES>
static void run(char* buf, size_t total, int blocks, int lds, const char* name) {
    const size_t per_block = total / blocks / (WAVES * 2048) * (WAVES * 2048);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)store_kernel<PATTERN, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    store_kernel<PATTERN, WAVES><<<blocks, WAVES * 64, lds>>>(buf, per_block, 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) store_kernel<PATTERN, WAVES><<<blocks, WAVES * 64, lds>>>(buf, per_block, 0);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 100.0;
    printf("%-44s blocks %4d waves/blk %2d lds %6d : %7.1f us  %5.2f TB/s\n", name, blocks, WAVES, lds, us, per_block * (double)blocks / us * 1e-6);
}

// tile loop: per iteration every wave issues K stores of 1 KiB, then s_waitcnt vmcnt(WAITN) [+ s_barrier] [+ SLEEP x s_sleep 8]
template <int K, int WAITN, int BARRIER, int SLEEP>
__global__ __launch_bounds__(512) void tile_kernel(char* __restrict__ out, size_t bytes_per_block) {
    extern __shared__ char pad[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* p = out + (size_t)blockIdx.x * bytes_per_block + wave * (K * 1024) + lane * 16;
    const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
    for (size_t o = 0; o < bytes_per_block; o += 8 * K * 1024) {
#pragma unroll
        for (int k = 0; k < K; ++k) *(uint4*)(p + o + k * 1024) = v;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");
        if (BARRIER) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int s = 0; s < SLEEP; ++s) __builtin_amdgcn_s_sleep(8);
    }
}
template <int K, int WAITN, int BARRIER, int SLEEP>
static void run_tile(char* buf, size_t total, const char* name) {
    const int blocks = 256;
    const size_t per_block = total / blocks / (8 * K * 1024) * (8 * K * 1024);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)tile_kernel<K, WAITN, BARRIER, SLEEP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    tile_kernel<K, WAITN, BARRIER, SLEEP><<<blocks, 512, 100 * 1024>>>(buf, per_block);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) tile_kernel<K, WAITN, BARRIER, SLEEP><<<blocks, 512, 100 * 1024>>>(buf, per_block);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 100.0;
    printf("%-40s K %d wait vmcnt(%2d) barrier %d sleep %2d : %7.1f us  %5.2f TB/s\n", name, K, WAITN, BARRIER, SLEEP, us, per_block * (double)blocks / us * 1e-6);
}

int main() {
    const size_t total = 328ull * 1000 * 1000;
    char* buf;
    CK(hipMalloc(&buf, total + (1 << 20)));
    for (int lds : {100 * 1024, 70 * 1024, 0}) {
        const int per_cu = lds == 0 ? 4 : lds > 80 * 1024 ? 1 : 2;
        run<0, 8>(buf, total, 256 * per_cu, lds, "p0 1KiB/instr, wave-private streams");
        run<1, 8>(buf, total, 256 * per_cu, lds, "p1 1KiB/instr, block-interleaved");
        run<2, 8>(buf, total, 256 * per_cu, lds, "p2 fragment-shaped 16 x 64 B");
        run<3, 8>(buf, total, 256 * per_cu, lds, "p3 dword/lane 256 B/instr");
        run<1, 4>(buf, total, 256 * per_cu, lds, "p1 4 waves");
        run<1, 16>(buf, total, 256 * per_cu, lds, "p1 16 waves");
    }
    run<1, 4>(buf, total, 4096, 0, "p1 4 waves, 4096 blocks");
    run<1, 4>(buf, total, 16384, 0, "p1 4 waves, 16384 blocks");
    run_tile<4, 0, 0, 0>(buf, total, "tile loop");
    run_tile<4, 0, 1, 0>(buf, total, "tile loop");
    run_tile<4, 4, 1, 0>(buf, total, "tile loop");
    run_tile<4, 12, 1, 0>(buf, total, "tile loop");
    run_tile<4, 12, 0, 0>(buf, total, "tile loop");
    run_tile<4, 12, 1, 4>(buf, total, "tile loop");
    run_tile<4, 12, 1, 16>(buf, total, "tile loop");
    run_tile<4, 12, 1, 32>(buf, total, "tile loop");
    run_tile<4, 40, 1, 32>(buf, total, "tile loop");
    run_tile<2, 6, 1, 16>(buf, total, "tile loop");
    return 0;
}
