// stream_order_probe.hip — library-independent check of in-order execution of ONE stream's kernels while several streams are busy.
// Every stream owns a buffer; kernel i of a stream checks that every element holds i (what kernel i - 1 wrote) and writes i + 1.
// In-order streams make a mismatch impossible whatever the other streams do.  Kernel shapes imitate the encoder's mix: "tile" kernels
// (one 512-thread workgroup per CU with 112 / 128 KiB of LDS walking the buffer, 16-byte accesses) alternate with "row" kernels (many
// 256-thread workgroups, no LDS, 8-byte accesses), over sub-ranges whose length depends on (stream, step) so that the streams drift.
// Build: hipcc -O2 --offload-arch=gfx950 tools/micro/stream_order_probe.hip -o tools/micro/stream_order_probe
// Run:   tools/micro/stream_order_probe [streams=8] [steps=400] [mb=64] [dma=0]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

// DMA: the element reaches LDS by global_load_lds_dwordx4 + s_waitcnt vmcnt(0) + barrier — the 16-bit tile kernels' staging
template <bool DMA>
__global__ __launch_bounds__(512) void tile_step(uint4* buf, long n4, unsigned expect, unsigned write, unsigned long long* err) {
    extern __shared__ uint4 stage[];
    unsigned bad = 0;
    const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)stage + (threadIdx.x >> 6) * 1024);
    for (long i0 = (long)blockIdx.x * 512; i0 < n4; i0 += (long)gridDim.x * 512) {  // (n4 % 512 == 0: whole workgroup passes)
        const long i = i0 + threadIdx.x;
        uint4 v;
        if (DMA) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(buf + i), "s"(lds_dst) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            v = stage[threadIdx.x];
        } else {
            v = buf[i];
            stage[threadIdx.x] = v;  // (through LDS like an operand tile)
        }
        __syncthreads();
        const uint4 w = stage[threadIdx.x ^ 1];
        bad += (v.x != expect) + (v.y != expect) + (v.z != expect) + (v.w != expect) + (w.x != expect);
        __syncthreads();
        buf[i] = make_uint4(write, write, write, write);
    }
    if (bad) atomicAdd(err, (unsigned long long)bad);
}
__global__ __launch_bounds__(256) void row_step(uint2* buf, long n2, unsigned expect, unsigned write, unsigned long long* err) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n2) return;
    const uint2 v = buf[i];
    const unsigned bad = (v.x != expect) + (v.y != expect);
    buf[i] = make_uint2(write, write);
    if (bad) atomicAdd(err, (unsigned long long)bad);
}

int main(int argc, char** argv) {
    const int NS = argc > 1 ? atoi(argv[1]) : 8, STEPS = argc > 2 ? atoi(argv[2]) : 400;
    const long MB = argc > 3 ? atol(argv[3]) : 64;
    const long n = MB << 18;  // u32 elements
    CK(hipFuncSetAttribute((const void*)tile_step<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CK(hipFuncSetAttribute((const void*)tile_step<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    const bool dma = argc > 4 && atoi(argv[4]) != 0;
    int cus = 256;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    std::vector<hipStream_t> st(NS);
    std::vector<unsigned*> buf(NS);
    unsigned long long* err;
    CK(hipMalloc(&err, NS * 2 * sizeof(unsigned long long)));
    CK(hipMemset(err, 0, NS * 2 * sizeof(unsigned long long)));
    for (int s = 0; s < NS; ++s) {
        CK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking));
        CK(hipMalloc(&buf[s], n * 4));
        CK(hipMemset(buf[s], 0, n * 4));
    }
    CK(hipDeviceSynchronize());
    // every step covers the WHOLE buffer (so that expect / write stay uniform) in one kernel whose shape depends on (s, i)
    for (int i = 0; i < STEPS; ++i)
        for (int s = 0; s < NS; ++s) {
            const int kind = (i * 7 + s * 3) % 5;
            if (kind < 3) {
                const int lds = kind == 0 ? 128 * 1024 : (kind == 1 ? 112 * 1024 : 8192);
                const int grid = kind == 2 ? cus * 3 : ((i + s) % 3 == 0 ? cus / 2 + 8 * s : cus);
                hipLaunchKernelGGL(dma ? tile_step<true> : tile_step<false>, dim3(grid), dim3(512), lds, st[s], (uint4*)buf[s], n / 4, (unsigned)i, (unsigned)(i + 1), err + 2 * s);
            } else {
                hipLaunchKernelGGL(row_step, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, st[s], (uint2*)buf[s], n / 2, (unsigned)i, (unsigned)(i + 1), err + 2 * s + 1);
            }
        }
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(NS * 2);
    CK(hipMemcpy(h.data(), err, NS * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long tot = 0;
    printf("{\"lds_dma\": %d, \"streams\": %d, \"steps\": %d, \"mb_per_stream\": %ld, \"mismatches (tile kernels, row kernels) per stream\": [", (int)dma, NS, STEPS, MB);
    for (int s = 0; s < NS; ++s) {
        printf("%s[%llu, %llu]", s ? ", " : "", h[2 * s], h[2 * s + 1]);
        tot += h[2 * s] + h[2 * s + 1];
    }
    printf("], \"total\": %llu}\n", tot);
    return 0;
}
