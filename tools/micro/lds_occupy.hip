// lds_occupy.hip — diagnostic helper (tools/lds_base_probe.py): workgroups that hold `lds_bytes` of LDS on their CU for `milliseconds`
// and do nothing else, so that the workgroups of a kernel launched beside them get a NON-ZERO LDS base address.
// Build: hipcc -O2 --offload-arch=gfx950 -shared -fPIC tools/micro/lds_occupy.hip -o tools/micro/liblds_occupy.so
#include <hip/hip_runtime.h>
namespace {
// mode 0: sleep; 1: every wave streams 16-byte LDS reads and writes over the held array for the whole time (the CU's LDS pipe is busy
// with FOREIGN traffic while the neighbour's LDS-DMA lands); 2: every wave streams 16-byte global loads (the CU's vector memory path)
__global__ void lds_hold_kernel(long long ticks, int words, int mode, const uint4* g, long gn) {
    extern __shared__ unsigned hold[];
    for (int i = threadIdx.x; i < words; i += blockDim.x) hold[i] = 0x7fc00000u + i;  // (NaN patterns: a stray read would show)
    __syncthreads();
    const long long t0 = wall_clock64();
    unsigned acc = 0;
    if (mode == 1) {
        uint4* h4 = (uint4*)hold;
        const int n4 = words / 4;
        int i = threadIdx.x % n4;
        while (wall_clock64() - t0 < ticks) {
#pragma unroll 8
            for (int r = 0; r < 64; ++r) {
                uint4 v = h4[i];
                v.x += r;
                h4[i] = v;
                acc += v.y;
                i += blockDim.x;
                i = i >= n4 ? i - n4 : i;
                i = i >= n4 ? threadIdx.x % n4 : i;
            }
        }
    } else if (mode == 2) {
        long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) % gn;
        while (wall_clock64() - t0 < ticks) {
#pragma unroll 8
            for (int r = 0; r < 64; ++r) {
                acc += g[i].x;
                i += 65536 + 1;
                i = i >= gn ? i - gn : i;
            }
        }
    } else {
        while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    }
    if (acc == 0x12345u || hold[threadIdx.x % (words > 0 ? words : 1)] == 1u) __builtin_trap();  // keeps the work alive
}
}  // namespace
extern "C" int lds_occupy2(int workgroups, int threads, int lds_bytes, double milliseconds, void* stream, int mode, const void* g, long g_bytes) {
    if (workgroups <= 0 || threads <= 0 || threads > 1024 || lds_bytes < 64 || (mode == 2 && (!g || g_bytes < (1 << 20))) || lds_bytes > 160 * 1024 || !(milliseconds >= 0) || milliseconds > 2000.0) return -1;
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -2;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;
    if (hipFuncSetAttribute((const void*)lds_hold_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -3;
    hipLaunchKernelGGL(lds_hold_kernel, dim3(workgroups), dim3(threads), lds_bytes, (hipStream_t)stream, (long long)(milliseconds * khz), lds_bytes / 4, mode,
                       (const uint4*)g, g_bytes / 16);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
extern "C" int lds_occupy(int workgroups, int threads, int lds_bytes, double milliseconds, void* stream) {
    return lds_occupy2(workgroups, threads, lds_bytes, milliseconds, stream, 0, nullptr, 0);
}
