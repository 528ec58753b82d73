// gemm_loop_probe.hip — the steady-state loop of gemm_kernel<float, 64, GLDS> (gemm.hip) with its ingredients switchable, on
// L2-resident operands: which of {LDS fragment reads, LDS-DMA staging, the per-stage wait + barrier} takes the matrix pipe
// from the 0.99 of a pure MFMA stream (mfma_peak.hip) to the 0.86 the GEMM measures.  Results are garbage by design.
// Build: tools/micro/build.sh (hipcc -O3 --offload-arch=gfx950 tools/micro/gemm_loop_probe.hip -o tools/micro/gemm_loop_probe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BM = 128, BN = 128, ROWB = 64, STAGE_BYTES = (BM + BN) * ROWB, SLOTS = 4, SMASK = 3, SSH = 2, RPT = 64, NLD = 2, NQ = 2;
constexpr int PANEL_STAGES = 64;  // the source panel wraps every 64 stages: 128 rows x 4 KiB per operand, L2 resident

// DMA: 0 none; 1 global_load_lds from inline asm with the M0 save / set / restore around every instruction (gemm.hip);
//      2 the same instruction with M0 set ONCE before the loop (every DMA lands at the same LDS address: timing only);
//      3 __builtin_amdgcn_global_load_lds (the compiler manages M0); 4 register staging: global_load_dwordx4 -> ds_write_b128
template <bool LDSREAD, int DMA, bool BARRIER>
__global__ __launch_bounds__(256, 4) void probe(const char* A, const char* W, float* out, int nk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1, half = lane >> 5, l31 = lane & 31;
    const int ps = tid & SMASK, lr = tid / SLOTS, ls = ps ^ ((lr >> SSH) & SMASK);
    const long rowbytes = (long)PANEL_STAGES * ROWB;
    const char* a_ptr[NLD];
    const char* w_ptr[NLD];
    for (int i = 0; i < NLD; ++i) {
        a_ptr[i] = A + (long)(lr + RPT * i) * rowbytes + ls * 16;
        w_ptr[i] = W + (long)(lr + RPT * i) * rowbytes + ls * 16;
    }
    const int swz = (l31 >> SSH) & SMASK;
    const int a_row0 = (wr * 64 + l31) * ROWB, w_row0 = BM * ROWB + (wc * 64 + l31) * ROWB;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // a defined LDS image for the no-DMA variants
    for (int o = tid * 16; o < 2 * STAGE_BYTES; o += 256 * 16) *(uint4*)(smem + o) = make_uint4(0x3f800000u, 0x3f000000u, 0x3e800000u, 0x3e000000u);
    __syncthreads();
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 1024);
    auto dma = [&](const char* gsrc, unsigned dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
    };
    auto dma_nom0 = [&](const char* gsrc) { asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(gsrc) : "memory"); };
    if (DMA == 2) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(lds0) : "memory");
    const int st_off = lr * ROWB + (ps << 4);
    uint4 ga[NLD], gw[NLD];
    auto issue = [&](int kt, int stage) {
        const long kb = (long)(kt % PANEL_STAGES) * ROWB;
        const unsigned sa = lds0 + stage * STAGE_BYTES, sw = sa + BM * ROWB;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            if (DMA == 1 || (DMA == 5 && !(kt & 1)) || (DMA == 6 && !(kt & 3))) {  // 5 / 6: half / a quarter of the traffic per MFMA
                dma(a_ptr[i] + kb, sa + i * 4096);
                dma(w_ptr[i] + kb, sw + i * 4096);
            } else if (DMA == 2) {
                dma_nom0(a_ptr[i] + kb);
                dma_nom0(w_ptr[i] + kb);
            } else if (DMA == 3) {
                char* base = smem + stage * STAGE_BYTES + wave * 1024;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_ptr[i] + kb),
                                                 (__attribute__((address_space(3))) void*)(base + i * 4096), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_ptr[i] + kb),
                                                 (__attribute__((address_space(3))) void*)(base + BM * ROWB + i * 4096), 16, 0, 0);
            } else if (DMA == 4) {
                ga[i] = *(const uint4*)(a_ptr[i] + kb);
                gw[i] = *(const uint4*)(w_ptr[i] + kb);
            }
        }
    };
    auto store_regs = [&](int stage) {
        char* sa = smem + stage * STAGE_BYTES;
        char* sw = sa + BM * ROWB;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            *(uint4*)(sa + st_off + i * RPT * ROWB) = ga[i];
            *(uint4*)(sw + st_off + i * RPT * ROWB) = gw[i];
        }
    };
    uint4 ca = make_uint4(0x3f800000u + tid, 0x3f000000u, 0x3e800000u, 0x3e000000u), cb = ca;
    auto compute = [&](int stage) {
        const char* st = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int so = ((half * NQ + q) ^ swz) << 4;
            uint4 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = LDSREAD ? *(const uint4*)(st + a_row0 + i * 32 * ROWB + so) : ca;
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = LDSREAD ? *(const uint4*)(st + w_row0 + j * 32 * ROWB + so) : cb;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x16& c = acc[i][j];
                    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fa[i].x), __uint_as_float(fb[j].x), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fa[i].y), __uint_as_float(fb[j].y), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fa[i].z), __uint_as_float(fb[j].z), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fa[i].w), __uint_as_float(fb[j].w), c, 0, 0, 0);
                }
        }
    };
    auto stage_barrier = [&]() {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    if (DMA) issue(0, 0);
    if (BARRIER) stage_barrier();
    for (int kt = 0; kt < nk; ++kt) {
        if (DMA && kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
        compute(kt & 1);
        if (DMA == 4 && kt + 1 < nk) store_regs((kt + 1) & 1);
        if (BARRIER) stage_barrier();
        else if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 12345.678f) out[0] = s;
}

template <typename K>
double run(K kern, int blocks, int nk, const char* A, const char* W, float* d, int lds) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, A, W, d, nk / 8);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, A, W, d, nk);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return (double)blocks * 4 * (double)nk * 32 * 4096 / (ms * 1e-3) / 1e12;
}

int main() {
    char *A, *W;
    float* d;
    const size_t panel = (size_t)128 * PANEL_STAGES * ROWB;
    hipMalloc(&A, panel);
    hipMalloc(&W, panel);
    hipMalloc(&d, 1024);
    hipMemset(A, 0x3c, panel);
    hipMemset(W, 0x3c, panel);
    const int nk = 3000;
    printf("| staging | LDS fragment reads | wait + barrier per stage | workgroups per CU | TFLOP/s |\n|---|---|---|---:|---:|\n");
    for (int wg : {4, 1}) {
        const int blocks = 256 * wg, lds = wg == 1 ? 120 * 1024 : 2 * STAGE_BYTES;
        printf("| none | yes | yes | %d | %.1f |\n", wg, run(probe<true, 0, true>, blocks, nk, A, W, d, lds));
        printf("| LDS-DMA, M0 saved / set / restored per instruction (gemm.hip) | yes | yes | %d | %.1f |\n", wg, run(probe<true, 1, true>, blocks, nk, A, W, d, lds));
        printf("| LDS-DMA, M0 set once (timing only) | yes | yes | %d | %.1f |\n", wg, run(probe<true, 2, true>, blocks, nk, A, W, d, lds));
        printf("| LDS-DMA as gemm.hip, every 2nd stage only (half the bytes per MFMA: a 256x256 tile) | yes | yes | %d | %.1f |\n", wg, run(probe<true, 5, true>, blocks, nk, A, W, d, lds));
        printf("| LDS-DMA as gemm.hip, every 4th stage only (a quarter of the bytes per MFMA) | yes | yes | %d | %.1f |\n", wg, run(probe<true, 6, true>, blocks, nk, A, W, d, lds));
        printf("| LDS-DMA through the builtin | yes | yes | %d | %.1f |\n", wg, run(probe<true, 3, true>, blocks, nk, A, W, d, lds));
        printf("| global_load -> VGPR -> ds_write | yes | yes | %d | %.1f |\n", wg, run(probe<true, 4, true>, blocks, nk, A, W, d, lds));
        printf("| LDS-DMA, M0 per instruction | no | no | %d | %.1f |\n", wg, run(probe<false, 1, false>, blocks, nk, A, W, d, lds));
        printf("| LDS-DMA, M0 set once | no | no | %d | %.1f |\n", wg, run(probe<false, 2, false>, blocks, nk, A, W, d, lds));
        fflush(stdout);
    }
    return 0;
}
