// gemm16_loop_probe.hip — the steady-state K loop of gemm16_big_kernel<bf16, 128, 128, 2, 2, 4> (gemm16.hip: 256x256 tile, 8 waves
// as 2 x 4, wave tile 128 x 64, two LDS stages of 64 k, 32 MFMAs per wave between two barriers) with its ingredients switchable:
// which of {LDS fragment reads, LDS-DMA staging (and where its bytes come from), the per-stage drain + barrier} takes the matrix
// pipe from the 2.13 PF of a pure v_mfma_f32_32x32x16_bf16 stream (profiles/r02_mfma_peak.md) to the 1.1-1.2 PF the kernel's K
// loop measures (profiles/r03_gemm16_probes.md).  Results are garbage by design.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/gemm16_loop_probe.hip -o tools/micro/gemm16_loop_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int BN = 256, NTHR = 512, PASS_BYTES = NTHR * 16;
constexpr int PANEL_BYTES = 4096;  // bytes of K per row of the source panel: 256 rows x 4 KiB per operand = 1 MiB + 1 MiB, then it wraps

// DMA: 0 none; 1 as gemm16.hip (global_load_lds_dwordx4 from inline asm, M0 saved / set / restored around every instruction, the 8
//      pieces of the next stage issued behind the fragment reads of the first two q steps); 2 M0 set once before the loop (every
//      piece lands at the same LDS address: timing only); 3 as 1 but all 8 pieces up front; 5 as 1 on every 2nd stage only (half the
//      bytes per MFMA); 6 as 1 spread over all four q steps (they then land later than the barrier would like)
// PRIV: every workgroup streams its OWN panel (512 MiB in all: HBM / Infinity Cache) instead of one shared, L2-resident panel
// ROWB / NST: bytes of K per row and stage, LDS stages (128 / 2 = the product; 64 / 3, 64 / 4 = rings with two / three K steps in
// flight and counted vmcnt, the pieces spread over the whole step); a_stride / w_stride: byte offset between the panels of
// consecutive workgroups (0 = shared)
// MI: 32-row accumulator blocks per wave (4: the 256 x 256 tile, 3: the 192 x 256 one — same W traffic, 3/4 of the MFMAs)
// PP (round 5, DMA == 1 only): which fragment steps carry a wave's pieces — 0: steps 0 / 1 for every wave; 1: waves 0-3 steps 0 / 1,
// waves 4-7 (their SIMD partners) steps 2 / 3; 2: 0 / 2 and 1 / 3; 3: 0 / 1 and 1 / 2 (gemm16.hip's PP)
template <bool LDSREAD, int DMA, bool BARRIER, int ROWB = 128, int NST = 2, int MI = 4, int PP = 0>
__global__ __launch_bounds__(512, 2) void probe(const char* A, const char* W, float* out, int nk, long a_stride, long w_stride) {
    constexpr int BM = 64 * MI;
    constexpr int A_BYTES = BM * ROWB, STAGE_BYTES = (BM + BN) * ROWB;
    constexpr int SLOTS = ROWB / 16, SMASK = SLOTS - 1, SSH = ROWB == 128 ? 1 : 2, RPP = NTHR / SLOTS, NLA = BM / RPP, NLB = BN / RPP, NL = NLA + NLB;
    constexpr int NQ = SLOTS / 2, PANEL_STAGES = PANEL_BYTES / ROWB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 2, wc = wave & 3, half = lane >> 5, l31 = lane & 31;
    const int ps = tid & SMASK, lr = tid / SLOTS, ls = ps ^ ((lr >> SSH) & SMASK);
    const long rowbytes = (long)PANEL_STAGES * ROWB;
    A += blockIdx.x * a_stride;
    W += blockIdx.x * w_stride;
    const char* a_ptr[NLA];
    const char* w_ptr[NLB];
#pragma unroll
    for (int i = 0; i < NLA; ++i) a_ptr[i] = A + (long)(lr + RPP * i) * rowbytes + ls * 16;
#pragma unroll
    for (int i = 0; i < NLB; ++i) w_ptr[i] = W + (long)(lr + RPP * i) * rowbytes + ls * 16;
    const int swz = (l31 >> SSH) & SMASK;
    const int a_row0 = (wr * 32 * MI + l31) * ROWB, w_row0 = A_BYTES + (wc * 64 + l31) * ROWB;
    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int o = tid * 16; o < NST * STAGE_BYTES; o += NTHR * 16) *(uint4*)(smem + o) = *(const uint4*)(A + (o & 0xfffff));  // the panel's kind of data (constant / random)
    __syncthreads();
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 1024);
    auto dma = [&](const char* gsrc, unsigned dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
    };
    auto dma_nom0 = [&](const char* gsrc) { asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(gsrc) : "memory"); };
    if (DMA == 2) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(lds0) : "memory");
    auto issue_piece = [&](int pc, int kt, int stage) {
        const long kb = (long)(kt % PANEL_STAGES) * ROWB;
        const unsigned sa = lds0 + stage * STAGE_BYTES;
        const char* src = pc < NLA ? a_ptr[pc] + kb : w_ptr[pc - NLA] + kb;
        if (DMA == 2) dma_nom0(src);
        else dma(src, pc < NLA ? sa + pc * PASS_BYTES : sa + A_BYTES + (pc - NLA) * PASS_BYTES);
    };
    const uint4 ca = make_uint4(0x3f803f80u + tid, 0x3f003f00u, 0x3e803e80u, 0x3e003e00u);
    auto compute = [&](int stage, bool pf, int kt_pf, int stage_pf) {
        const char* st = smem + stage * STAGE_BYTES;
        constexpr int NQI = (DMA == 6 || NST > 2) ? NQ : DMA == 3 ? 1 : NQ / 2;
        constexpr int PPQ = (NL + NQI - 1) / NQI;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int so = ((half * NQ + q) ^ swz) << 4;
            uint4 fa[MI], fb[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = LDSREAD ? *(const uint4*)(st + w_row0 + j * 32 * ROWB + so) : ca;
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = LDSREAD ? *(const uint4*)(st + a_row0 + i * 32 * ROWB + so) : ca;
            if (DMA && pf) {
                if constexpr (PP == 0) {
#pragma unroll
                    for (int pc = q * PPQ; pc < (q + 1) * PPQ && pc < NL; ++pc) issue_piece(pc, kt_pf, stage_pf);
                } else {
                    constexpr int q0[2] = {0, PP == 2 ? 2 : 1};
                    constexpr int q1[2] = {PP == 1 ? 2 : 1, PP == 2 ? 3 : (PP == 1 ? 3 : 2)};
                    const int wr_s = __builtin_amdgcn_readfirstlane(wr);
#pragma unroll
                    for (int g = 0; g < 2; ++g)
                        if ((q == q0[g] && wr_s == 0) || (q == q1[g] && wr_s != 0)) {
#pragma unroll
                            for (int pc = g * PPQ; pc < (g + 1) * PPQ && pc < NL; ++pc) issue_piece(pc, kt_pf, stage_pf);
                        }
                }
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
        }
    };
    auto stage_barrier = [&]() {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    auto barrier_keep = [&]() {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"((NST - 2) * NL) : "memory");
        __builtin_amdgcn_s_barrier();
    };
    if constexpr (NST == 2) {
        if (DMA) {
#pragma unroll
            for (int pc = 0; pc < NL; ++pc) issue_piece(pc, 0, 0);
        }
        if (BARRIER) stage_barrier();
        for (int kt = 0; kt < nk; ++kt) {
            const bool pf = kt + 1 < nk && !(DMA == 5 && (kt & 1));
            compute(kt & 1, pf, kt + 1, (kt + 1) & 1);
            if (BARRIER) stage_barrier();
            else if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    } else {  // ring, as gemm16.hip's NST > 2 path
#pragma unroll
        for (int i = 0; i < NST - 1; ++i) {
#pragma unroll
            for (int pc = 0; pc < NL; ++pc) issue_piece(pc, i, i);
        }
        barrier_keep();
        int cur = 0, nxt = NST - 1;
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = kt + NST - 1 < nk;
            compute(cur, more, kt + NST - 1, nxt);
            if (more) barrier_keep(); else stage_barrier();
            cur = cur == NST - 1 ? 0 : cur + 1;
            nxt = nxt == NST - 1 ? 0 : nxt + 1;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 12345.678f) out[0] = s;
}

template <typename K>
double run(K kern, int rowb, int nst, int blocks, int nk, const char* A, const char* W, float* d, long a_stride, long w_stride, int BM = 256) {
    const int lds = nst * (BM + BN) * rowb;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(NTHR), lds, 0, A, W, d, nk / 4, a_stride, w_stride);
    hipDeviceSynchronize();
    double best = 0;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(NTHR), lds, 0, A, W, d, nk, a_stride, w_stride);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double tf = (double)blocks * (double)nk * 2.0 * BM * BN * (rowb / 2) / (ms * 1e-3) / 1e12;
        best = tf > best ? tf : best;
    }
    return best;
}

// Operand data: the chip clocks to its power budget and MFMA power depends on the operand bits toggling — constant panels (the
// round-3 form of this probe: hipMemset 0x3c) run the same instruction stream at a ~15-20 % higher clock than the random operands of
// a real product (MI355X_MICROARCH.md, DVFS give-back).  `gemm16_loop_probe random` fills the panels with hashed bf16 values in
// [-1, 1) — the rows to hold against gemm16_lab, which multiplies random operands.
__global__ void fill_random(unsigned short* p, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + 12345u;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        const float f = ((int)(x & 0xffffff) - 0x800000) * (1.0f / 0x800000);
        p[i] = (unsigned short)(__float_as_uint(f) >> 16);
    }
}

int main(int argc, char** argv) {
    const bool random = argc > 1 && !strcmp(argv[1], "random");
    const int blocks = 256;
    const long panel = (long)256 * PANEL_BYTES;  // 1 MiB per operand
    char *A, *W;
    float* d;
    if (hipMalloc(&A, panel * blocks) != hipSuccess || hipMalloc(&W, panel * blocks) != hipSuccess || hipMalloc(&d, 64) != hipSuccess) {
        printf("alloc failed\n");
        return 1;
    }
    hipMemset(A, 0x3c, panel * blocks);
    hipMemset(W, 0x3c, panel * blocks);
    if (random) {
        hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (unsigned short*)A, panel * blocks / 2);
        hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (unsigned short*)W, panel * blocks / 2);
        hipDeviceSynchronize();
    }
    printf("operand panels: %s\n\n", random ? "random bf16 in [-1, 1)" : "constant (0x3c3c)");
    printf("| staging of the next stage(s) | source of A / W | LDS fragment reads | drain + barrier per stage | TFLOP/s |\n|---|---|---|---|---:|\n");
#define ROW(desc, src, rd, bar, L, D, B, RB, NS, as, ws) \
    printf("| %s | %s | %s | %s | %.0f |\n", desc, src, rd, bar, run(probe<L, D, B, RB, NS>, RB, NS, blocks, 384000 / RB, A, W, d, as, ws)); fflush(stdout);
    ROW("none", "-", "no", "no", false, 0, false, 128, 2, 0, 0)
    ROW("none", "-", "yes", "yes", true, 0, true, 128, 2, 0, 0)
    ROW("LDS-DMA as gemm16.hip: 2 stages of 64 k (8 pieces behind the reads of q = 0, 1)", "shared / shared (L2-resident)", "yes", "yes", true, 1, true, 128, 2, 0, 0)
    ROW("the same", "private / shared (A streams from memory, W from L2)", "yes", "yes", true, 1, true, 128, 2, panel, 0)
    ROW("the same", "private / private (both stream from memory)", "yes", "yes", true, 1, true, 128, 2, panel, panel)
    ROW("the same, M0 set once", "shared / shared", "yes", "yes", true, 2, true, 128, 2, 0, 0)
    ROW("the same, all 8 pieces before the first MFMA", "shared / shared", "yes", "yes", true, 3, true, 128, 2, 0, 0)
    ROW("the same on every 2nd stage only (half the bytes per MFMA)", "shared / shared", "yes", "yes", true, 5, true, 128, 2, 0, 0)
    ROW("the same on every 2nd stage only", "private / private", "yes", "yes", true, 5, true, 128, 2, panel, panel)
    ROW("the same", "shared / shared", "no", "yes", false, 1, true, 128, 2, 0, 0)
    ROW("the same", "shared / shared", "yes", "no (vmcnt(0) only)", true, 1, false, 128, 2, 0, 0)
    ROW("ring of 3 x 32 k (two K steps in flight, counted vmcnt)", "shared / shared", "yes", "yes", true, 1, true, 64, 3, 0, 0)
    ROW("ring of 3 x 32 k", "private / shared", "yes", "yes", true, 1, true, 64, 3, panel, 0)
    ROW("ring of 3 x 32 k", "private / private", "yes", "yes", true, 1, true, 64, 3, panel, panel)
    ROW("ring of 4 x 32 k (three K steps in flight)", "shared / shared", "yes", "yes", true, 1, true, 64, 4, 0, 0)
    ROW("ring of 4 x 32 k", "private / shared", "yes", "yes", true, 1, true, 64, 4, panel, 0)
    ROW("ring of 4 x 32 k", "private / private", "yes", "yes", true, 1, true, 64, 4, panel, panel)
    // round 5: the same tile as the product's two (256 / 192 rows), and the DMA pieces placed per wave half (PP)
    printf("\n| tile | staging | source of A / W | TFLOP/s | us per K step (256 workgroups) |\n|---|---|---|---:|---:|\n");
#define ROW5(desc, src, D, MIv, PPv, as, ws)                                                                                   \
    {                                                                                                                          \
        const double tf = run(probe<true, D, true, 128, 2, MIv, PPv>, 128, 2, blocks, 3000, A, W, d, as, ws, 64 * MIv);      \
        printf("| %d x 256 | %s | %s | %.0f | %.3f |\n", 64 * MIv, desc, src, tf, 256.0 * 2.0 * 64 * MIv * 256 * 64 / tf * 1e-6); \
        fflush(stdout);                                                                                                       \
    }
    ROW5("none", "-", 0, 4, 0, 0, 0)
    ROW5("PP 0: every wave issues behind steps 0 / 1 (the product, round 4)", "shared / shared", 1, 4, 0, 0, 0)
    ROW5("PP 1: waves 0-3 behind steps 0 / 1, waves 4-7 behind 2 / 3", "shared / shared", 1, 4, 1, 0, 0)
    ROW5("PP 2: 0 / 2 and 1 / 3", "shared / shared", 1, 4, 2, 0, 0)
    ROW5("PP 3: 0 / 1 and 1 / 2", "shared / shared", 1, 4, 3, 0, 0)
    ROW5("PP 0", "private / shared", 1, 4, 0, panel, 0)
    ROW5("PP 1", "private / shared", 1, 4, 1, panel, 0)
    ROW5("PP 3", "private / shared", 1, 4, 3, panel, 0)
    ROW5("PP 0", "private / private", 1, 4, 0, panel, panel)
    ROW5("PP 1", "private / private", 1, 4, 1, panel, panel)
    ROW5("none", "-", 0, 3, 0, 0, 0)
    ROW5("PP 0", "shared / shared", 1, 3, 0, 0, 0)
    ROW5("PP 1", "shared / shared", 1, 3, 1, 0, 0)
    ROW5("PP 2", "shared / shared", 1, 3, 2, 0, 0)
    ROW5("PP 3", "shared / shared", 1, 3, 3, 0, 0)
    ROW5("PP 0", "private / shared", 1, 3, 0, panel, 0)
    ROW5("PP 1", "private / shared", 1, 3, 1, panel, 0)
    return 0;
}
