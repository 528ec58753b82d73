// gemm32big.hip — exact-fp32 GEMM on a 256x256 output tile (opt-in: tuning key "gemm32_big"; gemm.hip's 128x128 kernel
// stays the default).
//
// Why: profiles/r02_mfma_peak.md — the 128x128 kernel's whole gap to the matrix peak is the L2 -> LDS staging traffic per
// MFMA (136 TF with it, 145 TF with half the bytes per MFMA, 153 for the bare MFMA loop).  A 256x256 tile stages
// (256 + 256) rows for 4x the MFMAs of (128 + 128): half the bytes per MFMA.  Same instruction (v_mfma_f32_32x32x2_f32),
// same per-accumulator k order as gemm_kernel<float>: results are bit-identical to the default kernel.
//
// Shape of the kernel: 512 threads = 8 waves as 2 (M) x 4 (N); a wave owns 128 x 64 = 4 x 2 MFMA tiles (128 accumulator
// VGPRs), two waves share a SIMD, one workgroup per CU.  64-byte K stages, double buffered by LDS-DMA exactly as in
// gemm.hip (lane-linear LDS image, XOR swizzle through the source address, one wait + barrier per stage); the epilogue is
// gemm.hip's vector epilogue (bias, erf-GELU, fp32 residual, padded-frame zeroing) through a wave-private LDS transpose.
// Only worth launching where the tile count divides the 256 CUs well (launch_gemm checks): conv1-5 and fc1 of the
// HuBERT-base forward; q|k|v / out_proj / fc2 (189-567 tiles) would quantise at 0.74 and stay on the 128x128 kernel until
// a persistent / split-tile schedule exists.
#include "kernels.h"

namespace s3 {

namespace {

constexpr int BM = 256, BN = 256, ROWB = 64;
constexpr int STAGE_BYTES = (BM + BN) * ROWB;  // 32 KiB
constexpr int SLOTS = ROWB / 16, SMASK = SLOTS - 1, SSH = 2;
constexpr int RPT = 512 / SLOTS;  // rows covered by one pass of the 512 loader threads (128)
constexpr int NLD = BM / RPT;     // DMA instructions per thread per operand per stage (2)
constexpr int NQ = SLOTS / 2;     // fragment steps per stage (2)

__global__ __launch_bounds__(512, 1) void gemm32_big_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int half = lane >> 5;
    const int l31 = lane & 31;

    // XCD-aware tile order (as gemm.hip): every XCD gets a contiguous range of the (batch, m-tile, n-tile) sequence, n fastest
    const int n_tiles = (p.N + BN - 1) / BN;
    const int m_tiles = (p.M + BM - 1) / BM;
    int tile;
    {
        const int nwg = gridDim.x, wg = blockIdx.x;
        const int q8 = nwg >> 3, r8 = nwg & 7, xcd = wg & 7, loc = wg >> 3;
        tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    }
    const int tn = tile % n_tiles;
    const int tmb = tile / n_tiles;
    const int tm = tmb % m_tiles, b = tmb / m_tiles;
    const int m0 = tm * BM, n0 = tn * BN;

    const long lda_b = p.lda * 4;
    const long kbytes = (long)p.K * 4;
    const char* Ab = (const char*)p.A + (long)b * p.a_bs * 4;
    const char* Wb = (const char*)p.W;
    const int nk = (int)(kbytes / ROWB);  // launcher: K * 4 is a multiple of the stage

    // loader: thread owns one 16-byte slot of rows lr, lr + RPT of both operand tiles (see gemm.hip)
    const int ps = tid & SMASK;
    const int lr = tid / SLOTS;
    const int ls = ps ^ ((lr >> SSH) & SMASK);
    const char* a_ptr[NLD];
    const char* w_ptr[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        int ra = m0 + lr + RPT * i;
        ra = ra < p.M ? ra : p.M - 1;
        int rw = n0 + lr + RPT * i;
        rw = rw < p.N ? rw : p.N - 1;
        a_ptr[i] = Ab + (long)ra * lda_b + ls * 16;
        w_ptr[i] = Wb + (long)rw * kbytes + ls * 16;
    }

    const int swz = (l31 >> SSH) & SMASK;
    const int a_row0 = (wr * 128 + l31) * ROWB;
    const int w_row0 = BM * ROWB + (wc * 64 + l31) * ROWB;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int stage) {
        const char* st = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int so = ((half * NQ + q) ^ swz) << 4;
            uint4 fa[4], fb[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = *(const uint4*)(st + a_row0 + i * 32 * ROWB + so);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = *(const uint4*)(st + w_row0 + j * 32 * ROWB + so);
#pragma unroll
            for (int i = 0; i < 4; ++i)
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

    // LDS-DMA staging: a wave instruction lands 64 x 16 B = 1 KiB (16 rows) at a wave-uniform LDS base; pass i of the 8 waves
    // covers rows 128 i .. 128 i + 127 of an operand tile
    const unsigned lds0 =
        __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 1024);
    auto dma = [&](const char* gsrc, unsigned dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(gsrc), "s"(dst)
                     : "memory");
    };
    auto issue = [&](int kt, int stage) {
        const long kb = (long)kt * ROWB;
        const unsigned sa = lds0 + stage * STAGE_BYTES;
        const unsigned sw = sa + BM * ROWB;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            dma(a_ptr[i] + kb, sa + i * (RPT * ROWB));
            dma(w_ptr[i] + kb, sw + i * (RPT * ROWB));
        }
    };
    auto stage_barrier = [&]() {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    issue(0, 0);
    stage_barrier();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
        compute(kt & 1);
        stage_barrier();
    }

    // ---- epilogue: acc[i][j][r] is (row = wr*128 + i*32 + (r&3) + 8*(r>>2) + 4*half, col = wc*64 + j*32 + l31) ----
    const int limit = p.row_limit ? p.row_limit[b] : p.M;
    const long ob = (long)b * p.o_bs;
    float* stg = (float*)(smem + wave * 8192);  // 32 x 64 fp32 per wave: the 8 waves use exactly the two stage buffers
    const int c4 = (lane & 15) * 4;
    const int n = n0 + wc * 64 + c4;
    const bool n_ok = n < p.N;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && n_ok) bias4 = *(const float4*)(p.bias + n);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * half) * 64 + j * 32 + l31] = acc[i][j][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int row = t * 4 + (lane >> 4);
            float4 v = *(const float4*)(stg + row * 64 + c4);
            const int m = m0 + wr * 128 + i * 32 + row;
            if (m < p.M && n_ok) {
                v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
                if (p.act) {
                    v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w);
                }
                const long o = ob + (long)m * p.ldo + n;
                if (p.residual) {
                    const float4 rs = *(const float4*)(p.residual + o);
                    v.x += rs.x; v.y += rs.y; v.z += rs.z; v.w += rs.w;
                }
                if (m >= limit) v = make_float4(0.f, 0.f, 0.f, 0.f);
                *(float4*)(p.out32 + o) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

}  // namespace

int g_gemm32_big = 0;  // 0 off (default), 1 = where the tile count fits the CUs, 2 = every eligible shape (measurement)

// fp32 output only, vector-epilogue alignment, K a multiple of the 64-byte stage; mode 1 additionally wants the 256x256
// tiles to fill whole rounds of the 256 CUs to >= 0.93
bool gemm32_big_eligible(int dtype, const GemmParams& p) {
    if (dtype != F32 || !g_gemm32_big || !p.out32 || p.out16) return false;
    if ((p.K & 15) || (p.N & 3) || (p.ldo & 3) || (p.o_bs & 3) || ((p.lda * 4) & 15) || ((p.a_bs * 4) & 15)) return false;
    const uintptr_t al = (uintptr_t)p.A | (uintptr_t)p.W | (uintptr_t)p.out32 | (uintptr_t)p.residual | (uintptr_t)p.bias;
    if (al & 15) return false;
    if (p.M < 256 || p.N < 256) return false;
    if (g_gemm32_big >= 2) return true;
    const long tiles = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * p.batches;
    const long rounds = (tiles + 255) / 256;
    return tiles * 100 >= rounds * 256 * 93;
}

hipError_t launch_gemm32_big(const GemmParams& p, hipStream_t stream) {
    constexpr int lds = 2 * STAGE_BYTES;  // 64 KiB: one workgroup per CU (512 threads, 128 accumulator VGPRs per lane)
    hipError_t e = ensure_dynamic_lds<gemm32_big_kernel>(lds);
    if (e != hipSuccess) return e;
    dim3 grid(((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * p.batches);
    hipLaunchKernelGGL(gemm32_big_kernel, grid, dim3(512), lds, stream, p);
    return hipGetLastError();
}

}  // namespace s3
