// gemm16.hip — large-tile variant of the dense-contraction kernel for the 16-bit operand modes (bf16 / f16).
//
//   out[b][m][n] = epilogue( sum_k A[b][m][k] * W[n][k] )        (same contract as gemm.hip)
//
// Why a second kernel: at the 16-bit MFMA rate (16x the fp32 rate) the 128x128 / 4-wave tile of gemm.hip is
// bounded by what feeds the matrix pipe, not by the pipe: 64 FLOP per byte staged from L2 and 1 KiB of LDS reads per
// MFMA.  Here a 512-thread workgroup owns a (2*WTM) x 256 tile, 8 wave64 as 2(M) x 4(N), each wave a WTM x 64
// sub-tile (WTM = 128: 8 accumulator tiles of 32x32 = 128 VGPRs; 128 FLOP per staged byte, 0.75 KiB LDS per MFMA).
//   * staging: LDS-DMA only (global_load_lds_dwordx4, 1 KiB per wave instruction, whole lines of a row per 8 / 4
//     lanes), issued from INLINE ASM with hand-counted vmcnt: through the builtin hipcc drains the DMA (vmcnt(0))
//     before the first fragment read of every K-step, which serialises load and compute.  Configurations:
//       mode 1  256x256 (or 192x256, whichever leaves fewer idle CU-rounds) tile, 2 stages of 64 k, one workgroup per CU
//               (128 FLOP per staged byte) — the default for every shape since the 16-bit epilogues store 16 bytes per
//               lane (round 2: q|k|v 64 vs 72 us, fc1 99 vs 104 us against mode 4);
//       mode 7  mode 1's tiles walked by ONE persistent workgroup per CU (the default since late round 3): the first K step of
//               the next tile is issued before the current tile's epilogue, bit-identical to mode 1 (tests), q|k|v 70.9 -> 66.4 us,
//               1 % on the other shapes (profiles/r03_gemm16_loop_probe.md);
//       mode 4  128x256 tile, ring of 3 stages of 32 k with two K-steps in flight, two workgroups per CU that hide
//               each other's barriers and epilogues — kept as a tuning option (`gemm16_big` = 4).
//     (Round 2's phase-pipelined variant, gemm16p.hip — staggered wave rows, region-granular DMA ring — measured equal or
//     slower on every shape and was removed in round 3.)
//     The DMA pieces are interleaved with the MFMA steps (an LDS-DMA instruction costs 60-180 issue cycles).  LDS rows
//     are XOR-swizzled through the per-lane SOURCE address (the DMA image is lane-linear) so every ds_read_b128
//     fragment read is bank-conflict free — same layout as gemm.hip.
//   * epilogue: accumulators go through a wave-private LDS transpose (32 x 64 fp32 per step) so that bias, GELU,
//     residual and the padded-frame zeroing run on row-contiguous float4s and every global access is a full
//     16-byte (fp32) / 8-byte (16-bit) vector; specialised at compile time on the four flag combinations of the path;
//     v_cvt_pk_bf16_f32 / v_cvt_f16_f32 for the 16-bit stores.
//   * GELU in the 16-bit modes is the one-transcendental form of common.h (gelu_fast: one v_exp_f32 + a degree-7
//     Horner, packed on pairs) instead of libm's erff (38 VALU with a divergent branch): at K = 768 the erff
//     epilogue costs about as many VALU cycles as the whole K loop costs MFMA cycles.  Its error is at the fp32
//     rounding level, and the fp32 mode uses it too (libm erff under the tuning key gelu32 = 0).
// Measured: profiles/r02_gemm16_variants.md, r02_pmc_gemm16.md (700-1000 TF on the shapes of the path, 1.15 PF at 8192^3:
// MFMA-busy 0.61 at a power-limited ~1.55 GHz; 18 % more on zero-filled operands).
// Requirements (checked by the launcher, which otherwise falls back to gemm.hip): K a multiple of 64, N / ldo /
// o_bs multiples of 4, 16-byte aligned operands and outputs.
#include <type_traits>

#include "kernels.h"

// Timing probes (garbage results by design; compiled in only by tools/micro/gemm16_lab.hip): GemmParams.variant bit 4 = the
// epilogue computes but does not store, bit 5 = no epilogue at all, bit 6 = no K loop (epilogue of zeros only)
#if defined(S3_GEMM_PROBE) && defined(__HIP_DEVICE_COMPILE__)
#define S3_GPROBE(p_, bit_) ((p_).variant & (bit_))
#define S3_GKEEP4(v_) asm volatile("" ::"v"((v_).x), "v"((v_).y), "v"((v_).z), "v"((v_).w))
#else
#define S3_GPROBE(p_, bit_) 0
#define S3_GKEEP4(v_) ((void)0)
#endif

namespace s3 {

namespace {


template <typename T> struct Mma16;
template <> struct Mma16<bf16_tag> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma16<f16_tag> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

// WTM: rows per wave (tile = 2*WTM x 256); ROWB: bytes of K per row per LDS stage (128 or 64); NST: LDS stages (2: the
// next stage lands while this one is multiplied; 3: two stages in flight, counted vmcnt); WPE: waves per SIMD the
// register budget is capped for (2 = one workgroup per CU, 4 = two).
// WN: waves along N (4: 8-wave workgroup, 256 columns; 2: 4-wave workgroup, 128 columns — two such workgroups per CU
// put ONE wave of each on every SIMD, so their barriers and epilogues interleave).
// PERSIST: one workgroup per CU walks ITS tiles (the XCD's contiguous range, strided by the XCD's workgroups) and issues the
// first K step of the next tile by LDS-DMA before it starts the epilogue of the current one — the per-tile prologue (arguments,
// addresses, a first stage with nothing to overlap it: ~2.5 us of a 24 us q|k|v tile, profiles/r03_gemm16_loop_probe.md) runs
// under the epilogue, whose LDS staging moves out of stage 0's way (behind it).
// OVL (PERSIST only; round 4): BOTH stages of the next tile are issued before the epilogue of the current one (its staging
// shrinks to 4 KiB per wave and sits behind the two stages) and the next tile starts on a COUNTED vmcnt that skips the epilogue's
// stores (vmcnt retires in order: the DMA pieces are older than the stores, so "all but the youngest S" = "the DMA has landed"),
// which lets the store drain run under the next tile's first two K steps instead of in front of them.  A schedule change only:
// same MFMA order per accumulator, bit-identical results.  Measured in profiles/r04_gemm16_overlap.md: -1.5...-4 % on the
// multi-round K = 768 / 1024 shapes, +0.5...2 % on single-round ones, nothing on the forward — opt-in (`gemm16_big` = 8).
// (A second option of that round — every thread touching one line of K step kt+2 a step ahead of its DMA — cost 12 % and was removed.)
// SWAP (round 4; only launched for the 16-bit-output epilogues: conv1-5, q|k|v, fc1): the two MFMA operands change places, so an
// accumulator block holds the TRANSPOSED 32 x 32 tile — a lane owns ONE ROW of the output (32 lanes = 32 rows) and 16 of its 32
// columns in groups of four consecutive ones.  Bias / GELU / conversion run on those registers and two v_permlane32_swap per
// 8 bytes hand each lane a contiguous 16-byte piece of its row: the epilogue needs NO LDS transpose (256 KiB written + read per
// tile before: the largest part of the 7 us a q|k|v tile spent in its epilogue).  Each product a * w and the order in which an
// accumulator sums them are unchanged — results bit-identical (tests).
// dynamic LDS of one workgroup (kernel and launcher agree through this one function)
constexpr int big_lds_bytes(int BM, int BN, int ROWB, int NST, int WN, bool PERSIST, bool OVL, bool SWAP, bool MXW) {
    const int stage = (BM + BN) * ROWB + (MXW ? BN * 32 : 0);
    const int staging = SWAP ? 0 : 2 * WN * (OVL ? 4096 : 8192);  // (SWAP: no LDS in the epilogue)
    // PERSIST: the epilogue's staging sits behind stage 0, which the next tile's first K step is landing in meanwhile;
    // OVL: behind BOTH stages (256 x 256: 128 + 32 KiB = all of a CU's LDS)
    const int body = OVL ? NST * stage + staging : (PERSIST ? (NST * stage > stage + staging ? NST * stage : stage + staging) : NST * stage);
    return body + (MXW ? 2 * BN * 4 : 0);
}

// PP (round 5; PERSIST, 2 stages of 4 fragment steps): WHICH fragment steps of a K step carry a wave's DMA pieces depends on the
// wave's half of the workgroup.  Waves w and w + 4 share a SIMD (and its matrix pipe); an LDS-DMA instruction costs its wave
// 60-185 issue cycles (MI355X_MICROARCH.md), during which that wave issues no MFMA.  With every wave issuing its 8 pieces behind
// the reads of steps 0 and 1 (PP = 0) both waves of a SIMD stall at the same time and the pipe idles for ~8 x 100 cycles of a
// 2048-cycle K step — the whole difference between this loop (1.4 us per step) and a pure MFMA stream (1.0 us: the loop probe,
// profiles/r03_gemm16_loop_probe.md, priced the pieces at ~55 cycles each with nothing else in the phase).  PP = 1: waves 0-3
// issue behind steps 0 / 1, waves 4-7 behind steps 2 / 3 — one wave of a SIMD multiplies while its partner issues; PP = 2:
// steps 0 / 2 and 1 / 3; PP = 3: steps 0 / 1 and 1 / 2 (the late pieces get one more step to land before the stage barrier).
// A schedule change only: same products, same order per accumulator — bit-identical (tests).
// MXW (round 5; fp16, the persistent 192 x 256 tile): S3ENC_F16X2's SECOND weight term on the MX pipe.  w = hi + lo with hi = fp16(w);
// instead of a second fp16 term multiplied in a second pass over K (the `wsplit` loop: twice the MFMAs, A staged twice), lo is an
// MX-fp4 image — per row and 32 k one E8M0 scale and 32 e2m1 nibbles (GemmParams.W4 / W4s, packed at s3enc_create) — and ONE
// v_mfma_scale_f32_32x32x64_f8f6f4 per 64 k and accumulator block multiplies it with an MX-fp4 image of A in the same K step as
// the four fp16 MFMAs (+25 % matrix work instead of +100 %).  The A image is built HERE, on registers: the lane's four fp16
// fragments of a K step (slot 4 half + q of the row's 128 bytes: k = 32 half + 8 q + j) ARE one 32-k block in natural order, so the
// block max, the scale and sixteen v_cvt_scalef32_pk_fp4_f16 need no producer and no A-side staging.  W_lo's tile (256 rows x
// 32 bytes per K step) rides along as one more DMA piece, its scales as a 4-byte piece every second K step.  The products a * lo
// carry ~2^-4 relative error on a term that is 2^-11 of the weight: 4.8e-5 of weight error against 2.2e-4 for one fp16 term and
// 4.8e-7 for two (profiles/r04_mx_gemm_lab.md, float64 reference).  K % 128 == 0.
template <typename T, int WTM, int ROWB, int NST, int WPE, int WN, bool PERSIST = false, bool OVL_ = false, bool SWAP = false, int PP = 0,
          bool MXW = false>
__global__ __launch_bounds__(128 * WN, WPE) void gemm16_big_kernel(GemmParams p) {
    constexpr bool OVL = PERSIST && OVL_;
    static_assert(!MXW || (PERSIST && !OVL_ && NST == 2 && ROWB == 128 && WN == 4 && std::is_same<T, f16_tag>::value),
                  "MXW: fp16, the persistent 2-stage loop");
    static_assert(PP == 0 || (PERSIST && NST == 2 && ROWB == 128 && WN == 4), "PP: the persistent 2-stage loop with 4 fragment steps");
    constexpr int NTHR = 128 * WN;  // 2 waves along M x WN along N
    constexpr int BM = 2 * WTM, BN = 64 * WN;
    constexpr int MI = WTM / 32;  // 32-row accumulator blocks per wave
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, W4_BYTES = MXW ? BN * 32 : 0, STAGE_BYTES = A_BYTES + B_BYTES + W4_BYTES;
    constexpr int SCB = BN * 4;  // MXW: scale bytes per buffer — 4 blocks (two K steps) per W row; two buffers behind everything else
    constexpr int SC_OFF = big_lds_bytes(BM, BN, ROWB, NST, WN, PERSIST, OVL, SWAP, MXW) - (MXW ? 2 * SCB : 0);
    constexpr int SLOTS = ROWB / 16, SMASK = SLOTS - 1;  // 16-byte slots per row per stage
    constexpr int SSH = ROWB == 128 ? 1 : 2;              // swizzle: slot ^= (row >> SSH) & SMASK
    constexpr int RPP = NTHR / SLOTS;                     // rows filled by one pass of the workgroup's waves
    constexpr int PASS_BYTES = NTHR * 16;                 // LDS bytes of one pass
    constexpr int NLA = BM / RPP, NLB = BN / RPP;         // LDS-DMA instructions per wave per stage and operand
    constexpr int NQ = SLOTS / 2;                         // 16-deep fragment steps per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave / WN, wc = wave % WN;
    const int wr_s = __builtin_amdgcn_readfirstlane(wr);  // (scalar: PP branches on it per fragment step)
    const int half = lane >> 5;
    const int l31 = lane & 31;
    // Round 6, fourth session (profiles/r06d_concurrent_forwards_exclusions.md): a wave of ANOTHER kernel that shares a SIMD with waves issuing
    // v_mfma_f32_32x32x16_{bf16,f16} was measured to compute single VALU results wrong (conv0_kernel beside these kernels' 182 / 198 / 218-register
    // instantiations: 4 of 4 trials; beside the 256-register ones, which leave no room, or beside the same GEMM on two 32x32x8 MFMAs: never).
    // The two-waves-per-SIMD instantiations therefore claim the whole register file (v255 reserved -> 256 allocated; their occupancy is two
    // waves either way): no foreign wave — this library's or a caller's — can be placed beside them.
    if constexpr (WPE == 2) asm volatile("" ::: "v255");

    // XCD-aware tile order (see gemm.hip): every XCD gets a contiguous range of (batch, m-tile, n-tile), n fastest
    const int n_tiles = (p.N + BN - 1) / BN;
    const int m_tiles = (p.M + BM - 1) / BM;
    int tile, tile_end, tile_step;  // this workgroup's tiles: tile, tile + tile_step, ... < tile_end
    {
        const int nwg = gridDim.x, wg = blockIdx.x, xcd = wg & 7, loc = wg >> 3;
        if constexpr (PERSIST) {
            const int total = n_tiles * m_tiles * p.batches;
            const int q8 = total >> 3, r8 = total & 7;
            const int start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
            tile_end = start + q8 + (xcd < r8 ? 1 : 0);
            tile_step = (nwg >> 3) + (xcd < (nwg & 7) ? 1 : 0);
            tile = start + loc;
            if (tile >= tile_end) return;
        } else {
            const int q8 = nwg >> 3, r8 = nwg & 7;
            tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
            tile_end = tile + 1;
            tile_step = 1;
        }
    }
#if defined(S3_GEMM_PROBE) && defined(__HIP_DEVICE_COMPILE__)
    // lab only (tools/micro/gemm16_lab `cmp ... <skew>0007`): every second workgroup of an XCD starts (variant >> 8) x 2 us late — what
    // would it buy if the workgroups' epilogue bursts did NOT land in the same microsecond (the premise of a skewed / stream-K walk)?
    if constexpr (PERSIST) {
        const int sk = (p.variant >> 8) & 0xff;
        if (sk && ((blockIdx.x >> 3) & 1)) {
            const long long t0 = wall_clock64();
            while (wall_clock64() - t0 < sk * 200LL) __builtin_amdgcn_s_sleep(8);
        }
    }
#endif
    int m0, n0, b;  // the tile being multiplied / written
    auto coords = [&](int t, int& cm0, int& cn0, int& cb) {
        const int tn = t % n_tiles;
        const int tmb = t / n_tiles;
        cm0 = (tmb % m_tiles) * BM;
        cn0 = tn * BN;
        cb = tmb / m_tiles;
    };
    coords(tile, m0, n0, b);

    const long lda_b = p.lda * 2;
    const long kbytes = (long)p.K * 2;                      // bytes of A's K extent
    const long wk = p.wsplit ? 2 * kbytes : kbytes;        // bytes of the contraction: [hi | lo] weights run A twice
    const long ldw_b = p.ldw ? p.ldw * 2 : wk;              // W row stride
    const char* Wb = (const char*)p.W;
    const int nk = (int)(wk / ROWB);

    // ---- loader: lane (lr, ps) fills physical 16-byte slot ps of row lr (+64 per pass) and FETCHES the logical slot
    //      ps ^ swizzle(row): 8 lanes read one whole 128-byte line of a row ----
    const int ps = tid & SMASK;
    const int lr = tid / SLOTS;  // 0..RPP-1
    const int ls = ps ^ ((lr >> SSH) & SMASK);
    const char* a_ptr[NLA];
    const char* w_ptr[NLB];
    const char *w4_ptr = nullptr, *sc_ptr = nullptr;  // MXW: lane t fills block t & 1 of W_lo row t >> 1; lane t < BN the 4 scale bytes of row t
    const int kblocks = p.K >> 5;
    auto set_ptrs = [&](int pm0, int pn0, int pb) {
        if constexpr (MXW) {
            int r4 = pn0 + (tid >> 1);
            r4 = r4 < p.N ? r4 : p.N - 1;
            w4_ptr = (const char*)p.W4 + ((long)r4 * kblocks + (tid & 1)) * 16;
            int rs = pn0 + tid;
            rs = rs < p.N ? rs : p.N - 1;
            sc_ptr = (const char*)p.W4s + (long)rs * kblocks;
        }
        const char* Ab = (const char*)p.A + (long)pb * p.a_bs * 2;
#pragma unroll
        for (int i = 0; i < NLA; ++i) {
            int ra = ((p.variant & 8) ? 0 : pm0) + lr + RPP * i;  // variant bit 3: every tile loads tile 0 (timing probe only)
            ra = ra < p.M ? ra : p.M - 1;
            a_ptr[i] = Ab + (long)ra * lda_b + ls * 16;
        }
#pragma unroll
        for (int i = 0; i < NLB; ++i) {
            int rw = ((p.variant & 8) ? 0 : pn0) + lr + RPP * i;
            rw = rw < p.N ? rw : p.N - 1;
            w_ptr[i] = Wb + (long)rw * ldw_b + ls * 16;
        }
    };
    set_ptrs(m0, n0, b);
    // LDS-DMA issued from inline asm: hipcc does not count it, so it inserts no vmcnt(0) in front of the fragment
    // ds_reads of the stage being multiplied (with the builtin it does — the DMA is a pending LDS write it cannot
    // disambiguate — which serialises load and compute); completion is waited for by hand before the stage barrier.
    // M0 (the wave-uniform LDS destination) is written in the same statement that uses it and restored.
    const unsigned lds_base =
        __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 1024);
    const unsigned lds_sc = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + SC_OFF + wave * 256);
    auto glds4 = [&](const char* gsrc, unsigned lds_dst) {  // 4 bytes per lane (the MX scales)
        unsigned keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" S3_GLDS_MOD "\n\ts_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(gsrc), "s"(lds_dst)
            : "memory");
    };
    auto glds16 = [&](const char* gsrc, unsigned lds_dst) {
        unsigned keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" S3_GLDS_MOD "\n\ts_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(gsrc), "s"(lds_dst)
            : "memory");
    };
    // piece pc of K-step kt -> LDS stage `stage`: pieces [0, NLA) are A passes, [NLA, NLA + NLB) W passes
    constexpr int NL = NLA + NLB + (MXW ? 1 : 0);  // (MXW: + the W_lo tile, one pass)
    auto issue_piece = [&](int pc, int kt, int stage) {
        const long kb = (long)kt * ROWB;
        const long kba = kb >= kbytes ? kb - kbytes : kb;    // (wsplit: the lo half re-reads A from its start; K % 64 == 0)
        const unsigned sa = lds_base + stage * STAGE_BYTES;  // wave-uniform; lane l lands at + l*16
        if (pc < NLA) glds16(a_ptr[pc] + kba, sa + pc * PASS_BYTES);
        else if (pc < NLA + NLB) glds16(w_ptr[pc - NLA] + kb, sa + A_BYTES + (pc - NLA) * PASS_BYTES);
        else {
            glds16(w4_ptr + (long)kt * 32, sa + A_BYTES + B_BYTES);
            // the scales of K steps kt, kt + 1 (blocks 2 kt .. 2 kt + 3) with the even step's last piece: buffer (kt / 2) & 1 was last
            // read two pairs ago
            if (!(kt & 1) && wave < BN / 64) glds4(sc_ptr + 2 * kt, lds_sc + ((kt >> 1) & 1) * SCB);
        }
    };
    auto issue = [&](int kt, int stage) {
#pragma unroll
        for (int pc = 0; pc < NL; ++pc) issue_piece(pc, kt, stage);
    };
    constexpr int STG_WAVE = (OVL && !SWAP) ? 4096 : 8192;  // epilogue staging per wave: 32 x 32 / 32 x 64 fp32 (SWAP: none)
    constexpr int STG_OFF = (OVL && !SWAP) ? 2 * STAGE_BYTES : (PERSIST ? STAGE_BYTES : 0);
    // my DMA (all of it, or all but the newest stage's NLA + NLB instructions) has landed and my fragment reads are
    // done; then everybody's
    auto barrier_all = [&]() {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    auto barrier_lgkm = [&]() {  // my fragment reads are done (nothing of mine in the VM queue is needed by anybody yet)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    // all but my youngest `n` VM operations have completed (n = the store instructions of the epilogue just behind the DMA)
    auto wait_vm_keep = [&](int n) {
        switch (n) {
            case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
            case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
            case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
            case 32: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
    };
    auto barrier_keep = [&]() {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"((NST - 2) * (NLA + NLB)) : "memory");
        __builtin_amdgcn_s_barrier();
    };

    // ---- fragment addresses ----
    const int swz = (l31 >> SSH) & SMASK;
    const int a_row0 = (wr * WTM + l31) * ROWB;
    const int w_row0 = A_BYTES + (wc * 64 + l31) * ROWB;

    f32x16 acc[MI][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    zero_acc();

    // multiply stage `stage`; when `pf`, also issue the DMA pieces of K-step kt_pf into stage_pf, spread over the
    // NQ fragment steps (right after each step's ds_reads): an LDS-DMA instruction costs 60-180 issue cycles
    // (MI355X_MICROARCH.md), so 8 of them issued up front would idle the matrix pipe for ~1/3 of a K-step
    // 2-stage pipeline: the pieces must land before THIS step's barrier, so they are issued in its first half;
    // rings of 3+ wait for an older K-step and spread them over the whole step
    constexpr int NQI = (NST == 2 && NQ >= 4) ? NQ / 2 : NQ;
    constexpr int PPQ = (NL + NQI - 1) / NQI;
    auto compute = [&](int stage, bool pf, int kt_pf, int stage_pf) {
        const char* st = smem + stage * STAGE_BYTES;
        uint4 keep[MXW ? MI : 1][4];  // MXW: the lane's 32 fp16 values of each row block over the four fragment steps
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int so = ((half * NQ + q) ^ swz) << 4;
            uint4 fa[MI], fb[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = *(const uint4*)(st + w_row0 + j * 32 * ROWB + so);
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = *(const uint4*)(st + a_row0 + i * 32 * ROWB + so);
            if (pf) {
                if constexpr (PP == 0) {
#pragma unroll
                    for (int pc = q * PPQ; pc < (q + 1) * PPQ && pc < NL; ++pc) issue_piece(pc, kt_pf, stage_pf);
                } else {
                    // piece group g (0 / 1) goes out behind the reads of step q0[g] for the first half of the waves, q1[g] for the second
                    constexpr int q0[2] = {0, PP == 2 ? 2 : 1};
                    constexpr int q1[2] = {PP == 1 ? 2 : 1, PP == 2 ? 3 : (PP == 1 ? 3 : 2)};
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const bool mine = (q == q0[g] && q == q1[g]) ? true : (q == q0[g] ? wr_s == 0 : (q == q1[g] ? wr_s != 0 : false));
                        if (q == q0[g] || q == q1[g]) {
                            if (mine) {
#pragma unroll
                                for (int pc = g * PPQ; pc < (g + 1) * PPQ && pc < NL; ++pc) issue_piece(pc, kt_pf, stage_pf);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                if constexpr (MXW) keep[i][q] = fa[i];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (SWAP) Mma16<T>::run(fb[j], fa[i], acc[i][j]);  // lane <-> row m, registers <-> columns n
                    else Mma16<T>::run(fa[i], fb[j], acc[i][j]);                 // lane <-> column n, registers <-> rows m
                }
            }
        }
        if constexpr (MXW) {
            // ---- the second weight term of this K step: acc += mx4(A block) x mx4(W_lo block), 64 k per instruction ----
            typedef int v8i __attribute__((ext_vector_type(8)));
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            const int kt = kt_pf - 1;
            const char* sc = smem + SC_OFF + ((kt >> 1) & 1) * SCB + (kt & 1) * 2 + half;
            uint4 a4[MI], b4[2];
            int sa[MI], sb[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = wc * 64 + j * 32 + l31;
                b4[j] = *(const uint4*)(st + A_BYTES + B_BYTES + row * 32 + half * 16);
                sb[j] = (int)(*(const unsigned char*)(sc + row * 4)) * 0x01010101;
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                // block max of the 32 halves (|x| as bit patterns: positive fp16 values order like unsigned integers), the scale
                // 2^e with max / 2^e <= 6 (e2m1's largest value), sixteen pair conversions
                unsigned mx = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned d[4] = {keep[i][q].x, keep[i][q].y, keep[i][q].z, keep[i][q].w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const unsigned a = d[c] & 0x7fff7fffu;
                        mx = max(mx, max(a & 0xffffu, a >> 16));
                    }
                }
                const float amax = (float)__builtin_bit_cast(_Float16, (unsigned short)mx);
                int ex = amax > 0.f ? __builtin_amdgcn_frexp_expf(amax * (1.f / 6.f)) : -127;
                ex = ex < -127 ? -127 : ex;
                const float scf = __builtin_amdgcn_ldexpf(1.f, ex);
                unsigned o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {  // elements 8 q .. 8 q + 7 of the block -> dword q
                    unsigned w_ = 0;
                    w_ = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w_, __builtin_bit_cast(h2, keep[i][q].x), scf, 0);
                    w_ = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w_, __builtin_bit_cast(h2, keep[i][q].y), scf, 1);
                    w_ = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w_, __builtin_bit_cast(h2, keep[i][q].z), scf, 2);
                    w_ = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w_, __builtin_bit_cast(h2, keep[i][q].w), scf, 3);
                    o[q] = w_;
                }
                a4[i] = make_uint4(o[0], o[1], o[2], o[3]);
                sa[i] = (ex + 127) * 0x01010101;
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const v8i av = {(int)a4[i].x, (int)a4[i].y, (int)a4[i].z, (int)a4[i].w, 0, 0, 0, 0};
                    const v8i bv = {(int)b4[j].x, (int)b4[j].y, (int)b4[j].z, (int)b4[j].w, 0, 0, 0, 0};
                    if constexpr (SWAP) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bv, av, acc[i][j], 4, 4, 0, sb[j], 0, sa[i]);
                    else acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc[i][j], 4, 4, 0, sa[i], 0, sb[j]);
                }
        }
    };

    if constexpr (PERSIST) {
        static_assert(!PERSIST || NST == 2, "the persistent tile loop is written for the 2-stage pipeline");
    } else if (S3_GPROBE(p, 64)) {
    } else if constexpr (NST == 2) {
        issue(0, 0);
        barrier_all();  // stage 0 is visible to every wave
        for (int kt = 0; kt < nk; ++kt) {
            compute(kt & 1, kt + 1 < nk, kt + 1, (kt + 1) & 1);  // K-step kt+1 lands while stage kt is multiplied
            barrier_all();
        }
    } else {
        // ring of NST stages: K-steps kt+1 .. kt+NST-2 are in flight while kt is multiplied; the DMA of kt+NST-1 is
        // issued (interleaved with the MFMAs) into the stage read during kt-1 — every wave is past that step's barrier.
        // The stage barrier lets the newest NST-2 K-steps' pieces stay in flight (counted vmcnt).
#pragma unroll
        for (int i = 0; i < NST - 1; ++i)
            if (i < nk) issue(i, i);
        if (nk >= NST - 1) barrier_keep(); else barrier_all();
        int cur = 0, nxt = NST - 1;
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = kt + NST - 1 < nk;
            compute(cur, more, kt + NST - 1, nxt);
            if (more) barrier_keep(); else barrier_all();  // tail: simply drain
            cur = cur == NST - 1 ? 0 : cur + 1;
            nxt = nxt == NST - 1 ? 0 : nxt + 1;
        }
    }

#if defined(S3_GEMM_PROBE) && defined(__HIP_DEVICE_COMPILE__)
    if (!PERSIST && S3_GPROBE(p, 32)) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
#endif
    // Returns the number of global STORE instructions this wave issued when that number is fixed and nothing else of the
    // epilogue entered the VM queue behind them (OVL: what the next tile's counted wait may leave in flight), else 0.
    // SWAP: the row-per-lane epilogue (16-bit output, optional bias / GELU; the launcher guarantees no residual, no fp32 output, no
    // row limit, N % 8 == 0 and 16-byte aligned output rows).  Returns the wave's store count when it is fixed (full tile).
    auto epilogue_rows = [&](auto full_c, auto act_c) -> int {
        constexpr bool FULL = decltype(full_c)::value;
        constexpr bool act = decltype(act_c)::value;
        typedef typename Cvt<T>::store_t store_t;
        const long ob = (long)b * p.o_bs;
        // my 16 columns of a 32-column block: 8 q + 4 half + {0, 1, 2, 3}, q = 0..3
        float4 bz[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wc * 64 + j * 32 + 8 * q + 4 * half;
                bz[j][q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.bias && (FULL || n < p.N)) bz[j][q] = *(const float4*)(p.bias + n);
            }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = m0 + wr * WTM + i * 32 + l31;
            const bool row_ok = FULL || m < p.M;
            store_t* orow = (store_t*)p.out16 + ob + (long)m * p.ldo;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                unsigned pk[4][2];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 v = make_float4(acc[i][j][4 * q] + bz[j][q].x, acc[i][j][4 * q + 1] + bz[j][q].y,
                                           acc[i][j][4 * q + 2] + bz[j][q].z, acc[i][j][4 * q + 3] + bz[j][q].w);
                    if (act) gelu_fast4(v);
                    pk[q][0] = Cvt<T>::pack2(v.x, v.y);
                    pk[q][1] = Cvt<T>::pack2(v.z, v.w);
                }
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    // lanes l (half 0) and l + 32 (half 1) own the same row: column groups q = 2 pr (half 0's four columns, then
                    // half 1's) and q = 2 pr + 1.  After swap(x = group 2 pr, y = group 2 pr + 1): the lower lane holds group 2 pr
                    // of BOTH halves (x: its own, y: its partner's) = columns 16 pr .. + 7, the upper lane group 2 pr + 1 of both
                    // = columns 16 pr + 8 .. + 15
                    const auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * pr][0], pk[2 * pr + 1][0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(pk[2 * pr][1], pk[2 * pr + 1][1], false, false);
                    const uint4 o = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                    const int n8 = n0 + wc * 64 + j * 32 + 16 * pr + 8 * half;
                    if (row_ok && (FULL || n8 < p.N)) {
                        if (S3_GPROBE(p, 16)) S3_GKEEP4(o);
                        else *(uint4*)(orow + n8) = o;
                    }
                }
            }
        }
        return FULL ? MI * 4 : 0;
    };
    // `pre`: what the caller wants issued as early as the epilogue allows (the persistent loop: the next tile's first DMA).  A
    // residual epilogue calls it BEHIND its first residual loads — vmcnt retires in order, so loads queued behind eight DMA
    // pieces would return a stage-fetch later — every other epilogue at once.
    auto run_epilogue = [&](bool want_count, auto&& pre) -> int {  // of tile (m0, n0, b)
    bool pre_done = false;
    auto do_pre = [&]() {
        if (!pre_done) pre();
        pre_done = true;
    };
    if constexpr (SWAP) {
        do_pre();
        const bool full_tile = m0 + BM <= p.M && n0 + BN <= p.N;
        const bool act = p.act != 0;  // workgroup-uniform: one branch per tile, not one per four values
        if (OVL && want_count && full_tile && !S3_GPROBE(p, 16))
            return act ? epilogue_rows(std::true_type{}, std::true_type{}) : epilogue_rows(std::true_type{}, std::false_type{});
        if (act) (void)epilogue_rows(std::false_type{}, std::true_type{});
        else (void)epilogue_rows(std::false_type{}, std::false_type{});
        return 0;
    } else {
    // ---- epilogue through a wave-private LDS transpose: 32 x SW fp32 per round (SW = 64: both 32-column accumulator blocks of
    // a 32-row block; OVL: 32, one accumulator block — 4 KiB per wave, so that the staging fits BESIDE both K stages) ----
    // Specialised at compile time on (GELU, residual, fp32 out, 16-bit out) for the four combinations the encoder uses
    // — the generic form tests five uniform flags per 4-row pass (168 branches per tile) — with a generic fallback.
    typedef typename Cvt<T>::store_t store_t;
    constexpr int NJ = OVL ? 1 : 2;   // accumulator blocks per staging round
    constexpr int SW = 32 * NJ;       // staging row width (floats)
    constexpr int NJB = 2 / NJ;       // staging rounds per 32-row block
    float* stg = (float*)(smem + STG_OFF + wave * STG_WAVE);
    const int limit = p.row_limit ? p.row_limit[b] : p.M;
    const long ob = (long)b * p.o_bs;
    // a tile with every row and column inside the product stores unconditionally: a fixed number of store instructions
    const bool full_tile = m0 + BM <= p.M && n0 + BN <= p.N;
    int n_stores = 0;
    auto epilogue = [&](auto spec, auto act_c, auto res_c, auto o32_c, auto o16_c, auto full_c, auto ln_c) {
        constexpr bool SPEC = decltype(spec)::value;
        constexpr bool FULL = decltype(full_c)::value;
        // residual epilogue (out_proj, fc2): the residual rows of a staging round are loaded one round AHEAD, before that round's
        // LDS transpose.  In the forward the residual (the previous LayerNorm's fp32 output) is cold — Infinity Cache at best —
        // and a load issued right before its use stalled every round for a memory round trip: out_proj / fc2 ran 46 / 105 us
        // in the forward against 31 / 76 us on cache-warm operands in the lab (profiles/r04_gemm16_residual.md)
        constexpr bool RESPF = SPEC && decltype(res_c)::value;
        if constexpr (!RESPF) do_pre();
        const bool act = SPEC ? decltype(act_c)::value : (p.act != 0);
        const bool res = SPEC ? decltype(res_c)::value : (p.residual != nullptr);
        const bool o32 = SPEC ? decltype(o32_c)::value : (p.out32 != nullptr);
        const bool o16 = SPEC ? decltype(o16_c)::value : (p.out16 != nullptr);
        if constexpr (SPEC && decltype(o16_c)::value && !decltype(o32_c)::value && !decltype(res_c)::value) {
            // 16-bit output only (conv1-5, q|k|v, fc1): SW / 8 lanes x 8 columns per row, ONE 16-byte store per lane and pass
            // (8-byte stores are issue-bound at 2.1-2.8 TB/s on this chip, 16-byte ones reach 5 TB/s: profiles/r02_gemm16_variants.md)
            // (16-byte stores: every row start of the 16-bit output must be 16-byte aligned, not just 8)
            if (!(p.N & 7) && !(p.ldo & 7) && !(p.o_bs & 7) && !((uintptr_t)p.out16 & 15)) {
                constexpr int LPR = SW / 8, RPS = 64 / LPR, NPASS = 32 / RPS;  // lanes per row, rows per pass, passes per round
                const int c8 = (lane % LPR) * 8;
                float4 b0[NJB], b1[NJB];
#pragma unroll
                for (int jb = 0; jb < NJB; ++jb) {
                    const int n8 = n0 + wc * 64 + jb * SW + c8;
                    b0[jb] = b1[jb] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (p.bias && (FULL || n8 < p.N)) {
                        b0[jb] = *(const float4*)(p.bias + n8);
                        b1[jb] = *(const float4*)(p.bias + n8 + 4);
                    }
                }
#pragma unroll
                for (int i = 0; i < MI; ++i) {
#pragma unroll
                    for (int jb = 0; jb < NJB; ++jb) {
                        const int n8 = n0 + wc * 64 + jb * SW + c8;
                        const bool n8_ok = FULL || n8 < p.N;
#pragma unroll
                        for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                stg[((r & 3) + 8 * (r >> 2) + 4 * half) * SW + jj * 32 + l31] = acc[i][jb * NJ + jj][r];
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                        for (int tt = 0; tt < NPASS; ++tt) {
                            const int row = tt * RPS + lane / LPR;
                            float4 v = *(const float4*)(stg + row * SW + c8);
                            float4 w = *(const float4*)(stg + row * SW + c8 + 4);
                            const int m = m0 + wr * WTM + i * 32 + row;
                            if (FULL || (m < p.M && n8_ok)) {
                                v.x += b0[jb].x; v.y += b0[jb].y; v.z += b0[jb].z; v.w += b0[jb].w;
                                w.x += b1[jb].x; w.y += b1[jb].y; w.z += b1[jb].z; w.w += b1[jb].w;
                                if (act) {
                                    gelu_fast4(v);
                                    gelu_fast4(w);
                                }
                                const long o = ob + (long)m * p.ldo + n8;
                                const uint4 pk = make_uint4(Cvt<T>::pack2(v.x, v.y), Cvt<T>::pack2(v.z, v.w), Cvt<T>::pack2(w.x, w.y),
                                                            Cvt<T>::pack2(w.z, w.w));
                                if (S3_GPROBE(p, 16)) S3_GKEEP4(pk);
                                else *(uint4*)((store_t*)p.out16 + o) = pk;
                            }
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                }
                if (FULL) n_stores = MI * NJB * NPASS;
                return;
            }
        }
        constexpr int LPR = SW / 4, RPS = 64 / LPR, NPASS = 32 / RPS;
        const int c4 = (lane % LPR) * 4;
        float4 bias4[NJB];
#pragma unroll
        for (int jb = 0; jb < NJB; ++jb) {
            const int n = n0 + wc * 64 + jb * SW + c4;
            bias4[jb] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias && (FULL || n < p.N)) bias4[jb] = *(const float4*)(p.bias + n);  // N % 4 == 0: a float4 is inside or outside as a whole
        }
        // (ONE register set: the row of pass t is re-loaded for the next round as soon as pass t has consumed it — a second set
        // pushes the 256 x 256 tile's epilogue over 256 registers)
        float4 rsb[RESPF ? NPASS : 1];
        auto res_load1 = [&](int rd, int t) -> float4 {  // the residual of pass t of staging round rd = i * NJB + jb
            const int i = rd / NJB, jb = rd % NJB;
            const int n = n0 + wc * 64 + jb * SW + c4;
            const int m = m0 + wr * WTM + i * 32 + t * RPS + lane / LPR;
            if (FULL || (m < p.M && n < p.N)) return *(const float4*)(p.residual + ob + (long)m * p.ldo + n);
            return make_float4(0.f, 0.f, 0.f, 0.f);
        };
        // res_ln_*: the residual rows are the INPUT of a LayerNorm; its output is rebuilt here from the rows' (mean, rstd) — they ride
        // with the residual prefetch — and the columns' gamma / beta (two float4 per staging round)
        // (lane l keeps the statistics of row l & 31 of the current 32-row block — two registers, fetched one block ahead — and a pass
        //  reads its row's pair from that lane by ds_bpermute: eight prefetched float2 per lane spilled the 256-row tile's epilogue)
        constexpr bool RESLN = RESPF && decltype(ln_c)::value;
        float2 rst_cur = make_float2(0.f, 1.f), rst_nxt = make_float2(0.f, 1.f);
        auto st_load = [&](int i) -> float2 {
            const int m = m0 + wr * WTM + i * 32 + l31;
            return p.res_ln_stats[m < p.M ? m : p.M - 1];
        };
        if constexpr (RESPF) {
#pragma unroll
            for (int t = 0; t < NPASS; ++t) rsb[t] = res_load1(0, t);
            if constexpr (RESLN) rst_nxt = st_load(0);
            do_pre();
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int jb = 0; jb < NJB; ++jb) {
                const int rd = i * NJB + jb;
                const int n = n0 + wc * 64 + jb * SW + c4;
                const bool n_ok = FULL || n < p.N;
                float4 lg = make_float4(1.f, 1.f, 1.f, 1.f), lb = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (RESLN) {
                    if (n_ok) {
                        lg = *(const float4*)(p.res_ln_g + n);
                        lb = *(const float4*)(p.res_ln_b + n);
                    }
                    if (jb == 0) {
                        rst_cur = rst_nxt;
                        if (i + 1 < MI) rst_nxt = st_load(i + 1);
                    }
                }
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        stg[((r & 3) + 8 * (r >> 2) + 4 * half) * SW + jj * 32 + l31] = acc[i][jb * NJ + jj][r];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int t = 0; t < NPASS; ++t) {
                    const int row = t * RPS + lane / LPR;
                    float4 v = *(const float4*)(stg + row * SW + c4);
                    const int m = m0 + wr * WTM + i * 32 + row;
                    if (FULL || (m < p.M && n_ok)) {
                        v.x += bias4[jb].x; v.y += bias4[jb].y; v.z += bias4[jb].z; v.w += bias4[jb].w;
                        if (act) {
                            gelu_fast4(v);
                        }
                        const long o = ob + (long)m * p.ldo + n;
                        if (res) {
                            float4 rs;
                            if constexpr (RESPF) rs = rsb[t];
                            else rs = *(const float4*)(p.residual + o);
                            if constexpr (RESLN) {  // the LayerNorm output of the stored row (norm.hip evaluates the same ln_affine)
                                const int src = (t * RPS + lane / LPR) * 4;  // (byte address of the lane that holds this row's pair)
                                const float mu = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(rst_cur.x)));
                                const float rr = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(rst_cur.y)));
                                rs.x = ln_affine(rs.x, mu, rr, lg.x, lb.x);
                                rs.y = ln_affine(rs.y, mu, rr, lg.y, lb.y);
                                rs.z = ln_affine(rs.z, mu, rr, lg.z, lb.z);
                                rs.w = ln_affine(rs.w, mu, rr, lg.w, lb.w);
                            }
                            v.x += rs.x; v.y += rs.y; v.z += rs.z; v.w += rs.w;
                        }
                        if (!SPEC && m >= limit) v = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (S3_GPROBE(p, 16)) {
                            S3_GKEEP4(v);
                        } else {
                            if (o32) *(float4*)(p.out32 + o) = v;
                            if (o16) *(uint2*)((store_t*)p.out16 + o) = make_uint2(Cvt<T>::pack2(v.x, v.y), Cvt<T>::pack2(v.z, v.w));
                        }
                    }
                    if constexpr (RESPF) {
                        if (rd + 1 < MI * NJB) rsb[t] = res_load1(rd + 1, t);  // in flight over the next round's LDS transpose
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
        // (a residual epilogue loads between its stores: each wait for a load retires every older store as well — nothing to count)
        if (FULL && SPEC && !decltype(res_c)::value && (decltype(o32_c)::value != decltype(o16_c)::value)) n_stores = MI * NJB * NPASS;
    };
    using TT = std::true_type;
    using FF = std::false_type;
    const bool a = p.act != 0, r = p.residual != nullptr, w32 = p.out32 != nullptr, w16 = p.out16 != nullptr;
    // FULL instantiations exist only where their fixed store count is used (OVL, a following tile, no residual loads)
    const bool full = OVL && want_count && full_tile && !r && !p.row_limit && !S3_GPROBE(p, 16);
    if (p.row_limit) epilogue(FF{}, FF{}, FF{}, FF{}, FF{}, FF{}, FF{});                         // generic (proj: padded-frame zeroing)
    else if (a && !r && !w32 && w16) {                                                     // conv1-5, fc1
        if constexpr (OVL) { if (full) { epilogue(TT{}, TT{}, FF{}, FF{}, TT{}, TT{}, FF{}); return n_stores; } }
        epilogue(TT{}, TT{}, FF{}, FF{}, TT{}, FF{}, FF{});
    } else if (!a && !r && !w32 && w16) {                                                  // q|k|v
        if constexpr (OVL) { if (full) { epilogue(TT{}, FF{}, FF{}, FF{}, TT{}, TT{}, FF{}); return n_stores; } }
        epilogue(TT{}, FF{}, FF{}, FF{}, TT{}, FF{}, FF{});
    } else if (!a && r && w32 && !w16) {                                                   // out_proj, fc2
        if (p.res_ln_stats) epilogue(TT{}, FF{}, TT{}, TT{}, FF{}, FF{}, TT{});            // (fc2 of a post-LN layer: the residual rows are LayerNorm inputs)
        else epilogue(TT{}, FF{}, TT{}, TT{}, FF{}, FF{}, FF{});
    }
    else if (a && !r && w32 && !w16) epilogue(TT{}, TT{}, FF{}, TT{}, FF{}, FF{}, FF{});         // last conv (feeds the fp32 LayerNorm)
    else epilogue(FF{}, FF{}, FF{}, FF{}, FF{}, FF{}, FF{});
    return 0;
    }
    };
    if constexpr (!PERSIST) {
        (void)run_epilogue(false, []() {});
    } else {
        const bool two = OVL && nk > 1;  // the first TWO K steps of a tile go out together
        auto issue_first = [&]() {
            issue(0, 0);
            if (two) issue(1, 1);
        };
        issue_first();
        int pend = 0;  // store instructions of my epilogue that sit behind the DMA of the tile about to start
        for (;;) {
            // the first stage(s) of this tile have landed; every wave is done with the previous tile's epilogue staging
            if constexpr (OVL) {
                wait_vm_keep(pend);
                barrier_lgkm();
            } else {
                barrier_all();
            }
            const int next = tile + tile_step;
            const bool has_next = next < tile_end;  // workgroup-uniform
            int nm0 = 0, nn0 = 0, nb = 0;
            if (has_next) coords(next, nm0, nn0, nb);
            for (int kt = 0; kt < nk; ++kt) {
                const bool pre = two && kt == 0;        // K step 1 is already in flight (or landed)
                compute(kt & 1, !pre && kt + 1 < nk, kt + 1, (kt + 1) & 1);
                if (pre) barrier_lgkm();  // nothing of mine is awaited: K step 1 landed with the wait that opened the tile
                else barrier_all();
            }
            // the next tile's first K step(s) go out before (a residual epilogue: just inside) this tile's epilogue: the stage
            // buffers are free
            pend = run_epilogue(has_next, [&]() {
                if (has_next) {
                    set_ptrs(nm0, nn0, nb);
                    issue_first();
                }
            });
            if (!has_next) break;
            tile = next;
            m0 = nm0;
            n0 = nn0;
            b = nb;
            zero_acc();
        }
    }
}

template <typename T, int WTM, int ROWB, int NST, int WPE, int WN = 4, bool PERSIST = false, bool OVL_ = false, bool SWAP = false, int PP = 0,
          bool MXW = false>
hipError_t big_go(const GemmParams& p, hipStream_t stream) {
    constexpr int BM = 2 * WTM, BN = 64 * WN, NTHR = 128 * WN;
    constexpr bool OVL = PERSIST && OVL_;
    constexpr int lds = big_lds_bytes(BM, BN, ROWB, NST, WN, PERSIST, OVL, SWAP, MXW);
    static_assert(!OVL || NST == 2, "OVL is written for the 2-stage pipeline");
    static_assert(!SWAP || PERSIST, "SWAP is instantiated for the persistent loop only");
    static_assert(lds * (WPE * 4 * 64 / NTHR) <= 160 * 1024, "workgroups per CU x LDS");
    auto kern = gemm16_big_kernel<T, WTM, ROWB, NST, WPE, WN, PERSIST, OVL_, SWAP, PP, MXW>;
    hipError_t e = ensure_dynamic_lds<gemm16_big_kernel<T, WTM, ROWB, NST, WPE, WN, PERSIST, OVL_, SWAP, PP, MXW>>(lds);
    if (e != hipSuccess) return e;
    long tiles = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * p.batches;
    if (PERSIST) {
        // `reserve_cus` (measurement knob, default 0): leave CUs out of the grid for somebody else's long-running workgroups (the
        // channel kernels of a collective).  Measured NOT to pay: foreign workgroups cost +2 %, a smaller grid +20 %
        // (profiles/r05_cu_contention.md)
        int cus = device_cus() - tuning().reserve_cus;
        cus = cus < 8 ? 8 : cus;
        if (tiles > cus) tiles = cus;
    }
    dim3 grid((unsigned)tiles);
    hipLaunchKernelGGL(kern, grid, dim3(NTHR), lds, stream, p);
    return hipGetLastError();
}

template <typename T>
hipError_t big_mode(int mode, const GemmParams& p, hipStream_t stream) {
    switch (mode) {
        case 1: {  // one workgroup per CU, 2 stages of 64 k: 256x256 (128 KiB) or 192x256 (112 KiB) tiles — whichever
                   // leaves fewer idle CU-rounds: time ~ ceil(tiles / 256 CUs) x tile rows (M = 15968, N = 768: 189 tiles
                   // of 256 rows use 74 % of the CUs, 252 tiles of 192 rows use 98 % and are 3/4 as long)
            const long nt = (p.N + 255) / 256;
            const long t256 = ((p.M + 255) / 256) * nt * p.batches, t192 = ((p.M + 191) / 192) * nt * p.batches;
            const long c256 = ((t256 + 255) / 256) * 256, c192 = ((t192 + 255) / 256) * 192;
            // (the smaller tile is ~8 % less efficient per row: conv1 890 vs 840 TF at equal CU utilisation)
            return c192 * 11 < c256 * 10 ? big_go<T, 96, 128, 2, 2>(p, stream) : big_go<T, 128, 128, 2, 2>(p, stream);
        }
        case 7: case 8: case 9: case 10: {  // mode 1 with the persistent tile loop (one workgroup per CU walks its tiles);
                                            // 8: + OVL; 9: + SWAP (row-per-lane epilogue) where the epilogue allows it; 10: + both
            const long nt = (p.N + 255) / 256;
            const long t256 = ((p.M + 255) / 256) * nt * p.batches, t192 = ((p.M + 191) / 192) * nt * p.batches;
            const long c256 = ((t256 + 255) / 256) * 256, c192 = ((t192 + 255) / 256) * 192;
            const bool small = c192 * 11 < c256 * 10;
            // the row-per-lane epilogue: 16-bit output only, no residual / row limit, whole 16-byte pieces of a row
            // Measured (profiles/r04_gemm16_epilogue.md): a GELU epilogue gains 3-5 % from it (fc1 93.2 -> 88.7 us), a plain one
            // LOSES 2-3 % (q|k|v 63.6 -> 64.7 us: its epilogue is store drain, and 32 rows x 32 bytes per store instruction drain
            // slower than 8 rows x 128) — mode 7 (the default) therefore takes it for GELU epilogues only, 9 / 10 wherever it is legal
            const bool rows_legal = p.out16 && !p.out32 && !p.residual && !p.row_limit && !(p.N & 7) && !(p.ldo & 7) &&
                                    !(p.o_bs & 7) && !((uintptr_t)p.out16 & 15) && !((uintptr_t)p.bias & 15);
            const bool rows_ok = rows_legal && (mode >= 9 || (mode == 7 && p.act != 0 && tuning().gemm16_rows != 0));
            const bool ovl = mode == 8 || mode == 10;
            if constexpr (std::is_same<T, f16_tag>::value) {
                if (p.mxw) {  // S3ENC_F16X2, second weight term on the MX pipe: the 192-row tile (its A-image registers fit beside 96 accumulators)
                    return rows_ok ? big_go<T, 96, 128, 2, 2, 4, true, false, true, 0, true>(p, stream)
                                   : big_go<T, 96, 128, 2, 2, 4, true, false, false, 0, true>(p, stream);
                }
            }
            if (!ovl && tuning().gemm16_pp) {  // DMA pieces placed per wave half (see PP above)
                const int pp = tuning().gemm16_pp;
#define S3_PP_GO(PPV)                                                                                                              \
    return rows_ok ? (small ? big_go<T, 96, 128, 2, 2, 4, true, false, true, PPV>(p, stream) : big_go<T, 128, 128, 2, 2, 4, true, false, true, PPV>(p, stream)) \
                   : (small ? big_go<T, 96, 128, 2, 2, 4, true, false, false, PPV>(p, stream) : big_go<T, 128, 128, 2, 2, 4, true, false, false, PPV>(p, stream))
#ifdef S3_GEMM_PP_LAB
                if (pp == 1) { S3_PP_GO(1); }
                if (pp == 2) { S3_PP_GO(2); }
#endif
                // the library builds PP 3 only (any non-zero key): the one placement that did not lose on random operands
                // (profiles/r05_gemm16_loop_probe.md); the lab binary builds all three
                S3_PP_GO(3);
#undef S3_PP_GO
            }
            if (rows_ok) {
                if (ovl) return small ? big_go<T, 96, 128, 2, 2, 4, true, true, true>(p, stream) : big_go<T, 128, 128, 2, 2, 4, true, true, true>(p, stream);
                return small ? big_go<T, 96, 128, 2, 2, 4, true, false, true>(p, stream) : big_go<T, 128, 128, 2, 2, 4, true, false, true>(p, stream);
            }
            if (ovl) return small ? big_go<T, 96, 128, 2, 2, 4, true, true>(p, stream) : big_go<T, 128, 128, 2, 2, 4, true, true>(p, stream);
            return small ? big_go<T, 96, 128, 2, 2, 4, true>(p, stream) : big_go<T, 128, 128, 2, 2, 4, true>(p, stream);
        }
        case 5: return big_go<T, 128, 128, 2, 2>(p, stream);  // 256x256 forced
        case 6: return big_go<T, 96, 128, 2, 2>(p, stream);   // 192x256 forced
        case 2: return big_go<T, 64, 128, 2, 2>(p, stream);   // 128x256,  96 KiB, one per CU
        case 4: return big_go<T, 64, 64, 3, 4>(p, stream);    // 128x256,  72 KiB ring of 3, two per CU
        // measured and dropped (profiles/r01_gemm16_variants.md): 256x256 with 64-byte stages in a ring of 3 / 4,
        // 128x256 ring of 4, 4-wave 256x128 two per CU — all 5-20 % behind modes 1 / 4 on every shape of the path
    }
    return hipErrorInvalidValue;
}

}  // namespace

bool gemm16_big_eligible(int dtype, const GemmParams& p) {
    if (dtype == F32 || tuning().gemm16_big == 0) return false;
    if ((p.K & 63) || (p.N & 3) || (p.ldo & 3) || (p.o_bs & 3)) return false;
    if (((p.lda * 2) & 15) || ((p.a_bs * 2) & 15)) return false;
    const uintptr_t al = (uintptr_t)p.A | (uintptr_t)p.W | (uintptr_t)p.out32 | (uintptr_t)p.residual | (uintptr_t)p.bias;
    if (al & 15) return false;
    if (((uintptr_t)p.out16) & 7) return false;
    // no lower bound on M: loads clamp and stores mask ragged rows, and a batch of few frames must take the kernel (and with it
    // the k grouping per MFMA, i.e. the rounding) its rows would take inside a large batch
    return p.N >= 128;
}

// GemmParams::res_ln_*: only the specialised residual epilogue of gemm16_big_kernel rebuilds LayerNorm rows — the combination the
// post-LN layers' fc2 has (fp32 output, no activation, no 16-bit copy, no row limit, one batch).  A property of the call's shape
// class, never of M: the engine asks before it decides what LayerNorm 1 writes.
bool gemm16_res_ln_ok(int dtype, const GemmParams& p) {
    return gemm16_big_eligible(dtype, p) && p.residual && p.out32 && !p.out16 && !p.act && !p.row_limit && p.batches == 1 && p.o_bs == 0 &&
           p.ldo == p.N;
}

bool gemm16_mx_eligible(int dtype, const GemmParams& p) {
    return dtype == F16 && p.W4 && p.W4s && !(p.K & 127) && !((uintptr_t)p.W4 & 15) && !((uintptr_t)p.W4s & 3) && gemm16_big_eligible(dtype, p) &&
           tuning().gemm16_mx != 0 && tuning().gemm16_big != 0;
}

// Which WEIGHTS get an MX image at all (s3enc_create).  The MX K step exists for the 192-row tile only (its A-image registers fit
// beside 96 accumulators, not beside 128), and a shape whose 256-row tiling needs fewer CU-rounds x rows is faster on the two-term
// loop (HuBERT-large fc2: ONE round of 252 tiles against two of 336 — 306 vs 237 us, profiles/r05_mx_second_term.md).  The rule is
// evaluated per weight over the path's reference batches (below), NOT per call: which kernel multiplies a row must
// not depend on the batch the row sits in — one utterance alone, a data-parallel shard and the full batch give the same bits.
// Round 6: summed over three reference batches (8 / 32 / 64 utterances of 10 s: M = 3 992 / 15 968 / 31 936 rows) instead of the
// bench's own 32 x 10 s alone — the decision is the same for every weight of the base and large models (base q|k|v, fc1, fc2,
// conv1 and large q|k|v take the 192-row MX tile; large fc1 / fc2 do not), so no result bit moved; it is a property of the
// weight's shape class, not of one benchmark batch.
bool gemm16_mx_weight_rule(long N, long K) {
    if ((K & 127) || N < 128) return false;
    const long nt = (N + 255) / 256;
    long c256 = 0, c192 = 0;  // row-rounds of the 256-row two-term tile against the 192-row MX tile over 256 CUs
    for (long M : {3992L, 15968L, 31936L}) {
        const long t256 = ((M + 255) / 256) * nt, t192 = ((M + 191) / 192) * nt;
        c256 += ((t256 + 255) / 256) * 256;
        c192 += ((t192 + 255) / 256) * 192;
    }
    return c192 <= c256;
}

hipError_t launch_gemm16_big(int dtype, const GemmParams& p, hipStream_t stream) {
    int mode = tuning().gemm16_big;
    if (p.mxw) mode = 7;  // (the MX K step lives in the persistent loop)  // 0 off, 3 = choose by shape, 1 / 2 / 4 / 5 / 6 = force one configuration (see big_mode)
    if (mode == 3) {
        // measured on MI355X (tools/gemm_bench.py, profiles/r02_gemm16_variants.md, and in the full forward with
        // `bench.py --tune gemm16_big=1|4`): with 16-byte stores in the 16-bit epilogues the one-workgroup-per-CU 256x256 /
        // 192x256 tile (mode 1 picks the one that leaves fewer idle CU-rounds) wins on every shape of the path —
        // q|k|v 64 vs 72 us, fc1 99 vs 104, out_proj 33 vs 37 against two 128x256 workgroups per CU (mode 4), which
        // round 1 preferred for K = 768 when the epilogue's 8-byte stores were the longer part of a tile
        mode = 7;  // round 3: mode 1's tiles, walked by one persistent workgroup per CU (bit-identical; q|k|v 70.9 -> 66.4 us)
    }
    return dtype == BF16 ? big_mode<bf16_tag>(mode, p, stream) : big_mode<f16_tag>(mode, p, stream);
}

}  // namespace s3
