// fbank.hip — the `fbank` baseline upstream (BASELINE configs[0]) on the GPU:
//   torchaudio.compliance.kaldi.fbank(80 mel bins, 25 ms / 10 ms, log) -> 2 x ComputeDeltas(win 5) -> CMVN over time
//   (upstream/baseline/extracter.py:32-90, baseline/fbank.yaml).
//
// Framing, DC removal, pre-emphasis, the povey window and the zero-padded 512-point real DFT are all linear in the
// 400 samples of a frame, so they are folded (in fp64, once per configuration) into ONE (2*257) x 400 matrix; the
// spectrum of every frame is then a GEMM whose A rows OVERLAP in the raw waveform (row t starts at sample 160 t:
// lda = 160 < K = 400) — the same implicit-GEMM addressing as the strided convs, served by gemm.hip's exact-fp32 MFMA
// kernel straight from the caller's PCM.  Three small HBM-bound kernels finish the job:
//   fbank_mel_kernel   |X|^2 -> 80 triangular mel filters -> log(max(., eps))          (one workgroup per frame)
//   fbank_delta_kernel delta and delta-delta with replicate padding                     (thread per (t, bin))
//   fbank_cmvn_kernel  per-dimension mean / unbiased std over time, two-pass, in place  (workgroup per dimension)
#include <cmath>
#include <map>
#include <mutex>
#include <vector>

#include "kernels.h"

namespace s3 {

namespace {

__global__ __launch_bounds__(128) void fbank_mel_kernel(const float* spec, int nbin, int ldspec, const float* banksT, int nmel,
                                                        float eps, float* out, int ldo) {
    extern __shared__ float pw[];
    const long t = blockIdx.x;
    const float* s = spec + t * ldspec;
    for (int k = threadIdx.x; k < nbin; k += blockDim.x) {
        const float re = s[k], im = s[nbin + k];
        pw[k] = re * re + im * im;
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nmel; b += blockDim.x) {
        float acc = 0.f;
        for (int k = 0; k < nbin; ++k) acc = fmaf(pw[k], banksT[k * nmel + b], acc);
        out[t * ldo + b] = logf(fmaxf(acc, eps));
    }
}

// out[t][nmel*(o+1) + b] = delta of order o+1; d[t] = sum_k k * x[clamp(t+k)] / denom  (replicate padding)
__global__ void fbank_delta_kernel(float* out, long T, int nmel, int ldo, int order, int n, float inv_denom) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T * nmel) return;
    const long t = idx / nmel;
    const int b = (int)(idx % nmel);
    auto x0 = [&](long tt) { return out[(tt < 0 ? 0 : (tt >= T ? T - 1 : tt)) * ldo + b]; };
    auto d1 = [&](long tt) {
        tt = tt < 0 ? 0 : (tt >= T ? T - 1 : tt);
        float a = 0.f;
        for (int k = -n; k <= n; ++k) a += (float)k * x0(tt + k);
        return a * inv_denom;
    };
    if (order >= 1) out[t * ldo + nmel + b] = d1(t);
    if (order >= 2) {
        float a = 0.f;
        for (int k = -n; k <= n; ++k) a += (float)k * d1(t + k);
        out[t * ldo + 2 * nmel + b] = a * inv_denom;
    }
}

__global__ __launch_bounds__(256) void fbank_cmvn_kernel(float* x, long T, int ldo, float eps) {
    __shared__ double red[4];
    const int f = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto block_sum = [&](double v) {
        v = wave_sum_d(v);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        return red[0] + red[1] + red[2] + red[3];
    };
    double s = 0.0;
    for (long t = threadIdx.x; t < T; t += 256) s += x[t * ldo + f];
    const double mean = block_sum(s) / (double)T;
    double q = 0.0;
    for (long t = threadIdx.x; t < T; t += 256) {
        const double d = x[t * ldo + f] - mean;
        q += d * d;
    }
    const double var = block_sum(q) / (double)(T - 1);  // unbiased (torch.std default); T == 1 -> nan like torch
    const float inv = (float)(1.0 / ((double)eps + sqrt(var)));
    for (long t = threadIdx.x; t < T; t += 256) x[t * ldo + f] = (float)((x[t * ldo + f] - mean)) * inv;
}

struct FbankPlan {
    int size, shift, padded, nbin, nmel;
    float* dft = nullptr;     // (2*nbin, size) fp32: rows 0..nbin-1 cos, nbin.. -sin, pre-processing folded in
    float* banksT = nullptr;  // (nbin, nmel)
};
// Scratch is per (device, STREAM): the plan's constants are shared, but two experts / two streams running the same
// configuration concurrently must not share the spectrum workspace (re-use within one stream is stream-ordered).
struct FbankScratch {
    float* spec = nullptr;    // workspace (frames, 2*nbin)
    size_t spec_elems = 0;
    float* stage = nullptr;   // aligned copy of a misaligned waveform
    size_t stage_elems = 0;
};

std::mutex g_mu;
std::map<std::vector<long>, FbankPlan> g_plans;
std::map<std::pair<int, uintptr_t>, FbankScratch> g_scratch;

hipError_t build_plan(FbankPlan& pl, int nmel, int size, int shift, double preemph, int sample_rate) {
    pl.size = size;
    pl.shift = shift;
    pl.padded = 1;
    while (pl.padded < size) pl.padded <<= 1;
    pl.nbin = pl.padded / 2 + 1;
    pl.nmel = nmel;
    const int N = size, nb = pl.nbin;
    // D = diag(window) * P(pre-emphasis, first sample replicated) * C(remove mean), as a dense N x N fp64 matrix
    std::vector<double> D((size_t)N * N, 0.0), tmp((size_t)N * N, 0.0);
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) tmp[(size_t)i * N + j] = (i == j ? 1.0 : 0.0) - 1.0 / N;  // C
    for (int i = 0; i < N; ++i) {
        const double w = std::pow(0.5 - 0.5 * std::cos(2.0 * M_PI * i / (N - 1)), 0.85);  // povey
        const int ip = i == 0 ? 0 : i - 1;
        for (int j = 0; j < N; ++j) D[(size_t)i * N + j] = w * (tmp[(size_t)i * N + j] - preemph * tmp[(size_t)ip * N + j]);
    }
    std::vector<float> M((size_t)2 * nb * N);
    std::vector<double> c(N), s(N);
    for (int r = 0; r < nb; ++r) {
        for (int i = 0; i < N; ++i) {
            const double a = 2.0 * M_PI * (double)((long)r * i % pl.padded) / pl.padded;
            c[i] = std::cos(a);
            s[i] = -std::sin(a);
        }
        for (int j = 0; j < N; ++j) {
            double re = 0.0, im = 0.0;
            for (int i = 0; i < N; ++i) {
                re += c[i] * D[(size_t)i * N + j];
                im += s[i] * D[(size_t)i * N + j];
            }
            M[(size_t)r * N + j] = (float)re;
            M[(size_t)(nb + r) * N + j] = (float)im;
        }
    }
    // kaldi mel banks (no VTLN): low 20 Hz, high = Nyquist, triangles in mel space; Nyquist column zero
    std::vector<float> bT((size_t)nb * nmel, 0.f);
    auto mel = [](double f) { return 1127.0 * std::log(1.0 + f / 700.0); };
    const double lo = mel(20.0), hi = mel(0.5 * sample_rate), delta = (hi - lo) / (nmel + 1);
    const double width = (double)sample_rate / pl.padded;
    for (int b = 0; b < nmel; ++b) {
        const double left = lo + b * delta, center = left + delta, right = center + delta;
        for (int k = 0; k < nb - 1; ++k) {
            const double m = mel(width * k);
            const double up = (m - left) / (center - left), down = (right - m) / (right - center);
            const double v = std::fmax(0.0, std::fmin(up, down));
            bT[(size_t)k * nmel + b] = (float)v;
        }
    }
    hipError_t e = hipMalloc((void**)&pl.dft, M.size() * 4);
    if (e != hipSuccess) return e;
    e = hipMemcpy(pl.dft, M.data(), M.size() * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) return e;
    e = hipMalloc((void**)&pl.banksT, bT.size() * 4);
    if (e != hipSuccess) return e;
    return hipMemcpy(pl.banksT, bT.data(), bT.size() * 4, hipMemcpyHostToDevice);
}

}  // namespace

long fbank_num_frames(long n, const FbankParams& c) {
    const int size = (int)(c.sample_rate * c.frame_length_ms * 0.001), shift = (int)(c.sample_rate * c.frame_shift_ms * 0.001);
    return n < size ? 0 : 1 + (n - size) / shift;  // snip_edges
}

// One utterance: wav (device, n samples) -> out (device, frames x ldo), columns [0, nmel*(order+1)).
hipError_t launch_fbank(const FbankParams& c, const float* wav, long n, float* out, int ldo, hipStream_t st) {
    const int size = (int)(c.sample_rate * c.frame_length_ms * 0.001), shift = (int)(c.sample_rate * c.frame_shift_ms * 0.001);
    const long T = fbank_num_frames(n, c);
    if (T <= 0) return hipSuccess;
    if (size <= 0 || shift <= 0 || (size & 3) || (shift & 3) || c.num_mel_bins <= 0 || c.delta_order < 0 || c.delta_order > 2)
        return hipErrorInvalidValue;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(g_mu);
    const std::vector<long> key{dev, c.num_mel_bins, size, shift, (long)std::lround(c.preemph * 1e6), c.sample_rate};
    FbankPlan& pl = g_plans[key];
    if (!pl.dft) {
        e = build_plan(pl, c.num_mel_bins, size, shift, c.preemph, c.sample_rate);
        if (e != hipSuccess) return e;
    }
    const int N2 = 2 * pl.nbin;
    FbankScratch& sc = g_scratch[std::make_pair(dev, (uintptr_t)st)];
    if ((size_t)T * N2 > sc.spec_elems) {
        if (sc.spec) (void)hipFree(sc.spec);  // device-synchronising: nothing in flight still uses it
        sc.spec = nullptr;
        sc.spec_elems = (size_t)T * N2 + (size_t)T * N2 / 4;
        e = hipMalloc((void**)&sc.spec, sc.spec_elems * 4);
        if (e != hipSuccess) return e;
    }
    const float* a = wav;
    if (((uintptr_t)wav) & 15) {  // the GEMM loads 16-byte vectors: stage a misaligned waveform once
        if ((size_t)n > sc.stage_elems) {
            if (sc.stage) (void)hipFree(sc.stage);
            sc.stage = nullptr;
            sc.stage_elems = (size_t)n + (size_t)n / 4;
            e = hipMalloc((void**)&sc.stage, sc.stage_elems * 4);
            if (e != hipSuccess) return e;
        }
        e = hipMemcpyAsync(sc.stage, wav, (size_t)n * 4, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return e;
        a = sc.stage;
    }
    GemmParams g{};
    g.A = a;
    g.lda = shift;  // overlapping rows: frame t starts at sample t * shift
    g.a_bs = 0;
    g.W = pl.dft;
    g.M = (int)T;
    g.N = N2;
    g.K = size;
    g.batches = 1;
    g.out32 = sc.spec;
    g.ldo = N2;
    g.o_bs = 0;
    e = launch_gemm(F32, g, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fbank_mel_kernel, dim3((unsigned)T), dim3(128), pl.nbin * sizeof(float), st, sc.spec, pl.nbin, N2,
                       pl.banksT, pl.nmel, 1.1920928955078125e-07f, out, ldo);
    if (c.delta_order > 0) {
        const int nw = (c.delta_win - 1) / 2;
        const float inv = 3.0f / (float)(nw * (nw + 1) * (2 * nw + 1));
        const long tot = T * pl.nmel;
        hipLaunchKernelGGL(fbank_delta_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, out, T, pl.nmel, ldo,
                           c.delta_order, nw, inv);
    }
    if (c.use_cmvn)
        hipLaunchKernelGGL(fbank_cmvn_kernel, dim3(pl.nmel * (c.delta_order + 1)), dim3(256), 0, st, out, T, ldo, c.cmvn_eps);
    return hipGetLastError();
}

}  // namespace s3
