// Host-only harness: which WEIGHTS get an MX image (gemm16.hip::gemm16_mx_weight_rule, per weight) and which calls the MX K step
// can take (gemm16_mx_eligible) — "M N K" triples in, "rule eligible" per triple out.  tests/test_boundary_cpu.py.
#include "kernels.h"

#include <cstdio>
#include <cstdlib>

int main(int argc, char** argv) {
    for (int i = 1; i + 2 < argc; i += 3) {
        s3::GemmParams g{};
        g.A = (const void*)(uintptr_t)256;
        g.W = (const void*)(uintptr_t)256;
        g.W4 = (const void*)(uintptr_t)256;
        g.W4s = (const void*)(uintptr_t)256;
        g.out16 = (void*)(uintptr_t)256;
        g.M = atoi(argv[i]);
        g.N = atoi(argv[i + 1]);
        g.K = atoi(argv[i + 2]);
        g.lda = g.K;
        g.batches = 1;
        g.ldo = g.N;
        g.o_bs = (long)g.M * g.N;
        g.wsplit = 1;
        g.mxw = 1;
        printf("%d %d\n", s3::gemm16_mx_weight_rule(g.N, g.K) ? 1 : 0, s3::gemm16_mx_eligible(s3::F16, g) ? 1 : 0);
    }
    return 0;
}
