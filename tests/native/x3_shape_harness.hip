// Host-only harness: the create-time predicate of the S3ENC_F16X2 hybrids (engine_internal.h::x3_shape_ok -> gemm_x3_eligible in
// libs3enc.so) on the shapes given as "N K lda" triples; prints one 0 / 1 per triple.  tests/test_boundary_cpu.py.
#include "engine_internal.h"

#include <cstdio>
#include <cstdlib>

int main(int argc, char** argv) {
    for (int i = 1; i + 2 < argc; i += 3) printf("%d\n", x3_shape_ok(atol(argv[i]), atol(argv[i + 1]), atol(argv[i + 2])) ? 1 : 0);
    return 0;
}
