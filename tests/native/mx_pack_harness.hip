// Host-only harness around pack_mx4_lo (s3prl_amd/csrc/engine_internal.h): reads an (N, K) fp32 matrix, writes the MX-fp4 image
// (data then scales) — tests/test_mx_pack_cpu.py compares it with a numpy restatement of the format.  No device code runs.
#include "engine_internal.h"

#include <cstdio>
#include <cstdlib>

int main(int argc, char** argv) {
    if (argc != 5) return 1;
    const long N = atol(argv[3]), K = atol(argv[4]);
    std::vector<float> w((size_t)N * K);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(w.data(), 4, w.size(), f) != w.size()) return 2;
    fclose(f);
    std::vector<uint8_t> d, s;
    pack_mx4_lo(w, N, K, d, s);
    f = fopen(argv[2], "wb");
    if (!f) return 3;
    fwrite(d.data(), 1, d.size(), f);
    fwrite(s.data(), 1, s.size(), f);
    fclose(f);
    return 0;
}
