// Host-side dump of the token codes csrc/twx_deflate.h emits (the __host__ __device__ functions the kernels call):
// every literal and every match length, as "kind value nbits bits".  tests/test_deflate_oracle.py compares them with the
// RFC 1951 tables of oracle/deflate_oracle.py and runs the piece tokenizer on sample bytes -- on the CPU, no GPU needed.
//   hipcc --offload-arch=gfx950 -Iinclude -Itopowx_amd/csrc tests/tools/deflate_codes.hip -o /tmp/deflate_codes
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "twx_deflate.h"

int main(int argc, char **argv)
{
    if (argc > 1) {      // tokens of the pieces of a byte file: "T nbits bits" per token, "P bits" per piece
        FILE *f = fopen(argv[1], "rb");
        if (!f) return 2;
        std::vector<uint8_t> b;
        int c;
        while ((c = fgetc(f)) != EOF) b.push_back((uint8_t)c);
        fclose(f);
        for (size_t p0 = 0; p0 < b.size(); p0 += TWX_DF_PIECE) {
            const int len = (int)(b.size() - p0 < TWX_DF_PIECE ? b.size() - p0 : TWX_DF_PIECE);
            const unsigned bits = df_piece<false>(&b[p0], len, p0 ? (int)b[p0 - 1] : 256, [](unsigned v, unsigned n) { printf("T %u %u\n", n, v); });
            const unsigned counted = df_piece<false>(&b[p0], len, p0 ? (int)b[p0 - 1] : 256, [](unsigned, unsigned) {});     // (the counting pass)
            printf("P %u %u\n", bits, counted);
        }
        return 0;
    }
    for (int c = 0; c < 256; ++c) { unsigned v; const unsigned n = df_literal(c, v); printf("L %d %u %u\n", c, n, v); }
    for (int L = 3; L <= 258; ++L) { unsigned v; const unsigned n = df_match(L, v); printf("M %d %u %u\n", L, n, v); }
    printf("S %d %d %d %lld %lld\n", TWX_DF_PIECE, TWX_DF_SEG, TWX_DF_SEG_OUT, (long long)df_lo_bytes(1000000), (long long)df_slot_bytes(1000000));
    return 0;
}
