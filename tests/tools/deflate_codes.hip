// Host-side dump of what the __host__ __device__ functions of csrc/twx_deflate.h produce -- the functions the kernels call --
// for tests/test_deflate_oracle.py to compare with oracle/deflate_oracle.py on the CPU, no GPU needed:
//   deflate_codes                      "M L symbol extra nextra" for every match length, "S piece segment nsym sample"
//   deflate_codes tokens FILE          "T symbol extra nextra" per token of the file's bytes, pieces of TWX_DF_PIECE
//   deflate_codes table FILE           FILE: TWX_DF_NSYM counts (text) -> "C symbol length code", "H bits", "W word" of df_build_table
//   hipcc --offload-arch=gfx950 -Iinclude -Itopowx_amd/csrc tests/tools/deflate_codes.hip -o /tmp/deflate_codes
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "twx_deflate.h"

int main(int argc, char **argv)
{
    if (argc > 2 && !strcmp(argv[1], "tokens")) {
        FILE *f = fopen(argv[2], "rb");
        if (!f) return 2;
        std::vector<uint8_t> b;
        int c;
        while ((c = fgetc(f)) != EOF) b.push_back((uint8_t)c);
        fclose(f);
        for (size_t p0 = 0; p0 < b.size(); p0 += TWX_DF_PIECE) {
            const int len = (int)(b.size() - p0 < TWX_DF_PIECE ? b.size() - p0 : TWX_DF_PIECE);
            df_piece<false>(&b[p0], len, p0 ? (int)b[p0 - 1] : 256, [](int run, int pv, int c) {
                df_step_tokens(run, pv, c, [](int sym, unsigned ex, unsigned ne) { printf("T %d %u %u\n", sym, ex, ne); });
            });
        }
        return 0;
    }
    if (argc > 2 && !strcmp(argv[1], "table")) {
        FILE *f = fopen(argv[2], "r");
        if (!f) return 2;
        std::vector<uint32_t> h(TWX_DF_NSYM, 0);
        for (int i = 0; i < TWX_DF_NSYM; ++i) if (fscanf(f, "%u", &h[i]) != 1) return 3;
        fclose(f);
        static DfTable t;
        df_build_table(h.data(), &t);
        for (int i = 0; i < TWX_DF_NSYM; ++i) printf("C %d %u %u\n", i, (unsigned)t.len[i], (unsigned)t.code[i]);
        printf("H %u\n", t.hdr_bits);
        for (unsigned i = 0; i < (t.hdr_bits + 31) / 32; ++i) printf("W %u\n", t.hdr[i]);
        return 0;
    }
    for (int L = 3; L <= TWX_DF_PIECE + 2; ++L) { unsigned ex, ne; const int s = df_len_sym(L, ex, ne); printf("M %d %d %u %u\n", L, s, ex, ne); }
    printf("S %d %d %d %d %lld %lld\n", TWX_DF_PIECE, TWX_DF_SEG, TWX_DF_NSYM, TWX_DF_SAMPLE, (long long)df_lo_bytes(1000000), (long long)df_slot_bytes(1000000));
    return 0;
}
