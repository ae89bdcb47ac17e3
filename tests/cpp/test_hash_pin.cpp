// test_hash_pin.cpp -- the C++ host mirror's cache-name hash against values the REFERENCE's own DJBHash
// (brutefir/hash.c:113-124, compiled in place: oracle/Makefile `ref`) returned; the cases come from
// tests/golden/djb_hash_ref.json, flattened by tests/test_wavio.py into "hexbytes expected" lines on stdin.
// No GPU call is made (the equalizer class itself needs a device; its static helpers do not).
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "../../foo-dsp-bfir_amd/host/equalizer_hip.hpp"

int main()
{
    std::string hex;
    unsigned long want;
    int n = 0, bad = 0;
    while (std::cin >> hex >> want) {
        if (hex == "-") hex.clear();
        std::vector<char> data(hex.size() / 2);
        for (size_t i = 0; i < data.size(); i++) data[i] = (char)strtoul(hex.substr(2 * i, 2).c_str(), nullptr, 16);
        const unsigned int got = equalizer::djb_hash(data.data(), data.size());
        if (got != (unsigned int)want) { printf("MISMATCH case %d: got %u want %lu\n", n, got, want); bad++; }
        // a band table: the file name carries the same hash in hex
        if (!data.empty() && data.size() % (3 * sizeof(double)) == 0) {
            const int nb = (int)(data.size() / (3 * sizeof(double)));
            const double *d = (const double *)data.data();
            std::vector<double> al(d, d + 3 * nb);   // aligned copy
            const std::string name = equalizer::make_filename_for(65536, 8, 2, 44100, nb, al.data(), al.data() + nb, al.data() + 2 * nb);
            char exp[128];
            snprintf(exp, sizeof(exp), "eq-%x-32768-8-2-44100.wav", (unsigned int)want);
            if (name != exp) { printf("MISMATCH name %s vs %s\n", name.c_str(), exp); bad++; }
        }
        n++;
    }
    printf("%d cases, %d mismatches\n", n, bad);
    if (n == 0 || bad) return 1;
    printf("ALL OK\n");
    return 0;
}
