// Mutation fuzzer of rt_image_read (host only): g++ -O1 -g -fsanitize=address,undefined -o /tmp/fuzz scripts/fuzz_image_read.cpp raytracer-rs_amd/csrc/rt_jpeg.cpp
// && /tmp/fuzz seed0.jpg seed1.ppm ...   (6 000 mutated copies of the seeds; clean under both sanitizers)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cstdint>
#include "../include/rtamd.h"
int main(int argc, char** argv) {
    std::vector<std::vector<uint8_t>> seeds;
    for (int i = 1; i < argc; i++) { FILE* f = fopen(argv[i], "rb"); if (!f) continue; std::vector<uint8_t> d; int c; while ((c = fgetc(f)) != EOF) d.push_back((uint8_t)c); fclose(f); seeds.push_back(d); }
    uint32_t x = 12345; auto rnd = [&]() { x = x * 1664525u + 1013904223u; return x >> 8; };
    std::vector<uint8_t> out(64 << 20);
    long ok = 0, bad = 0;
    for (int it = 0; it < 6000; it++) {
        std::vector<uint8_t> d = seeds[rnd() % seeds.size()];
        int nm = 1 + rnd() % 8;
        for (int k = 0; k < nm; k++) {
            int kind = rnd() % 4; size_t at = rnd() % d.size();
            if (kind == 0) d[at] = (uint8_t)rnd();
            else if (kind == 1) d[at] ^= 1u << (rnd() % 8);
            else if (kind == 2 && d.size() > 8) d.resize(at + 1);
            else { d[at] = 0xFF; if (at + 1 < d.size()) d[at + 1] = (uint8_t)(0xC0 + rnd() % 32); }
        }
        FILE* f = fopen("/tmp/fz.jpg", "wb"); fwrite(d.data(), 1, d.size(), f); fclose(f);
        uint32_t w = 0, h = 0;
        int rc = rt_image_read("/tmp/fz.jpg", &w, &h, nullptr, 0);
        if (rc == 0 && (uint64_t)w * h * 3 <= out.size()) rc = rt_image_read("/tmp/fz.jpg", &w, &h, out.data(), out.size());
        if (rc == 0) ok++; else bad++;
    }
    printf("decoded %ld, refused %ld\n", ok, bad);
    return 0;
}
