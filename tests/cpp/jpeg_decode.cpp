// test helper: decodes a JPEG file with include/hrbf_jpeg.h and writes "W H\n" + raw RGB to stdout (tests/test_cpp_io.py compares with Pillow)
#include <cstdio>
#include <fstream>
#include <iterator>
#include "hrbf_jpeg.h"
int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<uint8_t> b((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    try {
        int w = 0, h = 0;
        std::vector<uint8_t> rgb = hrbf_mi355::JpegDecoder::decodeRGB(b.data(), b.size(), w, h);
        std::printf("%d %d\n", w, h);
        std::fwrite(rgb.data(), 1, rgb.size(), stdout);
    } catch (const std::exception &e) { std::fprintf(stderr, "%s\n", e.what()); return 1; }
    return 0;
}
