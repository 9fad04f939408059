// io_fuzz.cpp -- malformed-file cases for the PNG and EXR readers of host/src/nsk_io.cpp (they replace OpenCV's imread and tinyexr in the
// drop-in): every truncation and a few thousand deterministic byte corruptions of the valid files tests/test_host_io.py wrote must come
// back as a clean failure or a decoded image, never as an out-of-bounds access.  Built with -fsanitize=address,undefined (make io_fuzz_asan),
// CPU only.   io_fuzz <dir with the fixture files>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "inputs/CoFusionReader.h"

static std::vector<unsigned char> slurp(const std::string& p)
{
    std::ifstream f(p, std::ios::binary);
    return std::vector<unsigned char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void spit(const std::string& p, const std::vector<unsigned char>& d, size_t n)
{
    std::ofstream f(p, std::ios::binary | std::ios::trunc);
    f.write((const char*)d.data(), (std::streamsize)n);
}

int main(int argc, char** argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: io_fuzz <dir>\n"); return 2; }
    const std::string d = std::string(argv[1]) + "/";
    const std::string tmp_png = d + "fuzz_tmp.png", tmp_exr = d + "fuzz_tmp.exr";
    unsigned long long lcg = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(lcg >> 33); };
    long cases = 0, decoded = 0;
    auto try_png = [&](const std::vector<unsigned char>& b, size_t n) {
        spit(tmp_png, b, n);
        cv::Mat m = cv::imread(tmp_png, cv::IMREAD_UNCHANGED);
        ++cases; if (!m.empty()) { ++decoded; volatile unsigned char sink = m.data[(size_t)m.rows * m.cols * m.channels() * (m.depth() == CV_16U ? 2 : 1) - 1]; (void)sink; }
    };
    auto try_exr = [&](const std::vector<unsigned char>& b, size_t n) {
        spit(tmp_exr, b, n);
        float* out = nullptr; int w = 0, h = 0; const char* err = nullptr;
        const int rc = LoadEXR(&out, &w, &h, tmp_exr.c_str(), &err);
        ++cases;
        if (rc == TINYEXR_SUCCESS) { ++decoded; volatile float sink = out[(size_t)w * h * 4 - 1]; (void)sink; std::free(out); }
        else if (err) FreeEXRErrorMessage(err);
    };
    for (const char* name : {"rgb8.png", "rgba8.png", "gray8.png", "gray16.png"}) {
        const std::vector<unsigned char> good = slurp(d + name);
        if (good.empty()) { std::fprintf(stderr, "missing fixture %s\n", name); return 1; }
        for (size_t n = 0; n <= good.size(); ++n) try_png(good, n);                       // every truncation
        for (int k = 0; k < 1500; ++k) {                                                     // corruptions: header bytes get most of them
            std::vector<unsigned char> b = good;
            const int nb = 1 + (int)(rnd() % 3);
            for (int q = 0; q < nb; ++q) { const size_t at = (k & 1) ? rnd() % std::min<size_t>(b.size(), 48) : rnd() % b.size(); b[at] = (unsigned char)rnd(); }
            try_png(b, b.size());
        }
    }
    for (const char* name : {"f_none.exr", "f_zip.exr", "f_zips.exr", "h_zip_rgba.exr", "f_rle.exr", "u_none.exr"}) {
        const std::vector<unsigned char> good = slurp(d + name);
        if (good.empty()) { std::fprintf(stderr, "missing fixture %s\n", name); return 1; }
        const size_t step = std::max<size_t>(1, good.size() / 400);
        for (size_t n = 0; n <= good.size(); n += (n < 512 ? 1 : step)) try_exr(good, n);
        for (int k = 0; k < 1500; ++k) {
            std::vector<unsigned char> b = good;
            const int nb = 1 + (int)(rnd() % 3);
            for (int q = 0; q < nb; ++q) { const size_t at = (k & 1) ? rnd() % std::min<size_t>(b.size(), 400) : rnd() % b.size(); b[at] = (unsigned char)rnd(); }
            try_exr(b, b.size());
        }
    }
    std::remove(tmp_png.c_str()); std::remove(tmp_exr.c_str());
    std::printf("io_fuzz ok: %ld malformed files, %ld still decoded\n", cases, decoded);
    return 0;
}
