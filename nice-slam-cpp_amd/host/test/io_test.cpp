// io_test.cpp -- CPU-only driver of the I/O stand-ins (host/src/nsk_io.cpp): decodes the PNG / EXR files tests/test_host_io.py wrote,
// runs cv::remap and the CoFusionReader on them and dumps what it read as .npy for the test to compare.  No GPU, no nsk context.
#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "inputs/CoFusionReader.h"
#include "inputs/SequenceReader.h"

static void save_npy(const std::string& path, const float* data, const std::vector<int>& shape)
{
    std::ostringstream sh; sh << "(";
    size_t n = 1;
    for (size_t i = 0; i < shape.size(); ++i) { sh << shape[i] << (shape.size() == 1 || i + 1 < shape.size() ? "," : ""); n *= (size_t)shape[i]; }
    sh << ")";
    std::string hdr = "{'descr': '<f4', 'fortran_order': False, 'shape': " + sh.str() + ", }";
    while ((10 + hdr.size() + 1) % 64 != 0) hdr += ' ';
    hdr += '\n';
    std::ofstream f(path, std::ios::binary);
    f.write("\x93NUMPY\x01\x00", 8);
    uint16_t hl = (uint16_t)hdr.size();
    f.write((const char*)&hl, 2); f.write(hdr.data(), hdr.size()); f.write((const char*)data, n * sizeof(float));
}
static void save_mat(const std::string& path, const cv::Mat& m)
{
    cv::Mat f; m.convertTo(f, CV_32F);
    std::vector<int> shape{m.rows, m.cols};
    if (m.channels() > 1) shape.push_back(m.channels());
    save_npy(path, f.ptr<float>(), shape);
}

int main(int argc, char** argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: io_test <dir>\n"); return 2; }
    const std::string d = std::string(argv[1]) + "/";
    try {
        for (const char* name : {"rgb8", "rgba8", "gray8", "gray16"}) {
            cv::Mat m = cv::imread(d + name + ".png", cv::IMREAD_UNCHANGED);
            if (m.empty()) { std::fprintf(stderr, "imread failed on %s\n", name); return 1; }
            save_mat(d + "out_" + name + ".npy", m);
        }
        { cv::Mat m = cv::imread(d + "gray8.png", cv::IMREAD_COLOR); if (m.channels() != 3) return 1; save_mat(d + "out_gray8_color.npy", m); }
        if (!cv::imread(d + "missing.png").empty() || !cv::imread(d + "not_a.png").empty()) { std::fprintf(stderr, "imread must return an empty Mat on failure\n"); return 1; }
        for (const char* name : {"f_none", "f_zip", "f_zips", "h_zip_rgba", "f_rle", "u_none"}) {
            float* out = nullptr; int w = 0, h = 0; const char* err = nullptr;
            if (LoadEXR(&out, &w, &h, (d + name + ".exr").c_str(), &err) != TINYEXR_SUCCESS) { std::fprintf(stderr, "LoadEXR %s: %s\n", name, err ? err : "?"); return 1; }
            save_npy(d + "out_" + name + ".npy", out, {h, w, 4});
            std::free(out);
        }
        { float* out = nullptr; int w, h; const char* err = nullptr; if (LoadEXR(&out, &w, &h, (d + "missing.exr").c_str(), &err) == TINYEXR_SUCCESS || !err) return 1; FreeEXRErrorMessage(err); }
        // cv::remap, INTER_LINEAR, zero border
        {
            cv::Mat src(3, 4, CV_32FC1);
            for (int i = 0; i < 12; ++i) src.ptr<float>()[i] = (float)(i * i);
            const float xs[6] = {0.f, 1.5f, 2.25f, 3.f, -0.5f, 3.5f}, ys[6] = {0.f, 0.5f, 1.75f, 2.f, 1.f, 2.5f};
            cv::Mat mx(1, 6, CV_32FC1), my(1, 6, CV_32FC1), dst;
            for (int i = 0; i < 6; ++i) { mx.ptr<float>()[i] = xs[i]; my.ptr<float>()[i] = ys[i]; }
            cv::remap(src, dst, mx, my, cv::INTER_LINEAR);
            save_mat(d + "out_remap.npy", dst);
            cv::Mat e = cv::Mat::eye(3, 3, CV_32F);
            if (e.at<float>(1, 1) != 1.f || e.at<float>(0, 1) != 0.f) return 1;
        }
        // the dataset reader on a two-frame CoFusion-style sequence (frames 3 and 4: the reference starts at fptr = 3)
        {
            CoFusionReader r(d + "seq/");
            r.n_imgs = 4;
            int n = 0;
            while (r.hasMore()) {
                const int idx = r.getIdx();
                r.getNext();
                save_mat(d + "out_seq_depth" + std::to_string(idx) + ".npy", r.depth);
                save_mat(d + "out_seq_rgb" + std::to_string(idx) + ".npy", r.rgb);
                ++n;
            }
            if (n != 2 || r.getIdx() != 5 || r.c2w(0, 0) != 1.f || r.c2w(0, 1) != 0.f || r.width != 8 || r.height != 6) { std::fprintf(stderr, "reader state wrong\n"); return 1; }
        }
        // the posed-sequence readers (Replica / ScanNet / TUM layouts): every frame's colour, metric depth and OpenGL pose
        {
            const struct { SequenceReader::Kind kind; const char* dir; } sets[3] = {{SequenceReader::Replica, "replica"}, {SequenceReader::ScanNet, "scannet"}, {SequenceReader::TUM, "tum"}};
            for (const auto& st : sets) {
                SequenceReader r(st.kind, d + st.dir, 10.f);
                std::vector<float> P;
                int n = 0;
                while (r.hasMore()) {
                    const int idx = r.getIdx();
                    r.getNext();
                    save_mat(d + "out_" + st.dir + "_depth" + std::to_string(idx) + ".npy", r.depth);
                    save_mat(d + "out_" + st.dir + "_rgb" + std::to_string(idx) + ".npy", r.rgb);
                    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) P.push_back(r.c2w(i, j));
                    ++n;
                }
                if (n != r.n_imgs || (int)r.poses.size() != n) { std::fprintf(stderr, "%s: frame count\n", st.dir); return 1; }
                save_npy(d + "out_" + st.dir + "_poses.npy", P.data(), {n, 4, 4});
                r.reset();
                if (!r.hasMore() || r.getIdx() != 0) return 1;
            }
            bool threw = false;
            try { SequenceReader bad(SequenceReader::Replica, d + "replica_jpg"); bad.getNext(); } catch (const std::exception& e) { threw = std::string(e.what()).find("PNG only") != std::string::npos; }
            if (!threw) { std::fprintf(stderr, "a JPEG colour image must be refused with a clear message\n"); return 1; }
        }
        std::printf("io_test ok\n");
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "io_test failed: %s\n", e.what());
        return 1;
    }
}
