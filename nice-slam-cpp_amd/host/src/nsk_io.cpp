// nsk_io.cpp -- the I/O side of the drop-in: the cv::Mat / cv::imread / cv::remap and LoadEXR stand-ins
// declared under host/include/{opencv2,tinyexr.h} and the CoFusion dataset reader with the reference's surface
// (include/inputs/CoFusionReader.h:17-37, src/inputs/CoFusionReader.cpp).  Written from the file formats' specifications (PNG: RFC 2083;
// OpenEXR file layout), zlib is the only dependency; nothing here is on the GPU path.
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <stdexcept>

#include "inputs/CoFusionReader.h"

// ---------------------------------------------------------------------------------------------------------
// cv::Mat
// ---------------------------------------------------------------------------------------------------------
namespace cv {

static inline double load_scalar(const unsigned char* p, int depth)
{
    switch (depth) {
    case CV_8U: return *p;
    case CV_16U: { uint16_t v; std::memcpy(&v, p, 2); return v; }
    default: { float v; std::memcpy(&v, p, 4); return v; }
    }
}
static inline void store_scalar(unsigned char* p, int depth, double v)
{
    switch (depth) {
    case CV_8U: *p = (unsigned char)std::min(255.0, std::max(0.0, std::nearbyint(v))); break;
    case CV_16U: { uint16_t u = (uint16_t)std::min(65535.0, std::max(0.0, std::nearbyint(v))); std::memcpy(p, &u, 2); break; }
    default: { float f = (float)v; std::memcpy(p, &f, 4); }
    }
}

void Mat::convertTo(Mat& dst, int rtype, double alpha, double beta) const
{
    const int ddepth = rtype & 7, cn = channels();
    Mat out(rows, cols, CV_MAKETYPE(ddepth, cn));
    const size_t n = total() * cn, s1 = elemSize1(), d1 = out.elemSize1();
    for (size_t i = 0; i < n; ++i) store_scalar(out.data + i * d1, ddepth, load_scalar(data + i * s1, depth()) * alpha + beta);
    dst = out;
}

void remap(const Mat& src, Mat& dst, const Mat& map_x, const Mat& map_y, int interpolation)
{
    if (src.type() != CV_32FC1 || map_x.type() != CV_32FC1 || map_y.type() != CV_32FC1) throw std::runtime_error("cv::remap stand-in: CV_32FC1 only");
    Mat out(map_x.rows, map_x.cols, CV_32FC1);
    const float* mx = map_x.ptr<float>();
    const float* my = map_y.ptr<float>();
    float* o = out.ptr<float>();
    auto px = [&](int y, int x) -> float { return (x < 0 || x >= src.cols || y < 0 || y >= src.rows) ? 0.f : src.ptr<float>(y)[x]; };
    for (size_t i = 0; i < out.total(); ++i) {
        const float u = mx[i], v = my[i];
        if (interpolation == INTER_NEAREST) { o[i] = px((int)std::nearbyint(v), (int)std::nearbyint(u)); continue; }
        const float x0 = std::floor(u), y0 = std::floor(v), ax = u - x0, ay = v - y0;
        const int ix = (int)x0, iy = (int)y0;
        o[i] = (1.f - ax) * (1.f - ay) * px(iy, ix) + ax * (1.f - ay) * px(iy, ix + 1) + (1.f - ax) * ay * px(iy + 1, ix) + ax * ay * px(iy + 1, ix + 1);
    }
    dst = out;
}

// ---- PNG (non-interlaced; gray / RGB / RGBA at 8 bits, gray at 16 bits) ----------------------------------------------------------
static bool inflate_all(const std::vector<unsigned char>& in, std::vector<unsigned char>& out, size_t expect)
{
    out.resize(expect);
    uLongf n = (uLongf)expect;
    const int rc = uncompress(out.data(), &n, in.data(), (uLong)in.size());
    out.resize(n);
    return rc == Z_OK;
}

Mat imread(const std::string& filename, int flags)
{
    std::ifstream f(filename, std::ios::binary);
    if (!f) return Mat();
    std::vector<unsigned char> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) return Mat();
    auto be32 = [&](size_t p) { return ((uint32_t)file[p] << 24) | ((uint32_t)file[p + 1] << 16) | ((uint32_t)file[p + 2] << 8) | file[p + 3]; };
    uint32_t W = 0, H = 0; int bits = 0, ctype = 0, interlace = 0;
    std::vector<unsigned char> idat;
    for (size_t p = 8; p + 12 <= file.size();) {
        const uint32_t len = be32(p);
        const std::string tag((const char*)&file[p + 4], 4);
        if (p + 12 + len > file.size()) return Mat();
        if (tag == "IHDR") {
            if (len != 13) return Mat();                                                // (a shorter chunk would be read past its end)
            W = be32(p + 8); H = be32(p + 12); bits = file[p + 16]; ctype = file[p + 17]; interlace = file[p + 20];
        }
        else if (tag == "IDAT") idat.insert(idat.end(), file.begin() + p + 8, file.begin() + p + 8 + len);
        else if (tag == "IEND") break;
        p += 12 + len;
    }
    const int cn = ctype == 0 ? 1 : (ctype == 2 ? 3 : (ctype == 6 ? 4 : 0));
    if (!W || !H || !cn || interlace || !((bits == 8) || (bits == 16 && cn == 1))) return Mat();    // palette / gray+alpha / interlaced: unsupported
    if (W > (1u << 15) || H > (1u << 15)) return Mat();                                   // a hostile header must not size the allocations below
    const size_t bpp = (size_t)cn * bits / 8, stride = (size_t)W * bpp;
    std::vector<unsigned char> raw;
    if (!inflate_all(idat, raw, (stride + 1) * H) || raw.size() != (stride + 1) * H) return Mat();
    std::vector<unsigned char> img(stride * H);
    for (uint32_t y = 0; y < H; ++y) {                                                  // undo the scanline filters
        const unsigned char* in = &raw[(stride + 1) * y + 1];
        unsigned char* cur = &img[stride * y];
        const unsigned char* up = y ? &img[stride * (y - 1)] : nullptr;
        const int ft = raw[(stride + 1) * y];
        for (size_t x = 0; x < stride; ++x) {
            const int a = x >= bpp ? cur[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0;
            int pred = 0;
            switch (ft) {
            case 0: pred = 0; break;
            case 1: pred = a; break;
            case 2: pred = b; break;
            case 3: pred = (a + b) >> 1; break;
            case 4: { const int pq = a + b - c, pa = std::abs(pq - a), pb = std::abs(pq - b), pc = std::abs(pq - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
            default: return Mat();
            }
            cur[x] = (unsigned char)(in[x] + pred);
        }
    }
    Mat m;
    if (bits == 16) {
        m.create((int)H, (int)W, CV_16UC1);
        uint16_t* o = m.ptr<uint16_t>();
        for (size_t i = 0; i < (size_t)W * H; ++i) o[i] = (uint16_t)((img[2 * i] << 8) | img[2 * i + 1]);
    } else {
        m.create((int)H, (int)W, CV_MAKETYPE(CV_8U, cn));
        for (size_t i = 0; i < (size_t)W * H; ++i)                                      // OpenCV hands colour out as B,G,R(,A)
            for (int k = 0; k < cn; ++k) m.data[i * cn + k] = img[i * cn + ((cn >= 3 && k < 3) ? 2 - k : k)];
    }
    if (flags == IMREAD_COLOR && !(m.depth() == CV_8U && m.channels() == 3)) {         // 3-channel 8-bit BGR, as OpenCV's default mode gives
        Mat c3((int)H, (int)W, CV_8UC3);
        for (size_t i = 0; i < (size_t)W * H; ++i)
            for (int k = 0; k < 3; ++k) {
                double v = load_scalar(m.data + (i * m.channels() + (m.channels() >= 3 ? k : 0)) * m.elemSize1(), m.depth());
                c3.data[i * 3 + k] = (unsigned char)(m.depth() == CV_16U ? v / 257.0 : v);
            }
        m = c3;
    }
    return m;
}

}  // namespace cv

// ---------------------------------------------------------------------------------------------------------
// OpenEXR (single-part scanline files; NONE / RLE / ZIPS / ZIP)
// ---------------------------------------------------------------------------------------------------------
static const char* exr_err(const char** err, const char* msg) { if (err) { char* m = (char*)std::malloc(std::strlen(msg) + 1); std::strcpy(m, msg); *err = m; } return msg; }
void FreeEXRErrorMessage(const char* msg) { std::free((void*)msg); }

static float half_to_float(uint16_t h)
{
    const uint32_t s = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31, m = h & 1023;
    uint32_t bits;
    if (e == 0) {
        if (m == 0) bits = s;
        else { int sh = 0; uint32_t mm = m; while (!(mm & 1024)) { mm <<= 1; ++sh; } bits = s | ((uint32_t)(113 - sh) << 23) | ((mm & 1023) << 13); }
    } else if (e == 31) bits = s | 0x7f800000u | (m << 13);
    else bits = s | ((e + 112) << 23) | (m << 13);
    float f; std::memcpy(&f, &bits, 4); return f;
}

int LoadEXR(float** out_rgba, int* width, int* height, const char* filename, const char** err)
{
    if (!out_rgba || !width || !height || !filename) { exr_err(err, "LoadEXR: null argument"); return TINYEXR_ERROR_INVALID_DATA; }
    std::ifstream f(filename, std::ios::binary);
    if (!f) { exr_err(err, "LoadEXR: cannot open file"); return TINYEXR_ERROR_CANT_OPEN_FILE; }
    std::vector<unsigned char> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    size_t p = 0;
    auto need = [&](size_t n) { return p + n <= d.size(); };
    auto i32 = [&](size_t q) { int32_t v; std::memcpy(&v, &d[q], 4); return v; };
    if (d.size() < 8 || d[0] != 0x76 || d[1] != 0x2f || d[2] != 0x31 || d[3] != 0x01) { exr_err(err, "LoadEXR: not an OpenEXR file"); return TINYEXR_ERROR_INVALID_DATA; }
    const uint32_t flags = (uint32_t)d[5] | ((uint32_t)d[6] << 8);       // bytes 5..7 of the version field
    if (flags & (0x02 | 0x08 | 0x10)) { exr_err(err, "LoadEXR: tiled, deep and multi-part files are not supported"); return TINYEXR_ERROR_UNSUPPORTED_FORMAT; }
    p = 8;
    struct Chan { std::string name; int type; };
    std::vector<Chan> chans;
    int comp = -1, xmin = 0, ymin = 0, xmax = -1, ymax = -1;
    for (;;) {
        if (!need(1)) { exr_err(err, "LoadEXR: truncated header"); return TINYEXR_ERROR_INVALID_DATA; }
        if (d[p] == 0) { ++p; break; }
        // every header string must end inside the file
        auto cstr = [&](size_t at, std::string& out_s) { const void* z = at < d.size() ? std::memchr(&d[at], 0, d.size() - at) : nullptr; if (!z) return false; out_s.assign((const char*)&d[at]); return true; };
        std::string name, type;
        if (!cstr(p, name)) { exr_err(err, "LoadEXR: unterminated attribute name"); return TINYEXR_ERROR_INVALID_DATA; }
        p += name.size() + 1;
        if (!cstr(p, type)) { exr_err(err, "LoadEXR: unterminated attribute type"); return TINYEXR_ERROR_INVALID_DATA; }
        p += type.size() + 1;
        if (!need(4)) return TINYEXR_ERROR_INVALID_DATA;
        const int32_t size = i32(p); p += 4;
        if (size < 0 || !need((size_t)size)) { exr_err(err, "LoadEXR: truncated attribute"); return TINYEXR_ERROR_INVALID_DATA; }
        if (name == "channels") {
            size_t q = p;
            while (q < p + size && d[q] != 0) {
                Chan c;
                const void* z = std::memchr(&d[q], 0, p + size - q);          // the name must end inside the attribute
                if (!z) { exr_err(err, "LoadEXR: unterminated channel name"); return TINYEXR_ERROR_INVALID_DATA; }
                c.name = std::string((const char*)&d[q]); q += c.name.size() + 1;
                if (q + 16 > p + size) { exr_err(err, "LoadEXR: truncated channel list"); return TINYEXR_ERROR_INVALID_DATA; }
                c.type = i32(q); q += 16;                                   // pixel type, pLinear + 3 reserved, xSampling, ySampling
                if (c.type < 0 || c.type > 2) { exr_err(err, "LoadEXR: unknown pixel type"); return TINYEXR_ERROR_INVALID_DATA; }
                chans.push_back(c);
            }
        } else if (name == "compression") { if (size < 1) return TINYEXR_ERROR_INVALID_DATA; comp = d[p]; }
        else if (name == "dataWindow") { if (size < 16) return TINYEXR_ERROR_INVALID_DATA; xmin = i32(p); ymin = i32(p + 4); xmax = i32(p + 8); ymax = i32(p + 12); }
        p += size;
    }
    const long long Wl = (long long)xmax - xmin + 1, Hl = (long long)ymax - ymin + 1;      // (in 64 bits: the window corners are arbitrary int32)
    if (Wl <= 0 || Hl <= 0 || Wl > (1 << 15) || Hl > (1 << 15) || chans.empty() || chans.size() > 64) { exr_err(err, "LoadEXR: missing or implausible dataWindow / channels"); return TINYEXR_ERROR_INVALID_DATA; }
    const int W = (int)Wl, H = (int)Hl;
    if (comp < 0 || comp > 3) { exr_err(err, "LoadEXR: only NONE, RLE, ZIPS and ZIP compression are supported"); return TINYEXR_ERROR_UNSUPPORTED_FORMAT; }
    const int lines_per_block = comp == 3 ? 16 : 1;
    const int nblocks = (H + lines_per_block - 1) / lines_per_block;
    if (!need((size_t)nblocks * 8)) return TINYEXR_ERROR_INVALID_DATA;
    std::vector<uint64_t> offs(nblocks);
    std::memcpy(offs.data(), &d[p], (size_t)nblocks * 8);
    size_t row_bytes = 0;
    std::vector<size_t> ch_off(chans.size());
    for (size_t c = 0; c < chans.size(); ++c) { ch_off[c] = row_bytes; row_bytes += (size_t)W * (chans[c].type == 1 ? 2 : 4); }
    // destination slot of each channel
    std::vector<int> slot(chans.size(), -1);
    for (size_t c = 0; c < chans.size(); ++c) {
        const std::string& n = chans[c].name;
        slot[c] = n == "R" ? 0 : n == "G" ? 1 : n == "B" ? 2 : n == "A" ? 3 : -1;
    }
    const bool single = chans.size() == 1 || std::count(slot.begin(), slot.end(), -1) == (long)chans.size();
    float* out = (float*)std::malloc((size_t)W * H * 4 * sizeof(float));
    if (!out) { exr_err(err, "LoadEXR: out of memory"); return TINYEXR_ERROR_INVALID_DATA; }
    for (size_t i = 0; i < (size_t)W * H; ++i) { out[4 * i] = out[4 * i + 1] = out[4 * i + 2] = 0.f; out[4 * i + 3] = 1.f; }
    std::vector<unsigned char> buf, tmp;
    for (int b = 0; b < nblocks; ++b) {
        size_t q = (size_t)offs[b];
        if (q + 8 > d.size()) { std::free(out); exr_err(err, "LoadEXR: bad block offset"); return TINYEXR_ERROR_INVALID_DATA; }
        const long long y0l = (long long)i32(q) - ymin; const int32_t dsize = i32(q + 4); q += 8;      // (64 bits: the block's line number is arbitrary)
        if (y0l < 0 || y0l >= H) { std::free(out); exr_err(err, "LoadEXR: bad block"); return TINYEXR_ERROR_INVALID_DATA; }
        const int y0 = (int)y0l;
        const int nl = std::min(lines_per_block, H - y0);
        const size_t expect = row_bytes * nl;
        if (dsize < 0 || q + (size_t)dsize > d.size() || nl <= 0) { std::free(out); exr_err(err, "LoadEXR: bad block"); return TINYEXR_ERROR_INVALID_DATA; }
        if (comp == 0 || (size_t)dsize == expect) buf.assign(d.begin() + q, d.begin() + q + dsize);      // stored raw
        else {
            tmp.resize(expect);
            if (comp == 1) {                                                                               // RLE
                size_t o = 0, i = 0;
                while (i < (size_t)dsize && o < expect) {
                    const int n = (signed char)d[q + i++];
                    if (n < 0) { const size_t k = (size_t)(-n); if (i + k > (size_t)dsize || o + k > expect) break; std::memcpy(&tmp[o], &d[q + i], k); o += k; i += k; }
                    else { const size_t k = (size_t)n + 1; if (i >= (size_t)dsize || o + k > expect) break; std::memset(&tmp[o], d[q + i++], k); o += k; }
                }
                if (o != expect) { std::free(out); exr_err(err, "LoadEXR: RLE block size mismatch"); return TINYEXR_ERROR_INVALID_DATA; }
            } else {
                uLongf n = (uLongf)expect;
                if (uncompress(tmp.data(), &n, &d[q], (uLong)dsize) != Z_OK || n != expect) { std::free(out); exr_err(err, "LoadEXR: zlib block failed"); return TINYEXR_ERROR_INVALID_DATA; }
            }
            for (size_t i = 1; i < expect; ++i) tmp[i] = (unsigned char)(tmp[i - 1] + tmp[i] - 128);         // predictor
            buf.resize(expect);                                                                                // de-interleave the two halves
            const size_t half = (expect + 1) / 2;
            for (size_t i = 0; i < expect; ++i) buf[i] = (i & 1) ? tmp[half + i / 2] : tmp[i / 2];
        }
        if (buf.size() != expect) { std::free(out); exr_err(err, "LoadEXR: block size mismatch"); return TINYEXR_ERROR_INVALID_DATA; }
        for (int l = 0; l < nl; ++l)
            for (size_t c = 0; c < chans.size(); ++c) {
                const unsigned char* row = &buf[row_bytes * l + ch_off[c]];
                for (int x = 0; x < W; ++x) {
                    float v;
                    if (chans[c].type == 1) { uint16_t h; std::memcpy(&h, row + 2 * x, 2); v = half_to_float(h); }
                    else if (chans[c].type == 2) std::memcpy(&v, row + 4 * x, 4);
                    else { uint32_t u; std::memcpy(&u, row + 4 * x, 4); v = (float)u; }
                    float* px = out + ((size_t)(y0 + l) * W + x) * 4;
                    if (single) { if (c == 0) px[0] = px[1] = px[2] = v; }
                    else if (slot[c] >= 0) px[slot[c]] = v;
                }
            }
    }
    *out_rgba = out; *width = W; *height = H;
    return TINYEXR_SUCCESS;
}

// ---------------------------------------------------------------------------------------------------------
// CoFusionReader (reference src/inputs/CoFusionReader.cpp:7-70)
// ---------------------------------------------------------------------------------------------------------
CoFusionReader::CoFusionReader(std::string inp)
{
    fptr = 3;                                   // :9 (the reference starts at frame 3)
    input_folder = inp;
    width = 640; height = 480;
    png_depth_scale = 6553.5f;
    c2w = Eigen::Matrix4f::Identity();          // :14 no poses in the dataset
    n_imgs = 849;
}
CoFusionReader::~CoFusionReader() {}
bool CoFusionReader::hasMore() { return fptr <= n_imgs; }      // :23-29
int CoFusionReader::getIdx() { return fptr; }
void CoFusionReader::getBack() {}
void CoFusionReader::reset() { fptr = 1; }

void CoFusionReader::getNext()                                  // :36-60
{
    if (!hasMore()) { std::cout << fptr << "! fptr size exceeded." << std::endl; return; }
    char num[8];
    std::snprintf(num, sizeof(num), "%d%d%d", fptr / 100, fptr / 10 % 10, fptr % 100 % 10);      // :40-41 (as written: three digits after a literal 0)
    const std::string rgb_f = input_folder + "colour/Color0" + num + ".png";
    const std::string depth_f = input_folder + "depth_noise/Depth0" + num + ".exr";
    const char* err = nullptr;
    float* out = nullptr;                        // width * height * RGBA
    if (LoadEXR(&out, &width, &height, depth_f.c_str(), &err) != TINYEXR_SUCCESS) {
        std::string msg = std::string("CoFusionReader: ") + (err ? err : "LoadEXR failed") + " (" + depth_f + ")";
        if (err) FreeEXRErrorMessage(err);
        throw std::runtime_error(msg);
    }
    depth = cv::Mat(height, width, CV_32FC1);    // channel 0 of the RGBA result (the reference wraps the RGBA buffer as 1-channel, D27)
    for (int i = 0; i < width * height; ++i) depth.ptr<float>()[i] = out[4 * i];
    std::free(out);
    rgb = cv::imread(rgb_f, cv::IMREAD_UNCHANGED);
    if (rgb.empty()) throw std::runtime_error("CoFusionReader: cannot read " + rgb_f);
    rgb.convertTo(rgb, CV_32FC3, 1.0 / 255.0);   // :49
    depth.convertTo(depth, CV_32FC1);            // :51
    fptr++;
}

// ---------------------------------------------------------------------------------------------------------
// SequenceReader: Replica / ScanNet / TUM-RGBD layouts (inputs/SequenceReader.h)
// ---------------------------------------------------------------------------------------------------------
#include "inputs/SequenceReader.h"
#include <algorithm>
#include <cmath>
#include <fstream>
#include <sstream>

namespace {
bool file_exists(const std::string& p) { std::ifstream f(p, std::ios::binary); return f.good(); }

std::string with_ext(const std::string& stem)
{
    for (const char* e : {".png", ".jpg", ".jpeg"}) if (file_exists(stem + e)) return stem + e;
    return std::string();
}

Eigen::Matrix4f to_opengl(Eigen::Matrix4f m)      // camera y and z axes negated (upstream NICE-SLAM datasets: c2w[:3, 1] *= -1, c2w[:3, 2] *= -1)
{
    for (int i = 0; i < 3; ++i) { m(i, 1) = -m(i, 1); m(i, 2) = -m(i, 2); }
    return m;
}

Eigen::Matrix4f read_4x4(std::istream& in, const std::string& what)
{
    Eigen::Matrix4f m;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { double v; if (!(in >> v)) throw std::runtime_error("SequenceReader: " + what + ": 16 numbers expected"); m(i, j) = (float)v; }
    return m;
}

struct Stamped { double t; std::string rest; };
std::vector<Stamped> read_stamped(const std::string& path)
{
    std::ifstream f(path);
    if (!f.good()) throw std::runtime_error("SequenceReader: cannot read " + path);
    std::vector<Stamped> v;
    std::string line;
    while (std::getline(f, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ss(line);
        Stamped s;
        if (!(ss >> s.t)) continue;
        std::getline(ss, s.rest);
        size_t a = s.rest.find_first_not_of(" \t"); s.rest = a == std::string::npos ? std::string() : s.rest.substr(a);
        while (!s.rest.empty() && (s.rest.back() == '\r' || s.rest.back() == ' ')) s.rest.pop_back();
        v.push_back(s);
    }
    return v;
}

int nearest(const std::vector<Stamped>& v, double t)
{
    int best = -1; double bd = 1e300;
    for (size_t i = 0; i < v.size(); ++i) { const double d = std::fabs(v[i].t - t); if (d < bd) { bd = d; best = (int)i; } }
    return best;
}

Eigen::Matrix4f pose_from_tq(const std::string& rest)      // "tx ty tz qx qy qz qw" -> camera-to-world
{
    std::istringstream ss(rest);
    double t[3], q[4];
    if (!(ss >> t[0] >> t[1] >> t[2] >> q[0] >> q[1] >> q[2] >> q[3])) throw std::runtime_error("SequenceReader: groundtruth.txt: 7 numbers expected after the timestamp");
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double x = q[0] / n, y = q[1] / n, z = q[2] / n, w = q[3] / n;
    Eigen::Matrix4f m = Eigen::Matrix4f::Identity();
    m(0, 0) = (float)(1 - 2 * (y * y + z * z)); m(0, 1) = (float)(2 * (x * y - z * w)); m(0, 2) = (float)(2 * (x * z + y * w));
    m(1, 0) = (float)(2 * (x * y + z * w)); m(1, 1) = (float)(1 - 2 * (x * x + z * z)); m(1, 2) = (float)(2 * (y * z - x * w));
    m(2, 0) = (float)(2 * (x * z - y * w)); m(2, 1) = (float)(2 * (y * z + x * w)); m(2, 2) = (float)(1 - 2 * (x * x + y * y));
    for (int i = 0; i < 3; ++i) m(i, 3) = (float)t[i];
    return m;
}
}  // namespace

SequenceReader::SequenceReader(Kind k, std::string inp, float tum_frame_rate) : kind(k), input_folder(inp), width(0), height(0), fptr(0)
{
    if (!input_folder.empty() && input_folder.back() != '/') input_folder += '/';
    c2w = Eigen::Matrix4f::Identity();
    if (kind == Replica) {
        png_depth_scale = 6553.5f;
        std::ifstream tr(input_folder + "traj.txt");
        if (!tr.good()) throw std::runtime_error("SequenceReader: cannot read " + input_folder + "traj.txt");
        for (int i = 0;; ++i) {
            char num[16]; std::snprintf(num, sizeof(num), "%06d", i);
            const std::string d = input_folder + "results/depth" + num + ".png";
            if (!file_exists(d)) break;
            const std::string cfile = with_ext(input_folder + "results/frame" + num);
            if (cfile.empty()) throw std::runtime_error("SequenceReader: no colour image for " + d);
            depth_files.push_back(d); color_files.push_back(cfile);
            poses.push_back(to_opengl(read_4x4(tr, "traj.txt")));
        }
    } else if (kind == ScanNet) {
        png_depth_scale = 1000.f;
        for (int i = 0;; ++i) {
            const std::string d = input_folder + "depth/" + std::to_string(i) + ".png";
            if (!file_exists(d)) break;
            const std::string cfile = with_ext(input_folder + "color/" + std::to_string(i));
            if (cfile.empty()) throw std::runtime_error("SequenceReader: no colour image for " + d);
            std::ifstream pf(input_folder + "pose/" + std::to_string(i) + ".txt");
            if (!pf.good()) throw std::runtime_error("SequenceReader: no pose file for " + d);
            depth_files.push_back(d); color_files.push_back(cfile);
            poses.push_back(to_opengl(read_4x4(pf, "pose/" + std::to_string(i) + ".txt")));
        }
    } else {
        png_depth_scale = 5000.f;
        const std::vector<Stamped> rgbs = read_stamped(input_folder + "rgb.txt"), deps = read_stamped(input_folder + "depth.txt"),
                                   gts = read_stamped(input_folder + "groundtruth.txt");
        const double max_dt = 0.08;
        double last_t = -1e300;
        bool have_first = false;
        Eigen::Matrix4f inv0 = Eigen::Matrix4f::Identity();
        for (const Stamped& r : rgbs) {
            const int j = nearest(deps, r.t), k = nearest(gts, r.t);
            if (j < 0 || k < 0 || std::fabs(deps[j].t - r.t) >= max_dt || std::fabs(gts[k].t - r.t) >= max_dt) continue;
            if (have_first && r.t - last_t <= 1.0 / tum_frame_rate) continue;          // thinned to frame_rate Hz
            Eigen::Matrix4f p = pose_from_tq(gts[k].rest);
            if (!have_first) { inv0 = p.inverse(); have_first = true; }
            last_t = r.t;
            color_files.push_back(input_folder + r.rest); depth_files.push_back(input_folder + deps[j].rest);
            poses.push_back(to_opengl(inv0 * p));
        }
    }
    n_imgs = (int)depth_files.size();
    if (n_imgs == 0) throw std::runtime_error("SequenceReader: no frames found under " + input_folder);
}
SequenceReader::~SequenceReader() {}
bool SequenceReader::hasMore() { return fptr < n_imgs; }
int SequenceReader::getIdx() { return fptr; }
void SequenceReader::getBack() { if (fptr > 0) --fptr; }
void SequenceReader::reset() { fptr = 0; }

void SequenceReader::getNext()
{
    if (!hasMore()) { std::cout << fptr << "! fptr size exceeded." << std::endl; return; }
    const std::string& cf = color_files[fptr];
    rgb = cv::imread(cf, cv::IMREAD_COLOR);
    if (rgb.empty()) {
        const bool jpg = cf.size() > 4 && (cf.substr(cf.size() - 4) == ".jpg" || cf.substr(cf.size() - 5) == ".jpeg");
        throw std::runtime_error("SequenceReader: cannot read " + cf + (jpg ? " (the cv::imread stand-in decodes PNG only: build against OpenCV or convert the colour images to PNG)" : ""));
    }
    cv::Mat d16 = cv::imread(depth_files[fptr], cv::IMREAD_UNCHANGED);
    if (d16.empty() || d16.channels() != 1) throw std::runtime_error("SequenceReader: cannot read " + depth_files[fptr] + " as a one-channel depth image");
    rgb.convertTo(rgb, CV_32FC3, 1.0 / 255.0);
    d16.convertTo(depth, CV_32FC1, 1.0 / png_depth_scale);
    width = depth.cols; height = depth.rows;
    c2w = poses[fptr];
    fptr++;
}
