// Minimal stand-in for the subset of OpenCV the reference's drivers touch (OpenCV is not installed in this image; SURVEY.md section
// 8f N4): cv::Mat as a reference-counted row-major buffer, convertTo, Mat::eye, cv::imread for PNG files (8-bit gray / RGB / RGBA and
// 16-bit gray, non-interlaced; colour comes back in OpenCV's B,G,R order) and cv::remap with INTER_LINEAR and a constant zero border.
// The implementations live in nice-slam-cpp_amd/host/src/nsk_io.cpp.  With the real OpenCV available, drop this directory from the
// include path.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#define CV_CN_SHIFT 3
#define CV_8U 0
#define CV_16U 2
#define CV_32F 5
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn) - 1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_8UC4 CV_MAKETYPE(CV_8U, 4)
#define CV_16UC1 CV_MAKETYPE(CV_16U, 1)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_32FC3 CV_MAKETYPE(CV_32F, 3)
#define CV_32FC4 CV_MAKETYPE(CV_32F, 4)

namespace cv {

enum ImreadModes { IMREAD_UNCHANGED = -1, IMREAD_GRAYSCALE = 0, IMREAD_COLOR = 1 };
enum InterpolationFlags { INTER_NEAREST = 0, INTER_LINEAR = 1 };

class Mat {
  public:
    int rows = 0, cols = 0;
    unsigned char* data = nullptr;

    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, void* external) : rows(r), cols(c), data((unsigned char*)external), type_(type) {}   // no copy, no ownership
    void create(int r, int c, int type)
    {
        rows = r; cols = c; type_ = type;
        buf_ = std::make_shared<std::vector<unsigned char>>((size_t)r * c * elemSize(), 0);
        data = buf_->data();
    }
    static Mat eye(int r, int c, int type)
    {
        Mat m(r, c, type);
        if ((type & 7) == CV_32F) for (int i = 0; i < (r < c ? r : c); ++i) m.at<float>(i, i * m.channels()) = 1.f;
        return m;
    }
    static Mat zeros(int r, int c, int type) { return Mat(r, c, type); }
    int type() const { return type_; }
    int depth() const { return type_ & 7; }
    int channels() const { return (type_ >> CV_CN_SHIFT) + 1; }
    size_t elemSize1() const { return depth() == CV_8U ? 1 : (depth() == CV_16U ? 2 : 4); }
    size_t elemSize() const { return elemSize1() * channels(); }
    size_t total() const { return (size_t)rows * cols; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    template <typename T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + (size_t)r * cols * elemSize()); }
    template <typename T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + (size_t)r * cols * elemSize()); }
    template <typename T> T& at(int r, int c) { return ptr<T>(r)[c]; }                 // c counts scalars of type T
    template <typename T> const T& at(int r, int c) const { return ptr<T>(r)[c]; }
    Mat clone() const { Mat m(rows, cols, type_); std::memcpy(m.data, data, total() * elemSize()); return m; }
    // dst = saturate(src * alpha + beta) in the depth of rtype (channel count is kept, as in OpenCV); dst may be *this
    void convertTo(Mat& dst, int rtype, double alpha = 1.0, double beta = 0.0) const;

  private:
    int type_ = 0;
    std::shared_ptr<std::vector<unsigned char>> buf_;
};

Mat imread(const std::string& filename, int flags = IMREAD_COLOR);
// dst(i) = src(map_y(i), map_x(i)) bilinearly, samples outside the image read 0 (BORDER_CONSTANT, the default the reference relies on
// at src/Mapper.cpp:93); src: CV_32FC1, maps: CV_32FC1 of equal size
void remap(const Mat& src, Mat& dst, const Mat& map_x, const Mat& map_y, int interpolation);

}  // namespace cv
