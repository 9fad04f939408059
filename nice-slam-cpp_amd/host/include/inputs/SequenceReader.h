// Readers for the posed RGB-D sequences that BASELINE.json's configs K2-K5 presuppose and the reference lacks:
// same surface as include/inputs/CoFusionReader.h (hasMore / getNext / getIdx / reset; rgb, depth, c2w), so Tracker::run and a Mapper
// loop take them the way src/main.cpp takes the CoFusion reader.  File layouts are those of the upstream NICE-SLAM datasets:
//   Replica  <dir>/results/frame%06d.png|jpg, <dir>/results/depth%06d.png (16 bit, / 6553.5), <dir>/traj.txt (one row-major 4x4 per line)
//   ScanNet  <dir>/color/%d.png|jpg, <dir>/depth/%d.png (16 bit, / 1000), <dir>/pose/%d.txt (4 lines of 4)
//   TUM      <dir>/rgb.txt, depth.txt, groundtruth.txt (timestamp file | timestamp tx ty tz qx qy qz qw), depth / 5000; frames are the
//            rgb stamps with a depth image and a pose within 0.08 s, thinned to `frame_rate` Hz; poses are taken relative to the first
//            frame (c2w_i = inv(c2w_0) c2w_i)
// Every pose is turned into the renderer's OpenGL camera (y and z axes of the camera negated: columns 1 and 2 of c2w), depth comes back in
// metres as CV_32FC1, colour as CV_32FC3 in [0, 1] in cv::imread's B,G,R order (what CoFusionReader hands out).
// Colour files are read with cv::imread: the stand-in decodes PNG only and says so for a JPEG; with the real OpenCV both work.
#ifndef SEQUENCEREADER_H_
#define SEQUENCEREADER_H_

#include <string>
#include <vector>
#include <opencv2/imgproc/imgproc.hpp>
#include <opencv2/highgui/highgui.hpp>
#include <Eigen/Core>

class SequenceReader {
  public:
    enum Kind { Replica, ScanNet, TUM };
    SequenceReader(Kind kind, std::string input_folder, float tum_frame_rate = 32.f);
    virtual ~SequenceReader();

    void getNext();
    void getBack();
    bool hasMore();
    void reset();
    int getIdx();

    Kind kind;
    std::string input_folder;
    cv::Mat depth, rgb;
    Eigen::Matrix4f c2w;                 // pose of the frame getNext() loaded

    int width, height;
    float png_depth_scale;
    int n_imgs;
    std::vector<Eigen::Matrix4f> poses;  // all frames' poses (OpenGL camera), known after construction

  private:
    int fptr;
    std::vector<std::string> color_files, depth_files;
};
#endif /* SEQUENCEREADER_H_ */
