// Same surface as the reference's include/inputs/CoFusionReader.h:17-37: CoFusion sequences (colour/ColorNNNN.png +
// depth_noise/DepthNNNN.exr, 640x480, 849 frames, no poses: c2w = identity), read through the cv::imread / LoadEXR stand-ins
// (or the real libraries when they are on the include path).
#ifndef COFUSIONREADER_H_
#define COFUSIONREADER_H_

#include <iostream>
#include <stdio.h>
#include <string>
#include <opencv2/imgproc/imgproc.hpp>
#include <opencv2/highgui/highgui.hpp>
#include <Eigen/Core>
#include "tinyexr.h"

class CoFusionReader {
  public:
    CoFusionReader(std::string input_folder);
    virtual ~CoFusionReader();

    void getNext();
    void getBack();
    bool hasMore();
    void reset();
    int getIdx();

    std::string input_folder;
    cv::Mat depth, rgb;
    Eigen::Matrix4f c2w;

    int width, height;
    float png_depth_scale;
    int n_imgs;

  private:
    int fptr;
};
#endif /* COFUSIONREADER_H_ */
