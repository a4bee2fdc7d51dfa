// MOCK (see ../core.hpp): cv::line_descriptor::KeyLine with the real field order (68 bytes)
#pragma once
#include "../core.hpp"
namespace cv { namespace line_descriptor {
struct KeyLine {
    float angle; int class_id; int octave; Point2f pt; float response; float size;
    float startPointX, startPointY, endPointX, endPointY, sPointInOctaveX, sPointInOctaveY, ePointInOctaveX, ePointInOctaveY, lineLength;
    int numOfPixels;
};
} }
