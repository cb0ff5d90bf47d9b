// image_size.h -- width/height of an image file from its header.  The reference calls cv::imread only to learn
// image.size() (data_io.cc:316-322); decoding pixels is not needed.  PNG, JPEG, BMP and TIFF, the extensions the
// reference accepts (data_io.cc:309).  Returns false for unreadable or unrecognised files (cv::imread -> empty -> skipped).
#pragma once

#include <string>

#include "types.h"

namespace ptzcalib {
bool ReadImageSize(const std::string& path, Size& size);
}
