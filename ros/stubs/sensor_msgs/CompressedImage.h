#ifndef SENSOR_MSGS_COMPRESSEDIMAGE_STUB_H
#define SENSOR_MSGS_COMPRESSEDIMAGE_STUB_H
#include <std_msgs/Header.h>
#include <vector>
namespace sensor_msgs {
struct CompressedImage {  // sensor_msgs/CompressedImage.msg
    std_msgs::Header header;
    std::string format;
    std::vector<uint8_t> data;
    typedef std::shared_ptr<const CompressedImage> ConstPtr;
};
}  // namespace sensor_msgs
#endif
