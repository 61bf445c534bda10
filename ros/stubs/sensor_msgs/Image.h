#ifndef SENSOR_MSGS_IMAGE_STUB_H
#define SENSOR_MSGS_IMAGE_STUB_H
#include <std_msgs/Header.h>
#include <vector>
namespace sensor_msgs {
struct Image {  // sensor_msgs/Image.msg
    std_msgs::Header header;
    uint32_t height = 0, width = 0;
    std::string encoding;
    uint8_t is_bigendian = 0;
    uint32_t step = 0;
    std::vector<uint8_t> data;
    typedef std::shared_ptr<const Image> ConstPtr;
};
}  // namespace sensor_msgs
#endif
