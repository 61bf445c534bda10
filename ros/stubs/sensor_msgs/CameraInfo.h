#ifndef SENSOR_MSGS_CAMERAINFO_STUB_H
#define SENSOR_MSGS_CAMERAINFO_STUB_H
#include <std_msgs/Header.h>
#include <array>
#include <vector>
namespace sensor_msgs {
struct CameraInfo {  // sensor_msgs/CameraInfo.msg (the fields a node can read; binning / roi left out)
    std_msgs::Header header;
    uint32_t height = 0, width = 0;
    std::string distortion_model;
    std::vector<double> D;
    std::array<double, 9> K{};   // (boost::array<double, 9> in roscpp)
    std::array<double, 9> R{};
    std::array<double, 12> P{};
    typedef std::shared_ptr<const CameraInfo> ConstPtr;
};
}  // namespace sensor_msgs
#endif
