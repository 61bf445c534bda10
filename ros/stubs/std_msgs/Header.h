#ifndef STD_MSGS_HEADER_STUB_H
#define STD_MSGS_HEADER_STUB_H
#include <ros/ros.h>
namespace std_msgs {
struct Header {  // std_msgs/Header.msg: uint32 seq, time stamp, string frame_id
    uint32_t seq = 0;
    ros::Time stamp;
    std::string frame_id;
};
}  // namespace std_msgs
#endif
