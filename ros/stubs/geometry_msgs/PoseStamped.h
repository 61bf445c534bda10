#ifndef GEOMETRY_MSGS_POSESTAMPED_STUB_H
#define GEOMETRY_MSGS_POSESTAMPED_STUB_H
#include <geometry_msgs/Transform.h>
#include <std_msgs/Header.h>
namespace geometry_msgs {
struct PoseStamped {  // geometry_msgs/PoseStamped.msg
    std_msgs::Header header;
    Pose pose;
};
}  // namespace geometry_msgs
#endif
