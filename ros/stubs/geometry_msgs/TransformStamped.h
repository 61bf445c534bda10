#ifndef GEOMETRY_MSGS_TRANSFORMSTAMPED_STUB_H
#define GEOMETRY_MSGS_TRANSFORMSTAMPED_STUB_H
#include <geometry_msgs/Transform.h>
#include <std_msgs/Header.h>
namespace geometry_msgs {
struct TransformStamped {
    std_msgs::Header header;
    std::string child_frame_id;
    Transform transform;
};
}  // namespace geometry_msgs
#endif
