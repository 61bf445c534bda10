#ifndef TF2_ROS_TRANSFORM_BROADCASTER_STUB_H
#define TF2_ROS_TRANSFORM_BROADCASTER_STUB_H
#include <geometry_msgs/TransformStamped.h>
#include <vector>
namespace tf2_ros {
class TransformBroadcaster {
   public:
    void sendTransform(const geometry_msgs::TransformStamped &transform);
    void sendTransform(const std::vector<geometry_msgs::TransformStamped> &transforms);
};
}  // namespace tf2_ros
#endif
