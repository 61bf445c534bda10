#ifndef VISION_MSGS_DETECTION2DARRAY_STUB_H
#define VISION_MSGS_DETECTION2DARRAY_STUB_H
#include <geometry_msgs/Transform.h>
#include <sensor_msgs/Image.h>
#include <std_msgs/Header.h>
#include <array>
#include <vector>
namespace geometry_msgs {
struct PoseWithCovariance {
    Pose pose;
    std::array<double, 36> covariance{};
};
struct Pose2D {
    double x = 0, y = 0, theta = 0;
};
}  // namespace geometry_msgs
namespace vision_msgs {  // vision_msgs 0.0.x (ROS Noetic): int64 id, float64 score, PoseWithCovariance pose
struct ObjectHypothesisWithPose {
    int64_t id = 0;
    double score = 0;
    geometry_msgs::PoseWithCovariance pose;
};
struct BoundingBox2D {
    geometry_msgs::Pose2D center;
    double size_x = 0, size_y = 0;
};
struct Detection2D {
    std_msgs::Header header;
    std::vector<ObjectHypothesisWithPose> results;
    BoundingBox2D bbox;
    sensor_msgs::Image source_img;
};
struct Detection2DArray {
    std_msgs::Header header;
    std::vector<Detection2D> detections;
};
}  // namespace vision_msgs
#endif
