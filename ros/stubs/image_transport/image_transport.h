// syntax-check stand-in for image_transport (ROS Noetic public API, the part stag_detect_amd_node.cpp uses)
#ifndef IMAGE_TRANSPORT_STUB_H
#define IMAGE_TRANSPORT_STUB_H
#include <ros/ros.h>
#include <sensor_msgs/Image.h>
#include <string>
namespace image_transport {
class TransportHints {
   public:
    explicit TransportHints(const std::string &default_transport = "raw");
};
class Subscriber {};
class Publisher {
   public:
    void publish(const sensor_msgs::Image &message) const;
};
class ImageTransport {
   public:
    explicit ImageTransport(const ros::NodeHandle &nh);
    template <class T>
    Subscriber subscribe(const std::string &base_topic, uint32_t queue_size, void (T::*fp)(const sensor_msgs::Image::ConstPtr &), T *obj,
                         const TransportHints &transport_hints = TransportHints());
    Publisher advertise(const std::string &base_topic, uint32_t queue_size, bool latch = false);
};
}  // namespace image_transport
#endif
