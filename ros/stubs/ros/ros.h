// syntax-check stand-in for roscpp (ROS Noetic public API, the part aruco_detect_amd_node.cpp uses)
#ifndef ROS_ROS_STUB_H
#define ROS_ROS_STUB_H
#include <cstdint>
#include <cstdio>
#include <functional>
#include <memory>
#include <string>
namespace ros {
struct Time {
    uint32_t sec = 0, nsec = 0;
};
class Publisher {
   public:
    template <typename M>
    void publish(const M &) const;
};
class Subscriber {};
class ServiceServer {};
class NodeHandle {
   public:
    NodeHandle();
    explicit NodeHandle(const std::string &ns);
    template <typename T>
    bool param(const std::string &name, T &val, const T &def) const;
    template <typename M>
    Publisher advertise(const std::string &topic, uint32_t queue_size, bool latch = false);
    template <typename M, typename T>
    Subscriber subscribe(const std::string &topic, uint32_t queue_size, void (T::*fp)(const std::shared_ptr<M const> &), T *obj);
    template <typename T, typename Req, typename Res>
    ServiceServer advertiseService(const std::string &service, bool (T::*fp)(Req &, Res &), T *obj);
};
void init(int &argc, char **argv, const std::string &name);
void spin();
}  // namespace ros
#define ROS_INFO(...) ((void)std::printf(__VA_ARGS__))
#define ROS_WARN(...) ((void)std::printf(__VA_ARGS__))
#define ROS_ERROR(...) ((void)std::printf(__VA_ARGS__))
#define ROS_FATAL(...) ((void)std::printf(__VA_ARGS__))
#define ROS_ERROR_THROTTLE(period, ...) ((void)(period), (void)std::printf(__VA_ARGS__))
#endif
