#ifndef GEOMETRY_MSGS_TRANSFORM_STUB_H
#define GEOMETRY_MSGS_TRANSFORM_STUB_H
namespace geometry_msgs {
struct Vector3 {
    double x = 0, y = 0, z = 0;
};
struct Point {
    double x = 0, y = 0, z = 0;
};
struct Quaternion {
    double x = 0, y = 0, z = 0, w = 0;
};
struct Transform {
    Vector3 translation;
    Quaternion rotation;
};
struct Pose {
    Point position;
    Quaternion orientation;
};
}  // namespace geometry_msgs
#endif
