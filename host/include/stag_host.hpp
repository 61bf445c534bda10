// stag_host.hpp -- the host side of the stag_detect drop-in in the reference's language: class Stag with the constructor and
// the two calls StagNode uses (stag_detect/include/stag/Stag.h:41-45), on top of fid_stag_* (include/fid_abi.h), and the
// message step of StagNode::imageCallback (stag_detect.cpp:110-217) producing the fiducial_msgs contract north_star asks for
// (vertices + transforms) instead of PoseStamped / Detection2DArray.  Header-only; no OpenCV: images are (pointer, cols, rows,
// step) and points are plain structs.
#ifndef STAG_HOST_HPP
#define STAG_HOST_HPP
#include <cmath>
#include <cstdint>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "fid_abi.h"
#include "fiducials_host.hpp"

namespace fiducials_amd {

struct Point2d {
    double x = 0, y = 0;
};

struct Marker {  // stag/Marker.h + Quad.h: what getMarkerList() hands out
    int id = 0;
    std::vector<Point2d> corners;  // 4, clockwise from the marker's first corner
    Point2d center;
    double H[9] = {0};
    double projectiveDistortion = 0;
};

class Stag {
   public:
    // Stag(int libraryHD = 15, int errorCorrection = 7, bool keepLogs = false); data_dir holds stag_HD<hd>.bin (the published
    // marker libraries, tools/make_stag_libraries.py).  Throws std::invalid_argument for an invalid library like Decoder does.
    Stag(int libraryHD = 15, int inErrorCorrection = 7, bool /*inKeepLogs*/ = false, const std::string &data_dir = "fiducials_amd/data",
         int max_width = 1920, int max_height = 1080, int device = 0)
    {
        if (libraryHD < 11 || libraryHD > 23 || !(libraryHD & 1))
            throw std::invalid_argument("Invalid library HD. Possible values are 11, 13, 15, 17, 19, 21, or 23");
        std::ifstream f(data_dir + "/stag_HD" + std::to_string(libraryHD) + ".bin", std::ios::binary);
        if (!f) throw std::runtime_error("marker library not found under " + data_dir);
        f.seekg(0, std::ios::end);
        words.resize((size_t)f.tellg() / 8);
        f.seekg(0);
        f.read((char *)words.data(), (std::streamsize)words.size() * 8);
        fid_status rc = fid_stag_create(libraryHD, inErrorCorrection, max_width, max_height, device, &ctx);
        if (rc == FID_OK) rc = fid_stag_load_library(ctx, words.data(), (int32_t)words.size());
        if (rc != FID_OK) {
            fid_stag_destroy(ctx);
            throw std::runtime_error(std::string("fid_stag_create: ") + fid_strerror(rc));
        }
    }
    ~Stag() { fid_stag_destroy(ctx); }
    Stag(const Stag &) = delete;
    Stag &operator=(const Stag &) = delete;

    // void detectMarkers(cv::Mat inImage): a mono8 image
    void detectMarkers(const uint8_t *data, int cols, int rows, int step)
    {
        std::vector<fid_stag_marker> m(256);
        int32_t n = 0;
        fid_status rc = fid_stag_detect_markers(ctx, data, cols, rows, step, m.data(), (int32_t)m.size(), &n);
        if (rc == FID_E_CAPACITY) {
            m.resize((size_t)n);
            rc = fid_stag_detect_markers(ctx, data, cols, rows, step, m.data(), (int32_t)m.size(), &n);
        }
        if (rc != FID_OK) throw std::runtime_error(std::string("fid_stag_detect_markers: ") + fid_strerror(rc));
        markers.clear();
        for (int i = 0; i < n; i++) {
            Marker k;
            k.id = m[i].id;
            k.corners.resize(4);
            for (int c = 0; c < 4; c++) {
                k.corners[c].x = m[i].corners[2 * c];
                k.corners[c].y = m[i].corners[2 * c + 1];
            }
            k.center.x = m[i].center[0];
            k.center.y = m[i].center[1];
            for (int j = 0; j < 9; j++) k.H[j] = m[i].H[j];
            k.projectiveDistortion = m[i].projectiveDistortion;
            markers.push_back(k);
        }
    }
    std::vector<Marker> getMarkerList() const { return markers; }

    // Common::solvePnpSingle for the markers of the last detectMarkers() (stag_detect.cpp:140-165)
    std::vector<fid_stag_pose_out> solvePnpSingle(const double K[9], const double D[5], double marker_size)
    {
        std::vector<fid_stag_pose_out> p(markers.empty() ? 1 : markers.size());
        int32_t n = 0;
        const fid_status rc = fid_stag_pose_last(ctx, K, D, marker_size, p.data(), (int32_t)p.size(), &n);
        if (rc != FID_OK) throw std::runtime_error(std::string("fid_stag_pose_last: ") + fid_strerror(rc));
        p.resize((size_t)n);
        return p;
    }

   private:
    fid_stag_ctx *ctx = nullptr;
    std::vector<uint64_t> words;
    std::vector<Marker> markers;
};

// tf::Matrix3x3::getRotation (what stag_detect.cpp:171-178 turns the pose matrix into)
inline void rotationToQuaternion(const double m[9], double q[4] /* x y z w */)
{
    const double trace = m[0] + m[4] + m[8];
    double temp[4];
    if (trace > 0.0) {
        double s = std::sqrt(trace + 1.0);
        temp[3] = s * 0.5;
        s = 0.5 / s;
        temp[0] = (m[7] - m[5]) * s;
        temp[1] = (m[2] - m[6]) * s;
        temp[2] = (m[3] - m[1]) * s;
    } else {
        const int i = m[0] < m[4] ? (m[4] < m[8] ? 2 : 1) : (m[0] < m[8] ? 2 : 0);
        const int j = (i + 1) % 3, k = (i + 2) % 3;
        double s = std::sqrt(m[3 * i + i] - m[3 * j + j] - m[3 * k + k] + 1.0);
        temp[i] = s * 0.5;
        s = 0.5 / s;
        temp[3] = (m[3 * k + j] - m[3 * j + k]) * s;
        temp[j] = (m[3 * j + i] + m[3 * i + j]) * s;
        temp[k] = (m[3 * k + i] + m[3 * i + k]) * s;
    }
    for (int a = 0; a < 4; a++) q[a] = temp[a];
}

// StagNode::imageCallback with the fiducial_msgs contract: vertices + transforms of one image
inline void stagImageCallback(Stag &stag, const Image &msg, const CameraInfo &cam, double marker_size, FiducialArray *fva, FiducialTransformArray *fta)
{
    stag.detectMarkers(msg.data.data(), (int)msg.width, (int)msg.height, (int)msg.step);
    const std::vector<Marker> markers = stag.getMarkerList();
    double D[5] = {0, 0, 0, 0, 0};
    for (size_t i = 0; i < cam.D.size() && i < 5; i++) D[i] = cam.D[i];
    const std::vector<fid_stag_pose_out> poses = stag.solvePnpSingle(cam.K.data(), D, marker_size);
    fva->header.sec = fta->header.sec = msg.header.sec;
    fva->header.nsec = fta->header.nsec = msg.header.nsec;
    fva->header.frame_id = fta->header.frame_id = cam.header.frame_id;
    fva->image_seq = fta->image_seq = (int32_t)msg.header.seq;
    fva->fiducials.clear();
    fta->transforms.clear();
    for (size_t i = 0; i < markers.size(); i++) {
        const Marker &m = markers[i];
        Fiducial f;
        f.fiducial_id = m.id;
        f.x0 = m.corners[0].x; f.y0 = m.corners[0].y; f.x1 = m.corners[1].x; f.y1 = m.corners[1].y;
        f.x2 = m.corners[2].x; f.y2 = m.corners[2].y; f.x3 = m.corners[3].x; f.y3 = m.corners[3].y;
        fva->fiducials.push_back(f);
        FiducialTransform t;
        t.fiducial_id = m.id;
        t.tx = poses[i].tvec[0]; t.ty = poses[i].tvec[1]; t.tz = poses[i].tvec[2];
        double q[4];
        rotationToQuaternion(poses[i].R, q);
        t.qx = q[0]; t.qy = q[1]; t.qz = q[2]; t.qw = q[3];
        // area as aruco_detect's calcFiducialArea (Heron on two triangles); the STag node has no error estimates
        auto dist = [](const Point2d &a, const Point2d &b) { return std::sqrt((a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y)); };
        double a1 = dist(m.corners[0], m.corners[1]), b1 = dist(m.corners[0], m.corners[3]), c1 = dist(m.corners[1], m.corners[3]);
        double a2 = dist(m.corners[1], m.corners[2]), b2 = dist(m.corners[2], m.corners[3]), c2 = c1;
        const double s1 = (a1 + b1 + c1) / 2.0, s2 = (a2 + b2 + c2) / 2.0;
        t.fiducial_area = std::sqrt(s1 * (s1 - a1) * (s1 - b1) * (s1 - c1)) + std::sqrt(s2 * (s2 - a2) * (s2 - b2) * (s2 - c2));
        fta->transforms.push_back(t);
    }
}

// ---- the reference's node itself, without ROS: StagNode of stag_detect/src/stag_ros/stag_detect.cpp with the outputs IT
// publishes -- one geometry_msgs/PoseStamped per marker on `stag_ros/markers` (header.frame_id = the marker id as text,
// common.hpp:72-82), one vision_msgs/Detection2DArray on `stag_ros/markers_array` (stag_detect.cpp:139-209; the shipped launch
// remaps it onto /fiducial_transforms, stag_detect.launch:10) and, with `publish_tf`, <image frame> -> <tag_tf_prefix><id>.
// ros/stag_detect_amd is the catkin glue around this class; host/test/stag_test.cpp runs it on the GPU box.
struct PoseStamped {  // geometry_msgs/PoseStamped
    Header header;
    Pose pose;
};

class StagNode {
   public:
    struct Params {  // StagNode::loadParameters (stag_detect.cpp:88-108) with its defaults
        int libraryHD = 15, errorCorrection = 7;
        std::string raw_image_topic = "image_raw", camera_info_topic = "camera_info";
        std::string markers_topic = "stag_ros/markers", markers_array_topic = "stag_ros/markers_array";
        bool is_compressed = false, show_markers = true, publish_tf = false;
        std::string tag_tf_prefix = "STag_";
        float marker_size = 0.18f;
    };
    struct Outputs {
        std::vector<PoseStamped> markers;  // Common::publishTransform, one message per marker, in marker order
        Detection2DArray array;
        std::vector<TransformStamped> tf;
        bool array_published = false;  // (the reference returns before markersArrayPub.publish when a pose comes back empty)
    };

    StagNode(const Params &p, const std::string &data_dir = "fiducials_amd/data", int max_width = 1920, int max_height = 1080, int device = 0)
        : params(p), stag(p.libraryHD, p.errorCorrection, false, data_dir, max_width, max_height, device)
    {
    }

    // StagNode::cameraInfoCallback (:219-263): the first message is kept, later ones are ignored
    void cameraInfoCallback(const CameraInfo &msg)
    {
        if (got_camera_info) return;
        for (int i = 0; i < 9; i++) K[i] = msg.K[(size_t)i];
        for (int i = 0; i < 5; i++) D[i] = (size_t)i < msg.D.size() ? msg.D[(size_t)i] : 0.0;
        got_camera_info = true;
    }

    // stag_ros::msgToGray (utility.hpp:8-20): bgr8 / rgb8 through cvtColor's 8-bit fixed point, mono8 as it is; anything else:
    // false (the reference goes on with an EMPTY image there; this node drops the frame)
    static bool msgToGray(const Image &msg, std::vector<uint8_t> *gray, const uint8_t **data, int *step)
    {
        if (msg.encoding == "mono8") {
            if (msg.step < msg.width || msg.data.size() < (size_t)msg.step * msg.height) return false;
            *data = msg.data.data();
            *step = (int)msg.step;
            return true;
        }
        const bool bgr = msg.encoding == "bgr8", rgb = msg.encoding == "rgb8";
        if (!bgr && !rgb) return false;
        if (msg.step < 3 * msg.width || msg.data.size() < (size_t)msg.step * msg.height) return false;
        gray->resize((size_t)msg.width * msg.height);
        for (uint32_t y = 0; y < msg.height; y++) {
            const uint8_t *s = msg.data.data() + (size_t)y * msg.step;
            uint8_t *o = gray->data() + (size_t)y * msg.width;
            for (uint32_t x = 0; x < msg.width; x++) {
                const int c0 = s[3 * x], g = s[3 * x + 1], c2 = s[3 * x + 2];
                const int b = bgr ? c0 : c2, r = bgr ? c2 : c0;
                o[x] = (uint8_t)((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14);  // imgproc color_rgb: RGB2Gray<uchar>, 14-bit coefficients
            }
        }
        *data = gray->data();
        *step = (int)msg.width;
        return true;
    }

    // StagNode::imageCallback (:110-217).  false: nothing is published (no CameraInfo yet, or an encoding msgToGray refuses).
    bool imageCallback(const Image &msg, Outputs *out)
    {
        out->markers.clear();
        out->tf.clear();
        out->array = Detection2DArray();
        out->array_published = false;
        if (!got_camera_info) return false;
        const uint8_t *data = nullptr;
        int step = 0;
        if (!msgToGray(msg, &gray_, &data, &step)) return false;
        stag.detectMarkers(data, (int)msg.width, (int)msg.height, step);
        const std::vector<Marker> markers = stag.getMarkerList();
        const std::vector<fid_stag_pose_out> poses = stag.solvePnpSingle(K, D, (double)params.marker_size);
        out->array.header = msg.header;
        for (size_t i = 0; i < markers.size(); i++) {
            double q[4];
            rotationToQuaternion(poses[i].R, q);  // tf::Matrix3x3::getRotation
            Pose pose;
            pose.px = poses[i].tvec[0]; pose.py = poses[i].tvec[1]; pose.pz = poses[i].tvec[2];
            pose.ox = q[0]; pose.oy = q[1]; pose.oz = q[2]; pose.ow = q[3];
            const std::string id = std::to_string(markers[i].id);
            if (params.publish_tf) {  // Common::publishTransform: tf first, then the PoseStamped
                TransformStamped t;
                t.header = msg.header;
                t.child_frame_id = params.tag_tf_prefix + id;
                t.tx = pose.px; t.ty = pose.py; t.tz = pose.pz;
                t.qx = pose.ox; t.qy = pose.oy; t.qz = pose.oz; t.qw = pose.ow;
                out->tf.push_back(t);
            }
            PoseStamped ps;
            ps.header.frame_id = id;  // (sic: the marker id, common.hpp:73)
            ps.header.sec = msg.header.sec;
            ps.header.nsec = msg.header.nsec;
            ps.pose = pose;
            out->markers.push_back(ps);
            Detection2D det;
            det.header = msg.header;
            ObjectHypothesisWithPose hyp;
            hyp.id = markers[i].id;
            hyp.pose = pose;
            det.results.push_back(hyp);
            out->array.detections.push_back(det);
        }
        out->array_published = true;
        return true;
    }

    const std::vector<Marker> lastMarkers() const { return stag.getMarkerList(); }

    Params params;
    bool got_camera_info = false;
    double K[9] = {0}, D[5] = {0};

   private:
    Stag stag;
    std::vector<uint8_t> gray_;
};

}  // namespace fiducials_amd
#endif
