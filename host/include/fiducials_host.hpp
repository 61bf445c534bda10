// fiducials_host.hpp -- the host side of the aruco_detect drop-in in the reference's own language: a ROS-free FiducialsNode
// with the callbacks, members and parameter names of aruco_detect/src/aruco_detect.cpp:88-147, on top of the C-ABI
// (include/fid_abi.h).  No ROS, catkin or OpenCV exists in this image, so the message types are plain structs with the fields
// of the .msg files and "publishing" is returning the message; everything else -- gating, latching of the first CameraInfo,
// ignore list, per-id lengths, header rules, quaternion -- is the node's logic, so that the node proper is this class plus
// ros::Subscriber / Publisher glue.  The arithmetic of the hot path is NOT here: it is behind fid_detect / fid_pose_last.
#ifndef FIDUCIALS_HOST_HPP
#define FIDUCIALS_HOST_HPP
#include <array>
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "fid_abi.h"

namespace fiducials_amd {

struct Header {  // std_msgs/Header
    uint32_t seq = 0;
    uint32_t sec = 0, nsec = 0;
    std::string frame_id;
};
struct Image {  // sensor_msgs/Image
    Header header;
    uint32_t height = 0, width = 0;
    std::string encoding;  // "mono8", "bgr8", "rgb8", "bgra8", "rgba8"; "mono16", "bgr16", "rgb16", "bgra16", "rgba16", "bayer_{rggb,bggr,gbrg,grbg}8"
    uint8_t is_bigendian = 0;
    uint32_t step = 0;
    std::vector<uint8_t> data;
};
struct CompressedImage {  // sensor_msgs/CompressedImage: what arrives with image_transport's `compressed` (the launch default)
    Header header;
    std::string format;  // "jpeg", "bgr8; jpeg compressed bgr8" or "mono8; png compressed " as compressed_image_transport writes it
                         // (the data decides: JPEG goes to the device decoder, PNG through fid_png_decode on the host)
    std::vector<uint8_t> data;
};
struct CameraInfo {  // sensor_msgs/CameraInfo (the fields the node reads)
    Header header;
    uint32_t height = 0, width = 0;
    std::string distortion_model;
    std::vector<double> D;
    std::array<double, 9> K{};
};
struct Fiducial {  // fiducial_msgs/Fiducial
    int32_t fiducial_id = 0, direction = 0;
    double x0 = 0, y0 = 0, x1 = 0, y1 = 0, x2 = 0, y2 = 0, x3 = 0, y3 = 0;
};
struct FiducialArray {
    Header header;
    int32_t image_seq = 0;
    std::vector<Fiducial> fiducials;
};
struct FiducialTransform {  // fiducial_msgs/FiducialTransform
    int32_t fiducial_id = 0;
    double tx = 0, ty = 0, tz = 0;          // geometry_msgs/Transform.translation
    double qx = 0, qy = 0, qz = 0, qw = 1;  // geometry_msgs/Transform.rotation
    double image_error = 0, object_error = 0, fiducial_area = 0;
};
struct FiducialTransformArray {
    Header header;
    int32_t image_seq = 0;
    std::vector<FiducialTransform> transforms;
};

// ---- the other output surface of poseEstimateCallback (aruco_detect.cpp:462-478, 501-524): vision_msgs and tf2
struct Pose {  // geometry_msgs/Pose
    double px = 0, py = 0, pz = 0;
    double ox = 0, oy = 0, oz = 0, ow = 1;
};
struct ObjectHypothesisWithPose {  // vision_msgs/ObjectHypothesisWithPose (Noetic: int64 id, float64 score, PoseWithCovariance)
    int64_t id = 0;
    double score = 0;
    Pose pose;
    std::array<double, 36> covariance{};
};
struct Detection2D {  // vision_msgs/Detection2D: the node fills `results` only (header, bbox and source_img stay default)
    Header header;
    std::vector<ObjectHypothesisWithPose> results;
};
struct Detection2DArray {
    Header header;
    std::vector<Detection2D> detections;
};
struct TransformStamped {  // geometry_msgs/TransformStamped, what tf2_ros::TransformBroadcaster::sendTransform is handed
    Header header;
    std::string child_frame_id;
    double tx = 0, ty = 0, tz = 0;
    double qx = 0, qy = 0, qz = 0, qw = 1;
};
// everything one poseEstimateCallback publishes: `~vis_msgs` selects which of the two arrays goes to fiducial_transforms
// (:666-669, 534-537), `~publish_fiducial_tf` whether `tf` is broadcast (:501-524)
struct PoseOutputs {
    bool vis_msgs = false;
    FiducialTransformArray fta;
    Detection2DArray vma;
    std::vector<TransformStamped> tf;
};

// ROS 1 wire format of the two output messages (little-endian, packed)
std::vector<uint8_t> serialize(const FiducialArray &m);
std::vector<uint8_t> serialize(const FiducialTransformArray &m);
bool deserialize(const std::vector<uint8_t> &b, FiducialTransformArray *m);

// aruco::getPredefinedDictionary(dicno) (aruco_detect.cpp:671): tables from <data_dir>/dict_*.txt (provenance:
// tools/make_dictionaries.py); the returned object owns the byte list fid_dict points into
struct Dictionary {
    int markerSize = 0, maxCorrectionBits = 0, nMarkers = 0;
    std::vector<uint8_t> bytesList;  // nMarkers x 4 rotations x nbytes
    fid_dict view() const;
};
Dictionary getPredefinedDictionary(int dicno, const std::string &data_dir);
// ... from a table file the deployer has (fid_dict_load_file: OpenCV's predefined_dictionaries.hpp as text, a FileStorage
// YAML, or a dict_*.txt): every enum value 0..16, or -1 for a custom dictionary whose sizes the YAML states
Dictionary loadDictionaryFile(int dicno, const std::string &table_file);

class FiducialsNode {
   public:
    struct Params {  // the private parameters of the node with their defaults (aruco_detect.cpp:609-627) ...
        bool publish_images = false;
        double fiducial_len = 0.14;
        int dictionary = 7;
        bool do_pose_estimation = true;
        bool publish_fiducial_tf = true;
        bool vis_msgs = false;
        bool verbose = false;
        std::string ignore_fiducials;
        std::string fiducial_len_override;
        fid_params detector;  // ... and the detector parameters as the node sets them (:690-727)
        std::string data_dir = "fiducials_amd/data";
        std::string dictionary_file;  // non-empty: the table of ~dictionary comes from this file (loadDictionaryFile) -- the way to
                                      // run the families whose shipped tables are fillers (4X4_1000, 6X6, 7X7, ARUCO_ORIGINAL)
        int device = 0, max_width = 1920, max_height = 1080;
        Params();
    };
    explicit FiducialsNode(const Params &p);  // throws std::runtime_error without a device: there is no CPU path
    ~FiducialsNode();
    FiducialsNode(const FiducialsNode &) = delete;
    FiducialsNode &operator=(const FiducialsNode &) = delete;

    // the callbacks (same names as the reference).  The two image-path callbacks return true when the node would have
    // published, the message is then in *out
    void configCallback(const fid_params &config, uint32_t level);  // :257-298
    void ignoreCallback(const std::string &msg);                     // :300-305
    void camInfoCallback(const CameraInfo &msg);                     // :307-330
    bool imageCallback(const Image &msg, FiducialArray *out);        // :332-395
    // ... and what image_pub publishes on /fiducial_images when ~publish_images is set (:381-387): the BGR8 copy of the frame
    // with aruco::drawDetectedMarkers' marker outlines on it (fid_draw_detected_markers: the exactly restated part of the
    // overlay; include/fid_abi.h lists what is not drawn).  *image is filled only when publish_images is set (returns the same
    // as the two-argument form).
    bool imageCallback(const Image &msg, FiducialArray *out, Image *image);
    // the same callback for a frame that arrives compressed (aruco_detect.launch:6 transport=compressed): the JPEG is decoded
    // on the device (fid_jpeg_decode: what the subscriber plugin's cv::imdecode + toCvCopy(BGR8) + BGR2GRAY produce) and the
    // detector runs on the device-resident gray image
    bool compressedImageCallback(const CompressedImage &msg, FiducialArray *out);
    bool poseEstimateCallback(const FiducialArray &msg, FiducialTransformArray *out);  // :397-538 (fiducial_msgs view)
    bool poseEstimateCallback(const FiducialArray &msg, PoseOutputs *out);             // ... everything it publishes
    bool enableDetectionsCallback(bool data, std::string *message);  // :573-588

    // state the reference keeps as members (read-only views for tests)
    const std::vector<int> &getIds() const { return ids; }
    const std::vector<int> &getIgnoreIds() const { return ignoreIds; }
    const std::map<int, double> &getFiducialLens() const { return fiducialLens; }
    bool haveCameraInfo() const { return haveCamInfo; }
    const std::string &lastError() const { return last_error; }

   private:
    void handleIgnoreString(const std::string &str);           // :540-571
    void handleLenOverrideString(const std::string &str);      // :627-660
    bool publishVertices(const Header &h, int32_t n, FiducialArray *out);  // the tail of imageCallback (:342-379)
    fid_ctx *ctx = nullptr;
    fid_jpeg_ctx *jctx = nullptr;  // made when the first compressed frame arrives
    std::vector<uint8_t> png_frame;  // a PNG frame decoded on the host (fid_png_decode), reused from frame to frame
    std::vector<uint8_t> converted;  // (round 5: the BGR8 copy of a 16-bit / Bayer frame made before the detection; unused since ABI 7)
    bool raw_encoding = false;       // the last frame came in a raw-camera encoding (Bayer / 16 bit / UYVY): detected on the message bytes
    int maxW = 0, maxH = 0, dev = 0;
    Dictionary dict;
    fid_params detectorParams;
    std::vector<fid_marker> markers;   // corners / ids of the last image (the reference's `corners`, `ids` members, :101-102)
    std::vector<int> ids;
    std::vector<int> ignoreIds;
    std::map<int, double> fiducialLens;
    double cameraMatrix[9] = {0}, distortionCoeffs[5] = {0};
    bool haveCamInfo = false, enable_detections = true, doPoseEstimation = true, verbose = false;
    bool vis_msgs = false, publishFiducialTf = true, publish_images = false;
    double fiducial_len = 0.14;
    int frameNum = 0;
    std::string frameId, last_error;
};

}  // namespace fiducials_amd
#endif
