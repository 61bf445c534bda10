// stag_detect_amd_node.cpp -- PARSED AND TYPE-CHECKED, NOT LINKED: no ROS exists in the authoring image, `make -C ros syntax`
// runs g++ -fsyntax-only -Wall -Wextra -Werror on this file against stand-in headers of the ROS API (ros/stubs/) and the real
// host/ and include/ headers (tests/test_host_and_abi.py runs the check).
//
// The catkin target of SURVEY.md section 8 rows f1 / f3 for the STag path: ROS glue around fiducials_amd::StagNode
// (host/include/stag_host.hpp), which carries the reference node's logic -- the CameraInfo latch, msgToGray's encodings, the
// 5-point pose, one PoseStamped per marker with the id as frame_id, the Detection2DArray, the optional TF -- and is compiled and
// run without ROS (host/test/stag_test.cpp).  Node name, private parameters, topics and transport hints are those of
// stag_detect (stag_detect/src/stag_ros/stag_detect.cpp:51-108, 265-277), so the shipped launch file's remap of
// `stag_ros/markers_array` onto /fiducial_transforms (stag_detect.launch:10) works as before.
// `~fiducial_msgs_output` (new, default false): publish fiducial_msgs/FiducialArray on `fiducial_vertices` and
// FiducialTransformArray on `fiducial_transforms` as well -- the contract fiducial_slam consumes (BASELINE north_star).
#include <fiducial_msgs/FiducialArray.h>
#include <fiducial_msgs/FiducialTransformArray.h>
#include <geometry_msgs/PoseStamped.h>
#include <geometry_msgs/TransformStamped.h>
#include <image_transport/image_transport.h>
#include <ros/ros.h>
#include <sensor_msgs/CameraInfo.h>
#include <sensor_msgs/Image.h>
#include <tf2_ros/transform_broadcaster.h>
#include <vision_msgs/Detection2DArray.h>

#include <memory>
#include <string>

#include "stag_host.hpp"

namespace fa = fiducials_amd;

namespace {

std_msgs::Header to_ros(const fa::Header &h)
{
    std_msgs::Header o;
    o.seq = h.seq;
    o.stamp.sec = h.sec;
    o.stamp.nsec = h.nsec;
    o.frame_id = h.frame_id;
    return o;
}
fa::Header to_host(const std_msgs::Header &h)
{
    fa::Header o;
    o.seq = h.seq;
    o.sec = h.stamp.sec;
    o.nsec = h.stamp.nsec;
    o.frame_id = h.frame_id;
    return o;
}
geometry_msgs::Pose to_ros(const fa::Pose &p)
{
    geometry_msgs::Pose o;
    o.position.x = p.px; o.position.y = p.py; o.position.z = p.pz;
    o.orientation.x = p.ox; o.orientation.y = p.oy; o.orientation.z = p.oz; o.orientation.w = p.ow;
    return o;
}

class RosStagNode {
   public:
    RosStagNode() : pnh_("~"), it_(nh_)
    {
        fa::StagNode::Params p;  // StagNode::loadParameters
        pnh_.param("libraryHD", p.libraryHD, 15);
        pnh_.param("errorCorrection", p.errorCorrection, 7);
        pnh_.param("raw_image_topic", p.raw_image_topic, std::string("image_raw"));
        pnh_.param("camera_info_topic", p.camera_info_topic, std::string("camera_info"));
        pnh_.param("markers_topic", p.markers_topic, std::string("stag_ros/markers"));
        pnh_.param("markers_array_topic", p.markers_array_topic, std::string("stag_ros/markers_array"));
        pnh_.param("is_compressed", p.is_compressed, false);
        pnh_.param("show_markers", p.show_markers, true);
        pnh_.param("publish_tf", p.publish_tf, false);
        pnh_.param("tag_tf_prefix", p.tag_tf_prefix, std::string("STag_"));
        pnh_.param("marker_size", p.marker_size, 0.18f);
        pnh_.param("fiducial_msgs_output", fiducial_msgs_output_, false);
        std::string data_dir;
        int max_width = 1920, max_height = 1080, device = 0;
        pnh_.param("data_dir", data_dir, std::string("fiducials_amd/data"));  // where stag_HD<hd>.bin lies
        pnh_.param("max_width", max_width, 1920);
        pnh_.param("max_height", max_height, 1080);
        pnh_.param("device", device, 0);
        node_.reset(new fa::StagNode(p, data_dir, max_width, max_height, device));  // (throws std::invalid_argument like Stag::Stag)

        image_sub_ = it_.subscribe(p.raw_image_topic, 1, &RosStagNode::imageCb, this,
                                   image_transport::TransportHints(p.is_compressed ? "compressed" : "raw"));
        caminfo_sub_ = nh_.subscribe(p.camera_info_topic, 1, &RosStagNode::camInfoCb, this);
        if (p.show_markers) debug_pub_ = it_.advertise("stag_ros/image_markers", 1);
        markers_pub_ = nh_.advertise<geometry_msgs::PoseStamped>(p.markers_topic, 10);
        array_pub_ = nh_.advertise<vision_msgs::Detection2DArray>(p.markers_array_topic, 10);
        if (fiducial_msgs_output_) {
            vertices_pub_ = nh_.advertise<fiducial_msgs::FiducialArray>("fiducial_vertices", 1);
            transforms_pub_ = nh_.advertise<fiducial_msgs::FiducialTransformArray>("fiducial_transforms", 1);
        }
    }

   private:
    void imageCb(const sensor_msgs::Image::ConstPtr &msg)
    {
        fa::Image im;
        im.header = to_host(msg->header);
        im.height = msg->height;
        im.width = msg->width;
        im.encoding = msg->encoding;
        im.is_bigendian = msg->is_bigendian;
        im.step = msg->step;
        im.data = msg->data;
        fa::StagNode::Outputs out;
        try {
            if (!node_->imageCallback(im, &out)) return;
        } catch (const std::exception &e) {
            ROS_ERROR("stag_detect_amd: %s", e.what());
            return;
        }
        if (node_->params.show_markers) {
            // the reference publishes Stag::drawMarkers() (circles, lines and id text through OpenCV's drawing tables); here: the
            // frame as bgr8 with the marker outlines as cv::line(LINE_8) draws them (fid_draw_detected_markers) -- a cue, not its pixels
            sensor_msgs::Image dbg;
            dbg.header = msg->header;
            dbg.height = msg->height;
            dbg.width = msg->width;
            dbg.encoding = "bgr8";
            dbg.step = msg->width * 3;
            dbg.data.resize((size_t)dbg.step * dbg.height);
            const fid_encoding enc = msg->encoding == "mono8" ? FID_ENC_MONO8 : (msg->encoding == "rgb8" ? FID_ENC_RGB8 : FID_ENC_BGR8);
            if (fid_to_bgr(msg->data.data(), (int32_t)msg->width, (int32_t)msg->height, (int32_t)msg->step, enc, dbg.data.data(),
                           (int64_t)dbg.data.size()) == FID_OK) {
                std::vector<fid_marker> mk;
                for (const fa::Marker &m : node_->lastMarkers()) {
                    fid_marker k;
                    k.id = m.id;
                    for (int c = 0; c < 4; c++) {
                        k.corners[2 * c] = (float)m.corners[(size_t)c].x;
                        k.corners[2 * c + 1] = (float)m.corners[(size_t)c].y;
                    }
                    mk.push_back(k);
                }
                (void)fid_draw_detected_markers(dbg.data.data(), (int32_t)dbg.width, (int32_t)dbg.height, (int32_t)dbg.step, mk.data(),
                                                (int32_t)mk.size(), 0);
                debug_pub_.publish(dbg);
            }
        }
        for (size_t i = 0; i < out.markers.size(); i++) {
            if (node_->params.publish_tf && i < out.tf.size()) {
                geometry_msgs::TransformStamped t;
                t.header = to_ros(out.tf[i].header);
                t.child_frame_id = out.tf[i].child_frame_id;
                t.transform.translation.x = out.tf[i].tx; t.transform.translation.y = out.tf[i].ty; t.transform.translation.z = out.tf[i].tz;
                t.transform.rotation.x = out.tf[i].qx; t.transform.rotation.y = out.tf[i].qy; t.transform.rotation.z = out.tf[i].qz;
                t.transform.rotation.w = out.tf[i].qw;
                broadcaster_.sendTransform(t);
            }
            geometry_msgs::PoseStamped ps;
            ps.header = to_ros(out.markers[i].header);
            ps.pose = to_ros(out.markers[i].pose);
            markers_pub_.publish(ps);
        }
        vision_msgs::Detection2DArray arr;
        arr.header = to_ros(out.array.header);
        for (const fa::Detection2D &d : out.array.detections) {
            vision_msgs::Detection2D o;
            o.header = to_ros(d.header);
            for (const fa::ObjectHypothesisWithPose &h : d.results) {
                vision_msgs::ObjectHypothesisWithPose r;
                r.id = h.id;
                r.score = h.score;
                r.pose.pose = to_ros(h.pose);
                o.results.push_back(r);
            }
            arr.detections.push_back(o);
        }
        if (out.array_published) array_pub_.publish(arr);
        if (fiducial_msgs_output_) {
            // the fiducial_msgs contract (vertices + transforms) from the same detections
            fiducial_msgs::FiducialArray fva;
            fiducial_msgs::FiducialTransformArray fta;
            fva.header = fta.header = msg->header;
            fva.image_seq = fta.image_seq = (int32_t)msg->header.seq;
            const std::vector<fa::Marker> markers = node_->lastMarkers();
            for (size_t i = 0; i < markers.size() && i < out.markers.size(); i++) {
                fiducial_msgs::Fiducial f;
                f.fiducial_id = markers[i].id;
                f.x0 = markers[i].corners[0].x; f.y0 = markers[i].corners[0].y; f.x1 = markers[i].corners[1].x; f.y1 = markers[i].corners[1].y;
                f.x2 = markers[i].corners[2].x; f.y2 = markers[i].corners[2].y; f.x3 = markers[i].corners[3].x; f.y3 = markers[i].corners[3].y;
                fva.fiducials.push_back(f);
                fiducial_msgs::FiducialTransform t;
                t.fiducial_id = markers[i].id;
                t.transform.translation.x = out.markers[i].pose.px; t.transform.translation.y = out.markers[i].pose.py;
                t.transform.translation.z = out.markers[i].pose.pz;
                t.transform.rotation.x = out.markers[i].pose.ox; t.transform.rotation.y = out.markers[i].pose.oy;
                t.transform.rotation.z = out.markers[i].pose.oz; t.transform.rotation.w = out.markers[i].pose.ow;
                fta.transforms.push_back(t);
            }
            vertices_pub_.publish(fva);
            transforms_pub_.publish(fta);
        }
    }
    void camInfoCb(const sensor_msgs::CameraInfo::ConstPtr &msg)
    {
        fa::CameraInfo ci;
        ci.header = to_host(msg->header);
        ci.height = msg->height;
        ci.width = msg->width;
        ci.distortion_model = msg->distortion_model;
        ci.D = msg->D;
        for (size_t k = 0; k < 9; k++) ci.K[k] = msg->K[k];
        node_->cameraInfoCallback(ci);
    }

    ros::NodeHandle nh_, pnh_;
    image_transport::ImageTransport it_;
    std::unique_ptr<fa::StagNode> node_;
    bool fiducial_msgs_output_ = false;
    image_transport::Subscriber image_sub_;
    image_transport::Publisher debug_pub_;
    ros::Subscriber caminfo_sub_;
    ros::Publisher markers_pub_, array_pub_, vertices_pub_, transforms_pub_;
    tf2_ros::TransformBroadcaster broadcaster_;
};

}  // namespace

int main(int argc, char **argv)
{
    ros::init(argc, argv, "stag_detect");
    try {
        RosStagNode node;
        ros::spin();  // one callback at a time: a fid_stag_ctx is single-threaded (include/fid_abi.h)
    } catch (const std::exception &e) {
        ROS_FATAL("stag_detect_amd: %s", e.what());  // (an invalid libraryHD: the reference prints the message and exits too)
        return 1;
    }
    return 0;
}
