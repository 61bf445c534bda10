// aruco_images_test.cpp -- the reference's own node test (aruco_detect/test/aruco_images_test.cpp) against the ROS-free
// FiducialsNode of host/: same camera info (:18-29), same images, same expected ids and vertices (:96-147, ASSERT_FLOAT_EQ =
// 4 ULP of float), plus the recorded bag frame (fiducial_slam/test/aruco_images.bag seq 4957 -> aruco_transforms.bag) and the
// node-surface behaviour around it (CameraInfo latching, ignore list, enable_detections, wire format).  No gtest / ROS in this
// image: plain checks, non-zero exit on the first failure.  Images come as binary PGM written by the pytest wrapper from
// tests/golden/*.npz (the gray the node's cv_bridge conversion produces).
//   usage: aruco_images_test <dir with tag_01.pgm tag_245_246.pgm bag_4957.pgm bag_4957.txt bag_4957_msg.hex> <data_dir>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

#include "fiducials_host.hpp"

using namespace fiducials_amd;

static int g_fail = 0;
#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) {                                                           \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);        \
            g_fail++;                                                            \
            return;                                                              \
        }                                                                        \
    } while (0)

// gtest's ASSERT_FLOAT_EQ: within 4 units in the last place of float
static bool float_eq(double expected, double actual)
{
    const float a = (float)expected, b = (float)actual;
    if (a == b) return true;
    int32_t ia, ib;
    std::memcpy(&ia, &a, 4);
    std::memcpy(&ib, &b, 4);
    if ((ia < 0) != (ib < 0)) return false;
    return std::abs(ia - ib) <= 4;
}
#define CHECK_FLOAT_EQ(e, a) CHECK(float_eq((e), (a)))

static Image load_pgm(const std::string &path, uint32_t seq)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::string magic;
    int w, h, maxv;
    f >> magic >> w >> h >> maxv;
    f.get();
    Image im;
    im.header.seq = seq;
    im.header.sec = 1491682360;
    im.header.nsec = 314066469;
    im.height = h;
    im.width = w;
    im.encoding = "mono8";
    im.step = w;
    im.data.resize((size_t)w * h);
    f.read((char *)im.data.data(), (std::streamsize)im.data.size());
    return im;
}

struct Fixture {  // ArucoImagesTest::SetUp
    CameraInfo c_info;
    std::string image_directory, data_dir;
    Fixture(const std::string &dir, const std::string &data) : image_directory(dir), data_dir(data)
    {
        c_info.height = 960;
        c_info.width = 1280;
        c_info.distortion_model = "plumb_bob";
        c_info.D = {0.1349735087283542, -0.2335869827451621, 0.0006697030315075139, 0.004846737465872353, 0.0};
        c_info.K = {1006.126285753055, 0.0, 655.8639244150409, 0.0, 1004.015433012594, 490.6140221242933, 0.0, 0.0, 1.0};
        c_info.header.frame_id = "camera";
    }
    FiducialsNode::Params params() const
    {
        FiducialsNode::Params p;
        p.dictionary = 7;        // aruco_images.test
        p.fiducial_len = 0.145;
        p.data_dir = data_dir;
        p.max_width = 1280;
        p.max_height = 960;
        return p;
    }
};

static void tag_01_d7_14cm(const Fixture &fx)
{
    FiducialsNode node(fx.params());
    FiducialArray fiducials;
    FiducialTransformArray fiducial_tfs;
    const Image img = load_pgm(fx.image_directory + "/tag_01.pgm", 1);
    CHECK(node.imageCallback(img, &fiducials));
    CHECK(!node.poseEstimateCallback(fiducials, &fiducial_tfs));  // no camera intrinsics yet: nothing published
    node.camInfoCallback(fx.c_info);
    CHECK(node.poseEstimateCallback(fiducials, &fiducial_tfs));
    CHECK(1 == fiducials.fiducials.size());
    CHECK(1 == fiducial_tfs.transforms.size());
    const Fiducial &vertices = fiducials.fiducials[0];
    CHECK(1 == vertices.fiducial_id);
    CHECK_FLOAT_EQ(569.89917, vertices.x0);
    CHECK_FLOAT_EQ(201.55890, vertices.y0);
    CHECK_FLOAT_EQ(777.42560, vertices.x1);
    CHECK_FLOAT_EQ(206.85025, vertices.y1);
    CHECK_FLOAT_EQ(767.95856, vertices.x2);
    CHECK_FLOAT_EQ(415.37830, vertices.y2);
    CHECK_FLOAT_EQ(565.75311, vertices.x3);
    CHECK_FLOAT_EQ(409.24496, vertices.y3);
    // header rules (:343-345, :409-411): stamp of the image, frame of the CameraInfo, image_seq = image seq
    CHECK(fiducials.image_seq == 1 && fiducials.header.sec == img.header.sec && fiducials.header.nsec == img.header.nsec);
    CHECK(fiducial_tfs.header.frame_id == "camera");
    const FiducialTransform &t = fiducial_tfs.transforms[0];
    CHECK(t.fiducial_id == 1 && t.tz > 0.3 && t.tz < 3.0 && t.fiducial_area > 30000 && t.image_error >= 0 && t.object_error >= 0);
    CHECK(std::fabs(t.qx * t.qx + t.qy * t.qy + t.qz * t.qz + t.qw * t.qw - 1.0) < 1e-12);
}

static void tag_245_246_d7_14cm(const Fixture &fx)
{
    FiducialsNode node(fx.params());
    node.camInfoCallback(fx.c_info);
    FiducialArray fiducials;
    FiducialTransformArray fiducial_tfs;
    CHECK(node.imageCallback(load_pgm(fx.image_directory + "/tag_245_246.pgm", 2), &fiducials));
    CHECK(node.poseEstimateCallback(fiducials, &fiducial_tfs));
    CHECK(2 == fiducials.fiducials.size());
    CHECK(2 == fiducial_tfs.transforms.size());
    for (auto &vertices : fiducials.fiducials) {
        if (vertices.fiducial_id == 245) {
            CHECK_FLOAT_EQ(307.68246, vertices.x0);
            CHECK_FLOAT_EQ(157.38346, vertices.y0);
            CHECK_FLOAT_EQ(545.10131, vertices.x1);
            CHECK_FLOAT_EQ(167.04420, vertices.y1);
            CHECK_FLOAT_EQ(540.11614, vertices.x2);
            CHECK_FLOAT_EQ(403.27578, vertices.y2);
            CHECK_FLOAT_EQ(305.64746, vertices.x3);
            CHECK_FLOAT_EQ(395.01422, vertices.y3);
        } else if (vertices.fiducial_id == 246) {
            CHECK_FLOAT_EQ(671.51892, vertices.x0);
            CHECK_FLOAT_EQ(173.46070, vertices.y0);
            CHECK_FLOAT_EQ(900.29650, vertices.x1);
            CHECK_FLOAT_EQ(178.44973, vertices.y1);
            CHECK_FLOAT_EQ(895.06933, vertices.x2);
            CHECK_FLOAT_EQ(407.39855, vertices.y2);
            CHECK_FLOAT_EQ(666.39910, vertices.x3);
            CHECK_FLOAT_EQ(403.12911, vertices.y3);
        } else {
            CHECK(false);
        }
    }
}

// the node-surface logic around the detector
static void node_surface(const Fixture &fx)
{
    FiducialsNode::Params p = fx.params();
    p.ignore_fiducials = "1,4,8,9-12,245";
    p.fiducial_len_override = "246: 0.29, 300-302: 0.5";
    FiducialsNode node(p);
    CHECK(node.getIgnoreIds().size() == 8 && node.getIgnoreIds()[3] == 9 && node.getIgnoreIds()[7] == 245);
    CHECK(node.getFiducialLens().size() == 4 && node.getFiducialLens().at(301) == 0.5 && node.getFiducialLens().at(246) == 0.29);
    CameraInfo zero = fx.c_info;
    zero.K.fill(0.0);
    node.camInfoCallback(zero);
    CHECK(!node.haveCameraInfo());  // all-zero K is refused (:313-328)
    node.camInfoCallback(fx.c_info);
    CameraInfo other = fx.c_info;
    other.header.frame_id = "other";
    other.K[0] = 1.0;
    node.camInfoCallback(other);  // latched: ignored
    FiducialArray fva;
    FiducialTransformArray fta, fta2;
    const Image img = load_pgm(fx.image_directory + "/tag_245_246.pgm", 7);
    CHECK(node.imageCallback(img, &fva));
    CHECK(fva.fiducials.size() == 1 && fva.fiducials[0].fiducial_id == 246);  // 245 is on the ignore list
    CHECK(node.getIds().size() == 2);                                          // ... but was detected
    CHECK(node.poseEstimateCallback(fva, &fta));
    CHECK(fta.transforms.size() == 1 && fta.transforms[0].fiducial_id == 246 && fta.header.frame_id == "camera");
    // doubling the marker length doubles the distance (per-id override, :241-244)
    node.ignoreCallback("");
    FiducialsNode plain(fx.params());
    plain.camInfoCallback(fx.c_info);
    FiducialArray f2;
    CHECK(plain.imageCallback(img, &f2) && plain.poseEstimateCallback(f2, &fta2));
    double tz_plain = 0;
    for (auto &t : fta2.transforms)
        if (t.fiducial_id == 246) tz_plain = t.tz;
    CHECK(tz_plain > 0 && std::fabs(fta.transforms[0].tz / tz_plain - 2.0) < 1e-6);
    std::string m;
    CHECK(node.enableDetectionsCallback(false, &m) && m == "Disabled aruco detections.");
    CHECK(!node.imageCallback(img, &fva));
    CHECK(node.enableDetectionsCallback(true, &m) && m == "Enabled aruco detections.");
    CHECK(node.imageCallback(img, &fva) && fva.fiducials.size() == 2);
    Image bad = img;
    bad.encoding = "32FC1";
    CHECK(!node.imageCallback(bad, &fva) && !node.lastError().empty());  // like the caught cv_bridge exception: frame dropped
    CHECK(node.lastError().find("cv_bridge exception") == 0);
    {
        // what else toCvCopy(msg, BGR8) converts (fid_image_to_bgr8): the frame as mono16 (v * 257: convertTo gives v back, so the
        // markers are those of the 8-bit frame to the last bit, both byte orders) and as a Bayer mosaic of its gray values (the
        // demosaiced copy is a slightly smoothed gray image: the same ids, corners within a pixel)
        FiducialArray ref;
        const Image &g8 = img;  // (the fixture is a mono8 frame)
        CHECK(node.imageCallback(g8, &ref) && ref.fiducials.size() == 2);
        for (int be = 0; be < 2; be++) {
            Image m16 = g8;
            m16.encoding = "mono16";  // (rows of img.step bytes in the source)
            m16.is_bigendian = (uint8_t)be;
            m16.step = img.width * 2 + 4;
            m16.data.assign((size_t)m16.step * img.height, 0);
            for (uint32_t y = 0; y < img.height; y++)
                for (uint32_t x = 0; x < img.width; x++) {
                    const unsigned v = g8.data[(size_t)y * img.step + x] * 257u;
                    m16.data[(size_t)y * m16.step + 2 * x + (be ? 1 : 0)] = (uint8_t)(v & 255);
                    m16.data[(size_t)y * m16.step + 2 * x + (be ? 0 : 1)] = (uint8_t)(v >> 8);
                }
            FiducialArray f16;
            CHECK(node.imageCallback(m16, &f16) && f16.fiducials.size() == ref.fiducials.size());
            for (size_t i = 0; i < f16.fiducials.size() && i < ref.fiducials.size(); i++)
                CHECK(f16.fiducials[i].fiducial_id == ref.fiducials[i].fiducial_id && f16.fiducials[i].x0 == ref.fiducials[i].x0 &&
                      f16.fiducials[i].y2 == ref.fiducials[i].y2);
        }
        const char *pats[4] = {"bayer_rggb8", "bayer_bggr8", "bayer_gbrg8", "bayer_grbg8"};
        for (const char *pat : pats) {
            Image by = g8;
            by.encoding = pat;
            FiducialArray fb;
            CHECK(node.imageCallback(by, &fb) && fb.fiducials.size() == ref.fiducials.size());
            for (size_t i = 0; i < fb.fiducials.size() && i < ref.fiducials.size(); i++)
                CHECK(fb.fiducials[i].fiducial_id == ref.fiducials[i].fiducial_id && std::fabs(fb.fiducials[i].x0 - ref.fiducials[i].x0) < 1.0 &&
                      std::fabs(fb.fiducials[i].y2 - ref.fiducials[i].y2) < 1.0);
        }
    }
    // ~publish_images: /fiducial_images = the BGR8 copy of the frame with the marker outlines on it (:381-387); and the third
    // corner refinement the node can select, cornerRefinementSubPix = false -> CORNER_REFINE_CONTOUR (:274-283, 700-711)
    {
        FiducialsNode::Params pi = fx.params();
        pi.publish_images = true;
        FiducialsNode inode(pi);
        FiducialArray fi;
        Image ov;
        CHECK(inode.imageCallback(img, &fi, &ov) && fi.fiducials.size() == 2);
        CHECK(ov.encoding == "bgr8" && ov.width == img.width && ov.height == img.height && ov.step == img.width * 3 &&
              ov.data.size() == (size_t)img.width * img.height * 3 && ov.header.seq == img.header.seq);
        size_t green = 0, changed = 0;
        for (uint32_t y = 0; y < img.height; y++)
            for (uint32_t x = 0; x < img.width; x++) {
                const uint8_t *p = &ov.data[((size_t)y * img.width + x) * 3];
                const uint8_t g = img.data[(size_t)y * img.step + x];
                if (p[0] == 0 && p[1] == 255 && p[2] == 0) green++;
                else if (p[0] != g || p[1] != g || p[2] != g) changed++;
            }
        CHECK(green > 400 && changed == 0);  // two marker outlines, every other pixel is the gray frame replicated
        // a corner of the first marker lies on its outline
        const int cx = (int)std::lrint(fi.fiducials[0].x0), cy = (int)std::lrint(fi.fiducials[0].y0);
        const uint8_t *pc = &ov.data[((size_t)cy * img.width + cx) * 3];
        CHECK(pc[0] == 0 && pc[1] == 255 && pc[2] == 0);
        {
            // a Bayer frame: /fiducial_images is the demosaiced BGR8 copy the detection ran on, with the outlines
            Image by = img, ovb;
            by.encoding = "bayer_grbg8";
            FiducialArray fb;
            CHECK(inode.imageCallback(by, &fb, &ovb) && fb.fiducials.size() == 2);
            CHECK(ovb.encoding == "bgr8" && ovb.step == img.width * 3 && ovb.data.size() == (size_t)img.width * img.height * 3);
            size_t g2 = 0;
            for (size_t k = 0; k + 2 < ovb.data.size(); k += 3) g2 += ovb.data[k] == 0 && ovb.data[k + 1] == 255 && ovb.data[k + 2] == 0;
            CHECK(g2 > 400);
        }
        Image none;
        FiducialsNode plain2(fx.params());
        CHECK(plain2.imageCallback(img, &fi, &none) && none.data.empty());  // publish_images = false: nothing is made
        // CORNER_REFINE_CONTOUR through dynamic_reconfigure: same ids, corners within a pixel of the SUBPIX ones, not the same
        fid_params cfg = pi.detector;
        cfg.cornerRefinementMethod = 2;
        inode.configCallback(cfg, 0);
        FiducialArray fc;
        CHECK(inode.imageCallback(img, &fc) && fc.fiducials.size() == 2 && inode.lastError().empty());
        CHECK(fc.fiducials[0].fiducial_id == fi.fiducials[0].fiducial_id);
        CHECK(std::fabs(fc.fiducials[0].x0 - fi.fiducials[0].x0) < 1.5 && std::fabs(fc.fiducials[0].y2 - fi.fiducials[0].y2) < 1.5);
        CHECK(fc.fiducials[0].x0 != fi.fiducials[0].x0 || fc.fiducials[0].y0 != fi.fiducials[0].y0);
    }
    // ~dictionary values whose shipped tables are fillers are refused by name ... unless the deployer's own table file is given
    {
        bool threw = false;
        try {
            (void)getPredefinedDictionary(10, fx.params().data_dir);
        } catch (const std::runtime_error &) {
            threw = true;
        }
        CHECK(threw);
        const Dictionary d6 = loadDictionaryFile(10, fx.params().data_dir + "/dict_6x6_1000.txt");
        CHECK(d6.markerSize == 6 && d6.nMarkers == 250 && d6.maxCorrectionBits == 5 && d6.bytesList.size() == (size_t)250 * 4 * 5);
        const Dictionary d5 = loadDictionaryFile(7, fx.params().data_dir + "/dict_5x5_1000.txt");
        const Dictionary ref5 = getPredefinedDictionary(7, fx.params().data_dir);
        CHECK(d5.bytesList == ref5.bytesList && d5.maxCorrectionBits == ref5.maxCorrectionBits);
        FiducialsNode::Params pf = fx.params();
        pf.dictionary_file = pf.data_dir + "/dict_5x5_1000.txt";
        FiducialsNode fnode(pf);
        FiducialArray ff;
        CHECK(fnode.imageCallback(img, &ff) && ff.fiducials.size() == 2);
    }
    // ~vis_msgs: vision_msgs/Detection2DArray instead of FiducialTransformArray, and the TF broadcasts (:462-478, 501-524)
    {
        FiducialsNode::Params pv = fx.params();
        pv.vis_msgs = true;
        pv.ignore_fiducials = "245";
        FiducialsNode vnode(pv);
        vnode.camInfoCallback(fx.c_info);
        FiducialArray fv;
        PoseOutputs po, pf;
        CHECK(vnode.imageCallback(img, &fv) && vnode.poseEstimateCallback(fv, &po));
        CHECK(po.vis_msgs && po.fta.transforms.empty() && po.vma.detections.size() == 1);  // 245 ignored, 246 published
        CHECK(po.vma.header.frame_id == "camera" && po.vma.header.seq == fv.header.seq);
        CHECK(plain.poseEstimateCallback(f2, &pf) && !pf.vis_msgs && pf.vma.detections.empty() && pf.fta.transforms.size() == 2);
        const ObjectHypothesisWithPose &h = po.vma.detections[0].results.at(0);
        CHECK(h.id == 246);
        for (auto &t : pf.fta.transforms)
            if (t.fiducial_id == 246) {
                CHECK(h.score == std::exp(-2 * t.object_error) && h.score > 0 && h.score <= 1);
                CHECK(h.pose.px == t.tx && h.pose.py == t.ty && h.pose.pz == t.tz && h.pose.ox == t.qx && h.pose.ow == t.qw);
            }
        CHECK(po.tf.size() == 1 && po.tf[0].child_frame_id == "fiducial_246" && po.tf[0].header.frame_id == "camera");
        CHECK(po.tf[0].tx == h.pose.px && po.tf[0].qw == h.pose.ow);
        CHECK(pf.tf.size() == 2 && pf.tf[0].child_frame_id == "fiducial_" + std::to_string(pf.fta.transforms[0].fiducial_id));
        FiducialsNode::Params pn = fx.params();
        pn.publish_fiducial_tf = false;
        FiducialsNode qnode(pn);
        qnode.camInfoCallback(fx.c_info);
        PoseOutputs pq;
        CHECK(qnode.imageCallback(img, &fv) && qnode.poseEstimateCallback(fv, &pq) && pq.tf.empty() && pq.fta.transforms.size() == 2);
    }
    // wire format: 72 / 84 bytes per element
    CHECK(serialize(fva).size() == 16 + fva.header.frame_id.size() + 8 + 72 * fva.fiducials.size());
    CHECK(serialize(fta).size() == 16 + fta.header.frame_id.size() + 8 + 84 * fta.transforms.size());
}

// the recorded bag frame: the node's output message, as recorded by the reference, re-serialised byte for byte; and this
// node's output for the recorded image next to it (ids equal; poses to the tolerance of tests/test_gpu_parity.py)
static void bag_4957(const Fixture &fx)
{
    std::ifstream hx(fx.image_directory + "/bag_4957_msg.hex");
    std::string hex;
    hx >> hex;
    std::vector<uint8_t> raw(hex.size() / 2);
    for (size_t i = 0; i < raw.size(); i++) raw[i] = (uint8_t)std::stoi(hex.substr(2 * i, 2), nullptr, 16);
    FiducialTransformArray rec;
    CHECK(deserialize(raw, &rec));
    CHECK(rec.image_seq == 4957 && rec.transforms.size() == 7 && rec.header.frame_id == "raspicam");
    CHECK(serialize(rec) == raw);
    std::ifstream kf(fx.image_directory + "/bag_4957.txt");
    CameraInfo ci;
    ci.header.frame_id = "raspicam";
    ci.D.resize(5);
    for (double &v : ci.K) kf >> v;
    for (double &v : ci.D) kf >> v;
    FiducialsNode::Params p = fx.params();
    p.fiducial_len = 0.14;
    FiducialsNode node(p);
    node.camInfoCallback(ci);
    FiducialArray fva;
    FiducialTransformArray fta;
    CHECK(node.imageCallback(load_pgm(fx.image_directory + "/bag_4957.pgm", 4957), &fva));
    CHECK(node.poseEstimateCallback(fva, &fta));
    // (:409-411) image_seq of the transforms = header.seq of the vertices message, which roscpp leaves as the node set it
    CHECK(fva.image_seq == 4957 && fta.image_seq == (int32_t)fva.header.seq && fta.header.frame_id == "raspicam");
    int matched = 0;
    for (auto &r : rec.transforms)
        for (auto &t : fta.transforms)
            if (t.fiducial_id == r.fiducial_id) {
                matched++;
                // recorded by an older OpenCV from a JPEG: a few percent (tests/test_oracle_golden.py has the tight check)
                const double dt = std::sqrt((t.tx - r.tx) * (t.tx - r.tx) + (t.ty - r.ty) * (t.ty - r.ty) + (t.tz - r.tz) * (t.tz - r.tz));
                CHECK(dt < 0.05 * std::sqrt(r.tx * r.tx + r.ty * r.ty + r.tz * r.tz));
                const double dot = std::fabs(t.qx * r.qx + t.qy * r.qy + t.qz * r.qz + t.qw * r.qw);
                CHECK(dot > 0.995);
            }
    CHECK(matched >= 6);
}

// a frame that arrives compressed (image_transport `compressed`, the launch default): decoded on the device, the result is
// what the same frame gives when it arrives raw -- decoded by the checker's restatement of libjpeg (tag_01_jpg.pgm, written
// by tests/test_gpu_host_cpp.py next to tag_01.jpg)
static void compressed_frame(const Fixture &fx)
{
    std::ifstream jf(fx.image_directory + "/tag_01.jpg", std::ios::binary);
    if (!jf) {
        std::printf("(no tag_01.jpg: compressed-frame check skipped)\n");
        return;
    }
    CompressedImage cm;
    cm.header.seq = 7;
    cm.format = "mono8; jpeg compressed mono8";
    cm.data.assign(std::istreambuf_iterator<char>(jf), std::istreambuf_iterator<char>());
    FiducialsNode node(fx.params());
    FiducialArray a, b;
    CHECK(node.compressedImageCallback(cm, &a));
    CHECK(node.imageCallback(load_pgm(fx.image_directory + "/tag_01_jpg.pgm", 7), &b));
    CHECK(a.image_seq == 7 && a.fiducials.size() == b.fiducials.size() && a.fiducials.size() == 1);
    for (size_t i = 0; i < a.fiducials.size() && i < b.fiducials.size(); i++) {
        CHECK(a.fiducials[i].fiducial_id == b.fiducials[i].fiducial_id && a.fiducials[i].fiducial_id == 1);
        CHECK(a.fiducials[i].x0 == b.fiducials[i].x0 && a.fiducials[i].y0 == b.fiducials[i].y0 && a.fiducials[i].x2 == b.fiducials[i].x2 &&
              a.fiducials[i].y2 == b.fiducials[i].y2);
    }
    // a frame that is not a JPEG is dropped with a message, as the subscriber plugin does
    cm.data.assign(64, 0x41);
    CHECK(!node.compressedImageCallback(cm, &a) && !node.lastError().empty());
    // the same frame with `format: png` (lossless: exactly what the raw frame gives), as a gray and as a colour file
    FiducialArray raw;
    CHECK(node.imageCallback(load_pgm(fx.image_directory + "/tag_01.pgm", 9), &raw));
    for (const char *name : {"/tag_01_gray.png", "/tag_01_rgb.png"}) {
        std::ifstream pf(fx.image_directory + name, std::ios::binary);
        if (!pf) {
            std::printf("(no %s: PNG frame check skipped)\n", name + 1);
            continue;
        }
        CompressedImage pm;
        pm.header.seq = 9;
        pm.format = "mono8; png compressed ";
        pm.data.assign(std::istreambuf_iterator<char>(pf), std::istreambuf_iterator<char>());
        FiducialArray c;
        CHECK(node.compressedImageCallback(pm, &c));
        CHECK(c.image_seq == 9 && c.fiducials.size() == raw.fiducials.size() && c.fiducials.size() == 1);
        for (size_t i = 0; i < c.fiducials.size() && i < raw.fiducials.size(); i++) {
            const auto &p = c.fiducials[i], &q = raw.fiducials[i];
            CHECK(p.fiducial_id == q.fiducial_id && p.x0 == q.x0 && p.y0 == q.y0 && p.x1 == q.x1 && p.y1 == q.y1 && p.x2 == q.x2 && p.y2 == q.y2 &&
                  p.x3 == q.x3 && p.y3 == q.y3);
        }
        pm.data.resize(pm.data.size() / 2);  // a truncated file is dropped with a message
        CHECK(!node.compressedImageCallback(pm, &c) && !node.lastError().empty());
    }
}

int main(int argc, char **argv)
{
    if (argc < 3) {
        std::printf("usage: %s <image dir> <data dir>\n", argv[0]);
        return 2;
    }
    try {
        Fixture fx(argv[1], argv[2]);
        tag_01_d7_14cm(fx);
        tag_245_246_d7_14cm(fx);
        node_surface(fx);
        bag_4957(fx);
        compressed_frame(fx);
    } catch (const std::exception &e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 3;
    }
    std::printf(g_fail ? "%d check(s) failed\n" : "all checks passed%.0d\n", g_fail);
    return g_fail ? 1 : 0;
}
