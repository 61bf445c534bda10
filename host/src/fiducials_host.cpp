// fiducials_host.cpp -- see fiducials_host.hpp.  Line references are to /root/reference/aruco_detect/src/aruco_detect.cpp.
#include "fiducials_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace fiducials_amd {

// ------------------------------------------------------------------------------------------------ wire format
namespace {
template <typename T>
void put(std::vector<uint8_t> &b, T v)
{
    uint8_t raw[sizeof(T)];
    std::memcpy(raw, &v, sizeof(T));
    b.insert(b.end(), raw, raw + sizeof(T));
}
void put_header(std::vector<uint8_t> &b, const Header &h)
{
    put<uint32_t>(b, h.seq);
    put<uint32_t>(b, h.sec);
    put<uint32_t>(b, h.nsec);
    put<uint32_t>(b, (uint32_t)h.frame_id.size());
    b.insert(b.end(), h.frame_id.begin(), h.frame_id.end());
}
template <typename T>
bool get(const std::vector<uint8_t> &b, size_t &p, T *v)
{
    if (p + sizeof(T) > b.size()) return false;
    std::memcpy(v, b.data() + p, sizeof(T));
    p += sizeof(T);
    return true;
}
}  // namespace

std::vector<uint8_t> serialize(const FiducialArray &m)
{
    std::vector<uint8_t> b;
    put_header(b, m.header);
    put<int32_t>(b, m.image_seq);
    put<uint32_t>(b, (uint32_t)m.fiducials.size());
    for (const Fiducial &f : m.fiducials) {  // 72 bytes each
        put<int32_t>(b, f.fiducial_id);
        put<int32_t>(b, f.direction);
        for (double v : {f.x0, f.y0, f.x1, f.y1, f.x2, f.y2, f.x3, f.y3}) put<double>(b, v);
    }
    return b;
}

std::vector<uint8_t> serialize(const FiducialTransformArray &m)
{
    std::vector<uint8_t> b;
    put_header(b, m.header);
    put<int32_t>(b, m.image_seq);
    put<uint32_t>(b, (uint32_t)m.transforms.size());
    for (const FiducialTransform &t : m.transforms) {  // 84 bytes each
        put<int32_t>(b, t.fiducial_id);
        for (double v : {t.tx, t.ty, t.tz, t.qx, t.qy, t.qz, t.qw, t.image_error, t.object_error, t.fiducial_area}) put<double>(b, v);
    }
    return b;
}

bool deserialize(const std::vector<uint8_t> &b, FiducialTransformArray *m)
{
    size_t p = 0;
    uint32_t n = 0, len = 0;
    if (!get(b, p, &m->header.seq) || !get(b, p, &m->header.sec) || !get(b, p, &m->header.nsec) || !get(b, p, &len) || p + len > b.size()) return false;
    m->header.frame_id.assign((const char *)b.data() + p, len);
    p += len;
    if (!get(b, p, &m->image_seq) || !get(b, p, &n)) return false;
    m->transforms.resize(n);
    for (FiducialTransform &t : m->transforms) {
        double *f[10] = {&t.tx, &t.ty, &t.tz, &t.qx, &t.qy, &t.qz, &t.qw, &t.image_error, &t.object_error, &t.fiducial_area};
        if (!get(b, p, &t.fiducial_id)) return false;
        for (double *v : f)
            if (!get(b, p, v)) return false;
    }
    return p == b.size();
}

// ------------------------------------------------------------------------------------------------ dictionary
fid_dict Dictionary::view() const
{
    fid_dict d;
    std::memset(&d, 0, sizeof(d));
    d.marker_size = markerSize;
    d.max_correction_bits = maxCorrectionBits;
    d.n_markers = nMarkers;
    d.bytes = bytesList.data();
    return d;
}

Dictionary getPredefinedDictionary(int dicno, const std::string &data_dir)
{
    // enum value -> (marker size, nMarkers, maxCorrectionBits) for the dictionaries whose tables ship with this repository
    struct Row { int dicno, n, count, maxc; };
    static const Row rows[] = {{0, 4, 50, 1}, {1, 4, 100, 1}, {2, 4, 250, 1}, {4, 5, 50, 3}, {5, 5, 100, 3}, {6, 5, 250, 2}, {7, 5, 1000, 2}};
    const Row *row = nullptr;
    for (const Row &r : rows)
        if (r.dicno == dicno) row = &r;
    if (!row)  // (4X4_1000, 6X6, 7X7, ARUCO_ORIGINAL: the shipped tables are labelled fillers, not OpenCV's codewords)
        throw std::runtime_error("dictionary " + std::to_string(dicno) + " not available in this build: give the node OpenCV's table (~dictionary_file, loadDictionaryFile)");
    const int n = row->n, nbytes = (n * n + 7) / 8;
    std::ifstream f(data_dir + (n == 4 ? "/dict_4x4_250.txt" : "/dict_5x5_1000.txt"));
    if (!f) throw std::runtime_error("dictionary table not found under " + data_dir);
    Dictionary d;
    d.markerSize = n;
    d.maxCorrectionBits = row->maxc;
    d.nMarkers = row->count;
    d.bytesList.assign((size_t)row->count * 4 * nbytes, 0);
    std::string line;
    int have = 0;
    while (have < row->count && std::getline(f, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream is(line);
        int idx;
        std::string flag, hex;
        is >> idx >> flag >> hex;
        const unsigned long long word = std::stoull(hex, nullptr, 16);
        std::vector<int> bits(n * n), rot(n * n);
        for (int k = 0; k < n * n; k++) bits[k] = (int)((word >> (n * n - 1 - k)) & 1);
        for (int r = 0; r < 4; r++) {  // Dictionary::getByteListFromBits: rotation r = r quarter turns counter-clockwise
            uint8_t *out = &d.bytesList[((size_t)have * 4 + r) * nbytes];
            int cur = 0;
            for (int i = 0; i < n * n; i++) {
                out[cur] = (uint8_t)((out[cur] << 1) | bits[i]);
                if (i % 8 == 7) cur++;
            }
            for (int i = 0; i < n; i++)
                for (int j = 0; j < n; j++) rot[i * n + j] = bits[j * n + (n - 1 - i)];
            bits = rot;
        }
        have++;
    }
    if (have != row->count) throw std::runtime_error("dictionary table too short");
    return d;
}

Dictionary loadDictionaryFile(int dicno, const std::string &table_file)
{
    fid_dict fd;
    fid_status rc = fid_dict_load_file(table_file.c_str(), dicno, nullptr, 0, &fd);  // sizes first
    if (rc != FID_E_CAPACITY) throw std::runtime_error("dictionary file " + table_file + ": " + fid_dict_last_error());
    Dictionary d;
    d.markerSize = fd.marker_size;
    d.maxCorrectionBits = fd.max_correction_bits;
    d.nMarkers = fd.n_markers;
    d.bytesList.assign((size_t)fd.n_markers * 4 * ((fd.marker_size * fd.marker_size + 7) / 8), 0);
    rc = fid_dict_load_file(table_file.c_str(), dicno, d.bytesList.data(), (int64_t)d.bytesList.size(), &fd);
    if (rc != FID_OK) throw std::runtime_error("dictionary file " + table_file + ": " + fid_dict_last_error());
    return d;
}

// ------------------------------------------------------------------------------------------------ node
FiducialsNode::Params::Params()
{
    fid_default_params(&detector);  // the node's values (:690-727), not OpenCV's defaults
}

FiducialsNode::FiducialsNode(const Params &p)
{
    frameNum = 0;
    haveCamInfo = false;
    enable_detections = true;
    fiducial_len = p.fiducial_len;
    doPoseEstimation = p.do_pose_estimation;
    verbose = p.verbose;
    publish_images = p.publish_images;      // (:609)
    vis_msgs = p.vis_msgs;                  // (:616)
    publishFiducialTf = p.publish_fiducial_tf;  // (:614)
    handleIgnoreString(p.ignore_fiducials);
    handleLenOverrideString(p.fiducial_len_override);
    dict = p.dictionary_file.empty() ? getPredefinedDictionary(p.dictionary, p.data_dir) : loadDictionaryFile(p.dictionary, p.dictionary_file);
    detectorParams = p.detector;
    fid_dict fd = dict.view();
    fid_limits lim;
    fid_default_limits(&lim);
    lim.max_width = p.max_width;
    lim.max_height = p.max_height;
    lim.max_batch = 1;
    maxW = p.max_width;
    maxH = p.max_height;
    dev = p.device;
    const fid_status rc = fid_create(&detectorParams, &fd, &lim, p.device, &ctx);
    if (rc != FID_OK) throw std::runtime_error(std::string("fid_create: ") + fid_strerror(rc));
    markers.resize(1024);
}

FiducialsNode::~FiducialsNode()
{
    if (jctx) fid_jpeg_destroy(jctx);
    fid_destroy(ctx);
}

static int stoi_like(const std::string &s) { return std::stoi(s); }  // the node uses std::stoi: same acceptance, same throws

static std::vector<std::string> split(const std::string &s, char c)  // boost::split(.., is_any_of(c)) without compression
{
    std::vector<std::string> out;
    std::string cur;
    for (char ch : s) {
        if (ch == c) {
            out.push_back(cur);
            cur.clear();
        } else
            cur.push_back(ch);
    }
    out.push_back(cur);
    return out;
}

void FiducialsNode::handleIgnoreString(const std::string &str)
{
    for (const std::string &element : split(str, ',')) {
        if (element == "") continue;
        const std::vector<std::string> range = split(element, '-');
        if (range.size() == 2) {
            const int start = stoi_like(range[0]), end = stoi_like(range[1]);
            for (int j = start; j <= end; j++) ignoreIds.push_back(j);
        } else if (range.size() == 1) {
            ignoreIds.push_back(stoi_like(range[0]));
        }  // else: malformed (the node logs an error)
    }
}

void FiducialsNode::handleLenOverrideString(const std::string &str)
{
    for (const std::string &element : split(str, ',')) {
        if (element == "") continue;
        const std::vector<std::string> parts = split(element, ':');
        if (parts.size() != 2) continue;  // malformed
        const double len = std::stod(parts[1]);
        const std::vector<std::string> range = split(element, '-');  // (the whole element, as the node does)
        if (range.size() == 2) {
            const int start = stoi_like(range[0]), end = stoi_like(range[1]);
            for (int j = start; j <= end; j++) fiducialLens[j] = len;
        } else if (range.size() == 1) {
            fiducialLens[stoi_like(range[0])] = len;
        }
    }
}

void FiducialsNode::configCallback(const fid_params &config, uint32_t level)
{
    if (level == 0xFFFFFFFF) return;  // don't load the initial config (:260-262)
    // the node's view of the parameters follows the context: a rejected reconfigure leaves both as they were
    const fid_status rc = fid_set_params(ctx, &config);
    if (rc == FID_OK)
        detectorParams = config;
    else
        last_error = fid_last_error(ctx);
}

void FiducialsNode::ignoreCallback(const std::string &msg)
{
    ignoreIds.clear();
    handleIgnoreString(msg);
}

void FiducialsNode::camInfoCallback(const CameraInfo &msg)
{
    if (haveCamInfo) return;  // the first valid one is latched
    bool all_zero = true;
    for (double v : msg.K) all_zero = all_zero && v == 0.0;
    if (all_zero) return;  // "CameraInfo message has invalid intrinsics, K matrix all zeros"
    for (int i = 0; i < 9; i++) cameraMatrix[i] = msg.K[i];
    for (int i = 0; i < 5; i++) distortionCoeffs[i] = msg.D.at(i);  // (the node indexes D[0..4] unchecked)
    haveCamInfo = true;
    frameId = msg.header.frame_id;
}

bool FiducialsNode::enableDetectionsCallback(bool data, std::string *message)
{
    enable_detections = data;
    if (message) *message = enable_detections ? "Enabled aruco detections." : "Disabled aruco detections.";
    return true;
}

bool FiducialsNode::publishVertices(const Header &h, int32_t n, FiducialArray *out)
{
    FiducialArray fva;
    fva.header.sec = h.sec;
    fva.header.nsec = h.nsec;
    fva.header.frame_id = frameId;
    fva.image_seq = (int32_t)h.seq;
    ids.resize(n);
    for (int i = 0; i < n; i++) ids[i] = markers[i].id;
    for (int i = 0; i < n; i++) {
        if (std::count(ignoreIds.begin(), ignoreIds.end(), ids[i]) != 0) continue;
        const float *c = markers[i].corners;
        Fiducial fid;
        fid.fiducial_id = ids[i];
        fid.x0 = c[0]; fid.y0 = c[1]; fid.x1 = c[2]; fid.y1 = c[3];
        fid.x2 = c[4]; fid.y2 = c[5]; fid.x3 = c[6]; fid.y3 = c[7];
        fva.fiducials.push_back(fid);
    }
    *out = fva;
    return true;
}

bool FiducialsNode::imageCallback(const Image &msg, FiducialArray *out)
{
    if (enable_detections == false) return false;
    // Every encoding cv_bridge::toCvCopy(msg, BGR8) converts (:348) goes to the device AS PUBLISHED: the five 8-bit layouts, and since
    // ABI 7 the 8-bit Bayer mosaics, the 16-bit gray / colour layouts and UYVY of raw camera drivers -- the conversion and BGR2GRAY
    // are the first kernel of fid_detect, so a 1080p mosaic crosses PCIe as 2.07 MB (round 5 made a 6.2 MB BGR8 copy on one host
    // thread first).  An encoding that is not restated ends like the cv_bridge exception the reference catches (:389-391).
    fid_encoding enc = FID_ENC_MONO8;
    int32_t bpp = 1;
    if (fid_encoding_from_string(msg.encoding.c_str(), msg.is_bigendian, &enc, &bpp) != FID_OK) {
        last_error = "cv_bridge exception: unsupported encoding " + msg.encoding;
        converted.clear();
        return false;
    }
    raw_encoding = (int)enc > (int)FID_ENC_RGBA8;
    if (raw_encoding && (msg.height == 0 || msg.width == 0 || (size_t)msg.step < (size_t)msg.width * (size_t)bpp ||
                         msg.data.size() < (size_t)msg.step * msg.height || (enc == FID_ENC_YUV422 && (msg.width & 1)))) {
        last_error = "cv_bridge exception: image is wrongly formed (" + msg.encoding + "): step * height exceeds the data";
        converted.clear();
        return false;
    }
    converted.clear();
    int32_t n = 0;
    const fid_status rc = fid_detect(ctx, msg.data.data(), (int32_t)msg.width, (int32_t)msg.height, (int32_t)msg.step, enc, markers.data(),
                                     (int32_t)markers.size(), &n);
    if (rc != FID_OK) {  // the node catches cv::Exception, logs and drops the frame (:392-394)
        last_error = fid_last_error(ctx);
        return false;
    }
    return publishVertices(msg.header, n, out);
}

bool FiducialsNode::imageCallback(const Image &msg, FiducialArray *out, Image *image)
{
    if (!imageCallback(msg, out)) return false;
    if (!publish_images || !image) return true;
    // cv_ptr = toCvCopy(msg, BGR8); if (ids.size() > 0) drawDetectedMarkers(cv_ptr->image, corners, ids); image_pub.publish(cv_ptr->toImageMsg())
    fid_encoding enc = FID_ENC_MONO8;
    (void)fid_encoding_from_string(msg.encoding.c_str(), msg.is_bigendian, &enc, nullptr);  // (imageCallback accepted it)
    image->header = msg.header;  // (cv_bridge keeps the source header)
    image->height = msg.height;
    image->width = msg.width;
    image->encoding = "bgr8";
    image->is_bigendian = 0;
    image->step = msg.width * 3;
    image->data.resize((size_t)msg.width * msg.height * 3);
    if (raw_encoding) {
        // a 16-bit / Bayer / UYVY frame: the detection ran on the message bytes; the overlay's BGR8 image is toCvCopy(msg, BGR8) made
        // here, on the host, only because /fiducial_images is a host-side message (~publish_images is a debugging aid)
        fid_status rcc = fid_image_to_bgr8(msg.data.data(), (int32_t)msg.width, (int32_t)msg.height, (int32_t)msg.step, msg.encoding.c_str(),
                                           msg.is_bigendian, image->data.data(), (int64_t)image->data.size());
        if (rcc == FID_OK && !ids.empty())
            rcc = fid_draw_detected_markers(image->data.data(), (int32_t)image->width, (int32_t)image->height, (int32_t)image->step, markers.data(),
                                            (int32_t)ids.size(), 0);
        if (rcc != FID_OK) {
            last_error = std::string("overlay: ") + fid_strerror(rcc);
            image->data.clear();
        }
        return true;
    }
    fid_status rc = fid_to_bgr(msg.data.data(), (int32_t)msg.width, (int32_t)msg.height, (int32_t)msg.step, enc, image->data.data(),
                               (int64_t)image->data.size());
    if (rc == FID_OK && !ids.empty())  // every detected marker is drawn, ignored ids included (the ignore list only filters the vertices)
        rc = fid_draw_detected_markers(image->data.data(), (int32_t)msg.width, (int32_t)msg.height, (int32_t)image->step, markers.data(),
                                       (int32_t)ids.size(), 0);
    if (rc != FID_OK) {
        last_error = std::string("overlay: ") + fid_strerror(rc);
        image->data.clear();
    }
    return true;
}

bool FiducialsNode::compressedImageCallback(const CompressedImage &msg, FiducialArray *out)
{
    if (enable_detections == false) return false;
    if (!jctx) {
        const fid_status rc = fid_jpeg_create(dev, maxW, maxH, 1, &jctx);
        if (rc != FID_OK) {
            last_error = std::string("fid_jpeg_create: ") + fid_strerror(rc);
            return false;
        }
    }
    const uint8_t *file = msg.data.data();
    const int64_t nbytes = (int64_t)msg.data.size();
    static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (nbytes >= 8 && !memcmp(file, png_sig, 8)) {
        // format "...; png compressed ...": a zlib stream, decoded on the host as the subscriber plugin's cv::imdecode does
        // (fid_png_decode); gray images stay one byte per pixel, colour ones go up as BGR and are converted on the device
        fid_png_info pi = {};
        fid_status rc = fid_png_probe(file, nbytes, &pi);
        const bool oversize = rc == FID_OK && (pi.width > maxW || pi.height > maxH);
        if (oversize) rc = FID_E_INVALID_ARG;
        fid_encoding enc = FID_ENC_MONO8;
        int px = 1;
        if (rc == FID_OK) {  // (the header is only looked at once it has been parsed)
            enc = pi.gray ? FID_ENC_MONO8 : FID_ENC_BGR8;
            px = pi.gray ? 1 : 3;
            png_frame.resize((size_t)pi.width * pi.height * px);
            rc = fid_png_decode(file, nbytes, enc, png_frame.data(), (int64_t)png_frame.size(), nullptr);
        }
        if (rc != FID_OK) {
            last_error = std::string("compressed frame: ") + (oversize ? "larger than the context" : fid_png_last_error());
            return false;
        }
        int32_t n = 0;
        rc = fid_detect(ctx, png_frame.data(), pi.width, pi.height, pi.width * px, enc, markers.data(), (int32_t)markers.size(), &n);
        if (rc != FID_OK) {
            last_error = fid_last_error(ctx);
            return false;
        }
        return publishVertices(msg.header, n, out);
    }
    // gray = cvtColor(BGR2GRAY) of what cv::imdecode returns, left on the device
    fid_status rc = fid_jpeg_decode(jctx, &file, &nbytes, 1, FID_ENC_MONO8, nullptr, 0);
    if (rc != FID_OK) {  // (the subscriber plugin logs and drops a frame it cannot decode)
        last_error = std::string("compressed frame: ") + fid_jpeg_last_error(jctx);
        return false;
    }
    int32_t w = 0, h = 0, stride = 0, n = 0;
    int64_t fstride = 0;
    const void *gray = fid_jpeg_device_ptr(jctx, &w, &h, &stride, &fstride);
    rc = fid_detect_device(ctx, gray, 1, w, h, stride, fstride, FID_ENC_MONO8, markers.data(), (int32_t)markers.size(), &n);
    if (rc != FID_OK) {
        last_error = fid_last_error(ctx);
        return false;
    }
    return publishVertices(msg.header, n, out);
}

bool FiducialsNode::poseEstimateCallback(const FiducialArray &msg, FiducialTransformArray *out)
{
    PoseOutputs po;
    const bool keep = vis_msgs;
    vis_msgs = false;  // this overload is the fiducial_msgs view whatever ~vis_msgs says
    const bool ok = poseEstimateCallback(msg, &po);
    vis_msgs = keep;
    if (ok) *out = po.fta;
    return ok;
}

bool FiducialsNode::poseEstimateCallback(const FiducialArray &msg, PoseOutputs *out)
{
    PoseOutputs po;
    po.vis_msgs = vis_msgs;
    if (vis_msgs) {  // (:403-407)
        po.vma.header.sec = msg.header.sec;
        po.vma.header.nsec = msg.header.nsec;
        po.vma.header.frame_id = frameId;
        po.vma.header.seq = msg.header.seq;
    } else {  // (:408-412)
        po.fta.header.sec = msg.header.sec;
        po.fta.header.nsec = msg.header.nsec;
        po.fta.header.frame_id = frameId;
        po.fta.image_seq = (int32_t)msg.header.seq;
    }
    frameNum++;
    if (doPoseEstimation) {
        if (!haveCamInfo) {
            if (frameNum > 5) last_error = "No camera intrinsics";
            return false;
        }
        const int n = (int)ids.size();
        std::vector<double> lens(n, fiducial_len);
        for (int i = 0; i < n; i++) {  // estimatePoseSingleMarkers: per-id length override (:241-244)
            auto it = fiducialLens.find(ids[i]);
            if (it != fiducialLens.end()) lens[i] = it->second;
        }
        std::vector<fid_pose_out> poses(n > 0 ? n : 1);
        const fid_status rc = fid_pose(ctx, cameraMatrix, distortionCoeffs, markers.data(), lens.data(), n, fiducial_len, poses.data());
        if (rc != FID_OK) {
            last_error = fid_last_error(ctx);
            return false;
        }
        for (int i = 0; i < n; i++) {
            if (std::count(ignoreIds.begin(), ignoreIds.end(), ids[i]) != 0) continue;
            const double *r = poses[i].rvec, *t = poses[i].tvec;
            const double angle = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);  // norm(rvecs[i])
            const double ax = r[0] / angle, ay = r[1] / angle, az = r[2] / angle;
            // tf2::Quaternion::setRotation(axis, angle)
            const double d = std::sqrt(ax * ax + ay * ay + az * az);
            const double s = std::sin(angle * 0.5) / d;
            const double qx = ax * s, qy = ay * s, qz = az * s, qw = std::cos(angle * 0.5);
            if (vis_msgs) {  // (:462-478)
                ObjectHypothesisWithPose vmh;
                vmh.id = ids[i];
                vmh.score = std::exp(-2 * poses[i].object_error);  // [0, infinity] -> [1, 0]
                vmh.pose.px = t[0]; vmh.pose.py = t[1]; vmh.pose.pz = t[2];
                vmh.pose.ox = qx; vmh.pose.oy = qy; vmh.pose.oz = qz; vmh.pose.ow = qw;
                Detection2D vm;
                vm.results.push_back(vmh);
                po.vma.detections.push_back(vm);
            } else {  // (:480-498)
                FiducialTransform ft;
                ft.fiducial_id = ids[i];
                ft.tx = t[0]; ft.ty = t[1]; ft.tz = t[2];
                ft.qx = qx; ft.qy = qy; ft.qz = qz; ft.qw = qw;
                ft.fiducial_area = poses[i].fiducial_area;
                ft.image_error = poses[i].image_error;
                ft.object_error = poses[i].object_error;
                po.fta.transforms.push_back(ft);
            }
            if (publishFiducialTf) {  // the fiducial relative to the camera (:501-524)
                TransformStamped ts;
                ts.tx = t[0]; ts.ty = t[1]; ts.tz = t[2];
                ts.qx = qx; ts.qy = qy; ts.qz = qz; ts.qw = qw;
                ts.header.frame_id = frameId;
                ts.header.sec = msg.header.sec;
                ts.header.nsec = msg.header.nsec;
                ts.child_frame_id = "fiducial_" + std::to_string(ids[i]);
                po.tf.push_back(ts);
            }
        }
    }
    *out = po;
    return true;
}

}  // namespace fiducials_amd
