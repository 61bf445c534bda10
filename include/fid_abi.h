/*
 * fid_abi.h -- C-ABI of the MI355X-native fiducial detection front-end (libfid_amd.so).
 *
 * This is the seam a maintainer of UbiquityRobotics/fiducials binds to in order to replace the
 * OpenCV calls on aruco_detect's hot path (reference paths relative to /root/reference):
 *
 *   fid_detect / fid_detect_batch / fid_detect_device
 *        replaces  cv_bridge::toCvCopy(msg, BGR8)  +  aruco::detectMarkers(image, dictionary,
 *        corners, ids, detectorParams)            aruco_detect/src/aruco_detect.cpp:348,350
 *   fid_pose
 *        replaces  FiducialsNode::estimatePoseSingleMarkers (cv::solvePnP per marker, :223-255,
 *        call :247), getReprojectionError (cv::projectPoints, :203-221), calcFiducialArea
 *        (:179-200) and the object_error formula (:455-457,:493-495)
 *   fid_jpeg_decode
 *        replaces  cv::imdecode in image_transport's compressed subscriber, in front of the callback when
 *        the node runs with the launch default transport:=compressed (aruco_detect.launch:6)
 *   fid_png_decode
 *        replaces  cv::imdecode for frames the same subscriber receives with format png (host code, like the reference's)
 *   fid_stag_*   the second front end: Stag::detectMarkers + the 5-point pose of stag_detect
 *   fid_params   mirrors aruco::DetectorParameters as the node fills it      (:690-727)
 *   fid_dict     mirrors aruco::Dictionary{bytesList, markerSize, maxCorrectionBits} as returned
 *                by aruco::getPredefinedDictionary(dicno)                     (:671)
 *
 * Rules: plain C, caller-allocated outputs with capacity + count, integer status codes, nothing
 * throws across the boundary.  A context is single-threaded (the node runs under ros::spin(),
 * aruco_detect.cpp:737); several contexts (one per GPU / stream) may be used concurrently.
 * Host image memory may be pageable.  Device pointers (fid_detect_device) must be on ctx's device.
 *
 * The reference-side bindings are shown in INTEGRATION.md.
 */
#ifndef FID_ABI_H
#define FID_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 1: round 1 (aruco path).  2: + fid_detect_device / fid_pose_last / limits, the fid_stag_* family, the fid_jpeg_* family (round 2,
 * which forgot to bump it).  3: fid_last_stage_ms reports 15 stages (seedless_chain); fid_pose_last may hand over poses that the
 * preceding fid_detect_* call already computed for the same camera; fid_stag_detect_markers_batch reports 0 markers for a frame
 * whose slot was too small (round 3).  4: + fid_submit_device / fid_submit_batch / fid_collect / fid_order_after, fid_png_* (round 3).  Entry points are only ever added: a caller
 * built against 1 runs against 5.  5: cornerRefinementMethod 2 (CORNER_REFINE_CONTOUR) is implemented instead of refused,
 * FID_E_CV_EXCEPTION, fid_refine_contour_corners, fid_to_bgr / fid_draw_detected_markers, fid_dict_load_file (round 4).
 * 6: + fid_stag_queue_stats (STag frames queued ahead of their own counts), fid_image_to_bgr8 (round 5).
 * 7: fid_detect* / fid_submit* take the raw-camera encodings themselves (FID_ENC_BAYER_*8, FID_ENC_MONO16 / BGR16 / RGB16 / BGRA16 /
 * RGBA16 [| FID_ENC_BIGENDIAN], FID_ENC_YUV422): the conversion cv_bridge::toCvCopy(msg, BGR8) + BGR2GRAY is folded into the
 * device's first kernel, so what crosses PCIe is the message's own bytes; + fid_encoding_from_string (round 6). */
#define FID_ABI_VERSION 7

typedef enum fid_status {
    FID_OK = 0,
    FID_E_INVALID_ARG = 1,   /* null pointer, bad size, unsupported encoding */
    FID_E_NO_DEVICE = 2,     /* no HIP device / kernels unavailable: the library never falls back to CPU */
    FID_E_HIP = 3,           /* a HIP runtime call failed (fid_last_error gives the text) */
    FID_E_CAPACITY = 4,      /* an internal or caller buffer was too small; outputs truncated */
    FID_E_OUT_OF_MEMORY = 5,
    FID_E_UNSUPPORTED = 6,   /* parameter combination outside what the kernels implement */
    FID_E_CV_EXCEPTION = 7   /* the reference's OpenCV call throws cv::Exception on this input: imageCallback's catch block logs it
                                and publishes nothing for the frame (aruco_detect.cpp:391-393).  n_per_frame[f] = -1 for such a
                                frame, the other frames of the call are valid.  Only CORNER_REFINE_CONTOUR can raise it (a marker
                                side of fewer than two contour points: cv::solve is handed one equation for two unknowns). */
} fid_status;

typedef enum fid_encoding {  /* sensor_msgs/Image encodings the node accepts via toCvCopy(BGR8) */
    FID_ENC_MONO8 = 0,
    FID_ENC_BGR8 = 1,
    FID_ENC_RGB8 = 2,
    FID_ENC_BGRA8 = 3, /* four bytes per pixel; toCvCopy(BGR8) drops the alpha channel (cvtColor BGRA2BGR / RGBA2BGR) */
    FID_ENC_RGBA8 = 4,
    /* ABI 7: what raw camera drivers publish.  cv_bridge::toCvCopy(msg, "bgr8") (aruco_detect.cpp:348) converts these on the host
     * before detectMarkers turns the BGR8 copy into gray; here both steps are one pass of the device's first kernel over the
     * MESSAGE bytes (a 2.07 MB mosaic crosses the link, not a 6.2 MB BGR8 copy).  The arithmetic is that of fid_image_to_bgr8
     * (below) followed by BGR2GRAY, bit for bit; tests/test_gpu_raw_encodings.py compares the gray tap with exactly that. */
    FID_ENC_BAYER_RGGB8 = 5, /* one byte per pixel: cv_bridge maps the four patterns onto COLOR_BayerBG / RG / GR / GB2BGR, */
    FID_ENC_BAYER_BGGR8 = 6, /* OpenCV's bilinear demosaicing; border columns, then border rows repeat their neighbours     */
    FID_ENC_BAYER_GBRG8 = 7,
    FID_ENC_BAYER_GRBG8 = 8,
    FID_ENC_MONO16 = 9,      /* two bytes per sample: convertTo(8U, 255. / 65535.) = cvRound(float(v) * float(255. / 65535.)) */
    FID_ENC_BGR16 = 10,
    FID_ENC_RGB16 = 11,
    FID_ENC_BGRA16 = 12,     /* (alpha dropped) */
    FID_ENC_RGBA16 = 13,
    FID_ENC_YUV422 = 14,     /* UYVY, two bytes per pixel, even width: cvtColor(COLOR_YUV2BGR_UYVY), BT.601 in 20-bit fixed point */
    FID_ENC_BIGENDIAN = 0x100 /* OR-ed onto a 16-bit encoding: sensor_msgs/Image.is_bigendian (cv_bridge swaps the bytes first) */
} fid_encoding;

/* aruco::DetectorParameters, fields and defaults as set by the node (aruco_detect.cpp:690-727) */
typedef struct fid_params {
    double adaptiveThreshConstant;                 /* 7    */
    int32_t adaptiveThreshWinSizeMin;              /* 3    */
    int32_t adaptiveThreshWinSizeMax;              /* 53   */
    int32_t adaptiveThreshWinSizeStep;             /* 4    */
    int32_t cornerRefinementMethod;                /* 1 = CORNER_REFINE_SUBPIX (node default), 0 = NONE, 2 = CORNER_REFINE_CONTOUR
                                                      (doCornerRefinement && !cornerRefinementSubPix, :274-283, :700-711) */
    int32_t cornerRefinementWinSize;               /* 5    */
    int32_t cornerRefinementMaxIterations;         /* 30   */
    double cornerRefinementMinAccuracy;            /* 0.01 */
    double errorCorrectionRate;                    /* 0.6  */
    double minCornerDistanceRate;                  /* 0.05 */
    int32_t markerBorderBits;                      /* 1    */
    int32_t minDistanceToBorder;                   /* 3    */
    double maxErroneousBitsInBorderRate;           /* 0.04 */
    double minMarkerDistanceRate;                  /* 0.05 */
    double minMarkerPerimeterRate;                 /* 0.1  */
    double maxMarkerPerimeterRate;                 /* 4.0  */
    double minOtsuStdDev;                          /* 5.0  */
    double perspectiveRemoveIgnoredMarginPerCell;  /* 0.13 */
    int32_t perspectiveRemovePixelPerCell;         /* 8    */
    int32_t reserved0;
    double polygonalApproxAccuracyRate;            /* 0.01 */
} fid_params;

/* aruco::Dictionary.  bytes = bytesList data: n_markers x 4 rotations x nbytes, nbytes =
 * (marker_size^2 + 7) / 8, rotation-major per marker (OpenCV >= 4.0 layout). Copied at fid_create. */
typedef struct fid_dict {
    int32_t marker_size;
    int32_t max_correction_bits;
    int32_t n_markers;
    int32_t reserved0;
    const uint8_t *bytes;
} fid_dict;

/* one detected marker: what imageCallback copies into fiducial_msgs/Fiducial (aruco_detect.cpp:366-377) */
typedef struct fid_marker {
    int32_t id;
    float corners[8]; /* x0,y0,x1,y1,x2,y2,x3,y3 */
} fid_marker;

/* per-marker pose record: what poseEstimateCallback puts into fiducial_msgs/FiducialTransform
 * (aruco_detect.cpp:480-497) before the axis-angle -> quaternion step */
typedef struct fid_pose_out {
    double rvec[3];
    double tvec[3];
    double image_error;   /* getReprojectionError: mean squared reprojection error, px^2 (:214-220) */
    double object_error;  /* (image_error / dist(c0,c2)) * (norm(tvec) / fiducial_len)  (:493-495) */
    double fiducial_area; /* calcFiducialArea (:179-200) */
} fid_pose_out;

typedef struct fid_ctx fid_ctx;

/* sizes fixed at creation; 0 selects the default in brackets */
typedef struct fid_limits {
    int32_t max_width;              /* [1920] */
    int32_t max_height;             /* [1080] */
    int32_t max_batch;              /* [1]   frames per fid_detect_batch call */
    int32_t max_starts_per_frame;   /* [262144] border-following start points, all scales */
    int32_t max_contours_per_frame; /* [16384; 65536 when max_batch <= 4, 32768 when <= 16: room for the denser seed grid
                                       of small calls]  probe survivors / tracing seeds / accepted contours, all scales (each) */
    int32_t max_candidates_per_frame; /* [2048] quads leaving _findMarkerContours, all scales */
    int32_t max_markers_per_frame;  /* [256] */
    int32_t max_points_per_frame;   /* [4194304; 16 Mi when max_batch <= 4, 8 Mi when <= 16] contour points kept while borders
                                       are followed, all scales */
} fid_limits;

void fid_default_params(fid_params *p);
void fid_default_limits(fid_limits *l);

fid_status fid_create(const fid_params *params, const fid_dict *dict, const fid_limits *limits /* may be NULL */,
                      int device, fid_ctx **out);
void fid_destroy(fid_ctx *ctx);
/* dynamic_reconfigure path (aruco_detect.cpp:257-298) */
fid_status fid_set_params(fid_ctx *ctx, const fid_params *params);

/* one frame from host memory (the imageCallback path). */
fid_status fid_detect(fid_ctx *ctx, const uint8_t *img, int32_t width, int32_t height, int32_t stride_bytes,
                      fid_encoding enc, fid_marker *out, int32_t cap, int32_t *n);
/* nframes contiguous frames (frame_stride_bytes apart) from host memory; out is nframes x cap_per_frame */
fid_status fid_detect_batch(fid_ctx *ctx, const uint8_t *imgs, int32_t nframes, int32_t width, int32_t height,
                            int32_t stride_bytes, int64_t frame_stride_bytes, fid_encoding enc, fid_marker *out,
                            int32_t cap_per_frame, int32_t *n_per_frame);
/* same, frames already resident in device memory (throughput mode, BASELINE cfg 3) */
fid_status fid_detect_device(fid_ctx *ctx, const void *d_imgs, int32_t nframes, int32_t width, int32_t height,
                             int32_t stride_bytes, int64_t frame_stride_bytes, fid_encoding enc, fid_marker *out,
                             int32_t cap_per_frame, int32_t *n_per_frame);
/* fid_detect_device in two halves, for a caller with a STREAM of batches (the node's frames keep coming: imageCallback,
 * aruco_detect.cpp:332-350, is called once per frame for as long as the camera runs).  fid_submit_device enqueues the whole
 * pipeline of one batch on the context's streams and returns without waiting; fid_collect waits for it and hands out what
 * fid_detect_device would have (fid_pose_last then refers to that batch).  With two or three contexts in turn -- submit k + 1,
 * collect k -- the latency-bound end of one batch runs under the front of the next.  One batch per context at a time:
 * fid_submit_* / fid_detect_* / fid_pose_last / fid_tap_read on a context with a batch in flight return
 * FID_E_INVALID_ARG; d_imgs must stay valid until fid_collect returns. */
fid_status fid_submit_device(fid_ctx *ctx, const void *d_imgs, int32_t nframes, int32_t width, int32_t height,
                             int32_t stride_bytes, int64_t frame_stride_bytes, fid_encoding enc);
/* the same first half for frames in HOST memory (fid_detect_batch's): imgs must stay valid until fid_collect returns; from pinned
 * memory the call returns at once and the copy runs under the kernels of the batch before it */
fid_status fid_submit_batch(fid_ctx *ctx, const uint8_t *imgs, int32_t nframes, int32_t width, int32_t height,
                            int32_t stride_bytes, int64_t frame_stride_bytes, fid_encoding enc);
fid_status fid_collect(fid_ctx *ctx, fid_marker *out, int32_t cap_per_frame, int32_t *n_per_frame);
/* Optional, before fid_submit_device(ctx): that batch's first kernel starts when the batch in flight on `prev` (another context of
 * the same device; NULL or nothing in flight: no order) has its chip-filling kernels behind it -- the fronts of two batches then
 * do not run beside each other, only a front beside the other's latency-bound end; and that batch is laid out as ONE piece
 * (fid_detect_device cuts a batch into two that overlap each other: in a chain the batch on the other context is the other
 * piece).  Holds for the next submit only.  Two contexts in turn, each ordered after the other, is the intended use:
 *     fid_order_after(b, a); fid_submit_device(b, k + 1); fid_collect(a, k); fid_order_after(a, b); fid_submit_device(a, k + 2); ... */
fid_status fid_order_after(fid_ctx *ctx, fid_ctx *prev);

/* poseEstimateCallback arithmetic for n markers.  K row-major 3x3, D = plumb-bob k1,k2,p1,p2,k3
 * (CameraInfo.K / .D[0..4], aruco_detect.cpp:315-323).  len_per_marker[i] is fiducial_len or its per-id
 * override (:241-244).  fiducial_len is the node's ~fiducial_len used in object_error. */
fid_status fid_pose(fid_ctx *ctx, const double K[9], const double D[5], const fid_marker *markers,
                    const double *len_per_marker, int32_t n, double fiducial_len, fid_pose_out *out);

/* Detection and pose of the LAST fid_detect* call fused on the device (no round trip of the corners):
 * poses for frame f start at out[f * cap_per_frame]. */
fid_status fid_pose_last(fid_ctx *ctx, const double K[9], const double D[5], double fiducial_len,
                         fid_pose_out *out, int32_t cap_per_frame);

/* aruco.cpp _refineCandidateLines on its own (what CORNER_REFINE_CONTOUR does to every marker inside fid_detect*): n markers,
 * contour i = points [offsets[i], offsets[i + 1]) of pts_xy (int32 x, y pairs in cv::findContours order, CHAIN_APPROX_NONE;
 * offsets[0] = 0), corners = 8 floats per marker, in: the quad (its corners are contour points), out: the crossings of the
 * four fitted side lines.  status_per_marker[i] = 0, or 1 where the reference throws (the corners are then left as they were;
 * the call returns FID_E_CV_EXCEPTION).  The same device code as inside the pipeline (k_refine_contour). */
fid_status fid_refine_contour_corners(fid_ctx *ctx, const int32_t *pts_xy, const int32_t *offsets, int32_t n, float *corners,
                                      int32_t *status_per_marker);

/* stage taps for parity tests (device -> host copies of intermediate buffers of the last call) */
typedef enum fid_tap {
    FID_TAP_MASKS = 0,       /* uint32 [nframes][nscales][height][words_per_row], bit x&31 of word x>>5 */
    FID_TAP_CANDIDATES = 1,  /* fid_candidate [nframes][max_candidates_per_frame], OpenCV order */
    FID_TAP_FILTERED = 2,    /* fid_candidate after reorder + too-close filter */
    FID_TAP_BITS = 3,        /* uint8 [nframes][max_candidates_per_frame][(ms+2)^2] for FILTERED entries */
    FID_TAP_IDENT = 4,       /* int32 [nframes][max_candidates_per_frame][2] id, rotation */
    FID_TAP_PRESUBPIX = 5,   /* fid_marker [nframes][max_markers_per_frame] */
    FID_TAP_COUNTS = 6,      /* int32 [nframes][12]: starts, contour slots, candidates, filtered, accepted, markers, overflow flags,
                                survivors of the long probe, point chunks (64 points each), survivors of the short probe, 0, 0 */
    FID_TAP_GRAY = 7         /* uint8 [nframes][height][width] the gray image the detector saw */
} fid_tap;

typedef struct fid_candidate {
    int32_t scale;
    int32_t contour_size;
    int32_t start_x, start_y;
    int32_t is_hole;
    float corners[8];
} fid_candidate;

int64_t fid_tap_bytes(fid_ctx *ctx, fid_tap which);
fid_status fid_tap_read(fid_ctx *ctx, fid_tap which, void *dst, int64_t dst_bytes);

/* timing of the last call's kernels, measured with hipEvents on the streams they were launched on (ms, needs
 * FID_PROFILE=1 in the environment at fid_create); names is a static table of nstages strings.  Returns the
 * number of stages. */
int32_t fid_last_stage_ms(fid_ctx *ctx, float *ms, int32_t cap, const char *const **names);
/* how many launches of each kernel the last fid_detect* call made: a large batch is cut into sub-batches that run
 * on separate streams so that the latency-bound tail of one overlaps the bulk of the next (stage times above are
 * summed over these launches) */
int32_t fid_last_launches(fid_ctx *ctx);

/* the HIP stream the context launches on (hipStream_t as void*), for callers that bracket it with events */
void *fid_stream(fid_ctx *ctx);

/* ------------------------------------------------------------------------------------------------------------------
 * STag (stag_detect, BASELINE cfg 5).  Stag::detectMarkers (stag_detect/src/stag/Stag.cpp:24-51) and the pose step of
 * StagNode::imageCallback (stag_detect/src/stag_ros/stag_detect.cpp:110-217), end to end on the device.  Every stage has its
 * own entry point so that it can be checked on its own against the reference's sources (tests/test_gpu_stag.py); each entry
 * point runs the pipeline from the frame up to and including its stage and leaves the results on the device (taps):
 *   fid_stag_edge_frontend             SmoothImage 5x5 + ComputeGradientMapByPrewitt + ComputeAnchorPoints +
 *                                      SortAnchorsByGradValue        ED/ED.cpp:144-187, ImageSmooth.cpp:43-55,
 *                                                                    GradientOperators.cpp:77-136, EDInternals.cpp:50-186
 *   fid_stag_detect_edges              JoinAnchorPointsUsingSortedAnchors (smart routing)   ED/EDInternals.cpp:842-1448
 *   fid_stag_detect_edges_validated    ValidateEdgeSegments                                 ED/ValidateEdgeSegments.cpp:365-413
 *   fid_stag_detect_lines / _validated SplitSegment2Lines, JoinCollinearLines, ValidateLineSegments   ED/EDLines.cpp:114-409
 *   fid_stag_detect_quads              QuadDetector::detectQuads                            QuadDetector.cpp:12-127
 *   fid_stag_detect_markers_unrefined  readCode, Decoder::decode, checkDuplicate            Stag.cpp:57-127
 *   fid_stag_detect_markers            + PoseRefiner::refineMarkerPose = Stag::detectMarkers PoseRefiner.cpp:12-190
 *   fid_stag_pose_last                 Common::solvePnpSingle on centre + 4 corners         common.hpp:34-46
 *   fid_stag_detect_markers_batch      frames over several contexts (host threads inside the library)
 * The constructor mirrors Stag::Stag(int libraryHD, int errorCorrection, bool keepLogs) (include/stag/Stag.h:41); the
 * marker library (the published HDxx codewords) is handed over with fid_stag_load_library like the aruco dictionary. */
typedef struct fid_stag_ctx fid_stag_ctx;
/* LineSegment (stag_detect/include/stag/ED/LineSegment.h:4-14): y = a + b x (invert 0) or x = a + b y (invert 1) */
typedef struct fid_stag_line {
    double a, b;
    double sx, sy, ex, ey;   /* end points */
    int32_t invert;
    int32_t segmentNo;       /* edge segment the line was cut from */
    int32_t firstPixelIndex; /* first pixel of the line inside that segment */
    int32_t len;             /* pixels of the segment that make up the line */
} fid_stag_line;
/* Quad (stag_detect/include/stag/Quad.h:9-26) as QuadDetector::detectQuads leaves it: corners clockwise, the vanishing
 * line and max / min corner distance to it */
typedef struct fid_stag_quad {
    double corners[8]; /* x0 y0 ... x3 y3 */
    double lineInf[3];
    double projectiveDistortion;
} fid_stag_quad;
/* Marker (stag_detect/include/stag/Marker.h:6-15) = Quad + id, after Marker::shiftCorners2 */
typedef struct fid_stag_marker {
    int32_t id;
    int32_t shift;      /* rotation the decoder found (corners are already shifted by it) */
    double corners[8];  /* x0 y0 ... x3 y3, clockwise from the marker's first corner */
    double center[2];
    double H[9];        /* unit square -> image, row-major */
    double lineInf[3];
    double projectiveDistortion;
    uint64_t code;      /* the 48 bits read (Stag::readCode) */
} fid_stag_marker;
/* what StagNode::imageCallback makes of a marker (stag_detect.cpp:140-209): the pose of Common::solvePnpSingle */
typedef struct fid_stag_pose_out {
    int32_t id;
    int32_t reserved;
    double rvec[3], tvec[3];
    double R[9]; /* cv::Rodrigues(rvec), row-major: marker_pose(:, 0:3) */
} fid_stag_pose_out;
fid_status fid_stag_create(int32_t libraryHD, int32_t errorCorrection, int32_t max_width, int32_t max_height, int32_t device,
                           fid_stag_ctx **out);
void fid_stag_destroy(fid_stag_ctx *ctx);
/* gray: host memory, mono8 (what StagNode::imageCallback hands to detectMarkers, stag_detect.cpp:110-131) */
fid_status fid_stag_edge_frontend(fid_stag_ctx *ctx, const uint8_t *gray, int32_t width, int32_t height, int32_t stride_bytes);
/* the front end + the edge routing JoinAnchorPointsUsingSortedAnchors (ED/EDInternals.cpp:842-1448): DoDetectEdgesByED
 * (ED/EDInternals.cpp:2598-2619) complete; the EdgeMap stays on the device.  FID_E_CAPACITY if a scratch array is too small --
 * among them the chain tree of ONE anchor's walk: 32 767 chains, where the reference's Chain::parent / children (short,
 * EDInternals.cpp:39-45) would wrap; frames of uniform noise can get there.  A refused frame leaves no stage readable. */
fid_status fid_stag_detect_edges(fid_stag_ctx *ctx, const uint8_t *gray, int32_t width, int32_t height, int32_t stride_bytes);
/* DetectEdgesByEDPF complete (ED/ED.cpp:144-187): the above + the second smoothing (sigma 1 / 2.5) and ValidateEdgeSegments
 * (ED/ValidateEdgeSegments.cpp:365-413): Helmholtz-principle validation of every segment, invalid pieces cut out */
fid_status fid_stag_detect_edges_validated(fid_stag_ctx *ctx, const uint8_t *gray, int32_t width, int32_t height, int32_t stride_bytes);
/* the above + the line fitting of DetectLinesByEDPF (ED/EDLines.cpp:849-941): SplitSegment2Lines (:162-268) and
 * JoinCollinearLines (:114-156) */
fid_status fid_stag_detect_lines(fid_stag_ctx *ctx, const uint8_t *gray, int32_t width, int32_t height, int32_t stride_bytes);
/* DetectLinesByEDPF complete = what EDInterface::runEDPFandEDLines produces (EDInterface.cpp:13-19): the above +
 * ValidateLineSegments (ED/EDLines.cpp:274-409).  EdgeMap (SEGPIX, VSEGMENTS) and EDLines (VLINES) stay on the device. */
fid_status fid_stag_detect_lines_validated(fid_stag_ctx *ctx, const uint8_t *gray, int32_t width, int32_t height, int32_t stride_bytes);
/* QuadDetector::detectQuads (QuadDetector.cpp:12-66): the above + line groups, corner detection and quad formation */
fid_status fid_stag_detect_quads(fid_stag_ctx *ctx, const uint8_t *gray, int32_t width, int32_t height, int32_t stride_bytes);
/* The tables the STag path makes on the host (functions of the image size alone; no device needed): kmin[n] = smallest number
 * of aligned pixels out of n that validates a line (NFALUT, ED/NFA.cpp:13-44, continued past the LUT size where the reference
 * calls nfa() directly), the 72 sample points of Stag::fillCodeLocations (Stag.cpp:129-277) as [72][3], and MIN_LINE_LEN of
 * DetectLinesByEDPF (ED/EDLines.cpp:694-703, 888-892).  Any output pointer may be NULL. */
fid_status fid_stag_host_tables(int32_t width, int32_t height, int32_t *kmin, int32_t kmin_cap, int32_t *kmin_n, int32_t *lut_size,
                                double *code_locations, int32_t *min_line_len);
/* the marker library of Decoder::Decoder(hd) (Decoder.cpp:14-43): n_codewords = 4 x number of markers, the four rotations
 * of every marker one block after the other, 48 bits each (the HDxx arrays of stag/MarkerIDs.h; fiducials_amd/data/
 * stag_HD<hd>.bin holds them as raw little-endian uint64 for the host sides shipped here) */
fid_status fid_stag_load_library(fid_stag_ctx *ctx, const uint64_t *codewords, int32_t n_codewords);
/* Stag::detectMarkers (Stag.cpp:24-51) up to, not including, PoseRefiner::refineMarkerPose: homography, code reading,
 * decoding with the context's errorCorrection, corner shift, duplicate removal */
fid_status fid_stag_detect_markers_unrefined(fid_stag_ctx *ctx, const uint8_t *gray, int32_t width, int32_t height, int32_t stride_bytes);
/* Stag::detectMarkers complete + getMarkerList() (Stag.h:42-45): the above + PoseRefiner::refineMarkerPose
 * (PoseRefiner.cpp:12-190) on every marker.  out may be NULL (read the MARKERS tap instead); FID_E_CAPACITY if cap is too small */
fid_status fid_stag_detect_markers(fid_stag_ctx *ctx, const uint8_t *gray, int32_t width, int32_t height, int32_t stride_bytes,
                                   fid_stag_marker *out, int32_t cap, int32_t *n_out);
typedef enum fid_stag_tap {
    FID_STAG_TAP_SMOOTH = 0,  /* uint8 [h][w] smoothed image */
    FID_STAG_TAP_GRAD = 1,    /* int16 [h][w] |gx| + |gy| (border: GRADIENT_THRESH - 1) */
    FID_STAG_TAP_DIR = 2,     /* uint8 [h][w] 1 = EDGE_VERTICAL, 2 = EDGE_HORIZONTAL, 0 = below GRADIENT_THRESH */
    FID_STAG_TAP_ANCHORS = 3, /* uint8 [h][w] 254 = ANCHOR_PIXEL */
    FID_STAG_TAP_SORTED = 4,  /* int32 [n_anchors] anchor offsets, ascending gradient (the routing consumes them from the end) */
    /* after fid_stag_detect_edges: */
    FID_STAG_TAP_EDGEIMG = 5,  /* uint8 [h][w] EdgeMap::edgeImg after the routing (255 = EDGE_PIXEL, 254 = anchor never reached) */
    FID_STAG_TAP_SEGMENTS = 6, /* int32 [noSegments][2]: index of the first pixel in SEGPIX, number of pixels (EdgeSegment) */
    FID_STAG_TAP_SEGPIX = 7,   /* int32 [][2]: (r, c) of EdgeMap::pixels, the segments one after the other */
    /* after fid_stag_detect_edges_validated (EDGEIMG then holds the validated edge image): */
    FID_STAG_TAP_SMOOTH2 = 8,   /* uint8 [h][w] image smoothed with sigma 0.4 */
    FID_STAG_TAP_VGRAD = 9,     /* int16 [h][w] Prewitt gradient of SMOOTH2 (0 on the image border) */
    FID_STAG_TAP_VPROB = 10,    /* double [1536] H[g] = P(gradient >= g) */
    FID_STAG_TAP_VSEGMENTS = 11, /* int32 [noSegments][2] validated segments: first pixel in SEGPIX, number of pixels */
    /* after fid_stag_detect_lines: */
    FID_STAG_TAP_LINES = 12,     /* fid_stag_line [noLines] */
    /* after fid_stag_detect_lines_validated: */
    FID_STAG_TAP_VLINES = 13,    /* fid_stag_line [noLines]: EDLines::lines as DetectLinesByEDPF returns them */
    /* after fid_stag_detect_quads (VLINES then carry the corrected line directions): */
    FID_STAG_TAP_QUADS = 14,     /* fid_stag_quad [noQuads]: QuadDetector::getQuads() */
    /* after fid_stag_detect_markers_unrefined: */
    FID_STAG_TAP_MARKERS = 15    /* fid_stag_marker [n]: Stag::markers before PoseRefiner::refineMarkerPose */
} fid_stag_tap;
/* Common::solvePnpSingle (stag_ros/common.hpp:34-46) for every marker of the last fid_stag_detect_markers* call: centre + four
 * corners against (0,0,0), (-h,h,0), (h,h,0), (h,-h,0), (-h,-h,0), h = float(marker_size / 2) (stag_detect.cpp:144-162) */
fid_status fid_stag_pose_last(fid_stag_ctx *ctx, const double K[9], const double D[5], double marker_size, fid_stag_pose_out *out,
                              int32_t cap, int32_t *n_out);
/* throughput mode (BASELINE cfg 5 as a batch): nframes frames, frame_stride_bytes apart, through nctx contexts side by side
 * (one host thread and one HIP stream per context; frame f goes to context f mod nctx).  A frame's work is a chain of small
 * kernels, several frames in flight fill the GPU.  markers / poses: nframes x cap_per_frame; K == NULL or poses == NULL skips
 * the pose step. */
fid_status fid_stag_detect_markers_batch(fid_stag_ctx *const *ctxs, int32_t nctx, const uint8_t *frames, int32_t nframes, int32_t width,
                                         int32_t height, int32_t stride_bytes, int64_t frame_stride_bytes, const double K[9], const double D[5],
                                         double marker_size, fid_stag_marker *markers, fid_stag_pose_out *poses, int32_t cap_per_frame,
                                         int32_t *n_per_frame);
int64_t fid_stag_tap_bytes(fid_stag_ctx *ctx, fid_stag_tap which);
fid_status fid_stag_tap_read(fid_stag_ctx *ctx, fid_stag_tap which, void *dst, int64_t dst_bytes);
/* Frames queued ahead (ABI 6).  A context that has finished a frame sizes the next frame's launches by that frame's counts (plus a
 * margin) and enqueues the whole frame without the nine host waits of the counted road; the true counts are checked at the
 * frame's one wait and a frame that did not fit is run again on the counted road -- the results are the same either way (every
 * kernel reads its counts from device memory).  Staged entry points and a context's first frame always take the counted road, and
 * so do the groups of fid_stag_detect_markers_batch unless FID_STAG_SPEC=1 is set (measured 2 - 4 % slower there: a group's waits
 * run under the other groups' kernels anyway); FID_STAG_SPEC=0 in the environment keeps every frame on it.  queued: frames enqueued that way since fid_stag_create; rerun: how
 * many of them had to be run again. */
fid_status fid_stag_queue_stats(const fid_stag_ctx *ctx, int32_t *queued, int32_t *rerun);

/* ------------------------------------------------------------------------------------------------------------------
 * JPEG ingest.  With the launch file's default `transport:=compressed` (aruco_detect/launch/aruco_detect.launch:6) the frames
 * reach FiducialsNode::imageCallback (aruco_detect.cpp:332) through image_transport's compressed subscriber, i.e. through
 * cv::imdecode = libjpeg(-turbo) with its defaults (JDCT_ISLOW, fancy upsampling, JFIF YCbCr -> RGB).  These entry points do
 * that decode on the device, bit for bit (oracle/jpeg_oracle.c, pinned on libjpeg-turbo's own output): entropy decoding by
 * self-synchronising sub-sequences, the integer IDCT, fancy chroma upsampling, colour conversion, and -- for the detector --
 * the gray image cvtColor(BGR2GRAY) would make of it, without the colour image ever being written.
 * Supported: baseline sequential DCT, 8 bit, Huffman, one or three components in one interleaved scan, luma sampling 1x1 /
 * 2x1 / 2x2 with 1x1 chroma (4:4:4, 4:2:2, 4:2:0), restart intervals.  Anything else: FID_E_UNSUPPORTED (never a wrong image).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct fid_jpeg_ctx fid_jpeg_ctx;
typedef struct fid_jpeg_info {
    int32_t width, height, components;   /* components: 1 or 3 */
    int32_t h_samp, v_samp;              /* luma sampling factors */
    int32_t restart_interval;            /* MCUs per restart interval, 0 = none */
    int32_t blocks_w[3], blocks_h[3];    /* 8 x 8 blocks per component row / column, MCU padded */
    int64_t scan_bytes;                  /* entropy-coded bytes */
} fid_jpeg_info;
typedef enum fid_jpeg_tap {
    FID_JPEG_TAP_COEFS = 0,  /* int16: quantised coefficients, natural order, DC prediction undone; component after component,
                                [blocks_h][blocks_w][64] each (jdhuff.c decode_mcu).  The IDCT consumes them (it zeroes what it has
                                read, so that the next call needs no fill): readable only on a context created with
                                FID_JPEG_KEEP_COEFS=1 in the environment, FID_E_UNSUPPORTED otherwise */
    FID_JPEG_TAP_PLANES = 1  /* uint8: IDCT output, component after component, [blocks_h * 8][blocks_w * 8] (jidctint.c) */
} fid_jpeg_tap;
/* header parse on the host (no device needed) */
fid_status fid_jpeg_probe(const uint8_t *data, int64_t nbytes, fid_jpeg_info *info);
fid_status fid_jpeg_create(int32_t device, int32_t max_width, int32_t max_height, int32_t max_batch, fid_jpeg_ctx **out);
void fid_jpeg_destroy(fid_jpeg_ctx *ctx);
/* n files of ONE image size (any mix of sampling layouts / tables), from host memory.  out_enc: FID_ENC_BGR8 = what
 * cv::imdecode(IMREAD_COLOR) returns, FID_ENC_MONO8 = cvtColor(BGR2GRAY) of that (what aruco::detectMarkers works on).  The
 * result stays on the device (fid_jpeg_device_ptr -> fid_detect_device) and is also copied to host_out if that is not
 * NULL (tightly packed rows, frames host_frame_stride bytes apart). */
fid_status fid_jpeg_decode(fid_jpeg_ctx *ctx, const uint8_t *const *files, const int64_t *nbytes, int32_t n, fid_encoding out_enc,
                           uint8_t *host_out, int64_t host_frame_stride);
const void *fid_jpeg_device_ptr(fid_jpeg_ctx *ctx, int32_t *width, int32_t *height, int32_t *stride_bytes, int64_t *frame_stride_bytes);
int64_t fid_jpeg_tap_bytes(fid_jpeg_ctx *ctx, fid_jpeg_tap which, int32_t frame);
fid_status fid_jpeg_tap_read(fid_jpeg_ctx *ctx, fid_jpeg_tap which, int32_t frame, void *dst, int64_t dst_bytes);
int32_t fid_jpeg_last_rounds(fid_jpeg_ctx *ctx); /* synchronisation rounds the last decode needed (diagnostic) */
const char *fid_jpeg_last_error(fid_jpeg_ctx *ctx);

/* ------------------------------------------------------------------------------------------------------------------
 * Frames that arrive as PNG (compressed_image_transport with format png = cv::imencode(".png") of the camera image).
 * HOST code: the zlib stream is sequential and the reference decodes it on the CPU as well (cv::imdecode in the subscriber
 * plugin, in front of imageCallback, aruco_detect.cpp:332); zlib inflates, this library does the container, the row filters and
 * the conversion.  out_enc FID_ENC_BGR8 = what cv::imdecode(IMREAD_COLOR) returns (alpha dropped, palettes expanded, 1 / 2 / 4
 * bit gray scaled, 16-bit samples cut to their high byte), FID_ENC_MONO8 = cvtColor(BGR2GRAY) of that; out is width * height *
 * (3 | 1) tightly packed bytes, then fid_detect(..., FID_ENC_BGR8 | FID_ENC_MONO8).  No device, no context.  Interlaced files:
 * FID_E_UNSUPPORTED; damaged files: FID_E_INVALID_ARG (fid_png_last_error says what) -- never a wrong image.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct fid_png_info {
    int32_t width, height, bit_depth, color_type, interlace;
    int32_t gray; /* 1: colour types 0 and 4 (every BGR pixel has three equal bytes) */
} fid_png_info;
fid_status fid_png_probe(const uint8_t *data, int64_t nbytes, fid_png_info *info);
fid_status fid_png_decode(const uint8_t *data, int64_t nbytes, fid_encoding out_enc, uint8_t *out, int64_t out_bytes,
                          fid_png_info *info /* may be NULL */);
const char *fid_png_last_error(void); /* of the calling thread */

/* ------------------------------------------------------------------------------------------------------------------
 * The /fiducial_images overlay (aruco_detect.cpp:381-387): imageCallback draws on the BGR8 copy cv_bridge made of the frame
 * (aruco::drawDetectedMarkers(cv_ptr->image, corners, ids)) and publishes it when ~publish_images is set.  HOST code, like the
 * reference's.  fid_to_bgr = that copy (toCvCopy(msg, BGR8): gray replicated, RGB swapped, alpha dropped; tightly packed rows);
 * fid_draw_detected_markers = the four sides of every marker, cv::line(..., Scalar(0, 255, 0), 1, LINE_8) restated exactly
 * (LineIterator, clipLine, Point2f -> Point by cvRound).  NOT drawn, because OpenCV's anti-aliasing and Hershey font tables are
 * third-party data absent from the reference tree and from this machine: the LINE_AA square on the first corner and the "id=<n>"
 * text; aruco::drawAxis (:431) likewise.  FID_DRAW_FIRST_CORNER_LINE8 adds that square with LINE_8 sides -- a cue for a human
 * viewer, not the reference's pixels.
 * ------------------------------------------------------------------------------------------------------------------ */
/* fid_image_to_bgr8 (ABI 6) = cv_bridge::toCvCopy(msg, "bgr8") by the message's encoding STRING, for what a raw camera driver
 * publishes beside the five encodings fid_detect takes itself: mono16 / bgr16 / rgb16 / bgra16 / rgba16 (layout, then
 * convertTo(8U, 255. / 65535.), is_bigendian honoured), bayer_rggb8 / bayer_bggr8 / bayer_gbrg8 / bayer_grbg8 (OpenCV's bilinear
 * demosaicing under cv_bridge's pattern mapping) and yuv422 (UYVY, BT.601 fixed point); the five 8-bit encodings go through fid_to_bgr.  The node converts such a frame
 * with this call and hands the BGR8 copy to fid_detect(FID_ENC_BGR8), which is the order the reference works in
 * (aruco_detect.cpp:348-350).  FID_E_UNSUPPORTED: an encoding that is not restated here (16-bit Bayer, float images, ...) -- the node
 * reports it like the cv_bridge exception it would catch (:389-391).  Restated from the published sources: parity unpinned. */
fid_status fid_image_to_bgr8(const uint8_t *img, int32_t width, int32_t height, int32_t stride_bytes, const char *encoding,
                             int32_t is_bigendian, uint8_t *out_bgr, int64_t out_bytes);
/* sensor_msgs/Image.encoding (+ is_bigendian) -> the fid_encoding fid_detect takes (ABI 7): every string fid_image_to_bgr8 converts.
 * FID_E_UNSUPPORTED for anything else (16-bit Bayer, float images, ...), which the node reports like the cv_bridge exception the
 * reference catches (aruco_detect.cpp:389-391).  bytes_per_pixel (may be NULL): what one pixel occupies in a row, for the
 * caller's step * height check. */
fid_status fid_encoding_from_string(const char *encoding, int32_t is_bigendian, fid_encoding *out_enc, int32_t *bytes_per_pixel);
#define FID_DRAW_FIRST_CORNER_LINE8 1u
fid_status fid_to_bgr(const uint8_t *img, int32_t width, int32_t height, int32_t stride_bytes, fid_encoding enc, uint8_t *out_bgr,
                       int64_t out_bytes);
fid_status fid_draw_detected_markers(uint8_t *bgr, int32_t width, int32_t height, int32_t stride_bytes, const fid_marker *markers,
                                     int32_t n, uint32_t flags);

/* ------------------------------------------------------------------------------------------------------------------
 * aruco::getPredefinedDictionary(dicno) (aruco_detect.cpp:671, ~dictionary :611) from a table file the DEPLOYER has.  OpenCV's
 * tables are third-party data that ship neither with the reference nor with this repository (fiducials_amd/data/ holds the
 * codewords the reference's fixtures pin + labelled fillers); a caller that links OpenCV passes Dictionary::bytesList to
 * fid_create directly, one that does not loads it here.  path =
 *   OpenCV's modules/aruco/src/predefined_dictionaries.hpp as text (the array DICT_<N>X<N>_1000_BYTES / DICT_ARUCO_BYTES of
 *     dicno's family is parsed: per marker four rotations x ceil(n^2 / 8) bytes = bytesList; the first n_markers(dicno) rows);
 *   a cv::FileStorage YAML as aruco::Dictionary::writeDictionary writes it (nmarkers, markersize, maxCorrectionBits,
 *     marker_<i>: "<bits>"); dicno -1 = a custom dictionary, sizes as the file says;
 *   this repository's dict_*.txt.
 * dicno: the node's ~dictionary enum 0..16 (marker size, count and maxCorrectionBits as dictionary.cpp's predefined objects).
 * bytes / bytes_cap: the caller's buffer for bytesList; *out points into it.  FID_E_CAPACITY (out->n_markers and marker_size
 * set, out->bytes NULL) when it is too small: n_markers * 4 * ((marker_size^2 + 7) / 8) bytes are needed.  Host code.
 * ------------------------------------------------------------------------------------------------------------------ */
fid_status fid_dict_load_file(const char *path, int32_t dicno, uint8_t *bytes, int64_t bytes_cap, fid_dict *out);
const char *fid_dict_last_error(void); /* of the calling thread */

const char *fid_strerror(fid_status s);
const char *fid_last_error(fid_ctx *ctx);
int32_t fid_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FID_ABI_H */
