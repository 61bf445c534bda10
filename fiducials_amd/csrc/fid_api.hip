// fid_api.hip -- the C-ABI of libfid_amd.so (include/fid_abi.h): context, buffers in HBM, the launch
// sequence of the detection pipeline on one HIP stream, result hand-back.  No CPU fallback: without a
// HIP device fid_create fails with FID_E_NO_DEVICE.
#include "fid_kernels.hip"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <string>
#include <vector>

namespace {

enum { ST_GRAY = 0, ST_THRESH, ST_STARTS, ST_PROBE, ST_WALK, ST_APPROX, ST_SORT, ST_NEAR, ST_RESOLVE, ST_IDENT, ST_FILTER, ST_SUBPIX, ST_POSE, ST_SEEDWALK,
       ST_SEEDLESS, ST_COUNT };
// (trace mode 2: walk_probe / walk_full / approx are the main stream's seed walk / link + cycles + copy / approx of the seed
//  cycles; seedless_chain is the auxiliary stream's probes + survivor walk + cycles + copy + approx beside them)
const char *const kStageNames[ST_COUNT] = {"to_gray", "threshold", "find_starts", "walk_probe", "walk_full", "approx", "sort_cands",
                                           "near", "resolve", "identify", "filter_markers", "subpix", "pose", "seed_walk", "seedless_chain"};

constexpr int TX = 128, TY = 32, NT = 256;

}  // namespace

struct fid_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    enum { MAX_SUB = 8 };
    hipStream_t sub_stream[MAX_SUB] = {};  // sub-batches of one call run on these, overlapping each other's tails
    int sub_prio[MAX_SUB] = {};
    hipStream_t aux_stream[MAX_SUB] = {};  // per sub-batch: the seed walk runs here, beside the probe passes and the survivor walk
    hipStream_t idx_stream = nullptr;      // calls laid out as ONE piece (a frame or two, a batch of a chain): k_seed_index on a stream of its own
    hipEvent_t aux_fork[MAX_SUB] = {}, aux_join[MAX_SUB] = {}, aux_idx[MAX_SUB] = {};
    hipEvent_t sub_done[MAX_SUB] = {}, fork_ev = nullptr;
    hipStream_t copy_stream = nullptr;     // fid_detect_batch: the frames go up sub-batch by sub-batch on this stream ...
    hipEvent_t in_ready[MAX_SUB] = {};     // ... and a sub-batch starts when its frames have landed (H2D of k + 1 under the compute of k)
    bool host_feed = false;                // this run_detect call is fed that way
    struct Pending {                       // the batch fid_submit_device enqueued and fid_collect has not yet fetched
        const uint8_t *d_src;
        int F, W, H, stride;
        long long fstride;
        fid_encoding enc;
    } pend = {};
    bool in_flight = false;
    // batches in turn on several contexts (fid_order_after): tail_ev is recorded where the LAST sub-batch of a batch has its
    // chip-filling kernels behind it (chain_at: 0 after find_starts, 1 seed walk, 2 copy, 3 approxPolyDP, 4 in front of the
    // candidate sort, 5 after the too-close filter); the next batch's first kernel, on another context, waits for wait_ev
    hipEvent_t tail_ev = nullptr, wait_ev = nullptr;
    // host-fed batches in turn: the copy of this context's next batch starts when the copy of the batch in flight on the other
    // context has landed (wait_copy_ev = that context's in_ready event).  Two copies side by side share the link, both take
    // twice as long, and the two contexts then run in lockstep: 13.8 k frames/s from pinned memory instead of 21 k.
    hipEvent_t wait_copy_ev = nullptr;
    bool fed_from_host = false;  // the batch in flight came through feed_and_enqueue
    int chain_at = 0;
    bool chained = false;  // the next submit is one of a chain of batches (fid_order_after): one sub-batch, see plan_sub_batches
    hipEvent_t walk_done[MAX_SUB] = {}, fs_done[MAX_SUB] = {}, front_done[MAX_SUB] = {};
    // ONE blocking call laid out like the batches in turn of two contexts (round 5): `pieces` pieces, piece k's front (threshold ..
    // approxPolyDP) on main stream k & 1, its seedless chain AND its candidate tail (sort .. pose) on auxiliary stream k & 1; piece
    // k + 1's threshold starts when piece k's find_starts is through -- four streams in all (the runtime's four hardware queues),
    // the latency-bound tail of a piece under the fronts of the pieces behind it.  0 / 1: the two sub-batches side by side.
    int pieces = 0;
    bool piece_chain = false;  // (this call is laid out that way)
    bool from_host_call = false;  // enqueue_detect is running under feed_and_enqueue: the sub-batches are the copy's pieces, never a piece chain
    int fs_barrier = 0;                    // FID_FS_BARRIER=1: the walks of every sub-batch wait for all find_starts (measured: find_starts
                                           // 3.6 -> 2.2 ms, the seed walks 3.3 -> 4.9 ms now side by side: the step is the same)     // a sub-batch has left its contour stage (staggered starts, FID_STAGGER)
    int stagger = 0;                        // sub-batch k starts when sub-batch k - stagger has left its contour stage (0: all at once)
    int resolve_lds_kb = 64;
    int walk2_div = 6;   // FID_WALK2_DIV: the survivor walk gets walk_blocks / this workgroups a frame (round 6: 2 -> 6, one workgroup a frame at 128 frames: 1.5 k
                         // survivors over 4 waves instead of 12 -- 1.13 M -> 0.70 M VALU wave-instructions a frame, lane utilisation 0.24 -> 0.33, same frame rate)
    int probe_refill0 = 0;  // FID_PROBE_REFILL0: the same for the first pass (0: k_probe_lut<6, 0>)
    int probe_refill = 4;  // FID_PROBE_REFILL: workgroups per frame of the refilled second probe pass (0: k_probe_lut<32, 1>)
    hipEvent_t sub_ev[MAX_SUB][20] = {};   // per sub-batch stage boundaries (FID_PROFILE)
    int sub_frames = 0;                    // frames per sub-batch (0 = automatic)
    fid_params params;
    fid_limits lim;
    DevParams P;
    std::vector<uint8_t> dict_host;
    int dict_ms = 0, dict_maxc = 0, dict_n = 0;
    // device buffers
    uint8_t *d_in = nullptr;
    size_t d_in_bytes = 0;
    uint8_t *d_gray = nullptr;
    uint32_t *d_masks = nullptr;
    size_t masks_bytes = 0;
    int masks_W = 0, masks_H = 0, masks_S = 0;
    uint2 *d_starts = nullptr, *d_surv1 = nullptr, *d_surv = nullptr;
    uint32_t *d_pool = nullptr;
    // seed-accelerated tracing
    DevSeg *d_segs = nullptr;
    DevPend *d_pend = nullptr;
    uint2 *d_seedq = nullptr;
    unsigned long long *d_seedhash = nullptr;  // per frame: seed state -> seed index (SeedHash, fid_kernels.hip)
    int seed_hash_cap = 0, seed_gen = 0;
    uint4 *d_wres = nullptr, *d_cinfo = nullptr;
    uint32_t *d_cbase = nullptr, *d_dense = nullptr;
    uint4 *d_recs = nullptr;  // copy records: pieces of the accepted contours
    int thr_mode = 1;    // node window table: 1 = k_threshold_stream (default), 0 (FID_THR=tile) = k_threshold_fixed
    // The probe survivors' walk.  Default: k_walk_full<2> (8 KB of LDS per wave).  FID_SURV_WALK=new: k_seed_walk<true> (round 5:
    // the seed walker's toroidal prefetched windows, 16 KB per wave) -- 1.21 -> 1.00 M VALU wave-instructions per frame at 0.30
    // instead of 0.23 of the lanes, the stage 9 % shorter, identical results; but on the two-context bench 33.2 - 33.9 k frames/s
    // against 33.9 - 34.3 k (five interleaved rounds; with half the waves 33.7 k): what it takes of the CUs' LDS costs the other
    // batch's kernels more than its instructions save.  Kept as an option and in the parity tests.
    int surv_walk_old = 1;
    int surv_blocks_x = 0;  // FID_SURV_BLOCKS_X: workgroups of k_seed_walk<true> = x times k_walk_full<2>'s waves (0: half of them)
    int thr_xcd = 1;  // strips dealt out so that an XCD works through neighbouring strips (FID_THR_XCD=0: plain grid order)
    int thr_nw = 3, thr_split = 0, thr_rows = 0;  // stream kernel: consumer waves per workgroup, un-fused LDS reads, rows per workgroup (0 = automatic)
    int trace_mode = 2;  // 2: cycle tracing (borders read off the seed cycles; starts only for borders without a seed);
                         // 1 (FID_TRACE=chain): round-2 seed tracing (every border found by a probe survivor); 0 (FID_TRACE=legacy):
                         // probe passes + whole-border walk only
    int sw_blocks = 0;   // FID_SW_BLOCKS: seed-walker workgroups per frame (0 = automatic)
    int probe_lut = 1;   // table-driven probe passes (FID_PROBE_LUT=0: the arithmetic ones)
    int tail_grid = 4096;  // FID_TAIL_GRID: workgroups of k_identify (x 4: k_subpix) -- 1024 left 5 k candidates five rounds of an 80 us chain: +5 %
    int filter_lds = 256;  // markers k_filter_markers keeps in LDS (FID_FILTER_LDS; more go through the global scratch)
    int light_x = 1;     // FID_LIGHT_X: grid multiplier of k_near / k_seg_cycles, 1024 threads for k_sort_cands
    long long fallbacks = 0;  // calls that fell back to the whole-border walk because the seed table was too small
    int max_chunks = 0;
    int walk_blocks = 0;  // one-wave workgroups per frame in the full walk pass (0 = automatic)
    int walk_blocks_cap = 0;   // FID_WALK_CAP: most walker workgroups per frame (0: 64, 256 for calls of one or two frames)
    int copy_blocks = 0;
    int resolve_serial = 0;    // FID_RESOLVE_SERIAL=1: k_resolve's single-wave path even when the near triangle fits LDS (tests)
    int resolve_reg_max = 64;  // FID_RESOLVE_REG_MAX: components of up to so many candidates are resolved in registers (0: none; tests)
    int seed_shift = 0;        // FID_SEED_SHIFT: force the seed grid spacing 8 << shift (0 = by call size)
    uint4 *d_contours = nullptr;
    uint32_t *d_ckpts = nullptr;
    size_t ckpts_elems = 0;
    DevCand *d_cands = nullptr, *d_sorted = nullptr, *d_filtered = nullptr;
    float4 *d_cmeta = nullptr;
    uint32_t *d_near = nullptr;
    DevIdent *d_ident = nullptr;
    fid_marker *d_pre = nullptr, *d_markers = nullptr, *d_filter_scratch = nullptr;
    int *d_accsrc = nullptr, *d_mksrc = nullptr;  // the filtered candidate behind every identified / kept marker (k_filter_markers -> k_refine_contour)
    fid_pose_out *d_poses = nullptr;
    DevCounts *d_counts = nullptr;
    DevGlobal *d_global = nullptr;
    unsigned *d_worklist = nullptr, *d_nwork = nullptr;
    uint8_t *d_dict = nullptr;
    float *d_subpix_mask = nullptr;
    uint8_t *d_probe_tables = nullptr;  // k_probe_lut's forward and backward step tables (4 KB, built once: k_probe_tables)
    double *d_lens = nullptr;
    fid_marker *d_pose_in = nullptr;
    int *d_pose_n = nullptr;
    int pose_cap = 0;
    // what a call hands back -- global flags, per-frame counters, markers, poses -- lies in ONE device block and ONE pinned host
    // block, laid out for the call's frame count (layout_results): one fill at the start of a call and one copy at its end instead
    // of three fills and three or four copies (each a ~5 us kernel on the stream of a 0.8 ms single-frame call)
    uint8_t *d_res = nullptr, *h_res = nullptr;
    size_t res_clear_bytes = 0, res_markers_end = 0, res_poses_end = 0, res_poses_off = 0;
    bool blocking = false;        // the call in progress is fid_detect*: nobody can chain behind it (no tail_ev on its stream)
    bool res_precleared = false;  // feed_and_enqueue has put this call's clear of d_res in front of its host -> device copy
    // pinned host staging
    fid_marker *h_markers = nullptr;
    DevCounts *h_counts = nullptr;
    DevGlobal *h_global = nullptr;
    fid_pose_out *h_poses = nullptr;
    // the camera of the last fid_pose_last call: the next fid_detect_* runs k_pose for it at the end of every sub-batch's stream
    // (under the other sub-batch's tail instead of after everything, and without a second host round trip); fid_pose_last with the
    // same camera then only hands the results over.  Same kernel, same arithmetic, same results.
    bool pose_cam_valid = false, pose_done = false;
    double pose_K[9] = {}, pose_D[5] = {}, pose_len = 0.;
    // last call
    int last_frames = 0, last_W = 0, last_H = 0, last_nsub = 1;
    const uint8_t *last_gray = nullptr;
    long long last_gfstride = 0;
    bool profile = false;
    hipEvent_t ev[ST_COUNT + 1] = {};
    bool ev_valid[ST_COUNT + 1] = {};
    float stage_ms[ST_COUNT] = {};
    std::string last_error;
};

namespace {

#define HIPCHK(ctx, expr)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(e_);                \
            return e_ == hipErrorOutOfMemory ? FID_E_OUT_OF_MEMORY : FID_E_HIP;                   \
        }                                                                                         \
    } while (0)

template <typename T>
fid_status dalloc(fid_ctx *c, T **p, size_t count)
{
    HIPCHK(c, hipMalloc((void **)p, count * sizeof(T)));
    return FID_OK;
}

int roundup(int v, int m) { return (v + m - 1) / m * m; }

size_t results_bytes(int F, int MM)
{
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    return up(sizeof(DevGlobal)) + up(sizeof(unsigned) * fid_ctx::MAX_SUB) + up(sizeof(DevCounts) * (size_t)F) + up(sizeof(fid_marker) * (size_t)F * MM) +
           up(sizeof(fid_pose_out) * (size_t)F * MM);
}
// [DevGlobal][nwork][DevCounts x F] (cleared at the start of a call) [fid_marker x F x MM][fid_pose_out x F x MM]
void layout_results(fid_ctx *c, int F)
{
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const int MM = c->lim.max_markers_per_frame;
    size_t off = 0;
    c->d_global = (DevGlobal *)(c->d_res + off);
    c->h_global = (DevGlobal *)(c->h_res + off);
    off += up(sizeof(DevGlobal));
    c->d_nwork = (unsigned *)(c->d_res + off);
    off += up(sizeof(unsigned) * fid_ctx::MAX_SUB);
    c->d_counts = (DevCounts *)(c->d_res + off);
    c->h_counts = (DevCounts *)(c->h_res + off);
    off += up(sizeof(DevCounts) * (size_t)F);
    c->res_clear_bytes = off;
    c->d_markers = (fid_marker *)(c->d_res + off);
    c->h_markers = (fid_marker *)(c->h_res + off);
    off += up(sizeof(fid_marker) * (size_t)F * MM);
    c->res_markers_end = off;
    c->res_poses_off = off;
    c->d_poses = (fid_pose_out *)(c->d_res + off);
    c->h_poses = (fid_pose_out *)(c->h_res + off);
    off += up(sizeof(fid_pose_out) * (size_t)F * MM);
    c->res_poses_end = off;
}

template <int NW, bool SPLIT>
void launch_thr_stream(dim3 grid, hipStream_t st, const uint8_t *g, long long gfstride, uint32_t *masks, const DevParams &P, int RS, int xcd_map)
{
    using S = ThrStream<3, 4, 13, NW>;
    fid_launch_log("k_threshold_stream", S::NT, S::LDS_BYTES);
    k_threshold_stream<3, 4, 13, NW, SPLIT><<<grid, dim3(S::NT), S::LDS_BYTES, st>>>(g, gfstride, masks, P, RS, xcd_map);
}

fid_status apply_params(fid_ctx *c, const fid_params *p)
{
    if (p->adaptiveThreshWinSizeMin < 3 || p->adaptiveThreshWinSizeMax < p->adaptiveThreshWinSizeMin ||
        p->adaptiveThreshWinSizeStep <= 0)
        return FID_E_INVALID_ARG;
    if (!(p->minMarkerPerimeterRate > 0 && p->maxMarkerPerimeterRate > 0 && p->polygonalApproxAccuracyRate > 0 &&
          p->minCornerDistanceRate >= 0 && p->minDistanceToBorder >= 0 && p->minMarkerDistanceRate >= 0))
        return FID_E_INVALID_ARG;  // the CV_Assert set of _findMarkerContours / _filterTooCloseCandidates
    if (p->markerBorderBits <= 0) return FID_E_INVALID_ARG;
    int nsc = (p->adaptiveThreshWinSizeMax - p->adaptiveThreshWinSizeMin) / p->adaptiveThreshWinSizeStep + 1;
    if (nsc > FID_MAX_SCALES) return FID_E_UNSUPPORTED;
    DevParams &P = c->P;
    P.nscales = nsc;
    P.rmax = 0;
    for (int i = 0; i < nsc; i++) {
        int w = p->adaptiveThreshWinSizeMin + i * p->adaptiveThreshWinSizeStep;
        if (w % 2 == 0) w++;
        P.win[i] = w;
        if (w / 2 > P.rmax) P.rmax = w / 2;
    }
    if (P.rmax > 40) return FID_E_UNSUPPORTED;  // LDS tile budget of k_threshold
    {
        // adaptiveThreshold: idelta = type == THRESH_BINARY ? cvCeil(delta) : cvFloor(delta); aruco passes THRESH_BINARY_INV
        double cdelta = p->adaptiveThreshConstant;
        int i = (int)cdelta;
        P.idelta = i - (i > cdelta);  // cvFloor
    }
    P.polyAcc = p->polygonalApproxAccuracyRate;
    P.minCornerDistRate = p->minCornerDistanceRate;
    P.minMarkerDistRate = p->minMarkerDistanceRate;
    P.minDistToBorder = p->minDistanceToBorder;
    P.markerSize = c->dict_ms;
    P.borderBits = p->markerBorderBits;
    P.cellSize = p->perspectiveRemovePixelPerCell;
    P.cellMargin = (int)(p->perspectiveRemoveIgnoredMarginPerCell * p->perspectiveRemovePixelPerCell);
    if (P.markerSize + 2 * P.borderBits > FID_MAX_CELLS) return FID_E_UNSUPPORTED;
    if (P.cellSize < 1 || P.cellSize - 2 * P.cellMargin < 1 || (P.markerSize + 2 * P.borderBits) * P.cellSize > 160) return FID_E_UNSUPPORTED;
    P.minOtsuStdDev = p->minOtsuStdDev;
    P.maxBorderErr = (int)(c->dict_ms * c->dict_ms * p->maxErroneousBitsInBorderRate);
    P.maxCorr = (int)((double)c->dict_maxc * p->errorCorrectionRate);
    P.nMarkers = c->dict_n;
    P.nbytes = (c->dict_ms * c->dict_ms + 7) / 8;
    // 0 CORNER_REFINE_NONE, 1 CORNER_REFINE_SUBPIX, 2 CORNER_REFINE_CONTOUR (the node reaches all three: aruco_detect.cpp:274-283,
    // 700-711).  3 = CORNER_REFINE_APRILTAG replaces the whole candidate search and is not something the node can select.
    if (p->cornerRefinementMethod < 0 || p->cornerRefinementMethod > 2) return FID_E_UNSUPPORTED;
    P.refine = p->cornerRefinementMethod;
    P.subpixWin = p->cornerRefinementWinSize;
    if (P.refine == 1 && (P.subpixWin < 1 || P.subpixWin > SP_MAXWIN || p->cornerRefinementMaxIterations < 1 ||
                     !(p->cornerRefinementMinAccuracy > 0)))
        return FID_E_INVALID_ARG;
    {
        int mi = p->cornerRefinementMaxIterations;
        P.subpixMaxIter = mi < 1 ? 1 : (mi > 100 ? 100 : mi);
        double e = p->cornerRefinementMinAccuracy > 0 ? p->cornerRefinementMinAccuracy : 0.;
        P.subpixEps = e * e;
    }
    c->params = *p;
    // cornerSubPix weight mask, computed once on the host exactly as cornersubpix.cpp does (float expf)
    if (P.refine == 1) {
        int win = P.subpixWin, ww = 2 * win + 1;
        std::vector<float> mask((size_t)ww * ww);
        for (int i = 0; i < ww; i++) {
            float y = (float)(i - win) / win;
            float vy = expf(-y * y);
            for (int j = 0; j < ww; j++) {
                float x = (float)(j - win) / win;
                mask[(size_t)i * ww + j] = (float)(vy * expf(-x * x));
            }
        }
        HIPCHK(c, hipMemcpy(c->d_subpix_mask, mask.data(), mask.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return FID_OK;
}

void set_geometry(fid_ctx *c, int W, int H, int gstride, int F)
{
    DevParams &P = c->P;
    P.W = W;
    P.H = H;
    P.gstride = gstride;
    P.WW = (W + 31) / 32;
    P.TC = MASK_PADW + roundup(W, 128) / 32 + 1;  // one zero tile column left and right (threshold tiles are 128 px wide)
    P.TR = (H + 2 + MT_ROWS - 1) / MT_ROWS + 1;  // + one zero tile row: a 2 x 2 tile window always exists
    P.nframes = F;
    int maxdim = W > H ? W : H;
    P.minPerim = (int)(unsigned int)(c->params.minMarkerPerimeterRate * maxdim);
    P.maxPerim = (int)(unsigned int)(c->params.maxMarkerPerimeterRate * maxdim);
    P.maxStarts = c->lim.max_starts_per_frame;
    P.maxContours = c->lim.max_contours_per_frame;
    P.maxCands = c->lim.max_candidates_per_frame;
    P.maxMarkers = c->lim.max_markers_per_frame;
    P.maxChunks = c->max_chunks;
    // tracing seeds: the denser the grid, the shorter the longest seed-free stretch (the latency of a single frame) and the
    // more segments (tables, link / chain work).  Small calls take the densest grid their tables have room for.
    {
        int sh = 4;  // 128 px: batches (measured on the 256-frame bench batch: 64 px and 256 px are both slower)
        if (F <= 16 && P.maxContours >= 32768 && c->max_chunks >= 2 * 65536) sh = 3;  // 64 px
        if (F <= 2 && P.maxContours >= 65536 && c->max_chunks >= 4 * 65536) sh = 2;    // 32 px (one frame: 0.98 ms against 1.00 at 64 px)
        if (c->seed_shift > 0) sh = c->seed_shift;
        P.seedShift = sh < SEED_SHIFT_MIN ? SEED_SHIFT_MIN : (sh > SEED_SHIFT_MAX ? SEED_SHIFT_MAX : sh);
        P.seedHashCap = c->seed_hash_cap;
    }
}

size_t masks_elems(const fid_ctx *c, int W, int H, int F)
{
    int TC = MASK_PADW + roundup(W, 128) / 32 + 1, TR = (H + 2 + MT_ROWS - 1) / MT_ROWS + 1;
    return (size_t)F * c->P.nscales * TR * TC * MT_ROWS;
}

// the streams of sub-batches 0 .. nsub - 1 (and their auxiliary streams in the traced modes), made on first use
fid_status ensure_streams(fid_ctx *c, int nsub, int F)
{
    if (c->piece_chain) nsub = 2;  // (piece k works on main stream k & 1 and auxiliary stream k & 1)
    for (int sb = 0; sb < nsub && sb < fid_ctx::MAX_SUB; sb++) {
        // (sub-batch 0 runs on the context's own stream, which has nothing else to do while a call is under way: a resident batch
        //  then works on four streams -- two sub-batches, two auxiliary -- which is what the HIP runtime's DEFAULT of four hardware
        //  queues carries side by side; with a fifth and sixth stream alive the default cost 17 % against GPU_MAX_HW_QUEUES >= 8)
        if (sb == 0) c->sub_stream[0] = c->stream;
        else if (!c->sub_stream[sb] && nsub > 1) HIPCHK(c, hipStreamCreateWithPriority(&c->sub_stream[sb], hipStreamNonBlocking, c->sub_prio[sb]));
        if (!c->aux_stream[sb] && c->trace_mode >= 1) HIPCHK(c, hipStreamCreateWithPriority(&c->aux_stream[sb], hipStreamNonBlocking, c->sub_prio[sb]));
    }
    // one piece = two streams so far: a third one still fits the runtime's default of four hardware queues
    // (only for calls of a few frames -- the node's shape: a 256-frame batch of a chain is one piece too, but two contexts in turn
    //  would then hold six streams, and more than four live streams cost the batch 17 % in round 3)
    if (nsub == 1 && F <= 16 && c->trace_mode == 2 && !c->idx_stream && !getenv("FID_NO_IDX_STREAM")) HIPCHK(c, hipStreamCreateWithFlags(&c->idx_stream, hipStreamNonBlocking));
    return FID_OK;
}

// how a call of F frames is cut into sub-batches (run_detect; fid_detect_batch sends the frames up in the same pieces):
// sub-batch sb = frames [f0[sb], f0[sb + 1])
struct SubPlan {
    int nsub;
    int f0[fid_ctx::MAX_SUB + 1];
};
SubPlan plan_sub_batches(const fid_ctx *c, int F)
{
    SubPlan pl;
    if (c->chained && c->sub_frames <= 0 && !getenv("FID_SUB_SHARES")) {
        // a batch in a chain of batches on two contexts (fid_order_after) IS the other half: the batch before it on the other
        // context plays the part of the first sub-batch, for ever, with no call boundary at which the last piece's latency-bound
        // end has the chip to itself.  Measured, 256 frames a batch, two contexts: whole batches 30.5 k frames/s, 64 % + 36 %
        // pieces 29.5 k (and 26.5 k for one context, one call after the other).
        pl.nsub = 1;
        pl.f0[0] = 0;
        pl.f0[1] = F;
        return pl;
    }
    if (c->sub_frames <= 0 && c->host_feed && F >= 64) {
        // frames that come up from the host while the call runs (the link, ~45 GB/s under load, is slower than the kernels):
        // four pieces -- the first kernels start after a fifth of the copy, the copy engine never idles, and what is left to
        // compute when the last byte has landed is a small piece.  The rate is insensitive to the split: the step is the arrival
        // of the first piece plus the kernels' time at the lower efficiency of small sub-batches
        int share[4] = {20, 30, 30, 20};  // (measured 15.7 k frames/s; 25-25-25-25: 15.3 k, 34-28-22-16: 15.4 k, 10-22-34-34: 15.0 k)
        if (const char *e = getenv("FID_FEED_SHARES")) (void)sscanf(e, "%d,%d,%d,%d", &share[0], &share[1], &share[2], &share[3]);
        pl.nsub = 4;
        int acc = 0;
        for (int k = 0; k < 4; k++) {
            pl.f0[k] = acc;
            acc += k == 3 ? F - acc : (F * share[k] + 50) / 100;
        }
        pl.f0[4] = F;
        return pl;
    }
    if (c->piece_chain) {
        pl.nsub = c->pieces < fid_ctx::MAX_SUB ? c->pieces : fid_ctx::MAX_SUB;
        for (int k = 0; k <= pl.nsub; k++) pl.f0[k] = (int)((long long)F * k / pl.nsub);
        return pl;
    }
    // resident frames: two halves measured best (more streams fight for CUs)
    int per = c->sub_frames > 0 ? c->sub_frames : (F >= 32 ? (F + 1) / 2 : F);
    int nsub = (F + per - 1) / per;
    if (nsub > fid_ctx::MAX_SUB) {
        nsub = fid_ctx::MAX_SUB;
        per = (F + nsub - 1) / nsub;
        nsub = (F + per - 1) / per;
    }
    pl.nsub = nsub;
    for (int k = 0; k <= nsub; k++) pl.f0[k] = k * per < F ? k * per : F;
    if (c->sub_frames <= 0 && F >= 32) {
        // the sub-batches do not run in step: the first one's kernels meet less of the others' (its find_starts runs beside a
        // threshold, the second one's beside a seed walk and the probes), so with equal pieces it is through ~2 ms earlier and
        // the last one's latency-bound tail has the chip to itself.  Decreasing pieces even that out (FID_SUB_SHARES, per cent).
        int sh[fid_ctx::MAX_SUB], n = 0, sum = 0;
        const char *e = getenv("FID_SUB_SHARES");
        const char *txt = e ? e : "64,36";
        for (const char *q = txt; *q && n < fid_ctx::MAX_SUB;) {
            const int v = atoi(q);
            if (v > 0) {
                sh[n++] = v;
                sum += v;
            }
            while (*q && *q != ',') q++;
            if (*q == ',') q++;
        }
        if (n >= 1 && sum > 0) {
            pl.nsub = n;
            int acc = 0;
            for (int k = 0; k < n; k++) {
                pl.f0[k] = (int)((long long)F * acc / sum);
                acc += sh[k];
            }
            pl.f0[n] = F;
            // (no empty piece)
            for (int k = 1; k <= n; k++)
                if (pl.f0[k] <= pl.f0[k - 1]) pl.f0[k] = pl.f0[k - 1] + 1 < F ? pl.f0[k - 1] + 1 : F;
            while (pl.nsub > 1 && pl.f0[pl.nsub - 1] >= F) pl.nsub--;
            pl.f0[pl.nsub] = F;
        }
    }
    return pl;
}

// The whole detection pipeline for F frames whose gray images are resident at d_gray, in two halves: enqueue_detect puts every
// kernel and the result copies on the context's streams and returns (no host wait anywhere), finish_detect waits for them and
// hands the markers out.  fid_detect_device / fid_detect_batch call one after the other; fid_submit_device / fid_collect expose
// the halves, so that a caller with a stream of batches keeps two or three contexts in flight and the latency-bound tail of one
// batch runs under the front of the next.
// Everything a detect call can be refused for, checked BEFORE anything is put on a stream: feed_and_enqueue queues host -> device
// copies of the caller's buffer in front of the kernels, and a call that is then refused must not leave DMA from that buffer in
// flight (the caller may free it as soon as it sees the error).
// bytes one pixel occupies in a row; 0 = not an encoding this library takes.  FID_ENC_BIGENDIAN is only meaningful (and only
// accepted) on the 16-bit layouts.
static int enc_bytes_per_pixel(int enc)
{
    const int base = enc & 0xff;
    if (enc & ~(0xff | FID_ENC_BIGENDIAN)) return 0;
    if ((enc & FID_ENC_BIGENDIAN) && !(base >= FID_ENC_MONO16 && base <= FID_ENC_RGBA16)) return 0;
    switch (base) {
    case FID_ENC_MONO8: case FID_ENC_BAYER_RGGB8: case FID_ENC_BAYER_BGGR8: case FID_ENC_BAYER_GBRG8: case FID_ENC_BAYER_GRBG8: return 1;
    case FID_ENC_BGR8: case FID_ENC_RGB8: return 3;
    case FID_ENC_BGRA8: case FID_ENC_RGBA8: return 4;
    case FID_ENC_MONO16: case FID_ENC_YUV422: return 2;
    case FID_ENC_BGR16: case FID_ENC_RGB16: return 6;
    case FID_ENC_BGRA16: case FID_ENC_RGBA16: return 8;
    default: return 0;
    }
}

fid_status check_call(fid_ctx *c, int F, int W, int H, int stride, fid_encoding enc)
{
    if (F < 1 || F > c->lim.max_batch || W < 8 || H < 8 || W > c->lim.max_width || H > c->lim.max_height || W > 8191 || H > 8191)  // 13-bit checkpoint packing
        return FID_E_INVALID_ARG;
    const int bpp = enc_bytes_per_pixel(enc);
    if (bpp == 0 || stride < W * bpp) return FID_E_INVALID_ARG;
    if ((enc & 0xff) == FID_ENC_YUV422 && (W & 1)) return FID_E_INVALID_ARG;  // (a UYVY row is whole pixel pairs)
    const int maxdim = W > H ? W : H;
    if ((unsigned)(c->params.maxMarkerPerimeterRate * maxdim) > 36000u) return FID_E_UNSUPPORTED;  // contour points live in LDS
    const size_t pitch = (size_t)((int)(unsigned)(c->params.maxMarkerPerimeterRate * maxdim) / CK + 3);  // chunk_tab_pitch
    if ((size_t)F * 2 * c->lim.max_contours_per_frame * pitch > c->ckpts_elems) {
        c->last_error = "chunk table too small for this image size / maxMarkerPerimeterRate";
        return FID_E_UNSUPPORTED;
    }
    return FID_OK;
}

fid_status enqueue_detect(fid_ctx *c, const uint8_t *d_src, int F, int W, int H, int stride, long long fstride, fid_encoding enc)
{
    {
        const fid_status rcv = check_call(c, F, W, H, stride, enc);
        if (rcv != FID_OK) return rcv;
    }
    hipStream_t st0 = c->stream;
    c->pose_done = false;
    // ---- K0 geometry: mono8 device input is used in place (any stride); colour goes through k_to_gray
    const bool to_gray = enc != FID_ENC_MONO8;
    const int gstride = to_gray ? W : stride;
    const long long gfstride = to_gray ? (long long)W * H : fstride;
    const uint8_t *gray = to_gray ? c->d_gray : d_src;
    set_geometry(c, W, H, gstride, F);
    if (c->wait_ev) {
        HIPCHK(c, hipStreamWaitEvent(st0, c->wait_ev, 0));
        c->wait_ev = nullptr;
    }
    if (c->d_seedhash) {
        // a new generation makes every entry of the seed hash tables stale; clear them when the 10-bit number wraps
        if (++c->seed_gen > 1023) {
            HIPCHK(c, hipMemsetAsync(c->d_seedhash, 0, (size_t)c->lim.max_batch * c->seed_hash_cap * sizeof(unsigned long long), st0));
            c->seed_gen = 1;
        }
        c->P.seedGen = c->seed_gen;
    }
    // masks pad words must be zero; re-zero when the layout changes
    if (c->masks_W != W || c->masks_H != H || c->masks_S != c->P.nscales) {
        HIPCHK(c, hipMemsetAsync(c->d_masks, 0, c->masks_bytes, st0));
        c->masks_W = W;
        c->masks_H = H;
        c->masks_S = c->P.nscales;
    }
    layout_results(c, F);
    if (!c->res_precleared) HIPCHK(c, hipMemsetAsync(c->d_res, 0, c->res_clear_bytes, st0));  // global flags, work-list counters, per-frame counters
    c->res_precleared = false;
    // ---- the batch is cut into sub-batches that run the whole pipeline on their own streams: the latency-bound
    //      tail of one sub-batch's kernels (the longest border, the last candidates) overlaps the next one's bulk
    // (decided HERE and nowhere else: frames that come up from the host are cut by the copy's pieces whether or not the copy overlaps)
    c->piece_chain = c->pieces > 1 && !c->chained && !c->host_feed && !c->from_host_call && c->trace_mode == 2 && c->sub_frames <= 0 &&
                     F >= 32 * c->pieces && c->stagger == 0;
    const SubPlan plan = plan_sub_batches(c, F);
    const int nsub = plan.nsub;
    const bool chainp = c->piece_chain;
    c->last_nsub = nsub;
    {
        const fid_status rcs = ensure_streams(c, nsub, F);
        if (rcs != FID_OK) return rcs;
    }
    if (nsub > 1) HIPCHK(c, hipEventRecord(c->fork_ev, st0));
    // The host enqueues in two rounds: first every sub-batch's gray conversion + threshold, then every sub-batch's rest.  (One
    // round -- a whole sub-batch, some thirty launches, before the next one's first kernel -- left the second sub-batch's stream
    // empty for the first 0.35 - 0.6 ms of every step.)
    bool tail_recorded = false;
    auto sub_phase = [&](int sb, int phase) -> fid_status {
        const int f0 = plan.f0[sb], Fs = plan.f0[sb + 1] - f0;
        hipStream_t st = nsub > 1 ? c->sub_stream[chainp ? (sb & 1) : sb] : st0;
        if (nsub > 1 && phase == 0) HIPCHK(c, hipStreamWaitEvent(st, c->fork_ev, 0));
        // (a chain of pieces: this one's threshold starts when the piece before it is through its find_starts)
        if (chainp && phase == 0 && sb > 0) HIPCHK(c, hipStreamWaitEvent(st, c->fs_done[sb - 1], 0));
        if (c->host_feed && phase == 0) HIPCHK(c, hipStreamWaitEvent(st, c->in_ready[sb], 0));
        // staggered starts: the contour stage of a sub-batch (threshold ... approx) keeps the whole chip busy, what follows
        // (candidates, identification, corners) is a chain of short latency-bound kernels -- let the next sub-batch's contour
        // stage run under that tail instead of beside another contour stage
        if (nsub > 1 && phase == 0 && c->stagger > 0 && sb >= c->stagger) HIPCHK(c, hipStreamWaitEvent(st, c->walk_done[sb - c->stagger], 0));
        DevParams P = c->P;
        P.nframes = Fs;
        const size_t MC = (size_t)P.maxCands, MM = (size_t)P.maxMarkers;
        const uint8_t *g = gray + (long long)f0 * gfstride;
        uint32_t *masks = c->d_masks + (size_t)f0 * P.nscales * P.TR * P.TC * MT_ROWS;
        uint2 *starts = c->d_starts + (size_t)f0 * P.maxStarts, *surv1 = c->d_surv1 + (size_t)f0 * P.maxStarts,
              *surv = c->d_surv + (size_t)f0 * P.maxStarts;
        uint4 *contours = c->d_contours + (size_t)f0 * P.maxContours;
        uint32_t *tab = c->d_ckpts + (size_t)f0 * 2 * P.maxContours * chunk_tab_pitch(P);
        uint32_t *pool = c->d_pool + (size_t)f0 * P.maxChunks * CKW;  // (chain codes: CKW words per chunk of CK points)
        DevCounts *counts = c->d_counts + f0;
        DevCand *cands = c->d_cands + f0 * MC, *sorted = c->d_sorted + f0 * MC, *filtered = c->d_filtered + f0 * MC;
        uint32_t *nearb = c->d_near + f0 * MC * (MC / 32);
        DevIdent *ident = c->d_ident + f0 * MC;
        fid_marker *pre = c->d_pre + f0 * MM, *markers = c->d_markers + f0 * MM;
        unsigned *worklist = c->d_worklist + f0 * MC, *nwork = c->d_nwork + sb;
        hipEvent_t *ev = c->sub_ev[sb];
        auto mark = [&](int idx) {
            if (c->profile) (void)hipEventRecord(ev[idx], st);
        };
        auto chain_point = [&](int pt) {  // (a point the mode in use does not pass falls through to the next one that it does)
            if (sb == nsub - 1 && !tail_recorded && !c->blocking && c->chain_at <= pt) {  // (an event record costs the stream ~6 us)
                (void)hipEventRecord(c->tail_ev, st);
                tail_recorded = true;
            }
        };
      if (phase == 0) {
        mark(0);
        if (to_gray && (int)enc <= FID_ENC_RGBA8)
            hipLaunchKernelGGL(k_to_gray, dim3(2048), dim3(256), 0, st, d_src + (long long)f0 * fstride, stride, fstride, (int)enc,
                               c->d_gray + (size_t)f0 * W * H, W, H, Fs);
        else if (to_gray)  // (ABI 7: Bayer mosaics, 16-bit layouts, UYVY -- cv_bridge's conversion and BGR2GRAY in one pass)
            hipLaunchKernelGGL(k_raw_to_gray, dim3((W + 255) / 256, (H + 3) / 4, Fs), dim3(64, 4), 0, st, d_src + (long long)f0 * fstride, stride,
                               fstride, (int)enc, c->d_gray + (size_t)f0 * W * H, W, H);
        mark(ST_GRAY + 1);
        // ---- K1
        {
            bool node_table = P.nscales == 13;  // 3, 7, ..., 51: the node defaults (aruco_detect.cpp:690-693)
            for (int i = 0; i < P.nscales && node_table; i++) node_table = P.win[i] == 3 + 4 * i;
            if (node_table && c->thr_mode == 1) {
                // strips of 64 * NW columns, cut into row segments so that the grid fills the chip (a segment pays a warm-up
                // of about ten steps, so they are kept as tall as the batch allows)
                const int cols = 64 * c->thr_nw, strips = (W + cols - 1) / cols;
                int RS = c->thr_rows;
                if (RS <= 0) {
                    int segs = (1536 + strips * Fs - 1) / (strips * Fs);
                    const int maxsegs = (H + 63) / 64;
                    segs = segs < 1 ? 1 : (segs > maxsegs ? maxsegs : segs);
                    RS = (H + segs - 1) / segs;
                }
                RS = (RS + 3) & ~3;
                dim3 grid(strips, (H + RS - 1) / RS, Fs);
                if (c->thr_nw == 5 && c->thr_split) launch_thr_stream<5, true>(grid, st, g, gfstride, masks, P, RS, c->thr_xcd);
                else if (c->thr_nw == 5) launch_thr_stream<5, false>(grid, st, g, gfstride, masks, P, RS, c->thr_xcd);
                else if (c->thr_split) launch_thr_stream<3, true>(grid, st, g, gfstride, masks, P, RS, c->thr_xcd);
                else launch_thr_stream<3, false>(grid, st, g, gfstride, masks, P, RS, c->thr_xcd);
            } else if (node_table) {
                using C = ThrCfg<3, 4, 13>;
                dim3 grid((W + C::TX - 1) / C::TX, (H + C::TY - 1) / C::TY, Fs);
                hipLaunchKernelGGL((k_threshold_fixed<3, 4, 13>), grid, dim3(C::NT), C::LDS_BYTES, st, g, gfstride, masks, P);
            } else {
                int R = P.rmax, RW = TX + 2 * R, RH = TY + 2 * R, PT = (RW + 1) | 1;
                size_t lds = (size_t)(RH + 1) * PT * sizeof(uint32_t);
                dim3 grid((W + TX - 1) / TX, (H + TY - 1) / TY, Fs);
                hipLaunchKernelGGL((k_threshold<TX, TY, NT>), grid, dim3(NT), lds, st, g, gfstride, masks, P);
            }
        }
        mark(ST_THRESH + 1);
      }
        // ---- K2
        long long k2blocks;
        {
            long long groups = (long long)P.nscales * P.TR * ((P.WW + 15) / 16);  // two groups per wave and iteration
            k2blocks = (groups + 7) / 8;
            if (k2blocks > 256) k2blocks = 256;
        }
      if (phase == 0) {
        // the traced modes' start / seed enumeration belongs to the first round as well: every sub-batch's find_starts is then
        // queued before any walk, and (fs_barrier) the walks of all sub-batches wait for the last find_starts -- beside another
        // sub-batch's seed walk and probes a find_starts took 2.5 ms for 92 frames, beside another find_starts 1.1 ms for 164
        if (c->trace_mode >= 1) {
            hipLaunchKernelGGL(k_find_starts<true>, dim3((unsigned)k2blocks, Fs), dim3(256), 0, st, masks, starts, counts, c->d_global,
                               c->d_seedq + f0 * (size_t)P.maxContours, P);
            mark(ST_STARTS + 1);
            if (nsub > 1) HIPCHK(c, hipEventRecord(c->fs_done[sb], st));
            chain_point(0);
        }
        return FID_OK;
      }
        if (c->trace_mode >= 1 && nsub > 1 && c->fs_barrier && !chainp)
            for (int o = 0; o < nsub; o++)
                if (o != sb) HIPCHK(c, hipStreamWaitEvent(st, c->fs_done[o], 0));
        // persistent walker workgroups (WALK_WAVES waves each) per frame: about 16 waves per CU over the sub-batch
        int wb = c->walk_blocks > 0 ? c->walk_blocks : (3072 / WALK_WAVES + Fs - 1) / Fs;  // (measured: 6 per frame at 128 frames beats 8 and 12)
        // (the walks of one frame want every seed in flight at once: one frame, 32 px grid: seed walk 0.25 -> 0.115 ms)
        const int wcap = c->walk_blocks_cap > 0 ? c->walk_blocks_cap : (Fs <= 2 ? 256 : 64);
        wb = wb < 2 ? 2 : (wb > wcap ? wcap : wb);
        // kernels with a fixed number of workgroups per frame (sized for batches): a call of a few frames gets more of them
        const int gm = Fs >= 16 ? 1 : 16 / Fs;
        const int wb2 = wb > 1 ? wb / 2 : 1;  // the two walks share the CUs' LDS
        const int cap1 = pts_cap_first(P);
        const size_t lds1 = (size_t)cap1 * sizeof(uint32_t) + (size_t)K4_SHORT_STACK * sizeof(int2);
        const size_t lds2 = (size_t)(P.maxPerim + 1) * sizeof(uint32_t) + (size_t)K4_LONG_STACK * sizeof(int2);
        if (c->trace_mode == 0) {
            hipLaunchKernelGGL(k_find_starts<false>, dim3((unsigned)k2blocks, Fs), dim3(256), 0, st, masks, starts, counts, c->d_global,
                               (uint2 *)nullptr, P);
            mark(ST_STARTS + 1);
            // ---- K3: sieve the starts twice, then walk the survivors to the end
            hipLaunchKernelGGL((k_probe<PROBE0_STEPS, 0>), dim3(64 * gm, Fs), dim3(256), 0, st, masks, starts, surv1, counts, c->d_global, P);
            hipLaunchKernelGGL((k_probe<PROBE1_STEPS, 1>), dim3(16 * gm, Fs), dim3(256), 0, st, masks, surv1, surv, counts, c->d_global, P);
            mark(ST_PROBE + 1);
            hipLaunchKernelGGL(k_walk_full<0>, dim3(wb, Fs), dim3(64 * WALK_WAVES), 0, st, masks, surv, contours, tab, pool, (DevSeg *)nullptr,
                               (DevPend *)nullptr, counts, c->d_global, P);
            mark(ST_WALK + 1);
            // ---- K4: short contours with a small LDS footprint first, then the long / flagged ones
            fid_launch_log("k_approx", 64, (size_t)(lds1));
            hipLaunchKernelGGL(k_approx, dim3(128 * gm, Fs), dim3(64), lds1, st, contours, tab, pool, cands, counts, c->d_global, P, cap1,
                               K4_SHORT_STACK, 0, (const uint32_t *)nullptr, (const uint32_t *)nullptr);
            fid_launch_log("k_approx", 64, (size_t)(lds2));
            hipLaunchKernelGGL(k_approx, dim3(16 * gm, Fs), dim3(64), lds2, st, contours, tab, pool, cands, counts, c->d_global, P,
                               P.maxPerim + 1, K4_LONG_STACK, 1, (const uint32_t *)nullptr, (const uint32_t *)nullptr);
            mark(ST_APPROX + 1);
        } else {
            // ---- seed-accelerated tracing: seeds walk their segments while the starts are sieved; survivors walk to the
            //      first seed; link -> chain -> flatten
            const size_t MCn = (size_t)P.maxContours;
            DevSeg *segs = c->d_segs + f0 * MCn;
            DevPend *pend = c->d_pend + f0 * MCn;
            uint2 *seedq = c->d_seedq + f0 * MCn;
            unsigned long long *seedhash = c->d_seedhash + (size_t)f0 * P.seedHashCap;
            uint4 *wres = c->d_wres + f0 * MCn, *cinfo = c->d_cinfo + f0 * MCn;
            uint32_t *cbase = c->d_cbase + f0 * MCn;
            uint32_t *dense = c->d_dense + (size_t)f0 * P.maxChunks * (CK / 4);  // (one code byte per point)
            // (k_find_starts<true> was queued in the first round)
          if (c->trace_mode == 2) {
            // ---- cycle tracing.  Main stream: the seeds walk their segments, link, the cycles become contour list A, copy,
            //      approxPolyDP.  Auxiliary stream, beside it: seed index, then the chain that only exists for borders WITHOUT a
            //      seed state (probe passes that drop a start at the first seed state, survivor walk, contour list B, copy,
            //      approxPolyDP) -- off the critical path, it used to be 2.9 ms of a 9.6 ms sub-batch.  Both append candidates;
            //      the streams join in front of k_sort_cands.
            hipStream_t sa = c->aux_stream[chainp ? (sb & 1) : sb];
            HIPCHK(c, hipEventRecord(c->aux_fork[sb], st));
            HIPCHK(c, hipStreamWaitEvent(sa, c->aux_fork[sb], 0));
            // about 4 walker waves per CU over the sub-batch (34 KB LDS per 2-wave workgroup: two of them leave a CU room for
            // the other sub-batch's kernels; 8 waves per CU measured 7 % slower end to end)
            int swb = c->sw_blocks > 0 ? c->sw_blocks : (1024 / SW_WAVES + Fs - 1) / Fs;
            swb = swb < 2 ? 2 : (swb > 4 * wcap ? 4 * wcap : swb);
            uint4 *recs = c->d_recs + 2 * f0 * MCn;
            const int cpb = c->copy_blocks > 0 ? c->copy_blocks : 256;
            // -- the main stream's first kernel goes out BEFORE the auxiliary chain's eight launches: the host needs ~15 us to
            //    enqueue those, and for a single frame the seed walk sat waiting behind them (round 4, cfg 2 timeline)
            if (c->profile) (void)hipEventRecord(ev[14], st);
            hipLaunchKernelGGL(k_seed_walk<false>, dim3(swb, Fs), dim3(64 * SW_WAVES), 0, st, masks, seedq, tab, pool, (DevSegC *)segs, counts,
                               c->d_global, P, (uint4 *)nullptr, (DevPend *)nullptr);
            if (c->profile) (void)hipEventRecord(ev[15], st);
            mark(ST_PROBE + 1);
            chain_point(1);
            // -- auxiliary stream.  k_seed_index (the map seed state -> seed index, which only k_seg_link2 on the main stream and
            //    the auxiliary k_seg_cycles need) is off the auxiliary chain's critical path when the call is one piece: a stream of
            //    its own beside the probes (39 us of a single frame's 314 us contour stage)
            if (c->profile) (void)hipEventRecord(ev[16], sa);
            hipStream_t si = (nsub == 1 && Fs <= 16 && c->idx_stream) ? c->idx_stream : sa;
            if (si != sa) HIPCHK(c, hipStreamWaitEvent(si, c->aux_fork[sb], 0));
            hipLaunchKernelGGL(k_seed_index, dim3(8 * gm, Fs), dim3(256), 0, si, seedq, seedhash, counts, P);
            HIPCHK(c, hipEventRecord(c->aux_idx[sb], si));
            if (c->probe_lut) {
                if (c->probe_refill0 && gm == 1)
                    hipLaunchKernelGGL((k_probe_refill<PROBE0_STEPS, 0>), dim3(c->probe_refill0, Fs), dim3(256), 0, sa, masks, starts, surv1, counts, c->d_global, P,
                                       (const uint4 *)c->d_probe_tables);
                else
                hipLaunchKernelGGL((k_probe_lut<PROBE0_STEPS, 0>), dim3(64 * gm, Fs), dim3(256), 0, sa, masks, starts, surv1, counts, c->d_global, P,
                                   (const uint4 *)c->d_probe_tables);
                if (c->probe_refill && gm == 1)  // (batches: 16 waves a frame that keep their lanes filled; a call of a few frames wants every start in flight at once)
                    hipLaunchKernelGGL((k_probe_refill<PROBE1_STEPS, 1>), dim3(c->probe_refill, Fs), dim3(256), 0, sa, masks, surv1, surv, counts, c->d_global, P,
                                       (const uint4 *)c->d_probe_tables);
                else
                    hipLaunchKernelGGL((k_probe_lut<PROBE1_STEPS, 1>), dim3(16 * gm, Fs), dim3(256), 0, sa, masks, surv1, surv, counts, c->d_global, P,
                                       (const uint4 *)c->d_probe_tables);
            } else {
                hipLaunchKernelGGL((k_probe<PROBE0_STEPS, 0, true>), dim3(64 * gm, Fs), dim3(256), 0, sa, masks, starts, surv1, counts, c->d_global, P);
                hipLaunchKernelGGL((k_probe<PROBE1_STEPS, 1, true>), dim3(16 * gm, Fs), dim3(256), 0, sa, masks, surv1, surv, counts, c->d_global, P);
            }
            const int wb3 = c->walk2_div > 0 ? (wb / c->walk2_div > 0 ? wb / c->walk2_div : 1) : wb2;
            if (c->surv_walk_old)
                hipLaunchKernelGGL(k_walk_full<2>, dim3(wb3, Fs), dim3(64 * WALK_WAVES), 0, sa, masks, surv, wres, tab, pool, segs, pend, counts,
                                   c->d_global, P);
            else  // (FID_SURV_WALK=new: the survivors on the seed walker's prefetched windows)
                hipLaunchKernelGGL(k_seed_walk<true>, dim3(c->surv_blocks_x > 0 ? wb3 * (WALK_WAVES / SW_WAVES) * c->surv_blocks_x : wb3, Fs), dim3(64 * SW_WAVES), 0, sa, masks,
                                   (const uint2 *)surv, tab, pool, (DevSegC *)nullptr, counts, c->d_global, P, wres, pend);
            if (si != sa) HIPCHK(c, hipStreamWaitEvent(sa, c->aux_idx[sb], 0));  // (the survivors' seed look-ups need the map)
            hipLaunchKernelGGL(k_seg_cycles<0>, dim3(8 * gm, Fs), dim3(64), 0, sa, seedq, (const DevSegC *)segs, pend, wres, contours, cinfo, cbase,
                               recs, counts, c->d_global, P, 1, 0);
            hipLaunchKernelGGL(k_seg_copy, dim3(cpb / 4 > 0 ? cpb / 4 : 1, Fs), dim3(256), 0, sa, recs, tab, pool, dense, counts, P, 1);
            fid_launch_log("k_approx", 64, (size_t)(lds1));
            hipLaunchKernelGGL(k_approx, dim3(32 * gm, Fs), dim3(64), lds1, sa, contours, tab, pool, cands, counts, c->d_global, P, cap1,
                               K4_SHORT_STACK, 0, dense, cbase, 2);
            fid_launch_log("k_approx", 64, (size_t)(lds2));
            hipLaunchKernelGGL(k_approx, dim3(8 * gm, Fs), dim3(64), lds2, sa, contours, tab, pool, cands, counts, c->d_global, P,
                               P.maxPerim + 1, K4_LONG_STACK, 1, dense, cbase, 2);
            if (c->profile) (void)hipEventRecord(ev[17], sa);
            HIPCHK(c, hipEventRecord(c->aux_join[sb], sa));
            // -- main stream, continued
            HIPCHK(c, hipStreamWaitEvent(st, c->aux_idx[sb], 0));
            hipLaunchKernelGGL(k_seg_link2, dim3(16 * gm, Fs), dim3(256), 0, st, seedq, (DevSegC *)segs, seedhash, counts, P);
            // (a call of one or two frames: a lane for every seed at once -- a workgroup that came round to a second 64 seeds walked a
            //  second longest cycle behind the first, 2 x 26 us of the single frame's 55 us)
            const int scb = Fs <= 2 ? (int)((P.maxContours + 63) / 64 < 4096 ? (P.maxContours + 63) / 64 : 4096) : 32 * gm * c->light_x;
            if (Fs <= 2)
                hipLaunchKernelGGL(k_seg_cycles<48>, dim3(scb, Fs), dim3(64), 0, st, seedq, (const DevSegC *)segs, pend, wres, contours, cinfo, cbase,
                                   recs, counts, c->d_global, P, 0, getenv("FID_NO_SEGC_WARM") ? 0 : 1);
            else
                hipLaunchKernelGGL(k_seg_cycles<0>, dim3(scb, Fs), dim3(64), 0, st, seedq, (const DevSegC *)segs, pend, wres, contours, cinfo, cbase,
                                   recs, counts, c->d_global, P, 0, 0);
            hipLaunchKernelGGL(k_seg_copy, dim3(cpb, Fs), dim3(256), 0, st, recs, tab, pool, dense, counts, P, 0);
            mark(ST_WALK + 1);
            chain_point(2);
            fid_launch_log("k_approx", 64, (size_t)(lds1));
            hipLaunchKernelGGL(k_approx, dim3(128 * gm, Fs), dim3(64), lds1, st, contours, tab, pool, cands, counts, c->d_global, P, cap1,
                               K4_SHORT_STACK, 0, dense, cbase, 1);
            fid_launch_log("k_approx", 64, (size_t)(lds2));
            hipLaunchKernelGGL(k_approx, dim3(16 * gm, Fs), dim3(64), lds2, st, contours, tab, pool, cands, counts, c->d_global, P,
                               P.maxPerim + 1, K4_LONG_STACK, 1, dense, cbase, 1);
            mark(ST_APPROX + 1);
            chain_point(3);
            if (chainp) {
                // the candidate tail of this piece moves to its auxiliary stream (behind the seedless chain, which is there already):
                // the main stream is free for the front of the piece after next
                HIPCHK(c, hipEventRecord(c->front_done[sb], st));
                HIPCHK(c, hipStreamWaitEvent(sa, c->front_done[sb], 0));
                st = sa;
            } else {
                HIPCHK(c, hipStreamWaitEvent(st, c->aux_join[sb], 0));
            }
          } else {
            // the seed walk needs only the seeds: it runs on its own stream beside the probe passes and the survivor walk
            // (both walks are a throughput phase followed by a tail of a few long walkers; side by side the tails overlap)
            hipStream_t sa = c->aux_stream[sb];
            HIPCHK(c, hipEventRecord(c->aux_fork[sb], st));
            HIPCHK(c, hipStreamWaitEvent(sa, c->aux_fork[sb], 0));
            if (c->profile) (void)hipEventRecord(ev[14], sa);
            hipLaunchKernelGGL(k_walk_full<1>, dim3(wb2, Fs), dim3(64 * WALK_WAVES), 0, sa, masks, seedq, wres, tab, pool, segs, pend, counts,
                               c->d_global, P);
            if (c->profile) (void)hipEventRecord(ev[15], sa);
            HIPCHK(c, hipEventRecord(c->aux_join[sb], sa));
            hipLaunchKernelGGL((k_probe<PROBE0_STEPS, 0>), dim3(64 * gm, Fs), dim3(256), 0, st, masks, starts, surv1, counts, c->d_global, P);
            hipLaunchKernelGGL((k_probe<PROBE1_STEPS, 1>), dim3(16 * gm, Fs), dim3(256), 0, st, masks, surv1, surv, counts, c->d_global, P);
            mark(ST_PROBE + 1);
            hipLaunchKernelGGL(k_walk_full<2>, dim3(wb2, Fs), dim3(64 * WALK_WAVES), 0, st, masks, surv, wres, tab, pool, segs, pend, counts,
                               c->d_global, P);
            HIPCHK(c, hipStreamWaitEvent(st, c->aux_join[sb], 0));
            hipLaunchKernelGGL(k_seed_index, dim3(8 * gm, Fs), dim3(256), 0, st, seedq, seedhash, counts, P);
            hipLaunchKernelGGL(k_seg_link, dim3(16 * gm, Fs), dim3(256), 0, st, seedq, segs, surv, pend, seedhash, counts, c->d_global, P);
            uint4 *recs = c->d_recs + 2 * f0 * MCn;
            hipLaunchKernelGGL(k_seg_chain, dim3(16 * gm, Fs), dim3(64), 0, st, surv, pend, wres, segs, contours, cinfo, cbase, recs, counts,
                               c->d_global, P);
            hipLaunchKernelGGL(k_seg_copy, dim3(c->copy_blocks > 0 ? c->copy_blocks : 256, Fs), dim3(256), 0, st, recs, tab, pool, dense, counts, P, -1);
            mark(ST_WALK + 1);
            fid_launch_log("k_approx", 64, (size_t)(lds1));
            hipLaunchKernelGGL(k_approx, dim3(128 * gm, Fs), dim3(64), lds1, st, contours, tab, pool, cands, counts, c->d_global, P, cap1,
                               K4_SHORT_STACK, 0, dense, cbase);
            fid_launch_log("k_approx", 64, (size_t)(lds2));
            hipLaunchKernelGGL(k_approx, dim3(16 * gm, Fs), dim3(64), lds2, st, contours, tab, pool, cands, counts, c->d_global, P,
                               P.maxPerim + 1, K4_LONG_STACK, 1, dense, cbase);
            mark(ST_APPROX + 1);
          }
        }
        if (nsub > 1 && c->stagger > 0) HIPCHK(c, hipEventRecord(c->walk_done[sb], st));
        chain_point(4);
        // ---- K5
        float4 *cmeta = c->d_cmeta + f0 * MC;
        // (a rank sort: every candidate is compared with all of them -- a call of a few frames gives the frame sixteen workgroups)
        fid_launch_log("k_sort_cands", 256, (size_t)(MC * 8));
        hipLaunchKernelGGL(k_sort_cands, dim3(Fs, (c->light_x > 1 || Fs < 16) ? 16 : 1), dim3(256), MC * 8, st, cands, sorted, cmeta, counts, P);
        mark(ST_SORT + 1);
        hipLaunchKernelGGL(k_near, dim3(32 * gm * c->light_x, Fs), dim3(256), 0, st, sorted, cmeta, nearb, counts, P);
        mark(ST_NEAR + 1);
        {
            // sizes, labels, component sizes; the rest holds the near triangle (64 KB: fits beside two seed-walker workgroups of
            // the other sub-batch on a CU; with 96 KB the kernel waited for whole CUs to drain)
            const size_t lds = (size_t)c->resolve_lds_kb * 1024;
            const int near_words = c->resolve_serial ? 0 : (int)((lds - 3 * MC * 4) / 4);
            fid_launch_log("k_resolve", 1024, (size_t)(lds));
            hipLaunchKernelGGL(k_resolve, dim3(Fs), dim3(1024), lds, st, sorted, nearb, filtered, counts, worklist, nwork, P, near_words, c->d_global,
                               c->resolve_reg_max);
        }
        mark(ST_RESOLVE + 1);
        chain_point(5);
        // ---- K6
        {
            int SZ = (P.markerSize + 2 * P.borderBits) * P.cellSize;
            fid_launch_log("k_identify", 64, (size_t)((size_t)SZ * SZ));
            hipLaunchKernelGGL(k_identify, dim3(c->tail_grid > 0 ? c->tail_grid : 256 * 4), dim3(64), (size_t)SZ * SZ, st, g, gfstride, filtered, worklist, nwork,
                               c->d_dict, ident, P);
        }
        mark(ST_IDENT + 1);
        // ---- K7
        fid_launch_log("k_filter_markers", 64, (size_t)((size_t)c->filter_lds * sizeof(fid_marker)));
        hipLaunchKernelGGL(k_filter_markers, dim3(Fs), dim3(64), (size_t)c->filter_lds * sizeof(fid_marker), st, filtered, ident, pre, counts, P,
                           c->d_filter_scratch + f0 * MC, c->filter_lds, c->d_accsrc + f0 * MC, c->d_mksrc + f0 * MM);
        mark(ST_FILTER + 1);
        {
            long long items = (long long)Fs * P.maxMarkers * 4;
            const long long bcap = c->tail_grid > 0 ? 4LL * c->tail_grid : 256 * 16;
            int blocks = (int)(items < bcap ? items : bcap);
            if (P.refine == 2)  // CORNER_REFINE_CONTOUR: a wave per marker fits the four sides of its contour (writes ids and corners)
                hipLaunchKernelGGL(k_refine_contour, dim3(P.maxMarkers < 64 ? P.maxMarkers : 64, Fs), dim3(64), 0, st, (const fid_marker *)pre, markers,
                                   (const int *)(c->d_mksrc + f0 * MM), (const DevCand *)filtered,
                                   (const uint32_t *)(c->d_dense ? c->d_dense + (size_t)f0 * P.maxChunks * (CK / 4) : nullptr), (const uint32_t *)tab,
                                   (const uint32_t *)pool, counts, P);
            else
                hipLaunchKernelGGL(k_subpix, dim3(blocks), dim3(64), 0, st, g, gfstride, pre, markers, counts, c->d_subpix_mask, P);
        }
        mark(ST_SUBPIX + 1);
        if (c->pose_cam_valid) {
            PoseCam cam;
            for (int i = 0; i < 9; i++) cam.K[i] = c->pose_K[i];
            for (int i = 0; i < 5; i++) cam.D[i] = c->pose_D[i];
            cam.fiducial_len = c->pose_len;
            int blocks = (Fs * P.maxMarkers + 7) / 8;  // eight lanes per marker, eight markers per wave
            blocks = blocks < 1 ? 1 : (blocks > 256 * 16 ? 256 * 16 : blocks);
            if (c->profile) (void)hipEventRecord(ev[18], st);
            hipLaunchKernelGGL(k_pose, dim3(blocks), dim3(64), 0, st, (const fid_marker *)markers, (const int *)&counts[0].nmark,
                               (int)(sizeof(DevCounts) / sizeof(int)), (const double *)nullptr, Fs, P.maxMarkers, cam, c->d_poses + f0 * MM);
            if (c->profile) (void)hipEventRecord(ev[19], st);
        }
        chain_point(99);
        if (nsub > 1) {
            HIPCHK(c, hipEventRecord(c->sub_done[sb], st));
            // (a chain of pieces: the context's stream carries the fronts of the even pieces -- it waits for the tails when all
            //  pieces are enqueued, below)
            if (!chainp) HIPCHK(c, hipStreamWaitEvent(st0, c->sub_done[sb], 0));
        }
        return FID_OK;
    };
    if (chainp) {
        // piece by piece: piece k + 2 goes behind piece k on the same two streams
        for (int sb = 0; sb < nsub; sb++)
            for (int phase = 0; phase < 2; phase++) {
                const fid_status rcs = sub_phase(sb, phase);
                if (rcs != FID_OK) return rcs;
            }
        for (int sb = 0; sb < nsub; sb++) HIPCHK(c, hipStreamWaitEvent(st0, c->sub_done[sb], 0));
    } else {
        for (int phase = 0; phase < 2; phase++)
            for (int sb = 0; sb < nsub; sb++) {
                const fid_status rcs = sub_phase(sb, phase);
                if (rcs != FID_OK) return rcs;
            }
    }
    hipStream_t st = st0;
    const DevParams &P = c->P;
    HIPCHK(c, hipGetLastError());
    // ---- results
    c->pose_done = false;
    HIPCHK(c, hipMemcpyAsync(c->h_res, c->d_res, c->pose_cam_valid ? c->res_poses_end : c->res_markers_end, hipMemcpyDeviceToHost, st));  // one copy
    c->last_frames = F;
    c->last_W = W;
    c->last_H = H;
    c->last_gray = gray;
    c->last_gfstride = gfstride;
    c->pend = {d_src, F, W, H, stride, fstride, enc};
    c->in_flight = true;
    c->chained = false;
    c->wait_ev = nullptr;       // fid_order_after holds for ONE submit: the other context's events are not kept beyond it (it may
    c->wait_copy_ev = nullptr;  // be destroyed before this context's next call)
    c->fed_from_host = false;  // (feed_and_enqueue sets it after this call)
    return FID_OK;
}

fid_status run_detect(fid_ctx *c, const uint8_t *d_src, int F, int W, int H, int stride, long long fstride, fid_encoding enc,
                      fid_marker *out, int cap_per_frame, int *n_per_frame);

fid_status finish_detect(fid_ctx *c, fid_marker *out, int cap_per_frame, int *n_per_frame)
{
    const DevParams &P = c->P;
    const int F = c->pend.F;
    c->in_flight = false;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->pose_done = c->pose_cam_valid;
    if (c->profile) {
        // a stage's time = its event-bracketed time on its own stream, summed over the sub-batches (with more than
        // one sub-batch the brackets of different streams overlap in wall time)
        for (int i = 0; i < ST_COUNT; i++) c->stage_ms[i] = 0.f;
        for (int sb = 0; sb < c->last_nsub; sb++)
            for (int i = 0; i <= ST_SUBPIX; i++) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, c->sub_ev[sb][i], c->sub_ev[sb][i + 1]) == hipSuccess) c->stage_ms[i] += ms;
            }
        if (c->trace_mode >= 1)  // the seed walk runs on the auxiliary streams, beside walk_probe and part of walk_full
            for (int sb = 0; sb < c->last_nsub; sb++) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, c->sub_ev[sb][14], c->sub_ev[sb][15]) == hipSuccess) c->stage_ms[ST_SEEDWALK] += ms;
                if (c->trace_mode == 2 && hipEventElapsedTime(&ms, c->sub_ev[sb][16], c->sub_ev[sb][17]) == hipSuccess) c->stage_ms[ST_SEEDLESS] += ms;
                if (c->pose_cam_valid && hipEventElapsedTime(&ms, c->sub_ev[sb][18], c->sub_ev[sb][19]) == hipSuccess) c->stage_ms[ST_POSE] += ms;
            }
    }
#ifdef FID_DEBUG_STATS
    {
        const unsigned long long *d = c->h_global->dbg;
        fprintf(stderr, "walk stats: ranges %llu iters %llu ckpts %llu (forced %llu) active-lanes/iter %.1f cyc/range %.0f ckpt-cyc/range %.0f wait-cyc/range %.0f\n",
                d[7], d[0], d[1], d[5], d[0] ? (double)d[2] / d[0] : 0., d[7] ? (double)d[6] / d[7] : 0., d[7] ? (double)d[3] / d[7] : 0.,
                d[7] ? (double)d[4] / d[7] : 0.);
        fprintf(stderr, "walk waves %llu: total cyc avg %.0f max %llu; own queue dry at avg %.0f max %llu\n", d[12],
                d[12] ? (double)d[8] / d[12] : 0., d[9], d[12] ? (double)d[10] / d[12] : 0., d[11]);
        fprintf(stderr, "walk longest %llu steps, total steps %llu\n", d[13], d[14]);
        fprintf(stderr, "resolve (frame 0, n %llu, %llu label rounds) cycles: load %llu label %llu count %llu components %llu places %llu move %llu\n", d[31] >> 32,
                d[31] & 0xffffffffull, d[25], d[26], d[27], d[28], d[29], d[30]);
        if (d[1])
            fprintf(stderr, "checkpoint cycles: wait %.0f activate %.0f retire %.0f hand-out %.0f chunks %.0f refill %.0f; lanes at a checkpoint: need %.1f loading %.1f final %.1f idle %.1f\n",
                    (double)d[4] / d[1], (double)d[16] / d[1], (double)d[17] / d[1], (double)d[18] / d[1], (double)d[19] / d[1], (double)d[20] / d[1],
                    (double)d[21] / d[1], (double)d[22] / d[1], (double)d[23] / d[1], (double)d[24] / d[1]);
    }
#endif
    if (c->trace_mode >= 1 && (c->h_global->overflow & (2u | 8u))) {
        // the tracing seeds and their points scale with the total border length of the frame, texture included: when they
        // do not fit max_contours_per_frame / max_points_per_frame, trace this call with the whole-border walk, which only
        // needs room for the probe survivors
        if (getenv("FID_VERBOSE"))
            fprintf(stderr, "fid: seed tracing overflow flags 0x%x (frame 0: seeds %d starts %d surv %d chunks %d; maxContours %d maxChunks %d shift %d): whole-border walk\n",
                    c->h_global->overflow, c->h_counts[0].nseeds, c->h_counts[0].nstarts, c->h_counts[0].nsurv, c->h_counts[0].npool, c->P.maxContours,
                    c->P.maxChunks, c->P.seedShift);
        const int tm = c->trace_mode;
        c->trace_mode = 0;
        c->fallbacks++;
        const fid_ctx::Pending pd = c->pend;
        const fid_status rc2 = run_detect(c, pd.d_src, pd.F, pd.W, pd.H, pd.stride, pd.fstride, pd.enc, out, cap_per_frame, n_per_frame);
        c->trace_mode = tm;
        return rc2;
    }
    fid_status rc = FID_OK;
    if (c->h_global->overflow) {
        const unsigned ov = c->h_global->overflow;
        if (ov & 16u) {
            c->last_error = "internal error: broken segment chain";
        } else {
            // which table: so that the caller knows which limit to raise
            c->last_error = "internal capacity exceeded:";
            if (ov & 1u) c->last_error += " max_starts_per_frame";
            if (ov & 2u) c->last_error += " max_contours_per_frame";
            if (ov & 4u) c->last_error += " approximation stack";
            if (ov & 8u) c->last_error += " max_points_per_frame";
            c->last_error += " (raise fid_limits)";
        }
        rc = FID_E_CAPACITY;
    }
    for (int f = 0; f < F; f++) {
        int n = c->h_counts[f].nmark;
        if (c->h_counts[f].overflow & 3) {
            c->last_error = "per-frame candidate/marker capacity exceeded: raise fid_limits";
            rc = FID_E_CAPACITY;
        }
        if (c->h_counts[f].overflow & 4) {
            // CORNER_REFINE_CONTOUR on a marker with a side of fewer than two contour points: cv::solve throws inside
            // aruco::detectMarkers, imageCallback's catch(cv::Exception&) logs it and publishes nothing for the frame
            if (rc == FID_OK) {
                rc = FID_E_CV_EXCEPTION;
                c->last_error = "cv exception: the reference's detectMarkers throws on frame " + std::to_string(f) +
                                " (CORNER_REFINE_CONTOUR: a marker side of fewer than two contour points)";
            }
            n_per_frame[f] = -1;
            continue;
        }
        if (n > cap_per_frame) {
            n = cap_per_frame;
            c->last_error = "caller marker capacity too small";
            rc = FID_E_CAPACITY;
        }
        n_per_frame[f] = n;
        memcpy(out + (size_t)f * cap_per_frame, c->h_markers + (size_t)f * P.maxMarkers, sizeof(fid_marker) * n);
    }
    return rc;
}

fid_status run_detect(fid_ctx *c, const uint8_t *d_src, int F, int W, int H, int stride, long long fstride, fid_encoding enc,
                      fid_marker *out, int cap_per_frame, int *n_per_frame)
{
    if (!out || !n_per_frame || cap_per_frame < 0 || c->in_flight) {
        if (c->in_flight) c->last_error = "a submitted batch is in flight: fid_collect first";
        c->wait_ev = c->wait_copy_ev = nullptr;  // (fid_order_after holds for one call, refused or not)
        c->chained = false;
        return FID_E_INVALID_ARG;
    }
    c->blocking = true;
    const fid_status rc = enqueue_detect(c, d_src, F, W, H, stride, fstride, enc);
    c->blocking = false;
    if (rc != FID_OK) {
        c->wait_ev = c->wait_copy_ev = nullptr;
        c->chained = false;
        if (c->last_frames > 0 && !c->in_flight) layout_results(c, c->last_frames);  // (the layout of the last call that ran)
        return rc;
    }
    return finish_detect(c, out, cap_per_frame, n_per_frame);
}

}  // namespace

extern "C" {

void fid_default_params(fid_params *p)
{
    if (!p) return;
    memset(p, 0, sizeof(*p));
    // node defaults: aruco_detect.cpp:690-727
    p->adaptiveThreshConstant = 7;
    p->adaptiveThreshWinSizeMin = 3;
    p->adaptiveThreshWinSizeMax = 53;
    p->adaptiveThreshWinSizeStep = 4;
    p->cornerRefinementMethod = 1;
    p->cornerRefinementWinSize = 5;
    p->cornerRefinementMaxIterations = 30;
    p->cornerRefinementMinAccuracy = 0.01;
    p->errorCorrectionRate = 0.6;
    p->minCornerDistanceRate = 0.05;
    p->markerBorderBits = 1;
    p->minDistanceToBorder = 3;
    p->maxErroneousBitsInBorderRate = 0.04;
    p->minMarkerDistanceRate = 0.05;
    p->minMarkerPerimeterRate = 0.1;
    p->maxMarkerPerimeterRate = 4.0;
    p->minOtsuStdDev = 5.0;
    p->perspectiveRemoveIgnoredMarginPerCell = 0.13;
    p->perspectiveRemovePixelPerCell = 8;
    p->polygonalApproxAccuracyRate = 0.01;
}

void fid_default_limits(fid_limits *l)
{
    if (!l) return;
    memset(l, 0, sizeof(*l));
    l->max_width = 1920;
    l->max_height = 1080;
    l->max_batch = 1;
    l->max_starts_per_frame = 262144;
    l->max_contours_per_frame = 16384;
    l->max_candidates_per_frame = 2048;
    l->max_markers_per_frame = 256;
    l->max_points_per_frame = 4 * 1024 * 1024;
}

fid_status fid_create(const fid_params *params, const fid_dict *dict, const fid_limits *limits, int device, fid_ctx **out)
{
    if (!params || !dict || !out || !dict->bytes || dict->n_markers <= 0 || dict->marker_size < 3 || dict->marker_size > 7)
        return FID_E_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FID_E_NO_DEVICE;
    if (device < 0 || device >= ndev) return FID_E_INVALID_ARG;
    fid_ctx *c = new (std::nothrow) fid_ctx();
    if (!c) return FID_E_OUT_OF_MEMORY;
    c->device = device;
    fid_limits L;
    fid_default_limits(&L);
    if (limits) {
        if (limits->max_width > 0) L.max_width = limits->max_width;
        if (limits->max_height > 0) L.max_height = limits->max_height;
        if (limits->max_batch > 0) L.max_batch = limits->max_batch;
        if (limits->max_starts_per_frame > 0) L.max_starts_per_frame = limits->max_starts_per_frame;
        if (limits->max_contours_per_frame > 0) L.max_contours_per_frame = limits->max_contours_per_frame;
        if (limits->max_candidates_per_frame > 0) L.max_candidates_per_frame = limits->max_candidates_per_frame;
        if (limits->max_markers_per_frame > 0) L.max_markers_per_frame = limits->max_markers_per_frame;
        if (limits->max_points_per_frame > 0) L.max_points_per_frame = limits->max_points_per_frame;
    }
    // contexts for a few frames at a time (the node's shape, max_batch 1) get room for the denser seed grid by default
    if (L.max_batch <= 4) {
        if (!limits || limits->max_contours_per_frame <= 0) L.max_contours_per_frame = 65536;
        if (!limits || limits->max_points_per_frame <= 0) L.max_points_per_frame = 16 * 1024 * 1024;
    } else if (L.max_batch <= 16) {
        if (!limits || limits->max_contours_per_frame <= 0) L.max_contours_per_frame = 32768;
        if (!limits || limits->max_points_per_frame <= 0) L.max_points_per_frame = 8 * 1024 * 1024;
    }
    L.max_candidates_per_frame = roundup(L.max_candidates_per_frame, 32);
    if (L.max_candidates_per_frame > 4096 || L.max_batch > 65535 || L.max_markers_per_frame > L.max_candidates_per_frame ||
        L.max_contours_per_frame > (1 << 20)) {  // (seed indices are 20-bit fields of the SeedHash entries)
        delete c;
        return FID_E_INVALID_ARG;
    }
    c->lim = L;
    c->dict_ms = dict->marker_size;
    c->dict_maxc = dict->max_correction_bits;
    c->dict_n = dict->n_markers;
    size_t dbytes = (size_t)dict->n_markers * 4 * ((dict->marker_size * dict->marker_size + 7) / 8);
    c->dict_host.assign(dict->bytes, dict->bytes + dbytes);
    c->profile = getenv("FID_PROFILE") && atoi(getenv("FID_PROFILE")) != 0;
    if (getenv("FID_SUB_FRAMES")) c->sub_frames = atoi(getenv("FID_SUB_FRAMES"));
    if (getenv("FID_WALK_CAP")) c->walk_blocks_cap = atoi(getenv("FID_WALK_CAP"));
    if (getenv("FID_SEED_SHIFT")) c->seed_shift = atoi(getenv("FID_SEED_SHIFT"));
    if (getenv("FID_RESOLVE_SERIAL")) c->resolve_serial = atoi(getenv("FID_RESOLVE_SERIAL"));
    if (getenv("FID_RESOLVE_REG_MAX")) {
        const int v = atoi(getenv("FID_RESOLVE_REG_MAX"));
        c->resolve_reg_max = v < 0 ? 0 : (v > 64 ? 64 : v);
    }
    if (getenv("FID_COPY_BLOCKS")) c->copy_blocks = atoi(getenv("FID_COPY_BLOCKS"));
    if (getenv("FID_THR")) c->thr_mode = strcmp(getenv("FID_THR"), "tile") ? 1 : 0;
    if (getenv("FID_THR_NW")) c->thr_nw = atoi(getenv("FID_THR_NW")) == 3 ? 3 : 5;
    if (getenv("FID_THR_SPLIT")) c->thr_split = atoi(getenv("FID_THR_SPLIT")) != 0;
    if (getenv("FID_THR_ROWS")) c->thr_rows = atoi(getenv("FID_THR_ROWS"));
    if (getenv("FID_THR_XCD")) c->thr_xcd = atoi(getenv("FID_THR_XCD")) != 0;
    if (getenv("FID_SURV_WALK")) c->surv_walk_old = !strcmp(getenv("FID_SURV_WALK"), "old");
    if (getenv("FID_SURV_BLOCKS_X")) c->surv_blocks_x = atoi(getenv("FID_SURV_BLOCKS_X"));  // (0: half as many waves as k_walk_full<2> had)
    if (getenv("FID_WALK_BLOCKS")) c->walk_blocks = atoi(getenv("FID_WALK_BLOCKS")) > 0 ? atoi(getenv("FID_WALK_BLOCKS")) : c->walk_blocks;
    memset(&c->P, 0, sizeof(c->P));

    fid_status rc = FID_OK;
    auto fail = [&](fid_status s) {
        fid_destroy(c);
        return s;
    };
#define TRY(expr)                         \
    do {                                  \
        rc = (expr);                      \
        if (rc != FID_OK) return fail(rc); \
    } while (0)
#define TRYHIP(expr)                                   \
    do {                                               \
        hipError_t e_ = (expr);                        \
        if (e_ != hipSuccess) {                        \
            fprintf(stderr, "fid_create: %s: %s\n", #expr, hipGetErrorString(e_)); \
            return fail(e_ == hipErrorOutOfMemory ? FID_E_OUT_OF_MEMORY : FID_E_HIP); \
        }                                              \
    } while (0)
    TRYHIP(hipSetDevice(device));
    TRYHIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (int i = 0; i <= ST_COUNT; i++) TRYHIP(hipEventCreate(&c->ev[i]));
    TRYHIP(hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming));
    TRYHIP(hipEventCreateWithFlags(&c->tail_ev, hipEventDisableTiming));
    int prio_lo = 0, prio_hi = 0;  // (numerically lower = more urgent)
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    // FID_PRIO=1: earlier sub-batches more urgent.  Measured slower (18.1k vs 18.8k frames/s) than equal priorities.
    const bool use_prio = getenv("FID_PRIO") && atoi(getenv("FID_PRIO"));
    for (int sb = 0; sb < fid_ctx::MAX_SUB; sb++) {
        int prio = use_prio ? prio_hi + sb : prio_lo;
        if (prio > prio_lo) prio = prio_lo;
        // (the sub-batch and auxiliary streams themselves are made when a call first needs them -- ensure_streams(): every live
        //  stream is a claim on the hardware queues; eighteen idle ones per context cost the STag path its rate in round 2, and
        //  creating eight more for an experiment cost this path a quarter of its own)
        c->sub_prio[sb] = prio;
        TRYHIP(hipEventCreateWithFlags(&c->aux_fork[sb], hipEventDisableTiming));
        TRYHIP(hipEventCreateWithFlags(&c->aux_join[sb], hipEventDisableTiming));
        TRYHIP(hipEventCreateWithFlags(&c->sub_done[sb], hipEventDisableTiming));
        TRYHIP(hipEventCreateWithFlags(&c->walk_done[sb], hipEventDisableTiming));
        TRYHIP(hipEventCreateWithFlags(&c->in_ready[sb], hipEventDisableTiming));
        TRYHIP(hipEventCreateWithFlags(&c->fs_done[sb], hipEventDisableTiming));
        TRYHIP(hipEventCreateWithFlags(&c->front_done[sb], hipEventDisableTiming));
        TRYHIP(hipEventCreateWithFlags(&c->aux_idx[sb], hipEventDisableTiming));
        for (int i = 0; i < 20; i++) TRYHIP(hipEventCreate(&c->sub_ev[sb][i]));
    }
    const size_t F = L.max_batch, MC = L.max_candidates_per_frame, MM = L.max_markers_per_frame;
    TRY(dalloc(c, &c->d_subpix_mask, (size_t)(2 * SP_MAXWIN + 1) * (2 * SP_MAXWIN + 1)));
    TRY(dalloc(c, &c->d_probe_tables, (size_t)4096));
    hipLaunchKernelGGL(k_probe_tables, dim3(1), dim3(256), 0, nullptr, c->d_probe_tables);
    TRYHIP(hipGetLastError());
    TRYHIP(hipDeviceSynchronize());
    TRY(dalloc(c, &c->d_dict, dbytes));
    TRYHIP(hipMemcpy(c->d_dict, c->dict_host.data(), dbytes, hipMemcpyHostToDevice));
    rc = apply_params(c, params);
    if (rc != FID_OK) return fail(rc);
    TRY(dalloc(c, &c->d_gray, F * L.max_width * L.max_height));
    c->masks_bytes = masks_elems(c, L.max_width, L.max_height, (int)F) * sizeof(uint32_t);
    // a narrower image can need a larger padded pitch only through rounding; keep a margin
    c->masks_bytes += (size_t)F * c->P.nscales * (L.max_height + 2) * 16 * sizeof(uint32_t);
    TRYHIP(hipMalloc((void **)&c->d_masks, c->masks_bytes));
    TRY(dalloc(c, &c->d_starts, F * L.max_starts_per_frame));
    TRY(dalloc(c, &c->d_surv1, F * L.max_starts_per_frame));
    TRY(dalloc(c, &c->d_surv, F * L.max_starts_per_frame));
    c->max_chunks = (L.max_points_per_frame + CK - 1) / CK;
    TRY(dalloc(c, &c->d_pool, F * (size_t)c->max_chunks * CKW));
    c->trace_mode = 2;
    if (const char *tm = getenv("FID_TRACE")) c->trace_mode = !strcmp(tm, "legacy") ? 0 : !strcmp(tm, "chain") ? 1 : 2;
    if (getenv("FID_SW_BLOCKS")) c->sw_blocks = atoi(getenv("FID_SW_BLOCKS"));
    if (getenv("FID_STAGGER")) c->stagger = atoi(getenv("FID_STAGGER"));
    if (getenv("FID_PIECES")) c->pieces = atoi(getenv("FID_PIECES"));
    if (getenv("FID_FS_BARRIER")) c->fs_barrier = atoi(getenv("FID_FS_BARRIER"));
    if (getenv("FID_PROBE_LUT")) c->probe_lut = atoi(getenv("FID_PROBE_LUT"));
    if (getenv("FID_TAIL_GRID")) c->tail_grid = atoi(getenv("FID_TAIL_GRID"));
    if (getenv("FID_FILTER_LDS")) c->filter_lds = atoi(getenv("FID_FILTER_LDS")) > 0 ? atoi(getenv("FID_FILTER_LDS")) : 1;
    if (getenv("FID_LIGHT_X")) c->light_x = atoi(getenv("FID_LIGHT_X")) > 0 ? atoi(getenv("FID_LIGHT_X")) : 1;
    if (getenv("FID_RESOLVE_LDS")) c->resolve_lds_kb = atoi(getenv("FID_RESOLVE_LDS"));
    if (getenv("FID_WALK2_DIV")) c->walk2_div = atoi(getenv("FID_WALK2_DIV"));
    if (getenv("FID_PROBE_REFILL")) c->probe_refill = atoi(getenv("FID_PROBE_REFILL"));
    if (getenv("FID_PROBE_REFILL0")) c->probe_refill0 = atoi(getenv("FID_PROBE_REFILL0"));
    if (getenv("FID_CHAIN_AT")) c->chain_at = atoi(getenv("FID_CHAIN_AT"));
    static_assert(sizeof(DevSegC) == sizeof(DevSeg), "the two segment records share one buffer");
    if (c->trace_mode >= 1) {
        TRY(dalloc(c, &c->d_segs, F * L.max_contours_per_frame));
        TRY(dalloc(c, &c->d_pend, F * L.max_contours_per_frame));
        TRY(dalloc(c, &c->d_seedq, F * L.max_contours_per_frame));
        c->seed_hash_cap = 1;
        while (c->seed_hash_cap < 2 * L.max_contours_per_frame) c->seed_hash_cap *= 2;
        TRY(dalloc(c, &c->d_seedhash, F * (size_t)c->seed_hash_cap));
        TRYHIP(hipMemset(c->d_seedhash, 0, F * (size_t)c->seed_hash_cap * sizeof(unsigned long long)));
        TRY(dalloc(c, &c->d_wres, F * L.max_contours_per_frame));
        TRY(dalloc(c, &c->d_cinfo, F * L.max_contours_per_frame));
        TRY(dalloc(c, &c->d_cbase, F * L.max_contours_per_frame));
        TRY(dalloc(c, &c->d_recs, 2 * F * L.max_contours_per_frame));
        TRY(dalloc(c, &c->d_dense, F * (size_t)c->max_chunks * (CK / 4) + 2));  // (+ 8 bytes: k_approx reads a contour's last codes eight at a time)
    }
    TRY(dalloc(c, &c->d_contours, F * L.max_contours_per_frame));
    {
        int maxdim = L.max_width > L.max_height ? L.max_width : L.max_height;
        size_t nck = (size_t)(params->maxMarkerPerimeterRate * maxdim) / 64 + 4;
        c->ckpts_elems = F * 2 * L.max_contours_per_frame * nck;  // rows: seeds, then survivors
        TRY(dalloc(c, &c->d_ckpts, c->ckpts_elems));
    }
    TRY(dalloc(c, &c->d_cands, F * MC));
    TRY(dalloc(c, &c->d_sorted, F * MC));
    TRY(dalloc(c, &c->d_cmeta, F * MC));
    TRY(dalloc(c, &c->d_filtered, F * MC));
    TRY(dalloc(c, &c->d_near, F * MC * (MC / 32)));
    TRY(dalloc(c, &c->d_ident, F * MC));
    TRY(dalloc(c, &c->d_pre, F * MM));
    TRY(dalloc(c, &c->d_filter_scratch, F * MC));
    TRY(dalloc(c, &c->d_accsrc, F * MC));
    TRY(dalloc(c, &c->d_mksrc, F * MM));
    TRY(dalloc(c, &c->d_res, results_bytes((int)F, (int)MM)));
    TRY(dalloc(c, &c->d_worklist, F * MC));
    TRY(dalloc(c, &c->d_pose_n, 1));
    TRYHIP(hipHostMalloc((void **)&c->h_res, results_bytes((int)F, (int)MM), hipHostMallocDefault));
    memset(c->h_res, 0, results_bytes((int)F, (int)MM));
    layout_results(c, (int)F);
    TRYHIP(hipMemset(c->d_masks, 0, c->masks_bytes));
    // opt in to large dynamic LDS where needed
    TRYHIP(hipFuncSetAttribute((const void *)k_threshold<TX, TY, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    TRYHIP(hipFuncSetAttribute((const void *)k_threshold_fixed<3, 4, 13>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)ThrCfg<3, 4, 13>::LDS_BYTES));
    TRYHIP(hipFuncSetAttribute((const void *)k_approx, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
    TRYHIP(hipFuncSetAttribute((const void *)k_resolve, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    TRYHIP(hipFuncSetAttribute((const void *)k_filter_markers, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
#undef TRY
#undef TRYHIP
    *out = c;
    return FID_OK;
}

void fid_destroy(fid_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    void *dev[] = {c->d_in, c->d_gray, c->d_masks, c->d_starts, c->d_surv1, c->d_surv, c->d_pool, c->d_segs, c->d_pend, c->d_seedq, c->d_seedhash, c->d_wres, c->d_cinfo, c->d_cbase, c->d_filter_scratch, c->d_accsrc, c->d_mksrc, c->d_dense, c->d_recs, c->d_contours, c->d_ckpts, c->d_cands, c->d_sorted, c->d_cmeta, c->d_filtered, c->d_near,
                   c->d_ident, c->d_pre, c->d_res, c->d_worklist, c->d_dict,
                   c->d_subpix_mask, c->d_probe_tables, c->d_lens, c->d_pose_in, c->d_pose_n};
    for (void *p : dev)
        if (p) (void)hipFree(p);
    void *host[] = {c->h_res};
    for (void *p : host)
        if (p) (void)hipHostFree(p);
    for (int i = 0; i <= ST_COUNT; i++)
        if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    if (c->fork_ev) (void)hipEventDestroy(c->fork_ev);
    if (c->tail_ev) (void)hipEventDestroy(c->tail_ev);
    for (int sb = 0; sb < fid_ctx::MAX_SUB; sb++) {
        if (c->sub_stream[sb]) (void)hipStreamSynchronize(c->sub_stream[sb]);
        if (sb == 0 && c->idx_stream) {
            (void)hipStreamSynchronize(c->idx_stream);
            (void)hipStreamDestroy(c->idx_stream);
            c->idx_stream = nullptr;
        }
        if (c->aux_stream[sb]) {
            (void)hipStreamSynchronize(c->aux_stream[sb]);
            (void)hipStreamDestroy(c->aux_stream[sb]);
        }
        if (c->aux_fork[sb]) (void)hipEventDestroy(c->aux_fork[sb]);
        if (c->aux_join[sb]) (void)hipEventDestroy(c->aux_join[sb]);
        if (c->aux_idx[sb]) (void)hipEventDestroy(c->aux_idx[sb]);
        if (c->sub_done[sb]) (void)hipEventDestroy(c->sub_done[sb]);
        if (c->walk_done[sb]) (void)hipEventDestroy(c->walk_done[sb]);
        if (c->in_ready[sb]) (void)hipEventDestroy(c->in_ready[sb]);
        if (c->fs_done[sb]) (void)hipEventDestroy(c->fs_done[sb]);
        if (c->front_done[sb]) (void)hipEventDestroy(c->front_done[sb]);
        for (int i = 0; i < 20; i++)
            if (c->sub_ev[sb][i]) (void)hipEventDestroy(c->sub_ev[sb][i]);
        if (c->sub_stream[sb] && c->sub_stream[sb] != c->stream) (void)hipStreamDestroy(c->sub_stream[sb]);
    }
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

fid_status fid_set_params(fid_ctx *c, const fid_params *p)
{
    if (!c || !p || c->in_flight) return FID_E_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int old_scales = c->P.nscales;
    fid_params keep = c->params;
    fid_status rc = apply_params(c, p);
    if (rc == FID_OK && c->P.nscales > old_scales) {
        // the masks buffer was sized for the scale count at creation (plus a margin): a larger scale count must fit
        size_t need = masks_elems(c, c->lim.max_width, c->lim.max_height, c->lim.max_batch) * sizeof(uint32_t);
        if (need > c->masks_bytes) {
            (void)apply_params(c, &keep);
            c->last_error = "more threshold scales than the context was created for";
            return FID_E_UNSUPPORTED;
        }
    }
    if (rc == FID_OK) {
        // what run_detect would refuse on every later frame is refused here, at reconfigure time: the chunk table was sized
        // for the creation-time maxMarkerPerimeterRate, and contour points live in LDS
        const int maxdim = c->lim.max_width > c->lim.max_height ? c->lim.max_width : c->lim.max_height;
        const unsigned maxPerim = (unsigned)(p->maxMarkerPerimeterRate * maxdim);
        DevParams Q = c->P;
        Q.maxPerim = (int)maxPerim;
        if (maxPerim > 36000u ||
            (size_t)c->lim.max_batch * 2 * c->lim.max_contours_per_frame * chunk_tab_pitch(Q) > c->ckpts_elems) {
            (void)apply_params(c, &keep);
            c->last_error = "maxMarkerPerimeterRate larger than the context was created for";
            return FID_E_UNSUPPORTED;
        }
    }
    if (rc != FID_OK) (void)apply_params(c, &keep);
    c->masks_W = 0;  // force re-zero
    return rc;
}

fid_status fid_detect_device(fid_ctx *c, const void *d_imgs, int32_t nframes, int32_t width, int32_t height, int32_t stride,
                             int64_t frame_stride, fid_encoding enc, fid_marker *out, int32_t cap_per_frame, int32_t *n_per_frame)
{
    if (!c) return FID_E_INVALID_ARG;
    const hipError_t dev_rc = d_imgs ? hipSetDevice(c->device) : hipSuccess;
    if (!d_imgs || dev_rc != hipSuccess) {
        c->wait_ev = c->wait_copy_ev = nullptr;  // (fid_order_after holds for one call, refused or not)
        c->chained = false;
        if (!d_imgs) return FID_E_INVALID_ARG;
        c->last_error = std::string("hipSetDevice: ") + hipGetErrorString(dev_rc);  // a runtime failure is not an argument error
        return FID_E_HIP;
    }
    return run_detect(c, (const uint8_t *)d_imgs, nframes, width, height, stride, frame_stride, enc, out, cap_per_frame, n_per_frame);
}

fid_status fid_submit_device(fid_ctx *c, const void *d_imgs, int32_t nframes, int32_t width, int32_t height, int32_t stride,
                             int64_t frame_stride, fid_encoding enc)
{
    if (!c) return FID_E_INVALID_ARG;
    fid_status rc = FID_E_INVALID_ARG;
    const bool was_in_flight = c->in_flight;
    if (was_in_flight) c->last_error = "a submitted batch is in flight: fid_collect first";
    else if (d_imgs) {
        const hipError_t dev_rc = hipSetDevice(c->device);
        if (dev_rc == hipSuccess) rc = enqueue_detect(c, (const uint8_t *)d_imgs, nframes, width, height, stride, frame_stride, enc);
        else {
            c->last_error = std::string("hipSetDevice: ") + hipGetErrorString(dev_rc);
            rc = FID_E_HIP;
        }
    }
    c->wait_ev = nullptr;  // (fid_order_after holds for one submit, refused or not)
    c->wait_copy_ev = nullptr;
    c->chained = false;
    if (rc != FID_OK && !was_in_flight && c->last_frames > 0) layout_results(c, c->last_frames);
    return rc;
}

fid_status fid_order_after(fid_ctx *c, fid_ctx *prev)
{
    if (!c || c == prev || c->in_flight || (prev && prev->device != c->device)) return FID_E_INVALID_ARG;
    c->wait_ev = prev && prev->in_flight ? prev->tail_ev : nullptr;
    c->wait_copy_ev = prev && prev->in_flight && prev->fed_from_host && prev->last_nsub >= 1 ? prev->in_ready[prev->last_nsub - 1] : nullptr;
    c->chained = prev != nullptr;
    return FID_OK;
}

fid_status fid_collect(fid_ctx *c, fid_marker *out, int32_t cap_per_frame, int32_t *n_per_frame)
{
    if (!c || !out || !n_per_frame || cap_per_frame < 0) return FID_E_INVALID_ARG;
    if (!c->in_flight) {
        c->last_error = "nothing was submitted";
        return FID_E_INVALID_ARG;
    }
    HIPCHK(c, hipSetDevice(c->device));
    return finish_detect(c, out, cap_per_frame, n_per_frame);
}

// frames in host memory: the copies go on the copy stream, the pipeline is enqueued behind them (fid_detect_batch, fid_submit_batch)
// fid_order_after holds for ONE submit, refused or not: the other context's events are not kept beyond it (that context may be
// destroyed before this one's next call), and a refused call leaves no result layout of its own behind (fid_pose_last /
// fid_tap_read address the result block through last_frames).
static void drop_order_and_layout(fid_ctx *c, bool layout_touched)
{
    c->wait_ev = c->wait_copy_ev = nullptr;
    c->chained = false;
    if (layout_touched) {
        if (c->last_frames > 0) layout_results(c, c->last_frames);
        c->res_precleared = false;
    }
}

static fid_status feed_and_enqueue_impl(fid_ctx *c, const uint8_t *imgs, int32_t nframes, int32_t width, int32_t height, int32_t stride,
                                        int64_t frame_stride, fid_encoding enc);

static fid_status feed_and_enqueue(fid_ctx *c, const uint8_t *imgs, int32_t nframes, int32_t width, int32_t height, int32_t stride,
                                   int64_t frame_stride, fid_encoding enc)
{
    if (!c) return FID_E_INVALID_ARG;
    const fid_status rc = feed_and_enqueue_impl(c, imgs, nframes, width, height, stride, frame_stride, enc);
    if (rc != FID_OK) drop_order_and_layout(c, true);
    return rc;
}

static fid_status feed_and_enqueue_impl(fid_ctx *c, const uint8_t *imgs, int32_t nframes, int32_t width, int32_t height, int32_t stride,
                                        int64_t frame_stride, fid_encoding enc)
{
    if (!c || !imgs || nframes < 1 || height < 1 || stride < 1) return FID_E_INVALID_ARG;
    if (nframes > c->lim.max_batch) return FID_E_INVALID_ARG;
    if (c->in_flight) {
        c->last_error = "a submitted batch is in flight: fid_collect first";
        return FID_E_INVALID_ARG;
    }
    HIPCHK(c, hipSetDevice(c->device));
    if (frame_stride < (int64_t)stride * height) return FID_E_INVALID_ARG;
    {
        // geometry, encoding and limits BEFORE any copy is queued: a refused call leaves no DMA from the caller's buffer behind
        const fid_status rcv = check_call(c, nframes, width, height, stride, enc);
        if (rcv != FID_OK) {
            c->wait_ev = c->wait_copy_ev = nullptr;
            c->chained = false;
            return rcv;
        }
    }
    size_t need = (size_t)frame_stride * (nframes - 1) + (size_t)stride * height;
    if (need > c->d_in_bytes) {
        if (c->d_in) (void)hipFree(c->d_in);
        c->d_in = nullptr;
        c->d_in_bytes = 0;
        HIPCHK(c, hipMalloc((void **)&c->d_in, need));
        c->d_in_bytes = need;
    }
    // The frames go up in the pieces run_detect works in, one asynchronous copy per sub-batch on the copy stream with an event
    // behind it; a sub-batch's stream waits for its own event only, so the copy of sub-batch k + 1 runs under the kernels of
    // sub-batch k (pinned host memory -- a capture ring buffer -- copies at link speed; pageable memory is staged by the runtime
    // and copies more slowly, the overlap is the same).  A batch of a chain (fid_order_after) is one piece: its copy runs
    // under the kernels of the batch before it, on the other context.
    c->host_feed = getenv("FID_NO_FEED_OVERLAP") == nullptr;
    c->piece_chain = false;  // (frames that come up from the host are cut by the copy's pieces)
    const SubPlan plan = plan_sub_batches(c, nframes);
    const int nsub = plan.nsub;
    if (c->blocking && nsub == 1 && !c->wait_ev && !c->wait_copy_ev) {
        // one piece, nothing to overlap with: the copy goes on the main stream itself (no event between it and the first kernel)
        // BEHIND the clear of the result block, which so runs under the copy instead of after it (the single-frame call: 34 us
        // between the end of the copy and the start of the threshold kernel, round 4 timeline)
        c->host_feed = false;
        layout_results(c, nframes);
        HIPCHK(c, hipMemsetAsync(c->d_res, 0, c->res_clear_bytes, c->stream));
        c->res_precleared = true;
    }
    if (c->host_feed && !c->copy_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    if (c->host_feed && c->wait_copy_ev) HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->wait_copy_ev, 0));
    c->wait_copy_ev = nullptr;
    if (!c->host_feed) {
        HIPCHK(c, hipMemcpyAsync(c->d_in, imgs, need, hipMemcpyHostToDevice, c->stream));
    } else {
        // (the copy stream must not overtake the previous call's kernels that still read d_in: they were synchronised at its end)
        for (int sb = 0; sb < nsub; sb++) {
            const int f0 = plan.f0[sb], Fs = plan.f0[sb + 1] - f0;
            const size_t off = (size_t)frame_stride * f0;
            const size_t bytes = (size_t)frame_stride * (Fs - 1) + (size_t)stride * height;
            HIPCHK(c, hipMemcpyAsync(c->d_in + off, imgs + off, bytes, hipMemcpyHostToDevice, c->copy_stream));
            HIPCHK(c, hipEventRecord(c->in_ready[sb], c->copy_stream));
        }
    }
    const bool fed = c->host_feed;
    c->from_host_call = true;  // (enqueue_detect plans the same sub-batches: no piece chain, FID_NO_FEED_OVERLAP or not)
    const fid_status rc = enqueue_detect(c, c->d_in, nframes, width, height, stride, frame_stride, enc);
    c->host_feed = c->from_host_call = false;
    if (rc == FID_OK) {
        c->fed_from_host = fed;
    } else {
        // (a HIP error half way through the enqueue: nothing will ever wait for the copies, so wait for them here -- the caller's
        //  buffer is not read after this function has returned an error)
        if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
        (void)hipStreamSynchronize(c->stream);
        c->wait_ev = c->wait_copy_ev = nullptr;
        c->chained = false;
    }
    return rc;
}

fid_status fid_detect_batch(fid_ctx *c, const uint8_t *imgs, int32_t nframes, int32_t width, int32_t height, int32_t stride,
                            int64_t frame_stride, fid_encoding enc, fid_marker *out, int32_t cap_per_frame, int32_t *n_per_frame)
{
    if (!out || !n_per_frame || cap_per_frame < 0) return FID_E_INVALID_ARG;
    if (c) c->blocking = true;
    const fid_status rc = feed_and_enqueue(c, imgs, nframes, width, height, stride, frame_stride, enc);
    if (c) c->blocking = c->res_precleared = false;
    if (rc != FID_OK) return rc;
    return finish_detect(c, out, cap_per_frame, n_per_frame);
}

fid_status fid_submit_batch(fid_ctx *c, const uint8_t *imgs, int32_t nframes, int32_t width, int32_t height, int32_t stride,
                            int64_t frame_stride, fid_encoding enc)
{
    const fid_status rc = feed_and_enqueue(c, imgs, nframes, width, height, stride, frame_stride, enc);
    if (c) {
        c->wait_ev = nullptr;
        c->wait_copy_ev = nullptr;
        c->chained = false;
    }
    return rc;
}

fid_status fid_detect(fid_ctx *c, const uint8_t *img, int32_t width, int32_t height, int32_t stride, fid_encoding enc,
                      fid_marker *out, int32_t cap, int32_t *n)
{
    return fid_detect_batch(c, img, 1, width, height, stride, (int64_t)stride * height, enc, out, cap, n);
}

static fid_status run_pose(fid_ctx *c, const fid_marker *d_markers, const int *d_n, int n_stride_ints, const double *d_lens,
                           int F, int per_frame, const double K[9], const double D[5], double fiducial_len, fid_pose_out *d_out)
{
    PoseCam cam;
    for (int i = 0; i < 9; i++) cam.K[i] = K[i];
    for (int i = 0; i < 5; i++) cam.D[i] = D ? D[i] : 0.;
    cam.fiducial_len = fiducial_len;
    int total = F * per_frame;
    int blocks = (total + 7) / 8;  // eight lanes per marker, eight markers per wave
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (c->profile) (void)hipEventRecord(c->ev[ST_POSE], c->stream);
    hipLaunchKernelGGL(k_pose, dim3(blocks), dim3(64), 0, c->stream, d_markers, d_n, n_stride_ints, d_lens, F, per_frame, cam, d_out);
    if (c->profile) (void)hipEventRecord(c->ev[ST_POSE + 1], c->stream);
    HIPCHK(c, hipGetLastError());
    return FID_OK;
}

fid_status fid_pose_last(fid_ctx *c, const double K[9], const double D[5], double fiducial_len, fid_pose_out *out,
                         int32_t cap_per_frame)
{
    if (!c || !K || !out || c->last_frames <= 0 || !(fiducial_len > 0)) return FID_E_INVALID_ARG;
    if (c->in_flight) {
        c->last_error = "a submitted batch is in flight: fid_collect first";
        return FID_E_INVALID_ARG;
    }
    HIPCHK(c, hipSetDevice(c->device));
    const int F = c->last_frames, MM = c->P.maxMarkers;
    double Dz[5] = {0., 0., 0., 0., 0.};
    if (D) memcpy(Dz, D, sizeof Dz);
    const bool same_cam = c->pose_cam_valid && !memcmp(c->pose_K, K, sizeof c->pose_K) && !memcmp(c->pose_D, Dz, sizeof Dz) && c->pose_len == fiducial_len;
    fid_status rc = FID_OK;
    if (!(same_cam && c->pose_done)) {  // (else: the detect call already ran k_pose for this camera on these markers)
        rc = run_pose(c, c->d_markers, &c->d_counts[0].nmark, (int)(sizeof(DevCounts) / sizeof(int)), nullptr, F, MM, K, D, fiducial_len, c->d_poses);
        if (rc != FID_OK) return rc;
        HIPCHK(c, hipMemcpyAsync(c->h_poses, c->d_poses, sizeof(fid_pose_out) * (size_t)F * MM, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->profile) (void)hipEventElapsedTime(&c->stage_ms[ST_POSE], c->ev[ST_POSE], c->ev[ST_POSE + 1]);
        memcpy(c->pose_K, K, sizeof c->pose_K);
        memcpy(c->pose_D, Dz, sizeof Dz);
        c->pose_len = fiducial_len;
        c->pose_cam_valid = getenv("FID_NO_POSE_AHEAD") == nullptr;
        c->pose_done = c->pose_cam_valid;
    }
    for (int f = 0; f < F; f++) {
        int n = c->h_counts[f].nmark;
        if (n > cap_per_frame) {
            n = cap_per_frame;
            rc = FID_E_CAPACITY;
        }
        memcpy(out + (size_t)f * cap_per_frame, c->h_poses + (size_t)f * MM, sizeof(fid_pose_out) * n);
    }
    return rc;
}

fid_status fid_pose(fid_ctx *c, const double K[9], const double D[5], const fid_marker *markers, const double *len_per_marker,
                    int32_t n, double fiducial_len, fid_pose_out *out)
{
    if (!c || !K || (n > 0 && (!markers || !out)) || n < 0 || !(fiducial_len > 0)) return FID_E_INVALID_ARG;
    if (n == 0) return FID_OK;
    if (len_per_marker)
        for (int i = 0; i < n; i++)
            if (!(len_per_marker[i] > 0)) return FID_E_INVALID_ARG;  // CV_Assert(markerLength > 0), aruco_detect.cpp:229
    if (c->in_flight) return FID_E_INVALID_ARG;  // (shares the context's stream and buffers with the batch in flight)
    HIPCHK(c, hipSetDevice(c->device));
    if (n > c->pose_cap) {
        if (c->d_pose_in) (void)hipFree(c->d_pose_in);
        if (c->d_lens) (void)hipFree(c->d_lens);
        c->d_pose_in = nullptr;
        c->d_lens = nullptr;
        c->pose_cap = 0;
        int cap = roundup(n, 256);
        HIPCHK(c, hipMalloc((void **)&c->d_pose_in, sizeof(fid_marker) * cap + sizeof(fid_pose_out) * cap));
        HIPCHK(c, hipMalloc((void **)&c->d_lens, sizeof(double) * cap));
        c->pose_cap = cap;
    }
    fid_pose_out *d_out = (fid_pose_out *)((char *)c->d_pose_in + sizeof(fid_marker) * c->pose_cap);
    HIPCHK(c, hipMemcpyAsync(c->d_pose_in, markers, sizeof(fid_marker) * n, hipMemcpyHostToDevice, c->stream));
    std::vector<double> lens(n);
    for (int i = 0; i < n; i++) lens[i] = len_per_marker ? len_per_marker[i] : fiducial_len;
    HIPCHK(c, hipMemcpyAsync(c->d_lens, lens.data(), sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    int nn = n;
    HIPCHK(c, hipMemcpyAsync(c->d_pose_n, &nn, sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // lens/nn are stack/heap temporaries
    fid_status rc = run_pose(c, c->d_pose_in, c->d_pose_n, 0, c->d_lens, 1, n, K, D, fiducial_len, d_out);
    if (rc != FID_OK) return rc;
    HIPCHK(c, hipMemcpyAsync(out, d_out, sizeof(fid_pose_out) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FID_OK;
}

fid_status fid_refine_contour_corners(fid_ctx *c, const int32_t *pts_xy, const int32_t *offsets, int32_t n, float *corners,
                                      int32_t *status_per_marker)
{
    if (!c || !pts_xy || !offsets || !corners || !status_per_marker || n < 0) return FID_E_INVALID_ARG;
    if (c->in_flight) {
        c->last_error = "a submitted batch is in flight: fid_collect first";
        return FID_E_INVALID_ARG;
    }
    if (n == 0) return FID_OK;
    HIPCHK(c, hipSetDevice(c->device));
    if (offsets[0] != 0) return FID_E_INVALID_ARG;
    for (int i = 0; i < n; i++)
        if (offsets[i + 1] < offsets[i]) return FID_E_INVALID_ARG;
    const int total = offsets[n];
    std::vector<uint32_t> packed((size_t)(total > 0 ? total : 1));
    for (int k = 0; k < total; k++) {
        const int x = pts_xy[2 * k], y = pts_xy[2 * k + 1];
        if (x < 0 || y < 0 || x > 0xffff || y > 0xffff) return FID_E_INVALID_ARG;
        packed[k] = (uint32_t)x | ((uint32_t)y << 16);
    }
    uint32_t *d_pts = nullptr;
    int *d_off = nullptr, *d_status = nullptr;
    float *d_corners = nullptr;
    fid_status rc = FID_OK;
    auto chk = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && rc == FID_OK) {
            c->last_error = std::string(what) + ": " + hipGetErrorString(e);
            rc = e == hipErrorOutOfMemory ? FID_E_OUT_OF_MEMORY : FID_E_HIP;
        }
    };
    chk(hipMalloc((void **)&d_pts, packed.size() * sizeof(uint32_t)), "hipMalloc");
    chk(hipMalloc((void **)&d_off, (size_t)(n + 1) * sizeof(int)), "hipMalloc");
    chk(hipMalloc((void **)&d_status, (size_t)n * sizeof(int)), "hipMalloc");
    chk(hipMalloc((void **)&d_corners, (size_t)n * 8 * sizeof(float)), "hipMalloc");
    if (rc == FID_OK) {
        chk(hipMemcpyAsync(d_pts, packed.data(), packed.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream), "hipMemcpyAsync");
        chk(hipMemcpyAsync(d_off, offsets, (size_t)(n + 1) * sizeof(int), hipMemcpyHostToDevice, c->stream), "hipMemcpyAsync");
        chk(hipMemcpyAsync(d_corners, corners, (size_t)n * 8 * sizeof(float), hipMemcpyHostToDevice, c->stream), "hipMemcpyAsync");
    }
    if (rc == FID_OK) {
        hipLaunchKernelGGL(k_refine_contour_pts, dim3(n), dim3(64), 0, c->stream, (const uint32_t *)d_pts, (const int *)d_off, d_corners, d_status);
        chk(hipGetLastError(), "k_refine_contour_pts");
        chk(hipMemcpyAsync(corners, d_corners, (size_t)n * 8 * sizeof(float), hipMemcpyDeviceToHost, c->stream), "hipMemcpyAsync");
        chk(hipMemcpyAsync(status_per_marker, d_status, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, c->stream), "hipMemcpyAsync");
        chk(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    }
    (void)hipFree(d_pts);
    (void)hipFree(d_off);
    (void)hipFree(d_status);
    (void)hipFree(d_corners);
    if (rc == FID_OK)
        for (int i = 0; i < n; i++)
            if (status_per_marker[i]) rc = FID_E_CV_EXCEPTION;
    return rc;
}

int64_t fid_tap_bytes(fid_ctx *c, fid_tap which)
{
    if (!c || c->last_frames <= 0) return 0;
    const DevParams &P = c->P;
    const int64_t F = c->last_frames;
    const int msb = P.markerSize + 2 * P.borderBits;
    switch (which) {
    case FID_TAP_MASKS: return F * P.nscales * P.H * P.WW * 4;
    case FID_TAP_CANDIDATES:
    case FID_TAP_FILTERED: return F * P.maxCands * (int64_t)sizeof(fid_candidate);
    case FID_TAP_BITS: return F * P.maxCands * msb * msb;
    case FID_TAP_IDENT: return F * P.maxCands * 8;
    case FID_TAP_PRESUBPIX: return F * P.maxMarkers * (int64_t)sizeof(fid_marker);
    case FID_TAP_COUNTS: return F * 12 * 4;
    case FID_TAP_GRAY: return F * (int64_t)P.W * P.H;
    }
    return 0;
}

fid_status fid_tap_read(fid_ctx *c, fid_tap which, void *dst, int64_t dst_bytes)
{
    if (!c || !dst || c->last_frames <= 0 || c->in_flight) return FID_E_INVALID_ARG;
    int64_t need = fid_tap_bytes(c, which);
    if (need <= 0 || dst_bytes < need) return FID_E_CAPACITY;
    HIPCHK(c, hipSetDevice(c->device));
    const DevParams &P = c->P;
    const int F = c->last_frames;
    const int msb = P.markerSize + 2 * P.borderBits;
    switch (which) {
    case FID_TAP_MASKS: {
        // un-tile on the host
        const size_t plane = (size_t)P.TR * P.TC * MT_ROWS;
        std::vector<uint32_t> tmp(plane);
        for (int fs = 0; fs < F * P.nscales; fs++) {
            HIPCHK(c, hipMemcpy(tmp.data(), c->d_masks + (size_t)fs * plane, plane * 4, hipMemcpyDeviceToHost));
            uint32_t *o = (uint32_t *)dst + (size_t)fs * P.H * P.WW;
            for (int y = 0; y < P.H; y++)
                for (int w = 0; w < P.WW; w++) o[(size_t)y * P.WW + w] = tmp[(size_t)mask_word(P.TC, y + 1, MASK_PADW + w)];
        }
        return FID_OK;
    }
    case FID_TAP_CANDIDATES:
    case FID_TAP_FILTERED: {
        std::vector<DevCand> tmp((size_t)F * P.maxCands);
        HIPCHK(c, hipMemcpy(tmp.data(), which == FID_TAP_CANDIDATES ? c->d_sorted : c->d_filtered, tmp.size() * sizeof(DevCand),
                            hipMemcpyDeviceToHost));
        fid_candidate *o = (fid_candidate *)dst;
        for (size_t i = 0; i < tmp.size(); i++) {
            o[i].scale = tmp[i].scale;
            o[i].contour_size = tmp[i].size;
            o[i].start_x = tmp[i].sx;
            o[i].start_y = tmp[i].sy;
            o[i].is_hole = tmp[i].hole;
            memcpy(o[i].corners, tmp[i].c, sizeof(float) * 8);
        }
        return FID_OK;
    }
    case FID_TAP_BITS:
    case FID_TAP_IDENT: {
        std::vector<DevIdent> tmp((size_t)F * P.maxCands);
        HIPCHK(c, hipMemcpy(tmp.data(), c->d_ident, tmp.size() * sizeof(DevIdent), hipMemcpyDeviceToHost));
        if (which == FID_TAP_BITS) {
            uint8_t *o = (uint8_t *)dst;
            for (size_t i = 0; i < tmp.size(); i++) memcpy(o + i * msb * msb, tmp[i].bits, (size_t)msb * msb);
        } else {
            int32_t *o = (int32_t *)dst;
            for (size_t i = 0; i < tmp.size(); i++) {
                o[2 * i] = tmp[i].id;
                o[2 * i + 1] = tmp[i].rot;
            }
        }
        return FID_OK;
    }
    case FID_TAP_PRESUBPIX:
        HIPCHK(c, hipMemcpy(dst, c->d_pre, (size_t)need, hipMemcpyDeviceToHost));
        return FID_OK;
    case FID_TAP_COUNTS: {
        int32_t *o = (int32_t *)dst;
        for (int f = 0; f < F; f++) {
            o[12 * f + 0] = c->h_counts[f].nstarts;
            o[12 * f + 1] = c->trace_mode == 0 ? (c->h_counts[f].nsurv < c->P.maxContours ? c->h_counts[f].nsurv : c->P.maxContours) : c->h_counts[f].ncontours + c->h_counts[f].ncontours2;
            o[12 * f + 10] = c->h_counts[f].nseeds;
            o[12 * f + 2] = c->h_counts[f].ncand;
            o[12 * f + 3] = c->h_counts[f].nfilt;
            o[12 * f + 4] = c->h_counts[f].nacc;
            o[12 * f + 5] = c->h_counts[f].nmark;
            o[12 * f + 6] = c->h_counts[f].overflow | ((int32_t)c->h_global->overflow << 8);
            o[12 * f + 7] = c->h_counts[f].nsurv;
            o[12 * f + 8] = c->h_counts[f].npool;
            o[12 * f + 9] = c->h_counts[f].nsurv1;
            o[12 * f + 11] = c->h_counts[f].ndense;  // contour points handed to approxPolyDP (every contour's rounded up to 8)
        }
        return FID_OK;
    }
    case FID_TAP_GRAY:
        for (int f = 0; f < F; f++)
            HIPCHK(c, hipMemcpy2D((char *)dst + (size_t)f * P.W * P.H, (size_t)P.W, c->last_gray + (size_t)f * c->last_gfstride,
                                  (size_t)P.gstride, (size_t)P.W, P.H, hipMemcpyDeviceToHost));
        return FID_OK;
    }
    return FID_E_INVALID_ARG;
}

int32_t fid_last_stage_ms(fid_ctx *c, float *ms, int32_t cap, const char *const **names)
{
    if (names) *names = kStageNames;
    if (!c || !ms) return ST_COUNT;
    for (int i = 0; i < ST_COUNT && i < cap; i++) ms[i] = c->profile ? c->stage_ms[i] : -1.f;
    return ST_COUNT;
}

int32_t fid_last_launches(fid_ctx *c) { return c ? c->last_nsub : 0; }

void *fid_stream(fid_ctx *c) { return c ? (void *)c->stream : nullptr; }

const char *fid_strerror(fid_status s)
{
    switch (s) {
    case FID_OK: return "ok";
    case FID_E_INVALID_ARG: return "invalid argument";
    case FID_E_NO_DEVICE: return "no HIP device (this library has no CPU fallback)";
    case FID_E_HIP: return "HIP runtime error";
    case FID_E_CAPACITY: return "capacity exceeded";
    case FID_E_OUT_OF_MEMORY: return "out of memory";
    case FID_E_CV_EXCEPTION:
        return "the reference's OpenCV call throws cv::Exception on this input (n = -1 for the frame)";
    case FID_E_UNSUPPORTED: return "unsupported parameter combination";
    }
    return "unknown status";
}

const char *fid_last_error(fid_ctx *c) { return c ? c->last_error.c_str() : ""; }

int32_t fid_abi_version(void) { return FID_ABI_VERSION; }

}  // extern "C"

#include "fid_stag.hip"
#include "fid_jpeg.hip"
#include "fid_png.hip"
#include "fid_draw.hip"
#include "fid_dict.hip"
