// batches_in_turn_test.cpp -- the C-ABI's split call from the reference's language: fid_submit_batch / fid_collect /
// fid_order_after on two contexts against fid_detect_batch, on batches made of the reference's own test image (the node's
// imageCallback is called for as long as the camera runs, aruco_detect.cpp:332-350: a stream of frames).  Plain C++ on
// include/fid_abi.h, no HIP, no ROS; the image comes as the binary PGM the pytest wrapper writes from tests/golden/tag_01.npz.
//   usage: batches_in_turn_test <tag_01.pgm> <data_dir>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "fiducials_host.hpp"

using namespace fiducials_amd;

static int g_fail = 0;
#define CHECK(cond)                                                       \
    do {                                                                  \
        if (!(cond)) {                                                    \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            g_fail++;                                                     \
            return;                                                       \
        }                                                                 \
    } while (0)

static void run(const std::string &pgm, const std::string &data_dir)
{
    std::ifstream f(pgm, std::ios::binary);
    CHECK((bool)f);
    std::string magic;
    int w = 0, h = 0, maxv = 0;
    f >> magic >> w >> h >> maxv;
    f.get();
    CHECK(magic == "P5" && w > 0 && h > 0);
    std::vector<uint8_t> img((size_t)w * h);
    f.read((char *)img.data(), (std::streamsize)img.size());

    const int F = 4, NB = 5, CAP = 16;
    // batch b: frame k is the image shifted down by (b + k) rows (so that every batch and frame differs)
    std::vector<std::vector<uint8_t>> batches(NB, std::vector<uint8_t>((size_t)F * w * h));
    for (int b = 0; b < NB; b++)
        for (int k = 0; k < F; k++) {
            const int s = b + k;
            uint8_t *dst = batches[b].data() + (size_t)k * w * h;
            std::memcpy(dst + (size_t)s * w, img.data(), (size_t)(h - s) * w);
            for (int y = 0; y < s; y++) std::memcpy(dst + (size_t)y * w, img.data(), (size_t)w);
        }

    const Dictionary dict = getPredefinedDictionary(7, data_dir);
    const fid_dict dv = dict.view();
    fid_params prm;
    fid_default_params(&prm);
    fid_limits lim;
    fid_default_limits(&lim);
    lim.max_width = w;
    lim.max_height = h;
    lim.max_batch = F;
    fid_ctx *a = nullptr, *b = nullptr;
    CHECK(fid_create(&prm, &dv, &lim, 0, &a) == FID_OK);
    CHECK(fid_create(&prm, &dv, &lim, 0, &b) == FID_OK);

    // one call after the other: what the batches must give
    std::vector<std::vector<fid_marker>> want(NB, std::vector<fid_marker>((size_t)F * CAP));
    std::vector<std::vector<int32_t>> want_n(NB, std::vector<int32_t>(F));
    for (int k = 0; k < NB; k++) {
        CHECK(fid_detect_batch(a, batches[k].data(), F, w, h, w, (int64_t)w * h, FID_ENC_MONO8, want[k].data(), CAP, want_n[k].data()) == FID_OK);
        for (int fr = 0; fr < F; fr++) CHECK(want_n[k][fr] == 1 && want[k][(size_t)fr * CAP].id == 1);
    }

    // the rules of the split call
    std::vector<fid_marker> got((size_t)F * CAP);
    std::vector<int32_t> got_n(F);
    CHECK(fid_collect(a, got.data(), CAP, got_n.data()) == FID_E_INVALID_ARG);  // nothing submitted
    CHECK(fid_order_after(a, a) == FID_E_INVALID_ARG);
    CHECK(fid_submit_batch(a, batches[0].data(), F, w, h, w, (int64_t)w * h, FID_ENC_MONO8) == FID_OK);
    CHECK(fid_submit_batch(a, batches[1].data(), F, w, h, w, (int64_t)w * h, FID_ENC_MONO8) == FID_E_INVALID_ARG);  // one batch per context
    CHECK(fid_detect(a, img.data(), w, h, w, FID_ENC_MONO8, got.data(), CAP, got_n.data()) == FID_E_INVALID_ARG);
    CHECK(fid_order_after(a, b) == FID_E_INVALID_ARG);  // (a has a batch in flight)
    CHECK(fid_collect(a, got.data(), CAP, got_n.data()) == FID_OK);
    CHECK(!std::memcmp(got_n.data(), want_n[0].data(), sizeof(int32_t) * F) && !std::memcmp(got.data(), want[0].data(), sizeof(fid_marker) * got.size()));

    // two contexts in turn: submit k, collect k - 1
    fid_ctx *ring[2] = {a, b};
    for (int k = 0; k <= NB; k++) {
        if (k < NB) {
            fid_ctx *cur = ring[k & 1], *prev = ring[(k + 1) & 1];
            CHECK(fid_order_after(cur, prev) == FID_OK);
            CHECK(fid_submit_batch(cur, batches[k].data(), F, w, h, w, (int64_t)w * h, FID_ENC_MONO8) == FID_OK);
        }
        if (k >= 1) {
            std::fill(got.begin(), got.end(), fid_marker{});
            CHECK(fid_collect(ring[(k - 1) & 1], got.data(), CAP, got_n.data()) == FID_OK);
            CHECK(!std::memcmp(got_n.data(), want_n[k - 1].data(), sizeof(int32_t) * F));
            for (int fr = 0; fr < F; fr++)
                CHECK(!std::memcmp(&got[(size_t)fr * CAP], &want[k - 1][(size_t)fr * CAP], sizeof(fid_marker) * got_n[fr]));
        }
    }
    fid_destroy(a);
    fid_destroy(b);
}

int main(int argc, char **argv)
{
    if (argc < 3) {
        std::printf("usage: %s <tag_01.pgm> <data_dir>\n", argv[0]);
        return 2;
    }
    run(argv[1], argv[2]);
    if (g_fail) return 1;
    std::printf("all checks passed\n");
    return 0;
}
