// stag_test.cpp -- the C++ host side of the stag_detect drop-in (host/include/stag_host.hpp) on a rendered frame:
//   usage: stag_test <frame.pgm> <expected.txt: "n  then n lines: id x0 y0 x1 y1 x2 y2 x3 y3 tz"> <data_dir> <hd> <errorCorrection>
// checks Stag::detectMarkers / getMarkerList (ids that were drawn, corners at the drawn places), the 5-point pose (distance
// as rendered) and the fiducial_msgs output of stagImageCallback.
#include <cstdio>
#include <fstream>
#include <iostream>

#include "stag_host.hpp"

using namespace fiducials_amd;

static int g_fail = 0;
#define CHECK(cond)                                                       \
    do {                                                                  \
        if (!(cond)) {                                                    \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            g_fail++;                                                     \
        }                                                                 \
    } while (0)

int main(int argc, char **argv)
{
    if (argc < 6) {
        std::printf("usage: %s <frame.pgm> <expected.txt> <data dir> <hd> <errorCorrection>\n", argv[0]);
        return 2;
    }
    try {
        std::ifstream f(argv[1], std::ios::binary);
        std::string magic;
        int w, h, maxv;
        f >> magic >> w >> h >> maxv;
        f.get();
        Image img;
        img.width = w; img.height = h; img.step = w; img.encoding = "mono8";
        img.header.seq = 42; img.header.sec = 100; img.header.nsec = 7;
        img.data.resize((size_t)w * h);
        f.read((char *)img.data.data(), (std::streamsize)img.data.size());
        struct Exp { int id; double c[8], tz; };
        std::vector<Exp> exp;
        {
            std::ifstream e(argv[2]);
            int n;
            e >> n;
            exp.resize(n);
            for (auto &x : exp) {
                e >> x.id;
                for (double &v : x.c) e >> v;
                e >> x.tz;
            }
        }
        bool threw = false;
        try {
            Stag bad(12, 2, false, argv[3]);
        } catch (const std::invalid_argument &) {
            threw = true;
        }
        CHECK(threw);  // Decoder::Decoder throws std::invalid_argument for an unknown library
        Stag stag(std::atoi(argv[4]), std::atoi(argv[5]), false, argv[3], w, h);
        CameraInfo cam;
        cam.header.frame_id = "camera";
        cam.K = {1400.0 * w / 1920.0, 0, w / 2.0, 0, 1400.0 * w / 1920.0, h / 2.0, 0, 0, 1};
        cam.D = {0, 0, 0, 0, 0};
        FiducialArray fva;
        FiducialTransformArray fta;
        stagImageCallback(stag, img, cam, 0.14, &fva, &fta);
        const std::vector<Marker> markers = stag.getMarkerList();
        CHECK(markers.size() >= 3 && fva.fiducials.size() == markers.size() && fta.transforms.size() == markers.size());
        CHECK(fva.image_seq == 42 && fta.header.frame_id == "camera" && fva.header.sec == 100);
        for (size_t i = 0; i < markers.size(); i++) {
            double best = 1e9, best_tz = 0;
            for (auto &x : exp) {
                if (x.id != markers[i].id) continue;
                double d = 0;
                for (int c = 0; c < 4; c++) {
                    d = std::fmax(d, std::fabs(markers[i].corners[c].x - x.c[2 * c]));
                    d = std::fmax(d, std::fabs(markers[i].corners[c].y - x.c[2 * c + 1]));
                }
                if (d < best) {
                    best = d;
                    best_tz = x.tz;
                }
            }
            CHECK(best < 2.0);  // an id that was drawn, at the drawn place
            const FiducialTransform &t = fta.transforms[i];
            CHECK(t.fiducial_id == markers[i].id && std::fabs(t.tz / best_tz - 1.0) < 0.03);
            CHECK(std::fabs(t.qx * t.qx + t.qy * t.qy + t.qz * t.qz + t.qw * t.qw - 1.0) < 1e-9 && t.fiducial_area > 1000);
            CHECK(fva.fiducials[i].x0 == markers[i].corners[0].x && fva.fiducials[i].y3 == markers[i].corners[3].y);
        }
        CHECK(serialize(fva).size() == 16 + 6 + 8 + 72 * fva.fiducials.size());
    } catch (const std::exception &e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 3;
    }
    std::printf(g_fail ? "%d check(s) failed\n" : "all checks passed%.0d\n", g_fail);
    return g_fail ? 1 : 0;
}
