// cv_adapter_demo.cpp — the reference's main() hot-path lines (W:217-233, W:271-313) with the two lines a maintainer changes
// (INTEGRATION.md §2): the warper and the blender are include/imagestitch_cv.hpp's subclasses of cv::detail::RotationWarper /
// cv::detail::Blender, used through base-class pointers as the demos' Ptr<RotationWarper> / Ptr<Blender> are.  cv::Mat in, cv::Mat out.
// Built by tests/test_gpu_cpp_mirror.py against tests/cpp/opencv_stub (this image has no OpenCV; with the real OpenCV 3.4.2 the same
// file compiles unchanged), which compares the files it writes with the CPU oracle.
//   usage: cv_adapter_demo <w> <h> <focal> <in0.raw> <in1.raw> <out_prefix>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "imagestitch_cv.hpp"

using namespace cv;
using namespace cv::detail;

static Mat rot(double yaw, double pitch, double roll) {
    double cy = cos(yaw), sy = sin(yaw), cp = cos(pitch), sp = sin(pitch), cr = cos(roll), sr = sin(roll);
    double Ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy}, Rx[9] = {1, 0, 0, 0, cp, -sp, 0, sp, cp}, Rz[9] = {cr, -sr, 0, sr, cr, 0, 0, 0, 1};
    double T[9], O[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { T[i * 3 + j] = 0; for (int k = 0; k < 3; ++k) T[i * 3 + j] += Ry[i * 3 + k] * Rx[k * 3 + j]; }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { O[i * 3 + j] = 0; for (int k = 0; k < 3; ++k) O[i * 3 + j] += T[i * 3 + k] * Rz[k * 3 + j]; }
    Mat R(3, 3, CV_32F);
    for (int i = 0; i < 9; ++i) R.at<float>(i / 3, i % 3) = (float)O[i];
    return R;
}

static void dump(const char* prefix, const char* name, const Mat& m) {
    char path[512];
    snprintf(path, sizeof(path), "%s_%s.raw", prefix, name);
    FILE* f = fopen(path, "wb");
    for (int y = 0; y < m.rows; ++y) fwrite(m.ptr<unsigned char>(y), 1, (size_t)m.cols * Mat::elemSize(m.type()), f);
    fclose(f);
    printf("%s %d %d %d\n", name, m.rows, m.cols, m.type());
}

int main(int argc, char** argv) {
    if (argc < 7) return 2;
    int w = atoi(argv[1]), h = atoi(argv[2]);
    float focal = (float)atof(argv[3]);
    try {
        const int num_images = 2;
        std::vector<Mat> imgs(num_images);
        for (int i = 0; i < num_images; ++i) {
            imgs[i].create(h, w, CV_8UC3);
            FILE* f = fopen(argv[4 + i], "rb");
            if (!f || fread(imgs[i].data, 1, (size_t)w * h * 3, f) != (size_t)w * h * 3) return 3;
            fclose(f);
        }
        Mat K(3, 3, CV_32F);
        const float kv[9] = {focal, 0, w / 2.0f, 0, focal, h / 2.0f, 0, 0, 1};
        for (int i = 0; i < 9; ++i) K.at<float>(i / 3, i % 3) = kv[i];
        Mat R[2] = {rot(-0.36, 0.010, 0.005), rot(0.36, 0.010, 0.005)};
        std::vector<Point> corners(num_images);                                        // W:206-210
        std::vector<Mat> masks_warped(num_images), images_warped(num_images), masks(num_images);
        std::vector<Size> sizes(num_images);
        for (int i = 0; i < num_images; ++i) { masks[i].create(h, w, CV_8U); memset(masks[i].data, 255, (size_t)w * h); }   // W:211-215
        // W:217-222   Ptr<RotationWarper> warper = warper_creator->create(static_cast<float>(cameras[0].focal));
        std::unique_ptr<RotationWarper> warper(new isx_cv::HipCylindricalWarper(focal));
        for (int i = 0; i < num_images; ++i) {
            corners[i] = warper->warp(imgs[i], K, R[i], 1 /* INTER_LINEAR */, 2 /* BORDER_REFLECT */, images_warped[i]);      // W:229
            sizes[i] = images_warped[i].size();                                          // W:230
            warper->warp(masks[i], K, R[i], 0 /* INTER_NEAREST */, 0 /* BORDER_CONSTANT */, masks_warped[i]);                 // W:232
            printf("corner %d %d %d\n", i, corners[i].x, corners[i].y);
        }
        {   // The reference's own declarations (W:206-207): vector<UMat> masks_warped, images_warped - the warped tiles are UMats, and
            // warp() receives them as OutputArray (W:229, 232); W:148 declares UMat xmap, ymap.  Same bytes as the Mat leg above,
            // through a mapping of the UMat (not through a Mat the caller owns).
            std::vector<UMat> images_warped_u(num_images), masks_warped_u(num_images);
            for (int i = 0; i < num_images; ++i) {
                Point c = warper->warp(imgs[i], K, R[i], 1, 2, images_warped_u[i]);                                         // W:229
                Size sz = images_warped_u[i].size();                                                                       // W:230
                warper->warp(masks[i], K, R[i], 0, 0, masks_warped_u[i]);                                                 // W:232
                Mat iu = images_warped_u[i].getMat(ACCESS_READ), mu = masks_warped_u[i].getMat(ACCESS_READ);
                const bool same = c.x == corners[i].x && c.y == corners[i].y && sz.width == sizes[i].width && sz.height == sizes[i].height &&
                                  iu.type() == CV_8UC3 && mu.type() == CV_8U && mu.rows == masks_warped[i].rows && mu.cols == masks_warped[i].cols &&
                                  memcmp(iu.data, images_warped[i].data, (size_t)iu.rows * iu.step) == 0 &&
                                  memcmp(mu.data, masks_warped[i].data, (size_t)mu.rows * mu.step) == 0;
                if (!same || images_warped_u[i].maps() < 2) return 6;
            }
            UMat xmap_u, ymap_u;                                                                                           // W:148
            Mat xmap_m, ymap_m;
            warper->buildMaps(Size(w, h), K, R[1], xmap_u, ymap_u);
            warper->buildMaps(Size(w, h), K, R[1], xmap_m, ymap_m);
            Mat xu = xmap_u.getMat(ACCESS_READ), yu = ymap_u.getMat(ACCESS_READ);
            if (xu.rows != xmap_m.rows || xu.cols != xmap_m.cols || memcmp(xu.data, xmap_m.data, (size_t)xu.rows * xu.step) != 0 ||
                memcmp(yu.data, ymap_m.data, (size_t)yu.rows * yu.step) != 0) return 6;
            // feed() from a UMat mask (W:302 passes masks_seam[k], a Mat; masks_warped stays a UMat until W:296 copies it) and blend() into UMats
            Ptr<Blender> ub = isx_cv::createDefaultBlender(Blender::NO);
            Ptr<Blender> mbu = isx_cv::createDefaultBlender(Blender::NO);
            ub->prepare(corners, sizes); mbu->prepare(corners, sizes);
            for (int k = 0; k < num_images; ++k) {
                Mat img_s(images_warped[k].rows, images_warped[k].cols, CV_16SC3);
                for (int y = 0; y < img_s.rows; ++y) {
                    const unsigned char* s = images_warped[k].ptr<unsigned char>(y);
                    short* d = img_s.ptr<short>(y);
                    for (int x = 0; x < img_s.cols * 3; ++x) d[x] = s[x];
                }
                ub->feed(img_s, masks_warped_u[k], corners[k]);
                mbu->feed(img_s, masks_warped[k], corners[k]);
            }
            UMat ur, um;
            Mat mr, mm;
            ub->blend(ur, um); mbu->blend(mr, mm);
            Mat urm = ur.getMat(ACCESS_READ), umm = um.getMat(ACCESS_READ);
            if (urm.rows != mr.rows || urm.cols != mr.cols || memcmp(urm.data, mr.data, (size_t)mr.rows * mr.step) != 0 ||
                memcmp(umm.data, mm.data, (size_t)mm.rows * mm.step) != 0) return 6;
            printf("umat-leg OK\n");
        }
        {   // buildMaps / warpRoi of the same interface (W:122): sizes as the stock class reports them
            Mat xmap, ymap;
            Rect r = warper->buildMaps(Size(w, h), K, R[0], xmap, ymap), q = warper->warpRoi(Size(w, h), K, R[0]);
            printf("maps %d %d %d\n", xmap.rows, xmap.cols, xmap.type());
            if (r.x != corners[0].x || r.y != corners[0].y || xmap.rows != images_warped[0].rows || xmap.cols != images_warped[0].cols ||
                q.width != xmap.cols || q.height != xmap.rows || r.width != xmap.cols - 1) return 5;
        }
        // seam masks: the left tile keeps x < mid, the right tile x >= mid of the overlap (stand-in for the seam finder)
        int mid = (corners[1].x + corners[0].x + sizes[0].width) / 2;
        for (int i = 0; i < num_images; ++i)
            for (int y = 0; y < masks_warped[i].rows; ++y) {
                unsigned char* p = masks_warped[i].ptr<unsigned char>(y);
                for (int x = 0; x < masks_warped[i].cols; ++x) {
                    bool keep = i == 0 ? corners[i].x + x < mid : corners[i].x + x >= mid;
                    if (!keep) p[x] = 0;
                }
            }
        // W:271-273   blender = Blender::createDefault(Blender::MULTI_BAND, false); mb->setNumBands(4);
        isx_cv::HipMultiBandBlender* mb = new isx_cv::HipMultiBandBlender(5);
        std::unique_ptr<Blender> blender(mb);
        mb->setNumBands(4);
        blender->prepare(corners, sizes);                                                // W:281
        for (int k = 0; k < num_images; ++k) {
            Mat img_s(images_warped[k].rows, images_warped[k].cols, CV_16SC3);          // convertTo(CV_16S)  W:294
            for (int y = 0; y < img_s.rows; ++y) {
                const unsigned char* s = images_warped[k].ptr<unsigned char>(y);
                short* d = img_s.ptr<short>(y);
                for (int x = 0; x < img_s.cols * 3; ++x) d[x] = s[x];
            }
            blender->feed(img_s, masks_warped[k], corners[k]);                           // W:302
            memset(img_s.data, 0x5a, (size_t)img_s.rows * img_s.step);                   // the fed mat is released at once (W:305-308)
            dump(argv[6], k == 0 ? "warped0" : "warped1", images_warped[k]);
            dump(argv[6], k == 0 ? "mask0" : "mask1", masks_warped[k]);
        }
        Mat result, result_mask;
        blender->blend(result, result_mask);                                             // W:313
        dump(argv[6], "result", result);
        dump(argv[6], "result_mask", result_mask);
        {   // W:276   blender = Blender::createDefault(Blender::NO, false);   (the line the demo runs before it settles on FEATHER)
            Ptr<Blender> nb = isx_cv::createDefaultBlender(Blender::NO);
            nb->prepare(corners, sizes);
            for (int k = 0; k < num_images; ++k) {
                Mat img_s(images_warped[k].rows, images_warped[k].cols, CV_16SC3);
                for (int y = 0; y < img_s.rows; ++y) {
                    const unsigned char* s = images_warped[k].ptr<unsigned char>(y);
                    short* d = img_s.ptr<short>(y);
                    for (int x = 0; x < img_s.cols * 3; ++x) d[x] = s[x];
                }
                nb->feed(img_s, masks_warped[k], corners[k]);
            }
            Mat nr, nm;
            nb->blend(nr, nm);
            dump(argv[6], "no_result", nr);
            dump(argv[6], "no_mask", nm);
        }
        try { blender->feed(result, result_mask, Point(0, 0)); printf("no-throw\n"); return 4; }
        catch (const isx::Exception& e) { printf("throws %d\n", e.code); }
    } catch (const std::exception& e) {
        fprintf(stderr, "exception: %s\n", e.what());
        return 1;
    }
    return 0;
}
