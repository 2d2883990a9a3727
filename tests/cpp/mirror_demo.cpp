// mirror_demo.cpp — the reference's main() hot-path lines (W:217-233, W:271-313) written against
// include/imagestitch.hpp.  Built with plain g++ and linked to libimagestitch_hip.so by
// tests/test_gpu_cpp_mirror.py, which compares the files it writes with the CPU oracle.
//   usage: mirror_demo <w> <h> <focal> <in0.raw> <in1.raw> <out_prefix>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "imagestitch.hpp"

static void rot(double yaw, double pitch, double roll, float R[9]) {
    double cy = cos(yaw), sy = sin(yaw), cp = cos(pitch), sp = sin(pitch), cr = cos(roll), sr = sin(roll);
    double Ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy}, Rx[9] = {1, 0, 0, 0, cp, -sp, 0, sp, cp}, Rz[9] = {cr, -sr, 0, sr, cr, 0, 0, 0, 1};
    double T[9], O[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { T[i * 3 + j] = 0; for (int k = 0; k < 3; ++k) T[i * 3 + j] += Ry[i * 3 + k] * Rx[k * 3 + j]; }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { O[i * 3 + j] = 0; for (int k = 0; k < 3; ++k) O[i * 3 + j] += T[i * 3 + k] * Rz[k * 3 + j]; }
    for (int i = 0; i < 9; ++i) R[i] = (float)O[i];
}

static void dump(const char* prefix, const char* name, const isx::Mat& m) {
    char path[512];
    snprintf(path, sizeof(path), "%s_%s.raw", prefix, name);
    FILE* f = fopen(path, "wb");
    for (int y = 0; y < m.rows(); ++y) fwrite(m.ptr<unsigned char>(y), 1, (size_t)m.cols() * isx::Mat::elemSize(m.type()), f);
    fclose(f);
    printf("%s %d %d %d\n", name, m.rows(), m.cols(), m.type());
}

int main(int argc, char** argv) {
    if (argc < 7) return 2;
    int w = atoi(argv[1]), h = atoi(argv[2]);
    float focal = (float)atof(argv[3]);
    try {
        const int num_images = 2;
        std::vector<isx::Mat> imgs(num_images);
        for (int i = 0; i < num_images; ++i) {
            imgs[i].create(h, w, ISX_8UC3);
            FILE* f = fopen(argv[4 + i], "rb");
            if (!f || fread(imgs[i].ptr<unsigned char>(0), 1, (size_t)w * h * 3, f) != (size_t)w * h * 3) return 3;
            fclose(f);
        }
        float K[9] = {focal, 0, w / 2.0f, 0, focal, h / 2.0f, 0, 0, 1};
        float R[2][9];
        rot(-0.36, 0.010, 0.005, R[0]);
        rot(0.36, 0.010, 0.005, R[1]);
        std::vector<isx::Point> corners(num_images);                                   // W:206-210
        std::vector<isx::Mat> masks_warped(num_images), images_warped(num_images), masks(num_images);
        std::vector<isx::Size> sizes(num_images);
        for (int i = 0; i < num_images; ++i) { masks[i].create(h, w, ISX_8UC1); masks[i].setTo(255); }   // W:211-215
        isx::CylindricalWarper warper_creator;                                          // W:219
        auto warper = warper_creator.create(focal);                                     // W:222
        for (int i = 0; i < num_images; ++i) {
            corners[i] = warper->warp(imgs[i], K, R[i], isx::INTER_LINEAR, isx::BORDER_REFLECT, images_warped[i]);   // W:229
            sizes[i] = images_warped[i].size();                                         // W:230
            warper->warp(masks[i], K, R[i], isx::INTER_NEAREST, isx::BORDER_CONSTANT, masks_warped[i]);           // W:232
            printf("corner %d %d %d\n", i, corners[i].x, corners[i].y);
        }
        // seam masks: the left tile keeps x < mid, the right tile x >= mid of the overlap (stand-in for the seam finder)
        int mid = (corners[1].x + corners[0].x + sizes[0].width) / 2;
        for (int i = 0; i < num_images; ++i)
            for (int y = 0; y < masks_warped[i].rows(); ++y) {
                unsigned char* p = masks_warped[i].ptr<unsigned char>(y);
                for (int x = 0; x < masks_warped[i].cols(); ++x) {
                    bool keep = i == 0 ? corners[i].x + x < mid : corners[i].x + x >= mid;
                    if (!keep) p[x] = 0;
                }
            }
        auto blender = isx::Blender::createDefault(isx::Blender::MULTI_BAND, false);   // W:271
        static_cast<isx::MultiBandBlender*>(blender.get())->setNumBands(4);             // W:272-273
        blender->prepare(corners, sizes);                                               // W:281
        for (int k = 0; k < num_images; ++k) {
            isx::Mat img_s(images_warped[k].rows(), images_warped[k].cols(), ISX_16SC3);   // convertTo(CV_16S)  W:294
            for (int y = 0; y < img_s.rows(); ++y) {
                const unsigned char* s = images_warped[k].ptr<unsigned char>(y);
                short* d = img_s.ptr<short>(y);
                for (int x = 0; x < img_s.cols() * 3; ++x) d[x] = s[x];
            }
            blender->feed(img_s, masks_warped[k], corners[k]);                          // W:302
            dump(argv[6], k == 0 ? "warped0" : "warped1", images_warped[k]);
            dump(argv[6], k == 0 ? "mask0" : "mask1", masks_warped[k]);
        }
        isx::Mat result, result_mask;
        blender->blend(result, result_mask);                                            // W:313
        dump(argv[6], "result", result);
        dump(argv[6], "result_mask", result_mask);
        // the same mosaic cut into two column strips (isx_blender_set_window): what two GPUs would each compute of ONE panorama
        for (int strip = 0; strip < 2; ++strip) {
            blender->prepare(corners, sizes);
            blender->setDeferredLevel0(2);
            blender->setWindow(strip * 3 * ISX_WINDOW_GRANULE, (strip + 1) * 3 * ISX_WINDOW_GRANULE);
            for (int k = 0; k < num_images; ++k) {
                isx::Mat img_s(images_warped[k].rows(), images_warped[k].cols(), ISX_16SC3);
                for (int y = 0; y < img_s.rows(); ++y) {
                    const unsigned char* s = images_warped[k].ptr<unsigned char>(y);
                    short* d = img_s.ptr<short>(y);
                    for (int x = 0; x < img_s.cols() * 3; ++x) d[x] = s[x];
                }
                blender->feed(img_s, masks_warped[k], corners[k]);
            }
            isx::Mat part, part_mask;
            blender->blend(part, part_mask);
            dump(argv[6], strip == 0 ? "strip0" : "strip1", part);
            dump(argv[6], strip == 0 ? "stripmask0" : "stripmask1", part_mask);
        }
        blender->setWindow(0, 0);
        // two mosaics of the rig in ONE chain of launches (Blender::blendBatch), fed the CV_8UC3 tiles as they come out of warp()
        {
            auto b0 = isx::Blender::createDefault(isx::Blender::MULTI_BAND, false), b1 = isx::Blender::createDefault(isx::Blender::MULTI_BAND, false);
            for (isx::Blender* b : {b0.get(), b1.get()}) {
                static_cast<isx::MultiBandBlender*>(b)->setNumBands(4);
                b->setDeferredLevel0(2);
                b->prepare(corners, sizes);
                for (int k = 0; k < num_images; ++k) b->feed(images_warped[k], masks_warped[k], corners[k]);
            }
            std::vector<isx::Mat> outs, out_masks;
            isx::Blender::blendBatch({b0.get(), b1.get()}, outs, out_masks);
            dump(argv[6], "batch0", outs[0]);
            dump(argv[6], "batch1", outs[1]);
            dump(argv[6], "batchmask1", out_masks[1]);
        }
        {   // W:276   Blender::createDefault(Blender::NO, false), the tiles converted with convertTo(CV_16S) (W:294)
            auto nb = isx::Blender::createDefault(isx::Blender::NO, false);
            nb->prepare(corners, sizes);
            for (int k = 0; k < num_images; ++k) {
                isx::Mat img_s;
                isx::convertTo(images_warped[k], img_s, ISX_16SC3);
                nb->feed(img_s, masks_warped[k], corners[k]);
            }
            isx::Mat nr, nm;
            nb->blend(nr, nm);
            dump(argv[6], "no_result", nr);
            dump(argv[6], "no_mask", nm);
        }
        // error behaviour: feed after blend must throw like a CV_Assert would
        try { blender->feed(result, result_mask, isx::Point(0, 0)); printf("no-throw\n"); return 4; }
        catch (const isx::Exception& e) { printf("throws %d\n", e.code); }
    } catch (const isx::Exception& e) {
        fprintf(stderr, "isx::Exception %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
