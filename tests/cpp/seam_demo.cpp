// The second half of the reference's DP-seam demo main() (S:1173-1283), written against the C++ mirror include/imagestitch.hpp
// exactly as the reference writes it against OpenCV:
//   masks_seam = copies of masks_warped -> convertTo(CV_32F) -> find(images_warped_f, corners, masks_seam) ->
//   imwrite("mask_seam[k].bmp") -> FeatherBlender(0.1), dilate 20x20 & masks_warped, convertTo(CV_16S), feed, blend -> imwrite
// usage: seam_demo <dir> <x0> <y0> <x1> <y1>     (inputs <dir>/images_warped[k].bmp, mask_warped[k].bmp)
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "imagestitch.hpp"

using namespace isx;

static Mat gray_of(const Mat& bgr) {
    Mat g(bgr.rows(), bgr.cols(), ISX_8UC1);
    for (int y = 0; y < bgr.rows(); ++y)
        for (int x = 0; x < bgr.cols(); ++x) g.ptr<unsigned char>(y)[x] = bgr.ptr<unsigned char>(y)[3 * x];
    return g;
}

int main(int argc, char** argv) {
    if (argc != 6) { std::fprintf(stderr, "usage\n"); return 2; }
    try {
        const std::string dir = argv[1];
        const int num_images = 2;
        std::vector<Point> corners = {Point(std::atoi(argv[2]), std::atoi(argv[3])), Point(std::atoi(argv[4]), std::atoi(argv[5]))};
        std::vector<Mat> images_warped(num_images), masks_warped(num_images), masks_seam(num_images), images_warped_f(num_images);
        std::vector<Size> sizes(num_images);
        for (int i = 0; i < num_images; ++i) {
            images_warped[i] = imread((dir + "/images_warped[" + std::to_string(i) + "].bmp").c_str());
            masks_warped[i] = gray_of(imread((dir + "/mask_warped[" + std::to_string(i) + "].bmp").c_str()));
            sizes[i] = images_warped[i].size();
            masks_seam[i] = Mat(masks_warped[i].rows(), masks_warped[i].cols(), ISX_8UC1);                       // masks_warped[i].copyTo(masks_seam[i]), S:1175-1176
            for (int y = 0; y < masks_seam[i].rows(); ++y)
                for (int x = 0; x < masks_seam[i].cols(); ++x) masks_seam[i].ptr<unsigned char>(y)[x] = masks_warped[i].ptr<unsigned char>(y)[x];
            images_warped_f[i] = Mat(images_warped[i].rows(), images_warped[i].cols(), ISX_32FC3);                // convertTo(CV_32F), S:1188-1190
            for (int y = 0; y < images_warped[i].rows(); ++y)
                for (int x = 0; x < 3 * images_warped[i].cols(); ++x) images_warped_f[i].ptr<float>(y)[x] = images_warped[i].ptr<unsigned char>(y)[x];
        }
        DpSeamFinder seam_finder;
        seam_finder.find(images_warped_f, corners, masks_seam);                                                       // S:1192
        for (int i = 0; i < num_images; ++i) imwrite((dir + "/mask_seam[" + std::to_string(i) + "].bmp").c_str(), masks_seam[i]);   // S:1197-1198
        std::shared_ptr<Blender> blender = Blender::createDefault(Blender::FEATHER, false);                           // S:1238
        dynamic_cast<FeatherBlender*>(blender.get())->setSharpness(0.1f);                                             // S:1239-1240
        blender->prepare(corners, sizes);                                                                             // S:1241
        for (int k = 0; k < num_images; ++k) {
            Mat images_warped_s(images_warped[k].rows(), images_warped[k].cols(), ISX_16SC3);                       // convertTo(CV_16S), S:1250
            for (int y = 0; y < images_warped_s.rows(); ++y)
                for (int x = 0; x < 3 * images_warped_s.cols(); ++x) images_warped_s.ptr<short>(y)[x] = (short)images_warped_f[k].ptr<float>(y)[x];
            Mat c;
            dilateAnd(masks_seam[k], 20, 20, &masks_warped[k], c);                                                    // S:1251-1256
            blender->feed(images_warped_s, c, corners[k]);                                                            // S:1257
        }
        Mat result, result_mask;
        blender->blend(result, result_mask);                                                                          // S:1281
        Mat result8(result.rows(), result.cols(), ISX_8UC3);
        for (int y = 0; y < result.rows(); ++y)
            for (int x = 0; x < 3 * result.cols(); ++x) {
                const int v = result.ptr<short>(y)[x];
                result8.ptr<unsigned char>(y)[x] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
        imwrite((dir + "/pano.bmp").c_str(), result8);                                                                // S:1283 (as .bmp)
        std::printf("result %d %d\n", result.rows(), result.cols());
    } catch (const Exception& e) {
        std::fprintf(stderr, "isx::Exception %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
