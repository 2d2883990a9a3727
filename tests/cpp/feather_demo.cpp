// The stage every reference demo actually runs after warping (W:241-244, W:278-313), written against the C++ mirror
// include/imagestitch.hpp exactly as the reference's main() writes it against OpenCV:
//   compensator->apply  ->  FeatherBlender(sharpness 0.1)  ->  dilate(seam mask, 20x20) & warped mask  ->
//   convertTo(CV_16S)  ->  prepare / feed / blend  ->  imwrite
// usage: feather_demo <dir> <x0> <y0> <gain0> <x1> <y1> <gain1>      (inputs <dir>/warped{i}.bmp, mask{i}.bmp, seam{i}.bmp)
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "imagestitch.hpp"

using namespace isx;

static Mat gray_of(const Mat& bgr) {   // masks are stored as 8-bit bitmaps; imread (IMREAD_COLOR) expands them to 3 equal channels
    Mat g(bgr.rows(), bgr.cols(), ISX_8UC1);
    for (int y = 0; y < bgr.rows(); ++y)
        for (int x = 0; x < bgr.cols(); ++x) g.ptr<unsigned char>(y)[x] = bgr.ptr<unsigned char>(y)[3 * x];
    return g;
}

int main(int argc, char** argv) {
    if (argc != 8) { std::fprintf(stderr, "usage\n"); return 2; }
    try {
        const std::string dir = argv[1];
        const int num_images = 2;
        std::vector<Point> corners = {Point(std::atoi(argv[2]), std::atoi(argv[3])), Point(std::atoi(argv[5]), std::atoi(argv[6]))};
        const double gains[2] = {std::atof(argv[4]), std::atof(argv[7])};
        std::vector<Mat> images_warped(num_images), masks_warped(num_images), masks_seam(num_images);
        std::vector<Size> sizes(num_images);
        for (int i = 0; i < num_images; ++i) {
            images_warped[i] = imread((dir + "/warped" + std::to_string(i) + ".bmp").c_str());           // W:166 style input
            masks_warped[i] = gray_of(imread((dir + "/mask" + std::to_string(i) + ".bmp").c_str()));
            masks_seam[i] = gray_of(imread((dir + "/seam" + std::to_string(i) + ".bmp").c_str()));
            sizes[i] = images_warped[i].size();
            gainApply(images_warped[i], gains[i]);                                                            // W:241-244
        }
        std::shared_ptr<Blender> blender = Blender::createDefault(Blender::FEATHER, false);                   // W:278
        FeatherBlender* fb = dynamic_cast<FeatherBlender*>(blender.get());                                    // W:279
        fb->setSharpness(0.1f);                                                                                // W:280
        blender->prepare(corners, sizes);                                                                      // W:281
        for (int img_idx = 0; img_idx < num_images; ++img_idx) {
            Mat img_warped_s(images_warped[img_idx].rows(), images_warped[img_idx].cols(), ISX_16SC3);        // convertTo(CV_16S) W:294
            for (int y = 0; y < img_warped_s.rows(); ++y)
                for (int x = 0; x < 3 * img_warped_s.cols(); ++x) img_warped_s.ptr<short>(y)[x] = images_warped[img_idx].ptr<unsigned char>(y)[x];
            Mat mask_warped;
            dilateAnd(masks_seam[img_idx], 20, 20, &masks_warped[img_idx], mask_warped);                        // W:295-301
            blender->feed(img_warped_s, mask_warped, corners[img_idx]);                                        // W:302
        }
        Mat result, result_mask;
        blender->blend(result, result_mask);                                                                   // W:313
        Mat result8(result.rows(), result.cols(), ISX_8UC3);                                                   // imwrite's convertTo(CV_8U), W:315
        for (int y = 0; y < result.rows(); ++y)
            for (int x = 0; x < 3 * result.cols(); ++x) {
                const int v = result.ptr<short>(y)[x];
                result8.ptr<unsigned char>(y)[x] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
        imwrite((dir + "/pano.bmp").c_str(), result8);
        imwrite((dir + "/pano.jpg").c_str(), result8);                                                         // S:1282 imwrite("pano.jpg", result)
        imwrite((dir + "/pano_mask.bmp").c_str(), result_mask);
        FILE* f = std::fopen((dir + "/pano_s16.raw").c_str(), "wb");
        for (int y = 0; y < result.rows(); ++y) std::fwrite(result.ptr<short>(y), 2, (size_t)3 * result.cols(), f);
        std::fclose(f);
        std::printf("result %d %d\n", result.rows(), result.cols());
        try { imread((dir + "/does_not_exist.bmp").c_str()); } catch (const Exception& e) { std::printf("throws %d\n", e.code); }
    } catch (const Exception& e) {
        std::fprintf(stderr, "isx::Exception %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
