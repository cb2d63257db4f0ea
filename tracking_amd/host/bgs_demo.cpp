// bgs_demo.cpp — the build's own small harness in the pattern of the reference's Demo2.cpp:142-168 (frames/N.png loop) and
// Main.cpp:63-72 (hard failures surface as one std::exception).  OpenCV is absent, so frames come from a raw file:
//     bgs_demo <frames.raw> <rows> <cols> <n_frames> <out_prefix> [ustc_type]
// frames.raw = n_frames x rows x cols x 3 bytes (BGR).  For every class enabled in ./config/FrameProcessor.xml the mask of
// each frame is appended to <out_prefix>.<ClassName>.raw (frames whose output the class leaves untouched are written as 0x07).
// With a 6th argument the frames go through USTC_BGS(type) instead (ustc_src/ustc_bgs.cpp) and GetMask() of every frame is
// written to <out_prefix>.ustc.raw.  With a 7th argument ("box" or "moments") the detector is a HipFGDetector and every frame's blob
// list (GetBlobs) is printed like ustc_src/trackingMain.cpp:189-190 prints the tracker's: "pBlob x,y,w,h,id is ...".
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

#include "FrameProcessor.h"
#include "ustc_bgs.h"

using namespace bgs_hip;

static void dump(std::ofstream& f, const Image& m, int rows, int cols) {
  std::vector<uint8_t> untouched((size_t)rows * cols, 7);
  if (m.empty()) {
    f.write((const char*)untouched.data(), untouched.size());
    return;
  }
  for (int y = 0; y < m.rows; ++y) f.write((const char*)m.ptr(y), m.cols);
}

int main(int argc, char** argv) {
  if (argc < 6) {
    std::fprintf(stderr, "usage: %s frames.raw rows cols n_frames out_prefix\n", argv[0]);
    return 2;
  }
  const int rows = std::atoi(argv[2]), cols = std::atoi(argv[3]), n = std::atoi(argv[4]);
  const std::string prefix = argv[5];
  try {
    std::ifstream in(argv[1], std::ios::binary);
    if (!in) throw Exception(BGS_ERR_INVALID, std::string("cannot open ") + argv[1]);
    if (argc >= 7) {  // the tracker's FG detector: Process(frame) then GetMask(), trackingMain.cpp:152-166
      HipFGDetector fg(std::atoi(argv[6]));
      const bool want_blobs = argc >= 8, from_moments = want_blobs && std::string(argv[7]) == "moments";
      std::ofstream out((prefix + ".ustc.raw").c_str(), std::ios::binary);
      Image frame(rows, cols, 3);
      CvBlobSeq blobs;
      for (int t = 0; t < n; ++t) {
        in.read((char*)frame.data, (size_t)rows * cols * 3);
        if (!in) throw Exception(BGS_ERR_INVALID, "short read on frame file");
        fg.Process(frame);
        const Image* m = fg.GetMask();
        dump(out, m ? *m : Image(), rows, cols);
        if (want_blobs) {
          fg.GetBlobs(&blobs, 8, from_moments);
          std::printf("frame %d blobs %d\n", t, blobs.GetBlobNum());
          for (int i = blobs.GetBlobNum(); i > 0; i--) {  // trackingMain.cpp:184-190
            CvBlob* pBlob = blobs.GetBlob(i - 1);
            std::printf("pBlob x,y,w,h,id is %.9g , %.9g , %.9g , %.9g , %d\n", pBlob->x, pBlob->y, pBlob->w, pBlob->h, pBlob->ID);
          }
        }
      }
      fg.Release();
      return 0;
    }
    FrameProcessor* fp = new FrameProcessor;
    fp->init();
    FramePrep prep;  // VideoCapture's resize / flip / ROI from ./config/VideoCapture.xml (defaults: none); BGSLIB_RAW geometry in, prepared out
    const char* names[] = {"FrameDifferenceBGS", "StaticFrameDifferenceBGS", "WeightedMovingMeanBGS", "WeightedMovingVarianceBGS",
                           "MixtureOfGaussianV1BGS", "MixtureOfGaussianV2BGS", "AdaptiveBackgroundLearning", "AdaptiveSelectiveBackgroundLearning",
                           "GMG", "DPAdaptiveMedianBGS", "DPGrimsonGMMBGS", "DPZivkovicAGMMBGS", "DPMeanBGS", "DPWrenGABGS", "SigmaDeltaBGS", "SuBSENSEBGS", "LOBSTERBGS"};
    Image* masks[] = {&fp->img_framediff, &fp->img_staticfdiff, &fp->img_wmovmean, &fp->img_movvar, &fp->img_mog1, &fp->img_mog2, &fp->img_bkgl_fgmask, &fp->img_asbl,
                     &fp->img_gmg, &fp->img_adpmed, &fp->img_grigmm, &fp->img_zivgmm, &fp->img_tmpmean, &fp->img_wrenga, &fp->img_sdbgs, &fp->img_ssbgs, &fp->img_lobgs};
    std::vector<std::ofstream> outs;
    for (const char* nm : names) outs.emplace_back((prefix + "." + nm + ".raw").c_str(), std::ios::binary);
    Image frame(rows, cols, 3);
    for (int t = 0; t < n; ++t) {
      in.read((char*)frame.data, (size_t)rows * cols * 3);
      if (!in) throw Exception(BGS_ERR_INVALID, "short read on frame file");
      Image img_input;
      prep.process(frame, img_input);  // VideoCapture.cpp:164-203
      fp->process(img_input);
      for (size_t i = 0; i < outs.size(); ++i) dump(outs[i], *masks[i], img_input.rows, img_input.cols);
    }
    fp->finish();
    delete fp;
  } catch (const std::exception& ex) {  // Main.cpp:63-72
    std::cout << "std::exception:" << ex.what() << std::endl;
    return 1;
  } catch (...) {
    std::cout << "Unknow error" << std::endl;
    return 1;
  }
  return 0;
}
