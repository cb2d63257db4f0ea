// FrameProcessor.h — host-side mirror of bgslibrary::FrameProcessor (FrameProcessor.{h,cpp}) for the classes on the hot path.
//
//   IFrameProcessor::process(const Image&)            IFrameProcessor.h:23-28
//   FrameProcessor::init / process / finish           FrameProcessor.cpp:35-155, 169-340, 342-482
//   per-algorithm helper with tic/toc                 FrameProcessor.cpp:157-167, 484-494
//   ./config/FrameProcessor.xml (tictoc + enable*)    FrameProcessor.cpp:496-610
//
// Same control flow as the reference: one IBGS instance per enabled class, created in init(), fed the same
// pre-processed frame in the reference's order, one mask per class kept in a member image.  The PreProcessor (PreProcessor.h:
// copy, optional equalizeHist / GaussianBlur from ./config/PreProcessor.xml) is created in init() like the reference's (:37-38).
#pragma once
#include <chrono>
#include <string>

#include "bgs_host.h"
#include "PreProcessor.h"

namespace bgs_hip {

class IFrameProcessor {  // IFrameProcessor.h:23-28
 public:
  virtual void process(const Image& input) = 0;
  virtual ~IFrameProcessor() {}
};

class FrameProcessor : public IFrameProcessor {
 public:
  FrameProcessor();
  ~FrameProcessor() override;
  long frameToStop;
  std::string imgref;

  void init();
  void process(const Image& img_input) override;
  void finish();

  // the masks the reference keeps as img_framediff, img_staticfdiff, ... (FrameProcessor.h:99-170)
  Image img_prep, img_framediff, img_staticfdiff, img_wmovmean, img_movvar, img_mog1, img_mog2, img_bkgl_fgmask, img_asbl;
  Image img_gmg, img_adpmed, img_grigmm, img_zivgmm, img_tmpmean, img_wrenga, img_sdbgs, img_ssbgs, img_lobgs;  // FrameProcessor.h:120-236
  double lastDuration() const { return duration; }
  int groupedClasses() const { return (int)grouped_.size(); }  // how many classes share the fused launch (0: none)

 private:
  bool firstTime;
  long frameNumber;
  std::string processname;
  double duration;
  std::chrono::steady_clock::time_point t0;
  std::string tictoc;

  PreProcessor* preProcessor;
  bool enablePreProcessor;
  FrameDifferenceBGS* frameDifference;
  bool enableFrameDifferenceBGS;
  StaticFrameDifferenceBGS* staticFrameDifference;
  bool enableStaticFrameDifferenceBGS;
  WeightedMovingMeanBGS* weightedMovingMean;
  bool enableWeightedMovingMeanBGS;
  WeightedMovingVarianceBGS* weightedMovingVariance;
  bool enableWeightedMovingVarianceBGS;
  MixtureOfGaussianV1BGS* mixtureOfGaussianV1BGS;
  bool enableMixtureOfGaussianV1BGS;
  MixtureOfGaussianV2BGS* mixtureOfGaussianV2BGS;
  bool enableMixtureOfGaussianV2BGS;
  AdaptiveBackgroundLearning* adaptiveBackgroundLearning;
  bool enableAdaptiveBackgroundLearning;
  GMG* gmg;
  bool enableGMG;
  DPAdaptiveMedianBGS* adaptiveMedian;
  bool enableDPAdaptiveMedianBGS;
  DPGrimsonGMMBGS* grimsonGMM;
  bool enableDPGrimsonGMMBGS;
  DPZivkovicAGMMBGS* zivkovicAGMM;
  bool enableDPZivkovicAGMMBGS;
  DPMeanBGS* temporalMean;
  bool enableDPMeanBGS;
  DPWrenGABGS* wrenGA;
  bool enableDPWrenGABGS;
  SigmaDeltaBGS* sdbgs;
  bool enableSigmaDeltaBGS;
  SuBSENSEBGS* ssbgs;
  bool enableSuBSENSEBGS;
  LOBSTERBGS* lobgs;
  bool enableLOBSTERBGS;
  AdaptiveSelectiveBackgroundLearning* adaptiveSelectiveBackgroundLearning;  // not in the reference's FrameProcessor (Demo.cpp / USTC_BGS type 7 only)
  bool enableAdaptiveSelectiveBackgroundLearning;

  // the enabled byte-stream classes share ONE launch per frame (bgs_group) when there are at least two of them
  bgs_group* group_ = nullptr;
  std::vector<HipBGSBase*> grouped_;
  void runGroup(const Image& img_input);

  void process(std::string name, IBGS* bgs, const Image& img_input, Image& img_bgs);
  void tic(std::string value);
  void toc();
  void saveConfig();
  void loadConfig();
};

}  // namespace bgs_hip
