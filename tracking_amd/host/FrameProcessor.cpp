// FrameProcessor.cpp — see FrameProcessor.h.  Statement order follows FrameProcessor.cpp of the reference.
#include "FrameProcessor.h"

#include <cstdlib>
#include <iomanip>

namespace bgs_hip {

FrameProcessor::FrameProcessor()
    : frameToStop(0), firstTime(true), frameNumber(0), duration(0), tictoc(""), preProcessor(nullptr), enablePreProcessor(true), frameDifference(nullptr),
      enableFrameDifferenceBGS(false), staticFrameDifference(nullptr), enableStaticFrameDifferenceBGS(false), weightedMovingMean(nullptr),
      enableWeightedMovingMeanBGS(false), weightedMovingVariance(nullptr), enableWeightedMovingVarianceBGS(false), mixtureOfGaussianV1BGS(nullptr),
      enableMixtureOfGaussianV1BGS(false), mixtureOfGaussianV2BGS(nullptr), enableMixtureOfGaussianV2BGS(false), adaptiveBackgroundLearning(nullptr),
      enableAdaptiveBackgroundLearning(false), adaptiveSelectiveBackgroundLearning(nullptr), enableAdaptiveSelectiveBackgroundLearning(false) {
  gmg = nullptr, enableGMG = false;
  adaptiveMedian = nullptr, enableDPAdaptiveMedianBGS = false;
  grimsonGMM = nullptr, enableDPGrimsonGMMBGS = false;
  zivkovicAGMM = nullptr, enableDPZivkovicAGMMBGS = false;
  temporalMean = nullptr, enableDPMeanBGS = false;
  wrenGA = nullptr, enableDPWrenGABGS = false;
  sdbgs = nullptr, enableSigmaDeltaBGS = false;
  ssbgs = nullptr, enableSuBSENSEBGS = false;
  lobgs = nullptr, enableLOBSTERBGS = false;

  std::cout << "FrameProcessor()" << std::endl;
  loadConfig();  // FrameProcessor.cpp:26-27
  saveConfig();
}

FrameProcessor::~FrameProcessor() { std::cout << "~FrameProcessor()" << std::endl; }

void FrameProcessor::init() {  // FrameProcessor.cpp:35-155
  if (enablePreProcessor) preProcessor = new PreProcessor;
  if (enableFrameDifferenceBGS) frameDifference = new FrameDifferenceBGS;
  if (enableStaticFrameDifferenceBGS) staticFrameDifference = new StaticFrameDifferenceBGS;
  if (enableWeightedMovingMeanBGS) weightedMovingMean = new WeightedMovingMeanBGS;
  if (enableWeightedMovingVarianceBGS) weightedMovingVariance = new WeightedMovingVarianceBGS;
  if (enableMixtureOfGaussianV1BGS) mixtureOfGaussianV1BGS = new MixtureOfGaussianV1BGS;
  if (enableMixtureOfGaussianV2BGS) mixtureOfGaussianV2BGS = new MixtureOfGaussianV2BGS;
  if (enableAdaptiveBackgroundLearning) adaptiveBackgroundLearning = new AdaptiveBackgroundLearning;
  if (enableGMG) gmg = new GMG;
  if (enableDPAdaptiveMedianBGS) adaptiveMedian = new DPAdaptiveMedianBGS;
  if (enableDPGrimsonGMMBGS) grimsonGMM = new DPGrimsonGMMBGS;
  if (enableDPZivkovicAGMMBGS) zivkovicAGMM = new DPZivkovicAGMMBGS;
  if (enableDPMeanBGS) temporalMean = new DPMeanBGS;
  if (enableDPWrenGABGS) wrenGA = new DPWrenGABGS;
  if (enableSigmaDeltaBGS) sdbgs = new SigmaDeltaBGS;
  if (enableSuBSENSEBGS) ssbgs = new SuBSENSEBGS;
  if (enableLOBSTERBGS) lobgs = new LOBSTERBGS;
  if (enableAdaptiveSelectiveBackgroundLearning) adaptiveSelectiveBackgroundLearning = new AdaptiveSelectiveBackgroundLearning;
  // Every enabled class gets the same pre-processed frame (:169-340).  The byte-stream classes among them run as ONE fused launch
  // over one upload and one read of that frame (bgs_group) when at least two are enabled; the reference's call shape - one
  // process(name, bgs, img_prep, img_xxx) per class, in its order - stays, each call then just delivers its class's share.
  // BGS_HOST_NO_GROUP=1 keeps every class on its own engine (A/B).
  HipBGSBase* cand[] = {frameDifference, staticFrameDifference, weightedMovingMean, weightedMovingVariance, adaptiveBackgroundLearning, sdbgs};
  for (HipBGSBase* c : cand)
    if (c) grouped_.push_back(c);
  const char* no_group = std::getenv("BGS_HOST_NO_GROUP");
  if (grouped_.size() < 2 || (no_group && no_group[0] == '1')) {
    grouped_.clear();
  } else {
    std::vector<bgs_algo> algos;
    for (HipBGSBase* c : grouped_) algos.push_back(c->algo());
    int rc = bgs_group_create(algos.data(), nullptr, (int)algos.size(), 0, 1, &group_);
    if (rc) throw Exception(rc, std::string("FrameProcessor: ") + bgs_last_error());
    for (size_t i = 0; i < grouped_.size(); ++i) grouped_[i]->attachGroup(group_, (int)i);
    std::cout << "FrameProcessor: " << grouped_.size() << " byte-stream classes share one launch per frame" << std::endl;
  }
}

void FrameProcessor::runGroup(const Image& img_input) {
  if (!group_ || img_input.empty()) return;
  const size_t n = grouped_.size();
  std::vector<uint8_t*> fg(n), bg(n);
  std::vector<size_t> fgs(n), bgs(n);
  std::vector<uint32_t> flags(n, 0);
  for (size_t i = 0; i < n; ++i) {
    grouped_[i]->groupPrepare(img_input);
    grouped_[i]->groupBuffers(&fg[i], &fgs[i], &bg[i], &bgs[i]);
  }
  int rc = bgs_group_process(group_, img_input.data, img_input.rows, img_input.cols, img_input.channels(), img_input.step, fg.data(), fgs.data(), bg.data(), bgs.data(),
                             flags.data());
  if (rc) throw Exception(rc, std::string("FrameProcessor: ") + bgs_last_error());
  for (size_t i = 0; i < n; ++i) grouped_[i]->groupDone(flags[i]);
}

void FrameProcessor::process(std::string name, IBGS* bgs, const Image& img_input, Image& img_bgs) {  // :157-167
  if (tictoc == name) tic(name);
  Image img_bkgmodel;
  bgs->process(img_input, img_bgs, img_bkgmodel);
  if (tictoc == name) toc();
}

void FrameProcessor::process(const Image& img_input) {  // :169-340
  frameNumber++;
  if (enablePreProcessor) preProcessor->process(img_input, img_prep);  // :173-174
  // with enablePreProcessor = 0 img_prep stays empty and every class returns at `if(img_input.empty()) return;`
  runGroup(img_prep);
  if (enableFrameDifferenceBGS) process("FrameDifferenceBGS", frameDifference, img_prep, img_framediff);
  if (enableStaticFrameDifferenceBGS) process("StaticFrameDifferenceBGS", staticFrameDifference, img_prep, img_staticfdiff);
  if (enableWeightedMovingMeanBGS) process("WeightedMovingMeanBGS", weightedMovingMean, img_prep, img_wmovmean);
  if (enableWeightedMovingVarianceBGS) process("WeightedMovingVarianceBGS", weightedMovingVariance, img_prep, img_movvar);
  if (enableMixtureOfGaussianV1BGS) process("MixtureOfGaussianV1BGS", mixtureOfGaussianV1BGS, img_prep, img_mog1);
  if (enableMixtureOfGaussianV2BGS) process("MixtureOfGaussianV2BGS", mixtureOfGaussianV2BGS, img_prep, img_mog2);
  if (enableAdaptiveBackgroundLearning) process("AdaptiveBackgroundLearning", adaptiveBackgroundLearning, img_prep, img_bkgl_fgmask);
  if (enableGMG) process("GMG", gmg, img_prep, img_gmg);
  if (enableDPAdaptiveMedianBGS) process("DPAdaptiveMedianBGS", adaptiveMedian, img_prep, img_adpmed);
  if (enableDPGrimsonGMMBGS) process("DPGrimsonGMMBGS", grimsonGMM, img_prep, img_grigmm);
  if (enableDPZivkovicAGMMBGS) process("DPZivkovicAGMMBGS", zivkovicAGMM, img_prep, img_zivgmm);
  if (enableDPMeanBGS) process("DPMeanBGS", temporalMean, img_prep, img_tmpmean);
  if (enableDPWrenGABGS) process("DPWrenGABGS", wrenGA, img_prep, img_wrenga);
  if (enableSigmaDeltaBGS) process("SigmaDeltaBGS", sdbgs, img_prep, img_sdbgs);
  if (enableSuBSENSEBGS) process("SuBSENSEBGS", ssbgs, img_prep, img_ssbgs);
  if (enableLOBSTERBGS) process("LOBSTERBGS", lobgs, img_prep, img_lobgs);
  if (enableAdaptiveSelectiveBackgroundLearning)
    process("AdaptiveSelectiveBackgroundLearning", adaptiveSelectiveBackgroundLearning, img_prep, img_asbl);
  firstTime = false;
}

void FrameProcessor::finish() {  // :342-482 (reverse order of init)
  delete adaptiveSelectiveBackgroundLearning, adaptiveSelectiveBackgroundLearning = nullptr;
  delete lobgs, lobgs = nullptr;
  delete ssbgs, ssbgs = nullptr;
  delete sdbgs, sdbgs = nullptr;
  delete wrenGA, wrenGA = nullptr;
  delete temporalMean, temporalMean = nullptr;
  delete zivkovicAGMM, zivkovicAGMM = nullptr;
  delete grimsonGMM, grimsonGMM = nullptr;
  delete adaptiveMedian, adaptiveMedian = nullptr;
  delete gmg, gmg = nullptr;
  delete adaptiveBackgroundLearning, adaptiveBackgroundLearning = nullptr;
  delete mixtureOfGaussianV2BGS, mixtureOfGaussianV2BGS = nullptr;
  delete mixtureOfGaussianV1BGS, mixtureOfGaussianV1BGS = nullptr;
  delete weightedMovingVariance, weightedMovingVariance = nullptr;
  delete weightedMovingMean, weightedMovingMean = nullptr;
  delete staticFrameDifference, staticFrameDifference = nullptr;
  delete frameDifference, frameDifference = nullptr;
  delete preProcessor, preProcessor = nullptr;  // :480-481
  if (group_) bgs_group_destroy(group_), group_ = nullptr;
  grouped_.clear();
}

void FrameProcessor::tic(std::string value) {  // :484-488
  processname = value;
  t0 = std::chrono::steady_clock::now();
}

void FrameProcessor::toc() {  // :490-494, same line format
  duration = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::cout << processname << "\ttime(sec):" << std::fixed << std::setprecision(6) << duration << std::endl;
}

void FrameProcessor::saveConfig() {  // :496-552 (keys of the classes this build provides)
  XmlConfig fs;
  fs.beginWrite();
  fs.writeString("tictoc", tictoc);
  fs.writeInt("enablePreProcessor", enablePreProcessor);
  fs.writeInt("enableFrameDifferenceBGS", enableFrameDifferenceBGS);
  fs.writeInt("enableStaticFrameDifferenceBGS", enableStaticFrameDifferenceBGS);
  fs.writeInt("enableWeightedMovingMeanBGS", enableWeightedMovingMeanBGS);
  fs.writeInt("enableWeightedMovingVarianceBGS", enableWeightedMovingVarianceBGS);
  fs.writeInt("enableMixtureOfGaussianV1BGS", enableMixtureOfGaussianV1BGS);
  fs.writeInt("enableMixtureOfGaussianV2BGS", enableMixtureOfGaussianV2BGS);
  fs.writeInt("enableAdaptiveBackgroundLearning", enableAdaptiveBackgroundLearning);
  fs.writeInt("enableGMG", enableGMG);
  fs.writeInt("enableDPAdaptiveMedianBGS", enableDPAdaptiveMedianBGS);
  fs.writeInt("enableDPGrimsonGMMBGS", enableDPGrimsonGMMBGS);
  fs.writeInt("enableDPZivkovicAGMMBGS", enableDPZivkovicAGMMBGS);
  fs.writeInt("enableDPMeanBGS", enableDPMeanBGS);
  fs.writeInt("enableDPWrenGABGS", enableDPWrenGABGS);
  fs.writeInt("enableSigmaDeltaBGS", enableSigmaDeltaBGS);
  fs.writeInt("enableSuBSENSEBGS", enableSuBSENSEBGS);
  fs.writeInt("enableLOBSTERBGS", enableLOBSTERBGS);
  fs.writeInt("enableAdaptiveSelectiveBackgroundLearning", enableAdaptiveSelectiveBackgroundLearning);
  fs.save("./config/FrameProcessor.xml");
}

void FrameProcessor::loadConfig() {  // :554-610 (defaults: PreProcessor and FrameDifferenceBGS on, everything else off)
  XmlConfig fs;
  fs.load("./config/FrameProcessor.xml");
  tictoc = fs.readString("tictoc", "");
  enablePreProcessor = fs.readInt("enablePreProcessor", true);
  enableFrameDifferenceBGS = fs.readInt("enableFrameDifferenceBGS", true);
  enableStaticFrameDifferenceBGS = fs.readInt("enableStaticFrameDifferenceBGS", false);
  enableWeightedMovingMeanBGS = fs.readInt("enableWeightedMovingMeanBGS", false);
  enableWeightedMovingVarianceBGS = fs.readInt("enableWeightedMovingVarianceBGS", false);
  enableMixtureOfGaussianV1BGS = fs.readInt("enableMixtureOfGaussianV1BGS", false);
  enableMixtureOfGaussianV2BGS = fs.readInt("enableMixtureOfGaussianV2BGS", false);
  enableAdaptiveBackgroundLearning = fs.readInt("enableAdaptiveBackgroundLearning", false);
  enableGMG = fs.readInt("enableGMG", false);
  enableDPAdaptiveMedianBGS = fs.readInt("enableDPAdaptiveMedianBGS", false);
  enableDPGrimsonGMMBGS = fs.readInt("enableDPGrimsonGMMBGS", false);
  enableDPZivkovicAGMMBGS = fs.readInt("enableDPZivkovicAGMMBGS", false);
  enableDPMeanBGS = fs.readInt("enableDPMeanBGS", false);
  enableDPWrenGABGS = fs.readInt("enableDPWrenGABGS", false);
  enableSigmaDeltaBGS = fs.readInt("enableSigmaDeltaBGS", false);
  enableSuBSENSEBGS = fs.readInt("enableSuBSENSEBGS", false);
  enableLOBSTERBGS = fs.readInt("enableLOBSTERBGS", false);
  enableAdaptiveSelectiveBackgroundLearning = fs.readInt("enableAdaptiveSelectiveBackgroundLearning", false);
}

}  // namespace bgs_hip
