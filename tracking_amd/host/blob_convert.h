// blob_convert.h — bgs_box / bgs_moments (include/bgs_hip.h) -> CvBlob {x, y, w, h, ID}: the blob list hand-off of SURVEY.md N2.
//
// CvBlob (opencv2/legacy/blobtrack.hpp) is {float x, y; float w, h; int ID;}: CENTRE and size of a blob.  OpenCV-legacy's blob
// detectors, which consume the foreground mask right after USTC_BGS::Process (ustc_src/trackingMain.cpp:166 through
// CvBlobTrackerAuto1), build one from a foreground region in two ways (recalled from modules/legacy/src/enteringblobdetection.cpp;
// OpenCV is not in the tree, so this is unpinned and both are offered):
//   * from the bounding rectangle r:          cvBlob(r.x + 0.5 r.width, r.y + 0.5 r.height, r.width, r.height)
//   * from the region's moments inside r:     X = m10/m00, Y = m01/m00, XX = m20/m00 - X*X, YY = m02/m00 - Y*Y,
//                                             cvBlob(r.x + X, r.y + Y, 4 sqrt(XX), 4 sqrt(YY))
// Templates on the blob type so that the same code serves the real CvBlob (tracking_amd/host/HipFGDetector.h, needs OpenCV) and
// this repository's mirror (tracking_amd/host/blob.h).  C++03.
#pragma once
#include <cmath>

#include "bgs_hip.h"

namespace bgs_hip_convert {

template <class BlobT>
inline void blob_from_box(const bgs_box& b, BlobT& out) {
  out.x = (float)b.x + 0.5f * (float)b.w, out.y = (float)b.y + 0.5f * (float)b.h;
  out.w = (float)b.w, out.h = (float)b.h;
}

// sums are over absolute image coordinates; the recalled code measures X, Y from the rectangle's corner and adds it back:
// the centre is the same, and a variance does not depend on the origin
template <class BlobT>
inline void blob_from_moments(const bgs_box& b, const bgs_moments& m, BlobT& out) {
  const double M00 = (double)b.area;
  const double X = (double)m.sx / M00 - (double)b.x, Y = (double)m.sy / M00 - (double)b.y;
  const double mx = (double)m.sx / M00, my = (double)m.sy / M00;
  double XX = (double)m.sxx / M00 - mx * mx, YY = (double)m.syy / M00 - my * my;
  if (XX < 0) XX = 0;
  if (YY < 0) YY = 0;
  out.x = (float)b.x + (float)X, out.y = (float)b.y + (float)Y;
  out.w = (float)(4 * std::sqrt(XX)), out.h = (float)(4 * std::sqrt(YY));
}

}  // namespace bgs_hip_convert
