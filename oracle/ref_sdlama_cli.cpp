// ref_sdlama_cli.cpp — TEST INFRASTRUCTURE.  Drives the REFERENCE'S OWN package_bgs/bl/sdLaMa091.cpp (included unmodified from
// /root/reference via the include path of oracle/Makefile) exactly the way SigmaDeltaBGS::process does
// (package_bgs/bl/SigmaDeltaBGS.cpp:20-55): ctor applyParams, per frame applyParams (loadConfig), first frame =
// sdLaMa091AllocInit_8u_C3R and no output, then sdLaMa091Update_8u_C3R and channel 0 of the 3-channel map.
//   ref_sdlama_cli frames.raw rows cols n_frames ampFactor minVar maxVar out.raw      (out = (n_frames-1) x rows x cols)
// A process of its own on purpose: sdLaMa091 leaves two thirds of its Vt buffer uninitialised (sdLaMa091.cpp:190-201 via
// :211-212); in a fresh process those mallocs are untouched zero pages, which is also what real frame sizes (mmap) get.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "sdLaMa091.cpp"

int main(int argc, char** argv) {
  if (argc < 9) return 2;
  const int rows = atoi(argv[2]), cols = atoi(argv[3]), n = atoi(argv[4]);
  const unsigned amp = (unsigned)atoi(argv[5]), vmin = (unsigned)atoi(argv[6]), vmax = (unsigned)atoi(argv[7]);
  const size_t fb = (size_t)rows * cols * 3;
  std::vector<unsigned char> frames(fb * n), seg(fb), out((size_t)rows * cols * (n - 1));
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(frames.data(), 1, frames.size(), f) != frames.size()) return 3;
  fclose(f);
  sdLaMa091_t* a = sdLaMa091New();
  bool firstTime = true;
  for (int t = 0; t < n; ++t) {
    sdLaMa091SetAmplificationFactor(a, amp);  // applyParams()
    sdLaMa091SetMinimalVariance(a, vmin);
    sdLaMa091SetMaximalVariance(a, vmax);
    const unsigned char* img = frames.data() + fb * t;
    if (firstTime) {
      sdLaMa091AllocInit_8u_C3R(a, img, cols, rows, cols * 3);
      firstTime = false;
      continue;
    }
    sdLaMa091Update_8u_C3R(a, img, seg.data());
    unsigned char* o = out.data() + (size_t)rows * cols * (t - 1);
    for (size_t i = 0; i < (size_t)rows * cols; ++i) o[i] = seg[3 * i];
  }
  sdLaMa091Free(a);
  f = fopen(argv[8], "wb");
  if (!f || fwrite(out.data(), 1, out.size(), f) != out.size()) return 4;
  fclose(f);
  return 0;
}
