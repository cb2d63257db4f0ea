// declaration-only mock, see ../README.md
#pragma once
#include <cstddef>
#include <string>
typedef unsigned char uchar;
struct IplImage {
  int nChannels, depth, width, height, widthStep;
  char* imageData;
};
struct CvFileStorage;
struct CvFileNode;
#define CV_STORAGE_READ 0
#define CV_STORAGE_WRITE 1
CvFileStorage* cvOpenFileStorage(const char* filename, void* memstorage, int flags, const char* encoding = 0);
void cvReleaseFileStorage(CvFileStorage** fs);
int cvReadIntByName(const CvFileStorage* fs, const CvFileNode* map, const char* name, int default_value = 0);
double cvReadRealByName(const CvFileStorage* fs, const CvFileNode* map, const char* name, double default_value = 0.);
void cvWriteInt(CvFileStorage* fs, const char* name, int value);
void cvWriteReal(CvFileStorage* fs, const char* name, double value);
#define CV_8U 0
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_StsError -2
#define CV_StsBadArg -5
namespace cv {
struct Size {
  int width, height;
};
class Exception {
 public:
  virtual ~Exception() throw();
  virtual const char* what() const throw();
};
void error(const Exception& exc);
class Mat {
 public:
  Mat();
  Mat(const IplImage* img, bool copyData = false);
  ~Mat();
  void create(Size size, int type);
  void create(int rows, int cols, int type);
  void copyTo(Mat& m) const;
  void release();
  bool empty() const;
  int channels() const;
  Size size() const;
  operator IplImage() const;
  int rows, cols;
  uchar* data;
  struct MStep {
    operator size_t() const;
  } step;
};
}  // namespace cv
void cv_mock_error(int code, const std::string& msg);
#define CV_Error(code, msg) cv_mock_error(code, msg)
#define CV_Assert(expr) \
  if (!!(expr))         \
    ;                   \
  else                  \
    cv_mock_error(CV_StsError, #expr)
