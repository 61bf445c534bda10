// opencv2/opencv.hpp -- TEST INFRASTRUCTURE ONLY.  A stand-in for the handful of OpenCV types the reference's STag sources
// use outside the ED library (OpenCV itself is not installed in this image), so that oracle/Makefile can compile
// stag_detect/src/stag/{Quad,QuadDetector,EDInterface,utility}.cpp where they lie and the parity tests can check the
// device code against the reference's OWN logic.  What is here is data plumbing with one obvious meaning (points, a dense
// row-major matrix, element access, the 3 x 3 / 3 x 1 double products of Quad.cpp, evaluated as a_i0 b_0j + a_i1 b_1j +
// a_i2 b_2j left to right).  Anything with numerical content of its own (filters, thresholds, solvers) is NOT provided
// here: callers of those are restated in oracle/stag_ref.cpp and say so.
#ifndef FID_ORACLE_CVSHIM_OPENCV_HPP
#define FID_ORACLE_CVSHIM_OPENCV_HPP
// <stdlib.h> / <math.h> under their C names, as the real OpenCV headers pull them in (core/cv_cpu_dispatch.h ->
// <emmintrin.h> -> <mm_malloc.h> -> <stdlib.h>; flann/lsh_table.h -> <math.h>): in C++ these are libstdc++'s wrappers, which
// put std::abs(double) into the global namespace.  The reference calls unqualified abs() on doubles (Quad.cpp:135,
// QuadDetector.cpp:118,162); without this it would bind to ::abs(int) here and truncate, unlike the real build.
#include <stdlib.h>
#include <math.h>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8UC1 0
#define CV_64FC1 6

namespace cv {

template <typename T>
struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T x_, T y_) : x(x_), y(y_) {}
};
typedef Point_<int> Point2i;
typedef Point_<int> Point;
typedef Point_<float> Point2f;
typedef Point_<double> Point2d;

template <typename T>
struct Point3_ {
    T x, y, z;
    Point3_() : x(0), y(0), z(0) {}
    Point3_(T x_, T y_, T z_) : x(x_), y(y_), z(z_) {}
};
typedef Point3_<double> Point3d;

struct Size {
    int width, height;
    Size() : width(0), height(0) {}
    Size(int w, int h) : width(w), height(h) {}
};

class Mat {
   public:
    int rows, cols, type_;
    unsigned char *data;
    Mat() : rows(0), cols(0), type_(CV_8UC1), data(nullptr) {}
    Mat(int r, int c, int type) : rows(r), cols(c), type_(type)
    {
        own_.reset(new std::vector<unsigned char>((size_t)r * c * esz(), 0));
        data = own_->data();
    }
    Mat(int r, int c, int type, void *ext) : rows(r), cols(c), type_(type), data((unsigned char *)ext) {}
    static Mat eye(int r, int c, int type)
    {
        Mat m(r, c, type);
        for (int i = 0; i < r && i < c; i++) m.at<double>(i, i) = 1.0;
        return m;
    }
    size_t esz() const { return type_ == CV_64FC1 ? 8 : 1; }
    Size size() const { return Size(cols, rows); }
    bool empty() const { return data == nullptr; }
    template <typename T>
    T &at(int i, int j) { return ((T *)data)[(size_t)i * cols + j]; }
    template <typename T>
    const T &at(int i, int j) const { return ((const T *)data)[(size_t)i * cols + j]; }
    template <typename T>
    T &at(int i) { return ((T *)data)[i]; }
    template <typename T>
    const T &at(int i) const { return ((const T *)data)[i]; }
    template <typename T>
    T *ptr(int r) { return (T *)data + (size_t)r * cols; }
    template <typename T>
    const T *ptr(int r) const { return (const T *)data + (size_t)r * cols; }
    Mat inv() const;  // DECLARED only (numerical content: restated in oracle/stag_ref.cpp)
    Mat clone() const
    {
        Mat m(rows, cols, type_);
        if (data) memcpy(m.data, data, (size_t)rows * cols * esz());
        return m;
    }

   private:
    std::shared_ptr<std::vector<unsigned char>> own_;
};

// cv::Mat_<double>(3, 1) << a, b, c  (PoseRefiner.cpp:93-96, 200)
template <typename T>
class Mat_ : public Mat {
   public:
    Mat_(int r, int c) : Mat(r, c, CV_64FC1), pos_(0) {}
    Mat_ &operator<<(T v) { at<T>(pos_++) = v; return *this; }
    Mat_ &operator,(T v) { at<T>(pos_++) = v; return *this; }

   private:
    int pos_;
};

inline void transpose(const Mat &a, Mat &d)
{
    Mat t(a.cols, a.rows, CV_64FC1);
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < a.cols; j++) t.at<double>(j, i) = a.at<double>(i, j);
    d = t;
}

// like OpenCV 4's cv::Ptr: a shared_ptr with an UNCONSTRAINED converting constructor (its body is instantiated at the end
// of the translation unit, which is what lets PoseRefiner.cpp convert Ptr<Refine> while Refine is still incomplete)
template <typename T>
struct Ptr : public std::shared_ptr<T> {
    Ptr() {}
    Ptr(T *p) : std::shared_ptr<T>(p) {}
    template <typename Y>
    Ptr(const Ptr<Y> &o) : std::shared_ptr<T>(static_cast<const std::shared_ptr<Y> &>(o)) {}
};
template <typename T>
Ptr<T> makePtr() { return Ptr<T>(new T()); }

class MinProblemSolver {
   public:
    class Function {
       public:
        virtual ~Function() {}
        virtual int getDims() const = 0;
        virtual double calc(const double *x) const = 0;
    };
};

// DECLARED only: cv::DownhillSolver (Nelder-Mead) has numerical content; its restatement is in oracle/stag_ref.cpp
class DownhillSolver {
   public:
    static Ptr<DownhillSolver> create();
    void setFunction(const Ptr<MinProblemSolver::Function> &f) { f_ = f; }
    void setInitStep(const Mat &step) { step_ = step.clone(); }
    double minimize(Mat &x);

   private:
    Ptr<MinProblemSolver::Function> f_;
    Mat step_;
};

struct Scalar {
    double val[4];
    Scalar(double a = 0, double b = 0, double c = 0, double d = 0) { val[0] = a; val[1] = b; val[2] = c; val[3] = d; }
};

// DECLARED only: the one cv:: call of Stag::readCode (Stag.cpp:119).  Its body has numerical content (Otsu's threshold) and
// is therefore a restatement, kept with the other restatements in oracle/stag_ref.cpp.
enum { THRESH_BINARY_INV = 1, THRESH_OTSU = 8 };
double threshold(std::vector<unsigned char> &src, std::vector<unsigned char> &dst, double thresh, double maxval, int type);

// double matrices only (the 3 x 3 and 3 x 1 products of Quad.cpp / Stag.cpp)
inline Mat operator*(const Mat &a, const Mat &b)
{
    Mat d(a.rows, b.cols, CV_64FC1);
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < b.cols; j++) {
            double s = a.at<double>(i, 0) * b.at<double>(0, j);
            for (int k = 1; k < a.cols; k++) s += a.at<double>(i, k) * b.at<double>(k, j);
            d.at<double>(i, j) = s;
        }
    return d;
}

}  // namespace cv
#endif
