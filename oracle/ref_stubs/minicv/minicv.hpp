/* oracle/ref_stubs/minicv/minicv.hpp -- the sliver of the OpenCV API that the reference's host pipeline
 * (voldor/voldor.cpp, geometry.cpp, utils.cpp, py_export.cpp) touches, so that those files compile IN PLACE from
 * /root/reference with g++ and the reference's own py_voldor_wrapper runs on the CPU on top of the emulated kernels
 * (ref_wrap_kernels.cpp).  TEST INFRASTRUCTURE ONLY (oracle/_ref).  Nothing here is OpenCV or reference code; the
 * numerical conventions that matter are restated from OpenCV 3.4's documented behaviour and marked "cv:" below:
 *   cv: Mat *= s / Mat /= s are convertTo(alpha = s / 1./s): float data times (float)alpha (so /= multiplies by a reciprocal)
 *   cv: sum / mean / norm accumulate in double; GEMM accumulates in double except the unrolled 2..4-wide float path
 *   cv: Rodrigues in double, matrix input orthonormalised first, theta from acos of the clamped trace
 *   cv: 3x3 inverse through the double-precision adjugate
 * Not provided (abort if reached; never on the tested path): resize, solvePnP, eigen, image IO / display, the two-view
 * geometry of calib3d -- findEssentialMat / recoverPose return the pose injected with minicv_set_two_view_pose(), which
 * is how deviation D5 (8-point LMedS bootstrap instead of OpenCV's 5-point) enters a reference run. */
#pragma once
#include <algorithm>
#include <cassert>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "../../../voldor_amd/csrc/vk_strict_math.h"
#include "../../../voldor_amd/csrc/vk_ref_cv.h"
extern "C" int ref_math_mode;  /* ref_wrap_kernels.cpp; 1 = strict math (ref_stubs/emul/cuda_emul.h) */

#define CV_32F 5
#define CV_64F 6
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn) - 1) << 3))
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_32FC2 CV_MAKETYPE(CV_32F, 2)
#define CV_32FC3 CV_MAKETYPE(CV_32F, 3)
#define CV_64FC1 CV_MAKETYPE(CV_64F, 1)

namespace cv {
typedef unsigned char uchar;
enum { NORM_L2 = 4, NORM_MINMAX = 32, LMEDS = 4, RANSAC = 8, COLOR_HSV2BGR = 54 };

[[noreturn]] static inline void minicv_unsupported(const char* what) {
    fprintf(stderr, "minicv: %s is not provided (off the tested path)\n", what);
    abort();
}

template <class T> struct Size_ { T width, height; Size_() : width(0), height(0) {} Size_(T w, T h) : width(w), height(h) {} };
typedef Size_<int> Size;
template <class T> struct Rect_ { T x, y, width, height; Rect_() : x(0), y(0), width(0), height(0) {} Rect_(T a, T b, T c, T d) : x(a), y(b), width(c), height(d) {} };
typedef Rect_<int> Rect;
template <class T> struct Point_ { T x, y; Point_() : x(0), y(0) {} Point_(T a, T b) : x(a), y(b) {} };
typedef Point_<float> Point2f;
template <class T> struct Point3_ {
    T x, y, z;
    Point3_() : x(0), y(0), z(0) {}
    Point3_(T a, T b, T c) : x(a), y(b), z(c) {}
    T dot(const Point3_& o) const { return x * o.x + y * o.y + z * o.z; }
    Point3_& operator+=(const Point3_& o) { x += o.x; y += o.y; z += o.z; return *this; }
    Point3_& operator/=(double s) { x = (T)(x / s); y = (T)(y / s); z = (T)(z / s); return *this; }
    Point3_ operator*(T s) const { return Point3_(x * s, y * s, z * s); }
};
typedef Point3_<float> Point3f;

template <class T, int n> struct Vec {
    T val[n];
    Vec() { for (int i = 0; i < n; i++) val[i] = 0; }
    Vec(T a, T b) : Vec() { val[0] = a; val[1] = b; }
    Vec(T a, T b, T c) : Vec() { val[0] = a; val[1] = b; val[2] = c; }
    Vec(T a, T b, T c, T d) : Vec() { val[0] = a; val[1] = b; val[2] = c; val[3] = d; }
    Vec(T a, T b, T c, T d, T e, T f) : Vec() { val[0] = a; val[1] = b; val[2] = c; val[3] = d; val[4] = e; val[5] = f; }
    explicit Vec(const T* p) { for (int i = 0; i < n; i++) val[i] = p[i]; }
    template <class U> Vec(const Vec<U, n>& o) { for (int i = 0; i < n; i++) val[i] = (T)o.val[i]; }
    T& operator[](int i) { return val[i]; }
    const T& operator[](int i) const { return val[i]; }
    T dot(const Vec& o) const { T s = 0; for (int i = 0; i < n; i++) s += val[i] * o.val[i]; return s; }
    template <class P> T dot(const Point3_<P>& p) const { return val[0] * p.x + val[1] * p.y + val[2] * p.z; }
    Vec& operator/=(double s) { for (int i = 0; i < n; i++) val[i] = (T)(val[i] / s); return *this; }
    Vec operator-() const { Vec r; for (int i = 0; i < n; i++) r.val[i] = -val[i]; return r; }
};
typedef Vec<float, 2> Vec2f;
typedef Vec<float, 3> Vec3f;
typedef Vec<double, 3> Vec3d;
typedef Vec<float, 4> Vec4f;
typedef Vec<float, 6> Vec6f;
template <class T, int n> static inline double norm(const Vec<T, n>& v, int = NORM_L2) {
    double s = 0; for (int i = 0; i < n; i++) s += (double)v.val[i] * v.val[i]; return std::sqrt(s);
}
template <class T, int n> static inline std::ostream& operator<<(std::ostream& o, const Vec<T, n>& v) {
    o << "["; for (int i = 0; i < n; i++) o << (i ? ", " : "") << v.val[i]; return o << "]";
}

template <class T, int m, int n> struct Matx {
    T val[m * n];
    Matx() { for (int i = 0; i < m * n; i++) val[i] = 0; }
    explicit Matx(const T* p) { for (int i = 0; i < m * n; i++) val[i] = p[i]; }
    Matx(T a0, T a1, T a2, T a3, T a4, T a5, T a6, T a7, T a8) { T a[9] = { a0, a1, a2, a3, a4, a5, a6, a7, a8 }; for (int i = 0; i < m * n && i < 9; i++) val[i] = a[i]; }
    T& operator()(int i, int j) { return val[i * n + j]; }
};
typedef Matx<float, 3, 3> Matx33f;
typedef Matx<float, 3, 1> Matx31f;

struct Scalar { double val[4]; Scalar(double a = 0) { val[0] = a; val[1] = val[2] = val[3] = 0; } double operator[](int i) const { return val[i]; } };

struct Mat {
    int rows, cols;
    size_t step;  // bytes per row
    uchar* data;
    int flags;    // type
    std::shared_ptr<uchar> buf;

    Mat() : rows(0), cols(0), step(0), data(nullptr), flags(CV_32F) {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(Size s, int type) { create(s.height, s.width, type); }
    Mat(int r, int c, int type, void* d) : rows(r), cols(c), step(0), data((uchar*)d), flags(type) { step = (size_t)c * elemSize(); }
    Mat(Size s, int type, void* d) : Mat(s.height, s.width, type, d) {}
    void create(int r, int c, int type) {
        rows = r; cols = c; flags = type; step = (size_t)c * elemSize();
        buf.reset((uchar*)calloc((size_t)r * step + 64, 1), free);
        data = buf.get();
    }
    int type() const { return flags; }
    int depth() const { return flags & 7; }
    int channels() const { return (flags >> 3) + 1; }
    size_t elemSize1() const { return depth() == CV_64F ? 8 : 4; }
    size_t elemSize() const { return elemSize1() * channels(); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    Size size() const { return Size(cols, rows); }
    bool isContinuous() const { return rows <= 1 || step == (size_t)cols * elemSize(); }
    uchar* ptr(int r) const { return data + (size_t)r * step; }

    template <class T> T& at(int r, int c) { return ((T*)ptr(r))[c]; }
    template <class T> T& at(int i) {  // cv: Mat::at(int i0): flat index if continuous or a single row, else row index of a column vector
        if (isContinuous() || rows == 1) return ((T*)data)[i];
        return *(T*)ptr(i);
    }
    double get(int r, int c) const { return depth() == CV_64F ? ((const double*)ptr(r))[c] : (double)((const float*)ptr(r))[c]; }
    void set(int r, int c, double v) { if (depth() == CV_64F) ((double*)ptr(r))[c] = v; else ((float*)ptr(r))[c] = (float)v; }
    int width1() const { return cols * channels(); }  // scalars per row

    Mat clone() const {
        Mat m(rows, cols, flags);
        for (int r = 0; r < rows; r++) memcpy(m.ptr(r), ptr(r), (size_t)cols * elemSize());
        return m;
    }
    void copyTo(Mat dst) const {  // destination must already have the size (views into a larger image)
        assert(dst.rows == rows && dst.cols == cols && dst.flags == flags);
        for (int r = 0; r < rows; r++) memcpy(dst.ptr(r), ptr(r), (size_t)cols * elemSize());
    }
    void convertTo(Mat& dst, int type) const {
        Mat out(rows, cols, CV_MAKETYPE(type & 7, channels()));
        for (int r = 0; r < rows; r++) for (int c = 0; c < width1(); c++) out.set(r, c, get(r, c));
        dst = out;
    }
    Mat view(int r0, int r1, int c0, int c1) const {
        Mat m = *this;
        m.rows = r1 - r0; m.cols = c1 - c0; m.data = data + (size_t)r0 * step + (size_t)c0 * elemSize();
        return m;
    }
    Mat rowRange(int a, int b) const { return view(a, b, 0, cols); }
    Mat colRange(int a, int b) const { return view(0, rows, a, b); }
    Mat operator()(const Rect& r) const { return view(r.y, r.y + r.height, r.x, r.x + r.width); }
    Mat diag() const { Mat m = *this; m.rows = std::min(rows, cols); m.cols = 1; m.step = step + elemSize(); return m; }

    Mat& operator=(double s) { for (int r = 0; r < rows; r++) for (int c = 0; c < width1(); c++) set(r, c, s); return *this; }
    // cv: a *= s is a.convertTo(a, -1, s): float data are multiplied by (float)s
    Mat& scale(double alpha) {
        for (int r = 0; r < rows; r++) for (int c = 0; c < width1(); c++) {
            if (depth() == CV_64F) ((double*)ptr(r))[c] *= alpha; else ((float*)ptr(r))[c] *= (float)alpha;
        }
        return *this;
    }
    Mat& operator*=(double s) { return scale(s); }
    Mat& operator/=(double s) { return scale(1. / s); }  // cv: a.convertTo(a, -1, 1./s)
    Mat& operator+=(const Mat& o) {
        assert(rows == o.rows && cols == o.cols && flags == o.flags);
        for (int r = 0; r < rows; r++) for (int c = 0; c < width1(); c++) {
            if (depth() == CV_64F) ((double*)ptr(r))[c] += ((const double*)o.ptr(r))[c]; else ((float*)ptr(r))[c] += ((const float*)o.ptr(r))[c];
        }
        return *this;
    }
    Mat inv() const {  // cv: 3x3 through the adjugate with a double determinant
        assert(rows == 3 && cols == 3 && channels() == 1);
        double m[9]; for (int i = 0; i < 9; i++) m[i] = get(i / 3, i % 3);
        const double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
        Mat o = Mat::zeros(3, 3, flags);
        if (d == 0) return o;
        const double id = 1. / d;
        const double a[9] = { (m[4] * m[8] - m[5] * m[7]) * id, (m[2] * m[7] - m[1] * m[8]) * id, (m[1] * m[5] - m[2] * m[4]) * id,
                              (m[5] * m[6] - m[3] * m[8]) * id, (m[0] * m[8] - m[2] * m[6]) * id, (m[2] * m[3] - m[0] * m[5]) * id,
                              (m[3] * m[7] - m[4] * m[6]) * id, (m[1] * m[6] - m[0] * m[7]) * id, (m[0] * m[4] - m[1] * m[3]) * id };
        for (int i = 0; i < 9; i++) o.set(i / 3, i % 3, a[i]);
        return o;
    }
    static Mat zeros(int r, int c, int type) { return Mat(r, c, type); }
    static Mat zeros(Size s, int type) { return Mat(s, type); }
    static Mat ones(int r, int c, int type) { Mat m(r, c, type); for (int y = 0; y < r; y++) for (int x = 0; x < c; x++) m.set(y, x * m.channels(), 1.0); return m; }
    static Mat ones(Size s, int type) { return ones(s.height, s.width, type); }
    static Mat eye(int r, int c, int type) { Mat m(r, c, type); for (int i = 0; i < std::min(r, c); i++) m.set(i, i, 1.0); return m; }
};

static inline Mat operator*(const Mat& a, double s) { Mat m = a.clone(); m *= s; return m; }
static inline Mat operator*(double s, const Mat& a) { return a * s; }
static inline Mat operator/(const Mat& a, double s) { Mat m = a.clone(); m /= s; return m; }
static inline Mat operator/(double s, const Mat& a) {  // cv::divide(scale, src): 0 where src == 0 (3.4)
    Mat m = a.clone();
    for (int r = 0; r < m.rows; r++) for (int c = 0; c < m.width1(); c++) {
        if (m.depth() == CV_64F) { double& v = ((double*)m.ptr(r))[c]; v = v != 0 ? s / v : 0; }
        else { float& v = ((float*)m.ptr(r))[c]; v = v != 0 ? (float)s / v : 0; }
    }
    return m;
}
static inline Mat operator*(const Mat& a, const Mat& b) {
    // cv: gemm.  Inner dimension 2..4 takes the unrolled small-matrix path of core/matmul.cpp, which for CV_32F sums the
    // products in float; everything else accumulates in double.
    assert(a.cols == b.rows && a.channels() == 1 && b.channels() == 1 && a.depth() == b.depth());
    Mat m(a.rows, b.cols, a.flags);
    const bool small_f32 = a.depth() == CV_32F && a.cols >= 2 && a.cols <= 4;
    for (int i = 0; i < a.rows; i++) for (int j = 0; j < b.cols; j++) {
        if (small_f32) {
            float s = 0;
            for (int k = 0; k < a.cols; k++) s += ((const float*)a.ptr(i))[k] * ((const float*)b.ptr(k))[j];
            m.set(i, j, (double)s);
        } else {
            double s = 0;
            for (int k = 0; k < a.cols; k++) s += a.get(i, k) * b.get(k, j);
            m.set(i, j, s);
        }
    }
    return m;
}

template <class T> struct MatCommaInitializer_ {
    Mat m; int idx;
    MatCommaInitializer_(const Mat& mm) : m(mm), idx(0) {}
    template <class V> MatCommaInitializer_& operator,(V v) { ((T*)m.data)[idx++] = (T)v; return *this; }
    operator Mat() const { return m; }
};
template <class T> struct Mat_ : Mat { Mat_(int r, int c) : Mat(r, c, sizeof(T) == 8 ? CV_64F : CV_32F) {} };
template <class T, class V> static inline MatCommaInitializer_<T> operator<<(const Mat_<T>& m, V v) { MatCommaInitializer_<T> ci(m); return (ci, v); }

static inline Scalar sum(const Mat& m) { double s = 0; for (int r = 0; r < m.rows; r++) for (int c = 0; c < m.width1(); c++) s += m.get(r, c); return Scalar(s); }
static inline Scalar mean(const Mat& m) { return Scalar(sum(m)[0] / std::max(1, m.rows * m.width1())); }
static inline double norm(const Mat& m, int = NORM_L2) { double s = 0; for (int r = 0; r < m.rows; r++) for (int c = 0; c < m.width1(); c++) s += m.get(r, c) * m.get(r, c); return std::sqrt(s); }
static inline bool checkRange(const Mat& m) { for (int r = 0; r < m.rows; r++) for (int c = 0; c < m.width1(); c++) if (!std::isfinite(m.get(r, c))) return false; return true; }

// ---- cv::Rodrigues (calib3d cvRodrigues2), double arithmetic: restated in voldor_amd/csrc/vk_ref_cv.h (one copy for this stand-in, the
// oracle and the strict kernels of the product; strict mode = ref_math_mode 1: sin / cos / acos from vk_strict_math.h)
static inline void minicv_rvec_to_R(const double r_in[3], double R[9]) { vrcv_rvec_to_R(r_in, R, ref_math_mode == 1); }
static inline void minicv_R_to_rvec(const double R_in[9], double r[3]) { vrcv_R_to_rvec(R_in, r, ref_math_mode == 1); }
static inline void Rodrigues(const Vec3f& rvec, Mat& R) {  // vector -> 3x3 CV_32F
    const double r[3] = { rvec.val[0], rvec.val[1], rvec.val[2] }; double Rd[9];
    minicv_rvec_to_R(r, Rd);
    R = Mat(3, 3, CV_32F);
    for (int i = 0; i < 9; i++) R.set(i / 3, i % 3, Rd[i]);
}
static inline void Rodrigues(const Mat& R, Vec3f& rvec) {  // 3x3 -> vector
    assert(R.rows == 3 && R.cols == 3);
    double Rd[9], r[3]; for (int i = 0; i < 9; i++) Rd[i] = R.get(i / 3, i % 3);
    minicv_R_to_rvec(Rd, r);
    for (int i = 0; i < 3; i++) rvec.val[i] = (float)r[i];
}
static inline void Rodrigues(const Matx33f& R, Vec3f& rvec) {
    double Rd[9], r[3]; for (int i = 0; i < 9; i++) Rd[i] = R.val[i];
    minicv_R_to_rvec(Rd, r);
    for (int i = 0; i < 3; i++) rvec.val[i] = (float)r[i];
}

// ---- calib3d two-view geometry: stand-in (deviation D5), see the header comment
}  // namespace cv
extern double minicv_two_view_R[9], minicv_two_view_t[3];  // defined in oracle/ref_wrap_host.cpp, set by minicv_set_two_view_pose
namespace cv {
static inline Mat findEssentialMat(const Mat&, const Mat&, const Mat&, int, double, double, Mat&) { return Mat::zeros(3, 3, CV_64F); }
static inline int recoverPose(const Mat&, const Mat&, const Mat&, const Mat&, Mat& R, Mat& t) {
    R = Mat(3, 3, CV_64F); t = Mat(3, 1, CV_64F);
    for (int i = 0; i < 9; i++) R.set(i / 3, i % 3, minicv_two_view_R[i]);
    for (int i = 0; i < 3; i++) t.set(i, 0, minicv_two_view_t[i]);
    return 0;
}

// ---- off the tested path
struct _InputArray { template <class T> _InputArray(const T*, int) {} _InputArray(const Mat&) {} };
template <class A, class B> static inline bool solvePnP(const _InputArray&, const _InputArray&, const Mat&, const Mat&, A&, B&, bool, int) { minicv_unsupported("solvePnP"); }
static inline void resize(const Mat&, Mat&, Size, double = 0, double = 0) { minicv_unsupported("resize"); }
static inline bool eigen(const Matx33f&, Matx31f&, Matx33f&) { minicv_unsupported("eigen"); }
static inline void split(const Mat&, Mat*) { minicv_unsupported("split"); }
static inline void merge(const std::vector<Mat>&, Mat&) { minicv_unsupported("merge"); }
static inline void cartToPolar(const Mat&, const Mat&, Mat&, Mat&, bool) { minicv_unsupported("cartToPolar"); }
static inline void normalize(const Mat&, Mat&, double, double, int) { minicv_unsupported("normalize"); }
static inline void cvtColor(const Mat&, Mat&, int) { minicv_unsupported("cvtColor"); }
static inline bool imwrite(const std::string&, const Mat&) { return false; }
static inline void imshow(const std::string&, const Mat&) {}
static inline int waitKey(int = 0) { return 0; }
static inline void destroyAllWindows() {}
}  // namespace cv
