// Minimal stand-ins for the handful of OpenCV value types that appear in the PUBLIC signatures of the
// stabilization path (cv::Size, Size2f, Point, Point2f, Rect, Rect2f, Scalar).  Used only when the build has no
// OpenCV (this image has none); with -DLVK_WITH_OPENCV the real <opencv2/core.hpp> types are used instead.
// Only what the reference's callers touch is provided (Modules/OBS-Plugin/Sources/Stabilisation/VSFilter.cpp:235-383).
#pragma once
#ifdef LVK_WITH_OPENCV
#include <opencv2/core.hpp>
#else
#include <cmath>
#include <cstdint>

namespace cv {

template <typename T> struct Size_
{
    T width{}, height{};
    Size_() = default;
    Size_(T w, T h) : width(w), height(h) {}
    template <typename U> Size_(const Size_<U>& o) : width(static_cast<T>(o.width)), height(static_cast<T>(o.height)) {}
    T area() const { return width * height; }
    bool operator==(const Size_& o) const { return width == o.width && height == o.height; }
    bool operator!=(const Size_& o) const { return !(*this == o); }
};
using Size = Size_<int>;
using Size2f = Size_<float>;

template <typename T> struct Point_
{
    T x{}, y{};
    Point_() = default;
    Point_(T x_, T y_) : x(x_), y(y_) {}
    Point_ operator+(const Point_& o) const { return {static_cast<T>(x + o.x), static_cast<T>(y + o.y)}; }
    Point_ operator-(const Point_& o) const { return {static_cast<T>(x - o.x), static_cast<T>(y - o.y)}; }
};
using Point = Point_<int>;
using Point2f = Point_<float>;
using Point2d = Point_<double>;

template <typename T> struct Rect_
{
    T x{}, y{}, width{}, height{};
    Rect_() = default;
    Rect_(T x_, T y_, T w, T h) : x(x_), y(y_), width(w), height(h) {}
    Point_<T> tl() const { return {x, y}; }
    Point_<T> br() const { return {static_cast<T>(x + width), static_cast<T>(y + height)}; }
    Size_<T> size() const { return {width, height}; }
};
using Rect = Rect_<int>;
using Rect2f = Rect_<float>;

struct Scalar
{
    double val[4]{0, 0, 0, 0};
    Scalar() = default;
    Scalar(double a, double b = 0, double c = 0, double d = 0) : val{a, b, c, d} {}
    double& operator[](int i) { return val[i]; }
    const double& operator[](int i) const { return val[i]; }
};

constexpr int CV_8UC1_ = 0, CV_8UC3_ = 16;

} // namespace cv

#ifndef CV_8UC3
#define CV_8UC1 0
#define CV_8UC3 16
#endif
#endif
