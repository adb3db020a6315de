// ros_stub_all.hpp -- minimal stand-ins for the ROS 2 / PCL / Eigen / tf2 / pluginlib declarations that the reference's
// local-planner and perception sources (and this repository's adapter sources and patches) use, so that those files can
// be SYNTAX-checked (g++ -fsyntax-only, tests/test_adapters_cpu.py) in an image that has none of the real packages
// ("compiled only syntactically against stub headers").  Declarations only: nothing here is ever linked
// or run, nothing here is a build of the reference (the reference is unbuildable in this image).
// Every <pkg/header> under tests/stubs/ forwards to this file.
#pragma once
#include <array>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <deque>
#include <functional>
#include <iostream>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

// ------------------------------------------------------------------ Eigen
namespace Eigen {
struct Vector3d {
  double v[3] = {0, 0, 0};
  Vector3d() {}
  Vector3d(double, double, double) {}
  double& x() { return v[0]; } double& y() { return v[1]; } double& z() { return v[2]; }
  double x() const { return v[0]; } double y() const { return v[1]; } double z() const { return v[2]; }
  double& operator[](int i) { return v[i]; } double operator[](int i) const { return v[i]; }
  double& operator()(int i) { return v[i]; } double operator()(int i) const { return v[i]; }
  double norm() const { return 0; } Vector3d normalized() const { return *this; } void normalize() {}
  double dot(const Vector3d&) const { return 0; } Vector3d cross(const Vector3d&) const { return *this; }
  Vector3d operator+(const Vector3d&) const { return *this; } Vector3d operator-(const Vector3d&) const { return *this; }
  Vector3d operator*(double) const { return *this; } Vector3d operator/(double) const { return *this; }
  static Vector3d UnitX() { return Vector3d(); } static Vector3d UnitY() { return Vector3d(); } static Vector3d UnitZ() { return Vector3d(); }
};
struct Vector3f {
  float v[3] = {0, 0, 0};
  Vector3f() {}
  Vector3f(float, float, float) {}
  float& x() { return v[0]; } float& y() { return v[1]; } float& z() { return v[2]; }
  float x() const { return v[0]; } float y() const { return v[1]; } float z() const { return v[2]; }
  float& operator[](int i) { return v[i]; } float operator[](int i) const { return v[i]; }
  float& operator()(int i) { return v[i]; } float operator()(int i) const { return v[i]; }
  float norm() const { return 0; } Vector3f normalized() const { return *this; } void normalize() {}
  float dot(const Vector3f&) const { return 0; } Vector3f cross(const Vector3f&) const { return *this; }
  Vector3f operator+(const Vector3f&) const { return *this; } Vector3f operator-(const Vector3f&) const { return *this; }
  Vector3f operator*(float) const { return *this; } Vector3f operator/(float) const { return *this; }
  static Vector3f Zero() { return Vector3f(); }
};
struct Vector4f { float v[4] = {0, 0, 0, 0}; float& operator[](int i) { return v[i]; } float& operator()(int i) { return v[i]; } float operator()(int i) const { return v[i]; } };
struct Matrix3d { double& operator()(int, int) { static double d; return d; } double operator()(int, int) const { return 0; } Matrix3d inverse() const { return *this; } Matrix3d transpose() const { return *this; } };
struct Matrix4f { float& operator()(int, int) { static float d; return d; } static Matrix4f Identity() { return Matrix4f(); } };
struct Matrix4d { double& operator()(int, int) { static double d; return d; } static Matrix4d Identity() { return Matrix4d(); } };
struct Quaterniond { Quaterniond() {} Quaterniond(double, double, double, double) {} double x() const { return 0; } double y() const { return 0; } double z() const { return 0; } double w() const { return 1; } Matrix3d toRotationMatrix() const { return Matrix3d(); } };
struct AngleAxisd { AngleAxisd() {} AngleAxisd(double, const Vector3d&) {} Matrix3d toRotationMatrix() const { return Matrix3d(); } };
struct Translation3d { Translation3d(double, double, double) {} };
struct Affine3d {
  Vector3d t;
  Affine3d() {}
  Affine3d(const AngleAxisd&) {}
  Affine3d(const Translation3d&) {}
  Vector3d& translation() { return t; } const Vector3d& translation() const { return t; }
  Matrix3d rotation() const { return Matrix3d(); } Matrix3d linear() const { return Matrix3d(); }
  Affine3d inverse() const { return *this; }
  Affine3d operator*(const Affine3d&) const { return *this; }
  Vector3d operator*(const Vector3d& p) const { return p; }
  Matrix4d matrix() const { return Matrix4d(); }
  static Affine3d Identity() { return Affine3d(); }
  void pretranslate(const Vector3d&) {} void prerotate(const AngleAxisd&) {} void rotate(const AngleAxisd&) {} void translate(const Vector3d&) {}
};
typedef Affine3d Isometry3d;
inline std::ostream& operator<<(std::ostream& o, const Matrix3d&) { return o; }
}  // namespace Eigen

// ------------------------------------------------------------------ messages
namespace builtin_interfaces { namespace msg { struct Time { int32_t sec = 0; uint32_t nanosec = 0; }; } }
namespace rclcpp {
struct Duration {
  Duration() {} Duration(int32_t, uint32_t) {} Duration(std::chrono::nanoseconds) {}
  double seconds() const { return 0; } int64_t nanoseconds() const { return 0; }
  static Duration from_seconds(double) { return Duration(); }
  bool operator>(const Duration&) const { return false; } bool operator<(const Duration&) const { return false; }
};
struct Time {
  Time() {} Time(int32_t, uint32_t) {} Time(const builtin_interfaces::msg::Time&) {} Time(int64_t) {}
  double seconds() const { return 0; } int64_t nanoseconds() const { return 0; }
  Duration operator-(const Time&) const { return Duration(); } Time operator+(const Duration&) const { return *this; } Time operator-(const Duration&) const { return *this; }
  bool operator>(const Time&) const { return false; } bool operator<(const Time&) const { return false; }
  operator builtin_interfaces::msg::Time() const { return builtin_interfaces::msg::Time(); }
};
}  // namespace rclcpp
namespace std_msgs { namespace msg {
struct Header { std::string frame_id; builtin_interfaces::msg::Time stamp; };
struct ColorRGBA { float r = 0, g = 0, b = 0, a = 0; };
} }
#define DDDMR_STUB_MSG_PTRS(T) typedef std::shared_ptr<T> SharedPtr; typedef std::shared_ptr<const T> ConstSharedPtr; typedef std::unique_ptr<T> UniquePtr;
namespace geometry_msgs { namespace msg {
struct Point { double x = 0, y = 0, z = 0; DDDMR_STUB_MSG_PTRS(Point) };
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
struct PoseStamped { std_msgs::msg::Header header; Pose pose; DDDMR_STUB_MSG_PTRS(PoseStamped) };
struct PointStamped { std_msgs::msg::Header header; Point point; DDDMR_STUB_MSG_PTRS(PointStamped) };
struct PoseArray { typedef std::vector<Pose> _poses_type; std_msgs::msg::Header header; _poses_type poses; DDDMR_STUB_MSG_PTRS(PoseArray) };
struct Transform { Vector3 translation; Quaternion rotation; };
struct TransformStamped { std_msgs::msg::Header header; std::string child_frame_id; Transform transform; DDDMR_STUB_MSG_PTRS(TransformStamped) };
struct Twist { Vector3 linear, angular; DDDMR_STUB_MSG_PTRS(Twist) };
struct TwistWithCovariance { Twist twist; };
struct PoseWithCovariance { Pose pose; };
} }
namespace nav_msgs { namespace msg {
struct Path { std_msgs::msg::Header header; std::vector<geometry_msgs::msg::PoseStamped> poses; DDDMR_STUB_MSG_PTRS(Path) };
struct Odometry { std_msgs::msg::Header header; std::string child_frame_id; geometry_msgs::msg::PoseWithCovariance pose; geometry_msgs::msg::TwistWithCovariance twist; DDDMR_STUB_MSG_PTRS(Odometry) };
} }
namespace sensor_msgs { namespace msg {
struct PointCloud2 { std_msgs::msg::Header header; uint32_t height = 0, width = 0; std::vector<uint8_t> data; DDDMR_STUB_MSG_PTRS(PointCloud2) };
} }
namespace visualization_msgs { namespace msg {
struct Marker {
  enum { ADD = 0, DELETE = 2, DELETEALL = 3, ARROW = 0, CUBE = 1, SPHERE = 2, CYLINDER = 3, LINE_STRIP = 4, LINE_LIST = 5, CUBE_LIST = 6, SPHERE_LIST = 7, POINTS = 8, TEXT_VIEW_FACING = 9 };
  std_msgs::msg::Header header; std::string ns, text; int id = 0, type = 0, action = 0; geometry_msgs::msg::Pose pose; geometry_msgs::msg::Vector3 scale;
  std_msgs::msg::ColorRGBA color; rclcpp::Duration lifetime; std::vector<geometry_msgs::msg::Point> points; std::vector<std_msgs::msg::ColorRGBA> colors;
  DDDMR_STUB_MSG_PTRS(Marker)
};
struct MarkerArray { std::vector<Marker> markers; DDDMR_STUB_MSG_PTRS(MarkerArray) };
} }
namespace std_srvs { namespace srv { struct Empty { struct Request {}; struct Response {}; }; } }

// ------------------------------------------------------------------ rclcpp
namespace rclcpp {
struct Logger { Logger get_child(const std::string&) const { return *this; } const char* get_name() const { return ""; } };
inline Logger get_logger(const std::string&) { return Logger(); }
struct Clock { typedef std::shared_ptr<Clock> SharedPtr; Time now() const { return Time(); } };
struct ParameterValue { template <class T> ParameterValue(const T&) {} ParameterValue() {} };
enum ParameterType { PARAMETER_NOT_SET, PARAMETER_BOOL, PARAMETER_INTEGER, PARAMETER_DOUBLE, PARAMETER_STRING, PARAMETER_BYTE_ARRAY, PARAMETER_BOOL_ARRAY, PARAMETER_INTEGER_ARRAY, PARAMETER_DOUBLE_ARRAY, PARAMETER_STRING_ARRAY };
struct Parameter { template <class T> T get_value() const { return T(); } std::vector<double> as_double_array() const { return {}; } std::vector<std::string> as_string_array() const { return {}; } std::vector<int64_t> as_integer_array() const { return {}; } double as_double() const { return 0; } int64_t as_int() const { return 0; } bool as_bool() const { return false; } std::string as_string() const { return ""; } };
enum class CallbackGroupType { MutuallyExclusive, Reentrant };
struct CallbackGroup { typedef std::shared_ptr<CallbackGroup> SharedPtr; };
struct SubscriptionOptions { CallbackGroup::SharedPtr callback_group; };
struct QoS { QoS(size_t) {} QoS& transient_local() { return *this; } QoS& reliable() { return *this; } QoS& best_effort() { return *this; } QoS& durability_volatile() { return *this; } QoS& keep_last(size_t) { return *this; } };
struct SensorDataQoS : QoS { SensorDataQoS() : QoS(5) {} };
struct KeepLast { KeepLast(size_t) {} operator QoS() const { return QoS(1); } };
template <class M> struct Publisher { typedef std::shared_ptr<Publisher<M>> SharedPtr; void publish(const M&) {} void publish(std::unique_ptr<M>) {} size_t get_subscription_count() const { return 0; } const char* get_topic_name() const { return ""; } };
template <class M> struct Subscription { typedef std::shared_ptr<Subscription<M>> SharedPtr; };
struct TimerBase { typedef std::shared_ptr<TimerBase> SharedPtr; void cancel() {} void reset() {} };
template <class S> struct Service { typedef std::shared_ptr<Service<S>> SharedPtr; };
namespace node_interfaces { struct NodeBaseInterface { typedef std::shared_ptr<NodeBaseInterface> SharedPtr; }; struct NodeTimersInterface { typedef std::shared_ptr<NodeTimersInterface> SharedPtr; };
  struct NodeLoggingInterface { typedef std::shared_ptr<NodeLoggingInterface> SharedPtr; Logger get_logger() const { return Logger(); } const char* get_logger_name() const { return ""; } };
  struct NodeParametersInterface { typedef std::shared_ptr<NodeParametersInterface> SharedPtr; }; struct NodeClockInterface { typedef std::shared_ptr<NodeClockInterface> SharedPtr; Clock::SharedPtr get_clock() { return nullptr; } }; }
struct NodeOptions {};
struct Node : std::enable_shared_from_this<Node> {
  typedef std::shared_ptr<Node> SharedPtr; typedef std::weak_ptr<Node> WeakPtr;
  Node(const std::string&) {} Node(const std::string&, const NodeOptions&) {} virtual ~Node() {}
  Logger get_logger() const { return Logger(); }
  Clock::SharedPtr get_clock() { return std::make_shared<Clock>(); }
  Time now() const { return Time(); }
  const char* get_name() const { return ""; } const char* get_namespace() const { return ""; }
  template <class T> T declare_parameter(const std::string&, const T& d) { return d; }
  ParameterValue declare_parameter(const std::string&, const ParameterValue& d) { return d; }
  ParameterValue declare_parameter(const std::string&, ParameterType) { return ParameterValue(); }
  bool has_parameter(const std::string&) const { return false; }
  template <class T> bool get_parameter(const std::string&, T&) const { return false; }
  Parameter get_parameter(const std::string&) const { return Parameter(); }
  template <class M, class... A> typename Publisher<M>::SharedPtr create_publisher(const std::string&, A&&...) { return nullptr; }
  template <class M, class CB, class... A> typename Subscription<M>::SharedPtr create_subscription(const std::string&, const QoS&, CB&&, A&&...) { return nullptr; }
  template <class D, class CB, class... A> TimerBase::SharedPtr create_wall_timer(D, CB&&, A&&...) { return nullptr; }
  template <class S, class CB, class... A> typename Service<S>::SharedPtr create_service(const std::string&, CB&&, A&&...) { return nullptr; }
  CallbackGroup::SharedPtr create_callback_group(CallbackGroupType) { return nullptr; }
  node_interfaces::NodeBaseInterface::SharedPtr get_node_base_interface() { return nullptr; }
  node_interfaces::NodeTimersInterface::SharedPtr get_node_timers_interface() { return nullptr; }
  node_interfaces::NodeLoggingInterface::SharedPtr get_node_logging_interface() { return nullptr; }
  node_interfaces::NodeParametersInterface::SharedPtr get_node_parameters_interface() { return nullptr; }
  node_interfaces::NodeClockInterface::SharedPtr get_node_clock_interface() { return nullptr; }
};
inline bool ok() { return true; }
struct Rate { Rate(double) {} bool sleep() { return true; } };
struct WallRate { WallRate(double) {} bool sleep() { return true; } };
}  // namespace rclcpp
#define DDDMR_STUB_LOG(...) do { (void)sizeof(::dddmr_stub::eat(__VA_ARGS__)); } while (0)
namespace dddmr_stub { template <class... A> int eat(A&&...) { return 0; } }
#define RCLCPP_DEBUG(...) DDDMR_STUB_LOG(__VA_ARGS__)
#define RCLCPP_INFO(...) DDDMR_STUB_LOG(__VA_ARGS__)
#define RCLCPP_WARN(...) DDDMR_STUB_LOG(__VA_ARGS__)
#define RCLCPP_ERROR(...) DDDMR_STUB_LOG(__VA_ARGS__)
#define RCLCPP_FATAL(...) DDDMR_STUB_LOG(__VA_ARGS__)
#define RCLCPP_DEBUG_THROTTLE(...) DDDMR_STUB_LOG(__VA_ARGS__)
#define RCLCPP_INFO_THROTTLE(...) DDDMR_STUB_LOG(__VA_ARGS__)
#define RCLCPP_WARN_THROTTLE(...) DDDMR_STUB_LOG(__VA_ARGS__)
#define RCLCPP_ERROR_THROTTLE(...) DDDMR_STUB_LOG(__VA_ARGS__)
#define RCLCPP_INFO_STREAM(l, s) do { std::ostringstream dddmr_stub_os; dddmr_stub_os << s; (void)l; } while (0)
#define RCLCPP_WARN_STREAM(l, s) RCLCPP_INFO_STREAM(l, s)
#define RCLCPP_ERROR_STREAM(l, s) RCLCPP_INFO_STREAM(l, s)
#define RCLCPP_DEBUG_STREAM(l, s) RCLCPP_INFO_STREAM(l, s)
namespace rclcpp_action { template <class A> struct Server { typedef std::shared_ptr<Server<A>> SharedPtr; }; template <class A> struct ServerGoalHandle { bool is_active() const { return false; } bool is_canceling() const { return false; } bool is_executing() const { return false; } template <class R> void canceled(R) {} template <class R> void succeed(R) {} template <class R> void abort(R) {} template <class F> void publish_feedback(F) {} std::shared_ptr<const typename A::Goal> get_goal() const { return nullptr; } }; template <class A> struct Client { typedef std::shared_ptr<Client<A>> SharedPtr; }; }

// ------------------------------------------------------------------ pluginlib
#define PLUGINLIB_EXPORT_CLASS(a, b)
namespace pluginlib {
template <class T> struct ClassLoader {
  ClassLoader(const std::string&, const std::string&) {}
  std::shared_ptr<T> createSharedInstance(const std::string&) { return nullptr; }
  std::vector<std::string> getDeclaredClasses() { return {}; }
  bool isClassAvailable(const std::string&) { return false; }
};
struct PluginlibException : std::runtime_error { PluginlibException(const std::string& s) : std::runtime_error(s) {} };
}

// ------------------------------------------------------------------ tf2
namespace tf2 {
typedef std::chrono::nanoseconds Duration;
typedef std::chrono::time_point<std::chrono::system_clock, std::chrono::nanoseconds> TimePoint;
static const TimePoint TimePointZero = TimePoint();
inline Duration durationFromSec(double) { return Duration(); }
struct TransformException : std::runtime_error { TransformException(const std::string& s) : std::runtime_error(s) {} };
struct LookupException : TransformException { using TransformException::TransformException; };
struct ConnectivityException : TransformException { using TransformException::TransformException; };
struct ExtrapolationException : TransformException { using TransformException::TransformException; };
struct Vector3 {
  Vector3() {} Vector3(double, double, double) {}
  double x() const { return 0; } double y() const { return 0; } double z() const { return 0; }
  double getX() const { return 0; } double getY() const { return 0; } double getZ() const { return 0; }
  void setX(double) {} void setY(double) {} void setZ(double) {} void setValue(double, double, double) {}
  double length() const { return 0; } double length2() const { return 0; } double dot(const Vector3&) const { return 0; } Vector3 cross(const Vector3&) const { return *this; }
  double angle(const Vector3&) const { return 0; } double distance(const Vector3&) const { return 0; }
  Vector3 normalized() const { return *this; } Vector3& normalize() { return *this; }
  Vector3 operator-(const Vector3&) const { return *this; } Vector3 operator+(const Vector3&) const { return *this; } Vector3 operator*(double) const { return *this; } Vector3 operator/(double) const { return *this; }
  Vector3 operator-() const { return *this; }
  double operator[](int) const { return 0; } double& operator[](int) { static double d; return d; }
};
struct Quaternion {
  Quaternion() {} Quaternion(double, double, double, double) {} Quaternion(const Vector3&, double) {}
  double x() const { return 0; } double y() const { return 0; } double z() const { return 0; } double w() const { return 1; }
  double getX() const { return 0; } double getY() const { return 0; } double getZ() const { return 0; } double getW() const { return 1; }
  void setRPY(double, double, double) {} void setRotation(const Vector3&, double) {} void setValue(double, double, double, double) {}
  Quaternion inverse() const { return *this; } Quaternion& normalize() { return *this; } Quaternion normalized() const { return *this; }
  Quaternion operator*(const Quaternion&) const { return *this; } double getAngle() const { return 0; } Vector3 getAxis() const { return Vector3(); }
  double angleShortestPath(const Quaternion&) const { return 0; }
};
inline Vector3 quatRotate(const Quaternion&, const Vector3& v) { return v; }
struct Matrix3x3 {
  Matrix3x3() {} Matrix3x3(const Quaternion&) {}
  void getRPY(double&, double&, double&) const {} void getEulerYPR(double&, double&, double&) const {} void setRotation(const Quaternion&) {} void getRotation(Quaternion&) const {}
  void setRPY(double, double, double) {} void setEulerYPR(double, double, double) {} Matrix3x3 inverse() const { return *this; } Matrix3x3 transpose() const { return *this; }
  Vector3 getColumn(int) const { return Vector3(); } Vector3 getRow(int) const { return Vector3(); } Matrix3x3 operator*(const Matrix3x3&) const { return *this; } Vector3 operator*(const Vector3& v) const { return v; }
};
struct Transform {
  Transform() {} Transform(const Quaternion&) {} Transform(const Quaternion&, const Vector3&) {} Transform(const Matrix3x3&, const Vector3&) {}
  Vector3 getOrigin() const { return Vector3(); } Quaternion getRotation() const { return Quaternion(); } Matrix3x3 getBasis() const { return Matrix3x3(); }
  void setOrigin(const Vector3&) {} void setRotation(const Quaternion&) {} void setBasis(const Matrix3x3&) {} void setIdentity() {}
  Transform inverse() const { return *this; } Transform operator*(const Transform&) const { return *this; } Vector3 operator*(const Vector3& v) const { return v; }
  void mult(const Transform&, const Transform&) {} Transform inverseTimes(const Transform&) const { return *this; }
};
template <class T> struct Stamped : T { std::string frame_id_; TimePoint stamp_; };
template <class A, class B> void fromMsg(const A&, B&) {}
template <class A, class B> void convert(const A&, B&) {}
template <class A> geometry_msgs::msg::Quaternion toMsg(const A&) { return geometry_msgs::msg::Quaternion(); }
template <class A, class B> B& toMsg(const A&, B& b) { return b; }
template <class T> void doTransform(const T&, T&, const geometry_msgs::msg::TransformStamped&) {}
inline Eigen::Affine3d transformToEigen(const geometry_msgs::msg::TransformStamped&) { return Eigen::Affine3d(); }
inline Eigen::Affine3d transformToEigen(const geometry_msgs::msg::Transform&) { return Eigen::Affine3d(); }
inline geometry_msgs::msg::TransformStamped eigenToTransform(const Eigen::Affine3d&) { return geometry_msgs::msg::TransformStamped(); }
inline double getYaw(const Quaternion&) { return 0; } inline double getYaw(const geometry_msgs::msg::Quaternion&) { return 0; }
namespace impl { inline double getYaw(const Quaternion&) { return 0; } template <class Q> void getEulerYPR(const Q&, double&, double&, double&) {} }
}  // namespace tf2
namespace tf2_ros {
struct CreateTimerInterface { typedef std::shared_ptr<CreateTimerInterface> SharedPtr; };
struct CreateTimerROS : CreateTimerInterface { template <class... A> CreateTimerROS(A&&...) {} };
struct Buffer {
  Buffer(rclcpp::Clock::SharedPtr) {} template <class... A> Buffer(rclcpp::Clock::SharedPtr, A&&...) {}
  geometry_msgs::msg::TransformStamped lookupTransform(const std::string&, const std::string&, const tf2::TimePoint&) const { return geometry_msgs::msg::TransformStamped(); }
  geometry_msgs::msg::TransformStamped lookupTransform(const std::string&, const std::string&, const tf2::TimePoint&, const tf2::Duration&) const { return geometry_msgs::msg::TransformStamped(); }
  geometry_msgs::msg::TransformStamped lookupTransform(const std::string&, const std::string&, const rclcpp::Time&) const { return geometry_msgs::msg::TransformStamped(); }
  bool canTransform(const std::string&, const std::string&, const tf2::TimePoint&, std::string* = nullptr) const { return false; }
  template <class T> T& transform(const T&, T& out, const std::string&) const { return out; }
  void setCreateTimerInterface(CreateTimerInterface::SharedPtr) {} void setUsingDedicatedThread(bool) {}
};
struct TransformListener { TransformListener(Buffer&) {} template <class... A> TransformListener(Buffer&, A&&...) {} };
struct TransformBroadcaster { template <class N> TransformBroadcaster(N) {} void sendTransform(const geometry_msgs::msg::TransformStamped&) {} };
}
namespace angles {
inline double normalize_angle(double a) { return a; } inline double normalize_angle_positive(double a) { return a; }
inline double shortest_angular_distance(double, double) { return 0; } inline double from_degrees(double d) { return d; } inline double to_degrees(double r) { return r; }
}

// ------------------------------------------------------------------ PCL
namespace pcl {
struct PCLHeader { std::string frame_id; uint64_t stamp = 0; uint32_t seq = 0; };
struct alignas(16) PointXYZ { float x = 0, y = 0, z = 0, pad_ = 1; PointXYZ() {} PointXYZ(float, float, float) {} Eigen::Vector3f getVector3fMap() const { return Eigen::Vector3f(); } };
struct alignas(16) PointXYZI { float x = 0, y = 0, z = 0, pad_ = 1; float intensity = 0; float pad2_[3] = {0, 0, 0}; PointXYZI() {} PointXYZI(float) {} Eigen::Vector3f getVector3fMap() const { return Eigen::Vector3f(); } };
struct alignas(16) PointXYZRGB { float x = 0, y = 0, z = 0, pad_ = 1; uint8_t r = 0, g = 0, b = 0, a = 0; float pad2_[3] = {0, 0, 0}; };
struct alignas(16) PointNormal { float x = 0, y = 0, z = 0, pad_ = 1; float normal_x = 0, normal_y = 0, normal_z = 0, pad1_ = 0; float curvature = 0, pad2_[3] = {0, 0, 0}; };
struct alignas(16) Normal { float normal_x = 0, normal_y = 0, normal_z = 0, pad_ = 0, curvature = 0, pad2_[3] = {0, 0, 0}; };
struct PointIndices { typedef std::shared_ptr<PointIndices> Ptr; typedef std::shared_ptr<const PointIndices> ConstPtr; PCLHeader header; std::vector<int> indices; };
typedef std::vector<int> Indices;
struct ModelCoefficients { typedef std::shared_ptr<ModelCoefficients> Ptr; PCLHeader header; std::vector<float> values; };
template <class P> struct PointCloud {
  typedef P PointType; typedef std::shared_ptr<PointCloud<P>> Ptr; typedef std::shared_ptr<const PointCloud<P>> ConstPtr;
  typedef typename std::vector<P>::iterator iterator; typedef typename std::vector<P>::const_iterator const_iterator;
  PCLHeader header; std::vector<P> points; uint32_t width = 0, height = 0; bool is_dense = true;
  size_t size() const { return points.size(); } bool empty() const { return points.empty(); } void clear() { points.clear(); }
  void push_back(const P& p) { points.push_back(p); } void resize(size_t n) { points.resize(n); } void reserve(size_t n) { points.reserve(n); }
  P& operator[](size_t i) { return points[i]; } const P& operator[](size_t i) const { return points[i]; } P& at(size_t i) { return points[i]; } const P& at(size_t i) const { return points[i]; }
  iterator begin() { return points.begin(); } iterator end() { return points.end(); } const_iterator begin() const { return points.begin(); } const_iterator end() const { return points.end(); }
  P& back() { return points.back(); } P& front() { return points.front(); }
  PointCloud& operator+=(const PointCloud&) { return *this; } PointCloud operator+(const PointCloud&) const { return *this; }
  Ptr makeShared() const { return Ptr(new PointCloud<P>(*this)); }
};
template <class P> struct KdTreeFLANN {
  typedef std::shared_ptr<KdTreeFLANN<P>> Ptr;
  void setInputCloud(const typename PointCloud<P>::ConstPtr&) {} void setInputCloud(const typename PointCloud<P>::Ptr&) {}
  int nearestKSearch(const P&, int, std::vector<int>&, std::vector<float>&) const { return 0; }
  int radiusSearch(const P&, double, std::vector<int>&, std::vector<float>&, unsigned = 0) const { return 0; }
  void setEpsilon(float) {} void setSortedResults(bool) {}
};
namespace search { template <class P> struct KdTree : KdTreeFLANN<P> { typedef std::shared_ptr<KdTree<P>> Ptr; }; template <class P> struct Search { typedef std::shared_ptr<Search<P>> Ptr; }; }
template <class P> struct Filter {
  void setInputCloud(const typename PointCloud<P>::ConstPtr&) {} void setInputCloud(const typename PointCloud<P>::Ptr&) {}
  void filter(PointCloud<P>&) {} void filter(std::vector<int>&) {} void setNegative(bool) {} void setIndices(const PointIndices::Ptr&) {} void setIndices(const std::shared_ptr<std::vector<int>>&) {} void setKeepOrganized(bool) {}
};
template <class P> struct PassThrough : Filter<P> { void setFilterFieldName(const std::string&) {} void setFilterLimits(float, float) {} void setFilterLimitsNegative(bool) {} };
template <class P> struct VoxelGrid : Filter<P> { void setLeafSize(float, float, float) {} void setDownsampleAllData(bool) {} void setMinimumPointsNumberPerVoxel(unsigned) {} };
template <class P> struct ExtractIndices : Filter<P> {};
template <class P> struct ProjectInliers : Filter<P> { void setModelType(int) {} void setModelCoefficients(const ModelCoefficients::Ptr&) {} void setCopyAllData(bool) {} };
template <class P> struct EuclideanClusterExtraction {
  void setClusterTolerance(double) {} void setMinClusterSize(int) {} void setMaxClusterSize(int) {}
  void setSearchMethod(const typename search::KdTree<P>::Ptr&) {} void setInputCloud(const typename PointCloud<P>::ConstPtr&) {} void setInputCloud(const typename PointCloud<P>::Ptr&) {}
  void extract(std::vector<PointIndices>&) {}
};
template <class P> struct SACSegmentation {
  void setOptimizeCoefficients(bool) {} void setModelType(int) {} void setMethodType(int) {} void setMaxIterations(int) {} void setDistanceThreshold(double) {} void setAxis(const Eigen::Vector3f&) {} void setEpsAngle(double) {}
  void setInputCloud(const typename PointCloud<P>::ConstPtr&) {} void setInputCloud(const typename PointCloud<P>::Ptr&) {} void segment(PointIndices&, ModelCoefficients&) {}
};
template <class P, class N> struct NormalEstimation { void setInputCloud(const typename PointCloud<P>::ConstPtr&) {} void setSearchMethod(const typename search::KdTree<P>::Ptr&) {} void setRadiusSearch(double) {} void setKSearch(int) {} void compute(PointCloud<N>&) {} };
enum { SACMODEL_PLANE = 0, SACMODEL_PERPENDICULAR_PLANE = 1, SACMODEL_PARALLEL_PLANE = 2, SAC_RANSAC = 0 };
template <class P, class T> void transformPointCloud(const PointCloud<P>&, PointCloud<P>&, const T&) {}
template <class A, class B> void copyPointCloud(const PointCloud<A>&, PointCloud<B>&) {}
template <class A> void copyPointCloud(const PointCloud<A>&, const std::vector<int>&, PointCloud<A>&) {}
template <class A> void copyPointCloud(const PointCloud<A>&, const PointIndices&, PointCloud<A>&) {}
template <class P> void getMinMax3D(const PointCloud<P>&, P&, P&) {}
template <class P> void getMinMax3D(const PointCloud<P>&, Eigen::Vector4f&, Eigen::Vector4f&) {}
template <class P> void fromROSMsg(const sensor_msgs::msg::PointCloud2&, PointCloud<P>&) {}
template <class P> void toROSMsg(const PointCloud<P>&, sensor_msgs::msg::PointCloud2&) {}
template <class P> unsigned compute3DCentroid(const PointCloud<P>&, Eigen::Vector4f&) { return 0; }
namespace geometry { template <class A, class B> float distance(const A&, const B&) { return 0; } template <class A, class B> float squaredDistance(const A&, const B&) { return 0; } }
template <class A, class B> float euclideanDistance(const A&, const B&) { return 0; }
}  // namespace pcl
namespace pcl_conversions { template <class A, class B> void toPCL(const A&, B&) {} template <class A, class B> void fromPCL(const A&, B&) {} }
