#pragma once
#include <iostream>
struct XmNullLog { template <class T> XmNullLog& operator<<(const T&) { return *this; } };
#define LOG(x) XmNullLog()
#define VLOG(x) XmNullLog()
#define LOG_IF(x, c) XmNullLog()
#define CHECK(c) XmNullLog()
#define DCHECK(c) XmNullLog()
#define CHECK_EQ(a, b) XmNullLog()
#define CHECK_NE(a, b) XmNullLog()
#define CHECK_GT(a, b) XmNullLog()
#define CHECK_GE(a, b) XmNullLog()
#define CHECK_LT(a, b) XmNullLog()
#define CHECK_LE(a, b) XmNullLog()
#define CHECK_NOTNULL(p) (p)
namespace google {
enum { GLOG_INFO = 0, GLOG_WARNING = 1, GLOG_ERROR = 2, GLOG_FATAL = 3 };
struct LogMessage {
  LogMessage(const char*, int, int = 0) {}
  std::ostream& stream() { return std::cerr; }
};
}  // namespace google
