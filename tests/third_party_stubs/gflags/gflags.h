#pragma once
#include <cstdint>
#include <string>
#define DECLARE_bool(n) extern bool FLAGS_##n
#define DECLARE_int32(n) extern int32_t FLAGS_##n
#define DECLARE_uint32(n) extern uint32_t FLAGS_##n
#define DECLARE_int64(n) extern int64_t FLAGS_##n
#define DECLARE_uint64(n) extern uint64_t FLAGS_##n
#define DECLARE_double(n) extern double FLAGS_##n
#define DECLARE_string(n) extern std::string FLAGS_##n
